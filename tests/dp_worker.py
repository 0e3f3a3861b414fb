"""Worker of test_parity_gpu.py::test_dp_world2_one_gpu_gloo: one rank of a 2-rank data-parallel run,
both ranks on GPU 0, torch.distributed backend gloo (it all-reduces CUDA tensors through the host), so
that the REAL engine + DataParallelTrainer bucket protocol runs with world_size 2 on a one-GPU box."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    out_dir, steps = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    import torch
    import torch.distributed as dist
    from l3embedding_amd import _lib
    from l3embedding_amd.training_utils import DataParallelTrainer, get_slice_bounds
    from oracle import l3_oracle as o
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    mt, GB = 'tiny_L3', 6
    v, a, l = o.synthetic_batch(GB, seed=31)
    lo, hi = get_slice_bounds(GB, world, rank)
    ts = torch.cuda.Stream(device=0)
    eng = _lib.Engine(mt, hi - lo, seed=13, stream=ts.cuda_stream, global_batch=GB)
    tr = DataParallelTrainer(eng, 0, world, rank, stream=ts)
    losses = []
    for _ in range(steps):
        eng.upload_batch(v[lo:hi], a[lo:hi], l[lo:hi])
        tr.step(1e-3)
        losses.append(eng.step_results()[0])
    W = eng.get_params()
    # every tensor, the BatchNorm moving statistics included: each rank applied BOTH replicas' updates (l3_config.dp_moving)
    # -- and the same validation rows through them, whichever rank evaluates (inference mode)
    ev = _lib.Engine(mt, 3, seed=1, stream=ts.cuda_stream)
    ev.copy_state_from(eng)
    _, val_logits = ev.forward(v[:3], a[:3], training=False)
    ev.close()
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), losses=np.asarray(losses), val_logits=val_logits,
             bn_updates=np.asarray(eng.optimizer_steps()), **{k.replace('/', '|'): W[k] for k in W})
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


if __name__ == '__main__':
    main()
