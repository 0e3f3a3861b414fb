#!/usr/bin/env python
"""bench.py -- AVC training pairs/sec of the MI355X-native L3-Net training step.

A "step" is one full training step (forward + backward + Adam + BN moving update, and
the bucketed RCCL gradient all-reduce when N > 1) of cnn_L3_melspec2 over one synthetic
batch of 64 pairs per GPU (BASELINE.json configs[2] at N=1, configs[3] at N=8), with the
inputs already resident in HBM.  fp32 throughout (fp32 MFMA; forward/dgrad as Winograd F(2x2,3x3)).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      dominant kernel (conv implicit-GEMM, forward+dgrad launches) timed with
                hipEvents on the stream each launch goes to.  The timed region runs the two
                towers on two streams (their kernels overlap, so a launch's begin-to-end time
                there includes the other tower's work: "roofline_in_timed_region"); the
                "roofline" object is the same launches timed over --roofline-steps further
                steps with the towers serialised on one stream, which is what a per-kernel
                duration means and what profiles/*kernel_stats.csv holds
  cpu_baseline  the same fp32 step in PyTorch-CPU/oneDNN (`oracle/torch_cpu.py`) and, as
                cpu_baseline_numpy, the NumPy oracle -- bounded samples on the host cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

F_TRAIN_GFLOP_PER_PAIR = 124.5519      # 3 x (convs + head) + DFT + mel
PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md chip table
PEAK_BF16_MFMA_TFLOPS = 2500.0         # dense bf16 MFMA (same table)


def synthetic_raw(batch, seed, rank):
    """int16 PCM U{-32768..32767}, uint8 frames U{0..255}, Bernoulli(0.5) labels."""
    rng = np.random.RandomState(seed + rank)
    pcm = rng.randint(-32768, 32768, size=(batch, 1, 48000)).astype(np.int16)
    frm = rng.randint(0, 256, size=(batch, 224, 224, 3)).astype(np.uint8)
    lab = rng.randint(0, 2, size=(batch,))
    labels = np.stack([lab, 1 - lab], axis=1).astype(np.int32)
    return frm, pcm, labels


def cpu_baseline(model_type, budget_s=12.0):
    """CPU stand-ins for the reference's (not installable) Keras/TF-1.4 CPU path, timed on this box's
    host cores on a bounded sample: the identical fp32 training step in PyTorch-CPU/oneDNN
    (`oracle/torch_cpu.py`, SURVEY 8(d) stand-in (ii), the stronger baseline -> `cpu_baseline`) and the
    NumPy oracle (`cpu_baseline_numpy`)."""
    import torch
    from oracle import l3_oracle as o
    from oracle.torch_cpu import TorchCpuTrainer
    P = o.init_params(model_type, seed=20180123)
    B = 8
    v, a, l = o.synthetic_batch(B)
    tr = TorchCpuTrainer(model_type, P)
    tr.step(v[:2], a[:2], l[:2], 1e-4)                       # warm-up (thread pools, oneDNN primitives)
    # oneDNN on a 2-socket host does not scale to all cores at this batch: take the best thread count
    default_threads = torch.get_num_threads()
    sweep = {}
    for nt in sorted(set(t for t in (8, 16, 32, default_threads) if t <= (os.cpu_count() or 1))):
        torch.set_num_threads(nt)
        tr.step(v, a, l, 1e-4)
        t0 = time.time()
        tr.step(v, a, l, 1e-4)
        sweep[nt] = time.time() - t0
        if sweep[nt] > 6.0:                                   # more threads only get slower from here
            break
    best_nt = min(sweep, key=sweep.get)
    torch.set_num_threads(best_nt)
    times = [sweep[best_nt]]
    t_start = time.time()
    while len(times) < 3 or (time.time() - t_start < budget_s and len(times) < 5):
        t0 = time.time()
        tr.step(v, a, l, 1e-4)
        times.append(time.time() - t0)
    med = float(np.median(times))
    torch.set_num_threads(default_threads)
    torch_cpu = {"value": B / med, "unit": "pairs/s", "cores": best_nt, "host_cores": os.cpu_count(), "kind": "port",
                 "thread_sweep_s_per_step": {str(k): round(val, 3) for k, val in sweep.items()},
                 "sample": "%d fp32 training steps (fwd+bwd+Adam) of %s at batch %d in PyTorch-CPU/oneDNN "
                           "(oracle/torch_cpu.py) with the best of %s threads (%d), median %.2f s/step; the reference "
                           "Keras/TF-1.4 CPU path is not installable, and oneDNN is a stronger CPU baseline than "
                           "TF-1.4 Eigen" % (len(times), model_type, B, sorted(sweep), best_nt, med)}
    B2 = 2
    adam, bn = o.AdamState(), o.BNMovingState()
    o.train_step(model_type, P, adam, bn, v[:1], a[:1], l[:1], 1e-4, np.float32)   # warm-up (BLAS init)
    times = []
    t_start = time.time()
    while len(times) < 1 or (time.time() - t_start < budget_s and len(times) < 3):
        t0 = time.time()
        o.train_step(model_type, P, adam, bn, v[:B2], a[:B2], l[:B2], 1e-4, np.float32)
        times.append(time.time() - t0)
    med = float(np.median(times))
    numpy_cpu = {"value": B2 / med, "unit": "pairs/s", "cores": os.cpu_count(), "kind": "port",
                 "sample": "%d fp32 training steps of %s at batch %d with the NumPy oracle (OpenBLAS threads), "
                           "median %.2f s/step" % (len(times), model_type, B2, med)}
    return torch_cpu, numpy_cpu


def tower_bench(args, eng, B, world, rank, dist):
    """One sub-network alone (replicas only: nothing is exchanged between ranks)."""
    import torch
    tower = 'audio' if args.workload == 'audio_tower' else 'vision'
    peak = PEAK_FP32_MFMA_TFLOPS if args.dtype == 'f32' else PEAK_BF16_MFMA_TFLOPS

    def run(backward, steps):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.tower_step(tower, backward)
        eng.sync()
        if dist is not None:
            dist.barrier()
        return time.perf_counter() - t0

    run(True, args.warmup)
    t_fwd = run(False, args.steps)
    eng.set_tower_overlap(False)
    eng.profile_enable(True)
    elapsed = run(True, args.steps)
    prof = eng.profile_read()
    if dist is not None:
        t = torch.tensor([elapsed, t_fwd], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, t_fwd = float(t[0].item()), float(t[1].item())
    if rank == 0:
        ig_ms = prof['conv_fwd']['ms'] + prof['conv_dgrad']['ms']
        ig_fl = prof['conv_fwd']['flops'] + prof['conv_dgrad']['flops']
        achieved = ig_fl / (ig_ms * 1e-3) / 1e12 if ig_ms > 0 else 0.0
        print(json.dumps({
            "metric": "%s-tower samples/sec, training-mode forward + backward (stand-in loss = mean of the tower output, "
                      "no optimizer step)" % tower,
            "value": B * world * args.steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "cnn_L3_melspec2 %s tower only (%s), batch %d per GPU, inputs resident in HBM" %
                                   (tower, "mel front-end + audio conv kernels" if tower == 'audio' else "vision conv kernels", B),
                       "parallelism": "replicas%d" % world},
            "forward_only": {"value": B * world * args.steps / t_fwd, "unit": "samples/s", "ms_per_step": 1e3 * t_fwd / args.steps},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": None, "kernel": "forward + dgrad convolution launches of the tower (algorithmic flops)",
                         "measured": "the timed fwd+bwd region (one stream)"},
            "kernel_ms_per_step": {k: v['ms'] / args.steps for k, v in prof.items()},
            "cpu_baseline": None}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch-per-gpu', type=int, default=64)
    ap.add_argument('--model', default='cnn_L3_melspec2')
    ap.add_argument('--lr', type=float, default=1e-4)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', default='full', choices=['full', 'audio_tower', 'vision_tower'],
                    help="full: the AVC training step (the metric).  audio_tower / vision_tower: one sub-network alone, "
                         "training-mode forward + backward from mean(output), no optimizer step -- SURVEY 8(d) config 2 "
                         "(BASELINE configs[1]); reported as its own line, never as the AVC metric")
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help="f32: the headline configuration (BASELINE.json configs[2]/[3]); bf16: mixed precision of "
                         "configs[4] (bf16 conv operands, fp32 accumulate) -- reported as its own line, never as the "
                         "fp32 metric")
    ap.add_argument('--roofline-steps', type=int, default=5,
                    help='extra, untimed-for-value steps with the towers serialised, for per-kernel durations')
    ap.add_argument('--serial', action='store_true', help='run the timed region with the towers serialised too')
    args = ap.parse_args()

    import torch
    from l3embedding_amd import _lib
    from l3embedding_amd.training_utils import DataParallelTrainer

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d for --gpus %d' % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an AMD GPU (no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl', rank=rank, world_size=world,
                                device_id=torch.device('cuda', local_rank))

    B = args.batch_per_gpu
    # a real side stream shared by the engine's kernels and (N > 1) the collectives' dependencies
    tstream = torch.cuda.Stream(device=local_rank)
    assert tstream.cuda_stream != 0
    eng = _lib.Engine(args.model, B, device=local_rank, global_batch=B * world, seed=20180123,
                      stream=tstream.cuda_stream, dtype=args.dtype)
    frm, pcm, lab = synthetic_raw(B, 20180123, rank)
    eng.upload_batch_raw(frm, pcm, lab)          # uint8/int16 -> fp32 on the GPU (train.py:186,189)
    trainer = DataParallelTrainer(eng, local_rank, world, rank, stream=tstream)
    if args.workload != 'full':
        return tower_bench(args, eng, B, world, rank, dist)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.serial:
        eng.set_tower_overlap(False)
    for _ in range(args.warmup):
        trainer.step(args.lr)
    eng.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step(args.lr)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss, acc = eng.step_results()
    prof_region = eng.profile_read()
    # per-kernel durations: same launches, towers serialised on the engine's stream
    prof = prof_region
    if not args.serial and args.roofline_steps > 0:
        eng.set_tower_overlap(False)
        eng.profile_enable(True)
        for _ in range(args.roofline_steps):
            trainer.step(args.lr)
        barrier()
        eng.sync()
        prof = eng.profile_read()
        eng.set_tower_overlap(True)

    if rank == 0:
        traffic = None
        tpath = os.path.join(HERE, 'profiles', 'pmc_traffic.json')     # committed PMC pass (scripts/pmc_conv.sh)
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath))['hbm_bytes_per_launch']
            except Exception:
                traffic = None
        peak = PEAK_FP32_MFMA_TFLOPS if args.dtype == 'f32' else PEAK_BF16_MFMA_TFLOPS
        pairs = B * world * args.steps
        value = pairs / elapsed
        def igemm(pr):
            ms = pr['conv_fwd']['ms'] + pr['conv_dgrad']['ms']
            fl = pr['conv_fwd']['flops'] + pr['conv_dgrad']['flops']
            ex = pr['conv_fwd']['executed_flops'] + pr['conv_dgrad']['executed_flops']
            n = pr['conv_fwd']['launches'] + pr['conv_dgrad']['launches']
            return ms, fl, n, (fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0), (ex / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)
        ig_ms, ig_fl, ig_n, achieved, executed = igemm(prof)
        r_ms, r_fl, r_n, r_achieved, _ = igemm(prof_region)
        wg = prof['conv_wgrad']
        wg_tf = wg['flops'] / (wg['ms'] * 1e-3) / 1e12 if wg['ms'] > 0 else 0.0
        prof_steps = args.steps if prof is prof_region else args.roofline_steps
        out = {
            "metric": "AVC training pairs/sec (1s audio + 224x224 frame)",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.dtype == 'f32' else "bf16 conv operands / f32 accumulate (everything else f32)",
            "data": "synthetic",
            "config": {"workload": "full %s AVC training step (audio+vision+fusion, fwd+bwd+Adam), batch %d per GPU, "
                                   "global batch %d, %s, inputs resident in HBM" %
                                   (args.model, B, B * world, "fp32" if args.dtype == 'f32' else "bf16 mixed precision"),
                       "global_batch": B * world, "parallelism": "dp%d" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic if args.dtype == 'f32' else None,
                         "kernel": ("conv_wino_kernel (Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32; forward + dgrad "
                                    "launches, incl. the two direct first-layer launches per pass)") if args.dtype == 'f32'
                                   else "conv_igemm_bf16_kernel (direct, v_mfma_f32_32x32x16_bf16; forward + dgrad launches)",
                         "note": "achieved counts ALGORITHMIC flops (direct convolution, SURVEY 8d); Winograd issues "
                                 "2.25x fewer, so frac may exceed 1 -- mfma_utilization is issued flops / peak",
                         "executed": executed, "mfma_utilization": executed / peak,
                         "launches": ig_n, "avg_launch_ms": ig_ms / ig_n if ig_n else None,
                         "alg_flop_per_launch": ig_fl / ig_n if ig_n else None,
                         "measured": ("timed region (towers serialised)" if prof is prof_region else
                                      "%d further steps with the towers serialised on one stream" % args.roofline_steps)},
            "roofline_in_timed_region": {"achieved": r_achieved, "frac": r_achieved / peak,
                                         "unit": "TFLOP/s", "launches": r_n,
                                         "avg_launch_ms": r_ms / r_n if r_n else None,
                                         "note": "towers overlap on two streams: durations include the other tower's kernels"},
            "tower_overlap": not args.serial,
            "wgrad": {"kernel": "conv_wgrad9t_kernel (direct, %s)" %
                                ("v_mfma_f32_32x32x2_f32" if args.dtype == 'f32' else "v_mfma_f32_32x32x16_bf16"),
                      "achieved": wg_tf, "frac": wg_tf / peak, "unit": "TFLOP/s", "ms_per_step": wg['ms'] / prof_steps},
            "step_fraction_of_mfma_peak": value / world * F_TRAIN_GFLOP_PER_PAIR * 1e9 / (peak * 1e12),
            "kernel_ms_per_step": {k: v['ms'] / prof_steps for k, v in prof.items()},
            "final_loss": loss,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["cpu_baseline_numpy"] = cpu_baseline(args.model)
            out["x_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            out["x_cpu_baseline_numpy"] = value / out["cpu_baseline_numpy"]["value"]
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == '__main__':
    main()
