"""Round 6 exploration behind test_bf16_engine_trains_like_the_fp32_engine_at_the_shard_size: per-tensor cosine of the bf16 engine's
gradient with the fp32 engine's at batch 128, beside two yardsticks on the same weights -- fp32 F(4x4,3x3) against fp32 F(2x2,3x3)
(two fp32 roundings of the same batch) and fp32 batch A against fp32 batch B (minibatch sampling noise) -- and the 30-step loss
trajectories of both engines.  usage: python scripts/probes/bf16_vs_f32_gradients.py [batch = 128]"""
import importlib.util, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from l3embedding_amd import _lib
from oracle import l3_oracle as o
spec = importlib.util.spec_from_file_location('make_golden', os.path.join(HERE, '..', '..', 'tests', 'golden', 'make_golden.py'))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
mt, B = 'cnn_L3_melspec2', int(sys.argv[1]) if len(sys.argv) > 1 else 128
P = mod.perturbed_params(mt, 101)
P['dense_2/kernel'] = (P['dense_2/kernel'] / np.float32(64)).astype(np.float32)
batches = [o.synthetic_batch(B, seed=500 + k) for k in range(4)]


def grads(dtype, conv, batch):
    e = _lib.Engine(mt, B, seed=0, dtype=dtype, fp32_conv=conv)
    e.set_params(P)
    v, a, l = batch
    e.upload_batch(v, a, l)
    e.step_forward(True)
    for b in range(1, e.bucket_count()):
        e.step_backward_bucket(b)
    e.sync()
    g = e.get_grads()
    e.step_update(0.0, 1.0)
    e.close()
    out = {}
    for n, x in g.items():
        x = x.astype(np.float64).ravel()
        if n.endswith('/kernel'):
            x = x - 2 * o.L2_WEIGHT * P[n].astype(np.float64).ravel()
        out[n] = x
    return out


def cos(a, b):
    na, nb = np.linalg.norm(a), np.linalg.norm(b)
    return float(a @ b / (na * nb)) if na > 0 and nb > 0 else float('nan')


g32 = grads('f32', 'f4x4', batches[0])
g16 = grads('bf16', 'f4x4', batches[0])
g22 = grads('f32', 'f2x2', batches[0])
g32b = grads('f32', 'f4x4', batches[1])
g16b = grads('bf16', 'f4x4', batches[1])
print('%-54s %9s %8s %8s %8s %8s' % ('tensor', 'numel', 'bf16', 'f2x2', 'batchB', 'bf16:B'))
for n in sorted(g32, key=lambda n: cos(g32[n], g16[n]) if g32[n].size > 1 else 9):
    print('%-54s %9d %8.4f %8.4f %8.4f %8.4f' % (n, g32[n].size, cos(g32[n], g16[n]), cos(g32[n], g22[n]), cos(g32[n], g32b[n]), cos(g16[n], g16b[n])))
allv = lambda g: np.concatenate([g[n] for n in sorted(g)])
print('whole gradient: bf16 %.4f  f2x2 %.4f  batch B %.4f' % (cos(allv(g32), allv(g16)), cos(allv(g32), allv(g22)), cos(allv(g32), allv(g32b))))
losses = {}
for dt in ('f32', 'bf16'):
    e = _lib.Engine(mt, B, seed=0, dtype=dt)
    e.set_params(P)
    losses[dt] = [e.train_step(*batches[k % 4], 1e-4)[0] for k in range(30)]
    e.close()
for k in range(30):
    print('step %2d  f32 %.5f  bf16 %.5f  rel %.3e' % (k, losses['f32'][k], losses['bf16'][k], abs(losses['bf16'][k] - losses['f32'][k]) / abs(losses['f32'][k])))
