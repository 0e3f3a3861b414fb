"""Cost of the data-parallel step path at world size 1: l3_step_dp (per-bucket events + ncclAllReduce on the
communicator stream, an identity with one rank) against l3_step_resident, alternating in ONE process so
that clock / power state is shared.  usage: python scripts/dp_overhead.py [batch] [dtype]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch  # noqa: F401  (first: one HIP runtime in the process)
from l3embedding_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dtype = sys.argv[2] if len(sys.argv) > 2 else 'f32'
rng = np.random.RandomState(0)
eng = _lib.Engine('cnn_L3_melspec2', B, global_batch=B, dtype=dtype)
lab = rng.randint(0, 2, B)
eng.upload_batch_raw(rng.randint(0, 256, (B, 224, 224, 3)).astype(np.uint8), rng.randint(-32768, 32768, (B, 1, 48000)).astype(np.int16),
                     np.stack([lab, 1 - lab], 1).astype(np.int32))
eng.comm_init(_lib.comm_unique_id(), 1, 0)
def run(fn, n=20):
    eng.sync(); t0 = time.perf_counter()
    for _ in range(n):
        fn(1e-4)
    eng.sync(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(5):
    eng.step_resident(1e-4)
res, dp = [], []
for rep in range(4):
    res.append(run(eng.step_resident)); dp.append(run(eng.step_dp))
print('resident ms/step', ['%.2f' % t for t in res], ' dp ms/step', ['%.2f' % t for t in dp])
print('median resident %.2f ms, dp %.2f ms, overhead %.2f %%' % (np.median(res), np.median(dp), 100 * (np.median(dp) / np.median(res) - 1)))
