"""bf16-operand conv kernels vs the oracle's emulation (dev check): python scripts/dev_bf16.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from l3embedding_amd import _lib
from oracle import l3_oracle as o

def relerr(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / (np.abs(b).max() + 1e-30))

shapes = [(1, 17, 13, 64, 128), (1, 8, 8, 128, 256), (1, 6, 6, 256, 512), (2, 5, 5, 64, 64), (1, 33, 31, 64, 64),
          (2, 28, 28, 64, 64), (3, 32, 24, 128, 192), (1, 1, 1, 64, 64), (1, 2, 70, 64, 64)]
for shp in shapes:
    n, h, w, ci, co = shp
    rng = np.random.RandomState(sum(shp))
    x = rng.randn(n, h, w, ci).astype(np.float32)
    wt = (rng.randn(3, 3, ci, co) / np.sqrt(9 * ci)).astype(np.float32)
    b = rng.randn(co).astype(np.float32)
    dy = rng.randn(n, h, w, co).astype(np.float32)
    with o.mixed_precision('bf16'):
        ref = o.conv2d_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64), 'same')
        dx_ref, dw_ref, db_ref = o.conv2d_bwd(x.astype(np.float64), wt.astype(np.float64), dy.astype(np.float64), 'same')
    ref32 = o.conv2d_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64), 'same')
    y = _lib.op_conv2d_fwd(x, wt, b, True, dtype='bf16')
    dx, dw, db = _lib.op_conv2d_bwd(x, wt, dy, True, dtype='bf16')
    print(shp, 'fwd %.2e' % relerr(y, ref), 'dx %.2e' % relerr(dx, dx_ref), 'dw %.2e' % relerr(dw, dw_ref),
          '| bf16-vs-fp32 oracle %.2e' % relerr(ref, ref32), flush=True)
