import numpy as np, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from l3embedding_amd import _lib
rng = np.random.RandomState(1)
n, h, w, ci = 16, 256, 199, 2
x = rng.randn(n, h, w, ci).astype(np.float32); x[..., -1] = 1.0
dy = (rng.randn(n, h, w, 64) * 1e-3).astype(np.float32)
wt = np.zeros((3, 3, ci, 64), np.float32)
# float64 reference by einsum per tap
xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
ref = np.zeros((3, 3, ci, 64))
d64 = dy.astype(np.float64)
for kh in range(3):
    for kw in range(3):
        ref[kh, kw] = np.einsum('nhwc,nhwk->ck', xp[:, kh:kh + h, kw:kw + w], d64)
for flag in ('1', '0'):
    os.environ['L3_FIRST_WGRAD'] = flag
    _, dw, _ = _lib.op_conv2d_bwd(x, wt, dy, True)
    print('L3_FIRST_WGRAD=%s  max|err|/max|ref| = %.3e   rel L2 = %.3e' % (flag, np.abs(dw - ref).max() / np.abs(ref).max(),
          np.sqrt(((dw - ref) ** 2).sum() / (ref ** 2).sum())))
