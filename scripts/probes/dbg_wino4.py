"""Developer probe: where does the F(4x4,3x3) forward differ from a float64 direct convolution?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from l3embedding_amd import _lib
rng = np.random.RandomState(0)
N, H, W, Ci, Co = 2, 16, 16, 64, 64
x = rng.randn(N, H, W, Ci).astype(np.float32)
w = (rng.randn(3, 3, Ci, Co) * 0.05).astype(np.float32)
b = np.zeros(Co, np.float32)
y = _lib.op_conv2d_fwd(x, w, b, True)
xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
ref = np.zeros((N, H, W, Co))
for kh in range(3):
    for kw in range(3):
        ref += np.einsum('nhwc,ck->nhwk', xp[:, kh:kh + H, kw:kw + W], w[kh, kw].astype(np.float64))
err = np.abs(y - ref)
print('max err', err.max(), 'ref max', np.abs(ref).max())
bad = err > 1e-3
print('bad fraction', bad.mean())
print('bad by sample', bad.mean(axis=(1, 2, 3)))
print('bad by channel (first 64):', ''.join('X' if v > 0 else '.' for v in bad.mean(axis=(0, 1, 2))))
print('bad by y:', ''.join('X' if v > 0 else '.' for v in bad.mean(axis=(0, 2, 3))))
print('bad by x:', ''.join('X' if v > 0 else '.' for v in bad.mean(axis=(0, 1, 3))))
# does y match ref at a permuted tile?
T = lambda a, n, ty, tx: a[n, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4, :]
for n in range(N):
    for ty in range(H // 4):
        for tx in range(W // 4):
            got = T(y, n, ty, tx)
            best = None
            for n2 in range(N):
                for ty2 in range(H // 4):
                    for tx2 in range(W // 4):
                        d = np.abs(got - T(ref, n2, ty2, tx2)).max()
                        if best is None or d < best[0]:
                            best = (d, n2, ty2, tx2)
            if best[1:] != (n, ty, tx) or best[0] > 1e-3:
                print('tile', (n, ty, tx), 'best match', best)
