import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from l3embedding_amd import _lib
for (n, h, w, ci, co) in [(16, 28, 28, 512, 512), (16, 56, 56, 128, 256), (8, 112, 112, 64, 128)]:
    rng = np.random.RandomState(n + h)
    x = rng.randn(n, h, w, ci).astype(np.float32)
    wt = (rng.randn(3, 3, ci, co) / np.sqrt(9 * ci)).astype(np.float32)
    dy = rng.randn(n, h, w, co).astype(np.float32)
    dx1, dw1, db1 = _lib.op_conv2d_bwd(x, wt, dy, True)
    R = 3
    x3 = np.concatenate([x] * R, 0); dy3 = np.concatenate([dy] * R, 0) / np.float32(R)
    dx3, dw3, db3 = _lib.op_conv2d_bwd(x3, wt, dy3, True)
    print((n, h, w, ci, co), 'dw rel', np.abs(dw3 - dw1).max() / np.abs(dw1).max(), 'dx rel', np.abs(dx3[:n] * R - dx1).max() / np.abs(dx1).max())
