import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from l3embedding_amd import _lib
rng = np.random.RandomState(0)
for ci in (8, 16, 64):
    x = rng.randn(16, 224, 224, ci).astype(np.float32)
    w = rng.randn(3, 3, ci, 64).astype(np.float32)
    b = np.zeros(64, np.float32)
    for _ in range(3):
        _lib.op_conv2d_fwd(x, w, b, True)
