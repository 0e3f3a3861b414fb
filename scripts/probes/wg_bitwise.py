"""Bitwise fingerprint of the fp32 weight gradient of every ledger layer (batch 2 and 16, each launch repeated): one line of
hashes per layer.  Run once per build (L3_DEBUG_KNOBS=1 L3_LIB_PATH=...) and diff the outputs -- two builds that issue the same fma
chain per element must agree bit for bit; repeats within a run must agree too (a hazard shows as a mismatch)."""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from l3embedding_amd import _lib
CONVS = [('A.conv1b', 256, 199, 64, 64), ('A.conv2a', 128, 99, 64, 128), ('A.conv2b', 128, 99, 128, 128), ('A.conv3a', 64, 49, 128, 256),
         ('A.conv3b', 64, 49, 256, 256), ('A.conv4a', 32, 24, 256, 512), ('A.conv4b', 32, 24, 512, 512),
         ('V.conv1b', 224, 224, 64, 64), ('V.conv2a', 112, 112, 64, 128), ('V.conv2b', 112, 112, 128, 128), ('V.conv3a', 56, 56, 128, 256),
         ('V.conv3b', 56, 56, 256, 256), ('V.conv4a', 28, 28, 256, 512), ('V.conv4b', 28, 28, 512, 512)]
for tag, h, w, ci, co in CONVS:
    out = []
    for n in (2, 16):
        rng = np.random.RandomState(h + ci + n)
        x = np.maximum(rng.randn(n, h, w, ci), 0).astype(np.float32)
        wt = (rng.randn(3, 3, ci, co) * 0.05).astype(np.float32)
        dy = (rng.randn(n, h, w, co) * 1e-3).astype(np.float32)
        hs = set()
        for rep in range(3):
            dx, dw, db = _lib.op_conv2d_bwd(x, wt, dy, True)
            hs.add(hashlib.sha256(np.ascontiguousarray(dw).tobytes()).hexdigest()[:12])
        out.append('n%d:%s' % (n, '|'.join(sorted(hs))))
    print(tag, ' '.join(out), flush=True)
