"""Developer probe: phase times inside conv_wino4_kernel (an instrumented build from scripts/probes/w4_instrument.py).
usage: python scripts/probes/w4_timing.py <libl3hip_timing.so>"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from l3embedding_amd import _lib
_lib.lib_path = lambda: os.path.abspath(sys.argv[1])
lib = _lib.load()
lib.l3_debug_w4_timing.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 16)()
N = 64
phases = [(9, 'top barrier'), (10, 'set-up'), (0, 'first loads landed (0 when prefetched)'), (11, 'first transform + barrier'), (1, 'stage loop'),
          (4, 'next block requested (waves 8-11)'), (2, 'exchange writes + barriers, 4 rounds'), (3, 'output transform, 4 rounds'),
          (6, 'statistics / end')]
for tag, h, w, ci, co in [('V.1b', 224, 224, 64, 64), ('V.2b', 112, 112, 128, 128), ('V.3b', 56, 56, 256, 256), ('V.4b', 28, 28, 512, 512)]:
    x = np.ones((N, h, w, ci), np.float32)
    wt = np.ones((3, 3, ci, co), np.float32) * 0.01
    b = np.zeros(co, np.float32)
    _lib.op_conv2d_fwd(x, wt, b, True)
    lib.l3_debug_w4_timing(buf, 1)
    _lib.op_conv2d_fwd(x, wt, b, True)
    lib.l3_debug_w4_timing(buf, 1)
    v = [int(buf[i]) for i in range(16)]
    blocks, stages = v[7], v[8]
    tot = sum(v[i] for i, _ in phases)
    print('%s: %d tile blocks, %.0f stages each, %.0f cycles per tile block' % (tag, blocks, stages / blocks, tot / blocks))
    for i, name in phases:
        print('   %-28s %9.0f cycles/block  %5.1f %%' % (name, v[i] / blocks, 100.0 * v[i] / tot))
    print('   stage loop per stage: %.0f cycles (the MFMAs alone: 4608)' % (v[1] / stages))
