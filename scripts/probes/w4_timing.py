"""Developer probe: phase times inside conv_wino4_kernel (a -DW4_TIMING build; s_memtime stamps by one wave of every tile block).
usage: python scripts/probes/w4_timing.py <libl3hip_timing.so>"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from l3embedding_amd import _lib
_lib.lib_path = lambda: os.path.abspath(sys.argv[1])
lib = _lib.load()
lib.l3_debug_w4_timing.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 16)()
extra = {10: 'first prep', 0: 'prologue', 11: 'loop: half 0', 12: 'loop: half 1', 13: 'loop: barrier'}
N = 64
names = ['prologue', 'stage loop', 'E write 0', 'transform 0', 'bar + E write 1', 'transform 1', 'stats/end', 'blocks', 'stages']
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
    tot = sum(v[:7]) + v[9] + v[10]
    print('%s: blocks %d, stages/block %.1f, ticks/block %.0f' % (tag, blocks, stages / blocks, tot / blocks))
    for i in range(7):
        print('   %-16s %10.1f ticks/block  %5.1f %%' % (names[i], v[i] / blocks, 100.0 * v[i] / tot))
    print('   per stage: %.1f ticks' % (v[1] / stages))
    for i in (10, 0):
        print('   [%-14s] %10.1f ticks/block' % (extra[i], v[i] / blocks))
    for i in (11, 12, 13):
        print('   [%-14s] %10.1f ticks/stage' % (extra[i], v[i] / (stages - blocks)))
