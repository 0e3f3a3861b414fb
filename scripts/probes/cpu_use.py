"""How much host CPU one rank's training loop takes (process CPU time / wall time over 40 steps without a sync in between, then the
final wait): a spinning stream synchronize shows as ~1.0 core."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from l3embedding_amd import _lib
B = 64
WITH_TORCH = 'torch' in sys.argv[1:]          # as bench.py does: the engine on a stream torch made, torch.cuda.synchronize() in the wait
if WITH_TORCH:
    import torch
    ts = torch.cuda.Stream(device=0)
    e = _lib.Engine('cnn_L3_melspec2', B, device=0, seed=1, stream=ts.cuda_stream)
elif 'torchown' in sys.argv[1:]:               # torch imported and initialised, the engine on its own stream
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device='cuda')
    e = _lib.Engine('cnn_L3_melspec2', B, device=0, seed=1)
elif 'torchprio' in sys.argv[1:]:              # a torch stream of high priority (another pool)
    import torch
    ts = torch.cuda.Stream(device=0, priority=-1)
    e = _lib.Engine('cnn_L3_melspec2', B, device=0, seed=1, stream=ts.cuda_stream)
elif 'rawstream' in sys.argv[1:]:             # a stream made with hipStreamCreateWithFlags(hipStreamNonBlocking) outside the library
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    raw = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(raw), 1) == 0
    e = _lib.Engine('cnn_L3_melspec2', B, device=0, seed=1, stream=raw.value)
elif 'rawdefault' in sys.argv[1:]:            # ... with hipStreamCreate (a stream that synchronises with the null stream)
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    raw = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(raw)) == 0
    e = _lib.Engine('cnn_L3_melspec2', B, device=0, seed=1, stream=raw.value)
else:
    e = _lib.Engine('cnn_L3_melspec2', B, device=0, seed=1)
rng = np.random.RandomState(0)
e.upload_batch_raw(rng.randint(0, 255, (B, 224, 224, 3)).astype(np.uint8), (rng.randn(B, 1, 48000) * 3000).astype(np.int16),
                   np.eye(2, dtype=np.int32)[rng.randint(0, 2, B)])
for _ in range(5):
    e.step_resident(1e-4)
e.sync()
def threads():
    out = {}
    for t in os.listdir('/proc/self/task'):
        try:
            f = open('/proc/self/task/%s/stat' % t).read()
            name = f[f.index('(') + 1:f.rindex(')')]
            v = f[f.rindex(')') + 2:].split()
            out[t] = (name, (int(v[11]) + int(v[12])) / os.sysconf('SC_CLK_TCK'))
        except Exception:
            pass
    return out
for mode in ('launch 40 then wait', 'wait after every step'):
    th0 = threads()
    w0, c0 = time.perf_counter(), time.process_time()
    for _ in range(40):
        e.step_resident(1e-4)
        if mode != 'launch 40 then wait':
            e.step_results()
    w1, c1 = time.perf_counter(), time.process_time()
    e.sync()
    if WITH_TORCH:
        torch.cuda.synchronize()
    w2, c2 = time.perf_counter(), time.process_time()
    print('%s: enqueue %.1f ms wall / %.1f ms cpu per step; final wait %.1f ms wall / %.1f ms cpu; whole loop %.2f cores' % (
        mode, 1e3 * (w1 - w0) / 40, 1e3 * (c1 - c0) / 40, 1e3 * (w2 - w1), 1e3 * (c2 - c1), (c2 - c0) / (w2 - w0)), flush=True)
    th1 = threads()
    print('   per thread (cpu s): ' + ', '.join('%s %s %.2f' % (t, th1[t][0], th1[t][1] - th0.get(t, ('', 0.0))[1]) for t in sorted(th1)
                                                 if th1[t][1] - th0.get(t, ('', 0.0))[1] > 0.02), flush=True)
