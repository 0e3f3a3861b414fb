"""Developer timing: full training step at batch B with per-family hipEvent breakdown."""
import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import l3_oracle as o
from l3embedding_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mt = sys.argv[2] if len(sys.argv) > 2 else 'cnn_L3_melspec2'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dtype = sys.argv[4] if len(sys.argv) > 4 else 'f32'
v, a, l = o.synthetic_batch(B)
fp32_conv = sys.argv[5] if len(sys.argv) > 5 else __import__('os').environ.get('L3_FP32_CONV_ARG', 'f4x4')
eng = _lib.Engine(mt, B, dtype=dtype, fp32_conv=fp32_conv)
eng.upload_batch(v, a, l)
if __import__('os').environ.get('L3_LIVE_HEAD', '1') != '0':      # dense_2/kernel x 1/64: live loss gradients (bench.py live_head)
    eng.set_param('dense_2/kernel', eng.get_param('dense_2/kernel', (128, 2)) / np.float32(64))
for _ in range(2):
    eng.step_resident(1e-4)
print('warm', eng.step_results())
t0 = time.time()
for _ in range(steps):
    eng.step_resident(1e-4)
loss, acc = eng.step_results()
dt = (time.time() - t0) / steps
print('B=%d %s %s: %.2f ms/step, %.1f pairs/s, loss %.4f' % (B, mt, dtype, dt * 1e3, B / dt, loss))
eng.profile_enable(True)
for _ in range(steps):
    eng.step_resident(1e-4)
eng.sync()
pr = eng.profile_read()
tot = sum(p['ms'] for p in pr.values())
for k, p in pr.items():
    tf = p['flops'] / (p['ms'] * 1e-3) / 1e12 if p['ms'] > 0 else 0
    print('  %-12s %8.2f ms/step  %5.1f%%  launches/step %5.1f  %.1f TFLOP/s' % (k, p['ms'] / steps, 100 * p['ms'] / tot, p['launches'] / steps, tf))
