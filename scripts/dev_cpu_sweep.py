import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from oracle import l3_oracle as o
from oracle.torch_cpu import TorchCpuTrainer
mt = 'cnn_L3_melspec2'
P = o.init_params(mt, seed=1)
for nt, B in ((16, 8), (32, 8), (64, 8), (64, 16), (128, 16)):
    torch.set_num_threads(nt)
    v, a, l = o.synthetic_batch(B)
    tr = TorchCpuTrainer(mt, P)
    tr.step(v[:2], a[:2], l[:2], 1e-4)
    ts = []
    for _ in range(2):
        t0 = time.time(); tr.step(v, a, l, 1e-4); ts.append(time.time() - t0)
    print('threads', nt, 'B', B, 'best %.2f s' % min(ts), '%.2f pairs/s' % (B / min(ts)), flush=True)
