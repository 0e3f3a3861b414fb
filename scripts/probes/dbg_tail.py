"""Developer probe: which part of the split-tail path moves the gradients (see tests/test_parity_gpu.py::test_split_tail_...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from l3embedding_amd import _lib
from oracle import l3_oracle as o
mt, B = 'cnn_L3_melspec2', 4
v, a, l = o.synthetic_batch(B, seed=21)
def run(env):
    for k, val in env.items(): os.environ[k] = val
    eng = _lib.Engine(mt, B, seed=6)
    loss, _ = eng.train_step(v, a, l, 1e-4)
    g = eng.get_grads(); eng.close()
    return loss, g
def cmp(tag, A, Bv):
    d = {n: float(np.abs(Bv[1][n] - A[1][n]).max() / (np.abs(A[1][n]).max() + 1e-30)) for n in A[1] if not ((n.endswith('/bias') and not n.startswith('dense')) or A[1][n].size == 1)}
    top = sorted(d.items(), key=lambda kv: -kv[1])[:5]
    print(tag, 'loss', A[0], Bv[0], ' | '.join('%s %.1e' % kv for kv in top))
base = run({'L3_W4_TAIL': '0', 'L3_BNBWD_FUSE': '1'})
cmp('tail=2 fuse=1', base, run({'L3_W4_TAIL': '2', 'L3_BNBWD_FUSE': '1'}))
nf = run({'L3_W4_TAIL': '0', 'L3_BNBWD_FUSE': '0'})
cmp('tail=0 fuse=0 vs base', base, nf)
cmp('tail=2 fuse=0 vs tail=0 fuse=0', nf, run({'L3_W4_TAIL': '2', 'L3_BNBWD_FUSE': '0'}))
os.environ['L3_W4_NCU'] = '24'
cmp('tail=2 fuse=1 ncu=24', base, run({'L3_W4_TAIL': '2', 'L3_BNBWD_FUSE': '1'}))
