import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import l3_oracle as o
from l3embedding_amd import _lib
mt, B = 'cnn_L3_melspec2', 4
v, a, l = o.synthetic_batch(B, seed=11)
G = {}
for fuse in ('1', '0'):
    os.environ['L3_BNBWD_FUSE'] = fuse
    eng = _lib.Engine(mt, B, seed=5, dtype='bf16')
    eng.train_step(v, a, l, 1e-4)
    G[fuse] = eng.get_grads(); eng.close()
for name in G['0']:
    if 'batch_normalization' in name:
        g0, g1 = G['0'][name], G['1'][name]
        print('%-50s %.2e' % (name, np.abs(g1 - g0).max() / (np.abs(g0).max() + 1e-30)))
