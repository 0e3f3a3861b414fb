import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from l3embedding_amd import _lib
from oracle import l3_oracle as o
mt = 'cnn_L3_melspec2'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = 3
v, a, l = o.synthetic_batch(B, seed=4)
e1 = _lib.Engine(mt, B, seed=3); P = e1.get_params()
e1.train_step(v, a, l, 1e-4); G1 = e1.get_grads(); e1.close()
v3, a3, l3 = (np.concatenate([t] * R, axis=0) for t in (v, a, l))
e3 = _lib.Engine(mt, B * R, seed=3); e3.set_params(P)
e3.train_step(v3, a3, l3, 1e-4); G3 = e3.get_grads(); e3.close()
worst = sorted(((np.abs(G3[n] - G1[n]).max() / (np.abs(G1[n]).max() + 1e-30), n) for n in G1 if 'bias' not in n), reverse=True)[:6]
print('B', B, worst)
# oracle check of the property itself (float64), tiny model
mt2 = 'tiny_L3'
P2 = o.init_params(mt2, seed=1)
v, a, l = o.synthetic_batch(4, seed=2)
_, g1 = o.loss_and_grads(mt2, P2, v, a, l)
v3, a3, l3 = (np.concatenate([t] * R, axis=0) for t in (v, a, l))
_, g3 = o.loss_and_grads(mt2, P2, v3, a3, l3)
print('oracle property', max(np.abs(g3[n] - g1[n]).max() / (np.abs(g1[n]).max() + 1e-30) for n in g1 if 'bias' not in n))
