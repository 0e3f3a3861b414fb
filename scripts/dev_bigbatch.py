import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from l3embedding_amd import _lib
from oracle import l3_oracle as o
mt, B, R = 'cnn_L3_melspec2', 64, 3
v, a, l = o.synthetic_batch(B, seed=4)
e1 = _lib.Engine(mt, B, seed=3)
P = e1.get_params()
_, lg1 = e1.forward(v, a, training=True)
names = ['vision_model/conv2d_%d' % i for i in (1, 2, 3, 4)] + ['audio_model/conv2d_%d' % i for i in (8, 9, 10)]
acts1 = {n: e1.activation(n) for n in names}
e1.close()
e3 = _lib.Engine(mt, B * R, seed=3)
e3.set_params(P)
v3, a3, l3 = (np.concatenate([t] * R, axis=0) for t in (v, a, l))
_, lg3 = e3.forward(v3, a3, training=True)
print('logit diff', np.abs(lg3[:B] - lg1).max(), np.abs(lg3[2*B:] - lg1).max(), 'scale', np.abs(lg1).max())
for n in names:
    a3_ = e3.activation(n)
    k = acts1[n].size
    for r in range(R):
        d = np.abs(a3_[r * k:(r + 1) * k] - acts1[n]).max()
        print(n, 'copy', r, 'max diff %.3e' % d, 'scale %.3f' % np.abs(acts1[n]).max())
e3.close()
e1 = _lib.Engine(mt, B, seed=3); e1.set_params(P)
e1.train_step(v, a, l, 1e-4); G1 = e1.get_grads(); e1.close()
e3 = _lib.Engine(mt, B * R, seed=3); e3.set_params(P)
e3.train_step(v3, a3, l3, 1e-4); G3 = e3.get_grads(); e3.close()
for n in G1:
    if 'kernel' in n or 'gamma' in n:
        print('%-50s diff/max %.2e  max %.2e' % (n, np.abs(G3[n] - G1[n]).max() / (np.abs(G1[n]).max() + 1e-30), np.abs(G1[n]).max()))
