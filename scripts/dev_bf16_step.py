import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from l3embedding_amd import _lib
from oracle import l3_oracle as o
mt = 'cnn_L3_melspec2'
for B, seed in ((2, 77), (4, 5)):
    P = o.init_params(mt, seed=seed)
    v, a, l = o.synthetic_batch(B, seed=seed + 1)
    with o.mixed_precision('bf16'):
        ref = o.forward(mt, P, v, a, True, np.float64)
        ref_e = o.forward(mt, P, v, a, False, np.float64)
    ref32 = o.forward(mt, P, v, a, True, np.float64)
    eng = _lib.Engine(mt, B, dtype='bf16'); eng.set_params(P)
    _, lg = eng.forward(v, a, training=True)
    _, lge = eng.forward(v, a, training=False)
    e32 = _lib.Engine(mt, B); e32.set_params(P)
    _, lg32 = e32.forward(v, a, training=True)
    print('B', B, 'logit scale', np.abs(ref['logits']).max())
    print('  gpu_bf16 - oracle_bf16 (train)', np.abs(lg - ref['logits']).max(), '(eval)', np.abs(lge - ref_e['logits']).max())
    print('  oracle_bf16 - oracle_fp32      ', np.abs(ref['logits'] - ref32['logits']).max())
    print('  gpu_fp32 - oracle_fp32         ', np.abs(lg32 - ref32['logits']).max())
    eng.close(); e32.close()
