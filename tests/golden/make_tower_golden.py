"""tests/golden/audio_tower_b64.npz: BASELINE.json configs[1] -- "cnn_L3_melspec2 audio tower only, batch=64, fp32 (mel front-end +
audio conv kernels)" -- as one float64 step of the oracle: the kapre front-end (audio_model.py:367-369), the audio tower in
training mode (audio_model.py:370-437) and its backward pass from the stand-in loss mean(tower output), at the
batch the configuration names.  The inputs are regenerated from the seeds; the file keeps the tower output, a seeded sample of every
gradient with its norm, and the BatchNorm batch statistics.  About 4 minutes and 20 GB of float64 NumPy.

    python tests/golden/make_tower_golden.py
"""
import os
import sys
from collections import OrderedDict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import l3_oracle as o  # noqa: E402
from make_golden import perturbed_params, sample_idx  # noqa: E402

MT, B, PARAM_SEED, DATA_SEED = 'cnn_L3_melspec2', 64, 121, 222


def main():
    P = perturbed_params(MT, PARAM_SEED)
    _, a, _ = o.synthetic_batch(B, seed=DATA_SEED)
    spec = o.model_spec(MT)
    consts = {k.rsplit('/', 1)[1]: v for k, v in P.items() if '/' + spec['frontend_name'] + '/' in k}
    fe = o.frontend_forward(spec['frontend'], a, consts, 'sample', np.float64)
    out, caches = o._tower_forward('audio_model', spec['audio'], fe, P, True)
    G = OrderedDict()
    o._tower_backward('audio_model', spec['audio'], np.full(out.shape, 1.0 / out.size), caches, True, G)
    rec = dict(model_type=MT, batch=B, param_seed=PARAM_SEED, data_seed=DATA_SEED, out=out.astype(np.float64),
               frontend_sample=fe.ravel()[sample_idx('frontend', fe.size, 4096)])
    for n, g in G.items():
        g = np.asarray(g, np.float64)
        rec['gsamp:' + n] = g.ravel()[sample_idx(n, g.size)]
        rec['gnorm:' + n] = np.sqrt((g ** 2).sum())
    for op, c in zip(spec['audio'], caches):
        if op[0] == 'bn':
            rec['bnmean:' + op[1]] = c[1][2]
            rec['bnvar:' + op[1]] = c[1][3]
    path = os.path.join(HERE, 'audio_tower_b64.npz')
    np.savez_compressed(path, **rec)
    print('wrote', path, os.path.getsize(path), 'bytes; |out| max %.4f' % np.abs(out).max())


if __name__ == '__main__':
    main()
