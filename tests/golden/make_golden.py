"""Generates tests/golden/*.npz from the numpy oracle (float64).

The reference itself cannot run here (keras/tensorflow/kapre absent: SURVEY.md 8c), so
these vectors pin the ORACLE (regression) and give the GPU tests a fixed target that
travels to the GPU box; they are not reference outputs.  Inputs are regenerated from the
recorded seeds (`oracle.synthetic_batch`), so the files stay small.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import l3_oracle as o  # noqa: E402

# batch 2 / 1: every BatchNorm normalises over one or two samples -- the worst-conditioned case, kept as the stress test;
# batch 8: a batch whose BatchNorm statistics are well conditioned, with tighter bounds;
# batch 64: the configuration the headline metric is quoted on (BASELINE.json configs[2]; train.py:408-414 at
# train_batch_size = 64) -- about 7 minutes and 35 GB of float64 NumPy, regenerated only on request
CASES = [('cnn_L3_melspec2', 2, 101, 202), ('tiny_L3', 3, 103, 204), ('cnn_L3_orig', 1, 105, 206),
         ('cnn_L3_melspec2', 8, 107, 208), ('cnn_L3_melspec2', 64, 109, 210),
         # the other two registry entries (model.py:220-262): a full training step each, not only the forward
         ('cnn_L3_kapredbinputbn', 2, 113, 214), ('cnn_L3_melspec1', 2, 115, 216)]
LR = 1e-3


def perturbed_params(model_type, seed):
    P = o.init_params(model_type, seed=seed)
    r = np.random.RandomState(seed + 1)
    for k in P:
        if k.endswith('/gamma'):
            P[k] = (1 + 0.1 * r.randn(*P[k].shape)).astype(np.float32)
        elif k.endswith('/beta') or k.endswith('/bias'):
            P[k] = (0.1 * r.randn(*P[k].shape)).astype(np.float32)
        elif k.endswith('/moving_mean'):
            P[k] = (0.05 * r.randn(*P[k].shape)).astype(np.float32)
        elif k.endswith('/moving_variance'):
            P[k] = (1 + 0.2 * r.rand(*P[k].shape)).astype(np.float32)
    return P


def sample_idx(name, size, k=256):
    """Seeded element sample of a tensor (whole tensor when it has <= k elements)."""
    if size <= k:
        return np.arange(size)
    return np.random.RandomState(sum(map(ord, name))).randint(0, size, size=k)


def grad_metrics(got, gsamp, gnorm, idx):
    """The two distances the GPU test bounds, for any implementation's data-term gradient `got`:
    max sampled |error| relative to the tensor's RMS, and the relative error of its L2 norm."""
    got = np.asarray(got, np.float64)
    rms = gnorm / np.sqrt(got.size)
    return (float(np.abs(got.ravel()[idx] - gsamp).max() / rms),
            float(abs(np.sqrt((got ** 2).sum()) - gnorm) / gnorm))


def main(only=None):
    for mt, B, pseed, dseed in CASES:
        if only and '%s_b%d' % (mt, B) not in only:
            continue
        P = perturbed_params(mt, pseed)
        v, a, l = o.synthetic_batch(B, seed=dseed)
        ev = o.forward(mt, P, v, a, False, np.float64)
        adam, bn = o.AdamState(), o.BNMovingState(zero_debias=True)
        P1 = {k: x.copy() for k, x in P.items()}
        out = o.train_step(mt, P1, adam, bn, v, a, l, LR, np.float64)
        rec = dict(model_type=mt, batch=B, param_seed=pseed, data_seed=dseed, lr=LR,
                   eval_logits=ev['logits'], eval_probs=ev['probs'],
                   train_logits=out['logits'], train_probs=out['probs'], loss=out['loss'],
                   data_loss=out['data_loss'], reg=out['reg'], acc=out['acc'])
        out.pop('fwd', None)          # the float64 activations (35 GB at batch 64) are not needed any more
        # the same step in float32 NumPy: how far a correct fp32 implementation of this graph lands from the
        # float64 answer (BatchNorm-backward cancellation).  The GPU test's gradient bounds are multiples of
        # these measured distances instead of constants.
        P32 = {k: x.copy() for k, x in P.items()}
        out32 = o.train_step(mt, P32, o.AdamState(), o.BNMovingState(zero_debias=True), v, a, l, LR, np.float32)
        rec['logits32_err'] = float(np.abs(out32['logits'] - out['logits']).max())
        for n, g in out['grads'].items():
            l2g = 2 * o.L2_WEIGHT * P[n].astype(np.float64) if n.endswith('/kernel') else 0
            gg = g - l2g                                                                           # data-term gradient
            idx = sample_idx(n, gg.size)
            rec['gnorm:' + n] = np.sqrt((gg ** 2).sum())
            rec['gsamp:' + n] = gg.ravel()[idx]
            rec['w1samp:' + n] = P1[n].astype(np.float64).ravel()[idx]
            if rec['gnorm:' + n] >= 1e-7:
                rec['gd32:' + n] = np.array(grad_metrics(out32['grads'][n].astype(np.float64) - l2g, rec['gsamp:' + n],
                                                         float(rec['gnorm:' + n]), idx))
            rec['w1d32:' + n] = float(np.abs(P32[n].astype(np.float64).ravel()[idx] - rec['w1samp:' + n]).max())
        for n in P1:
            if n.endswith('/moving_mean') or n.endswith('/moving_variance'):
                rec['mov:' + n] = P1[n].astype(np.float64)
        path = os.path.join(HERE, '%s_b%d.npz' % (mt, B))
        np.savez_compressed(path, **rec)
        print(path, os.path.getsize(path))


if __name__ == '__main__':
    main(sys.argv[1:])            # optional: the cases to (re)generate, e.g. cnn_L3_melspec2_b8
