"""tests/golden/cnn_L3_melspec2_b64_traj.npz: THREE consecutive training steps of the headline configuration (cnn_L3_melspec2,
batch 64 = BASELINE.json configs[2]; l3embedding/train.py:408-414 `fit_generator` at train_batch_size = 64) in the float64 NumPy
oracle -- Adam moments, BatchNorm moving statistics and the zero-debias accumulators carried from step to step -- followed by an
inference-mode forward of a fourth batch.  ~25 minutes and 35 GB; regenerated only on request:

    python tests/golden/make_traj_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from oracle import l3_oracle as o  # noqa: E402
import make_golden as mg  # noqa: E402

# lr 1e-5 (a tenth of train.py:222's default): at 1e-3 the first Adam step -- every weight by lr * sign(g) -- saturates the softmax of this random-data
# problem and the later steps see clipped probabilities (loss = 16.1 x error rate, zero data gradient): nothing left to compare
MT, B, PSEED, DSEED, LR, STEPS = 'cnn_L3_melspec2', 64, 111, 300, 1e-5, 3


def main():
    P = mg.perturbed_params(MT, PSEED)
    adam, bn = o.AdamState(), o.BNMovingState(zero_debias=True)
    rec = dict(model_type=MT, batch=B, param_seed=PSEED, data_seed=DSEED, lr=LR, steps=STEPS)
    losses, accs = [], []
    for s in range(STEPS):
        v, a, l = o.synthetic_batch(B, seed=DSEED + s)
        out = o.train_step(MT, P, adam, bn, v, a, l, LR, np.float64)
        losses.append(float(out['loss']))
        accs.append(float(out['acc']))
        rec['logits%d' % s] = out['logits']
        del out
        print('step', s, losses[-1], accs[-1], flush=True)
    rec['loss'] = np.array(losses)
    rec['acc'] = np.array(accs)
    v, a, l = o.synthetic_batch(B, seed=DSEED + STEPS)
    ev = o.forward(MT, P, v, a, False, np.float64)
    rec['eval_logits'] = ev['logits']
    for n in P:
        idx = mg.sample_idx(n, P[n].size)
        rec['w:' + n] = np.asarray(P[n], np.float64).ravel()[idx]
    path = os.path.join(HERE, 'cnn_L3_melspec2_b64_traj.npz')
    np.savez_compressed(path, **rec)
    print(path, os.path.getsize(path))


if __name__ == '__main__':
    main()
