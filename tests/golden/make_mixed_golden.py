"""tests/golden/cnn_L3_melspec2_b8_bf16.npz: one training step of cnn_L3_melspec2 at batch 8 in the oracle's MIXED-PRECISION mode
(BASELINE.json configs[4]: bfloat16 operands / fp32 accumulate on the 3x3 layers with 64 k input and output channels,
bfloat16-stored activations and data gradients -- oracle.mixed_precision('bf16'), rules (1)-(3) in oracle/l3_oracle.py), float64
arithmetic everywhere else; parameters and batch are those of cnn_L3_melspec2_b8.npz (seeds 107 / 208), whose float64 values give
the yardstick: every record carries how far the mixed-precision answer is from the float64 one, and the GPU test's bounds are
multiples of the distances it measures against THIS file, well below that yardstick.

    python tests/golden/make_mixed_golden.py          (~3 minutes)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from oracle import l3_oracle as o  # noqa: E402
import make_golden as mg  # noqa: E402

MT, B, PSEED, DSEED, LR = 'cnn_L3_melspec2', 8, 107, 208, 1e-3
# activations the test compares element by element: the first mixed-precision convolution of each tower (conv2d_2, conv2d_9: their
# inputs are still fp32-identical in any implementation) and the last one of each (the embedding layers: everything upstream has
# been through seven bfloat16 stores)
TAPS = ('conv2d_2', 'conv2d_9', 'vision_embedding_layer', 'audio_embedding_layer')


def main():
    P = mg.perturbed_params(MT, PSEED)
    v, a, l = o.synthetic_batch(B, seed=DSEED)
    z64 = np.load(os.path.join(HERE, 'cnn_L3_melspec2_b8.npz'))
    with o.mixed_precision('bf16'):
        ev = o.forward(MT, P, v, a, False, np.float64)
        fw = o.forward(MT, P, v, a, True, np.float64, want_taps=True)
        P1 = {k: x.copy() for k, x in P.items()}
        out = o.train_step(MT, P1, o.AdamState(), o.BNMovingState(zero_debias=True), v, a, l, LR, np.float64)
    rec = dict(model_type=MT, batch=B, param_seed=PSEED, data_seed=DSEED, lr=LR, mode='bf16',
               eval_logits=ev['logits'], train_logits=out['logits'], train_probs=out['probs'], loss=out['loss'],
               data_loss=out['data_loss'], acc=out['acc'],
               logits_vs_float64=float(np.abs(out['logits'] - z64['train_logits']).max()),
               eval_logits_vs_float64=float(np.abs(ev['logits'] - z64['eval_logits']).max()))
    for t in TAPS:
        if t in fw['taps']:
            x = np.asarray(fw['taps'][t], np.float64)
            idx = mg.sample_idx(t, x.size, 4096)
            rec['tap:' + t] = x.ravel()[idx]
            rec['tapabs:' + t] = float(np.abs(x).mean())
    for n, g in out['grads'].items():
        gg = g - (2 * o.L2_WEIGHT * P[n].astype(np.float64) if n.endswith('/kernel') else 0)
        idx = mg.sample_idx(n, gg.size)
        rec['gnorm:' + n] = np.sqrt((gg ** 2).sum())
        rec['gsamp:' + n] = gg.ravel()[idx]
        if float(z64['gnorm:' + n]) >= 1e-7:       # the mixed-precision gradient measured against the float64 one
            rec['gd64:' + n] = np.array(mg.grad_metrics(gg, z64['gsamp:' + n], float(z64['gnorm:' + n]), idx))
    path = os.path.join(HERE, 'cnn_L3_melspec2_b8_bf16.npz')
    np.savez_compressed(path, **rec)
    print(path, os.path.getsize(path), 'logits vs float64', rec['logits_vs_float64'], 'taps', [k for k in rec if k.startswith('tap:')])
    worst = max((float(rec[k][0]), k) for k in rec if k.startswith('gd64:'))
    print('worst sampled gradient distance to float64 (of the RMS):', worst)


if __name__ == '__main__':
    main()
