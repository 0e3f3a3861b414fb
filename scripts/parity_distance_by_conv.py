"""Where does the full-step gradient distance come from?  The BatchNorm-backward finals already sum their partials in
float64 (bn_fused.hip fast_final_kernel), so the reduction's LAST stage is not it.  This runs the golden training step at batch 8 and 64
with each fp32_conv algorithm -- F(4x4,3x3) fp32 (layer error ~8e-6 of the range), F(2x2,3x3) fp32 (~1e-6), F(2x2,3x3) split-bf16 (~5e-7) --
and prints, per algorithm, the worst sampled gradient error / RMS and the five worst tensors: if the distance follows the convolutions'
rounding error, the noise enters THERE and is amplified by the cancellation in the BatchNorm-backward sums (a beta gradient whose
N*H*W terms cancel to a tenth of their RMS), not in how those sums are added up.
usage (GPU box): python scripts/parity_distance_by_conv.py [golden file ...]"""
import importlib.util, os, sys
import numpy as np
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
from l3embedding_amd import _lib
from oracle import l3_oracle as o
GOLDEN = os.path.join(HERE, 'tests', 'golden')
spec = importlib.util.spec_from_file_location('make_golden', os.path.join(GOLDEN, 'make_golden.py'))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
for fname in (sys.argv[1:] or ['cnn_L3_melspec2_b8.npz', 'cnn_L3_melspec2_b64.npz']):
    z = np.load(os.path.join(GOLDEN, fname))
    mt, B = str(z['model_type']), int(z['batch'])
    P = mod.perturbed_params(mt, int(z['param_seed']))
    v, a, l = o.synthetic_batch(B, seed=int(z['data_seed']))
    for algo in ('f4x4', 'f2x2', 'f2x2_bf16x6'):
        eng = _lib.Engine(mt, B, fp32_conv=algo)
        eng.set_params(P)
        _, logits = eng.forward(v, a, training=True)
        dlog = float(np.abs(logits - z['train_logits']).max())
        eng.train_step(v, a, l, float(z['lr']))
        G = eng.get_grads()
        rows = []
        for n, _, tr in eng.param_table():
            if not tr or float(z['gnorm:' + n]) < 1e-7 or G[n].size == 1:
                continue
            idx = mod.sample_idx(n, G[n].size)
            err, nerr = mod.grad_metrics(G[n].astype(np.float64), z['gsamp:' + n], float(z['gnorm:' + n]), idx)
            rows.append((err, nerr, n))
        rows.sort(reverse=True)
        print('%s %-12s logits %.2e  worst grad err/rms %.2e  norm err %.2e   worst: %s' % (
            fname, algo, dlog, rows[0][0], max(r[1] for r in rows), ', '.join('%s %.1e' % (r[2].replace('_model/batch_normalization', '.bn').replace('_model/conv2d', '.conv'), r[0]) for r in rows[:5])))
        eng.close()
