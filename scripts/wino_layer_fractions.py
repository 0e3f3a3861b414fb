"""Developer timing: issued-flop fraction of the fp32 MFMA peak for every F(4x4,3x3) launch of a serial training step, from the per-launch
medians of scripts/wino_layers_by_order.py (launch k of the step = a fixed layer: vision forward 1b..4b, audio forward 1b..4b, then the
data gradients in backward order: vision 4b..1b, audio 4b..1b).  usage: python scripts/wino_layer_fractions.py <wino4_layer_durations.txt> [batch = 64]"""
import sys
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
PEAK = 157.3e12
V = [('V.conv1b', 224, 224, 64, 64), ('V.conv2a', 112, 112, 64, 128), ('V.conv2b', 112, 112, 128, 128), ('V.conv3a', 56, 56, 128, 256),
     ('V.conv3b', 56, 56, 256, 256), ('V.conv4a', 28, 28, 256, 512), ('V.conv4b', 28, 28, 512, 512)]
A = [('A.conv1b', 256, 199, 64, 64), ('A.conv2a', 128, 99, 64, 128), ('A.conv2b', 128, 99, 128, 128), ('A.conv3a', 64, 49, 128, 256),
     ('A.conv3b', 64, 49, 256, 256), ('A.conv4a', 32, 24, 256, 512), ('A.conv4b', 32, 24, 512, 512)]
order = [(n + ' forward', h, w, ci, co) for n, h, w, ci, co in V + A]
order += [(n + ' data gradient', h, w, co, ci) for n, h, w, ci, co in V[::-1] + A[::-1]]
med = [float(l.split('median')[1].split('us')[0]) for l in open(sys.argv[1]) if ' median ' in l and 'conv_wino4' in l]
assert len(med) == len(order), (len(med), len(order))
print('# issued MFMA flops = 2 x 36 x tiles x Cin x Cout per launch, tiles = N ceil(H / 4) ceil(W / 4); peak 157.3 TFLOP/s; blocks = tile blocks x Cout / 64 on 256 CUs')
tot_f = tot_t = 0.0
for (name, h, w, ci, co), us in zip(order, med):
    tiles = N * ((h + 3) // 4) * ((w + 3) // 4)
    fl = 2.0 * 36 * tiles * ci * co
    blocks = ((tiles + 31) // 32) * (co // 64)
    tot_f += fl; tot_t += us
    print('%-24s %8.1f us  %6.1f TFLOP/s issued  %.3f of peak   (%5d blocks = %.2f per CU)' % (name, us, fl / us / 1e6, fl / us / 1e6 / 157.3, blocks, blocks / 256.0))
print('all 28 launches: %.1f us, %.3f of peak issued' % (tot_t, tot_f / tot_t / 1e6 / 157.3))
