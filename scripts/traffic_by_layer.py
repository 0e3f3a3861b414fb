"""HBM traffic PER LAYER of the F(4x4,3x3) forward / data-gradient launches of one serialised fp32 training step:
rocprofv3 FETCH_SIZE and WRITE_SIZE per dispatch (separate --pmc passes, scripts/pmc_conv.sh; KiB units; FETCH_SIZE x 2 = the gfx950
wide-load correction of MI355X_MICROARCH.md, HBM section) joined, by the launch's position in the step's fixed launch order, with the
launch durations of a kernel trace of the same workload (scripts/wino_layers_by_order.py output) and the layer's ALGORITHMIC bytes
(input read once + output written once + the Winograd-domain filter once; a data gradient that carries the fused BatchNorm-backward
reduction also reads that BatchNorm's input at the output resolution -- listed separately).

usage: python scripts/traffic_by_layer.py <pmc dir with fetch/ and write/> <wino4_layer_durations.txt> [batch = 64]
Answers: is a layer ALU- or HBM-limited?  TB/s = counter bytes / launch duration against 8.0 (spec) and 6.29 (measured streaming)."""
import collections
import csv
import glob
import os
import sys

root, durfile = sys.argv[1], sys.argv[2]
N = int(sys.argv[3]) if len(sys.argv) > 3 else 64
V = [('V.conv1b', 224, 224, 64, 64), ('V.conv2a', 112, 112, 64, 128), ('V.conv2b', 112, 112, 128, 128), ('V.conv3a', 56, 56, 128, 256),
     ('V.conv3b', 56, 56, 256, 256), ('V.conv4a', 28, 28, 256, 512), ('V.conv4b', 28, 28, 512, 512)]
A = [('A.conv1b', 256, 199, 64, 64), ('A.conv2a', 128, 99, 64, 128), ('A.conv2b', 128, 99, 128, 128), ('A.conv3a', 64, 49, 128, 256),
     ('A.conv3b', 64, 49, 256, 256), ('A.conv4a', 32, 24, 256, 512), ('A.conv4b', 32, 24, 512, 512)]
order = [(n + ' forward', h, w, ci, co, False) for n, h, w, ci, co in V + A]
order += [(n + ' data gradient', h, w, co, ci, True) for n, h, w, ci, co in V[::-1] + A[::-1]]


def per_dispatch(sub, counter):
    rows = []
    for path in glob.glob(os.path.join(root, sub, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get('Counter_Name') == counter and 'conv_wino4_kernel' in r.get('Kernel_Name', ''):
                    rows.append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    agg = collections.OrderedDict()
    for d, v in sorted(rows):               # a counter may be reported per XCD / instance: add the rows of one dispatch
        agg[d] = agg.get(d, 0.0) + v
    return list(agg.values())


fetch, write = per_dispatch('fetch', 'FETCH_SIZE'), per_dispatch('write', 'WRITE_SIZE')
per = len(order)
med = [float(l.split('median')[1].split('us')[0]) for l in open(durfile) if ' median ' in l and 'conv_wino4' in l]
assert len(med) == per, (len(med), per)
assert len(fetch) % per == 0 and len(write) % per == 0 and fetch and write, (len(fetch), len(write))


def layer_mean(vals, k):
    steps = len(vals) // per
    return sum(vals[s * per + k] for s in range(steps)) / steps


print('# per launch: counter bytes = FETCH_SIZE x 2 KiB + WRITE_SIZE KiB (mean over %d / %d profiled steps); duration = median of the kernel trace; '
      'batch %d' % (len(fetch) // per, len(write) // per, N))
print('# algorithmic = input + output + 36-position filter, fp32, each once; "+bn" = what a data gradient with the fused BatchNorm-backward '
      'reduction reads on top (at most: the BatchNorm input at the output resolution); ratio = counter / algorithmic')
print('%-24s %8s %9s %9s %8s %7s %7s %9s %8s' % ('layer', 'us', 'fetch MB', 'write MB', 'alg MB', '+bn MB', 'ratio', 'TB/s', 'of 6.29'))
tot = collections.Counter()
for k, (name, h, w, ci, co, dgrad) in enumerate(order):
    f = layer_mean(fetch, k) * 1024 * 2
    wr = layer_mean(write, k) * 1024
    alg = 4.0 * N * h * w * (ci + co) + 4.0 * 36 * ci * co
    bn = 4.0 * N * h * w * co if dgrad else 0.0       # the BatchNorm input (or its pooled winners) a fused backward reduction reads
    us = med[k]
    tbs = (f + wr) / (us * 1e-6) / 1e12
    tot['f'] += f; tot['w'] += wr; tot['alg'] += alg; tot['bn'] += bn; tot['us'] += us
    print('%-24s %8.1f %9.1f %9.1f %8.1f %7.1f %7.2f %9.2f %8.2f' % (name, us, f / 1e6, wr / 1e6, alg / 1e6, bn / 1e6, (f + wr) / alg, tbs, tbs / 6.29))
print('%-24s %8.1f %9.1f %9.1f %8.1f %7.1f %7.2f %9.2f %8.2f' % ('all 28 launches', tot['us'], tot['f'] / 1e6, tot['w'] / 1e6, tot['alg'] / 1e6, tot['bn'] / 1e6,
                                                                   (tot['f'] + tot['w']) / tot['alg'], (tot['f'] + tot['w']) / (tot['us'] * 1e-6) / 1e12,
                                                                   (tot['f'] + tot['w']) / (tot['us'] * 1e-6) / 1e12 / 6.29))
