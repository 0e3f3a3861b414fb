"""Aggregates rocprofv3 counter_collection CSVs per kernel name (mean per dispatch)."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(root, '*', '*counter_collection.csv')) + glob.glob(os.path.join(root, '*', '*', '*counter_collection.csv')):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r.get('Kernel_Name', '')
            short = name.split('(')[0].replace('void l3::', '').replace('l3::', '')[:48]
            c = r.get('Counter_Name')
            v = float(r.get('Counter_Value', 0))
            a = agg[short][c]
            a[0] += v
            a[1] += 1
keys = ['conv_igemm_kernel<2, 2, 64, 64, 16, false, true>', 'conv_igemm_kernel<4, 1, 64, 64, 16, false, true>', 'conv_wgrad9_kernel']
for k in sorted(agg, key=lambda k: -sum(v[0] for v in agg[k].values())):
    if not any(k.startswith(x[:20]) for x in keys) and 'bn_' not in k:
        continue
    print('==', k)
    for c, (s, n) in sorted(agg[k].items()):
        print('   %-28s mean/dispatch %16.1f   dispatches %d' % (c, s / n, n))
