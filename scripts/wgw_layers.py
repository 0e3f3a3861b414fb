"""Developer timing: per-layer duration of the Winograd weight-gradient launches from a rocprofv3 kernel trace.
usage: python scripts/wgw_layers.py <kernel_trace.csv> [name substring]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else 'conv_wgrad_wino_kernel'
sel = [r for r in rows if pat in r['Kernel_Name']]
groups = collections.OrderedDict()
for r in sel:
    key = (r['Kernel_Name'][:70], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), r.get('Workgroup_Size_X'))
    groups.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0
for k, v in groups.items():
    v2 = sorted(v)[len(v) // 4: max(len(v) // 4 + 1, 3 * len(v) // 4)]
    m = sum(v2) / len(v2)
    tot += m * len(v) 
    print('%-72s grid %-8s n=%3d  median-ish %8.1f us' % (k[0], k[1], len(v), m))
print('total us per occurrence-set:', tot)
