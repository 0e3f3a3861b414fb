"""Developer timing: per-layer duration of the F(4x4,3x3) launches from a rocprofv3 kernel trace of `bench.py --serial`.
All launches share one (persistent) grid, so a launch is identified by its position in the step's fixed launch order:
14 forward launches (audio 1b..4b, vision 1b..4b), then 14 data-gradient launches in the backward order.
usage: python scripts/wino_layers_by_order.py <kernel_trace.csv> [launches per step = 28]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'conv_wino4_kernel' in r['Kernel_Name']]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 28
rows.sort(key=lambda r: int(r['Start_Timestamp']))
assert len(rows) % per == 0, (len(rows), per)
steps = len(rows) // per
tot = 0.0
for k in range(per):
    d = sorted((int(rows[s * per + k]['End_Timestamp']) - int(rows[s * per + k]['Start_Timestamp'])) / 1e3 for s in range(steps))
    name = rows[k]['Kernel_Name']
    inst = name[name.index('conv_wino4_kernel'):][:20]
    med = d[len(d) // 2]
    tot += med
    print('%2d %-22s median %8.1f us  (min %8.1f max %8.1f, %d steps)' % (k, inst, med, d[0], d[-1], steps))
print('sum of medians: %.1f us per step' % tot)
