"""Developer timing: per-layer duration of the F(4x4,3x3) launches from a rocprofv3 kernel trace of `bench.py --serial`.
A launch is identified by its position in the step's fixed launch order: 14 forward layers (vision 1b..4b, audio 1b..4b), then 14
data-gradient layers in the backward order.  A layer is one conv_wino4_kernel launch, or -- when its tile blocks do not divide by
the CU count (conv_wino4_launch, the tail) -- up to three: the full rounds, the channel-sliced tail (conv_wino4_kernel<0>) and
wino4_tail_reduce_kernel; their durations are added.
usage: python scripts/wino_layers_by_order.py <kernel_trace.csv> [layers per step = 28]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'conv_wino4_kernel' in r['Kernel_Name'] or 'wino4_tail_reduce' in r['Kernel_Name']]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 28
rows.sort(key=lambda r: int(r['Start_Timestamp']))
is_red = lambda r: 'wino4_tail_reduce' in r['Kernel_Name']
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
layers, i = [], 0
while i < len(rows):
    if i + 1 < len(rows) and is_red(rows[i + 1]):
        grp = rows[i:i + 2]
    elif i + 2 < len(rows) and is_red(rows[i + 2]) and not is_red(rows[i + 1]):
        grp = rows[i:i + 3]
    else:
        grp = rows[i:i + 1]
    i += len(grp)
    name = grp[-1]['Kernel_Name'] if len(grp) > 1 else grp[0]['Kernel_Name']
    key = 'wino4_tail_reduce_kernel' if len(grp) > 1 else 'conv_wino4_kernel'
    inst = name[name.index(key) + len(key):][:3]
    layers.append((sum(dur(r) for r in grp), 'conv_wino4_kernel' + inst + (' +tail' if len(grp) > 1 else ''), [dur(r) for r in grp]))
assert len(layers) % per == 0, (len(layers), per)
steps = len(layers) // per
tot = 0.0
for k in range(per):
    d = sorted(layers[s * per + k][0] for s in range(steps))
    med = d[len(d) // 2]
    tot += med
    parts = layers[(steps // 2) * per + k][2]
    print('%2d %-28s median %8.1f us  (min %8.1f max %8.1f, %d steps)%s' % (
        k, layers[k][1], med, d[0], d[-1], steps, '   [' + ' + '.join('%.0f' % x for x in parts) + ']' if len(parts) > 1 else ''))
print('sum of medians: %.1f us per step' % tot)
