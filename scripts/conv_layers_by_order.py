"""Developer timing: the forward / data-gradient convolution launches of a serialised step in launch order, from a rocprofv3 kernel trace
(any of conv_wino4_kernel, conv_wino_bx6_kernel, conv_wino_kernel; the F(4x4,3x3) tail launches are listed as their own rows).
usage: python scripts/conv_layers_by_order.py <kernel_trace.csv> <launches per step> [name substring ...]"""
import csv, sys
pats = sys.argv[3:] or ['conv_wino4_kernel', 'conv_wino_bx6_kernel', 'conv_wino_kernel', 'wino4_tail_reduce']
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(p in r['Kernel_Name'] for p in pats)]
per = int(sys.argv[2])
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
if len(rows) % per:
    rows = rows[len(rows) % per:]          # (drop the warm-up remainder)
steps = len(rows) // per
tot = 0.0
for k in range(per):
    d = sorted(dur(rows[s * per + k]) for s in range(steps))
    med = d[len(d) // 2]
    tot += med
    r = rows[k]
    nm = r['Kernel_Name']
    short = nm[nm.index('conv_w'):][:34] if 'conv_w' in nm else nm[:34]
    print('%2d %-36s grid %-7s median %8.1f us  (min %8.1f max %8.1f, %d steps)' % (k, short, r.get('Grid_Size_X', r.get('Grid_Size')), med, d[0], d[-1], steps))
print('sum of medians: %.1f us per step' % tot)
