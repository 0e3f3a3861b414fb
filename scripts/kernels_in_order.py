"""Developer timing: every kernel launch of ONE training step, in launch order, from a rocprofv3 kernel trace of a serialised run
(L3_TWO_STREAMS=0) -- median duration over the steps of the trace.  The step boundary is the front-end's framing kernel, which runs
once per step.   usage: python scripts/kernels_in_order.py <kernel_trace.csv> [substring filter ...]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'frame_audio' in r['Kernel_Name']]
steps = [rows[marks[i]:marks[i + 1]] for i in range(len(marks) - 1)]
L = len(steps[-1])
steps = [s for s in steps if len(s) == L and [r['Kernel_Name'] for r in s] == [r['Kernel_Name'] for r in steps[-1]]]
pats = sys.argv[2:]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = 0.0
for k in range(L):
    nm = steps[-1][k]['Kernel_Name']
    d = sorted(dur(s[k]) for s in steps)
    med = d[len(d) // 2]
    tot += med
    if pats and not any(p in nm for p in pats):
        continue
    short = re.sub(r'\(.*', '', nm.replace('l3::', '').replace('(anonymous namespace)::', '').replace('void ', ''))[:60]
    print('%3d %-60s grid %-8s wg %-5s median %8.1f us (min %8.1f)' % (k, short, steps[-1][k].get('Grid_Size_X', steps[-1][k].get('Grid_Size')),
                                                                     steps[-1][k].get('Workgroup_Size_X', steps[-1][k].get('Workgroup_Size')), med, d[0]))
print('%d launches per step, %d identical steps, sum of medians %.1f us' % (L, len(steps), tot))
