"""Developer timing: the kernels of ONE training step between the towers' forward passes and their backward passes (dense head, loss,
L2 sums) as a timeline -- start offset, duration, queue -- from a rocprofv3 --kernel-trace of `bench.py` (two streams).
usage: python scripts/head_section_from_trace.py <kernel_trace.csv> [step index]"""
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
which = int(sys.argv[2]) if len(sys.argv) > 2 else 12
soft = [i for i, r in enumerate(rows) if 'softmax_ce' in r[2]]
i = soft[which]
lo = max(0, i - 14); hi = min(len(rows), i + 26)
t0 = rows[lo][0]
for s, e, n, q in rows[lo:hi]:
    n = n.replace('l3::', '').replace('(anonymous namespace)::', '')
    print('%9.1f us  +%7.1f us  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, q, n[:90]))
# whole-step view: time with < 2 kernels in flight around the head
