"""Developer timing: every kernel of ONE training step (from one softmax_ce launch to the next) as a timeline -- start offset,
duration, queue -- from a rocprofv3 --kernel-trace of `bench.py`.  usage: python scripts/step_timeline_from_trace.py <kernel_trace.csv> [step index]"""
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
which = int(sys.argv[2]) if len(sys.argv) > 2 else 10
soft = [i for i, r in enumerate(rows) if 'softmax_ce' in r[2]]
lo, hi = soft[which], soft[which + 1]
t0 = rows[lo][0]
print('step of %.3f ms, %d kernels' % ((rows[hi][0] - t0) / 1e6, hi - lo))
for s, e, n, q in rows[lo:hi]:
    n = n.replace('l3::', '').replace('(anonymous namespace)::', '').replace('void ', '')
    print('%9.1f us  +%7.1f us  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, q, n[:80]))
