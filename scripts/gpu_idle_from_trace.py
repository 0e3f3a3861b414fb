"""Developer timing: how much of a training step the GPU runs NO kernel (union of the kernel intervals of a rocprofv3 --kernel-trace
of `bench.py`, two streams), and how much of it has two kernels in flight.  usage: python scripts/gpu_idle_from_trace.py <kernel_trace.csv> <steps>"""
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
steps = int(sys.argv[2])
# steady state: drop the first and last 15 % of the trace
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo, hi = t0 + 0.3 * (t1 - t0), t0 + 0.9 * (t1 - t0)
ev = []
for s, e, _ in rows:
    s, e = max(s, lo), min(e, hi)
    if e > s:
        ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, busy, multi = 0, lo, 0.0, 0.0
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: multi += t - last
    depth += d; last = t
span = hi - lo
print('window %.1f ms: busy %.2f %%, idle %.2f %% (= %.2f ms per 34-ms step), two or more kernels in flight %.1f %%' % (
    span / 1e6, 100 * busy / span, 100 * (1 - busy / span), 34.0 * (1 - busy / span), 100 * multi / span))
gaps = []
depth, last = 0, None
for t, d in ev:
    if depth == 0 and last is not None and d == 1: gaps.append(t - last)
    depth += d
    if depth == 0: last = t
gaps.sort()
if gaps:
    print('%d idle gaps: median %.2f us, 90 %% %.2f us, max %.1f us, sum %.2f ms' % (len(gaps), gaps[len(gaps)//2] / 1e3, gaps[int(len(gaps)*0.9)] / 1e3, gaps[-1] / 1e3, sum(gaps) / 1e6))
