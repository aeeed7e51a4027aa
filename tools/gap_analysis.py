"""GPU idle-gap analysis of a rocprofv3 kernel trace of bench.py (dev tool).
usage: python tools/gap_analysis.py <kernel_trace.csv>"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
idx = [i for i, x in enumerate(k) if 'FusedAdam' in x[2]]
per = 14
ends = [idx[i] for i in range(per - 1, len(idx), per)]
a, b = ends[-2] + 1, ends[-1] + 1
step = k[a:b]
span = (step[-1][1] - step[0][0]) / 1e6
busy = sum(e - s for s, e, _ in step) / 1e6
print("last step: span %.1f ms, GPU busy %.1f ms, idle %.1f ms, %d kernels" % (span, busy, span - busy, len(step)))
short = lambda n: n.replace('void ', '')[:60]
# timeline in 2 ms buckets: busy fraction + dominant kernel
t0 = step[0][0]
nb = int(span // 2) + 1
bus = [0.0] * nb
names = [collections.Counter() for _ in range(nb)]
for s, e, n in step:
    bi = int((s - t0) / 2e6)
    bus[bi] += (e - s) / 1e6
    names[bi][short(n)] += (e - s) / 1e6
for i in range(nb):
    top = names[i].most_common(1)
    print("  t=%5.1f ms  busy %4.0f%%  %s" % (i * 2.0, 100 * bus[i] / 2.0, top[0][0] if top else ""))
