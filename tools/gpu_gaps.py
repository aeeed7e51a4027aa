"""Where the GPU idles inside a step: the gaps of a `rocprofv3 --kernel-trace` timeline.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o bench -- python bench.py ...
    python tools/gpu_gaps.py DIR/.../bench_kernel_trace.csv [--steps K] [--min-gap-us 15]

The trace is cut into steps at `k_adamw_table` (one launch per optimiser step).  For the
last K steps it prints, per step: wall time, time with at least one kernel running (union over
all queues), time per queue, and the idle time; then the idle time summed by the kernel that
ENDED the gap (the launch the GPU was waiting for), largest first.
"""
import argparse
import collections
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--min-gap-us", type=float, default=15.0)
    ap.add_argument("--marker", default="k_adamw_table")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()

    ev = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    ev.sort()
    marks = [i for i, e in enumerate(ev) if e[2].startswith(a.marker)]
    if len(marks) < a.steps + 1:
        raise SystemExit(f"only {len(marks)} markers in the trace")
    marks = marks[-(a.steps + 1):]
    gaps_by = collections.Counter()
    gaps_n = collections.Counter()
    before_by = collections.Counter()
    tot_wall = tot_busy = 0
    queues = collections.Counter()
    for s in range(a.steps):
        lo, hi = marks[s] + 1, marks[s + 1] + 1
        step = ev[lo:hi]
        t0 = ev[marks[s]][1]
        t1 = step[-1][1]
        wall = t1 - t0
        busy = 0
        cur_end = t0
        prev_name = ev[marks[s]][2]
        for st, en, name, q in step:
            queues[q] += en - st
            if st > cur_end:
                g = st - cur_end
                if g >= a.min_gap_us * 1000:
                    gaps_by[name[:70]] += g
                    gaps_n[name[:70]] += 1
                    before_by[(prev_name[:50], name[:50])] += g
                busy += en - st
                cur_end = en
                prev_name = name
            elif en > cur_end:
                busy += en - cur_end
                cur_end = en
                prev_name = name
        tot_wall += wall
        tot_busy += busy
        print(f"step {s}: wall {wall / 1e6:7.3f} ms, some kernel running {busy / 1e6:7.3f} ms, idle {(wall - busy) / 1e6:6.3f} ms, {len(step)} launches")
    k = a.steps
    print(f"mean: wall {tot_wall / k / 1e6:.3f} ms, busy {tot_busy / k / 1e6:.3f} ms, idle {(tot_wall - tot_busy) / k / 1e6:.3f} ms")
    print("kernel time per queue and step:", {q: round(v / k / 1e6, 3) for q, v in queues.items()})
    print(f"\nidle time (gaps >= {a.min_gap_us} us) by the launch that ended the gap, per step:")
    for name, g in gaps_by.most_common(a.top):
        print(f"  {g / k / 1e3:8.1f} us  {gaps_n[name] / k:5.1f} gaps  {name}")
    print("\nby (last kernel before the gap -> first after):")
    for (p, n), g in before_by.most_common(a.top):
        print(f"  {g / k / 1e3:8.1f} us  {p}  ->  {n}")


if __name__ == "__main__":
    main()
