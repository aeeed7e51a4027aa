"""Per-queue timeline of the backbone passes in a `rocprofv3 --kernel-trace` of tools/backbone_lanes.py --marks.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o bl -- python tools/backbone_lanes.py --marks --iters 2 --wgrad-lanes 0
    python tools/lane_timeline.py DIR/.../bl_kernel_trace.csv [--pass -1] [--list]

Passes are cut at the k_clip_coef marker launches (forward: markers 4k, 4k+1; backward: 4k+2, 4k+3).  For the chosen passes
it prints the span, the kernel time per queue, the time with kernels of two queues running at once, the idle time, and (--list)
every launch with its queue, start and duration."""
import argparse
import collections
import csv


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        elif e > cur_e:
            cur_e = e
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--step", type=int, default=-1, help="which timed step (of all steps in the trace; negative: from the end)")
    ap.add_argument("--list", action="store_true")
    a = ap.parse_args()
    ev = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    ev.sort()
    marks = [i for i, e in enumerate(ev) if e[2].startswith("k_clip_coef")]
    nsteps = len(marks) // 4
    k = a.step % nsteps
    for what, m0, m1 in (("forward", marks[4 * k], marks[4 * k + 1]), ("backward", marks[4 * k + 2], marks[4 * k + 3])):
        t0, t1 = ev[m0][1], ev[m1][0]
        body = [e for e in ev[m0 + 1:m1] if not e[2].startswith("k_clip_coef")]
        perq = collections.defaultdict(list)
        for s, e, n, q in body:
            perq[q].append((s, e))
        busy = union([(s, e) for s, e, _, _ in body])
        sums = {q: sum(e - s for s, e in v) for q, v in perq.items()}
        print("%s of step %d/%d: span %.3f ms, %d launches, some kernel running %.3f ms, idle %.3f ms" % (
            what, k, nsteps, (t1 - t0) / 1e6, len(body), busy / 1e6, (t1 - t0 - busy) / 1e6))
        for q in sorted(sums):
            print("   queue %s: %4d launches, kernel time %.3f ms, busy (union) %.3f ms" % (q, len(perq[q]), sums[q] / 1e6, union(perq[q]) / 1e6))
        if len(perq) > 1:
            print("   time with kernels of more than one queue running: %.3f ms" % ((sum(union(v) for v in perq.values()) - busy) / 1e6))
        if a.list:
            for s, e, n, q in body:
                print("     q%-3s %9.1f us  %8.1f us  %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, n[:90]))


if __name__ == "__main__":
    main()
