"""Rebuild profiles/README.md from the rocprofv3 kernel-stats CSVs and bench JSON lines in profiles/."""
import csv, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
out = ["# profiles/ — round 1 (MI355X, 1 GPU, ROCm 7.2)\n\n",
       "Commands (on the GPU box, `cd /tmp && export TMPDIR=/tmp` first):\n\n",
       "```\nrocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o bench -- python bench.py --no-cpu-baseline [--precision fp32]\n"
       "rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'k_spconv_(implicit_bf16|pairs_bf16|pairs_wgrad_rows16)' --output-format csv ... -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline   # and a second pass with WRITE_SIZE\n"
       "python bench.py            # r01_bench_bf16.json (incl. cpu_baseline)\npython tools/stream_bw.py  # r01_stream_bw.txt\n```\n\n"
       "All of it is `tools/refresh_profiles.sh` (one gpurun call).  Device copy rate on this box: " + open(os.path.join(P, "r01_stream_bw.txt")).read().strip().splitlines()[-1] + ".\n\n",
       "13 steps per run (3 warm-up + 10 timed), batch = 4 synthetic S50k scenes, full training step (fwd + bwd + clip + AdamW).\n"]
for tag in ("bf16", "fp32"):
    rows = list(csv.DictReader(open(os.path.join(P, "r01_bench_%s_kernel_stats.csv" % tag))))
    steps = 13
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    calls = sum(int(r["Calls"]) for r in rows)
    b = json.loads(open(os.path.join(P, "r01_bench_%s.json" % tag)).read().strip().splitlines()[-1])
    r = b["roofline"]
    out.append("\n## %s operands — `r01_bench_%s_kernel_stats.csv`, `r01_bench_%s.json`\n\n" % (tag, tag, tag))
    out.append("bench line: **%.1f scenes/s**, %.1f ms/step. Dominant kernel `%s`: bound %s, achieved %.1f %s = **%.1f %%** of the %.0f %s peak; "
               "average launch %.3f ms over %d launches (HIP events, live in bench.py; the CSV's average for the same kernel agrees); "
               "it is %.0f %% of the step.\n\n" % (b["value"], b["ms_per_step"], r["kernel"].split(" ")[0], r["bound"], r["achieved"], r["unit"],
                                                100 * r["frac"], r["peak"], r["unit"], r["avg_launch_ms"], r["launches"], 100 * r["kernel_time_share"]))
    if not r.get("traffic") and tag == "bf16":
        import sys
        sys.path.insert(0, ROOT)
        import bench as _b
        r["traffic"] = _b.pmc_traffic("k_spconv_implicit_bf16")
    if r.get("traffic"):
        out.append("PMC (`r01_pmc_FETCH_SIZE.csv`, `r01_pmc_WRITE_SIZE.csv`, separate passes; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md): "
                   "%.0f MB HBM-side traffic per launch vs %.0f MB algorithmic (gathered rows + output rows + weights + map).\n\n"
                   % (r["traffic"] / 1e6, r["algorithmic_bytes_per_launch"] / 1e6))
    if "cpu_baseline" in b:
        c = b["cpu_baseline"]
        out.append("cpu_baseline (the oracle, `kind: port`): %.3f scenes/s on %d threads — %s. GPU / CPU = %.0fx.\n\n"
                   % (c["value"], c["cores"], c["sample"], b["value"] / c["value"]))
    out.append("GPU busy %.1f ms/step in %d launches/step.\n\n| ms/step | calls/step | avg µs | kernel |\n|---:|---:|---:|---|\n" % (tot / steps / 1e6, calls / steps))
    for x in rows[:24]:
        out.append("| %.2f | %d | %.1f | `%s` |\n" % (int(x["TotalDurationNs"]) / steps / 1e6, int(x["Calls"]) / steps, float(x["AverageNs"]) / 1e3,
                                                 x["Name"][:90].replace("|", "/")))
open(os.path.join(P, "README.md"), "w").write("".join(out))
print("".join(out)[:1800])
