"""Rebuild profiles/README.md from the rocprofv3 kernel-stats CSVs, PMC CSVs and bench JSON lines in profiles/ (round tag
from $ROUND, default r02)."""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = os.path.join(ROOT, "profiles")
RN = os.environ.get("ROUND", "r06")


def rd(name):
    return open(os.path.join(P, name)).read()


out = ["# profiles/ — round %s (MI355X, 1 GPU, ROCm 7.2)\n\n" % RN[1:].lstrip("0"),
       "Everything here is produced by `tools/refresh_profiles.sh` in one `gpurun` call (`cd /tmp && export TMPDIR=/tmp` first) and copied "
       "from `gpurun_out/refresh/`:\n\n```\n"
       "rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o bench -- python bench.py --no-cpu-baseline --no-fp32 [--precision fp32]\n"
       "rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'k_spconv_(tile|implicit_bf16|pairs_bf16|pairs_wgrad_rows16)' --output-format csv ... -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32   # second pass: WRITE_SIZE\n"
       "python bench.py                      # %s_bench_bf16.json (incl. fp32 sub-record and cpu_baseline)\n"
       "python tools/conv_shapes.py --wgrad  # %s_conv_shapes.txt: every conv launch shape, time, SURVEY 8(d) bound\n"
       "bash tools/pmc_sq.sh k_spconv_tile tools/mb_tile_one.py 4 128 128     # SQ counters, 128->128 layer (82 107 rows)\n"
       "bash tools/pmc_sq.sh wgrad_rows16 tools/mb_wgrad_one.py 4 128 128; bash tools/pmc_wgrad.sh 4 128 128   # weight gradient: SQ and L2 / memory-side counters\n"
       "python tools/mb_tile.py; python tools/mb_bn16.py; python tools/host_profile.py; python tools/stream_bw.py\n"
       "# the other configurations at the one-GPU size (BASELINE.json configs[3] / [4]): the same kernel-stats, conv-shape and FETCH / WRITE passes with\n"
       "#   --dataset sunrgbd --config S100k-yaw --batch 8   (%s_s100kyaw8_*)    and    --config S200k   (%s_s200k4_*, 0.01 m voxels)\n```\n\n" % (RN, RN, RN, RN),
       "Default precision of every run here: bf16 backbone with its rows stored as bf16 + split (fp32-accurate) heads -- `bench.py`'s default, BASELINE.json configs[1].  "
       "Earlier rounds' files (`r01_*` ... `r04_*`) are kept for comparison; `%s_synthetic_convergence.json`: `tools/synthetic_convergence.py` (fp32 / the bench precision / "
       "the same with fp32 rows / bf16 in every convolution / fp32 again, one seed, 1 008 iterations; indoor_eval mAP / recall and the loss curves).\n\n" % RN,
       "Device copy rate on the box: " + rd("%s_stream_bw.txt" % RN).strip().splitlines()[-1] + ".\n\n"]
import bench as _b
for tag in ("bf16", "fp32"):
    f = os.path.join(P, "%s_bench_%s_kernel_stats.csv" % (RN, tag))
    if not os.path.exists(f):
        continue
    rows = list(csv.DictReader(open(f)))
    f0 = os.path.join(P, "%s_bench_%s_lanes0_kernel_stats.csv" % (RN, tag))
    rows0 = list(csv.DictReader(open(f0))) if os.path.exists(f0) else rows       # the run compiled without lanes: every kernel alone
    b = json.loads(rd("%s_bench_%s.json" % (RN, tag)).strip().splitlines()[-1])
    steps = b["steps"] + b["warmup"]
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    calls = sum(int(r["Calls"]) for r in rows)
    r = b["roofline"]
    out.append("\n## %s operands — `%s_bench_%s_kernel_stats.csv`, `%s_bench_%s.json`\n\n" % (tag, RN, tag, RN, tag))
    out.append("bench line: **%.1f scenes/s**, %.1f ms/step (%d timed steps). Dominant kernel `%s`: bound %s (SURVEY 8(d): max(flops / peak, every-tensor-once bytes / 8 TB/s)), "
               "achieved %.1f %s = **%.1f %%** of the %.0f %s peak; average launch %.3f ms over %d launches (HIP events, live in bench.py; the CSV's average for the same kernel: %s); "
               "%.0f %% of the step.  Sum of the 8(d) bounds of ALL conv launches / their measured time: %.1f %%; / the whole step: %.1f %%.\n\n"
               % (b["value"], b["ms_per_step"], b["steps"], r["kernel"].split(" ")[0], r["bound"], r["achieved"], r["unit"], 100 * r["frac"], r["peak"], r["unit"],
                  r["avg_launch_ms"], r["launches"],
                  ", ".join("%.3f ms" % (float(x["AverageNs"]) / 1e6) for x in rows0 if r["kernel"].split(" ")[0] in x["Name"])[:60] or "n/a",
                  100 * r["kernel_time_share"], 100 * r.get("conv_bound_over_conv_time", 0), 100 * r.get("conv_bound_over_step_time", 0)))
    if tag == "bf16" and rows0 is not rows:
        ol = r.get("on_lanes") or {}
        out.append("The CSV averages quoted above are those of `%s_bench_bf16_lanes0_kernel_stats.csv` (the same command under `CG3D_LANES=0`: launch programs compiled "
                   "without lanes, every kernel alone on the one stream -- what `roofline` times, on steps compiled that way).  In the default run the backbone's coarse chain, "
                   "DAPPM's branches and the weight gradients run on queues of their own beside the stride-4 chain (engine.py, lanes): the table below is that run, and a launch's "
                   "duration in it is the time it took to get through a shared device (`roofline.on_lanes`: %.3f ms per launch of the dominant kernel, %.1f %% -- not a property of the kernel; "
                   "its averages in this table: %s).\n\n"
                   % (RN, ol.get("avg_launch_ms", float("nan")), 100 * ol.get("frac", float("nan")),
                      ", ".join("%.3f ms" % (float(x["AverageNs"]) / 1e6) for x in rows if r["kernel"].split(" ")[0] in x["Name"])[:60] or "n/a"))
    if tag == "bf16":
        for kn in ("k_spconv_tile", "k_spconv_implicit_bf16", "k_spconv_pairs_wgrad_rows16"):
            t, _src = _b.pmc_traffic(kn)
            if t:
                out.append("PMC `%s`: %.0f MB memory-side traffic per launch (2 x FETCH_SIZE + WRITE_SIZE, separate passes, averaged over the step's launches of that kernel; "
                           "FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md)%s.\n\n"
                           % (kn, t / 1e6, (" vs %.0f MB algorithmic (8(d)) for the dominant kernel" % (r["algorithmic_bytes_per_launch"] / 1e6)) if kn in r["kernel"] else ""))
        for k, v in sorted(r.get("conv_kernels", {}).items(), key=lambda kv: -kv[1]["ms_per_step"]):
            out.append("* `%s`: %.1f launches/step, %.2f ms/step, 8(d) bound / measured = %.1f %%, %.0f TF/s on the pairs, %.0f GB/s of 8(d) bytes\n"
                       % (k, v["launches_per_step"], v["ms_per_step"], 100 * v["bound_over_measured"], v["tflops"], v["gbytes_per_s_8d"]))
        out.append("\n")
    if "fp32" in b and isinstance(b["fp32"], dict):
        f32 = b["fp32"]
        out.append("fp32 sub-record of the same run: %.1f scenes/s, %.1f ms/step, dominant `%s` at %.1f %% of its %s roofline.\n\n"
                   % (f32["value"], f32["ms_per_step"], f32["dominant_kernel"].split(" ")[0], 100 * f32["frac"], f32["bound"]))
    if "cpu_baseline" in b:
        c = b["cpu_baseline"]
        out.append("cpu_baseline (the oracle, `kind: port`): %.3f scenes/s on %d threads — %s." % (c["value"], c["cores"], c["sample"]))
        s1 = c.get("single_thread_small_scene") or c.get("single_thread")
        if s1 and s1.get("value"):
            out.append(" One thread, on a 10 x smaller scene (its own unit, not comparable with the line above): %.3f %s — %s."
                       % (s1["value"], s1.get("unit", "S5k-scenes/s"), s1["sample"]))
        out.append("\n\n")
    out.append("GPU busy %.1f ms/step in %d launches/step.\n\n| ms/step | calls/step | avg µs | kernel |\n|---:|---:|---:|---|\n" % (tot / steps / 1e6, calls / steps))
    for x in sorted(rows, key=lambda x: -int(x["TotalDurationNs"]))[:28]:
        out.append("| %.2f | %.1f | %.1f | `%s` |\n" % (int(x["TotalDurationNs"]) / steps / 1e6, int(x["Calls"]) / steps, float(x["AverageNs"]) / 1e3,
                                                   re.sub(r"\(.*", "", x["Name"])[:90].replace("|", "/")))


def _counters(fname):
    d = {}
    f = os.path.join(P, fname)
    if os.path.exists(f):
        for ln in open(f):
            m = re.match(r"(\S+)\s+([0-9.e+]+) per launch", ln)
            if m:
                d[m.group(1)] = float(m.group(2))
    return d


def _pmc_per_kernel(tag):
    """{kernel name prefix: (launches, 2 x FETCH + WRITE bytes per launch)} of a FETCH_SIZE / WRITE_SIZE pair of passes."""
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = os.path.join(P, "%s_%spmc_%s.csv" % (RN, tag, c))
        if not os.path.exists(f):
            return {}
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:48]
            d = tot.setdefault(k, {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
            d[c][0] += float(r["Counter_Value"]) * 1024.0
            d[c][1] += 1
    return {k: (v["FETCH_SIZE"][1], 2.0 * v["FETCH_SIZE"][0] / max(v["FETCH_SIZE"][1], 1) + v["WRITE_SIZE"][0] / max(v["WRITE_SIZE"][1], 1))
            for k, v in tot.items()}


def _avg_us(stats_csv):
    d = {}
    f = os.path.join(P, stats_csv)
    if os.path.exists(f):
        for r in csv.DictReader(open(f)):
            d[re.sub(r"\(.*", "", r["Name"]).replace("void ", "")[:48]] = float(r["AverageNs"]) / 1e3
    return d


out.append("\n## Matrix-pipe occupancy and memory-side rate per kernel class\n\n")
for label, fn in (("`k_spconv_tile2` (128 -> 128 @ 82 107 rows, alone)", "%s_pmc_sq_tile_128.txt" % RN),
                  ("`k_spconv_pairs_wgrad_rows16` (same layer)", "%s_pmc_sq_wgrad_128.txt" % RN)):
    c = _counters(fn)
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0            # the counter sums the 8 XCDs
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
        wait = c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)
        conf = c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1)
        out.append("* %s: **MFMA pipe busy %.1f %%** of the SIMD cycles of the launch (`SQ_VALU_MFMA_BUSY_CYCLES` %.3g / (1 024 SIMDs x %.0f k cycles per XCD)), "
                   "waves waiting %.0f %% of their cycles, LDS bank-conflict cycles %.0f %% of the LDS cycles (`%s`)\n"
                   % (label, 100 * busy, c["SQ_VALU_MFMA_BUSY_CYCLES"], cyc / 1e3, 100 * wait, 100 * conf, fn))
out.append("\nMemory-side traffic per launch (2 x FETCH_SIZE + WRITE_SIZE, separate passes; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md) over the "
           "kernel's average duration in the kernel-stats table of the same configuration -- the HBM / Infinity-Fabric side rate of the conv kernels:\n\n")
for tag, cfgname, stats in (("", "ScanNet S50k x 4 (the benchmark)", "%s_bench_bf16_kernel_stats.csv" % RN),
                            ("s100kyaw8_", "SUN RGB-D S100k-yaw x 8", "%s_s100kyaw8_kernel_stats.csv" % RN),
                            ("s200k4_", "S200k x 4 @ 0.01 m", "%s_s200k4_kernel_stats.csv" % RN)):
    pk, av = _pmc_per_kernel(tag), _avg_us(stats)
    if not pk:
        continue
    out.append("| %s | launches (PMC pass) | MB / launch | avg us | GB/s |\n|---|---:|---:|---:|---:|\n" % cfgname)
    for k, (n, by) in sorted(pk.items(), key=lambda kv: -kv[1][1] * kv[1][0]):
        us = av.get(k)
        out.append("| `%s` | %d | %.1f | %s | %s |\n" % (k, n, by / 1e6, "%.1f" % us if us else "-", "%.0f" % (by / us / 1e3) if us else "-"))
    out.append("\n")

for name, what in (("%s_conv_shapes.txt" % RN, "per-shape conv table"), ("%s_pmc_sq_tile_128.txt" % RN, "SQ counters, tile kernel, 128->128 @ 82 107 rows"),
                   ("%s_pmc_sq_wgrad_128.txt" % RN, "SQ counters, weight gradient, same layer"), ("%s_pmc_l2_wgrad_128.txt" % RN, "L2 / memory-side counters, weight gradient"),
                   ("%s_tile_vs_dense_map.txt" % RN, "tile kernel vs the dense-map kernel per layer shape"), ("%s_bn_shapes.txt" % RN, "BatchNorm launches per shape, fp32 and bf16 row storage (tools/mb_bn16.py)"),
                   ("%s_s100kyaw8_conv_shapes.txt" % RN, "per-shape conv table, SUN RGB-D S100k-yaw x 8"),
                   ("%s_s200k4_conv_shapes.txt" % RN, "per-shape conv table, S200k x 4 at 0.01 m"),
                   ("%s_host_issue.txt" % RN, "host issue time vs step time"),
                   ("%s_tile_trace.txt" % RN, "per-workgroup trace of the tile kernel (dev build): effective clock, co-residency, time per phase, for 256 / 512 / 642 units"),
                   ("%s_tile_units_scaling.txt" % RN, "tile kernel time against the number of units in flight (one / two workgroups per CU, the tail round)"),
                   ("%s_tile_v1_knockout.txt" % RN, "round 2's tile kernel with its pieces knocked out (what motivated the rewrite)"),
                   ("%s_cpu_thread_scaling.txt" % RN, "thread scaling of the cpu_baseline leg (the oracle's S50k step) on the GPU host"),
                   ("%s_pmc_mem_tile_128.txt" % RN, "memory-side counters of the tile kernel, 128->128 @ 82 107 rows"),
                   ("%s_other_configs.txt" % RN, "bench lines of the other configurations and inference"),
                   ("%s_roi_contract.txt" % RN, "the per-RoI 7^3 contraction alone: library form vs cg3d_linear_fwd (stored partial products / atomics, by number of workgroups)"),
                   ("%s_host_sections.txt" % RN, "host wall-clock per section of the step, batch 1 (no GPU wait) and batch 4"),
                   ("%s_ops_by_site.txt" % RN, "framework ops, fills / copies, library GEMMs, reductions and sorts per source line; C-ABI calls per entry point (one step)"),
                   ("%s_sync_waits.txt" % RN, "every blocking read of the issuing thread: position in the step, time inside it (long = the host was ahead of the device, tens of us = the device was idle)"),
                   ("%s_gpu_gaps.txt" % RN, "kernel-trace timeline: launches per step, time with a kernel running on either queue, idle time by the launch that ended the gap"),
                   ("%s_backbone_lanes.txt" % RN, "device time of the backbone's forward / backward table on one stream and on their lanes (tools/backbone_lanes.py): DAPPM's branches with the coarse chain / on lanes 3, 2; weight gradients with their layer / on lane 2"),
                   ("%s_lane_timeline.txt" % RN, "one forward and one backward pass of the backbone under rocprofv3 --kernel-trace: kernel time per queue, time with more than one queue busy (tools/lane_timeline.py)"),
                   ("%s_lanes_ab.txt" % RN, "per-step wall time of the training step with and without lanes, alternating on one box (tools/ab_steps.sh -> tools/step_times.py, 100 steps each)"),
                   ("%s_backward_nodes.txt" % RN, "host time of the backward pass per autograd node (torch.profiler)"),
                   ("%s_class_branch_sections.txt" % RN, "host time and device tail per section of the class branches (a device sync per section)"),
                   ("%s_tile2_knockout.txt" % RN, "k_spconv_tile2 with parts knocked out at compile time: where a launch's time goes (DESIGN.md section 5, round 6)"),
                   ("%s_tile2_epilogue.txt" % RN, "the tile kernel's epilogue taken apart: statistics, exchange, stores, stagger, tail, 255-row passes"),
                   ("%s_tile2_stats_slots.txt" % RN, "statistics epilogue against the number of table slots: not a same-address queue"),
                   ("%s_tile2_early_stats.txt" % RN, "statistics taken from the accumulators right after the exchange"),
                   ("%s_pmc_sq_tile_128_stats_late.txt" % RN, "SQ counters of the tile kernel WITH the statistics epilogue, statistics at the end of the workgroup (round 5)"),
                   ("%s_pmc_sq_tile_128_stats_early.txt" % RN, "the same with the statistics taken right after the exchange (round 6)"),
                   ("%s_ab_bigk_two_wgs.txt" % RN, "two workgroups per CU for the 128-offset-block kernel (255-row plans): a loss"),
                   ("%s_ab_kb_k125.txt" % RN, "5^3 kernels on 128- vs 32-offset blocks"),
                   ("%s_tile2_setprio.txt" % RN, "multiply loop at wave priority 1"),
                   ("%s_ab_rocprof_tile.txt" % RN, "per-kernel totals of one box under five environments: early statistics, stage order, 128-offset table blocks, round 5's kernel"),
                   ("%s_ab_rocprof_tile_other_configs.txt" % RN, "the same A/B on S200k x 4 and S100k-yaw x 8"),
                   ("%s_ab_r05_vs_now.txt" % RN, "bench.py alternated between the round-5 build of the library and this round's"),
                   ("%s_wgrad_knockout.txt" % RN, "weight gradient without its atomic epilogue / with plain stores")):
    if os.path.exists(os.path.join(P, name)):
        out.append("\n### `%s` — %s\n\n```\n%s\n```\n" % (name, what, "\n".join(l for l in rd(name).splitlines() if "amdgpu.ids" not in l)[:6000]))
open(os.path.join(P, "README.md"), "w").write("".join(out))
print("".join(out)[:2500])
