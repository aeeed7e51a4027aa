#!/bin/bash
# dev tool (GPU box): bench.py alternated between builds of the HIP library on ONE box (boxes differ by +-5 %).
# usage: bash tools/ab_libs.sh <rounds> <lib A> <lib B> [...]     ("-" = the in-tree build)   [AB_STEPS=40] [AB_ARGS="--rotate 0"]
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$1; shift; mkdir -p gpurun_out/ab
for i in $(seq 1 $R); do
  k=0
  for lib in "$@"; do
    k=$((k+1))
    if [ "$lib" = "-" ]; then unset CG3D_HIP_LIB; else export CG3D_HIP_LIB=$(realpath $lib); fi
    timeout 600 python bench.py --steps ${AB_STEPS:-40} --warmup 8 --no-cpu-baseline --no-fp32 ${AB_ARGS} 2>gpurun_out/ab/lib${k}_$i.err | tail -1 > gpurun_out/ab/lib${k}_$i.json
    python - "$lib" gpurun_out/ab/lib${k}_$i.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    fb = d.get("fixed_batch") or {}
    print("%-40s %7.1f scenes/s %6.2f ms/step | fixed batch %7.1f | tile frac %.3f avg %.1f us" % (sys.argv[1], d["value"], d["ms_per_step"], fb.get("value", 0.0), d["roofline"]["frac"], d["roofline"]["avg_launch_ms"] * 1e3))
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
  done
done
