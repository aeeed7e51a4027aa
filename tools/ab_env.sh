#!/bin/bash
# A/B of bench.py under different environments, alternating inside ONE gpurun call (boxes differ by +-10 %):
#   tools/ab_env.sh <rounds> "<env A>" "<env B>" ...     (an env is a space-separated list of NAME=value, or "-" for none)
# prints scenes/s and ms/step per run; the JSON lines go to gpurun_out/ab/.
cd ${GRAFT_REPO_ROOT:-.}
rounds=$1; shift
mkdir -p gpurun_out/ab
for i in $(seq 1 $rounds); do
  k=0
  for e in "$@"; do
    k=$((k+1))
    [ "$e" = "-" ] && ee="" || ee="$e"
    env $ee timeout 600 python bench.py --steps ${AB_STEPS:-20} --warmup 5 --no-cpu-baseline --no-fp32 --rotate 0 ${AB_ARGS} > gpurun_out/ab/v${k}_$i.json 2> gpurun_out/ab/v${k}_$i.err
    python - "$e" $i gpurun_out/ab/v${k}_$i <<'P'
import json, sys
e, i, base = sys.argv[1:4]
try:
    d = json.loads(open(base + ".json").read().strip().splitlines()[-1])
    print("[%s] run %s: %.1f scenes/s  %.2f ms/step" % (e, i, d["value"], d["ms_per_step"]))
except Exception as ex:
    print("[%s] run %s FAILED: %s" % (e, i, ex)); print(open(base + ".err").read()[-1500:])
P
  done
done
