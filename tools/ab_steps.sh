#!/bin/bash
# A/B by per-step wall times (tools/step_times.py: mean / median / p10 / p90 over STEPS steps), environments alternating inside
# ONE gpurun call:   tools/ab_steps.sh <rounds> "<env A>" "<env B>" ...   ("-": no variables)
cd ${GRAFT_REPO_ROOT:-.}
rounds=$1; shift
for i in $(seq 1 $rounds); do
  for e in "$@"; do
    [ "$e" = "-" ] && ee="" || ee="$e"
    echo -n "[$e] run $i: "; env $ee STEPS=${STEPS:-120} timeout 600 python tools/step_times.py 2>&1 | grep "^mean" | cut -c1-60
  done
done
