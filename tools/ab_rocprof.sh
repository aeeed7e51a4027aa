#!/bin/bash
# dev tool (GPU box): rocprofv3 kernel statistics of a short fixed-batch bench.py run under several environments; prints the rows of
# the kernels matching $AB_KERNELS per environment.   usage: bash tools/ab_rocprof.sh "<env A>" "<env B>" ...   ("-" = none)
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; K=${AB_KERNELS:-k_spconv_tile2}
i=0
for e in "$@"; do
  i=$((i+1)); [ "$e" = "-" ] && ee="" || ee="$e"
  rm -rf /tmp/abp_$i
  env $ee CG3D_LANES=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp_$i -o b -- python $R/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-fp32 --rotate 0 ${AB_ARGS} > /tmp/abp_$i.log 2>&1
  f=$(find /tmp/abp_$i -name "*kernel_stats.csv" | head -1)
  echo "== [$e]  $(tail -1 /tmp/abp_$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f scenes/s %.2f ms/step (under rocprof, lanes off)' % (d['value'], d['ms_per_step']))" 2>/dev/null)"
  python - "$f" "$K" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    if any(k in r["Name"] for k in sys.argv[2].split(",")):
        n = r["Name"].split("(")[0].replace("void ", "")
        print("   %-44s calls %5s  avg %8.1f us  total %8.2f ms" % (n[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
        tot += float(r["TotalDurationNs"]) / 1e6
print("   sum %.2f ms;  all kernels %.2f ms" % (tot, sum(float(r["TotalDurationNs"]) for r in rows) / 1e6))
PY
done
