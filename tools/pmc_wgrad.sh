# dev tool (GPU box): L2 / memory-side counters of the bf16 weight-gradient kernel on one layer shape, with and without the
# XCD-aware segment order.  usage: bash tools/pmc_wgrad.sh [ts cin cout]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_wgrad.txt; : > $O
for x in 1; do
echo "=== CG3D_WGRAD_ROW_BLOCKS=${RB:-1}" >> $O
for c in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pw_$n
  CG3D_WGRAD_ROW_BLOCKS=${RB:-1} timeout 300 rocprofv3 --pmc $c --kernel-include-regex "wgrad_rows16" --output-format csv -d /tmp/pw_$n -o pmc -- python $R/tools/mb_wgrad_one.py "$@" > /tmp/pw_$n.log 2>&1
  f=$(find /tmp/pw_$n -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "== $c: no output" >> $O; tail -3 /tmp/pw_$n.log >> $O; continue; fi
  python - "$f" >> $O <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Counter_Name"]; tot[k] += float(r["Counter_Value"]); n[k] += 1
for k in sorted(tot): print("%-32s %14.5g per launch (%d launches)" % (k, tot[k] / n[k], n[k]))
PY
  grep "wgrad ts" /tmp/pw_$n.log | tail -1 >> $O
done
done
cat $O
