# dev tool (GPU box): memory-side traffic (FETCH_SIZE x 2, WRITE_SIZE; calibrated by tools/pmc_calibrate.sh) and L2 hit rate of one
# kernel on one layer shape.  usage: bash tools/pmc_mem.sh <kernel-regex> <script.py> [args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; RE=$1; SC=$2; shift 2
O=$R/gpurun_out/pmc_mem_$RE.txt; : > $O
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pm_$n
  timeout 300 rocprofv3 --pmc $c --kernel-include-regex "$RE" --output-format csv -d /tmp/pm_$n -o pmc -- python $R/$SC "$@" > /tmp/pm_$n.log 2>&1
  f=$(find /tmp/pm_$n -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "== $c: no output" >> $O; tail -3 /tmp/pm_$n.log >> $O; continue; fi
  python - "$f" >> $O <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Counter_Name"]; tot[k] += float(r["Counter_Value"]); n[k] += 1
for k in sorted(tot): print("%-32s %14.5g per launch (%d launches)" % (k, tot[k] / n[k], n[k]))
PY
done
cat $O
