# dev tool (GPU box): SQ counters of one kernel on one layer shape.
# usage: bash tools/pmc_sq.sh <kernel-regex> <script.py> [args]      e.g.  bash tools/pmc_sq.sh wgrad_rows16 tools/mb_wgrad_one.py 4 128 128
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; RE=$1; SC=$2; shift 2
O=$R/gpurun_out/pmc_sq_$RE.txt; : > $O
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/sq_$n
  timeout 300 rocprofv3 --pmc $c --kernel-include-regex "$RE" --output-format csv -d /tmp/sq_$n -o pmc -- python $R/$SC "$@" > /tmp/sq_$n.log 2>&1
  f=$(find /tmp/sq_$n -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "== $c: no output" >> $O; tail -5 /tmp/sq_$n.log >> $O; continue; fi
  python - "$f" >> $O <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Counter_Name"]; tot[k] += float(r["Counter_Value"]); n[k] += 1
for k in sorted(tot): print("%-32s %14.5g per launch (%d launches)" % (k, tot[k] / n[k], n[k]))
PY
done
cat $O
