# dev tool (GPU box): calibrates FETCH_SIZE / WRITE_SIZE on a kernel whose traffic is known -- k_bn_apply reads N*C*4 bytes and
# writes N*C*(4+2) (fp32 rows + bf16 copy), 16 bytes per lane, full lines.  usage: bash tools/pmc_calibrate.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_calibrate.txt; : > $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c
  timeout 300 rocprofv3 --pmc $c --kernel-include-regex "k_bn_apply" --output-format csv -d /tmp/cal_$c -o pmc -- python $R/tools/mb_bn.py > /tmp/cal_$c.log 2>&1
  f=$(find /tmp/cal_$c -name '*counter_collection.csv' | head -1)
  python - "$f" $c >> $O <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Grid_Size"]].append(float(r["Counter_Value"]))
shapes = {155773 * 64: "155773x64", 82107 * 128: "82107x128", 23015 * 256: "23015x256"}
for g, v in sorted(agg.items(), key=lambda kv: -int(kv[0])):
    print("%s grid %8s: %9.1f KB per launch (%d launches; min %.0f max %.0f)" % (sys.argv[2], g, sum(v) / len(v), len(v), min(v), max(v)))
PY
done
echo "expected per launch (no residual / with residual): 155773x64: read 39.9 / 79.8 MB, write 59.8 MB; 82107x128: read 42.0 / 84.1 MB, write 63.1 MB; 23015x256: read 23.6 / 47.1, write 35.4" >> $O
cat $O
