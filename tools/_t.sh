cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pp
rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 3 > /tmp/pp.log 2>&1
f=$(find /tmp/pp -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/gap_analysis.py $f 2>&1 | head -60
