cd /tmp; export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/prof_mid; mkdir -p $O
rm -rf /tmp/prof_bf16
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bf16 -o bench -- python $R/bench.py --no-cpu-baseline > $O/rocprof.log 2>&1
find /tmp/prof_bf16 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
tail -1 $O/rocprof.log | cut -c1-200
head -60 $O/kernel_stats.csv | cut -c1-170
