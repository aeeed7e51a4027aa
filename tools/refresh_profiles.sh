#!/bin/bash
# Runs on the GPU box (gpurun): regenerates everything profiles/ holds.  Outputs land in gpurun_out/refresh/.
set -x
R=/root/repo; O=$R/gpurun_out/refresh; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python $R/tools/stream_bw.py > $O/stream_bw.txt 2>&1
python $R/bench.py > $O/bench_bf16.log 2>&1; tail -1 $O/bench_bf16.log > $O/r01_bench_bf16.json
python $R/bench.py --precision fp32 --no-cpu-baseline > $O/bench_fp32.log 2>&1; tail -1 $O/bench_fp32.log > $O/r01_bench_fp32.json
for prec in bf16 fp32; do
  rm -rf /tmp/prof_$prec
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$prec -o bench -- python $R/bench.py --no-cpu-baseline --precision $prec > $O/rocprof_$prec.log 2>&1
  find /tmp/prof_$prec -name "*kernel_stats.csv" -exec cp {} $O/r01_bench_${prec}_kernel_stats.csv \;
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-include-regex "k_spconv_(implicit_bf16|pairs_bf16|pairs_wgrad_rows16)" --output-format csv -d /tmp/pmc_$c -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_$c.log 2>&1
  find /tmp/pmc_$c -name "*counter_collection.csv" -exec cp {} $O/r01_pmc_$c.csv \;
done
ls -la $O
