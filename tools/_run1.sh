cd /root/repo
python tools/mb_implicit.py > gpurun_out/r2_mb_base.log 2>&1
SORT=morton python tools/mb_implicit.py > gpurun_out/r2_mb_morton.log 2>&1
python bench.py --no-cpu-baseline > gpurun_out/r2_bench_base.log 2>&1
tail -3 gpurun_out/r2_mb_base.log gpurun_out/r2_mb_morton.log
