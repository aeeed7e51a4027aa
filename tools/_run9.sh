cd /root/repo
python bench.py --no-cpu-baseline > gpurun_out/r2_bench_a.log 2>&1
tail -n 1 gpurun_out/r2_bench_a.log | cut -c1-330
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2_gpu_tests.log 2>&1
tail -n 8 gpurun_out/r2_gpu_tests.log
