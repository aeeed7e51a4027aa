cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2_gpu_tests.log 2>&1
tail -n 8 gpurun_out/r2_gpu_tests.log
python bench.py --no-cpu-baseline > gpurun_out/r2_bench_a.log 2>&1
tail -n 1 gpurun_out/r2_bench_a.log | cut -c1-400
CG3D_TILE_KERNEL=0 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_b.log 2>&1
tail -n 1 gpurun_out/r2_bench_b.log | cut -c1-400
CG3D_TILE_KERNEL=0 CG3D_MORTON_ROWS=0 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_c.log 2>&1
tail -n 1 gpurun_out/r2_bench_c.log | cut -c1-400
