cd /root/repo
timeout 600 python -m pytest tests/test_tile_conv.py -x -q -m gpu > gpurun_out/r2_tile_tests.log 2>&1
timeout 300 python tools/mb_tile.py > gpurun_out/r2_mb_tile.log 2>&1
timeout 300 python tools/mb_tile_phases.py > gpurun_out/r2_phases.log 2>&1
tail -n 5 gpurun_out/r2_tile_tests.log
tail -n 13 gpurun_out/r2_mb_tile.log | cut -c1-150
cat gpurun_out/r2_phases.log
