cd /root/repo
timeout 600 python -m pytest tests/test_tile_conv.py -x -q -m gpu > gpurun_out/r2_tile_tests.log 2>&1
timeout 300 python tools/mb_tile.py > gpurun_out/r2_mb_tile.log 2>&1
timeout 300 python tools/mb_tile_dbg.py > gpurun_out/r2_tile_dbg.log 2>&1
tail -n 5 gpurun_out/r2_tile_tests.log
tail -n 14 gpurun_out/r2_mb_tile.log
cat gpurun_out/r2_tile_dbg.log
