#!/bin/bash
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_x
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-fp32 --steps 10 --warmup 3 > /root/repo/gpurun_out/prof_x.log 2>&1
mkdir -p /root/repo/gpurun_out/prof_x
find /tmp/prof_x -name "*kernel_stats.csv" -exec cp {} /root/repo/gpurun_out/prof_x/kernel_stats.csv \;
tail -1 /root/repo/gpurun_out/prof_x.log | cut -c1-200
