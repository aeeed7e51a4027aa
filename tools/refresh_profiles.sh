#!/bin/bash
# Runs on the GPU box (gpurun): regenerates everything profiles/ holds for this round.  Outputs land in gpurun_out/refresh/;
# copy them into profiles/ afterwards (tools/make_profiles_readme.py rebuilds the README from them).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/refresh; mkdir -p $O; RN=${ROUND:-r06}
cd /tmp; export TMPDIR=/tmp
python $R/tools/stream_bw.py > $O/${RN}_stream_bw.txt 2>&1
python $R/bench.py > $O/bench_bf16.log 2>&1; tail -1 $O/bench_bf16.log > $O/${RN}_bench_bf16.json
python $R/bench.py --precision fp32 --no-cpu-baseline --no-fp32 > $O/bench_fp32.log 2>&1; tail -1 $O/bench_fp32.log > $O/${RN}_bench_fp32.json
for prec in bf16 fp32; do
  rm -rf /tmp/prof_$prec
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$prec -o bench -- python $R/bench.py --no-cpu-baseline --no-fp32 --precision $prec > $O/rocprof_$prec.log 2>&1
  find /tmp/prof_$prec -name "*kernel_stats.csv" -exec cp {} $O/${RN}_bench_${prec}_kernel_stats.csv \;
done
# the same run with every launch program compiled without lanes: each kernel alone on the one stream -- the averages bench.py's
# `roofline` (timed on steps compiled that way) is to be compared with
rm -rf /tmp/prof_l0
CG3D_LANES=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l0 -o bench -- python $R/bench.py --no-cpu-baseline --no-fp32 > $O/rocprof_lanes0.log 2>&1
tail -1 $O/rocprof_lanes0.log | grep '^{' > $O/${RN}_bench_bf16_lanes0_under_rocprof.json
find /tmp/prof_l0 -name "*kernel_stats.csv" -exec cp {} $O/${RN}_bench_bf16_lanes0_kernel_stats.csv \;
# lanes: device time of the backbone passes on one queue / on their lanes, the per-queue timeline of one pass, the step A/B
( for d in "" "3,2"; do for w in 0 2; do python $R/tools/backbone_lanes.py --wgrad-lanes $w --dappm-lanes "$d" 2>&1 | grep "DAPPM lanes"; done; done ) > $O/${RN}_backbone_lanes.txt 2>&1
rm -rf /tmp/bl; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/bl -o bl -- python $R/tools/backbone_lanes.py --marks --iters 2 --wgrad-lanes 2 --dappm-lanes 3,2 > /dev/null 2>&1
python $R/tools/lane_timeline.py $(find /tmp/bl -name "bl_kernel_trace.csv" | head -1) --step -1 > $O/${RN}_lane_timeline.txt 2>&1
( cd $R; STEPS=100 bash tools/ab_steps.sh 2 "-" "CG3D_LANES=0" ) > $O/${RN}_lanes_ab.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-include-regex "k_spconv_(tile|implicit_bf16|pairs_bf16|pairs_wgrad_rows16)" --output-format csv -d /tmp/pmc_$c -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32 > $O/pmc_$c.log 2>&1
  find /tmp/pmc_$c -name "*counter_collection.csv" -exec cp {} $O/${RN}_pmc_$c.csv \;
done
python $R/tools/conv_shapes.py --wgrad > $O/${RN}_conv_shapes.txt 2>&1
bash $R/tools/pmc_sq.sh k_spconv_tile tools/mb_tile_one.py 4 128 128 > /dev/null 2>&1; cp $R/gpurun_out/pmc_sq_k_spconv_tile.txt $O/${RN}_pmc_sq_tile_128.txt
bash $R/tools/pmc_sq.sh wgrad_rows16 tools/mb_wgrad_one.py 4 128 128 > /dev/null 2>&1; cp $R/gpurun_out/pmc_sq_wgrad_rows16.txt $O/${RN}_pmc_sq_wgrad_128.txt
bash $R/tools/pmc_wgrad.sh 4 128 128 > /dev/null 2>&1; cp $R/gpurun_out/pmc_wgrad.txt $O/${RN}_pmc_l2_wgrad_128.txt
python $R/tools/mb_tile.py > $O/${RN}_tile_vs_dense_map.txt 2>&1
python $R/tools/mb_bn16.py > $O/${RN}_bn_shapes.txt 2>&1
python $R/tools/host_profile.py 2>&1 | head -12 > $O/${RN}_host_issue.txt
ls -la $O
bash $R/tools/pmc_mem.sh k_spconv_tile tools/mb_tile_one.py 4 128 128 > /dev/null 2>&1; cp $R/gpurun_out/pmc_mem_k_spconv_tile.txt $O/${RN}_pmc_mem_tile_128.txt
bash $R/tools/pmc_calibrate.sh > /dev/null 2>&1; cp $R/gpurun_out/pmc_calibrate.txt $O/${RN}_pmc_calibrate.txt
ls -la $O
python $R/tools/mb_roi_contract.py > $O/${RN}_roi_contract.txt 2>&1
python $R/tools/torch_ops_by_site.py > $O/${RN}_ops_by_site.txt 2>&1
ls -la $O
python $R/tools/host_sections.py > $O/${RN}_host_sections.txt 2>&1; BATCH=4 python $R/tools/host_sections.py >> $O/${RN}_host_sections.txt 2>&1
python $R/tools/sync_waits.py > $O/${RN}_sync_waits.txt 2>&1
python $R/tools/backward_nodes.py 2>/dev/null > $O/${RN}_backward_nodes.txt
python $R/tools/class_branch_sections.py > $O/${RN}_class_branch_sections.txt 2>&1
# rocprofv3 kernel trace of the default run: idle time by the launch that ended the gap
rm -rf /tmp/tr; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o bench -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-fp32 --rotate 0 > /dev/null 2>&1
python $R/tools/gpu_gaps.py $(find /tmp/tr -name "*kernel_trace.csv" | head -1) > $O/${RN}_gpu_gaps.txt 2>&1
# the other configurations (BASELINE.json configs[3] / configs[4] at the one-GPU size): kernel statistics, per-shape conv table,
# memory-side counters of the conv kernels
for spec in "sunrgbd S100k-yaw 8 s100kyaw8" "scannet S200k 4 s200k4"; do
  set -- $spec; DS=$1; CF=$2; BT=$3; TAG=$4
  rm -rf /tmp/prof_$TAG
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $R/bench.py --dataset $DS --config $CF --batch $BT --steps 8 --warmup 3 --no-cpu-baseline --no-fp32 --rotate 0 > $O/rocprof_$TAG.log 2>&1
  find /tmp/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $O/${RN}_${TAG}_kernel_stats.csv \;
  tail -1 $O/rocprof_$TAG.log | grep '^{' > $O/${RN}_${TAG}_bench_under_rocprof.json
  DATASET=$DS CFG=$CF BATCH=$BT python $R/tools/conv_shapes.py --wgrad > $O/${RN}_${TAG}_conv_shapes.txt 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${TAG}_$c
    timeout 600 rocprofv3 --pmc $c --kernel-include-regex "k_spconv_(tile|implicit_bf16|pairs_bf16|pairs_wgrad_rows16)" --output-format csv -d /tmp/pmc_${TAG}_$c -o pmc -- python $R/bench.py --dataset $DS --config $CF --batch $BT --steps 2 --warmup 1 --no-cpu-baseline --no-fp32 --rotate 0 > /dev/null 2>&1
    find /tmp/pmc_${TAG}_$c -name "*counter_collection.csv" -exec cp {} $O/${RN}_${TAG}_pmc_$c.csv \;
  done
done
# other configurations and inference, one line each
( echo "# other configurations at the end of round ${RN#r0} (bf16 backbone with bf16 rows + split heads unless stated; one box, one call)"
for a in "--dataset sunrgbd --config S100k-yaw --batch 8" "--config S200k" "--natural" "--batch 8" "--batch 2" "--batch 1" "--head-precision fp32" "--head-precision bf16"; do
  echo "bench.py $a"
  python $R/bench.py $a --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --rotate 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   %.1f scenes/s  %.1f ms/step  roofline.frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"
done
echo "CG3D_ACT_BF16=0 bench.py   (backbone rows stored as fp32: the arithmetic of rounds 1-4)"
CG3D_ACT_BF16=0 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --rotate 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   %.1f scenes/s  %.1f ms/step  roofline.frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"
echo "tools/eval_bench.py"; python $R/tools/eval_bench.py 2>/dev/null | tail -1 ) 2>/dev/null > $O/${RN}_other_configs.txt
cat $O/${RN}_other_configs.txt
