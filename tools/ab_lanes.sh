cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lanes
timeout 900 python -m pytest tests/test_engine.py -x -q -m gpu > gpurun_out/lanes/test_engine.log 2>&1; echo "test_engine rc=$?" 
tail -3 gpurun_out/lanes/test_engine.log
for i in 1 2; do
for L in 0 1; do
CG3D_LANES=$L timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --rotate 0 > gpurun_out/lanes/bench_L${L}_$i.json 2> gpurun_out/lanes/bench_L${L}_$i.err
python - <<P
import json
try:
    d=json.loads(open("gpurun_out/lanes/bench_L${L}_$i.json").read().strip().splitlines()[-1])
    print("LANES=$L run $i:", d["value"], d["ms_per_step"], {k:(v.get("value") if isinstance(v,dict) else v) for k,v in d.items() if k in ("bf16_backbone_fp32_heads","bf16_all_convolutions")})
except Exception as e:
    print("LANES=$L run $i failed", e); print(open("gpurun_out/lanes/bench_L${L}_$i.err").read()[-2000:])
P
done; done
