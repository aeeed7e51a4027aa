"""bench.py -- scenes/s forward+backward(+optimiser step) of CAGroup3D on synthetic ScanNet-shaped
scenes (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one training iteration of the ScanNetV2 CAGroup3D.yaml configuration on a batch of 4
synthetic 50k-point scenes per GPU (BASELINE.json configs[1]): voxelisation (hash build), BiResNet,
CAGroup3DHead (+ stage-1 NMS, target assignment, kNN), CAGroup3DRoIHead, all losses, backward,
grad-norm clip, AdamW step; inputs are resident in HBM when the timed region starts.  Scenes shard
one batch per rank (weak scaling); the only collectives are DDP's gradient all-reduce and the
fused 3-scalar reduce_mean per scene, over RCCL.

Prints ONE JSON line (rank 0).  `roofline` is for the conv kernel that accumulates the most time (bf16:
k_spconv_implicit_bf16_ad, fp32: k_spconv_pairs_lds), every conv launch timed live with HIP events on the launch stream on
up to six of the timed steps spread over the timed region (`roofline.timed_steps`); `cpu_baseline` times the CPU oracle
(oracle/liboracle.so, the checker -- never the product) on a bounded sample, in a child process.
"""
import argparse
import json
import os

# One host thread for torch's CPU-side ops: every host tensor of a step is tiny, and with the default (one OpenMP
# thread per core) a descheduled worker of some small parallel region stalled the launching thread for 50-200 ms a few
# times per 100 steps -- mean step 54-57 ms vs 44.5 ms with one thread (tools/alloc_steady.py).  torchrun sets the same
# default for its workers.  The cpu_baseline leg runs in its own process with all cores.
os.environ.setdefault("OMP_NUM_THREADS", "1")
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from cagroup3d_amd import _lib, build_model, me  # noqa: E402
from cagroup3d_amd.hostpin import pin_host_threads  # noqa: E402

_ALL_CPUS = None

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
HBM_PEAK_GBS = 8000.0
BF16_MFMA_PEAK_TFLOPS = 2500.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="scenes per GPU (CAGroup3D.yaml BATCH_SIZE_PER_GPU)")
    ap.add_argument("--config", default="S50k")
    ap.add_argument("--dataset", default="scannet")
    ap.add_argument("--natural", action="store_true", help="untrained-net selection instead of forced GT selection")
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16",
                    help="MFMA operand type of the sparse-conv forward/data-gradient (BASELINE.json configs[1]: bf16 backbone); "
                         "accumulation, storage and the weight gradient are fp32 in both")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="S50k:1", help="config:scenes timed on the CPU oracle")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)   # the child process of the cpu_baseline leg
    return ap.parse_args()


def fresh(batch):
    b = dict(batch)
    b["points"] = batch["points"].clone()   # forward normalises colours in place
    return b


def make_model(dataset, forced, device):
    model, cfg = build_model.build_cagroup3d(dataset, seed=0)
    if forced:
        model.dense_head.force_gt_selection = True
        # trained-like stage-1 scores: the map of class c fires for class c, so proposals survive
        # SCORE_THR and every per-class NMS sees up to NMS_PRE candidates per scene
        model.dense_head.force_class_logit_boost = 6.0
    return model.to(device), cfg


_PARAMS = {}
_PREPARED = {}
PREFETCH = os.environ.get("CG3D_PREFETCH", "1") != "0"


def train_step(model, opt, batch, clip):
    params = _PARAMS.get(id(model))
    if params is None:                      # walking the module tree every step costs ~2 ms of host time
        params = _PARAMS[id(model)] = [p for p in model.parameters() if p.requires_grad]
    opt.zero_grad(set_to_none=True)
    b = fresh(batch)
    core = model.module if hasattr(model, "module") else model
    if PREFETCH and _PREPARED.get(id(core)) is not None:
        b["prepared"] = _PREPARED.pop(id(core))
    ret, tb, disp = model(b)
    ret["loss"].backward()
    if getattr(core, "grad_sync", None) is not None:
        core.grad_sync.finish()                 # early/mid buckets were sent from the backward pass, late bucket here
    torch.nn.utils.clip_grad_norm_(params, clip)
    opt.step()
    if PREFETCH:
        # the NEXT batch's coordinate structures (here: the same synthetic scenes again), on a side stream while the
        # GPU still works through the backward just queued -- every step builds them anew, nothing is reused
        _PREPARED[id(core)] = core.prefetch_coordinates(batch)
    return tb


def cpu_baseline(args, forced):
    """Same training step on the host CPU with the oracle library bound in place of the HIP one."""
    oracle_so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(oracle_so):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    cfgname, nsc = args.cpu_sample.split(":")
    nsc = int(nsc)
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    torch.set_num_threads(cores)
    prec, me.PRECISION = me.PRECISION, 0      # the CPU port computes in fp32 (bf16 emulation would only slow it down)
    with _lib.use_library(_lib.bind(oracle_so)):
        model, cfg = make_model(args.dataset, forced, "cpu")
        model.train()
        opt = torch.optim.AdamW(model.parameters(), lr=cfg.OPTIMIZATION.LR, weight_decay=cfg.OPTIMIZATION.WEIGHT_DECAY)
        batch = build_model.synthetic_batch(cfgname, nsc, device="cpu")
        t0 = time.time()
        train_step(model, opt, batch, cfg.OPTIMIZATION.GRAD_NORM_CLIP)
        dt = time.time() - t0
    me.PRECISION = prec
    return {"value": nsc / dt, "unit": "scenes/s", "cores": cores, "kind": "port",
            "sample": "%d scene(s) of %s, one fwd+bwd+AdamW step of the full detector on the CPU oracle "
                      "(OpenMP + torch CPU threads), %.1f s" % (nsc, cfgname, dt)}


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/r01_pmc_{FETCH,WRITE}_SIZE.csv; separate runs, as the profiling guide prescribes).
    gfx950 correction: FETCH_SIZE counts 128-byte requests as 64 B for 16-B/lane reads -> doubled."""
    import csv
    try:
        tot = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            rows = [r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r01_pmc_%s.csv" % c)))
                    if kernel_substr in r["Kernel_Name"]]
            if not rows:
                return None
            tot[c] = sum(float(r["Counter_Value"]) for r in rows) * 1024.0 / len(rows)
        return 2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]
    except Exception:
        return None


def cpu_baseline_subprocess(args):
    """The cpu_baseline leg in a child process with every core (this process runs torch's host ops on one thread)."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(min(os.cpu_count() or 1, 64)))   # the oracle stops scaling beyond 64
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-sample", args.cpu_sample,
           "--dataset", args.dataset] + (["--natural"] if args.natural else [])
    every = (lambda: os.sched_setaffinity(0, _ALL_CPUS)) if _ALL_CPUS else None      # the child is not pinned
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, preexec_fn=every)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not lines:
        raise RuntimeError("cpu_baseline child failed: %s" % out.stderr[-400:])
    return json.loads(lines[-1])


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args, not args.natural)), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    global _ALL_CPUS
    _ALL_CPUS, _ = pin_host_threads(local_rank)        # before the runtime starts its helper threads
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    if os.environ.get("CG3D_SINGLE_DEVICE") == "1":      # test aid: every rank on cuda:0 (with CG3D_DIST_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("CG3D_FORCE_DDP") == "1"      # the env knob exercises the DDP path on one GPU
    if use_dist:
        dist.init_process_group(backend=os.environ.get("CG3D_DIST_BACKEND", "nccl"))   # "nccl" = RCCL on ROCm
    forced = not args.natural
    me.PRECISION = 1 if args.precision == "bf16" else 0

    model, cfg = make_model(args.dataset, forced, dev)
    model.train()
    net = model
    if use_dist and os.environ.get("CG3D_TORCH_DDP") == "1":      # A/B: torch DDP (per-parameter hooks + bucket copies)
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], gradient_as_bucket_view=True)
    elif use_dist:
        from cagroup3d_amd.grad_sync import TwoBucketGradSync
        model.grad_sync = TwoBucketGradSync(model)   # flat buckets over RCCL, sent from inside the backward pass
    opt = torch.optim.AdamW(model.parameters(), lr=cfg.OPTIMIZATION.LR, weight_decay=cfg.OPTIMIZATION.WEIGHT_DECAY,
                            fused=True)      # one multi-tensor launch set for the whole update
    clip = cfg.OPTIMIZATION.GRAD_NORM_CLIP
    # every rank owns different scenes (scene i -> rank i mod W), fixed across steps
    batch = build_model.synthetic_batch(args.config, args.batch, first_scene=rank * args.batch, device=dev)

    for _ in range(args.warmup):
        tb = train_step(net, opt, batch, clip)
    # the model, the optimizer state and the cached tables are permanent: take them out of the cyclic collector's
    # generations, or every gen-2 pass walks them again (measured: one 75 ms pause per ~100 steps)
    import gc
    gc.collect()
    gc.freeze()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # live HIP-event timing of every conv launch (the roofline figures) -- on at most ~6 of the timed steps, spread evenly
    # over the timed region: two events per launch add up (20 k live events in a 100-step run slowed the run itself)
    me.KernelProfile.reset()
    stride = max(1, -(-args.steps // 6))
    profiled_steps = 0
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        me.KernelProfile.enabled = i % stride == 0
        profiled_steps += int(me.KernelProfile.enabled)
        tb = train_step(net, opt, batch, clip)
    barrier()
    dt = time.perf_counter() - t0
    me.KernelProfile.enabled = False
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        kinds = me.KernelProfile.summary()
        kind = max(kinds, key=lambda k: kinds[k]["ms"])          # the dominant conv kernel of this run
        prof = kinds[kind]
        secs = prof["ms"] * 1e-3
        tf = prof["flops"] / secs / 1e12 if secs > 0 else 0.0
        gbs = prof["bytes"] / secs / 1e9 if secs > 0 else 0.0
        kname, ksym = {
            "tile_bf16": ("k_spconv_tile (sparse conv fwd + dgrad on LDS-staged neighbour tiles: persistent workgroups, loader waves "
                          "gather the tile's distinct rows into LDS, consumer waves run bf16 MFMA from LDS with fragment-ordered "
                          "weights streamed from L2, one store per output row)", "k_spconv_tile"),
            "implicit_bf16": ("k_spconv_implicit_bf16_ad (sparse conv fwd + dgrad, output-stationary: neighbour rows -> "
                              "registers -> bf16 MFMA -> one store per output row)", "k_spconv_implicit_bf16"),
            "pairs_bf16": ("k_spconv_pairs_bf16 (sparse conv fwd + dgrad: gather -> bf16 MFMA -> atomic scatter)",
                           "k_spconv_pairs_bf16"),
            "pairs": ("k_spconv_pairs_lds (sparse conv fwd + dgrad: gather -> fp32 MFMA -> atomic scatter)",
                      "k_spconv_pairs_lds"),
        }[kind]
        if me.PRECISION == 1:
            # bf16 MFMA runs at 16x the fp32 rate: the kernel is bound by the row gather (+ row scatter)
            roof = {"kernel": kname, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBS, "traffic": None, "achieved_tflops": tf,
                    "frac_of_bf16_mfma_peak": tf / BF16_MFMA_PEAK_TFLOPS}
        else:
            roof = {"kernel": kname, "bound": "mfma", "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": tf / FP32_MFMA_PEAK_TFLOPS, "traffic": None, "algorithmic_gbytes_per_s": gbs}
        roof.update(launches=prof["launches"], avg_launch_ms=prof["ms"] / max(prof["launches"], 1),
                    kernel_time_share=secs / (dt * profiled_steps / args.steps), timed_steps=profiled_steps,
                    algorithmic_bytes_per_launch=prof["bytes"] / max(prof["launches"], 1),
                    other_conv_kernels={k: {"launches": v["launches"], "ms_per_step": v["ms"] / profiled_steps}
                                        for k, v in kinds.items() if k != kind})
        roof["traffic"] = pmc_traffic(ksym)   # bytes per launch, from profiles/ (separate --pmc runs)
        # SURVEY 8(d) aggregate over every timed conv launch (forward + data gradient): sum of per-launch roofline
        # bounds max(flops / MFMA peak, bytes / HBM peak) over the sum of measured times
        mfma_peak = (BF16_MFMA_PEAK_TFLOPS if me.PRECISION == 1 else FP32_MFMA_PEAK_TFLOPS) * 1e12
        bound_s = sum(max(r[2] / mfma_peak, r[3] / (HBM_PEAK_GBS * 1e9)) for r in me.KernelProfile.records)
        meas_s = sum(v["ms"] for v in kinds.values()) * 1e-3
        roof["conv_fwd_dgrad_bound_over_measured"] = bound_s / meas_s if meas_s > 0 else None
        out = {"metric": "scenes/s fwd+bwd ScanNet ~50k pts", "value": world * args.batch * args.steps / dt,
               "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16" if me.PRECISION == 1 else "f32", "data": "synthetic",
               "config": {"workload": "%s CAGroup3D.yaml, %d x %s scenes per GPU, %s" % (
                   {"scannet": "ScanNetV2", "sunrgbd": "SUN RGB-D"}.get(args.dataset, args.dataset), args.batch, args.config, "forced GT selection (replaces the net's own selection: sizes independent of the weights) + own-class logit boost (trained-like loads)" if forced
                   else "natural selection of the untrained net"),
                          "scenes_per_gpu": args.batch, "points_per_scene": 50000 if args.config == "S50k" else args.config,
                          "voxel_size_m": float(model.voxel_size), "parallelism": "dp%d" % world, "optimizer": "AdamW+clip10",
                          "precision": ("bf16 MFMA operands in conv fwd/dgrad, fp32 accumulate/storage/wgrad"
                                        if me.PRECISION == 1 else "fp32 everywhere (parity configuration)"),
                          "last_loss": tb.get("loss_all")},
               "roofline": roof}
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline_subprocess(args)
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "scenes/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
