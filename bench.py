"""bench.py -- scenes/s forward+backward(+optimiser step) of CAGroup3D on synthetic ScanNet-shaped
scenes (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5        (the defaults)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one training iteration of the ScanNetV2 CAGroup3D.yaml configuration on a batch of 4
synthetic 50k-point scenes per GPU (BASELINE.json configs[1]): voxelisation (hash build), BiResNet,
CAGroup3DHead (+ stage-1 NMS, target assignment, kNN), CAGroup3DRoIHead, all losses, backward,
grad-norm clip, AdamW step; inputs are resident in HBM when the timed region starts.  Scenes shard
one batch per rank (weak scaling); the only collectives are DDP's gradient all-reduce and the
fused 3-scalar reduce_mean per scene, over RCCL.

Prints ONE JSON line (rank 0).  `roofline` is for the conv kernel that accumulates the most time (bf16: k_spconv_tile),
every conv launch (forward, data gradient, weight gradient) timed live with HIP events on the launch stream on two of
the timed steps spread over the timed region (`roofline.timed_steps`; those steps' launch programs are compiled without lanes,
so that a launch is timed alone -- `roofline.queues`, `roofline.on_lanes`).  Work per launch follows SURVEY.md 8(d): flops =
2 P Cin Cout, bytes = every tensor once (input rows + output rows + weights + the map); the bound of a launch is
max(flops / MFMA peak, bytes / HBM peak), `frac` = achieved / peak of whichever bounds the dominant kernel, and
`frac_8d_per_layer` = sum of the launches' bounds / sum of their measured times.  `fp32` is the parity configuration timed
in the same process; `cpu_baseline` times the CPU oracle (oracle/liboracle.so, the checker -- never the product) on a bounded
sample in child processes: 2 warm-up steps + the median of 5, all host cores and one thread.
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4, help="scenes per GPU (CAGroup3D.yaml BATCH_SIZE_PER_GPU)")
    ap.add_argument("--config", default="S50k")
    ap.add_argument("--dataset", default="scannet")
    ap.add_argument("--natural", action="store_true", help="untrained-net selection instead of forced GT selection")
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16",
                    help="MFMA operand type of the sparse-conv forward/data-gradient (BASELINE.json configs[1]: bf16 backbone); "
                         "accumulation, storage and the weight gradient are fp32 in both")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="S50k:4", help="config:scenes per step timed on the CPU oracle (the GPU's own batch)")
    ap.add_argument("--head-precision", choices=["split", "bf16", "fp32"], default="split",
                    help="arithmetic of the two heads' convolutions (class branches, RoI pooling, 1x1x1 layers) in the bf16 run.  "
                         "BASELINE.json configs[1] says \"bf16 backbone\" and the reference's heads are fp32: split (default) = "
                         "fp32-accurate products from three bf16 MFMA passes on split operands (me.PREC_SPLIT), fp32 = fp32 MFMA "
                         "operands, bf16 = bf16 operands in the heads too (a wider precision scope than configs[1] names).  The "
                         "two modes not chosen are timed as sub-records")
    ap.add_argument("--cpu-sample-1t", default="S5k:1", help="config:scenes of the single-thread CPU leg")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 sub-record")
    ap.add_argument("--rotate", type=int, default=8,
                    help="the headline cycles K DIFFERENT synthetic batches inside the timed region (map sizes, cache keys, allocator "
                         "requests and the class branches' loads change from step to step, as in training); the one fixed batch of "
                         "rounds 1-5 is timed as the sub-record `fixed_batch`.  0 / 1: the headline repeats the one fixed batch")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)   # the child process of the cpu_baseline leg
    return ap.parse_args()


def fresh(batch):
    b = dict(batch)
    b["points"] = batch["points"].clone()   # forward normalises colours in place
    return b


def make_model(dataset, forced, device, voxel_size=None):
    model, cfg = build_model.build_cagroup3d(dataset, seed=0, voxel_size=voxel_size)
    if forced:
        model.dense_head.force_gt_selection = True
        # trained-like stage-1 scores: the map of class c fires for class c, so proposals survive
        # SCORE_THR and every per-class NMS sees up to NMS_PRE candidates per scene
        model.dense_head.force_class_logit_boost = 6.0
    return model.to(device), cfg


_PARAMS = {}
_PREPARED = {}
PREFETCH = os.environ.get("CG3D_PREFETCH", "1") != "0"
PREFETCH_THREAD = os.environ.get("CG3D_PREFETCH_THREAD", "1") != "0"     # the dry run of the next batch on a worker thread


def train_step(model, opt, batch, clip, next_batch=None):
    """next_batch: the batch the FOLLOWING step will run (--rotate): its coordinate dry run is submitted here."""
    params = _PARAMS.get(id(model))
    if params is None:                      # walking the module tree every step costs ~2 ms of host time
        params = _PARAMS[id(model)] = [p for p in model.parameters() if p.requires_grad]
    opt.zero_grad(set_to_none=True)
    b = fresh(batch)
    core = model.module if hasattr(model, "module") else model
    nxt = batch if next_batch is None else next_batch
    if PREFETCH and _PREPARED.get(id(core)) is not None:
        handle, owner = _PREPARED.pop(id(core))
        prepared = handle.result()
        if owner is batch:                      # (a dry run made for another batch is dropped: this step builds its own)
            b["prepared"] = prepared
    if PREFETCH and PREFETCH_THREAD and batch["points"].is_cuda:
        # the NEXT batch's coordinate structures (fixed-batch runs: the same synthetic scenes again -- every step builds them
        # anew, nothing is reused) on the worker thread and the side stream, while this step is issued and runs
        _PREPARED[id(core)] = (core.prefetch_coordinates_async(nxt), nxt)
    ret, tb, disp = model(b)
    ret["loss"].backward()
    if getattr(core, "grad_sync", None) is not None:
        core.grad_sync.finish()                 # early/mid buckets were sent from the backward pass, late bucket here
    if hasattr(opt, "clip_and_step"):
        opt.clip_and_step(clip)                 # clip_grad_norm_ + fused AdamW, the same torch kernels on cached lists
    else:
        torch.nn.utils.clip_grad_norm_(params, clip)
        opt.step()
    if PREFETCH and not (PREFETCH_THREAD and batch["points"].is_cuda):
        # single-threaded variant: on the side stream while the GPU still works through the backward just queued
        from cagroup3d_amd.pcdet.models.detectors.cagroup3d import _Done
        _PREPARED[id(core)] = (_Done(core.prefetch_coordinates(nxt)), nxt)
    return tb


def finish_prefetch(model):
    """Wait for (and drop) the dry run submitted by the last step: nothing of this process may still be launching when the
    measurement ends."""
    core = model.module if hasattr(model, "module") else model
    h = _PREPARED.pop(id(core), None)
    if h is not None:
        h[0].result()
    w = getattr(core, "_prefetch_worker", None)
    if w is not None:
        w.close()


def cpu_baseline(args, forced):
    """Same training step on the host CPU with the oracle library bound in place of the HIP one: BASELINE.md section 2
    protocol -- 2 warm-up steps, then the median of up to 5 timed steps, bounded by a time budget (the repeats actually
    taken are stated in `sample`)."""
    oracle_so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(oracle_so):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    cfgname, nsc = args.cpu_sample.split(":")
    nsc = int(nsc)
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    budget = float(os.environ.get("CG3D_CPU_BUDGET_S", "45"))
    torch.set_num_threads(cores)
    prec, me.PRECISION = me.PRECISION, 0      # the CPU port computes in fp32 (bf16 emulation would only slow it down)
    with _lib.use_library(_lib.bind(oracle_so)):
        model, cfg = make_model(args.dataset, forced, "cpu")
        model.train()
        opt = torch.optim.AdamW(model.parameters(), lr=cfg.OPTIMIZATION.LR, weight_decay=cfg.OPTIMIZATION.WEIGHT_DECAY)
        batch = build_model.synthetic_batch(cfgname, nsc, device="cpu")
        t_start, warm, times = time.time(), 0, []
        while len(times) < 5:
            t0 = time.time()
            train_step(model, opt, batch, cfg.OPTIMIZATION.GRAD_NORM_CLIP)
            dt = time.time() - t0
            if warm < 2 and time.time() - t_start + 3 * dt < budget:     # warm-up only while the budget leaves room for timed steps
                warm += 1
                continue
            times.append(dt)
            if time.time() - t_start + dt > budget:
                break
    me.PRECISION = prec
    times.sort()
    med = times[len(times) // 2]
    return {"value": nsc / med, "unit": "scenes/s", "cores": cores, "kind": "port",
            "sample": "%d scene(s) of %s per step, fwd+bwd+AdamW of the full detector on the CPU oracle in fp32 (OpenMP + torch CPU "
                      "threads = %d); %d warm-up + median of %d timed step(s), %.1f s per step" % (nsc, cfgname, cores, warm, len(times), med)}


def pmc_traffic(kernel_substr):
    """(HBM bytes per launch of the dominant kernel, where the figure comes from).  NOT measured by this run: PMC counters
    need their own rocprofv3 passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, as the profiling
    guide prescribes), so the figure is read from the newest pair of CSVs committed under profiles/ and `traffic_source`
    names them.  gfx950 correction: FETCH_SIZE counts 128-byte requests as 64 B for 16-B/lane reads -> doubled."""
    import csv
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        names = [os.path.join("profiles", "%s_pmc_%s.csv" % (rnd, c)) for c in ("FETCH_SIZE", "WRITE_SIZE")]
        if not all(os.path.exists(os.path.join(ROOT, n)) for n in names):
            continue
        try:
            tot, variants = [], {}
            for i, n in enumerate(names):
                rows = [r for r in csv.DictReader(open(os.path.join(ROOT, n))) if kernel_substr in r["Kernel_Name"]]
                if not rows:
                    raise KeyError(kernel_substr)
                tot.append(sum(float(r["Counter_Value"]) for r in rows) * 1024.0 / len(rows))
                for r in rows:          # per template instance of the kernel (its launches differ in what they read: see DESIGN.md section 5)
                    v = variants.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), [[0.0, 0], [0.0, 0]])
                    v[i][0] += float(r["Counter_Value"]) * 1024.0
                    v[i][1] += 1
            pmc_traffic.by_variant = {k: {"launches_in_pass": v[0][1], "bytes_per_launch": 2.0 * v[0][0] / max(v[0][1], 1) + v[1][0] / max(v[1][1], 1)}
                                      for k, v in variants.items()}
            return 2.0 * tot[0] + tot[1], ("%s + %s (committed rocprofv3 --pmc passes of `bench.py --steps 2 --warmup 1`, builder-side; "
                                           "2 x FETCH_SIZE + WRITE_SIZE averaged over the kernel's launches)" % tuple(names))
        except Exception:
            continue
    return None, None


def cpu_baseline_subprocess(args, threads=None, sample=None, budget=80):
    """The cpu_baseline leg in a child process (this process runs torch's host ops on one thread), on `threads` or
    min(host cores, 16) threads: measured on the 256-thread GPU host the oracle's step takes 2.9 / 2.1 / 2.8 / 5.3 / 14.1 s
    at 8 / 16 / 32 / 64 / 128 threads (profiles/r03_cpu_thread_scaling.txt, tools/cpu_thread_scaling.py) -- 16 is its best."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(threads or min(os.cpu_count() or 1, 16)), CG3D_CPU_BUDGET_S=str(budget))
    env.pop("CG3D_ENGINE_ANY", None)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-sample", sample or args.cpu_sample,
           "--dataset", args.dataset] + (["--natural"] if args.natural else [])
    every = (lambda: os.sched_setaffinity(0, _ALL_CPUS)) if _ALL_CPUS else None      # the child is not pinned
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, preexec_fn=every)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not lines:
        raise RuntimeError("cpu_baseline child failed: %s" % out.stderr[-400:])
    return json.loads(lines[-1])


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves, one process per GPU, the way
    the reference's `tools/scripts/dist_train.sh:1-18` wraps `tools/train.py` (`python -m torch.distributed.launch
    --nproc_per_node=N train.py --launcher pytorch`, ranks picked up at `tools/train.py:59-74`).  The children see
    WORLD_SIZE and take the normal path below; rank 0's JSON line is this process's output, its exit code ours."""
    import subprocess
    single = os.environ.get("CG3D_SINGLE_DEVICE") == "1"          # test aid: every rank on cuda:0 (gloo)
    if not single:
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible -- refusing to report a %d-GPU number from fewer devices"
                             % (args.gpus, have, args.gpus))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC: RCCL across processes needs it on this stack
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if os.environ.get("CG3D_BENCH_WATCHDOG"):       # dev aid: every thread's stack to stderr after N seconds, then exit (a hung rank says where)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["CG3D_BENCH_WATCHDOG"]), exit=True)
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args, not args.natural)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d -- the line's n_gpus would not be the number asked for "
                         "(launch with --nproc-per-node %d, or drop the launcher and let --gpus start the ranks)"
                         % (args.gpus, world, args.gpus))
    if world > 1 and os.environ.get("CG3D_SINGLE_DEVICE") != "1" and torch.cuda.device_count() < min(world, int(os.environ.get("LOCAL_WORLD_SIZE", world))):
        raise SystemExit("bench.py: %d ranks on this node but only %d GPU(s) visible" % (world, torch.cuda.device_count()))
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
    HEADS = {"split": me.PREC_SPLIT, "fp32": 0, "bf16": None}
    me.HEAD_PRECISION = HEADS[args.head_precision] if args.precision == "bf16" else None

    model, cfg = make_model(args.dataset, forced, dev, build_model.VOXEL_SIZE_OF_CONFIG.get(args.config))
    model.train()
    net = model
    if use_dist and os.environ.get("CG3D_TORCH_DDP") == "1":      # A/B: torch DDP (per-parameter hooks + bucket copies)
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], gradient_as_bucket_view=True)
    elif use_dist:
        from cagroup3d_amd.grad_sync import TwoBucketGradSync
        model.grad_sync = TwoBucketGradSync(model)   # flat buckets over RCCL, sent from inside the backward pass
    from cagroup3d_amd.optim import ClippedAdamW
    if os.environ.get("CG3D_PLAIN_ADAMW") == "1":       # A/B: torch.optim.AdamW(fused) + clip_grad_norm_ called separately
        opt = torch.optim.AdamW(model.parameters(), lr=cfg.OPTIMIZATION.LR, weight_decay=cfg.OPTIMIZATION.WEIGHT_DECAY, fused=True)
    else:
        opt = ClippedAdamW(model.parameters(), lr=cfg.OPTIMIZATION.LR, weight_decay=cfg.OPTIMIZATION.WEIGHT_DECAY)
    if hasattr(model, "split_late_parameters"):
        model.split_late_parameters(opt)            # (only with CG3D_LATE_WEIGHTS=defer / stream: the class branches' AdamW rows and weight copies out of the device-bound half)
    clip = cfg.OPTIMIZATION.GRAD_NORM_CLIP
    # every rank owns different scenes (scene i -> rank i mod W), fixed across steps
    batch = build_model.synthetic_batch(args.config, args.batch, first_scene=rank * args.batch, device=dev)
    # the headline's batches: `batch` and rotate - 1 others (rank r's j-th batch starts at scene (r + W j) * batch: no scene twice)
    rot_batches = [batch] + [build_model.synthetic_batch(args.config, args.batch, first_scene=(rank + world * j) * args.batch, device=dev)
                             for j in range(1, max(args.rotate, 1))]
    nb = len(rot_batches)

    # backward nodes on the issuing thread: the autograd engine otherwise hands every node to its per-device thread, and the ~120
    # nodes of this step are Python functions -- an interpreter-lock hand-over each, inside the host-bound stretch of the step
    # (100 pinned steps, three alternating pairs: median 23.2 / 23.3 / 23.2 ms against 23.7 / 23.4 / 23.4)
    if os.environ.get("CG3D_AUTOGRAD_ST", "1") != "0":
        torch.autograd.set_multithreading_enabled(False)
    n_warm = max(args.warmup, nb if nb > 1 else 0)          # every batch at least once: its cache entries and allocator sizes exist
    for i in range(n_warm):
        tb = train_step(net, opt, rot_batches[i % nb], clip, rot_batches[(i + 1) % nb])
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

    # live HIP-event timing of every conv launch (the roofline figures) -- on two of the timed steps (CG3D_BENCH_PROFILE_STEPS),
    # spread evenly over the timed region: two events per launch add up (a profiled step records ~440 events and takes ~2 ms
    # longer; 20 k live events in a 100-step run slowed the run itself).  2 steps x 67 launches of the dominant kernel is plenty
    # for an average.
    rank_ms = []

    # Lanes (engine.py: the backbone's two chains, DAPPM's branches and the weight gradients on queues of their own): two
    # launches that run at once share the device, and an event pair around one of them then measures how long the PAIR took to
    # let it through -- not the kernel.  The profiled steps therefore run tables WITHOUT lanes (engine.LANES off while their
    # programs are compiled: the dry run of the step before, and the step itself -- emission order on one stream, what a
    # CG3D_LANES=0 run issues and what rocprofv3 sees in one, profiles/): `roofline` is the kernel alone;
    # `roofline.on_lanes` gives the same launches' average with the lanes on their queues (two extra, untimed steps).
    from cagroup3d_amd import engine as _engine

    def timed_run(steps, batches=None, lanes_while_profiling=False):
        batches = batches or [batch]
        me.KernelProfile.reset()
        me.KernelProfile.wgrad = True               # the weight gradient is part of the step's 8(d) work
        stride = max(1, -(-steps // int(os.environ.get("CG3D_BENCH_PROFILE_STEPS", "2"))))
        first = 1 if (stride > 1 and steps > 1) else 0       # (step 0's program was compiled before the run began)
        profiled = 0
        lanes = _engine.LANES
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            me.KernelProfile.enabled = i % stride == first
            profiled += int(me.KernelProfile.enabled)
            _engine.LANES = lanes and (lanes_while_profiling or not (me.KernelProfile.enabled or (i + 1) % stride == first))
            tb_ = train_step(net, opt, batches[i % len(batches)], clip, batches[(i + 1) % len(batches)])
        _engine.LANES = lanes
        pending = _PREPARED.get(id(net.module if hasattr(net, "module") else net))
        if pending is not None:
            pending[0].result()                     # the worker thread's dry run of the next batch belongs to the timed work
        barrier()
        dt_ = time.perf_counter() - t0
        me.KernelProfile.enabled = False
        if use_dist:
            t = torch.tensor([dt_], device=dev, dtype=torch.float64)
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            rank_ms[:] = [float(e.item()) / steps * 1e3 for e in every]
            dt_ = max(float(e.item()) for e in every)            # the job is as slow as its slowest rank
        return dt_, profiled, tb_

    KNAMES = {
        "tile_bf16": ("k_spconv_tile2 (sparse conv fwd + dgrad on LDS-staged neighbour tiles: one 4-wave workgroup per "
                      "(128-row tile, 128/64 output channels), two per CU; the tile's distinct input rows reach LDS by LDS-DMA, "
                      "bf16 MFMA from LDS with fragment-ordered weights streamed from L2, one store per output row, BatchNorm "
                      "statistics in the store phase)", "k_spconv_tile"),
        "implicit_bf16": ("k_spconv_implicit_bf16_ad (sparse conv fwd + dgrad, output-stationary: neighbour rows -> "
                          "registers -> bf16 MFMA -> one store per output row)", "k_spconv_implicit_bf16"),
        "pairs_bf16": ("k_spconv_pairs_bf16 (sparse conv fwd + dgrad: gather -> bf16 MFMA -> atomic scatter)", "k_spconv_pairs_bf16"),
        "pairs": ("k_spconv_pairs_lds (sparse conv fwd + dgrad: gather -> fp32 MFMA -> atomic scatter)", "k_spconv_pairs_lds"),
        "wgrad_bf16": ("k_spconv_pairs_wgrad_rows16 (sparse conv weight gradient, bf16 rows -> bf16 MFMA, fp32 accumulate)",
                       "k_spconv_pairs_wgrad_rows16"),
        "wgrad_bf16_fp32rows": ("k_spconv_pairs_wgrad_bf16 (weight gradient, fp32 rows rounded to bf16 operands)", "k_spconv_pairs_wgrad_bf16"),
        "wgrad": ("k_spconv_pairs_wgrad / _t128 (weight gradient, fp32 MFMA)", "k_spconv_pairs_wgrad"),
    }
    # "...x3" kinds: the same kernels on split operands (the heads, me.PREC_SPLIT): a launch multiplies a three times longer
    # contraction -- three bf16 products per fp32-accurate product.  Priced at 1 x their 2 P Cin Cout (flops_of);
    # `roofline.split_launches` states their share.

    def base_kind(k):
        return k[:-2] if k.endswith("x3") else k

    def flops_of(k, f):
        # 2 P Cin Cout per launch, whatever arithmetic delivers it.  A split-operand launch issues three bf16 products
        # per fp32-accurate product; rounds 4-5 priced those launches at 3 x (CG3D_ROOFLINE_SPLIT_3X=1 restores that): the extra
        # two thirds are the price of fp32 accuracy on bf16 pipes, not algorithmic work
        return 3.0 * f if (k.endswith("x3") and os.environ.get("CG3D_ROOFLINE_SPLIT_3X") == "1") else f

    def roofline_of(dt_, profiled, steps, precision):
        """per launch flops = 2 P Cin Cout, bytes = every tensor once; bound = max(flops / MFMA peak, bytes / HBM peak)."""
        kinds = me.KernelProfile.summary()
        # by KERNEL (the plain and the split-operand launches of one kernel are one line of a rocprofv3 kernel-stats table)
        merged = {}
        for k, v in kinds.items():
            d = merged.setdefault(base_kind(k), {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "bytes_per_pair": 0.0,
                                                 "split_launches": 0, "split_ms": 0.0})
            for f in ("launches", "ms", "bytes", "bytes_per_pair"):
                d[f] += v[f]
            d["flops"] += flops_of(k, v["flops"])
            if k.endswith("x3"):
                d["split_launches"] += v["launches"]
                d["split_ms"] += v["ms"]
        fwd = {k: v for k, v in merged.items() if not k.startswith("wgrad")}
        kind = max(fwd, key=lambda k: fwd[k]["ms"])                      # the dominant forward / data-gradient kernel of this run
        prof = merged[kind]
        secs = prof["ms"] * 1e-3
        bf16 = precision == 1

        def peak_of(k):      # operand type of the kernel: fp32 MFMA only for the fp32-operand kernels
            return (FP32_MFMA_PEAK_TFLOPS if k in ("pairs", "wgrad") else BF16_MFMA_PEAK_TFLOPS) * 1e12
        bw = HBM_PEAK_GBS * 1e9
        t_f, t_b = prof["flops"] / peak_of(kind), prof["bytes"] / bw
        tf = prof["flops"] / secs / 1e12 if secs > 0 else 0.0
        gbs = prof["bytes"] / secs / 1e9 if secs > 0 else 0.0
        if t_f >= t_b:
            roof = {"kernel": KNAMES[kind][0], "bound": "mfma", "achieved": tf, "peak": peak_of(kind) / 1e12, "unit": "TFLOP/s",
                    "frac": tf / (peak_of(kind) / 1e12), "traffic": None, "algorithmic_gbytes_per_s": gbs}
        else:
            roof = {"kernel": KNAMES[kind][0], "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBS, "traffic": None, "achieved_tflops": tf}
        per_kind = {}
        for r in me.KernelProfile.records:
            for key in {r[4][0], base_kind(r[4][0])}:
                d = per_kind.setdefault(key, [0.0, 0.0])
                d[0] += max(flops_of(r[4][0], r[2]) / peak_of(r[4][0]), r[3] / bw)
                d[1] += r[0].elapsed_time(r[1]) * 1e-3
        roof.update(launches=prof["launches"], avg_launch_ms=prof["ms"] / max(prof["launches"], 1),
                    kernel_time_share=secs / (dt_ * profiled / steps), timed_steps=profiled,
                    algorithmic_bytes_per_launch=prof["bytes"] / max(prof["launches"], 1),
                    algorithmic_flops_per_launch=prof["flops"] / max(prof["launches"], 1),
                    # sum of the launches' own bounds / sum of their measured times (a kernel mixing MFMA-bound and
                    # HBM-bound layers is priced layer by layer)
                    frac_8d_per_layer=per_kind[kind][0] / per_kind[kind][1] if per_kind[kind][1] > 0 else None,
                    # round 1's figure: a gathered row counted once per pair it takes part in (NOT the 8(d) numerator)
                    per_pair_gbytes_per_s=prof["bytes_per_pair"] / secs / 1e9 if secs > 0 else None,
                    conv_kernels={k: {"launches_per_step": v["launches"] / profiled, "ms_per_step": v["ms"] / profiled,
                                      "bound_over_measured": per_kind[k][0] / per_kind[k][1] if per_kind[k][1] > 0 else None,
                                      "tflops": flops_of(k, v["flops"]) / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else None,
                                      "gbytes_per_s_8d": v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else None}
                                  for k, v in kinds.items()})
        roof["traffic"], roof["traffic_source"] = pmc_traffic(KNAMES[kind][1]) if bf16 else (None, None)
        roof["traffic_by_variant"] = getattr(pmc_traffic, "by_variant", None) if bf16 else None
        roof["split_launches"] = {"launches": prof["split_launches"], "ms_per_step": prof["split_ms"] / profiled,
                                  "note": "launches of this kernel on split operands (the two heads): 3 bf16 products per fp32-accurate "
                                          "product, counted at 1 x 2 P Cin Cout (SURVEY 8(d)) since round 6 (rounds 4-5: 3 x)"}
        bound_all = sum(v[0] for k, v in per_kind.items() if k in kinds)
        meas_all = sum(v[1] for k, v in per_kind.items() if k in kinds)
        # every sparse convolution of the step (forward, data and weight gradient): their 8(d) bounds over their measured
        # time, and over the whole step (which also holds BN, the heads, losses, target assignment, AdamW, map building)
        roof["conv_bound_over_conv_time"] = bound_all / meas_all if meas_all > 0 else None
        roof["conv_bound_over_step_time"] = (bound_all / profiled) / (dt_ / steps)
        return roof

    dt, profiled_steps, tb = timed_run(args.steps, rot_batches)
    per_rank_ms = list(rank_ms)
    roof = roofline_of(dt, profiled_steps, args.steps, me.PRECISION) if rank == 0 else None
    on_lanes = bool(_engine.LANES and _engine.LANES_RUN and os.environ.get("CG3D_BENCH_ON_LANES", "1") != "0")
    if use_dist:
        # the lane tuner decides per process (two ranks on one device share its queues and may decide differently): the extra steps
        # hold collectives, so the ranks take them together or not at all (a rank that skipped them once left the other in a barrier)
        flag = torch.tensor([int(on_lanes)], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        on_lanes = bool(int(flag.item()))
    if on_lanes:
        keep_ms = list(rank_ms)
        dt_l, prof_l, _ = timed_run(2, lanes_while_profiling=True)       # (every rank: the steps hold collectives)
        rank_ms[:] = keep_ms
        if rank == 0:
            r_l = roofline_of(dt_l, prof_l, 2, me.PRECISION)
            roof["queues"] = ("profiled steps: launch programs compiled without lanes, every launch alone on the one stream; all other "
                              "steps: lanes on their own queues")
            roof["on_lanes"] = {"kernel": r_l["kernel"], "avg_launch_ms": r_l["avg_launch_ms"], "frac": r_l["frac"], "launches": r_l["launches"],
                                "note": "the same launches timed while the other lanes' kernels run beside them (2 untimed steps): the "
                                        "time a launch takes to get through a shared device, not a property of the kernel"}

    fp32 = None
    if me.PRECISION == 1 and not args.no_fp32 and os.environ.get("CG3D_BENCH_FP32", "1") != "0":
        # the parity configuration (fp32 operands everywhere) in the same process, same model and batch
        me.PRECISION = 0
        n32 = max(3, min(args.steps, 6))
        for _ in range(2):
            train_step(net, opt, batch, clip)
        dt32, prof32, _ = timed_run(n32)
        if rank == 0:
            r32 = roofline_of(dt32, prof32, n32, 0)
            fp32 = {"value": world * args.batch * n32 / dt32, "unit": "scenes/s", "ms_per_step": dt32 / n32 * 1e3, "steps": n32,
                    "warmup": 2, "dominant_kernel": r32["kernel"], "bound": r32["bound"], "achieved": r32["achieved"],
                    "roofline_unit": r32["unit"], "frac": r32["frac"], "frac_8d_per_layer": r32["frac_8d_per_layer"],
                    "avg_launch_ms": r32["avg_launch_ms"], "conv_bound_over_step_time": r32["conv_bound_over_step_time"]}
        me.PRECISION = 1
    other_heads = {}
    if me.PRECISION == 1 and not args.no_fp32 and os.environ.get("CG3D_BENCH_FP32", "1") != "0":
        # the same bf16 backbone with the heads in the two OTHER arithmetics, same process, model and batch, as many steps as
        # the headline (at most 20)
        keep = me.HEAD_PRECISION
        for name, hp in HEADS.items():
            if hp == keep:
                continue
            me.HEAD_PRECISION = hp
            n2 = max(3, min(args.steps, 20))
            for _ in range(3):
                train_step(net, opt, batch, clip)
            dt2, _, _ = timed_run(n2)
            if rank == 0:
                other_heads[name] = {"value": world * args.batch * n2 / dt2, "unit": "scenes/s", "ms_per_step": dt2 / n2 * 1e3, "steps": n2,
                                     "warmup": 3, "heads": name}
        me.HEAD_PRECISION = keep
    fp32_rows = None
    if me.PRECISION == 1 and _engine.ACT_BF16 and not args.no_fp32 and os.environ.get("CG3D_BENCH_FP32", "1") != "0":
        # the headline's arithmetic with the backbone's activations / activation gradients STORED as fp32 rows (CG3D_ACT_BF16=0:
        # BASELINE.json configs[1] read as "bf16 MFMA operands only"); the programs are compiled per step, so the switch applies
        # from the next step on
        _engine.ACT_BF16 = False
        n2 = max(3, min(args.steps, 20))
        try:
            for _ in range(3):
                train_step(net, opt, batch, clip)
            dt2, _, _ = timed_run(n2)
        finally:
            _engine.ACT_BF16 = True
        for _ in range(2):
            train_step(net, opt, batch, clip)
        if rank == 0:
            fp32_rows = {"value": world * args.batch * n2 / dt2, "unit": "scenes/s", "ms_per_step": dt2 / n2 * 1e3, "steps": n2, "warmup": 3,
                         "note": "CG3D_ACT_BF16=0: bf16 MFMA operands in the backbone, every feature matrix and gradient stored as fp32 rows"}
    fixed = None
    if nb > 1:
        # rounds 1-5's headline: ONE fixed batch repeated (the lightest of the nine S50k x 4 batches at tensor stride 4: 82 107
        # voxels against a mean of 89 081) -- sizes, cache keys and allocator requests never change
        nfix = max(3, min(args.steps, 20))
        for _ in range(2):
            train_step(net, opt, batch, clip)
        dtf, _, _ = timed_run(nfix)
        if rank == 0:
            fixed = {"value": world * args.batch * nfix / dtf, "unit": "scenes/s", "ms_per_step": dtf / nfix * 1e3, "steps": nfix, "warmup": 2,
                     "note": "the first of the headline's %d batches repeated (the headline of rounds 1-5)" % nb}
    finish_prefetch(net)            # the worker thread is done and joined before anything else happens (cpu_baseline, exit)

    if rank == 0:
        out = {"metric": "scenes/s fwd+bwd ScanNet ~50k pts", "value": world * args.batch * args.steps / dt,
               "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": n_warm,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16" if me.PRECISION == 1 else "f32", "data": "synthetic",
               "config": {"workload": "%s CAGroup3D.yaml, %d x %s scenes per GPU, %s" % (
                   {"scannet": "ScanNetV2", "sunrgbd": "SUN RGB-D"}.get(args.dataset, args.dataset), args.batch, args.config, "forced GT selection (replaces the net's own selection: sizes independent of the weights) + own-class logit boost (trained-like loads)" if forced
                   else "natural selection of the untrained net"),
                          "batches": nb, "batches_note": ("%d different synthetic batches of %d x %s scenes cycled inside the timed region, the next "
                                                          "batch's coordinate dry run on the worker thread as in training" % (nb, args.batch, args.config)
                                                          if nb > 1 else "one fixed batch repeated"),
                          "scenes_per_gpu": args.batch, "points_per_scene": int("".join(ch for ch in args.config.split("-")[0] if ch.isdigit())) * 1000,
                          "voxel_size_m": float(model.voxel_size), "parallelism": "dp%d" % world, "optimizer": "AdamW+clip10",
                          "precision": (("bf16 MFMA operands (fp32 accumulate) in the convolutions (>= 16 input channels) of the BACKBONE "
                                         "only -- forward, data gradient and weight gradient; both heads (class branches, RoI pooling, "
                                         "1x1x1 layers) compute fp32-accurate products from three bf16 MFMA passes on split operands "
                                         "(x = hi + lo: xhi whi + xlo whi + xhi wlo, ~1e-5 relative; tests/test_split_precision.py pins "
                                         "them against the fp32 oracle at rtol 1e-4): BASELINE.json configs[1] (\"bf16 backbone\")"
                                         if me.HEAD_PRECISION == me.PREC_SPLIT else
                                         "bf16 MFMA operands (fp32 accumulate) in the convolutions (>= 16 input channels) of the BACKBONE "
                                         "only -- forward, data gradient and weight gradient; both heads (class branches, RoI pooling, "
                                         "1x1x1 layers) run fp32 MFMA operands: BASELINE.json configs[1] read literally"
                                         if me.HEAD_PRECISION == 0 else
                                         "bf16 MFMA operands (fp32 accumulate) in every sparse convolution with >= 16 input channels -- "
                                         "backbone, class branches, RoI pooling and the 1x1x1 layers; forward, data gradient AND weight "
                                         "gradient (k_spconv_pairs_wgrad_rows16 on the bf16 row copies)") +
                                        ("; STORAGE inside the backbone's launch program: activations and activation gradients are "
                                         "bf16 rows only (feature matrices of >= %d rows; convolution sums rounded on the store, BatchNorm "
                                         "statistics summed from the unrounded fp32 sums, normalisation on the rounded rows), fp32 for "
                                         "smaller matrices, the 3-channel input, the 64-channel output, pooling / interpolation; weights, "
                                         "parameter gradients, statistics, both heads' rows, losses and the optimizer state are fp32 "
                                         "(CG3D_ACT_BF16=0 = fp32 rows everywhere: sub-record `fp32_rows` of this line)"
                                         % _engine.ACT16_MIN_ROWS if _engine.ACT_BF16 else
                                         "; activations, weights, gradients, BatchNorm, losses and the optimizer stay fp32 (CG3D_ACT_BF16=0)")
                                        if me.PRECISION == 1 else "fp32 everywhere (parity configuration)"),
                          "act_bf16": bool(_engine.ACT_BF16) and me.PRECISION == 1, "act16_min_rows": int(_engine.ACT16_MIN_ROWS),
                          "last_loss": tb.get("loss_all"),
                          "backbone_issue": dict(__import__("cagroup3d_amd.engine", fromlist=["STATS"]).STATS),
                          # launch programs on several queues (engine.py, lanes): compiled with lanes / run on them, and what the
                          # first passes' timing said (forward table in ms: on its lanes, on one stream)
                          "lanes": {"compiled": bool(_engine.LANES), "on_queues": bool(_engine.LANES and _engine.LANES_RUN),
                                    "weight_gradient_lane": _engine.WGRAD_LANE, "dappm_lanes": list(_engine.DAPPM_LANES),
                                    "autotune_forward_ms": list(_engine._LaneTuner.verdict) if _engine._LaneTuner.verdict else None}},
               "roofline": roof}
        if fp32 is not None:
            out["fp32"] = fp32
        if fixed is not None:
            out["fixed_batch"] = fixed
        if fp32_rows is not None:
            out["fp32_rows"] = fp32_rows
        for name, rec in other_heads.items():
            out[{"fp32": "bf16_backbone_fp32_heads", "bf16": "bf16_all_convolutions", "split": "bf16_backbone_split_heads"}[name]] = rec
        if use_dist and getattr(model, "grad_sync", None) is not None and hasattr(model.grad_sync, "report"):
            out["comm"] = model.grad_sync.report()
        if use_dist:
            out["per_rank_ms_per_step"] = per_rank_ms
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline_subprocess(args)
                # One thread, on a 10 x SMALLER scene (a 50 k-point step takes minutes on one core): its own key and its own
                # unit -- an S5k scene per second is not the unit of `value` (S50k scenes) and must not be read next to it.
                # Thread scaling of the S50k step itself: profiles/r03_cpu_thread_scaling.txt (tools/cpu_thread_scaling.py).
                try:
                    st = cpu_baseline_subprocess(args, threads=1, sample=args.cpu_sample_1t, budget=40)
                    st["unit"] = "%s-scenes/s (NOT the unit of cpu_baseline.value)" % args.cpu_sample_1t.split(":")[0]
                    out["cpu_baseline"]["single_thread_small_scene"] = st
                except Exception as e:
                    out["cpu_baseline"]["single_thread_small_scene"] = {"value": None, "sample": "failed: %r" % (e,)}
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "scenes/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
