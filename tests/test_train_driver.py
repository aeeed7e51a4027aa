"""Train / eval drivers (cagroup3d_amd/train.py) on the CPU oracle: schedule, checkpoint round trip, resume, evaluation."""
import os

import numpy as np
import pytest
import torch

from cagroup3d_amd import _lib, build_model, me, train
from cagroup3d_amd.pcdet.config import cfg_from_yaml_file


def test_step_decay_schedule_per_iteration():
    cfg = build_model.load_cfg("scannet").OPTIMIZATION
    w = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.AdamW([w], lr=cfg.LR)
    sched = train.build_scheduler(opt, iters_per_epoch=5, optim_cfg=cfg)
    lrs = []
    for _ in range(50):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
    # DECAY_STEP_LIST [7, 9] epochs x 5 iterations, LR_DECAY 0.1
    assert lrs[0] == pytest.approx(cfg.LR) and lrs[34] == pytest.approx(cfg.LR)
    assert lrs[35] == pytest.approx(cfg.LR * 0.1) and lrs[44] == pytest.approx(cfg.LR * 0.1)
    assert lrs[45] == pytest.approx(cfg.LR * 0.01)


def test_dataset_shards_like_distributed_sampler():
    a = train.SyntheticIndoorDataset("S5k", 10, 2, rank=0, world=2)
    b = train.SyntheticIndoorDataset("S5k", 10, 2, rank=1, world=2)
    assert a.scene_ids == [0, 2, 4, 6, 8] and b.scene_ids == [1, 3, 5, 7, 9] and len(a) == 3
    batch = next(a.batches(epoch=0))
    assert batch["points"].shape[1] == 7 and batch["gt_boxes"].shape[0] == 2 and len(batch["instance_mask"]) == 2
    annos = train.SyntheticIndoorDataset.gt_annos(batch)
    assert annos[0]["gt_num"] == len(annos[0]["class"]) > 0 and annos[0]["gt_boxes_upright_depth"].shape[1] == 7


def test_uneven_shards_are_padded_to_equal_shares():
    """n_scenes % world != 0: every rank must see the same number of scenes, batches and last-batch size (ADVICE r1:
    mismatched collectives at the end of the epoch otherwise) -- the deal torch's DistributedSampler makes."""
    from torch.utils.data.distributed import DistributedSampler
    for n, world, bs in ((10, 4, 2), (7, 2, 4), (13, 8, 4), (3, 4, 2)):
        shares = [train.shard_ids(range(n), r, world) for r in range(world)]
        assert len({len(s) for s in shares}) == 1 and set(sum(shares, [])) == set(range(n))
        for r in range(world):
            ref = list(DistributedSampler(list(range(n)), num_replicas=world, rank=r, shuffle=False))
            assert shares[r] == ref, (n, world, r)
        ds = [train.SyntheticIndoorDataset("S5k", n, bs, rank=r, world=world) for r in range(world)]
        assert len({len(d) for d in ds}) == 1
        assert len({tuple(len(d.scene_ids[i:i + bs]) for i in range(0, len(d.scene_ids), bs)) for d in ds}) == 1


def test_train_checkpoint_resume_eval(oracle, tmp_path):
    ck = str(tmp_path / "ck.pth")
    with _lib.use_library(oracle):
        np.random.seed(0)
        torch.manual_seed(0)
        train.main(["--config", "S5k", "--scenes", "2", "--batch", "2", "--epochs", "1", "--ckpt", ck, "--device", "cpu"])
        state = torch.load(ck, map_location="cpu", weights_only=False)
        assert state["epoch"] == 1 and state["it"] == 1 and "backbone_3d.conv1.0.kernel" in state["model_state"]
        assert state["optimizer_state"]["state"], "AdamW moments are part of the checkpoint"
        # resume: parameters and iteration counter come back, one more epoch runs, evaluation returns the metric dict
        model, cfg = build_model.build_cagroup3d("scannet", seed=1)
        it, ep = model.load_params_with_optimizer(ck, to_cpu=True)
        assert (it, ep) == (1, 1)
        for k, v in model.state_dict().items():
            assert torch.equal(v, state["model_state"][k]), k
        res = train.main(["--config", "S5k", "--scenes", "2", "--batch", "2", "--epochs", "2", "--resume", ck, "--device", "cpu",
                          "--eval"])
    assert set(k for k in res if k.startswith("m")) == {"mAP_0.25", "mAP_0.50", "mAR_0.25", "mAR_0.50"}


@pytest.mark.gpu
def test_train_and_eval_on_device(hip, tmp_path):
    ck = str(tmp_path / "ck.pth")
    with _lib.use_library(hip):
        res = train.main(["--config", "S5k", "--scenes", "4", "--batch", "2", "--epochs", "1", "--ckpt", ck, "--eval",
                          "--precision", "bf16"])
    from cagroup3d_amd import me
    me.PRECISION = 0
    assert os.path.exists(ck) and "mAP_0.25" in res


def test_clipped_adamw_is_clip_grad_norm_plus_torch_adamw():
    """cagroup3d_amd.optim.ClippedAdamW.clip_and_step == torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW(fused=True).step():
    bit-identical parameters over steps with and without active clipping, same reported norm, state_dict round trip."""
    import copy
    from cagroup3d_amd.optim import ClippedAdamW
    torch.manual_seed(0)
    m1 = torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.ReLU(), torch.nn.Linear(13, 3))
    m2 = copy.deepcopy(m1)
    o1 = torch.optim.AdamW(m1.parameters(), lr=1e-2, weight_decay=0.01, fused=True)
    o2 = ClippedAdamW(m2.parameters(), lr=1e-2, weight_decay=0.01)
    for it in range(6):
        x = torch.randn(16, 7) * (10 if it % 2 else 0.1)            # gradients above / below the clipping bar
        for m, o in ((m1, o1), (m2, o2)):
            o.zero_grad(set_to_none=True)
            (m(x) ** 2).sum().backward()
        n1 = torch.nn.utils.clip_grad_norm_(list(m1.parameters()), 1.0)
        o1.step()
        n2 = o2.clip_and_step(1.0)
        assert torch.allclose(n1, n2)
        for a, b in zip(m1.parameters(), m2.parameters()):
            assert torch.equal(a, b), it
        if it == 2:                                                 # a reloaded state continues identically
            o2.load_state_dict(copy.deepcopy(o2.state_dict()))
    for g in o2.param_groups:
        g["lr"] = 5e-3                                              # a scheduler's change is picked up
    o1.param_groups[0]["lr"] = 5e-3
    for m, o in ((m1, o1), (m2, o2)):
        o.zero_grad(set_to_none=True)
        (m(torch.ones(4, 7)) ** 2).sum().backward()
    torch.nn.utils.clip_grad_norm_(list(m1.parameters()), 1.0)
    o1.step()
    o2.clip_and_step(1.0)
    for a, b in zip(m1.parameters(), m2.parameters()):
        assert torch.equal(a, b)


def _adamw_case(dev, steps=4):
    """models with parameter sizes around the chunk / vector boundaries; returns (torch-stepped, library-stepped) params"""
    import copy
    from cagroup3d_amd.optim import ClippedAdamW
    torch.manual_seed(1)
    shapes = [(3,), (70001,), (129, 257), (27, 64, 64), (5,), (32768,), (32769,)]
    p1 = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
    p2 = [torch.nn.Parameter(p.detach().clone()) for p in p1]
    o1 = torch.optim.AdamW(p1, lr=1e-2, weight_decay=0.02, fused=True)
    o2 = ClippedAdamW(p2, lr=1e-2, weight_decay=0.02)
    for it in range(steps):
        gs = [torch.randn_like(p) * (30.0 if it % 2 else 0.01) for p in p1]
        for ps in (p1, p2):
            for p, g in zip(ps, gs):
                p.grad = g.clone()
        n1 = torch.nn.utils.clip_grad_norm_(p1, 1.0)
        o1.step()
        if it == 2:                                         # a plain step() in between: the step counters stay in sync
            n2 = torch.nn.utils.clip_grad_norm_(p2, 1.0)
            o2.step()
        else:
            n2 = o2.clip_and_step(1.0)
        assert torch.allclose(n1, n2)
    return p1, p2, o1, o2


def test_oracle_adamw_step_matches_torch_fused_adamw(oracle):
    """cg3d_adamw_step (the oracle's plain-C statement) against torch's fused AdamW on host tensors, clip scalar included."""
    from ctypes import c_float, c_int64
    import numpy as np
    torch.manual_seed(0)
    n = 10007
    p = torch.randn(n); g = torch.randn(n); m = torch.randn(n) * 0.1; v = torch.rand(n) * 0.1
    pt = torch.nn.Parameter(p.clone()); pt.grad = g.clone() * 0.37
    opt = torch.optim.AdamW([pt], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, fused=True)
    opt.state[pt] = {"step": torch.tensor(6.0), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
    opt.step()
    rows = torch.tensor([[p.data_ptr(), m.data_ptr(), v.data_ptr(), 0, 4000], [p.data_ptr(), m.data_ptr(), v.data_ptr(), 4000, n - 4000]], dtype=torch.int64)
    pid = torch.zeros(2, dtype=torch.int32)
    gp = torch.tensor([g.data_ptr()], dtype=torch.int64)
    clip = torch.tensor([0.37])
    oracle.call("cg3d_adamw_step", rows.data_ptr(), pid.data_ptr(), c_int64(2), gp.data_ptr(), clip.data_ptr(), c_float(3e-3), c_float(0.9),
                c_float(0.999), c_float(1e-8), c_float(0.05), c_float(1 - 0.9 ** 7), c_float(1 - 0.999 ** 7), None)
    torch.testing.assert_close(p, pt.detach(), rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(m, opt.state[pt]["exp_avg"], rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(v, opt.state[pt]["exp_avg_sq"], rtol=2e-6, atol=1e-7)


def _norm_case(lib, dev, max_norm, scale):
    from ctypes import c_float, c_int64
    g = torch.Generator().manual_seed(3)
    grads = [(torch.randn(n, generator=g) * scale).to(dev) for n in (70001, 5, 32768, 33)]
    rows, pid = [], []
    for k, t in enumerate(grads):
        for o in range(0, t.numel(), 32768):
            rows.append((0, 0, 0, o, min(32768, t.numel() - o)))          # (the parameter / moment columns are not read)
            pid.append(k)
    rows_t = torch.tensor(rows, dtype=torch.int64).to(dev)
    pid_t = torch.tensor(pid, dtype=torch.int32).to(dev)
    gp = torch.tensor([t.data_ptr() for t in grads], dtype=torch.int64).to(dev)
    sc = torch.empty(1, dtype=torch.float64, device=dev)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    with _lib.use_library(lib):
        lib.call("cg3d_grad_norm_clip", rows_t.data_ptr(), pid_t.data_ptr(), c_int64(len(rows)), gp.data_ptr(), c_float(max_norm),
                 sc.data_ptr(), out[0:1].data_ptr(), out[1:2].data_ptr(), lib.stream())
    params = [torch.nn.Parameter(torch.zeros_like(t)) for t in grads]
    for p, t in zip(params, grads):
        p.grad = t.clone()
    total = torch.nn.utils.clip_grad_norm_(params, max_norm)
    return out.cpu(), float(total), float(params[0].grad[0] / grads[0][0])


@pytest.mark.parametrize("max_norm,scale", [(10.0, 1.0), (10.0, 1e-3), (0.5, 30.0)])
def test_oracle_grad_norm_clip_is_clip_grad_norm(oracle, max_norm, scale):
    out, total, coef = _norm_case(oracle, "cpu", max_norm, scale)
    assert abs(float(out[0]) - total) <= 2e-6 * total and abs(float(out[1]) - coef) <= 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("max_norm,scale", [(10.0, 1.0), (10.0, 1e-3), (0.5, 30.0)])
def test_hip_grad_norm_clip_is_clip_grad_norm(hip, max_norm, scale):
    out, total, coef = _norm_case(hip, "cuda", max_norm, scale)
    assert abs(float(out[0]) - total) <= 2e-6 * total and abs(float(out[1]) - coef) <= 2e-6


def _nonfinite_norm_case(lib, dev, bad):
    """A NaN / Inf gradient: torch's clip coefficient clamp(max_norm / (norm + 1e-6), max=1) is NaN for a NaN norm and 0 for
    an infinite one -- the fused form must hand the optimiser the same coefficient (no half-poisoned parameter set)."""
    from ctypes import c_float, c_int64
    g = torch.randn(4096)
    g[17] = bad
    g = g.to(dev)
    rows_t = torch.tensor([(0, 0, 0, 0, 4096)], dtype=torch.int64).to(dev)
    pid_t = torch.zeros(1, dtype=torch.int32).to(dev)
    gp = torch.tensor([g.data_ptr()], dtype=torch.int64).to(dev)
    sc = torch.empty(1, dtype=torch.float64, device=dev)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    with _lib.use_library(lib):
        lib.call("cg3d_grad_norm_clip", rows_t.data_ptr(), pid_t.data_ptr(), c_int64(1), gp.data_ptr(), c_float(10.0),
                 sc.data_ptr(), out[0:1].data_ptr(), out[1:2].data_ptr(), lib.stream())
    p = torch.nn.Parameter(torch.zeros(4096))
    p.grad = g.cpu().clone()
    total = torch.nn.utils.clip_grad_norm_([p], 10.0)
    ref_coef = torch.clamp(10.0 / (total + 1e-6), max=1.0)
    out = out.cpu()
    assert torch.isnan(out[0]) == torch.isnan(total) and torch.isinf(out[0]) == torch.isinf(total)
    assert torch.isnan(out[1]) == torch.isnan(ref_coef) and (torch.isnan(ref_coef) or float(out[1]) == float(ref_coef))


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_oracle_grad_norm_clip_propagates_nonfinite(oracle, bad):
    _nonfinite_norm_case(oracle, "cpu", bad)


@pytest.mark.gpu
@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_hip_grad_norm_clip_propagates_nonfinite(hip, bad):
    _nonfinite_norm_case(hip, "cuda", bad)


@pytest.mark.gpu
def test_fused_clip_adamw_launch_matches_torch_on_device(hip):
    """ClippedAdamW on the device takes the library's one-launch step (after torch has created the state in step 1):
    parameters and moments follow torch.optim.AdamW(fused) + clip_grad_norm_ to fp32 rounding; the step counters agree."""
    from cagroup3d_amd import optim
    with _lib.use_library(hip):
        p1, p2, o1, o2 = _adamw_case("cuda", steps=5)
        assert o2._plan not in (None, False) and o2._host_step == 5, "the library step must have run"
    for a, b in zip(p1, p2):
        torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(o2.state[b]["exp_avg_sq"], o1.state[a]["exp_avg_sq"], rtol=1e-5, atol=1e-8)
        assert float(o2.state[b]["step"]) == float(o1.state[a]["step"]) == 5.0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["defer", "stream"])
def test_late_rows_of_the_fused_step_equal_the_one_launch_step(hip, mode):
    """ClippedAdamW.set_early: the early parameters' rows now, the others deferred to me.run_late() (or on the late stream) --
    the SAME kernel over two row ranges, so parameters and moments are bit-identical to the one-launch step; the gradients the
    late rows read survive zero_grad(set_to_none=True) and a burst of allocations on the current stream; in the defer mode the
    late parameters do not move until run_late()."""
    from cagroup3d_amd import me
    from cagroup3d_amd.optim import ClippedAdamW
    keep = (me.LATE_MODE, me.LATE_WEIGHTS)
    me.LATE_MODE, me.LATE_WEIGHTS = mode, True
    try:
        with _lib.use_library(hip):
            torch.manual_seed(3)
            shapes = [(3,), (70001,), (129, 257), (27, 64, 64), (5,), (32768,), (2_000_000,)]
            pa = [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]
            pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
            oa, ob = ClippedAdamW(pa, lr=1e-2, weight_decay=0.02), ClippedAdamW(pb, lr=1e-2, weight_decay=0.02)
            early = [pb[0], pb[2], pb[5]]                       # not a prefix of the parameter list: the late rows trail the TABLE
            ob.set_early(early)
            for it in range(5):
                gs = [torch.randn_like(p) * (30.0 if it % 2 else 0.01) for p in pa]
                for ps in (pa, pb):
                    for p, g in zip(ps, gs):
                        p.grad = g.clone()
                before = [p.detach().clone() for p in pb]
                na = oa.clip_and_step(1.0)
                nb = ob.clip_and_step(1.0)
                oa.zero_grad(set_to_none=True)
                ob.zero_grad(set_to_none=True)
                junk = [torch.full((2_000_000,), float("nan"), device="cuda") for _ in range(4)]      # would land in freed gradients
                del junk
                assert torch.equal(na, nb)
                if it >= 1:
                    assert ob._plan[5] > 0 and ob._late_hold is not None, "the split step must have run"
                    if mode == "defer":
                        torch.cuda.synchronize()
                        for p, q in zip(pb, before):
                            assert torch.equal(p, q) == (not any(p is e for e in early)), "late parameters wait for run_late()"
                        assert len(me._DEFERRED) == 1
                ob.finish_late()
                assert ob._late_hold is None and not me._DEFERRED and not me._LATE_PENDING
            torch.cuda.synchronize()
            for a, b in zip(pa, pb):
                assert torch.equal(a, b)
                assert torch.equal(oa.state[a]["exp_avg"], ob.state[b]["exp_avg"]) and torch.equal(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"])
            # state_dict() flushes by itself
            for p in pb:
                p.grad = torch.randn_like(p)
            ob.clip_and_step(1.0)
            assert ob._late_hold is not None
            ob.state_dict()
            assert ob._late_hold is None and not me._DEFERRED
    finally:
        me._DEFERRED.clear()
        me._LATE_PENDING.clear()
        me.LATE_MODE, me.LATE_WEIGHTS = keep


def _worker_eval(rank, world, port, out):
    """Two gloo ranks evaluate their shards of 5 scenes (uneven: rank 0 gets scenes 0,2,4, rank 1 gets 1,3 and the padding
    duplicate 0); the merged result on rank 0 must be exactly the single-process evaluation, and the non-zero rank gets
    None after taking part in the gather."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import _build_oracle
    oracle = _lib.bind(_build_oracle())
    with _lib.use_library(oracle):
        model, cfg = build_model.build_cagroup3d("scannet", seed=0)
        names = cfg.CLASS_NAMES
        full = train.eval_one_epoch(model, train.SyntheticIndoorDataset("S5k", 5, 2), names, "cpu", log=lambda *a: None) if rank == 0 else None
        ds = train.SyntheticIndoorDataset("S5k", 5, 2, rank, world)
        assert ds.eval_positions() == ([0, 2, 4] if rank == 0 else [1, 3, 0]) and ds.n_total() == 5
        res = train.eval_one_epoch(model, ds, names, "cpu", log=lambda *a: None, rank=rank, world=world)
    if rank == 0:
        assert res is not None and set(res) == set(full)
        for k in full:
            a, b = full[k], res[k]
            assert (a == b) or (a != a and b != b) or abs(a - b) < 1e-9, (k, a, b)
        open(out, "w").write("ok")
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_evaluation_merges_to_the_single_process_result(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok")
    mp.spawn(_worker_eval, args=(2, 29541, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_merge_eval_shards_places_scenes_by_index():
    # single process: identity; the multi-rank path is the spawn test above
    det, gt = train.merge_eval_shards(["a", "b", "c"], ["x", "y", "z"], [0, 1, 2], 3)
    assert det == ["a", "b", "c"] and gt == ["x", "y", "z"]
    with pytest.raises(AssertionError):
        train.merge_eval_shards(["a"], ["x"], [0], 2)             # a scene nobody evaluated


def _worker_sync_bn(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from conftest import _build_oracle
    oracle = _lib.bind(_build_oracle())
    calls = []
    orig = me._all_ranks_sums
    me._all_ranks_sums = lambda *a: (calls.append(1), orig(*a))[1]
    with _lib.use_library(oracle):
        train.main(["--config", "S5k", "--scenes", "4", "--batch", "2", "--epochs", "1", "--device", "cpu", "--sync_bn"])
    assert len(calls) > 100, len(calls)          # every BatchNorm of the step, forward and backward
    if rank == 0:
        open(out, "w").write("ok")


def test_two_rank_training_with_sync_bn(tmp_path):
    """train.main --sync_bn on two gloo ranks (oracle kernels): every BatchNorm -- backbone, class branches, RoI pooling, the
    RoI head's FC layers -- exchanges its statistics tables; the step runs through and both ranks leave together."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok")
    mp.spawn(_worker_sync_bn, args=(2, 29551, out), nprocs=2, join=True)
    assert open(out).read() == "ok"
