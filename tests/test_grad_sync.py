"""TwoBucketGradSync (cagroup3d_amd/grad_sync.py) on two gloo ranks: same averaged gradients as torch DDP, early bucket
sent from inside the backward pass."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from cagroup3d_amd.grad_sync import TwoBucketGradSync


class Backbone(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1, self.layer3 = nn.Linear(8, 16), nn.Linear(16, 16)
        self.grad_sync = None

    def forward(self, x):
        s = torch.relu(self.conv1(x))
        if self.grad_sync is not None:
            self.grad_sync.attach_mid(s)
        return self.layer3(s)


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone_3d = Backbone()
        self.dense_head = nn.Linear(16, 4)
        self.roi_head = nn.Linear(16, 2)
        self.unused = nn.Parameter(torch.zeros(3))
        self.grad_sync = None

    def forward(self, x):
        f = self.backbone_3d(x)
        if self.grad_sync is not None:
            self.grad_sync.attach(f)
        return self.dense_head(f).pow(2).sum() + self.roi_head(f).abs().sum()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    ref = Toy()
    torch.manual_seed(0)
    mine = Toy()
    ddp = nn.parallel.DistributedDataParallel(ref, find_unused_parameters=True)
    mine.grad_sync = TwoBucketGradSync(mine)
    sent_early = []
    orig = mine.grad_sync._on_backbone_output_grad
    mine.grad_sync._on_backbone_output_grad = lambda g: (sent_early.append(1), orig(g))[1]
    for step in range(2):
        x = torch.randn(5, 8, generator=torch.Generator().manual_seed(10 * step + rank))
        ddp.zero_grad(); mine.zero_grad()
        ddp(x).backward()
        mine(x).backward()
        mine.grad_sync.finish()
        for (n, a), (_, b) in zip(ref.named_parameters(), mine.named_parameters()):
            if a.grad is None:
                assert b.grad is None or float(b.grad.abs().max()) == 0.0, n
            else:
                torch.testing.assert_close(b.grad, a.grad, rtol=1e-6, atol=1e-7)
    gs = mine.grad_sync
    assert len(sent_early) == 2 and len(gs.early) == 4 and len(gs.mid) == 3 and len(gs.late) == 2
    if rank == 0:
        open(out, "w").write("ok")
    dist.destroy_process_group()


def test_two_bucket_sync_matches_ddp(tmp_path):
    out = str(tmp_path / "ok")
    mp.spawn(_worker, args=(2, 29533, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def _worker_uneven(rank, world, port, out):
    """The collectives must leave in the same order (early, mid, late) on every rank whichever hooks fire.  Rank 1 plays a
    rank whose heads got nothing to do: its backbone output is never hooked (the early bucket cannot leave from its hook)
    and in the second step its loss does not reach the roi head at all (zero gradients travel instead).  With the early
    bucket sent from finish() -- after the mid bucket from ITS hook -- rank 1 would issue (mid, early, late) against rank
    0's (early, mid, late): mismatched all-reduces, i.e. wrong sums on gloo and a hang on RCCL."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    ref = Toy()
    torch.manual_seed(0)
    mine = Toy()
    mine.grad_sync = TwoBucketGradSync(mine)
    gs = mine.grad_sync
    order = []
    real_reduce = gs._reduce
    gs._reduce = lambda flat, async_op: (order.append(next(k for k, v in gs._buf.items() if v[0] is flat)), real_reduce(flat, async_op))[1]
    if rank == 1:
        gs.attach = lambda tensor: (setattr(gs, "_early_sent", False), setattr(gs, "_work", None))      # never hooks the tensor
    for step in range(2):
        xs = [torch.randn(5, 8, generator=torch.Generator().manual_seed(10 * step + r)) for r in range(world)]

        def loss_of(m, r, x):
            f = m.backbone_3d(x)
            if m.grad_sync is not None:
                m.grad_sync.attach(f)
            out = m.dense_head(f).pow(2).sum()
            return out if (r == 1 and step == 1) else out + m.roi_head(f).abs().sum()
        ref.zero_grad(set_to_none=True)
        for r, x in enumerate(xs):                      # the average over ranks, computed locally
            (loss_of(ref, r, x) / world).backward()
        mine.zero_grad(set_to_none=True)
        del order[:]
        loss_of(mine, rank, xs[rank]).backward()
        gs.finish()
        assert order == ["early", "mid", "late"], (rank, step, order)
        for (n, a), (_, b) in zip(ref.named_parameters(), mine.named_parameters()):
            if a.grad is None:
                assert b.grad is None or float(b.grad.abs().max()) == 0.0, n
            else:
                torch.testing.assert_close(b.grad, a.grad, rtol=1e-5, atol=1e-6, msg=lambda m: "%s step %d: %s" % (n, step, m))
    if rank == 0:
        open(out, "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_collective_order_does_not_depend_on_which_hooks_fire(tmp_path):
    out = str(tmp_path / "ok")
    mp.spawn(_worker_uneven, args=(2, 29536, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def _worker_eight(rank, world, port, out):
    """Eight ranks (the node the bench scales to: reference tools/train.py:143-144, tools/scripts/dist_train.sh:1-18) with
    UNEVEN shards and one rank whose scenes hold no ground truth: its loss never reaches the roi head (zero gradients travel
    in the early bucket), its normaliser row is all zeros.  Checked on every rank: the collectives leave in the order early,
    mid, late; every parameter gradient equals the locally computed average over the eight ranks' losses; the (B, 3) loss
    normaliser of cagroup_utils.reduce_mean (cagroup_utils.py:6-12) is the mean over ranks, zeros included."""
    from cagroup3d_amd.pcdet.models.model_utils.cagroup_utils import reduce_mean
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    torch.manual_seed(0)
    ref = Toy()
    torch.manual_seed(0)
    mine = Toy()
    # ranks start from rank 0's parameters whatever they were initialised with (broadcast in the constructor)
    if rank != 0:
        with torch.no_grad():
            for p in mine.parameters():
                p.add_(float(rank))
    mine.grad_sync = TwoBucketGradSync(mine)
    for (n, a), (_, b) in zip(ref.named_parameters(), mine.named_parameters()):
        torch.testing.assert_close(b.detach(), a.detach(), rtol=0, atol=0, msg=lambda m: "broadcast of %s: %s" % (n, m))
    gs = mine.grad_sync
    order = []
    real_reduce = gs._reduce
    gs._reduce = lambda flat, async_op: (order.append(next(k for k, v in gs._buf.items() if v[0] is flat)), real_reduce(flat, async_op))[1]
    no_gt = 5
    rows = [3 + (r * 5) % 4 for r in range(world)]                     # 3..6 rows per rank
    for step in range(2):
        xs = [torch.randn(rows[r], 8, generator=torch.Generator().manual_seed(100 * step + r)) for r in range(world)]

        def loss_of(m, r, x):
            f = m.backbone_3d(x)
            if m.grad_sync is not None:
                m.grad_sync.attach(f)
            out = m.dense_head(f).pow(2).sum()
            return out if r == no_gt else out + m.roi_head(f).abs().sum()
        ref.zero_grad(set_to_none=True)
        for r, x in enumerate(xs):
            (loss_of(ref, r, x) / world).backward()
        mine.zero_grad(set_to_none=True)
        del order[:]
        loss_of(mine, rank, xs[rank]).backward()
        gs.finish()
        assert order == ["early", "mid", "late"], (rank, step, order)
        for (n, a), (_, b) in zip(ref.named_parameters(), mine.named_parameters()):
            if a.grad is None:
                assert b.grad is None or float(b.grad.abs().max()) == 0.0, n
            else:
                torch.testing.assert_close(b.grad, a.grad, rtol=1e-5, atol=1e-6, msg=lambda m: "%s rank %d step %d: %s" % (n, rank, step, m))
        # the loss normalisers: one (B, 3) all-reduce per step (positives, centerness sum, vote count per scene); the
        # ground-truth-free rank contributes zeros and still takes part
        B = 4
        stats = [torch.zeros(B, 3) if r == no_gt else torch.arange(B * 3, dtype=torch.float32).view(B, 3) * (r + 1) + step for r in range(world)]
        got = reduce_mean(stats[rank])
        torch.testing.assert_close(got, torch.stack(stats).mean(0), rtol=1e-6, atol=1e-6)
    if rank == 0:
        open(out, "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_uneven_shards_and_a_rank_without_ground_truth(tmp_path):
    out = str(tmp_path / "ok")
    mp.spawn(_worker_eight, args=(8, 29541, out), nprocs=8, join=True)
    assert open(out).read() == "ok"


def _worker_gpu(rank, world, port, out):
    """Both ranks on cuda:0 over gloo (RCCL refuses two ranks on one device): the bucket packing, the asynchronous
    exchange started inside backward and finish() run on device tensors and streams."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    ref = Toy().to(dev)
    torch.manual_seed(0)
    mine = Toy().to(dev)
    mine.grad_sync = TwoBucketGradSync(mine)
    for step in range(3):
        xs = [torch.randn(4096, 8, generator=torch.Generator().manual_seed(10 * step + r)).to(dev) for r in range(world)]
        ref.zero_grad(set_to_none=True)
        for x in xs:                                   # the average over ranks, computed locally
            (ref(x) / world).backward()
        mine.zero_grad(set_to_none=True)
        busy = torch.randn(2048, 2048, device=dev)
        for _ in range(8):
            busy = busy @ busy * 1e-3                  # a long queue in front of the backward pass
        (mine(xs[rank]) + 0.0 * busy.sum()).backward()
        mine.grad_sync.finish()
        torch.cuda.synchronize()
        for (n, a), (_, b) in zip(ref.named_parameters(), mine.named_parameters()):
            if a.grad is None:
                assert b.grad is None or float(b.grad.abs().max()) == 0.0, n
            else:
                torch.testing.assert_close(b.grad, a.grad, rtol=1e-4, atol=1e-5)
    if rank == 0:
        open(out, "w").write("ok")
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_bucket_sync_on_device_tensors(tmp_path):
    assert torch.cuda.is_available(), "gpu test needs a GPU"
    out = str(tmp_path / "ok")
    mp.spawn(_worker_gpu, args=(2, 29534, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def _worker_detector(rank, world, port, out):
    """The real detector, two ranks on cuda:0 over gloo, one training step each on its own scenes: the gradients after
    TwoBucketGradSync.finish() equal the average of the two ranks' gradients computed locally."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from cagroup3d_amd import build_model, me
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    me.PRECISION = 0
    # the loss normalisers are averaged over ranks (reduce_mean, cagroup_head.py:523-538 of the reference): a rank's loss is
    # then not the loss of its scenes alone.  Switched off here so that the locally computed average is the exact
    # reference for the gradient exchange, which is what this test is about.
    from cagroup3d_amd.pcdet.models.dense_heads import cagroup_head as H
    H.reduce_mean = lambda t: t
    ref, _ = bench.make_model("scannet", True, dev)
    mine, _ = bench.make_model("scannet", True, dev)
    mine.load_state_dict(ref.state_dict())
    ref.train(); mine.train()
    batches = [build_model.synthetic_batch("S5k", 2, first_scene=2 * r, device=dev) for r in range(world)]
    ref2, _ = bench.make_model("scannet", True, dev)
    ref2.load_state_dict(ref.state_dict())
    ref2.train()
    for m in (ref, ref2):                               # the average over ranks, computed locally -- twice, to calibrate
        for b in batches:                               # the run-to-run noise of the fp32 atomics
            ret, _, _ = m(bench.fresh(b))
            (ret["loss"] / world).backward()
    mine.grad_sync = TwoBucketGradSync(mine)
    ret, _, _ = mine(bench.fresh(batches[rank]))
    ret["loss"].backward()
    mine.grad_sync.finish()
    torch.cuda.synchronize()
    gs = mine.grad_sync
    assert gs._buf and len(gs.early) > 200 and len(gs.mid) > 50 and len(gs.late) > 10
    # per bucket: relative L2 distance between the exchanged gradients and the locally computed average.  (Per-parameter
    # maxima are useless here: BatchNorm over a class map of a handful of rows amplifies the fp32 summation-order noise
    # of the atomics to O(1) on a few tiny gradients, in the reference pass as much as in the exchanged one.)
    ids = {"early": {id(p) for p in gs.early}, "mid": {id(p) for p in gs.mid}, "late": {id(p) for p in gs.late}}
    def rel_l2(ma, mb):
        num = {k: 0.0 for k in ids}
        den = {k: 0.0 for k in ids}
        for a, b, c in zip(ma.parameters(), mb.parameters(), mine.parameters()):
            ga = torch.zeros_like(a) if a.grad is None else a.grad      # an unused parameter's bucket slot holds zeros
            gb = torch.zeros_like(b) if b.grad is None else b.grad
            k = next(k for k in ids if id(c) in ids[k])
            num[k] += float((ga - gb).double().pow(2).sum())
            den[k] += float(ga.double().pow(2).sum())
        assert all(den[k] > 0 for k in ids), den
        return {k: (num[k] / den[k]) ** 0.5 for k in ids}
    noise, rel = rel_l2(ref, ref2), rel_l2(ref, mine)
    worst = max(rel.values())
    assert all(rel[k] < max(5e-3, 5.0 * noise[k]) for k in ids), (rel, noise)     # the noise itself varies run to run (0.03-0.3 %)
    if rank == 0:
        open(out, "w").write("ok %g" % worst)
    dist.destroy_process_group()


@pytest.mark.gpu
def test_detector_gradients_two_ranks_on_device(tmp_path):
    assert torch.cuda.is_available(), "gpu test needs a GPU"
    out = str(tmp_path / "ok")
    mp.spawn(_worker_detector, args=(2, 29535, out), nprocs=2, join=True)
    assert open(out).read().startswith("ok")


def _worker_bf16_wire(rank, world, port, out):
    """CG3D_GRAD_BF16=1: the buckets travel as bf16; the averaged gradients agree with the fp32 exchange to bf16 precision
    and `report()` names the bucket bytes and the wire type."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CG3D_GRAD_BF16="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    mine = Toy()
    mine.grad_sync = TwoBucketGradSync(mine)
    assert mine.grad_sync.grad_dtype == torch.bfloat16
    xs = [torch.randn(64, 8, generator=torch.Generator().manual_seed(r)) for r in range(world)]
    ref = [None]
    torch.manual_seed(0)
    refm = Toy()
    for x in xs:
        (refm(x) / world).backward()
    mine(xs[rank]).backward()
    mine.grad_sync.finish()
    for (n, a), (_, b) in zip(refm.named_parameters(), mine.named_parameters()):
        if a.grad is not None:
            torch.testing.assert_close(b.grad, a.grad, rtol=2e-2, atol=2e-2 * float(a.grad.abs().max()) + 1e-6)
    rep = mine.grad_sync.report()
    assert rep["wire_dtype"] == "torch.bfloat16" and rep["world"] == 2 and sum(rep["bucket_bytes"].values()) > 0
    if rank == 0:
        open(out, "w").write("ok")
    dist.destroy_process_group()


def test_bf16_wire_format_and_report(tmp_path):
    out = str(tmp_path / "ok")
    mp.spawn(_worker_bf16_wire, args=(2, 29537, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def _worker_rccl(rank, world, port, out):
    """One rank per GPU over RCCL ("nccl" backend on ROCm): the product's data-parallel path as the driver's 2/4/8-GPU
    bench runs it -- bucket exchange with ReduceOp.AVG from inside backward, gradients == the locally computed average."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = torch.device("cuda", rank)
    torch.manual_seed(0)
    ref = Toy().to(dev)
    torch.manual_seed(0)
    mine = Toy().to(dev)
    mine.grad_sync = TwoBucketGradSync(mine)
    for step in range(3):
        xs = [torch.randn(4096, 8, generator=torch.Generator().manual_seed(10 * step + r)).to(dev) for r in range(world)]
        ref.zero_grad(set_to_none=True)
        for x in xs:
            (ref(x) / world).backward()
        mine.zero_grad(set_to_none=True)
        mine(xs[rank]).backward()
        mine.grad_sync.finish()
        torch.cuda.synchronize()
        for (n, a), (_, b) in zip(ref.named_parameters(), mine.named_parameters()):
            if a.grad is not None:
                torch.testing.assert_close(b.grad, a.grad, rtol=1e-4, atol=1e-5)
    rep = mine.grad_sync.report()
    assert rep["exposed_ms_per_step_median"] is not None
    if rank == 0:
        open(out, "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_over_rccl_when_two_gpus_are_visible(tmp_path):
    """Runs wherever >= 2 GPUs are visible (the driver's multi-GPU box); the 1-GPU gpurun box skips it -- RCCL refuses two
    ranks on one device, which is why the one-device tests above go over gloo."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL: one rank per device)")
    out = str(tmp_path / "ok")
    mp.spawn(_worker_rccl, args=(2, 29538, out), nprocs=2, join=True)
    assert open(out).read() == "ok"
