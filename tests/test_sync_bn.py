"""--sync_bn: me.fused_bn_act on torch.nn.SyncBatchNorm modules all-reduces its statistics tables, so that two ranks holding
halves of a batch normalise (forward AND backward) exactly like one process holding all rows.  Two gloo ranks on CPU, the
oracle as the kernel library (reference: tools/train.py:118-119 converts every BatchNorm of the model, the ones inside
ME.MinkowskiBatchNorm included)."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from cagroup3d_amd import _lib, me


def _case(G, C, rows, seed):
    g = torch.Generator().manual_seed(seed)
    n = sum(rows)
    return torch.randn(n, C, generator=g) * 2 + 0.5, torch.randn(n, C, generator=g), torch.randn(n, C, generator=g)


def _run(x, res, dy, bns, bounds, act):
    xs, rs = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    y = me.fused_bn_act(xs, bns, bounds, act, rs)
    (y * dy).sum().backward()
    return y.detach(), xs.grad, rs.grad, [b.weight.grad.clone() for b in bns], [b.bias.grad.clone() for b in bns]


def _worker(rank, world, port, out, liboracle):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = _lib.bind(liboracle)
    C = 64
    with _lib.use_library(lib):
        # rows per (rank, group): one group, then three groups of which one is EMPTY on rank 1
        for G, per_rank in ((1, [[37], [91]]), (3, [[20, 11, 5], [33, 0, 17]])):
            full_rows = [sum(per_rank[r][g] for r in range(world)) for g in range(G)]
            parts = [_case(G, C, per_rank[r], 100 + r) for r in range(world)]
            # the single-process reference: group-major concatenation of every rank's rows
            cat = [torch.cat([parts[r][k][sum(per_rank[r][:g]):sum(per_rank[r][:g + 1])] for g in range(G) for r in range(world)]) for k in range(3)]
            fb = (0,) + tuple(torch.tensor(full_rows).cumsum(0).tolist())
            torch.manual_seed(7)
            ref_bns = [nn.BatchNorm1d(C) for _ in range(G)]
            for b in ref_bns:
                nn.init.normal_(b.weight, 1.0, 0.2); nn.init.normal_(b.bias, 0.0, 0.2)
            my_bns = [nn.SyncBatchNorm(C) for _ in range(G)]
            for a, b in zip(my_bns, ref_bns):
                a.load_state_dict(b.state_dict())
            want = _run(*cat, ref_bns, fb, me.ACT_RELU)
            lb = (0,) + tuple(torch.tensor(per_rank[rank]).cumsum(0).tolist())
            got = _run(*parts[rank], my_bns, lb, me.ACT_RELU)
            # this rank's rows inside the reference's group-major order
            sel = torch.cat([torch.arange(per_rank[rank][g]) + fb[g] + sum(per_rank[r][g] for r in range(rank)) for g in range(G)])
            for k in range(3):
                torch.testing.assert_close(got[k], want[k][sel], rtol=1e-4, atol=1e-5)
            # parameter gradients are this rank's own sums: they add up to the single-process gradient
            for k in (3, 4):
                for g in range(G):
                    t = got[k][g].clone()
                    dist.all_reduce(t)
                    torch.testing.assert_close(t, want[k][g], rtol=1e-4, atol=1e-4)
            for a, b in zip(my_bns, ref_bns):
                torch.testing.assert_close(a.running_mean, b.running_mean, rtol=1e-5, atol=1e-6)
                torch.testing.assert_close(a.running_var, b.running_var, rtol=1e-5, atol=1e-6)
                assert int(a.num_batches_tracked) == int(b.num_batches_tracked) == 1
            # evaluation mode: running statistics, no collective
            for b in my_bns:
                b.eval()
            y = me.fused_bn_act(parts[rank][0], my_bns, lb, me.ACT_NONE)
            assert torch.isfinite(y).all()
        # A rank with NO rows at all (an RoI head that received no proposals): the
        # collectives of forward AND backward must still pair up (a hang here is the failure), the statistics are the
        # other rank's, and the running statistics stay identical on both ranks.
        torch.manual_seed(11)
        bn = nn.SyncBatchNorm(C)
        ref = nn.BatchNorm1d(C)
        ref.load_state_dict(bn.state_dict())
        x_full = torch.randn(23, C) * 3 + 1
        x = (x_full if rank == 0 else x_full[:0]).clone().requires_grad_(True)
        y = me.fused_bn_act(x, [bn], (0, x.shape[0]), me.ACT_RELU)
        follow = nn.SyncBatchNorm(C)                       # a second layer on top (rank 1: zero rows through both)
        z = me.fused_bn_act(y, [follow], (0, y.shape[0]), me.ACT_NONE)
        loss = z.square().sum() if rank == 0 else z.sum() * 0.0
        loss.backward()
        xr = x_full.clone().requires_grad_(True)
        yr = torch.relu(ref(xr))
        if rank == 0:
            torch.testing.assert_close(y.detach(), yr.detach(), rtol=1e-4, atol=1e-5)
            assert x.grad is not None and torch.isfinite(x.grad).all()
        torch.testing.assert_close(bn.running_mean, ref.running_mean, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(bn.running_var, ref.running_var, rtol=1e-5, atol=1e-6)
        assert int(bn.num_batches_tracked) == 1
    if rank == 0:
        open(out, "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_fused_bn_over_two_ranks_equals_one_process(tmp_path, oracle):
    out = str(tmp_path / "ok")
    mp.spawn(_worker, args=(2, 29547, out, oracle.path), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_convert_sync_batchnorm_reaches_the_minkowski_layers():
    net = nn.Sequential(me.MinkowskiBatchNorm(64), nn.BatchNorm1d(8))
    net = nn.SyncBatchNorm.convert_sync_batchnorm(net)
    assert isinstance(net[0].bn, nn.SyncBatchNorm) and isinstance(net[1], nn.SyncBatchNorm)
    assert me._sync_group_of(net[0].bn) is None            # no process group in this process: plain statistics
