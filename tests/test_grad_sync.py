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
