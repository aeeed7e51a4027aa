"""Fused focal loss (include/cagroup3d_hip.h: cg3d_focal_loss_fwd/bwd).
not gpu: the oracle against the golden value produced by the REFERENCE's FocalLoss and against the torch chain
(the mirror of py_sigmoid_focal_loss) forward and backward.  gpu: the HIP kernels against the oracle."""
import os

import numpy as np
import pytest
import torch

from cagroup3d_amd import _lib
from cagroup3d_amd.ops.focal_loss import sigmoid_focal_loss_rows
from cagroup3d_amd.pcdet.utils.loss_utils import py_sigmoid_focal_loss

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))


def chain(pred, lab, row_w, gamma, alpha):
    C = pred.shape[1]
    tgt = torch.where((lab < 0) | (lab >= C), torch.full_like(lab, C), lab)
    onehot = torch.nn.functional.one_hot(tgt, C + 1)[:, :C]
    el = py_sigmoid_focal_loss(pred, onehot, None, gamma=gamma, alpha=alpha, reduction="none")
    return (el * row_w.view(-1, 1)).sum()


def inputs(n, c, seed, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    pred = (torch.randn(n, c, generator=g) * 3).to(device).requires_grad_(True)
    lab = torch.randint(-1, c, (n,), generator=g).to(device)
    row_w = (torch.rand(n, generator=g) + 0.1).to(device)
    return pred, lab, row_w


def test_oracle_matches_reference_golden(oracle):
    # FocalLoss(gamma 2, alpha .25)(pred, target, avg_factor=7) of the reference = sum / 7 = row weight 1/7
    pred, tgt = torch.from_numpy(G["focal_pred"]), torch.from_numpy(G["focal_target"]).long()
    with _lib.use_library(oracle):
        out = sigmoid_focal_loss_rows(pred, tgt, torch.full((pred.shape[0],), 1.0 / 7.0), 2.0, 0.25)
    torch.testing.assert_close(out, torch.from_numpy(G["focal_loss"]).reshape(()), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n,c,gamma,alpha", [(257, 18, 2.0, 0.25), (64, 10, 1.5, 0.4), (1, 1, 2.0, 0.25)])
def test_oracle_matches_torch_chain(oracle, n, c, gamma, alpha):
    pred, lab, row_w = inputs(n, c, 3)
    ref = chain(pred, lab, row_w, gamma, alpha)
    (gref,) = torch.autograd.grad(ref * 1.7, pred)
    with _lib.use_library(oracle):
        out = sigmoid_focal_loss_rows(pred, lab, row_w, gamma, alpha)
        (gout,) = torch.autograd.grad(out * 1.7, pred)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gout, gref, rtol=1e-4, atol=1e-6)


def test_oracle_empty(oracle):
    with _lib.use_library(oracle):
        out = sigmoid_focal_loss_rows(torch.zeros(0, 18, requires_grad=True), torch.zeros(0, dtype=torch.long), torch.zeros(0))
    assert float(out.detach()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("n,c", [(100003, 18), (4097, 10), (0, 18)])
def test_hip_matches_oracle(hip, oracle, n, c):
    pred, lab, row_w = inputs(n, c, 5)
    with _lib.use_library(oracle):
        ref = sigmoid_focal_loss_rows(pred, lab, row_w, 2.0, 0.25)
        (gref,) = torch.autograd.grad(ref * 0.5, pred) if n else (torch.zeros_like(pred),)
    pd = pred.detach().cuda().requires_grad_(True)
    out = sigmoid_focal_loss_rows(pd, lab.cuda(), row_w.cuda(), 2.0, 0.25)
    # fp32 sums in a different order (per-block partials vs one fp64 accumulator): 1e-5 relative
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-5, atol=1e-5)
    if n:
        (gout,) = torch.autograd.grad(out * 0.5, pd)
        torch.testing.assert_close(gout.cpu(), gref, rtol=1e-4, atol=1e-6)
