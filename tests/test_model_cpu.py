"""not gpu: the host-side detector (pcdet-API mirror) end to end on the CPU oracle, the batched class
branches against the reference-order loop, and the multi-process (gloo, world_size 2) data-parallel path."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cagroup3d_amd import _lib, build_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_step(model, cfgname, batched, n_scenes=2, first=0):
    model.dense_head.batched = batched
    model.zero_grad()
    torch.manual_seed(1)
    np.random.seed(1)
    batch = build_model.synthetic_batch(cfgname, n_scenes, first_scene=first, device="cpu")
    ret, tb, disp = model(batch)
    ret["loss"].backward()
    grads = torch.cat([p.grad.flatten() for p in model.parameters()])
    return batch, tb, grads


@pytest.mark.parametrize("dataset,cfgname", [("scannet", "S5k"), ("sunrgbd", "S5k-yaw")])
def test_batched_class_branches_equal_reference_order_loop(oracle, dataset, cfgname):
    with _lib.use_library(oracle):
        model, cfg = build_model.build_cagroup3d(dataset)
        model.train()
        model.dense_head.force_gt_selection = True
        model.dense_head.force_class_logit_boost = 6.0
        b0, tb0, g0 = _run_step(model, cfgname, False)
        b1, tb1, g1 = _run_step(model, cfgname, True)
    for k in tb0:
        assert abs(tb0[k] - tb1[k]) <= 1e-4 * max(1.0, abs(tb0[k])), (k, tb0[k], tb1[k])
    x0, x1 = b0["one_stage_results"][0], b1["one_stage_results"][0]
    for li in range(4):
        for c in range(len(x0[li])):
            for s in range(2):
                torch.testing.assert_close(x1[li][c][s], x0[li][c][s], rtol=1e-4, atol=1e-4)
    for p0, p1 in zip(b0["pred_bbox_list"], b1["pred_bbox_list"]):
        assert len(p0[0]) > 10
        for a, b in zip(p0, p1):
            torch.testing.assert_close(b.float(), a.float(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(g1, g0, rtol=1e-3, atol=1e-5)
    assert tb0["loss_centerness"] > 0 and tb0["loss_bbox"] > 0      # positives exist: every loss term is exercised


def test_eval_mode_contract(oracle):
    with _lib.use_library(oracle):
        model, cfg = build_model.build_cagroup3d("scannet")
        model.eval()
        model.dense_head.force_gt_selection = True
        model.dense_head.force_class_logit_boost = 6.0
        batch = build_model.synthetic_batch("S5k", 2, device="cpu")
        with torch.no_grad():
            pred_dicts, recall = model(batch)
    assert len(pred_dicts) == 2
    for p in pred_dicts:
        n = p["pred_boxes"].shape[0]
        assert n > 0 and p["pred_boxes"].shape == (n, 7) and p["pred_scores"].shape == (n,) and p["pred_labels"].shape == (n,)
        assert p["pred_labels"].dtype == torch.long and (p["pred_scores"] > 0.01).all()
    assert "gt" in recall


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cagroup3d_amd import _lib as L, build_model as bm
    with L.use_library(L.bind(os.path.join(ROOT, "oracle", "liboracle.so"))):
        model, cfg = bm.build_cagroup3d("scannet", seed=0)
        model.train()
        model.dense_head.force_gt_selection = True
        model.dense_head.force_class_logit_boost = 6.0
        from cagroup3d_amd.grad_sync import TwoBucketGradSync
        model.grad_sync = TwoBucketGradSync(model)       # the product's exchange (bench.py / train.py at WORLD_SIZE > 1)
        torch.manual_seed(1)
        np.random.seed(1)
        batch = bm.synthetic_batch("S5k", 1, first_scene=rank, device="cpu")     # scene i -> rank i mod W
        ret, tb, _ = model(batch)
        ret["loss"].backward()
        model.grad_sync.finish()
        g = torch.cat([p.grad.flatten() for p in model.parameters()])
        q.put((rank, float(ret["loss"]), g[::997].clone().numpy(), float(g.abs().sum())))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_data_parallel_two_ranks_gloo(oracle):
    """One scene per rank, the bucketed gradient all-reduce (grad_sync.py) + the fused reduce_mean all-reduce on gloo:
    both ranks end with identical (averaged) gradients, different from a single-rank run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=800) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, l0, g0, s0), (_, l1, g1, s1) = res
    assert l0 != l1                                    # different scenes per rank
    np.testing.assert_allclose(g0, g1, rtol=1e-5, atol=1e-7)     # all-reduced gradients agree
    assert abs(s0 - s1) <= 1e-4 * s0 and s0 > 0


# ------------------------------------------------------------------ checkpoints written by the reference's stack
def test_me_kernel_order_converter_matches_the_stated_minkowski_convention(oracle):
    """convert_me_kernel_order: a 3^3 kernel stored in MinkowskiEngine's offset order (first spatial axis fastest, as
    the docstring restates ME v0.5.4 -- ME itself is not in the image, so the convention is restated, not pinned by one
    of its own vectors) gives, after conversion, the same convolution on THIS engine as a brute-force sum that walks the
    ME order directly.  Also: the conversion is an involution and leaves non-cubic tensors alone."""
    from cagroup3d_amd import me
    from cagroup3d_amd.pcdet.models.detectors.detector3d_template import convert_me_kernel_order
    torch.manual_seed(0)
    cin, cout, k = 4, 5, 3
    coords = torch.unique(torch.cat([torch.zeros(60, 1), torch.randint(0, 5, (60, 3)).float()], 1), dim=0)
    feats = torch.randn(coords.shape[0], cin)
    w_me = torch.randn(k ** 3, cin, cout)
    state = {"conv.kernel": w_me, "lin.kernel": torch.randn(cin, cout), "bn.weight": torch.randn(5), "odd.kernel": torch.randn(10, 2, 2)}
    conv_state = convert_me_kernel_order(state)
    assert torch.equal(conv_state["lin.kernel"], state["lin.kernel"]) and torch.equal(conv_state["odd.kernel"], state["odd.kernel"])
    assert torch.equal(convert_me_kernel_order(conv_state)["conv.kernel"], w_me)
    with _lib.use_library(oracle):
        x = me.SparseTensor(coordinates=coords, features=feats)
        conv = me.MinkowskiConvolution(cin, cout, kernel_size=k, dimension=3)
        with torch.no_grad():
            conv.kernel.copy_(conv_state["conv.kernel"])
        y = conv(x)
        C, F = y.C, y.F.detach()
    # brute force in ME order: offset index = ix + k*(iy + k*iz), offsets -1..1
    table = {tuple(c.tolist()): i for i, c in enumerate(x.C)}
    ref = torch.zeros(len(C), cout)
    for o, c in enumerate(C.tolist()):
        for iz in range(k):
            for iy in range(k):
                for ix in range(k):
                    nb = (c[0], c[1] + ix - 1, c[2] + iy - 1, c[3] + iz - 1)
                    j = table.get(nb)
                    if j is not None:
                        ref[o] += x.F[j] @ w_me[ix + k * (iy + k * iz)]
    torch.testing.assert_close(F, ref, rtol=1e-5, atol=1e-5)


def test_checkpoint_loader_converts_foreign_checkpoints_and_refuses_their_optimizer_state(oracle, tmp_path):
    from cagroup3d_amd.pcdet.models.detectors.detector3d_template import CHECKPOINT_VERSION, convert_me_kernel_order
    with _lib.use_library(oracle):
        model, _ = build_model.build_cagroup3d("scannet", seed=0)
        other, _ = build_model.build_cagroup3d("scannet", seed=1)
    native = {k: v.clone() for k, v in model.state_dict().items()}
    name = next(k for k, v in native.items() if k.endswith(".kernel") and v.dim() == 3 and v.shape[0] == 27)
    # (a) a checkpoint of this build: loaded as is
    f_native = str(tmp_path / "native.pth")
    torch.save({"model_state": native, "version": CHECKPOINT_VERSION, "optimizer_state": None, "it": 7, "epoch": 2}, f_native)
    other.load_params_from_file(f_native, to_cpu=True)
    assert torch.equal(other.state_dict()[name], native[name])
    # (b) the same weights as the reference's stack would have written them (ME order, its own / no version tag)
    f_me = str(tmp_path / "me.pth")
    torch.save({"model_state": convert_me_kernel_order(native), "version": "pcdet+0.5.2", "optimizer_state": {"state": {}, "param_groups": []}}, f_me)
    assert not torch.equal(convert_me_kernel_order(native)[name], native[name])
    with _lib.use_library(oracle):
        fresh, _ = build_model.build_cagroup3d("scannet", seed=2)
    fresh.load_params_from_file(f_me, to_cpu=True)
    assert torch.equal(fresh.state_dict()[name], native[name]), "foreign checkpoint must come out in this engine's order"
    fresh.load_params_from_file(f_me, to_cpu=True, kernel_order="native")
    assert torch.equal(fresh.state_dict()[name], convert_me_kernel_order(native)[name])
    opt = torch.optim.AdamW(fresh.parameters(), lr=1e-3)
    with pytest.raises(ValueError):
        fresh.load_params_with_optimizer(f_me, to_cpu=True, optimizer=opt)
    it, ep = fresh.load_params_with_optimizer(f_native, to_cpu=True, optimizer=None)
    assert (it, ep) == (7, 2)


def test_deferred_log_reads_the_device_once_and_behaves_like_a_dict():
    import json
    from cagroup3d_amd.pcdet.utils.common_utils import DeferredLog
    one = DeferredLog(("loss_a", "loss_b", "one_stage_loss"), torch.tensor([1.0, 2.0, 3.0]))
    two = DeferredLog(["rcnn", "loss_two_stage"], torch.tensor([0.5, 0.5]))
    tb = DeferredLog(("loss_all",), torch.tensor([3.5])).absorb(one).absorb(two).absorb({"extra": 7})
    assert len(tb._pend) == 3 and dict.__len__(tb) == 1              # nothing read yet
    assert tb["loss_all"] == 3.5 and not tb._pend                     # the first look materialises everything
    assert list(tb) == ["extra", "loss_all", "loss_a", "loss_b", "one_stage_loss", "rcnn", "loss_two_stage"]
    assert dict(tb)["rcnn"] == 0.5 and {**tb}["loss_b"] == 2.0 and tb.get("missing", -1) == -1 and "loss_a" in tb
    assert json.loads(json.dumps(tb))["one_stage_loss"] == 3.0 and isinstance(tb, dict) and len(tb) == 7
    assert one["loss_a"] == 1.0                                       # the absorbed log can still be read on its own


def test_deferred_log_write_and_remove_methods_see_the_pending_values():
    """pop / setdefault / popitem / == / | / update on a log whose numbers are still pending; a key the user wrote before the
    first read survives it; two logs that absorbed the same block share ONE host read."""
    from cagroup3d_amd.pcdet.utils.common_utils import DeferredLog
    mk = lambda: DeferredLog(("loss_bbox", "loss_cls"), torch.tensor([1.5, 2.5]))   # noqa: E731
    assert mk().pop("loss_bbox") == 1.5 and mk().pop("nope", None) is None
    assert mk().setdefault("loss_cls", 9.0) == 2.5 and mk().setdefault("new", 9.0) == 9.0
    assert mk().popitem() == ("loss_cls", 2.5)
    assert mk() == {"loss_bbox": 1.5, "loss_cls": 2.5} and mk() == mk() and mk() != {"loss_bbox": 1.5}
    assert (mk() | {"x": 1}) == {"loss_bbox": 1.5, "loss_cls": 2.5, "x": 1} and ({"x": 1} | mk())["loss_cls"] == 2.5
    d = mk()
    d["loss_bbox"] = -1.0                       # written BEFORE the first read: must not be overwritten by the device value
    d.update(loss_cls=-2.0, other=3)
    assert d == {"loss_bbox": -1.0, "loss_cls": -2.0, "other": 3}
    d = mk()
    del d["loss_bbox"]
    assert list(d) == ["loss_cls"]
    src = mk()
    a, b = DeferredLog().absorb(src), DeferredLog().absorb(src)
    assert a["loss_cls"] == 2.5 and src._pend[0].host == [1.5, 2.5]      # materialised once ...
    src._pend[0].values = None                                           # ... so the second log never touches the tensor
    assert b["loss_bbox"] == 1.5
