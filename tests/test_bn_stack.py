"""me.BNStack (per-class BatchNorm modules addressed as [G, C] arrays), me.pairs_many (one host read for several pair lists)
and engine._pg_views (parameter-gradient buffers reused per layout): host logic, on the CPU oracle."""
import copy

import numpy as np
import torch

from cagroup3d_amd import _lib, engine, me


def _bns(G, C, seed=0):
    torch.manual_seed(seed)
    bns = [torch.nn.BatchNorm1d(C) for _ in range(G)]
    for b in bns:
        b.weight.data.uniform_(0.5, 1.5)
        b.bias.data.normal_()
        b.running_mean.normal_()
        b.running_var.uniform_(0.5, 2)
    return bns


def test_grouped_bn_on_stacked_modules_equals_the_per_module_form(oracle):
    """Same outputs, input gradient, parameter gradients and running statistics whether the G modules' tensors are slices of
    one [G, C] array (statistics updated inside the apply launch) or separate tensors (stack + _foreach updates)."""
    with _lib.use_library(oracle):
        G, C = 5, 64
        bns = _bns(G, C)
        ref = copy.deepcopy(bns)
        bounds = (0, 10, 33, 40, 90, 128)
        x = torch.randn(128, C)
        x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        keep = me.GROUPED_BN_STACK
        try:
            me.GROUPED_BN_STACK = True
            y1 = me.fused_bn_act(x1, bns, bounds, me.ACT_ELU)
            me.GROUPED_BN_STACK = False
            y2 = me.fused_bn_act(x2, ref, bounds, me.ACT_ELU)
        finally:
            me.GROUPED_BN_STACK = keep
        assert torch.equal(y1, y2)
        y1.square().sum().backward()
        y2.square().sum().backward()
        assert torch.equal(x1.grad, x2.grad)
        for a, b in zip(bns, ref):
            torch.testing.assert_close(a.running_mean, b.running_mean, rtol=1e-6, atol=1e-6)
            torch.testing.assert_close(a.running_var, b.running_var, rtol=1e-6, atol=1e-6)
            assert int(a.num_batches_tracked) == int(b.num_batches_tracked) == 1
            torch.testing.assert_close(a.weight.grad, b.weight.grad, rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(a.bias.grad, b.bias.grad, rtol=1e-5, atol=1e-5)


def test_bn_stack_keeps_the_modules_and_follows_their_storage(oracle):
    G, C = 4, 32
    bns = _bns(G, C)
    before = [{k: v.clone() for k, v in b.state_dict().items()} for b in bns]
    params = [(b.weight, b.bias) for b in bns]
    st = me.BNStack.of(bns)
    # the modules keep their Parameter objects, keys, shapes and values; their storage is one array per attribute
    for g, b in enumerate(bns):
        assert b.weight is params[g][0] and b.bias is params[g][1]
        assert list(b.state_dict()) == list(before[g]) and all(torch.equal(b.state_dict()[k], before[g][k]) for k in before[g])
        assert b.weight.data_ptr() == st.weight.data_ptr() + g * C * 4 and b.running_var.data_ptr() == st.running_var.data_ptr() + g * C * 4
        assert b.num_batches_tracked.data_ptr() == st.num_batches_tracked.data_ptr() + g * 8
    assert me.BNStack.of(bns) is st
    # an in-place update of a module shows in the stack (load_state_dict copies in place) ...
    bns[2].load_state_dict({k: v + 1 for k, v in before[2].items()})
    assert torch.equal(st.weight[2], before[2]["weight"] + 1) and me.BNStack.of(bns) is st
    # ... fresh storage for any module (model.to(), a swapped buffer) is noticed and the stack rebuilt around the new values
    bns[1].running_mean.data = bns[1].running_mean.data.clone() + 3
    st2 = me.BNStack.of(bns)
    assert st2 is not st and st2.valid(bns) and torch.equal(st2.running_mean[1], before[1]["running_mean"] + 3)
    bns[3].weight.data = torch.full((C,), 7.0)
    st3 = me.BNStack.of(bns)
    assert st3 is not st2 and torch.equal(st3.weight[3], torch.full((C,), 7.0)) and bns[3].weight is params[3][0]


def test_pairs_many_equals_separate_calls(oracle):
    with _lib.use_library(oracle):
        g = torch.Generator().manual_seed(1)
        coords = torch.cat([torch.randint(0, 3, (400, 1), generator=g), torch.randint(0, 12, (400, 3), generator=g)], 1).float()
        feats = torch.randn(400, 4, generator=g)
        sp = me.SparseTensor(coordinates=coords, features=feats)
        mgr, key = sp.coordinate_manager, sp.coordinate_map_key
        n = sp.C.shape[0]
        bounds = (0, n // 3, n // 2, n)
        km_a = mgr.kernel_map(key, key, 3, 1, False)
        km_b = mgr.kernel_map(key, key, 5, 1, False)
        ident = me.KernelMap.identity(n, sp.C.device)
        many = me.pairs_many([(km_a, bounds), (km_b, None), (ident, bounds), (km_a, bounds)])
        assert many[0] is many[3]
        for (km, rb), got in zip(((km_a, bounds), (km_b, None), (ident, bounds)), many):
            fresh = me.KernelMap(km.nbr, km.K, km.n_in, km.n_out, None).pairs(rb)
            assert torch.equal(got[0][:got[3]], fresh[0][:fresh[3]]) and torch.equal(got[1][:got[3]], fresh[1][:fresh[3]])
            assert np.array_equal(got[2], fresh[2]) and got[3] == fresh[3]


class _Comp:
    def __init__(self, params):
        self.params, off = [], 0
        for p in params:
            self.params.append((p, off))
            off += (p.numel() * 4 + 255) & ~255
        self.size = {engine.R_PG: off}


def test_parameter_gradient_buffers_are_reused_only_when_no_gradient_is_pending():
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    comp = _Comp(ps)
    engine._PG_POOL.clear()
    pg1, v1, fresh1 = engine._pg_views(comp, torch.device("cpu"))
    assert not fresh1 and [tuple(v.shape) for v in v1] == [(5, 3), (7,)] and float(pg1.abs().sum()) == 0
    v1[0].fill_(2.0)
    for p, v in zip(ps, v1):
        p.grad = v
    # gradients pending (accumulation over two backward passes): a fresh buffer, the first pass's gradients untouched
    pg2, v2, fresh2 = engine._pg_views(comp, torch.device("cpu"))
    assert fresh2 and pg2.data_ptr() != pg1.data_ptr() and float(ps[0].grad.sum()) == 30.0
    # zero_grad(set_to_none=True): the pooled buffer comes back, zero-filled, with the same view objects
    for p in ps:
        p.grad = None
    pg3, v3, fresh3 = engine._pg_views(comp, torch.device("cpu"))
    assert not fresh3 and pg3 is pg1 and all(a is b for a, b in zip(v1, v3)) and float(pg3.abs().sum()) == 0
    # another parameter set of the same sizes does not get this one's views
    other = _Comp([torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))])
    assert engine._pg_views(other, torch.device("cpu"))[0] is not pg1
    engine._PG_POOL.clear()
