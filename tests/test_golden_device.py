"""-m gpu: the reference-generated golden vectors (tests/golden/reference_vectors.npz, produced by importing the
reference's own modules -- tests/golden/make_fixtures.py) checked on DEVICE tensors through the HIP library: the same
comparisons tests/test_golden.py makes on the CPU oracle, so the coder, assigner, losses, proposal target layer and the
rotated-IoU torch half meet the reference's outputs on the MI355X as well."""
import types

import pytest
import torch

import test_golden as tg
from cagroup3d_amd import _lib

pytestmark = pytest.mark.gpu

_ON_DEVICE = [name for name in dir(tg) if name.startswith("test_") and name not in (
    "test_bev_iou_oracle_vs_compiled_reference",)]      # that one compares the ORACLE with the compiled reference (CPU only)


@pytest.fixture()
def device_golden(hip, monkeypatch):
    """Golden arrays land on the GPU, comparisons copy back; the `oracle` fixture argument of the CPU tests is replaced by
    the HIP library."""
    real_t = tg.t
    monkeypatch.setattr(tg, "t", lambda name: real_t(name).cuda())
    real_close = tg.close

    def close(a, name, rtol=1e-5, atol=1e-6):
        torch.testing.assert_close(a.cpu() if torch.is_tensor(a) else a, real_t(name), rtol=rtol, atol=atol, equal_nan=True)
    monkeypatch.setattr(tg, "close", close)
    real_ones = torch.ones
    monkeypatch.setattr(torch, "ones", lambda *a, **k: real_ones(*a, **({"device": "cuda"} | k)))   # the tests' own helper tensors
    with _lib.use_library(hip):
        yield hip


@pytest.mark.parametrize("name", _ON_DEVICE)
def test_golden_on_device(device_golden, name):
    fn = getattr(tg, name)
    import inspect
    params = inspect.signature(fn).parameters
    kwargs = {}
    if "oracle" in params:
        kwargs["oracle"] = device_golden
    if any(p not in ("oracle",) for p in params):
        pytest.skip("needs fixtures this wrapper does not provide: %s" % list(params))
    fn(**kwargs)
