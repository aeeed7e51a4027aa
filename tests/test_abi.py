"""not gpu: the C-ABI shared libraries load and export every symbol include/cagroup3d_hip.h declares
(no compute calls on the HIP library without a GPU)."""
import os
import re

import pytest
import torch

from cagroup3d_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = "".join(open(os.path.join(ROOT, "include", f)).read() for f in ("cagroup3d_hip.h", "cagroup3d_stages.h", "cagroup3d_program.h"))
    src = src.split("#ifdef CG3D_PROGRAM_IMPL")[0] + src.split("#endif /* CG3D_PROGRAM_IMPL */")[-1]      # (the inline dispatcher is not a declaration)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cg3d_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    syms = header_symbols()
    assert len(syms) >= 27
    assert set(syms) == set(_lib.EXPORTED_SYMBOLS)


def test_hip_library_exports_every_symbol():
    assert os.path.exists(_lib.HIP_LIB_PATH), "run `python cagroup3d_amd/csrc/build.py`"
    lib = _lib.bind(_lib.HIP_LIB_PATH)            # resolves every symbol, sets signatures
    assert lib.is_device and lib.raw("cg3d_abi_version")() == 2
    assert lib.raw("cg3d_hash_capacity")(1000) == 2048   # host-only helper


def test_oracle_exports_every_symbol(oracle):
    assert not oracle.is_device and oracle.raw("cg3d_abi_version")() == 2
    assert oracle.raw("cg3d_coord_map_ws_bytes")(4096) == _lib.bind(_lib.HIP_LIB_PATH).raw("cg3d_coord_map_ws_bytes")(4096)


def test_product_path_has_no_cpu_fallback():
    """Ops handed CPU tensors must fail loudly while the HIP library is bound."""
    from cagroup3d_amd import me
    from cagroup3d_amd.ops import iou3d_nms_utils
    with _lib.use_library(_lib.bind(_lib.HIP_LIB_PATH)):
        with pytest.raises(_lib.CG3DError):
            me.SparseTensor(coordinates=torch.zeros(4, 4), features=torch.zeros(4, 3))
        with pytest.raises(_lib.CG3DError):
            iou3d_nms_utils.boxes_iou_bev(torch.zeros(2, 7), torch.zeros(2, 7))


def test_product_never_loads_oracle():
    """No module under cagroup3d_amd/ may load, import or link anything under oracle/ (comments that
    merely NAME the checker are fine)."""
    pkg = os.path.join(ROOT, "cagroup3d_amd")
    pat = re.compile(r"(CDLL|bind|import_module|__import__|open|check_call|system)\s*\([^)]*oracle|^\s*(from|import)\s+oracle|#include\s+[\"<][^\">]*oracle")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                for ln in open(os.path.join(dp, f)):
                    assert not pat.search(ln), (os.path.join(dp, f), ln)
