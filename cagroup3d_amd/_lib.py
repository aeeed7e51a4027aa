"""ctypes binding of the C-ABI in include/cagroup3d_hip.h.

The product binds `cagroup3d_amd/csrc/libcagroup3d_hip.so` (hand-written HIP for gfx950) and
FAILS LOUDLY if it is missing or if an op is handed a non-GPU tensor: there is no CPU fallback.

`bind(path)` returns a `Library` for any shared object exporting the same symbols; the parity
tests use it to drive the CPU oracle (oracle/liboracle.so) through the identical binding, and
`use_library()` lets tests / the bench's cpu_baseline leg run the host-side engine on it.  Nothing
in this package ever loads the oracle by itself.
"""
import ctypes
import os
from ctypes import c_float, c_int32, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# CG3D_HIP_LIB: dev aid -- another BUILD of the same HIP library (tools/ab_libs.sh: two builds alternated on one box); it must
# be a device library (cg3d_is_device_library), so this cannot route the product to the oracle
HIP_LIB_PATH = os.environ.get("CG3D_HIP_LIB") or os.path.join(_HERE, "csrc", "libcagroup3d_hip.so")

CG3D_OK = 0
_ERRORS = {-1: "CG3D_ERR_ARG (bad argument)", -2: "CG3D_ERR_LAUNCH (HIP launch/runtime error)",
           -3: "CG3D_ERR_RANGE (coordinate outside packable range)"}

P = c_void_p
_SIGNATURES = {
    "cg3d_is_device_library": (c_int32, []),
    "cg3d_abi_version": (c_int32, []),
    "cg3d_h2d_async": (c_int32, [P, P, c_int64, P]),
    "cg3d_hash_capacity": (c_int64, [c_int64]),
    "cg3d_coord_map_ws_bytes": (c_int64, [c_int64]),
    "cg3d_coord_map_build": (c_int32, [P, c_int64, c_int32, P, P, c_int64, P, P, P, P, P, P]),
    "cg3d_morton_order_ws_bytes": (c_int64, [c_int64]),
    "cg3d_morton_order": (c_int32, [P, c_int64, P, P, P]),
    "cg3d_kernel_map": (c_int32, [P, c_int64, P, c_int32, P, P, c_int64, P, P]),
    "cg3d_kernel_map_transpose": (c_int32, [P, c_int32, c_int64, c_int64, P, P]),
    "cg3d_kernel_map_self": (c_int32, [P, c_int64, P, c_int32, P, P, c_int64, P, P]),
    "cg3d_spconv_fwd": (c_int32, [P, P, P, P, P, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32, P]),
    "cg3d_spconv_fwd_tiled": (c_int32, [P, P, P, P, c_int64, P, P, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32, P]),
    "cg3d_spconv_wgrad": (c_int32, [P, P, P, P, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32, P]),
    "cg3d_spconv_prep_weights_bf16": (c_int32, [P, P, c_int64, c_int32, c_int32, P]),
    "cg3d_pairs_ws_bytes": (c_int64, [c_int64]),
    "cg3d_pairs_count": (c_int32, [P, c_int32, c_int64, P, c_int32, P, P, P]),
    "cg3d_gather_rows": (c_int32, [P, P, P, c_int64, c_int32, P]),
    "cg3d_scatter_add_rows": (c_int32, [P, P, P, c_int64, c_int32, P]),
    "cg3d_pairs_fill": (c_int32, [P, c_int32, c_int64, P, P, P, P]),
    "cg3d_spconv_pairs_fwd": (c_int32, [P, P, P, P, P, c_int64, P, P, c_int64, c_int32, c_int32, c_int32, c_int32, P]),
    "cg3d_spconv_pairs_wgrad": (c_int32, [P, P, P, P, P, c_int64, P, c_int32, c_int32, c_int32, c_int32, P]),
    "cg3d_interp_map": (c_int32, [P, c_int64, c_int32, P, P, c_int64, P, P, P]),
    "cg3d_interp_fwd": (c_int32, [P, P, P, P, c_int64, c_int32, P]),
    "cg3d_interp_bwd": (c_int32, [P, P, P, P, c_int64, c_int32, P]),
    "cg3d_pool_map": (c_int32, [P, c_int64, c_int32, c_int32, P, P, c_int64, P, P]),
    "cg3d_scatter_mean_fwd": (c_int32, [P, P, c_int32, P, P, c_int64, c_int64, c_int32, P]),
    "cg3d_scatter_mean_bwd": (c_int32, [P, P, P, c_int32, P, c_int64, c_int64, c_int32, P]),
    "cg3d_to_bf16": (c_int32, [P, P, c_int64, P]),
    "cg3d_to_bf16_split": (c_int32, [P, P, c_int64, c_int32, P]),
    "cg3d_from_bf16": (c_int32, [P, P, c_int64, P]),
    "cg3d_spconv_prep_weights_split": (c_int32, [P, P, P, P, c_int32, c_int64, c_int32, c_int32, c_int32, P]),
    "cg3d_points_in_boxes": (c_int32, [P, c_int64, P, c_int32, P, P, P, P]),
    "cg3d_focal_loss_nblocks": (c_int32, [c_int64, c_int32]),
    "cg3d_focal_loss_fwd": (c_int32, [P, P, P, c_int64, c_int32, c_float, c_float, P, P]),
    "cg3d_focal_loss_bwd": (c_int32, [P, P, P, P, c_int64, c_int32, c_float, c_float, P, P]),
    "cg3d_spconv_prep_weights_bf16_multi": (c_int32, [P, P, P, P, c_int32, c_int64, c_int32, c_int32, P]),
    "cg3d_spconv_prep_weights_bf16_table": (c_int32, [P, c_int64, P]),
    "cg3d_spconv_prep_weights_frag": (c_int32, [P, P, P, P, c_int32, c_int64, c_int32, c_int32, P]),
    "cg3d_tile_row_order": (c_int32, [P, c_int32, c_int64, c_int32, P, P]),
    "cg3d_tile_plan_build": (c_int32, [P, c_int32, c_int64, P, c_int64, c_int32, c_int32, P, P, P, P, P, c_int64, P, P, P]),
    "cg3d_spconv_tile_lds_bytes": (c_int64, [c_int32]),
    "cg3d_spconv_tile_fwd": (c_int32, [P, P, P, P, P, P, P, c_int32, c_int32, P, c_int64, P, P, P, c_int64, c_int64, c_int32,
                                       c_int32, c_int32, c_int32, c_int32, P, P]),
    "cg3d_spconv_tile_grid": (c_int32, [c_int64, c_int32, c_int32]),
    "cg3d_linear_fwd": (c_int32, [P, P, P, P, c_int64, c_int32, c_int32, c_int32, P, P, P]),
    "cg3d_fcos_centerness": (c_int32, [P, P, P, c_int64, P, P, P, c_int32, P, P]),
    "cg3d_fcos_assign": (c_int32, [P, P, c_int64, P, P, c_int32, P, P, P, P]),
    "cg3d_pos_loss_nblocks": (c_int32, [c_int64]),
    "cg3d_pos_loss_fwd": (c_int32, [P, P, P, P, P, c_int32, P, P, P, P, c_int64, c_float, c_float, c_float, P, P]),
    "cg3d_pos_loss_bwd": (c_int32, [P, P, P, P, P, c_int32, P, P, P, P, c_int64, c_float, c_float, c_float, P, P, P, P]),
    "cg3d_smooth_l1_rows_fwd": (c_int32, [P, P, P, c_int64, c_int32, c_float, P, P]),
    "cg3d_smooth_l1_rows_bwd": (c_int32, [P, P, P, P, c_int64, c_int32, c_float, P, P]),
    "cg3d_grad_norm_clip": (c_int32, [P, P, c_int64, P, c_float, P, P, P, P]),
    "cg3d_adamw_step": (c_int32, [P, P, c_int64, P, P, c_float, c_float, c_float, c_float, c_float, c_float, c_float, P]),
    "cg3d_bn_sums": (c_int32, [P, P, c_int64, c_int32, c_int32, P, P]),
    "cg3d_bn_apply_sums": (c_int32, [P, P, P, c_int64, c_int32, c_int32, P, P, c_float, P, P, c_int32, P, P, P, P, P, P, P, c_float, P]),
    "cg3d_bn_apply": (c_int32, [P, P, P, c_int64, c_int32, P, P, c_float, P, P, c_int32, P, P, P]),
    "cg3d_bn_bwd_sums": (c_int32, [P, P, P, P, c_int64, c_int32, c_int32, P, P, c_float, c_int32, P, P]),
    "cg3d_bn_bwd_apply_sums": (c_int32, [P, P, P, P, c_int64, c_int32, c_int32, P, P, c_float, P, P, P, c_int32, c_int32, P, P, P, P, P, P]),
    "cg3d_bn_bwd_apply": (c_int32, [P, P, P, P, c_int64, c_int32, P, P, c_float, P, P, P, P, c_int32, c_int32, P, P, P, P]),
    "cg3d_boxes_overlap_bev": (c_int32, [P, c_int64, P, c_int64, P, P]),
    "cg3d_boxes_iou_bev": (c_int32, [P, c_int64, P, c_int64, P, P]),
    "cg3d_boxes_iou_bev_cpu": (c_int32, [P, c_int64, P, c_int64, P]),
    "cg3d_nms": (c_int32, [P, c_int64, c_float, c_int32, P, P, P, P]),
    "cg3d_nms_gpu_ws_bytes": (c_int64, [c_int64]),
    "cg3d_nms_gpu": (c_int32, [P, c_int64, P, c_float, P, P]),
    "cg3d_nms_normal_gpu": (c_int32, [P, c_int64, P, c_float, P, P]),
    "cg3d_nms_batched": (c_int32, [P, P, P, c_int32, c_int64, c_float, c_int32, P, P, P, P]),
    "cg3d_knn_ws_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    "cg3d_knn": (c_int32, [c_int32, c_int32, c_int32, c_int32, P, P, P, P, P, P]),
    "cg3d_ball_query": (c_int32, [c_int32, c_int32, c_int32, c_float, c_int32, P, P, P, P]),
    "cg3d_sort_vertices": (c_int32, [c_int32, c_int32, c_int32, P, P, P, P, P]),
    # include/cagroup3d_stages.h
    "cg3d_roi_match": (c_int32, [P, P, P, c_int32, c_int32, c_float, P, c_int32, c_int32, P, P, P, P]),
    "cg3d_roi_targets": (c_int32, [P, P, P, P, c_int32, c_int32, c_float, P, c_int32, c_int32, P, P, P, c_int32, c_int32, c_float,
                                   c_float, c_float, c_float, P, P, P, P, P, P, P, P, P, P, P]),
    "cg3d_roi_grid_coords": (c_int32, [P, c_int64, c_int32, c_int32, c_int32, c_float, c_float, c_float, c_int32, P, P]),
    "cg3d_roi_reg_loss_fwd": (c_int32, [P, P, P, P, c_int64, c_int32, c_float, c_float, P, P]),
    "cg3d_roi_reg_loss_bwd": (c_int32, [P, P, P, P, c_int64, c_int32, c_float, c_float, P, P, P, P]),
    "cg3d_class_nblk": (c_int32, [c_int64]),
    "cg3d_class_count": (c_int32, [P, c_int64, c_int32, P, P, P, P]),
    "cg3d_class_rows": (c_int32, [P, c_int64, c_int32, c_int32, P, P, P, P, P, c_int32, c_float, c_int32, P, c_int32, P, P, P, P]),
    "cg3d_gather_rows2": (c_int32, [P, P, c_int64, P, P, c_int64, c_int32, P]),
    "cg3d_scatter_add_rows2": (c_int32, [P, P, P, P, c_int64, c_int64, c_int32, P]),
    "cg3d_count_ids": (c_int32, [P, c_int64, c_int32, c_int32, c_int32, P, P]),
    "cg3d_count_sorted_ids": (c_int32, [P, c_int64, c_int32, c_int32, c_int32, P, P]),
    "cg3d_prop_keys": (c_int32, [P, P, c_int64, P, P]),
    "cg3d_prop_entries": (c_int32, [P, P, P, c_int32, c_int32, c_int32, P, c_int32, c_float, P, P, P]),
    "cg3d_prop_gather": (c_int32, [P, c_int64, P, P, P, c_int32, c_int32, P, P, c_int32, P, P, P, P, P]),
    "cg3d_rotated_iou3d_fwd": (c_int32, [P, P, c_int64, P, P]),
    "cg3d_rotated_iou3d_bwd": (c_int32, [P, P, c_int64, P, P, P]),
    "cg3d_instance_centers": (c_int32, [P, P, P, c_int32, c_int32, c_int32, P, c_int32, P, c_int32, P, P, P]),
    "cg3d_vote_targets": (c_int32, [P, P, P, c_int64, P, c_int32, P, c_int32, P, P, P]),
    "cg3d_pos_loss_yaw_nblocks": (c_int32, [c_int64]),
    "cg3d_pos_loss_yaw_fwd": (c_int32, [P, P, P, P, P, c_int32, P, P, P, P, c_int64, c_float, c_float, c_float, P, P]),
    "cg3d_pos_loss_yaw_bwd": (c_int32, [P, P, P, P, P, c_int32, P, P, P, P, c_int64, c_float, c_float, c_float, P, P, P, P]),
    "cg3d_head_outputs_fwd": (c_int32, [P, c_int32, P, c_int64, c_int32, P, P, c_int32, c_float, P, P, P, P]),
    "cg3d_head_outputs_bwd": (c_int32, [P, P, P, c_int32, P, c_int64, c_int32, P, c_int32, P, P, P]),
    # include/cagroup3d_program.h
    "cg3d_run_program": (c_int32, [P, c_int64, P, P]),
    "cg3d_run_program_lanes": (c_int32, [P, c_int64, P, c_int32, P]),
    "cg3d_program_schedule": (c_int32, [P, c_int64, P, P, P, c_int32, P, c_int64, P, P, P, P]),
    "cg3d_program_roles": (c_int32, [c_int32, P, P]),
    "cg3d_run_program_bound": (c_int32, [P, c_int64, P, P, c_int64, P, c_int64, P, c_int32, P]),
    "cg3d_host_bn_chunks": (c_int32, [P, c_int32, c_int64, c_int64, c_int64, P, c_int64, P, P, P, P, P]),
    "cg3d_host_segments": (c_int32, [P, c_int32, c_int32, c_int64, c_int32, P, c_int64, P, c_int64, P]),
    "cg3d_event_create": (c_int32, [P]),
    "cg3d_event_create_sync": (c_int32, [P]),
    "cg3d_event_destroy": (c_int32, [c_int64]),
    "cg3d_event_elapsed_ms": (c_int32, [c_int64, c_int64, P]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class CG3DError(RuntimeError):
    pass


_DEBUG_SYNC = open(os.environ["CG3D_DEBUG_SYNC"], "w") if os.environ.get("CG3D_DEBUG_SYNC") else None
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None) or (lambda: torch.cuda.current_device())


class Library:
    """A loaded shared object exporting the cg3d_* C-ABI."""

    def __init__(self, path):
        self.path = path
        self._dll = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(self._dll, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        self.is_device = bool(self._dll.cg3d_is_device_library())
        self.device_type = "cuda" if self.is_device else "cpu"
        # Which KERNELS the host side picks (tile plans + fragment-ordered weights, bf16 row copies, the step's weight arena,
        # launch programs in the bench precision): the device library's choice.  `device_path()` lets tests bind a second
        # handle of the CPU oracle that takes the same choices, so that the path the benchmark times -- not only its kernels one
        # call at a time -- is compared with the oracle's arithmetic (tests/test_timed_path_oracle.py).  Never set by the product.
        self.device_kernels = self.is_device

    def raw(self, name):
        return getattr(self._dll, name)

    def call(self, name, *args):
        if _DEBUG_SYNC and self.is_device:
            # dev aid (CG3D_DEBUG_SYNC=<file>): the entry point's name goes to the file BEFORE the call and the device is waited
            # for after it, so that after a GPU memory fault (which kills the process without a traceback) the last line names
            # the faulting call (a line without its " ok" never came back)
            torch.cuda.synchronize()
            _DEBUG_SYNC.write("%s %s" % (name, " ".join(str(getattr(a, "value", a)) for a in args)))
            _DEBUG_SYNC.flush()
        rc = getattr(self._dll, name)(*args)
        if rc != CG3D_OK:
            raise CG3DError("%s failed: %s" % (name, _ERRORS.get(rc, rc)))
        if _DEBUG_SYNC and self.is_device:
            torch.cuda.synchronize()
            _DEBUG_SYNC.write(" ok\n")

    def stream(self):
        """Raw handle of torch's current stream on the current device (hipStream_t)."""
        if self.is_device:
            # (the two C entry points directly: torch.cuda.current_device() is a Python function with a lazy-init check,
            # ~600 calls per training step)
            return _RAW_STREAM(_GET_DEVICE())
        return None

    def check(self, *tensors):
        """Every tensor must live where this library computes and be contiguous."""
        dev_is_cuda = self.is_device
        for t in tensors:
            if t is None:
                continue
            if t.is_cuda != dev_is_cuda:
                raise CG3DError(
                    "cagroup3d_amd op got a %s tensor but the bound library (%s) computes on %s; "
                    "there is no CPU fallback in the product path" % (t.device.type, self.path, self.device_type))
            if not t.is_contiguous():
                raise CG3DError("cagroup3d_amd ops need contiguous tensors")


def ptr(t):
    """Address of a tensor's first element as a plain int (None for a missing optional argument): every entry point has its
    argtypes declared, so ctypes converts -- without a c_void_p object per argument (~2 200 of them per training step)."""
    return None if t is None else t.data_ptr()


_active = None


def bind(path):
    return Library(path)


def device_path(lib):
    """Test aid: `lib` (the CPU oracle) flagged to follow the device library's kernel selection.  Returns the same handle."""
    lib.device_kernels = True
    return lib


def get():
    """The library the ops run on.  Default: the HIP library; raises if it has not been built."""
    global _active
    if _active is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise CG3DError(
                "libcagroup3d_hip.so not found at %s -- build it with `python cagroup3d_amd/csrc/build.py` "
                "(hipcc --offload-arch=gfx950).  The product path has no CPU fallback." % HIP_LIB_PATH)
        _active = Library(HIP_LIB_PATH)
        if not _active.is_device:
            _active = None
            raise CG3DError("%s is not a device library -- the product path has no CPU fallback" % HIP_LIB_PATH)
    return _active


# Callbacks run whenever the active library CHANGES (me.py registers the reset of the step's weight plan: its arena and the
# recorded weights belong to one library's memory -- a plan that still listed device weights once handed the oracle a table of
# device addresses, which the CPU then read through the PCIe aperture for minutes: tests/test_timed_path_oracle.py, round 6).
on_switch = []


def _switched(old, new):
    if old is not new:
        for fn in on_switch:
            fn()


class use_library:
    """Context manager that points the host-side engine at another C-ABI implementation.

    Used only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg with the CPU
    oracle -- as the checker, never as the product."""

    def __init__(self, lib):
        self.lib = lib

    def __enter__(self):
        global _active
        self.prev = _active
        _active = self.lib
        _switched(self.prev, _active)
        return self.lib

    def __exit__(self, *exc):
        global _active
        was, _active = _active, self.prev
        _switched(was, _active)
        return False
