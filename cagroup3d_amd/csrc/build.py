"""Build libcagroup3d_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake: explicit commands.

    python cagroup3d_amd/csrc/build.py [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "libcagroup3d_hip.so")
ARCH = "gfx950"
# per-file flags: geometry / integer files are built with FMA contraction OFF so that their fp32
# results are bit-identical to the CPU oracle (tests/ compare keep masks and indices exactly).
SOURCES = {
    "coords.hip": ["-ffp-contract=off"],
    "spconv.hip": ["-munsafe-fp-atomics"],
    "spconv_tile.hip": ["-munsafe-fp-atomics"],
    "spconv_tile2.hip": ["-munsafe-fp-atomics"],
    "linear.hip": ["-munsafe-fp-atomics"],
    "gather_scatter.hip": ["-munsafe-fp-atomics"],
    "iou3d_nms.hip": ["-ffp-contract=off"],
    "stages.hip": ["-ffp-contract=off", "-munsafe-fp-atomics"],     # stage-level fused ops of the heads (RoI matching / targets / grid, class rows, proposals)
    "knn.hip": ["-ffp-contract=off"],
    "sort_vertices.hip": ["-ffp-contract=off"],
    "bn_act.hip": [],
    "loss.hip": [],
    "program.hip": [],                       # cg3d_run_program: the launch-table executor (include/cagroup3d_program.h)
    "optim.hip": ["-ffp-contract=off"],      # same rounding as the oracle's plain C (and torch's kernel: no contraction across statements)
}
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("CG3D_HIPCC_EXTRA", "").split()
# (CG3D_HIPCC_EXTRA: dev builds on the GPU box, e.g. -DCG3D_TILE_TRACE; the shipped library is built without it)


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    hdrs = [os.path.join(HERE, "cg3d_common.h"), os.path.join(HERE, "dg_geom.h"), os.path.join(HERE, "..", "..", "include", "cagroup3d_hip.h"),
            os.path.join(HERE, "..", "..", "include", "cagroup3d_program.h"), os.path.join(HERE, "..", "..", "include", "cagroup3d_stages.h")]
    objs = []
    for src, extra in SOURCES.items():
        s = os.path.join(HERE, src)
        o = os.path.join(HERE, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + COMMON + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
