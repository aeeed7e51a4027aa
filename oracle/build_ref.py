"""Build oracle/_ref/iou3d_cpu_ref*.so from the REFERENCE's own CPU BEV-IoU source, where it lies.

    /root/reference/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp   (boxes_iou_bev_cpu, :232-252)

The file includes <cuda.h> / <cuda_runtime_api.h>; this image has ROCm, so the source is passed through
torch's own hipify (torch.utils.hipify, part of the installed PyTorch-ROCm toolchain -- the same step
torch.utils.cpp_extension applies to every extension on ROCm) into oracle/_ref/, then compiled with g++
against the torch headers together with a 10-line pybind11 binding written here (the reference's binding
file iou3d_nms_api.cpp pulls in the CUDA kernels as well).  Nothing is copied into the repository:
oracle/_ref/ is git-ignored.  Only possible where /root/reference exists; tests skip the reference
cross-check when the built module is absent.
"""
import os
import shutil
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
SRC = "/root/reference/pcdet/ops/iou3d_nms/src"
BINDING = r'''
#include <torch/extension.h>
#include "iou3d_cpu.h"
PYBIND11_MODULE(iou3d_cpu_ref, m) {
    m.def("boxes_iou_bev_cpu", &boxes_iou_bev_cpu, "reference pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:232");
}
'''


def main():
    if not os.path.isdir(SRC):
        print("reference sources not present; skipping oracle/_ref")
        return 0
    import torch
    from torch.utils import cpp_extension
    from torch.utils.hipify import hipify_python
    os.makedirs(OUT, exist_ok=True)
    work = os.path.join(OUT, "build")
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    for f in ("iou3d_cpu.cpp", "iou3d_cpu.h"):
        shutil.copy(os.path.join(SRC, f), work)           # scratch copy inside the ignored _ref/ only
    hipify_python.hipify(project_directory=work, output_directory=work, includes=[os.path.join(work, "*")],
                         extra_files=[os.path.join(work, "iou3d_cpu.cpp"), os.path.join(work, "iou3d_cpu.h")],
                         show_detailed=False,
                         is_pytorch_extension=True, hipify_extra_files_only=True)
    cpp = os.path.join(work, "iou3d_cpu.cpp")
    hip_cpp = os.path.join(work, "iou3d_cpu_hip.cpp")
    src = hip_cpp if os.path.exists(hip_cpp) else cpp
    hdr = "iou3d_cpu_hip.h" if os.path.exists(os.path.join(work, "iou3d_cpu_hip.h")) else "iou3d_cpu.h"
    with open(os.path.join(work, "binding.cpp"), "w") as f:
        f.write(BINDING.replace("iou3d_cpu.h", hdr))
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    so = os.path.join(OUT, "iou3d_cpu_ref" + ext)
    inc = cpp_extension.include_paths(device_type="cuda") if "device_type" in cpp_extension.include_paths.__code__.co_varnames \
        else cpp_extension.include_paths(cuda=True)
    cmd = ["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-DTORCH_EXTENSION_NAME=iou3d_cpu_ref",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch.compiled_with_cxx11_abi()), "-I" + work,
           "-I" + sysconfig.get_paths()["include"], "-I/opt/rocm/include"]
    cmd += ["-I" + p for p in inc]
    cmd += [src, os.path.join(work, "binding.cpp"), "-o", so,
            "-L" + os.path.join(os.path.dirname(torch.__file__), "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python",
            "-Wl,-rpath," + os.path.join(os.path.dirname(torch.__file__), "lib")]
    print(" ".join(cmd))
    subprocess.check_call(cmd)
    shutil.rmtree(work, ignore_errors=True)               # keep only the binary
    print("built", so)
    return 0


if __name__ == "__main__":
    sys.exit(main())
