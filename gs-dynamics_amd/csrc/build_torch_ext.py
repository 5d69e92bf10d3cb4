"""Build the torch C++ layer (gsr_torch.cpp) in-tree: gs-dynamics_amd/diff_gaussian_rasterization/_C.so.

    python gs-dynamics_amd/csrc/build_torch_ext.py

Host compiler only (no device code: everything behind the C-ABI of libgsr_hip.so, which must have been built first); the result
links against ../csrc/libgsr_hip.so through an $ORIGIN-relative rpath, so the pair travels together with the repository snapshot.
"""
import os
import shutil
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.normpath(os.path.join(HERE, "..", "diff_gaussian_rasterization"))
OUT = os.path.join(PKG, "_C.so")
SRC = os.path.join(HERE, "gsr_torch.cpp")


def up_to_date():
    deps = [SRC, os.path.join(HERE, "..", "..", "include", "gsr.h"), os.path.join(HERE, "libgsr_hip.so")]
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps)


def build(verbose=False):
    if up_to_date():
        return OUT
    import torch
    from torch.utils import cpp_extension as ce
    inc = ce.include_paths() + [sysconfig.get_paths()["include"], "/opt/rocm/include"]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=_C", "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           "-Wno-deprecated-declarations"]
    cmd += [f"-I{p}" for p in inc]
    cmd += [SRC, "-o", OUT, f"-L{libdir}", "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-ltorch_python",
            f"-L{HERE}", "-lgsr_hip", "-Wl,-rpath,$ORIGIN/../csrc", f"-Wl,-rpath,{libdir}"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
