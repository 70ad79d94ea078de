"""Builds lib/libmvhmr_unproject.so with hipcc for gfx950 (in-tree, so the .so travels with the checkout)."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib", "libmvhmr_unproject.so")


def build(force=False, jobs=None, verbose=False):
    jobs = jobs or min(4, os.cpu_count() or 1)
    cmd = ["make", "-C", CSRC, "-j%d" % jobs]
    if force:
        cmd.append("-B")
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("hipcc build of libmvhmr_unproject.so failed (see output above)")
    return LIB


EXT_DIR = os.path.join(PKG, "lib_ext")
EXT = os.path.join(EXT_DIR, "mvhmr_torch_ext.so")


def build_ext(force=False, verbose=False):
    """The PyTorch-ROCm C++ extension over the C ABI (csrc_ext/mvhmr_torch_ext.cpp): host code only, built in-tree with
    torch.utils.cpp_extension (ninja + g++), linked against lib/libmvhmr_unproject.so through an $ORIGIN-relative rpath."""
    src = os.path.join(PKG, "csrc_ext", "mvhmr_torch_ext.cpp")
    hdr = os.path.join(os.path.dirname(PKG), "include", "mvhmr_unproject.h")
    if not force and os.path.exists(EXT) and os.path.getmtime(EXT) >= max(os.path.getmtime(src), os.path.getmtime(hdr)):
        return EXT
    if not os.path.exists(LIB):
        build()
    import torch
    from torch.utils import cpp_extension
    os.makedirs(EXT_DIR, exist_ok=True)
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cpp_extension.load(
        name="mvhmr_torch_ext", sources=[src], build_directory=EXT_DIR, is_python_module=False, with_cuda=False, verbose=verbose,
        extra_include_paths=[os.path.join(os.path.dirname(PKG), "include"), "/opt/rocm/include"],
        extra_cflags=["-O2", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"],
        extra_ldflags=["-L" + os.path.join(PKG, "lib"), "-lmvhmr_unproject", "-Wl,-rpath,\\$$ORIGIN/../lib",
                       "-L" + torch_lib, "-lc10_hip", "-ltorch_hip"])
    assert os.path.exists(EXT), EXT
    return EXT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_ext(force="--force" in sys.argv, verbose=True))
