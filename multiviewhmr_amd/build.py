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


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
