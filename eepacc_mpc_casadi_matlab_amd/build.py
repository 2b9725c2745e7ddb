"""Builds libeepacc.so (hand-written HIP for gfx950 + C-ABI) in-tree with hipcc."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libeepacc.so")
SOURCES = ["eepacc_kernels.hip", "eepacc_qp_dense.hip", "eepacc_fb.hip", "eepacc_capi.cpp"]
HEADERS = ["eepacc_device.h", "eepacc_qp_dense.h", "eepacc_fb.h", "eepacc_stage.h", "eepacc_ab_impl.inc", os.path.join("..", "..", "include", "eepacc.h")]


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip"]
    cmd += os.environ.get("EEPACC_EXTRA_FLAGS", "").split()
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
