"""Builds libeepacc.so (hand-written HIP for gfx950 + C-ABI) in-tree with hipcc.

Each source is compiled to an object under csrc/_obj/ (in parallel), then linked.  The flag string of the
build is embedded in the library (eepacc_build_flags()) and written beside it (libeepacc.flags): a library
built with other flags (e.g. the -DEEPACC_AB_TIMING / -DEEPACC_DEBUG_STATUS instrumented variants, which
change the meaning of the iteration and status outputs) is stale and gets rebuilt."""
from __future__ import annotations

import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libeepacc.so")
FLAGFILE = os.path.join(HERE, "libeepacc.flags")
SOURCES = ["eepacc_kernels.hip", "eepacc_qp_dense.hip", "eepacc_fb.hip", "eepacc_fbs.hip", "eepacc_capi.cpp", "eepacc_casadi_c.cpp", "eepacc_nlp.hip", "eepacc_nlp_tables.cpp"]
HEADERS = ["eepacc_device.h", "eepacc_qp_dense.h", "eepacc_fb.h", "eepacc_stage.h", "eepacc_wave.h", "eepacc_fbs.h", "eepacc_ab_impl.inc", "eepacc_nlp_solve.inc", os.path.join("..", "..", "include", "eepacc.h"),
           os.path.join("..", "..", "include", "eepacc_casadi_c.h"), os.path.join("..", "..", "include", "eepacc_nlp.h")]
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def extra_flags() -> str:
    return " ".join(os.environ.get("EEPACC_EXTRA_FLAGS", "").split())


def built_flags() -> str | None:
    try:
        with open(FLAGFILE) as f:
            return f.read().strip()
    except OSError:
        return None


def is_stale() -> bool:
    if not os.path.exists(LIB) or built_flags() != extra_flags():
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS
               if os.path.exists(os.path.join(CSRC, f)))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = extra_flags()
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    newest_hdr = max(os.path.getmtime(os.path.join(CSRC, f)) for f in HEADERS if os.path.exists(os.path.join(CSRC, f)))
    same_flags = built_flags() == flags

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        path = os.path.join(CSRC, src)
        if (not force and same_flags and os.path.exists(obj)
                and os.path.getmtime(obj) > max(os.path.getmtime(path), newest_hdr)):
            return obj
        cmd = [hipcc] + BASE_FLAGS + ["-x", "hip", "-c", path, "-o", obj,
                                      '-DEEPACC_BUILD_FLAGS="%s"' % flags.replace('"', "'")] + flags.split()
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(FLAGFILE, "w") as f:
        f.write(flags + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
