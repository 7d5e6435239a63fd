"""Build the gfx950 shared library ``libcavoid_hip.so`` in-tree with hipcc (no torch extension,
no JIT cache: the built .so travels with the source tree)."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libcavoid_hip.so")
SOURCES = [os.path.join(CSRC, "cavoid_capi.hip")]
DEPS = SOURCES + [os.path.join(CSRC, "cavoid_kernels.hpp"), os.path.join(CSRC, "cavoid_rollout.hpp"),
               os.path.join(ROOT, "include", "cavoid.h")]
# -ffp-contract=off: the reference env is unfused NumPy float64; keep mul/add separate so that the
# only numerical difference from the CPU oracle is the transcendental library.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built on this machine")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or is_stale():
        cmd = [hipcc()] + FLAGS + SOURCES + ["-o", LIB_PATH]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


def build_trace(verbose: bool = False) -> str:
    """Development variant with in-kernel phase time stamps (tools/trace_step.py); never loaded by
    the product (select it with CAVOID_LIB=<path>)."""
    out = os.path.join(PKG_DIR, "libcavoid_hip_trace.so")
    cmd = [hipcc()] + FLAGS + ["-DCAVOID_TRACE"] + SOURCES + ["-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    import sys
    print(build_trace(verbose=True) if "--trace" in sys.argv else build(force=True, verbose=True))
