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
SOURCES = [os.path.join(CSRC, "cavoid_capi.hip"), os.path.join(CSRC, "cavoid_multistep.hip"), os.path.join(CSRC, "cavoid_rvo.hip"),
           os.path.join(CSRC, "cavoid_relay.hip"), os.path.join(CSRC, "cavoid_quad.hip"), os.path.join(CSRC, "cavoid_rollout_capi.hip"),
           os.path.join(CSRC, "cavoid_policy_capi.hip"), os.path.join(CSRC, "cavoid_comm_capi.hip"), os.path.join(CSRC, "cavoid_actor.hip"),
           os.path.join(CSRC, "cavoid_actor_rvo.hip"), os.path.join(CSRC, "cavoid_actor_frozen.hip")]
HEADERS = {
    "cavoid_capi.hip": ["cavoid_kernels.hpp", "cavoid_launch.hpp", "cavoid_host.hpp"],
    "cavoid_multistep.hip": ["cavoid_kernels.hpp", "cavoid_launch.hpp", "cavoid_host.hpp"],
    "cavoid_rvo.hip": ["cavoid_kernels.hpp", "cavoid_launch.hpp", "cavoid_host.hpp"],
    "cavoid_relay.hip": ["cavoid_kernels.hpp", "cavoid_relay.hpp", "cavoid_launch.hpp", "cavoid_host.hpp"],
    "cavoid_quad.hip": ["cavoid_kernels.hpp", "cavoid_quad.hpp", "cavoid_launch.hpp", "cavoid_host.hpp"],
    "cavoid_rollout_capi.hip": ["cavoid_rollout.hpp", "cavoid_rollout_host.hpp", "cavoid_host.hpp"],
    "cavoid_policy_capi.hip": ["cavoid_policy.hpp", "cavoid_policy_split.hpp", "cavoid_policy_split8.hpp", "cavoid_policy_host.hpp", "cavoid_host.hpp"],
    "cavoid_actor.hip": ["cavoid_actor.hpp", "cavoid_actor_host.hpp", "cavoid_kernels.hpp", "cavoid_quad.hpp", "cavoid_launch.hpp", "cavoid_policy.hpp",
                         "cavoid_policy_split.hpp", "cavoid_policy_host.hpp", "cavoid_rollout.hpp", "cavoid_rollout_host.hpp", "cavoid_host.hpp"],
    "cavoid_actor_rvo.hip": ["cavoid_actor.hpp", "cavoid_actor_host.hpp", "cavoid_kernels.hpp", "cavoid_quad.hpp", "cavoid_launch.hpp", "cavoid_policy.hpp",
                             "cavoid_policy_split.hpp", "cavoid_rollout.hpp", "cavoid_host.hpp"],
    "cavoid_actor_frozen.hip": ["cavoid_actor.hpp", "cavoid_actor_host.hpp", "cavoid_kernels.hpp", "cavoid_quad.hpp", "cavoid_launch.hpp", "cavoid_policy.hpp",
                                "cavoid_policy_split.hpp", "cavoid_rollout.hpp", "cavoid_host.hpp"],
    "cavoid_comm_capi.hip": ["cavoid_host.hpp"],
}
# per-file extra flags.  The multi-step env kernels run their step loop inside the launch; MachineLICM would hoist every
# constant materialisation of the body (float64 polynomial coefficients, config scalars) out of that loop into
# registers live across it: 128 VGPRs + 276 B/lane of scratch instead of 128 VGPRs + 12 B (N = 4).
EXTRA_FLAGS = {"cavoid_multistep.hip": ["-mllvm", "-disable-machine-licm"], "cavoid_rvo.hip": ["-mllvm", "-disable-machine-licm"],
               "cavoid_relay.hip": ["-mllvm", "-disable-machine-licm"],
               # the fused actor kernel runs policy + env step + bookkeeping inside ONE step loop: same reason (without it the
               # GEMM loops' fragment addresses are hoisted across the loop: 256 VGPRs + 232 B/lane of scratch instead of 243 + 0)
               "cavoid_actor.hip": ["-mllvm", "-disable-machine-licm"], "cavoid_actor_rvo.hip": ["-mllvm", "-disable-machine-licm"],
               "cavoid_actor_frozen.hip": ["-mllvm", "-disable-machine-licm"]}
STAMP_PATH = os.path.join(PKG_DIR, "libcavoid_hip.so.stamp")
DEPS = SOURCES + [os.path.join(CSRC, h) for hs in HEADERS.values() for h in hs] + [os.path.join(ROOT, "include", "cavoid.h")]
OBJ_DIR = os.path.join(PKG_DIR, "build")
# Development variants (phase-trace build, fault-injection builds) are test / tooling artefacts, never the product: they are linked
# into tests/_variants/, NOT beside libcavoid_hip.so, so that the package directory holds exactly one library and a process that maps
# the product maps nothing else (selected with CAVOID_LIB=<path> by the tests and tools that want one, always in a child process).
VARIANT_DIR = os.path.join(ROOT, "tests", "_variants")


def variant_path(name: str) -> str:
    os.makedirs(VARIANT_DIR, exist_ok=True)
    return os.path.join(VARIANT_DIR, "libcavoid_hip_%s.so" % name)


# -ffp-contract=off: the reference env is unfused NumPy float64; keep mul/add separate so that the
# only numerical difference from the CPU oracle is the transcendental library.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built on this machine")


def source_digest() -> str:
    """sha256 over the flags and the contents of every source the library is built from (mtimes do not survive a
    copy of the tree to another box; contents do)."""
    import hashlib
    h = hashlib.sha256(repr((FLAGS[:5], sorted(EXTRA_FLAGS.items()))).encode())
    for d in sorted(DEPS):
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def is_stale() -> bool:
    """The in-tree binary does not match the in-tree sources (or is missing)."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(STAMP_PATH):
        return True
    with open(STAMP_PATH) as f:
        return f.read().strip() != source_digest()


def _compile_objects(extra_flags, tag: str, force: bool, verbose: bool):
    """One object per translation unit, rebuilt only when it (or a header it includes) changed; the stale ones are
    compiled concurrently (each hipcc is single-threaded and the env kernels take ~1.5 min)."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs, jobs = [], []
    for src in SOURCES:
        name = os.path.basename(src)
        obj = os.path.join(OBJ_DIR, name.replace(".hip", tag + ".o"))
        deps = [src, os.path.join(ROOT, "include", "cavoid.h")] + [os.path.join(CSRC, h) for h in HEADERS[name]]
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps):
            cmd = [hipcc()] + FLAGS + EXTRA_FLAGS.get(name, []) + list(extra_flags) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append(cmd)
        objs.append(obj)
    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as pool:
            list(pool.map(subprocess.check_call, jobs))
    return objs


def _link(objs, out: str, verbose: bool) -> str:
    # no -lrccl: cavoid_comm_capi.hip binds RCCL with dlopen at the first multi-rank call (single-GPU users and host-only
    # tests load without it; inside a PyTorch process it takes the librccl PyTorch has already mapped)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    if force or is_stale():
        digest = source_digest()             # of what is about to be compiled (a source edited meanwhile must read as stale)
        if os.path.exists(STAMP_PATH):
            os.remove(STAMP_PATH)
        _link(_compile_objects([], "", force, verbose), LIB_PATH, verbose)
        with open(STAMP_PATH, "w") as f:
            f.write(digest + "\n")
    return LIB_PATH


def build_trace(verbose: bool = False) -> str:
    """Development variant with in-kernel phase time stamps (tools/trace_step.py); never loaded by
    the product (select it with CAVOID_LIB=<path>)."""
    out = variant_path("trace")
    return _link(_compile_objects(["-DCAVOID_TRACE"], ".trace", False, verbose), out, verbose)


def build_fault(verbose: bool = False) -> str:
    """Development variant with a broken relay hand-over (-DCAVOID_FAULT_RELAY: the pair-pass wavefront of tile 0 leaves at step
    5), for tests/test_gpu_relay_fault.py: only cavoid_relay.hip is recompiled, the other objects are the product's."""
    objs = _compile_objects([], "", False, verbose)
    src = os.path.join(CSRC, "cavoid_relay.hip")
    obj = os.path.join(OBJ_DIR, "cavoid_relay.fault.o")
    cmd = [hipcc()] + FLAGS + EXTRA_FLAGS["cavoid_relay.hip"] + ["-DCAVOID_FAULT_RELAY", "-DCAVOID_DEV_ONLY_N", "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    objs = [obj if o.endswith("cavoid_relay.o") else o for o in objs]
    return _link(objs, variant_path("fault"), verbose)


def build_ulp_fault(kind: int, verbose: bool = False) -> str:
    """Development variants with ONE ulp-scale arithmetic fault in the env step (-DCAVOID_DEV_ULP_FAULT=kind: 1 = a float32 product in
    the pair pass's distance, 2 = the position update contracted into fused multiply-adds, 3 = the sort key's centimetre bucket through
    float32, 4 = a float32 product in the ORCA policy's squared distance), for tests/test_gpu_tie_classifier.py: what the parity harness and its tie classifier say about faults of the size the
    classifier excuses.  Only the env kernels' translation units are recompiled (dev-only N = 4, 10); never loaded by the product."""
    from concurrent.futures import ThreadPoolExecutor
    objs = _compile_objects([], "", False, verbose)
    jobs, swap = [], {}
    for name in ("cavoid_capi.hip", "cavoid_multistep.hip", "cavoid_rvo.hip", "cavoid_relay.hip"):
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJ_DIR, name.replace(".hip", ".ulp%d.o" % kind))
        deps = [src, os.path.join(ROOT, "include", "cavoid.h")] + [os.path.join(CSRC, h) for h in HEADERS[name]]
        if not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps):
            jobs.append([hipcc()] + FLAGS + EXTRA_FLAGS.get(name, []) + ["-DCAVOID_DEV_ULP_FAULT=%d" % kind, "-DCAVOID_DEV_ONLY_N", "-c", src, "-o", obj])
        swap[os.path.join(OBJ_DIR, name.replace(".hip", ".o"))] = obj
    if jobs:
        if verbose:
            for j in jobs:
                print(" ".join(j), flush=True)
        with ThreadPoolExecutor(max_workers=len(jobs)) as pool:
            list(pool.map(subprocess.check_call, jobs))
    return _link([swap.get(o, o) for o in objs], variant_path("ulp%d" % kind), verbose)


if __name__ == "__main__":
    import sys
    if "--trace" in sys.argv:
        print(build_trace(verbose=True))
    elif "--fault" in sys.argv:
        print(build_fault(verbose=True))
    elif "--ulp-faults" in sys.argv:
        for kind in (1, 2, 3, 4):
            print(build_ulp_fault(kind, verbose=True))
    else:
        print(build(force=True, verbose=True))
