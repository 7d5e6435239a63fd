"""RCCL behind the C ABI, EXECUTED: `cavoid_comm_create_ex(..., CAVOID_COMM_FORCE_RCCL)` makes a ONE-rank communicator a real RCCL
communicator (`ncclCommInitRank(nranks = 1)`), so that a 1-GPU box runs the `dlopen` / `dlsym` binding, `ncclGetUniqueId`,
`ncclAllGather` and the grouped `ncclSend` / `ncclRecv` of `cavoid_gatherv_begin` on the communicator's own stream, through the same
double-buffered event protocol the multi-rank hand-over uses.  (RCCL refuses two ranks on one device, so one rank is all a 1-GPU
box can run; what crosses a link first runs on the driver's node.)  Everything a forced communicator delivers must be bitwise what the
device-copy path delivers.  Reference hand-over: the mp.Queue of /root/reference/ga3c/GA3C/ProcessAgent.py:221,238."""
import ctypes as C
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _lib():
    from rl_collision_avoidance_amd import _lib
    return _lib, _lib.lib()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _info(lib, h):
    vals = [C.c_int32() for _ in range(4)]
    assert lib.cavoid_comm_info(h, *[C.byref(v) for v in vals]) == 0
    return [v.value for v in vals]


def test_unique_id_comes_from_rccl():
    """the binding itself: librccl is found (the copy PyTorch has mapped), ncclGetUniqueId answers, ids are fresh"""
    _, lib = _lib()
    a, b = (C.c_ubyte * 128)(), (C.c_ubyte * 128)()
    assert lib.cavoid_comm_unique_id(a) == 0 and lib.cavoid_comm_unique_id(b) == 0
    assert any(a) and any(b) and bytes(a) != bytes(b)
    assert lib.cavoid_comm_unique_id(None) == -1


def test_forced_one_rank_communicator_is_an_rccl_communicator():
    m, lib = _lib()
    plain, forced, by_env = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.cavoid_comm_create(None, 1, 0, 0, C.byref(plain)) == 0
    assert _info(lib, plain) == [1, 0, 0, 0]
    ident = (C.c_ubyte * 128)()
    assert lib.cavoid_comm_unique_id(ident) == 0
    assert lib.cavoid_comm_create_ex(ident, 1, 0, 0, m.COMM_FORCE_RCCL, C.byref(forced)) == 0, lib.cavoid_last_comm_error()
    n, r, uses, ver = _info(lib, forced)
    assert (n, r, uses) == (1, 0, 1) and ver >= 20000, ver            # ncclGetVersion: 2.x.y as 2xxyy
    os.environ["CAVOID_COMM_FORCE_RCCL"] = "1"                           # the environment switch of plain cavoid_comm_create
    try:
        assert lib.cavoid_comm_create(None, 1, 0, 0, C.byref(by_env)) == 0          # (no id given: one is made internally)
    finally:
        del os.environ["CAVOID_COMM_FORCE_RCCL"]
    assert _info(lib, by_env)[2] == 1
    bad = C.c_void_p()
    assert lib.cavoid_comm_create_ex(None, 1, 0, 0, 2, C.byref(bad)) == -1 and not bad.value     # unknown flag bit
    for h in (plain, forced, by_env):
        lib.cavoid_comm_destroy(h)


@pytest.mark.parametrize("floats", [0, 1, 12345, 8192 * 4 * 29])
def test_forced_gathers_equal_the_copy_path_bitwise(floats):
    """cavoid_gather_begin (ncclAllGather) and cavoid_gatherv_begin to every rank / to the root (grouped ncclSend + ncclRecv to
    itself) against the device-copy path: both slots, send != recv, sizes incl. empty, odd and the configs[1] record block."""
    m, lib = _lib()
    plain, forced = C.c_void_p(), C.c_void_p()
    assert lib.cavoid_comm_create_ex(None, 1, 0, 0, 0, C.byref(plain)) == 0
    assert lib.cavoid_comm_create_ex(None, 1, 0, 0, m.COMM_FORCE_RCCL, C.byref(forced)) == 0
    g = torch.Generator(device="cuda").manual_seed(floats + 1)
    counts = (C.c_int64 * 1)(floats)
    for slot in (0, 1, 0):
        send = torch.randn(max(floats, 1), generator=g, device="cuda")[:floats].contiguous() if floats else torch.zeros(1, device="cuda")
        outs = {}
        for name, h in (("plain", plain), ("forced", forced)):
            a, b, c = (torch.full((max(floats, 1),), -7.0, device="cuda") for _ in range(3))
            p = lambda t: C.c_void_p(t.data_ptr())
            assert lib.cavoid_gather_begin(h, slot, p(send), p(a), floats, _stream()) == 0, lib.cavoid_last_comm_error()
            assert lib.cavoid_gather_wait(h, slot, _stream()) == 0
            assert lib.cavoid_gatherv_begin(h, slot, p(send), p(b), counts, -1, _stream()) == 0, lib.cavoid_last_comm_error()
            assert lib.cavoid_gather_wait(h, slot, _stream()) == 0
            assert lib.cavoid_gatherv_begin(h, slot, p(send), p(c), counts, 0, _stream()) == 0, lib.cavoid_last_comm_error()
            assert lib.cavoid_gather_wait(h, slot, _stream()) == 0
            torch.cuda.synchronize()
            outs[name] = (a, b, c)
        for x, y in zip(outs["plain"], outs["forced"]):
            assert torch.equal(x, y)
            if floats:
                assert torch.equal(x[:floats], send)
    lib.cavoid_comm_destroy(plain)
    lib.cavoid_comm_destroy(forced)


@pytest.mark.parametrize("steps_per_launch,root", [(1, -1), (8, -1), (1, 0), (8, 0)])
def test_forced_rccl_carries_the_step_and_gather_protocol(steps_per_launch, root):
    """`ShardedEnv.step_and_gather` through a forced RCCL communicator -- one step and K-step blocks, to every rank (ncclAllGather)
    and to the trainer rank (point-to-point group) -- hands through, bit for bit, what an unsharded env produces; gather t is
    consumed while gather t+1 is in flight (the overlap protocol)."""
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    from rl_collision_avoidance_amd.config import EnvConfig
    from rl_collision_avoidance_amd.sharding import ShardedEnv
    W, N, seed, K = 2048, 4, 21, steps_per_launch
    launches = 12 if K > 1 else 80                                     # (>= 80 steps: past the first restarts)
    sh = ShardedEnv(W, EnvConfig(), device=torch.device("cuda", 0), seed=seed, force_rccl=True, gen_min_agents=2, gen_pool_size=512)
    ref = BatchedCollisionAvoidanceEnv(W, EnvConfig(), device="cuda:0", seed=seed, gen_min_agents=2, gen_pool_size=512)
    sh.reset(); ref.reset()
    g = torch.Generator(device="cuda").manual_seed(3)
    acts = torch.randint(0, 11, (launches, K, W, N), generator=g, device="cuda", dtype=torch.int32)
    slots = ref.new_step_slots(K, packed=True) if K > 1 else None
    pk = ref.new_packed()
    prev = None
    for l in range(launches):
        slot = sh.step_and_gather(acts[l] if K > 1 else acts[l, 0], root=root)
        if K > 1:
            ref.step_autoreset_packed(acts[l], slots)
            want = slots.packed.clone()
        else:
            ref.step_autoreset_packed(acts[l, 0], pk)
            want = pk.clone()
        if prev is not None:
            assert torch.equal(sh.gathered(prev[0]).reshape(prev[1].shape), prev[1]), l - 1
        prev = (slot, want)
    assert torch.equal(sh.gathered(prev[0]).reshape(prev[1].shape), prev[1])
    assert sh._native.uses_rccl and sh._native.rccl_version >= 20000
    assert ("ncclAllGather" in sh.gather_form) if root < 0 else ("point-to-point RCCL group" in sh.gather_form)
    assert ref.episode.max().item() >= 1                                   # restarts happened inside the compared steps
    sh.close(); ref.close()


_BAD_ID = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
from rl_collision_avoidance_amd import _lib
lib = _lib.lib()
h = C.c_void_p()
ident = (C.c_ubyte * 128)(*([%d] * 128))
rc = lib.cavoid_comm_create_ex(ident, 1, 0, 0, _lib.COMM_FORCE_RCCL, C.byref(h))
print("\nrc", rc, "nccl", lib.cavoid_last_comm_error(), "handle", bool(h.value), flush=True)
"""


@pytest.mark.parametrize("fill", [0, 255])
def test_a_bad_unique_id_is_an_error_not_a_hang(fill):
    """an id that no ncclGetUniqueId made (all zeros / all ones: no bootstrap root behind it) must come back as CAVOID_ECOMM with the
    raw ncclResult_t kept, promptly, and leave no handle -- in a child process with a deadline so that a hang is a failure"""
    out = subprocess.run([sys.executable, "-c", _BAD_ID % (ROOT, fill)], timeout=180, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, env=dict(os.environ, NCCL_DEBUG="WARN"))
    assert out.returncode == 0, out.stderr[-2000:]
    import re
    m = re.search(r"rc (-?\d+) nccl (-?\d+) handle (True|False)", out.stdout)      # (RCCL's own WARN lines share the stream)
    assert m, "no verdict line:\n" + out.stdout[-1500:] + out.stderr[-1500:]
    assert int(m.group(1)) == -6 and int(m.group(2)) != 0 and m.group(3) == "False", out.stdout[-1500:] + out.stderr[-1500:]
