"""GPU tests of the round-2 launch forms of the env step (through the C ABI):

* packed outputs -- the kernel writes one (obs | reward | done) record per agent -- must equal, bit for bit, the plain
  outputs packed by `sharding.pack_step_outputs` (the three-launch torch pack it replaces);
* multi-step launches (`cavoid_step_autoreset_n`: the world state stays in registers between steps) must equal the same
  steps launched one by one, in every form of the loop (env_relay_kernel: the step cut into roles on several wavefronts per
  tile; env_pipe_kernel: two wavefronts per tile; one wavefront per tile with register prefetch of the next pool record /
  on-demand gather) and for every wavefront geometry (worlds per wavefront);
* BASELINE configs[2] at full size on ONE GPU: 8 shard envs of 8192 worlds with world_offset = r*8192 against one
  65 536-world env, 50 auto-reset steps, concatenated packed buffers bitwise equal -- the workload and the gather layout
  of the 8-GPU run;
* the native all-gather (`cavoid_comm_*`, `cavoid_gather_*`) with one rank (a device copy on the communicator's stream),
  driven through the double-buffered overlap protocol of `ShardedEnv.step_and_gather`.

The oracle comparison of the same arithmetic is in test_gpu_parity.py; these tests pin the new launch forms to the
single-step plain path that those parity tests cover, plus one direct oracle check of a multi-step packed run."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import c_oracle as co

pytestmark = pytest.mark.gpu


def _env(W, N, M=None, seed=0, offset=0, **over):
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    from rl_collision_avoidance_amd.config import EnvConfig

    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
            self.MAX_NUM_OTHER_AGENTS_OBSERVED = N - 1 if M is None else M
            EnvConfig.__init__(self)
    return BatchedCollisionAvoidanceEnv(W, Cfg(), device="cuda:0", seed=seed, world_offset=offset, **over)


def _acts(T, W, N, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 11, size=(T, W, N))
    a[rng.random((T, W, N)) < 0.6] = 2                      # mostly straight ahead: goals are reached, worlds restart
    return torch.from_numpy(a.astype(np.int32)).cuda()


@pytest.mark.parametrize("N,M,W,nonl,gen_min", [(4, None, 1000, 0.0, 4), (10, None, 333, 0.3, 2), (7, 3, 129, 0.2, 3), (2, None, 65, 0.0, 2)])
def test_packed_outputs_equal_plain_outputs(N, M, W, nonl, gen_min):
    from rl_collision_avoidance_amd.sharding import pack_step_outputs
    kw = dict(gen_min_agents=gen_min, gen_nonlearning_fraction=nonl)
    a, b = _env(W, N, M, seed=3, **kw), _env(W, N, M, seed=3, **kw)
    pk = b.new_packed()
    acts = _acts(60, W, N, 1)
    a.reset()
    b.reset_packed(pk)
    width = a.obs_width
    assert torch.equal(pk[..., :width], a.obs) and torch.equal(pk[..., width], torch.zeros_like(a.rewards))
    assert torch.equal(pk[..., width + 1], ((a.get_state()[2].view(W, N) & 0x20) == 0).float())   # absent rows are 'done'
    for t in range(60):
        if t % 3 == 2:                                      # plain step (no restart) / auto-reset step alternate
            o, r, d, g = a.step(acts[t])
            p, g2 = b.step_packed(acts[t], pk)
        else:
            o, r, d, g = a.step_autoreset(acts[t])
            p, g2 = b.step_autoreset_packed(acts[t], pk)
        assert torch.equal(p, pack_step_outputs(o, r, d)), t
        assert torch.equal(g, g2), t
    assert torch.equal(b.observe_packed(pk)[..., :width], a.observe())
    for x, y in zip(a.get_state(), b.get_state()):
        assert torch.equal(x, y)
    assert a.episode.max().item() >= 1
    a.close(); b.close()


@pytest.mark.parametrize("N,W,pool,wpw,pipe", [
    (4, 300, 65536, None, None),   # latency mode, default form: env_relay_kernel (roles on wavefronts of one workgroup per tile)
    (4, 300, 65536, None, "1"),    # ... env_pipe_kernel (two wavefronts per tile)
    (4, 300, 65536, None, "0"),    # ... one wavefront per tile, next pool record in registers
    (4, 300, 0, None, None),       # in-kernel generator
    (4, 5000, 500, 16, None),      # full wavefronts
    (4, 5000, 500, 16, "1"),
    (4, 5000, 500, 1, None),       # one world per wavefront
    (10, 257, 300, None, None),
    (10, 257, 0, 3, None),
    (3, 1000, 7, 5, None),
    (3, 1000, 7, 5, "1"),
    (2, 777, 64, None, None),
    (5, 333, 100, None, None),
    (16, 130, 64, None, None),
])
def test_multi_step_launch_equals_single_steps(N, W, pool, wpw, pipe, monkeypatch):
    if wpw is not None:
        monkeypatch.setenv("CAVOID_WPW", str(wpw))
    if pipe is not None:
        monkeypatch.setenv("CAVOID_PIPELINE", pipe)
    T = 48
    kw = dict(gen_pool_size=pool, gen_min_agents=max(1, N // 2), gen_nonlearning_fraction=0.2)
    acts = _acts(T, W, N, 5)
    a = _env(W, N, seed=9, **kw)
    monkeypatch.delenv("CAVOID_WPW", raising=False)
    monkeypatch.delenv("CAVOID_PIPELINE", raising=False)
    b = _env(W, N, seed=9, **kw)                            # default geometry, single steps
    a.reset(); b.reset()
    for lo, n in ((0, 1), (1, 7), (8, 24), (32, 16)):       # chunks of different lengths, incl. n = 1
        a.step_autoreset_n(acts[lo:lo + n])
        for t in range(lo, lo + n):
            b.step_autoreset(acts[t])
        assert torch.equal(a.obs, b.obs) and torch.equal(a.rewards, b.rewards), (lo, n)
        assert torch.equal(a.done, b.done) and torch.equal(a.game_over, b.game_over), (lo, n)
        for x, y in zip(a.get_state(), b.get_state()):
            assert torch.equal(x, y), (lo, n)
        assert torch.equal(a.episode, b.episode)
    assert a.episode.max().item() >= 1
    a.close(); b.close()


@pytest.mark.parametrize("N,W,nc", [(4, 1, 1), (4, 17, 2), (4, 1000, 4), (2, 33, 3), (5, 64, 2), (1, 70, 2), (6, 100, 2), (4, 8192, 3)])
def test_relay_kernel_short_launches_and_consumer_counts(N, W, nc, monkeypatch):
    """env_relay_kernel's hand-over protocol at its edges: launches shorter than its rings (2 ... 6 steps), every number of
    observation wavefronts, tiles with one world, scripted (static / non-cooperative) agents in the tile, a time budget that
    restarts worlds every few steps.  Reference: the same steps one per launch."""
    monkeypatch.setenv("CAVOID_PIPELINE", "2")
    monkeypatch.setenv("CAVOID_RELAY_CONSUMERS", str(nc))
    kw = dict(gen_pool_size=257, gen_min_agents=max(1, N - 2), gen_nonlearning_fraction=0.3 if N > 1 else 0.0)
    a = _env(W, N, seed=31, **kw)
    monkeypatch.delenv("CAVOID_PIPELINE", raising=False)
    monkeypatch.delenv("CAVOID_RELAY_CONSUMERS", raising=False)
    b = _env(W, N, seed=31, **kw)
    a.reset(); b.reset()
    T = 120
    acts = _acts(T, W, N, 17)
    lo = 0
    for n in (2, 3, 4, 5, 6, 2, 9, 33, 3, 53):
        a.step_autoreset_n(acts[lo:lo + n])
        for t in range(lo, lo + n):
            b.step_autoreset(acts[t])
        assert torch.equal(a.obs, b.obs) and torch.equal(a.rewards, b.rewards), (lo, n)
        assert torch.equal(a.done, b.done) and torch.equal(a.game_over, b.game_over), (lo, n)
        for x, y in zip(a.get_state(), b.get_state()):
            assert torch.equal(x, y), (lo, n)
        assert torch.equal(a.episode, b.episode)
        lo += n
    assert a.episode.max().item() >= 1
    a.close(); b.close()


def test_multi_step_packed_run_against_the_oracle():
    """One direct check of the new launch form against the float64 oracle: 40 steps in one launch, packed record."""
    W, N, T, seed = 512, 4, 40, 21
    env = _env(W, N, seed=seed)
    env.reset()
    pk = env.new_packed()
    acts = _acts(T, W, N, 2)
    env.step_autoreset_packed(acts, pk)
    ocfg, ogen = co.default_cfg(N), co.default_gen(N, N, pool_size=int(env.cfg.gen_pool_size))
    st = co.State.empty(W, N)
    ep = np.zeros(W, np.uint32)
    co.generate(ocfg, ogen, seed, st, ep)
    for t in range(T):
        oobs, orew, odone, ogo = co.step_autoreset(ocfg, ogen, seed, st, ep, acts[t].cpu().numpy())
    p = pk.cpu().numpy()
    width = env.obs_width
    d = np.abs(p[..., :width] - oobs)
    d[..., 3] = np.minimum(d[..., 3], np.abs(d[..., 3] - 2 * np.pi))
    assert d.max() <= 1e-5 and np.abs(p[..., width] - orew).max() <= 1e-5
    assert np.array_equal(p[..., width + 1].astype(np.uint8), odone) and np.array_equal(env.game_over.cpu().numpy(), ogo)
    assert np.array_equal(env.get_state()[2].cpu().numpy().view(np.uint32), st.flags)
    assert np.array_equal(env.episode.cpu().numpy().view(np.uint32), ep) and ep.max() >= 1
    env.close()


def _oracle_run(env, N, seed, acts, gen_min=None, nonl=0.0, **cfg_over):
    ocfg = co.default_cfg(N, **cfg_over)
    ogen = co.default_gen(N if gen_min is None else gen_min, N, nonl, pool_size=int(env.cfg.gen_pool_size))
    W = env.num_worlds
    st = co.State.empty(W, N)
    ep = np.zeros(W, np.uint32)
    co.generate(ocfg, ogen, seed, st, ep)
    outs = []
    for t in range(acts.shape[0]):
        outs.append(co.step_autoreset(ocfg, ogen, seed, st, ep, acts[t].cpu().numpy()))
    return outs, st, ep


@pytest.mark.parametrize("N,W,nc,rvo_frac", [(4, 8192, 3, 0.5), (4, 1000, 3, 1.0), (4, 37, 2, 0.5), (3, 600, 3, 0.5), (2, 4096, 3, 0.7), (4, 2000, 4, 0.3)])
def test_k_step_launches_with_orca_agents_equal_single_steps(N, W, nc, rvo_frac, monkeypatch):
    """World sets with ORCA (policy 3) agents in the K-step launch forms small batches take (env_pipe_kernel<N, true>; written for an ORCA
    instantiation of the role-split relay kernel, which passed it and was dropped for being slower: profiles/r06_ac_relay_rvo.txt):
    launches shorter and longer than the hand-over rings, per-step slots and the overwrite form, scripted static / non-cooperative agents
    beside the ORCA ones, restarts that bring ORCA agents.  Reference: the same steps one per launch (env_kernel<N, 1, true>), bitwise."""
    monkeypatch.setenv("CAVOID_PIPELINE", "2")
    monkeypatch.setenv("CAVOID_RELAY_CONSUMERS", str(nc))
    kw = dict(rvo_enabled=1, gen_rvo_fraction=rvo_frac, gen_nonlearning_fraction=0.6, gen_static_fraction=0.1, gen_min_agents=2, gen_pool_size=4099)
    T = 70
    acts = _acts(T, W, N, 11)
    a = _env(W, N, seed=13, **kw)
    monkeypatch.setenv("CAVOID_PIPELINE", "0")
    b = _env(W, N, seed=13, **kw)
    monkeypatch.delenv("CAVOID_PIPELINE", raising=False)
    a.reset(); b.reset()
    slots = a.new_step_slots(24)
    lo = 0
    for n in (1, 2, 5, 24, 17, 21):                          # launches shorter and longer than the rings
        assert lo + n <= T
        if n == 24:
            oa = a.step_autoreset_n(acts[lo:lo + n], slots=slots)
            last = [x[n - 1] for x in oa]
        else:
            last = a.step_autoreset_n(acts[lo:lo + n])
        for t in range(lo, lo + n):
            ob = b.step_autoreset(acts[t])
            if n == 24:
                assert all(torch.equal(x[t - lo], y) for x, y in zip(oa, ob)), (n, t)
        assert all(torch.equal(x, y) for x, y in zip(last, ob)), n
        for x, y in zip(a.get_state(), b.get_state()):
            assert torch.equal(x, y), n
        assert torch.equal(a.episode, b.episode)
        lo += n
    fl = a.get_state()[2].cpu().numpy().view(np.uint32)
    assert ((fl >> 8) & 7 == 3).any() and a.episode.max().item() >= 1        # ORCA agents were there, worlds restarted
    a.close(); b.close()


@pytest.mark.parametrize("N,W,pipe,gen_min,nonl", [
    (4, 512, None, 4, 0.0),       # env_relay_kernel (BASELINE configs[1] shape; 32 tiles)
    (4, 512, "1", 4, 0.0),        # env_pipe_kernel
    (4, 512, "0", 4, 0.0),        # one wavefront per tile, register prefetch
    (4, 40000, None, 2, 0.3),     # beyond latency mode: MODE_STEP_AUTORESET_N, on-demand pool gather
    (10, 300, None, 2, 0.3),      # configs[3] shape (pipeline)
    (3, 200, None, 2, 0.2),
    (4, 8192, None, 4, 0.0),      # EXACTLY BASELINE configs[1]: 4 agents x 8192 worlds (512 tiles, env_relay_kernel) -- the bench's launch
    (10, 8192, None, 2, 0.0),     # EXACTLY BASELINE configs[3]: 10 agents x 8192 worlds, 2..10 agents per world (1366 tiles)
    (10, 8192, None, 2, 0.3),     # ... with scripted agents in the worlds
])
def test_every_step_of_a_multi_step_launch_lands_in_its_slot_and_matches_the_oracle(N, W, pipe, gen_min, nonl, monkeypatch):
    """cavoid_step_autoreset_n with out_step_stride: slot t of the [K,W,N,.] outputs holds what env.step returned AT step t
    (ProcessAgent.py:149-157 consumes rewards / done / the next observation of EVERY step) -- compared with the float64
    oracle's step t for every t, in every form of the in-launch step loop; and the packed-record form of the same."""
    if pipe is not None:
        monkeypatch.setenv("CAVOID_PIPELINE", pipe)
    K, seed = 37, 19
    kw = dict(gen_min_agents=gen_min, gen_nonlearning_fraction=nonl)
    env, twin = _env(W, N, seed=seed, **kw), _env(W, N, seed=seed, **kw)
    monkeypatch.delenv("CAVOID_PIPELINE", raising=False)
    env.reset(); twin.reset()
    acts = _acts(K, W, N, 4)
    slots = env.new_step_slots(K)
    pslots = twin.new_step_slots(K, packed=True)
    obs, rew, done, go = env.step_autoreset_n(acts, slots=slots)
    twin.step_autoreset_packed(acts, pslots)
    assert obs.shape == (K, W, N, env.obs_width) and go.shape == (K, W)
    outs, st, ep = _oracle_run(env, N, seed, acts, gen_min, nonl)
    width = env.obs_width
    restarts = 0
    for t, (oobs, orew, odone, ogo) in enumerate(outs):
        o = obs[t].cpu().numpy()
        d = np.abs(o - oobs)
        d[..., 3] = np.minimum(d[..., 3], np.abs(d[..., 3] - 2 * np.pi))
        assert d.max() <= 1e-5, (t, d.max())
        assert np.abs(rew[t].cpu().numpy() - orew).max() <= 1e-5, t
        assert np.array_equal(done[t].cpu().numpy(), odone), t
        assert np.array_equal(go[t].cpu().numpy(), ogo), t
        restarts += int(ogo.sum())
    assert restarts > 0
    assert np.array_equal(env.get_state()[2].cpu().numpy().view(np.uint32), st.flags)
    assert np.array_equal(env.episode.cpu().numpy().view(np.uint32), ep)
    # the packed form: the same bits, one record per agent and step
    assert torch.equal(pslots.packed[..., :width], obs) and torch.equal(pslots.packed[..., width], rew)
    assert torch.equal(pslots.packed[..., width + 1], done.float()) and torch.equal(pslots.game_over, go)
    # a shorter launch into the same slots touches only its own
    before = obs[5:].clone()
    env.step_autoreset_n(acts[:5], slots=slots)
    assert torch.equal(obs[5:], before)
    # and the overwrite form (out_step_stride = 0) leaves the LAST step's outputs, as before
    twin2 = _env(W, N, seed=seed, **kw)
    twin2.reset()
    o2, r2, d2, g2 = twin2.step_autoreset_n(acts)
    assert torch.equal(o2, pslots.packed[K - 1, ..., :width]) and torch.equal(g2, pslots.game_over[K - 1])
    for e in (env, twin, twin2):
        e.close()


def test_prepared_launch_equals_the_plain_call():
    """prepared_autoreset_n: arguments checked and converted once, then the same launch per call -- bitwise what step_autoreset_n does,
    into slots and into the env's own buffers; its argument checks are the plain call's."""
    W, N, K = 777, 4, 12
    acts = _acts(3 * K, W, N, 21)
    a, b = _env(W, N, seed=5), _env(W, N, seed=5)
    a.reset(); b.reset()
    sa, sb = a.new_step_slots(K), b.new_step_slots(K)
    go = a.prepared_autoreset_n(acts[:K], K, slots=sa)
    for rep in range(3):                                    # the same slices again: the launch is the same, the worlds move on
        oa = go()
        ob = b.step_autoreset_n(acts[:K], K, slots=sb)
        assert all(torch.equal(x, y) for x, y in zip(oa, ob)), rep
        assert oa[0] is sa.obs and oa[3] is sa.game_over
    go2 = a.prepared_autoreset_n(acts, 5)                   # no slots: the env's own buffers hold the last step's outputs
    oa, ob = go2(), b.step_autoreset_n(acts, 5)
    assert all(torch.equal(x, y) for x, y in zip(oa, ob)) and oa[0] is a.obs
    assert all(torch.equal(x, y) for x, y in zip(a.get_state(), b.get_state())) and torch.equal(a.episode, b.episode)
    with pytest.raises(ValueError):
        a.prepared_autoreset_n(acts[:4], 5)
    with pytest.raises(ValueError):
        a.prepared_autoreset_n(acts, K, slots=a.new_step_slots(K, packed=True))
    with pytest.raises(ValueError):
        a.prepared_autoreset_n(acts, K + 1, slots=sa)
    a.close(); b.close()


def test_step_slots_argument_checks():
    import ctypes as C
    from rl_collision_avoidance_amd import _lib
    W, N = 64, 4
    env = _env(W, N)
    env.reset()
    acts = _acts(8, W, N, 0)
    lib = _lib.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    slots = env.new_step_slots(8)
    # slots that overlap (stride < W) are refused; a NULL obs is fine in the single-wavefront forms and refused by none
    rc = lib.cavoid_step_autoreset_n(env._h, p(acts), W * N, 8, W - 1, p(slots.obs), p(slots.rewards), p(slots.done), p(slots.game_over), None)
    assert rc == -1
    rc = lib.cavoid_step_autoreset_n(env._h, p(acts), W * N, 8, W, None, p(slots.rewards), p(slots.done), p(slots.game_over), None)
    assert rc == 0                                            # obs = NULL: rewards / done / game_over only (no fault)
    torch.cuda.synchronize()
    with pytest.raises(ValueError):
        env.step_autoreset_n(acts, slots=env.new_step_slots(4))
    with pytest.raises(ValueError):
        env.step_autoreset_n(acts, slots=env.new_step_slots(8, packed=True))
    env.close()


def test_native_gather_of_multi_step_blocks_and_to_a_root_single_rank():
    """K steps per launch, every step's packed records gathered as one block (cavoid_gatherv_begin: the ragged / rooted /
    multi-step form) -- with one rank a device copy through the same double-buffered protocol."""
    from rl_collision_avoidance_amd.sharding import ShardedEnv
    from rl_collision_avoidance_amd.config import EnvConfig
    W, N, K, seed = 1000, 4, 6, 13
    for root in (-1, 0):
        sh = ShardedEnv(W, EnvConfig(), device=torch.device("cuda", 0), seed=seed)
        ref = _env(W, N, seed=seed)
        sh.reset(); ref.reset()
        acts = _acts(5 * K, W, N, 8)
        rs = ref.new_step_slots(K, packed=True)
        prev = None
        for r in range(5):
            slot = sh.step_and_gather(acts[r * K:(r + 1) * K], root=root)
            ref.step_autoreset_packed(acts[r * K:(r + 1) * K], rs)
            if prev is not None:
                assert torch.equal(sh.gathered(prev[0]), prev[1]), r - 1
            prev = (slot, rs.packed.clone())
        assert torch.equal(sh.gathered(prev[0]), prev[1])
        blocks = sh.gathered_blocks(prev[0])                 # the gather as it arrives: one [K, worlds of rank r, N, .] view per rank
        assert len(blocks) == 1 and torch.equal(blocks[0], prev[1]) and blocks[0].data_ptr() == sh._recv[prev[0]].data_ptr()
        sh.close(); ref.close()


def test_configs2_eight_shards_of_8192_equal_one_65536_world_env():
    """BASELINE configs[2] on one GPU: the concatenated packed (obs | reward | done) buffers of 8 shard envs equal the
    unsharded env's, bitwise, over 50 auto-reset steps (global-world-id RNG keying + the gather layout at full size)."""
    R, Wl, N, steps, seed = 8, 8192, 4, 50, 77
    W = R * Wl
    full = _env(W, N, seed=seed)
    shards = [_env(Wl, N, seed=seed, offset=r * Wl) for r in range(R)]
    pk_full = full.new_packed()
    pk = [e.new_packed() for e in shards]
    full.reset_packed(pk_full)
    for e, p in zip(shards, pk):
        e.reset_packed(p)
    assert torch.equal(torch.cat(pk), pk_full)
    g = torch.Generator(device="cuda").manual_seed(5)
    for t in range(steps):
        acts = torch.randint(0, 11, (W, N), generator=g, device="cuda", dtype=torch.int32)
        acts[torch.rand((W, N), generator=g, device="cuda") < 0.6] = 2
        full.step_autoreset_packed(acts, pk_full)
        for r, (e, p) in enumerate(zip(shards, pk)):
            e.step_autoreset_packed(acts[r * Wl:(r + 1) * Wl], p)
        assert torch.equal(torch.cat(pk), pk_full), t
        assert torch.equal(torch.cat([e.game_over for e in shards]), full.game_over), t
    assert torch.equal(torch.cat([e.episode for e in shards]), full.episode) and full.episode.max().item() >= 1
    # and the same 50 steps again as ONE multi-step launch per env (the bench's launch form)
    acts = _acts(10, W, N, 3)
    full.step_autoreset_packed(acts, pk_full)
    for r, (e, p) in enumerate(zip(shards, pk)):
        e.step_autoreset_packed(acts[:, r * Wl:(r + 1) * Wl].contiguous(), p)
    assert torch.equal(torch.cat(pk), pk_full)
    for e in shards + [full]:
        e.close()


def test_native_gather_single_rank_overlap_protocol():
    """cavoid_comm_* with one rank: the all-gather degenerates to a device copy on the communicator's own stream; the
    double-buffered step/gather protocol must hand every step's packed record through unchanged."""
    from rl_collision_avoidance_amd.sharding import ShardedEnv
    from rl_collision_avoidance_amd.config import EnvConfig
    W, N, steps, seed = 2048, 4, 30, 13
    sh = ShardedEnv(W, EnvConfig(), device=torch.device("cuda", 0), seed=seed)
    ref = _env(W, N, seed=seed)
    sh.reset(); ref.reset()
    acts = _acts(steps, W, N, 8)
    pk = ref.new_packed()
    prev = None
    for t in range(steps):
        slot = sh.step_and_gather(acts[t])
        ref.step_autoreset_packed(acts[t], pk)
        if prev is not None:                                  # consume gather t-1 while gather t is in flight
            assert torch.equal(sh.gathered(prev[0]), prev[1]), t - 1
        prev = (slot, pk.clone())
    assert torch.equal(sh.gathered(prev[0]), prev[1])
    sh.close(); ref.close()


def test_comm_rejects_bad_arguments():
    import ctypes as C
    from rl_collision_avoidance_amd import _lib
    lib = _lib.lib()
    h = C.c_void_p()
    assert lib.cavoid_comm_create(None, 2, 0, 0, C.byref(h)) == -1          # nranks > 1 needs an id
    assert lib.cavoid_comm_create(None, 1, 1, 0, C.byref(h)) == -1          # rank out of range
    assert lib.cavoid_comm_create(None, 1, 0, 0, C.byref(h)) == 0
    buf = torch.zeros(16, device="cuda")
    assert lib.cavoid_gather_begin(h, 2, C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr()), 16, None) == -1   # slot
    assert lib.cavoid_gather_wait(h, 0, None) == 0
    lib.cavoid_comm_destroy(h)
