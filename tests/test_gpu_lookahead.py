"""Scenario look-ahead (`cavoid_cfg::gen_lookahead`, include/cavoid.h): a FRESH scenario per episode -- the reference's reset semantics
(`TEST_CASE_FN = get_testcase_random`, a new random test case at every `reset`, /root/reference/ga3c/GA3C/ProcessAgent.py:107) -- without
the generator on the step's critical path: every world owns a ring of its own next episodes' scenarios, refilled between launches.  The
contract is exactness: an env with look-ahead must be BITWISE the env that generates inside the step (`gen_pool_size = 0`), which the
parity tests hold to the float64 oracle -- in every launch form (one step, K-step loops incl. the role-split relay kernel that the
in-kernel generator cannot use, the fused actor kernel), for GEN v1 and GEN v2, across resets, re-seeding and shards."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _cfg(N):
    from rl_collision_avoidance_amd.config import EnvConfig

    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
            EnvConfig.__init__(self)
    return Cfg()


def _env(W, N, seed, lookahead, offset=0, **over):
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    kw = dict(gen_pool_size=0, **over)
    if lookahead:
        kw["gen_lookahead"] = lookahead
    return BatchedCollisionAvoidanceEnv(W, _cfg(N), device="cuda:0", seed=seed, world_offset=offset, **kw)


def _same_state(a, b):
    return all(torch.equal(x, y) for x, y in zip(a.get_state(), b.get_state())) and torch.equal(a.episode, b.episode)


@pytest.mark.parametrize("N,mode,over", [(4, 0, {}), (4, 1, {}), (10, 0, dict(gen_min_agents=2, gen_nonlearning_fraction=0.3)),
                                         (10, 1, dict(gen_min_agents=2)), (3, 1, dict(gen_nonlearning_fraction=0.5, gen_static_fraction=0.5))])
def test_lookahead_equals_generation_inside_the_step_bitwise(N, mode, over):
    W, seed = 1500 if N == 4 else 700, 31 + N + mode
    a, b = _env(W, N, seed, 0, gen_mode=mode, **over), _env(W, N, seed, 64, gen_mode=mode, **over)
    assert torch.equal(a.reset(), b.reset()) and _same_state(a, b)
    g = torch.Generator(device="cuda").manual_seed(5)
    for t in range(150):                                                   # one step per launch (the ring is refilled every R - 2 steps)
        acts = torch.randint(0, 11, (W, N), generator=g, device="cuda", dtype=torch.int32)
        ra, rb = a.step_autoreset(acts), b.step_autoreset(acts)
        assert all(torch.equal(x, y) for x, y in zip(ra, rb)), t
    assert _same_state(a, b) and a.episode.max().item() >= 1
    K = 48
    sa, sb = a.new_step_slots(K), b.new_step_slots(K)
    for l in range(4):                                                     # K-step launches, every step's outputs in its own slot
        acts = torch.randint(0, 11, (K, W, N), generator=g, device="cuda", dtype=torch.int32)
        a.step_autoreset_n(acts, K, slots=sa)
        b.step_autoreset_n(acts, K, slots=sb)
        for name in ("obs", "rewards", "done", "game_over"):
            assert torch.equal(getattr(sa, name), getattr(sb, name)), (l, name)
    assert _same_state(a, b) and a.episode.max().item() >= 3
    a.close(); b.close()


def test_lookahead_runs_the_relay_kernel_and_matches_the_oracle():
    """at 4 x 8192 the look-ahead env takes the role-split relay kernel (the in-kernel generator keeps the single-wavefront loop); its
    per-step slots against the float64 oracle with NO pool: a fresh generator scenario at every restart on both sides"""
    from oracle import c_oracle as co
    W, N, seed, K = 8192, 4, 77, 40
    env = _env(W, N, seed, 64)
    obs0 = env.reset().cpu().numpy()
    ocfg, ogen = co.default_cfg(N), co.default_gen(N, N, pool_size=0)
    st, ep = co.State.empty(W, N), np.zeros(W, np.uint32)
    oobs0 = co.generate(ocfg, ogen, seed, st, ep)
    rng = np.random.default_rng(3)
    slots = env.new_step_slots(K)
    restarts = 0
    for l in range(3):
        acts = rng.integers(0, 11, size=(K, W, N)).astype(np.int32)
        env.step_autoreset_n(torch.from_numpy(acts).cuda(), K, slots=slots)
        obs, rew, done, go = slots.obs.cpu().numpy(), slots.rewards.cpu().numpy(), slots.done.cpu().numpy(), slots.game_over.cpu().numpy()
        for k in range(K):
            oobs, orew, odone, ogo = co.step_autoreset(ocfg, ogen, seed, st, ep, acts[k])
            assert np.array_equal(done[k], odone) and np.array_equal(go[k], ogo), (l, k)
            assert np.abs(obs[k].astype(np.float64) - oobs)[..., [0, 1, 2, 4, 5]].max() <= 1e-5 and np.abs(rew[k] - orew).max() <= 1e-5, (l, k)
            restarts += int(ogo.sum())
    assert restarts > 500 and np.array_equal(env.episode.cpu().numpy().view(np.uint32), ep)
    f64, f32, fl = [x.cpu().numpy() for x in env.get_state()]
    assert np.array_equal(fl.view(np.uint32), st.flags) and np.abs(f64 - st.f64).max() <= 1e-9
    env.close()


def test_lookahead_across_masked_resets_reseeding_and_shards():
    W, N, seed = 600, 4, 9
    a, b = _env(W, N, seed, 0), _env(W, N, seed, 16)
    a.reset(); b.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    mask = (torch.rand(W, generator=g, device="cuda") < 0.3).to(torch.uint8)
    for t in range(40):
        acts = torch.randint(0, 11, (W, N), generator=g, device="cuda", dtype=torch.int32)
        assert all(torch.equal(x, y) for x, y in zip(a.step_autoreset(acts), b.step_autoreset(acts)))
        if t % 7 == 3:                                                       # a masked reset starts those worlds' next episode from the ring
            assert torch.equal(a.reset(mask), b.reset(mask)) and _same_state(a, b)
    a.seed(1234); b.seed(1234)                                               # a new seed: nothing in the rings may survive
    assert torch.equal(a.reset(), b.reset())
    for t in range(30):
        acts = torch.randint(0, 11, (W, N), generator=g, device="cuda", dtype=torch.int32)
        assert all(torch.equal(x, y) for x, y in zip(a.step_autoreset(acts), b.step_autoreset(acts)))
    # a shard with a world offset is the slice of the whole (the rings are keyed on GLOBAL world ids)
    sh = _env(200, N, 1234, 16, offset=300)
    full = _env(W, N, 1234, 16)
    o_sh, o_full = sh.reset(), full.reset()
    assert torch.equal(o_sh, o_full[300:500])
    for t in range(60):
        acts = torch.randint(0, 11, (W, N), generator=g, device="cuda", dtype=torch.int32)
        r_full, r_sh = full.step_autoreset(acts), sh.step_autoreset(acts[300:500].contiguous())
        assert torch.equal(r_sh[0], r_full[0][300:500]) and torch.equal(r_sh[3], r_full[3][300:500]), t
    for e in (a, b, sh, full):
        e.close()


def test_a_launch_longer_than_the_ring_is_refused():
    env = _env(64, 4, 3, 16)
    env.reset()
    acts = torch.zeros((16, 64, 4), device="cuda", dtype=torch.int32)
    from rl_collision_avoidance_amd._lib import CavoidError
    with pytest.raises(CavoidError) as err:
        env.step_autoreset_n(acts, 16, slots=env.new_step_slots(16))         # 16 + 1 > R = 16
    assert err.value.code == -4                                               # CAVOID_EUNSUPPORTED
    env.step_autoreset_n(acts, 15, slots=env.new_step_slots(16))             # 15 + 1 fits
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    with pytest.raises(CavoidError):                                           # look-ahead beside a hashed pool / not a power of two
        BatchedCollisionAvoidanceEnv(8, _cfg(4), device="cuda:0", gen_lookahead=16)
    with pytest.raises(CavoidError):
        BatchedCollisionAvoidanceEnv(8, _cfg(4), device="cuda:0", gen_pool_size=0, gen_lookahead=24)
    env.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_fused_actor_kernel_with_lookahead_equals_generation_inside_the_step(mode):
    """the GA3C actor loop (cavoid_actor_run, K steps per launch, also captured into a hipGraph): worlds restart inside the launch from
    their rings -- bitwise the run whose restarts generate in the step (which, for box scenarios, needs the ORCA instantiation)"""
    from rl_collision_avoidance_amd.ga3c.network import NetworkVP_rnn
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
    from rl_collision_avoidance_amd.ga3c.rollout import BatchedRollout
    W, N, K = 1024, 4, 16
    torch.manual_seed(0)
    net = NetworkVP_rnn(_cfg(N)).cuda()
    out = []
    for look in (0, 64):
        env = _env(W, N, 21, look, gen_mode=mode, gen_min_agents=2)
        roll = BatchedRollout(env, FusedPolicy(net, seed=4), reflush_done=False, ring_len=4 * K + 64)
        roll.reset()
        assert roll.fused_available
        for _ in range(6):
            roll.run_fused(K)
        if look:
            roll.capture_fused(steps_per_graph=K)
            roll.replay(2)
        else:
            roll.run_fused(2); roll.run_fused(K); roll.run_fused(K)       # (capture_fused warms up with two steps)
        torch.cuda.synchronize()
        out.append((roll.obs.clone(), env.episode.clone(), [t.clone() for t in env.get_state()], roll.x.clone(), roll.ret.clone(), roll.emit_t.clone()))
        roll.close(); env.close()
    (o0, e0, s0, x0, r0, t0), (o1, e1, s1, x1, r1, t1) = out
    assert torch.equal(o0, o1) and torch.equal(e0, e1) and all(torch.equal(p, q) for p, q in zip(s0, s1))
    assert torch.equal(x0, x1) and torch.equal(r0, r1) and torch.equal(t0, t1) and e0.max().item() >= 2


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_launch_sequences_keep_the_rings_covered(seed):
    """the host's guaranteed-cover budget under fire: random sequences of one-step launches, K-step launches of random length, masked
    resets and re-seedings on rings as short as the launches allow (R = 4 .. 32, K up to R - 1) -- a world that restarts at every
    opportunity must always find its next scenario in its ring; bitwise against generation inside the step after every launch"""
    rng = np.random.default_rng(seed)
    N = int(rng.choice([2, 4, 5]))
    R = int(rng.choice([4, 8, 16, 32]))
    mode = int(rng.integers(0, 2))
    W = int(rng.integers(50, 700))
    over = dict(gen_mode=mode, gen_min_agents=int(rng.integers(1, N + 1)))
    # short episodes: a tiny time budget and agents that start close make worlds restart every few steps
    over.update(max_time_ratio=0.3 if rng.random() < 0.5 else 2.0)
    a, b = _env(W, N, 100 + seed, 0, **over), _env(W, N, 100 + seed, R, **over)
    assert torch.equal(a.reset(), b.reset())
    g = torch.Generator(device="cuda").manual_seed(seed)
    slots_a, slots_b = a.new_step_slots(R), b.new_step_slots(R)
    restarts_seen = 0
    for it in range(60):
        kind = rng.random()
        if kind < 0.45:                                                      # a burst of one-step launches
            for _ in range(int(rng.integers(1, 2 * R))):
                acts = torch.randint(0, 11, (W, N), generator=g, device="cuda", dtype=torch.int32)
                ra, rb = a.step_autoreset(acts), b.step_autoreset(acts)
                assert all(torch.equal(x, y) for x, y in zip(ra, rb)), (it, "single")
                restarts_seen += int(ra[3].sum().item())
        elif kind < 0.85:                                                    # a K-step launch, K anywhere up to the ring's limit
            K = int(rng.integers(2, R)) if R > 2 else 1
            acts = torch.randint(0, 11, (K, W, N), generator=g, device="cuda", dtype=torch.int32)
            a.step_autoreset_n(acts, K, slots=slots_a)
            b.step_autoreset_n(acts, K, slots=slots_b)
            for name in ("obs", "rewards", "done", "game_over"):
                assert torch.equal(getattr(slots_a, name)[:K], getattr(slots_b, name)[:K]), (it, K, name)
        elif kind < 0.95:                                                    # a masked reset
            mask = (torch.rand(W, generator=g, device="cuda") < rng.random()).to(torch.uint8)
            assert torch.equal(a.reset(mask), b.reset(mask)), (it, "reset")
        else:                                                                # a new seed
            s2 = int(rng.integers(0, 1 << 30))
            a.seed(s2); b.seed(s2)
            assert torch.equal(a.reset(), b.reset()), (it, "reseed")
        assert _same_state(a, b), it
    assert restarts_seen > 0
    a.close(); b.close()
