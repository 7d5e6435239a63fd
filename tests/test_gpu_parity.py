"""GPU parity tests proper: the HIP path (through the C ABI, via BatchedCollisionAvoidanceEnv)
against the float64 CPU oracle on identical seeded inputs.

Bar (BASELINE.json north_star): collision/goal/done/game_over flags BIT-EXACT, float observations
and rewards within 1e-5.  The env half of the oracle is parity-unpinned (reference env source
absent) -- these tests prove HIP == oracle, not oracle == upstream."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import c_oracle as co

pytestmark = pytest.mark.gpu

OBS_TOL = 1e-5      # tolerance stated by north_star for float observations / rewards
STATE_TOL = 1e-9    # float64 world state (positions, heading, time) after free-running both sides


def _env(W, N, M=None, seed=0, **over):
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    from rl_collision_avoidance_amd.config import EnvConfig

    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
            self.MAX_NUM_OTHER_AGENTS_OBSERVED = N - 1 if M is None else M
            EnvConfig.__init__(self)
    return BatchedCollisionAvoidanceEnv(W, Cfg(), device="cuda:0", seed=seed, **over)


DEFAULT_POOL = 65536     # cavoid_default_cfg's gen_pool_size


def _oracle(N, M=None, gen_min=None, nonl=0.0, pool=DEFAULT_POOL, **over):
    cfg = co.default_cfg(N, N - 1 if M is None else M, **over)
    gen = co.default_gen(N if gen_min is None else gen_min, N, nonl, pool_size=pool)
    return cfg, gen


def _push(env, st):
    env.set_state(torch.from_numpy(st.f64).cuda(), torch.from_numpy(st.f32).cuda(),
                  torch.from_numpy(st.flags.view(np.int32)).cuda())


def _pull(env):
    f64, f32, fl = env.get_state()
    return f64.cpu().numpy(), f32.cpu().numpy(), fl.cpu().numpy().view(np.uint32)


def _goal_seeking_actions(rng, W, N, p_straight=0.8):
    """uniform random actions, biased to 'full speed straight ahead' (index 2) so that agents also
    REACH goals (pure noise mostly times out or collides)."""
    acts = rng.integers(0, 11, size=(W, N))
    acts[rng.random((W, N)) < p_straight] = 2
    return acts.astype(np.int32)


def _compare_step(tag, env_out, ora_out, env, st):
    obs, rew, done, go = [t.cpu().numpy() for t in env_out]
    oobs, orew, odone, ogo = ora_out
    assert np.array_equal(done, odone), tag
    assert np.array_equal(go, ogo), tag
    f64, f32, fl = _pull(env)
    assert np.array_equal(fl, st.flags), tag                       # every flag bit, incl. collision / goal
    assert np.array_equal(f32, st.f32), tag
    np.testing.assert_allclose(f64, st.f64, rtol=0, atol=STATE_TOL, err_msg=str(tag))
    np.testing.assert_allclose(rew, orew, rtol=0, atol=OBS_TOL, err_msg=str(tag))
    # heading_ego_frame (col 3) is an ANGLE: the reference's wrap() has its branch cut at +-pi, and an
    # agent that has just run over its goal centre sits exactly on it (goal dead astern), where a
    # 1-ulp atan2 difference turns -pi into +pi.  Compare that column on the circle; all else plain.
    dh = np.abs(obs[..., 3].astype(np.float64) - oobs[..., 3])
    dh = np.minimum(dh, np.abs(dh - 2.0 * np.pi))
    assert dh.max() <= OBS_TOL, (tag, dh.max())
    keep = np.ones(obs.shape[-1], bool)
    keep[3] = False
    np.testing.assert_allclose(obs[..., keep], oobs[..., keep], rtol=0, atol=OBS_TOL, err_msg=str(tag))
    assert np.array_equal(obs[..., :2], oobs[..., :2].astype(np.float32)), tag   # is_learning, num_other exact


@pytest.mark.parametrize("N,M,sort,nonl,gen_min", [
    (4, None, 0, 0.0, 4),      # BASELINE configs[1] shape
    (4, None, 1, 0.4, 2),      # closest_first, scripted agents, 2..4 agents
    (2, None, 0, 0.0, 2),      # configs[0] shape
    (3, None, 2, 0.3, 2),      # time-to-impact ordering
    (10, None, 0, 0.3, 2),     # configs[3]: TrainPhase2-style, variable neighbour count
    (10, 4, 1, 0.0, 5),        # clipping to the 4 closest
    (5, 7, 0, 0.2, 1),         # M > N-1: padded slots (weight-sharing arch layout)
    (7, None, 0, 0.0, 7),      # 64 % N != 0: idle tail lanes
    (16, None, 0, 0.1, 9),
])
def test_trajectory_parity(N, M, sort, nonl, gen_min):
    W, steps, seed = 777, 140, 11
    ocfg, ogen = _oracle(N, M, gen_min, nonl, sort_method=sort)
    env = _env(W, N, M, seed=seed, sort_method=sort, gen_min_agents=gen_min, gen_nonlearning_fraction=nonl)
    st = co.State.empty(W, N)
    co.generate(ocfg, ogen, seed, st, np.zeros(W, np.uint32))
    _push(env, st)
    np.testing.assert_allclose(env.observe().cpu().numpy(), co.observe(ocfg, st), rtol=0, atol=OBS_TOL)
    rng = np.random.default_rng(seed)
    for t in range(steps):
        acts = _goal_seeking_actions(rng, W, N)
        out = env.step(torch.from_numpy(acts).cuda())
        _compare_step((N, M, sort, t), out, co.step(ocfg, st, acts), env, st)
    present = st.flags & 0x20 != 0
    assert (st.flags[present] & 7 != 0).mean() > 0.5       # most agents reached a terminal flag
    assert (st.flags & 4 != 0).any() and (st.flags & 1 != 0).any() and (st.flags & 2 != 0).any()
    env.close()


@pytest.mark.parametrize("pool", [0, 500, DEFAULT_POOL])
def test_reset_generator_parity(pool):
    W, N, seed = 1000, 6, 99
    ocfg, ogen = _oracle(N, None, 2, 0.3, pool=pool)
    env = _env(W, N, seed=seed, gen_min_agents=2, gen_nonlearning_fraction=0.3, gen_pool_size=pool)
    obs = env.reset().cpu().numpy()
    st = co.State.empty(W, N)
    co.generate(ocfg, ogen, seed, st, np.zeros(W, np.uint32))
    f64, f32, fl = _pull(env)
    assert np.array_equal(fl, st.flags)
    assert np.array_equal(f32, st.f32)
    np.testing.assert_allclose(f64, st.f64, rtol=0, atol=1e-12)
    np.testing.assert_allclose(obs, co.observe(ocfg, st), rtol=0, atol=OBS_TOL)
    assert np.array_equal(env.episode.cpu().numpy(), np.zeros(W, np.int32))
    # masked reset advances only the masked worlds
    mask = (np.arange(W) % 3 == 0).astype(np.uint8)
    obs2 = env.reset(torch.from_numpy(mask).cuda()).cpu().numpy()
    ep = mask.astype(np.uint32)
    co.generate(ocfg, ogen, seed, st, ep, mask)
    f64, f32, fl = _pull(env)
    assert np.array_equal(fl, st.flags) and np.array_equal(f32, st.f32)
    np.testing.assert_allclose(obs2, co.observe(ocfg, st), rtol=0, atol=OBS_TOL)
    assert np.array_equal(env.episode.cpu().numpy().view(np.uint32), ep)
    env.close()


@pytest.mark.parametrize("N,gen_min,nonl,pool", [(4, 4, 0.0, DEFAULT_POOL), (4, 4, 0.0, 0), (10, 2, 0.2, 300), (10, 2, 0.2, 0),
                                                 (3, 1, 0.5, 7)])
def test_autoreset_parity(N, gen_min, nonl, pool):
    W, steps, seed = 512, 300, 5
    ocfg, ogen = _oracle(N, None, gen_min, nonl, pool=pool)
    env = _env(W, N, seed=seed, gen_min_agents=gen_min, gen_nonlearning_fraction=nonl, gen_pool_size=pool)
    env.reset()
    st = co.State.empty(W, N)
    ep = np.zeros(W, np.uint32)
    co.generate(ocfg, ogen, seed, st, ep)
    rng = np.random.default_rng(seed)
    for t in range(steps):
        acts = _goal_seeking_actions(rng, W, N)
        out = env.step_autoreset(torch.from_numpy(acts).cuda())
        ora = co.step_autoreset(ocfg, ogen, seed, st, ep, acts)
        _compare_step(("autoreset", N, t), out, ora, env, st)
        assert np.array_equal(env.episode.cpu().numpy().view(np.uint32), ep)
    assert (ep >= 1).mean() > 0.8       # most worlds finished at least one episode
    env.close()


def test_step_autoreset_n_matches_single_steps():
    W, N, T, seed = 300, 4, 64, 3
    rng = np.random.default_rng(seed)
    acts = torch.from_numpy(rng.integers(0, 11, size=(T, W, N)).astype(np.int32)).cuda()
    a = _env(W, N, seed=seed)
    b = _env(W, N, seed=seed)
    a.reset(); b.reset()
    for t in range(T):
        a.step_autoreset(acts[t])
    b.step_autoreset_n(acts)
    for x, y in zip(a.get_state(), b.get_state()):
        assert torch.equal(x, y)
    assert torch.equal(a.obs, b.obs) and torch.equal(a.rewards, b.rewards) and torch.equal(a.episode, b.episode)
    a.close(); b.close()


@pytest.mark.parametrize("dyn", ["unicycle", "unicycle_max_turn_rate", "holonomic"])
def test_continuous_actions_parity(dyn):
    W, N, steps, seed = 400, 4, 80, 21
    code = {"unicycle": 0, "unicycle_max_turn_rate": 1, "holonomic": 2}[dyn]
    ocfg, ogen = _oracle(N, dynamics=code)
    env = _env(W, N, seed=seed, dynamics=dyn)
    st = co.State.empty(W, N)
    co.generate(ocfg, ogen, seed, st, np.zeros(W, np.uint32))
    _push(env, st)
    rng = np.random.default_rng(seed)
    for t in range(steps):
        if dyn == "holonomic":      # velocity toward the goal with noise
            g = np.stack([st.f32[0] - st.f64[0], st.f32[1] - st.f64[1]], -1)
            v = g / np.maximum(np.linalg.norm(g, axis=-1, keepdims=True), 1e-6) * st.f32[3][:, None]
            acts = (v + rng.normal(0, 0.3, size=v.shape)).astype(np.float32).reshape(W, N, 2)
        else:
            acts = np.stack([rng.uniform(0, 1.5, (W, N)), rng.uniform(-1.5, 1.5, (W, N))], -1).astype(np.float32)
        out = env.step_continuous(torch.from_numpy(acts).cuda())
        _compare_step((dyn, t), out, co.step(ocfg, st, None, acts), env, st)
    env.close()


def _cont_actions(rng, dyn, st, K, W, N):
    """K slices of continuous actions [K,W,N,2]: holonomic = a velocity towards the goal (as seen from `st`) + noise, unicycle =
    (speed, heading change) with the heading change biased towards the goal so that agents also arrive"""
    g = np.stack([st.f32[0] - st.f64[0], st.f32[1] - st.f64[1]], -1).reshape(W, N, 2)
    if dyn == "holonomic":
        v = g / np.maximum(np.linalg.norm(g, axis=-1, keepdims=True), 1e-6) * st.f32[3].reshape(W, N, 1)
        return (v[None] + rng.normal(0, 0.3, size=(K, W, N, 2))).astype(np.float32)
    to_goal = np.arctan2(g[..., 1], g[..., 0]) - st.f64[2].reshape(W, N)
    to_goal = (to_goal + np.pi) % (2 * np.pi) - np.pi
    dh = np.clip(to_goal, -0.5, 0.5)[None] * (rng.random((K, W, N)) < 0.7) + rng.uniform(-0.4, 0.4, (K, W, N))
    sp = st.f32[3].reshape(1, W, N) * rng.uniform(0.3, 1.0, (K, W, N))
    return np.stack([sp, dh], -1).astype(np.float32)


@pytest.mark.parametrize("source", ["pool", "lookahead", "instep"])
@pytest.mark.parametrize("dyn", ["unicycle", "unicycle_max_turn_rate", "holonomic"])
def test_continuous_actions_in_the_autoreset_and_k_step_launch_forms(dyn, source):
    """Round 6 (north_star "unicycle/holonomic dynamics"): continuous / velocity actions in EVERY launch form -- one
    auto-reset step per launch, K steps per launch with every step's outputs in its own slot (plain and packed records) -- with restarts
    from the pool, the look-ahead rings and the in-step generator; every step against the float64 oracle's step."""
    W, N, seed = 777, 4, 23
    code = {"unicycle": 0, "unicycle_max_turn_rate": 1, "holonomic": 2}[dyn]
    over = {"pool": dict(gen_pool_size=300), "lookahead": dict(gen_pool_size=0, gen_lookahead=64), "instep": dict(gen_pool_size=0)}[source]
    ocfg, ogen = _oracle(N, None, 2, 0.3, pool=over["gen_pool_size"], dynamics=code)
    env = _env(W, N, seed=seed, dynamics=dyn, gen_min_agents=2, gen_nonlearning_fraction=0.3, **over)
    env.reset()
    st, ep = co.State.empty(W, N), np.zeros(W, np.uint32)
    co.generate(ocfg, ogen, seed, st, ep)
    rng = np.random.default_rng(seed)

    def check(tag, obs, rew, done, go, ora):
        oobs, orew, odone, ogo = ora
        assert np.array_equal(done, odone) and np.array_equal(go, ogo), tag
        np.testing.assert_allclose(rew, orew, rtol=0, atol=OBS_TOL, err_msg=str(tag))
        d = np.abs(obs - oobs)
        d[..., 3] = np.minimum(d[..., 3], np.abs(d[..., 3] - 2.0 * np.pi))
        assert d.max() <= OBS_TOL, (tag, d.max())

    def check_state(tag):
        f64, f32, fl = _pull(env)
        assert np.array_equal(fl, st.flags) and np.array_equal(f32, st.f32), tag
        np.testing.assert_allclose(f64, st.f64, rtol=0, atol=STATE_TOL, err_msg=str(tag))
        assert np.array_equal(env.episode.cpu().numpy().view(np.uint32), ep), tag
    # ---- one auto-reset step per launch ----------------------------------------------------------------------------------
    for t in range(50):
        a = _cont_actions(rng, dyn, st, 1, W, N)[0]
        out = env.step_continuous_autoreset(torch.from_numpy(a).cuda())
        ora = co.step_autoreset(ocfg, ogen, seed, st, ep, None, cont=a)
        check((dyn, source, "single", t), *[v.cpu().numpy() for v in out], ora)
        check_state((dyn, source, "single", t))
    # ---- K steps per launch, per-step slots (plain, then packed records) --------------------------------------------------
    K = 16
    slots, pslots = env.new_step_slots(K), env.new_step_slots(K, packed=True)
    wdt = env.obs_width
    for launch in range(7):
        a = _cont_actions(rng, dyn, st, K, W, N)
        packed = launch >= 5
        n = K if launch != 3 else 11                        # (a launch shorter than its slots and its action slices)
        if packed:
            pk, go = env.step_continuous_autoreset(torch.from_numpy(a).cuda(), slots=pslots)
            obs, rew, done = pk[..., :wdt], pk[..., wdt], pk[..., wdt + 1].to(torch.uint8)
        else:
            obs, rew, done, go = env.step_continuous_autoreset(torch.from_numpy(a).cuda(), n_steps=n, slots=slots)
        for t in range(n):
            ora = co.step_autoreset(ocfg, ogen, seed, st, ep, None, cont=a[t])
            check((dyn, source, launch, t), obs[t].cpu().numpy(), rew[t].cpu().numpy(), done[t].cpu().numpy(), go[t].cpu().numpy(), ora)
        check_state((dyn, source, launch))
    assert ep.max() >= 1 and (ep >= 1).mean() > 0.3        # the restarts were exercised
    # table actions are refused for the holonomic dynamics in every form (they have no velocity meaning), loudly
    if dyn == "holonomic":
        ai = torch.zeros((W, N), dtype=torch.int32, device="cuda")
        with pytest.raises(RuntimeError):
            env.step_autoreset(ai)
    env.close()


def test_continuous_k_step_launch_at_the_benchmark_shape_and_argument_checks():
    """4 x 8192 (BASELINE configs[1]'s batch), 24 continuous-action steps in ONE launch with per-step slots, every slot against the oracle;
    == the same steps one launch each, bit for bit; overlapping float slices are refused"""
    import ctypes as C
    from rl_collision_avoidance_amd import _lib
    W, N, K, seed = 8192, 4, 24, 31
    ocfg, ogen = _oracle(N)
    env, twin = _env(W, N, seed=seed), _env(W, N, seed=seed)
    env.reset(); twin.reset()
    st, ep = co.State.empty(W, N), np.zeros(W, np.uint32)
    co.generate(ocfg, ogen, seed, st, ep)
    rng = np.random.default_rng(seed)
    for rnd in range(3):                                    # (three launches: past the first restarts)
        a = _cont_actions(rng, "unicycle", st, K, W, N)
        slots = env.new_step_slots(K)
        obs, rew, done, go = env.step_continuous_autoreset(torch.from_numpy(a).cuda(), slots=slots)
        for t in range(K):
            oobs, orew, odone, ogo = co.step_autoreset(ocfg, ogen, seed, st, ep, None, cont=a[t])
            d = np.abs(obs[t].cpu().numpy() - oobs)
            d[..., 3] = np.minimum(d[..., 3], np.abs(d[..., 3] - 2.0 * np.pi))
            assert d.max() <= OBS_TOL and np.abs(rew[t].cpu().numpy() - orew).max() <= OBS_TOL, (rnd, t)
            assert np.array_equal(done[t].cpu().numpy(), odone) and np.array_equal(go[t].cpu().numpy(), ogo), (rnd, t)
            o1 = twin.step_continuous_autoreset(torch.from_numpy(a[t]).cuda())
            assert torch.equal(o1[0], obs[t]) and torch.equal(o1[1], rew[t]) and torch.equal(o1[2], done[t]) and torch.equal(o1[3], go[t]), (rnd, t)
        assert np.array_equal(_pull(env)[2], st.flags) and np.array_equal(env.episode.cpu().numpy().view(np.uint32), ep)
    assert ep.max() >= 1
    lib = _lib.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    at = torch.from_numpy(a).cuda()
    rc = lib.cavoid_step_continuous_autoreset_n(env._h, p(at), 2 * W * N - 1, 4, W, p(slots.obs), p(slots.rewards), p(slots.done), p(slots.game_over), None)
    assert rc == -1                                         # float slices of consecutive steps must not overlap
    rc = lib.cavoid_step_continuous_autoreset_n(env._h, None, 2 * W * N, 4, W, p(slots.obs), p(slots.rewards), p(slots.done), p(slots.game_over), None)
    assert rc == -1
    torch.cuda.synchronize()
    env.close(); twin.close()


def test_u_switches_follow_the_oracle():
    """close-penalty sign (U5), float64 action array, timeout off (U1), time budget from the goal centre (U11):
    flipped on both sides."""
    W, N, steps, seed = 300, 4, 120, 8
    over = dict(close_penalty_slope=-0.5, actions_fp32=0, timeout_enabled=0, time_budget_from_goal_edge=0)
    ocfg, ogen = _oracle(N, **over)
    env = _env(W, N, seed=seed, **over)
    st = co.State.empty(W, N)
    co.generate(ocfg, ogen, seed, st, np.zeros(W, np.uint32))
    _push(env, st)
    rng = np.random.default_rng(seed)
    for t in range(steps):
        acts = rng.integers(0, 11, size=(W, N)).astype(np.int32)
        _compare_step(("U", t), env.step(torch.from_numpy(acts).cuda()), co.step(ocfg, st, acts), env, st)
    assert not (st.flags & 2).any()
    env.close()


@pytest.mark.parametrize("over", [dict(wrap_closed_end=1), dict(done_agents_collide=0), dict(sort_round_gap=0),
                                  dict(sort_tie_lateral=0),
                                  dict(wrap_closed_end=1, done_agents_collide=0, sort_round_gap=0, sort_tie_lateral=0)])
@pytest.mark.parametrize("N,sort", [(4, 0), (10, 1), (5, 2)])
def test_u2_u4_u7_switches_follow_the_oracle(over, N, sort):
    """SURVEY App. A U2 (wrap end), U4 (done agents still collide with movers), U7 (gap rounding / tie-break) are named
    switches of cavoid_cfg, mirrored in both oracles and the kernels: flipped on both sides, single steps and the in-launch
    step loop."""
    W, steps, seed = 300, 100, 8
    ocfg, ogen = _oracle(N, None, 2, 0.3, sort_method=sort, **over)
    env = _env(W, N, seed=seed, sort_method=sort, gen_min_agents=2, gen_nonlearning_fraction=0.3, **over)
    st = co.State.empty(W, N)
    ep = np.zeros(W, np.uint32)
    co.generate(ocfg, ogen, seed, st, ep)
    env.reset()
    _push(env, st)
    rng = np.random.default_rng(seed)
    for t in range(steps):
        acts = _goal_seeking_actions(rng, W, N)
        out = env.step_autoreset(torch.from_numpy(acts).cuda())
        _compare_step(("U247", over, t), out, co.step_autoreset(ocfg, ogen, seed, st, ep, acts), env, st)
    assert (st.flags & 7 != 0).any() and ep.max() >= 1
    K = 24
    acts = np.stack([_goal_seeking_actions(rng, W, N) for _ in range(K)])
    slots = env.new_step_slots(K)
    obs, rew, done, go = env.step_autoreset_n(torch.from_numpy(acts).cuda(), slots=slots)
    for t in range(K):
        oobs, orew, odone, ogo = co.step_autoreset(ocfg, ogen, seed, st, ep, acts[t])
        assert np.array_equal(done[t].cpu().numpy(), odone) and np.array_equal(go[t].cpu().numpy(), ogo), t
        d = np.abs(obs[t].cpu().numpy() - oobs)
        d[..., 3] = np.minimum(d[..., 3], np.abs(d[..., 3] - 2 * np.pi))
        assert d.max() <= OBS_TOL and np.abs(rew[t].cpu().numpy() - orew).max() <= OBS_TOL, t
    assert np.array_equal(_pull(env)[2], st.flags)
    env.close()


def test_u_switch_known_answers_on_the_gpu():
    """The hand-made cases of tests/test_oracle.py::test_u_switch_known_answers through the HIP path (no oracle in the loop)."""
    # U4: a timed-out agent in the mover's path
    for collide in (1, 0):
        env = _env(1, 2, done_agents_collide=collide)
        f64 = torch.tensor([[0.0, 1.15], [0.0, 0.0], [0.0, 0.0], [50.0, 50.0]], dtype=torch.float64).cuda()
        f32 = torch.tensor([[10.0, 10.0], [0.0, 5.0], [0.5, 0.5], [1.0, 1.0], [0.0, 0.0]], dtype=torch.float32).cuda()
        fl = torch.tensor([0x20 | 0x40, 0x20 | 0x40 | 0x02], dtype=torch.int32).cuda()
        env.set_state(f64, f32, fl)
        obs, rew, done, go = env.step(torch.tensor([[2, 2]], dtype=torch.int32).cuda())
        flags = env.get_state()[2].cpu().numpy()
        if collide:
            assert rew.cpu().tolist() == [[-0.25, -0.25]] and flags[0] & 4 and flags[1] & 4
        else:
            assert rew.cpu().tolist() == [[0.0, 0.0]] and not flags[0] & 4 and not flags[1] & 4
        assert obs[0, 0, 1].item() == 1.0 and obs[0, 0, 6 + 6].item() < 0.0
        env.close()
    # U7: two neighbours 4 mm apart in gap; slot k's p_orth identifies the neighbour
    for round_gap, tie_lat, want in ((1, 1, (2, 1)), (0, 1, (1, 2)), (1, 0, (1, 2))):
        env = _env(1, 3, sort_round_gap=round_gap, sort_tie_lateral=tie_lat)
        f64 = torch.tensor([[0.0, 0.0, 0.0], [0.0, 2.004, -2.0], [0.0, 0.0, 0.0], [50.0, 50.0, 50.0]], dtype=torch.float64).cuda()
        f32 = torch.tensor([[10.0, 5.0, 5.0], [0.0, 5.0, -5.0], [0.3, 0.3, 0.3], [1.0, 1.0, 1.0], [0.0, 0.0, 0.0]], dtype=torch.float32).cuda()
        env.set_state(f64, f32, torch.full((3,), 0x20 | 0x40, dtype=torch.int32).cuda())
        row = env.observe()[0, 0].cpu().numpy()
        order = tuple(1 if row[6 + 7 * k + 1] > 0 else 2 for k in range(2))
        assert order == want, (round_gap, tie_lat, order)
        env.close()
    # U2: an agent turned to exactly +-pi keeps the sign the switch says
    for closed, want in ((0, -np.pi), (1, np.pi)):
        env = _env(1, 1, wrap_closed_end=closed, actions=[[1.0, np.pi]], actions_fp32=0)
        f64 = torch.tensor([[0.0], [0.0], [0.0], [50.0]], dtype=torch.float64).cuda()
        f32 = torch.tensor([[10.0], [0.0], [0.3], [1.0], [0.0]], dtype=torch.float32).cuda()
        env.set_state(f64, f32, torch.tensor([0x20 | 0x40], dtype=torch.int32).cuda())
        env.step(torch.zeros((1, 1), dtype=torch.int32).cuda())
        assert env.get_state()[0][2, 0].item() == want
        env.close()


@pytest.mark.parametrize("N,mode,pool", [(4, 0, 300), (6, 1, 200)])
def test_frozen_network_agents_parity_and_row_list(N, mode, pool):
    """Scripted policy 4 (the GA3C-CADRL agent mechanism): non-learning agents whose action index the caller
    supplies; cavoid_policy_rows lists them for the frozen network.  HIP vs oracle incl. the generator's draw."""
    from rl_collision_avoidance_amd import _lib
    W, steps, seed = 400, 120, 5
    ocfg, _ = _oracle(N)
    ogen = co.default_gen(2, N, 0.6, 0.2, mode=mode, rvo_fraction=0.0, frozen_fraction=0.6, pool_size=pool)
    env = _env(W, N, seed=seed, gen_min_agents=2, gen_nonlearning_fraction=0.6, gen_static_fraction=0.2, gen_frozen_fraction=0.6,
               gen_mode=mode, gen_pool_size=pool)
    env.reset()
    st = co.State.empty(W, N)
    ep = np.zeros(W, np.uint32)
    co.generate(ocfg, ogen, seed, st, ep)
    f64, f32, fl = _pull(env)
    assert np.array_equal(fl, st.flags)
    pol = (st.flags >> 8) & 7
    assert (pol == 4).sum() > 50 and (pol == 2).any() and (pol == 1).any()
    _push(env, st)
    rng = np.random.default_rng(seed)
    for t in range(steps):
        rows, count = env.policy_rows(_lib.POLICY_FROZEN_NET)
        n = int(count.item())
        got = np.sort(rows[:n].cpu().numpy())
        want = np.nonzero(((st.flags >> 8) & 7 == 4) & (st.flags & 0x20 != 0) & (st.flags & 7 == 0))[0]
        assert np.array_equal(got, want), t
        acts = _goal_seeking_actions(rng, W, N)
        out = env.step_autoreset(torch.from_numpy(acts).cuda())
        _compare_step(("frozen", t), out, co.step_autoreset(ocfg, ogen, seed, st, ep, acts), env, st)
        obs = out[0].cpu().numpy()
        present4 = (((st.flags >> 8) & 7) == 4).reshape(W, N)
        assert (obs[..., 0][present4] == 0.0).all()                  # never "learning"
    assert ep.max() >= 1
    env.close()


def test_full_size_properties():
    """BASELINE configs[1] size (4 agents x 8192 worlds): size-independent properties.
    (a) worlds are independent: a 8192-world batch == the same worlds run as two 4096 shards with
        world_offset (bitwise) -- also the multi-GPU sharding invariant;
    (b) determinism: same seed, same result;  (c) permuting agents permutes rewards/done."""
    W, N, seed, steps = 8192, 4, 1234, 60
    rng = np.random.default_rng(seed)
    acts = torch.from_numpy(rng.integers(0, 11, size=(steps, W, N)).astype(np.int32)).cuda()
    full = _env(W, N, seed=seed)
    lo = _env(W // 2, N, seed=seed)
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    hi = BatchedCollisionAvoidanceEnv(W // 2, lo.config, device="cuda:0", world_offset=W // 2, seed=seed)
    full.reset(); lo.reset(); hi.reset()
    tot_rew = torch.zeros((), dtype=torch.float64, device="cuda")
    for t in range(steps):
        o, r, d, g = full.step_autoreset(acts[t])
        o1, r1, d1, g1 = lo.step_autoreset(acts[t, :W // 2])
        o2, r2, d2, g2 = hi.step_autoreset(acts[t, W // 2:])
        assert torch.equal(o, torch.cat([o1, o2])) and torch.equal(r, torch.cat([r1, r2]))
        assert torch.equal(d, torch.cat([d1, d2])) and torch.equal(g, torch.cat([g1, g2]))
        tot_rew += r.double().sum()
    assert torch.equal(full.episode, torch.cat([lo.episode, hi.episode]))
    assert full.episode.min().item() >= 0 and full.episode.max().item() >= 1
    assert torch.isfinite(full.obs).all()
    again = _env(W, N, seed=seed)
    again.reset()
    again.step_autoreset_n(acts)
    assert torch.equal(again.obs, full.obs) and torch.equal(again.episode, full.episode)
    # (c) permutation equivariance on a fresh state
    f64, f32, fl = full.get_state()
    perm = torch.tensor([2, 0, 3, 1], device="cuda")
    idx = (torch.arange(W, device="cuda")[:, None] * N + perm[None, :]).reshape(-1)
    p = _env(W, N, seed=seed)
    p.set_state(f64[:, idx], f32[:, idx], fl[idx])
    q = _env(W, N, seed=seed)
    q.set_state(f64, f32, fl)
    a = acts[0]
    _, rq, dq, gq = q.step(a)
    _, rp, dp, gp = p.step(a[:, perm])
    assert torch.equal(rp, rq[:, perm]) and torch.equal(dp, dq[:, perm]) and torch.equal(gp, gq)
    for e in (full, lo, hi, again, p, q):
        e.close()


def test_state_roundtrip_and_observe_idempotent():
    W, N, seed = 500, 4, 2
    env = _env(W, N, seed=seed)
    env.reset()
    rng = np.random.default_rng(0)
    for t in range(30):
        env.step(torch.from_numpy(rng.integers(0, 11, size=(W, N)).astype(np.int32)).cuda())
    obs = env.obs.clone()
    s1 = [x.clone() for x in env.get_state()]
    assert torch.equal(env.observe(), obs)             # observe() reproduces the step's observation
    env2 = _env(W, N, seed=seed)
    env2.set_state(*s1)
    assert torch.equal(env2.observe(), obs)
    for x, y in zip(env2.get_state(), s1):
        assert torch.equal(x, y)
    env.close(); env2.close()


def test_errors_are_codes_not_crashes():
    from rl_collision_avoidance_amd import _lib
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    env = _env(8, 4)
    with pytest.raises(ValueError):
        env.step(torch.zeros((7, 4), dtype=torch.int32, device="cuda"))
    with pytest.raises(_lib.CavoidError):
        BatchedCollisionAvoidanceEnv(8, env.config, gen_min_agents=9)
    h = _env(8, 4, dynamics="holonomic")
    with pytest.raises(_lib.CavoidError):
        h.step(torch.zeros((8, 4), dtype=torch.int32, device="cuda"))
    # out-of-range action indices are clamped, never read out of bounds
    env.reset()
    env.step(torch.full((8, 4), 999, dtype=torch.int32, device="cuda"))
    assert torch.isfinite(env.obs).all()
    env.close(); h.close()


def test_single_world_facade_and_environment_mirror_on_gpu():
    """create_env() -> the VecEnv-of-one facade over a real 1-world GPU env, driven like ProcessAgent does
    (BASELINE configs[0] plumbing), against the oracle stepping the same world."""
    from rl_collision_avoidance_amd.env_utils import create_env, run_episode
    from rl_collision_avoidance_amd.ga3c.environment import Environment
    seed = 4000
    game, one_env = create_env(seed=seed, gen_pool_size=0)
    env = Environment(0, game=game)
    ocfg, ogen = _oracle(4, None, 2, 0.0, pool=0)
    st = co.State.empty(1, 4)
    ep = np.zeros(1, np.uint32)
    rng = np.random.default_rng(1)
    for episode in range(3):
        env.reset()
        ep[0] = episode
        co.generate(ocfg, ogen, seed, st, ep)
        np.testing.assert_allclose(env.latest_observations, co.observe(ocfg, st)[0], rtol=0, atol=OBS_TOL)
        assert env.previous_state is None or env.current_state.shape == (1, 4, 26)
        over, t = False, 0
        while not over:
            acts = {i: int(rng.integers(0, 11)) for i in range(4) if env.latest_observations[i, 0]}
            full = np.zeros((1, 4), np.int32)
            for i, a in acts.items():
                full[0, i] = a
            rewards, over, info = env.step([acts], 0, t)
            oobs, orew, odone, ogo = co.step(ocfg, st, full)
            n = len(info[0]["which_agents_done"])
            assert over == bool(ogo[0]) and [info[0]["which_agents_done"][i] for i in range(n)] == list(odone[0, :n].astype(bool))
            np.testing.assert_allclose(rewards[0], orew[0, :n], rtol=0, atol=OBS_TOL)
            d = np.abs(env.latest_observations - oobs[0])
            d[:, 3] = np.minimum(d[:, 3], np.abs(d[:, 3] - 2 * np.pi))
            assert d.max() <= OBS_TOL
            assert env.current_state.shape == (1, 4, 26) and env.previous_state.shape == (1, 4, 26)
            t += 1
        assert t > 1
    total, steps = run_episode(game, one_env)
    assert steps > 0 and np.isfinite(total)
    one_env.close()


def test_seeding_is_reproducible_and_distinct():
    a = _env(256, 4, seed=1)
    b = _env(256, 4, seed=1)
    c = _env(256, 4, seed=2)
    oa, ob, oc = a.reset().clone(), b.reset().clone(), c.reset().clone()
    assert torch.equal(oa, ob) and not torch.equal(oa, oc)
    a.seed(2)                                   # re-seeding refills the scenario pool and restarts the episode counters
    assert torch.equal(a.reset(), oc)
    for e in (a, b, c):
        e.close()


def test_hip_path_reproduces_committed_env_golden():
    """The committed fixture (oracle-generated regression anchor, see tests/golden/make_env_golden.py): flags, done
    and game_over bit-exact, observations / rewards within 1e-5 (heading on the circle)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "env_golden.npz"))
    for name in g["cases"]:
        N, M, sort, gmin, W, steps, seed = [int(v) for v in g[name + "_cfg"]]
        env = _env(W, N, M, sort_method=sort)
        st = co.State(g[name + "_f64"].copy(), g[name + "_f32"].copy(), g[name + "_flags0"].copy())
        _push(env, st)
        for t in range(steps):
            obs, rew, done, go = [x.cpu().numpy() for x in env.step(torch.from_numpy(g[name + "_actions"][t]).cuda())]
            assert np.array_equal(done, g[name + "_done"][t]) and np.array_equal(go, g[name + "_over"][t]), (name, t)
            assert np.array_equal(_pull(env)[2].reshape(W, N), g[name + "_flags"][t]), (name, t)
            d = np.abs(obs.astype(np.float64) - g[name + "_obs"][t])
            d[..., 3] = np.minimum(d[..., 3], np.abs(d[..., 3] - 2 * np.pi))
            assert g[name + "_obs"].dtype == np.float64 and d.max() <= OBS_TOL, (name, t, d.max())
            assert np.abs(rew - g[name + "_rew"][t]).max() <= OBS_TOL
        env.close()


def test_worlds_without_agents_and_single_agent_worlds():
    """Edge shapes: an empty world (no agent present) and one-agent worlds inside a batch."""
    W, N = 64, 4
    env = _env(W, N, gen_min_agents=1)
    env.reset()
    f64, f32, fl = [x.clone() for x in env.get_state()]
    fl = fl.view(W, N)
    fl[0] = 0                                          # world 0: nobody
    fl[1, 1:] = 0                                      # world 1: a single agent
    env.set_state(f64, f32, fl.reshape(-1))
    st = co.State(f64.cpu().numpy(), f32.cpu().numpy(), fl.reshape(-1).cpu().numpy().view(np.uint32))
    ocfg, _ = _oracle(N)
    obs0 = env.observe().cpu().numpy()
    assert np.all(obs0[0] == 0) and obs0[1, 0, 1] == 0 and np.all(obs0[1, 1:] == 0)
    rng = np.random.default_rng(0)
    for t in range(40):
        acts = _goal_seeking_actions(rng, W, N)
        out = env.step(torch.from_numpy(acts).cuda())
        _compare_step(("edge", t), out, co.step(ocfg, st, acts), env, st)
    obs, rew, done, go = [x.cpu().numpy() for x in out]
    assert go[0] == 1 and np.all(done[0] == 1) and np.all(rew[0] == 0) and np.all(np.isfinite(obs))
    env.close()


def test_checkpoint_resume_is_bit_exact():
    """state_dict() mid-run -> a fresh env continues with identical observations, rewards and restarts."""
    W, N, seed = 700, 4, 77
    rng = np.random.default_rng(0)
    acts = torch.from_numpy(rng.integers(0, 11, size=(120, W, N)).astype(np.int32)).cuda()
    a = _env(W, N, seed=seed)
    a.reset()
    for t in range(60):
        a.step_autoreset(acts[t])
    sd = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in a.state_dict().items()}      # as if written to disk
    b = _env(W, N, seed=123)                     # different seed on purpose: the checkpoint carries its own
    b.load_state_dict(sd)
    assert torch.equal(b.observe(), a.obs)
    for t in range(60, 120):
        oa, ra, da, ga = a.step_autoreset(acts[t])
        ob, rb, db, gb = b.step_autoreset(acts[t])
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(ga, gb)
    assert torch.equal(a.episode, b.episode) and a.episode.max().item() >= 1
    a.close(); b.close()


def test_evaluate_mode_game_over_needs_every_agent():
    """EVALUATE_MODE: the episode ends when EVERY agent is done, not only the learning ones."""
    W, N, steps, seed = 300, 4, 120, 6
    ocfg, ogen = _oracle(N, None, 2, 0.6, evaluate_mode=1)
    env = _env(W, N, seed=seed, evaluate_mode=1)
    st = co.State.empty(W, N)
    co.generate(ocfg, ogen, seed, st, np.zeros(W, np.uint32))
    _push(env, st)
    rng = np.random.default_rng(seed)
    train_cfg, _ = _oracle(N, None, 2, 0.6)
    differs = False
    for t in range(steps):
        acts = _goal_seeking_actions(rng, W, N)
        shadow = st.copy()
        out = env.step(torch.from_numpy(acts).cuda())
        ora = co.step(ocfg, st, acts)
        _compare_step(("eval", t), out, ora, env, st)
        differs |= bool((co.step(train_cfg, shadow, acts)[3] != ora[3]).any())
    assert differs          # the two rules really disagree on this workload (scripted agents outlive the learners)
    env.close()


def test_baseline_config0_two_agent_single_world_1000_steps():
    """BASELINE configs[0]: 2-agent single world, 1000 steps of plumbing through create_env()'s facade, checked
    against the oracle stepping the same world (episodes restart through reset(), as ProcessAgent does)."""
    from rl_collision_avoidance_amd.config import EnvConfig
    from rl_collision_avoidance_amd.env_utils import create_env

    class TwoAgents(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = 2
            EnvConfig.__init__(self)
    seed = 12000
    game, one_env = create_env(TwoAgents(), seed=seed, gen_pool_size=0, gen_min_agents=2)
    ocfg, ogen = _oracle(2, None, 2, 0.0, pool=0)
    st = co.State.empty(1, 2)
    ep = np.zeros(1, np.uint32)
    rng = np.random.default_rng(3)
    steps, episode = 0, 0
    while steps < 1000:
        obs = game.reset()[0]
        ep[0] = episode
        co.generate(ocfg, ogen, seed, st, ep)
        assert obs.shape == (2, 13)
        over = False
        while not over and steps < 1000:
            acts = {i: (2 if rng.random() < 0.7 else int(rng.integers(0, 11))) for i in range(2)}
            o, r, over, info = game.step([acts])
            oo, orew, odone, ogo = co.step(ocfg, st, np.array([[acts[0], acts[1]]], np.int32))
            assert over == bool(ogo[0]) and [info[0]["which_agents_done"][i] for i in range(2)] == list(odone[0].astype(bool))
            d = np.abs(o[0] - oo[0])
            d[:, 3] = np.minimum(d[:, 3], np.abs(d[:, 3] - 2 * np.pi))
            assert d.max() <= OBS_TOL and np.abs(r[0] - orew[0]).max() <= OBS_TOL
            steps += 1
        episode += 1
    assert episode >= 5
    one_env.close()


@pytest.mark.parametrize("sort", [0, 1, 2])
def test_exact_sort_ties_take_the_generic_rank_path(sort):
    """Two neighbours mirrored about the host's goal axis have the SAME centimetre bucket and the SAME lateral offset
    (ry*tx - rx*ty, bit for bit): the 63-bit tournament keys coincide and the kernel must fall back to the exact rule
    (float64 lateral, then agent index -- what the oracle's stable sort does).  Worlds without ties share the tile."""
    W, N = 40, 4
    ocfg, _ = _oracle(N, sort_method=sort)
    env = _env(W, N, sort_method=sort)
    st = co.State.empty(W, N)
    rng = np.random.default_rng(4)
    for w in range(W):
        k = slice(w * N, (w + 1) * N)
        if w % 2 == 0:        # host 0 at the origin heading for (5, 0); others mirrored in x, one below
            st.f64[0, k] = [0.0, 1.0, -1.0, 0.0]
            st.f64[1, k] = [0.0, 0.8, 0.8, -2.0]
            st.f32[0, k] = [5.0, 1.0, -1.0, 0.0]
            st.f32[1, k] = [0.0, 5.0, 5.0, -6.0]
            st.f32[2, k] = 0.3
        else:
            st.f64[0, k] = rng.uniform(-4, 4, N)
            st.f64[1, k] = rng.uniform(-4, 4, N)
            st.f32[0, k] = rng.uniform(-6, 6, N)
            st.f32[1, k] = rng.uniform(-6, 6, N)
            st.f32[2, k] = rng.uniform(0.2, 0.5, N)
        st.f32[3, k] = 1.0
        st.f64[2, k] = np.arctan2(st.f32[1, k] - st.f64[1, k], st.f32[0, k] - st.f64[0, k])
        st.f64[3, k] = 50.0
        st.flags[k] = 0x20 | 0x40
    _push(env, st)
    want = co.observe(ocfg, st)
    got = env.observe().cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=OBS_TOL)
    # the tie really is one: host 0 of an even world sees its two mirrored neighbours at the same rounded gap
    row = want[0, 0]
    gaps = row[6 + 6::7][:3]
    assert np.min(np.abs(np.subtract.outer(gaps, gaps))[~np.eye(3, dtype=bool)]) < 1e-12
    # and a few steps from there (the tie persists while the mirrored pair moves symmetrically)
    for t in range(5):
        acts = np.full((W, N), 2, np.int32)
        _compare_step(("ties", sort, t), env.step(torch.from_numpy(acts).cuda()), co.step(ocfg, st, acts), env, st)
    env.close()


@pytest.mark.parametrize("N,M,mode,nonl,static,rvo,gen_min,sort", [
    (4, None, 1, 0.6, 0.2, 0.6, 2, 0),       # box scenarios, RVO + static + non-coop + learners
    (4, None, 0, 0.7, 0.0, 1.0, 4, 1),       # ring scenarios (head-on, crowded centre): ORCA's infeasible branch
    (10, None, 1, 0.5, 0.2, 0.5, 2, 0),      # configs[3] shape
    (6, 3, 1, 0.4, 0.3, 0.4, 3, 2),
    (15, None, 1, 0.3, 0.3, 0.5, 9, 0),      # the largest world the ORCA scratch fits
])
def test_rvo_agents_and_box_generator_parity(N, M, mode, nonl, static, rvo, gen_min, sort):
    """RVO (ORCA) scripted agents and the box-style generator GEN v2, HIP vs the float64 oracle:
    generated worlds bit-exact (statics) / 1e-12 (float64 state), trajectories to the usual bar."""
    W, steps, seed = 300, 120, 23
    ocfg, _ = _oracle(N, M, sort_method=sort)
    ogen = co.default_gen(gen_min, N, nonl, static, mode=mode, rvo_fraction=rvo)
    env = _env(W, N, M, seed=seed, sort_method=sort, gen_min_agents=gen_min, gen_nonlearning_fraction=nonl,
               gen_static_fraction=static, gen_rvo_fraction=rvo, rvo_enabled=1, gen_mode=mode, gen_pool_size=0)
    obs0 = env.reset().cpu().numpy()
    st = co.State.empty(W, N)
    co.generate(ocfg, ogen, seed, st, np.zeros(W, np.uint32))
    f64, f32, fl = _pull(env)
    assert np.array_equal(fl, st.flags) and np.array_equal(f32, st.f32)
    np.testing.assert_allclose(f64, st.f64, rtol=0, atol=1e-12)
    np.testing.assert_allclose(obs0, co.observe(ocfg, st), rtol=0, atol=OBS_TOL)
    assert ((st.flags >> 8) & 7 == 3).sum() > 20                   # RVO agents exist
    _push(env, st)                                                  # continue from the oracle's (1e-16 different) headings
    rng = np.random.default_rng(seed)
    for t in range(steps):
        acts = _goal_seeking_actions(rng, W, N)
        out = env.step(torch.from_numpy(acts).cuda())
        _compare_step(("rvo", N, mode, t), out, co.step(ocfg, st, acts), env, st)
    rvo_agents = (st.flags >> 8) & 7 == 3
    assert (st.flags[rvo_agents] & 1).mean() > 0.4                 # most RVO agents arrive
    env.close()


def test_box_generator_through_the_pool_and_refresh():
    """GEN v2 restarts come from the scenario pool; cavoid_pool_refresh re-fills it with the generator's next epoch."""
    W, N, seed, pool = 400, 4, 31, 200
    ocfg, _ = _oracle(N)
    env = _env(W, N, seed=seed, gen_min_agents=2, gen_mode=1, gen_pool_size=pool, gen_nonlearning_fraction=0.5,
               gen_rvo_fraction=0.5, rvo_enabled=1)
    env.reset()
    st = co.State.empty(W, N)
    ep = np.zeros(W, np.uint32)
    for epoch in (0, 7):
        ogen = co.default_gen(2, N, 0.5, pool_size=pool, mode=1, rvo_fraction=0.5, pool_epoch=epoch)
        if epoch:
            env.refresh_pool(epoch)
            env.seed(seed)                                          # (re-seeding keeps the refreshed epoch)
            env.refresh_pool(epoch)
            env.reset()
            ep[:] = 0
        co.generate(ocfg, ogen, seed, st, ep)
        f64, f32, fl = _pull(env)
        assert np.array_equal(fl, st.flags) and np.array_equal(f32, st.f32), epoch
        rng = np.random.default_rng(epoch)
        for t in range(150):
            acts = _goal_seeking_actions(rng, W, N)
            out = env.step_autoreset(torch.from_numpy(acts).cuda())
            _compare_step(("pool", epoch, t), out, co.step_autoreset(ocfg, ogen, seed, st, ep, acts), env, st)
        assert (ep >= 1).mean() > 0.5
    # a multi-step launch over the same pool equals single steps
    twin = _env(W, N, seed=seed, gen_min_agents=2, gen_mode=1, gen_pool_size=pool, gen_nonlearning_fraction=0.5,
                gen_rvo_fraction=0.5, rvo_enabled=1)
    twin.reset(); env.seed(seed); env.refresh_pool(0); twin.refresh_pool(0); env.reset()
    acts = torch.from_numpy(np.random.default_rng(1).integers(0, 11, size=(40, W, N)).astype(np.int32)).cuda()
    env.step_autoreset_n(acts)
    for t in range(40):
        twin.step_autoreset(acts[t])
    assert torch.equal(env.obs, twin.obs) and torch.equal(env.episode, twin.episode)
    for x, y in zip(env.get_state(), twin.get_state()):
        assert torch.equal(x, y)
    # 16-agent worlds have no room for the ORCA scratch: a code, not a crash
    from rl_collision_avoidance_amd import _lib
    with pytest.raises(_lib.CavoidError):
        _env(8, 16, rvo_enabled=1)
    for e in (env, twin):
        e.close()


@pytest.mark.parametrize("N,nonl,rvo", [(4, 0.0, 0.0), (4, 0.5, 0.4), (10, 0.3, 0.0), (7, 0.4, 0.3)])
def test_box_scenarios_generated_inside_the_step(N, nonl, rvo):
    """GEN v2 with gen_pool_size = 0: a world that ends restarts INSIDE the auto-reset step with a freshly generated box
    scenario (the wavefront that owns the world places its agents cooperatively) -- the reference's fresh test case per
    episode (TEST_CASE_FN, run-ws/config.yaml:281-283).  HIP vs the oracle, single steps and the in-launch step loop."""
    W, steps, seed = 300, 150, 41
    ocfg, _ = _oracle(N)
    ogen = co.default_gen(2, N, nonl, 0.3, mode=1, rvo_fraction=rvo, pool_size=0)
    env = _env(W, N, seed=seed, gen_min_agents=2, gen_mode=1, gen_pool_size=0, gen_nonlearning_fraction=nonl, gen_static_fraction=0.3,
               gen_rvo_fraction=rvo, rvo_enabled=1 if rvo > 0 else 0)
    env.reset()
    st = co.State.empty(W, N)
    ep = np.zeros(W, np.uint32)
    co.generate(ocfg, ogen, seed, st, ep)
    _push(env, st)
    rng = np.random.default_rng(seed)
    for t in range(steps):
        acts = _goal_seeking_actions(rng, W, N)
        out = env.step_autoreset(torch.from_numpy(acts).cuda())
        _compare_step(("box-in-step", N, t), out, co.step_autoreset(ocfg, ogen, seed, st, ep, acts), env, st)
        assert np.array_equal(env.episode.cpu().numpy().view(np.uint32), ep)
    assert (ep >= 1).mean() > 0.5
    K = 30
    acts = np.stack([_goal_seeking_actions(rng, W, N) for _ in range(K)])
    slots = env.new_step_slots(K)
    obs, rew, done, go = env.step_autoreset_n(torch.from_numpy(acts).cuda(), slots=slots)
    for t in range(K):
        oobs, orew, odone, ogo = co.step_autoreset(ocfg, ogen, seed, st, ep, acts[t])
        assert np.array_equal(done[t].cpu().numpy(), odone) and np.array_equal(go[t].cpu().numpy(), ogo), t
        d = np.abs(obs[t].cpu().numpy() - oobs)
        d[..., 3] = np.minimum(d[..., 3], np.abs(d[..., 3] - 2 * np.pi))
        assert d.max() <= OBS_TOL, (t, d.max())
    assert np.array_equal(_pull(env)[2], st.flags) and np.array_equal(env.episode.cpu().numpy().view(np.uint32), ep)
    env.close()
