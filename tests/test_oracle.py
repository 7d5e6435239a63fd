"""CPU tests of the oracle itself: the reference-style Python statement (oracle/cavoid_oracle.py)
and the batched C statement (oracle/cavoid_oracle.c) must agree BIT-FOR-BIT (same libm, no FMA),
plus analytic known-answer tests.  The env half is parity-unpinned:
nothing here can be compared with the absent reference env source."""
import math

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import cavoid_oracle as po


def _cross(N, M, sort, nonl, W=12, steps=100, seed=7, dyn=0, mode=0, rvo=0.0, frozen=0.0, **switches):
    pcfg = po.OracleConfig(max_agents=N, max_other_agents_observed=M, sort_method=sort, dynamics=dyn,
                           **{k: bool(v) for k, v in switches.items()})
    pgen = po.GenConfig(min_agents=2, max_agents=N, nonlearning_fraction=nonl, mode=mode, rvo_fraction=rvo, frozen_fraction=frozen)
    ccfg = co.default_cfg(N, M, sort_method=sort, dynamics=dyn, **{k: int(v) for k, v in switches.items()})
    cgen = co.default_gen(2, N, nonl, mode=mode, rvo_fraction=rvo, frozen_fraction=frozen)
    st = co.State.empty(W, N)
    ep = np.zeros(W, np.uint32)
    co.generate(ccfg, cgen, seed, st, ep)
    worlds = [po.generate_world(seed, w, 0, pcfg, pgen) for w in range(W)]
    for w, wd in enumerate(worlds):
        f64, f32, fl = po.world_to_arrays(wd)
        sl = slice(w * N, (w + 1) * N)
        assert np.array_equal(f64, st.f64[:, sl]) and np.array_equal(f32, st.f32[:, sl]) and np.array_equal(fl, st.flags[sl])
    o0 = co.observe(ccfg, st)
    for w, wd in enumerate(worlds):
        assert np.array_equal(wd.observe(), o0[w])
    rng = np.random.default_rng(seed)
    for t in range(steps):
        acts = rng.integers(0, 11, size=(W, N))
        obs, rew, done, go = co.step(ccfg, st, acts)
        for w, wd in enumerate(worlds):
            n = len(wd.agents)
            pobs, prew, pgo, info = wd.step({i: acts[w, i] for i in range(n)})
            assert np.array_equal(pobs, obs[w]), (t, w)
            assert np.array_equal(prew, rew[w, :n])
            assert pgo == bool(go[w])
            assert [info["which_agents_done"][i] for i in range(n)] == list(done[w, :n].astype(bool))
            assert np.all(done[w, n:] == 1) and np.all(rew[w, n:] == 0)
            f64, f32, fl = po.world_to_arrays(wd)
            sl = slice(w * N, (w + 1) * N)
            assert np.array_equal(f64, st.f64[:, sl]) and np.array_equal(fl, st.flags[sl])
            assert np.array_equal(f32, st.f32[:, sl])
    return st


@pytest.mark.parametrize("N,M,sort,nonl", [(4, 3, 0, 0.0), (4, 3, 1, 0.5), (10, 9, 0, 0.3), (10, 4, 1, 0.3),
                                           (6, 7, 2, 0.2), (2, 1, 0, 0.0), (10, 9, 2, 0.0)])
def test_python_and_c_statements_agree_bitwise(N, M, sort, nonl):
    st = _cross(N, M, sort, nonl)
    present = st.flags & po.F_PRESENT != 0
    assert (st.flags[present] & po.F_DONE_MASK != 0).mean() > 0.3      # episodes actually terminate


@pytest.mark.parametrize("N,M,sort,nonl,mode,rvo", [(4, 3, 0, 0.6, 0, 0.6), (4, 3, 0, 0.0, 1, 0.0), (10, 9, 0, 0.5, 1, 0.5),
                                                    (6, 3, 1, 0.7, 1, 0.3)])
def test_python_and_c_agree_on_rvo_agents_and_the_box_generator(N, M, sort, nonl, mode, rvo):
    """GEN v2 (rejection-sampled boxes) and the ORCA policy: both statements, bit for bit, incl. the infeasible-LP branch."""
    st = _cross(N, M, sort, nonl, W=10, steps=120, seed=11, mode=mode, rvo=rvo)
    pol = (st.flags >> po.F_POLICY_SHIFT) & 3
    if rvo > 0:
        assert (pol == po.POLICY_RVO).any()


@pytest.mark.parametrize("switches", [dict(wrap_closed_end=1), dict(done_agents_collide=0), dict(sort_round_gap=0),
                                      dict(sort_tie_lateral=0), dict(wrap_closed_end=1, done_agents_collide=0, sort_round_gap=0,
                                                                     sort_tie_lateral=0)])
def test_python_and_c_agree_with_the_u_switches_flipped(switches):
    """SURVEY App. A U2 (wrap end), U4 (done agents collide with movers), U7 (gap rounding / tie-break) as named switches:
    both statements follow them, bit for bit (scripted + RVO + frozen-network agents in the mix)."""
    for sort in (0, 1, 2):
        _cross(5, 3, sort, 0.5, W=10, steps=90, seed=5, rvo=0.3, frozen=0.2, **switches)


def test_u_switch_known_answers():
    """What each switch changes, on hand-made cases."""
    # U2: an angle exactly on the cut
    assert po.wrap(math.pi) == -math.pi and po.wrap(math.pi, True) == math.pi
    assert po.wrap(-math.pi) == -math.pi and po.wrap(-math.pi, True) == math.pi
    assert po.wrap(3 * math.pi) == -math.pi + 0.0 or abs(po.wrap(3 * math.pi)) <= math.pi
    # U4: agent 1 has timed out (frozen) in agent 0's path
    for collide in (True, False):
        cfg = po.OracleConfig(max_agents=2, max_other_agents_observed=1, done_agents_collide=collide)
        a = po.Agent(0.0, 0.0, 10.0, 0.0, 0.5, 1.0, None, po.POLICY_EXTERNAL, cfg)
        b = po.Agent(1.15, 0.0, 10.0, 5.0, 0.5, 1.0, None, po.POLICY_EXTERNAL, cfg)
        b.ran_out_of_time = True
        w = po.World([a, b], cfg)
        obs, rew, over, info = w.step({0: 2, 1: 2})               # agent 0 moves 0.2 m: gap 1.15 - 0.2 - 1.0 < 0
        if collide:
            assert rew[0] == -0.25 and a.in_collision and rew[1] == -0.25 and b.in_collision
        else:
            assert rew[0] == 0.0 and not a.in_collision and rew[1] == 0.0 and not b.in_collision
        assert obs[0, 1] == 1.0 and obs[0, 6 + 6] < 0.0               # still observed, gap negative
    # U7a: two neighbours 4 mm apart in gap: same centimetre bucket -> lateral decides; exact gaps -> the gap decides
    # U7b: without the lateral criterion the agent index decides
    for round_gap, tie_lat, want in ((True, True, (2, 1)), (False, True, (1, 2)), (True, False, (1, 2))):
        cfg = po.OracleConfig(max_agents=3, max_other_agents_observed=2, sort_round_gap=round_gap, sort_tie_lateral=tie_lat)
        host = po.Agent(0.0, 0.0, 10.0, 0.0, 0.3, 1.0, None, po.POLICY_EXTERNAL, cfg)
        n1 = po.Agent(0.0, 2.004, 5.0, 5.0, 0.3, 1.0, None, po.POLICY_EXTERNAL, cfg)      # farther by 4 mm, lateral +2
        n2 = po.Agent(0.0, -2.0, 5.0, -5.0, 0.3, 1.0, None, po.POLICY_EXTERNAL, cfg)      # lateral -2
        row = po.World([host, n1, n2], cfg).observe()[0]
        # closest_last: far ... near; slot k's p_orth identifies the neighbour
        order = tuple(1 if row[6 + 7 * k + 1] > 0 else 2 for k in range(2))
        assert order == want, (round_gap, tie_lat, order)


def test_frozen_network_agents_take_outside_actions_and_never_count_as_learning():
    cfg = po.OracleConfig(max_agents=2, max_other_agents_observed=1)
    a = po.Agent(0.0, 0.0, 3.0, 0.0, 0.3, 1.0, None, po.POLICY_EXTERNAL, cfg)
    b = po.Agent(0.0, 20.0, 9.0, 20.0, 0.3, 1.0, None, po.POLICY_FROZEN_NET, cfg)
    w = po.World([a, b], cfg)
    obs, rew, over, info = w.step({0: 2, 1: 0})                       # the frozen-network agent turns by -pi/6 at full speed
    assert obs[0, 0] == 1.0 and obs[1, 0] == 0.0
    assert info["which_agents_learning"] == {0: True, 1: False}
    assert abs(b.heading + math.pi / 6) < 1e-6 and abs(b.speed - 1.0) < 1e-12
    assert (b.flags() >> po.F_POLICY_SHIFT) & po.F_POLICY_MASK == po.POLICY_FROZEN_NET
    for _ in range(40):
        obs, rew, over, info = w.step({0: 2, 1: 2})
        if over:
            break
    assert over and a.is_at_goal and not b.is_done                    # TRAIN_MODE: only the learner ends the episode


def test_rvo_agents_avoid_each_other_and_arrive():
    """Behavioural known answer: in random box scenarios most RVO agents reach their goals and few collide -- and they
    collide less often than the same agents running the non-cooperative policy in the SAME scenarios."""
    cfg = po.OracleConfig()
    tally = {po.POLICY_RVO: [0, 0, 0], po.POLICY_NONCOOP: [0, 0, 0]}
    for policy in tally:
        for wd in range(60):
            gen = po.GenConfig(min_agents=3, max_agents=4, mode=1, nonlearning_fraction=1.0, static_fraction=0.0,
                               rvo_fraction=1.0 if policy == po.POLICY_RVO else 0.0)
            world = po.generate_world(9, wd, 0, cfg, gen)
            world.agents[0].policy = policy                      # agent 0 too (the generator keeps it a learner)
            for t in range(400):
                world.step({})
                if all(a.is_done for a in world.agents):
                    break
            for a in world.agents:
                tally[policy][0] += a.is_at_goal
                tally[policy][1] += a.in_collision
                tally[policy][2] += 1
    rvo, blind = tally[po.POLICY_RVO], tally[po.POLICY_NONCOOP]
    assert rvo[0] / rvo[2] > 0.7 and rvo[1] / rvo[2] < 0.1, tally
    assert rvo[1] < 0.5 * blind[1], tally
    cfg = po.OracleConfig(max_agents=6, max_other_agents_observed=5)
    # and the LP falls back to 'least penetration' when the half-planes admit nothing: a host boxed in by 3 approaching agents
    host = po.Agent(0.0, 0.0, 5.0, 0.0, 0.5, 1.0, None, po.POLICY_RVO, cfg)
    others = [po.Agent(1.2 * math.cos(a), 1.2 * math.sin(a), -math.cos(a), -math.sin(a), 0.5, 1.0, None, po.POLICY_EXTERNAL, cfg)
              for a in (0.0, 2.1, 4.2)]
    for o in others:
        o.vel[:] = (-1.5 * o.pos[0], -1.5 * o.pos[1])
    crowd = [host] + others
    lines = po.orca_lines(0, crowd, cfg)
    fail, vx, vy = po._lp_plane(lines, 1.0, 1.0, 0.0, False)
    assert fail < len(lines)                                   # infeasible
    vx, vy = po._lp_least_penetration(lines, fail, 1.0, vx, vy)
    assert math.hypot(vx, vy) <= 1.0 + 1e-9
    act = po.rvo_action(0, crowd, cfg)
    assert 0.0 <= act[0] <= 1.0 + 1e-9 and abs(act[1]) <= math.pi / 6 + 1e-12


def test_box_generator_keeps_its_separations():
    cfg = po.OracleConfig(max_agents=10, max_other_agents_observed=9)
    gen = po.GenConfig(min_agents=2, max_agents=10, mode=1)
    sizes = []
    for wd in range(300):
        world = po.generate_world(3, wd, 0, cfg, gen)
        sizes.append(len(world.agents))
        for i, a in enumerate(world.agents):
            assert np.hypot(*(a.goal - a.pos)) >= 1.0
            lim = 8.0 * 1.01 ** 10
            assert np.all(np.abs(a.pos) <= lim) and np.all(np.abs(a.goal) <= lim)
            for b in world.agents[:i]:
                assert np.hypot(*(a.pos - b.pos)) >= a.radius + b.radius + 0.2
                assert np.hypot(*(a.goal - b.goal)) >= a.radius + b.radius + 0.2
    assert min(sizes) == 2 and max(sizes) == 10


def test_python_and_c_agree_max_turn_rate():
    _cross(4, 3, 0, 0.0, dyn=po.DYN_UNICYCLE_MAX_TURN, steps=40)


def test_action_table():
    t = po.build_action_table()
    assert t.shape == (11, 2)
    assert list(t[:, 0]) == [1.0] * 5 + [0.5] * 3 + [0.0] * 3
    np.testing.assert_allclose(t[:5, 1], np.array([-2, -1, 0, 1, 2]) * math.pi / 12, atol=1e-15)
    np.testing.assert_allclose(t[5:8, 1], np.array([-1, 0, 1]) * math.pi / 6, atol=1e-15)
    c = co.default_cfg(4)
    assert np.array_equal(np.array([[c.actions[r][0], c.actions[r][1]] for r in range(11)]), t)


def test_philox_known_answers():
    # Random123 known-answer vectors for philox4x32-10
    assert po.philox4x32(0, 0, 0, 0, 0, 0) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    f = 0xFFFFFFFF
    assert po.philox4x32(f, f, f, f, f, f) == (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)
    assert po.philox4x32(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0) == (
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)


def _two_agents(cfg, d=10.0, r=0.5, v=1.0):
    a = po.Agent(-d / 2, 0.0, d / 2, 0.0, r, v, None, po.POLICY_EXTERNAL, cfg)
    b = po.Agent(d / 2, 0.0, -d / 2, 0.0, r, v, None, po.POLICY_EXTERNAL, cfg)
    return po.World([a, b], cfg)


def test_head_on_collision_step_index_and_reward():
    cfg = po.OracleConfig(max_agents=2, max_other_agents_observed=1)
    w = _two_agents(cfg, d=10.0, r=0.5, v=1.0)
    # closing speed 2 m/s, gap 9 m, 0.4 m per step -> overlap first at step ceil(9/0.4) = 23
    for k in range(1, 40):
        obs, rew, over, info = w.step({0: 2, 1: 2})
        if k < 22:
            assert not over and np.all(rew == 0.0)
        if k == 22:      # gap = 9 - 8.8 = 0.2 -> getting-close term -0.1 + slope * gap (U5: +0.5, the published sign, is the default)
            gap = obs[0, 6 + 6]
            assert abs(gap - 0.2) < 1e-9 and not over
            if gap <= 0.2:
                np.testing.assert_allclose(rew, -0.1 + cfg.close_penalty_slope * gap, atol=1e-12)
                assert cfg.close_penalty_slope == 0.5
        if over:
            break
    assert k == 23 and np.all(rew == -0.25)
    assert info["which_agents_done"] == {0: True, 1: True}
    # frozen afterwards: zero reward, zero velocity, no time spent
    t_before = [a.t_remaining for a in w.agents]
    obs, rew, over, info = w.step({0: 2, 1: 2})
    assert np.all(rew == 0.0) and over
    assert [a.t_remaining for a in w.agents] == t_before
    assert np.all(obs[:, 6 + 2:6 + 4] == 0.0)


def test_goal_reached_step_count_and_single_reward():
    cfg = po.OracleConfig(max_agents=2, max_other_agents_observed=1)
    # two agents far apart laterally so they never interact
    a = po.Agent(0.0, 0.0, 5.05, 0.0, 0.3, 1.0, None, po.POLICY_EXTERNAL, cfg)
    b = po.Agent(0.0, 50.0, 5.05, 50.0, 0.3, 0.5, None, po.POLICY_EXTERNAL, cfg)
    w = po.World([a, b], cfg)
    hit = {}
    for k in range(1, 200):
        obs, rew, over, info = w.step({0: 2, 1: 2})
        for i in (0, 1):
            if rew[i] == 1.0:
                assert i not in hit
                hit[i] = k
        if over:
            break
    # reach when 5.05 - v*dt*k <= 0.2  ->  k = ceil(4.85 / (v*0.2))  (kept off the exact threshold)
    assert hit == {0: 25, 1: 49}
    obs, rew, over, info = w.step({0: 2, 1: 2})
    assert np.all(rew == 0.0)       # goal reward is paid once


def test_timeout():
    cfg = po.OracleConfig(max_agents=2, max_other_agents_observed=1)
    a = po.Agent(0.0, 0.0, 5.05, 0.0, 0.3, 1.0, None, po.POLICY_EXTERNAL, cfg)
    b = po.Agent(0.0, 50.0, 5.05, 50.0, 0.3, 1.0, None, po.POLICY_EXTERNAL, cfg)
    w = po.World([a, b], cfg)
    budget = 2.0 * (5.05 - 0.2) / 1.0        # 9.7 s = 48.5 steps (kept off the exact threshold)
    assert a.t_remaining == pytest.approx(budget)
    for k in range(1, 200):        # action 9 = zero speed: never arrives
        obs, rew, over, info = w.step({0: 9, 1: 9})
        if over:
            break
    assert k == 49 == math.ceil(budget / 0.2)
    assert all(ag.ran_out_of_time and not ag.is_at_goal for ag in w.agents)


def test_time_budget_switch_u11():
    """U11: MAX_TIME_RATIO * (dist - NEAR_GOAL_THRESHOLD) / pref_speed (default, upstream agent.py as recalled) against
    SURVEY App. A's MAX_TIME_RATIO * dist / pref_speed; never below one DT; Python and C statements agree bitwise."""
    for edge, budget, steps in ((True, 2.0 * (5.05 - 0.2), 49), (False, 2.0 * 5.05, 51)):
        cfg = po.OracleConfig(max_agents=2, max_other_agents_observed=1, time_budget_from_goal_edge=edge)
        a = po.Agent(0.0, 0.0, 5.05, 0.0, 0.3, 1.0, None, po.POLICY_EXTERNAL, cfg)
        b = po.Agent(0.0, 50.0, 5.05, 50.0, 0.3, 1.0, None, po.POLICY_EXTERNAL, cfg)
        assert a.t_remaining == pytest.approx(budget, abs=1e-12)
        w = po.World([a, b], cfg)
        for k in range(1, 200):
            if w.step({0: 9, 1: 9})[2]:
                break
        assert k == steps == math.ceil(budget / 0.2)
        near = po.Agent(0.0, 0.0, 0.21, 0.0, 0.3, 1.0, None, po.POLICY_EXTERNAL, cfg)
        assert near.t_remaining == (0.2 if edge else pytest.approx(0.42))          # floor of one DT
        # generated worlds: the C statement makes the same choice, bit for bit
        ccfg = co.default_cfg(4, time_budget_from_goal_edge=int(edge))
        gen = po.GenConfig(min_agents=2, max_agents=4)
        cgen = co.default_gen(2, 4)
        st = co.State.empty(50, 4)
        co.generate(ccfg, cgen, 77, st, np.zeros(50, np.uint32))
        pcfg = po.OracleConfig(time_budget_from_goal_edge=edge)
        for wd in range(50):
            f64, _, _ = po.world_to_arrays(po.generate_world(77, wd, 0, pcfg, gen))
            assert np.array_equal(f64[3], st.f64[3, 4 * wd:4 * wd + 4])


def test_obs_layout_and_sorting():
    cfg = po.OracleConfig(max_agents=4, max_other_agents_observed=3)
    host = po.Agent(0.0, 0.0, 10.0, 0.0, 0.5, 1.0, None, po.POLICY_EXTERNAL, cfg)
    near = po.Agent(2.0, 0.0, -10.0, 0.0, 0.5, 1.0, None, po.POLICY_EXTERNAL, cfg)
    far = po.Agent(0.0, 6.0, 0.0, -10.0, 0.25, 1.0, None, po.POLICY_NONCOOP, cfg)
    w = po.World([host, near, far], cfg)
    obs = w.observe()
    assert obs.shape == (4, 27)
    assert obs[0, 0] == 1.0 and obs[2, 0] == 0.0 and np.all(obs[3] == 0.0)
    assert obs[0, 1] == 2 and obs[0, 2] == 10.0 and obs[0, 3] == 0.0 and obs[0, 4] == 1.0 and obs[0, 5] == 0.5
    # closest_last: far agent in slot 0, near agent in slot 1 (the last filled), slot 2 zero
    np.testing.assert_allclose(obs[0, 6:13], [0.0, 6.0, 0.0, 0.0, 0.25, 0.75, 5.25])
    np.testing.assert_allclose(obs[0, 13:20], [2.0, 0.0, 0.0, 0.0, 0.5, 1.0, 1.0])
    assert np.all(obs[0, 20:] == 0.0)
    cfg2 = po.OracleConfig(max_agents=4, max_other_agents_observed=3, sort_method=po.SORT_CLOSEST_FIRST)
    for ag in w.agents:
        ag.cfg = cfg2
    obs2 = po.World(w.agents, cfg2).observe()
    np.testing.assert_allclose(obs2[0, 6:13], obs[0, 13:20])
    np.testing.assert_allclose(obs2[0, 13:20], obs[0, 6:13])
    # clipping to M=1 keeps the closest
    cfg3 = po.OracleConfig(max_agents=4, max_other_agents_observed=1)
    for ag in w.agents:
        ag.cfg = cfg3
    obs3 = po.World(w.agents, cfg3).observe()
    assert obs3.shape == (4, 13) and obs3[0, 1] == 1
    np.testing.assert_allclose(obs3[0, 6:13], obs[0, 13:20])


def test_worlds_are_independent_and_permutation_equivariant():
    N = 4
    cfg = co.default_cfg(N)
    gen = co.default_gen(4, 4)
    W = 16
    st = co.State.empty(W, N)
    co.generate(cfg, gen, 3, st, np.zeros(W, np.uint32))
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 11, size=(W, N))
    full = st.copy()
    obs, rew, done, go = co.step(cfg, full, acts)
    for w in range(W):                              # batch of W == W single-world runs
        one = co.State(st.f64[:, w * N:(w + 1) * N].copy(), st.f32[:, w * N:(w + 1) * N].copy(), st.flags[w * N:(w + 1) * N].copy())
        o1, r1, d1, g1 = co.step(cfg, one, acts[w:w + 1])
        assert np.array_equal(o1[0], obs[w]) and np.array_equal(r1[0], rew[w]) and g1[0] == go[w]
    perm = np.array([2, 0, 3, 1])                   # permuting agents permutes rows
    w = 5
    sl = np.arange(w * N, (w + 1) * N)[perm]
    one = co.State(st.f64[:, sl].copy(), st.f32[:, sl].copy(), st.flags[sl].copy())
    o1, r1, d1, g1 = co.step(cfg, one, acts[w:w + 1, perm])
    assert np.array_equal(r1[0], rew[w][perm]) and np.array_equal(d1[0], done[w][perm])
    np.testing.assert_allclose(o1[0][:, :6], obs[w][perm][:, :6], rtol=0, atol=0)


def test_c_oracle_reproduces_committed_env_golden():
    """tests/golden/env_golden.npz was generated by the reference-STYLE Python oracle (parity unpinned: it is a
    regression anchor, not a reference output).  The C oracle must reproduce it: observations and rewards are
    stored in float64 and compared EXACTLY (the two statements are the same arithmetic), flags bit for bit."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "env_golden.npz"))
    for name in g["cases"]:
        N, M, sort, gmin, W, steps, seed = [int(v) for v in g[name + "_cfg"]]
        cfg = co.default_cfg(N, M, sort_method=sort)
        st = co.State(g[name + "_f64"].copy(), g[name + "_f32"].copy(), g[name + "_flags0"].copy())
        for t in range(steps):
            obs, rew, done, go = co.step(cfg, st, g[name + "_actions"][t])
            assert g[name + "_obs"].dtype == np.float64 and np.array_equal(obs, g[name + "_obs"][t]), (name, t)
            assert np.array_equal(rew, g[name + "_rew"][t]), (name, t)
            assert np.array_equal(done, g[name + "_done"][t]) and np.array_equal(go, g[name + "_over"][t])
            assert np.array_equal(st.flags.reshape(W, N), g[name + "_flags"][t])
