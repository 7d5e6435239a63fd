"""Shared checker code of the GPU parity tests (test infrastructure): the float64 env oracle configured from the same keyword
arguments a test hands to ``BatchedCollisionAvoidanceEnv``, and the replay of recorded per-step inputs through
``oracle/rollout_oracle.run_episode`` (itself bit-pinned to the reference's own ``ProcessAgent``,
/root/reference/ga3c/GA3C/ProcessAgent.py:105-211) against the rows a device-side rollout emitted."""
import os

import numpy as np

from oracle import c_oracle as co
from oracle import rollout_oracle as ro

R_TOL = 1e-6      # n-step returns: float64 on both sides, emitted as float32

# cavoid_cfg field -> OracleGen field
_GEN = {"gen_min_agents": "min_agents", "gen_nonlearning_fraction": "nonlearning_fraction", "gen_static_fraction": "static_fraction",
        "gen_rvo_fraction": "rvo_fraction", "gen_frozen_fraction": "frozen_fraction", "gen_mode": "mode", "gen_pool_size": "pool_size",
        "gen_goal_jitter": "goal_jitter", "gen_angle_jitter": "angle_jitter", "gen_min_trip": "min_trip"}
_SKIP = {"rvo_enabled"}          # env-side resource switches the oracle has no use for


def oracle_for(N, M=None, pool=65536, **over):
    """(OracleCfg, OracleGen) mirroring ``BatchedCollisionAvoidanceEnv(W, Cfg(N, M), **over)`` (pool: cavoid_default_cfg's
    gen_pool_size unless ``gen_pool_size`` is among the overrides)."""
    gen_kw = {"pool_size": pool}
    cfg_kw = {}
    for k, v in over.items():
        if k in _GEN:
            gen_kw[_GEN[k]] = v
        elif k not in _SKIP:
            cfg_kw[k] = v
    gen_kw.setdefault("min_agents", N)
    return co.default_cfg(N, N - 1 if M is None else M, **cfg_kw), co.default_gen(max_agents=N, **gen_kw)


def obs_diff(obs, oobs, heading_col=3):
    """|obs - oracle obs| with the ego heading compared on the circle (its branch cut at +-pi sits exactly where an agent
    that has just run over its goal centre is: a 1-ulp atan2 difference turns -pi into +pi)."""
    d = np.abs(np.asarray(obs, np.float64) - oobs)
    d[..., heading_col] = np.minimum(d[..., heading_col], np.abs(d[..., heading_col] - 2.0 * np.pi))
    return d


def replay_rollout(rec, rows, episodes, reflush, gamma, t_max):
    """rec[t] = (obs [W,N,1+D] the policy acted on at step t, actions [W,N], values [W,N], rewards [W,N], done bool [W,N],
    game_over bool [W]); rows = (x, r, a_index, src) of the batch the device handed to the trainer (src: world, agent,
    recorded-at, emitted-at); episodes = the device's episode log [k,3] or None.  Replays every FINISHED episode of every
    world through the rollout oracle and asserts that exactly its rows were emitted, bit-exact states and actions, returns
    to R_TOL.  Returns the number of rows matched."""
    x, r, a, src = rows
    steps, W = len(rec), rec[0][0].shape[0]
    got = {}
    for k in range(len(r)):
        got.setdefault(tuple(int(v) for v in src[k]), []).append(k)
    expect_rows, expect_eps = 0, []
    for w in range(W):
        start = 0
        for t in range(steps):
            if not rec[t][5][w]:
                continue
            ts = list(range(start, t + 1))
            obs_seq = np.stack([rec[k][0][w] for k in ts] + [rec[t][0][w]])   # last entry unused by the oracle
            learning = obs_seq[0][:, 0] > 0.5
            present = np.flatnonzero(obs_seq[0][:, 4] > 0)
            n_present = int(present.max()) + 1 if len(present) else 0
            rewards = np.stack([rec[k][3][w] for k in ts]).astype(np.float64)
            done = np.stack([rec[k][4][w] for k in ts])
            actions = np.stack([rec[k][1][w] for k in ts])
            values = np.stack([rec[k][2][w] for k in ts]).astype(np.float64)
            chunks = ro.run_episode(obs_seq.astype(np.float64), rewards, done, learning, n_present, actions, values, gamma, t_max)
            if not reflush:          # cleaned mode: drop what a done-and-trained agent would re-flush
                trained_at, kept = {}, []
                for c in chunks:
                    if c.agent in trained_at and c.emitted_t > trained_at[c.agent]:
                        continue
                    kept.append(c)
                    if done[c.emitted_t, c.agent]:
                        trained_at.setdefault(c.agent, c.emitted_t)
                chunks = kept
            total_reward, total_length = 0.0, 0
            for c in chunks:
                emitted = start + c.emitted_t
                for row, tl in enumerate(c.t):
                    key = (w, c.agent, start + tl, emitted)
                    assert key in got and got[key], (key, "missing row")
                    k = got[key].pop(0)
                    assert np.array_equal(x[k], c.x[row].astype(np.float32)), key
                    assert abs(r[k] - c.r[row]) <= R_TOL, (key, r[k], c.r[row])
                    assert a[k] == int(np.argmax(c.a[row])), key
                    expect_rows += 1
                total_reward += c.score
                total_length += len(c.r) + 1
            expect_eps.append((w, total_reward, total_length))
            start = t + 1
    # everything the device emitted for finished episodes was expected (rows of unfinished episodes remain)
    leftover = sum(len(v) for v in got.values())
    assert expect_rows + leftover == len(r)
    finished_until = {w: max([t for t in range(steps) if rec[t][5][w]], default=-1) for w in range(W)}
    for key, ks in got.items():
        if ks:
            assert key[3] > finished_until[key[0]], ("unexpected row", key)
    if episodes is not None and reflush:          # (cleaned mode logs the episodes too, with the cleaned chunks' lengths)
        assert len(episodes) == len(expect_eps)
        dev = sorted((int(e[0]), round(float(e[2]))) for e in episodes)
        assert dev == sorted((w, tl) for w, _, tl in expect_eps)
        np.testing.assert_allclose(sorted(float(e[1]) for e in episodes), sorted(tr for _, tr, _ in expect_eps), atol=1e-4)
    return expect_rows


def subset_worlds(rec, rows, episodes, worlds):
    """replay_rollout's inputs restricted to `worlds` (sorted world indices), renumbered 0..len-1: for batches too large to replay world
    by world in Python"""
    worlds = np.asarray(worlds)
    index = -np.ones(rec[0][0].shape[0], np.int64)
    index[worlds] = np.arange(len(worlds))
    rec = [tuple(v[worlds] for v in r) for r in rec]
    x, r, a, src = rows
    keep = index[src[:, 0]] >= 0
    src = src[keep].copy()
    src[:, 0] = index[src[:, 0]]
    if episodes is not None:
        ek = index[episodes[:, 0].astype(np.int64)] >= 0
        episodes = episodes[ek].copy()
        episodes[:, 0] = index[episodes[:, 0].astype(np.int64)]
    return rec, (x[keep], r[keep], a[keep], src), episodes


# ---- ties of a scripted policy vs real divergences -------------------------------------------------------------------------
def _explain(ocfg, ogen, seed, w, N, pre_s, pre_e, a, s, e, h, spread, eps, trials):
    """Diagnostics of a divergence the classifier calls real (CAVOID_STRESS_EXPLAIN=1): per agent, how far HIP's post-step state is from
    the oracle's; the oracle's action BEFORE its float32 cast (the step re-run with actions_fp32 = 0) and how close it sits to a
    float32 rounding boundary; how far +-eps moves that un-cast action (the conditioning of the agent's programme)."""
    h64, h32, hfl, hep = h
    pol = (pre_s.flags >> 8) & 7
    print("      flags equal %s, episode equal %s; oracle's own spread under +-%g: %.3g" % (np.array_equal(hfl, s.flags), np.array_equal(hep, e), eps, spread))
    o2 = type(ocfg).from_buffer_copy(ocfg)
    o2.actions_fp32 = 0
    s64, e64 = pre_s.copy(), pre_e.copy()
    co.step_autoreset(o2, ogen, seed, s64, e64, a, world_offset=w)
    wrap = lambda x: (x + np.pi) % (2 * np.pi) - np.pi
    prng = np.random.default_rng(2)
    moved = np.zeros(N)
    for _ in range(trials):
        s2, e2 = pre_s.copy(), pre_e.copy()
        s2.f64[0] += prng.choice([-eps, eps], N); s2.f64[1] += prng.choice([-eps, eps], N); s2.f64[2] += prng.choice([-eps, 0.0, eps], N)
        co.step_autoreset(o2, ogen, seed, s2, e2, a, world_offset=w)
        moved = np.maximum(moved, np.abs(wrap(s2.f64[2] - s64.f64[2])))
    for j in range(N):
        d = [float(h64[0, j] - s.f64[0, j]), float(h64[1, j] - s.f64[1, j]), float(wrap(h64[2, j] - s.f64[2, j]))]
        if max(abs(x) for x in d) == 0.0:
            continue
        a1 = float(wrap(s64.f64[2, j] - pre_s.f64[2, j]))                       # the oracle's heading change, not cast to float32
        f = np.float32(a1)
        other = np.nextafter(f, np.float32(np.inf if a1 > float(f) else -np.inf))
        mid, ulp = 0.5 * (float(f) + float(other)), abs(float(other) - float(f))
        a1_o, a1_h = float(wrap(s.f64[2, j] - pre_s.f64[2, j])), float(wrap(h64[2, j] - pre_s.f64[2, j]))
        print("      agent %d policy %d: HIP - oracle  dpx %.3g dpy %.3g dheading %.3g;  heading change: oracle un-cast %.17g, oracle %.9g, HIP %.9g "
              "(float32 ulp %.3g; un-cast value %.3g ulp from the rounding boundary; +-eps moves it by %.3g rad)"
              % (j, pol[j], d[0], d[1], d[2], a1, a1_o, a1_h, ulp, abs(a1 - mid) / ulp, moved[j]))


def classify_divergence(make_env, ocfg, ogen, seed, N, w, hip0, st0, ep0, acts_w, trials=63, eps=1e-13, pos_tol=1e-9, history=None):
    """A world whose HIP results left the oracle's: is it a TIE of the ORCA linear programme (the optimal velocity jumps between
    two vertices of the feasible region when the positions move by the ~1e-13 m the two transcendental libraries differ by anyway),
    or a real difference between the two implementations?  Decided by code, not by hand:

    world ``w`` is replayed alone, one step per launch, on both sides from their own states at the start of the launch that
    showed the mismatch (``hip0`` = (f64 [4,N], f32 [5,N], flags [N], episode) of the HIP side, ``st0`` / ``ep0`` the oracle's
    whole-batch state) over the launch's actions ``acts_w`` [n, N].  At the first step after which the two world states differ
    (a flag bit, the episode counter, or a position by more than ``pos_tol``) the ORACLE is re-run from its own pre-step state
    with that world's positions perturbed by +-``eps`` (``trials`` random sign patterns).  Verdict:
      ("tie", step)   an agent whose action passes through atan2 was running in the world -- ORCA (policy 3) or non-cooperative
                      (policy 2: straight at the goal, a1 = -heading_ego) -- AND one of the oracle's perturbed answers is HIP's
                      answer (every flag bit, the episode counter, positions / headings to pos_tol).  Besides the ORCA programme's
                      vertex jumps this covers the other thing 1e-13 m decides: such an action is cast to float32
                      (``actions_fp32``), and when its float64 value sits within an ulp of a float32 rounding boundary the two
                      atan2 implementations (<= 1 ulp apart) round it to neighbouring floats -- a 6e-8 rad step in the heading
                      -- and the continuous cousin of the vertex jump: two ORCA lines nearly parallel, whose intersection
                      amplifies the libraries' 1e-16 to 1e-8 (seen once in ~1.8 G agent-steps: parity stress pass 5, N = 4 box
                      scenarios, world 161, the round-3 kernels too): HIP's answer is then none of the oracle's perturbed answers
                      but lies INSIDE their spread (same flags, same episode, no further from the oracle than +-eps moves the
                      oracle itself);
                      -- or, with ``history`` (the oracle's whole-batch states and the actions of the launches BEFORE this one,
                      oldest first: [(st, ep, acts [n, W, N]), ...]), inside the spread of the oracle's own trajectories started
                      ``eps`` apart at the oldest of those launches and rolled through the same actions to the diverging step:
                      a DRIFT -- several moderately ill-conditioned steps in a row (each amplifying by 10 .. 100) carry the
                      libraries' 1e-16 past ``pos_tol`` without any single step being a jump; the one-step test then sees
                      two pre-step states that already differ by 1e-10 and an innocent step (seen twice in 5.9 G agent-steps:
                      stress passes 30 .. 59, N = 10 box scenarios with ORCA agents, seeds 844 / 851);
      ("real", step)  anything else -- including a world without such an agent, whatever the perturbations say;
      ("none", -1)    the replay shows no divergence (the batch-level mismatch was not reproduced: treat as real).
    ``make_env(num_worlds, world_offset)`` builds a HIP env configured like the one under test."""
    import torch
    sl = slice(w * N, (w + 1) * N)
    one = make_env(1, w)
    try:
        f64, f32, fl, ep_h = hip0
        one.seed(seed, torch.tensor([int(ep_h)], dtype=torch.int64).to(torch.int32).to(one.device))
        one.set_state(torch.from_numpy(np.ascontiguousarray(f64)).to(one.device), torch.from_numpy(np.ascontiguousarray(f32)).to(one.device),
                      torch.from_numpy(np.ascontiguousarray(fl).view(np.int32)).to(one.device))
        s = co.State(st0.f64[:, sl].copy(), st0.f32[:, sl].copy(), st0.flags[sl].copy())
        e = np.array([ep0[w]], np.uint32)

        def hip_state():
            h64, h32, hfl = [v.cpu().numpy() for v in one.get_state()]
            return h64, h32, hfl.view(np.uint32), one.episode.cpu().numpy().view(np.uint32)

        def same(h, s2, e2):
            h64, h32, hfl, hep = h
            return (np.array_equal(hfl, s2.flags) and np.array_equal(hep, e2) and np.array_equal(h32[:4], s2.f32[:4])
                    and float(np.abs(h64[:3] - s2.f64[:3]).max()) <= pos_tol)

        for k in range(len(acts_w)):
            a = np.ascontiguousarray(acts_w[k], np.int32).reshape(1, N)
            pre_s, pre_e = s.copy(), e.copy()
            one.step_autoreset(torch.from_numpy(a).to(one.device))
            co.step_autoreset(ocfg, ogen, seed, s, e, a, world_offset=w)
            h = hip_state()
            if same(h, s, e):
                continue
            pol = (pre_s.flags >> 8) & 7
            running_scripted = ((pol == 3) | (pol == 2)) & (pre_s.flags & 0x20 != 0) & (pre_s.flags & 7 == 0)
            if not running_scripted.any():
                return "real", k
            prng = np.random.default_rng(1)
            spread = 0.0                                      # how far +-eps moves the ORACLE's own answer
            for _ in range(trials):
                s2, e2 = pre_s.copy(), pre_e.copy()
                s2.f64[0] += prng.choice([-eps, eps], N)
                s2.f64[1] += prng.choice([-eps, eps], N)
                s2.f64[2] += prng.choice([-eps, 0.0, eps], N)      # (headings too: the two sincos differ by an ulp as well)
                co.step_autoreset(ocfg, ogen, seed, s2, e2, a, world_offset=w)
                if same(h, s2, e2):
                    return "tie", k
                if np.array_equal(s2.flags, s.flags) and np.array_equal(e2, e):
                    spread = max(spread, float(np.abs(s2.f64[:3] - s.f64[:3]).max()))
            # an ILL-CONDITIONED programme instead of a vertex jump (two ORCA lines nearly parallel: the intersection amplifies
            # 1e-16 to 1e-8 and beyond): no perturbed answer IS HIP's, but HIP's lies inside the cloud the oracle itself spreads
            # over under +-eps -- same flags, same episode, and no further from the oracle than the oracle is from itself
            h64, h32, hfl, hep = h
            if (np.array_equal(hfl, s.flags) and np.array_equal(hep, e) and np.array_equal(h32[:4], s.f32[:4])
                    and float(np.abs(h64[:3] - s.f64[:3]).max()) <= spread):
                return "tie", k
            # a DRIFT: the oracle against itself over the launches before this one (see the docstring)
            if history and np.array_equal(hfl, s.flags) and np.array_equal(hep, e) and np.array_equal(h32[:4], s.f32[:4]):
                def roll(hist, perturb):
                    st_h, ep_h0, _ = hist[0]
                    s2 = co.State(st_h.f64[:, sl].copy(), st_h.f32[:, sl].copy(), st_h.flags[sl].copy())
                    e2 = np.array([ep_h0[w]], np.uint32)
                    if perturb:
                        s2.f64[0] += prng.choice([-eps, eps], N)
                        s2.f64[1] += prng.choice([-eps, eps], N)
                        s2.f64[2] += prng.choice([-eps, 0.0, eps], N)
                    for _, _, acts_l in hist:
                        for kk in range(acts_l.shape[0]):
                            co.step_autoreset(ocfg, ogen, seed, s2, e2, np.ascontiguousarray(acts_l[kk, w], np.int32).reshape(1, N), world_offset=w)
                    for kk in range(k + 1):
                        co.step_autoreset(ocfg, ogen, seed, s2, e2, np.ascontiguousarray(acts_w[kk], np.int32).reshape(1, N), world_offset=w)
                    return s2, e2
                # the longest stretch of history from which the oracle reproduces its own state bit for bit (the stress re-synchronises
                # the oracle's copy of a world to HIP's after a tie: a stretch across such a point is not the oracle's trajectory)
                for j in range(len(history)):
                    s2, e2 = roll(history[j:], False)
                    if np.array_equal(s2.f64, s.f64) and np.array_equal(s2.flags, s.flags) and np.array_equal(e2, e):
                        far = 0.0
                        for _ in range(trials):
                            s2, e2 = roll(history[j:], True)
                            if np.array_equal(s2.flags, s.flags) and np.array_equal(e2, e):
                                far = max(far, float(np.abs(s2.f64[:3] - s.f64[:3]).max()))
                        if float(np.abs(h64[:3] - s.f64[:3]).max()) <= far:
                            return "tie", k
                        spread = max(spread, far)
                        break
            if os.environ.get("CAVOID_STRESS_EXPLAIN"):
                _explain(ocfg, ogen, seed, w, N, pre_s, pre_e, a, s, e, h, spread, eps, trials)
            return "real", k
        return "none", -1
    finally:
        one.close()
