"""GPU parity of the batched rollout (HIP, through the C ABI) against the rollout oracle, which is
itself pinned bit-for-bit to the reference's ProcessAgent (tests/test_rollout_oracle.py).  The real
GPU env drives it; the per-step inputs are recorded and replayed, world by world and episode by
episode, through oracle/rollout_oracle.run_episode."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import rollout_oracle as ro

pytestmark = pytest.mark.gpu

R_TOL = 1e-6      # n-step returns: float64 on both sides, emitted as float32


def _make(W, N, seed, **over):
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    from rl_collision_avoidance_amd.config import EnvConfig

    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
            EnvConfig.__init__(self)
    return BatchedCollisionAvoidanceEnv(W, Cfg(), seed=seed, **over)


@pytest.mark.parametrize("fuse_env_push", ["1", "0"])      # env.step + bookkeeping as ONE launch (cavoid_step_push, the default) / as three
@pytest.mark.parametrize("N,gen_min,nonl,reflush", [(4, 2, 0.3, True), (4, 4, 0.0, False), (10, 2, 0.2, True)])
def test_rollout_matches_oracle(N, gen_min, nonl, reflush, fuse_env_push, monkeypatch):
    monkeypatch.setenv("CAVOID_FUSE_ENV_PUSH", fuse_env_push)
    from rl_collision_avoidance_amd.ga3c.rollout import BatchedRollout
    W, steps, seed, T_MAX, GAMMA = 96, 260, 3, 20, 0.97
    env = _make(W, N, seed, gen_min_agents=gen_min, gen_nonlearning_fraction=nonl)
    roll = BatchedRollout(env, policy=None, time_max=T_MAX, discount=GAMMA, reflush_done=reflush, ring_len=steps + 8,
                          dup_capacity=1000000)
    roll.reset()
    rng = np.random.default_rng(seed)
    rec = []
    for t in range(steps):
        obs = roll.obs.cpu().numpy().copy()
        acts = rng.integers(0, 11, size=(W, N)).astype(np.int32)
        acts[rng.random((W, N)) < 0.7] = 2
        vals = np.round(rng.normal(0, 0.5, size=(W, N)), 3).astype(np.float32)
        rew, done, over = roll.step(torch.from_numpy(acts).cuda(), torch.from_numpy(vals).cuda())
        rec.append((obs, acts, vals, rew.cpu().numpy().copy(), done.cpu().numpy().astype(bool), over.cpu().numpy().astype(bool)))
    batch = roll.drain(flush_all=True)
    episodes = roll.drain_episodes().cpu().numpy()
    assert batch.dropped == 0 and len(batch) > W * 20
    x, r, a, src = [v.cpu().numpy() for v in (batch.x, batch.r, batch.a_index, batch.src)]
    got = {}
    for k in range(len(r)):
        got.setdefault(tuple(src[k]), []).append(k)          # (world, agent, recorded-at, emitted-at)

    # replay per world, episode by episode
    expect_rows, expect_eps = 0, []
    for w in range(W):
        start = 0
        for t in range(steps):
            if not rec[t][5][w]:
                continue
            ts = list(range(start, t + 1))
            obs_seq = np.stack([rec[k][0][w] for k in ts] + [rec[t][0][w]])   # last entry unused by the oracle
            learning = obs_seq[0][:, 0] > 0.5
            n_present = int(np.flatnonzero(obs_seq[0][:, 4] > 0).max()) + 1
            rewards = np.stack([rec[k][3][w] for k in ts]).astype(np.float64)
            done = np.stack([rec[k][4][w] for k in ts])
            actions = np.stack([rec[k][1][w] for k in ts])
            values = np.stack([rec[k][2][w] for k in ts]).astype(np.float64)
            chunks = ro.run_episode(obs_seq.astype(np.float64), rewards, done, learning, n_present, actions, values, GAMMA, T_MAX)
            if not reflush:          # cleaned mode: drop what a done-and-trained agent would re-flush
                trained_at, kept = {}, []
                for c in chunks:
                    if c.agent in trained_at and c.emitted_t > trained_at[c.agent]:
                        continue                      # a re-flush of an already trained agent
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
            if reflush:
                expect_eps.append((w, total_reward, total_length))
            start = t + 1
    # everything the device emitted for finished episodes was expected (rows of unfinished episodes remain)
    leftover = sum(len(v) for v in got.values())
    assert expect_rows + leftover == len(r)
    finished_until = {w: max([t for t in range(steps) if rec[t][5][w]], default=-1) for w in range(W)}
    for key, ks in got.items():
        if ks:
            assert key[3] > finished_until[key[0]], ("unexpected row", key)
    if reflush:
        assert len(episodes) == len(expect_eps)
        dev = sorted((int(e[0]), round(float(e[2]))) for e in episodes)
        assert dev == sorted((w, tl) for w, _, tl in expect_eps)
        np.testing.assert_allclose(sorted(float(e[1]) for e in episodes), sorted(tr for _, tr, _ in expect_eps), atol=1e-4)
    roll.close()
    env.close()


def test_graph_replay_equals_eager_steps():
    """A hipGraph of closed-loop steps replays to the same state / rows as stepping eagerly."""
    from rl_collision_avoidance_amd.ga3c.network import NetworkVP_rnn
    from rl_collision_avoidance_amd.ga3c.rollout import BatchedRollout
    W, N = 512, 4
    outs = []
    for graphed in (False, True):
        env = _make(W, N, 5)
        net = NetworkVP_rnn(env.config, seed=1).cuda()
        roll = BatchedRollout(env, net.predict_p_and_v, greedy=True, reflush_done=False, ring_len=64)   # argmax: no RNG in the loop
        roll.reset()
        if graphed:
            roll.capture(steps_per_graph=4)             # 2 warm-up steps happen here
            roll.replay(12)
        else:
            for _ in range(2 + 48):
                roll.step()
        b = roll.drain(flush_all=True)
        assert b.dropped == 0 and roll.lost_blocks == 0
        order = torch.argsort(b.src[:, 0].long() * 10**9 + b.src[:, 1].long() * 10**7 + b.src[:, 3].long() * 10**3 + b.src[:, 2].long() % 1000)
        outs.append((roll.obs.clone(), [t.clone() for t in env.get_state()], b.x[order], b.r[order], b.a_index[order], roll.step_index))
        roll.close(); env.close()
    (o0, s0, x0, r0, a0, n0), (o1, s1, x1, r1, a1, n1) = outs
    assert n0 == n1 == 50
    assert torch.equal(o0, o1) and all(torch.equal(u, v) for u, v in zip(s0, s1))
    assert torch.equal(x0, x1) and torch.equal(r0, r1) and torch.equal(a0, a1) and len(r0) > 0


def test_rollout_with_rvo_and_frozen_network_agents_on_box_scenarios():
    """SURVEY section 8f-N3 reached from the GA3C loop: box scenarios generated inside the step, a scripted mix of static / RVO /
    frozen-network / non-cooperative agents.  The frozen-network agents act by THEIR network's argmax (checked row by row against
    a direct forward pass of the frozen weights), never record experiences, and the step-by-step loop equals its hipGraph and the
    fused actor kernel (cavoid_actor_run_mix: the learner's and the frozen network's passes inside one launch)."""
    from rl_collision_avoidance_amd import _lib
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    from rl_collision_avoidance_amd.config import EnvConfig
    from rl_collision_avoidance_amd.ga3c.network import NetworkVP_rnn
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
    from rl_collision_avoidance_amd.ga3c.rollout import BatchedRollout
    W, N = 384, 4

    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
            self.TEST_CASE_GENERATOR = "box"
            self.SCRIPTED_AGENT_FRACTION = 0.7
            self.SCRIPTED_STATIC_FRACTION = 0.2
            self.SCRIPTED_RVO_FRACTION = 0.3
            self.SCRIPTED_FROZEN_NET_FRACTION = 0.3
            EnvConfig.__init__(self)
    outs = []
    for graphed in (False, True, "fused"):
        cfg = Cfg()
        env = BatchedCollisionAvoidanceEnv(W, cfg, seed=9, gen_min_agents=2, gen_pool_size=0)
        assert env.cfg.rvo_enabled == 1 and env.cfg.gen_mode == 1
        net, frozen_net = NetworkVP_rnn(cfg, seed=1).cuda(), NetworkVP_rnn(cfg, seed=2).cuda()
        pol, frozen = FusedPolicy(net, seed=5), FusedPolicy(frozen_net, seed=0)
        with pytest.raises(ValueError):
            BatchedRollout(env, pol)                          # the env generates frozen-network agents: their network is required
        roll = BatchedRollout(env, pol, reflush_done=False, frozen_policy=frozen, ring_len=64)
        assert roll.fused_available and "cavoid_actor_run_mix" in roll.actor_path    # the whole mix inside the fused launch (round 4)
        roll.reset()
        if graphed == "fused":
            roll.run_fused(2)
            for _ in range(10):
                roll.run_fused(4)
        elif not graphed:
            # one step by hand: the frozen rows carry the frozen network's argmax, the others the learner's draw
            obs = roll.obs.clone()
            flags = env.get_state()[2].view(W, N)
            pol4 = ((flags >> _lib.F_POLICY_SHIFT) & _lib.F_POLICY_MASK) == _lib.POLICY_FROZEN_NET
            assert pol4.sum().item() > 20 and ((flags >> 8) & 7 == 3).sum().item() > 20
            actions, _ = roll.act(obs)
            with torch.no_grad():
                p_frozen = frozen_net.predict_p_and_v(obs.view(W * N, -1)[:, 1:].contiguous())[0].view(W, N, -1)
            top2 = p_frozen.topk(2, dim=-1).values
            clear = pol4 & ((top2[..., 0] - top2[..., 1]) > 1e-4)          # (skip numerical near-ties between the two kernels)
            assert torch.equal(actions[clear].long(), p_frozen.argmax(dim=-1)[clear])
            assert (obs[..., 0][pol4] == 0).all()
            # (that act() consumed one policy launch; start over so that both runs see the same random stream)
            roll.close(); env.close()
            env = BatchedCollisionAvoidanceEnv(W, cfg, seed=9, gen_min_agents=2, gen_pool_size=0)
            pol, frozen = FusedPolicy(net, seed=5), FusedPolicy(frozen_net, seed=0)
            roll = BatchedRollout(env, pol, reflush_done=False, frozen_policy=frozen, ring_len=64)
            roll.reset()
            for _ in range(2 + 40):
                roll.step()
        else:
            roll.capture(steps_per_graph=4)
            roll.replay(10)
        b = roll.drain(flush_all=True)
        assert b.dropped == 0 and roll.lost_blocks == 0 and len(b) > 0
        order = torch.argsort(b.src[:, 0].long() * 10**9 + b.src[:, 1].long() * 10**7 + b.src[:, 3].long() * 10**3 + b.src[:, 2].long() % 1000)
        outs.append((roll.obs.clone(), [t.clone() for t in env.get_state()], b.x[order], b.r[order], b.a_index[order], env.episode.clone()))
        # nothing a scripted agent did became a training row
        flags = env.get_state()[2]
        roll.close(); env.close()
    (o0, s0, x0, r0, a0, e0) = outs[0]
    for (o1, s1, x1, r1, a1, e1) in outs[1:]:               # eager steps == their hipGraph == the fused actor kernel, bitwise
        assert torch.equal(o0, o1) and all(torch.equal(u, v) for u, v in zip(s0, s1)) and torch.equal(e0, e1)
        assert torch.equal(x0, x1) and torch.equal(r0, r1) and torch.equal(a0, a1)
    assert e0.max().item() >= 1


def test_rollout_with_policy_and_one_hot():
    from rl_collision_avoidance_amd.ga3c.rollout import BatchedRollout
    W, N = 256, 4
    env = _make(W, N, 1)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0)

    def policy(x):                       # uniform policy, zero value
        B = x.shape[0]
        return torch.full((B, 11), 1.0 / 11, device=x.device), torch.zeros(B, device=x.device)
    roll = BatchedRollout(env, policy, generator=gen)
    roll.reset()
    total = 0
    for k in range(120):
        roll.step()
        if k % 10 == 9:
            total += len(roll.drain())
    b = roll.drain(flush_all=True)
    assert total > 0 and roll.lost_blocks == 0
    assert len(b) > 0 and b.x.shape == (len(b), 26) and b.a.shape == (len(b), 11) and b.a.dtype == torch.float32
    assert torch.all(b.a.sum(dim=1) == 1) and torch.isfinite(b.r).all() and torch.isfinite(b.x).all()
    eps = roll.drain_episodes()
    assert eps.shape[1] == 3 and len(eps) > 0
    assert len(roll.drain(flush_all=True)) == 0
    # a ring that is too short for the drain cadence loses blocks and says so
    small = BatchedRollout(env, policy, generator=gen, ring_len=26)
    small.reset()
    for _ in range(80):
        small.step()
    b = small.drain()
    assert small.lost_blocks > 0 and len(b) > 0
    assert torch.all(b.x[:, 3] > 0) and torch.all(b.a.sum(dim=1) == 1)       # pref_speed column: every kept row is a real one
    small.close()
    roll.close(); env.close()


def test_fused_inference_path_matches_generic_forward():
    from rl_collision_avoidance_amd.config import EnvConfig
    from rl_collision_avoidance_amd.ga3c.network import NetworkVP_rnn

    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = 10
            EnvConfig.__init__(self)
    net = NetworkVP_rnn(Cfg(), seed=2).cuda()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4096, 68, generator=g)
    x[:, 0] = torch.randint(0, 10, (4096,), generator=g).float()
    x = x.cuda()
    p_fast, v_fast = net.predict_p_and_v(x)
    with torch.no_grad():
        _, p_ref, v_ref = net(x)                 # plain fp32 reference of the same graph
    assert torch.allclose(p_fast, p_ref, atol=2e-5) and torch.allclose(v_fast, v_ref, atol=2e-4)
    assert torch.allclose(p_fast.sum(1), torch.ones(4096, device="cuda"), atol=1e-5)
