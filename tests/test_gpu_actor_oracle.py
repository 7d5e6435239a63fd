"""The fused actor kernel (`cavoid_actor_run`: policy forward -> action draw -> env.step -> Experience bookkeeping, K steps in ONE
launch; the loop body of /root/reference/ga3c/GA3C/ProcessAgent.py:116-211) held DIRECTLY against the two CPU oracles -- not
against the step-by-step HIP path (tests/test_gpu_actor.py does that, bitwise):

  * env half: the actions the kernel drew are read back from its experience ring and replayed through the float64 env oracle
    (`oracle/cavoid_oracle.c`, step_autoreset): the observation the policy acted on at EVERY step (the ring's state rows), the
    reward of every step, and at every launch boundary done / game_over, every flag bit, the world state and the episode counters;
  * rollout half: the same recording replayed, world by world and episode by episode, through `oracle/rollout_oracle.run_episode`
    (bit-pinned to the reference's own ProcessAgent): exactly the oracle's training rows must have been emitted -- states and
    actions bit-exact, n-step returns to 1e-6 -- and the episode log must agree;
  * policy half: the values V(s_t) the flushes bootstrapped from and (greedy case) the actions are the PyTorch float32 graph's.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

import replay as rp
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu

OBS_TOL, STATE_TOL = 1e-5, 1e-9
T_MAX, GAMMA = 5, 0.97


def _make(W, N, seed, reflush, greedy, ring_len, **over):
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    from rl_collision_avoidance_amd.config import EnvConfig
    from rl_collision_avoidance_amd.ga3c.network import NetworkVP_rnn
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
    from rl_collision_avoidance_amd.ga3c.rollout import BatchedRollout

    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
            EnvConfig.__init__(self)
    cfg = Cfg()
    env = BatchedCollisionAvoidanceEnv(W, cfg, device="cuda:0", seed=seed, **over)
    torch.manual_seed(99)
    net = NetworkVP_rnn(cfg).to("cuda:0")
    pol = FusedPolicy(net, seed=5)
    frozen = None
    if over.get("gen_frozen_fraction", 0.0) > 0.0:          # the network behind the frozen-network agents (cavoid_actor_run_mix)
        torch.manual_seed(7)
        frozen = FusedPolicy(NetworkVP_rnn(cfg).to("cuda:0"), seed=0)
    roll = BatchedRollout(env, pol, reflush_done=reflush, greedy=greedy, time_max=T_MAX, discount=GAMMA, ring_len=ring_len,
                          dup_capacity=2000000 if reflush else None, episode_capacity=100 * W, frozen_policy=frozen)
    roll.reset()
    return env, net, pol, roll


@pytest.mark.parametrize("N,W,reflush,greedy,launches,over", [
    # BASELINE configs[4] shape, the reference's re-flush quirk on
    (4, 512, True, False, (1, 2, 7) + (16,) * 5 + (6,), dict()),
    # configs[3] shape: 2..10 agents per world, scripted (static / non-cooperative) agents, 6 worlds per tile
    (10, 300, False, False, (3, 16, 16, 16, 16, 16, 13), dict(gen_min_agents=2, gen_nonlearning_fraction=0.2)),
    # ORCA agents, box scenarios generated inside the step (actor_kernel<N, true>)
    (4, 300, False, False, (2, 16, 16, 16, 16, 16, 14),
     dict(rvo_enabled=1, gen_rvo_fraction=0.5, gen_nonlearning_fraction=0.5, gen_min_agents=2, gen_mode=1, gen_pool_size=0)),
    # a finite sensing horizon and a time-step reward inside the loop (run-ws/config.yaml:249-251,326-328)
    (4, 256, False, False, (4, 16, 16, 16, 16, 16, 12), dict(sensing_horizon=3.0, reward_time_step=-0.01, gen_min_agents=2)),
    # the training mix with frozen-network agents (scripted policy 4; their actions come from a second network inside the launch)
    (4, 300, False, False, (3, 16, 16, 16, 16, 16, 13),
     dict(gen_min_agents=2, gen_nonlearning_fraction=0.6, gen_static_fraction=0.2, gen_frozen_fraction=0.6)),
    # PLAY_MODE (argmax), in-kernel ring generator, ragged last tile
    (3, 130, False, True, (5, 16, 16, 16, 16, 16, 11), dict(gen_pool_size=0, gen_min_agents=2)),
])
def test_fused_actor_against_the_oracles(N, W, reflush, greedy, launches, over):
    _check_fused_actor(N, W, reflush, greedy, launches, over)


# Round 6: the same checks where the hardware hazard of DESIGN.md 3.7 (d) lived -- 512 tiles, i.e. TWO workgroups per CU, one tile's env
# step running beside the partner tile's matrix phase on the same SIMDs.  The nondeterministic v_par rows of round 4 appeared ONLY in this
# regime and were found by a soak, not by `pytest -m gpu`; here the float64 env oracle checks every observation row the policy acted on at
# every step of BASELINE configs[4]'s own shape (4 x 8192) and of 10 x 4096 (6 worlds per tile -> 683 tiles).  The env and policy halves run
# over all worlds; the rollout half (a pure-Python replay, world by world) over every 16th world.
@pytest.mark.parametrize("N,W,launches,over", [
    (4, 8192, (1, 16, 16, 16, 16, 16, 7), dict()),
    (10, 4096, (2, 16, 16, 16, 16, 16, 6), dict(gen_min_agents=2, gen_nonlearning_fraction=0.2)),
])
def test_fused_actor_against_the_oracles_two_workgroups_per_cu(N, W, launches, over):
    _check_fused_actor(N, W, False, False, launches, over, rollout_worlds=np.arange(0, W, 16))


def test_a_slice_of_the_actor_soak():
    """20 seconds of tools/actor_soak.py (random shapes incl. 512 tiles, scenario sources, ORCA agents, the re-flush quirk; the fused actor
    kernel against step-by-step stepping, bitwise) inside the suite: the run that found round 4's wrong rows was this soak"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "actor_soak.py"), "20"], cwd=root, timeout=600,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert out.returncode == 0 and "bitwise" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])


def _check_fused_actor(N, W, reflush, greedy, launches, over, rollout_worlds=None):
    seed = 33
    T = sum(launches)
    env, net, pol, roll = _make(W, N, seed, reflush, greedy, T + 8, **over)
    assert roll.fused_available
    ocfg, ogen = rp.oracle_for(N, **over)
    st = co.State.empty(W, N)
    ep = np.zeros(W, np.uint32)
    co.generate(ocfg, ogen, seed, st, ep)
    obs0 = roll.obs.cpu().numpy().copy()
    assert rp.obs_diff(obs0, co.observe(ocfg, st)).max() <= OBS_TOL
    is_learning = [obs0[..., :1].astype(np.float32)]             # column 0 of the observation acted on at step t (from the ORACLE)
    st0 = st.copy()
    frozen_at = []                                               # [t] -> bool [W, N]: a running frozen-network agent acts at step t
    ora = []
    t0 = 0
    for k in launches:
        roll.run_fused(k)
        acts = roll.act_ring[t0:t0 + k].cpu().numpy().astype(np.int32).reshape(k, W, N)
        assert acts.min() >= 0 and acts.max() < env.num_actions
        for j in range(k):
            frozen_at.append((((st.flags >> 8) & 7 == 4) & (st.flags & 0x20 != 0) & (st.flags & 7 == 0)).reshape(W, N))
            ora.append(co.step_autoreset(ocfg, ogen, seed, st, ep, acts[j]))
            is_learning.append(ora[-1][0][..., :1].astype(np.float32))
        t0 += k
        # ---- launch boundary: what the env half of the kernel left behind, against the oracle's last step -------------------
        oobs, orew, odone, ogo = ora[-1]
        tag = ("boundary", t0)
        assert np.array_equal(env.done.cpu().numpy(), odone), tag
        assert np.array_equal(env.game_over.cpu().numpy(), ogo), tag
        assert np.abs(env.rewards.cpu().numpy() - orew).max() <= OBS_TOL, tag
        assert rp.obs_diff(roll.obs.cpu().numpy(), oobs).max() <= OBS_TOL, tag
        f64, f32, fl = [v.cpu().numpy() for v in env.get_state()]
        assert np.array_equal(fl.view(np.uint32), st.flags), tag                      # every flag bit
        assert np.array_equal(f32, st.f32), tag
        np.testing.assert_allclose(f64, st.f64, rtol=0, atol=STATE_TOL, err_msg=str(tag))
        assert np.array_equal(env.episode.cpu().numpy().view(np.uint32), ep), tag
    assert roll.step_index == T and ep.max() >= 1

    # ---- every step: the state rows and rewards the kernel recorded, against the oracle's step --------------------------------
    D = env.obs_width - 1
    x_ring = roll.x[:T].cpu().numpy().reshape(T, W, N, D)
    rew_ring = roll.val[:T].cpu().numpy().reshape(T, W, N)
    assert np.array_equal(x_ring[0], obs0[..., 1:])
    for t in range(T):
        if t >= 1:          # the observation acted on at step t = what the oracle's step t-1 returned (restarted worlds: the new episode's)
            d = rp.obs_diff(x_ring[t], ora[t - 1][0][..., 1:], heading_col=2)
            assert d.max() <= OBS_TOL, (t, d.max())
            assert np.array_equal(x_ring[t][..., 0], ora[t - 1][0][..., 1].astype(np.float32)), t      # num_other_agents exact
        assert np.abs(rew_ring[t] - ora[t][1]).max() <= OBS_TOL, t

    # ---- the policy half: V(s_t) and (greedy) the actions are the PyTorch float32 graph's --------------------------------------
    acts_all = roll.act_ring[:T].cpu().numpy().astype(np.int32).reshape(T, W, N)
    vals_all = np.zeros((T, W, N), np.float32)
    worst_v, mism = 0.0, 0
    for t in range(T):
        xt = roll.x[t]
        p_k, v_k = pol(xt)                                  # the kernel stand-alone: bit-identical to the pass inside the loop
        vals_all[t] = v_k.cpu().numpy().reshape(W, N)
        with torch.no_grad():
            _, p_ref, v_ref = net.forward(xt)
        # the rows the loop runs the network for: a learning agent that has not finished (done in the step that produced this observation,
        # unless its world has just restarted) -- a finished agent is handed action 0 / value 0, which nothing reads.  With the re-flush
        # quirk or frozen-network agents in the batch every row runs.
        need = is_learning[t][..., 0] > 0.5
        if t >= 1 and not reflush and roll.frozen_policy is None:
            need = need & ((ora[t - 1][3].reshape(W, 1) != 0) | (ora[t - 1][2].reshape(W, N) == 0))
        running = torch.from_numpy(need.reshape(-1)).cuda()
        worst_v = max(worst_v, float((v_k - v_ref).abs()[running].max()))
        if greedy:
            top2 = p_ref.topk(2, dim=1).values
            clear = running & ((top2[:, 0] - top2[:, 1]) > 1e-4)
            want = p_ref.argmax(dim=1).to(torch.int32)
            mism += int((torch.from_numpy(acts_all[t].reshape(-1)).cuda()[clear] != want[clear]).sum())
        else:
            assert float((p_k - p_ref).abs().max()) <= 2e-5
    assert worst_v <= 2e-4 and mism == 0
    if roll.frozen_policy is not None:
        # the frozen-network agents took THEIR network's argmax (checked where the float32 graph's top two are clearly apart)
        fnet, bad, seen = roll.frozen_policy.net, 0, 0
        for t in range(T):
            with torch.no_grad():
                _, p_fz, _ = fnet.forward(roll.x[t])
            top2 = p_fz.topk(2, dim=1).values
            clear = ((top2[:, 0] - top2[:, 1]) > 1e-4).cpu().numpy().reshape(W, N)
            fz_rows = frozen_at[t] & clear
            want = p_fz.argmax(dim=1).cpu().numpy().reshape(W, N)
            bad += int((acts_all[t][fz_rows] != want[fz_rows]).sum())
            seen += int(fz_rows.sum())
        assert seen > 100 and bad == 0, (seen, bad)

    # ---- the rollout half: the training rows and the episode log, against the reference-pinned rollout oracle ------------------
    rec = []
    for t in range(T):
        obs_t = np.concatenate([is_learning[t], x_ring[t]], axis=-1)
        rec.append((obs_t, acts_all[t], vals_all[t], rew_ring[t], ora[t][2].astype(bool), ora[t][3].astype(bool)))
    batch = roll.drain(flush_all=True)
    episodes = roll.drain_episodes().cpu().numpy()
    assert batch.dropped == 0 and roll.lost_blocks == 0 and len(batch) > W * 5
    rows = [v.cpu().numpy() for v in (batch.x, batch.r, batch.a_index, batch.src)]
    n_eps = len(episodes)
    if rollout_worlds is not None:
        rec, rows, episodes = rp.subset_worlds(rec, rows, episodes, rollout_worlds)
    matched = rp.replay_rollout(rec, rows, episodes, reflush, GAMMA, T_MAX)
    assert matched > (W * 5 if rollout_worlds is None else len(rollout_worlds) // 2)   # (only FINISHED episodes' rows are replayed)
    assert n_eps == int(sum(o[3].sum() for o in ora))           # one log record per finished episode
    roll.close(); env.close()
