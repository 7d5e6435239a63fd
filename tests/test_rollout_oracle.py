"""The rollout oracle (oracle/rollout_oracle.py) is PINNED here: bit-for-bit against golden vectors
produced by the reference's own ProcessAgent (tests/golden/rollout_golden.npz), and -- when
/root/reference is present (this container, not the GPU box) -- against the reference run live."""
import os
import sys

import numpy as np
import pytest

from oracle import rollout_oracle as ro

GOLD = os.path.join(os.path.dirname(__file__), "golden", "rollout_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_survey_probe_returns(gold):
    # SURVEY.md section 8c probe: 5 exps, r = 0.1 t, gamma 0.97, bootstrap 0.5
    buf = [ro.Entry(np.zeros(26), 0, 0.1 * t, False) for t in range(5)]
    rows, left = ro.n_step_returns(buf, 0.97, 0.5, False, 20)
    got = np.array([e.reward for e in rows])
    assert left is None and np.array_equal(got, gold["probe_returns"])
    np.testing.assert_allclose(got, [1.001628, 1.032606, 0.96145, 0.785], atol=5e-7)


def test_accumulate_rewards_golden(gold):
    gamma, t_max = float(gold["discount"]), int(gold["time_max"])
    assert (gamma, t_max) == (0.97, 20)
    for k in range(int(gold["acc_n"])):
        g = lambda name: gold["acc_%d_%s" % (k, name)]
        rew, done, terminal = g("rew"), bool(g("done")), float(g("terminal"))
        buf = [ro.Entry(np.full(26, float(j)), j % 11, float(rew[j]), done and j == len(rew) - 1) for j in range(len(rew))]
        rows, left = ro.n_step_returns(buf, gamma, terminal, done, t_max)
        x, r, a = ro.to_arrays(rows, 11)
        assert np.array_equal(r, g("r_out")), k
        assert np.array_equal(x[:, 0].astype(np.int64), g("idx_out")), k
        assert a.dtype == np.float32 and np.array_equal(a, g("a_out")), k
        assert (-1 if left is None else int(left[0].state[0])) == int(g("leftover")), k
        assert np.array_equal(np.array([e.reward for e in buf]), g("rew_after")), k


def _episode(gold, e):
    g = lambda name: gold["ep_%d_%s" % (e, name)]
    return dict(obs=g("obs"), rewards=g("rewards"), done=g("done"), learning=g("learning"), n_present=int(g("n")),
                actions=g("actions"), values=g("values"))


def test_run_episode_golden(gold):
    saw_leftover = saw_reflush = saw_tmax = False
    for e in range(int(gold["num_episodes"])):
        g = lambda name: gold["ep_%d_%s" % (e, name)]
        chunks = ro.run_episode(gamma=0.97, t_max=20, **_episode(gold, e))
        assert len(chunks) == int(g("num_yields")), e
        assert [len(c.r) for c in chunks] == list(g("rows")), e
        D = g("obs").shape[-1] - 1
        x = np.concatenate([c.x.reshape(-1, D) for c in chunks])
        assert np.array_equal(x, g("x")), e
        assert np.array_equal(np.concatenate([c.r for c in chunks]), g("r")), e
        assert np.array_equal(np.concatenate([c.a for c in chunks]), g("a")), e
        assert np.array_equal(np.array([c.score for c in chunks]), g("reward_sum")), e
        saw_leftover |= any(c.leftover for c in chunks)
        saw_tmax |= any(len(c.r) == 20 and not c.leftover for c in chunks)
        per_agent = {}
        for c in chunks:
            per_agent.setdefault(c.agent, []).append(c)
        saw_reflush |= any(sum(1 for c in cs if len(c.r) == 2) >= 3 for cs in per_agent.values())
    assert saw_leftover and saw_reflush and saw_tmax      # the golden set exercises every quirk


@pytest.mark.skipif(not os.path.isdir("/root/reference/ga3c/GA3C"), reason="reference tree not present")
def test_against_live_reference():
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_rollout_golden as mk
    rng = np.random.default_rng(7)
    Config, PA, EX = mk.load_reference("TrainPhase1")
    for _ in range(10):
        n_present = int(rng.integers(1, 5))
        script = mk.make_script(rng, 4, Config.NN_INPUT_SIZE, n_present, 3 * Config.TIME_MAX, Config.TIME_MAX)
        ref = mk.run_reference_episode(Config, PA, script)
        mine = ro.run_episode(script["obs"], script["rewards"], script["done"], script["learning"], script["n"],
                              script["actions"], script["values"], Config.DISCOUNT, Config.TIME_MAX)
        assert len(ref) == len(mine)
        for (x, r, a, s), c in zip(ref, mine):
            assert np.array_equal(x, c.x) and np.array_equal(r, c.r) and np.array_equal(a, c.a) and s == c.score
