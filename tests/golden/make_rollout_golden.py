#!/usr/bin/env python
"""Generates tests/golden/rollout_golden.npz by running the REFERENCE's own GA3C rollout code
(/root/reference/ga3c/GA3C/ProcessAgent.py: _accumulate_rewards, convert_to_nparray, run_episode)
in this container.  The reference is imported, never copied: only the input scripts and the
values it returned are stored.  Needs /root/reference; not runnable on the GPU box.

    python tests/golden/make_rollout_golden.py

Stubs: the env package the reference imports (absent submodule) resolves to this repo's compat
shim; NumPy 2 dropped ``np.product`` (used at Config.py:69), aliased here; ``Environment`` is
replaced by a scripted fake env (the real one only forwards to ``game.step``) and ``predict`` by
a scripted one-hot policy so that ``np.random.choice`` is deterministic."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/ga3c"
sys.path[:0] = [os.path.join(ROOT, "rl_collision_avoidance_amd", "compat"), ROOT, REF, os.path.join(REF, "GA3C")]
np.product = np.prod


def load_reference(config_class):
    os.environ["GYM_CONFIG_CLASS"] = config_class
    os.environ["GYM_CONFIG_PATH"] = os.path.join(REF, "GA3C", "Config.py")
    for m in [m for m in sys.modules if m == "GA3C" or m.startswith("GA3C.") or m in ("ProcessAgent", "Environment", "Experience")]:
        del sys.modules[m]
    from GA3C import Config
    import ProcessAgent as PA
    import Experience as EX
    return Config, PA, EX


class ScriptedEnv(object):
    """What ProcessAgent needs from Environment: latest_observations, previous_state, reset, step."""

    def __init__(self, obs, rewards, done, learning, n_present):
        self.obs, self.rewards, self.done, self.learning, self.n = obs, rewards, done, learning, n_present
        self.t = 0
        self.latest_observations = None
        self.previous_state = self.current_state = None

    def _push(self, o):
        self.latest_observations = o
        self.previous_state = self.current_state
        self.current_state = np.array([o[:, 1:]])

    def reset(self):
        self.t = 0
        self.previous_state = self.current_state = None
        self._push(self.obs[0])

    def step(self, actions, pid, count):
        t = self.t
        self.taken = actions[0]
        rew = [self.rewards[t, :self.n]]
        info = [{"which_agents_done": {i: bool(self.done[t, i]) for i in range(self.n)},
                 "which_agents_learning": {i: bool(self.learning[i]) for i in range(self.n)}}]
        over = bool(np.all(self.done[t, :self.n][self.learning[:self.n]]))
        self.t += 1
        self._push(self.obs[self.t])
        return rew, over, info


def make_script(rng, N, D, n_present, max_len, time_max):
    """A random episode: who learns, per-step rewards, monotone done flags, scripted actions/values."""
    learning = np.zeros(N, bool)
    learning[:n_present] = rng.random(n_present) < 0.75
    learning[0] = True
    # episode lengths: some agents finish early (post-done re-flush quirk), some exactly at T_max
    finish = rng.integers(1, max_len + 1, size=N)
    if rng.random() < 0.5:
        finish[rng.integers(0, n_present)] = time_max + 1      # done with a full (T_max+1) buffer
    if rng.random() < 0.3:
        finish[rng.integers(0, n_present)] = 2 * time_max + 1
    T = int(finish[:n_present][learning[:n_present]].max())
    done = np.zeros((T, N), bool)
    for i in range(N):
        done[min(finish[i], T + 1) - 1:, i] = finish[i] <= T
    done[:, n_present:] = False
    rewards = np.round(rng.normal(0, 0.2, size=(T, N)), 3)
    obs = np.round(rng.normal(0, 1, size=(T + 1, N, 1 + D)), 1)   # coarse values: the fixture compresses well
    obs[:, :, 0] = 0.0
    obs[:, :n_present, 0] = learning[:n_present]
    obs[:, n_present:, :] = 0.0
    actions = rng.integers(0, 11, size=(T, N))
    values = np.round(rng.normal(0, 0.5, size=(T, N)), 3)
    return dict(obs=obs, rewards=rewards, done=done, learning=learning, n=n_present, actions=actions, values=values, T=T)


def run_reference_episode(Config, PA, script):
    agent = PA.ProcessAgent(0, None, None, None, Config.NUM_ACTIONS)
    env = ScriptedEnv(script["obs"], script["rewards"], script["done"], script["learning"], script["n"])
    agent.env = env
    state = {"calls": 0}
    order = [(t, i) for t in range(script["T"]) for i in range(Config.MAX_NUM_AGENTS_IN_ENVIRONMENT)
             if script["obs"][t, i, 0]]

    def predict(obs_row):
        t, i = order[state["calls"]]
        state["calls"] += 1
        assert np.array_equal(obs_row, script["obs"][t, i])
        p = np.zeros(Config.NUM_ACTIONS)
        p[script["actions"][t, i]] = 1.0
        return p, script["values"][t, i]
    agent.predict = predict
    out = []
    for x_, r_, a_, rs in agent.run_episode():
        out.append((np.array(x_, dtype=np.float64), np.array(r_, dtype=np.float64), np.array(a_, dtype=np.float32), float(rs)))
    assert env.t == script["T"]
    return out


def main():
    rng = np.random.default_rng(20240928)
    store = {}
    # ---- (1) _accumulate_rewards / convert_to_nparray known answers ---------------------------------
    Config, PA, EX = load_reference("TrainPhase1")
    agent = PA.ProcessAgent(0, None, None, None, Config.NUM_ACTIONS)
    acc_cases = []
    for length in (1, 2, 3, 5, Config.TIME_MAX - 1, Config.TIME_MAX, Config.TIME_MAX + 1):
        for done in (False, True):
            for terminal in (0.0, 0.5, -0.37):
                rew = np.round(rng.normal(0, 0.3, size=length), 3)
                exps = [EX.Experience(np.full(Config.NN_INPUT_SIZE, float(k)), int(k % 11), None, float(rew[k]), done and k == length - 1)
                        for k in range(length)]
                ret, left = agent._accumulate_rewards(exps, Config.DISCOUNT, terminal, done)
                x_, r_, a_ = agent.convert_to_nparray(ret)
                acc_cases.append(dict(rew=rew, done=done, terminal=terminal, r_out=r_, idx_out=x_[:, 0].astype(np.int64),
                                      a_out=a_, leftover=-1 if left is None else int(left[0].state_image[0]),
                                      rew_after=np.array([e.reward for e in exps])))
    store["acc_n"] = len(acc_cases)
    for k, c in enumerate(acc_cases):
        for key, val in c.items():
            store["acc_%d_%s" % (k, key)] = np.asarray(val)
    store["discount"] = Config.DISCOUNT
    store["time_max"] = Config.TIME_MAX
    # the survey's probe: 5 exps, r = 0.1 t, gamma 0.97, R = 0.5
    exps = [EX.Experience(np.zeros(Config.NN_INPUT_SIZE), 0, None, 0.1 * t, False) for t in range(5)]
    ret, _ = agent._accumulate_rewards(exps, 0.97, 0.5, False)
    store["probe_returns"] = np.array([e.reward for e in ret])

    # ---- (2) whole-episode control flow of run_episode ----------------------------------------------------
    ep = 0
    for config_class, n_eps in (("TrainPhase1", 20), ("TrainPhase2", 5)):
        Config, PA, EX = load_reference(config_class)
        N, D = Config.MAX_NUM_AGENTS_IN_ENVIRONMENT, Config.NN_INPUT_SIZE
        for _ in range(n_eps):
            n_present = int(rng.integers(1, N + 1))
            script = make_script(rng, N, D, n_present, max_len=3 * Config.TIME_MAX, time_max=Config.TIME_MAX)
            yields = run_reference_episode(Config, PA, script)
            pre = "ep_%d_" % ep
            store[pre + "N"] = N
            for key in ("obs", "rewards", "done", "learning", "n", "actions", "values", "T"):
                store[pre + key] = np.asarray(script[key])
            store[pre + "num_yields"] = len(yields)
            store[pre + "rows"] = np.array([len(y[1]) for y in yields], dtype=np.int64)
            store[pre + "x"] = np.concatenate([y[0].reshape(-1, D) for y in yields]) if yields else np.zeros((0, D))
            store[pre + "r"] = np.concatenate([y[1] for y in yields]) if yields else np.zeros(0)
            store[pre + "a"] = np.concatenate([y[2] for y in yields]) if yields else np.zeros((0, 11), np.float32)
            store[pre + "reward_sum"] = np.array([y[3] for y in yields])
            ep += 1
    store["num_episodes"] = ep
    out = os.path.join(ROOT, "tests", "golden", "rollout_golden.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, os.path.getsize(out), "bytes;", ep, "episodes,", len(acc_cases), "accumulate cases; probe", store["probe_returns"])


if __name__ == "__main__":
    main()
