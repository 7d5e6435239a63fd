#!/usr/bin/env python
"""Large seeded parity sweep: HIP path vs the float64 C oracle over many seeds / shapes, counting flag
mismatches (must be 0) and the worst observation / reward / state deviation.  Evidence for DESIGN.md."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import c_oracle as co
from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
from rl_collision_avoidance_amd.config import EnvConfig


def run(N, W, steps, seed, nonl, sort):
    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
            EnvConfig.__init__(self)
    pool = 4096
    env = BatchedCollisionAvoidanceEnv(W, Cfg(), seed=seed, gen_min_agents=2, gen_nonlearning_fraction=nonl,
                                       sort_method=sort, gen_pool_size=pool)
    ocfg = co.default_cfg(N, sort_method=sort)
    ogen = co.default_gen(2, N, nonl, pool_size=pool)
    env.reset()
    st = co.State.empty(W, N)
    ep = np.zeros(W, np.uint32)
    co.generate(ocfg, ogen, seed, st, ep)
    rng = np.random.default_rng(seed)
    worst = {"obs": 0.0, "rew": 0.0, "state": 0.0, "flag_mismatch": 0, "done_mismatch": 0, "episode_mismatch": 0}
    for t in range(steps):
        acts = rng.integers(0, 11, size=(W, N)).astype(np.int32)
        acts[rng.random((W, N)) < 0.75] = 2
        obs, rew, done, go = env.step_autoreset(torch.from_numpy(acts).cuda())
        oobs, orew, odone, ogo = co.step_autoreset(ocfg, ogen, seed, st, ep, acts)
        d = np.abs(obs.cpu().numpy().astype(np.float64) - oobs)
        d[..., 3] = np.minimum(d[..., 3], np.abs(d[..., 3] - 2 * np.pi))
        worst["obs"] = max(worst["obs"], float(d.max()))
        worst["rew"] = max(worst["rew"], float(np.abs(rew.cpu().numpy() - orew).max()))
        worst["done_mismatch"] += int((done.cpu().numpy() != odone).sum() + (go.cpu().numpy() != ogo).sum())
        if t % 25 == 24 or t == steps - 1:
            f64, f32, fl = env.get_state()
            worst["flag_mismatch"] += int((fl.cpu().numpy().view(np.uint32) != st.flags).sum())
            worst["state"] = max(worst["state"], float(np.abs(f64.cpu().numpy() - st.f64).max()))
            worst["episode_mismatch"] += int((env.episode.cpu().numpy().view(np.uint32) != ep).sum())
    env.close()
    worst["agent_steps"] = int(W * N * steps)
    return worst


def main():
    t0 = time.time()
    total = {"obs": 0.0, "rew": 0.0, "state": 0.0, "flag_mismatch": 0, "done_mismatch": 0, "episode_mismatch": 0, "agent_steps": 0}
    cases = [(4, 4096, 400, s, 0.0, 0) for s in range(6)] + [(4, 2048, 300, 100 + s, 0.4, 1) for s in range(3)] + \
            [(10, 1024, 300, 200 + s, 0.3, 0) for s in range(3)] + [(3, 2048, 300, 300, 0.3, 2), (16, 256, 200, 400, 0.1, 0)]
    for c in cases:
        r = run(*c)
        for k in ("obs", "rew", "state"):
            total[k] = max(total[k], r[k])
        for k in ("flag_mismatch", "done_mismatch", "episode_mismatch", "agent_steps"):
            total[k] += r[k]
        print(c, r, flush=True)
    total["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(total))


if __name__ == "__main__":
    main()
