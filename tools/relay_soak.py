#!/usr/bin/env python
"""Randomised soak of env_relay_kernel against one-step-per-launch stepping (bitwise): random agent counts, world counts, pool
sizes, launch lengths, consumer counts, scripted-agent fractions, packed / plain outputs.  usage: python tools/relay_soak.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
from rl_collision_avoidance_amd.config import EnvConfig


def make(W, N, seed, **kw):
    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
            EnvConfig.__init__(self)
    return BatchedCollisionAvoidanceEnv(W, Cfg(), device="cuda:0", seed=seed, **kw)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(2024)
    t0, cases, steps_total = time.time(), 0, 0
    while time.time() - t0 < budget:
        N = int(rng.integers(1, 7))
        wmax = 512 * (64 // N)
        W = int(rng.choice([1, 3, int(rng.integers(4, 200)), int(rng.integers(200, wmax + 1)), wmax]))
        nc = int(rng.integers(1, 5))
        kw = dict(gen_pool_size=int(rng.choice([1, 7, 300, 65536])), gen_min_agents=int(rng.integers(1, N + 1)),
                  gen_nonlearning_fraction=float(rng.choice([0.0, 0.3, 0.8])) if N > 1 else 0.0)
        if rng.random() < 0.3:
            kw["time_budget_from_goal_edge"] = 1
        if N > 1 and os.environ.get("RELAY_SOAK_RVO", "1") != "0" and rng.random() < float(os.environ.get("RELAY_SOAK_RVO_P", "0.3")):
            # ORCA agents: env_relay_kernel<N, true> (the state owner does not speculate across a step with a running ORCA agent in its tile)
            kw.update(rvo_enabled=1, gen_rvo_fraction=float(rng.choice([0.3, 0.6, 1.0])), gen_nonlearning_fraction=float(rng.choice([0.3, 0.6, 0.9])))
        seed = int(rng.integers(0, 1 << 30))
        os.environ["CAVOID_PIPELINE"] = "2"
        os.environ["CAVOID_RELAY_CONSUMERS"] = str(nc)
        a = make(W, N, seed, **kw)
        os.environ["CAVOID_PIPELINE"] = "0"
        b = make(W, N, seed, **kw)
        packed = rng.random() < 0.5
        a.reset(); b.reset()
        pk_a, pk_b = a.new_packed(), b.new_packed()
        T = int(rng.integers(40, 160))
        acts = rng.integers(0, 11, size=(T, W, N))
        acts[rng.random((T, W, N)) < 0.55] = 2
        acts = torch.from_numpy(acts.astype(np.int32)).cuda()
        lo = 0
        while lo < T:
            n = int(min(T - lo, rng.choice([2, 3, 5, 8, 13, 21, 34, 64])))
            if n < 2:
                break
            if packed:
                a.step_autoreset_packed(acts[lo:lo + n], pk_a)
                for t in range(lo, lo + n):
                    b.step_autoreset_packed(acts[t], pk_b)
                ok = torch.equal(pk_a, pk_b) and torch.equal(a.game_over, b.game_over)
            else:
                a.step_autoreset_n(acts[lo:lo + n])
                for t in range(lo, lo + n):
                    b.step_autoreset(acts[t])
                ok = torch.equal(a.obs, b.obs) and torch.equal(a.rewards, b.rewards) and torch.equal(a.done, b.done) and \
                    torch.equal(a.game_over, b.game_over)
            ok = ok and all(torch.equal(x, y) for x, y in zip(a.get_state(), b.get_state())) and torch.equal(a.episode, b.episode)
            if not ok:
                print("MISMATCH", dict(N=N, W=W, nc=nc, seed=seed, lo=lo, n=n, packed=packed, **kw), flush=True)
                sys.exit(1)
            lo += n
            steps_total += n * W * N
        a.close(); b.close()
        cases += 1
    print("relay soak: %d random cases, %.1f M agent-steps, 0 mismatches, %.0f s" % (cases, steps_total / 1e6, time.time() - t0))


if __name__ == "__main__":
    main()
