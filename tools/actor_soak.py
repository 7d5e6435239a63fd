#!/usr/bin/env python
"""Randomised soak of the fused actor kernel (cavoid_actor_run) against step-by-step stepping (BatchedRollout.step: one launch
per phase), bitwise: random agent counts, world counts, launch lengths, flush lengths, scripted-agent fractions, scenario
sources, the re-flush quirk, greedy / sampled actions.  usage: python tools/actor_soak.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
from rl_collision_avoidance_amd.config import EnvConfig
from rl_collision_avoidance_amd.ga3c.network import NetworkVP_rnn
from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
from rl_collision_avoidance_amd.ga3c.rollout import BatchedRollout


def make(W, N, seed, reflush, greedy, time_max, net_seed, **kw):
    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
            EnvConfig.__init__(self)
    cfg = Cfg()
    env = BatchedCollisionAvoidanceEnv(W, cfg, device="cuda:0", seed=seed, **kw)
    torch.manual_seed(net_seed)
    net = NetworkVP_rnn(cfg).to("cuda:0")
    with torch.no_grad():                                   # less uniform policies than the initialisation's: sharper heads
        net.p_kernel.mul_(6.0)
    # (the fused kernel runs the network for the rows that still need an action only -- every row with the re-flush quirk -- and so does
    #  the step-by-step path with skip_finished: like for like, down to what a finished agent's ring entry holds)
    roll = BatchedRollout(env, FusedPolicy(net, seed=net_seed + 1), reflush_done=reflush, greedy=greedy, time_max=time_max,
                          dup_capacity=4 * W * N * 64 if reflush else None, skip_finished=not reflush)
    roll.reset()
    return env, roll


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(31)
    t0, cases, steps_total, rows_total = time.time(), 0, 0, 0
    while time.time() - t0 < budget:
        N = int(rng.integers(2, 17))                       # (one agent: nothing to observe, cavoid_policy_create wants max_other >= 1)
        wpw = 64 // N
        W = int(rng.choice([1, int(rng.integers(2, 4 * wpw + 2)), int(rng.integers(50, 3000)), 512 * wpw]))
        kw = dict(gen_min_agents=int(rng.integers(1, N + 1)), gen_nonlearning_fraction=float(rng.choice([0.0, 0.3, 0.7])) if N > 1 else 0.0,
                  gen_pool_size=int(rng.choice([0, 5, 400, 20000])))
        if rng.random() < 0.3:
            kw["gen_mode"] = 1                                # box scenarios (from the pool, or generated inside the step)
        if rng.random() < 0.3 and 1 < N <= 12:
            kw.update(rvo_enabled=1, gen_rvo_fraction=float(rng.choice([0.3, 1.0])))    # ORCA agents among the scripted ones
            kw["gen_nonlearning_fraction"] = max(kw["gen_nonlearning_fraction"], 0.3)
        if rng.random() < 0.25:
            kw["sort_method"] = int(rng.integers(0, 3))
        if rng.random() < 0.2:
            kw.update(wrap_closed_end=1, sort_round_gap=0)
        reflush, greedy = bool(rng.random() < 0.4), bool(rng.random() < 0.25)
        time_max = int(rng.choice([2, 5, 8]))
        seed, net_seed = int(rng.integers(0, 1 << 30)), int(rng.integers(0, 1 << 20))
        ea, a = make(W, N, seed, reflush, greedy, time_max, net_seed, **kw)
        eb, b = make(W, N, seed, reflush, greedy, time_max, net_seed, **kw)
        if not a.fused_available:
            raise SystemExit("actor kernel unavailable for %r" % ((N, W, kw),))
        T, done = int(rng.integers(40, 200)), 0
        ok = True
        while done < T and ok:
            k = int(rng.choice([1, 2, 3, 5, 8, 16, 33]))
            a.run_fused(k)
            for _ in range(k):
                b.step()
            done += k
            ok = torch.equal(a.obs, b.obs) and torch.equal(ea.episode, eb.episode) and torch.equal(ea.rewards, eb.rewards) and \
                torch.equal(ea.done, eb.done) and all(torch.equal(x, y) for x, y in zip(ea.get_state(), eb.get_state())) and \
                all(torch.equal(getattr(a, n), getattr(b, n)) for n in ("x", "val", "ret", "act_ring", "emit_t"))
            if ok and rng.random() < 0.3:
                ba, bb = a.drain(), b.drain()
                ok = len(ba) == len(bb) and ba.dropped == bb.dropped == 0
                if ok and len(ba):
                    ka, kb = np.lexsort(ba.src.cpu().numpy().T[::-1]), np.lexsort(bb.src.cpu().numpy().T[::-1])
                    ok = all(np.array_equal(getattr(ba, n).cpu().numpy()[ka], getattr(bb, n).cpu().numpy()[kb]) for n in ("src", "x", "r", "a_index"))
                    rows_total += len(ba)
        if not ok:
            print("MISMATCH", dict(N=N, W=W, seed=seed, net_seed=net_seed, reflush=reflush, greedy=greedy, time_max=time_max, step=done, **kw))
            raise SystemExit(1)
        cases += 1
        steps_total += done * W * N
        for r in (a, b):
            r.close()
        ea.close(); eb.close()
    print({"cases": cases, "agent_steps_compared": steps_total, "training_rows_compared": rows_total, "seconds": round(time.time() - t0, 1),
           "result": "fused actor kernel == step-by-step path, bitwise (observations, world state, experience rings, training rows)"})


if __name__ == "__main__":
    main()
