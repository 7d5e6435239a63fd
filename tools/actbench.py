#!/usr/bin/env python
"""Actor-kernel micro-benchmark (development aid): K closed-loop GA3C actor steps per launch (`cavoid_actor_run`), HIP-event time
per env step, no hand-over.  usage: python tools/actbench.py [worlds] [agents] [steps per launch] [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
from rl_collision_avoidance_amd.config import EnvConfig
from rl_collision_avoidance_amd.ga3c.network import NetworkVP_rnn
from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
from rl_collision_avoidance_amd.ga3c.rollout import BatchedRollout


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    L = int(sys.argv[4]) if len(sys.argv) > 4 else 20

    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
            EnvConfig.__init__(self)
    cfg = Cfg()
    env = BatchedCollisionAvoidanceEnv(W, cfg, seed=3)
    torch.manual_seed(0)
    pol = FusedPolicy(NetworkVP_rnn(cfg).cuda(), seed=5)
    roll = BatchedRollout(env, pol, reflush_done=False, ring_len=4 * K + 64)
    roll.reset()
    for _ in range(3):
        roll.run_fused(K)
        roll.drain(provenance=False)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0.0
        for _ in range(L):
            e0.record()
            roll.run_fused(K)
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
            roll.drain(provenance=False)
        best = min(best, tot * 1e3 / (L * K))
    print({"W": W, "N": N, "steps_per_launch": K, "us_per_env_step": round(best, 2)})


if __name__ == "__main__":
    main()
