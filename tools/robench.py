#!/usr/bin/env python
"""Rollout-kernel micro-benchmark: env step + bookkeeping with scripted actions/values (no policy).
Run under rocprofv3 --kernel-trace --stats to read the per-kernel averages."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
from rl_collision_avoidance_amd.config import EnvConfig
from rl_collision_avoidance_amd.ga3c.rollout import BatchedRollout


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400

    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
            EnvConfig.__init__(self)
    env = BatchedCollisionAvoidanceEnv(W, Cfg(), seed=3)
    roll = BatchedRollout(env, None, reflush_done=False)
    roll.reset()
    acts = torch.randint(0, 11, (16, W, N), device="cuda", dtype=torch.int32)
    vals = torch.randn(16, W, N, device="cuda")
    for t in range(100):
        roll.step(acts[t % 16], vals[t % 16])
        if t % 8 == 7:
            roll.drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows = 0
    for t in range(steps):
        roll.step(acts[t % 16], vals[t % 16])
        if t % 8 == 7:
            rows += len(roll.drain())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("W=%d N=%d: %.1f us per step (wall, incl. python), %d rows drained" % (W, N, dt * 1e6 / steps, rows))


if __name__ == "__main__":
    main()
