#!/usr/bin/env python
"""Kernel micro-benchmark for optimisation work: mean HIP-event kernel time of the autoreset step
at several world counts (plus, under rocprofv3 --kernel-trace, the plain-step kernel)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", type=int, nargs="+", default=[1024, 8192, 65536, 1048576])
    ap.add_argument("--agents", type=int, default=4)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--plain", action="store_true", help="also launch 60 plain (no auto-reset) steps per size")
    ap.add_argument("--gen-min", type=int, default=None)
    ap.add_argument("--spl", type=int, nargs="+", default=[1, 32], help="steps per launch")
    args = ap.parse_args()
    import torch
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    from rl_collision_avoidance_amd.config import EnvConfig
    N = args.agents

    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
            EnvConfig.__init__(self)
    bytes_as = 4 * (11 + 1 + 7 + (6 + 7 * (N - 1)) + 1 + 1)
    out = []
    for W in args.worlds:
        over = {} if args.gen_min is None else {"gen_min_agents": args.gen_min}
        env = BatchedCollisionAvoidanceEnv(W, Cfg(), seed=7, **over)
        acts = torch.randint(0, 11, (32, W, N), device="cuda", dtype=torch.int32)
        env.reset()
        env.step_autoreset_n(acts)
        env.step_autoreset_n(acts)
        for spl in args.spl:
            ms = env.kernel_time_ms(acts, args.steps, spl) / min(spl, 32)
            rec = {"W": W, "N": N, "spl": spl, "us_per_step": round(ms * 1e3, 3), "Gagent_steps_s": round(W * N / ms / 1e6, 3),
                   "GBps": round(bytes_as * W * N / ms / 1e6, 1), "wpw": os.environ.get("CAVOID_WPW", "auto")}
            out.append(rec)
            print(json.dumps(rec), flush=True)
        if args.plain:
            env.reset()
            for t in range(60):
                env.step(acts[t % 32])
            torch.cuda.synchronize()
        env.close()


if __name__ == "__main__":
    main()
