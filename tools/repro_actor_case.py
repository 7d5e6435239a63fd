#!/usr/bin/env python
"""Replay ONE configuration of tools/actor_soak.py step group by step group and say which tensors differ between the fused actor
kernel and step-by-step stepping (development aid).  usage: python tools/repro_actor_case.py  (edit CASE below)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from actor_soak import make

CASE = {'N': 4, 'W': 8192, 'seed': 1005221373, 'net_seed': 557405, 'reflush': False, 'greedy': False, 'time_max': 5,
        'gen_min_agents': 4, 'gen_nonlearning_fraction': 0.7, 'gen_pool_size': 20000, 'rvo_enabled': 1, 'gen_rvo_fraction': 1.0}


def main():
    c = dict(CASE)
    N, W, seed, net_seed, reflush, greedy, time_max = (c.pop(k) for k in ("N", "W", "seed", "net_seed", "reflush", "greedy", "time_max"))
    ks = [int(x) for x in sys.argv[1:]] or [2, 1, 1, 3, 5, 8]
    ea, a = make(W, N, seed, reflush, greedy, time_max, net_seed, **c)
    eb, b = make(W, N, seed, reflush, greedy, time_max, net_seed, **c)
    done = 0
    for k in ks:
        a.run_fused(k)
        for _ in range(k):
            b.step()
        done += k
        bad = []
        for name, x, y in [("obs", a.obs, b.obs), ("episode", ea.episode, eb.episode), ("rewards", ea.rewards, eb.rewards), ("done", ea.done, eb.done)] + \
                [("state%d" % i, x, y) for i, (x, y) in enumerate(zip(ea.get_state(), eb.get_state()))] + \
                [(n, getattr(a, n), getattr(b, n)) for n in ("x", "val", "ret", "act_ring", "emit_t")]:
            if not torch.equal(x, y):
                d = (x != y)
                if x.dtype.is_floating_point:
                    d = d & ~(torch.isnan(x) & torch.isnan(y))
                idx = d.nonzero()
                bad.append((name, int(d.sum()), idx[0].tolist() if len(idx) else None))
        print("after %d steps:" % done, bad if bad else "equal", flush=True)
        if bad:
            break


if __name__ == "__main__":
    main()
