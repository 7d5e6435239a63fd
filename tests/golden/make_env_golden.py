#!/usr/bin/env python
"""Generates tests/golden/env_golden.npz from the reference-STYLE Python oracle (oracle/cavoid_oracle.py).

NOT a reference output: the env half of the reference is absent (empty submodule), so this fixture is
oracle-generated and "parity unpinned".  Its job is to be a committed regression anchor -- the C oracle
(CPU test) and the HIP path (GPU test) must both reproduce it, so oracle and kernels cannot drift together
unnoticed.    python tests/golden/make_env_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cavoid_oracle as po  # noqa: E402

CASES = [  # name, N, M, sort, nonlearning fraction, min agents, worlds, steps
    ("phase1", 4, 3, po.SORT_CLOSEST_LAST, 0.0, 4, 12, 90),
    ("phase1_ws", 4, 3, po.SORT_CLOSEST_FIRST, 0.4, 2, 12, 90),
    ("phase2", 10, 9, po.SORT_CLOSEST_LAST, 0.3, 2, 5, 80),
    ("tti_clip", 6, 3, po.SORT_TIME_TO_IMPACT, 0.2, 3, 6, 80),
]


def main():
    store = {"cases": np.array([c[0] for c in CASES])}
    for name, N, M, sort, nonl, gmin, W, steps in CASES:
        cfg = po.OracleConfig(max_agents=N, max_other_agents_observed=M, sort_method=sort)
        gen = po.GenConfig(min_agents=gmin, max_agents=N, nonlearning_fraction=nonl)
        seed = 31337
        worlds = [po.generate_world(seed, w, 0, cfg, gen) for w in range(W)]
        rng = np.random.default_rng(5)
        acts = rng.integers(0, 11, size=(steps, W, N)).astype(np.int32)
        acts[rng.random((steps, W, N)) < 0.7] = 2
        init = [po.world_to_arrays(wd) for wd in worlds]
        obs = np.zeros((steps, W, N, cfg.obs_width), np.float64)        # float64: the oracle's own precision (no storage rounding)
        rew = np.zeros((steps, W, N), np.float64)
        done = np.ones((steps, W, N), np.uint8)
        over = np.zeros((steps, W), np.uint8)
        flags = np.zeros((steps, W, N), np.uint32)
        for t in range(steps):
            for w, wd in enumerate(worlds):
                n = len(wd.agents)
                o, r, g, info = wd.step({i: acts[t, w, i] for i in range(n)})
                obs[t, w] = o
                rew[t, w, :n] = r
                done[t, w, :n] = [info["which_agents_done"][i] for i in range(n)]
                over[t, w] = g
                flags[t, w] = po.world_to_arrays(wd)[2]
        store.update({name + "_cfg": np.array([N, M, sort, gmin, W, steps, seed]), name + "_nonl": nonl,
                      name + "_f64": np.concatenate([i[0] for i in init], axis=1),
                      name + "_f32": np.concatenate([i[1] for i in init], axis=1),
                      name + "_flags0": np.concatenate([i[2] for i in init]),
                      name + "_actions": acts, name + "_obs": obs, name + "_rew": rew, name + "_done": done,
                      name + "_over": over, name + "_flags": flags})
    out = os.path.join(ROOT, "tests", "golden", "env_golden.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
