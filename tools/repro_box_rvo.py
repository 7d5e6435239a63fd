#!/usr/bin/env python
"""The one kind of HIP-vs-oracle divergence the 1-G-agent-step parity stress found (tests/parity_stress.py, seeds 21012 / 41012 of the
GEN v2 + ORCA case): is it a tie of the ORCA linear programme -- a configuration where a perturbation of the positions by 1e-13 m
makes the ORACLE ITSELF choose a different velocity -- or a difference between the two implementations?
For the first step at which HIP and the C oracle disagree, the oracle is re-run from the same pre-step state with the positions
of that world perturbed by +-1e-13 m (the size of the state differences the two transcendental libraries produce anyway, DESIGN.md
section 0) and the spread of the resulting positions is printed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import c_oracle as co
from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
from rl_collision_avoidance_amd.config import EnvConfig


def run(seed, chunk, N=10, W=512, steps=256, nonl=0.4, sort=1, mode=1, rvo=0.3, pool=0):
    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
            EnvConfig.__init__(self)
    env = BatchedCollisionAvoidanceEnv(W, Cfg(), seed=seed, gen_min_agents=2, gen_nonlearning_fraction=nonl, sort_method=sort, gen_pool_size=pool,
                                       gen_mode=mode, gen_rvo_fraction=rvo, rvo_enabled=1)
    ocfg = co.default_cfg(N, sort_method=sort)
    ogen = co.default_gen(2, N, nonl, pool_size=pool, mode=mode, rvo_fraction=rvo)
    env.reset()
    st = co.State.empty(W, N)
    ep = np.zeros(W, np.uint32)
    co.generate(ocfg, ogen, seed, st, ep)
    rng = np.random.default_rng(seed)
    block = None
    for t in range(steps):
        if t % chunk == 0:                                  # the action stream of the stress case (drawn `chunk` steps at a time)
            block = rng.integers(0, 11, size=(chunk, W, N)).astype(np.int32)
            block[rng.random((chunk, W, N)) < 0.75] = 2
        acts = block[t % chunk][None]
        st0, ep0 = st.copy(), ep.copy()
        obs, rew, done, go = [x.cpu().numpy() for x in env.step_autoreset(torch.from_numpy(acts[0]).cuda())]   # ONE step per launch
        oobs, orew, odone, ogo = co.step_autoreset(ocfg, ogen, seed, st, ep, acts[0])
        f64 = env.get_state()[0].cpu().numpy()
        dpos = np.abs(f64[:2] - st.f64[:2]).max(axis=0).reshape(W, N)
        if dpos.max() > 1e-9 and not ogo[np.unravel_index(dpos.argmax(), dpos.shape)[0]]:
            w, i = np.unravel_index(dpos.argmax(), dpos.shape)
            pol = (st0.flags.reshape(W, N)[w] >> 8) & 7
            print("seed %d: step %d world %d: HIP and oracle positions differ by %.3g m (agent %d, scripted policy %d; policies of the world: %s)"
                  % (seed, t, w, dpos.max(), i, pol[i], pol.tolist()))
            # the oracle against itself under perturbations of this world's positions
            sl = slice(w * N, (w + 1) * N)
            outs = []
            prng = np.random.default_rng(1)
            for trial in range(64):
                s2, e2 = st0.copy(), ep0.copy()
                if trial:
                    s2.f64[0, sl] += prng.choice([-1e-13, 1e-13], N)
                    s2.f64[1, sl] += prng.choice([-1e-13, 1e-13], N)
                co.step_autoreset(ocfg, ogen, seed, s2, e2, acts[0])
                outs.append(s2.f64[:2, sl].copy())
            outs = np.array(outs)
            spread = np.abs(outs - outs[0]).max(axis=(1, 2))
            print("   oracle vs oracle with the world's positions perturbed by +-1e-13 m: %d of 63 perturbed runs move an agent by more than 1e-6 m "
                  "(largest change %.3g m); HIP's result equals one of them to 1e-9: %s"
                  % (int((spread[1:] > 1e-6).sum()), spread.max(), bool((np.abs(outs - f64[:2, sl]).max(axis=(1, 2)) < 1e-9).any())))
            return
    print("seed", seed, ": no divergence in", steps, "steps")


if __name__ == "__main__":
    for seed in [int(x) for x in sys.argv[1:]] or [21012, 41012]:
        run(seed, 8)
