#!/usr/bin/env python
"""Where does one env-step wavefront spend its time?  Runs the -DCAVOID_TRACE build
(CAVOID_LIB=tests/_variants/libcavoid_hip_trace.so) and prints, per phase, the
shader-clock deltas (median / p90 / max over wavefronts) plus the launch span."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CAVOID_LIB", os.path.join(ROOT, "tests", "_variants", "libcavoid_hip_trace.so"))

import numpy as np
import torch

from rl_collision_avoidance_amd import _lib
from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
from rl_collision_avoidance_amd.config import EnvConfig

NAMES = ["start", "loads issued", "step begins", "dynamics", "ego+pairs", "reward/restart", "obs in LDS", "tile flushed", "stores issued"]


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # steps per launch (stamps 2..7 = the LAST step's)
    over = {"gen_min_agents": int(sys.argv[4])} if len(sys.argv) > 4 else {}   # worlds with gen_min..N agents present

    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
            EnvConfig.__init__(self)
    env = BatchedCollisionAvoidanceEnv(W, Cfg(), seed=7, **over)
    lib = _lib.lib()
    lib.cavoid_debug_trace.argtypes = [C.c_void_p]
    waves = (W + (64 // N) - 1) // (64 // N) + 8
    trace = torch.zeros((waves, 16), dtype=torch.int64, device="cuda")
    acts = torch.randint(0, 11, (32, W, N), device="cuda", dtype=torch.int32)
    env.reset()
    for _ in range(10):                                        # (past the first, synchronised wave of restarts)
        env.step_autoreset_n(acts)
    torch.cuda.synchronize()
    assert lib.cavoid_debug_trace(C.c_void_p(trace.data_ptr())) == 0
    for rep in range(3):
        trace.zero_()
        if K == 1:
            env.step_autoreset(acts[rep])
        else:
            env.step_autoreset_n(acts[:K])
        torch.cuda.synchronize()
        t = trace.cpu().numpy()[: waves - 8].astype(np.int64)
        used = list(range(9))
        if K > 1:
            print("   per-wave (end - start) / K steps: median %.0f clk" % np.median((t[:, 8] - t[:, 0]) / K))
        t0 = t[:, 0].min()
        print("rep %d: launch span (first wave start -> last wave end) = %d clk; per-wave total median %d max %d"
              % (rep, t[:, 8].max() - t0, np.median(t[:, 8] - t[:, 0]), (t[:, 8] - t[:, 0]).max()))
        print("   wave start spread: %d clk" % (t[:, 0].max() - t0))
        if K > 1 and (t[:, 1] > 0).all():      # the two-wavefront pipeline: producer stamps 2,3,4,5,8,1 / consumer 6,9,10,7,0
            for a, b, nm in ((2, 3, "P decode+dynamics"), (3, 4, "P stage+pairs"), (4, 5, "P rewards/restart"), (5, 8, "P hand-over"),
                             (8, 1, "P wait at barrier"), (6, 9, "C ego+ranking"), (9, 10, "C rows"), (10, 7, "C flush"), (7, 0, "C wait at barrier")):
                d = t[:, b] - t[:, a]
                print("   [pipe] %-20s median %6d  p90 %6d" % (nm, np.median(d), np.percentile(d, 90)))
        for a, b in ((6, 9), (9, 10), (10, 7)):
            d = t[:, b] - t[:, a]
            print("   [obs] %d -> %d median %6d  p90 %6d" % (a, b, np.median(d), np.percentile(d, 90)))
        for a, b in zip(used[:-1], used[1:]):
            d = t[:, b] - t[:, a]
            print("   %-16s -> %-16s median %6d  p90 %6d  max %6d" % (NAMES[a], NAMES[b], np.median(d), np.percentile(d, 90), d.max()))
    env.close()


if __name__ == "__main__":
    main()
