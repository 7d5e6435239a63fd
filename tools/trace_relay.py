#!/usr/bin/env python
"""Timeline of one step of env_relay_kernel (the -DCAVOID_TRACE build): shader-clock stamps of the middle step of a launch,
per role, relative to D's iteration start; medians over the tiles.  usage: python tools/trace_relay.py [W] [N] [K]"""
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

NAMES = {0: "D iteration begins", 1: "D successor posted", 2: "D verdict arrived", 3: "D slot final",
         8: "P waits for stage", 9: "P stage arrived", 12: "P own state read", 13: "P pair pass done", 10: "P verdict posted", 11: "P outputs stored",
         16: "C waits for final", 17: "C final arrived", 18: "C ego + keys", 19: "C rows flushed"}


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    K = int(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith('-') else 32

    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
            EnvConfig.__init__(self)
    env = BatchedCollisionAvoidanceEnv(W, Cfg(), seed=7)
    lib = _lib.lib()
    lib.cavoid_debug_trace.argtypes = [C.c_void_p]
    tiles = (W + (64 // N) - 1) // (64 // N)
    trace = torch.zeros((tiles + 8, 32), dtype=torch.int64, device="cuda")
    acts = torch.randint(0, 11, (K, W, N), device="cuda", dtype=torch.int32)
    env.reset()
    for _ in range(3):
        env.step_autoreset_n(acts)
    torch.cuda.synchronize()
    assert lib.cavoid_debug_trace(C.c_void_p(trace.data_ptr())) == 0
    for rep in range(2):
        trace.zero_()
        env.step_autoreset_n(acts)
        torch.cuda.synchronize()
        t = trace.cpu().numpy()[:tiles].astype(np.int64)
        ok = t[:, 0] > 0
        t = t[ok]
        print("rep %d: %d tiles stamped (step %d of %d)" % (rep, len(t), K // 2, K))
        for k in sorted(NAMES):
            d = t[:, k] - t[:, 0]
            print("   %-26s median %6d  p10 %6d  p90 %6d" % (NAMES[k], np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    if "--launch" in sys.argv:                             # launch-level marks: where a K-step launch's fixed cost goes
        marks = {20: "D kernel entry", 25: "L kernel entry", 26: "L first action batch in LDS", 21: "D prologue done (state, table, first actions)", 22: "D step 0 posted",
                 27: "L first pool records posted", 28: "C step 0's rows flushed", 23: "D last step settled", 29: "C last step's rows flushed", 30: "C0 stores completed",
                 24: "D write-back issued"}
        t0 = np.minimum(t[:, 20], t[:, 25])                # (per tile: the shader clock is per XCD, tiles do not share a time base)
        print("launch-level marks, cycles after the tile's own first kernel entry: median / p10 / p90 / max")
        for k in marks:
            d = t[:, k] - t0
            print("   %-48s %7d %7d %7d %7d" % (marks[k], np.median(d), np.percentile(d, 10), np.percentile(d, 90), d.max()))
        span = np.maximum(np.maximum(t[:, 24], t[:, 30]), t[:, 29]) - t0
        print("   span entry -> last mark per tile: median %d max %d cycles; D loop per step (mark 23 - mark 22) / (K - 1): %.0f cycles"
              % (np.median(span), span.max(), np.median(t[:, 23] - t[:, 22]) / max(K - 1, 1)))
    env.close()


if __name__ == "__main__":
    main()
