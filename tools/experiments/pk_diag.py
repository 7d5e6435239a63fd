#!/usr/bin/env python
"""What ARE the wrong values?  Replays tools/repro_actor_case.py's case for 2 steps (fused actor kernel vs step-by-step) and prints,
for the first mismatching entries of the experience rows `x`, the whole 7-feature group of that neighbour on both sides, the same
entries one step earlier, and how `got` relates to `want` (development aid for DESIGN.md 3.7 (d))."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from actor_soak import make
from repro_actor_case import CASE

c = dict(CASE)
N, W, seed, net_seed, reflush, greedy, time_max = (c.pop(k) for k in ("N", "W", "seed", "net_seed", "reflush", "greedy", "time_max"))
ea, a = make(W, N, seed, reflush, greedy, time_max, net_seed, **c)
eb, b = make(W, N, seed, reflush, greedy, time_max, net_seed, **c)
a.run_fused(2)
b.step(); b.step()
torch.cuda.synchronize()
xa, xb = a.x.cpu(), b.x.cpu()
d = (xa != xb) & ~(torch.isnan(xa) & torch.isnan(xb))
idx = d.nonzero()
print("x shape", tuple(xa.shape), "mismatches", int(d.sum()), "by ring slot", {int(k): int((idx[:, 0] == k).sum()) for k in idx[:, 0].unique()},
      "by column", {int(k): int((idx[:, 2] == k).sum()) for k in idx[:, 2].unique()})
rows = (idx[:, 1] // N).unique()
print("worlds hit", len(rows), "agents-in-world hit", {int(k): int(((idx[:, 1] % N) == k).sum()) for k in (idx[:, 1] % N).unique()},
      "lane-in-tile (row % 64)", sorted({int(r % 64) for r in idx[:, 1]})[:64])
for slot, row, col in idx[:12].tolist():
    g0 = 5 + 7 * ((col - 5) // 7)
    print("slot %d row %d (world %d agent %d) col %d:  got % .9g  want % .9g   diff % .3g" % (slot, row, row // N, row % N, col, xa[slot, row, col], xb[slot, row, col],
                                                                                  xa[slot, row, col] - xb[slot, row, col]))
    print("   group got ", ["% .6f" % v for v in xa[slot, row, g0:g0 + 7].tolist()], " num_other %g" % xa[slot, row, 0])
    print("   group want", ["% .6f" % v for v in xb[slot, row, g0:g0 + 7].tolist()])
    print("   the other groups' v_par (got/want):", [("% .6f" % xa[slot, row, 5 + 7 * k + 2], "% .6f" % xb[slot, row, 5 + 7 * k + 2]) for k in range(N - 1)])
    if slot > 0:
        print("   one step earlier  ", ["% .6f" % v for v in xb[slot - 1, row, g0:g0 + 7].tolist()])
