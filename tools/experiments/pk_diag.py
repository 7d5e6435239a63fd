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
STEPS = int(os.environ.get("PK_DIAG_STEPS", "2"))
SLOT = STEPS - 1
a.run_fused(STEPS)
prev = None
for _ in range(STEPS - 1):
    prev = [t.cpu() for t in eb.get_state()]
    b.step()
torch.cuda.synchronize()
st64, st32, stfl = (t.cpu() for t in eb.get_state())          # the state x[SLOT] was observed from
b.step()
torch.cuda.synchronize()
xa, xb = a.x.cpu(), b.x.cpu()
d = (xa != xb) & ~(torch.isnan(xa) & torch.isnan(xb))
d[:SLOT] = False                                                # (only the last observed slot: earlier ones were looked at with fewer steps)
if SLOT > 1:                                                    # ... and only worlds that had not parted before
    early = ((xa[1:SLOT] != xb[1:SLOT]).any(dim=0).any(dim=1)).view(-1, N).any(dim=1).repeat_interleave(N)
    d[:, early] = False
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

import math
print("\n-- what was added to vx*px in the wrong low half?  (want: vy*py of that neighbour)")
for slot, row, col in idx[:16].tolist():
    w, i = row // N, row % N
    g0 = 5 + 7 * ((col - 5) // 7)
    r_other = float(xb[slot, row, g0 + 4])
    js = [j for j in range(N) if j != i and abs(float(st32[2, w * N + j]) - r_other) < 1e-6]
    if len(js) != 1:
        print("row", row, "neighbour not identified", js); continue
    j = js[0]
    pxi, pyi, gx, gy = float(st64[0, row]), float(st64[1, row]), float(st32[0, row]), float(st32[1, row])
    d = math.hypot(gx - pxi, gy - pyi)
    ex, ey = (gx - pxi) / d, (gy - pyi) / d
    hj, sj = float(st64[2, w * N + j]), float(st32[4, w * N + j])
    vx, vy = sj * math.cos(hj), sj * math.sin(hj)
    got, want = float(xa[slot, row, col]), float(xb[slot, row, col])
    added = got - vx * ex
    if prev is not None:
        hp, sp = float(prev[0][2, w * N + j]), float(prev[1][4, w * N + j])
        print("      one step EARLIER this neighbour had vy*py(now's frame) % .6f, vx % .6f vy % .6f" % (sp * math.sin(hp) * ey, sp * math.cos(hp), sp * math.sin(hp)))
    others = {jj: float(st32[4, w * N + jj]) * math.sin(float(st64[2, w * N + jj])) * ey for jj in range(N) if jj != i}
    print("row %d nb %d: vx*px % .6f  vy*py % .6f (want % .6f = % .6f)  got % .6f -> added % .6f | -vx*py % .6f  vy*px % .6f  vx*py % .6f | vy'*py of all nbs %s"
          % (row, j, vx * ex, vy * ey, vx * ex + vy * ey, want, got, added, -vx * ey, vy * ex, vx * ey, {k: round(v, 6) for k, v in others.items()}))
