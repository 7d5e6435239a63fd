#!/usr/bin/env python
"""Randomised check of the fused inference kernel against the PyTorch float32 graph of the same network: random numbers of observed
agents (M = 1 .. 19), batch sizes, sequence-length mixes, weight scales; the bars of tests/test_gpu_policy.py (p 2e-5, v 2e-4).
usage: python tools/policy_soak.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rl_collision_avoidance_amd.config import EnvConfig
from rl_collision_avoidance_amd.ga3c.network import NetworkVP_rnn
from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(7)
    t0, cases, worst_p, worst_v = time.time(), 0, 0.0, 0.0
    while time.time() - t0 < budget:
        M = int(rng.integers(1, 20))                        # (kPolMaxOthers = 19)

        class Cfg(EnvConfig):
            def __init__(self):
                self.MAX_NUM_AGENTS_IN_ENVIRONMENT = M + 1
                EnvConfig.__init__(self)
        torch.manual_seed(int(rng.integers(0, 1 << 30)))
        net = NetworkVP_rnn(Cfg()).cuda()
        with torch.no_grad():
            for prm in net.parameters():
                if prm.dim() == 1:
                    prm.normal_(0.0, 0.1)                   # non-zero biases (the initialisation's are zero)
        pol = FusedPolicy(net)
        for _ in range(4):
            B = int(rng.choice([1, 2, 63, 64, 65, int(rng.integers(1, 5000)), int(rng.integers(5000, 40000))]))
            g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
            x = torch.randn((B, net.input_size), generator=g) * net.std.cpu() * float(rng.choice([0.3, 1.0, 3.0])) + net.avg.cpu()
            mode = int(rng.integers(0, 4))
            lens = (torch.randint(0, M + 1, (B,), generator=g) if mode == 0 else torch.full((B,), M) if mode == 1 else
                    torch.zeros(B, dtype=torch.long) if mode == 2 else torch.randint(0, M + 2, (B,), generator=g) - 0)   # (mode 3: one past M, clamped)
            x[:, 0] = lens.float()
            x = x.cuda()
            p, v = pol(x)
            with torch.no_grad():
                _, p_ref, v_ref = net.forward(x)
            dp, dv = (p - p_ref).abs().max().item(), (v - v_ref).abs().max().item()
            worst_p, worst_v = max(worst_p, dp), max(worst_v, dv)
            if not (dp <= 2e-5 and dv <= 2e-4 and torch.isfinite(p).all() and torch.isfinite(v).all()):
                print("MISMATCH", dict(M=M, B=B, mode=mode, dp=dp, dv=dv))
                raise SystemExit(1)
            cases += 1
        pol.close() if hasattr(pol, "close") else None
    print({"cases": cases, "worst_dp": worst_p, "worst_dv": worst_v, "seconds": round(time.time() - t0, 1),
           "result": "fused inference kernel within 2e-5 (p) / 2e-4 (v) of the PyTorch float32 graph"})


if __name__ == "__main__":
    main()
