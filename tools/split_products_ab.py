"""Accuracy (against the network in float64) and time of the bf16-split inference kernel for the library given in CAVOID_LIB --
development aid for the choice of CAVOID_SPLIT_PRODUCTS (csrc/cavoid_policy_split.hpp).  usage: python tools/split_products_ab.py"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rl_collision_avoidance_amd.config import EnvConfig
from rl_collision_avoidance_amd.ga3c.network import NetworkVP_rnn
from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy


def main():
    out = {}
    for M, B, scale in ((3, 32768, 1.0), (9, 8192, 1.0), (3, 32768, 4.0)):
        class Cfg(EnvConfig):
            def __init__(self):
                self.MAX_NUM_AGENTS_IN_ENVIRONMENT = M + 1
                EnvConfig.__init__(self)
        worst_p = worst_v = rel_v = 0.0
        for seed in range(3):
            net = NetworkVP_rnn(Cfg(), seed=40 + M + seed).cuda()
            pol = FusedPolicy(net)
            g = torch.Generator().manual_seed(7 + seed)
            x = (torch.randn((B, net.input_size), generator=g) * net.std.cpu() * scale + net.avg.cpu())
            x[:, 0] = torch.randint(0, M + 1, (B,), generator=g).float()
            x = x.cuda()
            p, v = pol(x)
            with torch.no_grad():
                _, p64, v64 = copy.deepcopy(net).double().forward(x.double())
            worst_p = max(worst_p, (p.double() - p64).abs().max().item())
            worst_v = max(worst_v, (v.double() - v64).abs().max().item())
            rel_v = max(rel_v, ((v.double() - v64).abs() / (1.0 + v64.abs())).max().item())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(10):
            pol.act(x)
        e0.record()
        for _ in range(100):
            pol.act(x)
        e1.record()
        torch.cuda.synchronize()
        out["M%d_B%d_s%g" % (M, B, scale)] = {"dp_max": "%.2e" % worst_p, "dv_max": "%.2e" % worst_v, "dv_rel": "%.2e" % rel_v,
                                              "kernel_us": round(e0.elapsed_time(e1) * 10, 1)}
    print(os.path.basename(os.environ.get("CAVOID_LIB", "product")), out)


if __name__ == "__main__":
    main()
