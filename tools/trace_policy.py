"""Per-workgroup timeline of the fused policy kernel (development aid; needs the -DCAVOID_TRACE build:
python -m rl_collision_avoidance_amd.build --trace; CAVOID_LIB=tests/_variants/libcavoid_hip_trace.so)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rl_collision_avoidance_amd import _lib
from rl_collision_avoidance_amd.config import EnvConfig
from rl_collision_avoidance_amd.ga3c.network import NetworkVP_rnn
from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    net = NetworkVP_rnn(EnvConfig()).cuda()
    pol = FusedPolicy(net)
    x = torch.randn((B, net.input_size)).cuda() * net.std + net.avg
    x[:, 0] = 3.0
    nb = (B + 63) // 64
    buf = torch.zeros((nb, 16), dtype=torch.int64, device="cuda")
    lib = _lib.lib()
    lib.cavoid_policy_debug_trace.argtypes = [C.c_void_p]
    for _ in range(3):
        pol.act(x)
    assert lib.cavoid_policy_debug_trace(C.c_void_p(buf.data_ptr())) == 0
    pol.act(x)
    torch.cuda.synchronize()
    t = buf.cpu().numpy()
    t0 = t[:, 0].min()
    us = (t[:, :5] - t0) / 100.0                       # wall clock: 100 MHz
    hw = t[:, 7]
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    xcc = (hw >> 32) & 0xF
    key = xcc * 1000 + se * 100 + sh * 10 + cu
    uniq, counts = np.unique(key, return_counts=True)
    print("blocks", nb, "distinct CUs", len(uniq), "blocks/CU histogram", np.bincount(counts).tolist())
    print("start us: min %.1f p50 %.1f max %.1f" % (us[:, 0].min(), np.median(us[:, 0]), us[:, 0].max()))
    print("end   us: min %.1f p50 %.1f max %.1f" % (us[:, 4].min(), np.median(us[:, 4]), us[:, 4].max()))
    dur = us[:, 4] - us[:, 0]
    print("duration us: min %.1f p50 %.1f max %.1f" % (dur.min(), np.median(dur), dur.max()))
    ph = np.diff(us[:, :5], axis=1)
    print("phase medians us (lstm, layer1, layer2+fc1, heads):", np.round(np.median(ph, axis=0), 1).tolist())
    pairs = []
    for k in uniq[counts == 2]:
        i, j = np.nonzero(key == k)[0]
        pairs.append(sorted([dur[i], dur[j]]))
    if pairs:
        pairs = np.array(pairs)
        print("CU pairs (shorter, longer) duration p10/p50/p90: %s / %s / %s" % tuple(
            np.round(np.percentile(pairs, q, axis=0), 1).tolist() for q in (10, 50, 90)))
        for q in range(0, len(pairs), max(1, len(pairs) // 8)):
            print("   pair", np.round(pairs[q], 1).tolist())
    late = us[:, 0] > 20
    if (t[:, 12] > 0).all():
        seg = np.diff(t[:, 8:13], axis=1) / 100.0
        print("LSTM step 1 (gemm 3 chunks, barrier, cell update, barrier) us p50:", np.round(np.median(seg, axis=0), 2).tolist())
    print("prologue us p50 %.1f; shader clock p50 %.0f MHz (s_memtime cycles / wall-clock time)" % (
        np.median((t[:, 5] - t[:, 0]) / 100.0), np.median(t[:, 6] / dur)))
    for xc in range(8):
        m = xcc == xc
        print("xcc %d: blocks %d, duration p50 %.1f max %.1f" % (xc, int(m.sum()), np.median(dur[m]), dur[m].max()))
    print("blocks starting after 20 us:", int(late.sum()))
    for c in (1, 2, 3, 4):
        m = np.isin(key, uniq[counts == c])
        if m.any():
            print("CUs with %d blocks: mean duration %.1f us, last end %.1f us" % (c, dur[m].mean(), us[m, 4].max()))


if __name__ == "__main__":
    main()
