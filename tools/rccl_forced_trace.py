#!/usr/bin/env python
"""What a rocprofv3 kernel trace is taken around (profiles/r05_*_rccl_forced_kernel_trace_stats.csv): ShardedEnv.step_and_gather at
4 x 8192 through a FORCED one-rank RCCL communicator -- K-step launches of the env step kernel on the producer stream, ncclAllGather
(to every rank) and the grouped ncclSend / ncclRecv (to the trainer rank) on the communicator's stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_collision_avoidance_amd.config import EnvConfig          # noqa: E402
from rl_collision_avoidance_amd.sharding import ShardedEnv       # noqa: E402

W, N, K = 8192, 4, 20
for root in (-1, 0):
    sh = ShardedEnv(W, EnvConfig(), device=torch.device("cuda", 0), seed=7, force_rccl=True)
    sh.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = torch.randint(0, 11, (K, W, N), generator=g, device="cuda", dtype=torch.int32)
    for _ in range(50):
        sh.gathered_blocks(sh.step_and_gather(acts, root=root))
    torch.cuda.synchronize()
    print(root, sh.gather_form, "rccl", sh._native.rccl_version)
    sh.close()
