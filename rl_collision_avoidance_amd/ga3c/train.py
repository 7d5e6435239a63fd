"""``python -m rl_collision_avoidance_amd.ga3c.train`` -- the GA3C training loop with every actor on the device
(BASELINE configs[4]).  What ``Server.main()`` + 32 ``ProcessAgent`` + 2 ``ThreadPredictor`` + 2 ``ThreadTrainer``
do in the reference (/root/reference/ga3c/GA3C/Server.py:129-170) as ONE loop per GPU:

    hipGraph replay  (policy forward -> sampling -> env.step -> experience bookkeeping, k steps)
    drain rows       -> A3C loss / Adam step (gradients all-reduced over RCCL when launched with torchrun)
    drain episodes   -> the reference's stats line

Multi-GPU: ``python -m torch.distributed.run --nproc-per-node 8 -m rl_collision_avoidance_amd.ga3c.train ...``;
worlds are sharded (globally keyed RNG), the policy is replicated, one flat gradient all-reduce per step."""
from __future__ import annotations

import argparse
import os
import time

import torch
import torch.distributed as dist

from ..batched_env import BatchedCollisionAvoidanceEnv
from ..config import EnvConfig
from ..sharding import shard_range
from .network import A3CTrainer, NetworkVP_rnn
from .policy_kernel import FusedPolicy
from .rollout import BatchedRollout
from .stats import EpisodeStats


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--worlds", type=int, default=8192, help="total worlds over all GPUs")
    ap.add_argument("--agents", type=int, default=4, help="MAX_NUM_AGENTS_IN_ENVIRONMENT (4 = TrainPhase1, 10 = TrainPhase2)")
    ap.add_argument("--min-agents", type=int, default=2)
    ap.add_argument("--episodes", type=int, default=20000, help="stop after this many finished episodes (all ranks)")
    ap.add_argument("--steps-per-graph", type=int, default=4)
    ap.add_argument("--train-rows", type=int, default=32768,
                    help="rows per Adam step: every drained row is trained on exactly once, in minibatches of this size")
    ap.add_argument("--torch-policy", action="store_true",
                    help="act with the PyTorch-ROCm graph of the network instead of the fused MFMA kernel (rnn arch only)")
    ap.add_argument("--lr", type=float, default=2e-5)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--print-every", type=int, default=2000, help="stats line every n episodes (rank 0)")
    ap.add_argument("--faithful-reflush", action="store_true", help="keep the reference's post-done re-flush quirk")
    args = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    size = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    N = args.agents

    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
            EnvConfig.__init__(self)
    cfg = Cfg()
    offset, count = shard_range(args.worlds, rank, size)
    env = BatchedCollisionAvoidanceEnv(count, cfg, device=device, world_offset=offset, seed=1000 * args.seed,
                                       gen_min_agents=min(args.min_agents, N))
    net = NetworkVP_rnn(cfg, seed=args.seed).to(device)
    trainer = A3CTrainer(net, learning_rate=args.lr)
    fused = None if (args.torch_policy or net.arch != "rnn") else FusedPolicy(net, seed=1000 * args.seed + rank)
    roll = BatchedRollout(env, fused if fused is not None else net.predict_p_and_v, reflush_done=args.faithful_reflush)
    stats = EpisodeStats(print_every=args.print_every if rank == 0 else 0, agents=count)
    roll.reset()
    roll.capture(steps_per_graph=args.steps_per_graph)
    done_flag = torch.zeros(1, device=device)
    t0 = time.time()
    while True:
        roll.replay(1)
        batch = roll.drain()
        # multi-GPU: every rank must enter the same number of gradient all-reduces
        n_chunks = max(1, -(-len(batch) // args.train_rows)) if (len(batch) > 0 or size > 1) else 0
        if size > 1:
            done_flag[0] = float(n_chunks)
            dist.all_reduce(done_flag, op=dist.ReduceOp.MAX)
            n_chunks = int(done_flag.item())
        for k in range(n_chunks):
            lo, hi = k * args.train_rows, min((k + 1) * args.train_rows, len(batch))
            lo = min(lo, hi)
            trainer.train(batch.x[lo:hi], batch.r[lo:hi], batch.a[lo:hi])
            stats.add_training_steps(1)
        if fused is not None and n_chunks:
            fused.refresh()                       # the actors see the new weights from the next replay on
        stats.add_episodes(roll.drain_episodes().tolist())
        finished = stats.episode_count
        if size > 1:
            done_flag[0] = float(finished)
            dist.all_reduce(done_flag)
            finished = int(done_flag.item())
        if finished >= args.episodes:
            break
    if rank == 0:
        dt = time.time() - t0
        print("finished %d episodes in %.1f s: %.0f learning-agent-steps/s per GPU, rolling reward %.4f, %d training steps"
              % (finished, dt, stats.total_frame_count / dt, stats.roll_reward_log, trainer.training_step), flush=True)
    roll.close()
    env.close()
    if size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
