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
from .policy_kernel import FusedA3CTrainer, FusedPolicy
from .rollout import BatchedRollout
from .stats import EpisodeStats


def save_checkpoint(directory: str, episode: int, net: NetworkVP_rnn, trainer: A3CTrainer) -> str:
    """``NetworkVPCore.save(episode)`` (:235-236): ``<dir>/network_%08d`` -- here a torch file with the variables under
    their TensorFlow names' attributes, the Adam state and the episode count."""
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, "network_%08d.pt" % episode)
    torch.save({"episode": int(episode), "model": net.state_dict(), "optimizer": trainer.opt.state_dict(),
                "training_step": trainer.training_step}, path)
    return path


def load_checkpoint(path: str, net: NetworkVP_rnn, trainer: A3CTrainer, device) -> int:
    """``NetworkVPCore.load`` (:238-262): returns the episode number the run resumes from."""
    if path.endswith(".npz"):                                  # {TF variable name: array}, see network.load_tf_variables
        import numpy as np
        from .network import load_tf_variables
        load_tf_variables(net, dict(np.load(path)))
        return 0
    ck = torch.load(path, map_location=device)
    net.load_state_dict(ck["model"])
    if "optimizer" in ck:
        trainer.opt.load_state_dict(ck["optimizer"])
    trainer.training_step = int(ck.get("training_step", 0))
    return int(ck.get("episode", 0))


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--worlds", type=int, default=8192, help="total worlds over all GPUs")
    ap.add_argument("--agents", type=int, default=4, help="MAX_NUM_AGENTS_IN_ENVIRONMENT (4 = TrainPhase1, 10 = TrainPhase2)")
    ap.add_argument("--min-agents", type=int, default=2)
    ap.add_argument("--episodes", type=int, default=20000, help="stop after this many finished episodes (all ranks)")
    ap.add_argument("--steps-per-graph", type=int, default=4)
    ap.add_argument("--scenario-pool", type=int, default=0,
                    help="0 (default): every episode runs the scenario generator; > 0: episodes gather from a pool of that many "
                         "pre-generated scenarios (benchmark aid, NOT the reference's distribution of fresh random test cases)")
    ap.add_argument("--train-rows", type=int, default=32768,
                    help="rows per Adam step: every drained row is trained on exactly once, in minibatches of this size")
    ap.add_argument("--torch-policy", action="store_true",
                    help="act with the PyTorch-ROCm graph of the network instead of the fused MFMA kernel (rnn arch only)")
    ap.add_argument("--autograd-trainer", action="store_true",
                    help="train through PyTorch autograd instead of the fused forward/backward kernels (rnn arch only)")
    ap.add_argument("--lr", type=float, default=2e-5, help="LEARNING_RATE_RL_START")
    ap.add_argument("--lr-end", type=float, default=None, help="LEARNING_RATE_RL_END (default: no annealing)")
    ap.add_argument("--beta", type=float, default=1e-4, help="BETA_START (entropy regularisation)")
    ap.add_argument("--beta-end", type=float, default=None, help="BETA_END")
    ap.add_argument("--annealing-episodes", type=int, default=None, help="ANNEALING_EPISODE_COUNT (default: --episodes)")
    ap.add_argument("--play", action="store_true", help="PLAY_MODE: argmax actions, trainers disabled (Server.py:134-137)")
    ap.add_argument("--checkpoint-dir", default=None, help="save network_%%08d.pt here (SAVE_MODELS)")
    ap.add_argument("--save-every", type=int, default=50000, help="SAVE_FREQUENCY, in episodes")
    ap.add_argument("--load", default=None, help="checkpoint file to start from (LOAD_CHECKPOINT): a network_%%08d.pt written "
                                                 "by this CLI, or an .npz of the reference's TensorFlow variables by name")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend under torchrun (nccl = RCCL over xGMI; gloo only for dry runs)")
    ap.add_argument("--share-device", action="store_true",
                    help="dry run of the multi-rank logic on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--evaluate", type=int, default=0, metavar="ROUNDS",
                    help="EVALUATE_MODE instead of training: ROUNDS x --worlds episodes with argmax actions, every agent must "
                         "finish; prints success / collision / timeout rates (use with --load)")
    ap.add_argument("--pretrain-steps", type=int, default=0,
                    help="supervised initialisation before RL (the role of Regression.py): Adam steps on teacher-driven rollouts")
    ap.add_argument("--scenario", default="ring", choices=["ring", "box"],
                    help="scenario generator: ring = GEN v1 (antipodal goals on a ring), box = GEN v2 (random starts / goals in a box "
                         "with minimum separations: the shape of the reference's get_testcase_random, TEST_CASE_FN run-ws/config.yaml:281-283)")
    ap.add_argument("--scripted-fraction", type=float, default=0.0,
                    help="P(an agent other than agent 0 runs a scripted policy) -- the reference trained against 'static / non-coop / RVO' "
                         "mixes (checkpoints/RL/wandb/run-2018-backup/checkpoints/index.txt:1-3)")
    ap.add_argument("--static-fraction", type=float, default=0.34, help="of the scripted agents: P(static)")
    ap.add_argument("--rvo-fraction", type=float, default=0.33, help="... P(RVO / ORCA)")
    ap.add_argument("--frozen-fraction", type=float, default=0.0,
                    help="... P(frozen network: a NON-learning agent driven by a frozen NetworkVP_rnn -- the GA3C-CADRL agent, "
                         "Server.py:36); the rest are non-cooperative")
    ap.add_argument("--frozen-policy", default=None,
                    help="checkpoint of the network behind the frozen-network agents (default: a frozen copy of the starting weights)")
    ap.add_argument("--no-actor-kernel", action="store_true",
                    help="run the actors as one launch per phase in a hipGraph instead of the fused actor kernel (cavoid_actor_run)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--print-every", type=int, default=2000, help="stats line every n episodes (rank 0)")
    ap.add_argument("--faithful-reflush", action="store_true", help="keep the reference's post-done re-flush quirk")
    args = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    size = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_device:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    N = args.agents

    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
            self.TEST_CASE_GENERATOR = args.scenario
            self.SCRIPTED_AGENT_FRACTION = args.scripted_fraction
            self.SCRIPTED_STATIC_FRACTION = args.static_fraction
            self.SCRIPTED_RVO_FRACTION = args.rvo_fraction if args.scripted_fraction > 0 else 0.0
            self.SCRIPTED_FROZEN_NET_FRACTION = args.frozen_fraction if args.scripted_fraction > 0 else 0.0
            EnvConfig.__init__(self)
    cfg = Cfg()
    offset, count = shard_range(args.worlds, rank, size)
    # training draws a FRESH scenario for every episode (in-kernel generator, gen_pool_size = 0): the 65 536-entry scenario
    # pool is a latency aid for benchmarks and would make a 20k..2M-episode run revisit a fixed scenario set
    env = BatchedCollisionAvoidanceEnv(count, cfg, device=device, world_offset=offset, seed=1000 * args.seed,
                                       gen_min_agents=min(args.min_agents, N), evaluate_mode=1 if args.evaluate else 0,
                                       gen_pool_size=args.scenario_pool)
    net = NetworkVP_rnn(cfg, seed=args.seed).to(device)
    fused = None if (args.torch_policy or net.arch != "rnn") else FusedPolicy(net, seed=1000 * args.seed + rank)
    if args.autograd_trainer or net.arch != "rnn":
        trainer = A3CTrainer(net, learning_rate=args.lr)
    else:
        trainer = FusedA3CTrainer(net, fused, learning_rate=args.lr)       # shares (or creates) the packed-weight handle
    episodes_before = 0
    if args.load:
        episodes_before = load_checkpoint(args.load, net, trainer, device)
    def make_frozen():
        """the network behind the frozen-network agents: its own weights (a checkpoint, or a copy of the weights RL starts from --
        loaded or regressed), never trained"""
        if not (args.scripted_fraction > 0 and args.frozen_fraction > 0):
            return None
        if net.arch != "rnn":
            raise SystemExit("--frozen-fraction needs the rnn architecture (FusedPolicy)")
        import copy
        frozen_net = copy.deepcopy(net)
        if args.frozen_policy:
            load_checkpoint(args.frozen_policy, frozen_net, A3CTrainer(frozen_net, distributed=False), device)
        for prm in frozen_net.parameters():
            prm.requires_grad_(False)
        return FusedPolicy(frozen_net, seed=0)
    if args.evaluate:
        frozen = make_frozen()
        from .evaluate import evaluate
        if fused is not None:
            fused.refresh()
        res = evaluate(env, fused if fused is not None else net.predict_p_and_v, rounds=args.evaluate, frozen_policy=frozen)
        if rank == 0:
            print("[Evaluate] " + "  ".join("%s %.4f" % (k, v) if isinstance(v, float) else "%s %d" % (k, v) for k, v in res.items()),
                  flush=True)
        env.close()
        if size > 1:
            dist.destroy_process_group()
        return
    if args.pretrain_steps > 0 and not args.load:
        from .regression import pretrain
        info = pretrain(net, env, steps=args.pretrain_steps, log_every=50 if rank == 0 and args.print_every else 0)
        if size > 1:                                           # every replica starts RL from rank 0's regression result
            for prm in net.parameters():
                dist.broadcast(prm.data, src=0)
        if rank == 0:
            print("[Regression] done: %s" % info, flush=True)
    steps_before = trainer.training_step
    frozen = make_frozen()
    if fused is not None:
        fused.refresh(with_backward=isinstance(trainer, FusedA3CTrainer))      # weights may have come from a checkpoint
    if args.steps_per_graph < 2 or args.steps_per_graph % 2:
        raise SystemExit("--steps-per-graph must be an even number >= 2")
    # the experience ring must hold every block from 'final' (older than TIME_MAX + 2 steps) back to the last drain, i.e.
    # one replay of steps_per_graph steps, plus slack; the re-flush quirk can emit up to one duplicate per slot and step
    time_max = int(getattr(cfg, "TIME_MAX", int(4 / cfg.DT)))
    ring_len = (time_max + 2) + args.steps_per_graph + 8
    roll = BatchedRollout(env, fused if fused is not None else net.predict_p_and_v, reflush_done=args.faithful_reflush,
                          greedy=args.play, ring_len=max(ring_len, 2 * (time_max + 2) + 8), frozen_policy=frozen,
                          dup_capacity=(count * N * (args.steps_per_graph + 1) + 1024) if args.faithful_reflush else None)
    stats = EpisodeStats(print_every=args.print_every if rank == 0 else 0, agents=count)
    anneal_over = args.annealing_episodes or args.episodes
    next_save = episodes_before + args.save_every
    finished = 0
    roll.reset()
    if roll.fused_available and not args.no_actor_kernel:
        roll.capture_fused(steps_per_graph=args.steps_per_graph)       # K closed-loop steps per launch (cavoid_actor_run)
        actors = "%s, %d env steps per launch" % (roll.actor_path, args.steps_per_graph)
    else:
        roll.capture(steps_per_graph=args.steps_per_graph)
        actors = "%s; in a hipGraph, %d env steps per graph" % (
            roll.actor_path if not roll.fused_available else "one launch per phase (--no-actor-kernel)", args.steps_per_graph)
    if rank == 0:
        print("actors: " + actors, flush=True)
    done_flag = torch.zeros(1, device=device)
    t0 = time.time()
    while True:
        # linear annealing of the learning rate and the entropy weight over the episode count (Server.py:139-147)
        # (a resumed run continues where the checkpoint stopped: Server.py sets episode_count to the loaded episode)
        frac = min(episodes_before + finished, anneal_over - 1) / float(anneal_over)
        trainer.opt.param_groups[0]["lr"] = args.lr + ((args.lr_end if args.lr_end is not None else args.lr) - args.lr) * frac
        net.beta = args.beta + ((args.beta_end if args.beta_end is not None else args.beta) - args.beta) * frac
        roll.replay(1)
        batch = roll.drain(provenance=False)
        if roll.lost_blocks or batch.dropped:      # never train on a silently thinned stream
            raise RuntimeError("rollout lost training rows (%d ring blocks overwritten, %d duplicate rows dropped): "
                               "raise ring_len / dup_capacity or lower --steps-per-graph" % (roll.lost_blocks, batch.dropped))
        # multi-GPU: every rank must enter the same number of gradient all-reduces
        n_chunks = max(1, -(-len(batch) // args.train_rows)) if (len(batch) > 0 or size > 1) else 0
        if args.play:
            n_chunks = 0
        if size > 1:
            done_flag[0] = float(n_chunks)
            dist.all_reduce(done_flag, op=dist.ReduceOp.MAX)
            n_chunks = int(done_flag.item())
        for k in range(n_chunks):
            lo, hi = k * args.train_rows, min((k + 1) * args.train_rows, len(batch))
            lo = min(lo, hi)
            trainer.train(batch.x[lo:hi], batch.r[lo:hi], batch.a_index[lo:hi] if isinstance(trainer, FusedA3CTrainer) else batch.a[lo:hi])
            stats.add_training_steps(1)
        if fused is not None and n_chunks and not isinstance(trainer, FusedA3CTrainer):
            fused.refresh()                       # the actors see the new weights from the next replay on
            # (the fused trainer re-packs after every optimiser step itself)
        stats.add_episodes(roll.drain_episodes().tolist())
        finished = stats.episode_count
        if size > 1:
            done_flag[0] = float(finished)
            dist.all_reduce(done_flag)
            finished = int(done_flag.item())
        if args.checkpoint_dir and rank == 0 and not args.play and episodes_before + finished >= next_save:
            save_checkpoint(args.checkpoint_dir, episodes_before + finished, net, trainer)      # Server.save_model (:126-127)
            next_save += args.save_every
        if episodes_before + finished >= args.episodes:      # EPISODES counts from the loaded checkpoint on (Server.py:135-147)
            break
    if args.checkpoint_dir and rank == 0 and not args.play:
        save_checkpoint(args.checkpoint_dir, episodes_before + finished, net, trainer)
    if rank == 0:
        dt = time.time() - t0
        print("finished %d episodes in %.1f s: %.0f learning-agent-steps/s per GPU, rolling reward %.4f, %d training steps"
              % (finished, dt, stats.total_frame_count / dt, stats.roll_reward_log, trainer.training_step - steps_before), flush=True)
    roll.close()
    env.close()
    if size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
