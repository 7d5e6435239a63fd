#!/usr/bin/env python
"""bench.py -- agent-steps/s of the batched env.step hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path (decode, dynamics, pairwise sensing, rewards, done flags, sorted
observation, in-kernel restart of finished worlds) over one batch of synthetic worlds: BASELINE
configs[1], 4 agents x 8192 worlds per GPU, unicycle dynamics, random actions pre-generated on the
device.  The K timed steps go through `cavoid_step_autoreset_n`: launches of up to --slices steps, the
world state staying in registers between the steps of a launch; EVERY step reads its action slice and
writes its observations, rewards, done flags and game_over.  Weak scaling: every rank steps its own
8192 worlds (RNG keyed on global world ids); no data-path collective in the headline region (for N>1
the packed (obs|reward|done) all-gather of configs[2] -- `cavoid_gather_*`, RCCL -- is timed afterwards
and reported under "extra").  `python bench.py --gpus N` without a launcher starts its own N ranks
(torch.distributed.run, 127.0.0.1).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E vendor peak (MI355X_MICROARCH.md); ~6300 achievable


def algorithmic_bytes_per_agent_step(M: int) -> int:
    """fp32 words -- state read 11 + action 1 + state write 7 + obs (2+4+7M)
    + reward 1 + done 1.  192 B at M=3, 360 B at M=9."""
    return 4 * (11 + 1 + 7 + (6 + 7 * M) + 1 + 1)


def cpu_baseline(N: int, W: int, budget_s: float, pool_size: int):
    """The C float64 oracle (a port/restatement -- the reference env source is absent) timed on this
    box's host cores, 1 thread, on the same workload shape, for about `budget_s` seconds."""
    import numpy as np
    from oracle import c_oracle as co
    cfg, gen = co.default_cfg(N), co.default_gen(N, N, pool_size=pool_size)
    st = co.State.empty(W, N)
    ep = np.zeros(W, np.uint32)
    co.generate(cfg, gen, 0, st, ep)
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 11, size=(16, W, N)).astype(np.int32)
    co.step_autoreset(cfg, gen, 0, st, ep, acts[0])
    n, t0 = 0, time.perf_counter()
    while True:
        co.step_autoreset(cfg, gen, 0, st, ep, acts[n % 16])
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    return {"value": W * N * n / dt, "unit": "agent-steps/s", "cores": 1, "kind": "port",
            "sample": "C float64 oracle (oracle/cavoid_oracle.c), %d agents x %d worlds, %d autoreset steps, %.1f s, 1 thread"
                      % (N, W, n, dt)}


def cpu_baseline_all_cores(N: int, W: int, budget_s: float, pool_size: int, max_procs: int = 64):
    """the same C oracle on many host cores at once (one process per core, the worlds split
    evenly -- how the reference scales: one env per ProcessAgent process, ProcessAgent.py:221).  Plain
    subprocesses (`bench.py --cpu-worker ...`): nothing is forked from the process that holds the HIP context."""
    import subprocess
    procs = max(1, min(max_procs, os.cpu_count() or 1))
    per = max(1, W // procs)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", str(N), str(per), str(budget_s), str(pool_size)]
    children = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(procs)]
    vals = []
    deadline = time.time() + budget_s + 90.0
    for ch in children:
        try:
            out, _ = ch.communicate(timeout=max(1.0, deadline - time.time()))
            vals.append(float(out.strip().splitlines()[-1]))
        except Exception:      # noqa: BLE001 -- a straggler must not cost the headline
            ch.kill()
    return {"value": float(sum(vals)), "unit": "agent-steps/s", "cores": len(vals), "kind": "port",
            "sample": "C float64 oracle, %d processes x %d worlds x %d agents, %.1f s each" % (len(vals), per, N, budget_s)}


def python_baseline(N: int, budget_s: float):
    """Reference-style per-object Python/NumPy oracle, one world, one process (baseline B1)."""
    import numpy as np
    from oracle import cavoid_oracle as po
    cfg = po.OracleConfig(max_agents=N, max_other_agents_observed=N - 1)
    gen = po.GenConfig(min_agents=N, max_agents=N)
    rng = np.random.default_rng(0)
    steps, ep, t0 = 0, 0, time.perf_counter()
    world = po.generate_world(0, 0, ep, cfg, gen)
    while time.perf_counter() - t0 < budget_s:
        _, _, over, _ = world.step({i: int(rng.integers(0, 11)) for i in range(N)})
        steps += 1
        if over:
            ep += 1
            world = po.generate_world(0, 0, ep, cfg, gen)
    dt = time.perf_counter() - t0
    return {"value": steps * N / dt, "unit": "agent-steps/s", "cores": 1,
            "sample": "oracle/cavoid_oracle.py, 1 world x %d agents, %d steps" % (N, steps)}


def full_loop(env_cls, cfg, device, W, N, rank, world_size, sync_all, steps: int = 240, train_rows: int = 32768, brief: bool = False):
    """BASELINE configs[4]: everything on the device -- policy inference + action selection (fused f32-MFMA kernel,
    cavoid_policy_*), env.step, experience store / n-step returns (HIP), and Adam steps (PyTorch-ROCm autograd) on
    EVERY drained row, in minibatches of `train_rows` (policy replica per GPU, no collective).  Reports the
    reference's PPS definition: learning-agent steps per second (ProcessStats.py:54-56), for four regimes:
    actors only (PLAY_MODE: no trainer); the full loop with the fused trainer pass (cavoid_policy_train + library
    weight-gradient GEMMs); with the PyTorch autograd trainer; and all-PyTorch (policy and trainer)."""
    import torch
    from rl_collision_avoidance_amd.ga3c.network import A3CTrainer, NetworkVP_rnn
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedA3CTrainer, FusedPolicy
    from rl_collision_avoidance_amd.ga3c.rollout import BatchedRollout
    per_graph = int(os.environ.get("CAVOID_STEPS_PER_GRAPH", "8"))   # env steps per hipGraph replay (even): one drain (a host sync) per replay

    def regime(fused: bool, train: bool, fused_trainer: bool = False):
        env = env_cls(W, cfg, device=device, world_offset=rank * W, seed=11)
        net = NetworkVP_rnn(cfg).to(device)
        pol = FusedPolicy(net, seed=rank) if fused else None
        # a policy replica per GPU and NO collective in this extra: ranks drain different row counts, so the number of
        # optimiser steps differs between them
        trainer = FusedA3CTrainer(net, pol, distributed=False) if fused_trainer else A3CTrainer(net, distributed=False)
        roll = BatchedRollout(env, pol if fused else net.predict_p_and_v, reflush_done=False)
        roll.reset()
        roll.capture(steps_per_graph=per_graph)              # policy + sampling + env.step + bookkeeping as ONE graph
        rows = [0]

        def run(n_replays):
            for _ in range(n_replays):
                roll.replay(1)
                b = roll.drain(provenance=False)             # hand-over of the rows that became training rows (the PPS numerator)
                rows[0] += len(b)
                if not train:
                    continue
                for lo in range(0, len(b), train_rows):
                    trainer.train(b.x[lo:lo + train_rows], b.r[lo:lo + train_rows],
                                  b.a_index[lo:lo + train_rows] if fused_trainer else b.a[lo:lo + train_rows])
                if pol is not None and len(b) and not fused_trainer:
                    pol.refresh()
            return n_replays * per_graph * W * N
        run(8)
        rows[0] = 0
        sync_all()
        t0 = time.perf_counter()
        policy_rows = run(steps // per_graph)
        sync_all()
        dt = time.perf_counter() - t0
        n = (steps // per_graph) * per_graph
        # PPS as the reference counts it (ProcessStats.py:54-56; ProcessAgent.py:237): experiences handed to the trainer.
        # Agents that are done and wait for their world to end still occupy a policy row but yield nothing.
        out = {"learning_agent_steps_per_s_per_gpu": rows[0] / dt, "policy_rows_per_s_per_gpu": policy_rows / dt,
               "ms_per_env_step": dt * 1e3 / n, "env_steps": n, "rows_handed_over": rows[0],
               "training_steps": trainer.training_step if train else 0}
        roll.close()
        env.close()
        return out

    def policy_kernel():
        """The fused inference kernel alone, on real observations: HIP events on the launch stream."""
        env = env_cls(W, cfg, device=device, world_offset=rank * W, seed=11)
        net = NetworkVP_rnn(cfg).to(device)
        pol = FusedPolicy(net, seed=rank)
        obs = env.reset().view(W * N, -1)[:, 1:]
        M = net.max_others

        def timed(fn, n=100):
            for _ in range(10):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize(device)
            return e0.elapsed_time(e1) * 1e3 / n
        fused_us = timed(lambda: pol.act(obs))
        torch_us = timed(lambda: net.predict_p_and_v(obs.contiguous()), 30)
        lens = obs[:, 0].clamp(0, M)
        # The inference kernel computes the float32 GEMMs by error-free bf16 splitting (csrc/cavoid_policy_split.hpp): every
        # float32 product is five bf16 partial products accumulated in float32.  Two figures:
        #  * f32-equivalent: the float32 work of the graph as the float32-MFMA kernel would issue it (16-wide K chunks x 256
        #    columns; LSTM step 0 needs the input chunk only; a 64-row tile runs as many LSTM steps as its longest row
        #    needs), against the float32-MFMA peak the same arithmetic would otherwise be bound by;
        #  * issued: the bf16 MFMA work really issued (32-wide K chunks x 5 partial products), against the dense bf16 peak.
        steps = lens.view(-1, 64).max(dim=1).values.mean().item() if (W * N) % 64 == 0 else float(M)
        chunks = (1 + 5 * max(steps - 1, 0)) + 5 + 16 + 16 + 1
        flop = W * N * chunks * 16 * 256 * 2
        split = os.environ.get("CAVOID_POLICY_F32", "0") in ("", "0")
        chunks32 = (1 + 3 * max(steps - 1, 0)) + 3 + 8 + 8
        flop_bf16 = W * N * 5 * 32 * 2 * (chunks32 * 256 + 8 * 16)
        env.close()
        out = {"rows": W * N, "kernel_us": fused_us, "pytorch_graph_us": torch_us,
               "f32_equivalent_TFLOPs": flop / fused_us * 1e-6, "f32_mfma_peak_TFLOPs": 157.3,
               "f32_equivalent_frac_of_f32_mfma_peak": flop / fused_us * 1e-6 / 157.3}
        if split:
            out.update({"issued_TFLOPs": flop_bf16 / fused_us * 1e-6, "peak_TFLOPs": 2500.0, "frac": flop_bf16 / fused_us * 1e-6 / 2500.0,
                        "bound": "mfma", "dtype": "f32 in/out; bf16 x 5 error-free split products, f32 accumulate",
                        "kernel": "cavoid::policy_forward_split_kernel"})
        else:
            out.update({"issued_TFLOPs": flop / fused_us * 1e-6, "peak_TFLOPs": 157.3, "frac": flop / fused_us * 1e-6 / 157.3,
                        "bound": "mfma", "dtype": "f32", "kernel": "cavoid::policy_forward_kernel"})
        return out
    res = {"policy_kernel": policy_kernel(), "actors_only_fused_policy": regime(True, False),
           "full_loop_fused_policy_fused_trainer": regime(True, True, True)}
    if not brief:                                            # the PyTorch comparison legs take most of the time
        res["full_loop_fused_policy_autograd_trainer"] = regime(True, True)
        res["full_loop_torch_policy_autograd_trainer"] = regime(False, True)
    res.update({
           "steps_per_graph": per_graph, "train_rows_per_adam_step": train_rows, "policy_dtype": "f32",
           "note": "one hipGraph per %d env steps (policy + action selection + env + experience store); every drained row "
                   "is trained on once; reference PPS datum: 563 (32 procs, laptop CPU)" % per_graph})
    return res


def step_kernel_name(N: int, W: int, spl: int) -> str:
    """Which instantiation cavoid_step_autoreset_n takes (csrc/cavoid_capi.hip, cavoid_launch.hpp)."""
    if spl == 1:
        return "cavoid::env_kernel<%d, MODE_STEP_AUTORESET, false>" % N
    if W * N > 131072:
        return "cavoid::env_kernel<%d, MODE_STEP_AUTORESET_N, false>" % N
    tiles = -(-W // (64 // N))
    pipe = os.environ.get("CAVOID_PIPELINE", "2")
    if tiles <= 512 and N <= 6 and pipe not in ("0", "1"):
        return "cavoid::env_relay_kernel<%d> (roles on the wavefronts of one workgroup per tile)" % N
    if tiles <= 1024 and pipe != "0":
        return "cavoid::env_pipe_kernel<%d, false> (two-wavefront pipeline per tile)" % N
    return "cavoid::env_kernel<%d, MODE_STEP_AUTORESET_PF, false>" % N


def self_launch(args) -> None:
    """`python bench.py --gpus N` with no launcher around it: start N ranks of this script through
    torch.distributed.run on 127.0.0.1 and pass their output through (rank 0 prints the JSON line)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")         # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    raise SystemExit(subprocess.call(cmd, env=env))


def pmc_child(args) -> None:
    """Child of `measure_traffic` (runs under `rocprofv3 --pmc ...`): the bench's launch pattern, nothing else."""
    import torch
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    from rl_collision_avoidance_amd.config import EnvConfig
    N, W = args.agents, args.worlds

    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = N
            EnvConfig.__init__(self)
    env = BatchedCollisionAvoidanceEnv(W, Cfg(), device="cuda:0", seed=7)
    g = torch.Generator(device="cuda").manual_seed(1234)
    acts = torch.randint(0, env.num_actions, (args.slices, W, N), generator=g, device="cuda", dtype=torch.int32)
    env.reset()
    done = 0
    while done < args.warmup + args.steps:
        n = min(args.slices, args.warmup + args.steps - done)
        env.step_autoreset_n(acts, n)
        done += n
    torch.cuda.synchronize()
    env.close()


def measure_traffic(N: int, W: int, slices: int, steps: int, timeout_s: float = 150.0):
    """HBM bytes per launch of the step kernel from the PMC counters, collected live: one `rocprofv3 --pmc` pass per
    counter (FETCH_SIZE and WRITE_SIZE do not fit one pass; only --kernel-trace beside --pmc) around a child that
    repeats this bench's launch pattern.  Counters are KiB; gfx950 tallies 128-B read requests as 64 B, so FETCH_SIZE
    is doubled (MI355X_MICROARCH.md, HBM section).  Returns None when rocprofv3 is not usable here."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    res = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="cavoid_pmc_", dir="/tmp")
        cmd = [prof, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.abspath(__file__), "--pmc-child", "--agents", str(N), "--worlds", str(W), "--slices", str(slices),
               "--steps", str(steps), "--warmup", str(slices)]
        try:
            subprocess.run(cmd, check=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=timeout_s,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            total, n = 0.0, 0
            for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        # the stepping instantiations: env_kernel<N, MODE in {1 single step, 4 / 5 step loop}, RVO>
                        hit = re.search(r"env_kernel<%d, (\d+)" % N, row["Kernel_Name"])
                        step = (hit and hit.group(1) in ("1", "4", "5")) or ("env_pipe_kernel<%d," % N) in row["Kernel_Name"] or ("env_relay_kernel<%d>" % N) in row["Kernel_Name"]
                        if step and row["Counter_Name"] == ctr:
                            total += float(row["Counter_Value"])
                            n += 1
            if n == 0:
                return None
            res[ctr] = (total / n, n)
        except Exception:      # noqa: BLE001 -- measurement aid only
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch_kib, nf = res["FETCH_SIZE"]
    write_kib, nw = res["WRITE_SIZE"]
    return {"traffic": 2.0 * fetch_kib * 1024.0 + write_kib * 1024.0, "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
            "dispatches": min(nf, nw), "steps_per_launch": min(slices, steps),
            "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace around `bench.py --pmc-child`; "
                      "mean per step-kernel dispatch; FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B)"}


def main() -> None:
    if len(sys.argv) >= 6 and sys.argv[1] == "--cpu-worker":        # child of cpu_baseline_all_cores: no torch, no GPU
        print(cpu_baseline(int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5]))["value"])
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--worlds", type=int, default=8192, help="worlds per GPU")
    ap.add_argument("--agents", type=int, default=4)
    ap.add_argument("--slices", type=int, default=64, help="distinct pre-generated action slices = max steps per launch")
    ap.add_argument("--no-gather", action="store_true",
                    help="N>1: skip the extra measurement of the per-step RCCL all-gather of (obs,reward,done) (configs[2])")
    ap.add_argument("--sweep", action="store_true", help="add a worlds-per-GPU saturation sweep to the JSON line")
    ap.add_argument("--full-loop", action="store_true",
                    help="also time BASELINE configs[4]: batched env + NetworkVP_rnn policy + rollout bookkeeping + Adam steps")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-loop", action="store_true", help="skip the brief configs[4] extra of the default N = 1 run")
    ap.add_argument("--no-configs3", action="store_true", help="skip the brief configs[3] (10 agents x 8192 worlds) extra")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect roofline.traffic live with rocprofv3 --pmc")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N>1 (nccl = RCCL over xGMI; gloo only for dry runs of the N>1 code path)")
    ap.add_argument("--share-device", action="store_true",
                    help="dry run: every rank uses cuda:0 (exercises the multi-rank logic on a 1-GPU box; needs --backend gloo)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launch / rendezvous / collective check of the N>1 path without touching a GPU (CPU boxes, gloo)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world_size))
    if args.rendezvous_only:
        # what every N>1 run does around the timed region, with no GPU in it: rendezvous, barrier, MAX over ranks
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world_size > 1:
            dist.init_process_group("gloo")
            t = torch.tensor([1.0 + rank], dtype=torch.float64)
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            assert float(t.item()) == float(world_size)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"rendezvous": "ok", "n_gpus": world_size, "backend": "gloo"}), flush=True)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    from rl_collision_avoidance_amd.config import EnvConfig
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    N, W = args.agents, args.worlds

    def cfg_for(n_agents):
        class Cfg(EnvConfig):
            def __init__(self):
                self.MAX_NUM_AGENTS_IN_ENVIRONMENT = n_agents
                EnvConfig.__init__(self)
        return Cfg()

    def sync_all():
        torch.cuda.synchronize(device)
        if world_size > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    def make(Wl, n_agents=N, **over):
        env = BatchedCollisionAvoidanceEnv(Wl, cfg_for(n_agents), device=device, world_offset=rank * Wl, seed=1000 * 0 + 7, **over)
        g = torch.Generator(device=device)
        g.manual_seed(1234 + rank)
        acts = torch.randint(0, env.num_actions, (args.slices, Wl, n_agents), generator=g, device=device, dtype=torch.int32)
        env.reset()
        return env, acts

    def run_steps(env, acts, k):
        """k auto-reset steps in launches of up to T = --slices steps (step t of a launch reads acts[t])."""
        T = acts.shape[0]
        done = 0
        while done < k:
            n = min(T, k - done)
            env.step_autoreset_n(acts, n)
            done += n

    def kernel_figures(env, acts, n_agents, Wl, k):
        """per-launch HIP-event durations of the step kernel, in launches shaped like the timed region's"""
        spl = min(acts.shape[0], k)
        launch_ms = env.kernel_time_ms(acts, max(k, spl), spl)
        bytes_per_launch = algorithmic_bytes_per_agent_step(n_agents - 1) * Wl * n_agents * spl
        achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "kernel": step_kernel_name(n_agents, Wl, spl),
                "steps_per_launch": spl, "kernel_us": launch_ms * 1e3, "kernel_us_per_step": launch_ms * 1e3 / spl,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "algorithmic_bytes_per_agent_step": algorithmic_bytes_per_agent_step(n_agents - 1)}

    env, acts = make(W)
    run_steps(env, acts, args.warmup)

    # ---- timed region: EXACTLY K steps, barrier + synchronize on both sides, max over ranks -------
    sync_all()
    t0 = time.perf_counter()
    run_steps(env, acts, args.steps)
    sync_all()
    elapsed = time.perf_counter() - t0
    if world_size > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    value = world_size * W * N * args.steps / elapsed

    # ---- roofline of the dominant (only) kernel ------------------------------------------------------
    roofline = kernel_figures(env, acts, N, W, args.steps)
    single = env.kernel_time_ms(acts, min(args.steps, 256), 1)          # the closed-loop form: one step per launch
    roofline["single_step_launch_us"] = single * 1e3
    roofline["single_step_launch_frac"] = algorithmic_bytes_per_agent_step(N - 1) * W * N / (single * 1e-3) / 1e9 / HBM_PEAK_GBS

    extra = {}
    if world_size > 1 and not args.no_gather:
        # BASELINE configs[2]: the one real exchange of the path -- the packed (obs | reward | done) record of every world
        # back to every rank, ONE all-gather per step.  nccl: the kernel writes the packed record and cavoid_gather_* issues
        # ncclAllGather on the communicator's own stream, gather(t) overlapping step(t+1).  gloo dry run: torch.distributed.
        # Timed OUTSIDE the headline region; a failure here must not cost the headline number.
        try:
            from rl_collision_avoidance_amd.sharding import ShardedEnv
            sh = ShardedEnv(world_size * W, cfg_for(N), device=device, seed=7)
            sh.reset()
            native = args.backend == "nccl"

            def one(t):
                if native:
                    sh.gathered(sh.step_and_gather(acts[t % acts.shape[0]]))
                else:
                    sh.step_autoreset(acts[t % acts.shape[0]])
                    sh.gather()
            n_g = 200 if native else 10
            for t in range(20 if native else 2):
                one(t)
            sync_all()
            tg = time.perf_counter()
            for t in range(n_g):
                one(t)
            sync_all()
            t_both = (time.perf_counter() - tg) / n_g
            tg = time.perf_counter()
            for t in range(n_g):
                sh.step_autoreset_packed(acts[t % acts.shape[0]], sh._send[0]) if native else sh.step_autoreset(acts[t % acts.shape[0]])
            sync_all()
            t_step = (time.perf_counter() - tg) / n_g
            extra["allgather"] = {"path": "cavoid_gather_* (ncclAllGather, own stream, double-buffered)" if native else "torch.distributed (gloo dry run)",
                                  "ms_per_step_with_gather": t_both * 1e3, "ms_per_single_step_launch_alone": t_step * 1e3,
                                  "pack_cost_ms": 0.0 if native else None,
                                  "bytes_per_rank": W * N * (env.obs_width + 2) * 4,
                                  "bytes_received_per_rank": world_size * W * N * (env.obs_width + 2) * 4,
                                  "agent_steps_per_s_with_gather": world_size * W * N / t_both}
            sh.close()
        except Exception as exc:      # noqa: BLE001 -- report, never lose the headline
            extra["allgather"] = {"error": repr(exc)}

    if rank == 0 and world_size == 1 and not args.no_pmc:
        try:
            pmc = measure_traffic(N, W, args.slices, max(args.slices * 4, 256))
            if pmc is not None:
                # the PMC child's launches are `slices` steps long; scale to this run's launch length
                per_step = pmc["traffic"] / pmc["steps_per_launch"]
                roofline["traffic"] = per_step * roofline["steps_per_launch"]
                roofline["traffic_source"] = pmc
        except Exception:      # noqa: BLE001
            pass
    if roofline["traffic"] is None:
        # fall back to the committed summary of a PMC run of this command (same kernel and shape), naming it
        try:
            import glob
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r02*_pmc_traffic*.json")), reverse=True):
                pmc = json.load(open(path))
                if pmc.get("worlds") == W and pmc.get("agents") == N and "traffic_bytes_per_step" in pmc:
                    roofline["traffic"] = pmc["traffic_bytes_per_step"] * roofline["steps_per_launch"]
                    roofline["traffic_source"] = os.path.relpath(path, ROOT)
                    break
        except Exception:      # noqa: BLE001
            pass

    if rank == 0 and not args.no_configs3 and N != 10:
        # BASELINE configs[3] beside the headline: 10 agents (TrainPhase2 shape, 2..10 agents per world, M = 9) x 8192 worlds
        try:
            e3, a3 = make(8192, 10, gen_min_agents=2)
            run_steps(e3, a3, 256)                          # (past the first, synchronised wave of restarts)
            torch.cuda.synchronize(device)
            t3 = time.perf_counter()
            run_steps(e3, a3, 640)
            torch.cuda.synchronize(device)
            dt3 = time.perf_counter() - t3
            r3 = kernel_figures(e3, a3, 10, 8192, 640)
            s3 = e3.kernel_time_ms(a3, 128, 1)
            c3 = {"workload": "BASELINE configs[3]: 10 agents (2..10 present) x 8192 worlds, M = 9, obs width 69",
                  "value": 8192 * 10 * 640 / dt3, "unit": "agent-steps/s", "ms_per_step": dt3 * 1e3 / 640, "roofline": r3,
                  "single_step_launch_us": s3 * 1e3,
                  "single_step_launch_frac": 360 * 81920 / (s3 * 1e-3) / 1e9 / HBM_PEAK_GBS}
            if not args.no_cpu_baseline:
                c3["cpu_baseline"] = cpu_baseline(10, 2048, min(3.0, args.cpu_seconds), int(e3.cfg.gen_pool_size))
            extra["configs3_n10"] = c3
            e3.close()
            del e3, a3
        except Exception as exc:      # noqa: BLE001
            extra["configs3_n10"] = {"error": repr(exc)}

    if rank == 0 and args.sweep:
        # transparency: the same step with NO scenario pool (every restart runs GEN v1 in-kernel, exact (world,
        # episode) scenarios) -- the pool only moves scenario generation (E2) off the step's critical path
        try:
            e0, a0 = make(W, N, gen_pool_size=0)
            run_steps(e0, a0, 128)
            extra["no_scenario_pool"] = kernel_figures(e0, a0, N, W, 256)
            e0.close()
        except Exception as exc:      # noqa: BLE001
            extra["no_scenario_pool"] = {"error": repr(exc)}

    if args.full_loop or not args.no_full_loop:
        # configs[4] beside the headline (a brief version: actors only + the fused-trainer loop, ~15 s; the PyTorch comparison
        # legs only with --full-loop).  At N > 1 every rank runs it on its own shard and policy replica, concurrently and
        # without any collective; rank 0 reports its own per-GPU figures.
        try:
            extra["full_ga3c_loop"] = full_loop(BatchedCollisionAvoidanceEnv, cfg_for(N), device, W, N, rank, world_size, sync_all,
                                                brief=not args.full_loop, steps=240 if args.full_loop else 120)
        except Exception as exc:      # noqa: BLE001  (an extra must never cost the contract line)
            extra["full_ga3c_loop"] = {"error": repr(exc)}

    if args.sweep and rank == 0:
        sweep = []
        for Ws in (1024, 8192, 65536, 262144, 1048576, 4194304):
            e2, a2 = make(Ws)
            run_steps(e2, a2, 64)
            f = kernel_figures(e2, a2, N, Ws, 128)
            k1 = e2.kernel_time_ms(a2, 64, 1)
            torch.cuda.synchronize(device)
            sweep.append({"worlds": Ws, "kernel_us_per_step": f["kernel_us_per_step"], "steps_per_launch": f["steps_per_launch"],
                          "agent_steps_per_s": Ws * N / (f["kernel_us_per_step"] * 1e-6), "GBps": f["achieved"], "frac": f["frac"],
                          "single_step_launch_us": k1 * 1e3,
                          "single_step_launch_frac": algorithmic_bytes_per_agent_step(N - 1) * Ws * N / (k1 * 1e-3) / 1e9 / HBM_PEAK_GBS})
            e2.close()
            del e2, a2
        extra["saturation_sweep"] = sweep

    which = {4: "configs[1]", 10: "configs[3]"}.get(N, "configs[1]-style")
    line = {
        "metric": "agent-steps/sec (env.step) at %d agents x %d worlds per GPU" % (N, W),
        "value": value, "unit": "agent-steps/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE %s: %d agents x %d worlds per GPU, unicycle dynamics, GEN v1 synthetic scenarios, "
                               "uniform random actions pre-staged on the device, in-kernel auto-reset; launches of up to %d steps "
                               "(world state in registers between the steps of a launch, every step's outputs written)"
                               % (which, N, W, args.slices),
                   "worlds_per_gpu": W, "agents_per_world": N, "obs_width": env.obs_width, "steps_per_launch": min(args.slices, args.steps),
                   "parallelism": "worlds sharded over %d GPU(s), no data-path collective" % world_size},
        "roofline": roofline,
    }
    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(N, W, args.cpu_seconds, int(env.cfg.gen_pool_size))
        line["cpu_baseline"]["host_cpus"] = os.cpu_count()
        extra["python_reference_style_baseline"] = python_baseline(N, min(3.0, args.cpu_seconds))
        try:
            extra["cpu_baseline_all_cores"] = cpu_baseline_all_cores(N, W, min(4.0, args.cpu_seconds), int(env.cfg.gen_pool_size))
        except Exception as exc:      # noqa: BLE001
            extra["cpu_baseline_all_cores"] = {"error": repr(exc)}
    if extra:
        line["extra"] = extra
    env.close()
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
