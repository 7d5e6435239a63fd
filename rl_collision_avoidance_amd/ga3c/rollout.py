"""``BatchedRollout`` -- every world's GA3C actor loop, on the device.

Replaces W x ``ProcessAgent`` (/root/reference/ga3c/GA3C/ProcessAgent.py): the per-step
predict -> sample -> env.step -> Experience bookkeeping of ``run_episode`` (:105-211), the n-step
return of ``_accumulate_rewards`` (:54-79), ``convert_to_nparray`` (:82-87) and the two queue
puts of ``run`` (:233-243).  Nothing crosses a process boundary: the policy is called once per
step on the whole ``[W*N, D]`` observation matrix (what 128-row ``ThreadPredictor`` batches
approximate, ThreadPredictor.py:40-75), the env steps in one kernel launch, and a second kernel
(csrc/cavoid_rollout.hpp) keeps a TIME-MAJOR experience store: step t's states, actions and
(later) n-step returns live in block ``t % ring_len``; a row becomes a training row (emit_t >= 0) the moment
the reference would have yielded it.  ``drain()`` compacts the final blocks into the shapes
``Server.train_model(x_, r_, a_)`` takes (Server.py:114-124; x [n, D], r [n], a one-hot f32
[n, num_actions])."""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional, Tuple

import torch

from .. import _lib
from ..batched_env import BatchedCollisionAvoidanceEnv

Policy = Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor]]   # x[B, D] -> (p[B, A], v[B])


class TrainingBatch(object):
    """Rows that became final since the last drain: ``x`` f32 [n, D], ``r`` f32 [n] (n-step returns),
    ``a_index`` int32 [n], ``src`` int32 [n, 4] (world, agent, recorded-at step, emitted-at step)."""

    def __init__(self, x, r, a_index, src, num_actions: int, dropped: int):
        self.x, self.r, self.a_index, self.src = x, r, a_index, src
        self.num_actions, self.dropped = num_actions, dropped

    def __len__(self) -> int:
        return int(self.r.shape[0])

    @property
    def a(self) -> torch.Tensor:
        """one-hot float32 [n, num_actions] -- ``np.eye(num_actions)[idx].astype(np.float32)`` (ProcessAgent.py:84)."""
        return torch.nn.functional.one_hot(self.a_index.long(), self.num_actions).to(torch.float32)


class BatchedRollout(object):
    def __init__(self, env: BatchedCollisionAvoidanceEnv, policy: Optional[Policy], time_max: Optional[int] = None,
                 discount: float = 0.97, ring_len: Optional[int] = None, dup_capacity: Optional[int] = None,
                 episode_capacity: Optional[int] = None, reflush_done: bool = True, greedy: bool = False,
                 generator: Optional[torch.Generator] = None, skip_finished: Optional[bool] = None, frozen_policy=None):
        self.env, self.policy = env, policy
        # the network behind the scripted "frozen network" agents (policy 4, SURVEY section 8f-N3: the GA3C-CADRL agent -- a
        # NON-learning agent driven by a frozen NetworkVP_rnn, /root/reference/ga3c/GA3C/Server.py:36): a FusedPolicy whose argmax
        # action replaces the learner's sample on exactly those rows (cavoid_policy_rows lists them on the device)
        self.frozen_policy = frozen_policy
        if policy is not None and frozen_policy is None and float(env.cfg.gen_frozen_fraction) > 0.0 and float(env.cfg.gen_nonlearning_fraction) > 0.0:
            raise ValueError("the env generates frozen-network agents: pass frozen_policy (a FusedPolicy)")
        cfg = env.config
        self.time_max = int(time_max if time_max is not None else getattr(cfg, "TIME_MAX", int(4 / cfg.DT)))
        self.discount = float(getattr(cfg, "DISCOUNT", discount))
        self.greedy = greedy                     # PLAY_MODE / EVALUATE_MODE: argmax instead of sampling (:98-103)
        self._reflush = bool(reflush_done)
        # step(): env.step and the Experience bookkeeping in one launch (cavoid_step_push) instead of three (env, push, episode log);
        # CAVOID_FUSE_ENV_PUSH=0 keeps the three launches (the form the goldens pin; both are compared bitwise in the tests)
        import os
        self.fuse_env_push = os.environ.get("CAVOID_FUSE_ENV_PUSH", "1") not in ("0", "")
        # fused policy only: no forward pass for absent agents and for agents that have finished and wait for their world to
        # end (the env ignores their action, nothing of theirs is recorded; ~30 % of the rows in the TrainPhase1 workload).
        # Never in the faithful re-flush mode, whose quirk rows carry V(s) of exactly those agents.  Default: only when the
        # batch is several rounds of 64-row tiles over the 256 CUs -- at 4 x 8192 (512 tiles = one round, two per CU) a
        # shorter tile list does not shorten the launch and the list itself costs 13 us.
        W_N = env.num_worlds * env.max_agents
        self.skip_finished = (not reflush_done and W_N > 65536) if skip_finished is None else bool(skip_finished)
        self.generator = generator
        W, N, D = env.num_worlds, env.max_agents, env.obs_width - 1
        self.slots = W * N
        # a block is final T_max + 2 steps after it was written; keep twice that between drains
        self.margin = self.time_max + 2
        self.ring_len = int(ring_len if ring_len is not None else 2 * self.margin + 8)
        if self.ring_len < self.margin + 2:
            raise ValueError("ring_len must exceed time_max + 3")
        self.dup_capacity = int(dup_capacity if dup_capacity is not None else 2 * self.slots + 1024)
        self.episode_capacity = int(episode_capacity if episode_capacity is not None else 4 * W + 1024)
        dev = env.device
        R, S = self.ring_len, self.slots
        self.x = torch.zeros((R, S, D), dtype=torch.float32, device=dev)
        self.val = torch.zeros((R, S), dtype=torch.float64, device=dev)
        self.ret = torch.zeros((R, S), dtype=torch.float32, device=dev)
        self.act_ring = torch.zeros((R, S), dtype=torch.uint8, device=dev)
        self.emit_t = torch.full((R, S), -1, dtype=torch.int32, device=dev)
        self.dup_x = torch.empty((self.dup_capacity, D), dtype=torch.float32, device=dev)
        self.dup_r = torch.empty((self.dup_capacity,), dtype=torch.float32, device=dev)
        self.dup_a = torch.empty((self.dup_capacity,), dtype=torch.int32, device=dev)
        self.dup_src = torch.empty((self.dup_capacity, 4), dtype=torch.int32, device=dev)
        self.dup_count = torch.zeros((2,), dtype=torch.int32, device=dev)
        self.ep_out = torch.empty((self.episode_capacity, 3), dtype=torch.float32, device=dev)
        self.ep_count = torch.zeros((2,), dtype=torch.int32, device=dev)
        self.row_index = torch.zeros((self.slots,), dtype=torch.int32, device=dev)
        self.row_count = torch.zeros((1,), dtype=torch.int32, device=dev)
        self._obs_buffers = [env.obs, torch.zeros_like(env.obs)]
        self._cur = 0
        self.step_index = 0                       # pushes done so far == the next step's index
        self.drained_until = 0                    # blocks with step < drained_until were handed out
        self.frames = 0                           # rows handed to the trainer (the reference's PPS numerator)
        self.lost_blocks = 0                      # blocks overwritten before a drain (drain more often / larger ring_len)
        self._lib = _lib.lib()
        h = C.c_void_p()
        _lib.check(self._lib.cavoid_rollout_create(W, N, env.obs_width, self.time_max, self.discount,
                                                   1 if reflush_done else 0, self.ring_len, dev.index, C.byref(h)),
                   "cavoid_rollout_create")
        self._h = h
        self._graph = None
        self._bufs = None
        self._actor_buffers()                     # (allocated here, never inside a hipGraph capture that calls step() first)

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            torch.cuda.synchronize(self.env.device)
            self._lib.cavoid_rollout_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def obs(self) -> torch.Tensor:
        """Observation the next ``step()`` will act on, [W, N, 1+D]."""
        return self._obs_buffers[self._cur]

    def reset(self) -> torch.Tensor:
        """Start every world's first episode (``env.reset()`` at ProcessAgent.py:107)."""
        self._cur = 0
        self.env.reset()
        self.env.game_over.fill_(1)                # "every world has just (re)started": see cavoid_rollout_active_rows
        _lib.check(self._lib.cavoid_rollout_reset(self._h, self.env._stream()), "cavoid_rollout_reset")
        self.emit_t.fill_(-1)
        self.dup_count.zero_()
        self.ep_count.zero_()
        self.step_index = 0
        self.drained_until = 0
        return self.obs

    def act(self, obs: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """predict + select_action for every agent row (ProcessAgent.py:89-103,128-144)."""
        W, N = self.env.num_worlds, self.env.max_agents
        if getattr(self.policy, "accepts_strided_obs", False):
            # fused kernel (ga3c/policy_kernel.py): reads the obs tensor in place and selects the action itself
            rows = None
            if self.skip_finished:
                p = BatchedCollisionAvoidanceEnv._ptr
                _lib.check(self._lib.cavoid_rollout_active_rows(self._h, p(obs), p(self.env.done), p(self.env.game_over),
                                                                p(self.row_index), p(self.row_count), self.env._stream()),
                           "cavoid_rollout_active_rows")
                rows = (self.row_index, self.row_count)
            x = obs.view(W * N, -1)[:, 1:]
            actions, _, v = self.policy.act(x, greedy=self.greedy, rows=rows)
            if self.frozen_policy is not None:             # policy-4 agents take the frozen network's argmax instead
                self.frozen_policy.act(x, greedy=True, rows=self.env.policy_rows(_lib.POLICY_FROZEN_NET), actions_out=actions)
            return actions.reshape(W, N), v.reshape(W, N)
        p, v = self.policy(obs[..., 1:].reshape(W * N, -1))
        if self.greedy:
            actions = p.argmax(dim=-1)
        else:
            actions = torch.multinomial(p, 1, generator=self.generator).squeeze(-1)
        actions = actions.to(torch.int32).contiguous()
        if self.frozen_policy is not None:
            self.frozen_policy.act(obs.view(W * N, -1)[:, 1:], greedy=True, rows=self.env.policy_rows(_lib.POLICY_FROZEN_NET),
                                   actions_out=actions)
        return actions.reshape(W, N), v.reshape(W, N).to(torch.float32)

    def step(self, actions: Optional[torch.Tensor] = None, values: Optional[torch.Tensor] = None):
        """One env step of every world + experience bookkeeping.  ``actions``/``values`` override the
        policy (scripted runs).  Returns ``(rewards, done, game_over)`` of the step."""
        env = self.env
        obs = self.obs
        if actions is None:
            actions, values = self.act(obs)
        actions = env._want(actions, (env.num_worlds, env.max_agents), torch.int32, "actions")
        values = env._want(values, (env.num_worlds, env.max_agents), torch.float32, "values")
        nxt = self._obs_buffers[1 - self._cur]
        p = BatchedCollisionAvoidanceEnv._ptr
        if self.fuse_env_push and env.cfg.dynamics != 2:
            # env.step + Experience bookkeeping as ONE launch (cavoid_step_push: the fused actor's env phase as a kernel of its own)
            _lib.check(self._lib.cavoid_step_push(self._h_env(), self._h, C.byref(self._actor_buffers()), p(obs), p(nxt), p(actions), p(values),
                                                  p(env.rewards), p(env.done), p(env.game_over), -1, env._stream()), "cavoid_step_push")
            self._cur = 1 - self._cur
            self.step_index += 1
            return env.rewards, env.done, env.game_over
        _, rew, done, game_over = env.step_autoreset(actions, obs_out=nxt)
        _lib.check(self._lib.cavoid_rollout_push(
            self._h, p(obs), p(actions), p(values), p(rew), p(done), p(game_over), -1,      # -1: device-side step counter
            p(self.x), p(self.val), p(self.ret), p(self.act_ring), p(self.emit_t),
            p(self.dup_x), p(self.dup_r), p(self.dup_a), p(self.dup_src), p(self.dup_count), self.dup_capacity,
            p(self.ep_out), p(self.ep_count), self.episode_capacity, env._stream()), "cavoid_rollout_push")
        self._cur = 1 - self._cur
        self.step_index += 1
        return rew, done, game_over

    # -- the fused actor: K closed-loop steps in ONE launch -------------------------------------------------------
    @property
    def fused_available(self) -> bool:
        """``run_fused`` applies: a ``FusedPolicy`` on the default (float16-split) inference form, no velocity actions; ORCA agents
        up to 12 agents per world (their line scratch must fit beside the env step's tile in the LDS the policy lends it); the
        network behind frozen-network agents, if any, must be a ``FusedPolicy`` too (``cavoid_actor_run_mix``)."""
        return self.fused_unavailable_reason is None

    @property
    def fused_unavailable_reason(self) -> Optional[str]:
        """Why ``step()`` / the hipGraph form must be used instead of the fused actor kernel (None: it applies)."""
        cfg = self.env.cfg
        # (skip_finished -- the step-by-step path's row list of the agents that still need an action -- does not matter here: the
        #  kernel packs the rows that still need an action to the front of their tile itself and skips the empty row tiles -- unless the
        #  re-flush quirk or frozen-network agents make every row count -- and hands the others action 0 / value 0 like the row-list pass)
        if not getattr(self.policy, "accepts_strided_obs", False):
            return "the policy is not a FusedPolicy"
        if cfg.rvo_enabled and cfg.max_agents > 12:
            return "ORCA agents with more than 12 agents per world (the line scratch does not fit the lent LDS)"
        if cfg.dynamics == 2:
            return "holonomic (velocity) actions"
        if self.frozen_policy is not None and not getattr(self.frozen_policy, "accepts_strided_obs", False):
            return "the frozen-network agents' policy is not a FusedPolicy"
        if self.frozen_policy is None and cfg.gen_frozen_fraction > 0.0 and cfg.gen_nonlearning_fraction > 0.0:
            # (cavoid_actor_run refuses exactly this with CAVOID_EUNSUPPORTED: those agents act by THEIR network)
            return "the env generates frozen-network agents but no frozen_policy was given"
        # the inference form is a property of the policy HANDLE, fixed at cavoid_policy_create (the environment switches are read there,
        # not here: changing them afterwards changes nothing)
        for who, pol in (("policy", self.policy), ("frozen policy", self.frozen_policy)):
            if pol is not None and getattr(pol, "inference_form", ("split", 16)) != ("split", 16):
                return "the %s runs a non-default inference form %r (CAVOID_POLICY_F32 / CAVOID_POLICY_PRODUCTS at its creation)" % (
                    who, pol.inference_form)
        return None

    @property
    def actor_path(self) -> str:
        """Which form of the actor loop this rollout runs with -- for train / bench logs (a run that silently falls from the fused
        kernel to one launch per phase loses ~40 % of its actor throughput)."""
        if self.fused_available:
            return "fused actor kernel (cavoid_actor_run%s)" % ("_mix: learner + frozen network" if self.frozen_policy is not None else "")
        why = self.fused_unavailable_reason
        return ("one launch per phase (policy, env + bookkeeping%s) -- fused kernel not applicable: %s"
                % (", frozen network" if self.frozen_policy is not None else "", why))

    def _h_env(self):
        return self.env._h

    def _actor_buffers(self):
        if getattr(self, "_bufs", None) is None:
            p = lambda t: t.data_ptr()
            b = _lib.CavoidRolloutBuffers()
            b.struct_size = C.sizeof(_lib.CavoidRolloutBuffers)
            b.x, b.val, b.ret, b.act, b.emit_t = p(self.x), p(self.val), p(self.ret), p(self.act_ring), p(self.emit_t)
            b.dup_x, b.dup_r, b.dup_a, b.dup_src, b.dup_count = p(self.dup_x), p(self.dup_r), p(self.dup_a), p(self.dup_src), p(self.dup_count)
            b.dup_capacity, b.ep_out, b.ep_count, b.ep_capacity = self.dup_capacity, p(self.ep_out), p(self.ep_count), self.episode_capacity
            W, N, dev = self.env.num_worlds, self.env.max_agents, self.env.device
            self._act_out = torch.zeros((W, N), dtype=torch.int32, device=dev)
            self._val_out = torch.zeros((W, N), dtype=torch.float32, device=dev)
            self._bufs = b
        return self._bufs

    def run_fused(self, n_steps: int) -> None:
        """``n_steps`` closed-loop steps of every world -- predict, select_action, env.step, Experience bookkeeping
        (ProcessAgent.py:116-211) -- in ONE launch (``cavoid_actor_run``): per tile of worlds a workgroup runs the policy
        on its own rows, steps its own worlds and records the step, with no kernel boundary and no host in between.
        Bit-identical to ``n_steps`` calls of ``step()``; capturable into a hipGraph."""
        if not self.fused_available:
            raise RuntimeError("run_fused needs a FusedPolicy and a configuration the fused kernel carries (see fused_available)")
        env, pol = self.env, self.policy
        b = self._actor_buffers()
        cur, nxt = self._obs_buffers[self._cur], self._obs_buffers[1 - self._cur]
        p = BatchedCollisionAvoidanceEnv._ptr
        if self.frozen_policy is not None:                  # learners + frozen-network agents (the GA3C-CADRL agent mechanism) in one launch
            _lib.check(self._lib.cavoid_actor_run_mix(env._h, pol._h, self.frozen_policy._h, self._h, C.byref(b), p(cur), p(nxt), p(env.rewards),
                                                      p(env.done), p(env.game_over), p(self._act_out), p(self._val_out), int(n_steps),
                                                      1 if self.greedy else 0, env._stream()), "cavoid_actor_run_mix")
        else:
            _lib.check(self._lib.cavoid_actor_run(env._h, pol._h, self._h, C.byref(b), p(cur), p(nxt), p(env.rewards), p(env.done),
                                                  p(env.game_over), p(self._act_out), p(self._val_out), int(n_steps), 1 if self.greedy else 0,
                                                  env._stream()), "cavoid_actor_run")
        self._cur = (self._cur + int(n_steps)) & 1
        self.step_index += int(n_steps)

    def capture_fused(self, steps_per_graph: int = 8) -> None:
        """``run_fused(steps_per_graph)`` as a one-node hipGraph (``replay`` then costs one graph launch per K steps)."""
        if steps_per_graph < 2 or steps_per_graph % 2:
            raise ValueError("steps_per_graph must be a positive even number")
        side = torch.cuda.Stream(device=self.env.device)
        side.wait_stream(torch.cuda.current_stream(self.env.device))
        with torch.cuda.stream(side):
            self.run_fused(2)
        torch.cuda.current_stream(self.env.device).wait_stream(side)
        torch.cuda.synchronize(self.env.device)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self.run_fused(steps_per_graph)
        self.step_index -= steps_per_graph                # capture records, it does not execute
        self._graph_steps = steps_per_graph

    # -- hipGraph path ---------------------------------------------------------------------------------
    def capture(self, steps_per_graph: int = 2) -> None:
        """Capture ``steps_per_graph`` closed-loop steps -- policy forward, action sampling, env step and
        experience bookkeeping -- into ONE hipGraph (HIP streams and graphs instead of a tracing
        compiler).  The per-step launch train (tens of small kernels) then costs one graph replay.
        Must be even: the two observation buffers alternate.  The experience store is addressed by the
        handle's device-side step counter, so a replay lands in the right blocks."""
        if steps_per_graph < 2 or steps_per_graph % 2:
            raise ValueError("steps_per_graph must be a positive even number")
        if self.policy is None:
            raise ValueError("capture() needs a policy")
        side = torch.cuda.Stream(device=self.env.device)
        side.wait_stream(torch.cuda.current_stream(self.env.device))
        with torch.cuda.stream(side):                      # warm-up outside capture (allocator, lazy inits)
            for _ in range(2):
                self.step()
        torch.cuda.current_stream(self.env.device).wait_stream(side)
        torch.cuda.synchronize(self.env.device)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            for _ in range(steps_per_graph):
                self.step()
        self.step_index -= steps_per_graph                # capture records, it does not execute
        self._graph_steps = steps_per_graph

    def replay(self, n_replays: int = 1) -> None:
        """Run ``n_replays * steps_per_graph`` env steps of every world."""
        for _ in range(n_replays):
            self._graph.replay()
        self.step_index += n_replays * self._graph_steps

    # -- hand-over to the trainer ------------------------------------------------------------------------
    def pending_final_steps(self) -> Tuple[int, int]:
        """[lo, hi): steps whose blocks are final (no slot can still hold a pending return in them)."""
        hi = max(self.step_index - self.margin, self.drained_until)
        lo = self.drained_until
        oldest_alive = self.step_index - self.ring_len       # anything older has been overwritten
        if lo < oldest_alive:
            self.lost_blocks += oldest_alive - lo
            lo = oldest_alive
        return lo, hi

    def drain(self, flush_all: bool = False, provenance: bool = True) -> TrainingBatch:
        """Hand the final rows to the trainer (``training_q.put((x_, r_, a_))``, :238).  ``flush_all``
        also takes the not-yet-final blocks' emitted rows (end of a run / tests).  One compaction launch
        (``cavoid_rollout_compact``) and one read-back of the row count; ``provenance=False`` skips the
        per-row (world, agent, recorded-at, emitted-at) record that only tests look at."""
        lo, hi = self.pending_final_steps()
        if flush_all:
            hi = self.step_index
        S, D, dev = self.slots, self.env.obs_width - 1, self.env.device
        cap = max(hi - lo, 0) * S
        out_x = torch.empty((cap, D), dtype=torch.float32, device=dev)
        out_r = torch.empty((cap,), dtype=torch.float32, device=dev)
        out_a = torch.empty((cap,), dtype=torch.int32, device=dev)
        out_src = torch.empty((cap, 4), dtype=torch.int32, device=dev) if provenance else None
        counts = torch.empty((2,), dtype=torch.int32, device=dev)
        p = BatchedCollisionAvoidanceEnv._ptr
        n = 0
        for s0 in range(lo, max(hi, lo), 65535):                            # (one launch unless a test flushes > 65535 steps)
            s1 = min(hi, s0 + 65535)
            _lib.check(self._lib.cavoid_rollout_compact(
                self._h, s0, s1, 1 if flush_all else 0, p(self.x), p(self.ret), p(self.act_ring), p(self.emit_t),
                C.c_void_p(out_x[n:].data_ptr()), C.c_void_p(out_r[n:].data_ptr()), C.c_void_p(out_a[n:].data_ptr()),
                C.c_void_p(out_src[n:].data_ptr()) if provenance else None, p(counts), cap - n, self.env._stream()),
                "cavoid_rollout_compact")
            if s1 < hi:
                n += int(counts[0].item())
        got, _, n_dup, dropped = torch.cat([counts, self.dup_count]).tolist() if hi > lo else [0, 0] + self.dup_count.tolist()
        n += int(got)                                                       # (the one read-back of a normal drain)
        if not flush_all:
            self.drained_until = max(hi, lo)
        xs, rs, as_, srcs = [out_x[:n]], [out_r[:n]], [out_a[:n]], [out_src[:n] if provenance else None]
        n_dup = min(n_dup, self.dup_capacity)
        if n_dup:
            xs.append(self.dup_x[:n_dup].clone()); rs.append(self.dup_r[:n_dup].clone())
            as_.append(self.dup_a[:n_dup].clone()); srcs.append(self.dup_src[:n_dup].clone() if provenance else None)
            self.dup_count.zero_()
        cat = lambda parts: parts[0] if len(parts) == 1 else torch.cat(parts)
        batch = TrainingBatch(cat(xs), cat(rs), cat(as_), cat(srcs) if provenance else None, self.env.num_actions, dropped + 0)
        self.frames += len(batch)
        return batch

    # -- the same hand-over, split in two so that the host never waits for the launch it has just enqueued --------------------
    def drain_begin(self, provenance: bool = False) -> dict:
        """First half of ``drain()``: enqueue the compaction of the final blocks and an asynchronous read-back of the row
        count; returns a handle for ``drain_end``.  Between the two calls the caller enqueues the NEXT replay, so the GPU goes
        from one launch to the next while the host collects the previous batch (cleaned mode only: the re-flush quirk's append
        buffer is read synchronously by ``drain()``)."""
        if self._reflush:
            raise RuntimeError("drain_begin / drain_end serve reflush_done=False; use drain()")
        lo, hi = self.pending_final_steps()
        if hi - lo > 65535:
            raise RuntimeError("too many undrained steps for one compaction launch: call drain()")
        S, D, dev = self.slots, self.env.obs_width - 1, self.env.device
        cap = max(hi - lo, 0) * S
        h = {"x": torch.empty((cap, D), dtype=torch.float32, device=dev), "r": torch.empty((cap,), dtype=torch.float32, device=dev),
             "a": torch.empty((cap,), dtype=torch.int32, device=dev),
             "src": torch.empty((cap, 4), dtype=torch.int32, device=dev) if provenance else None,
             "counts": torch.zeros((2,), dtype=torch.int32, device=dev), "host": torch.zeros((2,), dtype=torch.int32).pin_memory(),
             "event": torch.cuda.Event()}
        p = BatchedCollisionAvoidanceEnv._ptr
        if hi > lo:
            _lib.check(self._lib.cavoid_rollout_compact(self._h, lo, hi, 0, p(self.x), p(self.ret), p(self.act_ring), p(self.emit_t),
                                                        p(h["x"]), p(h["r"]), p(h["a"]), p(h["src"]) if provenance else None, p(h["counts"]), cap,
                                                        self.env._stream()), "cavoid_rollout_compact")
        h["host"].copy_(h["counts"], non_blocking=True)
        h["event"].record(torch.cuda.current_stream(dev))
        self.drained_until = max(hi, lo)
        return h

    def drain_end(self, h: dict) -> TrainingBatch:
        h["event"].synchronize()                              # (waits for THIS hand-over's count only, not for what was enqueued since)
        n = int(h["host"][0])                                 # (cleaned mode: no append buffer, nothing can be dropped)
        batch = TrainingBatch(h["x"][:n], h["r"][:n], h["a"][:n], h["src"][:n] if h["src"] is not None else None,
                              self.env.num_actions, 0)
        self.frames += len(batch)
        return batch

    def drain_episodes(self) -> torch.Tensor:
        """Finished-episode records [k, 3] = (world, total_reward, total_length): the payload of
        ``episode_log_q.put((now, total_reward, total_length))`` (:243)."""
        n = min(int(self.ep_count[0].item()), self.episode_capacity)
        out = self.ep_out[:n].clone()
        self.ep_count.zero_()
        return out
