"""``BatchedCollisionAvoidanceEnv`` -- W independent collision-avoidance worlds stepped by one
HIP kernel launch on an MI355X.

Host-side mirror of the reference's env seam: one world of this class behaves like the object
``create_env()`` hands to ``Environment`` (/root/reference/ga3c/GA3C/Environment.py:54-56):
``reset() -> obs`` (:106) and ``step(actions) -> (obs, rewards, game_over, which_agents_done)``
(:112; ProcessAgent.py:149-157), with the observation layout of Config.py:40,72-76.  Batched:
every return value gains a leading world dimension and lives on the GPU as a torch tensor.

PyTorch is plumbing only (device memory + the current HIP stream); the arithmetic is in
``csrc/cavoid_kernels.hpp`` behind the C ABI of ``include/cavoid.h``.  No CPU fallback exists.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from .config import EnvConfig

_SORT = {"closest_last": 0, "closest_first": 1, "time_to_impact": 2}
_DYN = {"unicycle": 0, "unicycle_max_turn_rate": 1, "holonomic": 2}


def make_cfg(config: Optional[EnvConfig] = None, **overrides) -> _lib.CavoidCfg:
    """Build the C-ABI config from an ``EnvConfig`` (attribute names of the reference's env Config);
    ``overrides`` set raw ``cavoid_cfg`` fields (e.g. ``gen_min_agents=2``)."""
    config = config or EnvConfig()
    cfg = _lib.CavoidCfg()
    N = int(config.MAX_NUM_AGENTS_IN_ENVIRONMENT)
    M = int(config.MAX_NUM_OTHER_AGENTS_OBSERVED)
    _lib.check(_lib.lib().cavoid_default_cfg(C.byref(cfg), N, M), "cavoid_default_cfg")
    cfg.dt = config.DT
    cfg.near_goal_threshold = config.NEAR_GOAL_THRESHOLD
    cfg.max_time_ratio = config.MAX_TIME_RATIO
    cfg.collision_dist = config.COLLISION_DIST
    cfg.getting_close_range = config.GETTING_CLOSE_RANGE
    cfg.sensing_horizon = float(config.SENSING_HORIZON)
    cfg.sort_method = _SORT[config.AGENT_SORTING_METHOD]
    cfg.reward_at_goal = config.REWARD_AT_GOAL
    cfg.reward_collision = config.REWARD_COLLISION_WITH_AGENT
    cfg.reward_getting_close = config.REWARD_GETTING_CLOSE
    cfg.reward_time_step = config.REWARD_TIME_STEP
    cfg.evaluate_mode = 1 if getattr(config, "EVALUATE_MODE", False) else 0
    possible = [config.REWARD_AT_GOAL, config.REWARD_COLLISION_WITH_AGENT, config.REWARD_TIME_STEP,
                config.REWARD_COLLISION_WITH_WALL, config.REWARD_WIGGLY_BEHAVIOR]
    cfg.reward_clip_lo, cfg.reward_clip_hi = min(possible), max(possible)
    # scenario generator / scripted agents
    cfg.gen_mode = {"ring": 0, "box": 1}[getattr(config, "TEST_CASE_GENERATOR", "ring")]
    cfg.gen_nonlearning_fraction = float(getattr(config, "SCRIPTED_AGENT_FRACTION", 0.0))
    cfg.gen_static_fraction = float(getattr(config, "SCRIPTED_STATIC_FRACTION", 0.5))
    cfg.gen_rvo_fraction = float(getattr(config, "SCRIPTED_RVO_FRACTION", 0.0))
    cfg.gen_frozen_fraction = float(getattr(config, "SCRIPTED_FROZEN_NET_FRACTION", 0.0))
    cfg.rvo_enabled = 1 if cfg.gen_rvo_fraction > 0.0 and cfg.gen_nonlearning_fraction > 0.0 else 0
    cfg.rvo_time_horizon = float(getattr(config, "RVO_TIME_HORIZON", 5.0))
    cfg.rvo_collab_coeff = float(getattr(config, "RVO_COLLAB_COEFF", 0.5))
    for key, val in overrides.items():
        if key == "actions":
            table = np.asarray(val, dtype=np.float64)
            if table.ndim != 2 or table.shape[1] != 2 or len(table) > _lib.MAX_ACTIONS:
                raise ValueError("actions must be [<=%d, 2]" % _lib.MAX_ACTIONS)
            cfg.num_actions = len(table)
            for r, (a0, a1) in enumerate(table):
                cfg.actions[r][0], cfg.actions[r][1] = a0, a1
        elif key == "dynamics" and isinstance(val, str):
            cfg.dynamics = _DYN[val]
        elif key == "sort_method" and isinstance(val, str):
            cfg.sort_method = _SORT[val]
        elif key in ("gen_box_small", "gen_box_large"):
            getattr(cfg, key)[0], getattr(cfg, key)[1] = float(val[0]), float(val[1])
        elif hasattr(cfg, key):
            setattr(cfg, key, val)
        else:
            raise AttributeError("cavoid_cfg has no field %r" % key)
    return cfg


class StepSlots(object):
    """Per-step outputs of a K-step launch: slot t holds what ``env.step`` returned at step t --
    ``obs [K,W,N,1+D]``, ``rewards [K,W,N]``, ``done [K,W,N]``, ``game_over [K,W]`` (what
    ``ProcessAgent.run_episode`` reads after every step, /root/reference/ga3c/GA3C/ProcessAgent.py:149-157), or with
    ``packed=True`` one ``packed [K,W,N,1+D+2]`` record tensor + ``game_over``."""

    def __init__(self, env: "BatchedCollisionAvoidanceEnv", steps: int, packed: bool = False):
        K, W, N, dev = int(steps), env.num_worlds, env.max_agents, env.device
        self.steps, self.is_packed = K, packed
        self.game_over = torch.zeros((K, W), dtype=torch.uint8, device=dev)
        if packed:
            self.packed = torch.zeros((K, W, N, env.packed_width), dtype=torch.float32, device=dev)
            self.obs = self.packed[..., :env.obs_width]
            self.rewards = self.packed[..., env.obs_width]
            self.done = self.packed[..., env.obs_width + 1]
        else:
            self.obs = torch.zeros((K, W, N, env.obs_width), dtype=torch.float32, device=dev)
            self.rewards = torch.zeros((K, W, N), dtype=torch.float32, device=dev)
            self.done = torch.zeros((K, W, N), dtype=torch.uint8, device=dev)


class BatchedCollisionAvoidanceEnv(object):
    """``num_worlds`` worlds of up to ``N`` agents on one GPU.

    world_offset: global id of this shard's first world.  Scenario RNG streams are keyed on
    (seed, global world id, episode), so a sharded run reproduces the unsharded one bit for bit.
    """

    def __init__(self, num_worlds: int, config: Optional[EnvConfig] = None, device="cuda:0",
                 world_offset: int = 0, seed: int = 0, **cfg_overrides):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BatchedCollisionAvoidanceEnv runs on an MI355X only (device=%r); "
                               "there is no CPU fallback" % (device,))
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the env.step hot path has no CPU fallback")
        self.config = config or EnvConfig()
        self.cfg = make_cfg(self.config, **cfg_overrides)
        self.num_worlds = int(num_worlds)
        self.max_agents = int(self.cfg.max_agents)
        self.max_other = int(self.cfg.max_other)
        self.obs_width = 6 + 7 * self.max_other
        self.num_actions = int(self.cfg.num_actions)
        self.world_offset = int(world_offset)
        self._lib = _lib.lib()
        handle = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", dev_index)
        _lib.check(self._lib.cavoid_create(C.byref(self.cfg), self.num_worlds, self.world_offset, dev_index,
                                           C.byref(handle)), "cavoid_create")
        self._h = handle
        W, N = self.num_worlds, self.max_agents
        self.obs = torch.zeros((W, N, self.obs_width), dtype=torch.float32, device=self.device)
        self.rewards = torch.zeros((W, N), dtype=torch.float32, device=self.device)
        self.done = torch.zeros((W, N), dtype=torch.uint8, device=self.device)
        self.game_over = torch.zeros((W,), dtype=torch.uint8, device=self.device)
        self.packed_width = self.obs_width + 2
        # the hot calls pass these as they are: the output tensors are allocated once and never replaced
        self._dev_index = dev_index
        self._p_obs, self._p_rew = C.c_void_p(self.obs.data_ptr()), C.c_void_p(self.rewards.data_ptr())
        self._p_done, self._p_go = C.c_void_p(self.done.data_ptr()), C.c_void_p(self.game_over.data_ptr())
        self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # (an int, without a Stream object around it)
        self.seed(seed)

    # -- packed outputs: one (obs | reward | done) record per agent, written by the kernel itself ---------------
    def new_packed(self) -> torch.Tensor:
        """A [W, N, obs_width + 2] float32 buffer for the ``*_packed`` calls (the multi-GPU gather record)."""
        return torch.zeros((self.num_worlds, self.max_agents, self.packed_width), dtype=torch.float32, device=self.device)

    def _packed(self, packed: torch.Tensor) -> torch.Tensor:
        want = (self.num_worlds, self.max_agents, self.packed_width)
        if packed.device != self.device or packed.dtype != torch.float32 or tuple(packed.shape) != want or not packed.is_contiguous():
            raise ValueError("packed must be a contiguous float32 %s tensor on %s" % (want, self.device))
        return packed

    def reset_packed(self, packed: torch.Tensor, world_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        mask = None if world_mask is None else self._want(world_mask, (self.num_worlds,), torch.uint8, "world_mask")
        _lib.check(self._lib.cavoid_reset_packed(self._h, self._ptr(mask), self._ptr(self._packed(packed)), self._stream()),
                   "cavoid_reset_packed")
        return packed

    def observe_packed(self, packed: torch.Tensor) -> torch.Tensor:
        _lib.check(self._lib.cavoid_observe_packed(self._h, self._ptr(self._packed(packed)), self._stream()), "cavoid_observe_packed")
        return packed

    def step_packed(self, actions: torch.Tensor, packed: torch.Tensor):
        """``step`` with ONE output tensor: packed[..., :width] = obs, [..., width] = reward, [..., width+1] = done."""
        a = self._actions(actions)
        _lib.check(self._lib.cavoid_step_packed(self._h, self._ptr(a), self._ptr(self._packed(packed)), self._ptr(self.game_over),
                                                self._stream()), "cavoid_step_packed")
        return packed, self.game_over

    def step_autoreset_packed(self, actions: torch.Tensor, packed, n_steps: Optional[int] = None):
        """``step_autoreset`` (actions [W,N]) or ``step_autoreset_n`` (actions [T,W,N]) into a packed record buffer
        ([W,N,1+D+2]: it holds the last step's records afterwards) or into ``StepSlots(packed=True)`` (every step's)."""
        if actions.dim() == 2:
            a, n, stride = self._actions(actions), 1, 0
        else:
            T = actions.shape[0]
            n = T if n_steps is None else int(n_steps)
            if n > T:
                raise ValueError("n_steps > number of action slices")
            a = self._want(actions, (T, self.num_worlds, self.max_agents), torch.int32, "actions")
            stride = self.num_worlds * self.max_agents
        if isinstance(packed, StepSlots):
            if not packed.is_packed or packed.steps < n:
                raise ValueError("need StepSlots(packed=True) of at least n_steps slots")
            _lib.check(self._lib.cavoid_step_autoreset_packed(self._h, self._ptr(a), stride, n, self.num_worlds, self._ptr(packed.packed),
                                                              self._ptr(packed.game_over), self._stream()), "cavoid_step_autoreset_packed")
            return packed.packed, packed.game_over
        _lib.check(self._lib.cavoid_step_autoreset_packed(self._h, self._ptr(a), stride, n, 0, self._ptr(self._packed(packed)),
                                                          self._ptr(self.game_over), self._stream()), "cavoid_step_autoreset_packed")
        return packed, self.game_over

    def new_step_slots(self, steps: int, packed: bool = False) -> StepSlots:
        """Output slots for ``steps`` steps of one multi-step launch (``step_autoreset_n(..., slots=...)``)."""
        return StepSlots(self, steps, packed)

    # -- lifetime ------------------------------------------------------------------------------
    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            torch.cuda.synchronize(self.device)
            self._lib.cavoid_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        if self._raw_stream is not None:
            return C.c_void_p(self._raw_stream(self._dev_index))
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _ptr(t: Optional[torch.Tensor]):
        return None if t is None else C.c_void_p(t.data_ptr())

    def _want(self, t: torch.Tensor, shape, dtype, name: str) -> torch.Tensor:
        if t.dtype is dtype and t.shape == shape and t.device == self.device and t.is_contiguous():
            return t                                       # (the usual case, checked first: this sits on the launch path)
        if t.device != self.device:
            raise ValueError("%s must live on %s (got %s)" % (name, self.device, t.device))
        if t.dtype != dtype:
            t = t.to(dtype)
        if tuple(t.shape) != tuple(shape):
            raise ValueError("%s must have shape %s (got %s)" % (name, tuple(shape), tuple(t.shape)))
        return t.contiguous()

    # -- RNG / episodes ----------------------------------------------------------------------------
    def seed(self, seed: int, episode: Optional[torch.Tensor] = None) -> None:
        """Seed the scenario generator; ``episode`` (uint32-valued int32/int64 tensor [W]) sets the
        per-world index of the *current* episode (default: before the first)."""
        self._seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        ep = None
        if episode is not None:
            ep = self._want(episode, (self.num_worlds,), torch.int32, "episode")
        _lib.check(self._lib.cavoid_seed(self._h, self._seed, self._ptr(ep), self._stream()), "cavoid_seed")

    def refresh_pool(self, epoch: int) -> None:
        """Re-fill the scenario pool with the generator's worlds of episode index ``epoch`` (fresh scenarios for a long
        run without the generator on the step's critical path)."""
        _lib.check(self._lib.cavoid_pool_refresh(self._h, int(epoch) & 0xFFFFFFFF, self._stream()), "cavoid_pool_refresh")

    @property
    def episode(self) -> torch.Tensor:
        out = torch.empty((self.num_worlds,), dtype=torch.int32, device=self.device)
        _lib.check(self._lib.cavoid_get_episode(self._h, self._ptr(out), self._stream()), "cavoid_get_episode")
        return out

    # -- state -------------------------------------------------------------------------------------
    def set_state(self, state_f64: torch.Tensor, state_f32: torch.Tensor, flags: torch.Tensor) -> None:
        """Inject explicit world states (SoA): f64 [4,W*N] px,py,heading,t_remaining; f32 [5,W*N]
        gx,gy,radius,pref_speed,speed; flags int32 [W*N] (CAVOID_F_* bits)."""
        A = self.num_worlds * self.max_agents
        f64 = self._want(state_f64, (4, A), torch.float64, "state_f64")
        f32 = self._want(state_f32, (5, A), torch.float32, "state_f32")
        fl = self._want(flags, (A,), torch.int32, "flags")
        _lib.check(self._lib.cavoid_set_state(self._h, self._ptr(f64), self._ptr(f32), self._ptr(fl), self._stream()),
                   "cavoid_set_state")

    def get_state(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        A = self.num_worlds * self.max_agents
        f64 = torch.empty((4, A), dtype=torch.float64, device=self.device)
        f32 = torch.empty((5, A), dtype=torch.float32, device=self.device)
        fl = torch.empty((A,), dtype=torch.int32, device=self.device)
        _lib.check(self._lib.cavoid_get_state(self._h, self._ptr(f64), self._ptr(f32), self._ptr(fl), self._stream()),
                   "cavoid_get_state")
        return f64, f32, fl

    def state_dict(self) -> dict:
        """Everything needed to resume this env bit-for-bit (the reference never checkpoints env state,
        SURVEY.md section 5; here it is three tensors + the episode counters + the seed)."""
        f64, f32, flags = self.get_state()
        return {"state_f64": f64, "state_f32": f32, "flags": flags, "episode": self.episode, "seed": self._seed,
                "num_worlds": self.num_worlds, "max_agents": self.max_agents, "world_offset": self.world_offset}

    def load_state_dict(self, sd: dict) -> None:
        if (sd["num_worlds"], sd["max_agents"]) != (self.num_worlds, self.max_agents):
            raise ValueError("checkpoint is for %dx%d worlds x agents, env is %dx%d"
                             % (sd["num_worlds"], sd["max_agents"], self.num_worlds, self.max_agents))
        self.seed(sd["seed"], sd["episode"].to(self.device))
        self.set_state(sd["state_f64"].to(self.device), sd["state_f32"].to(self.device), sd["flags"].to(self.device))

    # -- the gym-style surface ---------------------------------------------------------------------
    def reset(self, world_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Start the next episode in the masked worlds (all if None); returns obs [W,N,1+D]."""
        mask = None
        if world_mask is not None:
            mask = self._want(world_mask, (self.num_worlds,), torch.uint8, "world_mask")
        _lib.check(self._lib.cavoid_reset(self._h, self._ptr(mask), self._ptr(self.obs), self._stream()), "cavoid_reset")
        return self.obs

    def observe(self) -> torch.Tensor:
        _lib.check(self._lib.cavoid_observe(self._h, self._ptr(self.obs), self._stream()), "cavoid_observe")
        return self.obs

    def _actions(self, actions: torch.Tensor) -> torch.Tensor:
        return self._want(actions, (self.num_worlds, self.max_agents), torch.int32, "actions")

    def step(self, actions: torch.Tensor):
        """actions int [W,N] (indices into the action table; ignored for done / scripted agents)
        -> (obs f32 [W,N,1+D], rewards f32 [W,N], done u8 [W,N], game_over u8 [W]).
        The returned tensors are the env's own output buffers, overwritten by the next call."""
        a = self._actions(actions)
        _lib.check(self._lib.cavoid_step(self._h, self._ptr(a), self._ptr(self.obs), self._ptr(self.rewards),
                                         self._ptr(self.done), self._ptr(self.game_over), self._stream()), "cavoid_step")
        return self.obs, self.rewards, self.done, self.game_over

    def step_continuous(self, actions: torch.Tensor):
        """actions f32 [W,N,2]: (speed, delta_heading) for unicycle dynamics, (vx, vy) for holonomic."""
        a = self._want(actions, (self.num_worlds, self.max_agents, 2), torch.float32, "actions")
        _lib.check(self._lib.cavoid_step_continuous(self._h, self._ptr(a), self._ptr(self.obs), self._ptr(self.rewards),
                                                    self._ptr(self.done), self._ptr(self.game_over), self._stream()),
                   "cavoid_step_continuous")
        return self.obs, self.rewards, self.done, self.game_over

    def step_autoreset(self, actions: torch.Tensor, obs_out: Optional[torch.Tensor] = None):
        """``step`` + in-kernel restart of finished worlds: their obs rows hold the first observation
        of the next episode; rewards / done / game_over still describe the finished step.
        ``obs_out``: write the observation there instead of into ``self.obs`` (lets a rollout keep
        the previous observation alive without a copy)."""
        a = self._actions(actions)
        obs = self.obs if obs_out is None else self._want(obs_out, self.obs.shape, torch.float32, "obs_out")
        if obs_out is not None and obs.data_ptr() != obs_out.data_ptr():
            raise ValueError("obs_out must be a contiguous float32 tensor on the env's device")
        rc = self._lib.cavoid_step_autoreset(self._h, C.c_void_p(a.data_ptr()), self._p_obs if obs_out is None else C.c_void_p(obs.data_ptr()),
                                             self._p_rew, self._p_done, self._p_go, self._stream())
        if rc != 0:
            _lib.check(rc, "cavoid_step_autoreset")
        return obs, self.rewards, self.done, self.game_over

    def step_autoreset_n(self, actions: torch.Tensor, n_steps: Optional[int] = None, slots: Optional[StepSlots] = None):
        """Pre-staged actions int32 [T,W,N]; ``n_steps`` (default T, at most T) auto-reset steps in ONE launch -- the
        world state stays in registers between the steps, step t reads ``actions[t]``.  With ``slots`` (``new_step_slots``)
        step t's observations, rewards, done flags and game_over land in slot t; without, every step overwrites the env's
        own output buffers (they hold the last step's values afterwards)."""
        T = actions.shape[0]
        n = T if n_steps is None else int(n_steps)
        if n > T:
            raise ValueError("n_steps > number of action slices")
        a = self._want(actions, (T, self.num_worlds, self.max_agents), torch.int32, "actions")
        stride = self.num_worlds * self.max_agents
        if slots is not None:
            if slots.is_packed or slots.steps < n:
                raise ValueError("need plain StepSlots of at least n_steps slots")
            rc = self._lib.cavoid_step_autoreset_n(self._h, C.c_void_p(a.data_ptr()), stride, n, self.num_worlds,
                                                   C.c_void_p(slots.obs.data_ptr()), C.c_void_p(slots.rewards.data_ptr()),
                                                   C.c_void_p(slots.done.data_ptr()), C.c_void_p(slots.game_over.data_ptr()), self._stream())
            if rc != 0:
                _lib.check(rc, "cavoid_step_autoreset_n")
            return slots.obs, slots.rewards, slots.done, slots.game_over
        rc = self._lib.cavoid_step_autoreset_n(self._h, C.c_void_p(a.data_ptr()), stride, n, 0, self._p_obs, self._p_rew, self._p_done,
                                               self._p_go, self._stream())
        if rc != 0:
            _lib.check(rc, "cavoid_step_autoreset_n")
        return self.obs, self.rewards, self.done, self.game_over

    def prepared_autoreset_n(self, actions: torch.Tensor, n_steps: Optional[int] = None, slots: Optional[StepSlots] = None):
        """``step_autoreset_n`` with its arguments checked and converted ONCE: returns a callable that launches the same
        ``n_steps`` steps from the same action slices into the same slots every time it is called (a rollout loop that
        re-launches over fixed buffers; the launch-bound small-batch regime, where the per-call checks and pointer
        conversions are a visible part of a 30 us launch).  The callable keeps ``actions`` and ``slots`` alive, launches on the
        stream that is current when it is CALLED, and returns what ``step_autoreset_n`` returns."""
        T = actions.shape[0]
        n = T if n_steps is None else int(n_steps)
        if n > T:
            raise ValueError("n_steps > number of action slices")
        a = self._want(actions, (T, self.num_worlds, self.max_agents), torch.int32, "actions")
        stride = self.num_worlds * self.max_agents
        if slots is not None:
            if slots.is_packed or slots.steps < n:
                raise ValueError("need plain StepSlots of at least n_steps slots")
            out = (slots.obs, slots.rewards, slots.done, slots.game_over)
            args = (self._h, C.c_void_p(a.data_ptr()), stride, n, self.num_worlds, C.c_void_p(slots.obs.data_ptr()),
                    C.c_void_p(slots.rewards.data_ptr()), C.c_void_p(slots.done.data_ptr()), C.c_void_p(slots.game_over.data_ptr()))
        else:
            out = (self.obs, self.rewards, self.done, self.game_over)
            args = (self._h, C.c_void_p(a.data_ptr()), stride, n, 0, self._p_obs, self._p_rew, self._p_done, self._p_go)
        fn, stream, keep = self._lib.cavoid_step_autoreset_n, self._stream, (a, slots)

        def launch():
            rc = fn(*args, stream())
            if rc != 0:
                _lib.check(rc, "cavoid_step_autoreset_n")
            return out
        launch.keeps = keep
        return launch

    def step_continuous_autoreset(self, actions: torch.Tensor, n_steps: Optional[int] = None, slots: Optional[StepSlots] = None):
        """The auto-reset step with CONTINUOUS actions -- float32 ``[W,N,2]`` (one step) or pre-staged ``[T,W,N,2]`` (``n_steps`` <= T steps
        in ONE launch, the world state in registers between them): (speed, heading change) for the unicycle dynamics, a velocity for the
        holonomic ones.  ``slots``: plain ``StepSlots`` (every step's obs / rewards / done / game_over in its own slot) or
        ``StepSlots(packed=True)`` (the (obs | reward | done) records); without, the env's own output buffers hold the last step's."""
        W, N = self.num_worlds, self.max_agents
        if actions.dim() == 3:
            a, n, stride = self._want(actions, (W, N, 2), torch.float32, "actions"), 1, 0
        else:
            T = actions.shape[0]
            n = T if n_steps is None else int(n_steps)
            if n > T:
                raise ValueError("n_steps > number of action slices")
            a, stride = self._want(actions, (T, W, N, 2), torch.float32, "actions"), 2 * W * N
        if slots is not None and slots.steps < n:
            raise ValueError("need StepSlots of at least n_steps slots")
        if slots is not None and slots.is_packed:
            _lib.check(self._lib.cavoid_step_continuous_autoreset_packed(self._h, self._ptr(a), stride, n, W, self._ptr(slots.packed),
                                                                         self._ptr(slots.game_over), self._stream()),
                       "cavoid_step_continuous_autoreset_packed")
            return slots.packed, slots.game_over
        out = (slots.obs, slots.rewards, slots.done, slots.game_over) if slots is not None else (self.obs, self.rewards, self.done, self.game_over)
        if slots is None and actions.dim() == 3:
            _lib.check(self._lib.cavoid_step_continuous_autoreset(self._h, self._ptr(a), self._ptr(out[0]), self._ptr(out[1]), self._ptr(out[2]),
                                                                  self._ptr(out[3]), self._stream()), "cavoid_step_continuous_autoreset")
            return out
        _lib.check(self._lib.cavoid_step_continuous_autoreset_n(self._h, self._ptr(a), stride, n, W if slots is not None else 0,
                                                                self._ptr(out[0]), self._ptr(out[1]), self._ptr(out[2]), self._ptr(out[3]),
                                                                self._stream()), "cavoid_step_continuous_autoreset_n")
        return out

    def policy_rows(self, policy_id: int, only_running: bool = True):
        """(row_index int32 [W*N], row_count int32 [1]) on the device: the (world, agent) slots run by scripted policy
        ``policy_id`` (``_lib.POLICY_*``) -- for POLICY_FROZEN_NET the rows a frozen network must supply actions for."""
        if not hasattr(self, "_prow_index"):
            self._prow_index = torch.zeros((self.num_worlds * self.max_agents,), dtype=torch.int32, device=self.device)
            self._prow_count = torch.zeros((1,), dtype=torch.int32, device=self.device)
        _lib.check(self._lib.cavoid_policy_rows(self._h, int(policy_id), 1 if only_running else 0, self._ptr(self._prow_index),
                                                self._ptr(self._prow_count), self._stream()), "cavoid_policy_rows")
        return self._prow_index, self._prow_count

    # -- measurement -------------------------------------------------------------------------------
    def kernel_time_ms(self, actions: torch.Tensor, n_steps: int, steps_per_launch: int = 1, slots: Optional[StepSlots] = None) -> float:
        """Run ``n_steps`` autoreset steps (cycling through actions [T,W,N]) in launches of ``steps_per_launch`` steps,
        a HIP event pair per launch; returns the mean duration of ONE LAUNCH in ms (launch gaps excluded).  ``slots``:
        every launch writes its steps into these per-step output slots (else all into the env's own buffers)."""
        T = actions.shape[0]
        a = self._want(actions, (T, self.num_worlds, self.max_agents), torch.int32, "actions")
        stride = self.num_worlds * self.max_agents
        spl = max(1, min(int(steps_per_launch), T))
        total, launches, left = 0.0, 0, int(n_steps)
        while left > 0:
            n = min((T // spl) * spl, left)
            ms = C.c_float(0.0)
            o = slots if slots is not None else self
            if slots is not None and (slots.is_packed or slots.steps < spl):
                raise ValueError("need plain StepSlots of at least steps_per_launch slots")
            _lib.check(self._lib.cavoid_step_autoreset_n_timed(
                self._h, self._ptr(a), stride, n, spl, self.num_worlds if slots is not None else 0, self._ptr(o.obs), self._ptr(o.rewards),
                self._ptr(o.done), self._ptr(o.game_over), self._stream(), C.byref(ms)), "cavoid_step_autoreset_n_timed")
            k = -(-n // spl)
            total += ms.value * k
            launches += k
            left -= n
        return total / launches

    def timer_begin(self) -> None:
        _lib.check(self._lib.cavoid_timer_begin(self._h, self._stream()), "cavoid_timer_begin")

    def timer_end(self) -> float:
        ms = C.c_float(0.0)
        _lib.check(self._lib.cavoid_timer_end(self._h, self._stream(), C.byref(ms)), "cavoid_timer_end")
        return float(ms.value)
