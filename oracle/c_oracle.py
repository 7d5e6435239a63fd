"""ctypes front-end of the C oracle (test infrastructure, NOT product code).

Loads ``oracle/libcavoid_oracle.so`` (built by ``oracle/Makefile``) and exposes it on NumPy
arrays.  PARITY UNPINNED for the env half -- see ``cavoid_oracle.c``.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcavoid_oracle.so")
MAX_ACTIONS = 32


class OracleCfg(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "dt", "near_goal_threshold", "max_time_ratio", "collision_dist", "getting_close_range",
        "reward_at_goal", "reward_collision", "reward_getting_close", "reward_time_step",
        "sensing_horizon", "close_penalty_slope", "max_turn_rate", "reward_clip_lo", "reward_clip_hi",
        "rvo_time_horizon", "rvo_collab_coeff", "rvo_radius_scale", "rvo_max_delta_heading")] + [
        (n, C.c_int32) for n in ("max_agents", "max_other", "sort_method", "actions_fp32",
                                 "timeout_enabled", "dynamics", "num_actions", "evaluate_mode",
                                 "time_budget_from_goal_edge", "wrap_closed_end", "done_agents_collide",
                                 "sort_round_gap", "sort_tie_lateral", "_pad0")] + [
        ("actions", (C.c_double * 2) * MAX_ACTIONS)]


class OracleGen(C.Structure):
    _fields_ = [("min_agents", C.c_int32), ("max_agents", C.c_int32), ("nonlearning_fraction", C.c_double),
                ("static_fraction", C.c_double), ("goal_jitter", C.c_double), ("angle_jitter", C.c_double),
                ("pool_size", C.c_int32), ("mode", C.c_int32), ("rvo_fraction", C.c_double),
                ("box_small", C.c_double * 2), ("box_large", C.c_double * 2), ("min_trip", C.c_double),
                ("box_large_from", C.c_int32), ("pool_epoch", C.c_uint32), ("frozen_fraction", C.c_double)]


class _State(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("px", "py", "heading", "t_remaining", "gx", "gy", "radius",
                                          "pref_speed", "speed", "flags")]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "cavoid_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libcavoid_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_default_cfg.argtypes = [C.POINTER(OracleCfg), C.c_int32, C.c_int32]
        _lib.oracle_step.argtypes = [C.POINTER(OracleCfg), C.c_int64, C.POINTER(_State), C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.oracle_observe.argtypes = [C.POINTER(OracleCfg), C.c_int64, C.POINTER(_State), C.c_void_p]
        _lib.oracle_generate.argtypes = [C.POINTER(OracleCfg), C.POINTER(OracleGen), C.c_uint64, C.c_int64,
                                         C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(_State)]
        _lib.oracle_step_autoreset.argtypes = [C.POINTER(OracleCfg), C.POINTER(OracleGen), C.c_uint64, C.c_int64,
                                               C.c_void_p, C.c_int64, C.POINTER(_State), C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.oracle_step_autoreset_any.argtypes = [C.POINTER(OracleCfg), C.POINTER(OracleGen), C.c_uint64, C.c_int64,
                                                   C.c_void_p, C.c_int64, C.POINTER(_State), C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        for f in (_lib.oracle_default_cfg, _lib.oracle_step, _lib.oracle_observe, _lib.oracle_generate,
                  _lib.oracle_step_autoreset, _lib.oracle_step_autoreset_any):
            f.restype = None
    return _lib


def default_cfg(max_agents: int = 4, max_other: int | None = None, **overrides) -> OracleCfg:
    cfg = OracleCfg()
    lib().oracle_default_cfg(C.byref(cfg), max_agents, max_agents - 1 if max_other is None else max_other)
    for k, v in overrides.items():
        if k == "actions":
            set_actions(cfg, v)
        else:
            if not hasattr(cfg, k):
                raise AttributeError(k)
            setattr(cfg, k, v)
    return cfg


def set_actions(cfg: OracleCfg, table) -> None:
    table = np.asarray(table, dtype=np.float64)
    assert table.ndim == 2 and table.shape[1] == 2 and len(table) <= MAX_ACTIONS
    cfg.num_actions = len(table)
    for r, (a, b) in enumerate(table):
        cfg.actions[r][0], cfg.actions[r][1] = a, b


def default_gen(min_agents: int = 4, max_agents: int = 4, nonlearning_fraction: float = 0.0,
                static_fraction: float = 0.5, goal_jitter: float = 0.5, angle_jitter: float = 0.25,
                pool_size: int = 0, mode: int = 0, rvo_fraction: float = 0.0, box_small=(4.0, 5.0), box_large=(6.0, 8.0),
                box_large_from: int = 5, min_trip: float = 1.0, pool_epoch: int = 0, frozen_fraction: float = 0.0) -> OracleGen:
    g = OracleGen(min_agents, max_agents, nonlearning_fraction, static_fraction, goal_jitter, angle_jitter, pool_size, mode,
                  rvo_fraction)
    g.box_small[0], g.box_small[1] = box_small
    g.box_large[0], g.box_large[1] = box_large
    g.min_trip, g.box_large_from, g.pool_epoch, g.frozen_fraction = min_trip, box_large_from, pool_epoch, frozen_fraction
    return g


@dataclass
class State:
    """SoA state of W worlds x N agents (flat agent index a = w*N + i)."""
    f64: np.ndarray      # [4, W*N]  px, py, heading, t_remaining
    f32: np.ndarray      # [5, W*N]  gx, gy, radius, pref_speed, speed
    flags: np.ndarray    # [W*N] uint32

    @classmethod
    def empty(cls, W: int, N: int) -> "State":
        return cls(np.zeros((4, W * N), np.float64), np.zeros((5, W * N), np.float32), np.zeros(W * N, np.uint32))

    def copy(self) -> "State":
        return State(self.f64.copy(), self.f32.copy(), self.flags.copy())

    def _c(self) -> _State:
        assert self.f64.flags.c_contiguous and self.f32.flags.c_contiguous and self.flags.flags.c_contiguous
        assert self.f64.dtype == np.float64 and self.f32.dtype == np.float32 and self.flags.dtype == np.uint32
        p64 = [self.f64[k].ctypes.data for k in range(4)]
        p32 = [self.f32[k].ctypes.data for k in range(5)]
        return _State(*p64, *p32, self.flags.ctypes.data)


def _ptr(a):
    return None if a is None else a.ctypes.data


def step(cfg: OracleCfg, st: State, actions=None, cont=None):
    """-> obs f64 [W,N,width], rew f64 [W,N], done u8 [W,N], game_over u8 [W] (state updated in place)."""
    N = cfg.max_agents
    W = st.flags.size // N
    width = 6 + 7 * cfg.max_other
    obs = np.empty((W, N, width), np.float64)
    rew = np.empty((W, N), np.float64)
    done = np.empty((W, N), np.uint8)
    go = np.empty(W, np.uint8)
    if actions is not None:
        actions = np.ascontiguousarray(actions, dtype=np.int32).reshape(W, N)
        assert actions.min() >= 0 and actions.max() < cfg.num_actions
    if cont is not None:
        cont = np.ascontiguousarray(cont, dtype=np.float32).reshape(W, N, 2)
    cs = st._c()
    lib().oracle_step(C.byref(cfg), W, C.byref(cs), _ptr(actions), _ptr(cont), _ptr(obs), _ptr(rew), _ptr(done), _ptr(go))
    return obs, rew, done, go


def observe(cfg: OracleCfg, st: State) -> np.ndarray:
    N = cfg.max_agents
    W = st.flags.size // N
    obs = np.empty((W, N, 6 + 7 * cfg.max_other), np.float64)
    cs = st._c()
    lib().oracle_observe(C.byref(cfg), W, C.byref(cs), _ptr(obs))
    return obs


def generate(cfg: OracleCfg, gen: OracleGen, seed: int, st: State, episode: np.ndarray, mask=None,
             world_offset: int = 0) -> None:
    N = cfg.max_agents
    W = st.flags.size // N
    episode = np.ascontiguousarray(episode, dtype=np.uint32)
    assert episode.size == W and gen.max_agents <= N
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
    cs = st._c()
    lib().oracle_generate(C.byref(cfg), C.byref(gen), seed, world_offset, _ptr(episode), _ptr(mask), W, C.byref(cs))


def step_autoreset(cfg: OracleCfg, gen: OracleGen, seed: int, st: State, episode: np.ndarray, actions,
                   world_offset: int = 0, cont=None):
    N = cfg.max_agents
    W = st.flags.size // N
    width = 6 + 7 * cfg.max_other
    assert episode.dtype == np.uint32 and episode.flags.c_contiguous
    obs = np.empty((W, N, width), np.float64)
    rew = np.empty((W, N), np.float64)
    done = np.empty((W, N), np.uint8)
    go = np.empty(W, np.uint8)
    if actions is not None:
        actions = np.ascontiguousarray(actions, dtype=np.int32).reshape(W, N)
    if cont is not None:                                     # float [W,N,2] continuous / holonomic actions (oracle_step's second form)
        cont = np.ascontiguousarray(cont, dtype=np.float32).reshape(W, N, 2)
    assert (actions is None) != (cont is None)
    cs = st._c()
    lib().oracle_step_autoreset_any(C.byref(cfg), C.byref(gen), seed, world_offset, _ptr(episode), W, C.byref(cs),
                                    _ptr(actions), _ptr(cont), _ptr(obs), _ptr(rew), _ptr(done), _ptr(go))
    return obs, rew, done, go
