"""ctypes binding of the C ABI declared in ``include/cavoid.h``.

The product path has exactly one backend: the HIP library.  If ``libcavoid_hip.so`` is missing or
does not load, importing the env raises -- there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB_PATH as _DEFAULT_LIB_PATH

# CAVOID_LIB selects another build of the SAME HIP library (e.g. the phase-tracing development
# variant); it is not a fallback mechanism.
LIB_PATH = os.environ.get("CAVOID_LIB", _DEFAULT_LIB_PATH)

MAX_ACTIONS = 32
MAX_AGENTS = 16
ABI_VERSION = 3

F_AT_GOAL, F_RAN_OUT, F_IN_COLL, F_WAS_AT_GOAL, F_WAS_IN_COLL, F_PRESENT, F_LEARNING = 1, 2, 4, 8, 16, 32, 64
F_POLICY_SHIFT = 8
F_POLICY_MASK = 7
POLICY_EXTERNAL, POLICY_STATIC, POLICY_NONCOOP, POLICY_RVO, POLICY_FROZEN_NET = 0, 1, 2, 3, 4
F_DONE_MASK = 7
COMM_FORCE_RCCL = 1


class CavoidCfg(C.Structure):
    """Mirror of ``struct cavoid_cfg`` (include/cavoid.h)."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
        ("max_agents", C.c_int32), ("max_other", C.c_int32), ("sort_method", C.c_int32), ("dynamics", C.c_int32),
        ("actions_fp32", C.c_int32), ("timeout_enabled", C.c_int32), ("num_actions", C.c_int32), ("evaluate_mode", C.c_int32),
        ("time_budget_from_goal_edge", C.c_int32), ("wrap_closed_end", C.c_int32), ("done_agents_collide", C.c_int32),
        ("sort_round_gap", C.c_int32), ("sort_tie_lateral", C.c_int32), ("gen_lookahead", C.c_int32),
        ("dt", C.c_double), ("near_goal_threshold", C.c_double), ("max_time_ratio", C.c_double),
        ("collision_dist", C.c_double), ("getting_close_range", C.c_double), ("reward_at_goal", C.c_double),
        ("reward_collision", C.c_double), ("reward_getting_close", C.c_double), ("reward_time_step", C.c_double),
        ("close_penalty_slope", C.c_double), ("reward_clip_lo", C.c_double), ("reward_clip_hi", C.c_double),
        ("sensing_horizon", C.c_double), ("max_turn_rate", C.c_double),
        ("actions", (C.c_double * 2) * MAX_ACTIONS),
        ("gen_min_agents", C.c_int32), ("gen_max_agents", C.c_int32),
        ("gen_nonlearning_fraction", C.c_double), ("gen_static_fraction", C.c_double),
        ("gen_goal_jitter", C.c_double), ("gen_angle_jitter", C.c_double),
        ("gen_pool_size", C.c_int32),
        ("gen_mode", C.c_int32), ("gen_box_large_from", C.c_int32), ("gen_pool_epoch", C.c_uint32), ("rvo_enabled", C.c_int32),
        ("gen_rvo_fraction", C.c_double), ("gen_frozen_fraction", C.c_double), ("gen_box_small", C.c_double * 2), ("gen_box_large", C.c_double * 2),
        ("gen_min_trip", C.c_double),
        ("rvo_time_horizon", C.c_double), ("rvo_collab_coeff", C.c_double), ("rvo_radius_scale", C.c_double),
        ("rvo_max_delta_heading", C.c_double),
    ]


class CavoidPolicyWeights(C.Structure):
    """Mirror of ``struct cavoid_policy_weights`` (include/cavoid.h)."""
    _fields_ = [("struct_size", C.c_int32), ("min_policy", C.c_float), ("forget_bias", C.c_float), ("with_backward", C.c_int32)] + [
        (n, C.c_void_p) for n in ("avg", "std", "lstm_kernel", "lstm_bias", "layer1_kernel", "layer1_bias", "layer2_kernel",
                                  "layer2_bias", "fc1_kernel", "fc1_bias", "p_kernel", "p_bias", "v_kernel", "v_bias")]


class CavoidPolicyTrainBuffers(C.Structure):
    """Mirror of ``struct cavoid_policy_train_buffers`` (include/cavoid.h)."""
    _fields_ = [("struct_size", C.c_int32), ("reserved", C.c_int32), ("capacity_rows", C.c_int64)] + [
        (n, C.c_void_p) for n in ("z1", "z2", "z3", "l1_in", "h_in", "save", "gh", "loss", "g1", "g2", "g3", "gl", "db")]


class CavoidRolloutBuffers(C.Structure):
    """Mirror of ``struct cavoid_rollout_buffers`` (include/cavoid.h): the experience store handed to ``cavoid_actor_run``."""
    _fields_ = [("struct_size", C.c_int32), ("reserved", C.c_int32)] + [
        (n, C.c_void_p) for n in ("x", "val", "ret", "act", "emit_t", "dup_x", "dup_r", "dup_a", "dup_src", "dup_count")] + [
        ("dup_capacity", C.c_int64), ("ep_out", C.c_void_p), ("ep_count", C.c_void_p), ("ep_capacity", C.c_int64)]


class CavoidError(RuntimeError):
    def __init__(self, code: int, where: str):
        msg = lib().cavoid_strerror(code).decode()
        if code == -3:
            msg += " [hipError_t=%d]" % lib().cavoid_last_hip_error()
        if code == -6:
            msg += " [ncclResult_t=%d]" % lib().cavoid_last_comm_error()
        super().__init__("%s: %s (code %d)" % (where, msg, code))
        self.code = code


# every symbol include/cavoid.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("cavoid_abi_version", C.c_int, []),
    ("cavoid_strerror", C.c_char_p, [C.c_int]),
    ("cavoid_last_hip_error", C.c_int, []),
    ("cavoid_default_cfg", C.c_int, [C.POINTER(CavoidCfg), C.c_int32, C.c_int32]),
    ("cavoid_default_actions", C.c_int, [_P, C.POINTER(C.c_int32)]),
    ("cavoid_create", C.c_int, [C.POINTER(CavoidCfg), C.c_int64, C.c_int64, C.c_int, C.POINTER(_P)]),
    ("cavoid_destroy", None, [_P]),
    ("cavoid_num_worlds", C.c_int64, [_P]),
    ("cavoid_obs_width", C.c_int32, [_P]),
    ("cavoid_seed", C.c_int, [_P, C.c_uint64, _P, _P]),
    ("cavoid_get_episode", C.c_int, [_P, _P, _P]),
    ("cavoid_pool_refresh", C.c_int, [_P, C.c_uint32, _P]),
    ("cavoid_set_state", C.c_int, [_P, _P, _P, _P, _P]),
    ("cavoid_get_state", C.c_int, [_P, _P, _P, _P, _P]),
    ("cavoid_reset", C.c_int, [_P, _P, _P, _P]),
    ("cavoid_observe", C.c_int, [_P, _P, _P]),
    ("cavoid_step", C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    ("cavoid_step_continuous", C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    ("cavoid_step_autoreset", C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    ("cavoid_step_autoreset_n", C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int64, _P, _P, _P, _P, _P]),
    ("cavoid_step_continuous_autoreset", C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    ("cavoid_step_continuous_autoreset_n", C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int64, _P, _P, _P, _P, _P]),
    ("cavoid_step_continuous_autoreset_packed", C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int64, _P, _P, _P]),
    ("cavoid_step_autoreset_n_timed", C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int64, _P, _P, _P, _P, _P, C.POINTER(C.c_float)]),
    ("cavoid_policy_rows", C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P]),
    ("cavoid_packed_width", C.c_int32, [_P]),
    ("cavoid_reset_packed", C.c_int, [_P, _P, _P, _P]),
    ("cavoid_observe_packed", C.c_int, [_P, _P, _P]),
    ("cavoid_step_packed", C.c_int, [_P, _P, _P, _P, _P]),
    ("cavoid_step_autoreset_packed", C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int64, _P, _P, _P]),
    ("cavoid_comm_unique_id", C.c_int, [_P]),
    ("cavoid_comm_create", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int, C.POINTER(_P)]),
    ("cavoid_comm_create_ex", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int, C.c_uint32, C.POINTER(_P)]),
    ("cavoid_comm_info", C.c_int, [_P] + [C.POINTER(C.c_int32)] * 4),
    ("cavoid_comm_destroy", None, [_P]),
    ("cavoid_gather_begin", C.c_int, [_P, C.c_int32, _P, _P, C.c_int64, _P]),
    ("cavoid_gatherv_begin", C.c_int, [_P, C.c_int32, _P, _P, C.POINTER(C.c_int64), C.c_int32, _P]),
    ("cavoid_gather_wait", C.c_int, [_P, C.c_int32, _P]),
    ("cavoid_last_comm_error", C.c_int, []),
    ("cavoid_rollout_create", C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_int, C.POINTER(_P)]),
    ("cavoid_rollout_destroy", None, [_P]),
    ("cavoid_rollout_reset", C.c_int, [_P, _P]),
    ("cavoid_rollout_push", C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int32] + [_P] * 10 + [C.c_int64, _P, _P, C.c_int64, _P]),
    ("cavoid_actor_run", C.c_int, [_P, _P, _P, C.POINTER(CavoidRolloutBuffers)] + [_P] * 7 + [C.c_int32, C.c_int32, _P]),
    ("cavoid_actor_run_mix", C.c_int, [_P, _P, _P, _P, C.POINTER(CavoidRolloutBuffers)] + [_P] * 7 + [C.c_int32, C.c_int32, _P]),
    ("cavoid_step_push", C.c_int, [_P, _P, C.POINTER(CavoidRolloutBuffers)] + [_P] * 7 + [C.c_int32, _P]),
    ("cavoid_rollout_compact", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32] + [_P] * 9 + [C.c_int64, _P]),
    ("cavoid_rollout_active_rows", C.c_int, [_P] * 7),
    ("cavoid_policy_forward_rows", C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, _P, _P, _P, _P, C.c_int32, _P]),
    ("cavoid_policy_create", C.c_int, [C.c_int32, C.c_int32, C.c_int, C.POINTER(_P)]),
    ("cavoid_policy_destroy", None, [_P]),
    ("cavoid_policy_load", C.c_int, [_P, C.POINTER(CavoidPolicyWeights), _P]),
    ("cavoid_policy_seed", C.c_int, [_P, C.c_uint64, _P]),
    ("cavoid_policy_info", C.c_int, [_P, _P] + [C.POINTER(C.c_int32)] * 3),
    ("cavoid_policy_forward", C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, _P, _P, C.c_int32, _P]),
    ("cavoid_policy_train", C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, _P, C.c_float, C.c_float, C.POINTER(CavoidPolicyTrainBuffers), _P]),
    ("cavoid_timer_begin", C.c_int, [_P, _P]),
    ("cavoid_timer_end", C.c_int, [_P, _P, C.POINTER(C.c_float)]),
]

_lib = None


def lib():
    """Load ``libcavoid_hip.so`` (built in-tree by ``rl_collision_avoidance_amd.build``)."""
    global _lib
    if _lib is None:
        if "CAVOID_LIB" not in os.environ:
            from . import build as _build
            if os.path.exists(LIB_PATH) and _build.is_stale():
                # the binary does not match the in-tree sources.  Never rebuild behind an import's back (minutes of hipcc):
                # CAVOID_AUTO_REBUILD=1 opts a development box into it, everyone else gets told what to run.
                if os.environ.get("CAVOID_AUTO_REBUILD", "0") not in ("", "0"):
                    _build.build()
                else:
                    raise ImportError("%s does not match its sources: run `python -m rl_collision_avoidance_amd.build` "
                                      "(or set CAVOID_AUTO_REBUILD=1)" % LIB_PATH)
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -m rl_collision_avoidance_amd.build` (needs hipcc). "
                "There is no CPU fallback for the env.step hot path." % LIB_PATH)
        # PyTorch first: its wheel bundles the HIP runtime (libamdhip64) that owns the tensors we are
        # handed; loading it before our library makes the dynamic linker bind us to that SAME
        # runtime instead of pulling a second copy from /opt/rocm (two runtimes in one process:
        # hipGetDeviceCount fails / foreign pointers).
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(handle, name)      # AttributeError if the library lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.cavoid_abi_version() != ABI_VERSION:
            raise ImportError("libcavoid_hip.so ABI %d != binding ABI %d" % (handle.cavoid_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


def check(code: int, where: str) -> None:
    if code != 0:
        raise CavoidError(code, where)
