"""CPU-only tests of the host side: the C-ABI library loads without a GPU and exports every symbol
include/cavoid.h declares; error codes; config / action-table mirrors; the create_env() facade and
the Environment mirror against the REFERENCE's own Environment/ProcessAgent when /root/reference is
present (a CPU oracle world stands in for the GPU backend -- test infrastructure only)."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/ga3c"
HAVE_REF = os.path.isdir(os.path.join(REF, "GA3C"))


def test_library_exports_every_declared_symbol():
    from rl_collision_avoidance_amd import _lib
    header = open(os.path.join(ROOT, "include", "cavoid.h")).read()
    declared = set(re.findall(r"\b(cavoid_[a-z_0-9]+)\s*\(", header))
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    handle = _lib.lib()
    for name in declared:
        assert hasattr(handle, name)
    assert handle.cavoid_abi_version() == _lib.ABI_VERSION == 3


def test_error_codes_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this test is about the GPU-less box")
    from rl_collision_avoidance_amd import _lib
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv, make_cfg
    lib = _lib.lib()
    h = C.c_void_p()
    cfg = make_cfg()
    assert lib.cavoid_create(C.byref(cfg), 16, 0, 0, C.byref(h)) == -5          # CAVOID_ENODEVICE: no fallback
    cfg.max_agents = 99
    assert lib.cavoid_create(C.byref(cfg), 16, 0, 0, C.byref(h)) == -4
    cfg = make_cfg()
    cfg.struct_size = 8
    assert lib.cavoid_create(C.byref(cfg), 16, 0, 0, C.byref(h)) == -1
    assert b"invalid" in lib.cavoid_strerror(-1)
    with pytest.raises(RuntimeError):
        BatchedCollisionAvoidanceEnv(4, device="cpu")
    with pytest.raises(RuntimeError):
        BatchedCollisionAvoidanceEnv(4, device="cuda:0")


def test_config_and_actions_mirror():
    from oracle import cavoid_oracle as po
    from rl_collision_avoidance_amd.actions import Actions
    from rl_collision_avoidance_amd.batched_env import make_cfg
    from rl_collision_avoidance_amd.config import EnvConfig
    a = Actions()
    assert a.num_actions == 11 and np.array_equal(a.actions, po.build_action_table())
    cfg = make_cfg(EnvConfig())
    ocfg = po.OracleConfig()
    for mine, theirs in (("dt", "dt"), ("near_goal_threshold", "near_goal_threshold"), ("max_time_ratio", "max_time_ratio"),
                         ("collision_dist", "collision_dist"), ("getting_close_range", "getting_close_range"),
                         ("reward_at_goal", "reward_at_goal"), ("reward_collision", "reward_collision"),
                         ("reward_getting_close", "reward_getting_close"), ("reward_time_step", "reward_time_step"),
                         ("close_penalty_slope", "close_penalty_slope"), ("reward_clip_lo", "reward_clip_lo"),
                         ("reward_clip_hi", "reward_clip_hi"), ("max_turn_rate", "max_turn_rate")):
        assert getattr(cfg, mine) == getattr(ocfg, theirs), mine
    assert cfg.max_agents == 4 and cfg.max_other == 3 and EnvConfig().OBS_WIDTH == 27


class OracleBackend(object):
    """CPU stand-in with the BatchedCollisionAvoidanceEnv surface the facade uses (tests only)."""

    def __init__(self, N=4, seed=3):
        from oracle import c_oracle as co
        self.co, self.max_agents, self.num_worlds, self.num_actions = co, N, 1, 11
        self.cfg, self.gen, self.seed = co.default_cfg(N), co.default_gen(2, N, 0.3), seed
        self.st = co.State.empty(1, N)
        self.ep = np.full(1, 0xFFFFFFFF, np.uint32)

    def reset(self):
        self.ep += 1
        self.co.generate(self.cfg, self.gen, self.seed, self.st, self.ep)
        return self.co.observe(self.cfg, self.st)

    def step(self, actions):
        return self.co.step(self.cfg, self.st, np.asarray(actions))


def test_create_env_facade_contract():
    from rl_collision_avoidance_amd.env_utils import SingleWorldVecEnv
    env = SingleWorldVecEnv(OracleBackend())
    obs = env.reset()
    assert isinstance(obs, list) and obs[0].shape == (4, 27) and obs[0].dtype == np.float64
    learners = {i: 2 for i in range(4) if obs[0][i, 0]}
    o, r, over, info = env.step([learners])
    n = len(info[0]["which_agents_done"])
    assert o[0].shape == (4, 27) and len(r[0]) == n and isinstance(over, bool)
    assert set(info[0]) == {"which_agents_done", "which_agents_learning"}
    assert all(info[0]["which_agents_learning"][i] == bool(o[0][i, 0]) for i in range(n))
    assert np.all(o[0][n:] == 0)                      # absent agents: zero rows, is_learning == 0


@pytest.mark.skipif(not HAVE_REF, reason="reference tree not present")
def test_reference_code_runs_on_our_seam():
    """The reference's UNMODIFIED Config, Environment and ProcessAgent on top of create_env()'s facade."""
    sys.path[:0] = [os.path.join(ROOT, "rl_collision_avoidance_amd", "compat"), REF, os.path.join(REF, "GA3C")]
    np.product = np.prod
    os.environ["GYM_CONFIG_CLASS"] = "TrainPhase1"
    os.environ["GYM_CONFIG_PATH"] = os.path.join(REF, "GA3C", "Config.py")
    for m in [m for m in sys.modules if m == "GA3C" or m.startswith("GA3C.") or m in ("ProcessAgent", "Environment", "Experience")]:
        del sys.modules[m]
    from GA3C import Config
    assert Config.NN_INPUT_SIZE == 26 and Config.TIME_MAX == 20 and Config.NUM_ACTIONS == 11
    assert len(Config.NN_INPUT_AVG_VECTOR) == 26 and len(Config.NN_INPUT_STD_VECTOR) == 26
    import Environment as RefEnvironment
    import ProcessAgent as PA
    from rl_collision_avoidance_amd.env_utils import SingleWorldVecEnv
    from rl_collision_avoidance_amd.ga3c.environment import Environment as MyEnvironment

    ref_env = RefEnvironment.Environment.__new__(RefEnvironment.Environment)      # skip _set_env (would need a GPU)
    from queue import Queue
    ref_env.nb_frames, ref_env.frame_q, ref_env.total_reward = 1, Queue(maxsize=1), 0
    ref_env.previous_state = ref_env.current_state = None
    ref_env.game = SingleWorldVecEnv(OracleBackend(seed=9))
    mine = MyEnvironment(0, game=SingleWorldVecEnv(OracleBackend(seed=9)))

    # (a) my Environment mirror == the reference's Environment on the same game, step for step
    ref_env.reset(); mine.reset()
    rng = np.random.default_rng(0)
    for t in range(60):
        acts = {i: int(rng.integers(0, 11)) for i in range(4) if ref_env.latest_observations[i, 0]}
        r1, over1, info1 = ref_env.step([acts], 0, t)
        r2, over2, info2 = mine.step([acts], 0, t)
        assert over1 == over2 and info1 == info2 and np.array_equal(r1[0], r2[0])
        assert np.array_equal(ref_env.latest_observations, mine.latest_observations)
        assert np.array_equal(ref_env.previous_state, mine.previous_state)
        assert np.array_equal(ref_env.current_state, mine.current_state)
        assert ref_env.current_state.shape == (1, 4, 26)
        if over1:
            ref_env.reset(); mine.reset()

    # (b) the reference's actor loop runs episodes on the seam and yields trainer-shaped chunks
    agent = PA.ProcessAgent(0, None, None, None, Config.NUM_ACTIONS)
    agent.env = ref_env
    agent.predict = lambda obs_row: (np.full(11, 1.0 / 11), 0.1)
    np.random.seed(0)
    rows = 0
    for _ in range(3):
        for x_, r_, a_, score in agent.run_episode():
            assert x_.ndim == 2 and x_.shape[1] == 26 and a_.shape == (len(r_), 11) and a_.dtype == np.float32
            rows += len(r_)
    assert rows > 0


def test_header_is_plain_c(tmp_path):
    """include/cavoid.h is the drop-in boundary: it must compile as C99 (no C++, no torch types)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "hdr.c"
    src.write_text('#include <stdio.h>\n#include "cavoid.h"\nint main(void) { cavoid_cfg c; cavoid_policy_weights w; cavoid_policy_train_buffers b;\n'
                   '  (void)c; (void)w; (void)b; printf("%d %d %d\\n", (int)sizeof(c), (int)sizeof(w), (int)sizeof(b)); return 0; }\n')
    exe = tmp_path / "hdr"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                           str(src), "-o", str(exe)])
    from rl_collision_avoidance_amd import _lib
    sizes = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    # the ctypes mirrors of the binding must have the C compiler's layout
    assert sizes == [C.sizeof(_lib.CavoidCfg), C.sizeof(_lib.CavoidPolicyWeights), C.sizeof(_lib.CavoidPolicyTrainBuffers)]
