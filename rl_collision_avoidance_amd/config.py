"""``EnvConfig`` -- the env-side configuration class GA3C's ``Train`` config inherits from
(``from gym_collision_avoidance.envs.config import Config as EnvConfig``,
/root/reference/ga3c/GA3C/Config.py:29,31,52).

The upstream class is not in the reference tree (empty submodule); the scalar values below are
the ones recorded in ``ga3c/GA3C/checkpoints/regression/wandb/run-ws/config.yaml`` and the
attribute names are the ones GA3C reads: ``DT`` (Config.py:104), ``STATE_INFO_DICT``
(Config.py:67-71), ``PLAY_MODE / EVALUATE_MODE / TRAIN_MODE`` (ProcessAgent.py:99,228;
Server.py:135), ``TRAIN_SINGLE_AGENT`` (ProcessAgent.py:152).  ``STATE_INFO_DICT`` means/stds
are not recorded in-tree; the values here follow the published
GA3C-CADRL normalisation and only affect the network input scaling, never ``env.step``.
"""
from __future__ import annotations

import numpy as np


class EnvConfig(object):
    def __init__(self):
        # --- modes (run-ws/config.yaml:55-57,189-191,293-298) ----------------------------------
        self.TRAIN_MODE = True
        self.PLAY_MODE = False
        self.EVALUATE_MODE = False
        self.TRAIN_SINGLE_AGENT = False
        # --- simulation scalars ------------------------------------------------------------------
        self.DT = 0.2                         # :43-45
        self.NEAR_GOAL_THRESHOLD = 0.2        # :124-126
        self.MAX_TIME_RATIO = 2.0             # :115-117
        self.COLLISION_DIST = 0.0             # :31-33
        self.GETTING_CLOSE_RANGE = 0.2        # :64-66
        self.SENSING_HORIZON = np.inf         # :249-251
        self.AGENT_SORTING_METHOD = "closest_last"   # :9-11 (run-ws-4: closest_first)
        # --- rewards (:201-221) ------------------------------------------------------------------
        self.REWARD_AT_GOAL = 1.0
        self.REWARD_COLLISION_WITH_AGENT = -0.25
        self.REWARD_COLLISION_WITH_WALL = -0.25
        self.REWARD_GETTING_CLOSE = -0.1
        self.REWARD_ENTERED_NORM_ZONE = -0.05
        self.REWARD_TIME_STEP = 0.0
        self.REWARD_WIGGLY_BEHAVIOR = 0.0
        self.WIGGLY_BEHAVIOR_THRESHOLD = np.inf
        # --- sizes: a GA3C subclass sets these BEFORE calling EnvConfig.__init__ (Config.py:33-52)
        if not hasattr(self, "MAX_NUM_AGENTS_IN_ENVIRONMENT"):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = 4
        if not hasattr(self, "MAX_NUM_AGENTS_TO_SIM"):
            self.MAX_NUM_AGENTS_TO_SIM = self.MAX_NUM_AGENTS_IN_ENVIRONMENT
        if not hasattr(self, "MAX_NUM_OTHER_AGENTS_OBSERVED"):
            self.MAX_NUM_OTHER_AGENTS_OBSERVED = self.MAX_NUM_AGENTS_IN_ENVIRONMENT - 1
        self.MAX_NUM_OTHER_AGENTS_IN_ENVIRONMENT = self.MAX_NUM_AGENTS_IN_ENVIRONMENT - 1
        if not hasattr(self, "STATES_IN_OBS"):
            self.STATES_IN_OBS = ['is_learning', 'num_other_agents', 'dist_to_goal', 'heading_ego_frame',
                                  'pref_speed', 'radius', 'other_agents_states']
        if not hasattr(self, "STATES_NOT_USED_IN_POLICY"):
            self.STATES_NOT_USED_IN_POLICY = ['is_learning']
        self.TEST_CASE_FN = "get_testcase_random"   # :281-283 (upstream's generator and its np.random stream are unknowable:
        # SURVEY App. A U9; the library's own counter-based generators stand in, selected by the attributes below)
        # --- scenario generator and scripted agents (the reference's training data used "static / non-coop /
        #     RVO" mixes, checkpoints/RL/wandb/run-2018-backup/checkpoints/index.txt:1-3).  Subclasses may set them before __init__.
        for name, default in (("TEST_CASE_GENERATOR", "ring"),        # "ring" = GEN v1 (antipodal goals), "box" = GEN v2 (random boxes)
                              ("SCRIPTED_AGENT_FRACTION", 0.0),       # P(an agent other than agent 0 runs a scripted policy)
                              ("SCRIPTED_STATIC_FRACTION", 0.5),      # of those: P(static)
                              ("SCRIPTED_RVO_FRACTION", 0.0),         # ... P(RVO / ORCA)
                              ("SCRIPTED_FROZEN_NET_FRACTION", 0.0),  # ... P(driven by a frozen network: the GA3C-CADRL agent);
                                                                      #     the rest are non-cooperative
                              ("RVO_TIME_HORIZON", 5.0),              # run-ws/config.yaml:237-239
                              ("RVO_COLLAB_COEFF", 0.5)):             # :234-236
            if not hasattr(self, name):
                setattr(self, name, default)
        self.ACTION_SPACE_TYPE = 0            # continuous at the gym level; discretised by the policy
        self.NUM_TEST_CASES = 50
        self.USE_STATIC_MAP = False
        self.LASERSCAN_LENGTH = 512
        self.NUM_STEPS_IN_OBS_HISTORY = 1
        self.NUM_PAST_ACTIONS_IN_STATE = 0

        M = self.MAX_NUM_OTHER_AGENTS_OBSERVED
        other_mean = np.array([0.0, 0.0, 0.0, 0.0, 0.5, 0.0, 1.0], dtype=np.float32)
        other_std = np.array([5.0, 5.0, 1.0, 1.0, 1.0, 5.0, 1.0], dtype=np.float32)
        f32 = np.float32
        self.STATE_INFO_DICT = {
            'is_learning': {'dtype': f32, 'size': 1, 'bounds': [0., 1.]},
            'num_other_agents': {'dtype': f32, 'size': 1, 'bounds': [0, np.inf],
                                 'mean': np.array([1.0], f32), 'std': np.array([1.0], f32)},
            'dist_to_goal': {'dtype': f32, 'size': 1, 'bounds': [-np.inf, np.inf],
                             'mean': np.array([0.0], f32), 'std': np.array([5.0], f32)},
            'heading_ego_frame': {'dtype': f32, 'size': 1, 'bounds': [-np.pi, np.pi],
                                  'mean': np.array([0.0], f32), 'std': np.array([3.14], f32)},
            'pref_speed': {'dtype': f32, 'size': 1, 'bounds': [0, np.inf],
                           'mean': np.array([1.0], f32), 'std': np.array([1.0], f32)},
            'radius': {'dtype': f32, 'size': 1, 'bounds': [0, np.inf],
                       'mean': np.array([0.5], f32), 'std': np.array([1.0], f32)},
            'other_agents_states': {'dtype': f32, 'size': (M, 7), 'bounds': [-np.inf, np.inf],
                                    'mean': np.tile(other_mean, (M, 1)), 'std': np.tile(other_std, (M, 1))},
        }

    # ------------------------------------------------------------------------------------------
    @property
    def OBS_WIDTH(self) -> int:
        """Floats per agent row of the env observation (1 + NN_INPUT_SIZE)."""
        return 6 + 7 * self.MAX_NUM_OTHER_AGENTS_OBSERVED


Config = EnvConfig
