"""``Environment`` -- the env adapter GA3C's actor holds (mirror of
/root/reference/ga3c/GA3C/Environment.py:37-116, rows R1/R2 of SURVEY.md section 8a).

Same attributes and call signatures: ``latest_observations`` ([N, 1+D], with the is_learning
column), ``previous_state`` / ``current_state`` ([1, N, D], column 0 dropped -- the 1-deep frame
queue of Environment.py:41,64,88-91), ``total_reward``, ``reset()``,
``step(action, pid, count) -> (rewards, game_over, info)``."""
from __future__ import annotations

import numpy as np

from ..env_utils import create_env


class Environment(object):
    nb_frames = 1

    def __init__(self, id, game=None, **create_env_kwargs):
        self.id = id
        if game is None:
            game, self.one_env = create_env(**create_env_kwargs)
        self.game = game
        self.total_reward = 0
        self.latest_observations = None
        self.previous_state = self.current_state = None
        self._frames = []

    def _process_obs(self, observations) -> None:
        obs = observations[0]                     # undo the VecEnv nesting
        if obs.ndim == 3:
            obs = obs[0]                          # undo a multi-agent VecEnv wrapper
        self.latest_observations = obs
        self._frames = (self._frames + [obs[:, 1:]])[-self.nb_frames:]
        self.previous_state = self.current_state
        self.current_state = np.array(self._frames)

    def reset(self) -> None:
        self.total_reward = 0
        self._frames = []
        self._process_obs(self.game.reset())

    def step(self, action, pid=None, count=None):
        observations, rewards, game_over, info = self.game.step(action)
        self.total_reward += np.sum(rewards)
        self._process_obs(observations)
        return rewards, game_over, info

    def print_frame_q(self) -> int:
        return len(self._frames)
