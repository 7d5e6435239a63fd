"""``EpisodeStats`` -- the bookkeeping and console line of the reference's stats process, fed from the
device-side episode log instead of ``episode_log_q``.

Mirror of /root/reference/ga3c/GA3C/ProcessStats.py:54-111: total / rolling (window
``STAT_ROLLING_MEAN_WINDOW`` = 1000 episodes, Config.py:131) reward and frame counts, PPS =
total_frame_count / elapsed (:54-56), TPS = training steps / elapsed (:58-60), and the same table
line (:93-107) so logs stay comparable.  Frames are learning-agent steps: ``len(r_) + 1`` per
yielded chunk (ProcessAgent.py:237), which the rollout kernel accumulates per world."""
from __future__ import annotations

import time
from collections import deque
from typing import Deque, Iterable, Optional, Tuple

import numpy as np


class EpisodeStats(object):
    def __init__(self, window: int = 1000, print_every: int = 0, trainers: int = 1, predictors: int = 1, agents: int = 0):
        self.window = int(window)
        self.print_every = int(print_every)          # 0 = never print; the reference prints every episode
        self.trainer_count, self.predictor_count, self.agent_count = trainers, predictors, agents
        self.episode_count = 0
        self.training_count = 0
        self.total_frame_count = 0
        self.rolling_frame_count = 0
        self.rolling_reward = 0.0
        self.results: Deque[Tuple[float, float, int]] = deque()
        self.start_time = time.time()
        self.first_time = self.start_time
        self.reward_log = 0.0
        self.roll_reward_log = 0.0

    def PPS(self) -> float:
        return float(np.ceil(self.total_frame_count / max(time.time() - self.start_time, 1e-9)))

    FPS = PPS                                          # the reference calls it FPS()

    def TPS(self) -> float:
        return float(np.ceil(self.training_count / max(time.time() - self.start_time, 1e-9)))

    def add_training_steps(self, n: int = 1) -> None:
        self.training_count += int(n)

    def add_episode(self, reward: float, length: int, when: Optional[float] = None) -> Optional[str]:
        """One ``(episode_time, reward, length)`` record; returns the table line when it is due."""
        when = time.time() if when is None else when
        self.total_frame_count += int(length)
        self.episode_count += 1
        self.rolling_frame_count += int(length)
        self.rolling_reward += float(reward)
        if len(self.results) >= self.window:
            old_time, old_reward, old_length = self.results.popleft()
            self.rolling_frame_count -= old_length
            self.rolling_reward -= old_reward
            self.first_time = old_time
        self.results.append((when, float(reward), int(length)))
        self.reward_log = float(reward)
        self.roll_reward_log = self.rolling_reward / len(self.results)
        if self.print_every and self.episode_count % self.print_every == 0:
            return self.line(reward)
        return None

    def add_episodes(self, records: Iterable) -> None:
        """``records``: rows of (world, total_reward, total_length), e.g. ``BatchedRollout.drain_episodes().tolist()``."""
        for _, reward, length in records:
            line = self.add_episode(reward, int(round(length)))
            if line:
                print(line, flush=True)

    def line(self, reward: float) -> str:
        rpps = self.rolling_frame_count / max(time.time() - self.first_time, 1e-9)
        return ('[Time: %8d] [Episode: %8d Score: %10.4f] [RScore: %10.4f RPPS: %5d] [PPS: %5d TPS: %5d] '
                '[NT: %2d NP: %2d NA: %2d]' % (int(time.time() - self.start_time), self.episode_count, reward,
                                               self.roll_reward_log, rpps, self.PPS(), self.TPS(),
                                               self.trainer_count, self.predictor_count, self.agent_count))
