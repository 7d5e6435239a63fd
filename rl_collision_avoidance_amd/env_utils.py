"""``create_env()`` -- the constructor seam of the reference
(``from gym_collision_avoidance.experiments.src.env_utils import run_episode, create_env, store_stats``;
``env, one_env = create_env()``; /root/reference/ga3c/GA3C/Environment.py:54-56).

``SingleWorldVecEnv`` presents ONE world of a batched MI355X env with the nesting the reference's
consumers undo: ``observations[0]`` is the ``[N_max, 1+D]`` array (Environment.py:84-86),
``rewards[0]`` the per-agent rewards (ProcessAgent.py:151), ``game_over`` a scalar (:149),
``infos[0]`` the two dicts keyed by actual agent index (:155-157).  It is a plumbing adapter
(BASELINE configs[0]); throughput comes from driving ``BatchedCollisionAvoidanceEnv`` directly.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from .config import EnvConfig


def _load_config() -> EnvConfig:
    """Same selection rule as the reference's ``GA3C/__init__.py:3-12``: the class named by
    GYM_CONFIG_CLASS, loaded from GYM_CONFIG_PATH; default: the plain env Config."""
    cls_name = os.environ.get("GYM_CONFIG_CLASS")
    path = os.environ.get("GYM_CONFIG_PATH")
    if cls_name and path and os.path.exists(path):
        import importlib.util
        spec = importlib.util.spec_from_file_location("config_module", path)
        module = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(module)
        return getattr(module, cls_name)()
    return EnvConfig()


class SingleWorldVecEnv(object):
    """VecEnv-of-one facade over world ``world`` of a batched backend.

    backend: object with ``reset()``, ``step(actions[W,N] int)`` returning
    ``(obs[W,N,1+D], rewards[W,N], done[W,N], game_over[W])`` (torch tensors or ndarrays) and
    attributes ``max_agents`` / ``num_worlds`` -- i.e. ``BatchedCollisionAvoidanceEnv``."""

    def __init__(self, backend, world: int = 0):
        self.backend = backend
        self.world = int(world)
        self.max_agents = int(backend.max_agents)
        self._last_obs = None

    @staticmethod
    def _np(t) -> np.ndarray:
        if hasattr(t, "detach"):
            t = t.detach().cpu().numpy()
        return np.asarray(t)

    def _actions_tensor(self, actions: np.ndarray):
        try:
            import torch
            dev = getattr(self.backend, "device", None)
            if dev is not None:
                return torch.from_numpy(actions).to(dev)
        except ImportError:
            pass
        return actions

    def reset(self) -> List[np.ndarray]:
        obs = self._np(self.backend.reset())[self.world].astype(np.float64)
        self._last_obs = obs
        return [obs]

    def step(self, actions: Sequence[Dict[int, int]]):
        """``actions`` = ``[ {agent index: action index} ]`` holding the learning agents only
        (ProcessAgent.py:124,144,149)."""
        per_agent = actions[0] if isinstance(actions, (list, tuple)) else actions
        a = np.zeros((self.backend.num_worlds, self.max_agents), dtype=np.int32)
        for idx, act in dict(per_agent).items():
            a[self.world, int(idx)] = int(act)
        obs, rew, done, game_over = self.backend.step(self._actions_tensor(a))
        obs = self._np(obs)[self.world].astype(np.float64)
        rew = self._np(rew)[self.world].astype(np.float64)
        done = self._np(done)[self.world].astype(bool)
        present = np.flatnonzero(obs[:, 4] > 0.0)              # pref_speed > 0 <=> the row holds an agent
        n = int(present.max()) + 1 if present.size else 0
        info = {"which_agents_done": {i: bool(done[i]) for i in range(n)},
                "which_agents_learning": {i: bool(obs[i, 0] > 0.5) for i in range(n)}}
        self._last_obs = obs
        return [obs], [rew[:n]], bool(self._np(game_over)[self.world]), [info]


def create_env(config: Optional[EnvConfig] = None, device="cuda:0", seed: Optional[int] = None, **cfg_overrides):
    """-> ``(env, one_env)`` like the reference: ``env`` is the VecEnv-shaped object GA3C steps,
    ``one_env`` the underlying (here: batched, 1-world) environment."""
    from .batched_env import BatchedCollisionAvoidanceEnv
    config = config or _load_config()
    if seed is None:
        seed = int(getattr(config, "RANDOM_SEED_1000", 0)) * 1000
    cfg_overrides.setdefault("gen_min_agents", min(2, int(config.MAX_NUM_AGENTS_IN_ENVIRONMENT)))
    one_env = BatchedCollisionAvoidanceEnv(1, config, device=device, seed=seed, **cfg_overrides)
    return SingleWorldVecEnv(one_env, 0), one_env


def run_episode(env, one_env=None):
    """Roll one episode with uniformly random actions; returns ``(total_reward, steps)``.
    (Imported, unused, by Environment.py:54.)"""
    obs = env.reset()[0]
    total, steps, over = 0.0, 0, False
    num_actions = int(getattr(env.backend, "num_actions", 11))
    while not over:
        acts = {i: int(np.random.randint(num_actions)) for i in range(env.max_agents) if obs[i, 0] > 0.5}
        o, r, over, _ = env.step([acts])
        obs = o[0]
        total += float(np.sum(r[0]))
        steps += 1
    return total, steps


def store_stats(*args, **kwargs):
    """Imported by Environment.py:54 and never called there; kept so the import resolves."""
    return None
