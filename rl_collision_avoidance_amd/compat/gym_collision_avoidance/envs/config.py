# reference import: ga3c/GA3C/Config.py:29  `from gym_collision_avoidance.envs.config import Config as EnvConfig`
from rl_collision_avoidance_amd.config import EnvConfig as Config  # noqa: F401
