# reference import: ga3c/GA3C/Environment.py:54
#   `from gym_collision_avoidance.experiments.src.env_utils import run_episode, create_env, store_stats`
from rl_collision_avoidance_amd.env_utils import create_env, run_episode, store_stats  # noqa: F401
