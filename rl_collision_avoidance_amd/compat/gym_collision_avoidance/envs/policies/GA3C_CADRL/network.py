# reference import: ga3c/GA3C/Server.py:36  `from gym_collision_avoidance.envs.policies.GA3C_CADRL.network import Actions`
from rl_collision_avoidance_amd.actions import Actions  # noqa: F401
