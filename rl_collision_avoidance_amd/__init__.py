"""rl_collision_avoidance_amd -- MI355X-native batched collision-avoidance ``env.step`` hot path
and GA3C rollout, behind the reference's own Python seams."""
from .config import EnvConfig  # noqa: F401

__all__ = ["EnvConfig", "BatchedCollisionAvoidanceEnv", "Actions"]


def __getattr__(name):
    if name == "BatchedCollisionAvoidanceEnv":
        from .batched_env import BatchedCollisionAvoidanceEnv
        return BatchedCollisionAvoidanceEnv
    if name == "Actions":
        from .actions import Actions
        return Actions
    raise AttributeError(name)
