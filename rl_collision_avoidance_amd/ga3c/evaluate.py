"""``evaluate`` -- EVALUATE_MODE runs (Config.EVALUATE_MODE: every agent of a world must finish; actions are the argmax,
/root/reference/ga3c/GA3C/ProcessAgent.py:98-103) with the per-agent outcome statistics the collision-avoidance papers
report: reached the goal / collided / ran out of time, and the extra time to goal.  Worlds are stepped WITHOUT auto-reset
(``cavoid_step``), so that every agent's terminal flags can be read from the world buffer once its world is over."""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from .. import _lib
from ..batched_env import BatchedCollisionAvoidanceEnv


@torch.no_grad()
def evaluate(env: BatchedCollisionAvoidanceEnv, policy: Callable, rounds: int = 1, greedy: bool = True,
             max_steps: Optional[int] = None, frozen_policy=None) -> Dict[str, float]:
    """Run ``rounds`` batches of ``env.num_worlds`` episodes.  ``policy``: a ``FusedPolicy`` (its ``act`` is used) or any
    callable ``x [B, NN_INPUT_SIZE] -> (p [B, A], v [B])``.  Returns rates over the learning agents that took part."""
    W, N = env.num_worlds, env.max_agents
    dt = float(env.cfg.dt)
    tot = {"agents": 0, "goal": 0, "collision": 0, "timeout": 0, "reward": 0.0, "steps": 0, "time_to_goal": 0.0, "extra_time": 0.0}
    for _ in range(rounds):
        obs = env.reset()
        _, f32_0, fl_0 = env.get_state()
        f64_0 = env.get_state()[0]
        present = (fl_0 & _lib.F_PRESENT) != 0
        learning = present & ((fl_0 & _lib.F_LEARNING) != 0)
        # straight-line time at the preferred speed = the yardstick for "extra time to goal"
        dist = torch.sqrt((f64_0[0] - f32_0[0].double()) ** 2 + (f64_0[1] - f32_0[1].double()) ** 2)
        straight = ((dist - float(env.cfg.near_goal_threshold)).clamp_min(0.0) / f32_0[3].double().clamp_min(1e-6))
        t_goal = torch.zeros(W * N, dtype=torch.float64, device=env.device)
        reached = torch.zeros(W * N, dtype=torch.bool, device=env.device)
        ret = torch.zeros(W * N, dtype=torch.float64, device=env.device)
        limit = max_steps if max_steps is not None else 100000
        for step in range(1, limit + 1):
            x = obs.view(W * N, -1)[:, 1:]
            if hasattr(policy, "act"):
                actions = policy.act(x, greedy=greedy)[0]
            else:
                p = policy(x.contiguous())[0]
                actions = p.argmax(dim=1) if greedy else torch.multinomial(p, 1).squeeze(1)
            actions = actions.to(torch.int32).contiguous()
            if frozen_policy is not None:                  # frozen-network agents (scripted policy 4) take their own network's argmax
                frozen_policy.act(x, greedy=True, rows=env.policy_rows(_lib.POLICY_FROZEN_NET), actions_out=actions)
            obs, rew, done, over = env.step(actions.view(W, N))
            ret += rew.view(-1).double()
            fl = env.get_state()[2]
            now = ((fl & _lib.F_AT_GOAL) != 0) & ~reached
            t_goal[now] = step * dt
            reached |= now
            if bool(over.all()):
                break
        fl = env.get_state()[2]
        goal = learning & ((fl & _lib.F_AT_GOAL) != 0)
        coll = learning & ((fl & _lib.F_IN_COLL) != 0) & ~goal
        tout = learning & ((fl & _lib.F_RAN_OUT) != 0) & ~goal & ~coll
        tot["agents"] += int(learning.sum()); tot["goal"] += int(goal.sum()); tot["collision"] += int(coll.sum())
        tot["timeout"] += int(tout.sum()); tot["reward"] += float(ret[learning].sum()); tot["steps"] += step
        tot["time_to_goal"] += float(t_goal[goal].sum()); tot["extra_time"] += float((t_goal[goal] - straight[goal]).sum())
    n, g = max(tot["agents"], 1), max(tot["goal"], 1)
    return {"agents": tot["agents"], "success_rate": tot["goal"] / n, "collision_rate": tot["collision"] / n,
            "timeout_rate": tot["timeout"] / n, "unfinished_rate": 1.0 - (tot["goal"] + tot["collision"] + tot["timeout"]) / n,
            "mean_reward": tot["reward"] / n, "mean_time_to_goal": tot["time_to_goal"] / g,
            "mean_extra_time_to_goal": tot["extra_time"] / g, "env_steps": tot["steps"]}
