"""Supervised initialisation of the policy before the RL phases -- the role of
/root/reference/ga3c/GA3C/Regression.py (``train_with_regression``, :60-160) and of ``cost_regression``
(NetworkVPCore.py:90-100,123,198-200): softmax cross-entropy on a teacher's action + the value regression term.

The reference regresses onto a recorded CADRL dataset (``.../datasets/...`` -- a Git-LFS stub in this checkout).  The
stand-in teacher here is the env's own scripted "non-cooperative" behaviour (full preferred speed, turn towards the goal),
expressed on the observation alone, so a dataset is whatever the batched env produces while the teacher drives it; its
n-step returns (``BatchedRollout`` with V = 0) are the value targets.  Without some such start the +1 goal reward is never
found from random weights."""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from .network import NetworkVP_rnn
from .rollout import BatchedRollout


def find_action_index(actions: torch.Tensor, possible_actions: torch.Tensor) -> torch.Tensor:
    """Nearest discrete action in velocity space (Regression.py:164-176, the "complicated method"):
    actions [B, 2] (speed fraction, heading change) -> index into possible_actions [A, 2]."""
    ax, ay = actions[:, 0] * torch.cos(actions[:, 1]), actions[:, 0] * torch.sin(actions[:, 1])
    px, py = possible_actions[:, 0] * torch.cos(possible_actions[:, 1]), possible_actions[:, 0] * torch.sin(possible_actions[:, 1])
    d = (ax[:, None] - px[None, :]) ** 2 + (ay[:, None] - py[None, :]) ** 2
    return d.argmin(dim=1)


def teacher_actions(obs: torch.Tensor, possible_actions: torch.Tensor, max_turn: float = float(np.pi / 6)) -> torch.Tensor:
    """obs [W, N, 1+D] -> int32 [W, N]: go to the goal at full preferred speed.  obs column 3 is the agent's heading in
    the ego (goal-aligned) frame, so the heading change that points it at the goal is minus that, limited to one step's
    turn."""
    turn = (-obs[..., 3]).clamp(-max_turn, max_turn).reshape(-1)
    cont = torch.stack([torch.ones_like(turn), turn], dim=1)
    return find_action_index(cont, possible_actions).to(torch.int32).reshape(obs.shape[:-1])


def regression_loss(net: NetworkVP_rnn, x: torch.Tensor, y_r: torch.Tensor, a_index: torch.Tensor
                    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """cost_regression = cost_p_regression + cost_v (NetworkVPCore.py:90-100): sums over the batch."""
    logits, _, v = net.forward(x)
    cost_p = torch.nn.functional.cross_entropy(logits, a_index.long(), reduction="sum")
    cost_v = 0.5 * torch.sum((y_r.to(torch.float32) - v) ** 2)
    return cost_p + cost_v, cost_p, cost_v


def pretrain(net: NetworkVP_rnn, env, steps: int = 300, learning_rate: float = 1e-3, discount: float = 0.97,
             rows_per_step: int = 32768, log_every: int = 0) -> dict:
    """Drive ``env`` with the teacher, regress the network onto (observation -> teacher action, n-step return).
    Returns the last losses per row and the teacher's mean episode reward."""
    possible = torch.as_tensor(np.asarray(env.actions if hasattr(env, "actions") else _action_table()), dtype=torch.float32,
                               device=env.device)
    roll = BatchedRollout(env, policy=None, discount=discount, reflush_done=False)
    roll.reset()
    opt = torch.optim.Adam(net.parameters(), lr=learning_rate, eps=1e-8)
    zeros = torch.zeros((env.num_worlds, env.max_agents), dtype=torch.float32, device=env.device)
    done_steps, ep_reward, last = 0, [], (0.0, 0.0)
    while done_steps < steps:
        for _ in range(4):
            roll.step(teacher_actions(roll.obs, possible), zeros)
        b = roll.drain(provenance=False)
        e = roll.drain_episodes()
        if e.shape[0]:
            ep_reward.append(float(e[:, 1].mean()))
        for lo in range(0, len(b), rows_per_step):
            x, y, a = b.x[lo:lo + rows_per_step], b.r[lo:lo + rows_per_step], b.a_index[lo:lo + rows_per_step]
            opt.zero_grad(set_to_none=True)
            total, cost_p, cost_v = regression_loss(net, x, y, a)
            total.backward()
            opt.step()
            done_steps += 1
            last = (float(cost_p.detach()) / len(y), float(cost_v.detach()) / len(y))
            if log_every and done_steps % log_every == 0:
                print("[Regression] step %d  p-loss/row %.4f  v-loss/row %.5f  teacher episode reward %.3f"
                      % (done_steps, last[0], last[1], np.mean(ep_reward[-20:]) if ep_reward else float("nan")), flush=True)
            if done_steps >= steps:
                break
    roll.close()
    return {"p_loss_per_row": last[0], "v_loss_per_row": last[1],
            "teacher_episode_reward": float(np.mean(ep_reward[-50:])) if ep_reward else float("nan"), "steps": done_steps}


def _action_table():
    from ..actions import Actions
    return Actions().actions
