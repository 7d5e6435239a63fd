"""Which reward algebra can have produced a recorded GA3C episode score?  (test infrastructure, like everything under oracle/)

The reference's episode score (ProcessStats.py:98-109) is ProcessAgent.run's `total_reward` (ProcessAgent.py:230-243): the sum, over
every flush of every learning agent, of that agent's rewards since its last flush divided by the number of learning agents
(`reward_sum_logger[i] / num_agents_running_ga3c`, ProcessAgent.py:168,195).  With n learning agents throughout, and the env's reward
(oracle/cavoid_oracle.py `_compute_rewards`): +reward_at_goal once, reward_collision once, and on every other step
with the nearest gap d in (0, getting_close_range] the "getting close" term  reward_getting_close + slope * d  (U5: slope -0.5 as
recalled from the upstream code, +0.5 in the paper, arXiv:1805.01956),

    n * score = g * reward_at_goal + c * reward_collision + T,      g + c <= n,  T = a sum of k >= 0 getting-close terms.

One getting-close term lies in [lo, hi]; the sum of k of them in [k*lo, k*hi].  `explanations` enumerates (n, g, c, k) that reproduce a
printed score (4 decimals: +-0.00005) and `fewest_close_steps` is the most parsimonious one per sign of the slope."""
from dataclasses import dataclass
from typing import List, Optional

PRINT_HALF_ULP = 0.5e-4


@dataclass(frozen=True)
class RewardAlgebra:
    reward_at_goal: float = 1.0
    reward_collision: float = -0.25
    reward_getting_close: float = -0.1
    getting_close_range: float = 0.2
    close_penalty_slope: float = 0.5

    def close_term_range(self):
        """[lo, hi] of one getting-close term over gaps d in (0, range] (d = 0 is a collision: the end at d -> 0 is open)"""
        a, b = self.reward_getting_close, self.reward_getting_close + self.close_penalty_slope * self.getting_close_range
        return (min(a, b), max(a, b))


@dataclass(frozen=True)
class Explanation:
    n: int          # learning agents
    goals: int
    collisions: int
    close_steps: int
    close_sum: float


def explanations(score: float, alg: RewardAlgebra, max_agents: int = 4, max_close_steps: int = 60) -> List[Explanation]:
    lo, hi = alg.close_term_range()
    out = []
    for n in range(1, max_agents + 1):
        tol = PRINT_HALF_ULP * n
        for g in range(n + 1):
            for c in range(n + 1 - g):
                T = n * score - g * alg.reward_at_goal - c * alg.reward_collision
                if abs(T) <= tol:
                    out.append(Explanation(n, g, c, 0, 0.0))
                    continue
                for k in range(1, max_close_steps + 1):
                    if k * lo - tol <= T <= k * hi + tol:
                        out.append(Explanation(n, g, c, k, T))
                        break                                   # (the fewest steps for this (n, g, c))
    return out


def fewest_close_steps(score: float, alg: RewardAlgebra, max_agents: int = 4) -> Optional[Explanation]:
    ex = explanations(score, alg, max_agents)
    return min(ex, key=lambda e: (e.close_steps, e.n)) if ex else None
