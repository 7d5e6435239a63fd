"""CPU ORACLE (test infrastructure, NOT product code) -- float64 restatement of the
multi-agent collision-avoidance ``env.step`` hot path.

    *** PARITY UNPINNED for the env half (E1..E9 of SURVEY.md section 8a). ***

The reference tree ``/root/reference`` ships the GA3C half only; the env package
``gym_collision_avoidance`` is an empty, un-vendored git submodule
(``/root/reference/.gitmodules:1-3``; pinned SHA unrecoverable, era ~Mar-Apr 2020 per
``ga3c/GA3C/Config.py:161-163``).  There is therefore no reference source, golden vector or
fixture this file can be checked against.  It restates the *published* algorithm of
``mit-acl/gym-collision-avoidance`` (Everett et al., arXiv:1805.01956 / 1910.11689) anchored on
the in-tree evidence listed per function below:

  * call sites ............ ga3c/GA3C/Environment.py:54-56,84-86,106,112 ; ProcessAgent.py:124-157
  * obs layout ............ ga3c/GA3C/Config.py:40-41,66-76 ; NetworkVP_rnn.py:58-61
  * scalar constants ...... ga3c/GA3C/checkpoints/regression/wandb/run-ws/config.yaml
                            (DT :43-45, NEAR_GOAL_THRESHOLD :124-126, MAX_TIME_RATIO :115-117,
                             COLLISION_DIST :31-33, GETTING_CLOSE_RANGE :64-66, REWARD_* :201-221,
                             AGENT_SORTING_METHOD :9-11, SENSING_HORIZON :249-251)
  * action table .......... ga3c/GA3C/Server.py:36,51-52 ; Regression.py:157-160 ; Config.py:79

Every semantic that could differ from the true pinned upstream is a named switch in
:class:`OracleConfig` (the "U" items of SURVEY.md Appendix A) so it can be flipped if the source
ever becomes available.

Style: deliberately the reference's style -- one ``World`` object, one Python ``Agent`` object
per agent, Python loops over agents and pairs, NumPy float64 scalars -- so that timing it is a
fair stand-in for "the reference Python/NumPy env" (baseline B1 of BASELINE.md).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product package must never import it.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# ----------------------------------------------------------------------------------------------
# flag bits (shared vocabulary with include/cavoid.h; values restated there, not imported)
# ----------------------------------------------------------------------------------------------
F_AT_GOAL = 1 << 0          # Agent.is_at_goal
F_RAN_OUT = 1 << 1          # Agent.ran_out_of_time
F_IN_COLL = 1 << 2          # Agent.in_collision
F_WAS_AT_GOAL = 1 << 3      # Agent.was_at_goal_already
F_WAS_IN_COLL = 1 << 4      # Agent.was_in_collision_already
F_PRESENT = 1 << 5          # row holds a real agent (worlds may have fewer than N agents)
F_LEARNING = 1 << 6         # policy is the external (GA3C) learning policy -> obs col 0
F_POLICY_SHIFT = 8          # bits 8..10: 0 external/learning, 1 static, 2 non-cooperative, 3 RVO (ORCA), 4 frozen network
F_POLICY_MASK = 7
F_DONE_MASK = F_AT_GOAL | F_RAN_OUT | F_IN_COLL

# POLICY_FROZEN_NET: a NON-learning agent driven by a frozen copy of the GA3C-CADRL network (ga3c/GA3C/Server.py:36 imports
# that policy class from the env package): the env decodes its action index from the same table as a learner's, the index
# itself comes from outside (the caller evaluates the frozen network), it never counts as "learning".
POLICY_EXTERNAL, POLICY_STATIC, POLICY_NONCOOP, POLICY_RVO, POLICY_FROZEN_NET = 0, 1, 2, 3, 4
SORT_CLOSEST_LAST, SORT_CLOSEST_FIRST, SORT_TIME_TO_IMPACT = 0, 1, 2
DYN_UNICYCLE, DYN_UNICYCLE_MAX_TURN, DYN_HOLONOMIC = 0, 1, 2


def build_action_table() -> np.ndarray:
    """E4 -- the 11-row ``[speed_fraction, delta_heading]`` table of the GA3C-CADRL policy.

    Evidence: ``Server.py:51-52`` (``Actions().num_actions``), ``Config.py:79`` (11),
    ``Regression.py:157-160`` (column 0 = speed, column 1 = heading change).
    Rows: 5 headings at full speed (step pi/12 from -pi/6), 3 at half speed and 3 at zero speed
    (step pi/6).  Values are ``start + k*step`` in float64, i.e. what a NumPy grid produces.
    """
    rows = []
    for frac, step, count in ((1.0, math.pi / 12, 5), (0.5, math.pi / 6, 3), (0.0, math.pi / 6, 3)):
        for k in range(count):
            rows.append((frac, -math.pi / 6 + k * step))
    return np.array(rows, dtype=np.float64)


@dataclass
class OracleConfig:
    # --- constants recorded in run-ws/config.yaml ---------------------------------------------
    dt: float = 0.2
    near_goal_threshold: float = 0.2
    max_time_ratio: float = 2.0
    collision_dist: float = 0.0
    getting_close_range: float = 0.2
    reward_at_goal: float = 1.0
    reward_collision: float = -0.25
    reward_getting_close: float = -0.1
    reward_time_step: float = 0.0
    sensing_horizon: float = math.inf
    sort_method: int = SORT_CLOSEST_LAST
    # --- sizes (Config.py:34-49) ---------------------------------------------------------------
    max_agents: int = 4                  # MAX_NUM_AGENTS_IN_ENVIRONMENT (N)
    max_other_agents_observed: int = 3   # MAX_NUM_OTHER_AGENTS_OBSERVED (M)
    # --- U-switches ------------------------------------------------------
    close_penalty_slope: float = 0.5     # U5: r = reward_getting_close + slope*gap  (+0.5: the paper, favoured by the reference's recorded scores; -0.5: upstream code as recalled)
    actions_fp32: bool = True            # joint action array is float32 in the env's _take_action
    timeout_enabled: bool = True         # U1
    dynamics: int = DYN_UNICYCLE         # U3
    max_turn_rate: float = 3.0           # rad/s, only DYN_UNICYCLE_MAX_TURN
    evaluate_mode: bool = False          # EVALUATE_MODE: the episode ends when EVERY agent is done
    time_budget_from_goal_edge: bool = True   # U11: budget = ratio*(dist - NEAR_GOAL_THRESHOLD)/pref (upstream agent.py as
                                         # recalled) vs SURVEY App. A's ratio*dist/pref (False)
    wrap_closed_end: bool = False        # U2: angles wrap to [-pi, pi) (False) or to (-pi, pi] (True)
    done_agents_collide: bool = True     # U4: an agent that is already done still takes part in the others' collision test and
                                         # nearest gap (True); False: a pair with an agent that was done before the step is skipped
    sort_round_gap: bool = True          # U7a: neighbours ordered by the gap rounded to centimetres (True) or by the exact gap
    sort_tie_lateral: bool = True        # U7b: equal (rounded) gaps ordered by the lateral offset, then index (True); index alone
    # --- RVO scripted policy (run-ws/config.yaml:231-239: RVO_TIME_HORIZON 5.0, RVO_COLLAB_COEFF 0.5) -----------
    rvo_time_horizon: float = 5.0
    rvo_collab_coeff: float = 0.5        # share of the avoidance effort the RVO agent takes (0.5 = reciprocal)
    rvo_radius_scale: float = 1.05       # the policy inflates every radius by 5 % (upstream RVOPolicy as recalled)
    rvo_max_delta_heading: float = math.pi / 6   # larger turns: stop and turn in place
    # min/max of the env's list of possible reward values; rewards are clipped into it
    reward_clip_lo: float = -0.25
    reward_clip_hi: float = 1.0
    actions: np.ndarray = field(default_factory=build_action_table)

    @property
    def obs_width(self) -> int:          # 1 + NN_INPUT_SIZE  (Config.py:66-71)
        return 2 + 4 + 7 * self.max_other_agents_observed


def wrap(angle: float, closed_end: bool = False) -> float:
    """Wrap by repeated +-2*pi to [-pi, pi) (U2 default: half-open at +pi) or, ``closed_end``, to (-pi, pi]."""
    if closed_end:
        while angle > math.pi:
            angle -= 2.0 * math.pi
        while angle <= -math.pi:
            angle += 2.0 * math.pi
        return angle
    while angle >= math.pi:
        angle -= 2.0 * math.pi
    while angle < -math.pi:
        angle += 2.0 * math.pi
    return angle


class Agent:
    """One agent's global-frame state (E5).  ``policy`` picks who chooses its action."""

    def __init__(self, px, py, gx, gy, radius, pref_speed, heading=None, policy=POLICY_EXTERNAL,
                 cfg: Optional[OracleConfig] = None):
        cfg = cfg or OracleConfig()
        self.cfg = cfg
        self.pos = np.array([px, py], dtype=np.float64)
        self.goal = np.array([gx, gy], dtype=np.float64)
        self.vel = np.zeros(2, dtype=np.float64)
        self.speed = 0.0
        self.radius = float(radius)
        self.pref_speed = float(pref_speed)
        if heading is None:
            to_goal = self.goal - self.pos
            heading = math.atan2(to_goal[1], to_goal[0])
        self.heading = float(heading)
        self.policy = policy
        # time budget: MAX_TIME_RATIO x straight-line time, never below one step (run-ws/config.yaml:115-117)
        dxg, dyg = float(px) - float(gx), float(py) - float(gy)
        offset = cfg.near_goal_threshold if cfg.time_budget_from_goal_edge else 0.0
        straight = (math.sqrt(dxg * dxg + dyg * dyg) - offset) / self.pref_speed
        self.t_remaining = max(cfg.max_time_ratio * straight, cfg.dt)
        self.is_at_goal = False
        self.was_at_goal_already = False
        self.in_collision = False
        self.was_in_collision_already = False
        self.ran_out_of_time = False
        self.num_other_agents_observed = 0
        self.update_ego_frame()

    # -- bookkeeping ---------------------------------------------------------------------------
    @property
    def is_learning(self) -> bool:
        return self.policy == POLICY_EXTERNAL

    @property
    def is_done(self) -> bool:
        return self.is_at_goal or self.ran_out_of_time or self.in_collision

    def flags(self) -> int:
        f = F_PRESENT
        f |= F_AT_GOAL if self.is_at_goal else 0
        f |= F_RAN_OUT if self.ran_out_of_time else 0
        f |= F_IN_COLL if self.in_collision else 0
        f |= F_WAS_AT_GOAL if self.was_at_goal_already else 0
        f |= F_WAS_IN_COLL if self.was_in_collision_already else 0
        f |= F_LEARNING if self.is_learning else 0
        f |= self.policy << F_POLICY_SHIFT
        return f

    def set_flags(self, f: int) -> None:
        self.is_at_goal = bool(f & F_AT_GOAL)
        self.ran_out_of_time = bool(f & F_RAN_OUT)
        self.in_collision = bool(f & F_IN_COLL)
        self.was_at_goal_already = bool(f & F_WAS_AT_GOAL)
        self.was_in_collision_already = bool(f & F_WAS_IN_COLL)
        self.policy = (f >> F_POLICY_SHIFT) & F_POLICY_MASK

    # -- ego frame (E9 host part) --------------------------------------------------------------
    def update_ego_frame(self) -> None:
        """x-axis of the ego frame points at the goal (Config.py:72-73 'dist to goal, heading to goal')."""
        to_goal = self.goal - self.pos
        self.dist_to_goal = math.sqrt(to_goal[0] * to_goal[0] + to_goal[1] * to_goal[1])
        if self.dist_to_goal > 1e-8:
            self.ref_prll = to_goal / self.dist_to_goal
        else:
            self.ref_prll = to_goal.copy()
        self.ref_orth = np.array([-self.ref_prll[1], self.ref_prll[0]])
        self.heading_ego = wrap(self.heading - math.atan2(self.ref_prll[1], self.ref_prll[0]), self.cfg.wrap_closed_end)

    # -- E5: one dynamics step -----------------------------------------------------------------
    def take_action(self, action: Sequence[float], dt: float) -> None:
        """``action`` = [speed, delta_heading] (unicycle) or [vx, vy] (holonomic)."""
        cfg = self.cfg
        if self.is_done:
            # frozen agents neither move nor spend time; latch the 'already' flags
            if self.is_at_goal:
                self.was_at_goal_already = True
            if self.in_collision:
                self.was_in_collision_already = True
            self.vel[:] = 0.0
            self.speed = 0.0
            return
        if cfg.dynamics == DYN_HOLONOMIC:
            vx, vy = float(action[0]), float(action[1])
            self.speed = math.sqrt(vx * vx + vy * vy)
            if self.speed > 0.0:
                self.heading = math.atan2(vy, vx)
            self.pos += np.array([vx * dt, vy * dt])
            self.vel[0], self.vel[1] = vx, vy
        else:
            speed = float(action[0])
            dh = float(action[1])
            if cfg.dynamics == DYN_UNICYCLE_MAX_TURN:
                rate = min(max(dh / dt, -cfg.max_turn_rate), cfg.max_turn_rate)
                dh = rate * dt
            new_heading = wrap(dh + self.heading, cfg.wrap_closed_end)
            c, s = math.cos(new_heading), math.sin(new_heading)
            self.pos += np.array([speed * c * dt, speed * s * dt])
            self.vel[0], self.vel[1] = speed * c, speed * s
            self.speed = speed
            self.heading = new_heading
        self.update_ego_frame()
        d = self.pos - self.goal
        self.is_at_goal = bool(d[0] * d[0] + d[1] * d[1] <= cfg.near_goal_threshold * cfg.near_goal_threshold)
        self.t_remaining -= dt
        if cfg.timeout_enabled and self.t_remaining <= 0.0:
            self.ran_out_of_time = True


def time_to_impact(host: Agent, other: Agent) -> float:
    """U8 -- first time the relative motion brings the two discs into contact (inf if never,
    0 if already overlapping).  Ray/disc first-hit of v_rel = v_host - v_other against the
    combined radius; stated here in closed form."""
    rx, ry = other.pos[0] - host.pos[0], other.pos[1] - host.pos[1]
    vx, vy = host.vel[0] - other.vel[0], host.vel[1] - other.vel[1]
    R = host.radius + other.radius
    c = rx * rx + ry * ry - R * R
    if c <= 0.0:
        return 0.0
    a = vx * vx + vy * vy
    b = rx * vx + ry * vy            # closing speed x distance
    if a < 1e-10 or b <= 0.0:
        return math.inf
    disc = b * b - a * c
    if disc < 0.0:
        return math.inf
    return (b - math.sqrt(disc)) / a


# ----------------------------------------------------------------------------------------------
# RVO scripted policy: Optimal Reciprocal Collision Avoidance, van den Berg et al.,
# "Reciprocal n-body collision avoidance" (ISRR 2009) -- the algorithm of the RVO2 library the upstream RVOPolicy
# drives through its Python binding (absent here, like the whole env package: PARITY UNPINNED).  Restated in float64
# from the paper: one half-plane ("ORCA line") per neighbour, then the 2-D linear programme of its section 5.2
# (closest admissible velocity to the preferred one inside the max-speed disc), and, when the half-planes admit no
# velocity, the programme that minimises the worst penetration.  Neighbours are visited in agent-index order
# (RVO2 visits them nearest first; the optimum does not depend on the order, rounding does).
# ----------------------------------------------------------------------------------------------
RVO_EPSILON = 1e-5


def _det(ax: float, ay: float, bx: float, by: float) -> float:
    return ax * by - ay * bx


def orca_lines(hi: int, agents: List["Agent"], cfg: "OracleConfig") -> List[Tuple[float, float, float, float]]:
    """One line (point_x, point_y, dir_x, dir_y) per OTHER agent: velocities to the left of the line are admissible."""
    host = agents[hi]
    inv_h = 1.0 / cfg.rvo_time_horizon
    lines = []
    for j, other in enumerate(agents):
        if j == hi:
            continue
        rpx, rpy = other.pos[0] - host.pos[0], other.pos[1] - host.pos[1]
        rvx, rvy = host.vel[0] - other.vel[0], host.vel[1] - other.vel[1]
        dist_sq = rpx * rpx + rpy * rpy
        comb = cfg.rvo_radius_scale * host.radius + cfg.rvo_radius_scale * other.radius
        comb_sq = comb * comb
        if dist_sq > comb_sq:                              # not touching: velocity obstacle truncated at the horizon
            wx, wy = rvx - inv_h * rpx, rvy - inv_h * rpy
            w_sq = wx * wx + wy * wy
            dot1 = wx * rpx + wy * rpy
            if dot1 < 0.0 and dot1 * dot1 > comb_sq * w_sq:   # closest point is on the cut-off circle
                w_len = math.sqrt(w_sq)
                ux, uy = wx / w_len, wy / w_len
                dx, dy = uy, -ux
                scale = comb * inv_h - w_len
                ucx, ucy = scale * ux, scale * uy
            else:                                           # ... on one of the two legs
                leg = math.sqrt(dist_sq - comb_sq)
                if _det(rpx, rpy, wx, wy) > 0.0:
                    dx, dy = (rpx * leg - rpy * comb) / dist_sq, (rpx * comb + rpy * leg) / dist_sq
                else:
                    dx, dy = -(rpx * leg + rpy * comb) / dist_sq, -(-rpx * comb + rpy * leg) / dist_sq
                dot2 = rvx * dx + rvy * dy
                ucx, ucy = dot2 * dx - rvx, dot2 * dy - rvy
        else:                                               # already overlapping: get out within one time step
            inv_dt = 1.0 / cfg.dt
            wx, wy = rvx - inv_dt * rpx, rvy - inv_dt * rpy
            w_len = math.sqrt(wx * wx + wy * wy)
            ux, uy = wx / w_len, wy / w_len
            dx, dy = uy, -ux
            scale = comb * inv_dt - w_len
            ucx, ucy = scale * ux, scale * uy
        lines.append((host.vel[0] + cfg.rvo_collab_coeff * ucx, host.vel[1] + cfg.rvo_collab_coeff * ucy, dx, dy))
    return lines


def _lp_on_line(lines, k: int, radius: float, ox: float, oy: float, direction_opt: bool):
    """Optimise along line k inside the disc and the half-planes 0..k-1.  -> (ok, x, y)"""
    px, py, dx, dy = lines[k]
    dot = px * dx + py * dy
    disc = dot * dot + radius * radius - (px * px + py * py)
    if disc < 0.0:
        return False, 0.0, 0.0
    root = math.sqrt(disc)
    t_lo, t_hi = -dot - root, -dot + root
    for i in range(k):
        qx, qy, ex, ey = lines[i]
        den = _det(dx, dy, ex, ey)
        num = _det(ex, ey, px - qx, py - qy)
        if abs(den) <= RVO_EPSILON:                        # parallel lines
            if num < 0.0:
                return False, 0.0, 0.0
            continue
        t = num / den
        if den >= 0.0:
            t_hi = min(t_hi, t)
        else:
            t_lo = max(t_lo, t)
        if t_lo > t_hi:
            return False, 0.0, 0.0
    if direction_opt:
        t = t_hi if ox * dx + oy * dy > 0.0 else t_lo
    else:
        t = dx * (ox - px) + dy * (oy - py)
        t = t_lo if t < t_lo else (t_hi if t > t_hi else t)
    return True, px + t * dx, py + t * dy


def _lp_plane(lines, radius: float, ox: float, oy: float, direction_opt: bool):
    """-> (index of the first line that cannot be satisfied, or len(lines); x; y)"""
    if direction_opt:
        x, y = ox * radius, oy * radius
    elif ox * ox + oy * oy > radius * radius:
        n = math.sqrt(ox * ox + oy * oy)
        x, y = ox / n * radius, oy / n * radius
    else:
        x, y = ox, oy
    for k, (px, py, dx, dy) in enumerate(lines):
        if _det(dx, dy, px - x, py - y) > 0.0:             # current optimum violates half-plane k
            ok, nx, ny = _lp_on_line(lines, k, radius, ox, oy, direction_opt)
            if not ok:
                return k, x, y
            x, y = nx, ny
    return len(lines), x, y


def _lp_least_penetration(lines, begin: int, radius: float, x: float, y: float):
    distance = 0.0
    for k in range(begin, len(lines)):
        px, py, dx, dy = lines[k]
        if _det(dx, dy, px - x, py - y) > distance:
            proj = []
            for j in range(k):
                qx, qy, ex, ey = lines[j]
                den = _det(dx, dy, ex, ey)
                if abs(den) <= RVO_EPSILON:
                    if dx * ex + dy * ey > 0.0:             # same direction: line j adds nothing
                        continue
                    nx, ny = 0.5 * (px + qx), 0.5 * (py + qy)
                else:
                    t = _det(ex, ey, px - qx, py - qy) / den
                    nx, ny = px + t * dx, py + t * dy
                fx, fy = ex - dx, ey - dy
                fn = math.sqrt(fx * fx + fy * fy)
                proj.append((nx, ny, fx / fn, fy / fn))
            fail, nx, ny = _lp_plane(proj, radius, -dy, dx, True)
            if fail >= len(proj):
                x, y = nx, ny
            distance = _det(dx, dy, px - x, py - y)
    return x, y


def rvo_action(hi: int, agents: List["Agent"], cfg: "OracleConfig") -> np.ndarray:
    """[speed, delta_heading] of an RVO agent: the ORCA velocity for one DT, turned into the unicycle action the env
    integrates; a turn beyond rvo_max_delta_heading is clipped and taken standing still."""
    host = agents[hi]
    gx, gy = host.goal[0] - host.pos[0], host.goal[1] - host.pos[1]
    gn = math.sqrt(gx * gx + gy * gy)
    scale = host.pref_speed / gn if gn > 0.0 else 0.0
    pvx, pvy = scale * gx, scale * gy
    lines = orca_lines(hi, agents, cfg)
    fail, vx, vy = _lp_plane(lines, host.pref_speed, pvx, pvy, False)
    if fail < len(lines):
        vx, vy = _lp_least_penetration(lines, fail, host.pref_speed, vx, vy)
    speed = math.sqrt(vx * vx + vy * vy)
    delta = wrap(math.atan2(vy, vx) - host.heading, cfg.wrap_closed_end) if speed > 0.0 else 0.0
    if abs(delta) > cfg.rvo_max_delta_heading:
        delta = math.copysign(cfg.rvo_max_delta_heading, delta)
        speed = 0.0
    return np.array([speed, delta])


class World:
    """One simulated world = what one reference ``CollisionAvoidanceEnv`` instance holds."""

    def __init__(self, agents: List[Agent], cfg: Optional[OracleConfig] = None):
        self.cfg = cfg or (agents[0].cfg if agents else OracleConfig())
        self.agents = agents
        self._frozen = [False] * len(agents)
        assert len(agents) <= self.cfg.max_agents

    # -- E4: decode ----------------------------------------------------------------------------
    def _decode(self, agent: Agent, action_index: int) -> np.ndarray:
        raw = self.cfg.actions[int(action_index)]
        return np.array([agent.pref_speed * raw[0], raw[1]])

    def _policy_action(self, agent: Agent) -> np.ndarray:
        if agent.policy == POLICY_STATIC:
            return np.array([0.0, 0.0])
        if agent.policy == POLICY_RVO:
            return rvo_action(self.agents.index(agent), self.agents, self.cfg)
        # non-cooperative: full preferred speed straight at the goal
        return np.array([agent.pref_speed, -agent.heading_ego])

    # -- E3: the whole step ----------------------------------------------------------------------
    def step(self, actions, continuous: bool = False):
        """``actions``: dict/sequence ``agent index -> action index`` for the learning agents
        (ProcessAgent.py:124,144,149), or with ``continuous=True`` ``agent index -> (a0, a1)``.

        Returns ``(obs[N, 1+D] f64, rewards[n] f64, game_over bool, info)`` where ``info`` has
        the two dicts ProcessAgent.py:155-157 reads."""
        cfg = self.cfg
        n = len(self.agents)
        dtype = np.float32 if cfg.actions_fp32 else np.float64
        joint = np.zeros((n, 2), dtype=dtype)
        self._frozen = [ag.is_done for ag in self.agents]      # done BEFORE this step's move (U4)
        for i, ag in enumerate(self.agents):
            if ag.is_done:
                continue
            if ag.policy in (POLICY_EXTERNAL, POLICY_FROZEN_NET):
                a = actions[i]
                joint[i, :] = np.asarray(a, dtype=np.float64) if continuous else self._decode(ag, a)
            else:
                joint[i, :] = self._policy_action(ag)
        for i, ag in enumerate(self.agents):
            ag.take_action(joint[i, :], cfg.dt)
        rewards = self._compute_rewards()
        obs = self.observe()
        done = {i: ag.is_done for i, ag in enumerate(self.agents)}
        learning = {i: ag.is_learning for i, ag in enumerate(self.agents)}
        learners = [ag.is_done for ag in self.agents if ag.is_learning or cfg.evaluate_mode]
        game_over = bool(np.all(learners))       # TRAIN_MODE: every *learning* agent done (EVALUATE_MODE: every agent)
        return obs, rewards, game_over, {"which_agents_done": done, "which_agents_learning": learning}

    # -- E6: pairwise gaps / collisions ----------------------------------------------------------
    def _check_for_collisions(self) -> Tuple[List[bool], List[float]]:
        n = len(self.agents)
        hit = [False] * n
        min_gap = [math.inf] * n
        for i in range(n):
            for j in range(i + 1, n):
                a, b = self.agents[i], self.agents[j]
                if not self.cfg.done_agents_collide and (self._frozen[i] or self._frozen[j]):
                    continue                                   # U4 flipped: frozen agents are out of the collision check
                dx, dy = a.pos[0] - b.pos[0], a.pos[1] - b.pos[1]
                d = math.sqrt(dx * dx + dy * dy)
                gap = d - (a.radius + b.radius)
                min_gap[i] = min(min_gap[i], gap)
                min_gap[j] = min(min_gap[j], gap)
                if gap <= self.cfg.collision_dist:
                    hit[i] = hit[j] = True
        return hit, min_gap

    # -- E7: rewards (+ in_collision latch) ------------------------------------------------------
    def _compute_rewards(self) -> np.ndarray:
        cfg = self.cfg
        hit, min_gap = self._check_for_collisions()
        rewards = cfg.reward_time_step * np.ones(len(self.agents))
        for i, ag in enumerate(self.agents):
            if ag.is_at_goal:
                if not ag.was_at_goal_already:
                    rewards[i] = cfg.reward_at_goal          # paid once
            elif not ag.was_in_collision_already:
                if hit[i]:
                    rewards[i] = cfg.reward_collision
                    ag.in_collision = True
                elif min_gap[i] <= cfg.getting_close_range:
                    rewards[i] = cfg.reward_getting_close + cfg.close_penalty_slope * min_gap[i]
        return np.clip(rewards, cfg.reward_clip_lo, cfg.reward_clip_hi)

    # -- E9: sensing + observation assembly ------------------------------------------------------
    def _sense_others(self, hi: int) -> np.ndarray:
        cfg = self.cfg
        host = self.agents[hi]
        M = cfg.max_other_agents_observed
        crit = []
        for j, other in enumerate(self.agents):
            if j == hi:
                continue
            rel = other.pos - host.pos
            d = math.sqrt(rel[0] * rel[0] + rel[1] * rel[1])
            if d > cfg.sensing_horizon:
                continue
            gap = d - host.radius - other.radius
            p_orth = rel[0] * host.ref_orth[0] + rel[1] * host.ref_orth[1]
            tti = time_to_impact(host, other) if cfg.sort_method == SORT_TIME_TO_IMPACT else 0.0
            # U7a: gap rounded to centimetres (or the exact gap); U7b: lateral offset breaks ties (or nothing: index order)
            crit.append((j, np.rint(gap * 100.0) / 100.0 if cfg.sort_round_gap else gap,
                         p_orth if cfg.sort_tie_lateral else 0.0, tti))
        if cfg.sort_method == SORT_TIME_TO_IMPACT:
            far_to_near = sorted(crit, key=lambda c: (-c[3], -c[1], c[2]))
        else:
            far_to_near = sorted(crit, key=lambda c: (-c[1], c[2]))
        kept = far_to_near[-M:] if M > 0 else []
        if cfg.sort_method == SORT_CLOSEST_FIRST:
            kept = sorted(kept, key=lambda c: (c[1], c[2]))
        out = np.zeros((M, 7), dtype=np.float64)
        for slot, (j, _, _, _) in enumerate(kept):
            other = self.agents[j]
            rel = other.pos - host.pos
            p_prll = rel[0] * host.ref_prll[0] + rel[1] * host.ref_prll[1]
            p_orth = rel[0] * host.ref_orth[0] + rel[1] * host.ref_orth[1]
            v_prll = other.vel[0] * host.ref_prll[0] + other.vel[1] * host.ref_prll[1]
            v_orth = other.vel[0] * host.ref_orth[0] + other.vel[1] * host.ref_orth[1]
            gap = math.sqrt(rel[0] * rel[0] + rel[1] * rel[1]) - host.radius - other.radius
            out[slot] = (p_prll, p_orth, v_prll, v_orth, other.radius, host.radius + other.radius, gap)
        host.num_other_agents_observed = len(kept)
        return out

    def observe(self) -> np.ndarray:
        """``[N_max, 1+D]``; rows of absent agents are zero so ``is_learning == 0``
        (Environment.py:84-86, ProcessAgent.py:130-133)."""
        cfg = self.cfg
        obs = np.zeros((cfg.max_agents, cfg.obs_width), dtype=np.float64)
        for i, ag in enumerate(self.agents):
            ag.update_ego_frame()
            others = self._sense_others(i)
            obs[i, 0] = 1.0 if ag.is_learning else 0.0
            obs[i, 1] = ag.num_other_agents_observed
            obs[i, 2] = ag.dist_to_goal
            obs[i, 3] = ag.heading_ego
            obs[i, 4] = ag.pref_speed
            obs[i, 5] = ag.radius
            obs[i, 6:] = others.reshape(-1)
        return obs


# ----------------------------------------------------------------------------------------------
# Scenario generator "GEN v1" (E2).  The upstream random test-case generator and its np.random
# stream are unknowable here, so the build defines its own counter-based
# generator; this is its specification.  Philox4x32-10 (Salmon et al., SC'11), counter =
# (global world id, episode index, stream, agent index), key = 64-bit seed.
# ----------------------------------------------------------------------------------------------
_PHILOX_M0, _PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
_PHILOX_W0, _PHILOX_W1 = 0x9E3779B9, 0xBB67AE85
_MASK32 = 0xFFFFFFFF


def philox4x32(c0: int, c1: int, c2: int, c3: int, k0: int, k1: int) -> Tuple[int, int, int, int]:
    for _ in range(10):
        p0 = _PHILOX_M0 * c0
        p1 = _PHILOX_M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & _MASK32, p1 & _MASK32, ((p0 >> 32) ^ c3 ^ k1) & _MASK32, p0 & _MASK32
        k0 = (k0 + _PHILOX_W0) & _MASK32
        k1 = (k1 + _PHILOX_W1) & _MASK32
    return c0, c1, c2, c3


def _u01(r: int) -> float:
    """24-bit uniform in [0,1): exact in float32 and float64."""
    return (r >> 8) * (1.0 / 16777216.0)


@dataclass
class GenConfig:
    min_agents: int = 4
    max_agents: int = 4
    nonlearning_fraction: float = 0.0     # P(agent i>0 runs a scripted policy)
    static_fraction: float = 0.5          # of those, P(static)
    goal_jitter: float = 0.5              # GEN v1: half-width (m) of the uniform jitter on the antipodal goal
    angle_jitter: float = 0.25            # GEN v1: fraction of the angular slot
    pool_size: int = 0                    # > 0: scenario pool (episode ep of world gw = pool entry pool_index(seed, gw, ep, P))
    mode: int = 0                         # 0 = GEN v1 (ring, antipodal goals), 1 = GEN v2 (uniform boxes, rejection sampling)
    rvo_fraction: float = 0.0             # of the scripted agents, P(RVO)
    frozen_fraction: float = 0.0          # ... P(frozen network); the rest (1 - static - rvo - frozen) are non-cooperative
    box_small: Tuple[float, float] = (4.0, 5.0)    # GEN v2: half side of the box ~ U(lo, hi) for worlds of < box_large_from agents
    box_large: Tuple[float, float] = (6.0, 8.0)    # ... and for the larger worlds (keeps the density roughly constant)
    box_large_from: int = 5
    min_trip: float = 1.0                 # GEN v2: an agent's goal is at least this far from its start
    pool_epoch: int = 0                   # the pool holds generator worlds 0..P-1 of THIS episode index (refreshable)


GEN_V2_MAX_ATTEMPTS = 100


def pool_index(seed: int, world_id: int, episode: int, pool_size: int) -> int:
    """splitmix64-style finaliser of (seed, global world id, episode), reduced to [0, P) by multiply-shift."""
    m64 = 0xFFFFFFFFFFFFFFFF
    z = (seed + 0x9E3779B97F4A7C15 * ((world_id & _MASK32) + 1) + 0xC2B2AE3D27D4EB4F * ((episode & _MASK32) + 1)) & m64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m64
    z ^= z >> 31
    return ((z >> 32) * pool_size) >> 32


def _draw_policy(b, i: int, gen: GenConfig) -> int:
    if i > 0 and _u01(b[2]) < gen.nonlearning_fraction:
        u = _u01(b[3])
        if u < gen.static_fraction:
            return POLICY_STATIC
        if u < gen.static_fraction + gen.rvo_fraction:
            return POLICY_RVO
        return POLICY_FROZEN_NET if u < gen.static_fraction + gen.rvo_fraction + gen.frozen_fraction else POLICY_NONCOOP
    return POLICY_EXTERNAL


def generate_world(seed: int, world_id: int, episode: int, cfg: OracleConfig, gen: GenConfig) -> World:
    """GEN v1 (mode 0): n agents on a circle, goals roughly antipodal (every pair must negotiate the centre).
    GEN v2 (mode 1): starts and goals uniform in a box whose size grows with the agent count, placed one agent after the
    other by rejection sampling against the agents already placed (starts and goals at least the two radii +
    GETTING_CLOSE_RANGE apart, trips of at least min_trip) -- the shape of upstream's random test-case generator as
    recalled (its np.random stream is unknowable, so this is its own counter-based specification).
    Both: radius~U(0.2,0.8), pref_speed~U(0.5,2.0) stored as float32 values, goals float32, starts float64; heading points
    at the goal; time budget as in ``Agent``."""
    k0, k1 = seed & _MASK32, (seed >> 32) & _MASK32
    if gen.pool_size > 0:      # pool entry k is generator world k of the pool's epoch
        world_id = pool_index(seed, world_id, episode, gen.pool_size)
        episode = gen.pool_epoch
    wid, ep = world_id & _MASK32, episode & _MASK32
    w = philox4x32(wid, ep, 0, 0, k0, k1)
    span = gen.max_agents - gen.min_agents + 1
    n = gen.min_agents + (w[0] % span)
    agents = []
    if gen.mode == 0:
        base = max(4.0, 0.7 * n)
        ring = base * (1.0 + _u01(w[1]))
        phase = _u01(w[2])
        for i in range(n):
            a = philox4x32(wid, ep, 1, i, k0, k1)
            b = philox4x32(wid, ep, 2, i, k0, k1)
            radius = float(np.float32(0.2 + 0.6 * _u01(a[0])))
            pref_speed = float(np.float32(0.5 + 1.5 * _u01(a[1])))
            turn = phase + (i + (_u01(a[2]) - 0.5) * 2.0 * gen.angle_jitter) / n
            theta = 2.0 * math.pi * turn
            px, py = ring * math.cos(theta), ring * math.sin(theta)
            gx = float(np.float32(-px + (_u01(b[0]) - 0.5) * 2.0 * gen.goal_jitter))
            gy = float(np.float32(-py + (_u01(b[1]) - 0.5) * 2.0 * gen.goal_jitter))
            agents.append(Agent(px, py, gx, gy, radius, pref_speed, None, _draw_policy(b, i, gen), cfg))
        return World(agents, cfg)
    lo, hi = gen.box_small if n < gen.box_large_from else gen.box_large
    side = lo + (hi - lo) * _u01(w[1])
    for i in range(n):
        a = philox4x32(wid, ep, 1, i, k0, k1)
        b = philox4x32(wid, ep, 2, i, k0, k1)
        radius = float(np.float32(0.2 + 0.6 * _u01(a[0])))
        pref_speed = float(np.float32(0.5 + 1.5 * _u01(a[1])))
        attempt = 0
        while True:
            c = philox4x32(wid, ep, 3 + attempt, i, k0, k1)
            sx, sy = side * (2.0 * _u01(c[0]) - 1.0), side * (2.0 * _u01(c[1]) - 1.0)
            gx = float(np.float32(side * (2.0 * _u01(c[2]) - 1.0)))
            gy = float(np.float32(side * (2.0 * _u01(c[3]) - 1.0)))
            tx, ty = gx - sx, gy - sy
            ok = math.sqrt(tx * tx + ty * ty) >= gen.min_trip
            for other in agents:
                margin = (radius + other.radius) + cfg.getting_close_range
                ax, ay = sx - other.pos[0], sy - other.pos[1]
                bx, by = gx - other.goal[0], gy - other.goal[1]
                if math.sqrt(ax * ax + ay * ay) < margin or math.sqrt(bx * bx + by * by) < margin:
                    ok = False
            attempt += 1
            if ok or attempt >= GEN_V2_MAX_ATTEMPTS:
                break
            if attempt % 10 == 0:
                side = side * 1.01                           # a crowded box grows, for this and the later agents
        agents.append(Agent(sx, sy, gx, gy, radius, pref_speed, None, _draw_policy(b, i, gen), cfg))
    return World(agents, cfg)


# ----------------------------------------------------------------------------------------------
# flat-array helpers: the same SoA record the C oracle and the HIP library exchange
# ----------------------------------------------------------------------------------------------
STATE_F64_FIELDS = ("px", "py", "heading", "t_remaining")
STATE_F32_FIELDS = ("gx", "gy", "radius", "pref_speed", "speed")


def world_to_arrays(world: World):
    N = world.cfg.max_agents
    f64 = np.zeros((4, N), dtype=np.float64)
    f32 = np.zeros((5, N), dtype=np.float32)
    flags = np.zeros(N, dtype=np.uint32)
    for i, ag in enumerate(world.agents):
        f64[:, i] = (ag.pos[0], ag.pos[1], ag.heading, ag.t_remaining)
        f32[:, i] = (ag.goal[0], ag.goal[1], ag.radius, ag.pref_speed, ag.speed)
        flags[i] = ag.flags()
    return f64, f32, flags


def world_from_arrays(f64, f32, flags, cfg: OracleConfig) -> World:
    agents = []
    for i in range(cfg.max_agents):
        f = int(flags[i])
        if not f & F_PRESENT:
            break
        ag = Agent(f64[0, i], f64[1, i], float(f32[0, i]), float(f32[1, i]), float(f32[2, i]),
                   float(f32[3, i]), float(f64[2, i]), (f >> F_POLICY_SHIFT) & F_POLICY_MASK, cfg)
        ag.t_remaining = float(f64[3, i])
        ag.set_flags(f)
        ag.speed = float(f32[4, i])
        ag.vel[:] = (ag.speed * math.cos(ag.heading), ag.speed * math.sin(ag.heading))
        agents.append(ag)
    return World(agents, cfg)
