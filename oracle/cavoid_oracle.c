/* CPU ORACLE (test infrastructure, NOT product code).
 *
 *   *** PARITY UNPINNED for the env half. ***
 *
 * Batched float64 restatement, in plain C + libm, of the multi-agent collision-avoidance
 * env.step.  The env package (mit-acl/gym-collision-avoidance, pinned version unknown) is an
 * empty un-vendored submodule of the reference (/root/reference/.gitmodules:1-3), so no
 * reference source can be followed or compiled.  What is followed instead:
 *   - the call-site contract ........ ga3c/GA3C/Environment.py:84-86,106,112; ProcessAgent.py:124-157
 *   - the observation layout ........ ga3c/GA3C/Config.py:40-41,66-76; NetworkVP_rnn.py:58-61
 *   - the scalar constants .......... ga3c/GA3C/checkpoints/regression/wandb/run-ws/config.yaml
 *   - the published algorithm ....... arXiv:1805.01956, arXiv:1910.11689 (README.md:3-7)
 * It is arithmetic-for-arithmetic the same algorithm as oracle/cavoid_oracle.py (the
 * reference-style per-object statement); tests/test_oracle.py holds the two against each other.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA contraction, NumPy does not fuse).
 */
#include "cavoid_oracle.h"

#include <math.h>
#include <string.h>

enum {
    F_AT_GOAL = 1u << 0, F_RAN_OUT = 1u << 1, F_IN_COLL = 1u << 2, F_WAS_AT_GOAL = 1u << 3,
    F_WAS_IN_COLL = 1u << 4, F_PRESENT = 1u << 5, F_LEARNING = 1u << 6, F_POLICY_SHIFT = 8
};
#define F_DONE_MASK (F_AT_GOAL | F_RAN_OUT | F_IN_COLL)
enum { POLICY_EXTERNAL = 0, POLICY_STATIC = 1, POLICY_NONCOOP = 2, POLICY_RVO = 3, POLICY_FROZEN_NET = 4 };
#define F_POLICY_MASK 7
enum { SORT_CLOSEST_LAST = 0, SORT_CLOSEST_FIRST = 1, SORT_TTI = 2 };
enum { DYN_UNICYCLE = 0, DYN_UNICYCLE_MAX_TURN = 1, DYN_HOLONOMIC = 2 };

static const double PI = 3.14159265358979323846;

void oracle_default_cfg(oracle_cfg *c, int32_t max_agents, int32_t max_other) {
    memset(c, 0, sizeof(*c));
    /* run-ws/config.yaml: DT :43-45, NEAR_GOAL_THRESHOLD :124-126, MAX_TIME_RATIO :115-117,
     * COLLISION_DIST :31-33, GETTING_CLOSE_RANGE :64-66, REWARD_* :201-221 */
    c->dt = 0.2; c->near_goal_threshold = 0.2; c->max_time_ratio = 2.0; c->collision_dist = 0.0;
    c->getting_close_range = 0.2; c->reward_at_goal = 1.0; c->reward_collision = -0.25;
    c->reward_getting_close = -0.1; c->reward_time_step = 0.0; c->sensing_horizon = INFINITY;
    c->close_penalty_slope = 0.5; c->max_turn_rate = 3.0; c->reward_clip_lo = -0.25; c->reward_clip_hi = 1.0;
    /* RVO_TIME_HORIZON :237-239, RVO_COLLAB_COEFF :234-236; radius inflation and turn limit: upstream RVOPolicy as recalled */
    c->rvo_time_horizon = 5.0; c->rvo_collab_coeff = 0.5; c->rvo_radius_scale = 1.05; c->rvo_max_delta_heading = PI / 6;
    c->max_agents = max_agents; c->max_other = max_other; c->sort_method = SORT_CLOSEST_LAST;
    c->actions_fp32 = 1; c->timeout_enabled = 1; c->time_budget_from_goal_edge = 1; c->dynamics = DYN_UNICYCLE; c->num_actions = 11;
    c->wrap_closed_end = 0; c->done_agents_collide = 1; c->sort_round_gap = 1; c->sort_tie_lateral = 1;   /* U2, U4, U7a, U7b */
    /* E4: 5 headings at full speed (step pi/12), 3 at half speed, 3 at zero speed (step pi/6) */
    const double fr[3] = {1.0, 0.5, 0.0}, st[3] = {PI / 12, PI / 6, PI / 6};
    const int cnt[3] = {5, 3, 3};
    int r = 0;
    for (int g = 0; g < 3; ++g)
        for (int k = 0; k < cnt[g]; ++k, ++r) { c->actions[r][0] = fr[g]; c->actions[r][1] = -PI / 6 + k * st[g]; }
}

/* U2: [-pi, pi) by default, (-pi, pi] with wrap_closed_end */
static double wrap(const oracle_cfg *c, double a) {
    if (c->wrap_closed_end) {
        while (a > PI) a -= 2.0 * PI;
        while (a <= -PI) a += 2.0 * PI;
        return a;
    }
    while (a >= PI) a -= 2.0 * PI;
    while (a < -PI) a += 2.0 * PI;
    return a;
}

/* per-world scratch: one record per agent, the fields an Agent object would carry */
typedef struct {
    double px, py, gx, gy, vx, vy, heading, t_rem, radius, pref_speed, speed;
    double dist_to_goal, prll_x, prll_y, orth_x, orth_y, heading_ego;
    uint32_t flags;
} agent_t;

static void ego_frame(const oracle_cfg *c, agent_t *a) {
    double tx = a->gx - a->px, ty = a->gy - a->py;
    a->dist_to_goal = sqrt(tx * tx + ty * ty);
    if (a->dist_to_goal > 1e-8) { a->prll_x = tx / a->dist_to_goal; a->prll_y = ty / a->dist_to_goal; }
    else { a->prll_x = tx; a->prll_y = ty; }
    a->orth_x = -a->prll_y; a->orth_y = a->prll_x;
    a->heading_ego = wrap(c, a->heading - atan2(a->prll_y, a->prll_x));
}

static int load_world(const oracle_cfg *c, const oracle_state *s, int64_t w, agent_t *ag) {
    int n = 0;
    for (int i = 0; i < c->max_agents; ++i) {
        int64_t a = w * c->max_agents + i;
        if (!(s->flags[a] & F_PRESENT)) break;
        agent_t *g = &ag[n++];
        g->px = s->px[a]; g->py = s->py[a]; g->heading = s->heading[a]; g->t_rem = s->t_remaining[a];
        g->gx = s->gx[a]; g->gy = s->gy[a]; g->radius = s->radius[a]; g->pref_speed = s->pref_speed[a];
        g->speed = s->speed[a]; g->flags = s->flags[a];
        g->vx = g->speed * cos(g->heading); g->vy = g->speed * sin(g->heading);
        ego_frame(c, g);
    }
    return n;
}

static void store_world(const oracle_cfg *c, oracle_state *s, int64_t w, const agent_t *ag, int n) {
    for (int i = 0; i < n; ++i) {
        int64_t a = w * c->max_agents + i;
        s->px[a] = ag[i].px; s->py[a] = ag[i].py; s->heading[a] = ag[i].heading; s->t_remaining[a] = ag[i].t_rem;
        s->speed[a] = (float)ag[i].speed; s->flags[a] = ag[i].flags;
    }
}

/* E5 */
static void take_action(const oracle_cfg *c, agent_t *a, double a0, double a1) {
    const double dt = c->dt;
    if (a->flags & F_DONE_MASK) {
        if (a->flags & F_AT_GOAL) a->flags |= F_WAS_AT_GOAL;
        if (a->flags & F_IN_COLL) a->flags |= F_WAS_IN_COLL;
        a->vx = a->vy = 0.0; a->speed = 0.0;
        return;
    }
    if (c->dynamics == DYN_HOLONOMIC) {
        a->speed = sqrt(a0 * a0 + a1 * a1);
        if (a->speed > 0.0) a->heading = atan2(a1, a0);
        a->px += a0 * dt; a->py += a1 * dt; a->vx = a0; a->vy = a1;
    } else {
        double dh = a1;
        if (c->dynamics == DYN_UNICYCLE_MAX_TURN) {
            double rate = fmin(fmax(dh / dt, -c->max_turn_rate), c->max_turn_rate);
            dh = rate * dt;
        }
        double h = wrap(c, dh + a->heading), cs = cos(h), sn = sin(h);
        a->px += a0 * cs * dt; a->py += a0 * sn * dt;
        a->vx = a0 * cs; a->vy = a0 * sn; a->speed = a0; a->heading = h;
    }
    ego_frame(c, a);
    double dx = a->px - a->gx, dy = a->py - a->gy;
    if (dx * dx + dy * dy <= c->near_goal_threshold * c->near_goal_threshold) a->flags |= F_AT_GOAL;
    a->t_rem -= dt;
    if (c->timeout_enabled && a->t_rem <= 0.0) a->flags |= F_RAN_OUT;
}

static double time_to_impact(const agent_t *h, const agent_t *o) {
    double rx = o->px - h->px, ry = o->py - h->py, vx = h->vx - o->vx, vy = h->vy - o->vy;
    double R = h->radius + o->radius, cc = rx * rx + ry * ry - R * R;
    if (cc <= 0.0) return 0.0;
    double aa = vx * vx + vy * vy, bb = rx * vx + ry * vy;
    if (aa < 1e-10 || bb <= 0.0) return INFINITY;
    double disc = bb * bb - aa * cc;
    if (disc < 0.0) return INFINITY;
    return (bb - sqrt(disc)) / aa;
}

typedef struct { int j; double k0, k1, k2; } crit_t;

/* stable insertion sort, ascending lexicographic (k0,k1,k2) -- what a stable sort on a
 * tuple key does */
static void stable_sort(crit_t *v, int n) {
    for (int i = 1; i < n; ++i) {
        crit_t x = v[i];
        int p = i - 1;
        while (p >= 0) {
            const crit_t *y = &v[p];
            int gt = (y->k0 > x.k0) || (y->k0 == x.k0 && (y->k1 > x.k1 || (y->k1 == x.k1 && y->k2 > x.k2)));
            if (!gt) break;
            v[p + 1] = v[p]; --p;
        }
        v[p + 1] = x;
    }
}

/* E9 */
static void observe_world(const oracle_cfg *c, agent_t *ag, int n, double *obs /* [N, width] */) {
    const int M = c->max_other, width = 6 + 7 * M;
    memset(obs, 0, sizeof(double) * (size_t)c->max_agents * width);
    for (int i = 0; i < n; ++i) {
        agent_t *h = &ag[i];
        ego_frame(c, h);
        crit_t crit[ORACLE_MAX_AGENTS];
        int m = 0;
        for (int j = 0; j < n; ++j) {
            if (j == i) continue;
            const agent_t *o = &ag[j];
            double rx = o->px - h->px, ry = o->py - h->py, d = sqrt(rx * rx + ry * ry);
            if (d > c->sensing_horizon) continue;
            double gap = d - h->radius - o->radius;
            double p_orth = rx * h->orth_x + ry * h->orth_y;
            double gr = c->sort_round_gap ? rint(gap * 100.0) / 100.0 : gap;     /* U7a */
            if (!c->sort_tie_lateral) p_orth = 0.0;                               /* U7b: ties keep index order (stable sort) */
            crit[m].j = j;
            if (c->sort_method == SORT_TTI) { crit[m].k0 = -time_to_impact(h, o); crit[m].k1 = -gr; crit[m].k2 = p_orth; }
            else { crit[m].k0 = -gr; crit[m].k1 = p_orth; crit[m].k2 = 0.0; }
            ++m;
        }
        stable_sort(crit, m);                       /* far ... near */
        int first = m > M ? m - M : 0, kept = m - first;
        crit_t *kp = crit + first;
        if (c->sort_method == SORT_CLOSEST_FIRST) {
            for (int k = 0; k < kept; ++k) kp[k].k0 = -kp[k].k0;   /* (+gap, p_orth) */
            stable_sort(kp, kept);
        }
        double *row = obs + (size_t)i * width;
        row[0] = (h->flags & F_LEARNING) ? 1.0 : 0.0;
        row[1] = kept; row[2] = h->dist_to_goal; row[3] = h->heading_ego; row[4] = h->pref_speed; row[5] = h->radius;
        for (int k = 0; k < kept; ++k) {
            const agent_t *o = &ag[kp[k].j];
            double rx = o->px - h->px, ry = o->py - h->py;
            double *f = row + 6 + 7 * k;
            f[0] = rx * h->prll_x + ry * h->prll_y;
            f[1] = rx * h->orth_x + ry * h->orth_y;
            f[2] = o->vx * h->prll_x + o->vy * h->prll_y;
            f[3] = o->vx * h->orth_x + o->vy * h->orth_y;
            f[4] = o->radius;
            f[5] = h->radius + o->radius;
            f[6] = sqrt(rx * rx + ry * ry) - h->radius - o->radius;
        }
    }
}

/* ---- RVO scripted policy: ORCA (van den Berg et al., ISRR 2009), float64, neighbours in agent-index order.
 * Same statement as oracle/cavoid_oracle.py (orca_lines, _lp_on_line, _lp_plane, _lp_least_penetration, rvo_action). */
#define RVO_EPSILON 1e-5
typedef struct { double px, py, dx, dy; } line_t;
static double det2(double ax, double ay, double bx, double by) { return ax * by - ay * bx; }

static int orca_lines(const oracle_cfg *c, const agent_t *ag, int n, int hi, line_t *out) {
    const agent_t *h = &ag[hi];
    const double inv_h = 1.0 / c->rvo_time_horizon;
    int m = 0;
    for (int j = 0; j < n; ++j) {
        if (j == hi) continue;
        const agent_t *o = &ag[j];
        double rpx = o->px - h->px, rpy = o->py - h->py, rvx = h->vx - o->vx, rvy = h->vy - o->vy;
        double dist_sq = rpx * rpx + rpy * rpy;
        double comb = c->rvo_radius_scale * h->radius + c->rvo_radius_scale * o->radius, comb_sq = comb * comb;
        double dx, dy, ucx, ucy;
        if (dist_sq > comb_sq) {
            double wx = rvx - inv_h * rpx, wy = rvy - inv_h * rpy, w_sq = wx * wx + wy * wy, dot1 = wx * rpx + wy * rpy;
            if (dot1 < 0.0 && dot1 * dot1 > comb_sq * w_sq) {
                double w_len = sqrt(w_sq), ux = wx / w_len, uy = wy / w_len, scale = comb * inv_h - w_len;
                dx = uy; dy = -ux; ucx = scale * ux; ucy = scale * uy;
            } else {
                double leg = sqrt(dist_sq - comb_sq);
                if (det2(rpx, rpy, wx, wy) > 0.0) { dx = (rpx * leg - rpy * comb) / dist_sq; dy = (rpx * comb + rpy * leg) / dist_sq; }
                else { dx = -(rpx * leg + rpy * comb) / dist_sq; dy = -(-rpx * comb + rpy * leg) / dist_sq; }
                double dot2 = rvx * dx + rvy * dy;
                ucx = dot2 * dx - rvx; ucy = dot2 * dy - rvy;
            }
        } else {
            double inv_dt = 1.0 / c->dt, wx = rvx - inv_dt * rpx, wy = rvy - inv_dt * rpy;
            double w_len = sqrt(wx * wx + wy * wy), ux = wx / w_len, uy = wy / w_len, scale = comb * inv_dt - w_len;
            dx = uy; dy = -ux; ucx = scale * ux; ucy = scale * uy;
        }
        out[m].px = h->vx + c->rvo_collab_coeff * ucx; out[m].py = h->vy + c->rvo_collab_coeff * ucy;
        out[m].dx = dx; out[m].dy = dy;
        ++m;
    }
    return m;
}

static int lp_on_line(const line_t *ln, int k, double radius, double ox, double oy, int direction_opt, double *x, double *y) {
    const double px = ln[k].px, py = ln[k].py, dx = ln[k].dx, dy = ln[k].dy;
    double dot = px * dx + py * dy, disc = dot * dot + radius * radius - (px * px + py * py);
    if (disc < 0.0) return 0;
    double root = sqrt(disc), t_lo = -dot - root, t_hi = -dot + root;
    for (int i = 0; i < k; ++i) {
        double den = det2(dx, dy, ln[i].dx, ln[i].dy), num = det2(ln[i].dx, ln[i].dy, px - ln[i].px, py - ln[i].py);
        if (fabs(den) <= RVO_EPSILON) { if (num < 0.0) return 0; continue; }
        double t = num / den;
        if (den >= 0.0) t_hi = fmin(t_hi, t); else t_lo = fmax(t_lo, t);
        if (t_lo > t_hi) return 0;
    }
    double t;
    if (direction_opt) t = (ox * dx + oy * dy > 0.0) ? t_hi : t_lo;
    else { t = dx * (ox - px) + dy * (oy - py); t = t < t_lo ? t_lo : (t > t_hi ? t_hi : t); }
    *x = px + t * dx; *y = py + t * dy;
    return 1;
}

static int lp_plane(const line_t *ln, int m, double radius, double ox, double oy, int direction_opt, double *x, double *y) {
    if (direction_opt) { *x = ox * radius; *y = oy * radius; }
    else if (ox * ox + oy * oy > radius * radius) { double nn = sqrt(ox * ox + oy * oy); *x = ox / nn * radius; *y = oy / nn * radius; }
    else { *x = ox; *y = oy; }
    for (int k = 0; k < m; ++k)
        if (det2(ln[k].dx, ln[k].dy, ln[k].px - *x, ln[k].py - *y) > 0.0) {
            double nx, ny;
            if (!lp_on_line(ln, k, radius, ox, oy, direction_opt, &nx, &ny)) return k;
            *x = nx; *y = ny;
        }
    return m;
}

static void lp_least_penetration(const line_t *ln, int m, int begin, double radius, double *x, double *y) {
    double distance = 0.0;
    line_t proj[ORACLE_MAX_AGENTS];
    for (int k = begin; k < m; ++k) {
        const double px = ln[k].px, py = ln[k].py, dx = ln[k].dx, dy = ln[k].dy;
        if (det2(dx, dy, px - *x, py - *y) > distance) {
            int np = 0;
            for (int j = 0; j < k; ++j) {
                double den = det2(dx, dy, ln[j].dx, ln[j].dy), nx, ny;
                if (fabs(den) <= RVO_EPSILON) {
                    if (dx * ln[j].dx + dy * ln[j].dy > 0.0) continue;
                    nx = 0.5 * (px + ln[j].px); ny = 0.5 * (py + ln[j].py);
                } else {
                    double t = det2(ln[j].dx, ln[j].dy, px - ln[j].px, py - ln[j].py) / den;
                    nx = px + t * dx; ny = py + t * dy;
                }
                double fx = ln[j].dx - dx, fy = ln[j].dy - dy, fn = sqrt(fx * fx + fy * fy);
                proj[np].px = nx; proj[np].py = ny; proj[np].dx = fx / fn; proj[np].dy = fy / fn;
                ++np;
            }
            double nx, ny;
            if (lp_plane(proj, np, radius, -dy, dx, 1, &nx, &ny) >= np) { *x = nx; *y = ny; }
            distance = det2(dx, dy, px - *x, py - *y);
        }
    }
}

static void rvo_action(const oracle_cfg *c, const agent_t *ag, int n, int hi, double *a0, double *a1) {
    const agent_t *h = &ag[hi];
    double gx = h->gx - h->px, gy = h->gy - h->py, gn = sqrt(gx * gx + gy * gy);
    double scale = gn > 0.0 ? h->pref_speed / gn : 0.0, pvx = scale * gx, pvy = scale * gy;
    line_t ln[ORACLE_MAX_AGENTS];
    int m = orca_lines(c, ag, n, hi, ln);
    double vx, vy;
    int fail = lp_plane(ln, m, h->pref_speed, pvx, pvy, 0, &vx, &vy);
    if (fail < m) lp_least_penetration(ln, m, fail, h->pref_speed, &vx, &vy);
    double speed = sqrt(vx * vx + vy * vy);
    double delta = speed > 0.0 ? wrap(c, atan2(vy, vx) - h->heading) : 0.0;
    if (fabs(delta) > c->rvo_max_delta_heading) { delta = copysign(c->rvo_max_delta_heading, delta); speed = 0.0; }
    *a0 = speed; *a1 = delta;
}

static void step_world(const oracle_cfg *c, agent_t *ag, int n, const int32_t *act, const float *cont,
                       double *obs, double *rew, uint8_t *done, uint8_t *game_over) {
    const int N = c->max_agents;
    double a0[ORACLE_MAX_AGENTS], a1[ORACLE_MAX_AGENTS];
    /* E4: every agent picks its action first ... */
    int frozen[ORACLE_MAX_AGENTS];                       /* done BEFORE this step's move (U4) */
    for (int i = 0; i < n; ++i) frozen[i] = (ag[i].flags & F_DONE_MASK) ? 1 : 0;
    for (int i = 0; i < n; ++i) {
        a0[i] = a1[i] = 0.0;
        if (ag[i].flags & F_DONE_MASK) continue;
        int pol = (ag[i].flags >> F_POLICY_SHIFT) & F_POLICY_MASK;
        if (pol == POLICY_EXTERNAL || pol == POLICY_FROZEN_NET) {   /* frozen network: its action index comes in like a learner's */
            if (cont) { a0[i] = cont[2 * i]; a1[i] = cont[2 * i + 1]; }
            else { const double *r = c->actions[act[i]]; a0[i] = ag[i].pref_speed * r[0]; a1[i] = r[1]; }
        } else if (pol == POLICY_NONCOOP) { a0[i] = ag[i].pref_speed; a1[i] = -ag[i].heading_ego; }
        else if (pol == POLICY_RVO) rvo_action(c, ag, n, i, &a0[i], &a1[i]);
        if (c->actions_fp32) { a0[i] = (double)(float)a0[i]; a1[i] = (double)(float)a1[i]; }
    }
    /* ... then all move (E5) */
    for (int i = 0; i < n; ++i) take_action(c, &ag[i], a0[i], a1[i]);
    /* E6 */
    int hit[ORACLE_MAX_AGENTS];
    double min_gap[ORACLE_MAX_AGENTS];
    for (int i = 0; i < n; ++i) { hit[i] = 0; min_gap[i] = INFINITY; }
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            if (!c->done_agents_collide && (frozen[i] || frozen[j])) continue;      /* U4 flipped */
            double dx = ag[i].px - ag[j].px, dy = ag[i].py - ag[j].py;
            double gap = sqrt(dx * dx + dy * dy) - (ag[i].radius + ag[j].radius);
            min_gap[i] = fmin(min_gap[i], gap); min_gap[j] = fmin(min_gap[j], gap);
            if (gap <= c->collision_dist) hit[i] = hit[j] = 1;
        }
    /* E7 */
    for (int i = 0; i < N; ++i) { rew[i] = 0.0; done[i] = 1; }
    for (int i = 0; i < n; ++i) {
        double r = c->reward_time_step;
        if (ag[i].flags & F_AT_GOAL) { if (!(ag[i].flags & F_WAS_AT_GOAL)) r = c->reward_at_goal; }
        else if (!(ag[i].flags & F_WAS_IN_COLL)) {
            if (hit[i]) { r = c->reward_collision; ag[i].flags |= F_IN_COLL; }
            else if (min_gap[i] <= c->getting_close_range) r = c->reward_getting_close + c->close_penalty_slope * min_gap[i];
        }
        rew[i] = fmin(fmax(r, c->reward_clip_lo), c->reward_clip_hi);
    }
    /* E9 then E8 */
    observe_world(c, ag, n, obs);
    int all_done = 1;
    for (int i = 0; i < n; ++i) {
        done[i] = (ag[i].flags & F_DONE_MASK) ? 1 : 0;
        if (((ag[i].flags & F_LEARNING) || c->evaluate_mode) && !done[i]) all_done = 0;
    }
    *game_over = (uint8_t)all_done;
}

void oracle_step(const oracle_cfg *c, int64_t W, oracle_state *st, const int32_t *actions, const float *cont,
                 double *obs, double *rew, uint8_t *done, uint8_t *game_over) {
    const int N = c->max_agents, width = 6 + 7 * c->max_other;
    for (int64_t w = 0; w < W; ++w) {
        agent_t ag[ORACLE_MAX_AGENTS];
        int n = load_world(c, st, w, ag);
        step_world(c, ag, n, actions ? actions + w * N : 0, cont ? cont + w * N * 2 : 0,
                   obs + w * N * width, rew + w * N, done + w * N, game_over + w);
        store_world(c, st, w, ag, n);
    }
}

void oracle_observe(const oracle_cfg *c, int64_t W, const oracle_state *st, double *obs) {
    const int N = c->max_agents, width = 6 + 7 * c->max_other;
    for (int64_t w = 0; w < W; ++w) {
        agent_t ag[ORACLE_MAX_AGENTS];
        int n = load_world(c, st, w, ag);
        observe_world(c, ag, n, obs + w * N * width);
    }
}

/* ---- GEN v1 (E2): Philox4x32-10 counter-based scenario generator ---------------------------- */
void oracle_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double u01(uint32_t r) { return (double)(r >> 8) * (1.0 / 16777216.0); }

static uint32_t draw_policy(const oracle_gen *g, const uint32_t b[4], int i) {
    if (i > 0 && u01(b[2]) < g->nonlearning_fraction) {
        double u = u01(b[3]);
        if (u < g->static_fraction) return POLICY_STATIC;
        if (u < g->static_fraction + g->rvo_fraction) return POLICY_RVO;
        return u < g->static_fraction + g->rvo_fraction + g->frozen_fraction ? POLICY_FROZEN_NET : POLICY_NONCOOP;
    }
    return POLICY_EXTERNAL;
}

static void place_agent(const oracle_cfg *c, oracle_state *s, int64_t k, double px, double py, float gx, float gy,
                        float radius, float pref, uint32_t pol) {
    double tx = (double)gx - px, ty = (double)gy - py;
    double dxg = px - (double)gx, dyg = py - (double)gy;
    double straight = (sqrt(dxg * dxg + dyg * dyg) - (c->time_budget_from_goal_edge ? c->near_goal_threshold : 0.0)) / (double)pref;
    s->px[k] = px; s->py[k] = py; s->heading[k] = atan2(ty, tx);
    s->t_remaining[k] = fmax(c->max_time_ratio * straight, c->dt);
    s->gx[k] = gx; s->gy[k] = gy; s->radius[k] = radius; s->pref_speed[k] = pref; s->speed[k] = 0.0f;
    s->flags[k] = F_PRESENT | (pol == POLICY_EXTERNAL ? F_LEARNING : 0u) | (pol << F_POLICY_SHIFT);
}

static void generate_world(const oracle_cfg *c, const oracle_gen *g, uint64_t seed, uint32_t gw, uint32_t ep,
                           oracle_state *s, int64_t w) {
    const int N = c->max_agents;
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, ctr[4] = {gw, ep, 0, 0}, r[4], a[4], b[4];
    if (g->pool_size > 0) {            /* scenario pool: pick the pool entry, which is generator world k of the pool's epoch */
        /* splitmix64-style finaliser of (seed, gw, ep), reduced to [0, P) by multiply-shift */
        uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)gw + 1ull) + 0xC2B2AE3D27D4EB4Full * ((uint64_t)ep + 1ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        ctr[0] = gw = (uint32_t)(((z >> 32) * (uint64_t)(uint32_t)g->pool_size) >> 32);
        ctr[1] = ep = g->pool_epoch;
    }
    oracle_philox4x32(ctr, key, r);
    int span = g->max_agents - g->min_agents + 1;
    int n = g->min_agents + (int)(r[0] % (uint32_t)span);
    for (int i = n; i < N; ++i) {
        int64_t k = w * N + i;
        s->px[k] = s->py[k] = s->heading[k] = s->t_remaining[k] = 0.0;
        s->gx[k] = s->gy[k] = s->radius[k] = s->pref_speed[k] = s->speed[k] = 0.0f;
        s->flags[k] = 0;
    }
    if (g->mode == 0) {                /* GEN v1: ring, antipodal goals */
        double base = fmax(4.0, 0.7 * n), ring = base * (1.0 + u01(r[1])), phase = u01(r[2]);
        for (int i = 0; i < n; ++i) {
            ctr[2] = 1; ctr[3] = (uint32_t)i; oracle_philox4x32(ctr, key, a);
            ctr[2] = 2; oracle_philox4x32(ctr, key, b);
            float radius = (float)(0.2 + 0.6 * u01(a[0]));
            float pref = (float)(0.5 + 1.5 * u01(a[1]));
            double turn = phase + (i + (u01(a[2]) - 0.5) * 2.0 * g->angle_jitter) / n;
            double theta = 2.0 * PI * turn;
            double px = ring * cos(theta), py = ring * sin(theta);
            float gx = (float)(-px + (u01(b[0]) - 0.5) * 2.0 * g->goal_jitter);
            float gy = (float)(-py + (u01(b[1]) - 0.5) * 2.0 * g->goal_jitter);
            place_agent(c, s, w * N + i, px, py, gx, gy, radius, pref, draw_policy(g, b, i));
        }
        return;
    }
    /* GEN v2: uniform boxes, one agent after the other, rejection sampling against the agents already placed */
    const double *box = n < g->box_large_from ? g->box_small : g->box_large;
    double side = box[0] + (box[1] - box[0]) * u01(r[1]);
    for (int i = 0; i < n; ++i) {
        uint32_t cc[4];
        ctr[2] = 1; ctr[3] = (uint32_t)i; oracle_philox4x32(ctr, key, a);
        ctr[2] = 2; oracle_philox4x32(ctr, key, b);
        float radius = (float)(0.2 + 0.6 * u01(a[0]));
        float pref = (float)(0.5 + 1.5 * u01(a[1]));
        double sx, sy;
        float gx, gy;
        int attempt = 0;
        for (;;) {
            ctr[2] = 3u + (uint32_t)attempt; oracle_philox4x32(ctr, key, cc);
            sx = side * (2.0 * u01(cc[0]) - 1.0); sy = side * (2.0 * u01(cc[1]) - 1.0);
            gx = (float)(side * (2.0 * u01(cc[2]) - 1.0)); gy = (float)(side * (2.0 * u01(cc[3]) - 1.0));
            double tx = (double)gx - sx, ty = (double)gy - sy;
            int ok = sqrt(tx * tx + ty * ty) >= g->min_trip;
            for (int j = 0; j < i; ++j) {
                int64_t kj = w * N + j;
                double margin = ((double)radius + (double)s->radius[kj]) + c->getting_close_range;
                double ax = sx - s->px[kj], ay = sy - s->py[kj], bx = (double)gx - (double)s->gx[kj], by = (double)gy - (double)s->gy[kj];
                if (sqrt(ax * ax + ay * ay) < margin || sqrt(bx * bx + by * by) < margin) ok = 0;
            }
            ++attempt;
            if (ok || attempt >= 100) break;
            if (attempt % 10 == 0) side = side * 1.01;
        }
        place_agent(c, s, w * N + i, sx, sy, gx, gy, radius, pref, draw_policy(g, b, i));
    }
}

void oracle_generate(const oracle_cfg *c, const oracle_gen *g, uint64_t seed, int64_t world_offset,
                     const uint32_t *episode, const uint8_t *mask, int64_t W, oracle_state *st) {
    for (int64_t w = 0; w < W; ++w)
        if (!mask || mask[w]) generate_world(c, g, seed, (uint32_t)(world_offset + w), episode[w], st, w);
}

/* the auto-reset step with either action form: int32 [W,N] table indices (cont == NULL) or float [W,N,2] continuous actions
 * (speed, heading change) resp. holonomic velocities (cont != NULL) -- the VecEnv convention of oracle_step_autoreset around oracle_step's
 * two action forms */
void oracle_step_autoreset_any(const oracle_cfg *c, const oracle_gen *g, uint64_t seed, int64_t world_offset,
                               uint32_t *episode, int64_t W, oracle_state *st, const int32_t *actions, const float *cont,
                           double *obs, double *rew, uint8_t *done, uint8_t *game_over) {
    const int N = c->max_agents, width = 6 + 7 * c->max_other;
    for (int64_t w = 0; w < W; ++w) {
        agent_t ag[ORACLE_MAX_AGENTS];
        int n = load_world(c, st, w, ag);
        step_world(c, ag, n, actions ? actions + w * N : 0, cont ? cont + w * N * 2 : 0, obs + w * N * width, rew + w * N, done + w * N, game_over + w);
        store_world(c, st, w, ag, n);
        if (game_over[w]) {
            episode[w] += 1;
            generate_world(c, g, seed, (uint32_t)(world_offset + w), episode[w], st, w);
            n = load_world(c, st, w, ag);
            observe_world(c, ag, n, obs + w * N * width);
        }
    }
}

void oracle_step_autoreset(const oracle_cfg *c, const oracle_gen *g, uint64_t seed, int64_t world_offset,
                           uint32_t *episode, int64_t W, oracle_state *st, const int32_t *actions,
                           double *obs, double *rew, uint8_t *done, uint8_t *game_over) {
    oracle_step_autoreset_any(c, g, seed, world_offset, episode, W, st, actions, 0, obs, rew, done, game_over);
}
