/* CPU ORACLE (test infrastructure, NOT product code) -- see cavoid_oracle.c.
 * PARITY UNPINNED for the env half: the env source is absent from /root/reference
 * (.gitmodules:1-3, empty submodule); this restates the published algorithm.  */
#ifndef CAVOID_ORACLE_H
#define CAVOID_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MAX_ACTIONS 32
#define ORACLE_MAX_AGENTS 64

typedef struct oracle_cfg {
    double dt, near_goal_threshold, max_time_ratio, collision_dist, getting_close_range;
    double reward_at_goal, reward_collision, reward_getting_close, reward_time_step;
    double sensing_horizon, close_penalty_slope, max_turn_rate, reward_clip_lo, reward_clip_hi;
    double rvo_time_horizon, rvo_collab_coeff, rvo_radius_scale, rvo_max_delta_heading;   /* RVO scripted policy */
    int32_t max_agents;        /* N */
    int32_t max_other;         /* M */
    int32_t sort_method;       /* 0 closest_last, 1 closest_first, 2 time_to_impact */
    int32_t actions_fp32;
    int32_t timeout_enabled;
    int32_t dynamics;          /* 0 unicycle, 1 unicycle max-turn-rate, 2 holonomic */
    int32_t num_actions;
    int32_t evaluate_mode;     /* game over when EVERY agent is done (EVALUATE_MODE) instead of every learning agent */
    int32_t time_budget_from_goal_edge; /* U11: budget = ratio*(dist - near_goal)/pref (1, default) or ratio*dist/pref (0) */
    int32_t wrap_closed_end;     /* U2: 0 [-pi, pi) (default), 1 (-pi, pi] */
    int32_t done_agents_collide; /* U4: 1 (default) frozen agents still collide with movers, 0 pairs with a frozen agent are skipped */
    int32_t sort_round_gap;      /* U7a: 1 (default) order by the gap rounded to centimetres, 0 by the exact gap */
    int32_t sort_tie_lateral;    /* U7b: 1 (default) ties by lateral offset then index, 0 by index alone */
    int32_t _pad0;
    double actions[ORACLE_MAX_ACTIONS][2];
} oracle_cfg;

typedef struct oracle_gen {
    int32_t min_agents, max_agents;
    double nonlearning_fraction, static_fraction, goal_jitter, angle_jitter;
    int32_t pool_size;   /* > 0: episode ep of world gw is generator world pool_index(seed,gw,ep) (splitmix64 finaliser, multiply-shift) of episode pool_epoch */
    int32_t mode;        /* 0 GEN v1 (ring), 1 GEN v2 (boxes, rejection sampling) */
    double rvo_fraction; /* of the scripted agents: P(static) = static_fraction, P(RVO) = rvo_fraction, rest non-cooperative */
    double box_small[2], box_large[2], min_trip;
    int32_t box_large_from;
    uint32_t pool_epoch;
    double frozen_fraction; /* ... P(frozen network, policy 4) */
} oracle_gen;

/* SoA over A = W*N agents, agent a = w*N + i */
typedef struct oracle_state {
    double *px, *py, *heading, *t_remaining;
    float *gx, *gy, *radius, *pref_speed, *speed;
    uint32_t *flags;
} oracle_state;

void oracle_default_cfg(oracle_cfg *cfg, int32_t max_agents, int32_t max_other);

/* actions: int32 [W,N] indices (cont == NULL) or float [W,N,2] continuous (cont != NULL).
 * obs: double [W,N,2+4+7M]; rew: double [W,N]; done: u8 [W,N]; game_over: u8 [W]. */
void oracle_step(const oracle_cfg *cfg, int64_t W, oracle_state *st, const int32_t *actions,
                 const float *cont, double *obs, double *rew, uint8_t *done, uint8_t *game_over);

void oracle_observe(const oracle_cfg *cfg, int64_t W, const oracle_state *st, double *obs);

/* GEN v1: regenerate worlds whose mask byte is non-zero (mask == NULL: all).  World w uses the
 * Philox counter (world_offset + w, episode[w], stream, agent). */
void oracle_generate(const oracle_cfg *cfg, const oracle_gen *gen, uint64_t seed, int64_t world_offset,
                     const uint32_t *episode, const uint8_t *mask, int64_t W, oracle_state *st);

/* step, then for every game-over world: episode[w] += 1, regenerate, and overwrite its obs rows
 * with the fresh episode's first observation (VecEnv auto-reset convention). */
void oracle_step_autoreset(const oracle_cfg *cfg, const oracle_gen *gen, uint64_t seed, int64_t world_offset,
                           uint32_t *episode, int64_t W, oracle_state *st, const int32_t *actions,
                           double *obs, double *rew, uint8_t *done, uint8_t *game_over);

/* ... the same with either action form (cont != NULL: float [W,N,2] continuous / holonomic actions, like oracle_step) */
void oracle_step_autoreset_any(const oracle_cfg *cfg, const oracle_gen *gen, uint64_t seed, int64_t world_offset,
                               uint32_t *episode, int64_t W, oracle_state *st, const int32_t *actions, const float *cont,
                               double *obs, double *rew, uint8_t *done, uint8_t *game_over);

void oracle_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
