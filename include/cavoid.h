/* cavoid.h -- C ABI of the MI355X-native batched collision-avoidance env.step hot path.
 *
 * Drop-in boundary.  The reference has NO FFI for this path: the seam is a duck-typed Python
 * object.  Each entry point below names the reference interface it
 * stands in for (paths relative to /root/reference):
 *
 *   cavoid_create / cavoid_destroy ... `env, one_env = create_env()`            ga3c/GA3C/Environment.py:54-56
 *   cavoid_reset ..................... `observations = self.game.reset()`        ga3c/GA3C/Environment.py:106
 *   cavoid_step ...................... `self.game.step(action)` -> 4-tuple        ga3c/GA3C/Environment.py:112
 *                                      (obs, rewards, game_over, which_agents_done) ga3c/GA3C/ProcessAgent.py:149-157
 *   cavoid_step_autoreset ............ the per-episode `env.reset()` + step loop  ga3c/GA3C/ProcessAgent.py:105-116
 *   cavoid_*_packed .................. the same calls, returning ONE record per agent (obs | reward | done): what
 *                                      an actor hands back per step                  ga3c/GA3C/ProcessAgent.py:149-157
 *   cavoid_comm_* / cavoid_gather_* .. (new) the actors -> trainer hand-over across GPUs; the reference moves it
 *                                      through mp.Queue between processes            ga3c/GA3C/ProcessAgent.py:221,238
 *   cavoid_step_continuous ........... the env-level continuous action space      run-ws/config.yaml:3-5 (ACTION_SPACE_TYPE)
 *   cavoid_observe ................... the obs half of reset()/step()             ga3c/GA3C/Environment.py:81-91
 *   cavoid_set_state/get_state ....... (new) explicit initial states for parity runs; env checkpointing
 *   cavoid_default_cfg ............... the env's Config scalars                   ga3c/GA3C/Config.py:29,34-52 +
 *                                      checkpoints/regression/wandb/run-ws/config.yaml
 *   cavoid_default_actions ........... `Actions().actions[11,2]`                  ga3c/GA3C/Server.py:36,51-52
 *
 * Conventions: plain pointers and sizes, no torch types.  Every data pointer is a DEVICE pointer
 * on the env's device and is owned by the caller; the library owns only its internal world
 * buffer.  All work is enqueued asynchronously on the `hipStream_t` passed as `void *stream`
 * (NULL = the default stream).  Return 0 on success, a negative CAVOID_E* code on failure; the
 * library never throws, aborts or falls back to a CPU path.  A handle is not thread-safe; use
 * one per device/stream.
 *
 * Layouts (W worlds, N = max_agents, M = max_other, flat agent index a = w*N + i):
 *   obs        float  [W, N, 2+4+7M]  col0 is_learning, col1 num_other_agents, col2 dist_to_goal,
 *                                     col3 heading_ego_frame, col4 pref_speed, col5 radius, then M x
 *                                     [p_par, p_orth, v_par, v_orth, r_other, r_host+r_other, gap]
 *                                     (ga3c/GA3C/Config.py:40,72-76; NetworkVP_rnn.py:58-61)
 *   rewards    float  [W, N]
 *   done       u8     [W, N]          which_agents_done; 1 for absent agents
 *   game_over  u8     [W]             every *learning* agent of the world is done (TRAIN_MODE)
 *   actions    int32  [W, N]          index into the action table; ignored for done / scripted agents
 *   packed     float  [W, N, 2+4+7M+2] the obs row followed by the step's reward and done (0.0f / 1.0f): the
 *                                     per-agent record of the multi-GPU gather
 *   state_f64  double [4, W*N]        px, py, heading, t_remaining           (SoA, field-major)
 *   state_f32  float  [5, W*N]        gx, gy, radius, pref_speed, speed
 *   flags      u32    [W*N]           CAVOID_F_* bits
 */
#ifndef CAVOID_H
#define CAVOID_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAVOID_ABI_VERSION 3
#define CAVOID_MAX_ACTIONS 32
#define CAVOID_MAX_AGENTS 16

/* agent flag bits */
#define CAVOID_F_AT_GOAL 0x01u
#define CAVOID_F_RAN_OUT 0x02u
#define CAVOID_F_IN_COLL 0x04u
#define CAVOID_F_WAS_AT_GOAL 0x08u
#define CAVOID_F_WAS_IN_COLL 0x10u
#define CAVOID_F_PRESENT 0x20u
#define CAVOID_F_LEARNING 0x40u
#define CAVOID_F_POLICY_SHIFT 8 /* bits 8..10: 0 external (learning), 1 static, 2 non-cooperative, 3 RVO (ORCA),
                                   4 frozen network: a NON-learning agent whose action index the caller supplies like a learner's
                                   (from a second, frozen NetworkVP_rnn -- the GA3C-CADRL agent, ga3c/GA3C/Server.py:36) */
#define CAVOID_F_POLICY_MASK 7u
enum { CAVOID_POLICY_EXTERNAL = 0, CAVOID_POLICY_STATIC = 1, CAVOID_POLICY_NONCOOP = 2, CAVOID_POLICY_RVO = 3, CAVOID_POLICY_FROZEN_NET = 4 };
#define CAVOID_F_DONE_MASK 0x07u

enum { CAVOID_SORT_CLOSEST_LAST = 0, CAVOID_SORT_CLOSEST_FIRST = 1, CAVOID_SORT_TIME_TO_IMPACT = 2 };
enum { CAVOID_DYN_UNICYCLE = 0, CAVOID_DYN_UNICYCLE_MAX_TURN = 1, CAVOID_DYN_HOLONOMIC = 2 };

enum {
    CAVOID_OK = 0,
    CAVOID_EINVAL = -1,    /* bad argument / config */
    CAVOID_ENOMEM = -2,    /* device allocation failed */
    CAVOID_EHIP = -3,      /* a HIP runtime call failed (see cavoid_last_hip_error) */
    CAVOID_EUNSUPPORTED = -4, /* max_agents outside the compiled range */
    CAVOID_ENODEVICE = -5, /* no usable gfx950 device */
    CAVOID_ECOMM = -6      /* an RCCL call failed (see cavoid_last_comm_error) */
};

typedef struct cavoid_cfg {
    uint32_t struct_size;      /* = sizeof(cavoid_cfg); checked by cavoid_create */
    uint32_t abi_version;      /* = CAVOID_ABI_VERSION */
    int32_t max_agents;        /* N: MAX_NUM_AGENTS_IN_ENVIRONMENT   (Config.py:34-37) */
    int32_t max_other;         /* M: MAX_NUM_OTHER_AGENTS_OBSERVED   (Config.py:46-49) */
    int32_t sort_method;       /* AGENT_SORTING_METHOD               (run-ws/config.yaml:9-11) */
    int32_t dynamics;          /* CAVOID_DYN_* */
    int32_t actions_fp32;      /* joint action array is float32 (default 1) */
    int32_t timeout_enabled;   /* default 1 */
    int32_t num_actions;       /* NUM_ACTIONS (Config.py:79) */
    int32_t evaluate_mode;     /* 0 (TRAIN_MODE): game over when every LEARNING agent is done; 1 (EVALUATE_MODE): every agent */
    int32_t time_budget_from_goal_edge; /* U11: 1 (default): an agent's time budget is MAX_TIME_RATIO * (dist_to_goal -
                                   NEAR_GOAL_THRESHOLD) / pref_speed (upstream agent.py as recalled); 0: SURVEY App. A's
                                   MAX_TIME_RATIO * dist_to_goal / pref_speed.  Either way at least one DT. */
    int32_t wrap_closed_end;   /* U2: 0 (default) angles wrap to [-pi, pi); 1: to (-pi, pi] */
    int32_t done_agents_collide; /* U4: 1 (default) an agent that is already done (frozen) still takes part in the others' collision
                                   test and nearest gap; 0: a pair with an agent that was done before the step is skipped */
    int32_t sort_round_gap;    /* U7a: 1 (default) neighbours are ordered by the gap rounded to centimetres; 0: by the exact gap */
    int32_t sort_tie_lateral;  /* U7b: 1 (default) equal (rounded) gaps are ordered by the lateral offset p_orth, then agent index;
                                   0: by agent index alone (a stable sort on the gap) */
    int32_t gen_lookahead;     /* R (power of two, 0 = off; needs gen_pool_size == 0): a FRESH scenario per episode without the generator on the step's
                                * critical path -- see "scenario look-ahead" below.  (this word was padding up to ABI 3: zero = off) */
    double dt;                 /* DT 0.2 */
    double near_goal_threshold;/* 0.2 */
    double max_time_ratio;     /* 2.0 */
    double collision_dist;     /* 0.0 */
    double getting_close_range;/* 0.2 */
    double reward_at_goal;     /* 1.0 */
    double reward_collision;   /* -0.25 */
    double reward_getting_close;/* -0.1 */
    double reward_time_step;   /* 0.0 */
    double close_penalty_slope;/* +0.5 (the published reward, arXiv:1805.01956; -0.5 = upstream code as the survey recalls it): r = reward_getting_close + slope*gap */
    double reward_clip_lo, reward_clip_hi; /* [-0.25, 1.0] */
    double sensing_horizon;    /* +inf */
    double max_turn_rate;      /* 3.0 rad/s (CAVOID_DYN_UNICYCLE_MAX_TURN only) */
    double actions[CAVOID_MAX_ACTIONS][2]; /* [speed fraction, delta heading] */
    /* scenario generator "GEN v1" used by cavoid_reset / autoreset */
    int32_t gen_min_agents, gen_max_agents;
    double gen_nonlearning_fraction, gen_static_fraction, gen_goal_jitter, gen_angle_jitter;
    /* scenario pool: > 0 pre-generates that many GEN v1 scenarios at cavoid_seed() time (pool entry k
     * = generator world k, episode 0); the episode `ep` of global world `gw` then uses entry
     * hash(seed, gw, ep) -> [0, pool_size) -- a gather instead of the generator on the step's critical
     * path (cf. the reference env's fixed test-case sets, NUM_TEST_CASES run-ws/config.yaml:136-138).
     * 0 = every episode runs the generator in-kernel with counter (gw, ep).  Default 65536. */
    int32_t gen_pool_size;
    /* scenario look-ahead (gen_pool_size == 0 and gen_lookahead = R > 0): every world owns a ring of R pre-generated scenarios of ITS OWN next
     * episodes -- slot e % R holds the generator's scenario of (global world, episode e), the very one the in-kernel generator would make --
     * refilled by a small kernel in front of the stepping launches (only the slots consumed since the last refill are generated), so a
     * restart is a gather, as with the pool, but every episode is the fresh, exact scenario of gen_pool_size = 0 (bitwise: tests).  A launch
     * may hold at most R - 1 steps (a world can restart at most once per step): more is CAVOID_EUNSUPPORTED.  Memory: W * R * N * 64 bytes. */
    /* GEN v2 (gen_mode = 1): starts and goals uniform in a box (half side ~ U(gen_box_small) for worlds of fewer than
     * gen_box_large_from agents, ~ U(gen_box_large) otherwise), agents placed one after the other by rejection sampling
     * against those already placed (starts and goals at least r_i + r_j + getting_close_range apart, trips of at least
     * gen_min_trip) -- the shape of upstream's get_testcase_random as recalled (TEST_CASE_FN, run-ws/config.yaml:281-283).
     * cavoid_reset and the auto-reset steps generate in-kernel (the placement is sequential over a world's agents: the wavefront
     * that owns the world generates it cooperatively) or, with gen_pool_size > 0, take GEN v2 scenarios from the pool. */
    int32_t gen_mode;
    int32_t gen_box_large_from;   /* 5 */
    uint32_t gen_pool_epoch;      /* the pool holds generator worlds 0..P-1 of THIS episode index (cavoid_pool_refresh) */
    int32_t rvo_enabled;          /* agents with policy 3 may exist: the step kernels reserve the ORCA scratch (4 KiB of LDS per
                                     wavefront and neighbour; max_agents <= 15, else CAVOID_EUNSUPPORTED) */
    double gen_rvo_fraction;      /* of the scripted agents: P(static) = gen_static_fraction, P(RVO) = this, P(frozen network) =
                                     gen_frozen_fraction, the rest non-cooperative */
    double gen_frozen_fraction;   /* policy 4 agents (their actions come from the caller: cavoid_policy_rows lists them) */
    double gen_box_small[2], gen_box_large[2];   /* (4,5), (6,8) */
    double gen_min_trip;          /* 1.0 */
    /* RVO scripted policy: ORCA over the other agents' positions and last velocities */
    double rvo_time_horizon;      /* RVO_TIME_HORIZON 5.0   (run-ws/config.yaml:237-239) */
    double rvo_collab_coeff;      /* RVO_COLLAB_COEFF 0.5   (run-ws/config.yaml:234-236) */
    double rvo_radius_scale;      /* 1.05: the policy inflates every radius by 5 % */
    double rvo_max_delta_heading; /* pi/6: larger turns are clipped and taken standing still */
} cavoid_cfg;

typedef struct cavoid_env cavoid_env;
typedef struct cavoid_rollout cavoid_rollout;   /* (section 'rollout' below) */
typedef struct cavoid_policy cavoid_policy;     /* (section 'fused policy inference' below) */

int cavoid_abi_version(void);
const char *cavoid_strerror(int code);
int cavoid_last_hip_error(void);                /* raw hipError_t of the last CAVOID_EHIP */
int cavoid_default_cfg(cavoid_cfg *cfg, int32_t max_agents, int32_t max_other);
int cavoid_default_actions(double (*table)[2], int32_t *num_actions);

/* world_offset: global id of this handle's first world (RNG streams are keyed on global world
 * ids so that results do not depend on how worlds are sharded over GPUs). */
int cavoid_create(const cavoid_cfg *cfg, int64_t num_worlds, int64_t world_offset, int device, cavoid_env **out);
void cavoid_destroy(cavoid_env *env);
int64_t cavoid_num_worlds(const cavoid_env *env);
int32_t cavoid_obs_width(const cavoid_env *env);

/* seed the generator; episode (device u32 [W]) may be NULL = "before episode 0" for every world */
int cavoid_seed(cavoid_env *env, uint64_t seed, const uint32_t *episode, void *stream);
int cavoid_get_episode(cavoid_env *env, uint32_t *episode_out, void *stream);
/* re-fill the scenario pool with generator worlds 0..P-1 of episode index `epoch` (cavoid_seed fills epoch
 * cfg.gen_pool_epoch): a long run that wants fresh scenarios without the generator on the step's critical path
 * refreshes the pool every so often.  No-op without a pool. */
int cavoid_pool_refresh(cavoid_env *env, uint32_t epoch, void *stream);

/* the agents of a world must be packed: CAVOID_F_PRESENT rows first (indices 0..n-1), absent rows after */
int cavoid_set_state(cavoid_env *env, const double *state_f64, const float *state_f32, const uint32_t *flags, void *stream);
int cavoid_get_state(cavoid_env *env, double *state_f64, float *state_f32, uint32_t *flags, void *stream);

/* start the next episode in every world whose mask byte is non-zero (mask NULL = all worlds) and
 * write the observation of ALL worlds (obs may be NULL to skip) */
int cavoid_reset(cavoid_env *env, const uint8_t *world_mask, float *obs, void *stream);
int cavoid_observe(cavoid_env *env, float *obs, void *stream);

int cavoid_step(cavoid_env *env, const int32_t *actions, float *obs, float *rewards, uint8_t *done,
                uint8_t *game_over, void *stream);
int cavoid_step_continuous(cavoid_env *env, const float *actions /* [W,N,2] */, float *obs, float *rewards,
                           uint8_t *done, uint8_t *game_over, void *stream);
/* step; worlds that end are restarted in the same launch and their obs rows hold the first
 * observation of the new episode (rewards/done/game_over still describe the finished step) */
int cavoid_step_autoreset(cavoid_env *env, const int32_t *actions, float *obs, float *rewards, uint8_t *done,
                          uint8_t *game_over, void *stream);
/* n_steps back-to-back autoreset steps in ONE launch; step t reads actions + t*action_stride (int32 elements) and
 * writes its outputs into SLOT t: with S = out_step_stride (in WORLDS, >= num_worlds) the outputs are arrays
 *   obs [n_steps, S, N, width], rewards / done [n_steps, S, N], game_over [n_steps, S]
 * -- what ProcessAgent.run_episode consumes of EVERY step (rewards, which_agents_done, the next observation:
 * ga3c/GA3C/ProcessAgent.py:149-157).  out_step_stride = 0: every step overwrites slot 0 (after the call the outputs hold
 * the LAST step's).  Worlds are independent and a wavefront owns whole worlds, so the world state stays in registers
 * between the steps: per step only the action slice is read and the outputs are written; the world buffer is updated
 * once.  The actions are pre-staged (scripted / open-loop runs; a policy in the loop: cavoid_step_autoreset or
 * cavoid_actor_run). */
int cavoid_step_autoreset_n(cavoid_env *env, const int32_t *actions, int64_t action_stride, int32_t n_steps, int64_t out_step_stride,
                            float *obs, float *rewards, uint8_t *done, uint8_t *game_over, void *stream);

/* The env-level CONTINUOUS action space (run-ws/config.yaml:3-5, ACTION_SPACE_TYPE = 0: "continuous" at the gym level; SURVEY App. A:
 * the discretisation lives in the policy) in the auto-reset and K-step launch forms: float actions [n_steps][W,N,2] -- (speed,
 * heading change) for the unicycle dynamics, a velocity (vx, vy) for CAVOID_DYN_HOLONOMIC (which ONLY these entry points and
 * cavoid_step_continuous can step: it has no table actions) -- slice t at actions + t*action_stride FLOATS (>= 2*W*N, or 0 with
 * n_steps == 1); everything else as cavoid_step_autoreset / _n / _packed: in-kernel restarts from the pool, the look-ahead rings or the
 * in-step generator, every step's outputs in its own slot, scripted agents acting by their own rule.  (A continuous K-step launch
 * runs the single-wavefront step loop: the role-split and pipelined forms decode table actions.) */
int cavoid_step_continuous_autoreset(cavoid_env *env, const float *actions /* [W,N,2] */, float *obs, float *rewards, uint8_t *done,
                                     uint8_t *game_over, void *stream);
int cavoid_step_continuous_autoreset_n(cavoid_env *env, const float *actions, int64_t action_stride, int32_t n_steps, int64_t out_step_stride,
                                       float *obs, float *rewards, uint8_t *done, uint8_t *game_over, void *stream);

/* ---- packed outputs: one record per agent, [W, N, cavoid_packed_width()] floats = (obs row | reward | done) ----
 * The kernel assembles the record in its LDS tile and writes it with the same coalesced stores as the plain
 * observation: no separate reward / done arrays, no pack pass before the multi-GPU gather.  reset / observe write
 * reward 0 and the agents' current done state. */
int32_t cavoid_packed_width(const cavoid_env *env);
int cavoid_reset_packed(cavoid_env *env, const uint8_t *world_mask, float *packed, void *stream);
int cavoid_observe_packed(cavoid_env *env, float *packed, void *stream);
int cavoid_step_packed(cavoid_env *env, const int32_t *actions, float *packed, uint8_t *game_over, void *stream);
/* packed [n_steps, S, N, width + 2] and game_over [n_steps, S] with S = out_step_stride worlds, or one slot when it is 0 */
int cavoid_step_autoreset_packed(cavoid_env *env, const int32_t *actions, int64_t action_stride, int32_t n_steps, int64_t out_step_stride,
                                 float *packed, uint8_t *game_over, void *stream);
int cavoid_step_continuous_autoreset_packed(cavoid_env *env, const float *actions, int64_t action_stride, int32_t n_steps,
                                            int64_t out_step_stride, float *packed, uint8_t *game_over, void *stream);

/* ---- multi-GPU hand-over: ONE all-gather of every rank's packed shard (RCCL over xGMI) ------------------------------
 * One process per GPU, contiguous world ranges (rank r owns worlds [r*W, (r+1)*W)).  The communicator is created from
 * a 128-byte id made on rank 0 by cavoid_comm_unique_id and handed to the other ranks out of band (the host side uses
 * its process-group store).  cavoid_gather_begin(slot) enqueues, on the communicator's OWN stream, an ncclAllGather of
 * `floats_per_rank` floats from `send` into `recv` [nranks * floats_per_rank] after everything enqueued so far on
 * `producer_stream` (the step that wrote `send`); it returns at once.  cavoid_gather_wait(slot) makes a stream wait
 * for the last gather begun in that slot (no host synchronisation).  Two slots: with send / recv double-buffered and
 * slot = t % 2, step t+1 (writing the other buffer) overlaps gather t; before step t+2 re-uses the buffers the
 * producer stream waits on the slot.  With nranks == 1 the gather is a device copy (unless CAVOID_COMM_FORCE_RCCL, below).
 * Errors: CAVOID_ECOMM (see cavoid_last_comm_error). */
typedef struct cavoid_comm cavoid_comm;
#define CAVOID_COMM_ID_BYTES 128
#define CAVOID_COMM_SLOTS 2
int cavoid_comm_unique_id(void *id_out);
int cavoid_comm_create(const void *unique_id, int32_t nranks, int32_t rank, int device, cavoid_comm **out);
/* flags: CAVOID_COMM_FORCE_RCCL = a ONE-rank communicator is a real RCCL communicator too (ncclCommInitRank with nranks = 1;
 * cavoid_gather_begin issues ncclAllGather, cavoid_gatherv_begin a grouped ncclSend / ncclRecv to itself) instead of the device
 * copy -- a development switch that lets a 1-GPU box execute the very calls a multi-rank communicator makes.  unique_id may be
 * NULL for one rank (an id is made internally).  cavoid_comm_create = this with flags 0, or CAVOID_COMM_FORCE_RCCL when the
 * environment variable CAVOID_COMM_FORCE_RCCL is set to anything but "" / "0".  Unknown flag bits: CAVOID_EINVAL. */
#define CAVOID_COMM_FORCE_RCCL 1u
int cavoid_comm_create_ex(const void *unique_id, int32_t nranks, int32_t rank, int device, uint32_t flags, cavoid_comm **out);
/* what the handle is: any of the out pointers may be NULL.  uses_rccl = 1 when the gathers go through RCCL (0 = one rank, device
 * copy); rccl_version = ncclGetVersion() of the library bound (0 when uses_rccl is 0). */
int cavoid_comm_info(const cavoid_comm *comm, int32_t *nranks, int32_t *rank, int32_t *uses_rccl, int32_t *rccl_version);
void cavoid_comm_destroy(cavoid_comm *comm);
int cavoid_gather_begin(cavoid_comm *comm, int32_t slot, const float *send, float *recv, int64_t floats_per_rank,
                        void *producer_stream);
/* the same hand-over for RAGGED shards and / or to ONE rank: counts[r] (host array [nranks], identical on every rank) = floats of
 * rank r's shard; recv [sum(counts)] holds the shards in rank order.  root < 0: every rank receives every shard; root >= 0:
 * only that rank does (the trainer rank of the full GA3C loop; recv may be NULL on the others).  Point-to-point sends inside
 * one RCCL group: on the fully connected xGMI mesh every shard travels over its own link (a direct gather, no padding). */
int cavoid_gatherv_begin(cavoid_comm *comm, int32_t slot, const float *send, float *recv, const int64_t *counts, int32_t root,
                         void *producer_stream);
int cavoid_gather_wait(cavoid_comm *comm, int32_t slot, void *consumer_stream);
int cavoid_last_comm_error(void);               /* raw ncclResult_t of the last CAVOID_ECOMM */

/* as cavoid_step_autoreset_n in launches of `steps_per_launch` steps, every launch carrying its own HIP start/stop
 * event pair (recorded with the dispatch, so launch gaps are excluded); synchronises and returns the MEAN duration
 * of a launch in milliseconds.  Measurement aid for bench.py's roofline figure. */
int cavoid_step_autoreset_n_timed(cavoid_env *env, const int32_t *actions, int64_t action_stride, int32_t n_steps,
                                  int32_t steps_per_launch, int64_t out_step_stride, float *obs, float *rewards, uint8_t *done,
                                  uint8_t *game_over, void *stream, float *mean_launch_ms);

/* the (world, agent) slots whose agent runs scripted policy `policy_id` (CAVOID_POLICY_*) and is present and, with
 * only_running != 0, not done: row_index int32 [W*N] (unspecified order), row_count int32 [1], both on the device (feed them
 * to cavoid_policy_forward_rows of the frozen network that drives CAVOID_POLICY_FROZEN_NET agents). */
int cavoid_policy_rows(cavoid_env *env, int32_t policy_id, int32_t only_running, int32_t *row_index, int32_t *row_count, void *stream);

/* ---- batched GA3C actor bookkeeping (rollout) -------------------------------------------------------
 * Stands in for one ProcessAgent per world: ProcessAgent.run_episode / _accumulate_rewards /
 * convert_to_nparray (ga3c/GA3C/ProcessAgent.py:54-87,105-211) and the training_q / episode_log_q
 * puts (:238,243).  cavoid_rollout_push records one env step for every learning (world, agent) slot
 * in a caller-owned TIME-MAJOR experience store of `ring_len` step blocks (block = step % ring_len):
 *   x float [ring_len,W*N,D] (state the policy acted on), val double [ring_len,W*N] (the step's reward),
 *   ret float [ring_len,W*N] (n-step return y_r the row was emitted with), act u8, emit_t int32 (-1 = pending
 *   or nothing recorded, >= 0 = a training row, emitted at that step).  A block is final once it is older than
 *   time_max + 1 steps; the caller compacts the emit_t >= 0 rows of final blocks into (x, y_r, action) batches
 *   (the trainer's one-hot is eye(num_actions)[act]).
 *   prev_obs float [W,N,1+D] = the observation the policy acted on (Environment.previous_state plus col 0),
 *   actions int32 [W,N], values float [W,N] (V(s_t)), rewards/done/game_over = what the step returned.
 *   dup_* : append buffer for the rows only the reference's post-done re-flush quirk produces
 *   (reflush_done = 1: a done agent that is still reported learning re-emits its kept experience with
 *   every later step); dup_count int32 [2] = (appended, dropped), reset by the caller.
 *   ep_out float [ep_capacity,3] (world, total_reward, total_length), ep_count int32 [2].
 *   step < 0: use (and advance) the handle's device-side step counter (hipGraph replays). */
int cavoid_rollout_create(int64_t num_worlds, int32_t max_agents, int32_t obs_width, int32_t time_max, double discount,
                          int32_t reflush_done, int32_t ring_len, int device, cavoid_rollout **out);
void cavoid_rollout_destroy(cavoid_rollout *r);
int cavoid_rollout_reset(cavoid_rollout *r, void *stream);
int cavoid_rollout_push(cavoid_rollout *r, const float *prev_obs, const int32_t *actions, const float *values,
                        const float *rewards, const uint8_t *done, const uint8_t *game_over, int32_t step,
                        float *x, double *val, float *ret, uint8_t *act, int32_t *emit_t,
                        float *dup_x, float *dup_r, int32_t *dup_a, int32_t *dup_src, int32_t *dup_count, int64_t dup_capacity,
                        float *ep_out, int32_t *ep_count, int64_t ep_capacity, void *stream);

/* ---- the fused actor: K closed-loop steps of EVERY world in ONE launch ------------------------------------------------
 * observe -> predict_p_and_v -> select_action -> env.step -> Experience bookkeeping, K times (ProcessAgent.run_episode's loop body,
 * ga3c/GA3C/ProcessAgent.py:116-211, with the ThreadPredictor round trip of :89-96 and ThreadPredictor.py:61-75 inside it): per
 * tile of floor(64/N) worlds a workgroup runs cavoid_policy_forward's arithmetic on the tile's rows, draws the actions, steps
 * the tile's worlds (cavoid_step_autoreset's arithmetic) and records the step (cavoid_rollout_push's arithmetic) -- no kernel
 * boundary, no host, nothing between workgroups.  Results are bit-identical to calling those three entry points K times.
 *   obs_cur [W,N,1+D]: the observation to act on first; obs_next: a second buffer of the same shape -- step t reads one and
 *   writes the other, so after the call the current observation is in obs_cur when n_steps is even, in obs_next when odd.
 *   rewards / done / game_over / actions int32 [W,N] / values float [W,N]: per-step OUTPUTS, holding the LAST step's afterwards; they are
 *   never read (which rows need an action at a launch's first step is derived from the world state's own flags, so a cavoid_reset -- masked
 *   or not --, a cavoid_set_state or fresh buffers between two launches are all fine).
 *   The network runs only for the rows that still need an action -- what cavoid_rollout_active_rows lists: a learning agent (obs
 *   column 0) that was not done in the step that produced the observation, or whose world has just restarted; a finished agent
 *   waits for its world's last learning agent, a scripted agent acts by its own rule, the env ignores what either is given -- and
 *   the other rows are handed action 0 / value 0, like cavoid_policy_forward_rows' pass over that list (nothing reads them; the
 *   experience store's action entry of such a row is 0).  With reflush_done (the reference's re-flush quirk reads a done agent's
 *   value) and in cavoid_actor_run_mix every row runs.
 *   buffers: the experience store of cavoid_rollout_push.  The policy's launch counter and the rollout's step counter advance
 *   by n_steps on the device (the call can sit in a hipGraph).
 * Worlds with ORCA agents (rvo_enabled) and GEN v2 scenarios generated inside the step run over the env step's ORCA instantiation,
 * as in cavoid_step_autoreset.
 * CAVOID_EUNSUPPORTED (use the step-by-step entry points): holonomic dynamics, CAVOID_POLICY_F32 / a non-default
 * CAVOID_POLICY_PRODUCTS, rvo_enabled with so many agents per world (> 12) that the ORCA lines do not fit into the LDS the
 * policy lends the env step.  Frozen-network agents (CAVOID_POLICY_FROZEN_NET) act by a SECOND network: cavoid_actor_run_mix carries it;
 * this entry point refuses an env whose generator makes such agents (gen_frozen_fraction > 0 with gen_nonlearning_fraction > 0).  The
 * refusal looks at the GENERATOR's fractions only: frozen-network agents put into the worlds some other way -- cavoid_set_state, or a
 * scenario pool filled under another configuration -- are NOT detected here and would be handed the learner's sampled action; callers
 * that inject such agents must use cavoid_actor_run_mix.
 * Side effect: like every entry point that launches, the call leaves the CALLER's current HIP device set to the env's device. */
typedef struct cavoid_rollout_buffers {
    int32_t struct_size;             /* sizeof(cavoid_rollout_buffers) */
    int32_t reserved;
    float *x; double *val; float *ret; uint8_t *act; int32_t *emit_t;               /* the rings of cavoid_rollout_push */
    float *dup_x; float *dup_r; int32_t *dup_a; int32_t *dup_src; int32_t *dup_count; int64_t dup_capacity;
    float *ep_out; int32_t *ep_count; int64_t ep_capacity;
} cavoid_rollout_buffers;
int cavoid_actor_run(cavoid_env *env, cavoid_policy *policy, cavoid_rollout *rollout, const cavoid_rollout_buffers *buffers,
                     float *obs_cur, float *obs_next, float *rewards, uint8_t *done, uint8_t *game_over, int32_t *actions, float *values,
                     int32_t n_steps, int32_t greedy, void *stream);

/* cavoid_actor_run for the reference's TRAINING MIX (static / non-cooperative / RVO / CADRL agents around the learners,
 * ga3c/GA3C/checkpoints/RL/wandb/run-2018-backup/checkpoints/index.txt:1-3; the CADRL agent is a frozen network, ga3c/GA3C/Server.py:36):
 * `frozen` = a second cavoid_policy handle (same shapes, its own weights) that drives the CAVOID_POLICY_FROZEN_NET agents.  A tile that holds a
 * running frozen-network agent runs the forward pass once more on `frozen`'s weights and takes its ARGMAX for exactly those rows -- what the
 * step-by-step form does with cavoid_policy_rows + cavoid_policy_forward_rows(greedy) -- inside the same launch; other tiles skip it.
 * Runs over the env step's ORCA instantiation (ORCA agents and in-step box scenarios included).  Bit-identical to the step-by-step
 * form.  CAVOID_EUNSUPPORTED as cavoid_actor_run, and for a `frozen` handle of another inference form (CAVOID_POLICY_PRODUCTS / _F32). */
int cavoid_actor_run_mix(cavoid_env *env, cavoid_policy *policy, cavoid_policy *frozen, cavoid_rollout *rollout,
                         const cavoid_rollout_buffers *buffers, float *obs_cur, float *obs_next, float *rewards, uint8_t *done, uint8_t *game_over,
                         int32_t *actions, float *values, int32_t n_steps, int32_t greedy, void *stream);

/* cavoid_step_autoreset + cavoid_rollout_push as ONE launch: the env step of every world and the Experience bookkeeping of its slots
 * (ProcessAgent.py:149-211) -- the fused actor's env phase for actors whose policy is a launch of its own (frozen-network agents,
 * a caller-supplied policy).  obs_cur [W,N,1+D]: the observation `actions` / `values` (V(s_t)) were computed on; the step writes the
 * next one into obs_next (a different buffer) and rewards / done / game_over; buffers: the experience store of cavoid_rollout_push;
 * step as there (< 0: the handle's device-side counter, advanced by the call).  Bit-identical to the two calls it replaces.
 * CAVOID_EUNSUPPORTED: holonomic dynamics. */
int cavoid_step_push(cavoid_env *env, cavoid_rollout *rollout, const cavoid_rollout_buffers *buffers, const float *obs_cur, float *obs_next,
                     const int32_t *actions, const float *values, float *rewards, uint8_t *done, uint8_t *game_over, int32_t step, void *stream);

/* hand-over to the trainer (`training_q.put((x_, r_, a_))`, ProcessAgent.py:238): append the training rows (emit_t >= 0) of
 * the step blocks [step_lo, step_hi) to one batch -- out_x float [capacity, D], out_r float [capacity] (n-step returns),
 * out_a int32 [capacity], out_src int32 [capacity, 4] (world, agent, recorded-at, emitted-at; may be NULL),
 * out_count int32 [2] = (rows appended, rows dropped for lack of capacity), zeroed by the call.  Row order is unspecified.
 * mark_taken != 0 stamps the rows emit_t = -2 (they are not handed out again). */
int cavoid_rollout_compact(cavoid_rollout *r, int32_t step_lo, int32_t step_hi, int32_t mark_taken, const float *x, const float *ret,
                           const uint8_t *act, int32_t *emit_t, float *out_x, float *out_r, int32_t *out_a, int32_t *out_src,
                           int32_t *out_count, int64_t capacity, void *stream);

/* the (world, agent) slots that still need a policy output: learning agents that have not finished (an agent that is done
 * waits for its world's last learning agent; the env ignores its action -- ProcessAgent.py:149-211).  obs = the observation
 * the policy is about to act on [W,N,1+D]; done / game_over = the outputs of the step that produced it (after a reset:
 * game_over = 1 everywhere).  row_index int32 [W*N] receives the slot ids in unspecified order, row_count int32 [1] their
 * number; both stay on the device (cavoid_policy_forward_rows reads them there). */
int cavoid_rollout_active_rows(cavoid_rollout *r, const float *obs, const uint8_t *done, const uint8_t *game_over,
                               int32_t *row_index, int32_t *row_count, void *stream);

/* ---- fused policy inference (the actors' predict + select_action) ------------------------------------
 * Stands in for `NetworkVPCore.predict_p_and_v(x)` (ga3c/GA3C/NetworkVPCore.py:175-176) on the graph
 * `NetworkVP_rnn._create_graph` builds for MULTI_AGENT_ARCH 'RNN' (ga3c/GA3C/NetworkVP_rnn.py:50-67,103-105;
 * NetworkVPCore.py:66-75) and, optionally, `ProcessAgent.select_action` (ga3c/GA3C/ProcessAgent.py:89-103),
 * for every agent row of the batched env in ONE kernel on the matrix cores (float32 in, float32 accumulate).
 *   cavoid_policy_load   : hand over the variables in the reference checkpoint's layout (device pointers; dense
 *                          kernels [in,out], LSTM kernel [7+64, 4*64] in gate order i,j,f,o); they are re-packed
 *                          into MFMA fragment order inside the handle.  Call again whenever the trainer has
 *                          updated the weights.  avg/std = NN_INPUT_AVG_VECTOR / NN_INPUT_STD_VECTOR
 *                          (Config.py:64-71), both NULL = NORMALIZE_INPUT off.
 *   cavoid_policy_forward: x = first policy input (num_other_agents) of row 0, `row_stride` floats between
 *                          rows -- pass `obs + 1, 1 + D` to run straight on the env's observation tensor.
 *                          p_out float [rows, num_actions] (softmax_p incl. MIN_POLICY), v_out float [rows].
 *                          actions_out (nullable) int32 [rows]: greedy != 0 -> argmax p (PLAY_MODE /
 *                          EVALUATE_MODE), else one inverse-CDF sample per row from Philox4x32-10 keyed on
 *                          (seed, row, launch counter); the counter lives on the device (hipGraph replays). */
typedef struct cavoid_policy_weights {
    int32_t struct_size;             /* sizeof(cavoid_policy_weights) */
    float min_policy;                /* Config.MIN_POLICY */
    float forget_bias;               /* tf.contrib.rnn.LSTMCell default: 1.0 */
    int32_t with_backward;           /* != 0: also pack the transposed copies cavoid_policy_train_* needs */
    const float *avg, *std;          /* [5 + 7*max_other] or both NULL */
    const float *lstm_kernel, *lstm_bias;       /* rnn/lstm_cell/kernel [71,256], bias [256] */
    const float *layer1_kernel, *layer1_bias;   /* layer1 [68,256], [256] */
    const float *layer2_kernel, *layer2_bias;   /* layer2 [256,256], [256] */
    const float *fc1_kernel, *fc1_bias;         /* fullyconnected1 [256,256], [256] */
    const float *p_kernel, *p_bias;             /* logits_p [256,A], [A] */
    const float *v_kernel, *v_bias;             /* logits_v [256,1], [1] */
} cavoid_policy_weights;
int cavoid_policy_create(int32_t max_other, int32_t num_actions, int device, cavoid_policy **out);
void cavoid_policy_destroy(cavoid_policy *p);
int cavoid_policy_load(cavoid_policy *p, const cavoid_policy_weights *w, void *stream);
int cavoid_policy_seed(cavoid_policy *p, uint64_t seed, void *stream);
/* what the handle is (fixed at cavoid_policy_create: the environment switches CAVOID_POLICY_F32 / CAVOID_POLICY_PRODUCTS are read there,
 * never again): use_split = 1 when inference runs on the operand-split kernel (0: the float32-MFMA kernel); split_products = 16 for the
 * default float16 two-piece form, 3 / 4 / 5 for bf16 pieces.  RANGE LIMIT of the default form: a float16 piece saturates at +-65504, so a
 * weight beyond that (after the LSTM gate columns' scale of log2 e / 2 log2 e) is CLAMPED at load and the network then differs from the
 * reference's float32 predictor; clamped_weights = how many weights of the last cavoid_policy_load were (0 for every sane checkpoint;
 * reading it synchronises `stream`).  A caller that finds it non-zero should re-create the handle with CAVOID_POLICY_PRODUCTS=3 (bf16
 * pieces: float32's range) or CAVOID_POLICY_F32=1.  Inputs and hidden activations beyond +-65504 saturate in the kernel the same way.
 * Any out pointer may be NULL. */
int cavoid_policy_info(cavoid_policy *p, void *stream, int32_t *use_split, int32_t *split_products, int32_t *clamped_weights);
int cavoid_policy_forward(cavoid_policy *p, const float *x, int64_t rows, int64_t row_stride, float *p_out, float *v_out,
                          int32_t *actions_out, int32_t greedy, void *stream);
/* as cavoid_policy_forward, for the rows row_index[0 .. *row_count) only (device-side list and count: the launch geometry
 * does not depend on them, so the call can sit in a hipGraph); outputs of the other rows are left untouched */
int cavoid_policy_forward_rows(cavoid_policy *p, const float *x, int64_t rows, int64_t row_stride, const int32_t *row_index,
                               const int32_t *row_count, float *p_out, float *v_out, int32_t *actions_out, int32_t greedy,
                               void *stream);

/* ---- fused trainer pass (Server.train_model -> NetworkVPCore.train, ga3c/GA3C/Server.py:114-124, NetworkVPCore.py:71-100,178-187)
 * Forward + loss + the row-local part of the backward pass of the same network, for a batch of training rows
 * (x [rows, row_stride], y_r [rows] n-step returns, a_idx int32 [rows] actions taken), in two launches on the matrix
 * cores.  Needs cavoid_policy_load(with_backward = 1).  It leaves, in caller-owned buffers, the operand pair of every
 * weight-gradient GEMM (the layer's input rows and the gradient at its output rows); the caller forms X^T G with its GEMM
 * library, sums the bias gradients and takes the optimiser step:
 *   d fullyconnected1 = z2^T g3,  d layer2 = z1^T g2,  d layer1 (rows in packed order: 64 hidden, 4 host) = l1_in^T g1,
 *   d [logits_p | logits_v] = z3^T gh,  d lstm (rows: 64 hidden then 7 inputs; gate columns in packed order
 *   64w + 16 gate + u for unit 16w + u) = sum_t h_in[t]^T gl[t];   bias gradients (column sums of g3, g2, g1, gh, gl) come back in `db`.
 * All per-row buffers have `capacity_rows` rows (a multiple of 64, >= rows rounded up to 64); every one of them is
 * written by each call, rows past `rows` with zero gradients -- the GEMMs may run over all capacity_rows.  loss[0] = cost_p, loss[1] = cost_v (sums over rows, NetworkVPCore.py:71-100). */
typedef struct cavoid_policy_train_buffers {
    int32_t struct_size;             /* sizeof(cavoid_policy_train_buffers) */
    int32_t reserved;
    int64_t capacity_rows;
    float *z1, *z2, *z3;             /* [capacity_rows, 256] */
    float *l1_in;                    /* [capacity_rows, 72] */
    float *h_in;                     /* [max_other, capacity_rows, 72] */
    float *save;                     /* [capacity_rows / 64, max_other, 16, 256, 8] */
    float *gh;                       /* [capacity_rows, 16] */
    float *loss;                     /* [2] */
    float *g1, *g2, *g3;             /* [capacity_rows, 256] */
    float *gl;                       /* [max_other, capacity_rows, 256] */
    float *db;                       /* [1040] bias gradients: lstm 256 (packed gate order), layer1, layer2, fullyconnected1 256 each,
                                        heads 16 (A logits, value, padding) */
} cavoid_policy_train_buffers;
int cavoid_policy_train(cavoid_policy *p, const float *x, int64_t rows, int64_t row_stride, const float *y_r, const int32_t *a_idx,
                        float beta, float log_epsilon, const cavoid_policy_train_buffers *buffers, void *stream);

/* kernel timing helper: HIP events recorded on `stream` around the launches of the calls made
 * between begin and end; end synchronises and returns elapsed milliseconds */
int cavoid_timer_begin(cavoid_env *env, void *stream);
int cavoid_timer_end(cavoid_env *env, void *stream, float *elapsed_ms);

#ifdef __cplusplus
}
#endif
#endif /* CAVOID_H */
