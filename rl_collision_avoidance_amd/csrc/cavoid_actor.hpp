// cavoid_actor.hpp -- actor_kernel<N, RVO>: the GA3C actor's CLOSED loop, K env steps in ONE launch.
//
// What one reference ProcessAgent does per env step (ga3c/GA3C/ProcessAgent.py:116-211): observe -> predict (RPC to
// ThreadPredictor, ThreadPredictor.py:61-75) -> select_action -> env.step -> Experience bookkeeping / n-step returns.  Here, for
// a tile of floor(64/N) worlds x N agents = up to 64 policy rows, one workgroup of four wavefronts runs that loop K times
// without leaving the GPU or the launch:
//
//   all 4 wavefronts   NetworkVP_rnn forward of the tile's 64 rows on the matrix cores (policy_split_tile, the statements of
//                      policy_forward_split_kernel) + the Philox action draw  -> actions / values of the tile
//   wavefront 0        env.step of the tile (env_tile<N, MODE_STEP_AUTORESET>: the statements of env_kernel, LDS carved out of the
//                      policy's activation planes, which are idle now) -> obs(t+1), reward, done, game_over;
//                      then the experience bookkeeping of the tile's slots (rollout_push_slot: the statements of
//                      rollout_push_kernel) and the episode log
//   wavefronts 1..3    meanwhile: the step's state rows -> the time-major experience ring (rollout_copy_rows)
//
// Worlds are independent and a tile's policy rows are exactly its own agents, so NOTHING crosses workgroups: no grid barrier,
// no kernel boundary between policy, env and bookkeeping (five launches and ~8 dependent boundaries per env step in the
// hipGraph form).  Two workgroups share a CU; their phases drift apart, so one tile's env step (one busy wavefront) runs under
// the other tile's matrix phase.  State between the phases travels through global memory (L2-resident; same-CU visibility
// after the workgroup barrier), exactly as between the launches of the hipGraph form: every value is computed by the same
// statements in the same order, so trajectories, experience rings and episode logs are bit-identical to that form
// (tests/test_gpu_actor.py).
#pragma once
#include "cavoid_kernels.hpp"
#include "cavoid_quad.hpp"
#include "cavoid_policy_split.hpp"
#include "cavoid_rollout.hpp"

namespace cavoid {

#ifndef CAVOID_COPY_U
#define CAVOID_COPY_U 32        // loads in flight per lane of step_push_kernel's row-copy wavefronts
#endif
constexpr int kActorQuadMaxAgents = 4;   // the cooperative env step inside the fused actor kernel: instantiated up to this many agents per world
struct ActorIO {
    float *obs[2];            // [W,N,1+D] each: step t acts on obs[t & 1]; the env writes the next observation into obs[(t+1) & 1]
    float *rewards;           // [W,N]   the env's step outputs; after the launch they hold the LAST step's
    uint8_t *done;            // [W,N]
    uint8_t *game_over;       // [W]
    int32_t *actions;         // [W,N]   select_action's choice (hand-over policy -> env inside a step; last step's afterwards)
    float *values;            // [W,N]   V(s_t) (the bootstrap of a flush)
    int32_t *rollout_step;    // device-side step index of the experience store (advanced by actor_finish_kernel)
    int32_t n_steps;
    int32_t greedy;           // PLAY_MODE / EVALUATE_MODE: argmax instead of sampling
    int32_t quad;             // the tile's env step by all four wavefronts (cavoid_quad.hpp) instead of wavefront 0 alone -- the host sets it where that
                              // form carries the configuration (actor_run); same results
};

// LDS the env step of a tile needs inside the (idle) activation planes: the action table + one wavefront's staging arrays and
// obs tile
__host__ __device__ inline size_t actor_env_lds_bytes(int n_agents, int tile_floats, int rvo_floats = 0) {
    return (size_t)(lds_floats_block() + lds_floats_fixed(n_agents) + tile_floats + rvo_floats) * sizeof(float);
}

// env.step of ONE tile by ONE wavefront + the Experience bookkeeping of the tile's slots (the env step's lane mapping: one lane
// per slot): the statements of env_kernel<MODE_STEP_AUTORESET>, rollout_push_kernel and rollout_episode_kernel.  obs_t: the
// observation the actions were chosen on; the step writes the next one into obs_n.
template <int N, bool RVO, bool EARLY>
__device__ __forceinline__ void actor_env_push_tile(const KCfg &c, const KState &s, const PoolRec *pool, const RolloutCfg &rc, const RolloutState &rs,
                                                    const RolloutIO &rio_arg, const ActorIO &io, const float *obs_t, float *obs_n, double *lds_tab,
                                                    float *wbase, int lane, int64_t tile, int32_t step, int blk, int *live_next = nullptr) {
    const int wpw = c.wpw, ow = c.width;
    const int64_t w0 = tile * wpw;
    KIO k{};
    k.actions = io.actions; k.obs = obs_n; k.rew = io.rewards; k.done = io.done; k.game_over = io.game_over;
    k.obs_stride = ow; k.n_steps = 1;
    StepOut so{0.0f, true, false, false};
    const int lw = lane / N, i = lane - lw * N;
    const int64_t w = w0 + lw, a = w * N + i;
    const bool in_range = lane < wpw * N && w < c.num_worlds;
    // The bookkeeping's first trip to memory -- the slot's counters, is_learning of the state acted on, the policy's value and action:
    // nothing the env step changes -- is issued in front of the env step and lands under it.  (Measured and dropped: ALSO its second
    // trip, the pending rewards, issued under the env step through a hook in env_tile; every lane then loads its pending rewards
    // every step instead of the flushing lanes once per T_max steps: actor loop 41.4 -> 49.8 us per env step, step_push 15.3 -> 23.3 us.)
    // EARLY = false: the instantiations whose env step has no eight registers to spare across it (the fused actor kernel's 256-register
    // budget with many agents per world) keep the reads behind the step.
    RolloutSlot slot_in{0, 0, false, 0.0};
    float learn_f = 0.0f, value = 0.0f;                                // is_learning of the state acted on (ProcessAgent.py:130)
    int action = 0;
    auto first_trip = [&]() {
        slot_in = rollout_slot_load(rs, a, in_range);
        if (in_range) { learn_f = obs_t[a * ow]; value = io.values[a]; action = io.actions[a]; }
    };
    if (EARLY) first_trip();
    env_tile<N, MODE_STEP_AUTORESET, RVO>(c, s, pool, k, lds_tab, wbase, lane, tile, &so);
    POLICY_STAMP(0);                                       // (trace build) env.step of the tile done, its stores issued
    if (live_next) {        // the rows that need an action at the NEXT step (cavoid_rollout_active_rows' predicate on what this step produced), for the
        //                     fused kernel's next policy pass: in LDS, so that the pass need not go to memory for the flags
        const unsigned long long m = __ballot(in_range && so.learning_next && (so.game_over || !so.done));
        if (lane == 0) { live_next[0] = (int)(uint32_t)m; live_next[1] = (int)(uint32_t)(m >> 32); }
    }
    if (!EARLY) first_trip();
    const bool learning = in_range && learn_f > 0.5f;
    const int base = lane < wpw * N ? lw * N : 0;
    const int n_learning = __popcll(__ballot(learning) & (((1ull << N) - 1ull) << base));
    RolloutIO rio = rio_arg;
    rollout_push_slot(rc, rs, rio, a, in_range ? w : 0, i, in_range, learning, n_learning, so.done, so.game_over, so.reward, value,
                      action, step, blk, slot_in);
    // episode_log_q.put: the totals above were accumulated with atomics by this wavefront's own lanes -- drain them;
    // rollout_close_episode reads the sums at the cache the atomics went to
    if (__ballot(in_range && so.game_over) != 0ull) {
        __builtin_amdgcn_s_waitcnt(0);                       // (vmcnt 0: the atomics have been performed at the L2)
        if (in_range && i == 0 && so.game_over) rollout_close_episode(rc, rs, rio, w);
    }
}

// The same by the workgroup's four wavefronts (io.quad): quad_env_tile's cooperative step -- wavefront 0 hosts it and then does the bookkeeping
// exactly as above, wavefronts 1..3 take the neighbours' chains, the ranking, their slots of the rows and their share of the flush.  Contains
// workgroup barriers: every thread of the workgroup calls it.  smem: the (idle) activation planes.
template <int N, bool EARLY>
__device__ __forceinline__ void actor_env_push_tile_quad(const KCfg &c, const KState &s, const PoolRec *pool, const RolloutCfg &rc, const RolloutState &rs,
                                                         const RolloutIO &rio_arg, const ActorIO &io, const float *obs_t, float *obs_n,
                                                         unsigned char *smem, int role, int lane, int64_t tile, int32_t step, int blk, int *live_next) {
    const int wpw = c.wpw, ow = c.width;
    const int64_t w0 = tile * wpw;
    KIO k{};
    k.actions = io.actions; k.obs = obs_n; k.rew = io.rewards; k.done = io.done; k.game_over = io.game_over;
    k.obs_stride = ow; k.n_steps = 1;
    StepOut so{0.0f, true, false, false};
    const int lw = lane / N, i = lane - lw * N;
    const int64_t w = w0 + lw, a = w * N + i;
    const bool in_range = lane < wpw * N && w < c.num_worlds;
    RolloutSlot slot_in{0, 0, false, 0.0};
    float learn_f = 0.0f, value = 0.0f;
    int action = 0;
    auto first_trip = [&]() {
        slot_in = rollout_slot_load(rs, a, in_range);
        if (in_range) { learn_f = obs_t[a * ow]; value = io.values[a]; action = io.actions[a]; }
    };
    if (role == 0 && EARLY) first_trip();
    quad_env_tile<N>(c, s, pool, k, smem, role, lane, tile, &so);
    if (role != 0) return;
    POLICY_STAMP(0);
    if (live_next) {
        const unsigned long long m = __ballot(in_range && so.learning_next && (so.game_over || !so.done));
        if (lane == 0) { live_next[0] = (int)(uint32_t)m; live_next[1] = (int)(uint32_t)(m >> 32); }
    }
    if (!EARLY) first_trip();
    const bool learning = in_range && learn_f > 0.5f;
    const int base = lane < wpw * N ? lw * N : 0;
    const int n_learning = __popcll(__ballot(learning) & (((1ull << N) - 1ull) << base));
    RolloutIO rio = rio_arg;
    rollout_push_slot(rc, rs, rio, a, in_range ? w : 0, i, in_range, learning, n_learning, so.done, so.game_over, so.reward, value,
                      action, step, blk, slot_in);
    if (__ballot(in_range && so.game_over) != 0ull) {
        __builtin_amdgcn_s_waitcnt(0);
        if (in_range && i == 0 && so.game_over) rollout_close_episode(rc, rs, rio, w);
    }
}

// RVO = true: the env step's ORCA instantiation (scripted RVO agents; it is also the one that generates box scenarios inside the
// step) -- its line scratch comes out of the activation planes too (cavoid_actor_rvo.hip)
// FROZEN = true: the tile may hold frozen-network agents (scripted policy 4: NON-learning agents driven by a second, frozen
// NetworkVP_rnn -- the GA3C-CADRL agent mechanism of the reference's training mix, ga3c/GA3C/Server.py:36, index.txt:1-3).  After the
// learner's pass a tile with at least one RUNNING policy-4 agent runs the same forward pass once more on the frozen network's weights
// (`fz`) and takes its argmax for exactly those rows -- what BatchedRollout.act does with cavoid_policy_rows + cavoid_policy_forward_rows
// in the step-by-step form; tiles without such agents skip it (one ballot + a workgroup barrier per step).
template <int N, bool RVO, bool FROZEN = false>
__global__ void __launch_bounds__(256, 2) actor_kernel(const KCfg c, const KState s, const PoolRec *pool, const SplitArgs sa, const SplitArgs fz,
                                                       const RolloutCfg rc, const RolloutState rs, const RolloutIO rio_arg, const ActorIO io) {
    extern __shared__ __attribute__((aligned(16))) unsigned char planes[];      // the policy's activation planes ...
    float *len_f = reinterpret_cast<float *>(planes + 2 * kSpPlaneB);
    int *wave_max = reinterpret_cast<int *>(len_f + 64) + 64;
    int &ticket = wave_max[4];
    double *lds_tab = reinterpret_cast<double *>(planes);                        // ... lent to the env step while they are idle
    float *wbase = reinterpret_cast<float *>(planes) + lds_floats_block();

    const PolicyArgs &p = sa.p;
    const int tid0 = threadIdx.x;
    const int wpw = c.wpw, A = p.num_actions, ow = c.width;
    const int64_t tile = blockIdx.x, w0 = tile * wpw, a0 = w0 * N;
    int64_t worlds_here = c.num_worlds - w0;
    worlds_here = worlds_here > wpw ? wpw : (worlds_here < 0 ? 0 : worlds_here);
    const int rows = (int)worlds_here * N;                   // policy rows = agent slots of this tile
    const int32_t step0 = *io.rollout_step, pstep0 = *p.step_counter;

    if (tid0 == 0) {                                         // arrival parity on the CU -> static priority (see cavoid_policy.hpp)
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        const uint32_t key = ((xcc & 15u) << 8) | ((hw >> 8) & 0xFFu);
        ticket = (int)atomicAdd(p.cu_tickets + key, 1u);
    }
    __syncthreads();
    if (ticket & 1) __builtin_amdgcn_s_setprio(1);

#pragma unroll 1
    for (int t = 0; t < io.n_steps; ++t) {
        // Everything derived from the thread id is loop invariant, and the compiler would hoist all of it -- every weight-fragment
        // and LDS address of the GEMM loops, every lane-to-world index of the env step -- out of the step loop into registers live
        // across the whole body (measured: 256 VGPRs + 1.2 KB of scratch per lane).  Re-materialising the id per step keeps each
        // phase's register footprint that of its stand-alone kernel.
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int wave_in_block = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        const float *obs_t = io.obs[t & 1];
        float *obs_n = io.obs[(t + 1) & 1];
        const int32_t step = step0 + t;
        const int blk = step % rc.ring_len;
        POLICY_STAMP(13);                                    // (trace build, tools/trace_actor.py) the step begins

        // ---- predict_p_and_v + select_action for the tile's rows, read in place from the observation the env wrote -----------
        {
            const float *src = obs_t + a0 * ow + 1;            // column 0 (is_learning) is not a network input
            auto load = [&](int r, int k) -> float { return src[(int64_t)r * ow + k]; };
            auto emit = [&](int trow, int g, const float (&pj)[4], const f32x4 &logit) {
                const int64_t row = a0 + trow;
                const int action = split_select_action(pj, g, lane, A, io.greedy != 0, row, pstep0 + t, p.seed_lo, p.seed_hi);
                if (trow < rows) {
                    if (g == 0) io.actions[row] = action;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * g + r == A) io.values[row] = logit[r];
                }
            };
            if constexpr (!FROZEN) {
                // Only the rows that still need an action go through the network (policy_split_tile, COMPACT): a learning agent that has not
                // finished -- exactly the rows cavoid_rollout_active_rows lists for the step-by-step path (cavoid_rollout.hpp: obs column 0
                // set, and the agent not done in the step that produced this observation unless its world has just restarted).  A finished
                // agent waits for its world's last learning agent and a scripted agent acts by its own rule: the env ignores what either is
                // given, the bookkeeping never reads it.  With the reference's re-flush quirk a done agent's value IS read: every row runs.
                bool mine = lane < rows;
                if (rc.reflush_done == 0) {
                    if (t == 0) {
                        // (uniform) the first step of a launch: the same predicate from the WORLD STATE -- a learning agent whose state carries no
                        // terminal flag.  After a step that is what (game_over || !done) of that step says (a restarted world's agents are
                        // fresh; elsewhere done == a terminal flag), and unlike the caller's done / game_over buffers the state cannot be stale:
                        // a (masked) cavoid_reset, cavoid_set_state or freshly allocated output buffers between two launches change nothing
                        // here -- done / game_over stay pure OUTPUTS of this entry point, as include/cavoid.h documents them
                        if (mine) mine = obs_t[(a0 + lane) * ow] > 0.5f && (s.flags[a0 + lane] & CAVOID_F_DONE_MASK) == 0u;
                    } else {                                 // the later ones the mask the tile's own env step left in LDS
                        const unsigned long long m = (unsigned long long)(uint32_t)wave_max[12] | ((unsigned long long)(uint32_t)wave_max[13] << 32);
                        mine = (m >> lane) & 1ull;
                    }
                }
                // (a row that needs no action is handed action 0 / value 0, what the step-by-step path's row-list pass leaves there)
                if (wave_in_block == 0 && lane < rows && !mine) { io.actions[a0 + lane] = 0; io.values[a0 + lane] = 0.0f; }
                const unsigned long long live_mask = __ballot(mine);        // (every wavefront evaluates the same 64 rows: no trip through LDS)
                const int n_rt = (__popcll(live_mask) + 15) >> 4;           // row tiles the live rows fill: one instantiation of the pass each
                int *rmap = reinterpret_cast<int *>(len_f + 64);
                if (n_rt == 4) policy_split_tile<kSpDefaultProducts, 4>(sa, planes, len_f, wave_max, rows, tid, load, emit, live_mask, rmap);
                else if (n_rt == 3) policy_split_tile<kSpDefaultProducts, 3>(sa, planes, len_f, wave_max, rows, tid, load, emit, live_mask, rmap);
                else if (n_rt >= 1) policy_split_tile<kSpDefaultProducts, 2>(sa, planes, len_f, wave_max, rows, tid, load, emit, live_mask, rmap);
                // (n_rt == 0: nobody in the tile needs an action)
            } else {
                policy_split_tile<kSpDefaultProducts>(sa, planes, len_f, wave_max, rows, tid, load, emit);
            }
            if constexpr (FROZEN) {
                // the rows whose agent is a running frozen-network agent (the env state's flags: this step has not run yet)
                bool mine = false;
                if (tid < rows) {
                    const uint32_t f = s.flags[a0 + tid];
                    mine = (f & CAVOID_F_PRESENT) && ((f >> CAVOID_F_POLICY_SHIFT) & CAVOID_F_POLICY_MASK) == (uint32_t)CAVOID_POLICY_FROZEN_NET &&
                           (f & CAVOID_F_DONE_MASK) == 0u;
                }
                __syncthreads();                             // every wavefront is done with the learner's pass (planes, wave_max)
                if (tid < 64) { const unsigned long long m = __ballot(mine); if (lane == 0) { wave_max[12] = (int)(uint32_t)m; wave_max[13] = (int)(uint32_t)(m >> 32); } }
                __syncthreads();
                const unsigned long long frozen_rows = (unsigned long long)(uint32_t)wave_max[12] | ((unsigned long long)(uint32_t)wave_max[13] << 32);
                if (frozen_rows != 0ull) {                   // (uniform over the workgroup)
                    auto emit_fz = [&](int trow, int g, const float (&pj)[4], const f32x4 &) {
                        const int action = split_select_action(pj, g, lane, A, true, a0 + trow, 0, 0u, 0u);       // argmax: no random draw
                        if (trow < rows && g == 0 && ((frozen_rows >> trow) & 1ull)) io.actions[a0 + trow] = action;
                    };
                    policy_split_tile<kSpDefaultProducts>(fz, planes, len_f, wave_max, rows, tid, load, emit_fz);
                }
            }
        }
        __syncthreads();                                     // the tile's actions / values are in memory; the planes are idle
        POLICY_STAMP(14);                                    // policy pass + barrier done

        if (!RVO && N <= kActorQuadMaxAgents && io.quad) {   // (uniform)
            // ---- the step's state rows -> the experience store by wavefronts 1..3, then env.step of the tile by all four (cavoid_quad.hpp) and
            //      the Experience bookkeeping of its slots by wavefront 0
            if (wave_in_block != 0) rollout_copy_rows(rc, obs_t, rio_arg.x, a0, rows, blk, tid - 64, 192);
            actor_env_push_tile_quad<(N <= kActorQuadMaxAgents ? N : 1), true>(c, s, pool, rc, rs, rio_arg, io, obs_t, obs_n, planes, wave_in_block, lane, tile,
                                                                              step, blk, FROZEN ? nullptr : wave_max + 12);
            if (wave_in_block == 0) POLICY_STAMP(15);
        } else if (wave_in_block == 0) {
            // ---- env.step of the tile, then the Experience bookkeeping of its slots ---------------------------------------
            actor_env_push_tile<N, RVO, (N <= (RVO ? 9 : 13))>(c, s, pool, rc, rs, rio_arg, io, obs_t, obs_n, lds_tab, wbase, lane, tile, step, blk,
                                                               FROZEN ? nullptr : wave_max + 12);
            POLICY_STAMP(15);                                // env step + bookkeeping of the tile done (stores issued)
        } else {
            // ---- meanwhile: the step's state rows -> the time-major experience store -----------------------------------------
            rollout_copy_rows(rc, obs_t, rio_arg.x, a0, rows, blk, tid - 64, 192);
        }
        __syncthreads();                                     // obs(t+1), the world state and the slot state are in memory
        POLICY_STAMP(6);                                     // the step ends
#ifdef CAVOID_TRACE
        if (tid0 == 0 && g_pol_trace)
            g_pol_trace[(size_t)blockIdx.x * 16 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                                      ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
#endif
    }
}

// One auto-reset step of every world + its Experience bookkeeping in ONE launch (cavoid_step_push): the env phase of the loop
// above as a kernel of its own -- one wavefront per tile, env_kernel's launch shape -- for actors whose policy runs as its own
// launch (frozen-network agents, row lists, a caller-supplied policy): three launches (env, push, episode log) become one, the
// step's rewards / done flags go from the env step to the bookkeeping in registers.  Grid: (tiles / wavefronts per block, 2) --
// y = 0 runs env step + bookkeeping, y = 1 copies the step's state rows into the experience store at the same time.
template <int N, bool RVO>
__global__ void __launch_bounds__(256) step_push_kernel(const KCfg c, const KState s, const PoolRec *pool, const RolloutCfg rc, const RolloutState rs,
                                                        const RolloutIO rio_arg, const ActorIO io, const int32_t step_arg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave_in_block = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block;
    const int32_t step = step_arg >= 0 ? step_arg : *io.rollout_step;
    const int blk = step % rc.ring_len;
    if (blockIdx.y == 1) {
        // the second half of the grid: the step's state rows -> the time-major experience store, by a wavefront of its own per
        // tile (the rows are obs_cur, complete before the launch; nothing here depends on the env step the first half runs --
        // in sequence on one wavefront the copy was a third of the launch's chain)
        const int64_t a0 = tile * c.wpw * N;
        int64_t worlds_here = c.num_worlds - tile * c.wpw;
        worlds_here = worlds_here > c.wpw ? c.wpw : (worlds_here < 0 ? 0 : worlds_here);
        rollout_copy_rows<CAVOID_COPY_U>(rc, io.obs[0], rio_arg.x, a0, (int)worlds_here * N, blk, lane, 64);
        return;
    }
    const int tile_need = (c.tile_rows * c.width + 3) & ~3;
    const int tile_floats = tile_need > c.park_floats ? tile_need : c.park_floats;
    const int per_wave_floats = lds_floats_fixed(N) + tile_floats + c.rvo_lds_floats;
    double *lds_tab = reinterpret_cast<double *>(smem);
    float *wbase = reinterpret_cast<float *>(smem) + lds_floats_block() + (size_t)wave_in_block * per_wave_floats;
    actor_env_push_tile<N, RVO, true>(c, s, pool, rc, rs, rio_arg, io, io.obs[0], io.obs[1], lds_tab, wbase, lane, tile, step, blk);
}

#ifdef CAVOID_ACTOR_KERNELS      /* the non-template kernel is compiled by cavoid_actor.hip only */
// after the actor launch: advance the two device-side counters the next launch (or a hipGraph replay of this one) starts from
__global__ void actor_finish_kernel(int32_t *rollout_step, int32_t *policy_step, int32_t n_steps) {
    *rollout_step += n_steps;
    if (policy_step) *policy_step += n_steps;
}
#endif

}  // namespace cavoid

