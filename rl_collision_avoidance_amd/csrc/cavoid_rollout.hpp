// cavoid_rollout.hpp -- gfx950 kernels of the batched GA3C actor bookkeeping: what
// ProcessAgent.run_episode / _accumulate_rewards / convert_to_nparray do per actor process
// (/root/reference/ga3c/GA3C/ProcessAgent.py:54-87,105-211; rows R3-R5 of SURVEY.md section 8a),
// for every (world, agent) slot at once, on the device, with no queue hop.
//
// Layout: TIME-MAJOR experience store.  The training batch is a ring of `ring_len` step blocks
//   x   float  [ring_len][slots][D]   state the policy acted on at that step (copied coalesced, once)
//   val double [ring_len][slots]      single-step reward of the step
//   ret float  [ring_len][slots]      the n-step return the row was emitted with (y_r of the training row)
//   act u8     [ring_len][slots]      action index
//   emit_t i32 [ring_len][slots]      -1 = pending / nothing recorded; >= 0 = a training row, emitted at that step
// An experience is written where it will be trained from; a flush only walks the slot's <= T_max+1
// pending rewards (one load burst, the recurrence in registers, one store burst).  The reference
// overwrites Experience.reward with the return in place; no overwritten value is ever read again
// except the kept seed's, whose overwrite is the identity (bootstrap 0: R = 0*gamma + r), so the
// store keeps raw rewards in `val` and the emitted returns in `ret`.  The
// reference's "keep the last experience as the seed of the next chunk" costs nothing here: the
// seed simply stays pending.  Closed blocks (older than T_max+1 steps) are compacted by the host.
// Only the reference's post-done re-flush quirk produces rows that are not 1:1 with (step, slot);
// those rare duplicates go to a small append buffer.
//
// One lane per (world, agent) slot, flat index a = w*N + i: every per-slot array is coalesced.
// HBM-bound bookkeeping: no MFMA, no LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cavoid {

struct RolloutCfg {
    int64_t num_slots;       // W*N
    int32_t max_agents;      // N
    int32_t obs_width;       // 1 + D (column 0 = is_learning)
    int32_t time_max;        // T_max (Config.TIME_MAX, Config.py:104)
    int32_t reflush_done;    // 1: reference behaviour -- a done agent keeps flushing 2-long chunks
    int32_t ring_len;        // step blocks in the experience store (> T_max + 1)
    int32_t _pad;
    double discount;         // Config.DISCOUNT (Config.py:103)
    int64_t dup_capacity;    // rows the duplicate (re-flush) buffer can hold
    int64_t ep_capacity;     // episode-log records
};

struct RolloutState {
    uint8_t *len;            // [slots] pending experiences
    uint8_t *since_flush;    // [slots] time_counts[i]
    uint8_t *trained;        // [slots] which_agents_done_and_trained[i]
    double *score;           // [slots] reward_sum_logger[i]
    double *ep_reward;       // [W] total_reward of the running episode (ProcessAgent.py:236)
    int32_t *ep_length;      // [W] total_length (:237)
    int32_t *step_counter;   // [1] device-side step index, used (and advanced) when the host passes step < 0
};

struct RolloutIO {
    const float *prev_obs;   // [slots][1+D]  what the policy acted on (Environment.previous_state + col 0)
    const int32_t *actions;  // [slots]
    const float *values;     // [slots]
    const float *rewards;    // [slots]
    const uint8_t *done;     // [slots]
    const uint8_t *game_over;  // [W]
    int32_t step;            // global step index; < 0: use the device-side counter (graph replays)
    float *x;                // [ring_len][slots][D]
    double *val;             // [ring_len][slots]
    float *ret;              // [ring_len][slots]
    uint8_t *act;            // [ring_len][slots]
    int32_t *emit_t;         // [ring_len][slots]
    float *dup_x;            // [dup_capacity][D]   re-flushed duplicates (reference quirk only)
    float *dup_r;            // [dup_capacity]
    int32_t *dup_a;          // [dup_capacity]
    int32_t *dup_src;        // [dup_capacity][4]  world, agent, recorded-at step, emitted-at step
    int32_t *dup_count;      // [2]  rows appended, rows dropped for lack of capacity
    float *ep_out;           // [ep_capacity][3]  world, total_reward, total_length
    int32_t *ep_count;       // [2]  records appended, dropped
};

constexpr int kMaxRing = 32;   // T_max + 1 <= 32: the backward pass runs in registers (else a serial fallback)

// the step's state rows -> x[blk]: a coalesced sweep of `rows` contiguous slots starting at slot a0 by the `nlanes` lanes
// (lane = 0 .. nlanes-1) of the caller's group (a wavefront; the fused actor kernel uses three)
// U loads in flight per lane, then U stores: every round is a dependent trip to memory for the copying wavefront (a wavefront alone
// with a tile of 10-agent worlds: 4080 values = 8 rounds at U = 8)
template <int U = 8>
__device__ __forceinline__ void rollout_copy_rows(const RolloutCfg &c, const float *prev_obs, float *x, int64_t a0, int rows, int blk,
                                                  int lane, int nlanes) {
    const int D = c.obs_width - 1;
    const int total = rows * D;
    const uint32_t inv_d = (uint32_t)((1ull << 32) / (uint32_t)D) + 1u;   // idx / D by multiply-shift (idx < 2^16)
    const float *src = prev_obs + a0 * c.obs_width;
    float *dst = x + ((int64_t)blk * c.num_slots + a0) * D;
    for (int i0 = lane; i0 < total; i0 += nlanes * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = i0 + nlanes * u;
            const int r = (int)(((uint64_t)(uint32_t)idx * inv_d) >> 32), q = idx - r * D;
            v[u] = idx < total ? src[r * c.obs_width + 1 + q] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = i0 + nlanes * u;
            if (idx < total) dst[idx] = v[u];
        }
    }
}

// One slot's bookkeeping for one env step (ProcessAgent.run_episode's body for agent i of world w, :149-211): the append, the
// flush rule, the backward n-step return, the episode totals.  Inputs are VALUES (the stand-alone kernel reads them from the
// step's output tensors, the fused actor kernel has them in registers); the rings are written in place.
// (the slot's counters are VALUES too: rollout_slot_load reads them -- the fused kernels issue that read in front of the env step,
//  so that the bookkeeping's first trip to memory runs under it)
struct RolloutSlot { int len, since; bool trained; double score; };
__device__ __forceinline__ RolloutSlot rollout_slot_load(const RolloutState &s, int64_t a, bool in_range) {
    RolloutSlot q{0, 0, false, 0.0};
    if (in_range) { q.len = s.len[a]; q.since = s.since_flush[a]; q.trained = s.trained[a] != 0; q.score = s.score[a]; }
    return q;
}
__device__ __forceinline__ void rollout_push_slot(const RolloutCfg &c, const RolloutState &s, const RolloutIO &io, int64_t a, int64_t w, int i,
                                                  bool in_range, bool learning, int n_learning, bool done, bool over, float reward,
                                                  float value, int action, int32_t step, int blk, const RolloutSlot &slot_in) {
    const int D = c.obs_width - 1, L = c.time_max + 1, RL = c.ring_len;
    const int64_t slots = c.num_slots;
    int len = slot_in.len, since = slot_in.since;
    bool trained = slot_in.trained;
    double score = slot_in.score;
    const bool was_trained = trained;

    int n_rows = 0;            // rows of the main chunk
    bool leftover = false;     // + one separate 1-row chunk
    int count = 0;             // entries the backward pass covers
    bool flush = false;
    const bool frozen = !c.reflush_done && trained;          // cleaned mode: a trained agent records nothing more
    const bool record = learning && !frozen;
    const int64_t cur = (int64_t)blk * slots + a;
    if (in_range) {
        // ---- append (Experience(previous_state[0,i,:], action, prediction, reward, done), :172-177) ----
        io.val[cur] = (double)reward;
        io.act[cur] = (uint8_t)action;
        io.emit_t[cur] = -1;                                 // pending (or: nothing recorded at this step)
    }
    if (record) {
        score += (double)reward;
        len += 1;
        // ---- flush rule (:186, Python precedence: done OR (count == T_max AND NOT trained)) ------------
        flush = done || (since == c.time_max && !trained);
        if (flush) {
            if (len == 1) { n_rows = 1; count = 0; }
            else if (done && len == L) { leftover = true; n_rows = len - 1; count = len - 1; }
            else if (done) { n_rows = len; count = len; }
            else { n_rows = len - 1; count = len - 1; }
        }
    }
    const int mine = n_rows + (leftover ? 1 : 0);
    const int newest = len - 1;                              // the entry appended in this launch
    // pending entry k (0 = oldest) was recorded at step  step - newest + k
    // the reference's quirk: once trained, the kept (already emitted) seed is emitted AGAIN with every later
    // step's experience -- such a duplicate cannot live in the (step, slot) store
    const bool dup0 = flush && was_trained && len >= 2;

    if (flush) {
        // ---- n-step return, newest to oldest, overwriting the stored rewards (:54-79) ------------------
        double R = done ? 0.0 : (double)value;
        if (done) trained = true;
        double r0 = 0.0;                                      // return of the oldest pending entry (duplicate path)
        if (L <= kMaxRing) {
            // all pending rewards in flight at once, then the recurrence in registers
            double rr[kMaxRing];
            const int b0 = (step - newest) % RL;                 // block of the oldest pending entry; k-th: b0 + k (mod RL)
#pragma unroll
            for (int k = 0; k < kMaxRing; ++k) {
                const int bk = b0 + k >= RL ? b0 + k - RL : b0 + k;
                rr[k] = (k < newest) ? io.val[(int64_t)bk * slots + a] : (double)reward;
            }
#pragma unroll
            for (int k = kMaxRing - 1; k >= 0; --k)
                if (k < count) { R = c.discount * R + rr[k]; rr[k] = R; }
#pragma unroll
            for (int k = 0; k < kMaxRing; ++k) {
                const int bk = b0 + k >= RL ? b0 + k - RL : b0 + k;
                const int64_t e = (int64_t)bk * slots + a;
                if (k < mine && !(dup0 && k == 0)) {            // convert_to_nparray (:82-87): the row goes live
                    io.ret[e] = (float)rr[k]; io.emit_t[e] = step;
                }
            }
            r0 = rr[0];
        } else {
            for (int k = count - 1; k >= 0; --k) {
                const int64_t e = (int64_t)((step - newest + k) % RL) * slots + a;
                R = c.discount * R + (k == newest ? (double)reward : io.val[e]);
                if (k == 0) r0 = R;
                if (!(dup0 && k == 0)) { io.ret[e] = (float)R; io.emit_t[e] = step; }
            }
            for (int k = count; k < mine; ++k) {                  // rows outside the pass keep their raw reward
                const int64_t e = (int64_t)((step - newest + k) % RL) * slots + a;
                io.ret[e] = (float)(k == newest ? (double)reward : io.val[e]); io.emit_t[e] = step;
            }
        }
        if (dup0) {                                          // rare: one appended duplicate row
            const int slot = atomicAdd(io.dup_count, 1);
            if (slot < c.dup_capacity) {
                const int t0 = step - newest;
                const float *src = io.x + ((int64_t)(t0 % RL) * slots + a) * D;
                float *dst = io.dup_x + (int64_t)slot * D;
                for (int k0 = 0; k0 < D; k0 += 16) {
                    float v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = (k0 + u < D) ? src[k0 + u] : 0.f;
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (k0 + u < D) dst[k0 + u] = v[u];
                }
                io.dup_r[slot] = (float)r0;
                io.dup_a[slot] = (int32_t)io.act[(int64_t)(t0 % RL) * slots + a];
                io.dup_src[4 * slot + 0] = (int32_t)w;
                io.dup_src[4 * slot + 1] = i;
                io.dup_src[4 * slot + 2] = t0;
                io.dup_src[4 * slot + 3] = step;
            } else {
                atomicAdd(io.dup_count + 1, 1);
            }
        }
        // episode totals: total_reward += score / n_learning ; total_length += len(r_) + 1 per chunk
        // (the leftover chunk adds its own (already zeroed) score and 1 + 1 frames, :199-202,236-237)
        unsafeAtomicAdd(s.ep_reward + w, score / (double)n_learning);
        atomicAdd(s.ep_length + w, n_rows + 1 + (leftover ? 2 : 0));
        score = 0.0;
        // the newest experience stays pending as the seed of the next chunk (:205-208)
        len = 1;
        since = 0;
    }
    if (record) since += 1;

    if (in_range) {
        if (over) {                                            // the episode is over: run_episode starts afresh
            len = 0; since = 0; trained = false; score = 0.0;
        }
        s.len[a] = (uint8_t)len; s.since_flush[a] = (uint8_t)since; s.trained[a] = trained ? 1 : 0; s.score[a] = score;
    }
}

// close a finished episode of world w: episode_log_q.put((now, total_reward, total_length)) (:243)
__device__ __forceinline__ void rollout_close_episode(const RolloutCfg &c, const RolloutState &s, const RolloutIO &io, int64_t w) {
    // (read where the flush atomics accumulated them: agent-scope loads go to the L2, not to a stale L1 line -- the fused
    //  actor kernel closes an episode in the launch, and on the CU, that has just added to these totals.  Issued in front of the
    //  log slot's returning atomic: one trip to the L2 instead of two in a row)
    const double total_reward = __hip_atomic_load(s.ep_reward + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int total_length = __hip_atomic_load(s.ep_length + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int slot = atomicAdd(io.ep_count, 1);
    if (slot < c.ep_capacity) {
        io.ep_out[3 * slot + 0] = (float)w;
        io.ep_out[3 * slot + 1] = (float)total_reward;
        io.ep_out[3 * slot + 2] = (float)total_length;
    } else {
        atomicAdd(io.ep_count + 1, 1);
    }
    s.ep_reward[w] = 0.0;
    s.ep_length[w] = 0;
}

#ifdef CAVOID_ROLLOUT_KERNELS     /* the kernels themselves are compiled by cavoid_rollout_capi.hip only */
__global__ void __launch_bounds__(256) rollout_push_kernel(const RolloutCfg c, const RolloutState s, const RolloutIO io) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool in_range = a < c.num_slots;
    const int N = c.max_agents, RL = c.ring_len;
    const int64_t slots = c.num_slots;
    const int64_t w = in_range ? a / N : 0;
    const int i = in_range ? (int)(a - w * N) : 0;
    const int32_t step = io.step >= 0 ? io.step : *s.step_counter;
    const int blk = step % RL;

    // ---- the step's state rows -> x[blk]: a coalesced sweep of the wavefront's 64 contiguous rows --------
    {
        const int64_t a0 = a - lane;                             // first slot of this wavefront
        int64_t rows = slots - a0;
        rows = rows > 64 ? 64 : (rows < 0 ? 0 : rows);
        rollout_copy_rows(c, io.prev_obs, io.x, a0, (int)rows, blk, lane, 64);
    }

    bool learning = false, done = false, over = false;
    float reward = 0.f, value = 0.f;
    int action = 0;
    int n_learning = 0;
    if (in_range) {
        learning = io.prev_obs[a * c.obs_width] > 0.5f;      // is_learning column (ProcessAgent.py:130)
        done = io.done[a] != 0;
        over = io.game_over[w] != 0;
        reward = io.rewards[a];
        value = io.values[a];
        action = io.actions[a];
        // learning agents of this lane's world: the divisor of the chunk score (:157,195).  A world's N slots
        // are adjacent lanes but may straddle a wavefront edge, so read the is_learning column directly
        for (int k = 0; k < N; ++k) n_learning += io.prev_obs[(w * N + k) * c.obs_width] > 0.5f ? 1 : 0;
    }
    rollout_push_slot(c, s, io, a, w, i, in_range, learning, n_learning, done, over, reward, value, action, step, blk, rollout_slot_load(s, a, in_range));
}

// second, tiny pass (one lane per world, after the push kernel): close finished episodes.
// episode_log_q.put((now, total_reward, total_length)) (:243)
__global__ void __launch_bounds__(256) rollout_episode_kernel(const RolloutCfg c, const RolloutState s, const RolloutIO io) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t W = c.num_slots / c.max_agents;
    if (w == 0 && io.step < 0) *s.step_counter += 1;          // runs after every slot of the push kernel read it
    if (w >= W || io.game_over[w] == 0) return;
    rollout_close_episode(c, s, io, w);
}

// ---- the rows that still need a policy output ------------------------------------------------------------------
// (world, agent) slots whose agent is learning and has not finished: obs column 0 (is_learning) is set, and either the
// world has just restarted (game_over of the step that produced this observation) or the agent was not done in it.
// A finished agent waits for its world's last learning agent (ProcessAgent.py:149-211 keeps stepping the env with
// whatever action; the env ignores it) -- it needs no forward pass.  Order of the list is unspecified.
__global__ void rollout_zero_kernel(int32_t *counter) { *counter = 0; }

__global__ void __launch_bounds__(256) rollout_active_kernel(const RolloutCfg c, const float *obs, const uint8_t *done,
                                                             const uint8_t *game_over, int32_t *row_index, int32_t *row_count) {
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool active = false;
    if (slot < c.num_slots)
        active = obs[slot * c.obs_width] > 0.5f && (game_over[slot / c.max_agents] != 0 || done[slot] == 0);
    const unsigned long long mask = __ballot(active);
    if (mask == 0ull) return;
    const int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(row_count, __popcll(mask));
    base = __shfl(base, leader, 64);
    if (active) row_index[base + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)slot;
}

// ---- hand-over: compact the emitted rows of the step blocks [step_lo, step_hi) into one batch -----------------
struct CompactArgs {
    int32_t step_lo, step_hi;
    int32_t mark_taken;      // 1: stamp the rows emit_t = -2 so that they are not handed out twice (flush_all)
    const float *x;          // [ring_len][slots][D]
    const float *ret;        // [ring_len][slots]
    const uint8_t *act;      // [ring_len][slots]
    int32_t *emit_t;         // [ring_len][slots]
    float *out_x;            // [capacity][D]
    float *out_r;            // [capacity]
    int32_t *out_a;          // [capacity]
    int32_t *out_src;        // [capacity][4] (world, agent, recorded-at step, emitted-at step) or nullptr
    int32_t *out_count;      // [2] rows appended, rows dropped for lack of capacity
    int64_t capacity;
};

// A wavefront takes kCompactSpan groups of 64 consecutive slots of one step block (one lane per slot and group); the workgroup
// appends its emitted rows with ONE atomic (ballots, per-wavefront counts through LDS) -- one returning atomic per wavefront on
// the single counter cost 8192 of them per 16-step hand-over at 4 x 8192, ~90 per microsecond: 110 us for a 109 MB copy -- and
// each wavefront then copies its rows together: the destination is one contiguous run of rows, the sources are (mostly adjacent)
// rows of the wavefront's slots, so both sides of the copy are coalesced.  Row order inside the batch is unspecified (the A3C
// loss is a sum over rows).
constexpr int kCompactSpan = 4;
__global__ void __launch_bounds__(256) rollout_compact_kernel(const RolloutCfg c, const CompactArgs a) {
    __shared__ int slot_of_rank[4][64 * kCompactSpan];
    __shared__ int wave_count[4];
    __shared__ int block_base;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, D = c.obs_width - 1;
    const int step = a.step_lo + (int)blockIdx.y;
    const int64_t span0 = ((int64_t)blockIdx.x * 4 + wave) * (64 * kCompactSpan);     // first slot of this wavefront's span
    const int64_t block_row0 = (int64_t)(step % c.ring_len) * c.num_slots;
    int32_t emitted[kCompactSpan];
    int rank[kCompactSpan];
    int count = 0;
#pragma unroll
    for (int g = 0; g < kCompactSpan; ++g) {
        const int64_t slot = span0 + 64 * g + lane;
        emitted[g] = slot < c.num_slots ? a.emit_t[block_row0 + slot] : -1;
    }
#pragma unroll
    for (int g = 0; g < kCompactSpan; ++g) {
        const unsigned long long mask = __ballot(emitted[g] >= 0);
        rank[g] = count + __popcll(mask & ((1ull << lane) - 1ull));
        count += __popcll(mask);
    }
    if (lane == 0) wave_count[wave] = count;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        block_base = total ? atomicAdd(a.out_count, total) : 0;
    }
    __syncthreads();
    if (count == 0) return;                                // (wave-uniform)
    int64_t base = block_base;
    for (int w = 0; w < wave; ++w) base += wave_count[w];
    const int fit = base + count <= a.capacity ? count : (int)(a.capacity > base ? a.capacity - base : 0);
    if (lane == 0 && fit < count) atomicAdd(a.out_count + 1, count - fit);
#pragma unroll
    for (int g = 0; g < kCompactSpan; ++g)
        if (emitted[g] >= 0) slot_of_rank[wave][rank[g]] = 64 * g + lane;
    __builtin_amdgcn_wave_barrier();                       // wave-private table: LDS operations of a wavefront are in order
    const uint32_t inv_d = (uint32_t)((1ull << 32) / (uint32_t)D) + 1u;      // e / D by multiply-shift (e < 2^16)
    const float *__restrict__ src = a.x + (block_row0 + span0) * D;
    float *__restrict__ dstx = a.out_x + base * D;
    const int total = fit * D;
    for (int e0 = lane; e0 < total; e0 += 64 * 8) {        // 8 loads in flight per lane, then 8 stores
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 64 * u;
            const int r = (int)(((uint64_t)(uint32_t)e * inv_d) >> 32), k = e - r * D;
            v[u] = e < total ? src[slot_of_rank[wave][r] * D + k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 64 * u;
            if (e < total) dstx[e] = v[u];
        }
    }
#pragma unroll
    for (int g = 0; g < kCompactSpan; ++g) {
        if (emitted[g] < 0 || rank[g] >= fit) continue;
        const int64_t slot = span0 + 64 * g + lane, row = block_row0 + slot, dst = base + rank[g];
        a.out_r[dst] = a.ret[row];
        a.out_a[dst] = (int32_t)a.act[row];
        if (a.out_src) {
            a.out_src[4 * dst + 0] = (int32_t)(slot / c.max_agents);
            a.out_src[4 * dst + 1] = (int32_t)(slot % c.max_agents);
            a.out_src[4 * dst + 2] = step;
            a.out_src[4 * dst + 3] = emitted[g];
        }
        if (a.mark_taken) a.emit_t[row] = -2;
    }
}

#endif

}  // namespace cavoid
