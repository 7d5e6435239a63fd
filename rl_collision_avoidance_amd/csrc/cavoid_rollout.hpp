// cavoid_rollout.hpp -- gfx950 kernel of the batched GA3C actor bookkeeping: what
// ProcessAgent.run_episode / _accumulate_rewards / convert_to_nparray do per actor process
// (/root/reference/ga3c/GA3C/ProcessAgent.py:54-87,105-211; rows R3-R5 of SURVEY.md section 8a),
// for every (world, agent) slot at once, on the device, with no queue hop.
//
// One lane per (world, agent) slot (flat index a = w*N + i, so per-slot arrays are read and
// written coalesced).  Each slot owns a ring of T_max+1 experiences in HBM, laid out
// [entry][slot] so that all lanes touch the same entry row together.  A step appends one
// experience; a flush runs the backward n-step return over <= T_max+1 entries and appends the
// resulting training rows to a device-side batch through one atomic reservation per wavefront.
// HBM-bound integer/float bookkeeping: no MFMA, no LDS (nothing is shared between slots except
// the world-level episode counters, reduced with wave ballots / DPP-free shuffles).
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
    double discount;         // Config.DISCOUNT (Config.py:103)
    int64_t capacity;        // rows the output batch can hold
    int64_t ep_capacity;     // episode-log records
};

struct RolloutState {
    float *ring_x;           // [T_max+1][slots][D]
    double *ring_r;          // [T_max+1][slots]   single-step reward, overwritten by the n-step return
    int32_t *ring_t;         // [T_max+1][slots]   provenance: global step at which it was recorded
    uint8_t *ring_a;         // [T_max+1][slots]
    uint8_t *len;            // [slots] experiences held
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
    int32_t step;            // global step index (provenance); < 0: use the device-side counter (graph replays)
    float *out_x;            // [capacity][D]
    float *out_r;            // [capacity]
    int32_t *out_a;          // [capacity]
    int32_t *out_src;        // [capacity][4]  world, agent, recorded-at step, emitted-at step
    int32_t *out_count;      // [4]  rows reserved, rows dropped for lack of capacity, first dropped row (INT_MAX: none), -
    float *ep_out;           // [ep_capacity][3]  world, total_reward, total_length
    int32_t *ep_count;       // [2]  records appended, dropped
};

__global__ void __launch_bounds__(256) rollout_push_kernel(const RolloutCfg c, const RolloutState s, const RolloutIO io) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool in_range = a < c.num_slots;
    const int N = c.max_agents, D = c.obs_width - 1, L = c.time_max + 1;
    const int64_t slots = c.num_slots;
    const int64_t w = in_range ? a / N : 0;
    const int i = in_range ? (int)(a - w * N) : 0;
    const int32_t step = io.step >= 0 ? io.step : *s.step_counter;

    bool learning = false, done = false, over = false;
    float reward = 0.f, value = 0.f;
    int action = 0;
    if (in_range) {
        learning = io.prev_obs[a * c.obs_width] > 0.5f;      // is_learning column (ProcessAgent.py:130)
        done = io.done[a] != 0;
        over = io.game_over[w] != 0;
        reward = io.rewards[a];
        value = io.values[a];
        action = io.actions[a];
    }
    int len = 0, since = 0;
    bool trained = false;
    double score = 0.0;
    if (in_range) { len = s.len[a]; since = s.since_flush[a]; trained = s.trained[a] != 0; score = s.score[a]; }

    // number of learning agents of this lane's world (the divisor of the chunk score, :157,195);
    // a world's N slots are adjacent lanes but may straddle a wavefront edge, so count via memory-free
    // neighbour reads of the is_learning column instead of a ballot
    int n_learning = 0;
    if (in_range)
        for (int k = 0; k < N; ++k) n_learning += io.prev_obs[(w * N + k) * c.obs_width] > 0.5f ? 1 : 0;

    int n_rows = 0;            // rows of the main chunk
    bool leftover = false;     // + one separate 1-row chunk
    int count = 0;             // entries the backward pass covers
    bool flush = false;
    const bool frozen = !c.reflush_done && trained;          // cleaned mode: a trained agent records nothing more
    if (learning && !frozen) {
        score += (double)reward;
        // ---- append (Experience(previous_state[0,i,:], action, prediction, reward, done), :172-177) ----
        const int pos = len;
        const float *src = io.prev_obs + a * c.obs_width + 1;
        float *dst = s.ring_x + ((int64_t)pos * slots + a) * D;
        for (int k = 0; k < D; ++k) dst[k] = src[k];
        s.ring_r[(int64_t)pos * slots + a] = (double)reward;
        s.ring_a[(int64_t)pos * slots + a] = (uint8_t)action;
        s.ring_t[(int64_t)pos * slots + a] = step;
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

    // ---- reserve output rows: wave-level exclusive scan + one atomic per wavefront ----------------------
    const int mine = n_rows + (leftover ? 1 : 0);
    int prefix = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(prefix, off);
        if (lane >= off) prefix += v;
    }
    const int wave_total = __shfl(prefix, 63);
    int64_t base = 0;
    if (wave_total > 0) {
        int wave_base = 0;
        if (lane == 63) wave_base = atomicAdd(io.out_count, wave_total);
        wave_base = __shfl(wave_base, 63);
        base = (int64_t)wave_base + (prefix - mine);
    }

    if (flush) {
        // ---- n-step return, newest to oldest, overwriting the stored rewards (:54-79) ------------------
        double R = done ? 0.0 : (double)value;
        if (done) trained = true;
        for (int k = count - 1; k >= 0; --k) {
            const int64_t e = (int64_t)k * slots + a;
            R = c.discount * R + s.ring_r[e];
            s.ring_r[e] = R;
        }
        // ---- emit rows (convert_to_nparray, :82-87) -----------------------------------------------------
        const bool fits = base + mine <= c.capacity;
        if (fits) {
            for (int k = 0; k < mine; ++k) {                     // the leftover row is entry len-1
                const int64_t e = (int64_t)k * slots + a;
                const int64_t o = base + k;
                const float *src = s.ring_x + e * D;
                float *dst = io.out_x + o * D;
                for (int q = 0; q < D; ++q) dst[q] = src[q];
                io.out_r[o] = (float)s.ring_r[e];
                io.out_a[o] = (int32_t)s.ring_a[e];
                io.out_src[4 * o + 0] = (int32_t)w;
                io.out_src[4 * o + 1] = i;
                io.out_src[4 * o + 2] = s.ring_t[e];
                io.out_src[4 * o + 3] = step;
            }
        } else {                                                 // batch full: rows below the first failure stay valid
            atomicAdd(io.out_count + 1, mine);
            atomicMin(io.out_count + 2, (int32_t)(base < 0x7fffffff ? base : 0x7fffffff));
        }
        // episode totals: total_reward += score / n_learning ; total_length += len(r_) + 1 per chunk
        // (the leftover chunk adds its own (already zeroed) score and 1 + 1 frames, :199-202,236-237)
        atomicAdd(s.ep_reward + w, score / (double)n_learning);
        atomicAdd(s.ep_length + w, n_rows + 1 + (leftover ? 2 : 0));
        score = 0.0;
        // ---- keep the newest experience as the seed of the next chunk (:205-208) ------------------------
        if (len > 1) {
            const int64_t last = (int64_t)(len - 1) * slots + a, first = a;
            const float *src = s.ring_x + last * D;
            float *dst = s.ring_x + first * D;
            for (int q = 0; q < D; ++q) dst[q] = src[q];
            s.ring_r[first] = s.ring_r[last];
            s.ring_a[first] = s.ring_a[last];
            s.ring_t[first] = s.ring_t[last];
        }
        len = 1;
        since = 0;
    }
    if (learning && !frozen) since += 1;

    if (in_range) {
        if (over) {                                            // the episode is over: run_episode starts afresh
            len = 0; since = 0; trained = false; score = 0.0;
        }
        s.len[a] = (uint8_t)len; s.since_flush[a] = (uint8_t)since; s.trained[a] = trained ? 1 : 0; s.score[a] = score;
    }
}

// second, tiny pass (one lane per world, after the push kernel): close finished episodes.
// episode_log_q.put((now, total_reward, total_length)) (:243)
__global__ void __launch_bounds__(256) rollout_episode_kernel(const RolloutCfg c, const RolloutState s, const RolloutIO io) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t W = c.num_slots / c.max_agents;
    if (w == 0 && io.step < 0) *s.step_counter += 1;          // runs after every slot of the push kernel read it
    if (w >= W || io.game_over[w] == 0) return;
    const int slot = atomicAdd(io.ep_count, 1);
    if (slot < c.ep_capacity) {
        io.ep_out[3 * slot + 0] = (float)w;
        io.ep_out[3 * slot + 1] = (float)s.ep_reward[w];
        io.ep_out[3 * slot + 2] = (float)s.ep_length[w];
    } else {
        atomicAdd(io.ep_count + 1, 1);
    }
    s.ep_reward[w] = 0.0;
    s.ep_length[w] = 0;
}

}  // namespace cavoid
