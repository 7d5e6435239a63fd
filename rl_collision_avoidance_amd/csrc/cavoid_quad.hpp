// cavoid_quad.hpp -- quad_env_tile<N>: ONE auto-reset step of a tile (the closed-loop form: a policy in the loop, one `env.step` per policy
// step -- ga3c/GA3C/Environment.py:110-116) by FOUR cooperating wavefronts of a workgroup, and env_quad_kernel<N>, the launch of it.
//
// env_tile<N, MODE_STEP_AUTORESET> steps a tile on one wavefront: the step is that wavefront's dependent chain -- dynamics (2.0 k clocks at
// N = 4), then the N-1 neighbour chains of the pair pass interleaved on one issue port (2.6 k), rewards / restart (1.3 k), ranking + rows
// (2.5 k), tile flush (1.4 k) -- with three SIMDs of the CU idle (one-step launches: 512 tiles on 1024 SIMDs at 4 x 8192) or, in the fused
// actor kernel, with the tile's other three wavefronts waiting at the barrier.  Here the step's wide parts are spread over lanes of FOUR
// wavefronts ("ordered pairs (i, j) mapped to lanes"; the round-5 verdict's items 1 (b) and 4):
//
//   wavefront 0 (host)      loads, E4 decode, E5 dynamics, staging | own ego frame | E7 rewards, E8 done / game_over, restart, state
//                           write-back, the HEAD of its rows (6 values + the empty slots' zeros)
//   wavefront 1 + p, p<3    lane L = host lane L's neighbours o = p, p + 3, ..: the square-root chain, gap, collision test, sort key and the
//                           five observation features of each; then -- beside the host's reward phase -- the ranking of the lane's
//                           neighbours (every pair wavefront runs it, from the same keys) and ITS neighbours' seven values into the row
//   all four                the tile flush, 16 bytes per lane and round
//
// Hand-over through LDS, three workgroup barriers per step (staged state | pair results | rows); two more when a world of the tile
// restarts (the new episode's first observation needs the new agents' keys: the pair wavefronts redo their part on the restaged state --
// the rows they wrote before the restart was known are simply written again).  Every value is computed by the statements of env_tile /
// pair_pass_impl / assemble_obs on the same inputs -- value for value, like the pipeline and relay forms -- so the outputs are
// BIT-identical to env_kernel's (tests/test_gpu_quad.py); what changes is which lane computes them.
//
// Carried: table actions, unicycle dynamics (+ max turn rate), scripted static / non-cooperative agents (a frozen-network agent's
// index comes in like a learner's), restarts from the pool / the look-ahead rings / the per-agent generator, every sort order and
// U-switch, the packed record.  Not carried (the launchers say CAVOID_EUNSUPPORTED and the caller takes env_kernel / the single-wavefront
// env step): ORCA agents, box scenarios generated inside the step, continuous actions, holonomic dynamics, a null obs, tiles that go
// out in several passes (tile_rows < rows of the tile).
#pragma once
#include "cavoid_kernels.hpp"

namespace cavoid {

constexpr int kQuadPairWaves = 3;

template <int N>
struct QuadShared {
    static constexpr int K = Others<N>::K;
    double px[64], py[64], vx[64], vy[64], heading[64];  // the staged post-move state (ArrayStage's arrays + the heading)
    float r[64], rad[64], gx[64], gy[64];                 // r: radius, < 0 = absent row (the staging convention); rad: the radius itself
    uint64_t key[K][64];                                  // per neighbour: the sort key,
    double gap_c[K][64];                                  // the unordered-pair gap of the collision test,
    uint32_t bits[K][64];                                 // 1 = takes part in the collision test, 2 = seen (inside the sensing horizon)
    uint32_t frozen_w[64];                                // (U4 flipped) the lane's world: agents done before the step
    int again;                                            // a world of the tile restarted: one more pair round on the new state
};

template <int N>
__host__ __device__ constexpr size_t quad_lds_fixed_bytes() { return (size_t)lds_floats_block() * sizeof(float) + ((sizeof(QuadShared<N>) + 15) & ~(size_t)15); }
template <int N>
__host__ __device__ inline size_t quad_lds_bytes(int tile_floats) { return quad_lds_fixed_bytes<N>() + (size_t)tile_floats * sizeof(float); }

// pair_pass_impl's statements (FEAT form) for the neighbours o = pw, pw + 3, .. of the host agent in lane `lane`; the host's post-move state
// comes from the staging arrays like the neighbours', its ego axes are ego_from's values of the same inputs as the host wavefront's own
template <int N, bool SW>
__device__ __forceinline__ void quad_pair_round(const KCfg &c, QuadShared<N> &sh, const int lane, const int i, const int base, const bool active,
                                                const int pw, Agent &a, Ego &e, float (&gapf)[Others<N>::K], float (&feat)[Others<N>::K][kFeat]) {
    const ArrayStage<N> st{sh.px, sh.py, sh.vx, sh.vy, sh.r, i, base};
    a.px = sh.px[lane]; a.py = sh.py[lane]; a.vx = sh.vx[lane]; a.vy = sh.vy[lane]; a.heading = sh.heading[lane];
    a.radius = sh.rad[lane]; a.gx = sh.gx[lane]; a.gy = sh.gy[lane];
    const bool present = active && sh.r[lane] >= 0.0f;    // (staged r < 0 <=> the row is absent)
    a.flags = present ? (uint32_t)CAVOID_F_PRESENT : 0u;
    e = ego_from(c, (double)a.gx - a.px, (double)a.gy - a.py, a.heading);
    const double ri = (double)a.radius;
    const uint32_t frozen_w = SW ? sh.frozen_w[lane] : 0u;
#pragma unroll
    for (int o = 0; o < N - 1; ++o) {
        if (o % kQuadPairWaves != pw) continue;            // (wave-uniform)
        const OtherState q = st.other(o);
        const float rjf = q.r;
        const double rx = q.px - a.px, ry = q.py - a.py;
        const double d = sqrt_dist2(rx * rx + ry * ry);
        const bool other = present && (rjf >= 0.0f);
        bool collides = other;
        if (SW) {
            const int jj = other_index(i, o, N);
            collides = other && ((frozen_w >> jj) & 1u) == 0u && ((frozen_w >> i) & 1u) == 0u;
        }
        const double gap_c = d - (ri + (double)rjf);
        const bool seen = other && !(d > c.horizon);
        const double gap_o = d - ri - (double)rjf;
        uint32_t hi = kKeyBias - (uint32_t)(int)rint(gap_o * 100.0);
        uint32_t lo = orderable((float)(ry * e.tx - rx * e.ty));
        if (SW) {
            if (c.switches & kSwIndexTie) lo = (uint32_t)other_index(i, o, N);
            if (c.switches & kSwExactGap) { lo = 0u; hi = 0x7FFFFFFEu - (orderable((float)gap_o) >> 1); }
        }
        hi = seen ? hi : kKeySentinel + (uint32_t)o;
        sh.key[o][lane] = ((uint64_t)hi << 32) | lo;
        sh.gap_c[o][lane] = gap_c;
        sh.bits[o][lane] = (collides ? 1u : 0u) | (seen ? 2u : 0u);
        gapf[o] = (float)gap_o;
        neighbour_features(e, (float)e.prll_x, (float)e.prll_y, rx, ry, q, feat[o]);
    }
}

// the pair wavefront's part of E9: the ranking of the lane's neighbours (assemble_obs's, from the keys every pair wavefront left in LDS) and
// the seven values of ITS neighbours into the row
template <int N>
__device__ __forceinline__ void quad_rows(const KCfg &c, QuadShared<N> &sh, const int lane, const int i, const int base, const bool active, const int pw,
                                          const Agent &a, const Ego &e, const float (&gapf)[Others<N>::K], const float (&feat)[Others<N>::K][kFeat],
                                          float *tile, int rows_active, int ostride, int64_t wave) {
    constexpr int K = Others<N>::K;
    const ArrayStage<N> st{sh.px, sh.py, sh.vx, sh.vy, sh.r, i, base};
    Key key[K];
    uint32_t valid = 0u;
    if (N == 1) key[0].set(kKeySentinel, 0u);
#pragma unroll
    for (int o = 0; o < N - 1; ++o) {
        key[o].v = sh.key[o][lane];
        valid |= (sh.bits[o][lane] & 2u) ? (1u << o) : 0u;
    }
    assemble_obs<N, false, true, ArrayStage<N>, NoHook, false, PartOthers>(c, a, e, active, lane, st, key, gapf, feat, valid, tile, nullptr, rows_active,
                                                                           ostride, false, 0.0f, 0.0f, wave, NoHook(), false, nullptr,
                                                                           PartOthers{pw, kQuadPairWaves});
}

// the tile's rows out of LDS by every wavefront of the workgroup: flush_tile with tid / nthreads for lane / 64
__device__ __forceinline__ void quad_flush(const float *tile, float *dst, int n_floats, int tid, int nthreads) {
    if ((n_floats & 3) == 0 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
        const float4 *src4 = reinterpret_cast<const float4 *>(tile);
        float4 *dst4 = reinterpret_cast<float4 *>(dst);
        const int n4 = n_floats >> 2;
        for (int k0 = tid; k0 < n4; k0 += nthreads * 4) {       // up to 4 LDS reads in flight per lane, then the stores
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + nthreads * u;
                v[u] = k < n4 ? src4[k] : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + nthreads * u;
                if (k < n4) dst4[k] = v[u];
            }
        }
    } else {
        for (int k = tid; k < n_floats; k += nthreads) dst[k] = tile[k];
    }
}

// One auto-reset step of tile `wave` by the four wavefronts (role = 0..3, wave-uniform) of a workgroup.  smem: quad_lds_bytes<N>(tile
// floats) of the workgroup's LDS.  Contains workgroup barriers: every thread of the four wavefronts calls it.  out (role 0 only): what
// env_tile hands a caller that goes on in the same kernel.
template <int N>
__device__ __forceinline__ void quad_env_tile(const KCfg &c, const KState &s, const PoolRec *pool, const KIO &io, unsigned char *smem, const int role,
                                              const int lane, const int64_t wave, StepOut *out = nullptr) {
    constexpr int K = Others<N>::K;
    double *lds_tab = reinterpret_cast<double *>(smem);
    QuadShared<N> &sh = *reinterpret_cast<QuadShared<N> *>(smem + lds_floats_block() * sizeof(float));
    float *tile = reinterpret_cast<float *>(smem + quad_lds_fixed_bytes<N>());
    const int width = c.width, ostride = io.obs_stride;
    const int wpw = c.wpw, lanes_used = wpw * N;
    const int64_t w0 = wave * wpw;
    const int lw = lane / N, i = lane - lw * N;
    const int64_t w = w0 + lw;
    const bool active = lane < lanes_used && w < c.num_worlds;
    const int base = lane < lanes_used ? lw * N : 0;
    const int64_t a_idx = w * N + i;
    const bool packed = io.packed != 0;
    int64_t worlds_here = c.num_worlds - w0;
    if (worlds_here > wpw) worlds_here = wpw;
    if (worlds_here < 0) worlds_here = 0;
    const int rows_active = (int)worlds_here * N;
    const bool sw = c.switches != 0u;                      // (uniform)

    if (role == 0) {
        // ================================================ the host wavefront ===================================================
        CAVOID_STAMP(0);
        double tab_v = 0.0;
        if (lane < 2 * c.num_actions) tab_v = c.action_table[lane];
        Agent a;
        a.px = a.py = a.heading = a.t_rem = a.vx = a.vy = 0.0;
        a.gx = a.gy = a.radius = a.pref = a.speed = 0.0f;
        a.flags = 0u;
        uint32_t episode = 0u;
        int act = 0;
        if (active) {
            episode = s.episode[w];
            load_agent(s, a_idx, a);
            act = io.actions[a_idx];
        }
        lds_tab[lane] = tab_v;
        CAVOID_STAMP(1);
        const bool present_first = active && (a.flags & CAVOID_F_PRESENT);
        bool restarted_any = false, moved_any = false;
        CAVOID_STAMP(2);
        const uint32_t flags_in = a.flags;
        const bool present_in = active && (flags_in & CAVOID_F_PRESENT);
        const bool done_in = (flags_in & CAVOID_F_DONE_MASK) != 0u;
        // ---- E4 decode (env_tile's statements) -------------------------------------------------------------------------------
        wave_lds_sync();
        const uint32_t pol = (flags_in >> CAVOID_F_POLICY_SHIFT) & CAVOID_F_POLICY_MASK;
        double a0 = 0.0, a1 = 0.0;
        act = act < 0 ? 0 : (act >= c.num_actions ? c.num_actions - 1 : act);
        a0 = (double)a.pref * lds_tab[2 * act];
        a1 = lds_tab[2 * act + 1];
        if (CAVOID_RARE(__ballot(present_in && !done_in && pol != 0u) != 0ull)) {   // scripted agents in this tile
            if (pol == 1u) { a0 = 0.0; a1 = 0.0; }
            if (pol == 2u) {                                            // straight at the goal, full speed
                const Ego e0 = ego_frame_exact(c, a);
                a0 = (double)a.pref;
                a1 = -e0.heading_ego;
            }
        }
        if (c.actions_fp32) { a0 = (double)(float)a0; a1 = (double)(float)a1; }
        // ---- E5 dynamics -----------------------------------------------------------------------------------------------------
        const bool moving = present_in && !done_in;
        moved_any = moved_any || moving;
        {
            double dh = a1;
            if (CAVOID_RARE(c.dynamics == CAVOID_DYN_UNICYCLE_MAX_TURN)) {
                const double rate = fmin(fmax(dh / c.dt, -c.cold->max_turn_rate), c.cold->max_turn_rate);
                dh = rate * c.dt;
            }
            const double nh = wrap_angle(dh + a.heading, c.switches);
            double sn = 0.0, cs = 1.0;
            sincos_bounded(nh, &sn, &cs);
            const double npx = a.px + a0 * cs * c.dt, npy = a.py + a0 * sn * c.dt;
            const double nvx = a0 * cs, nvy = a0 * sn, nsp = a0;
            a.px = moving ? npx : a.px; a.py = moving ? npy : a.py; a.heading = moving ? nh : a.heading;
            a.vx = moving ? nvx : 0.0; a.vy = moving ? nvy : 0.0; a.speed = moving ? (float)nsp : 0.0f;
        }
        if (present_in && done_in) {                                    // frozen: latch the 'already' flags
            if (flags_in & CAVOID_F_AT_GOAL) a.flags |= CAVOID_F_WAS_AT_GOAL;
            if (flags_in & CAVOID_F_IN_COLL) a.flags |= CAVOID_F_WAS_IN_COLL;
        }
        if (moving) {
            const double dx = a.px - (double)a.gx, dy = a.py - (double)a.gy;
            if (dx * dx + dy * dy <= c.near_goal_sq) a.flags |= CAVOID_F_AT_GOAL;
            a.t_rem -= c.dt;
            if (c.timeout_enabled && a.t_rem <= 0.0) a.flags |= CAVOID_F_RAN_OUT;
        }
        CAVOID_STAMP(3);                                        // dynamics done
        bool present = active && (a.flags & CAVOID_F_PRESENT);
        auto stage_self = [&](bool is_present) {
            sh.px[lane] = a.px; sh.py[lane] = a.py; sh.vx[lane] = a.vx; sh.vy[lane] = a.vy; sh.heading[lane] = a.heading;
            sh.r[lane] = is_present ? a.radius : -1.0f;
            sh.rad[lane] = a.radius;
            sh.gx[lane] = a.gx; sh.gy[lane] = a.gy;
        };
        stage_self(present);
        if (CAVOID_RARE(c.switches & kSwSkipDonePairs))
            sh.frozen_w[lane] = (uint32_t)(__ballot(present_in && done_in) >> base) & ((1u << N) - 1u);
        else if (sw) sh.frozen_w[lane] = 0u;
        __syncthreads();                                        // S1: the staged state is in LDS
        Ego e = ego_frame_obs(c, a);                            // (beside the pair wavefronts' chains)
        uint32_t valid = 0u;
        __syncthreads();                                        // S2: the pair results are in LDS
        bool hit = false;
        double min_gap = INFINITY;
#pragma unroll
        for (int o = 0; o < N - 1; ++o) {                       // pair_pass_impl's accumulation, in neighbour order
            const uint32_t b = sh.bits[o][lane];
            const double gap_c = sh.gap_c[o][lane];
            const bool collides = (b & 1u) != 0u;
            min_gap = collides ? fmin(min_gap, gap_c) : min_gap;
            hit = hit || (collides && gap_c <= c.collision_dist);
            valid |= (b & 2u) ? (1u << o) : 0u;
        }
        CAVOID_STAMP(4);                                        // ego frame + pair results in
        // ---- E7 rewards, E8 done (env_tile's statements) -----------------------------------------------------------------------
        double r = 0.0;
        bool done = true;
        if (present) {
            r = c.r_step;
            if (a.flags & CAVOID_F_AT_GOAL) { if (!(a.flags & CAVOID_F_WAS_AT_GOAL)) r = c.r_goal; }
            else if (!(a.flags & CAVOID_F_WAS_IN_COLL)) {
                if (hit) { r = c.r_coll; a.flags |= CAVOID_F_IN_COLL; }
                else if (min_gap <= c.close_range) r = c.r_close + c.close_slope * min_gap;
            }
            r = fmin(fmax(r, c.clip_lo), c.clip_hi);
            done = (a.flags & CAVOID_F_DONE_MASK) != 0u;
        }
        const unsigned long long running = __ballot(present && ((a.flags & CAVOID_F_LEARNING) || c.evaluate_mode) && !done);
        const unsigned long long wmask = ((1ull << N) - 1ull) << base;
        const bool game_over = (running & wmask) == 0ull;
        const float rew_f = (float)r, done_f = done ? 1.0f : 0.0f;
        if (out) { out->reward = rew_f; out->done = done; out->game_over = game_over; }
        if (active) {
            if (!packed) {
                io.rew[a_idx] = rew_f;
                io.done[a_idx] = done ? 1 : 0;
            }
            if (i == 0) io.game_over[w] = game_over ? 1 : 0;
        }
        const bool restart = active && game_over;
        const bool any_restart = __ballot(restart) != 0ull;     // (wave-uniform)
        if (CAVOID_RARE(any_restart)) {                         // some world of this tile restarts: restage, one more pair round
            if (restart) {
                episode += 1u;
                restarted_any = true;
                new_episode<N>(c, pool, (uint32_t)(c.world_offset + w), episode, i, a);
                present = (a.flags & CAVOID_F_PRESENT) != 0u;
                stage_self(present);                            // (a fresh agent stands: a.vx = a.vy = 0)
                e = ego_frame_obs(c, a);
            }
            if (lane == 0) sh.again = 1;
            __syncthreads();                                    // S3: "again" + the restaged state
            __syncthreads();                                    // S4: the pair results of the new state
            valid = 0u;
#pragma unroll
            for (int o = 0; o < N - 1; ++o) valid |= (sh.bits[o][lane] & 2u) ? (1u << o) : 0u;
        } else if (lane == 0) {
            sh.again = 0;
        }
        CAVOID_STAMP(5);                                        // rewards / restart done
        if (out) out->learning_next = active && (a.flags & CAVOID_F_PRESENT) != 0u && (a.flags & CAVOID_F_LEARNING) != 0u;
        // the state is final: its stores complete under the rows (env_tile's one-step order)
        if (restarted_any) {
            store_agent(s, a_idx, a);
            if (i == 0) s.episode[w] = episode;
        } else if (present_first) {
            if (moved_any) { s.px[a_idx] = a.px; s.py[a_idx] = a.py; s.heading[a_idx] = a.heading; s.t_rem[a_idx] = a.t_rem; }
            s.speed[a_idx] = a.speed;
            s.flags[a_idx] = a.flags;
        }
        CAVOID_STAMP(6);
        // ---- E9, the host's part: the head of the row and the empty slots (assemble_obs's statements) ------------------------------
        if (active && lane < rows_active) {
            const int M = c.max_other;
            const int m = __popc(valid);
            const int first = m > M ? m - M : 0;
            const int kept = m - first;
            float *row = tile + lane * ostride;
            row[0] = (present && (a.flags & CAVOID_F_LEARNING)) ? 1.0f : 0.0f;
            row[1] = (float)kept;
            row[2] = present ? (float)e.dist : 0.0f;
            row[3] = present ? (float)e.heading_ego : 0.0f;
            row[4] = present ? a.pref : 0.0f;
            row[5] = present ? a.radius : 0.0f;
            for (int sl = kept; sl < M; ++sl) {
                float *z = row + 6 + 7 * sl;
#pragma unroll
                for (int q = 0; q < 7; ++q) z[q] = 0.0f;
            }
            if (packed) { row[width] = rew_f; row[width + 1] = done_f; }
        }
        __syncthreads();                                        // S3 (S5 after a restart): the rows are in the tile
    } else {
        // ================================================ a pair wavefront ============================================================
        const int pw = role - 1;
        Agent a;
        a.px = a.py = a.heading = a.t_rem = a.vx = a.vy = 0.0;
        a.gx = a.gy = a.radius = a.pref = a.speed = 0.0f;
        a.flags = 0u;
        Ego e;
        float gapf[K], feat[K][kFeat];
#pragma unroll
        for (int o = 0; o < K; ++o) {
            gapf[o] = 0.0f;
#pragma unroll
            for (int q = 0; q < kFeat; ++q) feat[o][q] = 0.0f;
        }
        __syncthreads();                                        // S1
        if (sw) quad_pair_round<N, true>(c, sh, lane, i, base, active, pw, a, e, gapf, feat);
        else quad_pair_round<N, false>(c, sh, lane, i, base, active, pw, a, e, gapf, feat);
        __syncthreads();                                        // S2
        quad_rows<N>(c, sh, lane, i, base, active, pw, a, e, gapf, feat, tile, rows_active, ostride, wave);   // (beside the host's reward phase)
        __syncthreads();                                        // S3
        if (CAVOID_RARE(sh.again != 0)) {                       // (uniform) a world restarted: the new agents' keys, ranks and rows
            if (sw) quad_pair_round<N, true>(c, sh, lane, i, base, active, pw, a, e, gapf, feat);
            else quad_pair_round<N, false>(c, sh, lane, i, base, active, pw, a, e, gapf, feat);
            __syncthreads();                                    // S4
            quad_rows<N>(c, sh, lane, i, base, active, pw, a, e, gapf, feat, tile, rows_active, ostride, wave);
            __syncthreads();                                    // S5
        }
    }
    // ---- every wavefront: its share of the tile flush --------------------------------------------------------------------------------
    if (worlds_here > 0) quad_flush(tile, io.obs + w0 * N * ostride, rows_active * ostride, role * 64 + lane, 256);
    if (role == 0) { CAVOID_STAMP(7); CAVOID_STAMP(8); }
}

// cavoid_step_autoreset's one-step launch with four wavefronts per tile
template <int N>
__global__ void __launch_bounds__(256) env_quad_kernel(const KCfg c, const KState s, const PoolRec *pool, const KIO io) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    quad_env_tile<N>(c, s, pool, io, smem, role, lane, (int64_t)blockIdx.x);
}

}  // namespace cavoid
