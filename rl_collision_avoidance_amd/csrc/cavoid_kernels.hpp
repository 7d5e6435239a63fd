// cavoid_kernels.hpp -- hand-written gfx950 (CDNA4) kernels of the batched env.step hot path.
//
// Replaces, for W worlds at once, what one reference `self.game.step(action)` call does for one
// world (ga3c/GA3C/Environment.py:112): E4 action decode, E5 dynamics, E6 pairwise gaps /
// collisions, E7 rewards, E8 done flags, E9 ego-frame neighbour-sorted observation
// (obs layout ga3c/GA3C/Config.py:40,72-76).
//
// Mapping (DESIGN.md "Kernels"):
//   * one lane per agent ("host"), one 64-lane wavefront per tile of floor(64/N) whole worlds;
//     flat agent index a = w*N + i, so a wavefront's agents are CONTIGUOUS in every SoA field
//     and each field is one coalesced global_load per wavefront;
//   * post-move agent state (pos, vel, radius) is staged in LDS, wave-private, and the O(N^2)
//     neighbour pass reads the other agents of the lane's world from there (same-world lanes
//     read the same address -> LDS broadcast);
//   * neighbour ordering by counting ranks (O(N^2) compares, no data-dependent control flow);
//   * the [agents, 1+D] observation tile is assembled in LDS and leaves as 16-byte coalesced
//     stores (a lane-owns-a-row store would touch 64 cache lines per instruction);
//   * float64 arithmetic throughout (the reference env is NumPy float64; flags are threshold
//     tests that flip on fp32 rounding), compiled with -ffp-contract=off so that the only
//     numerical difference from the float64 CPU oracle is sin/cos/atan2 (ocml vs libm, <=1 ulp).
//   No MFMA: there is no dense contraction anywhere on this path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cavoid.h"

namespace cavoid {

constexpr double kPi = 3.14159265358979323846;

// kernel-argument POD (by value).  The action table lives in device memory (per-lane index).
struct KCfg {
    double dt, near_goal_sq, near_goal, max_time_ratio, collision_dist, close_range;
    double r_goal, r_coll, r_close, r_step, close_slope, clip_lo, clip_hi, horizon, max_turn_rate;
    double gen_nonlearning, gen_static, gen_goal_jitter, gen_angle_jitter;
    int32_t max_other, width, sort_method, dynamics, actions_fp32, timeout_enabled, num_actions;
    int32_t gen_min_agents, gen_max_agents;
    uint32_t seed_lo, seed_hi;
    int64_t num_worlds, world_offset;
    const double *action_table;  // [num_actions][2]
};

struct KState {
    double *px, *py, *heading, *t_rem;
    float *gx, *gy, *radius, *pref, *speed;
    uint32_t *flags;
    uint32_t *episode;  // [W]
};

struct KIO {
    const int32_t *actions;  // [W,N] or null
    const float *cont;       // [W,N,2] or null
    const uint8_t *mask;     // reset mask [W] or null
    float *obs;              // [W,N,width] or null
    float *rew;              // [W,N]
    uint8_t *done;           // [W,N]
    uint8_t *game_over;      // [W]
};

__device__ __forceinline__ double wrap_angle(double a) {
    while (a >= kPi) a -= 2.0 * kPi;
    while (a < -kPi) a += 2.0 * kPi;
    return a;
}

// wave-private LDS hand-off: LDS ops of one wavefront execute in order; the fences only stop the
// compiler from moving accesses across the point.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Philox4x32-10 (Salmon et al. SC'11) -- counter-based, so a world's scenario depends only on
// (seed, global world id, episode), never on the launch geometry or on sharding.
struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}
__device__ __forceinline__ double u01(uint32_t r) { return (double)(r >> 8) * (1.0 / 16777216.0); }

// per-lane agent registers
struct Agent {
    double px, py, heading, t_rem, vx, vy;
    float gx, gy, radius, pref, speed;
    uint32_t flags;
};

template <int N>
struct Geometry {
    static constexpr int kWorldsPerWave = 64 / N;
    static constexpr int kLanes = kWorldsPerWave * N;  // active lanes per wavefront
};

// LDS carve per wavefront: 4 double[64] + 1 float[64] + obs tile float[kLanes*width]
__host__ __device__ constexpr int lds_floats_fixed() { return 64 * 2 * 4 + 64; }

// GEN v1 scenario generator (E2; own specification, see oracle/cavoid_oracle.py generate_world)
template <int N>
__device__ __forceinline__ void generate_agent(const KCfg &c, uint32_t gw, uint32_t ep, int i, Agent &a) {
    const U4 r = philox4x32(gw, ep, 0u, 0u, c.seed_lo, c.seed_hi);
    const int span = c.gen_max_agents - c.gen_min_agents + 1;
    const int n = c.gen_min_agents + (int)(r.x % (uint32_t)span);
    if (i >= n) {
        a.px = a.py = a.heading = a.t_rem = a.vx = a.vy = 0.0;
        a.gx = a.gy = a.radius = a.pref = a.speed = 0.0f;
        a.flags = 0u;
        return;
    }
    const double base = fmax(4.0, 0.7 * n), ring = base * (1.0 + u01(r.y)), phase = u01(r.z);
    const U4 p = philox4x32(gw, ep, 1u, (uint32_t)i, c.seed_lo, c.seed_hi);
    const U4 q = philox4x32(gw, ep, 2u, (uint32_t)i, c.seed_lo, c.seed_hi);
    a.radius = (float)(0.2 + 0.6 * u01(p.x));
    a.pref = (float)(0.5 + 1.5 * u01(p.y));
    const double turn = phase + (i + (u01(p.z) - 0.5) * 2.0 * c.gen_angle_jitter) / n;
    const double theta = 2.0 * kPi * turn;
    double sn, cs;
    sincos(theta, &sn, &cs);
    a.px = ring * cs;
    a.py = ring * sn;
    a.gx = (float)(-a.px + (u01(q.x) - 0.5) * 2.0 * c.gen_goal_jitter);
    a.gy = (float)(-a.py + (u01(q.y) - 0.5) * 2.0 * c.gen_goal_jitter);
    uint32_t pol = 0u;
    if (i > 0 && u01(q.z) < c.gen_nonlearning) pol = u01(q.w) < c.gen_static ? 1u : 2u;
    const double tx = (double)a.gx - a.px, ty = (double)a.gy - a.py;
    const double dxg = a.px - (double)a.gx, dyg = a.py - (double)a.gy;
    const double straight = (sqrt(dxg * dxg + dyg * dyg) - c.near_goal) / (double)a.pref;
    a.heading = atan2(ty, tx);
    a.t_rem = fmax(c.max_time_ratio * straight, c.dt);
    a.vx = a.vy = 0.0;
    a.speed = 0.0f;
    a.flags = CAVOID_F_PRESENT | (pol == 0u ? CAVOID_F_LEARNING : 0u) | (pol << CAVOID_F_POLICY_SHIFT);
}

// Ego frame of one host (x axis -> goal).
struct Ego { double dist, prll_x, prll_y, orth_x, orth_y, heading_ego; };
__device__ __forceinline__ Ego ego_frame(const Agent &a) {
    Ego e;
    const double tx = (double)a.gx - a.px, ty = (double)a.gy - a.py;
    e.dist = sqrt(tx * tx + ty * ty);
    if (e.dist > 1e-8) { e.prll_x = tx / e.dist; e.prll_y = ty / e.dist; }
    else { e.prll_x = tx; e.prll_y = ty; }
    e.orth_x = -e.prll_y;
    e.orth_y = e.prll_x;
    e.heading_ego = wrap_angle(a.heading - atan2(e.prll_y, e.prll_x));
    return e;
}

__device__ __forceinline__ double time_to_impact(double rx, double ry, double vx, double vy, double R) {
    const double cc = rx * rx + ry * ry - R * R;
    if (cc <= 0.0) return 0.0;
    const double aa = vx * vx + vy * vy, bb = rx * vx + ry * vy;
    if (aa < 1e-10 || bb <= 0.0) return INFINITY;
    const double disc = bb * bb - aa * cc;
    if (disc < 0.0) return INFINITY;
    return (bb - sqrt(disc)) / aa;
}

// E6 + E9 for the lane's host agent.  All 64 lanes of the wavefront must call this together.
//   lds_*: wave-private staging arrays; tile: wave-private obs tile [kLanes][width]
//   emit: this lane rewrites its obs row (false: keep what the tile holds)
// Outputs hit / min_gap feed the reward (E7).
template <int N>
__device__ __forceinline__ void sense_world(const KCfg &c, const Agent &a, const Ego &e, int lane, int i, int base,
                                            bool active, bool emit, double *lds_px, double *lds_py,
                                            double *lds_vx, double *lds_vy, float *lds_r, float *tile,
                                            bool &hit, double &min_gap) {
    const bool present = active && (a.flags & CAVOID_F_PRESENT);
    lds_px[lane] = a.px;
    lds_py[lane] = a.py;
    lds_vx[lane] = a.vx;
    lds_vy[lane] = a.vy;
    lds_r[lane] = present ? a.radius : -1.0f;     // radius < 0 marks an absent row
    wave_lds_sync();

    const int M = c.max_other, width = c.width;
    const double ri = (double)a.radius;
    double key0[N], key1[N], key2[N];
    uint32_t valid = 0u;
    hit = false;
    min_gap = INFINITY;
    const bool tti_sort = c.sort_method == CAVOID_SORT_TIME_TO_IMPACT;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float rjf = lds_r[base + j];
        const double rj = (double)rjf;
        const double rx = lds_px[base + j] - a.px, ry = lds_py[base + j] - a.py;
        const double d = sqrt(rx * rx + ry * ry);
        const bool other = present && (j != i) && (rjf >= 0.0f);
        // E6: unordered-pair gap d - (r_lo + r_hi); the sum is commutative so either end agrees
        const double gap_c = d - (ri + rj);
        if (other) {
            min_gap = fmin(min_gap, gap_c);
            hit = hit || (gap_c <= c.collision_dist);
        }
        // E9 sort criteria: gap rounded to centimetres, then lateral offset
        const double gap_s = d - ri - rj;
        const double p_orth = rx * e.orth_x + ry * e.orth_y;
        const double gr = rint(gap_s * 100.0);      // order-isomorphic to rint(.)/100
        const bool seen = other && !(d > c.horizon);
        if (tti_sort) {
            const double tti = time_to_impact(rx, ry, a.vx - lds_vx[base + j], a.vy - lds_vy[base + j], ri + rj);
            key0[j] = -tti; key1[j] = -gr; key2[j] = p_orth;
        } else {
            key0[j] = -gr; key1[j] = p_orth; key2[j] = 0.0;
        }
        valid |= seen ? (1u << j) : 0u;
    }

    // stable ranks by counting: pos = number of seen agents strictly before j in far->near order
    const int m = __popc(valid);
    const int first = m > M ? m - M : 0;
    const int kept = m - first;
    int slot[N];
    uint32_t keep = 0u;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        int pos = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if (k == j) continue;
            const bool lt = (key0[k] < key0[j]) ||
                            (key0[k] == key0[j] && (key1[k] < key1[j] || (key1[k] == key1[j] && (key2[k] < key2[j] || (key2[k] == key2[j] && k < j)))));
            pos += (lt && ((valid >> k) & 1u)) ? 1 : 0;
        }
        slot[j] = pos - first;
        keep |= (((valid >> j) & 1u) && pos >= first) ? (1u << j) : 0u;
    }
    if (c.sort_method == CAVOID_SORT_CLOSEST_FIRST) {
        // re-rank the kept set near->far on (+gap, p_orth); full ties keep index order
#pragma unroll
        for (int j = 0; j < N; ++j) {
            int pos = 0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                if (k == j) continue;
                const bool lt = (key0[k] > key0[j]) ||
                                (key0[k] == key0[j] && (key1[k] < key1[j] || (key1[k] == key1[j] && k < j)));
                pos += (lt && ((keep >> k) & 1u)) ? 1 : 0;
            }
            slot[j] = pos;
        }
    }

    if (emit && active) {
        float *row = tile + lane * width;
        for (int k = 0; k < width; ++k) row[k] = 0.0f;
        if (present) {
            row[0] = (a.flags & CAVOID_F_LEARNING) ? 1.0f : 0.0f;
            row[1] = (float)kept;
            row[2] = (float)e.dist;
            row[3] = (float)e.heading_ego;
            row[4] = a.pref;
            row[5] = a.radius;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (!((keep >> j) & 1u)) continue;
                const double rj = (double)lds_r[base + j];
                const double rx = lds_px[base + j] - a.px, ry = lds_py[base + j] - a.py;
                const double ovx = lds_vx[base + j], ovy = lds_vy[base + j];
                float *f = row + 6 + 7 * slot[j];
                f[0] = (float)(rx * e.prll_x + ry * e.prll_y);
                f[1] = (float)(rx * e.orth_x + ry * e.orth_y);
                f[2] = (float)(ovx * e.prll_x + ovy * e.prll_y);
                f[3] = (float)(ovx * e.orth_x + ovy * e.orth_y);
                f[4] = (float)rj;
                f[5] = (float)(ri + rj);
                f[6] = (float)(sqrt(rx * rx + ry * ry) - ri - rj);
            }
        }
    }
    wave_lds_sync();
}

// Coalesced write-out of the wave's obs tile: n_floats contiguous floats starting at dst.
__device__ __forceinline__ void flush_tile(const float *tile, float *dst, int n_floats, int lane) {
    if ((n_floats & 3) == 0 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
        const float4 *src4 = reinterpret_cast<const float4 *>(tile);
        float4 *dst4 = reinterpret_cast<float4 *>(dst);
        for (int k = lane; k < (n_floats >> 2); k += 64) dst4[k] = src4[k];
    } else {
        for (int k = lane; k < n_floats; k += 64) dst[k] = tile[k];
    }
}

enum : int { MODE_STEP = 0, MODE_STEP_AUTORESET = 1, MODE_OBSERVE = 2, MODE_RESET = 3 };

template <int N, int MODE>
__global__ void __launch_bounds__(256) env_kernel(const KCfg c, const KState s, const KIO io) {
    using G = Geometry<N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave_in_block = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int width = c.width;
    const int tile_floats = (G::kLanes * width + 3) & ~3;
    const int per_wave_floats = lds_floats_fixed() + tile_floats;
    float *wbase = reinterpret_cast<float *>(smem) + (size_t)wave_in_block * per_wave_floats;
    double *lds_px = reinterpret_cast<double *>(wbase);
    double *lds_py = lds_px + 64, *lds_vx = lds_py + 64, *lds_vy = lds_vx + 64;
    float *lds_r = reinterpret_cast<float *>(lds_vy + 64);
    float *tile = lds_r + 64;

    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block;
    const int64_t w0 = wave * G::kWorldsPerWave;           // first world of this wavefront
    const int lw = lane / N, i = lane - lw * N;
    const int64_t w = w0 + lw;
    const bool active = lane < G::kLanes && w < c.num_worlds;
    const int base = lane < G::kLanes ? lw * N : 0;     // first lane of this lane's world
    const int64_t a_idx = w * N + i;                       // == w0*N + lane: contiguous per wave

    Agent a;
    a.px = a.py = a.heading = a.t_rem = a.vx = a.vy = 0.0;
    a.gx = a.gy = a.radius = a.pref = a.speed = 0.0f;
    a.flags = 0u;
    uint32_t episode = 0u;
    bool regenerate = false;
    if (active) {
        if (MODE == MODE_RESET) {
            regenerate = io.mask == nullptr || io.mask[w] != 0;
            episode = s.episode[w] + (regenerate ? 1u : 0u);
        } else if (MODE == MODE_STEP_AUTORESET) {
            episode = s.episode[w];
        }
        if (!(MODE == MODE_RESET && regenerate)) {
            a.px = s.px[a_idx]; a.py = s.py[a_idx]; a.heading = s.heading[a_idx]; a.t_rem = s.t_rem[a_idx];
            a.gx = s.gx[a_idx]; a.gy = s.gy[a_idx]; a.radius = s.radius[a_idx]; a.pref = s.pref[a_idx];
            a.flags = s.flags[a_idx];
            if (MODE == MODE_OBSERVE || MODE == MODE_RESET) a.speed = s.speed[a_idx];
        }
    }

    const uint32_t flags_in = a.flags;
    const bool present = active && (flags_in & CAVOID_F_PRESENT);
    const bool done_in = (flags_in & CAVOID_F_DONE_MASK) != 0u;

    if (MODE == MODE_RESET) {
        if (active && regenerate) generate_agent<N>(c, (uint32_t)(c.world_offset + w), episode, i, a);
    }
    if (MODE == MODE_OBSERVE || MODE == MODE_RESET) {
        if (present || (MODE == MODE_RESET && regenerate)) {
            double sn, cs;
            sincos(a.heading, &sn, &cs);
            a.vx = (double)a.speed * cs;
            a.vy = (double)a.speed * sn;
        }
    }

    if (MODE == MODE_STEP || MODE == MODE_STEP_AUTORESET) {
        // ---- E4 decode + E5 dynamics --------------------------------------------------------
        if (present && done_in) {
            if (flags_in & CAVOID_F_AT_GOAL) a.flags |= CAVOID_F_WAS_AT_GOAL;
            if (flags_in & CAVOID_F_IN_COLL) a.flags |= CAVOID_F_WAS_IN_COLL;
            a.vx = a.vy = 0.0;
            a.speed = 0.0f;
        } else if (present) {
            const uint32_t pol = (flags_in >> CAVOID_F_POLICY_SHIFT) & 3u;
            double a0 = 0.0, a1 = 0.0;
            if (pol == 0u) {
                if (io.cont) { a0 = (double)io.cont[2 * a_idx]; a1 = (double)io.cont[2 * a_idx + 1]; }
                else {
                    int act = io.actions[a_idx];
                    act = act < 0 ? 0 : (act >= c.num_actions ? c.num_actions - 1 : act);
                    a0 = (double)a.pref * c.action_table[2 * act];
                    a1 = c.action_table[2 * act + 1];
                }
            } else if (pol == 2u) {
                const Ego e0 = ego_frame(a);            // non-cooperative: pref speed straight at the goal
                a0 = (double)a.pref;
                a1 = -e0.heading_ego;
            }
            if (c.actions_fp32) { a0 = (double)(float)a0; a1 = (double)(float)a1; }
            if (c.dynamics == CAVOID_DYN_HOLONOMIC) {
                const double sp = sqrt(a0 * a0 + a1 * a1);
                if (sp > 0.0) a.heading = atan2(a1, a0);
                a.px += a0 * c.dt; a.py += a1 * c.dt;
                a.vx = a0; a.vy = a1; a.speed = (float)sp;
            } else {
                double dh = a1;
                if (c.dynamics == CAVOID_DYN_UNICYCLE_MAX_TURN) {
                    const double rate = fmin(fmax(dh / c.dt, -c.max_turn_rate), c.max_turn_rate);
                    dh = rate * c.dt;
                }
                const double h = wrap_angle(dh + a.heading);
                double sn, cs;
                sincos(h, &sn, &cs);
                a.px += a0 * cs * c.dt; a.py += a0 * sn * c.dt;
                a.vx = a0 * cs; a.vy = a0 * sn; a.speed = (float)a0; a.heading = h;
            }
        }
    }

    Ego e = ego_frame(a);

    if (MODE == MODE_STEP || MODE == MODE_STEP_AUTORESET) {
        if (present && !done_in) {
            const double dx = a.px - (double)a.gx, dy = a.py - (double)a.gy;
            if (dx * dx + dy * dy <= c.near_goal_sq) a.flags |= CAVOID_F_AT_GOAL;
            a.t_rem -= c.dt;
            if (c.timeout_enabled && a.t_rem <= 0.0) a.flags |= CAVOID_F_RAN_OUT;
        }
    }

    // ---- E6 + E9 ----------------------------------------------------------------------------
    bool hit;
    double min_gap;
    sense_world<N>(c, a, e, lane, i, base, active, true, lds_px, lds_py, lds_vx, lds_vy, lds_r, tile, hit, min_gap);

    if (MODE == MODE_STEP || MODE == MODE_STEP_AUTORESET) {
        // ---- E7 rewards, E8 done -------------------------------------------------------------
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
        // game over <=> no learning agent of the world is still running
        const unsigned long long running = __ballot(present && (a.flags & CAVOID_F_LEARNING) && !done);
        const unsigned long long wmask = ((N == 64) ? ~0ull : ((1ull << N) - 1ull)) << base;
        const bool game_over = (running & wmask) == 0ull;
        if (active) {
            io.rew[a_idx] = (float)r;
            io.done[a_idx] = done ? 1 : 0;
            if (i == 0) io.game_over[w] = game_over ? 1 : 0;
        }

        bool restart = false;
        if (MODE == MODE_STEP_AUTORESET) {
            restart = active && game_over;
            if (__ballot(restart) != 0ull) {              // wave-uniform: some world of this tile restarts
                if (restart) {
                    episode += 1u;
                    generate_agent<N>(c, (uint32_t)(c.world_offset + w), episode, i, a);
                    e = ego_frame(a);
                }
                bool hit2;
                double gap2;
                sense_world<N>(c, a, e, lane, i, base, active, restart, lds_px, lds_py, lds_vx, lds_vy, lds_r, tile, hit2, gap2);
            }
        }
        if (restart) {                                   // fresh episode: every field of every row
            s.px[a_idx] = a.px; s.py[a_idx] = a.py; s.heading[a_idx] = a.heading; s.t_rem[a_idx] = a.t_rem;
            s.gx[a_idx] = a.gx; s.gy[a_idx] = a.gy; s.radius[a_idx] = a.radius; s.pref[a_idx] = a.pref;
            s.speed[a_idx] = a.speed; s.flags[a_idx] = a.flags;
            if (i == 0) s.episode[w] = episode;
        } else if (present) {
            if (!done_in) {                              // frozen agents keep pos / heading / time
                s.px[a_idx] = a.px; s.py[a_idx] = a.py; s.heading[a_idx] = a.heading; s.t_rem[a_idx] = a.t_rem;
            }
            s.speed[a_idx] = a.speed;
            s.flags[a_idx] = a.flags;
        }
    }

    if (MODE == MODE_RESET) {
        if (active && regenerate) {
            s.px[a_idx] = a.px; s.py[a_idx] = a.py; s.heading[a_idx] = a.heading; s.t_rem[a_idx] = a.t_rem;
            s.gx[a_idx] = a.gx; s.gy[a_idx] = a.gy; s.radius[a_idx] = a.radius; s.pref[a_idx] = a.pref;
            s.speed[a_idx] = a.speed; s.flags[a_idx] = a.flags;
            if (i == 0) s.episode[w] = episode;
        }
    }

    if (io.obs) {
        int64_t worlds_here = c.num_worlds - w0;
        if (worlds_here > G::kWorldsPerWave) worlds_here = G::kWorldsPerWave;
        if (worlds_here > 0)
            flush_tile(tile, io.obs + w0 * N * width, (int)worlds_here * N * width, lane);
    }
}

}  // namespace cavoid
