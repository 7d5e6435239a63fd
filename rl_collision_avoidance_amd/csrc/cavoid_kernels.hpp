// cavoid_kernels.hpp -- hand-written gfx950 (CDNA4) kernels of the batched env.step hot path.
//
// Replaces, for W worlds at once, what one reference `self.game.step(action)` call does for one
// world (ga3c/GA3C/Environment.py:112): E4 action decode, E5 dynamics, E6 pairwise gaps /
// collisions, E7 rewards, E8 done flags, E9 ego-frame neighbour-sorted observation
// (obs layout ga3c/GA3C/Config.py:40,72-76).
//
// Mapping (DESIGN.md "Kernels"):
//   * one lane per agent ("host"), one 64-lane wavefront per tile of up to floor(64/N) whole worlds
//     (fewer for small batches, so that every SIMD has wavefronts to interleave);
//     flat agent index a = w*N + i, so a wavefront's agents are CONTIGUOUS in every SoA field
//     and each field is one coalesced global_load per wavefront;
//   * post-move agent state is staged in LDS, wave-private, as one 32-byte record per agent (position | float32
//     velocity, radius), and the O(N^2) neighbour pass reads the other agents of the lane's world from there
//     (same-world lanes read the same address -> LDS broadcast); from 6 agents per world on the records of a world
//     are staged twice back to back, so that every neighbour address is the lane's own plus an immediate;
//   * the pair pass makes, per neighbour, the collision test AND what the observation needs of it (gap, sort key,
//     rotated features) while the neighbour's record is in registers;
//   * neighbour ordering by counting ranks (O(N^2) integer compares, no data-dependent control flow);
//   * the [agents, 1+D] observation tile is assembled in LDS and leaves as 16-byte coalesced stores (a
//     lane-owns-a-row store would touch 64 cache lines per instruction) -- non-temporal ones where nobody in the
//     launch reads the rows again (per-step output slots; batches whose observation exceeds the L2);
//   * float64 arithmetic throughout (the reference env is NumPy float64; flags are threshold
//     tests that flip on fp32 rounding), compiled with -ffp-contract=off so that everything that
//     decides a flag, a reward branch or a sort order is the oracle's operation sequence; what
//     differs is sin/cos/atan2 (<=1 ulp) and, on observation-only values, x*(1/d) for x/d;
//   * a restart (auto-reset) is decided BEFORE the observation pass, so a step assembles
//     observations exactly once; restarted worlds take their scenario from a pre-generated
//     pool (a gather) instead of running the generator on the step's critical path.
//   No MFMA: there is no dense contraction anywhere on this path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cavoid.h"

namespace cavoid {

// Development aid (never in the product build): -DCAVOID_TRACE makes lane 0 of every wavefront
// stamp the shader clock at the phase boundaries into g_trace[wave*16 + k] (tools/trace_step.py).
#ifdef CAVOID_TRACE
__device__ unsigned long long *g_trace = nullptr;
#define CAVOID_STAMP(k)                                                                \
    do {                                                                               \
        if (lane == 0 && g_trace) g_trace[wave * 16 + (k)] = (unsigned long long)clock64(); \
    } while (0)
#else
#define CAVOID_STAMP(k) do { } while (0)
#endif

// branch-probability hint: the block is laid out behind the hot path (the step loop of the larger instantiations is tens of
// KB of code, most of it rarely taken paths; the instruction cache is 64 KB per two CUs)
#define CAVOID_RARE(x) __builtin_expect(!!(x), 0)
#ifndef CAVOID_SKIP
#define CAVOID_SKIP 0   /* development: bit mask of phases left out (wrong results; instruction-count ablations) */
#endif

constexpr double kPi = 3.14159265358979323846;
constexpr int kRelayMaxConsumers = 4;   // observation wavefronts per tile of env_relay_kernel (cavoid_relay.hpp)

// Create-time constants of the rarely taken paths (scenario generators, a fresh agent's time budget, the ORCA policy, the
// max-turn-rate dynamics): in device memory behind KCfg::cold, read with scalar loads where they are used.  By value they were
// 42 more scalar registers live across the whole step (the kernel-argument struct is loaded up front), and the step's hot
// constants paid for them in spills to vector-register lanes (v_readlane per use).
struct KCold {
    double budget_offset, max_time_ratio, max_turn_rate;
    double gen_nonlearning, gen_static, gen_goal_jitter, gen_angle_jitter;
    double gen_rvo, gen_box_small_lo, gen_box_small_hi, gen_box_large_lo, gen_box_large_hi, gen_min_trip;
    double rvo_inv_horizon, rvo_collab, rvo_radius_scale, rvo_max_dh;
    double gen_frozen;           // P(frozen-network agent) among the scripted ones
    int32_t gen_min_agents, gen_max_agents, gen_box_large_from, pad;
};

// kernel-argument POD (by value).  The action table lives in device memory (per-lane index).
struct KCfg {
    double dt, near_goal_sq, collision_dist, close_range;
    double r_goal, r_coll, r_close, r_step, close_slope, clip_lo, clip_hi, horizon;
    const KCold *__restrict__ cold;
    int32_t max_other, width, sort_method, dynamics, actions_fp32, timeout_enabled, num_actions;
    int32_t gen_mode, rvo_enabled;
    uint32_t pool_epoch;         // the pool holds generator worlds 0..P-1 of this episode index
    int32_t pool_size;           // 0: restarts run the generator in-kernel; >0: gather from the pool
    int32_t ahead;               // R > 0: the "pool" is the look-ahead ring (slot ep % R of this world: exact fresh scenarios), not the hashed pool
    int32_t prefetch_pool;       // latency mode (small batches): every lane pre-loads its next pool entry
    int32_t tile_rows;           // rows of the LDS obs tile (one pass = tile_rows agents' rows)
    int32_t wpw;                 // worlds per wavefront, 1..floor(64/N): small batches spread over more, emptier wavefronts
    int32_t rvo_lds_floats;      // per-wavefront LDS floats of the ORCA line scratch (0 unless rvo_enabled)
    int32_t park_floats;         // least size of the tile region (parked sort keys / gaps, float64 velocities, field-major scratch)
    int32_t evaluate_mode;       // game over needs EVERY agent done (EVALUATE_MODE), not only the learning ones
    int32_t stream_obs;          // the batch's observation is larger than the L2 can hold for the next kernel (> 16 MB): one-step launches stream their rows out too
    uint32_t switches;           // kSw* bits: the rarely flipped U-switches in ONE word, tested by uniform branches that sit OUTSIDE the
                                 // unrolled hot loops (measured on the one-step launch at 4 x 8192, same box: separate scalars tested
                                 // per neighbour 7.4 us; pinned in scalar registers at the top 7.0; this form 6.3 -- round 2: 6.5)
    uint32_t seed_lo, seed_hi;
    int64_t num_worlds, world_offset;
    const double *action_table;  // [num_actions][2]
};

enum : uint32_t {
    kSwSkipDonePairs = 1u,       // U4 flipped: a pair with an agent that was done before the step takes no part in E6
    kSwExactGap = 2u,            // U7a flipped: neighbours ordered by the exact gap (default: rounded to centimetres)
    kSwIndexTie = 4u,            // U7b flipped: ties by agent index alone (default: lateral offset, then index)
    kSwWrapClosed = 8u,          // U2 flipped: angles wrap to (-pi, pi] (default: [-pi, pi))
};
struct KState {
    double *px, *py, *heading, *t_rem;
    float *gx, *gy, *radius, *pref, *speed;
    uint32_t *flags;
    uint32_t *episode;  // [W]
};

// One pre-generated agent of the scenario pool: a 64-byte record, because the pool is only ever GATHERED
// (one random entry per restarting world) -- an array-of-records costs one cache line per agent where the
// field-major world buffer layout would cost one line per field.
struct alignas(64) PoolRec {
    double px, py, heading, t_rem;
    float gx, gy, radius, pref;
    uint32_t flags;
    uint32_t pad[3];
};
static_assert(sizeof(PoolRec) == 64, "pool record must be one 64-byte line");

struct KIO {
    PoolRec *pool_out;       // MODE_RESET only: write the generated agents here (pool fill) instead of the world buffer
    const int32_t *actions;  // [n_steps][W,N] (slice t at actions + t*action_stride) or null
    const float *cont;       // continuous actions [n_steps][W,N,2] (slice t at cont + t*action_stride FLOATS) or null
    const uint8_t *mask;     // reset mask [W] or null
    float *obs;              // [W,N,obs_stride] or null
    float *rew;              // [W,N]; null in packed mode
    uint8_t *done;           // [W,N]; null in packed mode
    uint8_t *game_over;      // [W]
    int64_t action_stride;   // elements (int32 of `actions`, floats of `cont`) between the action slices of consecutive steps
    int64_t out_step_stride; // multi-step launches: step t writes its outputs into slot t -- obs + t*S*N*obs_stride, rew / done + t*S*N,
                             // game_over + t*S with S = out_step_stride WORLDS (>= num_worlds); 0: every step overwrites slot 0
    int32_t n_steps;         // steps taken by ONE launch (MODE_STEP_AUTORESET; 1 elsewhere)
    int32_t obs_stride;      // floats per output row: width, or width + 2 in packed mode
    int32_t packed;          // != 0: reward and done (as 0.0f / 1.0f) are columns width, width+1 of the agent's row
};

// wrap to [-pi, pi) by repeated +-2*pi, exactly the oracle's `while` loops: one branch-free fold each way covers every
// table action (|heading + delta| < 3*pi); anything still outside (huge continuous actions) takes the loops, as a
// wave-uniform branch.  A fold that does not apply leaves the value untouched, so the results are bit-identical.
// U2: the two conventions -- [-pi, pi) (default) and (-pi, pi] -- differ at ONE point: what the first calls -pi the second calls
// pi (pi - 2 pi and -pi + 2 pi are exact).  So the closed-end form is the default fold (literal constants on the step's dependent
// chain) plus a fix-up of that one value behind a uniform, rarely taken branch.
__device__ __forceinline__ double wrap_closed_fixup(double a, uint32_t switches) {
    if (CAVOID_RARE(switches & kSwWrapClosed)) a = a == -kPi ? kPi : a;
    return a;
}
__device__ __forceinline__ double wrap_angle(double a, uint32_t switches) {
    a = a >= kPi ? a - 2.0 * kPi : a;
    a = a < -kPi ? a + 2.0 * kPi : a;
    if (CAVOID_RARE(__ballot(a >= kPi || a < -kPi) != 0ull)) {
        while (a >= kPi) a -= 2.0 * kPi;
        while (a < -kPi) a += 2.0 * kPi;
    }
    return wrap_closed_fixup(a, switches);
}
// one fold each way: enough for a difference of two angles of magnitude <= pi
__device__ __forceinline__ double wrap_once(double h, uint32_t switches) {
    h = h >= kPi ? h - 2.0 * kPi : h;
    h = h < -kPi ? h + 2.0 * kPi : h;
    return wrap_closed_fixup(h, switches);
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

// One staged agent -- what the lanes of the other agents of its world read of it after the move: 32 bytes, so a neighbour
// costs two 16-byte LDS reads at ONE address (position | float32 velocity, radius) where field-major arrays cost five reads at
// two.  The velocity is the float32 the observation features are made of; the float64 velocities, which only the
// time-to-impact ordering and the ORCA policy read, live in the (then idle) tile region.  r < 0 marks an absent row.
struct alignas(16) StageRec { double px, py; float vxf, vyf, r, pad; };
static_assert(sizeof(StageRec) == 32, "two 16-byte reads per neighbour");
// N >= kRingDoubleFromN: a world's records are staged TWICE, back to back ([a0 .. aN-1 | a0 .. aN-1]), so that the N-1 others of
// agent i in ring order are simply the N-1 records behind its own: every neighbour address is the lane's own record address
// plus a compile-time offset -- no index arithmetic at all in the pair pass and the row pass (it was 5-6 vector instructions
// per neighbour and pass).  Smaller N keep one copy (the LDS of four wavefronts per SIMD is full at N = 4) and pay a compare
// and a select per neighbour for the wrap.
constexpr int kRingDoubleFromN = 6;
// LDS carve: per workgroup double[64] (the action table, 32 x 2: every wavefront writes the same values, so no
// workgroup barrier is needed), then per wavefront the staged records (64, or 128 when doubled) + the obs tile
// float[tile_rows * obs_stride] (whose region also holds, while no rows are in it, the parked keys, the float64 velocities and
// the scratch arrays of the ORCA policy / the box generator)
__host__ __device__ constexpr int lds_floats_block() { return 64 * 2; }
__host__ __device__ constexpr int lds_floats_fixed(int n) { return (n >= kRingDoubleFromN ? 128 : 64) * (int)(sizeof(StageRec) / sizeof(float)); }
// scratch the tile region must hold in instantiations that stage field-major arrays there (ORCA pre-move state, box generator):
// four double[64] + one float[64]
__host__ __device__ constexpr int lds_floats_scratch() { return 64 * 2 * 4 + 64; }

// sin/cos for |x| up to a few thousand: 2-term Cody-Waite reduction by pi/2 and the degree-13/14
// kernels of the classic fdlibm sin/cos (max error ~1 ulp).  Headings live in [-pi, pi), so the
// huge-argument path of a general sincos is dead weight on the step's critical path.
__device__ __forceinline__ void sincos_bounded(double x, double *sn, double *cs) {
    const double k = rint(x * 0.63661977236758134308);            // 2/pi
    double r = __builtin_fma(-k, 1.57079632673412561417e+00, x);   // pi/2, leading 33 bits
    r = __builtin_fma(-k, 6.07710050650619224932e-11, r);          // pi/2 tail
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
    ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
    ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
    ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
    ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
    const double s = __builtin_fma(r * z, ps, r);
    double pc = -1.13596475577881948265e-11;
    pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
    pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
    pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
    pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
    pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
    const double c = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
    const int q = (int)k & 3;
    const double s_out = (q & 1) ? c : s, c_out = (q & 1) ? s : c;
    *sn = (q & 2) ? -s_out : s_out;
    *cs = ((q + 1) & 2) ? -c_out : c_out;
}

// scripted-policy draw shared by both generators (oracle: _draw_policy)
__device__ __forceinline__ uint32_t draw_policy(const KCfg &c, const U4 &q, int i) {
    if (i > 0 && u01(q.z) < c.cold->gen_nonlearning) {
        const double u = u01(q.w);
        if (u < c.cold->gen_static) return 1u;
        if (u < c.cold->gen_static + c.cold->gen_rvo) return 3u;
        return u < c.cold->gen_static + c.cold->gen_rvo + c.cold->gen_frozen ? 4u : 2u;
    }
    return 0u;
}

// heading at the goal, time budget, flags of a freshly placed agent (oracle: Agent.__init__ / place_agent)
__device__ __forceinline__ void finish_agent(const KCfg &c, uint32_t pol, Agent &a) {
    const double tx = (double)a.gx - a.px, ty = (double)a.gy - a.py;
    const double dxg = a.px - (double)a.gx, dyg = a.py - (double)a.gy;
    const double straight = (sqrt(dxg * dxg + dyg * dyg) - c.cold->budget_offset) / (double)a.pref;   // U11 (x - 0.0 is exact)
    a.heading = atan2(ty, tx);
    a.t_rem = fmax(c.cold->max_time_ratio * straight, c.dt);
    a.vx = a.vy = 0.0;
    a.speed = 0.0f;
    a.flags = CAVOID_F_PRESENT | (pol == 0u ? CAVOID_F_LEARNING : 0u) | (pol << CAVOID_F_POLICY_SHIFT);
}

__device__ __forceinline__ void absent_agent(Agent &a) {
    a.px = a.py = a.heading = a.t_rem = a.vx = a.vy = 0.0;
    a.gx = a.gy = a.radius = a.pref = a.speed = 0.0f;
    a.flags = 0u;
}

// GEN v1 scenario generator (E2; own specification, see oracle/cavoid_oracle.py generate_world): a ring, antipodal goals
template <int N>
__device__ __forceinline__ void generate_agent(const KCfg &c, uint32_t gw, uint32_t ep, int i, Agent &a) {
    const U4 r = philox4x32(gw, ep, 0u, 0u, c.seed_lo, c.seed_hi);
    const int span = c.cold->gen_max_agents - c.cold->gen_min_agents + 1;
    const int n = c.cold->gen_min_agents + (int)(r.x % (uint32_t)span);
    if (i >= n) { absent_agent(a); return; }
    const double base = fmax(4.0, 0.7 * n), ring = base * (1.0 + u01(r.y)), phase = u01(r.z);
    const U4 p = philox4x32(gw, ep, 1u, (uint32_t)i, c.seed_lo, c.seed_hi);
    const U4 q = philox4x32(gw, ep, 2u, (uint32_t)i, c.seed_lo, c.seed_hi);
    a.radius = (float)(0.2 + 0.6 * u01(p.x));
    a.pref = (float)(0.5 + 1.5 * u01(p.y));
    const double turn = phase + (i + (u01(p.z) - 0.5) * 2.0 * c.cold->gen_angle_jitter) / n;
    const double theta = 2.0 * kPi * turn;
    double sn, cs;
    sincos_bounded(theta, &sn, &cs);
    a.px = ring * cs;
    a.py = ring * sn;
    a.gx = (float)(-a.px + (u01(q.x) - 0.5) * 2.0 * c.cold->gen_goal_jitter);
    a.gy = (float)(-a.py + (u01(q.y) - 0.5) * 2.0 * c.cold->gen_goal_jitter);
    finish_agent(c, draw_policy(c, q, i), a);
}

// GEN v2 (E2; oracle: generate_world, mode 1): uniform boxes, the agents of a world placed ONE AFTER THE OTHER by
// rejection sampling against those already placed.  Wave-cooperative (MODE_RESET only, every lane of the wavefront
// calls it): in round k the lanes holding agent k of a `fresh` world draw until accepted, reading the agents 0..k-1 of
// their world from the LDS staging arrays (starts in lds_px / lds_py, goals in lds_vx / lds_vy, radii in lds_r), then
// publish their own placement; the (possibly grown) box side travels to the next round by a lane shuffle.
template <int N>
__device__ __forceinline__ void generate_world_v2(const KCfg &c, uint32_t gw, uint32_t ep, int i, int base, int lane, bool fresh,
                                                  double *lds_px, double *lds_py, double *lds_gx, double *lds_gy, float *lds_r,
                                                  Agent &a) {
    const U4 r = philox4x32(gw, ep, 0u, 0u, c.seed_lo, c.seed_hi);
    const int span = c.cold->gen_max_agents - c.cold->gen_min_agents + 1;
    const int n = c.cold->gen_min_agents + (int)(r.x % (uint32_t)span);
    const double lo = n < c.cold->gen_box_large_from ? c.cold->gen_box_small_lo : c.cold->gen_box_large_lo;
    const double hi = n < c.cold->gen_box_large_from ? c.cold->gen_box_small_hi : c.cold->gen_box_large_hi;
    double side = lo + (hi - lo) * u01(r.y);
    const U4 p = philox4x32(gw, ep, 1u, (uint32_t)i, c.seed_lo, c.seed_hi);
    const U4 q = philox4x32(gw, ep, 2u, (uint32_t)i, c.seed_lo, c.seed_hi);
    const float radius = (float)(0.2 + 0.6 * u01(p.x));
    const float pref = (float)(0.5 + 1.5 * u01(p.y));
    if (fresh) absent_agent(a);
    for (int round = 0; round < N; ++round) {                  // wave-uniform
        // the box side agent round-1 left behind (a crowded box grows), straight from that lane's register
        const double side_prev = round > 0 ? __shfl(side, base + round - 1) : side;
        if (fresh && i == round && i < n) {
            side = side_prev;
            double sx = 0.0, sy = 0.0;
            float gx = 0.f, gy = 0.f;
            for (int attempt = 0;;) {
                const U4 d = philox4x32(gw, ep, 3u + (uint32_t)attempt, (uint32_t)i, c.seed_lo, c.seed_hi);
                sx = side * (2.0 * u01(d.x) - 1.0); sy = side * (2.0 * u01(d.y) - 1.0);
                gx = (float)(side * (2.0 * u01(d.z) - 1.0)); gy = (float)(side * (2.0 * u01(d.w) - 1.0));
                const double tx = (double)gx - sx, ty = (double)gy - sy;
                bool ok = sqrt(tx * tx + ty * ty) >= c.cold->gen_min_trip;
                for (int j = 0; j < round; ++j) {
                    const double margin = ((double)radius + (double)lds_r[base + j]) + c.close_range;
                    const double ax = sx - lds_px[base + j], ay = sy - lds_py[base + j];
                    const double bx = (double)gx - lds_gx[base + j], by = (double)gy - lds_gy[base + j];
                    if (sqrt(ax * ax + ay * ay) < margin || sqrt(bx * bx + by * by) < margin) ok = false;
                }
                ++attempt;
                if (ok || attempt >= 100) break;
                if (attempt % 10 == 0) side = side * 1.01;
            }
            a.px = sx; a.py = sy; a.gx = gx; a.gy = gy; a.radius = radius; a.pref = pref;
            finish_agent(c, draw_policy(c, q, i), a);
            lds_px[lane] = sx; lds_py[lane] = sy; lds_gx[lane] = (double)gx; lds_gy[lane] = (double)gy; lds_r[lane] = radius;
        }
        wave_lds_sync();
    }
}

// Ego frame of one host (x axis -> goal).  (tx, ty) is the un-normalised goal direction: the
// lateral-offset sort key uses it directly (same ordering as the normalised p_orth).
struct Ego { double dist, tx, ty, prll_x, prll_y, heading_ego; };

// exact form: correctly-rounded sqrt, float64 atan2.  Used where the result feeds the STATE (the
// non-cooperative policy steers by -heading_ego).
__device__ __forceinline__ Ego ego_frame_exact(const KCfg &c, const Agent &a) {
    Ego e;
    e.tx = (double)a.gx - a.px;
    e.ty = (double)a.gy - a.py;
    e.dist = sqrt(e.tx * e.tx + e.ty * e.ty);
    const double inv = e.dist > 1e-8 ? 1.0 / e.dist : 1.0;
    e.prll_x = e.tx * inv;
    e.prll_y = e.ty * inv;
    e.heading_ego = wrap_once(a.heading - atan2(e.prll_y, e.prll_x), c.switches);   // |heading|, |atan2| <= pi: one fold each way
    return e;
}

// observation form: everything here ends in a float32 observation (tolerance 1e-5) and never in a
// flag, a reward branch, a sort key or the state, so the long float64 sqrt/div/atan2 chains are
// replaced by a Newton-refined v_rsq_f32 seed (relative error ~1e-14) and a float32 atan2
// (absolute error < 1e-6 rad) -- a third of the dependent latency of the exact form.
__device__ __forceinline__ Ego ego_from(const KCfg &c, double tx, double ty, double heading) {
    Ego e;
    e.tx = tx;
    e.ty = ty;
    const double ss = e.tx * e.tx + e.ty * e.ty;
    const bool tiny = !(ss > 1e-16);                                  // dist <= 1e-8: axes stay un-normalised
    double y = (double)__builtin_amdgcn_rsqf((float)ss);              // ~1e-7 relative
    y = y * __builtin_fma(-0.5 * ss, y * y, 1.5);                     // -> ~1e-14
    y = y * __builtin_fma(-0.5 * ss, y * y, 1.5);
    e.dist = tiny ? sqrt(ss) : ss * y;
    const double inv = tiny ? 1.0 : y;
    e.prll_x = e.tx * inv;
    e.prll_y = e.ty * inv;
    e.heading_ego = wrap_once(heading - (double)atan2f((float)e.ty, (float)e.tx), c.switches);
    return e;
}
__device__ __forceinline__ Ego ego_frame_obs(const KCfg &c, const Agent &a) { return ego_from(c, (double)a.gx - a.px, (double)a.gy - a.py, a.heading); }

__device__ __forceinline__ double time_to_impact(double rx, double ry, double vx, double vy, double R) {
    const double cc = rx * rx + ry * ry - R * R;
    if (cc <= 0.0) return 0.0;
    const double aa = vx * vx + vy * vy, bb = rx * vx + ry * vy;
    if (aa < 1e-10 || bb <= 0.0) return INFINITY;
    const double disc = bb * bb - aa * cc;
    if (disc < 0.0) return INFINITY;
    return (bb - sqrt(disc)) / aa;
}

// sqrt of a squared distance, bit for bit the device library's correctly rounded sqrt(double) for x = 0 and x >= 2^-767
// (anything a sum of two squares of position differences can be): the library's iteration -- rsq seed, one coupled
// Goldschmidt step, two residual corrections -- without its rescaling of tiny arguments (a compare, two ldexp and a select
// per call; the pair pass calls it N-1 times per agent and step).  x = 0 (two agents on one point) needs no select either: the
// seed is taken of max(x, 2^-1000) -- finite -- so g = x * y is an exact 0 and every correction leaves it there (the library
// returns x itself for 0); for x >= 2^-1000 the max is the identity.  x = +inf does not occur (positions are finite).
// tests/test_gpu_parity.py holds the flags to the oracle's.
__device__ __forceinline__ double sqrt_dist2(double x) {
    const double y = __builtin_amdgcn_rsq(fmax(x, 0x1p-1000));
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return g;
}

// Others of host i, in ring order: o = 0..N-2  ->  agent j = (i + 1 + o) mod N.  Iterating the N-1
// OTHERS (instead of all N agents with the host masked out) saves a sqrt and, with the symmetric
// rank update below, three quarters of the key comparisons.
template <int N>
struct Others {
    static constexpr int K = N > 1 ? N - 1 : 1;     // array extent (N == 1: one dummy, never valid)
};

__device__ __forceinline__ int other_index(int i, int o, int n) {
    const int j = i + 1 + o;
    return j >= n ? j - n : j;
}

// float32 -> uint32 whose unsigned order is the float order (-0 canonicalised to +0 first)
__device__ __forceinline__ uint32_t orderable(float f) {
    const uint32_t u = __float_as_uint(f + 0.0f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// Sort key of one neighbour, 63 bits: hi = 2^30 - bucket (far -> near is ASCENDING in hi), lo = the orderable float32
// lateral offset.  bucket = rint(gap*100) (order-isomorphic to round(gap, 2)); |gap| < 1e7 m keeps hi in [1, 2^31).
// A neighbour that is absent or beyond the sensing horizon gets a sentinel above every real key (and distinct per
// slot, in the hi word: its lo word is whatever the arithmetic left there), so the ranking needs no validity masks:
// sentinels simply sort last.
// (one 64-bit value: the ranking compares whole keys with v_cmp_lt_u64, which wants the halves in a register pair)
struct Key {
    uint64_t v;
    __device__ __forceinline__ uint32_t hi() const { return (uint32_t)(v >> 32); }
    __device__ __forceinline__ uint32_t lo() const { return (uint32_t)v; }
    __device__ __forceinline__ void set(uint32_t hi, uint32_t lo) { v = ((uint64_t)hi << 32) | lo; }
};
constexpr uint32_t kKeyBias = 1u << 30;
// sentinels: hi = kKeySentinel + slot (distinct per slot, above every real key: a real hi is < 2^31 - 16), lo = anything
constexpr uint32_t kKeySentinel = 0x7FFFFFF0u;
__device__ __forceinline__ bool key_is_sentinel(uint32_t hi) { return hi >= kKeySentinel; }
__device__ __forceinline__ int key_bucket(const Key &k) { return (int)(kKeyBias - k.hi()); }

// How a lane finds the staged post-move state of the OTHER agents of its world, in ring order (o = 0..N-2 -> agent
// (i + 1 + o) mod N).  Two layouts behind one interface:
struct OtherState { double px, py; float vxf, vyf, r; };

// env_tile's wave-private records (StageRec; doubled from kRingDoubleFromN agents on)
template <int N>
struct RingStage {
    static constexpr bool kDouble = N >= kRingDoubleFromN;
    const StageRec *self;       // the lane's own record (its first copy)
    const double *vx64, *vy64;  // float64 velocities by lane (kept current only for the time-to-impact order)
    int i, base;                // agent index in its world, first lane of the world
    __device__ __forceinline__ OtherState other(int o) const {
        const StageRec *q = self + (o + 1);
        if (!kDouble) q = (o + 1 + i >= N) ? q - N : q;        // (one compare + select; the rest is the read's immediate offset)
        return OtherState{q->px, q->py, q->vxf, q->vyf, q->r};
    }
    __device__ __forceinline__ void vel64(int o, double &vx, double &vy) const {
        const int j = base + other_index(i, o, N);
        vx = vx64[j]; vy = vy64[j];
    }
};
// the field-major hand-over buffers of the pipeline and relay kernels (another wavefront staged them)
template <int N>
struct ArrayStage {
    const double *px, *py, *vx, *vy;
    const float *r;
    int i, base;
    __device__ __forceinline__ OtherState other(int o) const {
        const int j = base + other_index(i, o, N);
        return OtherState{px[j], py[j], (float)vx[j], (float)vy[j], r[j]};
    }
    __device__ __forceinline__ void vel64(int o, double &ovx, double &ovy) const {
        const int j = base + other_index(i, o, N);
        ovx = vx[j]; ovy = vy[j];
    }
};

// The four rotated features of one neighbour + its radius: everything here ends in a float32 observation (tolerance 1e-5) and
// nowhere else.  The positions take one fused multiply-add per feature in float64 (more accurate than the oracle's separate
// product and sum, an instruction less); the velocities (a few m/s at most: < 4e-7 of error) run in float32.  ONE statement of
// it for every launch form (they are held bit-identical to each other).
constexpr int kFeat = 5;   // p_par, p_orth, v_par, v_orth, r_other
__device__ __forceinline__ void neighbour_features(const Ego &e, float pxf, float pyf, double rx, double ry, const OtherState &q, float (&f)[kFeat]) {
    f[0] = (float)__builtin_fma(rx, e.prll_x, ry * e.prll_y);
    f[1] = (float)__builtin_fma(ry, e.prll_x, -(rx * e.prll_y));
    // (scalar float32 on purpose.  Left to itself the compiler packs the dot products of the THIRD neighbour into
    //      v_pk_mul_f32 t, (py, -py), (vx, vy) op_sel:[0,1] op_sel_hi:[1,0]     -- its SECOND source read with the halves swapped --
    //  and on gfx950 a v_pk_mul_f32 / v_pk_add_f32 whose LOW result takes the HIGH half of its second source (op_sel[1] = 1) reads that
    //  operand as 0 in lanes 48..63 now and then while another wavefront's MFMAs issue on the same SIMD: inside the fused actor kernel --
    //  two workgroups per CU, the other one in its policy phase -- v_par came out as vx*px alone, run to run.  Round 5 found the instruction
    //  by replacing it, and only it, with two v_mul_f32 in the compiler's assembly (tools/experiments/pk_isa_patch.py mul3_scalar), and
    //  reproduces it in 60 lines: tools/ubench/pk_mul_src1_swap.hip, profiles/r05_b_pk_src1_swap_hazard.txt; the first-source and
    //  third-source swaps, v_pk_mov_b32 and the swapped-halves v_pk_fma_f32 round 4 suspected are exact.  The barriers keep the chains out
    //  of the vectoriser's sight; tests/test_binary_guard.py fails the build if ANY packed float32 instruction with a low-half swap
    //  comes back.  -DCAVOID_DEV_PKFORM=0 restores the vectorised form for the bisect tools.)
#if !defined(CAVOID_DEV_PKFORM)
    float t_par = q.vyf * pyf, t_orth = q.vxf * pyf;
    asm volatile("" : "+v"(t_par), "+v"(t_orth));
    f[2] = __builtin_fmaf(q.vxf, pxf, t_par);
    f[3] = __builtin_fmaf(q.vyf, pxf, -t_orth);
#else
    f[2] = __builtin_fmaf(q.vxf, pxf, q.vyf * pyf);
    f[3] = __builtin_fmaf(q.vyf, pxf, -(q.vxf * pyf));
#endif
    f[4] = q.r;
}

// E6: centre distances to the other agents of the lane's world (from the staged positions), the
// collision test and the nearest gap.  All 64 lanes call this together.  What E9 needs later is kept
// in its cheapest form -- the gap as the float32 that goes into the observation and one 63-bit sort key
// per neighbour (its centimetre bucket and the float32 rounding of the lateral offset ry*tx - rx*ty, the
// tie-break inside a bucket) -- not the float64 values (the register budget decides how many wavefronts a
// SIMD holds, and with it how much memory latency hides).  float32 rounding is monotonic, so two DIFFERENT
// float32 laterals order exactly as the float64 ones do; equal keys fall back to the exact comparison.
// FEAT: the neighbour's observation features are made here too, while its staged state and the offset (rx, ry) are in
// registers (the row pass then only places values: no second read of the neighbour, no second subtraction) -- the forms that
// have the registers for it (everything but PARK).
// PARK (N >= kParkFromN): the keys and gaps do not stay in 3(N-1) registers from here to the observation rows; they are
// parked in the wave-private obs-tile region of LDS (idle until the rows are written), field-major: the keys [N-1][64] as
// 64-bit words, then the gaps [N-1][64].  The loop is then rolled three neighbours at a time (three square-root chains in flight instead of
// N-1), which is what lets the N = 10 kernels fit 128 registers.
constexpr int kParkFromN = 6;
// SW: some U-switch of c.switches is flipped (the caller tests the word once and picks the instantiation: inside the unrolled pair
// loop even a never-taken uniform branch per neighbour splits the N-1 square-root chains into separate scheduling regions and
// costs the one-step launch 9 %).
template <int N, bool PARK, bool SW, bool FEAT, class Stage>
__device__ __forceinline__ void pair_pass_impl(const KCfg &c, const Agent &a, const Ego &e, bool present, const Stage &st,
                                          Key (&key)[Others<N>::K], float (&gapf)[Others<N>::K], float (*feat)[kFeat],
                                          uint32_t &valid, bool &hit, double &min_gap, uint32_t *park, int lane, uint32_t frozen_w) {
    // frozen_w (U4 flipped, else 0): bit jj = agent jj of this lane's world was done before the step -- its pairs are skipped in
    // the collision test and the nearest gap (the observation still shows it)
    constexpr int K = Others<N>::K;
    const double ri = (double)a.radius;
    valid = 0u;
    hit = false;
    min_gap = INFINITY;
    if (!PARK) { key[0].set(kKeySentinel, 0u); gapf[0] = 0.0f; }
    auto one = [&](int o) {
        const OtherState q = st.other(o);
        const float rjf = q.r;
        const double rx = q.px - a.px, ry = q.py - a.py;
#if defined(CAVOID_DEV_ULP_FAULT) && CAVOID_DEV_ULP_FAULT == 1     /* injected fault (tests/test_gpu_tie_classifier.py): one float32 product in the distance */
        const double d = sqrt_dist2((double)((float)rx * (float)rx) + ry * ry);
#else
        const double d = sqrt_dist2(rx * rx + ry * ry);
#endif
        const bool other = present && (rjf >= 0.0f);
        bool collides = other;
        if (SW) {
            const int jj = other_index(st.i, o, N);
            collides = other && ((frozen_w >> jj) & 1u) == 0u && ((frozen_w >> st.i) & 1u) == 0u;    // (frozen_w = 0 unless U4 is flipped)
        }
        // unordered-pair gap d - (r_lo + r_hi): the sum is commutative, both ends agree bitwise
        const double gap_c = d - (ri + (double)rjf);
        min_gap = collides ? fmin(min_gap, gap_c) : min_gap;
        hit = hit || (collides && gap_c <= c.collision_dist);
        const bool seen = other && !(d > c.horizon);
        valid |= seen ? (1u << o) : 0u;
        // the observation's gap, host-side association (d - r_host) - r_other; rint(gap*100) is
        // order-isomorphic to round(gap, 2) and integer-valued: exact in int32 (|gap| < 1e7 m)
        const double gap_o = d - ri - (double)rjf;
#if defined(CAVOID_DEV_ULP_FAULT) && CAVOID_DEV_ULP_FAULT == 3     /* injected fault: the centimetre bucket of the sort key through float32 */
        uint32_t hi = kKeyBias - (uint32_t)(int)rintf((float)gap_o * 100.0f);
#else
        uint32_t hi = kKeyBias - (uint32_t)(int)rint(gap_o * 100.0);
#endif
        // U7a flipped (order by the exact gap): the float32 rounding of the gap, one bit dropped -- monotonic, so different values
        // order exactly as the float64 gaps; equal ones fall back to the exact comparison (assemble_obs, tie_first)
        uint32_t lo = orderable((float)(ry * e.tx - rx * e.ty));
        if (SW) {
            // U7b flipped: the agent index (distinct per neighbour) instead of the lateral offset -- a stable sort on the bucket;
            // with exact gaps: nothing -- equal float32 gaps must compare EQUAL so that the exact path decides
            if (c.switches & kSwIndexTie) lo = (uint32_t)other_index(st.i, o, N);
            if (c.switches & kSwExactGap) { lo = 0u; hi = 0x7FFFFFFEu - (orderable((float)gap_o) >> 1); }
        }
        hi = seen ? hi : kKeySentinel + (uint32_t)o;
        if (PARK) {
            reinterpret_cast<uint64_t *>(park)[o * 64 + lane] = ((uint64_t)hi << 32) | lo;
            park[(2 * K + o) * 64 + lane] = __float_as_uint((float)gap_o);
        } else {
            gapf[o] = (float)gap_o;
            key[o].set(hi, lo);
        }
        if (FEAT) neighbour_features(e, (float)e.prll_x, (float)e.prll_y, rx, ry, q, feat[o]);
    };
    if (PARK) {
        // (a scheduling fence per neighbour: left alone the scheduler interleaves all N-1 square-root chains of the unrolled loop
        //  and the one-step kernel needs 181 registers at N = 10 -- 2 wavefronts per SIMD instead of 3; fenced: 161)
#pragma unroll 3
        for (int o = 0; o < N - 1; ++o) { one(o); __builtin_amdgcn_sched_barrier(0); }
    } else {
#pragma unroll
        for (int o = 0; o < N - 1; ++o) if (!(CAVOID_SKIP & 16) || o == 0) one(o);
    }
}

template <int N, bool PARK, bool FEAT, class Stage>
__device__ __forceinline__ void pair_pass(const KCfg &c, const Agent &a, const Ego &e, bool present, const Stage &st,
                                          Key (&key)[Others<N>::K], float (&gapf)[Others<N>::K], float (*feat)[kFeat],
                                          uint32_t &valid, bool &hit, double &min_gap, uint32_t *park = nullptr, int lane = 0,
                                          uint32_t frozen_w = 0u) {
    if (CAVOID_RARE(c.switches != 0u))
        pair_pass_impl<N, PARK, true, FEAT>(c, a, e, present, st, key, gapf, feat, valid, hit, min_gap, park, lane, frozen_w);
    else
        pair_pass_impl<N, PARK, false, FEAT>(c, a, e, present, st, key, gapf, feat, valid, hit, min_gap, park, lane, 0u);
}

// Coalesced write-out of the wave's obs tile: n_floats contiguous floats starting at dst.
// stream: the rows are written once and not read again by this launch (per-step output slots of a K-step launch): non-temporal
// stores, so that K x 3.8 MB of output do not displace the L2's working set on their way to memory
#ifndef CAVOID_NT_STORES
#define CAVOID_NT_STORES 1
#endif
typedef float f32x4v __attribute__((ext_vector_type(4)));
// (a compile-time choice per flush: behind a run-time flag per store the optimiser merges the two stores of the diamond into one
//  plain store)
// CAVOID_STREAM_POLICY: what a streaming store of a tile that covers WHOLE 128-byte lines (WT: a compile-time property of the row width and rows per tile +
// the run-time alignment of its destination) is.  0: `nt` like every other streaming store -- the line stays DIRTY in the XCD's L2 until it is evicted
// or the launch's end-of-kernel release writes it back.  1 / 2 / 3: write-through (`sc1`, `sc0 sc1`, `sc0 sc1 nt`): the bytes leave for memory when the
// store is made and the line is dropped, nothing waits for the release.  Same box, bench.py's slot form: env_relay_kernel at 4 x 8192, 20-step launches: nt
// 29.9 us, sc1 28.2 (-5.7 %; 64-step launches unchanged; profiles/r06_x, r06_z); env_kernel<4, MODE_STEP_AUTORESET_N> at 4 x 65536: 485 -> 467 us per 64-step
// launch (-4 %), 4 x 1 M worlds +-0.  Tiles with RAGGED lines must not take it: 10 agents per world (60 rows x 69 floats = 129.4 lines per tile) 411 -> 614 us
// per 64-step launch, 10 x 262 144 2748 -> 4293 (profiles/r06_aa: every partial line becomes its own write at the memory side) -- hence WT.
#ifndef CAVOID_STREAM_POLICY
#define CAVOID_STREAM_POLICY 1
#endif
template <bool STREAM, bool WT = false>
__device__ __forceinline__ void store16(float4 *p, const float4 &v) {
    if (CAVOID_NT_STORES && STREAM) {
        const f32x4v x{v.x, v.y, v.z, v.w};
        if (WT && CAVOID_STREAM_POLICY != 0) {
            // (inline assembly: the compiler has no cache-policy argument for a plain vector store.  It does not see the instruction, so its hazard
            //  recogniser cannot keep the two wait states gfx940+ wants between a store of more than 64 bits and a vector write of its DATA
            //  registers -- and it does re-use them at once, as the next store's address: the s_nop is that distance, made by hand)
#if CAVOID_STREAM_POLICY == 2
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
#elif CAVOID_STREAM_POLICY == 3
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
#else
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
#endif
        } else {
            __builtin_nontemporal_store(x, reinterpret_cast<f32x4v *>(p));
        }
    } else {
        *p = v;
    }
}
template <bool STREAM = false>
__device__ __forceinline__ void flush_tile(const float *tile, float *dst, int n_floats, int lane) {
    if ((n_floats & 3) == 0 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
        const float4 *src4 = reinterpret_cast<const float4 *>(tile);
        float4 *dst4 = reinterpret_cast<float4 *>(dst);
        const int n4 = n_floats >> 2;
        for (int k0 = lane; k0 < n4; k0 += 64 * 8) {       // 8 LDS reads in flight, then 8 stores
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 64 * u;
                v[u] = k < n4 ? src4[k] : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 64 * u;
                if (k < n4) store16<STREAM>(dst4 + k, v[u]);
            }
        }
    } else {
        for (int k = lane; k < n_floats; k += 64) dst[k] = tile[k];
    }
}

// slot numbers of the (up to 15) others of a lane, 4 bits each, in two registers instead of N-1
struct Slots {
    uint32_t lo, hi;                                             // others 0..7, 8..15
    __device__ __forceinline__ void clear() { lo = 0u; hi = 0u; }
    __device__ __forceinline__ void bump(int o, uint32_t bit) {  // += bit at field o (one v_lshl_add_u32)
        if (o < 8) lo += bit << (4 * o); else hi += bit << (4 * (o - 8));
    }
    __device__ __forceinline__ void set(int o, int slot) { bump(o, (uint32_t)slot); }
    __device__ __forceinline__ int get(int o) const { return (int)(((o < 8 ? lo : hi) >> (4 * (o & 7))) & 15u); }
};
static_assert(CAVOID_MAX_AGENTS - 1 <= 15, "a slot number must fit 4 bits");

// Ranking by a round-robin tournament in integer arithmetic.  For every unordered pair (p, q), p < q, the sign of the 64-bit
// difference of their keys says "p comes first"; the position of a neighbour in the order = the number of pairs it lost:
// pos[q] += first, pos[p] -= first on top of the start value NO-1-p (p loses to every later neighbour unless the difference
// says otherwise).  Full-rate 32-bit instructions only (v_sub_co / v_subb_co / shift / add / sub), no wave mask parked in scalar
// registers, no branch: 5.2 vector instructions per pair measured at N = 10 (PMC), against 7.5 for the forms that packed the
// counters four bits each or kept one outcome bit per pair.  Measured and dropped: the borrow consumed as a carry
// (__builtin_usubll_overflow -> v_addc_co / v_subb_co: +0.9 instructions per pair after the hazard no-ops), and 64-bit compares
// (v_cmp_lt_u64 / v_cmp_eq_u64 issue at a quarter of the 32-bit rate on gfx950).  `tie` comes back true when two keys were EQUAL
// (difference zero): the caller then ranks the exact way.  With equal keys the later neighbour comes first.
template <int NO, bool ASC_BUCKET, int KK>
__device__ __forceinline__ bool tournament(const Key (&key)[KK], int (&pos)[KK]) {
    uint64_t k[KK];
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        // near -> far order (closest_first) flips the bucket half of the key; sentinels stay on top
        if (ASC_BUCKET) k[o] = ((uint64_t)(key_is_sentinel(key[o].hi()) ? key[o].hi() : 2u * kKeyBias - key[o].hi()) << 32) | key[o].lo();
        else k[o] = key[o].v;
        pos[o] = NO - 1 - o;
    }
    uint32_t differ = 0xFFFFFFFFu;
#pragma unroll
    for (int p = 0; p < NO; ++p)
#pragma unroll
        for (int q = p + 1; q < NO; ++q) {
            const uint64_t d = k[p] - k[q];                          // both < 2^63: bit 63 of d <=> kp < kq <=> p first
            const int p_first = (int)(d >> 63);
            pos[q] += p_first;
            pos[p] -= p_first;
            const uint32_t nz = (uint32_t)d | (uint32_t)(d >> 32);
            differ = nz < differ ? nz : differ;
        }
    return differ == 0u;
}

// the same for a compile-time float count (the common shape: a full wavefront of rows of the default width): every
// round but the last is unpredicated
// WT: the tile covers whole 128-byte lines at its destination (store16)
template <int NF, bool STREAM = false, bool WT = false>
__device__ __forceinline__ void flush_tile_fixed(const float *tile, float *dst, int lane) {
    static_assert(NF % 4 == 0, "whole float4s");
    constexpr int n4 = NF / 4, rounds = (n4 + 63) / 64;
    const float4 *src4 = reinterpret_cast<const float4 *>(tile);
    float4 *dst4 = reinterpret_cast<float4 *>(dst);
#pragma unroll
    for (int r0 = 0; r0 < rounds; r0 += 8) {                   // 8 LDS reads in flight, then 8 stores
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + u;
            if (r < rounds) {
                const int k = lane + 64 * r;
                v[u] = ((r + 1) * 64 <= n4 || k < n4) ? src4[k] : float4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // (left alone the scheduler pairs every read with its store -- read, wait for the LDS, store, 17 times over at N = 10:
        //  a serial LDS round trip per 16 bytes on the step's chain; fenced, the eight reads are in flight together)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + u;
            if (r < rounds) {
                const int k = lane + 64 * r;
                if ((r + 1) * 64 <= n4 || k < n4) store16<STREAM, WT>(dst4 + k, v[u]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// E9: neighbour ordering by counting ranks, the lane's observation row into the LDS tile, and the
// coalesced write-out.  The tile holds c.tile_rows rows; wide rows (large N) go out in several passes so
// that the LDS footprint -- and with it the wavefronts resident per CU -- does not scale with N*(1+D).
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// Which part of its row a lane writes (the cooperative step, cavoid_quad.hpp: the ranking is made by every wavefront that writes, from the
// same keys; the host wavefront writes the row's head and clears the empty slots itself, a pair wavefront the slots of ITS neighbours)
struct PartAll { static constexpr bool head = true; __device__ __forceinline__ constexpr bool other(int) const { return true; } };
struct PartOthers {                                          // neighbours o with o % every == mine
    static constexpr bool head = false;
    int mine, every;
    __device__ __forceinline__ bool other(int o) const { return o % every == mine; }
};
// PreFlush: called once, right before the first global store of the tile (env_relay_kernel orders the last step's rows behind
// the rows of the steps before it, which other wavefronts write)
// FLUSH = false: the rows stay in the LDS tile (row r at tile + r * ostride; needs c.tile_rows >= rows_active), nothing is written
// to obs_dst
// FEAT: the pair pass made the neighbours' features (feat_in); else they are made here from the staged state.
template <int N, bool PARK, bool FEAT, class Stage, class PreFlush = NoHook, bool FLUSH = true, class Part = PartAll>
__device__ __forceinline__ void assemble_obs(const KCfg &c, const Agent &a, const Ego &e, bool active, int lane, const Stage &st,
                                             const Key (&key_in)[Others<N>::K], const float (&gapf_in)[Others<N>::K], const float (*feat_in)[kFeat],
                                             uint32_t valid, float *tile,
                                             float *obs_dst, int rows_active, int ostride, bool packed, float rew_f, float done_f, int64_t wave,
                                             PreFlush pre_flush = PreFlush(), bool stream_out = false, int *prev_kept = nullptr, Part part = Part()) {
    // prev_kept (step loops that own their tile from step to step): how many slots of this lane's row the PREVIOUS step of the
    // launch filled -- the slots behind them are still zero in the tile, so only the slots [kept, *prev_kept) need zeroing now
    // (none, step after step, unless the lane's world restarted or a neighbour went out of sight).  Null: every slot behind `kept`.
    constexpr int K = Others<N>::K, NO = N - 1;
    const int i = st.i;
    // PARK: the pair pass left keys and gaps in the tile region (pair_pass); they come back into registers only now
    Key key[K];
    float gapf[K];
#pragma unroll
    for (int o = 0; o < K; ++o) {
        if (PARK) {
            if (o < NO) key[o].v = reinterpret_cast<const uint64_t *>(tile)[o * 64 + lane];
            else key[o].set(kKeySentinel + (uint32_t)o, 0u);
            gapf[o] = 0.0f;                                      // (fetched after the ranking, just before the rows are written)
        } else { key[o] = key_in[o]; gapf[o] = gapf_in[o]; }
    }
    const int M = c.max_other, width = c.width;
    const bool present = active && (a.flags & CAVOID_F_PRESENT);
    const double ri = (double)a.radius;
    // sort criteria: gap rounded to centimetres, then the lateral offset (its sign-preserving un-normalised form
    // ry*tx - rx*ty orders the same), then the agent index (what a stable sort does).  Fast path: the integer
    // tournament over the 63-bit keys of the pair pass.  Two neighbours with EQUAL keys (same bucket, same float32
    // lateral) need the exact float64 laterals and the index rule: the generic path below (wave-uniform branch).
    // The exact ranking (rare: two equal fast keys in the tile, the time_to_impact order): ROLLED loops over the others with every
    // criterion re-derived from the staged state -- O(N^2) square roots at run time, but a constant amount of code (unrolled
    // per pair, this path was a third of the kernel's code).  `among`: the candidates; near_first: the closest_first re-rank of the
    // kept set.  Criteria, in order: [time to impact, larger first]; the gap (its centimetre bucket, or -- U7a flipped -- the exact
    // float64 gap), larger first (smaller first when near_first); the lateral offset, smaller first (U7b flipped: skipped); the
    // agent index (what the oracle's stable sort leaves).  Neighbours outside `among` come behind the candidates, in ring order
    // (their slots, where the row has them, take zeros).
    auto rank_exact = [&](uint32_t among, bool near_first, bool use_tti, int (&out)[K]) {
        auto criteria = [&](int o, double &g, double &l, double &t, int &jj) {
            jj = other_index(i, o, N);
            const OtherState q = st.other(o);
            const double rj = (double)q.r;
            const double rx = q.px - a.px, ry = q.py - a.py;
            const double gap = sqrt_dist2(rx * rx + ry * ry) - ri - rj;
            g = (c.switches & kSwExactGap) ? gap : rint(gap * 100.0);
            l = (c.switches & kSwIndexTie) ? 0.0 : ry * e.tx - rx * e.ty;
            t = 0.0;
            if (use_tti) {
                double ovx, ovy;
                st.vel64(o, ovx, ovy);
                t = time_to_impact(rx, ry, a.vx - ovx, a.vy - ovy, ri + rj);
            }
        };
        Slots packed_out;
        packed_out.clear();
        const int n_among = __popc(among);
#pragma unroll 1
        for (int p = 0; p < NO; ++p) {
            double gp, lp, tp;
            int jp;
            criteria(p, gp, lp, tp, jp);
            int before = 0;
#pragma unroll 1
            for (int q = 0; q < NO; ++q) {
                double gq, lq, tq;
                int jq;
                criteria(q, gq, lq, tq, jq);
                const bool tie_break = (lq < lp) || (lq == lp && jq < jp);
                const bool by_gap = near_first ? (gq < gp) || (gq == gp && tie_break) : (gq > gp) || (gq == gp && tie_break);
                const bool q_first = use_tti ? (tq > tp) || (tq == tp && by_gap) : by_gap;
                before += (q != p && ((among >> q) & 1u) && q_first) ? 1 : 0;
            }
            const bool member = (among >> p) & 1u;
            packed_out.set(p, member ? before : n_among + __popc(~among & ((1u << p) - 1u)));
        }
#pragma unroll
        for (int o = 0; o < NO; ++o) out[o] = packed_out.get(o);
    };
    const int m = __popc(valid);
    const int first = m > M ? m - M : 0;
    const int kept = m - first;
    int pos[K];
#pragma unroll
    for (int o = 0; o < K; ++o) pos[o] = 0;
    uint32_t keep = 0u;
    bool generic = c.sort_method == CAVOID_SORT_TIME_TO_IMPACT;
    if (!generic) {
        // far -> near: larger bucket first, then smaller lateral
        const bool tie = (CAVOID_SKIP & 2) ? false : tournament<NO, false>(key, pos);
        generic = __ballot(tie) != 0ull;                        // wave-uniform: redo this tile's ranks the exact way
    }
    if (CAVOID_RARE(generic)) rank_exact(valid, false, c.sort_method == CAVOID_SORT_TIME_TO_IMPACT, pos);
#pragma unroll
    for (int o = 0; o < NO; ++o) keep |= (((valid >> o) & 1u) && pos[o] >= first) ? (1u << o) : 0u;
    // pos - first IS the slot (closest_last / time_to_impact); subtracted at the use
    int slot_bias = first;
    if (CAVOID_RARE(c.sort_method == CAVOID_SORT_CLOSEST_FIRST)) {      // kept set re-ranked near -> far; full ties keep index order
        slot_bias = 0;
        Key k2[K];
#pragma unroll
        for (int o = 0; o < K; ++o)                            // neighbours that were clipped away become sentinels
            k2[o].set(((keep >> o) & 1u) ? key[o].hi() : kKeySentinel + (uint32_t)o, key[o].lo());
        const bool tie = tournament<NO, true>(k2, pos);
        if (CAVOID_RARE(__ballot(tie) != 0ull)) rank_exact(keep, true, false, pos);
    }

    CAVOID_STAMP(9);                                             // ranks done
    if (PARK) {                                                  // the gaps come back into registers; then the region is free for the rows
        const uint32_t *park = reinterpret_cast<const uint32_t *>(tile);
#pragma unroll
        for (int o = 0; o < NO; ++o) gapf[o] = __uint_as_float(park[(2 * K + o) * 64 + lane]);
        wave_lds_sync();
    }
    const float pxf = (float)e.prll_x, pyf = (float)e.prll_y;
    const int rpp = c.tile_rows;                                 // rows per pass
    // Unfilled slots.  In the key-parking instantiations (one step per launch, N >= 6): when some row of the wavefront has any (a
    // wave-uniform test: worlds with fewer agents than N, neighbours out of sight), the whole tile is zeroed first, cooperatively,
    // with 16-byte LDS writes (N = 10: 17 per lane, no vector ALU work) and the rows then write only what they have -- 10 x 262144
    // with 2..10 agents present: 213 -> 194 us per step, 10 x 8192: 13.0 -> 12.8.  The step-loop instantiations keep the per-lane
    // loop over the slots behind `kept` (there the cooperative form measured 5 % slower: 6.63 -> 6.98 us per step at 10 x 8192).
    const bool zero_first = PARK && __ballot(active && kept < M) != 0ull;
    for (int p0 = 0; p0 < rows_active; p0 += rpp) {
    if (zero_first) {
        const int rows_z = rows_active - p0 < rpp ? rows_active - p0 : rpp;
        const int n4 = (rows_z * ostride + 3) >> 2;
        float4 *t4 = reinterpret_cast<float4 *>(tile);
        for (int k = lane; k < n4; k += 64) t4[k] = float4{0.f, 0.f, 0.f, 0.f};
        wave_lds_sync();
    }
    if (!(CAVOID_SKIP & 4) && active && lane >= p0 && lane < p0 + rpp) {
        float *row = tile + (lane - p0) * ostride;
        if (Part::head) {
        row[0] = (present && (a.flags & CAVOID_F_LEARNING)) ? 1.0f : 0.0f;
        row[1] = (float)kept;                                   // 0 for an absent agent (others == 0)
        row[2] = present ? (float)e.dist : 0.0f;
        row[3] = present ? (float)e.heading_ego : 0.0f;
        row[4] = present ? a.pref : 0.0f;
        row[5] = present ? a.radius : 0.0f;
        }
        // r_host + r_other as ONE float32 add is bit for bit the float32 rounding of the float64 sum (the sum of two floats is
        // exact in float64).
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            if (!((keep >> o) & 1u)) continue;
            if (!part.other(o)) continue;
            float f[kFeat];
            if (FEAT) {
#pragma unroll
                for (int q = 0; q < kFeat; ++q) f[q] = feat_in[o][q];
            } else {
                const OtherState q = st.other(o);
                neighbour_features(e, pxf, pyf, q.px - a.px, q.py - a.py, q, f);
            }
            float *dst = row + 6 + 7 * (pos[o] - slot_bias);
            dst[0] = f[0]; dst[1] = f[1]; dst[2] = f[2]; dst[3] = f[3]; dst[4] = f[4]; dst[5] = a.radius + f[4]; dst[6] = gapf[o];
        }
        // unfilled slots, a slot (seven floats) per iteration: the loop runs as long as the emptiest row of the wavefront -- an absent
        // agent's, all M slots -- so float by float it was 7 M dependent iterations per step at N = 10 with 2..10 agents present.
        // (Measured and dropped there: straight-line predicated zero writes into the slots the not-kept neighbours rank at, +80
        // vector instructions per wavefront-step.)
        if (Part::head) {
        const int zero_to = prev_kept ? *prev_kept : M;
        for (int sl = zero_first ? M : kept; sl < zero_to; ++sl) {
            float *z = row + 6 + 7 * sl;
#pragma unroll
            for (int q = 0; q < 7; ++q) z[q] = 0.0f;
        }
        if (packed) { row[width] = rew_f; row[width + 1] = done_f; }   // (obs | reward | done) gather record
        if (prev_kept) *prev_kept = kept;
        }
    }
    wave_lds_sync();
    CAVOID_STAMP(10);                                            // rows in the LDS tile
    if (p0 == 0) pre_flush();
    const int rows_here = rows_active - p0 < rpp ? rows_active - p0 : rpp;
    if (FLUSH && !(CAVOID_SKIP & 8)) {
        constexpr int kRows = (64 / N) * N, kW = 6 + 7 * (N - 1);   // a full wavefront, the default row widths
        constexpr bool kPlainOk = (kRows * kW) % 4 == 0, kPackedOk = (kRows * (kW + 2)) % 4 == 0;
        float *dst = obs_dst + (int64_t)p0 * ostride;
        const bool whole = rows_here == kRows && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
        if (stream_out) {                                        // (wave-uniform: a property of the launch)
            // whole 128-byte lines (32 floats) per tile and a destination on a line boundary: the write-through form of the streaming store (store16)
            constexpr bool kPlainLines = kPlainOk && (kRows * kW) % 32 == 0, kPackedLines = kPackedOk && (kRows * (kW + 2)) % 32 == 0;
            const bool lines = (reinterpret_cast<uintptr_t>(dst) & 127) == 0;
            if (kPlainOk && whole && ostride == kW) {
                if (kPlainLines && lines) flush_tile_fixed<kPlainOk ? kRows * kW : 4, true, kPlainLines>(tile, dst, lane);
                else flush_tile_fixed<kPlainOk ? kRows * kW : 4, true>(tile, dst, lane);
            } else if (kPackedOk && whole && ostride == kW + 2) {
                if (kPackedLines && lines) flush_tile_fixed<kPackedOk ? kRows * (kW + 2) : 4, true, kPackedLines>(tile, dst, lane);
                else flush_tile_fixed<kPackedOk ? kRows * (kW + 2) : 4, true>(tile, dst, lane);
            } else {
                flush_tile<true>(tile, dst, rows_here * ostride, lane);
            }
        } else {
            if (kPlainOk && whole && ostride == kW) flush_tile_fixed<kPlainOk ? kRows * kW : 4>(tile, dst, lane);
            else if (kPackedOk && whole && ostride == kW + 2) flush_tile_fixed<kPackedOk ? kRows * (kW + 2) : 4>(tile, dst, lane);
            else flush_tile(tile, dst, rows_here * ostride, lane);
        }
    }
    if (p0 + rpp < rows_active) wave_lds_sync();                 // the next pass overwrites the tile
    }
}

__device__ __forceinline__ void load_agent(const KState &s, int64_t k, Agent &a) {
    a.px = s.px[k]; a.py = s.py[k]; a.heading = s.heading[k]; a.t_rem = s.t_rem[k];
    a.gx = s.gx[k]; a.gy = s.gy[k]; a.radius = s.radius[k]; a.pref = s.pref[k];
    a.flags = s.flags[k];
}

__device__ __forceinline__ void store_agent(const KState &s, int64_t k, const Agent &a) {
    s.px[k] = a.px; s.py[k] = a.py; s.heading[k] = a.heading; s.t_rem[k] = a.t_rem;
    s.gx[k] = a.gx; s.gy[k] = a.gy; s.radius[k] = a.radius; s.pref[k] = a.pref;
    s.speed[k] = a.speed; s.flags[k] = a.flags;
}

// Pool entry used by episode `ep` of global world `gw`: a splitmix64-style finaliser of (seed, gw, ep)
// reduced to [0, P) by multiply-shift (no division).  Cheap on purpose: it sits on the critical path
// of every wavefront in which a world restarts.  (GEN v1 itself keeps Philox4x32-10.)
__device__ __forceinline__ uint32_t pool_index(const KCfg &c, uint32_t gw, uint32_t ep) {
    if (c.ahead > 0)                                            // look-ahead ring: this world's own slot of episode ep (R a power of two)
        return (gw - (uint32_t)c.world_offset) * (uint32_t)c.ahead + (ep & (uint32_t)(c.ahead - 1));
    uint64_t z = ((uint64_t)c.seed_hi << 32 | c.seed_lo) + 0x9E3779B97F4A7C15ull * ((uint64_t)gw + 1ull) +
                 0xC2B2AE3D27D4EB4Full * ((uint64_t)ep + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(((z >> 32) * (uint64_t)(uint32_t)c.pool_size) >> 32);
}

// Scenario of (global world gw, episode ep): a gather from the pre-generated pool (pool entry k is
// GEN v1's world k, episode 0) or, with no pool, the generator itself.
__device__ __forceinline__ void load_pool(const PoolRec *pool, int64_t k, Agent &a) {
    const PoolRec r = pool[k];                                  // four 16-byte loads, one cache line
    a.px = r.px; a.py = r.py; a.heading = r.heading; a.t_rem = r.t_rem;
    a.gx = r.gx; a.gy = r.gy; a.radius = r.radius; a.pref = r.pref;
    a.flags = r.flags;
    a.vx = a.vy = 0.0;
    a.speed = 0.0f;
}

template <int N>
__device__ __forceinline__ void new_episode(const KCfg &c, const PoolRec *pool, uint32_t gw, uint32_t ep, int i, Agent &a) {
    if (c.pool_size > 0) {
        load_pool(pool, (int64_t)pool_index(c, gw, ep) * N + i, a);
    } else {
        generate_agent<N>(c, gw, ep, i, a);
    }
}

// ---- scenario look-ahead (cavoid_cfg::gen_lookahead = R): refill of every world's ring with the scenarios of its NEXT episodes ----------
// One lane per (world, agent), the env step's lane mapping.  filled_hi[w] = the highest episode whose scenario is in world w's ring
// (slots of episodes (filled_hi - R, filled_hi] are valid; 0xFFFFFFFF = nothing yet); the launch generates episodes
// max(filled_hi, episode) + 1 .. episode + need -- in the steady state the one or two a world consumed since the last refill -- with the
// generator of (seed, GLOBAL world id, episode): GEN v1 per lane, GEN v2 wave-cooperatively, exactly what the in-kernel restart of
// gen_pool_size = 0 computes (tests/test_gpu_lookahead.py holds the two bitwise equal).
// one wavefront's share of the refill: `wave` = index of the 64-lane tile of (world, agent) slots, scratch = 4 x 64 doubles + 64 floats of
// wave-private LDS (GEN v2 only).  The missing episodes of a world are dealt over `ny` wavefronts (this one takes the y-th, y + ny-th, ...), so
// that a world which consumed several episodes since the last refill does not serialise their generation: hi_in is read by all of them, the
// y = 0 wavefront writes the new bookkeeping to hi_out (the host swaps the two arrays after every refill).
template <int N>
__device__ __forceinline__ void ahead_fill_wave(const KCfg &c, const uint32_t *episode, const uint32_t *hi_in, uint32_t *hi_out, PoolRec *ahead, const int need,
                                                const int64_t wave, const int lane, const int y, const int ny, double *sc_d, float *sc_r) {
    const int wpw = c.wpw, lanes_used = wpw * N;
    const int lw = lane / N, i = lane - lw * N;
    const int64_t w = wave * wpw + lw;
    const bool active = lane < lanes_used && w < c.num_worlds;
    const int base = lane < lanes_used ? lw * N : 0;
    uint32_t ep = 0u, fh = 0u;
    if (active) {
        ep = episode[w];
        fh = hi_in[w];
        if (fh == 0xFFFFFFFFu || (int32_t)(fh - ep) < 0) fh = ep;      // nothing valid ahead of this world's current episode
    }
    const uint32_t target = ep + (uint32_t)need;
    int missing = active ? (int)(int32_t)(target - fh) : 0;
    missing = missing < 0 ? 0 : missing;
    int trips = missing;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(trips, o, 64); trips = v > trips ? v : trips; }
    const uint32_t gw = (uint32_t)(c.world_offset + w);
    for (int k = y; k < trips; k += ny) {                       // (wave-uniform)
        const uint32_t e = fh + 1u + (uint32_t)k;
        const bool fresh = active && k < missing;
        Agent a;
        absent_agent(a);
        if (c.gen_mode == 1) generate_world_v2<N>(c, gw, e, i, base, lane, fresh, sc_d, sc_d + 64, sc_d + 128, sc_d + 192, sc_r, a);
        else if (fresh) generate_agent<N>(c, gw, e, i, a);
        if (fresh) {
            PoolRec r;
            r.px = a.px; r.py = a.py; r.heading = a.heading; r.t_rem = a.t_rem;
            r.gx = a.gx; r.gy = a.gy; r.radius = a.radius; r.pref = a.pref;
            r.flags = a.flags; r.pad[0] = r.pad[1] = r.pad[2] = 0u;
            ahead[((int64_t)w * c.ahead + (int64_t)(e & (uint32_t)(c.ahead - 1))) * N + i] = r;
        }
    }
    if (active && i == 0 && y == 0) hi_out[w] = missing > 0 ? target : fh;
}

template <int N>
__global__ void __launch_bounds__(256) ahead_fill_kernel(const KCfg c, const uint32_t *episode, const uint32_t *hi_in, uint32_t *hi_out, PoolRec *ahead, const int need) {
    __shared__ double sh_d[4][4 * 64];
    __shared__ float sh_r[4][64];
    const int wave_in_block = threadIdx.x >> 6;
    ahead_fill_wave<N>(c, episode, hi_in, hi_out, ahead, need, (int64_t)blockIdx.x * 4 + wave_in_block, threadIdx.x & 63, (int)blockIdx.y, (int)gridDim.y,
                       sh_d[wave_in_block], sh_r[wave_in_block]);
}

// ---- RVO scripted policy: ORCA, van den Berg et al., "Reciprocal n-body collision avoidance"
// (ISRR 2009) -- one half-plane per neighbour, the 2-D linear programme of its section 5.2, and the least-penetration
// programme when the half-planes admit no velocity.  float64, neighbours in agent-index order: the operation sequence of
// oracle/cavoid_oracle.py (orca_lines, _lp_on_line, _lp_plane, _lp_least_penetration, rvo_action).  The lines live in
// wave-private LDS (a lane indexes them dynamically): line k of set s of a lane at rvo[((s*(N-1) + k)*64 + lane)*4 ..+3].
constexpr double kRvoEps = 1e-5;
__device__ __forceinline__ double det2(double ax, double ay, double bx, double by) { return ax * by - ay * bx; }

struct Line { double px, py, dx, dy; };
template <int N>
struct RvoLines {
    double *mem; int lane;
    __device__ __forceinline__ Line get(int set, int k) const {
        const double *p = mem + ((size_t)(set * (N - 1) + k) * 64 + lane) * 4;
        return Line{p[0], p[1], p[2], p[3]};
    }
    __device__ __forceinline__ void put(int set, int k, const Line &l) const {
        double *p = mem + ((size_t)(set * (N - 1) + k) * 64 + lane) * 4;
        p[0] = l.px; p[1] = l.py; p[2] = l.dx; p[3] = l.dy;
    }
};

template <int N>
__device__ __forceinline__ bool lp_on_line(const RvoLines<N> &L, int set, int k, double radius, double ox, double oy, bool direction_opt,
                                           double &x, double &y) {
    const Line l = L.get(set, k);
    const double dot = l.px * l.dx + l.py * l.dy, disc = dot * dot + radius * radius - (l.px * l.px + l.py * l.py);
    if (disc < 0.0) return false;
    const double root = sqrt(disc);
    double t_lo = -dot - root, t_hi = -dot + root;
    for (int i = 0; i < k; ++i) {
        const Line o = L.get(set, i);
        const double den = det2(l.dx, l.dy, o.dx, o.dy), num = det2(o.dx, o.dy, l.px - o.px, l.py - o.py);
        if (fabs(den) <= kRvoEps) { if (num < 0.0) return false; continue; }
        const double t = num / den;
        if (den >= 0.0) t_hi = fmin(t_hi, t); else t_lo = fmax(t_lo, t);
        if (t_lo > t_hi) return false;
    }
    double t;
    if (direction_opt) t = (ox * l.dx + oy * l.dy > 0.0) ? t_hi : t_lo;
    else { t = l.dx * (ox - l.px) + l.dy * (oy - l.py); t = t < t_lo ? t_lo : (t > t_hi ? t_hi : t); }
    x = l.px + t * l.dx; y = l.py + t * l.dy;
    return true;
}

template <int N>
__device__ __forceinline__ int lp_plane(const RvoLines<N> &L, int set, int m, double radius, double ox, double oy, bool direction_opt,
                                        double &x, double &y) {
    if (direction_opt) { x = ox * radius; y = oy * radius; }
    else if (ox * ox + oy * oy > radius * radius) { const double nn = sqrt(ox * ox + oy * oy); x = ox / nn * radius; y = oy / nn * radius; }
    else { x = ox; y = oy; }
    for (int k = 0; k < m; ++k) {
        const Line l = L.get(set, k);
        if (det2(l.dx, l.dy, l.px - x, l.py - y) > 0.0) {
            double nx, ny;
            if (!lp_on_line<N>(L, set, k, radius, ox, oy, direction_opt, nx, ny)) return k;
            x = nx; y = ny;
        }
    }
    return m;
}

template <int N>
__device__ __forceinline__ void lp_least_penetration(const RvoLines<N> &L, int m, int begin, double radius, double &x, double &y) {
    double distance = 0.0;
    for (int k = begin; k < m; ++k) {
        const Line l = L.get(0, k);
        if (det2(l.dx, l.dy, l.px - x, l.py - y) > distance) {
            int np = 0;
            for (int j = 0; j < k; ++j) {
                const Line o = L.get(0, j);
                const double den = det2(l.dx, l.dy, o.dx, o.dy);
                double nx, ny;
                if (fabs(den) <= kRvoEps) {
                    if (l.dx * o.dx + l.dy * o.dy > 0.0) continue;
                    nx = 0.5 * (l.px + o.px); ny = 0.5 * (l.py + o.py);
                } else {
                    const double t = det2(o.dx, o.dy, l.px - o.px, l.py - o.py) / den;
                    nx = l.px + t * l.dx; ny = l.py + t * l.dy;
                }
                const double fx = o.dx - l.dx, fy = o.dy - l.dy, fn = sqrt(fx * fx + fy * fy);
                L.put(1, np, Line{nx, ny, fx / fn, fy / fn});
                ++np;
            }
            double nx, ny;
            if (lp_plane<N>(L, 1, np, radius, -l.dy, l.dx, true, nx, ny) >= np) { x = nx; y = ny; }
            distance = det2(l.dx, l.dy, l.px - x, l.py - y);
        }
    }
}

// [speed, delta_heading] of the RVO agent in this lane.  lds_* hold the PRE-move state of the wavefront's agents (positions,
// last velocities, radii; radius < 0 marks an absent row).  Called by the lanes that hold a running RVO agent only.
template <int N>
__device__ __forceinline__ void rvo_action(const KCfg &c, const Agent &a, int i, int base, int lane, const double *lds_px,
                                           const double *lds_py, const double *lds_vx, const double *lds_vy, const float *lds_r,
                                           double *rvo_mem, double &a0, double &a1) {
    const RvoLines<N> L{rvo_mem, lane};
    const double hvx = lds_vx[lane], hvy = lds_vy[lane];
    int m = 0;
    for (int jj = 0; jj < N; ++jj) {
        const int j = base + jj;
        const float rjf = lds_r[j];
        if (jj == i || rjf < 0.0f) continue;
        const double rpx = lds_px[j] - a.px, rpy = lds_py[j] - a.py, rvx = hvx - lds_vx[j], rvy = hvy - lds_vy[j];
#if defined(CAVOID_DEV_ULP_FAULT) && CAVOID_DEV_ULP_FAULT == 4     /* injected fault: one float32 product in the ORCA policy's squared distance */
        const double dist_sq = (double)((float)rpx * (float)rpx) + rpy * rpy;
#else
        const double dist_sq = rpx * rpx + rpy * rpy;
#endif
        const double comb = c.cold->rvo_radius_scale * (double)a.radius + c.cold->rvo_radius_scale * (double)rjf, comb_sq = comb * comb;
        double dx, dy, ucx, ucy;
        if (dist_sq > comb_sq) {
            const double wx = rvx - c.cold->rvo_inv_horizon * rpx, wy = rvy - c.cold->rvo_inv_horizon * rpy;
            const double w_sq = wx * wx + wy * wy, dot1 = wx * rpx + wy * rpy;
            if (dot1 < 0.0 && dot1 * dot1 > comb_sq * w_sq) {
                const double w_len = sqrt(w_sq), ux = wx / w_len, uy = wy / w_len, scale = comb * c.cold->rvo_inv_horizon - w_len;
                dx = uy; dy = -ux; ucx = scale * ux; ucy = scale * uy;
            } else {
                const double leg = sqrt(dist_sq - comb_sq);
                if (det2(rpx, rpy, wx, wy) > 0.0) { dx = (rpx * leg - rpy * comb) / dist_sq; dy = (rpx * comb + rpy * leg) / dist_sq; }
                else { dx = -(rpx * leg + rpy * comb) / dist_sq; dy = -(-rpx * comb + rpy * leg) / dist_sq; }
                const double dot2 = rvx * dx + rvy * dy;
                ucx = dot2 * dx - rvx; ucy = dot2 * dy - rvy;
            }
        } else {
            const double inv_dt = 1.0 / c.dt, wx = rvx - inv_dt * rpx, wy = rvy - inv_dt * rpy;
            const double w_len = sqrt(wx * wx + wy * wy), ux = wx / w_len, uy = wy / w_len, scale = comb * inv_dt - w_len;
            dx = uy; dy = -ux; ucx = scale * ux; ucy = scale * uy;
        }
        L.put(0, m, Line{hvx + c.cold->rvo_collab * ucx, hvy + c.cold->rvo_collab * ucy, dx, dy});
        ++m;
    }
    const double gx = (double)a.gx - a.px, gy = (double)a.gy - a.py, gn = sqrt(gx * gx + gy * gy);
    const double scale = gn > 0.0 ? (double)a.pref / gn : 0.0;
    double vx, vy;
    const int fail = lp_plane<N>(L, 0, m, (double)a.pref, scale * gx, scale * gy, false, vx, vy);
    if (fail < m) lp_least_penetration<N>(L, m, fail, (double)a.pref, vx, vy);
    double speed = sqrt(vx * vx + vy * vy);
    double delta = 0.0;
    if (speed > 0.0) {
        delta = atan2(vy, vx) - a.heading;
        while (delta >= kPi) delta -= 2.0 * kPi;
        while (delta < -kPi) delta += 2.0 * kPi;
        delta = wrap_closed_fixup(delta, c.switches);
    }
    if (fabs(delta) > c.cold->rvo_max_dh) { delta = copysign(c.cold->rvo_max_dh, delta); speed = 0.0; }
    a0 = speed; a1 = delta;
}

enum : int { MODE_STEP = 0, MODE_STEP_AUTORESET = 1, MODE_OBSERVE = 2, MODE_RESET = 3, MODE_STEP_AUTORESET_PF = 4, MODE_STEP_AUTORESET_N = 5 };

#ifndef CAVOID_OCC_LARGE_N
#define CAVOID_OCC_LARGE_N 2
#endif
#ifndef CAVOID_OCC4_MAX_N
#define CAVOID_OCC4_MAX_N 4
#endif
// MODE_STEP_AUTORESET    one auto-reset step per launch (a policy in the loop);
// MODE_STEP_AUTORESET_N  io.n_steps steps per launch: a wavefront owns whole worlds and nothing crosses worlds, so the
//                        world state stays in registers between steps; per step only the action slice is read and the
//                        step's outputs (obs, reward, done, game_over) are written; the world buffer is written back once,
//                        after the last step.  This removes the launch boundary and the state round trip from every step
//                        but the first;
// MODE_STEP_AUTORESET_PF the same with the NEXT episode's pool record of every lane held in registers (loaded with the
//                        state, re-loaded after a restart): latency mode for small batches, where a wavefront is alone on
//                        its SIMD and a restart must not cost a dependent trip to memory (may use more registers).
// RVO: the instantiation can drive policy-3 (ORCA) agents and generate box scenarios (GEN v2) inside the step.  A separate
// instantiation because the linear programmes and the generator, inlined into the step, cost every other configuration
// registers (N = 10: +40 VGPRs and scratch) for code it never runs.
// what the last step of a tile left in the lane's registers, for a caller that goes on in the same kernel (the fused actor:
// cavoid_actor.hpp hands it to the experience bookkeeping)
struct StepOut { float reward; bool done, game_over; bool learning_next; };   // learning_next: column 0 of the NEXT observation (after a restart: the new agent's)

// The body of env_kernel for ONE tile (one wavefront): lds_tab = the workgroup's action-table copy, wbase = the wavefront's
// private LDS (staging arrays, obs tile, ORCA scratch), wave = the tile's index.  A device function so that the fused actor
// kernel (policy -> sample -> THIS -> experience push, per tile, in one launch) runs the very same statements.
template <int N, int MODE, bool RVO>
__device__ __forceinline__ void env_tile(const KCfg &c, const KState &s, const PoolRec *pool, const KIO &io, double *lds_tab, float *wbase,
                                         const int lane, const int64_t wave, StepOut *out = nullptr) {
    constexpr bool kAuto = MODE == MODE_STEP_AUTORESET || MODE == MODE_STEP_AUTORESET_PF || MODE == MODE_STEP_AUTORESET_N;
    constexpr bool kLoop = MODE == MODE_STEP_AUTORESET_PF || MODE == MODE_STEP_AUTORESET_N;
    constexpr bool kStepping = MODE == MODE_STEP || kAuto;
    const int width = c.width, ostride = io.obs ? io.obs_stride : width;
    // parking pays where occupancy is the limit -- one step per launch (N = 10: 216 -> 162 VGPRs, 3 wavefronts/SIMD, saturated
    // 272 -> 235 us); the step-loop instantiations stay at 2 wavefronts/SIMD either way and lose ILP to the rolled pair loop
    // (204 -> 212 us), and the RVO instantiations need their LDS for the ORCA lines
    constexpr bool kPark = N >= kParkFromN && !RVO && MODE != MODE_STEP_AUTORESET_PF && MODE != MODE_STEP_AUTORESET_N;
    // the pair pass also makes the neighbours' observation features wherever their 5 (N-1) values can stay in registers
    constexpr bool kFused = !kPark;
    const int tile_need = (c.tile_rows * ostride + 3) & ~3;
    const int tile_floats = tile_need > c.park_floats ? tile_need : c.park_floats;
    StageRec *recs = reinterpret_cast<StageRec *>(wbase);  // the staged post-move records (twice per world from kRingDoubleFromN agents on)
    float *tile = wbase + lds_floats_fixed(N);
    // field-major scratch in the tile region while no rows are in it: the ORCA policy's pre-move state, the box generator's placements
    double *lds_px = reinterpret_cast<double *>(tile);
    double *lds_py = lds_px + 64, *lds_vx = lds_py + 64, *lds_vy = lds_vx + 64;
    float *lds_r = reinterpret_cast<float *>(lds_vy + 64);
    // ... and the float64 velocities the time-to-impact order reads (behind the parked keys / gaps where there are any)
    double *vx64 = reinterpret_cast<double *>(tile + (kPark ? 3 * Others<N>::K * 64 : 0)), *vy64 = vx64 + 64;
    double *rvo_mem = reinterpret_cast<double *>(tile + tile_floats);   // ORCA lines (only with c.rvo_enabled)

    const int wpw = c.wpw, lanes_used = wpw * N;          // worlds / lanes this wavefront really owns
    const int64_t w0 = wave * wpw;                         // first world of this wavefront
    const int lw = lane / N, i = lane - lw * N;
    const int64_t w = w0 + lw;
    const bool active = lane < lanes_used && w < c.num_worlds;
    const int base = lane < lanes_used ? lw * N : 0;       // first lane of this lane's world
    const int64_t a_idx = w * N + i;                       // == w0*N + lane: contiguous per wave
    const bool packed = io.packed != 0;
    int64_t worlds_here = c.num_worlds - w0;
    if (worlds_here > wpw) worlds_here = wpw;
    if (worlds_here < 0) worlds_here = 0;
    CAVOID_STAMP(0);

    // action table -> LDS: the load is issued first and lands together with the state loads; the
    // decode then needs no second trip to memory
    double tab_v = 0.0;
    const bool use_table = kStepping && io.actions != nullptr;
    if (use_table && lane < 2 * c.num_actions) tab_v = c.action_table[lane];

    Agent a;
    a.px = a.py = a.heading = a.t_rem = a.vx = a.vy = 0.0;
    a.gx = a.gy = a.radius = a.pref = a.speed = 0.0f;
    a.flags = 0u;
    uint32_t episode = 0u;
    bool fresh = false;                                    // this lane's world starts a new episode
    int act_next = 0;                                      // the NEXT step's action, loaded one step ahead: the table index -- or, with continuous
    float c1_next = 0.f;                                   // actions (io.cont), the BITS of its first component, the second one in c1_next
    if (active) {
        if (MODE == MODE_RESET) {
            fresh = io.mask == nullptr || io.mask[w] != 0;
            episode = s.episode[w] + (fresh ? 1u : 0u);
        } else if (kAuto) {
            episode = s.episode[w];
        }
        if (!fresh) {
            load_agent(s, a_idx, a);
            if (!kStepping || RVO) a.speed = s.speed[a_idx];   // (RVO agents read the others' last velocities)
        }
        if (kStepping) {
            if (io.cont) { act_next = __float_as_int(io.cont[2 * a_idx]); c1_next = io.cont[2 * a_idx + 1]; }
            else act_next = io.actions[a_idx];
        }
    }
    if (use_table) lds_tab[lane] = tab_v;
    CAVOID_STAMP(1);                                        // loads landed
#ifdef CAVOID_ABLATE
    // development aid: the memory skeleton of ONE step (same loads, same stores, no arithmetic) --
    // the launch + memory floor the real kernel is measured against
    if (kStepping) {
        if (active && lane < c.tile_rows) {
            float *row = tile + lane * ostride;
            for (int k = 0; k < ostride; ++k) row[k] = (float)a.px + (float)k + (float)act_next;
        }
        wave_lds_sync();
        if (io.obs && worlds_here > 0) flush_tile(tile, io.obs + w0 * N * ostride, (int)worlds_here * N * ostride, lane);
        if (active) {
            if (io.rew) io.rew[a_idx] = (float)a.py;
            if (io.done) io.done[a_idx] = 0;
            if (i == 0) io.game_over[w] = 0;
            s.px[a_idx] = a.px + 1.0; s.py[a_idx] = a.py + 1.0; s.heading[a_idx] = a.heading; s.t_rem[a_idx] = a.t_rem - 0.2;
            s.speed[a_idx] = a.pref; s.flags[a_idx] = a.flags;
        }
        return;
    }
#endif
    // latency mode: the NEXT episode's pool entry is fetched up front, together with the state loads, and again
    // right after a restart consumed it -- a restarting world never pays a second dependent trip to memory
    Agent nxt = a;
    constexpr bool kPrefetch = MODE == MODE_STEP_AUTORESET_PF;
    if (kPrefetch && active)
        load_pool(pool, (int64_t)pool_index(c, (uint32_t)(c.world_offset + w), episode + 1u) * N + i, nxt);
    const bool present_first = active && (a.flags & CAVOID_F_PRESENT);   // presence changes only at a restart

    if (MODE == MODE_RESET) {
        if (io.pool_out) episode = c.pool_epoch;               // pool fill: generator worlds 0..P-1 of the pool's epoch
        if (c.gen_mode == 1 && c.pool_size == 0)               // GEN v2 needs the whole wavefront (sequential placement)
            generate_world_v2<N>(c, (uint32_t)(c.world_offset + w), episode, i, base, lane, fresh, lds_px, lds_py, lds_vx, lds_vy,
                                 lds_r, a);
        else if (fresh) new_episode<N>(c, pool, (uint32_t)(c.world_offset + w), episode, i, a);
    }
    if (MODE == MODE_OBSERVE || MODE == MODE_RESET) {
        double sn, cs;
        sincos_bounded(a.heading, &sn, &cs);
        a.vx = (double)a.speed * cs;
        a.vy = (double)a.speed * sn;
    }

    const int n_steps = kLoop ? io.n_steps : 1;
    // (the step loops own their obs tile from step to step -- unless the time-to-impact order parks velocities in it, the ORCA /
    //  box-generator instantiations use it as scratch, or the rows go out in several passes: then every step zeroes all it must)
    const bool tile_persists = kLoop && !RVO && c.sort_method != CAVOID_SORT_TIME_TO_IMPACT && c.tile_rows >= lanes_used;
    int prev_kept = c.max_other;
    bool restarted_any = false, moved_any = false;         // what the write-back after the last step must cover
    const int lane0 = lane, i0 = i, base0 = base;
    const int64_t a_idx0 = a_idx;
    auto write_back = [&]() {
        if (restarted_any) {                               // a fresh episode started: every field of every row
            store_agent(s, a_idx0, a);
            if (i0 == 0) s.episode[w] = episode;
        } else if (present_first) {
            if (moved_any) {                               // agents frozen throughout keep pos / heading / time
                s.px[a_idx0] = a.px; s.py[a_idx0] = a.py; s.heading[a_idx0] = a.heading; s.t_rem[a_idx0] = a.t_rem;
            }
            s.speed[a_idx0] = a.speed;
            s.flags[a_idx0] = a.flags;
        }
    };
    for (int t = 0; t < n_steps; ++t) {
    // Everything derived from the lane id is loop invariant, and the compiler would hoist all of it (LDS addresses of
    // the N-1 others, every output address) out of the step loop into registers that stay live across the whole body --
    // measured: 128 VGPRs + 350 B/lane of scratch instead of 127 VGPRs.  Re-materialising the ids per step keeps the
    // body's register footprint that of the single-step kernel.
    int lane = lane0, i = i0, base = base0;
    int64_t a_idx = a_idx0;
    if (kLoop) asm volatile("" : "+v"(lane), "+v"(i), "+v"(base), "+v"(a_idx));
    CAVOID_STAMP(2);                                        // step t begins
    const uint32_t flags_in = a.flags;
    const bool present_in = active && (flags_in & CAVOID_F_PRESENT);
    const bool done_in = (flags_in & CAVOID_F_DONE_MASK) != 0u;
    int act = act_next;
    const float c1 = c1_next;
    if (kLoop && t + 1 < n_steps && active) {
        if (CAVOID_RARE(io.cont != nullptr)) {
            const float *cn = io.cont + (int64_t)(t + 1) * io.action_stride + 2 * a_idx;
            act_next = __float_as_int(cn[0]); c1_next = cn[1];
        } else act_next = io.actions[(int64_t)(t + 1) * io.action_stride + a_idx];
    }
    const int64_t slot_w = kLoop ? (int64_t)t * io.out_step_stride : 0;   // first world row of this step's output slot

    if (kStepping) {
        // ---- E4 decode ------------------------------------------------------------------------------
        wave_lds_sync();
        const uint32_t pol = (flags_in >> CAVOID_F_POLICY_SHIFT) & CAVOID_F_POLICY_MASK;   // (4 = frozen network: its action index comes in like a learner's)
        double a0 = 0.0, a1 = 0.0;
        if (io.cont) { a0 = (double)__int_as_float(act); a1 = (double)c1; }
        else {
            act = act < 0 ? 0 : (act >= c.num_actions ? c.num_actions - 1 : act);
            a0 = (double)a.pref * lds_tab[2 * act];
            a1 = lds_tab[2 * act + 1];
        }
        if (CAVOID_RARE(__ballot(present_in && !done_in && pol != 0u) != 0ull)) {   // scripted agents in this tile
            if (pol == 1u) { a0 = 0.0; a1 = 0.0; }
            if (pol == 2u) {                                            // straight at the goal, full speed
                const Ego e0 = ego_frame_exact(c, a);
                a0 = (double)a.pref;
                a1 = -e0.heading_ego;
            }
        }
        if (RVO && CAVOID_RARE(__ballot(present_in && !done_in && pol == 3u) != 0ull)) {   // RVO agents in this tile
            // stage the PRE-move state of every agent: position, last velocity (speed along the heading), radius
            double sn, cs;
            sincos_bounded(a.heading, &sn, &cs);
            lds_px[lane] = a.px; lds_py[lane] = a.py;
            lds_vx[lane] = present_in ? (double)a.speed * cs : 0.0;
            lds_vy[lane] = present_in ? (double)a.speed * sn : 0.0;
            lds_r[lane] = present_in ? a.radius : -1.0f;
            wave_lds_sync();
            if (present_in && !done_in && pol == 3u)
                rvo_action<N>(c, a, i, base, lane, lds_px, lds_py, lds_vx, lds_vy, lds_r, rvo_mem, a0, a1);
            wave_lds_sync();
        }
        if (c.actions_fp32) { a0 = (double)(float)a0; a1 = (double)(float)a1; }
        // ---- E5 dynamics (computed by every lane, committed only by agents still running) ------------
        const bool moving = present_in && !done_in;
        moved_any = moved_any || moving;
        double npx, npy, nh, nvx, nvy, nsp;
        if (CAVOID_RARE(c.dynamics == CAVOID_DYN_HOLONOMIC)) {
            nsp = sqrt(a0 * a0 + a1 * a1);
            nh = nsp > 0.0 ? atan2(a1, a0) : a.heading;
            npx = a.px + a0 * c.dt; npy = a.py + a1 * c.dt;
            nvx = a0; nvy = a1;
        } else {
            double dh = a1;
            if (CAVOID_RARE(c.dynamics == CAVOID_DYN_UNICYCLE_MAX_TURN)) {
                const double rate = fmin(fmax(dh / c.dt, -c.cold->max_turn_rate), c.cold->max_turn_rate);
                dh = rate * c.dt;
            }
            nh = wrap_angle(dh + a.heading, c.switches);
            double sn = 0.0, cs = 1.0;
            if (!(CAVOID_SKIP & 32)) sincos_bounded(nh, &sn, &cs);
#if defined(CAVOID_DEV_ULP_FAULT) && CAVOID_DEV_ULP_FAULT == 2     /* injected fault: the position update contracted into fused multiply-adds */
            npx = __builtin_fma(a0 * cs, c.dt, a.px); npy = __builtin_fma(a0 * sn, c.dt, a.py);
#else
            npx = a.px + a0 * cs * c.dt; npy = a.py + a0 * sn * c.dt;
#endif
            nvx = a0 * cs; nvy = a0 * sn; nsp = a0;
        }
        a.px = moving ? npx : a.px; a.py = moving ? npy : a.py; a.heading = moving ? nh : a.heading;
        a.vx = moving ? nvx : 0.0; a.vy = moving ? nvy : 0.0; a.speed = moving ? (float)nsp : 0.0f;
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
    }

    CAVOID_STAMP(3);                                        // dynamics done
    // ---- stage post-move state in LDS; E6 pair pass (ego frame in the same block: independent chains) ---
    bool present = active && (a.flags & CAVOID_F_PRESENT);
    constexpr bool kDouble = RingStage<N>::kDouble;
    const int rec_self = kDouble ? 2 * base + i : lane;    // doubled: world lw's two copies start at record 2 * base
    const bool tti = c.sort_method == CAVOID_SORT_TIME_TO_IMPACT;
    auto stage_self = [&](bool is_present) {               // (the doubled layout has no records for the lanes behind the last world)
        if (!kDouble || lane < lanes_used) {
            const StageRec me{a.px, a.py, (float)a.vx, (float)a.vy, is_present ? a.radius : -1.0f, 0.0f};   // radius < 0 marks an absent row
            recs[rec_self] = me;
            if (kDouble) recs[rec_self + N] = me;
        }
        if (CAVOID_RARE(tti)) { vx64[lane] = a.vx; vy64[lane] = a.vy; }
    };
    stage_self(present);
    wave_lds_sync();
    const RingStage<N> st{recs + rec_self, vx64, vy64, i, base};
    Ego e = ego_frame_obs(c, a);
    Key key[Others<N>::K];
    float gapf[Others<N>::K];
    float feat[Others<N>::K][kFeat];                       // the neighbours' features, made by the pair pass (kFused)
    uint32_t valid;
    bool hit;
    double min_gap;
    uint32_t frozen_w = 0u;                                 // U4 flipped: the agents of this lane's world that were done before the step
    if (kStepping && CAVOID_RARE(c.switches & kSwSkipDonePairs))
        frozen_w = (uint32_t)(__ballot(present_in && done_in) >> base) & ((1u << N) - 1u);
    pair_pass<N, kPark, kFused>(c, a, e, present, st, key, gapf, feat, valid, hit, min_gap, reinterpret_cast<uint32_t *>(tile), lane, frozen_w);

    CAVOID_STAMP(4);                                        // ego frame + pair pass done
    float rew_f = 0.0f, done_f = (present && (a.flags & CAVOID_F_DONE_MASK) == 0u) ? 0.0f : 1.0f;   // reset / observe, packed
    if (kStepping) {
        // ---- E7 rewards, E8 done ---------------------------------------------------------------------
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
        const unsigned long long running = __ballot(present && ((a.flags & CAVOID_F_LEARNING) || c.evaluate_mode) && !done);
        const unsigned long long wmask = ((1ull << N) - 1ull) << base;
        const bool game_over = (running & wmask) == 0ull;
        rew_f = (float)r;
        done_f = done ? 1.0f : 0.0f;
        if (out) { out->reward = rew_f; out->done = done; out->game_over = game_over; }
        if (active) {
            if (!packed) {
                io.rew[slot_w * N + a_idx] = rew_f;
                io.done[slot_w * N + a_idx] = done ? 1 : 0;
            }
            if (i == 0) io.game_over[slot_w + w] = game_over ? 1 : 0;
        }
        if (kAuto) {
            const bool restart = active && game_over;
            if (CAVOID_RARE(__ballot(restart) != 0ull)) {               // wave-uniform: some world of this tile restarts
                wave_lds_sync();                           // every lane is done reading the old positions
                if (restart) { episode += 1u; restarted_any = true; }
                // GEN v2 without a pool: the box generator places a world's agents one after the other, so the whole wavefront
                // takes part (the restarting worlds' lanes draw, round by round, against what their world has placed so far;
                // the staging arrays of the OTHER worlds' lanes are not touched)
                // (carried by the RVO = true instantiations only -- the 'everything' build of the step: inlined into the plain
                //  ones it costs the one-step kernel its occupancy, N = 10: 159 -> 210 VGPRs)
                const bool box_in_step = RVO && !kPrefetch && c.gen_mode == 1 && c.pool_size == 0;     // wave-uniform
                if (box_in_step)
                    generate_world_v2<N>(c, (uint32_t)(c.world_offset + w), episode, i, base, lane, restart, lds_px, lds_py, lds_vx, lds_vy,
                                         lds_r, a);
                if (restart) {
                    if (kPrefetch) {
                        a = nxt;
                        if (t + 1 < n_steps)               // re-arm: the record of the episode after this one
                            load_pool(pool, (int64_t)pool_index(c, (uint32_t)(c.world_offset + w), episode + 1u) * N + i, nxt);
                    } else if (!box_in_step) new_episode<N>(c, pool, (uint32_t)(c.world_offset + w), episode, i, a);
                    present = (a.flags & CAVOID_F_PRESENT) != 0u;
                    stage_self(present);                   // (a fresh agent stands: a.vx = a.vy = 0)
                }
                wave_lds_sync();
                if (restart) {
                    e = ego_frame_obs(c, a);
                    bool hit2;
                    double gap2;
                    pair_pass<N, kPark, kFused>(c, a, e, present, st, key, gapf, feat, valid, hit2, gap2, reinterpret_cast<uint32_t *>(tile), lane);
                }
            }
        }
    }

    CAVOID_STAMP(5);                                        // rewards / restart done
    if (out) out->learning_next = active && (a.flags & CAVOID_F_PRESENT) != 0u && (a.flags & CAVOID_F_LEARNING) != 0u;
    // one step per launch: the state is final here -- its stores complete under the observation phase instead of behind it, where the
    // kernel's end waited for them (same box: 4 x 8192 5.99 -> 5.69 us, 10 x 8192 12.8 -> 12.4; saturated launches pay 3 % for it:
    // 4 x 262144 44.9 -> 46.4 us.  Chosen at run time by batch size, the two copies of the stores cost more than either gains.)
    if (kStepping && !kLoop) write_back();
    // ---- E9 observation: once per step, after the restart decision -----------------------------------
    if (io.obs && !(CAVOID_SKIP & 1)) {
        CAVOID_STAMP(6);
        assemble_obs<N, kPark, kFused>(c, a, e, active, lane, st, key, gapf, feat, valid, tile,
                        io.obs + (slot_w + w0) * N * ostride, (int)worlds_here * N, ostride, packed, rew_f, done_f, wave, NoHook(),
                        kLoop ? (io.out_step_stride != 0 && !(MODE == MODE_STEP_AUTORESET_N && N <= CAVOID_OCC4_MAX_N))
                              : (N > CAVOID_OCC4_MAX_N && c.stream_obs != 0),
                        tile_persists ? &prev_kept : nullptr);
        // (a loop that overwrites ONE slot keeps it in the L2; the one-step kernels of up to 4 agents sit at the 128-register cliff of
        //  four wavefronts per SIMD -- a second, streaming copy of the flush cost env_kernel<4, 1> 194 spilled registers and 4 us,
        //  and the large-batch loop env_kernel<4, MODE_STEP_AUTORESET_N> 11 more, 1771 -> 2008 us per 16 steps at 4 x 1048576)
    }
    CAVOID_STAMP(7);                                        // tile flushed
    if (kLoop && n_steps > 1) wave_lds_sync();             // the next step re-stages the LDS arrays and the tile
    }   // step loop

    // ---- state write-back (once per launch; the one-step forms did it before the observation, see there) ---------------
    if (kStepping && kLoop) write_back();
    if (MODE == MODE_RESET) {
        if (fresh && io.pool_out) {                            // pool fill: one 64-byte record per agent
            PoolRec r;
            r.px = a.px; r.py = a.py; r.heading = a.heading; r.t_rem = a.t_rem;
            r.gx = a.gx; r.gy = a.gy; r.radius = a.radius; r.pref = a.pref;
            r.flags = a.flags; r.pad[0] = r.pad[1] = r.pad[2] = 0u;
            io.pool_out[a_idx] = r;
        } else if (fresh) {
            store_agent(s, a_idx, a);
            if (i == 0) s.episode[w] = episode;
        }
    }
    CAVOID_STAMP(8);
}

template <int N, int MODE, bool RVO>
// (second launch-bound = min wavefronts per SIMD: small-N instantiations sit right at the 128-VGPR cliff;
//  pin them to 4 wavefronts/SIMD -- the LDS tile admits no more anyway -- at the price of a 1-register spill)
__global__ void __launch_bounds__(256, (MODE == MODE_STEP_AUTORESET_PF ? 2 : (N <= CAVOID_OCC4_MAX_N ? 4 : CAVOID_OCC_LARGE_N)))
env_kernel(const KCfg c, const KState s, const PoolRec *pool, const KIO io) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave_in_block = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ostride = io.obs ? io.obs_stride : c.width;
    const int tile_need = (c.tile_rows * ostride + 3) & ~3;
    const int tile_floats = tile_need > c.park_floats ? tile_need : c.park_floats;
    const int per_wave_floats = lds_floats_fixed(N) + tile_floats + c.rvo_lds_floats;
    double *lds_tab = reinterpret_cast<double *>(smem);
    float *wbase = reinterpret_cast<float *>(smem) + lds_floats_block() + (size_t)wave_in_block * per_wave_floats;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block;
    env_tile<N, MODE, RVO>(c, s, pool, io, lds_tab, wbase, lane, wave);
}


// ---- MODE_STEP_AUTORESET_PIPE: the step loop as a two-wavefront pipeline (latency mode: small batches) ----------------
// At 4 x 8192 there are 512 tiles for 1024 SIMDs and a step is ONE wavefront's dependent chain.  Here a tile is owned by a
// workgroup of TWO wavefronts on two SIMDs of a CU:
//   wavefront P (producer): action decode, dynamics, staging, pair pass, rewards, done / game_over, restart -- step t;
//   wavefront C (consumer): ego frame, neighbour ranking, observation rows, tile flush -- step t-1,
// handing over through double-buffered LDS (the staged post-move state + one record per lane: goal offset, heading, the
// sort keys and gaps of the pair pass, flags, reward, done), one workgroup barrier per step.  Same arithmetic, same
// operation order per value as env_kernel: outputs are bit-identical (tests/test_gpu_packed.py).  The state lives in P's
// registers for the whole launch; P holds every lane's NEXT pool record like MODE_STEP_AUTORESET_PF.
template <int N>
struct PipeRec {                       // per-buffer hand-over record, field-major over the 64 lanes
    static constexpr int K = Others<N>::K;
    double tx[64], ty[64], heading[64];
    float pref[64], radius[64], rew[64], done[64];
    uint32_t flags[64], valid[64];
    uint64_t key[K][64];
    float gap[K][64];
};
struct PipeStage { double px[64], py[64], vx[64], vy[64]; float r[64]; };

template <int N>
__host__ __device__ constexpr size_t pipe_lds_fixed_bytes() {
    return (size_t)lds_floats_block() * sizeof(float) + 2 * sizeof(PipeStage) + 2 * sizeof(PipeRec<N>);
}

template <int N, bool RVO>
__global__ void __launch_bounds__(128, 1) env_pipe_kernel(const KCfg c, const KState s, const PoolRec *pool, const KIO io) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *lds_tab = reinterpret_cast<double *>(smem);
    PipeStage *stage = reinterpret_cast<PipeStage *>(smem + lds_floats_block() * sizeof(float));
    PipeRec<N> *rec = reinterpret_cast<PipeRec<N> *>(stage + 2);
    float *tile = reinterpret_cast<float *>(rec + 2);
    const int width = c.width, ostride = io.obs_stride;
    const int tile_need = (c.tile_rows * ostride + 3) & ~3;
    const int tile_floats = tile_need > c.park_floats ? tile_need : c.park_floats;
    double *rvo_mem = reinterpret_cast<double *>(tile + tile_floats);
    const bool producer = (threadIdx.x >> 6) == 0;
    const int lane0 = threadIdx.x & 63;
    const int wpw = c.wpw, lanes_used = wpw * N;
    const int64_t wave = blockIdx.x;                       // one tile per workgroup
    const int64_t w0 = wave * wpw;
    const int lw = lane0 / N, i0 = lane0 - lw * N;
    const int64_t w = w0 + lw;
    const bool active = lane0 < lanes_used && w < c.num_worlds;
    const int base0 = lane0 < lanes_used ? lw * N : 0;
    const int64_t a_idx0 = w * N + i0;
    const bool packed = io.packed != 0;
    int64_t worlds_here = c.num_worlds - w0;
    if (worlds_here > wpw) worlds_here = wpw;
    if (worlds_here < 0) worlds_here = 0;
    const int n_steps = io.n_steps;

    // ---- producer state ------------------------------------------------------------------------------------------------
    Agent a;
    a.px = a.py = a.heading = a.t_rem = a.vx = a.vy = 0.0;
    a.gx = a.gy = a.radius = a.pref = a.speed = 0.0f;
    a.flags = 0u;
    Agent nxt = a;
    uint32_t episode = 0u;
    int act_next = 0;
    bool restarted_any = false, moved_any = false, present_first = false;
    const bool prefetch = c.pool_size > 0;
    if (producer) {
        double tab_v = 0.0;
        if (lane0 < 2 * c.num_actions) tab_v = c.action_table[lane0];
        if (active) {
            episode = s.episode[w];
            load_agent(s, a_idx0, a);
            if (RVO) a.speed = s.speed[a_idx0];
            act_next = io.actions[a_idx0];
            if (prefetch) load_pool(pool, (int64_t)pool_index(c, (uint32_t)(c.world_offset + w), episode + 1u) * N + i0, nxt);
        }
        lds_tab[lane0] = tab_v;
        present_first = active && (a.flags & CAVOID_F_PRESENT);
    }

    for (int k = 0; k <= n_steps; ++k) {
        int lane = lane0, i = i0, base = base0;
        int64_t a_idx = a_idx0;
        asm volatile("" : "+v"(lane), "+v"(i), "+v"(base), "+v"(a_idx));   // keep lane-derived values out of loop-invariant registers
        if (producer) {
            if (k < n_steps) {
                const int t = k;
                PipeStage &st = stage[t & 1];
                PipeRec<N> &rc = rec[t & 1];
                const uint32_t flags_in = a.flags;
                const bool present_in = active && (flags_in & CAVOID_F_PRESENT);
                const bool done_in = (flags_in & CAVOID_F_DONE_MASK) != 0u;
                CAVOID_STAMP(2);
                int act = act_next;
                if (t + 1 < n_steps && active) act_next = io.actions[(int64_t)(t + 1) * io.action_stride + a_idx];
                // ---- E4 decode -----------------------------------------------------------------------------------------
                wave_lds_sync();
                const uint32_t pol = (flags_in >> CAVOID_F_POLICY_SHIFT) & CAVOID_F_POLICY_MASK;   // (4 = frozen network: its action index comes in like a learner's)
                double a0 = 0.0, a1 = 0.0;
                act = act < 0 ? 0 : (act >= c.num_actions ? c.num_actions - 1 : act);
                a0 = (double)a.pref * lds_tab[2 * act];
                a1 = lds_tab[2 * act + 1];
                if (CAVOID_RARE(__ballot(present_in && !done_in && pol != 0u) != 0ull)) {   // scripted agents in this tile
                    if (pol == 1u) { a0 = 0.0; a1 = 0.0; }
                    if (pol == 2u) {
                        const Ego e0 = ego_frame_exact(c, a);
                        a0 = (double)a.pref;
                        a1 = -e0.heading_ego;
                    }
                }
                if (RVO && __ballot(present_in && !done_in && pol == 3u) != 0ull) {
                    double sn, cs;
                    sincos_bounded(a.heading, &sn, &cs);
                    st.px[lane] = a.px; st.py[lane] = a.py;
                    st.vx[lane] = present_in ? (double)a.speed * cs : 0.0;
                    st.vy[lane] = present_in ? (double)a.speed * sn : 0.0;
                    st.r[lane] = present_in ? a.radius : -1.0f;
                    wave_lds_sync();
                    if (present_in && !done_in && pol == 3u)
                        rvo_action<N>(c, a, i, base, lane, st.px, st.py, st.vx, st.vy, st.r, rvo_mem, a0, a1);
                    wave_lds_sync();
                }
                if (c.actions_fp32) { a0 = (double)(float)a0; a1 = (double)(float)a1; }
                // ---- E5 dynamics ---------------------------------------------------------------------------------------
                const bool moving = present_in && !done_in;
                moved_any = moved_any || moving;
                double dh = a1;
                if (CAVOID_RARE(c.dynamics == CAVOID_DYN_UNICYCLE_MAX_TURN)) {
                    const double rate = fmin(fmax(dh / c.dt, -c.cold->max_turn_rate), c.cold->max_turn_rate);
                    dh = rate * c.dt;
                }
                const double nh = wrap_angle(dh + a.heading, c.switches);
                double sn, cs;
                sincos_bounded(nh, &sn, &cs);
                const double npx = a.px + a0 * cs * c.dt, npy = a.py + a0 * sn * c.dt;
                const double nvx = a0 * cs, nvy = a0 * sn, nsp = a0;
                a.px = moving ? npx : a.px; a.py = moving ? npy : a.py; a.heading = moving ? nh : a.heading;
                a.vx = moving ? nvx : 0.0; a.vy = moving ? nvy : 0.0; a.speed = moving ? (float)nsp : 0.0f;
                if (present_in && done_in) {
                    if (flags_in & CAVOID_F_AT_GOAL) a.flags |= CAVOID_F_WAS_AT_GOAL;
                    if (flags_in & CAVOID_F_IN_COLL) a.flags |= CAVOID_F_WAS_IN_COLL;
                }
                if (moving) {
                    const double dx = a.px - (double)a.gx, dy = a.py - (double)a.gy;
                    if (dx * dx + dy * dy <= c.near_goal_sq) a.flags |= CAVOID_F_AT_GOAL;
                    a.t_rem -= c.dt;
                    if (c.timeout_enabled && a.t_rem <= 0.0) a.flags |= CAVOID_F_RAN_OUT;
                }
                CAVOID_STAMP(3);
                // ---- stage, E6 pair pass -------------------------------------------------------------------------------
                bool present = active && (a.flags & CAVOID_F_PRESENT);
                st.px[lane] = a.px; st.py[lane] = a.py; st.vx[lane] = a.vx; st.vy[lane] = a.vy;
                st.r[lane] = present ? a.radius : -1.0f;
                wave_lds_sync();
                Ego e;                                                  // the pair pass needs the goal offset only
                e.tx = (double)a.gx - a.px; e.ty = (double)a.gy - a.py;
                Key key[Others<N>::K];
                float gapf[Others<N>::K];
                uint32_t valid;
                bool hit;
                double min_gap;
                uint32_t frozen_w = 0u;
                if (CAVOID_RARE(c.switches & kSwSkipDonePairs)) frozen_w = (uint32_t)(__ballot(present_in && done_in) >> base) & ((1u << N) - 1u);
                const ArrayStage<N> as{st.px, st.py, st.vx, st.vy, st.r, i, base};
                pair_pass<N, false, false>(c, a, e, present, as, key, gapf, nullptr, valid, hit, min_gap, nullptr, 0, frozen_w);
                CAVOID_STAMP(4);
                // ---- E7 rewards, E8 done -------------------------------------------------------------------------------
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
                if (active) {
                    const int64_t slot_w = (int64_t)t * io.out_step_stride;
                    if (!packed) {
                        io.rew[slot_w * N + a_idx] = rew_f;
                        io.done[slot_w * N + a_idx] = done ? 1 : 0;
                    }
                    if (i == 0) io.game_over[slot_w + w] = game_over ? 1 : 0;
                }
                const bool restart = active && game_over;
                if (__ballot(restart) != 0ull) {
                    wave_lds_sync();
                    if (restart) {
                        episode += 1u;
                        restarted_any = true;
                        if (prefetch) {
                            a = nxt;
                            if (t + 1 < n_steps)
                                load_pool(pool, (int64_t)pool_index(c, (uint32_t)(c.world_offset + w), episode + 1u) * N + i, nxt);
                        } else generate_agent<N>(c, (uint32_t)(c.world_offset + w), episode, i, a);
                        present = (a.flags & CAVOID_F_PRESENT) != 0u;
                        st.px[lane] = a.px; st.py[lane] = a.py; st.vx[lane] = 0.0; st.vy[lane] = 0.0;
                        st.r[lane] = present ? a.radius : -1.0f;
                    }
                    wave_lds_sync();
                    if (restart) {
                        e.tx = (double)a.gx - a.px; e.ty = (double)a.gy - a.py;
                        bool hit2;
                        double gap2;
                        pair_pass<N, false, false>(c, a, e, present, as, key, gapf, nullptr, valid, hit2, gap2);
                    }
                }
                CAVOID_STAMP(5);
                // ---- hand over to the consumer ---------------------------------------------------------------------------
                rc.tx[lane] = e.tx; rc.ty[lane] = e.ty; rc.heading[lane] = a.heading;
                rc.pref[lane] = a.pref; rc.radius[lane] = a.radius; rc.rew[lane] = rew_f; rc.done[lane] = done_f;
                rc.flags[lane] = a.flags; rc.valid[lane] = valid;
#pragma unroll
                for (int o = 0; o < N - 1; ++o) { rc.key[o][lane] = key[o].v; rc.gap[o][lane] = gapf[o]; }
                CAVOID_STAMP(8);
            }
        } else if (k >= 1) {
            // ---- consumer: E9 of step k-1 ----------------------------------------------------------------------------------
            const PipeStage &st = stage[(k - 1) & 1];
            const PipeRec<N> &rc = rec[(k - 1) & 1];
            CAVOID_STAMP(6);
            Agent ao;
            ao.px = st.px[lane]; ao.py = st.py[lane]; ao.vx = st.vx[lane]; ao.vy = st.vy[lane];
            ao.heading = rc.heading[lane]; ao.t_rem = 0.0;
            ao.gx = ao.gy = ao.speed = 0.0f;
            ao.radius = rc.radius[lane]; ao.pref = rc.pref[lane]; ao.flags = rc.flags[lane];
            const Ego e = ego_from(c, rc.tx[lane], rc.ty[lane], ao.heading);
            Key key[Others<N>::K];
            float gapf[Others<N>::K];
            key[0].set(kKeySentinel, 0u); gapf[0] = 0.0f;
#pragma unroll
            for (int o = 0; o < N - 1; ++o) { key[o].v = rc.key[o][lane]; gapf[o] = rc.gap[o][lane]; }
            const ArrayStage<N> as{st.px, st.py, st.vx, st.vy, st.r, i, base};
            assemble_obs<N, false, false>(c, ao, e, active, lane, as, key, gapf, nullptr, rc.valid[lane], tile,
                            io.obs + ((int64_t)(k - 1) * io.out_step_stride + w0) * N * ostride, (int)worlds_here * N, ostride, packed,
                            rc.rew[lane], rc.done[lane], wave, NoHook(), io.out_step_stride != 0);
            CAVOID_STAMP(7);
        }
        __syncthreads();
        if (producer) CAVOID_STAMP(1); else CAVOID_STAMP(0);                                   // buffer (k & 1) is published, buffer ((k-1) & 1) is free again
    }

    // ---- state write-back (producer, once per launch) ---------------------------------------------------------------------
    if (producer) {
        if (restarted_any) {
            store_agent(s, a_idx0, a);
            if (i0 == 0) s.episode[w] = episode;
        } else if (present_first) {
            if (moved_any) {
                s.px[a_idx0] = a.px; s.py[a_idx0] = a.py; s.heading[a_idx0] = a.heading; s.t_rem[a_idx0] = a.t_rem;
            }
            s.speed[a_idx0] = a.speed;
            s.flags[a_idx0] = a.flags;
        }
    }
}

}  // namespace cavoid
