// cavoid_relay.hpp -- env_relay_kernel<N>: the in-launch step loop for SMALL batches, one workgroup per tile, the step cut
// into roles that run on different wavefronts (different SIMDs of one CU) and relay through LDS.
//
// Why: at 4 agents x 8192 worlds there are 512 tiles for 1024 SIMDs, so the time of a step is the LENGTH OF ONE
// DEPENDENT CHAIN (action -> sincos -> position -> LDS -> sqrt -> gap -> flags -> ballot -> restart -> next action ...), not
// issue or memory bandwidth.  env_pipe_kernel took the observation off that chain (two wavefronts per tile); this kernel
// also takes the DYNAMICS off it.  The only true recurrence of the step is
//      positions(t) -> pair pass -> new collisions / game over (t) -> which agents move, which worlds restart at t+1,
// everything else of step t+1 -- decode, heading, sincos, position, goal test, time budget -- does not need the pair pass
// of step t unless an agent collided or the world restarted at t.  So:
//
//   D  (state owner)   holds the tile's state in registers for the whole launch.  While P works on step t it computes the
//                      successor T_c = advance(T(t), action(t+1)) as if nothing happened at t and POSTS it into ring slot t+1;
//                      then it waits for P's verdict on step t.  Only a surprise -- a collision found at t, or a world
//                      restarting (a wave-uniform test) -- makes it select (collided agent -> frozen copy, restarted world ->
//                      the next pool record, patched into slot t for the consumers and advanced on the spot) and re-post.
//   P  (pair pass)     distances, collision test, nearest gap, reward, done, game_over of step t from the posted tentative
//                      state; verdict to D; plain outputs.  It makes the same surprise test on its own verdict: no surprise ->
//                      it goes straight on with the successor D posted long ago, and the loop-carried chain is this wavefront
//                      alone; surprise -> it waits for D's corrected state.
//   C0..C{NC-1}        observation of step t (ego frame, distances and sort keys, ranking, rows, coalesced flush) from slot t
//                      (state + verdict) once D has declared it final, steps dealt round-robin: each consumer has NC step
//                      times per step.
//   L  (loader)        everything that touches memory inside the loop and is not an output: the action block (a 64-step
//                      byte ring in LDS, filled ahead of D) and the next scenario-pool record of every restarted lane.
//   D + P + L          (N >= 4) the observation of the LAST step, together: one neighbour slot in three each, the mapping of
//                      cavoid_quad.hpp -- all three are idle once the last step is settled, and the launch's tail is one
//                      observation latency behind that moment (relay_coop_last).
//
// Same arithmetic per value, in the same operation order, as env_kernel: outputs and state are bit-identical
// (tests/test_gpu_packed.py, test_gpu_parity.py run through this kernel by default).  Synchronisation: sequence counters in
// LDS, polled (s_sleep); an LDS write of a wavefront is visible to the workgroup in issue order, so "data, then counter" on
// the writer and "counter, then data" on the reader is enough -- no workgroup barrier and no vmcnt drain inside the loop.
#pragma once
#include "cavoid_kernels.hpp"

namespace cavoid {

constexpr int kRelayMaxAgents = 6;        // instantiated (and faster than the two-wavefront pipeline) up to this many agents per world
// depth of the state / verdict rings (steps in flight between D, P and the consumers).  D takes slot (t + 1) % ring when the consumer of
// step t + 1 - ring has flushed: with 4 slots an observation has 2 periods + D's advance to get from "final" to "flushed" -- 7.3 k clocks at
// the 2.9 k period of N = 4, against a median observation of 6.3 k and a 90th percentile of 7.3 k: the slowest tiles' D waited for a slot,
// and a shorter D iteration bought nothing.  8 slots (18 KB more LDS: two workgroups per CU still fit 160 KB up to N = 5) take the
// consumers' LATENCY out of the period; what is left is their throughput (a third of an observation per step each).
template <int N>
__host__ __device__ constexpr int relay_ring() { return N <= 5 ? 8 : 4; }
constexpr size_t kRelayLdsLimit = 80 * 1024;      // per workgroup: two of them resident per CU (160 KB)
constexpr int kRelayActRing = 64;        // action ring: steps
constexpr int kRelayActAhead = 48;       // the loader runs at most this many steps ahead of D
constexpr int kRelayEvq = 8;             // restart-event queue D -> L

struct RelaySeq {                        // sequence counters (each written by exactly one wavefront)
    int spec;                            // D: the SPECULATIVE tentative state of steps < spec is in `tent` (nothing happened at the step before)
    int stage;                           // D: ... and corrected for the verdict of the step before (collisions, restarts)
    int res;                             // P: 2 * (verdicts posted: steps < that many are in `res`) + (the last one holds a surprise)
    int fin;                             // D: the state and verdict slots of steps < fin are final (restarted worlds patched in)
    int act;                             // L: actions of steps < act are in the ring
    int ev;                              // D: restart events posted (it has read the old records of the restarted lanes)
    int nxt;                             // L: 1 + restart events served (1: the first pool records are in; then re-armed after every restart)
    int cons[kRelayMaxConsumers];        // C: steps < cons[c] of consumer c's share are flushed (ring slots free)
    int cfin[kRelayMaxConsumers];        // C: all of the consumer's stores have completed
    int coop[2];                         // D, P, L (the last step's observation, made together): arrivals at its two hand-overs (LDS atomics)
    int kernarg[2];                      // ... and what they start from: the kernel-argument segment's address and the tile (written once, at entry)
    int tile_id;
};
static_assert(sizeof(RelaySeq) % 16 == 0, "the event queue behind it holds 64-bit masks; the tiles further on are read 16 bytes at a time");
struct RelayTent { double px[64], py[64], vx[64], vy[64], heading[64]; float r[64], gx[64], gy[64], pref[64]; uint32_t flags[64]; };
struct RelayRes { uint32_t flags[64], ctl[64]; float rew[64]; };                            // ctl: bit 0 done, bit 1 the lane's world restarts
struct RelayNxt { double px[64], py[64], heading[64], t_rem[64]; float gx[64], gy[64], radius[64], pref[64]; uint32_t flags[64]; };
// Which wavefront of the workgroup carries which role (0 D, 1 P, 2 .. 1+NC consumers, 2+NC L).  Wavefront w of a workgroup lands on SIMD
// (first + w) % 4 and the next workgroup of the CU goes on where this one ended (tools/ubench/wave_placement.hip): with six wavefronts per
// workgroup and two workgroups per CU the EVEN wavefronts of both tiles share two SIMDs and the odd ones the other two.  Once the rings
// no longer hid it (relay_ring), the placement is worth +-10 % (profiles/r06_u_relay_role_order_ab.txt, r06_v): best is D, P and one
// consumer on the even wavefronts -- each D beside the OTHER tile's P -- and the loader with two consumers on the odd ones (-2 % against
// the plain order D P C0 C1 C2 L; P beside two consumers +12 %; D beside its own P +1 %; padding wavefronts that shift the pattern 0 %).
#ifndef CAVOID_RELAY_ORDER
#define CAVOID_RELAY_ORDER 0x432150      /* one nibble per wavefront, wavefront 0 lowest: D L P C0 C1 C2 */
#endif
// Issue priority of the roles (s_setprio: a SIMD's arbiter takes the ready wavefront of the highest priority).  D and P carry the loop-carried
// cycle and stay on top; the consumers ABOVE the loader (it only polls between its rare refills) is worth -1.4 % (K = 20) / -2.4 % (K = 64)
// against consumers 0 / loader 1; P or D one level down +-0.5 %; no priorities at all +4 ... +6 % (profiles/r06_w_relay_prio_ablation.txt); D, P and the
// loader all at 3 while they make the last step's observation together: +-0 (profiles/r06_ad_tail_prio.txt).
#ifndef CAVOID_RELAY_PRIO_D
#define CAVOID_RELAY_PRIO_D 3
#endif
#ifndef CAVOID_RELAY_PRIO_P
#define CAVOID_RELAY_PRIO_P 3
#endif
#ifndef CAVOID_RELAY_PRIO_L
#define CAVOID_RELAY_PRIO_L 0
#endif
#ifndef CAVOID_RELAY_PRIO_C
#define CAVOID_RELAY_PRIO_C 1
#endif
// development: timing-only ablations of the roles (WRONG results; profiles/r06_w_relay_prio_ablation.txt): 1 the consumers make no observation
// (they only free their ring slots), 2 P's distance loop left out, 4 D's advance without its sine / cosine, 8 D never waits for P's verdict and
// nobody acts on a surprise (the loop-carried cycle cut: what unbounded speculation would run at), 16 D waits for the verdict of the step BEFORE
// the one it waits for now (the timing of a speculation two steps deep)
#ifndef CAVOID_RELAY_ABL
#define CAVOID_RELAY_ABL 0
#endif
__device__ __forceinline__ int relay_role_of(int wv, int NC) {
    if (NC == 3) return (int)(((unsigned)CAVOID_RELAY_ORDER >> (4 * wv)) & 15u);
    return wv;                                              // (other consumer counts: the plain order)
}
// The LAST step's observation is made by three wavefronts together (D, P and L: all idle once the last step is settled), one
// neighbour slot in three each -- the mapping of cavoid_quad.hpp -- instead of by the consumer whose turn it would be: the launch's
// tail is one observation latency behind D's last iteration (profiles/r05_g_relay_launch_timeline.txt), and that latency is a
// single wavefront's dependent chain.  From kRelayCoopFromN agents per world on (every wavefront has a neighbour to work on).
constexpr int kRelayCoopWaves = 3;
constexpr int kRelayCoopFromN = 4;
template <int N>
struct RelayCoop {                       // what the three hand each other: every neighbour's sort key and "seen" bit, by host lane
    static constexpr int K = Others<N>::K;
    uint64_t key[K][64];
    uint32_t bits[K][64];
};
template <int N>
__host__ __device__ constexpr size_t relay_coop_bytes() { return N >= kRelayCoopFromN ? sizeof(RelayCoop<N>) : 0; }
template <int N>
__host__ __device__ constexpr size_t relay_lds_fixed_bytes() {
    return (size_t)lds_floats_block() * sizeof(float) + sizeof(RelaySeq) + kRelayActRing * 64 + relay_ring<N>() * (sizeof(RelayTent) + sizeof(RelayRes)) +
           sizeof(RelayNxt) + kRelayEvq * sizeof(unsigned long long) + relay_coop_bytes<N>();
}

// development build: lane 0 of a role stamps the shader clock of step n_steps/2 into g_trace[tile*32 + k] (tools/trace_relay.py)
#ifdef CAVOID_TRACE
#define RELAY_STAMP(k)                                                                                         \
    do {                                                                                                       \
        if (lane0 == 0 && g_trace && t == (n_steps >> 1)) g_trace[wave * 32 + (k)] = (unsigned long long)clock64(); \
    } while (0)
// launch-level marks (not tied to a step): kernel entry / prologue done / first step posted / loop done / exit, per role (tools/trace_relay.py --launch)
#define RELAY_MARK(k)                                                                                          \
    do {                                                                                                       \
        if (lane0 == 0 && g_trace) g_trace[wave * 32 + (k)] = (unsigned long long)clock64();                   \
    } while (0)
#else
#define RELAY_STAMP(k) do { } while (0)
#define RELAY_MARK(k) do { } while (0)
#endif

// the counters are read and written through explicit LDS (address space 3) volatile pointers: ds_read_b32 / ds_write_b32, not
// the flat system-coherent accesses a generic volatile pointer compiles to
typedef __attribute__((address_space(3))) volatile int relay_lds_int;
__device__ __forceinline__ int relay_peek(const int *p) { return __builtin_amdgcn_readfirstlane(*(relay_lds_int *)p); }
__device__ __forceinline__ void relay_wait(const int *p, int v) {
    while (relay_peek(p) < v) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
// The observation wavefronts and the loader wait with a bound: a wait that does not end within ~1 s (2^24 polls; a step is
// microseconds) is a broken hand-over somewhere in the workgroup, and the wavefront traps -- the process gets a launch failure
// instead of a GPU that never answers again.  (D and P wait unbounded: the bound costs their loops 5 %.  They are covered all
// the same: whichever of them stalls, `fin` stops advancing, and the consumers' and the loader's bounded waits on it trap --
// kept honest by the fault-injection build, tests/test_gpu_relay_fault.py.)
constexpr int kRelayPollLimit = 1 << 24;
__device__ __forceinline__ void relay_wait_bounded(const int *p, int v) {
    int polls = 0;
    while (relay_peek(p) < v) {
        if (++polls > kRelayPollLimit) __builtin_trap();
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}
// the two waits of the loop-carried chain (D for P's verdict, P for D's next state) poll without sleeping
__device__ __forceinline__ void relay_spin(const int *p, int v) {
    while (relay_peek(p) < v) { }
    asm volatile("" ::: "memory");
}
// issue the LDS read of a counter NOW, look at the value LATER (relay_seen): the round trip runs under whatever the wavefront does in between
// -- D's waits that are almost always satisfied already (a consumer's ring slot, P's verdict in the D-bound regime) cost an issue slot
// instead of an exposed LDS latency on the loop-carried chain
__device__ __forceinline__ int relay_peek_issue(const int *p) { return *(relay_lds_int *)p; }
__device__ __forceinline__ int relay_seen(int v) { return __builtin_amdgcn_readfirstlane(v); }
typedef __attribute__((address_space(3))) volatile uint32_t relay_lds_u32;
typedef __attribute__((address_space(3))) volatile double relay_lds_f64;
__device__ __forceinline__ void relay_post(int *p, int v) {
    asm volatile("" ::: "memory");        // the data writes are issued before the counter (LDS keeps a wavefront's order)
    *(relay_lds_int *)p = v;
    asm volatile("" ::: "memory");
}

// sincos_bounded (cavoid_kernels.hpp) with its fifteen constants handed in: D pins them in scalar registers once, instead of
// re-materialising each as two moves per step (the step-loop units are built without MachineLICM).  Same operations, same
// order: bit-identical.
struct RelayTrig {
    double two_over_pi, pio2_hi, pio2_lo, s1, s2, s3, s4, s5, s6, c1, c2, c3, c4, c5, c6;
};
__device__ __forceinline__ RelayTrig relay_trig_constants() {
    RelayTrig k;
    k.two_over_pi = 0.63661977236758134308; k.pio2_hi = 1.57079632673412561417e+00; k.pio2_lo = 6.07710050650619224932e-11;
    k.s6 = 1.58969099521155010221e-10; k.s5 = -2.50507602534068634195e-08; k.s4 = 2.75573137070700676789e-06;
    k.s3 = -1.98412698298579493134e-04; k.s2 = 8.33333333332248946124e-03; k.s1 = -1.66666666666666324348e-01;
    k.c6 = -1.13596475577881948265e-11; k.c5 = 2.08757232129817482790e-09; k.c4 = -2.75573143513906633035e-07;
    k.c3 = 2.48015872894767294178e-05; k.c2 = -1.38888888888741095749e-03; k.c1 = 4.16666666666666019037e-02;
    asm volatile("" : "+s"(k.two_over_pi), "+s"(k.pio2_hi), "+s"(k.pio2_lo), "+s"(k.s1), "+s"(k.s2), "+s"(k.s3), "+s"(k.s4), "+s"(k.s5),
                 "+s"(k.s6), "+s"(k.c1), "+s"(k.c2), "+s"(k.c3), "+s"(k.c4), "+s"(k.c5), "+s"(k.c6));
    return k;
}
// wrap_angle with its two folds made side by side instead of one behind the other (a value >= pi folded down lands in [-pi, pi): the
// second fold never applies to it, and the first never to a value < -pi): the same result for every input, half the dependent length
__device__ __forceinline__ double relay_wrap(double a, uint32_t switches) {
    const double down = a - 2.0 * kPi, up = a + 2.0 * kPi;
    double r = a < -kPi ? up : a;
    r = a >= kPi ? down : r;
    if (CAVOID_RARE(__ballot(r >= kPi || r < -kPi) != 0ull)) {
        while (r >= kPi) r -= 2.0 * kPi;
        while (r < -kPi) r += 2.0 * kPi;
    }
    return wrap_closed_fixup(r, switches);
}
__device__ __forceinline__ void relay_sincos(const RelayTrig &q, double x, double *sn, double *cs) {
    const double k = rint(x * q.two_over_pi);
    double r = __builtin_fma(-k, q.pio2_hi, x);
    r = __builtin_fma(-k, q.pio2_lo, r);
    const double z = r * r;
    double ps = q.s6;
    ps = __builtin_fma(ps, z, q.s5);
    ps = __builtin_fma(ps, z, q.s4);
    ps = __builtin_fma(ps, z, q.s3);
    ps = __builtin_fma(ps, z, q.s2);
    ps = __builtin_fma(ps, z, q.s1);
    const double s = __builtin_fma(r * z, ps, r);
    double pc = q.c6;
    pc = __builtin_fma(pc, z, q.c5);
    pc = __builtin_fma(pc, z, q.c4);
    pc = __builtin_fma(pc, z, q.c3);
    pc = __builtin_fma(pc, z, q.c2);
    pc = __builtin_fma(pc, z, q.c1);
    const double c = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
    const int qd = (int)k;
    const double s_out = (qd & 1) ? c : s, c_out = (qd & 1) ? s : c;
    // (the quadrant's sign flips as an exclusive-or of the sign bit -- what a negation is -- instead of compare + select: two
    //  vector-compare -> select hand-overs off the chain)
    *sn = __longlong_as_double(__double_as_longlong(s_out) ^ ((long long)(qd & 2) << 62));
    *cs = __longlong_as_double(__double_as_longlong(c_out) ^ ((long long)((qd + 1) & 2) << 62));
}

// One agent's step up to (not including) the pair pass: E4 decode, scripted policies 1 / 2, E5 unicycle dynamics, goal test,
// time budget -- env_kernel's statements, value for value.
// What an agent's step re-derives from values that change once per episode: kept in registers by D, re-made for the lanes of a restarted world
struct RelayStatics {
    double pref, gx, gy;                                   // the float32 statics widened once
    float r_staged;                                        // what the consumers and P find as the radius: < 0 for an absent row
};
__device__ __forceinline__ RelayStatics relay_statics(const Agent &a, bool active) {
    RelayStatics k;
    k.pref = (double)a.pref; k.gx = (double)a.gx; k.gy = (double)a.gy;
    k.r_staged = (active && (a.flags & CAVOID_F_PRESENT)) ? a.radius : -1.0f;
    return k;
}
// tab_dh: the table's heading change ALREADY rounded through float32 when c.actions_fp32 (D converts the table once, at the prologue:
// (double)(float)x of a table entry is the same value whether made there or here)
__device__ __forceinline__ Agent relay_advance(const KCfg &c, const RelayTrig &trig, const Agent &in, const RelayStatics &ks, double tab_speed, double tab_dh,
                                               bool active, bool &moving) {
    Agent a = in;
    const uint32_t flags_in = a.flags;
    const bool present_in = active && (flags_in & CAVOID_F_PRESENT);
    const bool done_in = (flags_in & CAVOID_F_DONE_MASK) != 0u;
    const uint32_t pol = (flags_in >> CAVOID_F_POLICY_SHIFT) & CAVOID_F_POLICY_MASK;
    double a0 = ks.pref * tab_speed;                       // the action's table row, read one step ahead by the caller
    double a1 = tab_dh;
    if (CAVOID_RARE(__ballot(present_in && !done_in && pol != 0u) != 0ull)) {      // scripted agents in this tile
        if (pol == 1u) { a0 = 0.0; a1 = 0.0; }
        if (pol == 2u) {
            const Ego e0 = ego_frame_exact(c, a);
            a0 = ks.pref;
            a1 = -e0.heading_ego;
            if (c.actions_fp32) a1 = (double)(float)a1;
        }
    }
    if (c.actions_fp32) a0 = (double)(float)a0;
    moving = present_in && !done_in;
    double dh = a1;
    if (CAVOID_RARE(c.dynamics == CAVOID_DYN_UNICYCLE_MAX_TURN)) {
        const double rate = fmin(fmax(dh / c.dt, -c.cold->max_turn_rate), c.cold->max_turn_rate);
        dh = rate * c.dt;
    }
    const double nh = relay_wrap(dh + a.heading, c.switches);
    double sn, cs;
    if (CAVOID_RELAY_ABL & 4) { sn = nh; cs = 1.0; }
    else relay_sincos(trig, nh, &sn, &cs);
    const double npx = a.px + a0 * cs * c.dt, npy = a.py + a0 * sn * c.dt;
    const double nvx = a0 * cs, nvy = a0 * sn, nsp = a0;
    a.px = moving ? npx : a.px; a.py = moving ? npy : a.py; a.heading = moving ? nh : a.heading;
    a.vx = moving ? nvx : 0.0; a.vy = moving ? nvy : 0.0; a.speed = moving ? (float)nsp : 0.0f;
    // (env_kernel's branches as selects: same values, no divergent control flow on the chain)
    uint32_t latch = 0u;
    latch |= (flags_in & CAVOID_F_AT_GOAL) ? CAVOID_F_WAS_AT_GOAL : 0u;
    latch |= (flags_in & CAVOID_F_IN_COLL) ? CAVOID_F_WAS_IN_COLL : 0u;
    a.flags |= (present_in && done_in) ? latch : 0u;
    const double dx = a.px - ks.gx, dy = a.py - ks.gy;
    a.flags |= (moving && dx * dx + dy * dy <= c.near_goal_sq) ? CAVOID_F_AT_GOAL : 0u;
    const double t_next = a.t_rem - c.dt;
    a.t_rem = moving ? t_next : a.t_rem;
    a.flags |= (moving && c.timeout_enabled && t_next <= 0.0) ? CAVOID_F_RAN_OUT : 0u;
    return a;
}

__device__ __forceinline__ void relay_read_nxt(const RelayNxt &nb, int lane, Agent &nxt) {
    nxt.px = nb.px[lane]; nxt.py = nb.py[lane]; nxt.heading = nb.heading[lane]; nxt.t_rem = nb.t_rem[lane];
    nxt.gx = nb.gx[lane]; nxt.gy = nb.gy[lane]; nxt.radius = nb.radius[lane]; nxt.pref = nb.pref[lane];
    nxt.flags = nb.flags[lane];
    nxt.vx = nxt.vy = 0.0;
    nxt.speed = 0.0f;
}


// ---- the last step's observation by D, P and L together -----------------------------------------------------------------------------
// a hand-over of the three: the arriving wavefront's LDS writes are issued before its count (LDS keeps a wavefront's order), the
// others read behind the count; bounded like the consumers' waits (a wavefront that never arrives traps the launch)
__device__ __forceinline__ void relay_coop_arrive(int *p, int lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add((__attribute__((address_space(3))) int *)p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    int polls = 0;
    while (relay_peek(p) < kRelayCoopWaves) {
        if (++polls > kRelayPollLimit) __builtin_trap();
    }
    asm volatile("" ::: "memory");
}
// pair_pass_impl's statements (FEAT form) for the neighbours o = pw, pw + 3, .. of the lane's agent: the square-root chain, the sort key
// and the five features; key and "seen" bit into LDS for the other two wavefronts' ranking (cavoid_quad.hpp's quad_pair_round on the
// relay's hand-over arrays; the collision test is P's and long done)
template <int N, bool SW>
__device__ __forceinline__ void relay_pair_part(const KCfg &c, const Agent &a, const Ego &e, const bool present, const ArrayStage<N> &st, const int pw,
                                                const int lane, RelayCoop<N> &co, float (&gapf)[Others<N>::K], float (&feat)[Others<N>::K][kFeat]) {
    const double ri = (double)a.radius;
#pragma unroll
    for (int o = 0; o < N - 1; ++o) {
        if (o % kRelayCoopWaves != pw) continue;           // (wave-uniform)
        const OtherState q = st.other(o);
        const float rjf = q.r;
        const double rx = q.px - a.px, ry = q.py - a.py;
#if defined(CAVOID_DEV_ULP_FAULT) && CAVOID_DEV_ULP_FAULT == 1     /* the injected fault of pair_pass_impl, here too */
        const double d = sqrt_dist2((double)((float)rx * (float)rx) + ry * ry);
#else
        const double d = sqrt_dist2(rx * rx + ry * ry);
#endif
        const bool other = present && (rjf >= 0.0f);
        const bool seen = other && !(d > c.horizon);
        const double gap_o = d - ri - (double)rjf;
#if defined(CAVOID_DEV_ULP_FAULT) && CAVOID_DEV_ULP_FAULT == 3
        uint32_t hi = kKeyBias - (uint32_t)(int)rintf((float)gap_o * 100.0f);
#else
        uint32_t hi = kKeyBias - (uint32_t)(int)rint(gap_o * 100.0);
#endif
        uint32_t lo = orderable((float)(ry * e.tx - rx * e.ty));
        if (SW) {
            if (c.switches & kSwIndexTie) lo = (uint32_t)other_index(st.i, o, N);
            if (c.switches & kSwExactGap) { lo = 0u; hi = 0x7FFFFFFEu - (orderable((float)gap_o) >> 1); }
        }
        hi = seen ? hi : kKeySentinel + (uint32_t)o;
        co.key[o][lane] = ((uint64_t)hi << 32) | lo;
        co.bits[o][lane] = seen ? 2u : 0u;
        gapf[o] = (float)gap_o;
        neighbour_features(e, (float)e.prll_x, (float)e.prll_y, rx, ry, q, feat[o]);
    }
}
// the tile's rows out of LDS by the three wavefronts (flush_tile with tid / nthreads for lane / 64)
template <bool STREAM, bool WT = false>
__device__ __forceinline__ void relay_flush_part(const float *tile, float *dst, int n_floats, int tid, int nthreads) {
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
                if (k < n4) store16<STREAM, WT>(dst4 + k, v[u]);
            }
        }
    } else {
        for (int k = tid; k < n_floats; k += nthreads) dst[k] = tile[k];
    }
}


// Everything the three need they derive again from the kernel arguments and their thread ids, starting from two words D left in LDS at
// kernel entry (the argument segment's address, the tile): nothing of it stays live in scalar registers across the role loops above
// (written inline it cost D's and P's loops 23 and 16 v_readlane of spilled scalars per iteration; even the argument pointer kept live
// put one of D's sine / cosine constants into a spill lane).
struct RelayArgs { KCfg c; KState s; const PoolRec *pool; KIO io; };      // the kernel's argument segment
template <class T>
__device__ __forceinline__ void relay_load_args(T &dst, const __attribute__((address_space(4))) T *src) {      // word by word from the constant address space
    static_assert(sizeof(T) % 4 == 0, "whole words");
    const __attribute__((address_space(4))) uint32_t *w = (const __attribute__((address_space(4))) uint32_t *)src;
    uint32_t d[sizeof(T) / 4];
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 4); ++k) d[k] = w[k];
    __builtin_memcpy(&dst, d, sizeof(T));
}
template <int N>
__device__ __forceinline__ void relay_coop_last(unsigned char *smem) {
    unsigned char *sp = smem + lds_floats_block() * sizeof(float);
    RelaySeq *seq = reinterpret_cast<RelaySeq *>(sp); sp += sizeof(RelaySeq) + kRelayEvq * sizeof(unsigned long long);
    const unsigned long long kp = (unsigned long long)(uint32_t)relay_peek(&seq->kernarg[0]) | ((unsigned long long)(uint32_t)relay_peek(&seq->kernarg[1]) << 32);
    typedef const __attribute__((address_space(4))) RelayArgs *ArgP;
    const ArgP ka = (ArgP)kp;
    KCfg c;
    KIO io;
    relay_load_args(c, &ka->c);                             // (scalar loads of the fields the observation uses)
    relay_load_args(io, &ka->io);
    RelayTent *tents = reinterpret_cast<RelayTent *>(sp); sp += relay_ring<N>() * sizeof(RelayTent) + sizeof(RelayNxt);
    RelayRes *ress = reinterpret_cast<RelayRes *>(sp); sp += relay_ring<N>() * sizeof(RelayRes) + kRelayActRing * 64;
    RelayCoop<N> *coopb = reinterpret_cast<RelayCoop<N> *>(sp); sp += relay_coop_bytes<N>();
    float *tiles = reinterpret_cast<float *>(sp);
    const int lane0 = threadIdx.x & 63;
    const int NC = (blockDim.x >> 6) - 3;
    const int role = relay_role_of(threadIdx.x >> 6, NC);
    const int pw = role < 2 ? role : 2;                     // D 0, P 1, L 2
    const int ostride = io.obs_stride;
    const int tile_floats = (c.tile_rows * ostride + 3) & ~3;
    const int wpw = c.wpw, lanes_used = wpw * N;
    const int64_t wave = relay_peek(&seq->tile_id);
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
    constexpr int K = Others<N>::K;
    const int t = n_steps - 1;
    const int cid_l = t % NC;                          // the tile of the consumer whose turn the step would have been
    float *tile = tiles + (size_t)cid_l * tile_floats;
    RelayCoop<N> &co = *coopb;
    relay_wait_bounded(&seq->fin, t + 1);              // (D: its own post)
    const RelayTent &f = tents[t & (relay_ring<N>() - 1)];
    const RelayRes &v = ress[t & (relay_ring<N>() - 1)];
    Agent ao;
    ao.px = f.px[lane0]; ao.py = f.py[lane0]; ao.vx = f.vx[lane0]; ao.vy = f.vy[lane0];
    ao.heading = f.heading[lane0]; ao.t_rem = 0.0;
    ao.gx = f.gx[lane0]; ao.gy = f.gy[lane0]; ao.speed = 0.0f;
    ao.radius = f.r[lane0]; ao.pref = f.pref[lane0]; ao.flags = v.flags[lane0];
    const uint32_t ctl_c = v.ctl[lane0];
    const float rew_c = v.rew[lane0], done_c = (ctl_c & 1u) ? 1.0f : 0.0f;
    const bool present = active && (ao.flags & CAVOID_F_PRESENT);
    const Ego e = ego_frame_obs(c, ao);
    const ArrayStage<N> as{f.px, f.py, f.vx, f.vy, f.r, i0, base0};
    float gapf[K], feat[K][kFeat];
#pragma unroll
    for (int o = 0; o < K; ++o) {
        gapf[o] = 0.0f;
#pragma unroll
        for (int q = 0; q < kFeat; ++q) feat[o][q] = 0.0f;
    }
    if (c.switches != 0u) relay_pair_part<N, true>(c, ao, e, present, as, pw, lane0, co, gapf, feat);
    else relay_pair_part<N, false>(c, ao, e, present, as, pw, lane0, co, gapf, feat);
    relay_coop_arrive(&seq->coop[0], lane0);           // every neighbour's key is in LDS
    Key key[K];
    uint32_t valid = 0u;
    if (N == 1) key[0].set(kKeySentinel, 0u);
#pragma unroll
    for (int o = 0; o < N - 1; ++o) {
        key[o].v = co.key[o][lane0];
        valid |= (co.bits[o][lane0] & 2u) ? (1u << o) : 0u;
    }
    if (t - NC >= 0) relay_wait_bounded(&seq->cons[cid_l], t - NC + 1);   // that consumer's last rows have left the tile
    const int rows_active = (int)worlds_here * N;
    // the ranking (every one of the three makes it, from the same keys) and this wavefront's neighbours into the rows
    assemble_obs<N, false, true, ArrayStage<N>, NoHook, false, PartOthers>(c, ao, e, active, lane0, as, key, gapf, feat, valid, tile, nullptr,
                                                                           rows_active, ostride, false, 0.0f, 0.0f, wave, NoHook(), false, nullptr,
                                                                           PartOthers{pw, kRelayCoopWaves});
    if (pw == 0 && active && lane0 < rows_active) {    // the head of the row and the empty slots (assemble_obs's statements)
        const int M = c.max_other;
        const int m = __popc(valid);
        const int first = m > M ? m - M : 0;
        const int kept = m - first;
        float *row = tile + lane0 * ostride;
        row[0] = (present && (ao.flags & CAVOID_F_LEARNING)) ? 1.0f : 0.0f;
        row[1] = (float)kept;
        row[2] = present ? (float)e.dist : 0.0f;
        row[3] = present ? (float)e.heading_ego : 0.0f;
        row[4] = present ? ao.pref : 0.0f;
        row[5] = present ? ao.radius : 0.0f;
        for (int sl = kept; sl < M; ++sl) {
            float *z = row + 6 + 7 * sl;
#pragma unroll
            for (int q = 0; q < 7; ++q) z[q] = 0.0f;
        }
        if (packed) { row[c.width] = rew_c; row[c.width + 1] = done_c; }
    }
    relay_coop_arrive(&seq->coop[1], lane0);           // the rows are in the tile
    if (io.out_step_stride == 0)                       // one output buffer for every step: the last step's rows land last
        for (int o = 0; o < NC; ++o) relay_wait_bounded(&seq->cfin[o], 1);
    const int64_t slot_w = (int64_t)t * io.out_step_stride;
    if (worlds_here > 0) {
        float *dst = io.obs + (slot_w + w0) * N * ostride;
        // (whole 128-byte lines per tile, on a line boundary: the write-through form of the streaming store, like assemble_obs's flush)
        const bool lines = ((rows_active * ostride) & 31) == 0 && (reinterpret_cast<uintptr_t>(dst) & 127) == 0;
        if (io.out_step_stride != 0 && lines) relay_flush_part<true, true>(tile, dst, rows_active * ostride, pw * 64 + lane0, 64 * kRelayCoopWaves);
        else if (io.out_step_stride != 0) relay_flush_part<true>(tile, dst, rows_active * ostride, pw * 64 + lane0, 64 * kRelayCoopWaves);
        else relay_flush_part<false>(tile, dst, rows_active * ostride, pw * 64 + lane0, 64 * kRelayCoopWaves);
    }
    if (pw == 0 && active) {                           // the step's plain outputs
        if (!packed) {
            io.rew[slot_w * N + a_idx0] = rew_c;
            io.done[slot_w * N + a_idx0] = (ctl_c & 1u) ? 1 : 0;
        }
        if (i0 == 0) io.game_over[slot_w + w] = (ctl_c & 2u) ? 1 : 0;
    }
    if (pw == 0) RELAY_MARK(29);                           // D: its share of the last step's rows flushed
}

template <int N>
__global__ void __launch_bounds__(64 * (3 + kRelayMaxConsumers), 1) env_relay_kernel(const KCfg c, const KState s, const PoolRec *pool, const KIO io) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sp = smem;
    double *lds_tab = reinterpret_cast<double *>(sp); sp += lds_floats_block() * sizeof(float);
    RelaySeq *seq = reinterpret_cast<RelaySeq *>(sp); sp += sizeof(RelaySeq);
    unsigned long long *evq = reinterpret_cast<unsigned long long *>(sp); sp += kRelayEvq * sizeof(unsigned long long);
    RelayTent *tents = reinterpret_cast<RelayTent *>(sp); sp += relay_ring<N>() * sizeof(RelayTent);   // state of step t in slot t % ring
    RelayNxt *nbuf = reinterpret_cast<RelayNxt *>(sp); sp += sizeof(RelayNxt);
    RelayRes *ress = reinterpret_cast<RelayRes *>(sp); sp += relay_ring<N>() * sizeof(RelayRes);    // verdict of step t in slot t % ring
    unsigned char *actring = sp; sp += kRelayActRing * 64;
    RelayCoop<N> *coopb = reinterpret_cast<RelayCoop<N> *>(sp); sp += relay_coop_bytes<N>();   // (N < kRelayCoopFromN: nothing, never touched)
    float *tiles = reinterpret_cast<float *>(sp);

    const int NC = (blockDim.x >> 6) - 3;
    const int role = relay_role_of(threadIdx.x >> 6, NC);   // 0 D, 1 P, 2 .. 1+NC consumers, 2+NC L
    const int lane0 = threadIdx.x & 63;
    const int ostride = io.obs_stride;
    const int tile_floats = (c.tile_rows * ostride + 3) & ~3;
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
    constexpr bool kCoop = N >= kRelayCoopFromN;            // the last step's observation: D, P and L together (below the roles)
    const bool coop_role = role < 2 || role == 2 + NC;      // (the consumers are still at their own steps when the last one is settled)

    if (role == 0 && lane0 < (int)(sizeof(RelaySeq) / sizeof(int))) reinterpret_cast<int *>(seq)[lane0] = 0;
    if constexpr (kCoop) {
        if (role == 0 && lane0 == 0) {                      // (behind the zeroes: the same wavefront's LDS writes, in order)
            const unsigned long long kp = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
            seq->kernarg[0] = (int)(uint32_t)kp; seq->kernarg[1] = (int)(uint32_t)(kp >> 32);
            seq->tile_id = (int)blockIdx.x;
        }
    }

    if (role == 0) {
        // ================================================ D: state owner =====================================================
        __builtin_amdgcn_s_setprio(CAVOID_RELAY_PRIO_D);
        RELAY_MARK(20);                                    // D: kernel entry
        KCfg cd = c;                                        // this role's constants, pinned in scalar registers (see P)
        asm volatile("" : "+s"(cd.dt), "+s"(cd.near_goal_sq), "+s"(cd.actions_fp32), "+s"(cd.dynamics),
                     "+s"(cd.timeout_enabled), "+s"(cd.switches));
        const RelayTrig trig = relay_trig_constants();
        Agent a;
        a.px = a.py = a.heading = a.t_rem = a.vx = a.vy = 0.0;
        a.gx = a.gy = a.radius = a.pref = a.speed = 0.0f;
        a.flags = 0u;
        uint32_t episode = 0u;
        double tab_v = 0.0;
        if (lane0 < 2 * c.num_actions) tab_v = c.action_table[lane0];
        int act0 = 0;                                       // step 0's action: D's own load, the same trip to memory as the state (the
        if (active) {                                       // loader's first batch is for steps 1..7 as far as D is concerned)
            episode = s.episode[w];
            load_agent(s, a_idx0, a);
            act0 = io.actions[a_idx0];
        }
        if (cd.actions_fp32 && (lane0 & 1)) tab_v = (double)(float)tab_v;   // the heading-change column, rounded through float32 ONCE (relay_advance)
        lds_tab[lane0] = tab_v;
        act0 = act0 < 0 ? 0 : (act0 >= c.num_actions ? c.num_actions - 1 : act0);     // (clamped like E4 does, like the loader)
        const bool present_first = active && (a.flags & CAVOID_F_PRESENT);
        bool restarted_any = false, moved_any = false;
        int events = 0;
        bool T_moving;
        Agent T;
        // the first advance runs IN FRONT of the workgroup barrier -- the table is this wavefront's own LDS write, the state and the
        // action its own loads -- so the barrier (the loader's first batch in LDS) is waited for under it, not before it
        wave_lds_sync();
        RelayStatics ks = relay_statics(a, active);
        T = relay_advance(cd, trig, a, ks, lds_tab[2 * act0], lds_tab[2 * act0 + 1], active, T_moving);   // step 0 is not speculative
        __syncthreads();                                   // table, counters, the loader's first actions
        RELAY_MARK(21);                                    // D: state, table and the first actions are in; step 0 advanced
        // the table row of the NEXT step's action is read one iteration ahead (two dependent LDS trips off the chain)
        double tab_s = 0.0, tab_h = 0.0;
        int act_q = 0;                                      // ... and the action INDEX of the step after that, two iterations ahead
        if (n_steps > 1) {
            const int actn = (int)actring[64 + lane0];
            tab_s = lds_tab[2 * actn]; tab_h = lds_tab[2 * actn + 1];
            if (n_steps > 2) act_q = (int)actring[2 * 64 + lane0];
        }
        auto stage_out = [&](RelayTent &tn, const Agent &x, float r_staged, int lane) {   // r_staged: relay_statics of x's episode
            tn.px[lane] = x.px; tn.py[lane] = x.py; tn.r[lane] = r_staged; tn.flags[lane] = x.flags;
            tn.vx[lane] = x.vx; tn.vy[lane] = x.vy; tn.heading[lane] = x.heading;
            tn.gx[lane] = x.gx; tn.gy[lane] = x.gy; tn.pref[lane] = x.pref;
        };
        stage_out(tents[0], T, ks.r_staged, lane0);
        relay_post(&seq->spec, 1);
        relay_post(&seq->stage, 1);
        RELAY_MARK(22);                                    // D: step 0 posted
        int cslot = 0;                                      // (t + 1 - ring) mod NC, kept by counting
        // steps 0 .. n-2: each iteration posts the successor (step t+1) and then settles step t; the last step is settled below
        for (int t = 0; t + 1 < n_steps; ++t) {
            int lane = lane0;
            asm volatile("" : "+v"(lane));
            RELAY_STAMP(0);                                // D: iteration begins (stage t posted)
            // ---- the successor of step t as if nothing happens at t, while P works on step t: posted at once, P takes it as it
            //      is when its own verdict says so (no new collision, no restart in the tile) ---------------------------------
            RelayTent &tn = tents[(t + 1) & (relay_ring<N>() - 1)];
            const double tab_s1 = tab_s, tab_h1 = tab_h;    // action(t+1)'s row
            const int act2 = act_q;                        // action(t+2)'s index, read one iteration ago (0 past the end: a valid row, unused)
            int act3 = 0;                                   // action(t+3)'s, for the next iteration: the ring read lands under the advance
            if (t + 3 < n_steps) {
                // (the loader's first batch is 8 steps -- iterations 0..3 read steps 3..6 --, from iteration 4 on every eighth iteration makes
                //  sure of the next eight reads; the loader runs up to 48 steps ahead of `fin`)
                if (CAVOID_RARE((t & 7) == 4)) relay_wait(&seq->act, t + 12 < n_steps ? t + 12 : n_steps);
                act3 = (int)actring[((t + 3) & (kRelayActRing - 1)) * 64 + lane];
            }
            // Round 6: this iteration is the launch's period (P idles a third of it), and five LDS round trips sat bare on it -- the action ring,
            // the table row behind it, the consumer's ring-slot counter, P's verdict counter, the verdict itself.  The ring read runs one more
            // iteration ahead (above); the table row and the slot counter are ISSUED here, in front of the advance, and looked at behind it;
            // the verdict's words are read right behind its counter (LDS returns a wavefront's reads in order: a counter that says "posted"
            // vouches for the words read after it)
            const bool need_slot = t + 1 >= relay_ring<N>();    // slot free: the consumer of step t+1-ring is done with it
            const int cons_early = need_slot ? relay_peek_issue(&seq->cons[cslot]) : 0;
            const double tab_s2 = *(relay_lds_f64 *)&lds_tab[2 * act2], tab_h2 = *(relay_lds_f64 *)&lds_tab[2 * act2 + 1];
            bool mn;
            Agent Tn = relay_advance(cd, trig, T, ks, tab_s1, tab_h1, active, mn);
            if (need_slot) {
                if (CAVOID_RARE(relay_seen(cons_early) < t + 2 - relay_ring<N>())) relay_wait(&seq->cons[cslot], t + 2 - relay_ring<N>());
                cslot = cslot + 1 == NC ? 0 : cslot + 1;
            }
            const int res_early = relay_peek_issue(&seq->res);   // P's counter for step t: the trip runs under the staging writes
            stage_out(tn, Tn, ks.r_staged, lane);
            relay_post(&seq->spec, t + 2);
            RELAY_STAMP(1);                                // D: successor computed and posted
            // ---- P's verdict on step t: ONE word -- 2 * (steps posted) + "the last of them holds a surprise" (a new collision, a
            //      restart: P's own test).  P is at most one step ahead of this point, and only behind a step WITHOUT a surprise (else it
            //      waits for the corrected state): a count beyond t + 1 says "none at t".  The verdict's words are read only when there
            //      is one: without a surprise they are T's own flags, and nobody restarts ---------------------------------------------
            RelayRes *res = &ress[t & (relay_ring<N>() - 1)];
            int posted = relay_seen(res_early);
            if (CAVOID_RELAY_ABL & 8) posted = 2 * (t + 2);
            if (CAVOID_RARE((posted >> 1) < t + 1 - ((CAVOID_RELAY_ABL & 16) ? 1 : 0)))   // (P is ahead of D in the steady state: the early read has it)
                do posted = relay_peek(&seq->res); while ((posted >> 1) < t + 1 - ((CAVOID_RELAY_ABL & 16) ? 1 : 0));
            asm volatile("" ::: "memory");
            RELAY_STAMP(2);                                // D: verdict arrived
            moved_any = moved_any || T_moving;
            const bool surprise = !(CAVOID_RELAY_ABL & 24) && (posted >> 1) == t + 1 && (posted & 1) != 0;
            if (CAVOID_RARE(surprise)) {
                const uint32_t vflags = *(relay_lds_u32 *)&res->flags[lane], ctl = *(relay_lds_u32 *)&res->ctl[lane];
                const bool restart = (ctl & 2u) != 0u;
                const unsigned long long rmask = __ballot(restart);
                Agent S = T;                               // the committed state of step t
                S.flags = vflags;
                if (rmask != 0ull) {                       // some world of the tile starts a new episode
                    relay_wait(&seq->nxt, events + 1);     // the first records are in / every earlier restart's are re-armed
                    Agent nx;
                    relay_read_nxt(*nbuf, lane, nx);
                    bool mr;
                    const RelayStatics ksn = relay_statics(nx, active);
                    const Agent Tr = relay_advance(cd, trig, nx, ksn, tab_s1, tab_h1, active, mr);
                    if (restart) {
                        S = nx; episode += 1u; restarted_any = true; Tn = Tr; mn = mr; ks = ksn;
                        stage_out(tents[t & (relay_ring<N>() - 1)], nx, ksn.r_staged, lane);   // the consumers see step t's FINAL state: the new episode
                        res->flags[lane] = nx.flags;
                    }
                }
                const bool s_present = active && (S.flags & CAVOID_F_PRESENT);
                const bool s_done = (S.flags & CAVOID_F_DONE_MASK) != 0u;
                const bool frozen = s_present && s_done && !restart;   // env_kernel: present_in && done_in
                uint32_t fflags = S.flags;
                if (S.flags & CAVOID_F_AT_GOAL) fflags |= CAVOID_F_WAS_AT_GOAL;
                if (S.flags & CAVOID_F_IN_COLL) fflags |= CAVOID_F_WAS_IN_COLL;
                Tn.px = frozen ? S.px : Tn.px; Tn.py = frozen ? S.py : Tn.py; Tn.heading = frozen ? S.heading : Tn.heading;
                Tn.t_rem = frozen ? S.t_rem : Tn.t_rem;
                Tn.vx = frozen ? 0.0 : Tn.vx; Tn.vy = frozen ? 0.0 : Tn.vy; Tn.speed = frozen ? 0.0f : Tn.speed;
                Tn.flags = frozen ? fflags : Tn.flags;      // (gx, gy, radius, pref: per-episode constants, S's == Tn's)
                mn = frozen ? false : mn;
                stage_out(tn, Tn, ks.r_staged, lane);      // the posted successor was wrong for some lane
                relay_post(&seq->stage, t + 2);
                if (rmask != 0ull) {                       // tell the loader which lanes need their next pool record
                    relay_wait(&seq->nxt, events + 1 - (kRelayEvq - 1));
                    if (lane == 0) evq[events & (kRelayEvq - 1)] = rmask;
                    events += 1;
                    relay_post(&seq->ev, events);
                }
            }
            // (no surprise: the posted successor stands -- an agent that is done was frozen by relay_advance itself, its flags
            //  were known; nobody else changed)
            T = Tn;
            T_moving = mn;
            tab_s = tab_s2; tab_h = tab_h2;
            act_q = act3;
            relay_post(&seq->fin, t + 1);                  // slot t (state + verdict) is final: the consumers may take it
            RELAY_STAMP(3);                                // D: slot t final
        }
        // ---- the last step: nothing to post, only its committed state ---------------------------------------------------------
        Agent S = T;
        {
            const int t = n_steps - 1;
            relay_spin(&seq->res, 2 * (t + 1));
            RelayRes *res = &ress[t & (relay_ring<N>() - 1)];
            const uint32_t vflags = res->flags[lane0], ctl = res->ctl[lane0];
            moved_any = moved_any || T_moving;
            S.flags = vflags;
            const bool restart = (ctl & 2u) != 0u;
            if (__ballot(restart) != 0ull) {
                relay_wait(&seq->nxt, events + 1);
                Agent nx;
                relay_read_nxt(*nbuf, lane0, nx);
                if (restart) {
                    S = nx; episode += 1u; restarted_any = true;
                    stage_out(tents[t & (relay_ring<N>() - 1)], nx, relay_statics(nx, active).r_staged, lane0);
                    res->flags[lane0] = nx.flags;
                }
            }
            relay_post(&seq->fin, t + 1);
        }
        RELAY_MARK(23);                                    // D: last step settled
        relay_post(&seq->stage, n_steps + 1);               // (the loader may leave: no restart is waiting for a record any more)
        // ---- state write-back (once per launch) ------------------------------------------------------------------------------
        if (restarted_any) {
            store_agent(s, a_idx0, S);
            if (i0 == 0) s.episode[w] = episode;
        } else if (present_first) {
            if (moved_any) {
                s.px[a_idx0] = S.px; s.py[a_idx0] = S.py; s.heading[a_idx0] = S.heading; s.t_rem[a_idx0] = S.t_rem;
            }
            s.speed[a_idx0] = S.speed;
            s.flags[a_idx0] = S.flags;
        }
        RELAY_MARK(24);                                    // D: write-back issued
    } else if (role == 1) {
        // ================================================ P: pair pass, rewards, done ==========================================
        __builtin_amdgcn_s_setprio(CAVOID_RELAY_PRIO_P);
        // the constants of this role, pinned in scalar registers for the whole loop (left to itself the compiler re-loads
        // them from the kernel-argument segment inside the reward branches: seven scalar loads + waits on the loop-carried chain)
        KCfg cp = c;
        asm volatile("" : "+s"(cp.r_step), "+s"(cp.r_goal), "+s"(cp.r_coll), "+s"(cp.r_close), "+s"(cp.close_slope), "+s"(cp.close_range),
                     "+s"(cp.clip_lo), "+s"(cp.clip_hi), "+s"(cp.collision_dist), "+s"(cp.horizon), "+s"(cp.evaluate_mode));
        bool prev_surprise = false;
        __syncthreads();
        for (int t = 0; t < n_steps; ++t) {
            int lane = lane0, i = i0, base = base0;
            asm volatile("" : "+v"(lane), "+v"(i), "+v"(base));
#ifdef CAVOID_FAULT_RELAY
            // development build only (tests/test_gpu_relay_fault.py): the pair-pass wavefront of tile 0 walks away at step 5 -- D then
            // spins on a verdict that never comes; the bounded waits of the observation wavefronts and of the loader must turn
            // that into a trap (a failed launch), not into a GPU that never answers
            if (t == 5 && blockIdx.x == 0) return;
#endif
            RELAY_STAMP(8);                                // P: waiting for stage t
            // the speculative successor D posted while this wavefront worked on step t-1 is exact unless the verdict of t-1 found
            // a new collision or restarted a world: only then wait for D's corrected state
            if (prev_surprise) relay_spin(&seq->stage, t + 1);
            else relay_spin(&seq->spec, t + 1);
            RELAY_STAMP(9);                                // P: stage t arrived
            const RelayTent *tent = &tents[t & (relay_ring<N>() - 1)];
            Agent a;
            a.px = tent->px[lane]; a.py = tent->py[lane];
            a.radius = tent->r[lane];
            uint32_t flags = tent->flags[lane];
            const uint32_t flags_t = flags;
            const bool present = active && (flags & CAVOID_F_PRESENT);
            // ---- distances, collision test, nearest gap (pair_pass without the sort keys: the consumers make those) -------------
            constexpr int K = Others<N>::K;
            bool hit = false;
            double min_gap = INFINITY;
            const double ri = (double)a.radius;
            RELAY_STAMP(12);                               // P: own state read
#pragma unroll
            for (int o = 0; o < ((CAVOID_RELAY_ABL & 2) ? 0 : N - 1); ++o) {
                const int j = base + other_index(i, o, N);
                const float rjf = tent->r[j];
                const double rx = tent->px[j] - a.px, ry = tent->py[j] - a.py;
                const double d = sqrt_dist2(rx * rx + ry * ry);
                const bool other = present && (rjf >= 0.0f);
                const double gap_c = d - (ri + (double)rjf);          // pair_pass: the unordered-pair gap
                min_gap = other ? fmin(min_gap, gap_c) : min_gap;
                hit = hit || (other && gap_c <= cp.collision_dist);
            }
            RELAY_STAMP(13);                               // P: pair pass done
            // E7 / E8 (env_kernel's branches as selects: same values, no divergent control flow on the chain)
            const bool at_goal = (flags & CAVOID_F_AT_GOAL) != 0u;
            const bool may_collide = present && !at_goal && (flags & CAVOID_F_WAS_IN_COLL) == 0u;
            const bool collides = may_collide && hit;
            const bool close = may_collide && !hit && min_gap <= cp.close_range;
            double r = present ? cp.r_step : 0.0;
            r = (present && at_goal && (flags & CAVOID_F_WAS_AT_GOAL) == 0u) ? cp.r_goal : r;
            r = collides ? cp.r_coll : r;
            const double r_near = cp.r_close + cp.close_slope * min_gap;
            r = close ? r_near : r;
            flags |= collides ? CAVOID_F_IN_COLL : 0u;
            const double r_clip = fmin(fmax(r, cp.clip_lo), cp.clip_hi);
            r = present ? r_clip : r;
            const bool done = present ? (flags & CAVOID_F_DONE_MASK) != 0u : true;
            const unsigned long long running = __ballot(present && ((flags & CAVOID_F_LEARNING) || cp.evaluate_mode) && !done);
            const unsigned long long wmask = ((1ull << N) - 1ull) << base;
            const bool game_over = (running & wmask) == 0ull;
            const bool restart = active && game_over;
            const float rew_f = (float)r;
            RelayRes *res = &ress[t & (relay_ring<N>() - 1)];
            res->flags[lane] = flags;
            res->ctl[lane] = (done ? 1u : 0u) | (restart ? 2u : 0u);
            res->rew[lane] = rew_f;
            const bool new_coll = (flags & CAVOID_F_IN_COLL) != 0u && (flags_t & CAVOID_F_IN_COLL) == 0u;
            prev_surprise = !(CAVOID_RELAY_ABL & 24) && __ballot(new_coll || restart) != 0ull;
            relay_post(&seq->res, 2 * (t + 1) + (prev_surprise ? 1 : 0));    // (D reads the verdict's words only behind a surprise)
            RELAY_STAMP(10);                               // P: verdict posted (the plain outputs go out with the consumer's rows)
        }
    } else if (role == 2 + NC) {
        // ================================================ L: actions and pool records ==========================================
        __builtin_amdgcn_s_setprio(CAVOID_RELAY_PRIO_L);
        RELAY_MARK(25);                                    // L: kernel entry
        uint32_t ep = 0u;
        if (active) ep = s.episode[w];
        int loaded = 0;
        const int64_t a_safe = active ? a_idx0 : 0;        // (idle lanes read lane 0's actions: no branch around the loads)
        auto load_actions = [&](int upto) {                 // steps [loaded, upto) -> ring, clamped like E4 does: the loads of up to
            while (loaded < upto) {                         // eight steps are in flight together, then the eight bytes are written
                int v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int sidx = loaded + q < upto ? loaded + q : upto - 1;
                    v[q] = io.actions[(int64_t)sidx * io.action_stride + a_safe];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    int x = v[q] < 0 ? 0 : (v[q] >= c.num_actions ? c.num_actions - 1 : v[q]);
                    if (loaded + q < upto) actring[((loaded + q) & (kRelayActRing - 1)) * 64 + lane0] = (unsigned char)x;
                }
                loaded = loaded + 8 < upto ? loaded + 8 : upto;
            }
        };
        load_actions(n_steps < 8 ? n_steps : 8);            // enough for D to start; the rest follows while the loop runs
        RELAY_MARK(26);                                    // L: first action batch in LDS
        __syncthreads();
        relay_post(&seq->act, loaded);
        // the next scenario-pool record of every lane: two dependent trips to memory (episode -> pool index -> record) that nobody
        // needs before the first restart, so they are taken AFTER the roles have started (D waits for nxt >= 1 when a world ends)
        {
            Agent r0;
            absent_agent(r0);
            if (active) load_pool(pool, (int64_t)pool_index(c, (uint32_t)(c.world_offset + w), ep + 1u) * N + i0, r0);
            RelayNxt &nb = *nbuf;
            nb.px[lane0] = r0.px; nb.py[lane0] = r0.py; nb.heading[lane0] = r0.heading; nb.t_rem[lane0] = r0.t_rem;
            nb.gx[lane0] = r0.gx; nb.gy[lane0] = r0.gy; nb.radius[lane0] = r0.radius; nb.pref[lane0] = r0.pref;
            nb.flags[lane0] = r0.flags;
            relay_post(&seq->nxt, 1);
            RELAY_MARK(27);                                // L: first pool records posted
        }
        int served = 0, idle_polls = 0;
        while (true) {
            const int ev = relay_peek(&seq->ev), fin = relay_peek(&seq->fin);
            if (served < ev) {
                asm volatile("" ::: "memory");
                const unsigned long long mask = evq[served & (kRelayEvq - 1)];
                if ((mask >> lane0) & 1ull) {
                    ep += 1u;
                    Agent r0;
                    load_pool(pool, (int64_t)pool_index(c, (uint32_t)(c.world_offset + w), ep + 1u) * N + i0, r0);
                    RelayNxt &nb = *nbuf;
                    nb.px[lane0] = r0.px; nb.py[lane0] = r0.py; nb.heading[lane0] = r0.heading; nb.t_rem[lane0] = r0.t_rem;
                    nb.gx[lane0] = r0.gx; nb.gy[lane0] = r0.gy; nb.radius[lane0] = r0.radius; nb.pref[lane0] = r0.pref;
                    nb.flags[lane0] = r0.flags;
                }
                served += 1;
                idle_polls = 0;
                relay_post(&seq->nxt, served + 1);
                continue;
            }
            if (loaded < n_steps && loaded < fin + kRelayActAhead) {
                int upto = fin + kRelayActAhead;
                if (upto > loaded + 8) upto = loaded + 8;
                if (upto > n_steps) upto = n_steps;
                load_actions(upto);
                idle_polls = 0;
                relay_post(&seq->act, loaded);
                continue;
            }
            if (fin >= n_steps && relay_peek(&seq->stage) > n_steps) break;
            if (++idle_polls > kRelayPollLimit) __builtin_trap();
            __builtin_amdgcn_s_sleep(2);
        }
    } else {
        // ================================================ C: observation of every NC-th step ===================================
        __builtin_amdgcn_s_setprio(CAVOID_RELAY_PRIO_C);
        KCfg cc = c;                                        // this role's switch word, pinned like the other roles' constants
        asm volatile("" : "+s"(cc.switches));
        const int cid = role - 2;
        float *tile = tiles + (size_t)cid * tile_floats;
        __syncthreads();
        const int n_mine = kCoop ? n_steps - 1 : n_steps;   // (kCoop: the last step's observation is D's, P's and L's)
        for (int t = cid; t < n_mine; t += NC) {
            int lane = lane0, i = i0, base = base0;
            asm volatile("" : "+v"(lane), "+v"(i), "+v"(base));
            RELAY_STAMP(16);                               // C: waiting for final state t
            relay_wait_bounded(&seq->fin, t + 1);
            RELAY_STAMP(17);                               // C: arrived
            if (CAVOID_RELAY_ABL & 1) { relay_post(&seq->cons[cid], t + 1); continue; }
            const RelayTent &f = tents[t & (relay_ring<N>() - 1)];
            const RelayRes &v = ress[t & (relay_ring<N>() - 1)];
            Agent ao;
            ao.px = f.px[lane]; ao.py = f.py[lane]; ao.vx = f.vx[lane]; ao.vy = f.vy[lane];
            ao.heading = f.heading[lane]; ao.t_rem = 0.0;
            ao.gx = f.gx[lane]; ao.gy = f.gy[lane]; ao.speed = 0.0f;
            ao.radius = f.r[lane]; ao.pref = f.pref[lane]; ao.flags = v.flags[lane];
            const uint32_t ctl_c = v.ctl[lane];
            const float rew_c = v.rew[lane], done_c = (ctl_c & 1u) ? 1.0f : 0.0f;
            const bool present = active && (ao.flags & CAVOID_F_PRESENT);
            const Ego e = ego_frame_obs(cc, ao);
            Key key[Others<N>::K];
            float gapf[Others<N>::K];
            uint32_t valid;
            bool hit;
            double min_gap;
            const ArrayStage<N> as{f.px, f.py, f.vx, f.vy, f.r, i, base};
            float feat[Others<N>::K][kFeat];
            pair_pass<N, false, true>(cc, ao, e, present, as, key, gapf, feat, valid, hit, min_gap);
            RELAY_STAMP(18);                               // C: ego frame + keys
            const bool last = t == n_steps - 1 && io.out_step_stride == 0;   // (with per-step slots no two steps share an address)
            const int64_t slot_w = (int64_t)t * io.out_step_stride;
            auto order_last = [&]() {                      // the last step's rows go out after every earlier step's have landed
                if (last)
                    for (int o = 0; o < NC; ++o)
                        if (o != cid) relay_wait_bounded(&seq->cfin[o], 1);
            };
            assemble_obs<N, false, true, ArrayStage<N>, decltype(order_last)>(cc, ao, e, active, lane, as, key, gapf, feat, valid, tile,
                                         io.obs + (slot_w + w0) * N * ostride, (int)worlds_here * N, ostride, packed, rew_c, done_c, wave,
                                         order_last, io.out_step_stride != 0);
            if (active) {                                  // the step's plain outputs (behind order_last, like the rows)
                if (!packed) {
                    io.rew[slot_w * N + a_idx0] = rew_c;
                    io.done[slot_w * N + a_idx0] = (ctl_c & 1u) ? 1 : 0;
                }
                if (i0 == 0) io.game_over[slot_w + w] = (ctl_c & 2u) ? 1 : 0;
            }
            relay_post(&seq->cons[cid], t + 1);
            RELAY_STAMP(19);                               // C: rows flushed
            if (t == 0) RELAY_MARK(28);                    // C: step 0's rows flushed
            if (t == n_steps - 1) RELAY_MARK(29);          // C: the last step's rows flushed
        }
        __builtin_amdgcn_s_waitcnt(0);                     // every store of this consumer has completed
        if (cid == 0) RELAY_MARK(30);                      // C0: its stores have completed
        relay_post(&seq->cfin[cid], 1);
    }
    // ================================================ D, P, L: the LAST step's observation, together =======================================
    if constexpr (kCoop) {
        if (coop_role) relay_coop_last<N>(smem);
    }
}

}  // namespace cavoid
