// cavoid_capi.hip -- the C ABI of include/cavoid.h over the gfx950 kernels of cavoid_kernels.hpp.
// Host side only: argument checking, the library-owned world buffer, launch geometry.
// There is deliberately no CPU path here: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

#include "cavoid.h"
#include "cavoid_host.hpp"
#include "cavoid_kernels.hpp"
#include "cavoid_launch.hpp"

using namespace cavoid;

thread_local int g_last_hip_error = 0;

extern "C" int cavoid_abi_version(void) { return CAVOID_ABI_VERSION; }

extern "C" int cavoid_last_hip_error(void) { return g_last_hip_error; }

extern "C" const char *cavoid_strerror(int code) {
    switch (code) {
        case CAVOID_OK: return "ok";
        case CAVOID_EINVAL: return "invalid argument or configuration";
        case CAVOID_ENOMEM: return "device memory allocation failed";
        case CAVOID_EHIP: return "HIP runtime call failed (see cavoid_last_hip_error)";
        case CAVOID_EUNSUPPORTED: return "max_agents outside the compiled range [1,16]";
        case CAVOID_ENODEVICE: return "no usable HIP device";
        case CAVOID_ECOMM: return "RCCL call failed (see cavoid_last_comm_error)";
        default: return "unknown error";
    }
}

extern "C" int cavoid_default_actions(double (*table)[2], int32_t *num_actions) {
    if (!table || !num_actions) return CAVOID_EINVAL;
    // E4: 5 headings at full speed (pi/12 apart), 3 at half speed, 3 at zero speed (pi/6 apart);
    // evidence Server.py:51-52, Config.py:79, Regression.py:157-160
    const double frac[3] = {1.0, 0.5, 0.0}, step[3] = {kPi / 12, kPi / 6, kPi / 6};
    const int count[3] = {5, 3, 3};
    int r = 0;
    for (int g = 0; g < 3; ++g)
        for (int k = 0; k < count[g]; ++k, ++r) {
            table[r][0] = frac[g];
            table[r][1] = -kPi / 6 + k * step[g];
        }
    *num_actions = r;
    return CAVOID_OK;
}

extern "C" int cavoid_default_cfg(cavoid_cfg *c, int32_t max_agents, int32_t max_other) {
    if (!c || max_agents < 1 || max_other < 0) return CAVOID_EINVAL;
    std::memset(c, 0, sizeof(*c));
    c->struct_size = (uint32_t)sizeof(cavoid_cfg);
    c->abi_version = CAVOID_ABI_VERSION;
    c->max_agents = max_agents;
    c->max_other = max_other;
    c->sort_method = CAVOID_SORT_CLOSEST_LAST;
    c->dynamics = CAVOID_DYN_UNICYCLE;
    c->actions_fp32 = 1;
    c->timeout_enabled = 1;
    c->time_budget_from_goal_edge = 1;
    c->wrap_closed_end = 0;             /* U2 */
    c->done_agents_collide = 1;         /* U4 */
    c->sort_round_gap = 1;              /* U7a */
    c->sort_tie_lateral = 1;            /* U7b */
    c->dt = 0.2;
    c->near_goal_threshold = 0.2;
    c->max_time_ratio = 2.0;
    c->collision_dist = 0.0;
    c->getting_close_range = 0.2;
    c->reward_at_goal = 1.0;
    c->reward_collision = -0.25;
    c->reward_getting_close = -0.1;
    c->reward_time_step = 0.0;
    c->close_penalty_slope = 0.5;                          // U5: the published sign (arXiv:1805.01956), which the reference's recorded scores favour too
    c->reward_clip_lo = -0.25;
    c->reward_clip_hi = 1.0;
    c->sensing_horizon = INFINITY;
    c->max_turn_rate = 3.0;
    cavoid_default_actions(c->actions, &c->num_actions);
    c->gen_min_agents = max_agents;
    c->gen_max_agents = max_agents;
    c->gen_nonlearning_fraction = 0.0;
    c->gen_static_fraction = 0.5;
    c->gen_goal_jitter = 0.5;
    c->gen_angle_jitter = 0.25;
    c->gen_pool_size = 65536;
    c->gen_mode = 0;
    c->gen_box_large_from = 5;
    c->gen_pool_epoch = 0;
    c->rvo_enabled = 0;
    c->gen_rvo_fraction = 0.0;
    c->gen_frozen_fraction = 0.0;
    c->gen_box_small[0] = 4.0; c->gen_box_small[1] = 5.0;
    c->gen_box_large[0] = 6.0; c->gen_box_large[1] = 8.0;
    c->gen_min_trip = 1.0;
    c->rvo_time_horizon = 5.0;          /* RVO_TIME_HORIZON  run-ws/config.yaml:237-239 */
    c->rvo_collab_coeff = 0.5;          /* RVO_COLLAB_COEFF  run-ws/config.yaml:234-236 */
    c->rvo_radius_scale = 1.05;
    c->rvo_max_delta_heading = kPi / 6;
    return CAVOID_OK;
}

static int validate(const cavoid_cfg *c) {
    if (!c || c->struct_size != sizeof(cavoid_cfg) || c->abi_version != CAVOID_ABI_VERSION) return CAVOID_EINVAL;
    if (c->max_agents < 1 || c->max_agents > CAVOID_MAX_AGENTS) return CAVOID_EUNSUPPORTED;
    if (c->max_other < 0 || c->max_other > 64) return CAVOID_EINVAL;
    if (c->num_actions < 1 || c->num_actions > CAVOID_MAX_ACTIONS) return CAVOID_EINVAL;
    if (c->sort_method < 0 || c->sort_method > 2 || c->dynamics < 0 || c->dynamics > 2) return CAVOID_EINVAL;
    if (!(c->dt > 0.0)) return CAVOID_EINVAL;
    if (c->gen_min_agents < 1 || c->gen_max_agents > c->max_agents || c->gen_min_agents > c->gen_max_agents) return CAVOID_EINVAL;
    if (c->gen_pool_size < 0 || c->gen_pool_size > (1 << 24)) return CAVOID_EINVAL;
    if (c->gen_lookahead != 0 && (c->gen_pool_size != 0 || c->gen_lookahead < 2 || c->gen_lookahead > 4096 || (c->gen_lookahead & (c->gen_lookahead - 1))))
        return CAVOID_EINVAL;                                                    /* look-ahead: no hashed pool beside it, R a power of two */
    if (c->gen_mode < 0 || c->gen_mode > 1) return CAVOID_EINVAL;
    if (c->gen_mode == 1 && !(c->gen_box_small[0] > 0.0 && c->gen_box_small[1] >= c->gen_box_small[0] &&
                              c->gen_box_large[0] > 0.0 && c->gen_box_large[1] >= c->gen_box_large[0] && c->gen_min_trip >= 0.0))
        return CAVOID_EINVAL;
    if (c->gen_rvo_fraction < 0.0 || c->gen_rvo_fraction > 1.0) return CAVOID_EINVAL;
    if (c->gen_frozen_fraction < 0.0 || c->gen_frozen_fraction > 1.0) return CAVOID_EINVAL;
    if (c->gen_rvo_fraction > 0.0 && !c->rvo_enabled) return CAVOID_EINVAL;    /* the generator would create agents the step cannot drive */
    if (c->rvo_enabled && !(c->rvo_time_horizon > 0.0 && c->rvo_radius_scale > 0.0)) return CAVOID_EINVAL;
    return CAVOID_OK;
}


// one device allocation holding the SoA world buffer of `worlds` worlds (+ `extra` trailing bytes)
static int alloc_state(size_t worlds, size_t agents, size_t extra, void **slab, KState *st, size_t *extra_off) {
    const size_t A = worlds * agents;
    size_t off = 0, o_f64 = off;
    off = align_up(off + 4 * A * sizeof(double), 256);
    const size_t o_f32 = off;
    off = align_up(off + 5 * A * sizeof(float), 256);
    const size_t o_flags = off;
    off = align_up(off + A * sizeof(uint32_t), 256);
    const size_t o_ep = off;
    off = align_up(off + worlds * sizeof(uint32_t), 256);
    *extra_off = off;
    off = align_up(off + extra, 256);
    if (hipMalloc(slab, off) != hipSuccess) return CAVOID_ENOMEM;
    if (hipMemset(*slab, 0, off) != hipSuccess) return CAVOID_EHIP;
    unsigned char *b = static_cast<unsigned char *>(*slab);
    double *f64 = reinterpret_cast<double *>(b + o_f64);
    float *f32 = reinterpret_cast<float *>(b + o_f32);
    st->px = f64; st->py = f64 + A; st->heading = f64 + 2 * A; st->t_rem = f64 + 3 * A;
    st->gx = f32; st->gy = f32 + A; st->radius = f32 + 2 * A; st->pref = f32 + 3 * A; st->speed = f32 + 4 * A;
    st->flags = reinterpret_cast<uint32_t *>(b + o_flags);
    st->episode = reinterpret_cast<uint32_t *>(b + o_ep);
    return CAVOID_OK;
}

static int grid_for(const cavoid_env *e, int64_t worlds) {
    const int wpw = e->k.wpw;
    const int64_t waves = (worlds + wpw - 1) / wpw;
    return (int)((waves + e->waves_per_block - 1) / e->waves_per_block);
}

extern "C" int cavoid_create(const cavoid_cfg *cfg, int64_t num_worlds, int64_t world_offset, int device, cavoid_env **out) {
    if (!out) return CAVOID_EINVAL;
    *out = nullptr;
    int rc = validate(cfg);
    if (rc != CAVOID_OK) return rc;
    if (num_worlds < 1 || world_offset < 0 || num_worlds > (int64_t)1 << 31) return CAVOID_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return CAVOID_ENODEVICE;
    HIP_TRY(hipSetDevice(device));

    cavoid_env *e = new (std::nothrow) cavoid_env();
    if (!e) return CAVOID_ENOMEM;
    e->device = device;
    e->W = num_worlds;
    e->A = num_worlds * cfg->max_agents;
    e->world_offset = world_offset;
    e->cfg = *cfg;

    const size_t W = (size_t)e->W;
    size_t o_act = 0;
    constexpr size_t kActBytes = CAVOID_MAX_ACTIONS * 2 * sizeof(double);                  // the action table, then the cold constants
    rc = alloc_state(W, (size_t)cfg->max_agents, kActBytes + sizeof(KCold), &e->slab, &e->st, &o_act);
    if (rc == CAVOID_OK && cfg->gen_pool_size > 0) {
        e->pool_size = cfg->gen_pool_size;
        const size_t recs = (size_t)e->pool_size * (size_t)cfg->max_agents * sizeof(PoolRec);
        if (hipMalloc(&e->pool_slab, recs + (size_t)e->pool_size * sizeof(uint32_t)) != hipSuccess) rc = CAVOID_ENOMEM;
        else {
            e->pool = static_cast<PoolRec *>(e->pool_slab);
            e->pool_episode = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(e->pool_slab) + recs);
        }
    }
    if (rc == CAVOID_OK && cfg->gen_lookahead > 0) {                                  // scenario look-ahead: W rings of R records + filled_hi [W]
        e->ahead_R = cfg->gen_lookahead;
        e->pool_size = (int64_t)W * e->ahead_R;                                          // (> 0: the restart paths gather, as from a pool)
        const size_t recs = (size_t)e->pool_size * (size_t)cfg->max_agents * sizeof(PoolRec);
        if (hipMalloc(&e->pool_slab, recs + 2 * W * sizeof(uint32_t)) != hipSuccess) rc = CAVOID_ENOMEM;
        else {
            e->pool = static_cast<PoolRec *>(e->pool_slab);
            e->ahead_hi[0] = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(e->pool_slab) + recs);
            e->ahead_hi[1] = e->ahead_hi[0] + W;
            if (hipMemset(e->ahead_hi[0], 0xFF, 2 * W * sizeof(uint32_t)) != hipSuccess) rc = CAVOID_EHIP;
        }
    }
    if (rc != CAVOID_OK) { g_last_hip_error = (int)hipGetLastError(); cavoid_destroy(e); return rc; }
    e->d_actions = reinterpret_cast<double *>(static_cast<unsigned char *>(e->slab) + o_act);
    if (hipMemset(e->st.episode, 0xFF, W * sizeof(uint32_t)) != hipSuccess ||
        hipMemcpy(e->d_actions, cfg->actions, sizeof(cfg->actions), hipMemcpyHostToDevice) != hipSuccess ||
        hipEventCreate(&e->ev0) != hipSuccess || hipEventCreate(&e->ev1) != hipSuccess) {
        g_last_hip_error = (int)hipGetLastError();
        cavoid_destroy(e);
        return CAVOID_EHIP;
    }

    KCold cold{};
    cold.budget_offset = cfg->time_budget_from_goal_edge ? cfg->near_goal_threshold : 0.0;
    cold.max_time_ratio = cfg->max_time_ratio;
    cold.max_turn_rate = cfg->max_turn_rate;
    cold.gen_nonlearning = cfg->gen_nonlearning_fraction; cold.gen_static = cfg->gen_static_fraction;
    cold.gen_goal_jitter = cfg->gen_goal_jitter; cold.gen_angle_jitter = cfg->gen_angle_jitter;
    cold.gen_rvo = cfg->gen_rvo_fraction; cold.gen_min_trip = cfg->gen_min_trip; cold.gen_frozen = cfg->gen_frozen_fraction;
    cold.gen_box_small_lo = cfg->gen_box_small[0]; cold.gen_box_small_hi = cfg->gen_box_small[1];
    cold.gen_box_large_lo = cfg->gen_box_large[0]; cold.gen_box_large_hi = cfg->gen_box_large[1];
    cold.gen_box_large_from = cfg->gen_box_large_from;
    cold.rvo_inv_horizon = cfg->rvo_enabled ? 1.0 / cfg->rvo_time_horizon : 0.0;
    cold.rvo_collab = cfg->rvo_collab_coeff; cold.rvo_radius_scale = cfg->rvo_radius_scale; cold.rvo_max_dh = cfg->rvo_max_delta_heading;
    cold.gen_min_agents = cfg->gen_min_agents; cold.gen_max_agents = cfg->gen_max_agents;
    KCold *d_cold = reinterpret_cast<KCold *>(static_cast<unsigned char *>(e->slab) + o_act + kActBytes);
    if (hipMemcpy(d_cold, &cold, sizeof(cold), hipMemcpyHostToDevice) != hipSuccess) {
        g_last_hip_error = (int)hipGetLastError();
        cavoid_destroy(e);
        return CAVOID_EHIP;
    }

    KCfg &k = e->k;
    k.cold = d_cold;
    k.dt = cfg->dt;
    k.near_goal_sq = cfg->near_goal_threshold * cfg->near_goal_threshold;
    k.collision_dist = cfg->collision_dist;
    k.close_range = cfg->getting_close_range;
    k.r_goal = cfg->reward_at_goal; k.r_coll = cfg->reward_collision; k.r_close = cfg->reward_getting_close;
    k.r_step = cfg->reward_time_step; k.close_slope = cfg->close_penalty_slope;
    k.clip_lo = cfg->reward_clip_lo; k.clip_hi = cfg->reward_clip_hi;
    k.horizon = cfg->sensing_horizon;
    k.switches = (cfg->done_agents_collide ? 0u : kSwSkipDonePairs) | (cfg->sort_round_gap ? 0u : kSwExactGap) |
                 (cfg->sort_tie_lateral ? 0u : kSwIndexTie) | (cfg->wrap_closed_end ? kSwWrapClosed : 0u);
    k.gen_mode = cfg->gen_mode; k.pool_epoch = cfg->gen_pool_epoch;
    k.rvo_enabled = cfg->rvo_enabled ? 1 : 0;
    k.max_other = cfg->max_other; k.width = 6 + 7 * cfg->max_other;
    k.sort_method = cfg->sort_method; k.dynamics = cfg->dynamics; k.actions_fp32 = cfg->actions_fp32;
    k.timeout_enabled = cfg->timeout_enabled; k.num_actions = cfg->num_actions;
    k.pool_size = cfg->gen_lookahead > 0 ? (int32_t)(e->pool_size > 0x7fffffffLL ? 0x7fffffffLL : e->pool_size) : cfg->gen_pool_size;   // (look-ahead: only its sign is used)
    k.ahead = cfg->gen_lookahead;
    k.evaluate_mode = cfg->evaluate_mode ? 1 : 0;
    k.stream_obs = (double)num_worlds * cfg->max_agents * (6 + 7 * cfg->max_other) * sizeof(float) > 16.0 * 1048576.0 ? 1 : 0;   // (measured: 10 x 262144 one step 213 -> 193 us, 4 x 65536 25.2 -> 24.5 us)
    if (const char *ov = std::getenv("CAVOID_STREAM_OBS")) k.stream_obs = std::atoi(ov) != 0;
    k.seed_lo = 0; k.seed_hi = 0;
    k.num_worlds = num_worlds; k.world_offset = world_offset;
    k.action_table = e->d_actions;

    // launch geometry: a wavefront owns floor(64/N) whole worlds.  (Measured at 4 x 8192, 32-step launches: 16 / 8 / 4 /
    // 2 / 1 worlds per wavefront -> 3.07 / 3.09 / 3.56 / 6.05 / 10.4 us per step: a wavefront's float64 chain costs the
    // same issue slots whether 16 or 64 lanes are live, so emptier wavefronts only multiply the issue work.)
    // CAVOID_WPW overrides it for such A/B runs.
    const int N = cfg->max_agents, wpw_max = 64 / N;
    int wpw = wpw_max;
    if (const char *ov = std::getenv("CAVOID_WPW")) { int v = std::atoi(ov); if (v >= 1 && v <= wpw_max) wpw = v; }
    k.wpw = wpw;
    const int lanes = wpw * N;
    // latency mode (at most ~2 full wavefronts per SIMD): multi-step launches keep every lane's NEXT pool record in
    // registers (64 B read per agent per LAUNCH and per restart), so a restart never costs a dependent trip to
    // memory.  Single-step launches gather on demand unless CAVOID_PREFETCH_POOL=1 (then +64 B per agent-step).
    e->latency_mode = (num_worlds * cfg->max_agents <= 64 * 2048 && (cfg->gen_pool_size > 0 || cfg->gen_lookahead > 0)) ? 1 : 0;
    if (const char *ov = std::getenv("CAVOID_PREFETCH_POOL")) {
        const int v = std::atoi(ov);
        e->latency_mode = (v != 0 && (cfg->gen_pool_size > 0 || cfg->gen_lookahead > 0)) ? 1 : 0;
        e->prefetch_single = e->latency_mode;
    }
    k.prefetch_pool = e->latency_mode;
    if (const char *ov = std::getenv("CAVOID_PIPELINE")) e->pipeline = std::atoi(ov);
    if (const char *ov = std::getenv("CAVOID_QUAD")) e->quad = std::atoi(ov);
    if (const char *ov = std::getenv("CAVOID_RELAY_CONSUMERS")) {
        const int v = std::atoi(ov);
        if (v >= 1 && v <= cavoid::kRelayMaxConsumers) e->relay_consumers = v;
    }
    // obs tile (rows of width + 2 floats: the packed record is the widest row): the wavefront's rows in ONE pass when
    // the batch is latency bound or when they fit ~9 KiB; else several passes of a multiple of 4 rows, so that the LDS
    // footprint (and the wavefronts resident per CU) does not scale with N*(1+D)   [N=10: +7 % at saturation]
    // ORCA scratch: two sets of N-1 lines (4 doubles) per lane, only when RVO agents may exist
    k.rvo_lds_floats = cfg->rvo_enabled ? 64 * 2 * (N - 1) * 4 * 2 : 0;
    // the tile region also parks the sort keys and gaps of the one-step / reset / observe instantiations (kPark).  With
    // rvo_enabled the STEP launches take the RVO instantiations (no parking, the ORCA scratch is theirs); reset / observe still
    // run the parking instantiations, whose parked floats may then run past a small tile into the wavefront's ORCA scratch --
    // by construction: that scratch (1024 (N-1) floats) is idle in those modes and always larger than the 192 (N-1) parked.
    // ... then the float64 velocities of the time-to-impact order (256 floats), and at least the field-major scratch of the ORCA
    // policy / the box generator
    k.park_floats = ((N >= kParkFromN && !cfg->rvo_enabled) ? 3 * (N - 1) * 64 : 0) + 256;
    if (k.park_floats < lds_floats_scratch()) k.park_floats = lds_floats_scratch();
    static_assert(64 * 2 * 4 * 2 * (kParkFromN - 1) >= 3 * 64 * (kParkFromN - 1) + 256, "reset / observe of an RVO env park keys, gaps and velocities in the idle ORCA scratch");
    const int row_floats = k.width + 2;
    int tile_rows = (int)(9216 / ((size_t)row_floats * sizeof(float))) & ~3;
    if (tile_rows < 4) tile_rows = 4;
    const bool one_pass_fits = (size_t)(lds_floats_block() + lds_floats_fixed(N) + k.rvo_lds_floats + lanes * row_floats + 4) * sizeof(float) <= 65536;
    if (tile_rows > lanes || (e->latency_mode && one_pass_fits)) tile_rows = lanes;
    if (const char *ov = std::getenv("CAVOID_TILE_ROWS")) { int v = std::atoi(ov); if (v >= 1 && v <= lanes) tile_rows = v; }
    // one wavefront must fit the 64 KiB a workgroup may ask for: with the ORCA scratch of a large N the tile shrinks
    auto wave_bytes = [&](int rows) {
        int tile = (rows * row_floats + 3) & ~3;
        if (tile < k.park_floats) tile = k.park_floats;
        return (size_t)(lds_floats_fixed(N) + k.rvo_lds_floats + tile) * sizeof(float);
    };
    while (tile_rows > 4 && wave_bytes(tile_rows) + lds_floats_block() * sizeof(float) > 65536) tile_rows -= 4;
    k.tile_rows = tile_rows;
    const size_t per_wave = wave_bytes(tile_rows);
    int wpb = (int)(((size_t)65536 - lds_floats_block() * sizeof(float)) / per_wave);
    if (wpb < 1) { cavoid_destroy(e); return CAVOID_EUNSUPPORTED; }
    if (wpb > 4) wpb = 4;
    if (const char *ov = std::getenv("CAVOID_WAVES_PER_BLOCK")) { int v = std::atoi(ov); if (v >= 1 && v <= wpb) wpb = v; }
    e->waves_per_block = wpb;
    e->grid = grid_for(e, num_worlds);
    *out = e;
    return CAVOID_OK;
}

extern "C" void cavoid_destroy(cavoid_env *e) {
    if (!e) return;
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->slab) (void)hipFree(e->slab);
    if (e->pool_slab) (void)hipFree(e->pool_slab);
    delete e;
}

extern "C" int64_t cavoid_num_worlds(const cavoid_env *e) { return e ? e->W : 0; }
extern "C" int32_t cavoid_obs_width(const cavoid_env *e) { return e ? e->k.width : 0; }

template <int MODE>
static int launch(cavoid_env *e, const KIO &io, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
    // the 'everything' instantiations (cavoid_rvo.hip): ORCA agents; box scenarios generated inside the auto-reset step
    if (((MODE == MODE_STEP || MODE == MODE_STEP_AUTORESET) && e->k.rvo_enabled) ||
        (MODE == MODE_STEP_AUTORESET && e->k.gen_mode == 1 && e->k.pool_size <= 0))
        return cavoid_launch_rvo(e, MODE, io, s, ev_start, ev_stop);
    if (MODE == MODE_STEP_AUTORESET) {                     // small batches: the step spread over four wavefronts per tile, where that form carries it
        const int rc = cavoid_launch_quad(e, io, s, ev_start, ev_stop);
        if (rc != CAVOID_EUNSUPPORTED) return rc;
    }
    return launch_on<MODE>(e, e->k, e->st, e->grid, io, s, ev_start, ev_stop);
}

// scenario look-ahead (cavoid.h, gen_lookahead): every world's ring must hold the scenarios of the episodes the next n_steps steps can start
// (at most one restart per world and step) + the one a restarted world prefetches.  The host keeps a guaranteed-cover budget: a refill launch
// (ahead_fill_kernel: only the episodes consumed since the last refill are generated, ~6 us) goes in front of a stepping launch when the budget
// does not cover it -- one per ~R / K launches of K steps, one per R - 2 one-step launches.  A sequence captured into a hipGraph always carries
// the refill (a no-op when nothing is missing), and once such a graph exists every launch does (replays consume episodes the host does not see).
// (Measured and dropped, profiles/r05_e_lookahead.txt: the refill on a side stream beside the previous launch -- the cross-stream event wait costs
// what the refill launch costs; the refill inside env_relay_kernel, by its loader role in idle time -- +5 % on the kernel -- or by extra
// workgroups of the same launch -- they displace tile workgroups, which must all be resident.)
int cavoid_ahead_prepare(cavoid_env *e, int32_t n_steps, hipStream_t s, hipEvent_t *timed_start) {
    if (!e || e->ahead_R <= 0) return CAVOID_OK;
    if (n_steps + 1 > e->ahead_R) return CAVOID_EUNSUPPORTED;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    const bool capturing = cap != hipStreamCaptureStatusNone;
    // a sequence being captured into a hipGraph cannot rely on host-side bookkeeping at replay time: it always carries the refill
    if (capturing) e->ahead_always = true;
    if (!e->ahead_always && e->ahead_primed && e->ahead_budget >= n_steps + 1) return CAVOID_OK;
    const int64_t waves = (e->W + e->k.wpw - 1) / e->k.wpw;
    // The missing episodes of a world are dealt over grid.y wavefronts (4 in the steady state -- a world rarely consumed more since the last
    // refill --, 32 for the first fill of R episodes per world): they all read the bookkeeping array `cur` and the y = 0 one writes the other,
    // which becomes `cur`.  Once a hipGraph holds a refill (its kernel arguments are frozen) every refill works IN PLACE on one array with
    // grid.y = 1, so that the graph's replays and the eager launches between them see the same bookkeeping.
    const bool in_place = e->ahead_always;
    const dim3 grid((unsigned)((waves + 3) / 4), in_place ? 1u : (e->ahead_primed ? 4u : 32u)), block(256);
    const uint32_t *hi_in = e->ahead_hi[e->ahead_cur];
    uint32_t *hi_out = e->ahead_hi[in_place ? e->ahead_cur : (e->ahead_cur ^ 1)];
    hipEvent_t ev_fill = (timed_start && !capturing) ? *timed_start : nullptr;
    if (ev_fill) *timed_start = nullptr;                     // (the refill kernel opens the timed interval)
#define CAVOID_AHEAD_CASE(NN) case NN: \
        if (ev_fill) hipExtLaunchKernelGGL((ahead_fill_kernel<NN>), grid, block, 0, s, ev_fill, nullptr, 0, e->k, e->st.episode, hi_in, hi_out, e->pool, e->ahead_R); \
        else hipLaunchKernelGGL((ahead_fill_kernel<NN>), grid, block, 0, s, e->k, e->st.episode, hi_in, hi_out, e->pool, e->ahead_R); \
        break;
    switch (e->cfg.max_agents) {
#ifdef CAVOID_DEV_ONLY_N
        CAVOID_AHEAD_CASE(4) CAVOID_AHEAD_CASE(10)
#else
        CAVOID_AHEAD_CASE(1) CAVOID_AHEAD_CASE(2) CAVOID_AHEAD_CASE(3) CAVOID_AHEAD_CASE(4) CAVOID_AHEAD_CASE(5) CAVOID_AHEAD_CASE(6)
        CAVOID_AHEAD_CASE(7) CAVOID_AHEAD_CASE(8) CAVOID_AHEAD_CASE(9) CAVOID_AHEAD_CASE(10) CAVOID_AHEAD_CASE(11) CAVOID_AHEAD_CASE(12)
        CAVOID_AHEAD_CASE(13) CAVOID_AHEAD_CASE(14) CAVOID_AHEAD_CASE(15) CAVOID_AHEAD_CASE(16)
#endif
        default: return CAVOID_EUNSUPPORTED;
    }
#undef CAVOID_AHEAD_CASE
    HIP_TRY(hipGetLastError());
    if (!in_place) e->ahead_cur ^= 1;
    if (!capturing) { e->ahead_primed = true; e->ahead_budget = e->ahead_R; }
    return CAVOID_OK;
}
// after a stepping launch of n_steps steps: what the rings are still guaranteed to cover (a world restarts at most once per step)
void cavoid_ahead_consumed(cavoid_env *e, int32_t n_steps) {
    if (!e || e->ahead_R <= 0) return;
    e->ahead_budget = e->ahead_budget > n_steps ? e->ahead_budget - n_steps : 0;
}

// (re)fill the scenario pool for the current seed: the RESET kernel run over the pool buffer as
// worlds 0..P-1 of episode k.pool_epoch, generator (GEN v1 or v2) in-kernel
static int fill_pool(cavoid_env *e, hipStream_t s) {
    if (e->ahead_R > 0) {                                      // look-ahead rings: nothing of the old seed / episodes stays valid
        e->ahead_budget = 0;
        e->ahead_primed = false;
        HIP_TRY(hipMemsetAsync(e->ahead_hi[0], 0xFF, 2 * (size_t)e->W * sizeof(uint32_t), s));
        return CAVOID_OK;
    }
    if (e->pool_size <= 0) return CAVOID_OK;
    HIP_TRY(hipMemsetAsync(e->pool_episode, 0xFF, (size_t)e->pool_size * sizeof(uint32_t), s));
    KCfg k = e->k;
    k.num_worlds = e->pool_size;
    k.world_offset = 0;
    k.pool_size = 0;
    KState st{};
    st.episode = e->pool_episode;                              // the only world-buffer field the fill launch touches
    KIO io{};
    io.pool_out = e->pool;
    io.n_steps = 1; io.obs_stride = k.width;
    return launch_on<MODE_RESET>(e, k, st, grid_for(e, e->pool_size), io, s, nullptr, nullptr);
}

extern "C" int cavoid_seed(cavoid_env *e, uint64_t seed, const uint32_t *episode, void *stream) {
    if (!e) return CAVOID_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    e->k.seed_lo = (uint32_t)seed;
    e->k.seed_hi = (uint32_t)(seed >> 32);
    if (episode) HIP_TRY(hipMemcpyAsync(e->st.episode, episode, (size_t)e->W * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    else HIP_TRY(hipMemsetAsync(e->st.episode, 0xFF, (size_t)e->W * sizeof(uint32_t), s));
    return fill_pool(e, s);
}

extern "C" int cavoid_pool_refresh(cavoid_env *e, uint32_t epoch, void *stream) {
    if (!e) return CAVOID_EINVAL;
    e->k.pool_epoch = epoch;
    e->cfg.gen_pool_epoch = epoch;
    return fill_pool(e, static_cast<hipStream_t>(stream));
}

extern "C" int cavoid_get_episode(cavoid_env *e, uint32_t *out, void *stream) {
    if (!e || !out) return CAVOID_EINVAL;
    HIP_TRY(hipMemcpyAsync(out, e->st.episode, (size_t)e->W * sizeof(uint32_t), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return CAVOID_OK;
}

extern "C" int cavoid_set_state(cavoid_env *e, const double *f64, const float *f32, const uint32_t *flags, void *stream) {
    if (!e || !f64 || !f32 || !flags) return CAVOID_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t A = (size_t)e->A;
    HIP_TRY(hipMemcpyAsync(e->st.px, f64, 4 * A * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(e->st.gx, f32, 5 * A * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(e->st.flags, flags, A * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    return CAVOID_OK;
}

extern "C" int cavoid_get_state(cavoid_env *e, double *f64, float *f32, uint32_t *flags, void *stream) {
    if (!e || !f64 || !f32 || !flags) return CAVOID_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t A = (size_t)e->A;
    HIP_TRY(hipMemcpyAsync(f64, e->st.px, 4 * A * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(f32, e->st.gx, 5 * A * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(flags, e->st.flags, A * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    return CAVOID_OK;
}

// output wiring of one launch: plain (obs / rewards / done arrays) or packed (one record per agent)
static KIO plain_io(const cavoid_env *e, float *obs, float *rew, uint8_t *done, uint8_t *game_over) {
    KIO io{};
    io.obs = obs; io.rew = rew; io.done = done; io.game_over = game_over;
    io.obs_stride = e->k.width; io.packed = 0; io.n_steps = 1;
    return io;
}
static KIO packed_io(const cavoid_env *e, float *packed, uint8_t *game_over) {
    KIO io{};
    io.obs = packed; io.game_over = game_over;
    io.obs_stride = e->k.width + 2; io.packed = 1; io.n_steps = 1;
    return io;
}

extern "C" int32_t cavoid_packed_width(const cavoid_env *e) { return e ? e->k.width + 2 : 0; }

extern "C" int cavoid_reset(cavoid_env *e, const uint8_t *world_mask, float *obs, void *stream) {
    if (!e) return CAVOID_EINVAL;
    KIO io = plain_io(e, obs, nullptr, nullptr, nullptr);
    io.mask = world_mask;
    if (int rc = cavoid_ahead_prepare(e, 1, static_cast<hipStream_t>(stream))) return rc;     // (a reset starts every masked world's NEXT episode)
    cavoid_ahead_consumed(e, 1);
    return launch<MODE_RESET>(e, io, static_cast<hipStream_t>(stream));
}

extern "C" int cavoid_reset_packed(cavoid_env *e, const uint8_t *world_mask, float *packed, void *stream) {
    if (!e || !packed) return CAVOID_EINVAL;
    KIO io = packed_io(e, packed, nullptr);
    io.mask = world_mask;
    if (int rc = cavoid_ahead_prepare(e, 1, static_cast<hipStream_t>(stream))) return rc;
    cavoid_ahead_consumed(e, 1);
    return launch<MODE_RESET>(e, io, static_cast<hipStream_t>(stream));
}

extern "C" int cavoid_observe(cavoid_env *e, float *obs, void *stream) {
    if (!e || !obs) return CAVOID_EINVAL;
    return launch<MODE_OBSERVE>(e, plain_io(e, obs, nullptr, nullptr, nullptr), static_cast<hipStream_t>(stream));
}

extern "C" int cavoid_observe_packed(cavoid_env *e, float *packed, void *stream) {
    if (!e || !packed) return CAVOID_EINVAL;
    return launch<MODE_OBSERVE>(e, packed_io(e, packed, nullptr), static_cast<hipStream_t>(stream));
}

static int step_args(cavoid_env *e, const void *actions, float *rew, uint8_t *done, uint8_t *go) {
    return (e && actions && rew && done && go) ? CAVOID_OK : CAVOID_EINVAL;
}

extern "C" int cavoid_step(cavoid_env *e, const int32_t *actions, float *obs, float *rew, uint8_t *done, uint8_t *game_over, void *stream) {
    if (step_args(e, actions, rew, done, game_over) != CAVOID_OK) return CAVOID_EINVAL;
    if (e->cfg.dynamics == CAVOID_DYN_HOLONOMIC) return CAVOID_EINVAL;   // holonomic needs velocity actions
    KIO io = plain_io(e, obs, rew, done, game_over);
    io.actions = actions;
    return launch<MODE_STEP>(e, io, static_cast<hipStream_t>(stream));
}

extern "C" int cavoid_step_packed(cavoid_env *e, const int32_t *actions, float *packed, uint8_t *game_over, void *stream) {
    if (!e || !actions || !packed || !game_over) return CAVOID_EINVAL;
    if (e->cfg.dynamics == CAVOID_DYN_HOLONOMIC) return CAVOID_EINVAL;
    KIO io = packed_io(e, packed, game_over);
    io.actions = actions;
    return launch<MODE_STEP>(e, io, static_cast<hipStream_t>(stream));
}

extern "C" int cavoid_step_continuous(cavoid_env *e, const float *actions, float *obs, float *rew, uint8_t *done, uint8_t *game_over, void *stream) {
    if (step_args(e, actions, rew, done, game_over) != CAVOID_OK) return CAVOID_EINVAL;
    KIO io = plain_io(e, obs, rew, done, game_over);
    io.cont = actions;
    return launch<MODE_STEP>(e, io, static_cast<hipStream_t>(stream));
}

// the auto-reset step, n_steps >= 1 steps in ONE launch.  Latency mode (small batch with a scenario pool) takes the
// register-prefetch instantiation for multi-step launches (and for single steps when CAVOID_PREFETCH_POOL=1).
// Continuous actions (`cont` != null: float [n_steps][W,N,2], (speed, heading change) for the unicycle dynamics, a velocity for the
// holonomic ones -- cavoid_step_continuous's action form) take the same launch forms: one step per launch, or the in-launch step loop with
// every step's outputs in its own slot and restarts from the pool / the look-ahead rings / the in-step generator.  (The role-split relay
// and the two-wavefront pipeline decode table actions only: a continuous K-step launch runs the single-wavefront loop kernels.)
static int launch_autoreset(cavoid_env *e, KIO io, const int32_t *actions, int64_t action_stride, int32_t n_steps, int64_t out_step_stride,
                            hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, const float *cont = nullptr) {
    if (e->cfg.dynamics == CAVOID_DYN_HOLONOMIC && !cont) return CAVOID_EINVAL;     // holonomic needs velocity actions
    if (n_steps < 1 || action_stride < 0) return CAVOID_EINVAL;
    if (cont && n_steps > 1 && action_stride != 0 && action_stride < 2 * e->A) return CAVOID_EINVAL;   // (float slices must not overlap)
    if (out_step_stride != 0 && out_step_stride < e->W) return CAVOID_EINVAL;   // slots of consecutive steps must not overlap
    io.out_step_stride = n_steps > 1 ? out_step_stride : 0;
    io.actions = actions;
    io.cont = cont;
    io.action_stride = action_stride;
    io.n_steps = n_steps;
    if (int rc = cavoid_ahead_prepare(e, n_steps, s, &ev_start)) return rc;      // (a timed launch includes its look-ahead refill, when it needs one)
    const int rc = (n_steps > 1 || e->prefetch_single)           // the in-launch step loop lives in cavoid_multistep.hip
                       ? cavoid_launch_multistep(e, io, e->latency_mode != 0, s, ev_start, ev_stop)
                       : launch<MODE_STEP_AUTORESET>(e, io, s, ev_start, ev_stop);
    cavoid_ahead_consumed(e, n_steps);
    return rc;
}

extern "C" int cavoid_step_autoreset(cavoid_env *e, const int32_t *actions, float *obs, float *rew, uint8_t *done, uint8_t *game_over, void *stream) {
    if (step_args(e, actions, rew, done, game_over) != CAVOID_OK) return CAVOID_EINVAL;
    return launch_autoreset(e, plain_io(e, obs, rew, done, game_over), actions, 0, 1, 0, static_cast<hipStream_t>(stream));
}

extern "C" int cavoid_step_autoreset_n(cavoid_env *e, const int32_t *actions, int64_t action_stride, int32_t n_steps, int64_t out_step_stride,
                                       float *obs, float *rew, uint8_t *done, uint8_t *game_over, void *stream) {
    if (step_args(e, actions, rew, done, game_over) != CAVOID_OK || n_steps < 0) return CAVOID_EINVAL;
    if (n_steps == 0) return CAVOID_OK;
    return launch_autoreset(e, plain_io(e, obs, rew, done, game_over), actions, action_stride, n_steps, out_step_stride,
                            static_cast<hipStream_t>(stream));
}

extern "C" int cavoid_step_autoreset_packed(cavoid_env *e, const int32_t *actions, int64_t action_stride, int32_t n_steps, int64_t out_step_stride,
                                            float *packed, uint8_t *game_over, void *stream) {
    if (!e || !actions || !packed || !game_over || n_steps < 0) return CAVOID_EINVAL;
    if (n_steps == 0) return CAVOID_OK;
    return launch_autoreset(e, packed_io(e, packed, game_over), actions, action_stride, n_steps, out_step_stride,
                            static_cast<hipStream_t>(stream));
}

extern "C" int cavoid_step_continuous_autoreset(cavoid_env *e, const float *actions, float *obs, float *rew, uint8_t *done, uint8_t *game_over,
                                               void *stream) {
    if (step_args(e, actions, rew, done, game_over) != CAVOID_OK) return CAVOID_EINVAL;
    return launch_autoreset(e, plain_io(e, obs, rew, done, game_over), nullptr, 0, 1, 0, static_cast<hipStream_t>(stream), nullptr, nullptr, actions);
}

extern "C" int cavoid_step_continuous_autoreset_n(cavoid_env *e, const float *actions, int64_t action_stride, int32_t n_steps,
                                                 int64_t out_step_stride, float *obs, float *rew, uint8_t *done, uint8_t *game_over, void *stream) {
    if (step_args(e, actions, rew, done, game_over) != CAVOID_OK || n_steps < 0) return CAVOID_EINVAL;
    if (n_steps == 0) return CAVOID_OK;
    return launch_autoreset(e, plain_io(e, obs, rew, done, game_over), nullptr, action_stride, n_steps, out_step_stride,
                            static_cast<hipStream_t>(stream), nullptr, nullptr, actions);
}

extern "C" int cavoid_step_continuous_autoreset_packed(cavoid_env *e, const float *actions, int64_t action_stride, int32_t n_steps,
                                                      int64_t out_step_stride, float *packed, uint8_t *game_over, void *stream) {
    if (!e || !actions || !packed || !game_over || n_steps < 0) return CAVOID_EINVAL;
    if (n_steps == 0) return CAVOID_OK;
    return launch_autoreset(e, packed_io(e, packed, game_over), nullptr, action_stride, n_steps, out_step_stride,
                            static_cast<hipStream_t>(stream), nullptr, nullptr, actions);
}

extern "C" int cavoid_step_autoreset_n_timed(cavoid_env *e, const int32_t *actions, int64_t action_stride, int32_t n_steps,
                                             int32_t steps_per_launch, int64_t out_step_stride, float *obs, float *rew, uint8_t *done,
                                             uint8_t *game_over, void *stream, float *mean_launch_ms) {
    if (step_args(e, actions, rew, done, game_over) != CAVOID_OK || n_steps < 1 || steps_per_launch < 1 || !mean_launch_ms) return CAVOID_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    constexpr int kPool = 128;
    hipEvent_t ev[2 * kPool];
    for (int i = 0; i < 2 * kPool; ++i) HIP_TRY(hipEventCreate(&ev[i]));
    const KIO io = plain_io(e, obs, rew, done, game_over);
    double total_ms = 0.0;
    int rc = CAVOID_OK, launches = 0;
    int32_t t0 = 0;
    while (t0 < n_steps && rc == CAVOID_OK) {
        int n = 0;
        for (; n < kPool && t0 < n_steps && rc == CAVOID_OK; ++n) {
            const int32_t k = (n_steps - t0) < steps_per_launch ? (n_steps - t0) : steps_per_launch;
            rc = launch_autoreset(e, io, actions + (int64_t)t0 * action_stride, action_stride, k, out_step_stride, s, ev[2 * n], ev[2 * n + 1]);
            t0 += k;
        }
        if (rc != CAVOID_OK) break;
        if (hipStreamSynchronize(s) != hipSuccess) { rc = CAVOID_EHIP; break; }
        for (int t = 0; t < n; ++t) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev[2 * t], ev[2 * t + 1]) != hipSuccess) { rc = CAVOID_EHIP; break; }
            total_ms += ms;
        }
        launches += n;
    }
    for (int i = 0; i < 2 * kPool; ++i) (void)hipEventDestroy(ev[i]);
    if (rc == CAVOID_OK) *mean_launch_ms = (float)(total_ms / launches);
    return rc;
}

// the slots driven by one scripted policy (CAVOID_POLICY_FROZEN_NET: the rows a frozen network must act for)
__global__ void __launch_bounds__(256) policy_rows_kernel(const uint32_t *flags, int64_t slots, uint32_t policy, int only_running,
                                                          int32_t *row_index, int32_t *row_count) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool take = false;
    if (a < slots) {
        const uint32_t f = flags[a];
        take = (f & CAVOID_F_PRESENT) && ((f >> CAVOID_F_POLICY_SHIFT) & CAVOID_F_POLICY_MASK) == policy &&
               !(only_running && (f & CAVOID_F_DONE_MASK));
    }
    const unsigned long long mask = __ballot(take);
    if (mask == 0ull) return;
    const int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(row_count, __popcll(mask));
    base = __shfl(base, leader, 64);
    if (take) row_index[base + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)a;
}
__global__ void zero_counter_kernel(int32_t *counter) { *counter = 0; }

extern "C" int cavoid_policy_rows(cavoid_env *e, int32_t policy_id, int32_t only_running, int32_t *row_index, int32_t *row_count, void *stream) {
    if (!e || !row_index || !row_count || policy_id < 0 || policy_id > (int32_t)CAVOID_F_POLICY_MASK) return CAVOID_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(zero_counter_kernel, dim3(1), dim3(1), 0, s, row_count);          // (a kernel, not a memset node: hipGraph replays)
    hipLaunchKernelGGL(policy_rows_kernel, dim3((unsigned)((e->A + 255) / 256)), dim3(256), 0, s, e->st.flags, e->A, (uint32_t)policy_id,
                       only_running ? 1 : 0, row_index, row_count);
    HIP_TRY(hipGetLastError());
    return CAVOID_OK;
}

#ifdef CAVOID_TRACE
// development build only: point the kernels' phase-stamp buffer (u64 [waves][16]) somewhere
int cavoid_debug_trace_multistep(unsigned long long *dev_ptr);
int cavoid_debug_trace_rvo(unsigned long long *dev_ptr);
int cavoid_debug_trace_relay(unsigned long long *dev_ptr);
extern "C" int cavoid_debug_trace(unsigned long long *dev_ptr) {
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &dev_ptr, sizeof(dev_ptr)));
    const int rc = cavoid_debug_trace_multistep(dev_ptr);
    if (rc != CAVOID_OK) return rc;
    const int rc2 = cavoid_debug_trace_rvo(dev_ptr);
    return rc2 != CAVOID_OK ? rc2 : cavoid_debug_trace_relay(dev_ptr);
}
#endif

extern "C" int cavoid_timer_begin(cavoid_env *e, void *stream) {
    if (!e) return CAVOID_EINVAL;
    HIP_TRY(hipEventRecord(e->ev0, static_cast<hipStream_t>(stream)));
    return CAVOID_OK;
}

extern "C" int cavoid_timer_end(cavoid_env *e, void *stream, float *elapsed_ms) {
    if (!e || !elapsed_ms) return CAVOID_EINVAL;
    HIP_TRY(hipEventRecord(e->ev1, static_cast<hipStream_t>(stream)));
    HIP_TRY(hipEventSynchronize(e->ev1));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, e->ev0, e->ev1));
    return CAVOID_OK;
}
