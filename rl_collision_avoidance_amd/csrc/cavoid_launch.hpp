// cavoid_launch.hpp -- host-side launch plumbing shared by the translation units that instantiate env_kernel:
// cavoid_capi.hip (single-step / reset / observe instantiations) and cavoid_multistep.hip (the instantiations with the
// in-launch step loop, compiled with -mllvm -disable-machine-licm, see build.py).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "cavoid.h"
#include "cavoid_host.hpp"
#include "cavoid_kernels.hpp"

struct cavoid_env {
    int device = 0;
    int64_t W = 0, A = 0, world_offset = 0;
    cavoid_cfg cfg{};
    cavoid::KCfg k{};
    cavoid::KState st{};
    cavoid::PoolRec *pool = nullptr;     // pre-generated scenarios (GEN v1 worlds 0..P-1, episode 0), 64-byte records
    uint32_t *pool_episode = nullptr;   // [P] scratch episode counters for the fill launch
    int64_t pool_size = 0;
    int ahead_R = 0;                    // scenario look-ahead (cfg.gen_lookahead): `pool` is then every world's ring of R records, filled by ahead_fill_kernel
    uint32_t *ahead_hi[2] = {nullptr, nullptr};   // [W] highest episode in each world's ring (0xFFFFFFFF: none): a refill reads [ahead_cur], writes the other
    int ahead_cur = 0;
    int ahead_budget = 0;               // restarts per world the ring is still guaranteed to cover without a refill
    bool ahead_primed = false;          // the rings have been filled once for the current seed / episodes
    bool ahead_always = false;          // a hipGraph holding stepping launches of this env exists: replays consume episodes the host does not see,
                                        // so from then on every launch carries the refill (a no-op when nothing is missing)
    void *slab = nullptr;
    void *pool_slab = nullptr;
    double *d_actions = nullptr;
    int waves_per_block = 4;
    int grid = 0;
    int pipeline = 2;            // latency mode, multi-step launches: 2 = env_relay_kernel (roles on 5-7 wavefronts per tile), 1 = env_pipe_kernel
                                 // (two wavefronts per tile), 0 = one wavefront per tile (CAVOID_PIPELINE)
    int relay_consumers = 3;     // observation wavefronts per tile of env_relay_kernel (CAVOID_RELAY_CONSUMERS, 1..4)
    int latency_mode = 0;        // small batch: multi-step launches keep the next pool record in registers (MODE_STEP_AUTORESET_PF)
    int quad = -1;               // one-step auto-reset launches with four cooperating wavefronts per tile (env_quad_kernel): -1 = where it pays
                                 // (<= 512 tiles), 0 / 1 = never / wherever it can run (CAVOID_QUAD)
    int prefetch_single = 0;     // ... and single-step launches too (CAVOID_PREFETCH_POOL=1; costs 64 B of reads per agent-step)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};


namespace cavoid {

template <int MODE, bool RVO = false>
static inline int launch_on(cavoid_env *e, const KCfg &k, const KState &st, int grid_x, const KIO &io, hipStream_t s,
                     hipEvent_t ev_start, hipEvent_t ev_stop) {
    const dim3 grid(grid_x), block(64 * e->waves_per_block);
    // dynamic LDS: the action table + per wavefront the staging arrays and an obs tile of this launch's row width
    const int row = io.obs ? io.obs_stride : k.width;
    int tile = (k.tile_rows * row + 3) & ~3;
    if (tile < k.park_floats) tile = k.park_floats;
    const size_t lds = (size_t)(lds_floats_block() + e->waves_per_block * (lds_floats_fixed(e->cfg.max_agents) + k.rvo_lds_floats + tile)) * sizeof(float);
#define CAVOID_CASE(NN) \
    case NN:                                                                                                            \
        if (ev_start || ev_stop)                                                                                        \
            hipExtLaunchKernelGGL((env_kernel<NN, MODE, RVO>), grid, block, lds, s, ev_start, ev_stop, 0, k, st, e->pool, io); \
        else /* plain launch: capturable into a hipGraph */                                                             \
            hipLaunchKernelGGL((env_kernel<NN, MODE, RVO>), grid, block, lds, s, k, st, e->pool, io);                         \
        break;
    switch (e->cfg.max_agents) {
#ifdef CAVOID_DEV_ONLY_N   /* development builds: instantiate two sizes only (compile time) */
        CAVOID_CASE(4) CAVOID_CASE(10)
#else
        CAVOID_CASE(1) CAVOID_CASE(2) CAVOID_CASE(3) CAVOID_CASE(4) CAVOID_CASE(5) CAVOID_CASE(6)
        CAVOID_CASE(7) CAVOID_CASE(8) CAVOID_CASE(9) CAVOID_CASE(10) CAVOID_CASE(11) CAVOID_CASE(12)
        CAVOID_CASE(13) CAVOID_CASE(14) CAVOID_CASE(15) CAVOID_CASE(16)
#endif
        default: return CAVOID_EUNSUPPORTED;
    }
#undef CAVOID_CASE
    HIP_TRY(hipGetLastError());
    return CAVOID_OK;
}


// The two-wavefront pipelined step loop (env_pipe_kernel): one 128-thread workgroup per tile.  Returns CAVOID_EUNSUPPORTED when
// its LDS (two staging + two hand-over buffers, the obs tile, the ORCA scratch) does not fit 64 KiB: the caller then takes
// the single-wavefront loop.
template <bool RVO>
static inline int launch_pipe(cavoid_env *e, const KIO &io, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const KCfg &k = e->k;
    if (!io.obs) return CAVOID_EUNSUPPORTED;               // (the consumer wavefront IS the observation: env_kernel guards a null obs)
    if (k.gen_mode == 1 && k.pool_size <= 0) return CAVOID_EUNSUPPORTED;   // (box scenarios generated inside the step: env_kernel's restart)
    const int64_t tiles = (e->W + k.wpw - 1) / k.wpw;
    // it pays while the chip can hold both wavefronts of every tile at <= 2 per SIMD (1024 SIMDs): measured at N = 4,
    // 32-step launches: 512 / 1024 tiles 2.50 vs 3.02 / 3.06 us per step, 2048 tiles 5.05 vs 3.75; N = 10, 1366 tiles 9.4 vs 8.2
    if (tiles > 1024) return CAVOID_EUNSUPPORTED;
    const int row = io.obs ? io.obs_stride : k.width;
    int tile = (k.tile_rows * row + 3) & ~3;
    if (tile < k.park_floats) tile = k.park_floats;
    const size_t tail = (size_t)(tile + k.rvo_lds_floats) * sizeof(float);
    const dim3 grid((unsigned)tiles), block(128);
#define CAVOID_PIPE_CASE(NN) \
    case NN: {                                                                                                          \
        const size_t lds = pipe_lds_fixed_bytes<NN>() + tail;                                                          \
        if (lds > 65536) return CAVOID_EUNSUPPORTED;                                                                    \
        if (ev_start || ev_stop)                                                                                        \
            hipExtLaunchKernelGGL((env_pipe_kernel<NN, RVO>), grid, block, lds, s, ev_start, ev_stop, 0, k, e->st, e->pool, io); \
        else                                                                                                            \
            hipLaunchKernelGGL((env_pipe_kernel<NN, RVO>), grid, block, lds, s, k, e->st, e->pool, io);                  \
        break;                                                                                                          \
    }
    switch (e->cfg.max_agents) {
#ifdef CAVOID_DEV_ONLY_N
        CAVOID_PIPE_CASE(4) CAVOID_PIPE_CASE(10)
#else
        CAVOID_PIPE_CASE(1) CAVOID_PIPE_CASE(2) CAVOID_PIPE_CASE(3) CAVOID_PIPE_CASE(4) CAVOID_PIPE_CASE(5) CAVOID_PIPE_CASE(6)
        CAVOID_PIPE_CASE(7) CAVOID_PIPE_CASE(8) CAVOID_PIPE_CASE(9) CAVOID_PIPE_CASE(10) CAVOID_PIPE_CASE(11) CAVOID_PIPE_CASE(12)
        CAVOID_PIPE_CASE(13) CAVOID_PIPE_CASE(14) CAVOID_PIPE_CASE(15) CAVOID_PIPE_CASE(16)
#endif
        default: return CAVOID_EUNSUPPORTED;
    }
#undef CAVOID_PIPE_CASE
    HIP_TRY(hipGetLastError());
    return CAVOID_OK;
}

}  // namespace cavoid

// scenario look-ahead: make sure every world's ring covers the restarts `n_steps` more steps can bring (a no-op without look-ahead);
// CAVOID_EUNSUPPORTED when n_steps + 1 > R (cavoid_capi.hip)
// timed_start (may be null): a start event the caller wants recorded where the launch's work begins -- when a refill is launched it is recorded
// in front of THAT kernel and *timed_start is set to null (the stepping kernel behind it then records only its stop event), so that a timed
// launch includes its refill (cavoid_step_autoreset_n_timed)
int cavoid_ahead_prepare(cavoid_env *e, int32_t n_steps, hipStream_t s, hipEvent_t *timed_start = nullptr);
void cavoid_ahead_consumed(cavoid_env *e, int32_t n_steps);     // call after the stepping launch that cavoid_ahead_prepare(n_steps) preceded
// env_relay_kernel (cavoid_relay.hip): CAVOID_EUNSUPPORTED when the batch is too large for it or its LDS does not fit
int cavoid_launch_relay(cavoid_env *e, const cavoid::KIO &io, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop);
// multi-step auto-reset launch (cavoid_multistep.hip): prefetch != 0 -> MODE_STEP_AUTORESET_PF, else MODE_STEP_AUTORESET_N
// env_quad_kernel (cavoid_quad.hip): CAVOID_EUNSUPPORTED when the configuration or the launch is not one it carries
int cavoid_launch_quad(cavoid_env *e, const cavoid::KIO &io, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop);
int cavoid_launch_multistep(cavoid_env *e, const cavoid::KIO &io, bool prefetch, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop);
// any stepping mode for an env with rvo_enabled (cavoid_rvo.hip: the instantiations that carry the ORCA policy)
int cavoid_launch_rvo(cavoid_env *e, int mode, const cavoid::KIO &io, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop);
