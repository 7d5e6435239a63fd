// cavoid_relay.hip -- env_relay_kernel instantiations (cavoid_relay.hpp): the in-launch step loop of small batches with the
// step cut into roles on several wavefronts of one workgroup per tile.  Own translation unit, compiled with
// -mllvm -disable-machine-licm like the other step-loop units (build.py).
#include "cavoid_launch.hpp"
#include "cavoid_relay.hpp"

using namespace cavoid;

int cavoid_launch_relay(cavoid_env *e, const KIO &io, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const KCfg &k = e->k;
    if (k.rvo_enabled || k.pool_size <= 0 || !io.obs || !io.actions || io.cont) return CAVOID_EUNSUPPORTED;
    if (k.switches & kSwSkipDonePairs) return CAVOID_EUNSUPPORTED;     // (U4 flipped: P would need to know who was frozen; the other loop forms do)
    const int64_t tiles = (e->W + k.wpw - 1) / k.wpw;
    // every tile's workgroup must be resident at once (256 CUs x 2 workgroups): beyond that the tiles run in rounds
    // (measured with 1024 tiles, N = 4: 3.87 vs 2.50 us per step for the two-wavefront pipeline; 1366 tiles, N = 10: 25 vs 8.3)
    if (tiles > 512) return CAVOID_EUNSUPPORTED;
    // wide rows make the observation wavefronts the limit (and N >= 9 spills): measured at 512 tiles, us per step, this kernel /
    // env_pipe_kernel: N = 2 1.42 / 1.97, 3 1.53 / 2.33, 5 1.86 / 2.91, 6 3.11 / 3.29, 8 4.35 / 3.88, 10 8.85 / 4.80
    if (e->cfg.max_agents > kRelayMaxAgents) return CAVOID_EUNSUPPORTED;
    const int row = io.obs_stride;
    const int tile_floats = (k.tile_rows * row + 3) & ~3;
    if (k.tile_rows < k.wpw * e->cfg.max_agents) return CAVOID_EUNSUPPORTED;   // one pass per step only
    const dim3 grid((unsigned)tiles);
#define CAVOID_RELAY_CASE(NN) \
    case NN: {                                                                                                          \
        int nc = e->relay_consumers;                                                                                    \
        while (nc > 1 && relay_lds_fixed_bytes<NN>() + (size_t)nc * tile_floats * sizeof(float) > kRelayLdsLimit) --nc; \
        const size_t lds = relay_lds_fixed_bytes<NN>() + (size_t)nc * tile_floats * sizeof(float);                      \
        /* one observation wavefront cannot keep up with the loop: the two-wavefront pipeline is the better form then */  \
        if (lds > kRelayLdsLimit || (nc < 2 && e->relay_consumers >= 2)) return CAVOID_EUNSUPPORTED;                    \
        if (lds > 65536) {                        /* beyond the default dynamic-LDS limit: opted into once per instantiation */ \
            static size_t allowed = 0;                                                                                  \
            if (allowed < lds) {                                                                                        \
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(env_relay_kernel<NN>),                       \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRelayLdsLimit));          \
                allowed = kRelayLdsLimit;                                                                               \
            }                                                                                                           \
        }                                                                                                               \
        const dim3 block(64 * (3 + nc));                                                                                \
        if (ev_start || ev_stop)                                                                                        \
            hipExtLaunchKernelGGL((env_relay_kernel<NN>), grid, block, lds, s, ev_start, ev_stop, 0, k, e->st, e->pool, io);  \
        else                                                                                                            \
            hipLaunchKernelGGL((env_relay_kernel<NN>), grid, block, lds, s, k, e->st, e->pool, io);                     \
        break;                                                                                                          \
    }
    switch (e->cfg.max_agents) {
#ifdef CAVOID_DEV_ONLY_N
        CAVOID_RELAY_CASE(4)
#else
        CAVOID_RELAY_CASE(1) CAVOID_RELAY_CASE(2) CAVOID_RELAY_CASE(3) CAVOID_RELAY_CASE(4) CAVOID_RELAY_CASE(5) CAVOID_RELAY_CASE(6)
#endif
        default: return CAVOID_EUNSUPPORTED;
    }
#undef CAVOID_RELAY_CASE
    HIP_TRY(hipGetLastError());
    return CAVOID_OK;
}

#ifdef CAVOID_TRACE
// development build only: this translation unit's copy of the phase-stamp pointer
int cavoid_debug_trace_relay(unsigned long long *dev_ptr) {
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &dev_ptr, sizeof(dev_ptr)));
    return CAVOID_OK;
}
#endif
