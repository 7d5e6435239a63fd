// cavoid_quad.hip -- env_quad_kernel instantiations (cavoid_quad.hpp): one auto-reset step per launch with four cooperating wavefronts per
// tile -- the closed-loop `env.step` of small batches, where a step is one wavefront's dependent chain per tile and most SIMDs idle.
#include <cstdlib>

#include "cavoid_launch.hpp"
#include "cavoid_quad.hpp"

using namespace cavoid;

int cavoid_launch_quad(cavoid_env *e, const KIO &io, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const KCfg &k = e->k;
    if (e->quad == 0) return CAVOID_EUNSUPPORTED;          // CAVOID_QUAD=0
    if (k.rvo_enabled || (k.gen_mode == 1 && k.pool_size <= 0) || !io.obs || !io.actions || io.cont) return CAVOID_EUNSUPPORTED;
    if (k.dynamics == CAVOID_DYN_HOLONOMIC) return CAVOID_EUNSUPPORTED;
    const int64_t tiles = (e->W + k.wpw - 1) / k.wpw;
    // it pays while every tile's four wavefronts are resident at <= 2 per SIMD (1024 SIMDs): beyond that the single-wavefront form's
    // throughput wins (CAVOID_QUAD=1 forces it for A/B runs)
    if (e->quad < 0 && tiles > 512) return CAVOID_EUNSUPPORTED;
    if (k.tile_rows < k.wpw * e->cfg.max_agents) return CAVOID_EUNSUPPORTED;   // one pass per step only
    const int row = io.obs_stride;
    const int tile_floats = (k.tile_rows * row + 3) & ~3;
    const dim3 grid((unsigned)tiles), block(256);
#define CAVOID_QUAD_CASE(NN) \
    case NN: {                                                                                                          \
        const size_t lds = quad_lds_bytes<NN>(tile_floats);                                                             \
        if (lds > 65536) return CAVOID_EUNSUPPORTED;                                                                    \
        if (ev_start || ev_stop)                                                                                        \
            hipExtLaunchKernelGGL((env_quad_kernel<NN>), grid, block, lds, s, ev_start, ev_stop, 0, k, e->st, e->pool, io);  \
        else                                                                                                            \
            hipLaunchKernelGGL((env_quad_kernel<NN>), grid, block, lds, s, k, e->st, e->pool, io);                      \
        break;                                                                                                          \
    }
    switch (e->cfg.max_agents) {
#ifdef CAVOID_DEV_ONLY_N
        CAVOID_QUAD_CASE(4)
#else
        CAVOID_QUAD_CASE(2) CAVOID_QUAD_CASE(3) CAVOID_QUAD_CASE(4) CAVOID_QUAD_CASE(5) CAVOID_QUAD_CASE(6) CAVOID_QUAD_CASE(10)
#endif
        default: return CAVOID_EUNSUPPORTED;
    }
#undef CAVOID_QUAD_CASE
    HIP_TRY(hipGetLastError());
    return CAVOID_OK;
}

#ifdef CAVOID_TRACE
// development build only: this translation unit's copy of the phase-stamp pointer
int cavoid_debug_trace_quad(unsigned long long *dev_ptr) {
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &dev_ptr, sizeof(dev_ptr)));
    return CAVOID_OK;
}
#endif
