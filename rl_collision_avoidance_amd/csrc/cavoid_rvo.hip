// cavoid_rvo.hip -- the env_kernel instantiations that can drive RVO (ORCA) scripted agents (cfg.rvo_enabled): every
// stepping mode once more with the linear programmes compiled in.  Own translation unit: the other configurations keep
// their register budget, and the three env-kernel units compile side by side.  Built with -mllvm -disable-machine-licm
// like cavoid_multistep.hip (it holds step-loop instantiations too).
#include "cavoid_launch.hpp"

using namespace cavoid;

int cavoid_launch_rvo(cavoid_env *e, int mode, const KIO &io, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (mode == MODE_STEP_AUTORESET_PF && e->pipeline && !io.cont) {       // (the pipelined form decodes table actions only)
        const int rc = launch_pipe<true>(e, io, s, ev_start, ev_stop);
        if (rc != CAVOID_EUNSUPPORTED) return rc;
    }
    switch (mode) {
        case MODE_STEP: return launch_on<MODE_STEP, true>(e, e->k, e->st, e->grid, io, s, ev_start, ev_stop);
        case MODE_STEP_AUTORESET: return launch_on<MODE_STEP_AUTORESET, true>(e, e->k, e->st, e->grid, io, s, ev_start, ev_stop);
        case MODE_STEP_AUTORESET_PF: return launch_on<MODE_STEP_AUTORESET_PF, true>(e, e->k, e->st, e->grid, io, s, ev_start, ev_stop);
        case MODE_STEP_AUTORESET_N: return launch_on<MODE_STEP_AUTORESET_N, true>(e, e->k, e->st, e->grid, io, s, ev_start, ev_stop);
        default: return CAVOID_EINVAL;
    }
}

#ifdef CAVOID_TRACE
int cavoid_debug_trace_rvo(unsigned long long *dev_ptr) {
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &dev_ptr, sizeof(dev_ptr)));
    return CAVOID_OK;
}
#endif
