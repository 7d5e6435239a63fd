// cavoid_multistep.hip -- the env_kernel instantiations that take several auto-reset steps inside ONE launch
// (cavoid_step_autoreset_n / cavoid_step_autoreset_packed with n_steps > 1): MODE_STEP_AUTORESET_PF (small batches: the next
// scenario-pool record of every lane is held in registers) and MODE_STEP_AUTORESET_N (restarts gather on demand).
// Own translation unit because it is compiled with -mllvm -disable-machine-licm (build.py): MachineLICM would hoist every
// constant materialisation of the step body out of the step loop into registers live across it.
#include "cavoid_launch.hpp"

using namespace cavoid;

int cavoid_launch_multistep(cavoid_env *e, const KIO &io, bool prefetch, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (e->k.rvo_enabled || (e->k.gen_mode == 1 && e->k.pool_size <= 0))       // ORCA agents / in-step box generator: cavoid_rvo.hip
        return cavoid_launch_rvo(e, prefetch ? MODE_STEP_AUTORESET_PF : MODE_STEP_AUTORESET_N, io, s, ev_start, ev_stop);
    // (continuous actions: the role-split and pipelined forms decode table actions only -- the single-wavefront loops carry them)
    if (prefetch && e->pipeline >= 2 && !io.cont) {                     // small batch: the step cut into roles on several wavefronts (env_relay_kernel)
        const int rc = cavoid_launch_relay(e, io, s, ev_start, ev_stop);
        if (rc != CAVOID_EUNSUPPORTED) return rc;
    }
    if (prefetch && e->pipeline && !io.cont) {                          // two wavefronts per tile, pipelined (env_pipe_kernel)
        const int rc = launch_pipe<false>(e, io, s, ev_start, ev_stop);
        if (rc != CAVOID_EUNSUPPORTED) return rc;
    }
    if (prefetch) return launch_on<MODE_STEP_AUTORESET_PF>(e, e->k, e->st, e->grid, io, s, ev_start, ev_stop);
    return launch_on<MODE_STEP_AUTORESET_N>(e, e->k, e->st, e->grid, io, s, ev_start, ev_stop);
}

#ifdef CAVOID_TRACE
// development build only: this translation unit's copy of the phase-stamp pointer
int cavoid_debug_trace_multistep(unsigned long long *dev_ptr) {
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &dev_ptr, sizeof(dev_ptr)));
    return CAVOID_OK;
}
#endif
