// cavoid_relay.hip -- env_relay_kernel instantiations (cavoid_relay.hpp): the in-launch step loop of small batches with the
// step cut into roles on several wavefronts of one workgroup per tile.  Own translation unit, compiled with
// -mllvm -disable-machine-licm like the other step-loop units (build.py).
#include "cavoid_relay_host.hpp"

using namespace cavoid;

int cavoid_launch_relay(cavoid_env *e, const KIO &io, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (e->k.rvo_enabled) return cavoid_launch_relay_rvo(e, io, s, ev_start, ev_stop);   // ORCA agents: the instantiations of cavoid_relay_rvo.hip
    return cavoid_relay_launch_impl<false>(e, io, s, ev_start, ev_stop);
}

#ifdef CAVOID_TRACE
// development build only: this translation unit's copy of the phase-stamp pointer
int cavoid_debug_trace_relay(unsigned long long *dev_ptr) {
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &dev_ptr, sizeof(dev_ptr)));
    return CAVOID_OK;
}
#endif
