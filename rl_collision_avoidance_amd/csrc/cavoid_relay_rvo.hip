// cavoid_relay_rvo.hip -- env_relay_kernel<N, true>: the role-split in-launch step loop for world sets with ORCA (policy 3) agents
// (cavoid_relay.hpp).  Own translation unit: the linear programmes inlined into the state owner's advance are compiled beside the plain
// instantiations, not into them; -mllvm -disable-machine-licm like the other step-loop units (build.py).
#include "cavoid_relay_host.hpp"

using namespace cavoid;

int cavoid_launch_relay_rvo(cavoid_env *e, const KIO &io, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
    return cavoid_relay_launch_impl<true>(e, io, s, ev_start, ev_stop);
}
