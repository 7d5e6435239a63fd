// cavoid_actor_rvo.hip -- actor_kernel<N, true> (cavoid_actor.hpp): the fused actor loop over the env step's ORCA instantiation --
// worlds with scripted RVO agents, and box scenarios (GEN v2) generated inside the step.  Own translation unit (compile time;
// built with -mllvm -disable-machine-licm like cavoid_actor.hip).
#include "cavoid_actor_host.hpp"

using namespace cavoid;

int cavoid_launch_actor_rvo(cavoid_env *e, const SplitArgs &sa, const RolloutCfg &rc, const RolloutState &rs, const RolloutIO &rio, const ActorIO &io,
                            hipStream_t s) {
    return launch_actor_any<true>(e, sa, sa, rc, rs, rio, io, s);
}

int cavoid_launch_step_push_rvo(cavoid_env *e, const RolloutCfg &rc, const RolloutState &rs, const RolloutIO &rio, const ActorIO &io, int32_t step,
                                hipStream_t s) {
    return launch_step_push_any<true>(e, rc, rs, rio, io, step, s);
}
