// cavoid_actor_frozen.hip -- actor_kernel<N, true, true> (cavoid_actor.hpp): the fused actor loop for worlds that hold
// frozen-network agents (scripted policy 4: the GA3C-CADRL agent mechanism) -- the env step's 'everything' instantiation (ORCA agents,
// box scenarios generated inside the step) plus a second forward pass on the frozen network's weights for the tiles that hold such an
// agent.  Own translation unit (compile time; built with -mllvm -disable-machine-licm like cavoid_actor.hip).
#include "cavoid_actor_host.hpp"

using namespace cavoid;

int cavoid_launch_actor_frozen(cavoid_env *e, const SplitArgs &sa, const SplitArgs &fz, const RolloutCfg &rc, const RolloutState &rs, const RolloutIO &rio,
                               const ActorIO &io, hipStream_t s) {
    return launch_actor_any<true, true>(e, sa, fz, rc, rs, rio, io, s);
}
