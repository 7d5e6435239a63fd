// cavoid_actor_host.hpp -- launch of actor_kernel<N, RVO, FROZEN> (cavoid_actor.hpp), shared by the two translation units that instantiate
// it: cavoid_actor.hip (RVO = false) and cavoid_actor_rvo.hip (RVO = true: ORCA agents, box scenarios generated inside the step).
#pragma once
#include <hip/hip_runtime.h>

#include "cavoid.h"
#include "cavoid_actor.hpp"
#include "cavoid_host.hpp"
#include "cavoid_launch.hpp"

namespace cavoid {

template <int N, bool RVO, bool FROZEN = false>
static int launch_actor(cavoid_env *e, const SplitArgs &sa, const SplitArgs &fz, const RolloutCfg &rc, const RolloutState &rs, const RolloutIO &rio, const ActorIO &io,
                        hipStream_t s) {
    // > 64 KiB of dynamic LDS: opted into once per instantiation AND device (the attribute belongs to the function on the current
    // device; a process may drive several).  The flags are only ever set, and setting the attribute twice is harmless: no lock.
    static bool opted_in[64] = {};
    const int dev = e->device;
    if (dev < 0 || dev >= 64 || !opted_in[dev]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(actor_kernel<N, RVO, FROZEN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)policy_split_lds_bytes()));
        if (dev >= 0 && dev < 64) opted_in[dev] = true;
    }
    const int64_t tiles = (e->W + e->k.wpw - 1) / e->k.wpw;
    KCfg k = e->k;
    k.stream_obs = 0;                                        // (the tile's own policy phase reads the rows next: they stay in the L2)
    hipLaunchKernelGGL((actor_kernel<N, RVO, FROZEN>), dim3((unsigned)tiles), dim3(256), policy_split_lds_bytes(), s, k, e->st, e->pool, sa, fz, rc, rs, rio, io);
    HIP_TRY(hipGetLastError());
    return CAVOID_OK;
}

template <bool RVO, bool FROZEN = false>
static int launch_actor_any(cavoid_env *e, const SplitArgs &sa, const SplitArgs &fz, const RolloutCfg &rc, const RolloutState &rs, const RolloutIO &rio, const ActorIO &io,
                            hipStream_t s) {
#define CAVOID_ACTOR_CASE(NN) case NN: return launch_actor<NN, RVO, FROZEN>(e, sa, fz, rc, rs, rio, io, s);
    switch (e->cfg.max_agents) {
#ifdef CAVOID_DEV_ONLY_N
        CAVOID_ACTOR_CASE(4) CAVOID_ACTOR_CASE(10)
#else
        CAVOID_ACTOR_CASE(1) CAVOID_ACTOR_CASE(2) CAVOID_ACTOR_CASE(3) CAVOID_ACTOR_CASE(4) CAVOID_ACTOR_CASE(5) CAVOID_ACTOR_CASE(6)
        CAVOID_ACTOR_CASE(7) CAVOID_ACTOR_CASE(8) CAVOID_ACTOR_CASE(9) CAVOID_ACTOR_CASE(10) CAVOID_ACTOR_CASE(11) CAVOID_ACTOR_CASE(12)
        CAVOID_ACTOR_CASE(13) CAVOID_ACTOR_CASE(14) CAVOID_ACTOR_CASE(15) CAVOID_ACTOR_CASE(16)
#endif
        default: break;
    }
#undef CAVOID_ACTOR_CASE
    return CAVOID_EUNSUPPORTED;
}

template <int N, bool RVO>
static int launch_step_push(cavoid_env *e, const RolloutCfg &rc, const RolloutState &rs, const RolloutIO &rio, const ActorIO &io, int32_t step, hipStream_t s) {
    const KCfg &k = e->k;
    int tile = (k.tile_rows * k.width + 3) & ~3;
    if (tile < k.park_floats) tile = k.park_floats;
    const size_t lds = (size_t)(lds_floats_block() + e->waves_per_block * (lds_floats_fixed(e->cfg.max_agents) + k.rvo_lds_floats + tile)) * sizeof(float);
    hipLaunchKernelGGL((step_push_kernel<N, RVO>), dim3((unsigned)e->grid, 2u), dim3(64 * e->waves_per_block), lds, s, k, e->st, e->pool, rc, rs, rio, io, step);
    HIP_TRY(hipGetLastError());
    return CAVOID_OK;
}

template <bool RVO>
static int launch_step_push_any(cavoid_env *e, const RolloutCfg &rc, const RolloutState &rs, const RolloutIO &rio, const ActorIO &io, int32_t step,
                                hipStream_t s) {
#define CAVOID_ACTOR_CASE(NN) case NN: return launch_step_push<NN, RVO>(e, rc, rs, rio, io, step, s);
    switch (e->cfg.max_agents) {
#ifdef CAVOID_DEV_ONLY_N
        CAVOID_ACTOR_CASE(4) CAVOID_ACTOR_CASE(10)
#else
        CAVOID_ACTOR_CASE(1) CAVOID_ACTOR_CASE(2) CAVOID_ACTOR_CASE(3) CAVOID_ACTOR_CASE(4) CAVOID_ACTOR_CASE(5) CAVOID_ACTOR_CASE(6)
        CAVOID_ACTOR_CASE(7) CAVOID_ACTOR_CASE(8) CAVOID_ACTOR_CASE(9) CAVOID_ACTOR_CASE(10) CAVOID_ACTOR_CASE(11) CAVOID_ACTOR_CASE(12)
        CAVOID_ACTOR_CASE(13) CAVOID_ACTOR_CASE(14) CAVOID_ACTOR_CASE(15) CAVOID_ACTOR_CASE(16)
#endif
        default: break;
    }
#undef CAVOID_ACTOR_CASE
    return CAVOID_EUNSUPPORTED;
}

}  // namespace cavoid

// cavoid_actor_rvo.hip
int cavoid_launch_step_push_rvo(cavoid_env *e, const cavoid::RolloutCfg &rc, const cavoid::RolloutState &rs, const cavoid::RolloutIO &rio,
                                const cavoid::ActorIO &io, int32_t step, hipStream_t s);
int cavoid_launch_actor_rvo(cavoid_env *e, const cavoid::SplitArgs &sa, const cavoid::RolloutCfg &rc, const cavoid::RolloutState &rs,
                            const cavoid::RolloutIO &rio, const cavoid::ActorIO &io, hipStream_t s);
// cavoid_actor_frozen.hip: actor_kernel<N, true, true> -- the 'everything' instantiation (ORCA agents, in-step box scenarios) + a
// second, frozen network for the policy-4 agents
int cavoid_launch_actor_frozen(cavoid_env *e, const cavoid::SplitArgs &sa, const cavoid::SplitArgs &fz, const cavoid::RolloutCfg &rc,
                               const cavoid::RolloutState &rs, const cavoid::RolloutIO &rio, const cavoid::ActorIO &io, hipStream_t s);
