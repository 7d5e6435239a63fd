// cavoid_rollout_host.hpp -- the handle behind `cavoid_rollout *` (include/cavoid.h), shared by cavoid_rollout_capi.hip and
// cavoid_actor.hip (the fused actor kernel keeps the same per-slot bookkeeping state).
#pragma once
#include <stddef.h>

#include "cavoid_rollout.hpp"

struct cavoid_rollout {
    int device = 0;
    cavoid::RolloutCfg c{};
    cavoid::RolloutState s{};
    void *slab = nullptr;
    size_t slab_bytes = 0;
};
