// cavoid_policy_host.hpp -- the handle behind `cavoid_policy *` (include/cavoid.h), shared by the translation units that launch on it:
// cavoid_policy_capi.hip (inference / trainer pass) and cavoid_actor.hip (the fused actor kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cavoid_policy.hpp"
#include "cavoid_policy_split.hpp"

using cavoid::f32x4;

struct cavoid_policy {
    int device = 0;
    int max_other = 0, num_actions = 0, in_size = 0;
    bool loaded = false, normalize = false, backward_loaded = false;
    float min_policy = 0.0f;
    uint64_t seed = 0;
    void *slab = nullptr;
    f32x4 *frags = nullptr;
    uint4 *sfrags = nullptr;         // split weight fragments of the inference kernel (cavoid_policy_split.hpp)
    float *sbias = nullptr;          // ... and its biases (packed order; the LSTM gates pre-scaled by log2 e / 2 log2 e like their weight columns)
    int split_products = cavoid::kSpDefaultProducts;   // 16 (default): float16 pieces, three products; 3 / 4 / 5: bf16 pieces (CAVOID_POLICY_PRODUCTS)
    int form = -1;                   // CAVOID_POLICY_FORM = quad (0) / oct (1) / duo (2) / unset (-1: duo from two tiles per CU on, else quad): the stand-alone inference
                                     // launch's tile-to-wavefront mapping (same results in every form)
    int num_cus = 256;
    bool use_split = true;           // CAVOID_POLICY_F32=1: run inference on the float32-MFMA kernel instead (A/B runs)
    float *bias = nullptr, *avg = nullptr, *std = nullptr;
    int32_t *step_counter = nullptr;
    uint32_t *clamped_weights = nullptr;   // device counter: weights the float16 split saturated at the last cavoid_policy_load
    uint32_t *blocks_done = nullptr, *cu_tickets = nullptr;
    int row_tiles = 4;               // 16-row tiles per workgroup (64 rows, 2 workgroups per CU); the 32-row / 4-per-CU
                                     // instantiation was measured and dropped: 175 vs 132 us
};
