// cavoid_host.hpp -- host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#include "cavoid.h"

extern thread_local int g_last_hip_error;      // raw hipError_t behind the last CAVOID_EHIP (cavoid_capi.hip)

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) {                        \
            g_last_hip_error = (int)_e;                \
            return CAVOID_EHIP;                        \
        }                                              \
    } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
