// cavoid_rollout_capi.hip -- C ABI (include/cavoid.h, cavoid_rollout_*) over the rollout kernels of
// cavoid_rollout.hpp.  Separate translation unit: it compiles in seconds, the env kernels do not.
#include <hip/hip_runtime.h>

#include <new>

#include "cavoid.h"
#include "cavoid_host.hpp"
#define CAVOID_ROLLOUT_KERNELS 1
#include "cavoid_rollout.hpp"
#include "cavoid_rollout_host.hpp"

using namespace cavoid;

extern "C" int cavoid_rollout_create(int64_t num_worlds, int32_t max_agents, int32_t obs_width, int32_t time_max, double discount,
                                     int32_t reflush_done, int32_t ring_len, int device, cavoid_rollout **out) {
    if (!out) return CAVOID_EINVAL;
    *out = nullptr;
    if (num_worlds < 1 || max_agents < 1 || max_agents > 64 || obs_width < 2 || time_max < 1 || time_max > 254 ||
        ring_len < time_max + 2 || !(discount >= 0.0 && discount <= 1.0))
        return CAVOID_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return CAVOID_ENODEVICE;
    HIP_TRY(hipSetDevice(device));
    cavoid_rollout *r = new (std::nothrow) cavoid_rollout();
    if (!r) return CAVOID_ENOMEM;
    r->device = device;
    RolloutCfg &c = r->c;
    c.num_slots = num_worlds * max_agents; c.max_agents = max_agents; c.obs_width = obs_width; c.time_max = time_max;
    c.reflush_done = reflush_done ? 1 : 0; c.ring_len = ring_len; c.discount = discount;
    const size_t S = (size_t)c.num_slots, W = (size_t)num_worlds;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_len = carve(S), o_since = carve(S), o_tr = carve(S), o_sc = carve(S * sizeof(double));
    const size_t o_er = carve(W * sizeof(double)), o_el = carve(W * sizeof(int32_t)), o_step = carve(sizeof(int32_t));
    if (hipMalloc(&r->slab, off) != hipSuccess) { delete r; return CAVOID_ENOMEM; }
    r->slab_bytes = off;
    if (hipMemset(r->slab, 0, off) != hipSuccess) { g_last_hip_error = (int)hipGetLastError(); cavoid_rollout_destroy(r); return CAVOID_EHIP; }
    unsigned char *b = static_cast<unsigned char *>(r->slab);
    r->s.len = b + o_len; r->s.since_flush = b + o_since; r->s.trained = b + o_tr;
    r->s.score = reinterpret_cast<double *>(b + o_sc);
    r->s.ep_reward = reinterpret_cast<double *>(b + o_er); r->s.ep_length = reinterpret_cast<int32_t *>(b + o_el);
    r->s.step_counter = reinterpret_cast<int32_t *>(b + o_step);
    *out = r;
    return CAVOID_OK;
}

extern "C" void cavoid_rollout_destroy(cavoid_rollout *r) {
    if (!r) return;
    if (r->slab) (void)hipFree(r->slab);
    delete r;
}

extern "C" int cavoid_rollout_reset(cavoid_rollout *r, void *stream) {
    if (!r) return CAVOID_EINVAL;
    HIP_TRY(hipMemsetAsync(r->slab, 0, r->slab_bytes, static_cast<hipStream_t>(stream)));
    return CAVOID_OK;
}

extern "C" int cavoid_rollout_push(cavoid_rollout *r, const float *prev_obs, const int32_t *actions, const float *values,
                                   const float *rewards, const uint8_t *done, const uint8_t *game_over, int32_t step,
                                   float *x, double *val, float *ret, uint8_t *act, int32_t *emit_t,
                                   float *dup_x, float *dup_r, int32_t *dup_a, int32_t *dup_src, int32_t *dup_count,
                                   int64_t dup_capacity, float *ep_out, int32_t *ep_count, int64_t ep_capacity, void *stream) {
    if (!r || !prev_obs || !actions || !values || !rewards || !done || !game_over || !x || !val || !ret || !act ||
        !emit_t || !dup_x || !dup_r || !dup_a || !dup_src || !dup_count || !ep_out || !ep_count || dup_capacity < 1 || ep_capacity < 1)
        return CAVOID_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    RolloutCfg c = r->c;
    c.dup_capacity = dup_capacity; c.ep_capacity = ep_capacity;
    RolloutIO io{};
    io.prev_obs = prev_obs; io.actions = actions; io.values = values; io.rewards = rewards; io.done = done; io.game_over = game_over;
    io.step = step; io.x = x; io.val = val; io.ret = ret; io.act = act; io.emit_t = emit_t;
    io.dup_x = dup_x; io.dup_r = dup_r; io.dup_a = dup_a; io.dup_src = dup_src; io.dup_count = dup_count;
    io.ep_out = ep_out; io.ep_count = ep_count;
    const int64_t W = c.num_slots / c.max_agents;
    hipLaunchKernelGGL(rollout_push_kernel, dim3((unsigned)((c.num_slots + 255) / 256)), dim3(256), 0, s, c, r->s, io);
    hipLaunchKernelGGL(rollout_episode_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, s, c, r->s, io);
    HIP_TRY(hipGetLastError());
    return CAVOID_OK;
}

extern "C" int cavoid_rollout_compact(cavoid_rollout *r, int32_t step_lo, int32_t step_hi, int32_t mark_taken, const float *x,
                                      const float *ret, const uint8_t *act, int32_t *emit_t, float *out_x, float *out_r,
                                      int32_t *out_a, int32_t *out_src, int32_t *out_count, int64_t capacity, void *stream) {
    if (!r || !x || !ret || !act || !emit_t || !out_x || !out_r || !out_a || !out_count || capacity < 0 || step_lo < 0 ||
        step_hi < step_lo || step_hi - step_lo > 65535)
        return CAVOID_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(out_count, 0, 2 * sizeof(int32_t), s));
    if (step_hi == step_lo) return CAVOID_OK;
    CompactArgs a{};
    a.step_lo = step_lo; a.step_hi = step_hi; a.mark_taken = mark_taken ? 1 : 0; a.x = x; a.ret = ret; a.act = act; a.emit_t = emit_t;
    a.out_x = out_x; a.out_r = out_r; a.out_a = out_a; a.out_src = out_src; a.out_count = out_count; a.capacity = capacity;
    if ((int64_t)(r->c.obs_width - 1) * 64 * kCompactSpan > 65535) return CAVOID_EUNSUPPORTED;   // (the copy loop's e / D trick)
    const int per_block = 256 * kCompactSpan;
    const dim3 grid((unsigned)((r->c.num_slots + per_block - 1) / per_block), (unsigned)(step_hi - step_lo));
    hipLaunchKernelGGL(rollout_compact_kernel, grid, dim3(256), 0, s, r->c, a);
    HIP_TRY(hipGetLastError());
    return CAVOID_OK;
}

extern "C" int cavoid_rollout_active_rows(cavoid_rollout *r, const float *obs, const uint8_t *done, const uint8_t *game_over,
                                          int32_t *row_index, int32_t *row_count, void *stream) {
    if (!r || !obs || !done || !game_over || !row_index || !row_count) return CAVOID_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(rollout_zero_kernel, dim3(1), dim3(1), 0, s, row_count);      // (a kernel, not a memset node: see DESIGN)
    hipLaunchKernelGGL(rollout_active_kernel, dim3((unsigned)((r->c.num_slots + 255) / 256)), dim3(256), 0, s, r->c, obs, done,
                       game_over, row_index, row_count);
    HIP_TRY(hipGetLastError());
    return CAVOID_OK;
}
