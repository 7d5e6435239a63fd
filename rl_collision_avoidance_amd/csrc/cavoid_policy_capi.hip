// cavoid_policy_capi.hip -- C ABI (include/cavoid.h, cavoid_policy_*) over the fused NetworkVP_rnn
// inference kernel of cavoid_policy.hpp.
#include <hip/hip_runtime.h>

#include <new>

#include "cavoid.h"
#include "cavoid_host.hpp"
#define CAVOID_POLICY_KERNELS 1
#include "cavoid_policy.hpp"
#include <cstring>
#include "cavoid_policy_split.hpp"
#include "cavoid_policy_split8.hpp"
#include "cavoid_policy_host.hpp"

#include <cstdlib>

using namespace cavoid;

extern "C" int cavoid_policy_create(int32_t max_other, int32_t num_actions, int device, cavoid_policy **out) {
    if (!out) return CAVOID_EINVAL;
    *out = nullptr;
    if (max_other < 1 || max_other > kPolMaxOthers || num_actions < 1 || num_actions > 15) return CAVOID_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return CAVOID_ENODEVICE;
    HIP_TRY(hipSetDevice(device));
    cavoid_policy *h = new (std::nothrow) cavoid_policy();
    if (!h) return CAVOID_ENOMEM;
    h->device = device; h->max_other = max_other; h->num_actions = num_actions;
    h->in_size = 1 + kPolHost + kPolOther * max_other;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_frag = carve((size_t)kPackFragsTrain * sizeof(f32x4)), o_bias = carve(kBiasFloats * sizeof(float));
    const size_t o_sfrag = carve((size_t)kSpPackFrags8 * sizeof(uint4)), o_sbias = carve(kBiasFloats8 * sizeof(float));
    const size_t o_avg = carve(h->in_size * sizeof(float)), o_std = carve(h->in_size * sizeof(float));
    const size_t o_step = carve(sizeof(int32_t)), o_done = carve(sizeof(uint32_t)), o_tick = carve(kPolCuSlots * sizeof(uint32_t));
    const size_t o_clamp = carve(sizeof(uint32_t));
    if (hipMalloc(&h->slab, off) != hipSuccess) { delete h; return CAVOID_ENOMEM; }
    if (hipMemset(h->slab, 0, off) != hipSuccess) { g_last_hip_error = (int)hipGetLastError(); (void)hipFree(h->slab); delete h; return CAVOID_EHIP; }
    unsigned char *b = static_cast<unsigned char *>(h->slab);
    h->frags = reinterpret_cast<f32x4 *>(b + o_frag); h->bias = reinterpret_cast<float *>(b + o_bias);
    h->sfrags = reinterpret_cast<uint4 *>(b + o_sfrag); h->sbias = reinterpret_cast<float *>(b + o_sbias);
    if (const char *ov = std::getenv("CAVOID_POLICY_F32")) h->use_split = std::atoi(ov) == 0;
    // 16 (default): two float16 pieces per operand, three products -- float32-grade; 3 / 4 / 5: bf16 pieces, that many products (A/B runs)
    if (const char *ov = std::getenv("CAVOID_POLICY_PRODUCTS")) { const int v = std::atoi(ov); if ((v >= 3 && v <= 5) || v == kSpF16) h->split_products = v; }
    // CAVOID_POLICY_FORM: how the stand-alone inference launch maps tiles to wavefronts (bit-identical results, A/B runs) -- quad: four wavefronts per
    // 64-row tile, two independent workgroups per CU; oct: eight wavefronts per tile (cavoid_policy_split8.hpp); duo: two tiles per workgroup, phases
    // locked one barrier apart (policy_forward_split_duo_kernel)
    // Default (-1): duo once the launch has at least two tiles per compute unit (below that a paired workgroup would leave CUs idle), else quad.
    if (const char *ov = std::getenv("CAVOID_POLICY_FORM")) h->form = !std::strcmp(ov, "oct") ? 1 : (!std::strcmp(ov, "duo") ? 2 : (!std::strcmp(ov, "pipe") ? 3 : (!std::strcmp(ov, "quad") ? 0 : -1)));
    if (hipDeviceGetAttribute(&h->num_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || h->num_cus <= 0) h->num_cus = 256;
    h->avg = reinterpret_cast<float *>(b + o_avg); h->std = reinterpret_cast<float *>(b + o_std);
    h->step_counter = reinterpret_cast<int32_t *>(b + o_step); h->blocks_done = reinterpret_cast<uint32_t *>(b + o_done);
    h->cu_tickets = reinterpret_cast<uint32_t *>(b + o_tick);
    h->clamped_weights = reinterpret_cast<uint32_t *>(b + o_clamp);
    // 70 KB of LDS per 64-row workgroup: above the 64 KB static limit, so it is dynamic and opted into here
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(policy_forward_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)policy_lds_bytes(4)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(policy_forward_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)policy_lds_bytes(4)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(policy_backward_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)policy_lds_bytes(4)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(policy_forward_split_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)policy_split_lds_bytes()) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(policy_forward_split_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)policy_split_lds_bytes()) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(policy_forward_split_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)policy_split_lds_bytes()) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(policy_forward_split_kernel<kSpF16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)policy_split_lds_bytes()) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(policy_forward_split_kernel<kSpF16, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)policy_split_lds_bytes()) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(policy_forward_split8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)policy_split_lds_bytes()) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(policy_forward_split_duo_kernel<kSpF16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)policy_split_duo_lds_bytes()) != hipSuccess) {
        g_last_hip_error = (int)hipGetLastError(); (void)hipFree(h->slab); delete h; return CAVOID_EHIP;
    }
    *out = h;
    return CAVOID_OK;
}

extern "C" void cavoid_policy_destroy(cavoid_policy *h) {
    if (!h) return;
    if (h->slab) (void)hipFree(h->slab);
    delete h;
}

extern "C" int cavoid_policy_load(cavoid_policy *h, const cavoid_policy_weights *w, void *stream) {
    if (!h || !w || w->struct_size != (int32_t)sizeof(cavoid_policy_weights)) return CAVOID_EINVAL;
    if (!w->lstm_kernel || !w->lstm_bias || !w->layer1_kernel || !w->layer1_bias || !w->layer2_kernel || !w->layer2_bias ||
        !w->fc1_kernel || !w->fc1_bias || !w->p_kernel || !w->p_bias || !w->v_kernel || !w->v_bias)
        return CAVOID_EINVAL;
    if ((w->avg == nullptr) != (w->std == nullptr)) return CAVOID_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(h->device));
    PolicyWeights k{};
    k.lstm_kernel = w->lstm_kernel; k.lstm_bias = w->lstm_bias; k.layer1_kernel = w->layer1_kernel; k.layer1_bias = w->layer1_bias;
    k.layer2_kernel = w->layer2_kernel; k.layer2_bias = w->layer2_bias; k.fc1_kernel = w->fc1_kernel; k.fc1_bias = w->fc1_bias;
    k.p_kernel = w->p_kernel; k.p_bias = w->p_bias; k.v_kernel = w->v_kernel; k.v_bias = w->v_bias;
    k.num_actions = h->num_actions; k.forget_bias = w->forget_bias;
    const int with_backward = w->with_backward ? 1 : 0;
    const unsigned blocks = (unsigned)(((with_backward ? kPackFragsTrain : kPackFrags) + 255) / 256);
    hipLaunchKernelGGL(policy_pack_kernel, dim3(blocks), dim3(256), 0, s, k, h->frags, h->bias, with_backward);
    h->backward_loaded = with_backward != 0;
    HIP_TRY(hipGetLastError());
    {   // the inference kernel's copy, fragment order: every weight split into two float16 pieces (22 bits; the default form) or
        // three bf16 pieces (exact; CAVOID_POLICY_PRODUCTS = 3 / 4 / 5)
        constexpr int64_t items = kSpOffHead / 3 + kSpChWide * 64;
        HIP_TRY(hipMemsetAsync(h->clamped_weights, 0, sizeof(uint32_t), s));
        hipLaunchKernelGGL(policy_pack_split_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, k, h->sfrags, h->sbias,
                           h->split_products == kSpF16 ? 1 : 0, h->clamped_weights);
        HIP_TRY(hipGetLastError());
        if (h->split_products == kSpF16) {                  // the LSTM once more in the eight-wavefront form's column order
            hipLaunchKernelGGL(policy_pack_split8_kernel, dim3((unsigned)((kSpChLstm * 16 * 64 + 255) / 256)), dim3(256), 0, s, k, h->sfrags, h->sbias);
            HIP_TRY(hipGetLastError());
        }
    }
    h->normalize = w->avg != nullptr;
    if (h->normalize) {
        HIP_TRY(hipMemcpyAsync(h->avg, w->avg, h->in_size * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(h->std, w->std, h->in_size * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    h->min_policy = w->min_policy;
    h->loaded = true;
    return CAVOID_OK;
}

extern "C" int cavoid_policy_info(cavoid_policy *h, void *stream, int32_t *use_split, int32_t *split_products, int32_t *clamped_weights) {
    if (!h) return CAVOID_EINVAL;
    if (use_split) *use_split = h->use_split ? 1 : 0;
    if (split_products) *split_products = h->split_products;
    if (clamped_weights) {                                  // (the one host read-back: waits for the load enqueued on `stream`)
        uint32_t n = 0;
        if (h->loaded) {
            hipStream_t s = static_cast<hipStream_t>(stream);
            HIP_TRY(hipMemcpyAsync(&n, h->clamped_weights, sizeof(n), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
        *clamped_weights = (int32_t)(n > 0x7fffffffu ? 0x7fffffffu : n);
    }
    return CAVOID_OK;
}

extern "C" int cavoid_policy_seed(cavoid_policy *h, uint64_t seed, void *stream) {
    if (!h) return CAVOID_EINVAL;
    h->seed = seed;
    HIP_TRY(hipMemsetAsync(h->step_counter, 0, sizeof(int32_t), static_cast<hipStream_t>(stream)));
    return CAVOID_OK;
}

static int policy_forward(cavoid_policy *h, const float *x, int64_t rows, int64_t row_stride, const int32_t *row_index,
                          const int32_t *row_count, float *p_out, float *v_out, int32_t *actions_out, int32_t greedy, void *stream) {
    if (!h || !x || !p_out || !v_out || rows < 0 || row_stride < h->in_size || row_stride > 256) return CAVOID_EINVAL;
    if (!h->loaded) return CAVOID_EINVAL;
    if (rows == 0) return CAVOID_OK;
    PolicyArgs a{};
    a.x = x; a.rows = rows; a.stride = row_stride; a.max_other = h->max_other; a.num_actions = h->num_actions; a.in_size = h->in_size;
    a.avg = h->normalize ? h->avg : nullptr; a.std = h->normalize ? h->std : nullptr;
    a.frags = h->frags; a.bias = h->bias; a.min_policy = h->min_policy; a.p_out = p_out; a.v_out = v_out;
    a.actions_out = actions_out; a.greedy = greedy ? 1 : 0;
    a.seed_lo = (uint32_t)h->seed; a.seed_hi = (uint32_t)(h->seed >> 32);
    a.step_counter = h->step_counter; a.blocks_done = h->blocks_done; a.cu_tickets = h->cu_tickets;
    a.row_index = row_index; a.row_count = row_count;
    const int tile = 16 * h->row_tiles;
    const int64_t blocks = (rows + tile - 1) / tile;
    if (blocks > 0x7fffffffLL) return CAVOID_EINVAL;
    if (h->use_split) {
        SplitArgs sa{a, h->sfrags, h->sbias};
        hipStream_t s = static_cast<hipStream_t>(stream);
        const int form = h->form >= 0 ? h->form : (blocks >= 2 * (int64_t)h->num_cus ? 2 : 0);
        if (h->split_products == kSpF16 && form == 1) hipLaunchKernelGGL(policy_forward_split8_kernel, dim3((unsigned)blocks), dim3(512), policy_split_lds_bytes(), s, sa);
        else if (h->split_products == kSpF16 && form == 3)
            hipLaunchKernelGGL((policy_forward_split_kernel<kSpF16, true>), dim3((unsigned)blocks), dim3(256), policy_split_lds_bytes(), s, sa);
        else if (h->split_products == kSpF16 && form == 2)
            hipLaunchKernelGGL(policy_forward_split_duo_kernel<kSpF16>, dim3((unsigned)((blocks + 1) / 2)), dim3(512), policy_split_duo_lds_bytes(), s, sa);
        else if (h->split_products == kSpF16) hipLaunchKernelGGL(policy_forward_split_kernel<kSpF16>, dim3((unsigned)blocks), dim3(256), policy_split_lds_bytes(), s, sa);
        else if (h->split_products == 5) hipLaunchKernelGGL(policy_forward_split_kernel<5>, dim3((unsigned)blocks), dim3(256), policy_split_lds_bytes(), s, sa);
        else if (h->split_products == 4) hipLaunchKernelGGL(policy_forward_split_kernel<4>, dim3((unsigned)blocks), dim3(256), policy_split_lds_bytes(), s, sa);
        else hipLaunchKernelGGL(policy_forward_split_kernel<3>, dim3((unsigned)blocks), dim3(256), policy_split_lds_bytes(), s, sa);
    } else {
        hipLaunchKernelGGL((policy_forward_kernel<4, false>), dim3((unsigned)blocks), dim3(256), policy_lds_bytes(4), static_cast<hipStream_t>(stream), a);
    }
    HIP_TRY(hipGetLastError());
    return CAVOID_OK;
}

extern "C" int cavoid_policy_forward(cavoid_policy *h, const float *x, int64_t rows, int64_t row_stride, float *p_out, float *v_out,
                                     int32_t *actions_out, int32_t greedy, void *stream) {
    return policy_forward(h, x, rows, row_stride, nullptr, nullptr, p_out, v_out, actions_out, greedy, stream);
}

extern "C" int cavoid_policy_forward_rows(cavoid_policy *h, const float *x, int64_t rows, int64_t row_stride, const int32_t *row_index,
                                          const int32_t *row_count, float *p_out, float *v_out, int32_t *actions_out, int32_t greedy,
                                          void *stream) {
    if (!row_index || !row_count) return CAVOID_EINVAL;
    return policy_forward(h, x, rows, row_stride, row_index, row_count, p_out, v_out, actions_out, greedy, stream);
}

extern "C" int cavoid_policy_train(cavoid_policy *h, const float *x, int64_t rows, int64_t row_stride, const float *y_r,
                                   const int32_t *a_idx, float beta, float log_epsilon, const cavoid_policy_train_buffers *b,
                                   void *stream) {
    if (!h || !x || !y_r || !a_idx || !b || b->struct_size != (int32_t)sizeof(cavoid_policy_train_buffers)) return CAVOID_EINVAL;
    if (!h->loaded || !h->backward_loaded || rows < 0 || row_stride < h->in_size) return CAVOID_EINVAL;
    const int64_t rows64 = (rows + 63) / 64 * 64;
    if (b->capacity_rows < rows64 || b->capacity_rows % 64 != 0 || !b->z1 || !b->z2 || !b->z3 || !b->l1_in || !b->h_in || !b->save || !b->gh || !b->loss ||
        !b->g1 || !b->g2 || !b->g3 || !b->gl || !b->db)
        return CAVOID_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(b->loss, 0, 2 * sizeof(float), s));
    HIP_TRY(hipMemsetAsync(b->db, 0, kBiasFloats * sizeof(float), s));
    if (rows == 0) return CAVOID_OK;
    const int64_t cap = b->capacity_rows;                  // leading dimension (in rows) of the per-step buffers
    PolicyArgs a{};
    a.x = x; a.rows = rows; a.stride = row_stride; a.max_other = h->max_other; a.num_actions = h->num_actions; a.in_size = h->in_size;
    a.avg = h->normalize ? h->avg : nullptr; a.std = h->normalize ? h->std : nullptr;
    a.frags = h->frags; a.bias = h->bias; a.min_policy = h->min_policy; a.cu_tickets = h->cu_tickets;
    a.y_r = y_r; a.a_idx = a_idx; a.beta = beta; a.log_eps = log_epsilon; a.rows64 = cap;
    a.z1 = b->z1; a.z2 = b->z2; a.z3 = b->z3; a.l1_in = b->l1_in; a.h_in = b->h_in; a.save = b->save; a.gh = b->gh; a.loss = b->loss; a.db = b->db;
    // every tile of the buffers is processed (tiles past `rows` carry zero gradients), so that the caller can run its
    // weight-gradient GEMMs over a convenient row count without ever reading stale rows
    const unsigned blocks = (unsigned)(cap / 64);
    hipLaunchKernelGGL((policy_forward_kernel<4, true>), dim3(blocks), dim3(256), policy_lds_bytes(4), s, a);
    HIP_TRY(hipGetLastError());
    PolicyBackArgs k{};
    k.x = x; k.rows = rows; k.stride = row_stride; k.rows64 = cap; k.max_other = h->max_other; k.frags = h->frags;
    k.z1 = b->z1; k.z2 = b->z2; k.z3 = b->z3; k.save = b->save; k.gh = b->gh; k.g1 = b->g1; k.g2 = b->g2; k.g3 = b->g3; k.gl = b->gl; k.db = b->db;
    hipLaunchKernelGGL((policy_backward_kernel<4>), dim3(blocks), dim3(256), policy_lds_bytes(4), s, k);
    HIP_TRY(hipGetLastError());
    return CAVOID_OK;
}

#ifdef CAVOID_TRACE
extern "C" int cavoid_policy_debug_trace(unsigned long long *dev_ptr) {
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_pol_trace), &dev_ptr, sizeof(dev_ptr)));
    return CAVOID_OK;
}
#endif
