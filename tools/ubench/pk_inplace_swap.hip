// v_pk_fma_f32 D, A, B, D op_sel:[0,0,1] -- a packed float32 fma whose DESTINATION pair is also its third source pair, read with the
// halves SWAPPED (low result = a.lo * b + D.hi, high result = a.hi * b - D.lo): is it exact?  This is what the compiler made of the env
// step's velocity dot products for neighbour slots 1 and 2 inside the fused actor kernel; round 4's probe
// (pk_opsel_hazard.hip) only ever wrote the result to a THIRD pair.  Wavefronts 4..7 of a 512-thread workgroup run the pair on
// changing inputs and check both halves against scalar v_mul_f32 / v_fma_f32; wavefronts 0..3 (their SIMD partners) idle or stream MFMAs.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pk_inplace tools/ubench/pk_inplace_swap.hip && /tmp/pk_inplace
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool INPLACE, bool MFMA>
__global__ void __launch_bounds__(512) probe(float *out, unsigned long long *bad, int iters) {
    if (threadIdx.x < 256) {                                            // the SIMD partners
        f32x4 acc = {1, 2, 3, 4}; f16x8 x, y;
        for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(0.5f + threadIdx.x); y[e] = (_Float16)0.25f; }
        if (MFMA) for (int it = 0; it < 8 * iters; ++it) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc, 0, 0, 0);
        out[blockIdx.x * 256 + threadIdx.x] = acc[0];
        return;
    }
    unsigned s = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u, lo = 0, hi = 0;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u; const float vx = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
        s = s * 1664525u + 1013904223u; const float vy = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
        s = s * 1664525u + 1013904223u; const float px = (float)(int)(s >> 8) * (1.0f / 16777216.0f), py = __builtin_sqrtf(1.0f - px * px);
        f32x2 v = {vx, vy}, pyy = {py, py}, pxx = {px, px}, m, r;
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(m) : "v"(v), "v"(pyy));                     // (vx py, vy py)
        if (INPLACE) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_hi:[0,0,1]" : "+v"(m) : "v"(v), "v"(pxx)); r = m; }
        else asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_hi:[0,0,1]" : "=&v"(r) : "v"(v), "v"(pxx), "v"(m));
        float t0, t1, e0, e1;                                           // the same two dot products by scalar instructions
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(vy), "v"(py));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(vx), "v"(py));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(vx), "v"(px), "v"(t0));
        asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(e1) : "v"(vy), "v"(px), "v"(t1));
        lo += __float_as_uint(r.x) != __float_as_uint(e0);
        hi += __float_as_uint(r.y) != __float_as_uint(e1);
    }
    if (lo) atomicAdd(bad, (unsigned long long)lo);
    if (hi) atomicAdd(bad + 1, (unsigned long long)hi);
}

template <bool INPLACE, bool MFMA>
void run(float *out, unsigned long long *bad, int iters) {
    unsigned long long h[2];
    (void)hipMemset(bad, 0, 16);
    probe<INPLACE, MFMA><<<512, 512>>>(out, bad, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    std::printf("%-12s partner %-5s  low-half mismatches %10llu  high-half mismatches %10llu  of %lld\n", INPLACE ? "in place" : "third pair",
                MFMA ? "mfma" : "idle", h[0], h[1], 512ll * 256 * iters);
}

int main() {
    float *out; unsigned long long *bad;
    if (hipMalloc(&out, 512 * 256 * 4) != hipSuccess || hipMalloc(&bad, 16) != hipSuccess) return 2;
    for (int rep = 0; rep < 3; ++rep) { run<false, false>(out, bad, 20000); run<false, true>(out, bad, 20000); run<true, false>(out, bad, 20000); run<true, true>(out, bad, 20000); }
    return 0;
}
