// Does a packed-float32 fma that reads its third operand with the halves SWAPPED (op_sel:[0,0,1] op_sel_hi:[1,0,0]) right behind
// the v_pk_mul_f32 that made that operand stay exact when the SIMD's other wavefront streams MFMAs?  (development aid; the pattern
// the compiler chose for the env step's two velocity dot products)  One workgroup of 512 threads per
// CU: wavefronts w and w+4 share a SIMD.  Wavefronts 4..7 run
//      m  = {vx, vy} * {-py, py}                  v_pk_mul_f32 ... op_sel_hi:[1,0] neg_lo:[0,1]
//      (FILL independent float64 instructions)
//      r  = {vx, vy} * {px, px} + {m.hi, m.lo}    v_pk_fma_f32 ... op_sel:[0,0,1] op_sel_hi:[1,0,0]
// on changing inputs and compare both lanes with the same two dot products made by scalar v_mul_f32 / v_fma_f32; wavefronts 0..3
// are idle or stream v_mfma_f32_16x16x32_f16.  Prints mismatches per lane half for every (partner, FILL).
// Result on MI355X (profiles/r04_pk_opsel_hazard.txt): exact in every configuration -- the bare pair is not what trips inside the
// fused actor kernel; the context is.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float mfma_stream(int iters, float a, float b) {
    f32x4 acc[16];
    for (int k = 0; k < 16; ++k) acc[k] = f32x4{a, a, a, a};
    f16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (_Float16)a; y[e] = (_Float16)b; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[k], 0, 0, 0);
    float s = 0;
    for (int k = 0; k < 16; ++k) s += acc[k][0];
    return s;
}

template <int FILL>
__device__ __forceinline__ void probe(int iters, unsigned seed, unsigned long long *bad_lo, unsigned long long *bad_hi) {
    unsigned s = seed * 2654435761u + 12345u;
    double d0 = 1.25 + seed, d1 = 0.75, d2 = 1.0000001;
    unsigned lo = 0, hi = 0;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u; const float vx = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
        s = s * 1664525u + 1013904223u; const float vy = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
        s = s * 1664525u + 1013904223u; const float px = (float)(int)(s >> 8) * (1.0f / 16777216.0f);
        const float py = __builtin_sqrtf(1.0f - px * px);
        f32x2 v = {vx, vy}, pyy = {py, 123.0f}, pxx = {px, -77.0f}, m, r;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(m) : "v"(v), "v"(pyy));
        if (FILL >= 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(d1));
        if (FILL >= 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d1) : "v"(d2));
        if (FILL >= 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d2) : "v"(d2));
        if (FILL >= 4) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(d2));
        if (FILL >= 5) asm volatile("s_nop 0\n\tv_add_f64 %0, %0, %1" : "+v"(d1) : "v"(d0));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(v), "v"(pxx), "v"(m));
        float t0, t1, e0, e1;                                   // the same values by scalar instructions
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(vy), "v"(py));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(vx), "v"(py));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(vx), "v"(px), "v"(t0));
        asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(e1) : "v"(vy), "v"(px), "v"(t1));
        lo += __float_as_uint(r.x) != __float_as_uint(e0);
        hi += __float_as_uint(r.y) != __float_as_uint(e1);
    }
    if (d0 == 0.123 && d1 == d2) lo += 1;                       // (keep the fillers alive)
    if (lo) atomicAdd(bad_lo, (unsigned long long)lo);
    if (hi) atomicAdd(bad_hi, (unsigned long long)hi);
}

template <int FILL, bool WITH_MFMA>
__global__ void __launch_bounds__(512) pair(float *out, unsigned long long *bad, int iters) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) { if (WITH_MFMA) out[blockIdx.x * 512 + threadIdx.x] = mfma_stream(iters / 2, 1.0f + threadIdx.x, 0.5f); }
    else probe<FILL>(iters, blockIdx.x * 512 + threadIdx.x, bad, bad + 1);
}

template <int FILL>
void run(float *out, unsigned long long *bad, int iters) {
    unsigned long long h[2];
    for (int with = 0; with < 2; ++with) {
        (void)hipMemset(bad, 0, 16);
        if (with) pair<FILL, true><<<256, 512>>>(out, bad, iters); else pair<FILL, false><<<256, 512>>>(out, bad, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
        std::printf("fill %d  partner %-5s  low-lane mismatches %10llu  high-lane mismatches %10llu  of %lld\n", FILL, with ? "mfma" : "idle", h[0], h[1],
                    256ll * 256 * iters);
    }
}

int main() {
    float *out; unsigned long long *bad;
    if (hipMalloc(&out, 256 * 512 * 4) != hipSuccess || hipMalloc(&bad, 16) != hipSuccess) return 2;
    const int iters = 20000;
    run<0>(out, bad, iters); run<1>(out, bad, iters); run<2>(out, bad, iters); run<3>(out, bad, iters); run<4>(out, bad, iters); run<5>(out, bad, iters);
    return 0;
}
