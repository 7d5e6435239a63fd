// Round 6: how much vector work fits BETWEEN the matrix instructions of ONE wavefront for free?  mfma32_valu_overlap.hip showed two plain
// v_fma_f32 per v_mfma_f32_16x16x32_f16 cost nothing.  The policy pass's vector phases are not plain fmas: the LSTM cell update is a third
// transcendentals (v_exp_f32 / v_rcp_f32, quarter rate), the epilogues are v_med3 / v_cvt_pk_f16_f32 / v_fma_mix_f32 + 8-byte LDS stores.
// Each stream below issues, per matrix instruction, a fixed filler mix from INDEPENDENT registers (the matrix stream uses 16 accumulators:
// no dependent issue).  One wavefront per SIMD (256 threads per CU), and the same with two (512): the second models the partner tile.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// fillers per matrix instruction: NF plain fmas, NT transcendentals (alternating exp / rcp), NC epilogue groups (med3 + cvt_pk + fma_mix),
// and one ds_write_b64 every WD-th matrix instruction (0: none)
template <int NF, int NT, int NC, int WD>
__global__ void __launch_bounds__(512) fill(float *out, long long *cyc, int iters, float a, float b) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 528];
    f32x4 acc[16];
    for (int k = 0; k < 16; ++k) acc[k] = f32x4{a, a, a, a};
    f16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (_Float16)a; y[e] = (_Float16)b; }
    float r[8], t[4], c[4];
    for (int k = 0; k < 8; ++k) r[k] = a + k;
    for (int k = 0; k < 4; ++k) { t[k] = 0.5f + 0.1f * k; c[k] = a + k; }
    unsigned char *p = lds + (threadIdx.x & 63) * 528 + ((threadIdx.x >> 6) & 7) * 64;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[k], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < NF; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[(k + f) & 7]) : "v"(b), "v"(a));
#pragma unroll
                for (int f = 0; f < NT; ++f) {
                    if ((k + f) & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(t[(k + f) & 3]));
                    else asm volatile("v_rcp_f32 %0, %0" : "+v"(t[(k + f) & 3]));
                }
#pragma unroll
                for (int f = 0; f < NC; ++f) {
                    asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(c[f & 3]) : "v"(b));
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(c[(f + 1) & 3]) : "v"(c[f & 3]), "v"(b));
                    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(c[(f + 2) & 3]) : "v"(c[(f + 1) & 3]), "v"(b));
                }
                if (WD && (k % WD) == 0) asm volatile("ds_write_b64 %0, %1" :: "v"((unsigned)(size_t)p), "v"(*reinterpret_cast<double *>(&r[0])) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int k = 0; k < 8; ++k) s += r[k];
    for (int k = 0; k < 4; ++k) s += t[k] + c[k];
    for (int k = 0; k < 16; ++k) s += acc[k][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NF, int NT, int NC, int WD>
void run(float *out, long long *cyc, int iters) {
    static long long h[256 * 8];
    for (int threads = 256; threads <= 512; threads += 256) {
        for (int rep = 0; rep < 2; ++rep) { fill<NF, NT, NC, WD><<<256, threads>>>(out, cyc, iters, 1.25f, 1.0000001f); hipDeviceSynchronize(); }
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0;
        const int waves = threads / 64;
        for (int i = 0; i < 256; ++i) for (int w = 0; w < waves; ++w) s += h[i * 8 + w];
        std::printf("per matrix instruction: %d fma + %d exp/rcp + %d x (med3, cvt_pk, fma_mix) + ds_write_b64 every %d   %d wavefront(s) per SIMD: %7.2f clocks per matrix instruction of a wavefront\n",
                    NF, NT, NC, WD, waves / 4, s / (256.0 * waves) / iters / 64);
    }
}

int main() {
    float *out; long long *cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int it = 1000;
    run<0, 0, 0, 0>(out, cyc, it);
    run<2, 0, 0, 0>(out, cyc, it); run<3, 0, 0, 0>(out, cyc, it); run<4, 0, 0, 0>(out, cyc, it); run<6, 0, 0, 0>(out, cyc, it);
    run<0, 1, 0, 0>(out, cyc, it); run<0, 2, 0, 0>(out, cyc, it);
    run<2, 1, 0, 0>(out, cyc, it); run<3, 1, 0, 0>(out, cyc, it);           // the LSTM cell update's mix: ~0.7 transcendentals + 2.5 plain per matrix instruction
    run<0, 0, 1, 0>(out, cyc, it); run<0, 0, 1, 4>(out, cyc, it); run<0, 0, 2, 2>(out, cyc, it);   // the relu + split epilogue: 0.5 groups per matrix instruction in the wide layers
    run<2, 0, 0, 4>(out, cyc, it);
    return 0;
}
