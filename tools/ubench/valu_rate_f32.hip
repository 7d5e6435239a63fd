// Issue cost of the float32 / packed / conversion vector instructions the policy kernel's epilogues are made of, on this GPU
// (development aid; companion of valu_rate.hip): one wavefront per workgroup, one workgroup per CU, each instruction in 8
// independent chains, 64 per loop body; shader clocks per INSTRUCTION (inline asm, so that the compiler cannot re-pack them).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CHAINS 8
#define KERNEL(name, DECL, OPS)                                                                   \
    __global__ void name(float *out, long long *cyc, int iters, float a, float b) {              \
        DECL;                                                                                     \
        const long long t0 = clock64();                                                           \
        for (int it = 0; it < iters; ++it) {                                                      \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) { _Pragma("unroll") for (int k = 0; k < CHAINS; ++k) { OPS; } } \
        }                                                                                         \
        const long long t1 = clock64();                                                           \
        float acc = 0.f; for (int k = 0; k < CHAINS; ++k) acc += FOLD;                            \
        out[blockIdx.x * 64 + threadIdx.x] = acc; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0; }

#define FOLD r[k]
KERNEL(k_fma, float r[CHAINS]; for (int k = 0; k < CHAINS; ++k) r[k] = a + k, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(b), "v"(a)))
KERNEL(k_mul, float r[CHAINS]; for (int k = 0; k < CHAINS; ++k) r[k] = a + k, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[k]) : "v"(b)))
KERNEL(k_add, float r[CHAINS]; for (int k = 0; k < CHAINS; ++k) r[k] = a + k, asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[k]) : "v"(b)))
KERNEL(k_med3, float r[CHAINS]; for (int k = 0; k < CHAINS; ++k) r[k] = a + k, asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(b), "v"(a)))
KERNEL(k_maxi, float r[CHAINS]; for (int k = 0; k < CHAINS; ++k) r[k] = a + k, asm volatile("v_max_i32 %0, %0, %1" : "+v"(r[k]) : "v"(b)))
KERNEL(k_exp, float r[CHAINS]; for (int k = 0; k < CHAINS; ++k) r[k] = a + k, asm volatile("v_exp_f32 %0, %0" : "+v"(r[k])))
KERNEL(k_rcp, float r[CHAINS]; for (int k = 0; k < CHAINS; ++k) r[k] = a + k, asm volatile("v_rcp_f32 %0, %0" : "+v"(r[k])))
KERNEL(k_cvt_f32_f16, float r[CHAINS]; for (int k = 0; k < CHAINS; ++k) r[k] = a + k, asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(r[k])))
KERNEL(k_cvt_pk_f16, float r[CHAINS]; for (int k = 0; k < CHAINS; ++k) r[k] = a + k, asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(r[k]) : "v"(b)))
KERNEL(k_cvt_pk_bf16, float r[CHAINS]; for (int k = 0; k < CHAINS; ++k) r[k] = a + k, asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[k]) : "v"(b)))
KERNEL(k_fma_mix, float r[CHAINS]; for (int k = 0; k < CHAINS; ++k) r[k] = a + k, asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(r[k]) : "v"(b), "v"(a)))
KERNEL(k_cndmask, float r[CHAINS]; for (int k = 0; k < CHAINS; ++k) r[k] = a + k, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[k]) : "v"(b)))
#undef FOLD
#define FOLD (r[k][0] + r[k][1])
__device__ inline f32x2 mk2(float x, float y) { f32x2 v; v[0] = x; v[1] = y; return v; }
KERNEL(k_pk_fma, f32x2 r[CHAINS]; f32x2 bb = mk2(b, b); f32x2 aa = mk2(a, a); for (int k = 0; k < CHAINS; ++k) r[k] = mk2(a + k, a - k),
       asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(bb), "v"(aa)))
KERNEL(k_pk_mul, f32x2 r[CHAINS]; f32x2 bb = mk2(b, b); for (int k = 0; k < CHAINS; ++k) r[k] = mk2(a + k, a - k),
       asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[k]) : "v"(bb)))
KERNEL(k_pk_add, f32x2 r[CHAINS]; f32x2 bb = mk2(b, b); for (int k = 0; k < CHAINS; ++k) r[k] = mk2(a + k, a - k),
       asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[k]) : "v"(bb)))

int main() {
    float *out; long long *cyc; hipMalloc(&out, 256 * 64 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000; long long h[256];
#define RUN(name) do { name<<<256, 64>>>(out, cyc, iters, 1.25f, 1.0000001f); hipDeviceSynchronize(); name<<<256, 64>>>(out, cyc, iters, 1.25f, 1.0000001f); hipDeviceSynchronize(); \
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost); double s = 0; for (int i = 0; i < 256; ++i) s += h[i]; \
    std::printf("%-16s %6.2f clocks per instruction\n", #name, s / 256 / iters / 64); } while (0)
    RUN(k_fma); RUN(k_mul); RUN(k_add); RUN(k_med3); RUN(k_maxi); RUN(k_exp); RUN(k_rcp); RUN(k_cvt_f32_f16); RUN(k_cvt_pk_f16); RUN(k_cvt_pk_bf16);
    RUN(k_fma_mix); RUN(k_cndmask); RUN(k_pk_fma); RUN(k_pk_mul); RUN(k_pk_add);
    return 0;
}
