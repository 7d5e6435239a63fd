// Round 6 question: the policy pass issues v_mfma_f32_16x16x32_f16 (16 clocks on the matrix pipe), beside which a SIMD hands out ONE vector
// slot per matrix instruction (mfma_valu_overlap.hip): matrix and vector time ADD.  Does the 32x32x16 form (32 clocks, the same flop per
// clock) leave more issue room -- to the SIMD's other wavefront (role B) and to independent vector instructions of the SAME wavefront?
// One workgroup of 512 threads per CU: wavefronts w and w + 4 share a SIMD; role A = wavefronts 0..3, role B = 4..7.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
enum { IDLE = 0, M16 = 1, M32 = 2, FMA = 3, EXP = 4, MIX = 5, F64 = 6, LDSR = 7, CVT = 8,
       M32_F1 = 9, M32_F2 = 10, M32_F3 = 11, M32_F4 = 12, M32_F5 = 13, M32_F6 = 14, M16_F1 = 15, M16_F2 = 16, M32_D2 = 17, M32_D4 = 18,
       PKFMA = 19, MED3 = 20, FMIX = 21, RCP = 22, MOV = 23, PERM = 24, CVTRTZ = 25 };

template <int KIND>
__device__ __forceinline__ float stream(int iters, float a, float b, const unsigned char *lds) {
    if (KIND == M16 || KIND == M16_F1 || KIND == M16_F2) {
        f32x4 acc[16];
        for (int k = 0; k < 16; ++k) acc[k] = f32x4{a, a, a, a};
        f16x8 x, y;
        for (int e = 0; e < 8; ++e) { x[e] = (_Float16)a; y[e] = (_Float16)b; }
        float r[4] = {a, a + 1, a + 2, a + 3};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[k], 0, 0, 0);
                    if (KIND != M16) {
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k & 3]) : "v"(b), "v"(a));
                        if (KIND == M16_F2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[(k + 2) & 3]) : "v"(b), "v"(a));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        }
        float s = r[0] + r[1] + r[2] + r[3]; for (int k = 0; k < 16; ++k) s += acc[k][0];
        return s;                                                            // 64 MFMAs (16 clocks each) per iteration
    } else if (KIND == M32 || (KIND >= M32_F1 && KIND <= M32_F6) || KIND == M32_D2 || KIND == M32_D4) {
        f32x16 acc[4];
        for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = a;
        f16x8 x, y;
        for (int e = 0; e < 8; ++e) { x[e] = (_Float16)a; y[e] = (_Float16)b; }
        float r[6] = {a, a + 1, a + 2, a + 3, a + 4, a + 5};
        double d[4] = {a, a + 1.0, a + 2.0, a + 3.0};
        constexpr int NF = KIND == M32 ? 0 : (KIND >= M32_F1 && KIND <= M32_F6 ? KIND - M32_F1 + 1 : 0);
        constexpr int ND = KIND == M32_D2 ? 2 : (KIND == M32_D4 ? 4 : 0);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[k], 0, 0, 0);
                    if (NF || ND) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int f = 0; f < NF; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[f]) : "v"(b), "v"(a));
#pragma unroll
                        for (int f = 0; f < ND; ++f) d[f] = __builtin_fma(d[f], (double)b, (double)a);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        }
        float s = r[0] + r[1] + r[2] + r[3] + r[4] + r[5] + (float)(d[0] + d[1] + d[2] + d[3]);
        for (int k = 0; k < 4; ++k) s += acc[k][0];
        return s;                                                            // 32 MFMAs (32 clocks each) per iteration
    } else if (KIND == FMA || KIND == EXP || KIND == MIX || KIND == CVT || KIND >= PKFMA) {
        float r[8];
        for (int k = 0; k < 8; ++k) r[k] = a + k;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(b), "v"(a));     // (plain C is packed into v_pk_fma_f32 by the compiler)
                    else if (KIND == PKFMA) r[k] = __builtin_fmaf(r[k], b, a);
                    else if (KIND == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(b), "v"(a));
                    else if (KIND == FMIX) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(r[k]) : "v"(b), "v"(a));
                    else if (KIND == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[k]));
                    else if (KIND == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(r[k]) : "v"(b));
                    else if (KIND == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(b), "v"(a));
                    else if (KIND == EXP) r[k] = __builtin_amdgcn_exp2f(r[k]);
                    else if (KIND == CVT) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(r[k]) : "v"(b));
                    else if (KIND == CVTRTZ) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(r[k]) : "v"(b));
                    else r[k] = (u & 3) == 0 ? __builtin_amdgcn_exp2f(r[k]) : __builtin_fmaf(r[k], b, a);
                }
        }
        float s = 0; for (int k = 0; k < 8; ++k) s += r[k];
        return s;                                                            // 64 vector instructions per iteration
    } else if (KIND == F64) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a + k;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) r[k] = __builtin_fma(r[k], (double)b, (double)a);
        }
        double s = 0; for (int k = 0; k < 8; ++k) s += r[k];
        return (float)s;                                                     // 64 v_fma_f64 per iteration
    } else if (KIND == LDSR) {
        u32x4 s4 = u32x4{0, 0, 0, 0};
        const unsigned char *p = lds + (threadIdx.x & 63) * 528;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) s4 ^= *reinterpret_cast<const u32x4 *>(p + u * 16);
            asm volatile("" ::: "memory");
        }
        return (float)(s4.x ^ s4.y ^ s4.z ^ s4.w);                           // 16 ds_read_b128 per iteration
    }
    return 0.0f;
}

template <int KA, int KB>
__global__ void __launch_bounds__(512) pair(float *out, long long *cyc, int iters, float a, float b) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 528];
    for (int i = threadIdx.x; i < 64 * 528 / 4; i += 512) reinterpret_cast<unsigned *>(lds)[i] = i;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    float r;
    const long long t0 = clock64();
    if (wave < 4) r = stream<KA>(iters, a, b, lds);
    else r = stream<KB>(iters, a, b, lds);
    const long long t1 = clock64();
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

static const char *name(int k) {
    static const char *n[] = {"idle", "mfma16x16x32", "mfma32x32x16", "fma32", "exp32", "mix(1 exp:3 fma)", "fma64", "lds b128", "cvt_pk_f16",
                              "m32 + 1 fma", "m32 + 2 fma", "m32 + 3 fma", "m32 + 4 fma", "m32 + 5 fma", "m32 + 6 fma", "m16 + 1 fma", "m16 + 2 fma",
                              "m32 + 2 fma64", "m32 + 4 fma64", "pk_fma32 (2 vals)", "med3", "fma_mix", "rcp32", "mov", "perm", "cvt_pkrtz_f16"};
    return n[k];
}
static double per(int k) { return k == LDSR ? 16 : ((k == M32 || (k >= M32_F1 && k <= M32_F6) || k == M32_D2 || k == M32_D4) ? 32 : 64); }

template <int KA, int KB>
void run(float *out, long long *cyc, int iters) {
    static long long h[256 * 8];
    for (int rep = 0; rep < 2; ++rep) { pair<KA, KB><<<256, 512>>>(out, cyc, iters, 1.25f, 1.0000001f); hipDeviceSynchronize(); }
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double sa = 0, sb = 0;
    for (int i = 0; i < 256; ++i) for (int w = 0; w < 4; ++w) { sa += h[i * 8 + w]; sb += h[i * 8 + 4 + w]; }
    std::printf("A = %-16s B = %-16s  A %7.2f clocks per (matrix) instruction, B %7.2f per instruction\n", name(KA), name(KB),
                sa / 1024 / iters / per(KA), sb / 1024 / iters / per(KB));
}

int main() {
    float *out; long long *cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int it = 2000;
    run<M16, IDLE>(out, cyc, it); run<M32, IDLE>(out, cyc, it);
    run<IDLE, FMA>(out, cyc, it); run<IDLE, PKFMA>(out, cyc, it); run<IDLE, EXP>(out, cyc, it); run<IDLE, RCP>(out, cyc, it); run<IDLE, MIX>(out, cyc, it); run<IDLE, F64>(out, cyc, it);
    run<IDLE, LDSR>(out, cyc, it); run<IDLE, CVT>(out, cyc, it); run<IDLE, CVTRTZ>(out, cyc, it); run<IDLE, MED3>(out, cyc, it); run<IDLE, FMIX>(out, cyc, it); run<IDLE, MOV>(out, cyc, it); run<IDLE, PERM>(out, cyc, it);
    run<FMA, FMA>(out, cyc, it); run<EXP, EXP>(out, cyc, it); run<CVT, CVT>(out, cyc, it);
    std::printf("-- the SIMD's other wavefront beside a matrix stream\n");
    run<M16, FMA>(out, cyc, it); run<M32, FMA>(out, cyc, it);
    run<M16, EXP>(out, cyc, it); run<M32, EXP>(out, cyc, it);
    run<M16, MIX>(out, cyc, it); run<M32, MIX>(out, cyc, it);
    run<M16, F64>(out, cyc, it); run<M32, F64>(out, cyc, it);
    run<M16, CVT>(out, cyc, it); run<M32, CVT>(out, cyc, it);
    run<M16, PKFMA>(out, cyc, it); run<M32, PKFMA>(out, cyc, it);
    run<M16, CVTRTZ>(out, cyc, it); run<M32, CVTRTZ>(out, cyc, it);
    run<M16, MED3>(out, cyc, it); run<M32, MED3>(out, cyc, it);
    run<M16, FMIX>(out, cyc, it); run<M32, FMIX>(out, cyc, it);
    run<M16, RCP>(out, cyc, it); run<M32, RCP>(out, cyc, it);
    run<M16, MOV>(out, cyc, it); run<M32, MOV>(out, cyc, it);
    run<M16, PERM>(out, cyc, it); run<M32, PERM>(out, cyc, it);
    run<M16, LDSR>(out, cyc, it); run<M32, LDSR>(out, cyc, it);
    run<M16, M16>(out, cyc, it); run<M32, M32>(out, cyc, it);
    std::printf("-- independent vector instructions of the SAME wavefront between its matrix instructions\n");
    run<M16_F1, IDLE>(out, cyc, it); run<M16_F2, IDLE>(out, cyc, it);
    run<M32_F1, IDLE>(out, cyc, it); run<M32_F2, IDLE>(out, cyc, it); run<M32_F3, IDLE>(out, cyc, it); run<M32_F4, IDLE>(out, cyc, it);
    run<M32_F5, IDLE>(out, cyc, it); run<M32_F6, IDLE>(out, cyc, it); run<M32_D2, IDLE>(out, cyc, it); run<M32_D4, IDLE>(out, cyc, it);
    std::printf("-- both: fillers in the matrix wavefront AND a vector stream in the other\n");
    run<M32_F2, FMA>(out, cyc, it); run<M32_F2, F64>(out, cyc, it); run<M32_F4, FMA>(out, cyc, it);
    return 0;
}
