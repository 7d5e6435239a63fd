// Do a wavefront's MFMAs and its SIMD partner's vector instructions overlap?  (development aid)  One workgroup of 512 threads
// per CU: wavefronts w and w+4 share a SIMD.  Role A (wavefronts 0..3) and role B (4..7) each run a stream of one kind; the
// shader clocks of each role alone and of both together say whether the pair costs max(A, B) or A + B.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
enum { IDLE = 0, MFMA = 1, FMA = 2, EXP = 3, MIX = 4, LDSR = 5, MFMA_N3 = 6, MFMA_N7 = 7, MFMA_N9 = 8, MFMA_N11 = 9, MFMA_N13 = 10 };

template <int KIND>
__device__ __forceinline__ float stream(int iters, float a, float b, const unsigned char *lds) {
    if (KIND == MFMA || KIND >= MFMA_N3) {
        f32x4 acc[16];
        for (int k = 0; k < 16; ++k) acc[k] = f32x4{a, a, a, a};
        bf16x8 x, y;
        for (int e = 0; e < 8; ++e) { x[e] = (__bf16)a; y[e] = (__bf16)b; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[k], 0, 0, 0);
                    if (KIND >= MFMA_N3) {                                   // keep the next MFMA out of the issue stage while the pipe is busy
                        __builtin_amdgcn_sched_barrier(0);
                        if (KIND == MFMA_N3) asm volatile("s_nop 3"); else if (KIND == MFMA_N7) asm volatile("s_nop 7");
                        else if (KIND == MFMA_N9) asm volatile("s_nop 9"); else if (KIND == MFMA_N11) asm volatile("s_nop 11"); else asm volatile("s_nop 13");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        }
        float s = 0; for (int k = 0; k < 16; ++k) s += acc[k][0];
        return s;                                                            // 64 MFMAs per iteration
    } else if (KIND == FMA || KIND == EXP || KIND == MIX) {
        float r[8];
        for (int k = 0; k < 8; ++k) r[k] = a + k;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (KIND == FMA) r[k] = __builtin_fmaf(r[k], b, a);
                    else if (KIND == EXP) r[k] = __builtin_amdgcn_exp2f(r[k]);
                    else r[k] = (u & 3) == 0 ? __builtin_amdgcn_exp2f(r[k]) : __builtin_fmaf(r[k], b, a);
                }
        }
        float s = 0; for (int k = 0; k < 8; ++k) s += r[k];
        return s;                                                            // 64 vector instructions per iteration
    } else if (KIND == LDSR) {
        u32x4 s4 = u32x4{0, 0, 0, 0};
        const unsigned char *p = lds + (threadIdx.x & 63) * 528;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) s4 ^= *reinterpret_cast<const u32x4 *>(p + u * 16);
            asm volatile("" ::: "memory");
        }
        return (float)(s4.x ^ s4.y ^ s4.z ^ s4.w);
    }
    return 0.0f;
}

template <int KA, int KB, int PRIO_B>
__global__ void __launch_bounds__(512) pair(float *out, long long *cyc, int iters, float a, float b) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 528];
    for (int i = threadIdx.x; i < 64 * 528 / 4; i += 512) reinterpret_cast<unsigned *>(lds)[i] = i;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    float r;
    const long long t0 = clock64();
    if (wave < 4) r = stream<KA>(iters, a, b, lds);
    else { if (PRIO_B) __builtin_amdgcn_s_setprio(PRIO_B); r = stream<KB>(iters, a, b, lds); }
    const long long t1 = clock64();
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

static const char *name(int k) { static const char *n[] = {"idle", "mfma", "fma32", "exp32", "mix(1 exp : 3 fma)", "lds b128", "mfma + s_nop 3", "mfma + s_nop 7", "mfma + s_nop 9", "mfma + s_nop 11", "mfma + s_nop 13"}; return n[k]; }

template <int KA, int KB, int PRIO_B = 0>
void run(float *out, long long *cyc, int iters) {
    static long long h[256 * 8];
    for (int rep = 0; rep < 2; ++rep) { pair<KA, KB, PRIO_B><<<256, 512>>>(out, cyc, iters, 1.25f, 1.0000001f); hipDeviceSynchronize(); }
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double sa = 0, sb = 0;
    for (int i = 0; i < 256; ++i) for (int w = 0; w < 4; ++w) { sa += h[i * 8 + w]; sb += h[i * 8 + 4 + w]; }
    const double per_a = KA == LDSR ? 16 : 64, per_b = KB == LDSR ? 16 : 64;
    std::printf("A = %-20s B = %-20s prio(B) %d:  A %7.2f clocks per instruction, B %7.2f\n", name(KA), name(KB), PRIO_B,
                sa / 1024 / iters / per_a, sb / 1024 / iters / per_b);
}

int main() {
    float *out; long long *cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int it = 2000;
    run<MFMA, IDLE>(out, cyc, it); run<IDLE, FMA>(out, cyc, it); run<IDLE, EXP>(out, cyc, it); run<IDLE, MIX>(out, cyc, it); run<IDLE, LDSR>(out, cyc, it);
    run<MFMA, MFMA>(out, cyc, it); run<FMA, FMA>(out, cyc, it);
    run<MFMA, FMA>(out, cyc, it); run<MFMA, FMA, 1>(out, cyc, it); run<MFMA, EXP>(out, cyc, it); run<MFMA, MIX>(out, cyc, it); run<MFMA, MIX, 1>(out, cyc, it);
    run<MFMA, LDSR>(out, cyc, it); run<FMA, LDSR>(out, cyc, it);
    run<MFMA_N3, IDLE>(out, cyc, it); run<MFMA_N7, IDLE>(out, cyc, it); run<MFMA_N9, IDLE>(out, cyc, it); run<MFMA_N11, IDLE>(out, cyc, it); run<MFMA_N13, IDLE>(out, cyc, it);
    run<MFMA_N3, FMA>(out, cyc, it); run<MFMA_N7, FMA>(out, cyc, it); run<MFMA_N9, FMA>(out, cyc, it); run<MFMA_N11, FMA>(out, cyc, it); run<MFMA_N13, FMA>(out, cyc, it);
    run<MFMA_N9, MIX>(out, cyc, it); run<MFMA_N11, MIX>(out, cyc, it); run<MFMA_N11, MIX, 1>(out, cyc, it); run<MFMA_N11, MFMA_N11>(out, cyc, it);
    return 0;
}
