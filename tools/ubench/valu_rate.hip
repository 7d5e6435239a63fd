// Issue cost of the vector instructions the env kernels are made of, on this GPU (development aid): one wavefront per
// workgroup, one workgroup per CU -- each instruction in 8 independent chains, 64 per loop body, shader clocks per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAINS 8
#define BODY(T, INIT, OP)                                                                        \
    T r[CHAINS];                                                                                 \
    for (int k = 0; k < CHAINS; ++k) r[k] = INIT;                                                \
    const long long t0 = clock64();                                                              \
    for (int it = 0; it < iters; ++it) {                                                         \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) { _Pragma("unroll") for (int k = 0; k < CHAINS; ++k) { OP; } } \
    }                                                                                            \
    const long long t1 = clock64();                                                              \
    T acc = r[0]; for (int k = 1; k < CHAINS; ++k) acc = acc + r[k];                             \
    out[blockIdx.x * 64 + threadIdx.x] = (double)acc; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;

__global__ void k_fma64(double *out, long long *cyc, int iters, double a, double b) { BODY(double, a + k, r[k] = __builtin_fma(r[k], b, a)) }
__global__ void k_mul64(double *out, long long *cyc, int iters, double a, double b) { BODY(double, a + k, r[k] = r[k] * b) }
__global__ void k_add64(double *out, long long *cyc, int iters, double a, double b) { BODY(double, a + k, r[k] = r[k] + b) }
__global__ void k_fma32(double *out, long long *cyc, int iters, double a, double b) { const float fa = (float)a, fb = (float)b; BODY(float, fa + k, r[k] = __builtin_fmaf(r[k], fb, fa)) }
__global__ void k_cvt(double *out, long long *cyc, int iters, double a, double b) { BODY(double, a + k, r[k] = (double)((float)r[k]) + b) }
__global__ void k_rsq64(double *out, long long *cyc, int iters, double a, double b) { BODY(double, a + k, r[k] = __builtin_amdgcn_rsq(r[k]) + b) }
__global__ void k_ldexp(double *out, long long *cyc, int iters, double a, double b) { const int e = (int)b; BODY(double, a + k, r[k] = __builtin_ldexp(r[k], e) + b) }
__global__ void k_rndne(double *out, long long *cyc, int iters, double a, double b) { BODY(double, a + k, r[k] = __builtin_rint(r[k]) + b) }
__global__ void k_min64(double *out, long long *cyc, int iters, double a, double b) { BODY(double, a + k, r[k] = __builtin_fmin(r[k], b) + b) }
__global__ void k_cmp64(double *out, long long *cyc, int iters, double a, double b) { BODY(double, a + k, r[k] = (r[k] < b ? a : b) + r[k]) }
__global__ void k_cmpu64(double *out, long long *cyc, int iters, double a, double b) { const unsigned long long ub = (unsigned long long)b; BODY(unsigned long long, (unsigned long long)(a + k), r[k] = r[k] + (r[k] < ub ? 3ull : 5ull)) }
__global__ void k_add32(double *out, long long *cyc, int iters, double a, double b) { const unsigned ub = (unsigned)b; BODY(unsigned, (unsigned)(a + k), r[k] = r[k] * 3u + ub) }
__global__ void k_exp32(double *out, long long *cyc, int iters, double a, double b) { const float fb = (float)b; BODY(float, (float)(a + k), r[k] = __builtin_amdgcn_exp2f(r[k]) * fb) }

int main() {
    double *out; long long *cyc; hipMalloc(&out, 256 * 64 * 8); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000; long long h[256];
#define RUN(name, per) do { name<<<256, 64>>>(out, cyc, iters, 1.25, 1.0000001); hipDeviceSynchronize(); name<<<256, 64>>>(out, cyc, iters, 1.25, 1.0000001); hipDeviceSynchronize(); \
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost); double s = 0; for (int i = 0; i < 256; ++i) s += h[i]; \
    std::printf("%-10s %6.2f clocks per loop-body operation (%d vector instructions each)\n", #name, s / 256 / iters / 64, per); } while (0)
    RUN(k_fma64, 1); RUN(k_mul64, 1); RUN(k_add64, 1); RUN(k_fma32, 1); RUN(k_cvt, 3); RUN(k_rsq64, 2); RUN(k_ldexp, 2); RUN(k_rndne, 2);
    RUN(k_min64, 2); RUN(k_cmp64, 4); RUN(k_cmpu64, 4); RUN(k_add32, 1); RUN(k_exp32, 2);
    return 0;
}
