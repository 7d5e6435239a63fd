// Per-CU scaling of vector throughput (development aid): W wavefronts per CU (one per SIMD up to 4, then two), each running the
// same dependent-chain-free stream of one instruction kind; wall-clock time per instruction and the shader clock it ran at.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void k(double *out, long long *cyc, int iters, double a, double b) {
    double r[8]; float f[8];
    for (int i = 0; i < 8; ++i) { r[i] = a + i; f[i] = (float)a + i; }
    const float fa = (float)a, fb = (float)b;
    const long long w0 = wall_clock64(), c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) r[i] = __builtin_fma(r[i], b, a);
                else if (KIND == 1) f[i] = __builtin_fmaf(f[i], fb, fa);
                else if (KIND == 2) { r[i] = __builtin_fma(r[i], b, a); f[i] = __builtin_fmaf(f[i], fb, fa); }
                else if (KIND == 3) r[i] = __builtin_amdgcn_rsq(r[i]) + b;
                else if (KIND == 4) r[i] = __builtin_amdgcn_rcp(r[i]) + b;
                else if (KIND == 5) r[i] = __builtin_rint(r[i]) + b;
                else if (KIND == 6) { unsigned u2 = (unsigned)f[i]; u2 = u2 * 2654435761u + (unsigned)it; f[i] = (float)__umulhi(u2, 40503u); }
                else if (KIND == 7) r[i] = (double)(float)r[i] + b;
                else if (KIND == 8) r[i] = r[i] < b ? r[i] + a : r[i] - a;
            }
    }
    const long long w1 = wall_clock64(), c1 = clock64();
    double acc = 0; for (int i = 0; i < 8; ++i) acc += r[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) { const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); cyc[2 * w] = w1 - w0; cyc[2 * w + 1] = c1 - c0; }
}
template <int KIND>
void run(const char *name, int blocks, int threads, double *out, long long *cyc) {
    static long long h[2 * 4096];
    const int iters = 4000, waves = blocks * threads / 64;
    for (int rep = 0; rep < 2; ++rep) { k<KIND><<<blocks, threads>>>(out, cyc, iters, 1.25, 1.0000001); hipDeviceSynchronize(); }
    hipMemcpy(h, cyc, sizeof(long long) * 2 * waves, hipMemcpyDeviceToHost);
    double wall = 0, shader = 0;
    for (int w = 0; w < waves; ++w) { wall += h[2 * w]; shader += h[2 * w + 1]; }
    const double n = (double)iters * 64;
    std::printf("%-12s %3d workgroups x %d wavefronts: %6.2f ns per instruction and wavefront (wall), %5.2f shader clocks, shader clock %4.0f MHz\n",
                name, blocks, threads / 64, wall / waves * 10.0 / n, shader / waves / n, shader / wall * 100.0);
}
int main() {
    double *out; long long *cyc; hipMalloc(&out, 512 * 512 * 8); hipMalloc(&cyc, 2 * 4096 * 8);
    for (int t : {64, 128, 192, 256, 512}) { run<0>("fma f64", 256, t, out, cyc); }
    for (int t : {64, 192, 256, 512}) { run<1>("fma f32", 256, t, out, cyc); }
    for (int t : {64, 192, 256, 512}) { run<2>("f64 + f32", 256, t, out, cyc); }
    for (int t : {64, 192, 256, 512}) { run<3>("rsq f64 + add", 256, t, out, cyc); }
    for (int t : {64, 192, 256, 512}) { run<4>("rcp f64 + add", 256, t, out, cyc); }
    for (int t : {64, 192, 256}) { run<5>("rndne f64+add", 256, t, out, cyc); }
    for (int t : {64, 192, 256}) { run<6>("u32 mul/mulhi", 256, t, out, cyc); }
    for (int t : {64, 192, 256}) { run<7>("cvt f64-f32-f64", 256, t, out, cyc); }
    for (int t : {64, 192, 256}) { run<8>("cmp+cndmask f64", 256, t, out, cyc); }
    return 0;
}
