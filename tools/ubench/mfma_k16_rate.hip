// How many cycles does the K = 16 float16 matrix instruction take on gfx950 next to the K = 32 one?  (development aid: the policy kernel's
// LSTM-input and host chunks carry 8 and 4 real k of the 32 a v_mfma_f32_16x16x32_f16 multiplies -- DESIGN.md section 8 item 2 (c).)
// One wavefront per SIMD (256 workgroups x 256 threads), 16 independent accumulators, s_memtime around the stream.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void __launch_bounds__(256) stream(float *out, long long *clk, int iters) {
    f32x4 acc[16];
    for (int k = 0; k < 16; ++k) acc[k] = f32x4{1.f + k, 2.f * k, 3.f - k, (float)(threadIdx.x + k)};   // (distinct: identical chains would be merged)
    f16x8 a8, b8; f16x4 a4, b4;
    for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(0.5f + e); b8[e] = (_Float16)(0.25f * e); }
    for (int e = 0; e < 4; ++e) { a4[e] = a8[e]; b4[e] = b8[e]; }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            // (inline asm: left to the compiler the accumulators wander between register files inside the loop)
            if (KIND == 32) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a8), "v"(b8));
            else asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a4), "v"(b4));
        }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int k = 0; k < 16; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
    float *out; long long *clk, h[256];
    if (hipMalloc(&out, 256 * 256 * 4) != hipSuccess || hipMalloc(&clk, 256 * 8) != hipSuccess) return 2;
    const int iters = 4000;
    for (int kind : {32, 16, 32, 16}) {
        if (kind == 32) stream<32><<<256, 256>>>(out, clk, iters); else stream<16><<<256, 256>>>(out, clk, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
        std::printf("v_mfma_f32_16x16x%d_f16: %.2f s_memtime ticks per instruction and wavefront (one wavefront per SIMD, 16 accumulators)\n", kind, s / 256 / iters / 16);
    }
    return 0;
}
