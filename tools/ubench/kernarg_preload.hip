// Does this box's firmware honour kernel-argument PRELOAD (gfx940+: the first kernel arguments arrive in SGPRs with the wavefront instead of
// through an s_load from the kernarg segment)?  One wavefront: wall clock at entry, one global load through the pointer argument, wall clock when
// it has landed.  Without preload the load's address waits for the kernarg s_load (two dependent trips), with it one.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/kernarg_preload.hip -o /tmp/kp0
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=8 tools/ubench/kernarg_preload.hip -o /tmp/kp1
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
struct Pad { int a[64]; };
__global__ void probe(const int *src, unsigned long long *out, int slot, Pad pad) {
    const unsigned long long t0 = wall_clock64();
    const int v = src[threadIdx.x + 64 * slot];
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(v) : "memory");
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[slot] = (t1 - t0) + (unsigned long long)(v == 12345 ? pad.a[5] : 0);
}
int main() {
    const int n = 400;
    int *src; unsigned long long *out;
    hipMalloc(&src, 64 * n * sizeof(int)); hipMemset(src, 0, 64 * n * sizeof(int));
    hipMalloc(&out, n * sizeof(unsigned long long));
    Pad pad{};
    for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out, i, pad); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(n);
    hipMemcpy(h.data(), out, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin() + 50, h.end());
    printf("entry -> first global load landed, wall clock ticks of 10 ns: p10 %llu p50 %llu p90 %llu\n", h[50 + 35], h[50 + 175], h[50 + 315]);
    return 0;
}
