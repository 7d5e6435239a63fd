// Standalone check (compiled and run by tests/test_gpu_sqrt.py): cavoid::sqrt_dist2 against the device library's correctly
// rounded sqrt(double), bit for bit, over random magnitudes, sums of squares of coordinate differences, exact squares, zero,
// the smallest argument it is specified for (2^-767) and signed zeros (+inf is outside its contract: a squared distance
// of finite positions is finite).  Prints "<inputs> <mismatches>".
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include "cavoid_kernels.hpp"

__global__ void compare(const double *in, unsigned long long *bad, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = sqrt(in[i]), b = cavoid::sqrt_dist2(in[i]);
    if (__double_as_longlong(a) != __double_as_longlong(b)) atomicAdd(bad, 1ull);
}

int main() {
    std::mt19937_64 rng(12345);
    std::vector<double> v;
    std::uniform_real_distribution<double> u(-50.0, 50.0), e(-700.0, 300.0), m(1.0, 2.0);
    for (int i = 0; i < 4000000; ++i) { const double dx = u(rng) - u(rng), dy = u(rng) - u(rng); v.push_back(dx * dx + dy * dy); }   // the pair pass's arguments
    for (int i = 0; i < 2000000; ++i) v.push_back(std::ldexp(m(rng), (int)e(rng)));                                              // any magnitude
    for (int i = 0; i < 1000000; ++i) { const double r = std::floor(std::fabs(u(rng)) * 1e6); v.push_back(r * r); }                 // exact squares
    for (int i = 0; i < 1000000; ++i) { const double r = 0.4 + 1e-9 * u(rng); v.push_back(r * r); }                                 // around a collision distance
    v.push_back(0.0); v.push_back(-0.0); v.push_back(std::ldexp(1.0, -767)); v.push_back(std::ldexp(1.5, -767)); v.push_back(std::ldexp(1.0, -1000)); v.push_back(std::ldexp(1.0, 1000));
    v.push_back(1.0); v.push_back(4.0); v.push_back(2.0); v.push_back(std::nextafter(1.0, 2.0)); v.push_back(std::nextafter(1.0, 0.0));
    const int n = (int)v.size();
    double *d_in; unsigned long long *d_bad, bad = 0;
    if (hipMalloc(&d_in, n * sizeof(double)) != hipSuccess || hipMalloc(&d_bad, sizeof(bad)) != hipSuccess) return 2;
    hipMemcpy(d_in, v.data(), n * sizeof(double), hipMemcpyHostToDevice);
    hipMemset(d_bad, 0, sizeof(bad));
    compare<<<(n + 255) / 256, 256>>>(d_in, d_bad, n);
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
    std::printf("%d %llu\n", n, bad);
    return bad == 0 ? 0 : 1;
}
