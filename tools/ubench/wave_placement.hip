// Where do the wavefronts of a workgroup land?  (development aid)  Every wavefront records its XCC / SE / CU / SIMD ids; the host
// prints how many distinct SIMDs the wavefronts of one workgroup use and how many workgroups share a CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
__global__ void k(unsigned long long *out, int spin) {
    extern __shared__ unsigned char smem[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = ((unsigned long long)(xcc & 15) << 32) | hw;
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }                       // stay resident so that the whole grid is placed at once
    if (smem[threadIdx.x] == 77) out[0] = 0;
}
int main() {
    for (int threads : {192, 256}) for (int lds : {1024, 84 * 1024}) for (int blocks : {192, 256, 342, 512}) {
        unsigned long long *d; const int waves = threads / 64;
        hipMalloc(&d, blocks * waves * 8);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        k<<<blocks, threads, lds>>>(d, 200000); hipDeviceSynchronize();
        unsigned long long *h = new unsigned long long[blocks * waves];
        hipMemcpy(h, d, blocks * waves * 8, hipMemcpyDeviceToHost);
        std::map<unsigned long long, int> per_cu; std::map<unsigned long long, int> per_simd; int blocks_all_distinct = 0;
        for (int b = 0; b < blocks; ++b) {
            std::set<unsigned> simds; unsigned long long cu = 0;
            for (int w = 0; w < waves; ++w) {
                const unsigned long long v = h[b * waves + w]; const unsigned hw = (unsigned)v;
                cu = ((v >> 32) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
                simds.insert((hw >> 4) & 3); per_simd[(cu << 4) | ((hw >> 4) & 3)]++;
            }
            per_cu[cu]++; blocks_all_distinct += (int)simds.size() == waves;
        }
        int cu_hist[8] = {0}, simd_hist[8] = {0};
        for (auto &p : per_cu) cu_hist[p.second < 7 ? p.second : 7]++;
        for (auto &p : per_simd) simd_hist[p.second < 7 ? p.second : 7]++;
        std::printf("threads %d lds %5d KiB blocks %3d: workgroups whose wavefronts sit on distinct SIMDs %3d; CUs used %3zu (with 1/2/3/4 workgroups: %d/%d/%d/%d); SIMDs with 1/2/3/4 wavefronts: %d/%d/%d/%d\n",
                    threads, lds / 1024, blocks, blocks_all_distinct, per_cu.size(), cu_hist[1], cu_hist[2], cu_hist[3], cu_hist[4], simd_hist[1], simd_hist[2], simd_hist[3], simd_hist[4]);
        hipFree(d); delete[] h;
    }
    return 0;
}
