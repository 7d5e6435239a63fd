// v_pk_mul_f32 D, A, B op_sel:[0,1] op_sel_hi:[1,0]   (D.lo = A.lo * B.HI, D.hi = A.hi * B.LO: the second source read with its halves
// swapped) -- the ONE instruction whose replacement by two v_mul_f32 at the ISA level cures the fused actor kernel's wrong v_par
// (tools/experiments/pk_isa_patch.py mul3_scalar): there its LOW result came out 0 in lanes 48..63, run to
// run, while the CU's other workgroup ran its policy phase (MFMAs, LDS reads, buffer loads) on the same SIMDs.  Does the bare
// instruction misbehave beside such a partner?  Wavefronts 4..7 of a 512-thread workgroup run it on changing inputs and check both
// halves against v_mul_f32; wavefronts 0..3 (their SIMD partners) idle, stream MFMAs, or stream MFMAs + LDS reads + global loads.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pk_mul_swap tools/ubench/pk_mul_src1_swap.hip && /tmp/pk_mul_swap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int PARTNER, int FORM>      // 0 idle, 1 MFMA stream, 2 MFMA + LDS reads + global loads (a policy phase's mix)
__global__ void __launch_bounds__(512) probe(float *out, const float4 *mem, unsigned long long *bad, int iters) {
    __shared__ float4 lds[1024];
    if (threadIdx.x < 256) {                                            // the SIMD partners
        f32x4 acc = {1, 2, 3, 4}; f16x8 x, y; float4 s = {0, 0, 0, 0};
        for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(0.5f + threadIdx.x); y[e] = (_Float16)0.25f; }
        lds[threadIdx.x] = lds[threadIdx.x + 256] = lds[threadIdx.x + 512] = lds[threadIdx.x + 768] = float4{1, 2, 3, 4};
        if (PARTNER) for (int it = 0; it < 4 * iters; ++it) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(y, x, acc, 0, 0, 0);
            if (PARTNER == 2) { const float4 a = lds[(threadIdx.x + 17 * it) & 1023], b = mem[(blockIdx.x * 256 + threadIdx.x + 64 * it) & 0xFFFFF]; s.x += a.x + b.y; x[it & 7] = (_Float16)s.x; }
        }
        out[blockIdx.x * 256 + threadIdx.x] = acc[0] + s.x;
        return;
    }
    unsigned s = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u, lo = 0, hi = 0, zero = 0, unswapped = 0, other = 0;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u; const float vx = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
        s = s * 1664525u + 1013904223u; const float vy = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
        s = s * 1664525u + 1013904223u; const float py = (float)(int)(s >> 8) * (1.0f / 16777216.0f) + 0.25f;
        f32x2 v = {vx, vy}, pp = {py, -py}, m;
        float e0, e1, u0;
        if (FORM == 0) {        // mul, second source fully swapped (the actor kernel's instruction)
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(m) : "v"(pp), "v"(v));     // (py vy, -py vx)
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(py), "v"(vy));
            asm volatile("v_mul_f32 %0, -%1, %2" : "=v"(e1) : "v"(py), "v"(vx));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u0) : "v"(py), "v"(vx));
        } else if (FORM == 1) { // mul, only the LOW result reads the high half (the high result reads it too)
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(m) : "v"(pp), "v"(v));     // (py vy, -py vy)
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(py), "v"(vy));
            asm volatile("v_mul_f32 %0, -%1, %2" : "=v"(e1) : "v"(py), "v"(vy));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u0) : "v"(py), "v"(vx));
        } else if (FORM == 2) { // mul, FIRST source swapped
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(m) : "v"(v), "v"(pp));     // (vy py, -vx py)
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(vy), "v"(py));
            asm volatile("v_mul_f32 %0, %1, -%2" : "=v"(e1) : "v"(vx), "v"(py));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u0) : "v"(vx), "v"(py));
        } else if (FORM == 3) { // add, second source swapped
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(m) : "v"(pp), "v"(v));     // (py + vy, -py + vx)
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(py), "v"(vy));
            asm volatile("v_sub_f32 %0, %2, %1" : "=v"(e1) : "v"(py), "v"(vx));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(u0) : "v"(py), "v"(vx));
        } else if (FORM == 5) { // v_pk_mov_b32, FIRST source's high half into the low result (the form the compiler emits all over the library)
            // (v_pk_mov_b32: D.lo = first source's half op_sel[0], D.hi = SECOND source's half op_sel[1])
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(m) : "v"(v), "v"(pp));                       // (vy, py)
            e0 = vy; e1 = py; u0 = vx;
        } else if (FORM == 6) { // v_pk_mov_b32, SECOND source's halves swapped
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(m) : "v"(pp), "v"(v));                       // (py, vy): the high result from the second source's high half
            e0 = py; e1 = vy; u0 = py;
            f32x2 m2;
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(m2) : "v"(v), "v"(pp));                      // (vy, -py)
            if (__float_as_uint(m2.x) != __float_as_uint(vy) || __float_as_uint(m2.y) != __float_as_uint(-py)) m.x = 12345.0f;
        } else {                // fma, THIRD source swapped (round 4's suspect)
            f32x2 one = {1.0f, 1.0f};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(m) : "v"(pp), "v"(one), "v"(v));   // (py + vy, -py + vx)
            asm volatile("v_fma_f32 %0, %1, 1.0, %2" : "=v"(e0) : "v"(py), "v"(vy));
            asm volatile("v_fma_f32 %0, -%1, 1.0, %2" : "=v"(e1) : "v"(py), "v"(vx));
            asm volatile("v_fma_f32 %0, %1, 1.0, %2" : "=v"(u0) : "v"(py), "v"(vx));
        }
        const bool bad_lo = __float_as_uint(m.x) != __float_as_uint(e0);
        lo += bad_lo;
        hi += __float_as_uint(m.y) != __float_as_uint(e1);
        if (bad_lo) { if (m.x == 0.0f) ++zero; else if (__float_as_uint(m.x) == __float_as_uint(u0)) ++unswapped; else ++other; }
    }
    if (lo) atomicAdd(bad + (threadIdx.x & 63) / 16, (unsigned long long)lo);        // low-half mismatches by quarter of the wavefront
    if (hi) atomicAdd(bad + 4, (unsigned long long)hi);
    if (zero) atomicAdd(bad + 5, (unsigned long long)zero);
    if (unswapped) atomicAdd(bad + 6, (unsigned long long)unswapped);
    if (other) atomicAdd(bad + 7, (unsigned long long)other);
}

template <int PARTNER, int FORM>
void run(float *out, const float4 *mem, unsigned long long *bad, int iters) {
    unsigned long long h[8];
    (void)hipMemset(bad, 0, 64);
    probe<PARTNER, FORM><<<512, 512>>>(out, mem, bad, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost);
    const char *names[] = {"idle", "mfma", "mfma+lds+global"};
    const char *forms[] = {"v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,1]", "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1]",
                           "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,1,0]", "v_pk_mov_b32 op_sel:[1,0]",
                           "v_pk_mov_b32 op_sel:[0,1] / [1,1]"};
    std::printf("%-46s partner %-16s low-half mismatches by lane quarter %llu %llu %llu %llu (result 0: %llu, result of the UNswapped halves: %llu, other: %llu)  "
                "high-half %llu  of %lld\n", forms[FORM], names[PARTNER], h[0], h[1], h[2], h[3], h[5], h[6], h[7], h[4], 512ll * 256 * iters);
}

template <int FORM>
void forms(float *out, const float4 *mem, unsigned long long *bad) { run<0, FORM>(out, mem, bad, 20000); run<1, FORM>(out, mem, bad, 20000); run<2, FORM>(out, mem, bad, 20000); }

int main() {
    float *out; float4 *mem; unsigned long long *bad;
    if (hipMalloc(&out, 512 * 256 * 4) != hipSuccess || hipMalloc(&mem, (1 << 20) * 16) != hipSuccess || hipMalloc(&bad, 64) != hipSuccess) return 2;
    (void)hipMemset(mem, 0, (1 << 20) * 16);
    for (int rep = 0; rep < 2; ++rep) { forms<0>(out, mem, bad); forms<1>(out, mem, bad); forms<2>(out, mem, bad); forms<3>(out, mem, bad); forms<4>(out, mem, bad); forms<5>(out, mem, bad); forms<6>(out, mem, bad); }
    return 0;
}
