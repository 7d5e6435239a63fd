// cavoid_policy_split.hpp -- the actors' NetworkVP_rnn inference (predict_p_and_v + select_action, see cavoid_policy.hpp for
// the graph and its citations) on the bf16 matrix pipe with float32 inputs, outputs and accumulation: operand splitting.
//
//   Every float32 operand is written as a sum of bf16 pieces (8 significant bits each):
//       weight  w = w1 + w2 + w3   (24 bits: exact)            -- split once, at cavoid_policy_load time
//       activation a = a1 + a2     (16 bits + rounding: |a - a1 - a2| <= 2^-17 |a|)  -- split in each layer's epilogue
//   and a product is the sum of its P largest partial products, accumulated in float32 by the MFMA:
//       w*a ~= w1*a1 + w1*a2 + w2*a1 [+ w3*a1 [+ w2*a2]]         P = 3 (default) [4 [5]]; w3*a2 ~ 2^-26 is never formed
//   v_mfma_f32_16x16x32_bf16 runs at 16x the rate of v_mfma_f32_16x16x4_f32 per unit of K, so P of them cost P/16 of the
//   float32 instruction's time.  The result differs from a float32 GEMM by the activation rounding (relative 2^-17 per
//   product, every P) plus the dropped products; the table below (kSpDefaultProducts) has the measured errors against the network
//   in float64 -- 3.4e-6 on p / 2.5e-5 on v at worst with P = 3, inside the 2e-5 / 2e-4 bar the float32 kernel is held to by
//   a factor 6 / 8.  CAVOID_POLICY_F32=1 keeps inference on the float32-MFMA kernel.
//
// Layout.  One workgroup = 64 rows x 4 wavefronts; the GEMMs are computed TRANSPOSED, D[m][n] = sum_k W[k][m] * act[n][k]:
//   * MFMA operand A = weights (lane: output column m = l%16 of a 16-column tile, k = 8*(l/16) .. +7), operand B =
//     activations (lane: batch row n = l%16 of a 16-row tile, same k), result lane: row n = l%16, columns m = 4*(l/16)+r:
//     a lane ends up with FOUR CONSECUTIVE output columns of one row = four consecutive k of the next layer, so the
//     epilogue packs them and writes 8 bytes per plane (ds_write_b64) -- no 2-byte scatter;
//   * activations live in LDS as two 16-bit planes [64][264] (row stride 528 B; inside a row the k-groups are placed by sp_phys()
//     so that the 16-byte fragment reads of a hardware lane group hit 16 different bank groups -- see there); during the LSTM
//     columns 0..63 hold h and columns 64+8s.. the input slots (s = 0: the 4
//     host values, s = 1+t: the 7 values of the t-th observed agent), so the "input chunk" of the two 71/68-wide layers is
//     one predicated 16-byte read by the lanes of k-group 0;
//   * wavefront w owns output columns 64w..64w+63 of every layer (4 column tiles x 4 row tiles, 64 accumulator
//     registers); for the LSTM its four column tiles are the i, j, f, o gates of hidden units 16w..16w+15, so the cell
//     update is per lane and the cell state never leaves registers;
//   * weights: fragment-ordered copy (policy_pack_split_kernel), [layer][chunk][plane][column tile][lane] x 16 bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "cavoid_policy.hpp"

namespace cavoid {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// How many of the partial products w_i * a_j are formed (largest first) is a template parameter P of the kernels.  With
// |w2| <= 2^-9 |w|, |w3| <= 2^-17 |w|, |a2| <= 2^-9 |a| and the activation split's own rounding |a - a1 - a2| <= 2^-17 |a| (which
// every variant carries), per product:
//   P = 5: w1a1 + w1a2 + w2a1 + w2a2 + w3a1   dropped: w3a2 (2^-26)                                      error ~2^-17
//   P = 4: w1a1 + w1a2 + w2a1 + w3a1          dropped: + w2a2 (2^-18)                                    ~1.5 x 2^-17
//   P = 3: w1a1 + w1a2 + w2a1                 dropped: + w3a1 (2^-17); the third weight plane is never read   ~2.5 x 2^-17
// Measured against the network in float64 (tools/split_products_ab.py; 3 seeds x 32 768 rows; M = 3 / M = 9 / inputs x 4):
//   P = 5: |dp| 3.6e-7 / 3.6e-7 / 2.1e-6, |dv| 3.3e-6 / 2.6e-6 / 1.3e-5;  76.7 us per 32 768 rows
//   P = 4: |dp| 4.6e-7 / 5.1e-7 / 2.8e-6, |dv| 4.1e-6 / 3.8e-6 / 1.7e-5;  70.2 us
//   P = 3: |dp| 6.2e-7 / 6.8e-7 / 3.4e-6, |dv| 4.9e-6 / 4.3e-6 / 2.5e-5;  58.8 us      (float32 PyTorch graph: ~1e-7 / ~1e-6)
// i.e. the two smallest products buy less than a factor 2 of an error that the 16-bit activation pieces set anyway, for a third
// more matrix time: P = 3 is the default (still 6x inside the 2e-5 bar on p, 8x inside 2e-4 on v, with saturating inputs);
// CAVOID_POLICY_PRODUCTS=5 (or 4) selects the others at cavoid_policy_create time.
//
// Round 4 -- the float32-GRADE form, P = kSpF16 (the default): the SAME three products on the SAME matrix instruction rate, with
// float16 pieces (11 significant bits each) instead of bf16 ones (8):
//       w = w1 + w2,  a = a1 + a2   (22 bits each: |x - x1 - x2| <= 2^-23 |x|, or 2^-25 absolute once x2 is a float16 subnormal)
//       w*a ~= w1*a1 + w1*a2 + w2*a1          dropped: w2*a2 (2^-22)  -- v_mfma_f32_16x16x32_f16 runs at the bf16 instruction's rate
// i.e. ~2^-21 per product where the bf16 form has ~2.5 x 2^-17: the reference's predictor is TensorFlow float32
// (ThreadPredictor.py:46,67; NetworkVPCore.py:160-176), and this is the form whose error sits at float32's own (measured against
// the network in float64, tools/split_products_ab.py, same cases as above: see profiles/r04_split_f16_vs_bf16.txt).  float16's
// range is the price: a piece saturates at +-65504 (inputs and relu outputs are clamped there -- one v_med3 where the bf16 form has a
// v_max -- so that an absurd activation degrades instead of turning into NaN; LSTM states are bounded by 1), weights beyond it do
// not occur.  No scaling is needed at the small end: a second piece that falls into float16's subnormals keeps an ABSOLUTE
// precision of 2^-25, which is below the float32 accumulator's own rounding of a sum of O(1) terms.
constexpr int kSpF16 = 16;                      // P: two float16 pieces per operand, three partial products
constexpr int kSpDefaultProducts = kSpF16;
template <int P> struct SplitFmt { static constexpr bool f16 = (P == kSpF16); static constexpr int planes = f16 ? 2 : (P >= 4 ? 3 : 2); };
constexpr float kSpF16Max = 65504.0f;
constexpr int kSpStrideB = 528;                 // bytes per LDS row of one plane (264 bf16)
constexpr int kSpPlaneB = 64 * kSpStrideB;      // 33 792 B
constexpr int kSpSlotCol = 64;                  // first input-slot column
constexpr int kSpZeroCol = 256;                 // columns 256..263 of every row (both planes) hold zeros for the whole pass: what the k-groups
                                                // 1..3 of an input-slot chunk read (a plain address select instead of 32 predicated moves)
constexpr int kSpMaxOthers = 23;                // slots 0..M must fit columns 64..255
// Where logical column `col` of a row lives inside the row's 528 bytes (one plane).  A K = 32 chunk c is four 16-byte k-groups g (8 columns
// each); gfx950 services a ds_read_b128 in four NON-contiguous 16-lane groups -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32
// (MI355X_MICROARCH.md, LDS) -- i.e. in fragment terms rows {0-3, 12-15} of k-group 0 TOGETHER WITH rows 4-11 of k-group 1: with the four
// k-groups of a chunk side by side (byte 64 c + 16 g, rounds 2-4) every fragment read had a two-way bank conflict in each of its four groups
// (8 LDS cycles instead of 4; PMC: 4.1 conflict cycles per LDS instruction).  Round 5: k-groups g and g ^ 1 sit 256 bytes (one bank row)
// apart, so that a lane group's sixteen rows land on sixteen different 16-byte slots:  byte = 32 c + 16 (g >> 1) + 256 (g & 1) + 2 e.
// Columns 256.. (the zero column) keep byte 2 col.
__host__ __device__ constexpr int sp_phys(int col) {
    return col >= 256 ? 2 * col : 32 * (col >> 5) + 16 * ((col >> 4) & 1) + 256 * ((col >> 3) & 1) + 2 * (col & 7);
}
static_assert(sp_phys(0) == 0 && sp_phys(8) == 256 && sp_phys(16) == 16 && sp_phys(24) == 272 && sp_phys(32) == 32 && sp_phys(255) == 510, "sp_phys");
// chunk counts (K = 32 each)
constexpr int kSpChLstm = 3, kSpChL1 = 3, kSpChWide = 8;
constexpr int kSpSlotChunk = 2;                 // the LSTM's / layer1's last chunk: the 8-wide input slot (chunks 0, 1: the hidden state)
constexpr int64_t kSpFragPerChunk = 3 * 16 * 64;          // planes x column tiles x lanes
constexpr int64_t kSpOffLstm = 0;
constexpr int64_t kSpOffL1 = kSpOffLstm + kSpChLstm * kSpFragPerChunk;
constexpr int64_t kSpOffL2 = kSpOffL1 + kSpChL1 * kSpFragPerChunk;
constexpr int64_t kSpOffFc1 = kSpOffL2 + kSpChWide * kSpFragPerChunk;
constexpr int64_t kSpOffHead = kSpOffFc1 + kSpChWide * kSpFragPerChunk;   // one column tile: 3 x 64 frags per chunk
constexpr int64_t kSpPackFrags = kSpOffHead + kSpChWide * 3 * 64;
// behind them: the LSTM once more in the EIGHT-wavefront form's column order (cavoid_policy_split8.hpp), and its 256 biases behind the packed biases
constexpr int64_t kSpOffLstm8 = kSpPackFrags;
constexpr int64_t kSpPackFrags8 = kSpOffLstm8 + kSpChLstm * kSpFragPerChunk;
constexpr int kBiasLstm8 = kBiasFloats, kBiasFloats8 = kBiasFloats + 256;
constexpr size_t policy_split_lds_bytes() { return (size_t)2 * kSpPlaneB + 64 * sizeof(float) + 64 * sizeof(int) + 64; }

// float -> (hi, lo) 16-bit pieces with hi = rn(x), lo = rn(x - hi); two values at a time (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32)
template <bool F16 = false>
__device__ __forceinline__ void split2(float x0, float x1, uint32_t &hi, uint32_t &lo) {
    if constexpr (F16) {
        const f16x2 h = __builtin_convertvector(f32x2{x0, x1}, f16x2);
        hi = __builtin_bit_cast(uint32_t, h);
        // x - rn16(x), exact in float32, as an fma on the float16 piece itself: v_fma_mix_f32 converts in the operand read (one
        // instruction per value instead of v_cvt_f32_f16 + half a v_pk_add_f32; the compiler does not form it from a subtraction)
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(x1));
        lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{r0, r1}, f16x2));
        // (v_fma_mixlo_f16 / v_fma_mixhi_f16 would round the residual inside the fma, 2.5 instead of 3 instructions per value: measured
        //  SLOWER -- +0.3 us per env step in the actor loop, +2.5 us stand-alone; the pair is a dependent chain on one register)
    } else {
        const bf16x2 h = __builtin_convertvector(f32x2{x0, x1}, bf16x2);
        hi = __builtin_bit_cast(uint32_t, h);
        const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xFFFF0000u);
        lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
    }
}
// float -> three bf16 pieces (exact: 3 x 8 bits cover the 24-bit significand)
__device__ __forceinline__ void split3(float x, uint32_t &p1, uint32_t &p2, uint32_t &p3) {
    uint32_t a, b, c;
    split2<false>(x, 0.0f, a, b);
    p1 = a & 0xFFFFu; p2 = b & 0xFFFFu;
    const float r = (x - __uint_as_float(p1 << 16)) - __uint_as_float(p2 << 16);
    const bf16x2 h = __builtin_convertvector(f32x2{r, 0.0f}, bf16x2);
    c = __builtin_bit_cast(uint32_t, h);
    p3 = c & 0xFFFFu;
}
// float -> two float16 pieces (22 bits; the third plane stays zero and is never read)
__device__ __forceinline__ void split3_f16(float x, uint32_t &p1, uint32_t &p2, uint32_t &p3) {
    uint32_t a, b;
    x = __builtin_amdgcn_fmed3f(x, -kSpF16Max, kSpF16Max);
    split2<true>(x, 0.0f, a, b);
    p1 = a & 0xFFFFu; p2 = b & 0xFFFFu; p3 = 0u;
}

// element (k-chunk c, k-group g, element e, packed column col) of a layer, in cavoid_policy.hpp's policy_weight() terms
// The LSTM gate pre-activations leave the GEMM already multiplied by what the cell update's 2^x needs: the i, f, o columns (and
// biases) carry log2 e, the j column 2 log2 e -- sigmoid(x) = 1 / (1 + 2^-(x log2 e)), tanh(x) = 1 - 2 / (1 + 2^(2 x log2 e)) -- folded
// into the packed weights at load time (one float32 rounding of each weight, 2^-24: below the split's own 2^-22).
constexpr float kSpLog2e = 1.4426950408889634f;
__device__ __forceinline__ float split_gate_scale(int packed_col) { return ((packed_col >> 4) & 3) == 1 ? 2.0f * kSpLog2e : kSpLog2e; }
__device__ __forceinline__ float split_weight(const PolicyWeights &w, int layer, int c, int g, int e, int col) {
    if (layer <= 1) {                                       // LSTM / layer1: two chunks of hidden state, then the input slot
        const float scale = layer == 0 ? split_gate_scale(col) : 1.0f;
        if (c < 2) return scale * policy_weight(w, layer, 32 * c + 8 * g + e, col);
        const int n_in = layer == 0 ? kPolOther : kPolHost;
        return (g == 0 && e < n_in) ? scale * policy_weight(w, layer, kPolHidden + e, col) : 0.0f;
    }
    return policy_weight(w, layer, 32 * c + 8 * g + e, col);
}

#ifdef CAVOID_POLICY_KERNELS     /* the non-template kernels are compiled by cavoid_policy_capi.hip only */
// `clamped` (may be null): incremented for every weight the float16 form saturates at +-65504 (after the LSTM gates' log2 e scale)
__global__ void __launch_bounds__(256) policy_pack_split_kernel(const PolicyWeights w, uint4 *frags, float *sbias, const int f16,
                                                                uint32_t *clamped) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one (layer, chunk, column tile, lane): all 3 planes
    constexpr int64_t kWide = kSpOffHead / 3, kAll = kWide + kSpChWide * 64;
    if (f < kBiasFloats) {                                                 // the biases in packed column order (as policy_pack_kernel), LSTM gates scaled
        const int i = (int)f;
        float b;
        if (i < kBiasL1) {
            const int wave = i >> 6, gate = (i >> 4) & 3, u = i & 15;
            b = split_gate_scale(i) * (w.lstm_bias[gate * kPolHidden + 16 * wave + u] + (gate == 2 ? w.forget_bias : 0.0f));
        } else if (i < kBiasL2) b = w.layer1_bias[i - kBiasL1];
        else if (i < kBiasFc1) b = w.layer2_bias[i - kBiasL2];
        else if (i < kBiasHead) b = w.fc1_bias[i - kBiasFc1];
        else {
            const int c = i - kBiasHead;
            b = c < w.num_actions ? w.p_bias[c] : (c == w.num_actions ? w.v_bias[0] : 0.0f);
        }
        sbias[i] = b;
    }
    if (f >= kAll) return;
    int layer, c, mt, lane;
    int64_t base;                                                          // frag index of plane 0
    if (f >= kWide) {
        const int64_t r = f - kWide;
        layer = 4; c = (int)(r >> 6); mt = 0; lane = (int)(r & 63);
        base = kSpOffHead + (int64_t)c * 3 * 64 + lane;
    } else {
        const int64_t per = 16 * 64;                                       // per chunk, per plane
        const int64_t ch = f / per, r = f - ch * per;                      // global chunk index over the four wide layers
        mt = (int)(r >> 6); lane = (int)(r & 63);
        if (ch < kSpChLstm) { layer = 0; c = (int)ch; base = kSpOffLstm; }
        else if (ch < kSpChLstm + kSpChL1) { layer = 1; c = (int)ch - kSpChLstm; base = kSpOffL1; }
        else if (ch < kSpChLstm + kSpChL1 + kSpChWide) { layer = 2; c = (int)ch - kSpChLstm - kSpChL1; base = kSpOffL2; }
        else { layer = 3; c = (int)ch - kSpChLstm - kSpChL1 - kSpChWide; base = kSpOffFc1; }
        base += (int64_t)c * kSpFragPerChunk + (int64_t)mt * 64 + lane;
    }
    const int g = lane >> 4, col = 16 * mt + (lane & 15);
    uint32_t pl[3][4];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        uint32_t a1, a2, a3, b1, b2, b3;
        if (f16) {
            const float w0 = split_weight(w, layer, c, g, e, col), w1 = split_weight(w, layer, c, g, e + 1, col);
            const int beyond = (int)!(__builtin_fabsf(w0) <= kSpF16Max) + (int)!(__builtin_fabsf(w1) <= kSpF16Max);   // (NaN counts)
            if (beyond && clamped) atomicAdd(clamped, (uint32_t)beyond);
            split3_f16(w0, a1, a2, a3);
            split3_f16(w1, b1, b2, b3);
        } else {
            split3(split_weight(w, layer, c, g, e, col), a1, a2, a3);
            split3(split_weight(w, layer, c, g, e + 1, col), b1, b2, b3);
        }
        if (f16 && layer <= 1 && c == kSpSlotChunk) {
            // the input-slot chunk's MIXED plane (see split_gemm3): k-groups 0 and 1 hold the slot's first weight pieces, k-group 2 the
            // second pieces, k-group 3 zeros -- against activation fragments (a1 | a2 | a1 | 0) ONE matrix instruction forms
            // w1 a1 + w1 a2 + w2 a1 of the slot's <= 8 inputs.  (The float16 form has no third piece: plane 2 is free.)
            uint32_t m1, m2, m3, n1, n2, n3;
            split3_f16(split_weight(w, layer, c, 0, e, col), m1, m2, m3);
            split3_f16(split_weight(w, layer, c, 0, e + 1, col), n1, n2, n3);
            a3 = g < 2 ? m1 : (g == 2 ? m2 : 0u);
            b3 = g < 2 ? n1 : (g == 2 ? n2 : 0u);
        }
        pl[0][e >> 1] = a1 | (b1 << 16); pl[1][e >> 1] = a2 | (b2 << 16); pl[2][e >> 1] = a3 | (b3 << 16);
    }
    const int64_t plane_stride = layer == 4 ? 64 : 16 * 64;
#pragma unroll
    for (int p = 0; p < 3; ++p) frags[base + p * plane_stride] = uint4{pl[p][0], pl[p][1], pl[p][2], pl[p][3]};
}

#endif

// Which of a tile's four 16-row tiles carry rows: all of them (the stand-alone kernel), or the first n (the fused actor kernel compacts the rows
// that still need an action to the front of the tile -- policy_split_tile, COMPACT -- and the GEMMs, epilogues and cell updates of the others are skipped;
// `n` is wave-uniform, every test a scalar branch)
struct SpAllRows { static constexpr bool dyn = false; __device__ __forceinline__ constexpr bool has(int) const { return true; } };
struct SpSomeRows { static constexpr bool dyn = true; int n; __device__ __forceinline__ bool has(int nt) const { return nt < n; } };
// ... the first NRT of them, known at compile time: every test folds away (the fused actor kernel instantiates the pass for NRT = 2, 3, 4 and
// picks one per tile and step by ONE scalar branch -- scalar tests around every group of matrix instructions cost more than the skipped
// row tiles saved: profiles/r05_l_row_compaction.txt)
template <int NRT> struct SpFirstRows { static constexpr bool dyn = true; __device__ __forceinline__ constexpr bool has(int nt) const { return nt < NRT; } };
struct SplitYes { static constexpr bool value = true; };
struct SplitNo { static constexpr bool value = false; };
struct SplitW { uint4 w[3][4]; };                            // weight fragments of one chunk: plane x column tile

template <bool F16 = false>
__device__ __forceinline__ f32x4 mfma_bf16(const uint4 &a, const uint4 &b, const f32x4 &c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Weight fragments and biases are read through BUFFER loads: resource descriptor + a scalar offset (layer, chunk, plane, wave: all
// uniform) + the lane's constant 32-bit offset -- no vector address arithmetic at all (the flat form spent ~170 64-bit vector adds per
// tile on addresses).  Offsets are in fragments (16 bytes) / floats.
struct SplitSrc { __amdgpu_buffer_rsrc_t w, b; };
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ SplitSrc split_src(const uint4 *sfrags, const float *bias) {
    return SplitSrc{__builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(sfrags), 0, (int)(kSpPackFrags8 * 16), 0x00020000),
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(bias), 0, (int)(kBiasFloats8 * sizeof(float)), 0x00020000)};
}
__device__ __forceinline__ uint4 split_buf16(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return uint4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ void split_load_w1(uint4 (&w)[4], const SplitSrc &src, int layer, int plane, int wave, int lane, int c) {
    const int soff = (layer + c * (int)kSpFragPerChunk + plane * 16 * 64 + (4 * wave) * 64) * 16;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) w[mt] = split_buf16(src.w, lane * 16, soff + mt * 1024);
}
template <int P>
__device__ __forceinline__ void split_load_w(SplitW &f, const SplitSrc &src, int layer, int wave, int lane, int c) {
#pragma unroll
    for (int pl = 0; pl < SplitFmt<P>::planes; ++pl) split_load_w1(f.w[pl], src, layer, pl, wave, lane, c);
}

// activation fragments (ONE plane) of the k-range starting at LDS column `col` (32 wide); `slot`: the input chunk --
// only k-group 0 holds data (one 8-value slot at column `col`), the other groups supply zeros
template <class RT = SpAllRows>
__device__ __forceinline__ void split_load_a(uint4 (&a)[4], const unsigned char *planes, int plane, int lane, int col, bool slot, RT rt = RT()) {
    const int g = lane >> 4;
    // (a chunk start is a multiple of 32 columns: sp_phys(col + 8 g) = col + the lane's constant part)
    const int off = slot ? (g == 0 ? sp_phys(col) : 2 * kSpZeroCol) : col + 16 * (g >> 1) + 256 * (g & 1);
    const unsigned char *p = planes + plane * kSpPlaneB + (lane & 15) * kSpStrideB + off;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) if (rt.has(nt)) a[nt] = *reinterpret_cast<const uint4 *>(p + nt * 16 * kSpStrideB);
}

// the MIXED activation fragments of the 8-wide input slot at LDS column `col`: k-group 0 = first pieces, 1 = second pieces, 2 = first
// pieces again, 3 = anything finite (its weights are zeros) -- the partner of the packed weights' mixed plane (policy_pack_split_kernel)
template <class RT = SpAllRows>
__device__ __forceinline__ void split_load_a_mix(uint4 (&a)[4], const unsigned char *planes, int lane, int col, RT rt = RT()) {
    const int g = lane >> 4;                               // (k-group 3 re-reads k-group 2's values: its weights are zeros and activations are finite)
    const unsigned char *p = planes + (g == 1 ? kSpPlaneB : 0) + (lane & 15) * kSpStrideB + sp_phys(col);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) if (rt.has(nt)) a[nt] = *reinterpret_cast<const uint4 *>(p + nt * 16 * kSpStrideB);
}

// one partial product for all 16 (column tile, row tile) pairs: consecutive MFMAs never share an accumulator
template <bool F16 = false, class RT = SpAllRows>
__device__ __forceinline__ void split_mfma_term(const uint4 (&w)[4], const uint4 (&a)[4], f32x4 (&acc)[4][4], RT rt = RT()) {
    if constexpr (RT::dyn) {                               // row tile outermost: one scalar test per row tile, four independent MFMAs behind it
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            if (rt.has(nt)) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = mfma_bf16<F16>(w[mt], a[nt], acc[mt][nt]);
            }
    } else {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma_bf16<F16>(w[mt], a[nt], acc[mt][nt]);
    }
}
// the first product of a GEMM: C operand = the layer's bias
template <bool F16, class RT>
__device__ __forceinline__ void split_mfma_first(const uint4 (&w)[4], const uint4 (&a)[4], f32x4 (&acc)[4][4], const f32x4 (&b4)[4], RT rt) {
    if constexpr (RT::dyn) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            if (rt.has(nt)) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = mfma_bf16<F16>(w[mt], a[nt], b4[mt]);
            }
    } else {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma_bf16<F16>(w[mt], a[nt], b4[mt]);
    }
}

// acc += W(chunks c0..c1-1 of `layer`) x act.  Chunk c reads LDS columns 32c.., except `slot_chunk`, which reads the
// 8-wide input slot at column `slot_col`.  w must already hold chunk c0's weights (issued before the barrier that
// publishes the activations).  ONE set of weight registers and ONE set of activation registers: every fragment group is
// re-loaded for the next chunk right after its last use in this one, ordered so that each load has >= 48 MFMAs (768
// cycles, L2) resp. >= 16 MFMAs (256 cycles, LDS) to land:
//     w1*a_lo   w1*a_hi   [w1 <- next]   w2*a_lo   [a_lo <- next]   w2*a_hi   [w2 <- next]   w3*a_hi   [w3, a_hi <- next]
// 144 live registers (64 accumulators, 48 weight, 32 activation) instead of 240 for a double-buffered pipeline.
//
// this lane's four bias values per column tile (columns 16*(4*wave+mt) + 4*(lane/16) + r); `bias` = the layer's first float
__device__ __forceinline__ void split_load_bias(f32x4 (&b4)[4], const SplitSrc &src, int bias, int wave, int lane) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const uint4 v = split_buf16(src.b, (lane >> 4) * 16, (bias + 64 * wave + 16 * mt) * 4);
        b4[mt] = f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
}

// P = 3 reads only two weight planes, so the third register group double-buffers plane 1 (the plane both of whose products come
// first): chunk c+1's w1 is requested at the START of chunk c into the idle buffer (48 MFMAs = 768 cycles to land instead of 16),
// w2 is re-loaded right after its one product (32 MFMAs), and the LAST chunk requests the first chunk of the layer that follows
// (`next_layer`, chunk `next_c`) into w.w[0] / w.w[1], which is where every call expects its first chunk.
// The accumulators are not initialised: the very first product takes the layer's bias (this lane's four columns per column
// tile, 16 registers: b4) as its C operand -- 64 register moves per GEMM less than broadcasting the bias into acc first.  b4 is
// loaded one GEMM ahead, like the first weight fragments: the last chunk requests `next_bias` into it.
template <bool F16, class RT = SpAllRows>
__device__ __forceinline__ void split_gemm3(const unsigned char *planes, const SplitSrc &src, int layer, int c0, int c1, int slot_chunk, int slot_col,
                                            int wave, int lane, SplitW &w, f32x4 (&acc)[4][4], int next_layer, int next_c,
                                            f32x4 (&b4)[4], int next_bias, RT rt = RT()) {
    // Round 5, float16 form: the input-slot chunk (<= 8 inputs in a K = 32 instruction; always the LAST chunk, c1 - 1) is ONE product
    // instead of three -- its three partial products laid side by side along K (mixed fragments: split_load_a_mix and the packed
    // weights' plane 2): 16 matrix instructions per slot chunk instead of 48, the same three partial sums in the same float32
    // accumulator.  A call that STARTS at the slot chunk (the first LSTM step) expects the mixed plane in w.w[0].
    const bool slot_last = F16 && slot_chunk >= 0;
    const int cm = slot_last ? c1 - 1 : c1;                // plain chunks: c0 .. cm - 1
    uint4 a_hi[4], a_lo[4];
    auto col_of = [&](int c) { return c == slot_chunk ? slot_col : 32 * c; };
    if (c0 >= cm) {                                        // (uniform) only the slot chunk: acc = bias + mixed product
        split_load_a_mix(a_lo, planes, lane, slot_col, rt);
        split_mfma_first<F16>(w.w[0], a_lo, acc, b4, rt);
        __builtin_amdgcn_sched_barrier(0);
        split_load_bias(b4, src, next_bias, wave, lane);
        split_load_w1(w.w[1], src, next_layer, 1, wave, lane, next_c);
        split_load_w1(w.w[0], src, next_layer, 0, wave, lane, next_c);
        __builtin_amdgcn_sched_barrier(0);
        return;
    }
    split_load_a(a_lo, planes, 1, lane, col_of(c0), c0 == slot_chunk, rt);
    split_load_a(a_hi, planes, 0, lane, col_of(c0), c0 == slot_chunk, rt);
    int c = c0;
    bool first = true;
#pragma unroll 1
    for (;;) {
        {                                                  // chunk c with w1 in w.w[0]; w.w[2] is idle
            const bool last = c + 1 >= cm;
            const bool to_slot = last && slot_last;        // the mixed product follows this chunk
            const int n = last ? c : c + 1;
            const int wl = last ? next_layer : layer;
            const int wc = last ? next_c : n;
            __builtin_amdgcn_sched_barrier(0);
            if (!last) split_load_w1(w.w[2], src, layer, 0, wave, lane, n);
            else if (to_slot) split_load_w1(w.w[2], src, layer, 2, wave, lane, cm);
            if (first) {                                   // (uniform) acc = bias + w1 * a_lo
                split_mfma_first<F16>(w.w[0], a_lo, acc, b4, rt);
                first = false;
            } else {
                split_mfma_term<F16>(w.w[0], a_lo, acc, rt);
            }
            if (last) split_load_bias(b4, src, next_bias, wave, lane);   // (b4 was consumed by the first product: free since then)
            __builtin_amdgcn_sched_barrier(0);
            if (to_slot) split_load_a_mix(a_lo, planes, lane, slot_col, rt);
            else split_load_a(a_lo, planes, 1, lane, col_of(n), n == slot_chunk, rt);
            split_mfma_term<F16>(w.w[1], a_hi, acc, rt);
            __builtin_amdgcn_sched_barrier(0);
            split_load_w1(w.w[1], src, wl, 1, wave, lane, wc);
            split_mfma_term<F16>(w.w[0], a_hi, acc, rt);
            __builtin_amdgcn_sched_barrier(0);
            if (!to_slot) split_load_a(a_hi, planes, 0, lane, col_of(n), n == slot_chunk, rt);
            if (last) {
                split_load_w1(w.w[0], src, next_layer, 0, wave, lane, next_c);
                if (to_slot) split_mfma_term<F16>(w.w[2], a_lo, acc, rt);
                break;
            }
        }
        ++c;
        {                                                  // chunk c with w1 in w.w[2]; w.w[0] is idle
            const bool last = c + 1 >= cm;
            const bool to_slot = last && slot_last;
            const int n = last ? c : c + 1;
            const int wl = last ? next_layer : layer;
            const int wc = last ? next_c : n;
            __builtin_amdgcn_sched_barrier(0);
            if (to_slot) split_load_w1(w.w[0], src, layer, 2, wave, lane, cm);      // the slot chunk's mixed plane: 48 matrix instructions ahead
            else split_load_w1(w.w[0], src, wl, 0, wave, lane, wc);
            split_mfma_term<F16>(w.w[2], a_lo, acc, rt);
            if (last) split_load_bias(b4, src, next_bias, wave, lane);
            __builtin_amdgcn_sched_barrier(0);
            if (to_slot) split_load_a_mix(a_lo, planes, lane, slot_col, rt);            // ... and its mixed activations: 32 ahead
            else split_load_a(a_lo, planes, 1, lane, col_of(n), n == slot_chunk, rt);
            split_mfma_term<F16>(w.w[1], a_hi, acc, rt);
            __builtin_amdgcn_sched_barrier(0);
            split_load_w1(w.w[1], src, wl, 1, wave, lane, wc);
            split_mfma_term<F16>(w.w[2], a_hi, acc, rt);
            __builtin_amdgcn_sched_barrier(0);
            if (!to_slot) split_load_a(a_hi, planes, 0, lane, col_of(n), n == slot_chunk, rt);
            if (last) {
                if (to_slot) {
                    split_mfma_term<F16>(w.w[0], a_lo, acc, rt);
                    __builtin_amdgcn_sched_barrier(0);
                    split_load_w1(w.w[0], src, next_layer, 0, wave, lane, next_c);
                }
                break;
            }
        }
        ++c;
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int P, class RT = SpAllRows>
__device__ __forceinline__ void split_gemm(const unsigned char *planes, const SplitSrc &src, int layer, int c0, int c1, int slot_chunk, int slot_col,
                                           int wave, int lane, SplitW &w, f32x4 (&acc)[4][4], int next_layer, int next_c,
                                           int bias, f32x4 (&b4)[4], int next_bias, RT rt = RT()) {
    if constexpr (P == 3 || P == kSpF16) {
        split_gemm3<P == kSpF16>(planes, src, layer, c0, c1, slot_chunk, slot_col, wave, lane, w, acc, next_layer, next_c, b4, next_bias, rt);
        return;
    }
    static_assert(!RT::dyn || P == 3 || P == kSpF16, "row compaction is carried by the three-product forms");
    {                                                      // acc[mt][nt] = bias of the lane's four columns 16*(4*wave+mt) + 4*(lane/16) + r
        f32x4 bb[4];
        split_load_bias(bb, src, bias, wave, lane);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = bb[mt];
    }
    uint4 a_hi[4], a_lo[4];
    auto col_of = [&](int c) { return c == slot_chunk ? slot_col : 32 * c; };
    split_load_a(a_lo, planes, 1, lane, col_of(c0), c0 == slot_chunk);
    split_load_a(a_hi, planes, 0, lane, col_of(c0), c0 == slot_chunk);
#pragma unroll 1
    for (int c = c0; c < c1; ++c) {
        const int n = c + 1 < c1 ? c + 1 : c;              // (the prefetches of the last chunk are harmless re-reads)
        __builtin_amdgcn_sched_barrier(0);
        split_mfma_term(w.w[0], a_lo, acc);
        split_mfma_term(w.w[0], a_hi, acc);
        __builtin_amdgcn_sched_barrier(0);
        split_load_w1(w.w[0], src, layer, 0, wave, lane, n);
        if (P >= 5) {
            split_mfma_term(w.w[1], a_lo, acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        split_load_a(a_lo, planes, 1, lane, col_of(n), n == slot_chunk);
        split_mfma_term(w.w[1], a_hi, acc);
        __builtin_amdgcn_sched_barrier(0);
        split_load_w1(w.w[1], src, layer, 1, wave, lane, n);
        if (P >= 4) {
            split_mfma_term(w.w[2], a_hi, acc);
            __builtin_amdgcn_sched_barrier(0);
            split_load_w1(w.w[2], src, layer, 2, wave, lane, n);
        }
        split_load_a(a_hi, planes, 0, lane, col_of(n), n == slot_chunk);
    }
    __builtin_amdgcn_sched_barrier(0);
    split_load_w<P>(w, src, next_layer, wave, lane, next_c);    // the first weight fragments of the layer that follows
}

// four consecutive columns of one row -> both planes, 8 bytes each
template <bool F16 = false>
__device__ __forceinline__ void split_store4(unsigned char *planes, int row, int col, const f32x4 &z) {
    uint32_t h0, l0, h1, l1;
    split2<F16>(z[0], z[1], h0, l0);
    split2<F16>(z[2], z[3], h1, l1);
    unsigned char *p = planes + row * kSpStrideB + sp_phys(col);
    *reinterpret_cast<uint2 *>(p) = uint2{h0, h1};
    *reinterpret_cast<uint2 *>(p + kSpPlaneB) = uint2{l0, l1};
}

template <bool F16 = false, class RT = SpAllRows>
__device__ __forceinline__ void split_store_relu(unsigned char *planes, int wave, int lane, const f32x4 (&acc)[4][4], RT rt = RT()) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        if (!rt.has(nt)) continue;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f32x4 z;
#pragma unroll
            for (int r = 0; r < 4; ++r) {                  // relu as ONE integer max on the bit pattern (negative floats are negative
                const float v = acc[mt][nt][r];            // integers; fmaxf costs a canonicalising max more).  (Through a scalar:
                const int bits = __float_as_int(v);        // __builtin_bit_cast straight on the vector element reads element 0.)
                if constexpr (F16) z[r] = __builtin_amdgcn_fmed3f(v, 0.0f, kSpF16Max);   // relu, saturating at float16's largest value
                else z[r] = __int_as_float(bits > 0 ? bits : 0);
            }
            split_store4<F16>(planes, 16 * nt + (lane & 15), 16 * (4 * wave + mt) + 4 * (lane >> 4), z);
        }
    }
}

struct SplitArgs {
    PolicyArgs p;                      // the float32 kernel's arguments (frags / bias are unused here)
    const uint4 *sfrags;               // split weight fragments
    const float *sbias;                // the biases in packed order, the LSTM gates' pre-scaled like their weight columns
};

// select_action (ProcessAgent.py:98-103) for the row whose softmax this 4-lane group holds (lane: columns 4g..4g+3 in pj): argmax
// (PLAY_MODE / EVALUATE_MODE) or one inverse-CDF draw from Philox4x32-10 keyed on (seed, global row, step).  Every lane of the
// group returns the same action.
__device__ __forceinline__ int split_select_action(const float (&pj)[4], int g, int lane, int A, bool greedy, int64_t row, int step,
                                                   uint32_t seed_lo, uint32_t seed_hi) {
    if (greedy) {                                          // np.argmax: first index of the maximum
        float best = fmaxf(fmaxf(pj[0], pj[1]), fmaxf(pj[2], pj[3]));
        best = fmaxf(best, __shfl_xor(best, 16, 64)); best = fmaxf(best, __shfl_xor(best, 32, 64));
        int idx = 99;
#pragma unroll
        for (int r = 3; r >= 0; --r) idx = (4 * g + r < A && pj[r] == best) ? 4 * g + r : idx;
        int o = __shfl_xor(idx, 16, 64); idx = o < idx ? o : idx;
        o = __shfl_xor(idx, 32, 64); idx = o < idx ? o : idx;
        return idx;
    }
    // inverse CDF: #{c : cdf_c <= u * cdf_{A-1}}
    // (the two partial sums are fenced from the final add: left alone the compiler makes it a packed add that reads its own result
    //  with the halves swapped -- v_pk_add_f32 ... op_sel:[0,1] -- the operand pattern that misbehaved next to matrix work in the
    //  env step, cavoid_kernels.hpp neighbour_features; same values, one scalar add)
    float s01 = pj[0] + pj[1], s23 = pj[2] + pj[3];
    asm volatile("" : "+v"(s01), "+v"(s23));
    const float t_g = s01 + s23;
    float before = 0.0f, total = 0.0f;                     // sum of the lower column groups / of all four, in group order
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
        const float tg = __shfl(t_g, (lane & 15) + 16 * gg, 64);
        before += gg < g ? tg : 0.0f;
        total += tg;
    }
    const uint32_t bits = policy_philox_x((uint32_t)row, (uint32_t)((uint64_t)row >> 32), (uint32_t)step, 0x504F4Cu, seed_lo, seed_hi);
    const float u = (float)(bits >> 8) * (1.0f / 16777216.0f);
    float cdf = before;
    int below = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) { cdf += pj[r]; below += (4 * g + r < A && cdf <= u * total) ? 1 : 0; }
    below += __shfl_xor(below, 16, 64); below += __shfl_xor(below, 32, 64);
    return below < A - 1 ? below : A - 1;
}

// The LSTM cell update of TWO cells at once (elements r, r+1 of a lane's accumulators: consecutive registers), on the packed
// float32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 issue at the plain instructions' rate for twice the values --
// tools/ubench/valu_rate_f32.hip: 5.0-5.1 clocks against 4.75-5.8; v_exp_f32 / v_rcp_f32: 8.75, no packed form):
//   sigmoid(x) = 1 / (1 + 2^(-x log2 e)),  tanh(x) = 1 - 2 / (1 + 2^(2 x log2 e));  c' = f c + i j;  h = o tanh(c')
// 10 transcendentals per cell and 9 packed operations per two cells (the gates' log2 e factors are folded into the weights).
__device__ __forceinline__ f32x2 sp_exp2(const f32x2 x) { f32x2 r; r[0] = __builtin_amdgcn_exp2f(x[0]); r[1] = __builtin_amdgcn_exp2f(x[1]); return r; }
__device__ __forceinline__ f32x2 sp_rcp(const f32x2 x) { f32x2 r; r[0] = __builtin_amdgcn_rcpf(x[0]); r[1] = __builtin_amdgcn_rcpf(x[1]); return r; }
__device__ __forceinline__ f32x2 sp_fma(const f32x2 a, const f32x2 b, const f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 sp_splat(float v) { f32x2 r; r[0] = v; r[1] = v; return r; }
__device__ __forceinline__ void split_lstm_cell2(const f32x2 xi, const f32x2 xj, const f32x2 xf, const f32x2 xo, const f32x2 c_old, f32x2 &c_new,
                                                 f32x2 &h_new) {
    // (xi, xf, xo arrive multiplied by log2 e, xj by 2 log2 e: split_gate_scale)
    // The cell state is carried MULTIPLIED BY 2 log2 e (c~ = 2 log2 e c: what tanh's 2^x wants; it is never read by anything else), which
    // the j gate supplies for free: tanh(j) 2 log2 e = k - 2k / (1 + 2^xj) is the same fma with other constants.
    constexpr float k = 2.0f * kSpLog2e;
    const f32x2 one = sp_splat(1.0f), m2 = sp_splat(-2.0f);
    const f32x2 gi = sp_rcp(sp_exp2(-xi) + one), gf = sp_rcp(sp_exp2(-xf) + one), go = sp_rcp(sp_exp2(-xo) + one);
    const f32x2 gj = sp_fma(sp_rcp(sp_exp2(xj) + one), sp_splat(-2.0f * k), sp_splat(k));
    c_new = sp_fma(gf, c_old, gi * gj);
    const f32x2 tc = sp_fma(sp_rcp(sp_exp2(c_new) + one), m2, one);
    h_new = go * tc;
}

}  // namespace cavoid
#include "cavoid_policy_pipe.hpp"
namespace cavoid {

// The forward pass of ONE 64-row tile by the 4 wavefronts of a workgroup (layout: header of this file).
//   load(r, k)  -> policy input k (0 = num_other_agents) of tile row r, r < rows_here; the rows may live in global memory (the
//                  stand-alone kernel) or in LDS (the fused actor kernel: the env step left them there);
//   emit(trow, g, pj, logit) is called by every lane of the heads' layout: tile row trow = 16 wave + lane%16, columns 4g..4g+3 --
//                  pj = softmax probabilities incl. MIN_POLICY, logit[r] = raw head output (column A = the value).
// planes / len_f / wave_max: the workgroup's LDS (policy_split_lds_bytes()).  Contains workgroup barriers: every thread calls it.
// COMPACT (the fused actor kernel): `live(r)` says which tile rows still need an action (a learning agent that has not finished: what
// cavoid_rollout_active_rows lists for the step-by-step path; a finished agent waits for its world's last learning agent, the env ignores
// whatever it is given).  The live rows are packed to the front of the tile -- input row r goes to tile row popcount(live rows below r); rmap[64]
// (LDS) holds the way back for emit() -- and only the first ceil(n_live / 16) row tiles are computed: GEMMs, cell updates, epilogues and heads of
// the others are skipped by scalar branches.  Rows are independent in every layer, so a live row's outputs are bit for bit what the full pass
// gives it; emit() is called for live rows only, with their ORIGINAL tile row (the action draw is keyed on it).
// NRT > 0 = COMPACT with the first NRT row tiles computed: `live_mask` (bit r: tile row r still needs an action; the same value in every
// wavefront; at least one bit, at most 16 NRT) comes from the caller, which picks the instantiation.
// PIPE: the LSTM steps and layer1 run as a software pipeline over row halves, their cell updates / epilogue inside the other half's matrix
// instructions (cavoid_policy_pipe.hpp); same results.
template <int P, int NRT = 0, bool PIPE = false, class Load, class Emit>
__device__ __forceinline__ void policy_split_tile(const SplitArgs &sa, unsigned char *planes, float *len_f, int *wave_max, int rows_here,
                                                  int tid, Load load, Emit emit, unsigned long long live_mask = ~0ull, int *rmap = nullptr,
                                                  int steps_total = -1) {
    const PolicyArgs &p = sa.p;
    constexpr bool F16 = SplitFmt<P>::f16;
    constexpr bool COMPACT = NRT > 0;
    using RT = typename std::conditional<COMPACT, SpFirstRows<(NRT > 0 ? NRT : 4)>, SpAllRows>::type;
    // (wave in a scalar register: every weight / bias address is then a uniform base + this lane's constant 32-bit offset, and the
    //  fragment loads need no vector address arithmetic)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4;
    const int M = p.max_other, A = p.num_actions;
    const SplitSrc src = split_src(sa.sfrags, sa.sbias);
    constexpr int w_lstm = (int)kSpOffLstm;
    int n_live = 64, pos = tid & 63;
    bool mine = (tid & 63) < rows_here;
    const RT rt{};
    if constexpr (COMPACT) {
        mine = (live_mask >> (tid & 63)) & 1ull;
        n_live = __popcll(live_mask);
        pos = __popcll(live_mask & ((1ull << (tid & 63)) - 1ull));
    }
    SplitW f0;
    f32x4 b4[4];                                           // the bias of the GEMM that comes next (P = 3: its first product's C operand)
    // first LSTM step: h == 0, only the input chunk contributes (float16 form: its mixed plane, one product)
    static_assert(!PIPE || F16, "the pipelined form carries the float16 pieces");
    PpW wb[2];                                             // (PIPE) the [h | slot] layer's weight fragments: chunk 0, chunk 1,
    uint4 wslot[4];                                        //        the slot chunk's mixed plane
    if constexpr (PIPE) split_load_w1(wslot, src, w_lstm, 2, wave, lane, kSpSlotChunk);
    else if constexpr (F16) split_load_w1(f0.w[0], src, w_lstm, 2, wave, lane, kSpSlotChunk);
    else split_load_w<P>(f0, src, w_lstm, wave, lane, kSpSlotChunk);
    split_load_bias(b4, src, kBiasLstm, wave, lane);

    // ---- input tile: gather + normalise + split into the slot columns ---------------------------------------------
    {
        int local_max = 0, local_min = 0;
        if (tid < 64) {
            const float v = (COMPACT ? mine : tid < rows_here) ? load(tid, 0) : 0.0f;
            if constexpr (COMPACT) {                        // tile row `pos` takes input row `tid`; the tile rows behind the live ones are empty
                if (mine) { len_f[pos] = v; rmap[pos] = tid; }
                if (tid >= n_live) len_f[tid] = 0.0f;
            } else {
                len_f[tid] = v;
            }
            int len = (int)v;
            len = len < 0 ? 0 : (len > M ? M : len);
            local_max = len;
            local_min = v >= (float)len ? len : len - 1;   // (a fractional count: the row's last step is live only up to floor)
            local_min = local_min < 0 ? 0 : local_min;
            if (COMPACT && !mine) local_min = M;           // (the empty tile rows do not decide whether every row is live at step t)
        }
        // h = 0 (columns 0..63 of both planes)
        for (int e = tid; e < 2 * 64 * 8; e += 256) {
            const int pl = e >> 9, r = (e >> 3) & 63, c16 = e & 7;
            *reinterpret_cast<uint4 *>(planes + pl * kSpPlaneB + r * kSpStrideB + sp_phys(8 * c16)) = uint4{0u, 0u, 0u, 0u};
        }
        if (tid < 128)                                      // the zero column (256..263) of every row of both planes
            *reinterpret_cast<uint4 *>(planes + (tid >> 6) * kSpPlaneB + (tid & 63) * kSpStrideB + kSpZeroCol * 2) = uint4{0u, 0u, 0u, 0u};
        const int items = 64 * (M + 1);                    // (row, slot): slot 0 = host (4 values), slot s = observed agent s-1 (7)
        for (int it = tid; it < items; it += 256) {
            const int r = it & 63, s = it >> 6;
            const int n_in = s == 0 ? kPolHost : kPolOther, sc0 = s == 0 ? 1 : 1 + kPolHost + kPolOther * (s - 1);
            // all eight loads (and the sixteen of avg / std) are issued before the first use: unused elements re-read element 0
            // of the slot and rows past the end re-read row 0 -- valid addresses, selected away below -- so nothing is conditional
            // COMPACT: r == this thread's lane; a live row writes tile row `pos`, a lane at or behind n_live zeroes tile row `lane`, the
            // lanes in between (not live, below n_live) own no tile row; a LIVE lane at or behind n_live also zeroes tile row `lane` (below)
            const bool row_ok = COMPACT ? mine : r < rows_here;
            if (COMPACT && !mine && r < n_live) continue;
            const int rr = row_ok ? r : 0;
            float v[8], av[8], sd[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = load(rr, e < n_in ? sc0 + e : sc0);
            if (p.avg) {                                   // uniform
#pragma unroll
                for (int e = 0; e < 8; ++e) { av[e] = p.avg[e < n_in ? sc0 + e : sc0]; sd[e] = p.std[e < n_in ? sc0 + e : sc0]; }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (v[e] - av[e]) * __builtin_amdgcn_rcpf(sd[e]);   // (v_rcp_f32, 1 ulp: ten instructions less per value than a float32 division)
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (row_ok && e < n_in) ? v[e] : 0.0f;
            if constexpr (F16) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], -kSpF16Max, kSpF16Max);
            }
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split2<F16>(v[2 * e], v[2 * e + 1], hi[e], lo[e]);
            unsigned char *d = planes + ((COMPACT && mine) ? pos : r) * kSpStrideB + sp_phys(kSpSlotCol + 8 * s);
            *reinterpret_cast<uint4 *>(d) = uint4{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<uint4 *>(d + kSpPlaneB) = uint4{lo[0], lo[1], lo[2], lo[3]};
            if (COMPACT && mine && r >= n_live) {           // a live lane behind the packed rows: tile row `lane` is nobody's -- it takes zeros like the rest of the tail
                unsigned char *z = planes + r * kSpStrideB + sp_phys(kSpSlotCol + 8 * s);
                *reinterpret_cast<uint4 *>(z) = uint4{0u, 0u, 0u, 0u};
                *reinterpret_cast<uint4 *>(z + kSpPlaneB) = uint4{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_xor(local_max, d, 64), q = __shfl_xor(local_min, d, 64);
            local_max = o > local_max ? o : local_max;
            local_min = q < local_min ? q : local_min;
        }
        if (lane == 0) { wave_max[wave] = local_max; wave_max[5 + wave] = local_min; }
    }
    __syncthreads();
    const int steps = wave_max[0];                         // rows 0..63 are all in wavefront 0's threads
    const int tile_min_len = wave_max[5];                  // every row of the tile (idle tail rows: 0) has more than t observed agents while t < this
    POLICY_STAMP(5);

    // this lane's rows (one per row tile) and their sequence lengths
    float len_r[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) len_r[nt] = len_f[16 * nt + (lane & 15)];

    // ---- LSTM over the observed agents ----------------------------------------------------------------------------
    f32x4 cell[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) cell[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (PIPE) {
        // ---- LSTM steps and layer1 in row halves: H0 = row tiles 0 .. NT0-1 (accumulators A), H1 = the rest (accumulators B) ----------
        constexpr int NRTc = COMPACT ? NRT : 4, NT0 = (NRTc + 1) / 2, NT1 = NRTc - NT0;
        static_assert(NT1 >= 1 && NT0 <= 2, "two to four row tiles");
        f32x4 accA[4][2], accB[4][2];
        auto load_wb = [&](int layer) {
#pragma unroll
            for (int c = 0; c < 2; ++c) { split_load_w1(wb[c].p0, src, layer, 0, wave, lane, c); split_load_w1(wb[c].p1, src, layer, 1, wave, lane, c); }
        };
        const int l1 = (int)kSpOffL1;
        if (steps > 0) {
            load_wb(steps > 1 ? w_lstm : l1);              // (the first bracket that has plain chunks: LSTM step 1, or layer1)
            // t = 0: h == 0, the slot's product only; no barrier between the halves (nothing reads h yet)
            pp_bracket<NT0, false, false>(planes, src, kSpSlotCol + 8, wave, lane, 0, wb, wslot, accA, b4, 0, kBiasLstm, PpNoFill{});
            pp_bracket<NT1, false, true>(planes, src, kSpSlotCol + 8, wave, lane, NT0, wb, wslot, accB, b4, steps > 1 ? w_lstm : l1,
                                         steps > 1 ? kBiasLstm : kBiasL1, PpCellFill<NT0, 0>{planes, accA, cell, len_f, 0, wave, lane});
            __syncthreads();
#pragma unroll 1
            for (int t = 1; t < steps; ++t) {
                if (t == 1) POLICY_STAMP(8);
                pp_bracket<NT0, true, false>(planes, src, kSpSlotCol + 8 * (1 + t), wave, lane, 0, wb, wslot, accA, b4, 0, kBiasLstm,
                                             PpCellFill<NT1, NT0>{planes, accB, cell, len_f, t - 1, wave, lane});
                if (t == 1) POLICY_STAMP(9);
                __syncthreads();
                if (t == 1) POLICY_STAMP(10);
                const bool last = t + 1 >= steps;
                pp_bracket<NT1, true, true>(planes, src, kSpSlotCol + 8 * (1 + t), wave, lane, NT0, wb, wslot, accB, b4, last ? l1 : w_lstm,
                                            last ? kBiasL1 : kBiasLstm, PpCellFill<NT0, 0>{planes, accA, cell, len_f, t, wave, lane});
                if (t == 1) POLICY_STAMP(11);
                __syncthreads();
                if (t == 1) POLICY_STAMP(12);
            }
            POLICY_STAMP(1);
            // layer1, H0 -- beside the last cell update of H1
            pp_bracket<NT0, true, false>(planes, src, kSpSlotCol, wave, lane, 0, wb, wslot, accA, b4, 0, kBiasL1,
                                         PpCellFill<NT1, NT0>{planes, accB, cell, len_f, steps - 1, wave, lane});
        } else {
            POLICY_STAMP(1);
            load_wb(l1);
            split_load_w1(wslot, src, l1, 2, wave, lane, kSpSlotChunk);
            split_load_bias(b4, src, kBiasL1, wave, lane);
            pp_bracket<NT0, true, false>(planes, src, kSpSlotCol, wave, lane, 0, wb, wslot, accA, b4, 0, kBiasL1, PpNoFill{});
        }
        __syncthreads();
        // layer1, H1 -- beside the relu + split of H0
        pp_bracket<NT1, true, false>(planes, src, kSpSlotCol, wave, lane, NT0, wb, wslot, accB, b4, 0, kBiasL2, PpReluFill<NT0, 0>{planes, accA, wave, lane});
        split_load_w1(f0.w[0], src, (int)kSpOffL2, 0, wave, lane, 0);      // layer2's first fragments (its bias is on its way), as split_gemm expects them
        split_load_w1(f0.w[1], src, (int)kSpOffL2, 1, wave, lane, 0);
        __syncthreads();                                   // every wavefront has read H1's [h | host]
        {
            const PpReluFill<NT1, NT0> tail{planes, accB, wave, lane};
#pragma unroll
            for (int u = 0; u < PpReluFill<NT1, NT0>::units; ++u) tail.unit(u);
        }
        __syncthreads();
    } else {
    // One LSTM step; ALL_LIVE (a type: two separately compiled bodies): every row of the tile has more than t observed agents -- the usual
        // case in full worlds -- so the cell update needs no selects.  (As ONE loop with a uniform `all_live` flag the compiler folded the two
        // cases into selects on every cell register plus ~100 register moves per step at the loop's end: 15 % of the tile's vector instructions.)
        auto lstm_step = [&](const int t, auto all_live_c) {
            constexpr bool ALL_LIVE = decltype(all_live_c)::value;
            f32x4 acc[4][4];
            if (t == 1) POLICY_STAMP(8);
            split_gemm<P>(planes, src, w_lstm, t == 0 ? 2 : 0, kSpChLstm, 2, kSpSlotCol + 8 * (1 + t), wave, lane, f0, acc,
                          t + 1 < steps ? w_lstm : (int)kSpOffL1, 0, kBiasLstm, b4,
                          t + 1 < steps ? kBiasLstm : kBiasL1, rt);             // (requests the next step's / layer1's first fragments and bias)
            if (t == 1) POLICY_STAMP(9);
            __syncthreads();                                   // every wavefront has read h
            if (t == 1) POLICY_STAMP(10);
            // lane: row 16nt + l%16, hidden units 16w + 4g + r; column tile = gate (i, j, f, o).  dynamic_rnn: rows past their own
            // length keep (c, h)
    #pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                if (!rt.has(nt)) continue;
                const bool live = ALL_LIVE || len_r[nt] > (float)t;
                f32x4 h_new;
    #pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    f32x2 c2, h2;
                    split_lstm_cell2(f32x2{acc[0][nt][r], acc[0][nt][r + 1]}, f32x2{acc[1][nt][r], acc[1][nt][r + 1]},
                                     f32x2{acc[2][nt][r], acc[2][nt][r + 1]}, f32x2{acc[3][nt][r], acc[3][nt][r + 1]},
                                     f32x2{cell[nt][r], cell[nt][r + 1]}, c2, h2);
                    h_new[r] = h2[0]; h_new[r + 1] = h2[1];
                    if constexpr (ALL_LIVE) { cell[nt][r] = c2[0]; cell[nt][r + 1] = c2[1]; }
                    else { cell[nt][r] = live ? c2[0] : cell[nt][r]; cell[nt][r + 1] = live ? c2[1] : cell[nt][r + 1]; }
                }
                if (live) split_store4<F16>(planes, 16 * nt + (lane & 15), 16 * wave + 4 * g, h_new);   // (else h stays as it is)
            }
            if (t == 1) POLICY_STAMP(11);
            __syncthreads();                                   // the new h is in place
            if (t == 1) POLICY_STAMP(12);
        };
        {
            const int t_all = tile_min_len < steps ? tile_min_len : steps;
            int t = 0;
    #pragma unroll 1
            for (; t < t_all; ++t) lstm_step(t, SplitYes{});
    #pragma unroll 1
            for (; t < steps; ++t) lstm_step(t, SplitNo{});
            // (the paired form, policy_forward_split_duo_kernel: the workgroup's other tile has more LSTM steps -- keep its barrier count)
    #pragma unroll 1
            for (; t < steps_total; ++t) { __syncthreads(); __syncthreads(); }
        }
        POLICY_STAMP(1);
        // ---- layer1 on [h | host] -------------------------------------------------------------------------------------
        {
            f32x4 acc[4][4];
            if (steps == 0) {                                  // (else the last LSTM step asked for them)
                split_load_w<P>(f0, src, (int)kSpOffL1, wave, lane, 0);
                split_load_bias(b4, src, kBiasL1, wave, lane);
            }
            split_gemm<P>(planes, src, (int)kSpOffL1, 0, kSpChL1, 2, kSpSlotCol, wave, lane, f0, acc, (int)kSpOffL2, 0, kBiasL1, b4, kBiasL2, rt);
            __syncthreads();
            split_store_relu<F16>(planes, wave, lane, acc, rt);
            __syncthreads();
        }
    }
    POLICY_STAMP(2);
    // ---- layer2, fullyconnected1 ----------------------------------------------------------------------------------
    {
        f32x4 acc[4][4];
        split_gemm<P>(planes, src, (int)kSpOffL2, 0, kSpChWide, -1, 0, wave, lane, f0, acc, (int)kSpOffFc1, 0, kBiasL2, b4, kBiasFc1, rt);
        __syncthreads();
        split_store_relu<F16>(planes, wave, lane, acc, rt);
        __syncthreads();
    }
    uint4 hw[kSpChWide][3];                                // the heads' weight fragments: half in flight across the epilogue
    auto head_frag = [&](int c, int pl) { return split_buf16(src.w, lane * 16, ((int)kSpOffHead + (c * 3 + pl) * 64) * 16); };
    {
        f32x4 acc[4][4];
        split_gemm<P>(planes, src, (int)kSpOffFc1, 0, kSpChWide, -1, 0, wave, lane, f0, acc, (int)kSpOffFc1, kSpChWide - 1, kBiasFc1, b4, kBiasFc1, rt);
#pragma unroll
        for (int c = 0; c < kSpChWide / 2; ++c)
#pragma unroll
            for (int pl = 0; pl < SplitFmt<P>::planes; ++pl) hw[c][pl] = head_frag(c, pl);
        __syncthreads();
        split_store_relu<F16>(planes, wave, lane, acc, rt);
        __syncthreads();
    }
    POLICY_STAMP(3);
    // ---- heads: wavefront w does rows 16w..16w+15 x 16 columns (A logits, the value, padding) ---------------------
    if (rt.has(wave)) {                                    // (COMPACT: a wavefront whose row tile is empty has no heads to make)
#pragma unroll
        for (int c = kSpChWide / 2; c < kSpChWide; ++c)
#pragma unroll
            for (int pl = 0; pl < SplitFmt<P>::planes; ++pl) hw[c][pl] = head_frag(c, pl);
        f32x4 acc[5];
        acc[0] = *reinterpret_cast<const f32x4 *>(p.bias + kBiasHead + 4 * g);
        acc[1] = acc[2] = acc[3] = acc[4] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char *arow = planes + (16 * wave + (lane & 15)) * kSpStrideB + sp_phys(8 * g);
#pragma unroll
        for (int c = 0; c < kSpChWide; ++c) {
            const uint4 a1 = *reinterpret_cast<const uint4 *>(arow + c * 32), a2 = *reinterpret_cast<const uint4 *>(arow + kSpPlaneB + c * 32);   // (sp_phys: 32 bytes per chunk)
            if (P == 4 || P == 5) acc[4] = mfma_bf16(hw[c][2], a1, acc[4]);
            if (P == 5) acc[3] = mfma_bf16(hw[c][1], a2, acc[3]);
            acc[2] = mfma_bf16<F16>(hw[c][1], a1, acc[2]);
            acc[1] = mfma_bf16<F16>(hw[c][0], a2, acc[1]);
            acc[0] = mfma_bf16<F16>(hw[c][0], a1, acc[0]);
        }
        const f32x4 logit = (acc[4] + acc[3]) + (acc[2] + acc[1]) + acc[0];
        // lane: row 16w + l%16, columns 4g + r.  Reductions over a row's 16 columns = over r in the lane and over the
        // four lanes l%16 + 16g'.
        const int trow = 16 * wave + (lane & 15);
        const float scale = 1.0f / (1.0f + p.min_policy * (float)A);
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) m = (4 * g + r < A) ? fmaxf(m, logit[r]) : m;
        m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
        float e[4], sum = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { e[r] = (4 * g + r < A) ? expf(logit[r] - m) : 0.0f; sum += e[r]; }
        sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
        float pj[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pj[r] = (4 * g + r < A) ? (e[r] / sum + p.min_policy) * scale : 0.0f;
        if constexpr (COMPACT) {                            // the tile row's input row; only live rows are emitted
            if (trow < n_live) emit(rmap[trow], g, pj, logit);
        } else {
            emit(trow, g, pj, logit);
        }
    }
    POLICY_STAMP(4);
}

template <int P, bool PIPE = false>
__global__ void __launch_bounds__(256, 2) policy_forward_split_kernel(const SplitArgs sa) {
    const PolicyArgs &p = sa.p;
    extern __shared__ __attribute__((aligned(16))) unsigned char planes[];      // plane 1 (hi), plane 2 (lo)
    float *len_f = reinterpret_cast<float *>(planes + 2 * kSpPlaneB);            // [64] raw num_other_agents
    int *tile_row = reinterpret_cast<int *>(len_f + 64);                          // [64] global row of each tile row
    int *wave_max = tile_row + 64;                                                // [4] + ticket
    int &ticket = wave_max[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    const int64_t n_rows = p.row_count ? (int64_t)*p.row_count : p.rows;
    const int rows_here = n_rows - row0 < 64 ? (int)(n_rows - row0 > 0 ? n_rows - row0 : 0) : 64;
    const int A = p.num_actions;
    const int step = p.actions_out ? *p.step_counter : 0;
    const bool listed = p.row_index != nullptr;
    if (listed && rows_here == 0) {                        // uniform over the workgroup: nothing listed for this tile
        if (p.actions_out) policy_finish(p, step, tid);
        return;
    }
    POLICY_STAMP(0);
#ifdef CAVOID_TRACE
    const unsigned long long trace_c0 = clock64();
    if (tid == 0 && g_pol_trace)
        g_pol_trace[(size_t)blockIdx.x * 16 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                                  ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
#endif
    if (tid < 64) tile_row[tid] = listed ? (tid < rows_here ? p.row_index[row0 + tid] : 0) : tid;   // row of `src` behind each tile row
    if (tid == 0) {                                        // arrival parity on the CU -> static priority (see cavoid_policy.hpp)
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        const uint32_t key = ((xcc & 15u) << 8) | ((hw >> 8) & 0xFFu);
        ticket = (int)atomicAdd(p.cu_tickets + key, 1u);
    }
    __syncthreads();
    if (ticket & 1) __builtin_amdgcn_s_setprio(1);
    const float *src = listed ? p.x : p.x + row0 * p.stride;
    auto load = [&](int r, int k) -> float { return src[(int64_t)tile_row[r] * p.stride + k]; };
    auto emit = [&](int trow, int g, const float (&pj)[4], const f32x4 &logit) {
        const bool in_tile = trow < rows_here;
        const int64_t row = listed ? (in_tile ? (int64_t)tile_row[trow] : p.rows) : row0 + trow;
        if (row < p.rows) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = 4 * g + r;
                if (col < A) p.p_out[row * A + col] = pj[r];
                else if (col == A) p.v_out[row] = logit[r];
            }
        }
        if (p.actions_out) {                               // wave-uniform
            const int action = split_select_action(pj, g, lane, A, p.greedy != 0, row, step, p.seed_lo, p.seed_hi);
            if (row < p.rows && g == 0) p.actions_out[row] = action;
        }
    };
    policy_split_tile<P, 0, PIPE>(sa, planes, len_f, wave_max, rows_here, tid, load, emit);
#ifdef CAVOID_TRACE
    if (tid == 0 && g_pol_trace) g_pol_trace[(size_t)blockIdx.x * 16 + 6] = clock64() - trace_c0;   // shader-clock cycles
#endif
    if (p.actions_out) policy_finish(p, step, tid);
}

// ---- the PAIRED form: two 64-row tiles per workgroup, their phases locked one barrier apart ------------------------------------------------
// policy_split_tile alternates matrix phases (a layer's GEMM) and vector phases (LSTM cell update / relu + split epilogue), a workgroup barrier
// between them.  With two independent workgroups per CU the two tiles of a CU drift: whether one tile's vector phase meets the other's matrix
// phase -- the only way the two pipes of a SIMD work at the same time (tools/ubench/mfma32_valu_overlap.hip: a matrix stream at full rate with
// the partner wavefront issuing a vector instruction every 8-16 clocks) -- is chance (profiles/r05_m_policy_phase_trace.txt: one tile alone on a
// CU 27.5 us, two drifting tiles 39.5 / 46 us).  Here ONE workgroup of eight wavefronts owns both tiles (wavefronts 0..3 tile 2b, 4..7 tile
// 2b+1: each SIMD holds one wavefront of either), every barrier of the pass is the workgroup's, and the second tile starts ONE barrier late: for
// the whole pass tile A's matrix phases run beside tile B's vector phases and vice versa.  The per-tile statements are policy_split_tile's,
// untouched: results are bit for bit the unpaired kernel's.  (A tile with fewer LSTM steps than its partner idles through the difference.)
constexpr size_t policy_split_duo_lds_bytes() { return 2 * policy_split_lds_bytes() + 16; }

template <int P>
__global__ void __launch_bounds__(512, 1) policy_forward_split_duo_kernel(const SplitArgs sa) {
    const PolicyArgs &p = sa.p;
    extern __shared__ __attribute__((aligned(16))) unsigned char duo_lds[];
    const int tid_wg = threadIdx.x;
    const int half = __builtin_amdgcn_readfirstlane(tid_wg >> 8), tid = tid_wg & 255, lane = tid & 63;
    unsigned char *planes = duo_lds + (size_t)half * policy_split_lds_bytes();
    float *len_f = reinterpret_cast<float *>(planes + 2 * kSpPlaneB);
    int *tile_row = reinterpret_cast<int *>(len_f + 64);
    int *wave_max = tile_row + 64;
    int *pair_steps = reinterpret_cast<int *>(duo_lds + 2 * policy_split_lds_bytes());
    const int64_t row0 = ((int64_t)blockIdx.x * 2 + half) * 64;
    const int64_t n_rows = p.row_count ? (int64_t)*p.row_count : p.rows;
    const int rows_here = n_rows - row0 < 64 ? (int)(n_rows - row0 > 0 ? n_rows - row0 : 0) : 64;
    const int A = p.num_actions;
    const int step = p.actions_out ? *p.step_counter : 0;
    const bool listed = p.row_index != nullptr;
    if (listed && n_rows <= (int64_t)blockIdx.x * 128) {  // uniform over the workgroup: nothing listed for either tile
        if (p.actions_out) policy_finish(p, step, tid_wg);
        return;
    }
    POLICY_STAMP(0);
    if (tid < 64) tile_row[tid] = listed ? (tid < rows_here ? p.row_index[row0 + tid] : 0) : (tid < rows_here ? tid : 0);
    const float *src = listed ? p.x : p.x + (rows_here > 0 ? row0 : 0) * p.stride;   // (an empty tile re-reads row 0 of the batch: valid, selected away)
    // the number of LSTM steps of either tile, before the phase offset (the tile function computes its own again: one more trip to the row)
    if (tid < 64) {
        int len = tid < rows_here ? (int)src[(int64_t)tile_row[tid] * p.stride] : 0;
        len = len < 0 ? 0 : (len > p.max_other ? p.max_other : len);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_xor(len, d, 64); len = o > len ? o : len; }
        if (lane == 0) pair_steps[half] = len;
    }
    __syncthreads();
    const int steps_total = pair_steps[0] > pair_steps[1] ? pair_steps[0] : pair_steps[1];
    auto load = [&](int r, int k) -> float { return src[(int64_t)tile_row[r] * p.stride + k]; };
    auto emit = [&](int trow, int g, const float (&pj)[4], const f32x4 &logit) {
        const bool in_tile = trow < rows_here;
        const int64_t row = listed ? (in_tile ? (int64_t)tile_row[trow] : p.rows) : (in_tile ? row0 + trow : p.rows);
        if (row < p.rows) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = 4 * g + r;
                if (col < A) p.p_out[row * A + col] = pj[r];
                else if (col == A) p.v_out[row] = logit[r];
            }
        }
        if (p.actions_out) {                               // wave-uniform
            const int action = split_select_action(pj, g, lane, A, p.greedy != 0, row, step, p.seed_lo, p.seed_hi);
            if (row < p.rows && g == 0) p.actions_out[row] = action;
        }
    };
    if (half == 1) __syncthreads();                        // the second tile runs one phase behind the first
    policy_split_tile<P>(sa, planes, len_f, wave_max, rows_here, tid, load, emit, ~0ull, nullptr, steps_total);
    if (half == 0) __syncthreads();
    if (p.actions_out) policy_finish(p, step, tid_wg);
}

}  // namespace cavoid
