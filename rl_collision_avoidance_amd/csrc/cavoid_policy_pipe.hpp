// cavoid_policy_pipe.hpp -- the LSTM steps and layer1 of policy_split_tile as a software pipeline over ROW HALVES (included by
// cavoid_policy_split.hpp in front of policy_split_tile; PIPE = true selects it).
//
// policy_split_tile alternates a matrix phase (a layer's GEMM for the tile's four row tiles) and a vector phase (LSTM cell update / relu +
// split epilogue), a workgroup barrier between them; tools/ubench/mfma_fill.hip: between two matrix instructions of ONE wavefront two plain
// vector instructions (or one transcendental) issue for free -- 16.6 clocks per v_mfma_f32_16x16x32_f16 with two v_fma_f32 behind each against
// 16.45 bare, and the same with a second wavefront on the SIMD (24.6 against 24.5) -- so a vector phase that runs INSIDE a matrix phase of the
// same wavefront costs (nearly) nothing.  Inside one GEMM that is impossible (the epilogue needs the finished sums, the next GEMM all columns
// of the epilogue's output), but rows are independent: with the tile's rows cut into halves H0 (row tiles 0 .. NT0-1) and H1 (the rest),
//
//     bracket (t, H0):   GEMM of step t for H0 -> accumulators A     beside   cell update of step t-1 for H1 (accumulators B)
//     -- barrier --
//     bracket (t, H1):   GEMM of step t for H1 -> accumulators B     beside   cell update of step t for H0 (accumulators A)
//     -- barrier --
//
// -- the same two barriers per step as before, every LDS access of a bracket's two halves on disjoint rows.  The LSTM's (and layer1's) whole
// weight set of a wavefront -- two plain K = 32 chunks x two planes + the input slot's mixed plane: 80 registers -- stays in registers across
// the two brackets of a step, so the halves cost NO second trip to the L2 for weights (the wide layers' 8 chunks do not fit: they keep the
// one-GEMM-per-layer form).  layer1 follows the same scheme: its H0 bracket runs beside the last cell update, its H1 bracket beside the
// relu + split of H0.  Every output element is the same float32 sum in the same order (bias + per chunk w1 a2 + w2 a1 + w1 a1, the slot's
// mixed product last) and the cell update / epilogue are the same statements: bit-identical results.
#pragma once

namespace cavoid {

struct PpW { uint4 p0[4], p1[4]; };                          // one plain chunk's weight fragments: plane 0 (first pieces), plane 1 (second pieces) x 4 column tiles

// activation fragments of row tiles nt0 .. nt0 + NT - 1 (ONE plane) for the K range starting at LDS column `col` (a multiple of 32)
template <int NT>
__device__ __forceinline__ void pp_load_a(uint4 (&a)[2], const unsigned char *planes, int plane, int lane, int col, int nt0) {
    const int g = lane >> 4;
    const unsigned char *p = planes + plane * kSpPlaneB + ((lane & 15) + 16 * nt0) * kSpStrideB + col + 16 * (g >> 1) + 256 * (g & 1);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) a[nt] = *reinterpret_cast<const uint4 *>(p + nt * 16 * kSpStrideB);
}
// ... the MIXED fragments of the 8-wide input slot at LDS column `col` (split_load_a_mix)
template <int NT>
__device__ __forceinline__ void pp_load_a_mix(uint4 (&a)[2], const unsigned char *planes, int lane, int col, int nt0) {
    const int g = lane >> 4;
    const unsigned char *p = planes + (g == 1 ? kSpPlaneB : 0) + ((lane & 15) + 16 * nt0) * kSpStrideB + sp_phys(col);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) a[nt] = *reinterpret_cast<const uint4 *>(p + nt * 16 * kSpStrideB);
}
template <int NT>
__device__ __forceinline__ void pp_term(const uint4 (&w)[4], const uint4 (&a)[2], f32x4 (&acc)[4][2]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = mfma_bf16<true>(w[mt], a[nt], acc[mt][nt]);
}
template <int NT>
__device__ __forceinline__ void pp_first(const uint4 (&w)[4], const uint4 (&a)[2], f32x4 (&acc)[4][2], const f32x4 (&b4)[4]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = mfma_bf16<true>(w[mt], a[nt], b4[mt]);
}

struct PpNoFill { static constexpr int units = 0, per_mfma = 0; __device__ __forceinline__ void unit(int) const {} };

// units of the vector work that ride along with the group of matrix instructions in front of this call (compile-time after unrolling); MF = how
// many matrix instructions that group has.  The scheduling region that ends here is laid out as MF x (one matrix instruction, VPM vector
// instructions): left to itself the scheduler clusters the matrix instructions and leaves the vector work in one run behind them.
template <int MF, class Fill>
__device__ __forceinline__ void pp_fill(const Fill &fill, int part, int parts) {
#ifdef PP_NOFILL
    return;
#endif
    constexpr int U = Fill::units;
    bool any = false;
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (u * parts / (U > 0 ? U : 1) == part) { fill.unit(u); any = true; }     // (unit u rides in part floor(u * parts / U): spread evenly, in order)
    if (any) {
#pragma unroll
        for (int m = 0; m < MF; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // one matrix instruction
            __builtin_amdgcn_sched_group_barrier(0x002, Fill::per_mfma, 0);     // vector instructions behind it
        }
    }
    __builtin_amdgcn_sched_barrier(0);                              // (the scheduler mixes a unit with the matrix instructions in front of it, not beyond)
}

// One bracket of a [h | input slot] layer (the LSTM step, layer1): acc = bias + W x [h | slot] for row tiles nt0 .. nt0 + NT - 1, the vector work
// of `fill` spread over its eight groups of matrix instructions.  FULL = false: the first LSTM step (h == 0: only the slot's mixed product).
// wb[0], wb[1], wslot: the layer's weight fragments (chunk 0, chunk 1, the slot chunk's mixed plane), loaded by the caller; untouched unless
// RELOAD: then every buffer is re-requested right after its last use here -- chunk 0 / 1 and the mixed plane of `next_layer` -- for the bracket
// that follows the barrier.  b4 <- `next_bias` (the next bracket's bias: this layer's again, or the next layer's) at the end, always.
template <int NT, bool FULL, bool RELOAD, class Fill>
__device__ __forceinline__ void pp_bracket(const unsigned char *planes, const SplitSrc &src, int slot_col, int wave, int lane, int nt0, PpW (&wb)[2],
                                           uint4 (&wslot)[4], f32x4 (&acc)[4][2], f32x4 (&b4)[4], int next_layer, int next_bias, const Fill &fill) {
    uint4 a_lo[2], a_hi[2];
    if (!FULL) {
        pp_load_a_mix<NT>(a_lo, planes, lane, slot_col, nt0);
        pp_fill<4 * NT>(fill, 0, 2);
        pp_first<NT>(wslot, a_lo, acc, b4);
        if (RELOAD) split_load_w1(wslot, src, next_layer, 2, wave, lane, kSpSlotChunk);
        pp_fill<4 * NT>(fill, 1, 2);
        split_load_bias(b4, src, next_bias, wave, lane);   // (the bias is the C operand of a bracket's FIRST product only: requested at the end of the
        return;                                            //  bracket before, it occupies its 16 registers across the barrier, not across a bracket)
    }
    pp_load_a<NT>(a_lo, planes, 1, lane, 0, nt0);
    pp_load_a<NT>(a_hi, planes, 0, lane, 0, nt0);
    pp_fill<4 * NT>(fill, 0, 8);                                    // (covers the fragments' trip from LDS behind the barrier)
    // ---- chunk 0 ----
    pp_first<NT>(wb[0].p0, a_lo, acc, b4);                  // w1 a2 (+ bias)
    pp_load_a<NT>(a_lo, planes, 1, lane, 32, nt0);
    pp_fill<4 * NT>(fill, 1, 8);
    pp_term<NT>(wb[0].p1, a_hi, acc);                       // w2 a1
    if (RELOAD) split_load_w1(wb[0].p1, src, next_layer, 1, wave, lane, 0);
    pp_fill<4 * NT>(fill, 2, 8);
    pp_term<NT>(wb[0].p0, a_hi, acc);                       // w1 a1
    if (RELOAD) split_load_w1(wb[0].p0, src, next_layer, 0, wave, lane, 0);
    pp_load_a<NT>(a_hi, planes, 0, lane, 32, nt0);
    pp_fill<4 * NT>(fill, 3, 8);
    // ---- chunk 1 ----
    pp_term<NT>(wb[1].p0, a_lo, acc);
    pp_load_a_mix<NT>(a_lo, planes, lane, slot_col, nt0);
    pp_fill<4 * NT>(fill, 4, 8);
    pp_term<NT>(wb[1].p1, a_hi, acc);
    if (RELOAD) split_load_w1(wb[1].p1, src, next_layer, 1, wave, lane, 1);
    pp_fill<4 * NT>(fill, 5, 8);
    pp_term<NT>(wb[1].p0, a_hi, acc);
    if (RELOAD) split_load_w1(wb[1].p0, src, next_layer, 0, wave, lane, 1);
    pp_fill<4 * NT>(fill, 6, 8);
    // ---- the input slot: its three partial products side by side along K ----
    pp_term<NT>(wslot, a_lo, acc);
    if (RELOAD) split_load_w1(wslot, src, next_layer, 2, wave, lane, kSpSlotChunk);
    pp_fill<4 * NT>(fill, 7, 8);
    split_load_bias(b4, src, next_bias, wave, lane);
}

// The LSTM cell update of row tiles nte0 .. nte0 + NTE - 1 from accumulators `acc` (the gates of step `t`), two cells (elements 2 rp, 2 rp + 1)
// per unit: policy_split_tile's statements, the new h stored four bytes per plane at a time.
template <int NTE, int NTE0>
struct PpCellFill {
    static constexpr int units = 2 * NTE, per_mfma = 8;
    static constexpr int nte0 = NTE0;
    unsigned char *planes;
    const f32x4 (&acc)[4][2];
    f32x4 (&cell)[4];
    const float *len_f;                                    // [64] the tile rows' sequence lengths (LDS)
    int t, wave, lane;
    __device__ __forceinline__ void unit(int u) const {
        const int nt = u >> 1, r = 2 * (u & 1), ntg = nte0 + nt;
        const bool live = len_f[16 * ntg + (lane & 15)] > (float)t;           // dynamic_rnn: rows past their own length keep (c, h)
        f32x2 c2, h2;
        split_lstm_cell2(f32x2{acc[0][nt][r], acc[0][nt][r + 1]}, f32x2{acc[1][nt][r], acc[1][nt][r + 1]}, f32x2{acc[2][nt][r], acc[2][nt][r + 1]},
                         f32x2{acc[3][nt][r], acc[3][nt][r + 1]}, f32x2{cell[ntg][r], cell[ntg][r + 1]}, c2, h2);
        cell[ntg][r] = live ? c2[0] : cell[ntg][r];
        cell[ntg][r + 1] = live ? c2[1] : cell[ntg][r + 1];
        // (no branch around the store -- it would cut the bracket into basic blocks the matrix instructions cannot be scheduled across: a row that is
        //  past its length writes its four bytes into its own corner of the row's zero column instead, bytes 512 + 4 (lane / 16) .., which the
        //  float16 form never reads: the slot chunk's idle k-groups take the mixed fragments' finite values, not zeros)
        uint32_t hi, lo;
        split2<true>(h2[0], h2[1], hi, lo);
        const int off = live ? sp_phys(16 * wave + 4 * (lane >> 4) + r) : 2 * kSpZeroCol + 4 * (lane >> 4);
        unsigned char *d = planes + (16 * ntg + (lane & 15)) * kSpStrideB + off;
        *reinterpret_cast<uint32_t *>(d) = hi;
        *reinterpret_cast<uint32_t *>(d + kSpPlaneB) = lo;
    }
};

// relu + split + store of row tiles nte0 .. nte0 + NTE - 1 from accumulators `acc`, one (column tile, row tile) per unit
template <int NTE, int NTE0>
struct PpReluFill {
    static constexpr int units = 4 * NTE, per_mfma = 3;
    static constexpr int nte0 = NTE0;
    unsigned char *planes;
    const f32x4 (&acc)[4][2];
    int wave, lane;
    __device__ __forceinline__ void unit(int u) const {
        const int nt = u >> 2, mt = u & 3;
        f32x4 z;
#pragma unroll
        for (int r = 0; r < 4; ++r) z[r] = __builtin_amdgcn_fmed3f(acc[mt][nt][r], 0.0f, kSpF16Max);
        split_store4<true>(planes, 16 * (nte0 + nt) + (lane & 15), 16 * (4 * wave + mt) + 4 * (lane >> 4), z);
    }
};

}  // namespace cavoid
