// cavoid_policy_split8.hpp -- policy_split_tile8: the pass of cavoid_policy_split.hpp (NetworkVP_rnn inference, float32 in and out on
// v_mfma_f32_16x16x32_f16 by the two-piece operand split, three partial products) by EIGHT wavefronts per 64-row tile instead of four.
//
// Why.  With four wavefronts per tile and two tiles per CU a SIMD holds two wavefronts; PMC and the phase trace of round 5 say the pass is the SUM
// of its matrix time and its vector time plus ~a third of idle issue slots at the ~14 phase boundaries per tile (barrier -> first fragment
// read -> first matrix instruction), which the ONE partner wavefront meets with nothing to issue half of the time.
// tools/ubench/mfma32_valu_overlap.hip: a matrix stream and a vector stream of DIFFERENT wavefronts of one SIMD do run side by side (16.4 clocks
// per v_mfma_f32_16x16x32_f16 with the partner issuing a vector instruction every 8 clocks).  Eight wavefronts per tile = four per SIMD: twice
// the candidates at every boundary, each with half the accumulators (the 128 registers a wavefront gets at that occupancy).
//
// What changes against the four-wavefront form -- and what does not:
//   * wavefront w (0..7) owns output columns 32w .. 32w+31 of every layer: TWO column tiles x four row tiles = 32 accumulator registers,
//     24 weight-fragment registers (three buffers), 32 activation-fragment registers; every wavefront still reads all of the tile's
//     activation fragments (twice the LDS fragment traffic per tile: the reads are conflict-free since round 5, sp_phys);
//   * the LSTM: a wavefront's 32 gate columns are the i, j, f, o gates of hidden units 8w .. 8w+7, ordered so that ONE lane holds all four
//     gates of units 8w + 2g, 8w + 2g + 1 (g = lane / 16): column tile 0 = (i_a i_b j_a j_b) per lane group, column tile 1 = (f_a f_b o_a o_b)
//     -- the cell update stays per lane on the packed float32 instructions, the cell state (two values per row tile) never leaves registers,
//     h goes out as one 4-byte store per plane.  The LSTM's weight fragments and biases exist a second time in that column order
//     (policy_pack_split8_kernel, behind the four-wavefront pack: kSpOffLstm8, kBiasLstm8); the other layers' fragments are shared;
//   * the heads stay on wavefronts 0..3 (one 16-row tile each, 24 matrix instructions);
//   * every output element is the SAME float32 sum in the SAME order as in policy_split_tile (bias + per K-chunk w1 a2 + w2 a1 + w1 a1, the
//     input slot's mixed product last): which wavefront owns a column changes nothing in it, so actions, values and probabilities are bit for
//     bit those of the four-wavefront kernel (tests/test_gpu_policy.py::test_eight_wavefront_form_is_bit_identical).
// float16 pieces only (kSpF16, the default form); the bf16 forms stay with the four-wavefront kernel.
#pragma once
#include "cavoid_policy_split.hpp"

namespace cavoid {

// packed LSTM column of this form -> the four-wavefront form's packed column of the same (gate, hidden unit)
__host__ __device__ constexpr int sp8_lstm_col4(int col8) {
    const int wave8 = col8 >> 5, mt = (col8 >> 4) & 1, q = col8 & 15, g = q >> 2, r = q & 3;
    const int gate = 2 * mt + (r >> 1), unit = 8 * wave8 + 2 * g + (r & 1);
    return 64 * (unit >> 4) + 16 * gate + (unit & 15);
}
static_assert(sp8_lstm_col4(0) == 0 && sp8_lstm_col4(1) == 1 && sp8_lstm_col4(2) == 16 && sp8_lstm_col4(16) == 32 && sp8_lstm_col4(18) == 48 &&
              sp8_lstm_col4(4) == 2 && sp8_lstm_col4(32) == 8 && sp8_lstm_col4(64) == 64, "sp8_lstm_col4");

#ifdef CAVOID_POLICY_KERNELS     /* the non-template kernels are compiled by cavoid_policy_capi.hip only */
// the LSTM's float16 pieces (planes 0, 1 and the slot chunk's mixed plane 2: policy_pack_split_kernel's statements) and biases in this
// form's column order; weights beyond +-65504 were counted by policy_pack_split_kernel (the same values)
__global__ void __launch_bounds__(256) policy_pack_split8_kernel(const PolicyWeights w, uint4 *frags, float *sbias) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one (chunk, column tile, lane): all 3 planes
    if (f < 256) {
        const int c4 = sp8_lstm_col4((int)f);
        const int wave = c4 >> 6, gate = (c4 >> 4) & 3, u = c4 & 15;
        sbias[kBiasLstm8 + f] = split_gate_scale(c4) * (w.lstm_bias[gate * kPolHidden + 16 * wave + u] + (gate == 2 ? w.forget_bias : 0.0f));
    }
    if (f >= (int64_t)kSpChLstm * 16 * 64) return;
    const int c = (int)(f >> 10), mt = (int)((f >> 6) & 15), lane = (int)(f & 63);
    const int64_t base = kSpOffLstm8 + (int64_t)c * kSpFragPerChunk + (int64_t)mt * 64 + lane;
    const int g = lane >> 4, col = sp8_lstm_col4(16 * mt + (lane & 15));
    uint32_t pl[3][4];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        uint32_t a1, a2, a3, b1, b2, b3;
        split3_f16(split_weight(w, 0, c, g, e, col), a1, a2, a3);
        split3_f16(split_weight(w, 0, c, g, e + 1, col), b1, b2, b3);
        if (c == kSpSlotChunk) {
            uint32_t m1, m2, m3, n1, n2, n3;
            split3_f16(split_weight(w, 0, c, 0, e, col), m1, m2, m3);
            split3_f16(split_weight(w, 0, c, 0, e + 1, col), n1, n2, n3);
            a3 = g < 2 ? m1 : (g == 2 ? m2 : 0u);
            b3 = g < 2 ? n1 : (g == 2 ? n2 : 0u);
        }
        pl[0][e >> 1] = a1 | (b1 << 16); pl[1][e >> 1] = a2 | (b2 << 16); pl[2][e >> 1] = a3 | (b3 << 16);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) frags[base + p * 16 * 64] = uint4{pl[p][0], pl[p][1], pl[p][2], pl[p][3]};
}
#endif

struct Split8W { uint4 w[3][2]; };                           // weight fragments: buffer x column tile

__device__ __forceinline__ void split8_load_w1(uint4 (&w)[2], const SplitSrc &src, int layer, int plane, int wave, int lane, int c) {
    const int soff = (layer + c * (int)kSpFragPerChunk + plane * 16 * 64 + (2 * wave) * 64) * 16;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) w[mt] = split_buf16(src.w, lane * 16, soff + mt * 1024);
}
// this lane's four bias values per column tile (columns 16*(2*wave+mt) + 4*(lane/16) + r)
__device__ __forceinline__ void split8_load_bias(f32x4 (&b4)[2], const SplitSrc &src, int bias, int wave, int lane) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const uint4 v = split_buf16(src.b, (lane >> 4) * 16, (bias + 32 * wave + 16 * mt) * 4);
        b4[mt] = f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    }
}
template <class RT>
__device__ __forceinline__ void split8_term(const uint4 (&w)[2], const uint4 (&a)[4], f32x4 (&acc)[2][4], RT rt) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
        if (rt.has(nt)) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = mfma_bf16<true>(w[mt], a[nt], acc[mt][nt]);
        }
}
template <class RT>
__device__ __forceinline__ void split8_first(const uint4 (&w)[2], const uint4 (&a)[4], f32x4 (&acc)[2][4], const f32x4 (&b4)[2], RT rt) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
        if (rt.has(nt)) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = mfma_bf16<true>(w[mt], a[nt], b4[mt]);
        }
}

// split_gemm3<true> for two column tiles per wavefront: the same chunk / product order (per chunk w1 a2, w2 a1, w1 a1; the slot chunk's mixed
// product last; the first product takes the bias as its C operand), the same one-chunk-ahead requests through three weight buffers
template <class RT>
__device__ __forceinline__ void split8_gemm(const unsigned char *planes, const SplitSrc &src, int layer, int c0, int c1, int slot_chunk, int slot_col,
                                            int wave, int lane, Split8W &w, f32x4 (&acc)[2][4], int next_layer, int next_c,
                                            f32x4 (&b4)[2], int next_bias, RT rt) {
    const bool slot_last = slot_chunk >= 0;
    const int cm = slot_last ? c1 - 1 : c1;                // plain chunks: c0 .. cm - 1
    uint4 a_hi[4], a_lo[4];
    auto col_of = [&](int c) { return c == slot_chunk ? slot_col : 32 * c; };
    if (c0 >= cm) {                                        // (uniform) only the slot chunk: acc = bias + mixed product
        split_load_a_mix(a_lo, planes, lane, slot_col, rt);
        split8_first(w.w[0], a_lo, acc, b4, rt);
        __builtin_amdgcn_sched_barrier(0);
        split8_load_bias(b4, src, next_bias, wave, lane);
        split8_load_w1(w.w[1], src, next_layer, 1, wave, lane, next_c);
        split8_load_w1(w.w[0], src, next_layer, 0, wave, lane, next_c);
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
            if (!last) split8_load_w1(w.w[2], src, layer, 0, wave, lane, n);
            else if (to_slot) split8_load_w1(w.w[2], src, layer, 2, wave, lane, cm);
            if (first) {                                   // (uniform) acc = bias + w1 * a_lo
                split8_first(w.w[0], a_lo, acc, b4, rt);
                first = false;
            } else {
                split8_term(w.w[0], a_lo, acc, rt);
            }
            if (last) split8_load_bias(b4, src, next_bias, wave, lane);
            __builtin_amdgcn_sched_barrier(0);
            if (to_slot) split_load_a_mix(a_lo, planes, lane, slot_col, rt);
            else split_load_a(a_lo, planes, 1, lane, col_of(n), n == slot_chunk, rt);
            split8_term(w.w[1], a_hi, acc, rt);
            __builtin_amdgcn_sched_barrier(0);
            split8_load_w1(w.w[1], src, wl, 1, wave, lane, wc);
            split8_term(w.w[0], a_hi, acc, rt);
            __builtin_amdgcn_sched_barrier(0);
            if (!to_slot) split_load_a(a_hi, planes, 0, lane, col_of(n), n == slot_chunk, rt);
            if (last) {
                split8_load_w1(w.w[0], src, next_layer, 0, wave, lane, next_c);
                if (to_slot) split8_term(w.w[2], a_lo, acc, rt);
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
            if (to_slot) split8_load_w1(w.w[0], src, layer, 2, wave, lane, cm);
            else split8_load_w1(w.w[0], src, wl, 0, wave, lane, wc);
            split8_term(w.w[2], a_lo, acc, rt);
            if (last) split8_load_bias(b4, src, next_bias, wave, lane);
            __builtin_amdgcn_sched_barrier(0);
            if (to_slot) split_load_a_mix(a_lo, planes, lane, slot_col, rt);
            else split_load_a(a_lo, planes, 1, lane, col_of(n), n == slot_chunk, rt);
            split8_term(w.w[1], a_hi, acc, rt);
            __builtin_amdgcn_sched_barrier(0);
            split8_load_w1(w.w[1], src, wl, 1, wave, lane, wc);
            split8_term(w.w[2], a_hi, acc, rt);
            __builtin_amdgcn_sched_barrier(0);
            if (!to_slot) split_load_a(a_hi, planes, 0, lane, col_of(n), n == slot_chunk, rt);
            if (last) {
                if (to_slot) {
                    split8_term(w.w[0], a_lo, acc, rt);
                    __builtin_amdgcn_sched_barrier(0);
                    split8_load_w1(w.w[0], src, next_layer, 0, wave, lane, next_c);
                }
                break;
            }
        }
        ++c;
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <class RT>
__device__ __forceinline__ void split8_store_relu(unsigned char *planes, int wave, int lane, const f32x4 (&acc)[2][4], RT rt) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        if (!rt.has(nt)) continue;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 z;
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = __builtin_amdgcn_fmed3f(acc[mt][nt][r], 0.0f, kSpF16Max);   // relu, saturating at float16's largest value
            split_store4<true>(planes, 16 * nt + (lane & 15), 16 * (2 * wave + mt) + 4 * (lane >> 4), z);
        }
    }
}

// The forward pass of ONE 64-row tile by the 8 wavefronts (512 threads) of a workgroup: policy_split_tile<kSpF16, NRT>'s contract -- load(),
// emit(), live_mask / rmap with NRT > 0 (COMPACT: the first NRT row tiles hold the rows that still need an action) -- and its results.
template <int NRT = 0, class Load, class Emit>
__device__ __forceinline__ void policy_split_tile8(const SplitArgs &sa, unsigned char *planes, float *len_f, int *wave_max, int rows_here,
                                                   int tid, Load load, Emit emit, unsigned long long live_mask = ~0ull, int *rmap = nullptr) {
    const PolicyArgs &p = sa.p;
    constexpr bool COMPACT = NRT > 0;
    using RT = typename std::conditional<COMPACT, SpFirstRows<(NRT > 0 ? NRT : 4)>, SpAllRows>::type;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4;
    const int M = p.max_other, A = p.num_actions;
    const SplitSrc src = split_src(sa.sfrags, sa.sbias);
    constexpr int w_lstm = (int)kSpOffLstm8;
    int n_live = 64, pos = lane;
    bool mine = lane < rows_here;
    const RT rt{};
    if constexpr (COMPACT) {
        mine = (live_mask >> lane) & 1ull;
        n_live = __popcll(live_mask);
        pos = __popcll(live_mask & ((1ull << lane) - 1ull));
    }
    Split8W f0;
    f32x4 b4[2];
    // first LSTM step: h == 0, only the input chunk contributes (its mixed plane, one product)
    split8_load_w1(f0.w[0], src, w_lstm, 2, wave, lane, kSpSlotChunk);
    split8_load_bias(b4, src, kBiasLstm8, wave, lane);

    // ---- input tile: gather + normalise + split into the slot columns (policy_split_tile's statements over 512 threads) ---------------
    {
        if (tid < 64) {
            int local_max = 0, local_min = 0;
            const float v = (COMPACT ? mine : tid < rows_here) ? load(tid, 0) : 0.0f;
            if constexpr (COMPACT) {
                if (mine) { len_f[pos] = v; rmap[pos] = tid; }
                if (tid >= n_live) len_f[tid] = 0.0f;
            } else {
                len_f[tid] = v;
            }
            int len = (int)v;
            len = len < 0 ? 0 : (len > M ? M : len);
            local_max = len;
            local_min = v >= (float)len ? len : len - 1;
            local_min = local_min < 0 ? 0 : local_min;
            if (COMPACT && !mine) local_min = M;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_xor(local_max, d, 64), q = __shfl_xor(local_min, d, 64);
                local_max = o > local_max ? o : local_max;
                local_min = q < local_min ? q : local_min;
            }
            if (lane == 0) { wave_max[0] = local_max; wave_max[5] = local_min; }
        }
        // h = 0 (columns 0..63 of both planes)
        for (int e = tid; e < 2 * 64 * 8; e += 512) {
            const int pl = e >> 9, r = (e >> 3) & 63, c16 = e & 7;
            *reinterpret_cast<uint4 *>(planes + pl * kSpPlaneB + r * kSpStrideB + sp_phys(8 * c16)) = uint4{0u, 0u, 0u, 0u};
        }
        if (tid < 128)                                      // the zero column (256..263) of every row of both planes
            *reinterpret_cast<uint4 *>(planes + (tid >> 6) * kSpPlaneB + (tid & 63) * kSpStrideB + kSpZeroCol * 2) = uint4{0u, 0u, 0u, 0u};
        const int items = 64 * (M + 1);                    // (row, slot): slot 0 = host (4 values), slot s = observed agent s-1 (7)
        for (int it = tid; it < items; it += 512) {
            const int r = it & 63, s = it >> 6;
            const int n_in = s == 0 ? kPolHost : kPolOther, sc0 = s == 0 ? 1 : 1 + kPolHost + kPolOther * (s - 1);
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
                for (int e = 0; e < 8; ++e) v[e] = (v[e] - av[e]) * __builtin_amdgcn_rcpf(sd[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (row_ok && e < n_in) ? v[e] : 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], -kSpF16Max, kSpF16Max);
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split2<true>(v[2 * e], v[2 * e + 1], hi[e], lo[e]);
            unsigned char *d = planes + ((COMPACT && mine) ? pos : r) * kSpStrideB + sp_phys(kSpSlotCol + 8 * s);
            *reinterpret_cast<uint4 *>(d) = uint4{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<uint4 *>(d + kSpPlaneB) = uint4{lo[0], lo[1], lo[2], lo[3]};
            if (COMPACT && mine && r >= n_live) {           // a live lane behind the packed rows: tile row `lane` is nobody's -- it takes zeros like the rest of the tail
                unsigned char *z = planes + r * kSpStrideB + sp_phys(kSpSlotCol + 8 * s);
                *reinterpret_cast<uint4 *>(z) = uint4{0u, 0u, 0u, 0u};
                *reinterpret_cast<uint4 *>(z + kSpPlaneB) = uint4{0u, 0u, 0u, 0u};
            }
        }
    }
    __syncthreads();
    const int steps = wave_max[0];
    const int tile_min_len = wave_max[5];
    POLICY_STAMP(5);

    float len_r[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) len_r[nt] = len_f[16 * nt + (lane & 15)];

    // ---- LSTM over the observed agents: lane = row 16nt + l%16, hidden units 8w + 2g, 8w + 2g + 1 ----------------------------------
    f32x2 cell[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) cell[nt] = f32x2{0.f, 0.f};
    auto lstm_step = [&](const int t, auto all_live_c) {
        constexpr bool ALL_LIVE = decltype(all_live_c)::value;
        f32x4 acc[2][4];
        if (t == 1) POLICY_STAMP(8);
        split8_gemm(planes, src, w_lstm, t == 0 ? 2 : 0, kSpChLstm, 2, kSpSlotCol + 8 * (1 + t), wave, lane, f0, acc,
                    t + 1 < steps ? w_lstm : (int)kSpOffL1, 0, b4, t + 1 < steps ? kBiasLstm8 : kBiasL1, rt);
        if (t == 1) POLICY_STAMP(9);
        __syncthreads();                                   // every wavefront has read h
        if (t == 1) POLICY_STAMP(10);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (!rt.has(nt)) continue;
            const bool live = ALL_LIVE || len_r[nt] > (float)t;
            f32x2 c2, h2;
            split_lstm_cell2(f32x2{acc[0][nt][0], acc[0][nt][1]}, f32x2{acc[0][nt][2], acc[0][nt][3]},
                             f32x2{acc[1][nt][0], acc[1][nt][1]}, f32x2{acc[1][nt][2], acc[1][nt][3]}, cell[nt], c2, h2);
            if constexpr (ALL_LIVE) cell[nt] = c2;
            else { cell[nt][0] = live ? c2[0] : cell[nt][0]; cell[nt][1] = live ? c2[1] : cell[nt][1]; }
            if (live) {                                    // (else h stays as it is)
                uint32_t hi, lo;
                split2<true>(h2[0], h2[1], hi, lo);
                unsigned char *d = planes + (16 * nt + (lane & 15)) * kSpStrideB + sp_phys(8 * wave + 2 * g);
                *reinterpret_cast<uint32_t *>(d) = hi;
                *reinterpret_cast<uint32_t *>(d + kSpPlaneB) = lo;
            }
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
    }
    POLICY_STAMP(1);
    // ---- layer1 on [h | host] -------------------------------------------------------------------------------------
    {
        f32x4 acc[2][4];
        if (steps == 0) {                                  // (else the last LSTM step asked for them)
            split8_load_w1(f0.w[0], src, (int)kSpOffL1, 0, wave, lane, 0);
            split8_load_w1(f0.w[1], src, (int)kSpOffL1, 1, wave, lane, 0);
            split8_load_bias(b4, src, kBiasL1, wave, lane);
        }
        split8_gemm(planes, src, (int)kSpOffL1, 0, kSpChL1, 2, kSpSlotCol, wave, lane, f0, acc, (int)kSpOffL2, 0, b4, kBiasL2, rt);
        __syncthreads();
        split8_store_relu(planes, wave, lane, acc, rt);
        __syncthreads();
    }
    POLICY_STAMP(2);
    // ---- layer2, fullyconnected1 ----------------------------------------------------------------------------------
    {
        f32x4 acc[2][4];
        split8_gemm(planes, src, (int)kSpOffL2, 0, kSpChWide, -1, 0, wave, lane, f0, acc, (int)kSpOffFc1, 0, b4, kBiasFc1, rt);
        __syncthreads();
        split8_store_relu(planes, wave, lane, acc, rt);
        __syncthreads();
    }
    const bool heads_here = wave < 4 && rt.has(wave);      // (uniform) the heads: wavefront w < 4 does rows 16w .. 16w+15
    uint4 hw[kSpChWide][2];                                // the heads' weight fragments: half in flight across the epilogue
    auto head_frag = [&](int c, int pl) { return split_buf16(src.w, lane * 16, ((int)kSpOffHead + (c * 3 + pl) * 64) * 16); };
    {
        f32x4 acc[2][4];
        split8_gemm(planes, src, (int)kSpOffFc1, 0, kSpChWide, -1, 0, wave, lane, f0, acc, (int)kSpOffFc1, kSpChWide - 1, b4, kBiasFc1, rt);
        if (heads_here) {
#pragma unroll
            for (int c = 0; c < kSpChWide / 2; ++c)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) hw[c][pl] = head_frag(c, pl);
        }
        __syncthreads();
        split8_store_relu(planes, wave, lane, acc, rt);
        __syncthreads();
    }
    POLICY_STAMP(3);
    if (heads_here) {
#pragma unroll
        for (int c = kSpChWide / 2; c < kSpChWide; ++c)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) hw[c][pl] = head_frag(c, pl);
        f32x4 acc[3];
        acc[0] = *reinterpret_cast<const f32x4 *>(p.bias + kBiasHead + 4 * g);
        acc[1] = acc[2] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char *arow = planes + (16 * wave + (lane & 15)) * kSpStrideB + sp_phys(8 * g);
#pragma unroll
        for (int c = 0; c < kSpChWide; ++c) {
            const uint4 a1 = *reinterpret_cast<const uint4 *>(arow + c * 32), a2 = *reinterpret_cast<const uint4 *>(arow + kSpPlaneB + c * 32);
            acc[2] = mfma_bf16<true>(hw[c][1], a1, acc[2]);
            acc[1] = mfma_bf16<true>(hw[c][0], a2, acc[1]);
            acc[0] = mfma_bf16<true>(hw[c][0], a1, acc[0]);
        }
        // (policy_split_tile's expression with its two absent products spelled out: (0 + 0) + (acc2 + acc1) + acc0)
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 logit = (zero + zero) + (acc[2] + acc[1]) + acc[0];
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
        if constexpr (COMPACT) {
            if (trow < n_live) emit(rmap[trow], g, pj, logit);
        } else {
            emit(trow, g, pj, logit);
        }
    }
    POLICY_STAMP(4);
}

// policy_forward_split_kernel<kSpF16> with eight wavefronts per tile (CAVOID_POLICY_WAVES=8 at cavoid_policy_create)
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) policy_forward_split8_kernel(const SplitArgs sa) {
    const PolicyArgs &p = sa.p;
    extern __shared__ __attribute__((aligned(16))) unsigned char planes[];
    float *len_f = reinterpret_cast<float *>(planes + 2 * kSpPlaneB);
    int *tile_row = reinterpret_cast<int *>(len_f + 64);
    int *wave_max = tile_row + 64;
    int &ticket = wave_max[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    const int64_t n_rows = p.row_count ? (int64_t)*p.row_count : p.rows;
    const int rows_here = n_rows - row0 < 64 ? (int)(n_rows - row0 > 0 ? n_rows - row0 : 0) : 64;
    const int A = p.num_actions;
    const int step = p.actions_out ? *p.step_counter : 0;
    const bool listed = p.row_index != nullptr;
    if (listed && rows_here == 0) {
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
    if (tid < 64) tile_row[tid] = listed ? (tid < rows_here ? p.row_index[row0 + tid] : 0) : tid;
    if (tid == 0) {
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
        if (p.actions_out) {
            const int action = split_select_action(pj, g, lane, A, p.greedy != 0, row, step, p.seed_lo, p.seed_hi);
            if (row < p.rows && g == 0) p.actions_out[row] = action;
        }
    };
    policy_split_tile8<0>(sa, planes, len_f, wave_max, rows_here, tid, load, emit);
#ifdef CAVOID_TRACE
    if (tid == 0 && g_pol_trace) g_pol_trace[(size_t)blockIdx.x * 16 + 6] = clock64() - trace_c0;
#endif
    if (p.actions_out) policy_finish(p, step, tid);
}

}  // namespace cavoid
