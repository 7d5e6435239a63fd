// cavoid_policy.hpp -- NetworkVP_rnn on the gfx950 matrix cores: the actors' predict_p_and_v + select_action as ONE
// kernel (policy_forward_kernel<RT, false>), and the trainer's pass over a batch as two (policy_forward_kernel<RT, true>:
// forward + A3C loss + gradient at the heads; policy_backward_kernel: everything row-local of the backward pass).
//
// What it computes (citations: /root/reference/ga3c/GA3C):
//   x [rows, 5+7M] -> (x - avg) / std                                        NetworkVP_rnn.py:50-53
//   num_other = x[:,0] raw; host = xn[:,1:5]; others = xn[:,5:] as [M,7]      :58-61
//   LSTMCell(64) over the others, state frozen past each row's own length      :64-66
//   [host | h] -> dense256 relu (layer1) -> dense256 relu (layer2)             :67,103-105
//   -> dense256 relu (fullyconnected1) -> logits_v, logits_p                   NetworkVPCore.py:66-75
//   softmax_p = (softmax(logits_p) + MIN_POLICY) / (1 + MIN_POLICY * A)         :75
//
// This part of the path IS a dense contraction (280 kFLOP per row, 9.2 GFLOP per step at 4 x 8192), so
// it runs on the matrix cores: v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate -- the reference policy
// is float32 TensorFlow, no precision is given up).  One workgroup = 64 rows x 4 wavefronts:
//   * activations live in ONE LDS buffer [64][260] floats (stride 260: ds_read_b128 of the A
//     fragments and ds_write_b32 of the C fragments are both bank-conflict free);
//   * wavefront w owns output columns 64w..64w+63 of every layer (4 row tiles x 4 column tiles of
//     16x16, 64 accumulator registers); for the LSTM the columns are permuted so that its 4 column
//     tiles are the i, j, f, o gates of hidden units 16w..16w+15 -- the cell update is then per lane,
//     the cell state never leaves registers;
//   * weights are read from a fragment-ordered copy (policy_pack_kernel): one global_load_dwordx4 per
//     lane feeds 4 MFMAs, 1 KB contiguous per wave instruction, L2-resident (688 KB in all);
//   * a K chunk is 16 wide: lane group g = lane/16 supplies k = 16*chunk + 4g + s in MFMA s = 0..3
//     (any fixed bijection of k works as long as A and B agree).
// The backward pass is the same machinery on transposed fragment packs (dX = dY x W^T); the weight gradients X^T G
// are large plain GEMMs over all rows and are left to the GEMM library (see policy_backward_kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cavoid {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPolStride = 260;        // LDS row stride in floats (260 % 64 == 4)
constexpr int kPolHidden = 64, kPolWidth = 256, kPolHost = 4, kPolOther = 7;
constexpr int kPolMaxOthers = 19;      // the padded input row parked at LDS columns 80.. must fit: 80 + 16 + 8M + 8 <= 260
constexpr int kPolCuSlots = 4096;      // (XCC id, SE, SH, CU) keys
constexpr int kPolXCol = 80;           // first LDS column of the normalised input row

// chunk counts (16 k each) and fragment offsets (in float4 = one lane's 4 k-values) of the packed weights
constexpr int kChLstm = 5, kChL1 = 5, kChWide = 16, kChHead = 16;
constexpr int64_t kFragPerChunk = 16 * 64;                     // 16 column tiles x 64 lanes
constexpr int64_t kOffLstm = 0;
constexpr int64_t kOffL1 = kOffLstm + kChLstm * kFragPerChunk;
constexpr int64_t kOffL2 = kOffL1 + kChL1 * kFragPerChunk;
constexpr int64_t kOffFc1 = kOffL2 + kChWide * kFragPerChunk;
constexpr int64_t kOffHead = kOffFc1 + kChWide * kFragPerChunk;  // one column tile only: 64 frags per chunk
constexpr int64_t kPackFrags = kOffHead + kChHead * 64;
// transposed copies for the trainer's backward pass (dX = dY x W^T: K = the layer's output width, N = its input width);
// N = 64 layers (inputs that are the LSTM state) have 4 column tiles per chunk instead of 16
constexpr int64_t kFragPerChunkNarrow = 4 * 64;
constexpr int64_t kOffTHead = kPackFrags;                                   // K = 16 (A logits, value, pad), N = 256
constexpr int64_t kOffTFc1 = kOffTHead + 1 * kFragPerChunk;                 // K = 256, N = 256
constexpr int64_t kOffTL2 = kOffTFc1 + kChWide * kFragPerChunk;
constexpr int64_t kOffTL1 = kOffTL2 + kChWide * kFragPerChunk;              // K = 256, N = 64 (the hidden-state inputs of layer1)
constexpr int64_t kOffTLstm = kOffTL1 + kChWide * kFragPerChunkNarrow;      // K = 256 gate columns (packed order), N = 64
constexpr int64_t kPackFragsTrain = kOffTLstm + kChWide * kFragPerChunkNarrow;
// biases, in packed column order: lstm 256 (forget bias folded in), l1 256, l2 256, fc1 256, head 16
constexpr int kBiasLstm = 0, kBiasL1 = 256, kBiasL2 = 512, kBiasFc1 = 768, kBiasHead = 1024, kBiasFloats = 1040;
constexpr size_t policy_lds_bytes(int row_tiles) {       // 64 rows: 71 008 B (2 workgroups per CU)
    return (size_t)(16 * row_tiles * kPolStride + kBiasFloats + 8 + 16 * row_tiles) * sizeof(float);
}

struct PolicyWeights {                 // device pointers, TensorFlow layout ([in, out] kernels)
    const float *lstm_kernel, *lstm_bias;      // [7+64, 256] rows: 7 inputs then 64 hidden; gate order i, j, f, o
    const float *layer1_kernel, *layer1_bias;  // [4+64, 256] rows: 4 host then 64 hidden
    const float *layer2_kernel, *layer2_bias;  // [256, 256]
    const float *fc1_kernel, *fc1_bias;        // [256, 256]
    const float *p_kernel, *p_bias;            // [256, A]
    const float *v_kernel, *v_bias;            // [256, 1]
    int num_actions;
    float forget_bias;
};

// ---- weight packing -----------------------------------------------------------------------------------
// packed frag (layer, chunk, column tile ct, lane) = 4 floats, s = 0..3:
//     W_layer[ kmap(16*chunk + 4*(lane/16) + s) ][ cmap(16*ct + lane%16) ]
__device__ __forceinline__ float policy_weight(const PolicyWeights &w, int layer, int k, int col) {
    switch (layer) {
    case 0: {                                               // LSTM: k 0..63 hidden, 64..70 input; col tiles = (wave, gate)
        const int wave = col >> 6, gate = (col >> 4) & 3, u = col & 15;
        const int src_col = gate * kPolHidden + 16 * wave + u;
        if (k < kPolHidden) return w.lstm_kernel[(int64_t)(kPolOther + k) * 256 + src_col];
        if (k < kPolHidden + kPolOther) return w.lstm_kernel[(int64_t)(k - kPolHidden) * 256 + src_col];
        return 0.0f;
    }
    case 1:                                                 // layer1: k 0..63 hidden, 64..67 host
        if (k < kPolHidden) return w.layer1_kernel[(int64_t)(kPolHost + k) * 256 + col];
        if (k < kPolHidden + kPolHost) return w.layer1_kernel[(int64_t)(k - kPolHidden) * 256 + col];
        return 0.0f;
    case 2: return w.layer2_kernel[(int64_t)k * 256 + col];
    case 3: return w.fc1_kernel[(int64_t)k * 256 + col];
    case 4:                                                 // heads: columns 0..A-1 logits_p, column A logits_v
        if (col < w.num_actions) return w.p_kernel[(int64_t)k * w.num_actions + col];
        if (col == w.num_actions) return w.v_kernel[k];
        return 0.0f;
    // ---- transposed (backward) layers: element [k][col] = W[col][k] --------------------------------------------
    case 5:                                                 // heads^T: k = logit / value index, col = fullyconnected1 unit
        if (k < w.num_actions) return w.p_kernel[(int64_t)col * w.num_actions + k];
        if (k == w.num_actions) return w.v_kernel[col];
        return 0.0f;
    case 6: return w.fc1_kernel[(int64_t)col * 256 + k];
    case 7: return w.layer2_kernel[(int64_t)col * 256 + k];
    case 8: return w.layer1_kernel[(int64_t)(kPolHost + col) * 256 + k];           // col < 64: the hidden-state inputs
    default: {                                              // LSTM^T: k = gate column in packed order, col = hidden unit
        const int wave = k >> 6, gate = (k >> 4) & 3, u = k & 15;
        return w.lstm_kernel[(int64_t)(kPolOther + col) * 256 + gate * kPolHidden + 16 * wave + u];
    }
    }
}

#ifdef CAVOID_POLICY_KERNELS     /* non-template kernel: compiled by cavoid_policy_capi.hip only */
__global__ void __launch_bounds__(256) policy_pack_kernel(const PolicyWeights w, f32x4 *frags, float *bias, int with_backward) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f < (with_backward ? kPackFragsTrain : kPackFrags)) {
        int layer, chunk, ct, lane;
        if (f >= kOffTL1) {                                 // narrow transposed layers: 4 column tiles per chunk
            layer = f >= kOffTLstm ? 9 : 8;
            const int64_t r = f - (f >= kOffTLstm ? kOffTLstm : kOffTL1);
            chunk = (int)(r / kFragPerChunkNarrow); ct = (int)((r >> 6) & 3); lane = (int)(r & 63);
        } else if (f >= kOffTHead) {
            layer = f >= kOffTL2 ? 7 : f >= kOffTFc1 ? 6 : 5;
            const int64_t r = f - (f >= kOffTL2 ? kOffTL2 : f >= kOffTFc1 ? kOffTFc1 : kOffTHead);
            chunk = (int)(r / kFragPerChunk); ct = (int)((r >> 6) & 15); lane = (int)(r & 63);
        } else if (f >= kOffHead) {
            const int64_t r = f - kOffHead;
            layer = 4; chunk = (int)(r >> 6); ct = 0; lane = (int)(r & 63);
        } else {
            const int64_t base = f >= kOffFc1 ? kOffFc1 : f >= kOffL2 ? kOffL2 : f >= kOffL1 ? kOffL1 : kOffLstm;
            layer = f >= kOffFc1 ? 3 : f >= kOffL2 ? 2 : f >= kOffL1 ? 1 : 0;
            const int64_t r = f - base;
            chunk = (int)(r / kFragPerChunk); ct = (int)((r >> 6) & 15); lane = (int)(r & 63);
        }
        const int k0 = 16 * chunk + 4 * (lane >> 4), col = 16 * ct + (lane & 15);
        f32x4 v;
        v.x = policy_weight(w, layer, k0 + 0, col); v.y = policy_weight(w, layer, k0 + 1, col);
        v.z = policy_weight(w, layer, k0 + 2, col); v.w = policy_weight(w, layer, k0 + 3, col);
        frags[f] = v;
    }
    if (f < kBiasFloats) {
        const int i = (int)f;
        float b;
        if (i < kBiasL1) {
            const int wave = i >> 6, gate = (i >> 4) & 3, u = i & 15;
            b = w.lstm_bias[gate * kPolHidden + 16 * wave + u] + (gate == 2 ? w.forget_bias : 0.0f);
        } else if (i < kBiasL2) b = w.layer1_bias[i - kBiasL1];
        else if (i < kBiasFc1) b = w.layer2_bias[i - kBiasL2];
        else if (i < kBiasHead) b = w.fc1_bias[i - kBiasFc1];
        else {
            const int c = i - kBiasHead;
            b = c < w.num_actions ? w.p_bias[c] : (c == w.num_actions ? w.v_bias[0] : 0.0f);
        }
        bias[i] = b;
    }
}
#endif

// Development aid (never in the product build): -DCAVOID_TRACE makes every workgroup record the constant-rate
// wall clock at its phase boundaries and the CU it ran on into g_pol_trace[block*16 + k] (tools/trace_policy.py).
#ifdef CAVOID_TRACE
__device__ unsigned long long *g_pol_trace = nullptr;
#define POLICY_STAMP(k)                                                                                      \
    do {                                                                                                     \
        if (threadIdx.x == 0 && g_pol_trace) g_pol_trace[(size_t)blockIdx.x * 16 + (k)] = wall_clock64();     \
    } while (0)
#else
#define POLICY_STAMP(k) do { } while (0)
#endif

// ---- forward ------------------------------------------------------------------------------------------
struct PolicyArgs {
    const float *x;                    // first policy input of row 0 (the env's obs row + 1: 'is_learning' skipped)
    int64_t rows, stride;              // stride between rows, in floats
    int max_other, num_actions, in_size;
    const float *avg, *std;            // [in_size]; nullptr = NORMALIZE_INPUT off
    const f32x4 *frags;
    const float *bias;
    float min_policy;
    float *p_out;                      // [rows, A]
    float *v_out;                      // [rows]
    // optional select_action (ProcessAgent.py:89-103): sample from p (or argmax when greedy) in the same launch
    int32_t *actions_out;              // [rows] or nullptr
    int greedy;
    uint32_t seed_lo, seed_hi;
    int32_t *step_counter;             // device-side: keys the random stream, advanced once per launch
    uint32_t *blocks_done;
    uint32_t *cu_tickets;              // [kPolCuSlots] arrival counters, one per compute unit (see the kernel)
    // optional row list (inference): process only rows row_index[0 .. *row_count), e.g. the agents that still act --
    // an agent that has finished waits for its world to end and needs no policy output.  Both live on the device, so
    // the launch geometry stays fixed (hipGraph replays): workgroups past the count leave at once.
    const int32_t *row_index;
    const int32_t *row_count;
    // TRAIN instantiation only (the trainer's forward pass): targets, the activations the backward pass and the
    // weight-gradient GEMMs need, and the gradient at the heads.  rows64 = rows rounded up to the 64-row tile.
    const float *y_r;                  // [rows] n-step returns
    const int32_t *a_idx;              // [rows] action taken
    float beta, log_eps;               // entropy weight, LOG_EPSILON
    int64_t rows64;
    float *z1, *z2, *z3;               // [rows64, 256] relu outputs of layer1, layer2, fullyconnected1
    float *l1_in;                      // [rows64, 72]  layer1's input in packed K order [h(64) | host(4) | 0]
    float *h_in;                       // [M, rows64, 72] LSTM step input [h_{t-1}(64) | x_t(7) | 0]
    float *save;                       // [tiles, M, 16, 256, 8] per lane and cell: i, j, f, o, c_{t-1}, tanh(c_t), 0, 0
    float *gh;                         // [rows64, 16] d cost / d (logits_p, logits_v)
    float *loss;                       // [2] cost_p, cost_v (summed over rows)
    float *db;                         // [kBiasFloats] bias gradients (column sums), same order as the packed biases
};

// Philox4x32-10, the same generator the scenario generator uses (cavoid_kernels.hpp)
__device__ __forceinline__ uint32_t policy_philox_x(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

// v_exp_f32 / v_rcp_f32 (1 ulp each): |error| of the gates ~1e-7, far inside the 1e-5 the outputs are held to
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// One 16-wide K chunk of fragments: RT row tiles of A (LDS) and 4 column tiles of B (packed weights, L2).
// RT = row tiles per workgroup (16*RT rows): 4 -> 64 rows, 2 workgroups per CU; 2 -> 32 rows, 4 workgroups per CU.
// CT = column tiles per wavefront: 4 for the 256-wide layers, 1 for the backward GEMMs whose output is the 64-wide LSTM state.
template <int RT, int CT = 4>
struct PolicyFrag { f32x4 a[RT], b[CT]; };

// A fragments of chunk ch start at LDS column 16*ch, except that chunk 4 (the "input" chunk of the two 80-wide
// layers) starts at `xcol`: the LSTM reads x_t and layer1 reads the host state where the prologue parked them.
template <int RT, int CT>
__device__ __forceinline__ void policy_load_a(PolicyFrag<RT, CT> &f, const float *arow, int ch, int xcol) {
    const int col = ch == 4 ? xcol : 16 * ch;
#pragma unroll
    for (int t = 0; t < RT; ++t) f.a[t] = *reinterpret_cast<const f32x4 *>(arow + 16 * t * kPolStride + col);
}

template <int RT, int CT>
__device__ __forceinline__ void policy_load_b(PolicyFrag<RT, CT> &f, const f32x4 *layer, int ct0, int lane, int ch) {
    const f32x4 *brow = layer + (int64_t)ct0 * 64 + lane;
    constexpr int64_t stride = CT == 4 ? kFragPerChunk : kFragPerChunkNarrow;
#pragma unroll
    for (int t = 0; t < CT; ++t) f.b[t] = brow[(int64_t)ch * stride + 64 * t];
}

template <int RT, int CT>
__device__ __forceinline__ void policy_mfma_chunk(const PolicyFrag<RT, CT> &f, f32x4 (&acc)[RT][CT]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[rt][s], f.b[ct][s], acc[rt][ct], 0, 0, 0);
}

// Issue order inside one chunk (16*RT MFMAs): the 4 weight loads and the RT LDS reads of the NEXT chunk go out one
// at a time, each after RT MFMAs (a wavefront can issue a few other instructions per 32-cycle MFMA slot; clustered at
// the chunk boundary they cost ~350 cycles per chunk), the remaining MFMAs cover their latency.
template <int RT, int CT>
__device__ __forceinline__ void policy_interleave() {
    if (CT != 4) return;                                   // the narrow GEMMs (16 MFMAs per chunk) are left to the compiler
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, RT, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, RT, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 16 * RT - RT * (4 + RT), 0);
}

// acc[rt][ct] += A(rows 16rt.., k chunks [c0, c1)) x B(column tiles ct0..ct0+3 of the packed layer).
// f0.b must already hold chunk c0's weight fragments (policy_load_b, issued BEFORE the barrier that publishes
// the activations: the weights do not depend on it, so their L2 latency hides behind the barrier).
// Two fragment sets ping-pong: the loads of chunk n+1 are issued before the 64 MFMAs (2048 cycles) of chunk n
// and are first needed after them, so neither the L2 nor the LDS latency is exposed.  The prefetches are
// unconditional (the last one is a harmless re-read): a branch around them makes the compiler wait for them
// at the join.
template <int RT, int CT>
__device__ __forceinline__ void policy_gemm(const float *act, const f32x4 *layer, int c0, int c1, int xcol, int ct0, int lane,
                                            PolicyFrag<RT, CT> &f0, f32x4 (&acc)[RT][CT]) {
    const float *arow = act + (lane & 15) * kPolStride + 4 * (lane >> 4);
    PolicyFrag<RT, CT> f1;
    policy_load_a(f0, arow, c0, xcol);
    int ch = c0;
    while (true) {
        const int n1 = ch + 1 < c1 ? ch + 1 : ch;
        policy_load_b(f1, layer, ct0, lane, n1);
        policy_load_a(f1, arow, n1, xcol);
        policy_mfma_chunk(f0, acc);
        policy_interleave<RT, CT>();
        if (ch + 1 >= c1) break;
        const int n2 = ch + 2 < c1 ? ch + 2 : ch;
        policy_load_b(f0, layer, ct0, lane, n2);
        policy_load_a(f0, arow, n2, xcol);
        policy_mfma_chunk(f1, acc);
        policy_interleave<RT, CT>();
        ch += 2;
        if (ch >= c1) break;
    }
}

template <int RT>
__device__ __forceinline__ void policy_init_acc(const float *bias, int ct0, int lane, f32x4 (&acc)[RT][4]) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const float b = bias[16 * (ct0 + ct) + (lane & 15)];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = f32x4{b, b, b, b};
    }
}

// relu(acc) -> act[row][col]: lane holds col = 16*(ct0+ct) + lane%16, rows 16rt + 4*(lane/16) + r.
// zg (trainer only): the same values to global memory [rows64, 256], first row of the tile = zg.
template <int RT>
__device__ __forceinline__ void policy_store_relu(float *act, int ct0, int lane, const f32x4 (&acc)[RT][4], float *zg = nullptr) {
    float *arow = act + (4 * (lane >> 4)) * kPolStride + 16 * ct0 + (lane & 15);
    float *grow = zg ? zg + (4 * (lane >> 4)) * kPolWidth + 16 * ct0 + (lane & 15) : nullptr;   // one base, constant offsets
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = fmaxf(acc[rt][ct][r], 0.0f);
                arow[(16 * rt + r) * kPolStride + 16 * ct] = z;
                if (zg) grow[(16 * rt + r) * kPolWidth + 16 * ct] = z;
            }
}

// e / d for 0 <= e < 2^20 and 1 <= d <= 512 without the integer-division sequence
__device__ __forceinline__ int policy_div(int e, int d, float inv_d) {
    int q = (int)((float)e * inv_d);
    q -= (q * d > e) ? 1 : 0;
    q += ((q + 1) * d <= e) ? 1 : 0;
    return q;
}

// the last workgroup of a launch that selects actions advances the device-side launch counter
__device__ __forceinline__ void policy_finish(const PolicyArgs &p, int step, int tid) {
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        if (atomicAdd(p.blocks_done, 1u) == gridDim.x - 1u) {
            *p.blocks_done = 0u;
            *p.step_counter = step + 1;
            __threadfence();
        }
    }
}

template <int RT, bool TRAIN>
__global__ void __launch_bounds__(256, TRAIN ? 1 : (RT == 4 ? 2 : 4)) policy_forward_kernel(const PolicyArgs p) {
    constexpr int kRows = 16 * RT;
    // LDS: activations [kRows][kPolStride], then the packed biases, then one int.  While the LSTM runs, a row is
    //   cols 0..63 h | 80 raw num_other | 84..87 host | 88+8t..94+8t x_t (t-th observed agent), zeros between
    extern __shared__ __attribute__((aligned(16))) float act[];
    float *lds_bias = act + kRows * kPolStride;
    int *wave_max = reinterpret_cast<int *>(lds_bias + kBiasFloats);
    int &ticket = wave_max[4];
    int *tile_row = wave_max + 8;                          // [kRows] global row of each tile row (identity without a row list)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t row0 = (int64_t)blockIdx.x * kRows;
    const int64_t n_rows = (!TRAIN && p.row_count) ? (int64_t)*p.row_count : p.rows;
    const int rows_here = n_rows - row0 < kRows ? (int)(n_rows - row0 > 0 ? n_rows - row0 : 0) : kRows;
    const int M = p.max_other;
    const int step = (!TRAIN && p.actions_out) ? *p.step_counter : 0;
    if (!TRAIN && p.row_index && rows_here == 0) {         // uniform over the workgroup: nothing listed for this tile
        if (p.actions_out) policy_finish(p, step, threadIdx.x);
        return;
    }
    POLICY_STAMP(0);
#ifdef CAVOID_TRACE
    const unsigned long long trace_c0 = clock64();
    if (tid == 0 && g_pol_trace)
        g_pol_trace[(size_t)blockIdx.x * 16 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                                  ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
#endif
    const f32x4 *w_lstm = p.frags + kOffLstm;
    PolicyFrag<RT, 4> f0;
    policy_load_b(f0, w_lstm, 4 * wave, lane, 4);          // first LSTM step: h == 0, only the input chunk contributes

    // ---- input tile: gather + normalise into the padded layout above ------------------------------------------
    // One trip to memory: every global load of the prologue (inputs, normalisation vectors, biases) is issued
    // before the first of them is consumed.
    if (!TRAIN && p.row_index) {                           // one extra (tiny) trip to memory: the tile's row list
        if (tid < kRows) tile_row[tid] = tid < rows_here ? p.row_index[row0 + tid] : 0;
        __syncthreads();
    }
    const bool listed = !TRAIN && p.row_index != nullptr;
    {
        const float *src = listed ? p.x : p.x + row0 * p.stride;
        const int wpad = 16 + 8 * M + 8;                   // padded row: [num,0,0,0, host(4), M x (x_t(7),0), 16 zeros]
        const float inv_wpad = 1.0f / (float)wpad;
        const int total = kRows * wpad;
        constexpr int U = 3 * RT;                          // M = 3: the whole tile in one pass
        int local_max = 0;
        float bias_v[(kBiasFloats + 255) / 256];
#pragma unroll
        for (int u = 0; u < (kBiasFloats + 255) / 256; ++u) bias_v[u] = tid + 256 * u < kBiasFloats ? p.bias[tid + 256 * u] : 0.0f;
        if (tid == 0) {
            // Two workgroups share a CU (one wavefront of each per SIMD).  Arrival parity on the CU decides a static
            // priority, so that the pair does not settle into lockstep (same phase at the same time, the matrix
            // pipe idle while both do their pointwise / barrier phases): without it CU pairs finish anywhere between
            // 85 and 131 us, with it every pair takes the same time.
            const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            const uint32_t key = ((xcc & 15u) << 8) | ((hw >> 8) & 0xFFu);        // cu_id[11:8] sh_id[12] se_id[15:13]
            ticket = (int)atomicAdd(p.cu_tickets + key, 1u);
        }
        for (int e0 = 0; e0 < total; e0 += 256 * U) {
            float v[U], av[U], sd[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                  // all the loads of the pass first
                const int e = e0 + u * 256 + tid;
                const int r = policy_div(e, wpad, inv_wpad), c = e - r * wpad;
                int sc = -1;                               // source column of this slot (-1: padding)
                if (c == 0) sc = 0;
                else if (c >= 4 && c < 8) sc = c - 3;
                else if (c >= 8 && c < 8 + 8 * M && (c & 7) != 7) sc = 1 + kPolHost + kPolOther * ((c - 8) >> 3) + (c & 7);
                const bool in = e < total && sc >= 0 && r < rows_here;
                dst[u] = e < total ? r * kPolStride + kPolXCol + c : -1;
                v[u] = in ? src[(int64_t)(listed ? tile_row[r] : r) * p.stride + sc] : 0.0f;
                const bool norm = in && sc > 0 && p.avg != nullptr;
                av[u] = norm ? p.avg[sc] : 0.0f;
                sd[u] = norm ? p.std[sc] : 1.0f;
                if (sc != 0) dst[u] |= dst[u] >= 0 ? 0x40000000 : 0;      // tag: not the length column
            }
            if (e0 == 0) {
                for (int e = tid; e < kRows * kPolHidden; e += 256) act[(e >> 6) * kPolStride + (e & 63)] = 0.0f;   // h = 0
#pragma unroll
                for (int u = 0; u < (kBiasFloats + 255) / 256; ++u)
                    if (tid + 256 * u < kBiasFloats) lds_bias[tid + 256 * u] = bias_v[u];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (dst[u] < 0) continue;
                if (!(dst[u] & 0x40000000)) {
                    int len = (int)v[u];
                    len = len < 0 ? 0 : (len > M ? M : len);
                    local_max = local_max > len ? local_max : len;
                }
                act[dst[u] & 0x3FFFFFFF] = (v[u] - av[u]) / sd[u];
            }
        }
        // tile-wide maximum without an initialising barrier: wavefront maxima into 4 LDS slots
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_xor(local_max, d, 64); local_max = o > local_max ? o : local_max; }
        if (lane == 0) wave_max[wave] = local_max;
    }
    __syncthreads();
    const int m01 = wave_max[0] > wave_max[1] ? wave_max[0] : wave_max[1], m23 = wave_max[2] > wave_max[3] ? wave_max[2] : wave_max[3];
    const int tile_max_len = m01 > m23 ? m01 : m23;
    const int steps = tile_max_len;                        // LSTM steps any row of this tile still needs
    if (ticket & 1) __builtin_amdgcn_s_setprio(1);
    POLICY_STAMP(5);

    // this lane's rows in the C layout and their sequence lengths
    float len_r[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) len_r[rt][r] = act[(16 * rt + 4 * (lane >> 4) + r) * kPolStride + kPolXCol];

    // ---- LSTM over the observed agents -------------------------------------------------------------------
    f32x4 cell[RT], hid[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { cell[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; hid[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int t = 0; t < steps; ++t) {
        if (TRAIN) {                                       // the step's input rows, for the LSTM weight gradient
            float *dst = p.h_in + ((int64_t)t * p.rows64 + row0) * 72;
            for (int e = tid; e < kRows * 72; e += 256) {
                const int r = e / 72, k = e - r * 72;
                dst[e] = k < kPolHidden ? act[r * kPolStride + k]
                                        : (k < kPolHidden + kPolOther ? act[r * kPolStride + kPolXCol + 8 + 8 * t + (k - kPolHidden)] : 0.0f);
            }
        }
        f32x4 acc[RT][4];
        policy_init_acc(lds_bias + kBiasLstm, 4 * wave, lane, acc);
        if (t == 1) POLICY_STAMP(8);
        policy_gemm(act, w_lstm, t == 0 ? 4 : 0, kChLstm, kPolXCol + 8 + 8 * t, 4 * wave, lane, f0, acc);
        if (t == 1) POLICY_STAMP(9);
        policy_load_b(f0, w_lstm, 4 * wave, lane, 0);      // the next step's first weight fragments
        __syncthreads();                                   // every wavefront has read h
        if (t == 1) POLICY_STAMP(10);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // dynamic_rnn: rows past their own length keep (c, h) -- selects, not branches
                const bool live = len_r[rt][r] > (float)t;
                const float gi = fast_sigmoid(acc[rt][0][r]), gj = fast_tanh(acc[rt][1][r]);
                const float gf = fast_sigmoid(acc[rt][2][r]), go = fast_sigmoid(acc[rt][3][r]);
                float keep = gf * cell[rt][r], add = gi * gj;
                if (TRAIN) asm volatile("" : "+v"(keep), "+v"(add));   // (no packed add with swapped halves: DESIGN.md 3.7 (d))
                const float c_new = keep + add;
                const float tc = fast_tanh(c_new);
                const float h_new = go * tc;
                if (TRAIN) {                               // lane-private 32-byte records: the backward pass reads them back as is
                    f32x4 *sv = reinterpret_cast<f32x4 *>(p.save) + ((((int64_t)blockIdx.x * M + t) * 16 + (rt * 4 + r)) * 256 + tid) * 2;
                    sv[0] = f32x4{gi, gj, gf, go};
                    sv[1] = f32x4{cell[rt][r], tc, 0.0f, 0.0f};
                }
                cell[rt][r] = live ? c_new : cell[rt][r];
                hid[rt][r] = live ? h_new : hid[rt][r];
                act[(16 * rt + 4 * (lane >> 4) + r) * kPolStride + 16 * wave + (lane & 15)] = hid[rt][r];
            }
        if (t == 1) POLICY_STAMP(11);
        __syncthreads();                                   // the new h is in place
        if (t == 1) POLICY_STAMP(12);
    }
    POLICY_STAMP(1);
    // ---- layer1 on [h | host] --------------------------------------------------------------------------------
    {
        if (TRAIN) {                                       // layer1's input rows [h | host | 0], for its weight gradient
            float *dst = p.l1_in + row0 * 72;
            for (int e = tid; e < kRows * 72; e += 256) {
                const int r = e / 72, k = e - r * 72;
                dst[e] = k < kPolHidden ? act[r * kPolStride + k]
                                        : (k < kPolHidden + kPolHost ? act[r * kPolStride + kPolXCol + 4 + (k - kPolHidden)] : 0.0f);
            }
        }
        f32x4 acc[RT][4];
        policy_load_b(f0, p.frags + kOffL1, 4 * wave, lane, 0);
        policy_init_acc(lds_bias + kBiasL1, 4 * wave, lane, acc);
        policy_gemm(act, p.frags + kOffL1, 0, kChL1, kPolXCol + 4, 4 * wave, lane, f0, acc);
        policy_load_b(f0, p.frags + kOffL2, 4 * wave, lane, 0);
        __syncthreads();
        policy_store_relu(act, 4 * wave, lane, acc, TRAIN ? p.z1 + row0 * kPolWidth : nullptr);
        __syncthreads();
    }
    POLICY_STAMP(2);
    // ---- layer2, fullyconnected1 -----------------------------------------------------------------------------
    {
        f32x4 acc[RT][4];
        policy_init_acc(lds_bias + kBiasL2, 4 * wave, lane, acc);
        policy_gemm(act, p.frags + kOffL2, 0, kChWide, 64, 4 * wave, lane, f0, acc);
        policy_load_b(f0, p.frags + kOffFc1, 4 * wave, lane, 0);
        __syncthreads();
        policy_store_relu(act, 4 * wave, lane, acc, TRAIN ? p.z2 + row0 * kPolWidth : nullptr);
        __syncthreads();
    }
    f32x4 hb[kChHead];                                     // the heads' weight fragments: in flight across the last layer's
    {                                                      // epilogue and barrier
        f32x4 acc[RT][4];
        policy_init_acc(lds_bias + kBiasFc1, 4 * wave, lane, acc);
        policy_gemm(act, p.frags + kOffFc1, 0, kChWide, 64, 4 * wave, lane, f0, acc);
        const f32x4 *brow = p.frags + kOffHead + lane;
        if (RT == 4 && !TRAIN) {                           // 2 wavefronts per SIMD: the registers are there
#pragma unroll
            for (int ch = 0; ch < kChHead; ++ch) hb[ch] = brow[64 * ch];
        }
        __syncthreads();
        policy_store_relu(act, 4 * wave, lane, acc, TRAIN ? p.z3 + row0 * kPolWidth : nullptr);
        __syncthreads();
        if (RT != 4 || TRAIN) {                            // tighter register budget: half now, half below
#pragma unroll
            for (int ch = 0; ch < kChHead / 2; ++ch) hb[ch] = brow[64 * ch];
        }
    }
    POLICY_STAMP(3);
    // ---- heads: wavefront w < RT does rows 16w..16w+15 x 16 columns (A logits, the value, padding) ------------
    if (wave < RT) {
        f32x4 acc[4];
        const float b = lds_bias[kBiasHead + (lane & 15)];
        acc[0] = f32x4{b, b, b, b};
        acc[1] = acc[2] = acc[3] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *arow = act + (16 * wave + (lane & 15)) * kPolStride + 4 * (lane >> 4);
#pragma unroll
        for (int g = 0; g < kChHead; g += 4) {
            if ((RT != 4 || TRAIN) && g == 0) {
#pragma unroll
                for (int ch = kChHead / 2; ch < kChHead; ++ch) hb[ch] = p.frags[kOffHead + lane + 64 * ch];
            }
            f32x4 ha[4];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) ha[ch] = *reinterpret_cast<const f32x4 *>(arow + 16 * (g + ch));
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[ch][s], hb[g + ch][s], acc[s], 0, 0, 0);
        }
        const f32x4 logit = acc[0] + acc[1] + acc[2] + acc[3];
        const int col = lane & 15, A = p.num_actions;
        const float scale = 1.0f / (1.0f + p.min_policy * (float)A);
        float cost_p = 0.0f, cost_v = 0.0f, gsum = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float z = logit[r];
            float m = col < A ? z : -INFINITY;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) m = fmaxf(m, __shfl_xor(m, d, 16));
            const float e = col < A ? expf(z - m) : 0.0f;
            float sum = e;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) sum += __shfl_xor(sum, d, 16);
            const int trow = 16 * wave + 4 * (lane >> 4) + r;             // row of the tile
            const bool in_tile = TRAIN || trow < rows_here;
            const int64_t row = listed ? (in_tile ? (int64_t)tile_row[trow] : p.rows) : row0 + trow;     // global row
            const float sm = e / sum;                      // softmax
            const float pj = col < A ? (sm + p.min_policy) * scale : 0.0f;
            if (!TRAIN && row < p.rows) {
                if (col < A) p.p_out[row * A + col] = pj;
                else if (col == A) p.v_out[row] = z;
            }
            if (TRAIN) {
                // A3C loss of NetworkVPCore.py:71-100 (sums over rows) and its gradient at the logits:
                //   cost_v = 0.5 (y - v)^2;  cost_p = -[ log(max(p_a, eps)) (y - v_detached) - beta sum_k log(max(p_k, eps)) p_k ]
                const bool valid = row < p.rows;
                const float y = valid ? p.y_r[row] : 0.0f;
                const int a = valid ? p.a_idx[row] : 0;
                const float v = __shfl(z, A, 16);
                const float sel = __shfl(pj, a, 16);
                const float lp = __logf(fmaxf(pj, p.log_eps));
                // d cost_p / d p'_k, then through p' = (softmax + MIN_POLICY) * scale and the softmax
                float dp = p.beta * (pj > p.log_eps ? lp + 1.0f : lp);
                if (col == a && sel > p.log_eps) dp -= (y - v) / sel;
                dp = col < A ? dp * scale : 0.0f;
                float dot = sm * dp;
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) dot += __shfl_xor(dot, d, 16);
                float g = col < A ? sm * (dp - dot) : (col == A ? v - y : 0.0f);
                if (!valid) g = 0.0f;
                p.gh[(row0 + 16 * wave + 4 * (lane >> 4) + r) * 16 + col] = g;
                gsum += g;
                float ent = col < A ? lp * pj : 0.0f;
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) ent += __shfl_xor(ent, d, 16);
                if (valid && col == 0) {
                    cost_p -= __logf(fmaxf(sel, p.log_eps)) * (y - v) - p.beta * ent;
                    cost_v += 0.5f * (y - v) * (y - v);
                }
            }
            if (!TRAIN && p.actions_out) {                 // wave-uniform
                int action;
                if (p.greedy) {                            // np.argmax: first index of the maximum
                    float best = pj;
#pragma unroll
                    for (int d = 1; d < 16; d <<= 1) best = fmaxf(best, __shfl_xor(best, d, 16));
                    int idx = (col < A && pj == best) ? col : 99;
#pragma unroll
                    for (int d = 1; d < 16; d <<= 1) { const int o = __shfl_xor(idx, d, 16); idx = o < idx ? o : idx; }
                    action = idx;
                } else {                                   // inverse CDF: #{c : cdf_c <= u * cdf_{A-1}}
                    float cdf = pj;
#pragma unroll
                    for (int d = 1; d < 16; d <<= 1) { const float t = __shfl_up(cdf, d, 16); if (col >= d) cdf += t; }
                    const float total = __shfl(cdf, A - 1, 16);
                    const uint32_t bits = policy_philox_x((uint32_t)row, (uint32_t)((uint64_t)row >> 32), (uint32_t)step, 0x504F4Cu,
                                                          p.seed_lo, p.seed_hi);
                    const float u = (float)(bits >> 8) * (1.0f / 16777216.0f);
                    int below = (col < A && cdf <= u * total) ? 1 : 0;
#pragma unroll
                    for (int d = 1; d < 16; d <<= 1) below += __shfl_xor(below, d, 16);
                    action = below < A - 1 ? below : A - 1;
                }
                if (row < p.rows && col == 0) p.actions_out[row] = action;
            }
        }
        if (TRAIN) {                                       // one atomic pair per wavefront
            cost_p += __shfl_xor(cost_p, 16, 64); cost_p += __shfl_xor(cost_p, 32, 64);
            cost_v += __shfl_xor(cost_v, 16, 64); cost_v += __shfl_xor(cost_v, 32, 64);
            if (lane == 0) { atomicAdd(p.loss, cost_p); atomicAdd(p.loss + 1, cost_v); }
            gsum += __shfl_xor(gsum, 16, 64); gsum += __shfl_xor(gsum, 32, 64);
            if (lane < 16) atomicAdd(p.db + kBiasHead + lane, gsum);
        }
    }
    POLICY_STAMP(4);
#ifdef CAVOID_TRACE
    if (tid == 0 && g_pol_trace) g_pol_trace[(size_t)blockIdx.x * 16 + 6] = clock64() - trace_c0;   // shader-clock cycles
#endif
    if (!TRAIN && p.actions_out) policy_finish(p, step, tid);
}

// ---- backward (trainer) ---------------------------------------------------------------------------------
// Everything of the backward pass that is row-local: the gradient at the heads (from the TRAIN forward) is pushed back
// through fullyconnected1, layer2, layer1 and the LSTM, 64 rows per workgroup, same LDS / fragment machinery as the
// forward (dX = dY x W^T with the transposed packs).  What it leaves in memory -- the masked gradient at every layer's
// output (g3, g2, g1, the per-step gate gradients gl) next to the forward's saved layer inputs -- is exactly the operand
// pair of each weight-gradient GEMM (X^T x G over all rows), which is a plain large GEMM and stays a library call.
struct PolicyBackArgs {
    const float *x;                    // [rows, stride]: column 0 = num_other_agents (the LSTM length)
    int64_t rows, stride, rows64;
    int max_other;
    const f32x4 *frags;
    const float *z1, *z2, *z3, *save, *gh;
    float *g1, *g2, *g3;               // [rows64, 256]
    float *gl;                         // [M, rows64, 256], gate columns in packed order (64w + 16 gate + u)
    float *db;                         // [kBiasFloats] bias gradients, accumulated with atomics (packed bias order)
};

// g = acc where the layer's relu was active, else 0 -> LDS (the next GEMM's A operand) and global memory
template <int RT>
__device__ __forceinline__ void policy_store_masked(float *act, int ct0, int lane, const f32x4 (&acc)[RT][4], const float *zg, float *gg,
                                                    float *db) {
    const int lane_off = (4 * (lane >> 4)) * kPolWidth + 16 * ct0 + (lane & 15);
    const float *zrow = zg + lane_off;
    float *grow = gg + lane_off;
    float *arow = act + (4 * (lane >> 4)) * kPolStride + 16 * ct0 + (lane & 15);
    float z[RT][4][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) z[rt][ct][r] = zrow[(16 * rt + r) * kPolWidth + 16 * ct];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float g = z[rt][ct][r] > 0.0f ? acc[rt][ct][r] : 0.0f;
                arow[(16 * rt + r) * kPolStride + 16 * ct] = g;
                grow[(16 * rt + r) * kPolWidth + 16 * ct] = g;
                z[rt][ct][r] = g;
            }
    // bias gradient = column sums: 16 rows per lane, then the 4 lane groups, one atomic per column and wavefront
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        float sum = 0.0f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sum += z[rt][ct][r];
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        if (lane < 16) atomicAdd(db + 16 * (ct0 + ct) + lane, sum);
    }
}

template <int RT>
__device__ __forceinline__ void policy_zero_acc(f32x4 (&acc)[RT][4]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int RT>
__global__ void __launch_bounds__(256, 2) policy_backward_kernel(const PolicyBackArgs p) {
    constexpr int kRows = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float act[];
    int *wave_max = reinterpret_cast<int *>(act + kRows * kPolStride);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t row0 = (int64_t)blockIdx.x * kRows;
    const int M = p.max_other;

    PolicyFrag<RT, 4> f0;
    policy_load_b(f0, p.frags + kOffTHead, 4 * wave, lane, 0);
    // sequence lengths of this lane's rows (C layout), the tile's step count, and the head gradient into LDS
    float len_r[RT][4];
    int local_max = 0;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + 16 * rt + 4 * (lane >> 4) + r;
            const float len = row < p.rows ? p.x[row * p.stride] : 0.0f;
            len_r[rt][r] = len;
            int li = (int)len;
            li = li < 0 ? 0 : (li > M ? M : li);
            local_max = local_max > li ? local_max : li;
        }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_xor(local_max, d, 64); local_max = o > local_max ? o : local_max; }
    if (lane == 0) wave_max[wave] = local_max;
    for (int e = tid; e < kRows * 16; e += 256) act[(e >> 4) * kPolStride + (e & 15)] = p.gh[row0 * 16 + e];
    __syncthreads();
    const int m01 = wave_max[0] > wave_max[1] ? wave_max[0] : wave_max[1], m23 = wave_max[2] > wave_max[3] ? wave_max[2] : wave_max[3];
    const int steps = m01 > m23 ? m01 : m23;

    // ---- heads^T, fullyconnected1^T, layer2^T: 256-wide outputs, masked by the forward's relu ---------------------
    {
        f32x4 acc[RT][4];
        policy_zero_acc(acc);
        policy_gemm(act, p.frags + kOffTHead, 0, 1, 64, 4 * wave, lane, f0, acc);
        policy_load_b(f0, p.frags + kOffTFc1, 4 * wave, lane, 0);
        __syncthreads();
        policy_store_masked(act, 4 * wave, lane, acc, p.z3 + row0 * kPolWidth, p.g3 + row0 * kPolWidth, p.db + kBiasFc1);
        __syncthreads();
    }
    {
        f32x4 acc[RT][4];
        policy_zero_acc(acc);
        policy_gemm(act, p.frags + kOffTFc1, 0, kChWide, 64, 4 * wave, lane, f0, acc);
        policy_load_b(f0, p.frags + kOffTL2, 4 * wave, lane, 0);
        __syncthreads();
        policy_store_masked(act, 4 * wave, lane, acc, p.z2 + row0 * kPolWidth, p.g2 + row0 * kPolWidth, p.db + kBiasL2);
        __syncthreads();
    }
    {
        f32x4 acc[RT][4];
        policy_zero_acc(acc);
        policy_gemm(act, p.frags + kOffTL2, 0, kChWide, 64, 4 * wave, lane, f0, acc);
        __syncthreads();
        policy_store_masked(act, 4 * wave, lane, acc, p.z1 + row0 * kPolWidth, p.g1 + row0 * kPolWidth, p.db + kBiasL1);
        __syncthreads();
    }
    // ---- layer1^T, hidden-state inputs only: d cost / d h_final, one column tile per wavefront = its 16 hidden units --
    PolicyFrag<RT, 1> n0;
    f32x4 dh[RT], dc[RT];
    {
        f32x4 acc1[RT][1];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc1[rt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        policy_load_b(n0, p.frags + kOffTL1, wave, lane, 0);
        policy_gemm(act, p.frags + kOffTL1, 0, kChWide, 64, wave, lane, n0, acc1);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) { dh[rt] = acc1[rt][0]; dc[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    __syncthreads();                                       // g1 has been read by every wavefront
    // ---- LSTM, backwards through the observed agents ------------------------------------------------------------
    for (int t = M - 1; t >= 0; --t) {
        float *glt = p.gl + ((int64_t)t * p.rows64 + row0) * kPolWidth;
        if (t >= steps) {                                  // no row of this tile took step t
            for (int e = tid; e < kRows * kPolWidth; e += 256) glt[e] = 0.0f;
            continue;
        }
        policy_load_b(n0, p.frags + kOffTLstm, wave, lane, 0);
        f32x4 dc_new[RT];
        float bsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};          // gate-bias gradient: column sums of this lane's 16 cells
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x4 *sv = reinterpret_cast<const f32x4 *>(p.save) + ((((int64_t)blockIdx.x * M + t) * 16 + (rt * 4 + r)) * 256 + tid) * 2;
                const f32x4 s0 = sv[0], s1 = sv[1];
                const float gi = s0[0], gj = s0[1], gf = s0[2], go = s0[3], cprev = s1[0], tc = s1[1];
                const bool live = len_r[rt][r] > (float)t;
                const float dhv = dh[rt][r];
                const float dct = dc[rt][r] + dhv * go * (1.0f - tc * tc);
                const float d_o = live ? dhv * tc * go * (1.0f - go) : 0.0f;
                const float d_i = live ? dct * gj * gi * (1.0f - gi) : 0.0f;
                const float d_j = live ? dct * gi * (1.0f - gj * gj) : 0.0f;
                const float d_f = live ? dct * cprev * gf * (1.0f - gf) : 0.0f;
                dc_new[rt][r] = live ? dct * gf : dc[rt][r];
                const int row = 16 * rt + 4 * (lane >> 4) + r, col = 64 * wave + (lane & 15);
                float *a = act + row * kPolStride + col;
                float *g = glt + (int64_t)row * kPolWidth + col;
                a[0] = d_i; a[16] = d_j; a[32] = d_f; a[48] = d_o;
                g[0] = d_i; g[16] = d_j; g[32] = d_f; g[48] = d_o;
                bsum[0] += d_i; bsum[1] += d_j; bsum[2] += d_f; bsum[3] += d_o;
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float sum = bsum[q];
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            if (lane < 16) atomicAdd(p.db + kBiasLstm + 64 * wave + 16 * q + lane, sum);
        }
        __syncthreads();
        f32x4 acc1[RT][1];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc1[rt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        policy_gemm(act, p.frags + kOffTLstm, 0, kChWide, 64, wave, lane, n0, acc1);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool live = len_r[rt][r] > (float)t;
                dh[rt][r] = live ? acc1[rt][0][r] : dh[rt][r];     // rows past their length: the step was the identity
                dc[rt][r] = dc_new[rt][r];
            }
        __syncthreads();
    }
}

}  // namespace cavoid
