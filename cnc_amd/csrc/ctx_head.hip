// ctx_head.hip — the context models' prediction heads and the Bernoulli rate, fused, for gfx950.
//
// In the reference (examples/utils_bpp_acc.py) every coded level runs, per training step,
//     cat([context, context_pn, Pg]) -> Linear(+LeakyReLU, Linear, LeakyReLU, Linear) (:378-393, :561-566, :689-692)
//     -> hash fusion (:567-572, :693-701) -> clamp / log2 / masks / sum (Bernoulli_entropy :1002-1013)
// as ~30 ATen launches forward and ~60 backward, 9 plane levels + the batched 3-D levels per step.  Here:
//
//   k_ctx_mlp_fwd<1,F>    the single-Linear heads (Linear(C->F)), one lane per vertex: the row [ctx | pn | Pg] is read in
//                         place (no cat), weights transposed in LDS (broadcast reads)
//   k_ctx_mlp_bwd<1,F>    back-propagates to the inputs and reduces the weight gradients over the 64 vertices of a wave
//                         through LDS tiles (MFMA outer products), one atomicAdd per element and block at the end
//   k_ctx_head3_fwd/bwd   the three-layer head (C->32->32->F, LeakyReLU 0.01) on the matrix cores, 16 vertices per wave: a
//                         layer's output is the next layer's operand register for register (see below)
//   k_bernoulli_bits      bits = sum over (slot, feature) of -log2(p) [x=+1] / -log2(1-p) [x=-1], p = clamp(mean),
//                         x gathered from the table by row; per-block partial sums (deterministic total)
//   k_bernoulli_bits_bwd  d bits / d mean and d bits / d x in one pass
//   k_segment_bwd         gradient of the hash fusion (cnc_segment_weighted_sum) w.r.t. the per-vertex values
//
// Arithmetic: fp32 with fmaf; the GEMMs these replace have no defined summation order either.  Tested against
// torch autograd of the op chain (tests/test_gpu_ctx_head.py).
#include "common.hpp"

namespace cnc {

constexpr int   kH = 32;            // hidden width of context_model_3D (utils_bpp_acc.py:378-384)
constexpr float kSlope = 0.01f;     // nn.LeakyReLU() default
constexpr int   kMaxC = 40;         // widest input row: 3 context levels x 8 + 8 (dimension-wise) + 1 (Pg) = 33

__device__ __forceinline__ float lrelu(float v) { return v > 0.0f ? v : kSlope * v; }

struct MlpArgs {
    const float* in_a; uint32_t lda, Ca;      // [N, Ca] rows with leading dimension lda
    const float* in_b; uint32_t ldb, Cb;      // [N, Cb] or null
    const float* pg;                          // device scalar appended as the last column, or null
    const int64_t* pg_index;                  // null, or per-row index into pg[] (rows of different levels)
    uint32_t     N, C;                        // C = Ca + Cb + (pg ? 1 : 0)
    const float *W1, *b1, *W2, *b2, *W3, *b3; // nn.Linear layout [out, in]; W2 / W3 unused for NL == 1
};

// weights into LDS, TRANSPOSED ([in][out]) so that the per-input-element inner loops read consecutive words
template <int NL, int F>
__device__ __forceinline__ void load_weights(const MlpArgs& a, float* sW1t, float* sb1, float* sW2t, float* sb2,
                                             float* sW3t, float* sb3)
{
    constexpr int H1 = NL == 1 ? F : kH;
    for (uint32_t e = threadIdx.x; e < H1 * a.C; e += blockDim.x) {
        const uint32_t j = e / a.C, c = e % a.C;
        sW1t[c * H1 + j] = a.W1[e];
    }
    for (uint32_t e = threadIdx.x; e < H1; e += blockDim.x) sb1[e] = a.b1[e];
    if constexpr (NL == 3) {
        for (uint32_t e = threadIdx.x; e < kH * kH; e += blockDim.x) sW2t[(e % kH) * kH + e / kH] = a.W2[e];
        for (uint32_t e = threadIdx.x; e < F * kH; e += blockDim.x) sW3t[(e % kH) * F + e / kH] = a.W3[e];
        for (uint32_t e = threadIdx.x; e < kH; e += blockDim.x) sb2[e] = a.b2[e];
        for (uint32_t e = threadIdx.x; e < F; e += blockDim.x) sb3[e] = a.b3[e];
    }
}

__device__ __forceinline__ float input_at(const MlpArgs& a, uint32_t row, uint32_t c)
{
    if (c < a.Ca) return a.in_a[(size_t)row * a.lda + c];
    if (c < a.Ca + a.Cb) return a.in_b[(size_t)row * a.ldb + (c - a.Ca)];
    return a.pg[a.pg_index ? a.pg_index[row] : 0];
}

// Columns [c0, c0 + 4) of a row as one 16-byte access when the segment allows it (base pointer and leading
// dimension multiples of 4 floats — true for the encoder outputs, which are 8 k wide), scalar otherwise.
__device__ __forceinline__ bool seg_vec_ok(const float* p, uint32_t ld, uint32_t n)
{
    return (((uintptr_t)p | (uintptr_t)(ld * sizeof(float))) & 15u) == 0 && (n & 3u) == 0;
}

// calls fn(c, value) for every column of the input row [in_a | in_b | pg]; 16-byte loads where possible
// (statically unrolled: a runtime-indexed float[4] here sent the NL = 3 kernels to scratch memory)
template <class Fn>
__device__ __forceinline__ void for_each_input(const MlpArgs& a, uint32_t row, Fn fn)
{
    uint32_t c = 0;
    auto seg = [&](const float* base, uint32_t ld, uint32_t n) {
        const float* r = base + (size_t)row * ld;
        if (seg_vec_ok(base, ld, n)) {
            for (uint32_t k = 0; k < n; k += 4, c += 4) {
                const float4 v = *reinterpret_cast<const float4*>(r + k);
                fn(c, v.x);
                fn(c + 1, v.y);
                fn(c + 2, v.z);
                fn(c + 3, v.w);
            }
        } else {
            for (uint32_t k = 0; k < n; k++, c++) fn(c, r[k]);
        }
    };
    seg(a.in_a, a.lda, a.Ca);
    if (a.Cb) seg(a.in_b, a.ldb, a.Cb);
    if (a.pg) fn(c, a.pg[a.pg_index ? a.pg_index[row] : 0]);
}

// forward of one vertex; h1 / h2 hold the POST-activation hidden values (NL == 3)
template <int NL, int F>
__device__ __forceinline__ void mlp_row(const MlpArgs& a, uint32_t row, const float* sW1t, const float* sb1,
                                        const float* sW2t, const float* sb2, const float* sW3t, const float* sb3,
                                        float (&h1)[NL == 1 ? F : kH], float (&h2)[kH], float (&out)[F])
{
    constexpr int H1 = NL == 1 ? F : kH;
#pragma unroll
    for (int j = 0; j < H1; j++) h1[j] = sb1[j];
    for_each_input(a, row, [&](uint32_t c, float v) {
        const float* w = sW1t + c * H1;
#pragma unroll
        for (int j = 0; j < H1; j++) h1[j] = __builtin_fmaf(w[j], v, h1[j]);
    });
    if constexpr (NL == 1) {
#pragma unroll
        for (int f = 0; f < F; f++) out[f] = h1[f];
    } else {
#pragma unroll
        for (int j = 0; j < kH; j++) { h1[j] = lrelu(h1[j]); h2[j] = sb2[j]; }
#pragma unroll
        for (int i = 0; i < kH; i++) {
            const float* w = sW2t + i * kH;
#pragma unroll
            for (int j = 0; j < kH; j++) h2[j] = __builtin_fmaf(w[j], h1[i], h2[j]);
            // keep the scheduler from hoisting all 1024 weight reads of the unrolled layer to the top (that took
            // 256 VGPRs + 256 AGPRs + 932 B of scratch per lane)
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int f = 0; f < F; f++) out[f] = sb3[f];
#pragma unroll
        for (int j = 0; j < kH; j++) {
            h2[j] = lrelu(h2[j]);
            const float* w = sW3t + j * F;
#pragma unroll
            for (int f = 0; f < F; f++) out[f] = __builtin_fmaf(w[f], h2[j], out[f]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int NL, int F>
__global__ __launch_bounds__(256) void k_ctx_mlp_fwd(MlpArgs a, float* __restrict__ out)
{
    constexpr int H1 = NL == 1 ? F : kH;
    __shared__ float sW1t[kMaxC * H1], sb1[H1], sW2t[NL == 3 ? kH * kH : 1], sb2[kH], sW3t[NL == 3 ? kH * F : 1], sb3[F];
    load_weights<NL, F>(a, sW1t, sb1, sW2t, sb2, sW3t, sb3);
    __syncthreads();
    for (uint32_t row = blockIdx.x * blockDim.x + threadIdx.x; row < a.N; row += gridDim.x * blockDim.x) {
        float h1[H1], h2[kH], o[F];
        mlp_row<NL, F>(a, row, sW1t, sb1, sW2t, sb2, sW3t, sb3, h1, h2, o);
#pragma unroll
        for (int f = 0; f < F; f++) out[(size_t)row * F + f] = o[f];
    }
}

struct MlpGrads {
    const float* g_out;                       // [N, F]
    float *g_a, *g_b, *g_pg;                  // [N, Ca], [N, Cb] or null, scalar accumulator or null
    uint32_t ldga, ldgb;                      // row pitch of g_a / g_b (Ca / Cb when packed)
    float *gW1, *gb1, *gW2, *gb2, *gW3, *gb3; // accumulated with atomicAdd: zeroed by the caller
    uint32_t n_rep, rep_stride;               // block b adds into copy b % n_rep, rep_stride floats further on
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One wave reduces outer products over its 64 vertices on the matrix cores:
//     G[i][j] += sum_v P[v][i] * Q[v][j],   i < NA <= 16 TA,  j < NB <= 16 TB,
// as 16 steps of v_mfma_f32_16x16x4_f32 per 16 x 16 tile of G (K = 4 vertices per step).  Lane l feeds vertex
// 4 s + l / 16 with column l % 16 of P (operand A) and of Q (operand B): ONE LDS read per lane, operand and step —
// 16 (TA + TB) reads per lane and batch where a lane-owns-elements loop read 2 x 64 words per owned element
// (2048 reads per lane for the 32 x 32 layer: the whole kernel sat on the LDS pipe, 9-13 x its forward).
// P / Q tiles live in the wave's LDS region ([64][pitch]); tile (ta, tb) of G ends up in acc[ta][tb]: element
// (row, col) in lane col + 16 (row / 4), register row % 4.  `ntb` = column tiles in use (wave-uniform).
template <int TA, int TB>
__device__ __forceinline__ void outer_mfma(const float* tP, uint32_t pitchP, uint32_t NA, const float* tQ,
                                           uint32_t pitchQ, uint32_t NB, uint32_t ntb, uint32_t lane,
                                           f32x4 (&acc)[TA][TB])
{
    const uint32_t c = lane & 15u, k = lane >> 4;
#pragma unroll 4
    for (uint32_t s = 0; s < 16; s++) {
        const uint32_t v = 4u * s + k;
        float a[TA], b[TB];
#pragma unroll
        for (int ta = 0; ta < TA; ta++) {
            const uint32_t i = 16u * ta + c;
            a[ta] = i < NA ? tP[v * pitchP + i] : 0.0f;
        }
#pragma unroll
        for (int tb = 0; tb < TB; tb++) {
            const uint32_t j = 16u * tb + c;
            b[tb] = j < NB ? tQ[v * pitchQ + j] : 0.0f;
        }
#pragma unroll
        for (int ta = 0; ta < TA; ta++)
#pragma unroll
            for (int tb = 0; tb < TB; tb++)
                if ((uint32_t)tb < ntb) acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
    }
}

// adds the accumulated tiles into g[NA][NB] (nn.Linear layout [out, in])
template <int TA, int TB>
__device__ __forceinline__ void flush_mfma(float* g, uint32_t NA, uint32_t NB, uint32_t lane, const f32x4 (&acc)[TA][TB])
{
#pragma unroll
    for (int ta = 0; ta < TA; ta++)
#pragma unroll
        for (int tb = 0; tb < TB; tb++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t i = 16u * ta + 4u * (lane >> 4) + r, j = 16u * tb + (lane & 15u);
                if (i < NA && j < NB) atomicAdd(g + i * NB + j, acc[ta][tb][r]);
            }
}

constexpr int kBwdThreads = 128;    // two waves per block: each needs two 64-row LDS tiles (2 x 10.5 KB)

template <int NL, int F>
__global__ __launch_bounds__(kBwdThreads) void k_ctx_mlp_bwd(MlpArgs a, MlpGrads g)
{
    constexpr int H1 = NL == 1 ? F : kH;
    constexpr int kPitch = kMaxC + 1;                         // tile pitch (words); covers 32 + 1 and C + 1
    __shared__ float sW1t[kMaxC * H1], sb1[H1], sW2t[NL == 3 ? kH * kH : 1], sb2[kH], sW3t[NL == 3 ? kH * F : 1], sb3[F];
    // per wave: tile A (gradients at a layer's output: <= 32 columns, F for the single-Linear heads) and tile B (that
    // layer's inputs: <= kMaxC columns).  The narrow tile A of NL == 1 takes 25.6 instead of 42 KB per block:
    // 6 instead of 3 blocks per CU for the nine 2-D heads of a step.
    constexpr int kPitchA = NL == 1 ? F + 1 : kPitch;
    __shared__ float tilesA[kBwdThreads / 64][64 * kPitchA];
    __shared__ float tilesB[kBwdThreads / 64][64 * kPitch];
    load_weights<NL, F>(a, sW1t, sb1, sW2t, sb2, sW3t, sb3);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* tA = tilesA[wave];
    float* tB = tilesB[wave];
    // weight-gradient tiles of this wave (accumulated over every batch of the block, MFMA layout: outer_mfma)
    constexpr int T1A = (H1 + 15) / 16, T1B = (kMaxC + 15) / 16, TH = kH / 16, TF = (F + 15) / 16;
    f32x4 aW1[T1A][T1B] = {}, aW2[NL == 3 ? TH : 1][NL == 3 ? TH : 1] = {}, aW3[NL == 3 ? TF : 1][NL == 3 ? TH : 1] = {};
    float ab1 = 0, ab2 = 0, ab3 = 0, apg = 0;
    const uint32_t ntb1 = (a.C + 15u) / 16u;
    int64_t pg_at = -1;            // pg_index mode: the table entry this WAVE is accumulating for (wave-uniform)

    const uint32_t n_batches = (a.N + kBwdThreads - 1) / kBwdThreads;
    for (uint32_t batch = blockIdx.x; batch < n_batches; batch += gridDim.x) {
        const uint32_t row = batch * kBwdThreads + threadIdx.x;
        const bool     on = row < a.N;
        float h1[H1], h2[kH], o[F], d_o[F];
        float d1[H1];                                          // gradient at the first layer's pre-activation
#pragma unroll
        for (int f = 0; f < F; f++) d_o[f] = on ? g.g_out[(size_t)row * F + f] : 0.0f;
        if (on && NL == 3) mlp_row<NL, F>(a, row, sW1t, sb1, sW2t, sb2, sW3t, sb3, h1, h2, o);   // a single Linear
        else {                                                                                   // needs no activations
#pragma unroll
            for (int j = 0; j < H1; j++) h1[j] = 0.0f;
#pragma unroll
            for (int j = 0; j < kH; j++) h2[j] = 0.0f;
        }
        if constexpr (NL == 1) {
#pragma unroll
            for (int f = 0; f < F; f++) d1[f] = d_o[f];
        } else {
            float d2[kH];
            // d2 = (W3^T d_out) * lrelu'(a2);  tiles: A = d_out [64][F], B = h2 [64][32] -> dW3
#pragma unroll
            for (int j = 0; j < kH; j++) {
                float s = 0.0f;
#pragma unroll
                for (int f = 0; f < F; f++) s = __builtin_fmaf(sW3t[j * F + f], d_o[f], s);
                d2[j] = h2[j] > 0.0f ? s : kSlope * s;
                tB[lane * kPitch + j] = h2[j];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int f = 0; f < F; f++) tA[lane * kPitchA + f] = d_o[f];
            __syncthreads();
            outer_mfma<TF, TH>(tA, kPitchA, F, tB, kPitch, kH, TH, lane, aW3);
            if (lane < F) {
                float s = 0.0f;
                for (uint32_t v = 0; v < 64; v++) s += tA[v * kPitchA + lane];
                ab3 += s;
            }
            __syncthreads();
            // d1 = (W2^T d2) * lrelu'(a1);  tiles: A = d2, B = h1 -> dW2
#pragma unroll
            for (int i = 0; i < kH; i++) {
                float s = 0.0f;
#pragma unroll
                for (int j = 0; j < kH; j++) s = __builtin_fmaf(sW2t[i * kH + j], d2[j], s);
                d1[i] = h1[i] > 0.0f ? s : kSlope * s;
                tA[lane * kPitchA + i] = d2[i];
                tB[lane * kPitch + i] = h1[i];
                __builtin_amdgcn_sched_barrier(0);       // as in mlp_row: do not hoist the whole layer's weight reads
            }
            __syncthreads();
            outer_mfma<TH, TH>(tA, kPitchA, kH, tB, kPitch, kH, TH, lane, aW2);
            if (lane < kH) {
                float s = 0.0f;
                for (uint32_t v = 0; v < 64; v++) s += tA[v * kPitchA + lane];
                ab2 += s;
            }
            __syncthreads();
        }
        // input gradient, and tiles A = d1 [64][H1], B = inputs [64][C] -> dW1
#pragma unroll
        for (int j = 0; j < H1; j++) tA[lane * kPitchA + j] = d1[j];
        float   s_pg = 0.0f;       // this row's gradient of the Pg column
        auto d_in = [&](uint32_t c) {      // d input[c] = sum_j W1[j][c] d1[j]
            float        s = 0.0f;
            const float* w = sW1t + c * H1;
#pragma unroll
            for (int j = 0; j < H1; j++) s = __builtin_fmaf(w[j], d1[j], s);
            return s;
        };
        if (on) {
            uint32_t c = 0;
            auto     seg = [&](const float* base, uint32_t ld, uint32_t n, float* gout, uint32_t ldg) {
                const float* r = base + (size_t)row * ld;
                float*       go = gout ? gout + (size_t)row * ldg : nullptr;
                if (seg_vec_ok(base, ld, n) && (!gout || seg_vec_ok(gout, ldg, n))) {
                    for (uint32_t k = 0; k < n; k += 4, c += 4) {
                        const float4 v = *reinterpret_cast<const float4*>(r + k);
                        tB[lane * kPitch + c] = v.x; tB[lane * kPitch + c + 1] = v.y;
                        tB[lane * kPitch + c + 2] = v.z; tB[lane * kPitch + c + 3] = v.w;
                        if (go) *reinterpret_cast<float4*>(go + k) = make_float4(d_in(c), d_in(c + 1), d_in(c + 2), d_in(c + 3));
                    }
                } else {
                    for (uint32_t k = 0; k < n; k++, c++) {
                        tB[lane * kPitch + c] = r[k];
                        if (go) go[k] = d_in(c);
                    }
                }
            };
            seg(a.in_a, a.lda, a.Ca, g.g_a, g.ldga);
            if (a.Cb) seg(a.in_b, a.ldb, a.Cb, g.g_b, g.ldgb);
            if (a.pg) {
                tB[lane * kPitch + c] = a.pg[a.pg_index ? a.pg_index[row] : 0];
                const float s = d_in(c);
                if (a.pg_index) s_pg = s;
                else apg += s;
            }
        } else {
            for (uint32_t c = 0; c < a.C; c++) tB[lane * kPitch + c] = 0.0f;
        }
        if (g.g_pg && a.pg_index) {
            // rows of one level are contiguous, so a wave almost always holds ONE table entry: reduce over the
            // wave and keep a running sum per wave, flushed with one atomic when the entry changes.  (One atomic
            // per lane on <= 16 addresses serialised the whole kernel: 9.6 ms instead of 0.7.)
            const int64_t  idx = on ? a.pg_index[row] : -1;
            const uint64_t live = __ballot(on);
            if (live) {
                const int64_t first = __shfl(idx, __builtin_ctzll(live));
                if (__ballot(on && idx != first) == 0) {
                    float v = on ? s_pg : 0.0f;
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
                    if (first != pg_at) {
                        if (pg_at >= 0 && lane == 0) atomicAdd(g.g_pg + pg_at, apg);
                        pg_at = first;
                        apg = 0.0f;
                    }
                    apg += v;
                } else if (on) {
                    atomicAdd(g.g_pg + idx, s_pg);       // a wave straddling two levels
                }
            }
        }
        __syncthreads();
        outer_mfma<T1A, T1B>(tA, kPitchA, H1, tB, kPitch, a.C, ntb1, lane, aW1);
        if (lane < H1) {
            float s = 0.0f;
            for (uint32_t v = 0; v < 64; v++) s += tA[v * kPitchA + lane];
            ab1 += s;
        }
        __syncthreads();
    }
    // ~1000 blocks adding into the same few hundred addresses serialise at the memory side (21 us of a 72 us
    // single-Linear call): the caller hands n_rep zeroed copies of the weight-gradient buffer and sums them
    const size_t rep = (size_t)(blockIdx.x % g.n_rep) * g.rep_stride;
    flush_mfma<T1A, T1B>(g.gW1 + rep, H1, a.C, lane, aW1);
    if (lane < H1) atomicAdd(g.gb1 + rep + lane, ab1);
    if constexpr (NL == 3) {
        flush_mfma<TH, TH>(g.gW2 + rep, kH, kH, lane, aW2);
        flush_mfma<TF, TH>(g.gW3 + rep, F, kH, lane, aW3);
        if (lane < kH) atomicAdd(g.gb2 + rep + lane, ab2);
        if (lane < F) atomicAdd(g.gb3 + rep + lane, ab3);
    }
    if (g.g_pg && a.pg_index) {
        if (pg_at >= 0 && lane == 0) atomicAdd(g.g_pg + pg_at, apg);
    } else if (g.g_pg) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) apg += __shfl_xor(apg, d);
        if (lane == 0) atomicAdd(g.g_pg, apg);
    }
}

// ---------------------------------------------------------------------------------------------
// The three-layer head (C -> 32 -> 32 -> F, LeakyReLU) on the matrix cores  (round 6)
//
// Rounds 2-5 ran these heads one LANE per vertex: a vertex's 32 + 32 hidden values in registers, every weight read from
// LDS (a broadcast read per multiply-add): 2080 LDS reads + 2080 FMAs per vertex forward, ~4200 backward, and the backward's
// 42 KB of weight-gradient tiles per two waves left six waves on a CU — 0.86 ms per training step for the 3-D context
// head's backward, 0.22 for its forward, on 0.75 M vertices whose arithmetic is 3 GFLOP (now 0.17 / 0.06 ms alone).
//
// Here a wave takes 16 vertices at a time through the layers as TRANSPOSED products on v_mfma_f32_16x16x4_f32 (fp32, an
// fmaf chain in k order): the weights are the A operand (rows = output features), the activations the B operand
// (columns = vertices).  With the instruction's four k slots standing for channels 4 q + s (q = lane / 16, s = step) the
// result — lane (vertex v = lane % 16, q) holds output features 4 q .. 4 q + 3 of a 16-feature tile — IS the next layer's B
// operand, register for register: a layer's output never leaves the registers, nothing is transposed, and the same holds
// backwards (d2 = W3^T d_out, d1 = W2^T d2, d_in = W1^T d1 with the transposed weight images).  Weights are read from LDS
// as float4 along k (one 16-byte read per four products).  Only the weight gradients — sums over VERTICES, which sit
// on the wrong axis for that — go through LDS: 16 rows per wave (5 KB), four k = 4 steps per 16 x 16 tile.
// Per 16 vertices: 40 products forward, 40 + 40 + 40 backward (recomputation, input gradients, weight gradients).
// ---------------------------------------------------------------------------------------------
constexpr int kP1 = 52;      // W1 image: 48 channels + 4 (rows 16-byte aligned, 13 sixteen-byte slots apart)
constexpr int kP2 = 36;      // 32 + 4
constexpr int kP3t = 20;     // W3^T image: 16 output features (F of them non-zero) + 4
constexpr int kTA = 33, kTB = 49;      // weight-gradient tiles of a wave: [16][32 + 1] and [16][48 + 1]

struct Head3Lds {
    float W1[kH * kP1];          // [j][c]   layer 1:   A[m = j][k = c]
    float W2[kH * kP2];          // [j][i]   layer 2
    float W3[16 * kP2];          // [f][j]   layer 3 (rows >= F zero)
    float b1[kH], b2[kH], b3[16];
};
struct Head3LdsBwd {
    float W3t[kH * kP3t];        // [j][f] = W3[f][j]     d2:   A[m = j][k = f]
    float W2t[kH * kP2];         // [i][j] = W2[j][i]     d1
    float W1t[48 * kP2];         // [c][j] = W1[j][c]     d_in
};

struct Head3LdsAll {
    Head3Lds    w;
    Head3LdsBwd wt;
};

template <int F>
__device__ __forceinline__ void head3_load(const MlpArgs& a, Head3Lds& w)
{
    for (uint32_t e = threadIdx.x; e < kH * kP1; e += blockDim.x) {
        const uint32_t j = e / kP1, c = e % kP1;
        w.W1[e] = c < a.C ? a.W1[j * a.C + c] : 0.0f;
    }
    for (uint32_t e = threadIdx.x; e < kH * kP2; e += blockDim.x) {
        const uint32_t j = e / kP2, i = e % kP2;
        w.W2[e] = i < kH ? a.W2[j * kH + i] : 0.0f;
    }
    for (uint32_t e = threadIdx.x; e < 16 * kP2; e += blockDim.x) {
        const uint32_t f = e / kP2, j = e % kP2;
        w.W3[e] = (f < F && j < kH) ? a.W3[f * kH + j] : 0.0f;
    }
    for (uint32_t e = threadIdx.x; e < kH; e += blockDim.x) { w.b1[e] = a.b1[e]; w.b2[e] = a.b2[e]; }
    for (uint32_t e = threadIdx.x; e < 16; e += blockDim.x) w.b3[e] = e < F ? a.b3[e] : 0.0f;
}

template <int F>
__device__ __forceinline__ void head3_load_bwd(const MlpArgs& a, Head3LdsBwd& w)
{
    for (uint32_t e = threadIdx.x; e < kH * kP3t; e += blockDim.x) {
        const uint32_t j = e / kP3t, f = e % kP3t;
        w.W3t[e] = f < F ? a.W3[f * kH + j] : 0.0f;
    }
    for (uint32_t e = threadIdx.x; e < kH * kP2; e += blockDim.x) {
        const uint32_t i = e / kP2, j = e % kP2;
        w.W2t[e] = j < kH ? a.W2[j * kH + i] : 0.0f;
    }
    for (uint32_t e = threadIdx.x; e < 48 * kP2; e += blockDim.x) {
        const uint32_t c = e / kP2, j = e % kP2;
        w.W1t[e] = (c < a.C && j < kH) ? a.W1[j * a.C + c] : 0.0f;
    }
}

// columns [c0, c0 + 4) of input row `row` ([in_a | in_b | pg | 0 ...]); zeros for a row that does not exist
__device__ __forceinline__ f32x4 head3_fetch4(const MlpArgs& a, uint32_t row, uint32_t c0, bool on, bool veca, bool vecb)
{
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (!on) return v;
    if (c0 + 4u <= a.Ca && veca) {
        const float4 t = *reinterpret_cast<const float4*>(a.in_a + (size_t)row * a.lda + c0);
        return f32x4{t.x, t.y, t.z, t.w};
    }
    if (c0 >= a.Ca && c0 + 4u <= a.Ca + a.Cb && vecb && ((c0 - a.Ca) & 3u) == 0u) {
        const float4 t = *reinterpret_cast<const float4*>(a.in_b + (size_t)row * a.ldb + (c0 - a.Ca));
        return f32x4{t.x, t.y, t.z, t.w};
    }
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = c0 + r < a.C ? input_at(a, row, c0 + r) : 0.0f;
    return v;
}

// A[m][4 q .. 4 q + 3] of a weight image with row pitch `pitch` (a float4: the four steps of one k block)
__device__ __forceinline__ f32x4 head3_a4(const float* img, uint32_t pitch, uint32_t m, uint32_t k0)
{
    const float4 t = *reinterpret_cast<const float4*>(img + m * pitch + k0);
    return f32x4{t.x, t.y, t.z, t.w};
}

__device__ __forceinline__ f32x4 mfma4(const f32x4& a, const f32x4& b, f32x4 acc)
{
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
    return acc;
}

// forward of 16 vertices: x[b] = columns 16 b + 4 q .. of this lane's vertex; h1 / h2 = post-activation hidden values
// (features 16 t + 4 q + r), o = outputs 4 q + r
__device__ __forceinline__ void head3_forward(const Head3Lds& w, const f32x4 (&x)[3], uint32_t nkb, uint32_t i16, uint32_t q,
                                              f32x4 (&h1)[2], f32x4 (&h2)[2], f32x4& o)
{
    f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < 2; t++) {
        f32x4 acc = z;
#pragma unroll
        for (int b = 0; b < 3; b++)
            if ((uint32_t)b < nkb) acc = mfma4(head3_a4(w.W1, kP1, 16u * t + i16, 16u * b + 4u * q), x[b], acc);
        const float4 bb = *reinterpret_cast<const float4*>(w.b1 + 16 * t + 4 * q);
        h1[t] = f32x4{lrelu(acc[0] + bb.x), lrelu(acc[1] + bb.y), lrelu(acc[2] + bb.z), lrelu(acc[3] + bb.w)};
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
        f32x4 acc = z;
#pragma unroll
        for (int b = 0; b < 2; b++) acc = mfma4(head3_a4(w.W2, kP2, 16u * t + i16, 16u * b + 4u * q), h1[b], acc);
        const float4 bb = *reinterpret_cast<const float4*>(w.b2 + 16 * t + 4 * q);
        h2[t] = f32x4{lrelu(acc[0] + bb.x), lrelu(acc[1] + bb.y), lrelu(acc[2] + bb.z), lrelu(acc[3] + bb.w)};
    }
    f32x4 acc = z;
#pragma unroll
    for (int b = 0; b < 2; b++) acc = mfma4(head3_a4(w.W3, kP2, i16, 16u * b + 4u * q), h2[b], acc);
    const float4 bb = *reinterpret_cast<const float4*>(w.b3 + 4 * q);
    o = f32x4{acc[0] + bb.x, acc[1] + bb.y, acc[2] + bb.z, acc[3] + bb.w};
}

template <int F>
__global__ __launch_bounds__(256) void k_ctx_head3_fwd(MlpArgs a, float* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) Head3Lds w;
    head3_load<F>(a, w);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, i16 = lane & 15u, q = lane >> 4;
    const uint32_t tiles = (a.N + 15u) / 16u, nkb = (a.C + 15u) / 16u;
    const bool     veca = seg_vec_ok(a.in_a, a.lda, 4), vecb = a.Cb != 0 && seg_vec_ok(a.in_b, a.ldb, 4);
    // a CONTIGUOUS range of tiles per wave
    const uint32_t per_wave = div_up(tiles, gridDim.x * 4u), t_lo = (blockIdx.x * 4u + wave) * per_wave;
    for (uint32_t tile = t_lo; tile < min(tiles, t_lo + per_wave); tile++) {
        const uint32_t row = tile * 16u + i16;
        const bool     on = row < a.N;
        f32x4 x[3], h1[2], h2[2], o;
#pragma unroll
        for (int b = 0; b < 3; b++) x[b] = head3_fetch4(a, row, 16u * b + 4u * q, on && (uint32_t)b < nkb, veca, vecb);
        head3_forward(w, x, nkb, i16, q, h1, h2, o);
        if (on) {
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (4u * q + r < (uint32_t)F) out[(size_t)row * F + 4u * q + r] = o[r];
        }
    }
}

// One wave's LDS writes followed by its own reads: DS operations of a wave execute in order, only the compiler has to be
// kept from moving them.
__device__ __forceinline__ void head3_lds_order()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// G[i][j] += sum over the 16 vertices of P[v][i] Q[v][j]: tiles of the wave's [16][pitch] LDS images, four k = 4 steps
template <int TA, int TB>
__device__ __forceinline__ void head3_outer(const float* tP, uint32_t pitchP, const float* tQ, uint32_t pitchQ, uint32_t ntb,
                                            uint32_t i16, uint32_t q, f32x4 (&acc)[TA][TB])
{
#pragma unroll
    for (uint32_t s = 0; s < 4; s++) {
        const uint32_t v = 4u * s + q;
        float pa[TA], qb[TB];
#pragma unroll
        for (int ta = 0; ta < TA; ta++) pa[ta] = tP[v * pitchP + 16u * ta + i16];
#pragma unroll
        for (int tb = 0; tb < TB; tb++) qb[tb] = (uint32_t)tb < ntb ? tQ[v * pitchQ + 16u * tb + i16] : 0.0f;
#pragma unroll
        for (int ta = 0; ta < TA; ta++)
#pragma unroll
            for (int tb = 0; tb < TB; tb++)
                if ((uint32_t)tb < ntb) acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[ta], qb[tb], acc[ta][tb], 0, 0, 0);
    }
}

__device__ __forceinline__ void head3_put(float* tile, uint32_t pitch, uint32_t v, uint32_t c0, const f32x4& x)
{
#pragma unroll
    for (int r = 0; r < 4; r++) tile[v * pitch + c0 + r] = x[r];
}

template <int F>
__global__ __launch_bounds__(256, 3) void k_ctx_head3_bwd(MlpArgs a, MlpGrads g)
{
    __shared__ __attribute__((aligned(16))) Head3LdsAll all;
    Head3Lds&    w = all.w;
    Head3LdsBwd& wt = all.wt;
    __shared__ float tilesA[4][16 * kTA], tilesB[4][16 * kTB];
    head3_load<F>(a, w);
    head3_load_bwd<F>(a, wt);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, i16 = lane & 15u, q = lane >> 4;
    float* const tA = tilesA[wave];
    float* const tB = tilesB[wave];
    const uint32_t tiles = (a.N + 15u) / 16u, nkb = (a.C + 15u) / 16u;
    const bool     veca = seg_vec_ok(a.in_a, a.lda, 4), vecb = a.Cb != 0 && seg_vec_ok(a.in_b, a.ldb, 4);
    const bool     vecga = seg_vec_ok(g.g_a, g.ldga, 4), vecgb = g.g_b != nullptr && seg_vec_ok(g.g_b, g.ldgb, 4);
    const uint32_t c_pg = a.Ca + a.Cb;                    // the Pg column (when there is one)
    f32x4 aW1[2][3] = {}, aW2[2][2] = {}, aW3[1][2] = {};
    f32x4 ab1[2] = {}, ab2[2] = {}, ab3 = {0.0f, 0.0f, 0.0f, 0.0f};
    float   apg = 0.0f;
    int64_t pg_at = -1;            // pg_index mode: the table entry this WAVE is accumulating for (wave-uniform)
    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};

    // A CONTIGUOUS range of tiles per wave: the vertices of one level are contiguous, so a wave's running sum of the Pg
    // column's gradient changes its table entry a handful of times per call (tiles dealt out round-robin changed it at every
    // tile: 47 k atomics onto 12 addresses, serialised at the memory side — 0.44 ms of a 0.67 ms call)
    const uint32_t per_wave = div_up(tiles, gridDim.x * 4u), t_lo = (blockIdx.x * 4u + wave) * per_wave;
    for (uint32_t tile = t_lo; tile < min(tiles, t_lo + per_wave); tile++) {
        const uint32_t row = tile * 16u + i16;
        const bool     on = row < a.N;
        f32x4 x[3], h1[2], h2[2], o;
#pragma unroll
        for (int b = 0; b < 3; b++) x[b] = head3_fetch4(a, row, 16u * b + 4u * q, on && (uint32_t)b < nkb, veca, vecb);
        head3_forward(w, x, nkb, i16, q, h1, h2, o);
        // d_out of this lane's vertex: features 4 q + r (zero beyond F, and for a row that does not exist: every gradient
        // below is linear in it)
        f32x4 d_o = z;
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (on && 4u * q + r < (uint32_t)F) d_o[r] = g.g_out[(size_t)row * F + 4u * q + r];
        // d2 = (W3^T d_out) lrelu'(a2), d1 = (W2^T d2) lrelu'(a1)   (lrelu keeps the sign: h > 0 <=> a > 0)
        f32x4 d2[2], d1[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const f32x4 acc = mfma4(head3_a4(wt.W3t, kP3t, 16u * t + i16, 4u * q), d_o, z);
#pragma unroll
            for (int r = 0; r < 4; r++) d2[t][r] = h2[t][r] > 0.0f ? acc[r] : kSlope * acc[r];
        }
#pragma unroll
        for (int t = 0; t < 2; t++) {
            f32x4 acc = z;
#pragma unroll
            for (int b = 0; b < 2; b++) acc = mfma4(head3_a4(wt.W2t, kP2, 16u * t + i16, 16u * b + 4u * q), d2[b], acc);
#pragma unroll
            for (int r = 0; r < 4; r++) d1[t][r] = h1[t][r] > 0.0f ? acc[r] : kSlope * acc[r];
        }
        // input gradient: columns 16 t + 4 q + r of this lane's vertex
        float s_pg = 0.0f;
#pragma unroll
        for (int t = 0; t < 3; t++) {
            if ((uint32_t)t >= nkb) continue;
            f32x4 acc = z;
#pragma unroll
            for (int b = 0; b < 2; b++) acc = mfma4(head3_a4(wt.W1t, kP2, 16u * t + i16, 16u * b + 4u * q), d1[b], acc);
            const uint32_t c0 = 16u * t + 4u * q;
            if (on) {
                if (c0 + 4u <= a.Ca && vecga) {
                    *reinterpret_cast<float4*>(g.g_a + (size_t)row * g.ldga + c0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                } else if (c0 >= a.Ca && c0 + 4u <= a.Ca + a.Cb && vecgb && ((c0 - a.Ca) & 3u) == 0u) {
                    *reinterpret_cast<float4*>(g.g_b + (size_t)row * g.ldgb + (c0 - a.Ca)) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const uint32_t c = c0 + r;
                        if (c < a.Ca) g.g_a[(size_t)row * g.ldga + c] = acc[r];
                        else if (c < a.Ca + a.Cb) { if (g.g_b) g.g_b[(size_t)row * g.ldgb + (c - a.Ca)] = acc[r]; }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (a.pg && c0 + r == c_pg) s_pg = acc[r];
            }
        }
        // bias gradients: per-lane partial sums (reduced over the 16 vertices of a lane row at the end)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            ab3[r] += d_o[r];
#pragma unroll
            for (int t = 0; t < 2; t++) { ab2[t][r] += d2[t][r]; ab1[t][r] += d1[t][r]; }
        }
        // the Pg column's gradient
        if (g.g_pg && a.pg) {
            const bool mine = on && 4u * q <= c_pg % 16u && c_pg % 16u < 4u * q + 4u;      // the lane row that holds column c_pg
            if (!a.pg_index) {
                apg += mine ? s_pg : 0.0f;
            } else {
                // rows of one level are contiguous, so a tile almost always holds ONE table entry: reduce over the wave and
                // keep a running sum per wave, flushed with one atomic when the entry changes
                const int64_t  idx = mine ? a.pg_index[row] : -1;
                const uint64_t live = __ballot(mine);
                if (live) {
                    const int64_t first = __shfl(idx, __builtin_ctzll(live));
                    if (__ballot(mine && idx != first) == 0) {
                        float v = mine ? s_pg : 0.0f;
#pragma unroll
                        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
                        if (first != pg_at) {
                            if (pg_at >= 0 && lane == 0) atomicAdd(g.g_pg + pg_at, apg);
                            pg_at = first;
                            apg = 0.0f;
                        }
                        apg += v;
                    } else if (mine) {
                        atomicAdd(g.g_pg + idx, s_pg);       // a tile straddling two levels
                    }
                }
            }
        }
        // weight gradients: the three outer products through the wave's LDS tiles
        head3_put(tA, kTA, i16, 4u * q, d_o);                                           // d_out [16][16]
#pragma unroll
        for (int t = 0; t < 2; t++) head3_put(tB, kTB, i16, 16u * t + 4u * q, h2[t]);   // h2 [16][32]
        head3_lds_order();
        head3_outer<1, 2>(tA, kTA, tB, kTB, 2, i16, q, aW3);
        head3_lds_order();
#pragma unroll
        for (int t = 0; t < 2; t++) {
            head3_put(tA, kTA, i16, 16u * t + 4u * q, d2[t]);
            head3_put(tB, kTB, i16, 16u * t + 4u * q, h1[t]);
        }
        head3_lds_order();
        head3_outer<2, 2>(tA, kTA, tB, kTB, 2, i16, q, aW2);
        head3_lds_order();
#pragma unroll
        for (int t = 0; t < 2; t++) head3_put(tA, kTA, i16, 16u * t + 4u * q, d1[t]);
#pragma unroll
        for (int b = 0; b < 3; b++)
            if ((uint32_t)b < nkb) head3_put(tB, kTB, i16, 16u * b + 4u * q, x[b]);
        head3_lds_order();
        head3_outer<2, 3>(tA, kTA, tB, kTB, nkb, i16, q, aW1);
        head3_lds_order();
    }
    // The block's four waves' weight-gradient tiles are summed through LDS (the weight images are dead by now: 28 KB, two
    // 12 KB regions) before ONE wave adds them to the caller's buffer: a quarter of the atomics (4096 waves flushing 3072
    // values each onto ~2 k addresses took 60 us of a 240 us call).
    {
        float* const scratch = reinterpret_cast<float*>(&all);
        static_assert(sizeof(Head3LdsAll) >= 2 * 12 * 256 * sizeof(float), "two waves' tiles fit the weight images");
        auto put_all = [&](float* dst) {
            uint32_t k = 0;
#pragma unroll
            for (int ta = 0; ta < 2; ta++)
#pragma unroll
                for (int tb = 0; tb < 3; tb++, k++)
#pragma unroll
                    for (int r = 0; r < 4; r++) dst[(k * 4 + r) * 64 + lane] = aW1[ta][tb][r];
#pragma unroll
            for (int ta = 0; ta < 2; ta++)
#pragma unroll
                for (int tb = 0; tb < 2; tb++, k++)
#pragma unroll
                    for (int r = 0; r < 4; r++) dst[(k * 4 + r) * 64 + lane] = aW2[ta][tb][r];
#pragma unroll
            for (int tb = 0; tb < 2; tb++, k++)
#pragma unroll
                for (int r = 0; r < 4; r++) dst[(k * 4 + r) * 64 + lane] = aW3[0][tb][r];
        };
        auto add_all = [&](const float* src) {
            uint32_t k = 0;
#pragma unroll
            for (int ta = 0; ta < 2; ta++)
#pragma unroll
                for (int tb = 0; tb < 3; tb++, k++)
#pragma unroll
                    for (int r = 0; r < 4; r++) aW1[ta][tb][r] += src[(k * 4 + r) * 64 + lane];
#pragma unroll
            for (int ta = 0; ta < 2; ta++)
#pragma unroll
                for (int tb = 0; tb < 2; tb++, k++)
#pragma unroll
                    for (int r = 0; r < 4; r++) aW2[ta][tb][r] += src[(k * 4 + r) * 64 + lane];
#pragma unroll
            for (int tb = 0; tb < 2; tb++, k++)
#pragma unroll
                for (int r = 0; r < 4; r++) aW3[0][tb][r] += src[(k * 4 + r) * 64 + lane];
        };
        __syncthreads();                                  // every wave is done with the weight images
        if (wave & 1u) put_all(scratch + (wave >> 1) * 3072);
        __syncthreads();
        if (!(wave & 1u)) add_all(scratch + (wave >> 1) * 3072);
        __syncthreads();
        if (wave == 2) put_all(scratch);
        __syncthreads();
        if (wave == 0) add_all(scratch);
    }
    const size_t rep = (size_t)(blockIdx.x % g.n_rep) * g.rep_stride;
    if (wave == 0) {
        flush_mfma<2, 3>(g.gW1 + rep, kH, a.C, lane, aW1);
        flush_mfma<2, 2>(g.gW2 + rep, kH, kH, lane, aW2);
        flush_mfma<1, 2>(g.gW3 + rep, F, kH, lane, aW3);
    }
    // bias sums over the 16 vertices of a lane row (lanes of equal q)
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) {
            ab3[r] += __shfl_xor(ab3[r], d);
#pragma unroll
            for (int t = 0; t < 2; t++) { ab2[t][r] += __shfl_xor(ab2[t][r], d); ab1[t][r] += __shfl_xor(ab1[t][r], d); }
        }
        if (i16 == 0) {
            if (4u * q + r < (uint32_t)F) atomicAdd(g.gb3 + rep + 4u * q + r, ab3[r]);
#pragma unroll
            for (int t = 0; t < 2; t++) {
                atomicAdd(g.gb2 + rep + 16u * t + 4u * q + r, ab2[t][r]);
                atomicAdd(g.gb1 + rep + 16u * t + 4u * q + r, ab1[t][r]);
            }
        }
    }
    if (g.g_pg && a.pg && a.pg_index) {
        if (pg_at >= 0 && lane == 0) atomicAdd(g.g_pg + pg_at, apg);
    } else if (g.g_pg && a.pg) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) apg += __shfl_xor(apg, d);
        if (lane == 0) atomicAdd(g.g_pg, apg);
    }
}

// ---------------------------------------------------------------------------------------------
// Bernoulli rate (utils_bpp_acc.py:1002-1013) with the table gather in front of it
// ---------------------------------------------------------------------------------------------
constexpr float kPmin = 1e-6f, kPmax = 1.0f - 1e-6f, kInvLn2 = 1.4426950408889634f;

__global__ __launch_bounds__(256) void k_bernoulli_bits(const float* __restrict__ table, const int64_t* __restrict__ rows,
                                                        const float* __restrict__ mean, uint64_t n, uint32_t F,
                                                        float* __restrict__ partial)
{
    float s = 0.0f;
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (uint64_t)gridDim.x * 256) {
        const uint64_t slot = e / F;
        const float    x = rows ? table[(size_t)rows[slot] * F + e % F] : table[e];
        const float    p = fminf(fmaxf(mean[e], kPmin), kPmax);
        s += -log2f(p) * ((1.0f + x) / 2.0f) + -log2f(1.0f - p) * ((1.0f - x) / 2.0f);
    }
    __shared__ float red[4];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void k_bernoulli_bits_bwd(const float* __restrict__ table, const int64_t* __restrict__ rows,
                                                            const float* __restrict__ mean, const float* __restrict__ g,
                                                            uint64_t n, uint32_t F, float* __restrict__ g_mean,
                                                            float* __restrict__ g_x)
{
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const float gs = g[0];
    const uint64_t slot = e / F;
    const float    x = rows ? table[(size_t)rows[slot] * F + e % F] : table[e];
    const float    m = mean[e];
    const float    p = fminf(fmaxf(m, kPmin), kPmax);
    const float    pos = (1.0f + x) / 2.0f, neg = (1.0f - x) / 2.0f;
    // clamp passes the gradient where min <= mean <= max
    const bool inside = m >= kPmin && m <= kPmax;
    if (g_mean) g_mean[e] = inside ? gs * (-pos / p + neg / (1.0f - p)) * kInvLn2 : 0.0f;
    if (g_x) g_x[e] = gs * 0.5f * (-log2f(p) + log2f(1.0f - p));
}

// d/d values of cnc_segment_weighted_sum: row t of slot s gets g[s] * scale_t, scale = w_t (mode 0),
// w_t / sum_slot(w) (mode 1), 1 / count (mode 2).  One lane per (slot, feature) walking the slot's rows, like the
// forward (a lane per (row, feature) had to find its slot by bisection: 17 dependent loads per element at 150 k
// slots — 0.33 ms per training step against 0.11 for the forward).
__global__ __launch_bounds__(256) void k_segment_bwd(const float* __restrict__ g, const int64_t* __restrict__ cumsum,
                                                     const float* __restrict__ weights, const float* __restrict__ wsum,
                                                     uint32_t n_slots, uint64_t T, uint32_t F, int mode,
                                                     float* __restrict__ g_values, const int64_t* __restrict__ order)
{
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (uint64_t)n_slots * F) return;
    const uint32_t s = (uint32_t)(e / F), f = (uint32_t)(e % F);
    const int64_t  t0 = cumsum[s], t1 = min(cumsum[s + 1], (int64_t)T);
    if (t1 <= t0) return;
    const float gs = g[e];
    float       per_slot = 1.0f;
    if (mode == 1) per_slot = wsum[s];
    else if (mode == 2) per_slot = (float)(t1 - t0);
    for (int64_t t = t0; t < t1; t++) {
        float scale = weights ? weights[t] : 1.0f;
        if (mode != 0) scale = scale / per_slot;
        // with a row permutation the reduction read values[order[t]]: that is where the gradient goes (one writer each)
        g_values[(uint64_t)(order ? order[t] : t) * F + f] = gs * scale;
    }
}

template <int NL>
static int launch_mlp(bool backward, uint32_t F, const MlpArgs& a, float* out, const MlpGrads& g, hipStream_t s)
{
    // The three-layer head: 16 vertices per wave on the matrix cores, four waves per workgroup, a contiguous range of
    // tiles per wave, three workgroups per CU resident.  The single-Linear heads stay on the lane-per-vertex kernels: built
    // on the same machinery (8 + 24 products per 16 vertices) they came out SLOWER alone — 30 / 115 us against 24 / 90 for
    // 0.64 M rows: those calls move rows and little else, and a lane per row moves them in fewer instructions.
    const uint32_t wgs = min(div_up(div_up(a.N, 16u), 4u), 768u);
    const uint32_t blocks = min(div_up(a.N, 256), 2048u);
    const uint32_t bwd_blocks = min(div_up(a.N, (uint32_t)kBwdThreads), 1024u);
#define CNC_CTX_CASE(FF)                                                                                              \
    if (F == FF) {                                                                                                    \
        if constexpr (NL == 3) {                                                                                      \
            if (backward) hipLaunchKernelGGL((k_ctx_head3_bwd<FF>), dim3(wgs), dim3(256), 0, s, a, g);                \
            else hipLaunchKernelGGL((k_ctx_head3_fwd<FF>), dim3(wgs), dim3(256), 0, s, a, out);                       \
        } else {                                                                                                      \
            if (backward) hipLaunchKernelGGL((k_ctx_mlp_bwd<1, FF>), dim3(bwd_blocks), dim3(kBwdThreads), 0, s, a, g); \
            else hipLaunchKernelGGL((k_ctx_mlp_fwd<1, FF>), dim3(blocks), dim3(256), 0, s, a, out);                   \
        }                                                                                                             \
        return launch_status();                                                                                       \
    }
    CNC_CTX_CASE(1) CNC_CTX_CASE(2) CNC_CTX_CASE(4) CNC_CTX_CASE(8)
#undef CNC_CTX_CASE
    return CNC_ERR_UNSUPPORTED;
}

static bool mlp_args_ok(const MlpArgs& a, uint32_t n_layers)
{
    if (!a.in_a || !a.W1 || !a.b1 || a.C == 0 || a.C > (uint32_t)kMaxC || a.lda < a.Ca) return false;
    if (a.Cb && (!a.in_b || a.ldb < a.Cb)) return false;
    if (n_layers == 3 && (!a.W2 || !a.b2 || !a.W3 || !a.b3)) return false;
    return n_layers == 1 || n_layers == 3;
}

}  // namespace cnc

using namespace cnc;

extern "C" int cnc_ctx_mlp_forward(const float* in_a, uint32_t lda, uint32_t Ca, const float* in_b, uint32_t ldb,
                                   uint32_t Cb, const float* pg, const int64_t* pg_index, uint32_t N,
                                   uint32_t n_layers, uint32_t F,
                                   const float* W1, const float* b1, const float* W2, const float* b2,
                                   const float* W3, const float* b3, float* out, void* stream)
{
    if (N == 0) return CNC_OK;
    MlpArgs a{in_a, lda, Ca, in_b, ldb, in_b ? Cb : 0u, pg, pg ? pg_index : nullptr, N,
              Ca + (in_b ? Cb : 0u) + (pg ? 1u : 0u), W1, b1, W2, b2, W3, b3};
    if (!out || !mlp_args_ok(a, n_layers)) return CNC_ERR_INVALID_VALUE;
    MlpGrads none{};
    return n_layers == 1 ? launch_mlp<1>(false, F, a, out, none, (hipStream_t)stream)
                         : launch_mlp<3>(false, F, a, out, none, (hipStream_t)stream);
}

extern "C" int cnc_ctx_mlp_backward(const float* in_a, uint32_t lda, uint32_t Ca, const float* in_b, uint32_t ldb,
                                    uint32_t Cb, const float* pg, const int64_t* pg_index, uint32_t N,
                                    uint32_t n_layers, uint32_t F,
                                    const float* W1, const float* b1, const float* W2, const float* b2,
                                    const float* W3, const float* b3, const float* grad_out, float* grad_a,
                                    float* grad_b, float* grad_pg, float* gW1, float* gb1, float* gW2, float* gb2,
                                    float* gW3, float* gb3, uint32_t n_replicas, uint32_t replica_stride,
                                    uint32_t ldga, uint32_t ldgb, void* stream)
{
    if (N == 0) return CNC_OK;
    if (n_replicas == 0) n_replicas = 1;
    if (ldga == 0) ldga = Ca;
    if (ldgb == 0) ldgb = Cb;
    if (ldga < Ca || (in_b && grad_b && ldgb < Cb)) return CNC_ERR_INVALID_VALUE;
    MlpArgs a{in_a, lda, Ca, in_b, ldb, in_b ? Cb : 0u, pg, pg ? pg_index : nullptr, N,
              Ca + (in_b ? Cb : 0u) + (pg ? 1u : 0u), W1, b1, W2, b2, W3, b3};
    if (!grad_out || !grad_a || !gW1 || !gb1 || !mlp_args_ok(a, n_layers)) return CNC_ERR_INVALID_VALUE;
    if (n_layers == 3 && (!gW2 || !gb2 || !gW3 || !gb3)) return CNC_ERR_INVALID_VALUE;
    MlpGrads g{grad_out, grad_a, grad_b, pg ? grad_pg : nullptr, ldga, ldgb, gW1, gb1, gW2, gb2, gW3, gb3, n_replicas, replica_stride};
    return n_layers == 1 ? launch_mlp<1>(true, F, a, nullptr, g, (hipStream_t)stream)
                         : launch_mlp<3>(true, F, a, nullptr, g, (hipStream_t)stream);
}

extern "C" uint32_t cnc_bernoulli_bits_partials(uint64_t n_slots, uint32_t F)
{
    const uint64_t blocks = (n_slots * F + 255) / 256;
    return (uint32_t)(blocks < 1024 ? (blocks ? blocks : 1) : 1024);
}

extern "C" int cnc_bernoulli_bits_forward(const float* table, const int64_t* rows, const float* mean, uint64_t n_slots,
                                          uint32_t F, float* partial, void* stream)
{
    if (!partial) return CNC_ERR_INVALID_VALUE;
    if (n_slots && (!table || !mean)) return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_bernoulli_bits, dim3(cnc_bernoulli_bits_partials(n_slots, F)), dim3(256), 0, (hipStream_t)stream,
                       table, rows, mean, n_slots * F, F, partial);
    return launch_status();
}

extern "C" int cnc_bernoulli_bits_backward(const float* table, const int64_t* rows, const float* mean,
                                           const float* grad_bits, uint64_t n_slots, uint32_t F, float* grad_mean,
                                           float* grad_x, void* stream)
{
    if (n_slots == 0) return CNC_OK;
    if (!table || !mean || !grad_bits) return CNC_ERR_INVALID_VALUE;
    const uint64_t n = n_slots * F;
    hipLaunchKernelGGL(k_bernoulli_bits_bwd, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table,
                       rows, mean, grad_bits, n, F, grad_mean, grad_x);
    return launch_status();
}

extern "C" int cnc_segment_weighted_sum_gathered_backward(const float* grad, const int64_t* order,
                                                          const int64_t* cumsum, const float* weights,
                                                          const float* wsum, uint32_t n_slots, uint64_t T, uint32_t F,
                                                          int32_t mode, float* grad_values, void* stream)
{
    if (T == 0 || n_slots == 0) return CNC_OK;
    if (!grad || !cumsum || !grad_values || (mode == 1 && (!weights || !wsum))) return CNC_ERR_INVALID_VALUE;
    // rows behind the last slot (none for a cumsum that ends at T) get no gradient: the buffer is the caller's
    hipLaunchKernelGGL(k_segment_bwd, dim3((uint32_t)(((uint64_t)n_slots * F + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad,
                       cumsum, weights, wsum, n_slots, T, F, mode, grad_values, order);
    return launch_status();
}

extern "C" int cnc_segment_weighted_sum_backward(const float* grad, const int64_t* cumsum, const float* weights,
                                                 const float* wsum, uint32_t n_slots, uint64_t T, uint32_t F,
                                                 int32_t mode, float* grad_values, void* stream)
{
    return cnc_segment_weighted_sum_gathered_backward(grad, nullptr, cumsum, weights, wsum, n_slots, T, F, mode,
                                                      grad_values, stream);
}

// ---------------------------------------------------------------------------------------------
// Level statistics of a binarised table (get_BiRF_wentropy_leveln, utils_bpp_acc.py:472-486) for ALL levels
// of a table in one pass: sums[l] = sum of the entries of level l (float64 accumulation: exact for +-1
// entries, so the total does not depend on the order of the atomics), then Pg and the zero-order bit count.
// ---------------------------------------------------------------------------------------------
namespace cnc {

constexpr int kMaxLevels = 32;

struct LevelOffsets {
    int64_t off[kMaxLevels + 1];      // row offsets, off[n_levels] = end
    int32_t n_levels;
};

__device__ __forceinline__ int level_of(const LevelOffsets& lo, int64_t row)
{
    int l = 0;
    while (l + 1 < lo.n_levels && row >= lo.off[l + 1]) l++;
    return l;
}

__global__ __launch_bounds__(256) void k_level_sums(const float* __restrict__ table, LevelOffsets lo, uint32_t F,
                                                    double* __restrict__ sums)
{
    __shared__ double s_acc[kMaxLevels];
    if (threadIdx.x < kMaxLevels) s_acc[threadIdx.x] = 0.0;
    __syncthreads();
    const int64_t r0 = lo.off[0], r1 = lo.off[lo.n_levels];
    // a block covers 256 x 8 consecutive rows; a lane sums rows r, r + 256, ...
    const int64_t base = r0 + (int64_t)blockIdx.x * 2048;
    int    cur = -1;
    double acc = 0.0;
    for (int k = 0; k < 8; k++) {
        const int64_t r = base + k * 256 + threadIdx.x;
        if (r >= r1) break;
        const int l = level_of(lo, r);
        if (l != cur) {
            if (cur >= 0) atomicAdd(&s_acc[cur], acc);
            cur = l;
            acc = 0.0;
        }
        float s = 0.0f;
        if ((F & 3u) == 0 && ((uintptr_t)table & 15u) == 0) {      // same order of additions, 16-byte loads
            const float4* p = reinterpret_cast<const float4*>(table + (size_t)r * F);
            for (uint32_t q = 0; q < F / 4; q++) {
                const float4 v = p[q];
                s += v.x; s += v.y; s += v.z; s += v.w;
            }
        } else {
            for (uint32_t f = 0; f < F; f++) s += table[(size_t)r * F + f];
        }
        acc += (double)s;
    }
    if (cur >= 0) atomicAdd(&s_acc[cur], acc);
    __syncthreads();
    if (threadIdx.x < lo.n_levels && s_acc[threadIdx.x] != 0.0) atomicAdd(&sums[threadIdx.x], s_acc[threadIdx.x]);
}

// pos = (ttl + s) / 2, neg = (ttl - s) / 2, Pg = pos / ttl,
// bits = pos * -log2(max(Pg, 1e-9)) + neg * -log2(max(1 - Pg, 1e-9))       (cnc_amd.context._zero_order_bits)
__global__ void k_level_stats(const double* __restrict__ sums, LevelOffsets lo, uint32_t F, float* __restrict__ Pg,
                              float* __restrict__ bits)
{
    const int l = threadIdx.x;
    if (l >= lo.n_levels) return;
    const float ttl = (float)((lo.off[l + 1] - lo.off[l]) * (int64_t)F);
    const float s = (float)sums[l];
    const float pos = (ttl + s) / 2.0f, neg = (ttl - s) / 2.0f;
    const float p = pos / ttl;
    Pg[l] = p;
    bits[l] = pos * (-log2f(fmaxf(p, 1e-9f))) + neg * (-log2f(fmaxf(1.0f - p, 1e-9f)));
}

// grad_table[r, :] = d sums[level(r)], d sums = g_Pg dPg/ds + g_bits dbits/ds
__global__ __launch_bounds__(256) void k_level_stats_bwd(const double* __restrict__ sums, LevelOffsets lo, uint32_t F,
                                                         const float* __restrict__ g_Pg, const float* __restrict__ g_bits,
                                                         uint64_t total_rows, float* __restrict__ g_table)
{
    __shared__ float s_d[kMaxLevels];
    if ((int)threadIdx.x < lo.n_levels) {
        const int   l = threadIdx.x;
        const float ttl = (float)((lo.off[l + 1] - lo.off[l]) * (int64_t)F);
        const float s = (float)sums[l];
        const float pos = (ttl + s) / 2.0f, neg = (ttl - s) / 2.0f;
        const float p = pos / ttl, q = 1.0f - p;
        const float A = -log2f(fmaxf(p, 1e-9f)), B = -log2f(fmaxf(q, 1e-9f));
        const float dA = p > 1e-9f ? -kInvLn2 / p : 0.0f;            // dA / dPg
        const float dB = q > 1e-9f ? kInvLn2 / q : 0.0f;             // dB / dPg
        const float dp = 0.5f / ttl;                                 // dPg / ds
        const float dbits = 0.5f * A - 0.5f * B + (pos * dA + neg * dB) * dp;
        s_d[l] = (g_Pg ? g_Pg[l] * dp : 0.0f) + (g_bits ? g_bits[l] * dbits : 0.0f);
    }
    __syncthreads();
    // one 16-byte piece of a row per lane (the level is found once per piece, stores are whole float4s): the
    // one-float-per-lane form ran at 1 TB/s (164 us for the 161 MB 3-D table of a training step)
    if ((F & 3u) == 0 && ((uintptr_t)g_table & 15u) == 0) {
        const uint32_t per_row = F / 4;
        const uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x;
        if (q >= total_rows * per_row) return;
        const int64_t r = (int64_t)(q / per_row);
        const float   v = (r >= lo.off[0] && r < lo.off[lo.n_levels]) ? s_d[level_of(lo, r)] : 0.0f;
        reinterpret_cast<float4*>(g_table)[q] = make_float4(v, v, v, v);
        return;
    }
    for (uint32_t k = 0; k < 4; k++) {
        const uint64_t e = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4 + k;
        if (e >= total_rows * F) return;
        const int64_t r = (int64_t)(e / F);
        g_table[e] = (r >= lo.off[0] && r < lo.off[lo.n_levels]) ? s_d[level_of(lo, r)] : 0.0f;
    }
}

}  // namespace cnc

namespace cnc {
__global__ void k_zero_f64(double* __restrict__ p, uint32_t n)
{
    for (uint32_t i = threadIdx.x; i < n; i += 64) p[i] = 0.0;
}
}  // namespace cnc

extern "C" int cnc_level_stats_forward(const float* table, const int64_t* offsets_host, uint32_t n_levels, uint32_t F,
                                       double* sums, float* Pg, float* bits, void* stream)
{
    if (n_levels == 0 || n_levels > (uint32_t)cnc::kMaxLevels || !table || !offsets_host || !sums || !Pg || !bits)
        return CNC_ERR_INVALID_VALUE;
    cnc::LevelOffsets lo{};
    lo.n_levels = (int32_t)n_levels;
    for (uint32_t i = 0; i <= n_levels; i++) lo.off[i] = offsets_host[i];
    const int64_t rows = lo.off[n_levels] - lo.off[0];
    // (a kernel, not hipMemsetAsync: recorded into a HIP graph — cnc_amd/_planes_graph.py — the memset node of these few
    // bytes was not reliably ordered in front of the atomics that follow when other streams were busy: level sums, and
    // with them Pg, came out as garbage once in a few dozen replays)
    hipLaunchKernelGGL(cnc::k_zero_f64, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, n_levels);
    if (rows > 0)
        hipLaunchKernelGGL(cnc::k_level_sums, dim3((uint32_t)((rows + 2047) / 2048)), dim3(256), 0, (hipStream_t)stream,
                           table, lo, F, sums);
    hipLaunchKernelGGL(cnc::k_level_stats, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, lo, F, Pg, bits);
    return cnc::launch_status();
}

extern "C" int cnc_level_stats_backward(const double* sums, const int64_t* offsets_host, uint32_t n_levels, uint32_t F,
                                        const float* grad_Pg, const float* grad_bits, uint64_t total_rows,
                                        float* grad_table, void* stream)
{
    if (total_rows == 0) return CNC_OK;
    if (n_levels == 0 || n_levels > (uint32_t)cnc::kMaxLevels || !sums || !offsets_host || !grad_table)
        return CNC_ERR_INVALID_VALUE;
    cnc::LevelOffsets lo{};
    lo.n_levels = (int32_t)n_levels;
    for (uint32_t i = 0; i <= n_levels; i++) lo.off[i] = offsets_host[i];
    // a lane writes four floats (one float4 when the rows allow it)
    hipLaunchKernelGGL(cnc::k_level_stats_bwd, dim3((uint32_t)(((total_rows * F + 3) / 4 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, sums, lo, F, grad_Pg, grad_bits, total_rows, grad_table);
    return cnc::launch_status();
}


// ---------------------------------------------------------------------------------------------
// The per-step sample of the 3-D context pass (utils_bpp_acc.py:619-667): for every coded level a window of
// hash slots [v0, v1) and the vertices [p0, p1) that land in them.  One kernel writes the concatenation over
// the levels of: the vertices (int16), their positions (x - 0.5) / (R - 2), level id and resolution per
// vertex, and per slot the collision count and the absolute table row.
// ---------------------------------------------------------------------------------------------
namespace cnc {

constexpr int kMaxWin = 16;

struct WindowDesc {
    const int16_t* pos[kMaxWin];       // pos_grid_sorted_list[n] + 3 * p0
    const int64_t* cnt[kMaxWin];       // unique_count_list[n] + v0
    const int64_t* val[kMaxWin];       // unique_value_list[n] + v0
    int64_t        p_at[kMaxWin + 1];  // output offset of the level's vertices (p_at[n_win] = P)
    int64_t        v_at[kMaxWin + 1];  // output offset of the level's slots (v_at[n_win] = V)
    int64_t        row0[kMaxWin];      // table row offset of the level
    int32_t        level[kMaxWin], res[kMaxWin];
    int32_t        n_win;
};

__global__ __launch_bounds__(256) void k_ctx_window_gather(WindowDesc d, int16_t* __restrict__ pts, float* __restrict__ pts_n,
                                                           int64_t* __restrict__ lvl, int64_t* __restrict__ res,
                                                           int64_t* __restrict__ cnts, int64_t* __restrict__ rows)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t P = d.p_at[d.n_win], V = d.v_at[d.n_win];
    if (t < P) {
        int w = 0;
        while (w + 1 < d.n_win && t >= d.p_at[w + 1]) w++;
        const int64_t  k = t - d.p_at[w];
        const float    scale = (float)(d.res[w] - 2);
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const int16_t q = d.pos[w][k * 3 + a];
            pts[t * 3 + a] = q;
            pts_n[t * 3 + a] = ((float)q - 0.5f) / scale;
        }
        lvl[t] = d.level[w];
        res[t] = d.res[w];
    }
    if (t < V) {
        int w = 0;
        while (w + 1 < d.n_win && t >= d.v_at[w + 1]) w++;
        const int64_t k = t - d.v_at[w];
        cnts[t] = d.cnt[w][k];
        rows[t] = d.val[w][k] + d.row0[w];
    }
}

}  // namespace cnc

extern "C" int cnc_ctx_window_gather(const cnc_ctx_window_t* win, int16_t* pts, float* pts_n, int64_t* level_ids,
                                     int64_t* resolutions, int64_t* slot_counts, int64_t* table_rows, void* stream)
{
    if (!win || win->n_win <= 0 || win->n_win > cnc::kMaxWin) return CNC_ERR_INVALID_VALUE;
    cnc::WindowDesc d{};
    d.n_win = win->n_win;
    for (int i = 0; i < win->n_win; i++) {
        d.pos[i] = win->pos[i]; d.cnt[i] = win->cnt[i]; d.val[i] = win->val[i];
        d.p_at[i] = win->p_at[i]; d.v_at[i] = win->v_at[i]; d.row0[i] = win->row0[i];
        d.level[i] = win->level[i]; d.res[i] = win->res[i];
    }
    d.p_at[win->n_win] = win->p_at[win->n_win];
    d.v_at[win->n_win] = win->v_at[win->n_win];
    const int64_t n = d.p_at[d.n_win] > d.v_at[d.n_win] ? d.p_at[d.n_win] : d.v_at[d.n_win];
    if (n == 0) return CNC_OK;
    if (!pts || !pts_n || !level_ids || !resolutions || !slot_counts || !table_rows) return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(cnc::k_ctx_window_gather, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d,
                       pts, pts_n, level_ids, resolutions, slot_counts, table_rows);
    return cnc::launch_status();
}


// ---------------------------------------------------------------------------------------------
// Vertices of one 2-D level inside / one ring around the occupied cells of a projected occupancy plane
// (utils_bpp_acc.py:431-456): cell c covers the (T+2)^2 vertices c T + {0 .. T+1}^2 (duplicates between neighbouring
// cells kept, as the reference keeps them); per vertex its table row (examples/utils.py:492-511, 64-bit arithmetic as
// the int64 tensors there) and its position (v - 0.5) / (R - 2).  Vertex order: cell-major, then ring row, ring column.
// ---------------------------------------------------------------------------------------------
namespace cnc {

__global__ __launch_bounds__(256) void k_plane_ring_vertices(const int32_t* __restrict__ cells, uint64_t n_points, uint32_t T,
                                                             uint32_t R, uint64_t hs, int32_t* __restrict__ rows,
                                                             float* __restrict__ points)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_points) return;
    const uint32_t W = T + 2, per = W * W;
    const uint64_t c = t / per;
    const uint32_t k = (uint32_t)(t - c * per), a = k / W, b = k - a * W;
    const uint64_t x = (uint64_t)(uint32_t)cells[c * 2] * T + a, y = (uint64_t)(uint32_t)cells[c * 2 + 1] * T + b;
    const uint64_t idx = (uint64_t)R * R <= hs ? x + y * R : x ^ (y * 2654435761ull);
    rows[t] = (int32_t)(idx % hs);
    const float scale = (float)(R - 2);
    points[t * 2] = ((float)(int64_t)x - 0.5f) / scale;
    points[t * 2 + 1] = ((float)(int64_t)y - 0.5f) / scale;
}

}  // namespace cnc

extern "C" int cnc_plane_ring_vertices(const int32_t* cells, uint64_t n_cells, uint32_t T, uint32_t resolution,
                                       uint64_t hashmap_size, int32_t* rows, float* points, void* stream)
{
    if (n_cells == 0) return CNC_OK;
    if (!cells || !rows || !points || resolution < 3 || hashmap_size == 0 || hashmap_size > 0x7fffffffull || T > 1022)
        return CNC_ERR_INVALID_VALUE;
    const uint64_t n = n_cells * (uint64_t)(T + 2) * (T + 2);
    if (n > 0xffffffffull * 256) return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(cnc::k_plane_ring_vertices, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cells,
                       n, T, resolution, hashmap_size, rows, points);
    return cnc::launch_status();
}


// ---------------------------------------------------------------------------------------------
// The vertices of the 3-D context windows that lie next to occupied space, compacted: for the M indices `idx` (ascending,
// from nonzero(mask)) the normalised position, the level, the first level of the vertex's context window (level - L) and
// the clamped overlap weight — one kernel for what was index_select x 3 (two of them the library's element-wise gather:
// 0.29 + 0.05 ms on 0.75 M rows), a subtraction, a clamp and two casts (utils_bpp_acc.py:680-690).
// ---------------------------------------------------------------------------------------------
namespace cnc {
__global__ __launch_bounds__(256) void k_ctx_compact(const int64_t* __restrict__ idx, const float* __restrict__ pts_n,
                                                     const int64_t* __restrict__ level_ids, const int32_t* __restrict__ overlap,
                                                     uint32_t M, int32_t L, float* __restrict__ pts_m, int64_t* __restrict__ level_m,
                                                     int32_t* __restrict__ min_level, float* __restrict__ overlap_w)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M) return;
    const int64_t i = idx[t];
    pts_m[(size_t)t * 3 + 0] = pts_n[(size_t)i * 3 + 0];
    pts_m[(size_t)t * 3 + 1] = pts_n[(size_t)i * 3 + 1];
    pts_m[(size_t)t * 3 + 2] = pts_n[(size_t)i * 3 + 2];
    const int64_t lv = level_ids[i];
    level_m[t] = lv;
    min_level[t] = (int32_t)lv - L;
    if (overlap_w) {
        const int32_t o = overlap[i];
        overlap_w[t] = (float)(o < 1 ? 1 : o);           // torch.clamp(min=1).to(float)
    }
}
}  // namespace cnc

extern "C" int cnc_ctx_compact(const int64_t* idx, const float* pts_n, const int64_t* level_ids, const int32_t* overlap,
                               uint64_t M, int32_t L, float* pts_m, int64_t* level_m, int32_t* min_level, float* overlap_w,
                               void* stream)
{
    if (M == 0) return CNC_OK;
    if (!idx || !pts_n || !level_ids || !pts_m || !level_m || !min_level || (overlap_w && !overlap) || M >= (1ull << 32))
        return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(cnc::k_ctx_compact, dim3((uint32_t)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, pts_n,
                       level_ids, overlap, (uint32_t)M, L, pts_m, level_m, min_level, overlap_w);
    return cnc::launch_status();
}


// table[rows[s], :] = values[s, :] for S distinct rows (the table-shaped gradient of the Bernoulli rate: zeros off the coded
// rows).  The library's index_put took 0.28 ms for 1.5e5 rows of 8 floats in the training step.
namespace cnc {
__global__ __launch_bounds__(256) void k_rows_scatter(const float* __restrict__ values, const int64_t* __restrict__ rows,
                                                      float* __restrict__ table, uint64_t n, uint32_t F)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    table[(size_t)rows[i / F] * F + i % F] = values[i];
}
}  // namespace cnc

extern "C" int cnc_rows_scatter(const float* values, const int64_t* rows, float* table, uint64_t n_rows, uint32_t F, void* stream)
{
    if (n_rows == 0) return CNC_OK;
    if (!values || !rows || !table || F == 0) return CNC_ERR_INVALID_VALUE;
    const uint64_t n = n_rows * F;
    hipLaunchKernelGGL(cnc::k_rows_scatter, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, values, rows,
                       table, n, F);
    return cnc::launch_status();
}
