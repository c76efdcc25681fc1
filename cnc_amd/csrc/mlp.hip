// mlp.hip — fused fp32 MLP forward on the MFMA units (gfx950), for the gradient-free evaluations of
// the radiance field: the density pre-pass over every marched sample, the occupancy-grid update and
// test rendering (reference: nn.Sequential(Linear, ReLU, Linear[, ReLU, Linear]) on cuBLAS,
// examples/radiance_fields/ngp.py:475-504, called from :514-566).
//
//   Y = W3 relu(W2 relu(W1 x + b1) + b2) + b3      (2 or 3 layers, widths <= 160)
//
// One wave owns 16 rows end to end: it stages them in LDS, and for each layer keeps one 16x16
// accumulator per 16 output columns (<= 10 tiles = 40 VGPRs) on v_mfma_f32_16x16x4_f32 — exact fp32
// (each MFMA is a k-ordered fmaf chain), 157 TFLOP/s peak, reached from one wave per SIMD with
// several independent accumulators.  Activations never leave the CU: the C-layout accumulators go
// back to LDS (bias + ReLU applied) and are re-read in A-layout for the next layer.  Weights stream
// from L2 as float4 along K (K is visited in a permuted order inside each 16-wide block so that a
// lane's four k-values are contiguous in the row-major [out, in] weight matrix).
//
// Waves are independent (no block barriers): one 64-thread workgroup per wave, 16 x (K0p + H1p
// [+ H2p]) floats of LDS each = 27 KB for the 255->160->80 base network, so five waves share a CU.
#include "common.hpp"

namespace cnc {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kMaxTiles = 10;   // widest layer: 160 = 10 x 16
constexpr int kPad = 4;         // LDS row padding (floats) against bank conflicts

struct MlpArgs {
    const float* X;   uint32_t N, ldx, K0, K0p;      // input [N, K0], row stride ldx; K0p = K0 rounded up to 16
    const float* W[3];                               // padded weights [Hp_l, Kp_l] row-major, zero filled
    const float* B[3];                               // padded biases [Hp_l]
    uint32_t     Hp[3];                              // padded widths (multiples of 16); Hp[2] = 0 for 2 layers
    uint32_t     n_layers;
    float*       Y;   uint32_t ldy, n_out;           // output [N, n_out]
    uint32_t     a_vec;                              // X rows 16-byte aligned and >= K0p floats long
};

// One layer for this wave's 16 rows: acc[t] = A(16 x Kp) * W^T tile t.
// NT = number of 16-column output tiles, a compile-time constant: the accumulators, the weight
// fragments and their prefetch copies are then exactly NT registers-quads each (with a runtime tile
// count the compiler kept all ten alive in every layer: 300 VGPRs + 40 AGPRs, one wave per SIMD).
// Weight addresses are a wave-uniform tile base plus one 32-bit lane offset.
// A_GLOBAL: the A operand (first layer: the input rows) is read straight from global memory in MFMA
// A layout — lane (r, g) needs x[row r][kb + 4g .. 4g+3] — and prefetched one K block ahead like the
// weights, so the input never occupies LDS.  k_valid guards the tail; a_row == nullptr = row >= N.
template <bool RELU, bool A_GLOBAL, int NT>
__device__ __forceinline__ void layer_nt(const float* __restrict__ a_lds, uint32_t lda, uint32_t Kp,
                                         const float* __restrict__ W, const float* __restrict__ bias,
                                         f32x4 (&acc)[kMaxTiles], uint32_t lane,
                                         const float* __restrict__ a_row, uint32_t k_valid)
{
    const uint32_t r = lane & 15, g = lane >> 4;
    const uint32_t lane_off = r * Kp + g * 4;            // floats, inside one 16-row weight tile
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = f32x4{0, 0, 0, 0};
    auto load_w = [&](uint32_t kb, float4 (&dst)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; t++)
            dst[t] = *reinterpret_cast<const float4*>(W + (size_t)t * 16 * Kp + kb + lane_off);
    };
    // k_valid bit 31 set: rows are 16-byte aligned and at least Kp floats long (the caller checked
    // ldx), so a fragment is one float4 load and only the columns >= K0 are zeroed afterwards
    const bool     a_vec = (k_valid >> 31) != 0;
    const uint32_t k_real = k_valid & 0x7FFFFFFFu;
    auto load_a = [&](uint32_t kb) -> float4 {
        float4 v{0, 0, 0, 0};
        const uint32_t k = kb + g * 4;
        if (a_row) {
            if (a_vec) {
                v = *reinterpret_cast<const float4*>(a_row + k);
                if (k + 4 > k_real) {
                    if (k >= k_real) v.x = 0;
                    if (k + 1 >= k_real) v.y = 0;
                    if (k + 2 >= k_real) v.z = 0;
                    v.w = 0;
                }
            } else if (k + 4 <= k_real) {
                v.x = a_row[k]; v.y = a_row[k + 1]; v.z = a_row[k + 2]; v.w = a_row[k + 3];
            } else {
                if (k < k_real) v.x = a_row[k];
                if (k + 1 < k_real) v.y = a_row[k + 1];
                if (k + 2 < k_real) v.z = a_row[k + 2];
            }
        }
        return v;
    };
    // software pipeline: the fragments of K block kb+16 are requested before the 4*NT MFMAs of block kb
    float4 wn[NT], an{0, 0, 0, 0};
    load_w(0, wn);
    if constexpr (A_GLOBAL) an = load_a(0);
    for (uint32_t kb = 0; kb < Kp; kb += 16) {
        float4 a, w[NT];
        if constexpr (A_GLOBAL) a = an;
        else a = *reinterpret_cast<const float4*>(a_lds + r * lda + kb + g * 4);
#pragma unroll
        for (int t = 0; t < NT; t++) w[t] = wn[t];
        if (kb + 16 < Kp) {
            load_w(kb + 16, wn);
            if constexpr (A_GLOBAL) an = load_a(kb + 16);
        }
        // k-step outermost: consecutive MFMAs go to different accumulators
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w[t].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w[t].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w[t].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w[t].w, acc[t], 0, 0, 0);
    }
    // bias (+ ReLU): C layout is col = lane & 15, row = (lane >> 4) * 4 + i
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const float b = bias[t * 16 + r];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float v = acc[t][i] + b;
            if (RELU) v = v > 0 ? v : 0;
            acc[t][i] = v;
        }
    }
}

template <bool RELU, bool A_GLOBAL = false>
__device__ __forceinline__ void layer(const float* __restrict__ a_lds, uint32_t lda, uint32_t Kp,
                                      const float* __restrict__ W, const float* __restrict__ bias,
                                      uint32_t Hp, f32x4 (&acc)[kMaxTiles], uint32_t lane,
                                      const float* __restrict__ a_row = nullptr, uint32_t k_valid = 0)
{
#define CNC_LAYER_NT(NTV) case NTV: layer_nt<RELU, A_GLOBAL, NTV>(a_lds, lda, Kp, W, bias, acc, lane, a_row, k_valid); break;
    switch (Hp / 16) {      // wave-uniform
        CNC_LAYER_NT(1) CNC_LAYER_NT(2) CNC_LAYER_NT(3) CNC_LAYER_NT(4) CNC_LAYER_NT(5)
        CNC_LAYER_NT(6) CNC_LAYER_NT(7) CNC_LAYER_NT(8) CNC_LAYER_NT(9) CNC_LAYER_NT(10)
    default: break;
    }
#undef CNC_LAYER_NT
}

__device__ __forceinline__ void acc_to_lds(float* __restrict__ dst, uint32_t ld, uint32_t Hp,
                                           const f32x4 (&acc)[kMaxTiles], uint32_t lane)
{
    const uint32_t c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < kMaxTiles; t++) {
        if ((uint32_t)t < Hp / 16) {
#pragma unroll
            for (int i = 0; i < 4; i++) dst[(g * 4 + i) * ld + t * 16 + c] = acc[t][i];
        }
    }
}

// NT0/NT1/NT2 > 0: the layer widths in 16-column tiles are compile-time (one lean kernel per network
// shape the field uses); 0 = decided at run time by the switch in layer().
template <int NT0, int NT1, int NT2>
__global__ __launch_bounds__(64) void k_mlp_forward(MlpArgs p)
{
    extern __shared__ float lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t ld1 = p.Hp[0] + kPad, ld2 = (p.n_layers == 3 ? p.Hp[1] : 0) + kPad;
    float* h1_lds = lds;
    float* h2_lds = h1_lds + 16 * ld1;

    const uint32_t tiles = (p.N + 15) / 16;
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint32_t row0 = tile * 16;
        const uint32_t my_row = row0 + (lane & 15);
        const float*   a_row = my_row < p.N ? p.X + (size_t)my_row * p.ldx : nullptr;

        f32x4 acc[kMaxTiles];
        // (the vector path reads the row's K0p floats — into the row behind it when K0 < K0p — so the matrix's LAST row,
        // which has none, takes the guarded loads: X may be a column window of a wider matrix that ends with the buffer)
        const uint32_t kv = p.K0 | ((p.a_vec && my_row + 1 < p.N) ? 0x80000000u : 0u);
        if constexpr (NT0 > 0) layer_nt<true, true, NT0>(nullptr, 0, p.K0p, p.W[0], p.B[0], acc, lane, a_row, kv);
        else layer<true, true>(nullptr, 0, p.K0p, p.W[0], p.B[0], p.Hp[0], acc, lane, a_row, kv);
        acc_to_lds(h1_lds, ld1, p.Hp[0], acc, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        uint32_t Hlast;
        if (p.n_layers == 3) {
            if constexpr (NT1 > 0) layer_nt<true, false, NT1>(h1_lds, ld1, p.Hp[0], p.W[1], p.B[1], acc, lane, nullptr, 0);
            else layer<true>(h1_lds, ld1, p.Hp[0], p.W[1], p.B[1], p.Hp[1], acc, lane);
            acc_to_lds(h2_lds, ld2, p.Hp[1], acc, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if constexpr (NT2 > 0) layer_nt<false, false, NT2>(h2_lds, ld2, p.Hp[1], p.W[2], p.B[2], acc, lane, nullptr, 0);
            else layer<false>(h2_lds, ld2, p.Hp[1], p.W[2], p.B[2], p.Hp[2], acc, lane);
            Hlast = p.Hp[2];
        } else {
            if constexpr (NT1 > 0) layer_nt<false, false, NT1>(h1_lds, ld1, p.Hp[0], p.W[1], p.B[1], acc, lane, nullptr, 0);
            else layer<false>(h1_lds, ld1, p.Hp[0], p.W[1], p.B[1], p.Hp[1], acc, lane);
            Hlast = p.Hp[1];
        }
        // store: lanes with the same i write 16 consecutive columns of one row
        const uint32_t c = lane & 15, g = lane >> 4;
#pragma unroll
        for (int t = 0; t < kMaxTiles; t++) {
            if ((uint32_t)t < Hlast / 16) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t row = row0 + g * 4 + i, col = t * 16 + c;
                    if (row < p.N && col < p.n_out) p.Y[(size_t)row * p.ldy + col] = acc[t][i];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();   // the wave's LDS region is reused by its next tile
    }
}

// ---------------------------------------------------------------------------------------------
// 32-row variant on v_mfma_f32_32x32x2_f32: one wave owns 32 rows, so every weight fragment fetched
// from L2 serves twice the rows, and a fragment is one float per (tile, k) instead of per 16 columns:
// 5 tiles x 4 k = 20 weight registers per K block of 8 for a 160-wide layer (40 in the 16-row kernel).
// Operand layouts (lane l, i = l & 31, h = l >> 5):  A[i][k = h], B[k = h][j = i],
// D[row = 8*(v >> 2) + 4*h + (v & 3)][col = i] for v = 0..15.  Inside a K block of 8 the lane's four
// k-values are kb + 4h .. 4h+3 (a float4), MFMA m taking the m-th of them from both half-waves.
// Widths are padded to multiples of 32 by the caller, K to multiples of 8.
// ---------------------------------------------------------------------------------------------
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kMaxTiles32 = 5;

template <bool RELU, bool A_GLOBAL, int NT>
__device__ __forceinline__ void layer32(const float* __restrict__ a_lds, uint32_t lda, uint32_t Kp,
                                        const float* __restrict__ W, const float* __restrict__ bias,
                                        f32x16 (&acc)[kMaxTiles32], uint32_t lane,
                                        const float* __restrict__ a_row, uint32_t k_valid)
{
    const uint32_t i = lane & 31, h = lane >> 5;
    const uint32_t lane_off = i * Kp + h * 4;
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int v = 0; v < 16; v++) acc[t][v] = 0;
    auto load_w = [&](uint32_t kb, float4 (&dst)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; t++)
            dst[t] = *reinterpret_cast<const float4*>(W + (size_t)t * 32 * Kp + kb + lane_off);
    };
    const bool     a_vec = (k_valid >> 31) != 0;
    const uint32_t k_real = k_valid & 0x7FFFFFFFu;
    auto load_a = [&](uint32_t kb) -> float4 {
        float4 v{0, 0, 0, 0};
        const uint32_t k = kb + h * 4;
        if (a_row) {
            if (a_vec) {
                v = *reinterpret_cast<const float4*>(a_row + k);
                if (k + 4 > k_real) {
                    if (k >= k_real) v.x = 0;
                    if (k + 1 >= k_real) v.y = 0;
                    if (k + 2 >= k_real) v.z = 0;
                    v.w = 0;
                }
            } else {
                if (k < k_real) v.x = a_row[k];
                if (k + 1 < k_real) v.y = a_row[k + 1];
                if (k + 2 < k_real) v.z = a_row[k + 2];
                if (k + 3 < k_real) v.w = a_row[k + 3];
            }
        }
        return v;
    };
    float4 wn[NT], an{0, 0, 0, 0};
    load_w(0, wn);
    if constexpr (A_GLOBAL) an = load_a(0);
    for (uint32_t kb = 0; kb < Kp; kb += 8) {
        float4 a, w[NT];
        if constexpr (A_GLOBAL) a = an;
        else a = *reinterpret_cast<const float4*>(a_lds + i * lda + kb + h * 4);
#pragma unroll
        for (int t = 0; t < NT; t++) w[t] = wn[t];
        if (kb + 8 < Kp) {
            load_w(kb + 8, wn);
            if constexpr (A_GLOBAL) an = load_a(kb + 8);
        }
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w[t].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w[t].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w[t].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w[t].w, acc[t], 0, 0, 0);
    }
    (void)bias;
}

// bias (+ ReLU) and the C-layout -> row-major store to LDS, one tile at a time: the accumulators are
// read once and never rewritten (modifying them in place first kept a second copy of all 80 alive)
template <bool RELU, int NT>
__device__ __forceinline__ void acc_to_lds32(float* __restrict__ dst, uint32_t ld,
                                             const float* __restrict__ bias,
                                             const f32x16 (&acc)[kMaxTiles32], uint32_t lane)
{
    const uint32_t i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const float b = bias[t * 32 + i];
#pragma unroll
        for (int v = 0; v < 16; v++) {
            float x = acc[t][v] + b;
            if (RELU) x = x > 0 ? x : 0;
            dst[(8 * (v >> 2) + 4 * h + (v & 3)) * ld + t * 32 + i] = x;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 64 rows per wave for the FIRST layer: two 32-row tiles share every weight fragment of the widest
// layer (the L2 weight stream, not the MFMA pipe, is what the 32-row kernel waits for), 160 accumulator
// registers in AGPRs.  The tiles then go through the remaining layers one after the other, so LDS
// still holds one 32-row activation block per wave.
// ---------------------------------------------------------------------------------------------
template <int NT, int T>
__device__ __forceinline__ void layer32_first_xT(uint32_t Kp, const float* __restrict__ W,
                                                 f32x16 (&acc)[T][kMaxTiles32], uint32_t lane,
                                                 const float* const (&rows)[T], uint32_t k_valid,
                                                 const float* __restrict__ last_row)
{
    const uint32_t i = lane & 31, h = lane >> 5;
    const uint32_t lane_off = i * Kp + h * 4;
#pragma unroll
    for (int m = 0; m < T; m++)
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int v = 0; v < 16; v++) acc[m][t][v] = 0;
    auto load_w = [&](uint32_t kb, float4 (&dst)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; t++)
            dst[t] = *reinterpret_cast<const float4*>(W + (size_t)t * 32 * Kp + kb + lane_off);
    };
    const bool     a_vec = (k_valid >> 31) != 0;
    const uint32_t k_real = k_valid & 0x7FFFFFFFu;
    auto load_a = [&](const float* __restrict__ a_row, uint32_t kb) -> float4 {
        float4 v{0, 0, 0, 0};
        const uint32_t k = kb + h * 4;
        if (a_row) {
            if (a_vec && a_row != last_row) {       // (the matrix's last row has no row behind it to read into)
                v = *reinterpret_cast<const float4*>(a_row + k);
                if (k + 4 > k_real) {
                    if (k >= k_real) v.x = 0;
                    if (k + 1 >= k_real) v.y = 0;
                    if (k + 2 >= k_real) v.z = 0;
                    v.w = 0;
                }
            } else {
                if (k < k_real) v.x = a_row[k];
                if (k + 1 < k_real) v.y = a_row[k + 1];
                if (k + 2 < k_real) v.z = a_row[k + 2];
                if (k + 3 < k_real) v.w = a_row[k + 3];
            }
        }
        return v;
    };
    float4 wn[NT], an[T];
#pragma unroll
    for (int m = 0; m < T; m++) an[m] = load_a(rows[m], 0);
    load_w(0, wn);
    for (uint32_t kb = 0; kb < Kp; kb += 8) {
        float4 w[NT], a[T];
#pragma unroll
        for (int m = 0; m < T; m++) a[m] = an[m];
#pragma unroll
        for (int t = 0; t < NT; t++) w[t] = wn[t];
        if (kb + 8 < Kp) {
            load_w(kb + 8, wn);
#pragma unroll
            for (int m = 0; m < T; m++) an[m] = load_a(rows[m], kb + 8);
        }
#define CNC_STEP(C)                                                                                     \
    _Pragma("unroll") for (int t = 0; t < NT; t++) {                                                    \
        _Pragma("unroll") for (int m = 0; m < T; m++)                                                   \
            acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].C, w[t].C, acc[m][t], 0, 0, 0);       \
    }
        CNC_STEP(x) CNC_STEP(y) CNC_STEP(z) CNC_STEP(w)
#undef CNC_STEP
    }
}

// T 32-row tiles per wave through the first layer (weights fetched once per 32*T rows), then one tile
// at a time through the rest, so LDS holds one 32-row activation block per wave.
template <int NT0, int NT1, int NT2, int T>
__global__ __launch_bounds__(64) void k_mlp_forward64(MlpArgs p)
{
    extern __shared__ float lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t ld1 = NT0 * 32 + kPad, ld2 = NT1 * 32 + kPad;
    float* h_lds = lds;

    const uint32_t tiles = (p.N + 32 * T - 1) / (32 * T);
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint32_t row0 = tile * 32 * T;
        const float* rows[T];
#pragma unroll
        for (int m = 0; m < T; m++) {
            const uint32_t r = row0 + m * 32 + (lane & 31);
            rows[m] = r < p.N ? p.X + (size_t)r * p.ldx : nullptr;
        }
        const uint32_t kv = p.K0 | (p.a_vec ? 0x80000000u : 0u);

        f32x16 accT[T][kMaxTiles32], acc[kMaxTiles32];
        layer32_first_xT<NT0, T>(p.K0p, p.W[0], accT, lane, rows, kv, p.X + (size_t)(p.N - 1) * p.ldx);
#pragma unroll
        for (int half = 0; half < T; half++) {
            const uint32_t base_row = row0 + half * 32;
            acc_to_lds32<true, NT0>(h_lds, ld1, p.B[0], accT[half], lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if constexpr (NT2 > 0) {
                layer32<true, false, NT1>(h_lds, ld1, NT0 * 32, p.W[1], p.B[1], acc, lane, nullptr, 0);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                acc_to_lds32<true, NT1>(h_lds, ld2, p.B[1], acc, lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                layer32<false, false, NT2>(h_lds, ld2, NT1 * 32, p.W[2], p.B[2], acc, lane, nullptr, 0);
            } else {
                layer32<false, false, NT1>(h_lds, ld1, NT0 * 32, p.W[1], p.B[1], acc, lane, nullptr, 0);
            }
            constexpr int NTL = NT2 > 0 ? NT2 : NT1;
            const float*   b_last = NT2 > 0 ? p.B[2] : p.B[1];
            const uint32_t i = lane & 31, h = lane >> 5;
#pragma unroll
            for (int t = 0; t < NTL; t++) {
                const uint32_t col = t * 32 + i;
                const float    b = b_last[col];
#pragma unroll
                for (int v = 0; v < 16; v++) {
                    const uint32_t row = base_row + 8 * (v >> 2) + 4 * h + (v & 3);
                    if (row < p.N && col < p.n_out) p.Y[(size_t)row * p.ldy + col] = acc[t][v] + b;
                }
            }
            __builtin_amdgcn_wave_barrier();   // the LDS block is rewritten by the next tile
        }
    }
}

}  // namespace cnc

using namespace cnc;

// Fused 2- or 3-layer fp32 MLP forward.  Weights must be pre-padded by the caller:
//   W_l : [Hp_l, Kp_l] row-major, zero filled outside [H_l, K_l];  b_l : [Hp_l];
//   Kp_0 = roundup16(K0), Kp_l = Hp_{l-1};  Hp_l = roundup16(H_l) <= 160.
extern "C" int cnc_mlp_forward(const float* X, uint32_t N, uint32_t ldx, uint32_t K0,
                               const float* W1, const float* b1, uint32_t H1p,
                               const float* W2, const float* b2, uint32_t H2p,
                               const float* W3, const float* b3, uint32_t H3p,
                               float* Y, uint32_t ldy, uint32_t n_out, void* stream)
{
    if (N == 0) return CNC_OK;
    if (!X || !W1 || !b1 || !W2 || !b2 || !Y) return CNC_ERR_INVALID_VALUE;
    const uint32_t n_layers = (W3 != nullptr) ? 3 : 2;
    if (n_layers == 3 && !b3) return CNC_ERR_INVALID_VALUE;
    const uint32_t K0p = (K0 + 15) / 16 * 16;
    auto bad = [](uint32_t h) { return h == 0 || h % 16 != 0 || h > 16 * kMaxTiles; };
    if (bad(H1p) || bad(H2p) || (n_layers == 3 && bad(H3p)) || K0p > 1024) return CNC_ERR_INVALID_VALUE;
    const uint32_t last = n_layers == 3 ? H3p : H2p;
    if (n_out == 0 || n_out > last) return CNC_ERR_INVALID_VALUE;

    MlpArgs p{};
    p.X = X; p.N = N; p.ldx = ldx; p.K0 = K0; p.K0p = K0p;
    p.W[0] = W1; p.B[0] = b1; p.Hp[0] = H1p;
    p.W[1] = W2; p.B[1] = b2; p.Hp[1] = H2p;
    p.W[2] = W3; p.B[2] = b3; p.Hp[2] = n_layers == 3 ? H3p : 0;
    p.n_layers = n_layers;
    p.Y = Y; p.ldy = ldy; p.n_out = n_out;

    const uint32_t per_wave = 16 * ((H1p + kPad) + ((n_layers == 3 ? H2p : 0) + kPad));
    const size_t   lds_bytes = (size_t)per_wave * sizeof(float);
    if (lds_bytes > 160 * 1024) return CNC_ERR_INVALID_VALUE;
    // > 64 KiB of dynamic LDS needs the opt-in (per device; cheap, so done on every call)
    p.a_vec = (ldx % 4 == 0 && ldx >= K0p && ((uintptr_t)X % 16) == 0) ? 1u : 0u;
    const uint32_t tiles = (N + 15) / 16;
    uint32_t       blocks = tiles;
    if (blocks > 256u * 32) blocks = 256u * 32;   // waves loop over tiles beyond that
#define CNC_MLP_LAUNCH(A, B, C)                                                                        \
    do {                                                                                               \
        if (hipFuncSetAttribute((const void*)k_mlp_forward<A, B, C>,                                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) \
            return CNC_ERR_LAUNCH;                                                                     \
        hipLaunchKernelGGL((k_mlp_forward<A, B, C>), dim3(blocks), dim3(64), lds_bytes,                \
                           (hipStream_t)stream, p);                                                    \
    } while (0)
    // the two networks of the radiance field get kernels with compile-time layer widths
    if (n_layers == 2 && H1p == 160 && H2p == 80) CNC_MLP_LAUNCH(10, 5, 0);
    else if (n_layers == 3 && H1p == 160 && H2p == 160 && H3p == 16) CNC_MLP_LAUNCH(10, 10, 1);
    else CNC_MLP_LAUNCH(0, 0, 0);
#undef CNC_MLP_LAUNCH
    return launch_status();
}

// The same network on the 32-row kernel.  Padding convention differs: W_l [Hp_l, Kp_l] with Hp_l a
// multiple of 32 (<= 160), Kp_0 = roundup8(K0), Kp_l = Hp_{l-1}.  Only the two shapes of the radiance
// field are instantiated: (160, 96[, -]) two layers and (160, 160, 32) three layers.
extern "C" int cnc_mlp_forward32(const float* X, uint32_t N, uint32_t ldx, uint32_t K0,
                                 const float* W1, const float* b1, uint32_t H1p,
                                 const float* W2, const float* b2, uint32_t H2p,
                                 const float* W3, const float* b3, uint32_t H3p,
                                 float* Y, uint32_t ldy, uint32_t n_out, void* stream)
{
    if (N == 0) return CNC_OK;
    if (!X || !W1 || !b1 || !W2 || !b2 || !Y) return CNC_ERR_INVALID_VALUE;
    const uint32_t n_layers = (W3 != nullptr) ? 3 : 2;
    if (n_layers == 3 && !b3) return CNC_ERR_INVALID_VALUE;
    const uint32_t K0p = (K0 + 7) / 8 * 8;
    MlpArgs p{};
    p.X = X; p.N = N; p.ldx = ldx; p.K0 = K0; p.K0p = K0p;
    p.W[0] = W1; p.B[0] = b1; p.Hp[0] = H1p;
    p.W[1] = W2; p.B[1] = b2; p.Hp[1] = H2p;
    p.W[2] = W3; p.B[2] = b3; p.Hp[2] = n_layers == 3 ? H3p : 0;
    p.n_layers = n_layers;
    p.Y = Y; p.ldy = ldy; p.n_out = n_out;
    p.a_vec = (ldx % 4 == 0 && ldx >= K0p && ((uintptr_t)X % 16) == 0) ? 1u : 0u;
    const size_t lds_bytes = (size_t)32 * (160 + kPad) * sizeof(float);
    // two 32-row tiles per wave through the first layer: its weight fragments are fetched once per 64 rows
    uint32_t bw = (N + 63) / 64;
    if (bw > 256u * 32) bw = 256u * 32;
    const bool base = n_layers == 2 && H1p == 160 && H2p == 96 && n_out <= 96;
    const bool head = n_layers == 3 && H1p == 160 && H2p == 160 && H3p == 32 && n_out <= 32;
    if (!base && !head) return CNC_ERR_UNSUPPORTED;
    if (base) hipLaunchKernelGGL((k_mlp_forward64<5, 3, 0, 2>), dim3(bw), dim3(64), lds_bytes, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((k_mlp_forward64<5, 5, 1, 2>), dim3(bw), dim3(64), lds_bytes, (hipStream_t)stream, p);
    return launch_status();
}
