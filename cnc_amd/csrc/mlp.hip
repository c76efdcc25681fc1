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
};

// One layer for this wave's 16 rows: acc[t] = A(16 x Kp) * W^T tile t, A read from LDS.
template <bool RELU>
__device__ __forceinline__ void layer(const float* __restrict__ a_lds, uint32_t lda, uint32_t Kp,
                                      const float* __restrict__ W, const float* __restrict__ bias,
                                      uint32_t Hp, f32x4 (&acc)[kMaxTiles], uint32_t lane)
{
    const uint32_t r = lane & 15, g = lane >> 4;
    const uint32_t n_tiles = Hp / 16;
#pragma unroll
    for (int t = 0; t < kMaxTiles; t++) acc[t] = f32x4{0, 0, 0, 0};
    // software pipeline: the B fragments (weights, from L2) of K-block kb+16 are requested before the
    // 4 * n_tiles MFMAs of block kb issue, so their latency hides under ~1280 cycles of matrix work
    float4 wn[kMaxTiles];
#pragma unroll
    for (int t = 0; t < kMaxTiles; t++)
        if ((uint32_t)t < n_tiles) wn[t] = *reinterpret_cast<const float4*>(W + (size_t)(t * 16 + r) * Kp + g * 4);
    for (uint32_t kb = 0; kb < Kp; kb += 16) {
        // A fragment for 4 MFMA steps: row r, k = kb + g*4 + {0,1,2,3}
        const float4 a = *reinterpret_cast<const float4*>(a_lds + r * lda + kb + g * 4);
        float4 w[kMaxTiles];
#pragma unroll
        for (int t = 0; t < kMaxTiles; t++) w[t] = wn[t];
        if (kb + 16 < Kp) {
#pragma unroll
            for (int t = 0; t < kMaxTiles; t++)
                if ((uint32_t)t < n_tiles)
                    wn[t] = *reinterpret_cast<const float4*>(W + (size_t)(t * 16 + r) * Kp + kb + 16 + g * 4);
        }
#pragma unroll
        for (int t = 0; t < kMaxTiles; t++) {
            if ((uint32_t)t < n_tiles) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w[t].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w[t].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w[t].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w[t].w, acc[t], 0, 0, 0);
            }
        }
    }
    // bias (+ ReLU): C layout is col = lane & 15, row = (lane >> 4) * 4 + i
#pragma unroll
    for (int t = 0; t < kMaxTiles; t++) {
        if ((uint32_t)t < n_tiles) {
            const float b = bias[t * 16 + r];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float v = acc[t][i] + b;
                if (RELU) v = v > 0 ? v : 0;
                acc[t][i] = v;
            }
        }
    }
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

__global__ __launch_bounds__(64) void k_mlp_forward(MlpArgs p)
{
    extern __shared__ float lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t ld0 = p.K0p + kPad, ld1 = p.Hp[0] + kPad, ld2 = (p.n_layers == 3 ? p.Hp[1] : 0) + kPad;
    float* x_lds = lds;
    float* h1_lds = x_lds + 16 * ld0;
    float* h2_lds = h1_lds + 16 * ld1;

    const uint32_t tiles = (p.N + 15) / 16;
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint32_t row0 = tile * 16;
        // stage 16 input rows (zero padded to K0p): lanes sweep each row contiguously; all loads
        // of a 64-column chunk are issued before any LDS store so they overlap
        for (uint32_t k0 = 0; k0 < p.K0p; k0 += 64) {
            const uint32_t k = k0 + lane;
            float v[16];
#pragma unroll
            for (uint32_t r = 0; r < 16; r++) {
                const uint32_t row = row0 + r;
                v[r] = (row < p.N && k < p.K0) ? p.X[(size_t)row * p.ldx + k] : 0.0f;
            }
            if (k < p.K0p) {
#pragma unroll
                for (uint32_t r = 0; r < 16; r++) x_lds[r * ld0 + k] = v[r];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

        f32x4 acc[kMaxTiles];
        layer<true>(x_lds, ld0, p.K0p, p.W[0], p.B[0], p.Hp[0], acc, lane);
        acc_to_lds(h1_lds, ld1, p.Hp[0], acc, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        uint32_t Hlast;
        if (p.n_layers == 3) {
            layer<true>(h1_lds, ld1, p.Hp[0], p.W[1], p.B[1], p.Hp[1], acc, lane);
            acc_to_lds(h2_lds, ld2, p.Hp[1], acc, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            layer<false>(h2_lds, ld2, p.Hp[1], p.W[2], p.B[2], p.Hp[2], acc, lane);
            Hlast = p.Hp[2];
        } else {
            layer<false>(h1_lds, ld1, p.Hp[0], p.W[1], p.B[1], p.Hp[1], acc, lane);
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

    const uint32_t per_wave = 16 * ((K0p + kPad) + (H1p + kPad) + ((n_layers == 3 ? H2p : 0) + kPad));
    const size_t   lds_bytes = (size_t)per_wave * sizeof(float);
    if (lds_bytes > 160 * 1024) return CNC_ERR_INVALID_VALUE;
    // > 64 KiB of dynamic LDS needs the opt-in (per device; cheap, so done on every call)
    if (hipFuncSetAttribute((const void*)k_mlp_forward, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
        return CNC_ERR_LAUNCH;
    const uint32_t tiles = (N + 15) / 16;
    uint32_t       blocks = tiles;
    if (blocks > 256u * 32) blocks = 256u * 32;   // waves loop over tiles beyond that
    hipLaunchKernelGGL(k_mlp_forward, dim3(blocks), dim3(64), lds_bytes, (hipStream_t)stream, p);
    return launch_status();
}
