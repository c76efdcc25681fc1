// grid_input_grad.hip — the dy_dx branch of kernel_grid (gridencoder.cu:319-395) and
// kernel_input_backward (gridencoder.cu:588-614).
//
// Dead in CNC (ngp.py:58-60 refuses calc_grad_inputs and :84 passes dy_dx=None), but part of the
// `_gridencoder` interface: cnc_grid_encode_forward fills dy_dx when it is given, and
// cnc_grid_encode_backward turns it into grad_inputs.  Not a hot path: one lane per (point, level) /
// per (point, axis), written for bit-parity with the oracle (same operation order, fmaf where nvcc
// contracts), not for speed.
#include "common.hpp"
#include "encoder_common.hpp"

namespace cnc {

// dy_dx [N, L, D, F].  Per axis gd: the 2^(D-1) edges along gd; weight (R-2) * product of the other
// axes' interpolation weights; (right - left) table values; no renormalisation over valid corners and
// no occupancy mask (the reference applies neither here); border-ring vertices read as 0.
template <uint32_t D, uint32_t F, bool STE>
__global__ __launch_bounds__(256) void k_grid_dy_dx(const float* __restrict__ inputs,
                                                    const float* __restrict__ emb,
                                                    const int32_t* __restrict__ offsets,
                                                    const int32_t* __restrict__ resolutions,
                                                    float* __restrict__ dy_dx, uint32_t N, uint32_t L,
                                                    const int32_t* __restrict__ min_level_id)
{
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= N) return;
    const uint32_t slot = blockIdx.y;
    const uint32_t level = slot + (min_level_id ? (uint32_t)min_level_id[b] : 0u);
    float* o = dy_dx + (((size_t)b * L + slot) * D) * F;
    float  x[D];
    if (!load_point<D>(inputs, b, x)) {                    // :143-158
#pragma unroll
        for (uint32_t j = 0; j < D * F; j++) o[j] = 0;
        return;
    }
    const uint32_t off = (uint32_t)offsets[level];
    const uint32_t hs = (uint32_t)offsets[level + 1] - off;
    const uint32_t R = (uint32_t)resolutions[level];
    const float*   table = emb + (size_t)off * F;
    float    pos[D];
    uint32_t g[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {                     // :171-177
        float p = x[d] * (float)(R - 2);
        p = p + 0.5f;
        const float fl = floorf(p);
        g[d] = (uint32_t)fl;
        pos[d] = p - fl;
    }
#pragma unroll
    for (uint32_t gd = 0; gd < D; gd++) {
        float acc[F];
#pragma unroll
        for (uint32_t ch = 0; ch < F; ch++) acc[ch] = 0;
#pragma unroll
        for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
            float    w = (float)(R - 2);                   // :333
            uint32_t q[D];
#pragma unroll
            for (uint32_t nd = 0; nd + 1 < D; nd++) {      // :337-347
                const uint32_t d = nd >= gd ? nd + 1 : nd;
                if ((idx & (1u << nd)) == 0) {
                    w *= 1 - pos[d];
                    q[d] = g[d];
                } else {
                    w *= pos[d];
                    q[d] = min(g[d] + 1, R - 1);
                }
            }
            bool edge_l = false, edge_r = false;
            q[gd] = g[gd];                                 // :349-361
#pragma unroll
            for (uint32_t d = 0; d < D; d++) edge_l |= (q[d] == 0) | (q[d] == R - 1);
            const uint32_t rl = edge_l ? 0u : grid_row<D>(q, hs, R);
            q[gd] = min(g[gd] + 1, R - 1);                 // :363-374
#pragma unroll
            for (uint32_t d = 0; d < D; d++) edge_r |= (q[d] == 0) | (q[d] == R - 1);
            const uint32_t rr = edge_r ? 0u : grid_row<D>(q, hs, R);
#pragma unroll
            for (uint32_t ch = 0; ch < F; ch++) {          // :377-387
                float vl = edge_l ? 0.0f : table[(size_t)rl * F + ch];
                float vr = edge_r ? 0.0f : table[(size_t)rr * F + ch];
                if (STE) {   // STE_binary.forward is applied to the table before the reference's call
                    if (!edge_l) vl = vl >= 0 ? 1.0f : -1.0f;
                    if (!edge_r) vr = vr >= 0 ? 1.0f : -1.0f;
                }
                const float t = w * (vr - vl);
                acc[ch] = __builtin_fmaf(t, 1.0f, acc[ch]);   // += t * pos_deriv, pos_deriv = 1
            }
        }
#pragma unroll
        for (uint32_t ch = 0; ch < F; ch++) o[gd * F + ch] = acc[ch];   // :390-393
    }
}

// grad_inputs[b][d] = sum_l sum_ch grad[l][b][ch] * dy_dx[b][l][d][ch], in that order (:596-613)
template <uint32_t D, uint32_t F>
__global__ __launch_bounds__(256) void k_input_backward(const float* __restrict__ grad,
                                                        const float* __restrict__ dy_dx,
                                                        float* __restrict__ grad_inputs, uint32_t N,
                                                        uint32_t L, FeatLayout lay)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= N * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float*   dy = dy_dx + (size_t)b * L * D * F;
    float          result = 0;
    for (uint32_t l = 0; l < L; l++) {
        const float* g = grad + feat_index(lay, l, N, b, F);
#pragma unroll
        for (uint32_t ch = 0; ch < F; ch++)
            result = __builtin_fmaf(g[ch], dy[((size_t)l * D + d) * F + ch], result);
    }
    grad_inputs[t] = result;
}

#define CNC_IG_F(F, CALL)                                 \
    switch (F) {                                          \
    case 1: { constexpr uint32_t FF = 1; CALL; } break;   \
    case 2: { constexpr uint32_t FF = 2; CALL; } break;   \
    case 4: { constexpr uint32_t FF = 4; CALL; } break;   \
    case 8: { constexpr uint32_t FF = 8; CALL; } break;   \
    case 16: { constexpr uint32_t FF = 16; CALL; } break; \
    case 32: { constexpr uint32_t FF = 32; CALL; } break; \
    default: return CNC_ERR_INVALID_VALUE;                \
    }

template <uint32_t D>
static int dy_dx_D(const float* inputs, const float* emb, const int32_t* offsets, const int32_t* resolutions,
                   float* dy_dx, uint32_t N, uint32_t F, uint32_t L, const int32_t* mli, bool ste,
                   hipStream_t s)
{
    const dim3 grid(div_up(N, 256), L);
    CNC_IG_F(F, {
        if (ste) hipLaunchKernelGGL((k_grid_dy_dx<D, FF, true>), grid, dim3(256), 0, s, inputs, emb, offsets, resolutions, dy_dx, N, L, mli);
        else hipLaunchKernelGGL((k_grid_dy_dx<D, FF, false>), grid, dim3(256), 0, s, inputs, emb, offsets, resolutions, dy_dx, N, L, mli);
    });
    return CNC_OK;
}

int launch_dy_dx(const float* inputs, const float* emb, const int32_t* offsets, const int32_t* resolutions,
                 float* dy_dx, uint32_t N, uint32_t D, uint32_t F, uint32_t L, const int32_t* mli, bool ste,
                 hipStream_t s)
{
    switch (D) {
    case 1: return dy_dx_D<1>(inputs, emb, offsets, resolutions, dy_dx, N, F, L, mli, ste, s);
    case 2: return dy_dx_D<2>(inputs, emb, offsets, resolutions, dy_dx, N, F, L, mli, ste, s);
    case 3: return dy_dx_D<3>(inputs, emb, offsets, resolutions, dy_dx, N, F, L, mli, ste, s);
    default: return CNC_ERR_INVALID_VALUE;
    }
}

template <uint32_t D>
static int input_backward_D(const float* grad, const float* dy_dx, float* grad_inputs, uint32_t N, uint32_t F,
                            uint32_t L, FeatLayout lay, hipStream_t s)
{
    const dim3 grid(div_up(N * D, 256));
    CNC_IG_F(F, hipLaunchKernelGGL((k_input_backward<D, FF>), grid, dim3(256), 0, s, grad, dy_dx, grad_inputs, N, L, lay));
    return CNC_OK;
}

int launch_input_backward(const float* grad, const float* dy_dx, float* grad_inputs, uint32_t N, uint32_t D,
                          uint32_t F, uint32_t L, FeatLayout lay, hipStream_t s)
{
    switch (D) {
    case 1: return input_backward_D<1>(grad, dy_dx, grad_inputs, N, F, L, lay, s);
    case 2: return input_backward_D<2>(grad, dy_dx, grad_inputs, N, F, L, lay, s);
    case 3: return input_backward_D<3>(grad, dy_dx, grad_inputs, N, F, L, lay, s);
    default: return CNC_ERR_INVALID_VALUE;
    }
}

}  // namespace cnc
