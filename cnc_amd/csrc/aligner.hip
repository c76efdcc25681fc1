// aligner.hip — context-model aligner for gfx950: occupancy-box query (mask + overlap volume)
// and ragged <-> padded packing of per-vertex features by hash slot.
//
// Stands in for my_cuda_backen/aligner_kernel.cu (query_mask_3D_kernel_* :4-326,
// align_and_pack_{forward,backward}_kernel :413-516) behind include/cnc_hip.h.
//
// Both are HBM-bound integer/byte kernels:
//   * query: 6 B in + 6 B out per vertex (+8 B for the per-point resolution list); the 2 MiB
//     occupancy grid stays resident in L2.  One lane per vertex, outputs written coalesced.
//   * pack: the padded [N, M, F] tensor is written once with consecutive lanes on consecutive
//     floats of a row (reference: one thread per element on a (2,128,1) block, i.e. strided).
#include "common.hpp"

namespace cnc {

template <uint32_t D, bool QLIST>
__global__ __launch_bounds__(256) void k_query_mask(const int16_t* __restrict__ points,
                                                    const uint8_t* __restrict__ vxl, uint32_t Rb,
                                                    int16_t* __restrict__ mask,
                                                    int32_t* __restrict__ overlap,
                                                    int32_t resolution,
                                                    const int64_t* __restrict__ res_list,
                                                    uint32_t N)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float Rb_re = 1.0f / (float)(int)Rb;
    const float R = QLIST ? (float)res_list[i] : (float)resolution;
    const float scale_re = 1.0f / (R - 2.0f);

    float    pn[D];
    uint32_t lo[D], hi[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++)
        box_range((float)points[(size_t)i * D + d], scale_re, Rb, lo[d], hi[d], pn[d]);

    bool  m = false;
    float area = 0;
    for (uint32_t a = lo[0]; a <= hi[0]; a++) {
        const float ra = fminf(__builtin_fmaf((float)(int)a, Rb_re, Rb_re), pn[0] + scale_re);
        const float la = fmaxf((float)(int)a * Rb_re, pn[0] - scale_re);
        const float oa = ra - la;
        for (uint32_t b = lo[1]; b <= hi[1]; b++) {
            const float rb = fminf(__builtin_fmaf((float)(int)b, Rb_re, Rb_re), pn[1] + scale_re);
            const float lb = fmaxf((float)(int)b * Rb_re, pn[1] - scale_re);
            const float ob = rb - lb;
            if constexpr (D == 2) {
                const bool mt = vxl[a * Rb + b] != 0;
                m |= mt;
                if (mt) area = __builtin_fmaf(oa, ob, area);
            } else {
                const float oab = oa * ob;
                for (uint32_t c = lo[2]; c <= hi[2]; c++) {
                    const float rc = fminf(__builtin_fmaf((float)(int)c, Rb_re, Rb_re), pn[2] + scale_re);
                    const float lc = fmaxf((float)(int)c * Rb_re, pn[2] - scale_re);
                    const float oc = rc - lc;
                    const bool  mt = vxl[(a * Rb + b) * Rb + c] != 0;
                    m |= mt;
                    if (mt) area = __builtin_fmaf(oab, oc, area);
                }
            }
        }
    }
    const float Rbf = (float)(int)Rb;
    area = area * Rbf * Rbf;
    if constexpr (D == 3) area = area * Rbf;
    mask[i] = (int16_t)m;
    overlap[i] = (int32_t)(area * 1000);
}

// packed[i][j][k] = j < cnt[i] ? feat[cumsum[i]+j][k] : V.  One lane per float of the output.
__global__ __launch_bounds__(256) void k_pack_fwd(const float* __restrict__ feat,
                                                  const int64_t* __restrict__ cnt,
                                                  const int64_t* __restrict__ cumsum,
                                                  float* __restrict__ packed, uint32_t N,
                                                  uint32_t M, uint32_t F, float V)
{
    const uint64_t total = (uint64_t)N * M * F;
    const uint64_t row_len = (uint64_t)M * F;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i = (uint32_t)(e / row_len);
        const uint32_t r = (uint32_t)(e - (uint64_t)i * row_len);
        const uint32_t j = r / F;
        packed[e] = ((int64_t)(j + 1) > cnt[i]) ? V : feat[(size_t)cumsum[i] * F + r];
    }
}

__global__ __launch_bounds__(256) void k_pack_bwd(const float* __restrict__ dpacked,
                                                  const int64_t* __restrict__ cnt,
                                                  const int64_t* __restrict__ cumsum,
                                                  float* __restrict__ dfeat, uint32_t N,
                                                  uint32_t M, uint32_t F)
{
    const uint64_t total = (uint64_t)N * M * F;
    const uint64_t row_len = (uint64_t)M * F;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i = (uint32_t)(e / row_len);
        const uint32_t r = (uint32_t)(e - (uint64_t)i * row_len);
        const uint32_t j = r / F;
        if ((int64_t)(j + 1) > cnt[i]) continue;
        dfeat[(size_t)cumsum[i] * F + r] = dpacked[e];
    }
}

// Hash fusion without the padded tensor: out[i][f] = sum_j w[s_i + j] * v[s_i + j][f]  (/ sum_j w)
// over the rows s_i .. s_i + cnt_i - 1 of slot i.  The reference materialises [N, max(cnt), F]
// (max(cnt) up to 288 at R=514), multiplies and reduces it (utils_bpp_acc.py:688-695); this reads
// every value once.  One lane per (slot, feature): the F lanes of a slot read a row's F floats as
// one contiguous segment.  mode: 0 = plain (weighted) sum, 1 = divide by the weight sum,
// 2 = divide by the row count (mean; weights ignored if NULL).
template <uint32_t F>
__global__ __launch_bounds__(256) void k_segment_wsum(const float* __restrict__ v,
                                                      const float* __restrict__ w,
                                                      const int64_t* __restrict__ cumsum,
                                                      float* __restrict__ out, uint32_t N,
                                                      int mode, const int64_t* __restrict__ order)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = t / F, f = t % F;
    if (i >= N) return;
    const int64_t s = cumsum[i], e = cumsum[i + 1];
    float acc = 0, wsum = 0;
    for (int64_t r = s; r < e; r++) {
        const float wr = w ? w[r] : 1.0f;
        acc += wr * v[(order ? order[r] : r) * F + f];
        wsum += wr;
    }
    if (mode == 1) acc = acc / wsum;
    else if (mode == 2) acc = acc / (float)(e - s);
    out[(size_t)i * F + f] = acc;
}

static uint32_t stream_grid(uint64_t total)
{
    const uint64_t want = (total + 255) / 256;
    const uint64_t cap = 256ull * 16;   // 256 CUs x 16 blocks; grid-stride the rest
    return (uint32_t)(want < cap ? (want ? want : 1) : cap);
}

}  // namespace cnc

using namespace cnc;

static int query_mask_impl(const int16_t* points, uint32_t D, const uint8_t* vxl, uint32_t Rb,
                           int16_t* mask, int32_t* overlap, int32_t resolution,
                           const int64_t* res_list, uint32_t N, void* stream)
{
    if (N == 0) return CNC_OK;
    if (!points || !vxl || !mask || !overlap || Rb == 0) return CNC_ERR_INVALID_VALUE;
    const dim3 grid(div_up(N, 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (D == 2) {
        if (res_list) hipLaunchKernelGGL((k_query_mask<2, true>), grid, block, 0, s, points, vxl, Rb, mask, overlap, resolution, res_list, N);
        else hipLaunchKernelGGL((k_query_mask<2, false>), grid, block, 0, s, points, vxl, Rb, mask, overlap, resolution, res_list, N);
    } else if (D == 3) {
        if (res_list) hipLaunchKernelGGL((k_query_mask<3, true>), grid, block, 0, s, points, vxl, Rb, mask, overlap, resolution, res_list, N);
        else hipLaunchKernelGGL((k_query_mask<3, false>), grid, block, 0, s, points, vxl, Rb, mask, overlap, resolution, res_list, N);
    } else {
        return CNC_ERR_INVALID_VALUE;   // reference: switch (num_dim) has only cases 2 and 3
    }
    return launch_status();
}

extern "C" int cnc_query_mask_3D(const int16_t* points, uint32_t D, const uint8_t* binary_vxl,
                                 uint32_t Rb, int16_t* mask, int32_t* overlap, int32_t resolution,
                                 uint32_t N, void* stream)
{
    return query_mask_impl(points, D, binary_vxl, Rb, mask, overlap, resolution, nullptr, N, stream);
}

extern "C" int cnc_query_mask_3D_qlist(const int16_t* points, uint32_t D,
                                       const uint8_t* binary_vxl, uint32_t Rb, int16_t* mask,
                                       int32_t* overlap, const int64_t* resolution_list,
                                       uint32_t N, void* stream)
{
    if (N != 0 && !resolution_list) return CNC_ERR_INVALID_VALUE;
    return query_mask_impl(points, D, binary_vxl, Rb, mask, overlap, 0, resolution_list, N, stream);
}

extern "C" int cnc_align_and_pack_forward(const float* feat, const int64_t* cnt,
                                          const int64_t* cumsum, float* packed, uint32_t N,
                                          uint32_t M, uint32_t F, float V, void* stream)
{
    const uint64_t total = (uint64_t)N * M * F;
    if (total == 0) return CNC_OK;
    if (!feat || !cnt || !cumsum || !packed) return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_pack_fwd, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       feat, cnt, cumsum, packed, N, M, F, V);
    return launch_status();
}

extern "C" int cnc_align_and_pack_backward(const float* dL_packed, const int64_t* cnt,
                                           const int64_t* cumsum, float* dL_feat, uint32_t N,
                                           uint32_t M, uint32_t F, void* stream)
{
    const uint64_t total = (uint64_t)N * M * F;
    if (total == 0) return CNC_OK;
    if (!dL_packed || !cnt || !cumsum || !dL_feat) return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_pack_bwd, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       dL_packed, cnt, cumsum, dL_feat, N, M, F);
    return launch_status();
}

extern "C" int cnc_segment_weighted_sum_gathered(const float* values, const int64_t* order, const float* weights,
                                                 const int64_t* cumsum, float* out, uint32_t N, uint32_t F,
                                                 int32_t mode, void* stream)
{
    if (N == 0) return CNC_OK;
    if (!values || !cumsum || !out || mode < 0 || mode > 2) return CNC_ERR_INVALID_VALUE;
    hipStream_t s = (hipStream_t)stream;
    const dim3  block(256);
#define CNC_SEG(FF) hipLaunchKernelGGL((k_segment_wsum<FF>), dim3(div_up(N * FF, 256)), block, 0, s, values, weights, cumsum, out, N, mode, order)
    switch (F) {
    case 1: CNC_SEG(1); break;
    case 2: CNC_SEG(2); break;
    case 4: CNC_SEG(4); break;
    case 8: CNC_SEG(8); break;
    case 16: CNC_SEG(16); break;
    case 32: CNC_SEG(32); break;
    default: return CNC_ERR_INVALID_VALUE;
    }
#undef CNC_SEG
    return launch_status();
}

extern "C" int cnc_segment_weighted_sum(const float* values, const float* weights,
                                        const int64_t* cumsum, float* out, uint32_t N, uint32_t F,
                                        int32_t mode, void* stream)
{
    return cnc_segment_weighted_sum_gathered(values, nullptr, weights, cumsum, out, N, F, mode, stream);
}
