// scan.hip — per-ray segmented inclusive/exclusive sum and product for gfx950.
//
// Stands in for nerfacc/cuda/csrc/scan.cu (:9-304) + include/utils_scan.cuh (:21-263) behind
// include/cnc_hip.h.
//
// The reference scans 32-element tiles with a shared-memory up-sweep/down-sweep tree and carries
// the running total into element 0 of the next tile.  Floating-point addition is not associative,
// so to return the SAME bits this kernel keeps that association but runs it in registers:
// a 64-lane wave holds two rays, 32 lanes (one tile element each) per ray, and every tree level is
// one lane-shift (ds_bpermute / DPP via __shfl_up, width 32) plus one predicated add — no LDS
// allocation, no block barriers.  Up-sweep level d: lanes with (j+1) % 2d == 0 take
// op(v[j-d], v[j]); down-sweep level d: lanes with (j+1) % 2d == d and j+1 >= 3d do the same.
//
// The product-backward entry points fuse the reference's two elementwise ATen passes
// (grad_outputs*outputs before, / inputs.clamp_min(1e-10) after; scan.cu:199-210) into the scan.
#include "common.hpp"

namespace cnc {

template <bool PROD>
__device__ __forceinline__ float op(float a, float b)
{
    return PROD ? a * b : a + b;
}

template <bool PROD>
__device__ __forceinline__ float tile_scan(float v, uint32_t j)
{
#pragma unroll
    for (uint32_t d = 1; d <= 16; d <<= 1) {
        const float up = __shfl_up(v, d, 32);
        if (((j + 1) & (2 * d - 1)) == 0) v = op<PROD>(up, v);
    }
#pragma unroll
    for (uint32_t d = 8; d >= 1; d >>= 1) {
        const float up = __shfl_up(v, d, 32);
        if (((j + 1) & (2 * d - 1)) == d && (j + 1) >= 3 * d) v = op<PROD>(up, v);
    }
    return v;
}

template <bool EXCL, bool PROD>
__global__ __launch_bounds__(256) void k_segmented_scan(
    const int64_t* __restrict__ starts, const int64_t* __restrict__ cnts,
    const float* __restrict__ in, const float* __restrict__ premul, const float* __restrict__ postdiv,
    float* __restrict__ out, uint32_t n_rays, int reverse, int normalize)
{
    const uint32_t j = threadIdx.x & 31;
    const uint32_t ray = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const bool     live = ray < n_rays;
    const int64_t  s0 = live ? starts[ray] : 0;
    const uint32_t n = live ? (uint32_t)cnts[ray] : 0u;
    const float    init = PROD ? 1.0f : 0.0f;

    // both rays of the wave run the same number of tiles (shuffles need all lanes converged)
    const uint32_t n_other = __shfl_xor(n, 32);
    const uint32_t n_max = n > n_other ? n : n_other;

    auto idx = [&](uint32_t e) -> int64_t { return reverse ? s0 + (int64_t)(n - 1 - e) : s0 + (int64_t)e; };

    float total = init;
    if (EXCL && n > 0 && j == 0) {
        float z = init;
        if (postdiv) z = z / fmaxf(postdiv[idx(0)], 1e-10f);
        out[idx(0)] = z;
    }
    for (uint32_t col = 0; col < n_max; col += 32) {
        const uint32_t e = col + j;
        float v = init;
        if (e < n) {
            const int64_t at = idx(e);
            v = in[at];
            if (premul) v = v * premul[at];
        }
        if (j == 0) v = op<PROD>(v, total);
        v = tile_scan<PROD>(v, j);
        total = __shfl(v, 31, 32);
        const uint32_t dst = EXCL ? e + 1 : e;
        if (col < n && dst < n) {
            float r = v;
            const int64_t at = idx(dst);
            if (postdiv) r = r / fmaxf(postdiv[at], 1e-10f);
            out[at] = r;
        }
    }
    if (normalize && n > 0) {
        // `total` here is the last tile's element 31 == the ray's grand total (tiles are padded
        // with the identity), same value the reference divides by (utils_scan.cuh:96,100-110)
        const float den = fmaxf(total, 1e-10f);
        // lane j rescales exactly the elements it stored above (same stride-32 ownership), so
        // program order already makes its own stores visible to it
        for (uint32_t e = (EXCL ? 1u : 0u) + j; e < n; e += 32) {
            const int64_t at = idx(e);
            out[at] = out[at] / den;
        }
    }
}

template <bool EXCL, bool PROD>
static int launch_scan(const int64_t* starts, const int64_t* cnts, const float* in,
                       const float* premul, const float* postdiv, float* out, uint32_t n_rays,
                       int64_t n_edges, int reverse, int normalize, void* stream)
{
    if (n_edges == 0 || n_rays == 0) return CNC_OK;
    if (!starts || !cnts || !in || !out) return CNC_ERR_INVALID_VALUE;
    const uint32_t rays_per_block = 256 / 32;
    hipLaunchKernelGGL((k_segmented_scan<EXCL, PROD>), dim3(div_up(n_rays, rays_per_block)),
                       dim3(256), 0, (hipStream_t)stream, starts, cnts, in, premul, postdiv, out,
                       n_rays, reverse, normalize);
    return launch_status();
}

}  // namespace cnc

using namespace cnc;

extern "C" int cnc_inclusive_sum(const int64_t* chunk_starts, const int64_t* chunk_cnts,
                                 const float* inputs, float* outputs, uint32_t n_rays,
                                 int64_t n_edges, int32_t normalize, int32_t backward, void* stream)
{
    return launch_scan<false, false>(chunk_starts, chunk_cnts, inputs, nullptr, nullptr, outputs,
                                     n_rays, n_edges, backward, normalize, stream);
}

extern "C" int cnc_exclusive_sum(const int64_t* chunk_starts, const int64_t* chunk_cnts,
                                 const float* inputs, float* outputs, uint32_t n_rays,
                                 int64_t n_edges, int32_t normalize, int32_t backward, void* stream)
{
    return launch_scan<true, false>(chunk_starts, chunk_cnts, inputs, nullptr, nullptr, outputs,
                                    n_rays, n_edges, backward, normalize, stream);
}

extern "C" int cnc_inclusive_prod_forward(const int64_t* chunk_starts, const int64_t* chunk_cnts,
                                          const float* inputs, float* outputs, uint32_t n_rays,
                                          int64_t n_edges, void* stream)
{
    return launch_scan<false, true>(chunk_starts, chunk_cnts, inputs, nullptr, nullptr, outputs,
                                    n_rays, n_edges, 0, 0, stream);
}

extern "C" int cnc_exclusive_prod_forward(const int64_t* chunk_starts, const int64_t* chunk_cnts,
                                          const float* inputs, float* outputs, uint32_t n_rays,
                                          int64_t n_edges, void* stream)
{
    return launch_scan<true, true>(chunk_starts, chunk_cnts, inputs, nullptr, nullptr, outputs,
                                   n_rays, n_edges, 0, 0, stream);
}

extern "C" int cnc_inclusive_prod_backward(const int64_t* chunk_starts, const int64_t* chunk_cnts,
                                           const float* inputs, const float* outputs,
                                           const float* grad_outputs, float* grad_inputs,
                                           uint32_t n_rays, int64_t n_edges, void* stream)
{
    if (n_edges != 0 && (!inputs || !outputs)) return CNC_ERR_INVALID_VALUE;
    return launch_scan<false, false>(chunk_starts, chunk_cnts, grad_outputs, outputs, inputs,
                                     grad_inputs, n_rays, n_edges, 1, 0, stream);
}

extern "C" int cnc_exclusive_prod_backward(const int64_t* chunk_starts, const int64_t* chunk_cnts,
                                           const float* inputs, const float* outputs,
                                           const float* grad_outputs, float* grad_inputs,
                                           uint32_t n_rays, int64_t n_edges, void* stream)
{
    if (n_edges != 0 && (!inputs || !outputs)) return CNC_ERR_INVALID_VALUE;
    return launch_scan<true, false>(chunk_starts, chunk_cnts, grad_outputs, outputs, inputs,
                                    grad_inputs, n_rays, n_edges, 1, 0, stream);
}
