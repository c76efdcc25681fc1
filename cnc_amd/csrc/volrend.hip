// volrend.hip — per-ray volume rendering, fused, for gfx950.
//
// The reference composes a ray's colour from a chain of ATen ops over the flattened sample arrays
// (nerfacc/volrend.py): pack_info (index_add_ + cumsum, pack.py:39-46) -> sigma*dt -> exclusive_sum
// (scan.cu) -> exp -> 1-exp -> trans*alpha (volrend.py:258-266,363) -> three index_add_ (:142-153,546)
// -> depth / opacity, + background (:136-140), and autograd replays the chain backwards.  Here one
// kernel per direction walks each ray once:
//
//   k_volrend_fwd   sigma, t -> alpha, transmittance, weight per sample; colour / opacity / depth per ray
//   k_volrend_bwd   dL/d(colour, opacity, depth) per ray -> dL/d(sigma), dL/d(rgb) per sample
//   k_visibility    the sampling-time visibility test (volrend.py:425-475) + kept count per ray
//   k_compact       stable compaction of (ray, t_start, t_end) by that mask, no boolean indexing
//   k_edges_to_samples  interval edges (is_left / is_right, data_specs.py) -> t_starts / t_ends per sample,
//                   from the two-pass or the over-allocated layout of traverse_grids (grid.cu:400-507)
//   k_pack_bounds   sorted ray_indices -> first / one-past-last sample of each ray (pack_info)
//
// Mapping: 32 lanes per ray, two rays per 64-wide wave (as scan.hip), samples in tiles of 32.  The
// running optical depth uses scan.hip's tile tree with the carry folded into element 0 — the SAME float
// association as the reference's exclusive_sum, so transmittance / weights are those of the op chain;
// the per-ray sums are a lane-strided partial sum + butterfly (the reference's atomics have no order).
#include "common.hpp"

namespace cnc {

__device__ __forceinline__ float tile_sum_scan(float v, uint32_t j)
{
#pragma unroll
    for (uint32_t d = 1; d <= 16; d <<= 1) {
        const float up = __shfl_up(v, d, 32);
        if (((j + 1) & (2 * d - 1)) == 0) v = up + v;
    }
#pragma unroll
    for (uint32_t d = 8; d >= 1; d >>= 1) {
        const float up = __shfl_up(v, d, 32);
        if (((j + 1) & (2 * d - 1)) == d && (j + 1) >= 3 * d) v = up + v;
    }
    return v;
}

__device__ __forceinline__ float half_wave_sum(float v)
{
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d, 32);
    return v;
}

struct RaySpan {
    int64_t  s0;
    uint32_t n, n_max;
};

// both rays of a wave run the same number of tiles (the shuffles need converged lanes)
__device__ __forceinline__ RaySpan ray_span(const int64_t* starts, const int64_t* cnts, uint32_t ray, uint32_t n_rays)
{
    RaySpan r;
    const bool live = ray < n_rays;
    r.s0 = live ? starts[ray] : 0;
    r.n = live ? (uint32_t)cnts[ray] : 0u;
    const uint32_t other = __shfl_xor(r.n, 32);
    r.n_max = r.n > other ? r.n : other;
    return r;
}

// exclusive running sum of `v` over the ray, tile by tile: returns the sum of all earlier elements
__device__ __forceinline__ float excl_step(float v, uint32_t j, float& total)
{
    const float before = total;
    if (j == 0) v = v + total;
    v = tile_sum_scan(v, j);
    total = __shfl(v, 31, 32);
    const float up = __shfl_up(v, 1, 32);
    return j == 0 ? before : up;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_volrend_fwd(
    const int64_t* __restrict__ starts, const int64_t* __restrict__ cnts,
    const float* __restrict__ t_starts, const float* __restrict__ t_ends, const float* __restrict__ sigmas,
    const float* __restrict__ rgbs, const float* opacity_in /* may alias `opacity` (iterative render) */, const float* __restrict__ prefix_trans,
    const float* __restrict__ bkgd,
    float* __restrict__ weights, float* __restrict__ trans_out, float* __restrict__ alphas,
    float* __restrict__ colors, float* opacity /* may alias `opacity_in` */, float* __restrict__ depth,
    uint32_t n_rays, uint32_t flags)
{
    const uint32_t j = threadIdx.x & 31;
    const uint32_t ray = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const RaySpan  sp = ray_span(starts, cnts, ray, n_rays);
    // prefix transmittance of an iterative render: what earlier rounds left of the ray (utils.py:431-436)
    const float prefix = (opacity_in && ray < n_rays) ? 1.0f - opacity_in[ray] : 1.0f;
    float total = 0.0f, cr = 0, cg = 0, cb = 0, co = 0, cd = 0;
    for (uint32_t col = 0; col < sp.n_max; col += 32) {
        const uint32_t e = col + j;
        const bool     on = e < sp.n;
        const int64_t  at = sp.s0 + e;
        float ts = 0, te = 0, sdt = 0;
        if (on) {
            ts = t_starts[at];
            te = t_ends[at];
            sdt = sigmas[at] * (te - ts);
        }
        const float before = excl_step(sdt, j, total);
        if (on) {
            const float alpha = 1.0f - expf(-sdt);
            float       tr = expf(-before);
            if (opacity_in) tr = tr * prefix;
            if (prefix_trans) tr = tr * prefix_trans[at];
            const float w = tr * alpha;
            if (weights) weights[at] = w;
            if (trans_out) trans_out[at] = tr;
            if (alphas) alphas[at] = alpha;
            if (rgbs) {
                cr += w * rgbs[at * 3 + 0];
                cg += w * rgbs[at * 3 + 1];
                cb += w * rgbs[at * 3 + 2];
            }
            co += w;
            cd += w * ((ts + te) / 2.0f);
        }
    }
    cr = half_wave_sum(cr);
    cg = half_wave_sum(cg);
    cb = half_wave_sum(cb);
    co = half_wave_sum(co);
    cd = half_wave_sum(cd);
    if (j != 0 || ray >= n_rays) return;
    if (flags & CNC_VOLREND_ACCUMULATE) {           // in-place accumulation of an iterative render
        if (colors) {
            colors[ray * 3 + 0] += cr;
            colors[ray * 3 + 1] += cg;
            colors[ray * 3 + 2] += cb;
        }
        if (opacity) opacity[ray] += co;
        if (depth) depth[ray] += cd;
        return;
    }
    if (flags & CNC_VOLREND_FINALIZE) {             // depth / clamp_min(opacity, eps); colour + bkgd * (1 - opacity)
        cd = cd / fmaxf(co, 1.1920928955078125e-07f);
        if (bkgd) {
            cr = cr + bkgd[0] * (1.0f - co);
            cg = cg + bkgd[1] * (1.0f - co);
            cb = cb + bkgd[2] * (1.0f - co);
        }
    }
    if (colors) {
        colors[ray * 3 + 0] = cr;
        colors[ray * 3 + 1] = cg;
        colors[ray * 3 + 2] = cb;
    }
    if (opacity) opacity[ray] = co;
    if (depth) depth[ray] = cd;
}

// dL/dsigma_k = ( (g_k T_k + gA_k) (1 - alpha_k) - sum_{i>k} (g_i w_i + gT_i T_i) ) * dt_k
//               g_i = dL/dw_i = gC.rgb_i + gO + gD tmid_i [+ grad_weights_i];  gT, gA = optional dL/dtrans, dL/dalphas
// dL/drgb_i   = w_i gC
// The suffix sum runs over the ray back to front with the same tile tree (as the reference's reverse
// exclusive_sum, scan.cu:42-48).  With FINALIZE the per-ray gradients are first pulled back through
// depth/opacity and the background blend.
__global__ __launch_bounds__(256) void k_volrend_bwd(
    const int64_t* __restrict__ starts, const int64_t* __restrict__ cnts,
    const float* __restrict__ t_starts, const float* __restrict__ t_ends, const float* __restrict__ rgbs,
    const float* __restrict__ weights, const float* __restrict__ trans, const float* __restrict__ alphas,
    const float* __restrict__ opacity, const float* __restrict__ depth, const float* __restrict__ bkgd,
    const float* __restrict__ g_colors, const float* __restrict__ g_opacity, const float* __restrict__ g_depth,
    const float* __restrict__ g_weights, const float* __restrict__ g_trans, const float* __restrict__ g_alphas,
    float* __restrict__ g_sigmas, float* __restrict__ g_rgbs, uint32_t n_rays, uint32_t flags)
{
    const uint32_t j = threadIdx.x & 31;
    const uint32_t ray = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const RaySpan  sp = ray_span(starts, cnts, ray, n_rays);
    float gc0 = 0, gc1 = 0, gc2 = 0, go = 0, gd = 0;
    if (ray < n_rays) {
        if (g_colors) {
            gc0 = g_colors[ray * 3 + 0];
            gc1 = g_colors[ray * 3 + 1];
            gc2 = g_colors[ray * 3 + 2];
        }
        if (g_opacity) go = g_opacity[ray];
        if (g_depth) gd = g_depth[ray];
        if (flags & CNC_VOLREND_FINALIZE) {
            const float eps = 1.1920928955078125e-07f;
            const float o = opacity[ray];
            if (bkgd) go -= gc0 * bkgd[0] + gc1 * bkgd[1] + gc2 * bkgd[2];
            const float den = fmaxf(o, eps);
            // depth_out = dsum / max(o, eps), and `depth` holds depth_out
            if (o > eps) go -= gd * depth[ray] / den;
            gd = gd / den;
        }
    }
    float total = 0.0f;
    for (uint32_t col = 0; col < sp.n_max; col += 32) {
        const uint32_t e = col + j;
        const bool     on = e < sp.n;
        const int64_t  at = sp.s0 + (int64_t)(sp.n - 1 - e);       // back to front
        float g = 0, w = 0, ts = 0, te = 0, tr = 0, carry = 0;
        if (on) {
            ts = t_starts[at];
            te = t_ends[at];
            w = weights[at];
            tr = trans[at];
            g = go + gd * ((ts + te) / 2.0f);
            if (rgbs) g += gc0 * rgbs[at * 3 + 0] + gc1 * rgbs[at * 3 + 1] + gc2 * rgbs[at * 3 + 2];
            if (g_weights) g += g_weights[at];
            carry = g * w;
            if (g_trans) carry += g_trans[at] * tr;
        }
        const float after = excl_step(carry, j, total);   // sum over the samples BEHIND this one
        if (on) {
            float own = g * tr;
            if (g_alphas) own += g_alphas[at];
            if (g_sigmas) g_sigmas[at] = (own * (1.0f - alphas[at]) - after) * (te - ts);
            if (g_rgbs) {
                g_rgbs[at * 3 + 0] = w * gc0;
                g_rgbs[at * 3 + 1] = w * gc1;
                g_rgbs[at * 3 + 2] = w * gc2;
            }
        }
    }
}

// render_visibility_from_density / _from_alpha (volrend.py:425-475): visible = T >= early_stop_eps
// (and alpha >= alpha_thre when alpha_thre > 0).  alpha_thre comes from the device (the reference clamps it
// with occs.mean().item(), occ_grid.py:189-190 — a host sync this path does not need).
template <bool FROM_ALPHA>
__global__ __launch_bounds__(256) void k_visibility(
    const int64_t* __restrict__ starts, const int64_t* __restrict__ cnts,
    const float* __restrict__ t_starts, const float* __restrict__ t_ends, const float* __restrict__ values,
    float early_stop_eps, const float* __restrict__ alpha_thre_dev, float alpha_thre_host,
    uint8_t* __restrict__ mask, int64_t* __restrict__ kept, uint32_t n_rays)
{
    const uint32_t j = threadIdx.x & 31;
    const uint32_t ray = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const RaySpan  sp = ray_span(starts, cnts, ray, n_rays);
    float thre = alpha_thre_host;
    if (alpha_thre_dev) thre = fminf(thre, alpha_thre_dev[0]);
    float    total = FROM_ALPHA ? 1.0f : 0.0f;
    uint32_t n_kept = 0;
    for (uint32_t col = 0; col < sp.n_max; col += 32) {
        const uint32_t e = col + j;
        const bool     on = e < sp.n;
        const int64_t  at = sp.s0 + e;
        float alpha = 0, tr;
        if constexpr (FROM_ALPHA) {
            // transmittance = exclusive product of (1 - alpha), scan.hip's product tree
            alpha = on ? values[at] : 0.0f;
            float v = 1.0f - alpha;
            const float before = total;
            if (j == 0) v = v * total;
#pragma unroll
            for (uint32_t d = 1; d <= 16; d <<= 1) {
                const float up = __shfl_up(v, d, 32);
                if (((j + 1) & (2 * d - 1)) == 0) v = up * v;
            }
#pragma unroll
            for (uint32_t d = 8; d >= 1; d >>= 1) {
                const float up = __shfl_up(v, d, 32);
                if (((j + 1) & (2 * d - 1)) == d && (j + 1) >= 3 * d) v = up * v;
            }
            total = __shfl(v, 31, 32);
            const float up = __shfl_up(v, 1, 32);
            tr = j == 0 ? before : up;
        } else {
            float sdt = 0;
            if (on) sdt = values[at] * (t_ends[at] - t_starts[at]);
            const float before = excl_step(sdt, j, total);
            alpha = 1.0f - expf(-sdt);
            tr = expf(-before);
        }
        bool vis = on && tr >= early_stop_eps;
        if (thre > 0.0f) vis = vis && alpha >= thre;
        if (on) mask[at] = vis ? 1 : 0;
        const uint64_t b = __ballot(vis);
        n_kept += __popc((uint32_t)(threadIdx.x & 32 ? b >> 32 : b));
    }
    if (j == 0 && ray < n_rays && kept) kept[ray] = (int64_t)n_kept;
}

// Stable compaction: sample e of ray r survives to out_starts[r] + (number of survivors before it).
__global__ __launch_bounds__(256) void k_compact(
    const int64_t* __restrict__ starts, const int64_t* __restrict__ cnts, const int64_t* __restrict__ out_starts,
    const uint8_t* __restrict__ mask, const float* __restrict__ t_starts, const float* __restrict__ t_ends,
    float* __restrict__ o_starts, float* __restrict__ o_ends, int64_t* __restrict__ o_ray, uint32_t n_rays)
{
    const uint32_t j = threadIdx.x & 31;
    const uint32_t ray = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const RaySpan  sp = ray_span(starts, cnts, ray, n_rays);
    const int64_t  o0 = ray < n_rays ? out_starts[ray] : 0;
    uint32_t       base = 0;
    for (uint32_t col = 0; col < sp.n_max; col += 32) {
        const uint32_t e = col + j;
        const int64_t  at = sp.s0 + e;
        const bool     keep = e < sp.n && mask[at];
        const uint64_t b64 = __ballot(keep);
        const uint32_t b = (uint32_t)(threadIdx.x & 32 ? b64 >> 32 : b64);
        if (keep) {
            const int64_t dst = o0 + base + __popc(b & ((1u << j) - 1u));
            o_starts[dst] = t_starts[at];
            o_ends[dst] = t_ends[at];
            o_ray[dst] = (int64_t)ray;
        }
        base += __popc(b);
    }
}

// Depth window of every ray: samples [win_lo[r], win_lo[r] + win_n[r]) of ray r go to out_starts[r] + k, with the
// position they came from — the sampler evaluates the density window by window and stops behind opaque surfaces.
__global__ __launch_bounds__(256) void k_window_samples(
    const int64_t* __restrict__ starts, const int64_t* __restrict__ win_lo, const int64_t* __restrict__ win_n,
    const int64_t* __restrict__ out_starts, const float* __restrict__ t_starts, const float* __restrict__ t_ends,
    float* __restrict__ o_starts, float* __restrict__ o_ends, int64_t* __restrict__ o_ray, int64_t* __restrict__ o_src,
    uint32_t n_rays)
{
    const uint32_t j = threadIdx.x & 31;
    const uint32_t ray = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    if (ray >= n_rays) return;
    const int64_t n = win_n[ray];
    if (n <= 0) return;
    const int64_t s0 = starts[ray] + win_lo[ray], o0 = out_starts[ray];
    for (int64_t k = j; k < n; k += 32) {
        o_starts[o0 + k] = t_starts[s0 + k];
        o_ends[o0 + k] = t_ends[s0 + k];
        o_ray[o0 + k] = (int64_t)ray;
        o_src[o0 + k] = s0 + k;
    }
}

// The same window as POSITIONS for the field (cnc_ray_window_positions): o + (d (t0 + t1)) / 2, the expression of
// k_sample_positions (march.hip), written where the running sum of the windows puts the ray — the window's sample count
// never leaves the device.
__global__ __launch_bounds__(256) void k_window_positions(
    const int64_t* __restrict__ starts, const int64_t* __restrict__ win_lo, const int64_t* __restrict__ win_n,
    const int64_t* __restrict__ win_ends, const float* __restrict__ t_starts, const float* __restrict__ t_ends,
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, float* __restrict__ positions,
    int64_t* __restrict__ o_src, uint32_t n_rays)
{
    const uint32_t j = threadIdx.x & 31;
    const uint32_t ray = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    if (ray >= n_rays) return;
    const int64_t n = win_n[ray];
    if (n <= 0) return;
    const int64_t s0 = starts[ray] + win_lo[ray], o0 = win_ends[ray] - n;
    float         o[3], d[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        o[a] = rays_o[(size_t)ray * 3 + a];
        d[a] = rays_d[(size_t)ray * 3 + a];
    }
    for (int64_t k = j; k < n; k += 32) {
        const float ta = t_starts[s0 + k], tb = t_ends[s0 + k];
#pragma unroll
        for (int a = 0; a < 3; a++) positions[(o0 + k) * 3 + a] = o[a] + (d[a] * (ta + tb)) / 2.0f;
        o_src[o0 + k] = s0 + k;
    }
}

__global__ __launch_bounds__(256) void k_scatter_counted(const float* __restrict__ values, const int64_t* __restrict__ src,
                                                         float* __restrict__ out, const int64_t* __restrict__ n_dev,
                                                         uint64_t capacity)
{
    const int64_t  n_ = *n_dev;
    const uint64_t n = n_ < 0 ? 0ull : ((uint64_t)n_ < capacity ? (uint64_t)n_ : capacity);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) out[src[i]] = values[i];
}

// exp(-sum of sigma dt over the first cnts[r] samples of ray r): what is left of the ray after them
__global__ __launch_bounds__(256) void k_ray_transmittance(
    const int64_t* __restrict__ starts, const int64_t* __restrict__ cnts, const float* __restrict__ t_starts,
    const float* __restrict__ t_ends, const float* __restrict__ sigmas, float* __restrict__ trans, uint32_t n_rays)
{
    const uint32_t j = threadIdx.x & 31;
    const uint32_t ray = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const bool     live = ray < n_rays;
    const int64_t  s0 = live ? starts[ray] : 0, n = live ? cnts[ray] : 0;
    float          acc = 0.0f;
    for (int64_t k = j; k < n; k += 32) acc += sigmas[s0 + k] * (t_ends[s0 + k] - t_starts[s0 + k]);
    acc = half_wave_sum(acc);
    if (live && j == 0) trans[ray] = expf(-acc);
}

// One step of the front-to-back sampler (occ_grid.py `_density_front_to_back`): the window just evaluated is added to
// the ray's evaluated prefix, what is left of the ray after it decides whether the ray goes on, and the next window is
// sized — `done` and `take` are updated in place (first call: done = 0, take ignored).
__global__ __launch_bounds__(256) void k_ray_window_next(
    const int64_t* __restrict__ starts, const int64_t* __restrict__ cnts, const float* __restrict__ t_starts,
    const float* __restrict__ t_ends, const float* __restrict__ sigmas, int64_t* __restrict__ done,
    int64_t* __restrict__ take, int64_t window, float threshold, int first, uint32_t n_rays)
{
    const uint32_t j = threadIdx.x & 31;
    const uint32_t ray = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const bool     live = ray < n_rays;
    const int64_t  s0 = live ? starts[ray] : 0, total = live ? cnts[ray] : 0;
    const int64_t  n = (live && !first) ? done[ray] + take[ray] : 0;
    float          acc = 0.0f;
    for (int64_t k = j; k < n; k += 32) acc += sigmas[s0 + k] * (t_ends[s0 + k] - t_starts[s0 + k]);
    acc = half_wave_sum(acc);
    if (live && j == 0) {
        const bool    alive = first ? total > 0 : (expf(-acc) >= threshold && n < total);
        const int64_t left = total - n;
        done[ray] = n;
        take[ray] = alive ? (window < 0 || left < window ? left : window) : 0;
    }
}

// t_starts = intervals.vals[is_left], t_ends = intervals.vals[is_right] (occ_grid.py:176-177,
// utils.py:408-409) and the samples' ray ids, without boolean indexing: the k-th left (right) edge of a ray
// is the start (end) of its k-th sample.  iv_starts may describe the over-allocated layout.
__global__ __launch_bounds__(256) void k_edges_to_samples(
    const int64_t* __restrict__ iv_starts, const int64_t* __restrict__ iv_cnts, const float* __restrict__ iv_vals,
    const uint8_t* __restrict__ is_left, const uint8_t* __restrict__ is_right,
    const int64_t* __restrict__ out_starts, float* __restrict__ o_starts, float* __restrict__ o_ends,
    int64_t* __restrict__ o_ray, uint32_t n_rays)
{
    const uint32_t j = threadIdx.x & 31;
    const uint32_t ray = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const RaySpan  sp = ray_span(iv_starts, iv_cnts, ray, n_rays);
    const int64_t  o0 = ray < n_rays ? out_starts[ray] : 0;
    uint32_t       nl = 0, nr = 0;
    for (uint32_t col = 0; col < sp.n_max; col += 32) {
        const uint32_t e = col + j;
        const int64_t  at = sp.s0 + e;
        const bool     on = e < sp.n;
        const bool     l = on && is_left[at], r = on && is_right[at];
        const float    v = on ? iv_vals[at] : 0.0f;
        const uint64_t bl64 = __ballot(l), br64 = __ballot(r);
        const uint32_t bl = (uint32_t)(threadIdx.x & 32 ? bl64 >> 32 : bl64);
        const uint32_t br = (uint32_t)(threadIdx.x & 32 ? br64 >> 32 : br64);
        const uint32_t below = (1u << j) - 1u;
        if (l) {
            const int64_t dst = o0 + nl + __popc(bl & below);
            o_starts[dst] = v;
            if (o_ray) o_ray[dst] = (int64_t)ray;
        }
        if (r) o_ends[o0 + nr + __popc(br & below)] = v;
        nl += __popc(bl);
        nr += __popc(br);
    }
}

// pack_info (pack.py:11-49) for SORTED ray_indices: first[r] / last[r] = first / one-past-last sample of ray r
// (both buffers pre-zeroed by the caller; rays without samples keep 0 / 0).
__global__ __launch_bounds__(256) void k_pack_bounds(const int64_t* __restrict__ ray_indices, int64_t n,
                                                     int64_t* __restrict__ first, int64_t* __restrict__ last,
                                                     int64_t n_rays)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = ray_indices[i];
    if (r < 0 || r >= n_rays) return;
    if (i == 0 || ray_indices[i - 1] != r) first[r] = i;
    if (i == n - 1 || ray_indices[i + 1] != r) last[r] = i + 1;
}

static inline dim3 ray_grid(uint32_t n_rays) { return dim3(div_up(n_rays, 256 / 32)); }

}  // namespace cnc

using namespace cnc;

extern "C" int cnc_volrend_forward(const int64_t* chunk_starts, const int64_t* chunk_cnts, const float* t_starts,
                                   const float* t_ends, const float* sigmas, const float* rgbs,
                                   const float* opacity_in, const float* prefix_trans, const float* render_bkgd,
                                   float* weights, float* trans, float* alphas, float* colors, float* opacity,
                                   float* depth, uint32_t n_rays, uint32_t flags, void* stream)
{
    if (n_rays == 0) return CNC_OK;
    if (!chunk_starts || !chunk_cnts || !t_starts || !t_ends || !sigmas) return CNC_ERR_INVALID_VALUE;
    if ((flags & CNC_VOLREND_ACCUMULATE) && (flags & CNC_VOLREND_FINALIZE)) return CNC_ERR_INVALID_VALUE;
    if (colors && !rgbs) return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_volrend_fwd, ray_grid(n_rays), dim3(256), 0, (hipStream_t)stream, chunk_starts, chunk_cnts,
                       t_starts, t_ends, sigmas, rgbs, opacity_in, prefix_trans, render_bkgd, weights, trans, alphas,
                       colors, opacity, depth, n_rays, flags);
    return launch_status();
}

extern "C" int cnc_volrend_backward(const int64_t* chunk_starts, const int64_t* chunk_cnts, const float* t_starts,
                                    const float* t_ends, const float* rgbs, const float* weights,
                                    const float* trans, const float* alphas, const float* opacity,
                                    const float* depth, const float* render_bkgd, const float* grad_colors,
                                    const float* grad_opacity, const float* grad_depth, const float* grad_weights,
                                    const float* grad_trans, const float* grad_alphas, float* grad_sigmas,
                                    float* grad_rgbs, uint32_t n_rays, uint32_t flags, void* stream)
{
    if (n_rays == 0) return CNC_OK;
    if (!chunk_starts || !chunk_cnts || !t_starts || !t_ends || !weights || !trans || !alphas)
        return CNC_ERR_INVALID_VALUE;
    if ((flags & CNC_VOLREND_FINALIZE) && (!opacity || !depth)) return CNC_ERR_INVALID_VALUE;
    if ((grad_colors || grad_rgbs) && !rgbs) return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_volrend_bwd, ray_grid(n_rays), dim3(256), 0, (hipStream_t)stream, chunk_starts, chunk_cnts,
                       t_starts, t_ends, rgbs, weights, trans, alphas, opacity, depth, render_bkgd, grad_colors,
                       grad_opacity, grad_depth, grad_weights, grad_trans, grad_alphas, grad_sigmas, grad_rgbs, n_rays,
                       flags);
    return launch_status();
}

extern "C" int cnc_render_visibility(const int64_t* chunk_starts, const int64_t* chunk_cnts, const float* t_starts,
                                     const float* t_ends, const float* sigmas_or_alphas, int32_t from_alpha,
                                     float early_stop_eps, float alpha_thre, const float* alpha_thre_cap,
                                     uint8_t* mask, int64_t* kept, uint32_t n_rays, void* stream)
{
    if (n_rays == 0) return CNC_OK;
    if (!chunk_starts || !chunk_cnts || !sigmas_or_alphas || !mask) return CNC_ERR_INVALID_VALUE;
    if (!from_alpha && (!t_starts || !t_ends)) return CNC_ERR_INVALID_VALUE;
    if (from_alpha)
        hipLaunchKernelGGL(k_visibility<true>, ray_grid(n_rays), dim3(256), 0, (hipStream_t)stream, chunk_starts,
                           chunk_cnts, t_starts, t_ends, sigmas_or_alphas, early_stop_eps, alpha_thre_cap,
                           alpha_thre, mask, kept, n_rays);
    else
        hipLaunchKernelGGL(k_visibility<false>, ray_grid(n_rays), dim3(256), 0, (hipStream_t)stream, chunk_starts,
                           chunk_cnts, t_starts, t_ends, sigmas_or_alphas, early_stop_eps, alpha_thre_cap,
                           alpha_thre, mask, kept, n_rays);
    return launch_status();
}

extern "C" int cnc_compact_samples(const int64_t* chunk_starts, const int64_t* chunk_cnts,
                                   const int64_t* out_starts, const uint8_t* mask, const float* t_starts,
                                   const float* t_ends, float* out_t_starts, float* out_t_ends,
                                   int64_t* out_ray_indices, uint32_t n_rays, void* stream)
{
    if (n_rays == 0) return CNC_OK;
    if (!chunk_starts || !chunk_cnts || !out_starts || !mask || !t_starts || !t_ends || !out_t_starts ||
        !out_t_ends || !out_ray_indices)
        return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_compact, ray_grid(n_rays), dim3(256), 0, (hipStream_t)stream, chunk_starts, chunk_cnts,
                       out_starts, mask, t_starts, t_ends, out_t_starts, out_t_ends, out_ray_indices, n_rays);
    return launch_status();
}

extern "C" int cnc_ray_window_samples(const int64_t* chunk_starts, const int64_t* window_first, const int64_t* window_cnts,
                                      const int64_t* out_starts, const float* t_starts, const float* t_ends,
                                      float* out_t_starts, float* out_t_ends, int64_t* out_ray_indices,
                                      int64_t* out_source_index, uint32_t n_rays, void* stream)
{
    if (n_rays == 0) return CNC_OK;
    if (!chunk_starts || !window_first || !window_cnts || !out_starts || !t_starts || !t_ends || !out_t_starts ||
        !out_t_ends || !out_ray_indices || !out_source_index)
        return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_window_samples, ray_grid(n_rays), dim3(256), 0, (hipStream_t)stream, chunk_starts, window_first,
                       window_cnts, out_starts, t_starts, t_ends, out_t_starts, out_t_ends, out_ray_indices,
                       out_source_index, n_rays);
    return launch_status();
}

extern "C" int cnc_ray_window_positions(const int64_t* chunk_starts, const int64_t* win_lo, const int64_t* win_n,
                                        const int64_t* win_ends, const float* t_starts, const float* t_ends,
                                        const float* rays_o, const float* rays_d, float* positions, int64_t* src,
                                        uint32_t n_rays, void* stream)
{
    if (n_rays == 0) return CNC_OK;
    if (!chunk_starts || !win_lo || !win_n || !win_ends || !t_starts || !t_ends || !rays_o || !rays_d || !positions || !src)
        return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_window_positions, ray_grid(n_rays), dim3(256), 0, (hipStream_t)stream, chunk_starts, win_lo, win_n,
                       win_ends, t_starts, t_ends, rays_o, rays_d, positions, src, n_rays);
    return launch_status();
}

extern "C" int cnc_scatter_counted(const float* values, const int64_t* src, float* out, const int64_t* n_dev,
                                   uint64_t capacity, void* stream)
{
    if (capacity == 0) return CNC_OK;
    if (!values || !src || !out || !n_dev) return CNC_ERR_INVALID_VALUE;
    const uint64_t blocks = (capacity + 255) / 256;
    hipLaunchKernelGGL(k_scatter_counted, dim3((uint32_t)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream,
                       values, src, out, n_dev, capacity);
    return launch_status();
}

extern "C" int cnc_ray_transmittance(const int64_t* chunk_starts, const int64_t* chunk_cnts, const float* t_starts,
                                     const float* t_ends, const float* sigmas, float* transmittance, uint32_t n_rays,
                                     void* stream)
{
    if (n_rays == 0) return CNC_OK;
    if (!chunk_starts || !chunk_cnts || !t_starts || !t_ends || !sigmas || !transmittance) return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_ray_transmittance, ray_grid(n_rays), dim3(256), 0, (hipStream_t)stream, chunk_starts, chunk_cnts,
                       t_starts, t_ends, sigmas, transmittance, n_rays);
    return launch_status();
}

extern "C" int cnc_ray_window_next(const int64_t* chunk_starts, const int64_t* chunk_cnts, const float* t_starts,
                                   const float* t_ends, const float* sigmas, int64_t* done, int64_t* take,
                                   int64_t window, float threshold, int first, uint32_t n_rays, void* stream)
{
    if (n_rays == 0) return CNC_OK;
    if (!chunk_starts || !chunk_cnts || !done || !take || (!first && (!t_starts || !t_ends || !sigmas)))
        return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_ray_window_next, ray_grid(n_rays), dim3(256), 0, (hipStream_t)stream, chunk_starts, chunk_cnts,
                       t_starts, t_ends, sigmas, done, take, window, threshold, first, n_rays);
    return launch_status();
}

extern "C" int cnc_interval_edges_to_samples(const int64_t* iv_chunk_starts, const int64_t* iv_chunk_cnts,
                                             const float* iv_vals, const uint8_t* is_left, const uint8_t* is_right,
                                             const int64_t* out_starts, float* out_t_starts, float* out_t_ends,
                                             int64_t* out_ray_indices, uint32_t n_rays, void* stream)
{
    if (n_rays == 0) return CNC_OK;
    if (!iv_chunk_starts || !iv_chunk_cnts || !iv_vals || !is_left || !is_right || !out_starts || !out_t_starts ||
        !out_t_ends)
        return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_edges_to_samples, ray_grid(n_rays), dim3(256), 0, (hipStream_t)stream, iv_chunk_starts,
                       iv_chunk_cnts, iv_vals, is_left, is_right, out_starts, out_t_starts, out_t_ends,
                       out_ray_indices, n_rays);
    return launch_status();
}

extern "C" int cnc_pack_bounds(const int64_t* ray_indices, int64_t n_samples, int64_t* first, int64_t* last,
                               int64_t n_rays, void* stream)
{
    if (n_samples == 0) return CNC_OK;
    if (!ray_indices || !first || !last) return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_pack_bounds, dim3((uint32_t)((n_samples + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       ray_indices, n_samples, first, last, n_rays);
    return launch_status();
}
