// cnt_votes.hip — cnt_np_embed / cnt_np_embed_backward (gridencoder.cu:873-1087) from a static plan,
// without atomics.
//
// The vertex list handed to cnt_np_embed (the occupied finest-level vertices, utils_bpp_acc.py:498-512)
// only changes when the occupancy grid does (every 16 training steps), while the votes are recounted
// three times per step (xy, xz, yz) and differentiated.  The atomic kernels of grid_encode.hip cost one
// memory-side atomic request per vertex (2.1e7 vertices -> ~1.1-1.3 ms per call at full size, the
// largest single item of the context pass).  With the list pre-sorted
//   * by pixel of the projection plane  -> forward  = a segmented count per pixel,
//   * by table row of the finest level  -> backward = a segmented sum per row,
// both are plain gathers + one plain store per pixel / row.  The plan (row and pixel of every vertex)
// comes from cnc_cnt_np_plan; the sorting itself is torch.sort in the host mirror, once per refresh.
// Forward counts are integers, so the result is bit-identical to the atomic kernel's.
#include "common.hpp"

namespace cnc {

// row / pixel of every vertex; 0xFFFFFFFF for vertices cnt_np_embed skips (not strictly inside)
__global__ __launch_bounds__(256) void k_cnt_plan(const int16_t* __restrict__ inputs, uint32_t N,
                                                  uint32_t R, uint32_t hs, uint32_t axis,
                                                  uint32_t* __restrict__ rows,
                                                  uint32_t* __restrict__ pixels)
{
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= N) return;
    uint32_t q[3];
    bool     inside = true;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        q[d] = (uint32_t)(int32_t)inputs[(size_t)b * 3 + d];
        inside &= !(q[d] <= 0 || q[d] >= R - 1);
    }
    const uint32_t u = axis == 2 ? q[1] : q[0];
    const uint32_t w = axis == 0 ? q[1] : q[2];
    if (rows) rows[b] = inside ? grid_row<3>(q, hs, R) : 0xFFFFFFFFu;
    if (pixels) pixels[b] = inside ? (u - 1) * (R - 2) + (w - 1) : 0xFFFFFFFFu;
}

// The whole plan input in one pass: table row and the three planes' pixels of every vertex, skipped vertices keyed
// PAST the last row / pixel (hs, (R-2)^2) so that a sort leaves them behind the last segment instead of having to be
// compacted away, and per plane whether the list is already in pixel order (unsorted[a] != 0 if some key is smaller
// than its predecessor's) — the host then sorts only what needs sorting and reads one 12-byte answer.
__global__ __launch_bounds__(256) void k_cnt_plan3(const int16_t* __restrict__ inputs, uint32_t N, uint32_t R,
                                                   uint32_t hs, int32_t* __restrict__ rows,
                                                   int32_t* __restrict__ pix_xy, int32_t* __restrict__ pix_xz,
                                                   int32_t* __restrict__ pix_yz, int32_t* __restrict__ unsorted)
{
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= N) return;
    const uint32_t S = R - 2, P = S * S;
    uint32_t       key[2][3];
#pragma unroll
    for (int w = 0; w < 2; w++) {                      // w = 1: the predecessor (for the order flags)
        const uint32_t i = b >= (uint32_t)w ? b - w : b;
        uint32_t       q[3];
        bool           inside = true;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            q[d] = (uint32_t)(int32_t)inputs[(size_t)i * 3 + d];
            inside &= !(q[d] <= 0 || q[d] >= R - 1);
        }
        key[w][0] = inside ? (q[0] - 1) * S + (q[1] - 1) : P;
        key[w][1] = inside ? (q[0] - 1) * S + (q[2] - 1) : P;
        key[w][2] = inside ? (q[1] - 1) * S + (q[2] - 1) : P;
        if (w == 0) rows[b] = inside ? (int32_t)grid_row<3>(q, hs, R) : (int32_t)hs;
    }
    pix_xy[b] = (int32_t)key[0][0];
    pix_xz[b] = (int32_t)key[0][1];
    pix_yz[b] = (int32_t)key[0][2];
#pragma unroll
    for (int a = 0; a < 3; a++)
        if (key[0][a] < key[1][a]) atomicOr(&unsorted[a], 1);     // rare (never, for a list in that plane's order)
}

// forward: one wave per pixel; lane = (vertex slot, channel); out[p][ch][{pos, neg}]
template <uint32_t F>
__global__ __launch_bounds__(64) void k_cnt_votes(const uint32_t* __restrict__ rows_by_pixel,
                                                  const int32_t* __restrict__ seg,
                                                  const float* __restrict__ emb,
                                                  float* __restrict__ out, uint32_t P)
{
    constexpr uint32_t VPI = 64 / F;                 // vertices per iteration
    const uint32_t p = blockIdx.x;
    if (p >= P) return;
    const uint32_t lane = threadIdx.x, ch = lane % F, vs = lane / F;
    const int32_t  s = seg[p], e = seg[p + 1];
    float          pos = 0;
    // four independent index -> row gathers in flight per lane (the loop is latency-bound otherwise)
    for (int32_t k = s + (int32_t)vs; k < e; k += (int32_t)(4 * VPI)) {
        uint32_t row[4];
        float    v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int32_t kk = k + u * (int32_t)VPI;
            row[u] = kk < e ? rows_by_pixel[kk] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = row[u] != 0xFFFFFFFFu ? emb[(size_t)row[u] * F + ch] : 0.0f;
#pragma unroll
        for (int u = 0; u < 4; u++)
            pos += ((double)v[u] > 0.9) ? 1.0f : 0.0f;   // float vs double literal, gridencoder.cu:909
    }
#pragma unroll
    for (uint32_t m = F; m < 64; m <<= 1) pos += __shfl_xor(pos, (int)m);
    if (vs == 0) {
        float2 r;
        r.x = pos;
        r.y = (float)(e - s) - pos;
        *reinterpret_cast<float2*>(out + ((size_t)p * F + ch) * 2) = r;
    }
}

// The three projections of a step count the SAME votes (embedding > 0.9 per (row, channel)): k_vote_masks packs them
// once into one word per table row, and the forward below gathers that word (4 bytes per vertex instead of a 4 F byte
// row, from a 2 MB array that stays in L2).
__global__ __launch_bounds__(256) void k_vote_masks(const float* __restrict__ emb, uint32_t rows, uint32_t F,
                                                    uint32_t* __restrict__ masks)
{
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    uint32_t m = 0;
    for (uint32_t ch = 0; ch < F; ch++) m |= ((double)emb[(size_t)r * F + ch] > 0.9 ? 1u : 0u) << ch;
    masks[r] = m;
}

// forward from the masks: one wave per pixel, one lane per vertex, a ballot per channel.  (A wave walking 8 pixels
// one after the other to save dispatches: 0.27 -> 0.48 ms for the three calls — the three dependent loads per pixel
// want many pixels in flight, not few dispatches.)
__global__ __launch_bounds__(64) void k_cnt_votes_masked(const uint32_t* __restrict__ rows_by_pixel,
                                                         const int32_t* __restrict__ seg,
                                                         const uint32_t* __restrict__ masks, float* __restrict__ out,
                                                         uint32_t P, uint32_t F)
{
    const uint32_t p = blockIdx.x;
    if (p >= P) return;
    const uint32_t lane = threadIdx.x;
    const int32_t  s = seg[p], e = seg[p + 1];
    uint32_t       cnt = 0;                       // lane ch < F keeps channel ch's count
    for (int32_t k = s; k < e; k += 64) {
        const int32_t  kk = k + (int32_t)lane;
        const uint32_t row = kk < e ? rows_by_pixel[kk] : 0xFFFFFFFFu;
        const uint32_t m = row != 0xFFFFFFFFu ? masks[row] : 0u;
        for (uint32_t ch = 0; ch < F; ch++) {
            const uint32_t c = (uint32_t)__popcll(__ballot((m >> ch) & 1u));
            if (lane == ch) cnt += c;
        }
    }
    if (lane < F) {
        float2 r;
        r.x = (float)cnt;
        r.y = (float)(e - s) - r.x;
        *reinterpret_cast<float2*>(out + ((size_t)p * F + lane) * 2) = r;
    }
}

// backward: one wave per table row; lane = (vertex slot, channel).  Every vertex of a row sees the
// same embedding, hence the same vote: grad_emb[row][ch] = +sum G[pixel][ch][0] or -sum G[pixel][ch][1],
// G = grad / outputs_sum (prepared by the caller).  Plain store: each row has one writer.
template <uint32_t F>
__global__ __launch_bounds__(64) void k_cnt_votes_bwd(const uint32_t* __restrict__ pixels_by_row,
                                                      const int32_t* __restrict__ seg,
                                                      const float* __restrict__ emb,
                                                      const float* __restrict__ G,
                                                      float* __restrict__ grad_emb, uint32_t rows)
{
    constexpr uint32_t VPI = 64 / F;
    const uint32_t r = blockIdx.x;
    if (r >= rows) return;
    const uint32_t lane = threadIdx.x, ch = lane % F, vs = lane / F;
    const int32_t  s = seg[r], e = seg[r + 1];
    if (s == e) return;                              // no vertex maps here: gradient stays as it is
    const bool     pos = (double)emb[(size_t)r * F + ch] > 0.9;
    const uint32_t sel = pos ? 0u : 1u;
    float          acc = 0;
    for (int32_t k = s + (int32_t)vs; k < e; k += (int32_t)(4 * VPI)) {
        uint32_t px[4];
        float    v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int32_t kk = k + u * (int32_t)VPI;
            px[u] = kk < e ? pixels_by_row[kk] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            v[u] = px[u] != 0xFFFFFFFFu ? G[((size_t)px[u] * F + ch) * 2 + sel] : 0.0f;
#pragma unroll
        for (int u = 0; u < 4; u++) acc += v[u];
    }
#pragma unroll
    for (uint32_t m = F; m < 64; m <<= 1) acc += __shfl_xor(acc, (int)m);
    if (vs == 0) grad_emb[(size_t)r * F + ch] += pos ? acc : -acc;
}

// backward of the three projections in one pass: the row's vertices are the same for the three planes (one row_seg),
// only their pixels differ.  grad_emb is WRITTEN (rows without vertices get 0): no zero-fill, no accumulation of three
// table-sized tensors by the caller.
struct Votes3 {
    const uint32_t* pixels_by_row[3];
    const float*    G[3];
    const uint32_t* xyz_by_row;      // XYZ form: one packed vertex (x | y << 10 | z << 20) instead of three pixels
    uint32_t        S;               // resolution - 2
};

template <uint32_t F, bool XYZ = false>
__global__ __launch_bounds__(64) void k_cnt_votes_bwd3(Votes3 v3, const int32_t* __restrict__ seg,
                                                       const float* __restrict__ emb, float* __restrict__ grad_emb,
                                                       uint32_t rows)
{
    constexpr uint32_t VPI = 64 / F;
    const uint32_t r = blockIdx.x;
    if (r >= rows) return;
    const uint32_t lane = threadIdx.x, ch = lane % F, vs = lane / F;
    const int32_t  s = seg[r], e = seg[r + 1];
    float          acc = 0;
    const bool     pos = (double)emb[(size_t)r * F + ch] > 0.9;
    const uint32_t sel = pos ? 0u : 1u;
    if (XYZ && s != e) {
        // one packed vertex per entry: loaded and decoded ONCE for the three planes, whose gathers are in flight together
        // (decoding inside the per-plane loop below: 0.63 ms per step against 0.54 for three pixel arrays)
        float part[3] = {0.0f, 0.0f, 0.0f};
        for (int32_t k = s + (int32_t)vs; k < e; k += (int32_t)(4 * VPI)) {
            uint32_t q[4];
            float    v[3][4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int32_t kk = k + u * (int32_t)VPI;
                q[u] = kk < e ? v3.xyz_by_row[kk] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t x = (q[u] & 1023u) - 1u, y = ((q[u] >> 10) & 1023u) - 1u, z = ((q[u] >> 20) & 1023u) - 1u;
                const bool     on = q[u] != 0xFFFFFFFFu;
                v[0][u] = on ? v3.G[0][((size_t)(x * v3.S + y) * F + ch) * 2 + sel] : 0.0f;
                v[1][u] = on ? v3.G[1][((size_t)(x * v3.S + z) * F + ch) * 2 + sel] : 0.0f;
                v[2][u] = on ? v3.G[2][((size_t)(y * v3.S + z) * F + ch) * 2 + sel] : 0.0f;
            }
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int u = 0; u < 4; u++) part[a] += v[a][u];
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
#pragma unroll
            for (uint32_t m = F; m < 64; m <<= 1) part[a] += __shfl_xor(part[a], (int)m);
            acc += pos ? part[a] : -part[a];          // the three planes' gradients, added in plane order
        }
    } else if (s != e) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const uint32_t* __restrict__ pbr = v3.pixels_by_row[a];
            const float* __restrict__    G = v3.G[a];
            float                        part = 0;
            for (int32_t k = s + (int32_t)vs; k < e; k += (int32_t)(4 * VPI)) {
                uint32_t px[4];
                float    v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int32_t kk = k + u * (int32_t)VPI;
                    px[u] = kk < e ? pbr[kk] : 0xFFFFFFFFu;
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    v[u] = px[u] != 0xFFFFFFFFu ? G[((size_t)px[u] * F + ch) * 2 + sel] : 0.0f;
#pragma unroll
                for (int u = 0; u < 4; u++) part += v[u];
            }
#pragma unroll
            for (uint32_t m = F; m < 64; m <<= 1) part += __shfl_xor(part, (int)m);
            acc += pos ? part : -part;          // the three planes' gradients, added in plane order
        }
    }
    if (vs == 0) grad_emb[(size_t)r * F + ch] = acc;
}

// ---- the plan straight from the occupancy grid (no vertex list, no vertex volume, no sorting by pixel) -------------
// utils_bpp_acc.py:498-512: the finest-level vertices inside / one ring around occupied cells.  Cell c of an axis covers
// the vertices c t .. c t + t + 1, so vertex u is in the set iff one of the FINE cells u - 2, u - 1, u (fine cell i =
// coarse cell i / t, 0 <= i < Rb t) is occupied on every axis, i.e. iff some coarse cell of the box A(x) x A(y) x A(z)
// is, A(u) = [(max(u, 2) - 2) / t, min(u, Rb t - 1) / t] (one to three cells).  For a LINE of vertices along one axis the
// other two boxes are fixed: the OR of the (at most 3 x 3) occupancy bit rows along the line's axis is a mask of Rb bits,
// and vertex v of the line is in the set iff the mask has a bit in A(v).  So the occupancy is bit-packed along each
// axis (3 x Rb^2 rows of 128 bits) and every pixel of every plane is a handful of bit operations on four words — no
// [R, R, R] volume (136 MB at R = 514), no vertex list.
constexpr uint32_t kOccWords = 4;            // Rb <= 128

// bits[axis][i][j][word]: axis 0 packs along z (row (x, y) = (i, j)), axis 1 along y (row (x, z)), axis 2 along x (row (y, z))
__global__ __launch_bounds__(256) void k_pack_occupancy_bits(const uint8_t* __restrict__ occ, uint32_t Rb,
                                                             uint32_t* __restrict__ bits)
{
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 3 * Rb * Rb * kOccWords) return;
    const uint32_t w = idx % kOccWords, j = (idx / kOccWords) % Rb, i = (idx / kOccWords / Rb) % Rb, axis = idx / kOccWords / Rb / Rb;
    uint32_t v = 0;
    for (uint32_t b = 0; b < 32; b++) {
        const uint32_t k = w * 32 + b;
        if (k >= Rb) break;
        const size_t at = axis == 0 ? ((size_t)i * Rb + j) * Rb + k : axis == 1 ? ((size_t)i * Rb + k) * Rb + j
                                                                               : ((size_t)k * Rb + i) * Rb + j;
        v |= (occ[at] ? 1u : 0u) << b;
    }
    bits[idx] = v;
}

struct OccLine {
    uint32_t m[kOccWords];
    __device__ __forceinline__ bool any(uint32_t lo, uint32_t hi) const       // a bit in [lo, hi] (hi - lo <= 2)?
    {
        bool r = false;
        for (uint32_t c = lo; c <= hi; c++) {
            const uint32_t w = c >> 5;
            const uint32_t word = w == 0 ? m[0] : w == 1 ? m[1] : w == 2 ? m[2] : m[3];
            r |= (word >> (c & 31u)) & 1u;
        }
        return r;
    }
};

// (t is a power of two in every configuration of the drivers — 514 = 128 x 4 + 2: a shift, not a ~25-instruction
// division twice per vertex)
__device__ __forceinline__ uint32_t div_t(uint32_t x, uint32_t t)
{
    return (t & (t - 1u)) == 0 ? x >> (31u - (uint32_t)__builtin_clz(t)) : x / t;
}

__device__ __forceinline__ void coarse_range(uint32_t u, uint32_t t, uint32_t n, uint32_t& lo, uint32_t& hi)
{
    lo = div_t(u >= 2 ? u - 2 : 0, t);
    hi = div_t(u < n ? u : n - 1, t);
}

// the line mask of pixel (u, w) of `plane` (0: xy, line along z; 1: xz, along y; 2: yz, along x)
__device__ __forceinline__ OccLine line_mask(const uint32_t* __restrict__ bits, uint32_t Rb, uint32_t t, uint32_t plane,
                                             uint32_t u, uint32_t w)
{
    const uint32_t  n = Rb * t;
    const uint32_t* base = bits + (size_t)plane * Rb * Rb * kOccWords;
    uint32_t        ulo, uhi, wlo, whi;
    coarse_range(u, t, n, ulo, uhi);
    coarse_range(w, t, n, wlo, whi);
    OccLine L{{0, 0, 0, 0}};
    for (uint32_t a = ulo; a <= uhi; a++)
        for (uint32_t b = wlo; b <= whi; b++) {
            const uint4 r = *reinterpret_cast<const uint4*>(base + ((size_t)a * Rb + b) * kOccWords);
            L.m[0] |= r.x; L.m[1] |= r.y; L.m[2] |= r.z; L.m[3] |= r.w;
        }
    return L;
}

// One wave per pixel of a plane (blockIdx.y = plane), lanes = 64 consecutive vertices of the pixel's line.
// FILL = false: counts[plane][pixel] = vertices of the set on the line (inner vertices 1 .. R-2 only: the ones
// cnt_np_embed does not skip).  FILL = true: their table rows at seg[pixel] + rank — the pixel-major order a stable sort
// by pixel of the (x, y, z)-ordered vertex list gives; the xy plane (whose order IS the list's) also writes the packed
// vertices x | y << 10 | z << 20.
template <bool FILL>
__global__ __launch_bounds__(256) void k_vote_plan_lines(const uint32_t* __restrict__ bits, uint32_t Rb, uint32_t t, uint32_t hs,
                                                         int32_t* __restrict__ counts, const int32_t* __restrict__ seg,
                                                         int32_t* __restrict__ rows_xy, int32_t* __restrict__ rows_xz,
                                                         int32_t* __restrict__ rows_yz, uint32_t* __restrict__ xyz)
{
    const uint32_t n = Rb * t, R = n + 2, S = R - 2, plane = blockIdx.y;
    const uint32_t p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (p >= S * S) return;
    const uint32_t u = p / S + 1, w = p % S + 1;
    const OccLine  L = line_mask(bits, Rb, t, plane, u, w);
    int32_t*       out = plane == 0 ? rows_xy : plane == 1 ? rows_xz : rows_yz;
    // Most lines of a projected scene are empty, and a line that is not touches a few coarse cells: coarse cell c holds the
    // vertices [c t, (c + 1) t + 1] (coarse_range read backwards), so only the span between the first and the last set bit is
    // walked — the same vertices in the same order (round 6: 0.58 + 0.64 -> see profiles/r06_refresh_step.md).
    uint32_t c_min = 0, c_max = 0;
    bool     some = false;
#pragma unroll
    for (uint32_t k = 0; k < kOccWords; k++)
        if (L.m[k]) {
            if (!some) c_min = 32 * k + (uint32_t)__builtin_ctz(L.m[k]);
            c_max = 32 * k + 31u - (uint32_t)__builtin_clz(L.m[k]);
            some = true;
        }
    if (!some) {
        if (!FILL && lane == 0) counts[(size_t)plane * S * S + p] = 0;
        return;
    }
    const uint32_t v_first = c_min * t > 1u ? c_min * t : 1u, v_last = (c_max + 1) * t + 1 < S ? (c_max + 1) * t + 1 : S;
    uint32_t       at = FILL ? (uint32_t)seg[(size_t)plane * (S * S + 1) + p] : 0u;
    for (uint32_t v0 = v_first; v0 <= v_last; v0 += 64) {
        const uint32_t v = v0 + lane;
        bool           in = false;
        if (v <= S) {
            uint32_t lo, hi;
            coarse_range(v, t, n, lo, hi);
            in = L.any(lo, hi);
        }
        const uint64_t b = __ballot(in);
        if (FILL && in) {
            uint32_t q[3];
            if (plane == 0) { q[0] = u; q[1] = w; q[2] = v; }
            else if (plane == 1) { q[0] = u; q[1] = v; q[2] = w; }
            else { q[0] = v; q[1] = u; q[2] = w; }
            const uint32_t k = at + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
            out[k] = (int32_t)grid_row<3>(q, hs, R);
            if (plane == 0) xyz[k] = q[0] | q[1] << 10 | q[2] << 20;
        }
        at += (uint32_t)__popcll(b);
    }
    if (!FILL && lane == 0) counts[(size_t)plane * S * S + p] = (int32_t)at;
}

}  // namespace cnc

using namespace cnc;

extern "C" int cnc_cnt_np_plan(const int16_t* inputs, uint32_t N, uint32_t resolution,
                               uint32_t hashmap_size, uint32_t axis, uint32_t* rows,
                               uint32_t* pixels, void* stream)
{
    if (N == 0) return CNC_OK;
    if (!inputs || (!rows && !pixels) || axis > 2 || resolution < 3 || hashmap_size == 0)
        return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_cnt_plan, dim3(div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, inputs, N,
                       resolution, hashmap_size, axis, rows, pixels);
    return launch_status();
}

extern "C" int cnc_cnt_np_plan3(const int16_t* inputs, uint32_t N, uint32_t resolution, uint32_t hashmap_size,
                                int32_t* rows, int32_t* pix_xy, int32_t* pix_xz, int32_t* pix_yz,
                                int32_t* unsorted, void* stream)
{
    if (N == 0) return CNC_OK;
    if (!inputs || !rows || !pix_xy || !pix_xz || !pix_yz || !unsorted || resolution < 3 || hashmap_size == 0 ||
        hashmap_size > 0x7fffffffu || (uint64_t)(resolution - 2) * (resolution - 2) > 0x7fffffffu)
        return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_cnt_plan3, dim3(div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, inputs, N, resolution,
                       hashmap_size, rows, pix_xy, pix_xz, pix_yz, unsorted);
    return launch_status();
}

extern "C" int cnc_vote_plan_count(const uint8_t* occupancy, uint32_t Rb, uint32_t t, uint32_t* bits, int32_t* counts,
                                   void* stream)
{
    if (!occupancy || !bits || !counts || Rb == 0 || Rb > 32 * kOccWords || t == 0 || (uint64_t)Rb * t + 2 > 1024)
        return CNC_ERR_INVALID_VALUE;
    hipStream_t    s = (hipStream_t)stream;
    const uint32_t S = Rb * t;
    hipLaunchKernelGGL(k_pack_occupancy_bits, dim3(div_up(3 * Rb * Rb * kOccWords, 256)), dim3(256), 0, s, occupancy, Rb, bits);
    hipLaunchKernelGGL((k_vote_plan_lines<false>), dim3(div_up(S * S, 4), 3), dim3(256), 0, s, bits, Rb, t, 1u, counts, nullptr,
                       nullptr, nullptr, nullptr, nullptr);
    return launch_status();
}

extern "C" int cnc_vote_plan_fill(const uint32_t* bits, uint32_t Rb, uint32_t t, uint32_t hashmap_size, const int32_t* seg,
                                  int32_t* rows_xy, int32_t* rows_xz, int32_t* rows_yz, uint32_t* xyz, void* stream)
{
    if (!bits || !seg || !rows_xy || !rows_xz || !rows_yz || !xyz || Rb == 0 || Rb > 32 * kOccWords || t == 0 ||
        (uint64_t)Rb * t + 2 > 1024 || hashmap_size == 0 || hashmap_size > 0x7fffffffu)
        return CNC_ERR_INVALID_VALUE;
    const uint32_t S = Rb * t;
    hipLaunchKernelGGL((k_vote_plan_lines<true>), dim3(div_up(S * S, 4), 3), dim3(256), 0, (hipStream_t)stream, bits, Rb, t,
                       hashmap_size, nullptr, seg, rows_xy, rows_xz, rows_yz, xyz);
    return launch_status();
}

#define CNC_VOTE_SWITCH(F, CALL)                                  \
    switch (F) {                                                  \
    case 1: { constexpr uint32_t FF = 1; CALL; } break;           \
    case 2: { constexpr uint32_t FF = 2; CALL; } break;           \
    case 4: { constexpr uint32_t FF = 4; CALL; } break;           \
    case 8: { constexpr uint32_t FF = 8; CALL; } break;           \
    case 16: { constexpr uint32_t FF = 16; CALL; } break;         \
    case 32: { constexpr uint32_t FF = 32; CALL; } break;         \
    default: return CNC_ERR_INVALID_VALUE;                        \
    }

extern "C" int cnc_cnt_np_embed_planned(const uint32_t* rows_by_pixel, const int32_t* pixel_seg,
                                        const float* embeddings_clip, float* outputs,
                                        uint32_t n_pixels, uint32_t F, void* stream)
{
    if (n_pixels == 0) return CNC_OK;
    if (!pixel_seg || !embeddings_clip || !outputs) return CNC_ERR_INVALID_VALUE;
    hipStream_t s = (hipStream_t)stream;
    CNC_VOTE_SWITCH(F, hipLaunchKernelGGL((k_cnt_votes<FF>), dim3(n_pixels), dim3(64), 0, s,
                                          rows_by_pixel, pixel_seg, embeddings_clip, outputs, n_pixels));
    return launch_status();
}

extern "C" int cnc_cnt_vote_masks(const float* embeddings_clip, uint32_t n_rows, uint32_t F, uint32_t* masks, void* stream)
{
    if (n_rows == 0) return CNC_OK;
    if (!embeddings_clip || !masks || F == 0 || F > 32) return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_vote_masks, dim3(div_up(n_rows, 256)), dim3(256), 0, (hipStream_t)stream, embeddings_clip, n_rows,
                       F, masks);
    return launch_status();
}

extern "C" int cnc_cnt_np_embed_planned_masked(const uint32_t* rows_by_pixel, const int32_t* pixel_seg,
                                               const uint32_t* masks, float* outputs, uint32_t n_pixels, uint32_t F,
                                               void* stream)
{
    if (n_pixels == 0) return CNC_OK;
    if (!pixel_seg || !masks || !outputs || F == 0 || F > 32) return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_cnt_votes_masked, dim3(n_pixels), dim3(64), 0, (hipStream_t)stream, rows_by_pixel, pixel_seg, masks,
                       outputs, n_pixels, F);
    return launch_status();
}

extern "C" int cnc_cnt_np_embed_planned_backward(const uint32_t* pixels_by_row, const int32_t* row_seg,
                                                 const float* embeddings_clip, const float* grad_over_sum,
                                                 float* grad_embeddings, uint32_t n_rows, uint32_t F,
                                                 void* stream)
{
    if (n_rows == 0) return CNC_OK;
    if (!row_seg || !embeddings_clip || !grad_over_sum || !grad_embeddings) return CNC_ERR_INVALID_VALUE;
    hipStream_t s = (hipStream_t)stream;
    CNC_VOTE_SWITCH(F, hipLaunchKernelGGL((k_cnt_votes_bwd<FF>), dim3(n_rows), dim3(64), 0, s,
                                          pixels_by_row, row_seg, embeddings_clip, grad_over_sum,
                                          grad_embeddings, n_rows));
    return launch_status();
}

namespace cnc {
// cnt [S, S, F, 2] -> table [R, R, F] (R = S + 2): the +1 fraction of every inner pixel, a ring of zero pixels around
// it; sums [S, S, F] = (cnt0 + cnt1) + 1e-6.  The reference spells it sum(-1, keepdim) + 1e-6, a division, a select of
// channel 0, two permutes and a pad (utils_bpp_acc.py:39-55, 515-526): five library launches per plane, one of them a
// reduction over an axis of length two that took 0.15 ms.
__global__ __launch_bounds__(256) void k_vote_fraction_table(const float* __restrict__ cnt, uint32_t S, uint32_t F,
                                                             float* __restrict__ table, float* __restrict__ sums)
{
    const uint32_t R = S + 2;
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;            // 32-bit index arithmetic (the entry checks the size)
    if (e >= R * R * F) return;
    const uint32_t f = e % F, px = e / F;
    const uint32_t u = px / R, w = px % R;
    float v = 0.0f;
    if (u >= 1 && u <= S && w >= 1 && w <= S) {
        const size_t  at = ((size_t)(u - 1) * S + (w - 1)) * F + f;
        const float2  c = *reinterpret_cast<const float2*>(cnt + at * 2);
        const float   sm = (c.x + c.y) + 1e-6f;
        sums[at] = sm;
        v = c.x / sm;
    }
    table[e] = v;
}

// g_table [R, R, F] -> grad_over_sum [S, S, F, 2] = [(1 / sums) * g, 0]: what cnt_np_embed_backward consumes
// (gridencoder.cu:1035-1040); channel 1 (the -1 votes) carries no gradient because only channel 0 is used downstream
__global__ __launch_bounds__(256) void k_vote_fraction_table_bwd(const float* __restrict__ g_table,
                                                                 const float* __restrict__ sums, uint32_t S, uint32_t F,
                                                                 float* __restrict__ grad_over_sum)
{
    const uint32_t R = S + 2;
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= S * S * F) return;
    const uint32_t f = e % F, px = e / F;
    const uint32_t u = px / S, w = px % S;
    const float    g = g_table[((size_t)(u + 1) * R + (w + 1)) * F + f];
    *reinterpret_cast<float2*>(grad_over_sum + e * 2) = make_float2((1.0f / sums[e]) * g, 0.0f);
}
}  // namespace cnc

extern "C" int cnc_vote_fraction_table(const float* cnt, uint32_t S, uint32_t F, float* table, float* sums, void* stream)
{
    if (S == 0 || F == 0) return CNC_OK;
    if (!cnt || !table || !sums) return CNC_ERR_INVALID_VALUE;
    const uint64_t n = (uint64_t)(S + 2) * (S + 2) * F;
    if (n >= (1ull << 31)) return CNC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_vote_fraction_table, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cnt, S, F,
                       table, sums);
    return launch_status();
}

extern "C" int cnc_vote_fraction_table_backward(const float* g_table, const float* sums, uint32_t S, uint32_t F,
                                                float* grad_over_sum, void* stream)
{
    if (S == 0 || F == 0) return CNC_OK;
    if (!g_table || !sums || !grad_over_sum) return CNC_ERR_INVALID_VALUE;
    const uint64_t n = (uint64_t)S * S * F;
    if (n >= (1ull << 31)) return CNC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_vote_fraction_table_bwd, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g_table,
                       sums, S, F, grad_over_sum);
    return launch_status();
}

extern "C" int cnc_cnt_np_embed_planned_backward3(const uint32_t* pixels_by_row_xy, const uint32_t* pixels_by_row_xz,
                                                  const uint32_t* pixels_by_row_yz, const int32_t* row_seg,
                                                  const float* embeddings_clip, const float* grad_over_sum_xy,
                                                  const float* grad_over_sum_xz, const float* grad_over_sum_yz,
                                                  float* grad_embeddings, uint32_t n_rows, uint32_t F, void* stream)
{
    if (n_rows == 0) return CNC_OK;
    if (!row_seg || !embeddings_clip || !grad_over_sum_xy || !grad_over_sum_xz || !grad_over_sum_yz || !grad_embeddings)
        return CNC_ERR_INVALID_VALUE;
    hipStream_t s = (hipStream_t)stream;
    Votes3      v3{{pixels_by_row_xy, pixels_by_row_xz, pixels_by_row_yz}, {grad_over_sum_xy, grad_over_sum_xz, grad_over_sum_yz},
               nullptr, 0};
    CNC_VOTE_SWITCH(F, hipLaunchKernelGGL((k_cnt_votes_bwd3<FF>), dim3(n_rows), dim3(64), 0, s, v3, row_seg,
                                          embeddings_clip, grad_embeddings, n_rows));
    return launch_status();
}

extern "C" int cnc_cnt_np_embed_planned_backward3_xyz(const uint32_t* xyz_by_row, const int32_t* row_seg,
                                                      const float* embeddings_clip, const float* grad_over_sum_xy,
                                                      const float* grad_over_sum_xz, const float* grad_over_sum_yz,
                                                      float* grad_embeddings, uint32_t n_rows, uint32_t F,
                                                      uint32_t resolution, void* stream)
{
    if (n_rows == 0) return CNC_OK;
    // xyz_by_row may be NULL: the plan of an all-empty occupancy grid has no vertices (every row segment is empty and the
    // kernel then reads nothing through it) — the forward and the three-array backward accept that too
    if (!row_seg || !embeddings_clip || !grad_over_sum_xy || !grad_over_sum_xz || !grad_over_sum_yz ||
        !grad_embeddings || resolution < 3 || resolution > 1024)
        return CNC_ERR_INVALID_VALUE;
    hipStream_t s = (hipStream_t)stream;
    Votes3      v3{{nullptr, nullptr, nullptr}, {grad_over_sum_xy, grad_over_sum_xz, grad_over_sum_yz}, xyz_by_row, resolution - 2};
    CNC_VOTE_SWITCH(F, hipLaunchKernelGGL((k_cnt_votes_bwd3<FF, true>), dim3(n_rows), dim3(64), 0, s, v3, row_seg,
                                          embeddings_clip, grad_embeddings, n_rows));
    return launch_status();
}
