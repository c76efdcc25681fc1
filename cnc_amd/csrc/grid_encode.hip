// grid_encode.hip — multiresolution hash-grid encoder for gfx950 (MI355X).
//
// Stands in for gridencoder/src/gridencoder.cu of the reference (kernel_grid :96-396,
// kernel_grid_backward :399-585, cnt_np_embed{,_backward} :873-1087) behind the C ABI of
// include/cnc_hip.h.  Design notes (docs/engineering_log.md §4.1-4.2):
//   * A table row is F floats.  A row is fetched by G = F/V adjacent lanes, V = min(F,4) floats
//     (<= 16 B, one global_load_dwordx4) per lane, so one wave-instruction touches 64/G rows and
//     the G lanes of a row land in one 16B-aligned span of a single cache line.  No lane ever
//     needs another lane's data: each lane owns V output features for all 2^D corners.
//   * Level-major launch (blockIdx.y = level slot) as in the reference, so at any moment the chip
//     is gathering from one level's table (<= 16 MiB at T=2^19, F=8): coarse levels live in L2,
//     fine ones in the 256 MiB Infinity Cache.  outputs are [L, N, F]: a wave stores 1 KiB
//     contiguous.
//   * All 2^D row addresses are formed first, then all gathers are issued back-to-back (8
//     independent dwordx4 loads in flight per lane), then the weighted sum runs in the
//     reference's corner order so results are bit-identical to the CPU oracle.
//   * Optional STE fusion (CNC_FLAG_STE_BINARY): sign() is applied to the gathered values, which
//     removes the reference's separate full-table STE_binary passes (ngp.py:22-39,244-245).
#include "common.hpp"
#include <cstdlib>

#include "encoder_common.hpp"

namespace cnc {

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <uint32_t D, uint32_t F, bool VXL, bool STE>
__global__ __launch_bounds__(256) void k_grid_encode_fwd(
    const float* __restrict__ inputs, const float* __restrict__ emb,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ resolutions,
    float* __restrict__ out, uint32_t N, uint32_t Rb, const uint8_t* __restrict__ vxl,
    const int32_t* __restrict__ min_level_id, const int32_t* __restrict__ sat, FeatLayout lay)
{
    constexpr uint32_t V = F < 4 ? F : 4;
    constexpr uint32_t G = F / V;
    constexpr uint32_t C = 1u << D;

    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = t / G;
    const uint32_t h = t % G;
    if (b >= N) return;
    const uint32_t slot = blockIdx.y;
    const uint32_t level = slot + (min_level_id ? (uint32_t)min_level_id[b] : 0u);

    float* o = out + feat_index(lay, slot, N, b, F) + h * V;
    float  acc[V];
#pragma unroll
    for (uint32_t k = 0; k < V; k++) acc[k] = 0;

    float x[D];
    if (!load_point<D>(inputs, b, x)) {   // out of [0,1]: zeros (gridencoder.cu:134-158)
        store_vec<V>(o, acc);
        return;
    }
    const uint32_t off = (uint32_t)offsets[level];
    const uint32_t hs = (uint32_t)offsets[level + 1] - off;
    const uint32_t R = (uint32_t)resolutions[level];
    const float* table = emb + (size_t)off * F + h * V;

    Corners<D, VXL> c;
    c.setup(x, R, hs, Rb, vxl, sat, vertex_plane(lay, level));

    float v[C][V];
#pragma unroll
    for (uint32_t i = 0; i < C; i++) {
        if (c.valid[i]) {
            load_vec<V>(table + (size_t)c.row[i] * F, v[i]);
        } else {
#pragma unroll
            for (uint32_t k = 0; k < V; k++) v[i][k] = 0;
        }
    }
#pragma unroll
    for (uint32_t i = 0; i < C; i++) {
        const float tw = c.w[i] * c.wn_re;
#pragma unroll
        for (uint32_t k = 0; k < V; k++) {
            float e = v[i][k];
            if constexpr (STE) e = (e >= 0) ? 1.0f : -1.0f;
            acc[k] = c.valid[i] ? __builtin_fmaf(tw, e, acc[k]) : acc[k];
        }
    }
    store_vec<V>(o, acc);
}

// ---------------------------------------------------------------------------------------------
// bit-plane forward: the binarised table as F bits per row.  One lane per point owns all F
// output features; a corner costs one 1/2/4-byte load from a table that is 32x smaller than the
// fp32 one (a whole level is <= 512 KiB at F=8: L2-resident), so the kernel is bound by its own
// output stream (L*F*4 bytes per sample).  Same corner order and fmaf chain as the fp32 kernel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_sign_bits(const float* __restrict__ emb,
                                                        uint8_t* __restrict__ bits,
                                                        uint64_t n_bytes, uint64_t n_vals,
                                                        uint32_t* __restrict__ clip_count)
{
    uint32_t clipped = 0;
    // one lane builds one output byte from 8 consecutive floats (two dwordx4 loads, coalesced)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_bytes;
         i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t b = 0;
        const uint64_t base = i * 8;
        if (base + 8 <= n_vals) {
            const float4 lo = *reinterpret_cast<const float4*>(emb + base);
            const float4 hi = *reinterpret_cast<const float4*>(emb + base + 4);
            b = (lo.x >= 0 ? 1u : 0u) | (lo.y >= 0 ? 2u : 0u) | (lo.z >= 0 ? 4u : 0u) |
                (lo.w >= 0 ? 8u : 0u) | (hi.x >= 0 ? 16u : 0u) | (hi.y >= 0 ? 32u : 0u) |
                (hi.z >= 0 ? 64u : 0u) | (hi.w >= 0 ? 128u : 0u);
            const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int k = 0; k < 8; k++) clipped += !(v[k] >= -1.0f && v[k] <= 1.0f);
        } else {
            for (uint32_t k = 0; k < 8 && base + k < n_vals; k++) {
                const float v = emb[base + k];
                b |= (v >= 0 ? 1u : 0u) << k;
                clipped += !(v >= -1.0f && v <= 1.0f);
            }
        }
        bits[i] = (uint8_t)b;
    }
    if (clip_count != nullptr && __any(clipped != 0)) {
        // rare: one integer atomic per lane that saw an out-of-range entry
        if (clipped) atomicAdd(clip_count, clipped);
    }
}

// P = level slots per lane.  Level-major output ([L, N, F], the drop-in layout): P = 1.  Point-major output (rows of
// a wider [N, ld] matrix, lay.ld != 0): P * F = 16 floats, so that a point's piece of the row is a whole 64-byte line.
//
// Stores.  A lane owns a point, i.e. P*F*4 = 32 or 64 bytes of output; stored straight from its registers, a wave's
// store instruction writes 16 bytes per lane at a 32 / 64-byte (level-major) or `ld`-float (point-major) stride: every
// instruction touches a quarter or a half of each 64-byte line, the halves reach the fabric as separate partial write
// requests (round 3: 11.7 M requests and WRITE_SIZE 735 MB for a 537 MB output; point-major 2x the level-major time).
// TR = true: the wave transposes its 64 x (P*F) tile through LDS (swizzled at 16-byte granularity: conflict-free
// ds_write_b128 / ds_read_b128), after which lane i of store s holds the 16-byte chunk s*64 + i of the tile in memory
// order: level-major a wave's instruction writes 1 KB contiguous, point-major every instruction writes whole lines
// (64 / Cp points x Cp*16 bytes).  The arithmetic is untouched (same corner order, same fmaf chain).
template <uint32_t D, uint32_t F, bool VXL, bool TR>
#ifndef CNC_FWD_BITS_BOUNDS
#define CNC_FWD_BITS_BOUNDS __launch_bounds__(256)
#endif
__global__ CNC_FWD_BITS_BOUNDS void k_grid_encode_fwd_bits(
    const float* __restrict__ inputs, const uint8_t* __restrict__ bits,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ resolutions,
    float* __restrict__ out, uint32_t N, uint32_t L, uint32_t P, uint32_t cp_log2, uint32_t Rb,
    const uint8_t* __restrict__ vxl, const int32_t* __restrict__ min_level_id, const int32_t* __restrict__ sat,
    FeatLayout lay)
{
    constexpr uint32_t C = 1u << D;
    constexpr uint32_t V = F < 4 ? F : 4;
    extern __shared__ float s_tile[];            // TR only: [4 waves][64 points][Cp chunks of 4 floats], swizzled
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t Cp = 1u << cp_log2;           // 16-byte chunks per point and pass
    float* tile = s_tile + (size_t)(threadIdx.x >> 6) * 64 * Cp * 4;
    const uint32_t swz = (lane >> (3u - cp_log2)) & (Cp - 1u);
    if constexpr (!TR) {
        if (b >= N) return;
    }
    float      x[D];
    const bool inside = b < N && load_point<D>(inputs, b, x);
    const uint32_t first = (min_level_id && b < N) ? (uint32_t)min_level_id[b] : 0u;
    for (uint32_t pl = 0; pl < P; pl++) {
        const uint32_t slot = blockIdx.y * P + pl;
        if (slot >= L) break;
        const uint32_t level = slot + first;
        float  acc[F];
#pragma unroll
        for (uint32_t k = 0; k < F; k++) acc[k] = 0;
        // The lean evaluator (encoder_common.hpp, unit_features_fast: per-axis index parts, no per-corner branch, x-pair
        // 16-bit gathers, shift + bfi signs — the same values bit for bit) whenever the level is the same for the whole
        // grid row (no per-point windows) and is what a GridEncoder makes: dense (R^D rows fit) or hashed into a
        // power-of-two table.  Anything else — odd caller-made tables, the occupancy mask — takes the general path below.
        bool done = false;
        if constexpr (!VXL && (D == 2 || D == 3)) {
            if (min_level_id == nullptr) {
                const uint32_t off = (uint32_t)offsets[slot];
                const uint32_t hs = (uint32_t)offsets[slot + 1] - off;
                const uint32_t R = (uint32_t)resolutions[slot];
                uint64_t rd = 1;
#pragma unroll
                for (uint32_t d = 0; d < D; d++) rd *= R;
                const bool dense = rd <= hs, pow2 = (hs & (hs - 1u)) == 0u;
                if ((dense || pow2) && (off & 7u) == 0u) {
                    unit_features_fast<D, F>(x, inside, bits, UnitRec{off, hs, R, 0u}, acc);
                    done = true;
                }
            }
        }
        if (inside && !done) {
            const uint32_t off = (uint32_t)offsets[level];
            const uint32_t hs = (uint32_t)offsets[level + 1] - off;
            const uint32_t R = (uint32_t)resolutions[level];
            Corners<D, VXL> c;
            c.setup(x, R, hs, Rb, vxl, sat, vertex_plane(lay, level));
            uint32_t rb[C];
#pragma unroll
            for (uint32_t i = 0; i < C; i++)
                rb[i] = c.valid[i] ? load_row_bits<F>(bits, (uint64_t)off + c.row[i]) : 0u;
#pragma unroll
            for (uint32_t i = 0; i < C; i++) {
                // an invalid corner gets weight +0: fmaf(+0, +-1, acc) == acc, no per-feature select
                const float tw = c.valid[i] ? c.w[i] * c.wn_re : 0.0f;
#pragma unroll
                for (uint32_t k = 0; k < F; k++) {
                    const float e = ((rb[i] >> k) & 1u) ? 1.0f : -1.0f;
                    acc[k] = __builtin_fmaf(tw, e, acc[k]);
                }
            }
        }
        float* o = TR ? nullptr : out + feat_index(lay, slot, N, b, F);
#pragma unroll
        for (uint32_t k = 0; k < F; k += V) {
            float v[V];
#pragma unroll
            for (uint32_t j = 0; j < V; j++) v[j] = acc[k + j];
            if constexpr (TR) {
                const uint32_t f = pl * F + k;                      // float index inside the point's piece
                store_vec<V>(tile + ((lane << cp_log2) + ((f >> 2) ^ swz)) * 4 + (f & 3u), v);
            } else {
                if (lay.nt) store_vec_nt<V>(o + k, v);
                else store_vec<V>(o + k, v);
            }
        }
    }
    if constexpr (TR) {
        __syncthreads();
        const uint32_t slot0 = blockIdx.y * P;
        const uint32_t n_floats = min(P, L - slot0) * F;           // of this pass, per point (a multiple of 4)
        const uint32_t b0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u);    // the wave's first point
        for (uint32_t s = 0; s < Cp; s++) {
            const uint32_t g = s * 64 + lane;
            const uint32_t pt = g >> cp_log2, c = g & (Cp - 1u);
            const uint32_t bp = b0 + pt;
            if (bp >= N || c * 4 >= n_floats) continue;
            const uint32_t psw = (pt >> (3u - cp_log2)) & (Cp - 1u);
            float v[4];
            load_vec<4>(tile + ((pt << cp_log2) + (c ^ psw)) * 4, v);
            float* o = lay.ld ? out + (size_t)bp * lay.ld + lay.col + slot0 * F + c * 4
                              : out + ((size_t)slot0 * N + bp) * F + c * 4;
            if (lay.nt) store_vec_nt<4>(o, v);
            else store_vec<4>(o, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward, generic fallback (row wider than a wave: C*F > 64): one lane per (point, 16-B chunk),
// hardware fp32 atomics (global_atomic_add_f32, no CAS loop).
// ---------------------------------------------------------------------------------------------
template <uint32_t D, uint32_t F, bool VXL, bool STE>
__global__ __launch_bounds__(256) void k_grid_encode_bwd_simple(
    const float* __restrict__ grad, const float* __restrict__ inputs,
    const float* __restrict__ emb, const int32_t* __restrict__ offsets,
    const int32_t* __restrict__ resolutions, float* __restrict__ grad_emb, uint32_t N,
    uint32_t Rb, const uint8_t* __restrict__ vxl, const int32_t* __restrict__ min_level_id,
    const uint32_t* __restrict__ clip_count, const int32_t* __restrict__ sat, FeatLayout lay)
{
    constexpr uint32_t V = F < 4 ? F : 4;
    constexpr uint32_t G = F / V;
    constexpr uint32_t C = 1u << D;
    const bool mask_on = STE && (clip_count == nullptr || *clip_count != 0);

    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = t / G;
    const uint32_t h = t % G;
    if (b >= N) return;
    const uint32_t slot = blockIdx.y;
    const uint32_t level = slot + (min_level_id ? (uint32_t)min_level_id[b] : 0u);

    float x[D];
    if (!load_point<D>(inputs, b, x)) return;   // gridencoder.cu:435-440

    float g[V];
    load_vec<V>(grad + feat_index(lay, slot, N, b, F) + h * V, g);

    const uint32_t off = (uint32_t)offsets[level];
    const uint32_t hs = (uint32_t)offsets[level + 1] - off;
    const uint32_t R = (uint32_t)resolutions[level];
    const size_t base = (size_t)off * F + h * V;

    Corners<D, VXL> c;
    c.setup(x, R, hs, Rb, vxl, sat, vertex_plane(lay, level));

#pragma unroll
    for (uint32_t i = 0; i < C; i++) {
        if (!c.valid[i]) continue;
        const float  tw = c.w[i] * c.wn_re;
        const size_t at = base + (size_t)c.row[i] * F;
        float keep[V];
#pragma unroll
        for (uint32_t k = 0; k < V; k++) keep[k] = 1.f;
        if (mask_on) {   // STE_binary.backward: pass gradient only where |param| <= 1
            float e[V];
            load_vec<V>(emb + at, e);
#pragma unroll
            for (uint32_t k = 0; k < V; k++) keep[k] = (e[k] >= -1.0f && e[k] <= 1.0f) ? 1.f : 0.f;
        }
#pragma unroll
        for (uint32_t k = 0; k < V; k++) {
            if (keep[k] == 0.f) continue;
            unsafeAtomicAdd(grad_emb + at + k, tw * g[k]);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// backward, run-aggregated scatter (the hot path).
//
// Measured on MI355X (tools/atomic_probe.hip): the L2 atomic path retires ~21 G requests/s, where
// a request is one (wave-instruction, 64-byte-aligned segment) pair, independent of how many of
// its 16 dwords are touched; same-address lanes inside one instruction serialise.  So the cost of
// the scatter is the number of row-requests, and the kernel is built to minimise that:
//   * phase A (lane = point): corner set-up once per (point, level); weights, gradient row,
//     absolute table rows and the cell key go to LDS.
//   * consecutive points that fall in the same cell of this level (ray-marched samples do, for
//     all but the finest levels) form a run; runs are found with a ballot prefix over head flags.
//   * phase B (lane = (corner, feature), C*F lanes per run): the run's C x F gradient block
//     sum_p w_p[c] * g_p[f] is accumulated from LDS (broadcast reads), then ONE atomic
//     instruction updates all C rows x F features: 8 lanes per 32-B row, and the two x-neighbour
//     rows sit on adjacent lanes so they share a request whenever they share a 64-B segment.
// ---------------------------------------------------------------------------------------------
template <uint32_t D, uint32_t F, bool VXL, bool STE>
__global__ __launch_bounds__(256) void k_grid_encode_bwd(
    const float* __restrict__ grad, const float* __restrict__ inputs,
    const float* __restrict__ emb, const int32_t* __restrict__ offsets,
    const int32_t* __restrict__ resolutions, float* __restrict__ grad_emb, uint32_t N,
    uint32_t Rb, const uint8_t* __restrict__ vxl, const int32_t* __restrict__ min_level_id,
    const uint32_t* __restrict__ clip_count, const int32_t* __restrict__ sat, FeatLayout lay)
{
    constexpr uint32_t C = 1u << D;
    constexpr uint32_t SLOTS = C * F;           // lanes per run in phase B (<= 64)
    // STE mask needed only if some parameter left [-1, 1] (counted by cnc_pack_sign_bits)
    const bool mask_on = STE && (clip_count == nullptr || *clip_count != 0);
    constexpr uint32_t GROUPS = 256 / SLOTS;    // runs processed concurrently by the block
    constexpr uint32_t V = F < 4 ? F : 4;
    static_assert(SLOTS <= 64, "use k_grid_encode_bwd_simple");

    __shared__ float    s_tw[256][C];
    __shared__ float    s_g[256][F];
    // with one run per wave (SLOTS == 64) the table rows are recomputed in phase B from the cell in
    // s_key: without the 8 KB of rows the block needs 19.5 KB of LDS and 8 instead of 5 blocks fit a
    // CU — the kernel is latency-bound (2 resident blocks: 1.6x slower), not issue-bound
    constexpr bool kRowsFromKey = SLOTS == 64;
    __shared__ uint32_t s_row[kRowsFromKey ? 1 : 256][C];
    __shared__ uint64_t s_key[256];
    __shared__ uint16_t s_run_start[257];
    __shared__ uint8_t  s_valid[256];
    __shared__ uint32_t s_wave_heads[4];

    const uint32_t tid = threadIdx.x;
    // Block -> (chunk of 256 points, level slot).  With the level as the FAST index the blocks
    // resident at any moment are spread over all levels instead of sharing one: the blocks of one
    // level hit the same table rows (every ray crosses the same coarse cells; neighbouring chunks
    // are the same ray), and the memory-side atomic units serialise same-address updates — with 16
    // private copies of the table the 9 coarse levels ran in 0.44 instead of 0.58 ms.
    const uint32_t n_slots = lay.n_slots ? lay.n_slots : gridDim.y;
    const uint32_t chunk = lay.n_slots ? blockIdx.x / n_slots : blockIdx.x;
    const uint32_t slot_raw = lay.n_slots ? blockIdx.x % n_slots : blockIdx.y;
    const uint32_t b = chunk * 256 + tid;
    // CNC_FLAG_LEVELS_FINEST_FIRST: the levels with the most atomic requests start first and the cheap
    // coarse ones fill the tail of the launch
    const uint32_t slot = lay.finest_first ? n_slots - 1 - slot_raw : slot_raw;

    // ---- phase A ----
    uint64_t key = ~0ull;   // out-of-range / padding points: no contribution
    uint32_t validmask = 0;
    {
        float x[D];
        bool  in_range = false;
        if (b < N) in_range = load_point<D>(inputs, b, x);
        if (in_range) {
            const uint32_t level = slot + (min_level_id ? (uint32_t)min_level_id[b] : 0u);
            const uint32_t off = (uint32_t)offsets[level];
            const uint32_t hs = (uint32_t)offsets[level + 1] - off;
            const uint32_t R = (uint32_t)resolutions[level];
            Corners<D, VXL> c;
            c.setup(x, R, hs, Rb, vxl, sat, vertex_plane(lay, level));
            uint64_t cell = 0;   // integer cell coordinates, 16 bits per axis (R <= 65535)
#pragma unroll
            for (uint32_t d = D; d-- > 0;) {
                float p = x[d] * (float)(R - 2);
                p = p + 0.5f;
                cell = (cell << 16) | (uint32_t)floorf(p);
            }
            key = ((uint64_t)level << 52) | cell;
#pragma unroll
            for (uint32_t i = 0; i < C; i++) {
                s_tw[tid][i] = c.valid[i] ? c.w[i] * c.wn_re : 0.0f;
                if constexpr (!kRowsFromKey) s_row[tid][i] = off + c.row[i];
                validmask |= (c.valid[i] ? 1u : 0u) << i;
            }
            const float* gp = grad + feat_index(lay, slot, N, b, F);
            bool         nonzero = false;
#pragma unroll
            for (uint32_t k = 0; k < F; k += V) {
                float gv[V];
                load_vec<V>(gp + k, gv);
#pragma unroll
                for (uint32_t j = 0; j < V; j++) {
                    s_g[tid][k + j] = gv[j];
                    nonzero |= gv[j] != 0.0f;
                }
            }
            // a point whose gradient row is all zeros adds nothing: no atomics for it (samples behind an opaque surface;
            // the levels outside a vertex's context window when several windows share one call, context.py _plane_bits)
            if (!nonzero) {
                key = ~0ull;
                validmask = 0;
            }
        }
        s_key[tid] = key;
        s_valid[tid] = (uint8_t)validmask;
    }
    __syncthreads();

    // ---- run detection: head = first point of a run of equal (level, cell) keys ----
    const uint64_t prev_key = tid == 0 ? ~0ull : s_key[tid - 1];
    const bool     head = (tid == 0) || (prev_key != key);
    // a head whose cell touches the previous point's cell (same level, every axis within +-1)
    // shares corner rows with it: count those to decide whether write-combining pays
    bool adjacent = false;
    if (head && key != ~0ull && prev_key != ~0ull && (key >> 52) == (prev_key >> 52)) {
        adjacent = true;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            const int a = (int)((key >> (16 * d)) & 0xFFFF), b2 = (int)((prev_key >> (16 * d)) & 0xFFFF);
            adjacent &= (a - b2 <= 1) && (b2 - a <= 1);
        }
    }
    const uint64_t hb = __ballot(head);
    const uint64_t ab = __ballot(adjacent);
    const uint32_t lane = tid & 63, wave = tid >> 6;
    if (lane == 0) s_wave_heads[wave] = (uint32_t)__popcll(hb) | ((uint32_t)__popcll(ab) << 16);
    __syncthreads();
    uint32_t before = 0, total = 0, n_adjacent = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
        const uint32_t h = s_wave_heads[w] & 0xFFFFu;
        before += (w < wave) ? h : 0u;
        total += h;
        n_adjacent += s_wave_heads[w] >> 16;
    }
    if (head) s_run_start[before + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull))] = (uint16_t)tid;
    if (tid == 0) s_run_start[total] = 256;
    __syncthreads();

    // ---- phase B ----
    // Each group of SLOTS lanes walks a CONTIGUOUS range of runs and keeps the previous run's
    // (row, sum) pending in registers: a ray leaves a cell through a face, so the next run shares
    // up to half of its corner rows with the pending one.  Rows of the pending run that reappear
    // are absorbed by the new run instead of being written (write-combining across adjacent
    // cells); only rows that do not reappear go out as atomics.
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    const uint32_t grp = tid / SLOTS, l = tid % SLOTS;
    const uint32_t c = l / F, f = l % F;

    auto flush = [&](uint32_t row, float v) {
        const size_t at = (size_t)row * F + f;
        if (mask_on) {   // STE_binary.backward: pass gradient only where |param| <= 1
            const float e = emb[at];
            if (!(e >= -1.0f && e <= 1.0f)) return;
        }
        unsafeAtomicAdd(grad_emb + at, v);
    };

    // absolute table row of corner c of the cell in `k` (the same arithmetic as Corners::setup)
    uint32_t u_off = 0, u_hs = 1, u_R = 2;
    if (!min_level_id) {
        u_off = (uint32_t)offsets[slot];
        u_hs = (uint32_t)offsets[slot + 1] - u_off;
        u_R = (uint32_t)resolutions[slot];
    }
    // Level geometry as wave-uniform scalars when the whole block works on one level (no per-point
    // level window): the dense / hashed decision and the modulo form of grid_row are then scalar
    // branches, and the row costs ~10 vector instructions instead of ~70 (it is evaluated once per
    // run by all 64 lanes, not once per 64 samples as in phase A).
    u_off = __builtin_amdgcn_readfirstlane(u_off);
    u_hs = __builtin_amdgcn_readfirstlane(u_hs);
    u_R = __builtin_amdgcn_readfirstlane(u_R);
    uint32_t u_stride[D];
    bool     u_dense;
    {
        uint32_t stride = 1;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {                 // the stride walk of grid_row, scalar
            u_stride[d] = stride <= u_hs ? stride : 0u;
            if (stride <= u_hs) stride *= u_R;
        }
        u_dense = !(stride > u_hs);
    }
    const bool u_pow2 = (u_hs & (u_hs - 1)) == 0;
    auto row_of = [&](uint64_t k) -> uint32_t {
        if (min_level_id) {                                // per-point level windows: level is in the key
            const uint32_t level = (uint32_t)(k >> 52);
            const uint32_t off = (uint32_t)offsets[level];
            const uint32_t hs = (uint32_t)offsets[level + 1] - off;
            const uint32_t R = (uint32_t)resolutions[level];
            uint32_t q[D];
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                const uint32_t gd = (uint32_t)(k >> (16 * d)) & 0xFFFFu;
                q[d] = ((c >> d) & 1u) ? min(gd + 1, R - 1) : gd;
            }
            return off + grid_row<D>(q, hs, R);
        }
        constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u,
                                        2097192037u, 1434869437u, 2165219737u};
        uint32_t index = 0;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            const uint32_t gd = (uint32_t)(k >> (16 * d)) & 0xFFFFu;
            const uint32_t qd = ((c >> d) & 1u) ? min(gd + 1, u_R - 1) : gd;
            if (u_dense) index += qd * u_stride[d];
            else index ^= qd * primes[d];
        }
        if (u_pow2) index &= u_hs - 1;
        else if (index >= u_hs) index %= u_hs;
        return u_off + index;
    };

    if (n_adjacent * 4 < total) {
        // few neighbouring runs (random points, or levels finer than the sample spacing): nothing
        // to combine, so every run goes straight out, runs interleaved over the groups
        for (uint32_t r = grp; r < total; r += GROUPS) {
            const uint32_t p0 = s_run_start[r], p1 = s_run_start[r + 1];
            if (!((s_valid[p0] >> c) & 1u)) continue;
            float acc = 0;
            for (uint32_t p = p0; p < p1; p++) acc += s_tw[p][c] * s_g[p][f];
            if constexpr (kRowsFromKey) flush(row_of(s_key[p0]), acc);
            else flush(s_row[p0][c], acc);
        }
        return;
    }
    const uint32_t gbase = lane - l;                       // first lane of my group inside the wave
    uint64_t fmask = 0;                                    // lanes (c', f) of my group, all c'
#pragma unroll
    for (uint32_t cc = 0; cc < C; cc++) fmask |= 1ull << (gbase + cc * F + f);
    const uint32_t rpg = (total + GROUPS - 1) / GROUPS;    // block-uniform trip count
    const uint32_t r_begin = grp * rpg;
    const uint32_t r_end = min(total, r_begin + rpg);

    if constexpr (SLOTS == 64) {
        // One run per wave at a time.  Which rows of the pending run reappear in the new one follows
        // from the two cells alone: corner b' of the new cell is corner b' + delta of the pending
        // cell whenever that is a corner at all (delta = cell' - cell, every axis within +-1).  The
        // keys are wave-uniform, so delta is scalar and every lane derives its partner from its own
        // corner bits — no row compares, cross-lane reads or ballots (PMC: the kernel was VALU-bound
        // on those, ~100 vector instructions per run at every level).  Equal rows of different
        // vertices (hash collisions) are simply not combined.
        uint32_t carry_row = NONE;
        float    carry_acc = 0;
        uint32_t carry_lo = ~0u, carry_hi = ~0u;           // key of the pending run (~0: none)
        // The run's head record (bounds, validity, key, first sample's weight and gradient) is read
        // one run ahead: a wave walks its runs serially, and the dependent LDS round trips
        // (bounds -> head -> samples) otherwise add up to most of a run's latency.
        auto head_of = [&](uint32_t r, uint32_t& p0, uint32_t& p1, uint64_t& k, bool& ok, float& tw0,
                           float& g0) {
            const bool     live = r < r_end;
            const uint32_t rr = live ? r : 0u;
            p0 = s_run_start[rr];
            p1 = live ? (uint32_t)s_run_start[rr + 1] : p0;
            const uint32_t q = p0 < 256u ? p0 : 255u;
            k = live ? s_key[q] : ~0ull;
            ok = live && ((s_valid[q] >> c) & 1u);
            tw0 = s_tw[q][c];
            g0 = s_g[q][f];
        };
        uint32_t n_p0, n_p1;
        uint64_t n_key;
        bool     n_ok;
        float    n_tw0, n_g0;
        head_of(r_begin, n_p0, n_p1, n_key, n_ok, n_tw0, n_g0);
        for (uint32_t i = 0; i <= rpg; i++) {              // one extra round drains the pending run
            const uint32_t p0 = n_p0, p1 = n_p1;
            const uint32_t k_lo = __builtin_amdgcn_readfirstlane((uint32_t)n_key);
            const uint32_t k_hi = __builtin_amdgcn_readfirstlane((uint32_t)(n_key >> 32));
            const bool     ok = n_ok;
            const float    tw0 = n_tw0, g0 = n_g0;
            head_of(i < rpg ? r_begin + i + 1 : r_end, n_p0, n_p1, n_key, n_ok, n_tw0, n_g0);
            uint32_t my_row = NONE;
            float    acc = 0;
            if (ok) {
                my_row = row_of((uint64_t)k_hi << 32 | k_lo);
                acc = tw0 * g0;
                for (uint32_t p = p0 + 1; p < p1; p++) acc += s_tw[p][c] * s_g[p][f];
            }
            // scalar: is the new cell a neighbour of the pending one (same level)?
            bool adj = (k_lo & k_hi) != ~0u && (carry_lo & carry_hi) != ~0u && (k_hi >> 20) == (carry_hi >> 20);
            int  delta[D];
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                const uint32_t a_new = d == 0 ? (k_lo & 0xFFFFu) : d == 1 ? (k_lo >> 16) : (k_hi & 0xFFFFu);
                const uint32_t a_old = d == 0 ? (carry_lo & 0xFFFFu) : d == 1 ? (carry_lo >> 16) : (carry_hi & 0xFFFFu);
                delta[d] = (int)a_new - (int)a_old;
                adj = adj && delta[d] >= -1 && delta[d] <= 1;
            }
            bool     shared = adj, claimed = adj;          // my new corner is a pending one / my pending
            uint32_t jm = 0;                               // corner is a new one
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                const int bit = (int)((c >> d) & 1u);
                const int as_old = bit + delta[d], as_new = bit - delta[d];
                shared = shared && (uint32_t)as_old <= 1u;
                claimed = claimed && (uint32_t)as_new <= 1u;
                jm |= ((uint32_t)as_old & 1u) << d;
            }
            const float ca = __shfl(carry_acc, (int)(jm * F + f));
            if (shared && ok) acc += ca;
            // a claimed pending lane is absorbed by the new run only if that corner is live there
            // (validity is a property of the vertex, so it is the pending lane's own validity)
            if (carry_row != NONE && !claimed) flush(carry_row, carry_acc);
            carry_row = my_row;
            carry_acc = acc;
            carry_lo = k_lo;
            carry_hi = k_hi;
        }
        return;
    }

    uint32_t carry_row = NONE;
    float    carry_acc = 0;
    for (uint32_t i = 0; i <= rpg; i++) {                  // one extra round drains the pending run
        const uint32_t r = r_begin + i;
        uint32_t my_row = NONE;
        float    acc = 0;
        if (i < rpg && r < r_end) {
            const uint32_t p0 = s_run_start[r], p1 = s_run_start[r + 1];
            if ((s_valid[p0] >> c) & 1u) {
                my_row = s_row[p0][c];
                for (uint32_t p = p0; p < p1; p++) acc += s_tw[p][c] * s_g[p][f];
            }
        }
        bool claimed = false;
#pragma unroll
        for (uint32_t j = 0; j < C; j++) {
            const uint32_t src = gbase + j * F + f;
            const uint32_t cr = __shfl(carry_row, src);
            const float    ca = __shfl(carry_acc, src);
            const bool     hit = (my_row != NONE) && (cr == my_row);
            const uint64_t cand = __ballot(hit) & fmask;
            if (cand != 0) {
                // pending lane (j, f) is taken over by exactly one new lane: the lowest one
                if (c == j) claimed = true;
                if ((uint32_t)__builtin_ctzll(cand) == lane) acc += ca;
            }
        }
        if (carry_row != NONE && !claimed) flush(carry_row, carry_acc);
        carry_row = my_row;
        carry_acc = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// cnt_np_embed: per fine-level vertex, vote +1 / -1 of each feature into the 2-D projection
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool cnt_loc(const int16_t* __restrict__ p, uint32_t R, uint32_t F,
                                        uint32_t axis, uint32_t (&q)[3], uint32_t& loc)
{
    q[0] = (uint32_t)(int32_t)p[0];
    q[1] = (uint32_t)(int32_t)p[1];
    q[2] = (uint32_t)(int32_t)p[2];
    const uint32_t scale = R - 2;
    bool inside = true;
#pragma unroll
    for (int d = 0; d < 3; d++) inside &= !(q[d] <= 0 || q[d] >= R - 1);
    const uint32_t u = axis == 2 ? q[1] : q[0];
    const uint32_t w = axis == 0 ? q[1] : q[2];
    loc = (u - 1) * scale * F * 2 + (w - 1) * F * 2;
    return inside;
}

// One request on the L2 atomic path is one (instruction, 64-byte segment) pair (see the backward
// kernel).  A pixel's counters are F x {pos, neg} floats = 64 B at F=8, so a vertex is handled by
// 2F adjacent lanes (one per counter) and costs ONE request instead of F; each lane group walks
// KV consecutive vertices of the (sorted) list and keeps the running counters in registers while
// the pixel does not change (vertices differing only along the projection axis are consecutive
// for the xy plane), so a column of up to R-2 vertices collapses to R/KV requests.
template <uint32_t F>
__global__ __launch_bounds__(256) void k_cnt_np_embed(const int16_t* __restrict__ inputs,
                                                      const float* __restrict__ emb,
                                                      float* __restrict__ out, uint32_t N,
                                                      uint32_t R, uint32_t hs, uint32_t axis)
{
    constexpr uint32_t LPV = 2 * F;                  // lanes per vertex: (channel, pos|neg)
    constexpr uint32_t KV = 8;                       // consecutive vertices per lane group
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = t / LPV, sub = t % LPV;
    const uint32_t ch = sub >> 1, neg = sub & 1u;
    const uint32_t v0 = grp * KV;
    if (v0 >= N) return;
    uint32_t carry_loc = NONE;
    float    carry = 0;
    for (uint32_t i = 0; i < KV; i++) {
        const uint32_t b = v0 + i;
        if (b >= N) break;
        uint32_t q[3], loc;
        if (!cnt_loc(inputs + (size_t)b * 3, R, F, axis, q, loc)) continue;
        const float e = emb[(size_t)grid_row<3>(q, hs, R) * F + ch];
        const bool  pos = (double)e > 0.9;           // float vs double literal, gridencoder.cu:909
        const float one = (pos != (neg != 0)) ? 1.0f : 0.0f;
        if (loc == carry_loc) {
            carry += one;
        } else {
            if (carry_loc != NONE) unsafeAtomicAdd(out + carry_loc + sub, carry);
            carry_loc = loc;
            carry = one;
        }
    }
    if (carry_loc != NONE) unsafeAtomicAdd(out + carry_loc + sub, carry);
}

// backward: one lane per (vertex, channel): the F lanes of a vertex update one 32-byte table row
// with a single request (the reference: one thread per vertex, F separate atomics).
template <uint32_t F>
__global__ __launch_bounds__(256) void k_cnt_np_embed_bwd(
    const int16_t* __restrict__ inputs, const float* __restrict__ emb,
    const float* __restrict__ out_sum, const float* __restrict__ grad,
    float* __restrict__ grad_emb, uint32_t N, uint32_t R, uint32_t hs, uint32_t axis)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = t / F, ch = t % F;
    if (b >= N) return;
    uint32_t q[3], loc;
    if (!cnt_loc(inputs + (size_t)b * 3, R, F, axis, q, loc)) return;
    const size_t   at = (size_t)grid_row<3>(q, hs, R) * F;
    const uint32_t half = loc / 2;
    const float gv = 1 / out_sum[half + ch];
    const bool  pos = (double)emb[at + ch] > 0.9;
    const float contrib = pos ? gv * grad[loc + ch * 2 + 0] : -gv * grad[loc + ch * 2 + 1];
    unsafeAtomicAdd(grad_emb + at + ch, contrib);
}

// ---------------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------------
struct EncArgs {
    const float*   inputs;
    const float*   emb;
    const int32_t* offsets;
    const int32_t* resolutions;
    float*         out;        // forward: outputs; backward: grad_embeddings
    const float*   grad;       // backward only
    uint32_t       N, L, Rb;
    const uint8_t* vxl;
    const int32_t* mli;
    hipStream_t    stream;
    const uint32_t* clip_count = nullptr;   // backward + STE only
    const int32_t*  sat = nullptr;          // optional summed-volume table of the occupancy grid
    FeatLayout      lay{0, 0};              // where outputs (forward) / gradients (backward) live
};

template <uint32_t D, uint32_t F, bool VXL, bool STE>
static void launch_fwd(const EncArgs& a)
{
    constexpr uint32_t V = F < 4 ? F : 4, G = F / V;
    const dim3 grid(div_up(a.N * G, 256), a.L, 1);
    hipLaunchKernelGGL((k_grid_encode_fwd<D, F, VXL, STE>), grid, dim3(256), 0, a.stream,
                       a.inputs, a.emb, a.offsets, a.resolutions, a.out, a.N, a.Rb, a.vxl, a.mli, a.sat, a.lay);
}

// grid_encode_cells.hip (CellsArgs: encoder_common.hpp)
bool launch_bwd_cells(const CellsArgs& a, uint32_t D, uint32_t F, bool ste, hipStream_t s);

// grid_encode_merge.hip
void launch_bwd_merge(const float* grad, const float* inputs, const float* emb, const int32_t* offsets,
                      const int32_t* resolutions, float* grad_emb, uint32_t N, uint32_t L,
                      const uint32_t* clip_count, FeatLayout lay, bool ste, hipStream_t s);

template <uint32_t D, uint32_t F, bool VXL, bool STE>
static void launch_bwd(const EncArgs& a)
{
    if constexpr (D == 3 && F == 8 && !VXL) {
        // coarse half of a binned call: runs merged across the rays of a 1024-sample block
        if (a.lay.finest_first && !a.mli && (uint64_t)div_up(a.N, 256) * a.L < (1ull << 31)) {
            launch_bwd_merge(a.grad, a.inputs, a.emb, a.offsets, a.resolutions, a.out, a.N, a.L, a.clip_count,
                             a.lay, STE, a.stream);
            return;
        }
    }
    if constexpr ((1u << D) * F <= 64) {
        const uint64_t blocks = (uint64_t)div_up(a.N, 256) * a.L;
        FeatLayout     lay = a.lay;
        dim3           grid(div_up(a.N, 256), a.L, 1);
        // level slot as the fast block index (see the kernel) for the coarse part of a binned call;
        // with all 16 levels on this kernel the interleaving costs L2/MALL locality on the big tables
        // (2.30 -> 2.83 ms), so the plain entry keeps the level-major order
        if (a.lay.finest_first && blocks < (1ull << 31)) {
            lay.n_slots = a.L;
            grid = dim3((uint32_t)blocks, 1, 1);
        }
        hipLaunchKernelGGL((k_grid_encode_bwd<D, F, VXL, STE>), grid, dim3(256), 0, a.stream,
                           a.grad, a.inputs, a.emb, a.offsets, a.resolutions, a.out, a.N, a.Rb,
                           a.vxl, a.mli, a.clip_count, a.sat, lay);
    } else {
        constexpr uint32_t V = F < 4 ? F : 4, G = F / V;
        const dim3 grid(div_up(a.N * G, 256), a.L, 1);
        hipLaunchKernelGGL((k_grid_encode_bwd_simple<D, F, VXL, STE>), grid, dim3(256), 0,
                           a.stream, a.grad, a.inputs, a.emb, a.offsets, a.resolutions, a.out, a.N,
                           a.Rb, a.vxl, a.mli, a.clip_count, a.sat, a.lay);
    }
}


template <bool BWD, uint32_t D, uint32_t F>
static void dispatch_flags(const EncArgs& a, bool ste)
{
    const bool vxl = a.vxl != nullptr;
#define CNC_GO(VX, ST)                                   \
    do {                                                 \
        if constexpr (BWD) launch_bwd<D, F, VX, ST>(a);  \
        else launch_fwd<D, F, VX, ST>(a);                \
    } while (0)
    if (vxl && ste) CNC_GO(true, true);
    else if (vxl) CNC_GO(true, false);
    else if (ste) CNC_GO(false, true);
    else CNC_GO(false, false);
#undef CNC_GO
}

template <bool BWD, uint32_t D>
static int dispatch_F(const EncArgs& a, uint32_t F, bool ste)
{
    switch (F) {
    case 1: dispatch_flags<BWD, D, 1>(a, ste); break;
    case 2: dispatch_flags<BWD, D, 2>(a, ste); break;
    case 4: dispatch_flags<BWD, D, 4>(a, ste); break;
    case 8: dispatch_flags<BWD, D, 8>(a, ste); break;
    case 16: dispatch_flags<BWD, D, 16>(a, ste); break;
    case 32: dispatch_flags<BWD, D, 32>(a, ste); break;
    default: return CNC_ERR_INVALID_VALUE;   // "n_fearures must be 1, 2, 4, 8, 16 or 32"
    }
    return CNC_OK;
}

template <bool BWD>
static int dispatch_D(const EncArgs& a, uint32_t D, uint32_t F, bool ste)
{
    switch (D) {
    case 1: return dispatch_F<BWD, 1>(a, F, ste);
    case 2: return dispatch_F<BWD, 2>(a, F, ste);
    case 3: return dispatch_F<BWD, 3>(a, F, ste);
    default: return CNC_ERR_INVALID_VALUE;   // "num_dim must be 1, 2, 3"
    }
}

// point-major layout: rows must hold the encoder's block and keep the vector accesses aligned
static bool layout_ok(FeatLayout lay, uint32_t F, uint32_t L)
{
    if (lay.ld == 0) return lay.col == 0;
    const uint32_t V = F < 4 ? F : 4;
    return lay.col + L * F <= lay.ld && lay.ld % V == 0 && lay.col % V == 0;
}

// grid_input_grad.hip
int launch_dy_dx(const float* inputs, const float* emb, const int32_t* offsets, const int32_t* resolutions,
                 float* dy_dx, uint32_t N, uint32_t D, uint32_t F, uint32_t L, const int32_t* mli, bool ste,
                 hipStream_t s);
int launch_input_backward(const float* grad, const float* dy_dx, float* grad_inputs, uint32_t N, uint32_t D,
                          uint32_t F, uint32_t L, FeatLayout lay, hipStream_t s);

}  // namespace cnc

using namespace cnc;

extern "C" int cnc_grid_encode_forward(const float* inputs, const float* embeddings,
                                       const int32_t* offsets, const int32_t* resolutions,
                                       float* outputs, uint32_t N, uint32_t D, uint32_t F,
                                       uint32_t L, uint32_t Rb, float PV, float* dy_dx,
                                       const uint8_t* binary_vxl, const int32_t* min_level_id,
                                       uint32_t flags, const int32_t* occ_sat,
                                       const uint32_t* vertex_bits, const int32_t* vertex_bit_offsets,
                                       uint32_t out_ld, uint32_t out_col, void* stream)
{
    (void)PV;
    if (N == 0 || L == 0) return CNC_OK;
    if (!inputs || !embeddings || !offsets || !resolutions || !outputs) return CNC_ERR_INVALID_VALUE;
    EncArgs a{inputs, embeddings, offsets, resolutions, outputs, nullptr, N, L, Rb,
              binary_vxl, min_level_id, (hipStream_t)stream, nullptr, binary_vxl ? occ_sat : nullptr,
              FeatLayout{out_ld, out_col}};
    if (!layout_ok(a.lay, F, L)) return CNC_ERR_INVALID_VALUE;
    if (binary_vxl && vertex_bits && vertex_bit_offsets) { a.lay.vbits = vertex_bits; a.lay.vboff = vertex_bit_offsets; }
    int rc = dispatch_D<false>(a, D, F, (flags & CNC_FLAG_STE_BINARY) != 0);
    if (rc == CNC_OK && dy_dx)   // the dy_dx branch of kernel_grid, as its own launch (not a hot path)
        rc = launch_dy_dx(inputs, embeddings, offsets, resolutions, dy_dx, N, D, F, L, min_level_id,
                          (flags & CNC_FLAG_STE_BINARY) != 0, (hipStream_t)stream);
    return rc != CNC_OK ? rc : launch_status();
}

extern "C" int cnc_grid_encode_backward(const float* grad, const float* inputs,
                                        const float* embeddings, const int32_t* offsets,
                                        const int32_t* resolutions, float* grad_embeddings,
                                        uint32_t N, uint32_t D, uint32_t F, uint32_t L,
                                        uint32_t Rb, const float* dy_dx, float* grad_inputs,
                                        const uint8_t* binary_vxl, const int32_t* min_level_id,
                                        uint32_t flags, const uint32_t* ste_clip_count,
                                        const int32_t* occ_sat, const uint32_t* vertex_bits,
                                        const int32_t* vertex_bit_offsets, uint32_t grad_ld,
                                        uint32_t grad_col, void* stream)
{
    if ((dy_dx == nullptr) != (grad_inputs == nullptr)) return CNC_ERR_INVALID_VALUE;   // both or neither
    if (N == 0 || L == 0) return CNC_OK;
    if (!grad || !inputs || !embeddings || !offsets || !resolutions || !grad_embeddings)
        return CNC_ERR_INVALID_VALUE;
    EncArgs a{inputs, embeddings, offsets, resolutions, grad_embeddings, grad, N, L, Rb,
              binary_vxl, min_level_id, (hipStream_t)stream, ste_clip_count,
              binary_vxl ? occ_sat : nullptr,
              FeatLayout{grad_ld, grad_col, (flags & CNC_FLAG_LEVELS_FINEST_FIRST) ? 1u : 0u}};
    if (!layout_ok(a.lay, F, L)) return CNC_ERR_INVALID_VALUE;
    if (binary_vxl && vertex_bits && vertex_bit_offsets) { a.lay.vbits = vertex_bits; a.lay.vboff = vertex_bit_offsets; }
    int rc = CNC_OK;
    bool done = false;
    if ((flags & CNC_FLAG_CELL_MERGE) && !dy_dx) {
        const CellsArgs ca{grad, inputs, embeddings, offsets, resolutions, grad_embeddings, binary_vxl, min_level_id,
                           ste_clip_count, a.sat, a.lay, N, L, Rb, (flags & CNC_FLAG_CELL_CARRY) ? 1u : 0u};
        done = launch_bwd_cells(ca, D, F, (flags & CNC_FLAG_STE_BINARY) != 0, (hipStream_t)stream);
    }
    if (!done) rc = dispatch_D<true>(a, D, F, (flags & CNC_FLAG_STE_BINARY) != 0);
    if (rc == CNC_OK && dy_dx)   // kernel_input_backward (gridencoder.cu:588-614)
        rc = launch_input_backward(grad, dy_dx, grad_inputs, N, D, F, L, FeatLayout{grad_ld, grad_col},
                                   (hipStream_t)stream);
    return rc != CNC_OK ? rc : launch_status();
}

#define CNC_F_SWITCH(F, CALL)                         \
    switch (F) {                                      \
    case 1: { constexpr uint32_t FF = 1; CALL; } break;   \
    case 2: { constexpr uint32_t FF = 2; CALL; } break;   \
    case 4: { constexpr uint32_t FF = 4; CALL; } break;   \
    case 8: { constexpr uint32_t FF = 8; CALL; } break;   \
    case 16: { constexpr uint32_t FF = 16; CALL; } break; \
    case 32: { constexpr uint32_t FF = 32; CALL; } break; \
    default: return CNC_ERR_INVALID_VALUE;            \
    }

extern "C" int cnc_cnt_np_embed(const int16_t* inputs, const float* embeddings_clip,
                                float* outputs, uint32_t N, uint32_t resolution, uint32_t F,
                                uint32_t hashmap_size, uint32_t axis, void* stream)
{
    if (N == 0) return CNC_OK;
    if (!inputs || !embeddings_clip || !outputs || axis > 2) return CNC_ERR_INVALID_VALUE;
    CNC_F_SWITCH(F, hipLaunchKernelGGL((k_cnt_np_embed<FF>), dim3(div_up(div_up(N, 8) * 2 * FF, 256)),
                                       dim3(256), 0, (hipStream_t)stream, inputs, embeddings_clip,
                                       outputs, N, resolution, hashmap_size, axis));
    return launch_status();
}

extern "C" int cnc_cnt_np_embed_backward(const int16_t* inputs, const float* embeddings_clip,
                                         const float* outputs_sum, const float* grad,
                                         float* grad_embeddings, uint32_t N, uint32_t resolution,
                                         uint32_t F, uint32_t hashmap_size, uint32_t axis,
                                         void* stream)
{
    if (N == 0) return CNC_OK;
    if (!inputs || !embeddings_clip || !outputs_sum || !grad || !grad_embeddings || axis > 2)
        return CNC_ERR_INVALID_VALUE;
    CNC_F_SWITCH(F, hipLaunchKernelGGL((k_cnt_np_embed_bwd<FF>), dim3(div_up(N * FF, 256)), dim3(256),
                                       0, (hipStream_t)stream, inputs, embeddings_clip, outputs_sum,
                                       grad, grad_embeddings, N, resolution, hashmap_size, axis));
    return launch_status();
}


extern "C" int cnc_pack_sign_bits(const float* embeddings, uint8_t* bits, uint64_t rows, uint32_t F,
                                  uint32_t* clip_count, void* stream)
{
    if (clip_count && hipMemsetAsync(clip_count, 0, sizeof(uint32_t), (hipStream_t)stream) != hipSuccess)
        return CNC_ERR_LAUNCH;
    if (rows == 0) return CNC_OK;
    if (!embeddings || !bits) return CNC_ERR_INVALID_VALUE;
    if (!(F == 1 || F == 2 || F == 4 || F == 8 || F == 16 || F == 32)) return CNC_ERR_INVALID_VALUE;
    const uint64_t n_vals = rows * F, n_bytes = (n_vals + 7) / 8;
    const uint64_t want = (n_bytes + 255) / 256;
    const uint32_t grid = (uint32_t)(want < 256ull * 32 ? want : 256ull * 32);
    hipLaunchKernelGGL(k_pack_sign_bits, dim3(grid), dim3(256), 0, (hipStream_t)stream, embeddings,
                       bits, n_bytes, n_vals, clip_count);
    return launch_status();
}

template <uint32_t D, uint32_t F>
static void launch_fwd_bits(const float* inputs, const uint8_t* bits, const int32_t* offsets,
                            const int32_t* resolutions, float* outputs, uint32_t N, uint32_t L,
                            uint32_t Rb, const uint8_t* vxl, const int32_t* mli, const int32_t* sat,
                            FeatLayout lay, hipStream_t s)
{
    // point-major rows: several level slots per lane (see the kernel)
    uint32_t P = 1;
    if (lay.ld != 0 && L > 1) {
        P = F >= 16 ? 1u : 16u / F;      // 16 floats = one 64-byte line per point and pass
        if (P > L) P = L;
    }
    // wave-transposed stores (see the kernel): whenever a point's piece is 2, 4 or 8 whole 16-byte chunks in every
    // pass and the rows keep them aligned (CNC_FWD_TR=0: measurement switch, the round-3 per-lane stores)
    const int tr_mode = getenv("CNC_FWD_TR") ? atoi(getenv("CNC_FWD_TR")) : 1;
    const uint32_t W = P * F, tail = (L % P) * F;
    const bool tr = tr_mode && (W == 8 || W == 16 || W == 32) && tail % 4 == 0 && lay.ld % 4 == 0 && lay.col % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(outputs) & 15u) == 0;
    const uint32_t cp_log2 = W == 8 ? 1u : (W == 16 ? 2u : 3u);
    // streaming stores of the level-major outputs (a wave writes whole lines): the bit plane keeps the L2
    // (CNC_FWD_NT=0: measurement switch).  Point-major rows: only with the transposed stores — a lane's own 16-byte
    // pieces need the L2 to merge into lines (streamed they cost 3.7x).
    const int nt_mode = getenv("CNC_FWD_NT") ? atoi(getenv("CNC_FWD_NT")) : 1;
    lay.nt = (nt_mode == 1 && lay.ld == 0) || (nt_mode == 2 && tr) ? 1u : 0u;
    const dim3 grid(div_up(N, 256), div_up(L, P), 1);
    const size_t lds = tr ? (size_t)256 * W * sizeof(float) : 0;
#define CNC_FWD_BITS(VX, TRV)                                                                                        \
    hipLaunchKernelGGL((k_grid_encode_fwd_bits<D, F, VX, TRV>), grid, dim3(256), TRV ? lds : 0, s, inputs, bits, offsets, \
                       resolutions, outputs, N, L, P, cp_log2, Rb, vxl, mli, VX ? sat : nullptr, lay)
    if (vxl) { if (tr) CNC_FWD_BITS(true, true); else CNC_FWD_BITS(true, false); }
    else     { if (tr) CNC_FWD_BITS(false, true); else CNC_FWD_BITS(false, false); }
#undef CNC_FWD_BITS
}

extern "C" int cnc_grid_encode_forward_bits(const float* inputs, const uint8_t* bits,
                                            const int32_t* offsets, const int32_t* resolutions,
                                            float* outputs, uint32_t N, uint32_t D, uint32_t F,
                                            uint32_t L, uint32_t Rb, const uint8_t* binary_vxl,
                                            const int32_t* min_level_id, const int32_t* occ_sat,
                                            const uint32_t* vertex_bits, const int32_t* vertex_bit_offsets,
                                            uint32_t out_ld, uint32_t out_col, void* stream)
{
    if (N == 0 || L == 0) return CNC_OK;
    if (!inputs || !bits || !offsets || !resolutions || !outputs) return CNC_ERR_INVALID_VALUE;
    FeatLayout lay{out_ld, out_col};
    if (!layout_ok(lay, F, L)) return CNC_ERR_INVALID_VALUE;
    if (binary_vxl && vertex_bits && vertex_bit_offsets) { lay.vbits = vertex_bits; lay.vboff = vertex_bit_offsets; }
    hipStream_t s = (hipStream_t)stream;
#define CNC_BITS_D(DD)                                                                              \
    CNC_F_SWITCH(F, (launch_fwd_bits<DD, FF>(inputs, bits, offsets, resolutions, outputs, N, L, Rb, \
                                             binary_vxl, min_level_id, occ_sat, lay, s)))
    switch (D) {
    case 1: CNC_BITS_D(1); break;
    case 2: CNC_BITS_D(2); break;
    case 3: CNC_BITS_D(3); break;
    default: return CNC_ERR_INVALID_VALUE;
    }
#undef CNC_BITS_D
    return launch_status();
}

// ---------------------------------------------------------------------------------------------
// Vertex bit plane of one level: bit q0 + R (q1 + R q2) = box_any(q) (gridencoder.cu:221-276) for EVERY vertex of
// the level, from the summed-volume table.  The masked kernels then read one bit per corner instead of 2^D table
// entries (192 loads per vertex of a 3-level context window: 0.69 ms forward / 1.15 ms backward per training step).
// Depends on (occupancy, R, Rb) only: rebuilt when the occupancy grid changes.
// ---------------------------------------------------------------------------------------------
namespace cnc {
template <uint32_t D>
__global__ __launch_bounds__(256) void k_vertex_bits(const int32_t* __restrict__ sat, uint32_t R, uint32_t Rb,
                                                     uint64_t n_vertices, uint32_t* __restrict__ words)
{
    const uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    bool bit = false;
    if (v < n_vertices) {
        uint32_t q[D];
        uint64_t r = v;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) { q[d] = (uint32_t)(r % R); r /= R; }
        bit = box_any_sat<D>(q, R, Rb, sat);
    }
    const uint64_t m = __ballot(bit);
    const uint32_t lane = threadIdx.x & 63u;
    // the plane holds ceil(n / 64) * 2 words (cnc_grid_vertex_bits_words): the waves of the last block that lie
    // wholly past the last vertex store nothing; lane 0 / 32 hold v = first vertex of their half-wave + 0 / 32 and
    // the half-waves of the last PARTIAL wave both have a word (the plane is rounded up to whole 64-vertex pairs)
    if (v - lane < n_vertices) {
        if (lane == 0) words[v >> 5] = (uint32_t)m;                   // v is a multiple of 64 here
        else if (lane == 32) words[v >> 5] = (uint32_t)(m >> 32);
    }
}
}  // namespace cnc

extern "C" uint64_t cnc_grid_vertex_bits_words(uint32_t D, uint32_t R)
{
    uint64_t n = 1;
    for (uint32_t d = 0; d < D; d++) n *= R;
    return (n + 63) / 64 * 2;
}

extern "C" int cnc_grid_vertex_bits(const int32_t* occ_sat, uint32_t D, uint32_t Rb, uint32_t R, uint32_t* words,
                                    void* stream)
{
    if (!occ_sat || !words || R < 3 || D < 1 || D > 3) return CNC_ERR_INVALID_VALUE;
    uint64_t n = 1;
    for (uint32_t d = 0; d < D; d++) n *= R;
    const uint64_t blocks = (n + 255) / 256;
    if (blocks >= (1ull << 31)) return CNC_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    switch (D) {
    case 1: hipLaunchKernelGGL((k_vertex_bits<1>), dim3((uint32_t)blocks), dim3(256), 0, s, occ_sat, R, Rb, n, words); break;
    case 2: hipLaunchKernelGGL((k_vertex_bits<2>), dim3((uint32_t)blocks), dim3(256), 0, s, occ_sat, R, Rb, n, words); break;
    default: hipLaunchKernelGGL((k_vertex_bits<3>), dim3((uint32_t)blocks), dim3(256), 0, s, occ_sat, R, Rb, n, words); break;
    }
    return launch_status();
}

extern "C" const char* cnc_error_string(int code)
{
    switch (code) {
    case CNC_OK: return "ok";
    case CNC_ERR_INVALID_VALUE:
        return "invalid argument (null pointer, bad size, n_features not in {1,2,4,8,16,32} or "
               "num_dim not in {1,2,3})";
    case CNC_ERR_UNSUPPORTED: return "argument combination not supported by libcnc_hip";
    case CNC_ERR_LAUNCH: return "HIP kernel launch failed";
    default: return "unknown error";
    }
}

extern "C" int cnc_abi_version(void) { return 31; }
