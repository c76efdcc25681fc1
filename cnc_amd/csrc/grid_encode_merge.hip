// grid_encode_merge.hip — backward scatter of the coarse levels with run merging across rays.
//
// k_grid_encode_bwd (grid_encode.hip) works on 256 consecutive samples per block: less than one ray of
// a marched frame (200-400 samples per ray), so the runs it merges are runs along ONE ray.  The rays of
// neighbouring pixels cross the same cells: of the runs in 1024 consecutive samples (3-4 rays) only 0.29
// (level 0) to 0.46 (resolution 214) open a cell no earlier run of the block visited; in 256 samples it is
// 0.93-0.96.  This kernel takes MB = 1024 (or 512) samples per block, chains the runs of equal cells through a
// small LDS hash table and sends ONE set of atomics per distinct cell — the coarse levels are bound by the
// memory-side atomic units (docs/engineering_log.md §4.2b), so the number of atomic instructions is what their time was
// made of.
//
// How it got here (9 coarse levels of the bench grid, ms per 2^20 marched samples; k_grid_encode_bwd: 0.523):
//   * first version, 8 stored weights per sample, block size 256 / 384 / 512 / 640 / 768 / 1024:
//     0.485 / 0.548 / 0.432 / 0.629 / 0.444 / 0.573 — more samples merge more, but the LDS they need
//     leaves fewer waves per CU;
//   * weights rebuilt from 4 floats per sample (half the LDS): 512 -> 0.426, 1024 -> 0.429; by then the
//     atomics were a sixth of the time (0.368 without them) and the per-sample loop was what was left
//     (a sample-minor LDS layout with 16-byte reads of 4 samples per lane was slower: 0.461);
//   * per-cell accumulation as an 8 x n by n x 8 product on v_mfma_f32_16x16x4_f32: 0.403;
//   * one packed LDS word per run and one 16-byte record per cell: 0.384;
//   * the cost now being per distinct cell, 1024 samples (75 KB of LDS, 2 blocks per CU): 0.365.
//
// D = 3, F = 8 (one cell per wave at a time: 64 lanes = 8 corners x 8 features), no occupancy mask, no
// per-point level window: the coarse half of a binned backward call.  Everything else stays on
// k_grid_encode_bwd.
#include <cstdlib>

#include "common.hpp"
#include "encoder_common.hpp"

namespace cnc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MB = samples (= threads) per block, a multiple of 64: 1024 for the frames of the bench (more samples per block merge
// more), 512 when the whole launch is only a few rounds of 1024-sample blocks — a training batch of 2^18 samples x 11
// levels is 2.8 k such blocks on 512 block slots, and ran 4x less efficiently than the 2^20-sample chunks.
template <bool STE, uint32_t MB>
__global__ __launch_bounds__(MB) void k_grid_encode_bwd_merge(
    const float* __restrict__ grad, const float* __restrict__ inputs, const float* __restrict__ emb,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ resolutions,
    float* __restrict__ grad_emb, uint32_t N, const uint32_t* __restrict__ clip_count, FeatLayout lay)
{
    constexpr uint32_t kMB = MB;
    constexpr uint32_t kMW = kMB / 64;        // waves per block
    constexpr uint32_t kMSlots = kMB <= 512 ? 1024 : 2048;   // hash slots, power of two, >= 2 x the most runs a block can have
    constexpr uint32_t D = 3, F = 8, C = 8, END = 0x7FFu;
    static_assert(kMB <= 1024, "run records pack start (10 bits) / end (11) / next (11)");
    // the three fractional positions and 1 / (sum of valid weights): the lane rebuilds its corner's
    // weight from them (same products, same order as Corners::setup) — half the LDS of 8 stored weights
    __shared__ __attribute__((aligned(16))) float s_w4[kMB][4];
    __shared__ float    s_g[kMB][F];
    // 16 bytes per thread, used twice: the sample keys (first half) and the hash table (second half)
    // until the runs are chained, then one record per distinct cell {first run record, key, validity}
    __shared__ __attribute__((aligned(16))) uint4 s_u[kMB];
    uint64_t* const s_key = reinterpret_cast<uint64_t*>(s_u);
    uint32_t* const h_slot = reinterpret_cast<uint32_t*>(s_u) + 2 * kMB;
    static_assert(kMSlots * 4 <= kMB * 8, "hash table fits the second half of s_u");
    __shared__ uint16_t s_run_start[kMB + 1];
    __shared__ uint32_t l_head[kMB];            // per representative run: last run chained to its cell
    __shared__ uint32_t s_run_rec[kMB];         // start | end << 10 | next run of the cell << 21: one read per run
    __shared__ uint32_t s_wave_heads[kMW], s_wave_claims[kMW];

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool     mask_on = STE && (clip_count == nullptr || *clip_count != 0);
    // 1-D grid, level slot as the fast index (see k_grid_encode_bwd), slots walked last to first
    const uint32_t n_slots = lay.n_slots;
    const uint32_t chunk = blockIdx.x / n_slots;
    const uint32_t slot = n_slots - 1 - blockIdx.x % n_slots;
    const uint32_t b = chunk * kMB + tid;
    const uint32_t off = (uint32_t)offsets[slot];
    const uint32_t hs = (uint32_t)offsets[slot + 1] - off;
    const uint32_t R = (uint32_t)resolutions[slot];

    for (uint32_t i = tid; i < kMSlots; i += kMB) h_slot[i] = 0;
    l_head[tid] = END;

    // ---- phase A: lane = sample ----
    uint64_t key = ~0ull;
    uint32_t validmask = 0;
    {
        float    x[D];
        if (b < N && load_point<D>(inputs, b, x)) {
            Corners<D, false> c;
            c.setup(x, R, hs, 0, nullptr);
            key = (uint64_t)c.cell[0] | (uint64_t)c.cell[1] << 16 | (uint64_t)c.cell[2] << 32;
#pragma unroll
            for (uint32_t i = 0; i < C; i++) validmask |= (c.valid[i] ? 1u : 0u) << i;
            *reinterpret_cast<float4*>(s_w4[tid]) = make_float4(c.frac[0], c.frac[1], c.frac[2], c.wn_re);
            const float* gp = grad + feat_index(lay, slot, N, b, F);
            float        g0[4], g1[4];
            load_vec<4>(gp, g0);
            load_vec<4>(gp + 4, g1);
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                s_g[tid][j] = g0[j];
                s_g[tid][4 + j] = g1[j];
            }
        }
        s_key[tid] = key;
    }
    __syncthreads();

    // ---- runs: consecutive samples with the same cell ----
    const bool     head = tid == 0 || s_key[tid - 1] != key;
    const uint64_t hb = __ballot(head);
    if (lane == 0) s_wave_heads[wave] = (uint32_t)__popcll(hb);
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kMW; w++) {
        const uint32_t h = s_wave_heads[w];
        before += w < wave ? h : 0u;
        total += h;
    }
    const uint32_t my_run = before + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull));
    if (head) s_run_start[my_run] = (uint16_t)tid;
    if (tid == 0) s_run_start[total] = (uint16_t)kMB;
    __syncthreads();

    // ---- cells: chain the runs of equal cells ----
    bool     claimer = false;
    if (head && key != ~0ull) {
        uint32_t sl = (((uint32_t)key ^ (uint32_t)(key >> 16) ^ (uint32_t)(key >> 32)) * 2654435761u) >> (32 - __builtin_ctz(kMSlots));
        uint32_t rep;
        for (;;) {
            const uint32_t seen = atomicCAS(&h_slot[sl], 0u, my_run + 1);
            if (seen == 0) { claimer = true; rep = my_run; break; }
            rep = seen - 1;
            if (s_key[s_run_start[rep]] == key) break;
            sl = (sl + 1) & (kMSlots - 1);
        }
        const uint32_t next = atomicExch(&l_head[rep], my_run);
        s_run_rec[my_run] = tid | (uint32_t)s_run_start[my_run + 1] << 10 | next << 21;
    }
    const uint64_t cb = __ballot(claimer);
    if (lane == 0) s_wave_claims[wave] = (uint32_t)__popcll(cb);
    __syncthreads();
    uint32_t g_before = 0, n_cells = 0;
#pragma unroll
    for (uint32_t w = 0; w < kMW; w++) {
        const uint32_t h = s_wave_claims[w];
        g_before += w < wave ? h : 0u;
        n_cells += h;
    }
    // (the sync above also ends the life of the keys and the hash table: s_u is rewritten here)
    uint4 my_cell = make_uint4(0, 0, 0, 0);
    if (claimer) my_cell = make_uint4(s_run_rec[l_head[my_run]], (uint32_t)key, (uint32_t)(key >> 32), validmask);
    if (claimer) s_u[g_before + (uint32_t)__popcll(cb & ((1ull << lane) - 1ull))] = my_cell;
    __syncthreads();

    // ---- phase B: lane = (corner, feature); each wave walks a contiguous range of cells ----
    const uint32_t c = lane / F, f = lane % F;
    const uint32_t mi = lane & 15u, mk = lane >> 4;        // MFMA operand index (corner / feature of run A | B), sample slot
    const bool     half_b = (mi & 8u) != 0;
    const float    sx = (mi & 1u) ? 1.0f : -1.0f, ox = (mi & 1u) ? 0.0f : 1.0f;
    const float    sy = (mi & 2u) ? 1.0f : -1.0f, oy = (mi & 2u) ? 0.0f : 1.0f;
    const float    sz = (mi & 4u) ? 1.0f : -1.0f, oz = (mi & 4u) ? 0.0f : 1.0f;
    auto flush = [&](uint32_t row, float v) {
        const size_t at = (size_t)row * F + f;
        if (mask_on) {
            const float e = emb[at];
            if (!(e >= -1.0f && e <= 1.0f)) return;
        }
        unsafeAtomicAdd(grad_emb + at, v);
    };
    // row of my corner in cell k (the arithmetic of Corners::setup, level geometry is block-uniform)
    uint32_t stride = 1, sd[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        sd[d] = stride;
        if (stride <= hs) stride *= R;
    }
    const bool hashed = stride > hs, pow2 = (hs & (hs - 1)) == 0;
    constexpr uint32_t primes[3] = {1u, 2654435761u, 805459861u};
    // the cell is wave-uniform: both candidates per axis (cell, cell + 1 clamped; times the stride or the
    // hash prime) are scalar work, the lane only selects by its corner bits
    const bool cx = (c & 1u) != 0, cy = (c & 2u) != 0, cz = (c & 4u) != 0;
    const uint32_t m1 = hashed ? primes[1] : sd[1], m2 = hashed ? primes[2] : sd[2];
    auto row_of = [&](uint32_t k_lo, uint32_t k_hi) -> uint32_t {
        const uint32_t gx = k_lo & 0xFFFFu, gy = k_lo >> 16, gz = k_hi & 0xFFFFu;
        const uint32_t x0 = gx, x1 = min(gx + 1, R - 1);
        const uint32_t y0 = gy * m1, y1 = min(gy + 1, R - 1) * m1;
        const uint32_t z0 = gz * m2, z1 = min(gz + 1, R - 1) * m2;
        const uint32_t px = cx ? x1 : x0, py = cy ? y1 : y0, pz = cz ? z1 : z0;
        uint32_t index = hashed ? (px ^ py ^ pz) : (px + py + pz);
        if (pow2) index &= hs - 1;
        else if (index >= hs) index %= hs;
        return off + index;
    };

    // Rows two consecutive cells share (a ray leaving through a face) are NOT combined in registers here as
    // k_grid_encode_bwd does: with the cells merged across rays the atomics are no longer what the time is
    // made of, and the match (cell delta, partner lane, one more cross-lane read per cell) cost more issue
    // slots than the saved requests: 0.330 -> 0.305 ms for the 9 coarse levels, 1.091 -> 1.060 ms for the
    // whole backward call next to the bin / owner passes.
    for (uint32_t g = wave; g < n_cells; g += kMW) {
        const uint4 cell = s_u[g];                         // {first run of the chain, key, valid corners}
        uint32_t    rec = cell.x;
        const uint32_t k_lo = __builtin_amdgcn_readfirstlane(cell.y);
        const uint32_t k_hi = __builtin_amdgcn_readfirstlane(cell.z);
        // S[corner][feature] = sum over the chain's samples of w[corner] * g[feature]: a K = n product of
        // an 8 x n and an n x 8 matrix on v_mfma_f32_16x16x4_f32.  Lane l feeds sample slot l / 16 with
        // operand index l % 16: its corner's weight (rebuilt from the 3 fractions — once per (corner, sample)
        // instead of once per (corner, feature, sample) as a lane-per-output loop would) and its feature's
        // gradient.  Two runs per step: operand rows / columns 0..7 carry run A, 8..15 run B (the next run
        // of the chain, or the second half of a lone run), so the tile's two diagonal 8 x 8 blocks are two
        // partial sums and every lane has work (8 samples per instruction; the off-diagonal A x B cross
        // terms are dropped).  The loop is wave-uniform: every lane walks the same chain.
        f32x4 S = {0.0f, 0.0f, 0.0f, 0.0f};
        for (;;) {
            uint32_t a0 = rec & 0x3FFu, a1 = (rec >> 10) & 0x7FFu, b0, b1, nxt = rec >> 21;
            if (nxt != END) {
                const uint32_t rb = s_run_rec[nxt];
                b0 = rb & 0x3FFu, b1 = (rb >> 10) & 0x7FFu, nxt = rb >> 21;
            } else {
                b1 = a1;
                a1 = b0 = min(a1, a0 + ((((a1 - a0 + 1u) >> 1) + 3u) & ~3u));
            }
            const uint32_t steps = max(a1 - a0, b1 - b0);
            const uint32_t m0 = half_b ? b0 : a0, m1 = half_b ? b1 : a1;
            for (uint32_t p = 0; p < steps; p += 4) {
                const uint32_t ps = m0 + p + mk;
                float          a = 0.0f, bv = 0.0f;
                if (ps < m1) {
                    const float4 q = *reinterpret_cast<const float4*>(s_w4[ps]);
                    // bit ? frac : 1 - frac, as one fma with (+1, 0) or (-1, 1): exact either way
                    const float wx = __builtin_fmaf(q.x, sx, ox), wy = __builtin_fmaf(q.y, sy, oy),
                                wz = __builtin_fmaf(q.z, sz, oz);
                    a = ((wx * wy) * wz) * q.w;
                    bv = s_g[ps][mi & 7u];
                }
                S = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, S, 0, 0, 0);
            }
            if (nxt == END) break;
            rec = s_run_rec[nxt];
        }
        // tile element (row, col) sits in lane col + 16 * (row / 4), register row % 4; block B is 8 rows
        // and 8 columns further on = 40 lanes
        const int   src = (int)(f + 16u * (c >> 2));
        const float e0 = __shfl(S[0], src) + __shfl(S[0], src + 40), e1 = __shfl(S[1], src) + __shfl(S[1], src + 40),
                    e2 = __shfl(S[2], src) + __shfl(S[2], src + 40), e3 = __shfl(S[3], src) + __shfl(S[3], src + 40);
        if ((cell.w >> c) & 1u)
            flush(row_of(k_lo, k_hi), (c & 2u) ? ((c & 1u) ? e3 : e2) : ((c & 1u) ? e1 : e0));
    }
}

// grid_encode.hip launches this for the coarse half of a binned call (D = 3, F = 8)
void launch_bwd_merge(const float* grad, const float* inputs, const float* emb, const int32_t* offsets,
                      const int32_t* resolutions, float* grad_emb, uint32_t N, uint32_t L,
                      const uint32_t* clip_count, FeatLayout lay, bool ste, hipStream_t s)
{
    lay.n_slots = L;
    // (round 3 measured this kernel with padded dynamic LDS — one block per CU, to leave room for the owner waves of
    // the binned levels: slower; the switch is gone, the library keeps no state between calls)
    const bool small = (uint64_t)div_up(N, 1024u) * L < 4096u;       // fewer than eight rounds of 1024-sample blocks
    if (small) {
        const dim3 grid(div_up(N, 512u) * L);
        if (ste) hipLaunchKernelGGL((k_grid_encode_bwd_merge<true, 512>), grid, dim3(512), 0, s, grad, inputs, emb, offsets, resolutions, grad_emb, N, clip_count, lay);
        else hipLaunchKernelGGL((k_grid_encode_bwd_merge<false, 512>), grid, dim3(512), 0, s, grad, inputs, emb, offsets, resolutions, grad_emb, N, clip_count, lay);
    } else {
        const dim3 grid(div_up(N, 1024u) * L);
        if (ste) hipLaunchKernelGGL((k_grid_encode_bwd_merge<true, 1024>), grid, dim3(1024), 0, s, grad, inputs, emb, offsets, resolutions, grad_emb, N, clip_count, lay);
        else hipLaunchKernelGGL((k_grid_encode_bwd_merge<false, 1024>), grid, dim3(1024), 0, s, grad, inputs, emb, offsets, resolutions, grad_emb, N, clip_count, lay);
    }
}

}  // namespace cnc
