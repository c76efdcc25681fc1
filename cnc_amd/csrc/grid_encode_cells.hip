// grid_encode_cells.hip — backward scatter for the calls of a TRAINING STEP: the equal cells of a block merged in LDS,
// with the occupancy mask and per-point level windows, for volumes and planes.
//
// k_grid_encode_bwd (grid_encode.hip) merges consecutive samples of one cell; k_grid_encode_bwd_merge
// (grid_encode_merge.hip) the equal cells of a 1024-sample block, but only for the unmasked 3-D levels of a bench
// frame.  What a step of the full model issues (tools/dump_bwd_calls.py + tools/count_bwd_calls.py, round 6, F = 8;
// M (row, 32 B) updates per call, blocks of 1024 points):
//
//   call                                     corner refs   runs of a cell   distinct cells x 2^D   distinct rows
//   render pass, 3-D, 269 k samples x 12         25.8           8.8                8.7                 5.3
//   render pass, one plane, x 4                   4.3           2.7                2.7                 1.8
//   context pass, 3-D, 749 k vertices x 3        18.0          17.8               10.5                 5.8
//   context pass, a plane's levels, 633 k x 3     6.5           2.3                0.68                0.23
//   context pass, a plane's vote table, x 1       2.5           1.7                0.91                0.37
//
// The render pass's samples are ~37 k short rays from random pixels: runs along a ray are all there is to merge and
// the run kernel has them.  The context pass's points are lattice vertices in hash-slot order: consecutive ones never
// share a cell, but a block of them does (x-neighbours sit in neighbouring hash slots — prime[0] = 1 — and a coarse
// cell holds many fine vertices).  These calls sit on the memory side's atomic request rate (21 G (instruction, 64-byte
// segment) pairs per second), so the number of distinct cells per block is what their time is made of.
//
// Why cells and not rows.  A table keyed by ROW (every corner of every point added into an LDS accumulator, one
// atomic per touched row at the end) was built first: LDS atomics — float adds and compare-and-swaps alike — retire
// ~0.3 lanes per clock and CU on this part (1.3 ms for the 206 M adds of the render pass's 3-D call against 0.31 ms
// for the whole run kernel; docs/engineering_log.md has the same figure from the owner pass's probe).  The merge here
// costs TWO LDS atomics per run (claim the cell, chain the run); the sums themselves are plain register
// accumulations by the lanes that own (corner, feature).
//
// Structure (one workgroup = MB consecutive points of one level slot):
//   A  lane = point: Corners<D, VXL> (mask bit planes, per-point level), key = level | cell; fractions + 1/sum of
//      valid weights and the gradient row go to LDS; a zero gradient row drops the point.
//   B  runs of equal consecutive keys (ballot prefix), runs of equal cells chained through an LDS hash table.
//   C  lanes = (corner, feature), 64 / (2^D F) cells per wave at a time: walk the cell's chain, rebuild the corner
//      weight from the fractions (same products, same order as Corners::setup), accumulate, ONE atomic per (cell,
//      corner row) with the x-neighbour rows on adjacent lanes.
#include "common.hpp"
#include "encoder_common.hpp"

namespace cnc {

template <uint32_t D, uint32_t F, bool VXL, bool STE, uint32_t MB>
__global__ __launch_bounds__(MB) void k_grid_encode_bwd_cells(const CellsArgs a)
{
    constexpr uint32_t C = 1u << D;
    constexpr uint32_t kW = MB / 64u;
    constexpr uint32_t kSlots = 2u * MB, END = 0x7FFu;
    constexpr uint32_t kLv = 64;                        // levels a key can name (6 bits)
    constexpr uint64_t kKeyBits = (1ull << 56) - 1ull;  // key = x | y << 16 | z << 32 | level << 48; a cell record adds
                                                        // the valid corners in bits 56..63
    static_assert(MB <= 1024 && (MB & (MB - 1)) == 0, "run records pack start (10 bits) / end (11) / next (11)");
    static_assert(D >= 2 && D <= 3 && 2 * F <= 64, "planes and volumes; a pair of x-neighbour rows fits a wave");
    __shared__ __attribute__((aligned(16))) float s_w4[MB][4];      // fractions [0, D), 1 / (sum of valid weights) at [3]
    __shared__ float    s_g[MB][F];
    __shared__ uint64_t s_key[MB];              // A-B: the points' keys; C: one record per distinct cell
    __shared__ uint32_t h_slot[kSlots];         // hash table: representative run + 1 of a cell (0 = empty); read again in C
    __shared__ uint16_t s_run_start[MB + 1];    // B: first point of a run; C: cell index of a representative run
    __shared__ uint32_t l_head[MB];             // B: per representative run, the last run chained to its cell; C: per cell,
                                                //    the record of the first run to walk
    __shared__ uint32_t s_run_rec[MB];          // start | end << 10 | next run of the cell << 21
    __shared__ uint32_t s_wave_heads[kW], s_wave_claims[kW], s_wave_lmax[kW];
    // per-point level windows: the geometry of the levels this block's points use (a cell's rows then need LDS reads only)
    __shared__ uint32_t s_lv_off[kLv + 1], s_lv_res[kLv];

    const CellsArgs& j = a;
    const uint32_t  tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const bool      mask_on = STE && (j.clip_count == nullptr || *j.clip_count != 0);
    // level slot as the fast block index, last slot first (resident blocks spread over the levels, the fine ones first)
    const uint32_t chunk = blockIdx.x / a.L;
    const uint32_t slot = a.L - 1u - blockIdx.x % a.L;
    const uint32_t b = chunk * MB + tid;
    const bool     per_point = j.mli != nullptr;

    for (uint32_t i = tid; i < kSlots; i += MB) h_slot[i] = 0;
    l_head[tid] = END;

    // ---- A: lane = point ----
    uint64_t key = ~0ull;
    uint32_t validmask = 0, my_level = 0;
    {
        float x[D];
        if (b < a.N && load_point<D>(j.inputs, b, x)) {
            const uint32_t level = slot + (per_point ? (uint32_t)j.mli[b] : 0u);
            my_level = level;
            const uint32_t off = (uint32_t)j.offsets[level];
            const uint32_t hs = (uint32_t)j.offsets[level + 1] - off;
            const uint32_t R = (uint32_t)j.resolutions[level];
            Corners<D, VXL> c;
            c.setup(x, R, hs, a.Rb, j.vxl, j.sat, vertex_plane(j.lay, level));
            constexpr uint32_t V = F < 4 ? F : 4;
            const float* gp = j.grad + feat_index(j.lay, slot, a.N, b, F);
            float        g[F];
            bool         nonzero = false;
#pragma unroll
            for (uint32_t k = 0; k < F; k += V) {
                float gv[V];
                load_vec<V>(gp + k, gv);
#pragma unroll
                for (uint32_t q = 0; q < V; q++) {
                    g[k + q] = gv[q];
                    nonzero |= gv[q] != 0.0f;
                }
            }
#pragma unroll
            for (uint32_t i = 0; i < C; i++) validmask |= (c.valid[i] ? 1u : 0u) << i;
            // a zero gradient row adds nothing (the levels outside a vertex's context window when several windows share
            // one call); neither does a point none of whose corners is valid
            if (nonzero && validmask != 0 && level >= kLv) {
                // a level the key cannot name (never the case for the encoders CNC builds: 12 to 16 levels): this
                // point's corners go out one by one
#pragma unroll
                for (uint32_t i = 0; i < C; i++) {
                    if (!c.valid[i]) continue;
                    const size_t at = (size_t)(off + c.row[i]) * F;
                    for (uint32_t k = 0; k < F; k++) {
                        if (mask_on && !(j.emb[at + k] >= -1.0f && j.emb[at + k] <= 1.0f)) continue;
                        unsafeAtomicAdd(j.grad_emb + at + k, (c.w[i] * c.wn_re) * g[k]);
                    }
                }
            } else if (nonzero && validmask != 0) {
                key = (uint64_t)level << 48 | (uint64_t)c.cell[0] | (uint64_t)c.cell[1] << 16;
                if constexpr (D == 3) key |= (uint64_t)c.cell[2] << 32;
                *reinterpret_cast<float4*>(s_w4[tid]) = make_float4(c.frac[0], c.frac[1], D == 3 ? c.frac[D - 1] : 0.0f, c.wn_re);
#pragma unroll
                for (uint32_t k = 0; k < F; k++) s_g[tid][k] = g[k];
            }
        }
        s_key[tid] = key;
    }
    {                                           // highest level of the block (wave reduction, one LDS word per wave)
        uint32_t m = my_level;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
        if (lane == 0) s_wave_lmax[wave] = m;
    }
    __syncthreads();
    uint32_t lmax = 0;
#pragma unroll
    for (uint32_t w = 0; w < kW; w++) lmax = max(lmax, s_wave_lmax[w]);
    if (tid <= lmax + 1u && tid <= kLv) s_lv_off[tid] = (uint32_t)j.offsets[tid];
    if (tid <= lmax && tid < kLv) s_lv_res[tid] = (uint32_t)j.resolutions[tid];

    // ---- B: runs of consecutive points with one key, chained per cell ----
    const bool     head = tid == 0 || s_key[tid - 1] != key;
    const uint64_t hb = __ballot(head);
    if (lane == 0) s_wave_heads[wave] = (uint32_t)__popcll(hb);
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kW; w++) {
        const uint32_t h = s_wave_heads[w];
        before += w < wave ? h : 0u;
        total += h;
    }
    const uint32_t my_run = before + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull));
    if (head) s_run_start[my_run] = (uint16_t)tid;
    if (tid == 0) s_run_start[total] = (uint16_t)MB;
    __syncthreads();

    auto slot_of = [](uint64_t k) -> uint32_t {
        return (((uint32_t)k ^ (uint32_t)(k >> 16) ^ (uint32_t)(k >> 32) ^ (uint32_t)(k >> 45)) * 2654435761u)
               >> (32 - __builtin_ctz(kSlots));
    };
    bool claimer = false;
    if (head && key != ~0ull) {
        uint32_t sl = slot_of(key);
        uint32_t rep;
        for (;;) {
            const uint32_t seen = atomicCAS(&h_slot[sl], 0u, my_run + 1);
            if (seen == 0) { claimer = true; rep = my_run; break; }
            rep = seen - 1;
            if (s_key[s_run_start[rep]] == key) break;
            sl = (sl + 1) & (kSlots - 1);
        }
        const uint32_t next = atomicExch(&l_head[rep], my_run);
        s_run_rec[my_run] = tid | (uint32_t)s_run_start[my_run + 1] << 10 | next << 21;
    }
    const uint64_t cb = __ballot(claimer);
    if (lane == 0) s_wave_claims[wave] = (uint32_t)__popcll(cb);
    __syncthreads();
    uint32_t g_before = 0, n_cells = 0;
#pragma unroll
    for (uint32_t w = 0; w < kW; w++) {
        const uint32_t h = s_wave_claims[w];
        g_before += w < wave ? h : 0u;
        n_cells += h;
    }
    // the cells, compacted: record (key | valid corners) where the keys were, the chain's entry where l_head was, and the
    // representative run -> cell index map where the run starts were (all three are dead: the sync above ended the claims)
    const uint32_t my_gi = g_before + (uint32_t)__popcll(cb & ((1ull << lane) - 1ull));
    const uint32_t my_first = claimer ? s_run_rec[l_head[my_run]] : 0u;
    __syncthreads();
    if (claimer) {
        s_key[my_gi] = key | (uint64_t)validmask << 56;
        l_head[my_gi] = my_first;
        s_run_start[my_run] = (uint16_t)my_gi;
    }
    __syncthreads();

    // ---- C: lane = (cell, x bit, feature): 64 / (2 F) cells per wave at a time, 2^(D-1) corners per lane ----
    // One cell per wave (lanes = (corner, feature)) made a cell cost a chain of dependent LDS round trips — record,
    // samples, next run, geometry — that nothing overlapped: 0.23 ms for the 1.3 M cells of the context pass's 3-D call
    // at full occupancy.  Here a wave walks CPW cells side by side; a lane keeps the 2^(D-1) corners that share its x
    // bit, so an atomic instruction still carries whole 32-byte rows with the two x-neighbours on adjacent lane groups.
    //
    // The x-neighbour carry.  Two cells that are neighbours along x share 2^(D-1) vertices; lattice vertices in hash-slot
    // order arrive in x-pairs (prime[0] = 1: the vertices x and x ^ 1 sit in neighbouring slots), so in the context pass
    // nearly every cell of a block has such a neighbour in the block.  A lane looks its neighbour up in the hash table
    // (x + 1 for the lanes of the upper x bit, x - 1 for the lower), and the shared vertex goes out ONCE, from the cell
    // whose other column holds the vertex's partner in its 64-byte segment (rows 2k and 2k + 1: the partner column is
    // x ^ 1 on a hashed level, the row index's parity decides on a dense one): that side walks the neighbour's samples
    // too, the other side drops the corner.  Both sides evaluate the same predicate on the same vertex, so exactly one
    // of them writes it — and writes it next to its segment partner: one request per touched segment.
    constexpr uint32_t CPW = 64u / (2u * F), NA = C / 2u, kCells = kW * CPW;
    const uint32_t ck = lane / (2u * F), xbit = (lane / F) & 1u, f = lane % F;
    const float    sx = xbit ? 1.0f : -1.0f, ox = xbit ? 0.0f : 1.0f;
    float* const       grad_emb = j.grad_emb;
    const float* const emb = j.emb;
    // level geometry: block-uniform without a per-point level window, else looked up per cell (the level is in the key)
    uint32_t u_off = 0, u_hs = 1, u_R = 2;
    if (!per_point && slot < kLv) {
        u_off = s_lv_off[slot];
        u_hs = s_lv_off[slot + 1] - u_off;
        u_R = s_lv_res[slot];
    }
    const bool carry = a.carry != 0;
    constexpr uint32_t primes[3] = {1u, 2654435761u, 805459861u};

    // One sample of a chain into NA partial sums, the corners' x weight being fma(frac_x, s, o)
    auto add_sample = [&](uint32_t p, float s, float o, float (&acc)[NA]) {
        const float4 q = *reinterpret_cast<const float4*>(s_w4[p]);
        const float  gv = s_g[p][f];
        // bit ? frac : 1 - frac as one fma with (+1, 0) or (-1, 1): exact either way; the products in the order of
        // Corners::setup ((wx wy) wz), then the 1 / sum factor, then the gradient
        const float wx = __builtin_fmaf(q.x, s, o);
        const float wy[2] = {__builtin_fmaf(q.y, -1.0f, 1.0f), q.y};
        const float wz[2] = {__builtin_fmaf(q.z, -1.0f, 1.0f), q.z};
#pragma unroll
        for (uint32_t i = 0; i < NA; i++) {
            float w = wx * wy[i & 1u];
            if constexpr (D == 3) w = w * wz[i >> 1];
            acc[i] += (w * q.w) * gv;
        }
    };
    // my cell's chain and (where I own shared vertices) my neighbour's, in lock step: two independent strings of LDS
    // round trips in flight instead of one after the other.  A record of END << 21 is an empty chain.
    auto walk2 = [&](uint32_t ra, uint32_t rb, float (&acc)[NA], float (&nacc)[NA]) {
        uint32_t pa = ra & 0x3FFu, ea = (ra >> 10) & 0x7FFu, na = ra >> 21;
        uint32_t pb = rb & 0x3FFu, eb = (rb >> 10) & 0x7FFu, nb2 = rb >> 21;
        for (;;) {
            const bool la = pa < ea, lb = pb < eb;
            if (la) add_sample(pa, sx, ox, acc);
            if (lb) add_sample(pb, -sx, 1.0f - ox, nacc);
            pa += la ? 1u : 0u;
            pb += lb ? 1u : 0u;
            const bool more_a = pa >= ea && na != END, more_b = pb >= eb && nb2 != END;
            if (more_a) {
                const uint32_t r = s_run_rec[na];
                pa = r & 0x3FFu, ea = (r >> 10) & 0x7FFu, na = r >> 21;
            }
            if (more_b) {
                const uint32_t r = s_run_rec[nb2];
                pb = r & 0x3FFu, eb = (r >> 10) & 0x7FFu, nb2 = r >> 21;
            }
            if (pa >= ea && pb >= eb) break;
        }
    };

    for (uint32_t base = wave * CPW; base < n_cells; base += kCells) {
        const uint32_t gi = base + ck;
        const bool     live = gi < n_cells;
        const uint64_t crec = live ? s_key[gi] : 0ull;
        const uint32_t c_lo = (uint32_t)crec, c_hi = (uint32_t)(crec >> 32);
        const uint32_t g0 = c_lo & 0xFFFFu, g1 = c_lo >> 16, g2 = c_hi & 0xFFFFu, valid = c_hi >> 24;
        // my x-neighbour's cell: probe for key +- 1 (the x coordinate is the key's low 16 bits)
        uint32_t nb = END;
        if (live && carry && (xbit || g0 != 0u)) {
            const uint64_t want = (crec & kKeyBits) + (xbit ? 1ull : ~0ull);
            uint32_t       sl = slot_of(want);
            for (;;) {
                const uint32_t seen = h_slot[sl];
                if (seen == 0) break;
                const uint32_t cand = s_run_start[seen - 1];
                if ((s_key[cand] & kKeyBits) == want) { nb = cand; break; }
                sl = (sl + 1) & (kSlots - 1);
            }
        }
        uint32_t off = u_off, hs = u_hs, R = u_R;
        if (per_point) {
            const uint32_t level = (c_hi >> 16) & (kLv - 1u);
            off = s_lv_off[level];
            hs = s_lv_off[level + 1] - off;
            R = s_lv_res[level];
        }
        if (!live) continue;
        const uint32_t rec_own = l_head[gi], rec_nb = nb != END ? l_head[nb] : (END << 21);
        // rows of my corners (the arithmetic of grid_row, per-axis parts shared by the corners)
        uint32_t stride = 1, sd[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            sd[d] = stride;
            if (stride <= hs) stride *= R;
        }
        const bool     hashed = stride > hs, pow2 = (hs & (hs - 1u)) == 0;
        const uint32_t m1 = hashed ? primes[1] : sd[1], m2 = hashed ? primes[2] : sd[D - 1];
        const uint32_t vx = xbit ? min(g0 + 1u, R - 1u) : g0;                 // (prime[0] = 1 = the dense x stride)
        const uint32_t py[2] = {g1 * m1, min(g1 + 1u, R - 1u) * m1};
        const uint32_t pz[2] = {g2 * m2, min(g2 + 1u, R - 1u) * m2};
        uint32_t index[NA];
        uint32_t emit = 0, take = 0;            // corners I write; of those, the ones whose neighbour's share I add
#pragma unroll
        for (uint32_t i = 0; i < NA; i++) {
            uint32_t ix = hashed ? (vx ^ py[i & 1u]) : (vx + py[i & 1u]);
            if constexpr (D == 3) ix = hashed ? (ix ^ pz[i >> 1]) : (ix + pz[i >> 1]);
            if (pow2) ix &= hs - 1u;
            else if (ix >= hs) ix %= hs;
            index[i] = ix;
            if (!((valid >> (xbit | (i << 1))) & 1u)) continue;
            bool mine = true;
            if (nb != END) {
                // the shared vertex's segment partner is the column to its left?
                const bool partner_left = hashed ? (vx & 1u) != 0 : (ix & 1u) != 0;
                mine = (xbit != 0) == partner_left;
                take |= (mine ? 1u : 0u) << i;
            }
            emit |= (mine ? 1u : 0u) << i;
        }
        float acc[NA], nacc[NA];
#pragma unroll
        for (uint32_t i = 0; i < NA; i++) acc[i] = nacc[i] = 0.0f;
        walk2(rec_own, take ? rec_nb : (END << 21), acc, nacc);
#pragma unroll
        for (uint32_t i = 0; i < NA; i++)
            if ((take >> i) & 1u) acc[i] += nacc[i];
#pragma unroll
        for (uint32_t i = 0; i < NA; i++) {
            if (!((emit >> i) & 1u)) continue;
            const size_t at = (size_t)(off + index[i]) * F + f;
            if (mask_on) {   // STE_binary.backward: pass gradient only where |param| <= 1
                const float e = emb[at];
                if (!(e >= -1.0f && e <= 1.0f)) continue;
            }
            unsafeAtomicAdd(grad_emb + at, acc[i]);
        }
    }
}

template <uint32_t D, uint32_t F, bool VXL, bool STE>
static void launch_cells_t(const CellsArgs& a, hipStream_t s)
{
    // 1024 points per workgroup (74 KB of LDS at F = 8, two workgroups per CU) unless that leaves the chip short of
    // workgroups
    const bool small = (uint64_t)div_up(a.N, 1024u) * a.L < 2048u;
    if (small) {
        const dim3 grid(div_up(a.N, 512u) * a.L);
        hipLaunchKernelGGL((k_grid_encode_bwd_cells<D, F, VXL, STE, 512>), grid, dim3(512), 0, s, a);
    } else {
        const dim3 grid(div_up(a.N, 1024u) * a.L);
        hipLaunchKernelGGL((k_grid_encode_bwd_cells<D, F, VXL, STE, 1024>), grid, dim3(1024), 0, s, a);
    }
}

template <uint32_t D, uint32_t F>
static void launch_cells_flags(const CellsArgs& a, bool ste, hipStream_t s)
{
    const bool vxl = a.vxl != nullptr;
    if (vxl && ste) launch_cells_t<D, F, true, true>(a, s);
    else if (vxl) launch_cells_t<D, F, true, false>(a, s);
    else if (ste) launch_cells_t<D, F, false, true>(a, s);
    else launch_cells_t<D, F, false, false>(a, s);
}

// D in {2, 3}, F in {2, 4, 8}, resolutions below 2^16; false = not built for this shape (the caller keeps its own kernel)
bool launch_bwd_cells(const CellsArgs& a, uint32_t D, uint32_t F, bool ste, hipStream_t s)
{
    if ((uint64_t)div_up(a.N, 512u) * a.L >= (1ull << 31)) return false;
#define CNC_CELLS(DD, FF)                            \
    if (D == DD && F == FF) {                        \
        launch_cells_flags<DD, FF>(a, ste, s);       \
        return true;                                 \
    }
    CNC_CELLS(3, 8) CNC_CELLS(3, 4) CNC_CELLS(3, 2) CNC_CELLS(2, 8) CNC_CELLS(2, 4) CNC_CELLS(2, 2)
#undef CNC_CELLS
    return false;
}

}  // namespace cnc
