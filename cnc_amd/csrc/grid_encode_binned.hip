// grid_encode_binned.hip — embedding-gradient scatter of the finest levels WITHOUT global atomics.
//
// Same result as kernel_grid_backward (gridencoder.cu:399-585) for the levels it is given; the
// coarser levels of the same call stay on the run-aggregated atomic kernel of grid_encode.hip.
//
// Why: on MI355X global fp32 atomics retire ~21 G (instruction, 64-byte segment) requests/s
// wherever the line lives (tools/atomic_probe.hip), and at the finest levels every sample sits in
// its own cell: 4 (y, z) corner pairs -> ~6 requests per (sample, level) that no amount of run
// merging removes.  tools/owner_probe.hip measured the alternative built here at 0.15 ms per
// 2^20-sample level against 0.31 ms for the atomic scatter:
//
//   pass 1  k_bwd_bin    every (sample, level) emits one 16-byte item per (dy, dz) corner pair —
//                        sample index, the two corner weights w/sum(w), the two slab-local rows —
//                        into the bin of the 256-row table slab that owns the pair's rows (two
//                        items if the x-neighbours straddle a slab edge).  Per workgroup the items are
//                        counted in an LDS histogram, space is reserved with ONE global atomic per
//                        (workgroup, non-empty bin), and the items are written at LDS-ranked slots.
//   pass 2  k_bwd_owner  one wave owns one slab (a bin far above the mean load is shared by several
//                        waves, which then add their partial slabs atomically): 256 rows x F floats
//                        of accumulators in LDS.  It
//                        streams its bin (items coalesced, the sample's gradient row gathered one
//                        batch ahead; the weights travel in the item, computed in pass 1 exactly as
//                        the scatter kernel computes them) and adds into LDS with plain
//                        read-modify-writes.  Claims of one batch on the same row are serialised
//                        with tickets from a per-row LDS counter (integer atomic), one round per
//                        ticket — LDS fp32 atomics were measured 5x slower than this.  The slab is then added to the gradient
//                        table with coalesced 16-byte accesses (STE mask applied there).
//
// A bin that fills up sends the excess items down the atomic path inside pass 1, so skewed inputs
// stay correct.  Summation order differs from the atomic kernel's (as it does between any two
// runs of that kernel); values agree to fp32 rounding.
#include "common.hpp"
#include "encoder_common.hpp"

namespace cnc {

constexpr uint32_t kSlabLog2 = 8;                  // rows per owner wave (<= 12: item row fields)
constexpr uint32_t kSlab = 1u << kSlabLog2;
constexpr uint32_t kBinSamplesPerThread = 4;       // pass 1: 4096 samples per 1024-thread block
constexpr uint32_t kMaxBins = 4096;                // LDS histogram size (level_rows <= 2^20)
constexpr uint32_t kHeadBytes = 16;                // workspace bytes per bin counter (4 used)

// One corner pair of one sample at one level, as the owner wave needs it.  16 bytes, so a wave reads
// 64 items as one contiguous KiB and only the gradient row is left to gather (carrying just the sample
// index and recomputing the weights in pass 2 cost a second 64-byte sector per item for the point:
// PMC showed that pass HBM-bound at 175 B fetched per item).
struct Item {
    uint32_t sample;
    float    w0, w1;      // weight / sum of valid weights of corner x and corner x+1
    uint32_t rows;        // r0 | r1 << 12 | mask << 24; mask bit 0: add row r0, bit 1: add row r1
};
static_assert(sizeof(Item) == 16, "Item is read as one dwordx4");

struct BinnedArgs {
    const float*    grad;
    const float*    inputs;
    const float*    emb;
    const int32_t*  offsets;
    const int32_t*  resolutions;
    float*          grad_emb;
    uint32_t        N;
    uint32_t        first_level;     // binned levels are [first_level, first_level + gridDim.y)
    uint32_t        bins;            // slabs per level = ceil(level_rows / 256)
    uint32_t        cap;             // item slots per bin
    uint32_t        part;            // items one owner wave takes (a fuller bin is shared by several waves)
    uint32_t*       bin_count;       // [n_binned][bins]
    Item*           items;           // [n_binned][bins][cap]
    const uint32_t* clip_count;
    FeatLayout      lay;
};

template <uint32_t F, bool STE>
__device__ __forceinline__ void atomic_row(const BinnedArgs& a, bool mask_on, uint32_t abs_row,
                                           float tw, const float* __restrict__ g)
{
#pragma unroll
    for (uint32_t f = 0; f < F; f++) {
        const size_t at = (size_t)abs_row * F + f;
        if (mask_on) {
            const float e = a.emb[at];
            if (!(e >= -1.0f && e <= 1.0f)) continue;
        }
        unsafeAtomicAdd(a.grad_emb + at, tw * g[f]);
    }
}

template <uint32_t F, bool STE>
__global__ __launch_bounds__(1024) void k_bwd_bin(BinnedArgs a)
{
    __shared__ uint32_t s_cnt[kMaxBins];
    const uint32_t slot = a.first_level + blockIdx.y;
    const uint32_t off = (uint32_t)a.offsets[slot];
    const uint32_t hs = (uint32_t)a.offsets[slot + 1] - off;
    const uint32_t R = (uint32_t)a.resolutions[slot];
    // a level with more rows than the caller sized the bins for cannot be binned: all atomics
    const bool     binnable = div_up(hs, kSlab) <= a.bins;
    const bool     mask_on = STE && (a.clip_count == nullptr || *a.clip_count != 0);
    uint32_t*      bin_count = a.bin_count + (size_t)blockIdx.y * a.bins;
    Item*          items = a.items + (size_t)blockIdx.y * a.bins * a.cap;
    const uint32_t base_i = blockIdx.x * 1024 * kBinSamplesPerThread;

    for (uint32_t b = threadIdx.x; b < a.bins; b += 1024) s_cnt[b] = 0;
    __syncthreads();

    // ---- count ----
    if (binnable) {
#pragma unroll
        for (uint32_t k = 0; k < kBinSamplesPerThread; k++) {
            const uint32_t i = base_i + k * 1024 + threadIdx.x;
            float x[3];
            if (i < a.N && load_point<3>(a.inputs, i, x)) {
                Corners<3, false> c;
                c.setup(x, R, hs, 0, nullptr);
#pragma unroll
                for (uint32_t p = 0; p < 4; p++) {
                    const bool     v0 = c.valid[2 * p], v1 = c.valid[2 * p + 1];
                    const uint32_t b0 = c.row[2 * p] >> kSlabLog2, b1 = c.row[2 * p + 1] >> kSlabLog2;
                    if (v0) atomicAdd(&s_cnt[b0], 1u);
                    if (v1 && !(v0 && b1 == b0)) atomicAdd(&s_cnt[b1], 1u);
                }
            }
        }
    }
    __syncthreads();
    // ---- reserve: one global atomic per non-empty bin of this block ----
    for (uint32_t b = threadIdx.x; b < a.bins; b += 1024) {
        const uint32_t n = s_cnt[b];
        s_cnt[b] = n ? atomicAdd(&bin_count[b], n) : 0u;
    }
    __syncthreads();
    // ---- emit ----
#pragma unroll
    for (uint32_t k = 0; k < kBinSamplesPerThread; k++) {
        const uint32_t i = base_i + k * 1024 + threadIdx.x;
        float x[3];
        if (!(i < a.N && load_point<3>(a.inputs, i, x))) continue;
        Corners<3, false> c;
        c.setup(x, R, hs, 0, nullptr);
        uint32_t spill = 0;   // corners whose item found no room (or level not binnable)
#pragma unroll
        for (uint32_t p = 0; p < 4; p++) {
            const bool     v0 = c.valid[2 * p], v1 = c.valid[2 * p + 1];
            const uint32_t b0 = c.row[2 * p] >> kSlabLog2, b1 = c.row[2 * p + 1] >> kSlabLog2;
            if (!binnable) {
                spill |= (v0 ? 1u : 0u) << (2 * p) | (v1 ? 1u : 0u) << (2 * p + 1);
                continue;
            }
            const bool together = v0 && v1 && b1 == b0;
            Item it;
            it.sample = i;
            it.w0 = c.w[2 * p] * c.wn_re;
            it.w1 = c.w[2 * p + 1] * c.wn_re;
            const uint32_t lr = (c.row[2 * p] & (kSlab - 1)) | (c.row[2 * p + 1] & (kSlab - 1)) << 12;
            if (v0) {
                const uint32_t at = atomicAdd(&s_cnt[b0], 1u);
                it.rows = lr | (together ? 3u : 1u) << 24;
                if (at < a.cap) items[(size_t)b0 * a.cap + at] = it;
                else spill |= (together ? 3u : 1u) << (2 * p);
            }
            if (v1 && !together) {
                const uint32_t at = atomicAdd(&s_cnt[b1], 1u);
                it.rows = lr | 2u << 24;
                if (at < a.cap) items[(size_t)b1 * a.cap + at] = it;
                else spill |= 2u << (2 * p);
            }
        }
        // the slow path: spilled corners are added atomically
        if (spill) {
            constexpr uint32_t V = F < 4 ? F : 4;
            float        g[F];
            const float* gp = a.grad + feat_index(a.lay, slot, a.N, i, F);
#pragma unroll
            for (uint32_t q = 0; q < F; q += V) {
                float gv[V];
                load_vec<V>(gp + q, gv);
#pragma unroll
                for (uint32_t j = 0; j < V; j++) g[q + j] = gv[j];
            }
#pragma unroll
            for (uint32_t q = 0; q < 8; q++)
                if ((spill >> q) & 1u)
                    atomic_row<F, STE>(a, mask_on, off + c.row[q], c.w[q] * c.wn_re, g);
        }
    }
}

// ---- pass 1, second form: the block's items leave in BIN ORDER ------------------------------------------------
// k_bwd_bin above lets every lane store its own items: the 8 or so items a block adds to one bin are contiguous in
// memory but written by different lanes of different waves at different times, and the L2 merges only part of them
// into full sectors — rocprofv3 counts 9.6 M write requests per 7-level pass for 235 MB of items (1.5 items per
// 64-byte request), and with everything else overlapped the whole backward call sits on the L2 -> fabric request
// rate (53 M requests in 1.06 ms = 50 G/s, the ceiling of tools/fetch_calib.hip).  Here the block sorts the SOURCE
// IDS of its items by bin in LDS (a 16-bit id per item: 40 KB where the 16-byte payloads would need 256 KB) and then
// walks the sorted list: lane t rebuilds item t from its sample (the corner set-up again: vector ALU is what this
// pass has to spare) and stores it, so that a wave-instruction writes 64 consecutive slots of the sorted order —
// whole (block, bin) runs, 128 contiguous bytes on average.
constexpr uint32_t kSortBins = 2048;                                  // three LDS arrays of that many words
#ifndef CNC_SORT_SAMPLES_PER_THREAD
#define CNC_SORT_SAMPLES_PER_THREAD 4
#endif
constexpr uint32_t kSortSamplesPerThread = CNC_SORT_SAMPLES_PER_THREAD;   // 4096 samples per block: runs of 8 items per bin
constexpr uint32_t kSortItems = 1024 * kSortSamplesPerThread * 5;     // source ids per block (typically 4.2 per sample)

// item of corner pair p (half 0: the x corner's bin, carrying both corners when they share it; half 1: the x+1
// corner's bin when it differs) — exactly what k_bwd_bin emits
__device__ __forceinline__ Item pair_item(const Corners<3, false>& c, uint32_t sample, uint32_t p, uint32_t half, uint32_t& bin,
                                          uint32_t& mask)
{
    // p may be a run-time value: pick the pair with selects (indexing the corner arrays dynamically would send them
    // to scratch memory)
    bool     v0 = false, v1 = false;
    uint32_t r0 = 0, r1 = 0;
    float    w0 = 0, w1 = 0;
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) {
        const bool hit = q == p;
        v0 = hit ? c.valid[2 * q] : v0;
        v1 = hit ? c.valid[2 * q + 1] : v1;
        r0 = hit ? c.row[2 * q] : r0;
        r1 = hit ? c.row[2 * q + 1] : r1;
        w0 = hit ? c.w[2 * q] : w0;
        w1 = hit ? c.w[2 * q + 1] : w1;
    }
    const uint32_t b0 = r0 >> kSlabLog2, b1 = r1 >> kSlabLog2;
    const bool     together = v0 && v1 && b1 == b0;
    Item           it;
    it.sample = sample;
    it.w0 = w0 * c.wn_re;
    it.w1 = w1 * c.wn_re;
    mask = half ? 2u : (together ? 3u : 1u);
    bin = half ? b1 : b0;
    it.rows = (r0 & (kSlab - 1)) | (r1 & (kSlab - 1)) << 12 | mask << 24;
    return it;
}

template <uint32_t F, bool STE>
__global__ __launch_bounds__(1024) void k_bwd_bin_sorted(BinnedArgs a)
{
    __shared__ uint32_t s_start[kSortBins];       // items of this block per bin, then their first position in s_src
    __shared__ uint32_t s_base[kSortBins];        // first slot of the block's run inside the bin (global reservation)
    __shared__ uint16_t s_src[kSortItems];        // source ids in bin order: sample-in-block | pair << 13 | half << 15
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_total;
    const uint32_t slot = a.first_level + blockIdx.y;
    const uint32_t off = (uint32_t)a.offsets[slot];
    const uint32_t hs = (uint32_t)a.offsets[slot + 1] - off;
    const uint32_t R = (uint32_t)a.resolutions[slot];
    const bool     binnable = div_up(hs, kSlab) <= a.bins;
    const bool     mask_on = STE && (a.clip_count == nullptr || *a.clip_count != 0);
    uint32_t*      bin_count = a.bin_count + (size_t)blockIdx.y * a.bins;
    Item*          items = a.items + (size_t)blockIdx.y * a.bins * a.cap;
    const uint32_t base_i = blockIdx.x * 1024 * kSortSamplesPerThread;
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t V = F < 4 ? F : 4;

    auto spill_corners = [&](const Corners<3, false>& c, uint32_t i, uint32_t corners) {
        float        g[F];
        const float* gp = a.grad + feat_index(a.lay, slot, a.N, i, F);
#pragma unroll
        for (uint32_t q = 0; q < F; q += V) {
            float gv[V];
            load_vec<V>(gp + q, gv);
#pragma unroll
            for (uint32_t j = 0; j < V; j++) g[q + j] = gv[j];
        }
#pragma unroll
        for (uint32_t q = 0; q < 8; q++)
            if ((corners >> q) & 1u) atomic_row<F, STE>(a, mask_on, off + c.row[q], c.w[q] * c.wn_re, g);
    };

    if (!binnable) {      // a level with more rows than the bins were sized for: every corner goes to atomics
#pragma unroll
        for (uint32_t k = 0; k < kSortSamplesPerThread; k++) {
            const uint32_t i = base_i + k * 1024 + tid;
            float x[3];
            if (!(i < a.N && load_point<3>(a.inputs, i, x))) continue;
            Corners<3, false> c;
            c.setup(x, R, hs, 0, nullptr);
            uint32_t all = 0;
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) all |= (c.valid[q] ? 1u : 0u) << q;
            if (all) spill_corners(c, i, all);
        }
        return;
    }

    for (uint32_t b = tid; b < kSortBins; b += 1024) s_start[b] = 0;
    __syncthreads();
    // ---- count: every item takes its rank inside its bin; (bin, rank) stay in registers ----
    uint32_t key[kSortSamplesPerThread][8];
#pragma unroll
    for (uint32_t k = 0; k < kSortSamplesPerThread; k++) {
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) key[k][q] = 0xFFFFFFFFu;
        const uint32_t i = base_i + k * 1024 + tid;
        float x[3];
        if (i < a.N && load_point<3>(a.inputs, i, x)) {
            Corners<3, false> c;
            c.setup(x, R, hs, 0, nullptr);
#pragma unroll
            for (uint32_t p = 0; p < 4; p++) {
                const bool     v0 = c.valid[2 * p], v1 = c.valid[2 * p + 1];
                const uint32_t b0 = c.row[2 * p] >> kSlabLog2, b1 = c.row[2 * p + 1] >> kSlabLog2;
                if (v0) key[k][2 * p] = b0 | atomicAdd(&s_start[b0], 1u) << 11;
                if (v1 && !(v0 && b1 == b0)) key[k][2 * p + 1] = b1 | atomicAdd(&s_start[b1], 1u) << 11;
            }
        }
    }
    __syncthreads();
    // ---- reserve (one global atomic per non-empty bin of this block) and exclusive prefix over the bins ----
    const uint32_t n0 = s_start[2 * tid], n1 = s_start[2 * tid + 1];
    s_base[2 * tid] = (n0 && 2 * tid < a.bins) ? atomicAdd(&bin_count[2 * tid], n0) : 0u;
    s_base[2 * tid + 1] = (n1 && 2 * tid + 1 < a.bins) ? atomicAdd(&bin_count[2 * tid + 1], n1) : 0u;
    uint32_t incl = n0 + n1;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if ((tid & 63) >= d) incl += up;
    }
    if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; w++) before += (w < (tid >> 6)) ? s_wave[w] : 0u;
    const uint32_t excl = before + incl - (n0 + n1);
    s_start[2 * tid] = excl;
    s_start[2 * tid + 1] = excl + n0;
    if (tid == 1023) s_total = before + incl;
    __syncthreads();
    const uint32_t total = s_total;
    if (total > kSortItems) {
        // more items than the id list holds (straddling pairs everywhere): this block stores lane by lane, as
        // k_bwd_bin does
#pragma unroll
        for (uint32_t k = 0; k < kSortSamplesPerThread; k++) {
            const uint32_t i = base_i + k * 1024 + tid;
            float x[3];
            if (!(i < a.N && load_point<3>(a.inputs, i, x))) continue;
            Corners<3, false> c;
            c.setup(x, R, hs, 0, nullptr);
            uint32_t spill = 0;
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) {
                if (key[k][q] == 0xFFFFFFFFu) continue;
                uint32_t bin, mask;
                const Item it = pair_item(c, i, q >> 1, q & 1u, bin, mask);
                const uint32_t at = s_base[bin] + (key[k][q] >> 11);
                if (at < a.cap) items[(size_t)bin * a.cap + at] = it;
                else spill |= mask << (q & ~1u);
            }
            if (spill) spill_corners(c, i, spill);
        }
        return;
    }
    // ---- source ids into bin order ----
#pragma unroll
    for (uint32_t k = 0; k < kSortSamplesPerThread; k++)
#pragma unroll
        for (uint32_t q = 0; q < 8; q++)
            if (key[k][q] != 0xFFFFFFFFu)
                s_src[s_start[key[k][q] & 0x7FFu] + (key[k][q] >> 11)] = (uint16_t)((k * 1024 + tid) | (q >> 1) << 13 | (q & 1u) << 15);
    __syncthreads();
    // ---- walk the sorted list: consecutive lanes, consecutive slots ----
    for (uint32_t pos = tid; pos < total; pos += 1024) {
        const uint32_t src = s_src[pos];
        const uint32_t i = base_i + (src & 0x1FFFu);
        float x[3];
        load_point<3>(a.inputs, i, x);
        Corners<3, false> c;
        c.setup(x, R, hs, 0, nullptr);
        uint32_t bin, mask;
        const Item it = pair_item(c, i, (src >> 13) & 3u, (src >> 15) & 1u, bin, mask);
        const uint32_t at = s_base[bin] + (pos - s_start[bin]);
        if (at < a.cap) items[(size_t)bin * a.cap + at] = it;
        else spill_corners(c, i, mask << (2 * ((src >> 13) & 3u)));
    }
}

// LDS accumulators are stored by 16-byte chunk: acc[chunk][row][4 floats].  With row-major
// [row][F] a 16-byte access of random rows only ever touches every other 16-byte bank group (the
// first halves of 32-byte rows), doubling the conflicts of the read-modify-write that bounds pass 2.
template <uint32_t F>
__device__ __forceinline__ uint32_t acc_index(uint32_t row, uint32_t q)
{
    constexpr uint32_t V = F < 4 ? F : 4;
    return ((q / V) * kSlab + row) * V + (q % V);
}

// read-modify-write of one accumulator row (F floats) in LDS
template <uint32_t F>
__device__ __forceinline__ void lds_row_add(float* __restrict__ acc, uint32_t row, const float (&v)[F])
{
    constexpr uint32_t V = F < 4 ? F : 4;
    using T = typename vecf<V>::type;
#pragma unroll
    for (uint32_t q = 0; q < F; q += V) {
        T*     p = reinterpret_cast<T*>(acc + acc_index<F>(row, q));
        T      t = *p;
        float* f = reinterpret_cast<float*>(&t);
#pragma unroll
        for (uint32_t j = 0; j < V; j++) f[j] += v[q + j];
        *p = t;
    }
}

template <uint32_t F, bool STE>
__global__ __launch_bounds__(64) void k_bwd_owner(BinnedArgs a)
{
    constexpr uint32_t V = F < 4 ? F : 4;
    __shared__ __attribute__((aligned(16))) float s_acc[kSlab * F];
    __shared__ __attribute__((aligned(16))) uint32_t s_cnt[kSlab];
    static_assert(kSlab * sizeof(uint32_t) == 64 * sizeof(uint4), "one uint4 per lane clears the tickets");
    // blockIdx.x = part * bins + bin: the waves that take the second, third ... `part` items of a bin are
    // scheduled after every bin's first wave — they exist only for bins far above the mean (thin slices of
    // space hash unevenly onto the slabs: up to 7.7x the mean at resolution 296 on the first chunk of the
    // bench frame, where one wave per bin made the pass 2x slower than on the other chunks)
    const uint32_t bin = blockIdx.x % a.bins, part = blockIdx.x / a.bins, lane = threadIdx.x;
    const uint32_t slot = a.first_level + blockIdx.y;
    const uint32_t off = (uint32_t)a.offsets[slot];
    const uint32_t hs = (uint32_t)a.offsets[slot + 1] - off;
    if (bin * kSlab >= hs) return;
    uint32_t n = a.bin_count[(size_t)blockIdx.y * a.bins + bin];
    n = n < a.cap ? n : a.cap;
    if (n <= part * a.part) return;     // nothing (more) landed here: the table slab stays as it is
    const bool  shared_slab = n > a.part;
    const Item* my = a.items + ((size_t)blockIdx.y * a.bins + bin) * a.cap + (size_t)part * a.part;
    n = min(n - part * a.part, a.part);

    for (uint32_t k = lane * 4; k < kSlab * F; k += 64 * 4)     // 16-byte LDS stores
        *reinterpret_cast<float4*>(s_acc + k) = make_float4(0, 0, 0, 0);

    // Two loads deep: while batch i is accumulated, the gradient rows of batch i+1 and the items of
    // batch i+2 are in flight (the gradient address needs the item, so a one-deep prefetch waits for
    // the two latencies back to back in every iteration — the pass was bound by exactly that).
    // A batch takes kRun consecutive items from each of 64 / kRun regions of the bin rather
    // than 64 consecutive ones: items of one cell arrive clustered (neighbouring rays of one pass-1
    // block), and every extra claim on a row costs the batch another round.
    constexpr uint32_t kRun = 8, kRegions = 64 / kRun;         // consecutive items per region and batch
    const uint32_t region = (div_up(n, kRegions) + kRun - 1) & ~(kRun - 1);   // items per region
    uint32_t       step = 0;                                   // item batches requested so far
    uint4          it2 = make_uint4(0, 0, 0, 0), it1 = make_uint4(0, 0, 0, 0);
    bool           valid2 = false, valid1 = false;
    float          g1[F];
    auto           fetch_items = [&]() {
        const uint32_t in_region = step * kRun + (lane % kRun);
        const uint32_t j = (lane / kRun) * region + in_region;
        valid2 = in_region < region && j < n;
        if (valid2) it2 = *reinterpret_cast<const uint4*>(my + j);
        step++;
    };
    auto fetch_grad = [&]() {
        if (valid1) {
            const float* gp = a.grad + feat_index(a.lay, slot, a.N, it1.x, F);
#pragma unroll
            for (uint32_t q = 0; q < F; q += V) {
                float gv[V];
                load_vec<V>(gp + q, gv);
#pragma unroll
                for (uint32_t t = 0; t < V; t++) g1[q + t] = gv[t];
            }
        }
    };
    fetch_items();
    it1 = it2;
    valid1 = valid2;
    fetch_items();
    fetch_grad();
    __syncthreads();   // single wave: orders the zero-fill before the first accumulate

    while (__ballot(valid1) != 0) {
        uint32_t pend = 0, r0 = 0, r1 = 0;
        float    v0[F], v1[F];
        if (valid1) {
            pend = (it1.w >> 24) & 3u;
            r0 = it1.w & 0xFFFu;
            r1 = (it1.w >> 12) & 0xFFFu;
            const float w0 = __uint_as_float(it1.y), w1 = __uint_as_float(it1.z);
#pragma unroll
            for (uint32_t f = 0; f < F; f++) {
                v0[f] = w0 * g1[f];
                v1[f] = w1 * g1[f];
            }
        }
        it1 = it2;
        valid1 = valid2;
        fetch_items();
        fetch_grad();
        // Claims on the same row must not share an LDS instruction.  Every claim takes a ticket from
        // a per-row counter (one integer LDS atomic); round k serves the claims holding ticket k, so
        // all rows touched in a round are distinct and its reads and writes need no ordering among
        // themselves.  Rounds = the largest number of claims on one row.  The x and the x+1 rows of
        // the items are served in two phases with their own tickets: 64 instead of 128 claims on the
        // 256 rows per phase (64 random claims already put 2-3 on one row), and a round is 4 instead
        // of 8 16-byte LDS operations — 8.1 half-rounds per batch where the joint scheme needed 6.5
        // full ones (marched rays, 7 levels).  LDS instructions of a wave execute in order, which
        // orders the counter reset, the tickets and the rounds.
        reinterpret_cast<uint4*>(s_cnt)[lane] = make_uint4(0, 0, 0, 0);
        asm volatile("" ::: "memory");
        const int32_t t0 = (pend & 1u) ? (int32_t)atomicAdd(&s_cnt[r0], 1u) : -1;
        for (int32_t k = 0; __ballot(t0 >= k) != 0; k++) {
            if (t0 == k) lds_row_add<F>(s_acc, r0, v0);
            asm volatile("" ::: "memory");
        }
        reinterpret_cast<uint4*>(s_cnt)[lane] = make_uint4(0, 0, 0, 0);
        asm volatile("" ::: "memory");
        const int32_t t1 = (pend & 2u) ? (int32_t)atomicAdd(&s_cnt[r1], 1u) : -1;
        for (int32_t k = 0; __ballot(t1 >= k) != 0; k++) {
            if (t1 == k) lds_row_add<F>(s_acc, r1, v1);
            asm volatile("" ::: "memory");
        }
    }
    __syncthreads();

    // ---- slab -> gradient table ----
    const bool     mask_on = STE && (a.clip_count == nullptr || *a.clip_count != 0);
    const uint32_t rows = min(kSlab, hs - bin * kSlab);
    const size_t   base = ((size_t)off + (size_t)bin * kSlab) * F;
    constexpr uint32_t NI = kSlab * F / (64 * V);     // 8 at F = 8: all loads issued before the first use
    if (shared_slab) {
        // several waves hold partial sums of this slab: coalesced atomics instead of the read-modify-write
#pragma unroll
        for (uint32_t i = 0; i < NI; i++) {
            const uint32_t k = (i * 64 + lane) * V;
            if (k < rows * F) {
                float e[V];
                if (mask_on) load_vec<V>(a.emb + base + k, e);
#pragma unroll
                for (uint32_t q = 0; q < V; q++)
                    if (!mask_on || (e[q] >= -1.0f && e[q] <= 1.0f))
                        unsafeAtomicAdd(a.grad_emb + base + k + q, s_acc[acc_index<F>(k / F, k % F + q)]);
            }
        }
        return;
    }
    // this wave is the only writer of these rows in this pass
    float t[NI][V], e[NI][V];
#pragma unroll
    for (uint32_t i = 0; i < NI; i++) {
        const uint32_t k = (i * 64 + lane) * V;
        if (k < rows * F) {
            load_vec<V>(a.grad_emb + base + k, t[i]);
            if (mask_on) load_vec<V>(a.emb + base + k, e[i]);
        }
    }
#pragma unroll
    for (uint32_t i = 0; i < NI; i++) {
        const uint32_t k = (i * 64 + lane) * V;
        if (k < rows * F) {
#pragma unroll
            for (uint32_t q = 0; q < V; q++) {
                const bool pass = !mask_on || (e[i][q] >= -1.0f && e[i][q] <= 1.0f);
                t[i][q] += pass ? s_acc[acc_index<F>(k / F, k % F + q)] : 0.0f;
            }
            store_vec<V>(a.grad_emb + base + k, t[i]);
        }
    }
}

template <uint32_t F>
static void launch_binned(const BinnedArgs& a, uint32_t n_binned, bool ste, bool sorted, hipStream_t s)
{
    sorted = sorted && a.bins <= kSortBins;
    const dim3 g1(div_up(a.N, 1024 * (sorted ? kSortSamplesPerThread : kBinSamplesPerThread)), n_binned),
        g2(a.bins * div_up(a.cap, a.part), n_binned);
    if (ste) {
        if (sorted) hipLaunchKernelGGL((k_bwd_bin_sorted<F, true>), g1, dim3(1024), 0, s, a);
        else hipLaunchKernelGGL((k_bwd_bin<F, true>), g1, dim3(1024), 0, s, a);
        hipLaunchKernelGGL((k_bwd_owner<F, true>), g2, dim3(64), 0, s, a);
    } else {
        if (sorted) hipLaunchKernelGGL((k_bwd_bin_sorted<F, false>), g1, dim3(1024), 0, s, a);
        else hipLaunchKernelGGL((k_bwd_bin<F, false>), g1, dim3(1024), 0, s, a);
        hipLaunchKernelGGL((k_bwd_owner<F, false>), g2, dim3(64), 0, s, a);
    }
}

}  // namespace cnc

using namespace cnc;

// mean bin load: 4 items per sample and level spread over the slabs
static uint64_t mean_load(uint32_t N, uint32_t bins) { return (4ull * N + bins - 1) / bins; }

// 8x the mean bin load, at least one batch.  Samples from a thin slice of space hash unevenly onto the slabs
// (the high bits of a row come from (y, z) alone): the fullest bin held 7.7x the mean on the first chunk
// of the bench frame at resolution 296, 2.4x in the middle of the frame.
static uint32_t default_cap(uint32_t N, uint32_t bins)
{
    const uint64_t cap = 8 * mean_load(N, bins) + 64;
    return (uint32_t)(cap > 0xFFFFFFFFull ? 0xFFFFFFFFull : cap);
}

// items per owner wave: 2x the mean (one wave per bin unless the bin is that far above it), whole batches
static uint32_t owner_part(uint32_t N, uint32_t bins, uint64_t cap)
{
    uint64_t part = (2 * mean_load(N, bins) + 63) / 64 * 64;     // 1.25x / 1.5x / 3x: the same; 1x: 13 % slower
    if (part < 64) part = 64;
    while ((cap + part - 1) / part > 16) part *= 2;      // a roomy caller-sized workspace: at most 16 waves per bin
    return (uint32_t)(part > 0xFFFFFFFFull ? 0xFFFFFFFFull : part);
}

extern "C" uint64_t cnc_grid_encode_backward_binned_workspace(uint32_t N, uint32_t n_binned,
                                                              uint32_t level_rows)
{
    if (n_binned == 0 || level_rows == 0) return 0;
    const uint64_t bins = div_up(level_rows, kSlab);
    return (uint64_t)n_binned * bins * (kHeadBytes + (uint64_t)default_cap(N, (uint32_t)bins) * sizeof(Item));
}

extern "C" int cnc_grid_encode_backward_binned(const float* grad, const float* inputs,
                                               const float* embeddings, const int32_t* offsets,
                                               const int32_t* resolutions, float* grad_embeddings,
                                               uint32_t N, uint32_t D, uint32_t F, uint32_t L,
                                               uint32_t flags, const uint32_t* ste_clip_count,
                                               uint32_t grad_ld, uint32_t grad_col,
                                               uint32_t n_binned, uint32_t level_rows,
                                               void* workspace, uint64_t workspace_bytes,
                                               void* stream)
{
    if (N == 0 || L == 0) return CNC_OK;
    if (!grad || !inputs || !embeddings || !offsets || !resolutions || !grad_embeddings)
        return CNC_ERR_INVALID_VALUE;
    if (n_binned > L) return CNC_ERR_INVALID_VALUE;
    // what is not binned goes through the atomic kernel, with the same arguments
    if (L - n_binned > 0) {
        const int rc = cnc_grid_encode_backward(grad, inputs, embeddings, offsets, resolutions,
                                                grad_embeddings, N, D, F, L - n_binned, 0, nullptr,
                                                nullptr, nullptr, nullptr,
                                                flags | CNC_FLAG_LEVELS_FINEST_FIRST, ste_clip_count,
                                                nullptr, nullptr, nullptr, grad_ld, grad_col, stream);
        if (rc != CNC_OK) return rc;
    }
    if (n_binned == 0) return CNC_OK;
    if (D != 3 || !(F == 2 || F == 4 || F == 8)) return CNC_ERR_UNSUPPORTED;
    if (grad_ld != 0) {
        const uint32_t V = F < 4 ? F : 4;
        if (grad_col + L * F > grad_ld || grad_ld % V || grad_col % V) return CNC_ERR_INVALID_VALUE;
    } else if (grad_col != 0) {
        return CNC_ERR_INVALID_VALUE;
    }
    const uint32_t bins = div_up(level_rows, kSlab);
    if (bins == 0 || bins > kMaxBins || !workspace) return CNC_ERR_INVALID_VALUE;
    // layout: [heads x 16 B: bin counters, padded so the items stay 16-byte aligned][heads x cap items]
    const uint64_t heads = (uint64_t)n_binned * bins;
    if ((uintptr_t)workspace % 16 != 0) return CNC_ERR_INVALID_VALUE;
    if (workspace_bytes < heads * (kHeadBytes + 64 * sizeof(Item))) return CNC_ERR_INVALID_VALUE;   // one batch per bin
    uint64_t cap = (workspace_bytes - heads * kHeadBytes) / (heads * sizeof(Item));
    if (cap > 0x0FFFFFFFull) cap = 0x0FFFFFFFull;

    hipStream_t s = (hipStream_t)stream;
    uint32_t*   ws = (uint32_t*)workspace;
    if (hipMemsetAsync(ws, 0, heads * 4, s) != hipSuccess) return CNC_ERR_LAUNCH;
    BinnedArgs a{grad, inputs, embeddings, offsets, resolutions, grad_embeddings, N, L - n_binned,
                 bins, (uint32_t)cap, owner_part(N, bins, cap), ws, (Item*)((char*)workspace + heads * kHeadBytes), ste_clip_count,
                 FeatLayout{grad_ld, grad_col}};
    const bool ste = (flags & CNC_FLAG_STE_BINARY) != 0;
    const bool sorted = (flags & CNC_FLAG_BIN_LANE_STORES) == 0;
    switch (F) {
    case 2: launch_binned<2>(a, n_binned, ste, sorted, s); break;
    case 4: launch_binned<4>(a, n_binned, ste, sorted, s); break;
    default: launch_binned<8>(a, n_binned, ste, sorted, s); break;
    }
    return launch_status();
}
