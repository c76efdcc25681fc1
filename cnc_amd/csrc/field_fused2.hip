// field_fused2.hip — the gradient-free radiance field (positions -> density -> rgb, ngp.py:506-547) with TWO cooperating
// waves per 32-sample tile.
//
// field_fused.hip gives one wave a whole tile: the 32 x H accumulators of the widest layer (H = 160: 80 registers) plus a
// K-step of weight fragments (80 more, double-buffered) put that kernel at 256 registers = two waves per SIMD, and with
// two waves the sign-plane gathers of the feature fill (an L2 round trip per unit) were what it waited for: 42 % of the
// wave cycles in s_waitcnt, 13-19 % MFMA busy (profiles/r04_mfma_utilisation.md).  Here a workgroup is two waves and a
// tile is split between them:
//   * fill: thread (sample i = tid & 31, window q = tid >> 5) computes the 8 columns [8 q, 8 q + 8) of sample i's row of
//     the current 32-column chunk — half the gather work per wave — into a double-buffered chunk tile (two half planes:
//     x = hi + lo), one workgroup barrier per chunk;
//   * matrix products on v_mfma_f32_16x16x32_f16, three per term as in field_fused.hip (hi hi + hi lo + lo hi).  A layer
//     whose width is H is split by COLUMNS: wave w owns output columns [w H / 2, (w + 1) H / 2) of all 32 rows =
//     2 row blocks x H / 32 column blocks x 4 = 40 accumulator registers at H = 160, and loads only its half of the
//     weight fragments (the workgroup reads each fragment once, as before).  The narrow layers (H -> 1 + geo, H -> 3)
//     are split by ROWS: wave w owns row block w and every column block;
//   * activations go between the layers through the workgroup's LDS planes (the A operand needs all K columns of a
//     row: both waves' halves), two barriers per layer.
// Half the accumulators and half the weight registers per wave: three to four waves per SIMD instead of two.
//
// Values: the features are those of k_grid_encode_fwd_bits (bit-identical: same Corners / fmaf chain); the layers are
// the three-product scheme of field_fused.hip with a different summation order (k in steps of 32 instead of 16).
// The fp16 range guard (field_fused_common.hpp) covers every value that is split into halves here.
#include "field_mma.hpp"

#include <type_traits>

namespace cnc {

constexpr uint32_t kMaxUnits = 64;        // rows of the unit table kept in LDS (the host refuses more)

// LDS layout in halves: [4 chunk buffers of two planes | colour: the activation planes alias them, + 32 floats] [unit table]
template <int NT, bool RGB>
__host__ __device__ constexpr uint32_t kUnitTableAt()
{
    uint32_t n = 4 * 32 * kCP;                                              // two chunk buffers of two planes
    if (n < 2 * 32 * 20 * 2) n = 2 * 32 * 20 * 2;                           // the density epilogue's partial sums (floats)
    if (RGB) {
        const uint32_t planes = 2 * 32 * Plane2<NT>::ld + 64;
        if (planes > n) n = planes;
    }
    return (n + 7u) & ~7u;
}

// x = acc / 2^8 + bias (ReLU) of a column-split layer -> the two half planes.  Lane (r, kq) holds sample rb * 16 + r,
// output features 16 (cb0 + cb) + 4 kq + v: four halves = 8 bytes inside one 16-byte chunk of the (swizzled) row.
// `save` (the gradient pass's forward): the same values as float32 into rows [row0, row0 + 32) of an [n_rows, 32 NT]
// matrix, 16 bytes per lane and column block.
template <int NCB, int NT, bool SAVE = false>
__device__ __forceinline__ void acc_to_planes(half_t* __restrict__ d_hi, half_t* __restrict__ d_lo,
                                              const float* __restrict__ bias, const f32x4 (&acc)[2][NCB], uint32_t cb0,
                                              uint32_t lane, float& mx, float* __restrict__ save = nullptr, uint32_t row0 = 0,
                                              uint32_t n_rows = 0)
{
    using P = Plane2<NT>;
    const uint32_t r = lane & 15u, kq = lane >> 4;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        const uint32_t col0 = (cb0 + cb) * 16u + 4u * kq;
        const float4   b4 = *reinterpret_cast<const float4*>(bias + col0);
        const float    bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            half4_t xh, xl;
            float   xs[4];
#pragma unroll
            for (int v = 0; v < 4; v++) {
                float x = __builtin_fmaf(acc[rb][cb][v], kWScaleInv, bb[v]);
                mx = fmaxf(mx, x);                                   // mx >= 0: the maximum of the ReLU's outputs
                // the gradient pass has no exact-fp32 kernel behind it: a value beyond fp16's range saturates (same
                // instruction count: one v_med3 for the v_max) and the guard word reports it
                x = SAVE ? __builtin_amdgcn_fmed3f(x, 0.0f, kHalfMax) : (x > 0 ? x : 0);
                half_t h, l;
                split_half(x, h, l);
                xh[v] = h;
                xl[v] = l;
                xs[v] = x;
            }
            if constexpr (SAVE) {
                const uint32_t row = row0 + rb * 16u + r;
                if (row < n_rows) store_vec<4>(save + (size_t)row * (32u * NT) + col0, xs);
            }
            const uint32_t at = P::at(rb * 16u + r, col0);
            *reinterpret_cast<half4_t*>(d_hi + at) = xh;
            *reinterpret_cast<half4_t*>(d_lo + at) = xl;
        }
    }
}

// MODE 0: the gradient-free evaluator.  1 (density, test hook): + the first layer's input rows into p.dbg_features.
// 2 (colour): the gradient pass's forward — + everything the backward reads into p.save (FieldSave), rows [n_live, N)
// evaluated as points outside the box, values beyond fp16's range saturated instead of recomputed.
template <uint32_t F, int NT, bool RGB, int WPE, int MODE = 0>
__global__ __launch_bounds__(128, WPE) void k_field_fused16w2(FusedFieldArgs p)
{
    constexpr bool DUMP = MODE == 1, SAVE = MODE == 2;
    static_assert(!SAVE || RGB, "the saving variant is the colour kernel");
    using RowT = std::conditional_t<DUMP, RowDump<RowF16>, std::conditional_t<SAVE, RowSave<RowF16>, RowF16>>;
    extern __shared__ float lds[];
    half_t* const lds16 = reinterpret_cast<half_t*>(lds);
    using P = Plane2<NT>;
    constexpr int      NCB = NT;                   // 16-column blocks a wave owns of a layer of width H = 32 NT
    constexpr uint32_t NCBT = 2 * NT;
    constexpr int      NB2 = NT == 5 ? 5 : 4;      // column blocks of the second layer: 1 + geo <= 80 (H = 160) / 64
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63u, r = lane & 15u, kq = lane >> 4;
    const uint32_t fi = tid & 31u, fq = tid >> 5;  // fill: sample of the tile, 8-column window of the chunk
    const uint32_t wu = __builtin_amdgcn_readfirstlane(w);
    half_t* const h_hi = lds16;                    // activation planes (colour variant) — alias the chunk buffers
    half_t* const h_lo = lds16 + 32 * P::ld;
    float* const  dens = reinterpret_cast<float*>(lds16 + 2 * 32 * P::ld);      // colour variant: 32 raw densities
    const uint32_t n_rows = rows_of(p);       // p.N, or a count the device holds (cnc_fused_field_t.n_rows_dev) — a LOCAL: writing
                                              // to the by-value argument block would move all of it into scratch memory
    const uint32_t tiles = (n_rows + 31u) / 32u;
    float amin[3], aext[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        amin[a] = p.aabb[a];
        aext[a] = p.aabb[3 + a] - p.aabb[a];
    }
    // (a flagged weight: the exact-fp32 kernel behind this one computes the call.  The saving form has none behind it
    // and goes on: its outputs are non-finite then, and the caller finds the flag — cnc_field_save_t.)
    if (!SAVE && guard_weights_flagged(p, RGB)) return;
    // the unit table (one 16-byte record per (encoder, level)) in LDS, behind everything else
    uint4* const unit_lds = reinterpret_cast<uint4*>(lds16 + kUnitTableAt<NT, RGB>());
    for (uint32_t u = tid; u < p.n_units; u += 128) unit_lds[u] = p.units[u];
    __syncthreads();
    const UnitTable units{p, unit_lds};
    const uint32_t n_chunks = p.nk16_1 / 2;
    const wrsrc_t  W1 = weight_rsrc(reinterpret_cast<const float*>(p.Wq16[0]), n_chunks * NCBT * 2048u);
    const uint32_t voff1 = lane * 16u + w * NCB * 2048u;
    float mx = 0.0f;
    // Register budget (tools/kernel_resources.py): the weight fragments of a chunk (40 registers) are requested AFTER its
    // features have been computed — in flight across the fill they push the density kernel from 128 to 168 registers
    // (three waves per SIMD instead of four) and the colour kernel into spills.
    constexpr bool kDB = WPE <= 3;                       // hidden layers: two sets of weight registers
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint32_t row0 = tile * 32, frow = row0 + fi;
        const bool     live = frow < n_rows;                                    // the row exists: its results are stored
        const bool     live_in = SAVE ? frow < p.save.n_live : live;         // ... and has a position / direction
        float xu[3] = {-1.0f, -1.0f, -1.0f};
        bool  sel = live_in;
        if (live_in) {
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const float v = (p.pos[(size_t)frow * 3 + a] - amin[a]) / aext[a];
                xu[a] = v;
                sel = sel && v > 0.0f && v < 1.0f;
            }
        }
        if constexpr (SAVE) {
            if (live && fq == 0) {                                           // what k_field_prepare writes, + the plane pairs
                p.save.selector[frow] = sel ? 1 : 0;
#pragma unroll
                for (int a = 0; a < 3; a++) p.save.xyz[(size_t)frow * 3 + a] = xu[a];
                *reinterpret_cast<float2*>(p.save.xy + (size_t)frow * 2) = make_float2(xu[0], xu[1]);
                *reinterpret_cast<float2*>(p.save.xz + (size_t)frow * 2) = make_float2(xu[0], xu[2]);
                *reinterpret_cast<float2*>(p.save.yz + (size_t)frow * 2) = make_float2(xu[1], xu[2]);
            }
        }

        // ---- layer 1: a chunk = 32 feature columns = one K-step; this wave's half of the output columns ----
        // (Two chunks per barrier with both units' gathers in flight before either is consumed was built and measured:
        // 34 more live registers — spills at four waves per SIMD, three waves without — 0.87 / 1.78 ms per 2^20 samples
        // against 0.82 / 1.45 for this loop.)
        f32x4 acc[2][NCB];
        zero_q<2, NCB>(acc);
        half8_t wh[NCB], wl[NCB];
        for (uint32_t c = 0; c < n_chunks; c++) {
            half_t* const c_hi = lds16 + (c & 1u) * (2 * 32 * kCP);
            half_t* const c_lo = c_hi + 32 * kCP;
            {   // what this wave's 16 columns of the chunk hold is a wave-uniform fact: branch on it (scalar), so that a
                // chunk of 3-D units issues the 3-D code only
                RowT trow;
                trow.hi = c_hi + fi * kCP;
                trow.lo = c_lo + fi * kCP;
                if constexpr (DUMP) trow.dbg = live ? p.dbg_features + (size_t)frow * p.dbg_ld + c * 32 : nullptr;
                if constexpr (SAVE) trow.out = live ? p.save.feat + (size_t)frow * p.save.ld_feat + c * 32 : nullptr;
                const uint32_t w0 = c * 32 + 8 * fq, col0 = c * 32 + 16 * wu;
                const uint32_t u_first = col0 / F, u_last = (col0 + 15) / F;
                uint32_t       kind = 3;                                         // 3: mixed -> the general fill
                const uint8_t* plane = nullptr;                                  // the wave's ONE sign plane, if it has one
                if (u_first >= p.n_units) kind = 2;                              // raw coordinates / sinusoids / padding
                else if (u_last < p.n_units) {
                    const uint32_t e_first = __builtin_amdgcn_readfirstlane(p.units[u_first].w);
                    const uint32_t e_last = __builtin_amdgcn_readfirstlane(p.units[u_last].w);
                    kind = e_last == 0 ? 0u : (e_first != 0 ? 1u : 3u);          // units are ordered 3-D first
                    // (scalar selects: the pointer stays in SGPRs and the gathers take the scalar-base form)
                    if (e_first == e_last) plane = e_first == 0 ? p.enc[0].bits : e_first == 1 ? p.enc[1].bits
                                                   : e_first == 2 ? p.enc[2].bits : p.enc[3].bits;
                }
                if (kind == 0) fill_units<F, 3, RowT, 8>(p, units, xu, w0, trow, plane);
                else if (kind == 1) fill_units<F, 2, RowT, 8>(p, units, xu, w0, trow, plane);
                else if (kind == 2) fill_tail_window<RowT, 8>(p, xu, w0, p.n_units * F, trow);
                else fill_window<F, false, RowT, 8>(p, xu, w0, trow);
            }
            load_wq<NCB>(W1, c, NCBT, voff1, wh, wl);
            __syncthreads();
            half8_t ah[2], al[2];
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                ah[rb] = *reinterpret_cast<const half8_t*>(c_hi + (rb * 16 + r) * kCP + 8 * kq);
                al[rb] = *reinterpret_cast<const half8_t*>(c_lo + (rb * 16 + r) * kCP + 8 * kq);
            }
            mfma3q<2, NCB>(ah, al, wh, wl, acc);
        }

        if constexpr (!RGB) {
            // density_raw = b2[0] + sum_j relu(h1[j]) W2[0][j]: lane (r, kq) holds features 4 kq + v of its column blocks for
            // samples r and 16 + r: two partial sums per lane, eight lanes-and-waves per sample through LDS
            float part[2] = {0.0f, 0.0f};
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) {
                const uint32_t col0 = (w * NCB + cb) * 16u + 4u * kq;
                const float4   b4 = *reinterpret_cast<const float4*>(p.Bp[0] + col0);
                const float4   w4 = *reinterpret_cast<const float4*>(p.w2row + col0);
                const float    bb[4] = {b4.x, b4.y, b4.z, b4.w}, ww[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int rb = 0; rb < 2; rb++)
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        float x = __builtin_fmaf(acc[rb][cb][v], kWScaleInv, bb[v]);
                        x = x > 0 ? x : 0;
                        part[rb] = __builtin_fmaf(x, ww[v], part[rb]);
                    }
            }
            __syncthreads();                                   // the last chunk has been read by both waves
            // [sample][wave * 4 + kq]: 8 partial sums per sample, 32-byte rows
#pragma unroll
            for (int rb = 0; rb < 2; rb++) lds[(rb * 16 + r) * 8 + w * 4 + kq] = part[rb];
            __syncthreads();
            if (tid < 32) {
                const float4 a4 = *reinterpret_cast<const float4*>(lds + tid * 8), c4 = *reinterpret_cast<const float4*>(lds + tid * 8 + 4);
                float s = a4.x;
                s += a4.y; s += a4.z; s += a4.w; s += c4.x; s += c4.y; s += c4.z; s += c4.w;
                if (live) p.density[frow] = sel ? expf((s + p.Bp[1][0]) - 1.0f) : 0.0f;
            }
            __syncthreads();                                   // before the next tile's fill overwrites the sums
        } else {
            __syncthreads();                                   // the last chunk has been read: the planes alias it
            acc_to_planes<NCB, NT, SAVE>(h_hi, h_lo, p.Bp[0], acc, w * NCB, lane, mx, p.save.h1, row0, n_rows);
            __syncthreads();
            // ---- layer 2 (H -> 1 + geo), split by rows: wave w owns rows [16 w, 16 w + 16) ----
            f32x4 acc2[1][NB2];
            layer_q<1, NB2, NT, kDB>(h_hi, h_lo, NT, p.Wq16[1], NB2, 0, w, acc2, lane);
            const uint32_t Kh = p.nk32_h * 32;                 // head input width: roundup32(17 + geo) <= H
            __syncthreads();                                   // every read of h1 has been issued and waited for
            // Lane (r, kq) holds outputs c = 16 cb + 4 kq + v of sample 16 w + r.  Head-input layout: [SH4 (16) | column 16 + c
            // for output c]: the geo features (c >= 1) land 4-aligned — one 8-byte write per half plane — and column 16
            // receives density_raw (c = 0), against which the packed head weights hold a zero column (cnc_field_pack_t:
            // k_gap = 16).  Outputs past 1 + geo are exact zeros (zero weight rows, zero bias): the padding up to Kh.
#pragma unroll
            for (int cb = 0; cb < NB2; cb++) {
                const uint32_t c0 = cb * 16u + 4u * kq;
                if (16u + c0 >= Kh) continue;
                const float4 b4 = *reinterpret_cast<const float4*>(p.Bp[1] + c0);
                const float  bb[4] = {b4.x, b4.y, b4.z, b4.w};
                half4_t xh, xl;
                float   xs[4];
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    float x = __builtin_fmaf(acc2[0][cb][v], kWScaleInv, bb[v]);
                    if (cb == 0 && v == 0 && kq == 0) dens[w * 16 + r] = x;
                    mx = fmaxf(mx, (c0 + v >= 1u && c0 + v <= p.geo) ? fabsf(x) : 0.0f);
                    // density_raw may be anything finite or not: what goes into its (zero-weight) column is 0
                    x = (c0 + v == 0u) ? 0.0f : x;
                    if constexpr (SAVE) x = __builtin_amdgcn_fmed3f(x, -kHalfMax, kHalfMax);
                    half_t h, l;
                    split_half(x, h, l);
                    xh[v] = h;
                    xl[v] = l;
                    xs[v] = x;
                }
                if constexpr (SAVE) {
                    const uint32_t row = row0 + w * 16u + r;
                    if (row < n_rows) store_vec<4>(p.save.head_in + (size_t)row * p.save.ld_head + 16u + c0, xs);
                }
                const uint32_t at = P::at(w * 16u + r, 16u + c0);
                *reinterpret_cast<half4_t*>(h_hi + at) = xh;
                *reinterpret_cast<half4_t*>(h_lo + at) = xl;
            }
            {   // SH4 of sample fi's direction: thread (fi, fq) writes harmonics 4 fq .. 4 fq + 3
                // (requesting the direction before layer 2, to take its round trip out of this phase, was built: the compiler
                // waits for it on the spot, nothing gained — and that build returned wrong colours for rows 16..31 of about
                // one tile per call, cause not found; tests/test_gpu_field_fused.py::test_fused_field_is_repeatable is the
                // detector that caught it)
                float d3[3] = {0.0f, 0.0f, 1.0f};
                if (live_in) {
#pragma unroll
                    for (int a = 0; a < 3; a++) d3[a] = ((p.dirs[(size_t)frow * 3 + a] + 1.0f) / 2.0f) * 2.0f - 1.0f;
                }
                const float4 v = sh4_quad(fq, d3[0], d3[1], d3[2]);
                float v4[4] = {v.x, v.y, v.z, v.w};
                if (p.sh_fp16) {
#pragma unroll
                    for (int j = 0; j < 4; j++) v4[j] = round_through_half(v4[j]);
                }
                const RowF16 hrow{h_hi, h_lo};
                hrow.put<4>(P::at(fi, 4 * fq), v4);                  // 4 halves inside one 16-byte chunk
                if constexpr (SAVE) {
                    if (live) store_vec<4>(p.save.head_in + (size_t)frow * p.save.ld_head + 4 * fq, v4);
                }
            }
            __syncthreads();
            if (tid < 32 && live) {
                p.density[frow] = sel ? expf(dens[tid] - 1.0f) : 0.0f;
                if constexpr (SAVE) p.save.raw[frow] = dens[tid];
            }
            // ---- head: (16 + geo) -> H -> H -> 3 ----
            layer_q<2, NCB, NT, kDB>(h_hi, h_lo, Kh / 32, p.Wq16[2], NCBT, w * NCB, 0, acc, lane);
            __syncthreads();
            acc_to_planes<NCB, NT, SAVE>(h_hi, h_lo, p.Bp[2], acc, w * NCB, lane, mx, p.save.h3, row0, n_rows);
            __syncthreads();
            layer_q<2, NCB, NT, kDB>(h_hi, h_lo, NT, p.Wq16[3], NCBT, w * NCB, 0, acc, lane);
            __syncthreads();
            acc_to_planes<NCB, NT, SAVE>(h_hi, h_lo, p.Bp[3], acc, w * NCB, lane, mx, p.save.h4, row0, n_rows);
            __syncthreads();
            f32x4 acc5[1][1];
            layer_q<1, 1, NT, kDB>(h_hi, h_lo, NT, p.Wq16[4], 1, 0, w, acc5, lane);
            if (kq == 0) {                                     // outputs 0..2 of sample 16 w + r: 12 contiguous bytes per lane
                const uint32_t row = row0 + w * 16 + r;
                if (row < n_rows) {
#pragma unroll
                    for (int v = 0; v < 3; v++) {
                        const float x = __builtin_fmaf(acc5[0][0][v], kWScaleInv, p.Bp[4][v]);
                        p.rgb[(size_t)row * 3 + v] = 1.0f / (1.0f + expf(-x));
                    }
                }
            }
            __syncthreads();                                   // the next tile's fill overwrites the planes
        }
    }
    guard_raise(p, mx);
}

// W [H, K] -> fragments of the 16x16x32 form: index ((ks * ncb + cb) * 2 + plane) * 512 + lane * 8 + e holds the hi / lo
// half of 2^8 W[16 cb + (lane & 15)][32 ks + 8 (lane >> 4) + e] (zero outside [H, K]).  A weight beyond fp16's range
// stamps `flag` with the pack's id (the range guard).
__device__ __forceinline__ void pack16x32_element(const float* __restrict__ W, uint32_t H, uint32_t K, uint32_t ldw,
                                                  uint32_t ncb, uint32_t idx, half_t* __restrict__ Wq, uint32_t* flag,
                                                  uint32_t pack_id, uint32_t k_gap, uint32_t tflags, uint32_t src_off)
{
    const uint32_t e = idx & 7u, lane = (idx >> 3) & 63u, q = idx >> 9;
    const uint32_t cb = q % ncb, ks = q / ncb;
    const uint32_t out = cb * 16 + (lane & 15u), kp = ks * 32 + 8 * (lane >> 4) + e;
    // k_gap > 0: packed column k_gap is a zero column, the source columns from k_gap on sit one to the right
    const bool     gap = k_gap != 0u && kp == k_gap;
    const uint32_t k = (k_gap != 0u && kp > k_gap) ? kp - 1u : kp;
    // the gradient chain's layers (field_bwd.hip) are the forward's transposed: packed W'[out][k] = W[k][src], with an
    // optional all-zero output 0 (the raw density's slot) in front and a first source column
    const bool     tr = (tflags & CNC_PACK_TRANSPOSE) != 0u, zf = (tflags & CNC_PACK_ZERO_FIRST) != 0u;
    const bool     live = out < H && k < K && !gap && !(zf && out == 0u);
    const uint32_t src = src_off + out - (zf ? 1u : 0u);
    const float    wv = live ? (tr ? W[(size_t)k * ldw + src] : W[(size_t)src * ldw + k]) * 256.0f : 0.0f;
    if (flag && !(fabsf(wv) <= kHalfMax)) atomicMax(flag, pack_id);
    half_t hi, lo;
    split_half(wv, hi, lo);
    Wq[((size_t)q * 2 + 0) * 512 + lane * 8 + e] = hi;
    Wq[((size_t)q * 2 + 1) * 512 + lane * 8 + e] = lo;
}

struct PackAllArgs {
    const float* W[5];
    const float* b[5];
    uint32_t     H[5], K[5], ldw[5];
    uint32_t     nt32[5], nk8[5], nk16[5];      // fp32 fragments (and biases), 32x32x16 half fragments
    uint32_t     ncb[5], nk32[5], k_gap[5];     // 16x16x32 half fragments
    uint32_t     tflags[5], src_off[5];
    float*       Wp[5];
    float*       Bp[5];
    half_t*      Wp16[5];                       // nullable
    half_t*      Wq16[5];                       // nullable
    float*       row0;                          // layer 1's (base.2) row 0, padded to row0_len
    uint32_t     row0_len;
    uint32_t     first_block[6];                // blocks of layer l: [first_block[l], first_block[l + 1])
    uint32_t*    guard;
    uint32_t     pack_id;
};

// All five layers in every fragment order the fused kernels read, in ONE launch (the per-layer entry points cost ten
// launches per optimiser step).
__global__ __launch_bounds__(256) void k_field_pack_all(PackAllArgs a)
{
    uint32_t l = 0;
#pragma unroll
    for (uint32_t j = 1; j < 5; j++) l = blockIdx.x >= a.first_block[j] ? j : l;
    const uint32_t idx = (blockIdx.x - a.first_block[l]) * 256 + threadIdx.x;
    const float*   W = a.W[l];
    const uint32_t H = a.H[l], K = a.K[l], ldw = a.ldw[l];
    if (a.Wp[l]) {   // fp32 fragments, biases, row 0 (k_field_pack_layer)
        const uint32_t NT = a.nt32[l], total = a.nk8[l] * NT * 256;
        if (idx < total) {
            const uint32_t m = idx & 3u, lane = (idx >> 2) & 63u, q = idx >> 8;
            const uint32_t t = q % NT, kb = q / NT;
            const uint32_t out = t * 32 + (lane & 31u), k = kb * 8 + 4 * (lane >> 5) + m;
            a.Wp[l][idx] = (out < H && k < K) ? W[(size_t)out * ldw + k] : 0.0f;
        }
        if (idx < NT * 32) a.Bp[l][idx] = idx < H ? a.b[l][idx] : 0.0f;
        if (l == 1 && a.row0 && idx < a.row0_len) a.row0[idx] = idx < K ? W[idx] : 0.0f;
    }
    uint32_t* const flag = a.guard ? a.guard + 1 + l : nullptr;
    if (a.Wp16[l]) {   // 32x32x16 half fragments (k_field_pack_layer16)
        const uint32_t NT = a.nt32[l], total = a.nk16[l] * NT * 512;
        if (idx < total) {
            const uint32_t e = idx & 7u, lane = (idx >> 3) & 63u, q = idx >> 9;
            const uint32_t t = q % NT, ks = q / NT;
            const uint32_t out = t * 32 + (lane & 31u), k = ks * 16 + 8 * (lane >> 5) + e;
            const float    wv = (out < H && k < K) ? W[(size_t)out * ldw + k] * 256.0f : 0.0f;
            if (flag && !(fabsf(wv) <= kHalfMax)) atomicMax(flag, a.pack_id);
            half_t hi, lo;
            split_half(wv, hi, lo);
            a.Wp16[l][((size_t)q * 2 + 0) * 512 + lane * 8 + e] = hi;
            a.Wp16[l][((size_t)q * 2 + 1) * 512 + lane * 8 + e] = lo;
        }
    }
    if (a.Wq16[l]) {
        const uint32_t total = a.nk32[l] * a.ncb[l] * 512;
        if (idx < total)
            pack16x32_element(W, H, K, ldw, a.ncb[l], idx, a.Wq16[l], flag, a.pack_id, a.k_gap[l], a.tflags[l], a.src_off[l]);
    }
}

template <typename Kern>
static int resident_grid(Kern kern, size_t lds_bytes, uint32_t tiles, uint32_t cap_per_cu, uint32_t* blocks)
{
    // (kernel, device, LDS size) -> resident workgroups: an immutable fact of the hardware and the binary, asked once
    // per thread and variant instead of three runtime calls per launch.  (Every instantiation has the same function
    // type, so the kernel's address is part of the key.)
    struct Slot { const void* kern; int dev; size_t lds; uint32_t n; };
    static thread_local Slot cache[8] = {};
    static thread_local uint32_t next = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return CNC_ERR_LAUNCH;
    const void* const key = reinterpret_cast<const void*>(kern);
    uint32_t n = 0;
    for (auto& s : cache)
        if (s.kern == key && s.dev == dev && s.lds == lds_bytes) n = s.n;
    if (n == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 128, lds_bytes) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu <= 0 || cus <= 0)
            return CNC_ERR_LAUNCH;
        if ((uint32_t)per_cu > cap_per_cu) per_cu = (int)cap_per_cu;
        n = (uint32_t)(per_cu * cus);
        cache[next++ & 7u] = Slot{key, dev, lds_bytes, n};
    }
    *blocks = tiles < n ? tiles : n;
    return CNC_OK;
}

// Launch of the two-wave kernels (called by cnc_field_fused_forward, field_fused.hip).  waves_per_simd: 3 or 4 (the
// register budget the variant was compiled for; the caller's choice, measured in DESIGN.md §4.5).
int launch_field_fused_w2(const FusedFieldArgs& p, bool rgb, uint32_t F, uint32_t H, uint32_t waves_per_simd, hipStream_t s)
{
    const uint32_t NT = H / 32, tiles = (p.N + 31u) / 32u;
    if (p.n_units > kMaxUnits) return CNC_ERR_UNSUPPORTED;
    const uint32_t table_at = NT == 5 ? (rgb ? kUnitTableAt<5, true>() : kUnitTableAt<5, false>())
                                      : (rgb ? kUnitTableAt<2, true>() : kUnitTableAt<2, false>());
    const size_t lds_bytes = (size_t)table_at * sizeof(half_t) + kMaxUnits * sizeof(uint4);
    uint32_t blocks = 0;
    int rc = CNC_OK;
#define CNC_W2(FV, NTV, RGBV, WV)                                                                   \
    do {                                                                                            \
        rc = resident_grid(k_field_fused16w2<FV, NTV, RGBV, WV>, lds_bytes, tiles, 16, &blocks);    \
        if (rc == CNC_OK) hipLaunchKernelGGL((k_field_fused16w2<FV, NTV, RGBV, WV>), dim3(blocks), dim3(128), lds_bytes, s, p); \
    } while (0)
#define CNC_W2_W(FV, NTV, RGBV)                     \
    do {                                            \
        if (waves_per_simd >= 4) CNC_W2(FV, NTV, RGBV, 4); \
        else CNC_W2(FV, NTV, RGBV, 3);              \
    } while (0)
#define CNC_W2_RGB(FV, NTV)                 \
    do {                                    \
        if (p.dbg_features) {               \
            rc = resident_grid(k_field_fused16w2<FV, NTV, false, 3, 1>, lds_bytes, tiles, 16, &blocks); \
            if (rc == CNC_OK) hipLaunchKernelGGL((k_field_fused16w2<FV, NTV, false, 3, 1>), dim3(blocks), dim3(128), lds_bytes, s, p); \
        } else if (p.save.feat) {           \
            rc = resident_grid(k_field_fused16w2<FV, NTV, true, 3, 2>, lds_bytes, tiles, 16, &blocks); \
            if (rc == CNC_OK) hipLaunchKernelGGL((k_field_fused16w2<FV, NTV, true, 3, 2>), dim3(blocks), dim3(128), lds_bytes, s, p); \
        } else if (rgb) CNC_W2_W(FV, NTV, true); \
        else CNC_W2_W(FV, NTV, false);      \
    } while (0)
#define CNC_W2_NT(FV)                   \
    do {                                \
        if (NT == 5) CNC_W2_RGB(FV, 5); \
        else CNC_W2_RGB(FV, 2);         \
    } while (0)
    if (F == 8) CNC_W2_NT(8);
    else if (F == 4) CNC_W2_NT(4);
    else CNC_W2_NT(2);
#undef CNC_W2_NT
#undef CNC_W2_RGB
#undef CNC_W2_W
#undef CNC_W2
    if (rc != CNC_OK) return rc;
    return launch_status();
}

}  // namespace cnc

using namespace cnc;

extern "C" int cnc_field_pack_all(const cnc_field_pack_t* d, void* stream)
{
    if (!d) return CNC_ERR_INVALID_VALUE;
    PackAllArgs a{};
    uint32_t blocks = 0;
    for (int l = 0; l < 5; l++) {
        const cnc_field_pack_layer_t& L = d->layer[l];
        const bool tr = (L.flags & CNC_PACK_TRANSPOSE) != 0u;
        if (!L.W || L.H == 0 || L.K == 0 || (!tr && L.ldw < L.K)) return CNC_ERR_INVALID_VALUE;
        if (L.Wp) {          // the fp32 fragments (and the biases): the forward's layers
            if (tr || !L.b || !L.Bp || L.n_tiles == 0 || L.n_ksteps == 0 || L.H > L.n_tiles * 32 || L.K > L.n_ksteps * 8)
                return CNC_ERR_INVALID_VALUE;
        } else if (!L.Wq16) {
            return CNC_ERR_INVALID_VALUE;
        }
        if (tr && (L.Wp16 || L.ldw < L.src_off + L.H - ((L.flags & CNC_PACK_ZERO_FIRST) ? 1u : 0u))) return CNC_ERR_INVALID_VALUE;
        if (L.Wp16 && (L.n_ksteps16 == 0 || L.K > L.n_ksteps16 * 16)) return CNC_ERR_INVALID_VALUE;
        if (L.Wq16 && (L.n_colblocks == 0 || L.n_ksteps32 == 0 || L.H > L.n_colblocks * 16 ||
                       L.K + (L.k_gap ? 1u : 0u) > L.n_ksteps32 * 32 || L.k_gap >= L.K))
            return CNC_ERR_INVALID_VALUE;
        a.W[l] = L.W; a.b[l] = L.b; a.H[l] = L.H; a.K[l] = L.K; a.ldw[l] = L.ldw;
        a.nt32[l] = L.n_tiles; a.nk8[l] = L.n_ksteps; a.nk16[l] = L.n_ksteps16;
        a.ncb[l] = L.n_colblocks; a.nk32[l] = L.n_ksteps32; a.k_gap[l] = L.k_gap;
        a.tflags[l] = L.flags; a.src_off[l] = L.src_off;
        a.Wp[l] = L.Wp; a.Bp[l] = L.Bp;
        a.Wp16[l] = reinterpret_cast<half_t*>(L.Wp16);
        a.Wq16[l] = reinterpret_cast<half_t*>(L.Wq16);
        uint32_t total = L.Wp ? L.n_ksteps * L.n_tiles * 256 : 0u;
        if (L.Wp16 && L.n_ksteps16 * L.n_tiles * 512 > total) total = L.n_ksteps16 * L.n_tiles * 512;
        if (L.Wq16 && L.n_ksteps32 * L.n_colblocks * 512 > total) total = L.n_ksteps32 * L.n_colblocks * 512;
        if (l == 1 && d->row0 && d->row0_len > total) total = d->row0_len;
        a.first_block[l] = blocks;
        blocks += div_up(total, 256);
    }
    a.first_block[5] = blocks;
    if (d->row0 && d->row0_len < d->layer[1].K) return CNC_ERR_INVALID_VALUE;
    a.row0 = d->row0; a.row0_len = d->row0_len;
    a.guard = d->guard; a.pack_id = d->pack_id;
    hipLaunchKernelGGL(k_field_pack_all, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return launch_status();
}
