// field_fused.hip — the gradient-free radiance field as ONE kernel: world positions -> density (-> rgb).
//
// Reference chain (examples/radiance_fields/ngp.py:506-547, compose_3D_2D_embed :620-645): normalise to the unit cube,
// four binarised hash-grid encoders (xyz + the xy / xz / yz planes) and the 63-wide sinusoid embedding concatenated
// into a [N, 255] matrix, base MLP 255 -> H (ReLU) -> 1 + geo, density = trunc_exp(x - 1) * selector; for colours
// [SH4(dir) | geo] -> H -> H -> 3, sigmoid.  The product ran that as encoder launches writing a [N, 256] matrix to HBM
// (1 KB per sample each way), library GEMMs and glue kernels.  Every sample of the sampler's visibility pass, of the
// occupancy refresh and of the evaluation render takes this path without gradients: 6-8x the samples of the
// gradient pass.
//
// Here one 64-lane wave owns 32 samples end to end; nothing but positions (and directions) is read and nothing but
// densities (and colours) is written:
//   * layer 1 runs K-chunk by K-chunk.  A chunk is 32 consecutive columns of the feature row; lane (i, h) computes the
//     16 columns [16 h, 16 h + 16) of sample i — whole (encoder, level) units through the same Corners / sign-bit-plane
//     / fmaf chain as k_grid_encode_fwd_bits (bit-identical features), or sinusoid columns — into a 32 x 32 LDS tile
//     that feeds v_mfma_f32_32x32x2_f32 as the A operand; the 32 x H accumulators stay in registers for all chunks.
//     fp32 MFMA = an exact k-ordered fmaf chain, 64 cycles per instruction and SIMD: with 8 K-steps x NT tiles = 80
//     MFMAs per chunk (H = 160) the matrix pipe is the floor (0.55 ms per 2^20 samples at K = 256) and the gather is
//     vector work that a second wave on the same SIMD overlaps with it — hence one-wave workgroups, no block barriers,
//     <= 256 registers.
//   * weights come from a buffer packed in fragment order (cnc_field_pack_layer): one wave-instruction reads 1 KB
//     contiguous, prefetched one K-step ahead, also across the gather of the next chunk.
//   * density only: the second layer's unit 0 is a dot product over the ReLU'd accumulators (vector ALU + an LDS
//     transpose), no further MFMA.  With colours: activations go through LDS (C layout -> row-major, bias + ReLU) between
//     layers; one LDS region per wave is reused for the chunk tile, h1, the head input and the head's hidden layers
//     (a wave's LDS operations execute in order, and every read of a layer is issued before its results exist).
#include "field_fused_common.hpp"

namespace cnc {


template <uint32_t F, int NT, bool RGB>
__global__ __launch_bounds__(64, 2) void k_field_fused(FusedFieldArgs p)
{
    extern __shared__ float lds[];
    // launched behind an fp16 kernel as its fallback: runs only if that kernel (or the weight packer) raised the flag
    if (p.only_if_flagged && *reinterpret_cast<volatile const uint32_t*>(p.guard) != p.call_id) return;
    const uint32_t lane = threadIdx.x, i = lane & 31u, h = lane >> 5;
    constexpr uint32_t ldh = NT * 32 + kPadH;
    const uint32_t n_rows = rows_of(p);       // p.N, or a count the device holds (cnc_fused_field_t.n_rows_dev) — a LOCAL: writing
                                              // to the by-value argument block would move all of it into scratch memory
    const uint32_t tiles = (n_rows + 31u) / 32u;
    float amin[3], aext[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        amin[a] = p.aabb[a];
        aext[a] = p.aabb[3 + a] - p.aabb[a];
    }
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint32_t row0 = tile * 32, row = row0 + i;
        const bool     live = row < n_rows;
        // unit-cube position and selector of sample i (k_field_prepare: same expression)
        float xu[3] = {-1.0f, -1.0f, -1.0f};
        bool  sel = live;
        if (live) {
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const float v = (p.pos[(size_t)row * 3 + a] - amin[a]) / aext[a];
                xu[a] = v;
                sel = sel && v > 0.0f && v < 1.0f;
            }
        }
        [[maybe_unused]] const uint64_t selmask = __ballot(sel);          // bit r (< 32) = selector of sample r

        // ---- layer 1, chunk by chunk ----
        f32x16 acc[NT];
        zero_acc<NT>(acc);
        float4 wn[NT];
        const wrsrc_t W1 = weight_rsrc(p.Wp[0]);
        load_w<NT>(W1, 0, lane, wn);
        float* trow = lds + i * kChunkPitch;
        for (uint32_t c = 0; c * 4 < p.nkb1; c++) {
            fill_window<F, !RGB>(p, xu, c * 32 + 16 * h, RowF32{trow});
            wave_lds_order();
#pragma unroll
            for (uint32_t kb = 0; kb < 4; kb++) {
                const uint32_t g = c * 4 + kb;
                const float4   a = *reinterpret_cast<const float4*>(trow + kb * 8 + 4 * h);
                float4 w[NT];
#pragma unroll
                for (int t = 0; t < NT; t++) w[t] = wn[t];
                if (g + 1 < p.nkb1) load_w<NT>(W1, g + 1, lane, wn);
                mfma_step<NT>(a, w, acc);
            }
            wave_lds_order();
        }

        if constexpr (!RGB) {
            // density_raw = b2[0] + sum_j relu(h1[j]) * W2[0][j]: per lane its 16 samples' partial sums over the
            // columns it holds, transposed through LDS, summed per sample
            float part[16];
#pragma unroll
            for (int v = 0; v < 16; v++) part[v] = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const float b = p.Bp[0][t * 32 + i], w2 = p.w2row[t * 32 + i];
#pragma unroll
                for (int v = 0; v < 16; v++) {
                    float x = acc[t][v] + b;
                    x = x > 0 ? x : 0;
                    part[v] = __builtin_fmaf(x, w2, part[v]);
                }
            }
#pragma unroll
            for (int v = 0; v < 16; v++) lds[(8 * (v >> 2) + 4 * h + (v & 3)) * kChunkPitch + i] = part[v];
            wave_lds_order();
            if (h == 0) {
                float s = 0.0f;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const float4 v4 = *reinterpret_cast<const float4*>(lds + i * kChunkPitch + 4 * q);
                    s += v4.x; s += v4.y; s += v4.z; s += v4.w;
                }
                if (live) p.density[row] = sel ? expf((s + p.Bp[1][0]) - 1.0f) : 0.0f;
            }
            wave_lds_order();
        } else {
            // ---- h1 -> LDS; layer 2 (H -> 1 + geo) ----
            acc_to_lds<true, NT>(lds, ldh, p.Bp[0], acc, lane);
            wave_lds_order();
            constexpr int NT2 = NT == 5 ? 3 : 2;          // 1 + geo <= 96 (H = 160) / 64 (H = 64)
            f32x16 acc2[NT2];
            layer_lds<NT2>(lds, ldh, NT * 4, p.Wp[1], acc2, lane);
            // outputs: column 0 = density_raw, columns 1..geo = geo features -> head input columns 16 + (c - 1);
            // head input = [SH4(dir) (16) | geo | zero padding], K = 8 nkbh <= H columns, in the region (and with the
            // pitch: every LDS offset stays an immediate) h1 occupied — every read of layer 2 has been issued
            const uint32_t Kh = p.nkbh * 8;
            constexpr uint32_t ldi = ldh;
            wave_lds_order();
#pragma unroll
            for (int t = 0; t < NT2; t++) {
                const uint32_t col = t * 32 + i;
                const float    b = p.Bp[1][col];
#pragma unroll
                for (int v = 0; v < 16; v++) {
                    const uint32_t r = 8 * (v >> 2) + 4 * h + (v & 3);
                    const float    x = acc2[t][v] + b;
                    if (col == 0) {
                        if (row0 + r < n_rows) p.density[row0 + r] = ((selmask >> r) & 1ull) ? expf(x - 1.0f) : 0.0f;
                    } else if (15 + col < Kh) {
                        lds[r * ldi + 15 + col] = col <= p.geo ? x : 0.0f;
                    }
                }
            }
            {   // SH4 of sample i's direction: lane (i, h) writes harmonics 8 h .. 8 h + 7
                float d3[3] = {0.0f, 0.0f, 1.0f};
                if (live) {
#pragma unroll
                    for (int a = 0; a < 3; a++) d3[a] = ((p.dirs[(size_t)row * 3 + a] + 1.0f) / 2.0f) * 2.0f - 1.0f;
                }
#pragma unroll
                for (uint32_t q = 0; q < 2; q++) {
                    float4 v = sh4_quad(2 * h + q, d3[0], d3[1], d3[2]);
                    if (p.sh_fp16) {
                        v.x = round_through_half(v.x); v.y = round_through_half(v.y);
                        v.z = round_through_half(v.z); v.w = round_through_half(v.w);
                    }
                    *reinterpret_cast<float4*>(lds + i * ldi + 8 * h + 4 * q) = v;
                }
            }
            wave_lds_order();
            // ---- head: (16 + geo) -> H -> H -> 3 ----
            layer_lds<NT>(lds, ldi, p.nkbh, p.Wp[2], acc, lane);
            wave_lds_order();
            acc_to_lds<true, NT>(lds, ldh, p.Bp[2], acc, lane);
            wave_lds_order();
            layer_lds<NT>(lds, ldh, NT * 4, p.Wp[3], acc, lane);
            wave_lds_order();
            acc_to_lds<true, NT>(lds, ldh, p.Bp[3], acc, lane);
            wave_lds_order();
            f32x16 acc5[1];
            layer_lds<1>(lds, ldh, NT * 4, p.Wp[4], acc5, lane);
            if (i < 3) {
                const float b = p.Bp[4][i];
#pragma unroll
                for (int v = 0; v < 16; v++) {
                    const uint32_t r = 8 * (v >> 2) + 4 * h + (v & 3);
                    if (row0 + r < n_rows) p.rgb[(size_t)(row0 + r) * 3 + i] = 1.0f / (1.0f + expf(-(acc5[0][v] + b)));
                }
            }
            wave_lds_order();
        }
    }
}

// -----------------------------------------------------------------------------------------------------------------
// The same network on the fp16 matrix pipe, three products per term.
//
// v_mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate and — measured (docs/engineering_log.md, round 4): the gather
// alone 0.77 ms, the MFMAs alone 0.91 ms, together 1.34 ms per 2^20 samples — does not overlap with the other wave's
// vector work: 43 % MFMA busy is what that kernel can do.  v_mfma_f32_32x32x16_f16 is 16x the rate on the real matrix
// pipe.  Every operand is split x = hi + lo, hi = half(x), lo = half(x - hi) (22 of fp32's 24 significand bits; the
// weights are scaled by 2^8 first so that their lo parts stay normal numbers) and a product becomes
//     x w ~= hi_x hi_w + hi_x lo_w + lo_x hi_w          (the dropped lo_x lo_w is 2^-22 relative)
// accumulated in fp32 by the MFMA: 3 instructions of 32 cycles per 16 k instead of 8 of 64 per 16 k, and a relative error
// per term of ~5e-7 (fp32 rounding: 6e-8) — two orders below north_star's 1e-4, checked against the chain in
// tests/test_gpu_field_fused.py at 1e-5.  Activations are assumed below fp16's 65504.
// -----------------------------------------------------------------------------------------------------------------
constexpr float kWeightScale = 256.0f, kWeightScaleInv = 1.0f / 256.0f;
constexpr uint32_t kChunkPitch16 = 40;    // halves per row of a 32 x 32 chunk plane (80 bytes: 16-byte aligned rows)
constexpr uint32_t kPadH16x = 8;

// Activation planes of the colour variant: 32 rows x H halves.  H = 160: no padding (two planes = 20 KB: eight waves per
// CU instead of seven) and a swizzle of the 16-byte chunks instead — chunk' = chunk ^ ((row >> 2) & 3): rows are 320
// bytes apart, i.e. rows r and r + 4 start on the same banks; the swizzle moves them to the four different 16-byte
// slots of a 64-byte group, so the 16 rows a ds_read_b128 lane group touches cover all 64 banks.  H = 64: padded rows.
template <int NT>
struct HPlane {
    static constexpr uint32_t ld = NT == 5 ? 160u : NT * 32u + kPadH16x;
    static __device__ __forceinline__ uint32_t at(uint32_t r, uint32_t c)
    {
        if constexpr (NT == 5) return r * ld + ((((c >> 3) ^ ((r >> 2) & 3u)) << 3) | (c & 7u));
        else return r * ld + c;
    }
};

template <int NT>
__device__ __forceinline__ void load_w16(wrsrc_t W, uint32_t ks, uint32_t lane, half8_t (&hi)[NT], half8_t (&lo)[NT])
{
    const int32_t soff = (int32_t)(ks * NT * 2048u);          // per (K-step, tile): 64 x 16 bytes hi, 64 x 16 bytes lo
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const f32x4_t a = llvm_raw_buffer_load_f32x4(W, (int32_t)(lane * 16u + t * 2048), soff, 0);
        const f32x4_t b = llvm_raw_buffer_load_f32x4(W, (int32_t)(lane * 16u + t * 2048 + 1024), soff, 0);
        hi[t] = __builtin_bit_cast(half8_t, a);
        lo[t] = __builtin_bit_cast(half8_t, b);
    }
}

template <int NT>
__device__ __forceinline__ void mfma3(const half8_t& a_hi, const half8_t& a_lo, const half8_t (&w_hi)[NT],
                                      const half8_t (&w_lo)[NT], f32x16 (&acc)[NT])
{
    // the two small products first, consecutive MFMAs on different accumulators
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, w_hi[t], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, w_lo[t], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, w_hi[t], acc[t], 0, 0, 0);
}

// acc = A * W^T with A in two half planes (pitch `ld` halves), K = 16 nks.  Weight fragments of the next K-step are
// requested before the current one's MFMAs (two register sets, the loop unrolled by two): without that every K-step
// waited for an L2 round trip — 36 of them per tile in the colour variant, 1.6 of its 2.5 ms.
template <int NT, int NTH>
__device__ __forceinline__ void layer_lds16(const half_t* __restrict__ a_hi, const half_t* __restrict__ a_lo,
                                            uint32_t nks, const half_t_* __restrict__ Wp_, f32x16 (&acc)[NT], uint32_t lane)
{
    using P = HPlane<NTH>;
    const uint32_t i = lane & 31u, g = lane >> 5;
    const wrsrc_t  Wp = weight_rsrc(reinterpret_cast<const float*>(Wp_));
    zero_acc<NT>(acc);
    half8_t wh0[NT], wl0[NT], wh1[NT], wl1[NT];
    load_w16<NT>(Wp, 0, lane, wh0, wl0);
    for (uint32_t ks = 0; ks < nks; ks += 2) {
        const bool second = ks + 1 < nks;
        if (second) load_w16<NT>(Wp, ks + 1, lane, wh1, wl1);
        {
            const uint32_t at = P::at(i, ks * 16 + 8 * g);
            const half8_t  ah = *reinterpret_cast<const half8_t*>(a_hi + at);
            const half8_t  al = *reinterpret_cast<const half8_t*>(a_lo + at);
            mfma3<NT>(ah, al, wh0, wl0, acc);
        }
        if (second) {
            if (ks + 2 < nks) load_w16<NT>(Wp, ks + 2, lane, wh0, wl0);
            const uint32_t at = P::at(i, ks * 16 + 16 + 8 * g);
            const half8_t  ah = *reinterpret_cast<const half8_t*>(a_hi + at);
            const half8_t  al = *reinterpret_cast<const half8_t*>(a_lo + at);
            mfma3<NT>(ah, al, wh1, wl1, acc);
        }
    }
}

// x = acc / 2^8 + bias (+ ReLU) -> the two half planes, C layout -> row-major
// C layout -> plane index of accumulator element v of lane (i, h), output column `col_lane` + 32 t (col_lane = i, or
// i + 15 for the head input): row r = 8 (v >> 2) + 4 h + (v & 3), so the row's swizzle (r >> 2) & 3 = (2 (v >> 2) + h) & 3
// = k | h with k = 2 ((v >> 2) & 1) known at compile time — two lane bases (k = 0, 2) and immediates, instead of an
// address computation per element (which cost the colour variant 100 spilled registers).
template <int NT>
struct CLayoutAt {
    uint32_t base[2];                    // k = 0, k = 2
    __device__ __forceinline__ CLayoutAt(uint32_t h, uint32_t col_lane)
    {
        using P = HPlane<NT>;
        const uint32_t chunk = col_lane >> 3, within = col_lane & 7u;
        if constexpr (NT == 5) {
            base[0] = 4 * h * P::ld + (((chunk & 4u) | ((chunk & 3u) ^ h)) << 3) + within;
            base[1] = 4 * h * P::ld + (((chunk & 4u) | ((chunk & 3u) ^ (2u | h))) << 3) + within;
        } else {
            base[0] = base[1] = 4 * h * P::ld + col_lane;
        }
    }
    __device__ __forceinline__ uint32_t operator()(int t, int v) const
    {
        return base[(v >> 2) & 1] + (uint32_t)(8 * (v >> 2) + (v & 3)) * HPlane<NT>::ld + (uint32_t)t * 32u;
    }
};

template <bool RELU, int NT>
__device__ __forceinline__ void acc_to_lds16(half_t* __restrict__ d_hi, half_t* __restrict__ d_lo,
                                             const float* __restrict__ bias, const f32x16 (&acc)[NT], uint32_t lane,
                                             float& mx)
{
    const uint32_t i = lane & 31u, h = lane >> 5;
    const CLayoutAt<NT> at(h, i);
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const float b = bias[t * 32 + i];
#pragma unroll
        for (int v = 0; v < 16; v++) {
            float x = __builtin_fmaf(acc[t][v], kWeightScaleInv, b);
            if (RELU) x = x > 0 ? x : 0;
            mx = fmaxf(mx, fabsf(x));
            half_t xh, xl;
            split_half(x, xh, xl);
            d_hi[at(t, v)] = xh;
            d_lo[at(t, v)] = xl;
        }
    }
}

template <uint32_t F, int NT, bool RGB>
__global__ __launch_bounds__(64, 2) void k_field_fused16(FusedFieldArgs p)
{
    extern __shared__ float lds[];
    half_t* lds16 = reinterpret_cast<half_t*>(lds);
    const uint32_t lane = threadIdx.x, i = lane & 31u, h = lane >> 5;
    using HP = HPlane<NT>;
    constexpr uint32_t ldh = HP::ld;                            // halves
    half_t* const c_hi = lds16;                                 // chunk planes
    half_t* const c_lo = lds16 + 32 * kChunkPitch16;
    half_t* const h_hi = lds16;                                 // activation planes (colour variant)
    half_t* const h_lo = lds16 + 32 * ldh;
    const uint32_t n_rows = rows_of(p);       // p.N, or a count the device holds (cnc_fused_field_t.n_rows_dev) — a LOCAL: writing
                                              // to the by-value argument block would move all of it into scratch memory
    const uint32_t tiles = (n_rows + 31u) / 32u;
    float amin[3], aext[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        amin[a] = p.aabb[a];
        aext[a] = p.aabb[3 + a] - p.aabb[a];
    }
    const wrsrc_t W1 = weight_rsrc(reinterpret_cast<const float*>(p.Wp16[0]));
    if (guard_weights_flagged(p, RGB)) return;
    float mx = 0.0f;                     // largest |value| split into halves by this lane (the range guard)
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint32_t row0 = tile * 32, row = row0 + i;
        const bool     live = row < n_rows;
        float xu[3] = {-1.0f, -1.0f, -1.0f};
        bool  sel = live;
        if (live) {
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const float v = (p.pos[(size_t)row * 3 + a] - amin[a]) / aext[a];
                xu[a] = v;
                sel = sel && v > 0.0f && v < 1.0f;
            }
        }

        // ---- layer 1: a chunk = 32 columns = two K-steps of 16 ----
        f32x16 acc[NT];
        zero_acc<NT>(acc);
        half8_t wh0[NT], wl0[NT], wh1[NT], wl1[NT];
        // density only: the first K-step's fragments of a chunk are in flight across its gather (40 registers; the
        // colour variant, at the register limit, requests them after the gather)
        constexpr bool kAcrossFill = !RGB;
        if constexpr (kAcrossFill) load_w16<NT>(W1, 0, lane, wh0, wl0);
        const RowF16 trow{c_hi + i * kChunkPitch16, c_lo + i * kChunkPitch16};
        const uint32_t n_chunks = p.nk16_1 / 2;
        for (uint32_t c = 0; c < n_chunks; c++) {
            fill_window<F, !RGB>(p, xu, c * 32 + 16 * h, trow);
            wave_lds_order();
            if constexpr (!kAcrossFill) load_w16<NT>(W1, 2 * c, lane, wh0, wl0);
            load_w16<NT>(W1, 2 * c + 1, lane, wh1, wl1);
            {
                const half8_t ah = *reinterpret_cast<const half8_t*>(trow.hi + 8 * h);
                const half8_t al = *reinterpret_cast<const half8_t*>(trow.lo + 8 * h);
                mfma3<NT>(ah, al, wh0, wl0, acc);
            }
            if constexpr (kAcrossFill) {
                if (c + 1 < n_chunks) load_w16<NT>(W1, 2 * c + 2, lane, wh0, wl0);
            }
            {
                const half8_t ah = *reinterpret_cast<const half8_t*>(trow.hi + 16 + 8 * h);
                const half8_t al = *reinterpret_cast<const half8_t*>(trow.lo + 16 + 8 * h);
                mfma3<NT>(ah, al, wh1, wl1, acc);
            }
            wave_lds_order();
        }

        if constexpr (!RGB) {
            float part[16];
#pragma unroll
            for (int v = 0; v < 16; v++) part[v] = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const float b = p.Bp[0][t * 32 + i], w2 = p.w2row[t * 32 + i];
#pragma unroll
                for (int v = 0; v < 16; v++) {
                    float x = __builtin_fmaf(acc[t][v], kWeightScaleInv, b);
                    x = x > 0 ? x : 0;
                    part[v] = __builtin_fmaf(x, w2, part[v]);
                }
            }
#pragma unroll
            for (int v = 0; v < 16; v++) lds[(8 * (v >> 2) + 4 * h + (v & 3)) * kChunkPitch + i] = part[v];
            wave_lds_order();
            if (h == 0) {
                float s = 0.0f;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const float4 v4 = *reinterpret_cast<const float4*>(lds + i * kChunkPitch + 4 * q);
                    s += v4.x; s += v4.y; s += v4.z; s += v4.w;
                }
                if (live) p.density[row] = sel ? expf((s + p.Bp[1][0]) - 1.0f) : 0.0f;
            }
            wave_lds_order();
        } else {
            acc_to_lds16<true, NT>(h_hi, h_lo, p.Bp[0], acc, lane, mx);
            wave_lds_order();
            constexpr int NT2 = NT == 5 ? 3 : 2;
            f32x16 acc2[NT2];
            layer_lds16<NT2, NT>(h_hi, h_lo, NT * 2, p.Wp16[1], acc2, lane);
            const uint32_t Kh = p.nk16_h * 16;
            wave_lds_order();
            const CLayoutAt<NT> hin_at(h, i + 15);             // output column c -> head-input column 15 + c
            // density_raw (output column 0, held by the two lanes with i = 0 for 16 samples each) goes through a float
            // slot at the end of each sample's row of the hi plane — the head input only uses the first Kh <= 96 (64)
            // columns and its swizzle stays inside them — so that ONE lane per sample evaluates the exponential and
            // the store is coalesced (16 inlined expf under a divergent branch cost the kernel 40 spilled registers)
            constexpr uint32_t kDensAt = NT == 5 ? 152u : 64u;
#pragma unroll
            for (int t = 0; t < NT2; t++) {
                const uint32_t col = t * 32 + i;
                const float    b = p.Bp[1][col];
#pragma unroll
                for (int v = 0; v < 16; v++) {
                    const uint32_t r = 8 * (v >> 2) + 4 * h + (v & 3);
                    const float    x = __builtin_fmaf(acc2[t][v], kWeightScaleInv, b);
                    if (col == 0) {
                        *reinterpret_cast<float*>(h_hi + r * ldh + kDensAt) = x;
                    } else if (15 + col < Kh) {
                        half_t xh, xl;
                        mx = fmaxf(mx, col <= p.geo ? fabsf(x) : 0.0f);
                        split_half(col <= p.geo ? x : 0.0f, xh, xl);
                        h_hi[hin_at(t, v)] = xh;
                        h_lo[hin_at(t, v)] = xl;
                    }
                }
            }
            {
                float d3[3] = {0.0f, 0.0f, 1.0f};
                if (live) {
#pragma unroll
                    for (int a = 0; a < 3; a++) d3[a] = ((p.dirs[(size_t)row * 3 + a] + 1.0f) / 2.0f) * 2.0f - 1.0f;
                }
                const RowF16 hrow{h_hi, h_lo};
#pragma unroll
                for (uint32_t q = 0; q < 2; q++) {
                    const float4 v = sh4_quad(2 * h + q, d3[0], d3[1], d3[2]);
                    float v4[4] = {v.x, v.y, v.z, v.w};
                    if (p.sh_fp16) {
#pragma unroll
                        for (int j = 0; j < 4; j++) v4[j] = round_through_half(v4[j]);
                    }
                    hrow.put<4>(HP::at(i, 8 * h + 4 * q), v4);       // 4 halves inside one 16-byte chunk
                }
            }
            wave_lds_order();
            if (h == 0 && live) {
                const float x = *reinterpret_cast<const float*>(h_hi + i * ldh + kDensAt);
                p.density[row] = sel ? expf(x - 1.0f) : 0.0f;
            }
            layer_lds16<NT, NT>(h_hi, h_lo, p.nk16_h, p.Wp16[2], acc, lane);
            wave_lds_order();
            acc_to_lds16<true, NT>(h_hi, h_lo, p.Bp[2], acc, lane, mx);
            wave_lds_order();
            layer_lds16<NT, NT>(h_hi, h_lo, NT * 2, p.Wp16[3], acc, lane);
            wave_lds_order();
            acc_to_lds16<true, NT>(h_hi, h_lo, p.Bp[3], acc, lane, mx);
            wave_lds_order();
            f32x16 acc5[1];
            layer_lds16<1, NT>(h_hi, h_lo, NT * 2, p.Wp16[4], acc5, lane);
            if (i < 3) {
                const float b = p.Bp[4][i];
#pragma unroll
                for (int v = 0; v < 16; v++) {
                    const uint32_t r = 8 * (v >> 2) + 4 * h + (v & 3);
                    const float    x = __builtin_fmaf(acc5[0][v], kWeightScaleInv, b);
                    if (row0 + r < n_rows) p.rgb[(size_t)(row0 + r) * 3 + i] = 1.0f / (1.0f + expf(-x));
                }
            }
            wave_lds_order();
        }
    }
    guard_raise(p, mx);
}

// W [H, K] -> fp16 fragments of `layer_lds16`: per (K-step of 16, tile): 64 lanes x 8 halves hi, then the same of lo,
// of 2^8 * W[32 t + (lane & 31)][16 ks + 8 (lane >> 5) + 0..7] (zero outside [H, K])
__global__ __launch_bounds__(256) void k_field_pack_layer16(const float* __restrict__ W, uint32_t H, uint32_t K,
                                                            uint32_t ldw, uint32_t NT, uint32_t nks,
                                                            half_t* __restrict__ Wp)
{
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nks * NT * 512) return;
    const uint32_t e = idx & 7u, lane = (idx >> 3) & 63u, q = idx >> 9;
    const uint32_t t = q % NT, ks = q / NT;
    const uint32_t out = t * 32 + (lane & 31u), k = ks * 16 + 8 * (lane >> 5) + e;
    const float    w = (out < H && k < K) ? W[(size_t)out * ldw + k] * kWeightScale : 0.0f;
    half_t hi, lo;
    split_half(w, hi, lo);
    Wp[((size_t)q * 2 + 0) * 512 + lane * 8 + e] = hi;
    Wp[((size_t)q * 2 + 1) * 512 + lane * 8 + e] = lo;
}

// W [H, K] (row stride ldw) -> fragment order for `layer_lds` / layer 1: float4 index (kb * NT + t) * 64 + lane holds
// W[32 t + (lane & 31)][8 kb + 4 (lane >> 5) + 0..3], zero outside [H, K]; bias padded to NT * 32; row0 (optional):
// W[0, :] padded to `row0_len` floats.
__global__ __launch_bounds__(256) void k_field_pack_layer(const float* __restrict__ W, const float* __restrict__ b,
                                                          uint32_t H, uint32_t K, uint32_t ldw, uint32_t NT,
                                                          uint32_t nkb, float* __restrict__ Wp, float* __restrict__ Bp,
                                                          float* __restrict__ row0, uint32_t row0_len)
{
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    const uint32_t total = nkb * NT * 256;
    if (idx < total) {
        const uint32_t m = idx & 3u, lane = (idx >> 2) & 63u, q = idx >> 8;
        const uint32_t t = q % NT, kb = q / NT;
        const uint32_t out = t * 32 + (lane & 31u), k = kb * 8 + 4 * (lane >> 5) + m;
        Wp[idx] = (out < H && k < K) ? W[(size_t)out * ldw + k] : 0.0f;
    }
    if (idx < NT * 32 && Bp) Bp[idx] = idx < H ? b[idx] : 0.0f;
    if (row0 && idx < row0_len) row0[idx] = idx < K ? W[idx] : 0.0f;
}

}  // namespace cnc

using namespace cnc;

extern "C" int cnc_field_pack_layer(const float* W, const float* b, uint32_t H, uint32_t K, uint32_t ldw,
                                    uint32_t n_tiles, uint32_t n_ksteps, float* Wp, float* Bp, float* row0,
                                    uint32_t row0_len, void* stream)
{
    if (!W || !b || !Wp || !Bp || H == 0 || K == 0 || n_tiles == 0 || n_ksteps == 0 || ldw < K) return CNC_ERR_INVALID_VALUE;
    if (H > n_tiles * 32 || K > n_ksteps * 8 || (row0 && row0_len < K)) return CNC_ERR_INVALID_VALUE;
    uint32_t total = n_ksteps * n_tiles * 256;
    if (row0 && row0_len > total) total = row0_len;
    hipLaunchKernelGGL(k_field_pack_layer, dim3(div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, W, b, H, K, ldw,
                       n_tiles, n_ksteps, Wp, Bp, row0, row0_len);
    return launch_status();
}

extern "C" int cnc_field_pack_layer16(const float* W, uint32_t H, uint32_t K, uint32_t ldw, uint32_t n_tiles,
                                      uint32_t n_ksteps16, void* Wp16, void* stream)
{
    if (!W || !Wp16 || H == 0 || K == 0 || n_tiles == 0 || n_ksteps16 == 0 || ldw < K) return CNC_ERR_INVALID_VALUE;
    if (H > n_tiles * 32 || K > n_ksteps16 * 16) return CNC_ERR_INVALID_VALUE;
    const uint32_t total = n_ksteps16 * n_tiles * 512;
    hipLaunchKernelGGL(k_field_pack_layer16, dim3(div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, W, H, K, ldw,
                       n_tiles, n_ksteps16, reinterpret_cast<half_t*>(Wp16));
    return launch_status();
}

extern "C" int cnc_field_fused_forward(const cnc_fused_field_t* f, const float* positions, const float* dirs, uint32_t N,
                                       float* density, float* rgb, void* stream)
{
    if (N == 0) return CNC_OK;
    if (!f || !positions || !density || !f->aabb) return CNC_ERR_INVALID_VALUE;
    const bool want_rgb = rgb != nullptr;
    if (want_rgb && !dirs) return CNC_ERR_INVALID_VALUE;
    const uint32_t F = f->n_features, H = f->n_neurons;
    if (!(F == 2 || F == 4 || F == 8) || !(H == 64 || H == 160)) return CNC_ERR_UNSUPPORTED;
    FusedFieldArgs p{};
    p.pos = positions; p.dirs = dirs; p.aabb = f->aabb; p.N = N;
    p.n_dev = f->n_rows_dev;
    uint32_t units = 0;
    for (int e = 0; e < 4; e++) {
        if (!f->bits[e] || !f->offsets[e] || !f->resolutions[e] || f->n_levels[e] == 0) return CNC_ERR_INVALID_VALUE;
        p.enc[e] = FieldEnc{f->bits[e], f->offsets[e], f->resolutions[e], f->n_levels[e]};
        units += f->n_levels[e];
    }
    if (f->n_levels[1] != f->n_levels[2] || f->n_levels[1] != f->n_levels[3]) return CNC_ERR_UNSUPPORTED;
    if (!f->units) return CNC_ERR_INVALID_VALUE;
    p.units = reinterpret_cast<const uint4*>(f->units);
    if (f->n_freqs == 0) return CNC_ERR_UNSUPPORTED;
    if (!f->freqs) return CNC_ERR_INVALID_VALUE;
    p.freqs = f->freqs; p.n_freqs = f->n_freqs; p.n_units = units;
    const uint32_t K0 = units * F + 3 + 6 * f->n_freqs;
    p.nkb1 = (K0 + 31) / 32 * 4;
    p.geo = f->geo_feat_dim;
    const uint32_t NT = H / 32, NT2 = NT == 5 ? 3u : 2u;
    p.nkbh = (16 + p.geo + 7) / 8;
    // the second layer's output tiles and the head's input (kept inside the hidden layers' LDS region) bound geo
    if (1 + p.geo > NT2 * 32 || p.nkbh * 8 > H) return CNC_ERR_UNSUPPORTED;
    for (int l = 0; l < 5; l++) {
        p.Wp[l] = f->packed_weights[l];
        p.Bp[l] = f->packed_biases[l];
    }
    if (!p.Wp[0] || !p.Bp[0] || !p.Bp[1]) return CNC_ERR_INVALID_VALUE;
    p.w2row = f->w2_row0;
    if (want_rgb) {
        for (int l = 1; l < 5; l++)
            if (!p.Wp[l] || !p.Bp[l]) return CNC_ERR_INVALID_VALUE;
    } else if (!p.w2row) {
        return CNC_ERR_INVALID_VALUE;
    }
    p.density = density; p.rgb = rgb;
    p.sh_fp16 = (f->flags & CNC_FIELD_SH_FP16) ? 1u : 0u;
    const bool two_waves = (f->flags & CNC_FIELD_TWO_WAVES) != 0;
    const bool f16x3 = two_waves || (f->flags & CNC_FIELD_MFMA_F16X3) != 0;
    p.nk16_1 = p.nkb1 / 2;
    p.nk16_h = (16 + p.geo + 15) / 16;
    p.nk32_h = (17 + p.geo + 31) / 32;              // two-wave kernels: [SH4 | raw density | geo]
    if (f16x3) {
        if (p.nk16_h * 16 > H || p.nk32_h * 32 > H) return CNC_ERR_UNSUPPORTED;
        // the range guard is part of the fp16 form: without it a value above 65504 would come out as inf / NaN
        if (!f->guard || f->call_id == 0 || f->pack_id == 0) return CNC_ERR_INVALID_VALUE;
        p.guard = f->guard; p.call_id = f->call_id; p.pack_id = f->pack_id;
        if (f->debug_features) {         // test hook: the two-wave density kernel only, rows wide enough for K padded to 32
            if (!two_waves || want_rgb || f->debug_ld < p.nkb1 * 8) return CNC_ERR_UNSUPPORTED;
            p.dbg_features = f->debug_features; p.dbg_ld = f->debug_ld;
        }
        for (int l = 0; l < (want_rgb ? 5 : 1); l++) {
            const void* w16 = two_waves ? f->packed_weights16q[l] : f->packed_weights16[l];
            if (!w16) return CNC_ERR_INVALID_VALUE;
            (two_waves ? p.Wq16[l] : p.Wp16[l]) = reinterpret_cast<const half_t_*>(w16);
        }
        if (two_waves && 1 + p.geo > (NT == 5 ? 80u : 64u)) return CNC_ERR_UNSUPPORTED;
    }
    const bool saving = f->save.feat != nullptr;
    if (saving) {                        // the gradient pass's forward: the two-wave colour kernel, saving variant
        const cnc_field_save_t& sv = f->save;
        if (!two_waves || !want_rgb || f->debug_features) return CNC_ERR_UNSUPPORTED;
        if (!sv.h1 || !sv.h3 || !sv.h4 || !sv.head_in || !sv.raw || !sv.selector || !sv.xyz || !sv.xy || !sv.xz || !sv.yz ||
            sv.ld_feat < p.nkb1 * 8 || sv.ld_feat % 4 != 0 || sv.ld_head != p.nk32_h * 32 || sv.n_live > N)
            return CNC_ERR_INVALID_VALUE;
        p.save = FieldSave{sv.feat, sv.ld_feat, sv.h1, sv.h3, sv.h4, sv.head_in, sv.ld_head, sv.raw, sv.selector,
                           sv.xyz, sv.xy, sv.xz, sv.yz, sv.n_live};
    }
    const uint32_t tiles = (N + 31) / 32;
    uint32_t lds_floats = 32 * kChunkPitch;
    if (want_rgb) lds_floats = 32 * (H + kPadH);
    const size_t lds32 = (size_t)lds_floats * sizeof(float);
    size_t lds16 = 0;
    if (f16x3) {        // two half planes: 32 x 40 (chunk; the density epilogue's 32 x 36 floats fit) or 32 x HPlane::ld
        const uint32_t ldh16 = NT == 5 ? 160u : H + kPadH16x;         // HPlane<NT>::ld
        lds16 = want_rgb ? (size_t)2 * 32 * ldh16 * sizeof(half_t) : (size_t)2 * 32 * kChunkPitch16 * sizeof(half_t);
        if (lds16 < 32 * kChunkPitch * sizeof(float)) lds16 = 32 * kChunkPitch * sizeof(float);
    }
    hipStream_t s = (hipStream_t)stream;
    // The grid is what is RESIDENT at once (the waves loop over the tiles): registers allow 8 one-wave workgroups per
    // CU, the colour variant's LDS 7 — with 8 per CU launched the eighth of every CU ran as a second round on an
    // otherwise idle chip (2.65 instead of 1.72 ms per 2^20 samples).  Residency is a fact of the binary and the
    // device: asked once per (thread, kernel, device), not at every launch.
    int rc = CNC_OK;
#define CNC_FF_GRID(K, LDS)                                                                                   \
    do {                                                                                                      \
        struct Slot { int dev; uint32_t n; };                                                                 \
        static thread_local Slot slot = {-1, 0};                                                              \
        int dev = 0;                                                                                          \
        if (hipGetDevice(&dev) != hipSuccess) { rc = CNC_ERR_LAUNCH; break; }                                 \
        if (slot.dev != dev) {                                                                                \
            int per_cu = 0, cus = 0;                                                                          \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, K, 64, LDS) != hipSuccess ||            \
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||      \
                per_cu <= 0 || cus <= 0) { rc = CNC_ERR_LAUNCH; break; }                                      \
            if (per_cu > 8) per_cu = 8;                                                                       \
            slot = Slot{dev, (uint32_t)(per_cu * cus)};                                                       \
        }                                                                                                     \
        uint32_t blocks = tiles < slot.n ? tiles : slot.n;                                                    \
        /* the guard's conditional launch returns at once in all but pathological calls: a quarter of the     \
           resident grid keeps the empty launch short (its waves loop over the tiles when it does run) */     \
        if (p.only_if_flagged && blocks > slot.n / 4) blocks = slot.n / 4;                                    \
        hipLaunchKernelGGL(K, dim3(blocks), dim3(64), LDS, s, p);                                             \
    } while (0)
#define CNC_FF(FV, NTV)                                                                                 \
    do {                                                                                                \
        if (two_waves) rc = launch_field_fused_w2(p, want_rgb, FV, H, (f->flags & CNC_FIELD_WAVES4) ? 4 : 3, s); \
        else if (f16x3 && want_rgb) CNC_FF_GRID((k_field_fused16<FV, NTV, true>), lds16);               \
        else if (f16x3) CNC_FF_GRID((k_field_fused16<FV, NTV, false>), lds16);                          \
        if (rc != CNC_OK || saving) break;           /* the saving variant saturates: nothing runs behind it */ \
        /* the exact form: the whole call without the fp16 flag, the guard's conditional fallback with it */ \
        p.only_if_flagged = f16x3 ? 1u : 0u;                                                            \
        if (want_rgb) CNC_FF_GRID((k_field_fused<FV, NTV, true>), lds32);                               \
        else CNC_FF_GRID((k_field_fused<FV, NTV, false>), lds32);                                       \
    } while (0)
#define CNC_FF_F(FV)          \
    do {                      \
        if (NT == 5) CNC_FF(FV, 5); \
        else CNC_FF(FV, 2);   \
    } while (0)
    if (F == 8) CNC_FF_F(8);
    else if (F == 4) CNC_FF_F(4);
    else CNC_FF_F(2);
#undef CNC_FF_F
#undef CNC_FF
#undef CNC_FF_GRID
    if (rc != CNC_OK) return rc;
    return launch_status();
}
