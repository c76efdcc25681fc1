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
#include "common.hpp"

#include "encoder_common.hpp"
#include "field_common.hpp"

namespace cnc {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct FieldEnc {
    const uint8_t* bits;
    const int32_t* offsets;
    const int32_t* res;
    uint32_t       n_levels;
};

struct FusedFieldArgs {
    const float* pos;
    const float* dirs;
    const float* aabb;
    uint32_t     N;
    FieldEnc     enc[4];          // xyz | xy | xz | yz
    const float* freqs;
    uint32_t     n_freqs;
    uint32_t     n_units;         // (encoder, level) units = sum of n_levels
    uint32_t     nkb1;            // K-steps of 8 of layer 1 (a multiple of 4: K padded to whole 32-column chunks)
    const float* Wp[5];           // packed weights (cnc_field_pack_layer)
    const float* Bp[5];           // padded biases
    const float* w2row;           // density only: W2[0, :] padded to NT * 32
    uint32_t     geo;
    uint32_t     nkbh;            // K-steps of the head's first layer: roundup8(16 + geo) / 8
    float*       density;
    float*       rgb;
    uint32_t     sh_fp16;
};

constexpr uint32_t kChunkPitch = 36;     // floats per row of the 32 x 32 chunk tile (+4: conflict-free b128 accesses)
constexpr uint32_t kPadH = 4;

// Weight fragments through a buffer resource: address = SGPR base + one VGPR (16 * lane) + a scalar K-step offset + an
// immediate per tile.  With flat pointers the compiler kept a 64-bit address pair per (layer, K-step, tile) alive across
// the persistent tile loop (hundreds of spilled registers); this way the whole weight stream costs one VGPR.
typedef int32_t i32x4_t __attribute__((ext_vector_type(4)));
using wrsrc_t = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ wrsrc_t weight_rsrc(const float* Wp)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wp), 0, 0x7FFFFFFF, 0x00020000);
}

template <int NT>
__device__ __forceinline__ void load_w(wrsrc_t W, uint32_t kb, uint32_t lane, float4 (&dst)[NT])
{
    const int32_t soff = (int32_t)(kb * NT * 1024u);          // 64 lanes x 16 bytes per (K-step, tile)
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const i32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(W, (int32_t)(lane * 16u + t * 1024), soff, 0);
        dst[t] = make_float4(__builtin_bit_cast(float, v.x), __builtin_bit_cast(float, v.y),
                             __builtin_bit_cast(float, v.z), __builtin_bit_cast(float, v.w));
    }
}

template <int NT>
__device__ __forceinline__ void mfma_step(const float4& a, const float4 (&w)[NT], f32x16 (&acc)[NT])
{
    // k-step outermost: consecutive MFMAs go to different accumulators
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w[t].x, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w[t].y, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w[t].z, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w[t].w, acc[t], 0, 0, 0);
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NT])
{
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int v = 0; v < 16; v++) acc[t][v] = 0;
}

// One wave's LDS writes followed by its own reads: DS operations of a wave execute in order, so only the compiler has
// to be kept from moving them (a workgroup fence would also wait for the weight prefetch in flight: vmcnt(0)).
__device__ __forceinline__ void wave_lds_order()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// acc (32 x NT*32, C layout) = A (32 x nkb*8, LDS row-major, pitch lda) * W^T
template <int NT>
__device__ __forceinline__ void layer_lds(const float* __restrict__ a_lds, uint32_t lda, uint32_t nkb,
                                          const float* __restrict__ Wp_, f32x16 (&acc)[NT], uint32_t lane)
{
    const wrsrc_t Wp = weight_rsrc(Wp_);
    const uint32_t i = lane & 31u, h = lane >> 5;
    zero_acc<NT>(acc);
    float4 wn[NT];
    load_w<NT>(Wp, 0, lane, wn);
    for (uint32_t kb = 0; kb < nkb; kb++) {
        const float4 a = *reinterpret_cast<const float4*>(a_lds + i * lda + kb * 8 + 4 * h);
        float4 w[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) w[t] = wn[t];
        if (kb + 1 < nkb) load_w<NT>(Wp, kb + 1, lane, wn);
        mfma_step<NT>(a, w, acc);
    }
}

// bias (+ ReLU), C layout -> row-major LDS: D[row = 8 (v >> 2) + 4 h + (v & 3)][col = 32 t + i]
template <bool RELU, int NT>
__device__ __forceinline__ void acc_to_lds(float* __restrict__ dst, uint32_t ld, const float* __restrict__ bias,
                                           const f32x16 (&acc)[NT], uint32_t lane)
{
    const uint32_t i = lane & 31u, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const float b = bias[t * 32 + i];
#pragma unroll
        for (int v = 0; v < 16; v++) {
            float x = acc[t][v] + b;
            if (RELU) x = x > 0 ? x : 0;
            dst[(8 * (v >> 2) + 4 * h + (v & 3)) * ld + t * 32 + i] = x;
        }
    }
}

// The F features of one (encoder, level) unit at a point: the body of k_grid_encode_fwd_bits (same corner order, same
// fmaf chain: bit-identical), no occupancy mask.
template <uint32_t D, uint32_t F>
__device__ __forceinline__ void unit_features(const float (&x)[D], bool inside, const FieldEnc& e, uint32_t level,
                                              float (&acc)[F])
{
    constexpr uint32_t C = 1u << D;
#pragma unroll
    for (uint32_t k = 0; k < F; k++) acc[k] = 0;
    if (!inside) return;
    const uint32_t off = (uint32_t)e.offsets[level];
    const uint32_t hs = (uint32_t)e.offsets[level + 1] - off;
    const uint32_t R = (uint32_t)e.res[level];
    Corners<D, false> c;
    c.setup(x, R, hs, 128u, nullptr);
    uint32_t rb[C];
#pragma unroll
    for (uint32_t q = 0; q < C; q++) rb[q] = c.valid[q] ? load_row_bits<F>(e.bits, (uint64_t)off + c.row[q]) : 0u;
#pragma unroll
    for (uint32_t q = 0; q < C; q++) {
        const float tw = c.valid[q] ? c.w[q] * c.wn_re : 0.0f;
#pragma unroll
        for (uint32_t k = 0; k < F; k++) {
            const float s = ((rb[q] >> k) & 1u) ? 1.0f : -1.0f;
            acc[k] = __builtin_fmaf(tw, s, acc[k]);
        }
    }
}

// Columns [w0, w0 + 16) of the feature row of one sample into its row of the chunk tile (`trow`, chunk-relative
// column w0 & 31).  Feature row = [units: n_units x F | x (3) | sin(f_k x) (3), cos(f_k x) (3) for k < n_freqs | 0 ...].
template <uint32_t F>
__device__ __forceinline__ void fill_window(const FusedFieldArgs& p, const float (&xu)[3], uint32_t w0,
                                            float* __restrict__ trow)
{
    constexpr uint32_t V = F < 4 ? F : 4;
    const uint32_t U = p.n_units * F;                 // first sinusoid column
    const bool in_x = xu[0] >= 0.0f && xu[0] <= 1.0f, in_y = xu[1] >= 0.0f && xu[1] <= 1.0f,
               in_z = xu[2] >= 0.0f && xu[2] <= 1.0f;
    const uint32_t L3 = p.enc[0].n_levels, L2 = p.enc[1].n_levels;
#pragma unroll
    for (uint32_t s = 0; s < 16 / F; s++) {
        const uint32_t col = w0 + s * F;
        const uint32_t u = col / F;
        if (u >= p.n_units) break;
        float a[F];
        if (u < L3) {
            unit_features<3, F>(xu, in_x && in_y && in_z, p.enc[0], u, a);
        } else {
            const uint32_t q = u - L3, pl = q / L2, level = q - pl * L2;      // plane 0 = xy, 1 = xz, 2 = yz
            const float    x2[2] = {pl == 2 ? xu[1] : xu[0], pl == 0 ? xu[1] : xu[2]};
            const bool     in2 = (pl == 2 ? in_y : in_x) && (pl == 0 ? in_y : in_z);
            unit_features<2, F>(x2, in2, p.enc[1 + pl], level, a);
        }
        float* o = trow + ((w0 + s * F) & 31u);
#pragma unroll
        for (uint32_t k = 0; k < F; k += V) {
            float v[V];
#pragma unroll
            for (uint32_t j = 0; j < V; j++) v[j] = a[k + j];
            store_vec<V>(o + k, v);
        }
    }
    // the part of the window behind the units: raw coordinates, sinusoids, zero padding
    const uint32_t lo = w0 > U ? w0 : U, hi = w0 + 16;
    if (lo >= hi) return;
    const uint32_t n_sin = 3 + 6 * p.n_freqs;
    for (uint32_t col = lo; col < hi; col++) {
        const uint32_t e = col - U;
        if (e < 3) trow[col & 31u] = e == 0 ? xu[0] : (e == 1 ? xu[1] : xu[2]);
        else if (e >= n_sin) trow[col & 31u] = 0.0f;
    }
    // sin column e = 3 + 6 k + a, its cos column e + 3: ONE argument reduction for both (sincosf returns the values of
    // sinf and cosf); a pair that straddles two windows is evaluated by both lanes
    const uint32_t e_lo = lo - U, e_hi = hi - U;
    for (uint32_t e = e_lo > 6 ? e_lo - 3 : 3; e < e_hi && e < n_sin; e++) {
        const uint32_t k = (e - 3) / 6, r = (e - 3) - 6 * k;
        if (r >= 3) continue;
        const float xa = r == 0 ? xu[0] : (r == 1 ? xu[1] : xu[2]);
        float sn, cs;
        sincosf(xa * p.freqs[k], &sn, &cs);
        if (e >= e_lo) trow[(e + U) & 31u] = sn;
        if (e + 3 >= e_lo && e + 3 < e_hi) trow[(e + 3 + U) & 31u] = cs;
    }
}

template <uint32_t F, int NT, bool RGB>
__global__ __launch_bounds__(64, 2) void k_field_fused(FusedFieldArgs p)
{
    extern __shared__ float lds[];
    const uint32_t lane = threadIdx.x, i = lane & 31u, h = lane >> 5;
    constexpr uint32_t ldh = NT * 32 + kPadH;
    const uint32_t tiles = (p.N + 31u) / 32u;
    float amin[3], aext[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        amin[a] = p.aabb[a];
        aext[a] = p.aabb[3 + a] - p.aabb[a];
    }
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint32_t row0 = tile * 32, row = row0 + i;
        const bool     live = row < p.N;
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
        const uint64_t selmask = __ballot(sel);          // bit r (< 32) = selector of sample r

        // ---- layer 1, chunk by chunk ----
        f32x16 acc[NT];
        zero_acc<NT>(acc);
        float4 wn[NT];
        const wrsrc_t W1 = weight_rsrc(p.Wp[0]);
        load_w<NT>(W1, 0, lane, wn);
        float* trow = lds + i * kChunkPitch;
        for (uint32_t c = 0; c * 4 < p.nkb1; c++) {
            fill_window<F>(p, xu, c * 32 + 16 * h, trow);
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
                if (live) p.density[row] = expf((s + p.Bp[1][0]) - 1.0f) * (sel ? 1.0f : 0.0f);
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
                        if (row0 + r < p.N) p.density[row0 + r] = expf(x - 1.0f) * (float)((selmask >> r) & 1ull);
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
                    if (row0 + r < p.N) p.rgb[(size_t)(row0 + r) * 3 + i] = 1.0f / (1.0f + expf(-(acc5[0][v] + b)));
                }
            }
            wave_lds_order();
        }
    }
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
    uint32_t units = 0;
    for (int e = 0; e < 4; e++) {
        if (!f->bits[e] || !f->offsets[e] || !f->resolutions[e] || f->n_levels[e] == 0) return CNC_ERR_INVALID_VALUE;
        p.enc[e] = FieldEnc{f->bits[e], f->offsets[e], f->resolutions[e], f->n_levels[e]};
        units += f->n_levels[e];
    }
    if (f->n_levels[1] != f->n_levels[2] || f->n_levels[1] != f->n_levels[3]) return CNC_ERR_UNSUPPORTED;
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
    const uint32_t tiles = (N + 31) / 32;
    // one wave per workgroup; registers allow two per SIMD, the colour variant's LDS (32 x (H + 4) floats) seven per CU
    uint32_t blocks = tiles < 256u * 8 ? tiles : 256u * 8;
    uint32_t lds_floats = 32 * kChunkPitch;
    if (want_rgb) lds_floats = 32 * (H + kPadH);
    const size_t lds_bytes = (size_t)lds_floats * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define CNC_FF(FV, NTV)                                                                                            \
    do {                                                                                                           \
        if (want_rgb) hipLaunchKernelGGL((k_field_fused<FV, NTV, true>), dim3(blocks), dim3(64), lds_bytes, s, p); \
        else hipLaunchKernelGGL((k_field_fused<FV, NTV, false>), dim3(blocks), dim3(64), lds_bytes, s, p);         \
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
    return launch_status();
}
