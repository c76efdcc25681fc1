// field_wgrad.hip — the five weight gradients of the radiance field's MLPs, dW_l = G_l^T A_l, as one kernel + one reduction.
//
// Reference: autograd's LinearBackward of `mlp_base` / `mlp_head` (ngp.py:506-547): for each Linear, grad_weight =
// grad_output^T @ input, a [out, in] matrix summed over ALL samples of the step (2^18 rows).  The product ran these as five
// library split-K batched GEMMs + five sums over the slabs (+ tails, slices, a cat): ~20 launches, ~0.6 ms of the render
// pass's backward.  Here: cnc_field_backward_chain has left G_1..G_5 in HBM and the forward (the saving fused kernel, or the
// op chain) the inputs A_l; one launch streams both once.
//
// Shape of the work: K = samples.  For v_mfma_f32_16x16x32_f16 a lane's operand is 8 consecutive k of ONE feature — a
// column of a row-major [N, features] matrix — so a workgroup stages a tile of TS = 64 (128) samples through LDS
// TRANSPOSED: thread (sample pair m, float4 column c) loads rows 2m, 2m + 1, splits the 8 values into fp16 hi / lo
// (three-product scheme: x y ~= hi hi + hi lo + lo hi, fp32 accumulation) and writes 4 + 4 dwords {x[2m][f], x[2m+1][f]}
// into feature rows f .. f + 3 of the [feature][sample] planes; fragments are then 16-byte LDS reads.
//   * Layers do not share operands (G_l and A_l are read by layer l only): the grid is split between the layers in
//     proportion to their bytes per sample, a workgroup serves ONE layer ("role") and walks that layer's sample tiles with
//     stride n_wg.  No byte is read twice: 1268 floats per sample at the headline shape = 1.33 GB per 2^18 samples.
//   * A workgroup is 16 waves (4 per SIMD) in a 2 x 8 grid over the (out, in) blocks of 16 x 16: 5 x 2 blocks = 40
//     accumulator registers per wave; the whole dW_l of the workgroup's samples stays in registers until the end.
//   * Memory-bound by design (0.31 ms = 4.3 TB/s; the matrix work is ~50 us at full rate): the next tile's rows are requested into registers
//     before the current tile's products are issued; two barriers per tile.
//   * Gradients are small (1e-3 .. 1e-9): G_l is scaled by ONE power of two per layer — chosen from max |G_l|, which
//     cnc_field_backward_chain leaves in g_max (no host round trip) — so that its largest entry is in [2^13, 2^14); an
//     entry within 2^17 of the largest keeps 22 bits, smaller ones an absolute error of 2^-38 of the largest.  The inputs
//     A_l are split as they are (as the forward does), clamped to fp16's range.
//   * Each workgroup writes its partial dW_l ([out, in] padded to 16 x 16 blocks); k_field_wgrad_reduce sums a layer's
//     partials, undoes the scale and writes the [out, in] gradient — dropping, for head.0, the raw-density slot of the
//     fused kernel's head-input layout (cnc_field_save_t).
#include "field_mma.hpp"

namespace cnc {

struct WGradRole {
    const float* G;          // [N, ldG] gradient w.r.t. the layer's output, columns [0, nO)
    const float* A;          // [N, ldA] the layer's input, columns [0, nI)
    uint32_t     ldG, ldA;
    uint32_t     nO, nI;     // nI % 4 == 0; columns [nO, roundup4(nO)) of G are readable (zero or not: they are dropped)
    uint32_t     kind;       // 0: 64-sample tiles; 1: 128-sample tiles (narrow G: nO <= 64 halves of ... see the host)
    uint32_t     wg0, n_wg;  // the role's workgroups [wg0, wg0 + n_wg)
    uint32_t     part_off;   // floats: this role's partials [n_wg][16 OB][16 IB] in the workspace
    float*       out;        // [nO, ld_out] the gradient
    uint32_t     ld_out, n_out_cols;
    uint32_t     gap_col;    // input column that is NOT a column of the weight (0xFFFFFFFF: none)
};

struct WGradArgs {
    uint32_t        N;
    WGradRole       role[5];
    float*          partial;
    const uint32_t* g_max;   // [5] float bits of max |G_l| (cnc_field_bwd_t.g_max)
};

constexpr int kWgWaves = 16, kWgThreads = 1024;
#ifndef CNC_WGRAD_ROW_LANES
#define CNC_WGRAD_ROW_LANES 4
#endif
constexpr uint32_t kRowLanes = CNC_WGRAD_ROW_LANES;
constexpr int kOBW = 5, kIBW = 2, kWO = 2, kWI = kWgWaves / kWO;      // blocks per wave, wave grid (out x in)

__device__ __forceinline__ float pow2_scale(uint32_t max_bits)
{
    const float m = __builtin_bit_cast(float, max_bits);
    if (!(m > 0.0f) || !(m < 3.0e38f)) return 1.0f;
    const int e = __builtin_amdgcn_frexp_expf(m);              // m = f 2^e, f in [0.5, 1)
    return __builtin_amdgcn_ldexpf(1.0f, 14 - e);
}

// One matrix's share of a tile: thread item (pair m, float4 column c) <- rows row0 + 2m, row0 + 2m + 1.
template <int NIT, uint32_t NP>
__device__ __forceinline__ void stage_request(wrsrc_t M, uint32_t ld, uint32_t c4, uint32_t row0, uint32_t tid, float4 (&v)[NIT][2])
{
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        const uint32_t it = tid + k * kWgThreads;
        // kRowLanes consecutive lanes read kRowLanes x 16 contiguous bytes of one row (then the next row pair).  Measured at 2^18
        // samples (ms): 1 lane per row — the conflict-free order for the LDS writes — 0.37, 2: 0.337, 4: 0.31, 8: 0.324, 16: 0.39
        const uint32_t m = (it / kRowLanes) % NP, c = ((it / kRowLanes) / NP) * kRowLanes + (it % kRowLanes);
        // an item past the matrix's columns reads beyond the records: zeros, never written to LDS
        const uint32_t off = c < c4 ? ((row0 + 2u * m) * ld + 4u * c) * 4u : 0xFFFFFFF0u;
        const f32x4_t  a = llvm_raw_buffer_load_f32x4(M, (int32_t)off, 0, 0);
        const f32x4_t  b = llvm_raw_buffer_load_f32x4(M, (int32_t)(c < c4 ? off + ld * 4u : 0xFFFFFFF0u), 0, 0);
        v[k][0] = make_float4(a.x, a.y, a.z, a.w);
        v[k][1] = make_float4(b.x, b.y, b.z, b.w);
    }
}

// ... x s (clamped to fp16's range), split, into feature rows frow0 + 4c .. + 3 of the planes: dword m of a row =
// {x[2m][f], x[2m + 1][f]}
template <int NIT, uint32_t NP, uint32_t P>
__device__ __forceinline__ void stage_write(const float4 (&v)[NIT][2], uint32_t c4, uint32_t frow0, float s, uint32_t tid,
                                            half_t* __restrict__ hi, half_t* __restrict__ lo)
{
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        const uint32_t it = tid + k * kWgThreads;
        const uint32_t m = (it / kRowLanes) % NP, c = ((it / kRowLanes) / NP) * kRowLanes + (it % kRowLanes);
        if (c >= c4) continue;
        const float x0[4] = {v[k][0].x, v[k][0].y, v[k][0].z, v[k][0].w}, x1[4] = {v[k][1].x, v[k][1].y, v[k][1].z, v[k][1].w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float a = __builtin_amdgcn_fmed3f(x0[j] * s, -kHalfMax, kHalfMax), b = __builtin_amdgcn_fmed3f(x1[j] * s, -kHalfMax, kHalfMax);
            half2_t h, l;
            half_t  t0, t1;
            split_half(a, t0, t1);
            h[0] = t0; l[0] = t1;
            split_half(b, t0, t1);
            h[1] = t0; l[1] = t1;
            const uint32_t at = (frow0 + 4u * c + j) * P + 2u * m;
            *reinterpret_cast<half2_t*>(hi + at) = h;
            *reinterpret_cast<half2_t*>(lo + at) = l;
        }
    }
}

// One role: this workgroup's tiles of TS = 32 KS samples.  NG / NA: items per thread of the G / A share of a tile.
template <int KS, int NG, int NA>
__device__ __forceinline__ void wgrad_role(const WGradArgs& p, const WGradRole& R, uint32_t l, uint32_t j, half_t* __restrict__ lds16)
{
    constexpr uint32_t TS = 32u * KS, NP = TS / 2u, P = TS + 8u;         // samples / pairs per tile, halves per LDS row
    const uint32_t tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u, r = lane & 15u, kq = lane >> 4;
    const uint32_t wo = w / kWI, wi = w % kWI;
    const uint32_t OB = (R.nO + 15u) / 16u, IB = (R.nI + 15u) / 16u, cG = (R.nO + 3u) / 4u, cA = R.nI / 4u;
    const uint32_t rows = 16u * (OB + IB);
    half_t* const  hi = lds16;
    half_t* const  lo = lds16 + rows * P;
    // feature rows past nO / nI and the rows' padding stay zero for the whole kernel
    for (uint32_t i = tid; i < rows * P; i += kWgThreads) reinterpret_cast<uint32_t*>(lds16)[i] = 0u;       // both planes: rows * P dwords
    const wrsrc_t rG = weight_rsrc(R.G, p.N * R.ldG * 4u), rA = weight_rsrc(R.A, p.N * R.ldA * 4u);
    const float   sG = pow2_scale(p.g_max[l]);
    const uint32_t tiles = (p.N + TS - 1u) / TS;
    f32x4 acc[kOBW][kIBW];
    zero_q<kOBW, kIBW>(acc);
    float4 vg[NG][2], va[NA][2];
    uint32_t tile = j;
    if (tile < tiles) {
        stage_request<NG, NP>(rG, R.ldG, cG, tile * TS, tid, vg);
        stage_request<NA, NP>(rA, R.ldA, cA, tile * TS, tid, va);
    }
    __syncthreads();
    for (; tile < tiles; tile += R.n_wg) {
        stage_write<NG, NP, P>(vg, cG, 0, sG, tid, hi, lo);
        stage_write<NA, NP, P>(va, cA, 16u * OB, 1.0f, tid, hi, lo);
        __syncthreads();
        if (tile + R.n_wg < tiles) {                   // the next tile's rows: in flight across the products
            stage_request<NG, NP>(rG, R.ldG, cG, (tile + R.n_wg) * TS, tid, vg);
            stage_request<NA, NP>(rA, R.ldA, cA, (tile + R.n_wg) * TS, tid, va);
        }
        if (wo * kOBW < OB && wi * kIBW < IB) {
#pragma unroll 1
            for (int ks = 0; ks < KS; ks++) {
                half8_t ah[kIBW], al[kIBW];
#pragma unroll
                for (int ib = 0; ib < kIBW; ib++) {
                    const uint32_t blk = wi * kIBW + ib;      // a block past IB reads the rows of padding behind: zeros? no — guarded below
                    const uint32_t row = 16u * OB + 16u * (blk < IB ? blk : wi * kIBW) + r;
                    ah[ib] = *reinterpret_cast<const half8_t*>(hi + row * P + 32u * ks + 8u * kq);
                    al[ib] = *reinterpret_cast<const half8_t*>(lo + row * P + 32u * ks + 8u * kq);
                }
#pragma unroll
                for (int ob = 0; ob < kOBW; ob++) {
                    const uint32_t blk = wo * kOBW + ob;
                    if (blk >= OB) break;
                    const uint32_t row = 16u * blk + r;
                    const half8_t  gh = *reinterpret_cast<const half8_t*>(hi + row * P + 32u * ks + 8u * kq);
                    const half8_t  gl = *reinterpret_cast<const half8_t*>(lo + row * P + 32u * ks + 8u * kq);
                    // D[in 4 kq + v][out lane & 15] += A^T G: the input fragment is the A operand
#pragma unroll
                    for (int ib = 0; ib < kIBW; ib++) acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ib], gh, acc[ob][ib], 0, 0, 0);
#pragma unroll
                    for (int ib = 0; ib < kIBW; ib++) acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ib], gl, acc[ob][ib], 0, 0, 0);
#pragma unroll
                    for (int ib = 0; ib < kIBW; ib++) acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ib], gh, acc[ob][ib], 0, 0, 0);
                }
            }
        }
        __syncthreads();                               // every fragment has been read: the next tile may be written
    }
    // the workgroup's partial: [16 OB][16 IB], lane (r, kq) holds out 16 ob + r, in 16 ib + 4 kq .. + 3
    float* const part = p.partial + R.part_off + (size_t)j * (256u * OB * IB);
#pragma unroll
    for (int ob = 0; ob < kOBW; ob++) {
        const uint32_t oblk = wo * kOBW + ob;
        if (oblk >= OB) break;
#pragma unroll
        for (int ib = 0; ib < kIBW; ib++) {
            const uint32_t iblk = wi * kIBW + ib;
            if (iblk >= IB) break;
            *reinterpret_cast<float4*>(part + (size_t)(16u * oblk + r) * (16u * IB) + 16u * iblk + 4u * kq) =
                make_float4(acc[ob][ib][0], acc[ob][ib][1], acc[ob][ib][2], acc[ob][ib][3]);
        }
    }
}

__global__ __launch_bounds__(kWgThreads) void k_field_wgrad(WGradArgs p)
{
    extern __shared__ float lds[];
    half_t* const lds16 = reinterpret_cast<half_t*>(lds);
    // (static indices only: a dynamically indexed kernel argument is copied to scratch memory)
    uint32_t  l = 0;
    WGradRole R = p.role[0];
#pragma unroll
    for (uint32_t k = 1; k < 5; k++)
        if (blockIdx.x >= p.role[k].wg0) {
            l = k;
            R = p.role[k];
        }
    const uint32_t j = blockIdx.x - R.wg0;
    if (j >= R.n_wg) return;
    if (R.kind == 1) wgrad_role<4, 1, 3>(p, R, l, j, lds16);
    else wgrad_role<2, 2, 2>(p, R, l, j, lds16);
}

// out[o][c] = (sum over the role's workgroups of partial[o][src(c)]) / scale
__global__ __launch_bounds__(256) void k_field_wgrad_reduce(WGradArgs p)
{
    WGradRole R = p.role[0];
#pragma unroll
    for (uint32_t k = 1; k < 5; k++)
        if (blockIdx.y == k) R = p.role[k];
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= R.nO * R.n_out_cols) return;
    const uint32_t o = e / R.n_out_cols, c = e % R.n_out_cols, src = c < R.gap_col ? c : c + 1u;
    const uint32_t OB = (R.nO + 15u) / 16u, IB = (R.nI + 15u) / 16u;
    const float*   q = p.partial + R.part_off + (size_t)o * (16u * IB) + src;
    const size_t   stride = 256u * OB * IB;
    float          s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    uint32_t       g = 0;
    for (; g + 4 <= R.n_wg; g += 4) {
        s0 += q[(size_t)g * stride];
        s1 += q[(size_t)(g + 1) * stride];
        s2 += q[(size_t)(g + 2) * stride];
        s3 += q[(size_t)(g + 3) * stride];
    }
    for (; g < R.n_wg; g++) s0 += q[(size_t)g * stride];
    const float inv = 1.0f / pow2_scale(p.g_max[blockIdx.y]);
    R.out[(size_t)o * R.ld_out + c] = ((s0 + s1) + (s2 + s3)) * inv;
}

// (layer l of the descriptor) -> role shape; returns false when the kernels are not built for it
static bool role_shape(const cnc_field_wgrad_t* f, int l, uint32_t* nO, uint32_t* nI, uint32_t* kind, size_t* lds_bytes)
{
    *nO = f->n_out[l];
    *nI = f->ldA[l];
    if (*nO == 0 || *nI == 0 || *nI % 4 != 0 || f->ldG[l] % 4 != 0 || f->ldG[l] < (*nO + 3) / 4 * 4) return false;
    const uint32_t OB = (*nO + 15) / 16, IB = (*nI + 15) / 16, cG = (*nO + 3) / 4, cA = *nI / 4;
    if (OB > (uint32_t)(kOBW * kWO) || IB > (uint32_t)(kIBW * kWI)) return false;
    const uint32_t rows = 16 * (OB + IB);
    // 128-sample tiles where the tile of 64 would be small (a narrow layer) and the planes fit
    if (cG <= 16 && cA <= 48 && (size_t)rows * 136 * 4 <= 100 * 1024) {
        *kind = 1;
        *lds_bytes = (size_t)rows * 136 * 4;
    } else {
        if (cG > 64 || cA > 64) return false;
        *kind = 0;
        *lds_bytes = (size_t)rows * 72 * 4;
    }
    return true;
}

static int wgrad_plan(const cnc_field_wgrad_t* f, uint32_t n_wg_total, WGradArgs* p, size_t* lds_max, uint64_t* ws_bytes)
{
    double   bytes[5], total = 0;
    uint32_t nO[5], nI[5], kind[5];
    *lds_max = 0;
    for (int l = 0; l < 5; l++) {
        size_t lds = 0;
        if (!role_shape(f, l, &nO[l], &nI[l], &kind[l], &lds)) return CNC_ERR_UNSUPPORTED;
        if (lds > *lds_max) *lds_max = lds;
        bytes[l] = (double)((nO[l] + 3) / 4 * 4 + nI[l]);
        total += bytes[l];
    }
    if (n_wg_total < 5) n_wg_total = 5;
    uint32_t wg0 = 0;
    uint64_t off = 0;
    for (int l = 0; l < 5; l++) {
        uint32_t n = (uint32_t)(n_wg_total * bytes[l] / total + 0.5);
        if (n < 1) n = 1;
        WGradRole& R = p->role[l];
        R.G = f->G[l]; R.A = f->A[l]; R.ldG = f->ldG[l]; R.ldA = f->ldA[l]; R.nO = nO[l]; R.nI = nI[l]; R.kind = kind[l];
        R.wg0 = wg0; R.n_wg = n; R.part_off = (uint32_t)off;
        R.out = f->dW[l]; R.ld_out = f->ld_dW[l]; R.n_out_cols = f->n_in[l];
        R.gap_col = l == 2 ? f->head_gap_col : 0xFFFFFFFFu;
        wg0 += n;
        off += (uint64_t)n * 256u * ((nO[l] + 15) / 16) * ((nI[l] + 15) / 16);
    }
    if (off >= (1ull << 32)) return CNC_ERR_UNSUPPORTED;
    *ws_bytes = off * sizeof(float);
    return CNC_OK;
}

static int wgrad_workgroups(uint32_t asked, uint32_t* n)
{
    if (asked) { *n = asked; return CNC_OK; }
    static thread_local int cached_dev = -1;
    static thread_local uint32_t cached_cus = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return CNC_ERR_LAUNCH;
    if (cached_dev != dev) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return CNC_ERR_LAUNCH;
        cached_cus = (uint32_t)cus;
        cached_dev = dev;
    }
    *n = cached_cus;                  // one 16-wave workgroup per CU
    return CNC_OK;
}

}  // namespace cnc

using namespace cnc;

extern "C" int cnc_field_weight_grads_workspace(const cnc_field_wgrad_t* f, uint64_t* bytes)
{
    if (!f || !bytes) return CNC_ERR_INVALID_VALUE;
    uint32_t n = 0;
    int rc = wgrad_workgroups(f->n_workgroups, &n);
    if (rc != CNC_OK) return rc;
    WGradArgs p{};
    size_t    lds = 0;
    return wgrad_plan(f, n, &p, &lds, bytes);
}

extern "C" int cnc_field_weight_grads(const cnc_field_wgrad_t* f, void* stream)
{
    if (!f) return CNC_ERR_INVALID_VALUE;
    if (f->N == 0) return CNC_ERR_INVALID_VALUE;
    uint32_t n = 0;
    int rc = wgrad_workgroups(f->n_workgroups, &n);
    if (rc != CNC_OK) return rc;
    WGradArgs p{};
    size_t    lds = 0;
    uint64_t  ws = 0;
    rc = wgrad_plan(f, n, &p, &lds, &ws);
    if (rc != CNC_OK) return rc;
    if (!f->workspace || f->workspace_bytes < ws || !f->g_max) return CNC_ERR_INVALID_VALUE;
    uint32_t max_elems = 0, wgs = 0;
    for (int l = 0; l < 5; l++) {
        const WGradRole& R = p.role[l];
        if (!R.G || !R.A || !R.out || R.ld_out < R.n_out_cols || R.n_out_cols + (R.gap_col < R.n_out_cols + 1 ? 1u : 0u) > R.nI)
            return CNC_ERR_INVALID_VALUE;
        // 32-bit byte offsets (buffer resources)
        if ((uint64_t)f->N * (R.ldG > R.ldA ? R.ldG : R.ldA) * 4u >= 0xFFFFFFF0ull) return CNC_ERR_UNSUPPORTED;
        if (R.nO * R.n_out_cols > max_elems) max_elems = R.nO * R.n_out_cols;
        wgs = R.wg0 + R.n_wg;
    }
    p.N = f->N; p.partial = f->workspace; p.g_max = f->g_max;
    hipStream_t s = (hipStream_t)stream;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_field_wgrad), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return CNC_ERR_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_field_wgrad, dim3(wgs), dim3(kWgThreads), lds, s, p);
    hipLaunchKernelGGL(k_field_wgrad_reduce, dim3((max_elems + 255) / 256, 5), dim3(256), 0, s, p);
    return launch_status();
}
