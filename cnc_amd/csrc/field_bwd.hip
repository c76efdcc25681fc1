// field_bwd.hip — the gradient chain of the radiance field's two MLPs as ONE kernel (the training step's render pass).
//
// Reference: autograd through `mlp_head` / `mlp_base` of NGPRadianceField_mygrid_2D3D (ngp.py:506-547): sigmoid' ->
// Linear(H, 3)^T -> ReLU' -> Linear(H, H)^T -> ReLU' -> Linear(16 + geo, H)^T -> [geo part] + trunc_exp' (ngp.py:318-334) ->
// Linear(H, 1 + geo)^T -> ReLU' -> Linear(K0, H)^T -> the encoders' scatter.  The product ran that as five library GEMMs
// (g @ W), three ReLU-backward passes, the field_post backward and a sigmoid backward, each through HBM: ~1.6 ms of the
// GPU-bound half of a 9.8 ms training step.  Here a workgroup of two waves takes a 32-sample tile through the whole
// chain: the gradient matrices G5 .. G1 go to HBM once (the weight gradients dW_l = G_l^T A_l need them: they stay library
// split-K products) and stay in LDS as the next stage's operand; nothing is read back.
//
// Arithmetic: the three-product fp16 scheme of field_fused2.hip (x w ~= hi hi + hi lo + lo hi, fp32 accumulation, weights
// scaled by 2^8) — with one addition.  Gradients are SMALL (1e-3 .. 1e-8): split into halves as they are, their lo parts
// (and soon their hi parts) would be fp16 subnormals.  Every stage therefore scales its tile by a power of two chosen from
// the tile's largest magnitude (largest -> [2^13, 2^14)) before the split and divides the accumulators by it afterwards:
// exact operations; a value keeps 22 significant bits as long as it is within 1e5 of the tile's largest, and an absolute
// error of 2^-38 of the largest below that.  The two waves exchange their maxima through LDS at the barrier the stage needs
// anyway.
#include "field_mma.hpp"

namespace cnc {

struct FieldBwdArgs {
    uint32_t       N;               // rows (samples incl. the bucketed padding rows)
    uint32_t       geo;
    uint32_t       n_enc;           // encoder columns of the feature row = columns of dX that are written (a multiple of 16)
    uint32_t       flags;           // CNC_FIELD_SH_FP16 has no bearing here; reserved
    const float*   g_rgb;           // [N, 3] gradient w.r.t. the sigmoid's output (nullable: zero)
    const float*   g_density;       // [N]    gradient w.r.t. the density (nullable: zero)
    const float*   rgb;             // [N, 3] the forward's colours
    const float*   base_out;        // [N, ld_base] the base network's output (column 0 = raw density)
    uint32_t       ld_base;
    const uint8_t* selector;        // [N]
    const float*   h1;              // [N, H] post-ReLU activations (their sign is the ReLU mask)
    const float*   h3;
    const float*   h4;
    const half_t_* Wt[5];           // transposed fragments: stage 4 (head.4^T), 3 (head.2^T), 2 (head.0^T geo rows), 1 (base.2^T), 0 (base.0^T)
    float*         G5;              // [N, 4]   gradient w.r.t. head.4's output (column 3 = 0)
    float*         G4;              // [N, H]
    float*         G3;              // [N, H]
    float*         G2;              // [N, ld_g2] gradient w.r.t. base.2's output (columns 0 .. geo; the rest of the row zero)
    uint32_t       ld_g2;
    float*         G1;              // [N, H]
    float*         dX;              // [N, ld_x] columns [0, n_enc)
    uint32_t       ld_x;
    float*         bias_grads;      // [3 H + 84] zero-initialised by the caller: column sums of G4 | G3 | G1 | G2 (80) | G5 (4)
    uint32_t*      g_max;           // [5] zero-initialised (nullable): float bits of max |G1| .. max |G5| (atomic max)
};

// Row-major [N, ld] float matrices through buffer resources: SGPR base + ONE 32-bit lane offset + constants — a flat pointer
// per (row block, column block) cost 20 address registers per matrix and the kernel its third wave per SIMD — and with the
// record count set to the matrix's size a row past N reads zeros / drops its store: no bounds branches.
__device__ void llvm_raw_buffer_store_f32x4(f32x4_t data, i32x4_t rsrc, int32_t voffset, int32_t soffset, int32_t aux)
    __asm("llvm.amdgcn.raw.buffer.store.v4f32");

__device__ __forceinline__ float4 buf_load4(wrsrc_t m, uint32_t byte_off)
{
    const f32x4_t v = llvm_raw_buffer_load_f32x4(m, (int32_t)byte_off, 0, 0);
    return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ void buf_store4(wrsrc_t m, uint32_t byte_off, float a, float b, float c, float d)
{
    llvm_raw_buffer_store_f32x4(f32x4_t{a, b, c, d}, m, (int32_t)byte_off, 0, 0);
}

// largest magnitude of the tile -> the power of two that puts it into [2^13, 2^14)
__device__ __forceinline__ float tile_scale(float m)
{
    if (!(m > 0.0f) || !(m < 3.0e38f)) return 1.0f;
    const int e = __builtin_amdgcn_frexp_expf(m);              // m = f 2^e, f in [0.5, 1)
    return __builtin_amdgcn_ldexpf(1.0f, 14 - e);
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Write the lane's 2 x NCB x 4 values (sample rb * 16 + r, features 16 (cb0 + cb) + 4 kq + v), scaled by s, as halves
template <int NCB, int NT>
__device__ __forceinline__ void vals_to_planes(half_t* __restrict__ d_hi, half_t* __restrict__ d_lo, const float (&x)[2][NCB][4],
                                               uint32_t cb0, uint32_t lane, float s)
{
    using P = Plane2<NT>;
    const uint32_t r = lane & 15u, kq = lane >> 4;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            half4_t xh, xl;
#pragma unroll
            for (int v = 0; v < 4; v++) {
                half_t h, l;
                split_half(x[rb][cb][v] * s, h, l);
                xh[v] = h;
                xl[v] = l;
            }
            const uint32_t at = P::at(rb * 16u + r, (cb0 + cb) * 16u + 4u * kq);
            *reinterpret_cast<half4_t*>(d_hi + at) = xh;
            *reinterpret_cast<half4_t*>(d_lo + at) = xl;
        }
}

// Sum over the 16 lanes of a DPP row (the 16 samples r of a row block that share kq): four v_add_f32 with row_shr, the
// total ends up in lane 15 of the row.
__device__ __forceinline__ float row16_sum_to_lane15(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));   // row_shr:1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));   // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));   // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));   // row_shr:8
    return v;
}

// The ReLU operand of a column-split stage (post-activation values: only their sign is used), requested right behind the
// ISSUE of the stage's products — the weight registers are free by then and the round trip runs while the matrix pipe
// drains (requested in front of the products it costs 40 more live registers: two waves per SIMD instead of three):
// lane (r, kq) -> rows rb * 16 + r, columns 16 (w NCB + cb) + 4 kq ..
template <int NCB>
struct ActTile {
    float4 a[2][NCB];
    // `lane_off`: byte offset of (row0 + r, 16 w NCB + 4 kq) in a [N, H] matrix
    __device__ __forceinline__ void load(wrsrc_t act, uint32_t H, uint32_t lane_off)
    {
#pragma unroll
        for (int cb = 0; cb < NCB; cb++)
#pragma unroll
            for (int rb = 0; rb < 2; rb++) a[rb][cb] = buf_load4(act, lane_off + (rb * 16u * H + cb * 16u) * 4u);
    }
};

// One column-split stage with a ReLU mask: X = (acc / (2^8 s_in)) where act > 0 else 0; to HBM (G), its column sums (the
// bias gradient) into the workgroup's LDS accumulators `bsum`, and, scaled by the new tile scale, into the planes.  Returns
// the new scale.  Barriers: [all reads of the planes done + maxima exchanged] ... writes ... [visible].
template <int NCB, int NT>
__device__ __forceinline__ float masked_stage_out(const f32x4 (&acc)[2][NCB], float inv_in, const ActTile<NCB>& act,
                                                  wrsrc_t G, uint32_t H, uint32_t lane_off, uint32_t w,
                                                  uint32_t lane, half_t* h_hi, half_t* h_lo, float* xch, float* bsum,
                                                  uint32_t& run_max)
{
    const uint32_t r = lane & 15u, kq = lane >> 4;
    float x[2][NCB][4];
    float m = 0.0f;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        const uint32_t col0 = (w * NCB + cb) * 16u + 4u * kq;
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            const float4   a4 = act.a[rb][cb];
            const float    av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const float g = av[v] > 0.0f ? acc[rb][cb][v] * inv_in : 0.0f;
                x[rb][cb][v] = g;
                m = fmaxf(m, fabsf(g));
            }
            buf_store4(G, lane_off + (rb * 16u * H + cb * 16u) * 4u, x[rb][cb][0], x[rb][cb][1], x[rb][cb][2], x[rb][cb][3]);
        }
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const float t = row16_sum_to_lane15(x[0][cb][v] + x[1][cb][v]);
            if (r == 15u) atomicAdd(bsum + col0 + v, t);          // ds_add_f32: four lanes, distinct addresses
        }
    }
    m = wave_max(m);
    if (lane == 0) xch[w] = m;
    __syncthreads();                                   // every read of the planes issued and done; both maxima written
    const float tmax = fmaxf(xch[0], xch[1]);
    run_max = max(run_max, (uint32_t)__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tmax)));    // >= 0: ordered as integers
    const float s = tile_scale(tmax);
    vals_to_planes<NCB, NT>(h_hi, h_lo, x, w * NCB, lane, s);
    __syncthreads();
    return s;
}

#ifndef CNC_BWD_WAVES
#define CNC_BWD_WAVES 3
#endif
#ifndef CNC_BWD_DB
#define CNC_BWD_DB false
#endif
template <int NT>
__global__ __launch_bounds__(128, CNC_BWD_WAVES) void k_field_bwd_chain(FieldBwdArgs p)
{
    extern __shared__ float lds[];
    half_t* const lds16 = reinterpret_cast<half_t*>(lds);
    using P = Plane2<NT>;
    constexpr int      NCB = NT;
    constexpr uint32_t NCBT = 2 * NT, H = 32 * NT;
    constexpr int      NB2 = NT == 5 ? 5 : 4;
    const uint32_t tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u, r = lane & 15u, kq = lane >> 4;
    const uint32_t fi = tid & 31u, fq = tid >> 5;
    half_t* const h_hi = lds16;
    half_t* const h_lo = lds16 + 32 * P::ld;
    float* const  xch = reinterpret_cast<float*>(lds16 + 2 * 32 * P::ld);       // two maxima (+ two spare words)
    // the workgroup's bias-gradient accumulators: column sums of G4 | G3 | G1 (H each) | G2 (80) | G5 (4), flushed at the end
    float* const  bs4 = xch + 4, * const bs3 = bs4 + H, * const bs1 = bs3 + H, * const bs2 = bs1 + H, * const bs5 = bs2 + 80;
    for (uint32_t i = tid; i < 3 * H + 84; i += 128) bs4[i] = 0.0f;
    const uint32_t tiles = (p.N + 31u) / 32u;
    const uint32_t K2 = ((1u + p.geo + 31u) / 32u) * 32u;                        // stage 1's K: 1 + geo padded to 32
    const uint32_t bytes_h = p.N * H * 4u;
    const wrsrc_t  rH1 = weight_rsrc(p.h1, bytes_h), rH3 = weight_rsrc(p.h3, bytes_h), rH4 = weight_rsrc(p.h4, bytes_h);
    const wrsrc_t  rG1 = weight_rsrc(p.G1, bytes_h), rG3 = weight_rsrc(p.G3, bytes_h), rG4 = weight_rsrc(p.G4, bytes_h);
    const wrsrc_t  rG2 = weight_rsrc(p.G2, p.N * p.ld_g2 * 4u), rX = weight_rsrc(p.dX, p.N * p.ld_x * 4u);
    uint32_t gm1 = 0, gm2 = 0, gm3 = 0, gm4 = 0, gm5 = 0;     // float bits of the largest |G_l| this workgroup has seen (scalar)
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint32_t row0 = tile * 32, frow = row0 + fi;
        const bool     live = frow < p.N;
        // ---- stage 5: g5 = g_rgb * y (1 - y) (SigmoidBackward), into columns 0..2 of a 32-column operand ----
        float g5[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (live && fq == 0 && p.g_rgb) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float y = p.rgb[(size_t)frow * 3 + c];
                g5[c] = p.g_rgb[(size_t)frow * 3 + c] * ((1.0f - y) * y);
            }
        }
        if (live && fq == 0) *reinterpret_cast<float4*>(p.G5 + (size_t)frow * 4) = make_float4(g5[0], g5[1], g5[2], 0.0f);
        float m5 = fmaxf(fmaxf(fabsf(g5[0]), fabsf(g5[1])), fabsf(g5[2]));
        m5 = wave_max(m5);
#pragma unroll
        for (int c = 0; c < 3; c++) {                   // column sums of G5 (head.4's bias gradient)
            float t = g5[c];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o);
            if (lane == 0) atomicAdd(bs5 + c, t);
        }
        if (lane == 0) xch[w] = m5;
        __syncthreads();                               // (also: the previous tile's last reads of the planes are done)
        const float t5 = fmaxf(xch[0], xch[1]);
        gm5 = max(gm5, (uint32_t)__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t5)));
        float s_in = tile_scale(t5);
        {
            const RowF16 row{h_hi, h_lo};
            float v8[4] = {g5[0] * s_in, g5[1] * s_in, g5[2] * s_in, 0.0f}, z4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            // thread (fi, fq) owns columns [8 fq, 8 fq + 8) of row fi: the three gradients sit in the first window
            row.put<4>(P::at(fi, 8 * fq), fq == 0 ? v8 : z4);
            row.put<4>(P::at(fi, 8 * fq + 4), z4);
        }
        __syncthreads();
        f32x4 acc[2][NCB];
        ActTile<NCB> act;
        const uint32_t lane_off = ((row0 + r) * H + w * NCB * 16u + 4u * kq) * 4u;    // (row0 + r, this wave's first column + 4 kq)
        // ---- stage 4: G4 = (g5 W5) where h4 > 0 ----
        layer_q<2, NCB, NT, CNC_BWD_DB>(h_hi, h_lo, 1, p.Wt[0], NCBT, w * NCB, 0, acc, lane);
        __builtin_amdgcn_sched_barrier(0);
        act.load(rH4, H, lane_off);                      // behind the products' issue: in flight while the matrix pipe drains
        s_in = masked_stage_out<NCB, NT>(acc, kWScaleInv / s_in, act, rG4, H, lane_off, w, lane, h_hi, h_lo, xch, bs4, gm4);
        // ---- stage 3: G3 = (G4 W4) where h3 > 0 ----
        layer_q<2, NCB, NT, CNC_BWD_DB>(h_hi, h_lo, NT, p.Wt[1], NCBT, w * NCB, 0, acc, lane);
        __builtin_amdgcn_sched_barrier(0);
        act.load(rH3, H, lane_off);                      // behind the products' issue: in flight while the matrix pipe drains
        s_in = masked_stage_out<NCB, NT>(acc, kWScaleInv / s_in, act, rG3, H, lane_off, w, lane, h_hi, h_lo, xch, bs3, gm3);
        // ---- stage 2: G2[:, c] = (G3 W3[:, 15 + c]) for the geo features c >= 1; G2[:, 0] = g_density * d density / d raw ----
        {
            f32x4 acc2[1][NB2];
            layer_q<1, NB2, NT, CNC_BWD_DB>(h_hi, h_lo, NT, p.Wt[2], NB2, 0, w, acc2, lane);
            const float    inv_in = kWScaleInv / s_in;
            const uint32_t row = row0 + w * 16u + r;           // rows split between the waves
            float x2[NB2][4];
            float m = 0.0f;
            float d0 = 0.0f;                                    // the density's own gradient (column 0), lanes kq == 0
            if (kq == 0 && row < p.N && p.g_density && p.selector[row]) {
                const float raw = p.base_out[(size_t)row * p.ld_base];
                d0 = p.g_density[row] * expf(fminf(raw - 1.0f, 15.0f));       // trunc_exp's clamped derivative (ngp.py:318-334)
            }
#pragma unroll
            for (int cb = 0; cb < NB2; cb++) {
                const uint32_t c0 = cb * 16u + 4u * kq;
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    float g = (c0 + v >= 1u && c0 + v <= p.geo) ? acc2[0][cb][v] * inv_in : 0.0f;
                    if (cb == 0 && v == 0 && kq == 0) g = d0;
                    x2[cb][v] = g;
                    m = fmaxf(m, fabsf(g));
                }
                if (c0 < p.ld_g2) buf_store4(rG2, (row * p.ld_g2 + c0) * 4u, x2[cb][0], x2[cb][1], x2[cb][2], x2[cb][3]);
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const float t = row16_sum_to_lane15(x2[cb][v]);
                    if (r == 15u && c0 + v < 80u) atomicAdd(bs2 + c0 + v, t);
                }
            }
            m = wave_max(m);
            if (lane == 0) xch[w] = m;
            __syncthreads();
            const float t2 = fmaxf(xch[0], xch[1]);
            gm2 = max(gm2, (uint32_t)__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t2)));
            s_in = tile_scale(t2);
#pragma unroll
            for (int cb = 0; cb < NB2; cb++) {
                const uint32_t c0 = cb * 16u + 4u * kq;
                if (c0 >= K2) continue;
                half4_t xh, xl;
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    half_t h, l;
                    split_half(x2[cb][v] * s_in, h, l);
                    xh[v] = h;
                    xl[v] = l;
                }
                const uint32_t at = P::at(w * 16u + r, c0);
                *reinterpret_cast<half4_t*>(h_hi + at) = xh;
                *reinterpret_cast<half4_t*>(h_lo + at) = xl;
            }
            // columns [16 NB2, K2) of the operand (K padded to 32) are zero
            for (uint32_t c = NB2 * 16u + 4u * fq; c < K2; c += 16u) {
                const RowF16 rowz{h_hi, h_lo};
                float        z4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                rowz.put<4>(P::at(fi, c), z4);
            }
            __syncthreads();
        }
        // ---- stage 1: G1 = (G2 W2) where h1 > 0 ----
        layer_q<2, NCB, NT, CNC_BWD_DB>(h_hi, h_lo, K2 / 32, p.Wt[3], NCBT, w * NCB, 0, acc, lane);
        __builtin_amdgcn_sched_barrier(0);
        act.load(rH1, H, lane_off);                      // behind the products' issue: in flight while the matrix pipe drains
        s_in = masked_stage_out<NCB, NT>(acc, kWScaleInv / s_in, act, rG1, H, lane_off, w, lane, h_hi, h_lo, xch, bs1, gm1);
        // ---- stage 0: dX[:, :n_enc] = G1 W1[:, :n_enc], rows split between the waves, the column blocks in passes of six
        // (a pass's blocks past the last one multiply whatever follows in the fragment stream: their results are dropped) ----
        {
            const float    inv_in = kWScaleInv / s_in;
            const uint32_t row = row0 + w * 16u + r;
            constexpr int  PASS = 6;
            const uint32_t nb0 = (p.n_enc + 15u) / 16u;
            for (uint32_t pass = 0; pass * PASS < nb0; pass++) {
                f32x4 acc0[1][PASS];
                layer_q<1, PASS, NT, CNC_BWD_DB>(h_hi, h_lo, NT, p.Wt[4], nb0, pass * PASS, w, acc0, lane);
#pragma unroll
                for (int cb = 0; cb < PASS; cb++) {
                    const uint32_t c0 = (pass * PASS + cb) * 16u + 4u * kq;
                    if (c0 < p.n_enc)
                        buf_store4(rX, (row * p.ld_x + c0) * 4u, acc0[0][cb][0] * inv_in, acc0[0][cb][1] * inv_in, acc0[0][cb][2] * inv_in,
                                   acc0[0][cb][3] * inv_in);
                }
            }
        }
        // the next tile's first barrier (after its stage-5 maxima) orders its plane writes behind these reads
    }
    // ---- the bias gradients: this workgroup's column sums into the global accumulators (zeroed by the caller) ----
    __syncthreads();
    if (p.bias_grads)
        for (uint32_t i = tid; i < 3 * H + 84; i += 128) {
            const float v = bs4[i];
            if (v != 0.0f) unsafeAtomicAdd(p.bias_grads + i, v);
        }
    if (p.g_max && tid == 0) {                          // what cnc_field_weight_grads scales the gradient matrices by
        atomicMax(p.g_max + 0, gm1);
        atomicMax(p.g_max + 1, gm2);
        atomicMax(p.g_max + 2, gm3);
        atomicMax(p.g_max + 3, gm4);
        atomicMax(p.g_max + 4, gm5);
    }
}

}  // namespace cnc

using namespace cnc;

extern "C" int cnc_field_backward_chain(const cnc_field_bwd_t* f, void* stream)
{
    if (!f) return CNC_ERR_INVALID_VALUE;
    if (f->N == 0) return CNC_OK;
    const uint32_t H = f->n_neurons, F = f->n_features;
    if (!(H == 64 || H == 160) || !(F == 2 || F == 4 || F == 8)) return CNC_ERR_UNSUPPORTED;
    const uint32_t NT = H / 32, n_enc = f->n_enc_columns;
    if (n_enc == 0 || n_enc % 4 != 0 || (n_enc + 15) / 16 > 12 || 1 + f->geo_feat_dim > (NT == 5 ? 80u : 64u)) return CNC_ERR_UNSUPPORTED;
    if (((1 + f->geo_feat_dim + 31) / 32) * 32 > H) return CNC_ERR_UNSUPPORTED;
    // the matrices are addressed with 32-bit byte offsets (buffer resources)
    if ((uint64_t)f->N * (f->ld_x > H ? f->ld_x : H) * 4u >= (1ull << 32)) return CNC_ERR_UNSUPPORTED;
    if (!f->rgb || !f->base_out || !f->selector || !f->h1 || !f->h3 || !f->h4 || !f->G5 || !f->G4 || !f->G3 || !f->G2 || !f->G1 ||
        !f->dX || f->ld_base < 1 || f->ld_g2 < 1 + f->geo_feat_dim || f->ld_g2 % 4 != 0 || f->ld_x < n_enc ||
        f->ld_x % 4 != 0)
        return CNC_ERR_INVALID_VALUE;
    FieldBwdArgs p{};
    p.N = f->N; p.geo = f->geo_feat_dim; p.n_enc = n_enc;
    p.g_rgb = f->grad_rgb; p.g_density = f->grad_density; p.rgb = f->rgb; p.base_out = f->base_out; p.ld_base = f->ld_base;
    p.selector = f->selector; p.h1 = f->h1; p.h3 = f->h3; p.h4 = f->h4;
    for (int l = 0; l < 5; l++) {
        if (!f->packed_weights_t[l]) return CNC_ERR_INVALID_VALUE;
        p.Wt[l] = reinterpret_cast<const half_t_*>(f->packed_weights_t[l]);
    }
    p.G5 = f->G5; p.G4 = f->G4; p.G3 = f->G3; p.G2 = f->G2; p.ld_g2 = f->ld_g2; p.G1 = f->G1; p.dX = f->dX; p.ld_x = f->ld_x;
    const uint32_t ld = NT == 5 ? 160u : NT * 32u + 8u;
    const size_t   lds_bytes = (size_t)2 * 32 * ld * sizeof(half_t) + 16 + (size_t)(3 * H + 84) * sizeof(float);
    p.bias_grads = f->bias_grads;
    p.g_max = f->g_max;
    const uint32_t tiles = (p.N + 31u) / 32u;
    hipStream_t    s = (hipStream_t)stream;
    // the grid is what is resident at once (the workgroups loop over the tiles); asked once per thread and device
    static thread_local int cached_dev = -1;
    static thread_local uint32_t cached_n[2] = {0, 0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return CNC_ERR_LAUNCH;
    if (cached_dev != dev) {
        int cus = 0, per5 = 0, per2 = 0;
        const size_t lds5 = (size_t)2 * 32 * 160 * sizeof(half_t) + 16 + (size_t)(3 * 160 + 84) * sizeof(float);
        const size_t lds2 = (size_t)2 * 32 * 72 * sizeof(half_t) + 16 + (size_t)(3 * 64 + 84) * sizeof(float);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per5, k_field_bwd_chain<5>, 128, lds5) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per2, k_field_bwd_chain<2>, 128, lds2) != hipSuccess || cus <= 0 ||
            per5 <= 0 || per2 <= 0)
            return CNC_ERR_LAUNCH;
        cached_n[0] = (uint32_t)(per5 * cus);
        cached_n[1] = (uint32_t)(per2 * cus);
        cached_dev = dev;
    }
    const uint32_t resident = cached_n[NT == 5 ? 0 : 1];
    const uint32_t blocks = tiles < resident ? tiles : resident;
    if (NT == 5) hipLaunchKernelGGL((k_field_bwd_chain<5>), dim3(blocks), dim3(128), lds_bytes, s, p);
    else hipLaunchKernelGGL((k_field_bwd_chain<2>), dim3(blocks), dim3(128), lds_bytes, s, p);
    return launch_status();
}
