// field_mma.hpp — the three-product fp16 matrix machinery shared by the fused field kernels (field_fused2.hip: the
// gradient-free forward; field_bwd.hip: the gradient chain of the training step): activation planes in LDS, weight
// fragments of the 16x16x32 form through a buffer resource, the transposed product, a layer over the planes.
#pragma once
#include "field_fused_common.hpp"

namespace cnc {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr uint32_t kCP = 40;              // halves per row of a chunk plane (80 bytes: conflict-free b128 rows)
constexpr float kWScaleInv = 1.0f / 256.0f;

// Activation planes: 32 rows x H halves.  H = 160: unpadded, 16-byte chunks XOR-swizzled by (row >> 2) & 3 (rows are
// 320 bytes apart: rows r and r + 4 would start on the same banks); H = 64: rows padded by 8 halves.
template <int NT>
struct Plane2 {
    static constexpr uint32_t ld = NT == 5 ? 160u : NT * 32u + 8u;
    static constexpr bool     swz = NT == 5;
    static __device__ __forceinline__ uint32_t at(uint32_t r, uint32_t c)
    {
        if constexpr (swz) return r * ld + ((((c >> 3) ^ ((r >> 2) & 3u)) << 3) | (c & 7u));
        else return r * ld + c;
    }
    // column part of `at` for a row whose (row >> 2) & 3 is `s`
    static __device__ __forceinline__ uint32_t col_at(uint32_t c, uint32_t s)
    {
        if constexpr (swz) return (((c >> 3) ^ s) << 3) | (c & 7u);
        else return c;
    }
};

// Weight fragments of the 16x16x32 form (cnc_field_pack_all): per (K-step of 32, column
// block of 16): 64 lanes x 8 halves hi, then lo, of 2^8 W[16 cb + (lane & 15)][32 ks + 8 (lane >> 4) + 0..7].
// voff = 16 lane + 2048 (first column block of this wave).
template <int NCB>
__device__ __forceinline__ void load_wq(wrsrc_t W, uint32_t ks, uint32_t ncbt, uint32_t voff, half8_t (&hi)[NCB],
                                        half8_t (&lo)[NCB])
{
    const int32_t soff = (int32_t)(ks * ncbt * 2048u);
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        const f32x4_t a = llvm_raw_buffer_load_f32x4(W, (int32_t)(voff + cb * 2048), soff, 0);
        const f32x4_t b = llvm_raw_buffer_load_f32x4(W, (int32_t)(voff + cb * 2048 + 1024), soff, 0);
        hi[cb] = __builtin_bit_cast(half8_t, a);
        lo[cb] = __builtin_bit_cast(half8_t, b);
    }
}

// The products run TRANSPOSED: the weight fragment is the MFMA's A operand (rows = output features), the activation fragment
// its B operand (columns = samples) — the fragments themselves are what they were — so that in the result lane (r, kq)
// holds, for sample r of the row block, the FOUR CONSECUTIVE output features 4 kq .. 4 kq + 3 of the column block.  They are
// four consecutive K of the next layer's row: bias, ReLU, split and ONE 8-byte LDS write per half plane instead of four
// 2-byte writes per plane and value (the colour kernel's write-backs were ~1000 LDS instructions per tile), and one bias /
// w2 vector load per column block instead of a scalar per lane.
template <int NRB, int NCB>
__device__ __forceinline__ void mfma3q(const half8_t (&ah)[NRB], const half8_t (&al)[NRB], const half8_t (&wh)[NCB],
                                       const half8_t (&wl)[NCB], f32x4 (&acc)[NRB][NCB])
{
    // the two small products first; consecutive instructions go to different accumulators
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int rb = 0; rb < NRB; rb++) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[cb], al[rb], acc[rb][cb], 0, 0, 0);
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int rb = 0; rb < NRB; rb++) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[cb], ah[rb], acc[rb][cb], 0, 0, 0);
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int rb = 0; rb < NRB; rb++) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[cb], ah[rb], acc[rb][cb], 0, 0, 0);
}


template <int NRB, int NCB>
__device__ __forceinline__ void zero_q(f32x4 (&acc)[NRB][NCB])
{
#pragma unroll
    for (int rb = 0; rb < NRB; rb++)
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) acc[rb][cb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}

// acc[rb][cb] (rows 16 (row_block0 + rb) + ..., this wave's NCB column blocks) = A * W^T with A in the activation planes,
// K = 32 nks.
template <int NRB, int NCB, int NT, bool DB>
__device__ __forceinline__ void layer_q(const half_t* __restrict__ a_hi, const half_t* __restrict__ a_lo, uint32_t nks,
                                        const half_t_* __restrict__ Wq, uint32_t ncbt, uint32_t cb0, uint32_t row_block0,
                                        f32x4 (&acc)[NRB][NCB], uint32_t lane)
{
    using P = Plane2<NT>;
    const uint32_t r = lane & 15u, kq = lane >> 4;
    const wrsrc_t  W = weight_rsrc(reinterpret_cast<const float*>(Wq), nks * ncbt * 2048u);
    const uint32_t voff = lane * 16u + cb0 * 2048u;
    const uint32_t abase = (row_block0 * 16u + r) * P::ld + (P::swz ? ((kq ^ ((r >> 2) & 3u)) << 3) : (kq << 3));
    zero_q<NRB, NCB>(acc);
    if constexpr (DB) {
        // the next K-step's fragments are requested before the current one's products (two register sets)
        half8_t wh0[NCB], wl0[NCB], wh1[NCB], wl1[NCB];
        load_wq<NCB>(W, 0, ncbt, voff, wh0, wl0);
        for (uint32_t ks = 0; ks < nks; ks += 2) {
            const bool second = ks + 1 < nks;
            if (second) load_wq<NCB>(W, ks + 1, ncbt, voff, wh1, wl1);
            {
                half8_t ah[NRB], al[NRB];
#pragma unroll
                for (int rb = 0; rb < NRB; rb++) {
                    ah[rb] = *reinterpret_cast<const half8_t*>(a_hi + abase + rb * 16 * P::ld + ks * 32);
                    al[rb] = *reinterpret_cast<const half8_t*>(a_lo + abase + rb * 16 * P::ld + ks * 32);
                }
                mfma3q<NRB, NCB>(ah, al, wh0, wl0, acc);
            }
            if (second) {
                if (ks + 2 < nks) load_wq<NCB>(W, ks + 2, ncbt, voff, wh0, wl0);
                half8_t ah[NRB], al[NRB];
#pragma unroll
                for (int rb = 0; rb < NRB; rb++) {
                    ah[rb] = *reinterpret_cast<const half8_t*>(a_hi + abase + rb * 16 * P::ld + ks * 32 + 32);
                    al[rb] = *reinterpret_cast<const half8_t*>(a_lo + abase + rb * 16 * P::ld + ks * 32 + 32);
                }
                mfma3q<NRB, NCB>(ah, al, wh1, wl1, acc);
            }
        }
    } else {
        // one register set: a K-step's fragments are requested as soon as the products of the step before have been
        // issued (they read their operands at issue); the other waves of the SIMD cover the round trip
        half8_t wh[NCB], wl[NCB];
        load_wq<NCB>(W, 0, ncbt, voff, wh, wl);
        for (uint32_t ks = 0; ks < nks; ks++) {
            half8_t ah[NRB], al[NRB];
#pragma unroll
            for (int rb = 0; rb < NRB; rb++) {
                ah[rb] = *reinterpret_cast<const half8_t*>(a_hi + abase + rb * 16 * P::ld + ks * 32);
                al[rb] = *reinterpret_cast<const half8_t*>(a_lo + abase + rb * 16 * P::ld + ks * 32);
            }
            mfma3q<NRB, NCB>(ah, al, wh, wl, acc);
            if (ks + 1 < nks) load_wq<NCB>(W, ks + 1, ncbt, voff, wh, wl);
        }
    }
}

}  // namespace cnc
