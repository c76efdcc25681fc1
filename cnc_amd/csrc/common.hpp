// common.hpp — shared device helpers for libcnc_hip.so (gfx950 only).
//
// Arithmetic policy: the library is compiled with -ffp-contract=off, so every float expression
// below is evaluated exactly as written (IEEE fp32, round-to-nearest, correctly rounded division).
// __builtin_fmaf appears only where nvcc's default -fmad=true would contract the reference's
// expression (DESIGN.md "Arithmetic policy").  Where the reference computes in double because of
// a C++ double literal and rounds back to float (gridencoder.cu:173,224,228,291;
// aligner_kernel.cu:14,19,25), the operation is one of +,-,*,/ on fp32-representable operands, for
// which rounding the fp64 result to fp32 equals the correctly rounded fp32 operation (53 >= 2*24+2
// bits; Figueroa 1995), so plain fp32 ops are used and fp64 never appears on the device.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "cnc_hip.h"

namespace cnc {

constexpr int kWave = 64;

__host__ __device__ inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// Where the F features of (level slot, point b) live.  ld == 0: the reference's level-major
// [L, N, F] tensor (gridencoder.cu:131).  ld != 0: point-major rows of a wider [N, ld] feature
// matrix, this encoder's block starting at column `col` — lets several encoders write straight
// into the MLP input (no permute / cat) and read its gradient in place.
struct FeatLayout {
    uint32_t ld;
    uint32_t col;
    uint32_t finest_first = 0;   // backward only: walk the level slots from the last one down
    uint32_t n_slots = 0;        // backward only: 1-D grid, block id = chunk * n_slots + level slot
    // optional per-level vertex bit planes of the occupancy mask (cnc_grid_vertex_bits): bit q0 + R (q1 + R q2) of
    // the level's plane = box_any(q); vboff[level] = first 32-bit word of the plane, < 0 = the level has none
    const uint32_t* vbits = nullptr;
    const int32_t*  vboff = nullptr;
    uint32_t        nt = 0;      // forward: streaming (non-temporal) stores of the outputs
};

__device__ __forceinline__ const uint32_t* vertex_plane(const FeatLayout& lay, uint32_t level)
{
    if (lay.vbits == nullptr) return nullptr;
    const int32_t w = lay.vboff[level];
    return w < 0 ? nullptr : lay.vbits + w;
}

__device__ __forceinline__ size_t feat_index(FeatLayout lay, uint32_t slot, uint32_t N, uint32_t b,
                                             uint32_t F)
{
    return lay.ld ? (size_t)b * lay.ld + lay.col + slot * F : ((size_t)slot * N + b) * F;
}

inline int launch_status()
{
    return hipGetLastError() == hipSuccess ? CNC_OK : CNC_ERR_LAUNCH;
}

// Row index of a grid vertex: dense (x + y*R + z*R^2) while the level fits its table, otherwise
// the xor-of-primes spatial hash, then modulo the table size (gridencoder.cu:45-87).  uint32 wrap.
template <uint32_t D>
__device__ __forceinline__ uint32_t grid_row(const uint32_t (&q)[D], uint32_t hashmap_size,
                                             uint32_t resolution)
{
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u,
                                    2097192037u, 1434869437u, 2165219737u};
    uint32_t stride = 1, index = 0;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (stride <= hashmap_size) {
            index += q[d] * stride;
            stride *= resolution;
        }
    }
    if (stride > hashmap_size) {
        index = 0;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) index ^= q[d] * primes[d];
    }
    // index % hashmap_size without the ~25-instruction u32 division in the common cases: hashed
    // levels have a power-of-two table (mask), dense levels have index < R^D <= hashmap_size
    // (identity).  Anything else (odd caller-made tables, uint32 wrap of R^D) takes the division.
    if ((hashmap_size & (hashmap_size - 1)) == 0) return index & (hashmap_size - 1);
    return index < hashmap_size ? index : index % hashmap_size;
}

// [lo, hi] range of occupancy cells covered by the +-1 vertex box of grid coordinate q along one
// axis (gridencoder.cu:224-240 == aligner_kernel.cu:19-41).  pn is returned for the overlap math.
__device__ __forceinline__ void box_range(float qf, float scale_re, uint32_t Rb, uint32_t& lo,
                                          uint32_t& hi, float& pn)
{
    pn = (qf - 0.5f) * scale_re;
    const float top = (float)(Rb - 1);
    float g1 = (pn - scale_re) * (float)Rb;
    g1 = g1 < 0 ? 0 : g1;
    g1 = g1 > top ? top : g1;
    lo = (uint32_t)(int)g1;
    float g2 = (pn + scale_re) * (float)Rb;
    g2 = g2 < 0 ? 0 : g2;
    g2 = g2 > top ? top : g2;
    hi = (uint32_t)(int)g2;
}

// true iff any occupancy cell in the vertex box is set (gridencoder.cu:221-276)
template <uint32_t D>
__device__ __forceinline__ bool box_any(const uint32_t (&q)[D], uint32_t R, uint32_t Rb,
                                        const uint8_t* __restrict__ vxl)
{
    const float scale_re = 1.0f / ((float)R - 2.0f);
    uint32_t lo[D], hi[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        float pn;
        box_range((float)q[d], scale_re, Rb, lo[d], hi[d], pn);
    }
    if constexpr (D == 1) {
        for (uint32_t a = lo[0]; a <= hi[0]; a++)
            if (vxl[a]) return true;
    } else if constexpr (D == 2) {
        for (uint32_t a = lo[0]; a <= hi[0]; a++)
            for (uint32_t b = lo[1]; b <= hi[1]; b++)
                if (vxl[a * Rb + b]) return true;
    } else {
        for (uint32_t a = lo[0]; a <= hi[0]; a++)
            for (uint32_t b = lo[1]; b <= hi[1]; b++)
                for (uint32_t c = lo[2]; c <= hi[2]; c++)
                    if (vxl[(a * Rb + b) * Rb + c]) return true;
    }
    return false;
}

// Same predicate from a summed-volume table of the occupancy grid: sat[(a*P + b)*P + c], P = Rb+1,
// = number of set cells with index < (a, b, c) per axis.  The box count is an integer
// inclusion-exclusion of 2^D entries, so the result equals the scan's, at 2^D loads instead of up
// to (2*Rb/(R-2)+1)^D byte reads (343 for a coarse context level against a 128^3 grid).
template <uint32_t D>
__device__ __forceinline__ bool box_any_sat(const uint32_t (&q)[D], uint32_t R, uint32_t Rb,
                                            const int32_t* __restrict__ sat)
{
    const float scale_re = 1.0f / ((float)R - 2.0f);
    uint32_t lo[D], hi[D];
    bool     empty = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        float pn;
        box_range((float)q[d], scale_re, Rb, lo[d], hi[d], pn);
        hi[d] += 1;   // exclusive
        empty |= hi[d] <= lo[d];
    }
    if (empty) return false;
    const uint32_t P = Rb + 1;
    int32_t cnt = 0;
#pragma unroll
    for (uint32_t m = 0; m < (1u << D); m++) {
        uint32_t idx = 0;
        int      sign = 1;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            const bool low = (m >> d) & 1u;
            idx = idx * P + (low ? lo[d] : hi[d]);
            if (low) sign = -sign;
        }
        cnt += sign * sat[idx];
    }
    return cnt > 0;
}

}  // namespace cnc
