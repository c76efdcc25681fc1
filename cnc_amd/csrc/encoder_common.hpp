// encoder_common.hpp — device code shared by the hash-grid encoder translation units:
// vector load/store helpers, the per-(point, level) corner set-up and the point loader.
#pragma once

#include "common.hpp"

namespace cnc {

template <uint32_t V> struct vecf;
template <> struct vecf<1> { using type = float; };
template <> struct vecf<2> { using type = float2; };
template <> struct vecf<4> { using type = float4; };

template <uint32_t V>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&v)[V])
{
    using T = typename vecf<V>::type;
    T t = *reinterpret_cast<const T*>(p);
    const float* f = reinterpret_cast<const float*>(&t);
#pragma unroll
    for (uint32_t i = 0; i < V; i++) v[i] = f[i];
}

template <uint32_t V>
__device__ __forceinline__ void store_vec(float* __restrict__ p, const float (&v)[V])
{
    using T = typename vecf<V>::type;
    T t;
    float* f = reinterpret_cast<float*>(&t);
#pragma unroll
    for (uint32_t i = 0; i < V; i++) f[i] = v[i];
    *reinterpret_cast<T*>(p) = t;
}

// Streaming store: the line is not kept in the L2 (the encoder's output is read back much later by another kernel; the
// bit plane / table it gathers from should keep the cache).
template <uint32_t V>
__device__ __forceinline__ void store_vec_nt(float* __restrict__ p, const float (&v)[V])
{
    if constexpr (V == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 t = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(t, reinterpret_cast<f4*>(p));
    } else if constexpr (V == 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 t = {v[0], v[1]};
        __builtin_nontemporal_store(t, reinterpret_cast<f2*>(p));
    } else {
        __builtin_nontemporal_store(v[0], p);
    }
}

// Corner set-up for one (point, level): weights, validity and row indices.
// Mirrors gridencoder.cu:166-291 (forward) / :443-562 (backward).
template <uint32_t D, bool VXL>
struct Corners {
    static constexpr uint32_t C = 1u << D;
    float    w[C];
    uint32_t row[C];
    bool     valid[C];
    float    wn_re;
    uint32_t cell[D];     // integer cell coordinates (floor of the scaled position)
    float    frac[D];     // fractional position inside the cell

    __device__ __forceinline__ void setup(const float (&x)[D], uint32_t R, uint32_t hs,
                                          uint32_t Rb, const uint8_t* __restrict__ vxl,
                                          const int32_t* __restrict__ sat = nullptr,
                                          const uint32_t* __restrict__ vplane = nullptr)
    {
        float    pos[D];
        uint32_t g[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            float p = x[d] * (float)(R - 2);   // float*float, rounded
            p = p + 0.5f;                      // == (float)((double)p + 0.5)
            const float fl = floorf(p);
            g[d] = (uint32_t)fl;
            cell[d] = g[d];
            pos[d] = p - fl;
            frac[d] = pos[d];
        }
        // Row index per corner = grid_row(q, hs, R), assembled from per-axis terms: every axis has
        // only two candidate coordinates, so the 32-bit multiplies (quarter rate on CDNA) are done
        // once per axis and value (2 D of them) instead of once per corner and axis (D 2^D), and the
        // dense / hashed decision is made once (it is wave-uniform whenever the level is).
        constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u,
                                        2097192037u, 1434869437u, 2165219737u};
        uint32_t qa[D][2], part[D][2];
        uint32_t stride = 1, sd[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {                 // the stride walk of grid_row
            sd[d] = stride;
            if (stride <= hs) stride *= R;
        }
        const bool hashed = stride > hs;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            qa[d][0] = g[d];
            qa[d][1] = min(g[d] + 1, R - 1);
            const uint32_t m = hashed ? primes[d] : sd[d];
#pragma unroll
            for (uint32_t b = 0; b < 2; b++) part[d][b] = d == 0 ? (hashed ? qa[d][b] : qa[d][b] * m) : qa[d][b] * m;
        }
        // dense vertex index per axis value for the vertex bit plane (x fastest: the two x-neighbours of a corner
        // pair share a word)
        uint32_t vpart[D][2];
        if constexpr (VXL) {
            uint32_t vs = 1;
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                vpart[d][0] = qa[d][0] * vs;
                vpart[d][1] = qa[d][1] * vs;
                vs *= R;
            }
        }
        const bool pow2 = (hs & (hs - 1)) == 0;
        float wn = 0;
#pragma unroll
        for (uint32_t i = 0; i < C; i++) {
            float    wi = 1;
            uint32_t q[D];
            uint32_t index = 0;
            bool     border = false;
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                const uint32_t bit = (i >> d) & 1u;
                wi *= bit ? pos[d] : 1 - pos[d];
                q[d] = qa[d][bit];
                index = hashed ? index ^ part[d][bit] : index + part[d][bit];
                border |= (q[d] == 0) | (q[d] == R - 1);
            }
            bool ok = !border;
            if constexpr (VXL) {
                // the reference evaluates the box for every corner; its result only matters
                // for non-border ones, so skip the (expensive) scan otherwise
                // (a level with a vertex bit plane — the same predicate evaluated once per vertex and occupancy
                // update — reads ONE bit here instead of 2^D table entries)
                if (ok) {
                    if (vplane) {
                        uint32_t vi = 0;
#pragma unroll
                        for (uint32_t d = 0; d < D; d++) vi += vpart[d][(i >> d) & 1u];
                        ok = (vplane[vi >> 5] >> (vi & 31u)) & 1u;
                    } else {
                        ok = sat ? box_any_sat<D>(q, R, Rb, sat) : box_any<D>(q, R, Rb, vxl);
                    }
                }
            }
            if (pow2) index &= hs - 1;
            else if (index >= hs) index %= hs;
            w[i] = wi;
            valid[i] = ok;
            row[i] = ok ? index : 0u;
            wn += ok ? wi : 0.0f;
        }
        if (wn == 0) wn = 1e-9f;   // (float)(0.0 + 1e-9)
        wn_re = 1.0f / wn;         // == (float)(1.0 / (double)wn)
    }
};

// F sign bits of one table row from the bit plane cnc_pack_sign_bits writes (bit k of row r = table[r][k] >= 0)
template <uint32_t F>
__device__ __forceinline__ uint32_t load_row_bits(const uint8_t* __restrict__ bits, uint64_t row)
{
    if constexpr (F == 32) return *reinterpret_cast<const uint32_t*>(bits + row * 4);
    else if constexpr (F == 16) return *reinterpret_cast<const uint16_t*>(bits + row * 2);
    else if constexpr (F == 8) return bits[row];
    else {
        const uint64_t bit = row * F;
        return (bits[bit >> 3] >> (bit & 7)) & ((1u << F) - 1u);
    }
}

// One 16-byte record per unit, built by the caller from the encoders' level tables (cnc_fused_field_t.units): a lane
// needs ONE L1-resident load before it can form its corner rows.  Reading the level tables through the encoder array
// of the kernel arguments (a dynamically indexed pointer, then the table entry, then the sign bytes) put three
// dependent memory round trips in front of every unit.
struct UnitRec {
    uint32_t off, hs, R, enc;
};

// The same F features with the vector work cut down (566 -> ~370 instructions per 3-D unit; the kernel was 53 % vector
// issue, tools/pmc_field.sh), BIT-IDENTICAL to `unit_features` / k_grid_encode_fwd_bits on the units a GridEncoder makes:
//   * a level is either dense (R^D <= rows: index = q0 + q1 R + q2 R^2 < rows) or hashed into a power-of-two table
//     (index = xor of primes & (rows - 1)) — the host refuses anything else (`FusedFieldForward._unit_table`) — so every
//     index is in range by construction: no modulo, no per-corner branch around the gather (an invalid corner's byte
//     is read and multiplied by a zero weight), coordinates of an outside point are replaced by 0 first;
//   * per-axis work is shared by the corners: 2 D multiplies for the index parts, the D = 3 weights as four x-y
//     products times two z factors (same association (wx wy) wz), border tests per axis value;
//   * the sign goes into the weight with shift + v_bfi (the weight is non-negative) and is ADDED: fmaf(tw, +-1, acc)
//     is acc +- tw rounded once — the same value — at 3 instead of 4 instructions per (corner, feature).
// ... in two halves: everything up to the gathers (their results stay in flight in `u.rb`), then the weights' sum, the
// division and the features — so that a lane can have two units' gathers under way before it consumes either.
struct UnitFast {
    float    m[8];        // corner weight, 0 for a border corner or an outside point
    uint32_t rb[8];       // the corner rows' sign bits (bits 0 .. F-1)
    float    wn;          // sum of m in corner order
};

template <uint32_t D, uint32_t F>
__device__ __forceinline__ void unit_issue_fast(const float (&x_)[D], bool inside, const uint8_t* __restrict__ bits,
                                                const UnitRec& r, UnitFast& u)
{
    static_assert(D == 2 || D == 3, "planes and volumes");
    constexpr uint32_t C = 1u << D;
    float (&m)[8] = u.m;
    uint32_t (&rb)[8] = u.rb;
    const uint32_t R = r.R, hs = r.hs;
    // the stride walk of grid_row: hashed iff the level does not fit its table
    uint32_t stride = 1, sd[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        sd[d] = stride;
        if (stride <= hs) stride *= R;
    }
    const bool     hashed = stride > hs;
    const uint32_t mask = hashed ? hs - 1u : 0xFFFFFFFFu;
    constexpr uint32_t primes[3] = {1u, 2654435761u, 805459861u};
    uint32_t pa[D][2];
    float    wa[D][2];
    bool     ba[D][2];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const float xd = inside ? x_[d] : 0.0f;
        float p = xd * (float)(R - 2);
        p = p + 0.5f;
        const float    fl = floorf(p);
        const uint32_t g = (uint32_t)fl, q1 = min(g + 1u, R - 1u);
        wa[d][1] = p - fl;
        wa[d][0] = 1 - wa[d][1];
        ba[d][0] = (g == 0u) | (g == R - 1u);
        ba[d][1] = (q1 == 0u) | (q1 == R - 1u);
        if (d == 0) {
            pa[d][0] = g;
            pa[d][1] = q1;
        } else {
            // q1 = g + 1 here (x in [0, 1] puts g at R - 2 at most: the clamp never bites), so its part is one ADD behind
            // g's multiply — the same uint32 value as (g + 1) m — instead of a second quarter-rate v_mul_lo_u32
            const uint32_t m = hashed ? primes[d] : sd[d];
            pa[d][0] = g * m;
            pa[d][1] = pa[d][0] + m;
        }
    }
    float w01[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) w01[j] = wa[0][j & 1u] * wa[1][j >> 1];
    float    wn = 0;
    // (the level's first row goes into the 32-bit row, not into the pointer: with a wave-uniform `bits` the gathers are
    // scalar-base + 32-bit-offset loads — no 64-bit address pair per corner)
    uint32_t index[C];
#pragma unroll
    for (uint32_t i = 0; i < C; i++) {
        const uint32_t b0 = i & 1u, b1 = (i >> 1) & 1u, b2 = D == 3 ? (i >> 2) & 1u : 0u;
        float    wi = w01[i & 3u];
        bool     border = ba[0][b0] | ba[1][b1];
        uint32_t ix = pa[0][b0] ^ pa[1][b1], ia = pa[0][b0] + pa[1][b1];
        if constexpr (D == 3) {
            wi = wi * wa[2][b2];
            border = border | ba[2][b2];
            ix ^= pa[2][b2];
            ia += pa[2][b2];
        }
        index[i] = (hashed ? ix : ia) & mask;
        m[i] = (!border && inside) ? wi : 0.0f;
        wn += m[i];
    }
    // (Serving the two x-neighbours of a corner pair with ONE 16-bit load — adjacent bytes on a dense level, an aligned
    // byte pair on a hashed level when the cell's x is even — was built and measured: level-major forward 0.218 -> 0.255 ms
    // per 2^20 marched samples, fused field unchanged at 0.855 ms on uniform points; the selects and the second,
    // conditional load cost more than the lookups saved.  One byte gather per corner it stays.)
#pragma unroll
    for (uint32_t i = 0; i < C; i++) rb[i] = load_row_bits<F>(bits, (uint64_t)(uint32_t)(r.off + index[i]));
    u.wn = wn;
}

template <uint32_t D, uint32_t F>
__device__ __forceinline__ void unit_finish_fast(const UnitFast& u, float (&acc)[F])
{
    constexpr uint32_t C = 1u << D;
    float wn = u.wn;
    if (wn == 0) wn = 1e-9f;
    const float wn_re = 1.0f / wn;
#pragma unroll
    for (uint32_t k = 0; k < F; k++) acc[k] = 0;
#pragma unroll
    for (uint32_t i = 0; i < C; i++) {
        const uint32_t tw = __builtin_bit_cast(uint32_t, u.m[i] * wn_re);        // >= +0
        const uint32_t nb = ~u.rb[i];                                            // bit k clear = feature +1
#pragma unroll
        for (uint32_t k = 0; k < F; k++) {
            // sign from bit k of nb, magnitude from tw: shift + v_bfi_b32 (the compiler's own choice for the C
            // expression is shift + and + or: VOP3 takes no literal on gfx9, so it will not form the bfi by itself)
            uint32_t sw;
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(sw) : "s"(0x80000000u), "v"(nb << (31u - k)), "v"(tw));
            acc[k] = acc[k] + __builtin_bit_cast(float, sw);
        }
    }
}

template <uint32_t D, uint32_t F>
__device__ __forceinline__ void unit_features_fast(const float (&x)[D], bool inside, const uint8_t* __restrict__ bits,
                                                   const UnitRec& r, float (&acc)[F])
{
    UnitFast u;
    unit_issue_fast<D, F>(x, inside, bits, r, u);
    unit_finish_fast<D, F>(u, acc);
}

// One backward call as the cell-merging scatter (grid_encode_cells.hip) takes it.
struct CellsArgs {
    const float*    grad;
    const float*    inputs;
    const float*    emb;
    const int32_t*  offsets;
    const int32_t*  resolutions;
    float*          grad_emb;
    const uint8_t*  vxl;
    const int32_t*  mli;
    const uint32_t* clip_count;
    const int32_t*  sat;
    FeatLayout      lay;
    uint32_t        N, L, Rb;
    uint32_t        carry;           // != 0: shared vertices of x-neighbour cells go out once (CNC_FLAG_CELL_CARRY)
};

template <uint32_t D>
__device__ __forceinline__ bool load_point(const float* __restrict__ inputs, uint32_t b,
                                           float (&x)[D])
{
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        x[d] = inputs[(size_t)b * D + d];
        oob |= (x[d] < 0) | (x[d] > 1);
    }
    return !oob;
}

}  // namespace cnc
