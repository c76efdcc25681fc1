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
