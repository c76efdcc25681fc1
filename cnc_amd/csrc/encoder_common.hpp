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

// Corner set-up for one (point, level): weights, validity and row indices.
// Mirrors gridencoder.cu:166-291 (forward) / :443-562 (backward).
template <uint32_t D, bool VXL>
struct Corners {
    static constexpr uint32_t C = 1u << D;
    float    w[C];
    uint32_t row[C];
    bool     valid[C];
    float    wn_re;

    __device__ __forceinline__ void setup(const float (&x)[D], uint32_t R, uint32_t hs,
                                          uint32_t Rb, const uint8_t* __restrict__ vxl,
                                          const int32_t* __restrict__ sat = nullptr)
    {
        float    pos[D];
        uint32_t g[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            float p = x[d] * (float)(R - 2);   // float*float, rounded
            p = p + 0.5f;                      // == (float)((double)p + 0.5)
            const float fl = floorf(p);
            g[d] = (uint32_t)fl;
            pos[d] = p - fl;
        }
        float wn = 0;
#pragma unroll
        for (uint32_t i = 0; i < C; i++) {
            float    wi = 1;
            uint32_t q[D];
            bool     border = false;
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                if ((i & (1u << d)) == 0) {
                    wi *= 1 - pos[d];
                    q[d] = g[d];
                } else {
                    wi *= pos[d];
                    q[d] = min(g[d] + 1, R - 1);
                }
                border |= (q[d] == 0) | (q[d] == R - 1);
            }
            bool ok = !border;
            if constexpr (VXL) {
                // the reference evaluates the box for every corner; its result only matters
                // for non-border ones, so skip the (expensive) scan otherwise
                if (ok) ok = sat ? box_any_sat<D>(q, R, Rb, sat) : box_any<D>(q, R, Rb, vxl);
            }
            w[i] = wi;
            valid[i] = ok;
            row[i] = ok ? grid_row<D>(q, hs, R) : 0u;
            wn += ok ? wi : 0.0f;
        }
        if (wn == 0) wn = 1e-9f;   // (float)(0.0 + 1e-9)
        wn_re = 1.0f / wn;         // == (float)(1.0 / (double)wn)
    }
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
