// Scratch probe: "bin by owner slab, accumulate in LDS" backward for hashed levels vs the atomic
// scatter, one 2^19-row F=8 level at a time.  Build: hipcc --offload-arch=gfx950 -O3 owner_probe.hip
//
// For a hashed level with R < 4096 the owner slab (row >> 12) of a corner does not depend on x
// (prime_x = 1 only touches bits 0..11), so a (sample, level) has 4 items, one per (dy, dz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <chrono>

constexpr uint32_t kP1 = 2654435761u, kP2 = 805459861u;
constexpr uint32_t kRowsLog2 = 19, kSlabLog2 = 12, kOwners = 1u << (kRowsLog2 - kSlabLog2);   // 128
constexpr uint32_t kF = 8;

struct Lvl { uint32_t R; };

__device__ __forceinline__ void cell(const float* __restrict__ pos, uint32_t i, uint32_t R, uint32_t (&pg)[3], float (&fr)[3])
{
    const float scale = (float)R - 2.0f;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float p = pos[(size_t)i * 3 + d] * scale + 0.5f;
        const float f = floorf(p);
        pg[d] = (uint32_t)f;
        fr[d] = p - f;
    }
}

// ---- reference: plain atomic scatter, lane = (sample, corner) x 8 features ----
__global__ void k_atomic_ref(const float* pos, const float* g, float* table, uint32_t N, uint32_t R)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = t >> 6, c = (t >> 3) & 7, f = t & 7;
    if (i >= N) return;
    uint32_t pg[3]; float fr[3];
    cell(pos, i, R, pg, fr);
    const uint32_t dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
    const float w = (dx ? fr[0] : 1 - fr[0]) * (dy ? fr[1] : 1 - fr[1]) * (dz ? fr[2] : 1 - fr[2]);
    const uint32_t row = ((pg[0] + dx) ^ ((pg[1] + dy) * kP1) ^ ((pg[2] + dz) * kP2)) & ((1u << kRowsLog2) - 1);
    unsafeAtomicAdd(table + (size_t)row * kF + f, w * g[(size_t)i * kF + f]);
}

// ---- pass 1: bin (sample, combo) items by owner; LDS histogram, one global reservation per (WG, bin) ----
__global__ __launch_bounds__(1024) void k_bin(const float* __restrict__ pos, uint32_t N, uint32_t R,
                                              uint32_t* __restrict__ bin_count, uint32_t* __restrict__ items,
                                              uint32_t cap, uint32_t* __restrict__ overflow)
{
    __shared__ uint32_t s_cnt[kOwners], s_base[kOwners];
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    if (threadIdx.x < kOwners) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t own[4], rank[4];
    const bool live = i < N;
    if (live) {
        uint32_t pg[3]; float fr[3];
        cell(pos, i, R, pg, fr);
        const uint32_t y0 = pg[1] * kP1, y1 = y0 + kP1, z0 = pg[2] * kP2, z1 = z0 + kP2;
        own[0] = ((y0 ^ z0) >> kSlabLog2) & (kOwners - 1);
        own[1] = ((y1 ^ z0) >> kSlabLog2) & (kOwners - 1);
        own[2] = ((y0 ^ z1) >> kSlabLog2) & (kOwners - 1);
        own[3] = ((y1 ^ z1) >> kSlabLog2) & (kOwners - 1);
#pragma unroll
        for (int k = 0; k < 4; k++) rank[k] = atomicAdd(&s_cnt[own[k]], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kOwners) s_base[threadIdx.x] = atomicAdd(&bin_count[threadIdx.x], s_cnt[threadIdx.x]);
    __syncthreads();
    if (live) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t at = s_base[own[k]] + rank[k];
            if (at < cap) items[(size_t)own[k] * cap + at] = (i << 2) | k;
            else atomicAdd(overflow, 1u);
        }
    }
}

// ---- pass 2: one workgroup per owner slab: 4096 rows x 8 floats in LDS ----
// MODE 0: AoS accumulators [row][f]; 1: loads only (no LDS atomics); 2: SoA [f][row];
// 3: SoA + software prefetch of the next item's data
template <int MODE>
__global__ __launch_bounds__(1024) void k_owner(const float* __restrict__ pos, const float* __restrict__ g,
                                                float* __restrict__ table, uint32_t R,
                                                const uint32_t* __restrict__ bin_count,
                                                const uint32_t* __restrict__ items, uint32_t cap)
{
    extern __shared__ float s_acc[];
    constexpr uint32_t S = 1u << kSlabLog2;
    const uint32_t owner = blockIdx.x;
    for (uint32_t k = threadIdx.x; k < S * kF; k += 1024) s_acc[k] = 0;
    __syncthreads();
    uint32_t n = bin_count[owner];
    n = n < cap ? n : cap;
    const uint32_t* my = items + (size_t)owner * cap;
    float sink = 0;
    const float scale = (float)R - 2.0f;

    uint32_t it = 0; float px = 0, py = 0, pz = 0; float4 g0 = {0,0,0,0}, g1 = {0,0,0,0};
    auto fetch = [&](uint32_t j) {
        if (j < n) {
            it = my[j];
            const uint32_t i = it >> 2;
            px = pos[(size_t)i * 3]; py = pos[(size_t)i * 3 + 1]; pz = pos[(size_t)i * 3 + 2];
            g0 = *reinterpret_cast<const float4*>(g + (size_t)i * kF);
            g1 = *reinterpret_cast<const float4*>(g + (size_t)i * kF + 4);
        }
    };
    if (MODE == 3) fetch(threadIdx.x);
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t j = base + threadIdx.x;
        if (MODE != 3) fetch(j);
        const uint32_t c_it = it; const float cx = px, cy = py, cz = pz; const float4 c0 = g0, c1 = g1;
        if (MODE == 3) fetch(j + 1024);
        if (j < n) {
            const uint32_t dy = c_it & 1, dz = (c_it >> 1) & 1;
            const float p0 = cx * scale + 0.5f, p1 = cy * scale + 0.5f, p2 = cz * scale + 0.5f;
            const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
            const float fr0 = p0 - f0, fr1 = p1 - f1, fr2 = p2 - f2;
            const float wyz = (dy ? fr1 : 1 - fr1) * (dz ? fr2 : 1 - fr2);
            const uint32_t A = (((uint32_t)f1 + dy) * kP1) ^ (((uint32_t)f2 + dz) * kP2);
            const uint32_t r0 = ((uint32_t)f0 ^ A) & (S - 1);
            const uint32_t r1 = (((uint32_t)f0 + 1) ^ A) & (S - 1);
            const float w0 = (1 - fr0) * wyz, w1 = fr0 * wyz;
            const float gv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int f = 0; f < 8; f++) {
                if (MODE == 1) { sink += w0 * gv[f] * (float)r0 + w1 * gv[f] * (float)r1; }
                else if (MODE == 0) { atomicAdd(&s_acc[r0 * kF + f], w0 * gv[f]); atomicAdd(&s_acc[r1 * kF + f], w1 * gv[f]); }
                else { atomicAdd(&s_acc[f * S + r0], w0 * gv[f]); atomicAdd(&s_acc[f * S + r1], w1 * gv[f]); }
            }
        }
    }
    if (MODE == 1) s_acc[threadIdx.x] = sink;
    __syncthreads();
    float* out = table + ((size_t)owner << kSlabLog2) * kF;
    if (MODE <= 1) {
        for (uint32_t k = threadIdx.x * 4; k < S * kF; k += 1024 * 4) {
            float4 t = *reinterpret_cast<float4*>(out + k);
            t.x += s_acc[k]; t.y += s_acc[k + 1]; t.z += s_acc[k + 2]; t.w += s_acc[k + 3];
            *reinterpret_cast<float4*>(out + k) = t;
        }
    } else {
        for (uint32_t k = threadIdx.x; k < S * kF; k += 1024) out[k] += s_acc[(k & 7) * S + (k >> 3)];
    }
}

// ---- pass 2, no LDS atomics: a lane claims a row by writing its id into tag[row]; after a barrier the
// winner does a plain LDS read-modify-write of the 8 floats, losers retry next round with the lanes
// that moved on to their next item ----
// workgroup barrier that orders LDS traffic only: outstanding global loads (the prefetch) stay in flight
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__global__ __launch_bounds__(1024) void k_owner_tag(const float* __restrict__ pos, const float* __restrict__ g,
                                                    float* __restrict__ table, uint32_t R,
                                                    const uint32_t* __restrict__ bin_count,
                                                    const uint32_t* __restrict__ items, uint32_t cap, uint32_t* __restrict__ rounds_out)
{
    extern __shared__ float s_acc[];   // [4096][8] floats, then uint16 tag[4096]
    constexpr uint32_t S = 1u << kSlabLog2;
    uint16_t* s_tag = reinterpret_cast<uint16_t*>(s_acc + S * kF);
    uint32_t* s_more = reinterpret_cast<uint32_t*>(s_tag + S);
    const uint32_t owner = blockIdx.x, tid = threadIdx.x;
    for (uint32_t k = tid; k < S * kF; k += 1024) s_acc[k] = 0;
    uint32_t n = bin_count[owner];
    n = n < cap ? n : cap;
    const uint32_t* my = items + (size_t)owner * cap;
    const float scale = (float)R - 2.0f;

    uint32_t j = tid;          // my next item
    uint32_t pending = 0;      // bit 0: row r0 still to add, bit 1: r1
    uint32_t r0 = 0, r1 = 0;
    float v0[8], v1[8];
    // next item's raw data, loaded one round ahead
    uint32_t nx_it = 0; float nx_p[3] = {0, 0, 0}; float4 nx_g0 = {0, 0, 0, 0}, nx_g1 = {0, 0, 0, 0};
    bool nx_valid = false;
    auto prefetch = [&]() {
        nx_valid = j < n;
        if (nx_valid) {
            nx_it = my[j];
            const uint32_t i = nx_it >> 2;
            nx_p[0] = pos[(size_t)i * 3]; nx_p[1] = pos[(size_t)i * 3 + 1]; nx_p[2] = pos[(size_t)i * 3 + 2];
            nx_g0 = *reinterpret_cast<const float4*>(g + (size_t)i * kF);
            nx_g1 = *reinterpret_cast<const float4*>(g + (size_t)i * kF + 4);
        }
        j += 1024;
    };
    prefetch();
    if (tid < 2) s_more[tid] = 0;
    __syncthreads();
    uint32_t rounds = 0;
    for (;;) {
        rounds++;
        if (pending == 0 && nx_valid) {
            const uint32_t dy = nx_it & 1, dz = (nx_it >> 1) & 1;
            const float p0 = nx_p[0] * scale + 0.5f, p1 = nx_p[1] * scale + 0.5f, p2 = nx_p[2] * scale + 0.5f;
            const float4 g0 = nx_g0, g1 = nx_g1;
            const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
            const float fr0 = p0 - f0, fr1 = p1 - f1, fr2 = p2 - f2;
            const float wyz = (dy ? fr1 : 1 - fr1) * (dz ? fr2 : 1 - fr2);
            const uint32_t A = (((uint32_t)f1 + dy) * kP1) ^ (((uint32_t)f2 + dz) * kP2);
            r0 = ((uint32_t)f0 ^ A) & (S - 1);
            r1 = (((uint32_t)f0 + 1) ^ A) & (S - 1);
            const float w0 = (1 - fr0) * wyz, w1 = fr0 * wyz;
            const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int f = 0; f < 8; f++) { v0[f] = w0 * gv[f]; v1[f] = w1 * gv[f]; }
            pending = 3;
            prefetch();
        }
        if (pending & 1) s_tag[r0] = (uint16_t)(tid * 2);
        if (pending & 2) s_tag[r1] = (uint16_t)(tid * 2 + 1);
        if (tid == 0) s_more[(rounds + 1) & 1] = 0;
        LDS_BARRIER();
        if ((pending & 1) && s_tag[r0] == (uint16_t)(tid * 2)) {
            float4* a = reinterpret_cast<float4*>(s_acc + r0 * kF);
            float4 x = a[0], y = a[1];
            x.x += v0[0]; x.y += v0[1]; x.z += v0[2]; x.w += v0[3];
            y.x += v0[4]; y.y += v0[5]; y.z += v0[6]; y.w += v0[7];
            a[0] = x; a[1] = y;
            pending &= ~1u;
        }
        if ((pending & 2) && s_tag[r1] == (uint16_t)(tid * 2 + 1)) {
            float4* a = reinterpret_cast<float4*>(s_acc + r1 * kF);
            float4 x = a[0], y = a[1];
            x.x += v1[0]; x.y += v1[1]; x.z += v1[2]; x.w += v1[3];
            y.x += v1[4]; y.y += v1[5]; y.z += v1[6]; y.w += v1[7];
            a[0] = x; a[1] = y;
            pending &= ~2u;
        }
        if (pending != 0 || nx_valid) s_more[rounds & 1] = 1;
        LDS_BARRIER();
        if (s_more[rounds & 1] == 0) break;
    }
    __syncthreads();
    if (tid == 0 && rounds_out) rounds_out[owner] = rounds;
    float* out = table + ((size_t)owner << kSlabLog2) * kF;
    for (uint32_t k = tid * 4; k < S * kF; k += 1024 * 4) {
        float4 t = *reinterpret_cast<float4*>(out + k);
        t.x += s_acc[k]; t.y += s_acc[k + 1]; t.z += s_acc[k + 2]; t.w += s_acc[k + 3];
        *reinterpret_cast<float4*>(out + k) = t;
    }
}

// =====================================================================================
// Design W: 2048 bins per level (row >> 8); one WAVE owns a 256-row sub-slab (8 KB of LDS), no barriers.
// item = sample << 4 | mask << 2 | combo; mask bit 0: add row r0 (x), bit 1: add row r1 (x+1)
// =====================================================================================
constexpr uint32_t kSubLog2 = 8, kBins = 1u << (kRowsLog2 - kSubLog2);   // 2048
constexpr uint32_t kSPT = 4;   // samples per thread in the bin pass

__device__ __forceinline__ void owners_of(const float* __restrict__ pos, uint32_t i, uint32_t R, uint32_t (&b0)[4], uint32_t (&b1)[4])
{
    uint32_t pg[3]; float fr[3];
    cell(pos, i, R, pg, fr);
    const uint32_t y0 = pg[1] * kP1, y1 = y0 + kP1, z0 = pg[2] * kP2, z1 = z0 + kP2;
    const uint32_t A[4] = {y0 ^ z0, y1 ^ z0, y0 ^ z1, y1 ^ z1};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        b0[k] = ((pg[0] ^ A[k]) >> kSubLog2) & (kBins - 1);
        b1[k] = (((pg[0] + 1) ^ A[k]) >> kSubLog2) & (kBins - 1);
    }
}

__global__ __launch_bounds__(1024) void k_bin2(const float* __restrict__ pos, uint32_t N, uint32_t R,
                                               uint32_t* __restrict__ bin_count, uint32_t* __restrict__ items,
                                               uint32_t cap, uint32_t* __restrict__ overflow)
{
    __shared__ uint32_t s_cnt[kBins];
    const uint32_t base_i = blockIdx.x * 1024 * kSPT;
    for (uint32_t b = threadIdx.x; b < kBins; b += 1024) s_cnt[b] = 0;
    __syncthreads();
    for (uint32_t k = 0; k < kSPT; k++) {
        const uint32_t i = base_i + k * 1024 + threadIdx.x;
        if (i < N) {
            uint32_t b0[4], b1[4];
            owners_of(pos, i, R, b0, b1);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                atomicAdd(&s_cnt[b0[c]], 1u);
                if (b1[c] != b0[c]) atomicAdd(&s_cnt[b1[c]], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < kBins; b += 1024) {
        const uint32_t c = s_cnt[b];
        s_cnt[b] = c ? atomicAdd(&bin_count[b], c) : 0;
    }
    __syncthreads();
    for (uint32_t k = 0; k < kSPT; k++) {
        const uint32_t i = base_i + k * 1024 + threadIdx.x;
        if (i < N) {
            uint32_t b0[4], b1[4];
            owners_of(pos, i, R, b0, b1);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const bool same = b1[c] == b0[c];
                uint32_t at = atomicAdd(&s_cnt[b0[c]], 1u);
                if (at < cap) items[(size_t)b0[c] * cap + at] = (i << 4) | ((same ? 3u : 1u) << 2) | c;
                else atomicAdd(overflow, 1u);
                if (!same) {
                    at = atomicAdd(&s_cnt[b1[c]], 1u);
                    if (at < cap) items[(size_t)b1[c] * cap + at] = (i << 4) | (2u << 2) | c;
                    else atomicAdd(overflow, 1u);
                }
            }
        }
    }
}

// one wave per bin; LDS: 256 rows x 8 floats + 256 tag bytes
__global__ __launch_bounds__(64) void k_owner_wave(const float* __restrict__ pos, const float* __restrict__ g,
                                                   float* __restrict__ table, uint32_t R_,
                                                   const uint32_t* __restrict__ bin_count,
                                                   const uint32_t* __restrict__ items, uint32_t cap, const uint32_t* __restrict__ Rl)
{
    constexpr uint32_t S = 1u << kSubLog2;
    const uint32_t R = Rl ? Rl[blockIdx.y] : R_;
    table += (size_t)blockIdx.y * (((size_t)1 << kRowsLog2) * kF);
    bin_count += blockIdx.y * kBins;
    items += (size_t)blockIdx.y * kBins * cap;
    __shared__ float s_acc[S * kF];
    __shared__ uint8_t s_tag[S];
    const uint32_t bin = blockIdx.x, lane = threadIdx.x;
    for (uint32_t k = lane; k < S * kF; k += 64) s_acc[k] = 0;
    uint32_t n = bin_count[bin];
    n = n < cap ? n : cap;
    const uint32_t* my = items + (size_t)bin * cap;
    const float scale = (float)R - 2.0f;

    uint32_t j = lane;
    uint32_t nx_it = 0; float nx_p[3] = {0, 0, 0}; float4 nx_g0 = {0, 0, 0, 0}, nx_g1 = {0, 0, 0, 0};
    bool nx_valid = false;
    auto prefetch = [&]() {
        nx_valid = j < n;
        if (nx_valid) {
            nx_it = my[j];
            const uint32_t i = nx_it >> 4;
            nx_p[0] = pos[(size_t)i * 3]; nx_p[1] = pos[(size_t)i * 3 + 1]; nx_p[2] = pos[(size_t)i * 3 + 2];
            nx_g0 = *reinterpret_cast<const float4*>(g + (size_t)i * kF);
            nx_g1 = *reinterpret_cast<const float4*>(g + (size_t)i * kF + 4);
        }
        j += 64;
    };
    prefetch();
    __syncthreads();
    while (__ballot(nx_valid) != 0) {
        uint32_t pend = 0, r0 = 0, r1 = 0;
        float v0[8], v1[8];
        if (nx_valid) {
            const uint32_t dy = nx_it & 1, dz = (nx_it >> 1) & 1;
            pend = (nx_it >> 2) & 3;
            const float p0 = nx_p[0] * scale + 0.5f, p1 = nx_p[1] * scale + 0.5f, p2 = nx_p[2] * scale + 0.5f;
            const float4 g0 = nx_g0, g1 = nx_g1;
            const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
            const float fr0 = p0 - f0, fr1 = p1 - f1, fr2 = p2 - f2;
            const float wyz = (dy ? fr1 : 1 - fr1) * (dz ? fr2 : 1 - fr2);
            const uint32_t A = (((uint32_t)f1 + dy) * kP1) ^ (((uint32_t)f2 + dz) * kP2);
            r0 = ((uint32_t)f0 ^ A) & (S - 1);
            r1 = (((uint32_t)f0 + 1) ^ A) & (S - 1);
            const float w0 = (1 - fr0) * wyz, w1 = fr0 * wyz;
            const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int f = 0; f < 8; f++) { v0[f] = w0 * gv[f]; v1[f] = w1 * gv[f]; }
        }
        prefetch();
        while (__ballot(pend != 0) != 0) {
            if (pend & 1) s_tag[r0] = (uint8_t)(lane * 2);
            if (pend & 2) s_tag[r1] = (uint8_t)(lane * 2 + 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if ((pend & 1) && s_tag[r0] == (uint8_t)(lane * 2)) {
                float4* a = reinterpret_cast<float4*>(s_acc + r0 * kF);
                float4 x = a[0], y = a[1];
                x.x += v0[0]; x.y += v0[1]; x.z += v0[2]; x.w += v0[3];
                y.x += v0[4]; y.y += v0[5]; y.z += v0[6]; y.w += v0[7];
                a[0] = x; a[1] = y;
                pend &= ~1u;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if ((pend & 2) && s_tag[r1] == (uint8_t)(lane * 2 + 1)) {
                float4* a = reinterpret_cast<float4*>(s_acc + r1 * kF);
                float4 x = a[0], y = a[1];
                x.x += v1[0]; x.y += v1[1]; x.z += v1[2]; x.w += v1[3];
                y.x += v1[4]; y.y += v1[5]; y.z += v1[6]; y.w += v1[7];
                a[0] = x; a[1] = y;
                pend &= ~2u;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    float* out = table + ((size_t)bin << kSubLog2) * kF;
    for (uint32_t k = lane * 4; k < S * kF; k += 64 * 4) {
        float4 t = *reinterpret_cast<float4*>(out + k);
        t.x += s_acc[k]; t.y += s_acc[k + 1]; t.z += s_acc[k + 2]; t.w += s_acc[k + 3];
        *reinterpret_cast<float4*>(out + k) = t;
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv)
{
    const uint32_t N = 1u << 20, NL = 4;
    const uint32_t Rs[NL] = {563, 778, 1484, 2049};
    const int ray_like = argc > 1 ? atoi(argv[1]) : 0;
    std::vector<float> hpos((size_t)N * 3), hg((size_t)N * kF);
    srand(1);
    if (!ray_like) {
        for (auto& v : hpos) v = (float)rand() / (float)RAND_MAX * 0.999f;
    } else {   // marched rays: 100 samples per ray, step 1/600
        for (uint32_t r = 0; r < N / 128; r++) {
            float o[3], d[3], n = 0;
            for (int k = 0; k < 3; k++) { o[k] = 0.3f + 0.2f * rand() / RAND_MAX; d[k] = (float)rand() / RAND_MAX - 0.3f; n += d[k] * d[k]; }
            for (int k = 0; k < 3; k++) d[k] /= sqrtf(n);
            for (uint32_t s = 0; s < 128; s++)
                for (int k = 0; k < 3; k++) { float v = o[k] + d[k] * s / 600.0f; hpos[((size_t)r * 128 + s) * 3 + k] = fminf(fmaxf(v, 0.f), 0.999f); }
        }
    }
    for (auto& v : hg) v = (float)rand() / (float)RAND_MAX - 0.5f;
    float *pos, *g, *t_ref, *t_own; uint32_t *bin_count, *items, *overflow;
    const size_t tbytes = ((size_t)1 << kRowsLog2) * kF * 4;
    const uint32_t cap = (N * 4 / kOwners) * 3 / 2;
    CK(hipMalloc(&pos, hpos.size() * 4)); CK(hipMalloc(&g, hg.size() * 4));
    CK(hipMalloc(&t_ref, tbytes * NL)); CK(hipMalloc(&t_own, tbytes * NL));
    CK(hipMalloc(&bin_count, NL * kOwners * 4)); CK(hipMalloc(&items, (size_t)NL * kOwners * cap * 4)); CK(hipMalloc(&overflow, 4 + 4 * NL * kOwners));
    CK(hipMemcpy(pos, hpos.data(), hpos.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(t_ref, 0, tbytes * NL)); CK(hipMemset(t_own, 0, tbytes * NL)); CK(hipMemset(overflow, 0, 4));
    CK(hipFuncSetAttribute((const void*)k_owner_tag, hipFuncAttributeMaxDynamicSharedMemorySize, 137 * 1024));
    CK(hipFuncSetAttribute((const void*)k_owner<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute((const void*)k_owner<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute((const void*)k_owner<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute((const void*)k_owner<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipEvent_t e0, e1, e2, e3; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3);
    hipStream_t st[NL]; for (auto& s : st) hipStreamCreate(&s);
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(t_ref, 0, tbytes * NL)); CK(hipMemset(t_own, 0, tbytes * NL));
        CK(hipMemset(bin_count, 0, NL * kOwners * 4));
        CK(hipDeviceSynchronize());
        hipEventRecord(e0);
        for (uint32_t l = 0; l < NL; l++)
            hipLaunchKernelGGL(k_atomic_ref, dim3(N * 64 / 256), dim3(256), 0, 0, pos, g, t_ref + l * (tbytes / 4), N, Rs[l]);
        hipEventRecord(e1);
        for (uint32_t l = 0; l < NL; l++)
            hipLaunchKernelGGL(k_bin, dim3(N / 1024), dim3(1024), 0, 0, pos, N, Rs[l], bin_count + l * kOwners, items + (size_t)l * kOwners * cap, cap, overflow);
        hipEventRecord(e2);
        CK(hipDeviceSynchronize());
        float ms_a, ms_b; hipEventElapsedTime(&ms_a, e0, e1); hipEventElapsedTime(&ms_b, e1, e2);
        printf("rep %d (%s): atomic scatter %.3f ms/level | bin %.3f ms/level |", rep, ray_like ? "rays" : "uniform", ms_a / NL, ms_b / NL);
        hipEventRecord(e2);
        for (uint32_t l = 0; l < NL; l++)
            hipLaunchKernelGGL(k_owner_tag, dim3(kOwners), dim3(1024), 137 * 1024, 0, pos, g, t_own + l * (tbytes / 4), Rs[l], bin_count + l * kOwners, items + (size_t)l * kOwners * cap, cap, overflow + 1 + l * kOwners);
        hipEventRecord(e3); CK(hipDeviceSynchronize());
        { float ms; hipEventElapsedTime(&ms, e2, e3); printf(" owner_tag %.3f ms/level", ms / NL); }
        printf("\n");
    }
    {
        float* t_scr; CK(hipMalloc(&t_scr, tbytes)); CK(hipMemset(t_scr, 0, tbytes));
        for (int m = 0; m < 4; m++) {
            hipEventRecord(e2);
            for (uint32_t l = 0; l < NL; l++) {
                const uint32_t* bc = bin_count + l * kOwners; const uint32_t* itp = items + (size_t)l * kOwners * cap;
                if (m == 0) hipLaunchKernelGGL(k_owner<0>, dim3(kOwners), dim3(1024), 128 * 1024, 0, pos, g, t_scr, Rs[l], bc, itp, cap);
                if (m == 1) hipLaunchKernelGGL(k_owner<1>, dim3(kOwners), dim3(1024), 128 * 1024, 0, pos, g, t_scr, Rs[l], bc, itp, cap);
                if (m == 2) hipLaunchKernelGGL(k_owner<2>, dim3(kOwners), dim3(1024), 128 * 1024, 0, pos, g, t_scr, Rs[l], bc, itp, cap);
                if (m == 3) hipLaunchKernelGGL(k_owner<3>, dim3(kOwners), dim3(1024), 128 * 1024, 0, pos, g, t_scr, Rs[l], bc, itp, cap);
            }
            hipEventRecord(e3); CK(hipDeviceSynchronize()); float ms; hipEventElapsedTime(&ms, e2, e3);
            printf("owner mode %d: %.3f ms/level (128 WGs per launch, sequential launches)\n", m, ms / NL);
        }
    }
    {
        const uint32_t cap2 = (N * 4 / kBins) * 2;
        uint32_t* dRs; CK(hipMalloc(&dRs, NL * 4)); CK(hipMemcpy(dRs, Rs, NL * 4, hipMemcpyHostToDevice));
        uint32_t *bc2, *items2; CK(hipMalloc(&bc2, NL * kBins * 4)); CK(hipMalloc(&items2, (size_t)NL * kBins * cap2 * 4));
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemset(t_own, 0, tbytes * NL)); CK(hipMemset(bc2, 0, NL * kBins * 4)); CK(hipMemset(overflow, 0, 4));
            CK(hipDeviceSynchronize());
            hipEventRecord(e0);
            for (uint32_t l = 0; l < NL; l++)
                hipLaunchKernelGGL(k_bin2, dim3((N + 1024 * kSPT - 1) / (1024 * kSPT)), dim3(1024), 0, 0, pos, N, Rs[l], bc2 + l * kBins, items2 + (size_t)l * kBins * cap2, cap2, overflow);
            hipEventRecord(e1);
            if (rep == 0) {
                for (uint32_t l = 0; l < NL; l++)
                    hipLaunchKernelGGL(k_owner_wave, dim3(kBins), dim3(64), 0, 0, pos, g, t_own + l * (tbytes / 4), Rs[l], bc2 + l * kBins, items2 + (size_t)l * kBins * cap2, cap2, (const uint32_t*)nullptr);
            } else {
                hipLaunchKernelGGL(k_owner_wave, dim3(kBins, NL), dim3(64), 0, 0, pos, g, t_own, 0, bc2, items2, cap2, (const uint32_t*)dRs);
            }
            hipEventRecord(e2); CK(hipDeviceSynchronize());
            float ms_a, ms_b; hipEventElapsedTime(&ms_a, e0, e1); hipEventElapsedTime(&ms_b, e1, e2);
            printf("design W rep %d: bin2 %.3f ms/level | owner_wave %.3f ms/level\n", rep, ms_a / NL, ms_b / NL);
        }
        std::vector<uint32_t> bc(NL * kBins); CK(hipMemcpy(bc.data(), bc2, NL * kBins * 4, hipMemcpyDeviceToHost));
        uint32_t mx = 0; uint64_t tot = 0; for (auto c : bc) { mx = c > mx ? c : mx; tot += c; }
        printf("design W: items/level %.0f, max bin %u, cap %u\n", (double)tot / NL, mx, cap2);
    }
    std::vector<float> a(tbytes / 4 * NL), b(tbytes / 4 * NL); uint32_t ov; std::vector<uint32_t> bc(NL * kOwners);
    CK(hipMemcpy(a.data(), t_ref, tbytes * NL, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), t_own, tbytes * NL, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&ov, overflow, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(bc.data(), bin_count, NL * kOwners * 4, hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0; for (size_t k = 0; k < a.size(); k++) { maxd = fmax(maxd, fabs((double)a[k] - b[k])); maxv = fmax(maxv, fabs((double)a[k])); }
    uint32_t mx = 0; for (auto c : bc) mx = c > mx ? c : mx;
    { std::vector<uint32_t> rr(NL * kOwners); CK(hipMemcpy(rr.data(), overflow + 1, 4 * NL * kOwners, hipMemcpyDeviceToHost)); uint32_t mr = 0; double ar = 0; for (auto c : rr) { mr = c > mr ? c : mr; ar += c; } printf("rounds avg %.1f max %u\n", ar / rr.size(), mr); }
    printf("max |diff| %.3g (max |ref| %.3g), overflow %u, max bin %u / cap %u (avg %u)\n", maxd, maxv, ov, mx, cap, N * 4 / kOwners);
    return 0;
}
