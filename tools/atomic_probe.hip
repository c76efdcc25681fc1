// Scratch probe: fp32 global atomic-add throughput on gfx950 for different lane->address layouts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// LPR lanes cooperate on one 8-float row; each lane adds 8/LPR consecutive floats. ROWS_PER_GROUP rows per lane-group.
// WIDE lanes cover WIDE/8 adjacent rows (aligned group of rows) with one dword each
template <int WIDE, int NROW>
__global__ void k_atomic_wide(float* __restrict__ t, uint32_t rows_mask, uint32_t seed, int misalign)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = tid / WIDE, sub = tid % WIDE;
#pragma unroll
    for (int r = 0; r < NROW; r++) {
        uint32_t row = (hash32(grp * 31u + r + seed) & rows_mask) & ~(uint32_t)(WIDE / 8 - 1);
        row += misalign;   // 1 => the row group straddles an aligned boundary
        unsafeAtomicAdd(t + (size_t)row * 8 + sub, 1.0f);
    }
}

template <int WIDE, int NROW>
double run_wide(float* t, uint32_t rows, int threads, int misalign)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_atomic_wide<WIDE, NROW>), dim3(threads / 256), dim3(256), 0, 0, t, rows / 2 - 1, i, misalign);
    (void)hipEventRecord(e0);
    const int it = 5;
    for (int i = 0; i < it; i++) hipLaunchKernelGGL((k_atomic_wide<WIDE, NROW>), dim3(threads / 256), dim3(256), 0, 0, t, rows / 2 - 1, 7 + i, misalign);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)threads * NROW * it / (ms * 1e-3) / 1e9;
}

template <int LPR, int NROW>
__global__ void k_atomic(float* __restrict__ t, uint32_t rows_mask, uint32_t seed, int run_len)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = tid / LPR, sub = tid % LPR;
    constexpr int PER = 8 / LPR;
#pragma unroll
    for (int r = 0; r < NROW; r++) {
        // run_len consecutive groups share the same row (contention / coalescing opportunity)
        const uint32_t row = hash32((grp / run_len) * 31u + r + seed) & rows_mask;
        float* p = t + (size_t)row * 8 + sub * PER;
#pragma unroll
        for (int k = 0; k < PER; k++) unsafeAtomicAdd(p + k, 1.0f);
    }
}

template <int LPR, int NROW>
double run(float* t, uint32_t rows, int groups, int run_len)
{
    const int threads = groups * LPR;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_atomic<LPR, NROW>), dim3(threads / 256), dim3(256), 0, 0, t, rows - 1, i, run_len);
    hipEventRecord(e0);
    const int it = 5;
    for (int i = 0; i < it; i++) hipLaunchKernelGGL((k_atomic<LPR, NROW>), dim3(threads / 256), dim3(256), 0, 0, t, rows - 1, 7 + i, run_len);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)groups * NROW * 8 * it / (ms * 1e-3) / 1e9;   // G float-atomics / s
}

int main()
{
    const uint32_t rows_big = 1u << 22;   // 128 MiB of 32-B rows
    float* t; hipMalloc(&t, (size_t)rows_big * 32); hipMemset(t, 0, (size_t)rows_big * 32);
    const int groups = 1 << 22;
    for (uint32_t rows : {1u << 19}) {
        for (int rl : {1, 4, 16}) {
            printf("rows=2^%d run_len=%2d  | LPR1 %.1f  LPR2 %.1f  LPR4 %.1f  LPR8 %.1f  Gatom/s\n", 31 - __builtin_clz(rows), rl,
                   run<1, 8>(t, rows, groups, rl), run<2, 8>(t, rows, groups, rl), run<4, 8>(t, rows, groups, rl), run<8, 8>(t, rows, groups, rl));
        }
    }
    for (int mis : {0, 1})
        printf("wide adjacent rows, misalign=%d | 8 lanes(1 row) %.1f  16 lanes(2 rows) %.1f  32 lanes(4 rows) %.1f  64 lanes(8 rows) %.1f Gatom/s\n", mis,
               run_wide<8, 8>(t, 1u << 19, 1 << 24, mis), run_wide<16, 8>(t, 1u << 19, 1 << 24, mis), run_wide<32, 8>(t, 1u << 19, 1 << 24, mis), run_wide<64, 8>(t, 1u << 19, 1 << 24, mis));
    return 0;
}
