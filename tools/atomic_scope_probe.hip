// Scratch probe: fp32 atomic-add throughput by memory scope (device vs workgroup = executes in the XCD's L2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int SCOPE, int NROW>
__global__ void k(float* __restrict__ t, uint32_t rows_mask, uint32_t seed, size_t copy_stride, int per_xcd)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = tid / 8, sub = tid % 8;
    uint32_t xcc = 0;
    if (per_xcd) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 7; }
    float* base = t + (size_t)xcc * copy_stride;
#pragma unroll
    for (int r = 0; r < NROW; r++) {
        const uint32_t row = hash32(grp * 31u + r + seed) & rows_mask;
        float* p = base + (size_t)row * 8 + sub;
        if (SCOPE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}

__global__ void k_sum(const float* t, size_t n, double* out) {
    double s = 0; for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += t[i];
    atomicAdd(out, s);
}

template <int SCOPE>
void run(float* t, uint32_t rows, size_t copy_stride, int per_xcd, const char* name, double* dsum)
{
    const int threads = 1 << 25; const int NROW = 8;
    (void)hipMemset(t, 0, copy_stride * 8 * sizeof(float)); (void)hipMemset(dsum, 0, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SCOPE, NROW>), dim3(threads / 256), dim3(256), 0, 0, t, rows - 1, 1u, copy_stride, per_xcd);
    (void)hipEventRecord(e0);
    const int it = 4;
    for (int i = 0; i < it; i++) hipLaunchKernelGGL((k<SCOPE, NROW>), dim3(threads / 256), dim3(256), 0, 0, t, rows - 1, 7u + i, copy_stride, per_xcd);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    hipLaunchKernelGGL(k_sum, dim3(1024), dim3(256), 0, 0, t, copy_stride * 8, dsum);
    double h; (void)hipMemcpy(&h, dsum, 8, hipMemcpyDeviceToHost);
    const double expect = (double)threads * NROW * (it + 1);
    printf("%-34s rows=2^%d per_xcd=%d : %.1f Gatom/s   sum %.0f expect %.0f %s\n", name, 31 - __builtin_clz(rows), per_xcd,
           (double)threads * NROW * it / (ms * 1e-3) / 1e9, h, expect, h == expect ? "OK" : "LOST UPDATES");
}

int main()
{
    const uint32_t rows = 1u << 19;               // one 16 MiB level table
    const size_t stride = (size_t)rows * 8;       // floats per copy
    float* t; (void)hipMalloc(&t, stride * 8 * sizeof(float));
    double* dsum; (void)hipMalloc(&dsum, 8);
    run<0>(t, rows, stride, 0, "agent scope, shared table", dsum);
    run<0>(t, rows, stride, 1, "agent scope, per-XCD copies", dsum);
    run<1>(t, rows, stride, 1, "workgroup scope, per-XCD copies", dsum);
    run<2>(t, rows, stride, 1, "wavefront scope, per-XCD copies", dsum);
    run<1>(t, rows, stride, 0, "workgroup scope, shared (UNSAFE)", dsum);
    // per-XCD PARTITIONS that fit the XCD's 4 MiB L2: 2^16 rows = 2 MiB, 2^15 rows = 1 MiB per XCD
    run<0>(t, 1u << 16, stride, 1, "agent scope, 2 MiB per-XCD partitions", dsum);
    run<1>(t, 1u << 16, stride, 1, "workgroup scope, 2 MiB per-XCD partitions", dsum);
    run<1>(t, 1u << 15, stride, 1, "workgroup scope, 1 MiB per-XCD partitions", dsum);
    run<1>(t, 1u << 17, stride, 1, "workgroup scope, 4 MiB per-XCD partitions", dsum);
    const uint32_t rows2 = 1u << 22;              // 128 MiB per copy: far beyond L2
    float* t2; (void)hipMalloc(&t2, (size_t)rows2 * 8 * 8 * sizeof(float));
    run<0>(t2, rows2, (size_t)rows2 * 8, 0, "agent scope, shared table", dsum);
    run<1>(t2, rows2, (size_t)rows2 * 8, 1, "workgroup scope, per-XCD copies", dsum);
    return 0;
}
