// fetch_calib.hip — what do rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of the
// encoder backward?  Kernels with KNOWN byte counts, one per access class:
//
//   calib_stream_read16    16 B / lane coalesced streaming read of 2 GiB       (the guide's calibrated case: reports 1/2)
//   calib_stream_write16   16 B / lane coalesced streaming write of 2 GiB
//   calib_gather32<B>      random 32-byte rows (two lanes x 16 B, as k_bwd_owner fetches a gradient row) from a table of
//                          B = 16 MiB (L2 / Infinity-Cache resident), 192 MiB (Infinity Cache), 2 GiB (HBM: every gather
//                          misses, so bytes per gather = the request size the fabric actually moves)
//   calib_scatter16        scattered 16-byte stores (as k_bwd_bin's item stores) into 1 GiB
//   calib_rmw32            read-modify-write of random 32-byte rows in a 192 MiB table (what an atomic-free scatter does)
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/fetch_calib tools/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/fetch -- tools/fetch_calib      (and WRITE_SIZE, and the raw
//   TCC_EA0_RDREQ / _32B / WRREQ / _64B request counters, each in its own pass)  -> tools/summarise_calib.py
//
// The program prints one JSON line per kernel: known algorithmic bytes, HIP-event time, GB/s.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ void __launch_bounds__(256) calib_stream_read16(const float4* __restrict__ src, uint64_t n16, float* __restrict__ sink)
{
    float acc = 0.f;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123456.789f) sink[0] = acc;      // never true: keeps the loads alive
}

__global__ void __launch_bounds__(256) calib_stream_write16(float4* __restrict__ dst, uint64_t n16)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x)
        dst[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}

// lane pair (2k, 2k+1) fetches the two 16-byte halves of one random 32-byte row; rows = power of two
template <int TAG>
__global__ void __launch_bounds__(256) calib_gather32(const float4* __restrict__ table, uint32_t row_mask, uint64_t n_rows,
                                                      float* __restrict__ sink)
{
    float acc = 0.f;
    const uint64_t lanes = n_rows * 2;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < lanes; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t row = mix((uint32_t)(i >> 1) * 2654435761u + TAG) & row_mask;
        const float4 v = table[(uint64_t)row * 2 + (i & 1)];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123456.789f) sink[0] = acc;
}

__global__ void __launch_bounds__(256) calib_scatter16(float4* __restrict__ dst, uint32_t slot_mask, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t slot = mix((uint32_t)i * 2654435761u + 17u) & slot_mask;
        dst[slot] = make_float4((float)i, 0.f, 1.f, 2.f);
    }
}

__global__ void __launch_bounds__(256) calib_rmw32(float4* __restrict__ table, uint32_t row_mask, uint64_t n_rows)
{
    const uint64_t lanes = n_rows * 2;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < lanes; i += (uint64_t)gridDim.x * blockDim.x) {
        // a permutation of the rows (odd multiplier): every row is touched exactly once, no two lanes race
        const uint32_t row = ((uint32_t)(i >> 1) * 2654435761u) & row_mask;
        float4* p = table + (uint64_t)row * 2 + (i & 1);
        float4 v = *p;
        v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
        *p = v;
    }
}

template <class F>
static void timed(const char* name, double bytes, int reps, F launch)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch();                                  // warm-up (also under the profiler: same kernel name, same bytes)
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a, 0));
    for (int r = 0; r < reps; ++r) launch();
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    printf("{\"kernel\": \"%s\", \"known_bytes\": %.0f, \"ms\": %.4f, \"GBps\": %.1f, \"launches\": %d}\n", name, bytes, ms,
           bytes / (ms * 1e6), reps + 1);
    fflush(stdout);
}

int main()
{
    const uint64_t big = 2ull << 30;           // 2 GiB: 8x the Infinity Cache
    float4 *buf = nullptr;
    float* sink = nullptr;
    CHECK(hipMalloc(&buf, big));
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipMemset(buf, 0, big));
    const int grid = 256 * 16, reps = 4;
    timed("calib_stream_read16", (double)big, reps, [&] { calib_stream_read16<<<grid, 256>>>(buf, big / 16, sink); });
    timed("calib_stream_write16", (double)big, reps, [&] { calib_stream_write16<<<grid, 256>>>(buf, big / 16); });
    const uint64_t n_rows = 1ull << 25;        // 32 Mi gathers = 1 GiB of rows per launch
    // table sizes in 32-byte rows: 16 MiB = 2^19, 256 MiB = 2^23 (the 16L table is 187 MiB; power of two for the mask), 2 GiB = 2^26
    timed("calib_gather32<0>/16MiB", (double)n_rows * 32, reps, [&] { calib_gather32<0><<<grid, 256>>>(buf, (1u << 19) - 1, n_rows, sink); });
    timed("calib_gather32<1>/256MiB", (double)n_rows * 32, reps, [&] { calib_gather32<1><<<grid, 256>>>(buf, (1u << 23) - 1, n_rows, sink); });
    timed("calib_gather32<2>/2GiB", (double)n_rows * 32, reps, [&] { calib_gather32<2><<<grid, 256>>>(buf, (1u << 26) - 1, n_rows, sink); });
    const uint64_t n_st = 1ull << 25;          // 32 Mi scattered 16-byte stores into 1 GiB (2^26 slots)
    timed("calib_scatter16/1GiB", (double)n_st * 16, reps, [&] { calib_scatter16<<<grid, 256>>>(buf, (1u << 26) - 1, n_st); });
    const uint64_t rmw_rows = 1ull << 23;      // every 32-byte row of a 256 MiB table once: 256 MiB read + 256 MiB written
    timed("calib_rmw32/256MiB", (double)rmw_rows * 64, reps, [&] { calib_rmw32<<<grid, 256>>>(buf, (1u << 23) - 1, rmw_rows); });
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
