// mix_probe.hip — do fp32 row atomics and 32-byte row gathers share ONE memory-side resource on MI355X?
//
// The encoder backward's overlapped call runs a kernel that ends in row atomics (coarse levels: 6.6 M
// per call) next to kernels whose traffic is random 32-byte gathers, item streams and slab read-modify-writes (finest
// levels: 31 M requests).  Alone, each class has its measured ceiling: 21 G row atomics/s (tools/atomic_probe.hip),
// ~50 G requests/s for gathers / streams (tools/fetch_calib.hip).  If the two classes used different units, the call
// could approach max(0.475, 0.645) ms; it takes 0.96.  This probe runs the two classes ALONE and CONCURRENTLY (two
// streams, grids sized so that each alone lasts about the same) and prints the rates: with one shared resource the
// normalised rates add up to ~1 (a'/a0 + b'/b0), with independent units to ~2.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/mix_probe tools/mix_probe.hip && tools/mix_probe
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

// 8 lanes per random 32-byte row, one dword each (what k_grid_encode_bwd_merge issues per distinct cell corner)
__global__ void __launch_bounds__(256) k_row_atomics(float* __restrict__ t, uint32_t row_mask, uint32_t reps, uint32_t seed)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, grp = tid >> 3, sub = tid & 7u;
    for (uint32_t r = 0; r < reps; r++) {
        const uint32_t row = mix(grp * 31u + r * 0x9e3779b9u + seed) & row_mask;
        unsafeAtomicAdd(t + (size_t)row * 8 + sub, 1.0f);
    }
}

// lane pair fetches the two 16-byte halves of one random 32-byte row (k_bwd_owner's gradient-row gather)
__global__ void __launch_bounds__(256) k_row_gathers(const float4* __restrict__ t, uint32_t row_mask, uint32_t reps, uint32_t seed,
                                                     float* __restrict__ sink)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (uint32_t r = 0; r < reps; r++) {
        const uint32_t row = mix((tid >> 1) * 2654435761u + r * 0x9e3779b9u + seed) & row_mask;
        const float4 v = t[(uint64_t)row * 2 + (tid & 1u)];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123456.789f) sink[0] = acc;
}

// 16 B / lane coalesced stream (the item lists and table slabs)
__global__ void __launch_bounds__(256) k_stream(const float4* __restrict__ src, uint64_t n16, uint32_t reps, float* __restrict__ sink)
{
    float acc = 0.f;
    for (uint32_t r = 0; r < reps; r++)
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
            const float4 v = src[i];
            acc += v.x + v.y + v.z + v.w;
        }
    if (acc == 123456.789f) sink[0] = acc;
}

struct Timed { float ms; };

int main()
{
    const uint32_t rows = 1u << 22;                       // 128 MiB of 32-byte rows, per table
    float *ta, *tg, *sink;
    CHECK(hipMalloc(&ta, (size_t)rows * 32)); CHECK(hipMemset(ta, 0, (size_t)rows * 32));
    CHECK(hipMalloc(&tg, (size_t)rows * 32)); CHECK(hipMemset(tg, 0, (size_t)rows * 32));
    CHECK(hipMalloc(&sink, 64));
    hipStream_t sa, sb;
    CHECK(hipStreamCreate(&sa)); CHECK(hipStreamCreate(&sb));
    hipEvent_t a0, a1, b0, b1;
    CHECK(hipEventCreate(&a0)); CHECK(hipEventCreate(&a1)); CHECK(hipEventCreate(&b0)); CHECK(hipEventCreate(&b1));
    for (uint32_t blocks : {256u * 4u, 256u * 8u}) {      // per kernel: half / all of the chip's wave slots
    auto run = [&](bool do_a, bool do_b, int mode_b, uint32_t reps_a, uint32_t reps_b, float* ms_a, float* ms_b) {
        for (int it = 0; it < 3; it++) {                  // the last iteration is the measurement
            if (do_a) {
                CHECK(hipEventRecord(a0, sa));
                hipLaunchKernelGGL(k_row_atomics, dim3(blocks), dim3(256), 0, sa, ta, rows - 1, reps_a, 17u * it);
                CHECK(hipEventRecord(a1, sa));
            }
            if (do_b) {
                CHECK(hipEventRecord(b0, sb));
                if (mode_b == 0)
                    hipLaunchKernelGGL(k_row_gathers, dim3(blocks), dim3(256), 0, sb, (const float4*)tg, rows - 1, reps_b, 29u * it, sink);
                else
                    hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, sb, (const float4*)tg, (uint64_t)rows * 2, reps_b, sink);
                CHECK(hipEventRecord(b1, sb));
            }
            CHECK(hipDeviceSynchronize());
        }
        *ms_a = *ms_b = 0;
        if (do_a) CHECK(hipEventElapsedTime(ms_a, a0, a1));
        if (do_b) CHECK(hipEventElapsedTime(ms_b, b0, b1));
    };
    const double threads = (double)blocks * 256;
    for (int mode_b = 0; mode_b < 2; mode_b++) {
        const char* name_b = mode_b == 0 ? "gather32" : "stream16";
        // size the two so that each alone lasts ~2 ms
        uint32_t reps_a = 64, reps_b = mode_b == 0 ? 128 : 8;
        float ma, mb, dummy;
        run(true, false, mode_b, reps_a, reps_b, &ma, &dummy);
        run(false, true, mode_b, reps_a, reps_b, &dummy, &mb);
        reps_a = (uint32_t)(reps_a * 2.0 / ma + 0.5); if (reps_a < 1) reps_a = 1;
        reps_b = (uint32_t)(reps_b * 2.0 / mb + 0.5); if (reps_b < 1) reps_b = 1;
        run(true, false, mode_b, reps_a, reps_b, &ma, &dummy);
        run(false, true, mode_b, reps_a, reps_b, &dummy, &mb);
        const double req_a = threads / 8.0 * reps_a;                                 // row atomics
        const double req_b = mode_b == 0 ? threads / 2.0 * reps_b                    // gathered rows (one request each)
                                         : (double)rows * 32.0 / 128.0 * reps_b;     // 128-byte stream requests
        const double a_alone = req_a / (ma * 1e-3) / 1e9, b_alone = req_b / (mb * 1e-3) / 1e9;
        float mac, mbc;
        run(true, true, mode_b, reps_a, reps_b, &mac, &mbc);
        const double a_both = req_a / (mac * 1e-3) / 1e9, b_both = req_b / (mbc * 1e-3) / 1e9;
        printf("{\"blocks_per_kernel\": %u, \"pair\": \"row_atomics + %s\", \"alone_ms\": [%.3f, %.3f], \"together_ms\": [%.3f, %.3f], "
               "\"G_requests_per_s_alone\": [%.2f, %.2f], \"G_requests_per_s_together\": [%.2f, %.2f], "
               "\"normalised_sum\": %.3f, \"wall_together_over_sum_alone\": %.3f}\n",
               blocks, name_b, ma, mb, mac, mbc, a_alone, b_alone, a_both, b_both, a_both / a_alone + b_both / b_alone,
               (mac > mbc ? mac : mbc) / (ma + mb));
    }
    }
    return 0;
}
