// cumask_probe.hip — can two kernels that lean on different units be given disjoint CU sets on MI355X?
//
//  1. hipExtStreamCreateWithCUMask: which (XCC, SE, CU) do the bits of the mask select?  A census kernel records
//     HW_REG_XCC_ID and HW_REG_HW_ID of every block.
//  2. how do the two kinds of work in the encoder backward scale with the number of CUs they get:
//       * fabric-request-bound: random 32-byte-row gathers from 256 MiB (as k_bwd_owner's gradient rows)
//       * issue-bound: a dependent FMA chain (as k_grid_encode_bwd_merge)
//     on streams masked to 256 / 192 / 160 / 128 / 96 / 64 CUs, spread evenly over the XCCs.
//  3. both at once on complementary masks against both on unmasked streams.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/cumask_probe tools/cumask_probe.hip && tools/cumask_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <set>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ void census(uint32_t* out, int spin)
{
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;             // stay resident for a while
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
    if (a == 12345.f) out[0] = 0;
}

__global__ void __launch_bounds__(256) gather32(const float4* __restrict__ table, uint32_t row_mask, uint64_t n_rows, float* sink)
{
    float acc = 0.f;
    const uint64_t lanes = n_rows * 2;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < lanes; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t row = mix((uint32_t)(i >> 1) * 2654435761u) & row_mask;
        const float4 v = table[(uint64_t)row * 2 + (i & 1)];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123456.789f) sink[0] = acc;
}

__global__ void __launch_bounds__(256) fma_chain(float* sink, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = a + 1.f, e = a + 2.f, f = a + 3.f;
    for (int i = 0; i < iters; ++i) {
        a = a * b + c; d = d * b + c; e = e * b + c; f = f * b + c;
    }
    if (a + d + e + f == 12345.f) sink[0] = a;
}

static hipStream_t masked_stream(const std::vector<uint32_t>& mask)
{
    hipStream_t s;
    CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    return s;
}

// `per_group` of every 32 consecutive mask bits, starting at bit `first`
static std::vector<uint32_t> mask_even(int n_cu, int first, int per_group)
{
    std::vector<uint32_t> m((n_cu + 31) / 32, 0u);
    for (int i = 0; i < n_cu; ++i) {
        const int k = i % 32;
        const bool on = first <= k && k < first + per_group;
        if (on) m[i / 32] |= 1u << k;
    }
    return m;
}

static float time_ms(hipStream_t s, int reps, const std::function<void()>& launch)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch();
    CHECK(hipStreamSynchronize(s));
    CHECK(hipEventRecord(a, s));
    for (int r = 0; r < reps; ++r) launch();
    CHECK(hipEventRecord(b, s));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("{\"multiProcessorCount\": %d}\n", n_cu);
    uint32_t* d_out;
    const int blocks = 8192;
    CHECK(hipMalloc(&d_out, blocks * 8));
    std::vector<uint32_t> h(blocks * 2);
    // ---- 1. census: which CUs does a mask select?
    struct Case { const char* name; std::vector<uint32_t> mask; };
    std::vector<Case> cases;
    cases.push_back({"bits 0..31 (first word)", [&] { std::vector<uint32_t> m(n_cu / 32, 0u); m[0] = ~0u; return m; }()});
    cases.push_back({"bits 0..7", [&] { std::vector<uint32_t> m(n_cu / 32, 0u); m[0] = 0xFFu; return m; }()});
    cases.push_back({"every 2nd bit", [&] { return std::vector<uint32_t>(n_cu / 32, 0x55555555u); }()});
    cases.push_back({"low 16 bits of each word", mask_even(n_cu, 0, 16)});
    for (auto& c : cases) {
        hipStream_t s = masked_stream(c.mask);
        CHECK(hipMemsetAsync(d_out, 0xFF, blocks * 8, s));
        census<<<blocks, 64, 0, s>>>(d_out, 20000);
        CHECK(hipStreamSynchronize(s));
        CHECK(hipMemcpy(h.data(), d_out, blocks * 8, hipMemcpyDeviceToHost));
        std::map<uint32_t, std::set<uint32_t>> per_xcc;
        for (int b = 0; b < blocks; ++b) per_xcc[h[2 * b] & 0xF].insert((h[2 * b + 1] >> 8) & 0xFF);   // cu_id | sh_id | se_id
        printf("{\"mask\": \"%s\", \"xccs\": %zu, \"distinct_cu_per_xcc\": [", c.name, per_xcc.size());
        bool first = true;
        for (auto& kv : per_xcc) { printf("%s[%u, %zu]", first ? "" : ", ", kv.first, kv.second.size()); first = false; }
        printf("]}\n");
        CHECK(hipStreamDestroy(s));
    }
    // ---- 2. scaling with the number of CUs
    float4* table;
    float* sink;
    CHECK(hipMalloc(&table, 256ull << 20));
    CHECK(hipMemset(table, 0, 256ull << 20));
    CHECK(hipMalloc(&sink, 256));
    const uint64_t n_rows = 1ull << 24;
    for (int per32 : {32, 24, 20, 16, 12, 8}) {
        hipStream_t s = masked_stream(mask_even(n_cu, 0, per32));
        const float tg = time_ms(s, 4, [&] { gather32<<<256 * 16, 256, 0, s>>>(table, (1u << 23) - 1, n_rows, sink); });
        const float tf = time_ms(s, 4, [&] { fma_chain<<<256 * 8, 256, 0, s>>>(sink, 20000); });
        printf("{\"cus\": %d, \"gather_ms\": %.4f, \"gather_G_per_s\": %.1f, \"fma_ms\": %.4f}\n", n_cu * per32 / 32, tg,
               n_rows / (tg * 1e6), tf);
        CHECK(hipStreamDestroy(s));
    }
    // ---- 3. both at once: complementary masks vs no masks
    for (int per32 : {0, 12, 16, 20}) {     // CUs (of every 32) for the FMA kernel; 0 = two unmasked streams
        hipStream_t sa, sb;
        if (per32 == 0) { CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); }
        else { sa = masked_stream(mask_even(n_cu, 0, per32)); sb = masked_stream(mask_even(n_cu, per32, 32 - per32)); }
        hipEvent_t a, b0, b1;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b0)); CHECK(hipEventCreate(&b1));
        auto both = [&] {
            fma_chain<<<256 * 8, 256, 0, sa>>>(sink, 8000);
            gather32<<<256 * 16, 256, 0, sb>>>(table, (1u << 23) - 1, n_rows, sink + 1);
        };
        both();
        CHECK(hipDeviceSynchronize());
        const int reps = 4;
        CHECK(hipEventRecord(a, sa));
        CHECK(hipStreamWaitEvent(sb, a, 0));
        for (int r = 0; r < reps; ++r) both();
        CHECK(hipEventRecord(b0, sa)); CHECK(hipEventRecord(b1, sb));
        CHECK(hipEventSynchronize(b0)); CHECK(hipEventSynchronize(b1));
        float m0, m1;
        CHECK(hipEventElapsedTime(&m0, a, b0)); CHECK(hipEventElapsedTime(&m1, a, b1));
        printf("{\"fma_cus_per_32\": %d, \"fma_stream_ms\": %.4f, \"gather_stream_ms\": %.4f, \"both_done_ms\": %.4f}\n", per32, m0 / reps,
               m1 / reps, std::max(m0, m1) / reps);
        CHECK(hipStreamDestroy(sa)); CHECK(hipStreamDestroy(sb));
    }
    return 0;
}
