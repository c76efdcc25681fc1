// grid_encode_overlap.hip — the backward of one encoder call scheduled over three HIP streams, behind the C ABI.
//
// The coarse levels (run-merging atomic kernel) and the finest levels (bin + owner passes, grid_encode_binned.hip)
// write disjoint table rows and lean on different units (memory-side atomics vs. HBM reads / writes), so they
// overlap: measured 1.12 -> 1.07 ms per 2^20 samples (docs/engineering_log.md §4.2b).  Until ABI v20 the fork / join lived in the
// Python mirror; a C or C++ integrator calling cnc_grid_encode_backward_binned got the serial 1.12 ms.  Here the
// streams and events belong to a plan object the caller creates once (no globals in the library), and one call
// does:   fork event on `stream`  ->  finest levels in one or two groups on the plan's side streams
//                                 ->  coarse levels on `stream`  ->  `stream` waits for the groups.
// Everything the call touches is ordered on `stream` again when it returns.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <new>

#include "cnc_hip.h"

struct cnc_backward_plan {
    hipStream_t side[2];
    hipEvent_t  fork;
    hipEvent_t  join[2];
    int         device;
};

namespace {
constexpr uint32_t kMinOverlapPoints = 1u << 16;   // below that the events cost more than the overlap returns
constexpr uint64_t kAlign = 256;

inline uint64_t round_up(uint64_t v) { return (v + kAlign - 1) / kAlign * kAlign; }

// the finest levels as one group, or two halves when there are four or more (the bin pass of one next to the owner
// pass of the other: scattered stores vs. gathers — 1.122 -> 1.083 ms in the bench)
inline int split_groups(uint32_t n_binned, uint32_t first[2], uint32_t count[2])
{
    if (n_binned >= 4) {
        count[0] = n_binned / 2; count[1] = n_binned - count[0];
        first[0] = 0; first[1] = count[0];
        return 2;
    }
    first[0] = 0; count[0] = n_binned;
    return 1;
}
}  // namespace

extern "C" int cnc_backward_plan_create(cnc_backward_plan** out)
{
    if (!out) return CNC_ERR_INVALID_VALUE;
    cnc_backward_plan* p = new (std::nothrow) cnc_backward_plan();
    if (!p) return CNC_ERR_LAUNCH;
    bool ok = hipGetDevice(&p->device) == hipSuccess;
    // The side streams (the finest levels: the longer half, request-bound) run at the LOWEST stream priority: the coarse
    // kernel on the caller's stream is dispatched first and the bin / owner blocks fill in around it — 0.983 -> 0.972 ms
    // per 2^20 samples (three alternating runs; above the caller's priority: 0.982).  CNC_BWD_SIDE_PRIORITY overrides.
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    const char* pr = getenv("CNC_BWD_SIDE_PRIORITY");
    const int   prio = pr ? atoi(pr) : least;
    for (int i = 0; i < 2 && ok; ++i) {
        ok = hipStreamCreateWithPriority(&p->side[i], hipStreamNonBlocking, prio) == hipSuccess
             && hipEventCreateWithFlags(&p->join[i], hipEventDisableTiming) == hipSuccess;
    }
    ok = ok && hipEventCreateWithFlags(&p->fork, hipEventDisableTiming) == hipSuccess;
    if (!ok) { delete p; return CNC_ERR_LAUNCH; }     // partially created handles are leaked only on a broken runtime
    *out = p;
    return CNC_OK;
}

extern "C" int cnc_backward_plan_destroy(cnc_backward_plan* p)
{
    if (!p) return CNC_OK;
    for (int i = 0; i < 2; ++i) {
        (void)hipStreamSynchronize(p->side[i]);
        (void)hipStreamDestroy(p->side[i]);
        (void)hipEventDestroy(p->join[i]);
    }
    (void)hipEventDestroy(p->fork);
    delete p;
    return CNC_OK;
}

extern "C" uint64_t cnc_grid_encode_backward_overlapped_workspace(uint32_t N, uint32_t n_binned, uint32_t level_rows)
{
    uint32_t first[2], count[2];
    const int groups = split_groups(n_binned, first, count);
    uint64_t total = 0;
    for (int g = 0; g < groups; ++g)
        total += round_up(cnc_grid_encode_backward_binned_workspace(N, count[g], level_rows));
    // the serial fallback (small N, no coarse levels) needs the whole set of bins in one piece
    const uint64_t serial = cnc_grid_encode_backward_binned_workspace(N, n_binned, level_rows);
    return total > serial ? total : serial;
}

extern "C" int cnc_grid_encode_backward_overlapped(cnc_backward_plan* plan, const float* grad, const float* inputs,
                                                   const float* embeddings, const int32_t* offsets,
                                                   const int32_t* resolutions, float* grad_embeddings,
                                                   uint32_t N, uint32_t D, uint32_t F, uint32_t L, uint32_t flags,
                                                   const uint32_t* ste_clip_count, uint32_t grad_ld, uint32_t grad_col,
                                                   uint32_t n_binned, uint32_t level_rows,
                                                   void* workspace, uint64_t workspace_bytes, void* stream)
{
    if (N == 0 || L == 0) return CNC_OK;
    if (n_binned > L) return CNC_ERR_INVALID_VALUE;
    const uint32_t coarse = L - n_binned;
    if (!plan || coarse == 0 || n_binned == 0 || N < kMinOverlapPoints)
        return cnc_grid_encode_backward_binned(grad, inputs, embeddings, offsets, resolutions, grad_embeddings, N, D, F, L,
                                               flags, ste_clip_count, grad_ld, grad_col, n_binned, level_rows, workspace,
                                               workspace_bytes, stream);
    if (!grad || !inputs || !embeddings || !offsets || !resolutions || !grad_embeddings || !workspace)
        return CNC_ERR_INVALID_VALUE;
    if ((uintptr_t)workspace % 16 != 0) return CNC_ERR_INVALID_VALUE;
    hipStream_t s = (hipStream_t)stream;
    uint32_t first[2], count[2];
    const int groups = split_groups(n_binned, first, count);
    // each group gets a share of the caller's scratch proportional to its level count (deeper bins when the caller
    // passes more than the minimum)
    if (hipEventRecord(plan->fork, s) != hipSuccess) return CNC_ERR_LAUNCH;
    int rc = CNC_OK;
    uint64_t used = 0;
    for (int g = 0; g < groups; ++g) {
        const uint32_t l0 = coarse + first[g];
        uint64_t share = g + 1 < groups ? workspace_bytes * count[g] / n_binned / kAlign * kAlign : workspace_bytes - used;
        char* ws = (char*)workspace + used;
        used += share;
        // level-major [L, N, F]: the group's levels start l0 * N * F floats in; point-major: l0 * F columns to the right
        const float*   g_grad = grad_ld == 0 ? grad + (uint64_t)l0 * N * F : grad;
        const uint32_t g_col = grad_ld == 0 ? grad_col : grad_col + l0 * F;
        if (hipStreamWaitEvent(plan->side[g], plan->fork, 0) != hipSuccess) return CNC_ERR_LAUNCH;
        const int rg = cnc_grid_encode_backward_binned(g_grad, inputs, embeddings, offsets + l0, resolutions + l0,
                                                       grad_embeddings, N, D, F, count[g], flags, ste_clip_count, grad_ld,
                                                       g_col, count[g], level_rows, ws, share, plan->side[g]);
        if (rg != CNC_OK && rc == CNC_OK) rc = rg;
        if (hipEventRecord(plan->join[g], plan->side[g]) != hipSuccess) return CNC_ERR_LAUNCH;
    }
    // the coarse levels fill in next to the (longer) bin + owner passes, on the caller's stream
    const int rc0 = cnc_grid_encode_backward(grad, inputs, embeddings, offsets, resolutions, grad_embeddings, N, D, F, coarse,
                                             0, nullptr, nullptr, nullptr, nullptr, flags | CNC_FLAG_LEVELS_FINEST_FIRST,
                                             ste_clip_count, nullptr, nullptr, nullptr, grad_ld, grad_col, stream);
    for (int g = 0; g < groups; ++g)
        if (hipStreamWaitEvent(s, plan->join[g], 0) != hipSuccess) return CNC_ERR_LAUNCH;
    return rc0 != CNC_OK ? rc0 : rc;
}
