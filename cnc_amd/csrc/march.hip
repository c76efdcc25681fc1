// march.hip — ray/AABB slab test and occupancy-grid ray marching for gfx950.
//
// Stands in for nerfacc/cuda/csrc/grid.cu (ray_aabb_intersect_kernel :320-349,
// traverse_grids_kernel :68-318) and include/utils_grid.cuh (:11-149) behind include/cnc_hip.h.
//
// One lane marches one ray (the traversal is a serial DDA whose every step depends on the
// previous one).  What is shaped for CDNA4 here:
//   * 64-thread workgroups: a wave retires as soon as its own 64 rays finish, instead of a
//     512-thread block waiting for its slowest ray, and N_rays/64 workgroups keep 256 CUs busy
//     even for an 8192-ray evaluation chunk;
//   * the 2 MiB occupancy grid is read-only and L2-resident; rays arrive in image order so the
//     lanes of a wave walk neighbouring cells;
//   * per-ray outputs are written in (ray, t) order at chunk_starts[ray] exactly as the
//     reference does, so the host mirror can reuse the reference's boolean-mask post-processing
//     (nerfacc/estimators/occ_grid.py:188-189).
#include <stdlib.h>

#include "common.hpp"

namespace cnc {

__device__ __forceinline__ float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
__device__ __forceinline__ int   clampi(int f, int a, int b) { return max(a, min(f, b)); }
__device__ __forceinline__ float calc_dt(float t, float cone, float dmin, float dmax)
{
    return clampf(t * cone, dmin, dmax);
}

// utils_grid.cuh:11-56
__device__ __forceinline__ bool slab(const float* __restrict__ o, const float (&inv)[3],
                                     const float* __restrict__ bb, float near, float far,
                                     float& tmin_o, float& tmax_o)
{
    float tmin, tmax;
    if (inv[0] >= 0) { tmin = (bb[0] - o[0]) * inv[0]; tmax = (bb[3] - o[0]) * inv[0]; }
    else             { tmin = (bb[3] - o[0]) * inv[0]; tmax = (bb[0] - o[0]) * inv[0]; }
#pragma unroll
    for (int d = 1; d < 3; d++) {
        float a, b;
        if (inv[d] >= 0) { a = (bb[d] - o[d]) * inv[d];     b = (bb[3 + d] - o[d]) * inv[d]; }
        else             { a = (bb[3 + d] - o[d]) * inv[d]; b = (bb[d] - o[d]) * inv[d]; }
        if (tmin > b || a > tmax) return false;
        if (a > tmin) tmin = a;
        if (b < tmax) tmax = b;
    }
    if (tmax <= 0) return false;
    tmin_o = fmaxf(tmin, near);
    tmax_o = fminf(tmax, far);
    return true;
}

__global__ __launch_bounds__(256) void k_ray_aabb(const float* __restrict__ rays_o,
                                                  const float* __restrict__ rays_d,
                                                  const float* __restrict__ aabbs, int32_t n_rays,
                                                  int32_t n_aabbs, float near, float far,
                                                  float miss, float* __restrict__ t_mins,
                                                  float* __restrict__ t_maxs,
                                                  uint8_t* __restrict__ hits)
{
    const int32_t numel = n_rays * n_aabbs;
    for (int32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < numel; t += blockDim.x * gridDim.x) {
        const int32_t r = t / n_aabbs, a = t % n_aabbs;
        const float inv[3] = {1.0f / rays_d[r * 3], 1.0f / rays_d[r * 3 + 1], 1.0f / rays_d[r * 3 + 2]};
        float t0, t1;
        const bool hit = slab(rays_o + (size_t)r * 3, inv, aabbs + a * 6, near, far, t0, t1);
        t_mins[t] = hit ? t0 : miss;
        t_maxs[t] = hit ? t1 : miss;
        hits[t] = (uint8_t)hit;
    }
}

struct Seg {
    float*   vals;
    int64_t* chunk_starts;
    int64_t* chunk_cnts;
    int64_t* ray_indices;
    uint8_t* is_left;
    uint8_t* is_right;
    uint8_t* is_valid;
};

// Samples / interval edges a lane may hold in LDS before the wave writes them out together.
constexpr int kStage = 16;
constexpr int kStagePitch = kStage + 1;   // +1: the flush reads a lane's row with 16 lanes, conflict-free

// LDS staging of one wave (FILL pass only): [64][17] sample mids, [64][17] interval edges,
// [64][17] interval flag bytes (bit 0 = is_left, bit 1 = is_right).
struct Stage {
    float*   sm;
    float*   iv;
    uint8_t* ivf;
};

// The wave writes what its 64 rays staged: four source rays per step, 16 lanes each, so a ray's
// entries leave as one contiguous 64-byte (vals), 128-byte (ray_indices) and 16-byte (flags) piece
// instead of one transaction per lane and array.  `cnt` entries of lane `src` go to out[g .. g+cnt).
template <bool IV>
__device__ __forceinline__ void flush_stage(const Stage& st, const Seg& o, uint32_t cnt, int64_t g,
                                            int32_t ray)
{
    const uint32_t lane = threadIdx.x, sub = lane >> 4, e = lane & 15;
#pragma unroll 1
    for (uint32_t it = 0; it < 16; it++) {
        const uint32_t src = it * 4 + sub;
        const uint32_t c = (uint32_t)__shfl((int)cnt, (int)src);
        const int64_t  gs = __shfl(g, (int)src);
        const int32_t  r = __shfl(ray, (int)src);
        if (e < c) {
            const int64_t k = gs + e;
            if constexpr (IV) {
                const uint8_t f = st.ivf[src * kStagePitch + e];
                o.vals[k] = st.iv[src * kStagePitch + e];
                o.ray_indices[k] = r;
                o.is_left[k] = f & 1;
                o.is_right[k] = (f >> 1) & 1;
            } else {
                o.vals[k] = st.sm[src * kStagePitch + e];
                o.ray_indices[k] = r;
                o.is_valid[k] = 1;
            }
        }
    }
}

// MODE 2 staging: 32 (t_start, t_end) pairs per lane.

// Flush of the (t_start, t_end) rows: only rows that hold something are visited (a wave pauses whenever ONE of
// its rays has filled its row; the rays that are still far from full keep theirs), two rows per step, 32 lanes
// each: a row leaves as 128 contiguous bytes per float array and 256 for the int64 ray ids.
// What cnc_march_samples may emit per sample besides (t_start, t_end): the sample's position — the expression of the
// reference's rgb_sigma_fn (examples/utils.py:251-262: o + d (t0 + t1) / 2), optionally mapped to the unit cube of
// `aabb` as the field does first thing (ngp.py:518-519), same operations in the same order as k_sample_positions —
// the ray direction, and the ray id as int32.  The marching lane has o and d in registers; a separate pass would
// re-read (ray, t0, t1) for every sample (554 MB of fetches per 800x800 frame) to recompute them.
struct SampleExtras {
    float*       positions;      // [S, 3] or null
    float*       dirs;           // [S, 3] or null
    int32_t*     ray_indices32;  // [S] or null
    const float* aabb;           // 6 floats or null: positions normalised to the box
};

__device__ __forceinline__ float sample_coord(float o, float d, float t0, float t1, const float* __restrict__ aabb, int a)
{
    float p = o + (d * (t0 + t1)) / 2.0f;
    if (aabb) p = (p - aabb[a]) / (aabb[3 + a] - aabb[a]);
    return p;
}

template <int PROW>
__device__ __forceinline__ void flush_pairs(const Stage& st, float* __restrict__ t_starts,
                                            float* __restrict__ t_ends, int64_t* __restrict__ ray_indices,
                                            uint32_t cnt, int64_t g, int32_t ray, const SampleExtras& ex,
                                            const float (&o)[3], const float (&dir)[3])
{
    constexpr uint32_t G = 64 / PROW;                    // rows written per step, PROW lanes each
    constexpr int      kPP = PROW + 1;
    const uint32_t lane = threadIdx.x, grp = lane / PROW, e = lane % PROW;
    const bool     want_od = ex.positions != nullptr || ex.dirs != nullptr;
    uint64_t todo = __ballot(cnt > 0);
    while (todo) {
        uint64_t m = todo;
        for (uint32_t k = 0; k < grp; k++) m &= m - 1;   // group g takes the g-th pending row
        const bool     has = m != 0;
        const uint32_t src = has ? (uint32_t)__builtin_ctzll(m) : 0u;
        const uint32_t c = (uint32_t)__shfl((int)cnt, (int)src);
        const int64_t  gs = __shfl(g, (int)src);
        const int32_t  r = __shfl(ray, (int)src);
        float so[3] = {0, 0, 0}, sd[3] = {0, 0, 0};
        if (want_od) {
#pragma unroll
            for (int a = 0; a < 3; a++) { so[a] = __shfl(o[a], (int)src); sd[a] = __shfl(dir[a], (int)src); }
        }
        if (has && e < c) {
            const int64_t k = gs + e;
            const float t0 = st.sm[src * kPP + e], t1 = st.iv[src * kPP + e];
            t_starts[k] = t0;
            t_ends[k] = t1;
            if (ray_indices) ray_indices[k] = r;
            if (ex.ray_indices32) ex.ray_indices32[k] = r;
            if (ex.positions) {
#pragma unroll
                for (int a = 0; a < 3; a++) ex.positions[k * 3 + a] = sample_coord(so[a], sd[a], t0, t1, ex.aabb, a);
            }
            if (ex.dirs) {
#pragma unroll
                for (int a = 0; a < 3; a++) ex.dirs[k * 3 + a] = sd[a];
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < G; k++) todo &= todo - 1;
    }
}

// One lane marches one ray (a serial DDA).  FILL = false: count only (first pass).  FILL = true:
// the march is resumable — a lane stops when its LDS staging row is full, the wave flushes all 64
// rows with coalesced stores (flush_stage), and the lane continues exactly where it stopped
// (same registers, same cell), so values and order equal the one-sweep march of grid.cu:68-318.
// MODE 0: count only.  MODE 1: fill the reference's RaySegmentsSpec pair (intervals + samples).
// CONE0: cone_angle == 0 with step_size > 0 (every CNC configuration): dt = clamp(t * 0, step, 1e10) = step for every finite t,
// so the three instructions per marching step that recompute it are dropped (same values).
// MODE 3 (extension, small batches of cnc_march_samples): MODE 2's outputs stored by each lane as it goes — no LDS, no
// pause / resume, full waves: for a training batch (~37 k rays, 260 k samples) the pass is a serial march per ray and
// the staging only adds latency (0.66 -> the count pass's 0.35 ms); the big frames keep the coalesced flushes.
// MODE 2 (extension, cnc_march_samples): fill (t_start, t_end, ray) per sample and nothing else — what the
// renderer consumes (occ_grid.py:176-178 derives exactly these from the edge flags) — 16 instead of 27 bytes
// per sample, 32-entry staging rows and a flush that only touches rows that have something to write.
template <int MODE, int PROW = 32, bool CONE0 = false>
__global__ __launch_bounds__(64) void k_traverse(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const uint8_t* __restrict__ rays_mask, int32_t n_rays, const uint8_t* __restrict__ binaries,
    int32_t n_grids, int32_t resx, int32_t resy, int32_t resz, const float* __restrict__ aabbs,
    const uint8_t* __restrict__ hits, const float* __restrict__ t_sorted,
    const int64_t* __restrict__ t_indices, const float* __restrict__ near_planes,
    const float* __restrict__ far_planes, float step_size, float cone_angle, int32_t limit,
    Seg iv, Seg sm, float* __restrict__ terminate_planes, int32_t rpb, uint32_t* __restrict__ rstate = nullptr,
    SampleExtras ex = SampleExtras{nullptr, nullptr, nullptr, nullptr}, const uint32_t* __restrict__ coarse = nullptr,
    uint32_t coarse_words = 0, uint32_t coarse_lds_off = 0)
{
    extern __shared__ float s_dyn[];
    // Coarse occupancy (cnc_occupancy_coarse_bits: one bit per block of 4 x 4 x 4 cells = "any cell of the block is set")
    // in LDS: a DDA step through empty space asks it first and goes to the cell's byte in memory only where the block
    // holds something.  The march is one dependent load per step and nothing else to hide it behind; most steps of a ray
    // are through empty blocks.  Same decisions, same values: the bit only ever answers for cells that are 0.
    uint32_t* const s_coarse = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(s_dyn) + coarse_lds_off);
    if (coarse != nullptr) {
        for (uint32_t k = threadIdx.x; k < coarse_words; k += 64) s_coarse[k] = coarse[k];
        __syncthreads();
    }
    constexpr bool DIRECT = MODE == 3;                  // fill without staging: every lane stores its own samples
    constexpr bool FILL = MODE != 0 && !DIRECT;         // "FILL" below = the staged (resumable) fill passes
    constexpr bool PAIRS = MODE == 2;
    constexpr int  kRow = PAIRS ? PROW : kStage;        // staged entries per lane
    constexpr int  kPP = PROW + 1;
    const float eps = 1e-6f;
    const int   res[3] = {resx, resy, resz};
    const bool  has_iv = !PAIRS && iv.chunk_cnts != nullptr, has_sm = sm.chunk_cnts != nullptr;
    const uint32_t lane = threadIdx.x;
    // rpb rays per 64-lane block: 64, or 16 for small batches (a training batch of 27 k rays is 420 full waves on
    // 256 CUs, each a long serial march: with a quarter of the lanes the batch spreads over four times the waves)
    const int32_t  tid = blockIdx.x * rpb + lane;

    Stage st{};
    if constexpr (PAIRS) {
        st.sm = s_dyn;                          // t_start of each staged sample
        st.iv = s_dyn + 64 * kPP;        // t_end
    } else if constexpr (FILL) {
        st.sm = s_dyn;
        st.iv = s_dyn + 64 * kStagePitch;
        st.ivf = (uint8_t*)(s_dyn + 2 * 64 * kStagePitch);
    }

    bool live = (int32_t)lane < rpb && tid < n_rays;
    if (live && rays_mask != nullptr && !rays_mask[tid]) live = false;
    if constexpr (FILL) {
        if (live && has_iv && iv.chunk_cnts[tid] == 0) live = false;
        if (live && has_sm && sm.chunk_cnts[tid] == 0) live = false;
    }
    if constexpr (DIRECT) {
        if (live && sm.chunk_cnts[tid] == 0) live = false;
    }
    int64_t cs_iv = 0, cs_sm = 0;
    if ((FILL || DIRECT) && live) {
        if (has_iv) cs_iv = iv.chunk_starts[tid];
        if (has_sm) cs_sm = sm.chunk_starts[tid];
    }
    // Resume state (cnc_march_samples, rstate != nullptr).  The COUNT pass leaves, per ray, where its first sample was
    // produced — the grid segment i, the cell, the three next-crossing distances and t_last — and the fill pass starts
    // there and stops after the ray's last sample (it knows the count): it replays only the span between a ray's first
    // and last sample, with the very same fp32 operations, not the empty space before and after it.
    constexpr bool SAVE = MODE == 0;
    constexpr bool RESTORE = MODE == 2 || MODE == 3;
    bool    restore = false;
    int64_t n_total = 0;
    int32_t r_i = 0, r_cur[3] = {0, 0, 0};
    float   r_t = 0, r_td[3] = {0, 0, 0};
    if constexpr (RESTORE) {
        if (rstate != nullptr && live) {
            const uint32_t* r = rstate + (size_t)tid * 8;
            r_i = (int32_t)r[0];
            r_cur[0] = (int32_t)r[1]; r_cur[1] = (int32_t)r[2]; r_cur[2] = (int32_t)r[3];
            r_t = __uint_as_float(r[4]);
            r_td[0] = __uint_as_float(r[5]); r_td[1] = __uint_as_float(r[6]); r_td[2] = __uint_as_float(r[7]);
            n_total = sm.chunk_cnts[tid];
            restore = true;
        }
    }
    float near_plane = 0, far_plane = 0, o[3] = {0, 0, 0}, dir[3] = {1, 1, 1};
    if (live) {
        near_plane = near_planes[tid];
        far_plane = far_planes[tid];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            o[a] = rays_o[(size_t)tid * 3 + a];
            dir[a] = rays_d[(size_t)tid * 3 + a];
        }
    }
    const float inv[3] = {1.0f / dir[0], 1.0f / dir[1], 1.0f / dir[2]};
    const int32_t base_hits = tid * n_grids, base_t = tid * n_grids * 2;
    const int32_t i_end = base_t + n_grids * 2 - 1;

    // march state, kept in registers across flushes
    int64_t n_iv = 0, n_sm = 0;
    float   t_last = near_plane;
    bool    continuous = false;
    int32_t i = restore ? r_i : base_t;
    bool    resume = false, finished = !live, stop = false;
    int64_t level = 0;
    float   this_tmax = 0;
    float   tdist[3] = {0, 0, 0}, delta[3] = {0, 0, 0};
    int     step_i[3] = {0, 0, 0}, cur[3] = {0, 0, 0}, over[3] = {0, 0, 0};
    uint32_t st_sm = 0, st_iv = 0;     // entries staged in LDS
    int64_t  fl_sm = 0, fl_iv = 0;     // entries already written out

    for (;;) {
        if (!finished) {
            bool paused = false;
            for (; i < i_end; i++) {
                if (!resume) {
                    const bool is_entering = t_indices[i] < n_grids;
                    level = t_indices[i] % n_grids;
                    if (!hits[base_hits + level]) continue;
                    if (!is_entering) {
                        if (t_indices[i + 1] < n_grids) continue;
                        level = t_indices[i + 1] % n_grids;
                        if (!hits[base_hits + level]) continue;
                    }
                    const float this_tmin = fmaxf(t_sorted[i], near_plane);
                    this_tmax = fminf(t_sorted[i + 1], far_plane);
                    if (this_tmin >= this_tmax) continue;

                    if (!continuous) {
                        if (step_size <= 0.0f) {
                            t_last = this_tmin;
                        } else {
                            const float dt = CONE0 ? step_size : calc_dt(t_last, cone_angle, step_size, 1e10f);
                            while (!(t_last + dt * 0.5f >= this_tmin)) t_last += dt;
                        }
                    }

                    const float* bb = aabbs + level * 6;
#pragma unroll
                    for (int a = 0; a < 3; a++) {   // setup_traversal, utils_grid.cuh:59-118
                        const float resf = (float)res[a];
                        const float ext = bb[3 + a] - bb[a];
                        const float voxel = ext / resf;
                        const float rs = __builtin_fmaf(dir[a], this_tmin + eps, o[a]);
                        const float re = __builtin_fmaf(dir[a], this_tmax - eps, o[a]);
                        cur[a] = clampi((int)(((rs - bb[a]) / ext) * resf), 0, res[a] - 1);
                        const int fin = clampi((int)(((re - bb[a]) / ext) * resf), 0, res[a] - 1);
                        const int start_index = cur[a] + (dir[a] > 0 ? 1 : 0);
                        const float txyz = __builtin_fmaf(
                            bb[a] + __builtin_fmaf((float)start_index, voxel, -rs), inv[a], this_tmin);
                        const bool  flat = dir[a] == 0.0f;
                        const float sf = flat ? 0.0f : (dir[a] > 0.0f ? 1.0f : -1.0f);
                        tdist[a] = flat ? this_tmax : txyz;
                        step_i[a] = (int)sf;
                        delta[a] = flat ? this_tmax : voxel * inv[a] * sf;
                        over[a] = fin + step_i[a];
                    }
                    if constexpr (RESTORE) {
                        if (restore) {      // this segment's set-up is done: continue from the first sample's cell
#pragma unroll
                            for (int a = 0; a < 3; a++) { cur[a] = r_cur[a]; tdist[a] = r_td[a]; }
                            t_last = r_t;
                            restore = false;
                        }
                    }
                }
                resume = false;

                while (limit <= 0 || n_sm < limit) {
                    float t_trav = fminf(tdist[0], fminf(tdist[1], tdist[2]));
                    t_trav = fminf(t_trav, this_tmax);
                    bool occupied = true;
                    if (coarse != nullptr) {
                        const uint32_t cb = (uint32_t)(((level * (res[0] >> 2) + (cur[0] >> 2)) * (res[1] >> 2) + (cur[1] >> 2))
                                                       * (res[2] >> 2) + (cur[2] >> 2));
                        occupied = ((s_coarse[cb >> 5] >> (cb & 31u)) & 1u) != 0u;
                    }
                    if (occupied) {
                        const int64_t cell = (int64_t)(cur[0] * res[1] * res[2] + cur[1] * res[2] + cur[2])
                                             + level * res[0] * res[1] * res[2];
                        occupied = binaries[cell] != 0;
                    }
                    if (!occupied) {
                        if (step_size <= 0.0f) {
                            t_last = t_trav;
                        } else {
                            const float dt = CONE0 ? step_size : calc_dt(t_last, cone_angle, step_size, 1e10f);
                            while (!(t_last + dt * 0.5f >= t_trav)) t_last += dt;
                        }
                        continuous = false;
                    } else {
                        while (limit <= 0 || n_sm < limit) {
                            if constexpr (FILL) {
                                // no room for this step's entries: stop here, flush, come back
                                if (st_sm >= (uint32_t)kRow || (!PAIRS && st_iv + 2 > (uint32_t)kStage)) {
                                    paused = true;
                                    break;
                                }
                            }
                            float t_next;
                            if (step_size <= 0.0f) {
                                t_next = t_trav;
                            } else {
                                const float dt = CONE0 ? step_size : calc_dt(t_last, cone_angle, step_size, 1e10f);
                                if (t_last + dt * 0.5f >= t_trav) break;
                                t_next = t_last + dt;
                            }
                            if constexpr (SAVE) {
                                if (rstate != nullptr && n_sm == 0) {
                                    uint32_t* r = rstate + (size_t)tid * 8;
                                    r[0] = (uint32_t)i;
                                    r[1] = (uint32_t)cur[0]; r[2] = (uint32_t)cur[1]; r[3] = (uint32_t)cur[2];
                                    r[4] = __float_as_uint(t_last);
                                    r[5] = __float_as_uint(tdist[0]); r[6] = __float_as_uint(tdist[1]);
                                    r[7] = __float_as_uint(tdist[2]);
                                }
                            }
                            if (has_iv) {
                                if (!continuous) {
                                    if constexpr (FILL) {
                                        float*   v = st.iv + lane * kStagePitch + st_iv;
                                        uint8_t* f = st.ivf + lane * kStagePitch + st_iv;
                                        v[0] = t_last; f[0] = 1;
                                        v[1] = t_next; f[1] = 2;
                                        st_iv += 2;
                                    }
                                    n_iv += 2;
                                } else {
                                    if constexpr (FILL) {
                                        // the previous edge is still staged (the flush keeps the
                                        // last one back), so it can become a left edge here
                                        st.iv[lane * kStagePitch + st_iv] = t_next;
                                        st.ivf[lane * kStagePitch + st_iv - 1] |= 1;
                                        st.ivf[lane * kStagePitch + st_iv] = 2;
                                        st_iv += 1;
                                    }
                                    n_iv += 1;
                                }
                            }
                            if constexpr (DIRECT) {
                                const int64_t k = cs_sm + n_sm;
                                sm.vals[k] = t_last;
                                iv.vals[k] = t_next;
                                if (sm.ray_indices) sm.ray_indices[k] = tid;
                                if (ex.ray_indices32) ex.ray_indices32[k] = tid;
                                if (ex.positions) {
#pragma unroll
                                    for (int a = 0; a < 3; a++)
                                        ex.positions[k * 3 + a] = sample_coord(o[a], dir[a], t_last, t_next, ex.aabb, a);
                                }
                                if (ex.dirs) {
#pragma unroll
                                    for (int a = 0; a < 3; a++) ex.dirs[k * 3 + a] = dir[a];
                                }
                            } else if constexpr (PAIRS) {
                                st.sm[lane * kPP + st_sm] = t_last;
                                st.iv[lane * kPP + st_sm] = t_next;
                                st_sm++;
                            } else if constexpr (FILL) {
                                if (has_sm) {
                                    st.sm[lane * kStagePitch + st_sm] = (t_next + t_last) * 0.5f;
                                    st_sm++;
                                }
                            }
                            n_sm++;
                            continuous = true;
                            t_last = t_next;
                            if constexpr (RESTORE) {
                                if (n_total > 0 && n_sm >= n_total) { stop = true; break; }     // the ray's last sample
                            }
                            if (t_next >= t_trav) break;
                        }
                        if (paused || stop) break;
                    }
                    // single_traversal, utils_grid.cuh:121-149 (strict '<' tie-break x, y, then z)
                    const int ax = (tdist[0] < tdist[1] && tdist[0] < tdist[2]) ? 0
                                   : (tdist[1] < tdist[2] ? 1 : 2);
                    bool done;
                    if (ax == 0)      { cur[0] += step_i[0]; tdist[0] += delta[0]; done = cur[0] == over[0]; }
                    else if (ax == 1) { cur[1] += step_i[1]; tdist[1] += delta[1]; done = cur[1] == over[1]; }
                    else              { cur[2] += step_i[2]; tdist[2] += delta[2]; done = cur[2] == over[2]; }
                    if (done) break;
                }
                if (paused) {
                    resume = true;
                    break;
                }
                if (stop) break;
            }
            if (!paused) finished = true;
        }
        if constexpr (!FILL) break;
        if constexpr (PAIRS) {
            __syncthreads();
            flush_pairs<PROW>(st, sm.vals, iv.vals, sm.ray_indices, st_sm, cs_sm + fl_sm, tid, ex, o, dir);
            fl_sm += st_sm;
            st_sm = 0;
            __syncthreads();
            if (__ballot(!finished) == 0) break;
        } else if constexpr (FILL) {
            __syncthreads();   // one wave per workgroup: orders the LDS writes before the reads
            if (has_sm) {
                flush_stage<false>(st, sm, st_sm, cs_sm + fl_sm, tid);
                fl_sm += st_sm;
                st_sm = 0;
            }
            if (has_iv) {
                // keep the newest edge back while the ray is still marching: the next sample may
                // turn it into a left edge
                const uint32_t keep = (!finished && st_iv > 0) ? 1u : 0u;
                const uint32_t out = st_iv - keep;
                flush_stage<true>(st, iv, out, cs_iv + fl_iv, tid);
                __syncthreads();
                if (keep && out > 0) {
                    st.iv[lane * kStagePitch] = st.iv[lane * kStagePitch + out];
                    st.ivf[lane * kStagePitch] = st.ivf[lane * kStagePitch + out];
                }
                fl_iv += out;
                st_iv = keep;
            }
            __syncthreads();
            if (__ballot(!finished) == 0) break;
        }
    }
    if (live) {
        if (terminate_planes != nullptr) terminate_planes[tid] = t_last;
        if (has_iv) iv.chunk_cnts[tid] = n_iv;
        if (has_sm) sm.chunk_cnts[tid] = n_sm;
    }
}

// Sample positions (and directions) for the field: positions[s] = o[ray] + d[ray] * t with
// t = t_a[s], or (d * (t_a[s] + t_b[s])) / 2 when t_b is given (the expression of the reference's
// rgb_sigma_fn, examples/utils.py:251-262, evaluated in the same order); optionally mapped to the
// unit cube of an aabb as the field does first thing (ngp.py:518-519).  One pass instead of the
// gather / mul / add / div chain of elementwise kernels.
__global__ __launch_bounds__(256) void k_sample_positions(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const int64_t* __restrict__ ray_indices, const float* __restrict__ t_a,
    const float* __restrict__ t_b, const float* __restrict__ aabb, int64_t S,
    float* __restrict__ positions, float* __restrict__ dirs)
{
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    const int64_t r = ray_indices[s];
    const float   ta = t_a[s];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float o = rays_o[r * 3 + a], d = rays_d[r * 3 + a];
        float p;
        if (t_b) p = o + (d * (ta + t_b[s])) / 2.0f;
        else p = o + d * ta;
        if (aabb) p = (p - aabb[a]) / (aabb[3 + a] - aabb[a]);
        positions[s * 3 + a] = p;
        if (dirs) dirs[s * 3 + a] = d;
    }
}

static Seg to_seg(const cnc_ray_segments_t* s)
{
    Seg r{};
    if (s) {
        r.vals = s->vals; r.chunk_starts = s->chunk_starts; r.chunk_cnts = s->chunk_cnts;
        r.ray_indices = s->ray_indices; r.is_left = s->is_left; r.is_right = s->is_right;
        r.is_valid = s->is_valid;
    }
    return r;
}

}  // namespace cnc

using namespace cnc;

extern "C" int cnc_ray_aabb_intersect(const float* rays_o, const float* rays_d,
                                      const float* aabbs, int32_t n_rays, int32_t n_aabbs,
                                      float near_plane, float far_plane, float miss_value,
                                      float* t_mins, float* t_maxs, uint8_t* hits, void* stream)
{
    const int64_t numel = (int64_t)n_rays * n_aabbs;
    if (numel <= 0) return CNC_OK;
    if (!rays_o || !rays_d || !aabbs || !t_mins || !t_maxs || !hits) return CNC_ERR_INVALID_VALUE;
    const uint32_t blocks = (uint32_t)((numel + 255) / 256);
    hipLaunchKernelGGL(k_ray_aabb, dim3(blocks < 65535u ? blocks : 65535u), dim3(256), 0,
                       (hipStream_t)stream, rays_o, rays_d, aabbs, n_rays, n_aabbs, near_plane,
                       far_plane, miss_value, t_mins, t_maxs, hits);
    return launch_status();
}

extern "C" int cnc_traverse_grids(const float* rays_o, const float* rays_d,
                                  const uint8_t* rays_mask, int32_t n_rays,
                                  const uint8_t* binaries, int32_t n_grids, int32_t resx,
                                  int32_t resy, int32_t resz, const float* aabbs,
                                  const uint8_t* hits, const float* t_sorted,
                                  const int64_t* t_indices, const float* near_planes,
                                  const float* far_planes, float step_size, float cone_angle,
                                  int32_t traverse_steps_limit, int32_t first_pass,
                                  const cnc_ray_segments_t* intervals,
                                  const cnc_ray_segments_t* samples, float* terminate_planes,
                                  void* stream)
{
    if (n_rays <= 0) return CNC_OK;
    if (!rays_o || !rays_d || !binaries || !aabbs || !hits || !t_sorted || !t_indices ||
        !near_planes || !far_planes || n_grids <= 0)
        return CNC_ERR_INVALID_VALUE;
    const Seg iv = to_seg(intervals), sm = to_seg(samples);
    if (!first_pass) {
        if (iv.chunk_cnts && (!iv.vals || !iv.chunk_starts || !iv.ray_indices || !iv.is_left || !iv.is_right))
            return CNC_ERR_INVALID_VALUE;
        if (sm.chunk_cnts && (!sm.vals || !sm.chunk_starts || !sm.ray_indices || !sm.is_valid))
            return CNC_ERR_INVALID_VALUE;
    }
    const uint32_t blocks = div_up((uint32_t)n_rays, 64);
    if (first_pass) {
        hipLaunchKernelGGL(k_traverse<0>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, rays_o,
                           rays_d, rays_mask, n_rays, binaries, n_grids, resx, resy, resz, aabbs,
                           hits, t_sorted, t_indices, near_planes, far_planes, step_size,
                           cone_angle, traverse_steps_limit, iv, sm, terminate_planes, 64);
    } else {
        const size_t lds = 64 * kStagePitch * (2 * sizeof(float) + 1);
        hipLaunchKernelGGL(k_traverse<1>, dim3(blocks), dim3(64), lds, (hipStream_t)stream, rays_o,
                           rays_d, rays_mask, n_rays, binaries, n_grids, resx, resy, resz, aabbs,
                           hits, t_sorted, t_indices, near_planes, far_planes, step_size,
                           cone_angle, traverse_steps_limit, iv, sm, terminate_planes, 64);
    }
    return launch_status();
}

namespace cnc {
// words of the coarse occupancy of n_grids grids of res^3 cells, or 0 when the kernels do not take one for this shape
static uint32_t coarse_words_of(int32_t n_grids, int32_t resx, int32_t resy, int32_t resz)
{
    if (n_grids <= 0 || resx < 4 || resy < 4 || resz < 4 || (resx & 3) || (resy & 3) || (resz & 3)) return 0;
    const uint64_t bits = (uint64_t)n_grids * (uint64_t)(resx >> 2) * (uint64_t)(resy >> 2) * (uint64_t)(resz >> 2);
    const uint64_t words = (bits + 31) / 32;
    return words <= 2048 ? (uint32_t)words : 0u;           // 8 KB of LDS per 64-lane workgroup at most
}

// one lane per block of 4 x 4 x 4 cells: 16 aligned dwords of the byte grid, the wave's answers packed by ballot
__global__ __launch_bounds__(64) void k_occupancy_coarse(const uint8_t* __restrict__ binaries, int32_t n_grids, int32_t resx,
                                                         int32_t resy, int32_t resz, uint32_t* __restrict__ words)
{
    const uint32_t cx = resx >> 2, cy = resy >> 2, cz = resz >> 2;
    const uint32_t n = (uint32_t)n_grids * cx * cy * cz, b = blockIdx.x * 64 + threadIdx.x;
    bool any = false;
    if (b < n) {
        const uint32_t z = b % cz, y = (b / cz) % cy, x = (b / (cz * cy)) % cx, g = b / (cz * cy * cx);
        const uint8_t* base = binaries + (size_t)g * resx * resy * resz;
        uint32_t acc = 0;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++)
#pragma unroll
            for (uint32_t j = 0; j < 4; j++)
                acc |= *reinterpret_cast<const uint32_t*>(base + ((size_t)(4 * x + i) * resy + (4 * y + j)) * resz + 4 * z);
        any = acc != 0;
    }
    const uint64_t m = __ballot(any);
    if (threadIdx.x == 0) words[blockIdx.x * 2] = (uint32_t)m;
    if (threadIdx.x == 32 && blockIdx.x * 2 + 1 < (n + 31) / 32) words[blockIdx.x * 2 + 1] = (uint32_t)(m >> 32);
}
}  // namespace cnc

extern "C" uint32_t cnc_occupancy_coarse_words(int32_t n_grids, int32_t resx, int32_t resy, int32_t resz)
{
    return cnc::coarse_words_of(n_grids, resx, resy, resz);
}

extern "C" int cnc_occupancy_coarse_bits(const uint8_t* binaries, int32_t n_grids, int32_t resx, int32_t resy, int32_t resz,
                                         uint32_t* words, void* stream)
{
    const uint32_t nw = cnc::coarse_words_of(n_grids, resx, resy, resz);
    if (nw == 0) return CNC_ERR_UNSUPPORTED;
    if (!binaries || !words || ((uintptr_t)binaries & 3u)) return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(cnc::k_occupancy_coarse, dim3((nw + 1) / 2), dim3(64), 0, (hipStream_t)stream, binaries, n_grids, resx, resy,
                       resz, words);
    return cnc::launch_status();
}

extern "C" int cnc_march_samples(const float* rays_o, const float* rays_d, const uint8_t* rays_mask, int32_t n_rays,
                                 const uint8_t* binaries, int32_t n_grids, int32_t resx, int32_t resy,
                                 int32_t resz, const float* aabbs, const uint8_t* hits, const float* t_sorted,
                                 const int64_t* t_indices, const float* near_planes, const float* far_planes,
                                 float step_size, float cone_angle, int32_t traverse_steps_limit,
                                 int64_t* chunk_cnts, const int64_t* chunk_starts, float* t_starts, float* t_ends,
                                 int64_t* ray_indices, float* terminate_planes, uint32_t* resume_state,
                                 float* positions, float* dirs, int32_t* ray_indices32, const float* aabb, void* stream)
{
    return cnc_march_samples_coarse(rays_o, rays_d, rays_mask, n_rays, binaries, n_grids, resx, resy, resz, aabbs, hits, t_sorted,
                                    t_indices, near_planes, far_planes, step_size, cone_angle, traverse_steps_limit, chunk_cnts,
                                    chunk_starts, t_starts, t_ends, ray_indices, terminate_planes, resume_state, positions, dirs,
                                    ray_indices32, aabb, nullptr, stream);
}

extern "C" int cnc_march_samples_coarse(const float* rays_o, const float* rays_d, const uint8_t* rays_mask, int32_t n_rays,
                                        const uint8_t* binaries, int32_t n_grids, int32_t resx, int32_t resy,
                                        int32_t resz, const float* aabbs, const uint8_t* hits, const float* t_sorted,
                                        const int64_t* t_indices, const float* near_planes, const float* far_planes,
                                        float step_size, float cone_angle, int32_t traverse_steps_limit,
                                        int64_t* chunk_cnts, const int64_t* chunk_starts, float* t_starts, float* t_ends,
                                        int64_t* ray_indices, float* terminate_planes, uint32_t* resume_state,
                                        float* positions, float* dirs, int32_t* ray_indices32, const float* aabb,
                                        const uint32_t* coarse_bits, void* stream)
{
    if (n_rays <= 0) return CNC_OK;
    if (!rays_o || !rays_d || !binaries || !aabbs || !hits || !t_sorted || !t_indices || !near_planes ||
        !far_planes || n_grids <= 0 || !chunk_cnts)
        return CNC_ERR_INVALID_VALUE;
    // the fill pass of a small batch runs 16 rays per 64-lane block (measured on 27 k rays: 0.86 -> 0.67 ms; 8 / 32 rays:
    // 0.80 / 0.75; the count pass, which stages nothing in LDS, prefers full waves: 0.33 vs 0.36)
    const bool     cone0 = cone_angle == 0.0f && step_size > 0.0f;      // dt is the constant step (k_traverse CONE0)
    const char*    dm = getenv("CNC_MARCH_DIRECT_MAX");        // measurement / test switch (0: always stage)
    const int      direct_max = dm ? atoi(dm) : (1 << 17);
    const bool     direct = chunk_starts && n_rays < direct_max;
    const int32_t  rpb = (chunk_starts && !direct && n_rays < (1 << 17)) ? 16 : 64;
    const uint32_t blocks = div_up((uint32_t)n_rays, (uint32_t)rpb);
    Seg none{}, sm{};
    sm.chunk_cnts = chunk_cnts;
    const uint32_t cw = coarse_bits ? cnc::coarse_words_of(n_grids, resx, resy, resz) : 0u;
    if (coarse_bits && cw == 0) return CNC_ERR_UNSUPPORTED;
    const SampleExtras no_ex{nullptr, nullptr, nullptr, nullptr};
    if (!chunk_starts) {        // pass 1: counts only
        if (cone0) hipLaunchKernelGGL((k_traverse<0, 32, true>), dim3(blocks), dim3(64), cw * 4, (hipStream_t)stream, rays_o, rays_d, rays_mask,
                           n_rays, binaries, n_grids, resx, resy, resz, aabbs, hits, t_sorted, t_indices,
                           near_planes, far_planes, step_size, cone_angle, traverse_steps_limit, none, sm,
                           terminate_planes, rpb, resume_state, no_ex, coarse_bits, cw, 0u);
        else hipLaunchKernelGGL(k_traverse<0>, dim3(blocks), dim3(64), cw * 4, (hipStream_t)stream, rays_o, rays_d, rays_mask,
                           n_rays, binaries, n_grids, resx, resy, resz, aabbs, hits, t_sorted, t_indices,
                           near_planes, far_planes, step_size, cone_angle, traverse_steps_limit, none, sm,
                           terminate_planes, rpb, resume_state, no_ex, coarse_bits, cw, 0u);
        return launch_status();
    }
    // the ray of a sample: int64 (the nerfacc boundary), int32 (internal consumers), or neither when the caller only
    // wants what the extras carry
    if (!t_starts || !t_ends || (!ray_indices && !ray_indices32 && !positions)) return CNC_ERR_INVALID_VALUE;
    const SampleExtras ex{positions, dirs, ray_indices32, aabb};
    Seg ends{};
    sm.chunk_starts = const_cast<int64_t*>(chunk_starts);
    sm.vals = t_starts;
    sm.ray_indices = ray_indices;
    ends.vals = t_ends;
    if (direct) {
        if (cone0) hipLaunchKernelGGL((k_traverse<3, 32, true>), dim3(blocks), dim3(64), cw * 4, (hipStream_t)stream, rays_o, rays_d, rays_mask,
                           n_rays, binaries, n_grids, resx, resy, resz, aabbs, hits, t_sorted, t_indices, near_planes,
                           far_planes, step_size, cone_angle, traverse_steps_limit, ends, sm, terminate_planes, rpb,
                           resume_state, ex, coarse_bits, cw, 0u);
        else hipLaunchKernelGGL((k_traverse<3>), dim3(blocks), dim3(64), cw * 4, (hipStream_t)stream, rays_o, rays_d, rays_mask,
                           n_rays, binaries, n_grids, resx, resy, resz, aabbs, hits, t_sorted, t_indices, near_planes,
                           far_planes, step_size, cone_angle, traverse_steps_limit, ends, sm, terminate_planes, rpb,
                           resume_state, ex, coarse_bits, cw, 0u);
        return launch_status();
    }
    // staging row length, measured on the 800x800 bench frame (count + fill, ms).  Marching whole rays: 8 -> 3.53,
    // 16 -> 2.55, 32 -> 2.32, 64 -> 2.89 (LDS then limits the waves per CU).  With the fill pass resumed at the first
    // sample (every step emits, rows fill evenly): 8 -> 2.27, 16 -> 2.04, 32 -> 2.15, 64 -> 2.54.
    const char* ps = getenv("CNC_PAIR_STAGE");
    const int   row = ps ? atoi(ps) : (resume_state ? 16 : 32);
#define CNC_LAUNCH_PAIRS_(R, C0)                                                                                    \
    hipLaunchKernelGGL((k_traverse<2, R, C0>), dim3(blocks), dim3(64), 2 * 64 * (R + 1) * sizeof(float) + cw * 4, \
                       (hipStream_t)stream, rays_o, rays_d, rays_mask, n_rays, binaries, n_grids, resx, resy, resz, \
                       aabbs, hits, t_sorted, t_indices, near_planes, far_planes, step_size, cone_angle,            \
                       traverse_steps_limit, ends, sm, terminate_planes, rpb, resume_state, ex, coarse_bits, cw,     \
                       (uint32_t)(2 * 64 * (R + 1) * sizeof(float)))
#define CNC_LAUNCH_PAIRS(R)                   \
    do {                                      \
        if (cone0) CNC_LAUNCH_PAIRS_(R, true); \
        else CNC_LAUNCH_PAIRS_(R, false);     \
    } while (0)
    if (row == 8) CNC_LAUNCH_PAIRS(8);
    else if (row == 16) CNC_LAUNCH_PAIRS(16);
    else if (row == 64) CNC_LAUNCH_PAIRS(64);
    else CNC_LAUNCH_PAIRS(32);
#undef CNC_LAUNCH_PAIRS
#undef CNC_LAUNCH_PAIRS_
    return launch_status();
}

extern "C" int cnc_sample_positions(const float* rays_o, const float* rays_d,
                                    const int64_t* ray_indices, const float* t_a, const float* t_b,
                                    const float* aabb, int64_t n_samples, float* positions,
                                    float* dirs, void* stream)
{
    if (n_samples <= 0) return CNC_OK;
    if (!rays_o || !rays_d || !ray_indices || !t_a || !positions) return CNC_ERR_INVALID_VALUE;
    const int64_t blocks = (n_samples + 255) / 256;
    if (blocks > 0x7FFFFFFF) return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_sample_positions, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream,
                       rays_o, rays_d, ray_indices, t_a, t_b, aabb, n_samples, positions, dirs);
    return launch_status();
}
