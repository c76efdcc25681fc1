/*
 * cnc_oracle.c — TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, scalar loops) of the
 * reference's CUDA kernels on the CNC hot path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load the library built from this file; the product
 * (cnc_amd/, libcnc_hip.so) never links, imports or calls it.
 *
 * Parity pinning: the reference ships no tests and no golden vectors,
 * and its native path is CUDA-only (unbuildable here: no nvcc, no CUDA headers).  What IS pinned,
 * by tests/test_oracle_pins.py against fixtures generated from the imported Python reference
 * (tests/golden/make_golden.py): the grid index/hash (examples/utils.py:492-511), the slab test
 * (nerfacc/grid.py:55-91), the four segmented scans and pack_info / render_* docstring examples
 * (nerfacc/scan.py, volrend.py).  The interpolation / marching / mask kernels have no executable
 * reference twin: for those the oracle is "parity unpinned" beyond those anchors.
 *
 * Arithmetic policy: this file is compiled with -ffp-contract=off; doubles appear exactly where
 * the reference's C++ literal promotion makes it compute in double; fmaf() appears exactly where
 * nvcc's default -fmad=true contracts a float multiply into the dependent add.
 *
 * Each function cites the reference lines it follows (paths relative to the reference root).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_D 3

static inline uint32_t u32min(uint32_t a, uint32_t b) { return a < b ? a : b; }

/* ---------------------------------------------------------------------------------------------
 * a1. fast_hash + get_grid_index — gridencoder/src/gridencoder.cu:45-87
 * Returns the ROW index (the reference multiplies by n_features and adds ch at :86).
 * ------------------------------------------------------------------------------------------- */
uint32_t orc_grid_index(uint32_t D, const uint32_t* q, uint32_t hashmap_size, uint32_t resolution)
{
    static const uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u,
                                       2097192037u, 1434869437u, 2165219737u};
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) { /* :72-77 */
        index += q[d] * stride;
        stride *= resolution;
    }
    if (stride > hashmap_size) { /* :80-82, gridtype is always 0 */
        index = 0;
        for (uint32_t d = 0; d < D; d++) index ^= q[d] * primes[d];
    }
    return index % hashmap_size; /* :86 */
}

/* vectorised helper for the index pins: pos [n, D] u32 -> rows [n] u32 */
void orc_grid_index_many(uint32_t D, const uint32_t* pos, uint32_t n, uint32_t hashmap_size,
                         uint32_t resolution, uint32_t* rows)
{
    for (uint32_t i = 0; i < n; i++) rows[i] = orc_grid_index(D, pos + (size_t)i * D, hashmap_size, resolution);
}

/* occupancy-box test shared by the encoder — gridencoder.cu:221-276 */
static int orc_box_any(uint32_t D, const uint32_t* q, uint32_t R, uint32_t Rb, const uint8_t* vxl)
{
    float scale_re = (float)(1.0 / ((double)(float)R - 2.0));                 /* :224 */
    uint32_t lo[ORC_MAX_D], hi[ORC_MAX_D];
    for (uint32_t d = 0; d < D; d++) {
        float pn = (float)(((double)(float)q[d] - 0.5) * (double)scale_re);  /* :228 */
        float g1 = pn - scale_re;                                             /* :230-234 */
        g1 = g1 * (float)Rb;
        g1 = g1 < 0 ? 0 : g1;
        g1 = g1 > (float)(Rb - 1) ? (float)(Rb - 1) : g1;
        lo[d] = (uint32_t)(int)g1;
        float g2 = pn + scale_re;                                             /* :236-240 */
        g2 = g2 * (float)Rb;
        g2 = g2 < 0 ? 0 : g2;
        g2 = g2 > (float)(Rb - 1) ? (float)(Rb - 1) : g2;
        hi[d] = (uint32_t)(int)g2;
    }
    if (D == 1) {
        for (uint32_t a = lo[0]; a <= hi[0]; a++) if (vxl[a]) return 1;
    } else if (D == 2) {
        for (uint32_t a = lo[0]; a <= hi[0]; a++)
            for (uint32_t b = lo[1]; b <= hi[1]; b++)
                if (vxl[a * Rb + b]) return 1;
    } else {
        for (uint32_t a = lo[0]; a <= hi[0]; a++)
            for (uint32_t b = lo[1]; b <= hi[1]; b++)
                for (uint32_t c = lo[2]; c <= hi[2]; c++)
                    if (vxl[a * Rb * Rb + b * Rb + c]) return 1;
    }
    return 0;
}

/* per-(point, level-slot) corner set-up shared by forward and backward.
 * gridencoder.cu:160-291 (forward) == :430-562 (backward).  Returns 0 for an out-of-range point. */
typedef struct {
    float    w[8];
    uint32_t row[8];
    int      valid[8];
    float    wn_re;
} orc_corners_t;

static int orc_corners(uint32_t D, const float* x, uint32_t R, uint32_t hs, uint32_t Rb,
                       const uint8_t* vxl, orc_corners_t* c)
{
    for (uint32_t d = 0; d < D; d++)
        if (x[d] < 0 || x[d] > 1) return 0;                                   /* :134-140 */
    float pos[ORC_MAX_D];
    uint32_t g[ORC_MAX_D];
    for (uint32_t d = 0; d < D; d++) {
        float prod = x[d] * (float)(R - 2);                                   /* :173, float*float */
        pos[d] = (float)((double)prod + 0.5);                                 /* + double literal  */
        g[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)g[d];
    }
    float wn = 0;
    for (uint32_t i = 0; i < (1u << D); i++) {
        float w = 1;
        uint32_t q[ORC_MAX_D];
        for (uint32_t d = 0; d < D; d++) {                                    /* :200-208 */
            if ((i & (1u << d)) == 0) { w *= 1 - pos[d]; q[d] = g[d]; }
            else                      { w *= pos[d];     q[d] = u32min(g[d] + 1, R - 1); }
        }
        int border = 0;
        for (uint32_t d = 0; d < D; d++)
            if (q[d] == 0 || q[d] == R - 1) { border = 1; break; }            /* :212-219 */
        int m = vxl ? orc_box_any(D, q, R, Rb, vxl) : 1;                      /* :221-276 */
        c->w[i] = w;
        c->valid[i] = (!border && m);
        c->row[i] = 0;
        if (c->valid[i]) { c->row[i] = orc_grid_index(D, q, hs, R); wn += w; } /* :281-285 */
    }
    if (wn == 0) wn = (float)((double)wn + 1e-9);                             /* :288-290 */
    c->wn_re = (float)(1.0 / (double)wn);                                     /* :291 */
    return 1;
}

/* ---------------------------------------------------------------------------------------------
 * a2. kernel_grid — gridencoder.cu:96-316 (dy_dx branch :319-395 is dead: ngp.py:84)
 * outputs [L, N, F].  threads: OpenMP threads for the cpu_baseline leg (1 = scalar).
 * ------------------------------------------------------------------------------------------- */
void orc_grid_encode_forward(const float* inputs, const float* emb, const int32_t* offsets,
                             const int32_t* resolutions, float* out, uint32_t N, uint32_t D,
                             uint32_t F, uint32_t L, uint32_t Rb, const uint8_t* vxl,
                             const int32_t* min_level_id, int ste_binary, int threads)
{
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (uint32_t k = 0; k < L; k++) {
        for (uint32_t b = 0; b < N; b++) {
            uint32_t level = (min_level_id ? (uint32_t)min_level_id[b] : 0u) + k; /* :118-126 */
            const float* table = emb + (size_t)(uint32_t)offsets[level] * F;
            float* o = out + ((size_t)k * N + b) * F;
            uint32_t hs = (uint32_t)(offsets[level + 1] - offsets[level]);
            uint32_t R = (uint32_t)resolutions[level];
            orc_corners_t c;
            for (uint32_t ch = 0; ch < F; ch++) o[ch] = 0;
            if (!orc_corners(D, inputs + (size_t)b * D, R, hs, Rb, vxl, &c)) continue;
            for (uint32_t i = 0; i < (1u << D); i++) {                        /* :294-303 */
                if (!c.valid[i]) continue;
                float t = c.w[i] * c.wn_re;
                const float* row = table + (size_t)c.row[i] * F;
                for (uint32_t ch = 0; ch < F; ch++) {
                    float v = row[ch];
                    if (ste_binary) v = (v >= 0) ? 1.0f : -1.0f; /* STE_binary.forward, ngp.py:26-30 */
                    o[ch] = fmaf(t, v, o[ch]);                    /* nvcc contracts  += t*v */
                }
            }
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * a3. kernel_grid_backward — gridencoder.cu:399-585.  grad [L, N, F]; grad_emb accumulated.
 * The reference scatters with float atomicAdd in nondeterministic order; here the order is
 * (level-slot, point, corner) and, when acc64 != NULL, a double-precision shadow is kept so
 * tests can bound the reordering error instead of guessing a tolerance.
 * ------------------------------------------------------------------------------------------- */
static void orc_bwd_point(uint32_t k, uint32_t b, const float* grad, const float* inputs, const float* emb,
                          const int32_t* offsets, const int32_t* resolutions, float* grad_emb,
                          double* acc64, uint32_t N, uint32_t D, uint32_t F, uint32_t Rb,
                          const uint8_t* vxl, const int32_t* min_level_id, int ste_binary, int atomic)
{
    uint32_t level = (min_level_id ? (uint32_t)min_level_id[b] : 0u) + k;
    size_t base = (size_t)(uint32_t)offsets[level] * F;
    uint32_t hs = (uint32_t)(offsets[level + 1] - offsets[level]);
    uint32_t R = (uint32_t)resolutions[level];
    const float* g = grad + ((size_t)k * N + b) * F;
    orc_corners_t c;
    if (!orc_corners(D, inputs + (size_t)b * D, R, hs, Rb, vxl, &c)) return;  /* :435-440 */
    for (uint32_t i = 0; i < (1u << D); i++) {
        if (!c.valid[i]) continue;
        float t = c.w[i] * c.wn_re;
        size_t at = base + (size_t)c.row[i] * F;
        for (uint32_t ch = 0; ch < F; ch++) {
            if (ste_binary) {           /* STE_binary.backward mask, ngp.py:35-39 */
                float v = emb[at + ch];
                if (!(v >= -1.0f && v <= 1.0f)) continue;
            }
            float contrib = t * g[ch];                                        /* :580 */
            if (atomic) {
#ifdef _OPENMP
#pragma omp atomic
#endif
                grad_emb[at + ch] += contrib;
            } else {
                grad_emb[at + ch] += contrib;
                if (acc64) acc64[at + ch] += (double)contrib;
            }
        }
    }
}

void orc_grid_encode_backward(const float* grad, const float* inputs, const float* emb,
                              const int32_t* offsets, const int32_t* resolutions, float* grad_emb,
                              double* acc64, uint32_t N, uint32_t D, uint32_t F, uint32_t L,
                              uint32_t Rb, const uint8_t* vxl, const int32_t* min_level_id,
                              int ste_binary, int threads)
{
    if (threads > 1 && acc64 && !min_level_id) {
        /* float64 shadow wanted AND threads: the level slots write disjoint table rows, so one
         * thread per level keeps the (point, corner) order of the serial loop inside each level —
         * the same sums, bit for bit, as threads == 1 */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
        for (uint32_t k = 0; k < L; k++)
            for (uint32_t b = 0; b < N; b++)
                orc_bwd_point(k, b, grad, inputs, emb, offsets, resolutions, grad_emb, acc64, N, D, F, Rb, vxl,
                              min_level_id, ste_binary, 0);
        return;
    }
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (uint32_t k = 0; k < L; k++)
        for (uint32_t b = 0; b < N; b++)
            orc_bwd_point(k, b, grad, inputs, emb, offsets, resolutions, grad_emb, acc64, N, D, F, Rb, vxl,
                          min_level_id, ste_binary, threads > 1);
}

/* ---------------------------------------------------------------------------------------------
 * a4. dy_dx branch of kernel_grid — gridencoder.cu:319-395 — and kernel_input_backward — :588-614.
 * Dead in CNC (ngp.py:58-60 refuses calc_grad_inputs, :84 passes dy_dx=None) but part of the
 * `_gridencoder` interface, so it is restated like everything else.
 * dy_dx [N, L, D, F].  Per input axis gd: the 2^(D-1) edges along gd, weight (R-2) * prod of the
 * other axes' interpolation weights, (right - left) table values; NO renormalisation over valid
 * corners and no occupancy mask here (the reference applies neither); vertices on the border ring
 * read as 0.  Out-of-range points give zeros (:143-158).
 * ------------------------------------------------------------------------------------------- */
void orc_grid_dy_dx(const float* inputs, const float* emb, const int32_t* offsets,
                    const int32_t* resolutions, float* dy_dx, uint32_t N, uint32_t D, uint32_t F,
                    uint32_t L, const int32_t* min_level_id, int ste_binary)
{
    for (uint32_t k = 0; k < L; k++) {
        for (uint32_t b = 0; b < N; b++) {
            uint32_t level = (min_level_id ? (uint32_t)min_level_id[b] : 0u) + k;
            const float* table = emb + (size_t)(uint32_t)offsets[level] * F;
            uint32_t hs = (uint32_t)(offsets[level + 1] - offsets[level]);
            uint32_t R = (uint32_t)resolutions[level];
            const float* x = inputs + (size_t)b * D;
            float* o = dy_dx + (((size_t)b * L + k) * D) * F;                  /* :323 */
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (x[d] < 0 || x[d] > 1) oob = 1;
            if (oob) { for (uint32_t j = 0; j < D * F; j++) o[j] = 0; continue; }
            float pos[ORC_MAX_D];
            uint32_t g[ORC_MAX_D];
            for (uint32_t d = 0; d < D; d++) {                                  /* :171-177 */
                float prod = x[d] * (float)(R - 2);
                pos[d] = (float)((double)prod + 0.5);
                g[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)g[d];
            }
            for (uint32_t gd = 0; gd < D; gd++) {
                float acc[32];
                for (uint32_t ch = 0; ch < F; ch++) acc[ch] = 0;
                for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                    float w = (float)(R - 2);                                   /* :333 */
                    uint32_t q[ORC_MAX_D];
                    for (uint32_t nd = 0; nd + 1 < D; nd++) {                   /* :337-347 */
                        uint32_t d = (nd >= gd) ? nd + 1 : nd;
                        if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; q[d] = g[d]; }
                        else                         { w *= pos[d];     q[d] = u32min(g[d] + 1, R - 1); }
                    }
                    int zl = 0, zr = 0;
                    uint32_t rl = 0, rr = 0;
                    q[gd] = g[gd];                                              /* :349-361 */
                    for (uint32_t d = 0; d < D; d++) if (q[d] == 0 || q[d] == R - 1) { zl = 1; break; }
                    if (!zl) rl = orc_grid_index(D, q, hs, R);
                    q[gd] = u32min(g[gd] + 1, R - 1);                           /* :363-374 */
                    for (uint32_t d = 0; d < D; d++) if (q[d] == 0 || q[d] == R - 1) { zr = 1; break; }
                    if (!zr) rr = orc_grid_index(D, q, hs, R);
                    for (uint32_t ch = 0; ch < F; ch++) {                       /* :377-387 */
                        float vl = 0, vr = 0;
                        if (!zl) vl = table[(size_t)rl * F + ch];
                        if (!zr) vr = table[(size_t)rr * F + ch];
                        if (ste_binary) {  /* STE_binary.forward is applied to the table before the call */
                            if (!zl) vl = (vl >= 0) ? 1.0f : -1.0f;
                            if (!zr) vr = (vr >= 0) ? 1.0f : -1.0f;
                        }
                        float t = w * (vr - vl);
                        acc[ch] = fmaf(t, 1.0f, acc[ch]);          /* += t * pos_deriv, pos_deriv = 1 */
                    }
                }
                for (uint32_t ch = 0; ch < F; ch++) o[gd * F + ch] = acc[ch];   /* :390-393 */
            }
        }
    }
}

void orc_input_backward(const float* grad, const float* dy_dx, float* grad_inputs, uint32_t N,
                        uint32_t D, uint32_t F, uint32_t L)
{
    for (uint32_t t = 0; t < N * D; t++) {                                      /* :596-613 */
        uint32_t b = t / D, d = t - b * D;
        const float* dy = dy_dx + (size_t)b * L * D * F;
        float result = 0;
        for (uint32_t l = 0; l < L; l++)
            for (uint32_t ch = 0; ch < F; ch++)
                result = fmaf(grad[((size_t)l * N + b) * F + ch], dy[((size_t)l * D + d) * F + ch], result);
        grad_inputs[t] = result;
    }
}

/* ---------------------------------------------------------------------------------------------
 * a6. cnt_np_embed / cnt_np_embed_backward — gridencoder.cu:873-915, 972-1020
 * ------------------------------------------------------------------------------------------- */
static int orc_cnt_loc(const int16_t* p, uint32_t R, uint32_t F, uint32_t axis, uint32_t* q,
                       uint32_t* ot_loc)
{
    /* short -> uint32 conversion as in `uint32_t pos_grid_local[3] = {inputs[0],...}` (:885) */
    q[0] = (uint32_t)(int32_t)p[0]; q[1] = (uint32_t)(int32_t)p[1]; q[2] = (uint32_t)(int32_t)p[2];
    uint32_t scale = R - 2;
    for (int d = 0; d < 3; d++)
        if (q[d] <= 0 || q[d] >= R - 1) return 0;                             /* :895-898 */
    switch (axis) {                                                           /* :902-906 */
    case 0: *ot_loc = (q[0] - 1) * scale * F * 2 + (q[1] - 1) * F * 2; break;
    case 1: *ot_loc = (q[0] - 1) * scale * F * 2 + (q[2] - 1) * F * 2; break;
    default: *ot_loc = (q[1] - 1) * scale * F * 2 + (q[2] - 1) * F * 2; break;
    }
    return 1;
}

void orc_cnt_np_embed(const int16_t* inputs, const float* emb, float* out, uint32_t N, uint32_t R,
                      uint32_t F, uint32_t hashmap_size, uint32_t axis)
{
    for (uint32_t b = 0; b < N; b++) {
        uint32_t q[3], loc;
        if (!orc_cnt_loc(inputs + (size_t)b * 3, R, F, axis, q, &loc)) continue;
        const float* row = emb + (size_t)orc_grid_index(3, q, hashmap_size, R) * F; /* :886 */
        for (uint32_t ch = 0; ch < F; ch++) {                                 /* :908-914 */
            if ((double)row[ch] > 0.9) out[loc + ch * 2 + 0] += 1.0f;
            else                       out[loc + ch * 2 + 1] += 1.0f;
        }
    }
}

void orc_cnt_np_embed_backward(const int16_t* inputs, const float* emb, const float* out_sum,
                               const float* grad, float* grad_emb, double* acc64, uint32_t N,
                               uint32_t R, uint32_t F, uint32_t hashmap_size, uint32_t axis)
{
    for (uint32_t b = 0; b < N; b++) {
        uint32_t q[3], loc;
        if (!orc_cnt_loc(inputs + (size_t)b * 3, R, F, axis, q, &loc)) continue;
        size_t at = (size_t)orc_grid_index(3, q, hashmap_size, R) * F;
        uint32_t half = loc / 2;                                              /* :1009 */
        for (uint32_t ch = 0; ch < F; ch++) {                                 /* :1011-1019 */
            float gv = 1 / out_sum[half + ch];
            float contrib = ((double)emb[at + ch] > 0.9) ? gv * grad[loc + ch * 2 + 0]
                                                         : -gv * grad[loc + ch * 2 + 1];
            grad_emb[at + ch] += contrib;
            if (acc64) acc64[at + ch] += (double)contrib;
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * a7. query_mask_3D{,_qlist} — my_cuda_backen/aligner_kernel.cu:4-326
 * res_list == NULL -> scalar `resolution`.
 * ------------------------------------------------------------------------------------------- */
/* `contraction`: which of the reference's multiply-add pairs are taken as ONE fused operation — what nvcc's default
 * -fmad=true does to a same-type multiply that feeds an add, and what cannot be observed without a CUDA build.
 *   bit 0: the cell's upper edge  float(idx) * Rb_re + Rb_re          (aligner_kernel.cu:57,71,216-232)
 *   bit 1: the accumulation       overlap += oa * ob (* oc)            (:73, :233)
 * 3 = both fused: the oracle's (and the HIP kernels') reading.  tools/fmaf_exposure.py counts how many outputs change under
 * the other three readings. */
void orc_query_mask_contraction(const int16_t* points, uint32_t D, const uint8_t* vxl, int Rb, int16_t* mask,
                                int32_t* overlap, int resolution, const int64_t* res_list, uint32_t N, int contraction);

void orc_query_mask(const int16_t* points, uint32_t D, const uint8_t* vxl, int Rb, int16_t* mask,
                    int32_t* overlap, int resolution, const int64_t* res_list, uint32_t N)
{
    orc_query_mask_contraction(points, D, vxl, Rb, mask, overlap, resolution, res_list, N, 3);
}

static inline float edge_(float idx, float Rb_re, int fused) { return fused ? fmaf(idx, Rb_re, Rb_re) : idx * Rb_re + Rb_re; }
static inline float acc_(float prod_a, float b, float area, int fused) { return fused ? fmaf(prod_a, b, area) : area + prod_a * b; }

void orc_query_mask_contraction(const int16_t* points, uint32_t D, const uint8_t* vxl, int Rb, int16_t* mask,
                                int32_t* overlap, int resolution, const int64_t* res_list, uint32_t N, int contraction)
{
    const int fe = contraction & 1, fa = (contraction >> 1) & 1;
    const float Rb_re = (float)(1.0 / (double)(float)Rb);                      /* :14 */
    for (uint32_t i = 0; i < N; i++) {
        float R = res_list ? (float)res_list[i] : (float)resolution;
        float scale_re = (float)(1.0 / ((double)R - 2.0));                    /* :19 / :98 */
        float pn[3] = {0, 0, 0};
        uint32_t lo[3], hi[3];
        for (uint32_t d = 0; d < D; d++)
            pn[d] = (float)(((double)(float)points[(size_t)i * D + d] - 0.5) * (double)scale_re); /* :25 */
        for (uint32_t d = 0; d < D; d++) {                                    /* :29-41 */
            float g1 = pn[d] - scale_re;
            g1 = g1 * (float)Rb;
            g1 = g1 < 0 ? 0 : g1;
            g1 = g1 > (float)(Rb - 1) ? (float)(Rb - 1) : g1;
            lo[d] = (uint32_t)(int)g1;
            float g2 = pn[d] + scale_re;
            g2 = g2 * (float)Rb;
            g2 = g2 < 0 ? 0 : g2;
            g2 = g2 > (float)(Rb - 1) ? (float)(Rb - 1) : g2;
            hi[d] = (uint32_t)(int)g2;
        }
        int m = 0;
        float area = 0;
        for (int a = (int)lo[0]; (uint32_t)a <= hi[0]; a++) {                 /* :56-73 / :213-236 */
            float ra = fminf(edge_((float)a, Rb_re, fe), pn[0] + scale_re);
            float la = fmaxf((float)a * Rb_re, pn[0] - scale_re);
            float oa = ra - la;
            for (int b = (int)lo[1]; (uint32_t)b <= hi[1]; b++) {
                float rb = fminf(edge_((float)b, Rb_re, fe), pn[1] + scale_re);
                float lb = fmaxf((float)b * Rb_re, pn[1] - scale_re);
                float ob = rb - lb;
                if (D == 2) {
                    int mt = vxl[(size_t)a * Rb + b] != 0;
                    m |= mt;
                    if (mt) area = acc_(oa, ob, area, fa);                    /* += oa*ob */
                } else {
                    for (int c = (int)lo[2]; (uint32_t)c <= hi[2]; c++) {
                        float rc = fminf(edge_((float)c, Rb_re, fe), pn[2] + scale_re);
                        float lc = fmaxf((float)c * Rb_re, pn[2] - scale_re);
                        float oc = rc - lc;
                        int mt = vxl[((size_t)a * Rb + b) * Rb + c] != 0;
                        m |= mt;
                        if (mt) area = acc_(oa * ob, oc, area, fa);           /* += oa*ob*oc */
                    }
                }
            }
        }
        area = area * (float)Rb * (float)Rb;                                  /* :75 / :238 */
        if (D == 3) area = area * (float)Rb;
        mask[i] = (int16_t)m;
        overlap[i] = (int32_t)(area * 1000);                                  /* :79 */
    }
}

/* ---------------------------------------------------------------------------------------------
 * a8. align_and_pack forward / backward — aligner_kernel.cu:413-435, 498-516
 * ------------------------------------------------------------------------------------------- */
void orc_align_and_pack_forward(const float* feat, const int64_t* cnt, const int64_t* cumsum,
                                float* packed, uint32_t N, uint32_t M, uint32_t F, float V)
{
    for (uint32_t i = 0; i < N; i++)
        for (uint32_t j = 0; j < M; j++)
            for (uint32_t k = 0; k < F; k++)
                packed[((size_t)i * M + j) * F + k] =
                    ((int64_t)(j + 1) > cnt[i]) ? V : feat[(size_t)(cumsum[i] + j) * F + k];
}

void orc_align_and_pack_backward(const float* dpacked, const int64_t* cnt, const int64_t* cumsum,
                                 float* dfeat, uint32_t N, uint32_t M, uint32_t F)
{
    for (uint32_t i = 0; i < N; i++)
        for (uint32_t j = 0; j < M; j++) {
            if ((int64_t)(j + 1) > cnt[i]) continue;
            for (uint32_t k = 0; k < F; k++)
                dfeat[(size_t)(cumsum[i] + j) * F + k] = dpacked[((size_t)i * M + j) * F + k];
        }
}

/* ---------------------------------------------------------------------------------------------
 * a11. ray_aabb_intersect — nerfacc/cuda/csrc/include/utils_grid.cuh:11-56, grid.cu:320-349
 * ------------------------------------------------------------------------------------------- */
static int orc_slab(const float* o, const float* inv, const float* bb, float near, float far,
                    float* tmin_o, float* tmax_o)
{
    float tmin, tmax, a, b;
    if (inv[0] >= 0) { tmin = (bb[0] - o[0]) * inv[0]; tmax = (bb[3] - o[0]) * inv[0]; }
    else             { tmin = (bb[3] - o[0]) * inv[0]; tmax = (bb[0] - o[0]) * inv[0]; }
    for (int d = 1; d < 3; d++) {
        if (inv[d] >= 0) { a = (bb[d] - o[d]) * inv[d];     b = (bb[3 + d] - o[d]) * inv[d]; }
        else             { a = (bb[3 + d] - o[d]) * inv[d]; b = (bb[d] - o[d]) * inv[d]; }
        if (tmin > b || a > tmax) return 0;
        if (a > tmin) tmin = a;
        if (b < tmax) tmax = b;
    }
    if (tmax <= 0) return 0;
    *tmin_o = fmaxf(tmin, near);
    *tmax_o = fminf(tmax, far);
    return 1;
}

void orc_ray_aabb_intersect(const float* rays_o, const float* rays_d, const float* aabbs,
                            int32_t n_rays, int32_t n_aabbs, float near, float far, float miss,
                            float* t_mins, float* t_maxs, uint8_t* hits)
{
    for (int32_t t = 0; t < n_rays * n_aabbs; t++) {
        int32_t r = t / n_aabbs, a = t % n_aabbs;
        float inv[3] = {1.0f / rays_d[r * 3], 1.0f / rays_d[r * 3 + 1], 1.0f / rays_d[r * 3 + 2]};
        float t0, t1;
        int hit = orc_slab(rays_o + r * 3, inv, aabbs + a * 6, near, far, &t0, &t1);
        t_mins[t] = hit ? t0 : miss;
        t_maxs[t] = hit ? t1 : miss;
        hits[t] = (uint8_t)hit;
    }
}

/* ---------------------------------------------------------------------------------------------
 * a12. traverse_grids_kernel — nerfacc/cuda/csrc/grid.cu:68-318,
 *      setup_traversal / single_traversal — include/utils_grid.cuh:59-149,
 *      roi_to_unit — include/utils_contraction.cuh:19-24, clamp/make_int3 — utils_math.cuh
 * One call == one kernel launch (first_pass semantics as at :100-114, :252-291).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    float*   vals;
    int64_t* chunk_starts;
    int64_t* chunk_cnts;
    int64_t* ray_indices;
    uint8_t* is_left;
    uint8_t* is_right;
    uint8_t* is_valid;
} orc_segments_t;

static inline float orc_clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
static inline int orc_clampi(int f, int a, int b) { int m = f < b ? f : b; return a > m ? a : m; }
static inline float orc_calc_dt(float t, float cone, float dmin, float dmax)
{
    return orc_clampf(t * cone, dmin, dmax);                                  /* grid.cu:23-28 */
}

void orc_traverse_grids(const float* rays_o, const float* rays_d, const uint8_t* rays_mask,
                        int32_t n_rays, const uint8_t* binaries, int32_t n_grids, int32_t resx,
                        int32_t resy, int32_t resz, const float* aabbs, const uint8_t* hits,
                        const float* t_sorted, const int64_t* t_indices, const float* near_planes,
                        const float* far_planes, float step_size, float cone_angle,
                        int32_t limit, int32_t first_pass, const orc_segments_t* iv,
                        const orc_segments_t* sm, float* terminate_planes)
{
    const float eps = 1e-6f;
    const int res[3] = {resx, resy, resz};
    const int has_iv = iv && iv->chunk_cnts, has_sm = sm && sm->chunk_cnts;
    /* rays are independent (each writes its own slots): one OpenMP thread per block of rays, like
     * one CUDA thread per ray */
    #pragma omp parallel for schedule(dynamic, 64) if (n_rays > 256)
    for (int32_t tid = 0; tid < n_rays; tid++) {
        if (rays_mask && !rays_mask[tid]) continue;                           /* :100 */
        if (has_iv && !first_pass && iv->chunk_cnts[tid] == 0) continue;      /* :103-106 */
        if (has_sm && !first_pass && sm->chunk_cnts[tid] == 0) continue;
        int64_t cs_iv = 0, cs_sm = 0;
        if (!first_pass) {
            if (has_iv) cs_iv = iv->chunk_starts[tid];
            if (has_sm) cs_sm = sm->chunk_starts[tid];
        }
        const float near_plane = near_planes[tid], far_plane = far_planes[tid];
        const float* o = rays_o + (size_t)tid * 3;
        const float* dir = rays_d + (size_t)tid * 3;
        const float inv[3] = {1.0f / dir[0], 1.0f / dir[1], 1.0f / dir[2]};   /* data_spec_packed.cuh:47-53 */
        const int32_t base_hits = tid * n_grids, base_t = tid * n_grids * 2;
        int64_t n_iv = 0, n_sm = 0;
        float t_last = near_plane;
        int continuous = 0;
        for (int32_t i = base_t; i < base_t + n_grids * 2 - 1; i++) {         /* :138 */
            int is_entering = t_indices[i] < n_grids;
            int64_t level = t_indices[i] % n_grids;
            if (!hits[base_hits + level]) continue;
            if (!is_entering) {                                               /* :150-158 */
                if (t_indices[i + 1] < n_grids) continue;
                level = t_indices[i + 1] % n_grids;
                if (!hits[base_hits + level]) continue;
            }
            float this_tmin = fmaxf(t_sorted[i], near_plane);                 /* :161-163 */
            float this_tmax = fminf(t_sorted[i + 1], far_plane);
            if (this_tmin >= this_tmax) continue;
            if (!continuous) {                                                /* :166-177 */
                if (step_size <= 0.0f) t_last = this_tmin;
                else {
                    float dt = orc_calc_dt(t_last, cone_angle, step_size, 1e10f);
                    while (!(t_last + dt * 0.5f >= this_tmin)) t_last += dt;
                }
            }
            const float* bb = aabbs + level * 6;
            /* setup_traversal, utils_grid.cuh:59-118 */
            float tdist[3], delta[3];
            int step_i[3], cur[3], fin[3], over[3];
            for (int a = 0; a < 3; a++) {
                float resf = (float)res[a];
                float voxel = (bb[3 + a] - bb[a]) / resf;                     /* :67 */
                float rs = fmaf(dir[a], this_tmin + eps, o[a]);               /* :68  o + d*(tmin+eps) */
                float re = fmaf(dir[a], this_tmax - eps, o[a]);               /* :69 */
                cur[a] = orc_clampi((int)(((rs - bb[a]) / (bb[3 + a] - bb[a])) * resf), 0, res[a] - 1);
                fin[a] = orc_clampi((int)(((re - bb[a]) / (bb[3 + a] - bb[a])) * resf), 0, res[a] - 1);
                int start_index = cur[a] + (dir[a] > 0 ? 1 : 0);              /* :86-89 */
                float txyz = fmaf(bb[a] + fmaf((float)start_index, voxel, -rs), inv[a], this_tmin); /* :92-93 */
                tdist[a] = (dir[a] == 0.0f) ? this_tmax : txyz;               /* :95-99 */
                float sf = (dir[a] == 0.0f) ? 0.0f : (dir[a] > 0.0f ? 1.0f : -1.0f);
                step_i[a] = (int)sf;
                float dtmp = voxel * inv[a] * sf;                             /* :111 */
                delta[a] = (dir[a] == 0.0f) ? this_tmax : dtmp;
                over[a] = fin[a] + step_i[a];                                 /* grid.cu:200 */
            }
            while (limit <= 0 || n_sm < limit) {                              /* :201 */
                float t_trav = fminf(tdist[0], fminf(tdist[1], tdist[2]));    /* :204 (min==fminf) */
                t_trav = fminf(t_trav, this_tmax);
                int64_t cell = (int64_t)(cur[0] * res[1] * res[2] + cur[1] * res[2] + cur[2]
                                         + level * res[0] * res[1] * res[2]);
                if (!binaries[cell]) {                                        /* :216-228 */
                    if (step_size <= 0.0f) t_last = t_trav;
                    else {
                        float dt = orc_calc_dt(t_last, cone_angle, step_size, 1e10f);
                        while (!(t_last + dt * 0.5f >= t_trav)) t_last += dt;
                    }
                    continuous = 0;
                } else {
                    while (limit <= 0 || n_sm < limit) {                      /* :234-297 */
                        float t_next;
                        if (step_size <= 0.0f) t_next = t_trav;
                        else {
                            float dt = orc_calc_dt(t_last, cone_angle, step_size, 1e10f);
                            if (t_last + dt * 0.5f >= t_trav) break;
                            t_next = t_last + dt;
                        }
                        if (has_iv) {                                         /* :252-278 */
                            if (!continuous) {
                                if (!first_pass) {
                                    int64_t k = cs_iv + n_iv;
                                    iv->vals[k] = t_last; iv->ray_indices[k] = tid; iv->is_left[k] = 1;
                                }
                                n_iv++;
                                if (!first_pass) {
                                    int64_t k = cs_iv + n_iv;
                                    iv->vals[k] = t_next; iv->ray_indices[k] = tid; iv->is_right[k] = 1;
                                }
                                n_iv++;
                            } else {
                                if (!first_pass) {
                                    int64_t k = cs_iv + n_iv;
                                    iv->vals[k] = t_next; iv->ray_indices[k] = tid;
                                    iv->is_left[k - 1] = 1; iv->is_right[k] = 1;
                                }
                                n_iv++;
                            }
                        }
                        if (has_sm && !first_pass) {                          /* :281-291 */
                            int64_t k = cs_sm + n_sm;
                            sm->vals[k] = (t_next + t_last) * 0.5f;
                            sm->ray_indices[k] = tid;
                            sm->is_valid[k] = 1;
                        }
                        n_sm++;
                        continuous = 1;
                        t_last = t_next;
                        if (t_next >= t_trav) break;
                    }
                }
                /* single_traversal, utils_grid.cuh:121-149 */
                int ax = (tdist[0] < tdist[1] && tdist[0] < tdist[2]) ? 0 : (tdist[1] < tdist[2] ? 1 : 2);
                cur[ax] += step_i[ax];
                tdist[ax] += delta[ax];
                if (cur[ax] == over[ax]) break;
            }
        }
        if (terminate_planes) terminate_planes[tid] = t_last;                 /* :310-316 */
        if (has_iv) iv->chunk_cnts[tid] = n_iv;
        if (has_sm) sm->chunk_cnts[tid] = n_sm;
    }
}

/* ---------------------------------------------------------------------------------------------
 * a13. segmented scans — nerfacc/cuda/csrc/include/utils_scan.cuh:21-263, scan.cu:9-304
 * The 32-element tile tree (up-sweep :75-81, down-sweep :84-90) is restated literally so the
 * float association equals the CUDA kernel's.
 * ------------------------------------------------------------------------------------------- */
static inline float orc_op(int prod, float a, float b) { return prod ? a * b : a + b; }

static void orc_scan_tile(float* buf, int prod)
{
    for (uint32_t s = 16, d = 1; s >= 1; s >>= 1, d <<= 1)
        for (uint32_t t = 0; t < s; t++) {
            uint32_t off = (2 * t + 1) * d - 1;
            buf[off + d] = orc_op(prod, buf[off], buf[off + d]);
        }
    for (uint32_t s = 2, d = 8; d >= 1; s <<= 1, d >>= 1)
        for (uint32_t t = 0; t + 1 < s; t++) {
            uint32_t off = 2 * (t + 1) * d - 1;
            buf[off + d] = orc_op(prod, buf[off], buf[off + d]);
        }
}

/* exclusive: 0/1; prod: 0 = sum (init 0), 1 = product (init 1); reverse: scan right-to-left
 * (reverse iterators, scan.cu:42-55); normalize as utils_scan.cuh:100-110 / 226-237.          */
void orc_segmented_scan(const int64_t* starts, const int64_t* cnts, const float* in, float* out,
                        uint32_t n_rays, int exclusive, int prod, int reverse, int normalize)
{
    const float init = prod ? 1.0f : 0.0f;
    for (uint32_t r = 0; r < n_rays; r++) {
        const int64_t s0 = starts[r];
        const uint32_t n = (uint32_t)cnts[r];
        if (n == 0) continue;
#define IDX(j) (reverse ? s0 + n - 1 - (j) : s0 + (j))
#define SRC(j) in[IDX(j)]
#define DST(j) out[IDX(j)]
        float total = init;
        if (exclusive) DST(0) = init;
        for (uint32_t col = 0; col < n; col += 32) {
            float buf[32];
            for (uint32_t t = 0; t < 32; t++) buf[t] = (col + t < n) ? SRC(col + t) : init;
            buf[0] = orc_op(prod, buf[0], total);
            orc_scan_tile(buf, prod);
            for (uint32_t t = 0; t < 32; t++) {
                if (exclusive) { if (col + t + 1 < n) DST(col + t + 1) = buf[t]; }
                else           { if (col + t < n)     DST(col + t) = buf[t]; }
            }
            total = buf[31];
        }
        if (normalize) {
            float den = fmaxf(total, 1e-10f);
            for (uint32_t j = exclusive ? 1 : 0; j < n; j++) DST(j) = DST(j) / den;
        }
#undef SRC
#undef DST
#undef IDX
    }
}

/* prod backward: reverse sum-scan of grad_out*out, divided by clamp_min(in, 1e-10) — scan.cu:199-210 */
void orc_prod_backward(const int64_t* starts, const int64_t* cnts, const float* in,
                       const float* outp, const float* gout, float* gin, uint32_t n_rays,
                       int64_t n_edges, int exclusive)
{
    float* tmp = (float*)malloc(sizeof(float) * (size_t)(n_edges > 0 ? n_edges : 1));
    for (int64_t i = 0; i < n_edges; i++) tmp[i] = gout[i] * outp[i];
    orc_segmented_scan(starts, cnts, tmp, gin, n_rays, exclusive, 0, 1, 0);
    for (int64_t i = 0; i < n_edges; i++) gin[i] = gin[i] / fmaxf(in[i], 1e-10f);
    free(tmp);
}

int orc_openmp_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
