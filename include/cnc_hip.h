/*
 * cnc_hip.h — C ABI of libcnc_hip.so, the MI355X (gfx950) implementation of the CNC hot path.
 *
 * Every entry point replaces one function of the reference's three pybind/torch extensions
 * (`_gridencoder`, `pack_and_align`, `nerfacc.csrc`); the reference interface each one stands in
 * for is cited next to it (paths relative to the reference checkout).  The ABI is plain C:
 * raw DEVICE pointers, sizes and a HIP stream (passed as void*, i.e. a hipStream_t; NULL = the
 * legacy default stream).  No torch types, no allocation inside the library, no state kept between
 * calls: all calls are re-entrant and stream-ordered (asynchronous; nothing here synchronises the
 * device).  One exception, opt-in: a cnc_backward_plan object the caller creates and owns.  (ABI v28 removed the two
 * process-wide setters of earlier versions, cnc_mlp_set_variant and cnc_set_persistent_share: measurement switches whose
 * experiments are closed — docs/engineering_log.md.)
 * Environment variables named CNC_* that a few entry points read are measurement switches too: they are
 * read on every call, never cached.
 *
 * Ownership: the caller owns every buffer.  Where the reference's host wrapper allocates its
 * result (torch::zeros / torch::empty inside the extension), the host-side mirror in
 * cnc_amd/backends/ allocates with torch and passes the pointer down, so the Python-visible
 * behaviour (callee returns a fresh tensor) is unchanged.
 *
 * Return value: CNC_OK (0) or a negative CNC_ERR_* code; cnc_error_string() gives the text the
 * Python mirror raises as RuntimeError (the reference raises RuntimeError via TORCH_CHECK /
 * std::runtime_error for the same conditions).
 *
 * Arithmetic policy (shared with oracle/): all float math is IEEE fp32 with contraction OFF,
 * except the handful of sites where nvcc's default -fmad=true fuses a same-type multiply into
 * the following add; those use an explicit fmaf (listed in DESIGN.md "Arithmetic policy").
 */
#ifndef CNC_HIP_H
#define CNC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNC_OK                 0
#define CNC_ERR_INVALID_VALUE -1   /* bad size / null pointer / unsupported n_features or num_dim */
#define CNC_ERR_UNSUPPORTED   -2   /* valid in the reference, not built here (e.g. half tables)  */
#define CNC_ERR_LAUNCH        -3   /* hipGetLastError() != hipSuccess after the launch           */

/* element type tags for the few polymorphic entry points */
#define CNC_F32 0

/* flags for cnc_grid_encode_{forward,backward} */
#define CNC_FLAG_NONE        0u
#define CNC_FLAG_STE_BINARY  1u   /* treat table value v as (v >= 0 ? +1 : -1) while gathering:
                                     fuses STE_binary.forward (examples/radiance_fields/ngp.py:22-31)
                                     into the gather so the 4-7 full-table elementwise passes of the
                                     reference disappear.  In backward, also applies the STE mask
                                     |v| <= 1 (ngp.py:33-39) to the scattered gradient.            */
#define CNC_FLAG_LEVELS_FINEST_FIRST 2u   /* backward scheduling hint for a call that only carries coarse levels
                                            * (same result): level slots last-to-first, and the level slot as
                                            * the fast block index so that resident blocks are spread over all
                                            * levels (fewer same-row atomics in flight; 0.58 -> 0.50 ms on the 9
                                            * coarse levels of the 16-level bench grid).  With fine levels in
                                            * the call the interleaving costs cache locality: leave it unset. */
#define CNC_FLAG_BIN_LANE_STORES 4u   /* measurement switch for cnc_grid_encode_backward_binned / _overlapped (same
                                       * result): the round-2 bin pass, every lane storing its own items, instead of
                                       * the default that writes a block's items in bin order (k_bwd_bin_sorted) */

#define CNC_FLAG_CELL_MERGE 8u   /* cnc_grid_encode_backward (same result to fp32 summation order): the points of a
                                 * 1024-point block that fall into one cell of a level are summed in LDS and every
                                 * distinct (cell, corner row) goes out as ONE atomic (grid_encode_cells.hip) — with or
                                 * without occupancy mask / per-point level windows, volumes and planes.  For the calls
                                 * of a training step's context pass: lattice vertices in hash-slot order, which share
                                 * cells inside a block but never consecutively (0.70 -> 0.43 ms for its 3-D call).
                                 * D in {2, 3}, F in {2, 4, 8}, no dy_dx, fewer than 64 levels of resolution < 2^16
                                 * (points on higher levels are scattered one by one); other shapes ignore the flag. */
#define CNC_FLAG_CELL_CARRY 16u  /* with CNC_FLAG_CELL_MERGE: two cells of a block that are neighbours along x share
                                 * 2^(D-1) vertices; each shared vertex is written once, next to its partner in the
                                 * 64-byte segment, by one of the two cells, which walks the other's points as well
                                 * (0.43 -> 0.28 ms for the context pass's 3-D call: the atomic requests are what that
                                 * call is made of).  Costs time where the requests are not the bound (many points per
                                 * cell: the planes' context levels).                                               */

const char* cnc_error_string(int code);
int         cnc_abi_version(void);              /* bumps when a signature below changes */

/* ------------------------------------------------------------------------------------------
 * Hash-grid encoder  —  replaces gridencoder/src/gridencoder.h:12-54 (bindings.cpp:5-9)
 * ---------------------------------------------------------------------------------------- */

/* grid_encode_forward (gridencoder.h:12-22, gridencoder.cu:752-806; kernel_grid :96-396).
 *   inputs      [N, D] f32 in [0,1]          embeddings [rows, F] f32
 *   offsets     [L+1] i32 (absolute rows; may be a slice of a longer table, ngp.py:90)
 *   resolutions [L] i32                      outputs    [L, N, F] f32 (level-major, :131)
 *   binary_vxl  NULL or bool[Rb^D]           min_level_id NULL or i32 [N] (per-point level window)
 *   dy_dx       NULL, or f32 [N, L, D, F]: d outputs / d inputs per level as the dy_dx branch of
 *               kernel_grid computes it (gridencoder.cu:319-395: edges along each axis, weight (R-2) x
 *               the other axes' weights, no renormalisation, no occupancy mask; zeros for points
 *               outside [0,1]).  CNC itself never asks for it (ngp.py:58-60,84).
 *   PV is accepted and ignored, as in the reference (gridencoder.cu:304-308).                  */
int cnc_grid_encode_forward(const float* inputs, const float* embeddings,
                            const int32_t* offsets, const int32_t* resolutions,
                            float* outputs,
                            uint32_t N, uint32_t D, uint32_t F, uint32_t L,
                            uint32_t Rb, float PV,
                            float* dy_dx, const uint8_t* binary_vxl, const int32_t* min_level_id,
                            uint32_t flags, const int32_t* occ_sat,
                            const uint32_t* vertex_bits, const int32_t* vertex_bit_offsets,
                            uint32_t out_ld, uint32_t out_col, void* stream);
/*   out_ld / out_col: 0 / 0 = the reference's level-major outputs [L, N, F].  out_ld != 0: write
 *   point-major into a wider feature matrix, outputs[b * out_ld + out_col + l * F + f], so several
 *   encoders fill the MLP input directly (no permute, no cat; ngp.py:111,631-642).  Requires
 *   out_col + L*F <= out_ld and both multiples of min(F, 4).                                       */
/*   occ_sat (may be NULL; only read when binary_vxl != NULL): summed-volume table of binary_vxl,
 *   int32 [(Rb+1)^D], sat[a][b][c] = #set cells with indices < (a,b,c).  With it the per-corner
 *   occupancy test costs 2^D loads instead of a scan of the whole +-1 vertex box; the result is
 *   identical (integer arithmetic).  The host mirror builds it with three cumsums per grid update.
 *   vertex_bits / vertex_bit_offsets (may be NULL; only read when binary_vxl != NULL): per-level bit planes of the
 *   same predicate, built once per grid update by cnc_grid_vertex_bits — bit q0 + R (q1 + R q2) of level l's plane,
 *   which starts at 32-bit word vertex_bit_offsets[l] of vertex_bits (i32 [L], indexed like `resolutions`; < 0 =
 *   no plane for that level, the test falls back to occ_sat / the scan).  One bit read per corner instead of 2^D
 *   table entries; identical results.                                                                        */

/* Vertex bit plane of ONE level of resolution R (ABI v22): words[cnc_grid_vertex_bits_words(D, R)] u32, bit
 * q0 + R (q1 + R q2) = 1 iff the +-1-vertex box of vertex q holds a set occupancy cell (gridencoder.cu:221-276,
 * the per-corner test of kernel_grid evaluated for every vertex).  occ_sat as above.                        */
uint64_t cnc_grid_vertex_bits_words(uint32_t D, uint32_t R);
int cnc_grid_vertex_bits(const int32_t* occ_sat, uint32_t D, uint32_t Rb, uint32_t R, uint32_t* words, void* stream);

/* grid_encode_backward (gridencoder.h:24-36, gridencoder.cu:808-866; kernel_grid_backward :399-585).
 *   grad [L, N, F] f32;  grad_embeddings [rows, F] f32, ACCUMULATED into (caller zero-fills,
 *   ngp.py:129).  dy_dx [N, L, D, F] and grad_inputs [N, D] are given together or both NULL:
 *   grad_inputs[b][d] = sum_l sum_f grad[l][b][f] * dy_dx[b][l][d][f] (kernel_input_backward :588-614). */
int cnc_grid_encode_backward(const float* grad, const float* inputs, const float* embeddings,
                             const int32_t* offsets, const int32_t* resolutions,
                             float* grad_embeddings,
                             uint32_t N, uint32_t D, uint32_t F, uint32_t L, uint32_t Rb,
                             const float* dy_dx, float* grad_inputs,
                             const uint8_t* binary_vxl, const int32_t* min_level_id,
                             uint32_t flags, const uint32_t* ste_clip_count,
                             const int32_t* occ_sat,
                             const uint32_t* vertex_bits, const int32_t* vertex_bit_offsets,
                             uint32_t grad_ld, uint32_t grad_col,
                             void* stream);   /* grad_ld/grad_col: layout of `grad`, as out_ld/out_col */
/*   ste_clip_count (device pointer, may be NULL): with CNC_FLAG_STE_BINARY, the number of table
 *   entries with |v| > 1 as counted by cnc_pack_sign_bits.  When it reads 0 the STE mask is the
 *   identity and the scatter skips the per-row parameter gather it otherwise needs.            */

/* ---- MI355X-specific fast path for binarised tables (no counterpart in the reference) ----
 * CNC always trains with ste_binary=True (train_CNC_nerf_synthetic.py:141): the encoder only ever
 * sees sign(table).  cnc_pack_sign_bits writes that as a bit plane (bit = value >= 0; F bits per
 * row, rows packed little-endian into bytes: 6.1 MB for the 16L x 2^19 x F8 table instead of 187 MiB)
 * which stays resident in L2 / Infinity Cache; cnc_grid_encode_forward_bits gathers from it and
 * returns exactly the bits cnc_grid_encode_forward(CNC_FLAG_STE_BINARY) returns.
 *   bits: uint8 [ceil(rows * F / 8)], rows = offsets[last]; F in {1,2,4,8,16,32}.              */
int cnc_pack_sign_bits(const float* embeddings, uint8_t* bits, uint64_t rows, uint32_t F,
                       uint32_t* clip_count /* NULL or device u32: set to #entries with |v| > 1 */,
                       void* stream);
int cnc_grid_encode_forward_bits(const float* inputs, const uint8_t* bits,
                                 const int32_t* offsets, const int32_t* resolutions,
                                 float* outputs,
                                 uint32_t N, uint32_t D, uint32_t F, uint32_t L, uint32_t Rb,
                                 const uint8_t* binary_vxl, const int32_t* min_level_id,
                                 const int32_t* occ_sat,
                                 const uint32_t* vertex_bits, const int32_t* vertex_bit_offsets,
                                 uint32_t out_ld, uint32_t out_col,
                                 void* stream);   /* out_ld/out_col, vertex_bits as cnc_grid_encode_forward */

/* Same gradient as cnc_grid_encode_backward (no binary_vxl / min_level_id), with the n_binned FINEST
 * levels taken off the global-atomic path (D = 3, F in {2,4,8}): their (sample, corner-pair)
 * contributions are binned by 256-row table slab and summed in LDS by one wave per slab (several for a
 * bin far above the mean load; cnc_amd/csrc/grid_encode_binned.hip).  Pays where every sample sits in its own cell, i.e. levels
 * finer than the sample spacing; coarser levels stay on the run-merging atomic kernel.
 *   level_rows: upper bound of rows per binned level (offsets[l+1]-offsets[l], <= 2^20); a level
 *               with more rows than that is routed to atomics on the device, results unchanged.
 *   workspace : device scratch of cnc_grid_encode_backward_binned_workspace(N, n_binned,
 *               level_rows) bytes = bins of 8x the mean load (more = deeper bins; a full bin spills to
 *               atomics).  The library
 *               clears the part it needs; contents are dead after the call.
 *   The slab owners add into grad_embeddings with plain read-modify-writes: calls that target the
 *   same grad_embeddings must be stream-ordered, not concurrent (the atomic entry point has no such
 *   restriction).                                                                                */
uint64_t cnc_grid_encode_backward_binned_workspace(uint32_t N, uint32_t n_binned, uint32_t level_rows);
int cnc_grid_encode_backward_binned(const float* grad, const float* inputs, const float* embeddings,
                                    const int32_t* offsets, const int32_t* resolutions,
                                    float* grad_embeddings,
                                    uint32_t N, uint32_t D, uint32_t F, uint32_t L, uint32_t flags,
                                    const uint32_t* ste_clip_count, uint32_t grad_ld, uint32_t grad_col,
                                    uint32_t n_binned, uint32_t level_rows,
                                    void* workspace, uint64_t workspace_bytes, void* stream);

/* The same call with its two halves overlapped on HIP streams the library owns through a PLAN object (ABI v21;
 * cnc_amd/csrc/grid_encode_overlap.hip).  The coarse levels run on `stream`, the n_binned finest levels in one or two
 * groups on the plan's side streams, forked from and joined to `stream` with events: when the call returns, all of
 * its work is ordered on `stream` again (no record_stream / synchronisation needed by the caller).  Same result as
 * cnc_grid_encode_backward_binned; 5 % faster at N = 2^20, 10-16 % at 2^16..2^18.  Falls back to the serial entry when
 * plan is NULL, N < 2^16, or there is nothing to overlap.
 *   plan      : cnc_backward_plan_create() once per (device, caller stream): the current device at creation owns
 *               the side streams.  A plan serves one call at a time (its events are re-recorded per call); use one
 *               plan per concurrently used caller stream / host thread.  No globals inside the library.
 *   workspace : cnc_grid_encode_backward_overlapped_workspace(N, n_binned, level_rows) bytes, 16-byte aligned.   */
typedef struct cnc_backward_plan cnc_backward_plan;
int      cnc_backward_plan_create(cnc_backward_plan** plan);
int      cnc_backward_plan_destroy(cnc_backward_plan* plan);
uint64_t cnc_grid_encode_backward_overlapped_workspace(uint32_t N, uint32_t n_binned, uint32_t level_rows);
int cnc_grid_encode_backward_overlapped(cnc_backward_plan* plan, const float* grad, const float* inputs,
                                        const float* embeddings, const int32_t* offsets,
                                        const int32_t* resolutions, float* grad_embeddings,
                                        uint32_t N, uint32_t D, uint32_t F, uint32_t L, uint32_t flags,
                                        const uint32_t* ste_clip_count, uint32_t grad_ld, uint32_t grad_col,
                                        uint32_t n_binned, uint32_t level_rows,
                                        void* workspace, uint64_t workspace_bytes, void* stream);

/* cnt_np_embed (gridencoder.h:39-44, gridencoder.cu:873-970): ±1 vote counts of the finest 3-D
 * level projected on a plane.  inputs i16 [N,3]; embeddings_clip [hashmap_size, F] f32;
 * outputs [res-2, res-2, F, 2] f32, ACCUMULATED into (caller zero-fills, utils_bpp_acc.py:39).
 * axis 0|1|2 = xy|xz|yz.                                                                       */
int cnc_cnt_np_embed(const int16_t* inputs, const float* embeddings_clip, float* outputs,
                     uint32_t N, uint32_t resolution, uint32_t F, uint32_t hashmap_size,
                     uint32_t axis, void* stream);

/* cnt_np_embed_backward (gridencoder.h:47-54, gridencoder.cu:972-1087).
 * outputs_sum [res-2,res-2,F,1]; grad [res-2,res-2,F,2]; grad_embeddings accumulated into.     */
int cnc_cnt_np_embed_backward(const int16_t* inputs, const float* embeddings_clip,
                              const float* outputs_sum, const float* grad, float* grad_embeddings,
                              uint32_t N, uint32_t resolution, uint32_t F, uint32_t hashmap_size,
                              uint32_t axis, void* stream);

/* cnt_np_embed from a static plan, without atomics (MI355X extension; cnc_amd/csrc/cnt_votes.hip).
 * The vertex list of cnt_np_embed changes only with the occupancy grid, so the host sorts it once per
 * refresh — by pixel of the projection plane for the forward count, by table row for the backward sum
 * — and each call is a segmented gather with one plain store per pixel / row.
 *   cnc_cnt_np_plan: per vertex, rows[i] = table row (gridencoder.cu:45-87) and pixels[i] =
 *     (u-1)*(res-2) + (w-1) of the plane `axis`; 0xFFFFFFFF for vertices cnt_np_embed skips.  Either
 *     output may be NULL.
 *   cnc_cnt_np_embed_planned: outputs [n_pixels, F, 2] is WRITTEN (not accumulated); rows_by_pixel
 *     holds the rows of the valid vertices ordered by pixel, pixel_seg [n_pixels+1] their segment
 *     starts.  Same counts as cnc_cnt_np_embed, bit for bit.
 *   cnc_cnt_np_embed_planned_backward: grad_over_sum [n_pixels, F, 2] = grad * (1 / outputs_sum);
 *     pixels_by_row / row_seg [n_rows+1] = the vertices' pixels ordered by table row;
 *     grad_embeddings [n_rows, F] is accumulated into (rows without vertices untouched).           */
int cnc_cnt_np_plan(const int16_t* inputs, uint32_t N, uint32_t resolution, uint32_t hashmap_size,
                    uint32_t axis, uint32_t* rows, uint32_t* pixels, void* stream);
/* (ABI v25) The plan's input in one pass: rows [N] and the pixels of the xy / xz / yz planes [N] each, as int32, with
 * the vertices cnt_np_embed skips keyed PAST the end (row = hashmap_size, pixel = (res-2)^2) so that sorting leaves
 * them behind the last segment — no compaction; unsorted [3] (zeroed by the caller) gets a non-zero word for every
 * plane whose pixel list is not already ascending.                                                               */
int cnc_cnt_np_plan3(const int16_t* inputs, uint32_t N, uint32_t resolution, uint32_t hashmap_size, int32_t* rows,
                     int32_t* pix_xy, int32_t* pix_xz, int32_t* pix_yz, int32_t* unsorted, void* stream);
int cnc_cnt_np_embed_planned(const uint32_t* rows_by_pixel, const int32_t* pixel_seg,
                             const float* embeddings_clip, float* outputs, uint32_t n_pixels,
                             uint32_t F, void* stream);
int cnc_cnt_np_embed_planned_backward(const uint32_t* pixels_by_row, const int32_t* row_seg,
                                      const float* embeddings_clip, const float* grad_over_sum,
                                      float* grad_embeddings, uint32_t n_rows, uint32_t F, void* stream);
/* The same forward counts in two steps, for callers that project one table onto several planes (the three calls
 * of utils_bpp_acc.py:590-600 vote on the same embeddings): masks[r] bit ch = (embeddings_clip[r][ch] > 0.9),
 * F <= 32, packed once; then a 4-byte gather per vertex instead of a 4 F byte row.  Counts are integers: equal. */
/* (ABI v24) counts [S, S, F, 2] of one projection -> the dense one-level table the dimension-wise context encodes from
 * (utils_bpp_acc.py:39-55, 515-526): table [R, R, F], R = S + 2 = cnt0 / ((cnt0 + cnt1) + 1e-6) on the inner S x S
 * pixels, zero on the ring; sums [S, S, F] = the denominators, kept for the backward.  And back: grad_over_sum
 * [S, S, F, 2] = [(1 / sums) * g_table(inner), 0], the input of cnc_cnt_np_embed_planned_backward{,3}.             */
int cnc_vote_fraction_table(const float* cnt, uint32_t S, uint32_t F, float* table, float* sums, void* stream);
int cnc_vote_fraction_table_backward(const float* g_table, const float* sums, uint32_t S, uint32_t F,
                                     float* grad_over_sum, void* stream);
/* Backward of the three projections (xy, xz, yz) of ONE table in one pass: grad_embeddings [n_rows, F] is WRITTEN
 * (0 for rows without vertices) = the sum of the three cnc_cnt_np_embed_planned_backward results on a zeroed table. */
int cnc_cnt_np_embed_planned_backward3(const uint32_t* pixels_by_row_xy, const uint32_t* pixels_by_row_xz,
                                       const uint32_t* pixels_by_row_yz, const int32_t* row_seg,
                                       const float* embeddings_clip, const float* grad_over_sum_xy,
                                       const float* grad_over_sum_xz, const float* grad_over_sum_yz,
                                       float* grad_embeddings, uint32_t n_rows, uint32_t F, void* stream);
/* (ABI v25) The plan straight from the occupancy grid — no vertex list, no sort by pixel.  The finest-level vertices
 * inside / one ring around occupied cells (utils_bpp_acc.py:498-512, the set get_idx_coords2 lists; resolution R = Rb t
 * + 2 <= 1024, Rb <= 128) are decided per line of vertices from the occupancy bit-packed along the line's axis.
 *   cnc_vote_plan_count: occupancy [Rb, Rb, Rb] bytes -> bits (scratch, 3 Rb^2 4 words, kept for the fill call) and
 *     counts [3, (R-2)^2] int32: per pixel of the xy / xz / yz plane the vertices of the set on its line, inner vertices
 *     only (the ones cnt_np_embed does not skip).
 *   cnc_vote_plan_fill: seg [3, (R-2)^2 + 1] = the caller's exclusive running sums of the counts; writes rows_* [n] =
 *     the table rows in pixel-major order of each plane (what a stable sort by pixel of the (x, y, z)-ordered list
 *     gives) and xyz [n] = the vertices x | y << 10 | z << 20 in (x, y, z) order.  Sorting xyz by row (rows_xy is that
 *     key) gives the backward plan;
 *   cnc_cnt_np_embed_planned_backward3_xyz = ..._backward3 reading one packed vertex per entry instead of three pixels. */
int cnc_vote_plan_count(const uint8_t* occupancy, uint32_t Rb, uint32_t t, uint32_t* bits, int32_t* counts, void* stream);
int cnc_vote_plan_fill(const uint32_t* bits, uint32_t Rb, uint32_t t, uint32_t hashmap_size, const int32_t* seg,
                       int32_t* rows_xy, int32_t* rows_xz, int32_t* rows_yz, uint32_t* xyz, void* stream);
int cnc_cnt_np_embed_planned_backward3_xyz(const uint32_t* xyz_by_row, const int32_t* row_seg,
                                           const float* embeddings_clip, const float* grad_over_sum_xy,
                                           const float* grad_over_sum_xz, const float* grad_over_sum_yz,
                                           float* grad_embeddings, uint32_t n_rows, uint32_t F, uint32_t resolution,
                                           void* stream);
int cnc_cnt_vote_masks(const float* embeddings_clip, uint32_t n_rows, uint32_t F, uint32_t* masks, void* stream);
int cnc_cnt_np_embed_planned_masked(const uint32_t* rows_by_pixel, const int32_t* pixel_seg, const uint32_t* masks,
                                    float* outputs, uint32_t n_pixels, uint32_t F, void* stream);

/* ------------------------------------------------------------------------------------------
 * Radiance-field MLP (gradient-free evaluations) — stands in for the cuBLAS GEMM chain behind
 * nn.Sequential(Linear, ReLU, Linear[, ReLU, Linear]) (examples/radiance_fields/ngp.py:475-504)
 * ---------------------------------------------------------------------------------------- */

/* Y[N, n_out] = W3 relu(W2 relu(W1 x + b1) + b2) + b3 (W3 == NULL: two layers), fp32 on
 * v_mfma_f32_16x16x4_f32, activations kept in LDS.  X [N, K0] with row stride ldx.  Weights are
 * passed PADDED: W_l [Hp_l, Kp_l] row-major zero-filled, b_l [Hp_l], Kp_0 = roundup16(K0),
 * Kp_l = Hp_{l-1}, Hp_l = roundup16(H_l) <= 160.  X may be a column window of a wider matrix: rows that are
 * 16-byte aligned with ldx >= Kp_0 are read Kp_0 floats at a time (into the row behind them), the matrix's last row only
 * up to K0.                                                                                    */
/* The same network on v_mfma_f32_32x32x2_f32: 32-row tiles, two of them per wave through the first
 * layer so that its weight fragments are fetched once per 64 rows.  Padding: Hp_l multiples of
 * 32, Kp_0 = roundup8(K0), Kp_l = Hp_{l-1}.  Only (160, 96) and (160, 160, 32) are instantiated (the
 * radiance field's base and head networks); anything else returns CNC_ERR_UNSUPPORTED.            */
int cnc_mlp_forward32(const float* X, uint32_t N, uint32_t ldx, uint32_t K0,
                      const float* W1, const float* b1, uint32_t H1p,
                      const float* W2, const float* b2, uint32_t H2p,
                      const float* W3, const float* b3, uint32_t H3p,
                      float* Y, uint32_t ldy, uint32_t n_out, void* stream);
int cnc_mlp_forward(const float* X, uint32_t N, uint32_t ldx, uint32_t K0,
                    const float* W1, const float* b1, uint32_t H1p,
                    const float* W2, const float* b2, uint32_t H2p,
                    const float* W3, const float* b3, uint32_t H3p,
                    float* Y, uint32_t ldy, uint32_t n_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Context-model aligner — replaces my_cuda_backen/aligner.cpp:4-79 (pack_and_align module)
 * ---------------------------------------------------------------------------------------- */

/* query_mask_3D (aligner.cpp:37-52, aligner_kernel.cu:4-80,161-242,328-367).
 *   points i16 [N, D] (D = 2 or 3), binary_vxl bool[Rb^D], mask i16 [N], overlap i32 [N].      */
int cnc_query_mask_3D(const int16_t* points, uint32_t D, const uint8_t* binary_vxl, uint32_t Rb,
                      int16_t* mask, int32_t* overlap, int32_t resolution, uint32_t N,
                      void* stream);

/* query_mask_3D_qlist (aligner.cpp:54-71, aligner_kernel.cu:82-158,244-326,370-409):
 * per-point resolution list i64 [N].                                                           */
int cnc_query_mask_3D_qlist(const int16_t* points, uint32_t D, const uint8_t* binary_vxl,
                            uint32_t Rb, int16_t* mask, int32_t* overlap,
                            const int64_t* resolution_list, uint32_t N, void* stream);

/* align_and_pack_forward (aligner.cpp:4-18, aligner_kernel.cu:413-495): ragged -> padded.
 *   feat [T, F] f32, cnt i64 [N], cumsum i64 [N+1] -> packed [N, M, F] f32, every element
 *   written (pad value V); `dim` only selected a launch shape in the reference and is ignored. */
int cnc_align_and_pack_forward(const float* feat, const int64_t* cnt, const int64_t* cumsum,
                               float* packed, uint32_t N, uint32_t M, uint32_t F, float V,
                               void* stream);

/* align_and_pack_backward (aligner.cpp:20-35, aligner_kernel.cu:498-565): dL_packed [N,M,F] ->
 * dL_feat [T, F]; rows not covered by any (i, j<cnt[i]) keep the caller's zero fill.           */
int cnc_align_and_pack_backward(const float* dL_packed, const int64_t* cnt, const int64_t* cumsum,
                                float* dL_feat, uint32_t N, uint32_t M, uint32_t F,
                                void* stream);

/* (extension) the reduction the reference always applies to a packed tensor, without packing:
 *   out[i][f] = sum_{r in [cumsum[i], cumsum[i+1])} w[r] * values[r][f]      mode 0
 *             ... / sum_r w[r]                                                mode 1
 *             ... / (cumsum[i+1]-cumsum[i])                                   mode 2
 * replaces align_and_pack -> (* weights) -> sum(dim=1) [-> / cnt] (utils_bpp_acc.py:564-566,
 * 668-669,681-682,688-695).  weights may be NULL (= 1).  values [T,F] f32, cumsum i64 [N+1].       */
int cnc_segment_weighted_sum(const float* values, const float* weights, const int64_t* cumsum,
                             float* out, uint32_t N, uint32_t F, int32_t mode, void* stream);
/* The same with the rows taken through a permutation: row r of the ragged list is values[order[r]] (weights stay
 * indexed by r).  Folds the `torch.index_select(mean, 0, order)` that sorts the per-vertex predictions by hash
 * slot (utils_bpp_acc.py:563) into the reduction.  order NULL = identity.                                        */
int cnc_segment_weighted_sum_gathered(const float* values, const int64_t* order, const float* weights,
                                      const int64_t* cumsum, float* out, uint32_t N, uint32_t F, int32_t mode,
                                      void* stream);

/* ------------------------------------------------------------------------------------------
 * Occupancy-grid marcher — replaces nerfacc/cuda/csrc/nerfacc.cpp:41-66 (grid.cu)
 * ---------------------------------------------------------------------------------------- */

/* ray_aabb_intersect (grid.cu:320-349,513-555; utils_grid.cuh:11-56).
 *   rays_o, rays_d [n_rays,3]; aabbs [n_aabbs,6] -> t_mins, t_maxs f32 [n_rays,n_aabbs],
 *   hits bool [n_rays,n_aabbs].                                                                */
int cnc_ray_aabb_intersect(const float* rays_o, const float* rays_d, const float* aabbs,
                           int32_t n_rays, int32_t n_aabbs, float near_plane, float far_plane,
                           float miss_value, float* t_mins, float* t_maxs, uint8_t* hits,
                           void* stream);

/* One launch of traverse_grids_kernel (grid.cu:68-318).  The reference's host function
 * (grid.cu:356-510) runs it twice (count, cumsum+alloc, fill) or once with over-allocation; the
 * allocation/cumsum stays on the host side of this ABI (cnc_amd/backends/nerfacc_cuda.py).
 * A segment group is "absent" when its chunk_cnts pointer is NULL (compute_intervals /
 * compute_samples == false).  first_pass != 0: only chunk_cnts are written.                    */
typedef struct {
    float*   vals;          /* [n_edges]  */
    int64_t* chunk_starts;  /* [n_rays]   */
    int64_t* chunk_cnts;    /* [n_rays]   */
    int64_t* ray_indices;   /* [n_edges]  */
    uint8_t* is_left;       /* [n_edges] or NULL */
    uint8_t* is_right;      /* [n_edges] or NULL */
    uint8_t* is_valid;      /* [n_edges] or NULL */
} cnc_ray_segments_t;       /* device-pointer view of RaySegmentsSpec, data_spec.hpp:6-13 */

int cnc_traverse_grids(const float* rays_o, const float* rays_d, const uint8_t* rays_mask,
                       int32_t n_rays,
                       const uint8_t* binaries, int32_t n_grids, int32_t resx, int32_t resy,
                       int32_t resz, const float* aabbs,
                       const uint8_t* hits, const float* t_sorted, const int64_t* t_indices,
                       const float* near_planes, const float* far_planes,
                       float step_size, float cone_angle, int32_t traverse_steps_limit,
                       int32_t first_pass,
                       const cnc_ray_segments_t* intervals, const cnc_ray_segments_t* samples,
                       float* terminate_planes, void* stream);

/* (extension) The march as the renderer consumes it: (t_start, t_end, ray) per sample, nothing else — what
 * OccGridEstimator.sampling / render_image_with_occgrid_test derive from traverse_grids' interval edges
 * (occ_grid.py:176-178, utils.py:408-410), 16 instead of 27 bytes per sample and no boolean indexing.
 * Same rays / grids / options as cnc_traverse_grids; same t values, same order.  Two calls:
 *   chunk_starts == NULL : count pass — chunk_cnts [n_rays] receives the samples per ray;
 *   chunk_starts != NULL : fill pass — t_starts / t_ends f32 [S], ray_indices i64 [S] are written at
 *                          chunk_starts[ray] (the caller's exclusive cumsum of chunk_cnts); rays whose
 *                          chunk_cnts is 0 are skipped.  With rays_mask + traverse_steps_limit the same
 *                          two calls give the iterative evaluation render its packed samples directly.
 * terminate_planes (nullable) [n_rays]: where each marched ray stopped (ask for it in the COUNT call).
 * resume_state (nullable, ABI v23) u32 [n_rays, 8]: scratch the count call fills with where each ray produced its
 *   first sample (grid segment, cell, the three next-crossing distances, t) and the fill call — given the same buffer —
 *   starts from; the fill call then also stops at the ray's last sample.  Same samples, same values: only the span
 *   between a ray's first and last sample is marched twice, not the empty space around it.
 * Fill-pass extras (ABI v24, all nullable): the marching lane has its ray's o and d in registers, so it can emit what
 *   the next pass would otherwise rebuild from (ray, t_start, t_end) per sample —
 *   positions f32 [S,3]: o + (d (t_start + t_end)) / 2 (rgb_sigma_fn, examples/utils.py:251-262), mapped to the unit
 *                       cube (p - min) / (max - min) of `aabb` (6 device floats, ngp.py:518-519) when aabb != NULL;
 *                       same operations in the same order as cnc_sample_positions: bit-equal;
 *   dirs f32 [S,3]: d of the sample's ray;  ray_indices32 i32 [S]: the ray id for internal consumers.
 *   ray_indices (i64, the nerfacc boundary) may be NULL when ray_indices32 or positions is given.             */
int cnc_march_samples(const float* rays_o, const float* rays_d, const uint8_t* rays_mask, int32_t n_rays,
                      const uint8_t* binaries, int32_t n_grids, int32_t resx, int32_t resy, int32_t resz,
                      const float* aabbs, const uint8_t* hits, const float* t_sorted,
                      const int64_t* t_indices, const float* near_planes, const float* far_planes,
                      float step_size, float cone_angle, int32_t traverse_steps_limit,
                      int64_t* chunk_cnts, const int64_t* chunk_starts, float* t_starts, float* t_ends,
                      int64_t* ray_indices, float* terminate_planes, uint32_t* resume_state,
                      float* positions, float* dirs, int32_t* ray_indices32, const float* aabb, void* stream);

/* (extension, ABI v27) Coarse occupancy for the march: one bit per block of 4 x 4 x 4 cells of the `binaries` byte grid
 * ([n_grids, resx, resy, resz], every res a multiple of 4) = "some cell of the block is set", bit (((g cx + x) cy + y) cz + z)
 * of the word array, c = res / 4.  cnc_occupancy_coarse_words: the array's length in 32-bit words, 0 when the march does
 * not take one for this shape (a resolution that is no multiple of 4, more than 2048 words).  cnc_march_samples_coarse =
 * cnc_march_samples with that array (nullable): a step of the march through an empty block is decided from LDS instead of
 * a dependent load of the cell's byte — same decisions, same samples, same values; the array must have been made from the
 * `binaries` the call is given.                                                                                   */
uint32_t cnc_occupancy_coarse_words(int32_t n_grids, int32_t resx, int32_t resy, int32_t resz);
int cnc_occupancy_coarse_bits(const uint8_t* binaries, int32_t n_grids, int32_t resx, int32_t resy, int32_t resz,
                              uint32_t* words, void* stream);
int cnc_march_samples_coarse(const float* rays_o, const float* rays_d, const uint8_t* rays_mask, int32_t n_rays,
                             const uint8_t* binaries, int32_t n_grids, int32_t resx, int32_t resy, int32_t resz,
                             const float* aabbs, const uint8_t* hits, const float* t_sorted,
                             const int64_t* t_indices, const float* near_planes, const float* far_planes,
                             float step_size, float cone_angle, int32_t traverse_steps_limit,
                             int64_t* chunk_cnts, const int64_t* chunk_starts, float* t_starts, float* t_ends,
                             int64_t* ray_indices, float* terminate_planes, uint32_t* resume_state,
                             float* positions, float* dirs, int32_t* ray_indices32, const float* aabb,
                             const uint32_t* coarse_bits, void* stream);

/* (extension) Sample positions for the field in one pass: positions[s] = o[ray] + d[ray] * t_a[s], or
 * o + (d * (t_a[s] + t_b[s])) / 2 when t_b != NULL (rgb_sigma_fn, examples/utils.py:251-262, same
 * evaluation order); aabb != NULL (6 floats on the device) maps them to the unit cube,
 * (p - min) / (max - min), as ngp.py:518-519.  dirs (may be NULL) receives d[ray].
 * rays_o, rays_d [n_rays,3]; ray_indices i64 [n_samples]; positions, dirs [n_samples,3].         */
int cnc_sample_positions(const float* rays_o, const float* rays_d, const int64_t* ray_indices,
                         const float* t_a, const float* t_b, const float* aabb, int64_t n_samples,
                         float* positions, float* dirs, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-ray segmented scans — replaces nerfacc/cuda/csrc/nerfacc.cpp:8-39 (scan.cu)
 * ---------------------------------------------------------------------------------------- */

/* inclusive_sum / exclusive_sum (scan.cu:9-128).  chunk_starts/cnts i64 [n_rays];
 * inputs/outputs f32 [n_edges].  backward != 0 scans each chunk right-to-left (scan.cu:42-55). */
int cnc_inclusive_sum(const int64_t* chunk_starts, const int64_t* chunk_cnts, const float* inputs,
                      float* outputs, uint32_t n_rays, int64_t n_edges, int32_t normalize,
                      int32_t backward, void* stream);
int cnc_exclusive_sum(const int64_t* chunk_starts, const int64_t* chunk_cnts, const float* inputs,
                      float* outputs, uint32_t n_rays, int64_t n_edges, int32_t normalize,
                      int32_t backward, void* stream);
/* inclusive_prod_forward / exclusive_prod_forward (scan.cu:130-170, 224-264). */
int cnc_inclusive_prod_forward(const int64_t* chunk_starts, const int64_t* chunk_cnts,
                               const float* inputs, float* outputs, uint32_t n_rays,
                               int64_t n_edges, void* stream);
int cnc_exclusive_prod_forward(const int64_t* chunk_starts, const int64_t* chunk_cnts,
                               const float* inputs, float* outputs, uint32_t n_rays,
                               int64_t n_edges, void* stream);
/* inclusive_prod_backward / exclusive_prod_backward (scan.cu:172-222, 266-304):
 * grad_inputs = reverse-scan-sum(grad_outputs * outputs) / max(inputs, 1e-10).                 */
int cnc_inclusive_prod_backward(const int64_t* chunk_starts, const int64_t* chunk_cnts,
                                const float* inputs, const float* outputs,
                                const float* grad_outputs, float* grad_inputs, uint32_t n_rays,
                                int64_t n_edges, void* stream);
int cnc_exclusive_prod_backward(const int64_t* chunk_starts, const int64_t* chunk_cnts,
                                const float* inputs, const float* outputs,
                                const float* grad_outputs, float* grad_inputs, uint32_t n_rays,
                                int64_t n_edges, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused per-ray volume rendering  —  replaces the ATen op chains of nerfacc/volrend.py and
 * nerfacc/pack.py (the reference has no native code here: SURVEY.md §7 item 5 asks for the fusion).
 * Samples are flattened and ray-ordered; chunk_starts / chunk_cnts i64 [n_rays] delimit each ray.
 * ---------------------------------------------------------------------------------------- */
#define CNC_VOLREND_ACCUMULATE 1u  /* colors/opacity/depth += this call's sums (iterative evaluation render,
                                      examples/utils.py:444-463: accumulate_along_rays_ x3)                  */
#define CNC_VOLREND_FINALIZE   2u  /* depth /= max(opacity, eps); colors += render_bkgd * (1 - opacity)
                                      (volrend.py:136-140); not with ACCUMULATE                              */

/* render_weight_from_density + accumulate_along_rays x3 (volrend.py:258-266,363 / :142-153,546 / :116-140):
 *   alpha_i = 1 - exp(-sigma_i dt_i),  T_i = exp(-sum_{j<i} sigma_j dt_j) [* (1 - opacity_in[ray])],  w_i = T_i alpha_i
 *   colors[ray] = sum w_i rgb_i,  opacity[ray] = sum w_i,  depth[ray] = sum w_i (t_start_i + t_end_i)/2
 * The running sum keeps the float association of exclusive_sum (scan.cu) — weights / trans / alphas equal the op
 * chain's; the per-ray sums have no defined order in the reference (atomics).
 * Nullable: rgbs (then colors must be NULL), opacity_in (per RAY: prefix transmittance = 1 - opacity_in,
 * utils.py:431-436 passes the same value per sample), render_bkgd [3], every output.                          */
int cnc_volrend_forward(const int64_t* chunk_starts, const int64_t* chunk_cnts, const float* t_starts,
                        const float* t_ends, const float* sigmas, const float* rgbs /* [S,3] */,
                        const float* opacity_in, const float* prefix_trans /* [S], nullable */,
                        const float* render_bkgd, float* weights, float* trans, float* alphas,
                        float* colors /* [n_rays,3] */, float* opacity, float* depth,
                        uint32_t n_rays, uint32_t flags, void* stream);
/* Backward of the above: dL/dsigma [S], dL/drgb [S,3] from dL/d(colors, opacity, depth) per ray (each
 * nullable = zero) and optionally dL/dweights, dL/dtrans, dL/dalphas per sample (the reference's op chain is
 * differentiable through all three).  weights / trans / alphas are the forward's outputs;
 * opacity / depth (the forward's FINAL outputs) are needed with CNC_VOLREND_FINALIZE only.               */
int cnc_volrend_backward(const int64_t* chunk_starts, const int64_t* chunk_cnts, const float* t_starts,
                         const float* t_ends, const float* rgbs, const float* weights, const float* trans,
                         const float* alphas, const float* opacity, const float* depth, const float* render_bkgd,
                         const float* grad_colors, const float* grad_opacity, const float* grad_depth,
                         const float* grad_weights, const float* grad_trans, const float* grad_alphas,
                         float* grad_sigmas, float* grad_rgbs, uint32_t n_rays, uint32_t flags, void* stream);
/* render_visibility_from_density / _from_alpha (volrend.py:425-475) as used by OccGridEstimator.sampling
 * (occ_grid.py:186-232): mask[s] = T_s >= early_stop_eps [&& alpha_s >= thre when thre > 0], thre =
 * min(alpha_thre, *alpha_thre_cap) with the cap read on the DEVICE (the reference's occs.mean().item());
 * kept [n_rays] (nullable) = survivors per ray.  from_alpha: values are alphas (t_* unused).              */
int cnc_render_visibility(const int64_t* chunk_starts, const int64_t* chunk_cnts, const float* t_starts,
                          const float* t_ends, const float* sigmas_or_alphas, int32_t from_alpha,
                          float early_stop_eps, float alpha_thre, const float* alpha_thre_cap, uint8_t* mask,
                          int64_t* kept, uint32_t n_rays, void* stream);
/* ray_indices[mask], t_starts[mask], t_ends[mask] (occ_grid.py:233-237) as one stable compaction:
 * survivor k of ray r goes to out_starts[r] + k (out_starts = exclusive cumsum of `kept`).               */
int cnc_compact_samples(const int64_t* chunk_starts, const int64_t* chunk_cnts, const int64_t* out_starts,
                        const uint8_t* mask, const float* t_starts, const float* t_ends, float* out_t_starts,
                        float* out_t_ends, int64_t* out_ray_indices, uint32_t n_rays, void* stream);
/* (extension) Depth windows for a sampler that evaluates the density front to back and stops behind opaque surfaces
 * (same survivors as evaluating every marched sample, volrend.py:425-475: a sample behind transmittance <
 * early_stop_eps is dropped whatever its density).
 *   cnc_ray_window_samples: samples [window_first[r], window_first[r] + window_cnts[r]) of ray r -> out_starts[r] + k
 *     (out_starts = exclusive cumsum of window_cnts), with out_source_index = their positions in the marched arrays;
 *   cnc_ray_transmittance : transmittance[r] = exp(-sum sigma dt) over the first chunk_cnts[r] samples of ray r.     */
int cnc_ray_window_samples(const int64_t* chunk_starts, const int64_t* window_first, const int64_t* window_cnts,
                           const int64_t* out_starts, const float* t_starts, const float* t_ends, float* out_t_starts,
                           float* out_t_ends, int64_t* out_ray_indices, int64_t* out_source_index, uint32_t n_rays,
                           void* stream);
int cnc_ray_transmittance(const int64_t* chunk_starts, const int64_t* chunk_cnts, const float* t_starts,
                          const float* t_ends, const float* sigmas, float* transmittance, uint32_t n_rays, void* stream);
/* (ABI v25) One step of that sampler as one kernel: done[r] += take[r] (the window just evaluated; first != 0: done = 0),
 * the ray goes on iff exp(-sum sigma dt over its first done[r] samples) >= threshold and done[r] < chunk_cnts[r] (first:
 * iff it has samples), take[r] = what it evaluates next: min(window, what is left), everything left for window < 0,
 * 0 for a ray that stopped.  done / take int64 [n_rays], updated in place.                                        */
int cnc_ray_window_next(const int64_t* chunk_starts, const int64_t* chunk_cnts, const float* t_starts,
                        const float* t_ends, const float* sigmas, int64_t* done, int64_t* take, int64_t window,
                        float threshold, int first, uint32_t n_rays, void* stream);
/* intervals.vals[is_left], intervals.vals[is_right], samples.ray_indices[is_valid] (occ_grid.py:176-178,
 * utils.py:408-410): the k-th left / right edge of a ray opens / closes its k-th sample.  iv_chunk_starts may be
 * the over-allocated layout of traverse_grids; out_starts [n_rays] = packed sample starts.                  */
int cnc_interval_edges_to_samples(const int64_t* iv_chunk_starts, const int64_t* iv_chunk_cnts,
                                  const float* iv_vals, const uint8_t* is_left, const uint8_t* is_right,
                                  const int64_t* out_starts, float* out_t_starts, float* out_t_ends,
                                  int64_t* out_ray_indices /* nullable */, uint32_t n_rays, void* stream);
/* pack_info (pack.py:11-49) on sorted ray_indices: first[r] / last[r] = first / one-past-last sample of
 * ray r; both [n_rays], zero-filled by the caller (rays without samples keep 0 / 0).                      */
int cnc_pack_bounds(const int64_t* ray_indices, int64_t n_samples, int64_t* first, int64_t* last,
                    int64_t n_rays, void* stream);

/* ------------------------------------------------------------------------------------------
 * Context-model heads and the Bernoulli rate  —  replace the ATen chains of examples/utils_bpp_acc.py
 * (context MLPs :378-393 applied at :561-566 / :689-692, hash fusion :567-572 / :693-701,
 * Bernoulli_entropy :1002-1013); SURVEY.md §7 item 6 / north_star: "the small context MLP ... as fused HIP
 * kernels".
 * ---------------------------------------------------------------------------------------- */

/* y = MLP([in_a | in_b | *pg]) row by row, no concatenated copy: in_a [N, Ca] (leading dimension lda), in_b
 * [N, Cb] or NULL, pg a DEVICE scalar appended as last column or NULL (with pg_index i64 [N]: a table, row i
 * takes pg[pg_index[i]] — rows of several levels in one call, each with its level's Pg); Ca + Cb + 1 <= 40.
 * n_layers 1: y = W1 x + b1 (the 2-D heads, Linear(C -> F));  n_layers 3: Linear(C,32) LeakyReLU(0.01)
 * Linear(32,32) LeakyReLU Linear(32,F) (context_model_3D).  Weights in nn.Linear layout [out, in].
 * F in {1,2,4,8}.  out [N, F].                                                                        */
int cnc_ctx_mlp_forward(const float* in_a, uint32_t lda, uint32_t Ca, const float* in_b, uint32_t ldb,
                        uint32_t Cb, const float* pg, const int64_t* pg_index, uint32_t N, uint32_t n_layers,
                        uint32_t F,
                        const float* W1, const float* b1, const float* W2, const float* b2,
                        const float* W3, const float* b3, float* out, void* stream);
/* Backward: grad_a [N, Ca] written; grad_b [N, Cb] written when non-NULL; *grad_pg and every weight / bias
 * gradient ACCUMULATED with atomics (the caller zero-fills them).  n_replicas (>= 1) zero-filled copies of the
 * weight / bias gradients, replica_stride floats apart (gW1 ... gb3 point into copy 0): workgroup b adds into copy
 * b % n_replicas and the caller sums the copies — ~1000 workgroups adding into the same few hundred addresses
 * serialise at the memory side.  (ABI v25) ldga / ldgb = row pitch of grad_a / grad_b in floats (0: packed, Ca /
 * Cb) — a caller that runs several heads on row ranges and column windows of one matrix (the three coded levels of a
 * plane, utils_bpp_acc.py:556-566) gets one gradient matrix for the encoder behind them.                    */
int cnc_ctx_mlp_backward(const float* in_a, uint32_t lda, uint32_t Ca, const float* in_b, uint32_t ldb,
                         uint32_t Cb, const float* pg, const int64_t* pg_index, uint32_t N, uint32_t n_layers,
                         uint32_t F,
                         const float* W1, const float* b1, const float* W2, const float* b2,
                         const float* W3, const float* b3, const float* grad_out, float* grad_a,
                         float* grad_b, float* grad_pg, float* gW1, float* gb1, float* gW2, float* gb2,
                         float* gW3, float* gb3, uint32_t n_replicas, uint32_t replica_stride,
                         uint32_t ldga, uint32_t ldgb, void* stream);
/* The per-step sample of the 3-D context pass (utils_bpp_acc.py:619-667), all coded levels at once: level i
 * contributes the vertices pos[i][0 .. p_at[i+1]-p_at[i]) (int16 triples, already offset to the window start)
 * and the slots cnt[i] / val[i][0 .. v_at[i+1]-v_at[i]).  Written, concatenated over the levels: pts i16 [P,3],
 * pts_n f32 [P,3] = (x - 0.5) / (res - 2), level_ids / resolutions i64 [P], slot_counts i64 [V],
 * table_rows i64 [V] = val + row0.  The descriptor lives on the HOST (device pointers inside).        */
typedef struct {
    const int16_t* pos[16];
    const int64_t* cnt[16];
    const int64_t* val[16];
    int64_t        p_at[17];
    int64_t        v_at[17];
    int64_t        row0[16];
    int32_t        level[16];
    int32_t        res[16];
    int32_t        n_win;
} cnc_ctx_window_t;
int cnc_ctx_window_gather(const cnc_ctx_window_t* win, int16_t* pts, float* pts_n, int64_t* level_ids,
                          int64_t* resolutions, int64_t* slot_counts, int64_t* table_rows, void* stream);

/* (ABI v28) The masked vertices of the 3-D context windows, compacted (utils_bpp_acc.py:680-690: `points_n[mask]`,
 * `n_list[mask] - L`, `clamp(overlap[mask], min = 1)`): for idx [M] i64 (the ascending indices of the vertices next to
 * occupied space) writes pts_m [M, 3] = pts_n[idx], level_m [M] = level_ids[idx], min_level [M] i32 = level_m - L and
 * (overlap_w != NULL) overlap_w [M] f32 = max(overlap[idx], 1).                                                    */
int cnc_ctx_compact(const int64_t* idx, const float* pts_n, const int64_t* level_ids, const int32_t* overlap, uint64_t M,
                    int32_t L, float* pts_m, int64_t* level_m, int32_t* min_level, float* overlap_w, void* stream);

/* (ABI v28) table[rows[s], 0..F) = values[s, 0..F) for n_rows DISTINCT rows (i64): the scatter behind
 * cnc_bernoulli_bits_backward's grad_x when the caller wants the table-shaped gradient (zero-filled by the caller).   */
int cnc_rows_scatter(const float* values, const int64_t* rows, float* table, uint64_t n_rows, uint32_t F, void* stream);

/* (ABI v29) The optimizer's update of the feature tables — torch.optim.Adam(lr, eps, weight_decay) as the reference steps it
 * (examples/train_CNC_nerf_synthetic.py:254-259,363: L2 decay into the gradient, no amsgrad) — for up to four tables in one
 * launch, with the step's gradient summed from up to four PIECES on the way in (what autograd left in `.grad`, the render
 * pass's scatter buffer, the entropy pass's, the planes' graph's static gradients): g = ((g0 + g1) + g2) + g3 in fp32, then
 * m += (1 - b1)(g + wd p - m); v = b2 v + (1 - b2) g^2; p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps),
 * scalar factors in double.  `step` = the count of THIS update (>= 1); table.step, when given, is the library optimizer's
 * own device-side counter (float32) and is incremented.  Pointers 16-byte aligned; a piece covers elements [g_lo, g_hi) of
 * its table, g_lo a multiple of 4, g_hi a multiple of 4 or n.  (ABI v31) the kernel can also leave the updated table's sign
 * bit plane and clip counter behind (sign_bits / clip_count below).                                                    */
typedef struct {
    float*       p;            /* [n] the table                                   */
    float*       m;            /* [n] exp_avg                                     */
    float*       v;            /* [n] exp_avg_sq                                  */
    float*       step;         /* device float32 step counter to increment, or NULL */
    const float* g[4];         /* gradient pieces, NULL = absent                  */
    uint64_t     g_lo[4];
    uint64_t     g_hi[4];
    uint64_t     n;
    /* ---- ABI v31 ---- */
    uint8_t*     sign_bits;    /* nullable (n a multiple of 8): receives cnc_pack_sign_bits' plane of the UPDATED table —
                                  byte i = the signs (>= 0) of elements 8 i .. 8 i + 7 — so that the next forward need not
                                  read the table again to make it                                                       */
    uint32_t*    clip_count;   /* nullable, with sign_bits: += the number of updated elements outside [-1, 1] (the
                                  caller zeroes it first), cnc_pack_sign_bits' counter                                  */
} cnc_adam_table_t;
typedef struct {
    cnc_adam_table_t table[4];
    uint32_t         n_tables;
    uint32_t         first_block[4];   /* filled by the library */
} cnc_adam_tables_t;
int cnc_table_adam(const cnc_adam_tables_t* tables, double lr, double beta1, double beta2, double eps, double weight_decay,
                   double step, void* stream);

/* (ABI v30) The front-to-back sampler's depth windows without a host round trip per window (nerfacc/estimators/occ_grid.py
 * `_density_front_to_back`; the reference evaluates sigma_fn on ALL marched samples at once, occ_grid.py:172-238).
 * cnc_ray_window_positions: samples [win_lo[r], win_lo[r] + win_n[r]) of ray r -> positions[o + k] = o_r + (d_r (t0 + t1)) / 2
 * (examples/utils.py:251-262, the arithmetic of cnc_sample_positions) and src[o + k] = their index in the sample stream, at
 * o = win_ends[r] - win_n[r] (win_ends = inclusive running sum of win_n, so win_ends[n_rays - 1] is the window's sample count,
 * on the device).  cnc_scatter_counted: out[src[i]] = values[i] for i < min(*n_dev, capacity).                        */
int cnc_ray_window_positions(const int64_t* chunk_starts, const int64_t* win_lo, const int64_t* win_n, const int64_t* win_ends,
                             const float* t_starts, const float* t_ends, const float* rays_o, const float* rays_d,
                             float* positions, int64_t* src, uint32_t n_rays, void* stream);
int cnc_scatter_counted(const float* values, const int64_t* src, float* out, const int64_t* n_dev, uint64_t capacity,
                        void* stream);

/* (ABI v25) Vertices of one 2-D level inside / one ring around the occupied cells of a projected occupancy plane
 * (utils_bpp_acc.py:431-456 `fetch_2D_batches`): cells [n_cells, 2] int32 = the occupied (i, j) of the plane, T =
 * (resolution - 2) / plane size; writes, cell-major then ring row / column, n_cells (T+2)^2 entries of rows (the
 * vertex's table row, examples/utils.py:492-511, int32) and points [., 2] = (vertex - 0.5) / (resolution - 2).   */
int cnc_plane_ring_vertices(const int32_t* cells, uint64_t n_cells, uint32_t T, uint32_t resolution,
                            uint64_t hashmap_size, int32_t* rows, float* points, void* stream);

/* bits = sum_{slot, f} -log2(p) (1 + x)/2 - log2(1 - p) (1 - x)/2, p = clamp(mean, 1e-6, 1 - 1e-6)
 * (utils_bpp_acc.py:1005-1013), x = table[rows[slot], f] (rows NULL: x = table[slot, f]).  The kernel writes
 * cnc_bernoulli_bits_partials(n_slots, F) per-block sums into `partial`; their sum is the result (summed by the
 * caller: deterministic).                                                                              */
uint32_t cnc_bernoulli_bits_partials(uint64_t n_slots, uint32_t F);
int cnc_bernoulli_bits_forward(const float* table, const int64_t* rows, const float* mean, uint64_t n_slots,
                               uint32_t F, float* partial, void* stream);
/* grad_mean [n_slots, F] and grad_x [n_slots, F] (each nullable) for d(bits) scaled by the device scalar
 * *grad_bits; the clamp passes the gradient where 1e-6 <= mean <= 1 - 1e-6.                            */
int cnc_bernoulli_bits_backward(const float* table, const int64_t* rows, const float* mean,
                                const float* grad_bits, uint64_t n_slots, uint32_t F, float* grad_mean,
                                float* grad_x, void* stream);
/* Gradient of cnc_segment_weighted_sum w.r.t. its values: grad_values [T, F] = grad[slot(t)] * scale_t,
 * scale = w_t (mode 0), w_t / wsum[slot] (mode 1; wsum = the forward applied to the weights), 1/count (mode 2). */
int cnc_segment_weighted_sum_backward(const float* grad, const int64_t* cumsum, const float* weights,
                                      const float* wsum, uint32_t n_slots, uint64_t T, uint32_t F,
                                      int32_t mode, float* grad_values, void* stream);
/* ... of cnc_segment_weighted_sum_gathered: the gradient of ragged row t is stored at grad_values[order[t]]
 * (order must be a permutation of [0, T): every row of grad_values is written exactly once).                  */
int cnc_segment_weighted_sum_gathered_backward(const float* grad, const int64_t* order, const int64_t* cumsum,
                                               const float* weights, const float* wsum, uint32_t n_slots,
                                               uint64_t T, uint32_t F, int32_t mode, float* grad_values,
                                               void* stream);

/* Level statistics of a binarised table, all levels in one pass (get_BiRF_wentropy_leveln,
 * utils_bpp_acc.py:472-486): sums[l] = sum of table[off[l]:off[l+1], :] (float64: exact for +-1 entries),
 * Pg[l] = #(+1) / numel, bits[l] = the zero-order bit count with the logarithms' arguments floored at 1e-9.
 * offsets_host: HOST array of n_levels + 1 row offsets (<= 32 levels).  sums f64 [n_levels] (scratch kept for
 * the backward), Pg / bits f32 [n_levels].                                                              */
int cnc_level_stats_forward(const float* table, const int64_t* offsets_host, uint32_t n_levels, uint32_t F,
                            double* sums, float* Pg, float* bits, void* stream);
/* grad_table [total_rows, F] = (grad_Pg dPg/dsum + grad_bits dbits/dsum)[level(row)], 0 outside the levels. */
int cnc_level_stats_backward(const double* sums, const int64_t* offsets_host, uint32_t n_levels, uint32_t F,
                             const float* grad_Pg, const float* grad_bits, uint64_t total_rows,
                             float* grad_table, void* stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise glue of the radiance field  —  replaces ATen chains of examples/radiance_fields/ngp.py
 * (no extension there).  Same float operations in the same order as the op chain.
 * ---------------------------------------------------------------------------------------- */

/* x_unit = (positions - aabb[:3]) / (aabb[3:] - aabb[:3]);  selector = all(0 < x_unit < 1)  (ngp.py:516-521).
 * positions, x_unit [N,3]; aabb 6 floats on the device; selector u8 [N].                             */
int cnc_field_prepare(const float* positions, const float* aabb, uint32_t N, float* x_unit,
                      uint8_t* selector, void* stream);
/* base_out [N, ld_base] = [density_raw | geo features (geo_feat_dim)]  ->
 *   density [N]        = exp(density_raw - 1) * selector                          (ngp.py:527-535; nullable)
 *   head_in [N, ld_head] = [SH degree-4 (16) of dirs | geo features | zeros]       (ngp.py:540-547; nullable)
 * dirs [N,3] are the raw view directions (the (dir + 1) / 2 and its inverse are applied inside, as the
 * reference and tiny-cuda-nn do between them).  selector nullable (= all ones).  ld_head a multiple of 4
 * and head_in 16-byte aligned (rows are written with 16-byte stores).
 * flags (ABI v24): CNC_FIELD_SH_FP16 rounds each of the 16 harmonics through IEEE half (round to nearest even)
 * before it is stored as float — tiny-cuda-nn's encoding writes a HALF tensor unless asked otherwise and the
 * reference asks for nothing (ngp.py:412-425), so the head MLP of the CUDA reference sees fp16-rounded inputs that
 * `cat` promoted back to float (ngp.py:540-547).  The directions carry no gradient: the backward is unchanged.  */
#define CNC_FIELD_SH_FP16 1u
int cnc_field_post(const float* base_out, uint32_t ld_base, uint32_t geo_feat_dim, const uint8_t* selector,
                   const float* dirs, uint32_t N, float* density, float* head_in, uint32_t ld_head,
                   uint32_t flags, void* stream);
/* ------------------------------------------------------------------------------------------
 * (extension, ABI v24) The gradient-free radiance field as one kernel: world positions -> density (-> rgb).
 * Replaces, for calls made without gradients, the whole chain of ngp.py:506-547 + compose_3D_2D_embed :620-645:
 * unit-cube mapping and selector, the four binarised grid encoders and the sinusoid embedding, base MLP
 * K0 -> H (ReLU) -> 1 + geo, density = exp(x - 1) * selector; with rgb != NULL also [SH4(dir) | geo] -> H -> H -> 3
 * and the sigmoid.  Features are computed into LDS and consumed by fp32 MFMA there: no [N, K0] matrix in HBM.
 * Values: the encoders' features are bit-identical to cnc_grid_encode_forward_bits; each layer is an exact fp32 fmaf
 * chain in k order (v_mfma_f32_32x32x2_f32), i.e. equal to the op chain up to the summation order of a GEMM.
 * ---------------------------------------------------------------------------------------- */
/* (ABI v27) The forward of the GRADIENT pass as the same one kernel: with `save.feat` set, cnc_field_fused_forward
 * (CNC_FIELD_TWO_WAVES, rgb != NULL) also writes everything the backward reads — what the op chain's autograd would have
 * kept (ngp.py:506-547: the MLPs' inputs and ReLU outputs) plus the inputs of the four encoders' backward — so that a
 * training step's render pass is: this kernel, cnc_field_backward_chain, cnc_field_weight_grads, the encoders' backward.
 * All matrices row-major float32 with N rows, N = the call's N (the caller's bucketed row count); positions / dirs are
 * read for rows < n_live only, rows [n_live, N) are evaluated as a point outside the box (zero grid features, selector
 * 0, density 0): what is stored for them is finite, and they receive zero gradient.
 * No exact-fp32 kernel runs behind this call: a hidden activation beyond fp16's range SATURATES at 65504 (and
 * guard[0] = call_id reports it; the caller reads the word when it next synchronises and leaves the fused path).   */
typedef struct {
    float*   feat;        /* [N, ld_feat] the first layer's input: [grid features | x | sin/cos | zeros to ld_feat]   */
    uint32_t ld_feat;     /* >= roundup32(K0), a multiple of 4                                                       */
    float*   h1;          /* [N, H] relu(base.0)                                                                     */
    float*   h3;          /* [N, H] relu(head.0)                                                                     */
    float*   h4;          /* [N, H] relu(head.2)                                                                     */
    float*   head_in;     /* [N, ld_head] = [SH4 (16) | 0 | geo features | 0 ...]: column 16 is the slot the kernel
                             keeps for the raw density (always 0 here); head.0's weight gradient drops it              */
    uint32_t ld_head;     /* roundup32(17 + geo)                                                                     */
    float*   raw;         /* [N] density before the activation (base.2's output 0)                                   */
    uint8_t* selector;    /* [N]                                                                                     */
    float*   xyz;         /* [N, 3] unit-cube positions (cnc_field_prepare's x_unit)                                 */
    float*   xy;          /* [N, 2] their (x, y) / (x, z) / (y, z) pairs: the plane encoders' inputs                 */
    float*   xz;
    float*   yz;
    uint32_t n_live;      /* <= N                                                                                    */
} cnc_field_save_t;

typedef struct {
    const float*   aabb;               /* 6 floats on the device: the field's box (ngp.py:516-519)                */
    const uint8_t* bits[4];            /* sign bit planes (cnc_pack_sign_bits) of the xyz | xy | xz | yz tables    */
    const int32_t* offsets[4];         /* per encoder: level offsets [n_levels + 1]                               */
    const int32_t* resolutions[4];     /* per encoder: resolutions [n_levels]                                     */
    const float*   freqs;              /* [n_freqs] on the device: the Embedder's frequency bands (ngp.py:583-599) */
    const float*   packed_weights[5];  /* cnc_field_pack_layer of: base.0, base.2, head.0, head.2, head.4          */
    const float*   packed_biases[5];   /*   (entries 2..4 may be NULL for density-only calls)                      */
    const float*   w2_row0;            /* base.2.weight[0, :] padded to n_neurons (density-only calls)             */
    const void*    packed_weights16[5];/* cnc_field_pack_layer16 of the same five layers (CNC_FIELD_MFMA_F16X3)     */
    const uint32_t* units;             /* [sum n_levels][4] on the device, in feature-row order (xyz levels, then the
                                          xy, xz, yz planes' levels): {offsets[l], offsets[l+1] - offsets[l],
                                          resolutions[l], encoder 0..3} — the level tables as one record per unit   */
    uint32_t       n_levels[4];        /* the three planes must have the same number of levels                    */
    uint32_t       n_features;         /* F per level: 2, 4 or 8                                                  */
    uint32_t       n_freqs;            /* > 0 (the reference always embeds, ngp.py:433)                           */
    uint32_t       n_neurons;          /* H: 64 or 160                                                            */
    uint32_t       geo_feat_dim;       /* 1 + geo <= 64 (H = 64) / 96 (H = 160) and roundup8(16 + geo) <= H        */
    uint32_t       flags;              /* CNC_FIELD_SH_FP16 | CNC_FIELD_MFMA_F16X3 | CNC_FIELD_TWO_WAVES | ...      */
    /* ---- ABI v26 ---- */
    const void*    packed_weights16q[5];/* cnc_field_pack_all's 16x16x32 fragments (CNC_FIELD_TWO_WAVES)              */
    uint32_t*      guard;              /* 8 zero-initialised words on the device, owned by the caller for the life of
                                          the field: the fp16 range guard (below).  Required with CNC_FIELD_MFMA_F16X3 */
    uint32_t       call_id;            /* a number the caller increases with every call (> 0)                      */
    uint32_t       pack_id;            /* cnc_field_pack_t.pack_id of the pack that wrote the fragments in use      */
    float*         debug_features;     /* test hook (nullable): [N, debug_ld] floats receive the first layer's INPUT row of
                                          every sample exactly as the kernel computed it — the four encoders' features,
                                          the raw coordinates, the sinusoids, zero padding to a multiple of 32 — before
                                          it is split into halves.  CNC_FIELD_TWO_WAVES density-only calls only.        */
    uint32_t       debug_ld;           /* >= roundup32(K0)                                                         */
    /* ---- ABI v27 ---- */
    cnc_field_save_t save;             /* save.feat != NULL: the gradient pass's forward (above)                   */
    /* ---- ABI v30 ---- */
    const int64_t* n_rows_dev;         /* nullable: a row count ON THE DEVICE — the call evaluates min(N, *n_rows_dev) rows
                                          (N = the capacity of the buffers).  For callers that size a batch on the device
                                          and must not wait for the number (the front-to-back sampler's depth windows)     */
} cnc_fused_field_t;

/* The layers' products on the fp16 matrix pipe, three per term: every operand split x = hi + lo into two halves
 * (22 significand bits), x w ~= hi hi + hi lo + lo hi accumulated in fp32 (v_mfma_f32_32x32x16_f16): ~5e-7 relative
 * per term against fp32's 6e-8, at 1/5 of the matrix cycles of the exact fp32 form and on a pipe that overlaps with the
 * gather's vector work.  Without the flag: v_mfma_f32_32x32x2_f32, an exact fp32 fmaf chain per output.        */
#define CNC_FIELD_MFMA_F16X3 2u
/* (ABI v26) Two cooperating waves per 32-sample tile (csrc/field_fused2.hip): each owns half the output columns of a
 * layer — half the accumulators and weight registers of the one-wave kernel, so 3-4 waves per SIMD hide the latency of
 * the feature gathers.  Implies the three-product fp16 scheme; needs packed_weights16q.  CNC_FIELD_WAVES4 selects the
 * variant compiled for four waves per SIMD (128 registers) instead of three (168).                                */
#define CNC_FIELD_TWO_WAVES 4u
#define CNC_FIELD_WAVES4 8u
/* The fp16 range guard.  The three-product kernels split every operand into two halves; a hidden activation above
 * fp16's 65504, or a weight with |2^8 w| above it, would become inf / NaN silently.  Both are detected exactly, on the
 * device: a kernel that split such a value (or whose fragments cnc_field_pack_all stamped) writes call_id into guard[0],
 * and cnc_field_fused_forward enqueues the exact-fp32 kernel behind every fp16 launch — it returns at once unless
 * guard[0] == call_id, and recomputes the call otherwise.  No host synchronisation; the cost is one empty launch.   */

/* W [H, K] row-major (row stride ldw), b [H]  ->  Wp: n_ksteps * n_tiles * 256 floats in MFMA fragment order (float4
 * (kb * n_tiles + t) * 64 + lane = W[32 t + (lane & 31)][8 kb + 4 (lane >> 5) + 0..3], zero outside [H, K]);
 * Bp: n_tiles * 32 floats; row0 (nullable): W[0, :] zero-padded to row0_len floats.
 * Layer shapes for cnc_field_fused_forward (H = n_neurons, T = H / 32, T2 = 3 if H == 160 else 2, K0 = feature width):
 *   base.0: n_tiles T,  n_ksteps roundup32(K0) / 8        base.2: n_tiles T2, n_ksteps H / 8  (+ row0, row0_len H)
 *   head.0: n_tiles T,  n_ksteps roundup8(16 + geo) / 8    head.2: n_tiles T,  n_ksteps H / 8
 *   head.4: n_tiles 1,  n_ksteps H / 8                                                                          */
int cnc_field_pack_layer(const float* W, const float* b, uint32_t H, uint32_t K, uint32_t ldw, uint32_t n_tiles,
                         uint32_t n_ksteps, float* Wp, float* Bp, float* row0, uint32_t row0_len, void* stream);

/* The same layer for CNC_FIELD_MFMA_F16X3: Wp16 = n_ksteps16 * n_tiles * 1024 halves (per (K-step of 16, tile): 64 x 8
 * halves hi, 64 x 8 halves lo of 2^8 W[32 t + (lane & 31)][16 ks + 8 (lane >> 5) + 0..7]).  n_ksteps16: base.0
 * roundup32(K0) / 16, base.2 / head.2 / head.4 H / 16, head.0 roundup16(16 + geo) / 16 (<= H / 16).  Biases: those of
 * cnc_field_pack_layer.                                                                                        */
int cnc_field_pack_layer16(const float* W, uint32_t H, uint32_t K, uint32_t ldw, uint32_t n_tiles, uint32_t n_ksteps16,
                           void* Wp16, void* stream);

/* (ABI v26) All five layers (base.0, base.2, head.0, head.2, head.4) into every fragment order the fused kernels read,
 * in ONE launch.  Per layer: W [H, K] (row stride ldw), b [H]; Wp / Bp as cnc_field_pack_layer (n_tiles, n_ksteps);
 * Wp16 (nullable) as cnc_field_pack_layer16 (n_ksteps16); Wq16 (nullable): n_ksteps32 * n_colblocks * 1024 halves — per
 * (K-step of 32, column block of 16): 64 x 8 halves hi, then lo, of 2^8 W[16 cb + (lane & 15)][32 ks + 8 (lane >> 4)
 * + 0..7].  Column blocks for CNC_FIELD_TWO_WAVES: base.0 / head.0 / head.2: H / 16; base.2: 5 (H = 160) or 4 (H = 64);
 * head.4: 1.  K-steps of 32: roundup32(K) / 32 (head.0: roundup32(K + 1) / 32 with k_gap = 16).  row0 / row0_len: base.2's row 0 as
 * in cnc_field_pack_layer.
 * guard / pack_id: a layer holding a weight with |2^8 w| > 65504 gets guard[1 + layer] = pack_id (see the guard). */
typedef struct {
    const float* W;
    const float* b;
    uint32_t     H, K, ldw;
    uint32_t     n_tiles, n_ksteps, n_ksteps16, n_colblocks, n_ksteps32;
    float*       Wp;
    float*       Bp;
    void*        Wp16;
    void*        Wq16;
    uint32_t     k_gap;     /* Wq16 only: != 0 inserts a zero column at packed k = k_gap (the source columns from k_gap on
                               move one to the right).  head.0 takes 16: the two-wave kernel lays the head's input out as
                               [SH4 (16) | base output c at column 16 + c], and output 0 is the raw density            */
    uint32_t     flags;     /* Wq16 only.  CNC_PACK_TRANSPOSE: packed W'[out][k] = W[k][src_off + out] (W: K rows of ldw
                               floats) — the layers of the gradient chain (cnc_field_backward_chain) are the forward's
                               transposed; Wp / Bp / Wp16 / b must be NULL then (Wp == NULL skips the fp32 fragments for
                               any layer).  CNC_PACK_ZERO_FIRST: packed output 0 is all zero and output o >= 1 reads source
                               src_off + o - 1 (the raw density's slot in front of the geo features)                 */
    uint32_t     src_off;
} cnc_field_pack_layer_t;
#define CNC_PACK_TRANSPOSE 1u
#define CNC_PACK_ZERO_FIRST 2u
typedef struct {
    cnc_field_pack_layer_t layer[5];
    float*                 row0;
    uint32_t               row0_len;
    uint32_t*              guard;
    uint32_t               pack_id;
} cnc_field_pack_t;
int cnc_field_pack_all(const cnc_field_pack_t* desc, void* stream);

/* positions [N,3] (world), dirs [N,3] (nullable unless rgb), density [N], rgb [N,3] (nullable: density only).
 * CNC_ERR_UNSUPPORTED for shapes outside the table above (the caller then runs the unfused chain).            */
int cnc_field_fused_forward(const cnc_fused_field_t* field, const float* positions, const float* dirs, uint32_t N,
                            float* density, float* rgb, void* stream);

/* ------------------------------------------------------------------------------------------
 * (extension, ABI v26) The gradient chain of the field's two MLPs as one kernel: autograd through mlp_head and mlp_base of
 * NGPRadianceField_mygrid_2D3D (ngp.py:506-547) from (d loss / d rgb, d loss / d density) down to the gradient of the
 * base network's INPUT columns that belong to the grid encoders — sigmoid', Linear^T, ReLU', Linear^T, ReLU', Linear^T,
 * the geo split + trunc_exp's clamped derivative (ngp.py:318-334), Linear^T, ReLU', Linear^T — and, on the way, the
 * gradients with respect to every Linear's output (G5 .. G1): what the weight gradients dW_l = G_l^T A_l and the bias
 * gradients need.  Replaces five `g @ W` GEMMs, three ReLU-backward passes, cnc_field_post_backward and the sigmoid's
 * backward.  Arithmetic: three fp16 products per term with fp32 accumulation (as CNC_FIELD_MFMA_F16X3) on operands
 * scaled per 32-sample tile by a power of two (gradients are small); ~5e-7 relative per term.
 * All matrices row-major float32; 16-byte aligned rows (ld multiples of 4).  packed_weights_t: cnc_field_pack_all with
 * CNC_PACK_TRANSPOSE of, in this order: head.4 (H outputs, K = 3), head.2 (H, K = H), head.0's geo columns (1 + geo outputs:
 * CNC_PACK_ZERO_FIRST, src_off = 16; K = H; 5 (H = 160) or 4 column blocks), base.2 (H outputs, K = 1 + geo), base.0's
 * first n_enc_columns columns (n_enc_columns outputs in roundup16(n_enc_columns) / 16 column blocks, K = H).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t       N;                 /* rows                                                                         */
    uint32_t       n_neurons;         /* H: 64 or 160                                                                 */
    uint32_t       n_features;        /* F: 2, 4, 8 (names the instantiation; n_enc_columns is what is used)           */
    uint32_t       n_enc_columns;     /* encoder columns of the feature row (a multiple of 4, <= 192)                  */
    uint32_t       geo_feat_dim;
    uint32_t       ld_base, ld_g2, ld_x;
    const float*   grad_rgb;          /* [N, 3]  (nullable: zero)                                                     */
    const float*   grad_density;      /* [N]     (nullable: zero)                                                     */
    const float*   rgb;               /* [N, 3]  the forward's sigmoid output                                         */
    const float*   base_out;          /* [N, ld_base] base.2's output: only column 0 (raw density) is read; ld_base >= 1   */
    const uint8_t* selector;          /* [N]                                                                          */
    const float*   h1;                /* [N, H] relu(base.0), relu(head.0), relu(head.2): only their signs are read    */
    const float*   h3;
    const float*   h4;
    const void*    packed_weights_t[5];
    float*         G5;                /* [N, 4]     out: gradient w.r.t. head.4's output (column 3 = 0)               */
    float*         G4;                /* [N, H]     ... head.2's                                                      */
    float*         G3;                /* [N, H]     ... head.0's                                                      */
    float*         G2;                /* [N, ld_g2] ... base.2's (columns 0 .. geo; up to 16 * blocks zero-filled)     */
    float*         G1;                /* [N, H]     ... base.0's                                                      */
    float*         dX;                /* [N, ld_x]  out: columns [0, n_enc_columns) of the base network's input gradient */
    float*         bias_grads;        /* (nullable) [3 H + 84] ZERO-INITIALISED by the caller; receives (atomic adds) the column
                                         sums of G4 | G3 | G1 (H each) | G2 (80 slots) | G5 (4 slots): the bias gradients   */
    /* ---- ABI v27 ---- */
    uint32_t*      g_max;             /* (nullable) 5 ZERO-INITIALISED words; receive (atomic max) the float bits of
                                         max |G1|, max |G2|, max |G3|, max |G4|, max |G5|: cnc_field_weight_grads' scales   */
} cnc_field_bwd_t;
int cnc_field_backward_chain(const cnc_field_bwd_t* chain, void* stream);

/* (extension, ABI v27) The five weight gradients of the field's Linears, dW_l = G_l^T A_l  (autograd's LinearBackward,
 * ngp.py:506-547), from the gradient matrices cnc_field_backward_chain left in HBM and the layers' inputs — one kernel
 * that streams both once (a layer's operands are read by that layer only: the grid is split between the layers by bytes),
 * products on the fp16 matrix pipe in the three-product form with G_l scaled by a power of two taken from g_max, fp32
 * accumulation in registers over a workgroup's samples, + one reduction over the workgroups' partial sums.
 * Layer order: base.0, base.2, head.0, head.2, head.4.  Row-major float32, N rows:
 *   G[l] [N, ldG[l]]: columns [0, n_out[l]) (ldG % 4 == 0, >= roundup4(n_out); n_out <= 160)
 *   A[l] [N, ldA[l]]: ALL ldA[l] columns are read (ldA % 4 == 0, <= 256; columns that are padding must hold finite values)
 *   dW[l] [n_out[l], ld_dW[l]]: columns [0, n_in[l]) are written = input columns [0, n_in[l]) — for head.0 (l = 2) with
 *     input column head_gap_col skipped (cnc_field_save_t.head_in: 16; 0xFFFFFFFF: none)
 * workspace: cnc_field_weight_grads_workspace bytes (the workgroups' partial sums), no initialisation needed.
 * n_workgroups: 0 = one per CU.  Values: within ~1e-6 of the largest entry of each dW of the float64 product.        */
typedef struct {
    uint32_t        N;
    const float*    G[5];
    uint32_t        ldG[5];
    uint32_t        n_out[5];
    const float*    A[5];
    uint32_t        ldA[5];
    uint32_t        n_in[5];
    uint32_t        head_gap_col;
    float*          dW[5];
    uint32_t        ld_dW[5];
    const uint32_t* g_max;            /* cnc_field_bwd_t.g_max after the chain kernel, same stream                         */
    float*          workspace;
    uint64_t        workspace_bytes;
    uint32_t        n_workgroups;
} cnc_field_wgrad_t;
int cnc_field_weight_grads_workspace(const cnc_field_wgrad_t* d, uint64_t* bytes);
int cnc_field_weight_grads(const cnc_field_wgrad_t* d, void* stream);

/* STE_binary of ngp.py:22-39 over n floats (16-byte aligned buffers), one pass each way:
 *   forward : out = (c >= 0) * 1 + (c < 0) * -1 with c = clamp(x, -1, 1)   (+1 / -1; NaN -> 0)
 *   backward: grad_in = grad_out * (clamp(x, -1, 1) == x)                                                     */
int cnc_ste_binary_forward(const float* x, float* out, uint64_t n, void* stream);
int cnc_ste_binary_backward(const float* x, const float* grad_out, float* grad_in, uint64_t n, void* stream);

/* Backward of y = relu(...) [N, C] (C a multiple of 4, <= 256; contiguous, 16-byte aligned) and the bias gradient of
 * the Linear in front of it, in one pass: grad_in = y > 0 ? grad_out : 0 (aten::threshold_backward, nn.ReLU's own);
 * partial [cnc_relu_backward_bias_partials(N), C] receives per-workgroup column sums of grad_in — their sum over the
 * first axis is the bias gradient (replaces threshold_backward + sum(0): one read of the gradient less).        */
uint32_t cnc_relu_backward_bias_partials(uint32_t N);
int cnc_relu_backward_bias(const float* grad_out, const float* y, uint32_t N, uint32_t C, float* grad_in,
                           float* partial, void* stream);

/* out[i, col:ld] = [x_i (3) | sin(freqs[k] x_i) (3), cos(freqs[k] x_i) (3) for k < n_freqs | zeros]: the Embedder
 * of ngp.py:583-599 (include_input, periodic_fns = [sin, cos]) written into the base MLP's input matrix
 * (row stride ld, first column col; everything from col to ld is written).  x [N,3], freqs [n_freqs] on the
 * device.  arg = x * freq in float32, sinf / cosf: the values of torch.sin / torch.cos on the same device.    */
int cnc_field_sinusoid(const float* x, const float* freqs, uint32_t n_freqs, uint32_t N, float* out, uint32_t ld,
                       uint32_t col, void* stream);

/* grad_base_out [N, ld_base] (the row stride of base_out; columns past 1 + geo are zero-filled) from
 * grad_density [N] (nullable) and grad_head_in [N, ld_head] (nullable):
 * column 0 = grad_density * selector * exp(min(density_raw - 1, 15)) (trunc_exp's clamped gradient,
 * ngp.py:318-334), columns 1.. = grad_head_in[:, 16:16+geo].                                          */
int cnc_field_post_backward(const float* base_out, uint32_t ld_base, uint32_t geo_feat_dim,
                            const uint8_t* selector, const float* grad_density, const float* grad_head_in,
                            uint32_t ld_head, uint32_t N, float* grad_base_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CNC_HIP_H */
