// field_fused_common.hpp — what the fused-field translation units (field_fused.hip: one wave per 32 samples;
// field_fused2.hip: two cooperating waves per 32 samples) share: the kernel arguments, the weight stream through a
// buffer resource, the per-unit gather, the chunk-tile rows and the window fill.
#pragma once
#include "common.hpp"
#include <cstdlib>

#include "encoder_common.hpp"
#include "field_common.hpp"

namespace cnc {

using f32x16 = __attribute__((ext_vector_type(16))) float;

typedef _Float16 half_t_;

struct FieldEnc {
    const uint8_t* bits;
    const int32_t* offsets;
    const int32_t* res;
    uint32_t       n_levels;
};

// The gradient pass's forward (k_field_fused16w2<.., SAVE>): everything the backward reads, written by the one kernel that
// computes it.  Row-major float32, `N` rows each (N = the bucketed row count: rows [n_live, N) are rows of padding,
// evaluated as a point outside the box — zero grid features, selector 0 — so that what is stored for them is finite).
struct FieldSave {
    float*   feat;        // [N, ld_feat] the first layer's input: [grid features | x | sinusoids | 0]
    uint32_t ld_feat;     // >= 32 * chunks, a multiple of 4
    float*   h1;          // [N, H] relu(layer 1)
    float*   h3;          // [N, H] relu(head layer 1)
    float*   h4;          // [N, H] relu(head layer 2)
    float*   head_in;     // [N, ld_head] [SH4 (16) | 0 | geo features | 0]: the kernel's own head-input layout
    uint32_t ld_head;     // roundup32(17 + geo)
    float*   raw;         // [N] density before its activation
    uint8_t* selector;    // [N]
    float*   xyz;         // [N, 3] unit-cube positions; [N, 2] their xy / xz / yz pairs (the four encoders' inputs)
    float*   xy;
    float*   xz;
    float*   yz;
    uint32_t n_live;      // positions / directions are read for rows < n_live only
};

struct FusedFieldArgs {
    const float* pos;
    const float* dirs;
    const float* aabb;
    uint32_t     N;
    FieldEnc     enc[4];          // xyz | xy | xz | yz
    const float* freqs;
    uint32_t     n_freqs;
    uint32_t     n_units;         // (encoder, level) units = sum of n_levels
    uint32_t     nkb1;            // K-steps of 8 of layer 1 (a multiple of 4: K padded to whole 32-column chunks)
    const float* Wp[5];           // packed weights (cnc_field_pack_layer)
    const float* Bp[5];           // padded biases
    const float* w2row;           // density only: W2[0, :] padded to NT * 32
    uint32_t     geo;
    uint32_t     nkbh;            // K-steps of the head's first layer: roundup8(16 + geo) / 8
    float*       density;
    float*       rgb;
    uint32_t     sh_fp16;
    const uint4* units;           // per unit {first row, rows, resolution, encoder} of its level (cnc_fused_field_t.units)
    const half_t_* Wp16[5];       // fp16 hi / lo fragments (cnc_field_pack_layer16), k_field_fused16
    uint32_t       nk16_1;        // K-steps of 16 of layer 1 (a multiple of 2)
    uint32_t       nk16_h;        // K-steps of 16 of the head's first layer: roundup16(16 + geo) / 16
    // fp16 range guard (cnc_fused_field_t.guard): word 0 = id of the last call in which a value left fp16's range
    // (written by the fp16 kernels, read by the exact-fp32 kernel launched behind them: it runs only then), words
    // 1..5 = "layer l has a weight with |256 w| > 65504" (cnc_field_pack_layer16*)
    uint32_t*      guard;
    uint32_t       call_id;
    uint32_t       pack_id;       // id of the cnc_field_pack_all that produced the fragments in use (> 0)
    uint32_t       nk32_h;        // K-steps of 32 of the head's first layer: roundup32(16 + geo) / 32
    uint32_t       only_if_flagged;   // exact-fp32 kernels: return at once unless guard[0] == call_id
    const half_t_* Wq16[5];       // fragments of the 16x16x32 form (cnc_field_pack_all), k_field_fused16w2
    float*         dbg_features;  // test hook (cnc_fused_field_t.debug_features): [N, dbg_ld] first-layer input rows
    uint32_t       dbg_ld;
    FieldSave      save;          // cnc_fused_field_t.save (feat != nullptr: the gradient pass's forward)
    const int64_t* n_dev;         // cnc_fused_field_t.n_rows_dev: N = min(N, *n_dev) (see `rows_of`)
};

// the number of rows a launch works on: the host's N, cut to a count the device holds (cnc_fused_field_t.n_rows_dev)
__device__ __forceinline__ uint32_t rows_of(const FusedFieldArgs& p)
{
    if (p.n_dev == nullptr) return p.N;
    const int64_t n = *p.n_dev;
    return n < 0 ? 0u : (n < (int64_t)p.N ? (uint32_t)n : p.N);
}

constexpr uint32_t kChunkPitch = 36;     // floats per row of the 32 x 32 chunk tile (+4: conflict-free b128 accesses)
constexpr uint32_t kPadH = 4;

// Weight fragments through a buffer resource: address = SGPR base + one VGPR (16 * lane) + a scalar K-step offset + an
// immediate per tile.  With flat pointers the compiler kept a 64-bit address pair per (layer, K-step, tile) alive across
// the persistent tile loop (hundreds of spilled registers); this way the whole weight stream costs one VGPR.
// (clang 22 / ROCm 7.2 lowers __builtin_amdgcn_raw_buffer_load_b128 to a ONE-dword load and splats it — checked in
// the ISA — so the intrinsic is declared by name, as composable_kernel does.)
typedef int32_t i32x4_t __attribute__((ext_vector_type(4)));
typedef float   f32x4_t __attribute__((ext_vector_type(4)));
using wrsrc_t = i32x4_t;
__device__ f32x4_t llvm_raw_buffer_load_f32x4(i32x4_t rsrc, int32_t voffset, int32_t soffset, int32_t aux)
    __asm("llvm.amdgcn.raw.buffer.load.v4f32");

__device__ __forceinline__ wrsrc_t weight_rsrc(const float* Wp, uint32_t bytes = 0x7FFFFFFFu)
{
    const uint64_t a = reinterpret_cast<uint64_t>(Wp);
    // base, stride 0, `bytes` of records (a read past them returns zero instead of touching memory), DATA_FORMAT 32
    // (the gfx9 raw-buffer word composable_kernel uses)
    return i32x4_t{(int32_t)(uint32_t)a, (int32_t)((uint32_t)(a >> 32) & 0xFFFFu), (int32_t)bytes, 0x00020000};
}

template <int NT>
__device__ __forceinline__ void load_w(wrsrc_t W, uint32_t kb, uint32_t lane, float4 (&dst)[NT])
{
    const int32_t soff = (int32_t)(kb * NT * 1024u);          // 64 lanes x 16 bytes per (K-step, tile)
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const f32x4_t v = llvm_raw_buffer_load_f32x4(W, (int32_t)(lane * 16u + t * 1024), soff, 0);
        dst[t] = make_float4(v.x, v.y, v.z, v.w);
    }
}

template <int NT>
__device__ __forceinline__ void mfma_step(const float4& a, const float4 (&w)[NT], f32x16 (&acc)[NT])
{
    // k-step outermost: consecutive MFMAs go to different accumulators
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w[t].x, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w[t].y, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w[t].z, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w[t].w, acc[t], 0, 0, 0);
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NT])
{
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int v = 0; v < 16; v++) acc[t][v] = 0;
}

// One wave's LDS writes followed by its own reads: DS operations of a wave execute in order, so only the compiler has
// to be kept from moving them (a workgroup fence would also wait for the weight prefetch in flight: vmcnt(0)).
__device__ __forceinline__ void wave_lds_order()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// acc (32 x NT*32, C layout) = A (32 x nkb*8, LDS row-major, pitch lda) * W^T
template <int NT>
__device__ __forceinline__ void layer_lds(const float* __restrict__ a_lds, uint32_t lda, uint32_t nkb,
                                          const float* __restrict__ Wp_, f32x16 (&acc)[NT], uint32_t lane)
{
    const wrsrc_t Wp = weight_rsrc(Wp_);
    const uint32_t i = lane & 31u, h = lane >> 5;
    zero_acc<NT>(acc);
    float4 wn[NT];
    load_w<NT>(Wp, 0, lane, wn);
    for (uint32_t kb = 0; kb < nkb; kb++) {
        const float4 a = *reinterpret_cast<const float4*>(a_lds + i * lda + kb * 8 + 4 * h);
        float4 w[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) w[t] = wn[t];
        if (kb + 1 < nkb) load_w<NT>(Wp, kb + 1, lane, wn);
        mfma_step<NT>(a, w, acc);
    }
}

// bias (+ ReLU), C layout -> row-major LDS: D[row = 8 (v >> 2) + 4 h + (v & 3)][col = 32 t + i]
template <bool RELU, int NT>
__device__ __forceinline__ void acc_to_lds(float* __restrict__ dst, uint32_t ld, const float* __restrict__ bias,
                                           const f32x16 (&acc)[NT], uint32_t lane)
{
    const uint32_t i = lane & 31u, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const float b = bias[t * 32 + i];
#pragma unroll
        for (int v = 0; v < 16; v++) {
            float x = acc[t][v] + b;
            if (RELU) x = x > 0 ? x : 0;
            dst[(8 * (v >> 2) + 4 * h + (v & 3)) * ld + t * 32 + i] = x;
        }
    }
}

// The F features of one (encoder, level) unit at a point: the body of k_grid_encode_fwd_bits (same corner order, same
// fmaf chain: bit-identical), no occupancy mask — in two halves, so that a lane can have the sign-plane gathers of
// BOTH units of its window in flight before it consumes either (two waves per SIMD do not hide an L2 round trip per
// unit: the gather alone ran at half the vector rate).
struct UnitGather {
    float    tw[8];       // weight / sum of the valid weights per corner, 0 for an invalid corner
    uint32_t rb[8];       // the corner rows' F sign bits
};

__device__ __forceinline__ UnitRec load_unit(const FusedFieldArgs& p, uint32_t u)
{
    const uint4 v = p.units[u];
    return UnitRec{v.x, v.y, v.z, v.w};
}

// ... or from the workgroup's copy of the table in LDS (field_fused2.hip): an LDS read instead of a memory round trip in
// front of every unit's index arithmetic
struct UnitTable {
    const FusedFieldArgs& p;
    const uint4*          lds;        // nullable: read through p.units
    __device__ __forceinline__ UnitRec operator()(uint32_t u) const
    {
        const uint4 v = lds ? lds[u] : p.units[u];
        return UnitRec{v.x, v.y, v.z, v.w};
    }
};

__device__ __forceinline__ const uint8_t* unit_bits(const FusedFieldArgs& p, uint32_t enc)
{
    const uint8_t* b = p.enc[0].bits;
    b = enc == 1 ? p.enc[1].bits : b;
    b = enc == 2 ? p.enc[2].bits : b;
    b = enc == 3 ? p.enc[3].bits : b;
    return b;
}

template <uint32_t D, uint32_t F>
__device__ __forceinline__ void unit_issue(const float (&x)[D], bool inside, const uint8_t* __restrict__ bits,
                                           const UnitRec& r, UnitGather& u)
{
    constexpr uint32_t C = 1u << D;
#pragma unroll
    for (uint32_t q = 0; q < 8; q++) { u.tw[q] = 0.0f; u.rb[q] = 0u; }
    if (!inside) return;
    const uint32_t off = r.off, hs = r.hs, R = r.R;
    Corners<D, false> c;
    c.setup(x, R, hs, 128u, nullptr);
#pragma unroll
    for (uint32_t q = 0; q < C; q++) {
        u.rb[q] = c.valid[q] ? load_row_bits<F>(bits, (uint64_t)off + c.row[q]) : 0u;
        u.tw[q] = c.valid[q] ? c.w[q] * c.wn_re : 0.0f;
    }
}

// ... and in one piece, for the colour variants (no room for a second unit's registers)
template <uint32_t D, uint32_t F>
__device__ __forceinline__ void unit_features(const float (&x)[D], bool inside, const uint8_t* __restrict__ bits,
                                              const UnitRec& r, float (&acc)[F])
{
    constexpr uint32_t C = 1u << D;
#pragma unroll
    for (uint32_t k = 0; k < F; k++) acc[k] = 0;
    if (!inside) return;
    const uint32_t off = r.off, hs = r.hs, R = r.R;
    Corners<D, false> c;
    c.setup(x, R, hs, 128u, nullptr);
    uint32_t rb[C];
#pragma unroll
    for (uint32_t q = 0; q < C; q++) rb[q] = c.valid[q] ? load_row_bits<F>(bits, (uint64_t)off + c.row[q]) : 0u;
#pragma unroll
    for (uint32_t q = 0; q < C; q++) {
        const float tw = c.valid[q] ? c.w[q] * c.wn_re : 0.0f;
#pragma unroll
        for (uint32_t k = 0; k < F; k++) {
            const float s = ((rb[q] >> k) & 1u) ? 1.0f : -1.0f;
            acc[k] = __builtin_fmaf(tw, s, acc[k]);
        }
    }
}

template <uint32_t C, uint32_t F>
__device__ __forceinline__ void unit_consume(const UnitGather& u, float (&acc)[F])
{
#pragma unroll
    for (uint32_t k = 0; k < F; k++) acc[k] = 0;
#pragma unroll
    for (uint32_t q = 0; q < C; q++) {
#pragma unroll
        for (uint32_t k = 0; k < F; k++) {
            const float s = ((u.rb[q] >> k) & 1u) ? 1.0f : -1.0f;
            acc[k] = __builtin_fmaf(u.tw[q], s, acc[k]);
        }
    }
}

// ---- the fp16 range guard --------------------------------------------------------------------------------------
// The three-product kernels split every operand into two halves; a magnitude above fp16's 65504 would turn into
// inf / NaN silently.  Features are bounded by construction (|feature| <= 1, raw coordinates clamped), so what can leave
// the range is a hidden activation or a weight.  Both are DETECTED, exactly and on the device: the kernels track the
// largest magnitude they split (one v_max per element) and the packer flags a layer with |2^8 w| > 65504; a call that
// saw either writes its id into guard[0], and the exact-fp32 kernel enqueued behind it — a few microseconds of an
// empty launch otherwise — recomputes that call.  No host synchronisation, no clamp, no false positives.
constexpr float kHalfMax = 65504.0f;

__device__ __forceinline__ bool guard_weights_flagged(const FusedFieldArgs& p, bool rgb)
{
    if (!p.guard) return false;
    bool f = p.guard[1] == p.pack_id;
    if (rgb) f = f || p.guard[2] == p.pack_id || p.guard[3] == p.pack_id || p.guard[4] == p.pack_id || p.guard[5] == p.pack_id;
    if (!f) return false;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(p.guard, p.call_id);
    return true;
}

__device__ __forceinline__ void guard_raise(const FusedFieldArgs& p, float mx)
{
    if (!p.guard) return;
    if (__any(!(mx <= kHalfMax)) && (threadIdx.x & 63u) == 0) atomicMax(p.guard, p.call_id);
}

// Where a sample's row of the 32 x 32 chunk tile lives.  Float tile: the A operand of the fp32 MFMA.  Half tile: two
// planes, x = hi + lo with hi = half(x), lo = half(x - hi) — 22 bits of x — the A operands of the three-product
// fp16 MFMA scheme (see k_field_fused16).
struct RowF32 {
    static constexpr bool kFastSin = false;
    static __device__ __forceinline__ float clamp_raw(float v) { return v; }
    float* row;
    template <uint32_t V>
    __device__ __forceinline__ void put(uint32_t col, const float (&v)[V]) const { store_vec<V>(row + col, v); }
    __device__ __forceinline__ void put1(uint32_t col, float v) const { row[col] = v; }
};

typedef _Float16 half_t;
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_half(float x, half_t& hi, half_t& lo)
{
    hi = (half_t)x;
    lo = (half_t)(x - (float)hi);
}

struct RowF16;
// RowF16 that also writes every value it is given, as float32, into a [N, ld] dump (the test hook that lets a parity test
// hold the kernel's OWN features against the oracle's encoder): `dbg` = the sample's dump row at this chunk's first column
template <typename Base>
struct RowDump : Base {
    float* dbg;
    template <uint32_t V>
    __device__ __forceinline__ void put(uint32_t col, const float (&v)[V]) const
    {
        Base::template put<V>(col, v);
        if (dbg) {
#pragma unroll
            for (uint32_t j = 0; j < V; j++) dbg[col + j] = v[j];
        }
    }
    __device__ __forceinline__ void put1(uint32_t col, float v) const
    {
        Base::put1(col, v);
        if (dbg) dbg[col] = v;
    }
};

// RowF16 that also writes the values as float32 into the saved feature matrix (FieldSave::feat), vector stores: `out` =
// the sample's row at this chunk's first column (16-byte aligned: ld_feat % 4 == 0), nullptr for a row that is not stored
template <typename Base>
struct RowSave : Base {
    float* out;
    template <uint32_t V>
    __device__ __forceinline__ void put(uint32_t col, const float (&v)[V]) const
    {
        Base::template put<V>(col, v);
        if (out) store_vec<V>(out + col, v);
    }
    __device__ __forceinline__ void put1(uint32_t col, float v) const
    {
        Base::put1(col, v);
        if (out) out[col] = v;
    }
};

struct RowF16 {
    static constexpr bool kFastSin = true;
    // a raw unit-cube coordinate of a sample far outside the box (selector 0: its density is exactly zero whatever
    // comes out) is clamped to +-2^14, inside fp16's range: its colour stays finite
    static __device__ __forceinline__ float clamp_raw(float v) { return fminf(fmaxf(v, -16384.0f), 16384.0f); }
    half_t* hi;
    half_t* lo;
    template <uint32_t V>
    __device__ __forceinline__ void put(uint32_t col, const float (&v)[V]) const
    {
        if constexpr (V == 4) {
            half4_t a, b;
#pragma unroll
            for (int j = 0; j < 4; j++) { half_t x, y; split_half(v[j], x, y); a[j] = x; b[j] = y; }
            *reinterpret_cast<half4_t*>(hi + col) = a;
            *reinterpret_cast<half4_t*>(lo + col) = b;
        } else if constexpr (V == 2) {
            half2_t a, b;
#pragma unroll
            for (int j = 0; j < 2; j++) { half_t x, y; split_half(v[j], x, y); a[j] = x; b[j] = y; }
            *reinterpret_cast<half2_t*>(hi + col) = a;
            *reinterpret_cast<half2_t*>(lo + col) = b;
        } else {
            put1(col, v[0]);
        }
    }
    __device__ __forceinline__ void put1(uint32_t col, float v) const
    {
        half_t x, y;
        split_half(v, x, y);
        hi[col] = x;
        lo[col] = y;
    }
};

// sin and cos of x in [0, 512] (a unit-cube coordinate times 2^k) on the hardware's v_sin_f32 / v_cos_f32 (arguments in
// revolutions) behind a two-term 1 / (2 pi) reduction: 8 instructions instead of ocml's ~80 for sincosf, 2.6e-7 from
// the float64 value where sincosf is 7e-8 (tools/sincos_probe.hip, 4 M arguments) — used by the fp16 kernels, whose
// products carry 5e-7 anyway; the exact-fp32 kernels keep sincosf.
__device__ __forceinline__ void fast_sincos(float x, float* s, float* c)
{
    const float hi = 0.15915494f, lo = 6.4206383e-09f;       // 1 / (2 pi) = hi + lo
    const float q = rintf(x * hi);
    float r = __builtin_fmaf(x, hi, -q);
    r = __builtin_fmaf(x, lo, r);
    *s = __builtin_amdgcn_sinf(r);
    *c = __builtin_amdgcn_cosf(r);
}

template <typename Row, uint32_t WC>
__device__ __forceinline__ void fill_tail(const FusedFieldArgs& p, const float (&xu)[3], uint32_t w0, uint32_t U, const Row& trow);

// Columns [w0, w0 + WC) of the feature row of one sample into its row of the chunk tile (`trow`, chunk-relative
// column w0 & 31).  Feature row = [units: n_units x F | x (3) | sin(f_k x) (3), cos(f_k x) (3) for k < n_freqs | 0 ...].
// WC = 16: one wave per tile (two lanes per sample); WC = 8: two waves per tile (four threads per sample).
template <uint32_t F, bool PAIR, typename Row, uint32_t WC = 16>
__device__ __forceinline__ void fill_window(const FusedFieldArgs& p, const float (&xu)[3], uint32_t w0, const Row& trow)
{
    static_assert(WC % F == 0 && (WC == 8 || WC == 16), "a window holds whole units");
    constexpr uint32_t B = PAIR ? 2u : 1u;            // units gathered before any is consumed
    constexpr uint32_t V = F < 4 ? F : 4;
    const uint32_t U = p.n_units * F;                 // first sinusoid column
    const bool in_x = xu[0] >= 0.0f && xu[0] <= 1.0f, in_y = xu[1] >= 0.0f && xu[1] <= 1.0f,
               in_z = xu[2] >= 0.0f && xu[2] <= 1.0f;
    if constexpr (!PAIR) {
#pragma unroll
        for (uint32_t s = 0; s < WC / F; s++) {
            const uint32_t u = (w0 + s * F) / F;
            if (u >= p.n_units) break;
            float a[F];
            const UnitRec  rec = load_unit(p, u);
            const uint8_t* bits = unit_bits(p, rec.enc);
            if (rec.enc == 0) {
                unit_features<3, F>(xu, in_x && in_y && in_z, bits, rec, a);
            } else {
                const uint32_t pl = rec.enc - 1;                                  // plane 0 = xy, 1 = xz, 2 = yz
                const float    x2[2] = {pl == 2 ? xu[1] : xu[0], pl == 0 ? xu[1] : xu[2]};
                const bool     in2 = (pl == 2 ? in_y : in_x) && (pl == 0 ? in_y : in_z);
                unit_features<2, F>(x2, in2, bits, rec, a);
            }
            const uint32_t o = (w0 + s * F) & 31u;
#pragma unroll
            for (uint32_t k = 0; k < F; k += V) {
                float v[V];
#pragma unroll
                for (uint32_t j = 0; j < V; j++) v[j] = a[k + j];
                trow.template put<V>(o + k, v);
            }
        }
    } else {
    // two units at a time: their gathers issued, then consumed (16 more registers: not in the colour variants, which
    // sit at the 256-register limit of two waves per SIMD)
#pragma unroll
    for (uint32_t s0 = 0; s0 < WC / F; s0 += B) {
        UnitGather ug[B];
        bool       is3[B], live[B];
#pragma unroll
        for (uint32_t j = 0; j < B; j++) { is3[j] = false; live[j] = false; }
#pragma unroll
        for (uint32_t j = 0; j < B; j++) {
            const uint32_t s = s0 + j;
            if (s >= WC / F) continue;
            const uint32_t u = (w0 + s * F) / F;
            if (u >= p.n_units) continue;
            live[j] = true;
            const UnitRec  rec = load_unit(p, u);
            const uint8_t* bits = unit_bits(p, rec.enc);
            if (rec.enc == 0) {
                is3[j] = true;
                unit_issue<3, F>(xu, in_x && in_y && in_z, bits, rec, ug[j]);
            } else {
                const uint32_t pl = rec.enc - 1;                                  // plane 0 = xy, 1 = xz, 2 = yz
                const float    x2[2] = {pl == 2 ? xu[1] : xu[0], pl == 0 ? xu[1] : xu[2]};
                const bool     in2 = (pl == 2 ? in_y : in_x) && (pl == 0 ? in_y : in_z);
                unit_issue<2, F>(x2, in2, bits, rec, ug[j]);
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < B; j++) {
            if (!live[j]) continue;
            float a[F];
            if (is3[j]) unit_consume<8, F>(ug[j], a);
            else unit_consume<4, F>(ug[j], a);
            const uint32_t o = (w0 + (s0 + j) * F) & 31u;
#pragma unroll
            for (uint32_t k = 0; k < F; k += V) {
                float v[V];
#pragma unroll
                for (uint32_t jj = 0; jj < V; jj++) v[jj] = a[k + jj];
                trow.template put<V>(o + k, v);
            }
        }
    }
    }
    fill_tail<Row, WC>(p, xu, w0, U, trow);
}

// The part of a window behind the units: raw coordinates, sinusoids, zero padding.  U = first such column.
template <typename Row, uint32_t WC>
__device__ __forceinline__ void fill_tail(const FusedFieldArgs& p, const float (&xu)[3], uint32_t w0, uint32_t U, const Row& trow)
{
    const uint32_t lo = w0 > U ? w0 : U, hi = w0 + WC;
    if (lo >= hi) return;
    const uint32_t n_sin = 3 + 6 * p.n_freqs;
    for (uint32_t col = lo; col < hi; col++) {
        const uint32_t e = col - U;
        if (e < 3) trow.put1(col & 31u, Row::clamp_raw(e == 0 ? xu[0] : (e == 1 ? xu[1] : xu[2])));
        else if (e >= n_sin) trow.put1(col & 31u, 0.0f);
    }
    // sin column e = 3 + 6 k + a, its cos column e + 3: ONE argument reduction for both (sincosf returns the values of
    // sinf and cosf); a pair that straddles two windows is evaluated by both lanes
    const uint32_t e_lo = lo - U, e_hi = hi - U;
    for (uint32_t e = e_lo > 6 ? e_lo - 3 : 3; e < e_hi && e < n_sin; e++) {
        const uint32_t k = (e - 3) / 6, r = (e - 3) - 6 * k;
        if (r >= 3) continue;
        const float xa = r == 0 ? xu[0] : (r == 1 ? xu[1] : xu[2]);
        float sn, cs;
        if constexpr (Row::kFastSin) fast_sincos(xa * p.freqs[k], &sn, &cs);
        else sincosf(xa * p.freqs[k], &sn, &cs);
        if (e >= e_lo) trow.put1((e + U) & 31u, sn);
        if (e + 3 >= e_lo && e + 3 < e_hi) trow.put1((e + 3 + U) & 31u, cs);
    }
}

// A whole window of WC columns behind the units (w0 >= U: raw coordinates, sinusoids, padding) as straight-line code:
// one column = one argument reduction, no loops, no per-column branches (the general `fill_tail` walks the columns and
// the sin / cos pairs in two data-dependent loops: ~2.5x the instructions for the two sinusoid chunks of a tile).
// Same values: fast_sincos / sincosf of the same float32 argument x_a * freq_k, the raw coordinate through clamp_raw.
template <typename Row, uint32_t WC>
__device__ __forceinline__ void fill_tail_window(const FusedFieldArgs& p, const float (&xu)[3], uint32_t w0, uint32_t U,
                                                 const Row& trow)
{
    const uint32_t n_sin = 3 + 6 * p.n_freqs, e0 = w0 - U;
    float v[WC];
#pragma unroll
    for (uint32_t j = 0; j < WC; j++) {
        const uint32_t e = e0 + j, t = e - 3u;                    // t wraps for the raw columns: masked below
        const uint32_t k = t / 6u, r = t - 6u * k, a = r >= 3u ? r - 3u : r;
        const bool     is_sin = e >= 3u && e < n_sin;
        const float    xa = a == 0 ? xu[0] : (a == 1 ? xu[1] : xu[2]);
        const float    fr = p.freqs[is_sin ? k : 0u];
        float sn, cs;
        if constexpr (Row::kFastSin) fast_sincos(xa * fr, &sn, &cs);
        else sincosf(xa * fr, &sn, &cs);
        const float raw = Row::clamp_raw(e == 0 ? xu[0] : (e == 1 ? xu[1] : xu[2]));
        v[j] = is_sin ? (r < 3u ? sn : cs) : (e < 3u ? raw : 0.0f);
    }
    constexpr uint32_t V = 4;
#pragma unroll
    for (uint32_t j = 0; j < WC; j += V) {
        float q[V];
#pragma unroll
        for (uint32_t i = 0; i < V; i++) q[i] = v[j + i];
        trow.template put<V>((w0 & 31u) + j, q);
    }
}

// The units of a window that holds only D-dimensional units (the caller knows: a wave-uniform fact, so the other
// dimension's code is not even issued).
// `bits_uniform`: the sign plane when the caller knows that every lane of the wave works on one encoder (nullptr: selected
// per lane from the unit's record)
template <uint32_t F, uint32_t D>
__device__ __forceinline__ void window_unit_issue(const FusedFieldArgs& p, const UnitTable& units, const float (&xu)[3],
                                                  uint32_t u, UnitFast& st, const uint8_t* bits_uniform = nullptr)
{
    const bool in_x = xu[0] >= 0.0f && xu[0] <= 1.0f, in_y = xu[1] >= 0.0f && xu[1] <= 1.0f,
               in_z = xu[2] >= 0.0f && xu[2] <= 1.0f;
    const UnitRec  rec = units(u);
    if constexpr (D == 3) {
        if (bits_uniform) unit_issue_fast<3, F>(xu, in_x && in_y && in_z, bits_uniform, rec, st);      // (two code paths: a
        else unit_issue_fast<3, F>(xu, in_x && in_y && in_z, unit_bits(p, rec.enc), rec, st);           // merged pointer is a vector one)
    } else {
        const uint32_t pl = rec.enc - 1;                                  // plane 0 = xy, 1 = xz, 2 = yz
        const float    x2[2] = {pl == 2 ? xu[1] : xu[0], pl == 0 ? xu[1] : xu[2]};
        const bool     in2 = (pl == 2 ? in_y : in_x) && (pl == 0 ? in_y : in_z);
        if (bits_uniform) unit_issue_fast<2, F>(x2, in2, bits_uniform, rec, st);
        else unit_issue_fast<2, F>(x2, in2, unit_bits(p, rec.enc), rec, st);
    }
}

template <uint32_t F, uint32_t D, typename Row>
__device__ __forceinline__ void window_unit_finish(const UnitFast& st, uint32_t col, const Row& trow)
{
    constexpr uint32_t V = F < 4 ? F : 4;
    float a[F];
    unit_finish_fast<D, F>(st, a);
#pragma unroll
    for (uint32_t k = 0; k < F; k += V) {
        float v[V];
#pragma unroll
        for (uint32_t j = 0; j < V; j++) v[j] = a[k + j];
        trow.template put<V>((col & 31u) + k, v);
    }
}

template <uint32_t F, uint32_t D, typename Row, uint32_t WC>
__device__ __forceinline__ void fill_units(const FusedFieldArgs& p, const UnitTable& units, const float (&xu)[3], uint32_t w0,
                                           const Row& trow, const uint8_t* bits_uniform = nullptr)
{
#pragma unroll
    for (uint32_t s = 0; s < WC / F; s++) {
        UnitFast st;
        window_unit_issue<F, D>(p, units, xu, (w0 + s * F) / F, st, bits_uniform);
        window_unit_finish<F, D, Row>(st, w0 + s * F, trow);
    }
}

// field_fused2.hip: the two-waves-per-tile kernels
int launch_field_fused_w2(const FusedFieldArgs& p, bool rgb, uint32_t F, uint32_t H, uint32_t waves_per_simd, hipStream_t s);

}  // namespace cnc
