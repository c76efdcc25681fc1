// field_glue.hip — the elementwise work around the radiance field's two MLPs, one kernel per stage.
//
// The reference (examples/radiance_fields/ngp.py) spells it as ~60 ATen launches per field evaluation:
//   :516-521  x = (x - aabb_min) / (aabb_max - aabb_min);  selector = ((x > 0) & (x < 1)).all(-1)
//   :527-535  density_before_activation, base_mlp_out = split(h, [1, geo]);  density = trunc_exp(d - 1) * selector
//   :540-547  d = SH4((dir + 1) / 2)  [tiny-cuda-nn, restated in closed form];  h = cat([d, base_mlp_out])
//   :583-599  embed(x) = [x | sin(2^k x) | cos(2^k x)], k = 0..9  (the Embedder's loop over frequencies and functions)
// Here:  k_field_prepare  positions -> unit-cube positions + selector
//        k_field_sinusoid [x | sin(f_k x) | cos(f_k x) | zero padding] into columns of the base MLP's input matrix
//        k_field_post     base MLP output [N, 1 + geo] (+ view directions) -> density [N], head input [N, ld]
//                         ( = [SH4(dir) | geo features | zero padding], no cat, no split)
//        k_field_post_bwd gradients of density / head input -> gradient of the base MLP output
// Same float operations in the same order as the op chain (the library is built with contraction off), so
// the values are those of the unfused path.
#include "common.hpp"
#include "field_common.hpp"

namespace cnc {

__global__ __launch_bounds__(256) void k_field_prepare(const float* __restrict__ pos, const float* __restrict__ aabb,
                                                       uint32_t N, float* __restrict__ x_unit,
                                                       uint8_t* __restrict__ selector)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    bool in = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float v = (pos[(size_t)i * 3 + a] - aabb[a]) / (aabb[3 + a] - aabb[a]);
        x_unit[(size_t)i * 3 + a] = v;
        in = in && v > 0.0f && v < 1.0f;
    }
    selector[i] = in ? 1 : 0;
}

// [x (3) | for each frequency k: sin(f_k x) (3), cos(f_k x) (3) | zeros]: one lane per (row, unit), a unit being
// one input coordinate, one (frequency, coordinate) pair — sin and cos from ONE argument reduction (sincosf
// returns the values of sinf and cosf) — or one padding column
__global__ __launch_bounds__(256) void k_field_sinusoid(const float* __restrict__ x, const float* __restrict__ freqs,
                                                        uint32_t n_freqs, uint32_t N, float* __restrict__ out,
                                                        uint32_t ld, uint32_t col, uint32_t width)
{
    const uint32_t units = 3 + 3 * n_freqs + (width - 3 - 6 * n_freqs);
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (uint64_t)N * units) return;
    const uint32_t i = (uint32_t)(t / units), u = (uint32_t)(t % units);
    float* o = out + (size_t)i * ld + col;
    if (u < 3) {
        o[u] = x[(size_t)i * 3 + u];
    } else if (u < 3 + 3 * n_freqs) {
        const uint32_t k = (u - 3) / 3, a = (u - 3) % 3;
        float sn, cs;
        sincosf(x[(size_t)i * 3 + a] * freqs[k], &sn, &cs);
        o[3 + 6 * k + a] = sn;
        o[3 + 6 * k + 3 + a] = cs;
    } else {
        o[3 + 6 * n_freqs + (u - 3 - 3 * n_freqs)] = 0.0f;
    }
}

// STE_binary (ngp.py:22-39) in one pass each way.  forward: (x >= 0) * 1 + (x < 0) * -1 of clamp(x, -1, 1) (NaN -> 0);
// backward: grad * (clamp(x, -1, 1) == x).  The op chains are 4 and 5 ATen kernels over the whole table.
__global__ __launch_bounds__(256) void k_ste_binary_fwd(const float* __restrict__ x, float* __restrict__ out, uint64_t n)
{
    const uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 4 <= n) {
        const float4 v = *reinterpret_cast<const float4*>(x + i);
        float4       o;
        o.x = (v.x >= 0.0f ? 1.0f : 0.0f) + (v.x < 0.0f ? -1.0f : 0.0f);
        o.y = (v.y >= 0.0f ? 1.0f : 0.0f) + (v.y < 0.0f ? -1.0f : 0.0f);
        o.z = (v.z >= 0.0f ? 1.0f : 0.0f) + (v.z < 0.0f ? -1.0f : 0.0f);
        o.w = (v.w >= 0.0f ? 1.0f : 0.0f) + (v.w < 0.0f ? -1.0f : 0.0f);
        *reinterpret_cast<float4*>(out + i) = o;
    } else {
        for (uint64_t k = i; k < n; k++) out[k] = (x[k] >= 0.0f ? 1.0f : 0.0f) + (x[k] < 0.0f ? -1.0f : 0.0f);
    }
}

__global__ __launch_bounds__(256) void k_ste_binary_bwd(const float* __restrict__ x, const float* __restrict__ g,
                                                        float* __restrict__ out, uint64_t n)
{
    const uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 4 <= n) {
        const float4 v = *reinterpret_cast<const float4*>(x + i), gv = *reinterpret_cast<const float4*>(g + i);
        float4       o;
        o.x = gv.x * ((v.x >= -1.0f && v.x <= 1.0f) ? 1.0f : 0.0f);
        o.y = gv.y * ((v.y >= -1.0f && v.y <= 1.0f) ? 1.0f : 0.0f);
        o.z = gv.z * ((v.z >= -1.0f && v.z <= 1.0f) ? 1.0f : 0.0f);
        o.w = gv.w * ((v.w >= -1.0f && v.w <= 1.0f) ? 1.0f : 0.0f);
        *reinterpret_cast<float4*>(out + i) = o;
    } else {
        for (uint64_t k = i; k < n; k++) out[k] = g[k] * ((x[k] >= -1.0f && x[k] <= 1.0f) ? 1.0f : 0.0f);
    }
}

// ReLU backward and the bias gradient of the layer in one pass: g = y > 0 ? grad_out : 0 (aten::threshold_backward),
// partial[block][c] = sum of g[:, c] over the block's rows (summed by the caller: ~1000 blocks adding into C addresses
// would serialise).  One lane per (row lane, 4 columns): a row is read and written as consecutive 16-byte pieces.
__global__ __launch_bounds__(256) void k_relu_bwd_bias(const float* __restrict__ grad_out, const float* __restrict__ y,
                                                       uint32_t N, uint32_t C, uint32_t rows_per_block,
                                                       float* __restrict__ g, float* __restrict__ partial)
{
    __shared__ float s_part[256 * 4];
    const uint32_t Q = C / 4, R = 256 / Q;                  // column quads, row lanes
    const uint32_t q = threadIdx.x % Q, rl = threadIdx.x / Q;
    const uint32_t r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
    float4 acc = make_float4(0, 0, 0, 0);
    if (rl < R) {
        for (uint32_t r = r0 + rl; r < r1; r += R) {
            const size_t at = (size_t)r * C + 4 * q;
            const float4 go = *reinterpret_cast<const float4*>(grad_out + at), yy = *reinterpret_cast<const float4*>(y + at);
            float4       v;
            v.x = yy.x > 0.0f ? go.x : 0.0f;
            v.y = yy.y > 0.0f ? go.y : 0.0f;
            v.z = yy.z > 0.0f ? go.z : 0.0f;
            v.w = yy.w > 0.0f ? go.w : 0.0f;
            *reinterpret_cast<float4*>(g + at) = v;
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    *reinterpret_cast<float4*>(s_part + 4 * threadIdx.x) = acc;
    __syncthreads();
    if (threadIdx.x < C) {
        const uint32_t c = threadIdx.x;
        float          s = 0.0f;
        for (uint32_t k = 0; k < R; k++) s += s_part[4 * (k * Q + c / 4) + c % 4];
        partial[(size_t)blockIdx.x * C + c] = s;
    }
}

// real spherical harmonics up to degree 4 of d (field.SHEncoding, term by term)
__device__ __forceinline__ void sh4(float x, float y, float z, float (&o)[16])
{
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    o[0] = 0.28209479177387814f;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * zz - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.54627421529603959f * xx - 0.54627421529603959f * yy;
    o[9] = 0.59004358992664352f * y * (-3.0f * xx + yy);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * zz);
    o[12] = 0.3731763325901154f * z * (5.0f * zz - 3.0f);
    o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * zz);
    o[14] = 1.4453057213202769f * z * (xx - yy);
    o[15] = 0.59004358992664352f * x * (-xx + 3.0f * yy);
}


// one lane per (row, 4 columns of the head input): 16-byte stores, a row's lanes write its 4 * ld_head bytes back to
// back; rows of `base` are [density_raw | geo features].  (One lane per column, each evaluating all 16 harmonics into
// a runtime-indexed array, put that array into scratch memory: 218 us per 2^19 rows instead of ~25.)
__global__ __launch_bounds__(256) void k_field_post(const float* __restrict__ base, uint32_t ld_base, uint32_t geo,
                                                    const uint8_t* __restrict__ selector, const float* __restrict__ dirs,
                                                    uint32_t N, float* __restrict__ density, float* __restrict__ head_in,
                                                    uint32_t ld_head, uint32_t sh_fp16)
{
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t quads = head_in ? ld_head / 4 : 1;
    if (e >= (uint64_t)N * quads) return;
    const uint32_t i = (uint32_t)(e / quads), q = (uint32_t)(e % quads);
    if (q == 0 && density) {
        const float d = expf(base[(size_t)i * ld_base] - 1.0f);          // trunc_exp(x - 1)
        density[i] = d * (selector ? (float)selector[i] : 1.0f);
    }
    if (!head_in) return;
    float4 v;
    if (q < 4) {
        // the reference hands (dir + 1) / 2 to the encoding, which maps it back with * 2 - 1
        float d3[3];
#pragma unroll
        for (int a = 0; a < 3; a++) d3[a] = ((dirs[(size_t)i * 3 + a] + 1.0f) / 2.0f) * 2.0f - 1.0f;
        v = sh4_quad(q, d3[0], d3[1], d3[2]);
        if (sh_fp16) {      // tiny-cuda-nn stores the encoding as half; the reference's cat promotes it back
            v.x = round_through_half(v.x); v.y = round_through_half(v.y);
            v.z = round_through_half(v.z); v.w = round_through_half(v.w);
        }
    } else {
        const float*   b = base + (size_t)i * ld_base + 1;
        const uint32_t k = 4 * q - 16;
        v.x = k < geo ? b[k] : 0.0f;
        v.y = k + 1 < geo ? b[k + 1] : 0.0f;
        v.z = k + 2 < geo ? b[k + 2] : 0.0f;
        v.w = k + 3 < geo ? b[k + 3] : 0.0f;
    }
    *reinterpret_cast<float4*>(head_in + (size_t)i * ld_head + 4 * q) = v;
}

// d base[:, 0] = g_density * exp(min(x - 1, 15)) * selector (the clamped-gradient exp, ngp.py:318-334);
// d base[:, 1 + k] = g_head[:, 16 + k]
__global__ __launch_bounds__(256) void k_field_post_bwd(const float* __restrict__ base, uint32_t ld_base, uint32_t geo,
                                                        const uint8_t* __restrict__ selector,
                                                        const float* __restrict__ g_density,
                                                        const float* __restrict__ g_head, uint32_t ld_head, uint32_t N,
                                                        float* __restrict__ g_base)
{
    // grad_base_out has the row stride of base_out (ld_base >= 1 + geo); columns past 1 + geo get zeros
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t cols = 1 + geo;
    if (e >= (uint64_t)N * ld_base) return;
    const uint32_t i = (uint32_t)(e / ld_base), c = (uint32_t)(e % ld_base);
    float v = 0.0f;
    if (c == 0) {
        if (g_density) {
            const float s = selector ? (float)selector[i] : 1.0f;
            v = (g_density[i] * s) * expf(fminf(base[(size_t)i * ld_base] - 1.0f, 15.0f));
        }
    } else if (c < cols && g_head) {
        v = g_head[(size_t)i * ld_head + 16 + (c - 1)];
    }
    g_base[(size_t)i * ld_base + c] = v;
}

}  // namespace cnc

using namespace cnc;

extern "C" int cnc_field_prepare(const float* positions, const float* aabb, uint32_t N, float* x_unit,
                                 uint8_t* selector, void* stream)
{
    if (N == 0) return CNC_OK;
    if (!positions || !aabb || !x_unit || !selector) return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_field_prepare, dim3(div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, positions, aabb, N,
                       x_unit, selector);
    return launch_status();
}

extern "C" int cnc_ste_binary_forward(const float* x, float* out, uint64_t n, void* stream)
{
    if (n == 0) return CNC_OK;
    if (!x || !out || ((uintptr_t)x | (uintptr_t)out) % 16) return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_ste_binary_fwd, dim3((uint32_t)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, x, out, n);
    return launch_status();
}

extern "C" int cnc_ste_binary_backward(const float* x, const float* grad_out, float* grad_in, uint64_t n, void* stream)
{
    if (n == 0) return CNC_OK;
    if (!x || !grad_out || !grad_in || ((uintptr_t)x | (uintptr_t)grad_out | (uintptr_t)grad_in) % 16)
        return CNC_ERR_INVALID_VALUE;
    hipLaunchKernelGGL(k_ste_binary_bwd, dim3((uint32_t)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, x, grad_out,
                       grad_in, n);
    return launch_status();
}

static uint32_t relu_bwd_blocks(uint32_t N) { return min(div_up(N, 256u), 1024u); }

extern "C" uint32_t cnc_relu_backward_bias_partials(uint32_t N) { return N ? relu_bwd_blocks(N) : 0; }

extern "C" int cnc_relu_backward_bias(const float* grad_out, const float* y, uint32_t N, uint32_t C, float* grad_in,
                                      float* partial, void* stream)
{
    if (N == 0) return CNC_OK;
    if (!grad_out || !y || !grad_in || !partial || C == 0 || C % 4 || C > 256 ||
        ((uintptr_t)grad_out | (uintptr_t)y | (uintptr_t)grad_in) % 16)
        return CNC_ERR_INVALID_VALUE;
    const uint32_t blocks = relu_bwd_blocks(N);
    hipLaunchKernelGGL(k_relu_bwd_bias, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grad_out, y, N, C,
                       div_up(N, blocks), grad_in, partial);
    return launch_status();
}

extern "C" int cnc_field_sinusoid(const float* x, const float* freqs, uint32_t n_freqs, uint32_t N, float* out,
                                  uint32_t ld, uint32_t col, void* stream)
{
    if (N == 0) return CNC_OK;
    if (!x || !freqs || !out || col > ld || ld - col < 3 + 6 * n_freqs) return CNC_ERR_INVALID_VALUE;
    const uint32_t width = ld - col;       // the columns behind the embedding are the matrix's zero padding
    const uint64_t total = (uint64_t)N * (width - 3 * n_freqs);
    hipLaunchKernelGGL(k_field_sinusoid, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       freqs, n_freqs, N, out, ld, col, width);
    return launch_status();
}

extern "C" int cnc_field_post(const float* base_out, uint32_t ld_base, uint32_t geo_feat_dim, const uint8_t* selector,
                              const float* dirs, uint32_t N, float* density, float* head_in, uint32_t ld_head,
                              uint32_t flags, void* stream)
{
    if (N == 0) return CNC_OK;
    if (!base_out || ld_base < 1 + geo_feat_dim || (!density && !head_in)) return CNC_ERR_INVALID_VALUE;
    if (head_in && (!dirs || ld_head < 16 + geo_feat_dim || ld_head % 4 || (uintptr_t)head_in % 16)) return CNC_ERR_INVALID_VALUE;
    const uint64_t n = (uint64_t)N * (head_in ? ld_head / 4 : 1);
    hipLaunchKernelGGL(k_field_post, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, base_out,
                       ld_base, geo_feat_dim, selector, dirs, N, density, head_in, ld_head,
                       (flags & CNC_FIELD_SH_FP16) ? 1u : 0u);
    return launch_status();
}

extern "C" int cnc_field_post_backward(const float* base_out, uint32_t ld_base, uint32_t geo_feat_dim,
                                       const uint8_t* selector, const float* grad_density, const float* grad_head_in,
                                       uint32_t ld_head, uint32_t N, float* grad_base_out, void* stream)
{
    if (N == 0) return CNC_OK;
    if (!base_out || !grad_base_out || ld_base < 1 + geo_feat_dim) return CNC_ERR_INVALID_VALUE;
    if (grad_head_in && ld_head < 16 + geo_feat_dim) return CNC_ERR_INVALID_VALUE;
    const uint64_t n = (uint64_t)N * ld_base;
    hipLaunchKernelGGL(k_field_post_bwd, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, base_out,
                       ld_base, geo_feat_dim, selector, grad_density, grad_head_in, ld_head, N, grad_base_out);
    return launch_status();
}
