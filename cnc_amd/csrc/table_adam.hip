// table_adam.hip — the optimizer's update of the feature tables, with the step's gradient pieces summed on the way in.
//
// The reference steps torch.optim.Adam(lr, eps=1e-15, weight_decay) over every parameter (train_CNC_nerf_synthetic.py:254-259,
// 363).  For the four tables (12 x 2^19 x F + 3 x 4 x 2^17 x F floats: 180 MB at F = 8) the training step here holds the
// gradient in PIECES — what autograd accumulated in `.grad` (the rate terms through the STE), the render pass's scatter
// buffer, the entropy pass's scatter buffer, the planes' graph's static gradients (cnc_amd._gradsink, cnc_amd._planes_graph)
// — and used to add them up first: a clone, a multi-tensor add of two more table-sized operands, then the library's fused
// Adam reading the sum again (clone 0.36 GB + add 0.54 GB + Adam 1.26 GB per step).  This kernel reads the pieces where they
// lie: p, m, v and up to four gradient sources in, p, m, v out — 7 (to 8) table-sized streams instead of 12, one launch for
// the four tables instead of ~12, at the very end of the step where nothing else is left to overlap with.
//
// Arithmetic: torch's Adam, single-tensor form (torch/optim/adam.py `_single_tensor_adam`, L2 weight decay into the gradient,
// no amsgrad, no maximize), with the scalar factors in double as the library's fused kernel has them:
//     g    = ((g0 + g1) + g2) + g3                  (fp32, in the order the pieces used to be added to `.grad`)
//     g   += wd p
//     m    = m + (1 - b1) (g - m)                   (lerp)
//     v    = b2 v + (1 - b2) g g
//     p   -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// tests/test_gpu_table_adam.py holds it against torch.optim.Adam(fused=True) and the plain float64 formula.
#include "common.hpp"

namespace cnc {

struct AdamScalars {
    double lr_over_bc1, one_minus_b1, b2, one_minus_b2, bc2_sqrt, eps, wd;
};

__device__ __forceinline__ void adam_one(float& p, float& m, float& v, float g, const AdamScalars& s)
{
    if (s.wd != 0.0) g = (float)((double)g + (double)p * s.wd);
    m = (float)((double)m + s.one_minus_b1 * (double)(g - m));
    v = (float)(s.b2 * (double)v + s.one_minus_b2 * (double)g * (double)g);
    const double denom = sqrt((double)v) / s.bc2_sqrt + s.eps;
    p = (float)((double)p - s.lr_over_bc1 * (double)m / denom);
}

constexpr uint32_t kAdamThreads = 256, kAdamVec = 4, kAdamPerBlock = kAdamThreads * kAdamVec * 4;   // 4096 elements a block

__global__ __launch_bounds__(kAdamThreads) void k_table_adam(cnc_adam_tables_t a, AdamScalars s)
{
    // which table this block works on (block ranges are consecutive per table)
    uint32_t t = 0;
#pragma unroll
    for (uint32_t k = 1; k < 4; k++) t += (k < a.n_tables && blockIdx.x >= a.first_block[k]) ? 1u : 0u;
    const cnc_adam_table_t& T = a.table[t];
    const uint64_t          base = (uint64_t)(blockIdx.x - a.first_block[t]) * kAdamPerBlock;
    if (blockIdx.x == a.first_block[t] && threadIdx.x == 0 && T.step) *T.step += 1.0f;    // the optimizer's step count
    uint32_t clipped = 0;
#pragma unroll
    for (uint32_t r = 0; r < 4; r++) {
        const uint64_t i = base + ((uint64_t)r * kAdamThreads + threadIdx.x) * kAdamVec;
        if (i >= T.n) break;
        if (i + kAdamVec <= T.n) {
            float4 p = *reinterpret_cast<const float4*>(T.p + i);
            float4 m = *reinterpret_cast<const float4*>(T.m + i);
            float4 v = *reinterpret_cast<const float4*>(T.v + i);
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            bool   first = true;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                // a source covers [lo, hi) of the table (multiples of 4: whole rows of F >= 4 features, checked by the host)
                if (T.g[k] && i >= T.g_lo[k] && i < T.g_hi[k]) {
                    const float4 q = *reinterpret_cast<const float4*>(T.g[k] + (i - T.g_lo[k]));
                    if (first) g = q;
                    else { g.x += q.x; g.y += q.y; g.z += q.z; g.w += q.w; }
                    first = false;
                }
            }
            adam_one(p.x, m.x, v.x, g.x, s);
            adam_one(p.y, m.y, v.y, g.y, s);
            adam_one(p.z, m.z, v.z, g.z, s);
            adam_one(p.w, m.w, v.w, g.w, s);
            *reinterpret_cast<float4*>(T.p + i) = p;
            *reinterpret_cast<float4*>(T.m + i) = m;
            *reinterpret_cast<float4*>(T.v + i) = v;
            if (T.sign_bits) {
                // cnc_pack_sign_bits' byte: 8 consecutive elements = this lane's four and the next lane's (consecutive lanes
                // hold consecutive float4s; n is a multiple of 8 and a block starts at a multiple of 4,096, so the pair
                // never straddles a wave or the table's end)
                const uint32_t nib = (p.x >= 0 ? 1u : 0u) | (p.y >= 0 ? 2u : 0u) | (p.z >= 0 ? 4u : 0u) | (p.w >= 0 ? 8u : 0u);
                const uint32_t hi = (uint32_t)__shfl_down((int)nib, 1);
                if ((threadIdx.x & 1u) == 0) T.sign_bits[i >> 3] = (uint8_t)(nib | hi << 4);
                clipped += !(p.x >= -1.0f && p.x <= 1.0f) + !(p.y >= -1.0f && p.y <= 1.0f) + !(p.z >= -1.0f && p.z <= 1.0f) +
                           !(p.w >= -1.0f && p.w <= 1.0f);
            }
        } else {
            for (uint64_t j = i; j < T.n; j++) {
                float g = 0.f;
                bool  first = true;
                for (uint32_t k = 0; k < 4; k++)
                    if (T.g[k] && j >= T.g_lo[k] && j < T.g_hi[k]) {
                        const float q = T.g[k][j - T.g_lo[k]];
                        g = first ? q : g + q;
                        first = false;
                    }
                float p = T.p[j], m = T.m[j], v = T.v[j];
                adam_one(p, m, v, g, s);
                T.p[j] = p; T.m[j] = m; T.v[j] = v;
            }
        }
    }
    if (T.sign_bits && T.clip_count && clipped) atomicAdd(T.clip_count, clipped);      // rare: |p| > 1 after the update
}

}   // namespace cnc

extern "C" int cnc_table_adam(const cnc_adam_tables_t* tables, double lr, double beta1, double beta2, double eps,
                              double weight_decay, double step, void* stream)
{
    if (!tables || tables->n_tables == 0 || tables->n_tables > 4 || !(step >= 1.0)) return CNC_ERR_INVALID_VALUE;
    cnc_adam_tables_t a = *tables;
    uint64_t          blocks = 0;
    for (uint32_t t = 0; t < a.n_tables; t++) {
        const cnc_adam_table_t& T = a.table[t];
        if (!T.p || !T.m || !T.v || T.n == 0) return CNC_ERR_INVALID_VALUE;
        if (((uintptr_t)T.p | (uintptr_t)T.m | (uintptr_t)T.v) & 15) return CNC_ERR_INVALID_VALUE;
        if (T.sign_bits && (T.n & 7)) return CNC_ERR_INVALID_VALUE;
        for (uint32_t k = 0; k < 4; k++) {
            if (!T.g[k]) continue;
            const bool hi_ok = (T.g_hi[k] & 3) == 0 || T.g_hi[k] == T.n;
            if (T.g_lo[k] > T.g_hi[k] || T.g_hi[k] > T.n || (T.g_lo[k] & 3) || !hi_ok || ((uintptr_t)T.g[k] & 15))
                return CNC_ERR_INVALID_VALUE;
        }
        a.first_block[t] = (uint32_t)blocks;
        blocks += (T.n + cnc::kAdamPerBlock - 1) / cnc::kAdamPerBlock;
    }
    if (blocks > 0x7fffffffull) return CNC_ERR_INVALID_VALUE;
    cnc::AdamScalars s;
    const double     bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    s.lr_over_bc1 = lr / bc1;
    s.one_minus_b1 = 1.0 - beta1;
    s.b2 = beta2;
    s.one_minus_b2 = 1.0 - beta2;
    s.bc2_sqrt = sqrt(bc2);
    s.eps = eps;
    s.wd = weight_decay;
    hipLaunchKernelGGL(cnc::k_table_adam, dim3((uint32_t)blocks), dim3(cnc::kAdamThreads), 0, (hipStream_t)stream, a, s);
    return cnc::launch_status();
}
