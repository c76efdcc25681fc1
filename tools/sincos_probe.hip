// sincos_probe.hip — how far v_sin_f32 / v_cos_f32 behind a two-term 1/(2 pi) reduction are from ocml's sincosf on the
// arguments the field's sinusoid embedding sees (x in [0, 1] times 2^k, k = 0..9).   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void fast_sincos(float x, float* s, float* c)
{
    const float hi = 0.15915494f, lo = 6.4206383e-09f;       // 1 / (2 pi) = hi + lo
    const float q = rintf(x * hi);
    float r = __builtin_fmaf(x, hi, -q);
    r = __builtin_fmaf(x, lo, r);
    *s = __builtin_amdgcn_sinf(r);
    *c = __builtin_amdgcn_cosf(r);
}

__global__ void k(const float* x, float* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s0, c0, s1, c1;
    sincosf(x[i], &s0, &c0);
    fast_sincos(x[i], &s1, &c1);
    out[4 * i] = s0; out[4 * i + 1] = c0; out[4 * i + 2] = s1; out[4 * i + 3] = c1;
}

int main()
{
    const int n = 1 << 22;
    std::vector<float> h(n);
    unsigned s = 12345;
    for (int i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        const float u = (s >> 8) * (1.0f / 16777216.0f);
        h[i] = u * (float)(1 << (i % 10));
    }
    float *dx, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 16);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(dx, dout, n);
    std::vector<float> o(4 * n);
    hipMemcpy(o.data(), dout, n * 16, hipMemcpyDeviceToHost);
    double e_ocml = 0, e_fast = 0, e_diff = 0;
    for (int i = 0; i < n; i++) {
        const double rs = sin((double)h[i]), rc = cos((double)h[i]);
        e_ocml = fmax(e_ocml, fmax(fabs(o[4 * i] - rs), fabs(o[4 * i + 1] - rc)));
        e_fast = fmax(e_fast, fmax(fabs(o[4 * i + 2] - rs), fabs(o[4 * i + 3] - rc)));
        e_diff = fmax(e_diff, fmax(fabs(o[4 * i + 2] - o[4 * i]), fabs(o[4 * i + 3] - o[4 * i + 1])));
    }
    printf("max |err| vs float64: sincosf %.3g, v_sin/v_cos behind a two-term reduction %.3g; max |fast - sincosf| %.3g\n", e_ocml, e_fast, e_diff);
    return 0;
}
