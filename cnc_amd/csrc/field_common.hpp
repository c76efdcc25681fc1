// field_common.hpp — device helpers shared by the radiance field's glue kernels (field_glue.hip) and the fused
// positions -> density / rgb kernel (field_fused.hip).
#pragma once

#include <hip/hip_fp16.h>

#include "common.hpp"

namespace cnc {

// four of the 16 terms of sh4 (same expressions): q = 0..3 -> terms 4q .. 4q+3
__device__ __forceinline__ float4 sh4_quad(uint32_t q, float x, float y, float z)
{
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    switch (q) {
    case 0: return make_float4(0.28209479177387814f, -0.48860251190291987f * y, 0.48860251190291987f * z,
                               -0.48860251190291987f * x);
    case 1: return make_float4(1.0925484305920792f * xy, -1.0925484305920792f * yz,
                               0.94617469575755997f * zz - 0.31539156525251999f, -1.0925484305920792f * xz);
    case 2: return make_float4(0.54627421529603959f * xx - 0.54627421529603959f * yy,
                               0.59004358992664352f * y * (-3.0f * xx + yy), 2.8906114426405538f * xy * z,
                               0.45704579946446572f * y * (1.0f - 5.0f * zz));
    default: return make_float4(0.3731763325901154f * z * (5.0f * zz - 3.0f), 0.45704579946446572f * x * (1.0f - 5.0f * zz),
                                1.4453057213202769f * z * (xx - yy), 0.59004358992664352f * x * (-xx + 3.0f * yy));
    }
}

// tiny-cuda-nn stores its encodings as half (CNC_FIELD_SH_FP16, include/cnc_hip.h)
__device__ __forceinline__ float round_through_half(float v) { return __half2float(__float2half_rn(v)); }

}  // namespace cnc
