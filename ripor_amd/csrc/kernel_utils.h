// Device helpers shared by the kernel translation units (t5_kernels.hip, tail_kernels.hip).
#pragma once
#include <hip/hip_fp16.h>

#include "common.h"

namespace rpr {

__device__ __forceinline__ void store_planes4(__half* out_h, size_t o_ps, size_t idx, float4 v, unsigned int* sat,
                                              float scale = A_PLANE_SCALE) {
  __half h[4], l[4];   // activation planes hold x * A_PLANE_SCALE (common.h)
  split_f16(v.x * scale, h[0], l[0], sat); split_f16(v.y * scale, h[1], l[1], sat);
  split_f16(v.z * scale, h[2], l[2], sat); split_f16(v.w * scale, h[3], l[3], sat);
  *reinterpret_cast<uint2*>(out_h + idx) = *reinterpret_cast<uint2*>(h);
  *reinterpret_cast<uint2*>(out_h + o_ps + idx) = *reinterpret_cast<uint2*>(l);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// One wave copies an embedding row into the residual stream; with the fused RMSNorm (XOut) it also emits the row's
// f16 planes and its fixed-point sum of squares (the wave owns the whole row: plain store, no atomic).
__device__ __forceinline__ void copy_row_x(const float4* __restrict__ src, float* __restrict__ out, int row, int d,
                                           int lane, const XOut& xo) {
  float4* dst = reinterpret_cast<float4*>(out + (size_t)row * d);   // only written when x_h == nullptr (fp32 mode)
  float ss = 0.f;
  for (int i = lane; i < (d >> 2); i += 64) {
    float4 v = src[i];
    if (xo.x_h) {
      const size_t idx = (size_t)row * d + 4 * (size_t)i;
      __half h[4], l[4];
      split_f16(v.x * X_PLANE_SCALE, h[0], l[0], xo.sat); split_f16(v.y * X_PLANE_SCALE, h[1], l[1], xo.sat);
      split_f16(v.z * X_PLANE_SCALE, h[2], l[2], xo.sat); split_f16(v.w * X_PLANE_SCALE, h[3], l[3], xo.sat);
      *reinterpret_cast<uint2*>(xo.x_h + idx) = *reinterpret_cast<uint2*>(h);
      *reinterpret_cast<uint2*>(xo.x_h + xo.x_ps + idx) = *reinterpret_cast<uint2*>(l);
      v = make_float4(x_from_planes(h[0], l[0]), x_from_planes(h[1], l[1]), x_from_planes(h[2], l[2]), x_from_planes(h[3], l[3]));
    } else {
      dst[i] = v;
    }
    ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (xo.ssq) {
    ss = wave_sum(ss);
    if (lane == 0) xo.ssq[row] = ssq_to_fix(ss);
  }
}

}  // namespace rpr
