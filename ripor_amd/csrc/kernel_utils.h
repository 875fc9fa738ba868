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

// x / d for a wave-uniform x by multiplication (SALU: s_mul_hi_u32) instead of the compiler's float-reciprocal sequence on the
// VALU: magic = 2^32 / d + 1 (host: div_magic), exact for x < 2^32 / d; d = 1 has no 32-bit magic
__device__ __forceinline__ int udiv_magic(unsigned x, int d, unsigned magic) { return d == 1 ? (int)x : (int)__umulhi(x, magic); }
static inline unsigned div_magic(int d) { return d <= 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)d) + 1u; }

// expf(x) for x <= 0 (or -inf, or NaN): the library's algorithm — 2^(x log2 e) with a two-term product, v_exp_f32 of the
// fraction, v_ldexp_f32 — without its overflow branch and with the underflow cut as the only select; same bits as expf.
__device__ __forceinline__ float exp_nonpos(float x) {
#pragma clang fp contract(off)   // ph - n must stay a subtraction of the ROUNDED product (contracted into an fma it is a different number)
  const float C = __uint_as_float(0x3fb8aa3bu), CL = __uint_as_float(0x32a5705fu), THR = __uint_as_float(0xc2ce8ed0u);
  const float ph = x * C;
  float t = fmaf(x, C, -ph);
  const float n = rintf(ph);
  t = fmaf(x, CL, t);
  const float r = (ph - n) + t;
  const float y = ldexpf(__builtin_amdgcn_exp2f(r), (int)n);
  return !(THR > x) ? y : 0.f;
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
    if (lane == 0) xo.ssq[row] = ssq_to_fix(ss, xo.sat, SSQ_ROW_CAP);
  }
}

}  // namespace rpr
