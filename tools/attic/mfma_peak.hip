// Diagnostic (GPU box): what does the chip sustain on v_mfma_f32_32x32x16_f16 alone, with no LDS / memory traffic,
// on random vs all-zero register operands? Separates "the GEMM is at the chip's MFMA power wall" from "the GEMM loses
// time around its MFMAs". Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_peak tools/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512, 2) void mfma_kernel(const _Float16* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const f16x8*>(src + ((size_t)tid * 8 + i) * 8);
    b[i] = *reinterpret_cast<const f16x8*>(src + ((size_t)tid * 8 + 4 + i) * 8);
  }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[tid] = s;
}

int main(int argc, char** argv) {
  const int blocks = 256 * 1, threads = 512, iters = 20000;
  const size_t n = (size_t)blocks * threads * 64;
  std::vector<_Float16> h(n);
  _Float16* d; float* o;
  hipMalloc(&d, n * 2); hipMalloc(&o, (size_t)blocks * threads * 4);
  // 0 zeros, 1 random in [-1, 1), 2 random small (lo-plane-like, ~2^-11); 3.. : random with the low mantissa bits of
  // the B operands (modes 3-5: 3 / 6 / 9 bits) or of both operands (6: 5 bits each) cleared, 7: B = 0 — how much of the
  // power wall is operand toggling that a coarser lo plane could avoid?
  const char* names[8] = {"zeros", "random", "random small", "B low 3 bits clear", "B low 6 bits clear", "B low 9 bits clear",
                          "A and B low 5 bits clear", "B zeros"};
  for (int mode = 0; mode < 8; ++mode) {
    for (size_t i = 0; i < n; ++i) {
      const float u = (float)rand() / RAND_MAX * 2.f - 1.f;
      _Float16 v = (_Float16)(mode == 0 ? 0.f : mode == 2 ? u * 4.8e-4f : u);
      const bool is_b = ((i / 8) % 8) >= 4;          // per thread: 4 A vectors of 8 halves, then 4 B vectors
      unsigned short bits; memcpy(&bits, &v, 2);
      if (mode >= 3 && mode <= 5 && is_b) bits &= (unsigned short)~((1u << (3 * (mode - 2))) - 1u);
      if (mode == 6) bits &= (unsigned short)~31u;
      if (mode == 7 && is_b) bits = 0;
      memcpy(&v, &bits, 2);
      h[i] = v;
    }
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
    for (int waves = 0; waves < 2; ++waves) {
      const int th = waves == 0 ? 256 : 512;      // 1 or 2 waves per SIMD
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      mfma_kernel<8><<<blocks, th>>>(d, o, 1000);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      mfma_kernel<8><<<blocks, th>>>(d, o, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)blocks * (th / 64) * iters * 8 * 2.0 * 32 * 32 * 16;
      printf("mode %d (%s) %d waves/SIMD: %.1f ms  %.0f TF/s f16 MFMA\n", mode, names[mode], th / 256, ms, flops / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
