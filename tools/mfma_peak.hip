// Diagnostic (GPU box): what does the chip sustain on v_mfma_f32_32x32x16_f16 alone, with no LDS / memory traffic,
// on random vs all-zero register operands? Separates "the GEMM is at the chip's MFMA power wall" from "the GEMM loses
// time around its MFMAs". Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_peak tools/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
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
  for (int mode = 0; mode < 3; ++mode) {   // 0 zeros, 1 random in [-1, 1), 2 random small (lo-plane-like, ~2^-11)
    for (size_t i = 0; i < n; ++i) {
      const float u = (float)rand() / RAND_MAX * 2.f - 1.f;
      h[i] = (_Float16)(mode == 0 ? 0.f : mode == 1 ? u : u * 4.8e-4f);
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
      printf("mode %d (%s) %d waves/SIMD: %.1f ms  %.0f TF/s f16 MFMA\n", mode, mode == 0 ? "zeros" : mode == 1 ? "random" : "random small",
             th / 256, ms, flops / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
