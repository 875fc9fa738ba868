// Diagnostic (GPU box): does the ORDER in which a wave's MFMAs reuse their register operands change what the chip
// sustains at its power cap? v_mfma_f32_32x32x16_f16 from registers only, random operands, 8 accumulators per wave,
// 4 A and 2 B fragments (the ping-pong GEMM's phase: 4 x 2 blocks), three issue orders:
//   0  A-major (the GEMM's order): (A0,B0) (A0,B1) (A1,B0) (A1,B1) ...   every other step changes both operands
//   1  Gray: (A0,B0) (A1,B0) (A2,B0) (A3,B0) (A3,B1) (A2,B1) (A1,B1) (A0,B1)   exactly one operand changes per step
//   2  both operands change on every step: (A0,B0) (A1,B1) (A2,B0) (A3,B1) (A0,B1) (A1,B0) (A2,B1) (A3,B0)
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_order tools/mfma_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(ai, bi, ci) acc[ci] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ai], b[bi], acc[ci], 0, 0, 0)
template <int ORDER>
__global__ __launch_bounds__(512, 2) void k(const _Float16* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  f16x8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f16x8*>(src + ((size_t)tid * 8 + i) * 8);
  for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const f16x8*>(src + ((size_t)tid * 8 + 4 + i) * 8);
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (ORDER == 0) { MF(0,0,0); MF(0,1,1); MF(1,0,2); MF(1,1,3); MF(2,0,4); MF(2,1,5); MF(3,0,6); MF(3,1,7); }
    if (ORDER == 1) { MF(0,0,0); MF(1,0,2); MF(2,0,4); MF(3,0,6); MF(3,1,7); MF(2,1,5); MF(1,1,3); MF(0,1,1); }
    if (ORDER == 2) { MF(0,0,0); MF(1,1,3); MF(2,0,4); MF(3,1,7); MF(0,1,1); MF(1,0,2); MF(2,1,5); MF(3,0,6); }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[tid] = s;
}
template <int ORDER> void run(const _Float16* d, float* o) {
  const int blocks = 256, th = 512, iters = 40000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<ORDER><<<blocks, th>>>(d, o, 2000); hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); k<ORDER><<<blocks, th>>>(d, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("order %d rep %d: %.2f ms  %.0f TF/s\n", ORDER, rep, ms, (double)blocks * 8 * iters * 8 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12);
  }
}
int main() {
  const size_t n = (size_t)256 * 512 * 64;
  std::vector<_Float16> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (_Float16)((float)rand() / RAND_MAX * 2.f - 1.f);
  _Float16* d; float* o; hipMalloc(&d, n * 2); hipMalloc(&o, (size_t)256 * 512 * 4);
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
  for (int round = 0; round < 2; ++round) { run<0>(d, o); run<1>(d, o); run<2>(d, o); }
  return 0;
}
