// Diagnostic (GPU box): can the split-precision GEMM of one in-flight batch overlap with the HBM-bound decoder
// self-attention of another batch when they are launched on two HIP streams?
// Measured on MI355X (round 1): no. Plain streams: both = 22.0 ms vs 11.1 (GEMM) + 12.0 (attention) alone; the
// same with the attention kernel made persistent at 256..2048 resident blocks. With CU-masked streams (every 3rd
// CU for attention) the kernels do overlap (41.6 ms vs 14.7 + 37.6) but each side scales with its CU share
// (GEMM 263 TF/s on 171 CUs, attention 1.55 TB/s on 85 CUs), so partitioning loses against running them back to back.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/overlap_test.hip ripor_amd/csrc/gemm_h2.hip ripor_amd/csrc/t5_kernels.hip -o tools/overlap_test
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../ripor_amd/csrc/common.h"
using namespace rpr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void fill_half(__half* p, size_t n, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = __float2half(scale * (float)((i * 2654435761u >> 8) & 1023) / 1024.f - scale * 0.5f);
}
__global__ void fill_f32(float* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (float)((i * 2246822519u >> 9) & 1023) / 1024.f - 0.5f;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const int maxb = argc > 1 ? atoi(argv[1]) : 0;    // persistent attention: max resident blocks (0 = plain launch)
  const int M = 20480, N = 3072, K = 768, Q = 2048, B = 10, H = 12, L = 32, t = 27, inner = H * 64;
  CK(init_t5_kernel_attributes());
  __half *Ah, *Wh; float* C;
  CK(hipMalloc(&Ah, (size_t)2 * M * K * 2)); CK(hipMalloc(&Wh, (size_t)2 * N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 4));
  fill_half<<<(2 * (size_t)M * K + 255) / 256, 256>>>(Ah, (size_t)2 * M * K, 1.f);
  fill_half<<<(2 * (size_t)N * K + 255) / 256, 256>>>(Wh, (size_t)2 * N * K, 0.05f);
  const size_t kvn = (size_t)Q * H * L * B * 64;
  float *kc, *vc, *qb, *ob, *relb; uint16_t* anc; int32_t* bucket;
  CK(hipMalloc(&kc, kvn * 4)); CK(hipMalloc(&vc, kvn * 4)); CK(hipMalloc(&qb, (size_t)Q * B * inner * 4));
  CK(hipMalloc(&ob, (size_t)Q * B * inner * 4)); CK(hipMalloc(&relb, 32 * H * 4)); CK(hipMalloc(&anc, (size_t)Q * B * L * 2));
  CK(hipMalloc(&bucket, 64 * 4));
  fill_f32<<<(kvn + 255) / 256, 256>>>(kc, kvn); fill_f32<<<(kvn + 255) / 256, 256>>>(vc, kvn);
  fill_f32<<<((size_t)Q * B * inner + 255) / 256, 256>>>(qb, (size_t)Q * B * inner);
  CK(hipMemset(relb, 0, 32 * H * 4)); CK(hipMemset(bucket, 0, 64 * 4));
  {  // ancestry: slot (r*7 + p) % B
    std::vector<uint16_t> h((size_t)Q * B * L);
    for (size_t r = 0; r < (size_t)Q * B; ++r) for (int p = 0; p < L; ++p) h[r * L + p] = (uint16_t)((r * 7 + p) % B);
    CK(hipMemcpy(anc, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  }
  CK(hipDeviceSynchronize());
  GemmH2Args g{};
  g.A = Ah; g.a_ps = (size_t)M * K; g.lda = K; g.W = Wh; g.w_ps = (size_t)N * K; g.ldw = K;
  g.out[0] = g.out[1] = g.out[2] = C; g.ldo[0] = g.ldo[1] = g.ldo[2] = N; g.split_n = N; g.M = M; g.N = N; g.K = K;
  DecSelfAttnArgs a{qb, kc, vc, (size_t)L * B * inner, (size_t)L * B * 64, (size_t)B * 64, 64, anc, L, relb, bucket, ob, Q, B, H, t,
                    nullptr, 0};
  (void)maxb;   // (a persistent variant of the attention kernel limited to maxb resident blocks was also tried: same result)
  hipStream_t s1, s2;
  const int every = argc > 2 ? atoi(argv[2]) : 0;   // > 0: CU masks — attention gets every `every`-th CU, the GEMM the rest
  if (every > 0) {
    uint32_t m1[8], m2[8];
    for (int w = 0; w < 8; ++w) { m1[w] = 0; m2[w] = 0; }
    for (int i = 0; i < 256; ++i) { if (i % every == every - 1) m2[i / 32] |= 1u << (i % 32); else m1[i / 32] |= 1u << (i % 32); }
    CK(hipExtStreamCreateWithCUMask(&s1, 8, m1)); CK(hipExtStreamCreateWithCUMask(&s2, 8, m2));
  } else {
    CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  }
  const int NG = 40, NA = 16;
  auto gemms = [&](hipStream_t s) { for (int i = 0; i < NG; ++i) CK(launch_gemm_h2(g, s)); };
  auto attns = [&](hipStream_t s) { for (int i = 0; i < NA; ++i) CK(launch_dec_self_attn(a, s)); };
  gemms(s1); attns(s2); CK(hipDeviceSynchronize());
  double t0 = now(); gemms(s1); CK(hipDeviceSynchronize()); const double tg = now() - t0;
  t0 = now(); attns(s2); CK(hipDeviceSynchronize()); const double ta = now() - t0;
  t0 = now();
  for (int rep = 0; rep < 8; ++rep) {   // interleave the enqueue order so neither queue gets a head start
    for (int i = 0; i < NG / 8; ++i) CK(launch_gemm_h2(g, s1));
    for (int i = 0; i < NA / 8; ++i) CK(launch_dec_self_attn(a, s2));
  }
  CK(hipDeviceSynchronize()); const double tb = now() - t0;
  const double gfl = 2.0 * M * N * K * NG, abytes = (2.0 * Q * B * (t + 1) * inner * 4 + 2.0 * Q * B * inner * 4) * NA;
  printf("every=%d max_blocks=%d  GEMM alone %.2f ms (%.0f TF/s)  attention alone %.2f ms (%.2f TB/s)  both %.2f ms  (sum %.2f, max %.2f)\n",
         every, maxb, tg * 1e3, gfl / tg / 1e12, ta * 1e3, abytes / ta / 1e12, tb * 1e3, (tg + ta) * 1e3, (tg > ta ? tg : ta) * 1e3);
  return 0;
}
