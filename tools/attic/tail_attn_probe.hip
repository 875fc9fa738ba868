// A/B of the two generations of the fp32-MFMA tail attention kernels (tail_kernels.hip) at the bench's shape, outside
// the search: same synthetic inputs, outputs compared bit for bit, launches timed with hipEvents on an idle chip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ripor_amd/csrc -I include tools/tail_attn_probe.hip \
//         -L ripor_amd -lripor_hip -Wl,-rpath,'$ORIGIN/../ripor_amd' -o tools/tail_attn_probe
//   tools/tail_attn_probe [queries = 2150] [half]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "common.h"

namespace rpr {
extern int g_tail_attn_gen, g_tail_attn_opt, g_tail_cross_tpw;
}
using namespace rpr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <class T> static T* dev(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); return p; }
template <class T> static T* up(const std::vector<T>& v) { T* p = dev<T>(v.size()); CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return p; }

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned x = (unsigned)(i * 2654435761u) ^ seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    p[i] = ((float)(x & 0xffffff) / 16777216.0f - 0.5f) * 1.5f;
  }
}
static float* dev_random(size_t n, unsigned seed) { float* p = dev<float>(n); hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, p, n, seed); CK(hipDeviceSynchronize()); return p; }

static hipStream_t g_stream = 0;
template <class F> static double time_us(F&& f, int reps = 10) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) CK(f());
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, g_stream));
  for (int i = 0; i < reps; ++i) CK(f());
  CK(hipEventRecord(e1, g_stream)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.0 / reps;
}

static bool same(const __half* a, const __half* b, size_t n, const char* what) {
  std::vector<__half> ha(n), hb(n);
  CK(hipMemcpy(ha.data(), a, n * sizeof(__half), hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), b, n * sizeof(__half), hipMemcpyDeviceToHost));
  const bool ok = memcmp(ha.data(), hb.data(), n * sizeof(__half)) == 0;
  size_t nz = 0; for (size_t i = 0; i < n; i += 97) nz += (__half2float(ha[i]) != 0.f);
  printf("  %-44s %s (%zu of %zu sampled values non-zero)\n", what, ok ? "bit-identical" : "DIFFERENT", nz, (n + 96) / 97);
  if (!ok) {
    size_t bad = 0, first = n;
    for (size_t i = 0; i < n; ++i) if (memcmp(&ha[i], &hb[i], 2)) { if (first == n) first = i; ++bad; }
    printf("    %zu differing values, first at %zu: %g vs %g\n", bad, first, __half2float(ha[first]), __half2float(hb[first]));
  }
  return ok;
}

// planes -> values, largest difference (kernels that sum in a different order)
static bool close_planes(const __half* a, const __half* b, size_t n_per_plane, const char* what, double tol) {
  std::vector<__half> ha(2 * n_per_plane), hb(2 * n_per_plane);
  CK(hipMemcpy(ha.data(), a, 2 * n_per_plane * sizeof(__half), hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), b, 2 * n_per_plane * sizeof(__half), hipMemcpyDeviceToHost));
  double worst = 0.0; size_t differ = 0;
  for (size_t i = 0; i < n_per_plane; ++i) {
    const double x = ((double)__half2float(ha[i]) + (double)__half2float(ha[n_per_plane + i])) / 16.0;
    const double y = ((double)__half2float(hb[i]) + (double)__half2float(hb[n_per_plane + i])) / 16.0;
    const double d = x > y ? x - y : y - x;
    if (d > 0) ++differ;
    if (!(d <= worst)) worst = d;
  }
  printf("  %-44s max |difference| %.3g (%zu of %zu values differ)%s\n", what, worst, differ, n_per_plane, worst <= tol ? "" : "  TOO LARGE");
  return worst <= tol;
}

int main(int argc, char** argv) {
  const int Q = argc > 1 ? atoi(argv[1]) : 2150, B = 10, H = 12, T = 4, L = 32, Lt = L - T, inner = H * DKV, Lq = 32, ND = 12;
  const int nseq = Q * B; const size_t rows = (size_t)nseq * Lt;
  printf("tail attention probe: %d queries x %d beams, T = %d, L = %d: %zu tail rows\n", Q, B, T, L, rows);
  std::mt19937 rng(1234);
  if (argc > 2 && !strcmp(argv[2], "half")) {   // the launches on a stream confined to the first half of the CUs, like a lane of the search
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t mask[32] = {0};
    for (int k = 0; k < cus / 2; ++k) mask[k >> 5] |= 1u << (k & 31);
    CK(hipExtStreamCreateWithCUMask(&g_stream, (uint32_t)((cus + 31) / 32), mask));
    printf("launches confined to %d of %d CUs\n", cus / 2, cus);
  }
  bool all_ok = true;
  unsigned int* sat = dev<unsigned int>(4); CK(hipMemset(sat, 0, 16));
  __half* out1 = dev<__half>(2 * rows * inner); __half* out2 = dev<__half>(2 * rows * inner);
  {  // ---------------------------------------------------------------- self-attention
    float* qkv = dev_random(rows * 3 * inner, 1);
    const size_t cache_n = (size_t)Q * T * B * inner;
    float* kc = dev_random(cache_n, 2); float* vc = dev_random(cache_n, 3);
    std::vector<uint16_t> anc((size_t)nseq * L); for (auto& x : anc) x = (uint16_t)(rng() % B);
    std::vector<int32_t> flist(Q); for (int i = 0; i < Q; ++i) flist[i] = i;
    std::vector<int> cnt{nseq};
    std::vector<float> bias(32 * H); for (auto& x : bias) x = ((int)(rng() % 2001) - 1000) / 500.0f;
    std::vector<int32_t> bucket(64); for (int i = 0; i < 64; ++i) bucket[i] = i < 16 ? i : 16 + (i - 16) / 4;
    TailSelfAttnArgs a{qkv, kc, vc, (size_t)T * B * inner, (size_t)T * B * DKV, (size_t)B * DKV, (size_t)DKV, up(anc), L, up(flist), up(cnt),
                       up(bias), up(bucket), nullptr, out1, rows * inner, sat, nseq, B, H, T, L};
    const double gb = ((double)rows * 4 * inner * 4 + 2.0 * nseq * T * inner * 4) / 1e9;
    printf("self-attention: %.2f GB algorithmic per launch\n", gb);
    auto run = [&](int gen, int opt, __half* out, const char* name) {
      g_tail_attn_gen = gen; g_tail_attn_opt = opt; a.out_h = out;
      CK(hipMemset(out, 0xff, 2 * rows * inner * sizeof(__half)));
      const double us = time_us([&] { return launch_tail_self_attn(a, g_stream); });
      printf("  %-44s %8.1f us  %6.2f TB/s\n", name, us, gb / us * 1e3);
    };
    run(1, 0, out1, "gen 1 (V through LDS, 2 waves / SIMD)");
    run(2, 0, out2, "gen 2 (scalar bases, V direct, 4 waves / SIMD)");
    all_ok &= same(out1, out2, 2 * rows * inner, "gen 2 vs gen 1");
    run(2, 1, out2, "gen 2 (launch bound 3 waves / SIMD)");
    all_ok &= same(out1, out2, 2 * rows * inner, "gen 2 (3) vs gen 1");
    CK(hipFree(qkv)); CK(hipFree(kc)); CK(hipFree(vc));
  }
  {  // ---------------------------------------------------------------- cross-attention
    std::vector<int32_t> last(Q), offs(Q), mask((size_t)Q * Lq, 0);
    size_t tp = 0;
    for (int q = 0; q < Q; ++q) {
      int len = 4 + (int)(rng() % 17); if (q % 97 == 0) len = 32; if (q % 101 == 0) len = 1;
      last[q] = len; offs[q] = (int32_t)tp; tp += len;
      for (int j = 0; j < len; ++j) mask[(size_t)q * Lq + j] = 1;
      if (q % 53 == 0 && len > 2) mask[(size_t)q * Lq + 1] = 0;   // a hole in the mask
    }
    const int xld = ND * 2 * inner;
    float* xkv = dev_random(tp * xld, 5);
    float* qb = dev_random(rows * inner, 6);
    std::vector<int> nq{Q};
    DecCrossAttnArgs a{qb, xkv + 3 * 2 * inner, xkv + 3 * 2 * inner + inner, xld, up(mask), nullptr, Q, B * Lt, H, Lq, out1, rows * inner,
                       up(last), up(offs), 0, sat, up(nq)};
    const double gb = ((double)rows * inner * 2 * 4 + 2.0 * tp * inner * 4) / 1e9;
    printf("cross-attention: %.2f GB algorithmic per launch (%zu encoder rows, mean %.1f per query)\n", gb, tp, (double)tp / Q);
    g_tail_attn_opt = 0;
    auto run = [&](int gen, int tpw, __half* out, const char* name) {
      g_tail_attn_gen = gen; g_tail_cross_tpw = tpw; a.out_h = out;
      CK(hipMemset(out, 0xff, 2 * rows * inner * sizeof(__half)));
      const double us = time_us([&] { return launch_tail_cross_attn(a, g_stream); });
      printf("  %-44s %8.1f us  %6.2f TB/s\n", name, us, gb / us * 1e3);
    };
    run(1, 0, out1, "gen 1 (wave per 32-row tile)");
    run(2, 1, out2, "gen 2, 1 tile per wave");
    all_ok &= same(out1, out2, 2 * rows * inner, "gen 2 tpw 1 vs gen 1");
    run(2, 3, out2, "gen 2, 3 tiles per wave, 3 waves / SIMD");
    all_ok &= same(out1, out2, 2 * rows * inner, "gen 2 tpw 3 vs gen 1");
    run(2, 9, out2, "gen 2, 9 tiles per wave, 3 waves / SIMD");
    all_ok &= same(out1, out2, 2 * rows * inner, "gen 2 tpw 9 vs gen 1");
    g_tail_attn_opt = 2;   // the other build of the tile loop
    run(2, 3, out2, "gen 2, 3 tiles per wave, Q prefetch, 2 waves / SIMD");
    all_ok &= same(out1, out2, 2 * rows * inner, "gen 2 tpw 3 occ 3 vs gen 1");
    run(2, 9, out2, "gen 2, 9 tiles per wave, Q prefetch, 2 waves / SIMD");
    all_ok &= same(out1, out2, 2 * rows * inner, "gen 2 tpw 9 occ 3 vs gen 1");
    g_tail_attn_opt = 0;
  }
  unsigned int hs = 0; CK(hipMemcpy(&hs, sat, 4, hipMemcpyDeviceToHost));
  printf("saturation word %u; %s\n", hs, all_ok ? "ALL IDENTICAL" : "MISMATCH");
  return all_ok ? 0 : 1;
}
