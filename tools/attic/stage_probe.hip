// Probe (GPU box): what bounds the operand feed of a mid-size split-precision GEMM launch (640 x 3072 x 768: 480 blocks of
// 64 x 64 outputs, four waves per block splitting K, 393 KB of operand rows per block) — the transport or the bytes in flight?
// No arithmetic: every block pulls exactly the rows its GEMM block would (A panel shared by the column tiles, its own W panel,
// two f16 planes each), K-tiles of 32 columns (64-byte row pieces), wave w takes K-tiles w, w + 4, ...
//   MODE 0: LDS-DMA into a private ring of ST K-tiles per wave (what gemm_h2_wsplit_kernel does)
//   MODE 1: global_load_dwordx4 into registers, G K-tiles in flight, consumed by ds_write_b128 into one 16-KB buffer per wave
//   MODE 2: the same with 128-byte row pieces (K-tiles of 64 columns, G counted in 64-column tiles)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stage_probe tools/attic/stage_probe.hip && /tmp/stage_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int M = 640, N = 3072, K = 768;

template <int MODE, int G, int BM = 64, int BN = 64>
__global__ __launch_bounds__(256, 1) void k_feed(const char* __restrict__ A, const char* __restrict__ W, unsigned* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int ROWS = 2 * (BM + BN);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = N / BN, tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const size_t a_plane = (size_t)M * K * 2, w_plane = (size_t)N * K * 2;
  constexpr int KB = MODE == 2 ? 128 : 64;           // bytes per row piece
  constexpr int RPI = 1024 / KB, LPR = KB / 16, PIECES = ROWS / RPI;
  const char* src[PIECES];
#pragma unroll
  for (int j = 0; j < PIECES; ++j) {
    const int lrow = RPI * j + lane / LPR;
    const char* base;
    if (lrow < BM) base = A + (size_t)(tm * BM + lrow) * K * 2;
    else if (lrow < 2 * BM) base = A + a_plane + (size_t)(tm * BM + lrow - BM) * K * 2;
    else if (lrow < 2 * BM + BN) base = W + (size_t)(tn * BN + lrow - 2 * BM) * K * 2;
    else base = W + w_plane + (size_t)(tn * BN + lrow - 2 * BM - BN) * K * 2;
    src[j] = base + (lane % LPR) * 16;
  }
  const int nkt = K * 2 / KB;                         // K-tiles of the block
  const int mine = wave < nkt ? (nkt - wave + 3) / 4 : 0;
  unsigned acc = 0;
  if (MODE == 0) {
    char* ring = lds + (size_t)wave * G * ROWS * KB;  // G = ring stages
    auto stage = [&](int buf, int kt) {
#pragma unroll
      for (int j = 0; j < PIECES; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (size_t)kt * KB),
                                         (__attribute__((address_space(3))) void*)(ring + (size_t)buf * ROWS * KB + j * 1024), 16, 0, 0);
    };
    for (int t = 0; t < G - 1; ++t) if (t < mine) stage(t, wave + 4 * t);
    for (int i = 0; i < mine; ++i) {
      constexpr int KEEP = PIECES * (G - 2) > 63 ? 63 : PIECES * (G - 2);
      if (i + G - 2 < mine) __builtin_amdgcn_s_waitcnt((KEEP & 15) | ((KEEP >> 4) << 14) | 0x0f70); else __builtin_amdgcn_s_waitcnt(0x0f70);
      if (i + G - 1 < mine) stage((i + G - 1) % G, wave + 4 * (i + G - 1));
      acc ^= *reinterpret_cast<const unsigned*>(ring + (size_t)(i % G) * ROWS * KB + lane * 16);   // "consume"
    }
  } else {
    char* buf = lds + (size_t)wave * ROWS * KB;
    uint4 v[G][PIECES];
#pragma unroll
    for (int g = 0; g < G - 1; ++g)
      if (g < mine) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) v[g][j] = *reinterpret_cast<const uint4*>(src[j] + (size_t)(wave + 4 * g) * KB);
      }
    for (int i0 = 0; i0 < mine; i0 += G) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int i = i0 + g;
        if (i < mine) {
          const int nx = i + G - 1;                      // the slot freed in the previous iteration: (g + G - 1) % G
          if (nx < mine) {
#pragma unroll
            for (int j = 0; j < PIECES; ++j) v[(g + G - 1) % G][j] = *reinterpret_cast<const uint4*>(src[j] + (size_t)(wave + 4 * nx) * KB);
          }
#pragma unroll
          for (int j = 0; j < PIECES; ++j) *reinterpret_cast<uint4*>(buf + j * 1024 + lane * 16) = v[g][j];
          acc ^= *reinterpret_cast<const unsigned*>(buf + lane * 16);
        }
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int G, int BM = 64, int BN = 64> static void run(const char* name, const char* A, const char* W, unsigned* sink) {
  constexpr int KB = MODE == 2 ? 128 : 64, ROWS = 2 * (BM + BN);
  const size_t smem = MODE == 0 ? (size_t)4 * G * ROWS * KB : (size_t)4 * ROWS * KB;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_feed<MODE, G, BM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int blocks = (M / BM) * (N / BN);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_feed<MODE, G, BM, BN>), dim3(blocks), dim3(256), smem, 0, A, W, sink);
  CK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0;
  for (int i = 0; i < 20; ++i) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_feed<MODE, G, BM, BN>), dim3(blocks), dim3(256), smem, 0, A, W, sink);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
  }
  const double kb_block = (double)ROWS * K * 2 / 1024.0, inflight = MODE == 0 ? (G - 1) * ROWS * KB / 1024.0 : (G - 1) * ROWS * KB / 1024.0;
  printf("%-58s %3d blocks x %.0f KB: %6.1f us avg %6.1f us best   %.0f KB in flight per wave\n", name, blocks, kb_block, sum / 20 * 1e3, best * 1e3, inflight);
}

int main() {
  char *A, *W; unsigned* sink;
  CK(hipMalloc(&A, (size_t)2 * M * K * 2)); CK(hipMalloc(&W, (size_t)2 * N * K * 2)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(A, 1, (size_t)2 * M * K * 2)); CK(hipMemset(W, 2, (size_t)2 * N * K * 2));
  run<0, 2>("LDS-DMA, 64 x 64 outputs, ring of 2 K-tiles per wave (128 KB)", A, W, sink);
  run<0, 2, 64, 32>("LDS-DMA, 64 x 32 outputs, ring of 2 K-tiles per wave (96 KB)", A, W, sink);
  run<0, 3, 64, 32>("LDS-DMA, 64 x 32 outputs, ring of 3 K-tiles per wave (144 KB)", A, W, sink);
  run<0, 2, 32, 32>("LDS-DMA, 32 x 32 outputs, ring of 2 K-tiles per wave (64 KB)", A, W, sink);
  run<0, 4, 32, 32>("LDS-DMA, 32 x 32 outputs, ring of 4 K-tiles per wave (128 KB)", A, W, sink);
  run<1, 2>("registers -> ds_write, 2 K-tiles (16 loads each)", A, W, sink);
  run<1, 3>("registers -> ds_write, 3 K-tiles", A, W, sink);
  run<1, 4>("registers -> ds_write, 4 K-tiles", A, W, sink);
  run<2, 2>("registers, 128-byte pieces, 2 tiles of 64 columns", A, W, sink);
  run<2, 3>("registers, 128-byte pieces, 3 tiles of 64 columns", A, W, sink);
  return 0;
}
