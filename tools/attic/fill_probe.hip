// Probe (GPU box): how fast can ONE CU pull L2-resident bytes towards its LDS, by transport? The wave-split GEMM tiles of
// round 5 run at ~25 GB/s per CU whatever the tile shape and ring depth — the rate of a single LDS-DMA stream. Candidates:
// global_load_lds_dwordx4 (what the kernels use), global_load_lds_dword, plain global_load_dwordx4 into registers followed by
// ds_write_b128, and the register loads alone. 256 threads per block, every block re-reads its own 96-KB region REPS times
// (3 MB per XCD: L2-resident), 8 loads in flight per wave.
//   hipcc --offload-arch=gfx950 -O3 -o fill_probe tools/fill_probe.hip && ./fill_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int REGION = 96 * 1024, REPS = 64;

// MODE 0: LDS-DMA 16 B per lane; 1: LDS-DMA 4 B per lane; 2: registers + ds_write_b128; 3: registers only;
// 4: registers + ds_write_b128, 128 contiguous bytes per 8 lanes (whole 128-B lines per row) instead of 64
template <int MODE>
__global__ __launch_bounds__(256, 1) void k_fill(const char* __restrict__ src, unsigned* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) char lds[128 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = src + (size_t)blockIdx.x * REGION + (size_t)wave * (REGION / 4);   // 24 KB per wave and pass
  char* wl = lds + wave * (32 * 1024);
  unsigned acc = 0;
  for (int rep = 0; rep < REPS; ++rep) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 24; ++i)      // 24 x 1 KB
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + i * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(wl + i * 1024), 16, 0, 0);
      __builtin_amdgcn_s_waitcnt(0x0f70);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 96; ++i)      // 96 x 256 B
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + i * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(wl + (i & 63) * 256), 4, 0, 0);
      __builtin_amdgcn_s_waitcnt(0x0f70);
    } else {
#pragma unroll
      for (int h = 0; h < 3; ++h) {     // 3 x 8 loads in flight
        uint4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const uint4*>(base + (h * 8 + i) * 1024 + lane * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (MODE == 2 || MODE == 4) *reinterpret_cast<uint4*>(wl + ((h * 8 + i) * 1024 + (lane ^ (i & 3)) * 16)) = v[i];
          else acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
        }
      }
    }
  }
  if (MODE == 2 || MODE == 4) { __syncthreads(); acc = *reinterpret_cast<unsigned*>(wl + lane * 4); }
  if (acc == 0x12345678u) sink[0] = acc;
}

// The GEMM tiles' access pattern: 128 rows of a row-major [rows][K] f16 plane, K-tiles of BPR bytes per row, one LDS-DMA
// instruction = (1024 / BPR) rows x BPR bytes; the four waves take K-tiles wave, wave + 4, ... and keep G of them in flight.
// Every block reads the SAME 128 rows (as column tiles share an activation panel): L2-resident.
template <int BPR, int G>
__global__ __launch_bounds__(256, 1) void k_tile(const char* __restrict__ src, int row_stride, int K_bytes, int reps) {
  __shared__ __attribute__((aligned(16))) char lds[128 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int RPI = 1024 / BPR, PIECES = 128 / RPI, LPR = BPR / 16;   // rows per instruction, instructions per K-tile, lanes per row
  char* wl = lds + wave * (32 * 1024);
  const char* lane_src = src + (size_t)(lane / LPR) * row_stride + (lane % LPR) * 16;
  const int nkt = K_bytes / BPR;
  for (int rep = 0; rep < reps; ++rep) {
    for (int kt0 = wave; kt0 < nkt; kt0 += 4 * G) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int kt = kt0 + 4 * g;
        if (kt < nkt) {
#pragma unroll
          for (int j = 0; j < PIECES; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(lane_src + (size_t)j * RPI * row_stride + (size_t)kt * BPR),
                                             (__attribute__((address_space(3))) void*)(wl + ((g * PIECES + j) & 31) * 1024), 16, 0, 0);
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0f70);
    }
  }
}

template <int BPR, int G> static void run_tile(const char* name, const char* src, int blocks, int row_stride, int K_bytes) {
  const int reps = 64;
  hipLaunchKernelGGL((k_tile<BPR, G>), dim3(blocks), dim3(256), 0, 0, src, row_stride, K_bytes, 2);
  CK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL((k_tile<BPR, G>), dim3(blocks), dim3(256), 0, 0, src, row_stride, K_bytes, reps);
  CK(hipDeviceSynchronize());
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  const double gb = (double)blocks * 128.0 * K_bytes * reps / 1e9;
  printf("%-34s stride %5d B, %3d-B pieces, %d K-tiles in flight, blocks %3d: %7.1f GB/s per block %7.2f TB/s total\n", name, row_stride, BPR, G, blocks,
         gb / blocks / (us * 1e-6), gb / (us * 1e-6) / 1e3);
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int MODE> static void run(const char* name, const char* src, unsigned* sink, int blocks) {
  hipLaunchKernelGGL(k_fill<MODE>, dim3(blocks), dim3(256), 0, 0, src, sink);
  CK(hipDeviceSynchronize());
  const double t0 = now_us();
  hipLaunchKernelGGL(k_fill<MODE>, dim3(blocks), dim3(256), 0, 0, src, sink);
  CK(hipDeviceSynchronize());
  const double us = now_us() - t0;
  const double gb = (double)blocks * REGION * REPS / 1e9;
  printf("%-52s blocks %3d: %8.1f us  %7.1f GB/s per block  %7.2f TB/s total\n", name, blocks, us, gb / blocks / (us * 1e-6), gb / (us * 1e-6) / 1e3);
}

int main() {
  char* src; unsigned* sink;
  CK(hipMalloc(&src, (size_t)256 * REGION)); CK(hipMemset(src, 1, (size_t)256 * REGION));
  CK(hipMalloc(&sink, 64));
  for (int blocks : {1, 8, 256}) {
    run<0>("LDS-DMA global_load_lds_dwordx4", src, sink, blocks);
    run<1>("LDS-DMA global_load_lds_dword", src, sink, blocks);
    run<2>("global_load_dwordx4 -> ds_write_b128", src, sink, blocks);
    run<3>("global_load_dwordx4 only", src, sink, blocks);
  }
  for (int blocks : {1, 256}) {
    run_tile<64, 3>("GEMM pattern K=768 (16 rows x 64 B)", src, blocks, 1536, 1536);
    run_tile<64, 3>("GEMM pattern K=3072", src, blocks, 6144, 6144);
    run_tile<64, 3>("rows padded by 128 B, K=768", src, blocks, 1536 + 128, 1536);
    run_tile<64, 3>("rows padded by 128 B, K=3072", src, blocks, 6144 + 128, 6144);
    run_tile<128, 2>("8 rows x 128 B, K=768", src, blocks, 1536, 1536);
    run_tile<128, 2>("8 rows x 128 B, K=3072", src, blocks, 6144, 6144);
    run_tile<128, 2>("8 rows x 128 B, padded, K=3072", src, blocks, 6144 + 128, 6144);
    run_tile<256, 1>("4 rows x 256 B, K=768", src, blocks, 1536, 1536);
    run_tile<256, 1>("4 rows x 256 B, K=3072", src, blocks, 6144, 6144);
  }
  return 0;
}
