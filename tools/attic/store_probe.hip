// Per-CU global store throughput probe (diagnostic, gfx950): how fast can ONE workgroup of 8 waves write a 256 x 256 fp32
// tile (256 KB) — the epilogue of gemm_h2_pp_kernel — as a function of the store pattern, and how does the rate change
// with the number of CUs storing at once. Build: hipcc --offload-arch=gfx950 -O3 -o tools/store_probe tools/store_probe.hip
// Usage: tools/store_probe   (prints us per 256 KB tile and bytes / cycle / CU at 2.4 GHz for each variant and grid size)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0: row-wise float4, 16 lanes per 256-B row piece, 4 rows per instruction (the current epilogue), row stride ld floats
// MODE 1: the same, nontemporal
// MODE 2: 64 lanes x 16 B contiguous (1 KB per wave-instruction), tile stored as one contiguous 256 KB block
// MODE 3: row-wise 8-byte stores (32 lanes per 256-B piece)
// MODE 4: MODE 0 through LDS first (write scattered, read row-wise) like the real epilogue
// MODE 5: 64 lanes per 1-KB row piece (a full 256-column row of the tile per instruction)
template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(float* out, int ld, int tiles_per_block, int iters) {
  __shared__ float stg[8 * 64 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  float4 v = make_float4((float)lane, (float)wave, 1.f, 2.f);
  for (int it = 0; it < iters; ++it)
  for (int t = 0; t < tiles_per_block; ++t) {
    const size_t tile = (size_t)blockIdx.x * tiles_per_block + t;
    if (MODE == 2) {
      float* base = out + tile * 65536 + (size_t)wave * 8192;
#pragma unroll 4
      for (int k = 0; k < 32; ++k) *reinterpret_cast<float4*>(base + k * 256 + lane * 4) = v;
      continue;
    }
    if (MODE == 5) {   // tile = 256 rows x 256 cols at row stride ld; wave w stores rows w*32..w*32+31, one row per instruction
      float* base = out + tile * 256 * (size_t)ld;
#pragma unroll 4
      for (int k = 0; k < 32; ++k) *reinterpret_cast<float4*>(base + (size_t)(wave * 32 + k) * ld + lane * 4) = v;
      continue;
    }
    // wave (wm, wn) owns rows wm*128.., cols wn*64..; two strips of 64 rows
    float* base = out + tile * 256 * (size_t)ld + (size_t)(wm * 128) * ld + wn * 64;
    for (int strip = 0; strip < 2; ++strip) {
      if (MODE == 4) {
        float* s = stg + wave * 4096;
        const int ncol = lane & 31, rsub = 4 * (lane >> 5);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[(ii * 32 + (r & 3) + 8 * (r >> 2) + rsub) * 64 + j * 32 + ncol] = v.x + r;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
      }
      if (MODE == 3) {
        const int rrow = lane >> 5, c2 = (lane & 31) * 2;
#pragma unroll 4
        for (int k = 0; k < 32; ++k)
          *reinterpret_cast<float2*>(base + (size_t)(strip * 64 + k * 2 + rrow) * ld + c2) = make_float2(v.x, v.y);
      } else {
        const int rrow = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
          float4 x = v;
          if (MODE == 4) x = *reinterpret_cast<const float4*>(stg + wave * 4096 + (k * 4 + rrow) * 64 + c4);
          float4* p = reinterpret_cast<float4*>(base + (size_t)(strip * 64 + k * 4 + rrow) * ld + c4);
          typedef float f4v __attribute__((ext_vector_type(4)));
          if (MODE == 1) { f4v y = {x.x, x.y, x.z, x.w}; __builtin_nontemporal_store(y, reinterpret_cast<f4v*>(p)); } else *p = x;
        }
      }
      if (MODE == 4) __builtin_amdgcn_wave_barrier();
    }
  }
}

template <int MODE>
float run(float* out, int ld, int blocks, int tiles, int iters) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(store_kernel<MODE>, dim3(blocks), dim3(512), 0, 0, out, ld, tiles, 1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  hipLaunchKernelGGL(store_kernel<MODE>, dim3(blocks), dim3(512), 0, 0, out, ld, tiles, iters);
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3f / (tiles * iters);   // us per tile per block
}

int main() {
  const int ld = 2304, tiles = 8, iters = 4;
  float* out = nullptr;
  const size_t bytes = (size_t)256 * tiles * 256 * ld * 4;   // 256 blocks x tiles x 256 rows x ld floats
  CK(hipMalloc(&out, bytes));
  CK(hipMemset(out, 0, bytes));
  const char* names[6] = {"row-wise float4 (epilogue pattern)", "row-wise float4 nontemporal", "contiguous 1 KB / instr", "row-wise float2",
                          "LDS-staged row-wise float4", "full 1-KB row / instr"};
  for (int blocks : {8, 32, 64, 128, 256}) {
    float us[6];
    us[0] = run<0>(out, ld, blocks, tiles, iters); us[1] = run<1>(out, ld, blocks, tiles, iters); us[2] = run<2>(out, ld, blocks, tiles, iters);
    us[3] = run<3>(out, ld, blocks, tiles, iters); us[4] = run<4>(out, ld, blocks, tiles, iters); us[5] = run<5>(out, ld, blocks, tiles, iters);
    for (int m = 0; m < 6; ++m)
      printf("blocks %3d  %-38s %7.2f us / 256 KB tile   %6.1f B/cyc/CU @2.4GHz   chip %6.2f TB/s\n", blocks, names[m], us[m],
             262144.0 / (us[m] * 2400.0), blocks * 262144.0 / us[m] / 1e6);
    printf("\n");
  }
  return 0;
}
