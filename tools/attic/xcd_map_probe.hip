// Probe (GPU box): which XCD does block b of a launch run on — on the whole chip and on the two CU-masked lane streams of
// rpr_search (hipExtStreamCreateWithCUMask, CUs 0..127 / 128..255 of the mask's bit order)? The persistent GEMM gives
// "XCD x = blockIdx & 7" a contiguous chunk of tiles so that tiles sharing an A panel share an L2; if the mapping differs on a
// masked stream the sharing is lost (profiles: fabric reads 2.07 x the algorithmic bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ __launch_bounds__(512) void k_where(unsigned* out, int spin) {
  __shared__ char big[120 * 1024];   // one block per CU
  big[threadIdx.x] = 1;
  if (threadIdx.x == 0) {
    unsigned x, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[2 * blockIdx.x] = x & 15u; out[2 * blockIdx.x + 1] = hw;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}     // keep the CU busy so that every block needs its own CU
  }
}
static void report(const char* name, hipStream_t s, int grid) {
  unsigned* d; CK(hipMalloc(&d, 8 * 1024)); CK(hipMemsetAsync(d, 0xff, 8 * 1024, s));
  hipLaunchKernelGGL(k_where, dim3(grid), dim3(512), 0, s, d, 20000);
  CK(hipStreamSynchronize(s));
  unsigned h[2048]; CK(hipMemcpy(h, d, 8 * 1024, hipMemcpyDeviceToHost));
  int match = 0, cnt[16] = {0};
  for (int b = 0; b < grid; ++b) { match += (h[2 * b] == (unsigned)(b & 7)); cnt[h[2 * b] & 15]++; }
  printf("%-28s grid %3d: XCD == blockIdx %% 8 for %3d blocks; blocks per XCD:", name, grid, match);
  for (int i = 0; i < 8; ++i) printf(" %d", cnt[i]);
  printf("; first 16 blocks ->");
  for (int b = 0; b < 16 && b < grid; ++b) printf(" %u", h[2 * b]);
  printf("\n");
  CK(hipFree(d));
}
int main() {
  hipStream_t s0; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  report("whole chip", s0, 256); report("whole chip", s0, 128); report("whole chip", s0, 210);
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, words = (cus + 31) / 32;
  for (int i = 0; i < 2; ++i) {
    uint32_t mask[32] = {0};
    for (int k = (i == 0 ? 0 : cus / 2); k < (i == 0 ? cus / 2 : cus); ++k) mask[k >> 5] |= 1u << (k & 31);
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask));
    char nm[64]; snprintf(nm, sizeof nm, "lane %d (CU mask half %d)", i, i);
    report(nm, s, 128); report(nm, s, 126); report(nm, s, 64);
  }
  return 0;
}
