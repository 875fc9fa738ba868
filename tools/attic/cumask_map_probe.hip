// Probe (GPU box): which XCD / CU does bit k of a hipExtStreamCreateWithCUMask mask enable?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void k_where(unsigned* out) {
  if (threadIdx.x == 0) {
    unsigned x, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[2 * blockIdx.x] = x; out[2 * blockIdx.x + 1] = hw;
  }
}
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, words = (cus + 31) / 32;
  unsigned* d; CK(hipMalloc(&d, 4096));
  printf("%d CUs, %d mask words\n", cus, words);
  for (int k = 0; k < cus; k += (k < 40 ? 1 : 13)) {
    uint32_t mask[32] = {0};
    mask[k >> 5] = 1u << (k & 31);
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask) != hipSuccess) { printf("bit %3d: stream creation failed\n", k); (void)hipGetLastError(); continue; }
    CK(hipMemsetAsync(d, 0xff, 4096, s));
    hipLaunchKernelGGL(k_where, dim3(4), dim3(64), 0, s, d);
    CK(hipStreamSynchronize(s));
    unsigned h[8]; CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
    printf("bit %3d: blocks ->", k);
    for (int b = 0; b < 4; ++b) printf(" xcc %u hw_id 0x%08x (cu %u sh %u se %u)", h[2 * b] & 15, h[2 * b + 1], (h[2 * b + 1] >> 8) & 15, (h[2 * b + 1] >> 12) & 1, (h[2 * b + 1] >> 13) & 7);
    printf("\n");
    CK(hipStreamDestroy(s));
  }
  return 0;
}
