// Probe (GPU box): does a chain of dependent small kernels run faster when the stream is confined to ONE XCD (CU mask), i.e. when
// producer and consumer blocks share an L2? Chain link = 48 blocks that each read the 30 KB their predecessor wrote and write 160
// floats (the shape of a 10-row decoder GEMM's activations), 2000 links as a hipGraph; the same with a streamed 48-KB weight
// slice per block. Streams: whole chip, one XCD (mask bits k % 8 == x), two XCDs.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ __launch_bounds__(256) void k_dep(const float* __restrict__ in, float* __restrict__ out, unsigned* where) {
  float acc = 0.f;
#pragma unroll 6
  for (int i = threadIdx.x; i < 7680; i += 256) acc += in[i];
  if (threadIdx.x < 160) out[blockIdx.x * 160 + threadIdx.x] = acc * 1e-9f;
  if (where && threadIdx.x == 0) { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); atomicOr(where, 1u << (x & 15)); }
}
__global__ __launch_bounds__(256) void k_dep_w(const float4* __restrict__ w, const float* __restrict__ in, float* __restrict__ out) {
  float acc = 0.f;
  const float4* src = w + (size_t)blockIdx.x * 3072;
#pragma unroll 4
  for (int i = threadIdx.x; i < 3072; i += 256) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
#pragma unroll 6
  for (int i = threadIdx.x; i < 7680; i += 256) acc += in[i];
  if (threadIdx.x < 160) out[blockIdx.x * 160 + threadIdx.x] = acc * 1e-9f;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double graph_chain(int N, hipStream_t s, const std::function<void(int, hipStream_t)>& launch) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < N; ++i) launch(i, s);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  const double t0 = now_us();
  for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  const double us = (now_us() - t0) / 3 / N;
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return us;
}
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, words = (cus + 31) / 32;
  float* f; CK(hipMalloc(&f, 1 << 20)); CK(hipMemset(f, 0, 1 << 20));
  float4* w; CK(hipMalloc(&w, (size_t)8 * 48 * 3072 * 16)); CK(hipMemset(w, 0, (size_t)8 * 48 * 3072 * 16));
  unsigned* where; CK(hipMalloc(&where, 64));
  struct Cfg { const char* name; int nx; } cfgs[] = {{"whole chip", 8}, {"one XCD (mask bits k % 8 == 0)", 1}, {"two XCDs (k % 8 in {0, 1})", 2}, {"four XCDs", 4}};
  for (const Cfg& c : cfgs) {
    hipStream_t s;
    if (c.nx == 8) { CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
    else {
      uint32_t mask[32] = {0};
      for (int k = 0; k < cus; ++k) if ((k & 7) < c.nx) mask[k >> 5] |= 1u << (k & 31);
      CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask));
    }
    CK(hipMemsetAsync(where, 0, 64, s));
    hipLaunchKernelGGL(k_dep, dim3(48), dim3(256), 0, s, f, f + 8192, where);
    CK(hipStreamSynchronize(s));
    unsigned xm = 0; CK(hipMemcpy(&xm, where, 4, hipMemcpyDeviceToHost));
    const double a = graph_chain(2000, s, [&](int i, hipStream_t st) { hipLaunchKernelGGL(k_dep, dim3(48), dim3(256), 0, st, (i & 1) ? f + 8192 : f, (i & 1) ? f : f + 8192, (unsigned*)nullptr); });
    const double a1 = graph_chain(2000, s, [&](int i, hipStream_t st) { hipLaunchKernelGGL(k_dep, dim3(3), dim3(256), 0, st, (i & 1) ? f + 8192 : f, (i & 1) ? f : f + 8192, (unsigned*)nullptr); });
    const double b = graph_chain(2000, s, [&](int i, hipStream_t st) { hipLaunchKernelGGL(k_dep_w, dim3(48), dim3(256), 0, st, w + (size_t)(i & 7) * 48 * 3072, (i & 1) ? f + 8192 : f, (i & 1) ? f : f + 8192); });
    const double b2 = graph_chain(2000, s, [&](int i, hipStream_t st) { hipLaunchKernelGGL(k_dep_w, dim3(192), dim3(256), 0, st, w, (i & 1) ? f + 8192 : f, (i & 1) ? f : f + 8192); });
    printf("%-34s XCDs seen 0x%02x: dep 48 blocks %5.2f us, dep 3 blocks %5.2f us, dep + 48-KB weight slice x 48 blocks %5.2f us, x 192 blocks %5.2f us\n", c.name, xm, a, a1, b, b2);
  }
  return 0;
}
