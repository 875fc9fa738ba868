// Probe 2 (GPU box): in the graph of a single-query search every kernel, however small, takes ~4.7 us start to start, while a
// graph of ONE trivial kernel repeated costs 1.5 us per node (tools/launch_probe.hip). Which property of the real sequence
// costs the 3 us? Variants of a 2000-node graph: distinct kernels alternating, LDS size alternating, grid size alternating,
// kernels with atomics / stores, large code, and a dependent chain through 30 KB.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int I> __global__ void k_triv(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[I] += I + 1; }
template <int I> __global__ __launch_bounds__(256) void k_ldsv(int* p) {
  __shared__ int s[(I + 1) * 6144];   // 24 / 48 / 72 / 96 KB
  s[threadIdx.x] = threadIdx.x; __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) p[I] += s[5];
}
__global__ __launch_bounds__(256) void k_atomic(unsigned long long* a, float* out) {
  out[blockIdx.x * 256 + threadIdx.x] = (float)threadIdx.x;
  if ((threadIdx.x & 63) == 0) atomicAdd(a + (blockIdx.x & 15), 1ull);
}
// ~N fused multiply-adds straight-line per instantiation: large code, executed once by every wave
template <int I, int N> __global__ __launch_bounds__(256) void k_code(float* p) {
  float a = p[threadIdx.x], b = (float)I;
#pragma unroll
  for (int i = 0; i < N; ++i) { a = a * b + (float)i; b = b * 1.0001f + a; }
  if (a == 12345.678f) p[0] = b;
}
__global__ __launch_bounds__(256) void k_dep(const float* __restrict__ in, float* __restrict__ out) {   // every block reads all 30 KB
  float acc = 0.f;
  for (int i = threadIdx.x; i < 7680; i += 256) acc += in[i];
  if (threadIdx.x < 160) out[blockIdx.x * 160 + threadIdx.x] = acc * 1e-9f;   // 48 blocks x 160 = 7680 outputs
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void graph_chain(const char* name, int N, hipStream_t s, const std::function<void(int, hipStream_t)>& launch) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < N; ++i) launch(i, s);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  const double t0 = now_us();
  for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  printf("%-64s graph %6.2f us/kernel\n", name, (now_us() - t0) / 3 / N);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int* p; CK(hipMalloc(&p, 1 << 20)); CK(hipMemset(p, 0, 1 << 20));
  float* f = reinterpret_cast<float*>(p) + 4096;
  unsigned long long* a64 = reinterpret_cast<unsigned long long*>(p) + 1024;
  const int N = 2000;
  graph_chain("one trivial kernel repeated (baseline)", N, s, [&](int, hipStream_t st) { hipLaunchKernelGGL(k_triv<0>, dim3(1), dim3(64), 0, st, p); });
  graph_chain("4 distinct trivial kernels alternating", N, s, [&](int i, hipStream_t st) {
    switch (i & 3) { case 0: hipLaunchKernelGGL(k_triv<0>, dim3(1), dim3(64), 0, st, p); break; case 1: hipLaunchKernelGGL(k_triv<1>, dim3(1), dim3(64), 0, st, p); break;
                     case 2: hipLaunchKernelGGL(k_triv<2>, dim3(1), dim3(64), 0, st, p); break; default: hipLaunchKernelGGL(k_triv<3>, dim3(1), dim3(64), 0, st, p); } });
  graph_chain("LDS size alternating 24/48/72/96 KB, 48 blocks", N, s, [&](int i, hipStream_t st) {
    switch (i & 3) { case 0: hipLaunchKernelGGL(k_ldsv<0>, dim3(48), dim3(256), 0, st, p); break; case 1: hipLaunchKernelGGL(k_ldsv<1>, dim3(48), dim3(256), 0, st, p); break;
                     case 2: hipLaunchKernelGGL(k_ldsv<2>, dim3(48), dim3(256), 0, st, p); break; default: hipLaunchKernelGGL(k_ldsv<3>, dim3(48), dim3(256), 0, st, p); } });
  graph_chain("LDS kernel (96 KB) alternating with a no-LDS kernel", N, s, [&](int i, hipStream_t st) {
    if (i & 1) hipLaunchKernelGGL(k_ldsv<3>, dim3(48), dim3(256), 0, st, p); else hipLaunchKernelGGL(k_triv<1>, dim3(3), dim3(256), 0, st, p); });
  graph_chain("one kernel, grid alternating 1/48/144/3 blocks", N, s, [&](int i, hipStream_t st) {
    const int g[4] = {1, 48, 144, 3}; hipLaunchKernelGGL(k_triv<0>, dim3(g[i & 3]), dim3(256), 0, st, p); });
  graph_chain("stores + u64 atomics, 48 blocks", N, s, [&](int, hipStream_t st) { hipLaunchKernelGGL(k_atomic, dim3(48), dim3(256), 0, st, a64, f); });
  graph_chain("stores + u64 atomics, 192 blocks", N, s, [&](int, hipStream_t st) { hipLaunchKernelGGL(k_atomic, dim3(192), dim3(256), 0, st, a64, f); });
  graph_chain("large code (8 x 2000 fma pairs) alternating, 48 blocks", N, s, [&](int i, hipStream_t st) {
    switch (i & 7) {
      case 0: hipLaunchKernelGGL((k_code<0, 2000>), dim3(48), dim3(256), 0, st, f); break; case 1: hipLaunchKernelGGL((k_code<1, 2000>), dim3(48), dim3(256), 0, st, f); break;
      case 2: hipLaunchKernelGGL((k_code<2, 2000>), dim3(48), dim3(256), 0, st, f); break; case 3: hipLaunchKernelGGL((k_code<3, 2000>), dim3(48), dim3(256), 0, st, f); break;
      case 4: hipLaunchKernelGGL((k_code<4, 2000>), dim3(48), dim3(256), 0, st, f); break; case 5: hipLaunchKernelGGL((k_code<5, 2000>), dim3(48), dim3(256), 0, st, f); break;
      case 6: hipLaunchKernelGGL((k_code<6, 2000>), dim3(48), dim3(256), 0, st, f); break; default: hipLaunchKernelGGL((k_code<7, 2000>), dim3(48), dim3(256), 0, st, f); } });
  graph_chain("large code, ONE instantiation repeated, 48 blocks", N, s, [&](int, hipStream_t st) { hipLaunchKernelGGL((k_code<0, 2000>), dim3(48), dim3(256), 0, st, f); });
  graph_chain("dependent chain: 48 blocks read all 30 KB of the predecessor", N, s, [&](int i, hipStream_t st) {
    hipLaunchKernelGGL(k_dep, dim3(48), dim3(256), 0, st, (i & 1) ? f + 8192 : f, (i & 1) ? f : f + 8192); });
  graph_chain("mixed: triv / lds96 / atomic / dep / code alternating", N, s, [&](int i, hipStream_t st) {
    switch (i % 5) { case 0: hipLaunchKernelGGL(k_triv<2>, dim3(1), dim3(256), 0, st, p); break; case 1: hipLaunchKernelGGL(k_ldsv<3>, dim3(48), dim3(256), 0, st, p); break;
                     case 2: hipLaunchKernelGGL(k_atomic, dim3(48), dim3(256), 0, st, a64, f); break; case 3: hipLaunchKernelGGL(k_dep, dim3(48), dim3(256), 0, st, f, f + 8192); break;
                     default: hipLaunchKernelGGL((k_code<3, 2000>), dim3(3), dim3(256), 0, st, f); } });
  return 0;
}
