// Probe (runs on the GPU box): what does a dependent kernel boundary cost on this stack, eager and graph-replayed, for the
// launch shapes of a single-query search (1 block, 48 blocks with 96 KB of LDS, a 336-byte argument struct), and what does
// a grid-wide barrier inside ONE launch cost (flat counter, XCD-hierarchical)? Decides between "more, leaner launches" and
// "one persistent launch per step" for the small-batch path.
//   hipcc --offload-arch=gfx950 -O3 -o launch_probe tools/launch_probe.hip && ./launch_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Big { int a[84]; };   // 336 bytes, like GemmH2Args

__global__ void k_trivial(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ __launch_bounds__(256) void k_lds(int* p) {
  __shared__ int s[24576];   // 96 KB
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += s[5];
}
__global__ __launch_bounds__(256) void k_big(Big b, int* p) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += b.a[threadIdx.x & 63];
}
// a dependent chain link that actually reads what its predecessor wrote (1 KB) and streams 48 KB per block like a skinny GEMM
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ w, const float* __restrict__ in, float* __restrict__ out, int n4) {
  float acc = in[threadIdx.x];
  const float4* src = w + (size_t)blockIdx.x * n4;
  for (int i = threadIdx.x; i < n4; i += 256) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
  if (blockIdx.x == 0) out[threadIdx.x] = acc * 1e-9f;
}

using gu32 = __attribute__((address_space(1))) unsigned;
__device__ __forceinline__ unsigned ld_relaxed(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u; }

// words: [0] arrivals (monotonic), [64] generation
template <int FENCE>
__global__ __launch_bounds__(256) void k_bar_flat(unsigned* bar, int nb, float* data) {
  const unsigned G = gridDim.x;
  for (int e = 1; e <= nb; ++e) {
    if (FENCE) data[(size_t)blockIdx.x * 256 + threadIdx.x] = (float)e;   // something to publish
    __syncthreads();
    if (threadIdx.x == 0) {
      if (FENCE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      const unsigned old = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (unsigned)e * G - 1) __hip_atomic_store(bar + 64, (unsigned)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else while (ld_relaxed(bar + 64) < (unsigned)e) __builtin_amdgcn_s_sleep(1);
      if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (FENCE && data[(size_t)((blockIdx.x + 37) % G) * 256 + threadIdx.x] != (float)e) bar[128] = 1;   // stale read detector
  }
}

// XCD-hierarchical: words [xcc*64] per-XCD arrivals, [512] top arrivals, [576] global generation, [640 + xcc*64] per-XCD generation,
// [1280 + xcc] blocks per XCD (census, filled by k_census before)
template <int FENCE>
__global__ __launch_bounds__(256) void k_bar_xcd(unsigned* bar, int nb, float* data) {
  const unsigned G = gridDim.x;
  __shared__ unsigned s_x, s_n, s_nx;
  if (threadIdx.x == 0) {   // census of THIS launch's placement (nothing promises it equals another launch's), then a flat barrier
    s_x = xcc_id();
    __hip_atomic_fetch_add(bar + 1280 + s_x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned old = __hip_atomic_fetch_add(bar + 1400, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == G - 1) __hip_atomic_store(bar + 1464, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else while (ld_relaxed(bar + 1464) < 1u) __builtin_amdgcn_s_sleep(1);
    s_n = ld_relaxed(bar + 1280 + s_x);
    unsigned nx = 0; for (int i = 0; i < 16; ++i) nx += ld_relaxed(bar + 1280 + i) ? 1 : 0; s_nx = nx;
  }
  __syncthreads();
  const unsigned x = s_x, n = s_n, nx = s_nx;
  for (int e = 1; e <= nb; ++e) {
    if (FENCE) data[(size_t)blockIdx.x * 256 + threadIdx.x] = (float)e;
    __syncthreads();
    if (threadIdx.x == 0) {
      if (FENCE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      const unsigned old = __hip_atomic_fetch_add(bar + x * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (unsigned)e * n - 1) {   // last arriver of this XCD
        const unsigned t = __hip_atomic_fetch_add(bar + 512, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned)e * nx - 1) __hip_atomic_store(bar + 576, (unsigned)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else while (ld_relaxed(bar + 576) < (unsigned)e) __builtin_amdgcn_s_sleep(1);
        __hip_atomic_store(bar + 640 + x * 64, (unsigned)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (ld_relaxed(bar + 640 + x * 64) < (unsigned)e) __builtin_amdgcn_s_sleep(1);
      }
      if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (FENCE && data[(size_t)((blockIdx.x + 37) % G) * 256 + threadIdx.x] != (float)e) bar[2000] = 1;
  }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F>
static void chain(const char* name, int N, hipStream_t s, F launch) {
  // eager
  for (int i = 0; i < 50; ++i) launch(s);
  CK(hipStreamSynchronize(s));
  double t0 = now_us();
  for (int i = 0; i < N; ++i) launch(s);
  const double t_issue = now_us() - t0;
  CK(hipStreamSynchronize(s));
  const double t_eager = now_us() - t0;
  // graph
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < N; ++i) launch(s);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  t0 = now_us();
  const int reps = 3;
  for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  const double t_graph = (now_us() - t0) / reps;
  printf("%-44s eager %6.2f us/kernel (host issue %5.2f)   graph %6.2f us/kernel\n", name, t_eager / N, t_issue / N, t_graph / N);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int* p; CK(hipMalloc(&p, 4096)); CK(hipMemset(p, 0, 4096));
  const int N = 2000;
  printf("env HIP_FORCE_DEV_KERNARG=%s\n", getenv("HIP_FORCE_DEV_KERNARG") ? getenv("HIP_FORCE_DEV_KERNARG") : "(unset)");
  chain("trivial, 1 block x 64", N, s, [&](hipStream_t st) { hipLaunchKernelGGL(k_trivial, dim3(1), dim3(64), 0, st, p); });
  chain("trivial, 256 blocks x 256", N, s, [&](hipStream_t st) { hipLaunchKernelGGL(k_trivial, dim3(256), dim3(256), 0, st, p); });
  chain("96 KB LDS, 48 blocks x 256", N, s, [&](hipStream_t st) { hipLaunchKernelGGL(k_lds, dim3(48), dim3(256), 0, st, p); });
  Big b{};
  chain("336-byte argument, 48 blocks x 256", N, s, [&](hipStream_t st) { hipLaunchKernelGGL(k_big, dim3(48), dim3(256), 0, st, b, p); });
  {
    float4* w; float *a, *o; const int n4 = 3072;   // 48 KB per block
    for (int blocks : {48, 192, 256}) {
      CK(hipMalloc(&w, (size_t)blocks * n4 * 16 * 8)); CK(hipMemset(w, 0, (size_t)blocks * n4 * 16 * 8));
      CK(hipMalloc(&a, 4096)); CK(hipMalloc(&o, 4096)); CK(hipMemset(a, 0, 4096)); CK(hipMemset(o, 0, 4096));
      int i = 0;
      char nm[80]; snprintf(nm, sizeof nm, "dependent stream link, %d blocks x 48 KB", blocks);
      chain(nm, N, s, [&](hipStream_t st) {   // ping-pong in/out, rotate over 8 weight sets (not L2 resident as a whole)
        hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, st, w + (size_t)(i & 7) * blocks * n4, (i & 1) ? o : a, (i & 1) ? a : o, n4);
        ++i;
      });
      CK(hipFree(w)); CK(hipFree(a)); CK(hipFree(o));
    }
  }
  // grid barriers inside one launch
  unsigned* bar; CK(hipMalloc(&bar, 16384));
  float* data; CK(hipMalloc(&data, (size_t)1024 * 256 * 4));
  const int NB = 1000;
  auto run_bar = [&](const char* name, int grid, auto kern) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemsetAsync(bar, 0, 16384, s));
      if (rep == 0) { }
      CK(hipStreamSynchronize(s));
      const double t0 = now_us();
      kern(grid);
      CK(hipStreamSynchronize(s));
      const double t = now_us() - t0;
      unsigned h[2048]; CK(hipMemcpy(h, bar, 8192 + 16, hipMemcpyDeviceToHost));
      if (rep == 1) printf("%-44s %6.2f us/barrier  (grid %d, stale flag %u/%u, blocks per XCD %u %u %u %u %u %u %u %u)\n", name, t / NB, grid,
                           h[128], h[2000], h[1280], h[1281], h[1282], h[1283], h[1284], h[1285], h[1286], h[1287]);
    }
  };
  for (int grid : {256, 512}) {
    run_bar("flat counter, no fences", grid, [&](int g) { hipLaunchKernelGGL(k_bar_flat<0>, dim3(g), dim3(256), 0, s, bar, NB, data); });
    run_bar("flat counter, release + acquire", grid, [&](int g) { hipLaunchKernelGGL(k_bar_flat<1>, dim3(g), dim3(256), 0, s, bar, NB, data); });
    run_bar("XCD-hierarchical, no fences", grid, [&](int g) { hipLaunchKernelGGL(k_bar_xcd<0>, dim3(g), dim3(256), 0, s, bar, NB, data); });
    run_bar("XCD-hierarchical, release + acquire", grid, [&](int g) { hipLaunchKernelGGL(k_bar_xcd<1>, dim3(g), dim3(256), 0, s, bar, NB, data); });
  }
  return 0;
}
