// Diagnostic (GPU box): what does the chip SUSTAIN under its power cap on the matrix instructions a projection GEMM could be
// built from — registers only, random operands, 1 and 2 waves per SIMD, ~0.5 s per point so that DVFS settles:
//   f16   v_mfma_f32_32x32x16_f16            (the split-precision GEMM: 3 per product)
//   bf16  v_mfma_f32_32x32x16_bf16
//   i8    v_mfma_i32_32x32x32_i8             (integer slices: exact accumulation, 6 products for 3 x 3 slices of 7 bits)
//   fp8   v_mfma_scale_f32_32x32x64_f8f6f4   (e4m3 x e4m3, unit scales)
// Reports tera-ops per second (2 * 32 * 32 * K per instruction) next to the nominal peak of the instruction at 2.4 GHz.
// Question behind it (DESIGN.md section 10 "what comes next"): the f16 pipe sustains ~0.6 of nominal on random data; if the
// integer pipe sustains a larger share, an int8-slice GEMM (6 MFMAs at twice the rate) would beat 3 f16 MFMAs per product.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_dtype_probe tools/mfma_dtype_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0 f16, 1 bf16, 2 i8, 3 fp8 (scaled instruction, 32 bytes per operand and lane)
template <int MODE>
__global__ __launch_bounds__(512, 2) void probe_kernel(const int* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  i32x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const i32x8*>(src + ((size_t)tid * 8 + i) * 8);
    b[i] = *reinterpret_cast<const i32x8*>(src + ((size_t)tid * 8 + 4 + i) * 8);
  }
  float s = 0.f;
  if (MODE == 2) {
    i32x16 acc[8];
    for (int i = 0; i < 8; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const i32x4 x = {a[i & 3][0], a[i & 3][1], a[i & 3][2], a[i & 3][3]}, y = {b[(i >> 1) & 3][0], b[(i >> 1) & 3][1], b[(i >> 1) & 3][2], b[(i >> 1) & 3][3]};
        acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, acc[i], 0, 0, 0);
      }
    }
    for (int i = 0; i < 8; ++i)
      for (int r = 0; r < 16; ++r) s += (float)acc[i][r];
  } else {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const i32x8 x = a[i & 3], y = b[(i >> 1) & 3];
        if (MODE == 0) {
          const i32x4 x4 = {x[0], x[1], x[2], x[3]}, y4 = {y[0], y[1], y[2], y[3]};
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x4), __builtin_bit_cast(f16x8, y4), acc[i], 0, 0, 0);
        } else if (MODE == 1) {
          const i32x4 x4 = {x[0], x[1], x[2], x[3]}, y4 = {y[0], y[1], y[2], y[3]};
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x4), __builtin_bit_cast(bf16x8, y4), acc[i], 0, 0, 0);
        } else {
          acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x, y, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
      }
    }
    for (int i = 0; i < 8; ++i)
      for (int r = 0; r < 16; ++r) s += acc[i][r];
  }
  out[tid] = s;
}

template <int MODE>
static void run(const char* name, int kdepth, double nominal_tops, const int* d, float* o, int blocks, int iters) {
  for (int th = 256; th <= 512; th += 256) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe_kernel<MODE><<<blocks, th>>>(d, o, 2000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe_kernel<MODE><<<blocks, th>>>(d, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * (th / 64) * iters * 8 * 2.0 * 32 * 32 * kdepth;
    const double tops = ops / (ms * 1e-3) / 1e12;
    printf("%-5s K=%2d  %d wave(s)/SIMD: %7.1f ms  %6.0f Tops/s sustained  = %.2f of the nominal %.0f\n", name, kdepth, th / 256, ms, tops,
           tops / nominal_tops, nominal_tops);
  }
}

int main(int argc, char** argv) {
  const int blocks = 256, iters = argc > 1 ? atoi(argv[1]) : 60000;
  const int zero = argc > 2 ? atoi(argv[2]) : 0;           // 1: all-zero operands (the pipe's rate without operand toggling)
  const size_t n = (size_t)blocks * 512 * 64;               // ints
  std::vector<int> h(n);
  int* d; float* o;
  hipMalloc(&d, n * 4); hipMalloc(&o, (size_t)blocks * 512 * 4);
  // f16 / bf16: random halves in [-1, 1); i8 / fp8: random bytes (fp8: e4m3 of magnitude 2^-2 .. 2^2)
  std::vector<int> hf(n), hb(n), hi(n), h8(n);
  srand(1);
  for (size_t i = 0; i < n; ++i) {
    unsigned w_f = 0, w_b = 0, w_i = 0, w_8 = 0;
    if (!zero) {
      for (int k = 0; k < 2; ++k) {
        const float u = (float)rand() / RAND_MAX * 2.f - 1.f;
        _Float16 v = (_Float16)u; unsigned short bits; memcpy(&bits, &v, 2);
        w_f |= (unsigned)bits << (16 * k);
        unsigned fb; memcpy(&fb, &u, 4);
        w_b |= (fb >> 16) << (16 * k);
      }
      for (int k = 0; k < 4; ++k) {
        w_i |= (unsigned)(rand() & 0xff) << (8 * k);
        const unsigned byte = ((unsigned)(rand() & 1) << 7) | ((5u + (rand() & 3)) << 3) | (rand() & 7);   // e4m3 in [2^-2, 2^2)
        w_8 |= byte << (8 * k);
      }
    }
    hf[i] = (int)w_f; hb[i] = (int)w_b; hi[i] = (int)w_i; h8[i] = (int)w_8;
  }
  printf("operands: %s; %d iterations x 8 MFMAs per wave\n", zero ? "zeros" : "random", iters);
  hipMemcpy(d, hf.data(), n * 4, hipMemcpyHostToDevice); run<0>("f16", 16, 2516.6, d, o, blocks, iters);
  hipMemcpy(d, hb.data(), n * 4, hipMemcpyHostToDevice); run<1>("bf16", 16, 2516.6, d, o, blocks, iters);
  hipMemcpy(d, hi.data(), n * 4, hipMemcpyHostToDevice); run<2>("i8", 32, 5033.2, d, o, blocks, iters);
  hipMemcpy(d, h8.data(), n * 4, hipMemcpyHostToDevice); run<3>("fp8", 64, 5033.2, d, o, blocks, iters);
  return 0;
}
