// fp32 linear layer on the gfx950 matrix cores: C[M,N] = act(A[M,K] @ W[N,K]^T) (+ residual).
//
// Every projection of the T5 encoder/decoder on the search path goes through this kernel
// (reference: the nn.Linear calls inside HF T5Attention / T5DenseActDense reached from
// t5_pretrainer/modeling/t5_generative_retriever.py:358-366,403-416, and get_lm_logits :250-262).
//
// Why fp32 MFMA: the reference computes logits in fp32 and parity demands beam scores within 1e-4
// and identical integer smtid sequences, which bf16 operands (~1e-2 on O(10) logits) cannot give.
// v_mfma_f32_32x32x2_f32 is exact fp32 (a k-ordered fmaf chain) at 157 TFLOP/s peak.
//
// Tiling (wave64): 128x128 block tile, BK = 32, 256 threads = 4 waves in a 2x2 grid, each wave
// owns a 64x64 sub-tile = 2x2 MFMA 32x32 accumulators (64 VGPRs). Operands are staged through LDS
// in rows padded to 36 floats so that the ds_read_b128 fragment reads (16-lane groups, distinct
// rows, same k-offset) hit 16 distinct 4-bank slots: conflict-free. The next K-tile is prefetched
// from global memory into registers while the current one is multiplied (one MFMA K-tile is 64
// MFMAs x 64 cycles per wave, which hides the global latency).
//
// Fragment trick: lanes 0-31 read k..k+3 and lanes 32-63 read k+4..k+7 of their row with one
// ds_read_b128; the j-th register of every lane then forms a valid (A[i][k'], B[k'][j]) operand pair
// for k' in {k+j, k+4+j}, so four MFMAs consume eight k values (the k summation order is free).
#include "common.h"

namespace rpr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, LDSP = BK + 4;

__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float As[BM * LDSP];
  __shared__ __attribute__((aligned(16))) float Ws[BN * LDSP];

  // XCD-aware tile order: blocks b, b+8, b+16, ... run on the same XCD (same L2); give each XCD a
  // contiguous chunk of the (m-major, n-minor) tile list so neighbours share A row-panels in L2.
  const int nt = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nt >> 3, r = nt & 7, x = bid & 7, k = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int bm = tm * BM, bn = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // global -> register staging: thread loads 4 float4 of A and 4 of W per K-tile
  const int c4 = (tid & 7) * 4, r0 = tid >> 3;
  float4 ra[4], rw[4];
  const float* Ap[4];
  const float* Wp[4];
  bool av[4], wv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = bm + r0 + 32 * i, n = bn + r0 + 32 * i;
    av[i] = m < g.M;
    wv[i] = n < g.N;
    Ap[i] = g.A + (size_t)(av[i] ? m : 0) * g.lda + c4;
    Wp[i] = g.W + (size_t)(wv[i] ? n : 0) * g.ldw + c4;
  }
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = av[i] ? *reinterpret_cast<const float4*>(Ap[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
      rw[i] = wv[i] ? *reinterpret_cast<const float4*>(Wp[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(&As[(r0 + 32 * i) * LDSP + c4]) = ra[i];
      *reinterpret_cast<float4*>(&Ws[(r0 + 32 * i) * LDSP + c4]) = rw[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fk = (lane >> 5) * 4;
  const float* a_frag = &As[(wm * 64 + frow) * LDSP + fk];
  const float* w_frag = &Ws[(wn * 64 + frow) * LDSP + fk];

  const int nkt = g.K / BK;
  gload(0);
  for (int kt = 0; kt < nkt; ++kt) {
    lstore();
    __syncthreads();
    if (kt + 1 < nkt) gload((kt + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const float4*>(a_frag + i * 32 * LDSP + kk * 8);
        b[i] = *reinterpret_cast<const float4*>(w_frag + i * 32 * LDSP + kk * 8);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int ncol = lane & 31, rsub = 4 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = bn + wn * 64 + j * 32 + ncol;
    if (n >= g.N) continue;
    const int oi = n / g.split_n, on = n - oi * g.split_n;
    float* outp = g.out[oi];
    const int ldo = g.ldo[oi];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = bm + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rsub;
        if (m < g.M) {
          float v = acc[i][j][r];
          if (g.relu) v = fmaxf(v, 0.f);
          if (g.resid) v = g.resid[(size_t)m * g.ldr + n] + v;
          outp[(size_t)m * ldo + on] = v;
        }
      }
    }
  }
}

hipError_t launch_gemm(const GemmArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  if (a.K % BK != 0 || a.K <= 0 || (a.lda & 3) || (a.ldw & 3)) return hipErrorInvalidValue;
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL(gemm_f32_kernel, dim3(tiles_m * tiles_n), dim3(256), 0, s, a, tiles_m, tiles_n);
  return hipGetLastError();
}

}  // namespace rpr
