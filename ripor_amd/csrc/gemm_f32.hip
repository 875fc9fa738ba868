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
// Tiling (wave64): BMxBN block tile (128x128 or 128x64), BK = 32, 256 threads = 4 waves in a 2x2
// grid, each wave owns (BM/2)x(BN/2) = TMxTN MFMA 32x32 accumulators. Operands are staged through
// LDS in rows padded to 36 floats so that the ds_read_b128 fragment reads (16-lane groups, distinct
// rows, same k-offset) hit 16 distinct 4-bank slots: conflict-free. LDS is double-buffered: the
// next K-tile is fetched from global memory into registers before the MFMAs of the current tile
// are issued and written to the other LDS buffer after them, one barrier per K-tile.
//
// Fragment trick: lanes 0-31 read k..k+3 and lanes 32-63 read k+4..k+7 of their row with one
// ds_read_b128; the j-th register of every lane then forms a valid (A[i][k'], B[k'][j]) operand pair
// for k' in {k+j, k+4+j}, so four MFMAs consume eight k values (the k summation order is free).
#include <cstdlib>

#include "common.h"

namespace rpr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32, LDSP = BK + 4;

template <int BM, int BN, bool FULL>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  constexpr int TM = BM / 64, TN = BN / 64;      // MFMA tiles per wave
  constexpr int PA = BM / 32, PW = BN / 32;      // staging passes (32 rows of 8 float4 per pass)
  constexpr int TILE = (BM + BN) * LDSP;
  __shared__ __attribute__((aligned(16))) float smem[2 * TILE];

  // XCD-aware tile order: blocks b, b+8, b+16, ... run on the same XCD (same L2); give each XCD a
  // contiguous chunk of the (m-major, n-minor) tile list so neighbours share A row-panels in L2.
  int nt = tiles_m * tiles_n;
  int bid = blockIdx.x;
  if (g.m_dev) {   // packed encoder: only the tiles holding live rows are distributed (evenly) over the XCDs
    nt = ((*g.m_dev + BM - 1) / BM) * tiles_n;
    if (bid >= nt) return;
  }
  {
    const int q = nt >> 3, r = nt & 7, x = bid & 7, k = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int bm = tm * BM, bn = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // global -> register staging: thread (r0, c4) loads float4 c4 of rows r0, r0+32, ... of both tiles
  const int c4 = (tid & 7) * 4, r0 = tid >> 3;
  float4 ra[PA], rw[PW];
  const float* Ab = g.A + (size_t)(bm + r0) * g.lda + c4;
  const float* Wb = g.W + (size_t)(bn + r0) * g.ldw + c4;
  const size_t a_step = (size_t)32 * g.lda, w_step = (size_t)32 * g.ldw;
#define RPR_GLOAD(k0)                                                                              \
  _Pragma("unroll") for (int i = 0; i < PA; ++i) {                                                 \
    if (FULL || bm + r0 + 32 * i < g.M) ra[i] = *reinterpret_cast<const float4*>(Ab + i * a_step + (k0)); \
    else ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);                                                  \
  }                                                                                                \
  _Pragma("unroll") for (int i = 0; i < PW; ++i) {                                                 \
    if (FULL || bn + r0 + 32 * i < g.N) rw[i] = *reinterpret_cast<const float4*>(Wb + i * w_step + (k0)); \
    else rw[i] = make_float4(0.f, 0.f, 0.f, 0.f);                                                  \
  }
#define RPR_LSTORE(buf)                                                                            \
  _Pragma("unroll") for (int i = 0; i < PA; ++i)                                                   \
      *reinterpret_cast<float4*>(&(buf)[(r0 + 32 * i) * LDSP + c4]) = ra[i];                       \
  _Pragma("unroll") for (int i = 0; i < PW; ++i)                                                   \
      *reinterpret_cast<float4*>(&(buf)[(BM + r0 + 32 * i) * LDSP + c4]) = rw[i];

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fk = (lane >> 5) * 4;
  const int a_off = (wm * (BM / 2) + frow) * LDSP + fk;
  const int w_off = (BM + wn * (BN / 2) + frow) * LDSP + fk;

  auto compute = [&](const float* cur) {
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(cur + a_off + i * 32 * LDSP + kk * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(cur + w_off + j * 32 * LDSP + kk * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
  };

  const int nkt = g.K / BK;
  RPR_GLOAD(0)
  RPR_LSTORE(smem)
  __syncthreads();
  for (int kt = 0; kt + 1 < nkt; ++kt) {   // steady state: fetch tile kt+1 while multiplying tile kt
    RPR_GLOAD((kt + 1) * BK)
    compute(smem + (kt & 1) * TILE);
    float* nxt = smem + ((kt + 1) & 1) * TILE;
    RPR_LSTORE(nxt)
    __syncthreads();
  }
  compute(smem + ((nkt - 1) & 1) * TILE);
#undef RPR_GLOAD
#undef RPR_LSTORE

  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int ncol = lane & 31, rsub = 4 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = bn + wn * (BN / 2) + j * 32 + ncol;
    if (!FULL && n >= g.N) continue;
    const int oi = n / g.split_n, on = n - oi * g.split_n;
    float* outp = g.out[oi];
    const int ldo = g.ldo[oi];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mbase = bm + wm * (BM / 2) + i * 32 + rsub;
      float res[16];
      if (g.resid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + (r & 3) + 8 * (r >> 2);
          res[r] = (FULL || m < g.M) ? g.resid[(size_t)m * g.ldr + n] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2);
        if (FULL || m < g.M) {
          float v = acc[i][j][r];
          if (g.relu) v = fmaxf(v, 0.f);
          if (g.resid) v = res[r] + v;
          outp[out_off(g, oi, m, ldo, on)] = v;
        }
      }
    }
  }
}

template <int BM, int BN>
static hipError_t launch_cfg(const GemmArgs& a, hipStream_t s) {
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  const bool full = (a.M % BM == 0) && (a.N % BN == 0);
  if (full)
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true>), dim3(tiles_m * tiles_n), dim3(256), 0, s, a, tiles_m, tiles_n);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false>), dim3(tiles_m * tiles_n), dim3(256), 0, s, a, tiles_m, tiles_n);
  return hipGetLastError();
}

hipError_t launch_gemm(const GemmArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  if (a.K % BK != 0 || a.K <= 0 || (a.lda & 3) || (a.ldw & 3)) return hipErrorInvalidValue;
  static const int force = [] { const char* e = getenv("RPR_GEMM_TILE"); return e ? atoi(e) : 0; }();
  // 128x128 tiles unless that leaves the 256 CUs with less than ~1.5 blocks each
  const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
  const bool narrow = force ? (force == 64) : (t128 < 384);
  return narrow ? launch_cfg<128, 64>(a, s) : launch_cfg<128, 128>(a, s);
}

}  // namespace rpr
