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

  // epilogue: the MFMA result layout (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) of a 32 x 32 block) is
  // transposed through a private LDS strip per wave (the operand buffers are free) and leaves row-wise: a lane handles
  // 16-byte pieces, every residual piece is requested before the first store (vmcnt counts loads and stores alike: a load
  // waited for with stores in flight is a write-acknowledge round trip). Straight from the registers a 64 x 64 wave tile was
  // 64 four-byte store instructions per 32 x 32 block and as many residual loads, batch by batch behind the stores before
  // them (the epilogue all split-precision kernels had until round 4; gemm_h2.hip).
  constexpr int SH = BM / 2, SW = BN / 2;                 // the wave's outputs: rows x columns
  static_assert((size_t)BM * BN <= 2 * TILE, "the strips of the four waves fit the operand buffers");
  __syncthreads();                                        // every wave is done reading operand tiles
  float* stg = smem + (size_t)wave * (SH * SW);
  const int ncol = lane & 31, rsub = 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) stg[(i * 32 + (r & 3) + 8 * (r >> 2) + rsub) * SW + j * 32 + ncol] = acc[i][j][r];
  __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): the wave's own LDS writes have landed
  __builtin_amdgcn_wave_barrier();
  constexpr int LPR = SW / 4, RPI = 64 / LPR, NK = SH / RPI;   // lanes per staged row, rows per instruction, instructions
  const int rrow = lane / LPR, rc4 = (lane % LPR) * 4;
  const int mrow0 = bm + wm * SH, n0 = bn + wn * SW + rc4;      // this lane's 4 consecutive output columns (N % 4 == 0)
  const bool ncol_ok = FULL || n0 < g.N;
  const int oi = ncol_ok ? n0 / g.split_n : 0, on = n0 - oi * g.split_n;   // split_n % 4 == 0: the 4 columns share an output
  float* outp = oi == 0 ? g.out[0] : oi == 1 ? g.out[1] : g.out[2];
  const int ldo = oi == 0 ? g.ldo[0] : oi == 1 ? g.ldo[1] : g.ldo[2];
  const float relu_lo = g.relu ? 0.f : -INFINITY;
  float4 res[NK];
  if (g.resid) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {   // clamped, unconditional loads
      const int m = min(mrow0 + k * RPI + rrow, g.M - 1), nc = (FULL || n0 + 3 < g.N) ? n0 : 0;   // (N % 4 == 0 with a residual)
      res[k] = *reinterpret_cast<const float4*>(g.resid + (size_t)m * g.ldr + nc);
    }
  }
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int rl = k * RPI + rrow, m = mrow0 + rl;
    float4 v = *reinterpret_cast<const float4*>(stg + rl * SW + rc4);
    v.x = fmaxf(v.x, relu_lo); v.y = fmaxf(v.y, relu_lo); v.z = fmaxf(v.z, relu_lo); v.w = fmaxf(v.w, relu_lo);
    if (g.resid) { v.x = res[k].x + v.x; v.y = res[k].y + v.y; v.z = res[k].z + v.z; v.w = res[k].w + v.w; }
    if (ncol_ok && (FULL || m < g.M)) {
      float* o = outp + out_off(g, oi, m, ldo, on);
      if ((FULL || n0 + 3 < g.N) && (ldo & 3) == 0) *reinterpret_cast<float4*>(o) = v;
      else {   // a decoder vocab size off the 4 grid in exact-fp32 mode (N, and possibly the row stride, not a multiple of 4):
               // column by column
        o[0] = v.x;
        if (FULL || n0 + 1 < g.N) o[1] = v.y;
        if (FULL || n0 + 2 < g.N) o[2] = v.z;
        if (FULL || n0 + 3 < g.N) o[3] = v.w;
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
  // the epilogue stores 16-byte pieces where it can: with several outputs or a residual four consecutive columns must
  // belong to one output and be 16-byte aligned in it (a single output with an odd width / row stride is stored by column)
  if ((a.resid && ((a.N & 3) || (a.ldr & 3))) || (a.split_n < a.N && ((a.split_n & 3) || (a.ldo[0] & 3) || (a.ldo[1] & 3) || (a.ldo[2] & 3))) ||
      (a.rm_B && ((a.rm_stride & 3) || (a.rm_slot & 3) || (a.rm_head & 3))))
    return hipErrorInvalidValue;
  static const int force = [] { const char* e = dev_getenv("RPR_GEMM_TILE"); return e ? atoi(e) : 0; }();
  // 128x128 tiles unless that leaves the 256 CUs with less than ~1.5 blocks each
  const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
  const bool narrow = force ? (force == 64) : (t128 < 384);
  return narrow ? launch_cfg<128, 64>(a, s) : launch_cfg<128, 128>(a, s);
}

}  // namespace rpr
