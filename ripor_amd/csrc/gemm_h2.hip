// Split-precision linear layer on the gfx950 matrix cores: C[M,N] = act(A[M,K] @ W[N,K]^T) (+ residual)
// with fp32-equivalent accuracy from f16 MFMA.
//
// Every fp32 operand x is carried as two f16 planes  s*x = hi + lo,  hi = f16(s*x), lo = f16(s*x - hi), with a
// power-of-two scale s per tensor class (weights 2^8, activations 2^4, FF intermediate 2^-4; common.h) that keeps
// lo out of the f16 subnormals and the FF intermediate inside the f16 range: 22 significant bits,
// |s*x - hi - lo| <= 2^-22 |s*x|. A product is evaluated as  hi*hi + hi*lo + lo*hi  with three
// v_mfma_f32_32x32x16_f16 accumulating in fp32 (the dropped lo*lo term is <= 2^-22 |xy|); the epilogue multiplies
// the accumulators by 1/(s_A s_W) (exact). Measured on the t5-base golden model (tools/precision_probe.py): logits
// differ from the exact-fp32 path by <= 7.4e-5 (mean 1.2e-5; two fp32 summation orders differ by <= 4e-5), beam
// scores by <= 5e-6 — inside the 1e-4 parity bar — while the matrix pipe runs at the f16 rate: 3 MFMAs of 32
// cycles instead of 8 fp32 MFMAs of 64 cycles per 32x32x16 block = 5.3x the fp32-MFMA roof. The exact fp32 kernel
// (gemm_f32.hip) stays selectable (RPR_PRECISION=f32) as the numerical reference.
//
// Planes are produced where the data is written (RMSNorm, attention and ReLU epilogues emit hi/lo
// planes; weights are split once at rpr_load_model), so this kernel only moves 16-bit data:
// operand traffic is 4 B/element, the same as fp32.
//
// Kernels: gemm_h2_pp_kernel (256x256x32 tiles, 8 waves, ping-pong schedule; the default whenever a launch has
// >= 112 such tiles) and gemm_h2_dma_kernel (128x128 / 128x64 tiles, 4 waves) for smaller launches; both stage
// K-tiles with LDS-DMA into unpadded XOR-swizzled rows. Earlier variants (register staging, in-phase software
// pipelining) are in the history and in DESIGN.md §5 / §8 with their measured numbers.
#include <algorithm>
#include <cstdio>
#include <type_traits>
#include <hip/hip_fp16.h>

#include <cstdlib>

#include "common.h"

#include <algorithm>
#include <vector>

namespace rpr {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// the LDS-transposed epilogue of the tile kernels (defined behind gemm_h2_dma_kernel)
template <bool FULL, int TM, int TN, int WM, int WN, int BM, int BN, bool BOUT = false>
__device__ __forceinline__ void h2_epilogue_256(const GemmH2Args& g, f32x16 (&acc)[TM][TN], __half* smem, int wave, int lane, int bm,
                                                int bn, int wm, int wn, const float* rs_tile, float acc_scale);

constexpr int HBK = 32;  // K-tile depth = halves per LDS row (64 B, unpadded)


// ---- LDS-DMA variant -------------------------------------------------------------------------------
// Same math, but the K-tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (no staging VGPRs, no
// ds_write pass — the ds_write_b128 of the register-staged kernel cost as many LDS cycles as all the
// fragment reads and made the LDS co-critical with the matrix pipe). LDS is double-buffered; the DMA
// of tile t+1 is issued before the MFMAs of tile t and drained by the single barrier that ends the
// iteration. An LDS-DMA instruction writes wave-uniform base + lane*16 B, so rows are stored
// unpadded (64 B) and the 16-byte segments of a row are XOR-swizzled with (row>>2)&3 on the SOURCE
// address and again on the fragment read: the 16-lane groups of ds_read_b128 then hit 16 distinct
// 4-bank slots.
// STAGES = LDS buffers. 2: one K-tile of prefetch, several blocks per CU hide the rest (large launches).
// 4: launches with fewer tiles than CUs (small in-flight batches) — one block per CU walks its whole K alone, and
// with one tile of prefetch every K-tile paid a full L2/HBM latency (~1.2 us x 24 K-tiles per GEMM); three tiles
// in flight and a counted s_waitcnt make the walk bandwidth-bound instead.
// BF16 = false: the split-precision f16 hi/lo planes (3 MFMAs per product, K-tile of 32). BF16 = true: ONE bf16 plane per
// operand and one v_mfma_f32_32x32x16_bf16 per product with fp32 accumulation — the arithmetic of the reference's bf16
// autocast for the training GEMMs (RPR_PREC_BF16; main.py:152, tasks/trainer.py:229), except that the result is not
// rounded to bf16. The LDS layout is the same: where the split kernel keeps the lo planes, this one keeps the NEXT 32
// columns of K, so a K-tile is 64 deep (16 MFMAs per wave and barrier for the 128x128 tile instead of 8).
template <int BM, int BN, int WM, int WN, bool FULL, int STAGES = 2, bool BF16 = false>
__global__ __launch_bounds__(64 * WM * WN, STAGES > 2 ? 1 : (WM * WN) / 4 * ((BM * BN >= 256 * 256) ? 1 : 2))
void gemm_h2_dma_kernel(GemmH2Args g, int tiles_m, int tiles_n) {
  // per-tensor dynamic plane scales (training): read from the device; g itself must stay untouched — a kernel that writes
  // to its by-value argument struct gets a private copy of all 320 bytes in scratch (measured: +20 % per launch)
  const float acc_scale = g.dyn_a ? 1.0f / (dyn_plane_scale(*g.dyn_a) * dyn_plane_scale(*g.dyn_b)) : g.acc_scale;
  constexpr int NW = WM * WN;                    // waves per block
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);  // MFMA 32x32 tiles per wave
  constexpr int ROWS = 2 * (BM + BN);            // LDS rows of 32 halves (64 B) per buffer
  constexpr int KSTEP = BF16 ? 2 * HBK : HBK;    // K columns per tile
  constexpr int NINST = ROWS / 16;               // DMA wave-instructions per K-tile (16 rows each)
  constexpr int PER_WAVE = NINST / NW;
  static_assert(NINST % NW == 0, "tile rows must split evenly over the waves");
  __shared__ __attribute__((aligned(16))) __half smem[STAGES * ROWS * HBK];
  __shared__ float rs_tile[BM];                      // fused RMSNorm: rsqrt(mean(x^2) + eps) of the tile's rows (split-precision launches)
  static_assert(64 * WM * WN >= BM, "one thread per tile row fills rs_tile");

  int nt = tiles_m * tiles_n;
  int bid = blockIdx.x;
  if (g.m_dev) {   // packed encoder: only the tiles holding live rows are distributed (evenly) over the XCDs
    const int live = max(0, min(*g.m_dev - g.m_base, g.M));
    if (g.live_hi > 0 && (*g.m_dev <= g.live_lo || *g.m_dev > g.live_hi)) return;   // the other kernel of the pair runs
    nt = ((live + BM - 1) / BM) * tiles_n;
    if (bid >= nt) return;
  }
  {
    const int q = nt >> 3, r = nt & 7, x = bid & 7, k = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int bm = tm * BM, bn = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  if (!BF16 && g.row_ssq && tid < BM) {   // read by the epilogue, behind the K-loop's barriers
    const int m = bm + tid;
    rs_tile[tid] = (m < g.M) ? ssq_rsqrt(g.row_ssq[m], g.inv_d_fix, g.eps) : 1.f;
  }
  // split-K (training weight gradients: few output tiles, K = thousands of rows): blockIdx.y walks its own range of
  // K-tiles and stores a partial result at out + blockIdx.y * part_stride; splitk_reduce_kernel adds them in order
  int nkt = g.K / KSTEP, kbeg = 0;
  if (g.ksplit > 1) {
    const int per = (nkt + g.ksplit - 1) / g.ksplit;
    kbeg = blockIdx.y * per;
    nkt = min(per, nkt - kbeg);
  }

  // per-instruction source pointers (k0 = 0): instruction j of this wave covers LDS rows
  // [16*(wave + NW*j), +16); lane -> row (lane>>2), physical segment (lane&3)
  const __half* src[PER_WAVE];
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int lrow = 16 * (wave + NW * j) + (lane >> 2);          // LDS row within the buffer
    const int seg = (lane & 3) ^ ((lrow >> 2) & 3);               // logical segment fetched into this slot
    const __half* base;
    int trow, limit;
    size_t ld;
    const size_t second = BF16 ? (size_t)HBK : 0;                   // the "second plane" of the bf16 mode: the next 32 columns
    if (lrow < BM) { base = g.A; trow = bm + lrow; limit = g.M; ld = g.lda; }
    else if (lrow < 2 * BM) { base = g.A + (BF16 ? second : g.a_ps); trow = bm + lrow - BM; limit = g.M; ld = g.lda; }
    else if (lrow < 2 * BM + BN) { base = g.W; trow = bn + lrow - 2 * BM; limit = g.N; ld = g.ldw; }
    else { base = g.W + (BF16 ? second : g.w_ps); trow = bn + lrow - 2 * BM - BN; limit = g.N; ld = g.ldw; }
    if (!FULL && trow >= limit) trow = limit - 1;                  // ragged tile: duplicate a valid row
    src[j] = base + (size_t)trow * ld + seg * 8 + (size_t)kbeg * KSTEP;
  }
  auto stage = [&](int buf, int k0) {
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
      __half* dst = smem + (size_t)buf * ROWS * HBK + 16 * (wave + NW * j) * HBK;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + k0),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment address: row = tile row, 16-B segment (2c + lane>>5) ^ swizzle(row); all row offsets used
  // below are multiples of 16, so swizzle(row) = (lane>>2)&3 for every fragment of this lane
  const int frow = lane & 31, sw = (lane >> 2) & 3, hf = lane >> 5;
  const int a_row = wm * (BM / WM) + frow, w_row = 2 * BM + wn * (BN / WN) + frow;

  auto compute = [&](int buf) {
    const __half* base = smem + (size_t)buf * ROWS * HBK;
#pragma unroll
    for (int c = 0; c < HBK / 16; ++c) {
      const int so = ((2 * c + hf) ^ sw) * 8;
      if (BF16) {   // columns k .. k+31 in the first row groups, k+32 .. k+63 in the second
        bf16x8 a0[TM], a1[TM], b0[TN], b1[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          a0[i] = *reinterpret_cast<const bf16x8*>(base + (a_row + i * 32) * HBK + so);
          a1[i] = *reinterpret_cast<const bf16x8*>(base + (BM + a_row + i * 32) * HBK + so);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          b0[j] = *reinterpret_cast<const bf16x8*>(base + (w_row + j * 32) * HBK + so);
          b1[j] = *reinterpret_cast<const bf16x8*>(base + (BN + w_row + j * 32) * HBK + so);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
        continue;
      }
      f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const f16x8*>(base + (a_row + i * 32) * HBK + so);
        al[i] = *reinterpret_cast<const f16x8*>(base + (BM + a_row + i * 32) * HBK + so);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const f16x8*>(base + (w_row + j * 32) * HBK + so);
        bl[j] = *reinterpret_cast<const f16x8*>(base + (BN + w_row + j * 32) * HBK + so);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    }
  };

  if (STAGES == 2) {
    stage(0, 0);
    __syncthreads();  // drains the DMA (vmcnt(0)) and publishes buffer 0
    for (int kt = 0; kt + 1 < nkt; ++kt) {
      stage((kt + 1) & 1, (kt + 1) * KSTEP);
      compute(kt & 1);
      __syncthreads();
    }
    compute((nkt - 1) & 1);
  } else {
    constexpr int AHEAD = STAGES - 1;                          // tiles requested before the first compute
    static_assert(PER_WAVE * (AHEAD - 1) <= 63, "vmcnt is a 6-bit counter");
    // vmcnt(N) immediate: bits [3:0] = N & 15, [15:14] = N >> 4; expcnt/lgkmcnt fields left at "no wait"
    constexpr int KEEP = PER_WAVE * (AHEAD - 1);
    constexpr int WAIT_KEEP = (KEEP & 15) | ((KEEP >> 4) << 14) | 0x0f70;
    constexpr int WAIT_NONE = 0x0f70;
#pragma unroll
    for (int t = 0; t < AHEAD; ++t)
      if (t < nkt) stage(t, t * KSTEP);
    for (int kt = 0; kt < nkt; ++kt) {
      // this wave's pieces of tile kt have landed once at most the pieces of the AHEAD-1 younger tiles are pending
      // (fewer tiles are in flight at the tail: wait for everything there)
      if (kt + AHEAD - 1 < nkt) __builtin_amdgcn_s_waitcnt(WAIT_KEEP); else __builtin_amdgcn_s_waitcnt(WAIT_NONE);
      __builtin_amdgcn_s_barrier();                            // everyone's pieces landed; compute(kt-1) is over everywhere
      __builtin_amdgcn_sched_barrier(0);
      if (kt + AHEAD < nkt) stage((kt + AHEAD) % STAGES, (kt + AHEAD) * KSTEP);   // refills the buffer of tile kt-1
      compute(kt % STAGES);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  if (BF16) {
    // bf16 training GEMMs (fp32 output, optional fp32 residual and ReLU, nothing else fused: launch_gemm_h2 checks): the wave's
    // (BM / WM) x (BN / WN) outputs go through a private LDS strip (the operand buffers are free) and leave row-wise with
    // 16-byte accesses, every residual piece requested before the first store. Straight from the MFMA layout (below) a wave
    // issued 64 four-byte stores per 32 x 32 block and as many residual loads, each batch waiting behind the stores before it.
    constexpr int SH = BM / WM, SW = BN / WN;             // strip rows x columns (floats)
    static_assert((size_t)BM * BN * sizeof(float) <= sizeof(smem), "the strips of all waves fit the operand buffers");
    __syncthreads();                                      // every wave is done reading operand tiles
    float* stg = reinterpret_cast<float*>(smem) + (size_t)wave * (SH * SW);
    const int ncol_ = lane & 31, rsub_ = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[(i * 32 + (r & 3) + 8 * (r >> 2) + rsub_) * SW + j * 32 + ncol_] = acc[i][j][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): the wave's own LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    constexpr int LPR = SW / 4, RPI = 64 / LPR, NK = SH / RPI;   // lanes per staged row, rows per instruction, instructions
    const int rrow = lane / LPR, rc4 = (lane % LPR) * 4;
    const int mrow0 = bm + wm * SH, n0 = bn + wn * SW + rc4;
    const bool ncol_ok = FULL || n0 < g.N;
    float* outp = g.out[0] + (size_t)blockIdx.y * g.part_stride;   // split-K: this block's partial result
    const int ldo = g.ldo[0];
    const float relu_lo = g.relu ? 0.f : -INFINITY;
    float4 res[NK];
    if (g.resid) {
#pragma unroll
      for (int k = 0; k < NK; ++k) {   // clamped, unconditional: a branch around a load brings the conservative vmcnt(0) back
        const int m = min(mrow0 + k * RPI + rrow, g.M - 1), nc = ncol_ok ? n0 : 0;
        res[k] = *reinterpret_cast<const float4*>(g.resid + (size_t)m * g.ldr + nc);
      }
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int rl = k * RPI + rrow, m = mrow0 + rl;
      float4 v = *reinterpret_cast<const float4*>(stg + rl * SW + rc4);
      v.x = fmaxf(v.x * acc_scale, relu_lo); v.y = fmaxf(v.y * acc_scale, relu_lo);
      v.z = fmaxf(v.z * acc_scale, relu_lo); v.w = fmaxf(v.w * acc_scale, relu_lo);
      if (g.resid) { v.x = res[k].x + v.x; v.y = res[k].y + v.y; v.z = res[k].z + v.z; v.w = res[k].w + v.w; }
      if (ncol_ok && (FULL || m < g.M)) *reinterpret_cast<float4*>(outp + (size_t)m * ldo + n0) = v;
    }
    return;
  }
  // split-precision launches (a few hundred to a few thousand rows in flight; every fused extra of the search path): the
  // LDS-transposed epilogue of the 256x256 kernels on this kernel's tile shape — 16-byte plane / fp32 stores, residual pieces
  // requested before the first store, row scales from LDS. (Until round 4 this kernel stored straight from the MFMA result
  // layout: two-byte plane stores and four-byte loads, 64 of each per 32 x 32 block.)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  h2_epilogue_256<FULL, TM, TN, WM, WN, BM, BN>(g, acc, smem, wave, lane_e, bm, bn, wm, wn, g.row_ssq ? rs_tile : nullptr, acc_scale);
}

// ---- shared epilogue of the 256x256 kernels: transpose through LDS, then row-wise 16-byte global accesses --
// A wave stages 64 rows at a time: the strips of the 8 waves fill both operand buffers (128 KB).
// One strip (64 rows of a wave's 128 x 64 outputs) of the epilogue below. The strip index is a template parameter: the
// accumulators are indexed with it, and a strip loop the compiler declines to unroll — it did once the output paths had
// grown — puts all 128 of them into scratch.
template <bool FULL, int TM, int TN, int WM, int WN, int strip, int BM = 256, int BN = 256, bool BOUT = false>
__device__ __forceinline__ void h2_epilogue_strip(const GemmH2Args& g, f32x16 (&acc)[TM][TN], float* stg, int lane, int bm, int bn,
                                                  int wm, int wn, const float* rs_tile, float acc_scale, int Mlim) {
  constexpr int SH = 64, SW = TN * 32;
  const int ncol = lane & 31, rsub = 4 * (lane >> 5);
#pragma unroll
    for (int ii = 0; ii < SH / 32; ++ii)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          stg[(ii * 32 + (r & 3) + 8 * (r >> 2) + rsub) * SW + j * 32 + ncol] = acc[strip * (SH / 32) + ii][j][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the wave's own LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    const int mrow0 = bm + wm * (BM / WM) + strip * SH;
    // How the output paths are written matters more than what they compute (ISA of the first version, measured with the
    // per-tile stamps of tools/gemm_tile_timeline.py: 6 us per tile on an idle chip, 10-14 us in a full launch, where the
    // same stores take 2.5 us in tools/store_probe.hip):
    //  * vmcnt counts loads AND stores in issue order. The residual pieces used to be loaded behind runtime branches in the
    //    same loop as the stores; at every join the compiler no longer knows what is pending and waits for vmcnt(0) — and
    //    it kept that wait on the path WITHOUT a residual too: each of the 32 stores of a wave waited for the previous
    //    one's write acknowledgement from the L2. Now the variants with and without a residual are separate
    //    instantiations (RES), a variant without one contains no global load at all, and one with a residual requests every
    //    piece of the strip with unconditional loads before the strip's first store. (Requesting strip 1's pieces before
    //    strip 0's stores as well — they otherwise complete behind them — needs registers the kernel does not have: all
    //    of them spill 120-290 B, half of them fits and measured -0.2 % same-box.)
    //  * every LDS read of a batch (staged rows, row scales) is issued before the first use (explicit arrays + a scheduling
    //    barrier; the compiler otherwise sinks each read next to its use behind an lgkmcnt(0)).
    auto planes_path = [&](auto res_tag) __attribute__((always_inline)) {
      // ---- f16-plane output (FF intermediate, residual stream): 8 columns per lane, 8 rows per pass
      constexpr bool RES = decltype(res_tag)::value;
      constexpr int LPR = SW / 8, RPI = 64 / LPR, NK = SH / RPI, KB = RES ? 2 : 4;   // (the residual pieces hold 64 registers)
      const int rrow = lane / LPR, rc8 = (lane % LPR) * 8;
      const int n0 = bn + wn * (BN / WN) + rc8;
      const bool ncol_ok = FULL || (n0 < g.N);          // N % 32 == 0
      const float ps = g.plane_scale;
      float amax = 0.f;                                 // split_f16's range check: one max per element, one compare per strip
      uint4 rh[RES ? NK : 1], rl_[RES ? NK : 1];
      if (RES) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          // unconditional loads from a clamped (always valid) address: a branch around a load brings the conservative
          // vmcnt(0) back; rows / columns past the limits are never stored
          const int m = min(mrow0 + k * RPI + rrow, Mlim - 1), nc = ncol_ok ? n0 : 0;
          rh[k] = *reinterpret_cast<const uint4*>(g.resid_h + (size_t)m * g.ldrh + nc);
          rl_[k] = *reinterpret_cast<const uint4*>(g.resid_h + g.r_ps + (size_t)m * g.ldrh + nc);
        }
      }
#pragma unroll
      for (int kb = 0; kb < NK; kb += KB) {
        float4 sa[KB], sb[KB];
        float scs[KB];
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          const int rl = (kb + k) * RPI + rrow;
          sa[k] = *reinterpret_cast<const float4*>(stg + rl * SW + rc8);
          sb[k] = *reinterpret_cast<const float4*>(stg + rl * SW + rc8 + 4);
          scs[k] = rs_tile ? rs_tile[wm * (BM / WM) + strip * SH + rl] : 1.0f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          const int rl = (kb + k) * RPI + rrow, m = mrow0 + rl;
          const bool ok = ncol_ok && (FULL || m < Mlim);
          const float sc = rs_tile ? acc_scale * scs[k] : acc_scale;
          // pairs of columns as 2-vectors: gfx950 multiplies / adds / fmas two fp32 per instruction (v_pk_*_f32) and
          // converts two floats to a packed f16 pair in one (v_cvt_pk_f16_f32, round to nearest) — about half the VALU
          // work of the element-by-element form, which also spent a shift + or per pair on packing
          f32x2 v[4];
          *reinterpret_cast<float4*>(&v[0]) = sa[k];
          *reinterpret_cast<float4*>(&v[2]) = sb[k];
          f16x2 h[4], l[4];
          f32x2 ss2 = {0.f, 0.f};
          float am = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f32x2 x = v[e] * sc;
            if (g.relu) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); }
            if (RES) {   // the residual from its planes: (hi + lo) / X_PLANE_SCALE, both steps exact in fp32
              x += (__builtin_convertvector(reinterpret_cast<const f16x2*>(&rh[RES ? kb + k : 0])[e], f32x2) +
                    __builtin_convertvector(reinterpret_cast<const f16x2*>(&rl_[RES ? kb + k : 0])[e], f32x2)) * (1.0f / X_PLANE_SCALE);
            }
            ss2 += x * x;
            f32x2 xs = x * ps;
            am = fmaxf(am, fmaxf(fabsf(xs.x), fabsf(xs.y)));
            xs.x = __builtin_amdgcn_fmed3f(xs.x, -65504.f, 65504.f);
            xs.y = __builtin_amdgcn_fmed3f(xs.y, -65504.f, 65504.f);
            h[e] = __builtin_convertvector(xs, f16x2);
            l[e] = __builtin_convertvector(xs - __builtin_convertvector(h[e], f32x2), f16x2);
          }
          float ss = ss2.x + ss2.y;
          if (ok) {
            *reinterpret_cast<uint4*>(g.out_h + (size_t)m * g.ldoh + n0) = *reinterpret_cast<uint4*>(h);
            *reinterpret_cast<uint4*>(g.out_h + g.o_ps + (size_t)m * g.ldoh + n0) = *reinterpret_cast<uint4*>(l);
            amax = fmaxf(amax, am);
          }
          if (g.ssq_out) {   // the LPR lanes of a staged row hold this wave's 64 columns of output row m
            if (!ok) ss = 0.f;
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            if ((lane % LPR) == 0 && (FULL || m < Mlim)) atomicAdd(g.ssq_out + m, ssq_to_fix(ss, g.sat));
          }
        }
      }
      if (amax > 65504.f && g.sat) *g.sat = 1u;
    };
    auto fp32_path = [&](auto res_tag, auto kv_tag) __attribute__((always_inline)) {
      // ---- fp32 output (q, K/V cache rows, logits, fp32 residual stream of callers without planes). KVMAP: the K/V-cache
      // element map of the sequential steps (an integer division per store); the variant without it keeps the store loop
      // free of branches (ReLU as a max against 0 or -inf, the row scale always multiplied): ~10 instructions per store
      // instead of ~30 with five branches
      constexpr bool RES = decltype(res_tag)::value, KVMAP = decltype(kv_tag)::value;
      const float relu_lo = g.relu ? 0.f : -INFINITY;
      constexpr int LPR = SW / 4, RPI = 64 / LPR, NK = SH / RPI, KB = RES ? 4 : 8;   // lanes per staged row, rows per read instruction
      const int rrow = lane / LPR, rc4 = (lane % LPR) * 4;
      const int n0 = bn + wn * (BN / WN) + rc4;           // first of this lane's 4 consecutive output columns
      const bool ncol_ok = FULL || (n0 < g.N);            // N % 4 == 0 is guaranteed (N % 32 == 0)
      const int oi = ncol_ok ? n0 / g.split_n : 0, on = n0 - oi * g.split_n;
      // (selects, not g.out[oi]: a variable index into the by-value argument struct makes the compiler keep a private copy
      // of all of it in scratch)
      float* outp = (oi == 0 ? g.out[0] : oi == 1 ? g.out[1] : g.out[2]) + (size_t)blockIdx.y * g.part_stride;   // split-K: this block's partial result
      const int ldo = oi == 0 ? g.ldo[0] : oi == 1 ? g.ldo[1] : g.ldo[2];
      float4 res[RES ? NK : 1];
      if (RES) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          const int m = min(mrow0 + k * RPI + rrow, Mlim - 1), nc = ncol_ok ? n0 : 0;   // clamped, unconditional (see planes_path)
          res[k] = *reinterpret_cast<const float4*>(g.resid + (size_t)m * g.ldr + nc);
        }
      }
#pragma unroll
      for (int kb = 0; kb < NK; kb += KB) {
        float4 st[KB];
        float scs[KB];
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          const int rl = (kb + k) * RPI + rrow;
          st[k] = *reinterpret_cast<const float4*>(stg + rl * SW + rc4);
          scs[k] = rs_tile ? rs_tile[wm * (BM / WM) + strip * SH + rl] : 1.0f;   // fused RMSNorm row scale
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          const int rl = (kb + k) * RPI + rrow, m = mrow0 + rl;
          float4 v = st[k];
          const float sc = acc_scale * scs[k];
          v.x = fmaxf(v.x * sc, relu_lo); v.y = fmaxf(v.y * sc, relu_lo); v.z = fmaxf(v.z * sc, relu_lo); v.w = fmaxf(v.w * sc, relu_lo);
          if (RES) { const float4 r = res[RES ? kb + k : 0]; v.x = r.x + v.x; v.y = r.y + v.y; v.z = r.z + v.z; v.w = r.w + v.w; }
          if (ncol_ok && (FULL || m < Mlim)) {
            const size_t off = KVMAP ? out_off(g, oi, m, ldo, on) : (size_t)m * ldo + on;
            *reinterpret_cast<float4*>(outp + off) = v;
          }
        }
      }
    };
    // ---- bf16 outputs (BOUT instantiations = the bf16 256x256 kernel of the training step; GemmH2Args::out_b / out_bt):
    // the staged rows leave as bf16 rows (8 columns = one 16-byte store per lane) and, after the final values have gone back
    // into the strip, as the transposed copy: a lane owns one column of a 32-row half of the strip (conflict-free 4-byte LDS
    // reads, bank = column), packs row pairs and stores 4 x 16 bytes; the two halves complete a column's 128-byte line.
    auto bf16_path = [&](auto mask_tag) __attribute__((always_inline)) {
      constexpr bool MASK = decltype(mask_tag)::value;
      constexpr int LPR = SW / 8, RPI = 64 / LPR, NK = SH / RPI, KB = MASK ? 2 : 4;
      const int rrow = lane / LPR, rc8 = (lane % LPR) * 8;
      const int n0 = bn + wn * (BN / WN) + rc8;
      const float relu_lo = g.relu ? 0.f : -INFINITY;
      __bf16* ob = reinterpret_cast<__bf16*>(g.out_b);
      float* of = g.out[0];
      float4 ma[MASK ? NK : 1], mb[MASK ? NK : 1];
      if (MASK) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {   // every mask piece of the strip requested before the first store (see planes_path)
          const float* mp = g.mask_src + (size_t)(mrow0 + k * RPI + rrow) * g.ldmask + n0;
          ma[k] = *reinterpret_cast<const float4*>(mp);
          mb[k] = *reinterpret_cast<const float4*>(mp + 4);
        }
      }
#pragma unroll
      for (int kb = 0; kb < NK; kb += KB) {
        float4 sa[KB], sb[KB];
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          const int rl = (kb + k) * RPI + rrow;
          sa[k] = *reinterpret_cast<const float4*>(stg + rl * SW + rc8);
          sb[k] = *reinterpret_cast<const float4*>(stg + rl * SW + rc8 + 4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          const int rl = (kb + k) * RPI + rrow, m = mrow0 + rl;
          float v[8] = {sa[k].x, sa[k].y, sa[k].z, sa[k].w, sb[k].x, sb[k].y, sb[k].z, sb[k].w};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] * acc_scale, relu_lo);
          if (MASK) {
            const float mk[8] = {ma[MASK ? kb + k : 0].x, ma[MASK ? kb + k : 0].y, ma[MASK ? kb + k : 0].z, ma[MASK ? kb + k : 0].w,
                                 mb[MASK ? kb + k : 0].x, mb[MASK ? kb + k : 0].y, mb[MASK ? kb + k : 0].z, mb[MASK ? kb + k : 0].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
          }
          if (g.out_bt) {   // the final values back into the strip for the transposed pass
            *reinterpret_cast<float4*>(stg + rl * SW + rc8) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(stg + rl * SW + rc8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
          }
          if (ob) {
            __bf16 b[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) b[e] = (__bf16)v[e];
            *reinterpret_cast<uint4*>(ob + (size_t)m * g.ldob + n0) = *reinterpret_cast<uint4*>(b);
          }
          if (of) {
            *reinterpret_cast<float4*>(of + (size_t)m * g.ldo[0] + n0) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(of + (size_t)m * g.ldo[0] + n0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
          }
        }
      }
      if (g.out_bt) {
        __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): the rewritten strip has landed
        __builtin_amdgcn_wave_barrier();
        __bf16* obt = reinterpret_cast<__bf16*>(g.out_bt);
        const int c = lane & 31, hrow = 32 * (lane >> 5);
#pragma unroll
        for (int cc = 0; cc < SW; cc += 32) {
          float col[32];
#pragma unroll
          for (int r = 0; r < 32; ++r) col[r] = stg[(hrow + r) * SW + cc + c];
          __bf16 b[32];
#pragma unroll
          for (int r = 0; r < 32; ++r) b[r] = (__bf16)col[r];
          __bf16* dst = obt + (size_t)(bn + wn * (BN / WN) + cc + c) * g.ldobt + mrow0 + hrow;
#pragma unroll
          for (int j = 0; j < 4; ++j) reinterpret_cast<uint4*>(dst)[j] = reinterpret_cast<const uint4*>(b)[j];
        }
      }
    };
    if (BOUT && (g.out_b || g.out_bt)) {
      if (g.mask_src) bf16_path(std::true_type{}); else bf16_path(std::false_type{});
    } else if (g.out_h) {
      if (g.resid_h) planes_path(std::true_type{}); else planes_path(std::false_type{});
    } else {
      if (g.rm_B) { if (g.resid) fp32_path(std::true_type{}, std::true_type{}); else fp32_path(std::false_type{}, std::true_type{}); }
      else if (g.resid) fp32_path(std::true_type{}, std::false_type{});      // (training GEMMs with an fp32 residual)
      else fp32_path(std::false_type{}, std::false_type{});
    }
    __builtin_amdgcn_wave_barrier();                  // stg is rewritten by the next strip
}

template <bool FULL, int TM, int TN, int WM, int WN, int BM, int BN, bool BOUT>
__device__ __forceinline__ void h2_epilogue_256(const GemmH2Args& g, f32x16 (&acc)[TM][TN], __half* smem, int wave,
                                                int lane, int bm, int bn, int wm, int wn, const float* rs_tile,
                                                float acc_scale) {
  constexpr int SH = 64;
  // The MFMA C layout gives a lane one column and 16 scattered rows per tile, so a direct epilogue is
  // 128 dword stores (+128 dword residual loads) per lane in 16-load batches, each batch exposing a full
  // memory latency: ~12 us per tile without and ~30 us with the residual, against ~55 us of K-loop at
  // K = 768. Here each wave stages SH x 64 outputs at a time in its private strip of the (now idle)
  // operand LDS and streams them out row-wise with 16-byte accesses. The epilogue is bound by the NUMBER of
  // memory instructions and by its VALU work, not by bytes (measured: two extra 8-byte plane stores per float4
  // tripled it; ~20 VALU ops per plane element cost ~5 us per tile; a global load of the fused-RMSNorm row scales
  // cost a full memory latency per half), so the f16-plane outputs give a lane 8 consecutive columns = one 16-byte
  // store per plane, the row scales come from LDS (rs_tile, filled before the K-loop) and the saturation check is
  // one max per element plus one compare per half.
  __syncthreads();                                   // all waves are done reading operand tiles
  const int Mlim = g.m_dev ? min(*g.m_dev - g.m_base, g.M) : g.M;   // packed encoder: rows past the live count are never stored
  constexpr int SW = TN * 32;                         // staged row width (floats)
  constexpr int NS = TM * 32 / SH;                    // strips per wave
  float* stg = reinterpret_cast<float*>(smem) + wave * (SH * SW);
  const int ncol = lane & 31, rsub = 4 * (lane >> 5);
  static_assert(NS == 1 || NS == 2, "one or two strips of 64 rows per wave");
  (void)ncol; (void)rsub;
  h2_epilogue_strip<FULL, TM, TN, WM, WN, 0, BM, BN, BOUT>(g, acc, stg, lane, bm, bn, wm, wn, rs_tile, acc_scale, Mlim);
  if (NS == 2) h2_epilogue_strip<FULL, TM, TN, WM, WN, NS - 1, BM, BN, BOUT>(g, acc, stg, lane, bm, bn, wm, wn, rs_tile, acc_scale, Mlim);
}

// ---- ping-pong 256x256 variant -------------------------------------------------------------------------
// 256x256x32 tiles, LDS-DMA feed into two 64-KB operand buffers, LDS-transposed epilogue. The 8 waves are two groups of
// four (wm = 0 / 1), one wave of each group per SIMD. A K-tile is two phases (one per 16-column k-chunk), each phase = a
// load segment (4 W + 8 A fragment ds_reads) and an MFMA segment (24 MFMAs), separated by raw s_barriers:
//       L_p | barrier | M_p | barrier | L_p+1 | ...
// Group 1 runs one barrier behind group 0, so on every SIMD one wave is in its MFMA segment while the other
// issues its loads: the LDS read latency (which an in-phase kernel pays with the MFMA pipe idle) is hidden behind the
// other group's MFMAs. The 8 LDS-DMA pieces of tile t+1 are issued in the MFMA shadows of M_0 of tile t, retired by
// `s_waitcnt vmcnt(0)` at the end of L_1 and first read in L_0 of tile t+1 (one phase after the wait, as the staggered
// groups need one barrier more); each L ends with lgkmcnt(0) BEFORE its barrier, so a buffer's last reads are retired
// before the other group starts overwriting it. (Rounds 1-3 ran four phases of 12 MFMAs: twice the barriers, 5076 instead
// of 4224 cycles per K-tile.)
// BF16 = true (training GEMMs, RPR_PREC_BF16): one bf16 plane per operand; the LDS rows of the lo planes hold the NEXT
// 32 columns of K instead, a K-tile is 64 deep and a phase issues 8 v_mfma_f32_32x32x16_bf16 (slice 0 x slice 0 and
// slice 1 x slice 1) on the same fragment reads — see gemm_h2_dma_kernel.
// TRACE = true: diagnostic instantiation (tools/gemm_trace_pp.py, tools/gemm_tile_timeline.py) that stamps cycle counters
// of block 0 into g.trace; never launched by a search.
// Schedules that were measured and removed (DESIGN.md "tried and rejected"; the code is in the history): a phase skew
// between the persistent blocks, a per-round barrier between the blocks of an XCD, and the next tile's first K-tile
// prefetched under the epilogue — all of them slower or equal on the power-capped headline step.
// The body is textually shared by two kernels (gemm_h2_pp_body.inc): gemm_h2_pp_kernel (one product, its arguments by value)
// and gemm_h2_pp_group_kernel (a table of products in device memory, blockIdx.y = product; see there).
template <bool FULL, bool TRACE = false, bool BF16 = false>
__global__ __launch_bounds__(512, 2) void gemm_h2_pp_kernel(GemmH2Args g, int tiles_m, int tiles_n) {
#include "gemm_h2_pp_body.inc"
}

// Grouped launch (bf16 training GEMMs): the weight-gradient products of one transformer layer, dW[N, K] = dY^T X for its six
// (encoder: four) linear layers, are 9 .. 36 output tiles each with a reduction over all rows of the batch. One launch per
// product cannot fill the chip without split-K (5-K-tile loops, fp32 partials, a reduce pass: 240 TF/s, and every
// 252-block launch takes the whole chip from the input-gradient chain on the main stream); here ONE launch walks all
// products of the layer, one block per tile (at most 384 tiles), each running the whole reduction of its tile in a single
// K-loop (t5-base decoder layer: 126 tiles x 128 K-tiles of 64 rows). A block reads (product, tile) from an assignment
// table and its product's argument struct from a table in device memory (scalar loads; a struct assembled in the kernel
// from by-value arrays stayed in scratch, 344 bytes per lane, because the epilogue indexes out[] / ldo[] at run time).
// Both tables are written by gemm_group_table_kernel from by-value arguments on the same stream, directly in front of the
// product launch: no host memory whose lifetime would have to outlast the enqueue.
// Assignment (launch_gemm_h2_group): workgroup i of a 1-D grid runs on XCD i % 8. Every product's tile grid is cut into
// super-tiles of up to 4 x 4 tiles, the tiles are listed super-tile by super-tile and every XCD takes an equal run of that
// list: the blocks of a super-tile run side by side on one XCD and share their A and W panels in its L2 (a 4 x 3 super-tile
// streams 7 panels for 12 tiles). With the products' tiles in
// plain row-major order over the XCDs the launch fetched 773 MB from the fabric for 276 MB of distinct operands
// (rocprofv3 FETCH_SIZE; 1.03 GB go through LDS).
template <bool FULL>
__global__ __launch_bounds__(512, 2) void gemm_h2_pp_group_kernel(const GemmH2Args* __restrict__ table, const int* __restrict__ assign) {
  constexpr bool TRACE = false, BF16 = true;
  const int asg = assign[blockIdx.x];
  if (asg < 0) return;
  const GemmH2Args& g = table[asg >> 16];
  const int tiles_m = (g.M + 255) >> 8, tiles_n = (g.N + 255) >> 8;
#define PP_GROUP_TILE (asg & 0xffff)
#include "gemm_h2_pp_body.inc"
#undef PP_GROUP_TILE
}

struct GroupAssign { int n; int v[GemmGroupArgs::MAX_BLOCKS]; };
__global__ void gemm_group_table_kernel(GemmGroupArgs p, GroupAssign a, GemmH2Args* __restrict__ table, int* __restrict__ assign) {
  const int i = threadIdx.x;
  for (int k = i; k < a.n; k += blockDim.x) assign[k] = a.v[k];
  if (i >= p.n) return;
  GemmH2Args g{};
  g.A = p.A[i]; g.W = p.W[i]; g.lda = p.lda; g.ldw = p.ldw;
  g.out[0] = g.out[1] = g.out[2] = p.out[i];
  g.ldo[0] = g.ldo[1] = g.ldo[2] = p.ldo[i];
  g.M = p.M[i]; g.N = p.N[i]; g.K = p.K; g.split_n = p.N[i];
  g.acc_scale = 1.0f; g.plane_scale = 1.0f; g.bf16 = 1;
  table[i] = g;
}

// ---- skinny variant: M <= 400 rows (one to a few dozen queries in flight) ---------------------------------------
// Such a launch is a weight stream: 2.4 MB of W for 10 live rows. The tile kernels give it N/64..N/32 blocks
// that each walk all of K through ONE LDS-DMA stream (~25 GB/s per CU): 18-29 us per launch, 2385 launches per
// search. Here a block is a 32x32 output tile whose four waves split K between them (wave w takes K-tiles
// w, w+4, ...), each with a private 4-stage LDS ring fed by its own LDS-DMA stream and no block barrier in the
// loop (a wave only reads what it loaded itself: s_waitcnt vmcnt is the whole synchronisation). The four partial
// accumulators are added in the fixed order 0..3 through LDS and wave 0 runs the epilogue — deterministic.
template <bool FULL>
__global__ __launch_bounds__(256, 1) void gemm_h2_skinny_kernel(GemmH2Args g, int tiles_m, int tiles_n) {
  // per-tensor dynamic plane scales (training): read from the device; g itself must stay untouched — a kernel that writes
  // to its by-value argument struct gets a private copy of all 320 bytes in scratch (measured: +20 % per launch)
  const float acc_scale = g.dyn_a ? 1.0f / (dyn_plane_scale(*g.dyn_a) * dyn_plane_scale(*g.dyn_b)) : g.acc_scale;
  constexpr int BM = 32, BN = 32, ST = 4, ROWS = 2 * (BM + BN), PIECES = ROWS / 16;   // 8 KB per stage, 8 pieces
  __shared__ __attribute__((aligned(16))) __half smem[4 * ST * ROWS * HBK];             // 128 KB
  const int tile = blockIdx.x, tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int bm = tm * BM, bn = tn * BN;
  // compacted stage / tail job: M is the capacity, the live row count is on the device (and, for one kernel of a gated
  // group, decides whether this kernel runs at all); row tiles past the live rows exit at once
  int Mlive = g.M;
  if (g.m_dev) {
    const int md = *g.m_dev;
    if (g.live_hi > 0 && (md <= g.live_lo || md > g.live_hi)) return;
    Mlive = min(md, g.M);
  }
  if (bm >= Mlive) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __half* wsm = smem + (size_t)wave * ST * ROWS * HBK;
  // Epilogue operands of wave 0, requested BEFORE the K walk so that their memory latency passes under it (a launch of
  // this kernel is a chain of latencies: ~1.5 us each for the row scales and for the residual planes when they were
  // loaded after the reduction): the fused-RMSNorm sums of squares of the lane's 16 rows and the residual of its 16
  // outputs (column bn + lane % 32, rows 4 * (lane / 32) + (r & 3) + 8 * (r >> 2)).
  unsigned long long e_ssq[16];
  unsigned int e_rh[16], e_rl[16];   // raw f16 bits, one 32-bit register each: as __half pairs the compiler packed them on arrival,
                                          // i.e. waited for every pair of loads before requesting the next (16 serial latencies in front of the K walk)
  float e_rf[16];
  if (wave == 0) {
    const int n = bn + (lane & 31), rsub = 4 * (lane >> 5);
    const bool nok = FULL || n < g.N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = bm + rsub + (r & 3) + 8 * (r >> 2);
      const bool mok = m < Mlive;
      e_ssq[r] = (g.row_ssq && mok) ? g.row_ssq[m] : 0ull;
      e_rf[r] = (g.resid && mok && nok) ? g.resid[(size_t)m * g.ldr + n] : 0.f;
      e_rh[r] = (g.resid_h && mok && nok) ? (unsigned int)reinterpret_cast<const unsigned short*>(g.resid_h)[(size_t)m * g.ldrh + n] : 0u;
      e_rl[r] = (g.resid_h && mok && nok) ? (unsigned int)reinterpret_cast<const unsigned short*>(g.resid_h)[g.r_ps + (size_t)m * g.ldrh + n] : 0u;
    }
  }

  const __half* src[PIECES];
#pragma unroll
  for (int j = 0; j < PIECES; ++j) {
    const int lrow = 16 * j + (lane >> 2);
    const int seg = (lane & 3) ^ ((lrow >> 2) & 3);
    const __half* base;
    int trow, limit;
    size_t ld;
    if (lrow < BM) { base = g.A; trow = bm + lrow; limit = g.M; ld = g.lda; }
    else if (lrow < 2 * BM) { base = g.A + g.a_ps; trow = bm + lrow - BM; limit = g.M; ld = g.lda; }
    else if (lrow < 2 * BM + BN) { base = g.W; trow = bn + lrow - 2 * BM; limit = g.N; ld = g.ldw; }
    else { base = g.W + g.w_ps; trow = bn + lrow - 2 * BM - BN; limit = g.N; ld = g.ldw; }
    if (!FULL && trow >= limit) trow = limit - 1;
    src[j] = base + (size_t)trow * ld + seg * 8;
  }
  auto stage = [&](int buf, int k0) {
#pragma unroll
    for (int j = 0; j < PIECES; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + k0),
                                       (__attribute__((address_space(3))) void*)(wsm + (size_t)buf * ROWS * HBK + 16 * j * HBK),
                                       16, 0, 0);
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int frow = lane & 31, sw = (lane >> 2) & 3, hf = lane >> 5;
  auto compute = [&](int buf) {
    const __half* base = wsm + (size_t)buf * ROWS * HBK;
#pragma unroll
    for (int c = 0; c < HBK / 16; ++c) {
      const int so = ((2 * c + hf) ^ sw) * 8;
      const f16x8 ah = *reinterpret_cast<const f16x8*>(base + frow * HBK + so);
      const f16x8 al = *reinterpret_cast<const f16x8*>(base + (BM + frow) * HBK + so);
      const f16x8 bh = *reinterpret_cast<const f16x8*>(base + (2 * BM + frow) * HBK + so);
      const f16x8 bl = *reinterpret_cast<const f16x8*>(base + (2 * BM + BN + frow) * HBK + so);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    }
  };
  const int nkt = g.K / HBK;
  const int mine = wave < nkt ? (nkt - wave + 3) / 4 : 0;         // K-tiles wave, wave + 4, ...
  constexpr int KEEP = PIECES * (ST - 2);                         // 16 pieces of the two younger tiles may be pending
  constexpr int WAIT_KEEP = (KEEP & 15) | ((KEEP >> 4) << 14) | 0x0f70, WAIT_NONE = 0x0f70;
#pragma unroll
  for (int t = 0; t < ST - 1; ++t)
    if (t < mine) stage(t, (wave + 4 * t) * HBK);
  for (int i = 0; i < mine; ++i) {
    if (i + ST - 2 < mine) __builtin_amdgcn_s_waitcnt(WAIT_KEEP); else __builtin_amdgcn_s_waitcnt(WAIT_NONE);
    __builtin_amdgcn_sched_barrier(0);
    if (i + ST - 1 < mine) stage((i + ST - 1) % ST, (wave + 4 * (i + ST - 1)) * HBK);   // buffer of tile i-1: its reads are done
    compute(i % ST);
    __builtin_amdgcn_sched_barrier(0);
  }
  // reduce the four waves' partial tiles in fixed order through LDS (the operand rings are idle now)
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int r = 0; r < 16; ++r)
    acc[r] = ((red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) + red[(2 * 16 + r) * 64 + lane]) +
             red[(3 * 16 + r) * 64 + lane];
  const int n = bn + (lane & 31), rsub = 4 * (lane >> 5);
  const bool nok = FULL || n < g.N;
  const int oi = nok ? n / g.split_n : 0, on = n - oi * g.split_n;
  float* outp = g.out[oi];
  const int ldo = g.ldo[oi];
  float ssr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = bm + rsub + (r & 3) + 8 * (r >> 2);
    const bool mok = m < Mlive, ok = nok && mok;
    float v = acc[r] * acc_scale;
    if (g.row_ssq && mok) v *= ssq_rsqrt(e_ssq[r], g.inv_d_fix, g.eps);
    if (g.relu) v = fmaxf(v, 0.f);
    if (g.resid && ok) v = e_rf[r] + v;
    if (g.resid_h && ok) v = x_from_planes(__ushort_as_half((unsigned short)e_rh[r]), __ushort_as_half((unsigned short)e_rl[r])) + v;
    if (ok) {
      if (g.out_h) {
        __half hi, lo;
        split_f16(v * g.plane_scale, hi, lo, g.sat);
        g.out_h[(size_t)m * g.ldoh + n] = hi;
        g.out_h[g.o_ps + (size_t)m * g.ldoh + n] = lo;
        v = (__half2float(hi) + __half2float(lo)) / g.plane_scale;
      } else {
        outp[out_off(g, oi, m, ldo, on)] = v;
      }
    }
    ssr[r] = ok ? v * v : 0.f;
  }
  if (g.ssq_out) {   // wave-uniform branch. The 16 butterflies are independent: issued level by level they overlap
                     // (one chain after the other cost ~4 us per launch — 12 % of a single-query search)
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < 16; ++r) ssr[r] += __shfl_xor(ssr[r], o, 64);
    if ((lane & 31) == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = bm + rsub + (r & 3) + 8 * (r >> 2);
        if (m < Mlive) atomicAdd(g.ssq_out + m, ssq_to_fix(ssr[r], g.sat));
      }
    }
  }
}

// ---- skinny variant for at most 32 rows (ONE query in flight: the beams of a step, its encoder tokens) ------------------
// The 32-row skinny kernel gives a 768 x 768 weight 24 blocks, and a wave's ring holds three of its six K-tiles (8-KB
// stages: 32 activation + 32 weight rows, two planes): two memory round trips per launch, 11 us, 413 launches = 85 % of a
// single-query search. Here a block is a 16 x 16 output tile (v_mfma_f32_16x16x32_f16: a K-tile of 32 is ONE MFMA per
// product term): twice the blocks, 4-KB stages, six stages per wave = all of K = 768 in flight at once (K = 3072: a ring of
// six). blockIdx.y = 16-row tile (one for a step's 10 beams, two for an encoder of 17 .. 32 tokens). Same K split over the
// four waves, same fixed-order reduction, same fused epilogue arithmetic as the 32-row kernel.
template <bool FULL>
__global__ __launch_bounds__(256, 1) void gemm_h2_skinny16_kernel(GemmH2Args g, int tiles_n) {
  const float acc_scale = g.dyn_a ? 1.0f / (dyn_plane_scale(*g.dyn_a) * dyn_plane_scale(*g.dyn_b)) : g.acc_scale;
  constexpr int BT = 16, ST = 6, ROWS = 4 * BT, PIECES = ROWS / 16;   // 4 KB per stage: A hi, A lo, W hi, W lo x 16 rows
  __shared__ __attribute__((aligned(16))) __half smem[4 * ST * ROWS * HBK];   // 96 KB
  const int bn = blockIdx.x * BT, bm = blockIdx.y * BT;
  int Mlive = g.M;
  if (g.m_dev) {
    const int md = *g.m_dev;
    if (g.live_hi > 0 && (md <= g.live_lo || md > g.live_hi)) return;
    Mlive = min(md, g.M);
  }
  if (bm >= Mlive) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __half* wsm = smem + (size_t)wave * ST * ROWS * HBK;
  // epilogue operands of wave 0, requested before the K walk: lane = column bn + lane % 16, rows 4 (lane / 16) + r
  unsigned long long e_ssq[4];
  unsigned int e_rh[4], e_rl[4];   // raw f16 bits, one 32-bit register each: as __half pairs the compiler packed them on arrival,
                                          // i.e. waited for every pair of loads before requesting the next (16 serial latencies in front of the K walk)
  float e_rf[4];
  if (wave == 0) {
    const int n = bn + (lane & 15), rsub = 4 * (lane >> 4);
    const bool nok = FULL || n < g.N;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = bm + rsub + r;
      const bool mok = m < Mlive;
      e_ssq[r] = (g.row_ssq && mok) ? g.row_ssq[m] : 0ull;
      e_rf[r] = (g.resid && mok && nok) ? g.resid[(size_t)m * g.ldr + n] : 0.f;
      e_rh[r] = (g.resid_h && mok && nok) ? (unsigned int)reinterpret_cast<const unsigned short*>(g.resid_h)[(size_t)m * g.ldrh + n] : 0u;
      e_rl[r] = (g.resid_h && mok && nok) ? (unsigned int)reinterpret_cast<const unsigned short*>(g.resid_h)[g.r_ps + (size_t)m * g.ldrh + n] : 0u;
    }
  }
  const __half* src[PIECES];
#pragma unroll
  for (int j = 0; j < PIECES; ++j) {
    const int lrow = 16 * j + (lane >> 2);
    const int seg = (lane & 3) ^ ((lrow >> 2) & 3);
    const bool is_a = j < 2, second = (j & 1) != 0;
    const __half* base = is_a ? g.A : g.W;
    const size_t plane = second ? (is_a ? g.a_ps : g.w_ps) : 0;
    const int limit = is_a ? g.M : g.N;
    const size_t ld = is_a ? (size_t)g.lda : (size_t)g.ldw;
    int trow = (is_a ? bm : bn) + (lane >> 2);
    if (trow >= limit) trow = limit - 1;                 // activation rows past M: clamped, their outputs are never stored
    src[j] = base + plane + (size_t)trow * ld + seg * 8;
  }
  auto stage = [&](int buf, int k0) {
#pragma unroll
    for (int j = 0; j < PIECES; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + k0),
                                       (__attribute__((address_space(3))) void*)(wsm + (size_t)buf * ROWS * HBK + 16 * j * HBK),
                                       16, 0, 0);
  };
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, kg = lane >> 4;
  const int so = (kg ^ ((frow >> 2) & 3)) * 8;           // the lane's 8 k values of its row: one swizzled 16-byte segment
  auto compute = [&](int buf) {
    const __half* base = wsm + (size_t)buf * ROWS * HBK + frow * HBK + so;
    const f16x8 ah = *reinterpret_cast<const f16x8*>(base);
    const f16x8 al = *reinterpret_cast<const f16x8*>(base + BT * HBK);
    const f16x8 bh = *reinterpret_cast<const f16x8*>(base + 2 * BT * HBK);
    const f16x8 bl = *reinterpret_cast<const f16x8*>(base + 3 * BT * HBK);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
  };
  const int nkt = g.K / HBK;
  const int mine = wave < nkt ? (nkt - wave + 3) / 4 : 0;         // K-tiles wave, wave + 4, ...
  constexpr int KEEP = PIECES * (ST - 2);                         // pieces of the younger tiles that may be pending
  constexpr int WAIT_KEEP = (KEEP & 15) | ((KEEP >> 4) << 14) | 0x0f70, WAIT_NONE = 0x0f70;
#pragma unroll
  for (int t = 0; t < ST - 1; ++t)
    if (t < mine) stage(t, (wave + 4 * t) * HBK);
  for (int i = 0; i < mine; ++i) {
    if (i + ST - 2 < mine) __builtin_amdgcn_s_waitcnt(WAIT_KEEP); else __builtin_amdgcn_s_waitcnt(WAIT_NONE);
    __builtin_amdgcn_sched_barrier(0);
    if (i + ST - 1 < mine) stage((i + ST - 1) % ST, (wave + 4 * (i + ST - 1)) * HBK);   // buffer of tile i-1: its reads are done
    compute(i % ST);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wave * 4 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    acc[r] = ((red[(0 * 4 + r) * 64 + lane] + red[(1 * 4 + r) * 64 + lane]) + red[(2 * 4 + r) * 64 + lane]) + red[(3 * 4 + r) * 64 + lane];
  const int n = bn + (lane & 15), rsub = 4 * (lane >> 4);
  const bool nok = FULL || n < g.N;
  const int oi = nok ? n / g.split_n : 0, on = n - oi * g.split_n;
  float* outp = g.out[oi];
  const int ldo = g.ldo[oi];
  float ssr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = bm + rsub + r;
    const bool mok = m < Mlive, ok = nok && mok;
    float v = acc[r] * acc_scale;
    if (g.row_ssq && mok) v *= ssq_rsqrt(e_ssq[r], g.inv_d_fix, g.eps);
    if (g.relu) v = fmaxf(v, 0.f);
    if (g.resid && ok) v = e_rf[r] + v;
    if (g.resid_h && ok) v = x_from_planes(__ushort_as_half((unsigned short)e_rh[r]), __ushort_as_half((unsigned short)e_rl[r])) + v;
    if (ok) {
      if (g.out_h) {
        __half hi, lo;
        split_f16(v * g.plane_scale, hi, lo, g.sat);
        g.out_h[(size_t)m * g.ldoh + n] = hi;
        g.out_h[g.o_ps + (size_t)m * g.ldoh + n] = lo;
        v = (__half2float(hi) + __half2float(lo)) / g.plane_scale;
      } else {
        outp[out_off(g, oi, m, ldo, on)] = v;
      }
    }
    ssr[r] = ok ? v * v : 0.f;
  }
  if (g.ssq_out) {   // the 16 lanes of a row group hold this block's 16 columns of rows rsub .. rsub + 3
#pragma unroll
    for (int o = 8; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < 4; ++r) ssr[r] += __shfl_xor(ssr[r], o, 64);
    if ((lane & 15) == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (bm + rsub + r < Mlive) atomicAdd(g.ssq_out + bm + rsub + r, ssq_to_fix(ssr[r], g.sat));
    }
  }
}

// ---- wave-split tiles for a few dozen to ~1500 rows (round 5) ------------------------------------------------------------------
// The skinny kernel's scheme (four waves split K, private LDS rings, no block barrier in the K walk, fixed-order reduction
// through LDS) on larger output tiles: (32 TM) x (32 TN) per block. Why: such a launch is bound by what ONE CU can keep in
// flight between L2 and its LDS (ring bytes / memory latency ~ 60-85 GB/s per CU), and with 32 x 32 tiles every column tile
// re-streams its activation rows and every row tile the weights: a 280 x 3072 x 768 product moved 165 MB through LDS-DMA
// (24 us per launch, 121 launches = the tail pass of a single-query search), 640 rows went to 128 x 64 tiles with split-K
// over blocks plus a separate epilogue launch (17 + 6 us). 64 x 32 and 64 x 64 tiles move 1/2 .. 1/3 of the bytes with the
// same number of bytes in flight per CU: ST stages of 2 (BM + BN) rows x 64 B per wave = 144 KB (64 x 32, three stages) or
// 128 KB (64 x 64, two stages) per block. The waves of quadrant (i, j) run the fused epilogue of their 32 x 32 part (the
// skinny kernel's arithmetic, in the same order: bit-identical results for every tile shape).
// blockIdx.y = K range of a split-K launch (ksplit > 1: raw partial sums to out[0] + y * part_stride, the caller runs
// splitk_epilogue*_kernel behind it).
template <bool FULL, int TM, int TN, int ST>
__global__ __launch_bounds__(256, 1) void gemm_h2_wsplit_kernel(GemmH2Args g, int tiles_m, int tiles_n) {
  const float acc_scale = g.dyn_a ? 1.0f / (dyn_plane_scale(*g.dyn_a) * dyn_plane_scale(*g.dyn_b)) : g.acc_scale;
  constexpr int BM = 32 * TM, BN = 32 * TN, ROWS = 2 * (BM + BN), PIECES = ROWS / 16, NQ = TM * TN;
  static_assert(NQ <= 4, "one epilogue wave per 32 x 32 quadrant");
  static_assert((size_t)4 * NQ * 16 * 64 * sizeof(float) <= (size_t)4 * ST * ROWS * HBK * sizeof(__half), "the partial tiles fit the rings");
  __shared__ __attribute__((aligned(16))) __half smem[4 * ST * ROWS * HBK];
  const int tile = blockIdx.x, tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int bm = tm * BM, bn = tn * BN;
  int Mlive = g.M;
  if (g.m_dev) {
    const int md = *g.m_dev;
    if (g.live_hi > 0 && (md <= g.live_lo || md > g.live_hi)) return;
    Mlive = min(md, g.M);
  }
  if (bm >= Mlive) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __half* wsm = smem + (size_t)wave * ST * ROWS * HBK;
  const int qi = wave / TN, qj = wave - qi * TN;            // the quadrant this wave finishes (waves >= NQ: none)
  const bool fused = g.ksplit <= 1;                          // split-K launches store raw partial sums
  // epilogue operands of the quadrant waves, requested before the K walk (see gemm_h2_skinny_kernel)
  unsigned long long e_ssq[16];
  unsigned int e_rh[16], e_rl[16];   // raw f16 bits, one 32-bit register each: as __half pairs the compiler packed them on arrival,
                                          // i.e. waited for every pair of loads before requesting the next (16 serial latencies in front of the K walk)
  float e_rf[16];
  if (wave < NQ && fused) {
    const int n = bn + 32 * qj + (lane & 31), rsub = bm + 32 * qi + 4 * (lane >> 5);
    const bool nok = FULL || n < g.N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = rsub + (r & 3) + 8 * (r >> 2);
      const bool mok = m < Mlive;
      e_ssq[r] = (g.row_ssq && mok) ? g.row_ssq[m] : 0ull;
      e_rf[r] = (g.resid && mok && nok) ? g.resid[(size_t)m * g.ldr + n] : 0.f;
      e_rh[r] = (g.resid_h && mok && nok) ? (unsigned int)reinterpret_cast<const unsigned short*>(g.resid_h)[(size_t)m * g.ldrh + n] : 0u;
      e_rl[r] = (g.resid_h && mok && nok) ? (unsigned int)reinterpret_cast<const unsigned short*>(g.resid_h)[g.r_ps + (size_t)m * g.ldrh + n] : 0u;
    }
  }
  // K range of this block, then K-tiles wave, wave + 4, ... of it
  int nkt = g.K / HBK, kbeg = 0;
  if (g.ksplit > 1) {
    const int per = (nkt + g.ksplit - 1) / g.ksplit;
    kbeg = blockIdx.y * per;
    nkt = min(per, nkt - kbeg);
  }
  // LDS rows of a stage: [A hi: BM][A lo: BM][W hi: BN][W lo: BN]; piece j = rows 16 j .. 16 j + 15
  const __half* src[PIECES];
#pragma unroll
  for (int j = 0; j < PIECES; ++j) {
    const int lrow = 16 * j + (lane >> 2);
    const int seg = (lane & 3) ^ ((lrow >> 2) & 3);
    const __half* base;
    int trow, limit;
    size_t ld;
    if (lrow < BM) { base = g.A; trow = bm + lrow; limit = g.M; ld = g.lda; }
    else if (lrow < 2 * BM) { base = g.A + g.a_ps; trow = bm + lrow - BM; limit = g.M; ld = g.lda; }
    else if (lrow < 2 * BM + BN) { base = g.W; trow = bn + lrow - 2 * BM; limit = g.N; ld = g.ldw; }
    else { base = g.W + g.w_ps; trow = bn + lrow - 2 * BM - BN; limit = g.N; ld = g.ldw; }
    if (!FULL && trow >= limit) trow = limit - 1;
    src[j] = base + (size_t)trow * ld + seg * 8 + (size_t)kbeg * HBK;
  }
  auto stage = [&](int buf, int k0) {
#pragma unroll
    for (int j = 0; j < PIECES; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + k0),
                                       (__attribute__((address_space(3))) void*)(wsm + (size_t)buf * ROWS * HBK + 16 * j * HBK),
                                       16, 0, 0);
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frow = lane & 31, sw = (lane >> 2) & 3, hf = lane >> 5;
  auto compute = [&](int buf) {
    const __half* base = wsm + (size_t)buf * ROWS * HBK;
#pragma unroll
    for (int c = 0; c < HBK / 16; ++c) {
      const int so = ((2 * c + hf) ^ sw) * 8;
      f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const f16x8*>(base + (32 * i + frow) * HBK + so);
        al[i] = *reinterpret_cast<const f16x8*>(base + (BM + 32 * i + frow) * HBK + so);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const f16x8*>(base + (2 * BM + 32 * j + frow) * HBK + so);
        bl[j] = *reinterpret_cast<const f16x8*>(base + (2 * BM + BN + 32 * j + frow) * HBK + so);
      }
      // per accumulator the skinny kernel's order of the three terms (lo x hi, hi x lo, hi x hi)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    }
  };
  const int mine = wave < nkt ? (nkt - wave + 3) / 4 : 0;
  constexpr int KEEP = PIECES * (ST - 2);                         // pieces of the younger tiles that may be pending
  static_assert(KEEP <= 63, "vmcnt is a 6-bit counter");
  constexpr int WAIT_KEEP = (KEEP & 15) | ((KEEP >> 4) << 14) | 0x0f70, WAIT_NONE = 0x0f70;
#pragma unroll
  for (int t = 0; t < ST - 1; ++t)
    if (t < mine) stage(t, (wave + 4 * t) * HBK);
  for (int i = 0; i < mine; ++i) {
    if (i + ST - 2 < mine) __builtin_amdgcn_s_waitcnt(WAIT_KEEP); else __builtin_amdgcn_s_waitcnt(WAIT_NONE);
    __builtin_amdgcn_sched_barrier(0);
    if (i + ST - 1 < mine) stage((i + ST - 1) % ST, (wave + 4 * (i + ST - 1)) * HBK);   // buffer of tile i-1: its reads are done
    compute(i % ST);
    __builtin_amdgcn_sched_barrier(0);
  }
  // the four waves' partial tiles through LDS (the rings are idle now): quadrant-major, added in the fixed order 0..3
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(((i * TN + j) * 4 + wave) * 16 + r) * 64 + lane] = acc[i][j][r];
  __syncthreads();
  if (wave >= NQ) return;
  f32x16 sum;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float* q = red + ((size_t)(wave * 4) * 16 + r) * 64 + lane;
    sum[r] = ((q[0] + q[16 * 64]) + q[2 * 16 * 64]) + q[3 * 16 * 64];
  }
  const int n = bn + 32 * qj + (lane & 31), rsub = bm + 32 * qi + 4 * (lane >> 5);
  const bool nok = FULL || n < g.N;
  if (!fused) {
    float* outp = g.out[0] + (size_t)blockIdx.y * g.part_stride;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = rsub + (r & 3) + 8 * (r >> 2);
      if (nok && m < Mlive) outp[(size_t)m * g.ldo[0] + n] = sum[r];
    }
    return;
  }
  const int oi = nok ? n / g.split_n : 0, on = n - oi * g.split_n;
  float* outp = g.out[oi];
  const int ldo = g.ldo[oi];
  float ssr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = rsub + (r & 3) + 8 * (r >> 2);
    const bool mok = m < Mlive, ok = nok && mok;
    float v = sum[r] * acc_scale;
    if (g.row_ssq && mok) v *= ssq_rsqrt(e_ssq[r], g.inv_d_fix, g.eps);
    if (g.relu) v = fmaxf(v, 0.f);
    if (g.resid && ok) v = e_rf[r] + v;
    if (g.resid_h && ok) v = x_from_planes(__ushort_as_half((unsigned short)e_rh[r]), __ushort_as_half((unsigned short)e_rl[r])) + v;
    if (ok) {
      if (g.out_h) {
        __half hi, lo;
        split_f16(v * g.plane_scale, hi, lo, g.sat);
        g.out_h[(size_t)m * g.ldoh + n] = hi;
        g.out_h[g.o_ps + (size_t)m * g.ldoh + n] = lo;
        v = (__half2float(hi) + __half2float(lo)) / g.plane_scale;
      } else {
        outp[out_off(g, oi, m, ldo, on)] = v;
      }
    }
    ssr[r] = ok ? v * v : 0.f;
  }
  if (g.ssq_out) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < 16; ++r) ssr[r] += __shfl_xor(ssr[r], o, 64);
    if ((lane & 31) == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = rsub + (r & 3) + 8 * (r >> 2);
        if (m < Mlive) atomicAdd(g.ssq_out + m, ssq_to_fix(ssr[r], g.sat));
      }
    }
  }
}

template <int TM, int TN, int ST>
static hipError_t launch_wsplit_cfg(const GemmH2Args& k, hipStream_t s) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  const int tiles_m = (k.M + BM - 1) / BM, tiles_n = (k.N + BN - 1) / BN;
  const bool full = (k.M % BM == 0) && (k.N % BN == 0) && !k.m_dev;
  const dim3 grid(tiles_m * tiles_n, k.ksplit > 1 ? k.ksplit : 1);
  if (full) hipLaunchKernelGGL((gemm_h2_wsplit_kernel<true, TM, TN, ST>), grid, dim3(256), 0, s, k, tiles_m, tiles_n);
  else hipLaunchKernelGGL((gemm_h2_wsplit_kernel<false, TM, TN, ST>), grid, dim3(256), 0, s, k, tiles_m, tiles_n);
  return hipGetLastError();
}

// Tile shape and K split of a wave-split launch. These launches are latency-bound by LDS capacity: a block keeps at most its
// rings in flight (64-96 KB) against a loaded L2 / Infinity-Cache latency of ~3 us, i.e. 40-50 GB/s per CU whatever the tile
// (tools/attic/fill_probe.hip: the LDS-DMA path itself sustains > 100 GB/s per CU from L2), one block per CU (128-144 KB of LDS).
// Model fitted to tools/wsplit_bench.sh on MI355X (profiles/archive/r05d_wsplit_gemm_bench.txt): launch = 5 us + rounds of blocks over
// the CUs x (3 us + KB per block / rate), + one reduction launch for a K split over blocks.
// cfg 0: 32 x 32 (four stages), 1: 64 x 32 (three), 2: 64 x 64 (two).
struct WsplitChoice { int cfg, ks; double us; long rounds; };
static WsplitChoice choose_wsplit(int M, int N, int K, int cus, bool can_split, size_t part_cap) {
  static const double lat_us = [] { const char* e = dev_getenv("RPR_WSPLIT_LAT_US"); return e ? atof(e) : 3.0; }();
  static const double split_us = [] { const char* e = dev_getenv("RPR_WSPLIT_SPLIT_US"); return e ? atof(e) : 4.5; }();
  const int bm[3] = {32, 64, 64}, bn[3] = {32, 32, 64};
  const double rate_gbs[3] = {48.0, 48.0, 41.0};
  WsplitChoice best{0, 1, 1e30, 1};
  for (int c = 0; c < 3; ++c) {
    const long tiles = (long)((M + bm[c] - 1) / bm[c]) * ((N + bn[c] - 1) / bn[c]);
    for (int ks = 1; ks <= 4; ++ks) {
      if (ks > 1 && (!can_split || K / ks < 256 || (size_t)M * N * ks > part_cap)) break;
      const int nkt = K / HBK;
      if (ks > 1 && (ks - 1) * ((nkt + ks - 1) / ks) >= nkt) continue;
      const long blocks = tiles * ks, rounds = (blocks + cus - 1) / cus;
      const double kb = (double)(bm[c] + bn[c]) * ((double)K / ks) * 4.0 * 1e-3;
      const double us = 5.0 + rounds * (lat_us + kb / rate_gbs[c]) + (ks > 1 ? split_us : 0.0);
      if (us < best.us) best = {c, ks, us, rounds};
    }
  }
  return best;
}

template <int BM, int BN, int WM = 2, int WN = 2, bool BF16 = false>
static hipError_t launch_cfg(const GemmH2Args& a, hipStream_t s) {
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  const bool full = (a.M % BM == 0) && (a.N % BN == 0) && !a.m_dev;
  const int ks = a.ksplit > 1 ? a.ksplit : 1;
  const dim3 grid(tiles_m * tiles_n, ks), blk(64 * WM * WN);
  static const int deep_max = [] { const char* e = dev_getenv("RPR_GEMM_DEEP"); return e ? atoi(e) : 128; }();
  if ((tiles_m * tiles_n * ks <= deep_max || BM < 128) && (!a.m_dev || a.live_hi > 0)) {   // fewer tiles than CUs: one block per CU, 3 K-tiles in flight
    if (full)
      hipLaunchKernelGGL((gemm_h2_dma_kernel<BM, BN, WM, WN, true, 4, BF16>), grid, blk, 0, s, a, tiles_m, tiles_n);
    else
      hipLaunchKernelGGL((gemm_h2_dma_kernel<BM, BN, WM, WN, false, 4, BF16>), grid, blk, 0, s, a, tiles_m, tiles_n);
    return hipGetLastError();
  }
  if (full)
    hipLaunchKernelGGL((gemm_h2_dma_kernel<BM, BN, WM, WN, true, 2, BF16>), grid, blk, 0, s, a, tiles_m, tiles_n);
  else
    hipLaunchKernelGGL((gemm_h2_dma_kernel<BM, BN, WM, WN, false, 2, BF16>), grid, blk, 0, s, a, tiles_m, tiles_n);
  return hipGetLastError();
}

// 256x256 tile, 8 waves (2x4) of 128x64, ping-pong schedule: half the staged bytes per MFMA of the 128x128 tile;
// >= ~112 tiles to beat the 128x128 kernel (measured), i.e. M = Q*B >= ~10k rows for N = 768
static hipError_t launch_256(const GemmH2Args& a_in, hipStream_t s) {
  GemmH2Args a = a_in;
  const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + 255) / 256;
  const bool full = (a.M % 256 == 0) && (a.N % 256 == 0) && !a.m_dev;
  // persistent blocks: one per CU of the stream (a whole number per XCD), fewer when the launch has fewer tiles
  const int cus = a.cus > 0 ? a.cus : 256, nt = tiles_m * tiles_n;
  const int grid = nt > cus ? cus : nt;
  const int ks = a.ksplit > 1 ? a.ksplit : 1;
  const dim3 gr(ks > 1 ? nt : grid, ks), bl(512);      // split-K launches are not persistent: one block per (tile, K range)
  // super-tile order for products more than four column tiles wide (N = 2304, 3072): column groups of 3 or 4 tiles, bands of
  // as many row panels as the blocks of one XCD fill with such a group. RPR_PP_SUPERTILE=0 (development builds): row-major
  static const int sup = [] { const char* e = dev_getenv("RPR_PP_SUPERTILE"); return e ? atoi(e) : 1; }();
  a.tile_cw = a.tile_rb = 0;
  if (sup && ks == 1 && tiles_n > 4 && nt > grid) {
    a.tile_cw = sup > 1 ? sup : (tiles_n % 4 == 0 ? 4 : (tiles_n % 3 == 0 ? 3 : 4));
    a.tile_rb = std::max(1, (grid / 8) / a.tile_cw);
  }
  if (a.bf16) {
    if (full) hipLaunchKernelGGL((gemm_h2_pp_kernel<true, false, true>), gr, bl, 0, s, a, tiles_m, tiles_n);
    else hipLaunchKernelGGL((gemm_h2_pp_kernel<false, false, true>), gr, bl, 0, s, a, tiles_m, tiles_n);
    return hipGetLastError();
  }
  if (full && a.trace)
    hipLaunchKernelGGL((gemm_h2_pp_kernel<true, true>), gr, bl, 0, s, a, tiles_m, tiles_n);
  else if (full)
    hipLaunchKernelGGL((gemm_h2_pp_kernel<true>), gr, bl, 0, s, a, tiles_m, tiles_n);
  else
    hipLaunchKernelGGL((gemm_h2_pp_kernel<false>), gr, bl, 0, s, a, tiles_m, tiles_n);
  return hipGetLastError();
}

hipError_t launch_gemm_h2_group(const GemmGroupArgs& p, void* scratch, hipStream_t s) {
  if (p.n <= 0) return hipSuccess;
  if (!scratch || p.n > GemmGroupArgs::MAXP || p.K <= 0 || (p.K & 63) || (p.lda & 7) || (p.ldw & 7)) return hipErrorInvalidValue;
  bool full = true;
  int total = 0;
  struct Super { int p, tm0, tn0, rm, rn; };
  std::vector<Super> sup;
  for (int i = 0; i < p.n; ++i) {
    if (!p.A[i] || !p.W[i] || !p.out[i] || p.M[i] <= 0 || p.N[i] <= 0 || (p.ldo[i] & 3) || (p.N[i] & 3)) return hipErrorInvalidValue;
    const int tm = (p.M[i] + 255) / 256, tn = (p.N[i] + 255) / 256;
    total += tm * tn;
    full = full && (p.M[i] % 256 == 0) && (p.N[i] % 256 == 0);
    // nearly equal parts of at most 4 tile rows / columns
    const int pm = (tm + 3) / 4, pn = (tn + 3) / 4, sm = (tm + pm - 1) / pm, sn = (tn + pn - 1) / pn;
    for (int a = 0; a < tm; a += sm)
      for (int b = 0; b < tn; b += sn) sup.push_back({i, a, b, std::min(sm, tm - a), std::min(sn, tn - b)});
  }
  if (total > GemmGroupArgs::MAX_TILES || total >= 65536) return hipErrorInvalidValue;
  // the tiles in super-tile order, cut into 8 equal runs: every XCD gets the same number of tiles (the main stream's kernels
  // run beside this launch and are spread evenly over the XCDs: whole super-tiles per XCD, 21 blocks on one XCD and 12 on
  // another, slowed those by 10 %), a run is one or two super-tiles plus parts of its neighbours
  std::vector<int> order;
  for (const Super& u : sup) {
    const int tn = (p.N[u.p] + 255) / 256;
    for (int a = 0; a < u.rm; ++a)
      for (int b = 0; b < u.rn; ++b) order.push_back((u.p << 16) | ((u.tm0 + a) * tn + u.tn0 + b));
  }
  std::vector<int> per_xcd[8];
  for (int x = 0; x < 8; ++x)
    for (size_t k = (size_t)x * order.size() / 8; k < (size_t)(x + 1) * order.size() / 8; ++k) per_xcd[x].push_back(order[k]);
  size_t slots = 0;
  for (int x = 0; x < 8; ++x) slots = std::max(slots, per_xcd[x].size());
  GroupAssign asg;
  asg.n = (int)slots * 8;
  if (asg.n > GemmGroupArgs::MAX_BLOCKS) return hipErrorInvalidValue;
  for (int i = 0; i < asg.n; ++i) asg.v[i] = -1;
  for (int x = 0; x < 8; ++x)
    for (size_t k = 0; k < per_xcd[x].size(); ++k) asg.v[x + 8 * k] = per_xcd[x][k];
  GemmH2Args* table = reinterpret_cast<GemmH2Args*>(scratch);
  int* assign = reinterpret_cast<int*>(static_cast<char*>(scratch) + GemmGroupArgs::TABLE_BYTES);
  hipLaunchKernelGGL(gemm_group_table_kernel, dim3(1), dim3(256), 0, s, p, asg, table, assign);
  const dim3 gr(asg.n), bl(512);
  if (full) hipLaunchKernelGGL((gemm_h2_pp_group_kernel<true>), gr, bl, 0, s, table, assign);
  else hipLaunchKernelGGL((gemm_h2_pp_group_kernel<false>), gr, bl, 0, s, table, assign);
  return hipGetLastError();
}

// out[m][n] = (resid ? resid[m][n] : 0) + sum over the splits, in split order (bitwise reproducible); N % 4 == 0
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int ks, size_t stride, int M, int N,
                                                             float* __restrict__ out, int ldo, const float* __restrict__ resid, int ldr) {
  const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int n4 = N >> 2;
  if (i4 >= (size_t)M * n4) return;
  const int m = (int)(i4 / n4), n = (int)(i4 - (size_t)m * n4) * 4;
  float4 a = resid ? *reinterpret_cast<const float4*>(resid + (size_t)m * ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < ks; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * stride + (size_t)m * N + n);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  *reinterpret_cast<float4*>(out + (size_t)m * ldo + n) = a;
}

// Fused epilogue of a split-K launch whose consumer needs more than a sum (search path, a few hundred to a few thousand
// rows in flight): out = epilogue(sum over the splits, in split order). One thread per output element, a wave covers 64
// consecutive columns of one row (N % 64 == 0). Same per-element arithmetic as the tile kernels' epilogues: accumulator
// scale, fused-RMSNorm row scale, ReLU, residual (fp32 or planes), f16-plane or fp32 output (K/V-cache layout included),
// fixed-point row sums of squares.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(GemmH2Args g, const float* __restrict__ part, int ks, size_t stride) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)g.M * g.N) return;                  // N % 64 == 0: a wave is inside or outside as a whole
  const int m = (int)(idx / g.N), n = (int)(idx - (size_t)m * g.N);
  float acc = 0.f;
  for (int k = 0; k < ks; ++k) acc += part[(size_t)k * stride + idx];
  const float acc_scale = g.dyn_a ? 1.0f / (dyn_plane_scale(*g.dyn_a) * dyn_plane_scale(*g.dyn_b)) : g.acc_scale;
  float v = acc * acc_scale;
  if (g.row_ssq) v *= ssq_rsqrt(g.row_ssq[m], g.inv_d_fix, g.eps);
  if (g.relu) v = fmaxf(v, 0.f);
  if (g.resid) v = g.resid[(size_t)m * g.ldr + n] + v;
  if (g.resid_h) v = x_from_planes(g.resid_h[(size_t)m * g.ldrh + n], g.resid_h[g.r_ps + (size_t)m * g.ldrh + n]) + v;
  if (g.out_h) {
    __half hi, lo;
    split_f16(v * g.plane_scale, hi, lo, g.sat);
    g.out_h[(size_t)m * g.ldoh + n] = hi;
    g.out_h[g.o_ps + (size_t)m * g.ldoh + n] = lo;
    v = (__half2float(hi) + __half2float(lo)) / g.plane_scale;   // the value the planes carry (row sums below)
  } else {
    const int oi = n / g.split_n, on = n - oi * g.split_n;
    g.out[oi][out_off(g, oi, m, g.ldo[oi], on)] = v;
  }
  if (g.ssq_out) {   // grid-uniform branch: all 64 lanes of the wave hold columns of row m
    float ss = v * v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(g.ssq_out + m, ssq_to_fix(ss, g.sat));
  }
}

// The same epilogue with FOUR consecutive columns per thread (N % 256 == 0: a wave is 256 consecutive columns of one row):
// 16-byte loads of the partials, 8-byte plane stores / 16-byte fp32 stores instead of one element per thread (2-byte plane
// stores). Per element the same arithmetic in the same order; the row sums of squares add four columns before the wave's
// butterfly.
__global__ __launch_bounds__(256) void splitk_epilogue4_kernel(GemmH2Args g, const float* __restrict__ part, int ks, size_t stride) {
  const size_t idx4 = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int n4 = g.N >> 2;
  if (idx4 >= (size_t)g.M * n4) return;                  // N % 256 == 0: a wave is inside or outside as a whole
  const int m = (int)(idx4 / n4), n = (int)(idx4 - (size_t)m * n4) * 4;
  const size_t idx = (size_t)m * g.N + n;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < ks; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * stride + idx);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  const float acc_scale = g.dyn_a ? 1.0f / (dyn_plane_scale(*g.dyn_a) * dyn_plane_scale(*g.dyn_b)) : g.acc_scale;
  const float rsc = g.row_ssq ? ssq_rsqrt(g.row_ssq[m], g.inv_d_fix, g.eps) : 1.0f;
  float v[4] = {a.x * acc_scale, a.y * acc_scale, a.z * acc_scale, a.w * acc_scale};
  float r[4] = {0.f, 0.f, 0.f, 0.f};
  if (g.resid) { const float4 q = *reinterpret_cast<const float4*>(g.resid + (size_t)m * g.ldr + n); r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w; }
  if (g.resid_h) {
    const uint2 hh = *reinterpret_cast<const uint2*>(g.resid_h + (size_t)m * g.ldrh + n);
    const uint2 ll = *reinterpret_cast<const uint2*>(g.resid_h + g.r_ps + (size_t)m * g.ldrh + n);
    const __half* h = reinterpret_cast<const __half*>(&hh); const __half* l = reinterpret_cast<const __half*>(&ll);
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = x_from_planes(h[e], l[e]);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (g.row_ssq) v[e] *= rsc;
    if (g.relu) v[e] = fmaxf(v[e], 0.f);
    if (g.resid || g.resid_h) v[e] = r[e] + v[e];
  }
  if (g.out_h) {
    __half hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      split_f16(v[e] * g.plane_scale, hi[e], lo[e], g.sat);
      v[e] = (__half2float(hi[e]) + __half2float(lo[e])) / g.plane_scale;   // the value the planes carry (row sums below)
    }
    *reinterpret_cast<uint2*>(g.out_h + (size_t)m * g.ldoh + n) = *reinterpret_cast<uint2*>(hi);
    *reinterpret_cast<uint2*>(g.out_h + g.o_ps + (size_t)m * g.ldoh + n) = *reinterpret_cast<uint2*>(lo);
  } else {
    const int oi = n / g.split_n, on = n - oi * g.split_n;
    *reinterpret_cast<float4*>(g.out[oi] + out_off(g, oi, m, g.ldo[oi], on)) = make_float4(v[0], v[1], v[2], v[3]);
  }
  if (g.ssq_out) {   // grid-uniform branch: all 64 lanes of the wave hold columns of row m
    float ss = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(g.ssq_out + m, ssq_to_fix(ss, g.sat));
  }
}

// Split-K through the 256x256 ping-pong kernel (weight gradients of the training step: dW[N, K] = dY^T X reduces over the
// 8192 rows of the batch into 9 .. 36 output tiles): blockIdx.y = K range, partial tiles to the caller's scratch,
// splitk_reduce_kernel adds them in split order (bitwise reproducible). Returns hipErrorNotSupported when the shape does
// not qualify (the caller falls back to the 128x64 split-K route).
static hipError_t launch_256_splitk(const GemmH2Args& a, hipStream_t s) {
  static const int on = [] { const char* e = dev_getenv("RPR_GEMM_PP_SPLITK"); return e ? atoi(e) : 1; }();
  const int kstep = a.bf16 ? 2 * HBK : HBK;
  if (!on || !a.part || (a.M & 255) || (a.N & 255) || (a.K % kstep) || a.relu || a.out_h || a.row_ssq || a.ssq_out || a.resid_h ||
      a.m_dev || a.rm_B || a.split_n < a.N || (a.ldo[0] & 3) || (a.resid && (a.ldr & 3)))
    return hipErrorNotSupported;
  const long tiles = (long)(a.M / 256) * (a.N / 256);
  const int cus = a.cus > 0 ? a.cus : 256, nkt = a.K / kstep;
  long ks = cus / tiles;                                        // one round of (tile, K range) blocks on the chip
  ks = std::min<long>(ks, nkt / 4);                             // at least 4 K-tiles per block
  ks = std::min<long>(ks, (long)(a.part_cap / ((size_t)a.M * a.N)));
  while (ks > 1 && (ks - 1) * ((nkt + ks - 1) / ks) >= nkt) --ks;
  if (tiles > 64 || ks < 2) return hipErrorNotSupported;
  GemmH2Args p = a;
  p.ksplit = (int)ks; p.part_stride = (size_t)a.M * a.N;
  p.out[0] = p.out[1] = p.out[2] = a.part; p.ldo[0] = p.ldo[1] = p.ldo[2] = a.N; p.split_n = a.N; p.resid = nullptr;
  hipError_t e = launch_256(p, s);
  if (e != hipSuccess) return e;
  const size_t n4 = (size_t)a.M * (a.N >> 2);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a.part, (int)ks, p.part_stride, a.M, a.N,
                     a.out[0], a.ldo[0], a.resid, a.ldr);
  return hipGetLastError();
}

hipError_t launch_gemm_h2(GemmH2Args& a_in, hipStream_t s) {
  GemmH2Args a = a_in;
  if (a.acc_scale == 0.f) a.acc_scale = 1.f;      // zero-initialised args mean "no scaling"
  if (a.plane_scale == 0.f) a.plane_scale = 1.f;
  a_in.kernel_cls = RPR_K_GEMM_SMALL;
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  if (a.K % HBK != 0 || a.K <= 0 || (a.lda & 7) || (a.ldw & 7)) return hipErrorInvalidValue;
  static const int skinny = [] { const char* e = dev_getenv("RPR_GEMM_SKINNY"); return e ? atoi(e) : 352; }();   // max rows (measured per search: 320 rows skinny 66.0 vs split-K route 68.5 ms, 400 rows 95.5 vs 71.8)
  static const int skinny16 = [] { const char* e = dev_getenv("RPR_GEMM_SKINNY16"); return e ? atoi(e) : 1; }();
  auto launch_skinny = [&](const GemmH2Args& k) {
    if (skinny16 && k.M <= 32 && (k.N & 15) == 0) {   // one query in flight: 16 x 16 tiles, all of K = 768 in flight
      const int tiles_n = k.N / 16;
      const dim3 grid(tiles_n, (k.M + 15) / 16);
      if (!k.m_dev) hipLaunchKernelGGL((gemm_h2_skinny16_kernel<true>), grid, dim3(256), 0, s, k, tiles_n);
      else hipLaunchKernelGGL((gemm_h2_skinny16_kernel<false>), grid, dim3(256), 0, s, k, tiles_n);
      return hipGetLastError();
    }
    const int tiles_m = (k.M + 31) / 32, tiles_n = (k.N + 31) / 32;
    const bool full = (k.M % 32 == 0) && (k.N % 32 == 0) && !k.m_dev;
    if (full) hipLaunchKernelGGL((gemm_h2_skinny_kernel<true>), dim3(tiles_m * tiles_n), dim3(256), 0, s, k, tiles_m, tiles_n);
    else hipLaunchKernelGGL((gemm_h2_skinny_kernel<false>), dim3(tiles_m * tiles_n), dim3(256), 0, s, k, tiles_m, tiles_n);
    return hipGetLastError();
  };
  if (a.m_dev && a.small_live > 0 && a.M > a.small_live && !a.bf16) {
    // a compacted stage: capacity M rows, usually a handful alive. The large-tile kernel would walk all of K with the
    // one or two blocks that hold live rows (60-250 us per launch). The launch is enqueued as a group of three, each
    // gated on the device-side live count (two of them exit at once): the large-tile kernel for more than small_live
    // rows, a 128x64 launch sized for small_live rows, and the skinny kernel for at most `skinny` rows (a few leftover
    // queries: 10 us instead of 17-20 for the 128x64 tile walking K alone).
    GemmH2Args big = a, mid = a, sk = a;
    const int sk_rows = std::min(skinny, a.small_live);
    big.small_live = 0; big.live_lo = a.small_live; big.live_hi = 0x7fffffff;
    mid.small_live = 0; mid.live_lo = sk_rows; mid.live_hi = a.small_live; mid.M = a.small_live;
    sk.small_live = 0; sk.live_lo = -1; sk.live_hi = sk_rows; sk.M = (sk_rows + 31) / 32 * 32;
    hipError_t e = launch_gemm_h2(big, s);
    if (e != hipSuccess) return e;
    a_in.kernel_cls = big.kernel_cls;
    if (sk_rows < a.small_live) { e = launch_cfg<128, 64>(mid, s); if (e != hipSuccess) return e; }
    return launch_skinny(sk);
  }
  if (a.bf16) {
    // one bf16 plane per operand (training GEMMs, RPR_PREC_BF16): fp32 output, optional residual / ReLU, split-K for the
    // long reductions into few tiles (weight gradients); K-tiles of 64 columns
    if (a.out_h || a.row_ssq || a.ssq_out || a.resid_h || a.m_dev || a.rm_B || (a.K & 63) || (a.N & 3) || (a.ldo[0] & 3) ||
        (a.resid && (a.ldr & 3)) || a.split_n < a.N)
      return hipErrorInvalidValue;                       // (the bf16 kernels' epilogues store 16-byte pieces of ONE fp32 output)
    if (a.out_b || a.out_bt) {
      // bf16 operands for the consumers straight from the epilogue (GemmH2Args::out_b): the 256 x 256 kernel's FULL instantiation only
      if ((a.M & 255) || (a.N & 255) || a.resid || a.ksplit > 1 || (a.out_b && (a.ldob & 7)) || (a.out_bt && ((a.ldobt & 7) || a.ldobt < a.M)) ||
          (a.mask_src && (a.ldmask & 3)))
        return hipErrorInvalidValue;
      a_in.kernel_cls = RPR_K_GEMM;
      return launch_256(a, s);
    }
    const long t128b = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    // (A kernel with 128x128 wave tiles — 256x256 block, four waves, one per SIMD, 512 registers: two thirds of the LDS reads
    // per MFMA — was built and measured: 31-34 us per 256x256x768 tile against 23 us for this shape on the 128-row kernel
    // and 60-65 us against 54-58 on the ping-pong kernel. With ONE wave per SIMD the 16 LDS-DMA pieces and 32 fragment reads
    // of a K-tile are issued by the wave that also issues the 64 MFMAs, in series: ~2.2 us per K-tile again. Source kept
    // as tools/gemm_bf16_w128.hip.txt; HISTORY.md.)
    if (a.prefer_pp) {
      static const int dwk = [] { const char* e = dev_getenv("RPR_TRAIN_DW_TILE"); return e ? atoi(e) : 256; }();   // experiment: 128 / 64
      if (dwk == 128) return launch_cfg<128, 128, 2, 2, true>(a, s);
      if (dwk == 64) return launch_cfg<128, 64, 2, 2, true>(a, s);
      a_in.kernel_cls = RPR_K_GEMM;
      return launch_256(a, s);
    }
    if (a.K >= 2048) {
      const hipError_t e = launch_256_splitk(a, s);
      if (e != hipErrorNotSupported) { if (e == hipSuccess) a_in.kernel_cls = RPR_K_GEMM; return e; }
    }
    if (a.part && a.K >= 2048 && t128b * 2 < 640 && !a.relu && a.split_n >= a.N && (a.N & 3) == 0 && (a.ldo[0] & 3) == 0 &&
        (!a.resid || (a.ldr & 3) == 0)) {
      const long t = (long)((a.M + 127) / 128) * ((a.N + 63) / 64);
      long ks = std::min<long>((640 + t - 1) / t, a.K / 1024);
      ks = std::min<long>(ks, (long)(a.part_cap / ((size_t)a.M * a.N)));
      const int nkt = a.K / (2 * HBK);
      while (ks > 1 && (ks - 1) * ((nkt + ks - 1) / ks) >= nkt) --ks;
      if (ks > 1) {
        GemmH2Args p = a;
        p.ksplit = (int)ks; p.part_stride = (size_t)a.M * a.N;
        p.out[0] = p.out[1] = p.out[2] = a.part; p.ldo[0] = p.ldo[1] = p.ldo[2] = a.N; p.split_n = a.N; p.resid = nullptr;
        hipError_t e = launch_cfg<128, 64, 2, 2, true>(p, s);
        if (e != hipSuccess) return e;
        const size_t n4 = (size_t)a.M * (a.N >> 2);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a.part, (int)ks, p.part_stride, a.M,
                           a.N, a.out[0], a.ldo[0], a.resid, a.ldr);
        return hipGetLastError();
      }
    }
    const long t256b = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    static const int bf_pp = [] { const char* e = dev_getenv("RPR_BF16_PP"); return e ? atoi(e) : 200; }();   // min tiles of 256^2 (0 = never)
    if (bf_pp > 0 && t256b >= bf_pp) return launch_256(a, s);     // ping-pong 256x256 tiles when they fill the chip
    return t128b < 256 ? launch_cfg<128, 64, 2, 2, true>(a, s) : launch_cfg<128, 128, 2, 2, true>(a, s);
  }
  static const int force = [] { const char* e = dev_getenv("RPR_GEMM_TILE"); return e ? atoi(e) : 0; }();
  const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
  // 256-tile rounds on the 256 CUs: a launch just over a whole number of rounds (e.g. 288 tiles) leaves most of the
  // chip idle in its last round; the 128-tile kernels quantise finer (measured M = 8192, N = 2304: 135 vs 151 us)
  const int cus = a.cus > 0 ? a.cus : 256;         // a lane stream owns part of the chip: thresholds scale with it
  const double round_eff = (double)t256 / (double)(((t256 + cus - 1) / cus) * cus);
  if (force == 256 || a.prefer_pp || (force == 0 && t256 >= 112L * cus / 256 && (round_eff >= 0.6 || a.out_h || a.row_ssq))) {
    a_in.kernel_cls = RPR_K_GEMM;
    // Row split of a launch just over a whole number of rounds (beam 1000 with one query: 318 tiles of 256^2 for the
    // N = 768 products = 1.24 rounds, the second one with 62 of 256 CUs busy): the row tiles that fill whole rounds go to
    // the ping-pong kernel, the rows behind them to the 128 x 128 tile kernel (a quarter of the work per block, 1.22 x the
    // time per flop), as a second launch on the same stream. Taken when the estimate — whole rounds + 0.43 per round of
    // 128^2 tiles (measured: 5240 rows x 768 columns = 246 such tiles in 35.7 us against 87.1 us for the 255 tiles of 256^2
    // in front of them, profiles/archive/r05x_rowsplit_gemm.txt) + ~6 us for the second launch — is under 0.9 of the rounds the
    // ping-pong kernel alone would need. RPR_GEMM_ROWSPLIT=0: off; =2 (tests): every launch of this route with two or more
    // row tiles is split in the middle.
    static const int row_split = [] { const char* e = dev_getenv("RPR_GEMM_ROWSPLIT"); return e ? atoi(e) : 1; }();
    const bool split_all = row_split == 2 && a.M > 256;
    if (row_split && (force == 0 || split_all) && !a.prefer_pp && !a.rm_B && a.ksplit <= 1 && !a.trace && a.small_live == 0 && (!a.no_row_split || split_all) &&
        (t256 > cus || split_all)) {
      const int tiles_n = (a.N + 255) / 256;
      const long rounds = t256 / cus;
      const int rows_main = split_all ? ((a.M + 255) / 256 / 2) * 256 : (int)((rounds * cus) / tiles_n) * 256, m_rest = a.M - rows_main;
      if ((t256 % cus != 0 || split_all) && rows_main > 0 && m_rest > 0) {
        const long t128r = (long)((m_rest + 127) / 128) * ((a.N + 127) / 128);
        const double tile_us = 78.0 * a.K / 768.0;
        const double cost_split = (double)rounds + 0.43 * (double)((t128r + cus - 1) / cus) + 6.0 / tile_us;
        if (split_all || cost_split < 0.9 * (double)(rounds + 1)) {
          static const int log_split = [] { const char* e = dev_getenv("RPR_GEMM_ROWSPLIT_LOG"); return e ? atoi(e) : 0; }();
          if (log_split) fprintf(stderr, "[rowsplit] M=%d N=%d K=%d: %d rows on 256x256 tiles, %d on 128x128\n", a.M, a.N, a.K, rows_main, m_rest);
          GemmH2Args main_p = a, rest = a;
          main_p.M = rows_main;
          rest.M = m_rest; rest.m_base = a.m_base + rows_main;
          rest.A = a.A + (size_t)rows_main * a.lda;
          for (int i = 0; i < 3; ++i) if (a.out[i]) rest.out[i] = a.out[i] + (size_t)rows_main * a.ldo[i];
          if (a.out_h) rest.out_h = a.out_h + (size_t)rows_main * a.ldoh;
          if (a.resid) rest.resid = a.resid + (size_t)rows_main * a.ldr;
          if (a.resid_h) rest.resid_h = a.resid_h + (size_t)rows_main * a.ldrh;
          if (a.row_ssq) rest.row_ssq = a.row_ssq + rows_main;
          if (a.ssq_out) rest.ssq_out = a.ssq_out + rows_main;
          hipError_t e = launch_256(main_p, s);
          if (e != hipSuccess) return e;
          return launch_cfg<128, 128>(rest, s);
        }
      }
    }
    return launch_256(a, s);
  }
  // a handful of rows (one to a few queries in flight): the launch is a weight stream; a 128-row tile would spend
  // most of the per-CU LDS-DMA rate (~25 GB/s) on padding rows, and 32-wide column tiles give 4x the blocks
  // 33 .. ~1500 rows (a handful to ~150 queries in flight, the tail pass of one query, beam 1000 at batch 1): wave-split tiles,
  // shape and K split from choose_wsplit (RPR_WSPLIT_CFG / RPR_WSPLIT_KS force them; RPR_GEMM_WSPLIT_MAX = 0: the routes below)
  static const int wsplit_max = [] { const char* e = dev_getenv("RPR_GEMM_WSPLIT_MAX"); return e ? atoi(e) : 1400; }();
  static const int wsplit_cfg = [] { const char* e = dev_getenv("RPR_WSPLIT_CFG"); return e ? atoi(e) : -1; }();
  static const int wsplit_ks = [] { const char* e = dev_getenv("RPR_WSPLIT_KS"); return e ? atoi(e) : 0; }();
  if (force == 0 && a.M > 32 && a.M <= wsplit_max) {
    const bool can_split = a.part && a.mid_split && !a.m_dev && (a.N & 63) == 0;
    WsplitChoice ch = choose_wsplit(a.M, a.N, a.K, a.cus > 0 ? a.cus : 256, can_split, a.part_cap);
    // beyond the skinny kernel's old range the 128 x 64 split-K route is as fast once the best wave-split shape needs a second
    // round of blocks (measured at 640 rows: N = 2304 / 3072 27.8 / 28.9 us against 28.9 / 30.0): those launches stay where they were
    const bool take = a.M <= skinny || ch.rounds <= 1 || wsplit_cfg >= 0;
    if (wsplit_cfg >= 0 && wsplit_cfg <= 2) ch.cfg = wsplit_cfg;
    if (wsplit_ks > 0 && (wsplit_ks == 1 || (can_split && (size_t)a.M * a.N * wsplit_ks <= a.part_cap && a.K / wsplit_ks >= 64))) ch.ks = wsplit_ks;
    auto go = [&](const GemmH2Args& k) {
      return ch.cfg == 0 ? launch_wsplit_cfg<1, 1, 4>(k, s) : ch.cfg == 1 ? launch_wsplit_cfg<2, 1, 3>(k, s) : launch_wsplit_cfg<2, 2, 2>(k, s);
    };
    if (take && ch.ks <= 1) return ch.cfg == 0 ? launch_skinny(a) : go(a);
    if (take) {
    GemmH2Args p = a;
    p.ksplit = ch.ks; p.part_stride = (size_t)a.M * a.N;
    p.out[0] = p.out[1] = p.out[2] = a.part; p.ldo[0] = p.ldo[1] = p.ldo[2] = a.N; p.split_n = a.N;
    p.out_h = nullptr; p.resid = nullptr; p.resid_h = nullptr; p.relu = 0; p.row_ssq = nullptr; p.ssq_out = nullptr;
    p.rm_B = 0; p.acc_scale = 1.0f; p.dyn_a = p.dyn_b = nullptr;
    hipError_t e = go(p);
    if (e != hipSuccess) return e;
    const size_t n = (size_t)a.M * a.N;
    const bool vec4 = (a.N & 255) == 0 && (a.split_n & 3) == 0 && (a.ldo[0] & 3) == 0 && (a.ldo[1] & 3) == 0 && (a.ldo[2] & 3) == 0 &&
                      (!a.resid || (a.ldr & 3) == 0) && (!a.resid_h || (a.ldrh & 3) == 0) && (!a.out_h || (a.ldoh & 3) == 0);
    if (vec4) hipLaunchKernelGGL(splitk_epilogue4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, a, a.part, ch.ks, p.part_stride);
    else hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, a.part, ch.ks, p.part_stride);
    return hipGetLastError();
    }
  }
  if (force == 0 && a.M <= skinny) return launch_skinny(a);   // (with m_dev: row tiles past the live rows exit)
  const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
  // A few hundred to a few thousand rows in flight (beam 1000 with one query, beam 100 with a dozen, beam 10 with
  // 40-400): the 128x64 launch has fewer blocks than CUs and each walks all of K alone (24-96 K-tiles at ~1 us).
  // Split K over blockIdx.y into the caller's scratch and run the fused epilogue as its own launch.
  static const int mid_split = [] { const char* e = dev_getenv("RPR_GEMM_MIDSPLIT"); return e ? atoi(e) : 1; }();
  if (mid_split && force == 0 && a.part && a.mid_split && !a.m_dev && (a.N & 63) == 0 && a.K >= 512) {
    const long t = (long)((a.M + 127) / 128) * ((a.N + 63) / 64);
    const int cus = a.cus > 0 ? a.cus : 256;
    static const int ks_cap = [] { const char* e = dev_getenv("RPR_GEMM_MIDSPLIT_CAP"); return e ? atoi(e) : 4; }();
    long ks = std::min<long>(std::min<long>((3L * cus / 2 + t - 1) / t, ks_cap), a.K / 128);
    ks = std::min<long>(ks, (long)(a.part_cap / ((size_t)a.M * a.N)));
    const int nkt = a.K / HBK;
    while (ks > 1 && (ks - 1) * ((nkt + ks - 1) / ks) >= nkt) --ks;
    if (ks > 1 && t < cus) {
      GemmH2Args p = a;
      p.ksplit = (int)ks; p.part_stride = (size_t)a.M * a.N;
      p.out[0] = p.out[1] = p.out[2] = a.part; p.ldo[0] = p.ldo[1] = p.ldo[2] = a.N; p.split_n = a.N;
      p.out_h = nullptr; p.resid = nullptr; p.resid_h = nullptr; p.relu = 0; p.row_ssq = nullptr; p.ssq_out = nullptr;
      p.rm_B = 0; p.acc_scale = 1.0f; p.dyn_a = p.dyn_b = nullptr;
      hipError_t e = launch_cfg<128, 64>(p, s);
      if (e != hipSuccess) return e;
      const size_t n = (size_t)a.M * a.N;
      const bool vec4 = (a.N & 255) == 0 && (a.split_n & 3) == 0 && (a.ldo[0] & 3) == 0 && (a.ldo[1] & 3) == 0 && (a.ldo[2] & 3) == 0 &&
                        (!a.resid || (a.ldr & 3) == 0) && (!a.resid_h || (a.ldrh & 3) == 0) && (!a.out_h || (a.ldoh & 3) == 0);
      if (vec4) hipLaunchKernelGGL(splitk_epilogue4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, a, a.part, (int)ks, p.part_stride);
      else hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, a.part, (int)ks, p.part_stride);
      return hipGetLastError();
    }
  }
  // split-K: the caller lent scratch for partial results and the launch is a long reduction into few tiles
  static const int split_target = [] { const char* e = dev_getenv("RPR_GEMM_SPLITK"); return e ? atoi(e) : 640; }();
  if (a.part && split_target > 0 && a.K >= 2048 && t128 * 2 < split_target && !a.mid_split) {
    const hipError_t e = launch_256_splitk(a, s);
    if (e != hipErrorNotSupported) { if (e == hipSuccess) a_in.kernel_cls = RPR_K_GEMM; return e; }
  }
  if (a.part && split_target > 0 && a.K >= 2048 && t128 * 2 < split_target && !a.out_h && !a.ssq_out && !a.row_ssq && !a.relu && !a.resid_h &&
      !a.m_dev && a.split_n >= a.N && (a.N & 3) == 0 && (a.ldo[0] & 3) == 0 && (!a.resid || (a.ldr & 3) == 0)) {
    const long t = (long)((a.M + 127) / 128) * ((a.N + 63) / 64);
    long ks = std::min<long>((split_target + t - 1) / t, a.K / 1024);
    ks = std::min<long>(ks, (long)(a.part_cap / ((size_t)a.M * a.N)));
    const int nkt = a.K / HBK;
    while (ks > 1 && (ks - 1) * ((nkt + ks - 1) / ks) >= nkt) --ks;   // every split owns at least one K-tile
    if (ks > 1) {
      GemmH2Args p = a;
      p.ksplit = (int)ks; p.part_stride = (size_t)a.M * a.N;
      p.out[0] = p.out[1] = p.out[2] = a.part; p.ldo[0] = p.ldo[1] = p.ldo[2] = a.N; p.split_n = a.N; p.resid = nullptr;
      hipError_t e = launch_cfg<128, 64>(p, s);
      if (e != hipSuccess) return e;
      const size_t n4 = (size_t)a.M * (a.N >> 2);
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a.part, (int)ks, p.part_stride, a.M,
                         a.N, a.out[0], a.ldo[0], a.resid, a.ldr);
      return hipGetLastError();
    }
  }
  const bool narrow = force ? (force == 64) : (t128 < 256);
  return narrow ? launch_cfg<128, 64>(a, s) : launch_cfg<128, 128>(a, s);
}

// fp32 [rows, cols] -> two f16 planes [2][rows][cols] (weights at load time, generic activations)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, __half* __restrict__ out,
                                                            size_t n, size_t plane_stride, float scale,
                                                            const float* __restrict__ colscale, int cols,
                                                            unsigned int* sat) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  float4 v = *reinterpret_cast<const float4*>(x + i);
  if (colscale) {   // cols % 4 == 0: the four elements stay inside one row
    const float4 c = *reinterpret_cast<const float4*>(colscale + (i % (size_t)cols));
    v.x *= c.x; v.y *= c.y; v.z *= c.z; v.w *= c.w;
  }
  __half h[4], l[4];
  split_f16(v.x * scale, h[0], l[0], sat); split_f16(v.y * scale, h[1], l[1], sat);
  split_f16(v.z * scale, h[2], l[2], sat); split_f16(v.w * scale, h[3], l[3], sat);
  *reinterpret_cast<uint2*>(out + i) = *reinterpret_cast<uint2*>(h);
  *reinterpret_cast<uint2*>(out + plane_stride + i) = *reinterpret_cast<uint2*>(l);
}

hipError_t launch_split_planes(const float* x, __half* out, size_t n, size_t plane_stride, hipStream_t s, float scale,
                               const float* colscale, int cols, unsigned int* sat) {
  if (n == 0) return hipSuccess;
  if ((n & 3) || (colscale && (cols <= 0 || (cols & 3)))) return hipErrorInvalidValue;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, x, out, n, plane_stride,
                     scale, colscale, cols, sat);
  return hipGetLastError();
}

}  // namespace rpr
