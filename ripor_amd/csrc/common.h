// Shared declarations for libripor_hip.so (gfx950 only; wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>
#include <string>

#include "../../include/ripor_hip.h"

namespace rpr {

constexpr int WAVE = 64;
constexpr int MAX_LQ = 256;          // encoder tokens per query supported by the attention kernels
constexpr int MAX_DEC_LEN = 64;      // decoder positions supported (reference uses 32 or 16)
constexpr int DKV = 64;              // head dim the fast attention kernels are written for (t5-base/large); d_kv = 128 (t5-3b) runs
                                     // on the generic kernels enc_attn_kernel<128> / dec_attn_kernel<., 128> without the forced tail

// Environment switches. The product library reads a handful (README: precision, forced tail, fork depths, lane split, trie
// threads, and the three selectors the test-suite compares bit for bit against the default: RPR_SELECT_RADIX,
// RPR_SELECT_LEVELS, RPR_TAIL_RANK_REPLAY) through getenv. Everything else — A/B switches of kernel routes and generations,
// tuning constants, debug traces — goes through dev_getenv, which is getenv only in a build with -DRPR_DEV_SWITCHES
// (ripor_amd/libripor_hip_dev.so: tools/ and the tests that compare kernel variants, loaded with RPR_DEV_LIB=1) and nullptr
// in the product library: there every such switch is its default and its name is not even in the binary
// (tests/test_abi.py checks the names the .so carries).
#ifdef RPR_DEV_SWITCHES
inline const char* dev_getenv(const char* name) { return getenv(name); }
#else
inline const char* dev_getenv(const char*) { return nullptr; }
#endif

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define RPR_HIP(call)                                                        \
  do {                                                                       \
    hipError_t _e = (call);                                                  \
    if (_e != hipSuccess) return ::rpr::hip_fail(_e, #call, __FILE__, __LINE__); \
  } while (0)

#define RPR_REQUIRE(cond, msg)                                               \
  do {                                                                       \
    if (!(cond)) {                                                           \
      ::rpr::set_error(std::string("invalid argument: ") + (msg));           \
      return RPR_ERR_INVALID;                                                \
    }                                                                        \
  } while (0)

// x = hi + lo carried as two f16 values (22 significant bits); see gemm_h2.hip.
// A value outside the f16 range is clamped AND reported: `sat` (nullable) is the ctx's sticky saturation word
// (rpr_get_status); the host then repeats the call on the exact fp32 path instead of returning clipped results.
// NaN/Inf inputs also count (the comparison is written so that NaN fails it).
__device__ __forceinline__ void split_f16(float x, __half& hi, __half& lo, unsigned int* sat = nullptr) {
  if (!(fabsf(x) <= 65504.f)) {
    if (sat) *sat = 1u;              // benign race: every writer stores 1
    x = fminf(fmaxf(x, -65504.f), 65504.f);
  }
  hi = __float2half_rn(x);
  lo = __float2half_rn(x - __half2float(hi));
}

// Per-tensor dynamic plane scale (training GEMMs: activations and gradients of unknown magnitude): the power of two s
// with amax * s in [2^13, 2^14) — below the f16 maximum with headroom, and far enough above the f16 subnormals that
// entries down to ~2^-17 of the largest one keep their lo plane normal. amax = 0 / non-finite: 1.
__host__ __device__ __forceinline__ float dyn_plane_scale(float amax) {
  union { float f; int i; } u; u.f = amax;
  const int e = (u.i >> 23) & 0xff;
  if (e == 0 || e == 255) return 1.0f;
  int be = 127 + 13 - (e - 127);
  be = be > 254 ? 254 : (be < 1 ? 1 : be);
  u.i = be << 23;
  return u.f;
}

// Row sum of squares in fixed point (2^-16 units, 64 bits): partial sums from different blocks are combined with
// integer atomics, so the total does not depend on the order of arrival (bitwise-reproducible RMSNorm scale).
// Range (round 6; the residual planes hold |x| < 1.05e6 since round 5, common.h X_PLANE_SCALE): a partial added by an
// epilogue is capped at 2^57 units (2.2e12: two elements at the plane limit) and a row has at most 64 of them (16-column
// tiles of d_model = 1024), so the 64-bit total cannot wrap; a row stored whole by an embedding kernel is capped at 2^63
// (1.4e14: RMS 4.3e5 at d = 768). A capped value raises the ctx's sticky saturation word like a clamped plane element does,
// and the boundary repeats the batch in exact fp32. (Until round 5: 2^-20 units read back as a signed value — the sum wrapped
// silently from sum x^2 = 8.8e12 on, RMS 1e5 at d = 768.) Partials are rounded to nearest: the error of a row total is
// ~6e-5 absolute, 1e-7 of a row of RMS 1.
constexpr float SSQ_FIX = 65536.0f;
constexpr float SSQ_PART_CAP = 2199023255552.0f;        // 2^41 = 2^57 units
constexpr float SSQ_ROW_CAP = 140737488355328.0f;       // 2^47 = 2^63 units
__device__ __forceinline__ unsigned long long ssq_to_fix(float ss, unsigned int* sat, float cap = SSQ_PART_CAP) {
  if (!(ss < cap)) {                 // also NaN
    if (sat) *sat = 1u;              // benign race: every writer stores 1
    ss = cap;
  }
  return (unsigned long long)(ss * SSQ_FIX + 0.5f);
}
__device__ __forceinline__ float ssq_rsqrt(unsigned long long fix, float inv_d_fix, float eps) {
  return rsqrtf(fmaf((float)fix, inv_d_fix, eps));   // inv_d_fix = 1 / (d * SSQ_FIX)
}

// ---- GEMM: C = act(A @ W^T) (+ residual), fp32 MFMA --------------------------------------------
struct GemmArgs {
  const float* A; int lda;      // [M, K]
  const float* W; int ldw;      // [N, K]  (torch Linear layout)
  const float* resid; int ldr;  // nullable, [M, N]
  float* out[3]; int ldo[3];    // column n is written to out[n / split_n][:, n % split_n]
  int split_n;                  // == N when there is a single output
  int M, N, K;
  int relu;
  // KV-cache element map for the outputs 1 and 2 (rm_B == 0: plain [M, ldo] rows): element (m, n) goes to
  // out[i] + (m / rm_B) * rm_stride + (m % rm_B) * rm_slot + (n / d_kv) * rm_head + n % d_kv   
  int rm_B; size_t rm_stride, rm_slot, rm_head;
  int rm_dshift;                           // log2(d_kv) of the map above (0 = 6)
  const int* m_dev;             // nullable: number of live rows on the device (packed encoder); tiles past it exit
};
hipError_t launch_gemm(const GemmArgs& a, hipStream_t s);

// ---- split-precision GEMM: operands as two f16 planes (hi, lo), 3 f16 MFMAs per product ---------
struct GemmH2Args {
  const __half* A; size_t a_ps; int lda;   // planes [2][M][lda], plane stride a_ps elements
  const __half* W; size_t w_ps; int ldw;   // planes [2][N][ldw]
  const float* resid; int ldr;
  float* out[3]; int ldo[3]; int split_n;  // fp32 outputs (used when out_h == nullptr)
  __half* out_h; size_t o_ps; int ldoh;    // f16-plane output [2][M][ldoh] (feeds the next GEMM)
  int M, N, K;
  int relu;
  unsigned long long* trace;               // diagnostic cycle stamps of block 0 (nullptr in production)
  int rm_B; size_t rm_stride, rm_slot, rm_head;  // KV-cache element map for out[1], out[2] (see GemmArgs)
  int rm_dshift;                           // log2(d_kv) of that map (0 = 6)
  const int* m_dev;                        // nullable: live row count on the device (see GemmArgs)
  int m_base;                              // rows in front of this launch's first row that *m_dev counts too (the second launch of a
                                           // row-split product, launch_gemm_h2; 128-row tile kernels only): live rows = *m_dev - m_base
  // Power-of-two scaling of the f16 planes (exact; see W_/A_/FF_PLANE_SCALE below): the accumulators are multiplied
  // by acc_scale = 1 / (scale of A's planes * scale of W's planes); planes written by the epilogue (out_h) are
  // scaled by plane_scale. 0 means 1.
  float acc_scale, plane_scale;
  // Fused RMSNorm. In split-precision mode the residual stream x lives in f16 planes only (hi + lo =
  // 22 bits, X_PLANE_SCALE) plus one fixed-point sum of squares per row and norm site.
  // Consumer side: A = the planes of the UN-normalised x, W = the planes of W * diag(ln_weight); row m of the result
  // is multiplied by rsqrt(row_ssq[m] / d + eps) in the epilogue.
  // Producer side (residual GEMMs): the residual is read from planes (resid_h, in place with out_h), the sum is
  // written back as planes (out_h) and the tile's part of every row's sum of squares is added to ssq_out.
  const unsigned long long* row_ssq; float inv_d_fix, eps;
  const __half* resid_h; size_t r_ps; int ldrh; unsigned long long* ssq_out;
  unsigned int* sat;                       // sticky saturation word of the ctx (split_f16)
  // dynamic plane scales (training GEMMs): device-side absolute maxima of A and B; when non-null the accumulators are
  // multiplied by 1 / (dyn_plane_scale(*dyn_a) * dyn_plane_scale(*dyn_b)) instead of acc_scale
  const float* dyn_a; const float* dyn_b;
  int cus;                                 // CUs the launch may use (0 = the whole chip): a lane stream's CU mask, for the tile choice
  // split-K (optional): scratch for partial results lent by the caller; launch_gemm_h2 decides whether to use it
  float* part; size_t part_cap;            // floats
  int mid_split;                           // search path: allow the split-K + separate fused-epilogue route (mid-size M)
  int ksplit; size_t part_stride;          // set by the launcher
  // Live-count window (launches with m_dev only; 0 / 0 = no window): the kernel runs iff live_lo < *m_dev <= live_hi.
  // launch_gemm_h2 uses it to enqueue a compacted stage's GEMM twice when small_live > 0 — the large-tile kernel for many
  // live rows and a small-tile one for at most small_live — one of which exits at once (the launch geometry of a
  // hipGraph is static, the number of queries a fork leaves over is not).
  int live_lo, live_hi, small_live;
  int bf16;                                // 1: A and W are single bf16 planes (training GEMMs, RPR_PREC_BF16); fp32 output only
  int prefer_pp;                           // 1: the 256x256 ping-pong kernel whatever the tile count, one K-loop per tile (weight gradients:
                                           // few tiles, thousands of K rows, several launches side by side on separate streams)
  int no_row_split;                        // 1: never split the rows of this launch over two kernels (packed encoder: M is a capacity far above
                                           // the live row count, which only the device knows)
  int kernel_cls;                          // out (host side): profile class of the kernel chosen (RPR_K_GEMM = 256x256 ping-pong, RPR_K_GEMM_SMALL = the others)
  // bf16 launches on the 256x256 kernel only (training step, round 6): the result leaves the epilogue in the operand formats
  // of the products that consume it, instead of one conversion launch per consumer reading the fp32 result back —
  // out_b [M][ldob] bf16 rows, out_bt [N][ldobt] bf16 = the transposed copy (the X^T / dY^T operand of a weight-gradient
  // product); mask_src [M][ldmask] fp32 (nullable): elements whose mask_src is not positive are written as zero (the ReLU
  // backward of the feed-forward block). out[0] may then be null (no fp32 result). M % 256 == 0, N % 256 == 0.
  void* out_b; int ldob; void* out_bt; int ldobt; const float* mask_src; int ldmask;
  // tile order of the persistent 256x256 kernel (set by its launcher): 0 = row-major; > 0 = bands of tile_rb row panels,
  // inside a band column groups of tile_cw tiles, inside a group panel by panel — the blocks of an XCD that run side by side
  // then cover tile_rb x tile_cw tiles and share tile_rb A panels and tile_cw W panels in their L2 (gemm_h2_pp_body.inc)
  int tile_cw, tile_rb;
};

// f16 has 5 exponent bits: a plane element below 2^-14 is subnormal, so the lo plane of x = hi + lo (|lo| ~ 2^-11 |x|)
// keeps its 11 bits only for |x| >= 0.125 — almost no T5 weight and few activations qualify. Planes therefore hold
// scaled values (powers of two, exact): weights x 2^8 (22 bits down to |w| ~ 5e-4, range |w| < 255), activations
// x 2^4 (|x| < 4094), and the FF intermediate relu(h Wi^T), the one tensor known to leave the f16 range on real T5
// checkpoints, x 2^-4 (|x| < 1.05e6). The GEMM epilogue undoes the product of the two scales.
constexpr float W_PLANE_SCALE = 256.0f, A_PLANE_SCALE = 16.0f, FF_PLANE_SCALE = 0.0625f;
// planes of the un-normalised residual stream x (fused RMSNorm): x 2^-4 like the FF intermediate, |x| < 1.05e6 — trained T5
// checkpoints carry a few residual channels of 1e3 .. 1e5 (the reason HF clamps fp16 T5); unscaled planes (round 4: |x| <
// 65504) sent every batch of such a model to the exact-fp32 path at 0.37 x the speed. Elements below 2 (x 2^-4 < 0.125) keep
// an absolute precision of 2^-21 (subnormal lo plane): 5e-7 relative to a row of O(1) RMS, against 6e-8 for an fp32 stream —
// measured on the goldens and on the heavy-tailed model of synth.make_state_dict(outliers=...): tests/test_gpu_edges.py,
// tools/precision_probe.py.
constexpr float X_PLANE_SCALE = 0.0625f;

// offset (in floats) of output element (m, on) in output block oi; on..on+3 stay inside one head
template <class G>
__device__ __forceinline__ size_t out_off(const G& g, int oi, int m, int ldo, int on) {
  if (g.rm_B && oi) {
    const int qi = m / g.rm_B;
    const int sh = g.rm_dshift ? g.rm_dshift : 6;   // log2 of the head dim: 6 (t5-base / large), 7 (t5-3b, d_kv = 128)
    return (size_t)qi * g.rm_stride + (size_t)(m - qi * g.rm_B) * g.rm_slot + (size_t)(on >> sh) * g.rm_head + (on & ((1 << sh) - 1));
  }
  return (size_t)m * ldo + on;
}
hipError_t launch_gemm_h2(GemmH2Args& a, hipStream_t s);
// several bf16 products C_i[M_i, N_i] = A_i[M_i, K] W_i[N_i, K]^T (fp32 output, no fused extras) in one launch of the 256x256
// ping-pong kernel, one block and one K-loop per tile (gemm_h2_pp_group_kernel): the weight gradients of one transformer
// layer. At most MAX_TILES tiles in total (the greedy assignment then stays within 64 per XCD); otherwise hipErrorInvalidValue.
struct GemmGroupArgs {
  static constexpr int MAXP = 8, MAX_TILES = 384, MAX_BLOCKS = 512;
  static constexpr size_t TABLE_BYTES = 4096, SCRATCH_BYTES = TABLE_BYTES + MAX_BLOCKS * sizeof(int);
  const __half* A[MAXP]; const __half* W[MAXP]; float* out[MAXP];
  int M[MAXP], N[MAXP], ldo[MAXP];
  int K, lda, ldw, n;
};
// scratch: SCRATCH_BYTES of device memory, written on stream s in front of the product launch (a later group on the same
// stream may reuse it)
hipError_t launch_gemm_h2_group(const GemmGroupArgs& p, void* scratch, hipStream_t s);
// colscale (nullable, length cols): element (r, c) is multiplied by colscale[c] before the split (folds a layer-norm
// weight into the columns of the consuming projection); cols is ignored when colscale is null
hipError_t launch_split_planes(const float* x, __half* out, size_t n, size_t plane_stride, hipStream_t s,
                               float scale = 1.0f, const float* colscale = nullptr, int cols = 0,
                               unsigned int* sat = nullptr);

// ---- T5 elementwise / attention kernels -----------------------------------------------------------
// post_scale: config.scaleup_output_hidden multiplies the final decoder norm by d_model**-0.5
// out (fp32) and/or out_h (two f16 planes, stride o_ps) are written; either may be null
// rows_dev (nullable): live row count on the device; rows past it are skipped (packed encoder)
// x_h (nullable): read the row from the residual-stream planes (stride x_ps) instead of x
hipError_t launch_rmsnorm(const float* x, const float* w, float* out, int rows, int d, float eps, hipStream_t s,
                          float post_scale = 1.0f, __half* out_h = nullptr, size_t o_ps = 0,
                          const int* rows_dev = nullptr, unsigned int* sat = nullptr, const __half* x_h = nullptr,
                          size_t x_ps = 0);
// Fused-RMSNorm producers (embedding kernels): when x_h is non-null the row goes to the f16 planes of the residual
// stream (X_PLANE_SCALE) instead of the fp32 buffer, and ssq receives the fixed-point sum of squares of the row as the
// planes hold it (stored — one wave owns a whole row).
struct XOut { __half* x_h; size_t x_ps; unsigned long long* ssq; unsigned int* sat; };
// a residual-stream element back from its planes (exact: hi and lo do not overlap)
__device__ __forceinline__ float x_from_planes(__half hi, __half lo) { return (__half2float(hi) + __half2float(lo)) * (1.0f / X_PLANE_SCALE); }
hipError_t init_t5_kernel_attributes();
hipError_t init_beam_kernel_attributes();
// row_src (nullable): out row p takes ids[row_src[p]] for p < *rows_dev (packed encoder)
hipError_t launch_embed_rows(const float* table, const int32_t* ids, float* out, int rows, int d, int vocab,
                             hipStream_t s, const int32_t* row_src = nullptr, const int* rows_dev = nullptr,
                             XOut xo = XOut{});
// Packed encoder rows: offs[q] = sum of lens[<q] (offs[Q] = live rows), row_src[offs[q] + j] = q * Lq + j
hipError_t launch_pack_rows(const int32_t* lens, int32_t* offs, int32_t* row_src, int Q, int Lq, hipStream_t s);
// x[r] = t==0 ? start : in_embeds[t-1][tokens[r][t-1]]
// rows_dev (nullable): live row count on the device (a compacted stage of the forced-tail search, api.hip)
hipError_t launch_dec_embed(const float* start, const float* in_embeds, const uint16_t* tokens, int tok_ld,
                            float* out, int R, int d, int V, int t, hipStream_t s, XOut xo = XOut{},
                            const int* rows_dev = nullptr);

struct EncAttnArgs {
  const float* qkv;        // [Q*Lq, 3*inner]  (q | k | v)
  const int32_t* mask;     // [Q, Lq]
  const float* rel_bias;   // [buckets, H]
  const int32_t* bucket;   // [2*MAX_LQ-1]: bucket of rel = key - query, index rel + MAX_LQ-1
  float* out;              // [Q*Lq, inner]
  int Q, Lq, H, buckets;
  __half* out_h; size_t o_ps;   // when non-null: write f16 planes instead of fp32
  const int32_t* offs;     // nullable (packed encoder): rows of query q are offs[q] .. offs[q] + lens[q] - 1
  const int32_t* lens;     //   instead of q*Lq .. q*Lq + Lq - 1
  unsigned int* sat;       // sticky saturation word (planes output)
  int causal;              // teacher-forced decoder self-attention (rpr_train_forward): key j <= query i only,
                           //   bucket table indexed by i - j (unidirectional)
  int mfma;                // 1: sequences of <= 32 padded positions with fp32 output may take the fp32-MFMA kernel (training
                           //   forward; the search encoder keeps the summation order of enc_attn_kernel)
  int dkv = 0;             // head dim (0 = 64); 128 (t5-3b) takes enc_attn_kernel<128>
};
hipError_t launch_enc_attn(const EncAttnArgs& a, hipStream_t s);
// tail_kernels.hip: the search encoder's attention (<= 32 positions) on the fp32-MFMA tile; false = shape not taken
bool launch_enc_attn_mfma_v2(const EncAttnArgs& a, hipStream_t s, hipError_t* err);
// training forward, Lq <= 32, padded layout, fp32 output: one wave per (sequence, head) on fp32 MFMA tiles (tail_kernels.hip)
hipError_t launch_train_self_attn_mfma(const EncAttnArgs& a, hipStream_t s);
// its backward (Ls <= 32): dqkv and the per-(sequence, head) bias-gradient parts, as self_attn_bwd_kernel writes them
hipError_t launch_train_self_attn_bwd_mfma(const float* qkv, const float* dO, const int32_t* mask, const float* rel_bias,
                                           const int32_t* bucket, float* dqkv, float* dbias_part, int S, int Ls, int H, int buckets,
                                           int causal, hipStream_t s);

struct DecSelfAttnArgs {
  const float* q;          // [R, inner]
  const float* kcache;     // this layer: the 64 floats of (q, head, p, slot) start at
  const float* vcache;     //   q*q_stride + head*h_stride + p*pos_stride + slot*slot_stride
  size_t q_stride, h_stride, pos_stride, slot_stride;
  const uint16_t* anc;     // [R, anc_ld]: slot (within the query) that produced position p < t
  int anc_ld;
  const float* rel_bias;   // [buckets, H]
  const int32_t* bucket;   // [MAX_DEC_LEN]: bucket of rel = -(n), n = t - p
  float* out;              // [R, inner]
  int Q, B, H, t;
  __half* out_h; size_t o_ps;
  unsigned int* sat;
  const int* nq_dev;       // nullable: live query count on the device (compacted stage); queries past it are skipped
  unsigned b_magic = 0, h_magic = 0;   // set by the launcher: reciprocals of B and H (kernel_utils.h udiv_magic)
  int dkv = 0;             // head dim (0 = 64); 128 (t5-3b) takes the generic kernel dec_attn_kernel<true, 128>
};
hipError_t launch_dec_self_attn(const DecSelfAttnArgs& a, hipStream_t s);

struct DecCrossAttnArgs {
  const float* q;          // [R, inner]
  const float* xk;         // K of this layer: row (q, j) at xk + (q*Lq + j) * xld
  const float* xv;
  int xld;
  const int32_t* mask;     // [Q, Lq]
  float* out;              // [R, inner]
  int Q, B, H, Lq;
  __half* out_h; size_t o_ps;
  const int32_t* last;     // [Q] index of the last attended key + 1 (launch_mask_lengths)
  const int32_t* offs;     // nullable (packed encoder): K/V row (q, j) is row offs[q] + j instead of q*Lq + j
  int bchunk;              // set by the launcher: beams per block when the beam is split over blockIdx.y (0 = all)
  unsigned int* sat;
  const int* nq_dev;       // nullable: live query count on the device; queries past it are skipped
  int dkv = 0;             // head dim (0 = 64); 128 (t5-3b) takes the generic kernel dec_attn_kernel<false, 128>
};
hipError_t launch_dec_cross_attn(const DecCrossAttnArgs& a, hipStream_t s);

// ---- beam state + trie-constrained selection ------------------------------------------------------
struct BeamState {           // one of two ping-pong buffers
  double* score;             // [R]
  int32_t* lo;               // [R] trie row range of the beam's prefix
  int32_t* hi;
  uint16_t* tokens;          // [R, ld]
  uint16_t* anc;             // [R, ld]
  int ld;
};
hipError_t launch_init_beams(const BeamState& st, int Q, int B, int64_t N, hipStream_t s);

// Scratch of the radix selection (select_radix.hip), carved from one buffer by select_radix_carve
struct RadixWs {
  unsigned* hist;            // [Q][3][2048] digit histograms of the three passes (zero between steps)
  unsigned* cnt;             // [Q][4] collected winners, collected ties
  unsigned long long* valid; // [Q][B * V / 64] child bitmap (bit = beam * V + token)
  float* lstat;              // [Q][B][2] log-softmax mode: row maximum, log of the sum
  int32_t* win;              // [Q][B] candidates above the threshold prefix
  int32_t* tie;              // [Q][B * V] candidates on it
  int32_t* chi;              // [R, V] end of the child's row range (SelectArgs::lb_scratch holds its start)
  int sort_cap;              // set by the launcher: entries of the finish kernel's LDS sort
};
constexpr int TRIE_NARROW = 64;   // trie nodes of at most this many rows have no entries in the deep child arrays (trie.h)
constexpr int TRIE_MAX_DEEP = 6;  // CSR levels 2 .. 7

struct SelectArgs {
  const float* logits;       // [R, V]
  const uint16_t* codes;     // sorted [N, Lc]
  int Lc;                    // row stride of codes (trie depth)
  BeamState cur, nxt;
  int32_t* lb_scratch;       // [R, V] lower bounds found by the mask phase
  const int32_t* lvl0;       // nullable: child arrays of trie levels 0 / 1 (rpr_trie::lvl0 / lvl1), row stride lvl_V
  const int32_t* lvl1;
  int lvl_V;
  // radix selection only: CSR child arrays of the levels 2 .. 2 + n_deep - 1 and the level-2 entry of every (c0, c1) node
  const int32_t* idx2;
  const int32_t* d_start[TRIE_MAX_DEEP]; const uint16_t* d_tok[TRIE_MAX_DEEP]; int d_n[TRIE_MAX_DEEP]; int n_deep;
  RadixWs rs;                // rs.hist != nullptr: the launch may take the radix path (launch_select decides)
  int Q, B, V, t;            // V: width of the token axis = the model's vocab rounded up to 64 (logits row stride)
  int Vreal;                 // the model's decoder vocab size (0 = V): tokens >= Vreal are padding and never selectable
  int log_softmax;
  int lds_logits;            // set by the launcher: stage the query's B*V logits in LDS
  int sort_lds, sort_off;    // set by the launcher: B > 256 -> bitonic sort of the candidate lists; byte offset of its LDS buffer
  int shared0;               // step 0 computed once per query: logits is [Q, V], position-0 K/V live in slot 0
  // debug taps for step t (nullable)
  double* tap_scores; int32_t* tap_tokens; int32_t* tap_parent;   // [Q, B]
  unsigned long long* tap_valid;                                   // [Q, B*V/64] phase-A child bitmap (bit = beam*V + token)
  unsigned long long* clk;   // debug (RPR_SELECT_CLOCK=1, eager mode): 8 wall-clock stamps of block 0 at the phase boundaries
  const int* nq_dev;         // nullable: live query count on the device; blocks past it exit
};
hipError_t launch_select(const SelectArgs& a, hipStream_t s);
// radix selection (select_radix.hip): many beams per query — five launches over all the CUs instead of one block per query
bool select_radix_fits(int B, int V);
bool select_radix_wanted(int B, int V);                  // the launcher's choice for this beam count (RPR_SELECT_RADIX overrides)
size_t select_radix_ws_bytes(int Q, int B, int V);
void select_radix_carve(RadixWs& w, void* base, int32_t* chi, int Q, int B, int V);
hipError_t init_select_radix_attributes();
hipError_t launch_select_radix_reset(const RadixWs& w, int Q, hipStream_t s);
hipError_t launch_select_radix(const SelectArgs& a, hipStream_t s);
bool select_fits(int B, int V);   // the beam's candidate bitmaps and state fit the 160 KB of LDS

struct FinalizeArgs {
  BeamState st;
  int Q, B, L;
  int32_t* out_tokens;       // [Q, B, L]
  float* out_scores;         // [Q, B]
  int64_t* out_lo;           // [Q, B]
  int64_t* out_hi;
  const int* nq_dev;         // nullable: live query count on the device; blocks past it exit
  const int32_t* qmap;       // nullable: stage query -> query of the call (output row block)
};
hipError_t launch_finalize(const FinalizeArgs& a, hipStream_t s);

// mask-only kernel for rpr_trie_mask: prefix rows -> child byte mask
hipError_t launch_prefix_mask(const uint16_t* codes, int Lc, int64_t N, const int32_t* prefix, int R, int T,
                              int V, uint8_t* out_mask, hipStream_t s);
hipError_t launch_zero_u64(unsigned long long* p, size_t n, hipStream_t s);
// status (nullable): the ctx's sticky "empty query" word, set to 1 when a query attends to no token
hipError_t launch_mask_lengths(const int32_t* mask, int32_t* lens, int Q, int Lq, hipStream_t s,
                               unsigned int* status = nullptr);

// ---- forced-tail evaluation (tail_kernels.hip; orchestration in api.hip) -------------------------------------------
// Once every beam of a query stands on a trie node under which a single distinct sequence remains, beam search can no
// longer prune for that query: each beam has exactly one valid child per step, the B valid candidates beat every
// masked one (-1e9) and the remaining tokens are the rest of the beam's code row. Such a query leaves the sequential
// steps at a FORK: its remaining positions are scored by one teacher-forced decoder pass (the "tail"), the others are
// compacted into the next stage and go on step by step. Same results as the sequential loop (reference
// generation.py:423-540); the reference itself has no counterpart — it recomputes the whole prefix every step.
struct ForkArgs {
  BeamState st;              // the stage's beams after step T-1 (T tokens each)
  const uint16_t* codes; int Lc;
  int Qcap; const int* nq_dev;   // stage capacity / live queries (nullable = Qcap)
  int B, T, L;
  double spread_max;         // forced only if max - min of the query's beam scores is below this (see api.hip)
  int32_t* flag;             // out [Qcap]: 1 = forced
};
hipError_t launch_fork_classify(const ForkArgs& a, hipStream_t s);
// exclusive scans of the flags: forced queries -> flist / tail_cnt {queries, sequences = x B, rows = x B x (L-T), 0},
// the others -> src (source stage query of every query of the next stage) / next_cnt {queries, rows = x B, 0, 0}
hipError_t launch_fork_scan(const int32_t* flag, int Qcap, const int* nq_dev, int B, int Lt, int32_t* flist, int32_t* tail_cnt,
                            int32_t* src, int32_t* next_cnt, hipStream_t s);
struct StageIO {             // per-query inputs of a stage (all indexed by the stage's query index)
  const int32_t* qmap;       // nullable = identity: query of the call
  const int32_t* offs;       // nullable = q * Lq: first encoder row
  const int32_t* last;       // attended length
  const int32_t* mask;       // [., Lq]
};
struct StageOut { int32_t* qmap; int32_t* offs; int32_t* last; int32_t* mask; };
// queries list[i] (i < *n_dev) of the source stage -> entry i of dst (qmap / offs / last / mask row)
hipError_t launch_gather_stage_io(const StageIO& src, const StageOut& dst, const int32_t* list, const int* n_dev, int Qcap, int Lq,
                                  hipStream_t s);
// beam state (T tokens per beam) of the source queries src[i] -> next stage's rows i*B ..
hipError_t launch_compact_beams(const BeamState& from, const BeamState& to, const int32_t* src, const int* n_dev, int Qcap, int B, int T,
                                hipStream_t s);
// K/V of positions < T of the source queries: (layer, q, head) regions [depth][B][64] -> the next stage's cache
struct KvCopyArgs {
  const float* k_from; const float* v_from; float* k_to; float* v_to;
  size_t layer_from, q_from, h_from;     // strides (floats) of the source cache
  size_t layer_to, q_to, h_to;
  const int32_t* src; const int* n_dev;
  int Qcap, nd, H, n;        // n = floats per (layer, q, head) to copy = T * B * 64
};
hipError_t launch_kv_copy(const KvCopyArgs& a, hipStream_t s);
// full token row of every forced beam: tokens[(i*B + b)*L + p] = p < T ? beam token : codes[lo_b][p]
hipError_t launch_tail_tokens(const BeamState& st, const uint16_t* codes, int Lc, const int32_t* flist, const int* nf_dev, int Qcap,
                              int B, int T, int L, uint16_t* tokens, hipStream_t s);
// decoder input embedding of the tail rows (sequence-major: row = seq * (L-T) + (p - T)): in_embeds[p-1][token p-1]
hipError_t launch_tail_embed(const float* in_embeds, const uint16_t* tokens, float* out, int rows, const int* rows_dev, int T, int L,
                             int d, int V, hipStream_t s, XOut xo = XOut{});
struct TailSelfAttnArgs {
  const float* qkv;          // [rows, 3*inner] of the tail rows (q | k | v)
  const float* kcache;       // the fork stage's cache of this layer (positions < T)
  const float* vcache;
  size_t q_stride, h_stride, pos_stride, slot_stride;
  const uint16_t* anc; int anc_ld;   // the fork stage's ancestry rows
  const int32_t* flist;      // tail query -> stage query
  const int* nseq_dev;       // live sequences
  const float* rel_bias; const int32_t* bucket;
  float* out; __half* out_h; size_t o_ps; unsigned int* sat;
  int nseq_cap, B, H, T, L;
  int dkv = 0;               // head dim (0 = 64); 128 (t5-3b) takes the VALU kernel tail_self_attn_kernel<128>
};
hipError_t launch_tail_self_attn(const TailSelfAttnArgs& a, hipStream_t s);
// cross-attention of the tail rows (a.B = rows per query): fp32-MFMA tiles for Lq <= 64, else the block kernel
hipError_t launch_tail_cross_attn(const DecCrossAttnArgs& a, hipStream_t s);
// cross-attention of a sequential step (a.B = beams per query): the MFMA tile kernel for Lq <= 32, else the block kernel
hipError_t launch_step_cross_attn(const DecCrossAttnArgs& a, hipStream_t s);
// gold[row] = <final RMSNorm of the row's stream (x post), E_out[p][token p]>, exact fp32 (as launch_gold_scores)
hipError_t launch_tail_gold(const float* x, const float* ln, const float* out_embeds, const uint16_t* tokens, float* gold, int rows,
                            const int* rows_dev, int T, int L, int d, int V, float eps, float post, hipStream_t s,
                            const __half* x_h = nullptr, size_t x_ps = 0);
// log-softmax mode: gold[row] = log_softmax(logits[row, :])[token of the row]
hipError_t launch_tail_logprob(const float* logits, const uint16_t* tokens, float* gold, int rows, const int* rows_dev, int T, int L,
                               int V, hipStream_t s);
struct TailRankArgs {
  BeamState st;              // the fork stage's beams (scores, ranges)
  const int32_t* flist; const int32_t* qmap; const int* nf_dev;   // qmap: tail query -> query of the call
  const uint16_t* tokens;    // [., B, L]
  const float* gold;         // [., B, L-T]
  int Qcap, B, T, L;
  int32_t* out_tokens; float* out_scores; int64_t* out_lo; int64_t* out_hi;
  int replay = 0;            // 1: always replay the L - T steps (RPR_TAIL_RANK_REPLAY=1; the tie path, for tests)
};
hipError_t launch_tail_rank(const TailRankArgs& a, hipStream_t s);
// max over the rows of |E[r] (*) w|_2 (w nullable): bound of the logits after the final RMSNorm (model load)
// *flag = 1 if *cnt != 0 (sticky status word)
hipError_t launch_flag_nonzero(const int* cnt, unsigned int* flag, hipStream_t s);
hipError_t launch_max_row_norm(const float* E, const float* w, int rows, int d, float* out /*zeroed*/, hipStream_t s);
hipError_t init_tail_kernel_attributes();

// ---- teacher-forced forward of the ranking fine-tune step (train_kernels.hip) -------------------
hipError_t launch_train_dec_embed(const float* start, const float* in_embeds, const int32_t* codes, float* out, int S,
                                  int L, int d, int V, hipStream_t s, XOut xo = XOut{});
hipError_t launch_gold_scores(const float* x, const float* ln, const float* out_embeds, const int32_t* codes, float* scores,
                              int S, int L, int d, int V, float eps, float post, hipStream_t s,
                              const __half* x_h = nullptr, size_t x_ps = 0);
hipError_t launch_margin_mse(const float* scores, const float* teacher_pos, const float* teacher_neg, const int32_t* prefix_lens,
                             int n_prefix, int bz, int L, float* losses, float* margins, hipStream_t s);
// ---- backward pass + optimizer of the same step (train_kernels.hip) ------------------------------------------------
hipError_t init_train_kernel_attributes();
// absolute maxima of one or two tensors -> out[0], out[1] by integer atomicMax: the slots must be zero beforehand
hipError_t launch_absmax2(const float* x0, size_t n0, const float* x1, size_t n1, float* out, hipStream_t s);
// fp32 [R, C] (row stride ldi) -> f16 hi/lo planes scaled by dyn_plane_scale(*amax):
//   plain:      out[2][R][C]            (plane stride R*C)
//   transposed: out_t[2][C][Rpad]       (plane stride C*Rpad, columns r >= R zero), optionally the plain planes as well
hipError_t launch_split_dyn(const float* x, int R, int C, int ldi, __half* out, const float* amax, hipStream_t s);
hipError_t launch_split_dyn_T(const float* x, int R, int C, int ldi, int Rpad, __half* out_t, __half* out_plain, const float* amax,
                              hipStream_t s);
// fp32 [R, C] (row stride ldi) -> one bf16 plane (round to nearest even): plain out[R][C] and / or transposed
// out_t[C][Rpad] (columns r >= R zero); RPR_PREC_BF16 training GEMMs
hipError_t launch_to_bf16(const float* x, int R, int C, int ldi, void* out, hipStream_t s);
hipError_t launch_to_bf16_T(const float* x, int R, int C, int ldi, int Rpad, void* out_t, void* out_plain, hipStream_t s,
                            const float* relu_act = nullptr, int ldt = 0);
// all GEMM weights at once: segs[i] = fp32 tensor [R][C] (R % 2 == 0, C % 4 == 0), off = element offset of its copies in
// `plain` ([R][C]) and `tr` ([C][R]); pref = exclusive prefix sums of ceil(R/64) * ceil(C/64)
struct WSeg { const float* src; int R, C; unsigned long long off; };
hipError_t launch_weights_bf16(const WSeg* segs, const int* pref, int nseg, int ntiles, void* plain, void* tr, hipStream_t s);
hipError_t launch_transpose_pad(const float* in, float* out, int R, int C, int ldi, int Rpad, hipStream_t s);
hipError_t launch_relu_bwd(float* dy, const float* act, size_t n, hipStream_t s);
// bf16 mode: RMSNorm written as the plain bf16 rows [rows][d] and their transposed copy [d][ldt] (columns rows .. Rpad zero)
hipError_t launch_rmsnorm_bf16_T(const float* x, const float* w, int rows, int d, float eps, float post, void* out_plain, void* out_t,
                                 int Rpad, int ldt, hipStream_t s);
hipError_t launch_rmsnorm_bwd(const float* x, const float* w, const float* dh, const float* dres, float* dx_out, float* w_part,
                              float* dw, int rows, int d, float eps, float post, int accumulate_dw, hipStream_t s);
// partials of several rmsnorm_bwd launches (dw == nullptr there) summed in one launch: out_i[k] = sum over p of part_i[p][k]
struct ColsumSites {
  static constexpr int MAXS = 4;
  const float* part[MAXS]; float* out[MAXS]; int nparts[MAXS]; int n;
};
int rmsnorm_bwd_parts(int rows);
hipError_t launch_colsum_multi(const ColsumSites& p, int d, hipStream_t s);
size_t self_attn_bwd_smem(int Ls, int buckets);
hipError_t launch_self_attn_bwd(const float* qkv, const float* dO, const int32_t* mask, const float* rel_bias, const int32_t* bucket,
                                float* dqkv, float* dbias_part, float* dbias, int S, int Ls, int H, int buckets, int causal,
                                hipStream_t s);
size_t cross_attn_bwd_smem(int n, int Lq);
hipError_t launch_cross_attn_bwd(const float* q, const float* xk, const float* xv, int xld, const int32_t* mask, const float* dO,
                                 float* dq, float* dxk, float* dxv, int bz, int n, int Lq, int H, hipStream_t s);
hipError_t launch_scatter_rows_fix(const float* src, const int32_t* idx, unsigned long long* acc, int rows, int d, hipStream_t s);
hipError_t launch_fix_flush(unsigned long long* acc, float* dst, size_t n, hipStream_t s);
hipError_t launch_train_indices(const int32_t* codes, int32_t* in_idx, int32_t* out_idx, int S, int L, int V, hipStream_t s);
hipError_t launch_sum_selected_rows(const float* src, const int32_t* sel, float* out, int rows, int d, hipStream_t s);
hipError_t launch_margin_mse_bwd(const float* margins, const float* teacher_pos, const float* teacher_neg, const int32_t* prefix_lens,
                                 int n_prefix, int bz, int L, float* dscores, hipStream_t s);
hipError_t launch_gold_score_bwd(const float* x, const float* ln, const float* out_embeds, const int32_t* out_idx, const float* dscores,
                                 float* dh, float* de, int rows, int d, float eps, float post, hipStream_t s);
hipError_t launch_grad_norm(const float* g, size_t n, double* part, int nparts, float max_norm, float* out, hipStream_t s);
hipError_t launch_adamw(float* p, const float* g, float* m, float* v, size_t n, const float* clip, float lr, float b1, float b2,
                        float eps, float wd, float bc1, float bc2_sqrt, hipStream_t s);
// every parameter tensor in one launch: segs[i] = {tensor, offset of its gradient / moments in the flat buffers, elements,
// decays?}; pref = exclusive prefix sums of ceil(n / 4096)
struct AdamSeg { float* p; unsigned long long off, n; int decay; };
hipError_t launch_adamw_multi(const AdamSeg* segs, const int* pref, int nseg, int nchunks, const float* g, float* m, float* v,
                              const float* clip, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                              hipStream_t s);

}  // namespace rpr
