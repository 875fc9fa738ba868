// Forced-tail evaluation of the trie-constrained beam search (gfx950, wave64): the fork (which queries can no longer
// be pruned, compaction of the others into the next stage) and the kernels of the teacher-forced tail pass that are
// not shared with the sequential steps. Orchestration: api.hip::enqueue_search.
//
// Semantics preserved (reference t5_pretrainer/tasks/generation.py): per step, candidate = ((double)logit_f32 +
// (valid ? 0 : -1e9)) + beam_score in float64 (:453-463); the first B of the sorted candidates become the new beams in
// that order, ties by ascending flat index beam*V + token (:484-503); finalize ranks by float64 sum/(L+1) descending
// with exact ties in reverse slot order and stores float32 (:532-540). For a forced query every beam has exactly one
// valid child per step, so the B winners of a step are those B candidates (fork_classify_kernel proves that no masked
// candidate can reach them) and only their ORDER has to be replayed: tail_rank_kernel.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "kernel_utils.h"

namespace rpr {

// ------------------------------------------------------------------------------------ fork
// One wave per stage query. forced <=> every beam is live (non-empty trie range), its range holds one distinct
// sequence over the columns T..L-1 (first row == last row there: the rows are sorted), and the spread of the beam
// scores is small enough that B valid continuations stay above every masked candidate for all remaining steps.
__global__ __launch_bounds__(256) void fork_classify_kernel(ForkArgs a) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= a.Qcap || (a.nq_dev && q >= *a.nq_dev)) return;
  const int r0 = q * a.B;
  bool ok = true;
  double smin = INFINITY, smax = -INFINITY;
  for (int b = lane; b < a.B; b += 64) {
    const int lo = a.st.lo[r0 + b], hi = a.st.hi[r0 + b];
    const double s = a.st.score[r0 + b];
    smin = fmin(smin, s); smax = fmax(smax, s);
    if (lo >= hi) { ok = false; continue; }
    if (hi - lo > 1) {
      const uint16_t* first = a.codes + (size_t)lo * a.Lc;
      const uint16_t* last = a.codes + (size_t)(hi - 1) * a.Lc;
      for (int p = a.T; p < a.L; ++p)
        if (first[p] != last[p]) { ok = false; break; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    smin = fmin(smin, __shfl_xor(smin, o, 64));
    smax = fmax(smax, __shfl_xor(smax, o, 64));
  }
  const bool all_ok = __all(ok);
  if (lane == 0) a.flag[q] = (all_ok && (smax - smin) < a.spread_max) ? 1 : 0;   // NaN scores compare false: not forced
}

hipError_t launch_fork_classify(const ForkArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(fork_classify_kernel, dim3((a.Qcap + 3) / 4), dim3(256), 0, s, a);
  return hipGetLastError();
}

// One block: the two lists in query order (forced -> flist, the others -> src) and the live counts of the tail pass
// and of the next stage.
__global__ __launch_bounds__(1024) void fork_scan_kernel(const int32_t* __restrict__ flag, int Qcap, const int* __restrict__ nq_dev,
                                                          int B, int Lt, int32_t* __restrict__ flist, int32_t* __restrict__ tail_cnt,
                                                          int32_t* __restrict__ src, int32_t* __restrict__ next_cnt) {
  __shared__ int part[1024];
  __shared__ int carry;
  const int tid = threadIdx.x;
  const int n = nq_dev ? min(*nq_dev, Qcap) : Qcap;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int q0 = 0; q0 < n; q0 += 1024) {
    const int q = q0 + tid;
    const int v = (q < n && flag[q] != 0) ? 1 : 0;
    part[tid] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {          // Hillis-Steele inclusive scan
      const int add = tid >= o ? part[tid - o] : 0;
      __syncthreads();
      part[tid] += add;
      __syncthreads();
    }
    if (q < n) {
      const int nf_before = carry + part[tid] - v;   // forced queries before q
      if (v) flist[nf_before] = q; else src[q - nf_before] = q;
    }
    __syncthreads();
    if (tid == 1023) carry += part[1023];
    __syncthreads();
  }
  if (tid == 0) {
    const int nf = carry, nu = n - carry;
    tail_cnt[0] = nf; tail_cnt[1] = nf * B; tail_cnt[2] = nf * B * Lt; tail_cnt[3] = 0;
    next_cnt[0] = nu; next_cnt[1] = nu * B; next_cnt[2] = 0; next_cnt[3] = 0;
  }
}

hipError_t launch_fork_scan(const int32_t* flag, int Qcap, const int* nq_dev, int B, int Lt, int32_t* flist, int32_t* tail_cnt,
                            int32_t* src, int32_t* next_cnt, hipStream_t s) {
  hipLaunchKernelGGL(fork_scan_kernel, dim3(1), dim3(1024), 0, s, flag, Qcap, nq_dev, B, Lt, flist, tail_cnt, src, next_cnt);
  return hipGetLastError();
}

// one wave per destination query
__global__ __launch_bounds__(256) void gather_stage_io_kernel(StageIO src, StageOut dst, const int32_t* __restrict__ list,
                                                               const int* __restrict__ n_dev, int Qcap, int Lq) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= Qcap || i >= *n_dev) return;
  const int q = list[i];
  if (lane == 0) {
    dst.qmap[i] = src.qmap ? src.qmap[q] : q;
    dst.offs[i] = src.offs ? src.offs[q] : q * Lq;
    dst.last[i] = src.last[q];
  }
  for (int j = lane; j < Lq; j += 64) dst.mask[(size_t)i * Lq + j] = src.mask[(size_t)q * Lq + j];
}

hipError_t launch_gather_stage_io(const StageIO& src, const StageOut& dst, const int32_t* list, const int* n_dev, int Qcap, int Lq,
                                  hipStream_t s) {
  hipLaunchKernelGGL(gather_stage_io_kernel, dim3((Qcap + 3) / 4), dim3(256), 0, s, src, dst, list, n_dev, Qcap, Lq);
  return hipGetLastError();
}

// one block per destination query
__global__ __launch_bounds__(256) void compact_beams_kernel(BeamState from, BeamState to, const int32_t* __restrict__ src,
                                                             const int* __restrict__ n_dev, int B, int T) {
  const int i = blockIdx.x, tid = threadIdx.x;
  if (i >= *n_dev) return;
  const int q = src[i];
  const size_t rf = (size_t)q * B, rt = (size_t)i * B;
  for (int b = tid; b < B; b += 256) {
    to.score[rt + b] = from.score[rf + b];
    to.lo[rt + b] = from.lo[rf + b];
    to.hi[rt + b] = from.hi[rf + b];
  }
  for (int k = tid; k < B * T; k += 256) {
    const int b = k / T, p = k - b * T;
    to.tokens[(rt + b) * to.ld + p] = from.tokens[(rf + b) * from.ld + p];
    to.anc[(rt + b) * to.ld + p] = from.anc[(rf + b) * from.ld + p];
  }
}

hipError_t launch_compact_beams(const BeamState& from, const BeamState& to, const int32_t* src, const int* n_dev, int Qcap, int B, int T,
                                hipStream_t s) {
  hipLaunchKernelGGL(compact_beams_kernel, dim3(Qcap), dim3(256), 0, s, from, to, src, n_dev, B, T);
  return hipGetLastError();
}

// block (i, layer * H + head): the first n floats of the (layer, query, head) region [depth][B][64] — positions < T
__global__ __launch_bounds__(256) void kv_copy_kernel(KvCopyArgs a) {
  const int i = blockIdx.x;
  if (i >= *a.n_dev) return;
  const int lh = blockIdx.y, layer = lh / a.H, h = lh - layer * a.H;
  const size_t of = (size_t)layer * a.layer_from + (size_t)a.src[i] * a.q_from + (size_t)h * a.h_from;
  const size_t ot = (size_t)layer * a.layer_to + (size_t)i * a.q_to + (size_t)h * a.h_to;
  const float4* kf = reinterpret_cast<const float4*>(a.k_from + of);
  const float4* vf = reinterpret_cast<const float4*>(a.v_from + of);
  float4* kt = reinterpret_cast<float4*>(a.k_to + ot);
  float4* vt = reinterpret_cast<float4*>(a.v_to + ot);
  for (int k = threadIdx.x; k < (a.n >> 2); k += 256) { kt[k] = kf[k]; vt[k] = vf[k]; }
}

hipError_t launch_kv_copy(const KvCopyArgs& a, hipStream_t s) {
  if (a.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(kv_copy_kernel, dim3(a.Qcap, a.nd * a.H), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------ tail pass
// one block per forced query: the whole token row of each of its beams
__global__ __launch_bounds__(256) void tail_tokens_kernel(BeamState st, const uint16_t* __restrict__ codes, int Lc,
                                                           const int32_t* __restrict__ flist, const int* __restrict__ nf_dev,
                                                           int B, int T, int L, uint16_t* __restrict__ tokens) {
  const int i = blockIdx.x;
  if (i >= *nf_dev) return;
  const int q = flist[i];
  for (int k = threadIdx.x; k < B * L; k += 256) {
    const int b = k / L, p = k - b * L;
    const size_t r = (size_t)q * B + b;
    tokens[((size_t)i * B + b) * L + p] = p < T ? st.tokens[r * st.ld + p] : codes[(size_t)st.lo[r] * Lc + p];
  }
}

hipError_t launch_tail_tokens(const BeamState& st, const uint16_t* codes, int Lc, const int32_t* flist, const int* nf_dev, int Qcap,
                              int B, int T, int L, uint16_t* tokens, hipStream_t s) {
  hipLaunchKernelGGL(tail_tokens_kernel, dim3(Qcap), dim3(256), 0, s, st, codes, Lc, flist, nf_dev, B, T, L, tokens);
  return hipGetLastError();
}

// decoder input embeddings of the tail rows (reference t5_generative_retriever.py:194-214: position p >= 1 takes
// list_decoder_embeds[p-1][token p-1]); one wave per row
__global__ __launch_bounds__(256) void tail_embed_kernel(const float* __restrict__ in_embeds, const uint16_t* __restrict__ tokens,
                                                          float* __restrict__ out, int rows, const int* __restrict__ rows_dev,
                                                          int T, int L, int d, int V, XOut xo) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows || row >= *rows_dev) return;
  const int Lt = L - T, seq = row / Lt, p = T + (row - seq * Lt);
  const int tok = tokens[(size_t)seq * L + (p - 1)];
  copy_row_x(reinterpret_cast<const float4*>(in_embeds + ((size_t)(p - 1) * V + tok) * d), out, row, d, lane, xo);
}

hipError_t launch_tail_embed(const float* in_embeds, const uint16_t* tokens, float* out, int rows, const int* rows_dev, int T, int L,
                             int d, int V, hipStream_t s, XOut xo) {
  if (rows <= 0 || T < 1) return rows <= 0 ? hipSuccess : hipErrorInvalidValue;
  hipLaunchKernelGGL(tail_embed_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, in_embeds, tokens, out, rows, rows_dev, T, L, d, V, xo);
  return hipGetLastError();
}

// Causal self-attention of the tail positions of one beam: one block per (sequence, head). K and V of all L positions
// are staged once in LDS — positions < T from the fork stage's KV cache through the beam's ancestry (written by the
// sequential steps, never moved), positions >= T from this pass's own q|k|v rows — then every wave handles query
// positions T + wave, T + wave + 4, ...: lane j scores key j (L <= 64), the q row is broadcast with v_readlane,
// softmax across the wave, P.V with lane = output dim. Arithmetic of dec_self_attn_fast_kernel / enc_attn_kernel
// (unscaled scores + unidirectional relative bias, fp32 softmax normalised before P.V).
// D = head dim: 64, or 128 (t5-3b; two q / output values per lane) — the fp32-MFMA tiles below are written for 64.
template <int D>
__global__ __launch_bounds__(256) void tail_self_attn_kernel(TailSelfAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NV = D / 64;           // q / output values per lane
  const int H = a.H, L = a.L, T = a.T, Lt = L - T, inner = H * D, ld = 3 * inner;
  const int seq = blockIdx.x / H, h = blockIdx.x - seq * H;
  if (seq >= *a.nseq_dev) return;
  float* Ks = smem;                          // [L][D + 1]
  float* Vs = smem + (size_t)L * (D + 1);    // [L][D]
  float* Ps = Vs + (size_t)L * D;            // [4][64]
  float* Bs = Ps + 4 * 64;                   // [buckets <= 64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = seq / a.B, b = seq - fi * a.B;
  const int qi = a.flist[fi];
  const uint16_t* ancr = a.anc + ((size_t)qi * a.B + b) * a.anc_ld;
  const size_t cbase = (size_t)qi * a.q_stride + (size_t)h * a.h_stride;
  const float* tbase = a.qkv + (size_t)seq * Lt * ld + h * D;
  for (int i = tid; i < L * (D / 4); i += 256) {
    const int j = i / (D / 4), c = (i - j * (D / 4)) * 4;
    float4 kv, vv;
    if (j < T) {
      const size_t off = cbase + (size_t)j * a.pos_stride + (size_t)ancr[j] * a.slot_stride + c;
      kv = *reinterpret_cast<const float4*>(a.kcache + off);
      vv = *reinterpret_cast<const float4*>(a.vcache + off);
    } else {
      const float* r = tbase + (size_t)(j - T) * ld + c;
      kv = *reinterpret_cast<const float4*>(r + inner);
      vv = *reinterpret_cast<const float4*>(r + 2 * inner);
    }
    float* kd = Ks + j * (D + 1) + c;
    kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
    *reinterpret_cast<float4*>(Vs + j * D + c) = vv;
  }
  if (tid < 64) Bs[tid] = a.rel_bias[a.bucket[tid] * H + h];   // bias of distance n = i - j (bucket table: MAX_DEC_LEN = 64 entries)
  __syncthreads();
  float* P = Ps + wave * 64;
  for (int i = T + wave; i < L; i += 4) {
    float qv[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) qv[v] = tbase[(size_t)(i - T) * ld + 64 * v + lane];  // lane d holds q_i[d], q_i[64 + d]
    const int jc = lane <= i ? lane : i;
    const float* kr = Ks + jc * (D + 1);
    float acc = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int d = 0; d < 64; ++d) {
        const float qd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qv[v]), d));
        acc = fmaf(qd, kr[64 * v + d], acc);
      }
    const float sc = lane <= i ? acc + Bs[i - lane] : -INFINITY;
    const float mx = wave_max(sc);
    const float e = (sc == -INFINITY) ? 0.f : expf(sc - mx);
    const float sum = wave_sum(e);
    P[lane] = e / sum;
    __builtin_amdgcn_wave_barrier();
    float o[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) o[v] = 0.f;
    for (int j = 0; j <= i; ++j)
#pragma unroll
      for (int v = 0; v < NV; ++v) o[v] = fmaf(P[j], Vs[j * D + 64 * v + lane], o[v]);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const size_t oidx = ((size_t)seq * Lt + (i - T)) * inner + h * D + 64 * v + lane;
      if (a.out_h) {
        __half hi, lo;
        split_f16(o[v] * A_PLANE_SCALE, hi, lo, a.sat);
        a.out_h[oidx] = hi;
        a.out_h[a.o_ps + oidx] = lo;
      } else {
        a.out[oidx] = o[v];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- fp32-MFMA attention tiles (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation) ---------------------
// One wave handles 32 query rows x up to 32*NKT keys of one head. Scores are computed TRANSPOSED, S^T = K Q^T, so that
// in the MFMA result layout a lane owns ONE query row (n = lane & 31) and 16 of its keys per key tile
// (m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), r = register): the softmax of a row is 16*NKT in-register values plus one
// exchange with the partner lane (lane ^ 32) instead of 32-lane butterflies. The probabilities then feed the P.V
// product as its A operand WITHOUT moving: an MFMA reduces over its k slots in any order, so slot (kk, half) is
// declared to be key kappa(kk, half) = (kk & 3) + 8 (kk >> 2) + 4 half — exactly the key register kk already holds —
// and the B operand reads V[kappa][d] from the wave's LDS strip. K and Q come straight from global memory: a lane
// reads the eight 16-byte pieces {8c + 4 half .. +3} of its row, MFMA 4c + x consumes component x of piece c (the
// same k-slot freedom). The VALU version of the tail self-attention spent 64 LDS reads + 64 FMAs per (row, key lane)
// and ran at 4.3 ms per layer; this one is bound by the 64 fp32 MFMAs per 32-row tile.
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int kappa(int kk, int half) { return (kk & 3) + 8 * (kk >> 2) + 4 * half; }

// the eight 16-byte pieces of a 64-float row owned by this lane half; null pointer -> zeros
__device__ __forceinline__ void load_row_pieces(const float* row, int half, float4 (&r)[8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c)
    r[c] = row ? *reinterpret_cast<const float4*>(row + c * 8 + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ void mfma_scores(const float4 (&k)[8], const float4 (&q)[8], f32x16& s) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k[c].x, q[c].x, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k[c].y, q[c].y, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k[c].z, q[c].z, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k[c].w, q[c].w, s, 0, 0, 0);
  }
}

// softmax of this lane's row over its 16*NKT keys and the partner lane's (scores of masked keys are -inf); returns
// the normalised probabilities in place. A row without any valid key gets all zeros.
template <int NKT>
__device__ __forceinline__ void softmax_rows(f32x16 (&s)[NKT]) {
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = (s[kt][r] == -INFINITY) ? 0.f : expf(s[kt][r] - mx);
      s[kt][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kt][r] *= inv;
}

// O[row][d] += sum_key P[row][key] V[key][d] for the two 32-column halves of the head; Vs = [32*NKT][64] in LDS
template <int NKT>
__device__ __forceinline__ void mfma_pv(const f32x16 (&p)[NKT], const float* Vs, int lane, f32x16 (&o)[2]) {
  const int d = lane & 31, half = lane >> 5;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float* vr = Vs + (kt * 32 + kappa(kk, half)) * 64 + d;
      o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p[kt][kk], vr[0], o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p[kt][kk], vr[32], o[1], 0, 0, 0);
    }
}

// The 32 x 64 output tile (MFMA result layout: a lane holds one column and 16 scattered rows per 32-column half) goes
// through the wave's LDS strip (the V rows are consumed by then) and leaves row-wise: a lane takes 8 consecutive columns
// of a row = one 16-byte store per f16 plane (or two float4). Storing straight from the MFMA layout was 64 two-byte
// stores per lane and tile.
__device__ __forceinline__ void store_o_tile(const f32x16 (&o)[2], float* strip, int lane, int i0, int nrows, size_t row_base, int inner,
                                             int hcol, float* out, __half* out_h, size_t o_ps, unsigned int* sat) {
  const int d = lane & 31, half = lane >> 5;
  __builtin_amdgcn_wave_barrier();                       // every P.V read of the strip has been issued
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
    strip[i * 64 + d] = o[0][r];
    strip[i * 64 + 32 + d] = o[1][r];
  }
  __builtin_amdgcn_wave_barrier();
  const int c8 = (lane & 7) * 8;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int il = k * 8 + (lane >> 3), i = i0 + il;     // 8 rows per pass, 8 lanes per row
    if (i >= nrows) continue;
    const float4 v0 = *reinterpret_cast<const float4*>(strip + il * 64 + c8);
    const float4 v1 = *reinterpret_cast<const float4*>(strip + il * 64 + c8 + 4);
    const size_t oidx = (row_base + i) * inner + hcol + c8;
    if (out_h) {
      __half h[8], l[8];
      const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) split_f16(x[e] * A_PLANE_SCALE, h[e], l[e], sat);
      *reinterpret_cast<uint4*>(out_h + oidx) = *reinterpret_cast<uint4*>(h);
      *reinterpret_cast<uint4*>(out_h + o_ps + oidx) = *reinterpret_cast<uint4*>(l);
    } else {
      *reinterpret_cast<float4*>(out + oidx) = v0;
      *reinterpret_cast<float4*>(out + oidx + 4) = v1;
    }
  }
  __builtin_amdgcn_wave_barrier();                       // the strip is rewritten by the next row tile's V / O
}

// Tail self-attention on the fp32 matrix cores: one wave per (sequence, head), NKT = ceil(L / 32) key tiles.
template <int NKT, int OCC = 2>
__global__ __launch_bounds__(256, OCC) void tail_self_attn_mfma_kernel(TailSelfAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H, L = a.L, T = a.T, Lt = L - T, inner = H * DKV, ld = 3 * inner;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5;
  const int w = blockIdx.x * 4 + wave;
  const int seq = w / H, h = w - seq * H;
  if (seq >= *a.nseq_dev) return;                       // wave-uniform
  // per wave: V rows [32 NKT][64], the bias table [64] and — only when a sequence has more than one tile of 32 tail rows
  // (NKT == 2), where V must survive the first tile's output — a separate 32 x 64 output strip
  constexpr int WAVE_FLOATS = NKT * 32 * 64 + 64 + (NKT > 1 ? 32 * 64 : 0);
  float* Vs = smem + (size_t)wave * WAVE_FLOATS;
  float* Bs = Vs + NKT * 32 * 64;
  float* Os = NKT > 1 ? Bs + 64 : Vs;
  const int fi = seq / a.B, b = seq - fi * a.B;
  const int qi = a.flist[fi];
  const uint16_t* ancr = a.anc + ((size_t)qi * a.B + b) * a.anc_ld;
  const size_t cbase = (size_t)qi * a.q_stride + (size_t)h * a.h_stride;
  const float* tbase = a.qkv + (size_t)seq * Lt * ld + h * DKV;
  {  // V rows -> LDS, four coalesced 256-B rows per instruction; rows past L are zero (0 * garbage must stay 0)
    const int g = lane >> 4, li = lane & 15;
#pragma unroll
    for (int it = 0; it < NKT * 8; ++it) {
      const int j = it * 4 + g;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < L) {
        const float* vr = j < T ? a.vcache + cbase + (size_t)j * a.pos_stride + (size_t)ancr[j] * a.slot_stride
                                : tbase + (size_t)(j - T) * ld + 2 * inner;
        v = *reinterpret_cast<const float4*>(vr + li * 4);
      }
      *reinterpret_cast<float4*>(Vs + j * 64 + li * 4) = v;
    }
  }
  Bs[lane] = a.rel_bias[a.bucket[lane] * H + h];         // bias of distance n = query position - key position
  float4 kreg[NKT][8];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    const int j = kt * 32 + (lane & 31);
    const float* kr = nullptr;
    if (j < L)
      kr = j < T ? a.kcache + cbase + (size_t)j * a.pos_stride + (size_t)ancr[j] * a.slot_stride : tbase + (size_t)(j - T) * ld + inner;
    load_row_pieces(kr, half, kreg[kt]);
  }
  __builtin_amdgcn_wave_barrier();
  for (int i0 = 0; i0 < Lt; i0 += 32) {
    const int i = i0 + (lane & 31);                      // this lane's query row (position T + i)
    float4 qreg[8];
    load_row_pieces(i < Lt ? tbase + (size_t)i * ld : nullptr, half, qreg);
    f32x16 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
      mfma_scores(kreg[kt], qreg, s[kt]);
    }
    const int pq = T + (i < Lt ? i : Lt - 1);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = kt * 32 + kappa(r, half);
        s[kt][r] = j <= pq ? s[kt][r] + Bs[pq - j] : -INFINITY;
      }
    softmax_rows<NKT>(s);
    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    mfma_pv<NKT>(s, Vs, lane, o);
    store_o_tile(o, Os, lane, i0, Lt, (size_t)seq * Lt, inner, h * DKV, a.out, a.out_h, a.o_ps, a.sat);
  }
}

// Cross-attention of the tail rows on the fp32 matrix cores: one wave per (query, head, tile of 32 of the query's
// rows); keys = the query's own encoder rows (<= 32 * NKT, padding keys masked), no position bias.
template <int NKT>
__global__ __launch_bounds__(256, 4) void tail_cross_attn_mfma_kernel(DecCrossAttnArgs a, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H, inner = H * DKV, nrows = a.B;       // a.B = rows of one query (beams x tail positions)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5;
  const int w = blockIdx.x * 4 + wave;
  const int tile = w % tiles, qh = w / tiles, h = qh % H, qi = qh / H;
  if (qi >= a.Q || (a.nq_dev && qi >= *a.nq_dev)) return;   // wave-uniform
  float* Vs = smem + (size_t)wave * (NKT * 32 * 64);
  const int nk = min(a.last[qi], NKT * 32);
  const int32_t* mrow = a.mask + (size_t)qi * a.Lq;
  const size_t xrow0 = a.offs ? (size_t)a.offs[qi] : (size_t)qi * a.Lq;
  const float* kb = a.xk + xrow0 * a.xld + h * DKV;
  const float* vb = a.xv + xrow0 * a.xld + h * DKV;
  {
    const int g = lane >> 4, li = lane & 15;
#pragma unroll
    for (int it = 0; it < NKT * 8; ++it) {
      const int j = it * 4 + g;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < nk && mrow[j] != 0) v = *reinterpret_cast<const float4*>(vb + (size_t)j * a.xld + li * 4);
      *reinterpret_cast<float4*>(Vs + j * 64 + li * 4) = v;
    }
  }
  float4 kreg[NKT][8];
  bool kok[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    const int j = kt * 32 + (lane & 31);
    kok[kt] = j < nk && mrow[j] != 0;
    load_row_pieces(kok[kt] ? kb + (size_t)j * a.xld : nullptr, half, kreg[kt]);
  }
  __builtin_amdgcn_wave_barrier();
  const int i0 = tile * 32, i = i0 + (lane & 31);
  const size_t row_base = (size_t)qi * nrows;
  float4 qreg[8];
  load_row_pieces(i < nrows ? a.q + (row_base + i) * inner + h * DKV : nullptr, half, qreg);
  f32x16 s[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
    mfma_scores(kreg[kt], qreg, s[kt]);
  }
  // validity of key kappa(r, half) of tile kt: held by the lane whose (lane & 31) is that key — one ballot per tile
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    const unsigned long long okm = __ballot(kok[kt]) & 0xffffffffull;   // bit j: key kt*32 + j is attended
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (!((okm >> kappa(r, half)) & 1ull)) s[kt][r] = -INFINITY;
  }
  softmax_rows<NKT>(s);
  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  mfma_pv<NKT>(s, Vs, lane, o);
  store_o_tile(o, Vs, lane, i0, nrows, row_base, inner, h * DKV, a.out, a.out_h, a.o_ps, a.sat);
}

// ---- second generation of the fp32-MFMA attention tiles ----------------------------------------------------------------
// On half of the chip (a lane of the search) the first generation is bound by instruction issue, not by memory: a wave
// spends ~1300 VALU instructions (x 4 cycles) beside its 64 MFMAs (x 64 cycles) — per-lane 64-bit address arithmetic
// (the wave index came from threadIdx, so nothing was known to be uniform), a branch and a flag store per value in the
// plane split, V staged through LDS with its own address math, mask and bias applied value by value — and runs two waves
// per SIMD (162 VGPRs). tools/tail_attn_probe.hip on 128 CUs: self 1503 us for 3.96 GB, cross 1009 us for 1.93 GB.
// This generation produces the same bits with a fraction of the instructions:
//  * the wave index goes through v_readfirstlane: sequence, head, query and every base pointer are scalars (SALU), the
//    loads take the scalar-base + 32-bit-offset form;
//  * V goes straight into the B operand of P.V (lane = (d, key slot): d is the contiguous index of a V row, 128
//    contiguous bytes per lane half and instruction) — no LDS staging, no fragment reads;
//  * the plane split is branch-free (one saturation test per lane at the end);
//  * cross-attention: a wave keeps K and V of its (query, head) in registers and walks up to TPW row tiles; the key mask is
//    the C operand of the first score MFMA (-inf + x = -inf), P.V skips the key slots beyond the query's length (a
//    12-token query uses 8 of 16), K and the Q tiles come through LDS-DMA (global_load_lds_dwordx4: whole 256-byte row
//    slices into an XOR-swizzled strip, conflict-free ds_read_b128 operand reads) with the next tile's Q in flight under
//    the current tile's products.
__device__ __forceinline__ void dma_rows16(const float* src, float* lds_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_uniform, 16, 0, 0);
}

// operand pieces of this lane's row out of a swizzled strip: piece 2c + half of row `row` sits in slot piece ^ (row & 15)
__device__ __forceinline__ void read_row_pieces(const float* strip, int row, int half, float4 (&r)[8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c)
    r[c] = *reinterpret_cast<const float4*>(strip + row * 64 + (((2 * c + half) ^ (row & 15)) << 2));
}

// scores with the C operand of the first product given (zeros, or the additive key mask)
__device__ __forceinline__ f32x16 mfma_scores_c(const float4 (&k)[8], const float4 (&q)[8], const f32x16& c0) {
  f32x16 s = __builtin_amdgcn_mfma_f32_32x32x2f32(k[0].x, q[0].x, c0, 0, 0, 0);
  s = __builtin_amdgcn_mfma_f32_32x32x2f32(k[0].y, q[0].y, s, 0, 0, 0);
  s = __builtin_amdgcn_mfma_f32_32x32x2f32(k[0].z, q[0].z, s, 0, 0, 0);
  s = __builtin_amdgcn_mfma_f32_32x32x2f32(k[0].w, q[0].w, s, 0, 0, 0);
#pragma unroll
  for (int c = 1; c < 8; ++c) {
    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k[c].x, q[c].x, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k[c].y, q[c].y, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k[c].z, q[c].z, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k[c].w, q[c].w, s, 0, 0, 0);
  }
  return s;
}

__device__ __forceinline__ void mfma_pv_regs(const f32x16& p, const float (&v0)[16], const float (&v1)[16], int kk_end, f32x16 (&o)[2]) {
  const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p[0], v0[0], z, 0, 0, 0);
  o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p[0], v1[0], z, 0, 0, 0);
#pragma unroll
  for (int kk = 1; kk < 16; ++kk) {
    if (kk < kk_end) {   // wave-uniform
      o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p[kk], v0[kk], o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p[kk], v1[kk], o[1], 0, 0, 0);
    }
  }
}

// split_f16 of eight values, two at a time (v_pk_mul_f32, v_cvt_pk_f16_f32, v_pk_add_f32) and without its branch: the
// same planes for every input; lanes with a value outside the f16 range (or NaN) are collected in `bad` (a wave mask: SALU)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8(const float4& a, const float4& b, uint4& hi, uint4& lo, unsigned long long& bad) {
  const f32x2 x[4] = {{a.x, a.y}, {a.z, a.w}, {b.x, b.y}, {b.z, b.w}};
  f16x2 h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f32x2 v = x[e] * A_PLANE_SCALE;
    bad |= __ballot(!(fabsf(v.x) <= 65504.f)) | __ballot(!(fabsf(v.y) <= 65504.f));
    v.x = fminf(fmaxf(v.x, -65504.f), 65504.f);
    v.y = fminf(fmaxf(v.y, -65504.f), 65504.f);
    h[e] = __builtin_convertvector(v, f16x2);
    l[e] = __builtin_convertvector(v - __builtin_convertvector(h[e], f32x2), f16x2);
  }
  hi = *reinterpret_cast<uint4*>(h); lo = *reinterpret_cast<uint4*>(l);
}

// store_o_tile with a scalar tile base (out_t / out_h_t point at row i0, column hcol of the head) and 32-bit offsets
__device__ __forceinline__ void store_o_tile_v2(const f32x16 (&o)[2], float* strip, int lane, int nlive, int inner, float* out_t,
                                                __half* out_h_t, size_t o_ps, unsigned long long& bad) {
  const int d = lane & 31, half = lane >> 5;
  __builtin_amdgcn_wave_barrier();
  float* wr = strip + (4 * half) * 64 + d;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2);
    wr[i * 64] = o[0][r];
    wr[i * 64 + 32] = o[1][r];
  }
  __builtin_amdgcn_wave_barrier();
  const int c8 = (lane & 7) * 8, il0 = lane >> 3;
  const float* rd = strip + il0 * 64 + c8;
  const int off0 = il0 * inner + c8;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k * 8 + il0 < nlive) {                           // 8 rows per pass, 8 lanes per row
      const float4 x0 = *reinterpret_cast<const float4*>(rd + k * 8 * 64);
      const float4 x1 = *reinterpret_cast<const float4*>(rd + k * 8 * 64 + 4);
      const int off = off0 + k * 8 * inner;
      if (out_h_t) {
        uint4 hi, lo;
        split8(x0, x1, hi, lo, bad);
        *reinterpret_cast<uint4*>(out_h_t + off) = hi;
        *reinterpret_cast<uint4*>(out_h_t + o_ps + off) = lo;
      } else {
        *reinterpret_cast<float4*>(out_t + off) = x0;
        *reinterpret_cast<float4*>(out_t + off + 4) = x1;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// softmax_rows<1> with exp_nonpos (a masked score is -inf: x = -inf - max falls under the cut; at least one key of a row
// is attended wherever this is called). Registers r >= r_end (wave-uniform, a multiple of 4) hold masked keys only: their
// probability is 0 without an exponential.
__device__ __forceinline__ void softmax_row16(f32x16& s, int r_end = 16) {
  float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
#pragma unroll
  for (int q4 = 1; q4 < 4; ++q4)
    if (q4 * 4 < r_end) mx = fmaxf(mx, fmaxf(fmaxf(s[q4 * 4], s[q4 * 4 + 1]), fmaxf(s[q4 * 4 + 2], s[q4 * 4 + 3])));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    if (q4 * 4 < r_end) {
#pragma unroll
      for (int r = q4 * 4; r < q4 * 4 + 4; ++r) {
        const float e = exp_nonpos(s[r] - mx);
        s[r] = e;
        sum += e;
      }
    } else {
#pragma unroll
      for (int r = q4 * 4; r < q4 * 4 + 4; ++r) s[r] = 0.f;
    }
  }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s[r] *= inv;
}

// Tail self-attention for L <= 32 and a fork depth T <= 8 (every search of the bench's kind; other shapes take the first
// generation): one wave per (sequence, head); block = four heads of one sequence (blockIdx.x = sequence * HB + head
// block, divided by multiplication). Per wave: one 8-KB strip — the reversed bias table, then the output tile. No
// branches on lanes: every address is valid (padding lanes and key slots past L repeat a live row: as query rows they are
// not stored, as keys the causal rule masks them; key slots past L are skipped four at a time).
template <int OCC>
__global__ __launch_bounds__(256, OCC) void tail_self_attn_mfma_v2_kernel(TailSelfAttnArgs a, int HB, unsigned hb_magic, unsigned b_magic) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H, L = a.L, T = a.T, Lt = L - T, inner = H * DKV, ld = 3 * inner;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, half = lane >> 5, ln = lane & 31;
  const int seq = udiv_magic(blockIdx.x, HB, hb_magic);
  const int h = ((int)blockIdx.x - seq * HB) * 4 + wave;
  if (h >= H || seq >= *a.nseq_dev) return;              // wave-uniform
  float* Os = smem + wave * (32 * 64);
  const int fi = udiv_magic((unsigned)seq, a.B, b_magic), b = seq - fi * a.B;
  const int qi = a.flist[fi];
  const uint16_t* ancr = a.anc + ((size_t)qi * a.B + b) * a.anc_ld;
  const float* kc = a.kcache + (size_t)qi * a.q_stride + (size_t)h * a.h_stride;
  const float* vc = a.vcache + (size_t)qi * a.q_stride + (size_t)h * a.h_stride;
  const float* tbase = a.qkv + (size_t)seq * Lt * ld + h * DKV;
  const int kk_end = L > 24 ? 16 : L > 16 ? 12 : L > 8 ? 8 : 4;   // key slots kk >= kk_end hold keys >= L in both halves
  float4 kreg[8], qreg[8];
  {
    const int j = min(ln, L - 1);
    const int slot = ancr[j];                            // defined for positions < T only; the pointer built on it is not used elsewhere
    const float* kr = j < T ? kc + (size_t)j * a.pos_stride + (size_t)slot * a.slot_stride : tbase + (j - T) * ld + inner;
    load_row_pieces(kr, half, kreg);
    load_row_pieces(tbase + min(ln, Lt - 1) * ld, half, qreg);
  }
  // bias of distance n = query position - key position, reversed: Os[31 - n]; the mask / bias pass reads Os[31 - pq + j]
  Os[lane < 32 ? 31 - lane : lane] = lane < 32 ? a.rel_bias[a.bucket[ln] * H + h] : 0.f;
  float v0[16], v1[16];                                  // V[kappa(kk, half)][d], [d + 32]: the B operand of P.V
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {                       // keys 0..7: cache rows below T
    const int j = min(kk + 4 * half, L - 1);
    const int slot = ancr[j];
    const float* vr = j < T ? vc + (size_t)j * a.pos_stride + (size_t)slot * a.slot_stride : tbase + (j - T) * ld + 2 * inner;
    v0[kk] = vr[ln]; v1[kk] = vr[ln + 32];
  }
  const int voff = 4 * half * ld + ln;
#pragma unroll
  for (int q4 = 1; q4 < 4; ++q4) {
    if (q4 * 4 < kk_end) {                               // wave-uniform; keys >= 8 are rows of this pass in both halves
#pragma unroll
      for (int kk = q4 * 4; kk < q4 * 4 + 4; ++kk) {
        const int j0 = kappa(kk, 0);
        if (j0 + 4 < L) {                                // wave-uniform: scalar row base + one per-lane offset for all slots
          const float* sb = tbase + 2 * inner + (j0 - T) * ld;
          v0[kk] = sb[voff]; v1[kk] = sb[voff + 32];
        } else {
          const float* vr = tbase + 2 * inner + (min(j0 + 4 * half, L - 1) - T) * ld;
          v0[kk] = vr[ln]; v1[kk] = vr[ln + 32];
        }
      }
    } else {
#pragma unroll
      for (int kk = q4 * 4; kk < q4 * 4 + 4; ++kk) { v0[kk] = 0.f; v1[kk] = 0.f; }
    }
  }
  const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 sc = mfma_scores_c(kreg, qreg, z);
  const int pq = T + min(ln, Lt - 1);
  {
    const float* brow = Os + (31 - pq + 4 * half);
    const int jl = pq - 4 * half;                        // key kappa(r, 0) + 4 half is visible iff kappa(r, 0) <= jl
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j0 = kappa(r, 0);
      sc[r] = j0 <= jl ? sc[r] + brow[j0] : -INFINITY;
    }
  }
  softmax_row16(sc, kk_end);
  f32x16 o[2];
  mfma_pv_regs(sc, v0, v1, kk_end, o);
  unsigned long long bad = 0ull;
  const size_t obase = (size_t)seq * Lt * inner + h * DKV;
  store_o_tile_v2(o, Os, lane, Lt, inner, a.out ? a.out + obase : nullptr, a.out_h ? a.out_h + obase : nullptr, a.o_ps, bad);
  if (bad != 0ull && a.sat && lane == 0) *a.sat = 1u;
}

// Cross-attention of the tail rows, Lq <= 32: one wave per (query, head, group of TPW row tiles); block = four heads
// (blockIdx.x = (query * groups + group) * HB + head block). K, V and the key mask of the (query, head) stay in registers
// for all of the wave's tiles; the next tile's Q rows are requested (into a second register set) before the current
// tile's products. Per wave: one 8-KB output strip. (A version with K and Q through LDS-DMA strips was no faster: the
// compiler fences every LDS read behind a pending LDS-DMA with vmcnt(0), which also waits for the tile's stores.)
template <int TPW, int OCC, bool PREF = true>
__global__ __launch_bounds__(256, OCC) void tail_cross_attn_mfma_v2_kernel(DecCrossAttnArgs a, int groups, int HB, unsigned hb_magic,
                                                                         unsigned g_magic) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H, inner = H * DKV, nrows = a.B;       // a.B = rows of one query (beams x tail positions)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, half = lane >> 5, ln = lane & 31;
  const int qg = udiv_magic(blockIdx.x, HB, hb_magic);    // query * groups + group
  const int h = ((int)blockIdx.x - qg * HB) * 4 + wave;
  const int qi = udiv_magic((unsigned)qg, groups, g_magic), grp = qg - qi * groups;
  if (h >= H || qi >= a.Q || (a.nq_dev && qi >= *a.nq_dev)) return;   // wave-uniform
  int i0 = grp * TPW * 32;
  if (i0 >= nrows) return;
  float* Os = smem + wave * (32 * 64);
  const int nk = min(a.last[qi], 32);
  const size_t obase = (size_t)qi * nrows * inner + h * DKV;
  unsigned long long bad = 0ull;
  if (nk == 0) {   // query without a single attended token: zeros (as the block kernel; its packed encoder has no rows to read)
    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    for (int t = 0; t < TPW && i0 < nrows; ++t, i0 += 32) {
      const size_t ob = obase + (size_t)i0 * inner;
      store_o_tile_v2(o, Os, lane, nrows - i0, inner, a.out ? a.out + ob : nullptr, a.out_h ? a.out_h + ob : nullptr, a.o_ps, bad);
    }
    return;
  }
  const int32_t* mrow = a.mask + (size_t)qi * a.Lq;
  const size_t xrow0 = a.offs ? (size_t)a.offs[qi] : (size_t)qi * a.Lq;
  const float* kb = a.xk + xrow0 * a.xld + h * DKV;
  const float* vb = a.xv + xrow0 * a.xld + h * DKV;
  const float* qb = a.q + (size_t)qi * nrows * inner + h * DKV;
  float4 kreg[8], qreg[8];
  load_row_pieces(kb + min(ln, nk - 1) * a.xld, half, kreg);          // unattended keys are masked below: any finite row will do
  // this lane's pieces of row ibase + ln; rows past the end repeat the last one (their output is not stored)
  auto q_load = [&](int ibase, float4 (&r)[8]) { load_row_pieces(qb + (size_t)ibase * inner + min(ln, nrows - 1 - ibase) * inner, half, r); };
  q_load(i0, qreg);
  const bool kok = ln < nk && mrow[min(ln, nk - 1)] != 0;
  const unsigned okm = (unsigned)(__ballot(kok) & 0xffffffffull);   // bit j: key j is attended
  // key slots kk >= kk_end hold keys >= nk in both halves (kappa(kk, 1) = kappa(kk, 0) + 4): their P is 0, skip them
  const int kk_end = nk > 24 ? 16 : nk > 16 ? 12 : nk > 8 ? 8 : 4;
  float v0[16], v1[16];
  f32x16 negm;                                            // additive key mask = the C operand of the first score product
  {
    const unsigned okh = half ? okm >> 4 : okm;           // key kappa(kk, 0) + 4 half
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      if (q4 * 4 < kk_end) {                              // wave-uniform
#pragma unroll
        for (int kk = q4 * 4; kk < q4 * 4 + 4; ++kk) {
          const int j0 = kappa(kk, 0);
          const bool live = (okh >> j0) & 1u;
          const float* vr = vb + min(j0 + 4 * half, nk - 1) * a.xld;
          const float x0 = vr[ln], x1 = vr[ln + 32];
          negm[kk] = live ? 0.f : -INFINITY;
          v0[kk] = live ? x0 : 0.f; v1[kk] = live ? x1 : 0.f;   // unattended keys: zero rows keep 0 * garbage out of the sum
        }
      } else {
#pragma unroll
        for (int kk = q4 * 4; kk < q4 * 4 + 4; ++kk) { negm[kk] = -INFINITY; v0[kk] = 0.f; v1[kk] = 0.f; }
      }
    }
  }
#pragma unroll 1
  for (int t = 0; t < TPW; ++t) {
    const int inext = i0 + 32;
    const bool more = t + 1 < TPW && inext < nrows;       // wave-uniform
    float4 qnext[8];
    if (PREF && more) q_load(inext, qnext);               // next tile's Q rows under this tile's products
    f32x16 sc = mfma_scores_c(kreg, qreg, negm);
    softmax_row16(sc, kk_end);
    f32x16 o[2];
    mfma_pv_regs(sc, v0, v1, kk_end, o);
    const size_t ob = obase + (size_t)i0 * inner;
    store_o_tile_v2(o, Os, lane, nrows - i0, inner, a.out ? a.out + ob : nullptr, a.out_h ? a.out_h + ob : nullptr, a.o_ps, bad);
    if (!more) break;
    i0 = inext;
    if (PREF) {
#pragma unroll
      for (int c = 0; c < 8; ++c) qreg[c] = qnext[c];
    } else {
      q_load(i0, qreg);                                   // three waves per SIMD instead of a second register set
    }
  }
  if (bad != 0ull && a.sat && lane == 0) *a.sat = 1u;
}

// Encoder self-attention of the search (bidirectional bias, key padding mask, packed or padded rows, <= 32 positions) on
// the same tile: one wave per (query, head). The VALU block kernel (enc_attn_kernel: 64 v_readlane + 64 LDS reads + 64
// FMAs per query row, half of the lanes idle at <= 32 keys) took 226 us per layer for a lane's 1075 packed queries.
__global__ __launch_bounds__(256, 4) void enc_attn_mfma_v2_kernel(EncAttnArgs a, int HB, unsigned hb_magic) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H, inner = H * DKV, ld = 3 * inner;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, half = lane >> 5, ln = lane & 31;
  const int qi = udiv_magic(blockIdx.x, HB, hb_magic);
  const int h = ((int)blockIdx.x - qi * HB) * 4 + wave;
  if (h >= H) return;                                     // wave-uniform
  const int nrow = a.offs ? a.lens[qi] : a.Lq;
  if (nrow == 0) return;                                  // a query without a token has no rows
  const size_t row0 = a.offs ? (size_t)a.offs[qi] : (size_t)qi * a.Lq;
  const float* base = a.qkv + row0 * ld + h * DKV;
  float* Os = smem + wave * (32 * 64);
  const int lnc = min(ln, nrow - 1);                      // padding lanes repeat the last row (masked as keys, not stored as rows)
  float4 kreg[8], qreg[8];
  load_row_pieces(base + lnc * ld + inner, half, kreg);
  load_row_pieces(base + lnc * ld, half, qreg);
  // bias of rel = key - query in [-31, 31]: Os[rel + 31]
  Os[lane] = lane < 63 ? a.rel_bias[a.bucket[lane - 31 + (MAX_LQ - 1)] * H + h] : 0.f;
  const bool kok = ln < nrow && a.mask[(size_t)qi * a.Lq + lnc] != 0;
  const unsigned okm = (unsigned)(__ballot(kok) & 0xffffffffull);   // bit j: key j is attended
  const unsigned okh = half ? okm >> 4 : okm;             // key kappa(kk, 0) + 4 half
  const int kk_end = nrow > 24 ? 16 : nrow > 16 ? 12 : nrow > 8 ? 8 : 4;
  float v0[16], v1[16];
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    if (q4 * 4 < kk_end) {                                // wave-uniform
#pragma unroll
      for (int kk = q4 * 4; kk < q4 * 4 + 4; ++kk) {
        const int j0 = kappa(kk, 0);
        const float* vr = base + min(j0 + 4 * half, nrow - 1) * ld + 2 * inner;
        const float x0 = vr[ln], x1 = vr[ln + 32];
        const bool live = (okh >> j0) & 1u;
        v0[kk] = live ? x0 : 0.f; v1[kk] = live ? x1 : 0.f;   // unattended keys: zero rows keep 0 * garbage out of the sum
      }
    } else {
#pragma unroll
      for (int kk = q4 * 4; kk < q4 * 4 + 4; ++kk) { v0[kk] = 0.f; v1[kk] = 0.f; }
    }
  }
  const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 sc = mfma_scores_c(kreg, qreg, z);
  {
    const float* brow = Os + (31 - lnc + 4 * half);       // key j = kappa(r, 0) + 4 half: rel + 31 = j - i + 31
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j0 = kappa(r, 0);
      sc[r] = ((okh >> j0) & 1u) ? sc[r] + brow[j0] : -INFINITY;
    }
  }
  softmax_row16(sc, kk_end);
  f32x16 o[2];
  mfma_pv_regs(sc, v0, v1, kk_end, o);
  unsigned long long bad = 0ull;
  const size_t obase = row0 * inner + h * DKV;
  store_o_tile_v2(o, Os, lane, nrow, inner, a.out ? a.out + obase : nullptr, a.out_h ? a.out_h + obase : nullptr, a.o_ps, bad);
  if (bad != 0ull && a.sat && lane == 0) *a.sat = 1u;
}

// Cross-attention of a sequential step for at most 32 encoder positions on v_mfma_f32_16x16x4_f32: one wave per (query,
// head, group of 16-beam tiles). The 32 x 32 tile above spends 64 MFMAs of 64 cycles on the 10 live rows of a beam-10 step (neutral
// against the VALU block kernel); a 16 x 16 tile is 16 + 16 MFMAs of 32 cycles for up to 16 keys. Layouts (lane l: c = l & 15,
// ks = l >> 4): S^T = K Q^T with A = K (key c of the tile, dims 16 ks .. 16 ks + 15: MFMA i consumes component i, the same
// k-slot freedom as above), B = Q (beam c, same dims); the result puts keys 4 ks + r (r = 0..3) of beam c in lane l, so a
// row's softmax is 4 (8) in-register values and two exchanges (lane ^ 16, lane ^ 32); P then feeds the A operand of
// O = P V without moving (slot ks of step r = key 4 ks + r), B = V[key][16 t + c] straight from memory for the four
// 16-column tiles t. The 16 x 64 output goes through a padded LDS strip (68 floats per row: the four ks groups hit
// disjoint banks) and leaves as 16-byte plane stores.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool MULTI>   // false: at most 16 beams — one tile, no tile loop (35.0 against 37.5 us per lane launch at beam 10)
__global__ __launch_bounds__(256, 4) void step_cross_attn_mfma16_kernel(DecCrossAttnArgs a, int HB, unsigned hb_magic, int groups,
                                                                      unsigned g_magic, int tpw) {
  // one wave per (query, head, group of tpw 16-row tiles): blockIdx.x = (query * groups + group) * HB + head block; K, V and
  // the key mask of the (query, head) stay in registers for all of the wave's tiles (beams > 16: 7 tiles at beam 100)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int SLD = 68;
  const int H = a.H, inner = H * DKV, B = a.B;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, c = lane & 15, ks = lane >> 4;
  const int qg = udiv_magic(blockIdx.x, HB, hb_magic);
  const int h = ((int)blockIdx.x - qg * HB) * 4 + wave;
  const int qi = MULTI ? udiv_magic((unsigned)qg, groups, g_magic) : qg, grp = MULTI ? qg - qi * groups : 0;
  if (h >= H || (a.nq_dev && qi >= *a.nq_dev)) return;    // wave-uniform
  int i0 = MULTI ? grp * tpw * 16 : 0;
  if (i0 >= B) return;
  float* Os = smem + wave * (16 * SLD);
  const int nk = min(a.last[qi], 32);
  const size_t obase = (size_t)qi * B * inner + h * DKV;
  unsigned long long bad = 0ull;
  // nk == 0 (a query without a single attended token): zeros, as the block kernel; nothing of its encoder is read
  const int nkt = nk > 16 ? 2 : nk > 0 ? 1 : 0;           // key tiles of 16
  const float* qb = a.q + (size_t)qi * B * inner + h * DKV + 16 * ks;
  // this lane's 16 dims of row ibase + c; rows past the end repeat the last one (not stored). The first tile's rows are
  // requested before K and V (they are needed first), a later tile's while the previous one is stored.
  float4 qreg[4];
  auto q_load = [&](int ibase) {
    const float* qr = qb + (size_t)ibase * inner + min(c, B - 1 - ibase) * inner;
#pragma unroll
    for (int u = 0; u < 4; ++u) qreg[u] = *reinterpret_cast<const float4*>(qr + 4 * u);
  };
  q_load(i0);
  float4 kreg[2][4];
  float vreg[2][4][4];                                    // [key tile][r][column tile]: V[16 kt + 4 ks + r][16 t + c]
  f32x4 negm[2];                                          // additive key mask = the C operand of the first score product
  if (nkt > 0) {
    const int32_t* mrow = a.mask + (size_t)qi * a.Lq;
    const size_t xrow0 = a.offs ? (size_t)a.offs[qi] : (size_t)qi * a.Lq;
    const float* kb = a.xk + xrow0 * a.xld + h * DKV;
    const float* vb = a.xv + xrow0 * a.xld + h * DKV;
    const bool kok = (lane & 31) < nk && mrow[min(lane & 31, nk - 1)] != 0;
    const unsigned okm = (unsigned)(__ballot(kok) & 0xffffffffull);   // bit j: key j is attended
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (kt < nkt) {                                     // wave-uniform; unattended keys are masked: any finite row will do
        const float* kr = kb + min(kt * 16 + c, nk - 1) * a.xld + 16 * ks;
#pragma unroll
        for (int u = 0; u < 4; ++u) kreg[kt][u] = *reinterpret_cast<const float4*>(kr + 4 * u);
        const unsigned bits = (okm >> (16 * kt + 4 * ks)) & 0xfu;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool live = (bits >> r) & 1u;
          const float* vr = vb + min(16 * kt + 4 * ks + r, nk - 1) * a.xld + c;
#pragma unroll
          for (int t = 0; t < 4; ++t) { const float x = vr[16 * t]; vreg[kt][r][t] = live ? x : 0.f; }
          negm[kt][r] = live ? 0.f : -INFINITY;
        }
      }
    }
  }
  const int c8 = (lane & 7) * 8, il0 = lane >> 3;
#pragma unroll 1
  for (int tl = 0; tl < (MULTI ? tpw : 1) && i0 < B; ++tl, i0 += 16) {
    f32x4 o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nkt > 0) {
      f32x4 sc[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        if (kt < nkt) {
          sc[kt] = negm[kt];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kreg[kt][u].x, qreg[u].x, sc[kt], 0, 0, 0);
            sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kreg[kt][u].y, qreg[u].y, sc[kt], 0, 0, 0);
            sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kreg[kt][u].z, qreg[u].z, sc[kt], 0, 0, 0);
            sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kreg[kt][u].w, qreg[u].w, sc[kt], 0, 0, 0);
          }
        }
      }
      // softmax of row c over its keys: 4 per key tile here, the others in lanes ^ 16, ^ 32
      float mx = fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3]));
      if (nkt > 1) mx = fmaxf(mx, fmaxf(fmaxf(sc[1][0], sc[1][1]), fmaxf(sc[1][2], sc[1][3])));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        if (kt < nkt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float e = exp_nonpos(sc[kt][r] - mx); sc[kt][r] = e; sum += e; }
        }
      }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        if (kt < nkt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = sc[kt][r] * inv;
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, vreg[kt][r][t], o[t], 0, 0, 0);
          }
        }
      }
    }
    if (MULTI && tl + 1 < tpw && i0 + 16 < B) q_load(i0 + 16);   // wave-uniform
    // o[t][r] = O[row 4 ks + r][16 t + c] -> strip -> rows of 8 lanes x 8 columns
    __builtin_amdgcn_wave_barrier();
    {
      float* wr = Os + (4 * ks) * SLD + c;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) wr[r * SLD + 16 * t] = o[t][r];
    }
    __builtin_amdgcn_wave_barrier();
    const size_t ob = obase + (size_t)i0 * inner;
    float* out_t = a.out ? a.out + ob : nullptr;
    __half* out_h_t = a.out_h ? a.out_h + ob : nullptr;
    const int nlive = B - i0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = k * 8 + il0;
      if (i < nlive) {
        const float4 x0 = *reinterpret_cast<const float4*>(Os + i * SLD + c8);
        const float4 x1 = *reinterpret_cast<const float4*>(Os + i * SLD + c8 + 4);
        const int off = i * inner + c8;
        if (out_h_t) {
          uint4 hi, lo;
          split8(x0, x1, hi, lo, bad);
          *reinterpret_cast<uint4*>(out_h_t + off) = hi;
          *reinterpret_cast<uint4*>(out_h_t + a.o_ps + off) = lo;
        } else {
          *reinterpret_cast<float4*>(out_t + off) = x0;
          *reinterpret_cast<float4*>(out_t + off + 4) = x1;
        }
      }
    }
  }
  if (bad != 0ull && a.sat && lane == 0) *a.sat = 1u;
}

// Which generation of the fp32-MFMA tail attention runs: 1 = direct K / Q loads and V through LDS, 2 = K (and Q) through
// LDS-DMA, V direct (bit-identical results). RPR_TAIL_ATTN_GEN; tools/tail_attn_probe.hip switches it per launch.
int g_tail_attn_gen = [] { const char* e = dev_getenv("RPR_TAIL_ATTN_GEN"); return e ? atoi(e) : 2; }();
int g_tail_attn_opt = [] { const char* e = dev_getenv("RPR_TAIL_ATTN_OPT"); return e ? atoi(e) : 0; }();
int g_tail_cross_tpw = [] { const char* e = dev_getenv("RPR_TAIL_CROSS_TPW"); return e ? atoi(e) : 0; }();   // 0 = by size

// the search encoder's attention on the MFMA tile (launch_enc_attn asks); false = not taken
bool launch_enc_attn_mfma_v2(const EncAttnArgs& a, hipStream_t s, hipError_t* err) {
  static const bool on = [] { const char* e = dev_getenv("RPR_ENC_ATTN_MFMA"); return !e || atoi(e) != 0; }();
  const int HB = (a.H + 3) / 4;
  if (!on || g_tail_attn_gen != 2 || a.causal || !a.mask || a.Lq > 32 || a.buckets > 64 || (long)a.Q * HB >= (1l << 31) / HB) return false;
  hipLaunchKernelGGL(enc_attn_mfma_v2_kernel, dim3((unsigned)(a.Q * HB)), dim3(256), 4 * (32 * 64) * sizeof(float), s, a, HB, div_magic(HB));
  *err = hipGetLastError();
  return true;
}

hipError_t launch_tail_cross_attn(const DecCrossAttnArgs& a, hipStream_t s) {
  static const bool off = [] { const char* e = dev_getenv("RPR_TAIL_ATTN_MFMA"); return e && atoi(e) == 0; }();
  if (off || a.Lq > 64 || a.dkv == 128) return launch_dec_cross_attn(a, s);   // long queries, 128-dim heads: the block kernel (any Lq <= 256)
  const int tiles = (a.B + 31) / 32;
  if (g_tail_attn_gen == 2 && a.Lq <= 32) {
    // tiles per wave: many tiles -> a wave keeps K / V for nine of them (fewer, longer waves: better on a lane's half of
    // the chip); few -> one tile per wave (more waves to fill the chip)
    const int want = g_tail_cross_tpw > 0 ? g_tail_cross_tpw : ((long)a.Q * a.H * tiles >= 32768 ? 9 : 1);
    const int tpw = want >= 9 ? 9 : want >= 3 ? 3 : 1;
    const int groups = (tiles + tpw - 1) / tpw, HB = (a.H + 3) / 4;
    const long blocks = (long)a.Q * groups * HB;
    if (blocks < (1l << 31) / HB && (long)a.Q * groups < (1l << 32) / groups && (long)a.B * a.H * DKV < (1l << 29)) {   // udiv_magic / 32-bit offsets
      const dim3 grid((unsigned)blocks), blk(256);
      const size_t smem = 4 * (32 * 64) * sizeof(float);
      const unsigned hm = div_magic(HB), gm = div_magic(groups);
      const bool o3 = !(g_tail_attn_opt & 2);   // default: three waves per SIMD, no second Q register set (658 vs 673 us per lane launch)
      if (tpw == 9 && o3) hipLaunchKernelGGL((tail_cross_attn_mfma_v2_kernel<9, 3, false>), grid, blk, smem, s, a, groups, HB, hm, gm);
      else if (tpw == 9) hipLaunchKernelGGL((tail_cross_attn_mfma_v2_kernel<9, 2>), grid, blk, smem, s, a, groups, HB, hm, gm);
      else if (tpw == 3 && o3) hipLaunchKernelGGL((tail_cross_attn_mfma_v2_kernel<3, 3, false>), grid, blk, smem, s, a, groups, HB, hm, gm);
      else if (tpw == 3) hipLaunchKernelGGL((tail_cross_attn_mfma_v2_kernel<3, 2>), grid, blk, smem, s, a, groups, HB, hm, gm);
      else hipLaunchKernelGGL((tail_cross_attn_mfma_v2_kernel<1, 4>), grid, blk, smem, s, a, groups, HB, hm, gm);
      return hipGetLastError();
    }
  }
  const long waves = (long)a.Q * a.H * tiles;
  const dim3 grid((unsigned)((waves + 3) / 4)), blk(256);
  if (a.Lq <= 32) hipLaunchKernelGGL(tail_cross_attn_mfma_kernel<1>, grid, blk, 4 * 32 * 64 * sizeof(float), s, a, tiles);
  else hipLaunchKernelGGL(tail_cross_attn_mfma_kernel<2>, grid, blk, 4 * 64 * 64 * sizeof(float), s, a, tiles);
  return hipGetLastError();
}

// Cross-attention of a sequential step (a.B = the beams of a query). RPR_STEP_CROSS_MFMA: 2 (default) = the 16 x 16 tile
// kernel above for at most 32 encoder positions, else the VALU block kernel; 1 = the 32-row tile kernel of the
// tail (measured neutral at beam 10: 4969-4974 vs 4983 queries/s same-box, 10 of a tile's 32 rows are live); 0 = always the
// block kernel.
hipError_t launch_step_cross_attn(const DecCrossAttnArgs& a, hipStream_t s) {
  static const int mode = [] { const char* e = dev_getenv("RPR_STEP_CROSS_MFMA"); return e ? atoi(e) : 2; }();
  const int HB = (a.H + 3) / 4;
  if (a.dkv == 128) return launch_dec_cross_attn(a, s);   // t5-3b heads: the generic kernel
  if (mode == 2 && g_tail_attn_gen == 2 && a.Lq <= 32 && (long)a.B * a.H * DKV < (1l << 29)) {
    // 16-row tiles per wave: one while that gives the chip enough waves, else up to eight (K / V / mask loaded once per wave)
    const int tiles = (a.B + 15) / 16;
    const int tpw = (long)a.Q * a.H * tiles >= 32768 ? std::min(tiles, 8) : 1;
    const int groups = (tiles + tpw - 1) / tpw;
    const long blocks = (long)a.Q * groups * HB;
    if (blocks < (1l << 31) / HB && (long)a.Q * groups < (1l << 32) / groups) {   // udiv_magic's exact range
      if (tiles == 1)
        hipLaunchKernelGGL(step_cross_attn_mfma16_kernel<false>, dim3((unsigned)blocks), dim3(256), 4 * (16 * 68) * sizeof(float), s, a, HB,
                           div_magic(HB), 1, 0u, 1);
      else
        hipLaunchKernelGGL(step_cross_attn_mfma16_kernel<true>, dim3((unsigned)blocks), dim3(256), 4 * (16 * 68) * sizeof(float), s, a, HB,
                           div_magic(HB), groups, div_magic(groups), tpw);
      return hipGetLastError();
    }
  }
  if (mode == 1 && g_tail_attn_gen == 2 && a.Lq <= 32) return launch_tail_cross_attn(a, s);
  return launch_dec_cross_attn(a, s);
}

// Self-attention of the training forward (teacher-forced decoder: causal, bias by distance i - j; encoder: key padding
// mask, bidirectional bias by j - i) for sequences of at most 32 positions, on the same fp32-MFMA tiles: one wave per
// (sequence, head), all keys in one tile. qkv [S * Ls, 3 inner] -> out [S * Ls, inner] (fp32). The block-per-head VALU kernel
// (enc_attn_kernel) spends 64 readlane + 64 LDS reads + 64 FMAs per query row with half of the lanes idle at 32 keys:
// 61 us per layer of 256 sequences x 12 heads, issue-bound; here a head is 64 MFMAs.
template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void train_self_attn_mfma_kernel(EncAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H, Ls = a.Lq, inner = H * DKV, ld = 3 * inner;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5;
  const int w = blockIdx.x * 4 + wave;
  const int seq = w / H, h = w - seq * H;
  if (seq >= a.Q) return;                               // wave-uniform
  float* Vs = smem + (size_t)wave * (32 * 64 + 64);
  float* Bs = Vs + 32 * 64;
  const float* base = a.qkv + (size_t)seq * Ls * ld + h * DKV;
  {  // V rows -> LDS, four coalesced 256-B rows per instruction; rows past Ls are zero (0 * garbage must stay 0)
    const int g = lane >> 4, li = lane & 15;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int j = it * 4 + g;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < Ls) v = *reinterpret_cast<const float4*>(base + (size_t)j * ld + 2 * inner + li * 4);
      *reinterpret_cast<float4*>(Vs + j * 64 + li * 4) = v;
    }
  }
  // bias per key offset: causal n = i - j in [0, Ls); bidirectional j - i + Ls - 1 in [0, 2 Ls - 1)
  if (lane < (CAUSAL ? Ls : 2 * Ls - 1)) Bs[lane] = a.rel_bias[a.bucket[CAUSAL ? lane : lane - (Ls - 1) + (MAX_LQ - 1)] * H + h];
  const int n = lane & 31;
  const bool kok = n < Ls && (CAUSAL || a.mask[(size_t)seq * Ls + n] != 0);
  float4 kreg[8], qreg[8];
  load_row_pieces(n < Ls ? base + (size_t)n * ld + inner : nullptr, half, kreg);
  load_row_pieces(n < Ls ? base + (size_t)n * ld : nullptr, half, qreg);
  __builtin_amdgcn_wave_barrier();
  f32x16 s[1];
#pragma unroll
  for (int r = 0; r < 16; ++r) s[0][r] = 0.f;
  mfma_scores(kreg, qreg, s[0]);
  const unsigned long long okm = __ballot(kok) & 0xffffffffull;   // bit j: key j is attended
  const int iq = n < Ls ? n : Ls - 1;                    // rows past Ls are computed on clamped indices and never stored
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = kappa(r, half);
    const bool ok = ((okm >> j) & 1ull) && (!CAUSAL || j <= iq);
    const int bi = CAUSAL ? iq - j : j - iq + Ls - 1;
    s[0][r] = ok ? s[0][r] + Bs[ok ? bi : 0] : -INFINITY;
  }
  softmax_rows<1>(s);
  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  mfma_pv<1>(s, Vs, lane, o);
  store_o_tile(o, Vs, lane, 0, Ls, (size_t)seq * Ls, inner, h * DKV, a.out, nullptr, 0, nullptr);
}

hipError_t launch_train_self_attn_mfma(const EncAttnArgs& a, hipStream_t s) {
  if (a.Lq > 32 || a.Lq < 1 || a.offs || a.out_h || a.buckets > 64) return hipErrorInvalidValue;
  const long waves = (long)a.Q * a.H;
  const dim3 grid((unsigned)((waves + 3) / 4)), blk(256);
  const size_t smem = 4 * (32 * 64 + 64) * sizeof(float);
  if (a.causal) hipLaunchKernelGGL(train_self_attn_mfma_kernel<true>, grid, blk, smem, s, a);
  else hipLaunchKernelGGL(train_self_attn_mfma_kernel<false>, grid, blk, smem, s, a);
  return hipGetLastError();
}

// Backward of the same attention (reference: autograd through T5Attention inside loss.backward(), tasks/trainer.py:203-275)
// for sequences of at most 32 positions: one wave per (sequence, head), seven 32 x 32 (x 64) products on the fp32 matrix
// cores. With P = softmax(S), S = Q K^T + bias, O = P V:
//   dP = dO V^T,  dS = P * (dP - rowsum(dP * P)),  dQ = dS K,  dK = dS^T Q,  dV = P^T dO,  dbias[bucket] += diagonals of dS.
// dQ wants dS with a lane per QUERY (its A operand's row), dK and dV want dS and P with a lane per KEY: both layouts are
// computed by the matrix cores — S^T = K Q^T and dP^T = V dO^T put a query in a lane (as the forward kernel), S = Q K^T and
// dP = dO V^T a key — and the per-query softmax statistics (maximum, 1 / sum, rowsum(dP * P)) found in the first layout are
// passed to the second through 96 floats of LDS; two more score products cost 64 MFMAs, a transposition of P and dS
// through LDS would cost 64 LDS accesses per lane and two more strips. K, Q and dO are staged row-major in LDS as the B
// operands of dQ / dK / dV (each strip then serves as the transposition scratch of its own product's output tile); the
// VALU kernel (self_attn_bwd_kernel, any length) took 75 us per layer of 256 sequences x 12 heads.
template <bool CAUSAL>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void train_self_attn_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dO,
                                                                        const int32_t* __restrict__ mask,
                                                                        const float* __restrict__ rel_bias,
                                                                        const int32_t* __restrict__ bucket, float* __restrict__ dqkv,
                                                                        float* __restrict__ dbias_part, int S, int Ls, int H, int buckets) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WAVE_FLOATS = 3 * 32 * 64 + 64 + 96 + 64 + 64;
  const int inner = H * DKV, ld = 3 * inner;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, n = lane & 31;
  const int w = blockIdx.x * 2 + wave;
  const int seq = w / H, h = w - seq * H;
  if (seq >= S) return;                                 // wave-uniform
  float* Ks = smem + (size_t)wave * WAVE_FLOATS;
  float* Qs = Ks + 32 * 64;
  float* Ds = Qs + 32 * 64;
  float* Bs = Ds + 32 * 64;          // [64] bias per key offset
  float* st = Bs + 64;               // [3][32] per query: row maximum, 1 / row sum, rowsum(dP * P)
  float* diag = st + 96;             // [64] sum of dS along each diagonal
  int* bk = reinterpret_cast<int*>(diag + 64);   // [64] bucket of each diagonal
  const float* base = qkv + (size_t)seq * Ls * ld + h * DKV;
  const float* dob = dO + (size_t)seq * Ls * inner + h * DKV;
  {  // K, Q, dO rows -> LDS, four coalesced 256-B rows per instruction; rows past Ls are zero
    const int g = lane >> 4, li = lane & 15;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int j = it * 4 + g;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), qv = kv, dv = kv;
      if (j < Ls) {
        kv = *reinterpret_cast<const float4*>(base + (size_t)j * ld + inner + li * 4);
        qv = *reinterpret_cast<const float4*>(base + (size_t)j * ld + li * 4);
        dv = *reinterpret_cast<const float4*>(dob + (size_t)j * inner + li * 4);
      }
      *reinterpret_cast<float4*>(Ks + j * 64 + li * 4) = kv;
      *reinterpret_cast<float4*>(Qs + j * 64 + li * 4) = qv;
      *reinterpret_cast<float4*>(Ds + j * 64 + li * 4) = dv;
    }
  }
  // diagonal t: causal i - j = t; bidirectional j - i = t - (Ls - 1)
  const int nd = CAUSAL ? Ls : 2 * Ls - 1;
  if (lane < nd) {
    const int b = bucket[CAUSAL ? lane : lane - (Ls - 1) + (MAX_LQ - 1)];
    bk[lane] = b;
    Bs[lane] = rel_bias[b * H + h];
  }
  const bool kok = n < Ls && (CAUSAL || mask[(size_t)seq * Ls + n] != 0);
  // row pieces of the score products: V from global memory; K, Q, dO from their LDS strips (rows past Ls are zero there).
  // Four sets of 32-byte pieces from global memory were 1024 cache-line requests per wave (32 rows per instruction) and made
  // this kernel as slow as the VALU one (76 us); the strip reads all fall on the same banks (row stride 256 B) and still
  // cost only ~64 cycles each.
  float4 kreg[8], qreg[8], vreg[8], greg[8];
  load_row_pieces(n < Ls ? base + (size_t)n * ld + 2 * inner : nullptr, half, vreg);
  __builtin_amdgcn_wave_barrier();
  load_row_pieces(Ks + n * 64, half, kreg);
  load_row_pieces(Qs + n * 64, half, qreg);
  load_row_pieces(Ds + n * 64, half, greg);
  const unsigned long long okm = __ballot(kok) & 0xffffffffull;   // bit j: key j is attended
  const int iq = n < Ls ? n : Ls - 1;                    // rows past Ls run on clamped indices and are never stored

  // ---- a lane per query: P^T, dS^T (registers = keys kappa(r, half)) ----------------------------------------------
  f32x16 p1[1], ds1[1];
#pragma unroll
  for (int r = 0; r < 16; ++r) { p1[0][r] = 0.f; ds1[0][r] = 0.f; }
  mfma_scores(kreg, qreg, p1[0]);                        // S^T[key][query]
  // the two products of the second layout (a lane per key) are issued here as well: the row pieces die early
  f32x16 p2[1], ds2[1];
#pragma unroll
  for (int r = 0; r < 16; ++r) { p2[0][r] = 0.f; ds2[0][r] = 0.f; }
  mfma_scores(qreg, kreg, p2[0]);                        // S[query][key]
  mfma_scores(vreg, greg, ds1[0]);                       // dP^T[key][query]
  mfma_scores(greg, vreg, ds2[0]);                       // dP[query][key]
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = kappa(r, half);
    const bool ok = ((okm >> j) & 1ull) && (!CAUSAL || j <= iq);
    const int bi = CAUSAL ? iq - j : j - iq + Ls - 1;
    p1[0][r] = ok ? p1[0][r] + Bs[ok ? bi : 0] : -INFINITY;
    mx = fmaxf(mx, p1[0][r]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float e = (p1[0][r] == -INFINITY) ? 0.f : expf(p1[0][r] - mx);
    p1[0][r] = e;
    sum += e;
  }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;
  float cq = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) { p1[0][r] *= inv; cq = fmaf(ds1[0][r], p1[0][r], cq); }
  cq += __shfl_xor(cq, 32, 64);
#pragma unroll
  for (int r = 0; r < 16; ++r) ds1[0][r] = p1[0][r] * (ds1[0][r] - cq);
  if (half == 0) { st[n] = mx; st[32 + n] = inv; st[64 + n] = cq; }
  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  mfma_pv<1>(ds1, Ks, lane, o);                          // dQ[query][d] = sum_key dS[query][key] K[key][d]
  store_o_tile(o, Ks, lane, 0, Ls, (size_t)seq * Ls, ld, h * DKV, dqkv, nullptr, 0, nullptr);
  // dS[i][j] -> the (now free) K strip for the diagonal sums
#pragma unroll
  for (int r = 0; r < 16; ++r) Ks[n * 33 + kappa(r, half)] = ds1[0][r];

  // ---- a lane per key: P, dS (registers = queries kappa(r, half)) ---------------------------------------------------
  __builtin_amdgcn_wave_barrier();                       // st[] written above
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = kappa(r, half);
    const bool ok = kok && i < Ls && (!CAUSAL || n <= i);
    const int bi = CAUSAL ? i - n : n - i + Ls - 1;
    const float pe = ok ? expf(p2[0][r] + Bs[ok ? bi : 0] - st[i]) * st[32 + i] : 0.f;
    p2[0][r] = pe;
    ds2[0][r] = ok ? pe * (ds2[0][r] - st[64 + i]) : 0.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  mfma_pv<1>(ds2, Qs, lane, o);                          // dK[key][d] = sum_query dS[query][key] Q[query][d]
  store_o_tile(o, Qs, lane, 0, Ls, (size_t)seq * Ls, ld, inner + h * DKV, dqkv, nullptr, 0, nullptr);
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  mfma_pv<1>(p2, Ds, lane, o);                           // dV[key][d] = sum_query P[query][key] dO[query][d]
  store_o_tile(o, Ds, lane, 0, Ls, (size_t)seq * Ls, ld, 2 * inner + h * DKV, dqkv, nullptr, 0, nullptr);

  // ---- bias gradient of this (sequence, head): every diagonal of dS in row order, then the diagonals of a bucket in order
  // (the order of self_attn_bwd_kernel)
  if (lane < nd) {
    const int off = CAUSAL ? -lane : lane - (Ls - 1);    // j - i
    float acc = 0.f;
    for (int i = 0; i < Ls; ++i) { const int j = i + off; if (j >= 0 && j < Ls) acc += Ks[i * 33 + j]; }
    diag[lane] = acc;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < buckets) {
    float acc = 0.f;
    for (int t = 0; t < nd; ++t) if (bk[t] == lane) acc += diag[t];
    dbias_part[((size_t)seq * H + h) * buckets + lane] = acc;
  }
}

hipError_t launch_train_self_attn_bwd_mfma(const float* qkv, const float* dO, const int32_t* mask, const float* rel_bias,
                                           const int32_t* bucket, float* dqkv, float* dbias_part, int S, int Ls, int H, int buckets,
                                           int causal, hipStream_t s) {
  if (Ls > 32 || Ls < 1 || buckets > 64) return hipErrorInvalidValue;
  const long waves = (long)S * H;
  const dim3 grid((unsigned)((waves + 1) / 2)), blk(128);
  const size_t smem = 2 * (3 * 32 * 64 + 64 + 96 + 64 + 64) * sizeof(float);
  if (causal) hipLaunchKernelGGL(train_self_attn_bwd_mfma_kernel<true>, grid, blk, smem, s, qkv, dO, mask, rel_bias, bucket, dqkv,
                                 dbias_part, S, Ls, H, buckets);
  else hipLaunchKernelGGL(train_self_attn_bwd_mfma_kernel<false>, grid, blk, smem, s, qkv, dO, mask, rel_bias, bucket, dqkv, dbias_part,
                          S, Ls, H, buckets);
  return hipGetLastError();
}

static size_t tail_self_attn_smem(int L, int D = 64) { return ((size_t)L * (D + 1) + (size_t)L * D + 4 * 64 + 64) * sizeof(float); }

hipError_t launch_tail_self_attn(const TailSelfAttnArgs& a, hipStream_t s) {
  if (a.L > MAX_DEC_LEN || a.T < 1 || a.T >= a.L) return hipErrorInvalidValue;
  if (a.dkv == 128) {   // t5-3b heads: the VALU kernel (the MFMA tiles below are written for 64-dim heads)
    hipLaunchKernelGGL(tail_self_attn_kernel<128>, dim3((unsigned)a.nseq_cap * a.H), dim3(256), tail_self_attn_smem(a.L, 128), s, a);
    return hipGetLastError();
  }
  static const bool off = [] { const char* e = dev_getenv("RPR_TAIL_ATTN_MFMA"); return e && atoi(e) == 0; }();
  if (!off) {
    const long waves = (long)a.nseq_cap * a.H;
    const dim3 grid((unsigned)((waves + 3) / 4)), blk(256);
    static const int occ = [] { const char* e = dev_getenv("RPR_TAIL_ATTN_OCC"); return e ? atoi(e) : 2; }();
    if (g_tail_attn_gen == 2 && a.L <= 32 && a.T <= 8) {
      const int HB = (a.H + 3) / 4;
      const long blocks = (long)a.nseq_cap * HB;
      if (blocks < (1l << 31) / HB && (long)a.nseq_cap < (1l << 32) / a.B) {   // udiv_magic's exact range
        const dim3 g2((unsigned)blocks);
        const unsigned hm = div_magic(HB), bm = div_magic(a.B);
        if (g_tail_attn_opt & 1) hipLaunchKernelGGL(tail_self_attn_mfma_v2_kernel<3>, g2, blk, 4 * (32 * 64) * sizeof(float), s, a, HB, hm, bm);
        else hipLaunchKernelGGL(tail_self_attn_mfma_v2_kernel<4>, g2, blk, 4 * (32 * 64) * sizeof(float), s, a, HB, hm, bm);
        return hipGetLastError();
      }
    }
    if (a.L <= 32 && occ == 3) hipLaunchKernelGGL((tail_self_attn_mfma_kernel<1, 3>), grid, blk, 4 * (32 * 64 + 64) * sizeof(float), s, a);
    else if (a.L <= 32) hipLaunchKernelGGL(tail_self_attn_mfma_kernel<1>, grid, blk, 4 * (32 * 64 + 64) * sizeof(float), s, a);
    else hipLaunchKernelGGL(tail_self_attn_mfma_kernel<2>, grid, blk, 4 * (64 * 64 + 64 + 32 * 64) * sizeof(float), s, a);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(tail_self_attn_kernel<64>, dim3((unsigned)a.nseq_cap * a.H), dim3(256), tail_self_attn_smem(a.L), s, a);
  return hipGetLastError();
}

// Gold-code score of a tail row: final RMSNorm of the row's stream (times d_model^-0.5 under scaleup_output_hidden)
// dotted with the OUTPUT codebook row of the token at that position = the logit the sequential step's GEMM would give
// the beam's only valid child (reference get_lm_logits, t5_generative_retriever.py:250-262), in exact fp32.
__global__ __launch_bounds__(256) void tail_gold_kernel(const float* __restrict__ x, const float* __restrict__ ln,
                                                         const float* __restrict__ out_embeds, const uint16_t* __restrict__ tokens,
                                                         float* __restrict__ gold, int rows, const int* __restrict__ rows_dev, int T,
                                                         int L, int d, int V, float eps, float post, const __half* __restrict__ x_h,
                                                         size_t x_ps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows || row >= *rows_dev) return;
  const int Lt = L - T, seq = row / Lt, p = T + (row - seq * Lt);
  const int tok = tokens[(size_t)seq * L + p];
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * d);   // only dereferenced when x_h == nullptr
  const float4* wr = reinterpret_cast<const float4*>(ln);
  const float4* er = reinterpret_cast<const float4*>(out_embeds + ((size_t)p * V + tok) * d);
  const int n4 = d >> 2;
  auto load4 = [&](int k) -> float4 {
    if (!x_h) return xr[k];
    const size_t idx = (size_t)row * d + 4 * (size_t)k;
    const uint2 hh = *reinterpret_cast<const uint2*>(x_h + idx), ll = *reinterpret_cast<const uint2*>(x_h + x_ps + idx);
    const __half* h = reinterpret_cast<const __half*>(&hh); const __half* l = reinterpret_cast<const __half*>(&ll);
    return make_float4(x_from_planes(h[0], l[0]), x_from_planes(h[1], l[1]), x_from_planes(h[2], l[2]), x_from_planes(h[3], l[3]));
  };
  float ss = 0.f;
  for (int k = lane; k < n4; k += 64) {
    const float4 v = load4(k);
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)d + eps);
  float acc = 0.f;
  for (int k = lane; k < n4; k += 64) {
    const float4 v = load4(k), g = wr[k], e = er[k];
    float4 hd = make_float4(g.x * (v.x * rs), g.y * (v.y * rs), g.z * (v.z * rs), g.w * (v.w * rs));
    if (post != 1.0f) { hd.x *= post; hd.y *= post; hd.z *= post; hd.w *= post; }
    acc += (hd.x * e.x + hd.y * e.y) + (hd.z * e.z + hd.w * e.w);
  }
  acc = wave_sum(acc);
  if (lane == 0) gold[row] = acc;
}

hipError_t launch_tail_gold(const float* x, const float* ln, const float* out_embeds, const uint16_t* tokens, float* gold, int rows,
                            const int* rows_dev, int T, int L, int d, int V, float eps, float post, hipStream_t s, const __half* x_h,
                            size_t x_ps) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(tail_gold_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ln, out_embeds, tokens, gold, rows, rows_dev, T, L, d, V,
                     eps, post, x_h, x_ps);
  return hipGetLastError();
}

// RPR_FLAG_LOG_SOFTMAX: the score of a position is the log-probability of its token (reference generation.py:453-455:
// log_softmax over the V logits of the position in fp32). One wave per tail row: logits = the row's V exact-fp32 logits
// (one GEMM per position, api.hip::enqueue_tail), arithmetic as select_kernel's: (x - max) - log(sum exp(x - max)).
__global__ __launch_bounds__(256) void tail_logprob_kernel(const float* __restrict__ logits, const uint16_t* __restrict__ tokens,
                                                            float* __restrict__ gold, int rows, const int* __restrict__ rows_dev, int T,
                                                            int L, int V) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows || row >= *rows_dev) return;
  const int Lt = L - T, seq = row / Lt, p = T + (row - seq * Lt);
  const int tok = tokens[(size_t)seq * L + p];
  const float* lr = logits + (size_t)row * V;
  float mx = -INFINITY;
  for (int c = lane; c < V; c += 64) mx = fmaxf(mx, lr[c]);
  mx = wave_max(mx);
  float sm = 0.f;
  for (int c = lane; c < V; c += 64) sm += expf(lr[c] - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);      // the reduction order of select_kernel
  if (lane == 0) gold[row] = (lr[tok] - mx) - logf(sm);
}

hipError_t launch_tail_logprob(const float* logits, const uint16_t* tokens, float* gold, int rows, const int* rows_dev, int T, int L,
                               int V, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(tail_logprob_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, logits, tokens, gold, rows, rows_dev, T, L, V);
  return hipGetLastError();
}

// One block per forced query: replay of the remaining L - T selection steps and the finalize step on the B forced
// candidates. Per step the B winners are the beams' single valid children; new slot order = (cumulative score desc,
// parent slot asc) — the sort order of the sequential select_kernel restricted to those candidates. Then
// finalize_kernel's rule: rank by float64 sum/(L+1) desc, exact ties in reverse slot order, float32 store.
// The slot order of the intermediate steps only ever decides exact ties, so the kernel first sums the scores (same
// additions in the same order), ranks the final values once and replays the L - T steps only if two of them are equal
// (B = 1000: the replay was 28 x B^2 comparisons = 9 ms for one query; the single ranking pass is 30 us).
__global__ __launch_bounds__(1024) void tail_rank_kernel(TailRankArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int i = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  if (i >= *a.nf_dev) return;
  const int B = a.B, T = a.T, L = a.L, Lt = L - T;
  double* S = reinterpret_cast<double*>(smem_raw);       // [B] cumulative score of the beam that started in slot b
  int* pos = reinterpret_cast<int*>(S + B);              // [B] its current slot
  int* npos = pos + B;                                   // [B]
  __shared__ int tie;
  const int q = a.flist[i];
  const size_t r0 = (size_t)q * B;
  if (tid == 0) tie = a.replay;
  for (int b = tid; b < B; b += nt) {
    double s = a.st.score[r0 + b];
    const float* g = a.gold + ((size_t)i * B + b) * Lt;
    for (int t = 0; t < Lt; ++t) s = ((double)g[t] + 0.0) + s;
    S[b] = s / (double)(L + 1);
    pos[b] = b;
  }
  __syncthreads();
  for (int b = tid; b < B; b += nt) {
    const double s = S[b];
    int gt = 0, eq = 0;
    for (int k = 0; k < B; ++k) { const double o = S[k]; gt += o > s; eq += o == s; }
    npos[b] = gt;
    if (eq > 1) tie = 1;                                 // benign race: every writer stores 1
  }
  __syncthreads();
  if (tie) {                                             // block-uniform: exact ties -> the full replay decides them
    for (int b = tid; b < B; b += nt) S[b] = a.st.score[r0 + b];
    __syncthreads();
    for (int t = 0; t < Lt; ++t) {
      for (int b = tid; b < B; b += nt) S[b] = ((double)a.gold[((size_t)i * B + b) * Lt + t] + 0.0) + S[b];
      __syncthreads();
      for (int b = tid; b < B; b += nt) {
        const double s = S[b];
        const int pb = pos[b];
        int rk = 0;
        for (int k = 0; k < B; ++k) rk += (S[k] > s) || (S[k] == s && pos[k] < pb);
        npos[b] = rk;
      }
      __syncthreads();
      for (int b = tid; b < B; b += nt) pos[b] = npos[b];
      __syncthreads();
    }
    for (int b = tid; b < B; b += nt) S[b] = S[b] / (double)(L + 1);
    __syncthreads();
    for (int b = tid; b < B; b += nt) {
      const double s = S[b];
      const int pb = pos[b];
      int rk = 0;
      for (int k = 0; k < B; ++k) rk += (S[k] > s) || (S[k] == s && pos[k] > pb);
      npos[b] = rk;
    }
    __syncthreads();
  }
  const size_t o0 = (size_t)a.qmap[i] * B;
  for (int b = tid; b < B; b += nt) {
    const int rk = npos[b];
    a.out_scores[o0 + rk] = (float)S[b];
    a.out_lo[o0 + rk] = a.st.lo[r0 + b];
    a.out_hi[o0 + rk] = a.st.hi[r0 + b];
  }
  for (int k = tid; k < B * L; k += nt) {
    const int b = k / L, p = k - b * L;
    a.out_tokens[(o0 + npos[b]) * L + p] = (int32_t)a.tokens[((size_t)i * B + b) * L + p];
  }
}

hipError_t launch_tail_rank(const TailRankArgs& a_in, hipStream_t s) {
  const char* rp = getenv("RPR_TAIL_RANK_REPLAY");          // tests: always take the tie path (read per call)
  const int replay = rp ? atoi(rp) : 0;
  TailRankArgs a = a_in;
  if (replay) a.replay = 1;
  const size_t smem = (size_t)a.B * (sizeof(double) + 2 * sizeof(int)) + 16;
  hipLaunchKernelGGL(tail_rank_kernel, dim3(a.Qcap), dim3(a.B > 256 ? 1024 : 256), smem, s, a);
  return hipGetLastError();
}

__global__ void flag_nonzero_kernel(const int* __restrict__ cnt, unsigned int* __restrict__ flag) {
  if (threadIdx.x == 0 && *cnt != 0) *flag = 1u;
}

hipError_t launch_flag_nonzero(const int* cnt, unsigned int* flag, hipStream_t s) {
  hipLaunchKernelGGL(flag_nonzero_kernel, dim3(1), dim3(64), 0, s, cnt, flag);
  return hipGetLastError();
}

// max over rows of || E[r] (*) w ||_2, as the bit pattern of a non-negative float through atomicMax (out zeroed)
__global__ __launch_bounds__(256) void max_row_norm_kernel(const float* __restrict__ E, const float* __restrict__ w, int rows, int d,
                                                            float* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float ss = 0.f;
  for (int k = lane; k < d; k += 64) {
    const float v = E[(size_t)row * d + k] * (w ? w[k] : 1.0f);
    ss = fmaf(v, v, ss);
  }
  ss = wave_sum(ss);
  if (lane == 0) {
    float n = sqrtf(ss);
    if (!(n >= 0.f)) n = INFINITY;   // NaN: no bound
    atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(n));
  }
}

hipError_t launch_max_row_norm(const float* E, const float* w, int rows, int d, float* out, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(max_row_norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, E, w, rows, d, out);
  return hipGetLastError();
}

hipError_t init_tail_kernel_attributes() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tail_self_attn_kernel<64>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(tail_self_attn_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)tail_self_attn_smem(MAX_DEC_LEN, 128));
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(tail_self_attn_mfma_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(tail_cross_attn_mfma_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(tail_rank_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
}

}  // namespace rpr
