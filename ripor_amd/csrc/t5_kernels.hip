// T5 elementwise + attention kernels of the search path (gfx950, wave64).
//
// Arithmetic follows HF T5Stack as the reference exercises it (reference
// t5_pretrainer/modeling/t5_generative_retriever.py:358-366 encoder, :403-416 decoder):
//   RMSNorm without mean/bias; attention scores NOT scaled by 1/sqrt(dkv); bucketed relative
//   position bias from block 0 shared by all blocks; additive masks that make masked keys
//   contribute exactly 0 after the fp32 softmax (here: masked keys are skipped); softmax
//   normalised before the PV product.
// All of these are HBM/L2-bound row kernels: 16-byte per-lane loads, one wave per row / per
// (beam, head), cross-lane reductions with DPP/shuffles, no LDS staging of data that is read once.
#include <hip/hip_fp16.h>

#include <algorithm>

#include "common.h"
#include "kernel_utils.h"

namespace rpr {

// ------------------------------------------------------------------------------------ RMSNorm
// one wave per row; d % 4 == 0
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       float* __restrict__ out, int rows, int d, float eps,
                                                       float post_scale, __half* __restrict__ out_h, size_t o_ps,
                                                       const int* __restrict__ rows_dev, unsigned int* sat,
                                                       const __half* __restrict__ x_h, size_t x_ps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows || (rows_dev && row >= *rows_dev)) return;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * d);   // only dereferenced when x_h == nullptr
  const float4* wr = reinterpret_cast<const float4*>(w);
  float4* orow = reinterpret_cast<float4*>(out + (size_t)row * d);  // only dereferenced when out != nullptr
  const int n4 = d >> 2;
  auto load4 = [&](int i) -> float4 {
    if (!x_h) return xr[i];
    const size_t idx = (size_t)row * d + 4 * (size_t)i;      // residual stream kept in f16 planes (split-precision mode)
    const uint2 hh = *reinterpret_cast<const uint2*>(x_h + idx), ll = *reinterpret_cast<const uint2*>(x_h + x_ps + idx);
    const __half* h = reinterpret_cast<const __half*>(&hh); const __half* l = reinterpret_cast<const __half*>(&ll);
    return make_float4(x_from_planes(h[0], l[0]), x_from_planes(h[1], l[1]), x_from_planes(h[2], l[2]), x_from_planes(h[3], l[3]));
  };
  float ss = 0.f;
  for (int i = lane; i < n4; i += 64) {
    const float4 v = load4(i);
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)d + eps);
  for (int i = lane; i < n4; i += 64) {
    const float4 v = load4(i), g = wr[i];
    float4 o = make_float4(g.x * (v.x * rs), g.y * (v.y * rs), g.z * (v.z * rs), g.w * (v.w * rs));
    if (post_scale != 1.0f) { o.x *= post_scale; o.y *= post_scale; o.z *= post_scale; o.w *= post_scale; }
    if (out) orow[i] = o;
    if (out_h) store_planes4(out_h, o_ps, (size_t)row * d + 4 * (size_t)i, o, sat);
  }
}

hipError_t launch_rmsnorm(const float* x, const float* w, float* out, int rows, int d, float eps, hipStream_t s,
                          float post_scale, __half* out_h, size_t o_ps, const int* rows_dev, unsigned int* sat,
                          const __half* x_h, size_t x_ps) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(rmsnorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, out, rows, d, eps, post_scale, out_h, o_ps,
                     rows_dev, sat, x_h, x_ps);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------ embeddings
__global__ __launch_bounds__(256) void embed_rows_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                                          float* __restrict__ out, int rows, int d, int vocab,
                                                          const int32_t* __restrict__ row_src,
                                                          const int* __restrict__ rows_dev, XOut xo) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows || (rows_dev && row >= *rows_dev)) return;
  int id = ids[row_src ? row_src[row] : row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  copy_row_x(reinterpret_cast<const float4*>(table + (size_t)id * d), out, row, d, lane, xo);
}

hipError_t launch_embed_rows(const float* table, const int32_t* ids, float* out, int rows, int d, int vocab,
                             hipStream_t s, const int32_t* row_src, const int* rows_dev, XOut xo) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(embed_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, table, ids, out, rows, d, vocab, row_src,
                     rows_dev, xo);
  return hipGetLastError();
}

// Packed encoder: exclusive scan of the per-query lengths (one block; Q is at most a few thousand per call,
// the chunk loop covers any Q) and the packed-row -> (q, j) source index.
__global__ __launch_bounds__(1024) void pack_scan_kernel(const int32_t* __restrict__ lens, int32_t* __restrict__ offs, int Q) {
  __shared__ int part[1024];
  __shared__ int carry;
  const int tid = threadIdx.x;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int q0 = 0; q0 < Q; q0 += 1024) {
    const int q = q0 + tid;
    const int v = q < Q ? lens[q] : 0;
    part[tid] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {          // Hillis-Steele inclusive scan
      const int add = tid >= o ? part[tid - o] : 0;
      __syncthreads();
      part[tid] += add;
      __syncthreads();
    }
    if (q < Q) offs[q] = carry + part[tid] - v;
    __syncthreads();
    if (tid == 1023) carry += part[1023];
    __syncthreads();
  }
  if (tid == 0) offs[Q] = carry;
}

__global__ __launch_bounds__(256) void pack_fill_kernel(const int32_t* __restrict__ lens, const int32_t* __restrict__ offs,
                                                         int32_t* __restrict__ row_src, int Q, int Lq) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= Q) return;
  const int n = lens[q], o = offs[q];
  for (int j = lane; j < n; j += 64) row_src[o + j] = q * Lq + j;
}

hipError_t launch_pack_rows(const int32_t* lens, int32_t* offs, int32_t* row_src, int Q, int Lq, hipStream_t s) {
  hipLaunchKernelGGL(pack_scan_kernel, dim3(1), dim3(1024), 0, s, lens, offs, Q);
  hipLaunchKernelGGL(pack_fill_kernel, dim3((Q + 3) / 4), dim3(256), 0, s, lens, offs, row_src, Q, Lq);
  return hipGetLastError();
}

// decoder input embedding of position t (reference t5_generative_retriever.py:194-214):
// position 0 is the constant start_token_embed, position t>=1 is list_decoder_embeds[t-1][token_t].
__global__ __launch_bounds__(256) void dec_embed_kernel(const float* __restrict__ start, const float* __restrict__ in_embeds,
                                                         const uint16_t* __restrict__ tokens, int tok_ld,
                                                         float* __restrict__ out, int R, int d, int V, int t, XOut xo,
                                                         const int* __restrict__ rows_dev) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R || (rows_dev && row >= *rows_dev)) return;
  const float* src = start;
  if (t > 0) {
    const int tok = tokens[(size_t)row * tok_ld + (t - 1)];
    src = in_embeds + ((size_t)(t - 1) * V + tok) * d;
  }
  copy_row_x(reinterpret_cast<const float4*>(src), out, row, d, lane, xo);
}

hipError_t launch_dec_embed(const float* start, const float* in_embeds, const uint16_t* tokens, int tok_ld,
                            float* out, int R, int d, int V, int t, hipStream_t s, XOut xo, const int* rows_dev) {
  hipLaunchKernelGGL(dec_embed_kernel, dim3((R + 3) / 4), dim3(256), 0, s, start, in_embeds, tokens, tok_ld, out,
                     R, d, V, t, xo, rows_dev);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------ encoder self-attention
// One block per (query, head). K and V of the head are staged once in LDS ([Lq][65] padded), every
// wave then handles query rows i = wave, wave+4, ...: lane j scores keys j, j+64, ...; the q row is
// broadcast through SGPRs (v_readlane); softmax across the wave; PV with lane = output dim.
// Nothing inside the row loop reads global memory: the relative-position bias of every key offset (bucket table + bias
// table, two dependent loads), the key mask and the q rows (eight rows of the wave per batch, requested together) are
// staged in LDS first. With those three loads in the loop a row cost two memory round trips: 61 us for the 3072 blocks of
// a teacher-forced decoder layer (32 positions), 290 us for 2150 packed queries of the search encoder.
// VG (128-dim heads, queries of more than ~150 tokens: K and V of a head together exceed the 160 KB of LDS): only K is staged,
// the weighted sum reads the V rows from global memory (512 contiguous bytes per row and wave, requested eight rows at a time;
// the rows of a query stay in the XCD's L2 between its heads' blocks).
template <int D, bool VG = false>   // head dim: 64 (every kernel argument / LDS row as before) or 128 (t5-3b): a lane holds D / 64 dims of a row
__global__ __launch_bounds__(256) void enc_attn_kernel(EncAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NV = D / 64;
  const int Lq = a.Lq, H = a.H, inner = H * D, ld = 3 * inner;
  const int qi = blockIdx.x / H, h = blockIdx.x - qi * H;
  float* Ks = smem;                         // [Lq][D + 1]
  float* Vs = smem + (size_t)Lq * (D + 1);  // [Lq][D] (VG: not staged)
  float* Ps = Vs + (VG ? 0 : (size_t)Lq * D);   // [4][Lq] normalised weights per wave
  float* RB = Ps + 4 * Lq;                  // [2 Lq] bias of this head per key offset (causal: i - j; else j - i + nrow - 1)
  float* Qs = RB + 2 * Lq;                  // [4][8][D] q rows of the wave's current batch
  int* Ms = reinterpret_cast<int*>(Qs + 4 * 8 * D);   // [Lq] key mask
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // packed encoder: the query's rows start at offs[qi] and only its own lens[qi] positions exist (the padded
  // positions behind them are masked keys and unused query rows in the padded layout)
  const int nrow = a.offs ? a.lens[qi] : Lq;
  const size_t row0 = a.offs ? (size_t)a.offs[qi] : (size_t)qi * Lq;
  const float* base = a.qkv + row0 * ld + h * D;
  for (int i = tid; i < nrow * (D / 4); i += 256) {
    const int j = i / (D / 4), c = (i - j * (D / 4)) * 4;
    const float4 kv = *reinterpret_cast<const float4*>(base + (size_t)j * ld + inner + c);
    const float4 vv = *reinterpret_cast<const float4*>(base + (size_t)j * ld + 2 * inner + c);
    float* kd = Ks + j * (D + 1) + c;
    kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
    if (!VG) *reinterpret_cast<float4*>(Vs + j * D + c) = vv;
  }
  const int nrb = a.causal ? nrow : 2 * nrow - 1, rb0 = a.causal ? 0 : MAX_LQ - nrow;
  for (int k = tid; k < nrb; k += 256) RB[k] = a.rel_bias[a.bucket[rb0 + k] * H + h];
  if (!a.causal) {
    const int32_t* mrow = a.mask + (size_t)qi * Lq;
    for (int j = tid; j < nrow; j += 256) Ms[j] = mrow[j];
  }
  __syncthreads();
  const int nchunk = (nrow + 63) >> 6;
  float* P = Ps + wave * Lq;
  float* Qw = Qs + wave * 8 * D;
  for (int i0 = wave; i0 < nrow; i0 += 32) {
    {   // the wave's next eight q rows: all requests go out before the first one is used
      float qb[8][NV];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 4 * u;
#pragma unroll
        for (int v = 0; v < NV; ++v) qb[u][v] = i < nrow ? base[(size_t)i * ld + 64 * v + lane] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v) Qw[u * D + 64 * v + lane] = qb[u][v];
    }
    __builtin_amdgcn_wave_barrier();
  for (int u = 0; u < 8; ++u) {
    const int i = i0 + 4 * u;
    if (i >= nrow) break;
    float qv[NV];                             // lane d holds q_i[d], q_i[64 + d]
#pragma unroll
    for (int v = 0; v < NV; ++v) qv[v] = Qw[u * D + 64 * v + lane];
    float sc[MAX_LQ / 64];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAX_LQ / 64; ++c) {
      if (c >= nchunk) break;
      const int j = c * 64 + lane;
      const int jc = j < nrow ? j : nrow - 1;
      const float* kr = Ks + jc * (D + 1);
      float acc = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int d = 0; d < 64; ++d) {
          const float qd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qv[v]), d));
          acc = fmaf(qd, kr[64 * v + d], acc);
        }
      float s = -INFINITY;
      if (a.causal) { if (j <= i) s = acc + RB[i - j]; }   // teacher-forced decoder: rel = j - i <= 0, table[n = i - j]
      else if (j < nrow && Ms[j] != 0) s = acc + RB[j - i + nrow - 1];
      sc[c] = s;
      mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < MAX_LQ / 64; ++c) {
      if (c >= nchunk) break;
      const float e = (sc[c] == -INFINITY) ? 0.f : expf(sc[c] - mx);
      sc[c] = e;
      sum += e;
    }
    sum = wave_sum(sum);
#pragma unroll
    for (int c = 0; c < MAX_LQ / 64; ++c) {
      if (c >= nchunk) break;
      const int j = c * 64 + lane;
      if (j < nrow) P[j] = sc[c] / sum;
    }
    __builtin_amdgcn_wave_barrier();
    float o[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) o[v] = 0.f;
    if (VG) {
      const float* vg = base + 2 * inner + lane;
      for (int j0 = 0; j0 < nrow; j0 += 8) {
        float vv[8][NV];
#pragma unroll
        for (int u2 = 0; u2 < 8; ++u2) {
          const int j = min(j0 + u2, nrow - 1);           // clamped, unconditional: all eight requests go out together
#pragma unroll
          for (int v = 0; v < NV; ++v) vv[u2][v] = vg[(size_t)j * ld + 64 * v];
        }
#pragma unroll
        for (int u2 = 0; u2 < 8; ++u2) {
          if (j0 + u2 < nrow) {
#pragma unroll
            for (int v = 0; v < NV; ++v) o[v] = fmaf(P[j0 + u2], vv[u2][v], o[v]);
          }
        }
      }
    } else {
      for (int j = 0; j < nrow; ++j)
#pragma unroll
        for (int v = 0; v < NV; ++v) o[v] = fmaf(P[j], Vs[j * D + 64 * v + lane], o[v]);
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const size_t oidx = (row0 + i) * inner + h * D + 64 * v + lane;
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
}

static size_t enc_attn_smem(int Lq, int D, bool stage_v = true) {
  return ((size_t)Lq * (D + 1) + (stage_v ? (size_t)Lq * D : 0) + 4 * (size_t)Lq + 2 * (size_t)Lq + 4 * 8 * (size_t)D + (size_t)Lq) * sizeof(float);
}
hipError_t launch_enc_attn(const EncAttnArgs& a, hipStream_t s) {
  if (a.Lq > MAX_LQ || a.buckets > 64) return hipErrorInvalidValue;
  if (a.dkv == 128) {   // t5-3b heads: the generic kernel only
    const size_t smem = enc_attn_smem(a.Lq, 128);
    if (smem <= 160 * 1024) {                               // Lq <= ~150: K and V of a head in LDS
      hipLaunchKernelGGL(enc_attn_kernel<128>, dim3(a.Q * a.H), dim3(256), smem, s, a);
      return hipGetLastError();
    }
    const size_t smem_k = enc_attn_smem(a.Lq, 128, false);  // up to MAX_LQ = 256 tokens: K in LDS (155 KB), V from global memory
    if (smem_k > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL((enc_attn_kernel<128, true>), dim3(a.Q * a.H), dim3(256), smem_k, s, a);
    return hipGetLastError();
  }
  static const bool mfma_off = [] { const char* e = dev_getenv("RPR_TRAIN_ATTN_MFMA"); return e && atoi(e) == 0; }();
  if (a.mfma && !mfma_off && a.Lq <= 32 && !a.offs && !a.out_h) return launch_train_self_attn_mfma(a, s);
  if (!a.mfma) { hipError_t e; if (launch_enc_attn_mfma_v2(a, s, &e)) return e; }
  hipLaunchKernelGGL(enc_attn_kernel<64>, dim3(a.Q * a.H), dim3(256), enc_attn_smem(a.Lq, 64), s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------ decoder attention
// One wave per (beam row r, head h). The 64 lanes form 4 groups of 16; a group reads one K (or V)
// row of the head as 16 x float4 = one coalesced 256-byte segment, so each wave instruction
// fetches 4 key rows. Scores are reduced inside the 16-lane group, kept in a per-wave LDS strip,
// normalised, then the same mapping accumulates P.V and the 4 groups are summed with shuffles.
//
// Work item order: w = ((q * H + h) * B + b): the B beams of one (query, head) are adjacent and are
// remapped so that one XCD (one L2) gets a contiguous chunk of work items — beams of a query share
// most of their ancestors' K/V rows and the encoder K/V rows, which then hit in that XCD's L2.
template <bool SELF, int D = DKV>   // D = head dim: 64, or 128 (t5-3b: 32 lanes per row, two rows per wave instruction)
__global__ __launch_bounds__(256) void dec_attn_kernel(const float* __restrict__ qbuf, const float* __restrict__ kbase,
                                                        const float* __restrict__ vbase, const uint16_t* __restrict__ anc,
                                                        int anc_ld, const float* __restrict__ rel_bias,
                                                        const int32_t* __restrict__ bucket, const int32_t* __restrict__ mask,
                                                        float* __restrict__ out, int Q, int B, int H, int t, int Lq,
                                                        int xld, __half* __restrict__ out_h, size_t o_ps,
                                                        size_t q_stride, size_t h_stride, size_t pos_stride,
                                                        size_t slot_stride, unsigned int* sat, const int* __restrict__ nq_dev,
                                                        const int32_t* __restrict__ offs = nullptr) {
  constexpr int LPR = D / 4, GP = 64 / LPR;   // lanes per K / V row (float4 each), rows per wave instruction
  __shared__ float Ss[4][MAX_LQ];
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q8 = nblk >> 3, r8 = nblk & 7, x = bid & 7, k = bid >> 3;
    bid = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + k;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int w = bid * 4 + wave;
  const int R = Q * B, inner = H * D;
  if (w >= R * H) return;
  const int b = w % B, qh = w / B, h = qh % H, qi = qh / H;
  if (nq_dev && qi >= *nq_dev) return;
  const int r = qi * B + b;
  const int g = lane / LPR, li = lane % LPR;
  float* S = Ss[wave];
  const size_t xrow0 = (!SELF && offs) ? (size_t)offs[qi] : (size_t)qi * Lq;   // packed encoder rows of the query

  const float4 q4 = *reinterpret_cast<const float4*>(qbuf + (size_t)r * inner + h * D + li * 4);
  const int nkeys = SELF ? (t + 1) : Lq;
  const uint16_t* ancr = SELF ? (anc + (size_t)r * anc_ld) : nullptr;
  const int32_t* mrow = SELF ? nullptr : (mask + (size_t)qi * Lq);

  auto row_off = [&](int p) -> size_t {
    if (SELF) {
      const int slot = (p == t) ? b : (int)ancr[p];
      return (size_t)qi * q_stride + (size_t)h * h_stride + (size_t)p * pos_stride + (size_t)slot * slot_stride + li * 4;
    } else {
      return (xrow0 + p) * (size_t)xld + h * D + li * 4;
    }
  };

  float mx = -INFINITY;
  for (int p0 = 0; p0 < nkeys; p0 += GP) {
    const int p = p0 + g;
    float s = -INFINITY;
    bool ok = p < nkeys;
    if (!SELF && ok) ok = mrow[p] != 0;
    float d = 0.f;
    if (ok) {
      const float4 k4 = *reinterpret_cast<const float4*>(kbase + row_off(p));
      d = q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);   // (LPR = 16: the additions of group16_sum, in its order)
    if (ok) {
      s = d;
      if (SELF) s += rel_bias[bucket[t - p] * H + h];
    }
    if (li == 0 && p < nkeys) S[p] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  __builtin_amdgcn_wave_barrier();
  float sum = 0.f;
  for (int p = lane; p < nkeys; p += 64) {
    const float s = S[p];
    const float e = (s == -INFINITY) ? 0.f : expf(s - mx);
    S[p] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __builtin_amdgcn_wave_barrier();
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p0 = 0; p0 < nkeys; p0 += GP) {
    const int p = p0 + g;
    if (p < nkeys) {
      const float wgt = sum > 0.f ? S[p] / sum : 0.f;   // every key masked (an empty query): zeros, like dec_cross_attn_block_kernel
      if (wgt != 0.f) {
        const float4 v4 = *reinterpret_cast<const float4*>(vbase + row_off(p));
        acc.x = fmaf(wgt, v4.x, acc.x);
        acc.y = fmaf(wgt, v4.y, acc.y);
        acc.z = fmaf(wgt, v4.z, acc.z);
        acc.w = fmaf(wgt, v4.w, acc.w);
      }
    }
  }
#pragma unroll
  for (int o = LPR; o <= 32; o <<= 1) {
    acc.x += __shfl_xor(acc.x, o, 64);
    acc.y += __shfl_xor(acc.y, o, 64);
    acc.z += __shfl_xor(acc.z, o, 64);
    acc.w += __shfl_xor(acc.w, o, 64);
  }
  if (g == 0) {
    const size_t oidx = (size_t)r * inner + h * D + li * 4;
    if (out_h) store_planes4(out_h, o_ps, oidx, acc, sat);
    else *reinterpret_cast<float4*>(out + oidx) = acc;
  }
}

// Self-attention, fast path (t + 1 <= 4 * SELF_MAXIT keys): same work split as dec_attn_kernel<true>
// (one wave per (beam, head), 4 groups x 16 lanes x float4 = 4 coalesced 256-B rows per load
// instruction), but every K and V row of the beam's ancestry is requested up front into registers,
// so a wave keeps up to 2 * SELF_MAXIT KB in flight instead of one row group at a time — the
// per-wave-serialised version was latency-bound at ~3.9 TB/s algorithmic. No LDS.
// K/V rows are read exactly once per step and never reused: stream them past the caches (nt)
__device__ __forceinline__ float4 ld_stream(const float* p) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

constexpr int SELF_MAXIT_MAX = 9;  // 36 keys: covers L <= 35 (the reference uses L = 32 or 16)

// SELF_MAXIT = row groups of 4 keys held in registers. Early steps use the small instantiations: fewer
// VGPRs -> 8 waves per SIMD instead of 4, which is what hides the anc -> K/V dependent-load chain when a
// wave only has a few hundred bytes to fetch.
template <int SELF_MAXIT, int D = DKV>   // D = 128 (t5-3b): 32 lanes per row, SELF_MAXIT groups of TWO keys
__global__ __launch_bounds__(256) void dec_self_attn_fast_kernel(DecSelfAttnArgs a) {
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q8 = nblk >> 3, r8 = nblk & 7, x = bid & 7, k = bid >> 3;
    bid = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + k;
  }
  // the wave index through v_readfirstlane: beam, head, query and every base below are scalars (the divisions by B and H
  // were float-reciprocal sequences on the VALU per lane: a third of the kernel's instructions at t <= 7)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int B = a.B, H = a.H, t = a.t;
  const int w = bid * 4 + wave;
  constexpr int LPR = D / 4, GP = 64 / LPR;   // lanes per K / V row, rows per load instruction
  const int R = a.Q * B, inner = H * D;
  if (w >= R * H) return;
  const int qh = udiv_magic((unsigned)w, B, a.b_magic), b = w - qh * B;
  const int qi = udiv_magic((unsigned)qh, H, a.h_magic), h = qh - qi * H;
  if (a.nq_dev && qi >= *a.nq_dev) return;
  const int r = qi * B + b;
  const int g = lane / LPR, li = lane % LPR;
  const int nkeys = t + 1;
  const uint16_t* ancr = a.anc + (size_t)r * a.anc_ld;

  float4 kreg[SELF_MAXIT], vreg[SELF_MAXIT];
  // scalar base of the (query, head) region + a 32-bit offset per lane (the region is depth * B * 64 floats)
  const size_t hbase = (size_t)qi * a.q_stride + (size_t)h * a.h_stride;
  const float* kb = a.kcache + hbase;
  const float* vb = a.vcache + hbase;
  const int pstr = (int)a.pos_stride, sstr = (int)a.slot_stride;
  int off[SELF_MAXIT];
  // the ancestry entries of all row groups first, unconditionally (entry t of the row is inside the row and unused): behind the
  // `pc == t` test every load was retired before the next one was requested — SELF_MAXIT serial latencies in front of the K / V loads
  int av[SELF_MAXIT];
#pragma unroll
  for (int it = 0; it < SELF_MAXIT; ++it) {
    const int p = it * GP + g;
    av[it] = (int)ancr[p < nkeys ? p : t];
  }
#pragma unroll
  for (int it = 0; it < SELF_MAXIT; ++it) {
    const int p = it * GP + g;
    const int pc = p < nkeys ? p : t;
    const int slot = (pc == t) ? b : av[it];
    off[it] = pc * pstr + slot * sstr + li * 4;
  }
  const float4 q4 = *reinterpret_cast<const float4*>(a.q + ((size_t)r * inner + h * D) + li * 4);
#pragma unroll
  for (int it = 0; it < SELF_MAXIT; ++it)
    if (it * GP < nkeys) kreg[it] = ld_stream(kb + off[it]);
#pragma unroll
  for (int it = 0; it < SELF_MAXIT; ++it)
    if (it * GP < nkeys) vreg[it] = ld_stream(vb + off[it]);

  float sc[SELF_MAXIT];
  float mx = -INFINITY;
#pragma unroll
  for (int it = 0; it < SELF_MAXIT; ++it) {
    sc[it] = -INFINITY;
    if (it * GP < nkeys) {  // wave-uniform
      const int p = it * GP + g;
      float d = q4.x * kreg[it].x + q4.y * kreg[it].y + q4.z * kreg[it].z + q4.w * kreg[it].w;
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);   // D = 64: group16_sum's additions in its order
      if (p < nkeys) sc[it] = d + a.rel_bias[a.bucket[t - p] * H + h];
      mx = fmaxf(mx, sc[it]);
    }
  }
#pragma unroll
  for (int o = LPR; o <= 32; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < SELF_MAXIT; ++it) {
    sc[it] = exp_nonpos(sc[it] - mx);   // the bits of expf; a masked score is -inf: 0 (mx is finite: key t is always there)
    sum += sc[it];
  }
#pragma unroll
  for (int o = LPR; o <= 32; o <<= 1) sum += __shfl_xor(sum, o, 64);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int it = 0; it < SELF_MAXIT; ++it) {
    if (it * GP < nkeys) {
      const float wgt = sc[it] / sum;
      if (wgt != 0.f) {  // lanes past nkeys hold a clamped duplicate row with weight 0
        acc.x = fmaf(wgt, vreg[it].x, acc.x);
        acc.y = fmaf(wgt, vreg[it].y, acc.y);
        acc.z = fmaf(wgt, vreg[it].z, acc.z);
        acc.w = fmaf(wgt, vreg[it].w, acc.w);
      }
    }
  }
#pragma unroll
  for (int o = LPR; o <= 32; o <<= 1) {
    acc.x += __shfl_xor(acc.x, o, 64);
    acc.y += __shfl_xor(acc.y, o, 64);
    acc.z += __shfl_xor(acc.z, o, 64);
    acc.w += __shfl_xor(acc.w, o, 64);
  }
  if (g == 0) {
    const size_t oidx = (size_t)r * inner + h * D + li * 4;
    if (a.out_h) store_planes4(a.out_h, a.o_ps, oidx, acc, a.sat);
    else *reinterpret_cast<float4*>(a.out + oidx) = acc;
  }
}

hipError_t launch_dec_self_attn(const DecSelfAttnArgs& a_in, hipStream_t s) {
  DecSelfAttnArgs a = a_in;
  a.b_magic = div_magic(a.B); a.h_magic = div_magic(a.H);
  const int items = a.Q * a.B * a.H;
  if ((long)items >= (1l << 32) / std::max(a.B, a.H)) return hipErrorInvalidValue;   // udiv_magic's exact range
  const dim3 grid((items + 3) / 4), blk(256);
  if (a.dkv == 128) {   // t5-3b heads: two keys per register group; beyond 36 keys the generic one-wave-per-(beam, head) kernel
    const int nk = a.t + 1;
    if (nk <= 8) { hipLaunchKernelGGL((dec_self_attn_fast_kernel<4, 128>), grid, blk, 0, s, a); return hipGetLastError(); }
    if (nk <= 16) { hipLaunchKernelGGL((dec_self_attn_fast_kernel<8, 128>), grid, blk, 0, s, a); return hipGetLastError(); }
    if (nk <= 24) { hipLaunchKernelGGL((dec_self_attn_fast_kernel<12, 128>), grid, blk, 0, s, a); return hipGetLastError(); }
    if (nk <= 36) { hipLaunchKernelGGL((dec_self_attn_fast_kernel<18, 128>), grid, blk, 0, s, a); return hipGetLastError(); }
    if (a.t + 1 > MAX_LQ) return hipErrorInvalidValue;
    hipLaunchKernelGGL((dec_attn_kernel<true, 128>), grid, blk, 0, s, a.q, a.kcache, a.vcache, a.anc, a.anc_ld, a.rel_bias, a.bucket,
                       (const int32_t*)nullptr, a.out, a.Q, a.B, a.H, a.t, 0, 0, a.out_h, a.o_ps, a.q_stride, a.h_stride, a.pos_stride,
                       a.slot_stride, a.sat, a.nq_dev, (const int32_t*)nullptr);
    return hipGetLastError();
  }
  const int nk = a.t + 1;
  if (nk <= 8) { hipLaunchKernelGGL(dec_self_attn_fast_kernel<2>, grid, blk, 0, s, a); return hipGetLastError(); }
  if (nk <= 16) { hipLaunchKernelGGL(dec_self_attn_fast_kernel<4>, grid, blk, 0, s, a); return hipGetLastError(); }
  if (nk <= 24) { hipLaunchKernelGGL(dec_self_attn_fast_kernel<6>, grid, blk, 0, s, a); return hipGetLastError(); }
  if (nk <= 32) { hipLaunchKernelGGL(dec_self_attn_fast_kernel<8>, grid, blk, 0, s, a); return hipGetLastError(); }
  if (nk <= 4 * SELF_MAXIT_MAX) {
    hipLaunchKernelGGL(dec_self_attn_fast_kernel<SELF_MAXIT_MAX>, grid, blk, 0, s, a);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(dec_attn_kernel<true>, dim3((items + 3) / 4), dim3(256), 0, s, a.q, a.kcache, a.vcache, a.anc,
                     a.anc_ld, a.rel_bias, a.bucket, (const int32_t*)nullptr, a.out, a.Q, a.B, a.H, a.t, 0, 0, a.out_h, a.o_ps,
                     a.q_stride, a.h_stride, a.pos_stride, a.slot_stride, a.sat, a.nq_dev);
  return hipGetLastError();
}

// Cross-attention: one block per (query, head). The encoder K/V rows of the head and the q rows of
// the query's B beams are staged once in LDS (the per-beam version re-read K/V B times through L2);
// then three block-wide phases: scores — one thread per (beam, key) pair, float4 LDS reads with rows
// padded to 68 floats (conflict-free); softmax — one wave per beam; P.V — one thread per (beam, 4 dims).
constexpr int XK_LD = DKV + 4, QS_LD = DKV + 4;

template <int D>   // head dim: 64, or 128 (t5-3b); rows padded by 4 floats either way (stride = 4 mod 32 banks)
__global__ __launch_bounds__(256) void dec_cross_attn_block_kernel(DecCrossAttnArgs a) {
  constexpr int XK_LD = D + 4, QS_LD = D + 4, D4 = D / 4;   // D4 = float4 pieces per row
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // Bq = beams of the query, B = the beams this block handles (all of them, or one chunk of a.bchunk when the
  // per-beam LDS rows of a large beam — topk = 1000 in the reference's retrieval script — would not fit)
  const int H = a.H, Bq = a.B, inner = H * D;
  const int b_first = a.bchunk ? (int)blockIdx.y * a.bchunk : 0;
  const int B = a.bchunk ? min(a.bchunk, Bq - b_first) : Bq;
  const int qi = blockIdx.x / H, h = blockIdx.x - qi * H;
  if (a.nq_dev && qi >= *a.nq_dev) return;
  const int SLD = a.Lq + 1;
  float* Ks = smem;                        // [Lq][68]
  float* Vs = Ks + (size_t)a.Lq * XK_LD;   // [Lq][64]
  float* Qs = Vs + (size_t)a.Lq * D;     // [B][68]
  float* S = Qs + (size_t)(a.bchunk ? a.bchunk : Bq) * QS_LD;       // [B][Lq+1]
  const int tid = threadIdx.x;
  const int32_t* mrow = a.mask + (size_t)qi * a.Lq;
  const size_t xrow0 = a.offs ? (size_t)a.offs[qi] : (size_t)qi * a.Lq;   // packed or padded encoder rows
  const float* kb = a.xk + xrow0 * a.xld + h * D;
  const float* vb = a.xv + xrow0 * a.xld + h * D;
  // keys at and beyond the last attended position are padding: every loop runs over that prefix only
  // (queries are padded to the batch maximum, typically 2-3x their own length); a.last[q] is computed
  // once per search. All global loads of the block are issued back to back before the first wait.
  const int Lq = a.last[qi];
  if (Lq == 0) {   // query without a single attended token (reported through the ctx status word by mask_lengths_kernel):
                   // its packed encoder has no rows — write zeros instead of reading a neighbour's K/V
    for (int item = tid; item < B * D4; item += 256) {
      const int b = item / D4, c = (item % D4) * 4;
      const size_t oidx = (size_t)(qi * Bq + b_first + b) * inner + h * D + c;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.out_h) store_planes4(a.out_h, a.o_ps, oidx, z, a.sat);
      else *reinterpret_cast<float4*>(a.out + oidx) = z;
    }
    return;
  }
  constexpr int PF = 2;  // float4 K and V items per thread held in registers (covers Lq <= 32 rows)
  float4 pk[PF], pv[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int i = tid + u * 256, j = i / D4, c = (i % D4) * 4;
    pk[u] = make_float4(0.f, 0.f, 0.f, 0.f); pv[u] = pk[u];
    if (j < Lq && mrow[j] != 0) {  // padded keys are never read; zero rows keep 0 * garbage out of the PV sum
      pk[u] = *reinterpret_cast<const float4*>(kb + (size_t)j * a.xld + c);
      pv[u] = *reinterpret_cast<const float4*>(vb + (size_t)j * a.xld + c);
    }
  }
  for (int i = tid; i < B * D4; i += 256) {
    const int b = i / D4, c = (i % D4) * 4;
    *reinterpret_cast<float4*>(Qs + b * QS_LD + c) =
        *reinterpret_cast<const float4*>(a.q + (size_t)(qi * Bq + b_first + b) * inner + h * D + c);
  }
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int i = tid + u * 256, j = i / D4, c = (i % D4) * 4;
    if (j < Lq) {
      *reinterpret_cast<float4*>(Ks + j * XK_LD + c) = pk[u];
      *reinterpret_cast<float4*>(Vs + j * D + c) = pv[u];
    }
  }
  for (int i = tid + PF * 256; i < Lq * D4; i += 256) {  // long queries: remaining rows
    const int j = i / D4, c = (i % D4) * 4;
    float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
    if (mrow[j] != 0) {
      kv = *reinterpret_cast<const float4*>(kb + (size_t)j * a.xld + c);
      vv = *reinterpret_cast<const float4*>(vb + (size_t)j * a.xld + c);
    }
    *reinterpret_cast<float4*>(Ks + j * XK_LD + c) = kv;
    *reinterpret_cast<float4*>(Vs + j * D + c) = vv;
  }
  __syncthreads();
  // scores: two threads per (beam, key) pair, 32 dims each (B * Lq is ~120 pairs for 10 beams: one thread per pair
  // left half of the block idle and made the 16-deep float4 loop the longest serial piece); four partial sums break
  // the dependent FMA chain, the two halves meet through one shuffle
  for (int p2 = tid; p2 < 2 * B * Lq; p2 += 256) {     // B * Lq * 2 and 256 are even: both halves of a pair are in range together
    const int pair = p2 >> 1, hlf = p2 & 1;
    const int b = pair / Lq, j = pair - b * Lq;
    float sv = 0.f;
    const bool live = mrow[j] != 0;
    if (live) {
      const float4* qr = reinterpret_cast<const float4*>(Qs + b * QS_LD) + hlf * (D / 8);
      const float4* kr = reinterpret_cast<const float4*>(Ks + j * XK_LD) + hlf * (D / 8);
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int d = 0; d < D / 8; ++d) {
        const float4 q4 = qr[d], k4 = kr[d];
        a0 = fmaf(q4.x, k4.x, a0); a1 = fmaf(q4.y, k4.y, a1);
        a2 = fmaf(q4.z, k4.z, a2); a3 = fmaf(q4.w, k4.w, a3);
      }
      sv = (a0 + a1) + (a2 + a3);
    }
    sv += __shfl_xor(sv, 1, 64);
    if (hlf == 0) S[b * SLD + j] = live ? sv : -INFINITY;
  }
  __syncthreads();
  // softmax numerators: the thread that owns a (beam, key) score turns it into exp(s - max of the beam's row) — one exp
  // per pair instead of one per (pair, 16 output lanes); the row maximum comes from broadcast LDS reads of the row
  // (a wave-shuffle reduction here was a chain of dependent ds_bpermute latencies and dominated the kernel). The
  // numerators go to a second array: other threads are still reading the scores of the row.
  float* P = S + (size_t)(a.bchunk ? a.bchunk : Bq) * SLD;      // [B][Lq+1]
  for (int pair = tid; pair < B * Lq; pair += 256) {
    const int b = pair / Lq, j = pair - b * Lq;
    const float* row = S + b * SLD;
    float mx = -INFINITY;
    for (int k = 0; k < Lq; ++k) mx = fmaxf(mx, row[k]);
    const float sv = row[j];
    P[b * SLD + j] = (sv == -INFINITY) ? 0.f : expf(sv - mx);
  }
  __syncthreads();
  // P.V: one thread per (beam, 4 output dims)
  for (int item = tid; item < B * D4; item += 256) {
    const int b = item / D4, c = (item % D4) * 4;
    const float* row = P + b * SLD;
    float sum = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < Lq; ++j) {
      const float e = row[j];
      sum += e;
      const float4 v4 = *reinterpret_cast<const float4*>(Vs + j * D + c);
      o.x = fmaf(e, v4.x, o.x); o.y = fmaf(e, v4.y, o.y); o.z = fmaf(e, v4.z, o.z); o.w = fmaf(e, v4.w, o.w);
    }
    const float inv = 1.0f / sum;
    o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv;
    const size_t oidx = (size_t)(qi * Bq + b_first + b) * inner + h * D + c;
    if (a.out_h) store_planes4(a.out_h, a.o_ps, oidx, o, a.sat);
    else *reinterpret_cast<float4*>(a.out + oidx) = o;
  }
}

hipError_t init_t5_kernel_attributes() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(enc_attn_kernel<64>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(enc_attn_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(enc_attn_kernel<128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_cross_attn_block_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(dec_cross_attn_block_kernel<64>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t launch_dec_cross_attn(const DecCrossAttnArgs& a_in, hipStream_t s) {
  DecCrossAttnArgs a = a_in;
  if (a.Lq > MAX_LQ) return hipErrorInvalidValue;
  if (a.dkv == 128) {   // t5-3b heads
    // the block kernel (K / V of the (query, head) staged once for all beams) when everything fits one block's LDS ...
    auto smem128 = [&](int nb) { return ((size_t)a.Lq * (132 + 128) + (size_t)nb * (132 + 2 * (a.Lq + 1)) + 4) * sizeof(float); };
    a.bchunk = 0;
    int chunks128 = 1;
    if (smem128(a.B) > 96 * 1024) {   // many rows per query (the tail pass: beams x remaining positions): chunks of the rows over blockIdx.y
      a.bchunk = 64;
      while (a.bchunk > 1 && smem128(a.bchunk) > 96 * 1024) a.bchunk >>= 1;
      chunks128 = (a.B + a.bchunk - 1) / a.bchunk;
    }
    if (smem128(a.bchunk ? a.bchunk : a.B) <= 160 * 1024) {
      hipLaunchKernelGGL(dec_cross_attn_block_kernel<128>, dim3(a.Q * a.H, chunks128), dim3(256), smem128(a.bchunk ? a.bchunk : a.B), s, a);
      return hipGetLastError();
    }
    a.bchunk = 0;
    // ... else one wave per (beam, head) over the query's encoder K / V rows (unattended keys masked)
    const long items = (long)a.Q * a.B * a.H;
    hipLaunchKernelGGL((dec_attn_kernel<false, 128>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, a.q, a.xk, a.xv,
                       (const uint16_t*)nullptr, 0, (const float*)nullptr, (const int32_t*)nullptr, a.mask, a.out, a.Q, a.B, a.H, 0, a.Lq,
                       a.xld, a.out_h, a.o_ps, (size_t)0, (size_t)0, (size_t)0, (size_t)0, a.sat, a.nq_dev, a.offs);
    return hipGetLastError();
  }
  auto smem_for = [&](int nb) { return ((size_t)a.Lq * (XK_LD + DKV) + (size_t)nb * (QS_LD + 2 * (a.Lq + 1)) + 4) * sizeof(float); };
  a.bchunk = 0;
  int chunks = 1;
  if (smem_for(a.B) > 64 * 1024) {   // large beams: split the query's beams over blockIdx.y (K/V re-staged per chunk)
    a.bchunk = 64;
    while (a.bchunk > 1 && smem_for(a.bchunk) > 64 * 1024) a.bchunk >>= 1;
    chunks = (a.B + a.bchunk - 1) / a.bchunk;
  }
  const size_t smem = smem_for(a.bchunk ? a.bchunk : a.B);
  if (smem > 160 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(dec_cross_attn_block_kernel<64>, dim3(a.Q * a.H, chunks), dim3(256), smem, s, a);
  return hipGetLastError();
}

// zeroes the per-site row sums of a pass (a kernel node rather than a memset node: memset nodes captured into the
// search graph did not re-execute reliably on replay with ROCm 7.2)
__global__ __launch_bounds__(256) void zero_u64_kernel(unsigned long long* __restrict__ p, size_t n) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (i + 1 < n) *reinterpret_cast<ulonglong2*>(p + i) = make_ulonglong2(0ull, 0ull);
  else if (i < n) p[i] = 0ull;
}

hipError_t launch_zero_u64(unsigned long long* p, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(zero_u64_kernel, dim3((unsigned)((n / 2 + 256) / 256)), dim3(256), 0, s, p, n);
  return hipGetLastError();
}

__global__ void mask_lengths_kernel(const int32_t* __restrict__ mask, int32_t* __restrict__ lens, int Q, int Lq,
                                    unsigned int* status) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= Q) return;
  int n = 0;  // index of the last attended key + 1 (the mask need not be a prefix)
  for (int j = 0; j < Lq; ++j) if (mask[(size_t)q * Lq + j] != 0) n = j + 1;
  lens[q] = n;
  if (n == 0 && status) *status = 1u;   // benign race: every writer stores 1
}

hipError_t launch_mask_lengths(const int32_t* mask, int32_t* lens, int Q, int Lq, hipStream_t s, unsigned int* status) {
  hipLaunchKernelGGL(mask_lengths_kernel, dim3((Q + 255) / 256), dim3(256), 0, s, mask, lens, Q, Lq, status);
  return hipGetLastError();
}

}  // namespace rpr
