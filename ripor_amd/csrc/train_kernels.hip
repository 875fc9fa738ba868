// Kernels of the teacher-forced forward of the prefix-oriented ranking fine-tune step:
// reference T5SeqAQEncoderForLngKnpMarginMSE.forward, t5_pretrainer/modeling/t5_generative_retriever.py:902-966.
// The decoder runs over all L positions of every (query, smtid) pair at once (rows = bz * n_docs * L), reusing the
// projection GEMMs, the block attention kernels (causal variant of enc_attn_kernel; cross-attention with the 2L rows
// of a query as its "beams") and the fused RMSNorm of the search path. What is specific to training lives here:
// the teacher-forced input embeddings, the gold-code scores and the margin-MSE losses.
#include <algorithm>

#include "common.h"

namespace rpr {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// decoder_inputs_embeds of every position (reference :194-214 with decoder_input_ids = [-1, c_1 .. c_{L-1}],
// dataset/dataset.py:497-500): row (s, i) = start_token_embed for i = 0, list_decoder_embeds[i-1][c_i] otherwise,
// where codes[s] = (c_1 .. c_L) is the doc encoding of sequence s. One wave per row; fused-RMSNorm outputs optional.
__global__ __launch_bounds__(256) void train_dec_embed_kernel(const float* __restrict__ start, const float* __restrict__ in_embeds,
                                                               const int32_t* __restrict__ codes, float* __restrict__ out,
                                                               int S, int L, int d, int V, XOut xo) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= S * L) return;
  const int s = row / L, i = row - s * L;
  const float* src = start;
  if (i > 0) {
    int tok = codes[(size_t)s * L + (i - 1)];
    tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
    src = in_embeds + ((size_t)(i - 1) * V + tok) * d;
  }
  const float4* s4 = reinterpret_cast<const float4*>(src);
  float4* dst = reinterpret_cast<float4*>(out + (size_t)row * d);
  float ss = 0.f;
  for (int k = lane; k < (d >> 2); k += 64) {
    float4 v = s4[k];
    if (xo.x_h) {   // split-precision mode: the residual stream lives in f16 planes
      __half h[4], l[4];
      split_f16(v.x * X_PLANE_SCALE, h[0], l[0], xo.sat); split_f16(v.y * X_PLANE_SCALE, h[1], l[1], xo.sat);
      split_f16(v.z * X_PLANE_SCALE, h[2], l[2], xo.sat); split_f16(v.w * X_PLANE_SCALE, h[3], l[3], xo.sat);
      const size_t idx = (size_t)row * d + 4 * (size_t)k;
      *reinterpret_cast<uint2*>(xo.x_h + idx) = *reinterpret_cast<uint2*>(h);
      *reinterpret_cast<uint2*>(xo.x_h + xo.x_ps + idx) = *reinterpret_cast<uint2*>(l);
      v = make_float4(x_from_planes(h[0], l[0]), x_from_planes(h[1], l[1]), x_from_planes(h[2], l[2]), x_from_planes(h[3], l[3]));
    } else {
      dst[k] = v;
    }
    ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (xo.ssq) {
    ss = wave_sum_f(ss);
    if (lane == 0) xo.ssq[row] = ssq_to_fix(ss, xo.sat, SSQ_ROW_CAP);
  }
}

hipError_t launch_train_dec_embed(const float* start, const float* in_embeds, const int32_t* codes, float* out, int S,
                                  int L, int d, int V, hipStream_t s, XOut xo) {
  const int rows = S * L;
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(train_dec_embed_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, start, in_embeds, codes, out, S, L, d, V, xo);
  return hipGetLastError();
}

// Gold-code score of every position: final RMSNorm of the decoder stream (x -> decoder_last_hidden_state, times
// d_model^-0.5 under config.scaleup_output_hidden) dotted with the OUTPUT codebook row of the doc's code at that
// position: (query_embeds * doc_embeds).sum(-1) of the reference (:917-918; decode() :812-826). One wave per row,
// exact fp32 in the reference's operation order (w * (x * rsqrt(mean(x^2) + eps)), then the product, then the sum).
__global__ __launch_bounds__(256) void gold_score_kernel(const float* __restrict__ x, const float* __restrict__ ln,
                                                          const float* __restrict__ out_embeds, const int32_t* __restrict__ codes,
                                                          float* __restrict__ scores, int S, int L, int d, int V, float eps,
                                                          float post, const __half* __restrict__ x_h, size_t x_ps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= S * L) return;
  const int s = row / L, i = row - s * L;
  int tok = codes[(size_t)s * L + i];
  tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * d);
  const float4* wr = reinterpret_cast<const float4*>(ln);
  const float4* er = reinterpret_cast<const float4*>(out_embeds + ((size_t)i * V + tok) * d);
  const int n4 = d >> 2;
  auto load4 = [&](int k) -> float4 {
    if (!x_h) return xr[k];
    const size_t idx = (size_t)row * d + 4 * (size_t)k;      // residual stream kept in f16 planes (split-precision mode)
    const uint2 hh = *reinterpret_cast<const uint2*>(x_h + idx), ll = *reinterpret_cast<const uint2*>(x_h + x_ps + idx);
    const __half* h = reinterpret_cast<const __half*>(&hh); const __half* l = reinterpret_cast<const __half*>(&ll);
    return make_float4(x_from_planes(h[0], l[0]), x_from_planes(h[1], l[1]), x_from_planes(h[2], l[2]), x_from_planes(h[3], l[3]));
  };
  float ss = 0.f;
  for (int k = lane; k < n4; k += 64) {
    const float4 v = load4(k);
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  ss = wave_sum_f(ss);
  const float rs = rsqrtf(ss / (float)d + eps);
  float acc = 0.f;
  for (int k = lane; k < n4; k += 64) {
    const float4 v = load4(k), g = wr[k], e = er[k];
    float4 h = make_float4(g.x * (v.x * rs), g.y * (v.y * rs), g.z * (v.z * rs), g.w * (v.w * rs));
    if (post != 1.0f) { h.x *= post; h.y *= post; h.z *= post; h.w *= post; }
    acc += (h.x * e.x + h.y * e.y) + (h.z * e.z + h.w * e.w);
  }
  acc = wave_sum_f(acc);
  if (lane == 0) scores[row] = acc;
}

hipError_t launch_gold_scores(const float* x, const float* ln, const float* out_embeds, const int32_t* codes, float* scores,
                              int S, int L, int d, int V, float eps, float post, hipStream_t s, const __half* x_h, size_t x_ps) {
  const int rows = S * L;
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(gold_score_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ln, out_embeds, codes, scores, S, L, d, V, eps,
                     post, x_h, x_ps);
  return hipGetLastError();
}

// Margin-MSE losses (reference :921-964): for every prefix length k_p, student margin = sum_{i<k_p} pos[b][i] -
// sum_{i<k_p} neg[b][i], loss_p = mean_b (student - (teacher_pos[p][b] - teacher_neg[p][b]))^2 (torch.nn.MSELoss).
// scores: [bz, 2, L] (d = 0 positive, 1 negative). One block; fp32 prefix sums like torch's .sum(-1), float64 mean.
__global__ __launch_bounds__(256) void margin_mse_kernel(const float* __restrict__ scores, const float* __restrict__ teacher_pos,
                                                          const float* __restrict__ teacher_neg, const int32_t* __restrict__ prefix_lens,
                                                          int n_prefix, int bz, int L, float* __restrict__ losses,
                                                          float* __restrict__ margins /* nullable [n_prefix, bz] */) {
  __shared__ double red[256];
  const int tid = threadIdx.x;
  for (int p = 0; p < n_prefix; ++p) {
    const int k = min(prefix_lens[p], L);
    double part = 0.0;
    for (int b = tid; b < bz; b += 256) {
      const float* ps = scores + (size_t)b * 2 * L;
      const float* ns = ps + L;
      float sp = 0.f, sn = 0.f;
      for (int i = 0; i < k; ++i) { sp += ps[i]; sn += ns[i]; }
      const float sm = sp - sn;
      const float tm = teacher_pos[(size_t)p * bz + b] - teacher_neg[(size_t)p * bz + b];
      const float dlt = sm - tm;
      if (margins) margins[(size_t)p * bz + b] = sm;
      part += (double)(dlt * dlt);
    }
    red[tid] = part;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    if (tid == 0) losses[p] = (float)(red[0] / (double)bz);
    __syncthreads();
  }
}

hipError_t launch_margin_mse(const float* scores, const float* teacher_pos, const float* teacher_neg, const int32_t* prefix_lens,
                             int n_prefix, int bz, int L, float* losses, float* margins, hipStream_t s) {
  hipLaunchKernelGGL(margin_mse_kernel, dim3(1), dim3(256), 0, s, scores, teacher_pos, teacher_neg, prefix_lens, n_prefix, bz, L,
                     losses, margins);
  return hipGetLastError();
}


// ====================================================================================================================
// Backward pass of the ranking fine-tune step (reference: loss.backward() in tasks/trainer.py:203-275 over
// T5SeqAQEncoderForLngKnpMarginMSE.forward). Activations are fp32 here; every matrix product of the backward pass is
// the forward GEMM kernel on explicitly transposed operands (dX = dY W -> dY (W^T)^T, dW = dY^T X -> (dY^T)(X^T)^T), so
// the only new arithmetic kernels are the row-wise / per-(sequence, head) ones below. Everything is deterministic:
// cross-row sums go through per-block partials reduced in a fixed order, scatter-adds into embedding tables through
// 2^-32 fixed-point integer atomics.
// ====================================================================================================================

// out[C, ldo] = in[R, C]^T, columns r >= R of the output zero-filled up to Rpad (the K dimension of a dW product must
// be a multiple of the GEMM's K-tile)
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C,
                                                             int ldi, int Rpad) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < R && c < C) ? in[(size_t)r * ldi + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (c < C && r < Rpad) out[(size_t)c * Rpad + r] = tile[tx][ty + 8 * k];
  }
}

hipError_t launch_transpose_pad(const float* in, float* out, int R, int C, int ldi, int Rpad, hipStream_t s) {
  if (R <= 0 || C <= 0) return hipSuccess;
  hipLaunchKernelGGL(transpose_pad_kernel, dim3((C + 31) / 32, (Rpad + 31) / 32), dim3(256), 0, s, in, out, R, C, ldi, Rpad);
  return hipGetLastError();
}

// y[i] = (mask[i] > 0) ? y[i] : 0     (ReLU backward on the stored post-activation)
__global__ __launch_bounds__(256) void relu_bwd_kernel(float* __restrict__ dy, const float* __restrict__ act, size_t n) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  float4 d = *reinterpret_cast<float4*>(dy + i);
  const float4 a = *reinterpret_cast<const float4*>(act + i);
  d.x = a.x > 0.f ? d.x : 0.f; d.y = a.y > 0.f ? d.y : 0.f; d.z = a.z > 0.f ? d.z : 0.f; d.w = a.w > 0.f ? d.w : 0.f;
  *reinterpret_cast<float4*>(dy + i) = d;
}
hipError_t launch_relu_bwd(float* dy, const float* act, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, dy, act, n);
  return hipGetLastError();
}

// RMSNorm backward. h = post * w * x * rs, rs = rsqrt(mean(x^2) + eps). Given dh:
//   g = dh * post * w;  dx = rs * g - x * rs^3 * mean(g * x);  dw += dh * post * x * rs (summed over rows)
// dx_out[row] = dx (+ dres[row]: the gradient arriving through the residual connection). One wave per row; the 16 rows
// of a block (4 per wave, in row order) leave one partial dw row (w_part[blk][d]); colsum_kernel adds the partials in
// block order.
constexpr int NB_ROWS = 16;   // rows per block of rmsnorm_bwd_kernel (4 waves x 4 rows)
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ dh, const float* __restrict__ dres,
                                                           float* __restrict__ dx_out, float* __restrict__ w_part, int rows,
                                                           int d, float eps, float post) {
  extern __shared__ float part[];   // [4][d]: each wave's dw contributions of its 4 rows, added in row order
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n4 = d >> 2;
  float* mypart = part + wave * d;
  for (int i = lane; i < n4; i += 64) reinterpret_cast<float4*>(mypart)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int NV = 4, NR = NB_ROWS / 4;
  const float4* wr = reinterpret_cast<const float4*>(w);
  if (n4 <= 64 * NV) {
    // d <= 1024: the wave requests x, dh and the residual gradient of ALL FOUR of its rows before it touches the first one
    // (192 registers). Row by row the kernel was a chain of load -> reduce -> reduce -> load -> store latencies with 8 waves
    // per CU in flight: 65 us for the 100 MB of a [8192, 768] site (1.5 TB/s), 4 ms of the 31-ms bf16 step.
    float4 xv[NR][NV], gv[NR][NV], rv[NR][NV];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
      const int row = blockIdx.x * NB_ROWS + rr * 4 + wave;
      const bool rok = row < rows;
      const float4* xr = reinterpret_cast<const float4*>(x + (size_t)(rok ? row : 0) * d);
      const float4* gr = reinterpret_cast<const float4*>(dh + (size_t)(rok ? row : 0) * d);
      const float4* rr4 = reinterpret_cast<const float4*>((dres ? dres : x) + (size_t)(rok ? row : 0) * d);
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        const bool ok = rok && i < n4;
        // unconditional loads of clamped addresses, zeroed by selects: behind the range test every load was retired
        // (s_waitcnt vmcnt(0)) before the next one was requested
        const int ic = i < n4 ? i : 0;
        const float4 lx = xr[ic], lg = gr[ic], lr = rr4[ic];
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        xv[rr][k] = ok ? lx : z4;
        gv[rr][k] = ok ? lg : z4;
        rv[rr][k] = (ok && dres) ? lr : z4;
      }
    }
    float4 wv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) { const int i = lane + 64 * k; const float4 lw = wr[i < n4 ? i : 0]; wv[k] = i < n4 ? lw : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
      const int row = blockIdx.x * NB_ROWS + rr * 4 + wave;
      if (row >= rows) break;
      float ss = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) ss += xv[rr][k].x * xv[rr][k].x + xv[rr][k].y * xv[rr][k].y + xv[rr][k].z * xv[rr][k].z + xv[rr][k].w * xv[rr][k].w;
      ss = wave_sum_f(ss);
      const float rs = rsqrtf(ss / (float)d + eps);
      float gx = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const float4 v = xv[rr][k], g = gv[rr][k], ww = wv[k];
        gx += (g.x * ww.x) * v.x + (g.y * ww.y) * v.y + (g.z * ww.z) * v.z + (g.w * ww.w) * v.w;
      }
      gx = wave_sum_f(gx) * post;
      const float c = gx * rs * rs * rs / (float)d;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        if (i < n4) {
          const float4 v = xv[rr][k], g = gv[rr][k], ww = wv[k], r = rv[rr][k];
          float4 o = make_float4(rs * post * g.x * ww.x - v.x * c, rs * post * g.y * ww.y - v.y * c,
                                 rs * post * g.z * ww.z - v.z * c, rs * post * g.w * ww.w - v.w * c);
          if (dres) { o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
          reinterpret_cast<float4*>(dx_out + (size_t)row * d)[i] = o;
          float4 acc = reinterpret_cast<float4*>(mypart)[i];
          acc.x += g.x * post * v.x * rs; acc.y += g.y * post * v.y * rs; acc.z += g.z * post * v.z * rs; acc.w += g.w * post * v.w * rs;
          reinterpret_cast<float4*>(mypart)[i] = acc;
        }
      }
    }
  } else {
    // wider rows: re-read (three passes over the row)
    for (int rr = 0; rr < NR; ++rr) {
      const int row = blockIdx.x * NB_ROWS + rr * 4 + wave;
      if (row >= rows) break;
      const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * d);
      const float4* gr = reinterpret_cast<const float4*>(dh + (size_t)row * d);
      float ss = 0.f;
      for (int i = lane; i < n4; i += 64) { const float4 v = xr[i]; ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
      ss = wave_sum_f(ss);
      const float rs = rsqrtf(ss / (float)d + eps);
      float gx = 0.f;
      for (int i = lane; i < n4; i += 64) {
        const float4 v = xr[i], g = gr[i], ww = wr[i];
        gx += (g.x * ww.x) * v.x + (g.y * ww.y) * v.y + (g.z * ww.z) * v.z + (g.w * ww.w) * v.w;
      }
      gx = wave_sum_f(gx) * post;
      const float c = gx * rs * rs * rs / (float)d;
      for (int i = lane; i < n4; i += 64) {
        const float4 v = xr[i], g = gr[i], ww = wr[i];
        float4 o = make_float4(rs * post * g.x * ww.x - v.x * c, rs * post * g.y * ww.y - v.y * c,
                               rs * post * g.z * ww.z - v.z * c, rs * post * g.w * ww.w - v.w * c);
        if (dres) { const float4 r = reinterpret_cast<const float4*>(dres + (size_t)row * d)[i]; o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
        reinterpret_cast<float4*>(dx_out + (size_t)row * d)[i] = o;
        float4 acc = reinterpret_cast<float4*>(mypart)[i];
        acc.x += g.x * post * v.x * rs; acc.y += g.y * post * v.y * rs; acc.z += g.z * post * v.z * rs; acc.w += g.w * post * v.w * rs;
        reinterpret_cast<float4*>(mypart)[i] = acc;
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < d; k += 256)
    w_part[(size_t)blockIdx.x * d + k] = (part[k] + part[d + k]) + (part[2 * d + k] + part[3 * d + k]);
}

// out[k] (+)= sum over p of part[p][k]: a block owns 16 columns; 16 row groups add their contiguous share of the partials
// in order, then the 16 group sums are added in order (deterministic, short dependent chains, d / 16 blocks)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, int d,
                                                      int accumulate) {
  __shared__ float red[16][16];
  const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int k = blockIdx.x * 16 + col;
  const int per = (nparts + 15) / 16, p0 = grp * per, p1 = min(nparts, p0 + per);
  float s = 0.f;
  if (k < d)
    for (int p = p0; p < p1; ++p) s += part[(size_t)p * d + k];
  red[grp][col] = s;
  __syncthreads();
  if (grp == 0 && k < d) {
    float t = 0.f;
#pragma unroll
    for (int g16 = 0; g16 < 16; ++g16) t += red[g16][col];
    out[k] = accumulate ? out[k] + t : t;
  }
}

// dw == nullptr: only the partials are written (w_part[nblk][d], nblk = rmsnorm_bwd_parts(rows)); the caller sums the
// partials of several norm sites in one launch_colsum_multi
int rmsnorm_bwd_parts(int rows) { return (rows + NB_ROWS - 1) / NB_ROWS; }
hipError_t launch_rmsnorm_bwd(const float* x, const float* w, const float* dh, const float* dres, float* dx_out, float* w_part,
                              float* dw, int rows, int d, float eps, float post, int accumulate_dw, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  const int nblk = rmsnorm_bwd_parts(rows);
  hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3(nblk), dim3(256), 4 * d * sizeof(float), s, x, w, dh, dres, dx_out, w_part, rows, d,
                     eps, post);
  if (dw) hipLaunchKernelGGL(colsum_kernel, dim3((d + 15) / 16), dim3(256), 0, s, w_part, dw, nblk, d, accumulate_dw);
  return hipGetLastError();
}

// colsum_kernel for up to four sites at once (blockIdx.y = site; the layer-norm weight gradients of one transformer layer):
// the same arithmetic and order per site
__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumSites p, int d) {
  __shared__ float red[16][16];
  const int y = blockIdx.y;
  const float* part = y == 0 ? p.part[0] : y == 1 ? p.part[1] : y == 2 ? p.part[2] : p.part[3];
  float* out = y == 0 ? p.out[0] : y == 1 ? p.out[1] : y == 2 ? p.out[2] : p.out[3];
  const int nparts = y == 0 ? p.nparts[0] : y == 1 ? p.nparts[1] : y == 2 ? p.nparts[2] : p.nparts[3];
  const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int k = blockIdx.x * 16 + col;
  const int per = (nparts + 15) / 16, p0 = grp * per, p1 = min(nparts, p0 + per);
  float s = 0.f;
  if (k < d)
    for (int q = p0; q < p1; ++q) s += part[(size_t)q * d + k];
  red[grp][col] = s;
  __syncthreads();
  if (grp == 0 && k < d) {
    float t = 0.f;
#pragma unroll
    for (int g16 = 0; g16 < 16; ++g16) t += red[g16][col];
    out[k] = t;
  }
}
hipError_t launch_colsum_multi(const ColsumSites& p, int d, hipStream_t s) {
  if (p.n <= 0) return hipSuccess;
  if (p.n > ColsumSites::MAXS) return hipErrorInvalidValue;
  hipLaunchKernelGGL(colsum_multi_kernel, dim3((d + 15) / 16, p.n), dim3(256), 0, s, p, d);
  return hipGetLastError();
}

// ---- attention backward, one block per (sequence, head), everything in LDS -----------------------------------------
// Self-attention (encoder: bidirectional buckets + key padding mask; decoder, teacher-forced: causal, unidirectional
// buckets). qkv: [S*Ls, 3*inner]; dO: [S*Ls, inner]; outputs dqkv [S*Ls, 3*inner] and this block's part of the
// relative-bias gradient dbias_part[(s * H + h)][buckets]. Recomputes P from q, k (nothing but q, k, v was saved).
// ---- attention backward: shared pieces ----------------------------------------------------------------------------------
// LDS rows of Q / K / V / dO are padded to AST = 65 floats: the score phase reads K[j][d] with j across the lanes, and an
// unpadded 64-float stride would put all of them on one bank.
constexpr int AST = DKV + 1;

__device__ __forceinline__ void stage_row4(float* dst, const float* src) {
  const float4 v = *reinterpret_cast<const float4*>(src);
  dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
}

// Ps[i][j] = score (or -inf where !ok), Gs[i][j] = dP = dO_i . V_j, 2 x 2 outputs per thread (half the LDS reads per FMA)
template <class OK, class BIAS>
__device__ __forceinline__ void attn_scores_dp(const float* Qs, const float* Ds, const float* Ks, const float* Vs, int nq, int nk,
                                               float* Ps, float* Gs, int PL, OK ok, BIAS bias) {
  const int hq = (nq + 1) >> 1, hk = (nk + 1) >> 1;
  for (int p = threadIdx.x; p < hq * hk; p += 256) {
    const int ti = p / hk, tj = p - ti * hk;
    const int i0 = ti, i1 = ti + hq, j0 = tj, j1 = tj + hk;
    const bool vi1 = i1 < nq, vj1 = j1 < nk;
    const float *q0 = Qs + i0 * AST, *q1 = Qs + (vi1 ? i1 : i0) * AST, *o0 = Ds + i0 * AST, *o1 = Ds + (vi1 ? i1 : i0) * AST;
    const float *k0 = Ks + j0 * AST, *k1 = Ks + (vj1 ? j1 : j0) * AST, *v0 = Vs + j0 * AST, *v1 = Vs + (vj1 ? j1 : j0) * AST;
    float s00 = 0.f, s01 = 0.f, s10 = 0.f, s11 = 0.f, g00 = 0.f, g01 = 0.f, g10 = 0.f, g11 = 0.f;
#pragma unroll 8
    for (int d = 0; d < DKV; ++d) {
      const float a0 = q0[d], a1 = q1[d], b0 = k0[d], b1 = k1[d], c0 = o0[d], c1 = o1[d], e0 = v0[d], e1 = v1[d];
      s00 = fmaf(a0, b0, s00); s01 = fmaf(a0, b1, s01); s10 = fmaf(a1, b0, s10); s11 = fmaf(a1, b1, s11);
      g00 = fmaf(c0, e0, g00); g01 = fmaf(c0, e1, g01); g10 = fmaf(c1, e0, g10); g11 = fmaf(c1, e1, g11);
    }
    auto put = [&](int i, int j, float sc, float dp) {
      const bool k = ok(i, j);
      Ps[i * PL + j] = k ? sc + bias(i, j) : -INFINITY;
      Gs[i * PL + j] = k ? dp : 0.f;
    };
    put(i0, j0, s00, g00);
    if (vj1) put(i0, j1, s01, g01);
    if (vi1) put(i1, j0, s10, g10);
    if (vi1 && vj1) put(i1, j1, s11, g11);
  }
}

// row softmax of Ps in place, then Gs = dS = P * (dP - sum_j dP P); eight lanes per row
__device__ __forceinline__ void attn_softmax_ds(float* Ps, float* Gs, int nq, int nk, int PL) {
  const int sub = threadIdx.x & 7;
  for (int i = threadIdx.x >> 3; i < nq; i += 32) {
    float* pr = Ps + i * PL; float* gr = Gs + i * PL;
    float mx = -INFINITY;
    for (int j = sub; j < nk; j += 8) mx = fmaxf(mx, pr[j]);
    for (int m = 1; m < 8; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 8));
    float sum = 0.f;
    for (int j = sub; j < nk; j += 8) { const float e = (pr[j] == -INFINITY) ? 0.f : expf(pr[j] - mx); pr[j] = e; sum += e; }
    for (int m = 1; m < 8; m <<= 1) sum += __shfl_xor(sum, m, 8);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    float c = 0.f;
    for (int j = sub; j < nk; j += 8) { const float pj = pr[j] * inv; pr[j] = pj; c = fmaf(gr[j], pj, c); }
    for (int m = 1; m < 8; m <<= 1) c += __shfl_xor(c, m, 8);
    for (int j = sub; j < nk; j += 8) gr[j] = pr[j] * (gr[j] - c);
  }
}

// out[r][c] = sum_j W[r][j] X[j][c] for two rows r0, r1 = r0 + hr of W (row stride PL) and columns c = lane16 + 16 k
__device__ __forceinline__ void attn_rows_times(const float* W, int PL, int r0, int r1, const float* X, int nj, int l16,
                                                float (&a0)[4], float (&a1)[4]) {
  for (int j = 0; j < nj; ++j) {
    const float w0 = W[r0 * PL + j], w1 = W[r1 * PL + j];
    const float* x = X + j * AST + l16;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float xv = x[16 * k]; a0[k] = fmaf(w0, xv, a0[k]); a1[k] = fmaf(w1, xv, a1[k]); }
  }
}
// out[r][c] = sum_i W[i][r] X[i][c] (the transposed product) for two columns r0, r1 of W
__device__ __forceinline__ void attn_cols_times(const float* W, int PL, int r0, int r1, const float* X, int ni, int l16,
                                                float (&a0)[4], float (&a1)[4]) {
  for (int i = 0; i < ni; ++i) {
    const float w0 = W[i * PL + r0], w1 = W[i * PL + r1];
    const float* x = X + i * AST + l16;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float xv = x[16 * k]; a0[k] = fmaf(w0, xv, a0[k]); a1[k] = fmaf(w1, xv, a1[k]); }
  }
}
__device__ __forceinline__ void attn_store16(float* dst, int l16, const float (&a)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) dst[l16 + 16 * k] = a[k];
}

// One block per (sequence, head). Everything of the head lives in LDS: Q, K, V, dO [Ls][65], P and dS [Ls][Ls + 1].
__global__ __launch_bounds__(256) void self_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dO,
                                                             const int32_t* __restrict__ mask, const float* __restrict__ rel_bias,
                                                             const int32_t* __restrict__ bucket, float* __restrict__ dqkv,
                                                             float* __restrict__ dbias_part, int S, int Ls, int H, int buckets,
                                                             int causal) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int s = blockIdx.x / H, h = blockIdx.x - s * H, inner = H * DKV, ld = 3 * inner, tid = threadIdx.x;
  const int PL = Ls + 1, nd = causal ? Ls : 2 * Ls - 1;
  float* Qs = sm;                       // [Ls][AST]
  float* Ks = Qs + Ls * AST;
  float* Vs = Ks + Ls * AST;
  float* Ds = Vs + Ls * AST;            // dO
  float* Ps = Ds + Ls * AST;            // [Ls][PL]  probabilities
  float* Gs = Ps + Ls * PL;             // [Ls][PL]  dP, then dS
  float* Bs = Gs + Ls * PL;             // [buckets] bias of this head
  float* diag = Bs + 64;                // [nd] sum of dS along each diagonal j - i
  int* bk = reinterpret_cast<int*>(diag + 2 * Ls);   // [nd] bucket of each diagonal
  int* mk = bk + 2 * Ls;                // [Ls] key mask
  const size_t row0 = (size_t)s * Ls;
  for (int i = tid; i < Ls * 16; i += 256) {
    const int r = i >> 4, c = (i & 15) * 4;
    const float* base = qkv + (row0 + r) * ld + h * DKV + c;
    stage_row4(Qs + r * AST + c, base); stage_row4(Ks + r * AST + c, base + inner); stage_row4(Vs + r * AST + c, base + 2 * inner);
    stage_row4(Ds + r * AST + c, dO + (row0 + r) * inner + h * DKV + c);
  }
  if (tid < buckets) Bs[tid] = rel_bias[tid * H + h];
  // diagonal t: causal i - j = t; bidirectional j - i = t - (Ls - 1)
  for (int t = tid; t < nd; t += 256) bk[t] = causal ? bucket[t] : bucket[t - (Ls - 1) + (MAX_LQ - 1)];
  for (int j = tid; j < Ls; j += 256) mk[j] = (causal || mask[row0 + j] != 0) ? 1 : 0;
  __syncthreads();
  attn_scores_dp(Qs, Ds, Ks, Vs, Ls, Ls, Ps, Gs, PL,
                 [&](int i, int j) { return causal ? (j <= i) : (mk[j] != 0); },
                 [&](int i, int j) { return Bs[bk[causal ? i - j : j - i + Ls - 1]]; });
  __syncthreads();
  attn_softmax_ds(Ps, Gs, Ls, Ls, PL);
  __syncthreads();
  {  // dQ_i = sum_j dS_ij K_j ; dK_j = sum_i dS_ij Q_i ; dV_j = sum_i P_ij dO_i — two rows per thread
    const int hr = (Ls + 1) >> 1, l16 = tid & 15;
    for (int t = tid >> 4; t < hr; t += 16) {
      const int r0 = t, r1 = (t + hr < Ls) ? t + hr : t;
      float dq0[4] = {}, dq1[4] = {}, dk0[4] = {}, dk1[4] = {}, dv0[4] = {}, dv1[4] = {};
      attn_rows_times(Gs, PL, r0, r1, Ks, Ls, l16, dq0, dq1);
      attn_cols_times(Gs, PL, r0, r1, Qs, Ls, l16, dk0, dk1);
      attn_cols_times(Ps, PL, r0, r1, Ds, Ls, l16, dv0, dv1);
      float* ob = dqkv + (row0 + r0) * ld + h * DKV;
      attn_store16(ob, l16, dq0); attn_store16(ob + inner, l16, dk0); attn_store16(ob + 2 * inner, l16, dv0);
      if (r1 != r0) {
        ob = dqkv + (row0 + r1) * ld + h * DKV;
        attn_store16(ob, l16, dq1); attn_store16(ob + inner, l16, dk1); attn_store16(ob + 2 * inner, l16, dv1);
      }
    }
  }
  // bias gradient of this block: dS is zero wherever the score was masked, so each diagonal is summed whole (fixed
  // order), then the diagonals of one bucket
  for (int t = tid; t < nd; t += 256) {
    const int off = causal ? -t : t - (Ls - 1);      // j - i
    float acc = 0.f;
    for (int i = 0; i < Ls; ++i) { const int j = i + off; if (j >= 0 && j < Ls) acc += Gs[i * PL + j]; }
    diag[t] = acc;
  }
  __syncthreads();
  if (tid < buckets) {
    float acc = 0.f;
    for (int t = 0; t < nd; ++t) if (bk[t] == tid) acc += diag[t];
    dbias_part[(size_t)blockIdx.x * buckets + tid] = acc;
  }
}

// dbias[bucket][h] += sum over sequences s of part[(s*H + h)][bucket]: eight groups of sequences in parallel, each in s
// order, then the eight partial sums in group order (fixed order: bitwise reproducible)
__global__ __launch_bounds__(512) void bias_reduce_kernel(const float* __restrict__ part, float* __restrict__ dbias, int S, int H,
                                                           int buckets) {
  __shared__ float red[8][64];
  const int h = blockIdx.x, b = threadIdx.x & 63, grp = threadIdx.x >> 6;
  float acc = 0.f;
  if (b < buckets)
    for (int s = grp; s < S; s += 8) acc += part[((size_t)s * H + h) * buckets + b];
  red[grp][b] = acc;
  __syncthreads();
  if (grp == 0 && b < buckets) {
    float t = 0.f;
    for (int k = 0; k < 8; ++k) t += red[k][b];
    dbias[b * H + h] += t;
  }
}

size_t self_attn_bwd_smem(int Ls, int buckets) {
  (void)buckets;   // <= 64, fixed slot
  return ((size_t)4 * Ls * AST + 2 * (size_t)Ls * (Ls + 1) + 64 + 5 * (size_t)Ls) * sizeof(float);
}

hipError_t launch_self_attn_bwd(const float* qkv, const float* dO, const int32_t* mask, const float* rel_bias, const int32_t* bucket,
                                float* dqkv, float* dbias_part, float* dbias, int S, int Ls, int H, int buckets, int causal,
                                hipStream_t s) {
  const size_t smem = self_attn_bwd_smem(Ls, buckets);
  if (smem > 160 * 1024 || buckets > 64) return hipErrorInvalidValue;
  static const bool mfma_off = [] { const char* e = dev_getenv("RPR_TRAIN_ATTN_MFMA"); return e && atoi(e) == 0; }();
  if (Ls <= 32 && !mfma_off) {   // one wave per (sequence, head) on the fp32 matrix cores (tail_kernels.hip)
    const hipError_t e = launch_train_self_attn_bwd_mfma(qkv, dO, mask, rel_bias, bucket, dqkv, dbias_part, S, Ls, H, buckets, causal, s);
    if (e != hipSuccess) return e;
  } else
  hipLaunchKernelGGL(self_attn_bwd_kernel, dim3(S * H), dim3(256), smem, s, qkv, dO, mask, rel_bias, bucket, dqkv, dbias_part, S, Ls,
                     H, buckets, causal);
  hipLaunchKernelGGL(bias_reduce_kernel, dim3(H), dim3(512), 0, s, dbias_part, dbias, S, H, buckets);
  return hipGetLastError();
}

// Cross-attention backward, block per (query, head): the n = ndoc*L decoder rows of the query against its Lq encoder
// keys. q: [bz*n, inner]; K/V of this layer: rows (query, j) at xk/xv + (query*Lq + j) * xld; dO: [bz*n, inner].
// Outputs dq [bz*n, inner] and dK/dV into dxk/dxv (same layout as xk/xv): every (query, head) block owns its rows
// and columns, so there is no accumulation across blocks.
__global__ __launch_bounds__(256) void cross_attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ xk,
                                                              const float* __restrict__ xv, int xld, const int32_t* __restrict__ mask,
                                                              const float* __restrict__ dO, float* __restrict__ dq,
                                                              float* __restrict__ dxk, float* __restrict__ dxv, int n, int Lq, int H) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int qi = blockIdx.x / H, h = blockIdx.x - qi * H, inner = H * DKV, tid = threadIdx.x;
  const int PL = Lq + 1;
  float* Qs = sm;                     // [n][AST]
  float* Ds = Qs + n * AST;           // dO [n][AST]
  float* Ks = Ds + n * AST;           // [Lq][AST]
  float* Vs = Ks + Lq * AST;
  float* Ps = Vs + Lq * AST;          // [n][PL]
  float* Gs = Ps + n * PL;            // [n][PL]
  int* mk = reinterpret_cast<int*>(Gs + n * PL);   // [Lq]
  for (int i = tid; i < n * 16; i += 256) {
    const int r = i >> 4, c = (i & 15) * 4;
    const size_t o = ((size_t)qi * n + r) * inner + h * DKV + c;
    stage_row4(Qs + r * AST + c, q + o); stage_row4(Ds + r * AST + c, dO + o);
  }
  for (int i = tid; i < Lq * 16; i += 256) {
    const int j = i >> 4, c = (i & 15) * 4;
    const size_t o = ((size_t)qi * Lq + j) * xld + h * DKV + c;
    stage_row4(Ks + j * AST + c, xk + o); stage_row4(Vs + j * AST + c, xv + o);
  }
  for (int j = tid; j < Lq; j += 256) mk[j] = mask[(size_t)qi * Lq + j];
  __syncthreads();
  attn_scores_dp(Qs, Ds, Ks, Vs, n, Lq, Ps, Gs, PL, [&](int, int j) { return mk[j] != 0; }, [](int, int) { return 0.f; });
  __syncthreads();
  attn_softmax_ds(Ps, Gs, n, Lq, PL);
  __syncthreads();
  const int l16 = tid & 15;
  {  // dq_i = sum_j dS_ij K_j
    const int hr = (n + 1) >> 1;
    for (int t = tid >> 4; t < hr; t += 16) {
      const int r0 = t, r1 = (t + hr < n) ? t + hr : t;
      float a0[4] = {}, a1[4] = {};
      attn_rows_times(Gs, PL, r0, r1, Ks, Lq, l16, a0, a1);
      attn_store16(dq + ((size_t)qi * n + r0) * inner + h * DKV, l16, a0);
      if (r1 != r0) attn_store16(dq + ((size_t)qi * n + r1) * inner + h * DKV, l16, a1);
    }
  }
  {  // dK_j = sum_i dS_ij q_i ; dV_j = sum_i P_ij dO_i
    const int hr = (Lq + 1) >> 1;
    for (int t = tid >> 4; t < hr; t += 16) {
      const int j0 = t, j1 = (t + hr < Lq) ? t + hr : t;
      float dk0[4] = {}, dk1[4] = {}, dv0[4] = {}, dv1[4] = {};
      attn_cols_times(Gs, PL, j0, j1, Qs, n, l16, dk0, dk1);
      attn_cols_times(Ps, PL, j0, j1, Ds, n, l16, dv0, dv1);
      size_t o = ((size_t)qi * Lq + j0) * xld + h * DKV;
      attn_store16(dxk + o, l16, dk0); attn_store16(dxv + o, l16, dv0);
      if (j1 != j0) { o = ((size_t)qi * Lq + j1) * xld + h * DKV; attn_store16(dxk + o, l16, dk1); attn_store16(dxv + o, l16, dv1); }
    }
  }
}

size_t cross_attn_bwd_smem(int n, int Lq) {
  return ((size_t)2 * n * AST + 2 * (size_t)Lq * AST + 2 * (size_t)n * (Lq + 1) + (size_t)Lq) * sizeof(float);
}

hipError_t launch_cross_attn_bwd(const float* q, const float* xk, const float* xv, int xld, const int32_t* mask, const float* dO,
                                 float* dq, float* dxk, float* dxv, int bz, int n, int Lq, int H, hipStream_t s) {
  const size_t smem = cross_attn_bwd_smem(n, Lq);
  if (smem > 160 * 1024) return hipErrorInvalidValue;
  // (an fp32-MFMA version — one wave per (query, head), dK / dV accumulated over the row tiles in the matrix cores — measured
  // 91 us against this kernel's 88: one wave per SIMD by its 33 KB of strips; removed)
  hipLaunchKernelGGL(cross_attn_bwd_kernel, dim3(bz * H), dim3(256), smem, s, q, xk, xv, xld, mask, dO, dq, dxk, dxv, n, Lq, H);
  return hipGetLastError();
}

// ---- scatter-add of rows into an embedding-table gradient, deterministic ---------------------------------------------
// acc: int64 [table_rows, d], 2^-32 fixed point. Row r of src goes to table row idx(r) (< 0: skipped).
constexpr double GRAD_FIX = 4294967296.0;
__global__ __launch_bounds__(256) void scatter_rows_fix_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                                unsigned long long* __restrict__ acc, int rows, int d) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int t = idx[row];
  if (t < 0) return;
  for (int k = lane; k < d; k += 64)
    atomicAdd(acc + (size_t)t * d + k, (unsigned long long)(long long)llrint((double)src[(size_t)row * d + k] * GRAD_FIX));
}
// dst[i] += acc[i] * 2^-32 ; acc[i] = 0
__global__ __launch_bounds__(256) void fix_flush_kernel(unsigned long long* __restrict__ acc, float* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long v = (long long)acc[i];
  if (v != 0) { dst[i] += (float)((double)v / GRAD_FIX); acc[i] = 0ull; }
}
hipError_t launch_scatter_rows_fix(const float* src, const int32_t* idx, unsigned long long* acc, int rows, int d, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(scatter_rows_fix_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, src, idx, acc, rows, d);
  return hipGetLastError();
}
hipError_t launch_fix_flush(unsigned long long* acc, float* dst, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(fix_flush_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, acc, dst, n);
  return hipGetLastError();
}

// row r = (s, i) of the teacher-forced decoder -> index of its input-embedding row in the stacked [L, V] tables
// (i = 0: the start embedding, index -1) and of its gold code's output-codebook row
__global__ __launch_bounds__(256) void train_indices_kernel(const int32_t* __restrict__ codes, int32_t* __restrict__ in_idx,
                                                             int32_t* __restrict__ out_idx, int S, int L, int V) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= S * L) return;
  const int s = r / L, i = r - s * L;
  auto clampv = [&](int t) { return t < 0 ? 0 : (t >= V ? V - 1 : t); };
  in_idx[r] = i == 0 ? -1 : (i - 1) * V + clampv(codes[(size_t)s * L + i - 1]);
  out_idx[r] = i * V + clampv(codes[(size_t)s * L + i]);
}
hipError_t launch_train_indices(const int32_t* codes, int32_t* in_idx, int32_t* out_idx, int S, int L, int V, hipStream_t s) {
  hipLaunchKernelGGL(train_indices_kernel, dim3((S * L + 255) / 256), dim3(256), 0, s, codes, in_idx, out_idx, S, L, V);
  return hipGetLastError();
}

// out[k] += sum over rows r with sel[r] < 0 of src[r][k] (gradient of the start embedding). A block owns 16 columns; 16 row
// groups add their contiguous share of the rows in increasing order, then the 16 group sums are added in group order
// (deterministic; one thread walking all rows took 480 us of the step)
__global__ __launch_bounds__(256) void sum_selected_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ sel,
                                                                 float* __restrict__ out, int rows, int d) {
  __shared__ float red[16][16];
  const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int k = blockIdx.x * 16 + col;
  const int per = (rows + 15) / 16, r0 = grp * per, r1 = min(rows, r0 + per);
  float a = 0.f;
  if (k < d)
    for (int r = r0; r < r1; ++r) if (sel[r] < 0) a += src[(size_t)r * d + k];
  red[grp][col] = a;
  __syncthreads();
  if (grp == 0 && k < d) {
    float t = 0.f;
#pragma unroll
    for (int g16 = 0; g16 < 16; ++g16) t += red[g16][col];
    out[k] += t;
  }
}
hipError_t launch_sum_selected_rows(const float* src, const int32_t* sel, float* out, int rows, int d, hipStream_t s) {
  hipLaunchKernelGGL(sum_selected_rows_kernel, dim3((d + 15) / 16), dim3(256), 0, s, src, sel, out, rows, d);
  return hipGetLastError();
}

// ---- loss and gold-score backward -------------------------------------------------------------------------------------
// dscore[b][side][i] = sum_p [i < k_p] * sign(side) * 2 / bz * (student_margin[p][b] - teacher_margin[p][b])
__global__ __launch_bounds__(256) void margin_mse_bwd_kernel(const float* __restrict__ margins, const float* __restrict__ teacher_pos,
                                                              const float* __restrict__ teacher_neg, const int32_t* __restrict__ prefix_lens,
                                                              int n_prefix, int bz, int L, float* __restrict__ dscores) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= bz * 2 * L) return;
  const int b = idx / (2 * L), rem = idx - b * 2 * L, side = rem / L, i = rem - side * L;
  float g = 0.f;
  for (int p = 0; p < n_prefix; ++p)
    if (i < prefix_lens[p]) {
      const float tm = teacher_pos[(size_t)p * bz + b] - teacher_neg[(size_t)p * bz + b];
      g += 2.0f / (float)bz * (margins[(size_t)p * bz + b] - tm);
    }
  dscores[idx] = side == 0 ? g : -g;
}
hipError_t launch_margin_mse_bwd(const float* margins, const float* teacher_pos, const float* teacher_neg, const int32_t* prefix_lens,
                                 int n_prefix, int bz, int L, float* dscores, hipStream_t s) {
  hipLaunchKernelGGL(margin_mse_bwd_kernel, dim3((bz * 2 * L + 255) / 256), dim3(256), 0, s, margins, teacher_pos, teacher_neg,
                     prefix_lens, n_prefix, bz, L, dscores);
  return hipGetLastError();
}

// score_r = <hF_r, E_r>, hF = post * w * x * rs. Given dscore_r: dE_r = dscore * hF (scattered into the output codebook
// gradient by the caller: dE rows written to de[r]), dhF = dscore * E_r -> dh[r] (the caller runs rmsnorm_bwd on it).
__global__ __launch_bounds__(256) void gold_score_bwd_kernel(const float* __restrict__ x, const float* __restrict__ ln,
                                                              const float* __restrict__ out_embeds, const int32_t* __restrict__ out_idx,
                                                              const float* __restrict__ dscores, float* __restrict__ dh,
                                                              float* __restrict__ de, int rows, int d, float eps, float post) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * d);
  const float4* wr = reinterpret_cast<const float4*>(ln);
  const float4* er = reinterpret_cast<const float4*>(out_embeds + (size_t)out_idx[row] * d);
  const int n4 = d >> 2;
  float ss = 0.f;
  for (int k = lane; k < n4; k += 64) { const float4 v = xr[k]; ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
  ss = wave_sum_f(ss);
  const float rs = rsqrtf(ss / (float)d + eps), g = dscores[row];
  for (int k = lane; k < n4; k += 64) {
    const float4 v = xr[k], w = wr[k], e = er[k];
    reinterpret_cast<float4*>(de + (size_t)row * d)[k] = make_float4(g * post * w.x * (v.x * rs), g * post * w.y * (v.y * rs),
                                                                     g * post * w.z * (v.z * rs), g * post * w.w * (v.w * rs));
    reinterpret_cast<float4*>(dh + (size_t)row * d)[k] = make_float4(g * e.x, g * e.y, g * e.z, g * e.w);
  }
}
hipError_t launch_gold_score_bwd(const float* x, const float* ln, const float* out_embeds, const int32_t* out_idx, const float* dscores,
                                 float* dh, float* de, int rows, int d, float eps, float post, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(gold_score_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ln, out_embeds, out_idx, dscores, dh, de, rows, d,
                     eps, post);
  return hipGetLastError();
}

// ---- optimizer ---------------------------------------------------------------------------------------------------------
// sum of squares of a flat buffer: block partials in double, then one block adds them in order (deterministic)
__global__ __launch_bounds__(256) void sumsq_part_kernel(const float* __restrict__ g, size_t n, double* __restrict__ part) {
  __shared__ double red[256];
  // 16-byte loads (the flat gradient buffer is 16-byte aligned), four independent double accumulators per thread; the
  // n % 4 trailing elements go to the last thread of the grid. (One 4-byte load per iteration ran at 2.2 TB/s.)
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const size_t n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = g4[i];
    a0 += (double)v.x * (double)v.x; a1 += (double)v.y * (double)v.y; a2 += (double)v.z * (double)v.z; a3 += (double)v.w * (double)v.w;
  }
  double a = (a0 + a1) + (a2 + a3);
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255)
    for (size_t i = n4 << 2; i < n; ++i) a += (double)g[i] * (double)g[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void sumsq_final_kernel(const double* __restrict__ part, int n, float max_norm, float* __restrict__ out /*[2]: norm, clip*/) {
  double a = 0.0;
  for (int i = 0; i < n; ++i) a += part[i];
  const double nrm = sqrt(a);
  out[0] = (float)nrm;
  // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
  const double c = max_norm > 0.f ? (double)max_norm / (nrm + 1e-6) : 1.0;
  out[1] = (float)(c < 1.0 ? c : 1.0);
}
hipError_t launch_grad_norm(const float* g, size_t n, double* part, int nparts, float max_norm, float* out, hipStream_t s) {
  hipLaunchKernelGGL(sumsq_part_kernel, dim3(nparts), dim3(256), 0, s, g, n, part);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(1), 0, s, part, nparts, max_norm, out);
  return hipGetLastError();
}

// torch.optim.AdamW (decoupled weight decay, no amsgrad), gradient scaled by the clip coefficient on the device:
//   p *= 1 - lr * wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, size_t n, const float* __restrict__ clip, float lr,
                                                     float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * clip[1];
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  float pi = p[i] * (1.f - lr * wd);
  pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
  p[i] = pi;
}
hipError_t launch_adamw(float* p, const float* g, float* m, float* v, size_t n, const float* clip, float lr, float b1, float b2,
                        float eps, float wd, float bc1, float bc2_sqrt, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, clip, lr, b1, b2, eps, wd, bc1,
                     bc2_sqrt);
  return hipGetLastError();
}

// The same update for every parameter tensor in ONE launch (189 tensors of t5-base: 189 launches of 9 us before). A block
// owns one chunk of 4096 consecutive elements of one tensor: block -> tensor by binary search over the prefix sums of the
// tensors' chunk counts; gradients and moments live at the tensor's offset of the flat buffers.
__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamSeg* __restrict__ segs, const int* __restrict__ pref, int nseg,
                                                           const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                           const float* __restrict__ clip, float lr, float b1, float b2, float eps,
                                                           float wd, float bc1, float bc2_sqrt) {
  const int b = blockIdx.x;
  int lo = 0, hi = nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (pref[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const AdamSeg sg = segs[lo];
  const size_t c0 = (size_t)(b - pref[lo]) * 4096;
  const float decay = 1.f - lr * (sg.decay ? wd : 0.f), cl = clip[1], step = lr / bc1;
  float* p = sg.p;
  const float* gs = g + sg.off;
  float* ms = m + sg.off;
  float* vs = v + sg.off;
  auto upd = [&](float pi, float gi, float& mi, float& vi) {
    gi *= cl;
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    pi = pi * decay;
    pi -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
    return pi;
  };
  if (((sg.off | sg.n) & 3) == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t i = c0 + ((size_t)k * 256 + threadIdx.x) * 4;
      if (i >= sg.n) continue;
      float4 pv = *reinterpret_cast<float4*>(p + i), mv = *reinterpret_cast<float4*>(ms + i), vv = *reinterpret_cast<float4*>(vs + i);
      const float4 gv = *reinterpret_cast<const float4*>(gs + i);
      pv.x = upd(pv.x, gv.x, mv.x, vv.x); pv.y = upd(pv.y, gv.y, mv.y, vv.y);
      pv.z = upd(pv.z, gv.z, mv.z, vv.z); pv.w = upd(pv.w, gv.w, mv.w, vv.w);
      *reinterpret_cast<float4*>(p + i) = pv; *reinterpret_cast<float4*>(ms + i) = mv; *reinterpret_cast<float4*>(vs + i) = vv;
    }
  } else {
    for (int k = 0; k < 16; ++k) {
      const size_t i = c0 + (size_t)k * 256 + threadIdx.x;
      if (i >= sg.n) continue;
      float mi = ms[i], vi = vs[i];
      p[i] = upd(p[i], gs[i], mi, vi);
      ms[i] = mi; vs[i] = vi;
    }
  }
}
hipError_t launch_adamw_multi(const AdamSeg* segs, const int* pref, int nseg, int nchunks, const float* g, float* m, float* v,
                              const float* clip, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                              hipStream_t s) {
  if (nseg <= 0 || nchunks <= 0) return hipSuccess;
  hipLaunchKernelGGL(adamw_multi_kernel, dim3(nchunks), dim3(256), 0, s, segs, pref, nseg, g, m, v, clip, lr, b1, b2, eps, wd, bc1,
                     bc2_sqrt);
  return hipGetLastError();
}

// ---- per-tensor dynamic f16 planes for the training GEMMs ---------------------------------------------------------------
// Absolute maxima of up to two tensors in one launch (blockIdx.y picks the tensor). The result is combined with an integer
// atomicMax on the bit pattern (non-negative floats order like unsigned integers; a maximum does not depend on the order
// of arrival), so the slots must be zero before the launch — the caller hands out fresh slots of a ring zeroed per step.
__global__ __launch_bounds__(256) void absmax2_kernel(const float* __restrict__ x0, size_t n0, const float* __restrict__ x1, size_t n1,
                                                       unsigned int* __restrict__ out) {
  __shared__ float red[4];
  const float* x = blockIdx.y ? x1 : x0;
  const size_t n = blockIdx.y ? n1 : n0;
  if ((size_t)blockIdx.x * 1024 >= n) return;
  float a = 0.f;
  const size_t stride = (size_t)gridDim.x * 1024;
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  auto amax4 = [](const float4& v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); };
  for (; i + 3 * stride + 3 < n; i += 4 * stride) {      // four independent loads in flight per lane
    const float4 v0 = *reinterpret_cast<const float4*>(x + i), v1 = *reinterpret_cast<const float4*>(x + i + stride);
    const float4 v2 = *reinterpret_cast<const float4*>(x + i + 2 * stride), v3 = *reinterpret_cast<const float4*>(x + i + 3 * stride);
    a = fmaxf(a, fmaxf(fmaxf(amax4(v0), amax4(v1)), fmaxf(amax4(v2), amax4(v3))));
  }
  for (; i < n; i += stride) {
    if (i + 3 < n) a = fmaxf(a, amax4(*reinterpret_cast<const float4*>(x + i)));
    else for (size_t j = i; j < n; ++j) a = fmaxf(a, fabsf(x[j]));
  }
  for (int m = 32; m > 0; m >>= 1) a = fmaxf(a, __shfl_xor(a, m));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out + blockIdx.y, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}
hipError_t launch_absmax2(const float* x0, size_t n0, const float* x1, size_t n1, float* out, hipStream_t s) {
  const size_t n = std::max(n0, x1 ? n1 : 0);
  if (n == 0) return hipSuccess;
  const int nb = (int)std::min<size_t>(2048, (n + 8191) / 8192);   // >= 8 float4 per thread before another block pays off
  hipLaunchKernelGGL(absmax2_kernel, dim3(nb, x1 ? 2 : 1), dim3(256), 0, s, x0, n0, x1, x1 ? n1 : 0,
                     reinterpret_cast<unsigned int*>(out));
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void split_dyn_kernel(const float* __restrict__ x, int R, int C, int ldi, __half* __restrict__ out,
                                                         const float* __restrict__ amax) {
  const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;          // float4 index over R * C / 4
  const int c4n = C >> 2;
  if (i4 >= (size_t)R * c4n) return;
  const int r = (int)(i4 / c4n), c = (int)(i4 - (size_t)r * c4n) * 4;
  const float sc = dyn_plane_scale(*amax);
  const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * ldi + c);
  __half h[4], l[4];
  split_f16(v.x * sc, h[0], l[0]); split_f16(v.y * sc, h[1], l[1]); split_f16(v.z * sc, h[2], l[2]); split_f16(v.w * sc, h[3], l[3]);
  const size_t o = (size_t)r * C + c;
  *reinterpret_cast<uint2*>(out + o) = *reinterpret_cast<uint2*>(h);
  *reinterpret_cast<uint2*>(out + (size_t)R * C + o) = *reinterpret_cast<uint2*>(l);
}
hipError_t launch_split_dyn(const float* x, int R, int C, int ldi, __half* out, const float* amax, hipStream_t s) {
  if (R <= 0 || C <= 0) return hipSuccess;
  if (C & 3) return hipErrorInvalidValue;
  const size_t n4 = (size_t)R * (C >> 2);
  hipLaunchKernelGGL(split_dyn_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, R, C, ldi, out, amax);
  return hipGetLastError();
}

// 64 x 64 tile through LDS: transposed planes out_t[2][C][Rpad] (pairs of rows as one half2 store), and, when out_p is
// given, the plain planes out_p[2][R][C] from the same read of x
__global__ __launch_bounds__(256) void split_dyn_T_kernel(const float* __restrict__ x, int R, int C, int ldi, int Rpad,
                                                           __half* __restrict__ out_t, __half* __restrict__ out_p,
                                                           const float* __restrict__ amax) {
  __shared__ float tile[64][65];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64, tid = threadIdx.x;
  const float sc = dyn_plane_scale(*amax);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = (tid >> 4) + 16 * k, c = (tid & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < R && c0 + c < C) {          // C % 4 == 0: a float4 is inside or outside as a whole
      v = *reinterpret_cast<const float4*>(x + (size_t)(r0 + r) * ldi + c0 + c);
      v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
      if (out_p) {
        __half h[4], l[4];
        split_f16(v.x, h[0], l[0]); split_f16(v.y, h[1], l[1]); split_f16(v.z, h[2], l[2]); split_f16(v.w, h[3], l[3]);
        const size_t o = (size_t)(r0 + r) * C + c0 + c;
        *reinterpret_cast<uint2*>(out_p + o) = *reinterpret_cast<uint2*>(h);
        *reinterpret_cast<uint2*>(out_p + (size_t)R * C + o) = *reinterpret_cast<uint2*>(l);
      }
    }
    tile[r][c] = v.x; tile[r][c + 1] = v.y; tile[r][c + 2] = v.z; tile[r][c + 3] = v.w;
  }
  __syncthreads();
  const size_t ps = (size_t)C * Rpad;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = (tid >> 5) + 8 * k, r = (tid & 31) * 2;        // Rpad % 2 == 0
    if (c0 + c < C && r0 + r < Rpad) {
      __half2 hi, lo;
      __half h0, l0, h1, l1;
      split_f16(tile[r][c], h0, l0); split_f16(tile[r + 1][c], h1, l1);
      hi = __halves2half2(h0, h1); lo = __halves2half2(l0, l1);
      const size_t o = (size_t)(c0 + c) * Rpad + r0 + r;
      *reinterpret_cast<__half2*>(out_t + o) = hi;
      *reinterpret_cast<__half2*>(out_t + ps + o) = lo;
    }
  }
}
hipError_t launch_split_dyn_T(const float* x, int R, int C, int ldi, int Rpad, __half* out_t, __half* out_plain, const float* amax,
                              hipStream_t s) {
  if (R <= 0 || C <= 0) return hipSuccess;
  if ((C & 3) || (ldi & 3) || (Rpad & 1)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(split_dyn_T_kernel, dim3((C + 63) / 64, (Rpad + 63) / 64), dim3(256), 0, s, x, R, C, ldi, Rpad, out_t, out_plain,
                     amax);
  return hipGetLastError();
}

// ---- bf16 operands for the training GEMMs (RPR_PREC_BF16): one conversion pass per operand, no maxima, no second plane --
__global__ __launch_bounds__(256) void to_bf16_kernel(const float* __restrict__ x, int R, int C, int ldi, __bf16* __restrict__ out) {
  const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;          // float4 index over R * C / 4
  const int c4n = C >> 2;
  if (i4 >= (size_t)R * c4n) return;
  const int r = (int)(i4 / c4n), c = (int)(i4 - (size_t)r * c4n) * 4;
  const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * ldi + c);
  __bf16 b[4] = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  *reinterpret_cast<uint2*>(out + (size_t)r * C + c) = *reinterpret_cast<uint2*>(b);
}
hipError_t launch_to_bf16(const float* x, int R, int C, int ldi, void* out, hipStream_t s) {
  if (R <= 0 || C <= 0) return hipSuccess;
  if ((C & 3) || (ldi & 3)) return hipErrorInvalidValue;
  const size_t n4 = (size_t)R * (C >> 2);
  hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, R, C, ldi, reinterpret_cast<__bf16*>(out));
  return hipGetLastError();
}

// 64 x 64 tile through LDS (as split_dyn_T_kernel): transposed out_t[C][Rpad] and, when given, the plain out_p[R][C].
// RELU: x is the gradient w.r.t. a ReLU output and act (same shape and row stride) the stored activation: elements whose
// activation is not positive convert as zero — the ReLU backward of the feed-forward block folded into the conversion of
// its result (the fp32 gradient itself is not read again in bf16 mode), one pass over x less per feed-forward block.
template <bool RELU>
__global__ __launch_bounds__(256) void to_bf16_T_kernel(const float* __restrict__ x, int R, int C, int ldi, int Rpad, int ldt,
                                                         __bf16* __restrict__ out_t, __bf16* __restrict__ out_p,
                                                         const float* __restrict__ act) {
  __shared__ float tile[64][65];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64, tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = (tid >> 4) + 16 * k, c = (tid & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < R && c0 + c < C) {
      v = *reinterpret_cast<const float4*>(x + (size_t)(r0 + r) * ldi + c0 + c);
      if (RELU) {
        const float4 a = *reinterpret_cast<const float4*>(act + (size_t)(r0 + r) * ldi + c0 + c);
        v.x = a.x > 0.f ? v.x : 0.f; v.y = a.y > 0.f ? v.y : 0.f; v.z = a.z > 0.f ? v.z : 0.f; v.w = a.w > 0.f ? v.w : 0.f;
      }
      if (out_p) {
        __bf16 b[4] = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        *reinterpret_cast<uint2*>(out_p + (size_t)(r0 + r) * C + c0 + c) = *reinterpret_cast<uint2*>(b);
      }
    }
    tile[r][c] = v.x; tile[r][c + 1] = v.y; tile[r][c + 2] = v.z; tile[r][c + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = (tid >> 5) + 8 * k, r = (tid & 31) * 2;        // Rpad % 2 == 0
    if (c0 + c < C && r0 + r < Rpad) {
      __bf16 b[2] = {(__bf16)tile[r][c], (__bf16)tile[r + 1][c]};
      *reinterpret_cast<unsigned int*>(out_t + (size_t)(c0 + c) * ldt + r0 + r) = *reinterpret_cast<unsigned int*>(b);
    }
  }
}
// ldt: row stride of out_t in elements (0 = Rpad); see ldT() in train_api.hip.
hipError_t launch_to_bf16_T(const float* x, int R, int C, int ldi, int Rpad, void* out_t, void* out_plain, hipStream_t s,
                            const float* relu_act, int ldt) {
  if (R <= 0 || C <= 0) return hipSuccess;
  if (ldt == 0) ldt = Rpad;
  if ((C & 3) || (ldi & 3) || (Rpad & 1) || (ldt & 1) || ldt < Rpad) return hipErrorInvalidValue;
  const dim3 grid((C + 63) / 64, (Rpad + 63) / 64);
  if (relu_act)
    hipLaunchKernelGGL(to_bf16_T_kernel<true>, grid, dim3(256), 0, s, x, R, C, ldi, Rpad, ldt, reinterpret_cast<__bf16*>(out_t),
                       reinterpret_cast<__bf16*>(out_plain), relu_act);
  else
    hipLaunchKernelGGL(to_bf16_T_kernel<false>, grid, dim3(256), 0, s, x, R, C, ldi, Rpad, ldt, reinterpret_cast<__bf16*>(out_t),
                       reinterpret_cast<__bf16*>(out_plain), relu_act);
  return hipGetLastError();
}

// RMSNorm of the training forward in bf16 mode, written straight in the two operand formats of the layer's products: the
// plain bf16 rows [rows][d] (A operand of the forward GEMM) and the transposed copy [d][ldt] (X^T operand of the weight
// gradient), columns rows .. Rpad of the latter zero. Replaces rmsnorm_kernel (fp32 out) + to_bf16_T_kernel (fp32 in): one
// read of x instead of a write and a read of the normalised fp32 rows in between. A block owns 32 rows (8 waves x 4 rows,
// all of a wave's loads requested up front); the transposed copy goes through a bf16 tile in LDS and leaves as 64-byte
// row pieces (two columns per thread). Arithmetic as rmsnorm_kernel: w * (x * rsqrt(mean(x^2) + eps)) (* post). d <= 1024.
__global__ __launch_bounds__(512) void rmsnorm_bf16_T_kernel(const float* __restrict__ x, const float* __restrict__ w, int rows, int d,
                                                              float eps, float post, __bf16* __restrict__ out_p,
                                                              __bf16* __restrict__ out_t, int Rpad, int ldt) {
  extern __shared__ __attribute__((aligned(16))) __bf16 nt_tile[];   // [32][d + 4]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n4 = d >> 2, tld = d + 4;
  const int r0 = blockIdx.x * 32;
  constexpr int NV = 4, NR = 4;
  const float4* wr = reinterpret_cast<const float4*>(w);
  float4 xv[NR][NV];
#pragma unroll
  for (int rr = 0; rr < NR; ++rr) {
    const int row = r0 + wave * NR + rr;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)(row < rows ? row : 0) * d);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = lane + 64 * k;
      const float4 lx = xr[i < n4 ? i : 0];   // unconditional (clamped) load + select: see rmsnorm_bwd_kernel
      xv[rr][k] = (row < rows && i < n4) ? lx : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  float4 wv[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) { const int i = lane + 64 * k; const float4 lw = wr[i < n4 ? i : 0]; wv[k] = i < n4 ? lw : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
  for (int rr = 0; rr < NR; ++rr) {
    const int lr = wave * NR + rr, row = r0 + lr;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) { const float4 v = xv[rr][k]; ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
    ss = wave_sum_f(ss);
    const float rs = rsqrtf(ss / (float)d + eps);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = lane + 64 * k;
      if (i < n4) {
        const float4 v = xv[rr][k], g = wv[k];
        float4 o = make_float4(g.x * (v.x * rs), g.y * (v.y * rs), g.z * (v.z * rs), g.w * (v.w * rs));
        if (post != 1.0f) { o.x *= post; o.y *= post; o.z *= post; o.w *= post; }
        if (row >= rows) o = make_float4(0.f, 0.f, 0.f, 0.f);
        __bf16 b[4] = {(__bf16)o.x, (__bf16)o.y, (__bf16)o.z, (__bf16)o.w};
        if (row < rows) *reinterpret_cast<uint2*>(out_p + (size_t)row * d + 4 * i) = *reinterpret_cast<uint2*>(b);
        *reinterpret_cast<uint2*>(nt_tile + lr * tld + 4 * i) = *reinterpret_cast<uint2*>(b);
      }
    }
  }
  __syncthreads();
  // columns (c, c + 1) of the tile -> 32 consecutive row entries of out_t[c] and out_t[c + 1]
  for (int c = 2 * threadIdx.x; c < d; c += 2 * 512) {
    unsigned int lo[16], hi[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const unsigned int a = *reinterpret_cast<const unsigned int*>(nt_tile + (2 * p) * tld + c);
      const unsigned int b = *reinterpret_cast<const unsigned int*>(nt_tile + (2 * p + 1) * tld + c);
      lo[p] = (a & 0xffffu) | (b << 16);       // rows 2p, 2p + 1 of column c
      hi[p] = (a >> 16) | (b & 0xffff0000u);   // ... of column c + 1
    }
    uint4* d0 = reinterpret_cast<uint4*>(out_t + (size_t)c * ldt + r0);
    uint4* d1 = reinterpret_cast<uint4*>(out_t + (size_t)(c + 1) * ldt + r0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      d0[q] = make_uint4(lo[4 * q], lo[4 * q + 1], lo[4 * q + 2], lo[4 * q + 3]);
      d1[q] = make_uint4(hi[4 * q], hi[4 * q + 1], hi[4 * q + 2], hi[4 * q + 3]);
    }
  }
}
hipError_t launch_rmsnorm_bf16_T(const float* x, const float* w, int rows, int d, float eps, float post, void* out_plain, void* out_t,
                                 int Rpad, int ldt, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (d > 1024 || (d & 7) || (Rpad & 31) || (ldt & 7) || ldt < Rpad || Rpad < rows) return hipErrorInvalidValue;
  const size_t smem = (size_t)32 * (d + 4) * sizeof(__bf16);
  hipLaunchKernelGGL(rmsnorm_bf16_T_kernel, dim3(Rpad / 32), dim3(512), smem, s, x, w, rows, d, eps, post,
                     reinterpret_cast<__bf16*>(out_plain), reinterpret_cast<__bf16*>(out_t), Rpad, ldt);
  return hipGetLastError();
}

// Every GEMM weight of the model in one launch: plain bf16 copy [N][K] (forward operand) and transposed copy [K][N]
// (operand of the input-gradient product), both at the tensor's offset in the flat parameter layout. Block -> tensor by
// binary search over the prefix sums of the tensors' 64 x 64 tile counts.
__global__ __launch_bounds__(256) void weights_bf16_kernel(const WSeg* __restrict__ segs, const int* __restrict__ pref, int nseg,
                                                            __bf16* __restrict__ plain, __bf16* __restrict__ tr) {
  __shared__ float tile[64][65];
  const int b = blockIdx.x, tid = threadIdx.x;
  int lo = 0, hi = nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (pref[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const WSeg sg = segs[lo];
  const int t = b - pref[lo], tc = (sg.C + 63) >> 6;
  const int c0 = (t % tc) * 64, r0 = (t / tc) * 64, R = sg.R, C = sg.C;
  __bf16* out_p = plain + sg.off;
  __bf16* out_t = tr + sg.off;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = (tid >> 4) + 16 * k, c = (tid & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < R && c0 + c < C) {
      v = *reinterpret_cast<const float4*>(sg.src + (size_t)(r0 + r) * C + c0 + c);
      __bf16 q[4] = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
      *reinterpret_cast<uint2*>(out_p + (size_t)(r0 + r) * C + c0 + c) = *reinterpret_cast<uint2*>(q);
    }
    tile[r][c] = v.x; tile[r][c + 1] = v.y; tile[r][c + 2] = v.z; tile[r][c + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = (tid >> 5) + 8 * k, r = (tid & 31) * 2;        // R % 2 == 0
    if (c0 + c < C && r0 + r < R) {
      __bf16 q[2] = {(__bf16)tile[r][c], (__bf16)tile[r + 1][c]};
      *reinterpret_cast<unsigned int*>(out_t + (size_t)(c0 + c) * R + r0 + r) = *reinterpret_cast<unsigned int*>(q);
    }
  }
}
hipError_t launch_weights_bf16(const WSeg* segs, const int* pref, int nseg, int ntiles, void* plain, void* tr, hipStream_t s) {
  if (nseg <= 0 || ntiles <= 0) return hipSuccess;
  hipLaunchKernelGGL(weights_bf16_kernel, dim3(ntiles), dim3(256), 0, s, segs, pref, nseg, reinterpret_cast<__bf16*>(plain),
                     reinterpret_cast<__bf16*>(tr));
  return hipGetLastError();
}

hipError_t init_train_kernel_attributes() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(self_attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(rmsnorm_bf16_T_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(cross_attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace rpr
