// Kernels of the teacher-forced forward of the prefix-oriented ranking fine-tune step:
// reference T5SeqAQEncoderForLngKnpMarginMSE.forward, t5_pretrainer/modeling/t5_generative_retriever.py:902-966.
// The decoder runs over all L positions of every (query, smtid) pair at once (rows = bz * n_docs * L), reusing the
// projection GEMMs, the block attention kernels (causal variant of enc_attn_kernel; cross-attention with the 2L rows
// of a query as its "beams") and the fused RMSNorm of the search path. What is specific to training lives here:
// the teacher-forced input embeddings, the gold-code scores and the margin-MSE losses.
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
    if (lane == 0) xo.ssq[row] = ssq_to_fix(ss);
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

}  // namespace rpr
