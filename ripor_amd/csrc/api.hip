// C-ABI of libripor_hip.so (see include/ripor_hip.h): context, model binding, trie, and the
// orchestration of one constrained beam search = T5 encoder once + L KV-cached decoder steps, each
// fused with the trie mask / top-B / beam expand, all enqueued on one HIP stream and replayed as a
// hipGraph (no host synchronisation inside the search; the reference syncs >= 1 + 2*B*Q times per
// step).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <tuple>
#include <vector>

#include "internal.h"

namespace rpr {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  g_err = std::string("HIP error ") + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ") in " + what + " at " +
          file + ":" + std::to_string(line);
  (void)hipGetLastError();
  return e == hipErrorOutOfMemory ? RPR_ERR_OOM : RPR_ERR_HIP;
}

// HF T5Attention._relative_position_bucket in float32 (log in float32, truncation toward zero);
// rel = key_pos - query_pos. Pinned against the torch expression in tests/test_host_logic.py.
int rel_bucket(int rel, int bidirectional, int num_buckets, int max_distance) {
  int bucket = 0, n;
  if (bidirectional) {
    num_buckets /= 2;
    if (rel > 0) bucket += num_buckets;
    n = rel < 0 ? -rel : rel;
  } else {
    n = rel < 0 ? -rel : 0;
  }
  const int max_exact = num_buckets / 2;
  if (n < max_exact) return bucket + n;
  const float v = logf((float)n / (float)max_exact) / (float)log((double)max_distance / (double)max_exact) *
                  (float)(num_buckets - max_exact);
  int large = max_exact + (int)v;
  if (large > num_buckets - 1) large = num_buckets - 1;
  return bucket + large;
}

// |logit| <= sqrt(d_model) * max_row |E_out[r] * ln_final| * scaleup factor for ANY decoder state (Cauchy-Schwarz on the
// RMS-normalised hidden state): the bound behind the forced-tail fork's masked-candidate proof (enqueue_fork).
int compute_logit_bound(rpr_ctx* c, rpr_model* m, hipStream_t s, float* out) {
  const auto& d = m->d;
  float* slot = reinterpret_cast<float*>(c->status + 9);   // device word next to the weight-range probe
  RPR_HIP(hipMemsetAsync(slot, 0, 4, s));
  RPR_HIP(launch_max_row_norm(d.out_embeds, d.dec_final_ln, d.L * d.V, d.d_model, slot, s));
  float mx = 0.f;
  RPR_HIP(hipMemcpyAsync(&mx, slot, 4, hipMemcpyDeviceToHost, s));
  RPR_HIP(hipStreamSynchronize(s));
  const float post = d.scaleup_output_hidden ? (float)pow((double)d.d_model, -0.5) : 1.0f;
  *out = 1.001f * mx * sqrtf((float)d.d_model) * post + 1.0f;   // slack for the split-precision arithmetic
  return RPR_OK;
}

int refresh_weight_planes(rpr_ctx* c, rpr_model* m, hipStream_t s) {
  // the weight-range probe has its own device word (status[8]): the ctx's sticky flags (status[0..3]) may hold something
  // nobody has read yet — searches on another model, a forward enqueued before the optimizer step
  unsigned int* probe = c->status + 8;
  RPR_HIP(hipMemsetAsync(probe, 0, 4, s));
  for (const auto& j : m->plane_jobs)
    RPR_HIP(launch_split_planes(j.w, j.dst, j.n, j.plane_stride ? j.plane_stride : j.n, s, W_PLANE_SCALE * j.pre, j.ln, m->d.d_model, probe));
  unsigned int sat = 0;
  RPR_HIP(hipMemcpyAsync(&sat, probe, 4, hipMemcpyDeviceToHost, s));
  RPR_HIP(hipStreamSynchronize(s));
  m->f32_only = sat != 0;
  // the weights changed (optimizer step, or a caller writing through rpr_param_info's pointers): the logit bound of the
  // forced-tail proof follows them, and so do the graphs that hold the spread limit derived from it by value
  float lb = m->logit_bound;
  const int e = compute_logit_bound(c, m, s, &lb);
  if (e) return e;
  if (lb != m->logit_bound) {
    m->logit_bound = lb;
    for (auto it = c->graphs.begin(); it != c->graphs.end();) {
      if (it->first.m == m) { (void)hipGraphExecDestroy(it->second); it = c->graphs.erase(it); } else ++it;
    }
  }
  return RPR_OK;
}

}  // namespace rpr

using namespace rpr;

namespace {

// One linear layer C = act(A @ W^T) (+ residual). A and W are given in both representations; the
// ctx precision picks the exact fp32 MFMA kernel or the f16x2 split kernel.
struct LinIn {                                                               // activation [M, K]
  const float* f; const __half* h; size_t ps; int ld; float scale = A_PLANE_SCALE;   //   fp32 / planes (+ their scale)
  const unsigned long long* ssq = nullptr; float inv_d_fix = 0.f, eps = 0.f;         //   fused RMSNorm: h = planes of x, W folded
};
struct LinW { const float* f; const __half* h; int N, K; };                  // weight [N, K] (+ planes, stride N*K)
struct LinOut {                                                              // destination
  float* f[3]; int ldo[3]; int split_n;                                      //   fp32 (up to 3 column blocks)
  __half* h; size_t ps; int ldh;                                             //   or f16 planes (next GEMM's input)
  const float* resid; int relu;
  int rm_B; size_t rm_stride, rm_slot, rm_head;                              //   KV-cache element map (common.h)
  int rm_dshift;                                                             //   log2(d_kv) of the map (0 = 6)
  float plane_scale;                                                         //   scale of the planes written to h (0 = 1)
  const __half* resid_h; unsigned long long* ssq_out;                        //   fused RMSNorm producer: residual read from the
};                                                                           //   planes h (in place), row sums accumulated

LinOut out_f32(float* p, int ld, int N, const float* resid = nullptr, int relu = 0) {
  LinOut o{};
  o.f[0] = o.f[1] = o.f[2] = p; o.ldo[0] = o.ldo[1] = o.ldo[2] = ld; o.split_n = N; o.resid = resid; o.relu = relu;
  return o;
}

// m_dev (nullable): device-side live row count (packed encoder); m_acc = rows to account flops/bytes for
void linear(Launcher& L, const LinIn& A, const LinW& W, int M, const LinOut& O, const int* m_dev = nullptr, int m_acc = -1) {
  const double Ma = m_acc >= 0 ? m_acc : M;
  const double fl = 2.0 * Ma * (double)W.N * W.K;
  const double by = 4.0 * (Ma * W.K + (double)W.N * W.K + Ma * W.N * ((O.resid || O.resid_h) ? 2 : 1));
  hipStream_t s = L.s;
  if (L.c->precision == RPR_PREC_F16X2) {
    GemmH2Args g{};
    g.A = A.h; g.a_ps = A.ps; g.lda = A.ld; g.W = W.h; g.w_ps = (size_t)W.N * W.K; g.ldw = W.K;
    g.resid = O.resid; g.ldr = O.ldo[0];
    for (int i = 0; i < 3; ++i) { g.out[i] = O.f[i]; g.ldo[i] = O.ldo[i]; }
    g.split_n = O.split_n; g.out_h = O.h; g.o_ps = O.ps; g.ldoh = O.ldh;
    g.M = M; g.N = W.N; g.K = W.K; g.relu = O.relu;
    g.trace = L.c->trace_buf;
    g.rm_B = O.rm_B; g.rm_stride = O.rm_stride; g.rm_slot = O.rm_slot; g.rm_head = O.rm_head; g.rm_dshift = O.rm_dshift;
    g.m_dev = m_dev; g.acc_scale = 1.0f / (W_PLANE_SCALE * A.scale); g.plane_scale = O.plane_scale;
    g.row_ssq = A.ssq; g.inv_d_fix = A.inv_d_fix; g.eps = A.eps;
    g.resid_h = O.resid_h; g.r_ps = O.ps; g.ldrh = O.ldh; g.ssq_out = O.ssq_out;
    // timing ablations (results are wrong with any bit set): 1 = no row-sum atomics, 2 = no consumer row scale
    static const int dbg = [] { const char* e = dev_getenv("RPR_DEBUG_FUSED"); return e ? atoi(e) : 0; }();
    if (dbg & 1) g.ssq_out = nullptr;
    if (dbg & 2) g.row_ssq = nullptr;
    g.sat = L.c->status;
    g.cus = L.c->cur_cus;
    g.small_live = m_dev ? L.c->cur_small_live : 0;
    g.no_row_split = L.c->cur_no_row_split;
    g.part = P<float>(L.c->ws.part); g.part_cap = L.c->ws.part.cap / sizeof(float); g.mid_split = 1;
    L.run(RPR_K_GEMM, fl, by, [&] { return launch_gemm_h2(g, s); }, &g.kernel_cls);
  } else {
    GemmArgs g{};
    g.A = A.f; g.lda = A.ld; g.W = W.f; g.ldw = W.K; g.resid = O.resid; g.ldr = O.ldo[0];
    for (int i = 0; i < 3; ++i) { g.out[i] = O.f[i]; g.ldo[i] = O.ldo[i]; }
    g.split_n = O.split_n; g.M = M; g.N = W.N; g.K = W.K; g.relu = O.relu;
    g.rm_B = O.rm_B; g.rm_stride = O.rm_stride; g.rm_slot = O.rm_slot; g.rm_head = O.rm_head; g.rm_dshift = O.rm_dshift;
    g.m_dev = m_dev;
    L.run(RPR_K_GEMM, fl, by, [&] { return launch_gemm(g, s); });
  }
}

int flush_profile(rpr_ctx* c) {
  std::map<const int*, int> live;   // device counters of the pass, read once each after the events have completed
  for (auto& r : c->recs) {
    RPR_HIP(hipEventSynchronize(r.b));
    float ms = 0.f;
    RPR_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    double scale = 1.0;
    if (r.live_dev && r.live_static > 0) {
      auto it = live.find(r.live_dev);
      if (it == live.end()) {
        int n = r.live_static;
        RPR_HIP(hipMemcpy(&n, r.live_dev, sizeof(int), hipMemcpyDeviceToHost));
        it = live.emplace(r.live_dev, n).first;
      }
      scale = (double)std::min(std::max(it->second, 0), r.live_static) / (double)r.live_static;
    }
    auto& d = c->done[r.cls];
    d.total_ms += ms; d.launches += 1; d.flops += r.flops * scale; d.bytes += r.bytes * scale;
    c->pool.push_back(r.a); c->pool.push_back(r.b);
  }
  c->recs.clear();
  return 0;
}

// forks: depths at which forced queries leave the sequential steps (ascending, each in [1, L-1]; empty = plain search)
// drop_last: no stage after the last fork (optimistic mode, see choose_forks): its caches are not needed
int alloc_workspace(rpr_ctx* c, const rpr_model* m, int Q, int Lq, int B, int L, const std::vector<int>& forks = {},
                    bool drop_last = false, bool log_softmax = false) {
  const auto& d = m->d;
  const size_t T = (size_t)Q * Lq, R = (size_t)Q * B, inner = m->inner(), dm = d.d_model, dff = d.d_ff;
  const size_t nd = d.num_decoder_layers, ne = d.num_layers, f = sizeof(float);
  Workspace& w = c->ws;
  int e = 0;
  auto E = [&](DevBuf& b, size_t bytes) { if (!e) e = ensure(c, b, bytes); };
  E(w.ids, T * 4); E(w.mask, T * 4); E(w.last, (size_t)Q * 4); E(w.offs, ((size_t)Q + 1) * 4); E(w.row_src, T * 4);
  E(w.ex, T * dm * f); E(w.eh, T * dm * f); E(w.eqkv, T * 3 * inner * f); E(w.eattn, T * inner * f);
  E(w.eff, T * dff * f); E(w.enc_out, T * dm * f); E(w.xkv, T * nd * 2 * inner * f);
  E(w.x, R * dm * f); E(w.h, R * dm * f); E(w.q, R * inner * f); E(w.attn, R * inner * f);
  E(w.ff, R * dff * f); E(w.logits, R * (size_t)m->Vp() * f);
  const size_t depth0 = forks.empty() ? (size_t)L : (size_t)forks[0];   // stage 0 stops at the first fork
  E(w.kcache, nd * depth0 * R * inner * f); E(w.vcache, nd * depth0 * R * inner * f);
  E(w.lb, R * (size_t)m->Vp() * 4 * 2);   // start and end of every child's row range
  if (select_radix_wanted(B, m->Vp())) E(w.sel_rs, select_radix_ws_bytes(Q, B, m->Vp()));
  for (int i = 0; i < 2; ++i) {
    E(w.score[i], R * 8); E(w.lo[i], R * 4); E(w.hi[i], R * 4);
    E(w.tokens[i], R * (size_t)L * 2); E(w.anc[i], R * (size_t)L * 2);
  }
  E(w.o_tokens, R * (size_t)L * 4); E(w.o_scores, R * 4); E(w.o_lo, R * 8); E(w.o_hi, R * 8);
  const size_t hb = sizeof(__half) * 2;  // two planes
  E(w.eattn_h, T * inner * hb); E(w.eff_h, T * dff * hb); E(w.enc_out_h, T * dm * hb);
  E(w.attn_h, R * inner * hb); E(w.ff_h, R * dff * hb);
  E(w.ex_h, T * dm * hb); E(w.x_h, R * dm * hb);
  E(w.ssq_e, (2 * ne + 1) * T * 8); E(w.ssq_d, (3 * nd + 1) * R * 8);
  E(w.part, (size_t)9 << 20 << 2);   // split-K partials of the mid-size GEMM route: < 256 tiles of 128 x 64, up to 4 splits
  // forced-tail search: one compacted stage and one tail job per fork, tail activations for the longest tail
  for (size_t k = 0; k < forks.size(); ++k) {
    const size_t depth = k + 1 < forks.size() ? (size_t)forks[k + 1] : (size_t)L, Lt = (size_t)(L - forks[k]);
    StageBufs& sb = w.stage[k];
    E(sb.cnt, 16); E(sb.src, (size_t)Q * 4);
    if (!(drop_last && k + 1 == forks.size())) {
      E(sb.qmap, (size_t)Q * 4); E(sb.offs, (size_t)Q * 4); E(sb.last, (size_t)Q * 4); E(sb.mask, T * 4);
      E(sb.kcache, nd * depth * R * inner * f); E(sb.vcache, nd * depth * R * inner * f);
      for (int i = 0; i < 2; ++i) {
        E(sb.score[i], R * 8); E(sb.lo[i], R * 4); E(sb.hi[i], R * 4);
        E(sb.tokens[i], R * (size_t)L * 2); E(sb.anc[i], R * (size_t)L * 2);
      }
    }
    TailBufs& tb = w.tail[k];
    E(tb.flag, (size_t)Q * 4); E(tb.flist, (size_t)Q * 4); E(tb.cnt, 16);
    E(tb.qmap, (size_t)Q * 4); E(tb.offs, (size_t)Q * 4); E(tb.last, (size_t)Q * 4); E(tb.mask, T * 4);
    E(tb.tokens, R * (size_t)L * 2); E(tb.gold, R * Lt * f);
  }
  if (!forks.empty()) {
    const size_t Rt = R * (size_t)(L - forks[0]);
    E(w.t_qkv, Rt * 3 * inner * f); E(w.t_q, Rt * inner * f);
    if (c->precision == RPR_PREC_F16X2 && !m->f32_only) {
      E(w.t_x_h, Rt * dm * hb); E(w.t_attn_h, Rt * inner * hb); E(w.t_ff_h, Rt * dff * hb); E(w.t_ssq, (3 * nd + 1) * Rt * 8);
    } else {
      E(w.t_x, Rt * dm * f); E(w.t_h, Rt * dm * f); E(w.t_attn, Rt * inner * f); E(w.t_ff, Rt * dff * f);
    }
    if (log_softmax) { E(w.t_h, Rt * dm * f); E(w.t_logits, Rt * (size_t)d.V * f); }   // normalised rows, V logits per row
  }
  return e;
}

// Fused RMSNorm plumbing of the split-precision mode: the residual stream x lives in two f16 planes
// (hi + lo = 22 bits; operand of the next projection AND residual of the next producer) plus one fixed-point sum of
// squares per row and norm site; the projection that follows a norm runs on the x planes against W * diag(ln_weight)
// and scales its output rows by rsqrt(ssq / d + eps). Sites are numbered in program order; every site of a pass has
// its own ssq slot, zeroed by one small kernel per pass (site 0 is stored by the embedding kernel, the others are
// accumulated with integer atomics by the residual GEMMs' epilogues, so the sums do not depend on the order of
// arrival). The exact-fp32 mode keeps the fp32 stream and the separate RMSNorm kernel.
struct XStream {
  __half* x_h; size_t ps; unsigned long long* ssq; size_t rows; int dm; float eps;
  LinIn in(int site) const {
    LinIn a{nullptr, x_h, ps, dm, X_PLANE_SCALE};
    a.ssq = ssq + (size_t)site * rows; a.inv_d_fix = 1.0f / ((float)dm * SSQ_FIX); a.eps = eps;
    return a;
  }
  LinOut out(int site) const {          // x += projection (in place in the planes); the site's row sums
    LinOut o{};
    o.split_n = dm; o.h = x_h; o.ps = ps; o.ldh = dm; o.plane_scale = X_PLANE_SCALE;
    o.resid_h = x_h; o.ssq_out = ssq + (size_t)site * rows;
    return o;
  }
};

// Encoder forward into ws.enc_out (reference generation.py:132-137 -> model.encoder(...)).
// packed = false: rows are [Q, Lq] padded (taps / rpr_encode return that layout).
// packed = true (the search path): only the positions before each query's last attended token exist, as rows
//   offs[q] .. offs[q] + last[q] - 1 (ws.offs / ws.last / ws.row_src, launch_pack_rows). The row count is only
//   known on the device, so every launch keeps its padded grid (hipGraph-safe) and tiles / rows past offs[Q] exit.
//   Padded positions are exp(-inf) keys and unused query rows in the padded layout, so results are identical.
void enqueue_encoder(Launcher& Ln, rpr_ctx* c, const rpr_model* m, int Q, int Lq, bool packed) {
  const auto& d = m->d;
  Workspace& w = c->ws;
  const int T = Q * Lq, inner = m->inner(), dm = d.d_model, dff = d.d_ff;
  const bool h2 = c->precision == RPR_PREC_F16X2;
  hipStream_t s = Ln.s;
  float *x = P<float>(w.ex), *h = P<float>(w.eh), *qkv = P<float>(w.eqkv), *attn = P<float>(w.eattn),
        *ff = P<float>(w.eff);
  __half *attn_h = P<__half>(w.eattn_h), *ff_h = P<__half>(w.eff_h);
  const size_t ps_d = (size_t)T * dm, ps_i = (size_t)T * inner, ps_f = (size_t)T * dff;
  const float eps = d.layer_norm_eps;
  const int32_t* offs = packed ? P<int32_t>(w.offs) : nullptr;
  const int* live = packed ? P<int>(w.offs) + Q : nullptr;   // device-side number of live rows
  int Ta = T;                                                  // rows accounted in the profile (flops / bytes)
  if (packed) {
    Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_pack_rows(P<int32_t>(w.last), P<int32_t>(w.offs), P<int32_t>(w.row_src), Q, Lq, s); });
  }
  Ln.account_live(live, T);   // profile pass: flops / bytes below are stated for T rows and scaled by the live count at flush
  const XStream xs{P<__half>(w.ex_h), ps_d, P<unsigned long long>(w.ssq_e), (size_t)T, dm, eps};
  if (h2) Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_zero_u64(P<unsigned long long>(w.ssq_e), (size_t)(2 * d.num_layers + 1) * T, s); });
  auto norm = [&](const float* wgt) {   // exact-fp32 mode only: the split mode folds the norms into the projections
    Ln.run(RPR_K_RMSNORM, 0, 2.0 * Ta * dm * 4, [&] { return launch_rmsnorm(x, wgt, h, T, dm, eps, s, 1.0f, nullptr, 0, live); });
  };
  Ln.run(RPR_K_OTHER, 0, 2.0 * Ta * dm * 4, [&] {
    return launch_embed_rows(d.shared, P<int32_t>(w.ids), x, T, dm, d.vocab_size, s,
                             packed ? P<int32_t>(w.row_src) : nullptr, live,
                             h2 ? XOut{xs.x_h, ps_d, xs.ssq, c->status} : XOut{});
  });
  const LinIn in_attn{attn, attn_h, ps_i, inner}, in_ff{ff, ff_h, ps_f, dff, FF_PLANE_SCALE};
  for (int i = 0; i < d.num_layers; ++i) {
    if (!h2) norm(m->enc_ln0[i]);
    linear(Ln, h2 ? xs.in(2 * i) : LinIn{h, nullptr, 0, dm}, {m->enc_qkv[i], m->h_enc_qkv[i], 3 * inner, dm}, T,
           out_f32(qkv, 3 * inner, 3 * inner), live, Ta);
    EncAttnArgs a{qkv, P<int32_t>(w.mask), d.enc_rel_bias, m->enc_bucket, attn, Q, Lq, d.num_heads, d.rel_buckets,
                  h2 ? attn_h : nullptr, ps_i, offs, P<int32_t>(w.last), c->status, 0};
    a.dkv = d.d_kv;
    Ln.run(RPR_K_ENC_ATTN, 4.0 * Q * d.num_heads * (double)Lq * Lq * d.d_kv * ((double)Ta / T) * ((double)Ta / T), 4.0 * Ta * 4 * inner,
           [&] { return launch_enc_attn(a, s); });
    linear(Ln, in_attn, {m->enc_o[i], m->h_enc_o[i], dm, inner}, T, h2 ? xs.out(2 * i + 1) : out_f32(x, dm, dm, x), live, Ta);
    if (!h2) norm(m->enc_ln1[i]);
    LinOut o = out_f32(ff, dff, dff, nullptr, 1);
    if (h2) { o.h = ff_h; o.ps = ps_f; o.ldh = dff; o.plane_scale = FF_PLANE_SCALE; }
    linear(Ln, h2 ? xs.in(2 * i + 1) : LinIn{h, nullptr, 0, dm}, {m->enc_wi[i], m->h_enc_wi[i], dff, dm}, T, o, live, Ta);
    linear(Ln, in_ff, {m->enc_wo[i], m->h_enc_wo[i], dm, dff}, T, h2 ? xs.out(2 * i + 2) : out_f32(x, dm, dm, x), live, Ta);
  }
  // final norm: fp32 copy always (taps / rpr_encode), planes for the cross-K/V GEMM in split mode
  Ln.run(RPR_K_RMSNORM, 0, 2.0 * Ta * dm * 4, [&] {
    return launch_rmsnorm(x, d.enc_final_ln, P<float>(w.enc_out), T, dm, eps, s, 1.0f,
                          h2 ? P<__half>(w.enc_out_h) : nullptr, ps_d, live, c->status, h2 ? xs.x_h : nullptr, ps_d);
  });
  c->enc_rows_accounted = Ta;
  Ln.account_live(nullptr, 0);
}

BeamState beam_state(DevBuf (&score)[2], DevBuf (&lo)[2], DevBuf (&hi)[2], DevBuf (&tokens)[2], DevBuf (&anc)[2], int i, int L) {
  BeamState st;
  st.score = P<double>(score[i]); st.lo = P<int32_t>(lo[i]); st.hi = P<int32_t>(hi[i]);
  st.tokens = P<uint16_t>(tokens[i]); st.anc = P<uint16_t>(anc[i]); st.ld = L;
  return st;
}

// A batch of queries stepping through the decoder one position at a time: stage 0 = all queries of the call, later
// stages = the queries left over by a fork, compacted (live counts on the device, static launch geometry).
struct StageView {
  int Qcap;                        // query capacity = grid size of every launch
  const int* nq_dev;               // live queries / live rows (queries x beams) on the device; null = Qcap (stage 0)
  const int* nrows_dev;
  StageIO io;                      // qmap (null = identity), first encoder row (null = q * Lq), attended length, mask rows
  float* kcache; float* vcache;    // [nd][Qcap][H][depth][B][64] fp32: everything one (query, head) can touch is one
  int depth;                       //   contiguous depth*B*256-B region and the B rows of a position are adjacent
  BeamState st[2];                 // ping-pong by step parity
  size_t kv_q(int B, int inner) const { return (size_t)depth * B * inner; }
  int dkv = DKV;                   // head dim of the caches (64; 128 = t5-3b, which runs without forks)
  size_t kv_h(int B) const { return (size_t)depth * B * dkv; }
  size_t kv_layer(int B, int inner) const { return (size_t)Qcap * depth * B * inner; }
};

// Profile accounting of a compacted stage / tail job: the launches that follow state their flops and bytes for the
// static capacity `rows`; the record keeps the device counter and flush_profile scales by live / rows afterwards. Nothing
// is read back while the step is being enqueued (a synchronisation here would run the two lanes one after the other).
int live_count(Launcher& Ln, const int* dev, int rows) {
  Ln.account_live(dev, rows);
  return rows;
}

struct SearchDims { int Q, Lq, B, L, xld; bool packed; unsigned flags; };

// Decoder steps [t0, t1) of one stage: embed, nd x {self-attention over the beam's ancestry, cross-attention, FF},
// logits of position t, fused trie mask / top-B / beam expand (reference generation.py:423-526, one iteration per step).
void enqueue_steps(Launcher& Ln, rpr_ctx* c, const rpr_model* m, const rpr_trie* tr, const SearchDims& sd, const StageView& sv,
                   int t0, int t1, bool shared0, const rpr_debug_taps* taps, unsigned long long* sel_clk) {
  const auto& d = m->d;
  Workspace& w = c->ws;
  const int Q = sv.Qcap, B = sd.B, L = sd.L, Lq = sd.Lq, R = Q * B, inner = m->inner(), dm = d.d_model, dff = d.d_ff, H = d.num_heads;
  const int nd = d.num_decoder_layers, V = d.V, Vp = m->Vp(), xld = sd.xld;   // Vp: logits row / selection width (V padded to 64)
  const bool h2 = c->precision == RPR_PREC_F16X2;
  const float eps = d.layer_norm_eps;
  hipStream_t s = Ln.s;
  float *x = P<float>(w.x), *h = P<float>(w.h), *qb = P<float>(w.q), *attn = P<float>(w.attn), *ff = P<float>(w.ff),
        *logits = P<float>(w.logits);
  __half *attn_h = P<__half>(w.attn_h), *ff_h = P<__half>(w.ff_h);
  const size_t ps_d = (size_t)R * dm, ps_i = (size_t)R * inner, ps_f = (size_t)R * dff;
  const size_t layer_stride = sv.kv_layer(B, inner), kv_q = sv.kv_q(B, inner), kv_h = sv.kv_h(B), kv_pos = (size_t)B * sv.dkv, kv_slot = sv.dkv;
  int Rt = R, Bt = B;   // rows / beams per query of the current step's decoder pass
  const int Racc = live_count(Ln, sv.nrows_dev, R);   // rows the profile accounts for
  const float post = d.scaleup_output_hidden ? (float)pow((double)dm, -0.5) : 1.0f;
  auto norm = [&](const float* wgt, float post_scale = 1.0f) {   // exact-fp32 mode only (see XStream)
    Ln.run(RPR_K_RMSNORM, 0, 2.0 * Racc * dm * 4, [&] { return launch_rmsnorm(x, wgt, h, Rt, dm, eps, s, post_scale, nullptr, 0, sv.nrows_dev); });
  };
  const XStream xs{P<__half>(w.x_h), ps_d, P<unsigned long long>(w.ssq_d), (size_t)R, dm, eps};
  const LinIn in_h{h, nullptr, 0, dm}, in_attn{attn, attn_h, ps_i, inner}, in_ff{ff, ff_h, ps_f, dff, FF_PLANE_SCALE};
  if (Vp != V && !h2)   // exact-fp32 logits GEMM writes the V real columns of a row only: the padding must read as finite
    // (a kernel node: memset nodes captured into the search graph did not re-execute reliably on replay, see launch_zero_u64)
    Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_zero_u64(reinterpret_cast<unsigned long long*>(logits), (size_t)R * Vp / 2, s); });
  for (int t = t0; t < t1; ++t) {
    const BeamState cur = sv.st[t & 1], nxt = sv.st[(t + 1) & 1];
    Bt = (t == 0 && shared0) ? 1 : B; Rt = Q * Bt;
    const int Ma = (Bt == B) ? Racc : Rt;
    if (h2) Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_zero_u64(P<unsigned long long>(w.ssq_d), (size_t)(3 * nd + 1) * R, s); });
    Ln.run(RPR_K_OTHER, 0, 2.0 * Ma * dm * 4, [&] {
      return launch_dec_embed(d.start_embed, d.in_embeds, cur.tokens, L, x, Rt, dm, V, t, s,
                              h2 ? XOut{xs.x_h, ps_d, xs.ssq, c->status} : XOut{}, sv.nrows_dev);
    });
    for (int i = 0; i < nd; ++i) {
      float* kc = sv.kcache + i * layer_stride;
      float* vc = sv.vcache + i * layer_stride;
      if (!h2) norm(m->dec_ln0[i]);
      {  // q -> qb, k/v -> cache row block of position t
        LinOut o{};
        o.f[0] = qb; o.f[1] = kc + (size_t)t * kv_pos; o.f[2] = vc + (size_t)t * kv_pos;
        o.ldo[0] = o.ldo[1] = o.ldo[2] = inner; o.split_n = inner;
        o.rm_B = Bt; o.rm_stride = kv_q; o.rm_slot = kv_slot; o.rm_head = kv_h; o.rm_dshift = sv.dkv == 128 ? 7 : 6;
        linear(Ln, h2 ? xs.in(3 * i) : in_h, {m->dec_qkv[i], m->h_dec_qkv[i], 3 * inner, dm}, Rt, o, sv.nrows_dev, Ma);
      }
      {
        DecSelfAttnArgs a{qb, kc, vc, kv_q, kv_h, kv_pos, kv_slot, cur.anc, L, d.dec_rel_bias, m->dec_bucket, attn, Q, Bt, H, t,
                          h2 ? attn_h : nullptr, ps_i, c->status, sv.nq_dev};
        a.dkv = d.d_kv;
        Ln.run(RPR_K_DEC_SELF_ATTN, 4.0 * Ma * H * (double)(t + 1) * d.d_kv,
               4.0 * ((double)Ma * inner * 2 + 2.0 * Ma * (double)(t + 1) * inner), [&] { return launch_dec_self_attn(a, s); });
      }
      linear(Ln, in_attn, {m->dec_o[i], m->h_dec_o[i], dm, inner}, Rt, h2 ? xs.out(3 * i + 1) : out_f32(x, dm, dm, x), sv.nrows_dev, Ma);
      if (!h2) norm(m->dec_ln1[i]);
      linear(Ln, h2 ? xs.in(3 * i + 1) : in_h, {m->dec_xq[i], m->h_dec_xq[i], inner, dm}, Rt, out_f32(qb, inner, inner), sv.nrows_dev, Ma);
      {
        const float* xk = P<float>(w.xkv) + (size_t)i * 2 * inner;
        DecCrossAttnArgs a{qb, xk, xk + inner, xld, sv.io.mask, attn, Q, Bt, H, Lq, h2 ? attn_h : nullptr, ps_i,
                           sv.io.last, sv.io.offs, 0, c->status, sv.nq_dev};
        a.dkv = d.d_kv;
        Ln.run(RPR_K_DEC_CROSS_ATTN, 4.0 * Ma * H * (double)Lq * d.d_kv,
               4.0 * ((double)Ma * inner * 2 + 2.0 * (Ma / Bt) * (double)Lq * inner), [&] { return launch_step_cross_attn(a, s); });
      }
      linear(Ln, in_attn, {m->dec_xo[i], m->h_dec_xo[i], dm, inner}, Rt, h2 ? xs.out(3 * i + 2) : out_f32(x, dm, dm, x), sv.nrows_dev, Ma);
      if (!h2) norm(m->dec_ln2[i]);
      LinOut o = out_f32(ff, dff, dff, nullptr, 1);
      if (h2) { o.h = ff_h; o.ps = ps_f; o.ldh = dff; o.plane_scale = FF_PLANE_SCALE; }
      linear(Ln, h2 ? xs.in(3 * i + 2) : in_h, {m->dec_wi[i], m->h_dec_wi[i], dff, dm}, Rt, o, sv.nrows_dev, Ma);
      linear(Ln, in_ff, {m->dec_wo[i], m->h_dec_wo[i], dm, dff}, Rt, h2 ? xs.out(3 * i + 3) : out_f32(x, dm, dm, x), sv.nrows_dev, Ma);
    }
    if (!h2) norm(d.dec_final_ln, post);
    // logits of position t only (the reference computes every position and keeps [-1])
    float* lg = (taps && taps->step_logits) ? taps->step_logits + (size_t)t * R * V : logits;   // taps: V == Vp (rpr_search)
    {
      LinW wt{d.out_embeds + (size_t)t * V * dm, nullptr, V, dm};
      if (h2) {
        // planes of codebook t (times the final layer-norm weight and the scaleup factor) inside the stacked
        // [2][L*V][d] buffer: plane stride is L*V*d
        const LinIn a = xs.in(3 * nd);
        GemmH2Args g{};
        g.A = a.h; g.a_ps = a.ps; g.lda = dm;
        g.W = m->h_out_embeds + (size_t)t * Vp * dm; g.w_ps = (size_t)d.L * Vp * dm; g.ldw = dm;
        g.out[0] = g.out[1] = g.out[2] = lg; g.ldo[0] = g.ldo[1] = g.ldo[2] = Vp; g.split_n = Vp;
        g.M = Rt; g.N = Vp; g.K = dm; g.acc_scale = 1.0f / (W_PLANE_SCALE * a.scale);
        g.row_ssq = a.ssq; g.inv_d_fix = a.inv_d_fix; g.eps = a.eps; g.sat = c->status;
        g.m_dev = sv.nrows_dev; g.cus = c->cur_cus; g.small_live = sv.nrows_dev ? c->cur_small_live : 0;
        Ln.run(RPR_K_GEMM, 2.0 * Ma * (double)V * dm, 4.0 * ((double)Ma * dm + (double)V * dm + (double)Ma * V),
               [&] { return launch_gemm_h2(g, s); }, &g.kernel_cls);
      } else {
        linear(Ln, in_h, wt, Rt, out_f32(lg, Vp, V), sv.nrows_dev, Ma);   // V real columns into rows of Vp (the pad stays 0)
      }
    }
    SelectArgs sa{};
    sa.logits = lg; sa.codes = tr->codes; sa.Lc = tr->L; sa.cur = cur; sa.nxt = nxt;
    sa.lb_scratch = P<int32_t>(w.lb); sa.Q = Q; sa.B = B; sa.V = Vp; sa.Vreal = V; sa.t = t;
    if (tr->lvl_V == tr->V) {
      sa.lvl0 = tr->lvl0; sa.lvl1 = tr->lvl1; sa.lvl_V = tr->lvl_V;
      sa.idx2 = tr->idx2; sa.n_deep = tr->n_deep;
      for (int i = 0; i < tr->n_deep; ++i) { sa.d_start[i] = tr->d_start[i]; sa.d_tok[i] = tr->d_tok[i]; sa.d_n[i] = tr->d_n[i]; }
    }
    if (w.sel_rs.p && select_radix_wanted(B, Vp) && w.sel_rs.cap >= select_radix_ws_bytes(Q, B, Vp))
      select_radix_carve(sa.rs, w.sel_rs.p, P<int32_t>(w.lb) + (size_t)R * Vp, Q, B, Vp);
    sa.log_softmax = (sd.flags & RPR_FLAG_LOG_SOFTMAX) ? 1 : 0;
    sa.shared0 = (Bt != B) ? 1 : 0;
    sa.nq_dev = sv.nq_dev;
    if (taps) {
      sa.tap_scores = taps->step_scores ? taps->step_scores + (size_t)t * R : nullptr;
      sa.tap_tokens = taps->step_tokens ? taps->step_tokens + (size_t)t * R : nullptr;
      sa.tap_parent = taps->step_parent ? taps->step_parent + (size_t)t * R : nullptr;
      sa.tap_valid = taps->step_valid ? reinterpret_cast<unsigned long long*>(taps->step_valid) + (size_t)t * ((size_t)R * V / 64) : nullptr;
    }
    if (sel_clk) sa.clk = sel_clk + (size_t)t * 8;
    Ln.run(RPR_K_SELECT, 0, (double)Ma * V * 4 + (double)Ma * 40, [&] { return launch_select(sa, s); });
  }
  Ln.account_live(nullptr, 0);   // the live counter of this stage scales THIS stage's records only (fork, tail, finalize follow)
}

// The fork after step T-1 of stage `sv`: which of its queries are forced (tail job `tb`), the others compacted into
// the next stage (`nb` / returned view): beam state, the K/V of the T positions walked so far, the cross-attention inputs.
// compact = false (optimistic mode, last fork): nobody walks on — a query that is not forced here only raises the ctx's
// sticky RPR_STATUS_TAIL_LEFTOVER word and the caller repeats the batch in the exact mode.
StageView enqueue_fork(Launcher& Ln, rpr_ctx* c, const rpr_model* m, const rpr_trie* tr, const SearchDims& sd, const StageView& sv,
                       int T, int next_depth, TailBufs& tb, StageBufs& nb, bool compact) {
  const auto& d = m->d;
  const int Q = sv.Qcap, B = sd.B, L = sd.L, Lq = sd.Lq, inner = m->inner(), nd = d.num_decoder_layers, H = d.num_heads;
  hipStream_t s = Ln.s;
  const BeamState st = sv.st[T & 1];
  // No masked candidate may overtake a valid one during the remaining n = L - T steps. With |logit| <= bound
  // (rpr_model::logit_bound) the valid candidates of a step are >= smin - n*bound and the masked ones
  // <= smax + n*bound - 1e9, so forced requires (smax - smin) + 2*n*bound < 1e9; a tenth of that is demanded.
  // Log-softmax scores: a step adds a log-probability in [-(2*bound + ln V), 0] instead of a logit in [-bound, bound].
  const double per_step = (sd.flags & RPR_FLAG_LOG_SOFTMAX) ? 2.0 * (double)m->logit_bound + log((double)d.V) : 2.0 * (double)m->logit_bound;
  const double spread_max = 1e8 - (double)(L - T) * per_step;
  ForkArgs fa{st, tr->codes, tr->L, Q, sv.nq_dev, B, T, L, spread_max, P<int32_t>(tb.flag)};
  Ln.run(RPR_K_FORK, 0, 0, [&] { return launch_fork_classify(fa, s); });
  Ln.run(RPR_K_FORK, 0, 0, [&] {
    return launch_fork_scan(P<int32_t>(tb.flag), Q, sv.nq_dev, B, L - T, P<int32_t>(tb.flist), P<int32_t>(tb.cnt), P<int32_t>(nb.src),
                            P<int32_t>(nb.cnt), s);
  });
  // tail job: per-query inputs and the full token rows of the forced beams
  Ln.run(RPR_K_FORK, 0, 0, [&] {
    return launch_gather_stage_io(sv.io, StageOut{P<int32_t>(tb.qmap), P<int32_t>(tb.offs), P<int32_t>(tb.last), P<int32_t>(tb.mask)},
                                  P<int32_t>(tb.flist), P<int>(tb.cnt), Q, Lq, s);
  });
  Ln.run(RPR_K_FORK, 0, 0, [&] {
    return launch_tail_tokens(st, tr->codes, tr->L, P<int32_t>(tb.flist), P<int>(tb.cnt), Q, B, T, L, P<uint16_t>(tb.tokens), s);
  });
  // next stage
  StageView nv{};
  if (!compact) {
    Ln.run(RPR_K_FORK, 0, 0, [&] { return launch_flag_nonzero(P<int>(nb.cnt), c->status + 2, s); });
    return nv;
  }
  nv.Qcap = Q; nv.nq_dev = P<int>(nb.cnt); nv.nrows_dev = P<int>(nb.cnt) + 1;
  nv.io = StageIO{P<int32_t>(nb.qmap), P<int32_t>(nb.offs), P<int32_t>(nb.last), P<int32_t>(nb.mask)};
  nv.kcache = P<float>(nb.kcache); nv.vcache = P<float>(nb.vcache); nv.depth = next_depth; nv.dkv = sv.dkv;
  for (int i = 0; i < 2; ++i) nv.st[i] = beam_state(nb.score, nb.lo, nb.hi, nb.tokens, nb.anc, i, L);
  Ln.run(RPR_K_FORK, 0, 0, [&] {
    return launch_gather_stage_io(sv.io, StageOut{P<int32_t>(nb.qmap), P<int32_t>(nb.offs), P<int32_t>(nb.last), P<int32_t>(nb.mask)},
                                  P<int32_t>(nb.src), nv.nq_dev, Q, Lq, s);
  });
  Ln.run(RPR_K_FORK, 0, 0, [&] { return launch_compact_beams(st, nv.st[T & 1], P<int32_t>(nb.src), nv.nq_dev, Q, B, T, s); });
  KvCopyArgs kc{sv.kcache, sv.vcache, nv.kcache, nv.vcache, sv.kv_layer(B, inner), sv.kv_q(B, inner), sv.kv_h(B),
                nv.kv_layer(B, inner), nv.kv_q(B, inner), nv.kv_h(B), P<int32_t>(nb.src), nv.nq_dev, Q, nd, H, T * B * sv.dkv};
  Ln.run(RPR_K_FORK, 0, 0, [&] { return launch_kv_copy(kc, s); });
  return nv;
}

// Tail pass of one fork: the remaining positions T..L-1 of every forced beam in ONE teacher-forced decoder pass
// (rows = forced queries x beams x (L - T), sequence-major), then the replay of the selection order and finalize.
// Same layer arithmetic as the sequential steps; self-attention reads the positions < T from the fork stage's KV cache
// through the beams' ancestry and the positions >= T from this pass's own K/V rows; the B*(L-T) rows of a query share
// its encoder K/V in cross-attention; instead of V logits per row only the logit of the row's (only valid) token is
// computed, in exact fp32 (tail_gold_kernel).
void enqueue_tail(Launcher& Ln, rpr_ctx* c, const rpr_model* m, const SearchDims& sd, const StageView& sv, int T, TailBufs& tb) {
  const auto& d = m->d;
  Workspace& w = c->ws;
  const int Q = sv.Qcap, B = sd.B, L = sd.L, Lq = sd.Lq, Lt = L - T, S = Q * B, R = S * Lt;
  const int inner = m->inner(), dm = d.d_model, dff = d.d_ff, H = d.num_heads, nd = d.num_decoder_layers, V = d.V, xld = sd.xld;
  const bool h2 = c->precision == RPR_PREC_F16X2;
  const float eps = d.layer_norm_eps;
  hipStream_t s = Ln.s;
  const int* nf_dev = P<int>(tb.cnt);
  const int* nseq_dev = nf_dev + 1;
  const int* nrows_dev = nf_dev + 2;
  const int Ra = live_count(Ln, nrows_dev, R);
  float *x = P<float>(w.t_x), *h = P<float>(w.t_h), *qkv = P<float>(w.t_qkv), *qb = P<float>(w.t_q), *attn = P<float>(w.t_attn),
        *ff = P<float>(w.t_ff);
  __half *attn_h = P<__half>(w.t_attn_h), *ff_h = P<__half>(w.t_ff_h);
  const size_t ps_d = (size_t)R * dm, ps_i = (size_t)R * inner, ps_f = (size_t)R * dff;
  const float post = d.scaleup_output_hidden ? (float)pow((double)dm, -0.5) : 1.0f;
  auto norm = [&](const float* wgt) {   // exact-fp32 mode only (see XStream)
    Ln.run(RPR_K_RMSNORM, 0, 2.0 * Ra * dm * 4, [&] { return launch_rmsnorm(x, wgt, h, R, dm, eps, s, 1.0f, nullptr, 0, nrows_dev); });
  };
  const XStream xs{P<__half>(w.t_x_h), ps_d, P<unsigned long long>(w.t_ssq), (size_t)R, dm, eps};
  const LinIn in_h{h, nullptr, 0, dm}, in_attn{attn, attn_h, ps_i, inner}, in_ff{ff, ff_h, ps_f, dff, FF_PLANE_SCALE};
  if (h2) Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_zero_u64(P<unsigned long long>(w.t_ssq), (size_t)(3 * nd + 1) * R, s); });
  Ln.run(RPR_K_OTHER, 0, 2.0 * Ra * dm * 4, [&] {
    return launch_tail_embed(d.in_embeds, P<uint16_t>(tb.tokens), x, R, nrows_dev, T, L, dm, V, s,
                             h2 ? XOut{xs.x_h, ps_d, xs.ssq, c->status} : XOut{});
  });
  const size_t kv_pos = (size_t)B * sv.dkv;
  for (int i = 0; i < nd; ++i) {
    if (!h2) norm(m->dec_ln0[i]);
    linear(Ln, h2 ? xs.in(3 * i) : in_h, {m->dec_qkv[i], m->h_dec_qkv[i], 3 * inner, dm}, R, out_f32(qkv, 3 * inner, 3 * inner), nrows_dev, Ra);
    {
      const size_t ls = sv.kv_layer(B, inner);
      TailSelfAttnArgs a{qkv, sv.kcache + i * ls, sv.vcache + i * ls, sv.kv_q(B, inner), sv.kv_h(B), kv_pos, (size_t)sv.dkv,
                         sv.st[T & 1].anc, L, P<int32_t>(tb.flist), nseq_dev, d.dec_rel_bias, m->dec_bucket, attn,
                         h2 ? attn_h : nullptr, ps_i, c->status, S, B, H, T, L};
      a.dkv = d.d_kv;
      Ln.run(RPR_K_TAIL_SELF_ATTN, 2.0 * Ra * H * (double)(L + T + 1) * sv.dkv, 4.0 * ((double)Ra * 4 * inner + 2.0 * (Ra / Lt) * (double)T * inner),
             [&] { return launch_tail_self_attn(a, s); });
    }
    linear(Ln, in_attn, {m->dec_o[i], m->h_dec_o[i], dm, inner}, R, h2 ? xs.out(3 * i + 1) : out_f32(x, dm, dm, x), nrows_dev, Ra);
    if (!h2) norm(m->dec_ln1[i]);
    linear(Ln, h2 ? xs.in(3 * i + 1) : in_h, {m->dec_xq[i], m->h_dec_xq[i], inner, dm}, R, out_f32(qb, inner, inner), nrows_dev, Ra);
    {
      const float* xk = P<float>(w.xkv) + (size_t)i * 2 * inner;
      DecCrossAttnArgs a{qb, xk, xk + inner, xld, P<int32_t>(tb.mask), attn, Q, B * Lt, H, Lq, h2 ? attn_h : nullptr, ps_i,
                         P<int32_t>(tb.last), P<int32_t>(tb.offs), 0, c->status, nf_dev};
      a.dkv = d.d_kv;
      Ln.run(RPR_K_DEC_CROSS_ATTN, 4.0 * Ra * H * (double)Lq * sv.dkv, 4.0 * ((double)Ra * inner * 2 + 2.0 * (Ra / (B * Lt)) * (double)Lq * inner),
             [&] { return launch_tail_cross_attn(a, s); });
    }
    linear(Ln, in_attn, {m->dec_xo[i], m->h_dec_xo[i], dm, inner}, R, h2 ? xs.out(3 * i + 2) : out_f32(x, dm, dm, x), nrows_dev, Ra);
    if (!h2) norm(m->dec_ln2[i]);
    LinOut o = out_f32(ff, dff, dff, nullptr, 1);
    if (h2) { o.h = ff_h; o.ps = ps_f; o.ldh = dff; o.plane_scale = FF_PLANE_SCALE; }
    linear(Ln, h2 ? xs.in(3 * i + 2) : in_h, {m->dec_wi[i], m->h_dec_wi[i], dff, dm}, R, o, nrows_dev, Ra);
    linear(Ln, in_ff, {m->dec_wo[i], m->h_dec_wo[i], dm, dff}, R, h2 ? xs.out(3 * i + 3) : out_f32(x, dm, dm, x), nrows_dev, Ra);
  }
  if (sd.flags & RPR_FLAG_LOG_SOFTMAX) {
    // the score of a position is the log-probability of its token: final RMSNorm of every row, the V logits of the rows
    // of one position per launch of the exact-fp32 GEMM (rows of a position are Lt apart; its codebook is out_embeds[p]),
    // then log_softmax at the token (tail_logprob_kernel)
    float* lg = P<float>(w.t_logits);
    Ln.run(RPR_K_RMSNORM, 0, 2.0 * Ra * dm * 4, [&] {
      return launch_rmsnorm(x, d.dec_final_ln, h, R, dm, eps, s, post, nullptr, 0, nrows_dev, nullptr, h2 ? xs.x_h : nullptr, ps_d);
    });
    for (int p = T; p < L; ++p) {
      GemmArgs g{};
      g.A = h + (size_t)(p - T) * dm; g.lda = Lt * dm;
      g.W = d.out_embeds + (size_t)p * V * dm; g.ldw = dm;
      for (int i = 0; i < 3; ++i) { g.out[i] = lg + (size_t)(p - T) * V; g.ldo[i] = Lt * V; }
      g.split_n = V; g.M = S; g.N = V; g.K = dm; g.m_dev = nseq_dev;
      Ln.run(RPR_K_GEMM_SMALL, 2.0 * (Ra / Lt) * (double)V * dm, 4.0 * ((double)(Ra / Lt) * (dm + V) + (double)V * dm),
             [&] { return launch_gemm(g, s); });
    }
    Ln.run(RPR_K_OTHER, 0, 4.0 * Ra * V, [&] {
      return launch_tail_logprob(lg, P<uint16_t>(tb.tokens), P<float>(tb.gold), R, nrows_dev, T, L, V, s);
    });
  } else {
    Ln.run(RPR_K_OTHER, 2.0 * Ra * dm, 4.0 * 2 * Ra * dm, [&] {
      return launch_tail_gold(x, d.dec_final_ln, d.out_embeds, P<uint16_t>(tb.tokens), P<float>(tb.gold), R, nrows_dev, T, L, dm, V, eps, post,
                              s, h2 ? xs.x_h : nullptr, ps_d);
    });
  }
  TailRankArgs ra{sv.st[T & 1], P<int32_t>(tb.flist), P<int32_t>(tb.qmap), nf_dev, P<uint16_t>(tb.tokens), P<float>(tb.gold), Q, B, T, L,
                  P<int32_t>(w.o_tokens), P<float>(w.o_scores), P<int64_t>(w.o_lo), P<int64_t>(w.o_hi)};
  Ln.run(RPR_K_FORK, 0, 0, [&] { return launch_tail_rank(ra, s); });
  Ln.account_live(nullptr, 0);
}

// Everything between the staged inputs (ws.ids/ws.mask) and the staged outputs (ws.o_*).
void enqueue_search(Launcher& Ln, rpr_ctx* c, const rpr_model* m, const rpr_trie* tr, int Q, int Lq, int B, int L,
                    unsigned flags, const rpr_debug_taps* taps, const std::vector<int>& forks, bool drop_last) {
  const auto& d = m->d;
  Workspace& w = c->ws;
  const int T = Q * Lq, inner = m->inner(), dm = d.d_model;
  const int nd = d.num_decoder_layers;
  hipStream_t s = Ln.s;
  // debug: RPR_SELECT_CLOCK=1 prints the phase durations of the selection kernel (eager launches only)
  unsigned long long* sel_clk = nullptr;
  {
    static const bool clk_env = [] { const char* e = dev_getenv("RPR_SELECT_CLOCK"); return e && atoi(e) != 0; }();
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (clk_env && hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone &&
        hipMalloc(&sel_clk, (size_t)L * 8 * sizeof(unsigned long long)) == hipSuccess)
      (void)hipMemset(sel_clk, 0, (size_t)L * 8 * sizeof(unsigned long long));
  }
  // index of the last attended key + 1 per query: row packing of the encoder and the cross-attention loop bound
  Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_mask_lengths(P<int32_t>(w.mask), P<int32_t>(w.last), Q, Lq, s, c->status + 1); });
  static const bool packed_env = [] { const char* e = dev_getenv("RPR_PACKED_ENCODER"); return !(e && atoi(e) == 0); }();
  const bool packed = packed_env && !taps;   // taps return the padded [Q, Lq, d] encoder output
  c->cur_no_row_split = packed ? 1 : 0;   // the packed rows' capacity says nothing about the live rows (reset below, after the cross-K/V product)
  enqueue_encoder(Ln, c, m, Q, Lq, packed);
  if (taps && taps->encoder_out && !Ln.err) {
    hipError_t e = hipMemcpyAsync(taps->encoder_out, w.enc_out.p, (size_t)T * dm * 4, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) { Ln.err = hip_fail(e, "tap copy", __FILE__, __LINE__); return; }
  }
  // cross-attention K/V of every decoder layer in one GEMM (shared by the B beams of a query;
  // the reference recomputes them for every beam at every step)
  const int xld = nd * 2 * inner;
  Ln.account_live(packed ? P<int>(w.offs) + Q : nullptr, T);
  linear(Ln, {P<float>(w.enc_out), P<__half>(w.enc_out_h), (size_t)T * dm, dm}, {d.dec_xkv, m->h_dec_xkv, xld, dm}, T,
         out_f32(P<float>(w.xkv), xld, xld), packed ? P<int>(w.offs) + Q : nullptr, c->enc_rows_accounted);
  Ln.account_live(nullptr, 0);
  c->cur_no_row_split = 0;

  const SearchDims sd{Q, Lq, B, L, xld, packed, flags};
  StageView sv{};
  sv.Qcap = Q;
  sv.io = StageIO{nullptr, packed ? P<int32_t>(w.offs) : nullptr, P<int32_t>(w.last), P<int32_t>(w.mask)};
  sv.kcache = P<float>(w.kcache); sv.vcache = P<float>(w.vcache); sv.depth = forks.empty() ? L : forks[0]; sv.dkv = d.d_kv;
  for (int i = 0; i < 2; ++i) sv.st[i] = beam_state(w.score, w.lo, w.hi, w.tokens, w.anc, i, L);
  Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_init_beams(sv.st[0], Q, B, tr->N, s); });
  if (w.sel_rs.p && select_radix_wanted(B, m->Vp()) && w.sel_rs.cap >= select_radix_ws_bytes(Q, B, m->Vp())) {
    RadixWs rs;   // the radix selection's histograms and counters start at zero (every step leaves them so)
    select_radix_carve(rs, w.sel_rs.p, nullptr, Q, B, m->Vp());
    Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_select_radix_reset(rs, Q, s); });
  }

  // Step 0: every beam of a query starts from the same start embedding and the same encoder states, so the
  // decoder pass is computed once per query (Q rows, "one beam") and select reads the shared logits row; the
  // position-0 K/V exist in slot 0 only and every beam's ancestry points there. (The reference recomputes
  // the B identical rows; beams 1..B-1 differ only by their -1e9 initial score, generation.py:418-420.)
  // Off when debug taps are requested (they expect [Q*B, V] logits per step) or RPR_STEP0_SHARED=0.
  static const bool step0_env = [] { const char* e = dev_getenv("RPR_STEP0_SHARED"); return !(e && atoi(e) == 0); }();
  const bool shared0 = step0_env && !taps && B > 1;
  // Stages: stage 0 (all queries) walks steps [0, forks[0]); at every fork the forced queries get their tail pass and
  // the others are compacted into the next stage, which walks on to the next fork (or to L); finalize ranks whoever is
  // still stepping at L. Without forks this is the plain loop of the reference.
  // Everything after the first fork works on what that fork left over — usually a handful of queries in buffers sized for
  // all of them: those GEMMs are enqueued as large-tile / small-tile pairs gated on the live count (GemmH2Args.small_live)
  struct SmallLive { rpr_ctx* c; ~SmallLive() { c->cur_small_live = 0; } } small_guard{c};
  static const int small_live_rows = [] { const char* e = dev_getenv("RPR_SMALL_LIVE"); return e ? atoi(e) : 1024; }();
  int t0 = 0;
  for (size_t k = 0; k <= forks.size(); ++k) {
    const int t1 = k < forks.size() ? forks[k] : L;
    c->cur_small_live = k >= 1 ? small_live_rows : 0;
    enqueue_steps(Ln, c, m, tr, sd, sv, t0, t1, shared0, taps, sel_clk);
    if (k < forks.size()) {
      const int next_depth = k + 1 < forks.size() ? forks[k + 1] : L;
      const bool last_dropped = drop_last && k + 1 == forks.size();
      const StageView nv = enqueue_fork(Ln, c, m, tr, sd, sv, t1, next_depth, w.tail[k], w.stage[k], !last_dropped);
      enqueue_tail(Ln, c, m, sd, sv, t1, w.tail[k]);
      if (last_dropped) return;   // every query was finished by a tail pass (or flagged)
      sv = nv;
    }
    t0 = t1;
  }
  if (sel_clk) {   // debug: phase durations of the selection kernel (block 0), 100 MHz wall clock
    (void)hipStreamSynchronize(s);
    std::vector<unsigned long long> hb((size_t)L * 8);
    (void)hipMemcpy(hb.data(), sel_clk, hb.size() * 8, hipMemcpyDeviceToHost);
    for (int t = 0; t < L; ++t) {
      fprintf(stderr, "[select t=%2d] us:", t);
      for (int k = 0; k < 6; ++k) fprintf(stderr, " %7.1f", (double)(hb[t * 8 + k + 1] - hb[t * 8 + k]) * 0.01);
      fprintf(stderr, "  rounds=%llu\n", hb[t * 8 + 7]);
    }
    (void)hipFree(sel_clk);
  }
  FinalizeArgs fa{sv.st[L & 1], sv.Qcap, B, L, P<int32_t>(w.o_tokens), P<float>(w.o_scores),
                  P<int64_t>(w.o_lo), P<int64_t>(w.o_hi), sv.nq_dev, sv.io.qmap};
  Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_finalize(fa, s); });
}

// Teacher-forced forward of the prefix-oriented ranking fine-tune step: reference
// T5SeqAQEncoderForLngKnpMarginMSE.forward (modeling/t5_generative_retriever.py:902-966). The encoder runs once per
// query (the positive and the negative example of a row carry the same query text, dataset/dataset.py:502-503: the
// reference encodes it twice); the decoder runs over all L positions of the n_docs smtids of every query at once:
// rows (q, doc, position) = bz * n_docs * L, causal block self-attention per (sequence, head), cross-attention with
// the n_docs * L rows of a query sharing its encoder K/V. Output: the gold-code score of every position.
void enqueue_train_forward(Launcher& Ln, rpr_ctx* c, const rpr_model* m, int bz, int Lq, int ndoc, int L,
                           const int32_t* codes /*[bz, ndoc, L]*/, float* pos_scores /*[bz, ndoc, L]*/) {
  const auto& d = m->d;
  Workspace& w = c->ws;
  const int T = bz * Lq, S = bz * ndoc, R = S * L, inner = m->inner(), dm = d.d_model, dff = d.d_ff, H = d.num_heads;
  const int nd = d.num_decoder_layers, V = d.V;
  const bool h2 = c->precision == RPR_PREC_F16X2;
  const float eps = d.layer_norm_eps;
  hipStream_t s = Ln.s;
  Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_mask_lengths(P<int32_t>(w.mask), P<int32_t>(w.last), bz, Lq, s, c->status + 1); });
  c->cur_no_row_split = 1;
  enqueue_encoder(Ln, c, m, bz, Lq, true);
  const int xld = nd * 2 * inner;
  linear(Ln, {P<float>(w.enc_out), P<__half>(w.enc_out_h), (size_t)T * dm, dm}, {d.dec_xkv, m->h_dec_xkv, xld, dm}, T,
         out_f32(P<float>(w.xkv), xld, xld), P<int>(w.offs) + bz, c->enc_rows_accounted);
  Ln.account_live(nullptr, 0);
  c->cur_no_row_split = 0;

  float *x = P<float>(w.x), *h = P<float>(w.h), *qkv = P<float>(w.tr_x), *qb = P<float>(w.q), *attn = P<float>(w.attn),
        *ff = P<float>(w.ff);
  __half *attn_h = P<__half>(w.attn_h), *ff_h = P<__half>(w.ff_h);
  const size_t ps_d = (size_t)R * dm, ps_i = (size_t)R * inner, ps_f = (size_t)R * dff;
  const float post = d.scaleup_output_hidden ? (float)pow((double)dm, -0.5) : 1.0f;
  auto norm = [&](const float* wgt) {   // exact-fp32 mode only (see XStream)
    Ln.run(RPR_K_RMSNORM, 0, 2.0 * R * dm * 4, [&] { return launch_rmsnorm(x, wgt, h, R, dm, eps, s); });
  };
  const XStream xs{P<__half>(w.x_h), ps_d, P<unsigned long long>(w.ssq_d), (size_t)R, dm, eps};
  const LinIn in_h{h, nullptr, 0, dm}, in_attn{attn, attn_h, ps_i, inner}, in_ff{ff, ff_h, ps_f, dff, FF_PLANE_SCALE};
  if (h2) Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_zero_u64(P<unsigned long long>(w.ssq_d), (size_t)(3 * nd + 1) * R, s); });
  Ln.run(RPR_K_OTHER, 0, 2.0 * R * dm * 4, [&] {
    return launch_train_dec_embed(d.start_embed, d.in_embeds, codes, x, S, L, dm, V, s,
                                  h2 ? XOut{xs.x_h, ps_d, xs.ssq, c->status} : XOut{});
  });
  for (int i = 0; i < nd; ++i) {
    if (!h2) norm(m->dec_ln0[i]);
    linear(Ln, h2 ? xs.in(3 * i) : in_h, {m->dec_qkv[i], m->h_dec_qkv[i], 3 * inner, dm}, R, out_f32(qkv, 3 * inner, 3 * inner));
    {  // causal self-attention of every sequence over its own L positions (decoder relative-position table)
      EncAttnArgs a{qkv, nullptr, d.dec_rel_bias, m->dec_bucket, attn, S, L, H, d.rel_buckets,
                    h2 ? attn_h : nullptr, ps_i, nullptr, nullptr, c->status, 1};
      Ln.run(RPR_K_ENC_ATTN, 2.0 * S * H * (double)L * L * DKV, 4.0 * R * 4 * inner, [&] { return launch_enc_attn(a, s); });
    }
    linear(Ln, in_attn, {m->dec_o[i], m->h_dec_o[i], dm, inner}, R, h2 ? xs.out(3 * i + 1) : out_f32(x, dm, dm, x));
    if (!h2) norm(m->dec_ln1[i]);
    linear(Ln, h2 ? xs.in(3 * i + 1) : in_h, {m->dec_xq[i], m->h_dec_xq[i], inner, dm}, R, out_f32(qb, inner, inner));
    {
      const float* xk = P<float>(w.xkv) + (size_t)i * 2 * inner;
      DecCrossAttnArgs a{qb, xk, xk + inner, xld, P<int32_t>(w.mask), attn, bz, ndoc * L, H, Lq, h2 ? attn_h : nullptr, ps_i,
                         P<int32_t>(w.last), P<int32_t>(w.offs), 0, c->status};
      Ln.run(RPR_K_DEC_CROSS_ATTN, 4.0 * R * H * (double)Lq * DKV, 4.0 * ((double)R * inner * 2 + 2.0 * bz * (double)Lq * inner),
             [&] { return launch_dec_cross_attn(a, s); });
    }
    linear(Ln, in_attn, {m->dec_xo[i], m->h_dec_xo[i], dm, inner}, R, h2 ? xs.out(3 * i + 2) : out_f32(x, dm, dm, x));
    if (!h2) norm(m->dec_ln2[i]);
    LinOut o = out_f32(ff, dff, dff, nullptr, 1);
    if (h2) { o.h = ff_h; o.ps = ps_f; o.ldh = dff; o.plane_scale = FF_PLANE_SCALE; }
    linear(Ln, h2 ? xs.in(3 * i + 2) : in_h, {m->dec_wi[i], m->h_dec_wi[i], dff, dm}, R, o);
    linear(Ln, in_ff, {m->dec_wo[i], m->h_dec_wo[i], dm, dff}, R, h2 ? xs.out(3 * i + 3) : out_f32(x, dm, dm, x));
  }
  // decoder_last_hidden_state (final RMSNorm, scaleup factor) dotted with the gold codes' OUTPUT codebook rows
  Ln.run(RPR_K_OTHER, 2.0 * R * dm, 4.0 * 2 * R * dm, [&] {
    return launch_gold_scores(x, d.dec_final_ln, d.out_embeds, codes, pos_scores, S, L, dm, V, eps, post, s,
                              h2 ? xs.x_h : nullptr, ps_d);
  });
}

int alloc_train_workspace(rpr_ctx* c, const rpr_model* m, int bz, int Lq, int ndoc, int L) {
  // the search workspace for bz queries with ndoc * L "beams" of one position covers every shared buffer
  int e = alloc_workspace(c, m, bz, Lq, ndoc * L, 1);
  if (e) return e;
  const size_t R = (size_t)bz * ndoc * L;
  e = ensure(c, c->ws.tr_x, R * 3 * m->inner() * sizeof(float));
  if (e) return e;
  return ensure(c, c->ws.tr_misc, R * sizeof(float) + R * sizeof(int32_t) + 4096);
}

}  // namespace

extern "C" {

int rpr_abi_version(void) { return 3; }
const char* rpr_last_error(void) { return g_err.c_str(); }

int rpr_rel_bucket(int rel, int bidirectional, int num_buckets, int max_distance) {
  return rel_bucket(rel, bidirectional, num_buckets, max_distance);
}

int rpr_init(int device, rpr_ctx** out_ctx) {
  RPR_REQUIRE(out_ctx != nullptr, "out_ctx is NULL");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    set_error("no HIP device visible: libripor_hip.so has no CPU fallback");
    return RPR_ERR_NO_DEVICE;
  }
  RPR_REQUIRE(device >= 0 && device < n, "device index out of range");
  RPR_HIP(hipSetDevice(device));
  RPR_HIP(init_t5_kernel_attributes());
  RPR_HIP(init_beam_kernel_attributes());
  RPR_HIP(init_train_kernel_attributes());
  RPR_HIP(init_tail_kernel_attributes());
  auto* c = new rpr_ctx();
  c->device = device;
  if (const char* e = getenv("RPR_PRECISION"))
    c->precision = (std::string(e) == "f32") ? RPR_PREC_F32 : (std::string(e) == "bf16") ? RPR_PREC_BF16 : RPR_PREC_F16X2;
  if (const char* e = getenv("RPR_LANE_MIN_ROWS")) c->lane_min_rows = atoi(e) > 0 ? atoi(e) : 0;
  if (const char* e = getenv("RPR_FORCED_TAIL")) c->forced_tail = atoi(e) < 0 ? 0 : (atoi(e) > 2 ? 2 : atoi(e));
  if (const char* e = getenv("RPR_FORK_DEPTHS")) {   // "4,6": explicit fork depths; "" or "0": none
    c->n_fork_override = 0;
    for (const char* p = e; *p && c->n_fork_override < MAX_FORKS;) {
      const int v = atoi(p);
      if (v > 0) c->fork_override[c->n_fork_override++] = v;
      while (*p && *p != ',') ++p;
      if (*p == ',') ++p;
    }
  }
  if (dev_getenv("RPR_GEMM_TRACE")) {
    void* p = nullptr;
    if (hipMalloc(&p, 2 << 20) == hipSuccess) { (void)hipMemset(p, 0, 2 << 20); c->trace_buf = (unsigned long long*)p; }
  }
  std::memset(c->done, 0, sizeof(c->done));
  hipError_t e = hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->status), 256);
  if (e == hipSuccess) e = hipMemset(c->status, 0, 256);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->status_host), 64, hipHostMallocDefault);
  if (e != hipSuccess) {
    if (c->status) (void)hipFree(c->status);
    if (c->cap_stream) (void)hipStreamDestroy(c->cap_stream);
    delete c;
    return hip_fail(e, "ctx setup", __FILE__, __LINE__);
  }
  *out_ctx = c;
  return RPR_OK;
}

void rpr_free_ctx(rpr_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.second);
  auto free_ws = [](Workspace& w) {   // a Workspace is nothing but DevBufs (static_assert in internal.h)
    DevBuf* all = reinterpret_cast<DevBuf*>(&w);
    for (size_t i = 0; i < sizeof(Workspace) / sizeof(DevBuf); ++i) if (all[i].p) (void)hipFree(all[i].p);
  };
  free_ws(c->ws);
  for (Lane& ln : c->lanes) {
    free_ws(ln.ws);
    if (ln.done) (void)hipEventDestroy(ln.done);
    if (ln.stream) (void)hipStreamDestroy(ln.stream);
  }
  if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
  free_train_ws(c);
  if (c->status) (void)hipFree(c->status);
  if (c->status_host) (void)hipHostFree(c->status_host);
  if (c->trace_buf) (void)hipFree(c->trace_buf);
  for (auto& r : c->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto e : c->pool) (void)hipEventDestroy(e);
  if (c->cap_stream) (void)hipStreamDestroy(c->cap_stream);
  delete c;
}

int64_t rpr_workspace_bytes(const rpr_ctx* c) { return c ? (int64_t)c->ws_bytes : 0; }

int rpr_set_precision(rpr_ctx* c, int precision) {
  RPR_REQUIRE(c, "NULL ctx");
  RPR_REQUIRE(precision == RPR_PREC_F32 || precision == RPR_PREC_F16X2 || precision == RPR_PREC_BF16, "unknown precision");
  c->precision = precision;
  return RPR_OK;
}
int rpr_get_precision(const rpr_ctx* c) { return c ? c->precision : -1; }

int rpr_load_model(rpr_ctx* c, const rpr_model_desc* d, rpr_model** out) {
  RPR_REQUIRE(c && d && out, "NULL argument");
  RPR_REQUIRE(d->d_kv == DKV || d->d_kv == 128, "d_kv must be 64 (t5-base / t5-large) or 128 (t5-3b)");
  RPR_REQUIRE(d->d_model % 32 == 0 && d->d_ff % 32 == 0, "d_model and d_ff must be multiples of 32");
  RPR_REQUIRE(d->V >= 2 && d->V <= 65536, "decoder vocab size out of range (2..65536)");
  RPR_REQUIRE(d->L >= 1 && d->L <= MAX_DEC_LEN, "decoder length out of range");
  RPR_REQUIRE(d->rel_buckets >= 2 && d->rel_buckets <= 64, "relative_attention_num_buckets out of range");
  RPR_REQUIRE(d->num_layers >= 1 && d->num_decoder_layers >= 1 && d->num_heads >= 1, "bad layer/head count");
  RPR_REQUIRE(d->shared && d->enc_rel_bias && d->dec_rel_bias && d->enc_final_ln && d->dec_final_ln &&
                  d->start_embed && d->in_embeds && d->out_embeds && d->dec_xkv, "NULL weight pointer");
  RPR_HIP(hipSetDevice(c->device));
  auto m = std::make_unique<rpr_model>();
  m->ctx = c;
  m->d = *d;
  auto copyv = [](std::vector<const float*>& v, const float* const* src, int n) -> bool {
    if (!src) return false;
    v.assign(src, src + n);
    for (auto p : v) if (!p) return false;
    return true;
  };
  bool ok = copyv(m->enc_ln0, d->enc_ln0, d->num_layers) && copyv(m->enc_qkv, d->enc_qkv, d->num_layers) &&
            copyv(m->enc_o, d->enc_o, d->num_layers) && copyv(m->enc_ln1, d->enc_ln1, d->num_layers) &&
            copyv(m->enc_wi, d->enc_wi, d->num_layers) && copyv(m->enc_wo, d->enc_wo, d->num_layers);
  const int nd = d->num_decoder_layers;
  ok = ok && copyv(m->dec_ln0, d->dec_ln0, nd) && copyv(m->dec_qkv, d->dec_qkv, nd) && copyv(m->dec_o, d->dec_o, nd) &&
       copyv(m->dec_ln1, d->dec_ln1, nd) && copyv(m->dec_xq, d->dec_xq, nd) && copyv(m->dec_xo, d->dec_xo, nd) &&
       copyv(m->dec_ln2, d->dec_ln2, nd) && copyv(m->dec_wi, d->dec_wi, nd) && copyv(m->dec_wo, d->dec_wo, nd);
  RPR_REQUIRE(ok, "NULL per-layer weight pointer");
  // the desc's host arrays need not outlive this call
  m->d.enc_ln0 = m->d.enc_qkv = m->d.enc_o = m->d.enc_ln1 = m->d.enc_wi = m->d.enc_wo = nullptr;
  m->d.dec_ln0 = m->d.dec_qkv = m->d.dec_o = m->d.dec_ln1 = m->d.dec_xq = m->d.dec_xo = nullptr;
  m->d.dec_ln2 = m->d.dec_wi = m->d.dec_wo = nullptr;
  std::vector<int32_t> eb(2 * MAX_LQ - 1), db(MAX_DEC_LEN);
  for (int rel = -(MAX_LQ - 1); rel <= MAX_LQ - 1; ++rel)
    eb[rel + MAX_LQ - 1] = rel_bucket(rel, 1, d->rel_buckets, d->rel_max_distance);
  for (int n = 0; n < MAX_DEC_LEN; ++n) db[n] = rel_bucket(-n, 0, d->rel_buckets, d->rel_max_distance);
  RPR_HIP(hipMalloc(&m->enc_bucket, eb.size() * 4));
  RPR_HIP(hipMalloc(&m->dec_bucket, db.size() * 4));
  RPR_HIP(hipMemcpy(m->enc_bucket, eb.data(), eb.size() * 4, hipMemcpyHostToDevice));
  RPR_HIP(hipMemcpy(m->dec_bucket, db.data(), db.size() * 4, hipMemcpyHostToDevice));
  // hi/lo f16 planes of every GEMM weight for the split-precision kernels
  {
    const size_t inner = (size_t)m->inner(), dm = d->d_model, dff = d->d_ff;
    int err = 0;
    unsigned int* probe = c->status + 8;   // weight-range probe word, apart from the sticky flags (refresh_weight_planes)
    RPR_HIP(hipMemset(probe, 0, 4));
    // ln (nullable): layer-norm weight folded into the columns (length = the projection's input dim, always d_model);
    // pre = extra scalar on the weights (scaleup_output_hidden on the codebooks)
    auto mk = [&](const float* wf, size_t n, __half** outp, const float* ln = nullptr, float pre = 1.0f) {
      if (err) return;
      void* p = nullptr;
      hipError_t e = hipMalloc(&p, n * 2 * sizeof(__half));
      if (e == hipSuccess) {
        m->owned.push_back(p);
        m->plane_jobs.push_back({wf, n, (__half*)p, ln, pre, 0});
        e = launch_split_planes(wf, (__half*)p, n, n, nullptr, W_PLANE_SCALE * pre, ln, (int)dm, probe);
      }
      if (e != hipSuccess) { err = hip_fail(e, "weight split", __FILE__, __LINE__); return; }
      *outp = (__half*)p;
    };
    auto mkv = [&](const std::vector<const float*>& src, size_t n, std::vector<__half*>& dst,
                   const std::vector<const float*>* ln = nullptr) {
      dst.assign(src.size(), nullptr);
      for (size_t i = 0; i < src.size(); ++i) mk(src[i], n, &dst[i], ln ? (*ln)[i] : nullptr);
    };
    mkv(m->enc_qkv, 3 * inner * dm, m->h_enc_qkv, &m->enc_ln0); mkv(m->enc_o, dm * inner, m->h_enc_o);
    mkv(m->enc_wi, dff * dm, m->h_enc_wi, &m->enc_ln1); mkv(m->enc_wo, dm * dff, m->h_enc_wo);
    mkv(m->dec_qkv, 3 * inner * dm, m->h_dec_qkv, &m->dec_ln0); mkv(m->dec_o, dm * inner, m->h_dec_o);
    mkv(m->dec_xq, inner * dm, m->h_dec_xq, &m->dec_ln1); mkv(m->dec_xo, dm * inner, m->h_dec_xo);
    mkv(m->dec_wi, dff * dm, m->h_dec_wi, &m->dec_ln2); mkv(m->dec_wo, dm * dff, m->h_dec_wo);
    mk(d->dec_xkv, (size_t)nd * 2 * inner * dm, &m->h_dec_xkv);
    if (!err) {   // the output codebooks, each padded to Vp rows (zero rows: their logits are 0 and never selectable)
      const size_t Vp = (size_t)m->Vp(), ps = (size_t)d->L * Vp * dm, n = (size_t)d->V * dm;
      const float pre = d->scaleup_output_hidden ? (float)pow((double)dm, -0.5) : 1.0f;
      void* p = nullptr;
      hipError_t e = hipMalloc(&p, ps * 2 * sizeof(__half));
      if (e == hipSuccess) { m->owned.push_back(p); e = hipMemset(p, 0, ps * 2 * sizeof(__half)); }
      for (int l = 0; l < d->L && e == hipSuccess; ++l) {
        __half* dst = (__half*)p + (size_t)l * Vp * dm;
        m->plane_jobs.push_back({d->out_embeds + (size_t)l * n, n, dst, d->dec_final_ln, pre, ps});
        e = launch_split_planes(d->out_embeds + (size_t)l * n, dst, n, ps, nullptr, W_PLANE_SCALE * pre, d->dec_final_ln, (int)dm, probe);
      }
      if (e != hipSuccess) err = hip_fail(e, "codebook split", __FILE__, __LINE__);
      m->h_out_embeds = (__half*)p;
    }
    if (!err) { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) err = hip_fail(e, "sync", __FILE__, __LINE__); }
    if (err) return err;
    // a weight (times its folded layer-norm weight, times 2^8) outside the f16 range cannot be carried by the planes:
    // the model is pinned to the exact-fp32 kernels instead of being clipped silently
    unsigned int sat = 0;
    RPR_HIP(hipMemcpy(&sat, probe, 4, hipMemcpyDeviceToHost));
    if (sat) m->f32_only = true;
  }
  {  // bound of |logit| (forced-tail fork, see internal.h)
    const int e = compute_logit_bound(c, m.get(), nullptr, &m->logit_bound);
    if (e) return e;
  }
  *out = m.release();
  return RPR_OK;
}

void rpr_free_model(rpr_model* m) {
  if (!m) return;
  (void)hipSetDevice(m->ctx->device);
  (void)hipDeviceSynchronize();
  // graphs that reference this model's tables are dropped
  for (auto it = m->ctx->graphs.begin(); it != m->ctx->graphs.end();) {
    if (it->first.m == m) { (void)hipGraphExecDestroy(it->second); it = m->ctx->graphs.erase(it); } else ++it;
  }
  train_forget_model(m->ctx, m);
  delete m;
}

// child arrays of the trie for its current vocab size (see rpr_trie / trie.h ChildLevels): built on the host in one pass over
// the sorted rows, uploaded once; rpr_trie_set_vocab rebuilds them (the dense tables are indexed with V)
static int upload_levels(rpr_trie* t) {
  t->free_levels();
  static const bool levels_on = [] { const char* e = getenv("RPR_SELECT_LEVELS"); return !(e && atoi(e) == 0); }();
  if (!levels_on || t->N >= ((int64_t)1 << 31) - 1) return RPR_OK;
  ChildLevels cl;
  // at most two entries per doc over all the deep levels: 8.8 M x 32 MS MARCO codes need 6.9 M (level 2 only)
  build_child_levels(t->host_sorted.data(), t->N, t->L, t->V, TRIE_NARROW, TRIE_MAX_DEEP, 2 * t->N + 1024, cl);
  auto up = [](auto*& dst, const auto& v) -> hipError_t {
    if (v.empty()) return hipSuccess;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&dst), v.size() * sizeof(v[0]));
    if (e != hipSuccess) return e;
    return hipMemcpy(dst, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice);
  };
  RPR_HIP(up(t->lvl0, cl.lvl0));
  RPR_HIP(up(t->lvl1, cl.lvl1));
  if (!cl.deep.empty()) RPR_HIP(up(t->idx2, cl.idx2));
  for (size_t i = 0; i < cl.deep.size() && i < (size_t)TRIE_MAX_DEEP; ++i) {
    RPR_HIP(up(t->d_start[i], cl.deep[i].start));
    RPR_HIP(up(t->d_tok[i], cl.deep[i].tok));
    t->d_n[i] = (int)cl.deep[i].start.size();
    t->n_deep = (int)i + 1;
  }
  t->lvl_V = t->V;
  return RPR_OK;
}

static int upload_trie(rpr_ctx* c, std::unique_ptr<rpr_trie>& t) {
  RPR_HIP(hipSetDevice(c->device));
  RPR_HIP(hipMalloc(&t->codes, t->host_sorted.size() * sizeof(uint16_t)));
  RPR_HIP(hipMemcpy(t->codes, t->host_sorted.data(), t->host_sorted.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  return upload_levels(t.get());
}

int rpr_build_trie(rpr_ctx* c, const uint16_t* codes, int64_t N, int32_t L, int32_t V, rpr_trie** out) {
  RPR_REQUIRE(c && codes && out, "NULL argument");
  RPR_REQUIRE(N > 0 && N < ((int64_t)1 << 31) - 1, "N out of range");
  RPR_REQUIRE(L >= 1 && L <= 4096 && V >= 1 && V <= 65536, "L or V out of range");
  for (int64_t i = 0; i < N * L; ++i) RPR_REQUIRE(codes[i] < V, "code >= V");
  auto t = std::make_unique<rpr_trie>();
  t->ctx = c; t->N = N; t->L = L; t->V = V;
  sort_codes(codes, N, L, t->host_sorted, t->perm);
  int e = upload_trie(c, t);
  if (e) return e;
  *out = t.release();
  return RPR_OK;
}

void rpr_free_trie(rpr_trie* t) {
  if (!t) return;
  (void)hipSetDevice(t->ctx->device);
  (void)hipDeviceSynchronize();
  for (auto it = t->ctx->graphs.begin(); it != t->ctx->graphs.end();) {
    if (it->first.t == t) { (void)hipGraphExecDestroy(it->second); it = t->ctx->graphs.erase(it); } else ++it;
  }
  delete t;
}

int64_t rpr_trie_num_rows(const rpr_trie* t) { return t ? t->N : 0; }
const int64_t* rpr_trie_perm(const rpr_trie* t) { return t ? t->perm.data() : nullptr; }

int rpr_trie_save(const rpr_trie* t, const char* path) {
  RPR_REQUIRE(t && path, "NULL argument");
  if (save_trie_file(path, t->host_sorted, t->perm, t->N, t->L, t->V, t->keys, 0, 0) != 0) {
    set_error(std::string("cannot write trie file ") + path);
    return RPR_ERR_INVALID;
  }
  return RPR_OK;
}

int rpr_trie_build_file(const uint16_t* codes, int64_t N, int32_t L, int32_t V, const char* keys, int64_t key_bytes,
                        int64_t src_size, int64_t src_mtime_ns, const char* path) {
  RPR_REQUIRE(codes && path, "NULL argument");
  RPR_REQUIRE(N > 0 && N < ((int64_t)1 << 31) - 1, "N out of range");
  RPR_REQUIRE(L >= 1 && L <= 4096 && V >= 1 && V <= 65536, "L or V out of range");
  RPR_REQUIRE(key_bytes >= 0 && (key_bytes == 0 || keys), "keys missing");
  for (int64_t i = 0; i < N * L; ++i) RPR_REQUIRE(codes[i] < V, "code >= V");
  try {
    std::vector<uint16_t> sorted;
    std::vector<int64_t> perm;
    sort_codes(codes, N, L, sorted, perm);
    const std::string k = key_bytes ? std::string(keys, (size_t)key_bytes) : std::string();
    if (save_trie_file(path, sorted, perm, N, L, V, k, src_size, src_mtime_ns) != 0) {
      set_error(std::string("cannot write trie file ") + path);
      return RPR_ERR_INVALID;
    }
  } catch (const std::exception& ex) {
    set_error(std::string("rpr_trie_build_file: ") + ex.what());
    return RPR_ERR_OOM;
  }
  return RPR_OK;
}

int rpr_trie_single_frac(const uint16_t* codes, int64_t N, int32_t Lc, int32_t L, double* out_frac) {
  RPR_REQUIRE(codes && out_frac, "NULL argument");
  RPR_REQUIRE(N > 0 && N < ((int64_t)1 << 31) - 1 && Lc >= 1 && Lc <= 4096 && L >= 1 && L <= Lc, "N, Lc or L out of range");
  try {
    std::vector<uint16_t> sorted;
    std::vector<int64_t> perm;
    sort_codes(codes, N, Lc, sorted, perm);
    std::vector<double> f;
    trie_single_frac(sorted.data(), N, Lc, L, f);
    for (int t = 0; t <= L; ++t) out_frac[t] = f[(size_t)t];
  } catch (const std::exception& ex) {
    set_error(std::string("rpr_trie_single_frac: ") + ex.what());
    return RPR_ERR_OOM;
  }
  return RPR_OK;
}

int rpr_trie_file_info(const char* path, int64_t* N, int32_t* L, int32_t* V, int64_t* key_bytes, int64_t* src_size,
                       int64_t* src_mtime_ns) {
  RPR_REQUIRE(path, "NULL argument");
  int64_t h[6];
  if (trie_file_info(path, h) != 0) { set_error(std::string("not a readable RPRTRIE2 file: ") + path); return RPR_ERR_INVALID; }
  if (N) *N = h[0];
  if (L) *L = (int32_t)h[1];
  if (V) *V = (int32_t)h[2];
  if (key_bytes) *key_bytes = h[3];
  if (src_size) *src_size = h[4];
  if (src_mtime_ns) *src_mtime_ns = h[5];
  return RPR_OK;
}

int rpr_trie_file_validate(const char* path) {
  RPR_REQUIRE(path, "NULL argument");
  try {
    std::vector<uint16_t> sorted; std::vector<int64_t> perm; std::string keys, err;
    int64_t N; int L, V;
    if (load_trie_file(path, sorted, perm, N, L, V, keys, err) != 0) {
      set_error(std::string("invalid trie file ") + path + ": " + err);
      return RPR_ERR_INVALID;
    }
  } catch (const std::exception& ex) {
    set_error(std::string("rpr_trie_file_validate: ") + ex.what());
    return RPR_ERR_OOM;
  }
  return RPR_OK;
}

int rpr_trie_load(rpr_ctx* c, const char* path, rpr_trie** out) {
  RPR_REQUIRE(c && path && out, "NULL argument");
  try {
    auto t = std::make_unique<rpr_trie>();
    t->ctx = c;
    std::string err;
    if (load_trie_file(path, t->host_sorted, t->perm, t->N, t->L, t->V, t->keys, err) != 0) {
      set_error(std::string("cannot load trie file ") + path + ": " + err);
      return RPR_ERR_INVALID;
    }
    int e = upload_trie(c, t);
    if (e) return e;
    *out = t.release();
  } catch (const std::exception& ex) {   // nothing may cross the C ABI
    set_error(std::string("rpr_trie_load: ") + ex.what());
    return RPR_ERR_OOM;
  }
  return RPR_OK;
}

int rpr_trie_dims(const rpr_trie* t, int64_t* N, int32_t* L, int32_t* V, int64_t* key_bytes) {
  RPR_REQUIRE(t, "NULL trie");
  if (N) *N = t->N;
  if (L) *L = t->L;
  if (V) *V = t->V;
  if (key_bytes) *key_bytes = (int64_t)t->keys.size();
  return RPR_OK;
}

int rpr_trie_keys(const rpr_trie* t, char* out) {
  RPR_REQUIRE(t && out, "NULL argument");
  std::memcpy(out, t->keys.data(), t->keys.size());
  return RPR_OK;
}

int rpr_trie_set_vocab(rpr_trie* t, int32_t V) {
  RPR_REQUIRE(t, "NULL trie");
  RPR_REQUIRE(V >= 1 && V <= 65536, "V out of range");
  uint16_t mx = 0;
  for (uint16_t v : t->host_sorted) mx = v > mx ? v : mx;
  RPR_REQUIRE((int)mx < V, "a code of the trie is >= the requested vocab size");
  if (V == t->V) return RPR_OK;
  t->V = V;
  // the child arrays are indexed with the vocab size they were built for: rebuild them (searches captured against the old
  // tables are dropped with them)
  RPR_HIP(hipSetDevice(t->ctx->device));
  RPR_HIP(hipDeviceSynchronize());
  for (auto it = t->ctx->graphs.begin(); it != t->ctx->graphs.end();) {
    if (it->first.t == t) { (void)hipGraphExecDestroy(it->second); it = t->ctx->graphs.erase(it); } else ++it;
  }
  return upload_levels(t);
}

int rpr_d2s_open(const char* path, rpr_d2s** out) {
  RPR_REQUIRE(path && out, "NULL argument");
  auto h = std::make_unique<rpr_d2s>();
  std::string err;
  if (read_docid_to_smtid(path, h->codes, h->keys, h->N, h->L, err) != 0) {
    set_error("docid_to_smtid: " + err);
    return RPR_ERR_INVALID;
  }
  *out = h.release();
  return RPR_OK;
}

int rpr_d2s_dims(const rpr_d2s* h, int64_t* N, int32_t* L, int64_t* key_bytes) {
  RPR_REQUIRE(h, "NULL handle");
  if (N) *N = h->N;
  if (L) *L = h->L;
  if (key_bytes) *key_bytes = (int64_t)h->keys.size();
  return RPR_OK;
}

int rpr_d2s_copy(const rpr_d2s* h, uint16_t* codes, char* keys) {
  RPR_REQUIRE(h, "NULL handle");
  if (codes) std::memcpy(codes, h->codes.data(), h->codes.size() * sizeof(uint16_t));
  if (keys) std::memcpy(keys, h->keys.data(), h->keys.size());
  return RPR_OK;
}

void rpr_d2s_close(rpr_d2s* h) { delete h; }

int rpr_trie_mask(rpr_ctx* c, const rpr_trie* t, const int32_t* prefix, int32_t R, int32_t T, uint8_t* out_mask) {
  RPR_REQUIRE(c && t && prefix && out_mask, "NULL argument");
  RPR_REQUIRE(R >= 1 && T >= 1, "R and T must be >= 1");
  RPR_HIP(hipSetDevice(c->device));
  DevTmp dp, dm;
  RPR_HIP(dp.alloc((size_t)R * T * 4));
  RPR_HIP(dm.alloc((size_t)R * t->V));
  RPR_HIP(hipMemcpy(dp.p, prefix, (size_t)R * T * 4, hipMemcpyHostToDevice));
  RPR_HIP(launch_prefix_mask(t->codes, t->L, t->N, dp.as<int32_t>(), R, T, t->V, dm.as<uint8_t>(), nullptr));
  RPR_HIP(hipMemcpy(out_mask, dm.p, (size_t)R * t->V, hipMemcpyDeviceToHost));
  return RPR_OK;
}

}  // extern "C"

namespace {

// the two CU-masked lane streams of a ctx (created on first use)
bool ensure_lanes(rpr_ctx* c) {
  if (c->lanes_state) return c->lanes_state > 0;
  c->lanes_state = -1;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) { (void)hipGetLastError(); return false; }
  const int cus = prop.multiProcessorCount, words = (cus + 31) / 32;
  if (cus < 64 || words > 32) return false;
  c->lane_cus = cus / 2;
  for (int i = 0; i < 2; ++i) {
    uint32_t mask[32] = {0};
    for (int k = (i == 0 ? 0 : cus / 2); k < (i == 0 ? cus / 2 : cus); ++k) mask[k >> 5] |= 1u << (k & 31);
    if (hipExtStreamCreateWithCUMask(&c->lanes[i].stream, (uint32_t)words, mask) != hipSuccess ||
        hipEventCreateWithFlags(&c->lanes[i].done, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
  }
  if (hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
  c->lanes_state = 1;
  return true;
}

// Fork depths of a search (ascending, at most MAX_FORKS; empty = every query walks all L steps). Explicit depths
// (rpr_set_fork_depths / RPR_FORK_DEPTHS) win; otherwise they come from the trie: with f_t = the share of depth-t nodes
// under which one distinct sequence remains (trie_single_frac), a query whose B beams sit on random depth-t nodes is
// forced with probability ~ f_t^B. First fork: the first depth where that reaches one half; second fork: the first
// depth after it where fewer than 0.05 queries of the call are expected to stay unforced, so that the last stage is
// almost always empty (a stage with a handful of live rows still pays ~100 launches per step).
std::vector<int> choose_forks(rpr_ctx* c, const rpr_model* m, rpr_trie* tr, int Q, int B, int L, unsigned flags, bool taps,
                              bool* drop_last) {
  std::vector<int> forks;
  *drop_last = false;
  if (!c->forced_tail || taps || L < 3 || !std::isfinite(m->logit_bound)) return forks;
  const double per_step = 2.0 * (double)m->logit_bound + ((flags & RPR_FLAG_LOG_SOFTMAX) ? log((double)m->d.V) : 0.0);
  if (1e8 - L * per_step <= 1e7) return forks;   // logits too large for the masked-candidate proof
  if (c->n_fork_override >= 0) {
    int prev = 0;
    for (int i = 0; i < c->n_fork_override; ++i) {
      const int t = c->fork_override[i];
      if (t > prev && t <= L - 1) { forks.push_back(t); prev = t; }
    }
    *drop_last = c->forced_tail == 2 && !forks.empty();
    return forks;
  }
  auto it = tr->single_frac.find(L);
  if (it == tr->single_frac.end()) {
    std::vector<double> f;
    trie_single_frac(tr->host_sorted.data(), tr->N, tr->L, L, f);
    it = tr->single_frac.emplace(L, std::move(f)).first;
  }
  const std::vector<double>& f = it->second;
  auto p_forced = [&](int t) { return std::pow(f[(size_t)t], (double)B); };
  int t0 = 0;
  for (int t = 1; t <= L - 2 && !t0; ++t) if (p_forced(t) >= 0.5) t0 = t;
  // a tail pass costs what its positions cost step by step minus the K/V gathering, plus a fork (~100 launches, two
  // partly filled launches for the leftovers): with thousands of decoder rows in flight — steps bound by the matrix
  // pipes — the plain loop is as fast below 8 remaining positions (measured at beam 100, len 8, 214 queries: 1890
  // queries/s without forks, 1510 with). A few hundred rows (the reference's rank-data flags: beam 100, batch 4, len 8)
  // are bound by the launch chain instead: a step of 12 layers costs 1.6 ms whatever it computes, the four remaining
  // positions as ONE pass of 1600 rows cost as much as one and a half steps (round 6: 306 -> 378 queries/s)
  const int min_tail = (int64_t)Q * B <= 4096 ? 2 : 8;
  if (!t0 || L - t0 < min_tail) return forks;
  forks.push_back(t0);
  // Optimistic mode (rpr_set_forced_tail(ctx, 2)): when the statistics promise an (almost always) empty last stage, that
  // stage is not enqueued at all — ~100 launches per step for nobody — and a query that is still unforced at the last
  // fork raises RPR_STATUS_TAIL_LEFTOVER instead; the caller then repeats the batch in the exact mode (1).
  // A handful of queries in flight: the first fork already leaves fewer than 0.05 queries behind in expectation, so the
  // second fork (a compacted stage, its steps and a second tail pass: ~300 launches that almost always work on nothing,
  // 2 of the 9.8 ms of a single-query search) is not enqueued either.
  if (c->forced_tail == 2 && (double)Q * (1.0 - p_forced(t0)) <= 0.05) { *drop_last = true; return forks; }
  for (int t = t0 + 1; t <= L - 2 && t <= t0 + 12; ++t)
    if ((double)Q * (1.0 - p_forced(t)) <= 0.05) { forks.push_back(t); break; }
  *drop_last = c->forced_tail == 2 && forks.size() == 2;
  return forks;
}

int pack_forks(const std::vector<int>& forks, bool drop_last) {
  int v = drop_last ? (1 << 30) : 0;
  for (size_t i = 0; i < forks.size(); ++i) v |= forks[i] << (8 * i);
  return v;
}

// one search on stream s; lane >= 0: in that lane's workspace (swapped into c->ws for the duration of the call)
int search_one(rpr_ctx* c, rpr_model* m, rpr_trie* tr, const int32_t* input_ids, const int32_t* attention_mask, int32_t Q,
               int32_t Lq, int32_t B, int32_t L, uint32_t flags, int32_t* out_tokens, float* out_scores, int64_t* out_row_lo,
               int64_t* out_row_hi, const rpr_debug_taps* taps, hipStream_t s, int lane) {
  struct WsGuard {
    rpr_ctx* c; int lane;
    WsGuard(rpr_ctx* c_, int l) : c(c_), lane(l) { if (lane >= 0) { std::swap(c->ws, c->lanes[lane].ws); c->cur_cus = c->lane_cus; c->cur_lane = lane; } }
    ~WsGuard() { if (lane >= 0) { std::swap(c->ws, c->lanes[lane].ws); c->cur_cus = 0; c->cur_lane = -1; } }
  } ws_guard(c, lane);
  // a model whose weights do not fit the f16 planes runs on the exact-fp32 kernels whatever the ctx setting
  struct PrecGuard { rpr_ctx* c; int saved; ~PrecGuard() { c->precision = saved; } } guard{c, c->precision};
  if (c->precision == RPR_PREC_BF16) c->precision = RPR_PREC_F16X2;   // bf16 is a training-GEMM mode; scores need fp32-equivalent
  if (m->f32_only) c->precision = RPR_PREC_F32;
  bool drop_last = false;
  const std::vector<int> forks = choose_forks(c, m, tr, Q, B, L, flags, taps != nullptr, &drop_last);
  int e = alloc_workspace(c, m, Q, Lq, B, L, forks, drop_last, (flags & RPR_FLAG_LOG_SOFTMAX) != 0);
  if (e) return e;
  c->last_forks = forks;
  c->last_ws_mask |= lane >= 0 ? (2 << lane) : 1;
  Workspace& w = c->ws;
  const size_t T = (size_t)Q * Lq, R = (size_t)Q * B;
  RPR_HIP(hipMemcpyAsync(w.ids.p, input_ids, T * 4, hipMemcpyDeviceToDevice, s));
  RPR_HIP(hipMemcpyAsync(w.mask.p, attention_mask, T * 4, hipMemcpyDeviceToDevice, s));

  const bool eager = (flags & RPR_FLAG_NO_GRAPH) || taps || c->profiling;
  if (eager) {
    Launcher Ln{c, s};
    enqueue_search(Ln, c, m, tr, Q, Lq, B, L, flags, taps, forks, drop_last);
    if (Ln.err) return Ln.err;
  } else {
    // the per-call debug switches of the selection / ranking kernels are part of the key (tests flip them between calls)
    auto env_int = [](const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; };
    const unsigned dbg = (env_int("RPR_TAIL_RANK_REPLAY", 0) ? 1u : 0u) | ((unsigned)(env_int("RPR_SELECT_RADIX", -1) + 1) << 1);
    GraphKey key{m, tr, Q, Lq, B, L, flags | ((unsigned)c->precision << 16) | (dbg << 20), lane, pack_forks(forks, drop_last)};
    auto it = c->graphs.find(key);
    if (it == c->graphs.end()) {
      hipGraph_t graph = nullptr;
      RPR_HIP(hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal));
      Launcher Ln{c, c->cap_stream};
      enqueue_search(Ln, c, m, tr, Q, Lq, B, L, flags, nullptr, forks, drop_last);
      hipError_t ce = hipStreamEndCapture(c->cap_stream, &graph);
      if (Ln.err) { if (graph) (void)hipGraphDestroy(graph); return Ln.err; }
      if (ce != hipSuccess) return hip_fail(ce, "hipStreamEndCapture", __FILE__, __LINE__);
      hipGraphExec_t exec = nullptr;
      hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (ie != hipSuccess) return hip_fail(ie, "hipGraphInstantiate", __FILE__, __LINE__);
      it = c->graphs.emplace(key, exec).first;
    }
    RPR_HIP(hipGraphLaunch(it->second, s));
  }
  RPR_HIP(hipMemcpyAsync(out_tokens, w.o_tokens.p, R * (size_t)L * 4, hipMemcpyDeviceToDevice, s));
  RPR_HIP(hipMemcpyAsync(out_scores, w.o_scores.p, R * 4, hipMemcpyDeviceToDevice, s));
  if (out_row_lo) RPR_HIP(hipMemcpyAsync(out_row_lo, w.o_lo.p, R * 8, hipMemcpyDeviceToDevice, s));
  if (out_row_hi) RPR_HIP(hipMemcpyAsync(out_row_hi, w.o_hi.p, R * 8, hipMemcpyDeviceToDevice, s));
  return RPR_OK;
}

}  // namespace

extern "C" {

int rpr_search(rpr_ctx* c, rpr_model* m, rpr_trie* tr, const int32_t* input_ids, const int32_t* attention_mask,
               int32_t Q, int32_t Lq, int32_t B, int32_t L, uint32_t flags, int32_t* out_tokens, float* out_scores,
               int64_t* out_row_lo, int64_t* out_row_hi, const rpr_debug_taps* taps, void* stream) {
  RPR_REQUIRE(c && m && tr && input_ids && attention_mask && out_tokens && out_scores, "NULL argument");
  RPR_REQUIRE(m->ctx == c && tr->ctx == c, "model/trie belong to another ctx");
  RPR_REQUIRE(Q >= 1 && B >= 1 && B <= 65535, "Q or B out of range");
  RPR_REQUIRE(Lq >= 1 && Lq <= MAX_LQ, "Lq out of range (1..256)");
  RPR_REQUIRE(L >= 1 && L <= m->d.L && L <= tr->L, "L exceeds the model's decoder length or the trie depth");
  RPR_REQUIRE(tr->V == m->d.V, "trie V differs from the model's decoder vocab size");
  RPR_REQUIRE((int64_t)Q * B < ((int64_t)1 << 24), "Q*B too large");
  // the step self-attention splits its wave index (query, beam, head) by reciprocal multiplication, exact below
  // 2^32 / max(B, H) (launch_dec_self_attn): said here, not as a launch error in the middle of a capture
  RPR_REQUIRE((int64_t)Q * B * m->d.num_heads < ((int64_t)1 << 32) / std::max<int64_t>(B, m->d.num_heads),
              "Q * num_beams * num_heads too large for this beam count: search fewer queries per call");
  RPR_REQUIRE(select_fits(B, m->Vp()), "num_beams * decoder vocab size too large for the select kernel (about 1600 beams at V=256)");
  RPR_REQUIRE(!taps || m->Vp() == m->d.V, "debug taps need a decoder vocab size that is a multiple of 64");
  RPR_HIP(hipSetDevice(c->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  { const int pe = ensure_weight_planes(c, m, s); if (pe) return pe; }   // after an optimizer step
  c->last_ws_mask = 0;
  // Large batches: two halves on the two CU-masked lanes, side by side (see Lane). Results are those of one call: every
  // query is processed on its own rows. The caller's stream waits for both lanes.
  if (c->lane_min_rows > 0 && (int64_t)Q * B >= c->lane_min_rows && Q >= 2 && !taps && ensure_lanes(c)) {
    const int32_t Qh[2] = {(Q + 1) / 2, Q / 2};
    // both lane workspaces are sized BEFORE either half is enqueued: growing a buffer drops every cached graph of the
    // ctx (ensure()), which must not happen while the other lane's graph is in flight — and an allocation failure then
    // leaves nothing running
    for (int i = 0; i < 2; ++i) {
      std::swap(c->ws, c->lanes[i].ws);
      const int saved_prec = c->precision;
      if (c->precision == RPR_PREC_BF16) c->precision = RPR_PREC_F16X2;
      if (m->f32_only) c->precision = RPR_PREC_F32;
      bool drop_last = false;
      const std::vector<int> forks = choose_forks(c, m, tr, Qh[i], B, L, flags, false, &drop_last);
      const int e = alloc_workspace(c, m, Qh[i], Lq, B, L, forks, drop_last, (flags & RPR_FLAG_LOG_SOFTMAX) != 0);
      c->precision = saved_prec;
      std::swap(c->ws, c->lanes[i].ws);
      if (e) return e;
    }
    RPR_HIP(hipEventRecord(c->fork_ev, s));
    int32_t q0 = 0;
    for (int i = 0; i < 2; ++i) {
      Lane& ln = c->lanes[i];
      RPR_HIP(hipStreamWaitEvent(ln.stream, c->fork_ev, 0));
      const size_t r0 = (size_t)q0 * B;
      int e = search_one(c, m, tr, input_ids + (size_t)q0 * Lq, attention_mask + (size_t)q0 * Lq, Qh[i], Lq, B, L, flags,
                         out_tokens + r0 * L, out_scores + r0, out_row_lo ? out_row_lo + r0 : nullptr,
                         out_row_hi ? out_row_hi + r0 : nullptr, nullptr, ln.stream, i);
      if (e) {   // the other half may already be writing the caller's buffers: let it finish before reporting the error
        for (int k = 0; k < 2; ++k) (void)hipStreamSynchronize(c->lanes[k].stream);
        return e;
      }
      RPR_HIP(hipEventRecord(ln.done, ln.stream));
      q0 += Qh[i];
    }
    for (int i = 0; i < 2; ++i) RPR_HIP(hipStreamWaitEvent(s, c->lanes[i].done, 0));
    return RPR_OK;
  }
  return search_one(c, m, tr, input_ids, attention_mask, Q, Lq, B, L, flags, out_tokens, out_scores, out_row_lo, out_row_hi, taps,
                    s, -1);
}

int rpr_set_lane_split(rpr_ctx* c, int32_t min_rows) {
  RPR_REQUIRE(c && min_rows >= 0, "NULL ctx or negative threshold");
  c->lane_min_rows = min_rows;
  return RPR_OK;
}
int32_t rpr_lane_split(rpr_ctx* c) {
  if (!c || c->lane_min_rows <= 0) return 0;
  (void)hipSetDevice(c->device);
  return ensure_lanes(c) ? c->lane_min_rows : 0;
}

int rpr_set_forced_tail(rpr_ctx* c, int32_t mode) {
  RPR_REQUIRE(c && mode >= 0 && mode <= 2, "mode must be 0 (off), 1 (exact) or 2 (optimistic)");
  c->forced_tail = mode;
  return RPR_OK;
}
int32_t rpr_forced_tail(const rpr_ctx* c) { return c ? c->forced_tail : -1; }

int rpr_set_fork_depths(rpr_ctx* c, int32_t n, const int32_t* depths) {
  RPR_REQUIRE(c && n >= -1 && n <= MAX_FORKS && (n <= 0 || depths), "n out of range (-1 = automatic, 0..2 explicit depths)");
  c->n_fork_override = n;
  for (int i = 0; i < n; ++i) {
    RPR_REQUIRE(depths[i] >= 1 && depths[i] < 256 && (i == 0 || depths[i] > depths[i - 1]), "fork depths must be ascending and >= 1");
    c->fork_override[i] = depths[i];
  }
  return RPR_OK;
}

int rpr_fork_depths(rpr_ctx* c, rpr_model* m, rpr_trie* tr, int32_t Q, int32_t B, int32_t L, uint32_t flags, int32_t* out_depths) {
  RPR_REQUIRE(c && m && tr && out_depths, "NULL argument");
  RPR_REQUIRE(Q >= 1 && B >= 1 && L >= 1 && L <= tr->L, "Q, B or L out of range");
  bool drop_last = false;
  const std::vector<int> f = choose_forks(c, m, tr, Q, B, L, flags, false, &drop_last);
  for (size_t i = 0; i < f.size(); ++i) out_depths[i] = f[i];
  return (int)f.size();
}

int rpr_last_fork_stats(rpr_ctx* c, int32_t* out_depths, int32_t* out_forced, int32_t* out_left) {
  RPR_REQUIRE(c && out_depths && out_forced && out_left, "NULL argument");
  RPR_HIP(hipSetDevice(c->device));
  RPR_HIP(hipDeviceSynchronize());
  const int n = (int)c->last_forks.size();
  for (int k = 0; k < n; ++k) {
    out_depths[k] = c->last_forks[(size_t)k]; out_forced[k] = 0; out_left[k] = 0;
    const Workspace* wss[3] = {&c->ws, &c->lanes[0].ws, &c->lanes[1].ws};
    for (int i = 0; i < 3; ++i) {
      if (!(c->last_ws_mask & (1 << i)) || !wss[i]->tail[k].cnt.p || !wss[i]->stage[k].cnt.p) continue;
      int32_t a = 0, b = 0;
      RPR_HIP(hipMemcpy(&a, wss[i]->tail[k].cnt.p, 4, hipMemcpyDeviceToHost));
      RPR_HIP(hipMemcpy(&b, wss[i]->stage[k].cnt.p, 4, hipMemcpyDeviceToHost));
      out_forced[k] += a; out_left[k] += b;
    }
  }
  return n;
}

int rpr_lngknp_forward(rpr_ctx* c, rpr_model* m, const int32_t* input_ids, const int32_t* attention_mask, int32_t bz,
                       int32_t Lq, const int32_t* doc_codes, int32_t n_docs, int32_t L, const float* teacher_pos,
                       const float* teacher_neg, const int32_t* prefix_lens, int32_t n_prefix, float* out_losses,
                       float* out_position_scores, void* stream) {
  RPR_REQUIRE(c && m && input_ids && attention_mask && doc_codes, "NULL argument");
  RPR_REQUIRE(m->ctx == c, "model belongs to another ctx");
  RPR_REQUIRE(bz >= 1 && Lq >= 1 && Lq <= MAX_LQ, "bz or Lq out of range");
  RPR_REQUIRE(L >= 1 && L <= m->d.L && L <= MAX_LQ, "smtid length exceeds the model's decoder length");
  RPR_REQUIRE(n_docs >= 1 && (int64_t)bz * n_docs * L < ((int64_t)1 << 24), "n_docs out of range");
  RPR_REQUIRE(m->d.d_kv == DKV, "the teacher-forced kernels are written for d_kv == 64 (t5-base / t5-large)");
  RPR_REQUIRE(n_prefix >= 0 && n_prefix <= 8, "n_prefix out of range (0..8)");
  RPR_REQUIRE(n_prefix == 0 || (n_docs == 2 && teacher_pos && teacher_neg && prefix_lens && out_losses),
              "the margin losses need n_docs == 2 (positive, negative), teacher scores, prefix lengths and out_losses");
  RPR_REQUIRE(out_losses || out_position_scores, "nothing to return");
  RPR_HIP(hipSetDevice(c->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int e = ensure_weight_planes(c, m, s);
  if (e) return e;
  e = alloc_train_workspace(c, m, bz, Lq, n_docs, L);
  if (e) return e;
  Workspace& w = c->ws;
  const size_t T = (size_t)bz * Lq, R = (size_t)bz * n_docs * L;
  RPR_HIP(hipMemcpyAsync(w.ids.p, input_ids, T * 4, hipMemcpyDeviceToDevice, s));
  RPR_HIP(hipMemcpyAsync(w.mask.p, attention_mask, T * 4, hipMemcpyDeviceToDevice, s));
  float* scores = P<float>(w.tr_misc);
  int32_t* codes = reinterpret_cast<int32_t*>(scores + R);
  RPR_HIP(hipMemcpyAsync(codes, doc_codes, R * 4, hipMemcpyDeviceToDevice, s));
  struct PrecGuard { rpr_ctx* c; int saved; ~PrecGuard() { c->precision = saved; } } guard{c, c->precision};
  if (c->precision == RPR_PREC_BF16) c->precision = RPR_PREC_F16X2;
  if (m->f32_only) c->precision = RPR_PREC_F32;
  Launcher Ln{c, s};
  enqueue_train_forward(Ln, c, m, bz, Lq, n_docs, L, codes, scores);
  if (Ln.err) return Ln.err;
  if (n_prefix > 0)
    RPR_HIP(launch_margin_mse(scores, teacher_pos, teacher_neg, prefix_lens, n_prefix, bz, L, out_losses, nullptr, s));
  if (out_position_scores) RPR_HIP(hipMemcpyAsync(out_position_scores, scores, R * 4, hipMemcpyDeviceToDevice, s));
  return RPR_OK;
}

int rpr_encode(rpr_ctx* c, rpr_model* m, const int32_t* input_ids, const int32_t* attention_mask, int32_t Q,
               int32_t Lq, float* out, void* stream) {
  RPR_REQUIRE(c && m && input_ids && attention_mask && out, "NULL argument");
  RPR_REQUIRE(Q >= 1 && Lq >= 1 && Lq <= MAX_LQ, "Q or Lq out of range");
  RPR_HIP(hipSetDevice(c->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int e = ensure_weight_planes(c, m, s);
  if (e) return e;
  e = alloc_workspace(c, m, Q, Lq, 1, 1);
  if (e) return e;
  Workspace& w = c->ws;
  const size_t T = (size_t)Q * Lq;
  RPR_HIP(hipMemcpyAsync(w.ids.p, input_ids, T * 4, hipMemcpyDeviceToDevice, s));
  RPR_HIP(hipMemcpyAsync(w.mask.p, attention_mask, T * 4, hipMemcpyDeviceToDevice, s));
  struct PrecGuard { rpr_ctx* c; int saved; ~PrecGuard() { c->precision = saved; } } guard{c, c->precision};
  if (c->precision == RPR_PREC_BF16) c->precision = RPR_PREC_F16X2;
  if (m->f32_only) c->precision = RPR_PREC_F32;
  Launcher Ln{c, s};
  enqueue_encoder(Ln, c, m, Q, Lq, false);
  if (Ln.err) return Ln.err;
  RPR_HIP(hipMemcpyAsync(out, w.enc_out.p, T * m->d.d_model * 4, hipMemcpyDeviceToDevice, s));
  return RPR_OK;
}

int rpr_op_linear(rpr_ctx* c, const float* A, const float* W, const float* residual, float* C, int32_t M, int32_t N,
                  int32_t K, int32_t relu, void* stream) {
  RPR_REQUIRE(c && A && W && C, "NULL argument");
  RPR_REQUIRE(M >= 1 && N >= 1 && K >= 32 && K % 32 == 0, "bad GEMM shape (K must be a multiple of 32)");
  RPR_HIP(hipSetDevice(c->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Launcher Ln{c, s};
  DevTmp At, Wt;
  struct PrecGuard { rpr_ctx* c; int saved; ~PrecGuard() { c->precision = saved; } } guard{c, c->precision};
  if (c->precision == RPR_PREC_BF16) c->precision = RPR_PREC_F16X2;
  if (c->precision == RPR_PREC_F16X2) {  // test hook: split the operands on the fly
    { const int pe = ensure(c, c->ws.part, (size_t)9 << 20 << 2); if (pe) return pe; }   // split-K scratch, as a search has it
    RPR_HIP(At.alloc((size_t)M * K * 2 * sizeof(__half)));
    RPR_HIP(Wt.alloc((size_t)N * K * 2 * sizeof(__half)));
    RPR_HIP(launch_split_planes(A, At.as<__half>(), (size_t)M * K, (size_t)M * K, s, 1.0f, nullptr, 0, c->status));
    RPR_HIP(launch_split_planes(W, Wt.as<__half>(), (size_t)N * K, (size_t)N * K, s, W_PLANE_SCALE, nullptr, 0, c->status));
  }
  linear(Ln, {A, At.as<__half>(), (size_t)M * K, K, 1.0f}, {W, Wt.as<__half>(), N, K}, M, out_f32(C, N, N, residual, relu));
  if (At.p) RPR_HIP(hipStreamSynchronize(s));   // the temporaries are freed when this scope ends
  if (c->trace_buf) {  // dump the stamps of this launch: K/32 tiles x 8 waves x 18 slots (gemm_h2_pp_kernel<.., TRACE>)
    const char* we = dev_getenv("RPR_GEMM_TRACE_W");
    const size_t tw = we ? (size_t)atoi(we) : 18;
    const size_t n = (size_t)(K / 32) * 8 * tw;
    std::vector<unsigned long long> hbuf(n);
    RPR_HIP(hipMemcpy(hbuf.data(), c->trace_buf, n * 8, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(dev_getenv("RPR_GEMM_TRACE"), "w")) {
      for (size_t i = 0; i < n; ++i) fprintf(f, "%llu%c", hbuf[i], (i % tw == tw - 1) ? '\n' : ' ');
      fclose(f);
    }
    // per-tile wall-clock stamps of block 0 (persistent kernel): start, first K-tile landed, K-loop done, epilogue issued
    std::vector<unsigned long long> tb(4 * 4096);
    RPR_HIP(hipMemcpy(tb.data(), c->trace_buf + 100000, tb.size() * 8, hipMemcpyDeviceToHost));
    if (FILE* f = fopen((std::string(dev_getenv("RPR_GEMM_TRACE")) + ".tiles").c_str(), "w")) {
      for (size_t i = 0; i + 3 < tb.size() && tb[i]; i += 4) fprintf(f, "%llu %llu %llu %llu\n", tb[i], tb[i + 1], tb[i + 2], tb[i + 3]);
      fclose(f);
    }
    RPR_HIP(hipMemset(c->trace_buf + 100000, 0, tb.size() * 8));
  }
  return Ln.err;
}

int rpr_op_linear_bf16(rpr_ctx* c, const float* A, const float* W, const float* residual, float* C, int32_t M, int32_t N, int32_t K,
                       int32_t relu, int32_t n_products, void* stream) {
  RPR_REQUIRE(c && A && W && C, "NULL argument");
  RPR_REQUIRE(M >= 1 && N >= 1 && K >= 64 && K % 64 == 0, "bad GEMM shape (K must be a multiple of 64)");
  RPR_REQUIRE(n_products >= 0 && n_products <= GemmGroupArgs::MAXP, "n_products out of range");
  RPR_REQUIRE(n_products == 0 || (!residual && !relu && (N & 3) == 0), "the grouped launch has no fused extras");
  RPR_HIP(hipSetDevice(c->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DevTmp At, Wt, part, tab;
  RPR_HIP(At.alloc((size_t)M * K * sizeof(__half)));
  RPR_HIP(Wt.alloc((size_t)N * K * sizeof(__half)));
  RPR_HIP(launch_to_bf16(A, M, K, K, At.p, s));
  RPR_HIP(launch_to_bf16(W, N, K, K, Wt.p, s));
  if (n_products == 0) {
    const size_t part_bytes = (size_t)64 << 20;
    RPR_HIP(part.alloc(part_bytes));
    GemmH2Args g{};
    g.A = At.as<__half>(); g.lda = K; g.W = Wt.as<__half>(); g.ldw = K;
    g.resid = residual; g.ldr = N;
    g.out[0] = g.out[1] = g.out[2] = C; g.ldo[0] = g.ldo[1] = g.ldo[2] = N; g.split_n = N;
    g.M = M; g.N = N; g.K = K; g.relu = relu; g.acc_scale = 1.0f; g.bf16 = 1;
    g.part = part.as<float>(); g.part_cap = part_bytes / sizeof(float);
    Launcher Ln{c, s};                                    // (profile accounting: tools/gemm_bf16_bench.py times the launch alone)
    Ln.run(RPR_K_GEMM, 2.0 * M * (double)N * K, 2.0 * ((double)M * K + (double)N * K) + 4.0 * (double)M * N, [&] { return launch_gemm_h2(g, s); },
           &g.kernel_cls);
    if (Ln.err) return Ln.err;
  } else {
    RPR_HIP(tab.alloc(GemmGroupArgs::SCRATCH_BYTES));
    GemmGroupArgs p{};
    p.K = K; p.lda = K; p.ldw = K;
    for (int i = 0; i < n_products && M - 256 * i > 0; ++i) {
      p.A[i] = At.as<__half>(); p.W[i] = Wt.as<__half>(); p.out[i] = C + (size_t)i * M * N;
      p.M[i] = M - 256 * i; p.N[i] = N; p.ldo[i] = N;
      p.n = i + 1;
    }
    RPR_HIP(launch_gemm_h2_group(p, tab.p, s));
  }
  RPR_HIP(hipStreamSynchronize(s));   // the temporaries are freed when this scope ends
  return RPR_OK;
}

int rpr_op_rmsnorm(rpr_ctx* c, const float* x, const float* w, float* out, int32_t rows, int32_t d, float eps,
                   void* stream) {
  RPR_REQUIRE(c && x && w && out, "NULL argument");
  RPR_REQUIRE(rows >= 1 && d >= 4 && d % 4 == 0, "bad shape");
  RPR_HIP(hipSetDevice(c->device));
  RPR_HIP(launch_rmsnorm(x, w, out, rows, d, eps, reinterpret_cast<hipStream_t>(stream)));
  return RPR_OK;
}

int rpr_get_status(rpr_ctx* c, void* stream, uint32_t* out_flags, int clear) {
  RPR_REQUIRE(c && out_flags, "NULL argument");
  RPR_HIP(hipSetDevice(c->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  RPR_HIP(hipMemcpyAsync(c->status_host, c->status, 12, hipMemcpyDeviceToHost, s));
  if (clear) RPR_HIP(hipMemsetAsync(c->status, 0, 12, s));
  RPR_HIP(hipStreamSynchronize(s));
  *out_flags = (c->status_host[0] ? RPR_STATUS_SATURATED : 0u) | (c->status_host[1] ? RPR_STATUS_EMPTY_QUERY : 0u) |
               (c->status_host[2] ? RPR_STATUS_TAIL_LEFTOVER : 0u);
  return RPR_OK;
}

int rpr_status_words_async(rpr_ctx* c, void* stream, uint32_t* host_words, int clear) {
  RPR_REQUIRE(c, "NULL ctx");
  RPR_HIP(hipSetDevice(c->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (host_words) RPR_HIP(hipMemcpyAsync(host_words, c->status, 16, hipMemcpyDeviceToHost, s));
  if (clear) RPR_HIP(hipMemsetAsync(c->status, 0, 16, s));
  return RPR_OK;
}

int rpr_model_f32_only(const rpr_model* m) { return m ? (m->f32_only ? 1 : 0) : -1; }   // as of the last plane split

int rpr_profile_enable(rpr_ctx* c, int enable) {
  RPR_REQUIRE(c, "NULL ctx");
  if (!enable && c->profiling) { int e = flush_profile(c); if (e) return e; }
  c->profiling = enable != 0;
  return RPR_OK;
}

int rpr_profile_reset(rpr_ctx* c) {
  RPR_REQUIRE(c, "NULL ctx");
  int e = flush_profile(c);
  if (e) return e;
  std::memset(c->done, 0, sizeof(c->done));
  return RPR_OK;
}

int rpr_profile_get(rpr_ctx* c, int cls, rpr_kernel_stats* out) {
  RPR_REQUIRE(c && out, "NULL argument");
  RPR_REQUIRE(cls >= 0 && cls < RPR_K_COUNT, "bad kernel class");
  int e = flush_profile(c);
  if (e) return e;
  *out = c->done[cls];
  return RPR_OK;
}

}  // extern "C"
