// Training step of the prefix-oriented ranking fine-tune (BASELINE config 5) behind the C ABI:
//   rpr_lngknp_backward  = forward (activations kept) + backward of T5SeqAQEncoderForLngKnpMarginMSE
//                          (reference modeling/t5_generative_retriever.py:902-966; loss.backward() in
//                          tasks/trainer.py:203-275) into one flat fp32 gradient buffer,
//   rpr_adamw_step       = clip_grad_norm_ + torch.optim.AdamW as HF Trainer configures them for main.py:131-155,
//   rpr_param_*          = the layout of the flat buffers (so the host can all-reduce them over RCCL and map them back
//                          to the checkpoint's tensor names).
// Arithmetic: fp32 activations and gradients. Matrix products: the search path's split-precision GEMM (f16 hi/lo planes,
// 3 f16 MFMAs per product, fp32 accumulation) with PER-TENSOR dynamic plane scales — gradients span ten orders of
// magnitude, so every operand's absolute maximum is found on the device and a power of two brings it to [2^13, 2^14)
// before the split; in RPR_PREC_F32 mode the exact-fp32 MFMA kernel (gemm_f32.hip). The reference trains under bf16
// autocast: both are the more precise side. The backward products reuse the forward kernels on transposed operands
// (train_kernels.hip); the reductions are deterministic (fixed-order partials, fixed-point integer atomics).
#include <cmath>
#include <cstring>
#include <unordered_map>

#include "internal.h"

using namespace rpr;

struct TrainWs {
  // saved forward activations
  DevBuf enc_act, dec_act, enc_out, xkv, x_last, scores, margins, dscores, in_idx, out_idx, tok_idx;
  // scratch
  DevBuf h, dxa, dxb, dbig, dattn, dxkv, denc, tA, wT, w_part, bias_part, fix, gn_part, gn_out, amax, part, part2;
  // The weight-gradient GEMMs run on side streams beside the input-gradient chain. dW[N, K] = dY^T X reduces over all
  // rows of the batch into 9 .. 36 tiles of 256 x 256: one launch cannot fill the chip, so the launches of consecutive
  // sites go round-robin to NSIDE streams and run next to each other, each walking the whole reduction in one K-loop
  // (no split-K partials, no reduce pass). One set of transposed-operand scratch per stream: a set is reused NSIDE
  // dxdw calls later, after its product has finished (ev_done).
  static constexpr int NSIDE = 4;
  DevBuf tB[NSIDE], tC[NSIDE];
  hipStream_t side[NSIDE] = {};
  hipEvent_t ev_fork[NSIDE] = {}, ev_done[NSIDE] = {};
  bool done_pending[NSIDE] = {};
  int flip = 0;
  std::vector<hipEvent_t> bucket_ev;   // gradient buckets handed to the caller's communication stream (2 events each)
  // bf16 mode: the GEMM weights are converted once per step (plain + transposed copies at the tensors' offsets of the flat
  // parameter layout) instead of once per GEMM call, and the forward pass leaves the transposed bf16 copy of every
  // linear layer's input behind — the X^T operand of its weight-gradient product — so the backward pass neither converts
  // X again nor recomputes the normalised inputs it would only need for that
  DevBuf wc, wcT, wseg, wpref, xT;
  // bf16 mode, grouped weight gradients: the dY^T operands of one layer's products live side by side in one of two sets
  // (a set is rewritten two layers later, after its group launch has finished: ev_gdone), the products are collected in
  // `grp` while the layer's input-gradient chain is enqueued and go out as ONE launch on the side stream (gtab = the
  // device table the launch reads its argument structs from)
  DevBuf dyT[2], gtab;
  // bf16 mode, round 6: the feed-forward block's wide intermediate leaves its producing GEMM's epilogue in bf16 (rows here, the
  // transposed copy in the consumer's X^T / dY^T slot) instead of through a conversion launch: relu(h Wi^T) in the forward
  // pass, the masked gradient w.r.t. it in the backward pass (GemmH2Args::out_b / out_bt / mask_src)
  DevBuf bfb;
  struct Pre { const float* dY = nullptr; const void* py = nullptr; __half* pyt = nullptr; int gset = -1; } pre;   // a dY the producer left converted
  hipEvent_t ev_gfork[2] = {}, ev_gdone[2] = {};
  bool gdone_pending[2] = {};
  GemmGroupArgs grp = {};
  int gset = 0, grp_tiles = 0;
  size_t dyT_used = 0;
  double grp_flops = 0, grp_bytes = 0;
  DevBuf aseg, apref;                  // rpr_adamw_step: one launch over all tensors (table built once per model)
  const rpr_model* aw_model = nullptr;
  int aw_nseg = 0, aw_chunks = 0;
  const rpr_model* wc_model = nullptr;
  int wc_nseg = 0, wc_tiles = 0;
  std::unordered_map<const float*, size_t> wc_off;
  int amax_next = 0;
  // forward GEMM site (keyed by its weight tensor) -> {amax of its input activations, amax of the weight}: the backward
  // multiplies the same two tensors again (dW = dY^T X, dX = dY W) and reuses both maxima
  std::unordered_map<const float*, float*> site_amax;
  size_t bytes = 0;
};

namespace {

enum { K_SHARED, K_ENC_REL, K_DEC_REL, K_ENC_FLN, K_DEC_FLN, K_START, K_IN_EMB, K_OUT_EMB, K_XKV, K_ENC_LN0, K_ENC_QKV, K_ENC_O,
       K_ENC_LN1, K_ENC_WI, K_ENC_WO, K_DEC_LN0, K_DEC_QKV, K_DEC_O, K_DEC_LN1, K_DEC_XQ, K_DEC_XO, K_DEC_LN2, K_DEC_WI, K_DEC_WO };

void build_params(rpr_model* m) {
  if (!m->params.empty()) return;
  const auto& d = m->d;
  const size_t dm = d.d_model, inner = m->inner(), dff = d.d_ff, nd = d.num_decoder_layers;
  size_t off = 0;
  auto add = [&](int kind, int layer, const float* p, size_t n) {
    m->params.push_back({kind, layer, const_cast<float*>(p), n, off});
    off += n;
  };
  add(K_SHARED, -1, d.shared, (size_t)d.vocab_size * dm);
  add(K_ENC_REL, -1, d.enc_rel_bias, (size_t)d.rel_buckets * d.num_heads);
  add(K_DEC_REL, -1, d.dec_rel_bias, (size_t)d.rel_buckets * d.num_heads);
  add(K_ENC_FLN, -1, d.enc_final_ln, dm);
  add(K_DEC_FLN, -1, d.dec_final_ln, dm);
  add(K_START, -1, d.start_embed, dm);
  add(K_IN_EMB, -1, d.in_embeds, (size_t)d.L * d.V * dm);
  if (d.out_embeds != d.in_embeds) add(K_OUT_EMB, -1, d.out_embeds, (size_t)d.L * d.V * dm);
  add(K_XKV, -1, d.dec_xkv, nd * 2 * inner * dm);
  for (int i = 0; i < d.num_layers; ++i) {
    add(K_ENC_LN0, i, m->enc_ln0[i], dm); add(K_ENC_QKV, i, m->enc_qkv[i], 3 * inner * dm); add(K_ENC_O, i, m->enc_o[i], dm * inner);
    add(K_ENC_LN1, i, m->enc_ln1[i], dm); add(K_ENC_WI, i, m->enc_wi[i], dff * dm); add(K_ENC_WO, i, m->enc_wo[i], dm * dff);
  }
  for (int i = 0; i < (int)nd; ++i) {
    add(K_DEC_LN0, i, m->dec_ln0[i], dm); add(K_DEC_QKV, i, m->dec_qkv[i], 3 * inner * dm); add(K_DEC_O, i, m->dec_o[i], dm * inner);
    add(K_DEC_LN1, i, m->dec_ln1[i], dm); add(K_DEC_XQ, i, m->dec_xq[i], inner * dm); add(K_DEC_XO, i, m->dec_xo[i], dm * inner);
    add(K_DEC_LN2, i, m->dec_ln2[i], dm); add(K_DEC_WI, i, m->dec_wi[i], dff * dm); add(K_DEC_WO, i, m->dec_wo[i], dm * dff);
  }
  m->params_total = off;
}

size_t param_offset(const rpr_model* m, int kind, int layer) {
  for (const auto& p : m->params)
    if (p.kind == kind && p.layer == layer) return p.offset;
  return (size_t)-1;
}

int tensure(rpr_ctx* c, DevBuf& b, size_t bytes) {   // like ensure(), without touching the search graphs
  if (bytes <= b.cap) return 0;
  if (b.p) { RPR_HIP(hipFree(b.p)); c->tws->bytes -= b.cap; b.p = nullptr; b.cap = 0; }
  const size_t want = (bytes + 255) & ~(size_t)255;
  RPR_HIP(hipMalloc(&b.p, want));
  b.cap = want;
  c->tws->bytes += want;
  return 0;
}

inline int pad32(int n) { return (n + 31) & ~31; }
inline int pad64(int n) { return (n + 63) & ~63; }
// row stride (elements) of the transposed bf16 operands [cols][pad64(rows)] of the weight-gradient products. (Padding it off
// the power of two — 8192 rows = 16 KB looked like a memory-channel hazard for K-tiles that are 128-byte pieces of 256 rows —
// measured no different: 27.4 ms per step unpadded, 27.6 with 64 elements, 28.3 with 32; HISTORY.md.)
inline int ldT(int rows) { return pad64(rows); }

struct Dims {
  int bz, Lq, L, S, R, T, dm, inner, dff, H, ne, nd, V, xld, buckets;
  float eps, post;
  size_t enc_stride, dec_stride;   // floats per saved layer
};

// Split-precision training GEMM: both operands are fp32 tensors of unknown magnitude (activations, gradients, weights
// that change every step), so each gets a per-tensor power-of-two scale from its absolute maximum, found on the device
// (no host round trip; slots of a ring that is zeroed once per pass) and undone in the GEMM epilogue.
struct Planes { __half* p; size_t ps; int ld; const float* amax; };
constexpr int AMAX_SLOTS = 8192;
float* amax_slots(rpr_ctx* c, int n) {          // n fresh (zero) slots
  TrainWs& w = *c->tws;
  if (w.amax_next + n > AMAX_SLOTS) return nullptr;
  float* p = P<float>(w.amax) + w.amax_next;
  w.amax_next += n;
  return p;
}
void amax_reset(Launcher& Ln) {
  TrainWs& w = *Ln.c->tws;
  w.amax_next = 0;
  w.site_amax.clear();
  Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_zero_u64(P<unsigned long long>(w.amax), AMAX_SLOTS / 2, Ln.s); });
}
void gemm_planes(Launcher& Ln, const Planes& A, const Planes& B, float* C, int ldc, int M, int N, int K, const float* resid, int relu,
                 DevBuf* part = nullptr, bool whole_k = false) {
  GemmH2Args g{};
  g.A = A.p; g.a_ps = A.ps; g.lda = A.ld; g.W = B.p; g.w_ps = B.ps; g.ldw = B.ld;
  g.resid = resid; g.ldr = ldc;
  g.out[0] = g.out[1] = g.out[2] = C; g.ldo[0] = g.ldo[1] = g.ldo[2] = ldc; g.split_n = N;
  g.M = M; g.N = N; g.K = K; g.relu = relu; g.acc_scale = 1.0f; g.sat = Ln.c->status;
  g.dyn_a = A.amax; g.dyn_b = B.amax;
  if (!part) part = &Ln.c->tws->part;
  if (whole_k) g.prefer_pp = 1;        // one K-loop per tile on the 256 x 256 kernel, no split-K (weight gradients)
  else { g.part = P<float>(*part); g.part_cap = part->cap / sizeof(float); }
  Ln.run(RPR_K_GEMM, 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), [&] { return launch_gemm_h2(g, Ln.s); },
         &g.kernel_cls);
}

// C[M, N] = act(A[M, K] B[N, K]^T) (+ resid): exact fp32 MFMA, or (split-precision mode) f16x2 planes with dynamic scales
// one bf16 plane per operand, one MFMA per product (RPR_PREC_BF16)
// bf16 outputs of the epilogue (256 x 256 kernel only; see GemmH2Args::out_b): rows [M][N], transposed [N][ldt], optional mask
struct BOut { void* rows = nullptr; void* tr = nullptr; int ldt = 0; const float* mask = nullptr; };
// a product whose result may leave its epilogue as bf16 operands: whole 256 x 256 tiles, and as many of them as send a bf16
// product to the ping-pong kernel anyway (launch_gemm_h2: 200) — the fusion never changes which kernel computes the product,
// so the gradients are the conversion launches' bit for bit (RPR_TRAIN_FUSE_FF=0, development builds, selects those).
// Measured, t5-base bz 128 (profiles/r06f_train_fuse_ab.txt): 23.2 -> 22.6 ms per step; the conversion launches of the
// feed-forward blocks were 1.8 ms, the two extra output formats cost the 48 producing launches 0.85 ms of epilogue
// (32 instead of 16 store instructions per lane and strip).
bool fused_ok(const rpr_ctx* c, int M, int N) {
  static const bool on = [] { const char* e = dev_getenv("RPR_TRAIN_FUSE_FF"); return !(e && atoi(e) == 0); }();
  return on && c->precision == RPR_PREC_BF16 && M % 256 == 0 && N % 256 == 0 && (long)(M / 256) * (N / 256) >= 200;
}
void gemm_bf16(Launcher& Ln, const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K, const float* resid,
               int relu, DevBuf* part = nullptr, bool whole_k = false, const BOut* bo = nullptr) {
  GemmH2Args g{};
  g.A = reinterpret_cast<const __half*>(A); g.lda = lda; g.W = reinterpret_cast<const __half*>(B); g.ldw = ldb;
  g.resid = resid; g.ldr = ldc;
  g.out[0] = g.out[1] = g.out[2] = C; g.ldo[0] = g.ldo[1] = g.ldo[2] = ldc; g.split_n = N;
  g.M = M; g.N = N; g.K = K; g.relu = relu; g.acc_scale = 1.0f; g.bf16 = 1;
  if (bo) { g.out_b = bo->rows; g.ldob = N; g.out_bt = bo->tr; g.ldobt = bo->ldt; g.mask_src = bo->mask; g.ldmask = N; }
  if (!part) part = &Ln.c->tws->part;
  if (whole_k) g.prefer_pp = 1;
  else { g.part = P<float>(*part); g.part_cap = part->cap / sizeof(float); }
  Ln.run(RPR_K_GEMM, 2.0 * M * (double)N * K, 2.0 * ((double)M * K + (double)N * K) + 4.0 * (double)M * N, [&] { return launch_gemm_h2(g, Ln.s); },
         &g.kernel_cls);
}

// save_xt (bf16 mode): where to leave the transposed copy [K][pad64(M)] (row stride ldT(M)) of A for the weight-gradient product
// a_ready (bf16 mode): A's bf16 rows — and its transposed copy in save_xt — were written by the epilogue that produced A
void gemm(Launcher& Ln, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
          const float* resid = nullptr, int relu = 0, void* save_xt = nullptr, const void* a_ready = nullptr) {
  if (Ln.c->precision == RPR_PREC_BF16) {
    // the reference's bf16 autocast (main.py:152 bf16=args.use_fp16; tasks/trainer.py:229): operands rounded to bf16, fp32
    // accumulation. One conversion pass per operand, no maxima, no second plane.
    TrainWs& w = *Ln.c->tws;
    if (lda != K || ldb != K) { Ln.err = RPR_ERR_INVALID; return; }
    hipStream_t s = Ln.s;
    const void* ab = a_ready ? a_ready : w.tA.p;
    if (a_ready) {}
    else if (save_xt) Ln.run(RPR_K_OTHER, 0, 8.0 * M * K, [&] { return launch_to_bf16_T(A, M, K, lda, pad64(M), save_xt, w.tA.p, s, nullptr, ldT(M)); });
    else Ln.run(RPR_K_OTHER, 0, 6.0 * M * K, [&] { return launch_to_bf16(A, M, K, lda, w.tA.p, s); });
    const void* wb = w.wT.p;
    auto it = w.wc_off.find(B);
    if (it != w.wc_off.end() && w.wc.p) wb = reinterpret_cast<const __half*>(w.wc.p) + it->second;   // converted once per step
    else Ln.run(RPR_K_OTHER, 0, 6.0 * N * K, [&] { return launch_to_bf16(B, N, K, ldb, w.wT.p, s); });
    gemm_bf16(Ln, ab, K, wb, K, C, ldc, M, N, K, resid, relu);
    return;
  }
  if (Ln.c->precision == RPR_PREC_F16X2) {
    TrainWs& w = *Ln.c->tws;
    float* am = amax_slots(Ln.c, 2);
    if (!am || lda != K || ldb != K) { Ln.err = RPR_ERR_INVALID; return; }
    __half *pa = P<__half>(w.tA), *pb = P<__half>(w.wT);
    hipStream_t s = Ln.s;
    Ln.run(RPR_K_OTHER, 0, 4.0 * (M + N) * K, [&] { return launch_absmax2(A, (size_t)M * K, B, (size_t)N * K, am, s); });
    w.site_amax[B] = am;
    Ln.run(RPR_K_OTHER, 0, 8.0 * M * K, [&] { return launch_split_dyn(A, M, K, lda, pa, am, s); });
    Ln.run(RPR_K_OTHER, 0, 8.0 * N * K, [&] { return launch_split_dyn(B, N, K, ldb, pb, am + 1, s); });
    gemm_planes(Ln, {pa, (size_t)M * K, K, am}, {pb, (size_t)N * K, K, am + 1}, C, ldc, M, N, K, resid, relu);
    return;
  }
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = B; g.ldw = ldb; g.resid = resid; g.ldr = ldc;
  g.out[0] = g.out[1] = g.out[2] = C; g.ldo[0] = g.ldo[1] = g.ldo[2] = ldc; g.split_n = N;
  g.M = M; g.N = N; g.K = K; g.relu = relu;
  Ln.run(RPR_K_GEMM, 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), [&] { return launch_gemm(g, Ln.s); });
}

// Saved X^T operands (bf16 mode): encoder layer i holds [qkv | o | wi | wo], then the cross K/V projection, then decoder
// layer i [qkv | o | xq | xo | wi | wo]; a slot is [K][pad64(rows)] bf16 with row stride ldT(rows).
enum XtSite { XT_QKV, XT_O, XT_XQ, XT_XO, XT_WI, XT_WO };
struct XtLayout {
  size_t Tp, Rp, enc_layer, dec_layer, xkv_off, dec_off, total;
  int dm, inner, dff;
  explicit XtLayout(const Dims& D) : Tp(ldT(D.T)), Rp(ldT(D.R)), dm(D.dm), inner(D.inner), dff(D.dff) {   // Tp, Rp: row strides
    enc_layer = Tp * (size_t)(2 * dm + inner + dff);
    dec_layer = Rp * (size_t)(3 * dm + 2 * inner + dff);
    xkv_off = enc_layer * D.ne;
    dec_off = xkv_off + Tp * (size_t)dm;
    total = dec_off + dec_layer * D.nd;
  }
  size_t enc(int layer, XtSite st) const {
    const size_t o = enc_layer * layer;
    switch (st) {
      case XT_QKV: return o;
      case XT_O: return o + Tp * dm;
      case XT_WI: return o + Tp * (size_t)(dm + inner);
      default: return o + Tp * (size_t)(2 * dm + inner);   // XT_WO
    }
  }
  size_t dec(int layer, XtSite st) const {
    const size_t o = dec_off + dec_layer * layer;
    switch (st) {
      case XT_QKV: return o;
      case XT_O: return o + Rp * dm;
      case XT_XQ: return o + Rp * (size_t)(dm + inner);
      case XT_XO: return o + Rp * (size_t)(2 * dm + inner);
      case XT_WI: return o + Rp * (size_t)(2 * dm + 2 * inner);
      default: return o + Rp * (size_t)(3 * dm + 2 * inner);   // XT_WO
    }
  }
};
struct XtSlots {     // null base: nothing is saved (every mode but bf16)
  __half* base; XtLayout lay;
  void* enc(int l, XtSite st) const { return base ? base + lay.enc(l, st) : nullptr; }
  void* dec(int l, XtSite st) const { return base ? base + lay.dec(l, st) : nullptr; }
  void* xkv() const { return base ? base + lay.xkv_off : nullptr; }
};

// bf16 copies of every GEMM weight, refreshed at the start of each pass (the weights change between passes: AdamW, or the
// caller writing through rpr_param_info's pointers). The table of tensors is built once per model.
int refresh_weight_cache(Launcher& Ln, rpr_ctx* c, rpr_model* m) {
  TrainWs& w = *c->tws;
  if (w.wc_model != m) {
    std::vector<WSeg> segs;
    std::vector<int> pref;
    w.wc_off.clear();
    int tiles = 0;
    const auto& d = m->d;
    const int dm = d.d_model, inner = m->inner(), dff = d.d_ff, nd = d.num_decoder_layers;
    auto add = [&](const float* p, int R, int C) {
      size_t off = (size_t)-1;
      for (const auto& pr : m->params) if (pr.ptr == p) off = pr.offset;
      if (off == (size_t)-1 || (R & 1) || (C & 3)) return;
      segs.push_back(WSeg{p, R, C, (unsigned long long)off});
      pref.push_back(tiles);
      tiles += ((R + 63) / 64) * ((C + 63) / 64);
      w.wc_off[p] = off;
    };
    for (int i = 0; i < d.num_layers; ++i) {
      add(m->enc_qkv[i], 3 * inner, dm); add(m->enc_o[i], dm, inner); add(m->enc_wi[i], dff, dm); add(m->enc_wo[i], dm, dff);
    }
    add(d.dec_xkv, nd * 2 * inner, dm);
    for (int i = 0; i < nd; ++i) {
      add(m->dec_qkv[i], 3 * inner, dm); add(m->dec_o[i], dm, inner); add(m->dec_xq[i], inner, dm); add(m->dec_xo[i], dm, inner);
      add(m->dec_wi[i], dff, dm); add(m->dec_wo[i], dm, dff);
    }
    int e = tensure(c, w.wseg, segs.size() * sizeof(WSeg));
    if (!e) e = tensure(c, w.wpref, pref.size() * sizeof(int));
    if (!e) e = tensure(c, w.wc, m->params_total * sizeof(__half));
    if (!e) e = tensure(c, w.wcT, m->params_total * sizeof(__half));
    if (e) { w.wc_off.clear(); return e; }
    RPR_HIP(hipMemcpyAsync(w.wseg.p, segs.data(), segs.size() * sizeof(WSeg), hipMemcpyHostToDevice, Ln.s));
    RPR_HIP(hipMemcpyAsync(w.wpref.p, pref.data(), pref.size() * sizeof(int), hipMemcpyHostToDevice, Ln.s));
    RPR_HIP(hipStreamSynchronize(Ln.s));              // the host vectors go out of scope
    w.wc_model = m; w.wc_nseg = (int)segs.size(); w.wc_tiles = tiles;
  }
  Ln.run(RPR_K_OTHER, 0, 8.0 * (double)m->params_total, [&] {
    return launch_weights_bf16(P<WSeg>(w.wseg), P<int>(w.wpref), w.wc_nseg, w.wc_tiles, w.wc.p, w.wcT.p, Ln.s);
  });
  return Ln.err;
}

struct Bwd {
  Launcher& Ln; rpr_ctx* c; TrainWs& w; const Dims& D;
  // RPR_TRAIN_DW_SPLITK=1: one side stream, split-K over the rows + a reduce pass per weight gradient; =0: NSIDE streams,
  // one K-loop per tile
  // Measured, t5-base bz 128: bf16 32.4 ms whole-K on 2-4 streams vs 32.7 split-K on one; f16x2 56.3 vs 52.3 (a lone
  // 256 x 256 block walks 256 K-tiles of the two-plane operands in 670 us and the side streams fall behind the main chain).
  // And once the search path's two CU-masked lane streams exist in the process, a second side stream lands on the main
  // stream's hardware queue and the step serialises (bf16 50 ms). The split-K route on ONE side stream does not depend on
  // how the runtime maps streams to queues: it is the default, RPR_TRAIN_DW_SPLITK=0 selects the whole-K route.
  static bool whole_k() {
    static const int v = [] { const char* e = dev_getenv("RPR_TRAIN_DW_SPLITK"); return e ? atoi(e) : 1; }();
    return v == 0;
  }
  static int side_streams() {   // streams the whole-K products rotate over (the scratch sets always rotate over NSIDE)
    static const int v = [] { const char* e = dev_getenv("RPR_TRAIN_SIDE_STREAMS"); const int n = e ? atoi(e) : 2;
                              return n < 1 ? 1 : (n > TrainWs::NSIDE ? TrainWs::NSIDE : n); }();
    return whole_k() ? v : 1;
  }
  // bf16 mode: the weight gradients of a layer as one grouped launch (gemm_h2_pp_group_kernel); RPR_TRAIN_DW_GROUP=0 selects
  // the per-product routes above. Measured, t5-base bz 128: see DESIGN.md section 9.
  static bool grouped() {
    static const int v = [] { const char* e = dev_getenv("RPR_TRAIN_DW_GROUP"); return e ? atoi(e) : 1; }();
    return v != 0;
  }
  // enqueue the products collected since the last flush on the side stream; the main stream goes on
  void flush_group() {
    if (w.grp.n == 0 || Ln.err) return;
    static const bool side_on = [] { const char* e = dev_getenv("RPR_TRAIN_SIDE"); return !(e && atoi(e) == 0); }();   // 0: in line on the main stream (diagnostic)
    hipStream_t s = Ln.s, side = side_on ? w.side[0] : Ln.s;
    const int gs = w.gset;
    if (hipEventRecord(w.ev_gfork[gs], s) != hipSuccess || hipStreamWaitEvent(side, w.ev_gfork[gs], 0) != hipSuccess) { Ln.err = RPR_ERR_HIP; return; }
    Launcher L2{c, side};
    const GemmGroupArgs grp = w.grp;
    L2.run(RPR_K_GEMM, w.grp_flops, w.grp_bytes, [&] { return launch_gemm_h2_group(grp, w.gtab.p, side); });
    if (L2.err) { Ln.err = L2.err; return; }
    if (hipEventRecord(w.ev_gdone[gs], side) != hipSuccess) { Ln.err = RPR_ERR_HIP; return; }
    w.gdone_pending[gs] = true;
    w.grp.n = 0; w.grp_tiles = 0; w.grp_flops = w.grp_bytes = 0; w.dyT_used = 0;
    w.gset = gs ^ 1;
    if (w.gdone_pending[w.gset]) {   // the next layer writes into the other set: its last group launch must be over
      if (hipStreamWaitEvent(s, w.ev_gdone[w.gset], 0) != hipSuccess) { Ln.err = RPR_ERR_HIP; return; }
      w.gdone_pending[w.gset] = false;
    }
  }
  // dX[M, K] = dY[M, N] W[N, K]  and  dW[N, K] = dY[M, N]^T X[M, K]  (dX may alias X: X is consumed first)
  // relu_act (bf16 mode only): dY is the gradient w.r.t. relu(.) and relu_act the stored activation; the mask is applied
  // while dY is converted (the caller skips launch_relu_bwd)
  // fuse_mask (bf16 mode, grouped route): dX is only read by the NEXT dxdw call, as its dY with relu_act = fuse_mask (the
  // feed-forward block: dX = gradient w.r.t. relu(.), [M, K]); when the shapes allow it the dX product's epilogue writes that
  // dY's bf16 rows and transposed copy itself (masked), no fp32 dX is written, and the next call finds them in w.pre
  void dxdw(const float* dY, const float* W, const float* X, float* dX, float* dW, int M, int N, int K, const void* saved_xt = nullptr,
            const float* relu_act = nullptr, const float* fuse_mask = nullptr) {
    const int Mp = pad32(M);
    hipStream_t s = Ln.s;
    if (c->precision == RPR_PREC_BF16) {
      // bf16 operands (see gemm()): one read of dY gives its plain and its transposed copy; dW on the side stream.
      // The bf16 kernel walks K in tiles of 64: the reduction length of the dW product (the rows) is padded to 64.
      const int Mp = (M + 63) & ~63;
      auto wit_g = w.wc_off.find(W);
      const int Ml = ldT(M);
      const size_t need = (((size_t)N * Ml * sizeof(__half)) + 255) & ~(size_t)255;
      const int ptiles = ((N + 255) / 256) * ((K + 255) / 256);
      // a dY the previous call's dX product left converted (rows + transposed copy in this group's set): nothing to convert,
      // and the group has room (checked when the slot was reserved)
      const bool pre = w.pre.dY == dY && w.pre.gset == w.gset && grouped() && saved_xt && wit_g != w.wc_off.end() && w.wcT.p;
      if (w.pre.dY && !pre) { Ln.err = RPR_ERR_INVALID; set_error("dxdw: a pre-converted gradient was not consumed by the next product"); return; }
      if (!pre && grouped() && saved_xt && wit_g != w.wc_off.end() && w.wcT.p && w.dyT[w.gset].p && ptiles <= GemmGroupArgs::MAX_TILES &&
          (w.grp.n == 0 || w.grp.K != Mp || w.grp.n >= GemmGroupArgs::MAXP || w.grp_tiles + ptiles > GemmGroupArgs::MAX_TILES ||
           w.dyT_used + need > w.dyT[w.gset].cap))
        flush_group();   // the product does not fit the group being collected: send that one off, start the next
      if (pre || (grouped() && saved_xt && wit_g != w.wc_off.end() && w.wcT.p && w.dyT[w.gset].p && w.dyT_used + need <= w.dyT[w.gset].cap &&
                  ptiles <= GemmGroupArgs::MAX_TILES)) {
        // grouped route: dY^T into this layer's set, the product into the group, dX on the main stream at once
        const void* py = w.tA.p;
        __half* pyt;
        if (pre) { py = w.pre.py; pyt = w.pre.pyt; w.pre = TrainWs::Pre{}; }
        else {
          pyt = reinterpret_cast<__half*>(static_cast<char*>(w.dyT[w.gset].p) + w.dyT_used);
          w.dyT_used += need;
          void* pyw = w.tA.p;
          Ln.run(RPR_K_OTHER, 0, (relu_act ? 12.0 : 8.0) * M * N, [&] { return launch_to_bf16_T(dY, M, N, N, Mp, pyt, pyw, s, relu_act, Ml); });
        }
        GemmGroupArgs& gp = w.grp;
        const int i = gp.n++;
        gp.A[i] = pyt; gp.W[i] = reinterpret_cast<const __half*>(saved_xt); gp.out[i] = dW;
        gp.M[i] = N; gp.N[i] = K; gp.ldo[i] = K; gp.K = Mp; gp.lda = Ml; gp.ldw = Ml;
        w.grp_tiles += ptiles;
        w.grp_flops += 2.0 * N * (double)K * Mp;
        w.grp_bytes += 2.0 * ((double)N * Mp + (double)K * Mp) + 4.0 * (double)N * K;
        const void* pwt = reinterpret_cast<const __half*>(w.wcT.p) + wit_g->second;
        // the next product (dY = this dX, [M, K]) joins the same group: its dY^T slot is reserved now and this product's
        // epilogue fills it, with the rows for its dX product beside it
        const size_t need2 = (((size_t)K * Ml * sizeof(__half)) + 255) & ~(size_t)255;
        const int ptiles2 = ((K + 255) / 256) * ((N + 255) / 256);    // the next product is dW2[K, N2]; N2 is not known here: bound by this one's N
        if (fuse_mask && fused_ok(c, M, K) && Mp == M && w.bfb.p && w.bfb.cap >= (size_t)M * K * sizeof(__half) && w.grp.n < GemmGroupArgs::MAXP &&
            w.grp_tiles + ptiles2 <= GemmGroupArgs::MAX_TILES && w.dyT_used + need2 <= w.dyT[w.gset].cap) {
          __half* pyt2 = reinterpret_cast<__half*>(static_cast<char*>(w.dyT[w.gset].p) + w.dyT_used);
          w.dyT_used += need2;
          const BOut bo{w.bfb.p, pyt2, Ml, fuse_mask};
          gemm_bf16(Ln, py, N, pwt, N, nullptr, K, M, K, N, nullptr, 0, nullptr, false, &bo);
          w.pre.dY = dX; w.pre.py = w.bfb.p; w.pre.pyt = pyt2; w.pre.gset = w.gset;
          return;
        }
        gemm_bf16(Ln, py, N, pwt, N, dX, K, M, K, N, nullptr, 0);
        return;
      }
      flush_group();   // (a product the group cannot take: keep the order of the side stream's work)
      const int f = w.flip; w.flip = (w.flip + 1) % TrainWs::NSIDE;
      hipStream_t side = whole_k() ? w.side[f % side_streams()] : w.side[0];
      void *py = w.tA.p, *pyt = w.tC[f].p;
      const void *pxt = w.tB[f].p, *pwt = w.wT.p;
      if (w.done_pending[f]) {
        if (hipStreamWaitEvent(s, w.ev_done[f], 0) != hipSuccess) { Ln.err = RPR_ERR_HIP; return; }
        w.done_pending[f] = false;
      }
      Ln.run(RPR_K_OTHER, 0, (relu_act ? 12.0 : 8.0) * M * N, [&] { return launch_to_bf16_T(dY, M, N, N, Mp, pyt, py, s, relu_act); });
      if (saved_xt) pxt = saved_xt;                       // left behind by the forward pass
      else Ln.run(RPR_K_OTHER, 0, 6.0 * M * K, [&] { return launch_to_bf16_T(X, M, K, K, Mp, w.tB[f].p, nullptr, s); });
      auto wit = w.wc_off.find(W);
      if (wit != w.wc_off.end() && w.wcT.p) pwt = reinterpret_cast<const __half*>(w.wcT.p) + wit->second;   // once per step
      else Ln.run(RPR_K_OTHER, 0, 6.0 * N * K, [&] { return launch_to_bf16_T(W, N, K, K, N, w.wT.p, nullptr, s); });
      if (hipEventRecord(w.ev_fork[f], s) != hipSuccess || hipStreamWaitEvent(side, w.ev_fork[f], 0) != hipSuccess) {
        Ln.err = RPR_ERR_HIP; return;
      }
      static const bool side_on_b = [] { const char* e = dev_getenv("RPR_TRAIN_SIDE"); return !(e && atoi(e) == 0); }();
      {
        Launcher L2{c, side_on_b ? side : s};
        gemm_bf16(L2, pyt, Mp, pxt, saved_xt ? Ml : Mp, dW, K, N, K, Mp, nullptr, 0, &w.part2, whole_k());
        if (L2.err) { Ln.err = L2.err; return; }
        if (hipEventRecord(w.ev_done[f], side_on_b ? side : s) != hipSuccess) { Ln.err = RPR_ERR_HIP; return; }
        w.done_pending[f] = true;
      }
      gemm_bf16(Ln, py, N, pwt, N, dX, K, M, K, N, nullptr, 0);
      return;
    }
    if (c->precision == RPR_PREC_F16X2) {
      // one read of dY gives its plain planes (for dX) and its transposed planes (for dW); W is transposed straight
      // from the fp32 weight
      float* am = amax_slots(c, 3);
      if (!am) { Ln.err = RPR_ERR_INVALID; return; }
      const int f = w.flip; w.flip = (w.flip + 1) % TrainWs::NSIDE;
      hipStream_t side = whole_k() ? w.side[f % side_streams()] : w.side[0];
      __half *py = P<__half>(w.tA), *pyt = P<__half>(w.tC[f]), *pxt = P<__half>(w.tB[f]), *pwt = P<__half>(w.wT);
      if (w.done_pending[f]) {   // the dW product that last read this scratch set (NSIDE calls ago) must be over
        if (hipStreamWaitEvent(s, w.ev_done[f], 0) != hipSuccess) { Ln.err = RPR_ERR_HIP; return; }
        w.done_pending[f] = false;
      }
      const float *am_x = am + 2, *am_w = am + 1;
      auto site = w.site_amax.find(W);
      if (site != w.site_amax.end()) {   // X and W are the forward GEMM's operands: their maxima are known
        am_x = site->second; am_w = site->second + 1;
        Ln.run(RPR_K_OTHER, 0, 4.0 * M * N, [&] { return launch_absmax2(dY, (size_t)M * N, nullptr, 0, am, s); });
      } else {
        Ln.run(RPR_K_OTHER, 0, 4.0 * M * N + 4.0 * N * K, [&] { return launch_absmax2(dY, (size_t)M * N, W, (size_t)N * K, am, s); });
        Ln.run(RPR_K_OTHER, 0, 4.0 * M * K, [&] { return launch_absmax2(X, (size_t)M * K, nullptr, 0, am + 2, s); });
      }
      Ln.run(RPR_K_OTHER, 0, 12.0 * M * N, [&] { return launch_split_dyn_T(dY, M, N, N, Mp, pyt, py, am, s); });
      Ln.run(RPR_K_OTHER, 0, 8.0 * M * K, [&] { return launch_split_dyn_T(X, M, K, K, Mp, pxt, nullptr, am_x, s); });
      Ln.run(RPR_K_OTHER, 0, 8.0 * N * K, [&] { return launch_split_dyn_T(W, N, K, K, N, pwt, nullptr, am_w, s); });
      // dW on the side stream, dX on the main one
      if (hipEventRecord(w.ev_fork[f], s) != hipSuccess || hipStreamWaitEvent(side, w.ev_fork[f], 0) != hipSuccess) {
        Ln.err = RPR_ERR_HIP; return;
      }
      static const bool side_on = [] { const char* e = dev_getenv("RPR_TRAIN_SIDE"); return !(e && atoi(e) == 0); }();
      {
        Launcher L2{c, side_on ? side : s};
        gemm_planes(L2, {pyt, (size_t)N * Mp, Mp, am}, {pxt, (size_t)K * Mp, Mp, am_x}, dW, K, N, K, Mp, nullptr, 0, &w.part2, whole_k());
        if (L2.err) { Ln.err = L2.err; return; }
        if (hipEventRecord(w.ev_done[f], side_on ? side : s) != hipSuccess) { Ln.err = RPR_ERR_HIP; return; }
        w.done_pending[f] = true;
      }
      gemm_planes(Ln, {py, (size_t)M * N, N, am}, {pwt, (size_t)K * N, N, am_w}, dX, K, M, K, N, nullptr, 0);
      return;
    }
    Ln.run(RPR_K_OTHER, 0, 8.0 * M * N, [&] { return launch_transpose_pad(dY, P<float>(w.tA), M, N, N, Mp, s); });
    Ln.run(RPR_K_OTHER, 0, 8.0 * M * K, [&] { return launch_transpose_pad(X, P<float>(w.tB[0]), M, K, K, Mp, s); });
    gemm(Ln, P<float>(w.tA), Mp, P<float>(w.tB[0]), Mp, dW, K, N, K, Mp);
    Ln.run(RPR_K_OTHER, 0, 8.0 * N * K, [&] { return launch_transpose_pad(W, P<float>(w.wT), N, K, K, N, s); });
    gemm(Ln, dY, N, P<float>(w.wT), N, dX, K, M, K, N);
  }
  void norm(const float* x, const float* ln, int rows, float post = 1.0f) {   // recompute the normalised input into w.h
    Ln.run(RPR_K_RMSNORM, 0, 8.0 * rows * D.dm, [&] { return launch_rmsnorm(x, ln, P<float>(w.h), rows, D.dm, D.eps, Ln.s, post); });
  }
  // The layer-norm weight gradient of a site is the column sum of its blocks' partials: the sites of a layer keep their
  // partials in separate regions of w_part and are summed by ONE launch at the end of the layer (flush_norms; 62 launches
  // of 10 us per step before)
  ColsumSites cs = {};
  void norm_bwd(const float* x, const float* ln, const float* dh, const float* dres, float* dx_out, float* dln, int rows,
                float post = 1.0f) {
    if (cs.n == ColsumSites::MAXS) flush_norms();
    const size_t region = ((size_t)(std::max(D.R, D.T) + 3) / 4) * D.dm;
    float* part = P<float>(w.w_part) + (size_t)cs.n * region;
    Ln.run(RPR_K_RMSNORM, 0, 16.0 * rows * D.dm, [&] {
      return launch_rmsnorm_bwd(x, ln, dh, dres, dx_out, part, nullptr, rows, D.dm, D.eps, post, 0, Ln.s);
    });
    cs.part[cs.n] = part; cs.out[cs.n] = dln; cs.nparts[cs.n] = rmsnorm_bwd_parts(rows); ++cs.n;
  }
  void flush_norms() {
    if (cs.n == 0 || Ln.err) return;
    const ColsumSites p = cs;
    Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_colsum_multi(p, D.dm, Ln.s); });
    cs.n = 0;
  }
};

int alloc_train(rpr_ctx* c, const rpr_model* m, const Dims& D) {
  if (!c->tws) c->tws = new TrainWs();
  TrainWs& w = *c->tws;
  const size_t f = sizeof(float), R = D.R, T = D.T, dm = D.dm, inner = D.inner, dff = D.dff;
  const size_t rows = std::max(R, T), rp = ((int)rows + 63) & ~63;   // rows padded to the K-tile of the bf16 kernel (64) / the split kernel (32)
  const size_t wide = std::max<size_t>(std::max<size_t>(dff, 3 * inner), (size_t)D.xld);
  int e = 0;
  auto E = [&](DevBuf& b, size_t bytes) { if (!e) e = tensure(c, b, bytes); };
  E(w.enc_act, (size_t)D.ne * D.enc_stride * f + T * dm * f);     // + the encoder's last stream
  E(w.dec_act, (size_t)D.nd * D.dec_stride * f);
  E(w.enc_out, T * dm * f); E(w.xkv, T * (size_t)D.xld * f); E(w.x_last, R * dm * f);
  E(w.scores, R * f); E(w.margins, 8 * (size_t)D.bz * f); E(w.dscores, R * f);
  E(w.in_idx, R * 4); E(w.out_idx, R * 4); E(w.tok_idx, T * 4);
  E(w.h, rows * dm * f); E(w.dxa, rows * dm * f); E(w.dxb, rows * dm * f); E(w.dbig, rows * wide * f);
  E(w.dattn, rows * inner * f); E(w.dxkv, T * (size_t)D.xld * f); E(w.denc, T * dm * f);
  E(w.tA, wide * rp * f);
  // the transposed operands of a weight-gradient product (fp32, two f16 planes or bf16), one set per side stream
  for (int i = 0; i < TrainWs::NSIDE; ++i) { E(w.tB[i], wide * rp * f); E(w.tC[i], wide * rp * f); }
  if (c->precision == RPR_PREC_BF16) {
    E(w.xT, XtLayout(D).total * sizeof(__half));
    if (Bwd::grouped()) {
      // dY^T of one layer's products: [N_out][pad64(rows)] bf16 each, 256-byte aligned
      const size_t Rp = ldT(D.R), Tp = ldT(D.T);
      const size_t dec = (3 * dm + dff + 4 * inner) * Rp, enc = (2 * dm + dff + 3 * inner) * Tp, xkv = (size_t)D.xld * Tp;
      const size_t need = std::max(std::max(dec, enc), xkv) * sizeof(__half) + 8 * 256;
      E(w.dyT[0], need); E(w.dyT[1], need);
      E(w.bfb, rp * dff * sizeof(__half));
      static_assert(GemmGroupArgs::MAXP * sizeof(GemmH2Args) <= GemmGroupArgs::TABLE_BYTES, "argument table");
      E(w.gtab, GemmGroupArgs::SCRATCH_BYTES);
    }
  }
  E(w.wT, std::max<size_t>(std::max<size_t>(dff * dm, 3 * inner * dm), (size_t)D.xld * dm) * f);
  E(w.w_part, ColsumSites::MAXS * ((rows + 3) / 4) * dm * f);   // the partials of up to four norm sites (Bwd::norm_bwd)
  E(w.bias_part, std::max<size_t>((size_t)D.S, (size_t)D.bz) * D.H * D.buckets * f);
  E(w.fix, std::max<size_t>((size_t)m->d.vocab_size, (size_t)m->d.L * D.V) * dm * 8);
  E(w.gn_part, 1024 * 8); E(w.gn_out, 16); E(w.amax, AMAX_SLOTS * f); E(w.part, (size_t)16 << 20 << 2); E(w.part2, (size_t)16 << 20 << 2);   // split-K partials: 16 M floats per stream
  if (!e && !w.side[0]) {
    for (int i = 0; i < TrainWs::NSIDE; ++i) {
      if (i < Bwd::side_streams()) RPR_HIP(hipStreamCreateWithFlags(&w.side[i], hipStreamNonBlocking));   // only the streams in use
      RPR_HIP(hipEventCreateWithFlags(&w.ev_fork[i], hipEventDisableTiming));
      RPR_HIP(hipEventCreateWithFlags(&w.ev_done[i], hipEventDisableTiming));
    }
    for (int i = 0; i < 2; ++i) {
      RPR_HIP(hipEventCreateWithFlags(&w.ev_gfork[i], hipEventDisableTiming));
      RPR_HIP(hipEventCreateWithFlags(&w.ev_gdone[i], hipEventDisableTiming));
    }
  }
  return e;
}

// saved activations of one layer (floats, row-major)
struct EncAct { float *x, *qkv, *attn, *xm, *ff; };
struct DecAct { float *x0, *qkv, *a0, *x1, *qx, *a1, *x2, *ff; };
EncAct enc_act(const TrainWs& w, const Dims& D, int i) {
  float* b = P<float>(w.enc_act) + (size_t)i * D.enc_stride;
  const size_t T = D.T;
  EncAct a; a.x = b; a.qkv = a.x + T * D.dm; a.attn = a.qkv + T * 3 * D.inner; a.xm = a.attn + T * D.inner; a.ff = a.xm + T * D.dm;
  return a;
}
DecAct dec_act(const TrainWs& w, const Dims& D, int i) {
  float* b = P<float>(w.dec_act) + (size_t)i * D.dec_stride;
  const size_t R = D.R;
  DecAct a; a.x0 = b; a.qkv = a.x0 + R * D.dm; a.a0 = a.qkv + R * 3 * D.inner; a.x1 = a.a0 + R * D.inner; a.qx = a.x1 + R * D.dm;
  a.a1 = a.qx + R * D.inner; a.x2 = a.a1 + R * D.inner; a.ff = a.x2 + R * D.dm;
  return a;
}

void forward(Launcher& Ln, rpr_ctx* c, const rpr_model* m, const Dims& D, const int32_t* ids, const int32_t* mask,
             const int32_t* codes, int32_t* last) {
  TrainWs& w = *c->tws;
  const auto& d = m->d;
  hipStream_t s = Ln.s;
  const int T = D.T, R = D.R, dm = D.dm, inner = D.inner, dff = D.dff;
  float* h = P<float>(w.h);
  auto norm = [&](const float* x, const float* ln, float* out, int rows, float post = 1.0f) {
    Ln.run(RPR_K_RMSNORM, 0, 8.0 * rows * dm, [&] { return launch_rmsnorm(x, ln, out, rows, dm, D.eps, s, post); });
  };
  const XtSlots xt{c->precision == RPR_PREC_BF16 ? P<__half>(w.xT) : nullptr, XtLayout(D)};
  // out = act(norm(x) W^T): in bf16 mode the norm writes the product's bf16 operand and its transposed copy itself
  // (rmsnorm_bf16_T_kernel); otherwise norm into h, then gemm() converts. RPR_TRAIN_NORM_FUSE=0: the two-kernel route.
  static const bool norm_fuse = [] { const char* e = dev_getenv("RPR_TRAIN_NORM_FUSE"); return !(e && atoi(e) == 0); }();
  // next_xt: where the NEXT product (the one that reads C) wants C's transposed bf16 copy; when the shapes allow it (fused_ok)
  // this product's epilogue writes it, and C's bf16 rows into w.bfb: returns those rows (the next gemm()'s a_ready) or null
  auto norm_gemm = [&](const float* x, const float* ln, const float* W, float* C, int rows, int N, int relu, void* save_xt,
                       void* next_xt = nullptr) -> const void* {
    auto it = w.wc_off.find(W);
    if (norm_fuse && c->precision == RPR_PREC_BF16 && save_xt && it != w.wc_off.end() && w.wc.p && dm <= 1024 && (dm & 63) == 0) {
      Ln.run(RPR_K_RMSNORM, 0, 8.0 * rows * dm, [&] {
        return launch_rmsnorm_bf16_T(x, ln, rows, dm, D.eps, 1.0f, w.tA.p, save_xt, pad64(rows), ldT(rows), s);
      });
      const bool fuse = next_xt && fused_ok(c, rows, N) && w.bfb.p && w.bfb.cap >= (size_t)rows * N * sizeof(__half) && pad64(rows) == rows;
      const BOut bo{w.bfb.p, next_xt, ldT(rows), nullptr};
      gemm_bf16(Ln, w.tA.p, dm, reinterpret_cast<const __half*>(w.wc.p) + it->second, dm, C, N, rows, N, dm, nullptr, relu, nullptr, false,
                fuse ? &bo : nullptr);
      return fuse ? w.bfb.p : nullptr;
    }
    norm(x, ln, h, rows);
    gemm(Ln, h, dm, W, dm, C, N, rows, N, dm, nullptr, relu, save_xt);
    return nullptr;
  };
  Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_mask_lengths(mask, last, D.bz, D.Lq, s, c->status + 1); });
  // ---- encoder over the padded [bz, Lq] layout (padded positions get no gradient: nothing downstream reads them)
  float* xe_last = P<float>(w.enc_act) + (size_t)D.ne * D.enc_stride;
  Ln.run(RPR_K_OTHER, 0, 8.0 * T * dm, [&] { return launch_embed_rows(d.shared, ids, enc_act(w, D, 0).x, T, dm, d.vocab_size, s); });
  for (int i = 0; i < D.ne; ++i) {
    EncAct a = enc_act(w, D, i);
    float* xnext = i + 1 < D.ne ? enc_act(w, D, i + 1).x : xe_last;
    norm_gemm(a.x, m->enc_ln0[i], m->enc_qkv[i], a.qkv, T, 3 * inner, 0, xt.enc(i, XT_QKV));
    EncAttnArgs ea{a.qkv, mask, d.enc_rel_bias, m->enc_bucket, a.attn, D.bz, D.Lq, D.H, d.rel_buckets, nullptr, 0, nullptr, nullptr,
                   nullptr, 0, 1};
    Ln.run(RPR_K_ENC_ATTN, 0, 0, [&] { return launch_enc_attn(ea, s); });
    gemm(Ln, a.attn, inner, m->enc_o[i], inner, a.xm, dm, T, dm, inner, a.x, 0, xt.enc(i, XT_O));
    const void* ffb = norm_gemm(a.xm, m->enc_ln1[i], m->enc_wi[i], a.ff, T, dff, 1, xt.enc(i, XT_WI), xt.enc(i, XT_WO));
    gemm(Ln, a.ff, dff, m->enc_wo[i], dff, xnext, dm, T, dm, dff, a.xm, 0, xt.enc(i, XT_WO), ffb);
  }
  norm(xe_last, d.enc_final_ln, P<float>(w.enc_out), T);
  gemm(Ln, P<float>(w.enc_out), dm, d.dec_xkv, dm, P<float>(w.xkv), D.xld, T, D.xld, dm, nullptr, 0, xt.xkv());
  // ---- teacher-forced decoder over all positions of the positive and the negative smtid
  Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_train_indices(codes, P<int32_t>(w.in_idx), P<int32_t>(w.out_idx), D.S, D.L, D.V, s); });
  Ln.run(RPR_K_OTHER, 0, 8.0 * R * dm, [&] { return launch_train_dec_embed(d.start_embed, d.in_embeds, codes, dec_act(w, D, 0).x0, D.S, D.L, dm, D.V, s); });
  for (int i = 0; i < D.nd; ++i) {
    DecAct a = dec_act(w, D, i);
    float* xnext = i + 1 < D.nd ? dec_act(w, D, i + 1).x0 : P<float>(w.x_last);
    norm_gemm(a.x0, m->dec_ln0[i], m->dec_qkv[i], a.qkv, R, 3 * inner, 0, xt.dec(i, XT_QKV));
    EncAttnArgs sa{a.qkv, nullptr, d.dec_rel_bias, m->dec_bucket, a.a0, D.S, D.L, D.H, d.rel_buckets, nullptr, 0, nullptr, nullptr,
                   nullptr, 1, 1};
    Ln.run(RPR_K_ENC_ATTN, 0, 0, [&] { return launch_enc_attn(sa, s); });
    gemm(Ln, a.a0, inner, m->dec_o[i], inner, a.x1, dm, R, dm, inner, a.x0, 0, xt.dec(i, XT_O));
    norm_gemm(a.x1, m->dec_ln1[i], m->dec_xq[i], a.qx, R, inner, 0, xt.dec(i, XT_XQ));
    const float* xk = P<float>(w.xkv) + (size_t)i * 2 * inner;
    DecCrossAttnArgs ca{a.qx, xk, xk + inner, D.xld, mask, a.a1, D.bz, 2 * D.L, D.H, D.Lq, nullptr, 0, last, nullptr, 0, nullptr};
    // (the query's 2 L decoder rows as 32-row tiles on the fp32-MFMA kernel of the search tail when Lq <= 64)
    Ln.run(RPR_K_DEC_CROSS_ATTN, 0, 0, [&] { return launch_tail_cross_attn(ca, s); });
    gemm(Ln, a.a1, inner, m->dec_xo[i], inner, a.x2, dm, R, dm, inner, a.x1, 0, xt.dec(i, XT_XO));
    const void* ffb = norm_gemm(a.x2, m->dec_ln2[i], m->dec_wi[i], a.ff, R, dff, 1, xt.dec(i, XT_WI), xt.dec(i, XT_WO));
    gemm(Ln, a.ff, dff, m->dec_wo[i], dff, xnext, dm, R, dm, dff, a.x2, 0, xt.dec(i, XT_WO), ffb);
  }
  Ln.run(RPR_K_OTHER, 0, 0, [&] {
    return launch_gold_scores(P<float>(w.x_last), d.dec_final_ln, d.out_embeds, codes, P<float>(w.scores), D.S, D.L, dm, D.V, D.eps,
                              D.post, s);
  });
}

// Gradient buckets for the data-parallel exchange (reference: DDP's bucketed all-reduce overlapped with the backward pass,
// tasks/trainer.py:486 wraps the model in DistributedDataParallel): one bucket per transformer layer — its nine (six)
// tensors are contiguous in the flat buffer — handed over the moment the layer's last gradient kernel has been ENQUEUED,
// plus one for everything that is only final at the end (embeddings, codebooks, cross K/V, final norms, bias tables).
// "Handed over" = the caller's communication stream is made to wait for the layer's producers on the main stream and on
// the weight-gradient side stream, then the host callback runs: it enqueues the bucket's all-reduce on that stream, where
// it overlaps the rest of the backward pass.
struct BucketHook {
  rpr_grad_bucket_cb cb; void* user; hipStream_t comm; int next = 0;
};

void backward(Launcher& Ln, rpr_ctx* c, rpr_model* m, const Dims& D, const int32_t* ids, const int32_t* mask, float* G,
              BucketHook* hook = nullptr) {
  TrainWs& w = *c->tws;
  const auto& d = m->d;
  hipStream_t s = Ln.s;
  auto bucket = [&](size_t off, size_t numel) {
    if (!hook || !hook->cb || Ln.err) return;
    while ((int)w.bucket_ev.size() < 2 * (hook->next + 1)) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { Ln.err = RPR_ERR_HIP; return; }
      w.bucket_ev.push_back(e);
    }
    hipEvent_t e0 = w.bucket_ev[(size_t)2 * hook->next], e1 = w.bucket_ev[(size_t)2 * hook->next + 1];
    ++hook->next;
    (void)e1;
    if (hipEventRecord(e0, s) != hipSuccess || hipStreamWaitEvent(hook->comm, e0, 0) != hipSuccess) { Ln.err = RPR_ERR_HIP; return; }
    // ... and for the weight gradients in flight: the last product of every side stream (stream order covers the earlier ones)
    for (int i = 0; i < TrainWs::NSIDE; ++i)
      if (w.done_pending[i] && hipStreamWaitEvent(hook->comm, w.ev_done[i], 0) != hipSuccess) { Ln.err = RPR_ERR_HIP; return; }
    for (int i = 0; i < 2; ++i)     // grouped weight gradients: the launch of every set still in flight
      if (w.gdone_pending[i] && hipStreamWaitEvent(hook->comm, w.ev_gdone[i], 0) != hipSuccess) { Ln.err = RPR_ERR_HIP; return; }
    hook->cb(hook->user, (int64_t)off, (int64_t)numel);
  };
  auto layer_numel = [&](int first_kind, int last_kind, int layer) {
    size_t lo = param_offset(m, first_kind, layer), hi = param_offset(m, last_kind, layer);
    for (const auto& p : m->params) if (p.kind == last_kind && p.layer == layer) hi += p.numel;
    return std::make_pair(lo, hi - lo);
  };
  const int T = D.T, R = D.R, dm = D.dm, inner = D.inner, dff = D.dff, H = D.H;
  // a backward that aborted on Ln.err leaves collected-but-unflushed products behind: never carry them into this step's
  // gradient buffer
  w.grp.n = 0; w.grp_tiles = 0; w.grp_flops = w.grp_bytes = 0; w.dyT_used = 0; w.pre = TrainWs::Pre{};
  Bwd B{Ln, c, w, D};
  const XtSlots xt{c->precision == RPR_PREC_BF16 ? P<__half>(w.xT) : nullptr, XtLayout(D)};
  const bool saved = xt.base != nullptr;   // the normalised inputs are only recomputed for their weight-gradient products
  const bool bf16 = c->precision == RPR_PREC_BF16;   // ReLU backward folded into the conversion of its result (dxdw)
  auto g = [&](int kind, int layer = -1) { return G + param_offset(m, kind, layer); };
  float *dxa = P<float>(w.dxa), *dxb = P<float>(w.dxb), *dbig = P<float>(w.dbig), *dattn = P<float>(w.dattn), *h = P<float>(w.h);
  unsigned long long* fix = P<unsigned long long>(w.fix);

  // ---- gold scores: dscore -> (dE rows, dhF) -> final RMSNorm backward
  Ln.run(RPR_K_OTHER, 0, 0, [&] {   // dE rows go to dxb, dhF to dxa
    return launch_gold_score_bwd(P<float>(w.x_last), d.dec_final_ln, d.out_embeds, P<int32_t>(w.out_idx), P<float>(w.dscores), dxa, dxb,
                                 R, dm, D.eps, D.post, s);
  });
  float* g_out = d.out_embeds != d.in_embeds ? g(K_OUT_EMB) : g(K_IN_EMB);
  Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_scatter_rows_fix(dxb, P<int32_t>(w.out_idx), fix, R, dm, s); });
  Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_fix_flush(fix, g_out, (size_t)d.L * D.V * dm, s); });
  B.norm_bwd(P<float>(w.x_last), d.dec_final_ln, dxa, nullptr, dxb, g(K_DEC_FLN), R, D.post);
  float *dx = dxb, *dx2 = dxa;   // dx = gradient w.r.t. the current layer's output stream
  // ---- decoder layers, last to first
  Ln.run(RPR_K_OTHER, 0, 0, [&] { return hipMemsetAsync(w.dxkv.p, 0, (size_t)T * D.xld * sizeof(float), s); });
  for (int i = D.nd - 1; i >= 0; --i) {
    DecAct a = dec_act(w, D, i);
    // feed-forward: x3 = x2 + relu(norm(x2) Wi^T) Wo^T
    B.dxdw(dx, m->dec_wo[i], a.ff, dbig, g(K_DEC_WO, i), R, dm, dff, xt.dec(i, XT_WO), nullptr, bf16 ? a.ff : nullptr);
    if (!bf16) Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_relu_bwd(dbig, a.ff, (size_t)R * dff, s); });
    if (!saved) B.norm(a.x2, m->dec_ln2[i], R);
    B.dxdw(dbig, m->dec_wi[i], h, h, g(K_DEC_WI, i), R, dff, dm, xt.dec(i, XT_WI), bf16 ? a.ff : nullptr);     // dh into the (now free) h buffer
    B.norm_bwd(a.x2, m->dec_ln2[i], h, dx, dx2, g(K_DEC_LN2, i), R);
    std::swap(dx, dx2);                                            // dx = gradient w.r.t. x2
    // cross-attention: x2 = x1 + CrossAttn(norm(x1) Wq^T, Kx, Vx) Wo^T
    B.dxdw(dx, m->dec_xo[i], a.a1, dattn, g(K_DEC_XO, i), R, dm, inner, xt.dec(i, XT_XO));
    {
      const float* xk = P<float>(w.xkv) + (size_t)i * 2 * inner;
      float* dxk = P<float>(w.dxkv) + (size_t)i * 2 * inner;
      Ln.run(RPR_K_DEC_CROSS_ATTN, 0, 0, [&] {   // dq into dbig (as [R, inner])
        return launch_cross_attn_bwd(a.qx, xk, xk + inner, D.xld, mask, dattn, dbig, dxk, dxk + inner, D.bz, 2 * D.L, D.Lq, H, s);
      });
    }
    if (!saved) B.norm(a.x1, m->dec_ln1[i], R);
    B.dxdw(dbig, m->dec_xq[i], h, h, g(K_DEC_XQ, i), R, inner, dm, xt.dec(i, XT_XQ));
    B.norm_bwd(a.x1, m->dec_ln1[i], h, dx, dx2, g(K_DEC_LN1, i), R);
    std::swap(dx, dx2);                                            // dx = gradient w.r.t. x1
    // self-attention: x1 = x0 + SelfAttn(norm(x0) Wqkv^T) Wo^T
    B.dxdw(dx, m->dec_o[i], a.a0, dattn, g(K_DEC_O, i), R, dm, inner, xt.dec(i, XT_O));
    Ln.run(RPR_K_ENC_ATTN, 0, 0, [&] {
      return launch_self_attn_bwd(a.qkv, dattn, nullptr, d.dec_rel_bias, m->dec_bucket, dbig, P<float>(w.bias_part), g(K_DEC_REL), D.S,
                                  D.L, H, d.rel_buckets, 1, s);
    });
    if (!saved) B.norm(a.x0, m->dec_ln0[i], R);
    B.dxdw(dbig, m->dec_qkv[i], h, h, g(K_DEC_QKV, i), R, 3 * inner, dm, xt.dec(i, XT_QKV));
    B.norm_bwd(a.x0, m->dec_ln0[i], h, dx, dx2, g(K_DEC_LN0, i), R);
    std::swap(dx, dx2);                                            // dx = gradient w.r.t. x0 = the previous layer's output
    B.flush_group();                                               // the layer's six weight gradients: one launch on the side stream
    B.flush_norms();                                               // ... and its three layer-norm weight gradients
    { const auto b = layer_numel(K_DEC_LN0, K_DEC_WO, i); bucket(b.first, b.second); }
  }
  // decoder input embeddings: codebook rows and the start embedding
  Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_scatter_rows_fix(dx, P<int32_t>(w.in_idx), fix, R, dm, s); });
  Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_fix_flush(fix, g(K_IN_EMB), (size_t)d.L * D.V * dm, s); });
  Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_sum_selected_rows(dx, P<int32_t>(w.in_idx), g(K_START), R, dm, s); });
  // ---- cross K/V projection and the encoder's final norm
  float* denc = P<float>(w.denc);
  B.dxdw(P<float>(w.dxkv), d.dec_xkv, P<float>(w.enc_out), denc, g(K_XKV), T, D.xld, dm, xt.xkv());
  B.flush_group();
  float* xe_last = P<float>(w.enc_act) + (size_t)D.ne * D.enc_stride;
  B.norm_bwd(xe_last, d.enc_final_ln, denc, nullptr, dxa, g(K_ENC_FLN), T);
  dx = dxa; dx2 = dxb;
  for (int i = D.ne - 1; i >= 0; --i) {
    EncAct a = enc_act(w, D, i);
    B.dxdw(dx, m->enc_wo[i], a.ff, dbig, g(K_ENC_WO, i), T, dm, dff, xt.enc(i, XT_WO), nullptr, bf16 ? a.ff : nullptr);
    if (!bf16) Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_relu_bwd(dbig, a.ff, (size_t)T * dff, s); });
    if (!saved) B.norm(a.xm, m->enc_ln1[i], T);
    B.dxdw(dbig, m->enc_wi[i], h, h, g(K_ENC_WI, i), T, dff, dm, xt.enc(i, XT_WI), bf16 ? a.ff : nullptr);
    B.norm_bwd(a.xm, m->enc_ln1[i], h, dx, dx2, g(K_ENC_LN1, i), T);
    std::swap(dx, dx2);
    B.dxdw(dx, m->enc_o[i], a.attn, dattn, g(K_ENC_O, i), T, dm, inner, xt.enc(i, XT_O));
    Ln.run(RPR_K_ENC_ATTN, 0, 0, [&] {
      return launch_self_attn_bwd(a.qkv, dattn, mask, d.enc_rel_bias, m->enc_bucket, dbig, P<float>(w.bias_part), g(K_ENC_REL), D.bz,
                                  D.Lq, H, d.rel_buckets, 0, s);
    });
    if (!saved) B.norm(a.x, m->enc_ln0[i], T);
    B.dxdw(dbig, m->enc_qkv[i], h, h, g(K_ENC_QKV, i), T, 3 * inner, dm, xt.enc(i, XT_QKV));
    B.norm_bwd(a.x, m->enc_ln0[i], h, dx, dx2, g(K_ENC_LN0, i), T);
    std::swap(dx, dx2);
    B.flush_group();
    B.flush_norms();
    { const auto b = layer_numel(K_ENC_LN0, K_ENC_WO, i); bucket(b.first, b.second); }
  }
  // token embeddings (the encoder's table is the shared one)
  Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_scatter_rows_fix(dx, ids, fix, T, dm, s); });
  Ln.run(RPR_K_OTHER, 0, 0, [&] { return launch_fix_flush(fix, g(K_SHARED), (size_t)d.vocab_size * dm, s); });
  // the weight gradients still in flight on the side stream belong to this pass: join
  for (int i = 0; i < TrainWs::NSIDE; ++i)
    if (w.done_pending[i]) {
      if (hipStreamWaitEvent(s, w.ev_done[i], 0) != hipSuccess) Ln.err = RPR_ERR_HIP;
      w.done_pending[i] = false;
    }
  B.flush_group();
  B.flush_norms();                                                 // the encoder's final norm (the decoder's went with its last layer)
  for (int i = 0; i < 2; ++i)
    if (w.gdone_pending[i]) {
      if (hipStreamWaitEvent(s, w.ev_gdone[i], 0) != hipSuccess) Ln.err = RPR_ERR_HIP;
      w.gdone_pending[i] = false;
    }
  bucket(0, param_offset(m, K_ENC_LN0, 0));   // everything in front of the first layer: final only now
}

}  // namespace

void rpr::train_forget_model(rpr_ctx* c, const rpr_model* m) {
  if (c && c->tws && c->tws->wc_model == m) { c->tws->wc_model = nullptr; c->tws->wc_off.clear(); }
  if (c && c->tws && c->tws->aw_model == m) c->tws->aw_model = nullptr;
}

void rpr::free_train_ws(rpr_ctx* c) {
  if (!c->tws) return;
  TrainWs& w = *c->tws;
  DevBuf* all[] = {&w.enc_act, &w.dec_act, &w.enc_out, &w.xkv, &w.x_last, &w.scores, &w.margins, &w.dscores, &w.in_idx, &w.out_idx,
                   &w.tok_idx, &w.h, &w.dxa, &w.dxb, &w.dbig, &w.dattn, &w.dxkv, &w.denc, &w.tA, &w.wT, &w.w_part, &w.bias_part,
                   &w.fix, &w.gn_part, &w.gn_out, &w.amax, &w.part, &w.part2, &w.wc, &w.wcT, &w.wseg, &w.wpref, &w.xT, &w.aseg, &w.apref, &w.bfb};
  for (DevBuf* b : all) if (b->p) (void)hipFree(b->p);
  for (int i = 0; i < 2; ++i) {
    if (w.dyT[i].p) (void)hipFree(w.dyT[i].p);
    if (w.ev_gfork[i]) (void)hipEventDestroy(w.ev_gfork[i]);
    if (w.ev_gdone[i]) (void)hipEventDestroy(w.ev_gdone[i]);
  }
  if (w.gtab.p) (void)hipFree(w.gtab.p);
  for (int i = 0; i < TrainWs::NSIDE; ++i) {
    if (w.tB[i].p) (void)hipFree(w.tB[i].p);
    if (w.tC[i].p) (void)hipFree(w.tC[i].p);
    if (w.ev_fork[i]) (void)hipEventDestroy(w.ev_fork[i]);
    if (w.ev_done[i]) (void)hipEventDestroy(w.ev_done[i]);
    if (w.side[i]) (void)hipStreamDestroy(w.side[i]);
  }
  for (hipEvent_t e : w.bucket_ev) (void)hipEventDestroy(e);
  delete c->tws;
  c->tws = nullptr;
}

extern "C" {

int64_t rpr_param_count(rpr_model* m) { if (!m) return 0; build_params(m); return (int64_t)m->params.size(); }
int64_t rpr_param_total(rpr_model* m) { if (!m) return 0; build_params(m); return (int64_t)m->params_total; }
int rpr_param_info(rpr_model* m, int64_t index, const float** ptr, int64_t* numel, int64_t* offset) {
  RPR_REQUIRE(m, "NULL model");
  build_params(m);
  RPR_REQUIRE(index >= 0 && index < (int64_t)m->params.size(), "parameter index out of range");
  const auto& p = m->params[(size_t)index];
  if (ptr) *ptr = p.ptr;
  if (numel) *numel = (int64_t)p.numel;
  if (offset) *offset = (int64_t)p.offset;
  return RPR_OK;
}

int rpr_lngknp_backward(rpr_ctx* c, rpr_model* m, const int32_t* input_ids, const int32_t* attention_mask, int32_t bz, int32_t Lq,
                        const int32_t* doc_codes, int32_t L, const float* teacher_pos, const float* teacher_neg,
                        const int32_t* prefix_lens, int32_t n_prefix, float* out_losses, float* flat_grads, void* stream) {
  return rpr_lngknp_backward_buckets(c, m, input_ids, attention_mask, bz, Lq, doc_codes, L, teacher_pos, teacher_neg, prefix_lens,
                                     n_prefix, out_losses, flat_grads, stream, nullptr, nullptr, nullptr);
}

int rpr_lngknp_backward_buckets(rpr_ctx* c, rpr_model* m, const int32_t* input_ids, const int32_t* attention_mask, int32_t bz,
                                int32_t Lq, const int32_t* doc_codes, int32_t L, const float* teacher_pos, const float* teacher_neg,
                                const int32_t* prefix_lens, int32_t n_prefix, float* out_losses, float* flat_grads, void* stream,
                                void* comm_stream, rpr_grad_bucket_cb on_bucket, void* user) {
  RPR_REQUIRE(c && m && input_ids && attention_mask && doc_codes && teacher_pos && teacher_neg && prefix_lens && out_losses &&
                  flat_grads, "NULL argument");
  RPR_REQUIRE(m->ctx == c, "model belongs to another ctx");
  RPR_REQUIRE(bz >= 1 && Lq >= 1 && Lq <= 128, "bz or Lq out of range (the training kernels hold Lq <= 128 keys in LDS)");
  RPR_REQUIRE(m->d.d_kv == DKV, "the training kernels are written for d_kv == 64 (t5-base / t5-large)");
  RPR_REQUIRE(L >= 1 && L <= m->d.L && L <= MAX_DEC_LEN, "smtid length exceeds the model's decoder length");
  RPR_REQUIRE(n_prefix >= 1 && n_prefix <= 8, "n_prefix out of range (1..8)");
  RPR_HIP(hipSetDevice(c->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  build_params(m);
  const auto& d = m->d;
  Dims D{};
  D.bz = bz; D.Lq = Lq; D.L = L; D.S = bz * 2; D.R = D.S * L; D.T = bz * Lq; D.dm = d.d_model; D.inner = m->inner(); D.dff = d.d_ff;
  D.H = d.num_heads; D.ne = d.num_layers; D.nd = d.num_decoder_layers; D.V = d.V; D.xld = D.nd * 2 * D.inner; D.buckets = d.rel_buckets;
  D.eps = d.layer_norm_eps; D.post = d.scaleup_output_hidden ? (float)pow((double)D.dm, -0.5) : 1.0f;
  D.enc_stride = (size_t)D.T * (2 * D.dm + 4 * D.inner + D.dff);
  D.dec_stride = (size_t)D.R * (3 * D.dm + 6 * D.inner + D.dff);
  RPR_REQUIRE(self_attn_bwd_smem(std::max(L, Lq), d.rel_buckets) <= 160 * 1024 && cross_attn_bwd_smem(2 * L, Lq) <= 160 * 1024,
              "sequence lengths too large for the LDS-resident attention backward");
  int e = alloc_train(c, m, D);
  if (e) return e;
  // the forward's cross-attention kernel needs the per-query key counts: reuse the search workspace's small buffers
  e = ensure(c, c->ws.last, (size_t)bz * 4);
  if (e) return e;
  TrainWs& w = *c->tws;
  RPR_HIP(hipMemsetAsync(flat_grads, 0, m->params_total * sizeof(float), s));
  RPR_HIP(hipMemsetAsync(w.fix.p, 0, w.fix.cap, s));
  Launcher Ln{c, s};
  amax_reset(Ln);
  if (c->precision == RPR_PREC_BF16) {
    e = refresh_weight_cache(Ln, c, m);
    if (e) return e;
  } else {
    w.wc_off.clear(); w.wc_model = nullptr;   // the other modes convert per call
  }
  forward(Ln, c, m, D, input_ids, attention_mask, doc_codes, P<int32_t>(c->ws.last));
  if (Ln.err) return Ln.err;
  RPR_HIP(launch_margin_mse(P<float>(w.scores), teacher_pos, teacher_neg, prefix_lens, n_prefix, bz, L, out_losses, P<float>(w.margins), s));
  // total loss = sum of the task losses with weight 1 (reference arguments.py:109-119, trainer.py:228-240)
  RPR_HIP(launch_margin_mse_bwd(P<float>(w.margins), teacher_pos, teacher_neg, prefix_lens, n_prefix, bz, L, P<float>(w.dscores), s));
  BucketHook hook{on_bucket, user, reinterpret_cast<hipStream_t>(comm_stream)};
  backward(Ln, c, m, D, input_ids, attention_mask, flat_grads, on_bucket ? &hook : nullptr);
  return Ln.err;
}

int rpr_adamw_step(rpr_ctx* c, rpr_model* m, const float* flat_grads, float* exp_avg, float* exp_avg_sq, int64_t step, float lr,
                   float beta1, float beta2, float eps, float weight_decay, float max_grad_norm, float* out_grad_norm,
                   void* stream) {
  RPR_REQUIRE(c && m && flat_grads && exp_avg && exp_avg_sq, "NULL argument");
  RPR_REQUIRE(m->ctx == c && step >= 1, "bad model or step (1-based)");
  RPR_HIP(hipSetDevice(c->device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  build_params(m);
  if (!c->tws) c->tws = new TrainWs();
  TrainWs& w = *c->tws;
  int e = tensure(c, w.gn_part, 1024 * 8);
  if (!e) e = tensure(c, w.gn_out, 16);
  if (e) return e;
  RPR_HIP(launch_grad_norm(flat_grads, m->params_total, P<double>(w.gn_part), 1024, max_grad_norm, P<float>(w.gn_out), s));
  const float bc1 = 1.0f - (float)pow((double)beta1, (double)step);
  const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  // Weight decay follows HF 4.17's Trainer.create_optimizer (what the reference trains with, tasks/trainer.py:477):
  // every parameter decays except those of nn.LayerNorm modules and those whose name contains "bias". T5LayerNorm is
  // not an nn.LayerNorm in that version, so the T5 layer-norm weights DO decay; `relative_attention_bias.weight` is
  // excluded by its name. (Restated from memory of transformers 4.17 — its source is not available offline; the
  // reference's default weight_decay is 0, where the rule is moot.)
  static const bool per_tensor = [] { const char* e = dev_getenv("RPR_ADAMW_PER_TENSOR"); return e && atoi(e) != 0; }();
  if (per_tensor) {
    for (const auto& p : m->params) {
      const float wd = (p.kind == K_ENC_REL || p.kind == K_DEC_REL) ? 0.0f : weight_decay;
      RPR_HIP(launch_adamw(p.ptr, flat_grads + p.offset, exp_avg + p.offset, exp_avg_sq + p.offset, p.numel, P<float>(w.gn_out), lr, beta1,
                           beta2, eps, wd, bc1, bc2s, s));
    }
  } else {
    if (w.aw_model != m) {   // tensor table: pointer, offset in the flat buffers, elements, decays?
      std::vector<AdamSeg> segs;
      std::vector<int> pref;
      int chunks = 0;
      for (const auto& p : m->params) {
        if (!p.numel) continue;
        segs.push_back(AdamSeg{p.ptr, (unsigned long long)p.offset, (unsigned long long)p.numel,
                               (p.kind == K_ENC_REL || p.kind == K_DEC_REL) ? 0 : 1});
        pref.push_back(chunks);
        chunks += (int)((p.numel + 4095) / 4096);
      }
      e = tensure(c, w.aseg, segs.size() * sizeof(AdamSeg));
      if (!e) e = tensure(c, w.apref, pref.size() * sizeof(int));
      if (e) return e;
      RPR_HIP(hipMemcpyAsync(w.aseg.p, segs.data(), segs.size() * sizeof(AdamSeg), hipMemcpyHostToDevice, s));
      RPR_HIP(hipMemcpyAsync(w.apref.p, pref.data(), pref.size() * sizeof(int), hipMemcpyHostToDevice, s));
      RPR_HIP(hipStreamSynchronize(s));              // the host vectors go out of scope
      w.aw_model = m; w.aw_nseg = (int)segs.size(); w.aw_chunks = chunks;
    }
    RPR_HIP(launch_adamw_multi(P<AdamSeg>(w.aseg), P<int>(w.apref), w.aw_nseg, w.aw_chunks, flat_grads, exp_avg, exp_avg_sq,
                               P<float>(w.gn_out), lr, beta1, beta2, eps, weight_decay, bc1, bc2s, s));
  }
  if (out_grad_norm) RPR_HIP(hipMemcpyAsync(out_grad_norm, w.gn_out.p, 4, hipMemcpyDeviceToDevice, s));
  // the search / inference paths read the f16 planes of the weights: they are stale now and are re-split by the next call
  // that needs them (ensure_weight_planes) — a training loop never does, and the refresh costs a pass over every weight
  // plus a stream synchronisation per step
  m->planes_dirty = true;
  return RPR_OK;
}

}  // extern "C"
