// Internal declarations shared by the translation units behind the C ABI (api.hip: search path, train_api.hip: the
// training step). Not part of the ABI: include/ripor_hip.h is.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

#include "common.h"
#include "trie.h"

namespace rpr {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace rpr

using rpr::DevBuf;

struct rpr_model {
  rpr_ctx* ctx;
  rpr_model_desc d;
  std::vector<const float*> enc_ln0, enc_qkv, enc_o, enc_ln1, enc_wi, enc_wo;
  std::vector<const float*> dec_ln0, dec_qkv, dec_o, dec_ln1, dec_xq, dec_xo, dec_ln2, dec_wi, dec_wo;
  int32_t* enc_bucket = nullptr;  // [2*MAX_LQ-1]
  int32_t* dec_bucket = nullptr;  // [MAX_DEC_LEN]
  // f16 hi/lo planes of every GEMM weight (split once at load; [2][N][K], plane stride N*K)
  std::vector<__half*> h_enc_qkv, h_enc_o, h_enc_wi, h_enc_wo;
  std::vector<__half*> h_dec_qkv, h_dec_o, h_dec_xq, h_dec_xo, h_dec_wi, h_dec_wo;
  __half* h_dec_xkv = nullptr;
  __half* h_out_embeds = nullptr;  // [2][L*Vp][d]: every codebook padded to Vp = V rounded up to 64 rows (zero rows)
  int Vp() const { return (d.V + 63) & ~63; }   // width of the selection kernel's token axis and of a logits row
  // Fused RMSNorm: the planes of every projection that consumes a normalised input hold W * diag(ln_weight)
  // (enc_qkv: ln0, enc_wi: ln1, dec_qkv: ln0, dec_xq: ln1, dec_wi: ln2, out_embeds: final ln * scaleup factor);
  // the fp32 weights of the caller stay untouched and serve the exact-fp32 mode.
  bool f32_only = false;           // a weight does not fit the f16 planes: every search of this model runs exact fp32
  bool planes_dirty = false;       // the fp32 weights changed (rpr_adamw_step) since the planes were split
  std::vector<void*> owned;
  // how every plane buffer was produced (replayed by refresh_weight_planes after an optimizer step changed the weights)
  struct PlaneJob { const float* w; size_t n; __half* dst; const float* ln; float pre; size_t plane_stride; };   // plane_stride 0 = n
  std::vector<PlaneJob> plane_jobs;
  // trainable tensors in the order of the flat gradient / optimizer-state buffers (train_api.hip)
  struct ParamRef { int kind; int layer; float* ptr; size_t numel; size_t offset; };
  std::vector<ParamRef> params;
  size_t params_total = 0;
  // bound of |logit| for any decoder state: sqrt(d_model) * max row norm of the output codebooks times the final
  // layer-norm weight (Cauchy-Schwarz on the RMS-normalised hidden state), times the scaleup factor; computed at load.
  // The forced-tail fork uses it to prove that no masked (-1e9) candidate can overtake a valid one (api.hip).
  float logit_bound = INFINITY;
  int inner() const { return d.num_heads * d.d_kv; }
  ~rpr_model() {   // device memory goes with the object, also on the error paths of rpr_load_model
    if (enc_bucket) (void)hipFree(enc_bucket);
    if (dec_bucket) (void)hipFree(dec_bucket);
    for (void* p : owned) (void)hipFree(p);
  }
};

struct rpr_trie {
  rpr_ctx* ctx;
  int64_t N;
  int L, V;
  uint16_t* codes = nullptr;  // [dev] sorted [N, L]
  // [dev] child arrays of the first two trie levels (round 5): lvl0[c] = first sorted row whose code 0 is >= c (V + 1 entries,
  // lvl0[V] = N), lvl1[c0 * V + c1] = first row >= (c0, c1) (V * V + 1 entries; only for V <= 1024, L >= 2). The selection
  // kernel reads a child's row range from them at steps 0 and 1 instead of walking 23 / 16 dependent probes of a binary
  // search over the code matrix. Built for the trie's V at upload; dropped by rpr_trie_set_vocab.
  int32_t* lvl0 = nullptr;
  int32_t* lvl1 = nullptr;
  int lvl_V = 0;
  // [dev] round 6: CSR child arrays of the deeper levels (trie.h ChildLevels::deep: children of every node of more than
  // TRIE_NARROW rows) and the level-2 entry of every (c0, c1) node; read by the radix selection (select_radix.hip)
  int32_t* idx2 = nullptr;
  int32_t* d_start[rpr::TRIE_MAX_DEEP] = {};
  uint16_t* d_tok[rpr::TRIE_MAX_DEEP] = {};
  int d_n[rpr::TRIE_MAX_DEEP] = {};
  int n_deep = 0;
  void free_levels() {
    if (lvl0) (void)hipFree(lvl0);
    if (lvl1) (void)hipFree(lvl1);
    if (idx2) (void)hipFree(idx2);
    for (int i = 0; i < rpr::TRIE_MAX_DEEP; ++i) {
      if (d_start[i]) (void)hipFree(d_start[i]);
      if (d_tok[i]) (void)hipFree(d_tok[i]);
      d_start[i] = nullptr; d_tok[i] = nullptr; d_n[i] = 0;
    }
    lvl0 = lvl1 = idx2 = nullptr; n_deep = 0; lvl_V = 0;
  }
  std::vector<int64_t> perm;
  std::vector<uint16_t> host_sorted;
  std::string keys;           // docid strings in original row order, '\n'-joined (only when loaded from a file that has them)
  std::map<int, std::vector<double>> single_frac;   // per search length L: trie_single_frac (lazily, first search of that length)
  ~rpr_trie() { if (codes) (void)hipFree(codes); free_levels(); }
};

struct rpr_d2s {
  std::vector<uint16_t> codes;
  std::string keys;
  int64_t N = 0;
  int L = 0;
};

struct GraphKey {
  const rpr_model* m; const rpr_trie* t; int Q, Lq, B, L; unsigned flags; int lane;   // lane: -1 = the ctx workspace
  int forks;                                                                           // fork depths, 8 bits each (0 = none)
  bool operator<(const GraphKey& o) const {
    return std::tie(m, t, Q, Lq, B, L, flags, lane, forks) < std::tie(o.m, o.t, o.Q, o.Lq, o.B, o.L, o.flags, o.lane, o.forks);
  }
};

// Forced-tail search (api.hip::enqueue_search): a compacted batch of queries that goes on step by step after a fork
struct StageBufs {
  DevBuf qmap;              // int32 [cap]: stage query -> query of the call
  DevBuf cnt;               // int32 [4]: live queries, live rows (queries x beams)
  DevBuf src;               // int32 [cap]: source query (in the previous stage) of every query
  DevBuf offs, last, mask;  // first encoder row / attended length / mask row of every stage query
  DevBuf kcache, vcache;    // [nd][cap][H][depth][B][64]
  DevBuf score[2], lo[2], hi[2], tokens[2], anc[2];
};
// ... and the queries that leave at that fork: their remaining positions are scored in one teacher-forced pass
struct TailBufs {
  DevBuf flag, flist;       // int32 [cap]: forced?, tail query -> stage query
  DevBuf cnt;               // int32 [4]: forced queries, sequences (x B), rows (x B x (L - T))
  DevBuf qmap, offs, last, mask;
  DevBuf tokens;            // uint16 [cap * B][L]
  DevBuf gold;              // float [cap * B][L - T]
};
constexpr int MAX_FORKS = 2;

struct Workspace {
  // encoder
  DevBuf ids, mask, last, offs, row_src, ex, eh, eqkv, eattn, eff, enc_out, xkv;
  // decoder
  DevBuf x, h, q, attn, ff, logits, kcache, vcache, lb;
  DevBuf sel_rs;          // radix selection (many beams): RadixWs scratch (select_radix_carve)
  // beam state (2 ping-pong buffers)
  DevBuf score[2], lo[2], hi[2], tokens[2], anc[2];
  // staged outputs
  DevBuf o_tokens, o_scores, o_lo, o_hi;
  // f16 hi/lo planes of the GEMM inputs (split-precision mode): attention outputs, FF intermediates, the final
  // encoder states, and the UN-normalised residual streams (fused RMSNorm) with their row sums of squares
  DevBuf eattn_h, eff_h, enc_out_h, attn_h, ff_h, ex_h, x_h, ssq_e, ssq_d;
  DevBuf tr_x, tr_misc;   // rpr_train_forward scratch (teacher-forced decoder)
  DevBuf part;            // split-K partial sums of the mid-size GEMM route (gemm_h2.hip): 9 M floats
  // forced-tail search: stages 1.. (stage 0 = the buffers above), one tail job per fork, and the activations of a tail
  // pass (rows = queries x beams x remaining positions), shared by the forks
  StageBufs stage[MAX_FORKS];
  TailBufs tail[MAX_FORKS];
  DevBuf t_x, t_h, t_qkv, t_q, t_attn, t_ff, t_x_h, t_attn_h, t_ff_h, t_ssq;
  DevBuf t_logits;        // log-softmax mode: the V logits of every tail row (fp32)
};

static_assert(sizeof(Workspace) % sizeof(DevBuf) == 0 && std::is_standard_layout<Workspace>::value,
              "Workspace must consist of DevBuf members only (rpr_free_ctx walks it as an array)");

// Half of a large search batch: its own workspace (KV cache, graphs are keyed by the lane) and a HIP stream confined to
// half of the CUs (hipExtStreamCreateWithCUMask). The two halves of a batch run side by side: the HBM-bound attention of
// one under the power-bound GEMMs of the other — on one stream every kernel has all CUs in the same phase.
struct Lane {
  Workspace ws;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
};

struct rpr_ctx {
  int device;
  int precision = RPR_PREC_F16X2;
  unsigned int* status = nullptr;       // [dev, 64 words] [8] weight-range probe; sticky words: [0] a value left the f16 plane range, [1] a query attends to nothing, [2] a query was left unforced by the last fork of an optimistic forced-tail search
  unsigned int* status_host = nullptr;  // pinned mirror filled by rpr_get_status
  struct TrainWs* tws = nullptr;        // activations / scratch of the training step (train_api.hip), freed by free_train_ws
  unsigned long long* trace_buf = nullptr;  // diagnostic (RPR_GEMM_TRACE): cycle stamps of block 0 of the last f16x2 GEMM
  Workspace ws;
  Lane lanes[2];
  int lanes_state = 0;          // 0 not tried yet, 1 ready, -1 masked streams unavailable on this device
  int lane_min_rows = 10240;    // batches of at least this many decoder rows (queries x beams) are split over the two lanes (0 = never)
  hipEvent_t fork_ev = nullptr;
  int cur_cus = 0;              // CUs of the lane the current enqueue runs on (0 = the whole chip)
  int lane_cus = 0;             // CUs per lane
  int cur_lane = -1;            // lane of the current enqueue (-1 = the ctx stream)
  int cur_no_row_split = 0;     // 1 while the packed encoder (and the cross-K/V product on its rows) is enqueued: GemmH2Args.no_row_split
  int cur_small_live = 0;       // > 0 while a leftover stage is enqueued: its GEMMs are paired (GemmH2Args.small_live)
  int forced_tail = 1;          // 0 = every query runs all L steps sequentially, 1 = exact forced tail, 2 = optimistic (see choose_forks)
  int fork_override[MAX_FORKS] = {0, 0};   // explicit fork depths (rpr_set_fork_depths / RPR_FORK_DEPTHS); 0 = from the trie statistics
  int n_fork_override = -1;     // -1 = automatic
  std::vector<int> last_forks;  // fork depths of the last rpr_search and the workspaces it ran in (bit 0: ctx, 1 / 2: lanes)
  int last_ws_mask = 0;         //   -> rpr_last_fork_stats
  size_t ws_bytes = 0;
  int enc_rows_accounted = 0;   // live encoder rows of the last enqueue (profile accounting)
  hipStream_t cap_stream = nullptr;
  std::map<GraphKey, hipGraphExec_t> graphs;
  // profiling
  bool profiling = false;
  // one timed launch of the eager profile pass. live_dev (nullable): device word holding the live rows of a compacted /
  // packed launch whose flops and bytes were accounted for live_static rows at enqueue time; read back when the
  // records are flushed — no synchronisation while a step is being enqueued, so both lanes run side by side exactly as
  // in the timed region
  struct Rec { int cls; hipEvent_t a, b; double flops, bytes; const int* live_dev; int live_static; };
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  rpr_kernel_stats done[RPR_K_COUNT];
};

namespace rpr {

// re-split every GEMM weight into its f16 planes (api.hip); sets model->f32_only if a weight no longer fits
int refresh_weight_planes(rpr_ctx* c, rpr_model* m, hipStream_t s);
inline int ensure_weight_planes(rpr_ctx* c, rpr_model* m, hipStream_t s) {
  if (!m->planes_dirty) return 0;
  const int e = refresh_weight_planes(c, m, s);
  if (!e) m->planes_dirty = false;
  return e;
}
void free_train_ws(rpr_ctx* c);   // train_api.hip
void train_forget_model(rpr_ctx* c, const rpr_model* m);   // train_api.hip: drop the per-model weight cache table

inline int ensure(rpr_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return 0;
  if (b.p) {
    RPR_HIP(hipFree(b.p));
    c->ws_bytes -= b.cap;
    b.p = nullptr; b.cap = 0;
    // graphs captured against the old pointers are stale
    for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.second);
    c->graphs.clear();
  }
  const size_t want = (bytes + 255) & ~(size_t)255;
  RPR_HIP(hipMalloc(&b.p, want));
  b.cap = want;
  c->ws_bytes += want;
  return 0;
}

template <class T> T* P(const DevBuf& b) { return reinterpret_cast<T*>(b.p); }

// scoped temporary device buffer (test hooks): freed on every return path
struct DevTmp {
  void* p = nullptr;
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
  ~DevTmp() { if (p) (void)hipFree(p); }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Launch wrapper: optional hipEvent timing per kernel class (bench.py roofline leg).
struct Launcher {
  rpr_ctx* c;
  hipStream_t s;
  int err = 0;
  const int* live_dev = nullptr;   // profile accounting of the launches that follow (see rpr_ctx::Rec)
  int live_static = 0;
  void account_live(const int* dev, int rows_static) { live_dev = dev; live_static = rows_static; }
  hipEvent_t get_event() {
    if (!c->pool.empty()) { hipEvent_t e = c->pool.back(); c->pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
  }
  template <class F> void run(int cls, double flops, double bytes, F&& f, const int* cls_after = nullptr) {
    if (err) return;
    hipEvent_t ea = nullptr, eb = nullptr;
    if (c->profiling) {
      ea = get_event(); eb = get_event();
      if (ea) (void)hipEventRecord(ea, s);
    }
    hipError_t e = f();
    if (e != hipSuccess) { err = hip_fail(e, "kernel launch", __FILE__, __LINE__); return; }
    if (c->profiling && ea && eb) {
      (void)hipEventRecord(eb, s);
      c->recs.push_back({cls_after ? *cls_after : cls, ea, eb, flops, bytes, live_dev, live_static});
    }
  }
};

}  // namespace rpr
