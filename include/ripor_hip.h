/*
 * ripor_hip.h — C ABI of libripor_hip.so: the MI355X (gfx950) trie-constrained beam-search
 * generative-retrieval hot path.
 *
 * The reference (HansiZeng/RIPOR) has no FFI layer: its boundary for this path is the Python call
 *   generate_for_constrained_prefix_beam_search(model, prefix_constrain_processor, input_ids,
 *       attention_mask, max_new_tokens=L, num_beams=B, num_return_sequences=B, ...)
 *   (reference t5_pretrainer/tasks/generation.py:35-78; callers t5_pretrainer/evaluate.py:60,102,149)
 * plus the constructors T5SeqAQEncoder.from_pretrained (modeling/t5_generative_retriever.py:772-855)
 * and PrefixConstrainLogitProcessorFastSparse (tasks/generation.py:603-642).
 * Each entry point below names the reference interface it replaces. The Python mirror of those
 * interfaces (ripor_amd/tasks/generation.py, ripor_amd/modeling/t5_generative_retriever.py)
 * binds these symbols with ctypes (ripor_amd/_lib.py); INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions: every function returns 0 on success or a negative rpr_status; nothing throws
 * across the ABI; rpr_last_error() returns a human-readable message for the last failure on the
 * calling thread. All tensor arguments are plain pointers + sizes. Pointers marked [dev] are HIP
 * device pointers (e.g. torch tensor .data_ptr()), [host] are host pointers. Caller owns every
 * in/out buffer; the library owns its workspaces, KV cache and hipGraphs. One rpr_ctx per device,
 * not thread-safe per ctx. `stream` is a hipStream_t passed as void* (NULL = default stream).
 */
#ifndef RIPOR_HIP_H
#define RIPOR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  RPR_OK = 0,
  RPR_ERR_INVALID = -1,  /* bad argument / unsupported configuration */
  RPR_ERR_HIP = -2,      /* a HIP runtime call failed */
  RPR_ERR_NO_DEVICE = -3,
  RPR_ERR_OOM = -4
} rpr_status;

typedef struct rpr_ctx rpr_ctx;
typedef struct rpr_model rpr_model;
typedef struct rpr_trie rpr_trie;

/* Weights of a T5ForDocIDGeneration checkpoint (reference modeling/t5_generative_retriever.py:84-112;
 * SURVEY.md §8 row a14). All matrices are float32, row-major [out_features, in_features] exactly
 * as torch.nn.Linear stores them; q/k/v of one attention are concatenated along out_features by
 * the host (rows 0..inner-1 = q, then k, then v). Per-layer pointers are host arrays of device
 * pointers with num_layers / num_decoder_layers entries. */
typedef struct {
  int32_t vocab_size, d_model, d_kv, d_ff, num_heads;   /* d_kv: 64 (t5-base / t5-large) or 128 (t5-3b: rpr_search / rpr_encode only,
                                                         * no training entry points) */
  int32_t num_layers, num_decoder_layers;
  int32_t rel_buckets, rel_max_distance;
  int32_t L;                     /* len(config.decoder_vocab_sizes) = max decoder positions      */
  int32_t V;                     /* decoder_vocab_sizes[i], must be equal for all i (evaluate.py:433); 2..65536 —
                                  * sizes off the 64 grid are padded internally, only rpr_debug_taps need V % 64 == 0 */
  int32_t scaleup_output_hidden; /* config.scaleup_output_hidden (t5_generative_retriever.py:427)  */
  float layer_norm_eps;
  const float* shared;           /* [dev] shared.weight [vocab_size, d_model]                     */
  const float* enc_rel_bias;     /* [dev] encoder.block.0...relative_attention_bias [buckets, H]   */
  const float* dec_rel_bias;     /* [dev] decoder.block.0...relative_attention_bias [buckets, H]   */
  const float* enc_final_ln;     /* [dev] [d_model] */
  const float* dec_final_ln;     /* [dev] [d_model] */
  const float* start_embed;      /* [dev] start_token_embed [d_model]                             */
  const float* in_embeds;        /* [dev] list_decoder_embeds stacked [L, V, d_model]              */
  const float* out_embeds;       /* [dev] list_output_embeds stacked [L, V, d_model] (== in_embeds if shared) */
  const float* dec_xkv;          /* [dev] all decoder layers' EncDecAttention k,v stacked
                                    [num_decoder_layers * 2 * inner, d_model]: layer i rows
                                    [2i*inner, (2i+1)*inner) = k, next inner rows = v               */
  const float* const* enc_ln0;   /* [host][num_layers] -> [dev] layer.0.layer_norm [d_model]       */
  const float* const* enc_qkv;   /*                    -> [dev] [3*inner, d_model]                 */
  const float* const* enc_o;     /*                    -> [dev] [d_model, inner]                   */
  const float* const* enc_ln1;
  const float* const* enc_wi;    /* [d_ff, d_model] */
  const float* const* enc_wo;    /* [d_model, d_ff] */
  const float* const* dec_ln0;   /* [host][num_decoder_layers] */
  const float* const* dec_qkv;   /* SelfAttention [3*inner, d_model] */
  const float* const* dec_o;
  const float* const* dec_ln1;
  const float* const* dec_xq;    /* EncDecAttention.q [inner, d_model] */
  const float* const* dec_xo;    /* EncDecAttention.o [d_model, inner] */
  const float* const* dec_ln2;
  const float* const* dec_wi;
  const float* const* dec_wo;
} rpr_model_desc;

/* Arithmetic of the projection GEMMs (everything else — attention, norms, scores — is fp32/fp64):
 *  RPR_PREC_F32    exact fp32 MFMA (v_mfma_f32_32x32x2_f32), the numerical reference;
 *  RPR_PREC_F16X2  every fp32 operand carried as two f16 planes (hi + lo, 22 significant bits), products
 *                  evaluated as hi*hi + hi*lo + lo*hi with three f16 MFMAs accumulating in fp32:
 *                  fp32-equivalent to ~2^-22 at 5.3x the fp32-MFMA rate (default). */
#define RPR_PREC_F32 0
#define RPR_PREC_F16X2 1
/*  RPR_PREC_BF16   (training step only: rpr_lngknp_backward) every GEMM operand rounded to bf16, one bf16 MFMA per
 *                  product, fp32 accumulation and fp32 results — the arithmetic of the reference's bf16 autocast
 *                  (main.py:152 bf16=args.use_fp16, full_lng_knp_train_pipline.sh:93; tasks/trainer.py:229), BASELINE
 *                  config 5 as stated. The search / inference entry points need fp32-equivalent scores (1e-4) and run as
 *                  RPR_PREC_F16X2 under this setting. */
#define RPR_PREC_BF16 2

/* rpr_search flags */
#define RPR_FLAG_LOG_SOFTMAX 1u /* apply_log_softmax_for_scores (generation.py:453-455)           */
#define RPR_FLAG_NO_GRAPH 2u    /* launch kernels eagerly instead of replaying a hipGraph           */

/* Optional per-step taps for parity tests (all [dev], any may be NULL). */
typedef struct {
  float* encoder_out;   /* [Q, Lq, d_model] final encoder hidden states                            */
  float* step_logits;   /* [L, Q*B, V] logits of position t for the beams alive at step t           */
  double* step_scores;  /* [L, Q, B] cumulative float64 beam scores after step t (slot order)        */
  int32_t* step_tokens; /* [L, Q, B] token chosen for each new slot at step t                       */
  int32_t* step_parent; /* [L, Q, B] parent slot of each new slot at step t                         */
  uint64_t* step_valid; /* [L, Q, B*V/64] the trie child bitmap the selection of step t works from: bit
                           (beam*V + token) of query q is set iff token is a child of the beam's trie node —
                           the mask of PrefixConstrainLogitProcessorFastSparse.__call__ (generation.py:666-677)
                           for the beams alive at step t, in slot order                                  */
} rpr_debug_taps;

/* Timing of one kernel class, accumulated with hipEvents on the launch stream while
 * profiling is enabled (bench.py's roofline leg). */
typedef struct {
  double total_ms;
  int64_t launches;
  double flops;  /* algorithmic flops of those launches */
  double bytes;  /* algorithmic bytes of those launches */
} rpr_kernel_stats;

/* RPR_K_GEMM = the dominant projection kernel (256x256 ping-pong tiles in f16x2 mode, the fp32 MFMA kernel in exact
 * mode); RPR_K_GEMM_SMALL = the launches that fall to the 128-row / skinny tile kernels (few rows or few tiles). */
enum { RPR_K_GEMM = 0, RPR_K_DEC_SELF_ATTN = 1, RPR_K_DEC_CROSS_ATTN = 2, RPR_K_ENC_ATTN = 3,
       RPR_K_RMSNORM = 4, RPR_K_SELECT = 5, RPR_K_OTHER = 6, RPR_K_GEMM_SMALL = 7,
       RPR_K_TAIL_SELF_ATTN = 8, /* causal block attention of the forced-tail pass */
       RPR_K_FORK = 9,           /* fork classification / compaction / rank replay of the forced-tail search */
       RPR_K_COUNT = 10 };

/* ---- lifecycle (replaces: model.to(local_rank), evaluate.py:470; ddp_setup device binding) ---- */
int rpr_init(int device, rpr_ctx** out_ctx);
void rpr_free_ctx(rpr_ctx* ctx);
const char* rpr_last_error(void);
/* Library/ABI version; bumped when a signature changes. */
int rpr_abi_version(void);
/* Select the GEMM arithmetic (RPR_PREC_*); default RPR_PREC_F16X2, or RPR_PRECISION=f32|f16x2 in the
 * environment at rpr_init. Takes effect on the next rpr_search/rpr_encode/rpr_op_linear. */
int rpr_set_precision(rpr_ctx* ctx, int precision);
int rpr_get_precision(const rpr_ctx* ctx);

/* Host-only: HF T5Attention._relative_position_bucket evaluated the way the library fills its
 * device lookup tables (float32 log, truncation); rel = key_pos - query_pos. Needs no GPU. */
int rpr_rel_bucket(int rel, int bidirectional, int num_buckets, int max_distance);

/* ---- model (replaces T5SeqAQEncoder.from_pretrained(...).base_model, t5_generative_retriever.py:772-784) ---- */
int rpr_load_model(rpr_ctx* ctx, const rpr_model_desc* desc, rpr_model** out_model);
void rpr_free_model(rpr_model* model);

/* ---- trie (replaces list_smtid_to_nextids + PrefixConstrainLogitProcessorFastSparse(list, V),
 *      evaluate.py:404-434, generation.py:603-642, and smtid_to_docids, evaluate.py:439-446) ----
 * codes: [host] [N, L] row-major, codes[i*L + l] in [0, V); row i is the smtid of docid index i.
 * The library sorts the rows lexicographically (stable), keeps the sorted matrix on the device and
 * the permutation on the host. A beam's trie node is the half-open range [lo, hi) of sorted rows
 * sharing its prefix; the docids under a final smtid are perm[lo..hi). */
int rpr_build_trie(rpr_ctx* ctx, const uint16_t* codes, int64_t N, int32_t L, int32_t V, rpr_trie** out_trie);
void rpr_free_trie(rpr_trie* trie);
int64_t rpr_trie_num_rows(const rpr_trie* trie);
/* [host] perm[N]: perm[sorted_row] = original row index (docid index). Valid until rpr_free_trie. */
const int64_t* rpr_trie_perm(const rpr_trie* trie);
/* Binary trie cache replacing list_smtid_to_nextids.pkl (evaluate.py:404-408,428-432): sorted code matrix,
 * permutation and (optionally) the docid strings, so that a run needs neither the JSON parse nor the sort.
 * rpr_trie_build_file is HOST ONLY (no ctx, no GPU): the replacement of
 * `python -m t5_pretrainer.aq_preprocess.build_list_smtid_to_nextids` (aq_preprocess/build_list_smtid_to_nextids.py:13-41).
 *   keys: [host] docid strings in row order joined by '\n' (key_bytes = 0: none); src_size / src_mtime_ns: identity of
 *   the docid_to_smtid.json the codes came from (0 = unknown), returned by rpr_trie_file_info so callers can detect a
 *   stale cache. rpr_trie_load validates everything it reads (sizes vs file length, codes < V, sorted rows, perm). */
int rpr_trie_save(const rpr_trie* trie, const char* path);
int rpr_trie_build_file(const uint16_t* codes, int64_t N, int32_t L, int32_t V, const char* keys, int64_t key_bytes,
                        int64_t src_size, int64_t src_mtime_ns, const char* path);
int rpr_trie_file_info(const char* path, int64_t* N, int32_t* L, int32_t* V, int64_t* key_bytes, int64_t* src_size,
                       int64_t* src_mtime_ns);
int rpr_trie_load(rpr_ctx* ctx, const char* path, rpr_trie** out_trie);
/* HOST ONLY: the full validation rpr_trie_load performs, without a device (0 = the file is well-formed). */
int rpr_trie_file_validate(const char* path);
/* Dimensions of a trie object (the FILE's L and V after rpr_trie_load); key_bytes = size of its docid blob. */
int rpr_trie_dims(const rpr_trie* trie, int64_t* N, int32_t* L, int32_t* V, int64_t* key_bytes);
/* [host] out: key_bytes chars, the docid strings in original row order joined by '\n'. */
int rpr_trie_keys(const rpr_trie* trie, char* out);
/* A cache is built without knowing the model: widen V to the model's decoder vocab size (must exceed every code). */
int rpr_trie_set_vocab(rpr_trie* trie, int32_t V);
/* Host-side child mask of arbitrary prefixes — the processor's __call__ (generation.py:666-677)
 * without a model, for callers that use the processor object on its own. prefix: [host] [R, T] with
 * column 0 ignored (start id); out_mask: [host] [R, V] bytes 0/1. Runs a stand-alone device kernel
 * (prefix_mask_kernel: walks the prefix by binary search, then one search per token); the mask the SEARCH
 * uses is built inside select_kernel and is exposed through rpr_debug_taps.step_valid. */
int rpr_trie_mask(rpr_ctx* ctx, const rpr_trie* trie, const int32_t* prefix, int32_t R, int32_t T,
                  uint8_t* out_mask);

/* ---- docid_to_smtid.json reader (host only, no ctx) ------------------------------------------------
 * Replaces `ujson.load(docid_to_smtid_path)` + the list comprehension over 8.8 M Python lists in
 * evaluate.py:400-402,439-446 and aq_preprocess/build_list_smtid_to_nextids.py:21-27: one streaming pass
 * over {"docid": [-1, c1, ..., cL], ...} (format: aq_preprocess/create_customized_smtid_file.py:47-59)
 * into a [N, L] uint16 code matrix in file order plus the docid strings. */
typedef struct rpr_d2s rpr_d2s;
int rpr_d2s_open(const char* path, rpr_d2s** out);
/* N docs, L codes per doc (the leading -1 dropped), key_bytes = length of the '\n'-joined docid strings */
int rpr_d2s_dims(const rpr_d2s* h, int64_t* N, int32_t* L, int64_t* key_bytes);
/* codes: [host] [N, L] uint16; keys: [host] key_bytes chars ('\n'-separated, file order). Either may be NULL. */
int rpr_d2s_copy(const rpr_d2s* h, uint16_t* codes, char* keys);
void rpr_d2s_close(rpr_d2s* h);

/* ---- the hot path (replaces generate_for_constrained_prefix_beam_search, generation.py:35-251,
 *      -> beam_search_for_constrained_prefix :253-575 incl. BeamSearchScorer.process/finalize) ----
 * input_ids, attention_mask: [dev] int32 [Q, Lq] (pad id 0 / mask 0 on padding, any Lq <= 256).
 * out_tokens: [dev] int32 [Q, B, L] generated smtid tokens, beams ranked best-first per query
 *             (= outputs.sequences[:, 1:] of the reference, which prepends the start id 0).
 * out_scores: [dev] float32 [Q, B] = float32(sum of step scores (float64) / (L+1))
 *             (= outputs.sequences_scores).
 * out_row_lo/out_row_hi: [dev] int64 [Q, B] sorted-row range of each returned smtid (empty range:
 *             the smtid is not in the trie; the reference prints "smtid not in smtid_to_docid").
 * L may be smaller than the model's decoder length (prefix search, evaluate.py:134-178). */
int rpr_search(rpr_ctx* ctx, rpr_model* model, rpr_trie* trie, const int32_t* input_ids,
               const int32_t* attention_mask, int32_t Q, int32_t Lq, int32_t B, int32_t L, uint32_t flags,
               int32_t* out_tokens, float* out_scores, int64_t* out_row_lo, int64_t* out_row_hi,
               const rpr_debug_taps* taps, void* stream);

/* Lane split of large batches. A call of rpr_search with at least `min_rows` decoder rows (queries x beams; default
 * 10240; 0 = never; env RPR_LANE_MIN_ROWS) runs as two halves on two internal HIP streams, each confined to half of the CUs
 * (hipExtStreamCreateWithCUMask) and each with its own workspace: the HBM-bound attention of one half overlaps the
 * power-bound GEMMs of the other (+3.6 % queries/s at 2150 queries in flight; on one stream all CUs are in the same
 * phase). Results are identical to the unsplit call; the caller's stream waits for both halves. Calls with debug taps
 * are never split. rpr_lane_split returns the threshold in force, 0 when splitting is off or masked streams are
 * unavailable on the device. No counterpart in the reference (its loop is host-bound at batch 1). */
int rpr_set_lane_split(rpr_ctx* ctx, int32_t min_rows);
int32_t rpr_lane_split(rpr_ctx* ctx);

/* Forced-tail evaluation (default on; env RPR_FORCED_TAIL=0 / RPR_FORK_DEPTHS="4,6"). Beam search over a docid trie
 * stops deciding early: once every beam of a query stands on a trie node under which a single distinct smtid remains
 * (8.8 M docs under 256^4 depth-4 prefixes: after 4 steps for 99 % of the queries), each beam has exactly one valid
 * child per step, nothing can be pruned any more and the remaining tokens are the rest of the beam's code row — only
 * the scores are missing. rpr_search therefore walks the first steps sequentially, FORKS at up to two depths chosen from
 * the trie (queries that are forced there leave; the others are compacted and walk on), and scores the remaining
 * positions of the forced queries in one teacher-forced decoder pass per fork instead of L - T KV-gathering steps.
 * Results are those of the step-by-step loop (generation.py:423-540: float64 cumulative scores, the slot order of every
 * step replayed, sum/(L+1) finalize with ties in reverse slot order); a fork only takes a query whose beams are all
 * live, single-sequence and close enough in score that no masked (-1e9) candidate could be selected (bound on |logit|
 * from the output codebooks, computed at rpr_load_model). Calls with debug taps never fork; RPR_FLAG_LOG_SOFTMAX
 * forks too (the tail pass then computes the V logits of every forced position for the log-sum-exp).
 *   rpr_set_forced_tail(ctx, mode)           0 = off; 1 = exact (default): whoever is still unforced after the last fork
 *                                            walks on to L on the device; 2 = optimistic: when the trie statistics promise
 *                                            an (almost always) empty last stage, it is not enqueued at all (~100 launches
 *                                            per remaining step for nobody) — a query left unforced by the last fork raises
 *                                            the sticky RPR_STATUS_TAIL_LEFTOVER and the results of that call are NOT valid:
 *                                            repeat it in mode 1 (ripor_amd/tasks/generation.py and bench.py do).
 *   rpr_forced_tail(ctx)                     the mode in force;
 *   rpr_set_fork_depths(ctx, n, depths)      n = -1: depths from the trie statistics (default); n = 0..2: explicit,
 *                                            ascending, each in [1, L-1] (entries >= L are ignored at search time);
 *   rpr_fork_depths(...)                     the depths a search of this shape would use -> out_depths[2]; returns
 *                                            their number (>= 0) or a negative rpr_status. */
int rpr_set_forced_tail(rpr_ctx* ctx, int32_t mode);
int32_t rpr_forced_tail(const rpr_ctx* ctx);
int rpr_set_fork_depths(rpr_ctx* ctx, int32_t n, const int32_t* depths);
int rpr_fork_depths(rpr_ctx* ctx, rpr_model* model, rpr_trie* trie, int32_t Q, int32_t B, int32_t L, uint32_t flags,
                    int32_t* out_depths);
/* HOST ONLY (no ctx, no GPU): the statistic the automatic fork depths come from. codes: [host] [N, Lc] in any order;
 * out_frac: [host] L + 1 doubles, out_frac[t] = share of the trie nodes at depth t (distinct t-prefixes) under which
 * exactly one distinct L-token sequence remains. A query whose B beams stand on random depth-t nodes is forced with
 * probability ~ out_frac[t]^B: the first fork is the first depth where that reaches 1/2, the second the first later
 * depth where fewer than 0.05 queries of the call are expected to stay unforced. */
int rpr_trie_single_frac(const uint16_t* codes, int64_t N, int32_t Lc, int32_t L, double* out_frac);
/* Diagnostic (synchronises the device): for every fork of the last rpr_search its depth, the number of queries that
 * were forced there and the number that walked on; [host] arrays of 2 entries each; returns the number of forks. */
int rpr_last_fork_stats(rpr_ctx* ctx, int32_t* out_depths, int32_t* out_forced, int32_t* out_left);

/* ---- forward of the prefix-oriented ranking fine-tune step (BASELINE config 5) ----------
 * Replaces the forward of T5SeqAQEncoderForLngKnpMarginMSE (modeling/t5_generative_retriever.py:902-966; also
 * T5SeqAQEncoderForMarginMSE :847-882 with n_prefix = 1 and T5SeqAQEncoder.rerank_forward :788-792 with n_prefix = 0):
 * teacher-forced decoder passes of base_model over the smtids of n_docs documents per query
 * (decoder_input_ids = [-1, c_1 .. c_{L-1}], dataset/dataset.py:497-500), decode() of the doc encodings through the
 * OUTPUT codebooks (:812-826), per-position scores <decoder_last_hidden_state[i], E_out[i][c_i]>, and the MSE losses
 * between the student margins over the first prefix_lens[p] positions and the teacher margins.
 *   input_ids, attention_mask: [dev] int32 [bz, Lq]      (the query; the positive and the negative pass share it)
 *   doc_codes:   [dev] int32 [bz, n_docs, L]   doc_codes[b][0] = positive, [b][1] = negative smtid (codes in [0, V))
 *   teacher_pos, teacher_neg: [dev] float [n_prefix, bz]  teacher scores per prefix length (row p belongs to
 *                prefix_lens[p]; the reference's order is rank (= L), rank_4, rank_8, rank_16)
 *   prefix_lens: [dev] int32 [n_prefix]
 *   out_losses:  [dev] float [n_prefix]        mean_b (student_margin - teacher_margin)^2   (torch.nn.MSELoss)
 *   out_position_scores: [dev] float [bz, n_docs, L], nullable
 * Inference-style forward (split-precision GEMMs, nothing kept for a backward pass); the training step is
 * rpr_lngknp_backward + rpr_adamw_step below. Eager launches on `stream`; asynchronous. */
int rpr_lngknp_forward(rpr_ctx* ctx, rpr_model* model, const int32_t* input_ids, const int32_t* attention_mask,
                       int32_t bz, int32_t Lq, const int32_t* doc_codes, int32_t n_docs, int32_t L,
                       const float* teacher_pos, const float* teacher_neg, const int32_t* prefix_lens, int32_t n_prefix,
                       float* out_losses, float* out_position_scores, void* stream);

/* ---- the training step of the same model (config 5): backward pass + optimizer --------------------------------
 * Replaces, for loss_type t5seq_aq_encoder_lng_knp_margin_mse, what HF Trainer does per step around the forward above
 * (tasks/trainer.py:203-275 training_step: loss = sum of the task losses, loss.backward(); Trainer defaults behind
 * main.py:131-155: clip_grad_norm_(1.0), torch.optim.AdamW(lr, betas (0.9, 0.999), eps 1e-8, weight_decay 0)).
 *
 * Trainable tensors and the flat buffers. The model's tensors (the device pointers of rpr_model_desc: fp32, caller
 * owned) are updated IN PLACE by rpr_adamw_step. Gradients and the two AdamW moments live in caller-owned flat fp32
 * buffers of rpr_param_total(model) elements; tensor i occupies [offset_i, offset_i + numel_i) in each of them and
 * rpr_param_info returns its device pointer (so the host can map it back to a checkpoint name), numel and offset.
 * Order: shared, encoder / decoder relative bias, final layer norms, start embedding, stacked input codebooks, stacked
 * output codebooks (absent when shared with the input ones), stacked cross-attention k/v, then per encoder layer
 * ln0, qkv, o, ln1, wi, wo and per decoder layer ln0, qkv, o, ln1, xq, xo, ln2, wi, wo (the concatenated layouts of
 * rpr_model_desc). The flat gradient buffer is what a data-parallel caller all-reduces (RCCL) between the two calls.
 *
 * rpr_lngknp_backward: forward (activations kept in library-owned memory) + backward of the sum of the n_prefix
 *   margin-MSE losses; same batch arguments as rpr_lngknp_forward with n_docs = 2; writes out_losses [dev, n_prefix]
 *   and overwrites flat_grads [dev, rpr_param_total]. fp32 activations and gradients; the matrix products follow the
 *   context's precision (rpr_set_precision): RPR_PREC_F16X2 (default) = split-precision f16 MFMA with per-tensor
 *   dynamic plane scales found on the device, RPR_PREC_F32 = exact-fp32 MFMA. Deterministic reductions in both.
 *   Lq <= 128, L <= the model's decoder length.
 * rpr_adamw_step: global L2 norm of flat_grads (-> out_grad_norm [dev, 1], nullable), clip coefficient
 *   min(1, max_grad_norm / (norm + 1e-6)) (max_grad_norm <= 0: no clipping), AdamW update of every tensor with the
 *   bias corrections of `step` (1-based); the f16 weight planes of the search / inference paths are marked stale and
 *   re-split by the next rpr_search / rpr_lngknp_forward / rpr_encode on this model (a training loop never pays for it). weight_decay applies to every tensor except the two relative-attention-bias tables (HF 4.17
 *   Trainer.create_optimizer excludes nn.LayerNorm parameters and names containing "bias"; T5LayerNorm is not an
 *   nn.LayerNorm there, so layer-norm weights decay). The reference's default is weight_decay = 0.
 *   exp_avg / exp_avg_sq: [dev, rpr_param_total], zero before the first step. */
int64_t rpr_param_count(rpr_model* model);
int64_t rpr_param_total(rpr_model* model);
int rpr_param_info(rpr_model* model, int64_t index, const float** ptr, int64_t* numel, int64_t* offset);
int rpr_lngknp_backward(rpr_ctx* ctx, rpr_model* model, const int32_t* input_ids, const int32_t* attention_mask, int32_t bz,
                        int32_t Lq, const int32_t* doc_codes, int32_t L, const float* teacher_pos, const float* teacher_neg,
                        const int32_t* prefix_lens, int32_t n_prefix, float* out_losses, float* flat_grads, void* stream);
/* The same pass with the gradient exchange overlapped (reference: DistributedDataParallel's bucketed all-reduce running
 * under the backward pass, tasks/trainer.py:486). The flat gradient buffer is handed over in BUCKETS: one per
 * transformer layer (decoder layers last to first, then encoder layers last to first: the order the backward finishes
 * them; a layer's tensors are contiguous in the flat layout) and a final one for everything in front of the first layer
 * (embeddings, codebooks, cross K/V, final norms, bias tables). For each bucket the library makes `comm_stream` wait for
 * the kernels that produce it (main stream and the internal weight-gradient stream) and then calls, on the calling host
 * thread and before returning, on_bucket(user, offset, numel): the callback enqueues the bucket's all-reduce (RCCL) on
 * comm_stream, where it runs beside the remaining backward kernels. The caller joins comm_stream before rpr_adamw_step.
 * on_bucket == NULL: exactly rpr_lngknp_backward. */
typedef void (*rpr_grad_bucket_cb)(void* user, int64_t offset, int64_t numel);
int rpr_lngknp_backward_buckets(rpr_ctx* ctx, rpr_model* model, const int32_t* input_ids, const int32_t* attention_mask,
                                int32_t bz, int32_t Lq, const int32_t* doc_codes, int32_t L, const float* teacher_pos,
                                const float* teacher_neg, const int32_t* prefix_lens, int32_t n_prefix, float* out_losses,
                                float* flat_grads, void* stream, void* comm_stream, rpr_grad_bucket_cb on_bucket, void* user);
int rpr_adamw_step(rpr_ctx* ctx, rpr_model* model, const float* flat_grads, float* exp_avg, float* exp_avg_sq, int64_t step,
                   float lr, float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
                   float* out_grad_norm, void* stream);

/* ---- sticky status of a ctx ------------------------------------------------------------------------
 * RPR_STATUS_SATURATED: in RPR_PREC_F16X2 mode an activation left the range of the f16 planes (|x| > 65504 after
 *   the plane scale: 4094 for normalised activations and attention outputs, 65504 for the residual stream, 1.05e6
 *   for the FF intermediate) and was clamped — the results of the searches since the last clear are NOT trustworthy;
 *   repeat them with rpr_set_precision(RPR_PREC_F32) (ripor_amd/tasks/generation.py does so automatically).
 *   Weights are checked at rpr_load_model: a model that does not fit is pinned to RPR_PREC_F32 (rpr_model_f32_only).
 * RPR_STATUS_EMPTY_QUERY: a query's attention mask has no attended position (its cross-attention output is defined
 *   as zero here; the reference would average the padded positions).
 * rpr_get_status synchronises `stream`, returns the flags accumulated by the work enqueued so far and, if clear != 0,
 * resets them. */
#define RPR_STATUS_SATURATED 1u
#define RPR_STATUS_EMPTY_QUERY 2u
/* RPR_STATUS_TAIL_LEFTOVER: a search in the optimistic forced-tail mode (rpr_set_forced_tail(ctx, 2)) met a query that
 *   was still unforced at its last fork; the outputs of that query are unspecified — repeat the call in mode 1. */
#define RPR_STATUS_TAIL_LEFTOVER 4u
int rpr_get_status(rpr_ctx* ctx, void* stream, uint32_t* out_flags, int clear);
/* The same words WITHOUT a synchronisation: enqueues on `stream` a copy of the four raw status words ([0] != 0:
 * SATURATED, [1] != 0: EMPTY_QUERY, [2] != 0: TAIL_LEFTOVER, [3] reserved) into host_words ([host], pinned, 16 bytes;
 * NULL = no copy) and, if clear != 0, their reset behind it. A caller that records an event after this call reads the
 * flags of exactly the work enqueued before it once the event has completed — the search loop of
 * ripor_amd/evaluate.py checks batch n this way while batch n + 1 runs (the reference loop synchronises >= 2*B*Q times
 * per step). */
int rpr_status_words_async(rpr_ctx* ctx, void* stream, uint32_t* host_words, int clear);
int rpr_model_f32_only(const rpr_model* model);

/* ---- measurement ---- */
/* Enable/disable per-kernel-class hipEvent timing (forces eager launches while enabled). */
int rpr_profile_enable(rpr_ctx* ctx, int enable);
int rpr_profile_reset(rpr_ctx* ctx);
/* Synchronises the recorded events and returns the accumulated statistics for one class. */
int rpr_profile_get(rpr_ctx* ctx, int kernel_class, rpr_kernel_stats* out);
/* Bytes of device memory the ctx currently holds for workspaces + KV cache. */
int64_t rpr_workspace_bytes(const rpr_ctx* ctx);

/* ---- single-operator entry points (kernel-level parity tests; same kernels the search uses) ---- */
/* C[M,N] = act(A[M,K] @ W[N,K]^T) (+ residual[M,N]); fp32 MFMA. K % 32 == 0, N % 32 == 0. */
int rpr_op_linear(rpr_ctx* ctx, const float* A, const float* W, const float* residual, float* C,
                  int32_t M, int32_t N, int32_t K, int32_t relu, void* stream);
/* The bf16 GEMM kernels of the fine-tune step (RPR_PREC_BF16; row f4), as a kernel-level parity hook: operands rounded to
 * bf16, one bf16 MFMA per product, fp32 accumulation. n_products = 0: C[M,N] = act(A W^T) (+ residual) through the step's
 * own kernel choice (128-row LDS-DMA tiles, 256x256 ping-pong tiles, split-K for long reductions into few tiles).
 * n_products = 1..8: the grouped launch of the weight-gradient products (gemm_h2_pp_group_kernel, one K-loop per tile):
 * product i = rows [0, M - 256 i) of A against W, written to C + i * M * N (row stride N); residual and relu must be 0.
 * K % 64 == 0. Reference arithmetic: torch bf16 autocast matmul with fp32 accumulation (tasks/trainer.py:229). */
int rpr_op_linear_bf16(rpr_ctx* ctx, const float* A, const float* W, const float* residual, float* C,
                       int32_t M, int32_t N, int32_t K, int32_t relu, int32_t n_products, void* stream);
/* out[rows,d] = w * x * rsqrt(mean(x^2) + eps) */
int rpr_op_rmsnorm(rpr_ctx* ctx, const float* x, const float* w, float* out, int32_t rows, int32_t d,
                   float eps, void* stream);
/* Encoder forward only (reference generation.py:132-137): out [Q, Lq, d_model]. */
int rpr_encode(rpr_ctx* ctx, rpr_model* model, const int32_t* input_ids, const int32_t* attention_mask,
               int32_t Q, int32_t Lq, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RIPOR_HIP_H */
