// Trie-constrained top-B for large beams as a radix select (gfx950, wave64).
//
// The evaluation script of the reference runs the retrieval with --topk=1000 (full_evaluate_t5seq_aq_encoder.sh:191-199):
// a step then ranks B * V = 256 000 float64 candidates per query (tasks/generation.py:453-463 mask + combine, :484-492
// torch.topk(2B), :496-503 BeamSearchScorer.process = the first B of them). One block per query (select_kernel,
// beam_kernels.hip) sorts per-thread lists in LDS; with many beams that is hundreds of microseconds per step on a handful
// of CUs. Here the same selection runs as five small launches over all the CUs:
//
//   rs_mask_kernel     per block RS_BPB beams: children of every beam from the trie's child arrays (coalesced reads of
//                      lvl0 / lvl1 / the CSR levels, trie.h; one wave per beam) -> child bitmap + child row ranges; the
//                      candidates' order-preserving 64-bit keys; histogram of the top 12 key bits = sign and exponent (wave
//                      ballot + popcount, LDS, one global atomic per non-empty bin)
//   rs_hist_kernel x2  the bin holding the B-th best candidate is found from the previous histogram (every block redoes the
//                      scan, ~1 us); histogram of the next 6, then 11 bits over the candidates inside it. (Scores of one step
//                      share an exponent or two: the 12-bit pass leaves ~all valid candidates in one bin; 64 bins for the
//                      next digit keep that pass at 64 global atomics per block — with 2048 bins it was one per candidate,
//                      27-45 us at beam 1000 —, and what falls into one of them is a few hundred candidates.)
//   rs_collect_kernel  candidates above the 29-bit threshold prefix -> winners, candidates equal to it -> tie list
//   rs_finish_kernel   one block per query: winners + ties (ties beyond the sort buffer: a block-level radix select on the
//                      remaining 35 key bits and the candidate index first), bitonic sort by (key desc, candidate index asc) —
//                      select_kernel's order — and the next beam state (select_kernel's phase D)
//   rs_step0_kernel    step 0 when it is computed once per query (SelectArgs::shared0): beam 0 is alive, beams 1 .. B-1 are
//                      B - 1 copies of one dead beam (-1e9, generation.py:418-420) — (B - 1) * V candidates in exact ties,
//                      index order decides. One block per query sorts the 2 V distinct candidates and writes the B winners
//                      from the run structure of the dead ones (the five launches above took 0.7 ms on it at beam 1000).
//
// Same float semantics as select_kernel: candidate = ((double)logit_f32 + (valid ? 0 : -1e9)) + beam_score; ties by
// ascending flat index beam * V + token. Keys are recomputed from the logits in every pass (4 bytes per candidate instead
// of an 8-byte key array).
#include <cstdlib>

#include "common.h"

namespace rpr {

namespace {

constexpr int RS_D0 = 12, RS_D1 = 6, RS_D2 = 11;      // digits of the three passes (key bits 63..52, 51..46, 45..35)
constexpr int RS_N0 = 1 << RS_D0, RS_N1 = 1 << RS_D1, RS_N2 = 1 << RS_D2;
constexpr int RS_HIST = RS_N0 + RS_N1 + RS_N2;        // histogram words per query
constexpr int RS_PREFIX = RS_D0 + RS_D1 + RS_D2;      // key bits the passes fix
constexpr int RS_BPB = 8;                             // beams per block of the candidate passes
constexpr int RS_NARROW = TRIE_NARROW;                // rows of a trie node one wave enumerates directly (one row per lane)
constexpr int RS_SORT_CAP = 8192;                     // entries of the finish kernel's LDS sort (12 bytes each)
constexpr int RS_MAX_V = 2048;                        // widest token axis (the step-0 kernel sorts 2 V entries in LDS: 82 KB)

__device__ __forceinline__ unsigned long long key_of_score(double s) {
  s = s + 0.0;                                        // -0.0 -> +0.0: equal doubles have equal keys
  const unsigned long long u = (unsigned long long)__double_as_longlong(s);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double score_of_key(unsigned long long k) {
  const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)u);
}

// everything a pass needs to recompute the key of candidate `item` (= beam * V + token) of one query
struct CandCtx {
  const float* lg_q;                  // logits of the query: [B][V], or [V] when shared0
  const unsigned long long* valid_q;  // child bitmap [B * V / 64]
  const double* score_q;              // beam scores [B]
  const float* lstat_q;               // log-softmax: (max, log sum exp) per beam
  int V, Vr, shared0, log_softmax;
};
__device__ __forceinline__ unsigned long long cand_key(const CandCtx& x, int item, unsigned long long valid_word, double bscore, int b) {
  const int c = item - b * x.V;
  float lg = x.lg_q[x.shared0 ? c : item];
  if (x.log_softmax) lg = (lg - x.lstat_q[2 * b]) - x.lstat_q[2 * b + 1];
  const bool ok = (valid_word >> (item & 63)) & 1ull;
  double s = ((double)lg + (ok ? 0.0 : -1e9)) + bscore;
  if (c >= x.Vr) s = -INFINITY;       // padding column of the token axis: behind every candidate of the reference
  return key_of_score(s);
}
__device__ __forceinline__ unsigned long long cand_key_any(const CandCtx& x, int item) {
  const int b = item / x.V;
  return cand_key(x, item, x.valid_q[item >> 6], x.score_q[b], b);
}

__device__ __forceinline__ CandCtx make_ctx(const SelectArgs& a, int q) {
  CandCtx x;
  const size_t r0 = (size_t)q * a.B;
  x.lg_q = a.logits + (a.shared0 ? (size_t)q * a.V : r0 * a.V);
  x.valid_q = a.rs.valid + (size_t)q * ((size_t)a.B * a.V >> 6);
  x.score_q = a.cur.score + r0;
  x.lstat_q = a.rs.lstat + 2 * r0;
  x.V = a.V; x.Vr = a.Vreal > 0 ? a.Vreal : a.V; x.shared0 = a.shared0; x.log_softmax = a.log_softmax;
  return x;
}

// histogram of digit d over the active lanes of a wave: one LDS atomic per distinct digit (ballot + popcount)
__device__ __forceinline__ void wave_hist(unsigned* h, bool active, unsigned d, int lane) {
  unsigned long long todo = __ballot(active);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned d0 = (unsigned)__shfl((int)d, leader, 64);
    const unsigned long long m = __ballot(active && d == d0);
    if (lane == leader) atomicAdd(&h[d0], (unsigned)__popcll(m));
    todo &= ~m;
  }
}

// The bin of an NB-bin histogram (global memory) that holds the k-th largest element, bins counted from the top, and how
// many elements are still wanted from inside it. All 256 threads of the block call it; sh = 8 ints of LDS.
template <int NB>
__device__ __forceinline__ void find_digit(const unsigned* __restrict__ gh, int k, int* sh, int tid, int& digit, int& krem) {
  constexpr int PER = NB >= 1024 ? NB / 256 : 4;      // bins per thread (a multiple of 4: 16-byte loads), NB / PER threads work
  constexpr int NT = NB / PER;
  unsigned v[PER];
  unsigned sum = 0;
  const int top = NB - 1 - PER * tid;                 // this thread owns bins top, top - 1, .., top - PER + 1
  if (tid < NT) {
#pragma unroll
    for (int j = 0; j < PER / 4; ++j) {
      const uint4 x = *reinterpret_cast<const uint4*>(gh + top - 4 * j - 3);
      v[4 * j] = x.w; v[4 * j + 1] = x.z; v[4 * j + 2] = x.y; v[4 * j + 3] = x.x;
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) sum += v[i];
  }
  const int lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { sh[4] = 0; sh[5] = 1; }
  unsigned incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned up = (unsigned)__shfl_up((int)incl, o, 64);
    if (lane >= o) incl += up;
  }
  if (lane == 63) sh[wave] = (int)incl;
  __syncthreads();
  if (tid < NT) {
    unsigned above = incl - sum;
    for (int w = 0; w < wave; ++w) above += (unsigned)sh[w];
    if ((int)above < k && k <= (int)(above + sum)) {   // exactly one thread
      unsigned acc = above;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        if ((int)acc < k && k <= (int)(acc + v[i])) { sh[4] = top - i; sh[5] = k - (int)acc; }
        acc += v[i];
      }
    }
  }
  __syncthreads();
  digit = sh[4]; krem = sh[5];
  __syncthreads();                                    // sh may be reused by the next call
}

__device__ __forceinline__ int lower_bound_col2(const uint16_t* __restrict__ codes, int Lc, int col, int lo, int hi, int c) {
  while (lo < hi) {
    const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
    if ((int)codes[(size_t)mid * Lc + col] < c) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Children of one beam, by one wave: the child bitmap (LDS words vb, zeroed by the caller), and for every child token c the
// row range [lb_b[c], chi_b[c]). r = the beam's row in the beam state (tokens of its prefix).
__device__ __forceinline__ void beam_children(const SelectArgs& a, size_t r, int lo, int hi, int lane, unsigned long long* vb,
                                              int32_t* __restrict__ lb_b, int32_t* __restrict__ chi_b) {
  const int V = a.V, t = a.t, Lc = a.Lc;
  if (!(t < Lc && lo < hi)) return;
  const bool tabs = a.lvl_V > 0;                        // the trie's child arrays were built for this vocab size
  const int Vt = a.lvl_V;
  const int n = hi - lo;
  if (n <= RS_NARROW) {
    // narrow node: one row per lane, a child starts where the code of column t changes
    const bool in = lane < n;
    const int c = in ? (int)a.codes[(size_t)(lo + lane) * Lc + t] : -1;
    const int cprev = __shfl_up(c, 1, 64);
    const bool st = in && (lane == 0 || c != cprev);
    const unsigned long long m = __ballot(st);
    if (st) {
      const unsigned long long higher = lane == 63 ? 0ull : (m >> (lane + 1)) << (lane + 1);
      const int nxt = higher ? __ffsll((long long)higher) - 1 : n;
      lb_b[c] = lo + lane; chi_b[c] = lo + nxt;
      atomicOr(&vb[c >> 6], 1ull << (c & 63));
    }
    return;
  }
  if (tabs && (t == 0 || (t == 1 && a.lvl1))) {
    // levels 0 / 1: dense tables of lower bounds
    const int32_t* tab = t == 0 ? a.lvl0 : a.lvl1 + (size_t)a.cur.tokens[r * a.cur.ld] * Vt;
    for (int c = lane; c < V; c += 64) {
      const bool in = c < Vt;
      const int l = in ? tab[c] : 0, h = in ? tab[c + 1] : 0;
      const bool ok = h > l;
      if (ok) { lb_b[c] = l; chi_b[c] = h; }
      const unsigned long long m = __ballot(ok);
      if (lane == 0) vb[c >> 6] = m;
    }
    return;
  }
  // deeper levels: the node's children are consecutive entries of the level's CSR arrays
  const int di = t - 2;
  if (tabs && di >= 0 && di < a.n_deep && a.d_n[di] > 0) {
    const int32_t* st = a.d_start[di];
    const int nt = a.d_n[di];
    int k0;
    if (t == 2 && a.idx2) {
      k0 = a.idx2[(size_t)a.cur.tokens[r * a.cur.ld] * Vt + a.cur.tokens[r * a.cur.ld + 1]];
    } else {
      // the entry that starts at row lo: lower bound over the level's start rows, 64 probes per round (one per lane)
      int l = 0, h = nt;                                // entries < l start before lo, entries >= h start at or behind it
      while (h - l > 64) {
        const long span = (long)h - l;
        const bool less = st[l + (int)((span * (lane + 1)) / 65)] < lo;
        const int cnt = __popcll(__ballot(less));       // probes ascend with the lane: the first cnt of them are below
        const int nl = cnt > 0 ? l + (int)((span * cnt) / 65) + 1 : l;
        const int nh = cnt < 64 ? l + (int)((span * (cnt + 1)) / 65) : h;
        l = nl; h = nh;
      }
      const bool less = l + lane < h && st[l + lane] < lo;
      k0 = l + __popcll(__ballot(less));
    }
    if (k0 >= 0 && k0 < nt && st[k0] == lo) {
      const uint16_t* tk = a.d_tok[di];
      for (int k = k0 + lane;; k += 64) {
        const int s0 = k < nt ? st[k] : 0x7fffffff;
        const bool in = s0 < hi;
        if (in) {
          const int c = tk[k];
          int nxt = k + 1 < nt ? st[k + 1] : hi;
          nxt = nxt < hi ? nxt : hi;
          lb_b[c] = s0; chi_b[c] = nxt;
          atomicOr(&vb[c >> 6], 1ull << (c & 63));
        }
        if (__ballot(in) != ~0ull) break;
      }
      return;
    }
  }
  // no child array for this node (vocab mismatch, level beyond the build): two binary searches per token
  for (int c = lane; c < V; c += 64) {
    const int l = lower_bound_col2(a.codes, Lc, t, lo, hi, c);
    const int h = lower_bound_col2(a.codes, Lc, t, l, hi, c + 1);
    const bool ok = h > l;
    if (ok) { lb_b[c] = l; chi_b[c] = h; }
    const unsigned long long m = __ballot(ok);
    if (lane == 0) vb[c >> 6] = m;
  }
}

// fp32 log_softmax statistics of one logits row over its real columns (generation.py:453-455), as select_kernel; one wave
__device__ __forceinline__ void row_lstat(const float* __restrict__ row, int Vr, int lane, float* out2) {
  float mx = -INFINITY;
  for (int c = lane; c < Vr; c += 64) mx = fmaxf(mx, row[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sm = 0.f;
  for (int c = lane; c < Vr; c += 64) sm += expf(row[c] - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
  if (lane == 0) { out2[0] = mx; out2[1] = logf(sm); }
}

// ---- pass 0: children of every beam, child bitmap, first histogram ------------------------------------------------------
__global__ __launch_bounds__(256) void rs_mask_kernel(SelectArgs a, int nbq) {
  __shared__ unsigned hist[RS_N0];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* vbits = reinterpret_cast<unsigned long long*>(smem_raw);   // child bitmaps of the block's beams [RS_BPB][V / 64]
  __shared__ float lst[RS_BPB * 2];
  const int q = blockIdx.x / nbq, bb = blockIdx.x - q * nbq;
  if (a.nq_dev && q >= *a.nq_dev) return;               // compacted stage: block-uniform
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = a.B, V = a.V, wpb = V >> 6;             // bitmap words per beam
  const int b0 = bb * RS_BPB, b1 = min(B, b0 + RS_BPB), nb = b1 - b0;
  const size_t r0 = (size_t)q * B;
  for (int i = tid; i < RS_N0; i += 256) hist[i] = 0u;
  for (int i = tid; i < nb * wpb; i += 256) vbits[i] = 0ull;
  __syncthreads();
  for (int bl = wave; bl < nb; bl += 4) {               // one wave per beam
    const size_t r = r0 + b0 + bl;
    beam_children(a, r, a.cur.lo[r], a.cur.hi[r], lane, vbits + bl * wpb, a.lb_scratch + r * V, a.rs.chi + r * V);
    if (a.log_softmax) row_lstat(a.logits + (a.shared0 ? (size_t)q * V : r * V), a.Vreal > 0 ? a.Vreal : V, lane, lst + 2 * bl);
  }
  __syncthreads();
  CandCtx x = make_ctx(a, q);
  unsigned long long* vg = a.rs.valid + (size_t)q * ((size_t)B * wpb) + (size_t)b0 * wpb;
  for (int i = tid; i < nb * wpb; i += 256) vg[i] = vbits[i];
  if (a.tap_valid) for (int i = tid; i < nb * wpb; i += 256) a.tap_valid[(size_t)q * ((size_t)B * wpb) + (size_t)b0 * wpb + i] = vbits[i];
  if (a.log_softmax) {
    for (int i = tid; i < 2 * nb; i += 256) a.rs.lstat[2 * (r0 + b0) + i] = lst[i];
    x.lstat_q = lst - 2 * b0;                           // this block's beams from LDS (the global copy is for the later passes)
  }
  for (int it = b0 * V + tid; it < b1 * V; it += 256) {  // V % 64 == 0: a wave stays inside one beam and one bitmap word
    const int b = it / V;
    const unsigned long long key = cand_key(x, it, vbits[(it - b0 * V) >> 6], x.score_q[b], b);
    wave_hist(hist, true, (unsigned)(key >> (64 - RS_D0)), lane);
  }
  __syncthreads();
  unsigned* gh = a.rs.hist + (size_t)q * RS_HIST;
  for (int i = tid; i < RS_N0; i += 256) { const unsigned v = hist[i]; if (v) atomicAdd(&gh[i], v); }
}

// ---- passes 1 and 2: histogram of the next digit inside the threshold bin(s) --------------------------------------------
template <int PASS>
__global__ __launch_bounds__(256) void rs_hist_kernel(SelectArgs a, int nbq) {
  constexpr int NB = PASS == 1 ? RS_N1 : RS_N2;
  constexpr int FIXED = PASS == 1 ? RS_D0 : RS_D0 + RS_D1;          // key bits fixed by the earlier passes
  constexpr int DIG = PASS == 1 ? RS_D1 : RS_D2;
  __shared__ unsigned hist[NB];
  __shared__ int sh[8];
  const int q = blockIdx.x / nbq, bb = blockIdx.x - q * nbq;
  if (a.nq_dev && q >= *a.nq_dev) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int B = a.B, V = a.V;
  const int b0 = bb * RS_BPB, b1 = min(B, b0 + RS_BPB);
  for (int i = tid; i < NB; i += 256) hist[i] = 0u;
  unsigned* gh = a.rs.hist + (size_t)q * RS_HIST;
  int d0, d1 = 0, k;
  find_digit<RS_N0>(gh, B, sh, tid, d0, k);
  if (PASS == 2) find_digit<RS_N1>(gh + RS_N0, k, sh, tid, d1, k);
  const unsigned long long want = PASS == 1 ? (unsigned long long)d0 : (((unsigned long long)d0 << RS_D1) | (unsigned long long)d1);
  const CandCtx x = make_ctx(a, q);
  for (int it = b0 * V + tid; it < b1 * V; it += 256) {
    const int b = it / V;
    const unsigned long long key = cand_key(x, it, x.valid_q[it >> 6], x.score_q[b], b);
    // (one LDS atomic per candidate: inside the threshold bin the next digit differs from lane to lane, and the ballot loop of
    // wave_hist — made for pass 0, where a wave holds one or two exponents — ran once per distinct digit: 28 us at beam 1000)
    if ((key >> (64 - FIXED)) == want) atomicAdd(&hist[(unsigned)(key >> (64 - FIXED - DIG)) & (NB - 1)], 1u);
  }
  __syncthreads();
  unsigned* go = gh + (PASS == 1 ? RS_N0 : RS_N0 + RS_N1);
  for (int i = tid; i < NB; i += 256) { const unsigned v = hist[i]; if (v) atomicAdd(&go[i], v); }
}

// ---- collect: candidates above the threshold prefix, and the ties on it -------------------------------------------------
__global__ __launch_bounds__(256) void rs_collect_kernel(SelectArgs a, int nbq) {
  __shared__ int sh[8];
  const int q = blockIdx.x / nbq, bb = blockIdx.x - q * nbq;
  if (a.nq_dev && q >= *a.nq_dev) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int B = a.B, V = a.V;
  const int b0 = bb * RS_BPB, b1 = min(B, b0 + RS_BPB);
  const unsigned* gh = a.rs.hist + (size_t)q * RS_HIST;
  int d0, d1, d2, k;
  find_digit<RS_N0>(gh, B, sh, tid, d0, k);
  find_digit<RS_N1>(gh + RS_N0, k, sh, tid, d1, k);
  find_digit<RS_N2>(gh + RS_N0 + RS_N1, k, sh, tid, d2, k);
  const unsigned long long thr = ((unsigned long long)d0 << (RS_D1 + RS_D2)) | ((unsigned long long)d1 << RS_D2) | (unsigned long long)d2;
  const CandCtx x = make_ctx(a, q);
  unsigned* cnt = a.rs.cnt + (size_t)q * 4;
  int32_t* win = a.rs.win + (size_t)q * B;
  int32_t* tie = a.rs.tie + (size_t)q * ((size_t)B * V);
  for (int it = b0 * V + tid; it < b1 * V; it += 256) {
    const int b = it / V;
    const unsigned long long p = cand_key(x, it, x.valid_q[it >> 6], x.score_q[b], b) >> (64 - RS_PREFIX);
    const bool w = p > thr, e = p == thr;
    const unsigned long long mw = __ballot(w), me = __ballot(e);   // one atomic per wave and list
    unsigned bw = 0, be = 0;
    if (lane == 0) {
      if (mw) bw = atomicAdd(&cnt[0], (unsigned)__popcll(mw));
      if (me) be = atomicAdd(&cnt[1], (unsigned)__popcll(me));
    }
    bw = (unsigned)__shfl((int)bw, 0, 64); be = (unsigned)__shfl((int)be, 0, 64);
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    if (w) win[bw + (unsigned)__popcll(mw & below)] = it;
    if (e) tie[be + (unsigned)__popcll(me & below)] = it;
  }
}

struct Ent { unsigned long long k; int it; };
__device__ __forceinline__ bool ent_better(const Ent& a, const Ent& b) { return a.k > b.k || (a.k == b.k && a.it < b.it); }

// bitonic sort of the P2 (power of two) entries (ck, ci) in LDS by (key descending, item ascending); NT threads
template <int NT>
__device__ __forceinline__ void lds_sort(unsigned long long* ck, int* ci, int P2, int tid) {
  for (int k = 2; k <= P2; k <<= 1) {
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      for (int p = tid; p < P2 / 2; p += NT) {
        const int lo_i = 2 * p - (p & (jj - 1));
        const int hi_i = lo_i + jj;
        const bool desc = (lo_i & k) == 0;
        Ent e0{ck[lo_i], ci[lo_i]}, e1{ck[hi_i], ci[hi_i]};
        if (ent_better(e1, e0) == desc) { ck[lo_i] = e1.k; ci[lo_i] = e1.it; ck[hi_i] = e0.k; ci[hi_i] = e0.it; }
      }
      __syncthreads();
    }
  }
}

// ---- finish: sort the winners, write the next beam state ------------------------------------------------------------------
__global__ __launch_bounds__(1024) void rs_finish_kernel(SelectArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ unsigned h8[256];
  __shared__ int sh[8];
  const int q = blockIdx.x, tid = threadIdx.x;
  if (a.nq_dev && q >= *a.nq_dev) return;
  const int B = a.B, V = a.V, t = a.t;
  unsigned* cnt = a.rs.cnt + (size_t)q * 4;
  const int n_gt = (int)cnt[0], n_tie = (int)cnt[1], need = B - n_gt;   // n_gt < B <= n_gt + n_tie
  const CandCtx x = make_ctx(a, q);
  const int32_t* win = a.rs.win + (size_t)q * B;
  int32_t* tie = a.rs.tie + (size_t)q * ((size_t)B * V);
  int m = n_tie;                                       // ties that enter the sort
  // LDS: keys [cap] + items [cap], cap = the launch's sort capacity (>= next power of two of B)
  const int cap = a.rs.sort_cap;
  unsigned long long* ck = reinterpret_cast<unsigned long long*>(smem_raw);
  int* ci = reinterpret_cast<int*>(ck + cap);
  if (n_gt + n_tie > cap) {
    // More ties than the sort buffer holds (degenerate scores: whole codebooks tie). Block-level radix select of the `need`
    // best ties by the composite (low 35 key bits, then descending candidate index — 29 bits, select_radix_fits): 8 passes
    // of 8 bits over the tie list in global memory; composites are unique, so exactly `need` ties are >= the threshold.
    constexpr unsigned long long LOW = (1ull << (64 - RS_PREFIX)) - 1ull;
    auto comp_of = [&](unsigned long long key, int it) { return ((key & LOW) << 29) | (unsigned long long)(0x1fffffffu - (unsigned)it); };
    unsigned long long prefix = 0ull;                  // bits fixed so far, at the top of the composite
    int k = need;
    for (int pass = 0; pass < 8; ++pass) {
      const int shift = 56 - 8 * pass;
      if (tid < 256) h8[tid] = 0u;
      __syncthreads();
      for (int i = tid; i < n_tie; i += 1024) {
        const int it = tie[i];
        const unsigned long long comp = comp_of(cand_key_any(x, it), it);
        if (pass == 0 || (comp >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&h8[(unsigned)(comp >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        int acc = 0, d = 255;
        for (; d > 0; --d) { if (acc + (int)h8[d] >= k) break; acc += (int)h8[d]; }
        sh[0] = d; sh[1] = k - acc;
      }
      __syncthreads();
      prefix |= (unsigned long long)sh[0] << shift;
      k = sh[1];
      __syncthreads();
    }
    // prefix == composite of the need-th best tie
    if (tid == 0) sh[2] = 0;
    __syncthreads();
    for (int i = tid; i < n_tie; i += 1024) {
      const int it = tie[i];
      const unsigned long long key = cand_key_any(x, it);
      if (comp_of(key, it) >= prefix) { const int slot = atomicAdd(&sh[2], 1); ck[n_gt + slot] = key; ci[n_gt + slot] = it; }
    }
    __syncthreads();
    m = need;
  } else {
    for (int i = tid; i < n_tie; i += 1024) { const int it = tie[i]; ck[n_gt + i] = cand_key_any(x, it); ci[n_gt + i] = it; }
  }
  for (int i = tid; i < n_gt; i += 1024) { const int it = win[i]; ck[i] = cand_key_any(x, it); ci[i] = it; }
  const int n = n_gt + m;
  int P2 = 64;                                         // (beam 100: 128 entries, 28 compare-exchange rounds instead of the 55 of 1024)
  while (P2 < n) P2 <<= 1;
  for (int i = n + tid; i < P2; i += 1024) { ck[i] = 0ull; ci[i] = 0x7fffffff; }
  __syncthreads();
  lds_sort<1024>(ck, ci, P2, tid);
  // ---- next beam state: new slot j <- winner j (select_kernel's phase D) ----
  const int ld = a.cur.ld;
  const size_t r0 = (size_t)q * B;
  for (int j = tid; j < B; j += 1024) {
    const int item = ci[j];
    const int b = item / V, c = item - b * V;
    const bool ok = (x.valid_q[item >> 6] >> (item & 63)) & 1ull;
    const size_t r = r0 + j;
    const double s = score_of_key(ck[j]);
    a.nxt.score[r] = s;
    a.nxt.lo[r] = ok ? a.lb_scratch[(r0 + b) * V + c] : 0;
    a.nxt.hi[r] = ok ? a.rs.chi[(r0 + b) * V + c] : 0;
    a.nxt.tokens[r * ld + t] = (uint16_t)c;
    a.nxt.anc[r * ld + t] = (uint16_t)(a.shared0 ? 0 : b);
    if (a.tap_scores) a.tap_scores[r] = s;
    if (a.tap_tokens) a.tap_tokens[r] = c;
    if (a.tap_parent) a.tap_parent[r] = b;
  }
  for (int i = tid; i < B * t; i += 1024) {
    const int j = i / t, p = i - j * t;
    const int b = ci[j] / V;
    a.nxt.tokens[(r0 + j) * ld + p] = a.cur.tokens[(r0 + b) * ld + p];
    a.nxt.anc[(r0 + j) * ld + p] = a.cur.anc[(r0 + b) * ld + p];
  }
  // leave the query's histograms and counters zeroed for the next step
  unsigned* gh = a.rs.hist + (size_t)q * RS_HIST;
  for (int i = tid; i < RS_HIST; i += 1024) gh[i] = 0u;
  if (tid < 4) cnt[tid] = 0u;
}

// ---- step 0 of a search whose step 0 runs once per query (shared0) -------------------------------------------------------
// Every beam stands on the root with the query's one logits row; beam 0 has score 0, beams 1 .. B-1 the same dead score
// (init_beams_kernel). The B * V candidates are therefore 2 V distinct (key, token) pairs: V of beam 0 and V of "a dead beam",
// each of the latter standing for B - 1 candidates that differ in the beam only. In select order (key descending, then index
// beam * V + token ascending) the dead candidates of a maximal run of g tokens with one key come out beam-major: (1, c_1),
// .., (1, c_g), (2, c_1), .. One block: sort the 2 V pairs, walk the runs until B slots are covered, fill the slots.
__global__ __launch_bounds__(256) void rs_step0_kernel(SelectArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ float lst[2];
  __shared__ int nseg_s;
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (a.nq_dev && q >= *a.nq_dev) return;
  const int B = a.B, V = a.V, wpb = V >> 6;
  int P2 = 256;
  while (P2 < 2 * V) P2 <<= 1;
  // LDS: keys [P2] | entries [P2] | bitmap [V / 64] | first slot of every segment [2 V + 1] | its first sorted position [2 V + 1]
  unsigned long long* ck = reinterpret_cast<unsigned long long*>(smem_raw);
  int* ci = reinterpret_cast<int*>(ck + P2);
  unsigned long long* vb = reinterpret_cast<unsigned long long*>(ci + P2);
  int* seg_off = reinterpret_cast<int*>(vb + wpb);
  int* seg_pos = seg_off + 2 * V + 1;
  const size_t r0 = (size_t)q * B;
  for (int i = tid; i < wpb; i += 256) vb[i] = 0ull;
  __syncthreads();
  if (wave == 0) beam_children(a, r0, a.cur.lo[r0], a.cur.hi[r0], lane, vb, a.lb_scratch + r0 * V, a.rs.chi + r0 * V);
  if (wave == 1 && a.log_softmax) row_lstat(a.logits + (size_t)q * V, a.Vreal > 0 ? a.Vreal : V, lane, lst);
  __syncthreads();
  CandCtx x = make_ctx(a, q);
  x.lstat_q = lst;                                      // one row: "beam 0" for both halves below
  const double s_live = a.cur.score[r0], s_dead = B > 1 ? a.cur.score[r0 + 1] : 0.0;
  for (int e = tid; e < P2; e += 256) {                 // entry e < V: token e of beam 0; V <= e < 2 V: token e - V of a dead beam
    unsigned long long key = 0ull;
    if (e < 2 * V && (e < V || B > 1)) { const int c = e < V ? e : e - V; key = cand_key(x, c, vb[c >> 6], e < V ? s_live : s_dead, 0); }
    ck[e] = key; ci[e] = (e < 2 * V && (e < V || B > 1)) ? e : 0x7fffffff;
  }
  __syncthreads();
  lds_sort<256>(ck, ci, P2, tid);                       // ties: entry index ascending = beam 0 first, then tokens ascending
  // segments in sorted order: a live entry = 1 slot; a run of g dead entries with one key = g * (B - 1) slots
  if (tid == 0) {
    int ns = 0, off = 0, p = 0;
    const int n = B > 1 ? 2 * V : V;
    while (p < n && off < B) {
      seg_off[ns] = off; seg_pos[ns] = p;
      if (ci[p] < V) { off += 1; p += 1; }
      else {
        int g = 1;
        while (p + g < n && ci[p + g] >= V && ck[p + g] == ck[p]) ++g;
        off += g * (B - 1); p += g;
      }
      ++ns;
    }
    seg_off[ns] = off; seg_pos[ns] = p; nseg_s = ns;
  }
  __syncthreads();
  const int ns = nseg_s, ld = a.cur.ld;
  for (int j = tid; j < B; j += 256) {
    int lo = 0, hi = ns;                                // last segment with seg_off <= j
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_off[mid] <= j) lo = mid; else hi = mid; }
    const int p = seg_pos[lo], k = j - seg_off[lo];
    int b, c;
    unsigned long long key;
    if (ci[p] < V) { b = 0; c = ci[p]; key = ck[p]; }
    else {
      const int g = seg_pos[lo + 1] - p;                // tokens of the run
      b = 1 + k / g; c = ci[p + k % g] - V; key = ck[p];
    }
    const bool ok = (vb[c >> 6] >> (c & 63)) & 1ull;
    const size_t r = r0 + j;
    a.nxt.score[r] = score_of_key(key);
    a.nxt.lo[r] = ok ? a.lb_scratch[r0 * V + c] : 0;    // every beam stands on the root: beam 0's child ranges
    a.nxt.hi[r] = ok ? a.rs.chi[r0 * V + c] : 0;
    a.nxt.tokens[r * ld] = (uint16_t)c;
    a.nxt.anc[r * ld] = 0;                              // shared0: position 0 lives in slot 0
    (void)b;
  }
}

int next_pow2(int n) { int p = 1024; while (p < n) p <<= 1; return p; }

}  // namespace

// RPR_SELECT_RADIX: 0 = never, 1 = every selection that fits (tests), unset = from 32 beams on. Measured on MI355X (t5-base
// dims, 8.8 M-doc trie, profiles/r06e_radix_beam100.txt): beam 100 at the reference script's batch of 4 (prefix 4 / 8 / 16)
// 531 / 274 / 305 -> 606 / 307 / 330 queries/s (the single block spends 100 rounds of a block-wide arg-max per step: 290 us,
// the five launches 60 us); beam 100 at 162-214 queries in flight and beam 10 at 2150: within +-1 %. Beam 10 stays on the
// single block (one launch instead of five on the single-query path).
bool select_radix_wanted(int B, int V) {
  if (!select_radix_fits(B, V)) return false;
  const char* e = getenv("RPR_SELECT_RADIX");   // read per call: a test switches it between searches
  if (e) return atoi(e) != 0;
  return B >= 32;
}

bool select_radix_fits(int B, int V) { return V % 64 == 0 && V <= RS_MAX_V && B >= 1 && B <= RS_SORT_CAP && (long)B * V < (1L << 29); }

// bytes of RadixWs scratch for Q queries (everything but chi, which has the shape of SelectArgs::lb_scratch)
size_t select_radix_ws_bytes(int Q, int B, int V) {
  const size_t n = (size_t)B * V;
  return (size_t)Q * (RS_HIST * 4 + 16 + n / 8 + (size_t)B * 8 + (size_t)B * 4 + n * 4) + 256;
}

void select_radix_carve(RadixWs& w, void* base, int32_t* chi, int Q, int B, int V) {
  const size_t n = (size_t)B * V;
  unsigned char* p = static_cast<unsigned char*>(base);
  w.hist = reinterpret_cast<unsigned*>(p); p += (size_t)Q * RS_HIST * 4;
  w.valid = reinterpret_cast<unsigned long long*>(p); p += (size_t)Q * (n / 8);
  w.lstat = reinterpret_cast<float*>(p); p += (size_t)Q * B * 8;
  w.cnt = reinterpret_cast<unsigned*>(p); p += (size_t)Q * 16;
  w.win = reinterpret_cast<int32_t*>(p); p += (size_t)Q * B * 4;
  w.tie = reinterpret_cast<int32_t*>(p);
  w.chi = chi;
  w.sort_cap = 0;
}

static size_t step0_smem(int V) {
  int P2 = 256;
  while (P2 < 2 * V) P2 <<= 1;
  return (size_t)P2 * 12 + (size_t)(V / 64) * 8 + ((size_t)4 * V + 2) * 4 + 16;
}

hipError_t init_select_radix_attributes() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rs_finish_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RS_SORT_CAP * 12);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(rs_step0_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)step0_smem(RS_MAX_V));
}

// zero the histograms and counters of Q queries (once per search; rs_finish_kernel re-zeroes them after every step)
hipError_t launch_select_radix_reset(const RadixWs& w, int Q, hipStream_t s) {
  hipError_t e = launch_zero_u64(reinterpret_cast<unsigned long long*>(w.hist), (size_t)Q * RS_HIST / 2, s);
  if (e != hipSuccess) return e;
  return launch_zero_u64(reinterpret_cast<unsigned long long*>(w.cnt), (size_t)Q * 2, s);
}

hipError_t launch_select_radix(const SelectArgs& a_in, hipStream_t s) {
  SelectArgs a = a_in;
  if (!select_radix_fits(a.B, a.V) || !a.rs.hist || !a.rs.chi) return hipErrorInvalidValue;
  if (a.t == 0 && a.shared0 && !a.tap_valid && !a.tap_scores && !a.tap_tokens && !a.tap_parent) {
    hipLaunchKernelGGL(rs_step0_kernel, dim3((unsigned)a.Q), dim3(256), step0_smem(a.V), s, a);
    return hipGetLastError();
  }
  const int nbq = (a.B + RS_BPB - 1) / RS_BPB;
  // the sort holds the winners plus the ties on the threshold prefix: twice the beam count covers every ordinary step,
  // more ties than that go through the finish kernel's own radix select first
  a.rs.sort_cap = std::min(RS_SORT_CAP, next_pow2(2 * a.B));
  const dim3 grid((unsigned)(a.Q * nbq)), blk(256);
  hipLaunchKernelGGL(rs_mask_kernel, grid, blk, (size_t)RS_BPB * (a.V / 64) * 8, s, a, nbq);
  hipLaunchKernelGGL(rs_hist_kernel<1>, grid, blk, 0, s, a, nbq);
  hipLaunchKernelGGL(rs_hist_kernel<2>, grid, blk, 0, s, a, nbq);
  hipLaunchKernelGGL(rs_collect_kernel, grid, blk, 0, s, a, nbq);
  hipLaunchKernelGGL(rs_finish_kernel, dim3((unsigned)a.Q), dim3(1024), (size_t)a.rs.sort_cap * 12, s, a);
  return hipGetLastError();
}

}  // namespace rpr
