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
//                      candidates' order-preserving 64-bit keys; histogram of the top 11 key bits (wave ballot + popcount,
//                      LDS, one global atomic per non-empty bin)
//   rs_hist_kernel x2  the bin holding the B-th best candidate is found from the previous histogram (every block redoes the
//                      2048-bin scan, ~1 us); histogram of the next 11 bits over the candidates inside it
//   rs_collect_kernel  candidates above the 33-bit threshold prefix -> winners, candidates equal to it -> tie list
//   rs_finish_kernel   one block per query: winners + ties (ties beyond the sort buffer: a block-level radix select on the
//                      remaining 31 key bits and the candidate index first), bitonic sort by (key desc, candidate index asc) —
//                      select_kernel's order — and the next beam state (select_kernel's phase D)
//
// Same float semantics as select_kernel: candidate = ((double)logit_f32 + (valid ? 0 : -1e9)) + beam_score; ties by
// ascending flat index beam * V + token. Keys are recomputed from the logits in every pass (4 bytes per candidate instead
// of an 8-byte key array).
#include <cstdlib>

#include "common.h"

namespace rpr {

namespace {

constexpr int RS_BITS = 11, RS_BINS = 1 << RS_BITS;   // digit of one pass
constexpr int RS_BPB = 8;                             // beams per block of the candidate passes
constexpr int RS_NARROW = TRIE_NARROW;                // rows of a trie node one wave enumerates directly (one row per lane)
constexpr int RS_SORT_CAP = 8192;                     // entries of the finish kernel's LDS sort (12 bytes each)

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

// The bin of a 2048-bin histogram (global memory) that holds the k-th largest element, bins counted from the top, and
// how many elements are still wanted from inside it. All threads of the block call it (first 256 work); sh = 8 ints of LDS.
__device__ __forceinline__ void find_digit(const unsigned* __restrict__ gh, int k, int* sh, int tid, int& digit, int& krem) {
  unsigned v[8];
  unsigned sum = 0;
  const int top = RS_BINS - 1 - 8 * tid;              // this thread owns bins top, top - 1, .., top - 7
  if (tid < 256) {
    const uint4 lo4 = *reinterpret_cast<const uint4*>(gh + top - 7), hi4 = *reinterpret_cast<const uint4*>(gh + top - 3);
    v[0] = hi4.w; v[1] = hi4.z; v[2] = hi4.y; v[3] = hi4.x; v[4] = lo4.w; v[5] = lo4.z; v[6] = lo4.y; v[7] = lo4.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += v[i];
  }
  const int lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { sh[4] = 0; sh[5] = 1; }
  unsigned incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned up = (unsigned)__shfl_up((int)incl, o, 64);
    if (lane >= o) incl += up;
  }
  if (tid < 256 && lane == 63) sh[wave] = (int)incl;
  __syncthreads();
  if (tid < 256) {
    unsigned above = incl - sum;
    for (int w = 0; w < wave; ++w) above += (unsigned)sh[w];
    if ((int)above < k && k <= (int)(above + sum)) {   // exactly one thread
      unsigned acc = above;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
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

// ---- pass 0: children of every beam, child bitmap, first histogram ------------------------------------------------------
__global__ __launch_bounds__(256) void rs_mask_kernel(SelectArgs a, int nbq) {
  __shared__ unsigned hist[RS_BINS];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* vbits = reinterpret_cast<unsigned long long*>(smem_raw);   // child bitmaps of the block's beams [RS_BPB][V / 64]
  __shared__ float lst[RS_BPB * 2];
  const int q = blockIdx.x / nbq, bb = blockIdx.x - q * nbq;
  if (a.nq_dev && q >= *a.nq_dev) return;               // compacted stage: block-uniform
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = a.B, V = a.V, t = a.t, Lc = a.Lc, wpb = V >> 6;   // bitmap words per beam
  const int b0 = bb * RS_BPB, b1 = min(B, b0 + RS_BPB), nb = b1 - b0;
  const size_t r0 = (size_t)q * B;
  for (int i = tid; i < RS_BINS; i += 256) hist[i] = 0u;
  for (int i = tid; i < nb * wpb; i += 256) vbits[i] = 0ull;
  __syncthreads();
  const bool tabs = a.lvl_V > 0;                        // the trie's child arrays were built for this vocab size
  const int Vt = a.lvl_V;
  for (int bl = wave; bl < nb; bl += 4) {               // one wave per beam
    const int b = b0 + bl;
    const size_t r = r0 + b;
    const int lo = a.cur.lo[r], hi = a.cur.hi[r];
    int32_t* lb_b = a.lb_scratch + r * V;
    int32_t* chi_b = a.rs.chi + r * V;
    unsigned long long* vb = vbits + bl * wpb;
    if (t < Lc && lo < hi) {
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
      } else if (tabs && (t == 0 || (t == 1 && a.lvl1))) {
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
      } else {
        // deeper levels: the node's children are consecutive entries of the level's CSR arrays
        const int di = t - 2;
        int k0 = -1;
        if (tabs && di >= 0 && di < a.n_deep && a.d_n[di] > 0) {
          const int32_t* st = a.d_start[di];
          const int nt = a.d_n[di];
          if (t == 2 && a.idx2) {
            k0 = a.idx2[(size_t)a.cur.tokens[r * a.cur.ld] * Vt + a.cur.tokens[r * a.cur.ld + 1]];
          } else {
            // the entry that starts at row lo: lower bound over the level's start rows, 64 probes per round (one per lane)
            int l = 0, h = nt;                          // entries < l start before lo, entries >= h start at or behind it
            while (h - l > 64) {
              const long span = (long)h - l;
              const bool less = st[l + (int)((span * (lane + 1)) / 65)] < lo;
              const int cnt = __popcll(__ballot(less));  // probes ascend with the lane: the first cnt of them are below
              const int nl = cnt > 0 ? l + (int)((span * cnt) / 65) + 1 : l;
              const int nh = cnt < 64 ? l + (int)((span * (cnt + 1)) / 65) : h;
              l = nl; h = nh;
            }
            const bool less = l + lane < h && st[l + lane] < lo;
            k0 = l + __popcll(__ballot(less));
          }
          if (k0 >= 0 && (k0 >= nt || st[k0] != lo)) k0 = -1;
          if (k0 >= 0) {
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
          }
        }
        if (k0 < 0) {
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
      }
    }
    if (a.log_softmax) {   // fp32 log_softmax over the real columns (generation.py:453-455), as select_kernel
      const int Vr = a.Vreal > 0 ? a.Vreal : V;
      const float* row = a.logits + (a.shared0 ? (size_t)q * V : r * V);
      float mx = -INFINITY;
      for (int c = lane; c < Vr; c += 64) mx = fmaxf(mx, row[c]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      float sm = 0.f;
      for (int c = lane; c < Vr; c += 64) sm += expf(row[c] - mx);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
      if (lane == 0) { lst[2 * bl] = mx; lst[2 * bl + 1] = logf(sm); }
    }
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
    wave_hist(hist, true, (unsigned)(key >> (64 - RS_BITS)), lane);
  }
  __syncthreads();
  unsigned* gh = a.rs.hist + (size_t)q * 3 * RS_BINS;
  for (int i = tid; i < RS_BINS; i += 256) { const unsigned v = hist[i]; if (v) atomicAdd(&gh[i], v); }
}

// ---- passes 1 and 2: histogram of the next digit inside the threshold bin(s) --------------------------------------------
__global__ __launch_bounds__(256) void rs_hist_kernel(SelectArgs a, int nbq, int pass) {
  __shared__ unsigned hist[RS_BINS];
  __shared__ int sh[8];
  const int q = blockIdx.x / nbq, bb = blockIdx.x - q * nbq;
  if (a.nq_dev && q >= *a.nq_dev) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int B = a.B, V = a.V;
  const int b0 = bb * RS_BPB, b1 = min(B, b0 + RS_BPB);
  for (int i = tid; i < RS_BINS; i += 256) hist[i] = 0u;
  unsigned* gh = a.rs.hist + (size_t)q * 3 * RS_BINS;
  int d0, d1 = 0, k;
  find_digit(gh, B, sh, tid, d0, k);
  if (pass == 2) find_digit(gh + RS_BINS, k, sh, tid, d1, k);
  const unsigned long long want = pass == 1 ? (unsigned long long)d0 : (((unsigned long long)d0 << RS_BITS) | (unsigned long long)d1);
  const int shift_prefix = 64 - RS_BITS * pass, shift_digit = 64 - RS_BITS * (pass + 1);
  const CandCtx x = make_ctx(a, q);
  for (int it = b0 * V + tid; it < b1 * V; it += 256) {
    const int b = it / V;
    const unsigned long long key = cand_key(x, it, x.valid_q[it >> 6], x.score_q[b], b);
    wave_hist(hist, (key >> shift_prefix) == want, (unsigned)(key >> shift_digit) & (RS_BINS - 1), lane);
  }
  __syncthreads();
  unsigned* go = gh + (size_t)pass * RS_BINS;
  for (int i = tid; i < RS_BINS; i += 256) { const unsigned v = hist[i]; if (v) atomicAdd(&go[i], v); }
}

// ---- collect: candidates above the 33-bit threshold prefix, and the ties on it ------------------------------------------
__global__ __launch_bounds__(256) void rs_collect_kernel(SelectArgs a, int nbq) {
  __shared__ int sh[8];
  const int q = blockIdx.x / nbq, bb = blockIdx.x - q * nbq;
  if (a.nq_dev && q >= *a.nq_dev) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int B = a.B, V = a.V;
  const int b0 = bb * RS_BPB, b1 = min(B, b0 + RS_BPB);
  const unsigned* gh = a.rs.hist + (size_t)q * 3 * RS_BINS;
  int d0, d1, d2, k;
  find_digit(gh, B, sh, tid, d0, k);
  find_digit(gh + RS_BINS, k, sh, tid, d1, k);
  find_digit(gh + 2 * RS_BINS, k, sh, tid, d2, k);
  const unsigned long long thr = ((unsigned long long)d0 << (2 * RS_BITS)) | ((unsigned long long)d1 << RS_BITS) | (unsigned long long)d2;
  const CandCtx x = make_ctx(a, q);
  unsigned* cnt = a.rs.cnt + (size_t)q * 4;
  int32_t* win = a.rs.win + (size_t)q * B;
  int32_t* tie = a.rs.tie + (size_t)q * ((size_t)B * V);
  for (int it = b0 * V + tid; it < b1 * V; it += 256) {
    const int b = it / V;
    const unsigned long long p = cand_key(x, it, x.valid_q[it >> 6], x.score_q[b], b) >> (64 - 3 * RS_BITS);
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
  int P2 = 1024;
  // LDS: keys [cap] + items [cap], cap = the launch's sort capacity (>= next power of two of B)
  const int cap = a.rs.sort_cap;
  unsigned long long* ck = reinterpret_cast<unsigned long long*>(smem_raw);
  int* ci = reinterpret_cast<int*>(ck + cap);
  if (n_gt + n_tie > cap) {
    // More ties than the sort buffer holds (degenerate scores: whole codebooks tie). Block-level radix select of the `need`
    // best ties by the composite (low 31 key bits, then descending candidate index): 8 passes of 8 bits over the tie list
    // in global memory; composites are unique, so exactly `need` ties are >= the threshold.
    unsigned long long prefix = 0ull;                  // bits fixed so far, at the top of the composite
    int k = need;
    for (int pass = 0; pass < 8; ++pass) {
      const int shift = 56 - 8 * pass;
      if (tid < 256) h8[tid] = 0u;
      __syncthreads();
      for (int i = tid; i < n_tie; i += 1024) {
        const int it = tie[i];
        const unsigned long long comp = ((cand_key_any(x, it) & 0x7fffffffull) << 32) | (unsigned long long)(0xffffffffu - (unsigned)it);
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
      const unsigned long long comp = ((key & 0x7fffffffull) << 32) | (unsigned long long)(0xffffffffu - (unsigned)it);
      if (comp >= prefix) { const int slot = atomicAdd(&sh[2], 1); ck[n_gt + slot] = key; ci[n_gt + slot] = it; }
    }
    __syncthreads();
    m = need;
  } else {
    for (int i = tid; i < n_tie; i += 1024) { const int it = tie[i]; ck[n_gt + i] = cand_key_any(x, it); ci[n_gt + i] = it; }
  }
  for (int i = tid; i < n_gt; i += 1024) { const int it = win[i]; ck[i] = cand_key_any(x, it); ci[i] = it; }
  const int n = n_gt + m;
  while (P2 < n) P2 <<= 1;
  for (int i = n + tid; i < P2; i += 1024) { ck[i] = 0ull; ci[i] = 0x7fffffff; }
  __syncthreads();
  for (int k = 2; k <= P2; k <<= 1) {
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      for (int p = tid; p < P2 / 2; p += 1024) {
        const int lo_i = 2 * p - (p & (jj - 1));
        const int hi_i = lo_i + jj;
        const bool desc = (lo_i & k) == 0;
        Ent e0{ck[lo_i], ci[lo_i]}, e1{ck[hi_i], ci[hi_i]};
        if (ent_better(e1, e0) == desc) { ck[lo_i] = e1.k; ci[lo_i] = e1.it; ck[hi_i] = e0.k; ci[hi_i] = e0.it; }
      }
      __syncthreads();
    }
  }
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
  unsigned* gh = a.rs.hist + (size_t)q * 3 * RS_BINS;
  for (int i = tid; i < 3 * RS_BINS; i += 1024) gh[i] = 0u;
  if (tid < 4) cnt[tid] = 0u;
}

int next_pow2(int n) { int p = 1024; while (p < n) p <<= 1; return p; }

}  // namespace

// RPR_SELECT_RADIX: 0 = never, 1 = every selection that fits (tests), unset = from 256 beams on
bool select_radix_wanted(int B, int V) {
  if (!select_radix_fits(B, V)) return false;
  const char* e = getenv("RPR_SELECT_RADIX");   // read per call: a test switches it between searches
  if (e) return atoi(e) != 0;
  return B >= 256;
}

bool select_radix_fits(int B, int V) { return V % 64 == 0 && V <= 4096 && B >= 1 && B <= RS_SORT_CAP && (long)B * V < (1L << 31); }

// bytes of RadixWs scratch for Q queries (everything but chi, which has the shape of SelectArgs::lb_scratch)
size_t select_radix_ws_bytes(int Q, int B, int V) {
  const size_t n = (size_t)B * V;
  return (size_t)Q * (3 * RS_BINS * 4 + 16 + n / 8 + (size_t)B * 8 + (size_t)B * 4 + n * 4) + 256;
}

void select_radix_carve(RadixWs& w, void* base, int32_t* chi, int Q, int B, int V) {
  const size_t n = (size_t)B * V;
  unsigned char* p = static_cast<unsigned char*>(base);
  w.hist = reinterpret_cast<unsigned*>(p); p += (size_t)Q * 3 * RS_BINS * 4;
  w.valid = reinterpret_cast<unsigned long long*>(p); p += (size_t)Q * (n / 8);
  w.lstat = reinterpret_cast<float*>(p); p += (size_t)Q * B * 8;
  w.cnt = reinterpret_cast<unsigned*>(p); p += (size_t)Q * 16;
  w.win = reinterpret_cast<int32_t*>(p); p += (size_t)Q * B * 4;
  w.tie = reinterpret_cast<int32_t*>(p);
  w.chi = chi;
  w.sort_cap = 0;
}

hipError_t init_select_radix_attributes() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(rs_finish_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RS_SORT_CAP * 12);
}

// zero the histograms and counters of Q queries (once per search; rs_finish_kernel re-zeroes them after every step)
hipError_t launch_select_radix_reset(const RadixWs& w, int Q, hipStream_t s) {
  hipError_t e = launch_zero_u64(reinterpret_cast<unsigned long long*>(w.hist), (size_t)Q * 3 * RS_BINS / 2, s);
  if (e != hipSuccess) return e;
  return launch_zero_u64(reinterpret_cast<unsigned long long*>(w.cnt), (size_t)Q * 2, s);
}

hipError_t launch_select_radix(const SelectArgs& a_in, hipStream_t s) {
  SelectArgs a = a_in;
  if (!select_radix_fits(a.B, a.V) || !a.rs.hist || !a.rs.chi) return hipErrorInvalidValue;
  const int nbq = (a.B + RS_BPB - 1) / RS_BPB;
  // the sort holds the winners plus the ties on the threshold prefix: twice the beam count covers every ordinary step,
  // more ties than that go through the finish kernel's own radix select first
  a.rs.sort_cap = std::min(RS_SORT_CAP, next_pow2(2 * a.B));
  const dim3 grid((unsigned)(a.Q * nbq)), blk(256);
  hipLaunchKernelGGL(rs_mask_kernel, grid, blk, (size_t)RS_BPB * (a.V / 64) * 8, s, a, nbq);
  hipLaunchKernelGGL(rs_hist_kernel, grid, blk, 0, s, a, nbq, 1);
  hipLaunchKernelGGL(rs_hist_kernel, grid, blk, 0, s, a, nbq, 2);
  hipLaunchKernelGGL(rs_collect_kernel, grid, blk, 0, s, a, nbq);
  hipLaunchKernelGGL(rs_finish_kernel, dim3((unsigned)a.Q), dim3(1024), (size_t)a.rs.sort_cap * 12, s, a);
  return hipGetLastError();
}

}  // namespace rpr
