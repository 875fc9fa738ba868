// Trie-constrained beam selection on the device (gfx950, wave64).
//
// Replaces, per decode step, the reference's host-side pipeline
//   PrefixConstrainLogitProcessorFastSparse.__call__   (tasks/generation.py:666-677: D2H ids -> string
//       keys -> dict -> scipy CSR rows -> float64 mask -> H2D)
//   mask/score combine                                   (generation.py:453-463)
//   torch.topk(2B) + // and %                             (generation.py:484-492)
//   BeamSearchScorer.process (first B of the sorted 2B)   (generation.py:496-503; HF 4.17)
//   input_ids = cat(input_ids[beam_idx], tokens)          (generation.py:511)
// and at the end BeamSearchScorer.finalize                (generation.py:532-540).
//
// Trie representation: the docid code matrix sorted lexicographically ([N, L] uint16). A beam's
// trie node is the half-open row range [lo, hi) sharing its prefix; inside that range column t is
// sorted, so "token c is a child" <=> lower_bound(c) lands on a row whose column t equals c, and
// the child's range is [lower_bound(c), lower_bound(c+1)). A beam that ever took a masked token has
// an empty range, which reproduces the reference's all-zero mask row for unknown prefixes.
//
// Float semantics kept bit-for-bit: candidate = ((double)logit_f32 + (valid ? 0 : -1e9)) + beam_score
// in float64; ties broken by ascending flat index beam*V + token (torch.topk leaves ties
// unspecified); finalize ranks by float64 sum/(L+1) descending with exact ties in reverse slot order
// (stable ascending sort + pop), scores rounded once to float32.
#include <cstdlib>

#include "common.h"

namespace rpr {

__device__ __forceinline__ int lower_bound_col(const uint16_t* __restrict__ codes, int Lc, int col, int lo, int hi,
                                               int c) {
  while (lo < hi) {
    const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
    if ((int)codes[(size_t)mid * Lc + col] < c) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ void init_beams_kernel(BeamState st, int Q, int B, int N) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= Q * B) return;
  st.score[r] = (r % B == 0) ? 0.0 : -1e9;  // generation.py:418-420
  st.lo[r] = 0;
  st.hi[r] = N;
}

hipError_t launch_init_beams(const BeamState& st, int Q, int B, int64_t N, hipStream_t s) {
  const int R = Q * B;
  hipLaunchKernelGGL(init_beams_kernel, dim3((R + 255) / 256), dim3(256), 0, s, st, Q, B, (int)N);
  return hipGetLastError();
}

struct Cand {
  double s;
  int item;
};
__device__ __forceinline__ bool better(const Cand& a, const Cand& b) {
  return a.s > b.s || (a.s == b.s && a.item < b.item);
}
__device__ __forceinline__ Cand wave_best(Cand c) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Cand d;
    d.s = __shfl_xor(c.s, o, 64);
    d.item = __shfl_xor(c.item, o, 64);
    if (better(d, c)) c = d;
  }
  return c;
}

// One block (256 threads) per query. TK = length of the per-thread candidate lists (8 for the rounds of small
// beams, 16 for the sorted path of large ones).
template <int TK>
__global__ __launch_bounds__(256) void select_kernel(SelectArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // (Many beams per query — the evaluation script's --topk=1000 — go through the radix selection, select_radix.hip; this
  // kernel is the path of small beams and, from 32 beams on, the single-block reference the radix path is tested against.)
  const int Bw = a.B, B = a.B, V = a.V, t = a.t, Lc = a.Lc;
  const int Vr = a.Vreal > 0 ? a.Vreal : V;   // real vocab; columns Vr..V-1 of a logits row are padding (zero logits)
  const int q = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (a.nq_dev && q >= *a.nq_dev) return;   // compacted stage: block-uniform
  const int items = B * V, words = items >> 6;
  // LDS carve (all 8-byte aligned)
  double* bscore = reinterpret_cast<double*>(smem_raw);                  // [B]
  double* wscore = bscore + B;                                           // [Bw]
  double* red_s = wscore + Bw;                                           // [4]
  unsigned long long* valid = reinterpret_cast<unsigned long long*>(red_s + 4);  // [words]
  unsigned long long* taken = valid + words;                             // [words]
  int* blo = reinterpret_cast<int*>(taken + words);                      // [B]
  int* bhi = blo + B;                                                    // [B]
  int* widx = bhi + B;                                                   // [Bw]
  int* red_i = widx + Bw;                                                // [8] (+ 4 floats lmax_w)
  float* lmax_w = reinterpret_cast<float*>(red_i + 8);                   // [4]   per-wave logit maxima (compact path)
  float* lmax = lmax_w + 4;                                              // [B]   (red_i[4] = rescan flag)
  float* lsum = lmax + B;                                                // [B] log(sum exp)
  float* slog = lsum + B;                                                // [B*V] logits of the query (a.lds_logits)

  const int r0 = q * B;                     // first beam row of this query
  auto stamp = [&](int k) { if (a.clk && blockIdx.x == 0 && tid == 0) a.clk[k] = wall_clock64(); };
  stamp(0);
  for (int b = tid; b < B; b += 256) {
    bscore[b] = a.cur.score[r0 + b];
    blo[b] = a.cur.lo[r0 + b];
    bhi[b] = a.cur.hi[r0 + b];
  }
  for (int w = tid; w < words; w += 256) { taken[w] = 0ull; valid[w] = 0ull; }
  int* any_wide = red_i + 6;             // LDS flag: some beam's range is wider than NARROW rows
  if (tid == 0) *any_wide = 0;
  __syncthreads();
  for (int b = tid; b < B; b += 256)
    if (bhi[b] - blo[b] > 32) *any_wide = 1;   // benign race: every writer stores 1
  __syncthreads();

  // ---- phase A: child mask of every beam (64 consecutive tokens of one beam per wave) ----
  const float* lg_q = a.logits + (a.shared0 ? (size_t)q * V : (size_t)r0 * V);   // shared0: one row per query
  int* lb_q = a.lb_scratch + (size_t)r0 * V;
  // Wide ranges (the first trie levels): one binary search per (beam, token) over the sorted column, the bitmap
  // word of 64 tokens comes from a ballot. Narrow ranges (<= NARROW rows — every beam from depth ~3 on, one doc
  // each): the children are enumerated from the rows themselves, <= NARROW reads per beam instead of V searches
  // (at B = 1000 the V searches per beam were ~1000 dependent global reads per thread and step).
  constexpr int NARROW = 32;
  // step 0: every beam starts at the root range (init_beams_kernel), so only beam 0 is searched and copied below
  const int search_items = (t == 0) ? V : items;
  // from depth ~3 on every range is narrow (one doc each): the (beam, token) loop has nothing to search then and, with
  // the logits left in global memory (large beams), nothing to stage either — at B = 1000 it cost 190 us per step
  // step 1 with the trie's child arrays: the first token of every beam once, into LDS (widx is free until the selection),
  // so that the table lookups below depend on no other global load
  const bool tab1 = t == 1 && a.lvl1 != nullptr && t < Lc;      // block-uniform
  const bool tab0 = t == 0 && a.lvl0 != nullptr;
  if (tab1) {
    for (int b = tid; b < B; b += 256) widx[b] = a.cur.tokens[(size_t)(r0 + b) * a.cur.ld];
    __syncthreads();
  }
  if (*any_wide || a.lds_logits)
  for (int item = tid; item < items; item += 4 * 256) {
    // four (beam, token) pairs per thread advance their binary searches in lockstep: the four probes of a step
    // are independent loads (one search at a time was one dependent L2/HBM latency per probe)
    int lo4[4], hi4[4], end4[4], c4[4], tab4[4], nxt4[4];
    bool act[4], use_tab[4];
    const int32_t* tptr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int it = item + u * 256;
      const bool in = it < items;                      // wave-uniform (V % 64 == 0)
      const int b = in ? it / V : 0, c = it - b * V;
      const int lo = blo[b], hi = bhi[b];
      if (in && a.lds_logits) slog[it] = lg_q[a.shared0 ? c : it];
      const bool narrow = hi - lo <= NARROW;           // wave-uniform: a wave covers 64 tokens of one beam
      act[u] = in && !narrow && it < search_items;
      lo4[u] = lo; hi4[u] = (act[u] && t < Lc) ? hi : lo; end4[u] = hi; c4[u] = c;
      tab4[u] = -1; nxt4[u] = 0;
      // levels 0 / 1: the lower bound of (prefix, c) and of its successor straight from the trie's child arrays (the range of
      // this beam is [lvl0[c0], lvl0[c0 + 1]) at step 1: its rows share code 0 = c0). Address by selects, loads unconditional
      // (a pair without a table entry reads the scratch array): inside a branch per pair the compiler retired every load before
      // the next pair's was issued — 248 serial latencies per thread, 838 us of phase A at beam 1000 with one query.
      use_tab[u] = act[u] && t < Lc && c < a.lvl_V && (tab0 || tab1);
      tptr[u] = !use_tab[u] ? a.lb_scratch : tab0 ? a.lvl0 + c : a.lvl1 + (size_t)widx[b] * a.lvl_V + c;
    }
    {
      int e0[4], e1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { e0[u] = tptr[u][0]; e1[u] = tptr[u][1]; }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (use_tab[u]) { tab4[u] = e0[u]; nxt4[u] = e1[u]; hi4[u] = lo4[u]; }
    }
    for (;;) {
      int v4[4];
      bool go = false;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v4[u] = 0;
        if (lo4[u] < hi4[u]) {
          go = true;
          v4[u] = a.codes[(size_t)(int)(((unsigned)lo4[u] + (unsigned)hi4[u]) >> 1) * Lc + t];
        }
      }
      if (!go) break;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (lo4[u] < hi4[u]) {
          const int mid = (int)(((unsigned)lo4[u] + (unsigned)hi4[u]) >> 1);
          if (v4[u] < c4[u]) lo4[u] = mid + 1; else hi4[u] = mid;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int it = item + u * 256;
      if (!act[u]) continue;                           // wave-uniform
      const bool tab = tab4[u] >= 0;                   // (per lane: tokens >= the trie's vocab size have no table entry)
      const int l = tab ? tab4[u] : lo4[u];
      const bool ok = tab ? nxt4[u] > l : (t < Lc && l < end4[u] && (int)a.codes[(size_t)l * Lc + t] == c4[u]);
      lb_q[it] = l;
      const unsigned long long m = __ballot(ok);
      if (lane == 0) valid[it >> 6] = m;
    }
  }
  stamp(1);
  if (t == 0 && B > 1) {   // replicate beam 0's row (same root range for every beam)
    __syncthreads();
    if (bhi[0] - blo[0] > NARROW) {
      for (int it = V + tid; it < items; it += 256) {
        const int c = it % V;
        lb_q[it] = lb_q[c];
        if ((it & 63) == 0) valid[it >> 6] = valid[c >> 6];
      }
    }
  }
  __syncthreads();
  if (t < Lc) {
    for (int idx = tid; idx < B * NARROW; idx += 256) {
      const int b = idx / NARROW, k = idx - b * NARROW;
      const int lo = blo[b], hi = bhi[b];
      const int r = lo + k;
      if (hi - lo > NARROW || r >= hi) continue;
      const int c = a.codes[(size_t)r * Lc + t];
      const int item = b * V + c;
      atomicOr(&valid[item >> 6], 1ull << (item & 63));
      if (k == 0 || (int)a.codes[(size_t)(r - 1) * Lc + t] != c) lb_q[item] = r;   // first row of this child
    }
  }
  if (a.log_softmax) {  // fp32 log_softmax over V (generation.py:453-455): (x - max) - log(sum exp(x - max))
    for (int b = wave; b < B; b += 4) {
      const float* row = a.shared0 ? lg_q : lg_q + (size_t)b * V;
      float mx = -INFINITY;
      for (int c = lane; c < Vr; c += 64) mx = fmaxf(mx, row[c]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      float sm = 0.f;
      for (int c = lane; c < Vr; c += 64) sm += expf(row[c] - mx);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
      if (lane == 0) { lmax[b] = mx; lsum[b] = logf(sm); }
    }
  }
  __syncthreads();
  stamp(2);
  if (a.tap_valid)   // debug tap: the child bitmap the selection below works from (tests/test_gpu_parity.py)
    for (int w = tid; w < words; w += 256) a.tap_valid[(size_t)q * words + w] = valid[w];

  auto raw_logit = [&](int item) -> float {
    if (a.lds_logits) return slog[item];
    return lg_q[a.shared0 ? item % V : item];
  };
  auto cand_score = [&](int item, float lg) -> double {
    const int b = item / V;
    if (Vr != V && item - b * V >= Vr) return -INFINITY;   // padding column: behind every candidate of the reference
    if (a.log_softmax) lg = (lg - lmax[b]) - lsum[b];
    const bool ok = (valid[item >> 6] >> (item & 63)) & 1ull;
    return ((double)lg + (ok ? 0.0 : -1e9)) + bscore[b];
  };
  // best untaken candidate among items first, first + stride, ... The logits of four candidates are requested
  // before any of them is looked at: at B = 1000 they live in global memory (L2), and one dependent read per
  // candidate made every arg-max round cost ~16 L2 latencies.
  // Ownership: thread o owns one candidate of every stripe of 256 consecutive items, rotated by the stripe index:
  // item_k(o) = ((o + k) & 255) + 256 k. With V = 256 a stripe is a beam and the thread meets a different token in every
  // beam — logits of the beams of a query are strongly correlated, so "thread = token" put most of the top-B on a few
  // threads (list overflow, rescans), the diagonal spreads them.
  auto owned = [&](int o, int k) -> int { return ((o + k) & 255) + (k << 8); };
  auto owner_of = [&](int item) -> int { return ((item & 255) - (item >> 8)) & 255; };
  const int stripes = (items + 255) >> 8;
  // best untaken candidate of thread o, its stripes split over the 64 lanes of a wave
  auto scan_owner = [&](int o) -> Cand {
    Cand best; best.s = -INFINITY; best.item = 0x7fffffff;
    for (int k0 = lane; k0 < stripes; k0 += 4 * 64) {
      float lg[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int it = owned(o, k0 + u * 64);
        lg[u] = (k0 + u * 64 < stripes && it < items) ? raw_logit(it) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int it = owned(o, k0 + u * 64);
        if (k0 + u * 64 >= stripes || it >= items) continue;
        if ((taken[it >> 6] >> (it & 63)) & 1ull) continue;
        Cand c; c.s = cand_score(it, lg[u]); c.item = it;
        if (better(c, best)) best = c;
      }
    }
    return best;
  };

  // ---- phase B/C: B rounds of block-wide argmax ----
  // Every thread owns the candidates tid, tid + 256, ... and keeps its TK best in a sorted register list, built in
  // one pass.
  //  * B <= 256: B rounds of block-wide arg-max over the list heads; the winner's thread pops its list (a thread
  //    that has won TK times and owns more candidates has its wave rescan them — never happens for B*V/256 <= TK).
  //  * B > 256 (a.sort_lds): the 256 lists are written to LDS and sorted with one bitonic network (4096 entries);
  //    the first B entries are the winners in rank order. This is exact unless some thread owns more than TK of
  //    the top B (its last list entry is inside the top B and it has more candidates) — then the rounds run
  //    instead. B = 1000 spent 1000 sequential rounds (2.8 us each) per step before.
  // ---- large beams, usual case: the valid candidates alone ----
  // From depth ~3 on every beam has a handful of children: ~B valid candidates among B*V. They are compacted from the
  // bitmap into the sort buffer and sorted; the result is exact iff the B-th of them beats every INVALID candidate, whose
  // scores are bounded by ((double)max logit + -1e9) + max beam score (the additions are monotone) — checked, else the
  // general path below runs. (B = 1000: the scan of all 256 000 candidates was 0.85 of the kernel's 1.3 ms per step.)
  bool use_compact = false;
  if (a.sort_lds) {
    constexpr int NS = 256 * TK;
    double* cs = reinterpret_cast<double*>(smem_raw + a.sort_off);   // [NS]
    int* ci = reinterpret_cast<int*>(cs + NS);                       // [NS]
    int* cnt_s = red_i + 7;
    int cnt = 0;
    for (int w = tid; w < words; w += 256) cnt += __popcll(valid[w]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) red_i[wave] = cnt;
    if (tid == 0) *cnt_s = 0;
    __syncthreads();
    const int nv = red_i[0] + red_i[1] + red_i[2] + red_i[3];
    __syncthreads();                                                 // red_i is reused below
    if (nv >= Bw && nv <= NS) {                                      // block-uniform
      float lm = -INFINITY;
      if (a.log_softmax) lm = 0.f;                                   // log-probabilities are <= 0
      else {
        const int n4 = (a.shared0 ? V : items) >> 2;
        const float4* l4 = reinterpret_cast<const float4*>(lg_q);
        for (int i = tid; i < n4; i += 256) { const float4 v = l4[i]; lm = fmaxf(lm, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w))); }
      }
      double bm = -INFINITY;
      for (int b = tid; b < B; b += 256) bm = fmax(bm, bscore[b]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { lm = fmaxf(lm, __shfl_xor(lm, o, 64)); bm = fmax(bm, __shfl_xor(bm, o, 64)); }
      if (lane == 0) { red_s[wave] = bm; lmax_w[wave] = lm; }
      for (int w = tid; w < words; w += 256) {
        unsigned long long bits = valid[w];
        while (bits) {
          const int bit = __ffsll((long long)bits) - 1;
          bits &= bits - 1;
          const int item = (w << 6) + bit;
          const int slot = atomicAdd(cnt_s, 1);
          cs[slot] = cand_score(item, raw_logit(item));
          ci[slot] = item;
        }
      }
      int P2 = 256;
      while (P2 < nv) P2 <<= 1;
      for (int i = nv + tid; i < P2; i += 256) { cs[i] = -INFINITY; ci[i] = 0x7fffffff; }
      __syncthreads();
      for (int k = 2; k <= P2; k <<= 1) {
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
          for (int p = tid; p < P2 / 2; p += 256) {
            const int lo_i = 2 * p - (p & (jj - 1));
            const int hi_i = lo_i + jj;
            const bool desc = (lo_i & k) == 0;
            Cand x; x.s = cs[lo_i]; x.item = ci[lo_i];
            Cand y; y.s = cs[hi_i]; y.item = ci[hi_i];
            if (better(y, x) == desc) { cs[lo_i] = y.s; ci[lo_i] = y.item; cs[hi_i] = x.s; ci[hi_i] = x.item; }
          }
          __syncthreads();
        }
      }
      const double bmax = fmax(fmax(red_s[0], red_s[1]), fmax(red_s[2], red_s[3]));
      const float lmaxq = fmaxf(fmaxf(lmax_w[0], lmax_w[1]), fmaxf(lmax_w[2], lmax_w[3]));
      const double bound = ((double)lmaxq + -1e9) + bmax;
      if (cs[Bw - 1] > bound) {                                      // block-uniform (LDS values)
        for (int j = tid; j < Bw; j += 256) { wscore[j] = cs[j]; widx[j] = ci[j]; }
        use_compact = true;
      }
      __syncthreads();
    }
  }

  double ts[TK];
  int ti[TK];
#pragma unroll
  for (int i = 0; i < TK; ++i) { ts[i] = -INFINITY; ti[i] = 0x7fffffff; }
  int own = 0;   // candidates owned by this thread
  for (int k0 = 0; k0 < stripes && !use_compact; k0 += 4) {
    float lg[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int it = owned(tid, k0 + u);
      lg[u] = (k0 + u < stripes && it < items) ? raw_logit(it) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int it = owned(tid, k0 + u);
      if (k0 + u >= stripes || it >= items) continue;
      ++own;
      Cand c; c.s = cand_score(it, lg[u]); c.item = it;
      Cand last; last.s = ts[TK - 1]; last.item = ti[TK - 1];
      if (!better(c, last)) continue;
      ts[TK - 1] = c.s; ti[TK - 1] = c.item;
#pragma unroll
      for (int i = TK - 1; i > 0; --i) {   // bubble up (static register indices)
        Cand lo_; lo_.s = ts[i]; lo_.item = ti[i];
        Cand hi_; hi_.s = ts[i - 1]; hi_.item = ti[i - 1];
        if (better(lo_, hi_)) { ts[i] = hi_.s; ti[i] = hi_.item; ts[i - 1] = lo_.s; ti[i - 1] = lo_.item; }
      }
    }
  }
  stamp(3);
  int left = own < TK ? own : TK;        // entries of the list not yet consumed
  const bool more = own > TK;            // candidates beyond the list exist
  int* resc = red_i + 4;                 // LDS flag: the winner's wave must rescan for it
  int* overflow = red_i + 5;             // LDS flag: the sorted union of the lists may miss a top-B candidate
  if (tid == 0) { *resc = 0; *overflow = 0; }
  bool need_rounds = !use_compact;
  if (a.sort_lds && !use_compact) {
    constexpr int NS = 256 * TK;
    double* cs = reinterpret_cast<double*>(smem_raw + a.sort_off);   // [NS]
    int* ci = reinterpret_cast<int*>(cs + NS);                       // [NS]
#pragma unroll
    for (int i = 0; i < TK; ++i) { cs[tid * TK + i] = ts[i]; ci[tid * TK + i] = ti[i]; }
    __syncthreads();
    for (int k = 2; k <= NS; k <<= 1) {
      for (int jj = k >> 1; jj > 0; jj >>= 1) {
        for (int p = tid; p < NS / 2; p += 256) {
          const int lo_i = 2 * p - (p & (jj - 1));        // the pair (lo_i, lo_i + jj)
          const int hi_i = lo_i + jj;
          const bool desc = (lo_i & k) == 0;               // best-first in the blocks that end up in front
          Cand x; x.s = cs[lo_i]; x.item = ci[lo_i];
          Cand y; y.s = cs[hi_i]; y.item = ci[hi_i];
          if (better(y, x) == desc) { cs[lo_i] = y.s; ci[lo_i] = y.item; cs[hi_i] = x.s; ci[hi_i] = x.item; }
        }
        __syncthreads();
      }
    }
    Cand kth; kth.s = cs[Bw - 1]; kth.item = ci[Bw - 1];
    Cand mylast; mylast.s = ts[TK - 1]; mylast.item = ti[TK - 1];
    if (more && better(mylast, kth)) *overflow = 1;      // benign race: every writer stores 1
    __syncthreads();
    if (*overflow == 0) {
      for (int j = tid; j < Bw; j += 256) { wscore[j] = cs[j]; widx[j] = ci[j]; }
      need_rounds = false;
    }
  }
  stamp(4);
  if (a.clk && blockIdx.x == 0 && tid == 0) a.clk[7] = need_rounds ? 1 : 0;
  if (need_rounds) {   // block-uniform
  Cand mine; mine.s = ts[0]; mine.item = ti[0];
  for (int j = 0; j < Bw; ++j) {
    const Cand wb = wave_best(mine);
    if (lane == 0) { red_s[wave] = wb.s; red_i[wave] = wb.item; }
    __syncthreads();
    Cand win; win.s = red_s[0]; win.item = red_i[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      Cand c; c.s = red_s[w]; c.item = red_i[w];
      if (better(c, win)) win = c;
    }
    const bool any = win.item != 0x7fffffff;
    const int owner_tid = owner_of(win.item);
    const bool owner = any && owner_tid == tid;
    if (tid == 0) { wscore[j] = win.s; widx[j] = win.item; }
    if (owner) {
      taken[win.item >> 6] |= 1ull << (win.item & 63);  // single writer per round (one winner), fenced by the barriers
      if (left > 0) {                     // pop the list head
#pragma unroll
        for (int i = 0; i < TK - 1; ++i) { ts[i] = ts[i + 1]; ti[i] = ti[i + 1]; }
        ts[TK - 1] = -INFINITY; ti[TK - 1] = 0x7fffffff;
        --left;
      }
      mine.s = ts[0]; mine.item = ti[0];
      *resc = (left == 0 && more) ? 1 : 0;
    }
    __syncthreads();
    if (any && *resc && wave == (owner_tid >> 6)) {   // list exhausted: the wave rescans that thread's candidates
      Cand best = scan_owner(owner_tid);
      best = wave_best(best);
      if (owner) mine = best;
    }
  }
  }
  __syncthreads();
  stamp(5);

  // ---- phase D: write the next beam state (new slot j <- winner j) ----
  const int ld = a.cur.ld;
  for (int j = tid; j < Bw; j += 256) {
    const int item = widx[j];
    const int b = item / V, c = item - b * V;
    const bool ok = (valid[item >> 6] >> (item & 63)) & 1ull;
    int nlo = 0, nhi = 0;
    if (ok) {
      nlo = lb_q[item];
      if (bhi[b] - blo[b] <= 32) {   // narrow range (phase A enumerated it): the child ends where the token changes
        nhi = nlo + 1;
        while (nhi < bhi[b] && (int)a.codes[(size_t)nhi * Lc + t] == c) ++nhi;
      } else {
        nhi = (c + 1 < V) ? lb_q[item + 1] : bhi[b];
      }
    }
    const int r = r0 + j;
    a.nxt.score[r] = wscore[j];
    a.nxt.lo[r] = nlo;
    a.nxt.hi[r] = nhi;
    a.nxt.tokens[(size_t)r * ld + t] = (uint16_t)c;
    a.nxt.anc[(size_t)r * ld + t] = (uint16_t)(a.shared0 ? 0 : b);   // slot holding this position's K/V
    if (a.tap_scores) a.tap_scores[r] = wscore[j];
    if (a.tap_tokens) a.tap_tokens[r] = c;
    if (a.tap_parent) a.tap_parent[r] = b;
  }
  for (int i = tid; i < B * t; i += 256) {
    const int j = i / t, p = i - j * t;
    const int b = widx[j] / V;
    a.nxt.tokens[(size_t)(r0 + j) * ld + p] = a.cur.tokens[(size_t)(r0 + b) * ld + p];
    a.nxt.anc[(size_t)(r0 + j) * ld + p] = a.cur.anc[(size_t)(r0 + b) * ld + p];
  }
  stamp(6);
}

constexpr int SEL_TK_HOST = 16;   // list length of the sorted path
// Bs = beams a block selects among, Bw = winners it emits (== Bs unless grouped)
static size_t select_smem(int Bs, int Bw, int V) {
  const size_t words = (size_t)Bs * V / 64;
  return ((size_t)Bs + Bw + 4) * sizeof(double) + 2 * words * sizeof(unsigned long long) +
         (2 * (size_t)Bs + Bw + 12) * sizeof(int) + 2 * (size_t)Bs * sizeof(float) + 16;
}

bool select_fits(int B, int V) { return V % 64 == 0 && select_smem(B, B, V) <= 160 * 1024; }

hipError_t init_beam_kernel_attributes() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(select_kernel<8>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(select_kernel<SEL_TK_HOST>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          160 * 1024);
  if (e != hipSuccess) return e;
  return init_select_radix_attributes();
}

hipError_t launch_select(const SelectArgs& a_in, hipStream_t s) {
  SelectArgs a = a_in;
  if (a.V % 64 != 0) return hipErrorInvalidValue;
  if (a.rs.hist && select_radix_wanted(a.B, a.V)) return launch_select_radix(a, s);   // many beams: select_radix.hip
  size_t smem = select_smem(a.B, a.B, a.V);
  if (smem > 160 * 1024) return hipErrorInvalidValue;
  smem = (smem + 15) & ~(size_t)15;
  const size_t sort_bytes = (size_t)256 * SEL_TK_HOST * (sizeof(double) + sizeof(int));
  // measured: B = 100 -> rounds 8.8 ms per search vs sort 10.9 ms; B = 1000 -> rounds 87 ms vs sort 62 ms
  a.sort_lds = (a.B > 256 && a.B <= 256 * SEL_TK_HOST && smem + sort_bytes <= 160 * 1024) ? 1 : 0;
  a.sort_off = (int)smem;
  const size_t logits_bytes = (size_t)a.B * a.V * sizeof(float);
  // the logits strip sits right behind the fixed carve (slog = lsum + B in the kernel); the sort buffer follows it
  a.lds_logits = (smem + logits_bytes + (a.sort_lds ? sort_bytes : 0) <= 160 * 1024) ? 1 : 0;
  if (a.lds_logits) { smem = (smem + logits_bytes + 15) & ~(size_t)15; a.sort_off = (int)smem; }
  if (a.sort_lds) smem += sort_bytes;
  if (a.sort_lds) hipLaunchKernelGGL(select_kernel<SEL_TK_HOST>, dim3(a.Q), dim3(256), smem, s, a);
  else hipLaunchKernelGGL(select_kernel<8>, dim3(a.Q), dim3(256), smem, s, a);
  return hipGetLastError();
}

// One block per query: rank slots by float64 score/(L+1) descending, exact ties in reverse slot order.
__global__ __launch_bounds__(256) void finalize_kernel(FinalizeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* sc = reinterpret_cast<double*>(smem_raw);  // [B]
  const int B = a.B, L = a.L, q = blockIdx.x, tid = threadIdx.x, r0 = q * B;
  if (a.nq_dev && q >= *a.nq_dev) return;
  const size_t o0 = (size_t)(a.qmap ? a.qmap[q] : q) * B;   // output rows of this query
  for (int j = tid; j < B; j += 256) sc[j] = a.st.score[r0 + j] / (double)(L + 1);
  __syncthreads();
  int* rank = reinterpret_cast<int*>(sc + B);  // [B]
  for (int j = tid; j < B; j += 256) {
    const double s = sc[j];
    int rk = 0;
    for (int i = 0; i < B; ++i) rk += (sc[i] > s) || (sc[i] == s && i > j);
    rank[j] = rk;
    const size_t o = o0 + rk;
    a.out_scores[o] = (float)s;
    a.out_lo[o] = a.st.lo[r0 + j];
    a.out_hi[o] = a.st.hi[r0 + j];
  }
  __syncthreads();
  for (int i = tid; i < B * L; i += 256) {
    const int j = i / L, p = i - j * L;
    a.out_tokens[(o0 + rank[j]) * L + p] = (int32_t)a.st.tokens[(size_t)(r0 + j) * a.st.ld + p];
  }
}

hipError_t launch_finalize(const FinalizeArgs& a, hipStream_t s) {
  const size_t smem = (size_t)a.B * (sizeof(double) + sizeof(int)) + 16;
  hipLaunchKernelGGL(finalize_kernel, dim3(a.Q), dim3(256), smem, s, a);
  return hipGetLastError();
}

// Processor-only entry (rpr_trie_mask): one block per prefix row.
__global__ __launch_bounds__(256) void prefix_mask_kernel(const uint16_t* __restrict__ codes, int Lc, int N,
                                                           const int32_t* __restrict__ prefix, int R, int T, int V,
                                                           uint8_t* __restrict__ out_mask) {
  __shared__ int s_lo, s_hi;
  const int r = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    int lo = 0, hi = N;
    for (int p = 1; p < T && lo < hi; ++p) {
      const int c = prefix[(size_t)r * T + p];
      if (c < 0 || c >= V || p - 1 >= Lc) { hi = lo; break; }
      const int l = lower_bound_col(codes, Lc, p - 1, lo, hi, c);
      const int h = lower_bound_col(codes, Lc, p - 1, l, hi, c + 1);
      lo = l; hi = h;
    }
    s_lo = lo; s_hi = hi;
  }
  __syncthreads();
  const int lo = s_lo, hi = s_hi, col = T - 1;
  for (int c = tid; c < V; c += 256) {
    bool ok = false;
    if (col < Lc && lo < hi) {
      const int l = lower_bound_col(codes, Lc, col, lo, hi, c);
      ok = (l < hi) && ((int)codes[(size_t)l * Lc + col] == c);
    }
    out_mask[(size_t)r * V + c] = ok ? 1 : 0;
  }
}

hipError_t launch_prefix_mask(const uint16_t* codes, int Lc, int64_t N, const int32_t* prefix, int R, int T,
                              int V, uint8_t* out_mask, hipStream_t s) {
  if (R <= 0) return hipSuccess;
  hipLaunchKernelGGL(prefix_mask_kernel, dim3(R), dim3(256), 0, s, codes, Lc, (int)N, prefix, R, T, V, out_mask);
  return hipGetLastError();
}

}  // namespace rpr
