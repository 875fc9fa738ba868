// Host side of the device trie: sort the docid code matrix, keep the permutation, (de)serialise.
//
// Replaces the reference's per-level dict-of-strings build (t5_pretrainer/evaluate.py:410-424,
// aq_preprocess/build_list_smtid_to_nextids.py:21-41), its pickle cache (evaluate.py:404-408,
// 428-432) and the smtid -> docids dict (evaluate.py:439-446): after sorting, every trie node and
// every smtid is a contiguous row range, and the docids of a range are perm[lo..hi).
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <string>
#include <thread>
#include <cstring>
#include <numeric>
#include <vector>

#include "trie.h"

namespace rpr {

struct SortKey {
  uint64_t key;  // first four codes, 16 bits each (most significant first)
  int32_t idx;
};

// Sort = bucket by the first code (counting sort, stable), then sort every bucket on its own thread: the trie's
// first level splits 8.8 M MS MARCO docids into V = 256 independent ranges (RQ codes are roughly balanced there;
// a skewed first level only costs parallelism, not correctness). 8 841 823 x 32 codes: 4.0 s on one thread,
// see DESIGN.md for the threaded figure.
int sort_codes(const uint16_t* codes, int64_t N, int L, std::vector<uint16_t>& sorted, std::vector<int64_t>& perm) {
  std::vector<SortKey> keys((size_t)N);
  const int pk = L < 4 ? L : 4;
  // bucket boundaries by first code
  std::vector<int64_t> start(65537, 0);
  for (int64_t i = 0; i < N; ++i) start[(size_t)codes[i * L] + 1]++;
  for (size_t v = 0; v < 65536; ++v) start[v + 1] += start[v];
  {
    std::vector<int64_t> fill(start.begin(), start.end() - 1);
    for (int64_t i = 0; i < N; ++i) {   // stable scatter: docid order is kept inside a bucket
      uint64_t k = 0;
      for (int l = 0; l < 4; ++l) k = (k << 16) | (l < pk ? codes[i * L + l] : 0);
      keys[(size_t)fill[codes[i * L]]++] = {k, (int32_t)i};
    }
  }
  auto cmp = [codes, L, pk](const SortKey& a, const SortKey& b) {
    if (a.key != b.key) return a.key < b.key;
    const uint16_t* ra = codes + (int64_t)a.idx * L;
    const uint16_t* rb = codes + (int64_t)b.idx * L;
    for (int l = pk; l < L; ++l)
      if (ra[l] != rb[l]) return ra[l] < rb[l];
    return a.idx < b.idx;  // stable: equal smtids keep docid order (evaluate.py:443-446 appends in file order)
  };
  unsigned nthreads = std::thread::hardware_concurrency();
  if (const char* e = std::getenv("RPR_TRIE_THREADS")) nthreads = (unsigned)std::atoi(e);
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 32) nthreads = 32;
  if (N < (1 << 16)) nthreads = 1;
  std::atomic<int> next{0};
  auto worker = [&] {
    for (;;) {
      const int v = next.fetch_add(1);
      if (v >= 65536) break;
      if (start[(size_t)v + 1] - start[(size_t)v] > 1)
        std::sort(keys.begin() + start[(size_t)v], keys.begin() + start[(size_t)v + 1], cmp);
    }
  };
  if (nthreads == 1) {
    worker();
  } else {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nthreads; ++t) pool.emplace_back(worker);
    for (auto& th : pool) th.join();
  }
  sorted.resize((size_t)N * L);
  perm.resize((size_t)N);
  auto gather = [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i) {
      perm[(size_t)i] = keys[(size_t)i].idx;
      std::memcpy(&sorted[(size_t)i * L], codes + (int64_t)keys[(size_t)i].idx * L, sizeof(uint16_t) * L);
    }
  };
  if (nthreads == 1) {
    gather(0, N);
  } else {
    std::vector<std::thread> pool;
    const int64_t chunk = (N + nthreads - 1) / nthreads;
    for (unsigned t = 0; t < nthreads; ++t) {
      const int64_t lo = (int64_t)t * chunk, hi = lo + chunk < N ? lo + chunk : N;
      if (lo < hi) pool.emplace_back(gather, lo, hi);
    }
    for (auto& th : pool) th.join();
  }
  return 0;
}

// ---- forced-tail statistics ----------------------------------------------------------------------------------------------
// For a search of length L over the sorted matrix: frac[t] (t = 0..L) = the fraction of the trie nodes at depth t
// (distinct t-prefixes) under which exactly ONE distinct L-token sequence remains. A beam standing on such a node has
// a single valid child at every remaining step, so its tokens are known and only its scores are missing; a query
// whose beams all stand on such nodes is "forced" (api.hip: forced-tail evaluation). One pass over adjacent rows:
// c_i = min(lcp(row i-1, row i), L); the rows between two boundaries with c < L are one distinct sequence ("run");
// with a / b the c of its left / right boundary (-1 at the ends of the matrix) the run opens a new node at every
// depth t > a and is alone in its node at every depth t > max(a, b).
void trie_single_frac(const uint16_t* sorted, int64_t N, int Lc, int L, std::vector<double>& frac) {
  std::vector<int64_t> nodes((size_t)L + 2, 0), single((size_t)L + 2, 0);   // histograms over a + 1 and max(a, b) + 1
  int a = -1;
  for (int64_t i = 1; i <= N; ++i) {
    int c = -1;                                  // right boundary of the run that ends at row i - 1
    if (i < N) {
      const uint16_t* p = sorted + (size_t)(i - 1) * Lc;
      const uint16_t* q = p + Lc;
      c = 0;
      while (c < L && p[c] == q[c]) ++c;
      if (c == L) continue;                      // same sequence: the run goes on
    }
    nodes[(size_t)(a + 1)] += 1;
    single[(size_t)((a > c ? a : c) + 1)] += 1;
    a = c;
  }
  frac.assign((size_t)L + 1, 0.0);
  int64_t n = 0, s = 0;
  for (int t = 0; t <= L; ++t) {                 // nodes at depth t: runs with a < t; single: runs with max(a, b) < t
    n += nodes[(size_t)t];
    s += single[(size_t)t];
    frac[(size_t)t] = n ? (double)s / (double)n : 0.0;
  }
}

// ---- child arrays ------------------------------------------------------------------------------------------------------
// One pass computes lcp[r] = common prefix length of rows r - 1 and r (capped at 255; lcp[0] = 0): row r opens a new depth-t
// node iff r == 0 or lcp[r] < t. Level t lists the rows that open a depth-(t + 1) node inside a depth-t node of more than
// `narrow` rows. Wide nodes nest, so the first level without any ends the build.
void build_child_levels(const uint16_t* sorted, int64_t N, int Lc, int V, int narrow, int max_levels, int64_t max_entries,
                        ChildLevels& out) {
  out = ChildLevels{};
  if (N <= 0 || Lc < 1 || V < 1) return;
  out.lvl0.assign((size_t)V + 1, (int32_t)N);
  {
    int next = 0;                                   // tokens < next have their lower bound
    for (int64_t r = 0; r < N; ++r) {
      const int v = sorted[(size_t)r * Lc];
      while (next <= v) out.lvl0[(size_t)next++] = (int32_t)r;
    }
  }
  if (Lc < 2 || V > 1024) return;
  out.lvl1.assign((size_t)V * V + 1, (int32_t)N);
  {
    int64_t next = 0;
    for (int64_t r = 0; r < N; ++r) {
      const int64_t v = (int64_t)sorted[(size_t)r * Lc] * V + sorted[(size_t)r * Lc + 1];
      while (next <= v) out.lvl1[(size_t)next++] = (int32_t)r;
    }
  }
  if (Lc < 3 || max_levels < 1) return;
  std::vector<uint8_t> lcp((size_t)N, 0);
  for (int64_t r = 1; r < N; ++r) {
    const uint16_t* p = sorted + (size_t)(r - 1) * Lc;
    const uint16_t* q = p + Lc;
    int c = 0;
    while (c < Lc && c < 255 && p[c] == q[c]) ++c;
    lcp[(size_t)r] = (uint8_t)c;
  }
  out.idx2.assign((size_t)V * V, -1);
  int64_t total = 0;
  for (int t = 2; t < Lc && t < 255 && (int)out.deep.size() < max_levels; ++t) {
    ChildLevels::Level lv;
    int64_t a = 0;                                  // first row of the current depth-t node
    for (int64_t r = 1; r <= N; ++r) {
      if (r < N && lcp[(size_t)r] >= t) continue;   // same depth-t node
      if (r - a > narrow) {
        if (t == 2) out.idx2[(size_t)sorted[(size_t)a * Lc] * V + sorted[(size_t)a * Lc + 1]] = (int32_t)lv.start.size();
        for (int64_t k = a; k < r; ++k)
          if (k == a || lcp[(size_t)k] < t + 1) { lv.start.push_back((int32_t)k); lv.tok.push_back(sorted[(size_t)k * Lc + t]); }
      }
      a = r;
    }
    if (lv.start.empty()) break;
    total += (int64_t)lv.start.size();
    if (total > max_entries) { if (t == 2) std::fill(out.idx2.begin(), out.idx2.end(), -1); break; }
    out.deep.push_back(std::move(lv));
  }
}

// ---- docid_to_smtid.json ------------------------------------------------------------------------------------
// One pass over the file with a 4 MB read buffer; no DOM. The reference loads this file with ujson into a dict of
// 8.8 M Python lists (minutes and tens of GB, evaluate.py:400-402); here it becomes a uint16 matrix directly.
namespace {
struct Reader {
  FILE* f;
  std::vector<char> buf;
  size_t pos = 0, len = 0;
  explicit Reader(FILE* fp) : f(fp), buf(4 << 20) {}
  int peek() {
    if (pos == len) { len = std::fread(buf.data(), 1, buf.size(), f); pos = 0; if (len == 0) return -1; }
    return (unsigned char)buf[pos];
  }
  int get() { const int c = peek(); if (c >= 0) ++pos; return c; }
  int skip_ws() { int c; while ((c = peek()) == ' ' || c == '\n' || c == '\r' || c == '\t') ++pos; return c; }
};
}  // namespace

int read_docid_to_smtid(const char* path, std::vector<uint16_t>& codes, std::string& keys, int64_t& N, int& L,
                        std::string& err) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { err = std::string("cannot open ") + path; return -1; }
  Reader r(f);
  auto fail = [&](const std::string& m) { err = m + " (entry " + std::to_string(N) + ")"; std::fclose(f); return -2; };
  codes.clear(); keys.clear(); N = 0; L = -1;
  if (r.skip_ws() != '{') return fail("expected '{'");
  r.get();
  if (r.skip_ws() == '}') { std::fclose(f); err = "empty docid_to_smtid"; return -2; }
  for (;;) {
    if (r.skip_ws() != '"') return fail("expected a quoted docid");
    r.get();
    if (N) keys.push_back('\n');
    for (int c; (c = r.get()) != '"';) {
      if (c < 0) return fail("unterminated key");
      if (c == '\\' || c == '\n') return fail("escaped characters in docids are not supported");
      keys.push_back((char)c);
    }
    if (r.skip_ws() != ':') return fail("expected ':'");
    r.get();
    if (r.skip_ws() != '[') return fail("expected '['");
    r.get();
    int n = 0;
    for (;;) {
      int c = r.skip_ws();
      bool neg = false;
      if (c == '-') { neg = true; r.get(); c = r.peek(); }
      if (c < '0' || c > '9') return fail("expected an integer");
      long v = 0;
      while ((c = r.peek()) >= '0' && c <= '9') { v = v * 10 + (c - '0'); r.get(); if (v > 1000000) return fail("code out of range"); }
      if (n == 0) {
        if (!(neg && v == 1)) return fail("smtid lists must start with -1");
      } else {
        if (neg || v > 65535) return fail("code out of range (0..65535)");
        codes.push_back((uint16_t)v);
      }
      ++n;
      c = r.skip_ws();
      if (c == ',') { r.get(); continue; }
      if (c == ']') { r.get(); break; }
      return fail("expected ',' or ']'");
    }
    if (L < 0) L = n - 1;
    if (n - 1 != L || L < 1) return fail("ragged or empty smtid list");
    ++N;
    const int c = r.skip_ws();
    if (c == ',') { r.get(); continue; }
    if (c == '}') break;
    return fail("expected ',' or '}'");
  }
  std::fclose(f);
  return 0;
}

// ---- binary trie file -----------------------------------------------------------------------------------------------
// "RPRTRIE2" | int64 N, L, V, key_bytes, src_size, src_mtime_ns | sorted codes uint16[N*L] | perm int64[N] | keys
// keys = the docid strings in ORIGINAL row order joined by '\n' (key_bytes may be 0: no docids stored);
// src_size / src_mtime_ns identify the docid_to_smtid.json the file was built from (0 = unknown), so that a caller can
// tell a stale cache from a fresh one without parsing the JSON.
static const char kMagic[8] = {'R', 'P', 'R', 'T', 'R', 'I', 'E', '2'};

int save_trie_file(const char* path, const std::vector<uint16_t>& sorted, const std::vector<int64_t>& perm,
                   int64_t N, int L, int V, const std::string& keys, int64_t src_size, int64_t src_mtime_ns) {
  FILE* f = std::fopen(path, "wb");
  if (!f) return -1;
  int64_t hdr[6] = {N, L, V, (int64_t)keys.size(), src_size, src_mtime_ns};
  bool ok = std::fwrite(kMagic, 1, 8, f) == 8 && std::fwrite(hdr, sizeof(int64_t), 6, f) == 6 &&
            std::fwrite(sorted.data(), sizeof(uint16_t), sorted.size(), f) == sorted.size() &&
            std::fwrite(perm.data(), sizeof(int64_t), perm.size(), f) == perm.size() &&
            (keys.empty() || std::fwrite(keys.data(), 1, keys.size(), f) == keys.size());
  ok = (std::fclose(f) == 0) && ok;
  return ok ? 0 : -1;
}

int trie_file_info(const char* path, int64_t hdr_out[6]) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return -1;
  char magic[8];
  const bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, kMagic, 8) == 0 &&
                  std::fread(hdr_out, sizeof(int64_t), 6, f) == 6;
  std::fclose(f);
  return ok ? 0 : -1;
}

// Nothing in the file is trusted: sizes are checked against the file length before any allocation, codes against V,
// rows against the sorted order the kernels' binary searches rely on, perm against [0, N) and for duplicates.
int load_trie_file(const char* path, std::vector<uint16_t>& sorted, std::vector<int64_t>& perm, int64_t& N, int& L,
                   int& V, std::string& keys, std::string& err) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { err = "cannot open file"; return -1; }
  auto fail = [&](const char* m) { err = m; std::fclose(f); return -1; };
  char magic[8];
  int64_t hdr[6];
  if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, kMagic, 8) != 0) return fail("not an RPRTRIE2 file");
  if (std::fread(hdr, sizeof(int64_t), 6, f) != 6) return fail("truncated header");
  N = hdr[0];
  const int64_t kb = hdr[3];
  if (!(N > 0 && N < ((int64_t)1 << 31) - 1)) return fail("N out of range");
  if (!(hdr[1] > 0 && hdr[1] <= 4096 && hdr[2] > 0 && hdr[2] <= 65536)) return fail("L or V out of range");
  if (kb < 0 || kb > ((int64_t)1 << 40)) return fail("key_bytes out of range");
  L = (int)hdr[1]; V = (int)hdr[2];
  const int64_t want = 8 + 6 * 8 + N * L * 2 + N * 8 + kb;
  if (std::fseek(f, 0, SEEK_END) != 0) return fail("seek failed");
  const int64_t have = (int64_t)std::ftell(f);
  if (have != want) return fail("file size does not match its header");
  if (std::fseek(f, 8 + 6 * 8, SEEK_SET) != 0) return fail("seek failed");
  try {
    sorted.resize((size_t)N * L);
    perm.resize((size_t)N);
    keys.resize((size_t)kb);
  } catch (const std::exception&) {
    return fail("out of host memory");
  }
  if (std::fread(sorted.data(), sizeof(uint16_t), sorted.size(), f) != sorted.size() ||
      std::fread(perm.data(), sizeof(int64_t), perm.size(), f) != perm.size() ||
      (kb && std::fread(&keys[0], 1, (size_t)kb, f) != (size_t)kb))
    return fail("short read");
  std::fclose(f);
  for (size_t i = 0; i < sorted.size(); ++i)
    if (sorted[i] >= V) { err = "code >= V"; return -1; }
  for (int64_t i = 1; i < N; ++i) {
    const uint16_t* a = &sorted[(size_t)(i - 1) * L];
    const uint16_t* b = a + L;
    int l = 0;
    while (l < L && a[l] == b[l]) ++l;
    if (l < L && a[l] > b[l]) { err = "rows are not sorted"; return -1; }
  }
  std::vector<bool> seen((size_t)N, false);
  for (int64_t i = 0; i < N; ++i) {
    const int64_t p = perm[(size_t)i];
    if (p < 0 || p >= N || seen[(size_t)p]) { err = "perm is not a permutation of [0, N)"; return -1; }
    seen[(size_t)p] = true;
  }
  if (kb) {
    int64_t lines = 1;
    for (char c : keys) lines += c == '\n';
    if (lines != N) { err = "docid key count differs from N"; return -1; }
  }
  return 0;
}

}  // namespace rpr
