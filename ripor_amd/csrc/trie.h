#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace rpr {
int sort_codes(const uint16_t* codes, int64_t N, int L, std::vector<uint16_t>& sorted, std::vector<int64_t>& perm);
int save_trie_file(const char* path, const std::vector<uint16_t>& sorted, const std::vector<int64_t>& perm,
                   int64_t N, int L, int V, const std::string& keys, int64_t src_size, int64_t src_mtime_ns);
int load_trie_file(const char* path, std::vector<uint16_t>& sorted, std::vector<int64_t>& perm, int64_t& N, int& L,
                   int& V, std::string& keys, std::string& err);
// forced-tail statistics: frac[t] = share of the depth-t trie nodes that hold a single distinct L-token sequence
void trie_single_frac(const uint16_t* sorted, int64_t N, int Lc, int L, std::vector<double>& frac);
// Child arrays of the trie (the selection kernels read a node's children from them instead of probing the code matrix):
//   lvl0[c]            first sorted row whose code 0 is >= c            (V + 1 entries, lvl0[V] = N)
//   lvl1[c0 * V + c1]  first row >= (c0, c1)                            (V * V + 1 entries; only for V <= 1024, L >= 2)
//   deep[i]            level t = 2 + i in CSR form over the sorted rows: one entry per child of every depth-t node of more
//                      than `narrow` rows (narrower nodes are enumerated from their rows), in row order: start[k] = first row
//                      of the child, tok[k] = its code at position t. The children of node [lo, hi) are the entries from the
//                      one with start == lo up to the last with start < hi; a child ends where the next entry starts (or at hi).
//   idx2[c0 * V + c1]  entry of deep[0] holding the first child of node (c0, c1), -1 if that node is narrow or empty
//                      (V * V entries, only with lvl1)
// Levels are built while nodes of more than `narrow` rows exist, at most max_levels of them and max_entries entries in total.
struct ChildLevels {
  std::vector<int32_t> lvl0, lvl1, idx2;
  struct Level { std::vector<int32_t> start; std::vector<uint16_t> tok; };
  std::vector<Level> deep;
};
void build_child_levels(const uint16_t* sorted, int64_t N, int Lc, int V, int narrow, int max_levels, int64_t max_entries,
                        ChildLevels& out);
// header words {N, L, V, key_bytes, src_size, src_mtime_ns} of a trie file (no payload read)
int trie_file_info(const char* path, int64_t hdr_out[6]);
// Streaming reader of the reference's docid_to_smtid.json: {"docid": [-1, c1, ..., cL], ...}
// (aq_preprocess/create_customized_smtid_file.py:47-59). codes: [N, L] row-major in file order (the leading -1 is
// dropped), keys: the docid strings joined by '\n'. Returns 0, or a negative code with *err set.
int read_docid_to_smtid(const char* path, std::vector<uint16_t>& codes, std::string& keys, int64_t& N, int& L,
                        std::string& err);
}  // namespace rpr
