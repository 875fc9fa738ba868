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
// header words {N, L, V, key_bytes, src_size, src_mtime_ns} of a trie file (no payload read)
int trie_file_info(const char* path, int64_t hdr_out[6]);
// Streaming reader of the reference's docid_to_smtid.json: {"docid": [-1, c1, ..., cL], ...}
// (aq_preprocess/create_customized_smtid_file.py:47-59). codes: [N, L] row-major in file order (the leading -1 is
// dropped), keys: the docid strings joined by '\n'. Returns 0, or a negative code with *err set.
int read_docid_to_smtid(const char* path, std::vector<uint16_t>& codes, std::string& keys, int64_t& N, int& L,
                        std::string& err);
}  // namespace rpr
