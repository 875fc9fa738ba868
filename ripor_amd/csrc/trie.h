#pragma once
#include <stdint.h>
#include <vector>

namespace rpr {
int sort_codes(const uint16_t* codes, int64_t N, int L, std::vector<uint16_t>& sorted, std::vector<int64_t>& perm);
int save_trie_file(const char* path, const std::vector<uint16_t>& sorted, const std::vector<int64_t>& perm,
                   int64_t N, int L, int V);
int load_trie_file(const char* path, std::vector<uint16_t>& sorted, std::vector<int64_t>& perm, int64_t& N, int& L,
                   int& V);
}  // namespace rpr
