// kmtricks/kmer.hpp -- minimal k-mer value type for merge plugins (2-bit, A0 C1 T2 G3, little-endian
// 64-bit words; reference include/kmtricks/kmer.hpp:155-889 is the full class).  Only what a plugin
// needs to look at the key the driver hands to process_kmer: set_k, set64_p, at, to_string, words.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>

namespace km {

template <size_t MAX_K>
class Kmer {
 public:
  static constexpr size_t NWORDS = (MAX_K + 31) / 32;
  Kmer() { std::memset(m_data, 0, sizeof(m_data)); }
  void set_k(size_t k) { m_k = k; }
  size_t k() const { return m_k; }
  void set64_p(const uint64_t* p) { std::memcpy(m_data, p, ((m_k + 31) / 32) * 8); }
  const uint64_t* get_data64() const { return m_data; }
  // nucleotide i counted from the first (most significant) one
  char at(size_t i) const {
    static const char alpha[4] = {'A', 'C', 'T', 'G'};
    const size_t d = m_k - 1 - i;
    return alpha[(m_data[d >> 5] >> ((d & 31) * 2)) & 3u];
  }
  std::string to_string() const { std::string s(m_k, 'A'); for (size_t i = 0; i < m_k; i++) s[i] = at(i); return s; }

 private:
  uint64_t m_data[NWORDS];
  size_t m_k {0};
};

}  // namespace km
