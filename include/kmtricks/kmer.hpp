// kmtricks/kmer.hpp -- the k-mer value type merge plugins are written against: km::Kmer<MAX_K> and km::Mmer with the
// public surface of reference include/kmtricks/kmer.hpp:92-152 (Mmer) and :155-632 (Kmer): 2-bit nucleotides A0 C1 T2 G3,
// nucleotide i of a k-mer (from its first) is digit k-1-i, little-endian 64-bit words, low word first -- the layout of the
// keys the kmx driver hands to IMergePlugin::process_kmer and of the .kmer / matrix files.  One generic word-array
// implementation for every MAX_K (the reference specialises 32 and 64 on uint64_t / __uint128_t for speed; same values).
// Own implementation: arithmetic on the word array, reverse complement by digit reversal with shifts and masks.
#pragma once
#include <algorithm>
#include <bitset>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <istream>
#include <limits>
#include <ostream>
#include <sstream>
#include <string>
#include <vector>

namespace km {

inline char nt_of(unsigned code) { return "ACTG"[code & 3u]; }
inline unsigned code_of(char c) { return ((unsigned char)c >> 1) & 3u; }      // A/a 0, C/c 1, T/t 2, G/g 3
inline std::string str_rev_comp(const std::string& s) {
  std::string r; r.reserve(s.size());
  for (auto it = s.rbegin(); it != s.rend(); ++it) r.push_back(nt_of(code_of(*it) ^ 2u));
  return r;
}

// an m-mer may be a minimizer unless it holds "AA" anywhere but at its very start (kmer.hpp:75-83; gatb Model.hpp:1220-1251)
inline bool is_valid_minimizer(uint32_t value, uint8_t size) {
  const uint32_t mask = (0xFFFFFFFFu >> ((32 - size * 2) + 4)) & 0x55555555u;
  uint32_t v = ~(value | (value >> 2));
  v = ((v >> 1) & v) & mask;
  return v == 0;
}

class Mmer {
 public:
  Mmer() {}
  Mmer(uint32_t value, uint8_t size) { set(value, size); }
  void set(uint32_t value, uint8_t size) { m_size = size; m_data = value; }
  Mmer rev_comp() const {
    uint32_t rev = 0, tmp = m_data;
    for (int i = 0; i < m_size; i++) { rev = (rev << 2) | ((tmp & 3u) ^ 2u); tmp >>= 2; }
    return Mmer(rev, m_size);
  }
  std::string to_string() const {
    std::string s(m_size, 'A'); uint32_t tmp = m_data;
    for (int i = m_size - 1; i >= 0; i--) { s[i] = nt_of(tmp & 3u); tmp >>= 2; }
    return s;
  }
  bool operator>(const Mmer& m) const { return m_data > m.m_data; }
  bool operator<(const Mmer& m) const { return m_data < m.m_data; }
  bool operator==(const Mmer& m) const { return m_data == m.m_data; }
  uint32_t value() const { return m_data; }
 private:
  uint32_t m_data {0};
  uint8_t m_size {0};
};

template <size_t MAX_K>
class Kmer {
 public:
  static constexpr size_t NWORDS = (MAX_K + 31) / 32;
  typedef const uint64_t* data_ptr64;
  typedef const uint8_t* data_ptr8;
  // (the reference's 32 and 64 specialisations name their storage types, kmer.hpp:645, 919; the generic one its array, :185)
  static std::string name() { return "Kmer<" + std::to_string(MAX_K) + "> - " + (MAX_K == 32 ? std::string("uint64_t") : MAX_K == 64 ? std::string("__uint128_t") : "uint64_t[" + std::to_string(NWORDS) + "]"); }
  static const size_t get_size_bits() { return 64 * NWORDS; }

  Kmer() { zero(); }
  Kmer(size_t kmer_size) { zero(); set_k(kmer_size); }
  Kmer(const std::string& str_kmer) { zero(); set_polynom(str_kmer); }

  void zero() { std::memset(m_data, 0, sizeof(m_data)); }
  void set_k(size_t k) { m_k = k; m_n = (k + 31) / 32; }
  size_t k() const { return m_k; }
  void set64(uint64_t v) { m_data[0] = v; }
  void set64_p(const uint64_t* p) { std::memcpy(m_data, p, m_n * 8); }
  void set_polynom(const char* data, size_t kmer_size) {
    set_k(kmer_size); zero();
    for (size_t i = 0; i < kmer_size; i++) { *this = (*this << 2); m_data[0] |= code_of(data[i]); }
  }
  void set_polynom(const std::string& s) { set_polynom(s.data(), s.size()); }

  uint64_t get64() const { return m_data[0]; }
  data_ptr64 get_data64() const { return m_data; }
  data_ptr8 get_data8() const { return reinterpret_cast<const uint8_t*>(m_data); }
  uint64_t* get_data64_unsafe() { return m_data; }

  // digit i (0 = the LAST nucleotide)
  uint8_t operator[](size_t i) const { return (uint8_t)((m_data[i / 32] >> (2 * (i % 32))) & 3u); }
  // nucleotide i counted from the first one
  char at(size_t i) const { return nt_of((*this)[m_k - i - 1]); }
  uint8_t at2bit(size_t i) const { return (*this)[m_k - i - 1]; }
  uint8_t byte_at(size_t i) const { return (*this)[m_k - i - 1]; }

  // comparisons: most significant word first
  bool operator<(const Kmer& o) const { for (size_t i = NWORDS; i-- > 0;) if (m_data[i] != o.m_data[i]) return m_data[i] < o.m_data[i]; return false; }
  bool operator==(const Kmer& o) const { for (size_t i = 0; i < NWORDS; i++) if (m_data[i] != o.m_data[i]) return false; return true; }
  bool operator!=(const Kmer& o) const { return !(*this == o); }
  bool operator<=(const Kmer& o) const { return !(o < *this); }
  bool operator>(const Kmer& o) const { return o < *this; }
  bool operator>=(const Kmer& o) const { return !(*this < o); }

  // arithmetic on the word array (carries ripple upwards)
  Kmer operator+(const Kmer& o) const { Kmer r = like(); unsigned c = 0; for (size_t i = 0; i < NWORDS; i++) { const uint64_t a = m_data[i], s = a + o.m_data[i], t = s + c; c = (s < a) | (t < s); r.m_data[i] = t; } return r; }
  Kmer operator+(uint64_t v) const { Kmer o = like(); o.m_data[0] = v; return *this + o; }
  Kmer operator-(const Kmer& o) const { Kmer r = like(); unsigned b = 0; for (size_t i = 0; i < NWORDS; i++) { const uint64_t a = m_data[i], d = a - o.m_data[i], t = d - b; b = (a < o.m_data[i]) | (d < b); r.m_data[i] = t; } return r; }
  Kmer operator-(uint64_t v) const { Kmer o = like(); o.m_data[0] = v; return *this - o; }
  Kmer operator*(uint32_t f) const { Kmer r = like(); unsigned __int128 c = 0; for (size_t i = 0; i < NWORDS; i++) { c += (unsigned __int128)m_data[i] * f; r.m_data[i] = (uint64_t)c; c >>= 64; } return r; }
  Kmer operator/(uint32_t f) const { Kmer r = like(); unsigned __int128 rem = 0; for (size_t i = NWORDS; i-- > 0;) { rem = (rem << 64) | m_data[i]; r.m_data[i] = (uint64_t)(rem / f); rem %= f; } return r; }
  uint32_t operator%(uint32_t& f) const { unsigned __int128 rem = 0; for (size_t i = NWORDS; i-- > 0;) { rem = ((rem << 64) | m_data[i]) % f; } return (uint32_t)rem; }
  Kmer operator^(const Kmer& o) const { Kmer r = like(); for (size_t i = 0; i < NWORDS; i++) r.m_data[i] = m_data[i] ^ o.m_data[i]; return r; }
  Kmer operator|(const Kmer& o) const { Kmer r = like(); for (size_t i = 0; i < NWORDS; i++) r.m_data[i] = m_data[i] | o.m_data[i]; return r; }
  Kmer operator&(const Kmer& o) const { Kmer r = like(); for (size_t i = 0; i < NWORDS; i++) r.m_data[i] = m_data[i] & o.m_data[i]; return r; }
  Kmer operator&(char c) const { Kmer r = like(); r.m_data[0] = m_data[0] & (uint64_t)(unsigned char)c; return r; }
  Kmer operator~() const { Kmer r = like(); for (size_t i = 0; i < NWORDS; i++) r.m_data[i] = ~m_data[i]; return r; }
  Kmer operator>>(uint32_t s) const {
    Kmer r = like(); const size_t w = s / 64, b = s % 64;
    for (size_t i = 0; i + w < NWORDS; i++) { uint64_t v = m_data[i + w] >> b; if (b && i + w + 1 < NWORDS) v |= m_data[i + w + 1] << (64 - b); r.m_data[i] = v; }
    return r;
  }
  Kmer operator<<(uint32_t s) const {
    Kmer r = like(); const size_t w = s / 64, b = s % 64;
    for (size_t i = NWORDS; i-- > w;) { uint64_t v = m_data[i - w] << b; if (b && i - w >= 1) v |= m_data[i - w - 1] >> (64 - b); r.m_data[i] = v; }
    return r;
  }
  Kmer& operator+=(const Kmer& o) { *this = *this + o; return *this; }
  Kmer& operator-=(const Kmer& o) { *this = *this - o; return *this; }
  Kmer& operator*=(uint32_t f) { *this = *this * f; return *this; }
  Kmer& operator/=(uint32_t f) { *this = *this / f; return *this; }
  Kmer& operator&=(const Kmer& o) { *this = *this & o; return *this; }
  Kmer& operator|=(const Kmer& o) { *this = *this | o; return *this; }
  Kmer& operator^=(const Kmer& o) { *this = *this ^ o; return *this; }
  Kmer& operator<<=(uint32_t s) { *this = *this << s; return *this; }
  Kmer& operator>>=(uint32_t s) { *this = *this >> s; return *this; }

  // reverse complement: digits reversed over the words in use, complemented (digit ^ 2), shifted down to k digits
  Kmer rev_comp() const {
    Kmer r = like();
    for (size_t i = 0; i < m_n; i++) {
      uint64_t x = m_data[i];
      x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
      x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
      x = __builtin_bswap64(x);
      r.m_data[m_n - 1 - i] = x ^ 0xAAAAAAAAAAAAAAAAULL;
    }
    return r.shr_used((uint32_t)(2 * (32 * m_n - m_k)));
  }
  Kmer canonical() const { Kmer r = rev_comp(); return (r < *this) ? r : *this; }

  std::string to_string() const { std::string s(m_k, 'A'); for (size_t i = 0; i < m_k; i++) s[m_k - i - 1] = nt_of((*this)[i]); return s; }
  std::string to_bit_string() const {
    std::stringstream ss;
    for (size_t i = 0; i < m_n; i++) ss << i << " " << std::bitset<64>(m_data[i]).to_string() << "\n";
    return ss.str();
  }
  void dump(std::ostream& stream) { stream.write(reinterpret_cast<char*>(m_data), (std::streamsize)(m_n * 8)); }
  void load(std::istream& stream) { stream.read(reinterpret_cast<char*>(m_data), (std::streamsize)(m_n * 8)); }

  std::vector<Mmer> mmers(uint8_t size) const {
    const size_t nb = m_k - size + 1;
    std::vector<Mmer> out(nb);
    for (size_t i = 0; i < nb; i++) { uint32_t v = 0; for (size_t j = i; j < i + size; j++) v = (v << 2) | byte_at(j); out[i].set(v, size); }
    return out;
  }
  // the smallest of the k-mer's m-mers, each taken as min(m-mer, its reverse complement), an m-mer that may not be a minimizer
  // counting as 4^m - 1 (kmer.hpp:592-632)
  Mmer minimizer(uint8_t size) const {
    const uint32_t def = (uint32_t)(((uint64_t)1 << (2 * size)) - 1);
    Mmer best(std::numeric_limits<uint32_t>::max(), size);
    for (const Mmer& m : mmers(size)) {
      const uint32_t rev = m.rev_comp().value(), v = rev < m.value() ? rev : m.value();
      const Mmer cand(is_valid_minimizer(v, size) ? v : def, size);
      if (cand < best) best = cand;
    }
    return best;
  }

 private:
  Kmer like() const { Kmer r; r.m_k = m_k; r.m_n = m_n; return r; }
  // shift right over the m_n words in use only (the reverse complement is built in them)
  Kmer shr_used(uint32_t s) const {
    Kmer r = like(); const size_t w = s / 64, b = s % 64;
    for (size_t i = 0; i + w < m_n; i++) { uint64_t v = m_data[i + w] >> b; if (b && i + w + 1 < m_n) v |= m_data[i + w + 1] << (64 - b); r.m_data[i] = v; }
    return r;
  }
  uint64_t m_data[NWORDS];
  size_t m_k {0};
  size_t m_n {0};
};

}  // namespace km
