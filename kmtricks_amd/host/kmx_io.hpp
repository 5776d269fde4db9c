// kmx_io.hpp -- byte-exact readers/writers of the kmtricks run-directory files the counting/merge
// path touches (layouts: SURVEY.md Appendix A; reference include/kmtricks/io/{io_common,kmer_file,
// hash_file,matrix_file,pa_matrix_file,vector_matrix_file,superk_file}.hpp, hash.hpp:52-60,
// repartition.hpp:58-67, merge.hpp:72-83).  Plain structs + free functions; headers are written field
// by field exactly as the reference's serialize() methods do.  Uncompressed bodies only (--cpr is out of scope).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>
#include <zlib.h>

namespace kmxio {

constexpr uint64_t MAGIC_BASE = 0x736b636972746d6bULL;         // "kmtricks"
constexpr uint64_t MAGIC_KMER = 0x72656d6bULL, MAGIC_HASH = 0x68736168ULL;
constexpr uint64_t MAGIC_MATRIX = 0x6b5f78697274616dULL, MAGIC_MATRIX_HASH = 0x685f78697274616dULL;
constexpr uint64_t MAGIC_PA = 0x6b5f74616d6170ULL, MAGIC_PA_HASH = 0x685f74616d6170ULL;
constexpr uint64_t MAGIC_BITMATRIX = 0x74616d746962ULL, MAGIC_SUPERK = 0x6b7265707573ULL;

struct IoError : std::runtime_error { using std::runtime_error::runtime_error; };

class Out {
 public:
  explicit Out(const std::string& path) : f_(fopen(path.c_str(), "wb")), path_(path) {
    if (!f_) throw IoError("Unable to write at " + path);
    setvbuf(f_, nullptr, _IOFBF, 1 << 20);
  }
  ~Out() { if (f_) fclose(f_); }
  template <typename T> void put(T v) { raw(&v, sizeof(T)); }
  void raw(const void* p, size_t n) { if (n && fwrite(p, 1, n, f_) != n) throw IoError("write failed: " + path_); }
  void base_header(bool compressed = false) { put<uint64_t>(MAGIC_BASE); put<uint32_t>(0); put<uint8_t>(compressed ? 1 : 0); }
 private:
  FILE* f_; std::string path_;
};

inline std::vector<uint8_t> slurp(const std::string& path) {
  std::ifstream in(path, std::ios::binary);
  if (!in) throw IoError("Unable to read at " + path);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}
template <typename T> inline T rd(const uint8_t* p) { T v; memcpy(&v, p, sizeof(T)); return v; }

// ---- counts/partition_<p>/<id>.kmer (io/kmer_file.hpp:31-40, 102-108) -----------------------------
inline void write_kmer_file(const std::string& path, uint32_t k, uint32_t id, uint32_t part,
                            const uint64_t* keys, const uint32_t* counts, uint64_t n) {
  const uint32_t slots = (k + 31) / 32;
  Out o(path); o.base_header();
  o.put<uint64_t>(MAGIC_KMER); o.put<uint32_t>(k); o.put<uint32_t>(slots); o.put<uint32_t>(4); o.put<uint32_t>(id); o.put<uint32_t>(part);
  std::vector<uint8_t> rec((size_t)n * (slots * 8 + 4));
  for (uint64_t i = 0; i < n; i++) { memcpy(&rec[i * (slots * 8 + 4)], keys + i * slots, slots * 8); memcpy(&rec[i * (slots * 8 + 4) + slots * 8], counts + i, 4); }
  o.raw(rec.data(), rec.size());
}
// -> packed records (key words + u32 count) = what kmx_merge takes; widens 1/2-byte counts
inline std::vector<uint8_t> read_kmer_records(const std::string& path, uint32_t* k_out, uint32_t* slots_out) {
  std::vector<uint8_t> raw = slurp(path);
  if (raw.size() < 41 || rd<uint64_t>(&raw[0]) != MAGIC_BASE || rd<uint64_t>(&raw[13]) != MAGIC_KMER) throw IoError("Invalid file format: " + path);
  if (raw[12]) throw IoError("compressed count files (--cpr) are not supported: " + path);
  const uint32_t k = rd<uint32_t>(&raw[21]), slots = rd<uint32_t>(&raw[25]), cs = rd<uint32_t>(&raw[29]);
  if (k_out) *k_out = k;
  if (slots_out) *slots_out = slots;
  const size_t rin = slots * 8 + cs, rout = slots * 8 + 4, n = (raw.size() - 41) / rin;
  std::vector<uint8_t> out(n * rout, 0);
  for (size_t i = 0; i < n; i++) { memcpy(&out[i * rout], &raw[41 + i * rin], slots * 8); memcpy(&out[i * rout + slots * 8], &raw[41 + i * rin + slots * 8], cs); }
  return out;
}

// ---- counts/partition_<p>/<id>.hash (io/hash_file.hpp:31-38, 91-131): blocks [u64 n][n x u64][n x count] ----
inline void write_hash_file(const std::string& path, uint32_t id, uint32_t part, const uint64_t* h, const uint32_t* c, uint64_t n) {
  Out o(path); o.base_header();
  o.put<uint64_t>(MAGIC_HASH); o.put<uint32_t>(4); o.put<uint32_t>(id); o.put<uint32_t>(part);
  for (uint64_t i = 0; i < n; i += 4096) {
    const uint64_t m = std::min<uint64_t>(4096, n - i);
    o.put<uint64_t>(m); o.raw(h + i, m * 8); o.raw(c + i, m * 4);
  }
}
inline std::vector<uint8_t> read_hash_records(const std::string& path, uint32_t* part_out) {
  std::vector<uint8_t> raw = slurp(path);
  if (raw.size() < 33 || rd<uint64_t>(&raw[0]) != MAGIC_BASE || rd<uint64_t>(&raw[13]) != MAGIC_HASH) throw IoError("Invalid file format: " + path);
  if (raw[12]) throw IoError("TurboPFor-compressed hash files (--cpr) are not supported: " + path);
  const uint32_t cs = rd<uint32_t>(&raw[21]);
  if (part_out) *part_out = rd<uint32_t>(&raw[29]);
  std::vector<uint8_t> out; size_t off = 33;
  while (off + 8 <= raw.size()) {
    const uint64_t n = rd<uint64_t>(&raw[off]); off += 8;
    if (off + n * (8 + cs) > raw.size()) throw IoError("truncated hash file: " + path);
    const size_t base = out.size(); out.resize(base + n * 12, 0);
    for (uint64_t i = 0; i < n; i++) { memcpy(&out[base + i * 12], &raw[off + i * 8], 8); memcpy(&out[base + i * 12 + 8], &raw[off + n * 8 + i * cs], cs); }
    off += n * (8 + cs);
  }
  return out;
}

// ---- matrices (io/matrix_file.hpp:31-41, 199-207; pa_matrix_file.hpp:31-41, 178-186; vector_matrix_file.hpp:31-40) ----
inline void matrix_count_header(Out& o, uint32_t k, uint32_t n, uint32_t part) {   // merge.hpp:264: count_slots is the literal 1, id 0
  o.base_header(); o.put<uint64_t>(MAGIC_MATRIX); o.put<uint32_t>(k); o.put<uint32_t>((k + 31) / 32); o.put<uint32_t>(1); o.put<uint32_t>(n); o.put<uint32_t>(0); o.put<uint32_t>(part);
}
inline void matrix_count_hash_header(Out& o, uint32_t n, uint32_t part) {           // merge.hpp:521
  o.base_header(); o.put<uint64_t>(MAGIC_MATRIX_HASH); o.put<uint32_t>(4); o.put<uint32_t>(n); o.put<uint32_t>(0); o.put<uint32_t>(part);
}
inline void matrix_pa_header(Out& o, uint32_t k, uint32_t n, uint32_t part) {
  o.base_header(); o.put<uint64_t>(MAGIC_PA); o.put<uint32_t>(k); o.put<uint32_t>((k + 31) / 32); o.put<uint32_t>(n); o.put<uint32_t>((n + 7) / 8); o.put<uint32_t>(0); o.put<uint32_t>(part);
}
inline void matrix_pa_hash_header(Out& o, uint32_t n, uint32_t part) {
  o.base_header(); o.put<uint64_t>(MAGIC_PA_HASH); o.put<uint32_t>(n); o.put<uint32_t>((n + 7) / 8); o.put<uint32_t>(0); o.put<uint32_t>(part);
}
inline void matrix_bf_header(Out& o, uint32_t bits, uint64_t first, uint64_t window, uint32_t part) {
  o.base_header(); o.put<uint64_t>(MAGIC_BITMATRIX); o.put<uint32_t>(bits); o.put<uint64_t>(first); o.put<uint64_t>(window); o.put<uint32_t>(0); o.put<uint32_t>(part);
}

// ---- superkmers/<id>/skp.<p> (io/superk_file.hpp:30-35; superk_storage.hpp:187-225, 296-309) ----
// blocks of <= 32768 bytes of whole records, each preceded by its u32 size
struct SuperkBlockWriter {
  Out out; std::vector<uint8_t> buf; uint64_t kmers = 0, bytes = 0;
  SuperkBlockWriter(const std::string& path, uint32_t part) : out(path) { out.base_header(); out.put<uint64_t>(MAGIC_SUPERK); out.put<uint32_t>(part); buf.reserve(32768); }
  void add_stream(const uint8_t* s, uint64_t len, uint32_t k) {   // concatenated records [u8 n][bytes]
    uint64_t pos = 0;
    while (pos < len) {
      const uint32_t n = s[pos]; const uint64_t nb = ((uint64_t)k + n - 1 + 3) / 4;   // payload bytes
      if (buf.size() + nb + 1 > 32768) flush();
      buf.insert(buf.end(), s + pos, s + pos + 1 + nb);
      kmers += n; pos += 1 + nb;
    }
  }
  void flush() { if (buf.empty()) return; out.put<uint32_t>((uint32_t)buf.size()); out.raw(buf.data(), buf.size()); bytes += buf.size() + 4; buf.clear(); }
};
inline std::vector<uint8_t> read_superk_stream(const std::string& path) {
  std::vector<uint8_t> raw = slurp(path);
  if (raw.size() < 25 || rd<uint64_t>(&raw[0]) != MAGIC_BASE || rd<uint64_t>(&raw[13]) != MAGIC_SUPERK) throw IoError("Invalid file format: " + path);
  std::vector<uint8_t> out; size_t off = 25;
  while (off + 4 <= raw.size()) { const uint32_t n = rd<uint32_t>(&raw[off]); off += 4; out.insert(out.end(), raw.begin() + off, raw.begin() + off + n); off += n; }
  return out;
}

// ---- hash.info (hash.hpp:31-60) -------------------------------------------------------------------
struct HashWindow {
  uint64_t bloom = 0, parts = 0, wbits = 0, wbytes = 0; uint32_t msize = 0;
  HashWindow() {}
  HashWindow(uint64_t bloom_size, uint64_t nb_parts, uint32_t m) : parts(nb_parts), msize(m) {
    const uint64_t per = (bloom_size + nb_parts - 1) / nb_parts;        // ceil(bloom / parts)
    wbits = (per + 63) / 64 * 64; wbytes = (wbits + 7) / 8; bloom = wbits * nb_parts;
  }
  uint64_t lower(uint32_t p) const { return p * wbits; }
  uint64_t upper(uint32_t p) const { return (p + 1) * wbits - 1; }
  void save(const std::string& path) const { Out o(path); o.put(bloom); o.put(parts); o.put(wbits); o.put(wbytes); o.put(msize); }
};

// ---- repartition_gatb/repartition.minimRepart (gatb PartiInfo.cpp:271-297, repartition.hpp:58-67) ----
inline void write_repartition(const std::string& path, uint16_t nb_part, const std::vector<uint16_t>& table) {
  Out o(path); o.put<uint16_t>(nb_part); o.put<uint64_t>(table.size()); o.put<uint16_t>(1);
  o.raw(table.data(), table.size() * 2); o.put<uint8_t>(0); o.put<uint32_t>(0x12345678);
}
inline std::vector<uint16_t> read_repartition(const std::string& path, uint16_t* nb_part) {
  std::vector<uint8_t> raw = slurp(path);
  if (raw.size() < 12) throw IoError("bad repartition file: " + path);
  *nb_part = rd<uint16_t>(&raw[0]); const uint64_t n = rd<uint64_t>(&raw[2]);
  if (raw.size() < 12 + n * 2 + 5 || rd<uint32_t>(&raw[12 + n * 2 + 1]) != 0x12345678) throw IoError("bad repartition file: " + path);
  std::vector<uint16_t> t(n); memcpy(t.data(), &raw[12], n * 2); return t;
}

// ---- merge_infos/partition<p>.merge_info (merge.hpp:72-83) -------------------------------------------
inline void write_merge_info(const std::string& path, const uint64_t* stats, uint32_t n) {
  static const char* names[6] = {"NON_SOLID", "RESCUED", "UNIQUE_WO_RESCUE", "UNIQUE_W_RESCUE", "TOTAL_WO_RESCUE", "TOTAL_W_RESCUE"};
  std::ofstream out(path); if (!out) throw IoError("Unable to write at " + path);
  for (int r = 0; r < 6; r++) { out << names[r] << '\t'; for (uint32_t i = 0; i < n; i++) out << stats[(size_t)r * n + i] << '\t'; out << "\n"; }
}

// ---- FASTA / FASTQ (plain or gz) reader: kseq-style records, sequence lines joined (gatb BankFasta.cpp:390-570) ----
class SeqReader {
 public:
  explicit SeqReader(const std::string& path) : gz_(gzopen(path.c_str(), "rb")), path_(path) {
    if (!gz_) throw IoError("Unable to read at " + path);
    gzbuffer(gz_, 1 << 20);
  }
  ~SeqReader() { if (gz_) gzclose(gz_); }
  bool next(std::string& seq) {
    seq.clear();
    std::string line;
    if (!have_hdr_) { while (getline(line)) if (!line.empty() && (line[0] == '>' || line[0] == '@')) { hdr_ = line[0]; have_hdr_ = true; break; } if (!have_hdr_) return false; }
    have_hdr_ = false;
    if (hdr_ == '>') {
      while (getline(line)) { if (!line.empty() && line[0] == '>') { have_hdr_ = true; hdr_ = '>'; break; } append(seq, line); }
      return true;
    }
    // FASTQ: sequence lines until '+', then as many quality characters as bases
    while (getline(line)) { if (!line.empty() && line[0] == '+') break; append(seq, line); }
    size_t q = 0;
    while (q < seq.size() && getline(line)) q += line.size();
    return true;
  }
 private:
  static void append(std::string& s, const std::string& l) { for (char c : l) if (c != ' ' && c != '\t' && c != '\r') s.push_back(c); }
  bool getline(std::string& line) {
    line.clear();
    char buf[65536];
    for (;;) {
      if (!gzgets(gz_, buf, sizeof(buf))) return !line.empty();
      const size_t n = strlen(buf);
      line.append(buf, n);
      if (n && buf[n - 1] == '\n') { line.pop_back(); if (!line.empty() && line.back() == '\r') line.pop_back(); return true; }
    }
  }
  gzFile gz_; std::string path_; char hdr_ = 0; bool have_hdr_ = false;
};

}  // namespace kmxio
