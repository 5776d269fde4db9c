// kmx_io.hpp -- byte-exact readers/writers of the kmtricks run-directory files the counting/merge
// path touches (layouts: SURVEY.md Appendix A; reference include/kmtricks/io/{io_common,kmer_file,
// hash_file,matrix_file,pa_matrix_file,vector_matrix_file,superk_file}.hpp, hash.hpp:52-60,
// repartition.hpp:58-67, merge.hpp:72-83, howde_utils.hpp:56-187; gatb PartiInfo.hpp:266-287,
// Configuration.cpp:145-178).  Plain structs + free functions; headers are written field by field exactly
// as the reference's serialize() methods do.  Bodies are raw, or one lz4 frame with --cpr
// (io/lz4_stream.hpp:89-159: any LZ4F reader reads it; the compressed bytes themselves are not pinned).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cerrno>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>
#include <memory>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

// The lz4 frame API (liblz4 >= 1.7, stable ABI): the image ships liblz4.so.1 without its development header, so the
// few entry points used are declared here (signatures of lz4frame.h; preferences / options are always NULL).
extern "C" {
typedef struct LZ4F_dctx_s LZ4F_dctx;
size_t LZ4F_compressFrameBound(size_t srcSize, const void* preferences);
size_t LZ4F_compressFrame(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const void* preferences);
unsigned LZ4F_isError(size_t code);
const char* LZ4F_getErrorName(size_t code);
size_t LZ4F_createDecompressionContext(LZ4F_dctx** dctx, unsigned version);
size_t LZ4F_freeDecompressionContext(LZ4F_dctx* dctx);
size_t LZ4F_decompress(LZ4F_dctx* dctx, void* dst, size_t* dstSize, const void* src, size_t* srcSize, const void* options);
typedef struct LZ4F_cctx_s LZ4F_cctx;
size_t LZ4F_createCompressionContext(LZ4F_cctx** cctx, unsigned version);
size_t LZ4F_freeCompressionContext(LZ4F_cctx* cctx);
size_t LZ4F_compressBegin(LZ4F_cctx* cctx, void* dst, size_t dstCapacity, const void* preferences);
size_t LZ4F_compressBound(size_t srcSize, const void* preferences);
size_t LZ4F_compressUpdate(LZ4F_cctx* cctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize, const void* options);
size_t LZ4F_compressEnd(LZ4F_cctx* cctx, void* dst, size_t dstCapacity, const void* options);
}
#define KMX_LZ4F_VERSION 100

namespace kmxio {

constexpr uint64_t MAGIC_BASE = 0x736b636972746d6bULL;         // "kmtricks"
constexpr uint64_t MAGIC_KMER = 0x72656d6bULL, MAGIC_HASH = 0x68736168ULL;
constexpr uint64_t MAGIC_MATRIX = 0x6b5f78697274616dULL, MAGIC_MATRIX_HASH = 0x685f78697274616dULL;
constexpr uint64_t MAGIC_PA = 0x6b5f74616d6170ULL, MAGIC_PA_HASH = 0x685f74616d6170ULL;
constexpr uint64_t MAGIC_BITMATRIX = 0x74616d746962ULL, MAGIC_SUPERK = 0x6b7265707573ULL;

struct IoError : std::runtime_error { using std::runtime_error::runtime_error; };

// ---- lz4 frames (--cpr) ---------------------------------------------------------------------------------
inline std::vector<uint8_t> lz4_compress(const void* src, size_t n) {
  std::vector<uint8_t> out(LZ4F_compressFrameBound(n, nullptr));
  const size_t r = LZ4F_compressFrame(out.data(), out.size(), src, n, nullptr);
  if (LZ4F_isError(r)) throw IoError(std::string("lz4 compression failed: ") + LZ4F_getErrorName(r));
  out.resize(r);
  return out;
}
inline std::vector<uint8_t> lz4_decompress(const uint8_t* src, size_t n, const std::string& what) {
  LZ4F_dctx* d = nullptr;
  if (LZ4F_isError(LZ4F_createDecompressionContext(&d, KMX_LZ4F_VERSION))) throw IoError("lz4: no decompression context");
  std::vector<uint8_t> out; std::vector<uint8_t> buf(1 << 20);
  size_t pos = 0;
  size_t r = 0;
  while (pos < n) {
    size_t dn = buf.size(), sn = n - pos;
    r = LZ4F_decompress(d, buf.data(), &dn, src + pos, &sn, nullptr);
    if (LZ4F_isError(r)) { LZ4F_freeDecompressionContext(d); throw IoError("corrupt lz4 body: " + what); }
    out.insert(out.end(), buf.begin(), buf.begin() + dn);
    pos += sn;
    if (r == 0 && sn == 0 && dn == 0) break;
  }
  LZ4F_freeDecompressionContext(d);
  if (n && r != 0) throw IoError("truncated lz4 body (the frame does not end): " + what);      // (a disk that filled up, a crashed writer)
  return out;
}

// A kmtricks file being written: raw header (first layer), then the body -- streamed as is, or, when the file is compressed, as
// ONE lz4 frame fed piece by piece (LZ4F_compressBegin / Update / End with bounded buffers, as the reference's lz4 stream feeds 8 KB
// blocks into its frame, io/lz4_stream.hpp:89-159): nothing the size of the body is ever held.
class Out {
 public:
  explicit Out(const std::string& path) : f_(fopen(path.c_str(), "wb")), path_(path) {
    if (!f_) throw IoError("Unable to write at " + path);
    setvbuf(f_, nullptr, _IOFBF, 1 << 20);
  }
  Out(const Out&) = delete;
  ~Out() { try { close(); } catch (...) {} if (cctx_) LZ4F_freeCompressionContext(cctx_); }
  template <typename T> void put(T v) { raw(&v, sizeof(T)); }
  void raw(const void* p, size_t n) {
    if (!n) return;
    if (cpr_) {
      const uint8_t* b = (const uint8_t*)p;
      while (n) {      // pieces of at most 4 MB into the frame
        const size_t m = std::min<size_t>(n, PIECE);
        const size_t r = LZ4F_compressUpdate(cctx_, zbuf_.data(), zbuf_.size(), b, m, nullptr);
        if (LZ4F_isError(r)) throw IoError(std::string("lz4 compression failed: ") + LZ4F_getErrorName(r));
        if (r && fwrite(zbuf_.data(), 1, r, f_) != r) throw IoError("write failed: " + path_);
        b += m; n -= m;
      }
      return;
    }
    if (fwrite(p, 1, n, f_) != n) throw IoError("write failed: " + path_);
  }
  void base_header(bool compressed = false) { put<uint64_t>(MAGIC_BASE); put<uint32_t>(0); put<uint8_t>(compressed ? 1 : 0); hdr_cpr_ = compressed; }
  void begin_body() {          // everything after this call is body
    if (!hdr_cpr_ || cpr_) return;
    if (LZ4F_isError(LZ4F_createCompressionContext(&cctx_, KMX_LZ4F_VERSION))) throw IoError("lz4: no compression context");
    zbuf_.resize(std::max<size_t>(LZ4F_compressBound(PIECE, nullptr), 64) + 64);
    const size_t r = LZ4F_compressBegin(cctx_, zbuf_.data(), zbuf_.size(), nullptr);
    if (LZ4F_isError(r)) throw IoError(std::string("lz4 compression failed: ") + LZ4F_getErrorName(r));
    if (fwrite(zbuf_.data(), 1, r, f_) != r) throw IoError("write failed: " + path_);
    cpr_ = true;
  }
  void close() {
    if (!f_) return;
    if (cpr_) {
      cpr_ = false;
      const size_t r = LZ4F_compressEnd(cctx_, zbuf_.data(), zbuf_.size(), nullptr);
      if (LZ4F_isError(r)) throw IoError(std::string("lz4 compression failed: ") + LZ4F_getErrorName(r));
      if (fwrite(zbuf_.data(), 1, r, f_) != r) throw IoError("write failed: " + path_);
    }
    if (fclose(f_) != 0) { f_ = nullptr; throw IoError("write failed: " + path_); }
    f_ = nullptr;
  }
 private:
  static constexpr size_t PIECE = (size_t)4 << 20;
  FILE* f_; std::string path_; bool hdr_cpr_ = false, cpr_ = false; LZ4F_cctx* cctx_ = nullptr; std::vector<uint8_t> zbuf_;
};

inline std::vector<uint8_t> slurp(const std::string& path) {
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) throw IoError("Unable to read at " + path);
  struct stat st; if (fstat(fd, &st) != 0) { close(fd); throw IoError("Unable to read at " + path); }
  std::vector<uint8_t> v((size_t)st.st_size);
  size_t got = 0;
  while (got < v.size()) { const ssize_t r = read(fd, v.data() + got, v.size() - got); if (r <= 0) break; got += (size_t)r; }
  close(fd);
  if (got != v.size()) throw IoError("short read: " + path);
  return v;
}
template <typename T> inline T rd(const uint8_t* p) { T v; memcpy(&v, p, sizeof(T)); return v; }
// header check + the (decompressed) body of a kmtricks file whose header is hdr bytes long
inline std::vector<uint8_t> body_of(std::vector<uint8_t>& raw, size_t hdr, uint64_t magic, const std::string& path) {
  if (raw.size() < hdr || rd<uint64_t>(&raw[0]) != MAGIC_BASE || rd<uint64_t>(&raw[13]) != magic) throw IoError("Invalid file format: " + path);
  if (raw[12]) return lz4_decompress(raw.data() + hdr, raw.size() - hdr, path);
  return std::vector<uint8_t>(raw.begin() + hdr, raw.end());
}

// ---- counts/partition_<p>/<id>.kmer[.lz4] (io/kmer_file.hpp:31-40, 102-108) -----------------------------
inline void write_kmer_file(const std::string& path, uint32_t k, uint32_t id, uint32_t part,
                            const uint64_t* keys, const uint32_t* counts, uint64_t n, bool cpr = false) {
  const uint32_t slots = (k + 31) / 32;
  Out o(path); o.base_header(cpr);
  o.put<uint64_t>(MAGIC_KMER); o.put<uint32_t>(k); o.put<uint32_t>(slots); o.put<uint32_t>(4); o.put<uint32_t>(id); o.put<uint32_t>(part);
  o.begin_body();
  std::vector<uint8_t> rec((size_t)n * (slots * 8 + 4));
  for (uint64_t i = 0; i < n; i++) { memcpy(&rec[i * (slots * 8 + 4)], keys + i * slots, slots * 8); memcpy(&rec[i * (slots * 8 + 4) + slots * 8], counts + i, 4); }
  o.raw(rec.data(), rec.size());
  o.close();
}
// -> packed records (key words + u32 count) = what kmx_merge takes; widens 1/2-byte counts
inline std::vector<uint8_t> read_kmer_records(const std::string& path, uint32_t* k_out, uint32_t* slots_out) {
  std::vector<uint8_t> raw = slurp(path);
  std::vector<uint8_t> body = body_of(raw, 41, MAGIC_KMER, path);
  const uint32_t k = rd<uint32_t>(&raw[21]), slots = rd<uint32_t>(&raw[25]), cs = rd<uint32_t>(&raw[29]);
  if (k_out) *k_out = k;
  if (slots_out) *slots_out = slots;
  if (slots == 0 || slots > 16 || (cs != 1 && cs != 2 && cs != 4)) throw IoError("Invalid file format: " + path);
  if (body.size() % (slots * 8 + cs) != 0) throw IoError("truncated count file (its body is no whole number of records): " + path);
  if (cs == 4) return body;
  const size_t rin = slots * 8 + cs, rout = slots * 8 + 4, n = body.size() / rin;
  std::vector<uint8_t> out(n * rout, 0);
  for (size_t i = 0; i < n; i++) { memcpy(&out[i * rout], &body[i * rin], slots * 8); memcpy(&out[i * rout + slots * 8], &body[i * rin + slots * 8], cs); }
  return out;
}

// ---- counts/partition_<p>/<id>.hash (io/hash_file.hpp:31-38, 91-131): blocks [u64 n][n x u64][n x count] ----
// histograms/<id>.hist (io/hist_file.hpp:30-116): base header, then magic, k, sample index, lower, upper, unique, total,
// oob lower total, oob lower unique, oob upper total, oob upper unique (the order HistFileHeader::serialize writes them),
// then the unique bins and the total bins (u64 each, upper - lower + 1 of them).  oob = {lower unique, upper unique, lower total, upper total}.
constexpr uint64_t MAGIC_HIST = 0x747369686bULL;      // "khist" (io/io_common.hpp:58)
inline void write_hist_file(const std::string& path, uint32_t k, uint32_t idx, uint64_t lower, uint64_t upper,
                            const uint64_t* uniq_bins, const uint64_t* total_bins, const uint64_t* oob, const uint64_t* sums) {
  Out o(path); o.base_header(false);
  o.put<uint64_t>(MAGIC_HIST); o.put<uint32_t>(k); o.put<uint32_t>(idx); o.put<uint64_t>(lower); o.put<uint64_t>(upper);
  o.put<uint64_t>(sums[0]); o.put<uint64_t>(sums[1]); o.put<uint64_t>(oob[2]); o.put<uint64_t>(oob[0]); o.put<uint64_t>(oob[3]); o.put<uint64_t>(oob[1]);
  o.begin_body();
  o.raw(uniq_bins, (size_t)(upper - lower + 1) * 8); o.raw(total_bins, (size_t)(upper - lower + 1) * 8);
  o.close();
}
struct HistFile { uint32_t k = 0, idx = 0; uint64_t lower = 0, upper = 0, uniq = 0, total = 0, oob_ln = 0, oob_lu = 0, oob_un = 0, oob_uu = 0; std::vector<uint64_t> u, n; };
inline HistFile read_hist_file(const std::string& path) {
  std::vector<uint8_t> raw = slurp(path);
  std::vector<uint8_t> body = body_of(raw, 93, MAGIC_HIST, path);
  HistFile h; h.k = rd<uint32_t>(&raw[21]); h.idx = rd<uint32_t>(&raw[25]); h.lower = rd<uint64_t>(&raw[29]); h.upper = rd<uint64_t>(&raw[37]);
  h.uniq = rd<uint64_t>(&raw[45]); h.total = rd<uint64_t>(&raw[53]); h.oob_ln = rd<uint64_t>(&raw[61]); h.oob_lu = rd<uint64_t>(&raw[69]);
  h.oob_un = rd<uint64_t>(&raw[77]); h.oob_uu = rd<uint64_t>(&raw[85]);
  if (h.upper < h.lower || body.size() != (h.upper - h.lower + 1) * 16) throw IoError("Invalid file format: " + path);
  const size_t nb = (size_t)(h.upper - h.lower + 1);
  h.u.resize(nb); h.n.resize(nb);
  memcpy(h.u.data(), body.data(), nb * 8); memcpy(h.n.data(), body.data() + nb * 8, nb * 8);
  return h;
}
inline void write_hash_file(const std::string& path, uint32_t id, uint32_t part, const uint64_t* h, const uint32_t* c, uint64_t n) {
  Out o(path); o.base_header();
  o.put<uint64_t>(MAGIC_HASH); o.put<uint32_t>(4); o.put<uint32_t>(id); o.put<uint32_t>(part);
  for (uint64_t i = 0; i < n; i += 4096) {
    const uint64_t m = std::min<uint64_t>(4096, n - i);
    o.put<uint64_t>(m); o.raw(h + i, m * 8); o.raw(c + i, m * 4);
  }
  o.close();
}
inline std::vector<uint8_t> read_hash_records(const std::string& path, uint32_t* part_out) {
  std::vector<uint8_t> raw = slurp(path);
  if (raw.size() < 33 || rd<uint64_t>(&raw[0]) != MAGIC_BASE || rd<uint64_t>(&raw[13]) != MAGIC_HASH) throw IoError("Invalid file format: " + path);
  if (raw[12]) throw IoError("TurboPFor-compressed hash files (.hash.p4) are not supported: " + path);
  const uint32_t cs = rd<uint32_t>(&raw[21]);
  if (cs != 1 && cs != 2 && cs != 4) throw IoError("Invalid file format: " + path);
  if (part_out) *part_out = rd<uint32_t>(&raw[29]);
  std::vector<uint8_t> out; size_t off = 33;
  while (off + 8 <= raw.size()) {
    const uint64_t n = rd<uint64_t>(&raw[off]); off += 8;
    if (n > (raw.size() - off) / (8 + cs)) throw IoError("truncated hash file: " + path);
    const size_t base = out.size(); out.resize(base + n * 12, 0);
    for (uint64_t i = 0; i < n; i++) { memcpy(&out[base + i * 12], &raw[off + i * 8], 8); memcpy(&out[base + i * 12 + 8], &raw[off + n * 8 + i * cs], cs); }
    off += n * (8 + cs);
  }
  return out;
}

// ---- matrices (io/matrix_file.hpp:31-41, 199-207; pa_matrix_file.hpp:31-41, 178-186; vector_matrix_file.hpp:31-40) ----
inline void matrix_count_header(Out& o, uint32_t k, uint32_t n, uint32_t part, bool cpr = false) {   // merge.hpp:264: count_slots is the literal 1, id 0
  o.base_header(cpr); o.put<uint64_t>(MAGIC_MATRIX); o.put<uint32_t>(k); o.put<uint32_t>((k + 31) / 32); o.put<uint32_t>(1); o.put<uint32_t>(n); o.put<uint32_t>(0); o.put<uint32_t>(part);
  o.begin_body();
}
inline void matrix_count_hash_header(Out& o, uint32_t n, uint32_t part, bool cpr = false) {           // merge.hpp:521
  o.base_header(cpr); o.put<uint64_t>(MAGIC_MATRIX_HASH); o.put<uint32_t>(4); o.put<uint32_t>(n); o.put<uint32_t>(0); o.put<uint32_t>(part);
  o.begin_body();
}
inline void matrix_pa_header(Out& o, uint32_t k, uint32_t n, uint32_t part, bool cpr = false) {
  o.base_header(cpr); o.put<uint64_t>(MAGIC_PA); o.put<uint32_t>(k); o.put<uint32_t>((k + 31) / 32); o.put<uint32_t>(n); o.put<uint32_t>((n + 7) / 8); o.put<uint32_t>(0); o.put<uint32_t>(part);
  o.begin_body();
}
inline void matrix_pa_hash_header(Out& o, uint32_t n, uint32_t part, bool cpr = false) {
  o.base_header(cpr); o.put<uint64_t>(MAGIC_PA_HASH); o.put<uint32_t>(n); o.put<uint32_t>((n + 7) / 8); o.put<uint32_t>(0); o.put<uint32_t>(part);
  o.begin_body();
}
inline void matrix_bf_header(Out& o, uint32_t bits, uint64_t first, uint64_t window, uint32_t part) {   // never compressed (task.hpp:828-834)
  o.base_header(); o.put<uint64_t>(MAGIC_BITMATRIX); o.put<uint32_t>(bits); o.put<uint64_t>(first); o.put<uint64_t>(window); o.put<uint32_t>(0); o.put<uint32_t>(part);
}

// ---- superkmers/<id>/skp.<p> (io/superk_file.hpp:30-35; superk_storage.hpp:187-225, 296-309) ----
// blocks of <= 32768 bytes of whole records, each preceded by its u32 size
struct SuperkBlockWriter {
  Out out; std::vector<uint8_t> buf; uint64_t kmers = 0, bytes = 0;
  SuperkBlockWriter(const std::string& path, uint32_t part, bool cpr = false) : out(path) { out.base_header(cpr); out.put<uint64_t>(MAGIC_SUPERK); out.put<uint32_t>(part); out.begin_body(); buf.reserve(32768); }
  void add_stream(const uint8_t* s, uint64_t len, uint32_t k) {   // concatenated records [u8 n][bytes]
    uint64_t pos = 0;
    while (pos < len) {
      const uint32_t n = s[pos]; const uint64_t nb = ((uint64_t)k + n - 1 + 3) / 4;   // payload bytes
      if (pos + 1 + nb > len) throw IoError("malformed super-k-mer stream");
      if (buf.size() + nb + 1 > 32768) flush();
      buf.insert(buf.end(), s + pos, s + pos + 1 + nb);
      kmers += n; pos += 1 + nb;
    }
  }
  void flush() { if (buf.empty()) return; out.put<uint32_t>((uint32_t)buf.size()); out.raw(buf.data(), buf.size()); bytes += buf.size() + 4; buf.clear(); }
};
inline std::vector<uint8_t> read_superk_stream(const std::string& path) {
  std::vector<uint8_t> raw = slurp(path);
  std::vector<uint8_t> body = body_of(raw, 25, MAGIC_SUPERK, path);
  std::vector<uint8_t> out; size_t off = 0;
  while (off + 4 <= body.size()) {
    const uint32_t n = rd<uint32_t>(&body[off]); off += 4;
    if (n > body.size() - off) throw IoError("truncated super-k-mer file: " + path);
    out.insert(out.end(), body.begin() + off, body.begin() + off + n); off += n;
  }
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
  void save(const std::string& path) const { Out o(path); o.put(bloom); o.put(parts); o.put(wbits); o.put(wbytes); o.put(msize); o.close(); }
};

// ---- repartition_gatb/repartition.minimRepart (gatb PartiInfo.cpp:271-297, repartition.hpp:58-67) ----
inline void write_repartition(const std::string& path, uint16_t nb_part, const std::vector<uint16_t>& table) {
  Out o(path); o.put<uint16_t>(nb_part); o.put<uint64_t>(table.size()); o.put<uint16_t>(1);
  o.raw(table.data(), table.size() * 2); o.put<uint8_t>(0); o.put<uint32_t>(0x12345678);
  o.close();
}
inline std::vector<uint16_t> read_repartition(const std::string& path, uint16_t* nb_part) {
  std::vector<uint8_t> raw = slurp(path);
  if (raw.size() < 12) throw IoError("bad repartition file: " + path);
  *nb_part = rd<uint16_t>(&raw[0]); const uint64_t n = rd<uint64_t>(&raw[2]);
  if (n > raw.size() || raw.size() < 12 + n * 2 + 5 || rd<uint32_t>(&raw[12 + n * 2 + 1]) != 0x12345678) throw IoError("bad repartition file: " + path);
  std::vector<uint16_t> t(n); memcpy(t.data(), &raw[12], n * 2); return t;
}

// ---- config_gatb/gatb.config (gatb Configuration.cpp:145-178; members Configuration.hpp) ---------------------
// size_t kmerSize, minim_size, repartitionType, minimizerType; u64 max_disk_space; u32 max_memory; size_t nbCores,
// nb_partitions_in_parallel, abundanceUserNb, nbCores_per_partition; u64 estimateSeqNb, estimateSeqTotalSize,
// estimateSeqMaxSize, available_space, volume, kmersNb; u32 nb_passes, nb_partitions; u16 nb_bits_per_kmer, nb_banks;
// u32 nb_cached_items_per_core_per_part.  The sizing fields are system dependent in the reference (not byte-pinned):
// the fields a later stage reads back (k, m, nb_partitions, nb_banks, nb_passes) are exact.
struct GatbConfig {
  uint64_t kmer_size = 0, minim_size = 0, nb_cores = 1, est_seq_nb = 0, est_seq_total = 0, est_seq_max = 0, kmers_nb = 0;
  uint32_t nb_partitions = 0; uint16_t nb_banks = 0;
  void save(const std::string& path) const {
    Out o(path);
    o.put<uint64_t>(kmer_size); o.put<uint64_t>(minim_size); o.put<uint64_t>(0); o.put<uint64_t>(0);
    o.put<uint64_t>(0); o.put<uint32_t>(0); o.put<uint64_t>(nb_cores); o.put<uint64_t>(1); o.put<uint64_t>(1);
    o.put<uint64_t>(1); o.put<uint64_t>(est_seq_nb); o.put<uint64_t>(est_seq_total); o.put<uint64_t>(est_seq_max);
    o.put<uint64_t>(0); o.put<uint64_t>(0); o.put<uint64_t>(kmers_nb);
    o.put<uint32_t>(1); o.put<uint32_t>(nb_partitions); o.put<uint16_t>((uint16_t)(2 * kmer_size)); o.put<uint16_t>(nb_banks); o.put<uint32_t>(1);
    o.close();
  }
  static bool load(const std::string& path, GatbConfig& c) {
    std::vector<uint8_t> raw;
    try { raw = slurp(path); } catch (const IoError&) { return false; }
    if (raw.size() < 136) return false;
    c.kmer_size = rd<uint64_t>(&raw[0]); c.minim_size = rd<uint64_t>(&raw[8]); c.nb_partitions = rd<uint32_t>(&raw[128]);
    return true;
  }
};

// ---- superkmers/<id>/PartiInfoFile (gatb PartiInfo.hpp:266-287): one number per line -------------------------
inline void write_parti_info(const std::string& path, uint32_t nb_parts, uint64_t nb_minims, uint64_t nb_superk_total, uint64_t nb_kmer_total,
                             const uint64_t* part_counters /* nb_parts * (2 + 5*256) */, const uint64_t* minim_superks, const uint64_t* minim_kmers) {
  std::string s; s.reserve((size_t)nb_parts * 1282 * 3 + nb_minims * 6 + 64);
  char b[24];
  auto num = [&](uint64_t v) { if (v == 0) { s += "0\n"; return; } int n = 0; while (v) { b[n++] = (char)('0' + v % 10); v /= 10; } while (n) s += b[--n]; s += '\n'; };
  num(nb_parts); num(nb_minims); num(nb_superk_total); num(nb_kmer_total);
  for (size_t i = 0; i < (size_t)nb_parts * 1282; i++) num(part_counters[i]);
  for (uint64_t i = 0; i < nb_minims; i++) { num(minim_superks[i]); num(minim_kmers[i]); s += "0\n"; }   // nb_kxmers per minimizer stays 0 in this stage
  Out o(path); o.raw(s.data(), s.size()); o.close();
}

// the same file from the device's own u32 tables of one sample (kmx_superk_raw): part_radix[p][x][radix] = kx-mers of x + 1 k-mers;
// a partition's nb_kmers / nb_kxmers are sums over its 1280 counters (PartiInfo::incKmer_and_rad)
inline void write_parti_info_raw(const std::string& path, uint32_t nb_parts, uint64_t nb_minims, uint64_t nb_superk_total,
                                 const uint32_t* part_radix, const uint32_t* minim_superks, const uint32_t* minim_kmers) {
  std::vector<char> s((size_t)nb_parts * 1282 * 11 + nb_minims * 24 + 128);      // (a u32 is at most 10 digits + the newline)
  char* w = s.data();
  auto num = [&](uint64_t v) {
    if (v < 10) { *w++ = (char)('0' + v); *w++ = '\n'; return; }
    char b[24]; int n = 0; while (v) { b[n++] = (char)('0' + v % 10); v /= 10; } while (n) *w++ = b[--n]; *w++ = '\n';
  };
  uint64_t nk_total = 0;
  for (uint32_t p = 0; p < nb_parts; p++) for (uint32_t x = 0; x < 5; x++) { uint64_t c = 0; const uint32_t* r = part_radix + ((size_t)p * 5 + x) * 256; for (int i = 0; i < 256; i++) c += r[i]; nk_total += c * (x + 1); }
  num(nb_parts); num(nb_minims); num(nb_superk_total); num(nk_total);
  for (uint32_t p = 0; p < nb_parts; p++) {
    const uint32_t* r = part_radix + (size_t)p * 1280;
    uint64_t nk = 0, nx = 0;
    for (uint32_t x = 0; x < 5; x++) { uint64_t c = 0; for (int i = 0; i < 256; i++) c += r[x * 256 + i]; nx += c; nk += c * (x + 1); }
    num(nk); num(nx);
    for (int i = 0; i < 1280; i++) num(r[i]);
  }
  for (uint64_t i = 0; i < nb_minims; i++) { num(minim_superks[i]); num(minim_kmers[i]); *w++ = '0'; *w++ = '\n'; }
  const int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0666);
  if (fd < 0) throw IoError("Unable to write at " + path);
  const size_t n = (size_t)(w - s.data()); size_t done = 0;
  while (done < n) { const ssize_t r = write(fd, s.data() + done, n - done); if (r <= 0) break; done += (size_t)r; }
  close(fd);
  if (done != n) throw IoError("write failed: " + path);
}

// ... and with the per-minimizer records in sparse form ({minimizer, super-k-mers, k-mers} triples in any order,
// kmx_superk_raw::minim_sparse): the minimizers that do not occur are runs of "0\n0\n0\n"
inline void write_parti_info_sparse(const std::string& path, uint32_t nb_parts, uint64_t nb_minims, uint64_t nb_superk_total,
                                    const uint32_t* part_radix, const uint32_t* triples, uint64_t n_triples) {
  // the triples' positions in minimizer order: an LSD radix sort, 11 bits a pass (10^5 entries: a std::sort of pairs took 6 ms per
  // sample of the pool's time, this takes under one)
  std::vector<std::pair<uint32_t, uint32_t>> idx(n_triples), tmp(n_triples);      // minimizer, position
  for (uint64_t i = 0; i < n_triples; i++) idx[i] = {triples[3 * i], (uint32_t)i};
  { unsigned bits = 0; while (bits < 32 && (nb_minims >> bits) > 1) bits++;
    for (unsigned sh = 0; sh < bits; sh += 11) {
      uint32_t cnt[2049] = {0};
      for (auto& e : idx) cnt[((e.first >> sh) & 2047u) + 1]++;
      for (int b = 0; b < 2048; b++) cnt[b + 1] += cnt[b];
      for (auto& e : idx) tmp[cnt[(e.first >> sh) & 2047u]++] = e;
      idx.swap(tmp);
    } }
  // (6 MB of text per sample at m = 10, most of it the "0\n0\n0\n" of the minimizers that do not occur: the buffer is not
  //  cleared first, and the runs of zeros are copied from a block of them)
  const size_t cap = (size_t)nb_parts * 1282 * 11 + nb_minims * 6 + n_triples * 24 + 128;
  std::unique_ptr<char[]> s(new char[cap]);
  char* w = s.get();
  static const std::vector<char> zero_block = []() { std::vector<char> z(6 * 8192); for (size_t i = 0; i < z.size(); i += 6) memcpy(&z[i], "0\n0\n0\n", 6); return z; }();
  auto num = [&](uint64_t v) {
    if (v < 10) { *w++ = (char)('0' + v); *w++ = '\n'; return; }
    char b[24]; int n = 0; while (v) { b[n++] = (char)('0' + v % 10); v /= 10; } while (n) *w++ = b[--n]; *w++ = '\n';
  };
  uint64_t nk_total = 0;
  for (uint32_t p = 0; p < nb_parts; p++) for (uint32_t x = 0; x < 5; x++) { uint64_t c = 0; const uint32_t* r = part_radix + ((size_t)p * 5 + x) * 256; for (int i = 0; i < 256; i++) c += r[i]; nk_total += c * (x + 1); }
  num(nb_parts); num(nb_minims); num(nb_superk_total); num(nk_total);
  for (uint32_t p = 0; p < nb_parts; p++) {
    const uint32_t* r = part_radix + (size_t)p * 1280;
    uint64_t nk = 0, nx = 0;
    for (uint32_t x = 0; x < 5; x++) { uint64_t c = 0; for (int i = 0; i < 256; i++) c += r[x * 256 + i]; nx += c; nk += c * (x + 1); }
    num(nk); num(nx);
    for (int i = 0; i < 1280; i++) num(r[i]);
  }
  auto zeros = [&](uint64_t n) { while (n) { const uint64_t c = std::min<uint64_t>(n, 8192); memcpy(w, zero_block.data(), c * 6); w += c * 6; n -= c; } };
  uint64_t at = 0;
  for (auto& e : idx) {
    if (e.first >= nb_minims) throw IoError("minimizer out of range in the statistics");
    zeros(e.first - at);
    num(triples[3 * e.second + 1]); num(triples[3 * e.second + 2]); *w++ = '0'; *w++ = '\n';
    at = (uint64_t)e.first + 1;
  }
  zeros(nb_minims - at);
  const int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0666);
  if (fd < 0) throw IoError("Unable to write at " + path);
  const size_t n = (size_t)(w - s.get()); size_t done = 0;
  while (done < n) { const ssize_t r = write(fd, s.get() + done, n - done); if (r <= 0) break; done += (size_t)r; }
  close(fd);
  if (done != n) throw IoError("write failed: " + path);
}

// ---- merge_infos/partition<p>.merge_info (merge.hpp:72-83) -------------------------------------------
inline void write_merge_info(const std::string& path, const uint64_t* stats, uint32_t n) {
  static const char* names[6] = {"NON_SOLID", "RESCUED", "UNIQUE_WO_RESCUE", "UNIQUE_W_RESCUE", "TOTAL_WO_RESCUE", "TOTAL_W_RESCUE"};
  std::string s;
  for (int r = 0; r < 6; r++) { s += names[r]; s += '\t'; for (uint32_t i = 0; i < n; i++) { s += std::to_string(stats[(size_t)r * n + i]); s += '\t'; } s += "\n"; }
  Out o(path); o.raw(s.data(), s.size()); o.close();
}

// ---- filters/<id>.bf : one sample's Bloom filter in HowDeSBT's file format (howde_utils.hpp:56-187) --------------
// bffileheader (km_howdesbt bloom_filter_file.h -- that header is NOT in the reference snapshot, so this layout is
// restated from HowDeSBT's published format and is UNPINNED): u64 magic, u32 headerSize, u32 version, u32 bfKind,
// u32 padding, u32 smerSize, u32 numHashes, u64 hashSeed1, u64 hashSeed2, u64 hashModulus, u64 numBits,
// u32 numVectors, u32 setSizeKnown(+padding), u64 setSize, then one bfvectorinfo {u32 compressor, u32 name, u64 offset,
// u64 numBytes, u64 filterInfo}: 112 bytes (already a multiple of 16).  The vector itself is an uncompressed sdsl
// bit_vector: u64 number of bits, then the bits -- for p = 0..P-1 the W/8 bytes of sample s in partition p's
// transposed matrix (BloomBuilderFromHash::build, howde_utils.hpp:142-187: offset 49 + s * W/8 of matrix_<p>.cmbf).
constexpr uint64_t BF_MAGIC = 0xD532006662544253ULL;     // "SBTbf\0" + version tag (HowDeSBT bffileheaderMagic)
constexpr uint32_t BF_HEADER_BYTES = 112, BF_VERSION = 2, BF_KIND_SIMPLE = 1, BF_COMP_UNCOMPRESSED = 1;
inline void bf_header(uint8_t* h /* BF_HEADER_BYTES */, uint32_t kmer_size, uint64_t bloom_bits) {
  memset(h, 0, BF_HEADER_BYTES);
  auto w32 = [&](size_t o, uint32_t v) { memcpy(h + o, &v, 4); };
  auto w64 = [&](size_t o, uint64_t v) { memcpy(h + o, &v, 8); };
  w64(0, BF_MAGIC); w32(8, BF_HEADER_BYTES); w32(12, BF_VERSION); w32(16, BF_KIND_SIMPLE); w32(24, kmer_size); w32(28, 1);
  w64(32, 0); w64(40, 0); w64(48, bloom_bits); w64(56, bloom_bits); w32(64, 1); w32(68, 0); w64(72, 0);
  w32(80, BF_COMP_UNCOMPRESSED); w32(84, 0); w64(88, BF_HEADER_BYTES); w64(96, bloom_bits / 8 + 8); w64(104, 0);
}

// ---- FASTA / FASTQ (plain or gz) reader: kseq-style records, sequence lines joined (gatb BankFasta.cpp:390-570) ----
// Round 6: a sample's 30 MB of FASTA took a reader thread 20-35 ms (every byte through zlib's pass-through buffer, a line string, a
// per-character copy and the record's string before it reached the page-locked batch): 12-24 readers were the count stage's pace
// at 1000 x 5 Mbp.  Now a plain file is read with read(2) into a 1 MB block (gzread only when the file starts with the gzip magic),
// a line goes from there into the record with one memchr and one append, and its bytes are looked at one by one only when it holds a
// blank or a carriage return.
class SeqReader {
 public:
  explicit SeqReader(const std::string& path) : path_(path), buf_(new char[BUF]) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) throw IoError("Unable to read at " + path);
    unsigned char m[2] = {0, 0};
    const ssize_t r = ::pread(fd_, m, 2, 0);
    if (r == 2 && m[0] == 0x1f && m[1] == 0x8b) {
      gz_ = gzdopen(fd_, "rb");
      if (!gz_) { ::close(fd_); throw IoError("Unable to read at " + path); }
      gzbuffer(gz_, 1 << 20);
    } else {
#ifdef POSIX_FADV_SEQUENTIAL
      (void)posix_fadvise(fd_, 0, 0, POSIX_FADV_SEQUENTIAL);
#endif
    }
  }
  SeqReader(const SeqReader&) = delete;
  ~SeqReader() { if (gz_) gzclose(gz_); else if (fd_ >= 0) ::close(fd_); }
  // the next record's sequence as a VIEW: into the read block itself when the record is one header line and one sequence line that lie
  // whole in the block with nothing to strip (short reads: every record but the ones that straddle a refill) -- the caller copies it
  // once, to where it goes; else assembled in `tmp` as next() does.  The view holds until the next call.
  bool next_view(std::string& tmp, const char*& p, size_t& n) {
    if (hdr_ != '@' && pos_ < end_) {
      const char* const b = buf_.get(); const char* const e = b + end_;
      const char* s0 = nullptr;
      if (have_hdr_ && hdr_ == '>') s0 = b + pos_;                                  // the header is behind us: at the sequence line
      else if (!have_hdr_ && b[pos_] == '>') {                                      // at a header that lies in the block
        const char* h = (const char*)memchr(b + pos_, '\n', end_ - pos_);
        if (h && h + 1 < e) s0 = h + 1;
      }
      if (s0 && *s0 != '>') {
        const char* l = (const char*)memchr(s0, '\n', (size_t)(e - s0));
        // (the line behind it must start a new record: a second sequence line means joining, the end of the block means looking further)
        if (l && l + 1 < e && l[1] == '>') {
          const size_t m = (size_t)(l - s0);
          unsigned low = 0;
          for (size_t i = 0; i < m; i++) low |= (unsigned)((unsigned char)s0[i] <= ' ');
          if (!low) { p = s0; n = m; pos_ = (size_t)(l + 1 - b); hdr_ = '>'; have_hdr_ = false; return true; }
        }
      }
    }
    if (!next(tmp)) return false;
    p = tmp.data(); n = tmp.size();
    return true;
  }
  bool next(std::string& seq) {
    seq.clear();
    char c;
    if (!have_hdr_) {      // up to the next header line
      while (peek(c)) { if (c == '>' || c == '@') { hdr_ = c; have_hdr_ = true; skip_line(); break; } skip_line(); }
      if (!have_hdr_) return false;
    }
    have_hdr_ = false;
    if (hdr_ == '>') {
      while (peek(c)) { if (c == '>') { have_hdr_ = true; hdr_ = '>'; skip_line(); break; } take_line(seq); }
      return true;
    }
    // FASTQ: sequence lines until '+', then as many quality characters as bases
    while (peek(c)) { if (c == '+') { skip_line(); break; } take_line(seq); }
    size_t q = 0;
    while (q < seq.size() && peek(c)) q += skip_line();
    return true;
  }
 private:
  static constexpr size_t BUF = 1 << 20;
  bool fill() {
    if (eof_) return false;
    ssize_t r;
    if (gz_) r = gzread(gz_, buf_.get(), (unsigned)BUF);
    else { do r = ::read(fd_, buf_.get(), BUF); while (r < 0 && errno == EINTR); }
    if (r <= 0) { eof_ = true; return false; }
    pos_ = 0; end_ = (size_t)r;
    return true;
  }
  // the first character of the next line (false: end of file); an empty line gives '\n'
  bool peek(char& c) { if (pos_ == end_ && !fill()) return false; c = buf_[pos_]; return true; }
  // the rest of the current line is dropped; -> its length without the line end (and without a closing carriage return)
  size_t skip_line() {
    size_t n = 0; char last = 0;
    for (;;) {
      if (pos_ == end_ && !fill()) break;
      const char* b = buf_.get() + pos_;
      const char* nl = (const char*)memchr(b, '\n', end_ - pos_);
      const size_t m = nl ? (size_t)(nl - b) : end_ - pos_;
      if (m) last = b[m - 1];
      n += m; pos_ += m + (nl ? 1 : 0);
      if (nl) break;
    }
    return n - (n && last == '\r' ? 1 : 0);
  }
  // the rest of the current line goes behind `s`, blanks, tabs and carriage returns left out
  void take_line(std::string& s) {
    for (;;) {
      if (pos_ == end_ && !fill()) return;
      const char* b = buf_.get() + pos_;
      const char* nl = (const char*)memchr(b, '\n', end_ - pos_);
      const size_t m = nl ? (size_t)(nl - b) : end_ - pos_;
      unsigned low = 0;
      for (size_t i = 0; i < m; i++) low |= (unsigned)((unsigned char)b[i] <= ' ');      // (vectorised: a line with nothing but letters is appended as it is)
      if (!low) s.append(b, m);
      else { const size_t o = s.size(); s.resize(o + m); size_t n = o; for (size_t i = 0; i < m; i++) { const char c = b[i]; if (c != ' ' && c != '\t' && c != '\r') s[n++] = c; } s.resize(n); }
      pos_ += m + (nl ? 1 : 0);
      if (nl) return;
    }
  }
  gzFile gz_ = nullptr; int fd_ = -1; std::string path_; char hdr_ = 0; bool have_hdr_ = false, eof_ = false;
  std::unique_ptr<char[]> buf_; size_t pos_ = 0, end_ = 0;
};

}  // namespace kmxio
