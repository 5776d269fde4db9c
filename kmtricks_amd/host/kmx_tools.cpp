// kmx_tools.cpp -- `kmx dump`, `kmx aggregate` and `kmx combine`: the reader-side commands of kmtricks over kmx run directories
// (reference src/cli.cpp:648-776 flags; include/kmtricks/cmd.hpp:275-369 main_dump, 441-607 main_agg; text forms:
// io/kmer_file.hpp:140-148, io/hash_file.hpp:211-219, io/matrix_file.hpp:169-180, 293-304, io/pa_matrix_file.hpp:134-152,
// 267-285; sorted aggregation = a merge of the partitions' ascending files, io/kmer_file.hpp:171-290, matrix_file.hpp:307-460).
// combine: matrix.hpp:396-886 (MatrixMerger: the matrices of several runs that share a repartition, joined column block after
// column block), cmd.hpp:371-437, src/cli.cpp:669-700.
// Host-only file conversion: no GPU work here (nothing data-parallel is timed on this path).
#include <algorithm>
#include <filesystem>
#include <iostream>
#include <queue>
#include <sstream>
#include "kmx_io.hpp"

namespace fs = std::filesystem;
using namespace kmxio;

[[noreturn]] static void tdie(const std::string& msg) { std::cerr << "[error] " << msg << std::endl; std::exit(EXIT_FAILURE); }

static std::string kmer_string(const uint8_t* key, uint32_t k)
{ // Kmer::to_string: nucleotide i (from the first) = digit k-1-i, A0 C1 T2 G3 (kmer.hpp:797-810)
  std::string s(k, 'A');
  for (uint32_t i = 0; i < k; i++) { const uint32_t d = k - 1 - i; uint64_t w; memcpy(&w, key + 8 * (d >> 5), 8); s[i] = "ACTG"[(w >> ((d & 31) * 2)) & 3]; }
  return s;
}

// one kmtricks file, decoded: rows of `key_bytes` key + payload
struct KmFile {
  enum Kind { KMER, HASH, MATRIX, MATRIX_HASH, PA, PA_HASH } kind;
  uint32_t k = 0, key_bytes = 8, cols = 0, count_slots = 4, pa_bytes = 0, partition = 0, id = 0, hdr_count_slots = 0;
  std::vector<uint8_t> body;          // HASH files: re-packed to hash + count records
  size_t row_bytes() const { return kind == KMER || kind == HASH ? key_bytes + count_slots : kind == MATRIX || kind == MATRIX_HASH ? key_bytes + (size_t)cols * count_slots : key_bytes + pa_bytes; }
  size_t rows() const { return row_bytes() ? body.size() / row_bytes() : 0; }
  bool hashed() const { return kind == HASH || kind == MATRIX_HASH || kind == PA_HASH; }
};

static KmFile load(const std::string& path)
{
  std::vector<uint8_t> raw = slurp(path);
  if (raw.size() < 21 || rd<uint64_t>(&raw[0]) != MAGIC_BASE) tdie("Invalid file format: " + path);
  const uint64_t magic = rd<uint64_t>(&raw[13]);
  KmFile f;
  if (magic == MAGIC_KMER) { f.kind = KmFile::KMER; f.k = rd<uint32_t>(&raw[21]); f.key_bytes = rd<uint32_t>(&raw[25]) * 8; f.count_slots = rd<uint32_t>(&raw[29]); f.id = rd<uint32_t>(&raw[33]); f.partition = rd<uint32_t>(&raw[37]); f.body = body_of(raw, 41, magic, path); }
  else if (magic == MAGIC_HASH) { f.kind = KmFile::HASH; f.count_slots = 4; f.partition = rd<uint32_t>(&raw[29]); f.body = read_hash_records(path, nullptr); }
  else if (magic == MAGIC_MATRIX) { f.kind = KmFile::MATRIX; f.k = rd<uint32_t>(&raw[21]); f.key_bytes = rd<uint32_t>(&raw[25]) * 8; f.count_slots = 4; f.hdr_count_slots = rd<uint32_t>(&raw[29]); f.cols = rd<uint32_t>(&raw[33]); f.id = rd<uint32_t>(&raw[37]); f.partition = rd<uint32_t>(&raw[41]); f.body = body_of(raw, 45, magic, path); }   // (count_slots is the literal 1 in the header, the counts are 4 bytes: merge.hpp:264)
  else if (magic == MAGIC_MATRIX_HASH) { f.kind = KmFile::MATRIX_HASH; f.count_slots = rd<uint32_t>(&raw[21]); f.hdr_count_slots = f.count_slots; f.cols = rd<uint32_t>(&raw[25]); f.id = rd<uint32_t>(&raw[29]); f.partition = rd<uint32_t>(&raw[33]); f.body = body_of(raw, 37, magic, path); }
  else if (magic == MAGIC_PA) { f.kind = KmFile::PA; f.k = rd<uint32_t>(&raw[21]); f.key_bytes = rd<uint32_t>(&raw[25]) * 8; f.cols = rd<uint32_t>(&raw[29]); f.pa_bytes = rd<uint32_t>(&raw[33]); f.id = rd<uint32_t>(&raw[37]); f.partition = rd<uint32_t>(&raw[41]); f.body = body_of(raw, 45, magic, path); }
  else if (magic == MAGIC_PA_HASH) { f.kind = KmFile::PA_HASH; f.cols = rd<uint32_t>(&raw[21]); f.pa_bytes = rd<uint32_t>(&raw[25]); f.id = rd<uint32_t>(&raw[29]); f.partition = rd<uint32_t>(&raw[33]); f.body = body_of(raw, 37, magic, path); }
  else tdie("this file type doesn't support text conversion: " + path);
  if (f.key_bytes == 0 || f.key_bytes > 128 || (f.count_slots != 1 && f.count_slots != 2 && f.count_slots != 4)) tdie("Invalid file format: " + path);
  if (f.row_bytes() && f.body.size() % f.row_bytes() != 0) tdie("truncated file (its body is no whole number of rows): " + path);
  return f;
}

static void row_text(const KmFile& f, const uint8_t* row, bool no_count, std::string& out)
{
  if (f.hashed()) out += std::to_string(rd<uint64_t>(row)); else out += kmer_string(row, f.k);
  if (!no_count) {
    const uint8_t* p = row + f.key_bytes;
    if (f.kind == KmFile::PA || f.kind == KmFile::PA_HASH) { for (uint32_t i = 0; i < f.cols; i++) { out += ' '; out += ((p[i >> 3] >> (i & 7)) & 1) ? '1' : '0'; } }
    else {
      const uint32_t n = f.kind == KmFile::KMER || f.kind == KmFile::HASH ? 1 : f.cols;
      for (uint32_t i = 0; i < n; i++) { uint32_t c = 0; memcpy(&c, p + (size_t)i * f.count_slots, f.count_slots); out += ' '; out += std::to_string(c); }
    }
  }
  out += '\n';
}

struct Sink {
  std::ostream* os = &std::cout; std::ofstream file;
  explicit Sink(const std::string& path) { if (path != "stdout") { file.open(path, std::ios::binary); if (!file) tdie("Unable to write at " + path); os = &file; } }
  void put(const std::string& s) { os->write(s.data(), (std::streamsize)s.size()); }
};

static bool key_less(const uint8_t* a, const uint8_t* b, uint32_t key_bytes)
{ // most significant word first (kmer.hpp:262-268)
  for (int w = (int)key_bytes / 8 - 1; w >= 0; w--) { const uint64_t x = rd<uint64_t>(a + 8 * w), y = rd<uint64_t>(b + 8 * w); if (x != y) return x < y; }
  return false;
}

static int cmd_dump(int argc, char** argv)
{
  std::string input, output = "stdout";
  for (int i = 2; i < argc; i++) {
    const std::string a = argv[i];
    auto need = [&]() -> std::string { if (i + 1 >= argc) tdie("missing value for " + a); return argv[++i]; };
    if (a == "--input") input = need(); else if (a == "-o" || a == "--output") output = need(); else if (a == "--run-dir") need(); else if (a == "-v" || a == "--verbose" || a == "-t" || a == "--threads") need();
    else tdie("unknown option " + a);
  }
  if (input.empty()) tdie("--input is required");
  {   // histograms/<id>.hist: HistReader::write_as_text(stream, false) -- the unique counts (io/hist_file.hpp:143-171, cmd.hpp:351-360)
    std::ifstream probe(input, std::ios::binary); char h[21] = {0}; probe.read(h, 21);
    if (probe.gcount() == 21 && rd<uint64_t>((const uint8_t*)h + 13) == MAGIC_HIST) {
      const HistFile hf = read_hist_file(input);
      Sink out(output); std::string buf;
      buf += "@LOWER=" + std::to_string(hf.lower) + "\n@UPPER=" + std::to_string(hf.upper) + "\n@OOB_L=" + std::to_string(hf.oob_lu) + "\n@OOB_U=" + std::to_string(hf.oob_uu) + "\n";
      for (size_t i = 0; i < hf.u.size(); i++) buf += std::to_string(hf.lower + i) + " " + std::to_string(hf.u[i]) + "\n";
      out.put(buf);
      return 0;
    }
  }
  const KmFile f = load(input);
  Sink out(output); std::string buf;
  for (size_t r = 0; r < f.rows(); r++) { row_text(f, f.body.data() + r * f.row_bytes(), false, buf); if (buf.size() > (1u << 20)) { out.put(buf); buf.clear(); } }
  out.put(buf);
  return 0;
}

static int cmd_aggregate(int argc, char** argv)
{
  std::string dir, count, matrix, pa, format = "text", output = "stdout";
  bool sorted = false, cpr_in = false, cpr_out = false, no_count = false;
  for (int i = 2; i < argc; i++) {
    const std::string a = argv[i];
    auto need = [&]() -> std::string { if (i + 1 >= argc) tdie("missing value for " + a); return argv[++i]; };
    if (a == "--run-dir") dir = need(); else if (a == "--count") count = need(); else if (a == "--matrix") matrix = need(); else if (a == "--pa-matrix") pa = need();
    else if (a == "--format") format = need(); else if (a == "--sorted") sorted = true; else if (a == "--cpr-in") cpr_in = true; else if (a == "--cpr-out") cpr_out = true;
    else if (a == "--no-count") no_count = true; else if (a == "--output") output = need(); else if (a == "-v" || a == "--verbose" || a == "-t" || a == "--threads") need();
    else tdie("unknown option " + a);
  }
  if (dir.empty()) tdie("--run-dir is required");
  if (format != "text" && format != "bin") tdie("--format must be text or bin");
  if ((int)!count.empty() + (int)!matrix.empty() + (int)!pa.empty() != 1) tdie("exactly one of --count, --matrix, --pa-matrix is required");
  GatbConfig gc; if (!GatbConfig::load(dir + "/config_gatb/gatb.config", gc)) tdie("Unable to read at " + dir + "/config_gatb/gatb.config");
  std::vector<std::string> paths;
  for (uint32_t p = 0; p < gc.nb_partitions; p++) {
    std::string f;
    if (!count.empty()) {   // id:kmer|hash
      const size_t c = count.find(':'); if (c == std::string::npos) tdie("--count takes id:kmer|hash");
      const std::string id = count.substr(0, c), kind = count.substr(c + 1);
      f = dir + "/counts/partition_" + std::to_string(p) + "/" + id + (kind == "hash" ? ".hash" : (cpr_in ? ".kmer.lz4" : ".kmer"));
    } else if (!matrix.empty()) f = dir + "/matrices/matrix_" + std::to_string(p) + (matrix == "hash" ? ".count_hash" : (cpr_in ? ".count.lz4" : ".count"));
    else f = dir + "/matrices/matrix_" + std::to_string(p) + (pa == "hash" ? ".pa_hash" : (cpr_in ? ".pa.lz4" : ".pa"));
    if (fs::exists(f)) paths.push_back(f);
  }
  if (paths.empty()) tdie("No files found for these parameters.");
  std::vector<KmFile> files; for (auto& p : paths) files.push_back(load(p));
  const KmFile& f0 = files[0];
  for (auto& f : files) if (f.kind != f0.kind || f.row_bytes() != f0.row_bytes()) tdie("partition files of different shapes");
  if (sorted && f0.hashed()) sorted = false;            // (hash windows of the partitions are disjoint and ascending already)
  // row order: partition after partition, or one ascending stream (a merge of the partitions' ascending files)
  std::vector<std::pair<uint32_t, size_t>> order;
  if (!sorted) { for (uint32_t i = 0; i < files.size(); i++) for (size_t r = 0; r < files[i].rows(); r++) order.push_back({i, r}); }
  else {
    auto cmp = [&](const std::pair<uint32_t, size_t>& a, const std::pair<uint32_t, size_t>& b) {
      return key_less(files[b.first].body.data() + b.second * f0.row_bytes(), files[a.first].body.data() + a.second * f0.row_bytes(), f0.key_bytes); };
    std::priority_queue<std::pair<uint32_t, size_t>, std::vector<std::pair<uint32_t, size_t>>, decltype(cmp)> pq(cmp);
    for (uint32_t i = 0; i < files.size(); i++) if (files[i].rows()) pq.push({i, 0});
    while (!pq.empty()) { auto t = pq.top(); pq.pop(); order.push_back(t); if (t.second + 1 < files[t.first].rows()) pq.push({t.first, t.second + 1}); }
  }
  if (format == "text") {
    Sink out(output); std::string buf;
    for (auto& o : order) { row_text(f0, files[o.first].body.data() + o.second * f0.row_bytes(), no_count, buf); if (buf.size() > (1u << 20)) { out.put(buf); buf.clear(); } }
    out.put(buf);
  } else {
    if (output == "stdout") tdie("--format bin needs --output");
    Out out(output);
    switch (f0.kind) {   // the aggregated file carries partition 0 (KmerFileAggregator::write_as_bin and friends)
      case KmFile::KMER: out.base_header(cpr_out); out.put<uint64_t>(MAGIC_KMER); out.put<uint32_t>(f0.k); out.put<uint32_t>(f0.key_bytes / 8); out.put<uint32_t>(f0.count_slots); out.put<uint32_t>(0); out.put<uint32_t>(0); out.begin_body(); break;
      case KmFile::HASH: tdie("aggregated hash count files (--format bin) are not supported"); break;
      case KmFile::MATRIX: matrix_count_header(out, f0.k, f0.cols, 0, cpr_out); break;
      case KmFile::MATRIX_HASH: matrix_count_hash_header(out, f0.cols, 0, cpr_out); break;
      case KmFile::PA: matrix_pa_header(out, f0.k, f0.cols, 0, cpr_out); break;
      case KmFile::PA_HASH: matrix_pa_hash_header(out, f0.cols, 0, cpr_out); break;
    }
    for (auto& o : order) out.raw(files[o.first].body.data() + o.second * f0.row_bytes(), f0.row_bytes());
    out.close();
  }
  return 0;
}

// ---- kmx combine --fof <runs, one per line> --output <dir> [--cpr] ------------------------------------------------------
static std::string trim(const std::string& s) { const size_t a = s.find_first_not_of(" \t\r\n"); if (a == std::string::npos) return ""; return s.substr(a, s.find_last_not_of(" \t\r\n") - a + 1); }

static int cmd_combine(int argc, char** argv)
{
  std::string fof, output; bool cpr = false, compat = false;
  for (int i = 2; i < argc; i++) {
    const std::string a = argv[i];
    auto need = [&]() -> std::string { if (i + 1 >= argc) tdie("missing value for " + a); return argv[++i]; };
    if (a == "--fof") fof = need(); else if (a == "--output") output = need(); else if (a == "--cpr") cpr = true;
    else if (a == "--reference-compat") compat = true;      // kmx extension: reproduce PartitionMerger::next's dropped last row (below)
    else if (a == "-v" || a == "--verbose" || a == "-t" || a == "--threads") need(); else tdie("unknown option " + a);
  }
  if (fof.empty() || output.empty()) tdie("--fof and --output are required");
  std::vector<std::string> runs;
  { std::ifstream in(fof); if (!in) tdie("Unable to read at " + fof); for (std::string l; std::getline(in, l);) if (!trim(l).empty()) runs.push_back(trim(l)); }
  if (runs.empty()) tdie("--fof names no run");
  // mode and count format of the first run (cmd.hpp:379-399: the first line of options.txt, entries split at ',' and '=')
  std::string mode, cformat;
  { std::ifstream in(runs[0] + "/options.txt"); std::string line; std::getline(in, line); std::stringstream ss(line); std::string e;
    while (std::getline(ss, e, ',')) { const size_t q = e.find('='); if (q == std::string::npos) continue; const std::string key = trim(e.substr(0, q)), v = trim(e.substr(q + 1)); if (key == "mode") mode = v; else if (key == "count_format") cformat = v; } }
  if ((mode != "count" && mode != "pa") || (cformat != "kmer" && cformat != "hash")) tdie(runs[0] + ": matrix format not supported by 'kmtricks combine'.");
  const bool pa = mode == "pa", hashed = cformat == "hash";
  // the runs must share their repartition (matrix.hpp:717-733)
  const std::string rp = "/repartition_gatb/repartition.minimRepart";
  if (!fs::exists(runs[0] + rp)) tdie(runs[0] + ": not a kmtricks directory.");
  { std::vector<uint8_t> t0 = slurp(runs[0] + rp);
    for (size_t i = 1; i < runs.size(); i++) { if (!fs::exists(runs[i] + rp)) tdie(runs[i] + ": not a kmtricks directory."); if (slurp(runs[i] + rp) != t0) tdie(runs[0] + " and " + runs[i] + " are not mergeable."); } }
  // the new run directory (matrix.hpp:735-747)
  fs::create_directory(output); fs::create_directory(output + "/matrices"); fs::create_directory(output + "/repartition_gatb"); fs::create_directory(output + "/config_gatb");
  fs::copy(runs[0] + "/hash.info", output, fs::copy_options::recursive);
  fs::copy(runs[0] + "/config_gatb", output + "/config_gatb", fs::copy_options::recursive);
  fs::copy(runs[0] + "/repartition_gatb", output + "/repartition_gatb", fs::copy_options::recursive);
  fs::copy(runs[0] + "/options.txt", output, fs::copy_options::recursive);
  uint64_t nb_parts = 0;
  { std::vector<uint8_t> hi = slurp(runs[0] + "/hash.info"); if (hi.size() < 16) tdie(runs[0] + "/hash.info: Invalid file format."); nb_parts = rd<uint64_t>(&hi[8]); }      // (matrix.hpp:866-871)
  // kmtricks.fof: the runs' fofs one after the other; duplicate ids get the run number appended (matrix.hpp:817-864)
  { std::vector<std::vector<std::string>> lines(runs.size()); std::vector<std::string> ids; bool dup = false;
    for (size_t r = 0; r < runs.size(); r++) { std::ifstream in(runs[r] + "/kmtricks.fof"); for (std::string l; std::getline(in, l);) if (!l.empty()) { lines[r].push_back(l); ids.push_back(trim(l.substr(0, l.find(':')))); } }
    { std::vector<std::string> s2 = ids; std::sort(s2.begin(), s2.end()); dup = std::adjacent_find(s2.begin(), s2.end()) != s2.end(); }
    std::ofstream out(output + "/kmtricks.fof");
    for (size_t r = 0; r < runs.size(); r++) for (auto& l : lines[r]) {
      if (!dup) out << l << '\n';
      else { const size_t q = l.find(':'); out << trim(l.substr(0, q)) << "_" << r << ": " << (q == std::string::npos ? "" : l.substr(q + 1, l.find(':', q + 1) == std::string::npos ? std::string::npos : l.find(':', q + 1) - q - 1)) << '\n'; }
    } }
  for (uint64_t p = 0; p < nb_parts; p++) {
    // the partition's files: a run that still holds count files contributes each of them as a one-column matrix (its
    // counts/partition_<p>/ entries; listed by name here, the reference takes the directory's order), any other run the p-th
    // of its matrices/ entries sorted BY NAME -- matrix_10 before matrix_2, as in matrix.hpp:786-798
    std::vector<std::string> paths; std::vector<uint8_t> from_counts;
    for (auto& r : runs) {
      const std::string c0 = r + "/counts/partition_0";
      if (fs::exists(c0) && !fs::is_empty(c0)) {
        std::vector<std::string> kp; for (auto& e : fs::directory_iterator(r + "/counts/partition_" + std::to_string(p))) kp.push_back(e.path().string());
        std::sort(kp.begin(), kp.end());
        for (auto& x : kp) { paths.push_back(x); from_counts.push_back(1); }
      } else {
        std::vector<std::string> mp; for (auto& e : fs::directory_iterator(r + "/matrices")) mp.push_back(e.path().string());
        std::sort(mp.begin(), mp.end());
        if (p >= mp.size()) tdie(r + ": no matrix for partition " + std::to_string(p));
        paths.push_back(mp[p]); from_counts.push_back(0);
      }
    }
    std::vector<KmFile> files; for (auto& x : paths) files.push_back(load(x));
    // column blocks: a file's columns start where the previous file's end (PartitionMerger::init, matrix.hpp:514-532)
    std::vector<size_t> pos(files.size()); size_t total = 0;
    for (size_t i = 0; i < files.size(); i++) {
      const KmFile& f = files[i];
      const bool ok = pa ? (f.kind == (hashed ? KmFile::PA_HASH : KmFile::PA)) : (f.kind == (hashed ? KmFile::MATRIX_HASH : KmFile::MATRIX) || (!hashed && f.kind == KmFile::KMER));
      if (!ok || f.key_bytes != files[0].key_bytes) tdie(paths[i] + ": not a " + cformat + " " + mode + " matrix like the first run's");
      pos[i] = total; total += f.kind == KmFile::KMER ? 1 : f.cols;
    }
    const KmFile& last = files.back();
    std::string op = output + "/matrices/matrix_" + std::to_string(p) + (pa ? (hashed ? ".pa_hash" : ".pa") : (hashed ? ".count_hash" : ".count")) + (cpr ? ".lz4" : "");
    Out out(op);
    // header fields come from the LAST file (write_k_c .. write_h_p, matrix.hpp:632-680); a count file read as a matrix has its
    // id / partition / count_slots fields shifted by one (MatrixFileHeader::deserialize(stream, kasm), io/matrix_file.hpp:54-69)
    uint32_t h_cs = last.hdr_count_slots, h_id = last.id, h_part = last.partition;
    if (last.kind == KmFile::KMER) { h_cs = last.partition; h_id = last.count_slots; h_part = last.id; }
    out.base_header(cpr);
    if (!pa && !hashed) { out.put<uint64_t>(MAGIC_MATRIX); out.put<uint32_t>(last.k); out.put<uint32_t>((last.k + 31) / 32); out.put<uint32_t>(h_cs); out.put<uint32_t>((uint32_t)total); out.put<uint32_t>(h_id); out.put<uint32_t>(h_part); }
    else if (!pa) { out.put<uint64_t>(MAGIC_MATRIX_HASH); out.put<uint32_t>(h_cs); out.put<uint32_t>((uint32_t)total); out.put<uint32_t>(h_id); out.put<uint32_t>(h_part); }
    else if (!hashed) { out.put<uint64_t>(MAGIC_PA); out.put<uint32_t>(last.k); out.put<uint32_t>((last.k + 31) / 32); out.put<uint32_t>((uint32_t)total); out.put<uint32_t>((uint32_t)((total + 7) / 8)); out.put<uint32_t>(h_id); out.put<uint32_t>(h_part); }
    else { out.put<uint64_t>(MAGIC_PA_HASH); out.put<uint32_t>((uint32_t)total); out.put<uint32_t>((uint32_t)((total + 7) / 8)); out.put<uint32_t>(h_id); out.put<uint32_t>(h_part); }
    out.begin_body();
    // PartitionMerger::next (matrix.hpp:534-583): the smallest key of the queue starts a row, every file at that key adds its
    // columns.  In the reference a row whose first file leaves the queue EMPTY is not written (`if (m_queue.empty()) return
    // false` sits before the row is handed out): the last key of a partition is lost unless two files hold it.  kmx writes that
    // row -- combine(runA, runB) then equals one run over both sample sets -- and reproduces the reference's bytes only with
    // --reference-compat.
    const uint32_t kb = files[0].key_bytes;
    std::vector<size_t> cur(files.size(), 0);
    auto key_of = [&](size_t i) { return files[i].body.data() + cur[i] * files[i].row_bytes(); };
    auto cmp = [&](size_t a, size_t b) { return key_less(key_of(b), key_of(a), kb); };      // (min-heap)
    std::priority_queue<size_t, std::vector<size_t>, decltype(cmp)> q(cmp);
    for (size_t i = 0; i < files.size(); i++) if (files[i].rows()) q.push(i);
    const size_t data_bytes = pa ? (total + 7) / 8 : total * 4;
    std::vector<uint8_t> row(kb + data_bytes), obuf;
    auto add = [&](size_t i) {
      const KmFile& f = files[i]; const uint8_t* d = key_of(i) + kb;
      if (!pa) { const uint32_t n = f.kind == KmFile::KMER ? 1 : f.cols; for (uint32_t c = 0; c < n; c++) { uint32_t v = 0; memcpy(&v, d + (size_t)c * f.count_slots, f.count_slots); memcpy(&row[kb + (pos[i] + c) * 4], &v, 4); } }
      else for (uint32_t j = 0; j < f.cols; j++) if ((d[j >> 3] >> (j & 7)) & 1) row[kb + ((pos[i] + j) >> 3)] |= (uint8_t)(1u << ((pos[i] + j) & 7));      // (copy_pa_vec, matrix.hpp:605-614)
    };
    auto advance = [&](size_t i) { q.pop(); if (++cur[i] < files[i].rows()) q.push(i); };
    while (!q.empty()) {
      std::fill(row.begin() + kb, row.end(), 0);
      size_t e = q.top();
      memcpy(row.data(), key_of(e), kb);
      add(e); advance(e);
      if (q.empty() && compat) break;                              // (this row is lost in the reference: see above)
      while (!q.empty() && memcmp(key_of(q.top()), row.data(), kb) == 0) { e = q.top(); add(e); advance(e); }
      obuf.insert(obuf.end(), row.begin(), row.end());
      if (obuf.size() > (4u << 20)) { out.raw(obuf.data(), obuf.size()); obuf.clear(); }
    }
    out.raw(obuf.data(), obuf.size());
    out.close();
  }
  return 0;
}

int kmx_tools_main(int argc, char** argv)
{
  try {
    const std::string cmd = argv[1];
    if (cmd == "dump") return cmd_dump(argc, argv);
    if (cmd == "aggregate") return cmd_aggregate(argc, argv);
    if (cmd == "combine") return cmd_combine(argc, argv);
  } catch (const std::exception& e) { tdie(e.what()); }
  return -1;
}
