// kmx_pipeline.cpp -- `kmx pipeline ...`: the `kmtricks pipeline` command line, run-directory layout
// and plugin loading over libkmx (the MI355X engine).  Mirrors reference src/cli.cpp:117-382 (flags and
// defaults), include/kmtricks/kmdir.hpp:195-241 (directory tree), task.hpp:98-124, 170-225, 255-320,
// 367-392, 447-481, 690-743, 787-863 (what each stage reads and writes), task_scheduler.hpp:115-160, 251-417
// (stage order, --restrict-to), io/fof.hpp:39-43 (fof grammar), plugin_manager.hpp:38-113 (plugin symbols),
// howde_utils.hpp:133-187 (per-sample Bloom filter files).  All compute goes through the C ABI of
// include/kmx.h; this file parses, schedules, reads and writes files.
//
// Scheduling (the role of task_scheduler.hpp + task_pool.hpp): --gpus G shards, each with --gpu-workers count workers
// (a host thread + a libkmx context each: one worker's host work overlaps the other's kernels); samples go round-robin
// over the workers for split + count, partitions to shard (index mod G) for the merge.  The count lists stay in HBM
// between the two stages (a kmx_store per shard; partition p's list is written into the store of the shard that merges
// it, over xGMI when that is another GPU) -- the reference's count files exist only with --keep-tmp / --until count, or
// for the samples a full store turns away (the merge then takes both kinds in one batch).  A pool of -t host threads
// parses reads ahead of the devices, reads the count files of the NEXT merge batch into pinned memory while the
// current one merges (kmx_merge_host uploads on its own stream), and writes outputs behind them: matrix bodies come
// off the device in pieces into a ring of pinned buffers and go to their files with pwrite.
// Errors: message on stderr + exit(EXIT_FAILURE) (reference src/kmtricks.cpp:109-123).
#include <kmx.h>
#include <kmtricks/plugin.hpp>

#include <dlfcn.h>
#include <unistd.h>
#include <sys/resource.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <filesystem>
#include <iomanip>
#include <iostream>
#include <map>
#include <memory>
#include <random>
#include <regex>
#include <set>
#include <sstream>

#include "kmx_io.hpp"
#include "kmx_pool.hpp"
#include "kmx_repart.hpp"

namespace fs = std::filesystem;
using namespace kmxio;
using clk = std::chrono::steady_clock;

struct Sample { std::string id; std::vector<std::string> files; uint32_t hard_min; };

struct Opt {
  bool merge_only = false;      // `kmx merge --run-dir <dir>`: the merge module by itself over the count files of an existing run directory (src/cli.cpp:526-646)
  bool text = false;      // --mode <count-format>:<count|pa>:text (src/cli.cpp:151-157): rows as text lines, merge.hpp:288-316, 531-572
  std::string fof, dir, mode = "kmer:count:bin", until = "all", plugin, plugin_config, repart_from, repart_file, bf_format = "howdesbt", soft_min_path;
  double soft_f = 0.0; bool soft_float = false;      // --soft-min <fraction> (src/cli.cpp:234-240)
  uint32_t k = 31, hard_min = 2, soft_min = 1, rec_min = 1, share_min = 0, nb_parts = 0, msize = 10, bitw = 2, threads = 8, gpus = 1, gpu_workers = 2, per_call = 0;
  uint64_t bloom = 10000000, merge_batch_mb = 4096;
  double restrict_to = 1.0, focus = 0.5;
  std::vector<uint32_t> restrict_list;
  bool static_repart = false, keep_tmp = false, cpr = false, skip_pinfo = false, hist = false, no_resident = false;
};

// (a worker thread cannot unwind the others: print and leave without running destructors under them)
[[noreturn]] static void die(const std::string& msg) { std::cerr << "[error] " << msg << std::endl; std::cerr.flush(); _exit(EXIT_FAILURE); }
static double since(clk::time_point t) { return std::chrono::duration<double>(clk::now() - t).count(); }

static std::vector<Sample> parse_fof(const std::string& path, uint32_t default_hard_min)
{ // grammar `ID : path[ ; path...][ ! hardmin]` (io/fof.hpp:39-43, 126-134)
  std::ifstream in(path); if (!in) die("Unable to read at " + path);
  static const std::regex pat(R"((^[A-Za-z0-9_-]+)[\s]*:[\s]*([.A-Za-z0-9\/_\-; ]+)([\s]*![\s]*)?([0-9]+$)?)");
  std::vector<Sample> out; std::map<std::string, int> seen; std::string line;
  const fs::path base = fs::absolute(fs::path(path)).parent_path();
  while (std::getline(in, line)) {
    while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
    if (line.empty()) continue;
    std::smatch m;
    if (!std::regex_match(line, m, pat)) die("fof: invalid line: " + line);
    Sample s; s.id = m[1]; s.hard_min = m[4].matched ? (uint32_t)std::stoul(m[4]) : default_hard_min;
    if (seen[s.id]++) die("fof: duplicate id " + s.id);
    std::stringstream ss(m[2]); std::string f;
    while (std::getline(ss, f, ';')) {
      f.erase(0, f.find_first_not_of(" \t")); f.erase(f.find_last_not_of(" \t") + 1);
      if (f.empty()) continue;
      fs::path p(f); if (p.is_relative() && !fs::exists(p)) p = base / p;   // fixtures use paths relative to the fof
      s.files.push_back(p.string());
    }
    if (s.files.empty()) die("fof: no file for " + s.id);
    out.push_back(s);
  }
  if (out.empty()) die("fof: empty");
  return out;
}

static Opt parse_cli(int argc, char** argv)
{
  if (argc < 2 || (std::string(argv[1]) != "pipeline" && std::string(argv[1]) != "merge")) die("usage: kmx pipeline --file <fof> --run-dir <dir> [options] | kmx merge --run-dir <dir> [options] | kmx dump --input <file> [-o out] | kmx aggregate --run-dir <dir> --matrix kmer|hash ...  (see INTEGRATION.md)");
  Opt o;
  // `kmtricks merge` (src/cli.cpp:526-646): --run-dir, --partition-id, --soft-min, --recurrence-min, --share-min, --mode, --clear, --cpr,
  // -t; what the run was made with (k, partitions, Bloom size, minimizer size) is read back from its options.txt below
  o.merge_only = std::string(argv[1]) == "merge";
  if (o.merge_only) { o.until = "merge"; o.keep_tmp = true; o.no_resident = true; o.k = 0; }
  auto need = [&](int& i) -> std::string { if (i + 1 >= argc) die(std::string("missing value for ") + argv[i]); return argv[++i]; };
  auto num = [&](int& i) -> unsigned long { const std::string v = need(i); try { size_t n = 0; const unsigned long x = std::stoul(v, &n); if (n != v.size()) throw 1; return x; } catch (...) { die(std::string("bad number for ") + argv[i - 1] + ": " + v); } };
  auto real = [&](int& i) -> double { const std::string v = need(i); try { return std::stod(v); } catch (...) { die(std::string("bad number for ") + argv[i - 1] + ": " + v); } };
  for (int i = 2; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "--file") o.fof = need(i);
    else if (a == "--run-dir") o.dir = need(i);
    else if (a == "--kmer-size") o.k = num(i);
    else if (a == "--hard-min") o.hard_min = num(i);
    else if (a == "--mode") o.mode = need(i);
    else if (a == "--soft-min") {      // INT / STR / FLOAT (src/cli.cpp:228-248): an integer, a file with one threshold per sample, or a fraction
      const std::string v = need(i);
      if (fs::is_regular_file(v)) o.soft_min_path = v;
      else if (v.find('.') != std::string::npos) {
        // a fraction: thresholds "derived from the abundance histograms" (histogram.hpp:218-243).  The reference's computation pushes
        // its thresholds onto a vector it has already sized to one entry per sample, so the entries the merge reads -- the first N
        // -- are all 0: it behaves as --soft-min 0, implies --hist, and leaves merge_amin.txt with N zeros followed by the values it
        // computed.  Reproduced as it is (drop-in), with a note on stderr.
        double f = -1; try { size_t n = 0; f = std::stod(v, &n); if (n != v.size()) f = -1; } catch (...) { f = -1; }
        if (!(f >= 0.0 && f <= 1.0)) die("--abundance-min<float>: Not in range [0.0, 1.0] : " + v);      // (bcli's range check, src/cli.cpp:237)
        o.soft_f = f; o.soft_float = true; o.hist = true;
        fprintf(stderr, "[kmx] --soft-min %s: as in kmtricks 1.6.0 the thresholds computed from the histograms do not reach the merge (histogram.hpp:221-234 appends them "
                        "behind one zero per sample): every sample is merged with soft-min 0; merge_amin.txt holds the zeros and the computed values\n", v.c_str());
      }
      else { try { size_t n = 0; o.soft_min = (uint32_t)std::stoul(v, &n); if (n != v.size()) throw 1; } catch (...) { die("bad number for --soft-min: " + v); } }
    }
    else if (a == "--recurrence-min") o.rec_min = num(i);
    else if (a == "--share-min") o.share_min = num(i);
    else if (a == "--nb-partitions") o.nb_parts = num(i);
    else if (a == "--minimizer-size") o.msize = num(i);
    else if (a == "--minimizer-type") { if (num(i) != 0) die("--minimizer-type: only 0 (lexicographic) is supported"); }
    else if (a == "--repartition-type") { if (num(i) != 0) die("--repartition-type: only 0 (unordered) is supported"); }
    else if (a == "--bloom-size") o.bloom = (uint64_t)real(i);
    else if (a == "--bf-format") o.bf_format = need(i);
    else if (a == "--bitw") o.bitw = num(i);
    else if (a == "--until") o.until = need(i);
    else if (a == "--static-repart") o.static_repart = true;
    else if (a == "--repart-from") o.repart_from = need(i);
    else if (a == "--repart-file") o.repart_file = need(i);     // kmx extension: use this minimRepart table as is
    else if (a == "--restrict-to") { o.restrict_to = real(i); if (o.restrict_to < 0.05 || o.restrict_to > 1.0) die("--restrict-to must be in [0.05, 1.0]"); }
    else if (a == "--restrict-to-list") { std::stringstream ss(need(i)); std::string t; while (std::getline(ss, t, ',')) { try { o.restrict_list.push_back((uint32_t)std::stoul(t)); } catch (...) { die("--restrict-to-list: bad partition " + t); } } }
    else if (a == "--focus") { o.focus = real(i); if (o.focus < 0.0 || o.focus > 1.0) die("--focus must be in [0.0, 1.0]"); }
    else if (a == "--keep-tmp") o.keep_tmp = true;
    else if (a == "--clear" && o.merge_only) o.keep_tmp = false;      // clear the partition files once merged (src/cli.cpp:640-642)
    else if (a == "--partition-id" && o.merge_only) { const std::string v = need(i); if (v != "-1") { try { o.restrict_list.assign(1, (uint32_t)std::stoul(v)); } catch (...) { die("bad number for --partition-id: " + v); } } }
    else if (a == "--cpr") o.cpr = true;
    else if (a == "--hist") o.hist = true;                      // histograms/<id>.hist (src/cli.cpp:207-209)
    else if (a == "--plugin") o.plugin = need(i);
    else if (a == "--plugin-config") o.plugin_config = need(i);
    else if (a == "-t" || a == "--threads") o.threads = num(i);
    else if (a == "--gpus") o.gpus = num(i);                    // kmx extension: samples / partitions shard round-robin over this many GPUs
    else if (a == "--samples-per-call") o.per_call = num(i);      // kmx extension: whole samples a count worker hands the GPU in one call (0: the default)
    else if (a == "--gpu-workers") o.gpu_workers = num(i);      // kmx extension: count workers (host thread + context) per shard
    else if (a == "--no-resident") o.no_resident = true;        // kmx extension: count lists go through count files (as with --keep-tmp) instead of staying in HBM
    else if (a == "--merge-batch-mb") o.merge_batch_mb = num(i); // kmx extension: count-list bytes per merge batch and GPU
    else if (a == "--skip-partiinfo") o.skip_pinfo = true;      // kmx extension: do not write superkmers/<id>/PartiInfoFile (6 MB of text per sample)
    else if (a == "-v" || a == "--verbose") need(i);
    else die("unknown option " + a);
  }
  if (o.merge_only) {
    // the run directory must be one (kmtricks.fof marks it, src/cli.cpp:106-115); its options.txt (cmd/all.hpp:85-125, one line of
    // key=value pairs that both kmtricks and kmx write) says what the counts were made with
    if (o.dir.empty()) die("--run-dir is required");
    if (!fs::exists(o.dir + "/kmtricks.fof")) die(o.dir + " is not a kmtricks runtime directory.");
    o.fof = o.dir + "/kmtricks.fof";
    std::string opt; { std::ifstream f(o.dir + "/options.txt"); std::getline(f, opt); }
    auto val = [&](const std::string& key) -> std::string {
      const size_t at = opt.find(" " + key + "="); if (at == std::string::npos) return "";
      const size_t b = at + key.size() + 2, e = opt.find(',', b); return opt.substr(b, e == std::string::npos ? std::string::npos : e - b);
    };
    auto uval = [&](const std::string& key, uint64_t dflt) -> uint64_t { const std::string v = val(key); if (v.empty()) return dflt; try { return (uint64_t)std::stod(v); } catch (...) { return dflt; } };
    if (o.k == 0) o.k = (uint32_t)uval("kmer_size", 31);
    if (o.nb_parts == 0) o.nb_parts = (uint32_t)uval("nb_parts", 0);
    o.msize = (uint32_t)uval("minim_size", o.msize);
    o.bloom = uval("bloom_size", o.bloom);
    o.bitw = (uint32_t)uval("bwidth", o.bitw);
    if (!o.cpr) o.cpr = uval("lz4", 0) != 0;
    if (o.nb_parts == 0) {      // (no options.txt: the partitions are the directories that are there)
      while (fs::exists(o.dir + "/counts/partition_" + std::to_string(o.nb_parts))) o.nb_parts++;
      if (o.nb_parts == 0) die(o.dir + " holds no counts/partition_<p>.");
    }
  } else {
  if (o.fof.empty() || o.dir.empty()) die("--file and --run-dir are required");
  if (fs::exists(o.dir)) die("--run-dir already exists: " + o.dir);                  // src/cli.cpp:101-104
  }
  if (o.k < 8 || o.k > 127) die("--kmer-size must be in [8, 127]: the reference's default KMER_LIST \"32 64 96 128\" (keys of one to four 64-bit words, loop_executor.hpp:47-63)");
  if (o.msize < 4 || o.msize > 15 || o.msize >= o.k) die("--minimizer-size must be in [4, 15] and < k");
  {   // the four text modes: the same merge, another writer (the mode string is the :bin one from here on)
    static const char* tmodes[] = {"kmer:count:text", "kmer:pa:text", "hash:count:text", "hash:pa:text"};
    for (const char* m : tmodes) if (o.mode == m) { o.text = true; o.mode = o.mode.substr(0, o.mode.size() - 4) + "bin"; break; }
  }
  static const char* modes[] = {"kmer:count:bin", "kmer:pa:bin", "hash:count:bin", "hash:pa:bin", "hash:bf:bin", "hash:bfc:bin", "hash:bft:bin"};
  if (std::find_if(std::begin(modes), std::end(modes), [&](const char* m) { return o.mode == m; }) == std::end(modes))
    die("--mode " + o.mode + " is not supported (kmer:{count,pa}:{bin,text}, hash:{count,pa}:{bin,text}, hash:{bf,bfc,bft}:bin)");
  static const char* untils[] = {"all", "repart", "superk", "count", "merge"};
  if (std::find_if(std::begin(untils), std::end(untils), [&](const char* m) { return o.until == m; }) == std::end(untils)) die("bad --until");
  if (o.bf_format != "howdesbt") die("--bf-format " + o.bf_format + " is not supported (howdesbt)");
  if (o.nb_parts > 65535) die("--nb-partitions too large");
  if ((o.mode == "hash:bf:bin" || o.mode == "hash:bft:bin") && (o.restrict_to != 1.0 || !o.restrict_list.empty())) die("--mode bf|bft requires all partitions.");   // cmd/all.hpp:137-143
  if (o.mode == "hash:bfc:bin" && (o.bitw < 1 || o.bitw > 32)) die("--bitw must be in [1, 32]");
  if (o.threads == 0) o.threads = 1;
  if (o.gpus == 0) o.gpus = 1;
  if (o.gpu_workers == 0) o.gpu_workers = 1;
  if (o.gpu_workers > 16) o.gpu_workers = 16;
  return o;
}

static void chk(kmx_ctx* c, int rc, const char* what) { if (rc != KMX_OK) die(std::string(what) + ": " + kmx_last_error(c)); }

// ---- plugin (plugin_manager.hpp:38-113) --------------------------------------------------------------
struct Plugin {
  void* h = nullptr; km::IMergePlugin* (*create)() = nullptr; void (*destroy)(km::IMergePlugin*) = nullptr;
  void load(const std::string& path, uint32_t k) {
    h = dlopen(path.c_str(), RTLD_LAZY); if (!h) die(std::string("plugin: ") + dlerror());
    auto use_template = (int (*)())dlsym(h, "use_template"); if (!use_template) die("plugin: use_template() missing");
    const std::string sym = "create" + std::to_string(use_template() ? (k < 32 ? 32 : 64) : 0);
    create = (km::IMergePlugin * (*)()) dlsym(h, sym.c_str()); if (!create) die("plugin: " + sym + "() missing");
    destroy = (void (*)(km::IMergePlugin*))dlsym(h, "destroy"); if (!destroy) die("plugin: destroy() missing");
  }
};

// what SuperKStorageWriter::SaveInfoFile reports for a partition stream (io/superk_storage.hpp:205-225, 328-340; Appendix B-4 of
// SURVEY.md): the info file is saved before the final flush, so it holds the k-mers since the last full block and the bytes
// of the blocks flushed so far
static void superk_info_numbers(const uint8_t* s, uint64_t len, uint32_t k, uint64_t* kmers_pending, uint64_t* bytes_flushed)
{
  uint64_t pos = 0, buf = 0, km = 0, flushed = 0, km_total_in_flushed_reset = 0;
  (void)km_total_in_flushed_reset;
  while (pos < len) {
    const uint32_t n = s[pos]; const uint64_t nb = ((uint64_t)k + n - 1 + 3) / 4;
    if (buf + nb + 1 > 32768) { flushed += buf + 4; buf = 0; km = 0; }
    buf += nb + 1; km += n; pos += 1 + nb;
  }
  *kmers_pending = km; *bytes_flushed = flushed;
}

struct Stage { double read = 0, split = 0, count = 0, merge_io = 0, merge = 0, format = 0, repart = 0, setup_wall = 0, count_wall = 0, merge_wall = 0; std::atomic<uint64_t> bases{0}, kmers{0}, merge_recs{0}, count_calls{0}; };

int run(int argc, char** argv)
{
  const auto t0 = clk::now();
  if (getenv("KMX_TRACE")) { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); fprintf(stderr, "[kmx epoch] pipeline starts %.6f\n", (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec); }
  Opt o = parse_cli(argc, argv);
  std::vector<Sample> samples = parse_fof(o.fof, o.hard_min);
  const uint32_t N = (uint32_t)samples.size(), kw = (o.k + 31) / 32;
  const bool hash_mode = o.mode.rfind("hash:", 0) == 0;
  const std::string what = o.mode.substr(o.mode.find(':') + 1, o.mode.rfind(':') - o.mode.find(':') - 1);   // count|pa|bf|bfc|bft
  if (o.cpr && hash_mode && !getenv("KMX_QUIET")) std::cerr << "[kmx pipeline] --cpr: .hash count files stay uncompressed (their .p4 form needs TurboPFor); matrix bodies are lz4 frames\n";

  // ---- number of partitions (task.hpp:108-115): 0 = gatb's ConfigurationAlgorithm sizes it from the estimated volume,
  // memory and open-file limits of the host (system dependent); this build uses volume / 2^30 k-mers, at least 4 ----
  std::vector<std::string> all_files; uint64_t in_bytes = 0;
  if (!o.merge_only) for (auto& s : samples) for (auto& f : s.files) { all_files.push_back(f); std::error_code ec; const auto sz = fs::file_size(f, ec); if (ec) die("Unable to read at " + f); in_bytes += sz * (f.size() > 3 && f.substr(f.size() - 3) == ".gz" ? 4 : 1); }
  if (o.nb_parts == 0) { uint64_t p = in_bytes / (1ull << 30) + 1; o.nb_parts = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(p, 4), 2000); }
  if (o.nb_parts < 4 && !o.merge_only) o.nb_parts = 4;                                 // task.hpp:112-113
  const uint32_t P = o.nb_parts;
  for (uint32_t p : o.restrict_list) if (p >= P) die("Ask to process part " + std::to_string(p) + " but nb_partitions is " + std::to_string(P));   // task_scheduler.hpp:152-158

  // ---- run directory (kmdir.hpp:195-241) ----
  const std::string root = fs::absolute(o.dir).string();
  for (const char* d : {"", "/superkmers", "/counts", "/matrices", "/filters", "/histograms", "/merge_infos", "/howde_index",
                        "/partition_infos", "/fpr", "/plugin_output", "/repartition_gatb", "/config_gatb"})
    fs::create_directories(root + d);
  if (!o.merge_only) {
  fs::copy_file(o.fof, root + "/kmtricks.fof");
  { std::ofstream b(root + "/build_infos.txt"); b << "kmx (MI355X-native kmtricks pipeline), libkmx ABI " << kmx_version() << "\n"; }
  { std::ofstream f(root + "/options.txt");   // cmd/all.hpp:85-125: `kmtricks combine` re-parses mode= from this line
    f << "Options: dir=" << root << ", verbosity=info, nb_threads=" << o.threads << ", fof=" << o.fof << ", kmer_size=" << o.k << ", c_ab_min=" << o.hard_min
      << ", m_ab_min=" << o.soft_min << ", r_min=" << o.rec_min << ", m_ab_min_path=" << o.soft_min_path << ", m_ab_min_f=" << o.soft_f << ", m_ab_float=" << o.soft_float << ", save_if=" << o.share_min << ", minim_size=" << o.msize
      << ", minim_type=0, repart_type=0, nb_parts=" << P << ", bloom_size=" << o.bloom << ", keep_tmp=" << o.keep_tmp << ", lz4=" << o.cpr << ", kff=0, hist=" << o.hist << ", static_repart=" << o.static_repart
      << ", focus=" << o.focus << ", restrict_to=" << o.restrict_to << ", bwidth=" << o.bitw << ", bam_exclude_refs=, bam_include_flags=0, bam_exclude_flags=0, mode=" << what      // (mode_to_str: count | pa | bf | bfc; cmd/all.hpp:119)
      << ", format=" << (o.text ? "text" : "bin") << ", bf_format=" << o.bf_format << ", count_format=" << (hash_mode ? "hash" : "kmer") << ", until=" << o.until << "\n"; }
  }
  for (uint32_t p = 0; p < P; p++) fs::create_directories(root + "/counts/partition_" + std::to_string(p));
  HashWindow hw(o.bloom, P, o.msize);
  if (o.merge_only) {      // the windows the hash counts were made with: hash.info's (hash.hpp:52-60: bloom size, partitions, window bits, ...)
    std::ifstream hi(root + "/hash.info", std::ios::binary); uint64_t h4[4] = {0, 0, 0, 0};
    if (hi.read(reinterpret_cast<char*>(h4), 32) && h4[1] == P && h4[0]) hw = HashWindow(h4[0], P, o.msize);
    for (uint32_t p = 0; p < P; p++) for (auto& sm : samples) { std::error_code ec; const auto sz = fs::file_size(root + "/counts/partition_" + std::to_string(p) + "/" + sm.id + (o.mode.rfind("hash:", 0) == 0 ? ".hash" : (o.cpr ? ".kmer.lz4" : ".kmer")), ec); if (!ec) in_bytes += sz; }
  }
  if (!o.merge_only) hw.save(root + "/hash.info");                                    // task.hpp:98-124
  if (!o.merge_only) { GatbConfig gc; gc.kmer_size = o.k; gc.minim_size = o.msize; gc.nb_cores = o.threads; gc.nb_partitions = P; gc.nb_banks = (uint16_t)std::min<size_t>(all_files.size(), 65535);
    gc.est_seq_total = in_bytes; gc.save(root + "/config_gatb/gatb.config"); }

  // ---- devices and host threads ----
  std::vector<kmx_ctx*> gpu;
  // G shards (--gpus beyond the devices present: the extra shards share the devices round-robin), NW = G x --gpu-workers count
  // workers: worker w belongs to shard w mod G and runs on that shard's device; the merge of shard g runs on worker g's context
  const int ndev = std::max(1, kmx_device_count());
  const uint32_t G = o.gpus, NW = G * o.gpu_workers;
  for (uint32_t w = 0; w < NW; w++) { kmx_ctx* x = nullptr; if (kmx_create((int)((w % G) % (uint32_t)ndev), &x) != KMX_OK) die(kmx_last_error(nullptr)); gpu.push_back(x); }
  // count lists resident in HBM between count and merge: one store per shard.  Not with --keep-tmp / --until count (the count
  // files are the product then) nor with --no-resident.
  const bool resident_mode = !o.keep_tmp && !o.no_resident && (o.until == "all" || o.until == "merge");
  std::vector<kmx_store*> stores;
  if (resident_mode) {
    for (uint32_t g = 0; g < G; g++) {
      const int dev = (int)(g % (uint32_t)ndev);
      uint64_t fr = 0, tot = 0; kmx_device_memory(dev, &fr, &tot);
      const uint32_t on_dev = (G - (uint32_t)dev + (uint32_t)ndev - 1) / (uint32_t)ndev;      // shards that share this device
      uint64_t lim = tot / 10 * 6 / std::max(1u, on_dev);
      if (const char* e = getenv("KMX_STORE_LIMIT_MB")) lim = (uint64_t)std::max(0L, atol(e)) << 20;      // (for the tests of the mixed merge: lists + count files)
      kmx_store* st_ = nullptr;
      if (kmx_store_create(dev, std::max<uint64_t>(lim, 1), &st_) != KMX_OK) die(kmx_last_error(nullptr));
      stores.push_back(st_);
    }
  }
  // (round 6) the devices' memory handed out once and given back, on threads of their own, while the first samples are read: a fresh
  // box clears HBM the first time it is allocated -- 8 ms a store chunk, 18 ms a row arena, inside the count and merge calls
  // (kmx_device_warm; KMX_WARM_GB=0: never)
  std::vector<std::thread> warmers;
  static volatile int warm_stop = 0;      // (raised when the count stage ends: the merge stage's allocations must not meet the warmers')
  struct WarmJoin { std::vector<std::thread>& v; ~WarmJoin() { warm_stop = 1; for (auto& t : v) if (t.joinable()) t.join(); } } warm_join{warmers};
  if (resident_mode) {
    const char* e = getenv("KMX_WARM_GB");
    const uint64_t gbs = e ? (uint64_t)std::max(0L, atol(e)) : 64;
    const uint32_t nd = std::min<uint32_t>(G, (uint32_t)ndev);
    if (gbs) for (uint32_t d = 0; d < nd; d++) warmers.emplace_back([d, gbs]() { (void)kmx_device_warm((int)d, gbs << 30, &warm_stop); });
  }
  // how a count worker on one GPU fills the store of another: peer access is asked for here, once per ordered pair of devices in
  // use ("p2p": copies over xGMI; "staged": through host memory); the summary line says which
  uint32_t peer_pairs = 0, peer_direct = 0;
  if (resident_mode) {
    const uint32_t nd = std::min<uint32_t>(G, (uint32_t)ndev);
    for (uint32_t a = 0; a < nd; a++) for (uint32_t b = 0; b < nd; b++) if (a != b) { peer_pairs++; if (kmx_peer_access((int)a, (int)b) == 1) peer_direct++; }
  }
  struct CtxGuard { std::vector<kmx_ctx*>& v; std::vector<kmx_store*>& s; ~CtxGuard() { for (auto x : s) kmx_store_destroy(x); for (auto x : v) kmx_destroy(x); } } ctx_guard{gpu, stores};
  // where sample i's count list of partition p is: resident (res_flag[i]) -> res_lists[i * P + p], else its count file
  std::vector<kmx_list> res_lists(resident_mode ? (size_t)samples.size() * o.nb_parts : 0, kmx_list{nullptr, 0});
  std::vector<uint8_t> res_flag(samples.size(), 0);
  Pool pool(o.threads);
  Stage st;
  const char* trace = getenv("KMX_TRACE");
  const auto t_start = clk::now();
  auto tlog = [&](uint32_t g, const char* what_, uint64_t id) { if (trace) fprintf(stderr, "[kmx trace] %.6f gpu=%u %s %llu\n", since(t_start), g, what_, (unsigned long long)id); };

  // ---- repartition (task.hpp:170-225) ----
  std::vector<uint16_t> table;
  const std::string rpath = root + "/repartition_gatb/repartition.minimRepart";
  if (!o.merge_only) {
    const auto t_rep = clk::now();
    if (!o.repart_file.empty() || !o.repart_from.empty()) {
      uint16_t np = 0;
      if (!o.repart_from.empty()) {   // check_repart_compatibility (task.hpp:136-147)
        GatbConfig fc;
        if (GatbConfig::load(o.repart_from + "/config_gatb/gatb.config", fc)) {
          if (fc.kmer_size != o.k) die("Unable to use repartition from " + o.repart_from + ", kmer sizes differ.");
          if (fc.minim_size != o.msize) die("Unable to use repartition from " + o.repart_from + ", minimizer sizes differ.");
          if (fc.nb_partitions != P) die("Unable to use repartition from " + o.repart_from + ", numbers of partitions differ.");
        }
      }
      table = read_repartition(o.repart_file.empty() ? o.repart_from + "/repartition_gatb/repartition.minimRepart" : o.repart_file, &np);
      if (np != P || table.size() != (1ULL << (2 * o.msize))) die("repartition table does not match --nb-partitions / --minimizer-size");
    } else if (o.static_repart) {
      table = repart_static(o.msize, P);
    } else {
      // sampled repartition (gatb RepartitionAlgorithm.cpp:395-496): from every bank (file), the reads up to the one that
      // brings the super-k-mers seen past the sample size; their kx-mers per minimizer (counted on the GPU) are balanced
      // over the partitions by Repartitor::computeDistrib.  Sample size: max(1 % of the bank's estimated reads, 100000)
      // with several banks, max(5 %, 1000000) with one -- the estimate is gatb's, from file sizes, and only exceeds the
      // floor for banks of more than 10 M reads; this build uses the floor.
      const uint64_t nm = 1ULL << (2 * o.msize);
      std::vector<uint64_t> kx(nm, 0);
      std::vector<uint16_t> zero(nm, 0);
      const uint64_t budget = all_files.size() > 1 ? 100000ULL : 1000000ULL;
      kmx_superk_stats S{}; S.minim_kxmers = kx.data();
      for (const std::string& f : all_files) {
        SeqReader rd(f); std::string seq, bases; std::vector<uint64_t> offs(1, 0);
        uint64_t seen = 0; bool done = false;
        auto flush = [&]() {
          if (offs.size() <= 1) return;
          uint64_t used = 0, nsk = 0;
          chk(gpu[0], kmx_superk_sample(gpu[0], bases.data(), offs.data(), offs.size() - 1, o.k, o.msize, budget - seen, &S, &used, &nsk), "kmx_superk_sample");
          seen += nsk;
          if (used < offs.size() - 1 || seen > budget) done = true;
          bases.clear(); offs.assign(1, 0);
        };
        while (!done && rd.next(seq)) {
          bases += seq; offs.push_back(bases.size()); st.bases += seq.size();
          if (offs.size() > 20000 || bases.size() > (64u << 20)) flush();
        }
        if (!done) flush();
      }
      table = repart_from_kxmers(kx.data(), nm, P);
    }
    write_repartition(rpath, (uint16_t)P, table);
    if (o.msize <= 12) {   // minimizers/minimizers.<p>: every m-mer assigned to partition p, one per line (task.hpp:160-168, repartition.hpp:116-124)
      fs::create_directories(root + "/minimizers");
      std::vector<std::string> buf(P);
      std::string mm(o.msize, 'A');
      for (uint64_t v = 0; v < table.size(); v++) {
        uint64_t t = v;
        for (int i = (int)o.msize - 1; i >= 0; i--) { mm[i] = "ACTG"[t & 3]; t >>= 2; }   // Mmer::to_string (kmer.hpp:115-127)
        if (table[v] >= P) die("repartition table names partition " + std::to_string(table[v]));
        std::string& b = buf[table[v]]; b += mm; b += '\n';
      }
      pool.for_each(P, [&](size_t p) { Out f(root + "/minimizers/minimizers." + std::to_string(p)); f.raw(buf[p].data(), buf[p].size()); f.close(); });
    }
    st.repart = since(t_rep);
  }
  if (o.until == "repart") return 0;

  // ---- the partitions to process (task_scheduler.hpp:121-160) ----
  std::vector<uint32_t> plist = o.restrict_list;
  if (plist.empty()) {
    for (uint32_t p = 0; p < P; p++) plist.push_back(p);
    if (o.restrict_to != 1.0) {
      std::random_device rdev; std::mt19937 gen(rdev()); std::shuffle(plist.begin(), plist.end(), gen);
      size_t n = (size_t)(P * o.restrict_to); if (n == 0) n = 1;
      plist.resize(n);
    }
  }
  std::vector<uint8_t> selected(P, 0); for (uint32_t p : plist) selected[p] = 1;
  const bool restricted = plist.size() != P;
  // --hist: the sample's abundance histogram comes off the device after its count calls (kmx_hist_reset / kmx_hist_read) and goes
  // to histograms/<id>.hist as KHist(i, k, 1, 255) leaves it (task_scheduler.hpp:54-57, 103)
  std::vector<std::vector<uint64_t>> hist_u(o.soft_float ? N : 0);      // --soft-min <fraction>: every sample's unique bins 1..255, [255] = its distinct k-mers
  auto save_hist = [&](kmx_ctx* c, uint32_t si) {
    auto ub = std::make_shared<std::vector<uint64_t>>(255), tb = std::make_shared<std::vector<uint64_t>>(255), ex = std::make_shared<std::vector<uint64_t>>(6);
    chk(c, kmx_hist_read(c, 1, 255, ub->data(), tb->data(), ex->data(), ex->data() + 4), "kmx_hist_read");
    if (o.soft_float) { hist_u[si] = *ub; hist_u[si].push_back((*ex)[4]); }
    chk(c, kmx_hist_off(c), "kmx_hist_off");
    const std::string path = root + "/histograms/" + samples[si].id + ".hist"; const uint32_t k = o.k;
    return pool.submit([=]() { try { write_hist_file(path, k, si, 1, 255, ub->data(), tb->data(), ex->data(), ex->data() + 4); } catch (const std::exception& e) { die(e.what()); } });
  };

  auto report = [&]() {
    fprintf(stderr, "[kmx pipeline] {\"samples\": %u, \"partitions\": %u, \"gpus\": %u, \"threads\": %u, \"bases\": %llu, \"kmers\": %llu, \"merge_records\": %llu, "
                    "\"repart_s\": %.4f, \"read_s\": %.4f, \"superk_s\": %.4f, \"count_s\": %.4f, \"merge_io_s\": %.4f, \"merge_s\": %.4f, \"format_s\": %.4f, "
                    "\"gpu_workers\": %u, \"resident_samples\": %u, \"resident_count_calls\": %llu, \"devices\": %d, \"peer_pairs\": %u, \"peer_pairs_direct\": %u, \"setup_wall_s\": %.4f, \"count_wall_s\": %.4f, \"merge_wall_s\": %.4f, \"total_s\": %.4f}\n",
            N, P, G, o.threads, (unsigned long long)st.bases.load(), (unsigned long long)st.kmers.load(), (unsigned long long)st.merge_recs.load(),
            st.repart, st.read, st.split, st.count, st.merge_io, st.merge, st.format,
            o.gpu_workers, (unsigned)std::count(res_flag.begin(), res_flag.end(), (uint8_t)1), (unsigned long long)st.count_calls.load(), ndev, peer_pairs, peer_direct, st.setup_wall, st.count_wall, st.merge_wall, since(t0));
  };
  auto count_path = [&](uint32_t p, uint32_t si) {
    return root + "/counts/partition_" + std::to_string(p) + "/" + samples[si].id + (hash_mode ? ".hash" : (o.cpr ? ".kmer.lz4" : ".kmer"));
  };

  // matrix bodies leave the device in pieces: a ring of pinned buffers shared by the shards (KMX_OUT_RING_MB, default 2048, in
  // pieces of 32 MB); a piece is handed to the pool (pwrite at its place in the file) and comes back to the ring when written --
  // what is pending is bounded in bytes, whatever the cohort.  Page-locking memory is slow (~10 ms per piece, 0.6 s for the ring:
  // the whole merge stage of the 1000 x 1 Mbp cohort, as the first runs of round 3 showed): a thread of its own fills the ring
  // while the samples are counted.
  // Page-locked memory is asked for by readers, writers and workers alike -- and a thread's FIRST call into the HIP runtime costs it
  // ~28 ms under the runtime's lock (its per-thread state), during which every other thread's launches and copies wait: 24 readers
  // pinning their first read buffer were 24 x 28 ms of stalled count calls (0.45 s of the 2.6 s count stage of 1000 x 5 Mbp; the
  // stage clocks of KMX_TRACE=1 showed them).  All page-locking therefore goes through ONE thread that has paid that once.
  struct PinServer {
    struct Req { size_t bytes; void* p = nullptr; bool done = false; };
    std::mutex m; std::condition_variable cv_req, cv_done; std::deque<Req*> q, q_low; bool stop = false; std::thread th;      // (q_low: the output ring's pieces asked for ahead -- behind whatever somebody waits for)
    PinServer() { th = std::thread([this]() { run(); }); }
    void run() {
      std::unique_lock<std::mutex> lk(m);
      for (;;) {
        cv_req.wait(lk, [&]() { return stop || !q.empty() || !q_low.empty(); });
        if (q.empty() && q_low.empty()) return;
        Req* r;
        if (!q.empty()) { r = q.front(); q.pop_front(); } else { r = q_low.front(); q_low.pop_front(); }
        lk.unlock();
        void* p = kmx_alloc_pinned(r->bytes);
        lk.lock();
        r->p = p; r->done = true; cv_done.notify_all();
      }
    }
    void* alloc(size_t bytes, bool ahead = false) {
      Req r{bytes};
      std::unique_lock<std::mutex> lk(m);
      (ahead ? q_low : q).push_back(&r); cv_req.notify_one();
      cv_done.wait(lk, [&]() { return r.done; });
      return r.p;
    }
    ~PinServer() { { std::lock_guard<std::mutex> lk(m); stop = true; } cv_req.notify_one(); if (th.joinable()) th.join(); }
  };
  static PinServer pins;      // (static: alive until the process leaves, whatever the order the pools below are destroyed in)
  struct Ring {
    std::mutex m; std::condition_variable cv; std::vector<uint8_t*> free_; size_t made = 0, cap = 64, bytes = (size_t)32 << 20;
    uint8_t* get() {
      std::unique_lock<std::mutex> lk(m);
      for (;;) {
        if (!free_.empty()) { uint8_t* p = free_.back(); free_.pop_back(); return p; }
        if (made < cap) { made++; lk.unlock(); uint8_t* p = (uint8_t*)pins.alloc(bytes); if (!p) die("pinned host allocation failed"); return p; }
        cv.wait(lk);
      }
    }
    void put(uint8_t* p) { { std::lock_guard<std::mutex> lk(m); free_.push_back(p); } cv.notify_one(); }
    void prefill(const std::atomic<bool>& stop) {      // (pieces the writers will want, made ahead of them)
      for (;;) {
        { std::lock_guard<std::mutex> lk(m); if (made >= cap || stop.load()) return; made++; }
        uint8_t* p = (uint8_t*)pins.alloc(bytes, true);
        if (!p) { std::lock_guard<std::mutex> lk(m); made--; return; }
        put(p);
      }
    }
    ~Ring() { for (auto p : free_) kmx_free_pinned(p); }
  } ring;
  // (round 4) A piece as large as a partition's matrix is expected to be (input bytes x 3 / partitions: count rows of a cohort;
  // 32 MB .. 512 MB), 18 of them (2 .. 9 GB; a file's copy takes 6 ms, its pwrite ~90: 14 in flight keep the link busy, and what is
  // page-locked is paid for once more when the process leaves -- 0.023 s per GB: 32 pieces = 16 GB merged no faster and left 0.3 s
  // later): a file then leaves in ONE copy and ONE pwrite.  The pieces of one file are written one
  // after the other whatever the number of threads (a write holds the file's inode lock), and every further piece of a file costs
  // ~0.6 ms of wall clock: 1000 x 5 Mbp (92 GB of matrices of 360 MB; the copies by themselves take 1.63 s at the link's 56.5 GB/s,
  // scripts/dev/d2h_bench.cpp) leaves in 3.6-3.8 s through 32 MB pieces, 2.6 s through 64 MB, 2.0-2.1 s through 128 MB (the size of
  // this round's first half), 1.9 s through 256 MB and 1.72 s through 384 or 512 MB -- a piece per file.
  size_t ring_total_mb = 0; bool ring_piece_set = false;
  if (const char* e = getenv("KMX_OUT_RING_MB")) ring_total_mb = (size_t)std::max(64L, atol(e));
  if (const char* e = getenv("KMX_OUT_PIECE_KB")) { ring.bytes = std::max<size_t>(32768, (size_t)atol(e) << 10); ring_piece_set = true; }      // (small pieces: for the tests)
  auto size_ring = [&](uint64_t body_bytes) {      // body_bytes: what a partition's matrix is expected to hold (0: not known)
    if (!ring_piece_set && body_bytes) ring.bytes = (size_t)std::min<uint64_t>((uint64_t)512 << 20, std::max<uint64_t>((uint64_t)32 << 20, (body_bytes + ((32u << 20) - 1)) & ~(uint64_t)((32u << 20) - 1)));
    // (never more than an eighth of the host's memory: the pieces are page-locked)
    const size_t ram8 = (size_t)sysconf(_SC_PHYS_PAGES) / 8 * (size_t)sysconf(_SC_PAGE_SIZE);
    const size_t total = ring_total_mb ? ring_total_mb << 20 : std::min<size_t>(std::max<size_t>(ram8, (size_t)512 << 20), std::min<size_t>((size_t)9216 << 20, std::max<size_t>((size_t)2048 << 20, 18 * ring.bytes)));
    if (!ring_piece_set && ring.bytes > total / 4) ring.bytes = std::max<size_t>((size_t)32 << 20, (total / 4) & ~(size_t)((32u << 20) - 1));
    ring.cap = std::max<size_t>(4, total / ring.bytes);
  };
  size_ring(0);
  std::atomic<bool> ring_stop{false};
  std::thread ring_filler;
  {
    const bool will_merge = o.until == "all" || o.until == "merge";
    const bool streams = will_merge && o.plugin.empty() && !o.text && !(o.cpr && !(o.mode == "hash:bf:bin" || o.mode == "hash:bfc:bin")) && o.mode != "hash:bft:bin";
    // (as many pieces as the matrices can fill: input bytes bound them loosely; small runs do not pin 2 GB for nothing)
    // A thread asks for the pieces while the samples are counted (KMX_RING_PREFILL=0: the writers ask when they need them).  Round 3
    // measured that as a loss (the merge stage gained 0.1 s, the count stage lost 0.3 s: hipHostMalloc under the runtime's lock, and the
    // filler's own first HIP call); with the pieces page-locked from huge pages by the one pin server thread it costs the count stage
    // nothing, and the merge stage of 1000 x 5 Mbp no longer starts with 0.4 s of pinning 32 pieces of 128 MB (the link's timeline:
    // scripts/dev/merge_d2h.sh).
    size_ring(in_bytes * 3 / std::max<uint32_t>(1, o.nb_parts));
    if (streams) { ring.cap = std::max<size_t>(4, std::min<size_t>(ring.cap, (size_t)(in_bytes * 4 / ring.bytes) + 4)); if (!(getenv("KMX_RING_PREFILL") && getenv("KMX_RING_PREFILL")[0] == '0')) ring_filler = std::thread([&]() { ring.prefill(ring_stop); }); }
  }
  struct RingJoin { std::atomic<bool>& stop; std::thread& t; ~RingJoin() { stop = true; if (t.joinable()) t.join(); } } ring_join{ring_stop, ring_filler};
  st.setup_wall = since(t0);
  const auto t_count_stage = clk::now();
  // ================= superk + count, sample by sample (task_scheduler.hpp:251-348) =================
  // Readers (pool threads) parse a sample's files into batches of reads; the worker of GPU (sample mod G) splits every batch
  // (kmx_superk_partition[_stats]) and, at the sample's last batch, counts all its partitions (kmx_count_batch) and hands the
  // count files to the pool for writing.
  if (!o.merge_only) {
    // a batch of reads: the bases lie in page-locked memory (the upload is a DMA at the link's rate and does not hold the worker's
    // thread; from a std::string the runtime stages it through its own pinned block, synchronously), blocks from a pool
    struct PinStr { char* p = nullptr; size_t cap = 0, len = 0; const char* data() const { return p; } size_t size() const { return len; } };
    struct PinPool {
      std::mutex m; std::vector<PinStr> free_;
      PinStr get(size_t want) {
        {
          std::lock_guard<std::mutex> lk(m);
          int best = -1;
          for (size_t i = 0; i < free_.size(); i++) if (free_[i].cap >= want && (best < 0 || free_[i].cap < free_[best].cap)) best = (int)i;
          if (best >= 0) { PinStr r = free_[best]; free_.erase(free_.begin() + best); r.len = 0; return r; }
        }
        PinStr r; r.cap = want + want / 8 + (1u << 16); r.p = (char*)pins.alloc(r.cap);
        if (!r.p) die("pinned host allocation failed");
        return r;
      }
      void put(PinStr b) { if (!b.p) return; std::lock_guard<std::mutex> lk(m); free_.push_back(b); }
      ~PinPool() { for (auto& b : free_) kmx_free_pinned(b.p); }
    } pinpool;
    // (offs_pin: the offsets once more, behind the bases in the page-locked block when there is room -- 8 bytes per read in the block's
    //  spare eighth --: the call uploads them by DMA instead of through the runtime's staging copy on the worker's thread)
    struct ReadBatch {
      uint32_t si = 0; bool last = false; PinStr bases; std::vector<uint64_t> offs; const uint64_t* offs_pin = nullptr;
      void seal() {
        const size_t at = (bases.len + 63) & ~(size_t)63, nb = offs.size() * 8;
        if (bases.p && at + nb <= bases.cap) { memcpy(bases.p + at, offs.data(), nb); offs_pin = reinterpret_cast<const uint64_t*>(bases.p + at); }
      }
      const uint64_t* offsets() const { return offs_pin ? offs_pin : offs.data(); }
    };
    std::vector<std::unique_ptr<Channel<ReadBatch>>> chan;
    // (a queue of 3 batches per worker; more when a worker takes several samples per call, KMX_COUNT_SAMPLES_PER_CALL below)
    { const char* e = getenv("KMX_COUNT_SAMPLES_PER_CALL"); const size_t qcap = e && atol(e) > 1 ? (size_t)atol(e) + 2 : 3;
      for (uint32_t w = 0; w < NW; w++) chan.emplace_back(new Channel<ReadBatch>(qcap)); }
    std::mutex tm; double s_read = 0, s_split = 0, s_count = 0;
    // readers: threads that parse samples in fof order, each into the queue of the sample's worker (bounded by the channel)
    const uint32_t readers = getenv("KMX_READERS") ? (uint32_t)std::max(1L, atol(getenv("KMX_READERS"))) : std::max<uint32_t>(1, std::min<uint32_t>(o.threads > 1 ? o.threads / 2 : 1, 24));
    // pinned blocks for the statistics tables of a sample (kmx_superk_raw: P * 1280 + 2 * 4^m u32), handed back by the task that
    // has written the sample's PartiInfoFile
    const uint64_t nm_ = 1ULL << (2 * o.msize);
    const size_t raw_words = (size_t)P * 1280 + 3 * nm_;      // the partitions' radix counters + {minimizer, super-k-mers, k-mers} triples (as many as occur)
    struct RawPool {
      std::mutex m; std::condition_variable cv; std::vector<uint32_t*> free_; size_t made = 0, cap, words;
      uint32_t* get() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
          if (!free_.empty()) { uint32_t* p = free_.back(); free_.pop_back(); return p; }
          if (made < cap) { made++; lk.unlock(); uint32_t* p = (uint32_t*)pins.alloc(words * 4); if (!p) die("pinned host allocation failed"); return p; }
          cv.wait(lk);
        }
      }
      void put(uint32_t* p) { { std::lock_guard<std::mutex> lk(m); free_.push_back(p); } cv.notify_one(); }
      ~RawPool() { for (auto p : free_) kmx_free_pinned(p); }
    } rawpool;
    // KMX_COUNT_SAMPLES_PER_CALL=n: up to n whole samples that wait in a worker's queue go to the GPU in ONE call
    // (kmx_count_reads_dev_multi: a 1 Mbp sample is a few dozen small kernels and four host round trips, which a call pays once --
    // 0.71 -> 0.49 ms per sample at n = 4 when the call is timed by itself, scripts/bench_count_multi.py).  Through this driver it does
    // not pay yet: 1000 x 1 Mbp count in 0.89-0.95 s at n = 4 against 0.63 s at n = 1 (--skip-partiinfo; two workers), so the default is 1.
    uint32_t per_call = o.per_call ? o.per_call : getenv("KMX_COUNT_SAMPLES_PER_CALL") ? (uint32_t)std::max(1L, atol(getenv("KMX_COUNT_SAMPLES_PER_CALL"))) : 1u;
    if (o.k >= 64) per_call = 1;      // (kmx_count_reads_dev_multi refuses k >= 64 -- the split of such k-mers, k_superk_wide, takes one sample a call; k = 64 is a two-word key but a wide split: the guard stays >= 64)
    rawpool.cap = ((size_t)per_call + 2) * NW + 2; rawpool.words = raw_words;
    std::atomic<uint32_t> next_sample{0};
    // A sample's batches must reach its worker in order; samples are taken in fof order by `readers` threads per round so the
    // GPU workers see them nearly in order too.
    auto reader_fn = [&]() {
      for (uint32_t si; (si = next_sample++) < N;) {
        Channel<ReadBatch>& ch = *chan[si % NW];
        ReadBatch b; b.si = si; b.offs.assign(1, 0);
        b.offs.reserve(1u << 18);      // (short reads: a few hundred thousand a sample -- no regrowth under the parse)
        double rs = 0;
        // (a sample's reads reach the GPU in batches of 256 MB of bases; KMX_READ_BATCH_BYTES lowers that -- for the tests of the
        //  path that adds a sample's batches up)
        static const size_t batch_bytes = getenv("KMX_READ_BATCH_BYTES") ? (size_t)std::max(1L, atol(getenv("KMX_READ_BATCH_BYTES"))) : (size_t)(256u << 20);
        // room for the sample in one block when it fits a batch (its files' sizes bound its bases; 8x for gzip)
        uint64_t est = 4096;
        for (const std::string& f : samples[si].files) { std::error_code ec; const uint64_t sz = fs::file_size(f, ec); est += ec ? 0 : sz * (f.size() > 3 && f.substr(f.size() - 3) == ".gz" ? 8 : 1); }
        const size_t want = (size_t)std::min<uint64_t>(batch_bytes, est);
        b.bases = pinpool.get(want);
        try {
          for (const std::string& f : samples[si].files) {
            SeqReader rd(f); std::string seq; const char* sp = nullptr; size_t sn = 0;
            auto t = clk::now();
            while (rd.next_view(seq, sp, sn)) {      // (round 6: a short read is copied once, from the read block to the page-locked batch)
              if (b.bases.len + sn > b.bases.cap || (b.bases.len + sn > batch_bytes && b.offs.size() > 1)) {
                rs += since(t);
                b.seal(); ch.push(std::move(b));
                b = ReadBatch(); b.si = si; b.offs.assign(1, 0); b.bases = pinpool.get(std::max(want, sn));
                t = clk::now();
              }
              memcpy(b.bases.p + b.bases.len, sp, sn); b.bases.len += sn; b.offs.push_back(b.bases.len);
            }
            rs += since(t);
          }
        } catch (const std::exception& e) { die(e.what()); }
        b.last = true; b.seal(); ch.push(std::move(b));
        std::lock_guard<std::mutex> lk(tm); s_read += rs;
      }
    };
    std::vector<std::thread> rthreads;
    for (uint32_t r = 0; r < readers; r++) rthreads.emplace_back(reader_fn);

    struct SampleState {
      std::vector<std::vector<uint8_t>> streams; std::vector<uint64_t> nk, nk_all;      // k-mers per selected partition / per partition
      std::vector<uint64_t> pc, ms, mk; uint64_t nb_superk = 0;
    };
    std::atomic<uint32_t> samples_left{N};
    std::vector<uint32_t> per_gpu(NW, 0); for (uint32_t si = 0; si < N; si++) per_gpu[si % NW]++;
    const size_t list_rec_bytes = (hash_mode ? 1 : kw) * 8 + 4;
    auto worker_fn = [&](uint32_t g) {
      kmx_ctx* c = gpu[g];
      std::map<uint32_t, SampleState> open;
      std::deque<std::future<void>> writes;
      uint32_t done = 0;
      double w_split = 0, w_count = 0;
      const uint64_t nm = 1ULL << (2 * o.msize);
      std::unique_ptr<ReadBatch> pending;      // (a batch taken from the queue that did not fit the call being put together, or taken ahead: below)
      // the NEXT sample's bases travel to the device while this one is counted (kmx_reads_upload): two workers that each upload
      // and then compute fall into step -- both upload, the GPU idles; both compute, the link idles: 1.0 of every 4.3 ms at
      // 5 Mbp per sample (the GPU's timeline, scripts/dev/gantt.sh).  KMX_READS_AHEAD=0: off
      static const bool reads_ahead = !(getenv("KMX_READS_AHEAD") && getenv("KMX_READS_AHEAD")[0] == '0');
      // (released with the worker whatever way its loop ends -- the queue closed behind an upload, an early exit: ADVICE r4)
      struct DevAhead { kmx_ctx* c; const char* d; ~DevAhead() { if (d) kmx_reads_release(c, d); } } pend_dev{c, nullptr};
      const char*& pending_dev = pend_dev.d;
      // a whole sample through count FILES (no room in the stores, --keep-tmp, --no-resident): split + count in one call, the
      // super-k-mer streams stay in HBM (kmx_count_reads), the counts come back and are written as counts/partition_<p>/<id>.kmer
      auto whole_to_files = [&](const ReadBatch& b) {
        // ---- the whole sample in one batch: split + count in one call, the super-k-mer streams stay in HBM (kmx_count_reads) ----
        const auto t = clk::now();
        const uint32_t si = b.si; const Sample& smp = samples[si];
        tlog(g, "split_begin", si);
        st.bases += b.bases.size();
        std::vector<uint64_t*> keys(P, nullptr); std::vector<uint32_t*> cnts(P, nullptr); std::vector<uint64_t> cnt(P, 0), nkp(P, 0), info(2 * (size_t)P, 0);
        std::vector<uint8_t*> ob(P, nullptr); std::vector<uint64_t> ol(P, 0);
        auto pc = std::make_shared<std::vector<uint64_t>>(), ms = std::make_shared<std::vector<uint64_t>>(), mk = std::make_shared<std::vector<uint64_t>>();
        kmx_superk_stats ks{};
        if (!o.skip_pinfo) { pc->assign((size_t)P * KMX_PINFO_STRIDE, 0); ms->assign(nm, 0); mk->assign(nm, 0); ks.part_counters = pc->data(); ks.minim_superks = ms->data(); ks.minim_kmers = mk->data(); }
        if (o.hist) chk(c, kmx_hist_reset(c), "kmx_hist_reset");
        chk(c, kmx_count_reads(c, b.bases.data(), b.offs.data(), b.offs.size() - 1, o.k, o.msize, table.data(), P, hash_mode ? 1 : 0, hash_mode ? hw.wbits : 0, smp.hard_min,
                               keys.data(), cnts.data(), cnt.data(), nkp.data(), o.keep_tmp ? ob.data() : nullptr, o.keep_tmp ? ol.data() : nullptr, info.data(),
                               o.skip_pinfo ? nullptr : &ks), "kmx_count_reads");
        if (o.hist) writes.push_back(save_hist(c, si));
        uint64_t nkt = 0; for (uint32_t p = 0; p < P; p++) if (selected[p]) nkt += nkp[p];
        st.kmers += nkt;
        { std::string s2; for (uint32_t p = 0; p < P; p++) { s2 += std::to_string(nkp[p]); s2 += "\n"; }      // every partition, as dump_pinfo does (gatb_utils.hpp:46-51; the split counts them all, fill_partitions.hpp:59-105)
          Out pi(root + "/partition_infos/" + smp.id + ".pinfo"); pi.raw(s2.data(), s2.size()); pi.close(); }
        const std::string sd = root + "/superkmers/" + smp.id; fs::create_directories(sd);
        { std::string inf = "skp\n" + sd + "\n" + std::to_string(P) + "\n";
          for (uint32_t p = 0; p < P; p++) {
            inf += std::to_string(selected[p] ? info[2 * p] : 0) + "\n" + std::to_string(selected[p] ? info[2 * p + 1] : 0) + "\n";
            if (o.keep_tmp && selected[p]) { SuperkBlockWriter w(sd + "/skp." + std::to_string(p), p, o.cpr); w.add_stream(ob[p], ol[p], o.k); w.flush(); w.out.close(); }
            if (o.keep_tmp) kmx_free(ob[p]);
          }
          Out f(sd + "/SuperKmerBinInfoFile"); f.raw(inf.data(), inf.size()); f.close(); }
        if (!o.skip_pinfo) {
          uint64_t nk_all = 0; for (uint32_t p = 0; p < P; p++) nk_all += nkp[p];
          const uint64_t nsk = ks.nb_superk;
          writes.push_back(pool.submit([=]() { try { write_parti_info(sd + "/PartiInfoFile", P, nm, nsk, nk_all, pc->data(), ms->data(), mk->data()); } catch (const std::exception& e) { die(e.what()); } }));
        }
        for (uint32_t p = 0; p < P; p++) {
          uint64_t* kk = keys[p]; uint32_t* cc = cnts[p]; const uint64_t nn = cnt[p];
          if (!selected[p]) { kmx_free(kk); kmx_free(cc); continue; }
          writes.push_back(pool.submit([=, &o]() {
            try { if (hash_mode) write_hash_file(count_path(p, si), si, p, kk, cc, nn); else write_kmer_file(count_path(p, si), o.k, si, p, kk, cc, nn, o.cpr); }
            catch (const std::exception& e) { die(e.what()); }
            kmx_free(kk); kmx_free(cc);
          }));
        }
        tlog(g, "split_end", si);
        w_count += since(t);
        done++;
        while (writes.size() > 4u * P) { writes.front().get(); writes.pop_front(); }
      };
      while (done < per_gpu[g]) {
        ReadBatch b;
        if (pending) { b = std::move(*pending); pending.reset(); }
        else if (!chan[g]->pop(b)) break;
        struct Ahead { kmx_ctx* c; const char* d; ~Ahead() { if (d) kmx_reads_release(c, d); } } ahead{c, pending_dev};      // (b's bases on the device, when they were sent ahead)
        pending_dev = nullptr;
        struct Back { PinPool& pp; PinStr s; ~Back() { pp.put(s); } } back{pinpool, b.bases};      // (the block goes back to the pool when this batch is done with)
        const bool whole_sample = b.last && b.offs.size() > 1 && open.find(b.si) == open.end() && o.until != "superk" && !(o.hist && restricted);      // (the fused calls count every partition: not what a histogram of the selected ones needs)
        bool fits = resident_mode && whole_sample;
        if (fits) {   // a k-mer per base at most, a record per k-mer at most: room for that in every store (its share of the partitions)
          const uint64_t worst = (uint64_t)b.bases.size() * list_rec_bytes / G + (1u << 20);
          for (uint32_t d = 0; d < G; d++) if (kmx_store_used(stores[d]) + worst > kmx_store_limit(stores[d])) fits = false;
        }
        if (fits) {
          // ---- whole samples, one batch each, their counts stay in HBM: split + count in one call (kmx_count_reads_dev[_multi]),
          //      partition p's list lands in the store of shard p mod G; only numbers and the statistics tables come back ----
          const auto t = clk::now();
          std::vector<ReadBatch> more;      // the samples that ride along with b
          std::vector<std::unique_ptr<Back>> more_back;
          if (per_call > 1 && !o.hist && (uint64_t)per_call * P <= 65535) {
            uint64_t worst = (uint64_t)b.bases.size() * list_rec_bytes / G + (1u << 20);
            while (more.size() + 1 < per_call) {
              ReadBatch b2;
              if (!chan[g]->try_pop(b2)) break;
              const bool whole2 = b2.last && b2.offs.size() > 1 && open.find(b2.si) == open.end() && samples[b2.si].hard_min == samples[b.si].hard_min;
              const uint64_t w2 = worst + (uint64_t)b2.bases.size() * list_rec_bytes / G;
              bool room = whole2;
              for (uint32_t d = 0; d < G && room; d++) if (kmx_store_used(stores[d]) + w2 > kmx_store_limit(stores[d])) room = false;
              if (!room) { pending.reset(new ReadBatch(std::move(b2))); break; }
              worst = w2;
              more_back.emplace_back(new Back{pinpool, b2.bases});
              more.push_back(std::move(b2));
            }
          }
          const uint32_t S = 1 + (uint32_t)more.size();
          if (reads_ahead && per_call == 1 && !pending) {      // the next batch, if one waits: its bases start their way now
            ReadBatch b2;
            if (chan[g]->try_pop(b2)) {
              pending.reset(new ReadBatch(std::move(b2)));
              if (pending->last && pending->offs.size() > 1 && open.find(pending->si) == open.end())
                chk(c, kmx_reads_upload(c, pending->bases.data(), pending->bases.size(), &pending_dev), "kmx_reads_upload");
            }
          }
          st.count_calls++;
          std::vector<const ReadBatch*> grp; grp.push_back(&b); for (auto& x : more) grp.push_back(&x);
          std::vector<kmx_list> ls((size_t)S * P); std::vector<uint64_t> nkp_all((size_t)S * P, 0), info_all(2 * (size_t)S * P, 0);
          std::vector<uint32_t*> rawbufs(S, nullptr); std::vector<kmx_superk_raw> raws(S);
          for (uint32_t i = 0; i < S; i++) {
            tlog(g, "split_begin", grp[i]->si);
            st.bases += grp[i]->bases.size();
            rawbufs[i] = o.skip_pinfo ? nullptr : rawpool.get();
            raws[i] = kmx_superk_raw{};
            if (rawbufs[i]) { raws[i].part_radix = rawbufs[i]; raws[i].minim_sparse = rawbufs[i] + (size_t)P * 1280; raws[i].minim_sparse_cap = nm; }      // (sparse: ~10^5 of the 4^m minimizers occur in a sample)
          }
          if (o.hist) chk(c, kmx_hist_reset(c), "kmx_hist_reset");
          int crc = KMX_OK;
          if (S == 1)
            crc = kmx_count_reads_dev(c, ahead.d ? ahead.d : b.bases.data(), b.offsets(), b.offs.size() - 1, o.k, o.msize, table.data(), P, hash_mode ? 1 : 0, hash_mode ? hw.wbits : 0, samples[b.si].hard_min,
                                      stores.data(), G, ls.data(), nkp_all.data(), nullptr, nullptr, info_all.data(), nullptr, rawbufs[0] ? &raws[0] : nullptr);
          else {
            std::vector<const char*> bp(S); std::vector<const uint64_t*> op(S); std::vector<uint64_t> ns(S);
            for (uint32_t i = 0; i < S; i++) { bp[i] = grp[i]->bases.data(); op[i] = grp[i]->offs.data(); ns[i] = grp[i]->offs.size() - 1; }
            crc = kmx_count_reads_dev_multi(c, S, bp.data(), op.data(), ns.data(), o.k, o.msize, table.data(), P, hash_mode ? 1 : 0, hash_mode ? hw.wbits : 0, samples[b.si].hard_min,
                                            stores.data(), G, ls.data(), nkp_all.data(), info_all.data(), rawbufs[0] ? raws.data() : nullptr);
          }
          if (crc == KMX_E_NOMEM) {
            // a store filled up under us (the room check above is not atomic with the other workers' calls, and worst-case sizes
            // are estimates): these samples go through count files, as a sample that finds the stores full up front does
            for (uint32_t i = 0; i < S; i++) { if (rawbufs[i]) rawpool.put(rawbufs[i]); st.bases -= grp[i]->bases.size(); }
            if (o.hist) chk(c, kmx_hist_off(c), "kmx_hist_off");
            for (uint32_t i = 0; i < S; i++) whole_to_files(*grp[i]);
            continue;
          }
          chk(c, crc, S == 1 ? "kmx_count_reads_dev" : "kmx_count_reads_dev_multi");
          if (o.hist) writes.push_back(save_hist(c, b.si));
          for (uint32_t i = 0; i < S; i++) {
            const uint32_t si = grp[i]->si; const Sample& smp = samples[si];
            for (uint32_t p = 0; p < P; p++) res_lists[(size_t)si * P + p] = selected[p] ? ls[(size_t)i * P + p] : kmx_list{nullptr, 0};
            res_flag[si] = 1;
            std::vector<uint64_t> nkp(nkp_all.begin() + (size_t)i * P, nkp_all.begin() + (size_t)(i + 1) * P);
            auto info = std::make_shared<std::vector<uint64_t>>(info_all.begin() + 2 * (size_t)i * P, info_all.begin() + 2 * (size_t)(i + 1) * P);
            uint64_t nkt = 0; for (uint32_t p = 0; p < P; p++) if (selected[p]) nkt += nkp[p];
            st.kmers += nkt;
            uint32_t* const rawbuf = rawbufs[i];
            const uint64_t nsk = raws[i].nb_superk, n_sparse = raws[i].minim_sparse_n;
            auto nkp_s = std::make_shared<std::vector<uint64_t>>(std::move(nkp));
            const std::string sid = smp.id;
            writes.push_back(pool.submit([=, &rawpool, &selected]() {
              try {
                { std::string s2; for (uint32_t p = 0; p < P; p++) { s2 += std::to_string((*nkp_s)[p]); s2 += "\n"; }      // every partition, as dump_pinfo does (gatb_utils.hpp:46-51)
                  Out pi(root + "/partition_infos/" + sid + ".pinfo"); pi.raw(s2.data(), s2.size()); pi.close(); }
                const std::string sd = root + "/superkmers/" + sid; fs::create_directories(sd);
                { std::string inf = "skp\n" + sd + "\n" + std::to_string(P) + "\n";
                  for (uint32_t p = 0; p < P; p++) inf += std::to_string(selected[p] ? (*info)[2 * p] : 0) + "\n" + std::to_string(selected[p] ? (*info)[2 * p + 1] : 0) + "\n";
                  Out f(sd + "/SuperKmerBinInfoFile"); f.raw(inf.data(), inf.size()); f.close(); }
                if (rawbuf) write_parti_info_sparse(sd + "/PartiInfoFile", P, nm, nsk, rawbuf, rawbuf + (size_t)P * 1280, n_sparse);
              } catch (const std::exception& e) { die(e.what()); }
              if (rawbuf) rawpool.put(rawbuf);
            }));
            tlog(g, "split_end", si);
            done++;
          }
          w_count += since(t);
          while (writes.size() > 64) { writes.front().get(); writes.pop_front(); }
          continue;
        }
        if (whole_sample) { whole_to_files(b); continue; }
        SampleState& S = open[b.si];
        if (S.streams.empty()) { S.streams.resize(P); S.nk.assign(P, 0); S.nk_all.assign(P, 0); if (!o.skip_pinfo) { S.pc.assign((size_t)P * KMX_PINFO_STRIDE, 0); S.ms.assign(nm, 0); S.mk.assign(nm, 0); } }
        if (b.offs.size() > 1) {
          const auto t = clk::now();
          tlog(g, "split_begin", b.si);
          st.bases += b.bases.size();
          std::vector<uint8_t*> ob(P); std::vector<uint64_t> ol(P), ok(P);
          if (o.skip_pinfo) chk(c, kmx_superk_partition(c, b.bases.data(), b.offs.data(), b.offs.size() - 1, o.k, o.msize, table.data(), P, ob.data(), ol.data(), ok.data()), "kmx_superk_partition");
          else {
            kmx_superk_stats ks{}; ks.part_counters = S.pc.data(); ks.minim_superks = S.ms.data(); ks.minim_kmers = S.mk.data();
            chk(c, kmx_superk_partition_stats(c, b.bases.data(), b.offs.data(), b.offs.size() - 1, o.k, o.msize, table.data(), P, ob.data(), ol.data(), ok.data(), &ks), "kmx_superk_partition_stats");
            S.nb_superk += ks.nb_superk;
          }
          for (uint32_t p = 0; p < P; p++) {
            S.nk_all[p] += ok[p];
            if (selected[p]) { S.streams[p].insert(S.streams[p].end(), ob[p], ob[p] + ol[p]); S.nk[p] += ok[p]; }   // --restrict-to: other partitions are dropped (superk_storage.hpp:301)
            kmx_free(ob[p]);
          }
          tlog(g, "split_end", b.si);
          w_split += since(t);
        }
        if (!b.last) continue;
        // ---- the sample is complete ----
        const uint32_t si = b.si; const Sample& smp = samples[si];
        uint64_t nkt = 0, nkt_all = 0; for (uint32_t p = 0; p < P; p++) { nkt += S.nk[p]; nkt_all += S.nk_all[p]; }
        st.kmers += nkt;
        { std::string s; for (uint32_t p = 0; p < P; p++) { s += std::to_string(S.nk_all[p]); s += "\n"; }   // gatb_utils.hpp:46-51 (every partition: the split counts them all)
          Out pi(root + "/partition_infos/" + smp.id + ".pinfo"); pi.raw(s.data(), s.size()); pi.close(); }
        const std::string sd = root + "/superkmers/" + smp.id; fs::create_directories(sd);
        {
          std::string info = "skp\n" + sd + "\n" + std::to_string(P) + "\n";
          const bool files = o.keep_tmp || o.until == "superk";
          for (uint32_t p = 0; p < P; p++) {
            uint64_t kmp = 0, bfl = 0;
            if (files && selected[p]) {
              SuperkBlockWriter w(sd + "/skp." + std::to_string(p), p, o.cpr);
              w.add_stream(S.streams[p].data(), S.streams[p].size(), o.k);
              kmp = w.kmers; bfl = w.bytes;
              uint64_t a = 0, bb = 0; superk_info_numbers(S.streams[p].data(), S.streams[p].size(), o.k, &a, &bb);
              kmp = a; bfl = bb;                      // saved before the final flush, like task.hpp:315-316 (Appendix B-4)
              w.flush(); w.out.close();
            } else if (selected[p]) superk_info_numbers(S.streams[p].data(), S.streams[p].size(), o.k, &kmp, &bfl);
            info += std::to_string(kmp) + "\n" + std::to_string(bfl) + "\n";
          }
          Out f(sd + "/SuperKmerBinInfoFile"); f.raw(info.data(), info.size()); f.close();
        }
        if (!o.skip_pinfo) {
          auto pc = std::make_shared<std::vector<uint64_t>>(std::move(S.pc)); auto ms = std::make_shared<std::vector<uint64_t>>(std::move(S.ms)); auto mk = std::make_shared<std::vector<uint64_t>>(std::move(S.mk));
          const uint64_t nsk = S.nb_superk;
          writes.push_back(pool.submit([=]() { try { write_parti_info(sd + "/PartiInfoFile", P, nm, nsk, nkt_all, pc->data(), ms->data(), mk->data()); } catch (const std::exception& e) { die(e.what()); } }));
        }
        if (o.until != "superk") {
          const auto t = clk::now();
          tlog(g, "count_begin", si);
          // CountTask / HashCountTask over the sample's partitions (task.hpp:367-392, 447-481): groups of partitions whose
          // k-mers and bytes stay below libkmx's 2^32 per-call limits; a partition beyond them by itself goes alone
          std::vector<uint32_t> grp; uint64_t gk = 0, gb = 0;
          auto run_group = [&]() {
            if (grp.empty()) return;
            const uint32_t n = (uint32_t)grp.size();
            std::vector<const uint8_t*> sp(n); std::vector<uint64_t> sl(n), pid(n), cnt(n);
            std::vector<uint64_t*> keys(n, nullptr); std::vector<uint32_t*> cnts(n, nullptr);
            for (uint32_t i = 0; i < n; i++) { sp[i] = S.streams[grp[i]].data(); sl[i] = S.streams[grp[i]].size(); pid[i] = grp[i]; }
            chk(c, kmx_count_batch(c, n, sp.data(), sl.data(), o.k, hash_mode ? 1 : 0, hash_mode ? hw.wbits : 0, pid.data(), smp.hard_min,
                                   keys.data(), cnts.data(), cnt.data()), "kmx_count_batch");
            for (uint32_t i = 0; i < n; i++) {
              const uint32_t p = grp[i]; uint64_t* kk = keys[i]; uint32_t* cc = cnts[i]; const uint64_t nn = cnt[i];
              std::vector<uint8_t>().swap(S.streams[p]);
              writes.push_back(pool.submit([=, &o]() {
                try { if (hash_mode) write_hash_file(count_path(p, si), si, p, kk, cc, nn); else write_kmer_file(count_path(p, si), o.k, si, p, kk, cc, nn, o.cpr); }
                catch (const std::exception& e) { die(e.what()); }
                kmx_free(kk); kmx_free(cc);
              }));
            }
            grp.clear(); gk = 0; gb = 0;
          };
          // (KMX_COUNT_GROUP_LIMIT lowers the limit for the test of the grouping)
          static const uint64_t LIM = getenv("KMX_COUNT_GROUP_LIMIT") ? (uint64_t)std::max(1LL, atoll(getenv("KMX_COUNT_GROUP_LIMIT"))) : 0x7FFFFFFFULL;
          if (o.hist) chk(c, kmx_hist_reset(c), "kmx_hist_reset");
          for (uint32_t p : plist) {
            const uint64_t nkp = S.nk[p], nbp = S.streams[p].size();
            if (nkp >= 0xFFFFFF00ULL || nbp >= 0xFFFFFF00ULL) die("sample " + smp.id + ", partition " + std::to_string(p) + ": more than 2^32 k-mers in one partition; use more partitions");
            if (!grp.empty() && (gk + nkp > LIM || gb + nbp > LIM)) run_group();
            grp.push_back(p); gk += nkp; gb += nbp;
          }
          run_group();
          if (o.hist) writes.push_back(save_hist(c, si));
          tlog(g, "count_end", si);
          w_count += since(t);
        }
        open.erase(si);
        done++;
        while (writes.size() > 4u * P) { writes.front().get(); writes.pop_front(); }
      }
      for (auto& w : writes) w.get();
      std::lock_guard<std::mutex> lk(tm); s_split += w_split; s_count += w_count;
      (void)samples_left;
    };
    std::vector<std::thread> wthreads;
    for (uint32_t w = 0; w < NW; w++) wthreads.emplace_back(worker_fn, w);
    for (auto& t : rthreads) t.join();
    for (auto& t : wthreads) t.join();
    st.read = s_read; st.split = s_split; st.count = s_count;
  }
  st.count_wall = since(t_count_stage);
  warm_stop = 1; for (auto& t : warmers) if (t.joinable()) t.join();      // (before the merge stage allocates)
  const auto t_merge_stage = clk::now();
  if (o.until == "superk" || o.until == "count") { report(); return 0; }

  // ================= merge, one task per partition (task_scheduler.hpp:381-417), batched per GPU =================
  Plugin plug; if (!o.plugin.empty()) plug.load(o.plugin, o.k);
  std::vector<uint32_t> soft(N, o.soft_min);
  if (!o.soft_min_path.empty()) {      // one threshold per sample, in fof order (cmd/all.hpp:150-162)
    soft.clear();
    std::ifstream in(o.soft_min_path); if (!in) die("Unable to read at " + o.soft_min_path);
    for (std::string line; std::getline(in, line);) { if (line.empty()) continue; try { soft.push_back((uint32_t)std::stol(line)); } catch (...) { die("bad threshold in " + o.soft_min_path + ": " + line); } }
    if (soft.size() != N) die("The number of thresholds in " + o.soft_min_path + " is different from the number of samples.");
  }
  if (o.soft_float) {
    // compute_merge_thresholds (histogram.hpp:218-243) to the letter: `thresholds` starts as N zeros; per sample, the first index i
    // of the unique bins at which the running sum (32-bit) exceeds unique() * p is APPENDED; the file gets every entry, the merge
    // the first N (all zero)
    std::vector<uint32_t> th(N, 0);
    for (uint32_t h = 0; h < N; h++) {
      if (hist_u[h].size() != 256) die("--soft-min <fraction>: no histogram for sample " + samples[h].id);
      uint32_t sum = 0; const uint32_t n = (uint32_t)((double)hist_u[h][255] * o.soft_f);
      for (size_t i = 0; i < 255; i++) { if (sum > n) { th.push_back((uint32_t)i); break; } sum += (uint32_t)hist_u[h][i]; }
    }
    { Out f(root + "/merge_amin.txt"); std::string t; for (uint32_t v : th) t += std::to_string(v) + "\n"; f.raw(t.data(), t.size()); f.close(); }      // kmdir.hpp:183
    soft.assign(th.begin(), th.begin() + N);
  }
  const bool is_bloom = what == "bf" || what == "bfc" || what == "bft";
  const uint32_t mkw = hash_mode ? 1 : kw;
  const size_t rec_bytes = mkw * 8 + 4;
  // per-sample Bloom filter files (hash:bft:bin): filters/<id>.bf = header + u64 number of bits + for p = 0..P-1 the sample's row of
  // partition p's transposed matrix (howde_utils.hpp:133-187)
  if (what == "bft") {
    const auto t = clk::now();
    uint8_t hdr[BF_HEADER_BYTES]; bf_header(hdr, o.k, hw.bloom);
    pool.for_each(N, [&](size_t i) {
      const std::string path = root + "/filters/" + samples[i].id + ".bf";
      const int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0666);
      if (fd < 0) die("Unable to write at " + path);
      const uint64_t bits = hw.bloom;
      if (pwrite(fd, hdr, BF_HEADER_BYTES, 0) != (ssize_t)BF_HEADER_BYTES || pwrite(fd, &bits, 8, BF_HEADER_BYTES) != 8 ||
          ftruncate(fd, (off_t)(BF_HEADER_BYTES + 8 + hw.bloom / 8)) != 0) die("write failed: " + path);
      close(fd);
    });
    st.format += since(t);
  }
  {
    std::mutex tm; double s_io = 0, s_merge = 0, s_format = 0;
    struct PartIn { uint32_t p = 0; std::vector<kmx_list> lists; std::vector<uint8_t> on_dev; };
    struct Batch { std::vector<PartIn> parts; uint8_t* buf = nullptr; uint64_t bytes = 0; double io_s = 0; };
    struct Pinned { uint8_t* p = nullptr; uint64_t cap = 0; void need(uint64_t n) { if (n <= cap) return; kmx_free_pinned(p); cap = n + n / 8 + (1u << 20); p = (uint8_t*)pins.alloc(cap); if (!p) die("pinned host allocation failed"); } ~Pinned() { kmx_free_pinned(p); } };

    auto worker_fn = [&](uint32_t g) {
      kmx_ctx* c = gpu[g];
      std::vector<uint32_t> mine;
      for (size_t i = g; i < plist.size(); i += G) mine.push_back(plist[i]);       // partitions shard round-robin, no collective
      if (mine.empty()) return;
      // batches: count-list bytes (+ Bloom image bytes) up to the budget.  A list is resident (kmx_store) or a count file.
      // COUNT / PA: libkmx sizes its row arenas from the rows-per-record ratio of the batches a context has completed, and a first
      // batch of full size would run twice if the default guess is too small (a cohort keeps several rows per record of a list):
      // the first batch is ONE partition and is waited for before the second is submitted.
      const uint64_t budget = std::max<uint64_t>(o.merge_batch_mb, 16) << 20;
      std::vector<std::vector<uint32_t>> batches; std::vector<std::vector<uint64_t>> fsize;   // per batch: partitions, file sizes [part][sample] (0: resident)
      const bool calibrate = !is_bloom && mine.size() > 2;
      {
        std::vector<uint32_t> cur; std::vector<uint64_t> cur_sz; uint64_t acc = 0;
        for (uint32_t p : mine) {
          uint64_t pb = 0; std::vector<uint64_t> sz(N, 0);
          for (uint32_t i = 0; i < N; i++) {
            if (res_flag[i]) { pb += res_lists[(size_t)i * P + p].n * rec_bytes; continue; }
            std::error_code ec; sz[i] = fs::file_size(count_path(p, i), ec); if (ec) die(count_path(p, i) + " is missing."); pb += sz[i];   // kmdir.hpp:69-70
          }
          if (is_bloom) pb += hw.wbits * (uint64_t)(what == "bfc" ? ((uint64_t)N * o.bitw + 7) / 8 : (N + 7) / 8) * (what == "bft" ? 2 : 1);
          if (!cur.empty() && (acc + pb > budget || cur.size() >= 64 || (calibrate && batches.empty()))) { batches.push_back(cur); fsize.push_back(cur_sz); cur.clear(); cur_sz.clear(); acc = 0; }
          cur.push_back(p); cur_sz.insert(cur_sz.end(), sz.begin(), sz.end()); acc += pb;
        }
        if (!cur.empty()) { batches.push_back(cur); fsize.push_back(cur_sz); }
      }
      Pinned pin[2];
      // loader: the count files of a batch into one pinned buffer, lists back to back (kmx_merge_host then needs one copy)
      auto load = [&](size_t bi, Batch& B) {
        const auto t = clk::now();
        const auto& parts = batches[bi]; const auto& sz = fsize[bi];
        // our own .kmer files: 41-byte header, then the records exactly as kmx_list wants them.  `kmx merge` over a directory somebody
        // else counted: the files of a build with MAX_C <= 65535 hold 1- or 2-byte counts (CMakeLists.txt:25-41, utils.hpp:311-327; the
        // reference's own fixtures do) -- those go through read_kmer_records, which widens them
        bool direct = !hash_mode && !o.cpr;
        if (direct && o.merge_only) {      // (every file's header: a directory may mix builds, or hold an lz4 file among plain ones -- ADVICE r5)
          std::atomic<int> widen{0};
          pool.for_each(parts.size() * N, [&](size_t j) {
            if (res_flag[j % N] || sz[j] < 41 || widen.load(std::memory_order_relaxed)) return;
            const std::string path = count_path(parts[j / N], (uint32_t)(j % N));
            const int fd = open(path.c_str(), O_RDONLY); uint8_t h[41];
            if (fd < 0 || pread(fd, h, 41, 0) != 41 || rd<uint32_t>(h + 29) != 4 || h[12] != 0) widen = 1;
            if (fd >= 0) close(fd);
          });
          if (widen) direct = false;
        }
        B.parts.resize(parts.size());
        for (size_t a = 0; a < parts.size(); a++) {      // the resident lists: merged where they lie
          B.parts[a].p = parts[a]; B.parts[a].lists.assign(N, kmx_list{nullptr, 0}); B.parts[a].on_dev.assign(N, 0);
          for (uint32_t i = 0; i < N; i++) if (res_flag[i]) { B.parts[a].lists[i] = res_lists[(size_t)i * P + parts[a]]; B.parts[a].on_dev[i] = 1; }
        }
        bool any_file = false; for (uint32_t i = 0; i < N; i++) any_file |= !res_flag[i];
        if (!any_file) { B.bytes = 0; B.buf = nullptr; }
        else if (direct) {
          std::vector<uint64_t> off(parts.size() * N + 1, 0);
          for (size_t j = 0; j < parts.size() * N; j++) {
            if (res_flag[j % N]) { off[j + 1] = off[j]; continue; }
            if (sz[j] < 41) die("Invalid file format: " + count_path(parts[j / N], (uint32_t)(j % N)));
            if ((sz[j] - 41) % rec_bytes) die("truncated count file (its body is no whole number of records): " + count_path(parts[j / N], (uint32_t)(j % N)));
            off[j + 1] = off[j] + (sz[j] - 41);
          }
          B.bytes = off.back(); pin[bi & 1].need(B.bytes + 64); B.buf = pin[bi & 1].p;
          pool.for_each(parts.size() * N, [&](size_t j) {
            const uint32_t p = parts[j / N], i = (uint32_t)(j % N);
            if (res_flag[i]) return;
            const std::string path = count_path(p, i);
            const int fd = open(path.c_str(), O_RDONLY); if (fd < 0) die("Unable to read at " + path);
            uint8_t h[41]; if (pread(fd, h, 41, 0) != 41 || rd<uint64_t>(h) != MAGIC_BASE || rd<uint64_t>(h + 13) != MAGIC_KMER || h[12] != 0 || rd<uint32_t>(h + 29) != 4 || rd<uint32_t>(h + 25) != kw) { close(fd); die("Invalid file format: " + path); }
            const uint64_t nb = off[j + 1] - off[j]; uint64_t got = 0;
            while (got < nb) { const ssize_t r = pread(fd, B.buf + off[j] + got, nb - got, (off_t)(41 + got)); if (r <= 0) break; got += (uint64_t)r; }
            close(fd);
            if (got != nb) die("short read: " + path);
            B.parts[j / N].lists[i].recs = B.buf + off[j]; B.parts[j / N].lists[i].n = nb / rec_bytes;
          });
        } else {
          std::vector<std::vector<uint8_t>> recs(parts.size() * N);
          pool.for_each(parts.size() * N, [&](size_t j) {
            if (res_flag[j % N]) return;
            const std::string path = count_path(parts[j / N], (uint32_t)(j % N));
            try { recs[j] = hash_mode ? read_hash_records(path, nullptr) : read_kmer_records(path, nullptr, nullptr); } catch (const std::exception& e) { die(e.what()); }
          });
          std::vector<uint64_t> off(recs.size() + 1, 0);
          for (size_t j = 0; j < recs.size(); j++) off[j + 1] = off[j] + recs[j].size();
          B.bytes = off.back(); pin[bi & 1].need(B.bytes + 64); B.buf = pin[bi & 1].p;
          pool.for_each(recs.size(), [&](size_t j) {
            if (res_flag[j % N]) return;
            if (!recs[j].empty()) memcpy(B.buf + off[j], recs[j].data(), recs[j].size());
            B.parts[j / N].lists[j % N].recs = B.buf + off[j]; B.parts[j / N].lists[j % N].n = recs[j].size() / rec_bytes;
          });
        }
        B.io_s = since(t);
      };
      struct Flight { kmx_merge_result* R = nullptr; Batch B; std::vector<kmx_merge_task> tasks; };
      std::atomic<uint64_t> pending_bytes{0};
      std::deque<std::future<void>> writes;
      double w_io = 0, w_merge = 0, w_format = 0;
      // a matrix body leaves in pieces, two copies in flight (kmx_copy_to_host_async): the link runs the next piece while this one is
      // handed to its writer -- a synchronous copy per piece cost ~0.6 ms of idle link each, whatever its size (32 MB pieces: 3.6 s
      // for the 92 GB of 1000 x 5 Mbp, 128 MB: 2.05 s, 256 MB: 1.9 s; the copies themselves take 1.63 s)
      struct Landing { uint8_t* piece; uint32_t ticket; uint64_t n, off, hl; int fd; std::shared_ptr<std::atomic<uint64_t>> left; std::string path; };
      std::deque<Landing> fly;
      auto land = [&](size_t keep) {
        while (fly.size() > keep) {
          Landing f = std::move(fly.front()); fly.pop_front();
          chk(c, kmx_copy_wait(c, f.ticket), "kmx_copy_wait");
          writes.push_back(pool.submit([f, &ring]() {
            uint64_t done = 0;
            while (done < f.n) { const ssize_t r = pwrite(f.fd, f.piece + done, f.n - done, (off_t)(f.hl + f.off + done)); if (r <= 0) break; done += (uint64_t)r; }
            ring.put(f.piece);
            if (done != f.n) die("write failed: " + f.path);
            if (--*f.left == 0) close(f.fd);
          }));
        }
      };
      // output of a finished batch: bodies + statistics back, files written on the pool
      auto finish = [&](Flight& F) {
        auto t = clk::now();
        chk(c, kmx_result_wait(F.R), "kmx_merge");
        w_merge += since(t);
        t = clk::now();
        // the plain case -- no plugin, no lz4 body, not the per-sample filters -- streams the body from the device to its file
        const bool stream_out = !plug.create && !o.text && !(o.cpr && !is_bloom) && what != "bft";
        for (size_t a = 0; a < F.B.parts.size(); a++) {
          const uint32_t p = F.B.parts[a].p;
          const uint64_t nbytes = kmx_result_body_bytes(F.R, (uint32_t)a), rows = kmx_result_rows(F.R, (uint32_t)a);
          auto stats = std::make_shared<std::vector<uint64_t>>((size_t)6 * N);
          chk(c, kmx_result_copy_stats(F.R, (uint32_t)a, stats->data()), "kmx_result_copy_stats");
          const kmx_merge_task T = F.tasks[a];
          if (stream_out) {
            const std::string ext = what == "count" ? (hash_mode ? "count_hash" : "count") : what == "pa" ? (hash_mode ? "pa_hash" : "pa") : "cmbf";
            const std::string path = root + "/matrices/matrix_" + std::to_string(p) + "." + ext;
            uint64_t hl = 0;
            try {
              Out out(path);
              if (what == "count") { if (hash_mode) matrix_count_hash_header(out, N, p, false); else matrix_count_header(out, o.k, N, p, false); hl = hash_mode ? 37 : 45; }
              else if (what == "pa") { if (hash_mode) matrix_pa_hash_header(out, N, p, false); else matrix_pa_header(out, o.k, N, p, false); hl = hash_mode ? 37 : 45; }
              else { matrix_bf_header(out, what == "bfc" ? N * o.bitw : N, T.lower, T.upper - T.lower + 1, p); hl = 49; }
              out.close();
            } catch (const std::exception& e) { die(e.what()); }
            tlog(g, "merge_done", p);
            // count / pa rows reach their file in one of two ways.  By default the body is put in file order on the device first
            // (kmx_result_body_dev: one device-to-device pass per result, a second copy of the batch's matrices in HBM while they are
            // written) and leaves in large contiguous pieces.  KMX_OUT_ORDER=1 (a cohort whose batches fill the HBM): the arena leaves
            // as the kernels left it and every run of rows is written at its place (below) -- 1000 x 1 Mbp: 0.65 s against 1.35 s for
            // the merge stage, the price of ~10^6 small pwrites instead of 600 large ones.
            static const bool by_order = getenv("KMX_OUT_ORDER") != nullptr;
            if (nbytes && (is_bloom || !by_order)) {      // a dense image: pieces of it straight to their place in the file
              if (!is_bloom && a + 1 < F.B.parts.size()) chk(c, kmx_result_prepare_body(F.R, (uint32_t)a + 1), "kmx_result_prepare_body");      // (the next task's ordering pass runs behind this task's copies)
              const uint8_t* dbody = (const uint8_t*)kmx_result_body_dev(F.R, (uint32_t)a);
              if (!dbody) die(std::string("kmx_result_body_dev: ") + kmx_last_error(c));
              const int fd = open(path.c_str(), O_WRONLY); if (fd < 0) die("Unable to write at " + path);
              auto left = std::make_shared<std::atomic<uint64_t>>((nbytes + ring.bytes - 1) / ring.bytes);
              for (uint64_t off = 0; off < nbytes; off += ring.bytes) {
                const uint64_t n = std::min<uint64_t>(ring.bytes, nbytes - off);
                uint8_t* piece = ring.get();
                uint32_t ticket = 0;
                chk(c, kmx_copy_to_host_async(c, piece, dbody + off, n, &ticket), "kmx_copy_to_host_async");
                fly.push_back(Landing{piece, ticket, n, off, hl, fd, left, path});
                land(1);      // (this piece travels while the one before it is handed on)
              }
            } else if (nbytes) {
              // count / pa rows: the arena as the kernels left it (k_merge_cols: the row keys' rows, then the rows out of k_cols_sparse;
              // the other kernels: row segments) comes to the host piece by piece, and every run of rows that is a run of the body
              // too goes to its place in the file with one pwrite -- the file order comes to exist in the file; no pass over the
              // matrix on the device, no second copy of it in HBM
              const uint64_t rb = kmx_result_row_bytes(F.R, (uint32_t)a);
              const void* dar = nullptr; uint64_t arows = 0;
              chk(c, kmx_result_arena(F.R, (uint32_t)a, &dar, &arows), "kmx_result_arena");
              std::vector<uint32_t> order(rows);
              chk(c, kmx_result_copy_order(F.R, (uint32_t)a, order.data()), "kmx_result_copy_order");
              auto inv = std::make_shared<std::vector<uint32_t>>(arows, 0xFFFFFFFFu);      // arena row -> row of the body (unused arena rows: none)
              for (uint64_t d = 0; d < rows; d++) { if (order[d] >= arows) die("corrupt row order"); (*inv)[order[d]] = (uint32_t)d; }
              const int fd = open(path.c_str(), O_WRONLY); if (fd < 0) die("Unable to write at " + path);
              const uint64_t prow = std::max<uint64_t>(1, ring.bytes / rb);               // arena rows per piece
              auto left = std::make_shared<std::atomic<uint64_t>>((arows + prow - 1) / prow);
              for (uint64_t r0 = 0; r0 < arows; r0 += prow) {
                const uint64_t nr = std::min<uint64_t>(prow, arows - r0);
                uint8_t* piece = ring.get();
                chk(c, kmx_copy_to_host(c, piece, (const uint8_t*)dar + r0 * rb, nr * rb), "kmx_copy_to_host");
                writes.push_back(pool.submit([=, &ring]() {
                  const uint32_t* iv = inv->data();
                  for (uint64_t r = r0; r < r0 + nr;) {
                    if (iv[r] == 0xFFFFFFFFu) { r++; continue; }
                    uint64_t e = r + 1; while (e < r0 + nr && iv[e] == iv[e - 1] + 1) e++;      // rows r .. e - 1 are rows iv[r] .. of the body
                    const uint64_t n = (e - r) * rb; uint64_t done = 0;
                    while (done < n) { const ssize_t w = pwrite(fd, piece + (r - r0) * rb + done, n - done, (off_t)(hl + (uint64_t)iv[r] * rb + done)); if (w <= 0) break; done += (uint64_t)w; }
                    if (done != n) die("write failed: " + path);
                    r = e;
                  }
                  ring.put(piece);
                  if (--*left == 0) close(fd);
                }));
              }
            }
            writes.push_back(pool.submit([=, &o, &hw, &res_flag]() {
              try {
                write_merge_info(root + "/merge_infos/partition" + std::to_string(p) + ".merge_info", stats->data(), N);
                if (what == "bf") {   // task.hpp:849-860 + utils.hpp:239-243
                  std::ostringstream fp;
                  for (uint32_t i = 0; i < N; i++) fp << std::fixed << std::pow(1.0 - std::pow(std::exp(1.0), -(double)(*stats)[(size_t)3 * N + i] / (double)hw.wbits), 1.0) << "\n";
                  Out f(root + "/fpr/partition_" + std::to_string(p) + ".txt"); const std::string s2 = fp.str(); f.raw(s2.data(), s2.size()); f.close();
                }
                if (!o.keep_tmp) for (uint32_t i = 0; i < N; i++) if (!res_flag[i]) fs::remove(count_path(p, i));   // task.hpp:676-688
              } catch (const std::exception& e) { die(e.what()); }
            }));
            while (writes.size() > 512) { writes.front().get(); writes.pop_front(); }
            continue;
          }
          auto body = std::make_shared<std::vector<uint8_t>>(nbytes);
          chk(c, kmx_result_copy_body(F.R, (uint32_t)a, body->data(), nbytes), "kmx_result_copy_body");
          tlog(g, "merge_done", p);
          bool plugin_done = false;
          if (plug.create && is_bloom) {
            // A plugin in a Bloom mode (merge.hpp:509-514: process_hash sees EVERY hash of the window's lists, in ascending order, with
            // the counts the soft-min / rescue rules left; its answer replaces the recurrence test, write_as_bf / bfc / bft then pack
            // what it left in the counts, merge.hpp:575-644).  The batch ran as count rows with recurrence-min 0 (every hash is a
            // row); here, on the worker's thread (the context is this thread's): the plugin over the rows, the Bloom rows packed
            // into the window's dense image, and for hash:bft:bin the transpose (kmx_transpose_bits).
            const uint64_t W = T.upper - T.lower + 1;
            const size_t rb_in = 8 + 4ull * N, rb_out = what == "bfc" ? ((size_t)N * o.bitw + 7) / 8 : (N + 7) / 8;
            auto img = std::make_shared<std::vector<uint8_t>>((size_t)((W + 7) & ~7ULL) * rb_out, 0);
            km::IMergePlugin* pl = plug.create(); pl->configure(o.plugin_config);
            pl->set_out_dir(root + "/plugin_output"); pl->set_kmer_size(0); pl->set_partition(p);
            std::vector<km::IMergePlugin::count_type> cv(N);
            uint64_t last = T.lower;
            for (uint64_t r = 0; r <= rows; r++) {      // r == rows: the reference's extra call after the last row
              const uint8_t* row = body->data() + r * rb_in;
              if (r < rows) { memcpy(&last, row, 8); for (uint32_t i = 0; i < N; i++) { uint32_t v; memcpy(&v, row + 8 + 4 * i, 4); cv[i] = (km::IMergePlugin::count_type)v; } }
              else std::fill(cv.begin(), cv.end(), 0);
              const bool keep = pl->process_hash(last, cv);
              if (r == rows || !keep || last < T.lower || last > T.upper) continue;
              uint8_t* out_row = img->data() + (last - T.lower) * rb_out;
              if (what == "bfc") {      // pack_v: to_n_b(c, w) = min(bit_length(c), 2^w - 1), MSB first at bit i * w (packc.hpp:26-43)
                for (uint32_t i = 0; i < N; i++) {
                  uint32_t cc = (uint32_t)cv[i], bl = 0; while (cc) { bl++; cc >>= 1; }
                  const uint32_t v = std::min<uint32_t>(bl, o.bitw >= 32 ? 0xFFFFFFFFu : (1u << o.bitw) - 1);
                  for (uint32_t b = 0; b < o.bitw; b++) if ((v >> (o.bitw - 1 - b)) & 1u) { const uint64_t bit = (uint64_t)i * o.bitw + b; out_row[bit >> 3] |= (uint8_t)(0x80u >> (bit & 7)); }
                }
              } else for (uint32_t i = 0; i < N; i++) if (cv[i]) out_row[i >> 3] |= (uint8_t)(1u << (i & 7));
            }
            plug.destroy(pl);
            if (what == "bft") {      // write_as_bft: BitMatrix(ROUND_UP(W, 8), ROUND_UP(N, 8) / 8) transposed (merge.hpp:631-644)
              const uint64_t W8 = (W + 7) & ~7ULL, N8 = (uint64_t)rb_out * 8;
              auto tr = std::make_shared<std::vector<uint8_t>>((size_t)(N8 * (W8 / 8)), 0);
              chk(c, kmx_transpose_bits(c, img->data(), W8, N8, tr->data()), "kmx_transpose_bits");
              img = tr;
            } else img->resize((size_t)W * rb_out);
            body = img;
            plugin_done = true;
          }
          pending_bytes += body->size();
          writes.push_back(pool.submit([=, &o, &plug, &samples, &hw, &tm, &s_format, &res_flag, &pending_bytes]() {
            struct Done { std::atomic<uint64_t>& p; uint64_t n; ~Done() { p -= n; } } done_{pending_bytes, (uint64_t)body->size()};
            try {
              const std::string ext = what == "count" ? (hash_mode ? "count_hash" : "count") : what == "pa" ? (hash_mode ? "pa_hash" : "pa") : "cmbf";
              const bool lz4_name = o.cpr && !hash_mode && !is_bloom && !o.text;             // hash-mode matrices never get the suffix (task.hpp:794-795), text ones neither (kmdir.hpp:127-130)
              const bool cpr_body = o.cpr && !is_bloom && !o.text;
              Out out(root + "/matrices/matrix_" + std::to_string(p) + "." + ext + (o.text ? ".txt" : "") + (lz4_name ? ".lz4" : ""));
              if (o.text) {}      // (a text matrix has no header: a plain ofstream, merge.hpp:288-316)
              else if (what == "count") { if (hash_mode) matrix_count_hash_header(out, N, p, cpr_body); else matrix_count_header(out, o.k, N, p, cpr_body); }
              else if (what == "pa") { if (hash_mode) matrix_pa_hash_header(out, N, p, cpr_body); else matrix_pa_header(out, o.k, N, p, cpr_body); }
              else matrix_bf_header(out, what == "bfc" ? N * o.bitw : N, T.lower, T.upper - T.lower + 1, p);
              km::IMergePlugin* pl = nullptr;
              if (plug.create && !plugin_done) {
                pl = plug.create(); pl->configure(o.plugin_config);                              // plugin_manager.hpp:106-111
                pl->set_out_dir(root + "/plugin_output"); pl->set_kmer_size(hash_mode ? 0 : o.k); pl->set_partition(p);   // task.hpp:701-712
              }
              // a row as a line of text (write_as_text / write_as_pa_text, merge.hpp:288-316, 531-572): the k-mer as letters
              // (Kmer::to_string, kmer.hpp:541-550: most significant digit first, A C T G = 0 1 2 3) or the hash in decimal, then
              // " <count>" -- " 1" / " 0" for presence/absence -- per sample.  The batch ran as count rows (key + N x u32).
              std::string line;
              auto text_row = [&](const uint64_t* key, auto&& count_of) {
                line.clear();
                if (hash_mode) line += std::to_string(key[0]);
                else for (uint32_t d = o.k; d-- > 0;) line += "ACTG"[(key[d >> 5] >> (2 * (d & 31))) & 3];
                for (uint32_t i = 0; i < N; i++) { line += ' '; const uint32_t cnt = count_of(i); if (what == "pa") line += cnt ? '1' : '0'; else line += std::to_string(cnt); }
                line += '\n';
                out.raw(line.data(), line.size());
              };
              if (!pl && o.text) {
                const uint32_t kb = T.key_words * 8; const size_t rb = kb + 4ull * N;
                uint64_t key[4] = {0, 0, 0, 0};
                for (uint64_t r = 0; r < rows; r++) {
                  const uint8_t* row = body->data() + r * rb;
                  memcpy(key, row, kb);
                  text_row(key, [&](uint32_t i) { uint32_t v; memcpy(&v, row + kb + 4 * i, 4); return v; });
                }
              } else if (!pl) out.raw(body->data(), body->size());
              else {   // the plugin's return value replaces the recurrence test: every row was produced, filter here
                const uint32_t kb = T.key_words * 8; const size_t rb = kb + 4ull * N;
                std::vector<km::IMergePlugin::count_type> cv(N); std::vector<uint8_t> pa((N + 7) / 8);
                uint64_t last_key[4] = {0, 0, 0, 0};      // (up to Kmer<128>)
                for (uint64_t r = 0; r <= rows; r++) {   // r == rows: the reference's extra call after the last row (merge.hpp:185-259)
                  const uint8_t* row = body->data() + r * rb;
                  if (r < rows) { memcpy(last_key, row, kb); for (uint32_t i = 0; i < N; i++) { uint32_t v; memcpy(&v, row + kb + 4 * i, 4); cv[i] = (km::IMergePlugin::count_type)v; } }
                  else std::fill(cv.begin(), cv.end(), 0);
                  const bool keep = hash_mode ? pl->process_hash(last_key[0], cv) : pl->process_kmer(last_key, cv);
                  if (r == rows || !keep) continue;
                  if (o.text) { text_row(last_key, [&](uint32_t i) { return (uint32_t)cv[i]; }); continue; }
                  out.raw(last_key, kb);
                  if (what == "count") for (uint32_t i = 0; i < N; i++) { const uint32_t v = (uint32_t)cv[i]; out.raw(&v, 4); }
                  else { std::fill(pa.begin(), pa.end(), 0); for (uint32_t i = 0; i < N; i++) if (cv[i]) pa[i >> 3] |= (uint8_t)(1u << (i & 7)); out.raw(pa.data(), pa.size()); }
                }
                plug.destroy(pl);
              }
              out.close();
              write_merge_info(root + "/merge_infos/partition" + std::to_string(p) + ".merge_info", stats->data(), N);
              if (what == "bf" || what == "bft") {   // task.hpp:849-860 + utils.hpp:239-243
                std::ostringstream fp;
                for (uint32_t i = 0; i < N; i++) fp << std::fixed << std::pow(1.0 - std::pow(std::exp(1.0), -(double)(*stats)[(size_t)3 * N + i] / (double)hw.wbits), 1.0) << "\n";
                Out f(root + "/fpr/partition_" + std::to_string(p) + ".txt"); const std::string s = fp.str(); f.raw(s.data(), s.size()); f.close();
              }
              if (what == "bft") {   // sample i's row of this partition -> its place in filters/<id>.bf
                const auto tf = clk::now();
                const uint64_t rowb = hw.wbits / 8;
                for (uint32_t i = 0; i < N; i++) {
                  const std::string path = root + "/filters/" + samples[i].id + ".bf";
                  const int fd = open(path.c_str(), O_WRONLY); if (fd < 0) die("Unable to write at " + path);
                  const off_t at = (off_t)(BF_HEADER_BYTES + 8 + (uint64_t)p * rowb); uint64_t done = 0;
                  while (done < rowb) { const ssize_t r = pwrite(fd, body->data() + (uint64_t)i * rowb + done, rowb - done, at + (off_t)done); if (r <= 0) break; done += (uint64_t)r; }
                  close(fd);
                  if (done != rowb) die("write failed: " + path);
                }
                std::lock_guard<std::mutex> lk(tm); s_format += since(tf);
              }
              if (!o.keep_tmp) for (uint32_t i = 0; i < N; i++) if (!res_flag[i]) fs::remove(count_path(p, i));   // task.hpp:676-688
            } catch (const std::exception& e) { die(e.what()); }
          }));
        }
        land(0);      // (the last pieces of the batch: the device bodies go with the result)
        kmx_result_free(F.R); F.R = nullptr;
        w_io += since(t);
        // (bodies handed to the pool whole -- plugin, lz4, per-sample filters -- are bounded in BYTES: at most ~8 GB wait to be written)
        while (!writes.empty() && (writes.size() > 512 || pending_bytes.load() > ((uint64_t)8 << 30))) { writes.front().get(); writes.pop_front(); }
      };
      Batch next; std::future<void> fut;
      auto start_load = [&](size_t bi) { next = Batch(); fut = std::async(std::launch::async, [&, bi]() { load(bi, next); }); };
      start_load(0);
      std::unique_ptr<Flight> prev;
      for (size_t bi = 0; bi < batches.size(); bi++) {
        fut.get();
        std::unique_ptr<Flight> F(new Flight()); F->B = std::move(next);
        w_io += F->B.io_s;
        F->tasks.resize(F->B.parts.size());
        for (size_t a = 0; a < F->B.parts.size(); a++) {
          kmx_merge_task& t = F->tasks[a]; t = kmx_merge_task{};
          const uint32_t p = F->B.parts[a].p;
          t.n_lists = N; t.key_words = mkw; t.lists = F->B.parts[a].lists.data(); t.soft_min = soft.data();
          t.rec_min = o.rec_min; t.share_min = o.share_min; t.bitw = o.bitw;
          t.mode = what == "count" ? KMX_MODE_COUNT : what == "pa" ? KMX_MODE_PA : what == "bf" ? KMX_MODE_BF : what == "bfc" ? KMX_MODE_BFC : KMX_MODE_BFT;
          if (is_bloom) { t.lower = hw.lower(p); t.upper = hw.upper(p); }
          if (plug.create) { t.rec_min = 0; t.mode = KMX_MODE_COUNT; }
          if (o.text) t.mode = KMX_MODE_COUNT;      // (the text writers print the counts, or whether they are zero)
          t.list_on_device = F->B.parts[a].on_dev.data();
          for (uint32_t i = 0; i < N; i++) st.merge_recs += t.lists[i].n;
        }
        tlog(g, "merge_submit", F->B.parts[0].p);
        const auto t = clk::now();
        chk(c, kmx_merge_host(c, F->tasks.data(), (uint32_t)F->tasks.size(), &F->R), "kmx_merge_host");
        w_merge += since(t);
        if (prev) finish(*prev);                 // (its pinned buffer is free again: the next load may take it)
        if (bi + 1 < batches.size()) start_load(bi + 1);
        prev = std::move(F);
        if (bi == 0 && calibrate && batches.size() > 1) { finish(*prev); prev.reset(); }      // (the one-partition batch the arenas of the others are sized from)
      }
      if (prev) finish(*prev);
      for (auto& w : writes) w.get();
      std::lock_guard<std::mutex> lk(tm); s_io += w_io; s_merge += w_merge; s_format += w_format;
    };
    std::vector<std::thread> wthreads;
    for (uint32_t g = 0; g < G; g++) wthreads.emplace_back(worker_fn, g);
    for (auto& t : wthreads) t.join();
    st.merge_io = s_io; st.merge = s_merge; st.format += s_format;
  }
  st.merge_wall = since(t_merge_stage);
  report();
  struct rusage ru; getrusage(RUSAGE_SELF, &ru);
  { std::ofstream ri(root + "/run_infos.txt");                                         // task_scheduler.hpp:453-457
    ri << "Time: " << std::chrono::duration_cast<std::chrono::seconds>(clk::now() - t0).count() << " seconds\n"
       << "Memory: " << ru.ru_maxrss / 1024 << "MB\n"; }
  // every file is written and closed, every worker joined: leave without giving tens of GB of device and pinned memory back
  // block by block (the driver and the OS take them back with the process; KMX_SLOW_EXIT=1 runs the destructors, for leak checks)
  if (getenv("KMX_TRACE")) { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); fprintf(stderr, "[kmx epoch] leaving %.6f\n", (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec); }
  if (!getenv("KMX_SLOW_EXIT")) { fflush(stdout); fflush(stderr); _exit(0); }
  return 0;
}

int kmx_tools_main(int argc, char** argv);      // kmx_tools.cpp: dump, aggregate

int main(int argc, char** argv)
{
  if (argc >= 2 && (std::string(argv[1]) == "dump" || std::string(argv[1]) == "aggregate" || std::string(argv[1]) == "combine")) return kmx_tools_main(argc, argv);
  try { return run(argc, argv); }
  catch (const std::exception& e) { die(e.what()); }
}
