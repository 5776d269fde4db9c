// kmx_pipeline.cpp -- `kmx pipeline ...`: the `kmtricks pipeline` command line, run-directory layout
// and plugin loading over libkmx (the MI355X engine).  Mirrors reference src/cli.cpp:117-382 (flags and
// defaults), include/kmtricks/kmdir.hpp:195-241 (directory tree), task.hpp:98-124, 170-225, 255-320,
// 367-392, 447-481, 690-743, 787-863 (what each stage reads and writes), io/fof.hpp:39-43 (fof grammar),
// plugin_manager.hpp:38-113 (plugin symbols).  All compute goes through the C ABI of include/kmx.h;
// this file only parses, reads and writes files.  Errors: message on stderr + exit(EXIT_FAILURE)
// (reference src/kmtricks.cpp:109-123).
#include <kmx.h>
#include <kmtricks/plugin.hpp>

#include <dlfcn.h>
#include <sys/resource.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <future>
#include <mutex>
#include <thread>
#include <cmath>
#include <filesystem>
#include <iomanip>
#include <iostream>
#include <map>
#include <regex>
#include <sstream>

#include "kmx_io.hpp"

namespace fs = std::filesystem;
using namespace kmxio;

struct Sample { std::string id; std::vector<std::string> files; uint32_t hard_min; };

struct Opt {
  std::string fof, dir, mode = "kmer:count:bin", until = "all", plugin, plugin_config, repart_from, repart_file;
  uint32_t k = 31, hard_min = 2, soft_min = 1, rec_min = 1, share_min = 0, nb_parts = 0, msize = 10, bitw = 2, threads = 1, gpus = 1;
  uint64_t bloom = 10000000;
  bool static_repart = false, keep_tmp = false, cpr = false;
};

[[noreturn]] static void die(const std::string& msg) { std::cerr << "[error] " << msg << std::endl; std::exit(EXIT_FAILURE); }

static uint64_t xxh64_u32(uint32_t v)
{ // XXH64(&v, 4, seed 0) -- static repartition (reference include/kmtricks/repartition.hpp:45-56)
  const uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P5 = 0x27D4EB2F165667C5ULL;
  uint64_t h = P5 + 4;
  h ^= (uint64_t)v * P1; h = ((h << 23) | (h >> 41)) * P2 + P3;
  h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
  return h;
}

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
  if (argc < 2 || std::string(argv[1]) != "pipeline") die("usage: kmx pipeline --file <fof> --run-dir <dir> [options]  (see INTEGRATION.md)");
  Opt o;
  auto need = [&](int& i) -> std::string { if (i + 1 >= argc) die(std::string("missing value for ") + argv[i]); return argv[++i]; };
  for (int i = 2; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "--file") o.fof = need(i);
    else if (a == "--run-dir") o.dir = need(i);
    else if (a == "--kmer-size") o.k = std::stoul(need(i));
    else if (a == "--hard-min") o.hard_min = std::stoul(need(i));
    else if (a == "--mode") o.mode = need(i);
    else if (a == "--soft-min") o.soft_min = std::stoul(need(i));
    else if (a == "--recurrence-min") o.rec_min = std::stoul(need(i));
    else if (a == "--share-min") o.share_min = std::stoul(need(i));
    else if (a == "--nb-partitions") o.nb_parts = std::stoul(need(i));
    else if (a == "--minimizer-size") o.msize = std::stoul(need(i));
    else if (a == "--bloom-size") o.bloom = (uint64_t)std::stod(need(i));
    else if (a == "--bitw") o.bitw = std::stoul(need(i));
    else if (a == "--until") o.until = need(i);
    else if (a == "--static-repart") o.static_repart = true;
    else if (a == "--repart-from") o.repart_from = need(i);
    else if (a == "--repart-file") o.repart_file = need(i);     // kmx extension: use this minimRepart table as is
    else if (a == "--keep-tmp") o.keep_tmp = true;
    else if (a == "--cpr") o.cpr = true;
    else if (a == "--plugin") o.plugin = need(i);
    else if (a == "--plugin-config") o.plugin_config = need(i);
    else if (a == "-t" || a == "--threads") o.threads = std::stoul(need(i));
    else if (a == "--gpus") o.gpus = std::stoul(need(i));       // kmx extension: partitions p -> GPU p % gpus
    else if (a == "-v" || a == "--verbose") need(i);
    else die("unknown option " + a);
  }
  if (o.fof.empty() || o.dir.empty()) die("--file and --run-dir are required");
  if (fs::exists(o.dir)) die("--run-dir already exists: " + o.dir);                  // src/cli.cpp:101-104
  if (o.k < 8 || o.k > 63) die("--kmer-size must be in [8, 63] for this build");
  if (o.msize < 4 || o.msize > 15 || o.msize >= o.k) die("--minimizer-size must be in [4, 15] and < k");
  if (o.cpr) die("--cpr (lz4 / TurboPFor bodies) is not supported by this build");
  static const char* modes[] = {"kmer:count:bin", "kmer:pa:bin", "hash:count:bin", "hash:pa:bin", "hash:bf:bin", "hash:bfc:bin"};
  if (std::find_if(std::begin(modes), std::end(modes), [&](const char* m) { return o.mode == m; }) == std::end(modes))
    die("--mode " + o.mode + " is not supported (kmer:{count,pa}:bin, hash:{count,pa,bf,bfc}:bin)");
  static const char* untils[] = {"all", "repart", "superk", "count", "merge"};
  if (std::find_if(std::begin(untils), std::end(untils), [&](const char* m) { return o.until == m; }) == std::end(untils)) die("bad --until");
  if (o.nb_parts == 0) o.nb_parts = 4;                                                // task.hpp:112-115 forces >= 4; auto-sizing is system dependent
  if (o.nb_parts > 65535) die("--nb-partitions too large");
  if (!o.plugin.empty() && (o.mode == "hash:bf:bin" || o.mode == "hash:bfc:bin")) die("--plugin with Bloom modes is not supported by this build");
  return o;
}

struct Ctx {
  std::vector<kmx_ctx*> c;
  explicit Ctx(uint32_t gpus) { for (uint32_t g = 0; g < gpus; g++) { kmx_ctx* x = nullptr; if (kmx_create((int)g, &x) != KMX_OK) die(kmx_last_error(nullptr)); c.push_back(x); } }
  ~Ctx() { for (auto x : c) kmx_destroy(x); }
  kmx_ctx* of_partition(uint32_t p) const { return c[p % c.size()]; }   // partitions shard round-robin, no collective
};
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

int main(int argc, char** argv)
{
  const auto t0 = std::chrono::steady_clock::now();
  Opt o = parse_cli(argc, argv);
  std::vector<Sample> samples = parse_fof(o.fof, o.hard_min);
  const uint32_t N = (uint32_t)samples.size(), P = o.nb_parts, kw = (o.k + 31) / 32;
  const bool hash_mode = o.mode.rfind("hash:", 0) == 0;
  const std::string what = o.mode.substr(o.mode.find(':') + 1, o.mode.rfind(':') - o.mode.find(':') - 1);   // count|pa|bf|bfc

  // ---- run directory (kmdir.hpp:195-241) ----
  const std::string root = fs::absolute(o.dir).string();
  for (const char* d : {"", "/superkmers", "/counts", "/matrices", "/filters", "/histograms", "/merge_infos", "/howde_index",
                        "/partition_infos", "/fpr", "/plugin_output", "/repartition_gatb", "/config_gatb"})
    fs::create_directories(root + d);
  fs::copy_file(o.fof, root + "/kmtricks.fof");
  { std::ofstream b(root + "/build_infos.txt"); b << "kmx (MI355X-native kmtricks pipeline), libkmx ABI " << kmx_version() << "\n"; }
  { std::ofstream f(root + "/options.txt");
    f << "Options: dir=" << root << ", nb_threads=" << o.threads << ", fof=" << o.fof << ", kmer_size=" << o.k << ", c_ab_min=" << o.hard_min
      << ", m_ab_min=" << o.soft_min << ", r_min=" << o.rec_min << ", save_if=" << o.share_min << ", minim_size=" << o.msize
      << ", nb_parts=" << P << ", bloom_size=" << o.bloom << ", keep_tmp=" << o.keep_tmp << ", static_repart=" << o.static_repart
      << ", bwidth=" << o.bitw << ", mode=" << o.mode << ", until=" << o.until << "\n"; }
  for (uint32_t p = 0; p < P; p++) fs::create_directories(root + "/counts/partition_" + std::to_string(p));
  HashWindow hw(o.bloom, P, o.msize);
  hw.save(root + "/hash.info");                                                       // task.hpp:98-124

  // ---- repartition (task.hpp:170-225) ----
  std::vector<uint16_t> table;
  const std::string rpath = root + "/repartition_gatb/repartition.minimRepart";
  if (!o.repart_file.empty() || !o.repart_from.empty()) {
    uint16_t np = 0;
    table = read_repartition(o.repart_file.empty() ? o.repart_from + "/repartition_gatb/repartition.minimRepart" : o.repart_file, &np);
    if (np != P || table.size() != (1ULL << (2 * o.msize))) die("repartition table does not match --nb-partitions / --minimizer-size");   // task.hpp:136-147
  } else if (o.static_repart) {
    table.resize(1ULL << (2 * o.msize));
    for (uint64_t m = 0; m < table.size(); m++) table[m] = (uint16_t)(xxh64_u32((uint32_t)m) % P);
  } else die("the sampled repartition is not built yet: pass --static-repart (or --repart-from / --repart-file)");
  write_repartition(rpath, (uint16_t)P, table);
  if (o.msize <= 12) {   // minimizers/minimizers.<p>: every m-mer assigned to partition p, one per line (task.hpp:160-168, repartition.hpp:116-124)
    fs::create_directories(root + "/minimizers");
    std::vector<std::string> buf(P);
    std::string mm(o.msize, 'A');
    for (uint64_t v = 0; v < table.size(); v++) {
      uint64_t t = v;
      for (int i = (int)o.msize - 1; i >= 0; i--) { mm[i] = "ACTG"[t & 3]; t >>= 2; }   // Mmer::to_string (kmer.hpp:115-127)
      std::string& b = buf[table[v]]; b += mm; b += '\n';
    }
    for (uint32_t p = 0; p < P; p++) { std::ofstream f(root + "/minimizers/minimizers." + std::to_string(p)); f.write(buf[p].data(), (std::streamsize)buf[p].size()); }
  }
  if (o.until == "repart") return 0;

  // wall-clock per stage (one line on stderr at the end; parsed by scripts/bench_pipeline.py)
  using clk = std::chrono::steady_clock;
  const auto t_start = clk::now();
  double s_read = 0, s_split = 0, s_count = 0, s_merge_io = 0, s_merge = 0;
  uint64_t n_bases = 0, n_kmers = 0, n_merge_recs = 0;
  auto since = [](clk::time_point t) { return std::chrono::duration<double>(clk::now() - t).count(); };
  auto report = [&]() {
    fprintf(stderr, "[kmx pipeline] {\"samples\": %u, \"partitions\": %u, \"bases\": %llu, \"kmers\": %llu, \"merge_records\": %llu, "
                    "\"read_s\": %.4f, \"superk_s\": %.4f, \"count_s\": %.4f, \"merge_io_s\": %.4f, \"merge_s\": %.4f, \"total_s\": %.4f}\n",
            N, P, (unsigned long long)n_bases, (unsigned long long)n_kmers, (unsigned long long)n_merge_recs,
            s_read, s_split, s_count, s_merge_io, s_merge, since(t_start));
  };
  Ctx gpu(o.gpus);
  // ---- superk + count, sample by sample (task_scheduler.hpp:251-348) ----
  // A reader thread parses the FASTA/FASTQ files one batch ahead of the device (a sample's files are read one
  // after another, io/fof.hpp:82-87); the main thread splits and counts.
  struct ReadBatch { uint32_t si; bool last; std::string bases; std::vector<uint64_t> offs; };
  std::mutex qm; std::condition_variable qcv; std::deque<ReadBatch> rq; bool reader_done = false;
  std::thread reader([&]() {
    auto push = [&](ReadBatch&& b) {
      std::unique_lock<std::mutex> lk(qm);
      qcv.wait(lk, [&]() { return rq.size() < 2; });
      rq.push_back(std::move(b)); qcv.notify_all();
    };
    for (uint32_t si = 0; si < N; si++) {
      ReadBatch b; b.si = si; b.last = false; b.offs.assign(1, 0);
      for (const std::string& f : samples[si].files) {
        SeqReader rd(f); std::string seq;
        auto t0 = clk::now();
        while (rd.next(seq)) {
          b.bases += seq; b.offs.push_back(b.bases.size());
          if (b.bases.size() > (256u << 20)) {
            s_read += since(t0);
            push(std::move(b));
            b = ReadBatch(); b.si = si; b.last = false; b.offs.assign(1, 0);
            t0 = clk::now();
          }
        }
        s_read += since(t0);
      }
      b.last = true; push(std::move(b));
    }
    { std::lock_guard<std::mutex> lk(qm); reader_done = true; } qcv.notify_all();
  });
  struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{reader};
  for (uint32_t si = 0; si < N; si++) {
    const Sample& S = samples[si];
    kmx_ctx* c = gpu.c[si % gpu.c.size()];
    std::vector<std::vector<uint8_t>> streams(P);
    std::vector<uint64_t> nk(P, 0);
    for (;;) {
      ReadBatch b;
      { std::unique_lock<std::mutex> lk(qm); qcv.wait(lk, [&]() { return !rq.empty(); }); b = std::move(rq.front()); rq.pop_front(); qcv.notify_all(); }
      if (b.offs.size() > 1) {
        const auto t0 = clk::now();
        n_bases += b.bases.size();
        std::vector<uint8_t*> ob(P); std::vector<uint64_t> ol(P), ok(P);
        chk(c, kmx_superk_partition(c, b.bases.data(), b.offs.data(), b.offs.size() - 1, o.k, o.msize, table.data(), P, ob.data(), ol.data(), ok.data()), "kmx_superk_partition");
        for (uint32_t p = 0; p < P; p++) { streams[p].insert(streams[p].end(), ob[p], ob[p] + ol[p]); nk[p] += ok[p]; kmx_free(ob[p]); }
        s_split += since(t0);
      }
      if (b.last) break;
    }
    for (uint32_t p = 0; p < P; p++) n_kmers += nk[p];
    { std::ofstream pi(root + "/partition_infos/" + S.id + ".pinfo"); for (uint32_t p = 0; p < P; p++) pi << nk[p] << "\n"; }   // gatb_utils.hpp:46-51
    if (o.keep_tmp || o.until == "superk") {
      const std::string sd = root + "/superkmers/" + S.id; fs::create_directories(sd);
      std::ofstream info(sd + "/SuperKmerBinInfoFile"); info << "skp\n" << sd << "\n" << P << "\n";
      for (uint32_t p = 0; p < P; p++) {
        SuperkBlockWriter w(sd + "/skp." + std::to_string(p), p);
        w.add_stream(streams[p].data(), streams[p].size(), o.k);
        info << w.kmers << "\n" << w.bytes << "\n";       // saved before the final flush, like task.hpp:315-316 (Appendix B-4)
        w.flush();
      }
    }
    if (o.until == "superk") continue;
    const auto t_count = clk::now();
    {   // every partition of the sample in one device pass (CountTask / HashCountTask, task.hpp:367-392, 447-481)
      std::vector<const uint8_t*> sp(P); std::vector<uint64_t> sl(P), pid(P), n(P);
      std::vector<uint64_t*> keys(P, nullptr); std::vector<uint32_t*> cnts(P, nullptr);
      for (uint32_t p = 0; p < P; p++) { sp[p] = streams[p].data(); sl[p] = streams[p].size(); pid[p] = p; }
      chk(c, kmx_count_batch(c, P, sp.data(), sl.data(), o.k, hash_mode ? 1 : 0, hash_mode ? hw.wbits : 0, pid.data(), S.hard_min,
                             keys.data(), cnts.data(), n.data()), "kmx_count_batch");
      for (uint32_t p = 0; p < P; p++) {
        const std::string cp = root + "/counts/partition_" + std::to_string(p) + "/" + S.id + (hash_mode ? ".hash" : ".kmer");
        if (hash_mode) write_hash_file(cp, si, p, keys[p], cnts[p], n[p]);
        else write_kmer_file(cp, o.k, si, p, keys[p], cnts[p], n[p]);
        kmx_free(keys[p]); kmx_free(cnts[p]);
      }
    }
    s_count += since(t_count);
  }
  if (o.until == "superk" || o.until == "count") { report(); return 0; }

  // ---- merge, one task per partition (task_scheduler.hpp:381-417) ----
  Plugin plug; if (!o.plugin.empty()) plug.load(o.plugin, o.k);
  std::vector<uint32_t> soft(N, o.soft_min);
  // the count files of partition p + 1 are read while partition p is merged
  auto load_partition = [&](uint32_t p) {
    std::vector<std::vector<uint8_t>> r(N);
    for (uint32_t i = 0; i < N; i++) {
      const std::string cp = root + "/counts/partition_" + std::to_string(p) + "/" + samples[i].id + (hash_mode ? ".hash" : ".kmer");
      if (!fs::exists(cp)) die(cp + " is missing.");                                  // kmdir.hpp:69-70
      r[i] = hash_mode ? read_hash_records(cp, nullptr) : read_kmer_records(cp, nullptr, nullptr);
    }
    return r;
  };
  std::future<std::vector<std::vector<uint8_t>>> next_recs;
  if (P) next_recs = std::async(std::launch::async, load_partition, 0u);
  for (uint32_t p = 0; p < P; p++) {
    kmx_ctx* c = gpu.of_partition(p);
    auto t_io = clk::now();
    std::vector<std::vector<uint8_t>> recs = next_recs.get(); std::vector<kmx_list> lists(N);
    if (p + 1 < P) next_recs = std::async(std::launch::async, load_partition, p + 1);
    for (uint32_t i = 0; i < N; i++) {
      lists[i].recs = recs[i].data(); lists[i].n = recs[i].size() / ((hash_mode ? 1 : kw) * 8 + 4);
      n_merge_recs += lists[i].n;
    }
    s_merge_io += since(t_io);
    kmx_merge_task t{};
    t.n_lists = N; t.key_words = hash_mode ? 1 : kw; t.lists = lists.data(); t.soft_min = soft.data();
    t.rec_min = o.rec_min; t.share_min = o.share_min; t.bitw = o.bitw;
    t.mode = what == "count" ? KMX_MODE_COUNT : what == "pa" ? KMX_MODE_PA : what == "bf" ? KMX_MODE_BF : KMX_MODE_BFC;
    if (what == "bf" || what == "bfc") { t.lower = hw.lower(p); t.upper = hw.upper(p); }
    km::IMergePlugin* pl = nullptr;
    if (plug.create) {   // the plugin's return value replaces the recurrence test: produce every row, filter on the host
      pl = plug.create(); pl->configure(o.plugin_config);                              // plugin_manager.hpp:106-111
      pl->set_out_dir(root + "/plugin_output"); pl->set_kmer_size(hash_mode ? 0 : o.k); pl->set_partition(p);   // task.hpp:701-712
      t.rec_min = 0; t.mode = KMX_MODE_COUNT;
    }
    void* body = nullptr; uint64_t nbytes = 0, rows = 0; std::vector<uint64_t> stats((size_t)6 * N);
    const auto t_m = clk::now();
    chk(c, kmx_merge(c, &t, &body, &nbytes, &rows, stats.data()), "kmx_merge");
    s_merge += since(t_m);
    t_io = clk::now();
    const std::string ext = what == "count" ? (hash_mode ? "count_hash" : "count") : what == "pa" ? (hash_mode ? "pa_hash" : "pa") : "cmbf";
    Out out(root + "/matrices/matrix_" + std::to_string(p) + "." + ext);
    if (what == "count") { if (hash_mode) matrix_count_hash_header(out, N, p); else matrix_count_header(out, o.k, N, p); }
    else if (what == "pa") { if (hash_mode) matrix_pa_hash_header(out, N, p); else matrix_pa_header(out, o.k, N, p); }
    else matrix_bf_header(out, what == "bf" ? N : N * o.bitw, t.lower, t.upper - t.lower + 1, p);
    if (!pl) out.raw(body, nbytes);
    else {
      const uint32_t kb = t.key_words * 8; const size_t rb = kb + 4ull * N;
      std::vector<km::IMergePlugin::count_type> cv(N); std::vector<uint8_t> pa((N + 7) / 8);
      uint64_t last_key[2] = {0, 0};
      for (uint64_t r = 0; r <= rows; r++) {   // r == rows: the reference's extra call after the last row (merge.hpp:185-259)
        const uint8_t* row = (const uint8_t*)body + r * rb;
        if (r < rows) { memcpy(last_key, row, kb); for (uint32_t i = 0; i < N; i++) { uint32_t v; memcpy(&v, row + kb + 4 * i, 4); cv[i] = (km::IMergePlugin::count_type)v; } }
        else std::fill(cv.begin(), cv.end(), 0);
        const bool keep = hash_mode ? pl->process_hash(last_key[0], cv) : pl->process_kmer(last_key, cv);
        if (r == rows || !keep) continue;
        out.raw(last_key, kb);
        if (what == "count") for (uint32_t i = 0; i < N; i++) { const uint32_t v = (uint32_t)cv[i]; out.raw(&v, 4); }
        else { std::fill(pa.begin(), pa.end(), 0); for (uint32_t i = 0; i < N; i++) if (cv[i]) pa[i >> 3] |= (uint8_t)(1u << (i & 7)); out.raw(pa.data(), pa.size()); }
      }
      plug.destroy(pl);
    }
    kmx_free(body);
    write_merge_info(root + "/merge_infos/partition" + std::to_string(p) + ".merge_info", stats.data(), N);
    if (what == "bf") {   // task.hpp:849-860 + utils.hpp:239-243
      std::ofstream fp(root + "/fpr/partition_" + std::to_string(p) + ".txt");
      for (uint32_t i = 0; i < N; i++) fp << std::fixed << std::pow(1.0 - std::pow(std::exp(1.0), -(double)stats[(size_t)3 * N + i] / (double)hw.wbits), 1.0) << "\n";
    }
    if (!o.keep_tmp)                                                                  // task.hpp:676-688
      for (uint32_t i = 0; i < N; i++) fs::remove(root + "/counts/partition_" + std::to_string(p) + "/" + samples[i].id + (hash_mode ? ".hash" : ".kmer"));
    s_merge_io += since(t_io);
  }
  report();
  struct rusage ru; getrusage(RUSAGE_SELF, &ru);
  { std::ofstream ri(root + "/run_infos.txt");                                         // task_scheduler.hpp:453-457
    ri << "Time: " << std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t0).count() << " seconds\n"
       << "Memory: " << ru.ru_maxrss / 1024 << "MB\n"; }
  return 0;
}
