// kmx_host.hpp -- host-side declarations shared by the translation units of libkmx.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <functional>
#include <vector>
#include "kmx_dev.hpp"
#include "skf.hpp"
#include "../../include/kmx.h"

namespace kmx {

// kernel launchers (defined next to their kernels)
int rows_lds_bytes(int kw, u32 n_lists);
int rows_cap(int kw, u32 n_lists);
int rows_wgs_per_cu(int kw);
u32 rows_chunk_rows(u32 row_bytes);
u32 rows_image_bytes(int kw);
hipError_t launch_range_bounds(int kw, const TaskDev* tasks, u32 n_tasks, u32 max_n, u32 max_c, hipStream_t st);
hipError_t launch_merge_rows(int kw, int mode, const TaskDev* tasks, const uint2* items, u32 n_items, u32* ticket,
                             u32 grid_x, u32 max_n, hipStream_t st);
// the same kernel built for cohorts of up to 256 lists (merge_rows_small.hip: 512 threads, 2048 slots, two workgroups a CU; keys of one and two words)
int rows_s_cap(int kw, u32 n_lists);
int rows_s_wgs_per_cu(int kw);
u32 rows_s_image_bytes(int kw);
hipError_t launch_merge_rows_s(int kw, int mode, const TaskDev* tasks, const uint2* items, u32 n_items, u32* ticket, u32 grid_x, u32 max_n, hipStream_t st);
int pivot_lds_bytes(int kw);
u32 pivot_max_lists();
hipError_t launch_merge_pivot(int kw, int mode, const TaskDev* tasks, const uint2* items, u32 n_items, u32* ticket,
                              u32 grid_x, hipStream_t st);
// the column-blocked merge is compiled once per key width (merge_cols.hip, merge_cols_k2.hip): the entry points of a build
struct ColsOps {
  int (*lds_bytes)();
  u32 (*block_lists)();
  u32 (*wgs_per_cu)();
  u32 (*tile_rows)(u32 nb);
  u64 (*scratch_keys)(u32 slots, u32 nblk);       // u64 words of the set-aside entries
  u64 (*scratch_counts)(u32 slots, u32 nblk);
  u64 (*ext_entries)(u32 slots, u32 nblk);        // entries of the pool the slices' extensions come from
  u32 (*skel_cap)();
  hipError_t (*skel)(const TaskDev* subs, const uint2* items, u32 n_items, hipStream_t st);
  hipError_t (*prep)(const TaskDev* tasks, const TaskDev* subs, const ColsDev* cols, u32 n_tasks, hipStream_t st);
  hipError_t (*merge)(int mode, int ext, const TaskDev* tasks, const ColsDev* cols, const uint2* items, u32 n_items, u32* ticket, u32 grid_x, hipStream_t st);
  hipError_t (*sparse)(int mode, const TaskDev* tasks, const ColsDev* cols, const uint2* range_items, u32 n_items, u32 n_tasks, u32* ticket, u32 n_cu, hipStream_t st);
  u64 (*dir_bytes)(u32 slots);
  u32 (*groups)(u32 slots);
  hipError_t (*offsets)(const ColsDev* cols, u32 task, u64* goff, hipStream_t st);
  hipError_t (*gather)(const TaskDev* tasks, const ColsDev* cols, u32 task, u32 n_groups, const u64* goff, u8* body, hipStream_t st);
  hipError_t (*order)(const TaskDev* tasks, const ColsDev* cols, u32 task, u32 n_groups, const u64* goff, u32* order, hipStream_t st);
  void (*dbg_dump)();
  void (*phase_prof_dump)();
  u32 key_words;
};
const ColsOps& cols_ops_k1();
const ColsOps& cols_ops_k2();
inline const ColsOps& cols_ops(int kw) { return kw == 2 ? cols_ops_k2() : cols_ops_k1(); }
int bf_lds_bytes(u32 rt, u32 nb, u32 n_lists);
hipError_t launch_range_bounds_bf(const TaskDev* tasks, u32 n_tasks, u32 max_n, u32 max_c, hipStream_t st);
hipError_t launch_merge_bf(int bfc, const TaskDev* tasks, const uint2* items, u32 n_items, u32* ticket,
                           u32 grid_x, int lds, hipStream_t st);
u32 bft_tile_rows(u32 rec_min, u32 share_min);
u32 bft_block_lists();
u32 bft_max_lists();
u32 bft_round_records(bool wide);
u32 bft_fit_rows(u32 max_n, bool bits);
hipError_t launch_merge_bft(const TaskDev* tasks, const uint2* items, u32 n_items, u32* ticket, u32 grid_x, u32 max_n, u32 rt_max, bool rec_bits, bool wide, u64* rem, u32 rem_cap, hipStream_t st);
hipError_t launch_bit_transpose(const u8* in, u8* out, u64 nrows, u64 ncols, hipStream_t st);

}  // namespace kmx

// ---- split -> count without the super-k-mer streams leaving HBM (kmx_count_reads): superk.hip hands the packed, partition-ordered
//      record stream, every record's (k-mers << 32 | bytes) prefix and partition to count.hip ----
//      Results: host arrays (keys / counts / n_out, kmx_count_reads) or packed records in device stores (lists, kmx_count_reads_dev).
struct kmx_count_req { kmx::u32 k; int hash_mode; kmx::u64 window; kmx::u32 hard_min; uint64_t** keys; uint32_t** counts; uint64_t* n_out;
                       kmx_store* const* stores = nullptr; kmx::u32 n_stores = 0; kmx_list* lists = nullptr;
                       kmx::u32 inner_parts = 0; };      // several samples in one call: partition p' = sample * inner_parts + p (store and window id from p)
int kmx_count_from_device(kmx_ctx* ctx, const kmx::u8* d_recs, const kmx::u64* d_prefix, const kmx::u16* d_part, kmx::u32 n_recs,
                          kmx::u64 total_kmers, kmx::u32 n_parts, const kmx::u64* part_kmer_off /* n_parts + 1, host */, const kmx_count_req& rq,
                          const kmx::u32* d_sbase = nullptr /* set: no record stream -- d_recs are the batch's bases packed by kmx_launch_pack_bases, record i starts at base d_sbase[i] */);
void kmx_launch_pack_bases(const char* d_bases, kmx::u64 n, kmx::u64* out /* (n + 31) / 32 + 2 words */, hipStream_t st);
// ---- round 6: the same hand-over without a host round trip (superk_fast.hpp -> count.hip).  Everything is on the device, sizes
//      included (d_ctl); the arrays are sized for bounds.  kmx_count_fast_tail queues decode + partition-local sample sort + count,
//      reads the control block, the sample sort's tables and the kept sizes back ONCE, and packs the lists into the stores.
//      Returns KMX_OK, a negative error, or 1: a status bit was raised on the device (h_ctl holds it) -- the caller takes the old path.
struct kmx_fast_split {
  const kmx::u64* d_words;        // the batch's bases, 2 bits each (kmx_launch_pack_bases)
  const kmx::u32* d_sbase;        // [nd] first base of sorted record i
  const kmx::u64* d_boff;         // [nd + 1] k-mers << 32 | record bytes in front of sorted record i
  const kmx::u16* d_part16;       // [nd] its partition (hash mode; else null)
  const kmx::u32* d_blk;          // first record of every block of SKF_DK k-mers
  kmx::SkfCtl* d_ctl;             // control block; 8 more bytes behind it ride along in the read-back (64 bytes in all)
  const uint4* d_parts;           // CsPart[n_parts]: the sample sort's layout (k_sk_scan)
  const kmx::u32* d_cfirst;       // [n_parts + 1] first walk chunk of every partition
  kmx::u32* d_cnt;                // [tb_max + 2] zeroed: the buckets' counters
  kmx::u32* d_sflags;             // [2 * 256] zeroed: the flags of the two bucket scans (k_cs_scan_mw)
  kmx::u32 n_parts; kmx::u64 kmer_bound; kmx::u32 tb_max, nc_max, nb_max;      // bounds: k-mers, buckets, walk chunks, decode blocks
  kmx::SkfCtl* h_ctl;             // page-locked, 64 bytes: the control block as read back
  const uint4* h_parts;           // page-locked: d_parts as read back (part of the one copy below)
  unsigned long long* d_strand;   // [kmer_bound / 64 + 2] or null: the decode leaves bit g = "k-mer g of the batch is its own canonical form" (the PartiInfo statistics' strands)
  std::function<int()> behind_scatter;   // run right after the scatter walk is queued: the caller starts its work on the second stream there (the kernels up to the scatter fill the LDS, the wave sort behind it uses none)
  std::function<int()> before_wait;      // run once everything is queued, before the stream is waited for (the caller's work on its second stream)
  // the call's ONE read-back: back_bytes from d_ctl on -- the control block, the caller's tables (h_parts among them) and, at d_koff / h_koff,
  // the kept pairs' offsets per bucket ([tb_max + 2]) that kmx_count_fast_tail fills -- to h_ctl on; queued by kmx_count_fast_tail at its end
  kmx::u32* d_koff; kmx::u32* h_koff; size_t back_bytes;
};
kmx::SkfLayout kmx_fast_layout(int key_words /* of the sort's keys: 1 (k <= 32, hashes) or 2 */);
int kmx_count_fast_tail(kmx_ctx* ctx, const kmx_fast_split& F, const kmx_count_req& rq);
void kmx_phase_mark(int i);      // KMX_COUNT_PHASES (superk.hip)
// the count calls of one GPU's contexts (the pipeline runs two or three workers a GPU, a context and a stream each) in a CHAIN: a call's
// kernels start when the call queued before it -- whichever context's -- has left the GPU.  Side by side two calls' kernels share the
// CUs, finish together, and their hosts then read back, pack and queue the next call at the same time with the GPU idle (36-47 % of
// the count stage, profiles/r06_pipeline_count_trace.txt); chained, one call's host work lies under the other's kernels.
void kmx_count_chain_begin(kmx_ctx* ctx);      // takes the GPU's chain lock, makes ctx->stream wait for the last queued call
void kmx_count_chain_end(kmx_ctx* ctx);        // records this call's end on ctx->stream as the chain's new tail, releases the lock
void kmx_count_chain_forget(kmx_ctx* ctx);     // (kmx_destroy)

// page-locked host memory (kmx_api.hip: transparent huge pages + hipHostRegister for blocks of 2 MB and more, hipHostMalloc else)
int kmx_peer_path(int from, int to);      // 1: GPU `from` reaches GPU `to`'s memory directly (peer access enabled on first use), 0: staged
void* kmx_pinned_alloc(size_t bytes);
bool kmx_is_pinned(const void* p, size_t n);      // inside a page-locked block of kmx_alloc_pinned's (the mapped + registered kind)
void kmx_pinned_free(void* p);

// ---- context -------------------------------------------------------------------------------------
struct kmx_pool_block { void* p; size_t bytes; bool used; };

// ---- device arena of count lists (kmx_store_*): chunks of HBM on one GPU, bump-allocated, freed together ----
#include <mutex>
#include <thread>
#include <condition_variable>
struct kmx_store {
  int device = 0;
  size_t limit = 0, used = 0, chunk_bytes = 0;
  struct Chunk { kmx::u8* p; size_t cap, fill; };
  std::vector<Chunk> chunks;
  std::mutex mu;
  void* alloc(size_t bytes);      // 256-byte aligned; nullptr: over the limit or out of device memory
  // round 6: a count call that does not know its lists' size yet takes room for an estimate, has its kernel write there, and gives
  // back what it did not need once the size is read back.  A reservation is the tail of a chunk, one per chunk; while it is open alloc()
  // and the other reservations leave that chunk alone -- the calls of a GPU's workers run side by side, each with its own (up to 8 open)
  struct Resv { int chunk; size_t off, bytes; };
  std::vector<Resv> resvs;
  bool chunk_reserved(size_t i) const { for (auto& r : resvs) if (r.chunk == (int)i) return true; return false; }
  // a chunk made ahead of its need by a thread of the store's own (round 6): on a box whose HBM has not been touched since boot a
  // hipMalloc of 256 MB takes ~8 ms (the driver clears it), 223 of them for configs[2]'s lists -- inside count calls, with the GPU idle
  std::vector<Chunk> spare; std::thread ahead; std::condition_variable cv; bool stop = false, ahead_on = false;
  void start_ahead();      // (first allocation; KMX_STORE_AHEAD=0: never)
  bool take_spare(size_t bytes, Chunk& out);      // with mu held
  void* try_reserve(size_t bytes);              // nullptr: eight reservations are open, or no room
  void commit(void* p, size_t used_bytes);      // keeps the first used_bytes of the reservation (0: none of it)
};

struct kmx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool stream_shared = false;           // the caller asked for the stream (kmx_stream): its own work may be queued on it
  hipStream_t copy = nullptr;           // third stream: results are read back without queueing behind later batches
  hipStream_t up = nullptr;             // uploads of kmx_merge_host: the next batch's lists travel while this batch merges
  hipEvent_t copy_ev[8] = {}; bool copy_out[8] = {};      // kmx_copy_to_host_async: tickets
  struct ReadsAhead { char* d = nullptr; size_t bytes = 0; hipEvent_t ev = nullptr; bool live = false; };      // kmx_reads_upload
  ReadsAhead ahead[4];
  hipStream_t aux = nullptr;            // second stream: meta uploads, scratch clears (and, with KMX_COLS_PREP_OVERLAP, the small kernels that prepare a batch)
  hipEvent_t ev_chain = nullptr;        // kmx_count_reads_dev: this context's last call has left the GPU (kmx_count_chain_*: the calls of a GPU's contexts run one behind the other)
  hipEvent_t ev_split = nullptr;        // kmx_count_reads_dev: the split is through -- the PartiInfo statistics run on `aux` beside the count kernels from here
  int n_cu = 0;
  std::string err;
  bool profiling = false;
  std::vector<kmx_pool_block> pool;     // device blocks kept for reuse (bench steps allocate nothing)
  size_t big_max = 0;                   // the largest block of 64 MB or more asked for so far (dalloc)
  std::vector<kmx_pool_block> hpool;    // pinned host blocks
  // the pivot merge kernel handed a batch back: the next `pivot_skip` eligible batches go straight to k_merge_rows
  // (doubling back-off, reset by the first batch the pivot kernel completes)
  unsigned pivot_backoff = 0, pivot_skip = 0;
  unsigned cols_backoff = 0, cols_skip = 0;   // the same for the column-blocked kernel
  bool cols_min_env = false;
  unsigned cols_min_lists = 192, cols_min_lists_ord = 257;   // lists per task from which libkmx picks the column-blocked pair (rows left where they fall / in file order: k_merge_rows' windows halve above 256 lists, and up to there it is the faster one, profiles/r04_crossover.txt); KMX_COLS_MIN_LISTS[_ORD]
  // rows kept per record of a task's longest list, as the batches completed so far had it (COUNT/PA arenas are sized from it:
  // a cohort whose samples share their private k-mers pairwise keeps several times more rows than a list is long, and an arena
  // sized for 2 x the longest list would make every batch run twice)
  double rows_per_longest = 0.0;
  bool cols_ext = false;                // a batch was handed back for full set-aside slices: k_merge_cols with slice extensions from here on
  // COUNT / PA rows of the column-blocked pair come out in file order (the matrix body as the reference streams it,
  // merge.hpp:262-272): kmx_set_file_order, KMX_FILE_ORDER=0 for the rows where the kernels leave them + a directory
  bool file_order = true;
  double hash_lost_frac = 0.0;          // kmx_count_reads_dev: the share of the last call's buckets whose distinct keys did not fit k_cs_wave_count's tables (a quarter: the next call sorts)
  double kept_per_kmer = 0.0;           // kmx_count_reads_dev: distinct kept k-mers per k-mer as the last call had it (the next one reserves room in the store for 1.25 x that)
  double keys_per_longest = 0.0;        // row keys per record of the longest list they were merged from, as completed batches had it
  // abundance histogram (kmx_hist_reset / kmx_hist_read): distinct keys per count 0..255, [256] = keys counted more than 255
  // times, [257] = the sum of those counts.  Every count call adds to it while it is on.
  unsigned long long* d_hist = nullptr;
  bool hist_on = false;
  // the minimizer -> partition table of the split stays on the device between calls: a sample after a sample hands the same table
  // (2 MB at m = 10) -- re-uploaded only when its address, size or digest (every entry folded: kmx_rep_digest) changes
  kmx::u16* d_rep = nullptr; const void* rep_host = nullptr; size_t rep_n = 0; kmx::u64 rep_digest = 0;
  // the statistics tables of kmx_superk_raw's sparse mode stay with the context: the kernel that compacts the per-minimizer tables
  // puts the entries it read back to zero, so a call clears only the partitions' counters (1.3 MB instead of 9.3 MB at m = 10)
  kmx::u32* d_stat = nullptr; size_t stat_cap = 0, stat_parts = 0, stat_nm = 0; bool stat_dirty = true;

  void* dalloc(size_t bytes);
  void dfree(void* p);
  void* halloc(size_t bytes);
  void hfree(void* p);
  int fail(int code, const std::string& msg) { err = msg; return code; }
};

#define KMX_HIP(ctx, call)                                                                         \
  do {                                                                                             \
    hipError_t e__ = (call);                                                                       \
    if (e__ != hipSuccess)                                                                         \
      return (ctx)->fail(KMX_E_HIP, std::string(#call) + ": " + hipGetErrorString(e__));           \
  } while (0)

#include <chrono>
#include <cstdio>
#include <cstdlib>
struct StageClock {     // KMX_TRACE=1: per-stage wall times of the batched count on stderr
  const char* name; bool on; hipStream_t st; std::chrono::steady_clock::time_point t0; std::string log;
  StageClock(hipStream_t s, const char* nm = "count_batch") : name(nm), on(getenv("KMX_TRACE") != nullptr), st(s), t0(std::chrono::steady_clock::now()) {}
  void mark(const char* what) {
    if (!on) return;
    (void)hipStreamSynchronize(st);
    auto t1 = std::chrono::steady_clock::now();
    char b[96]; snprintf(b, sizeof b, " %s=%.2fms", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    log += b; t0 = t1;
  }
  ~StageClock() { if (on) fprintf(stderr, "[kmx %s]%s\n", name, log.c_str()); }
};

