/*
 * kmx.h -- C ABI of the MI355X-native kmtricks counting/merge engine (libkmx.so).
 *
 * This is the drop-in boundary: the entry points below are what the
 * reference's tasks would bind instead of running their CPU classes
 * (file:line relative to the kmtricks source tree):
 *
 *   kmx_merge / kmx_merge_dev   replace km::KmerMerger<MAX_K,MAX_C>::{next,write_as_bin,write_as_pa}
 *                               (include/kmtricks/merge.hpp:102-361) and km::HashMerger<MAX_C>::
 *                               {next,write_as_bin,write_as_pa,write_as_bf,write_as_bfc}
 *                               (merge.hpp:363-629), called from KmerMergeTask::exec / HashMergeTask::exec
 *                               (include/kmtricks/task.hpp:690-743, 787-863)
 *   kmx_count_kmer              replaces km::KmerPartCounter::execute + KmerCountProcessor::process
 *                               (include/kmtricks/gatb/sorting_count.hpp:637-884,
 *                                include/kmtricks/gatb/count_processor.hpp:135-146), CountTask::exec (task.hpp:367-392)
 *   kmx_count_hash              replaces km::HashPartCounter::execute + HashCountProcessor::process
 *                               (sorting_count.hpp:346-363, 908-997; count_processor.hpp:61-70),
 *                               HashCountTask::exec (task.hpp:447-481)
 *   kmx_transpose_bits          replaces km::BitMatrix::transpose / __sse_trans
 *                               (include/kmtricks/bitmatrix.hpp:209-214, 238-289); HashMerger::write_as_bft (merge.hpp:631-644)
 *                               as a whole is kmx_merge* with KMX_MODE_BFT (merge + transpose without leaving HBM)
 *   kmx_superk_partition        replaces KmFillPartitions / Sequence2SuperKmer / SuperKmer::save
 *                               (include/kmtricks/gatb/fill_partitions.hpp:59-105, gatb kmer/impl/Sequence2SuperKmer.hpp:80-158,
 *                                gatb kmer/impl/Model.hpp:1086-1139, 1388-1433), SuperKTask::exec (task.hpp:255-320)
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success or a negative KMX_E_* code, with a message in kmx_last_error().
 * There is NO CPU fallback: without a HIP device kmx_create fails with
 * KMX_E_NODEVICE.  All integers little-endian.  A ctx is used by one host
 * thread at a time (one ctx per pool thread, like the reference's tasks).
 */
#ifndef KMX_H
#define KMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: kmx_merge_task carries list_on_device (the struct grew: a caller built against version 1 hands tasks of the wrong stride);
 *    kmx_set_file_order.  Callers check kmx_version() == KMX_VERSION before anything else. */
#define KMX_VERSION 2

enum {
  KMX_OK = 0,
  KMX_E_NODEVICE = -1,   /* no HIP device / HIP runtime error at init */
  KMX_E_INVAL    = -2,   /* bad argument */
  KMX_E_NOMEM    = -3,   /* host or device allocation failed */
  KMX_E_HIP      = -4,   /* HIP runtime error during a call */
  KMX_E_UNSUPPORTED = -5 /* configuration outside what this build handles (message says which) */
};

/* matrix row encodings; names follow kmtricks' --mode <kmer|hash>:<count|pa|bf|bfc>:bin */
enum {
  KMX_MODE_COUNT = 0,  /* row = key + N * u32           (.count / .count_hash body) */
  KMX_MODE_PA    = 1,  /* row = key + ceil(N/8) bytes   (.pa / .pa_hash body)       */
  KMX_MODE_BF    = 2,  /* one ceil(N/8)-byte row per hash of [lower, upper] (.cmbf) */
  KMX_MODE_BFC   = 3,  /* one ceil(N*w/8)-byte row per hash, bitpacker MSB-first    */
  KMX_MODE_BFT   = 4   /* the BF matrix bit-transposed on the device: round_up8(N) rows (row s = sample s) of
                          round_up8(W)/8 bytes -- what HashMerger::write_as_bft dumps (merge.hpp:631-644), and
                          the layout the per-sample Bloom filter files are cut from (howde_utils.hpp:133-187) */
};

typedef struct kmx_ctx kmx_ctx;

int  kmx_version(void);
int  kmx_device_count(void);   /* HIP devices visible to this process (0: none -- libkmx has no CPU fallback) */
/* free and total bytes of a device's memory (hipMemGetInfo) */
int  kmx_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes);
/* Round 6.  Takes `bytes` of the device's memory in blocks of 1 GiB, holds them all, and gives them back: on a box whose HBM has not
 * been used since boot the driver clears memory the first time it is handed out (~30 us a MB: 8 ms for a 256 MB store chunk, 18 ms
 * for a 600 MB row arena, with the GPU idle behind the allocating call) and not again afterwards.  `kmx pipeline` calls it on a
 * thread of its own while the first samples are read (KMX_WARM_GB, default 96, 0: never); the count and merge stages' own
 * allocations then find memory that is handed out at once.  Returns the bytes it took (0: nothing to do, or no memory to spare). */
uint64_t kmx_device_warm(int device, uint64_t bytes, const volatile int* stop /* or NULL: checked before every block; non-zero ends the call */);
/* (it times its first block: memory that comes at once -- under 5 ms a GiB -- has been handed out before, and the call returns) */
int  kmx_create(int device, kmx_ctx** out);
void kmx_destroy(kmx_ctx* ctx);
/* last error message of this ctx (or of the failed kmx_create when ctx == NULL) */
const char* kmx_last_error(const kmx_ctx* ctx);
/* When on, the merge driver brackets its dominant kernel with HIP events on the ctx stream so that bench.py can report
 * the kernel's launch duration.  Which kernel a batch runs (libkmx chooses; KMX_MERGE_KERNEL=rows|pivot|cols forces one):
 *   COUNT / PA   k_merge_cols + k_cols_sparse  every task has >= 192 lists (>= 257 for count rows with 64-bit keys in file
 *                                              order, >= 160 / 96 for PA rows with 128-bit keys in file order / not;
 *                                              KMX_COLS_MIN_LISTS[_ORD]), recurrence-min <= 21, share-min <= max(1,
 *                                              recurrence-min), >= 1 M records in the batch; 64- and 128-bit keys
 *                k_merge_pivot                 otherwise, tasks of more than 512 lists, 64-bit keys, no share-min
 *                k_merge_rows                  everything else -- and the tasks the two above hand back (lists that do not
 *                                              resemble each other): results never depend on the choice
 *   BF / BFC     k_merge_bf;   BFT  k_merge_bft */
int  kmx_set_profiling(kmx_ctx* ctx, int on);
/* COUNT / PA rows in FILE ORDER out of the merge itself (default on; KMX_FILE_ORDER=0 in the environment or on = 0 here turns it
 * off).  The reference's output IS the ascending row stream (merge.hpp:262-272 write_as_bin -> io/matrix_file.hpp:120-127).  On:
 * the column-blocked pair writes every row at its final place -- k_cols_sparse learns each slice group's offset by a decoupled
 * look-back and puts the group's rows, the row keys' rows among them, in key order there: the arena IS the matrix body
 * (kmx_result_body_dev returns it, no second copy, no gather pass).  Off: the row keys' rows first, the other rows behind them in
 * runs with a directory (kmx_result_arena + kmx_result_copy_order; kmx_result_body_dev then assembles a copy on the device). */
int  kmx_set_file_order(kmx_ctx* ctx, int on);
/* HIP stream the ctx launches on (a hipStream_t), so callers can order their own work after it */
void* kmx_stream(kmx_ctx* ctx);

/* ------------------------------------------------------------------ merge */

/* One sample's sorted count list of one partition: `n` packed records of
 * key_words*8 key bytes (low word first) + a u32 count = the body of a
 * .kmer count file written with 4-byte counts (io/kmer_file.hpp:102-108).
 * Keys strictly ascending (most significant word first, kmer.hpp:262-268). */
typedef struct {
  const void* recs;
  uint64_t    n;
} kmx_list;

/* One merge task = one partition (merge.hpp:115-125, 376-385). */
typedef struct {
  uint32_t        n_lists;     /* N samples, fof order = column order (kmdir.hpp:65-72).
                                  LIMIT (a deliberate deviation from KmerMerger, which takes any vector of paths, merge.hpp:115-125): a COUNT / PA
                                  task takes at most 4096 lists (3072 with keys of four words) -- a row's cursors and its image live in one
                                  workgroup's LDS; a hash:bft task at most 18396 (the cursors of every sample beside the tile).  Beyond:
                                  KMX_E_UNSUPPORTED, nothing is merged.  BASELINE's largest cohort is 2500.  (BF / BFC windows: no limit of that kind.) */
  uint32_t        key_words;   /* ceil(k / 32) (kmer.hpp:215 m_n_data, io/kmer_file.hpp:84 kmer_slots): 1 for k <= 32 and hash keys, 2 up to 64,
                                  3 up to 96, 4 up to 128.  3 and 4 (the reference's Kmer<96> / Kmer<128>): COUNT / PA rows, at most 4096 (three words) / 3072 (four) lists a task */
  const kmx_list* lists;       /* [n_lists] */
  const uint32_t* soft_min;    /* [n_lists] per-sample abundance min (m_a_min_vec) */
  uint32_t        rec_min;     /* recurrence-min (m_r_min) */
  uint32_t        share_min;   /* share-min / save_if (m_save_if), 0 = no rescue */
  uint32_t        mode;        /* KMX_MODE_* */
  uint32_t        bitw;        /* BFC bits per count (--bitw), else ignored */
  uint64_t        lower, upper;/* BF/BFC: first and last hash of the window (hash.hpp:77-85) */
  uint64_t        rows_hint;   /* COUNT/PA: expected kept rows (0 = let the engine guess) */
  const uint8_t*  list_on_device; /* kmx_merge_host only: NULL = every lists[i].recs is a host pointer; else [n_lists],
                                     non-zero = lists[i].recs is a DEVICE pointer already (a list of a kmx_store) and is
                                     merged where it lies.  Ignored by kmx_merge_dev / kmx_merge. */
} kmx_merge_task;

/* statistics layout: 6 * n_lists u64, rows in the order of merge.hpp:72-83:
 * NON_SOLID, RESCUED, UNIQUE_WO_RESCUE, UNIQUE_W_RESCUE, TOTAL_WO_RESCUE, TOTAL_W_RESCUE */
#define KMX_STATS_ROWS 6

typedef struct kmx_merge_result kmx_merge_result;

/* Device-resident batch merge: every lists[i].recs is a DEVICE pointer (4-byte
 * aligned) whose records are complete, or produced by work queued on kmx_stream(ctx)
 * (libkmx prepares a batch on a second stream of the context; once kmx_stream has been
 * called that stream is ordered behind the first).  Enqueues the whole batch and returns; the
 * result stays in HBM until freed.  COUNT/PA rows are produced in row segments
 * that kmx_result_* hands back in ascending key order. */
int kmx_merge_dev(kmx_ctx* ctx, const kmx_merge_task* tasks, uint32_t n_tasks, kmx_merge_result** out);
/* blocks until the batch has finished on the GPU; returns KMX_OK or the error of the run */
int      kmx_result_wait(kmx_merge_result* r);
/* duration in ms of the batch's merge kernel launch (needs kmx_set_profiling(ctx, 1)); < 0 if unavailable */
double   kmx_result_kernel_ms(kmx_merge_result* r);
/* the same launch in its two kernels when the column-blocked pair produced the result: k_merge_cols (the column blocks' walk over the
 * lists), then k_cols_sparse (the rows of the keys outside the row keys; with kmx_set_file_order on also the row keys' rows at their
 * final place).  -1 in both for every other kernel, without profiling, or when tasks were re-run behind the pair. */
int      kmx_result_kernel_parts_ms(kmx_merge_result* r, double* first_ms, double* second_ms);
/* name of the device kernel that produced (most of) the result: "k_merge_cols", "k_merge_pivot", "k_merge_rows",
 * "k_merge_bf" or "k_merge_bft" (see kmx_set_profiling for when each is chosen); valid after kmx_result_wait (tasks a cohort
 * kernel handed back count for the kernel that completed them) */
const char* kmx_result_kernel(const kmx_merge_result* r);
/* duration in ms of a separate transpose pass behind the merge; < 0 when there is none (KMX_MODE_BFT results come out
 * of k_merge_bft sample-major already: kmx_result_kernel_ms covers k_bf_rowrec + k_merge_bft) */
double   kmx_result_transpose_ms(kmx_merge_result* r);
/* DEVICE pointer to the task's body in file order (rows * row_bytes bytes), valid until kmx_result_free.  BF / BFC / BFT
 * bodies are dense as the kernel leaves them; COUNT / PA rows are put in ascending key order on the device the first time
 * the body is asked for (rows of k_merge_rows / k_merge_pivot lie in arena segments, those of k_merge_cols in two ascending
 * lists): a caller can then bring it to the host in pieces (kmx_copy_to_host) or send it from where it lies. */
const void* kmx_result_body_dev(kmx_merge_result* r, uint32_t task);
/* queues that ordering for a COUNT / PA task without waiting for it (a writer asks for task i + 1 before it brings task i's body
 * over: the pass hides behind the copies); kmx_result_body_dev / kmx_result_copy_body wait for it.  No-op for Bloom results. */
int kmx_result_prepare_body(kmx_merge_result* r, uint32_t task);
uint64_t kmx_result_rows(const kmx_merge_result* r, uint32_t task);        /* kept rows (COUNT/PA), window rows (BF/BFC), round_up8(N) (BFT) */
/* COUNT/PA results of k_merge_cols: how many of the task's rows came out of k_cols_sparse (keys outside the row keys the
 * column blocks are built on: sample-private k-mers, k-mers a few samples share); 0 for the other kernels */
uint64_t kmx_result_sparse_rows(const kmx_merge_result* r, uint32_t task);
uint64_t kmx_result_row_bytes(const kmx_merge_result* r, uint32_t task);
uint64_t kmx_result_body_bytes(const kmx_merge_result* r, uint32_t task);  /* rows * row_bytes */
/* algorithmic bytes moved for this task: input records + output rows (DESIGN.md roofline) */
uint64_t kmx_result_algo_bytes(const kmx_merge_result* r, uint32_t task);
/* copies the matrix file body (rows in ascending key order, exactly what the
 * reference's writer streams after its header) into host memory -- by DMA straight into host_dst when that is
 * page-locked (kmx_alloc_pinned), through a pinned staging buffer otherwise */
int kmx_result_copy_body(kmx_merge_result* r, uint32_t task, void* host_dst, uint64_t dst_bytes);
/* COUNT / PA results WITHOUT any device-side pass over the rows: the arena as the kernels left it (arena_rows rows of row_bytes,
 * some unused) and the order of its rows -- host_order[d] (kmx_result_rows entries) = the arena row that is row d of the body.
 * What a file writer takes: it brings the arena to the host in pieces (kmx_copy_to_host) and writes every run of rows at its
 * place (pwrite at d * row_bytes): the file order comes to exist in the file, never as a second copy of the matrix in HBM. */
int kmx_result_arena(kmx_merge_result* r, uint32_t task, const void** dev_arena, uint64_t* arena_rows);
int kmx_result_copy_order(kmx_merge_result* r, uint32_t task, uint32_t* host_order);
/* the same body into DEVICE memory of the caller (e.g. a buffer an RCCL collective sends from) */
int kmx_result_copy_body_dev(kmx_merge_result* r, uint32_t task, void* dev_dst, uint64_t dst_bytes);
int kmx_result_copy_stats(kmx_merge_result* r, uint32_t task, uint64_t* host_stats /* 6 * n_lists */);
void kmx_result_free(kmx_merge_result* r);

/* kmx_merge_dev with HOST list pointers (what a merge task that has just read its count files holds): the
 * lists are uploaded on a stream of their own -- in ONE copy when they lie back to back in one buffer, best a
 * pinned one (kmx_alloc_pinned) -- so a batch travels while the previous one merges; the host buffers may be
 * reused once kmx_result_wait has returned.  Results are read with the kmx_result_* calls above. */
int kmx_merge_host(kmx_ctx* ctx, const kmx_merge_task* tasks, uint32_t n_tasks, kmx_merge_result** out);
/* page-locked host memory for list buffers / result bodies (plain malloc memory works too, slower) */
void* kmx_alloc_pinned(size_t bytes);
void  kmx_free_pinned(void* p);

/* Host-buffer convenience around kmx_merge_dev for ONE task: lists[i].recs are
 * HOST pointers; uploads, merges, returns the body in a buffer to release with
 * kmx_free.  stats may be NULL. */
int kmx_merge(kmx_ctx* ctx, const kmx_merge_task* task, void** body, uint64_t* body_bytes,
              uint64_t* rows, uint64_t* stats);

/* ------------------------------------------------------------------ count */

/* superk: concatenated super-k-mer records [u8 n][2-bit nts] of one
 * (sample, partition) with the u32 block-size framing of skp.<p> removed
 * (io/superk_storage.hpp:215-225).  HOST pointers.  Output: ascending
 * canonical k-mers (key_words = ceil(k/32) words each) with
 * count >= hard_min, saturated to u32; buffers released with kmx_free.
 * 8 <= kmer_size <= 127: the reference's default KMER_LIST "32 64 96 128" (CMakeLists.txt:25-27; loop_executor.hpp:47-63 picks
 * the first entry above k, and that type's width bounds a record: 28 k-mers for k < 32, 60 below 64, 92 below 96, 124 beyond --
 * Sequence2SuperKmer.hpp:146).  A record that claims more is refused (KMX_E_INVAL), not decoded. */
int kmx_count_kmer(kmx_ctx* ctx, const uint8_t* superk, uint64_t len, uint32_t kmer_size,
                   uint32_t hard_min, uint64_t** keys, uint32_t** counts, uint64_t* n_out);
/* window hashes XXH64(words, 8*ceil(k/32), 0) % window + window * partition */
int kmx_count_hash(kmx_ctx* ctx, const uint8_t* superk, uint64_t len, uint32_t kmer_size,
                   uint64_t window, uint64_t partition, uint32_t hard_min,
                   uint64_t** hashes, uint32_t** counts, uint64_t* n_out);

/* Batched form (one call per sample instead of one per (sample, partition)): superk[p] / len[p] are the
 * n_parts partition streams of one sample; hash_mode != 0 selects window hashes with
 * partition_ids[p] as the window index of stream p.  keys / counts / n_out are arrays of n_parts
 * entries; every keys[p] and counts[p] is released with kmx_free. */
int kmx_count_batch(kmx_ctx* ctx, uint32_t n_parts, const uint8_t* const* superk, const uint64_t* len,
                    uint32_t kmer_size, int hash_mode, uint64_t window, const uint64_t* partition_ids,
                    uint32_t hard_min, uint64_t** keys, uint32_t** counts, uint64_t* n_out);

/* -------------------------------------------------------------- transpose */

/* out[c][r] = in[r][c], bits LSB-first in each byte; nrows, ncols multiples
 * of 8; in row stride ncols/8 bytes, out row stride nrows/8 bytes.  HOST pointers. */
int kmx_transpose_bits(kmx_ctx* ctx, const uint8_t* in, uint64_t nrows, uint64_t ncols, uint8_t* out);

/* ------------------------------------------------------ super-k-mer split */

/* Splits `n_seqs` reads (concatenated in `bases`, read i = bases[offsets[i] .. offsets[i+1]))
 * into super-k-mers and returns, per partition, the concatenated 2-bit records
 * (same bytes the reference buffers before block framing).  repart: u16[4^m]
 * minimizer -> partition table.  out_bytes[p] / out_len[p] / out_kmers[p] are
 * arrays of nb_parts entries; each out_bytes[p] is released with kmx_free.
 * 8 <= kmer_size <= 127, 4 <= minim_size <= 15.  From k = 64 on (Kmer<96> / Kmer<128>) the split, its statistics, the sampling pass
 * kmx_count_reads and kmx_count_reads_dev work as below that; kmx_count_reads_dev_multi (several samples a call) answers
 * KMX_E_UNSUPPORTED there. */
int kmx_superk_partition(kmx_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n_seqs,
                         uint32_t kmer_size, uint32_t minim_size, const uint16_t* repart,
                         uint32_t nb_parts, uint8_t** out_bytes, uint64_t* out_len, uint64_t* out_kmers);

/* The statistics the reference keeps beside the split (gatb PartiInfo<5>): every array is optional (NULL = not wanted)
 * and is ADDED to, so that the batches of a sample -- or the samples of a cohort -- accumulate:
 *   part_counters  [nb_parts * KMX_PINFO_STRIDE]: per partition nb_kmers, nb_kxmers, then nbk_per_radix[x * 256 + radix]
 *                  for x = 0..4 (kx-mers of x + 1 k-mers; fill_partitions.hpp:67-102, PartiInfo.hpp:266-287 PartiInfoFile order)
 *   minim_superks / minim_kmers [4^m]: super-k-mers and k-mers per minimizer (incSuperKmer_per_minimBin)
 *   minim_kxmers   [4^m]: kx-mers per minimizer -- what the sampled repartition balances
 *                  (SampleRepart, gatb RepartitionAlgorithm.cpp:182-215; Repartitor::computeDistrib, PartiInfo.cpp:48-103)
 *   nb_superk      running total of super-k-mers (needs minim_superks) */
#define KMX_PINFO_STRIDE (2 + 5 * 256)
typedef struct {
  uint64_t* part_counters;
  uint64_t* minim_superks;
  uint64_t* minim_kmers;
  uint64_t* minim_kxmers;
  uint64_t  nb_superk;
} kmx_superk_stats;
/* kmx_superk_partition + statistics.  out_bytes == NULL (then out_len / out_kmers are ignored): statistics only,
 * nothing is packed -- the sampling pass of the repartition, where `repart` may be any table (all zeros). */
int kmx_superk_partition_stats(kmx_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n_seqs,
                               uint32_t kmer_size, uint32_t minim_size, const uint16_t* repart,
                               uint32_t nb_parts, uint8_t** out_bytes, uint64_t* out_len, uint64_t* out_kmers,
                               kmx_superk_stats* stats);

/* The sampling pass of the sampled repartition (gatb RepartitionAlgorithm.cpp:182-215, 395-496): statistics
 * (normally minim_kxmers only) of the SHORTEST PREFIX of the reads that holds more than `budget` super-k-mers -- the
 * reference's bank iterator is cancelled by the super-k-mer that brings its count past the sample size and stops
 * before the next read.  n_used = reads in that prefix (n_seqs when the batch does not reach the budget),
 * n_superk = their super-k-mers. */
int kmx_superk_sample(kmx_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n_seqs,
                      uint32_t kmer_size, uint32_t minim_size, uint64_t budget, kmx_superk_stats* stats,
                      uint64_t* n_used, uint64_t* n_superk);

/* One sample (or one batch of its reads) from reads to counts in ONE call: kmx_superk_partition[_stats] + kmx_count_batch
 * with the super-k-mer streams never leaving HBM (what SuperKTask + CountTask / HashCountTask do through the skp files,
 * task.hpp:255-320, 367-392, 447-481).  keys / counts / n_out / out_kmers: arrays of nb_parts entries as in kmx_count_batch
 * (partition p's window index is p in hash mode); superk_bytes / superk_len: NULL, or arrays that receive the streams as
 * kmx_superk_partition returns them (--keep-tmp); superk_info: NULL or 2 * nb_parts numbers, per partition what
 * SuperKStorageWriter::SaveInfoFile reports for its skp file (io/superk_storage.hpp:205-225, 328-340: k-mers since the last
 * full 32 KB block, bytes of the blocks flushed before it); stats: NULL or as in kmx_superk_partition_stats. */
int kmx_count_reads(kmx_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n_seqs,
                    uint32_t kmer_size, uint32_t minim_size, const uint16_t* repart, uint32_t nb_parts,
                    int hash_mode, uint64_t window, uint32_t hard_min,
                    uint64_t** keys, uint32_t** counts, uint64_t* n_out, uint64_t* out_kmers,
                    uint8_t** superk_bytes, uint64_t* superk_len, uint64_t* superk_info, kmx_superk_stats* stats);

/* ---- count lists that stay in HBM between the count and the merge stage ------------------------------------------
 * The reference's CountTask writes counts/partition_<p>/<id>.kmer and its merge task reads them back
 * (task.hpp:367-392, 690-743), erasing them afterwards unless --keep-tmp (task.hpp:676-688).  On a GPU with 288 GB of
 * HBM the lists of a whole cohort fit: a kmx_store is an arena of device memory on one GPU that holds packed
 * (key, count) records -- exactly a .kmer file body, what kmx_merge_dev takes -- until the store is destroyed.
 * Thread-safe (the contexts of several host threads append to one store); limit_bytes = 0: 60 % of the device's
 * memory.  A store on another GPU than the counting context is filled with a peer copy over xGMI. */
typedef struct kmx_store kmx_store;
int      kmx_store_create(int device, uint64_t limit_bytes, kmx_store** out);
void     kmx_store_destroy(kmx_store* s);
uint64_t kmx_store_used(const kmx_store* s);
/* How a context on GPU from_device fills a store on GPU to_device: 1 = peer access between the two is enabled (asked for on first
 * use: hipDeviceCanAccessPeer + hipDeviceEnablePeerAccess) and the copy is one DMA over their xGMI link; 0 = the runtime stages it
 * through host memory; negative = KMX_E_INVAL.  `kmx pipeline --gpus G` asks for every pair when it creates its stores and prints
 * the outcome in its summary line. */
int      kmx_peer_access(int from_device, int to_device);
uint64_t kmx_store_limit(const kmx_store* s);

/* kmx_superk_stats without the host arithmetic: the device's own u32 tables of ONE call, copied (not added) into the
 * caller's buffers -- best pinned (kmx_alloc_pinned).  part_radix: [nb_parts][5][256] kx-mers of x + 1 k-mers per radix
 * (PartiInfoFile's nbk_per_radix; nb_kmers / nb_kxmers of a partition are sums over it); minim_superks / minim_kmers:
 * [4^m].  Any pointer may be NULL.  nb_superk: super-k-mers of the call. */
typedef struct {
  uint32_t* part_radix;
  uint32_t* minim_superks;
  uint32_t* minim_kmers;
  uint64_t  nb_superk;
  /* the per-minimizer records in SPARSE form instead (most of the 4^m minimizers never occur in a sample: 8 MB of tables for
   * ~10^5 entries at m = 10): minim_sparse != NULL (room for 3 * minim_sparse_cap u32; minim_superks / minim_kmers are then
   * ignored) receives minim_sparse_n triples {minimizer, super-k-mers, k-mers} in no particular order.  More minimizers
   * than minim_sparse_cap: KMX_E_INVAL (4^m entries always suffice). */
  uint32_t* minim_sparse;
  uint64_t  minim_sparse_cap;
  uint64_t  minim_sparse_n;
} kmx_superk_raw;

/* kmx_count_reads with the results left on the device: partition p's (key, count) records -- ascending, packed as a
 * .kmer body with 4-byte counts -- go to stores[p % n_stores] (the merge stage shards partitions round-robin over the
 * GPUs: the list already lies where it will be merged), lists[p] = {device pointer, records}.  stats (added to, u64)
 * or raw (copied, u32) or neither.  KMX_E_NOMEM when a store is full: the call's results are dropped, but a store is a bump arena
 * without rollback -- what the call had already placed in other stores stays consumed until kmx_store_destroy (the caller sends the
 * sample through count files: kmx_count_reads; `kmx pipeline` does).
 * With superk_bytes == NULL (here and in kmx_count_reads) no super-k-mer record stream is built at all: the k-mers are cut
 * straight from the batch's bases, the counts are the same (SuperKmerBinInfoFile's numbers still come back in superk_info:
 * they are computed from the records' sizes).  raw's buffers are filled when the call returns. */
int kmx_count_reads_dev(kmx_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n_seqs,
                        uint32_t kmer_size, uint32_t minim_size, const uint16_t* repart, uint32_t nb_parts,
                        int hash_mode, uint64_t window, uint32_t hard_min,
                        kmx_store* const* stores, uint32_t n_stores, kmx_list* lists, uint64_t* out_kmers,
                        uint8_t** superk_bytes, uint64_t* superk_len, uint64_t* superk_info,
                        kmx_superk_stats* stats, kmx_superk_raw* raw);

/* The bases of a batch on their way to the device AHEAD of the call that counts them (no reference counterpart: the reference reads
 * its super-k-mer files while it counts, task.hpp:367-392).  The copy is queued on a stream of its own and runs beside the kernels
 * of the call before it -- two count workers that each upload and then compute fall into step and leave the GPU idle for the length
 * of every upload (a third of the count stage of 1000 x 5 Mbp).  bases: page-locked (kmx_alloc_pinned), unchanged until
 * the counting call that takes them has returned.  *dev_bases is what kmx_count_reads_dev / kmx_count_reads of the SAME context then
 * get as `bases` (with the same offsets as for the host copy); kmx_reads_release gives the device block back after that call (or
 * instead of it).  At most KMX_READS_AHEAD uploads per context are alive at a time (KMX_E_INVAL beyond). */
#define KMX_READS_AHEAD 4
int  kmx_reads_upload(kmx_ctx* ctx, const char* bases, uint64_t n_bytes, const char** dev_bases);
void kmx_reads_release(kmx_ctx* ctx, const char* dev_bases);

/* kmx_count_reads_dev for SEVERAL samples in one call (no reference counterpart: SuperKTask + CountTask run per sample,
 * task.hpp:250-392; a small sample -- 1 Mbp -- is a few dozen kernels of 5-150 us each and four host round trips, which one call
 * for several samples pays once).  bases[i] / offsets[i] / n_seqs[i]: sample i's reads as in kmx_count_reads.  Results are
 * sample-major: lists[i * nb_parts + p], out_kmers[i * nb_parts + p], superk_info[2 * (i * nb_parts + p)], raw[i] (every sample's
 * statistics in the same form -- all sparse or all dense -- or raw == NULL).  Partition p of every sample goes to
 * stores[p % n_stores]; a hash window's id is p.  n_samples * nb_parts <= 65535; not while the abundance histogram is on
 * (kmx_hist_reset: it is per call).  The same lists, numbers and tables as n_samples calls of kmx_count_reads_dev. */
int kmx_count_reads_dev_multi(kmx_ctx* ctx, uint32_t n_samples, const char* const* bases, const uint64_t* const* offsets,
                              const uint64_t* n_seqs, uint32_t kmer_size, uint32_t minim_size, const uint16_t* repart_table,
                              uint32_t nb_parts, int hash_mode, uint64_t window, uint32_t hard_min,
                              kmx_store* const* stores, uint32_t n_stores, kmx_list* lists, uint64_t* out_kmers,
                              uint64_t* superk_info, kmx_superk_raw* raw);
/* device memory (a list of a store, a result body) into host memory; blocks until it is there */
int kmx_copy_to_host(kmx_ctx* ctx, void* host_dst, const void* dev_src, uint64_t bytes);
/* ... without the wait: the copy is queued behind the context's earlier ones and runs back to back with them -- a writer that
 * brings a matrix body over in pieces keeps two in flight, so the link does not idle while it hands a piece on (0.6 ms per piece
 * otherwise: 0.4 s of the 2.0 s that 92 GB of matrices take).  host_dst: page-locked (kmx_alloc_pinned).
 * kmx_copy_wait(ctx, ticket) blocks until that copy (and every one queued before it) is in host memory and gives the ticket back;
 * at most KMX_COPIES_AHEAD tickets are out at a time (KMX_E_INVAL beyond). */
#define KMX_COPIES_AHEAD 8
int kmx_copy_to_host_async(kmx_ctx* ctx, void* host_dst, const void* dev_src, uint64_t bytes, uint32_t* ticket);
int kmx_copy_wait(kmx_ctx* ctx, uint32_t ticket);

/* The abundance histogram of a sample (`--hist`): the reference's KHist (histogram.hpp:35-68) is fed EVERY distinct k-mer /
 * hash of the sample with its count, before the hard-min filter (count_processor.hpp:61, 135), one clone per partition,
 * summed (histogram.hpp:113-136).  kmx_hist_reset zeroes the context's device histogram and turns accumulation on: every
 * kmx_count_kmer / _hash / _batch / _reads call after it adds its distinct keys (on the device, where the run lengths are).
 * kmx_hist_read waits for them and fills uniq_bins / total_bins (upper - lower + 1 entries each: keys with count c, and c times
 * that), oob[4] = {keys below lower, keys above upper, the sums of their counts: lower, upper} and sums[2] = {distinct keys,
 * sum of counts} -- the fields of HistFileHeader + the two vectors of a .hist file (io/hist_file.hpp:30-116).
 * lower <= upper <= 255 (the reference always builds KHist(id, k, 1, 255): task_scheduler.hpp:103).  kmx_hist_off stops it. */
int kmx_hist_reset(kmx_ctx* ctx);
int kmx_hist_read(kmx_ctx* ctx, uint32_t lower, uint32_t upper, uint64_t* uniq_bins, uint64_t* total_bins, uint64_t* oob, uint64_t* sums);
int kmx_hist_off(kmx_ctx* ctx);

void kmx_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
