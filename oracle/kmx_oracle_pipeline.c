/*
 * kmx_oracle_pipeline.c -- the CPU oracle's split + count + merge over whole samples on host threads, timed.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (see kmx_oracle.h): this is bench.py's `pipeline.cpu_baseline` leg -- the
 * restatement of the reference's CPU path (SuperKTask + CountTask per sample, task.hpp:255-392; one KmerMergeTask per
 * partition, task.hpp:690-743; a pool of worker threads, task_pool.hpp:36-120) run the way the reference runs it: one
 * task per sample handed to a pool of T threads, then one merge task per partition on the same pool.  Plain C and
 * pthreads, no interpreter in the loop; T is chosen by the caller (at most one thread per physical core).
 *
 * usage: kmx_oracle_pipeline K M PARTS HARD_MIN REC_MIN THREADS file.fa [file.fa ...]
 * prints one JSON object: wall clocks, the per-core rate of the split + count leg (CPU seconds of the worker threads) and
 * the rate of ONE thread working alone on the first sample (the two must agree within 2x: asserted by the caller).
 */
#define _GNU_SOURCE
#include "kmx_oracle.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(clockid_t c) { struct timespec t; clock_gettime(c, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

typedef struct { uint64_t* keys; uint32_t* counts; uint64_t n; } plist;

static int g_k, g_m, g_parts, g_hard_min, g_rec_min, g_kw;
static uint32_t* g_lut;
static uint16_t* g_rep;
static char** g_files;
static int g_nfiles;
static plist* g_lists;          /* [sample][partition] */
static uint64_t* g_bases;       /* per sample */
static uint64_t* g_kmers;

/* one sample: every read of the FASTA / FASTQ-free file (header lines start with '>') through the split, then every
 * partition's record stream through the counter */
static void do_sample(int s)
{
  FILE* f = fopen(g_files[s], "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", g_files[s]); exit(2); }
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  char* buf = (char*)malloc((size_t)sz + 1);
  if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fprintf(stderr, "short read %s\n", g_files[s]); exit(2); }
  fclose(f); buf[sz] = '\n';
  orc_buf* out = (orc_buf*)calloc((size_t)g_parts, sizeof(orc_buf));
  uint64_t bases = 0, kmers = 0;
  /* a record = a header line and the sequence lines up to the next header, joined (Bank iteration, gatb bank/impl/BankFasta.cpp) */
  char* seq = (char*)malloc((size_t)sz + 1);
  long i = 0;
  while (i < sz) {
    if (buf[i] == '>') { while (i < sz && buf[i] != '\n') i++; i++; }
    size_t n = 0;
    while (i < sz && buf[i] != '>') { long e = i; while (buf[e] != '\n') e++; memcpy(seq + n, buf + i, (size_t)(e - i)); n += (size_t)(e - i); i = e + 1; }
    if (n) { orc_superk_partition(seq, n, g_k, g_m, g_lut, g_rep, (uint32_t)g_parts, out, NULL); bases += n; }
  }
  free(seq); free(buf);
  for (int p = 0; p < g_parts; p++) {
    plist* L = &g_lists[(size_t)s * g_parts + p];
    kmers += out[p].nb_kmers;
    orc_count_kmer(out[p].data, out[p].len, g_k, (uint32_t)g_hard_min, &L->keys, &L->counts, &L->n);
    orc_buf_free(&out[p]);
  }
  free(out);
  g_bases[s] = bases; g_kmers[s] = kmers;
}

static int g_next;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static double* g_cpu;           /* CPU seconds per worker thread */
static uint64_t g_rows;

static void* count_worker(void* arg)
{
  const int w = (int)(intptr_t)arg;
  const double c0 = now_s(CLOCK_THREAD_CPUTIME_ID);
  for (;;) {
    pthread_mutex_lock(&g_mu); const int s = g_next++; pthread_mutex_unlock(&g_mu);
    if (s >= g_nfiles) break;
    do_sample(s);
  }
  g_cpu[w] = now_s(CLOCK_THREAD_CPUTIME_ID) - c0;
  return NULL;
}

static void* merge_worker(void* arg)
{
  const int w = (int)(intptr_t)arg;
  const double c0 = now_s(CLOCK_THREAD_CPUTIME_ID);
  orc_list* ls = (orc_list*)malloc((size_t)g_nfiles * sizeof(orc_list));
  uint32_t* soft = (uint32_t*)malloc((size_t)g_nfiles * 4);
  uint64_t* stats = (uint64_t*)malloc((size_t)g_nfiles * 6 * 8);
  for (int i = 0; i < g_nfiles; i++) soft[i] = 1;
  uint64_t rows_mine = 0;
  for (;;) {
    pthread_mutex_lock(&g_mu); const int p = g_next++; pthread_mutex_unlock(&g_mu);
    if (p >= g_parts) break;
    for (int s = 0; s < g_nfiles; s++) { const plist* L = &g_lists[(size_t)s * g_parts + p]; ls[s].keys = L->keys; ls[s].counts = L->counts; ls[s].n = L->n; }
    uint8_t* body = NULL; uint64_t blen = 0, rows = 0;
    orc_merge_matrix(ls, (uint32_t)g_nfiles, g_kw, soft, (uint32_t)g_rec_min, 0, ORC_MODE_COUNT, 0, 0, 2, &body, &blen, &rows, stats);
    orc_free(body);
    rows_mine += rows;
  }
  pthread_mutex_lock(&g_mu); g_rows += rows_mine; pthread_mutex_unlock(&g_mu);
  free(ls); free(soft); free(stats);
  g_cpu[w] = now_s(CLOCK_THREAD_CPUTIME_ID) - c0;
  return NULL;
}

static double run_pool(void* (*fn)(void*), int T, double* cpu_sum)
{
  pthread_t* th = (pthread_t*)malloc((size_t)T * sizeof(pthread_t));
  g_next = 0;
  const double t0 = now_s(CLOCK_MONOTONIC);
  for (int w = 0; w < T; w++) pthread_create(&th[w], NULL, fn, (void*)(intptr_t)w);
  for (int w = 0; w < T; w++) pthread_join(th[w], NULL);
  const double wall = now_s(CLOCK_MONOTONIC) - t0;
  *cpu_sum = 0; for (int w = 0; w < T; w++) *cpu_sum += g_cpu[w];
  free(th);
  return wall;
}

int main(int argc, char** argv)
{
  if (argc < 8) { fprintf(stderr, "usage: %s K M PARTS HARD_MIN REC_MIN THREADS file.fa ...\n", argv[0]); return 2; }
  g_k = atoi(argv[1]); g_m = atoi(argv[2]); g_parts = atoi(argv[3]); g_hard_min = atoi(argv[4]); g_rec_min = atoi(argv[5]);
  int T = atoi(argv[6]);
  g_files = argv + 7; g_nfiles = argc - 7; g_kw = (g_k + 31) / 32;
  if (T > g_nfiles) T = g_nfiles;
  if (T < 1) T = 1;
  g_lut = (uint32_t*)malloc(((size_t)1 << (2 * g_m)) * 4); orc_minimizer_lut(g_m, g_lut);
  g_rep = (uint16_t*)malloc(((size_t)1 << (2 * g_m)) * 2); orc_repart_static(g_m, (uint32_t)g_parts, g_rep);
  g_lists = (plist*)calloc((size_t)g_nfiles * g_parts, sizeof(plist));
  g_bases = (uint64_t*)calloc((size_t)g_nfiles, 8); g_kmers = (uint64_t*)calloc((size_t)g_nfiles, 8);
  g_cpu = (double*)calloc((size_t)(T > g_parts ? T : g_parts) + 1, sizeof(double));

  /* one thread alone on the first sample: what a core does when nothing else runs (the result is thrown away) */
  const double s0 = now_s(CLOCK_MONOTONIC);
  do_sample(0);
  const double single_s = now_s(CLOCK_MONOTONIC) - s0;
  const uint64_t single_bases = g_bases[0];
  for (int p = 0; p < g_parts; p++) { orc_free(g_lists[p].keys); orc_free(g_lists[p].counts); g_lists[p].keys = NULL; g_lists[p].counts = NULL; g_lists[p].n = 0; }

  double cpu_count = 0, cpu_merge = 0;
  const double wall_count = run_pool(count_worker, T, &cpu_count);
  uint64_t bases = 0, kmers = 0, recs = 0;
  for (int s = 0; s < g_nfiles; s++) { bases += g_bases[s]; kmers += g_kmers[s]; }
  for (size_t j = 0; j < (size_t)g_nfiles * g_parts; j++) recs += g_lists[j].n;
  const int Tm = T < g_parts ? T : g_parts;
  const double wall_merge = run_pool(merge_worker, Tm, &cpu_merge);
  printf("{\"samples\": %d, \"threads\": %d, \"merge_threads\": %d, \"bases\": %llu, \"kmers\": %llu, \"merge_records\": %llu, \"rows\": %llu, "
         "\"split_count_wall_s\": %.4f, \"split_count_cpu_s\": %.4f, \"merge_wall_s\": %.4f, \"merge_cpu_s\": %.4f, "
         "\"split_count_Mbases_per_s_per_core\": %.3f, \"single_core_Mbases_per_s\": %.3f, \"merge_Mrecords_per_s_per_core\": %.3f}\n",
         g_nfiles, T, Tm, (unsigned long long)bases, (unsigned long long)kmers, (unsigned long long)recs, (unsigned long long)g_rows,
         wall_count, cpu_count, wall_merge, cpu_merge,
         cpu_count > 0 ? (double)bases / cpu_count / 1e6 : 0.0, single_s > 0 ? (double)single_bases / single_s / 1e6 : 0.0,
         cpu_merge > 0 ? (double)recs / cpu_merge / 1e6 : 0.0);
  return 0;
}
