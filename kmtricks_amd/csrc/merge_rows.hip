// merge_rows.hip -- N-way merge of sorted per-sample count lists into count / presence-absence
// matrix rows on gfx950.  Replaces km::KmerMerger::next + write_as_bin/write_as_pa and
// km::HashMerger::next + write_as_bin/write_as_pa (reference include/kmtricks/merge.hpp:183-286,
// 441-558): for every distinct key, counts[i] = c_i if c_i >= soft_min[i]; non-solid entries are
// rescued when share_min > 0 and recurrence >= share_min; the row is kept iff recurrence >= rec_min.
//
// Decomposition (see DESIGN.md "merge kernel"):
//   * the key space of a partition is cut into c ranges at quantiles of one pivot list
//     (k_range_bounds: one lower_bound per (range, list));
//   * a workgroup owns one range and walks it tile by tile.  A tile gives every list a circular
//     window of w = 2^wl record slots held in registers (w adjacent lanes, 12/20-byte records);
//     the tile's key bound b is the smallest "last record of a window that has more records behind
//     it", so every record <= b of every list is inside its window: tiles are disjoint, ascending
//     key intervals that always fit the 4096 register slots, whatever the skew.  Consumed slots are
//     refilled by a prefetch issued right after b is known, so every record is loaded exactly once
//     and the load latency hides behind the rest of the tile;
//   * inside a tile the distinct keys are found with an LDS hash set (owner slot + recurrence
//     packed in one u32, claimed with ds_cmpst; reads first, so dense keys cost no atomics), the
//     owners publish the kept keys, those are ranked, and rows are assembled as a byte-exact file
//     image in LDS, then streamed out with coalesced stores;
//   * rows go to chunks of the task's row arena claimed with one global atomic per chunk; the
//     (range, seq) chunk directory restores ascending key order when the body is copied out.
//     Input is read once, output written once, no inter-workgroup dependency.
#include "kmx_dev.hpp"
#include <algorithm>
#include <cstdio>
#include <cstring>

// Round 6: a second build of this file (merge_rows_small.hip: KMX_ROWS_SMALL, 512 threads, 2048 record slots, two workgroups a CU) for
// cohorts of up to 256 lists -- there a list's window is 8-16 slots whatever the tile, and what the kernel waits for is its barriers
// and the next records: two smaller workgroups a CU hide one another's (64 lists 0.18 -> 0.24 of the roofline, 128: 0.23 -> 0.27,
// 256: 0.25 -> 0.27; profiles/r06_mid_cohorts.txt).  The build's names differ so that both live in one library.
#ifdef KMX_ROWS_SMALL
#define k_merge_rows k_merge_rows_s
#define cap_of cap_of_s
#define ts_of ts_of_s
#define rows_emit_bytes rows_emit_bytes_s
#define rows_fixed_bytes rows_fixed_bytes_s
#define rows_big rows_big_s
#define rows_lds_bytes rows_s_lds_bytes
#define rows_cap rows_s_cap
#define rows_wgs_per_cu rows_s_wgs_per_cu
#define rows_image_bytes rows_s_image_bytes
#define launch_merge_rows launch_merge_rows_s
#define kmx_phase_prof kmx_phase_prof_s
#define rows_phase_prof_dump rows_s_phase_prof_dump
#endif

namespace kmx {

#ifndef KMX_ROWS_TPB
#define KMX_ROWS_TPB 1024
#endif
#ifndef KMX_ROWS_CAP
#define KMX_ROWS_CAP 4096
#endif
constexpr int TPB = KMX_ROWS_TPB;  // 8 or 16 waves
constexpr int CAP = KMX_ROWS_CAP;  // record slots per tile
// per tile and key width (cap_of below): M = CAP / TPB record slots per thread, TS = 2 CAP hash set entries (load factor <= 0.5),
// KLBYTES = CAP * 2 + 4096 for the kept keys: u16 table slots [CAP] + fast-path key copies (4 KiB)
constexpr int WGS_PER_CU = (TPB <= 512 && (CAP * 8 + 2 * CAP * 4 + CAP * 2 + 4096 + 384) * 2 <= 160 * 1024) ? 2 : 1;   // KW = 1: LDS and 128-VGPR budget
constexpr int NWAVE = TPB / 64;
// keys of three and four words (k = 65 ... 96 and 97 ... 127: ceil(k / 32) words of a Kmer<96> / Kmer<128>; include/kmtricks/kmer.hpp:164-630, :215 of the reference): half the record
// slots per tile -- the staged keys of 4096 slots alone would be 128 KB of the 160 KB, and a thread's keys leave the registers
// (BIG, round 5: for tasks of more lists than CAP / 2 -- three-word keys, k = 65 ... 96, with all CAP slots, four-word keys with 3 CAP / 4,
// both under 140 KB of LDS; a thread's four / three records spill some 80 registers to scratch: the build for cohorts beyond 2048
// samples, not for speed)
__host__ __device__ constexpr int cap_of(int kw, bool big = false) { return kw <= 2 || (big && kw == 3) ? CAP : (big && kw == 4) ? 3 * CAP / 4 : CAP / 2; }
__host__ __device__ constexpr int ts_of(int cap) { int t = 1; while (t < 2 * cap) t *= 2; return t; }      // hash set entries: a power of two, load factor <= 0.5

// LDS image of the row batch being assembled: aliases the staged keys
__host__ __device__ inline int rows_emit_bytes(int kw, bool big = false) { return cap_of(kw, big) * kw * 8; }
__host__ __device__ inline int rows_fixed_bytes(int kw, bool big = false) { return cap_of(kw, big) * kw * 8 + ts_of(cap_of(kw, big)) * 4 + cap_of(kw, big) * 2 + 4096; }

#ifndef KMX_ROWS_SMALL
// ---- range bounds ------------------------------------------------------------------------------
// bounds[j*N + i] = first record of list i whose key >= Q_j, Q_j = pivot[j * len_pivot / c].
template <int KW>
__global__ void k_range_bounds(const TaskDev* __restrict__ tasks, u32 max_c)
{
  const TaskDev& T = tasks[blockIdx.z];
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 j = blockIdx.y;
  if (i >= T.N || j > T.c) return;
  const u32 n = T.len[i];
  u32 res;
  if (j == 0) res = 0;
  else if (j == T.c) res = n;
  else {
    constexpr int RB = KW * 8 + 4;
    const u32 np = T.len[T.pivot];
    const u32 pos = (u32)(((u64)j * np) / T.c);
    const Key<KW> q = load_key<KW>(T.recs[T.pivot] + (u64)pos * RB);
    const u8* base = T.recs[i];
    u32 lo = 0, hi = n;
    if (n) {
      // lists of a cohort resemble their pivot: gallop outwards from the proportional position, then
      // bisect the bracket (a handful of dependent loads instead of log2(n))
      const u32 g = (u32)min((u64)n - 1, ((u64)pos * n) / np);
      if (key_less<KW>(load_key<KW>(base + (u64)g * RB), q)) {
        lo = g + 1;
        u32 step = 1, pr = g + 1;
        while (pr < n && key_less<KW>(load_key<KW>(base + (u64)pr * RB), q)) { lo = pr + 1; step <<= 1; pr = g + step; }
        hi = min(pr, n);
      } else {
        hi = g;
        u32 step = 1;
        while (step <= g && !key_less<KW>(load_key<KW>(base + (u64)(g - step) * RB), q)) { hi = g - step; step <<= 1; }
        lo = step <= g ? g - step + 1 : 0;
      }
    }
    while (lo < hi) {
      u32 mid = lo + ((hi - lo) >> 1);
      Key<KW> k = load_key<KW>(base + (u64)mid * RB);
      if (key_less<KW>(k, q)) lo = mid + 1; else hi = mid;
    }
    res = lo;
  }
  T.bounds[(u64)j * T.N + i] = res;
}
#endif

#ifdef KMX_PHASE_PROF
__device__ u64 kmx_phase_prof[16];
#endif

// ---- the merge kernel ----------------------------------------------------------------------------
template <int KW, int TS> __device__ __forceinline__ u32 key_hash(const Key<KW>& k)
{ // cheap 32-bit mix (a handful of VALU ops); quality only matters for the probe length
  u32 x = (u32)k.w[0] ^ ((u32)(k.w[0] >> 32) * 0x9E3779B1u);
  if (KW >= 2) x ^= ((u32)k.w[KW - 1] * 0x85EBCA77u) ^ ((u32)(k.w[KW - 1] >> 32) * 0xC2B2AE3Du);
  if (KW >= 3) x ^= ((u32)k.w[1] * 0x27D4EB2Fu) ^ ((u32)(k.w[1] >> 32) * 0x165667B1u);
  if (KW >= 4) x ^= ((u32)k.w[2] * 0x9E3779B1u) ^ ((u32)(k.w[2] >> 32) * 0x85EBCA77u);
  x *= 0x85EBCA6Bu; x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 13;
  return x & (TS - 1);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt on gfx950
// (loads and stores share the counter), which would wait for the record prefetch at every barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// linear probing past a hash collision (rare): returns table slot (low 32) | previous entry (high 32)
// of `k`, claiming an empty slot for record slot `s` if the key is new (previous entry 0) -- out of line
template <int KW, int TS>
__device__ __noinline__ u64 probe_slow(u32* tab, const Key<KW>* keysL, Key<KW> k, u32 h, u32 s)
{
  for (;;) {
    h = (h + 1) & (TS - 1);
    u32 o = tab[h];
    if (o == 0) {
      o = atomicCAS(&tab[h], 0u, s + 1);
      if (o == 0) return (u64)h;
    }
    if (key_eq<KW>(keysL[(o & 0xFFFFu) - 1], k)) return (u64)h | ((u64)o << 32);
  }
}

// One workgroup per CU: the per-slot state (record being merged, record in flight, statistics)
// lives in registers.  The body is written for a low instruction count per record: the kernel is
// issue-bound long before it is LDS- or HBM-bound.
template <int KW, int MODE, bool BIG>
__global__ __launch_bounds__(TPB, (KW == 1 ? WGS_PER_CU : 1) * TPB / 256)
void k_merge_rows(const TaskDev* __restrict__ tasks, const uint2* __restrict__ items, u32 n_items, u32* ticket)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CAP = cap_of(KW, BIG), M = CAP / TPB, TS = ts_of(CAP), KLBYTES = CAP * 2 + 4096;      // (the file's constants, for this key width)
  constexpr int WMIN = KW <= 2 ? 256 : NWAVE * KW * 8;                                          // bytes of the waves' candidate keys
  constexpr int RB4 = (KW * 8 + 4) / 4;
  constexpr int KEYS_BYTES = CAP * KW * 8;
  constexpr int DKMAX = 4096 / (KW * 8);

  Key<KW>* keysL = reinterpret_cast<Key<KW>*>(smem);               // staged keys of the tile ...
  unsigned char* const img = smem;                                  // ... later the row image (aliased)
  u32* tab = reinterpret_cast<u32*>(smem + KEYS_BYTES);             // hash set, all zero between tiles
  u16* dslot = reinterpret_cast<u16*>(smem + KEYS_BYTES + TS * 4);  // table slots of the kept keys (8 KiB)
  Key<KW>* dkeys = reinterpret_cast<Key<KW>*>(smem + KEYS_BYTES + TS * 4 + CAP * 2);   // their keys, fast path (4 KiB)
  unsigned char* misc = smem + KEYS_BYTES + TS * 4 + KLBYTES;
  Key<KW>* wmin = reinterpret_cast<Key<KW>*>(misc);                 // NWAVE keys (<= 256 B)
  u32* wany = reinterpret_cast<u32*>(misc + WMIN);                  // NWAVE flags (64 B)
  u64* bc64 = reinterpret_cast<u64*>(misc + WMIN + 64);             // broadcast: row offset
  u32* bc32 = reinterpret_cast<u32*>(misc + WMIN + 80);             // [0] item  [1] can-write  [2] kept counter

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef KMX_PHASE_PROF
  long long pt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; long long pc = clock64();
#define PH(i) do { const long long n_ = clock64(); pt[i] += n_ - pc; pc = n_; } while (0)
#else
#define PH(i) do {} while (0)
#endif
  {
    uint4* t4 = reinterpret_cast<uint4*>(tab);
    for (int t = tid; t < TS / 4; t += TPB) t4[t] = make_uint4(0, 0, 0, 0);
    if (tid == 0) bc32[2] = 0;
  }

  for (;;) {
    // ---- next work item (dynamic, ascending ids) ----
    if (tid == 0) bc32[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    const u32 item = (u32)__builtin_amdgcn_readfirstlane((int)bc32[0]);   // scalar: the task descriptor is read with scalar loads
    __syncthreads();
    if (item >= n_items) {
#ifdef KMX_PHASE_PROF
      if (tid == 0) for (int i = 0; i < 9; i++) atomicAdd(&kmx_phase_prof[i], (u64)pt[i]);
#endif
      return;
    }
    const TaskDev& T = tasks[items[item].x];
    const u32 range = items[item].y;
    const u32 N = T.N, wl = T.wl, w = 1u << wl;      // w <= 64: a list's window never leaves its wave
    const u32 rec_min = T.rec_min, share_min = T.share_min, row_bytes = T.row_bytes;
    const u32 sat = max(rec_min, share_min);          // recurrence only matters up to this value
    const u32 chunk_rows = max(64u, (u32)KMX_CHUNK_BYTES / row_bytes);
    const u32 g0 = lane & ~(w - 1);                   // first lane of my list's lane group
    const u32 nxt = (lane + 1) & (w - 1);             // group-relative lane holding the next window position
    const u64 wmask = (w == 64) ? ~0ULL : ((1ULL << w) - 1);

    // Slot s = tid + m*TPB serves list s >> wl.  A list's window is circular: lane residue
    // rr = s & (w-1) always holds the record whose index is == rr (mod w) inside [cur, cur + w),
    // so a consumed record is replaced in place by record idx + w and every record is loaded once.
    // lastm bit m: this slot holds the LAST record of its window (its key bounds the tile).
    gu32* ptr[M]; u32 endv[M], idx[M], smin[M], nso[M]; u64 two[M];
    Key<KW> key[M]; u32 cnt[M];
    u32 lastm = 0;
#pragma unroll
    for (int m = 0; m < M; m++) {
      const u32 s = tid + m * TPB, li = s >> wl, rr = s & (w - 1);
      nso[m] = 0; two[m] = 0; idx[m] = 0; endv[m] = 0; ptr[m] = (gu32*)nullptr; smin[m] = 0;
      key[m] = key_inf<KW>(); cnt[m] = 0;
      if (li < N) {
        ptr[m] = (gu32*)(uintptr_t)T.recs[li];
        smin[m] = T.soft_min[li];
        endv[m] = T.bounds[(u64)(range + 1) * N + li];
        const u32 c0 = T.bounds[(u64)range * N + li];
        idx[m] = c0 + ((rr - c0) & (w - 1));
        if (idx[m] == c0 + w - 1) lastm |= 1u << m;
        if (idx[m] < endv[m]) {
          gu32* p = ptr[m] + (u64)idx[m] * RB4;
#pragma unroll
          for (int q = 0; q < KW; q++) key[m].w[q] = (u64)p[2 * q] | ((u64)p[2 * q + 1] << 32);
          cnt[m] = p[2 * KW];
        }
      }
    }
    // row-space allocator state (thread 0): rows are claimed in chunks, one directory entry per chunk
    u64 ch_base = 0; u32 ch_used = 0, ch_cap = 0, ch_seq = 0, ch_ok = 1;
    __syncthreads();

    u32 seq = 0;
    PH(0);
    for (;;) {
      // ---- 1. tile bound b = smallest "last record of a window that has more records behind it" ----
      Key<KW> cand = key_inf<KW>();
      u32 anyv = 0;
#pragma unroll
      for (int m = 0; m < M; m++) {
        anyv |= (idx[m] < endv[m]) ? 1u : 0u;
        if (((lastm >> m) & 1u) && idx[m] + 1 < endv[m]) cand = key_min<KW>(cand, key[m]);
      }
      cand = wave_min_key<KW>(cand);
      const u64 vbal = __ballot(anyv != 0);
      if (lane == 0) { wmin[wave] = cand; wany[wave] = vbal != 0; }
      lds_barrier();
      PH(1);
      Key<KW> b = wmin[0];
      u32 any = wany[0];
#pragma unroll
      for (int v = 1; v < NWAVE; v++) { b = key_min<KW>(b, wmin[v]); any |= wany[v]; }
      if (!any) break;

      // ---- 2. consume keys <= b, stage them, prefetch the replacements ----
      Key<KW> nkey[M]; u32 ncnt[M];
      u32 consm = 0;
#pragma unroll
      for (int m = 0; m < M; m++) {
        const bool c = idx[m] < endv[m] && key_le<KW>(key[m], b);
        const u64 g = (__ballot(c) >> g0) & wmask;          // consumed lanes of my list
        if (g && g != wmask) {                              // the last consumed slot becomes the window's last
          if (c && !((g >> nxt) & 1ULL)) lastm |= 1u << m; else lastm &= ~(1u << m);
        }                                                   // (whole window consumed: the last stays the last)
        if (c) {
          consm |= 1u << m;
          keysL[tid + m * TPB] = key[m];
          const u32 ni = idx[m] + w;
          if (ni < endv[m]) {
            gu32* p = ptr[m] + (u64)ni * RB4;
#pragma unroll
            for (int q = 0; q < KW; q++) nkey[m].w[q] = (u64)p[2 * q] | ((u64)p[2 * q + 1] << 32);
            ncnt[m] = p[2 * KW];
          }
        }
      }
      lds_barrier();
      PH(2);

      // ---- 3. hash-set insert: entry = owner slot + 1 (low 16) | recurrence (high 16) ----
      // Dense keys are inserted by hundreds of lists at once, so every access reads first: only the
      // first arrivals issue the ds_cmpst claim and the recurrence counter stops being incremented
      // at max(rec_min, share_min) -- same-address LDS atomics serialise, same-address reads broadcast.
      u32 hs[M]; u32 ownm = 0, solidm = 0;
      {
        u32 old[M];
#pragma unroll
        for (int m = 0; m < M; m++) { hs[m] = key_hash<KW, TS>(key[m]); old[m] = 1; }
#pragma unroll
        for (int m = 0; m < M; m++) if ((consm >> m) & 1u) old[m] = tab[hs[m]];
#pragma unroll
        for (int m = 0; m < M; m++) {
          if ((consm >> m) & 1u) {
            bool own = false;
            if (old[m] == 0) { old[m] = atomicCAS(&tab[hs[m]], 0u, (u32)(tid + m * TPB) + 1); own = old[m] == 0; }
            if (!own && !key_eq<KW>(keysL[(old[m] & 0xFFFFu) - 1], key[m])) {
              const u64 pr = probe_slow<KW, TS>(tab, keysL, key[m], hs[m], (u32)(tid + m * TPB));
              hs[m] = (u32)pr; old[m] = (u32)(pr >> 32); own = old[m] == 0;
            }
            if (own) ownm |= 1u << m;
            if (cnt[m] >= smin[m]) {
              solidm |= 1u << m;
              if ((old[m] >> 16) < sat) atomicAdd(&tab[hs[m]], 1u << 16);
            }
          }
        }
      }
      lds_barrier();
      PH(3);

      // ---- 4. owners publish the kept keys (recurrence >= rec_min) ----
      if (ownm) {
#pragma unroll
        for (int m = 0; m < M; m++) {
          if ((ownm >> m) & 1u) {
            const u32 e = tab[hs[m]];
            if ((e >> 16) >= rec_min) {
              const u32 pos = atomicAdd(&bc32[2], 1u);
              dslot[pos] = (u16)hs[m];
              if (pos < (u32)DKMAX) dkeys[pos] = key[m];
            } else tab[hs[m]] = (e & 0xFFFF0000u) | 0xFFFFu;
          }
        }
      }
      lds_barrier();
      const u32 dk = (u32)__builtin_amdgcn_readfirstlane((int)bc32[2]);
      if (tid == 0) {   // row space for this tile (ascending inside a chunk; one directory entry per chunk)
        u64 off = 0;
        if (dk) {
          if (ch_used + dk > ch_cap) {
            if (ch_used) {
              const u64 sidx = atomicAdd(&T.ctrl[1], 1ULL);
              if (sidx < T.seg_cap) { Seg sg; sg.range = range; sg.seq = ch_seq; sg.row_off = ch_base; sg.nrows = ch_used; sg.pad = 0; T.segs[sidx] = sg; }
              else atomicOr(&T.ctrl[2], (u64)ERR_SEGS_OVERFLOW);
              atomicAdd(&T.ctrl[3], (u64)ch_used);
            }
            ch_cap = max(chunk_rows, dk);
            ch_base = atomicAdd(&T.ctrl[0], (u64)ch_cap);
            ch_used = 0; ch_seq = seq;
            ch_ok = (ch_base + ch_cap <= T.out_cap_rows) ? 1u : 0u;
            if (!ch_ok) atomicOr(&T.ctrl[2], (u64)ERR_ROWS_OVERFLOW);
          }
          off = ch_base + ch_used; ch_used += dk;
        }
        bc64[0] = off; bc32[1] = ch_ok;
      }
      // ---- 5. rank the kept keys ----
      if (dk <= (u32)DKMAX) {
        // counting rank: kept-rank = number of kept keys smaller than mine (broadcast LDS reads)
        for (u32 p = tid; p < dk; p += TPB) {
          const Key<KW> mine = dkeys[p];
          u32 r = 0;
          for (u32 q = 0; q < dk; q++) r += key_less<KW>(dkeys[q], mine) ? 1u : 0u;
          const u32 t = dslot[p];
          tab[t] = (tab[t] & 0xFFFF0000u) | r;
        }
      } else {
        // slow path (more than DKMAX kept keys in one tile): LDS bitonic sort of table slots by key
        u32 p2 = 1; while (p2 < dk) p2 <<= 1;
        for (u32 p = dk + tid; p < p2; p += TPB) dslot[p] = 0xFFFFu;   // +inf padding
        lds_barrier();
        for (u32 k2 = 2; k2 <= p2; k2 <<= 1) {
          for (u32 j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (u32 p = tid; p < p2; p += TPB) {
              const u32 q = p ^ j2;
              if (q > p) {
                const u32 a = dslot[p], bb = dslot[q];
                bool a_gt_b;
                if (a == 0xFFFFu) a_gt_b = (bb != 0xFFFFu);
                else if (bb == 0xFFFFu) a_gt_b = false;
                else a_gt_b = key_less<KW>(keysL[(tab[bb] & 0xFFFFu) - 1], keysL[(tab[a] & 0xFFFFu) - 1]);
                const bool up = (p & k2) == 0;
                if (a_gt_b == up) { dslot[p] = (u16)bb; dslot[q] = (u16)a; }
              }
            }
            lds_barrier();
          }
        }
        for (u32 p = tid; p < dk; p += TPB) { const u32 t = dslot[p]; tab[t] = (tab[t] & 0xFFFF0000u) | p; }
      }
      const u32 rows_per = (u32)KEYS_BYTES / row_bytes;
      {   // the staged keys are dead: zero the first row batch of the image
        const u32 zb = min(dk, rows_per) * row_bytes;
        uint4* z = reinterpret_cast<uint4*>(img);
        for (u32 t = tid; t < (zb + 15) / 16; t += TPB) z[t] = make_uint4(0, 0, 0, 0);
      }
      lds_barrier();
      PH(4);
      const u64 row_off = bc64[0];
      const bool can_write = bc32[1] != 0;
      if (tid == 0) bc32[2] = 0;

      // ---- 6. per-record decision (merge.hpp:199-247), statistics; rows as a file-body image ----
      u32 kr[M];
#pragma unroll
      for (int m = 0; m < M; m++) kr[m] = ((consm >> m) & 1u) ? tab[hs[m]] : 0xFFFFu;
#pragma unroll
      for (int m = 0; m < M; m++) {
        if ((consm >> m) & 1u) {
          const u32 rec = kr[m] >> 16;
          kr[m] &= 0xFFFFu;
          if ((solidm >> m) & 1u) two[m] += cnt[m];
          else {
            nso[m] += 1;
            if (share_min && rec >= share_min) {
              const u32 li = (tid + m * TPB) >> wl;
              atomicAdd(&T.stats[1 * (u64)N + li], 1ULL);
              atomicAdd(&T.stats[5 * (u64)N + li], (u64)cnt[m]);
            } else cnt[m] = 0;                       // neither solid nor rescued: contributes nothing
          }
        }
      }
      PH(5);
      if (dk && can_write) {
        u8* const dst0 = T.out + row_off * row_bytes;
        for (u32 b0 = 0; b0 < dk; b0 += rows_per) {
          const u32 nb = min(rows_per, dk - b0);
          const u32 bytes = nb * row_bytes;
          if (b0) {
            lds_barrier();
            uint4* z = reinterpret_cast<uint4*>(img);
            for (u32 t = tid; t < (bytes + 15) / 16; t += TPB) z[t] = make_uint4(0, 0, 0, 0);
            lds_barrier();
          }
#pragma unroll
          for (int m = 0; m < M; m++) {
            const u32 r = kr[m] - b0;   // wraps for 0xFFFF / other batches
            if (kr[m] != 0xFFFFu && r < nb) {
              u8* row = img + r * row_bytes;
              if ((ownm >> m) & 1u) {
                if (MODE == 0) {
                  u32* rw = reinterpret_cast<u32*>(row);
#pragma unroll
                  for (int q = 0; q < KW; q++) { rw[2 * q] = (u32)key[m].w[q]; rw[2 * q + 1] = (u32)(key[m].w[q] >> 32); }
                } else {
#pragma unroll
                  for (int q = 0; q < KW * 8; q++) row[q] = (u8)(key[m].w[q >> 3] >> ((q & 7) * 8));
                }
              }
              if (cnt[m]) {
                const u32 li = (tid + m * TPB) >> wl;
                if (MODE == 0) reinterpret_cast<u32*>(row + KW * 8)[li] = cnt[m];
                else {
                  const u32 ob = r * row_bytes + KW * 8 + (li >> 3);
                  atomicOr(reinterpret_cast<u32*>(img) + (ob >> 2), 1u << (((ob & 3u) << 3) + (li & 7u)));
                }
              }
            }
          }
          lds_barrier();
          if (b0 == 0) {   // the hash set is fully read: owners clear their entries for the next tile
#pragma unroll
            for (int m = 0; m < M; m++) if ((ownm >> m) & 1u) tab[hs[m]] = 0;
          }
          u8* dst = dst0 + (u64)b0 * row_bytes;
          if (MODE == 0) {
            if (((reinterpret_cast<uintptr_t>(dst) | bytes) & 7u) == 0) {
              const u64* src = reinterpret_cast<const u64*>(img);
              u64* d64 = reinterpret_cast<u64*>(dst);
              for (u32 t = tid; t < bytes / 8; t += TPB) d64[t] = src[t];
            } else {
              const u32* src = reinterpret_cast<const u32*>(img);
              u32* d32 = reinterpret_cast<u32*>(dst);
              for (u32 t = tid; t < bytes / 4; t += TPB) d32[t] = src[t];
            }
          } else {
            for (u32 t = tid; t < bytes; t += TPB) dst[t] = img[t];
          }
        }
      } else {
        lds_barrier();
#pragma unroll
        for (int m = 0; m < M; m++) if ((ownm >> m) & 1u) tab[hs[m]] = 0;
      }
      // ---- 7. the prefetched records take the consumed slots ----
#pragma unroll
      for (int m = 0; m < M; m++) {
        if ((consm >> m) & 1u) {
          idx[m] += w;
          if (idx[m] < endv[m]) { key[m] = nkey[m]; cnt[m] = ncnt[m]; }
        }
      }
      PH(6);
      seq++;
    }

    // ---- range done: close the open chunk, flush per-list statistics (NON_SOLID, TOTAL_WO) ----
    if (tid == 0 && ch_used) {
      const u64 sidx = atomicAdd(&T.ctrl[1], 1ULL);
      if (sidx < T.seg_cap) { Seg sg; sg.range = range; sg.seq = ch_seq; sg.row_off = ch_base; sg.nrows = ch_used; sg.pad = 0; T.segs[sidx] = sg; }
      else atomicOr(&T.ctrl[2], (u64)ERR_SEGS_OVERFLOW);
      atomicAdd(&T.ctrl[3], (u64)ch_used);
    }
#pragma unroll
    for (int m = 0; m < M; m++) {
      u32 a = nso[m]; u64 t2 = two[m];
      for (u32 off = 1; off < w; off <<= 1) {      // reduce over the w adjacent lanes of a list
        a += __shfl_xor(a, (int)off); t2 += shfl_xor_u64(t2, (int)off);
      }
      const u32 s = tid + m * TPB, li = s >> wl;
      if (li < N && (s & (w - 1)) == 0 && (a | t2)) {
        if (a) atomicAdd(&T.stats[0 * (u64)N + li], (u64)a);
        atomicAdd(&T.stats[4 * (u64)N + li], t2);
      }
    }
    __syncthreads();
  }
}

// explicit instantiations used by the host side
#ifndef KMX_ROWS_SMALL
template __global__ void k_range_bounds<1>(const TaskDev*, u32);
template __global__ void k_range_bounds<2>(const TaskDev*, u32);
template __global__ void k_range_bounds<3>(const TaskDev*, u32);
template __global__ void k_range_bounds<4>(const TaskDev*, u32);
#endif
template __global__ void k_merge_rows<1, 0, false>(const TaskDev*, const uint2*, u32, u32*);
template __global__ void k_merge_rows<1, 1, false>(const TaskDev*, const uint2*, u32, u32*);
template __global__ void k_merge_rows<2, 0, false>(const TaskDev*, const uint2*, u32, u32*);
template __global__ void k_merge_rows<2, 1, false>(const TaskDev*, const uint2*, u32, u32*);
#ifndef KMX_ROWS_SMALL      // (the small build: keys of one and two words)
template __global__ void k_merge_rows<3, 0, false>(const TaskDev*, const uint2*, u32, u32*);
template __global__ void k_merge_rows<3, 1, false>(const TaskDev*, const uint2*, u32, u32*);
template __global__ void k_merge_rows<4, 0, false>(const TaskDev*, const uint2*, u32, u32*);
template __global__ void k_merge_rows<4, 1, false>(const TaskDev*, const uint2*, u32, u32*);
template __global__ void k_merge_rows<3, 0, true>(const TaskDev*, const uint2*, u32, u32*);
template __global__ void k_merge_rows<3, 1, true>(const TaskDev*, const uint2*, u32, u32*);
template __global__ void k_merge_rows<4, 0, true>(const TaskDev*, const uint2*, u32, u32*);
template __global__ void k_merge_rows<4, 1, true>(const TaskDev*, const uint2*, u32, u32*);
#endif

}  // namespace kmx

// ---- host-side launchers (plain functions so other translation units need no device code) -----
namespace kmx {

static bool rows_big(int kw, u32 n) { return kw >= 3 && n > (u32)cap_of(kw); }
int rows_lds_bytes(int kw, u32 n) { return rows_fixed_bytes(kw, rows_big(kw, n)) + (kw <= 2 ? 384 : NWAVE * kw * 8 + 128); }
int rows_cap(int kw, u32 n) { return cap_of(kw, rows_big(kw, n)); }      // record slots of a tile = the most lists of a task (n: the task's lists; ~0u: the limit)
int rows_wgs_per_cu(int kw) { return kw == 1 ? WGS_PER_CU : 1; }
#ifndef KMX_ROWS_SMALL
u32 rows_chunk_rows(u32 row_bytes) { return std::max(64u, (u32)KMX_CHUNK_BYTES / row_bytes); }
#endif
u32 rows_image_bytes(int kw) { return (u32)rows_emit_bytes(kw); }
#ifdef KMX_PHASE_PROF
void rows_phase_prof_dump()
{
  u64 h[16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(kmx_phase_prof), sizeof(h)) != hipSuccess) return;
  u64 tot = 0; for (int i = 0; i < 9; i++) tot += h[i];
  static const char* nm[9] = {"setup", "bound(+wait)", "consume+pref", "insert", "publish+rank", "decide", "emit+rotate", "-", "-"};
  for (int i = 0; i < 8; i++) fprintf(stderr, "[phase] %-14s %6.2f%%  %llu\n", nm[i], tot ? 100.0 * h[i] / tot : 0.0, h[i]);
  memset(h, 0, sizeof(h));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(kmx_phase_prof), h, sizeof(h));
}
#endif

#ifndef KMX_ROWS_SMALL
hipError_t launch_range_bounds(int kw, const TaskDev* tasks, u32 n_tasks, u32 max_n, u32 max_c, hipStream_t st)
{
  dim3 grid((max_n + 255) / 256, max_c + 1, n_tasks), block(256);
  if (kw == 1) hipLaunchKernelGGL(k_range_bounds<1>, grid, block, 0, st, tasks, max_c);
  else if (kw == 2) hipLaunchKernelGGL(k_range_bounds<2>, grid, block, 0, st, tasks, max_c);
  else if (kw == 3) hipLaunchKernelGGL(k_range_bounds<3>, grid, block, 0, st, tasks, max_c);
  else hipLaunchKernelGGL(k_range_bounds<4>, grid, block, 0, st, tasks, max_c);
  return hipGetLastError();
}
#endif

hipError_t launch_merge_rows(int kw, int mode, const TaskDev* tasks, const uint2* items, u32 n_items, u32* ticket,
                             u32 grid_x, u32 max_n, hipStream_t st)
{
  const int lds = rows_lds_bytes(kw, max_n);
  dim3 grid(grid_x), block(TPB);
#define KMX_LAUNCH(KW_, MODE_, BIG_)                                                                              \
  do {                                                                                                      \
    hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_merge_rows<KW_, MODE_, BIG_>),           \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);                   \
    if (e_ != hipSuccess) return e_;                                                                        \
    hipLaunchKernelGGL((k_merge_rows<KW_, MODE_, BIG_>), grid, block, lds, st, tasks, items, n_items, ticket);    \
  } while (0)
  if (kw == 1 && mode == 0) KMX_LAUNCH(1, 0, false);
  else if (kw == 1 && mode == 1) KMX_LAUNCH(1, 1, false);
  else if (kw == 2 && mode == 0) KMX_LAUNCH(2, 0, false);
  else if (kw == 2 && mode == 1) KMX_LAUNCH(2, 1, false);
#ifndef KMX_ROWS_SMALL
  else if (kw == 3 && mode == 0 && rows_big(3, max_n)) KMX_LAUNCH(3, 0, true);
  else if (kw == 3 && mode == 1 && rows_big(3, max_n)) KMX_LAUNCH(3, 1, true);
  else if (kw == 4 && mode == 0 && rows_big(4, max_n)) KMX_LAUNCH(4, 0, true);
  else if (kw == 4 && mode == 1 && rows_big(4, max_n)) KMX_LAUNCH(4, 1, true);
  else if (kw == 3 && mode == 0) KMX_LAUNCH(3, 0, false);
  else if (kw == 3 && mode == 1) KMX_LAUNCH(3, 1, false);
  else if (kw == 4 && mode == 0) KMX_LAUNCH(4, 0, false);
  else if (kw == 4 && mode == 1) KMX_LAUNCH(4, 1, false);
#endif
  else return hipErrorInvalidValue;
#undef KMX_LAUNCH
  return hipGetLastError();
}

}  // namespace kmx
