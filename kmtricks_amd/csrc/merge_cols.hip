// merge_cols.hip -- column-blocked streaming merge for large cohorts of related samples (COUNT and PA rows) on gfx950.
// Same results as k_merge_rows / k_merge_pivot (reference include/kmtricks/merge.hpp:183-286, 441-558).
//
// k_merge_pivot gives every list a 16-record register window and a workgroup all N lists, so a list hands
// over ~13 records (156 bytes) per tile: each 128-byte line of a list is touched in two tiles and, with
// 32 workgroups x 1000 lists per L2, fetched twice.  Here a workgroup owns a BLOCK of <= 128 lists (columns)
// of a key range and gives each a 128-record window (8 adjacent lanes x 16 slots): ~112 records (1.3 KB) per
// list and tile, 1.13x line traffic instead of 1.85x, and an eighth of the per-tile fixed work per record.
// What makes that possible:
//   * the ROW KEYS are known before the merge runs: a handful of the task's lists are merged first
//     (k_cols_skel, same recurrence-min), and the keys that merge keeps -- every one of them is kept by the
//     full merge too -- are the rows (k_cols_prep gathers them into one ascending array and cuts it into
//     the task's key ranges).  Row r of the result is row key r: every column block writes its slice of the
//     row at a position known up front, no cross-block ranking, no row directory;
//   * a tile is rt consecutive row keys; a record below the tile's upper key is consumed, its slot is
//     refilled in place with the record one window further (one global_load_dwordx3), a record whose key
//     is a row key (one read of a collision-free LDS table built per tile) is deposited into the block's LDS
//     image of the tile, and the image leaves as rt slices of ~500 bytes;
//   * a solid record whose key is NOT a row key (sample-private k-mers, and the k-mers two or three samples share) is
//     appended to a per-(half tile, block, wave) slice in HBM.  k_cols_sparse then takes those entries a slice group at a
//     time across the blocks, sorts the keys that can reach the recurrence-min in LDS and writes THEIR rows behind the row
//     keys' rows (with a directory: k_cols_gather interleaves the two when the body is asked for).  A task whose slices
//     overflow (beyond the extension a wave can claim in the EXT build), or with a tile no collision-free table was found for, is handed back (ERR_FALLBACK: the driver re-runs it
//     with k_merge_pivot / k_merge_rows).  Results never depend on how well the row keys cover the lists.
// Share-min (rescue, merge.hpp:210-247; round 3): the RESC builds.  A row key has at least recurrence-min solid records, so with
// share-min <= recurrence-min EVERY record of a row key's row is written (a non-solid one is rescued); the records that are no row
// keys are set aside solid or not (a flag in the entry), and k_cols_sparse, which sees all records of such a key together, counts
// the solid ones: the run is a row from recurrence-min of them, its non-solid records are rescued from share-min of them (statistics
// for every key, kept or not, as the reference accumulates them).  Recurrence-min 0 takes the same builds: a key only non-solid
// records hold is a row of zeros there.
// Applicable to COUNT and PA rows, 64- and 128-bit keys, share-min <= max(1, recurrence-min) (count rows: any share-min, with k_share_fix
// of kmx_api.hip behind the pair); chosen from 192 lists and recurrence-min <= 21
// (the row keys come from 8..32 of the lists, more for a larger recurrence-min: cols_row_lists in kmx_api.hip).
#include "kmx_host.hpp"
#include <algorithm>
#include <cstdio>
#include <cstring>

// This file is compiled twice: as it is for 64-bit keys (k <= 31, hashes), and through merge_cols_k2.hip with KMX_CL_KW = 2 for
// 128-bit keys (32 <= k <= 63: 5-dword records, 8-slot windows, one 32-byte-entry row table).  Each build lives in its own
// namespace and hands kmx_api.hip its entry points through a ColsOps table.
#ifndef KMX_CL_KW
#define KMX_CL_KW 1
#endif
#if KMX_CL_KW == 1
#define CLNS cols_k1
#elif KMX_CL_KW == 2
#define CLNS cols_k2
#elif KMX_CL_KW == 3
#define CLNS cols_k3
#else
#define CLNS cols_k4
#endif

namespace kmx { namespace CLNS {

#ifndef KMX_CL_TPB
#define KMX_CL_TPB 1024
#endif
constexpr int CL_TPB = KMX_CL_TPB;       // 1024: one workgroup per CU; 512: two, each with half the lists of a block and half the image
constexpr int CL_WGS = 1024 / CL_TPB;
#ifndef KMX_CL_G
#define KMX_CL_G 8
#endif
constexpr int CL_G = KMX_CL_G;           // adjacent lanes per list (4: 64-record windows, 256 lists per block; 8: 128 and 128)
constexpr int KW = KMX_CL_KW;            // 64-bit words per key
constexpr int RB = KW * 8 + 4;           // bytes per record (key words, u32 count)
constexpr int RD = RB / 4;               // ... in dwords
constexpr int EW = KW + 1;               // u64 words per set-aside entry: the key, then list << 32 | count
constexpr int CL_U = KW == 1 ? 16 : 8;   // window slots per lane (a 128-bit record is 5 registers: 8 slots keep the kernel under 128 VGPRs)
constexpr int CL_W = CL_G * CL_U;        // records per window
constexpr int CL_NB = CL_TPB / CL_G;     // lists per column block
constexpr int CL_IMG = 61440 / CL_WGS;   // LDS image bytes (rt rows x nb u32 counts)
constexpr int CL_RT = CL_W * 7 / 8;      // row keys per tile (< window: a similar list needs no second round)
constexpr int CL_KPL = (CL_RT + 63) / 64;   // row keys per lane of wave 0 (which builds the row table)
constexpr int CL_NT = (CL_KPL == 1 && CL_WGS == 1 && KW == 1) ? 2 : 1;  // row tables: two of 2048 entries (the next tile's is built beside this tile's), or one
constexpr int CL_PT = CL_KPL == 1 ? 2048 : 4096;      // row-key table entries
constexpr int CL_PTSHIFT = CL_PT == 2048 ? 21 : 20;
constexpr int CL_HALVES = CL_KPL;        // set-aside slices per tile: one per 56 rows (k_cols_sparse takes a slice group at a time)
constexpr int CL_SEEDS = 256;            // hashes tried per tile for a collision-free table: 64 cheap ones, then 64-bit multiplicative ones
constexpr int CL_OVW = 128;              // keys per (slice group, block, wave) of records that are not row keys
constexpr int CL_XS = 3 * CL_OVW;        // ... and of the extension a wave claims when its slice is full (an outlier sample among its lists)
constexpr int CL_NW = CL_TPB / 64;
constexpr int CK_BITS = 1 << 17;         // k_cols_sparse: bits of the key map (16 KB of LDS)
constexpr int CP_MAXSEG = 2048;          // k_cols_prep: segments of the row-key merge

__device__ u32 kmx_cols_dbg[16];     // why tasks were handed back (KMX_TRACE=1 prints and clears them); [8..11]: look-backs given up, the last one's task order / group / entries still to add; [12]: tasks handed back for it

namespace {

typedef u32 u32x3 __attribute__((ext_vector_type(3), aligned(4)));
typedef u32 u32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef __attribute__((address_space(1))) const u32x3 gu32x3;
typedef __attribute__((address_space(1))) const u32x4 gu32x4;
typedef __attribute__((address_space(1))) u64 gu64w;
typedef u64 u64x2 __attribute__((ext_vector_type(2), aligned(8)));      // a 16-byte store at an 8-byte aligned place (a count row starts at a multiple of its 8 * k bytes)
__device__ __forceinline__ u32 cl_uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 cl_uni64(u64 v) { return (u64)cl_uni((u32)v) | ((u64)cl_uni((u32)(v >> 32)) << 32); }

// ---- the key type of this build: u64, or two words compared most significant first (kmer.hpp:262-268) ----
#if KMX_CL_KW == 1
typedef u64 CKey;
struct CRec { u32x3 v; };                                      // key low dword, key high dword, count
struct ClEnt { u32 klo, khi, idx, pad; };                      // row table entry: idx = row + 1, 0 = empty
__device__ __forceinline__ bool ck_lt(CKey a, CKey b) { return a < b; }
__device__ __forceinline__ bool ck_eq(CKey a, CKey b) { return a == b; }
__device__ __forceinline__ CKey ck_inf() { return ~0ULL; }
__device__ __forceinline__ u64 ck_fold(CKey k) { return k; }    // the 64 bits the hashes work on
__device__ __forceinline__ CKey ck_uni(CKey k) { return cl_uni64(k); }
__device__ __forceinline__ CKey ck_load(const u8* p) { return load_key<1>(p).w[0]; }
__device__ __forceinline__ void ck_store(u32* p, CKey k) { p[0] = (u32)k; p[1] = (u32)(k >> 32); }
__device__ __forceinline__ CKey cl_key(const CRec& r) { return (u64)r.v.x | ((u64)r.v.y << 32); }
__device__ __forceinline__ u32 cl_cnt(const CRec& r) { return r.v.z; }
__device__ __forceinline__ CRec cl_none() { CRec r; r.v.x = ~0u; r.v.y = ~0u; r.v.z = 0; return r; }
__device__ __forceinline__ CRec cl_load(gu32* p) { CRec r; r.v = *(gu32x3*)p; return r; }
__device__ __forceinline__ bool ent_hit(const ClEnt& e, CKey k) { return (((u64)e.khi << 32) | e.klo) == k && e.idx != 0; }      // (one 64-bit compare)
__device__ __forceinline__ void ent_set(ClEnt& e, CKey k) { e.klo = (u32)k; e.khi = (u32)(k >> 32); }
__device__ __forceinline__ ClEnt ent_load(const ClEnt* tab, u32 h) { const uint4 v = reinterpret_cast<const uint4*>(tab)[h]; ClEnt e; e.klo = v.x; e.khi = v.y; e.idx = v.z; e.pad = v.w; return e; }      // one 16-byte LDS read
#else
struct CKey { u64 lo, hi; };
struct CRec { u32x4 k; u32 c; };                               // key (low word first), count
struct ClEnt { u64 lo, hi; u32 idx, pad[3]; };                 // 32 bytes
__device__ __forceinline__ bool ck_lt(CKey a, CKey b) { return a.hi != b.hi ? a.hi < b.hi : a.lo < b.lo; }
__device__ __forceinline__ bool ck_eq(CKey a, CKey b) { return a.lo == b.lo && a.hi == b.hi; }
__device__ __forceinline__ CKey ck_inf() { CKey k; k.lo = ~0ULL; k.hi = ~0ULL; return k; }
// (the product's high part folded down: a plain lo ^ hi * c cancels when the low word is itself a multiple of the high one.  By 29,
//  not 32: a difference in the high word's upper half must not put the same bits into both halves of the fold -- cl_thash xors them)
__device__ __forceinline__ u64 ck_fold(CKey k) { const u64 h = k.hi * 0x9E3779B97F4A7C15ULL; return k.lo ^ h ^ (h >> 29); }
__device__ __forceinline__ CKey ck_uni(CKey k) { CKey r; r.lo = cl_uni64(k.lo); r.hi = cl_uni64(k.hi); return r; }
__device__ __forceinline__ CKey ck_load(const u8* p) { const Key<2> k = load_key<2>(p); CKey r; r.lo = k.w[0]; r.hi = k.w[1]; return r; }
__device__ __forceinline__ void ck_store(u32* p, CKey k) { p[0] = (u32)k.lo; p[1] = (u32)(k.lo >> 32); p[2] = (u32)k.hi; p[3] = (u32)(k.hi >> 32); }
__device__ __forceinline__ CKey cl_key(const CRec& r) { CKey k; k.lo = (u64)r.k.x | ((u64)r.k.y << 32); k.hi = (u64)r.k.z | ((u64)r.k.w << 32); return k; }
__device__ __forceinline__ u32 cl_cnt(const CRec& r) { return r.c; }
__device__ __forceinline__ CRec cl_none() { CRec r; r.k.x = ~0u; r.k.y = ~0u; r.k.z = ~0u; r.k.w = ~0u; r.c = 0; return r; }
__device__ __forceinline__ CRec cl_load(gu32* p) { CRec r; r.k = *(gu32x4*)p; r.c = p[4]; return r; }
__device__ __forceinline__ bool ent_hit(const ClEnt& e, CKey k) { return e.lo == k.lo && e.hi == k.hi && e.idx != 0; }
__device__ __forceinline__ void ent_set(ClEnt& e, CKey k) { e.lo = k.lo; e.hi = k.hi; }
__device__ __forceinline__ ClEnt ent_load(const ClEnt* tab, u32 h)
{ // the key and the row: 20 of the entry's 32 bytes
  const uint4 v = reinterpret_cast<const uint4*>(tab)[2 * h]; const u32 ix = reinterpret_cast<const u32*>(tab)[8 * h + 4];
  ClEnt e; e.lo = (u64)v.x | ((u64)v.y << 32); e.hi = (u64)v.z | ((u64)v.w << 32); e.idx = ix; return e;
}
#endif
__device__ __forceinline__ bool ck_le(CKey a, CKey b) { return !ck_lt(b, a); }
__device__ __forceinline__ void cl_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// the row keys of a tile sit in a PERFECT hash table: wave 0 tries multipliers until the (<= 56) keys land in
// distinct entries of the 1024, so a lookup is one LDS read and one compare -- no probing, no branch
// (a hash = 24-bit multiplier | shift << 24: the shift picks which key bits are folded into the 24 that get multiplied,
//  so two keys that agree in those 24 bits under one hash do not under the next)
// Keys of real minimizer partitions are structured (the row keys of a tile share their leading nucleotides, neighbours in
// the genome are shifted copies of each other): for a tile in a thousand none of the cheap hashes is collision free.  Those
// tiles use a second family -- a 64-bit multiplicative hash of the whole key (bit 31 of the hash word set; a uniform branch).
__device__ __forceinline__ u32 cl_thash(CKey key, u32 hf)
{
  const u64 k = ck_fold(key);
  if (hf & 0x80000000u) {
    const u64 m = 0x9E3779B97F4A7C15ULL + 2ULL * (u64)(hf & 0xFFFFu) * 0xBF58476D1CE4E5B9ULL;      // odd
    return (u32)((k * m) >> (CL_PT == 2048 ? 53 : 52));      // the product's TOP bits: every key bit counts (a k-mer and its variant with one
                                                             // substitution near the front differ in one high bit and sit in the same tile)
  }
#if KMX_CL_KW == 1
  const u32 x = ((u32)k ^ (u32)(k >> (hf >> 24))) & 0xFFFFFFu;
#else
  // 128-bit keys of a partition are ~2^100 apart: a k-mer and its variant with ONE substitution anywhere in its low 50 nucleotides
  // are neighbours in a tile, so every bit of the fold has to reach the 24 that get multiplied: f ^ f >> 8 takes bit i of either half
  // to bit i or i - 8.  (With the 64-bit family's window of 48 bits three tiles in four of a real cohort went through all the cheap
  // tries first, the workgroup waiting: 63 % of the kernel on configs[4]'s lists from the count stage.)
  const u32 f = (u32)k ^ (u32)(k >> 32);
  const u32 x = (f ^ (f >> 8)) & 0xFFFFFFu;
#endif
  return ((u32)__umul24(x, hf) >> CL_PTSHIFT) & (u32)(CL_PT - 1);      // (__umul24 takes the low 24 bits of hf, and returns int)
}
__device__ __forceinline__ u32 cl_mult(u32 seed)
{
  if (KW == 2 && seed >= 16u) return 0x80000000u | (seed - 15u);      // (its cheap hashes differ in the multiplier only: 16 tries tell)
  if (seed >= 64u) return 0x80000000u | (seed - 63u);
  return ((0x9E3779u + seed * 0x5A6B2u) & 0xFFFFFFu) | ((13u + (seed * 7u) % 19u) << 24);
}
__device__ __forceinline__ u32 cl_mix(CKey key)
{
  const u64 k = ck_fold(key);
  u32 x = (u32)k ^ ((u32)(k >> 32) * 0x9E3779B1u);
  x *= 0x85EBCA6Bu; x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 13;
  return x;
}
// (Round 5 tried four hashes at once -- a bit map of LDS per hash, a key sets its bit under each and sees whether it was set, the first
//  hash nobody clashed under wins: k_merge_cols 2.29 -> 2.50 ms.  On real tiles the first or second cheap hash is collision free; four
//  hashes' worth of work every time is more than the one or two tries below.)
// wave 0: lane j holds row keys j, j + 64, ...  -> the hash, 0 when no try worked; slot[x] = entry of my key x
__device__ __forceinline__ u32 cl_build(ClEnt* tab, const CKey (&key)[CL_KPL], const bool (&have)[CL_KPL], u32 lane, u32 (&slot)[CL_KPL])
{
  for (u32 s = 0; s < (u32)CL_SEEDS; s++) {
    const u32 mult = cl_mult(s);
    u32 h[CL_KPL], old[CL_KPL];
    bool clash = false;
#pragma unroll
    for (int x = 0; x < CL_KPL; x++) {
      h[x] = cl_thash(key[x], mult);
      old[x] = have[x] ? atomicCAS(&tab[h[x]].idx, 0u, lane + 64u * x + 1) : 0u;
      clash |= old[x] != 0;
    }
    if (__ballot(clash) == 0) {
#pragma unroll
      for (int x = 0; x < CL_KPL; x++) { if (have[x]) ent_set(tab[h[x]], key[x]); slot[x] = h[x]; }
      return mult;
    }
#pragma unroll
    for (int x = 0; x < CL_KPL; x++) if (have[x] && old[x] == 0) atomicExch(&tab[h[x]].idx, 0u);      // take my claims back, next hash
  }
  return 0;
}

}  // namespace

// ---- row keys, step 1: the keys a merge of a few lists keeps.  One workgroup per key range of <= SK_CAP records:
//      the solid keys of the range are sorted in LDS (bitonic), a key is kept when its run is at least
//      recurrence-min long, kept keys leave in order.  Segment j of the result = range j (k_cols_prep strings them
//      together). ----
#ifndef KMX_SK_TPB
#define KMX_SK_TPB 256
#endif
constexpr int SK_TPB = KMX_SK_TPB;
constexpr int SK_CAP = 2048;             // records per range (all lists), and the row capacity of a range

__global__ __launch_bounds__(SK_TPB)
void k_cols_skel(const TaskDev* __restrict__ subs, const uint2* __restrict__ items, u32 n_items)
{
  __shared__ CKey ks[SK_CAP];
  __shared__ u32 wsum[SK_TPB / 64];
  const u32 item = blockIdx.x;
  if (item >= n_items) return;
  const TaskDev& S = subs[items[item].x];
  const u32 range = items[item].y, N = S.N, rec_min = max(1u, S.rec_min);
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  u32 total = 0;
  for (u32 i = 0; i < N; i++) total += S.bounds[(u64)(range + 1) * N + i] - S.bounds[(u64)range * N + i];
  Seg sg; sg.range = range; sg.seq = 0; sg.row_off = (u64)range * SK_CAP; sg.nrows = 0; sg.pad = 0;
  if (total > (u32)SK_CAP) {   // lists that do not line up with the range cuts: no row keys, the general kernels take the task
    if (tid == 0) { atomicOr(&S.ctrl[2], (u64)ERR_ROWS_OVERFLOW); S.segs[range] = sg; atomicAdd(&S.ctrl[1], 1ULL); }
    return;
  }
  // The lists are sorted already: each gets a run of L slots (ascending runs and descending ones in turn, padded with
  // the largest key), and only the MERGING stages of the bitonic network run (30 steps instead of 66 for 8 x 256).  A
  // non-solid record (it counts for nothing: largest key in the middle of its run) or runs that do not fit make it a
  // plain sort of the packed keys.
  __shared__ u32 irregular;
  u32 maxn = 1;
  for (u32 i = 0; i < N; i++) maxn = max(maxn, S.bounds[(u64)(range + 1) * N + i] - S.bounds[(u64)range * N + i]);
  u32 L = 1; while (L < maxn) L <<= 1;
  u32 SP = 1; while (SP < N) SP <<= 1;
  const bool runs = (u64)SP * L <= (u64)SK_CAP;
  u32 P = 2;
  if (runs) P = max(2u, SP * L); else while (P < total) P <<= 1;
  if (tid == 0) irregular = 0;
  if (runs) for (u32 t = tid; t < P; t += SK_TPB) ks[t] = ck_inf();
  __syncthreads();
  {   // SK_TPB / N threads per list, every list at once
    const u32 tpl = max(1u, (u32)SK_TPB / N);
    for (u32 i = tid / tpl; i < N; i += (u32)SK_TPB / tpl) {
      u32 off = 0;
      if (!runs) for (u32 j = 0; j < i; j++) off += S.bounds[(u64)(range + 1) * N + j] - S.bounds[(u64)range * N + j];
      const u32 lo = S.bounds[(u64)range * N + i], n = S.bounds[(u64)(range + 1) * N + i] - lo, smin = S.soft_min[i];
      const u8* base = S.recs[i] + (u64)lo * RB;
      for (u32 e = tid % tpl; e < n; e += tpl) {
        const u32* rp = reinterpret_cast<const u32*>(base + (u64)e * RB);
        const bool solid = rp[RD - 1] >= smin;
        const CKey key = solid ? ck_load(reinterpret_cast<const u8*>(rp)) : ck_inf();      // (a non-solid record counts for nothing)
        if (!runs) ks[off + e] = key;
        else { ks[i * L + ((i & 1u) ? L - 1 - e : e)] = key; if (!solid) irregular = 1; }
      }
    }
  }
  if (!runs) for (u32 t = total + tid; t < P; t += SK_TPB) ks[t] = ck_inf();
  __syncthreads();
  const u32 k0 = (runs && !irregular) ? 2 * L : 2;      // runs of L are sorted, in the directions the network expects
  for (u32 k = k0; k <= P; k <<= 1) {
    for (u32 j = k >> 1; j > 0; j >>= 1) {
      for (u32 t = tid; t < P / 2; t += SK_TPB) {
        const u32 a = ((t & ~(j - 1)) << 1) | (t & (j - 1)), b = a | j;      // the pair (a, a + j)
        const CKey x = ks[a], y = ks[b];
        const bool up = (a & k) == 0;
        if (ck_lt(y, x) == up) { ks[a] = y; ks[b] = x; }
      }
      __syncthreads();
    }
  }
  // kept: first record of a run of >= recurrence-min equal keys
  const u32 per = P / SK_TPB ? P / SK_TPB : 1;
  u32 mine = 0, keptm = 0, nsolid = 0, ncov = 0;      // ... and how many of the lists' solid records the kept keys cover
  for (u32 x = 0; x < per; x++) {
    const u32 i = tid * per + x;
    if (i < P) {
      const CKey k = ks[i];
      const bool real = !ck_eq(k, ck_inf());
      const bool kept = real && (i == 0 || !ck_eq(ks[i - 1], k)) && i + rec_min - 1 < P && ck_eq(ks[i + rec_min - 1], k);
      keptm |= (kept ? 1u : 0u) << x; mine += kept ? 1u : 0u;
      if (real && (range & 7u) == 0) {      // (an estimate: every eighth range)
        nsolid++;
        bool cov = false;      // my record's run of equal keys is at least recurrence-min long
        for (u32 j = 0; j < rec_min && !cov; j++) cov = i >= j && i - j + rec_min - 1 < P && ck_eq(ks[i - j], k) && ck_eq(ks[i - j + rec_min - 1], k);
        ncov += cov ? 1u : 0u;
      }
    }
  }
  if ((range & 7u) == 0) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { nsolid += __shfl_xor(nsolid, off); ncov += __shfl_xor(ncov, off); }
    if (lane == 0 && nsolid) { atomicAdd(&S.ctrl[4], (u64)nsolid); atomicAdd(&S.ctrl[5], (u64)ncov); }
  }
  const u32 incl = wave_incl_scan(mine, (int)lane);
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  u32 base = incl - mine, nk = 0;
  for (u32 w = 0; w < SK_TPB / 64; w++) { if (w < wave) base += wsum[w]; nk += wsum[w]; }
  CKey* out = reinterpret_cast<CKey*>(S.out) + (u64)range * SK_CAP;
  for (u32 x = 0; x < per; x++) if ((keptm >> x) & 1u) out[base++] = ks[tid * per + x];
  if (tid == 0) { sg.nrows = nk; S.segs[range] = sg; atomicAdd(&S.ctrl[1], 1ULL); atomicAdd(&S.ctrl[3], (u64)nk); }
}

// ---- row keys: gather the kept keys of the few-lists merge into one ascending array, cut it into ranges ----
constexpr int CP_TPB = 1024;
__global__ __launch_bounds__(CP_TPB)
void k_cols_prep(const TaskDev* __restrict__ tasks, const TaskDev* __restrict__ subs, const ColsDev* __restrict__ cols)
{
  const TaskDev& T = tasks[blockIdx.x];
  const TaskDev& S = subs[blockIdx.x];
  const ColsDev& C = cols[blockIdx.x];
  __shared__ u32 s_off[CP_MAXSEG];
  __shared__ u64 s_ord[CP_MAXSEG];
  __shared__ u32 s_n[CP_MAXSEG];
  const u32 tid = threadIdx.x;
  const u64 serr = S.ctrl[2], nseg64 = S.ctrl[1], rows = S.ctrl[3];
  if (tid == 0) T.ctrl[7] = rows;      // (the host sizes the next batches' stores of row keys' rows from it, whatever becomes of this task)
  const u64 slots = rows / C.rt + T.c + 2;
  const bool dense_small = C.dense != nullptr && rows > (u64)C.dense_cap;      // (the host's estimate of the row keys fell short: its own flag, see ERR_DENSE_CAP)
  const bool bad = serr != 0 || nseg64 > (u64)CP_MAXSEG || nseg64 > S.seg_cap || rows > T.out_cap_rows || rows > 0xFFFFFF00ULL ||
                   slots > C.slots_cap || dense_small;
  // the share of the merged lists' solid records that the row keys do not cover estimates what every list would set aside:
  // above 1/8 the slices would overflow (and k_merge_pivot gives up at the same point): straight to k_merge_rows
  const u64 nsolid = S.ctrl[4], ncov = S.ctrl[5];
  const bool divergent = (nsolid - ncov) * 8 > nsolid;
  if (divergent && !bad) {
    if (tid == 0) { atomicOr(&T.ctrl[2], (u64)(ERR_FALLBACK | ERR_DIVERGENT)); *C.nskel = 0; atomicAdd(&kmx_cols_dbg[4], 1u); }
    for (u32 j = tid; j <= T.c; j += CP_TPB) { C.rbounds[j] = 0; if (C.gbase) C.gbase[j] = 0; }
    return;
  }
  if (bad) {   // the row keys could not be built (arena too small, ...): the general kernels take the task
    const bool only_dense = dense_small && !(serr != 0 || nseg64 > (u64)CP_MAXSEG || nseg64 > S.seg_cap || rows > T.out_cap_rows || rows > 0xFFFFFF00ULL || slots > C.slots_cap);
    if (tid == 0) { atomicOr(&T.ctrl[2], (u64)(ERR_FALLBACK | (only_dense ? ERR_DENSE_CAP : 0))); *C.nskel = 0; }
    for (u32 j = tid; j <= T.c; j += CP_TPB) { C.rbounds[j] = 0; if (C.gbase) C.gbase[j] = 0; }      // (no slice groups: k_cols_sparse has nothing to wait for)
    return;
  }
  const u32 nseg = (u32)nseg64;
  const u32 srb = S.row_bytes;
  for (u32 i = tid; i < nseg; i += CP_TPB) { const Seg a = S.segs[i]; s_ord[i] = ((u64)a.range << 32) | a.seq; s_n[i] = a.nrows; }
  __syncthreads();
  for (u32 i = tid; i < nseg; i += CP_TPB) {   // position of a segment = rows of the segments in front of it in (range, seq) order
    const u64 mine = s_ord[i];
    u32 off = 0;
    for (u32 j = 0; j < nseg; j++) off += s_ord[j] < mine ? s_n[j] : 0u;
    s_off[i] = off;
  }
  __syncthreads();
  for (u32 i = tid >> 6; i < nseg; i += CP_TPB / 64) {   // a wave per segment
    const Seg a = S.segs[i];
    const u8* src = S.out + a.row_off * srb;
    for (u32 r = tid & 63u; r < a.nrows; r += 64) reinterpret_cast<CKey*>(C.skel)[(u64)s_off[i] + r] = ck_load(src + (u64)r * srb);
  }
  __syncthreads();
  const u32 np = T.len[T.pivot];
  for (u32 j = tid; j <= T.c; j += CP_TPB) {
    u32 res;
    if (j == 0) res = 0;
    else if (j == T.c) res = (u32)rows;
    else {   // same boundary keys as k_range_bounds: Q_j = pivot[j * len_pivot / c]
      const u32 pos = (u32)(((u64)j * np) / T.c);
      const CKey q = ck_load(T.recs[T.pivot] + (u64)pos * RB);
      u32 lo = 0, hi = (u32)rows;
      while (lo < hi) { const u32 mid = lo + ((hi - lo) >> 1); if (ck_lt(reinterpret_cast<const CKey*>(C.skel)[mid], q)) lo = mid + 1; else hi = mid; }
      res = lo;
    }
    C.rbounds[j] = res;
  }
  if (tid == 0) {
    *C.nskel = (u32)rows;
    T.ctrl[0] = rows; T.ctrl[1] = 1; T.ctrl[3] = rows;
    Seg sg; sg.range = 0; sg.seq = 0; sg.row_off = 0; sg.nrows = (u32)rows; sg.pad = 0;
    T.segs[0] = sg;
  }
  if (C.gbase) {
    // file order out of the kernels: the task's slice groups numbered in key order -- k_cols_sparse hands them out by ticket in that
    // order (a group's place in the arena is the sum of the rows of the groups in front of it: decoupled look-back)
    __shared__ u32 s_ng;
    __syncthreads();
    if (tid == 0) {
      u32 g = 0;
      for (u32 j = 0; j < T.c; j++) { C.gbase[j] = g; const u32 n = C.rbounds[j + 1] - C.rbounds[j]; g += max(1u, (n + C.rt - 1) / C.rt) * (u32)CL_HALVES; }
      C.gbase[T.c] = g; s_ng = g;
      atomicMax(C.gmax, g);
    }
    __syncthreads();
    const u32 ng = s_ng;
    for (u32 g = tid; g < ng; g += CP_TPB) {
      u32 lo = 0, hi = T.c;      // the last range with gbase <= g
      while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (C.gbase[mid] <= g) lo = mid; else hi = mid; }
      C.gmap[g] = make_uint4(lo + 1, g - C.gbase[lo], C.rbounds[lo], C.rbounds[lo + 1]);      // (range + 1: 0 = no such group; all k_cols_sparse needs to start on the group, in one load)
    }
  }
}

__device__ const u32 kmx_cols_sentinel[8] = {~0u, ~0u, KW == 1 ? 0u : ~0u, KW == 1 ? 0u : ~0u, 0u, 0u, 0u, 0u};      // the record "past the end of a list": largest key, count 0

#ifdef KMX_PHASE_PROF
__device__ u64 kmx_cols_prof[8];
__device__ u64 kmx_sparse_prof[8];
#ifndef KMX_PROF_TID
#define KMX_PROF_TID 512
#endif
#endif

// ---- the merge: work item = (task, key range, column block) ----------------------------------------------
// (Round 5 measured what each part of the slot walk costs on configs[2], profiles/r05b_cols_levels.jsonl: the window's loads and the tile
//  skeleton alone 1.84 ms, + the table lookups 1.85, + the deposits 2.03, + the records that are no row keys 2.68 -- the branch per
//  window slot, not its instructions: staged per lane in LDS and appended once per round they cost half as much as long as nothing
//  overflows the stage, but every way of handling the overflow that was tried -- a branch per slot, per four slots, a second walk --
//  gave it back in spilled registers around the loop, and the window's loads issued any later than they are cost more than all of it.)
// (Also round 5, on the NAR build: two byte images in the image's room, a tile's image streamed out during the NEXT tile's walk -- right
//  behind the walk's vmcnt(0), so that no later wait meets the stores -- instead of between two barriers of its own: correct, 2.29 ->
//  2.37 ms, five registers spilled around the loop.  A walk skipped by the waves that have nothing below the tile's upper key in a
//  further round: +-0, further rounds are rare.)
// EXT: a wave whose set-aside slice is full claims an extension (cohorts with outlier samples; the plain build hands such a task
// back and the context's next batches use this one: 1-2 % slower on cohorts that never need it)
constexpr u64 CL_NONSOLID = 1ULL << 63;      // set-aside entry of a RESC build: the record is below its list's soft-min
// ORD: the row keys' rows go to the side store k_cols_sparse copies them from (rows at their final place, kmx_set_file_order) -- a
// build of its own: as a run-time switch the side store's code cost the other build seven more spilled registers (round 4: 2.53 -> 2.70 ms)
// NAR (count rows of an ORD build): the side store holds ONE BYTE per count (C.dnarrow) -- and so does the tile's image in LDS: a
// deposit is a byte (255 = "the count is in the 4-byte row", written there by the lane that holds it: rare), a row's slice of the
// block leaves as 16-byte pieces (round 4 kept the u32 image and squeezed it on the way out: a row at a time per wave, 2-byte
// stores)
template <int MODE, bool EXT, bool RESC, bool ORD, bool NAR>      // MODE 0: count rows (u32 per list), 1: presence/absence rows (a bit per list, LSB first)
__global__ __launch_bounds__(CL_TPB, CL_WGS)
void k_merge_cols(const TaskDev* __restrict__ tasks, const ColsDev* __restrict__ cols, const uint2* __restrict__ items,
                  u32 n_items, u32* ticket)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32* const img = reinterpret_cast<u32*>(smem);                                   // [rt][nb] counts of the tile
  ClEnt* const ptab = reinterpret_cast<ClEnt*>(smem + CL_IMG);                     // [CL_NT][CL_PT] row key -> row
  u32* const sh = reinterpret_cast<u32*>(smem + CL_IMG + CL_NT * CL_PT * sizeof(ClEnt)); // [0] item [1..3] "another round" flags, used in turn [4],[5] the tables' multipliers
  const u32 dummy = (u32)(CL_IMG + CL_NT * CL_PT * sizeof(ClEnt)) / 4 + 16 + (u32)(threadIdx.x & 63);   // image index of a scratch word of my own (deposits that are none)

  const int tid = threadIdx.x, lane = tid & 63;
  const u32 wave = cl_uni((u32)tid >> 6);
  for (int t = tid; t < CL_IMG / 16; t += CL_TPB) reinterpret_cast<uint4*>(img)[t] = make_uint4(0, 0, 0, 0);
  for (int t = tid; t < CL_NT * CL_PT; t += CL_TPB) ptab[t].idx = 0;
  if (tid == 0) { sh[1] = 0; sh[2] = 0; sh[3] = 0; }
#ifdef KMX_PHASE_PROF
  long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long pc = clock64();
#define CLPH(i) do { const long long n_ = clock64(); pt[i] += n_ - pc; pc = n_; } while (0)
#else
#define CLPH(i) do {} while (0)
#endif

  for (;;) {
    if (tid == 0) sh[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    const u32 item = cl_uni(sh[0]);
    __syncthreads();
    if (item >= n_items) {
#ifdef KMX_PHASE_PROF
      if (tid == KMX_PROF_TID) for (int i = 0; i < 8; i++) atomicAdd(&kmx_cols_prof[i], (u64)pt[i]);
#endif
      return;
    }
    const TaskDev& T = tasks[items[item].x];
    const ColsDev& C = cols[items[item].x];
    // (ONE decision per workgroup: the word is raised by other workgroups while this one reads it -- threads that saw it and threads that
    //  did not went different ways through the barriers below, and a launch in a thousand took minutes: round 5, scripts/dev/stress_ord.py)
    if (__syncthreads_or((__hip_atomic_load(&T.ctrl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (u64)(ERR_FALLBACK | ERR_ROWS_OVERFLOW)) != 0)) continue;
    // (everything per work item is uniform; the readfirstlanes tell the compiler, which otherwise treats the tile loop's
    //  conditions as divergent and branches on them per window slot)
    const u32 N = cl_uni(T.N), row_bytes = cl_uni(T.row_bytes), nblk = cl_uni(C.nblk), nbs = cl_uni(C.nb), rt = cl_uni(C.rt);
    const u32 range = cl_uni(items[item].y) / nblk, blk = cl_uni(items[item].y) - range * nblk;
    const u32 col0 = blk * nbs, nbl = min(nbs, N - col0);
    const u32 iw = MODE == 0 ? nbs : (nbs + 31) / 32;      // image words per row
    const u32 s_lo = cl_uni(C.rbounds[range]), s_hi = cl_uni(C.rbounds[range + 1]);
    const u32 ntiles = max(1u, (s_hi - s_lo + rt - 1) / rt);
    const u32 slot0 = s_lo / rt + range;
    const CKey* const skel = reinterpret_cast<const CKey*>(C.skel);
    // where the row keys' rows go: the arena, row r at row r (key + payload) -- or, when the rows are to come out in file order,
    // the side store k_cols_sparse copies them from (payload only: the key is in skel)
    constexpr bool ord = ORD;
    u8* const obase = ord ? C.dense : T.out + KW * 8;
    const u32 opitch = ord ? cl_uni(C.dpitch) : row_bytes;
    // ... count rows there as ONE BYTE per count where a block's slice of the row holds none above 254 (the usual case by far): the
    // side store is written here and read again by k_cols_sparse -- a quarter of the bytes both ways.  C.dnarrow row r = [N count
    // bytes, padded to 8][a flag byte per column block: 1 = this block's counts of the row are in the 4-byte row]
    static_assert(!NAR || (MODE == 0 && ORD), "the byte-wide side store is for count rows in file order");
    u8* const nar = NAR ? C.dnarrow : nullptr;
    const u32 npitch = NAR ? cl_uni(C.npitch) : 0u;
    u8* const img8 = reinterpret_cast<u8*>(img);                  // NAR: [rt][nbs] count bytes
    u8* const rowbig = reinterpret_cast<u8*>(smem) + CL_RT * CL_NB;   // NAR: [rt] "this row of the tile holds a count above 254" (behind the image's rt x nb bytes)
    const u32 dummyb = dummy * 4u;

    // CL_G adjacent lanes per list; circular window: lane r, slot u holds the record whose index is == r + CL_G * u
    // (mod CL_W) inside [cur, cur + CL_W)
    const u32 lg = (u32)tid / CL_G, r = (u32)tid & (CL_G - 1);
    const bool on = lg < nbl;
    const u32 li = col0 + (on ? lg : 0u);
    u32 cur = on ? T.bounds[(u64)range * N + li] : 0u;
    const u32 end = on ? T.bounds[(u64)(range + 1) * N + li] : 0u;
    const u32 smin = T.soft_min[li];
    gu32* const base = (gu32*)(uintptr_t)T.recs[li];
    gu32* const sentinel = (gu32*)(uintptr_t)kmx_cols_sentinel;
    CRec rec[CL_U];
#pragma unroll
    for (int u = 0; u < CL_U; u++) {
      const u32 ix = cur + ((r + CL_G * u - cur) & (CL_W - 1));
      rec[u] = cl_none();
      if (ix < end) rec[u] = cl_load(base + (u64)ix * RD);
    }
    u64 tsum = 0; u32 tn = 0;            // TOTAL_WO / NON_SOLID of my share of my list
    u64 rsum = 0; u32 rn = 0;            // RESC: the rescued records of the row keys' rows (their counts, their number)
    const bool rescue = RESC && cl_uni(T.share_min) != 0;      // (a RESC build with share-min 0: recurrence-min 0 -- nothing is rescued, everything is set aside)
    // first tile: row keys, table, (block 0) the key column of the result
    CKey skn[CL_KPL];
    u32 myslot[CL_KPL];                  // wave 0: the table entries of my row keys of the tile in hand
    bool failed = false;
#pragma unroll
    for (int x = 0; x < CL_KPL; x++) { skn[x] = ck_inf(); myslot[x] = 0; }
    if (tid < 64) {
      bool have[CL_KPL];
#pragma unroll
      for (int x = 0; x < CL_KPL; x++) {
        const u32 j = (u32)tid + 64u * x;
        have[x] = j < rt && s_lo + j < s_hi;
        if (have[x]) {
          skn[x] = skel[s_lo + j];
          if (blk == 0 && !ord) ck_store(reinterpret_cast<u32*>(T.out + (u64)(s_lo + j) * row_bytes), skn[x]);
        }
      }
      const u32 mult = cl_build(ptab, skn, have, (u32)tid, myslot);
      if (tid == 0) { sh[4] = mult; if (mult == 0) { failed = true; atomicAdd(&kmx_cols_dbg[0], 1u); } }
    }
    CKey khi_n = ntiles > 1 ? skel[s_lo + rt] : ck_inf();     // upper key of tile 0 (uniform address: scalar load)
    CKey kmid_n = (CL_HALVES > 1 && s_lo + 56 < s_hi) ? skel[s_lo + 56] : ck_inf();      // ... and the key its second slice group starts at
    u32 rnd = 0;                         // round number mod 3
    cl_barrier();
    CLPH(0);

    for (u32 q = 0; q < ntiles; q++) {
      const u32 s0 = s_lo + q * rt;
      const u32 rte = min(rt, s_hi - s0);
      const bool last = cl_uni(q + 1 == ntiles ? 1u : 0u) != 0;            // takes everything the lists have left in the range
      const CKey khi = ck_uni(khi_n);
      const CKey kmid = ck_uni(kmid_n);
      if (!last) {
        // the next tile's row keys are wave 0's business alone: nobody else ever waits for these loads
        if (tid < 64) {
#pragma unroll
          for (int x = 0; x < CL_KPL; x++) { const u32 j = (u32)tid + 64u * x; skn[x] = ck_inf(); if (j < rt && s0 + rt + j < s_hi) skn[x] = skel[s0 + rt + j]; }
        }
        khi_n = q + 2 < ntiles ? skel[s0 + 2 * rt] : ck_inf();
        kmid_n = (CL_HALVES > 1 && s0 + rt + 56 < s_hi) ? skel[s0 + rt + 56] : ck_inf();
      }
      const ClEnt* const tab = ptab + (q % CL_NT) * CL_PT;
      const u32 mult = cl_uni(sh[4 + (q % CL_NT)]);
      // the wave's slices of the tile's slice groups (records below / from the tile's middle row key)
      // (an entry = the key and (list << 32 | count): k_cols_sparse builds the rows of the keys that reach the recurrence from them)
      gu64w* const ovk0 = (gu64w*)(uintptr_t)(C.ovkeys + (((((u64)(slot0 + q) * CL_HALVES) * nblk + blk) * CL_NW + wave) * CL_OVW) * EW);
      gu64w* const ovk1 = ovk0 + (u64)nblk * CL_NW * CL_OVW * EW;
      const u64 li_hi = (u64)li << 32;
      u32 wov = 0, wov1 = 0;                        // records of this wave that are not row keys (uniform), per slice group
      u32 xb0 = 0, xb1 = 0;                         // 0, or 1 + the first entry of the slice's extension in C.ovx (uniform)
      // a slice about to run over claims an extension (rare: a sample with several times the cohort's k-mers among the wave's lists)
      auto extend = [&](u32& xb) {
        u32 b = 0;
        if (lane == 0) b = atomicAdd(C.xcur, (u32)CL_XS);
        b = cl_uni(b);
        xb = (b + (u32)CL_XS <= C.xcap) ? b + 1u : 0xFFFFFFFFu;      // (pool exhausted: the task is handed back)
      };
      gu64w* const ovx = (gu64w*)(uintptr_t)C.ovx;

      for (;;) {   // rounds: one, unless a list has more than a window of records below the upper key
        // (three flags in turn: the one cleared here was last read two rounds ago, a barrier away)
        if (tid == 0) sh[1 + (rnd == 2 ? 0 : rnd + 1)] = 0;
        // which slots are consumed (this is where the window loads are waited for, all at once: the
        // refills issued further down then never stall a slot that is looked at after them)
        u32 consm = 0;
        if (last) {
#pragma unroll
          for (int u = 0; u < CL_U; u++) consm |= ((cur + ((r + CL_G * u - cur) & (CL_W - 1))) < end ? 1u : 0u) << u;
        } else {
#pragma unroll
          for (int u = 0; u < CL_U; u++) consm |= (ck_lt(cl_key(rec[u]), khi) ? 1u : 0u) << u;      // (an empty slot holds the largest key)
        }
        asm volatile("" : "+v"(consm));      // one bit mask in a vector register, not 16 lane masks in scalar registers
#if KMX_CL_KW == 1
        // (the next tile's row keys, requested at the tile's start, have landed with the window: from here on they are plain registers.
        //  Used where they were loaded, wave 0 met an s_waitcnt vmcnt(0) in front of its table build -- a wait for the window refills
        //  requested in between, with fifteen waves at the barrier behind it: 2.30 -> 2.28 ms)
#pragma unroll
        for (int x = 0; x < CL_KPL; x++) asm volatile("" : "+v"(skn[x]));
#endif
        CLPH(1);
#pragma unroll
        for (int g = 0; g < CL_U; g += 4) {
          __builtin_amdgcn_sched_barrier(0);
          u32 curg = cur; asm volatile("" : "+v"(curg));      // (re-derived per group: 16 slot indices kept live get spilled)
          ClEnt pe[4];
#pragma unroll
          for (int j = 0; j < 4; j++) pe[j] = ent_load(tab, cl_thash(cl_key(rec[g + j]), mult));
          u32 ovm = 0;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            // straight-line: masks and selects, no branches (a deposit that is none goes to a scratch word)
            const int u = g + j;
            const bool cons = (consm >> u) & 1u;
            const CKey k = cl_key(rec[u]);
            const u32 c = cl_cnt(rec[u]);
            const bool solid = cons && c >= smin;
            const bool hit = ent_hit(pe[j], k);
            tsum += solid ? c : 0u;
            tn += (cons && !solid) ? 1u : 0u;
            // (RESC: a row key's row has recurrence-min >= share-min solid records: its non-solid records are rescued, written like the others)
            const bool dep = RESC ? (cons && hit && (solid || rescue)) : (solid && hit);
            if (RESC) { const bool rs = cons && hit && !solid && rescue; rsum += rs ? c : 0u; rn += rs ? 1u : 0u; }
            bool big = false;      // NAR: a count that does not fit its byte -- to the 4-byte row, by the lane that holds it (below, in the slot's one branch)
            if (NAR) { img8[dep ? __umul24(pe[j].idx - 1, nbs) + lg : dummyb] = (u8)min(c, 255u); big = dep && c > 254u; }
            else if (MODE == 0) img[dep ? __umul24(pe[j].idx - 1, iw) + lg : dummy] = c;
            else if (dep) atomicOr(&img[__umul24(pe[j].idx - 1, iw) + (lg >> 5)], 1u << (lg & 31u));
            ovm |= (((RESC ? cons : solid) && !hit) ? 1u : 0u) << j;
            if (NAR) ovm |= (big ? 0x11u : 0u) << j;      // (bit 4 + j: the slot's record is such a count, not a record for the slices)
          }
          asm volatile("" : "+v"(tsum), "+v"(tn));      // summed up here, not at the end of the scan (with every count kept until then)
          if (RESC) asm volatile("" : "+v"(rsum), "+v"(rn));
          // solid records that are not row keys: appended to the wave's slice of the tile (positions from ballots)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            u64 bal = __ballot((ovm >> j) & 1u);
            if (bal) {      // (the same block without this branch around it: no difference, 2.29 ms either way)
              if (NAR) {
                const bool isbig = (ovm >> (4 + j)) & 1u;
                if (__builtin_expect(__ballot(isbig) != 0, 0)) {
                  if (isbig) { const u32 rw = ent_load(tab, cl_thash(cl_key(rec[g + j]), mult)).idx - 1; reinterpret_cast<u32*>(C.dense + (u64)(s0 + rw) * opitch)[li] = cl_cnt(rec[g + j]); rowbig[rw] = 1; }      // (its row: looked up again -- rare)
                  ovm &= isbig ? ~(1u << j) : ~0u;
                  bal = __ballot((ovm >> j) & 1u);
                }
              }
              const CKey kk = cl_key(rec[g + j]);
              const u64 pay = li_hi | cl_cnt(rec[g + j]) | ((RESC && cl_cnt(rec[g + j]) < smin) ? CL_NONSOLID : 0ULL);
#if KMX_CL_KW == 1
#define CL_PUT(o, pos) do { (o)[2 * (pos)] = kk; (o)[2 * (pos) + 1] = pay; } while (0)
#else
#define CL_PUT(o, pos) do { (o)[3 * (pos)] = kk.lo; (o)[3 * (pos) + 1] = kk.hi; (o)[3 * (pos) + 2] = pay; } while (0)
#endif
              if (CL_HALVES == 1) {
                if (EXT && __builtin_expect(wov + (u32)__popcll(bal) > (u32)CL_OVW && xb0 == 0, 0)) extend(xb0);
                if ((ovm >> j) & 1u) {
                  const u32 pos = wov + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
                  if (pos < (u32)CL_OVW) CL_PUT(ovk0, pos);
                  else if (EXT && xb0 - 1u < 0xFFFFFFFEu && pos - (u32)CL_OVW < (u32)CL_XS) { gu64w* const o = ovx + (u64)(xb0 - 1u) * EW; const u32 px = pos - (u32)CL_OVW; CL_PUT(o, px); }
                }
                wov += (u32)__popcll(bal);
              } else {
                const u64 hi = __ballot(((ovm >> j) & 1u) && !ck_lt(kk, kmid)), lo = bal & ~hi;
                if (EXT && __builtin_expect(wov + (u32)__popcll(lo) > (u32)CL_OVW && xb0 == 0, 0)) extend(xb0);
                if (EXT && __builtin_expect(wov1 + (u32)__popcll(hi) > (u32)CL_OVW && xb1 == 0, 0)) extend(xb1);
                if ((ovm >> j) & 1u) {
                  const bool up = !ck_lt(kk, kmid);
                  const u64 m = up ? hi : lo;
                  const u32 pos = (up ? wov1 : wov) + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
                  if (pos < (u32)CL_OVW) { gu64w* const o = up ? ovk1 : ovk0; CL_PUT(o, pos); }
                  else if (EXT) {
                    const u32 xb = up ? xb1 : xb0, px = pos - (u32)CL_OVW;
                    if (xb - 1u < 0xFFFFFFFEu && px < (u32)CL_XS) { gu64w* const o = ovx + (u64)(xb - 1u) * EW; CL_PUT(o, px); }
                  }
                }
                wov += (u32)__popcll(lo); wov1 += (u32)__popcll(hi);
              }
            }
          }
          // refill in place: the consumed records are a prefix of the window, so a consumed slot's next record is
          // the one 64 positions further (one past the list's end: the sentinel).  A slot that was not consumed is left
          // alone: loading it again would fetch the window's last line twice (it is evicted from L2 by the next tile).
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int u = g + j;
            if ((consm >> u) & 1u) {
              const u32 ix = curg + ((r + CL_G * u - curg) & (CL_W - 1)) + (u32)CL_W;
              gu32* const src = ix < end ? base + (u64)ix * RD : sentinel;
              rec[u] = cl_load(src);
            }
          }
        }
        u32 c = __popc(consm);
        c += (u32)__builtin_amdgcn_update_dpp(0, (int)c, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
        c += (u32)__builtin_amdgcn_update_dpp(0, (int)c, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
        if (CL_G == 8) c += (u32)__builtin_amdgcn_update_dpp(0, (int)c, 0x141, 0xF, 0xF, true);   // row_half_mirror: the other quad of my 8
        cur += c;
        if (c == (u32)CL_W && cur < end) sh[1 + rnd] = 1;
        CLPH(2);
        cl_barrier();
        CLPH(3);
        const u32 more = cl_uni(sh[1 + rnd]);
        rnd = rnd == 2 ? 0 : rnd + 1;
        if (!more) break;
      }
      if (lane == 0) {
        const u64 s0i = (((u64)(slot0 + q) * CL_HALVES) * nblk + blk) * CL_NW + wave, s1i = (((u64)(slot0 + q) * CL_HALVES + 1) * nblk + blk) * CL_NW + wave;
        // (a slice's count and its extension's place side by side: one load for k_cols_sparse)
        reinterpret_cast<uint2*>(C.ovcnt)[s0i] = make_uint2(wov, xb0 == 0xFFFFFFFFu ? 0u : xb0);
        if (CL_HALVES > 1) reinterpret_cast<uint2*>(C.ovcnt)[s1i] = make_uint2(wov1, xb1 == 0xFFFFFFFFu ? 0u : xb1);
        const bool over0 = wov > (u32)CL_OVW && (xb0 - 1u >= 0xFFFFFFFEu || wov > (u32)(CL_OVW + CL_XS));
        const bool over1 = wov1 > (u32)CL_OVW && (xb1 - 1u >= 0xFFFFFFFEu || wov1 > (u32)(CL_OVW + CL_XS));
        if (over0 || over1) {
          if (lane == 0) atomicOr(&T.ctrl[2], (u64)ERR_SLICES); failed = true; atomicAdd(&kmx_cols_dbg[1], 1u); atomicMax(&kmx_cols_dbg[5], max(wov, wov1)); if (last) atomicAdd(&kmx_cols_dbg[6], 1u); if (q == 0) atomicAdd(&kmx_cols_dbg[7], 1u); }
      }

      // ---- tile out: wave 0 turns the row table over, the others stream the image out (and leave it zeroed) ----
      if (wave == 0) {
        ClEnt* const old = ptab + (q % CL_NT) * CL_PT;
#pragma unroll
        for (int x = 0; x < CL_KPL; x++) if ((u32)lane + 64u * x < rte) old[myslot[x]].idx = 0;        // (keys may stay: an entry without a row is never a hit)
        if (!last) {
          const u32 sn = s0 + rt;
          bool have[CL_KPL];
#pragma unroll
          for (int x = 0; x < CL_KPL; x++) {
            const u32 j = (u32)lane + 64u * x;
            have[x] = j < rt && sn + j < s_hi;
            if (have[x] && blk == 0 && !ord) ck_store(reinterpret_cast<u32*>(T.out + (u64)(sn + j) * row_bytes), skn[x]);
          }
          const u32 m2 = cl_build(ptab + ((q + 1) % CL_NT) * CL_PT, skn, have, (u32)lane, myslot);
          if (lane == 0) { sh[4 + ((q + 1) % CL_NT)] = m2; if (m2 == 0) { failed = true; atomicAdd(&kmx_cols_dbg[0], 1u); } }
        }
      } else {
        if (NAR) {
          // a row's slice of the block: nbl count bytes, 16 per lane, eight lanes a row, eight rows a wave and step
          const u32 lr = (u32)lane >> 3, lc = (u32)lane & 7u, nch = (nbl + 15u) >> 4;
          for (u32 j = (wave - 1) * 8u + lr; j < rte; j += (CL_NW - 1) * 8u) {
            u8* const nrow = nar + (u64)(s0 + j) * npitch;
            if (lc < nch) {
              uint4* const src = reinterpret_cast<uint4*>(img8 + j * nbs + 16u * lc);
              const uint4 v = *src; *src = make_uint4(0, 0, 0, 0);
              *reinterpret_cast<uint4*>(nrow + col0 + 16u * lc) = v;
            }
            if (lc == 0) { const u8 f = rowbig[j]; rowbig[j] = 0; nrow[npitch - 16u + blk] = f; }      // (the row's flag byte of this block: some count of it is in the 4-byte row)
          }
        } else
        if (MODE == 0) {
          u8* const out0 = obase + (u64)s0 * opitch + 4ull * col0;
          const bool wide = ((opitch | (4u * col0) | (4u * nbs)) & 7u) == 0;
          for (u32 j = wave - 1; j < rte; j += CL_NW - 1) {
            u32* const src = img + j * nbs;
            u8* const dst = out0 + (u64)j * opitch;
            if (wide) {
              const u32 n2 = nbl >> 1;
              for (u32 t0 = 0; t0 < n2; t0 += 256) {
                u64 w[4];
#pragma unroll
                for (int x = 0; x < 4; x++) { const u32 t = t0 + 64 * x + lane; w[x] = 0; if (t < n2) { w[x] = reinterpret_cast<u64*>(src)[t]; reinterpret_cast<u64*>(src)[t] = 0; } }
#pragma unroll
                for (int x = 0; x < 4; x++) { const u32 t = t0 + 64 * x + lane; if (t < n2) reinterpret_cast<u64*>(dst)[t] = w[x]; }
              }
              if ((nbl & 1u) && lane == 0) { reinterpret_cast<u32*>(dst)[nbl - 1] = src[nbl - 1]; src[nbl - 1] = 0; }
            } else {
              for (u32 t = lane; t < nbl; t += 64) { const u32 w = src[t]; src[t] = 0; reinterpret_cast<u32*>(dst)[t] = w; }
            }
          }
        } else {
          // a row's slice is (lists of the block) / 8 bytes (the block starts at a multiple of 8 lists): a byte per lane
          u8* const out0 = obase + (u64)s0 * opitch + (col0 >> 3);
          const u32 nby = (nbl + 7) >> 3;
          if (ORD && nbs == 128u && (opitch & 15u) == 0 && (col0 >> 3) + 16u <= opitch) {      // (blocks of exactly 128 lists: every block's slice of a row is its own 16 bytes)
            // the side store's rows are 16-byte aligned and a block's slice of a row is 16 bytes: a LANE per row, one 16-byte load from
            // the image and one 16-byte store (round 5; a wave per row and a byte per lane before: 56 store instructions a tile
            // where one does; the bits behind the last list are zeros in the row's padding)
            for (u32 j = (wave - 1) * 64u + (u32)lane; j < rte; j += (CL_NW - 1) * 64u) {
              uint4* const src = reinterpret_cast<uint4*>(img + j * 4u);
              const uint4 v = *src; *src = make_uint4(0, 0, 0, 0);
              *reinterpret_cast<uint4*>(out0 + (u64)j * opitch) = v;
            }
          } else
          for (u32 j = wave - 1; j < rte; j += CL_NW - 1) {
            u32* const src = img + j * iw;
            u8* const dst = out0 + (u64)j * opitch;
            for (u32 t = lane; t < nby; t += 64) dst[t] = (u8)(src[t >> 2] >> ((t & 3u) * 8));
            for (u32 t = lane; t < iw; t += 64) src[t] = 0;      // (same wave, behind the reads)
          }
        }
      }
      CLPH(4);
      cl_barrier();
      CLPH(5);
    }

    // ---- item done: statistics of my list (the 4 lanes of a list add up) ----
    {
      u32 tlo = (u32)tsum, thi = (u32)(tsum >> 32);
      u64 ts = tsum;
#pragma unroll
      for (int off = 1; off < CL_G; off <<= 1) {
        const u32 olo = __shfl_xor(tlo, off), ohi = __shfl_xor(thi, off);
        ts += (u64)olo | ((u64)ohi << 32);
        tlo = (u32)ts; thi = (u32)(ts >> 32);
        tn += __shfl_xor(tn, off);
      }
      if (on && r == 0) {
        if (tn) atomicAdd(&T.stats[0 * (u64)N + li], (u64)tn);
        if (ts) atomicAdd(&T.stats[4 * (u64)N + li], ts);
      }
      if (RESC) {
        u32 rlo = (u32)rsum, rhi = (u32)(rsum >> 32);
        u64 rs = rsum;
#pragma unroll
        for (int off = 1; off < CL_G; off <<= 1) {
          const u32 olo = __shfl_xor(rlo, off), ohi = __shfl_xor(rhi, off);
          rs += (u64)olo | ((u64)ohi << 32);
          rlo = (u32)rs; rhi = (u32)(rs >> 32);
          rn += __shfl_xor(rn, off);
        }
        if (on && r == 0 && rn) { atomicAdd(&T.stats[1 * (u64)N + li], (u64)rn); atomicAdd(&T.stats[5 * (u64)N + li], rs); }
      }
    }
    if (failed) atomicOr(&T.ctrl[2], (u64)ERR_FALLBACK);
    __syncthreads();
  }
}

// ---- after the merge: the rows of the keys OUTSIDE the row keys.  A cohort's samples share most of their private k-mers with
//      nobody, some with one or two other samples: those keys reach a small recurrence-min without being in the few lists the row
//      keys were taken from (with recurrence-min 1 -- kmtricks' default -- every key set aside is a row).  One workgroup per
//      (task, range), a slice group (the records the column blocks set aside for half a tile: the keys between two row keys 56
//      rows apart) at a time:
//        1. every entry sets its key's bit of a 256-Kbit LDS map; an entry that finds the bit set marks the key in a second, small
//           map: the keys marked there occur at least twice (plus a few chance pairs) -- the CANDIDATES; for a recurrence-min of 1
//           every entry is one;
//        2. the candidates are gathered in LDS and sorted by key (bitonic); a run of >= recurrence-min equal keys is a row;
//        3. the group's rows are claimed from the task's arena BEHIND the row keys' rows with one atomic; a wave per row writes the
//           key, zeroes the counts and drops the run's counts in (PA: sets its bits);
//        4. a directory entry per (group, pass) says where they are: the rows of a task are the row keys' rows (row r = row key r)
//           and these, each list ascending; k_cols_gather interleaves them when the body is asked for.
//      Groups with many entries are done in 2..8 passes split by hash bits (equal keys meet in the same pass). ----
#ifndef KMX_CK_TPB0
#define KMX_CK_TPB0 256
#endif
#ifndef KMX_CK_TPB1
#define KMX_CK_TPB1 512
#endif
#ifndef KMX_CK_Z
#define KMX_CK_Z 16
#endif
#ifndef KMX_CK_OCC1
#define KMX_CK_OCC1 4
#endif
// threads per workgroup of k_cols_sparse, by the rows' kind.  Count rows (4 N bytes each): 256, two workgroups per CU -- a group costs its
// workgroup four dependent round trips (ticket, descriptor, slices, look-back) and a sort before it writes a byte, and the second workgroup's
// stores fill that time: configs[2] in file order 2.68 -> 2.49 ms (round 5; with 128-VGPR workgroups of 512 it was 3.3).  PA rows: 512 --
// their time is the candidates' sort, which half the threads run twice as long (configs[4]: 2.51 -> 2.78 ms with 256).
template <int MODE> __host__ __device__ constexpr int ck_tpb() { return MODE == 0 ? KMX_CK_TPB0 : KMX_CK_TPB1; }
constexpr int CK_Z = KMX_CK_Z;           // workgroups sharing the slice groups of a range
#ifndef KMX_CK_SPEC2
#define KMX_CK_SPEC2 (KW == 1 ? 7 : 4)      // (64-bit keys: 14 of a slice's usual ~14 entries in the first round trip -- with 5 the tail's second one cost 0.05 ms of configs[2]; 256 VGPRs, 32 bytes of scratch)
#endif
constexpr int CK_SPEC4 = KW == 1 ? 5 : 4;              // entries per thread requested together with the slice's count: four threads a slice ...
constexpr int CK_SPEC2 = KMX_CK_SPEC2;                 // ... two threads a slice (workgroups of 256)
constexpr int CK_CAND = 2048;            // candidates per pass (32 or 48 KB of LDS with their payloads)
constexpr int CK_B2 = 1 << 15;           // bits of the candidate map
constexpr int CK_NPASS = 8;              // directory entries per group
constexpr int CK_UNI = CK_BITS / 8 + CK_B2 / 8;      // bytes of the LDS block that is the key maps while candidates are chosen and the rows' staging once they are sorted

struct SpDir { u32 base, n, dense_first, dense_n; };      // rows [base, base + n) of the arena: a pass's rows; dense_* filled in entry 0 of a group

__device__ __forceinline__ bool ck_less(CKey ka, u64 pa, CKey kb, u64 pb) { return !ck_eq(ka, kb) ? ck_lt(ka, kb) : pa < pb; }

// ---- the candidates' sort: a bitonic network over P = 128 .. CK_CAND entries in LDS.  Steps between entries less than 128
//      apart stay inside a wave -- a wave holds a chunk of 128 entries in registers (two per lane) and exchanges them with
//      lane shuffles, no workgroup barrier -- so a sort of 1024 entries meets 10 barriers, not 55. ----
__device__ __forceinline__ CKey ck_shfl_xor(CKey k, int m)
{
#if KMX_CL_KW == 1
  return (u64)__shfl_xor((unsigned long long)k, m);
#else
  CKey r; r.lo = (u64)__shfl_xor((unsigned long long)k.lo, m); r.hi = (u64)__shfl_xor((unsigned long long)k.hi, m); return r;
#endif
}
// steps j = jtop .. 1 of stage k2 on the chunk [base, base + 128) (jtop <= 64)
__device__ __forceinline__ void ck_chunk_steps(CKey (&k)[2], u64 (&p)[2], u32 base, u32 lane, u32 k2, u32 jtop)
{
  if (jtop >= 64u) {
    const bool asc = ((base + lane) & k2) == 0;      // (k2 >= 128 here: both of my entries sort the same way)
    if (ck_less(k[1], p[1], k[0], p[0]) == asc) { const CKey tk = k[0]; k[0] = k[1]; k[1] = tk; const u64 tp = p[0]; p[0] = p[1]; p[1] = tp; }
    jtop = 32;
  }
  for (u32 j = jtop; j > 0; j >>= 1) {
#pragma unroll
    for (int x = 0; x < 2; x++) {
      const CKey ok = ck_shfl_xor(k[x], (int)j); const u64 op = (u64)__shfl_xor((unsigned long long)p[x], (int)j);
      const bool asc = ((base + 64u * x + lane) & k2) == 0;
      const bool keep_min = ((lane & j) == 0) == asc;
      const bool other_less = ck_less(ok, op, k[x], p[x]);
      if (other_less == keep_min) { k[x] = ok; p[x] = op; }
    }
  }
}
template <int CK_TPB>
__device__ __forceinline__ void ck_sort_block(CKey* ck, u64* cp, u32 P, u32 tid)      // P: a power of two >= 128, pads = ck_inf()
{
  const u32 lane = tid & 63u, wave = tid >> 6;
  for (u32 base = wave * 128u; base < P; base += (CK_TPB / 64) * 128u) {
    CKey k[2] = {ck[base + lane], ck[base + 64u + lane]}; u64 p[2] = {cp[base + lane], cp[base + 64u + lane]};
    for (u32 k2 = 2; k2 <= 128u; k2 <<= 1) ck_chunk_steps(k, p, base, lane, k2, k2 >> 1);
    ck[base + lane] = k[0]; ck[base + 64u + lane] = k[1]; cp[base + lane] = p[0]; cp[base + 64u + lane] = p[1];
  }
  __syncthreads();
  for (u32 k2 = 256; k2 <= P; k2 <<= 1) {
    for (u32 j = k2 >> 1; j >= 128u; j >>= 1) {
      for (u32 t = tid; t < P / 2; t += CK_TPB) {
        const u32 a = ((t & ~(j - 1)) << 1) | (t & (j - 1)), b = a | j;
        const CKey ka = ck[a], kb = ck[b]; const u64 pa = cp[a], pb = cp[b];
        const bool up = (a & k2) == 0;
        if (ck_less(kb, pb, ka, pa) == up) { ck[a] = kb; ck[b] = ka; cp[a] = pb; cp[b] = pa; }
      }
      __syncthreads();
    }
    for (u32 base = wave * 128u; base < P; base += (CK_TPB / 64) * 128u) {
      CKey k[2] = {ck[base + lane], ck[base + 64u + lane]}; u64 p[2] = {cp[base + lane], cp[base + 64u + lane]};
      ck_chunk_steps(k, p, base, lane, k2, 64u);
      ck[base + lane] = k[0]; ck[base + 64u + lane] = k[1]; cp[base + lane] = p[0]; cp[base + 64u + lane] = p[1];
    }
    __syncthreads();
  }
}

// a kept run in runs[]: its first entry (11 bits: < CK_CAND) | its entries << 11 (13 bits: one per list, <= 4096) | (ORD) how many of the
// pass's row keys lie below its key << 24 (<= 56)
static_assert(CK_CAND <= 2048, "11 bits for a run's first entry");
#define CK_RUN_I0(v) ((v) & 0x7FFu)
#define CK_RUN_LEN(v) (((v) >> 11) & 0x1FFFu)
#define CK_RUN_ND(v) ((v) >> 24)
constexpr u32 CK_OVF = 0xFFFFFFFFu;      // a pass found more candidates than the LDS holds

// ---- ORD: the rows in front of a slice group.  A task's groups are numbered in key order (k_cols_prep) and handed out by ticket in
//      that order; chain[g] = status << 62 | rows.  Wave 0 of the workgroup that holds group g publishes the group's own rows
//      (status 1) as soon as it knows them -- before it waits for anything --, adds up its predecessors' entries backwards until it
//      meets one that carries a whole prefix (status 2), and publishes its own prefix.  The holder of the lowest unfinished ticket
//      never waits: no deadlock whatever the number of resident workgroups. ----
// A look-back that meets an unpublished entry sleeps and looks again -- at most CK_LB_SPINS times (seconds; a whole launch is
// milliseconds): then it gives up (returns ~0: the caller hands the task back to the general kernel and the chain goes on with a prefix
// of 0), so that whatever keeps an entry from being published costs a task its fast path, not the launch its end.
constexpr u32 CK_LB_SPINS = 1u << 20;
__device__ __forceinline__ u64 ck_lookback(u64* chain, const u32 g, const u64 mine, const u32 lane, u64* err /* the task's error word: ERR_FALLBACK is raised there BEFORE a give-up publishes its empty prefix */)
{
  constexpr u64 VAL = (1ULL << 62) - 1ULL;
  u32 spins = 0;
  if (g == 0) { if (lane == 0) __hip_atomic_store(&chain[0], (2ULL << 62) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return 0; }
  if (lane == 0) __hip_atomic_store(&chain[g], (1ULL << 62) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  u64 excl = 0;
  u32 i = g;      // the entries below i are still to be added
  for (;;) {
    const u64 v = lane < i ? __hip_atomic_load(&chain[i - 1 - lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ULL << 62);      // (in front of group 0: a prefix of 0)
    const u32 st = (u32)(v >> 62);
    const u64 inv = __ballot(st == 0), pfx = __ballot(st == 2);
    const u32 lim = inv ? (u32)__builtin_ctzll(inv) : 64u;      // the lanes below lim hold published entries
    const u64 pm = lim == 64u ? pfx : (pfx & ((1ULL << lim) - 1ULL));
    const u32 take = pm ? (u32)__builtin_ctzll(pm) + 1u : lim;
    u64 s = lane < take ? (v & VAL) : 0ULL;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += shfl_xor_u64(s, off);
    excl += s;
    if (pm) break;
    i -= take;
    if (take == 0) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > CK_LB_SPINS) {
        if (lane == 0) {
          atomicOr(err, (u64)ERR_FALLBACK);      // (first: a later group that reads the empty prefix below finds the task handed back already)
          __threadfence();
          __hip_atomic_store(&chain[g], 2ULL << 62, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicAdd(&kmx_cols_dbg[8], 1u); kmx_cols_dbg[10] = g; kmx_cols_dbg[11] = i;
        }
        return ~0ULL;
      }
    }
  }
  if (lane == 0) __hip_atomic_store(&chain[g], (2ULL << 62) | (excl + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return excl;
}

// ORD (rows at their final place): the kernel is persistent, a slice group per ticket; a group's passes are cut by KEY RANGE (at the
// group's row keys), so that a pass's rows -- the row keys' rows of its stretch, copied from C.dense, and the rows of the keys set
// aside there, interleaved by key -- are one contiguous run of the body; the group's place comes from the look-back above.  A
// group of several passes counts its rows first (every pass sorted once without writing), publishes, and sorts again to write.
template <int MODE, bool RESC, bool ORD>
__global__ __launch_bounds__(ck_tpb<MODE>(), MODE == 1 ? KMX_CK_OCC1 : ORD ? 2 : 3)      // (waves per SIMD.  PA rows: <= 128 VGPRs, two workgroups of 512 per CU; count rows: two workgroups of 256, the ORD build with 234 VGPRs -- it keeps four row keys' rows per wave in flight)
void k_cols_sparse(const TaskDev* __restrict__ tasks, const ColsDev* __restrict__ cols, const uint2* __restrict__ items, u32 n_items, u32 n_tasks, u32* tkt)
{
  constexpr int CK_TPB = ck_tpb<MODE>();
  constexpr u32 CK_TS = CK_TPB >= 512 ? 4 : 2;         // threads per slice of a group when every slice has its threads at once (<= 128 slices)
  constexpr int CK_SPEC = CK_TS == 4 ? CK_SPEC4 : CK_SPEC2;
  constexpr int CK_PTM = CK_CAND / CK_TPB;             // sorted candidates per thread, at most
  constexpr int CK_STAGE = CK_UNI / (CK_TPB / 64);     // a wave's part of the staging block (512 threads: 2560 bytes: 32 PA rows of 500 lists and a 128-bit key)
  __shared__ __attribute__((aligned(16))) u32 uni[CK_UNI / 4];      // the key maps | the row keys and interval counters (recurrence-min 1) | the rows' staging
  u32* const bits = uni;
  u32* const bits2 = uni + CK_BITS / 32;
  __shared__ CKey ck[CK_CAND];           // candidate keys
  __shared__ u64 cp[CK_CAND];            // ... and their (list << 32 | count)
  __shared__ u32 runs[CK_CAND];          // kept runs (CK_RUN_*)
  __shared__ u32 wsum[CK_TPB / 64];
  __shared__ u32 total, ncand, rowbase, sover;
  __shared__ CKey dkeys[ORD ? 64 : 1];   // ORD: the group's row keys
  __shared__ u32 dpos[ORD ? 64 : 1];     // ... and where their rows go among the pass's rows
  __shared__ u32 s_tk;
  __shared__ u64 s_base;
  u32* const parow = uni;                // PA, rows too long for the staging below: a row per lane group is assembled here (<= 4096 lists + key)
  static_assert((CK_TPB / 16) * 136 * 4 <= CK_UNI, "the lane groups' rows fit the block");
  auto ent_key = [](const u64* kp, u32 e) -> CKey {
#if KMX_CL_KW == 1
    return kp[EW * e];
#else
    CKey k; k.lo = kp[EW * e]; k.hi = kp[EW * e + 1]; return k;
#endif
  };
#ifdef KMX_PHASE_PROF
  long long spt[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long spc = clock64();
#define SPPH(i) do { const long long n_ = clock64(); spt[i] += n_ - spc; spc = n_; } while (0)
#else
#define SPPH(i) do {} while (0)
#endif
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto pl_list = [](u64 pl) -> u32 { return (u32)(pl >> 32) & 0x7FFFFFFFu; };
  auto pl_solid = [](u64 pl) -> bool { return !RESC || !(pl & CL_NONSOLID); };
  if (tid == 0) { total = 0; sover = 0; }
  __syncthreads();

  // ---- one slice group (the records the column blocks set aside for half a tile); false: the task is handed back ----
  auto group = [&](const TaskDev& T, const ColsDev& C, const u32 range, const u32 q, const u32 s_lo, const u32 s_hi, const u32 gord) -> bool {
    // thr: solid records that make a key a row (RESC: 0 is taken literally -- a key only non-solid records hold is a row of zeros);
    // share: solid records from which a key's non-solid records are rescued (0: never)
    const u32 thr = RESC ? T.rec_min : max(1u, T.rec_min), share = RESC ? T.share_min : 0u, rt = C.rt, nsl = C.nblk * CL_NW, row_bytes = T.row_bytes;
    const u32 slot0 = (s_lo / rt + range) * CL_HALVES;
    SpDir* const dir = reinterpret_cast<SpDir*>(C.spdir);
    const bool single = nsl <= (u32)CK_TPB / CK_TS;
    const u64 sbase = (u64)(slot0 + q) * nsl;
    const u32 gid = slot0 + q;
    auto hand_back = [&](int why) { if (tid == 0) { atomicOr(&T.ctrl[2], (u64)ERR_FALLBACK); atomicAdd(&kmx_cols_dbg[why], 1u); } };
    // the group's row keys: rows [d0, d0 + dn) (a tile's first half: 56 rows, second: the rest)
    const u32 tq = q / CL_HALVES, hq = q % CL_HALVES;
    const u32 t0 = s_lo + tq * rt, te = min(s_hi, t0 + rt);
    u32 d0 = t0, dn = te - t0;
    if (CL_HALVES > 1) { const u32 mid = min(te, t0 + 56u); d0 = hq ? mid : t0; dn = hq ? te - mid : mid - t0; }
    if (s_hi == s_lo) { d0 = s_lo; dn = 0; }
    if (ORD && tid < dn) dkeys[tid] = reinterpret_cast<const CKey*>(C.skel)[d0 + tid];
    // four threads per (block, wave) slice; when every slice has its four threads at once (<= 1024 lists) a slice's count and its
    // first 20 entries (5 per thread; the usual slice holds ~14; the slice's memory is there whatever the count) are requested
    // together and kept in registers for both walks over the entries: one memory round trip per group
    u32 n0 = 0; CKey kk0[CK_SPEC]; u64 pp0[CK_SPEC];
    const u64* kp0 = C.ovkeys; const u64* kpx0 = C.ovx;
    if (single) {
      const u32 sl = tid / CK_TS, sub = tid % CK_TS;
      const bool ok = sl < nsl;
      const uint2 ce = ok ? reinterpret_cast<const uint2*>(C.ovcnt)[sbase + sl] : make_uint2(0u, 0u);
      const u32 nraw = ce.x, xb = ce.y;                                 // the slice's entries, and its extension if it has one
      kpx0 = C.ovx + (u64)(xb ? xb - 1u : 0u) * EW;
      const u32 lim0 = xb ? (u32)(CL_OVW + CL_XS) : (u32)CL_OVW;
      kp0 = C.ovkeys + (sbase + (ok ? sl : 0u)) * CL_OVW * EW;
#pragma unroll
      for (int x = 0; x < CK_SPEC; x++) { kk0[x] = ent_key(kp0, sub + CK_TS * x); pp0[x] = kp0[EW * (sub + CK_TS * x) + KW]; }
      n0 = min(nraw, lim0);
      if (nraw > lim0) sover = 1;
      if (sub == 0 && nraw) atomicAdd(&total, nraw);
    } else {
      for (u32 sl = tid; sl < nsl; sl += CK_TPB) {
        const uint2 ce = reinterpret_cast<const uint2*>(C.ovcnt)[sbase + sl];
        const u32 nraw = ce.x;
        if (nraw > (ce.y ? (u32)(CL_OVW + CL_XS) : (u32)CL_OVW)) sover = 1;
        if (nraw) atomicAdd(&total, nraw);
      }
    }
    __syncthreads();
    const u32 tot = total;
    const bool over = sover != 0;
    __syncthreads();
    SPPH(0);
    if (tid == 0) { total = 0; sover = 0; if (!ORD) { dir[(u64)gid * CK_NPASS].dense_first = d0; dir[(u64)gid * CK_NPASS].dense_n = dn; } }
    bool failed = false;
    if (over) { hand_back(1); if (!ORD) return false; failed = true; }
    if (!ORD && tot == 0) return true;
    // passes: ~1400 entries each when every entry is a candidate, ~3000 when only the keys seen twice are
#ifndef KMX_CK_PER1
#define KMX_CK_PER1 1400
#endif
    const u32 per = (thr <= 1 ? (u32)KMX_CK_PER1 : 3000u) * (u32)CK_CAND / 2048u;
    u32 npass = 1;
    if (!(MODE == 1 && thr <= 1 && tot <= (u32)CK_CAND)) while (npass < (u32)CK_NPASS && tot > per * npass) npass <<= 1;      // (PA rows at recurrence-min 1 -- every entry a candidate, placed by the sample sort -- when all of them fit: one pass, no margin needed)
    if (!ORD && tot > per * npass * 2) { hand_back(3); return false; }
    if (ORD) {
      if (thr <= 1 && tot <= (u32)CK_CAND) npass = 1;      // (every entry a candidate, and they fit: a pass too full would only be cut finer and sorted again)
      while (npass > 1 && npass > dn) npass >>= 1;         // (cut at the group's row keys: a pass holds at least one)
    }
    // each of my entries of the pass in hand through f(key, payload): four threads per slice
    auto each = [&](auto&& in_pass, auto&& f) {
      if (single) {
        const u32 sub = tid % CK_TS;
#pragma unroll
        for (int x = 0; x < CK_SPEC; x++)
          if (sub + CK_TS * x < n0 && in_pass(kk0[x])) f(kk0[x], pp0[x]);
        for (u32 e = sub + CK_TS * CK_SPEC; e < n0; e += CK_TS) {
          const u64* const kp = e < (u32)CL_OVW ? kp0 : kpx0; const u32 ee = e < (u32)CL_OVW ? e : e - (u32)CL_OVW;      // (the slice, then its extension)
          const CKey k = ent_key(kp, ee);
          if (!in_pass(k)) continue;
          f(k, kp[EW * ee + KW]);
        }
        return;
      }
      for (u32 sl0 = 0; sl0 < nsl; sl0 += CK_TPB / 4) {
        const u32 sl = sl0 + (tid >> 2), sub = tid & 3u;
        if (sl >= nsl) continue;
        const uint2 ce = reinterpret_cast<const uint2*>(C.ovcnt)[sbase + sl];
        const u32 xb = ce.y;
        const u32 n = min(ce.x, xb ? (u32)(CL_OVW + CL_XS) : (u32)CL_OVW);
        const u64* const kpb = C.ovkeys + (sbase + sl) * CL_OVW * EW;
        const u64* const kpx = C.ovx + (u64)(xb ? xb - 1u : 0u) * EW;
        for (u32 e = sub; e < n; e += 4) {
          const u64* const kp = e < (u32)CL_OVW ? kpb : kpx; const u32 ee = e < (u32)CL_OVW ? e : e - (u32)CL_OVW;
          const CKey k = ent_key(kp, ee);
          if (!in_pass(k)) continue;
          f(k, kp[EW * ee + KW]);
        }
      }
    };
    // a candidate's place: one LDS atomic per wave and call site, not one per entry (they all hit the same word)
    auto push = [&](CKey k, u64 pl) {
      const u64 act = __ballot(1);
      const u32 ldr = (u32)__builtin_ctzll(act);
      u32 b0 = 0;
      if (lane == ldr) b0 = atomicAdd(&ncand, (u32)__popcll(act));
      b0 = (u32)__shfl((int)b0, (int)ldr);
      const u32 ps = b0 + (u32)__popcll(act & ((1ULL << lane) - 1ULL));
      if (ps < (u32)CK_CAND) { ck[ps] = k; cp[ps] = pl; }
    };

    // ---- a pass, first half: its candidates gathered and sorted, the kept runs in runs[] -> how many (CK_OVF: more candidates than
    //      fit).  final: the rescue's statistics are added and the entries marked for the rows (a counting sweep must do neither).
    //      ORD: the pass holds the keys between row keys dlo and dlo + dnp of the group. ----
    auto sort_pass = [&](const u32 pass, const bool final, const u32 dlo, const u32 dnp) -> u32 {
      constexpr bool BY_INTERVAL = MODE == 1;      // (count rows -- 4 N bytes each -- are bound by their stores: +-0 there, 20 more registers)
      CKey plo = ck_inf(), phi = ck_inf(); bool has_lo = false, has_hi = false;
      if (ORD && npass > 1) { has_lo = pass > 0; has_hi = pass + 1 < npass; if (has_lo) plo = dkeys[dlo]; if (has_hi) phi = dkeys[dlo + dnp]; }
      auto in_pass = [&](CKey k) -> bool {
        if (npass <= 1) return true;
        if (ORD) return !(has_lo && ck_lt(k, plo)) && !(has_hi && !ck_lt(k, phi));
        return ((cl_mix(k) >> 24) & (npass - 1)) == pass;      // (by hash bits: equal keys meet in the same pass)
      };
      if (thr > 1) {
        for (u32 t = tid; t < (u32)CK_BITS / 128; t += CK_TPB) reinterpret_cast<uint4*>(bits)[t] = make_uint4(0, 0, 0, 0);
        for (u32 t = tid; t < (u32)CK_B2 / 128; t += CK_TPB) reinterpret_cast<uint4*>(bits2)[t] = make_uint4(0, 0, 0, 0);
      } else {
        for (u32 t = tid; t < 512u; t += CK_TPB) bits[t] = 0;      // (the interval counters below live in the key map's room)
      }
      if (tid == 0) ncand = 0;
      __syncthreads();
      SPPH(1);
      bool sorted = false;
      u32 nc1 = 0;
      if (thr <= 1 && !BY_INTERVAL) {
        each(in_pass, push);
        __syncthreads();
      }
      if (thr <= 1 && BY_INTERVAL) {
        // recurrence-min 1: every entry is a row, the sort is all there is to do -- a SAMPLE SORT inside the workgroup.  Every
        // (n / 256)-th entry, as gathered (slice order: no order in the keys), is a sample; the <= 256 samples are sorted (the
        // network below, a few barriers at that size), an entry finds its interval by binary search among them, the intervals are
        // laid out one after the other (count, scan, place) and an entry's place inside its interval is the number of the interval's
        // entries below it (full compare: ~8 of them).  Balanced whatever the keys look like -- the k-mers of a minimizer partition
        // crowd a few prefixes: intervals cut by the group's row keys (round 2) or by equal steps of the key's top word came out so
        // uneven that counting "entries below me" was half of the kernel (phase profile of configs[4]: 48 % of its cycles).
        constexpr u32 NS = 256, NIV = 320;                                   // samples; interval counters (5 per lane of wave 0)
        u32* const ioff = bits;                                               // [NIV + 1] entries per interval, then where each interval starts
        CKey* const sk = reinterpret_cast<CKey*>(bits + 1024);                // [NS] the samples
        u64* const sp = reinterpret_cast<u64*>(bits + 1024 + NS * (sizeof(CKey) / 4));      // [NS] (their payloads: the network sorts pairs)
        // (the entries are gathered first, as they come: then every thread of the workgroup has its four to work on at once)
        each(in_pass, push);
        __syncthreads();
        SPPH(7);
        nc1 = ncand;
        if (nc1 <= (u32)CK_CAND) {
          constexpr int PER = CK_CAND / CK_TPB;      // entries per thread
          const u32 stride = (nc1 + NS - 1) / NS, ns = stride ? (nc1 + stride - 1) / stride : 0u, SP = ns <= 128u ? 128u : 256u;
          for (u32 t = tid; t < SP; t += CK_TPB) { sk[t] = t < ns ? ck[t * stride] : ck_inf(); sp[t] = t < ns ? cp[t * stride] : ~0ULL; }      // (key AND payload: a key that hundreds of lists hold is cut into intervals like any other stretch)
          __syncthreads();
          ck_sort_block<CK_TPB>(sk, sp, SP, tid);
          CKey mk[PER]; u64 mp[PER]; u32 mb[PER], mo[PER];
#pragma unroll
          for (int x = 0; x < PER; x++) {
            const u32 i = tid + (u32)x * CK_TPB;
            mk[x] = ck_inf(); mp[x] = 0; mb[x] = 0xFFFFFFFFu; mo[x] = 0;
            if (i < nc1) { mk[x] = ck[i]; mp[x] = cp[i]; }
          }
          {   // the binary searches of my entries side by side (9 fixed steps cover 256 samples: their LDS reads overlap)
            u32 lo[PER], hi[PER];
#pragma unroll
            for (int x = 0; x < PER; x++) { lo[x] = 0; hi[x] = ns; }
#pragma unroll
            for (int it = 0; it < 9; it++) {
#pragma unroll
              for (int x = 0; x < PER; x++) {
                const u32 m = (lo[x] + hi[x]) >> 1;
                const bool open = lo[x] < hi[x];
                const bool below = open && ck_less(sk[min(m, NS - 1u)], sp[min(m, NS - 1u)], mk[x], mp[x]);
                lo[x] = below ? m + 1 : lo[x];
                hi[x] = (open && !below) ? m : hi[x];
              }
            }
#pragma unroll
            for (int x = 0; x < PER; x++) if (tid + (u32)x * CK_TPB < nc1) { mb[x] = lo[x]; mo[x] = atomicAdd(&ioff[lo[x]], 1u); }      // my number in my interval
          }
          __syncthreads();
          if (tid < 64) {      // wave 0: sizes -> offsets, five intervals per lane
            u32 v[5]; u32 sum = 0;
#pragma unroll
            for (int x = 0; x < 5; x++) { v[x] = ioff[tid * 5 + x]; sum += v[x]; }
            const u32 incl = wave_incl_scan(sum, (int)tid);
            u32 a = incl - sum;
#pragma unroll
            for (int x = 0; x < 5; x++) { ioff[tid * 5 + x] = a; a += v[x]; }
            if (tid == 63) ioff[NIV] = incl;
          }
          __syncthreads();
#pragma unroll
          for (int x = 0; x < PER; x++) if (mb[x] != 0xFFFFFFFFu) { const u32 ps = ioff[mb[x]] + mo[x]; ck[ps] = mk[x]; cp[ps] = mp[x]; }      // interval after interval
          __syncthreads();
          SPPH(2);
          u32 mr[PER];
#pragma unroll
          for (int x = 0; x < PER; x++) {
            mr[x] = 0xFFFFFFFFu;
            if (mb[x] != 0xFFFFFFFFu) {
              const u32 lo = ioff[mb[x]], hi = ioff[mb[x] + 1];
              u32 rr = lo;
              for (u32 j = lo; j < hi; j++) rr += ck_less(ck[j], cp[j], mk[x], mp[x]) ? 1u : 0u;
              mr[x] = rr;
            }
          }
          __syncthreads();
#pragma unroll
          for (int x = 0; x < PER; x++) if (mr[x] != 0xFFFFFFFFu) { ck[mr[x]] = mk[x]; cp[mr[x]] = mp[x]; }
          sorted = true;
        }
        __syncthreads();
      }
      if (thr > 1) {
        each(in_pass, [&](CKey k, u64) {
          const u32 hx = cl_mix(k);
          const u32 bit = hx & (CK_BITS - 1);
          const u32 old = atomicOr(&bits[bit >> 5], 1u << (bit & 31u));
          if ((old >> (bit & 31u)) & 1u) { const u32 b2 = (hx >> 7) & (CK_B2 - 1); atomicOr(&bits2[b2 >> 5], 1u << (b2 & 31u)); }
        });
        __syncthreads();
        SPPH(2);
      }
      if (thr > 1) {
        each(in_pass, [&](CKey k, u64 pl) {
          const u32 b2 = (cl_mix(k) >> 7) & (CK_B2 - 1); if (!((bits2[b2 >> 5] >> (b2 & 31u)) & 1u)) return;
          push(k, pl);
        });
        __syncthreads();
      }
      SPPH(3);
      const u32 nc = ncand;
      if (nc > (u32)CK_CAND) return CK_OVF;
      if (nc == 0) return 0u;
      if (!sorted) {
        u32 P = 128; while (P < nc) P <<= 1;
        for (u32 t = nc + tid; t < P; t += CK_TPB) { ck[t] = ck_inf(); cp[t] = ~0ULL; }      // (pads: larger than any entry, the key of all ones included)
        __syncthreads();
        ck_sort_block<CK_TPB>(ck, cp, P, tid);
      }
      SPPH(4);
      // kept runs: first entry of a run of >= thr equal keys (entries of one key come from different lists).  RESC: of >= thr SOLID
      // entries -- the payload sorts a key's non-solid entries behind its solid ones, so entry i + thr - 1 of the run decides; and a
      // run with at least `share` solid entries has its non-solid ones rescued: their statistics here, for every run, kept or not
      // (merge.hpp:234-247), their counts in the row below
      u32 mine = 0, km = 0, rl[CK_PTM];
      const u32 pt = (nc + CK_TPB - 1) / CK_TPB;     // consecutive entries per thread (<= CK_PTM)
#pragma unroll
      for (u32 x = 0; x < (u32)CK_PTM; x++) {
        rl[x] = 0;
        const u32 i = tid * pt + x;
        if (x < pt && i < nc) {
          const CKey k = ck[i];
          const bool first = i == 0 || !ck_eq(ck[i - 1], k);
          bool kept = first && (thr == 0 || (i + thr - 1 < nc && ck_eq(ck[i + thr - 1], k) && pl_solid(cp[i + thr - 1])));
          u32 len = 1;
          if (RESC && first) {
            while (i + len < nc && ck_eq(ck[i + len], k)) len++;
            u32 ns = 0; while (ns < len && pl_solid(cp[i + ns])) ns++;
            if (final && share && ns >= share) for (u32 e = ns; e < len; e++) { const u64 pl = cp[i + e]; atomicAdd(&T.stats[1 * (u64)T.N + pl_list(pl)], 1ULL); atomicAdd(&T.stats[5 * (u64)T.N + pl_list(pl)], (u64)(u32)pl); }
            // the row's entries: the solid ones, and all of them when the rescue applies -- marked by clearing / keeping the flags:
            // a non-solid entry that is NOT rescued gets count 0 (it must not reach the row)
            if (final && kept && !(share && ns >= share)) for (u32 e = ns; e < len; e++) cp[i + e] &= ~0xFFFFFFFFULL;
          } else if (kept) { while (i + len < nc && ck_eq(ck[i + len], k)) len++; }      // (len <= lists <= 4096, i < 2048)
          if (kept) {
            rl[x] = i | (len << 11);
            if (ORD) {      // the pass's row keys below this key (<= 56: six steps)
              u32 lo = 0, hi = dnp;
              while (lo < hi) { const u32 m = (lo + hi) >> 1; if (ck_lt(dkeys[dlo + m], k)) lo = m + 1; else hi = m; }
              rl[x] |= lo << 24;
            }
          }
          km |= (kept ? 1u : 0u) << x; mine += kept ? 1u : 0u;
        }
      }
      const u32 incl = wave_incl_scan(mine, (int)lane);
      if (lane == 63) wsum[wave] = incl;
      __syncthreads();
      u32 rank = incl - mine, nk = 0;
      for (u32 w = 0; w < CK_TPB / 64; w++) { if (w < wave) rank += wsum[w]; nk += wsum[w]; }
#pragma unroll
      for (u32 x = 0; x < (u32)CK_PTM; x++) if ((km >> x) & 1u) runs[rank++] = rl[x];
      return nk;      // (runs[] is complete behind the caller's next barrier)
    };

    // ---- a pass, second half: its nk kept runs become rows [rb, rb + nk) of the arena -- ORD: rows [rb, rb + nk + dnp), the rows of
    //      the pass's row keys (payload in C.dense) among them, by key ----
    auto dense_places = [&](const u32 t, const u32 nk, const u32 dlo, const u32 dnp) {      // thread t < dnp: where row key dlo + t's row goes among the pass's rows -- behind the kept runs with a smaller key
      const CKey k = dkeys[dlo + t];
      u32 lo = 0, hi = nk;
      while (lo < hi) { const u32 m = (lo + hi) >> 1; if (ck_lt(ck[CK_RUN_I0(runs[m])], k)) lo = m + 1; else hi = m; }
      dpos[t] = t + lo;
    };
    auto write_pass = [&](const u64 rb, const u32 nk, const u32 dlo, const u32 dnp, const bool have_dpos) {
      const u32 nrows = ORD ? nk + dnp : nk;
      const u32 dwords = (row_bytes - (u32)KW * 8u + 3u) / 4u, dtail = (row_bytes - (u32)KW * 8u) & 3u;      // ORD: payload dwords of a row of C.dense, bytes of the last one (0: all four)
      auto dense_src = [&](u32 i) -> const u32* { return reinterpret_cast<const u32*>(C.dense + (u64)(d0 + dlo + i) * C.dpitch); };
      auto dense_word = [&](const u32* src, u32 w) -> u32 {      // dword w of the payload, the bytes behind it masked off
        if (w >= dwords) return 0u;
        const u32 v = src[w];
        return (w + 1 == dwords && dtail) ? v & ((1u << (8u * dtail)) - 1u) : v;
      };
      if (ORD && !have_dpos) {
        if (tid < dnp) dense_places(tid, nk, dlo, dnp);
        __syncthreads();
      }
      // lanes per row: a wave for a count row (4 N bytes); 16 for a PA row (N / 8 bytes), 8 when that is at most 128 bytes
      const u32 SG = MODE == 0 ? 64u : row_bytes <= 128u ? 8u : 16u;
      const u32 sg = tid / SG, sl = tid % SG;      // my lane group, my lane in it
      // a PA row starts at any byte (rows are row_bytes apart): the aligned dwords inside the row are stored whole, each taken
      // from two words of the row as word(w) gives them, the <= 3 bytes before and after them one by one
      auto put_bytes = [&](u8* const row, auto&& word) {
        const u32 head = (4u - (u32)((uintptr_t)row & 3u)) & 3u;
        const u32 nd = (row_bytes - head) / 4, tail0 = head + 4 * nd;
        if (sl < head) row[sl] = (u8)(word(0) >> (sl * 8));
        for (u32 t = sl; t < nd; t += SG) {
          const u32 off = head + 4 * t, w = off >> 2;
          const u64 two = (u64)word(w) | ((u64)word(w + 1) << 32);
          reinterpret_cast<u32*>(row + off)[0] = (u32)(two >> ((off & 3u) * 8));
        }
        if (sl < row_bytes - tail0) { const u32 t = tail0 + sl; row[t] = (u8)(word(t >> 2) >> ((t & 3u) * 8)); }
      };
      // ORD: the row keys' rows -- key from the group's row keys, payload from C.dense -- a lane group per row.  Count rows: four
      // rows of a wave in flight at once (a row is read at the latency of HBM: one at a time a wave would spend its time waiting)
      auto dense_rows = [&]() {
        if (MODE == 0) {
          constexpr int RF = 4;
          const bool wide = (row_bytes & 7u) == 0;      // (C.dense's rows are 8-byte aligned)
          const u32 n8 = row_bytes / 8, n4 = row_bytes / 4;
          for (u32 i0 = wave * RF; i0 < dnp; i0 += (CK_TPB / 64) * RF) {
            for (u32 t0 = 0; t0 < (wide ? n8 : n4); t0 += 8 * 64) {
              u64 w[RF][8];
#pragma unroll
              for (int r = 0; r < RF; r++) {
                const u32 i = i0 + r;
                const u32* const src = dense_src(i < dnp ? i : 0);
                u32 kw4[4] = {0, 0, 0, 0};
                ck_store(kw4, dkeys[dlo + (i < dnp ? i : 0)]);
#pragma unroll
                for (int x = 0; x < 8; x++) {
                  const u32 t = t0 + 64 * x + lane;
                  w[r][x] = 0;
                  if (i < dnp) {
                    if (wide) { if (t < n8) w[r][x] = t < (u32)KW ? ((u64)kw4[2 * (t & 1u)] | ((u64)kw4[2 * (t & 1u) + 1] << 32)) : reinterpret_cast<const u64*>(src)[t - KW]; }
                    else if (t < n4) w[r][x] = t < 2u * KW ? kw4[t & 3u] : src[t - 2u * KW];
                  }
                }
              }
#pragma unroll
              for (int r = 0; r < RF; r++) {
                const u32 i = i0 + r;
                if (i < dnp) {
                  u8* const row = T.out + (rb + dpos[i]) * row_bytes;
#pragma unroll
                  for (int x = 0; x < 8; x++) {
                    const u32 t = t0 + 64 * x + lane;
                    if (wide) { if (t < n8) reinterpret_cast<u64*>(row)[t] = w[r][x]; }
                    else if (t < n4) reinterpret_cast<u32*>(row)[t] = (u32)w[r][x];
                  }
                }
              }
            }
          }
        } else {
          for (u32 i = sg; i < dnp; i += CK_TPB / SG) {
            u8* const row = T.out + (rb + dpos[i]) * row_bytes;
            const u32* const src = dense_src(i);
            u32 kw4[4] = {0, 0, 0, 0};
            ck_store(kw4, dkeys[dlo + i]);
            put_bytes(row, [&](u32 w) -> u32 { return w < 2u * KW ? kw4[w & 3u] : dense_word(src, w - 2u * KW); });
          }
        }
      };
      if (MODE == 1 && nrows && row_bytes * 8u + 16u <= (u32)CK_STAGE) {
        // PA rows (N / 8 bytes behind the key, not a multiple of anything): the pass's rows are one contiguous run of the arena.  The
        // run's bytes are cut into 16-byte-aligned pieces of CB bytes; a wave assembles a piece in its own LDS block exactly as it
        // will lie in memory -- a lane per row that touches the piece (a row across a cut is assembled, in part, by both sides):
        // the key's dwords and the lists' bits are OR-ed into the zeroed block, <= 9 LDS atomics for a private k-mer's row -- and
        // streams it out as aligned 16-byte words.  Only the run's first and last word share their 16 bytes with other groups'
        // rows and go out byte by byte.  (Before: 8 lanes per row storing single dwords at odd offsets.)  No workgroup barrier.
        u32* const st = uni + wave * (CK_STAGE / 4);
        constexpr u32 CB = (u32)CK_STAGE & ~15u;
        const uintptr_t A0 = (uintptr_t)(T.out + rb * row_bytes), Ae = A0 + (uintptr_t)nrows * row_bytes, As = A0 & ~(uintptr_t)15;
        const u32 npieces = (u32)((Ae - As + CB - 1) / CB);
        for (u32 pc = wave; pc < npieces; pc += CK_TPB / 64) {
          const uintptr_t lo_a = As + (uintptr_t)pc * CB, hi_a = lo_a + CB < Ae ? lo_a + CB : Ae;      // the piece: [lo_a, hi_a), lo_a aligned
          const u32 nby = (u32)(hi_a - lo_a), nw16 = (nby + 15u) / 16u;
          for (u32 w = lane; w < nw16; w += 64) reinterpret_cast<uint4*>(st)[w] = make_uint4(0, 0, 0, 0);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the zeros are in before the ORs
          // rows that touch the piece: [r0, r1]
          const u32 r0 = lo_a > A0 ? (u32)((lo_a - A0) / row_bytes) : 0u, r1 = (u32)((hi_a - 1 - A0) / row_bytes);
          auto or_dword = [&](int o, u32 w) {      // the 4 bytes of w at byte o of the piece, clipped to it
            const int ix = o >> 2; const u32 sh = ((u32)o & 3u) * 8u;
            const u32 w0 = w << sh, w1 = sh ? w >> (32u - sh) : 0u;
            if (w0 && ix >= 0 && (u32)ix < nw16 * 4u) atomicOr(&st[ix], w0);
            if (w1 && ix + 1 >= 0 && (u32)(ix + 1) < nw16 * 4u) atomicOr(&st[ix + 1], w1);
          };
          auto sparse_row = [&](const u32 r, const u32 rv) {      // row r of the pass's rows: the kept run rv
            const u32 i0 = CK_RUN_I0(rv), len = CK_RUN_LEN(rv);
            const int b = (int)(long long)((long long)(A0 + (uintptr_t)r * row_bytes) - (long long)lo_a);      // the row's first byte in the piece (negative: it began in the piece before)
            u32 kw4[4] = {0, 0, 0, 0};
            ck_store(kw4, ck[i0]);
#pragma unroll
            for (u32 d = 0; d < 2u * KW; d++) or_dword(b + 4 * (int)d, kw4[d]);
            for (u32 e = 0; e < len; e++) {
              const u64 pl = cp[i0 + e];
              if (RESC && !(u32)pl) continue;      // (a non-solid record that is not rescued, or recurrence-min 0's lone non-solid record: no bit)
              const u32 li = pl_list(pl);
              const int bo = b + 8 * (int)KW + (int)(li >> 3);
              if (bo >= 0 && (u32)bo < nw16 * 16u) atomicOr(&st[bo >> 2], 1u << (((u32)bo & 3u) * 8u + (li & 7u)));
            }
          };
          if (!ORD) {
            for (u32 r = r0 + lane; r <= r1; r += 64) sparse_row(r, runs[r]);
          } else if (nk) {
            // the kept runs whose rows touch the piece: run j's row is j + (row keys below it, <= dnp), so they are among runs
            // r0 - dnp .. r1 -- looked at without a search.  (The row keys' rows are left as zeros here and written afterwards.)
            const u32 jmin = r0 > dnp ? r0 - dnp : 0u, jmax = min(nk - 1u, r1);
            for (u32 j = jmin + lane; j <= jmax; j += 64) { const u32 rv = runs[j], r = j + CK_RUN_ND(rv); if (r >= r0 && r <= r1) sparse_row(r, rv); }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          // (bytes of the block beyond [A0, Ae) belong to other rows: whatever was OR-ed there is not written)
          u8* const a0 = reinterpret_cast<u8*>(lo_a);
          const u32 vlo = A0 > lo_a ? (u32)(A0 - lo_a) : 0u;       // valid bytes of the piece: [vlo, nby)
          for (u32 w = lane; w < nw16; w += 64) {
            const uint4 v = reinterpret_cast<const uint4*>(st)[w];
            const u32 lo = w * 16u, hi = lo + 16u;
            if (lo >= vlo && hi <= nby) reinterpret_cast<uint4*>(a0)[w] = v;
            else {
              const u32 vv[4] = {v.x, v.y, v.z, v.w};
              for (u32 t = max(lo, vlo); t < min(hi, nby); t++) a0[t] = (u8)(vv[(t - lo) >> 2] >> ((t & 3u) * 8u));
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (my reads of the block are done before the next round's zeros)
        }
        if (ORD && dnp) {
          __syncthreads();      // (the pieces' zeros at the row keys' rows are out before the rows themselves)
          dense_rows();
        }
      } else if (nrows) {
        // kept runs [jlo, jhi) -> rows, a lane group per row
        auto sparse_rows = [&](const u32 jlo, const u32 jhi) {
        for (u32 j = jlo + sg; j < jhi; j += CK_TPB / SG) {
          const u32 rv = runs[j], i0 = CK_RUN_I0(rv), len = CK_RUN_LEN(rv);
          const CKey key = ck[i0];
          u8* const row = T.out + (rb + (ORD ? j + CK_RUN_ND(rv) : j)) * row_bytes;
          u32 kw4[4] = {0, 0, 0, 0};
          ck_store(kw4, key);                      // the key's dwords (2 or 4 of them)
          auto kword = [&](u32 w) -> u32 { return w == 0 ? kw4[0] : w == 1 ? kw4[1] : w == 2 ? kw4[2] : kw4[3]; };
          if (len == 1) {
            // the usual row here: a key one list holds (a private k-mer) -- every lane knows the whole row, no staging
            const u64 pl = cp[i0];
            const u32 li = pl_list(pl), cnt = (u32)pl;
            const u32 bitv = (MODE == 1 && RESC && cnt == 0) ? 0u : 1u;      // (RESC, recurrence-min 0: a lone non-solid record is a row of zeros)
            if (MODE == 0) {
              if ((row_bytes & 7u) == 0) {
                // 16-byte stores (round 5): the rows are what this kernel is made of -- 10 GB of them per launch of configs[2] --, and a
                // store INSTRUCTION is what it pays for: with 8-byte stores it ran at the ~7 bytes per cycle and CU that is the issue
                // limit of dwordx2 stores (MI355X_MICROARCH.md), not at the memory's rate
                auto word = [&](u32 t) -> u64 { return t < (u32)KW ? ((u64)kword(2 * t) | ((u64)kword(2 * t + 1) << 32)) : (t - KW == (li >> 1) ? (u64)cnt << ((li & 1u) * 32) : 0ULL); };
                const u32 n8 = row_bytes / 8, n16 = n8 >> 1;
                for (u32 t = sl; t < n16; t += SG) { u64x2 v; v.x = word(2 * t); v.y = word(2 * t + 1); *reinterpret_cast<u64x2*>(row + 16u * t) = v; }
                if ((n8 & 1u) && sl == 0) reinterpret_cast<u64*>(row)[n8 - 1] = word(n8 - 1);
              } else {
                u32* const r4 = reinterpret_cast<u32*>(row);
                for (u32 t = sl; t < row_bytes / 4; t += SG) r4[t] = t < 2u * KW ? kword(t) : (t - 2u * KW == li ? cnt : 0u);
              }
            } else {
              put_bytes(row, [&](u32 w) -> u32 { return w < 2u * KW ? kword(w) : (w - 2u * KW == (li >> 5) ? bitv << (li & 31u) : 0u); });
            }
            continue;
          }
          if (MODE == 0) {
            // row = key + N counts: 8-byte stores (rows start at multiples of 8 when row_bytes is one: N even), else 4-byte ones
            if ((row_bytes & 7u) == 0) {
              auto word = [&](u32 t) -> u64 { return t < (u32)KW ? ((u64)kword(2 * t) | ((u64)kword(2 * t + 1) << 32)) : 0ULL; };
              const u32 n8 = row_bytes / 8, n16 = n8 >> 1;
              for (u32 t = sl; t < n16; t += SG) { u64x2 v; v.x = word(2 * t); v.y = word(2 * t + 1); *reinterpret_cast<u64x2*>(row + 16u * t) = v; }
              if ((n8 & 1u) && sl == 0) reinterpret_cast<u64*>(row)[n8 - 1] = word(n8 - 1);
            } else {
              u32* const r4 = reinterpret_cast<u32*>(row);
              for (u32 t = sl; t < row_bytes / 4; t += SG) r4[t] = t < 2u * KW ? kword(t) : 0u;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            for (u32 e = sl; e < len; e += SG) { const u64 pl = cp[i0 + e]; reinterpret_cast<u32*>(row + KW * 8)[pl_list(pl)] = (u32)pl; }
          } else {
            u32* const pr = parow + sg * (SG == 8u ? 40u : 136u);
            const u32 nby = row_bytes - KW * 8, nw = (nby + 3) / 4;
            for (u32 t = sl; t < nw + 2 * KW + 1; t += SG) pr[t] = t < 2u * KW ? kword(t) : 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            for (u32 e = sl; e < len; e += SG) { const u64 pl = cp[i0 + e]; const u32 li = pl_list(pl); if (!RESC || (u32)pl) atomicOr(&pr[2 * KW + (li >> 5)], 1u << (li & 31u)); }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            put_bytes(row, [&](u32 w) -> u32 { return pr[w]; });
          }
        }
        };
        if (ORD && MODE == 0 && (row_bytes & 7u) == 0 && row_bytes / 8 <= 512u) {
          // count rows in file order: the row keys' rows are READ (C.dense) and a wave waits for them at the latency of HBM, the
          // other rows are only written.  So the two are interleaved: a wave asks for RF of its row keys' rows, writes its share of
          // a stretch of the kept runs' rows while they travel, stores them, asks for the next RF.  (All of them up front, the wave
          // waiting: rows phase 2.8x the arena build's for a fifth more rows.)
          constexpr int RF = 4;
          const u32 n8 = row_bytes / 8;
          const u32 rounds = max(1u, (dnp + (CK_TPB / 64) * RF - 1) / ((CK_TPB / 64) * RF));
          if (C.dnarrow) {
            // the NARROW side store: a byte per count (+ a flag byte per column block: 1 = some count of that block's slice of the row is
            // above 254: its byte is 255 and the count lies in the 4-byte row).  A lane takes two groups of eight lists: an 8-byte load
            // each, four 8-byte stores each; the row's eight flag bytes in one (uniform) load.
            const u8* const nar = C.dnarrow; const u32 npitch = C.npitch, nbs = C.nb, NL = T.N, FO = npitch - 16u;      // (the row's last 16 bytes: a flag byte per column block)
            // (a lane takes SIXTEEN lists: one 16-byte load, four 16-byte stores -- a block is a multiple of 16 lists wide, so the sixteen
            //  share its flag byte)
            const u32 ng = (NL + 15u) / 16u;      // (<= 64: row_bytes / 8 <= 512)
            const u32 l0 = min(16u * lane, NL - 1u), fsh = 8u * (l0 / nbs);
            const u32 nval = lane < ng ? min(16u, NL - 16u * lane) : 0u;      // my lists (even: the rows are whole 8-byte words)
            for (u32 c = 0; c < rounds; c++) {
              const u32 i0 = c * (CK_TPB / 64) * RF + wave * RF;
              uint4 nv[RF]; u64 fl[RF];
#pragma unroll
              for (int r = 0; r < RF; r++) {
                const u32 i = i0 + r;
                const u8* const nrow = nar + (u64)(d0 + dlo + (i < dnp ? i : 0)) * npitch;
                fl[r] = i < dnp ? *reinterpret_cast<const u64*>(nrow + FO) : 0ULL;
                nv[r] = (i < dnp && lane < ng) ? *reinterpret_cast<const uint4*>(nrow + 16u * lane) : make_uint4(0, 0, 0, 0);
              }
              sparse_rows((u32)(((u64)nk * c) / rounds), (u32)(((u64)nk * (c + 1)) / rounds));
#pragma unroll
              for (int r = 0; r < RF; r++) {
                const u32 i = i0 + r;
                if (i >= dnp) continue;
                u8* const row = T.out + (rb + dpos[i]) * row_bytes;
                if (lane == 0) {
                  u32 kw4[4] = {0, 0, 0, 0};
                  ck_store(kw4, dkeys[dlo + i]);
#pragma unroll
                  for (u32 d = 0; d < 2u * KW; d++) reinterpret_cast<u32*>(row)[d] = kw4[d];
                }
                if (nval == 0) continue;
                u8* const out = row + 8u * KW + 64u * lane;      // my sixteen counts' place
                const u32 by[4] = {nv[r].x, nv[r].y, nv[r].z, nv[r].w};
                const bool flagged = ((fl[r] >> fsh) & 0xFFULL) != 0;
                if (!flagged) {
#pragma unroll
                  for (u32 q4 = 0; q4 < 4; q4++) {      // four lists a store
                    const u32 b = by[q4];
                    if (4 * q4 + 4 <= nval) { u64x2 v; v.x = (u64)(b & 0xFFu) | ((u64)((b >> 8) & 0xFFu) << 32); v.y = (u64)((b >> 16) & 0xFFu) | ((u64)(b >> 24) << 32); *reinterpret_cast<u64x2*>(out + 16u * q4) = v; }
                    else if (4 * q4 + 2 <= nval) *reinterpret_cast<u64*>(out + 16u * q4) = (u64)(b & 0xFFu) | ((u64)((b >> 8) & 0xFFu) << 32);
                  }
                } else {      // a block some count of whose slice of this row lies in the 4-byte row (its byte is 255)
                  const u32* const wsrc = dense_src(i); u32* const cnt = reinterpret_cast<u32*>(row + 8u * KW);
                  for (u32 l = 0; l < nval; l++) { const u32 b = (by[l >> 2] >> (8u * (l & 3u))) & 0xFFu; cnt[16u * lane + l] = b == 255u ? wsrc[16u * lane + l] : b; }
                }
              }
            }
          } else
          for (u32 c = 0; c < rounds; c++) {
            const u32 i0 = c * (CK_TPB / 64) * RF + wave * RF;
            u64 w[RF][8];
#pragma unroll
            for (int r = 0; r < RF; r++) {
              const u32 i = i0 + r;
              const u64* const s8 = reinterpret_cast<const u64*>(dense_src(i < dnp ? i : 0));
              u32 kw4[4] = {0, 0, 0, 0};
              ck_store(kw4, dkeys[dlo + (i < dnp ? i : 0)]);
#pragma unroll
              for (int x = 0; x < 8; x++) {
                const u32 t = 64 * x + lane;
                w[r][x] = 0;
                if (i < dnp && t < n8) w[r][x] = t < (u32)KW ? ((u64)kw4[2 * (t & 1u)] | ((u64)kw4[2 * (t & 1u) + 1] << 32)) : s8[t - KW];
              }
            }
            sparse_rows((u32)(((u64)nk * c) / rounds), (u32)(((u64)nk * (c + 1)) / rounds));
#pragma unroll
            for (int r = 0; r < RF; r++) {
              const u32 i = i0 + r;
              if (i < dnp) {
                u64* const row = reinterpret_cast<u64*>(T.out + (rb + dpos[i]) * row_bytes);
#pragma unroll
                for (int x = 0; x < 8; x++) { const u32 t = 64 * x + lane; if (t < n8) row[t] = w[r][x]; }
              }
            }
          }
        } else {
          if (ORD && MODE == 0) dense_rows();
          sparse_rows(0, nk);
          if (ORD && MODE == 1) dense_rows();
        }
      }
      __syncthreads();
      SPPH(6);
    };

    if (!ORD) {
      for (u32 pass = 0; pass < npass; pass++) {
        const u32 nk = sort_pass(pass, true, 0, 0);
        if (nk == CK_OVF) { hand_back(3); return false; }
        if (tid == 0) { dir[(u64)gid * CK_NPASS + pass].base = 0; dir[(u64)gid * CK_NPASS + pass].n = 0; }
        if (nk == 0) { __syncthreads(); continue; }
        if (tid == 0) {
          const u64 at = atomicAdd(&T.ctrl[0], (u64)nk);      // rows of the arena: behind the row keys' rows
          atomicAdd(&T.ctrl[3], (u64)nk); atomicAdd(&T.ctrl[6], (u64)nk);
          if (at + nk > T.out_cap_rows) { atomicOr(&T.ctrl[2], (u64)ERR_ROWS_OVERFLOW); rowbase = 0xFFFFFFFFu; }
          else { rowbase = (u32)at; dir[(u64)gid * CK_NPASS + pass].base = (u32)at; dir[(u64)gid * CK_NPASS + pass].n = nk; }
        }
        __syncthreads();
        const u32 rb = rowbase;
        SPPH(5);
        if (rb != 0xFFFFFFFFu) write_pass((u64)rb, nk, 0, 0, false);
        else __syncthreads();
      }
      return true;
    }
    // ---- ORD: count the group's rows, publish, learn its place, write ----
    __syncthreads();      // (dkeys)
    u32 nks = 0;
    while (!failed && tot) {
      bool ovf = false;
      nks = 0;
      if (npass == 1) { const u32 r = sort_pass(0, true, 0, dn); if (r == CK_OVF) ovf = true; else nks = r; __syncthreads(); }
      else for (u32 p = 0; p < npass; p++) {
        const u32 dlo = (u32)(((u64)dn * p) / npass), dhi = (u32)(((u64)dn * (p + 1)) / npass);
        const u32 r = sort_pass(p, false, dlo, dhi - dlo);
        __syncthreads();
        if (r == CK_OVF) { ovf = true; break; }
        nks += r;
      }
      if (!ovf) break;
      if (npass * 2 > (u32)CK_NPASS || npass * 2 > dn) { hand_back(3); failed = true; break; }
      npass *= 2;      // (an uneven stretch of keys: finer passes)
    }
    const u64 grows = failed ? 0ULL : (u64)dn + nks;
    if (wave == 0) { const u64 b = ck_lookback(C.chain, gord, grows, lane, &T.ctrl[2]); if (lane == 0) s_base = b; }
    else if (npass == 1 && !failed && tid - 64u < dn) dense_places(tid - 64u, nks, 0, dn);      // (beside the look-back: where the row keys' rows go)
    __syncthreads();
    SPPH(5);
    if (failed) return false;
    if (s_base == ~0ULL) { hand_back(12); return false; }      // (the look-back gave up: see ck_lookback; its own counter -- [2] counts kept keys outside the row keys)
    // (the task's row counters: behind the rows, where no barrier waits for the atomics -- the next one is the next ticket's)
    auto count_rows = [&]() { if (tid == 0 && nks) { atomicAdd(&T.ctrl[0], (u64)nks); atomicAdd(&T.ctrl[3], (u64)nks); atomicAdd(&T.ctrl[6], (u64)nks); } };
    const u64 base = s_base;
    if (base + grows > T.out_cap_rows) { if (tid == 0) atomicOr(&T.ctrl[2], (u64)ERR_ROWS_OVERFLOW); count_rows(); return true; }      // (the batch is re-run with arenas of the size ctrl[0] asks for)
    if (npass == 1) write_pass(base, nks, 0, dn, true);
    else {
      u64 b = base;
      for (u32 p = 0; p < npass; p++) {
        const u32 dlo = (u32)(((u64)dn * p) / npass), dhi = (u32)(((u64)dn * (p + 1)) / npass);
        const u32 nk = sort_pass(p, true, dlo, dhi - dlo);
        __syncthreads();
        write_pass(b, nk, dlo, dhi - dlo, false);
        b += nk + (dhi - dlo);
      }
    }
    count_rows();
    return true;
  };

  if (!ORD) {
    // one workgroup per (task, range) x CK_Z, sharing the range's slice groups
    const u32 item = blockIdx.x;
    if (item < n_items) {
      const TaskDev& T = tasks[items[item].x];
      const ColsDev& C = cols[items[item].x];
      if (!__syncthreads_or((__hip_atomic_load(&T.ctrl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (u64)(ERR_FALLBACK | ERR_ROWS_OVERFLOW)) != 0)) {      // (one decision per workgroup: see k_merge_cols)
        const u32 range = items[item].y, rt = C.rt;
        const u32 s_lo = C.rbounds[range], s_hi = C.rbounds[range + 1];
        const u32 ngroups = max(1u, (s_hi - s_lo + rt - 1) / rt) * CL_HALVES;
        for (u32 q = blockIdx.y; q < ngroups; q += CK_Z) if (!group(T, C, range, q, s_lo, s_hi, 0u)) break;
      }
    }
  } else {
    // persistent: ticket t -> group t / n_tasks of task t % n_tasks (the tasks' chains advance side by side).  A ticket is taken
    // only when the workgroup is ready to start on it: requesting the next one ahead (to hide the atomic's round trip) was tried and
    // made the kernel 30 % slower -- a ticket held back for the length of a group holds back every later group of its task's chain.
    for (;;) {
      __syncthreads();
      if (tid == 0) s_tk = atomicAdd(&tkt[1], 1u);
      __syncthreads();
      const u32 t = s_tk;
      const u32 ti = t % n_tasks, g = t / n_tasks;
      if (g >= tkt[2] || g >= tkt[3]) {      // (k_cols_prep: the largest number of groups a task of the batch has; the host: the most any task has room for)
        if (g < tkt[2] && tid == 0) atomicAdd(&kmx_cols_dbg[9], 1u);
        break;
      }
      const TaskDev& T = tasks[ti];
      const ColsDev& C = cols[ti];
      if (g >= C.ngcap) continue;
      // (the group's descriptor and the task's error word in one round trip: no branch between the two loads)
      const uint4 rq = C.gmap[g];
      const u64 terr = __hip_atomic_load(&T.ctrl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (one decision per workgroup -- other workgroups raise the word while this one reads it: see k_merge_cols)
      const bool dead = __syncthreads_or((terr & (u64)(ERR_FALLBACK | ERR_ROWS_OVERFLOW)) != 0) != 0;
      if (rq.x == 0) continue;      // (beyond the task's groups: the map is zeroed before every batch)
      if (dead) {
        // a task that is handed back anyway: nobody may wait for this group
        if (tid == 0) __hip_atomic_store(&C.chain[g], 2ULL << 62, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
      }
      (void)group(T, C, rq.x - 1u, rq.y, rq.z, rq.w, g);
    }
  }
#ifdef KMX_PHASE_PROF
  __syncthreads();
  if (tid == 0) for (int i = 0; i < 8; i++) atomicAdd(&kmx_sparse_prof[i], (u64)spt[i]);
#endif
}

// ---- the body of a task the column-blocked merge completed: row keys' rows and the sparse rows, interleaved by key.
//      k_cols_offsets: position of every slice group's first row in the body (a workgroup per task walks the directory);
//      k_cols_gather: a workgroup per slice group ranks its (<= 9) ascending lists against each other and copies the rows. ----
__global__ __launch_bounds__(1024)
void k_cols_offsets(const ColsDev* __restrict__ cols, u32 task, u64* __restrict__ goff)
{
  const ColsDev& C = cols[task];
  const SpDir* dir = reinterpret_cast<const SpDir*>(C.spdir);
  __shared__ u64 part[1024];
  const u32 ng = C.slots_cap * CL_HALVES, tid = threadIdx.x;
  const u32 per = (ng + 1023) / 1024;
  u64 sum = 0;
  for (u32 g = tid * per; g < min(ng, (tid + 1) * per); g++) { sum += dir[(u64)g * CK_NPASS].dense_n; for (int p = 0; p < CK_NPASS; p++) sum += dir[(u64)g * CK_NPASS + p].n; }
  part[tid] = sum;
  __syncthreads();
  if (tid == 0) { u64 a = 0; for (u32 t = 0; t < 1024; t++) { const u64 v = part[t]; part[t] = a; a += v; } }
  __syncthreads();
  u64 a = part[tid];
  for (u32 g = tid * per; g < min(ng, (tid + 1) * per); g++) { goff[g] = a; a += dir[(u64)g * CK_NPASS].dense_n; for (int p = 0; p < CK_NPASS; p++) a += dir[(u64)g * CK_NPASS + p].n; }
}

constexpr int GA_TPB = 512;
constexpr int GA_KEYS = 4096 / KW;      // keys of a group held in LDS (more: ranks come from global memory)
// COPY: the rows go to their places in `body`.  !COPY: only the ORDER is written -- order[d] = the arena row that is row d of the body
// (body then points at a u32 array): a consumer that writes the rows to a file anyway puts them in order there (pwrite at d * row
// bytes), and no second copy of the matrix ever exists in HBM.
template <bool COPY>
__global__ __launch_bounds__(GA_TPB)
void k_cols_gather(const TaskDev* __restrict__ tasks, const ColsDev* __restrict__ cols, u32 task, const u64* __restrict__ goff, u8* __restrict__ body)
{
  const TaskDev& T = tasks[task];
  const ColsDev& C = cols[task];
  const SpDir* dir = reinterpret_cast<const SpDir*>(C.spdir) + (u64)blockIdx.x * CK_NPASS;
  __shared__ CKey keys[GA_KEYS];
  __shared__ u32 lo[CK_NPASS + 2];      // list l = keys [lo[l], lo[l + 1]): 0 the row keys, 1.. the passes
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row_bytes = T.row_bytes;
  if (tid == 0) { u32 a = 0; lo[0] = 0; a += dir[0].dense_n; lo[1] = a; for (int p = 0; p < CK_NPASS; p++) { a += dir[p].n; lo[2 + p] = a; } }
  __syncthreads();
  const u32 n = lo[CK_NPASS + 1];
  if (n == 0) return;
  const u32 d0 = dir[0].dense_first, dn = dir[0].dense_n;
  auto src_row = [&](u32 i) -> const u8* {      // i-th row of the group in list order
    if (i < dn) return T.out + (u64)(d0 + i) * row_bytes;
    int p = 0; while (i >= lo[2 + p]) p++;
    return T.out + (u64)(dir[p].base + (i - lo[1 + p])) * row_bytes;
  };
  auto key_at = [&](u32 i) -> CKey { return i < (u32)GA_KEYS ? keys[i] : ck_load(src_row(i)); };
  for (u32 i = tid; i < min(n, (u32)GA_KEYS); i += GA_TPB) keys[i] = i < dn ? reinterpret_cast<const CKey*>(C.skel)[d0 + i] : ck_load(src_row(i));
  __syncthreads();
  if (!COPY) {      // a thread per row: its rank, and where it lies in the arena
    u32* const order = reinterpret_cast<u32*>(body) + goff[blockIdx.x];
    for (u32 i = tid; i < n; i += GA_TPB) {
      const CKey k = key_at(i);
      u32 rank = 0;
      for (int l = 0; l <= CK_NPASS; l++) {
        u32 a = lo[l], b = lo[l + 1];
        if (i >= a && i < b) { rank += i - a; continue; }
        while (a < b) { const u32 m = (a + b) >> 1; if (ck_lt(key_at(m), k)) a = m + 1; else b = m; }
        rank += a - lo[l];
      }
      order[rank] = (u32)((u64)(src_row(i) - T.out) / row_bytes);
    }
    return;
  }
  u8* const dst0 = body + goff[blockIdx.x] * row_bytes;
  // (round 3) the ranks of the group's rows first, a THREAD per row (round 2: every lane of a wave repeated its row's binary searches),
  // kept in LDS for the rows whose keys are there; then the rows are copied, a wave per row
  __shared__ u32 rnk[GA_KEYS];
  auto rank_of = [&](u32 i) -> u32 {
    const CKey k = key_at(i);
    u32 rank = 0;
    for (int l = 0; l <= CK_NPASS; l++) {
      u32 a = lo[l], b = lo[l + 1];
      if (i >= a && i < b) { rank += i - a; continue; }
      while (a < b) { const u32 m = (a + b) >> 1; if (ck_lt(key_at(m), k)) a = m + 1; else b = m; }
      rank += a - lo[l];
    }
    return rank;
  };
  for (u32 i = tid; i < min(n, (u32)GA_KEYS); i += GA_TPB) rnk[i] = rank_of(i);
  __syncthreads();
  for (u32 i = wave; i < n; i += GA_TPB / 64) {
    const u32 rank = i < (u32)GA_KEYS ? rnk[i] : rank_of(i);
    const u8* s = src_row(i);
    u8* d = dst0 + (u64)rank * row_bytes;
    if ((row_bytes & 7u) == 0) {      // (count rows of an even number of lists: 8-byte pieces, four loads in flight per lane)
      const u32 n8 = row_bytes / 8;
      for (u32 t0 = 0; t0 < n8; t0 += 512) {      // (a row of 1000 counts is 501 words: all of it requested before the first store)
        u64 w[8];
#pragma unroll
        for (int x = 0; x < 8; x++) { const u32 t = t0 + 64 * x + lane; w[x] = t < n8 ? reinterpret_cast<const u64*>(s)[t] : 0ULL; }
#pragma unroll
        for (int x = 0; x < 8; x++) { const u32 t = t0 + 64 * x + lane; if (t < n8) reinterpret_cast<u64*>(d)[t] = w[x]; }
      }
    } else if ((row_bytes & 3u) == 0) for (u32 t = lane; t < row_bytes / 4; t += 64) reinterpret_cast<u32*>(d)[t] = reinterpret_cast<const u32*>(s)[t];
    else for (u32 t = lane; t < row_bytes; t += 64) d[t] = s[t];
  }
}

void cols_dbg_dump()
{
  u32 h[16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(kmx_cols_dbg), sizeof(h)) != hipSuccess) return;
  if (h[9]) fprintf(stderr, "[kmx merge] k_cols_sparse: the group count of the batch was beyond every task's room (%u workgroups stopped at the room)\n", h[9]);
  if (h[8]) fprintf(stderr, "[kmx merge] k_cols_sparse: %u look-backs given up (last: group %u, %u entries still unpublished in front of it)\n", h[8], h[10], h[11]);
  fprintf(stderr, "[kmx merge] k_merge_cols hand-back reasons: no collision-free row table %u, slice overflow %u (wave-tiles; largest %u, in a range's last tile %u, first tile %u), kept key outside the row keys %u, check table full %u, lists too divergent %u (tasks)\n", h[0], h[1], h[5], h[6], h[7], h[2], h[3], h[4]);
  memset(h, 0, sizeof(h)); (void)hipMemcpyToSymbol(HIP_SYMBOL(kmx_cols_dbg), h, sizeof(h));
}
#ifdef KMX_PHASE_PROF
void cols_phase_prof_dump()
{
  u64 h[8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(kmx_cols_prof), sizeof(h)) != hipSuccess) return;
  u64 tot = 0; for (int i = 0; i < 8; i++) tot += h[i];
  static const char* nm[8] = {"item setup", "window wait", "scan", "bar(scan)", "tile out", "bar(out)", "-", "-"};
  for (int i = 0; i < 6; i++) fprintf(stderr, "[cols] %-12s %6.2f%%  %llu\n", nm[i], tot ? 100.0 * h[i] / tot : 0.0, h[i]);
  memset(h, 0, sizeof(h));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(kmx_cols_prof), h, sizeof(h));
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(kmx_sparse_prof), sizeof(h)) != hipSuccess) return;
  tot = 0; for (int i = 0; i < 8; i++) tot += h[i];
  static const char* sn[8] = {"group entries", "zero maps", "mark twice | sample + place", "candidates | rank", "sort", "runs + claim", "rows", "gather (rec-min 1)"};
  for (int i = 0; i < 8; i++) fprintf(stderr, "[sparse] %-14s %6.2f%%  %llu\n", sn[i], tot ? 100.0 * h[i] / tot : 0.0, h[i]);
  memset(h, 0, sizeof(h));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(kmx_sparse_prof), h, sizeof(h));
}
#endif

// ---- host side ------------------------------------------------------------------------------------------
int cols_lds_bytes() { return CL_IMG + CL_NT * CL_PT * (int)sizeof(ClEnt) + 64 + 256; }
u32 cols_halves() { return CL_HALVES; }
u32 cols_wgs_per_cu() { return CL_WGS; }
u32 cols_block_lists() { return CL_NB; }
u32 cols_tile_rows(u32 nb) { return std::max(1u, std::min<u32>((u32)CL_RT, (u32)CL_IMG / (4u * std::max(1u, nb)))); }      // (sized for count rows; PA rows need less)
u64 cols_scratch_keys(u32 slots, u32 nblk) { return (u64)slots * CL_HALVES * nblk * CL_NW * CL_OVW * EW; }      // (u64 words: key + payload per entry)
u64 cols_scratch_counts(u32 slots, u32 nblk) { return (u64)slots * CL_HALVES * nblk * CL_NW; }
// the pool the slices' extensions come from: 1/16 of the slices' room (a sample in 300 of three times the cohort's size fills 1/50)
u64 cols_ext_entries(u32 slots, u32 nblk) { return std::max<u64>(64 * 1024, (u64)slots * CL_HALVES * nblk * CL_NW * CL_OVW / 16); }

u32 cols_skel_cap() { return SK_CAP; }
hipError_t launch_cols_skel(const TaskDev* subs, const uint2* items, u32 n_items, hipStream_t st)
{
  hipLaunchKernelGGL(k_cols_skel, dim3(n_items), dim3(SK_TPB), 0, st, subs, items, n_items);
  return hipGetLastError();
}
hipError_t launch_cols_prep(const TaskDev* tasks, const TaskDev* subs, const ColsDev* cols, u32 n_tasks, hipStream_t st)
{
  hipLaunchKernelGGL(k_cols_prep, dim3(n_tasks), dim3(CP_TPB), 0, st, tasks, subs, cols);
  return hipGetLastError();
}
template <int MODE, bool EXT, bool RESC, bool ORD, bool NAR>
static hipError_t launch_merge_cols_as(const TaskDev* tasks, const ColsDev* cols, const uint2* items, u32 n_items, u32* ticket, u32 grid_x, hipStream_t st)
{
  const int lds = cols_lds_bytes();
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_merge_cols<MODE, EXT, RESC, ORD, NAR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_merge_cols<MODE, EXT, RESC, ORD, NAR>), dim3(grid_x), dim3(CL_TPB), lds, st, tasks, cols, items, n_items, ticket);
  return hipGetLastError();
}
// ext: bit 0 = slice extensions (outlier samples), bit 1 = the RESC build (share-min, recurrence-min 0), bit 2 = the ORD build (the row
// keys' rows into the side store: every task of the batch has one), bit 3 = NAR (count rows, ORD: every task's side store is the byte-wide one)
hipError_t launch_merge_cols(int mode, int ext, const TaskDev* tasks, const ColsDev* cols, const uint2* items, u32 n_items, u32* ticket, u32 grid_x, hipStream_t st)
{
  const bool x = ext & 1, r = (ext & 2) != 0, o = (ext & 4) != 0, n = (ext & 8) != 0 && o && mode == 0;
#define KMX_CL_L3(M, X, R) (o ? launch_merge_cols_as<M, X, R, true, false>(tasks, cols, items, n_items, ticket, grid_x, st) : launch_merge_cols_as<M, X, R, false, false>(tasks, cols, items, n_items, ticket, grid_x, st))
#define KMX_CL_LAUNCH(M) (r ? (x ? KMX_CL_L3(M, true, true) : KMX_CL_L3(M, false, true)) : (x ? KMX_CL_L3(M, true, false) : KMX_CL_L3(M, false, false)))
  if (n) return r ? (x ? launch_merge_cols_as<0, true, true, true, true>(tasks, cols, items, n_items, ticket, grid_x, st) : launch_merge_cols_as<0, false, true, true, true>(tasks, cols, items, n_items, ticket, grid_x, st))
                  : (x ? launch_merge_cols_as<0, true, false, true, true>(tasks, cols, items, n_items, ticket, grid_x, st) : launch_merge_cols_as<0, false, false, true, true>(tasks, cols, items, n_items, ticket, grid_x, st));
  return mode == 0 ? KMX_CL_LAUNCH(0) : KMX_CL_LAUNCH(1);
#undef KMX_CL_L3
#undef KMX_CL_LAUNCH
}
template <int MODE, bool RESC>
static hipError_t launch_cols_sparse_ord(const TaskDev* tasks, const ColsDev* cols, u32 n_tasks, u32* ticket, u32 n_cu, hipStream_t st)
{
  // persistent: as many workgroups as the device holds at once (a slice group per ticket)
  static int per_cu = 0;
  if (per_cu == 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&k_cols_sparse<MODE, RESC, true>), ck_tpb<MODE>(), 0) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 1; }
    per_cu = n;
  }
  hipLaunchKernelGGL((k_cols_sparse<MODE, RESC, true>), dim3(n_cu * (u32)per_cu), dim3(ck_tpb<MODE>()), 0, st, tasks, cols, (const uint2*)nullptr, 0u, n_tasks, ticket);
  return hipGetLastError();
}
// mode: bit 0 = PA rows, bit 1 = the RESC build, bit 2 = rows at their final place (ORD: persistent, tickets from ticket[1])
hipError_t launch_cols_sparse(int mode, const TaskDev* tasks, const ColsDev* cols, const uint2* range_items, u32 n_items, u32 n_tasks, u32* ticket, u32 n_cu, hipStream_t st)
{
  if (mode & 4) {
    switch (mode & 3) {
      case 0: return launch_cols_sparse_ord<0, false>(tasks, cols, n_tasks, ticket, n_cu, st);
      case 1: return launch_cols_sparse_ord<1, false>(tasks, cols, n_tasks, ticket, n_cu, st);
      case 2: return launch_cols_sparse_ord<0, true>(tasks, cols, n_tasks, ticket, n_cu, st);
      default: return launch_cols_sparse_ord<1, true>(tasks, cols, n_tasks, ticket, n_cu, st);
    }
  }
  const dim3 grid(n_items, CK_Z), block(ck_tpb<0>()), block1(ck_tpb<1>());
  switch (mode & 3) {
    case 0: hipLaunchKernelGGL((k_cols_sparse<0, false, false>), grid, block, 0, st, tasks, cols, range_items, n_items, n_tasks, ticket); break;
    case 1: hipLaunchKernelGGL((k_cols_sparse<1, false, false>), grid, block1, 0, st, tasks, cols, range_items, n_items, n_tasks, ticket); break;
    case 2: hipLaunchKernelGGL((k_cols_sparse<0, true, false>), grid, block, 0, st, tasks, cols, range_items, n_items, n_tasks, ticket); break;
    default: hipLaunchKernelGGL((k_cols_sparse<1, true, false>), grid, block1, 0, st, tasks, cols, range_items, n_items, n_tasks, ticket); break;
  }
  return hipGetLastError();
}
u64 cols_dir_bytes(u32 slots) { return (u64)slots * CL_HALVES * CK_NPASS * sizeof(SpDir); }
u32 cols_groups(u32 slots) { return slots * CL_HALVES; }
hipError_t launch_cols_offsets(const ColsDev* cols, u32 task, u64* goff, hipStream_t st)
{
  hipLaunchKernelGGL(k_cols_offsets, dim3(1), dim3(1024), 0, st, cols, task, goff);
  return hipGetLastError();
}
hipError_t launch_cols_gather(const TaskDev* tasks, const ColsDev* cols, u32 task, u32 n_groups, const u64* goff, u8* body, hipStream_t st)
{
  hipLaunchKernelGGL(k_cols_gather<true>, dim3(n_groups), dim3(GA_TPB), 0, st, tasks, cols, task, goff, body);
  return hipGetLastError();
}
hipError_t launch_cols_order(const TaskDev* tasks, const ColsDev* cols, u32 task, u32 n_groups, const u64* goff, u32* order, hipStream_t st)
{
  hipLaunchKernelGGL(k_cols_gather<false>, dim3(n_groups), dim3(GA_TPB), 0, st, tasks, cols, task, goff, reinterpret_cast<u8*>(order));
  return hipGetLastError();
}

static const ColsOps g_ops = {cols_lds_bytes, cols_block_lists, cols_wgs_per_cu, cols_tile_rows, cols_scratch_keys, cols_scratch_counts, cols_ext_entries, cols_skel_cap,
                              launch_cols_skel, launch_cols_prep, launch_merge_cols, launch_cols_sparse, cols_dir_bytes, cols_groups, launch_cols_offsets,
                              launch_cols_gather, launch_cols_order, cols_dbg_dump,
#ifdef KMX_PHASE_PROF
                              cols_phase_prof_dump,
#else
                              nullptr,
#endif
                              (u32)KMX_CL_KW};
}  // namespace CLNS
#if KMX_CL_KW == 1
const ColsOps& cols_ops_k1() { return cols_k1::g_ops; }
#elif KMX_CL_KW == 2
const ColsOps& cols_ops_k2() { return cols_k2::g_ops; }
#elif KMX_CL_KW == 3
const ColsOps& cols_ops_k3() { return cols_k3::g_ops; }
#else
const ColsOps& cols_ops_k4() { return cols_k4::g_ops; }
#endif
}  // namespace kmx
