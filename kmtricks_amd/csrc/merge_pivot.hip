// merge_pivot.hip -- pivot-tiled streaming merge for related samples (COUNT / PA rows) on gfx950.
// Same results as k_merge_rows (reference include/kmtricks/merge.hpp:183-286, 441-558), different
// decomposition, built for the case the metric is quoted on: many lists that share most of their keys.
//
//   * The task's longest list is the PIVOT.  A tile is rt consecutive pivot records; its key range
//     [first key of the tile, key of the pivot record after the tile) is known before any other list
//     is looked at.  A list whose whole window lies below that key (it is denser than the pivot here)
//     cuts the tile at its last window key, so everything below the tile's limit is always inside
//     every list's window: no second round, no retry.
//   * Every list keeps a circular window of 16 records in registers (4 adjacent lanes x 4 slots, a
//     lane serves up to 4 lists).  A record below the tile's upper key is consumed and its slot is
//     refilled IN PLACE with the record 16 positions further: every record is loaded exactly once
//     (one global_load_dwordx3 straight into the slot), and the refill has a whole tile to land.
//   * A consumed record whose key IS one of the tile's pivot keys (one probe of a read-only 128-entry
//     LDS table built per tile: no atomics, same-address reads broadcast) is deposited straight into
//     row j of the tile's LDS image.  The recurrence of a pivot row is a popcount of its image row.
//   * A record whose key is NOT a pivot key (sample-private k-mers, keys the pivot lacks) goes to an
//     LDS overflow buffer; after the scan the (few) overflow records are merged among themselves
//     with the hash set of k_merge_rows, kept keys are ranked together with the kept pivot rows, and
//     their (sparse) rows are written straight to HBM.
//   * A key the pivot lacks but at least two of four helper lists have is adopted as an extra image
//     row before the scan (a key missing from the pivot is usually present in ~all other lists).
//   * If the overflow does not fit, or more than 1/8 of the records are not row keys (lists that do
//     not resemble each other), the task is flagged and the driver re-runs the batch with
//     k_merge_rows -- results never depend on how well the pivot covers the other lists.
// Rows leave through the same chunked arena + (range, seq) directory as k_merge_rows.
#include "kmx_dev.hpp"
#include <algorithm>
#include <cstdio>
#include <cstring>

namespace kmx {

#ifndef KMX_PV_RT
#define KMX_PV_RT 13
#endif
#ifndef KMX_PV_TPB
#define KMX_PV_TPB 1024
#endif
constexpr int PV_TPB = KMX_PV_TPB;  // 512: 8 waves with 256 VGPRs each -- the windows of 16 lists per lane plus room to overlap two passes
constexpr int PV_RTMAX = KMX_PV_RT; // pivot records per tile (< window: a similar list rarely needs a second round)
constexpr int PV_IMG = 61440;       // LDS row image bytes (15 rows of 1000 u32 counts)
constexpr int PV_OVCAP = 2048;      // overflow records per tile
constexpr int PV_OT = 2 * PV_OVCAP; // overflow hash set entries
constexpr int PV_OVW = PV_OVCAP / (KMX_PV_TPB / 64);   // ... of which every wave owns a private slice (no atomics to append)
#ifndef KMX_PV_G
#define KMX_PV_G 4
#endif
constexpr int PV_G = KMX_PV_G;      // adjacent lanes per list: one wave load covers 64/G lists x 12G contiguous bytes
constexpr int PV_U = 16 / PV_G;     // window slots per lane
constexpr int PV_W = PV_G * PV_U;   // records per window (power of two)
constexpr int PV_LPP = PV_TPB / PV_G;   // lists per pass
constexpr int PV_MAXN = 1024;       // lists per task
constexpr int PV_NP = PV_MAXN / PV_LPP;   // passes
constexpr int PV_NH = 8;             // helper lists
constexpr int PV_HSTRIDE = 64 / PV_G;   // lists per wave and pass: helpers sit one wave apart
constexpr int PV_NC = 64;            // candidate keys per tile (one per lane of wave 0)
constexpr int PV_PT = 128;          // pivot lookup table entries (load factor <= 1/8)
static_assert(PV_RTMAX <= 16 && (PV_W & (PV_W - 1)) == 0, "tile geometry");

// a 12-byte record as one register triple: the refill is ONE global_load_dwordx3 straight into the
// window slot (key and count allocated apart cost a register move, i.e. a wait, right behind the load)
typedef u32 u32x3 __attribute__((ext_vector_type(3), aligned(4)));
typedef __attribute__((address_space(1))) const u32x3 gu32x3;
__device__ __forceinline__ Key<1> rec_key(const u32x3& v) { Key<1> k; k.w[0] = (u64)v.x | ((u64)v.y << 32); return k; }
__device__ __forceinline__ u32x3 rec_none() { u32x3 v; v.x = ~0u; v.y = ~0u; v.z = 0; return v; }

template <int KW> struct OvRec { Key<KW> key; u32 cnt; u32 list; };
template <int KW> struct PvEnt { Key<KW> key; u32 idx; u32 pad; };   // idx = row + 1, 0 = empty

// values every lane reads from the same LDS word are uniform, but only a readfirstlane tells the compiler: they
// then live in scalar registers instead of occupying (and spilling) vector registers across the scan
__device__ __forceinline__ u32 pv_uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 pv_uni64(u64 v) { return (u64)pv_uni((u32)v) | ((u64)pv_uni((u32)(v >> 32)) << 32); }

// a value the compiler must re-derive here: what is computed from it is not hoisted out of the tile loop (hoisted
// addresses get spilled, and a scratch reload waits for every record load in flight)
__device__ __forceinline__ int pv_fresh(int v) { asm volatile("" : "+v"(v)); return v; }

__device__ __forceinline__ void pv_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int KW> __device__ __forceinline__ u32 pv_mix(const Key<KW>& k)
{
  u32 x = (u32)k.w[0] ^ ((u32)(k.w[0] >> 32) * 0x9E3779B1u);
  if (KW == 2) x ^= ((u32)k.w[KW - 1] * 0x85EBCA77u) ^ ((u32)(k.w[KW - 1] >> 32) * 0xC2B2AE3Du);
  x *= 0x85EBCA6Bu; x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 13;
  return x;
}
template <int KW> __device__ __forceinline__ u32 pv_hash(const Key<KW>& k) { return pv_mix<KW>(k) & (PV_OT - 1); }
// the tile's pivot keys are neighbours in key order: a multiplicative hash of the low bits spreads them
template <int KW> __device__ __forceinline__ u32 pv_thash(const Key<KW>& k)
{ return (__umul24(((u32)k.w[0] ^ (u32)(k.w[0] >> 23)) & 0xFFFFFFu, 0x9E3779u) >> 15) & (PV_PT - 1); }

template <int KW> __device__ __forceinline__ Key<KW> gload_key(gu32* p)
{
  Key<KW> k;
#pragma unroll
  for (int q = 0; q < KW; q++) k.w[q] = (u64)p[2 * q] | ((u64)p[2 * q + 1] << 32);
  return k;
}

// probing past a collision in the pivot table (rare): row + 1 of `k`, 0 if it is not a pivot key
template <int KW>
__device__ __noinline__ u32 pv_lookup_slow(const PvEnt<KW>* ptab, Key<KW> k, u32 h)
{
  for (;;) {
    h = (h + 1) & (PV_PT - 1);
    const u32 idx = ptab[h].idx;
    if (idx == 0) return 0;
    if (key_eq<KW>(ptab[h].key, k)) return idx;
  }
}

#ifdef KMX_PHASE_PROF
__device__ u64 kmx_pivot_prof[16];
#ifndef KMX_PROF_TID
#define KMX_PROF_TID 0      // the thread whose clock64 deltas are summed (wave 0 does the serial work; try 512 for a typical wave)
#endif
#endif

template <int KW, int MODE>
__global__ __launch_bounds__(PV_TPB, PV_TPB / 256)
void k_merge_pivot(const TaskDev* __restrict__ tasks, const uint2* __restrict__ items, u32 n_items, u32* ticket)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int RB4 = (KW * 8 + 4) / 4;
  static_assert(KW == 1, "the register windows hold 12-byte records");
  unsigned char* const img = smem;                                                        // [PV_IMG]
  OvRec<KW>* ov = reinterpret_cast<OvRec<KW>*>(smem + PV_IMG);                            // [PV_OVCAP]
  u32* otab = reinterpret_cast<u32*>(smem + PV_IMG + PV_OVCAP * sizeof(OvRec<KW>));       // [PV_OT]
  unsigned char* misc = smem + PV_IMG + PV_OVCAP * sizeof(OvRec<KW>) + PV_OT * 4;
  Key<KW>* pk = reinterpret_cast<Key<KW>*>(misc);                                         // [32] keys of the image rows (pivot rows, then adopted rows)
  u32* prec = reinterpret_cast<u32*>(misc + 256);                                         // [32] recurrence
  u32* prank = reinterpret_cast<u32*>(misc + 384);                                        // [32] final row or ~0
  u32* sh = reinterpret_cast<u32*>(misc + 512);                                           // [0] item [1] overflow records [2] nok [3] can-write [4] adopted rows [5] candidates [6] records consumed [7] task error bits [8] some overflow key may be kept [9] fullest overflow slice
  u64* sh64 = reinterpret_cast<u64*>(misc + 576);                                         // [0] tile row base [1] upper key [2] cut
  u32* wcnt = reinterpret_cast<u32*>(misc + 1152);                                        // [waves] overflow records in each wave's slice
  u64* cand = reinterpret_cast<u64*>(misc + 640);                                        // [PV_NC] helper keys the pivot lacks
  PvEnt<KW>* ptab = reinterpret_cast<PvEnt<KW>*>(misc + 1280);                             // [PV_PT] row key -> row
  unsigned char* m2 = misc + 1280 + PV_PT * sizeof(PvEnt<KW>);
  u16* okl = reinterpret_cast<u16*>(m2);                                                  // [PV_OVCAP] table slots of kept overflow keys
  u16* orank = reinterpret_cast<u16*>(m2 + PV_OVCAP * 2);                                 // [PV_OVCAP] their final rows
  // per-list state (a list is served by 8 lanes, a lane serves up to 8 lists)
  unsigned char* lt = m2 + PV_OVCAP * 4;
  u64* lt_base = reinterpret_cast<u64*>(lt);                    // [1024] record base
  uint4* lt_st = reinterpret_cast<uint4*>(lt + 8192);           // [1024] x: range end, y: soft-min, z: cursor (first record not yet consumed)
  u64* lt_two = reinterpret_cast<u64*>(lt + 24576);             // [1024] TOTAL_WO of the range so far
  u32* lt_nso = reinterpret_cast<u32*>(lt + 32768);             // [1024] NON_SOLID so far
  // row-space allocator state (thread 0 only; kept out of the register file)
  u64* al64 = reinterpret_cast<u64*>(misc + 608);               // [0] chunk base
  u32* al = reinterpret_cast<u32*>(misc + 616);                 // [0] used [1] cap [2] seq [3] ok

  const int tid = threadIdx.x, lane = tid & 63;
  for (int t = tid; t < PV_OT; t += PV_TPB) otab[t] = 0;
  for (int t = tid; t < PV_PT; t += PV_TPB) ptab[t].idx = 0;
#ifdef KMX_PHASE_PROF
  long long pt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long pc = clock64();
#define PVPH(i) do { const long long n_ = clock64(); pt[i] += n_ - pc; pc = n_; } while (0)
#else
#define PVPH(i) do {} while (0)
#endif

  for (;;) {
    if (tid == 0) sh[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    const u32 item = pv_uni(sh[0]);      // scalar: the task descriptor is then read with scalar loads into SGPRs
    __syncthreads();
    if (item >= n_items) {
#ifdef KMX_PHASE_PROF
      if (tid == KMX_PROF_TID) for (int i = 0; i < 16; i++) atomicAdd(&kmx_pivot_prof[i], (u64)pt[i]);
#endif
      return;
    }
    const TaskDev& T = tasks[items[item].x];
    const u32 range = items[item].y;
    const u32 N = T.N, rec_min = T.rec_min, share_min = T.share_min, row_bytes = T.row_bytes;
    const u32 sat = max(rec_min, share_min);
    const u32 chunk_rows = max(64u, (u32)KMX_CHUNK_BYTES / row_bytes);
    const u32 rows_cap = max(1u, min(32u, (u32)PV_IMG / row_bytes));      // image rows
    const u32 rt_cap = min((u32)PV_RTMAX, rows_cap);                      // pivot rows per tile; the rest is for adopted rows
    if (__syncthreads_or((__hip_atomic_load(&T.ctrl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (u64)ERR_FALLBACK) != 0)) continue;   // the batch is re-run anyway (one decision per workgroup: the word is raised while it is read)

    // PV_G adjacent lanes per list; list p*PV_LPP + tid/PV_G in pass p
    const u32 npass = (N + PV_LPP - 1) / PV_LPP;
    const u32 lg = tid / PV_G, r = tid & (PV_G - 1);
    for (u32 i = tid; i < N; i += PV_TPB) {
      lt_base[i] = (u64)(uintptr_t)T.recs[i];
      lt_st[i] = make_uint4(T.bounds[(u64)(range + 1) * N + i], T.soft_min[i], T.bounds[(u64)range * N + i], 0u);
      lt_nso[i] = 0; lt_two[i] = 0;
    }
    // circular windows: lane r, slot u of a list holds the record whose index is == r + 8u (mod 16)
    // inside [cur, cur + 16)
    u32x3 rec[PV_NP][PV_U];
#pragma unroll
    for (int p = 0; p < PV_NP; p++) {
      const u32 li = p * PV_LPP + lg;
      const bool on = (u32)p < npass && li < N;
      const u32 c0 = on ? T.bounds[(u64)range * N + li] : 0;
      const u32 e = on ? T.bounds[(u64)(range + 1) * N + li] : 0;
      gu32* base = (gu32*)(uintptr_t)(on ? T.recs[li] : nullptr);
#pragma unroll
      for (int u = 0; u < PV_U; u++) {
        const u32 ix = c0 + ((r + PV_G * u - c0) & (PV_W - 1));
        rec[p][u] = rec_none();
        if (ix < e) rec[p][u] = *(gu32x3*)(base + (u64)ix * RB4);
      }
    }
    // the pivot's records of this range define the tiles; a tile's keys are fetched one tile ahead
    gu32* pbase = (gu32*)(uintptr_t)pv_uni64((u64)(uintptr_t)T.recs[T.pivot]);   // scalar: it lives across every tile
    const u32 pend = T.bounds[(u64)(range + 1) * N + T.pivot];
    u32 ppos = T.bounds[(u64)range * N + T.pivot];
    Key<KW> pkn = key_inf<KW>();
    if ((u32)tid <= rt_cap && ppos + tid < pend) pkn = gload_key<KW>(pbase + (u64)(ppos + tid) * RB4);
    // A few more lists lend their keys: a key the pivot lacks is usually present in ~all other lists, and
    // ~N overflow records per missing key would swamp the overflow buffer.  A key below the tile's
    // upper key that the pivot does not have but at least two of the helper lists do is ADOPTED as an
    // extra image row (one helper alone would mostly contribute its private keys); when the rows run
    // out, the tile is cut in front of the first key that did not get one.
    // helpers = the first PV_NH lists that are not the pivot (all served in pass 0)
    // helpers: every PV_HSTRIDE-th list of the first pass (one per wave, so the candidate probes run in parallel
    // instead of all landing on wave 0), the pivot excepted
    const u32 hmax = min(N, (u32)(PV_NH + 1) * PV_HSTRIDE);
    if (tid == 0) { al64[0] = 0; al[0] = 0; al[1] = 0; al[2] = 0; al[3] = 1; }
    u32 seq = 0, ovsum = 0, conssum = 0;
    u32 hb = 0;                          // thread 0: the task's error word as of the start of the tile
    u32 rt_cur = rt_cap;                 // pivot rows per tile: shrinks where the lists carry many keys that are not row keys
    bool failed = false;
    __syncthreads();

    PVPH(0);
    for (;;) {
      // ---- tile = pivot records [ppos, ppos + rte); key range up to the next pivot key ----
      const int tidA = pv_fresh(tid);
      const u32 rte = min(rt_cur, pend - ppos);
      const bool open_end = ppos + rte >= pend;     // last tile of the range: bounded by the lists' range ends
      if ((u32)tidA < rte) {
        pk[tidA] = pkn;
        u32 h = pv_thash<KW>(pkn);
        while (atomicCAS(&ptab[h].idx, 0u, (u32)tidA + 1) != 0) h = (h + 1) & (PV_PT - 1);
        ptab[h].key = pkn;
      }
      if ((u32)tidA == rte) { sh64[1] = pkn.w[0]; sh64[2] = open_end ? ~0ULL : pkn.w[0]; }   // upper key; exclusive limit (cuts lower it)
      {
        const u32 zb = rows_cap * row_bytes;
        uint4* z = reinterpret_cast<uint4*>(img);
        for (u32 t = tidA; t < (zb + 15) / 16; t += PV_TPB) z[t] = make_uint4(0, 0, 0, 0);
        if (tidA == 0) { sh[1] = 0; sh[2] = 0; sh[4] = 0; sh[5] = 0; sh[6] = 0; sh[8] = rec_min <= 1 ? 1u : 0u; sh[9] = 0; }
        // hand-back flag of the task: loaded now, stored to LDS only behind the scan, so nobody waits for the load
        if (tidA == 0) hb = (u32)__hip_atomic_load(&T.ctrl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      pv_lds_barrier();
      PVPH(1);
      Key<KW> khi; khi.w[0] = open_end ? ~0ULL : pv_uni64(sh64[1]);
      if ((u32)tidA < rte) {   // row keys
        const Key<KW> mine = pkn;
        u8* row = img + tidA * row_bytes;
        if (MODE == 0) { u32* rw = reinterpret_cast<u32*>(row);
#pragma unroll
          for (int q = 0; q < KW; q++) { rw[2 * q] = (u32)mine.w[q]; rw[2 * q + 1] = (u32)(mine.w[q] >> 32); } }
        else {
#pragma unroll
          for (int q = 0; q < KW * 8; q++) row[q] = (u8)(mine.w[q >> 3] >> ((q & 7) * 8));
        }
      }
      // ---- window check: a list whose whole window lies below khi with records behind it cuts the tile
      //      at its last window key (everything up to that key is inside every list's window) ----
      uint4 stn = lg < N ? lt_st[lg] : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int p = 0; p < PV_NP; p++) {
        __builtin_amdgcn_sched_barrier(0);
        if ((u32)p < npass) {
          u32 lgx = lg; asm volatile("" : "+v"(lgx));
          const u32 li = p * PV_LPP + lgx;
          const uint4 st = stn;                          // this pass's list state was read one pass ahead
          if (p + 1 < PV_NP && (u32)(p + 1) < npass) stn = li + PV_LPP < N ? lt_st[li + PV_LPP] : make_uint4(0, 0, 0, 0);
          const u32 end = st.x, cur = st.z;
          // lists are sorted: the whole window lies below khi iff its LAST record does -- one compare in the
          // one slot that holds it, no ballots
#pragma unroll
          for (int u = 0; u < PV_U; u++) {
            const u32 ix = cur + ((r + PV_G * u - cur) & (PV_W - 1));
            const Key<KW> k = rec_key(rec[p][u]);
            const bool below = ix < end && (open_end || key_less<KW>(k, khi));
            if (below && ix == cur + PV_W - 1 && cur + PV_W < end) atomicMin(&sh64[2], k.w[0] + 1);
            if (p == 0 && below && li < hmax && (li % PV_HSTRIDE) == 0 && li != T.pivot) {   // a helper's key the pivot lacks: candidate row
              u32 h = pv_thash<KW>(k);
              u32 idx = ptab[h].idx;
              if (idx != 0 && !key_eq<KW>(ptab[h].key, k)) idx = pv_lookup_slow<KW>(ptab, k, h);
              if (idx == 0) { const u32 pos = atomicAdd(&sh[5], 1u); if (pos < (u32)PV_NC) cand[pos] = k.w[0]; }
            }
          }
        }
      }
      PVPH(9);
      pv_lds_barrier();
      PVPH(10);
      {   // candidates seen in >= 2 helpers become rows, smallest keys first; the first one left without a row cuts the
          // tile.  Wave 0 alone: one candidate per lane, the others read with v_readlane.
        const u32 nc = min(pv_uni(sh[5]), (u32)PV_NC);
        if (nc && tid < 64) {
          const u64 mine = (u32)lane < nc ? cand[lane] : ~0ULL;
          u32 cnt = 0, earlier = 0;
          for (u32 j = 0; j < nc; j++) {
            const u64 o = (u64)(u32)__builtin_amdgcn_readlane((int)(u32)mine, (int)j) | ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(mine >> 32), (int)j) << 32);
            const bool e = o == mine;
            cnt += e ? 1u : 0u; earlier += (e && j < (u32)lane) ? 1u : 0u;
          }
          const bool qual = (u32)lane < nc && cnt >= 2 && earlier == 0;
          const u64 qmask = __ballot(qual);
          u32 rank = 0;    // qualifying distinct keys below mine
          for (u32 j = 0; j < nc; j++) {
            const u64 o = (u64)(u32)__builtin_amdgcn_readlane((int)(u32)mine, (int)j) | ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(mine >> 32), (int)j) << 32);
            rank += (((qmask >> j) & 1ULL) && o < mine) ? 1u : 0u;
          }
          if (qual) {
            const u32 row = rte + rank;
            if (row < rows_cap) {
              Key<KW> k; k.w[0] = mine;
              u32 h2 = pv_thash<KW>(k);
              while (atomicCAS(&ptab[h2].idx, 0u, row + 1) != 0) h2 = (h2 + 1) & (PV_PT - 1);
              ptab[h2].key = k;
              pk[row] = k;
              atomicMax(&sh[4], rank + 1);
              u8* rowp = img + row * row_bytes;
              if (MODE == 0) { u32* rw = reinterpret_cast<u32*>(rowp);
#pragma unroll
                for (int q = 0; q < KW; q++) { rw[2 * q] = (u32)k.w[q]; rw[2 * q + 1] = (u32)(k.w[q] >> 32); } }
              else {
#pragma unroll
                for (int q = 0; q < KW * 8; q++) rowp[q] = (u8)(k.w[q >> 3] >> ((q & 7) * 8));
              }
            } else atomicMin(&sh64[2], mine);
          }
        }
      }
      PVPH(11);
      pv_lds_barrier();
      PVPH(12);
      const u64 lim = pv_uni64(sh64[2]);           // exclusive upper key of the tile
      const bool unbounded = lim == ~0ULL;         // open-ended tile, not cut
      const u32 nrows = rte + min(pv_uni(sh[4]), rows_cap - rte);
      // pivot rows below the limit (lane j looks at row j: one LDS read per lane, not rte per lane)
      const u32 rte_eff = (u32)__popcll(__ballot((u32)lane < rte && (unbounded || pk[lane < 32 ? lane : 0].w[0] < lim)));
      const bool done = unbounded;
      {   // next tile's pivot keys
        const u32 npos = ppos + rte_eff;
        pkn = key_inf<KW>();
        if (!done && (u32)tid <= rt_cap && npos + tid < pend) pkn = gload_key<KW>(pbase + (u64)(npos + tid) * RB4);
      }
      PVPH(13);

      // ---- scan: every list consumes its records of the tile (keys below lim) ----
      stn = lg < N ? lt_st[lg] : make_uint4(0, 0, 0, 0);
      u32 lcons = 0, wov = 0;                   // records consumed by this lane / overflow records of this wave (uniform)
#pragma unroll
      for (int p = 0; p < PV_NP; p++) {
        __builtin_amdgcn_sched_barrier(0);      // keep the passes apart: interleaving them only costs registers
        if ((u32)p < npass) {
          // li is recomputed here on purpose: hoisted out of the tile loop, the per-pass LDS addresses
          // spill, and a scratch reload has to wait for every record load in flight
          u32 lgx = lg; asm volatile("" : "+v"(lgx));
          const u32 li = p * PV_LPP + lgx;
          const uint4 st = stn;
          if (p + 1 < PV_NP && (u32)(p + 1) < npass) stn = li + PV_LPP < N ? lt_st[li + PV_LPP] : make_uint4(0, 0, 0, 0);
          const u32 end = st.x, smin = st.y, cur = st.z;
          u32 consm = 0, ovm = 0, tn = 0; u64 tsum = 0;
          // both slots probe the row table before either result is looked at
          uint4 pe[PV_U];
#pragma unroll
          for (int u = 0; u < PV_U; u++) pe[u] = reinterpret_cast<const uint4*>(ptab)[pv_thash<KW>(rec_key(rec[p][u]))];
#pragma unroll
          for (int u = 0; u < PV_U; u++) {
            // straight-line on purpose: one table probe (16-byte entry in one LDS read) for every lane,
            // masks instead of nested branches
            const u32 ix = cur + ((r + PV_G * u - cur) & (PV_W - 1));
            const Key<KW> k = rec_key(rec[p][u]);
            const bool cons = ix < end && (unbounded || k.w[0] < lim);
            const u32 c = rec[p][u].z;
            const bool solid = cons && c >= smin;
            const uint4 e = pe[u];                                       // key lo, key hi, idx, pad
            u32 idx = e.z;
            if (((e.x ^ (u32)k.w[0]) | (e.y ^ (u32)(k.w[0] >> 32))) != 0 && idx != 0) idx = pv_lookup_slow<KW>(ptab, k, pv_thash<KW>(k));
            consm |= (cons ? 1u : 0u) << u;
            ovm |= ((cons && idx == 0) ? 1u : 0u) << u;
            tsum += solid ? c : 0u;
            tn += (cons && !solid) ? 1u : 0u;
            if (solid && idx != 0) {
              const u32 row = idx - 1;
              if (MODE == 0) reinterpret_cast<u32*>(img + __umul24(row, row_bytes) + KW * 8)[li] = c;
              else { const u32 ob = __umul24(row, row_bytes) + KW * 8 + (li >> 3);
                     atomicOr(reinterpret_cast<u32*>(img) + (ob >> 2), 1u << (((ob & 3u) << 3) + (li & 7u))); }
            }
          }
          // overflow records of this pass go to the wave's own slice of the buffer: positions from ballots,
          // the running count stays in a scalar register -- no atomic, no LDS round trip
          {
#pragma unroll
            for (int u = 0; u < PV_U; u++) {
              const u64 bal = __ballot((ovm >> u) & 1u);
              if ((ovm >> u) & 1u) {
                const u32 pos = wov + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
                if (pos < (u32)PV_OVW) { OvRec<KW> o; o.key = rec_key(rec[p][u]); o.cnt = rec[p][u].z; o.list = li; ov[(tid >> 6) * PV_OVW + pos] = o; }
              }
              wov += __popcll(bal);
            }
          }
          // records consumed by my list (a prefix of its window): statistics and the new cursor
          // (count = the list's group sum of per-lane counts: lane-crossing adds, no ballot / scalar round trips)
          u32 c = __popc(consm);
          lcons += c;
          if (PV_G == 4) {   // quad sum with DPP lane permutes: no LDS round trip on the way to the refill addresses
            c += (u32)__builtin_amdgcn_update_dpp(0, (int)c, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
            c += (u32)__builtin_amdgcn_update_dpp(0, (int)c, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
          } else {
#pragma unroll
            for (int off = 1; off < PV_G; off <<= 1) c += __shfl_xor(c, off);
          }
          if (consm) {
            if (tsum) atomicAdd(&lt_two[li], tsum);
            if (tn) atomicAdd(&lt_nso[li], tn);
          }
          if (r == 0 && c) reinterpret_cast<u32*>(lt_st + li)[2] = cur + c;
          // refill in place: a consumed slot takes the record 16 positions further.  Every window slot
          // was touched (and its pending load waited for) by the window sweep, so these loads do not
          // stall the passes behind this one; they land while the rest of the tile is processed.
          if (consm) {
            gu32* base = (gu32*)(uintptr_t)lt_base[li];
            const u32 ncur = cur + c;
#pragma unroll
            for (int u = 0; u < PV_U; u++) {
              if ((consm >> u) & 1u) {
                const u32 tix = ncur + ((r + PV_G * u - ncur) & (PV_W - 1));    // the slot's record in the new window
                rec[p][u] = rec_none();
                if (tix < end) rec[p][u] = *(gu32x3*)(base + (u64)tix * RB4);
              }
            }
          }
        }
      }
      {
        // wave sum of the per-lane counts (<= 16 each) bit by bit with ballots: scalar work, no cross-lane LDS traffic
        u32 wcons = 0;
#pragma unroll
        for (int b = 0; b < 5; b++) wcons += (u32)__popcll(__ballot((lcons >> b) & 1u)) << b;
        if (lane == 0) { if (wcons) atomicAdd(&sh[6], wcons); wcnt[tid >> 6] = wov; if (wov) { atomicAdd(&sh[1], wov); atomicMax(&sh[9], wov); } }
        if (tid == 0) sh[7] = hb;
      }
      PVPH(8);
      PVPH(14);
      pv_lds_barrier();
      PVPH(15);
      const int tidB = pv_fresh(tid), laneB = tidB & 63;
      const u32 ovn = pv_uni(sh[1]), ovmax = pv_uni(sh[9]);     // overflow records of the tile, fullest slice
      ovsum += ovn; conssum += pv_uni(sh[6]);
      const bool unfit = ovmax > (u32)PV_OVW || (conssum > 65536u && ovsum * 8u > conssum);
      if (unfit || (pv_uni(sh[7]) & (u32)ERR_FALLBACK)) {   // ... or another workgroup already handed this task back
        // the rows do not cover the other lists here (overflow buffer full, or more than 1/8 of the
        // range's records so far are not row keys: lists that do not resemble each other --
        // k_merge_rows does better there): flag the task, the driver re-runs the batch with
        // k_merge_rows.  Leave the tables clean for the next work item.
        for (int t = tidB; t < PV_PT; t += PV_TPB) ptab[t].idx = 0;
        failed = unfit;
        break;
      }
      // the fuller the fullest overflow slice, the fewer pivot rows the next tile takes (and back up when it empties)
      if (ovmax > (u32)PV_OVW * 5 / 8) rt_cur = max(3u, rt_cur - 3u);
      else if (ovmax < (u32)PV_OVW / 4 && rt_cur < rt_cap) rt_cur++;

      for (int t = tidB; t < PV_PT; t += PV_TPB) ptab[t].idx = 0;   // the row table is dead after the scan: clean for the next tile
      // recurrence of the image rows = number of lists that deposited a (solid) count: wave j counts row j
      {
        // (wave index recomputed here: as a loop invariant the row addresses derived from it get spilled, and
        //  a scratch reload right behind the scan waits for every refill load in flight)
        u32 wv = pv_uni((u32)tidB >> 6); asm volatile("" : "+s"(wv));
        u32 ln = laneB; asm volatile("" : "+v"(ln));
        for (u32 j = wv; j < nrows; j += PV_TPB / 64) {
          // ballots + scalar popcounts: no cross-laneB reduction (its LDS round trips cost more than the row)
          u32 nz = 0;
          if (MODE == 0) {
            const u32* rowc = reinterpret_cast<const u32*>(img + j * row_bytes + KW * 8);   // 8-byte aligned: row_bytes = 8 + 4N
            if ((row_bytes & 7u) == 0) {
              const uint2* row2 = reinterpret_cast<const uint2*>(rowc);
              for (u32 t0 = 0; t0 < N / 2; t0 += 256) {      // four reads in flight per lane
                uint2 v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) { const u32 t = t0 + 64 * q + ln; v[q] = make_uint2(0, 0); if (t < N / 2) v[q] = row2[t]; }
#pragma unroll
                for (int q = 0; q < 4; q++) nz += __popcll(__ballot(v[q].x != 0)) + __popcll(__ballot(v[q].y != 0));
              }
            } else {
              for (u32 t0 = 0; t0 < N; t0 += 64) { const u32 t = t0 + ln; nz += __popcll(__ballot(t < N && rowc[t] != 0)); }
            }
          } else {
            const u8* rowb = img + j * row_bytes + KW * 8;
            u32 part = 0;
            for (u32 t = ln; t < (N + 7) / 8; t += 64) part += __popc((u32)rowb[t]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
            nz = part;
          }
          if (laneB == 0) prec[j] = nz;
        }
      }
      PVPH(2);   // no barrier here: the overflow insert below does not touch the image or prec
      // ---- overflow records: merge them among themselves (hash set, as k_merge_rows) ----
      constexpr int OQ = PV_OVW / 64;            // a wave merges its own slice: lane l takes slots l, l + 64, ... of it
      const u32 wov_mine = pv_uni(wcnt[(u32)tidB >> 6]);
      u32 hs[OQ]; u32 ownm = 0, solidm = 0;
#pragma unroll
      for (int q = 0; q < OQ; q++) {
        hs[q] = 0;
        const u32 t = ((u32)tidB >> 6) * PV_OVW + (u32)laneB + 64u * q;
        if (64u * q < wov_mine && (u32)laneB + 64u * q < wov_mine) {
          const OvRec<KW> o = ov[t];
          u32 h = pv_hash<KW>(o.key), old;
          for (;;) {
            old = otab[h];
            if (old == 0) { old = atomicCAS(&otab[h], 0u, t + 1); if (old == 0) { ownm |= 1u << q; break; } }
            if (key_eq<KW>(ov[(old & 0xFFFFu) - 1].key, o.key)) break;
            h = (h + 1) & (PV_OT - 1);
          }
          hs[q] = h;
          if (o.cnt >= lt_st[o.list].y) {
            solidm |= 1u << q;
            if ((old >> 16) < sat) { const u32 prev = atomicAdd(&otab[h], 1u << 16); if ((prev >> 16) + 1 >= rec_min) sh[8] = 1; }
          }
        }
      }
      pv_lds_barrier();
      // the usual tile has no overflow key that reaches the recurrence: then nothing is published, ranked or scattered
      const bool anykept = pv_uni(sh[8]) != 0;
      if (anykept) {
#pragma unroll
        for (int q = 0; q < OQ; q++) {
          if ((ownm >> q) & 1u) {
            const u32 e = otab[hs[q]];
            if ((e >> 16) >= rec_min) { const u32 pos = atomicAdd(&sh[2], 1u); okl[pos] = (u16)hs[q]; }
            else otab[hs[q]] = (e & 0xFFFF0000u) | 0xFFFFu;
          }
        }
        pv_lds_barrier();
      }
      PVPH(3);
      const u32 nok = anykept ? pv_uni(sh[2]) : 0u;
      // ---- final row order: kept image rows and kept overflow keys together ----
      u32 nk = 0;
      if (nrows + nok <= 64) {
        // the usual case, done by wave 0 alone: every laneB holds one key, the others are read with
        // v_readlane -- no LDS round trip per comparison
        if (tidB < 64) {
          const u32 it = laneB, n = nrows + nok;
          Key<KW> mine = key_inf<KW>(); bool kept = false;
          if (it < nrows) { mine = pk[it]; kept = prec[it] >= rec_min; }
          else if (it < n) { mine = ov[(otab[okl[it - nrows]] & 0xFFFFu) - 1].key; kept = true; }
          const u64 keptmask = __ballot(kept);
          u32 rk = 0;
          for (u32 j = 0; j < n; j++) {
            const u64 kj = (u64)(u32)__builtin_amdgcn_readlane((int)(u32)mine.w[0], (int)j) | ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(mine.w[0] >> 32), (int)j) << 32);
            rk += (((keptmask >> j) & 1ULL) && kj < mine.w[0]) ? 1u : 0u;
          }
          if (it < nrows) prank[it] = kept ? rk : 0xFFFFFFFFu;
          else if (it < n) orank[it - nrows] = (u16)rk;
          nk = (u32)__popcll(keptmask);
        }
      } else {
        u32 nkp = 0;
        for (u32 j = 0; j < nrows; j++) nkp += prec[j] >= rec_min ? 1u : 0u;
        nk = nkp + nok;
        for (u32 it = tidB; it < nrows + nok; it += PV_TPB) {
          Key<KW> mine; bool kept = true;
          if (it < nrows) { mine = pk[it]; kept = prec[it] >= rec_min; }
          else mine = ov[(otab[okl[it - nrows]] & 0xFFFFu) - 1].key;
          u32 rk = 0;
          for (u32 j = 0; j < nrows; j++) rk += (prec[j] >= rec_min && key_less<KW>(pk[j], mine)) ? 1u : 0u;
          for (u32 j = 0; j < nok; j++) rk += key_less<KW>(ov[(otab[okl[j]] & 0xFFFFu) - 1].key, mine) ? 1u : 0u;
          if (it < nrows) prank[it] = kept ? rk : 0xFFFFFFFFu;
          else orank[it - nrows] = (u16)rk;
        }
      }
      if (tidB == 0) {
        u64 off = 0;
        if (nk) {
          u64 ch_base = al64[0]; u32 ch_used = al[0], ch_cap = al[1], ch_ok = al[3];
          if (ch_used + nk > ch_cap) {
            if (ch_used) {
              const u64 sidx = atomicAdd(&T.ctrl[1], 1ULL);
              if (sidx < T.seg_cap) { Seg sg; sg.range = range; sg.seq = al[2]; sg.row_off = ch_base; sg.nrows = ch_used; sg.pad = 0; T.segs[sidx] = sg; }
              else atomicOr(&T.ctrl[2], (u64)ERR_SEGS_OVERFLOW);
              atomicAdd(&T.ctrl[3], (u64)ch_used);
            }
            ch_cap = max(chunk_rows, nk);
            ch_base = atomicAdd(&T.ctrl[0], (u64)ch_cap);
            ch_used = 0; al[2] = seq;
            ch_ok = (ch_base + ch_cap <= T.out_cap_rows) ? 1u : 0u;
            if (!ch_ok) atomicOr(&T.ctrl[2], (u64)ERR_ROWS_OVERFLOW);
          }
          off = ch_base + ch_used; ch_used += nk;
          al64[0] = ch_base; al[0] = ch_used; al[1] = ch_cap; al[3] = ch_ok;
        }
        sh64[0] = off; sh[3] = al[3];
      }
      pv_lds_barrier();
      // overflow-key ranks go into their table entries (low 16 bits; the owner index is no longer needed)
      for (u32 q = tidB; q < nok; q += PV_TPB) {
        const u32 t = okl[q];
        otab[t] = (otab[t] & 0xFFFF0000u) | orank[q];
      }
      PVPH(4);   // no barrier: rows-out reads prank/orank (complete since the barrier above), not the table
      const u64 tile_base = pv_uni64(sh64[0]);
      const bool can_write = pv_uni(sh[3]) != 0;
      u8* const out0 = T.out + tile_base * row_bytes;

      // ---- rows out: kept pivot rows from the LDS image; overflow rows zero-filled in HBM ----
      if (can_write) {
        const int wave = tidB >> 6;
        for (u32 j = wave; j < nrows + nok; j += PV_TPB / 64) {
          if (j < nrows) {
            const u32 rk = prank[j];
            if (rk == 0xFFFFFFFFu) continue;
            const u8* src = img + j * row_bytes;
            u8* dst = out0 + (u64)rk * row_bytes;
            if (MODE == 0) {
              if (((reinterpret_cast<uintptr_t>(dst) | (uintptr_t)(j * row_bytes) | row_bytes) & 7u) == 0) {
                for (u32 t0 = 0; t0 < row_bytes / 8; t0 += 256) {      // four row words in flight per lane
                  u64 w[4];
#pragma unroll
                  for (int q = 0; q < 4; q++) { const u32 t = t0 + 64 * q + laneB; w[q] = t < row_bytes / 8 ? reinterpret_cast<const u64*>(src)[t] : 0ULL; }
#pragma unroll
                  for (int q = 0; q < 4; q++) { const u32 t = t0 + 64 * q + laneB; if (t < row_bytes / 8) reinterpret_cast<u64*>(dst)[t] = w[q]; }
                }
              } else {
                for (u32 t = laneB; t < row_bytes / 4; t += 64) reinterpret_cast<u32*>(dst)[t] = reinterpret_cast<const u32*>(src)[t];
              }
            }
            else { for (u32 t = laneB; t < row_bytes; t += 64) dst[t] = src[t]; }
          } else {
            u8* dst = out0 + (u64)orank[j - nrows] * row_bytes;
            if (MODE == 0) { for (u32 t = laneB; t < row_bytes / 4; t += 64) reinterpret_cast<u32*>(dst)[t] = 0; }
            else { for (u32 t = laneB; t < row_bytes; t += 64) dst[t] = 0; }
          }
        }
      }
      if (nok) __syncthreads();   // zero-filled overflow rows are in memory, and their ranks in the table, before the scatter
      PVPH(5);
      if (anykept) {
#pragma unroll
      for (int q = 0; q < OQ; q++) {
        const u32 t = ((u32)tidB >> 6) * PV_OVW + (u32)laneB + 64u * q;
        if (64u * q < wov_mine && (u32)laneB + 64u * q < wov_mine) {
          const OvRec<KW> o = ov[t];
          const u32 e = otab[hs[q]];
          const u32 rk = e & 0xFFFFu;
          const u32 outc = ((solidm >> q) & 1u) ? o.cnt : 0u;   // no rescue here: tasks with share-min go to k_merge_rows
          if (can_write && rk != 0xFFFFu) {
            u8* dst = out0 + (u64)rk * row_bytes;
            if ((ownm >> q) & 1u) {   // the owner writes the row key
              if (MODE == 0) { u32* rw = reinterpret_cast<u32*>(dst);
#pragma unroll
                for (int w2 = 0; w2 < KW; w2++) { rw[2 * w2] = (u32)o.key.w[w2]; rw[2 * w2 + 1] = (u32)(o.key.w[w2] >> 32); } }
              else {
#pragma unroll
                for (int b = 0; b < KW * 8; b++) dst[b] = (u8)(o.key.w[b >> 3] >> ((b & 7) * 8));
              }
            }
            if (outc) {
              if (MODE == 0) reinterpret_cast<u32*>(dst + KW * 8)[o.list] = outc;
              else {
                u8* bp = dst + KW * 8 + (o.list >> 3);
                const uintptr_t a = reinterpret_cast<uintptr_t>(bp);
                atomicOr(reinterpret_cast<u32*>(a & ~(uintptr_t)3), 1u << (((a & 3u) << 3) + (o.list & 7u)));
              }
            }
          }
        }
      }
      }
      pv_lds_barrier();
#pragma unroll
      for (int q = 0; q < OQ; q++) if ((ownm >> q) & 1u) otab[hs[q]] = 0;   // hash set clean for the next tile
      PVPH(6);
      ppos += rte_eff;
      seq++;
      // the range ends with its open-ended tile (that one takes everything the lists have left)
      if (done) break;
      // no barrier: the next tile's set-up writes pk, the (already cleared) row table, the image and the
      // counters, all of which were last read before the barrier above; the hash set is next touched
      // several barriers from here
    }

    // ---- range done ----
    if (tid == 0) {
      if (failed) atomicOr(&T.ctrl[2], (u64)ERR_FALLBACK);
      const u32 ch_used = al[0];
      if (ch_used) {
        const u64 sidx = atomicAdd(&T.ctrl[1], 1ULL);
        if (sidx < T.seg_cap) { Seg sg; sg.range = range; sg.seq = al[2]; sg.row_off = al64[0]; sg.nrows = ch_used; sg.pad = 0; T.segs[sidx] = sg; }
        else atomicOr(&T.ctrl[2], (u64)ERR_SEGS_OVERFLOW);
        atomicAdd(&T.ctrl[3], (u64)ch_used);
      }
    }
    pv_lds_barrier();
    for (u32 i = tid; i < N; i += PV_TPB) {
      if (lt_nso[i]) atomicAdd(&T.stats[0 * (u64)N + i], (u64)lt_nso[i]);
      if (lt_two[i]) atomicAdd(&T.stats[4 * (u64)N + i], lt_two[i]);
    }
    __syncthreads();
  }
}

template __global__ void k_merge_pivot<1, 0>(const TaskDev*, const uint2*, u32, u32*);
template __global__ void k_merge_pivot<1, 1>(const TaskDev*, const uint2*, u32, u32*);

#ifdef KMX_PHASE_PROF
void pivot_phase_prof_dump()
{
  u64 h[16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(kmx_pivot_prof), sizeof(h)) != hipSuccess) return;
  u64 tot = 0; for (int i = 0; i < 16; i++) tot += h[i];
  static const char* nm[16] = {"setup", "T0+barrier", "popcount+bar", "ov-hash", "publish+rank", "rows-out", "ov-scatter+clr", "-", "process-sweep",
                               "window-sweep", "bar(window)", "candidates", "bar(cand)", "lim+prefetch", "refill-sweep", "bar(scan)"};
  for (int i = 0; i < 16; i++) fprintf(stderr, "[pivot] %-14s %6.2f%%  %llu\n", nm[i], tot ? 100.0 * h[i] / tot : 0.0, h[i]);
  memset(h, 0, sizeof(h));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(kmx_pivot_prof), h, sizeof(h));
}
#endif

int pivot_lds_bytes(int kw)
{ return PV_IMG + PV_OVCAP * (kw * 8 + 8) + PV_OT * 4 + 1280 + PV_PT * (kw * 8 + 8) + PV_OVCAP * 4 + 36864; }   // image + overflow + hash set + tables + list state
u32 pivot_max_lists() { return PV_MAXN; }

hipError_t launch_merge_pivot(int kw, int mode, const TaskDev* tasks, const uint2* items, u32 n_items, u32* ticket,
                              u32 grid_x, hipStream_t st)
{
  const int lds = pivot_lds_bytes(kw);
  dim3 grid(grid_x), block(PV_TPB);
#define KMX_LAUNCH(KW_, MODE_)                                                                              \
  do {                                                                                                      \
    hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_merge_pivot<KW_, MODE_>),          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);                   \
    if (e_ != hipSuccess) return e_;                                                                        \
    hipLaunchKernelGGL((k_merge_pivot<KW_, MODE_>), grid, block, lds, st, tasks, items, n_items, ticket);   \
  } while (0)
  if (kw == 1 && mode == 0) KMX_LAUNCH(1, 0);
  else if (kw == 1 && mode == 1) KMX_LAUNCH(1, 1);
  else return hipErrorInvalidValue;
#undef KMX_LAUNCH
  return hipGetLastError();
}

}  // namespace kmx
