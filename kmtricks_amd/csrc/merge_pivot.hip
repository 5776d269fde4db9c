// merge_pivot.hip -- pivot-tiled streaming merge for related samples (COUNT / PA rows) on gfx950.
// Same results as k_merge_rows (reference include/kmtricks/merge.hpp:183-286, 441-558), different
// decomposition, built for the case the metric is quoted on: many lists that share most of their keys.
//
//   * The task's longest list is the PIVOT.  A tile is RT consecutive pivot records; its key range
//     [first key of the range, key of the pivot record after the tile) is walked by every list
//     with its own sequential cursor, exactly like the Bloom kernel: g adjacent lanes stream a
//     list's records with 8 loads in flight each (one 12/20-byte record per load, a list's lines
//     are consumed in one or two visits instead of five).
//   * A record whose key IS one of the tile's pivot keys (binary search over <= 16 keys in LDS)
//     goes straight to row j of the tile's LDS image: no hash set, no ranking, no WG-wide bound.
//     The recurrence counter of a row saturates at max(recurrence-min, share-min) and is read
//     before it is incremented, so the ~N lanes that hit the same row do not serialise.
//   * A record whose key is NOT a pivot key (sample-private k-mers, keys the pivot lacks) is put
//     in an LDS overflow buffer; after the scan the (few) overflow records are merged among
//     themselves with the hash set of k_merge_rows, kept keys are ranked together with the kept
//     pivot rows, and their (sparse) rows are written straight to HBM.
//   * If a tile's overflow does not fit, the tile is retried with half as many pivot records; if
//     one pivot gap alone does not fit, the task is flagged and the driver re-runs it with
//     k_merge_rows -- results never depend on how well the pivot covers the other lists.
// Rows leave through the same chunked arena + (range, seq) directory as k_merge_rows.
#include "kmx_dev.hpp"
#include <algorithm>
#include <cstdio>
#include <cstring>

namespace kmx {

constexpr int PV_TPB = 1024;
constexpr int PV_RTMAX = 16;        // pivot records per tile
constexpr int PV_IMG = 61440;       // LDS row image bytes (15 rows of 1000 u32 counts)
constexpr int PV_OVCAP = 2048;      // overflow records per tile
constexpr int PV_OT = 2 * PV_OVCAP; // overflow hash set entries
constexpr int PV_G = 8;             // adjacent lanes per list: one wave load covers 8 lists x 96 contiguous bytes
constexpr int PV_LPP = PV_TPB / PV_G;   // lists per pass (128)
constexpr int PV_PB = 2;            // passes whose records are in flight together (8 loads per lane)
constexpr int PV_U = 3;             // records per lane and pass in the prefetch batch (24 per list: the tail loop is rare)

template <int KW> struct OvRec { Key<KW> key; u32 cnt; u32 list; };

__device__ __forceinline__ void pv_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int KW> __device__ __forceinline__ u32 pv_hash(const Key<KW>& k)
{
  u32 x = (u32)k.w[0] ^ ((u32)(k.w[0] >> 32) * 0x9E3779B1u);
  if (KW == 2) x ^= ((u32)k.w[KW - 1] * 0x85EBCA77u) ^ ((u32)(k.w[KW - 1] >> 32) * 0xC2B2AE3Du);
  x *= 0x85EBCA6Bu; x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 13;
  return x & (PV_OT - 1);
}

template <int KW> __device__ __forceinline__ Key<KW> gload_key(gu32* p)
{
  Key<KW> k;
#pragma unroll
  for (int q = 0; q < KW; q++) k.w[q] = (u64)p[2 * q] | ((u64)p[2 * q + 1] << 32);
  return k;
}

#ifdef KMX_PHASE_PROF
__device__ u64 kmx_pivot_prof[16];
#endif

template <int KW, int MODE>
__global__ __launch_bounds__(PV_TPB, 4)
void k_merge_pivot(const TaskDev* __restrict__ tasks, const uint2* __restrict__ items, u32 n_items, u32* ticket)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int RB4 = (KW * 8 + 4) / 4;
  unsigned char* const img = smem;                                                        // [PV_IMG]
  OvRec<KW>* ov = reinterpret_cast<OvRec<KW>*>(smem + PV_IMG);                            // [PV_OVCAP]
  u32* otab = reinterpret_cast<u32*>(smem + PV_IMG + PV_OVCAP * sizeof(OvRec<KW>));       // [PV_OT]
  unsigned char* misc = smem + PV_IMG + PV_OVCAP * sizeof(OvRec<KW>) + PV_OT * 4;
  Key<KW>* pk = reinterpret_cast<Key<KW>*>(misc);                                         // [PV_RTMAX] pivot keys of the tile
  u32* prec = reinterpret_cast<u32*>(misc + 256);                                         // [PV_RTMAX] recurrence (saturating)
  u32* prank = reinterpret_cast<u32*>(misc + 320);                                        // [PV_RTMAX] final row or ~0
  u16* okl = reinterpret_cast<u16*>(misc + 384);                                          // [PV_OVCAP] table slots of kept overflow keys
  u16* orank = reinterpret_cast<u16*>(misc + 384 + PV_OVCAP * 2);                         // [PV_OVCAP] their final rows
  u32* sh = reinterpret_cast<u32*>(misc + 384 + PV_OVCAP * 4);                            // [0] item [1] ovn [2] nok [3] can-write
  u64* sh64 = reinterpret_cast<u64*>(misc + 384 + PV_OVCAP * 4 + 32);                     // [0] tile row base
  // per-list state (a list is served by 8 lanes, a lane serves up to 8 lists: the state lives in LDS)
  unsigned char* lt = misc + 384 + PV_OVCAP * 4 + 64;
  u64* lt_base = reinterpret_cast<u64*>(lt);                    // [1024] record base
  u64* lt_two = reinterpret_cast<u64*>(lt + 8192);              // [1024] TOTAL_WO of the range so far
  u64* pd_two = reinterpret_cast<u64*>(lt + 16384);             // [1024] ... of the tile attempt (committed on success)
  u32* lt_cur = reinterpret_cast<u32*>(lt + 24576);             // [1024] cursor
  u32* pd_nxt = reinterpret_cast<u32*>(lt + 28672);             // [1024] cursor after the tile attempt
  u32* lt_nso = reinterpret_cast<u32*>(lt + 32768);             // [1024] NON_SOLID so far
  u32* pd_nso = reinterpret_cast<u32*>(lt + 36864);             // [1024] ... of the tile attempt

  const int tid = threadIdx.x, lane = tid & 63;
  for (int t = tid; t < PV_OT; t += PV_TPB) otab[t] = 0;
#ifdef KMX_PHASE_PROF
  long long pt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; long long pc = clock64();
#define PVPH(i) do { const long long n_ = clock64(); pt[i] += n_ - pc; pc = n_; } while (0)
#else
#define PVPH(i) do {} while (0)
#endif

  for (;;) {
    if (tid == 0) sh[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    const u32 item = sh[0];
    __syncthreads();
    if (item >= n_items) {
#ifdef KMX_PHASE_PROF
      if (tid == 0) for (int i = 0; i < 9; i++) atomicAdd(&kmx_pivot_prof[i], (u64)pt[i]);
#endif
      return;
    }
    const TaskDev& T = tasks[items[item].x];
    const u32 range = items[item].y;
    const u32 N = T.N, rec_min = T.rec_min, share_min = T.share_min, row_bytes = T.row_bytes;
    const u32 sat = max(rec_min, share_min);
    const u32 chunk_rows = max(64u, 262144u / row_bytes);
    const u32 rt_cap = min((u32)PV_RTMAX, (u32)PV_IMG / row_bytes);

    // PV_G adjacent lanes per list; list p*PV_LPP + tid/PV_G in pass p (N <= 1024 -> <= 8 passes)
    const u32 npass = (N + PV_LPP - 1) / PV_LPP;
    const u32 lg = tid / PV_G, r = tid & (PV_G - 1);
    const u32* const endp = T.bounds + (u64)(range + 1) * N;
    for (u32 i = tid; i < N; i += PV_TPB) {
      lt_base[i] = (u64)(uintptr_t)T.recs[i];
      lt_cur[i] = T.bounds[(u64)range * N + i];
      lt_nso[i] = 0; lt_two[i] = 0;
    }
    // the pivot's records of this range define the tiles
    gu32* pbase = (gu32*)(uintptr_t)T.recs[T.pivot];
    const u32 pend = T.bounds[(u64)(range + 1) * N + T.pivot];
    u32 ppos = T.bounds[(u64)range * N + T.pivot];
    u64 ch_base = 0; u32 ch_used = 0, ch_cap = 0, ch_seq = 0, ch_ok = 1;
    u32 seq = 0, rt_try = rt_cap;
    bool failed = false, split = false;
    Key<KW> ksplit = key_inf<KW>();   // artificial upper key of a tile that cuts an oversized pivot gap

    PVPH(0);
    for (;;) {
      // ---- tile = pivot records [ppos, ppos + rte); key range up to the next pivot key ----
      const u32 rte = min(rt_try, pend - ppos);
      const bool open_end = !split && ppos + rte >= pend;     // last tile of the range: bounded by the lists' range ends
      Key<KW> khi = key_inf<KW>();
      if (split) khi = ksplit;
      else if (!open_end) khi = gload_key<KW>(pbase + (u64)(ppos + rte) * RB4);
      if (tid < (int)rte) {
        const Key<KW> k = gload_key<KW>(pbase + (u64)(ppos + tid) * RB4);
        pk[tid] = k; prec[tid] = 0;
      }
      {
        const u32 zb = rte * row_bytes;
        uint4* z = reinterpret_cast<uint4*>(img);
        for (u32 t = tid; t < (zb + 15) / 16; t += PV_TPB) z[t] = make_uint4(0, 0, 0, 0);
        if (tid == 0) { sh[1] = 0; sh[2] = 0; sh[4] = 0; }
      }
      pv_lds_barrier();
      if (tid < (int)rte) {   // row keys
        u8* row = img + tid * row_bytes;
        const Key<KW> k = pk[tid];
        if (MODE == 0) { u32* rw = reinterpret_cast<u32*>(row);
#pragma unroll
          for (int q = 0; q < KW; q++) { rw[2 * q] = (u32)k.w[q]; rw[2 * q + 1] = (u32)(k.w[q] >> 32); } }
        else {
#pragma unroll
          for (int q = 0; q < KW * 8; q++) row[q] = (u8)(k.w[q >> 3] >> ((q & 7) * 8));
        }
      }

      PVPH(1);
      // ---- scan: every list streams its records of the tile's key range ----
      // The tile's pivot keys sit in registers (uniform values): a record's row is the number of pivot
      // keys below its key, found by brute-force compares -- no memory access, no dependent chain.
      Key<KW> pkr[PV_RTMAX];
#pragma unroll
      for (int j = 0; j < PV_RTMAX; j++) pkr[j] = j < (int)rte ? pk[j] : key_inf<KW>();
      // exact lookup for the (rare) tail loop
      auto row_of = [&](const Key<KW>& k, bool& found) -> u32 {
        u32 rank = 0, eq = 0;
#pragma unroll
        for (int j = 0; j < PV_RTMAX; j++) { rank += key_less<KW>(pkr[j], k) ? 1u : 0u; eq += key_eq<KW>(pkr[j], k) ? 1u : 0u; }
        found = eq != 0;
        return rank;
      };
      auto deposit = [&](u32 row, u32 c, u32 li) {
        if (MODE == 0) reinterpret_cast<u32*>(img + row * row_bytes + KW * 8)[li] = c;
        else { const u32 ob = row * row_bytes + KW * 8 + (li >> 3);
               atomicOr(reinterpret_cast<u32*>(img) + (ob >> 2), 1u << (((ob & 3u) << 3) + (li & 7u))); }
      };
      for (u32 p0 = 0; p0 < npass; p0 += PV_PB) {
        // prefetch batch: PV_U records per lane for PV_PB passes before anything is processed
        Key<KW> kk[PV_PB][PV_U]; u32 cc[PV_PB][PV_U]; u32 c0[PV_PB], ee[PV_PB], sm[PV_PB];
#pragma unroll
        for (int p = 0; p < PV_PB; p++) {
          const u32 li = (p0 + p) * PV_LPP + lg;
          const bool on = p0 + p < npass && li < N;
          c0[p] = on ? lt_cur[li] : 0;
          ee[p] = on ? endp[li] : 0;
          sm[p] = on ? T.soft_min[li] : 0;
          gu32* base = (gu32*)(uintptr_t)(on ? lt_base[li] : 0);
#pragma unroll
          for (int u = 0; u < PV_U; u++) {
            const u32 ix = c0[p] + r + u * PV_G;
            kk[p][u] = key_inf<KW>(); cc[p][u] = 0;
            if (ix < ee[p]) { gu32* q = base + (u64)ix * RB4; kk[p][u] = gload_key<KW>(q); cc[p][u] = q[2 * KW]; }
          }
        }
#ifdef KMX_PHASE_PROF
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PVPH(7);
#endif
        u32 ovm = 0;                     // bit p*PV_U+u: overflow record
        u32 nx[PV_PB], tn[PV_PB]; u64 tt[PV_PB];
#pragma unroll
        for (int p = 0; p < PV_PB; p++) {
          nx[p] = c0[p]; tn[p] = 0; tt[p] = 0;
          const u32 li = (p0 + p) * PV_LPP + lg;
#pragma unroll
          for (int u = 0; u < PV_U; u++) {
            const u32 ix = c0[p] + r + u * PV_G;
            const Key<KW> k = kk[p][u];
            const bool valid = ix < ee[p] && (open_end || key_less<KW>(k, khi));
            u32 rank = 0, eq = 0;
#pragma unroll
            for (int j = 0; j < PV_RTMAX; j++) { rank += key_less<KW>(pkr[j], k) ? 1u : 0u; eq += key_eq<KW>(pkr[j], k) ? 1u : 0u; }
            const bool solid = cc[p][u] >= sm[p];
            if (valid) {
              nx[p] = ix + 1;
              if (solid) tt[p] += cc[p][u]; else tn[p]++;
              if (eq) { if (solid) deposit(rank, cc[p][u], li); }
              else ovm |= 1u << (p * PV_U + u);
            }
          }
        }
        // lists with more than PV_U * PV_G records in the tile (much denser than the pivot here): keep going
#pragma unroll
        for (int p = 0; p < PV_PB; p++) {
          const u32 li = (p0 + p) * PV_LPP + lg;
          bool more = nx[p] == c0[p] + r + (PV_U - 1) * PV_G + 1;       // my last prefetched record was consumed
          for (u32 i0 = c0[p] + r + PV_U * PV_G; __any(more && i0 < ee[p]); i0 += PV_G) {
            if (more && i0 < ee[p]) {
              gu32* q = (gu32*)(uintptr_t)lt_base[li] + (u64)i0 * RB4;
              const Key<KW> k = gload_key<KW>(q); const u32 c = q[2 * KW];
              if (!open_end && !key_less<KW>(k, khi)) more = false;
              else {
                nx[p] = i0 + 1;
                const bool solid = c >= sm[p];
                if (solid) tt[p] += c; else tn[p]++;
                bool found; const u32 row = row_of(k, found);
                if (found) { if (solid) deposit(row, c, li); }
                else { const u32 pos = atomicAdd(&sh[1], 1u); if (pos < (u32)PV_OVCAP) { OvRec<KW> o; o.key = k; o.cnt = c; o.list = li; ov[pos] = o; } }
              }
            } else more = false;
          }
        }
        {   // aggregated append of this batch's overflow records: one LDS atomic per wave
          const u32 mine = __popc(ovm), incl = wave_incl_scan(mine, lane), total = __shfl(incl, 63);
          if (total) {
            u32 bpos = 0; if (lane == 63) bpos = atomicAdd(&sh[1], total); bpos = __shfl(bpos, 63);
            u32 pos = bpos + incl - mine;
#pragma unroll
            for (int p = 0; p < PV_PB; p++) {
#pragma unroll
              for (int u = 0; u < PV_U; u++) {
                if ((ovm >> (p * PV_U + u)) & 1u) {
                  if (pos < (u32)PV_OVCAP) { OvRec<KW> o; o.key = kk[p][u]; o.cnt = cc[p][u]; o.list = (p0 + p) * PV_LPP + lg; ov[pos] = o; }
                  pos++;
                }
              }
            }
          }
        }
        // the lists' 8 lanes: new cursor + statistics of this attempt -> pending slots (committed on success)
#pragma unroll
        for (int off = 1; off < PV_G; off <<= 1) {
#pragma unroll
          for (int p = 0; p < PV_PB; p++) {
            nx[p] = max(nx[p], (u32)__shfl_xor(nx[p], off)); tn[p] += __shfl_xor(tn[p], off); tt[p] += shfl_xor_u64(tt[p], off);
          }
        }
#pragma unroll
        for (int p = 0; p < PV_PB; p++) {
          const u32 li = (p0 + p) * PV_LPP + lg;
          if (p0 + p < npass && li < N && r == 0) { pd_nxt[li] = nx[p]; pd_nso[li] = tn[p]; pd_two[li] = tt[p]; }
        }
#ifdef KMX_PHASE_PROF
        PVPH(8);
#endif
      }
      // recurrence of the pivot rows = number of lists that deposited a (solid) count: wave j counts row j
      pv_lds_barrier();
      {
        const u32 wv = tid >> 6;
        for (u32 j = wv; j < rte; j += PV_TPB / 64) {
          u32 nz = 0;
          if (MODE == 0) { const u32* rowc = reinterpret_cast<const u32*>(img + j * row_bytes + KW * 8);
                           for (u32 t = lane; t < N; t += 64) nz += rowc[t] != 0 ? 1u : 0u; }
          else { const u8* rowb = img + j * row_bytes + KW * 8;
                 for (u32 t = lane; t < (N + 7) / 8; t += 64) nz += __popc((u32)rowb[t]); }
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) nz += __shfl_xor(nz, off);
          if (lane == 0) prec[j] = nz;
        }
      }
      pv_lds_barrier();
      PVPH(2);
      const u32 ovn = sh[1];
      if (ovn > (u32)PV_OVCAP) {
        // the overflow does not fit: retry this tile with fewer pivot records (the image, counters
        // and overflow buffer are rebuilt; cursors and statistics were not committed)
        if (rte > 1 && !split) rt_try = max(1u, rte >> 1);
        else {
          // one pivot gap alone holds more than the buffer: cut it at the largest buffered key.  A key
          // has at most N <= 1024 records, so the buffer holds >= 2 distinct keys and the part below
          // the cut is strictly smaller -- repeated cuts always terminate.
          Key<KW> mx; for (int q = 0; q < KW; q++) mx.w[q] = 0;
          for (u32 t = tid; t < (u32)PV_OVCAP; t += PV_TPB) { const Key<KW> k = ov[t].key; if (key_less<KW>(mx, k)) mx = k; }
          // max = min of the complemented key
          Key<KW> cm; for (int q = 0; q < KW; q++) cm.w[q] = ~mx.w[q];
          cm = wave_min_key<KW>(cm);
          if (lane == 0) pk[tid >> 6] = cm;        // pk is rebuilt at the top of the next attempt
          pv_lds_barrier();
          Key<KW> best = pk[0];
          for (int v = 1; v < PV_TPB / 64; v++) best = key_min<KW>(best, pk[v]);
          for (int q = 0; q < KW; q++) ksplit.w[q] = ~best.w[q];
          split = true;
        }
        pv_lds_barrier();
        continue;
      }
      for (u32 i = tid; i < N; i += PV_TPB) { lt_cur[i] = pd_nxt[i]; lt_nso[i] += pd_nso[i]; lt_two[i] += pd_two[i]; }   // commit

      // ---- overflow records: merge them among themselves (hash set, as k_merge_rows) ----
      u32 hs[2] = {0, 0}; u32 ownm = 0, solidm = 0;
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const u32 t = tid + q * PV_TPB;
        if (t < ovn) {
          const OvRec<KW> o = ov[t];
          u32 h = pv_hash<KW>(o.key), old;
          for (;;) {
            old = otab[h];
            if (old == 0) { old = atomicCAS(&otab[h], 0u, t + 1); if (old == 0) { ownm |= 1u << q; break; } }
            if (key_eq<KW>(ov[(old & 0xFFFFu) - 1].key, o.key)) break;
            h = (h + 1) & (PV_OT - 1);
          }
          hs[q] = h;
          if (o.cnt >= T.soft_min[o.list]) { solidm |= 1u << q; if ((old >> 16) < sat) atomicAdd(&otab[h], 1u << 16); }
        }
      }
      pv_lds_barrier();
#pragma unroll
      for (int q = 0; q < 2; q++) {
        if ((ownm >> q) & 1u) {
          const u32 e = otab[hs[q]];
          if ((e >> 16) >= rec_min) { const u32 pos = atomicAdd(&sh[2], 1u); okl[pos] = (u16)hs[q]; }
          else otab[hs[q]] = (e & 0xFFFF0000u) | 0xFFFFu;
        }
      }
      pv_lds_barrier();
      PVPH(3);
      const u32 nok = sh[2];
      // ---- final row order: kept pivot rows and kept overflow keys together ----
      u32 nkp = 0;
      for (u32 j = 0; j < rte; j++) nkp += prec[j] >= rec_min ? 1u : 0u;
      const u32 nk = nkp + nok;
      for (u32 it = tid; it < rte + nok; it += PV_TPB) {
        Key<KW> mine; bool kept = true;
        if (it < rte) { mine = pk[it]; kept = prec[it] >= rec_min; }
        else mine = ov[(otab[okl[it - rte]] & 0xFFFFu) - 1].key;
        u32 rk = 0;
        for (u32 j = 0; j < rte; j++) rk += (prec[j] >= rec_min && key_less<KW>(pk[j], mine)) ? 1u : 0u;
        for (u32 j = 0; j < nok; j++) rk += key_less<KW>(ov[(otab[okl[j]] & 0xFFFFu) - 1].key, mine) ? 1u : 0u;
        if (it < rte) prank[it] = kept ? rk : 0xFFFFFFFFu;
        else orank[it - rte] = (u16)rk;
      }
      if (tid == 0) {
        u64 off = 0;
        if (nk) {
          if (ch_used + nk > ch_cap) {
            if (ch_used) {
              const u64 sidx = atomicAdd(&T.ctrl[1], 1ULL);
              if (sidx < T.seg_cap) { Seg sg; sg.range = range; sg.seq = ch_seq; sg.row_off = ch_base; sg.nrows = ch_used; sg.pad = 0; T.segs[sidx] = sg; }
              else atomicOr(&T.ctrl[2], (u64)ERR_SEGS_OVERFLOW);
              atomicAdd(&T.ctrl[3], (u64)ch_used);
            }
            ch_cap = max(chunk_rows, nk);
            ch_base = atomicAdd(&T.ctrl[0], (u64)ch_cap);
            ch_used = 0; ch_seq = seq;
            ch_ok = (ch_base + ch_cap <= T.out_cap_rows) ? 1u : 0u;
            if (!ch_ok) atomicOr(&T.ctrl[2], (u64)ERR_ROWS_OVERFLOW);
          }
          off = ch_base + ch_used; ch_used += nk;
        }
        sh64[0] = off; sh[3] = ch_ok;
      }
      pv_lds_barrier();
      // overflow-key ranks go into their table entries (low 16 bits; the owner index is no longer needed)
      for (u32 q = tid; q < nok; q += PV_TPB) {
        const u32 t = okl[q];
        otab[t] = (otab[t] & 0xFFFF0000u) | orank[q];
      }
      __syncthreads();
      PVPH(4);
      const u64 tile_base = sh64[0];
      const bool can_write = sh[3] != 0;
      u8* const out0 = T.out + tile_base * row_bytes;

      // ---- rows out: kept pivot rows from the LDS image; overflow rows zero-filled in HBM ----
      if (can_write) {
        const int wave = tid >> 6;
        for (u32 j = wave; j < rte + nok; j += PV_TPB / 64) {
          if (j < rte) {
            const u32 rk = prank[j];
            if (rk == 0xFFFFFFFFu) continue;
            const u8* src = img + j * row_bytes;
            u8* dst = out0 + (u64)rk * row_bytes;
            if (MODE == 0) { for (u32 t = lane; t < row_bytes / 4; t += 64) reinterpret_cast<u32*>(dst)[t] = reinterpret_cast<const u32*>(src)[t]; }
            else { for (u32 t = lane; t < row_bytes; t += 64) dst[t] = src[t]; }
          } else {
            u8* dst = out0 + (u64)orank[j - rte] * row_bytes;
            if (MODE == 0) { for (u32 t = lane; t < row_bytes / 4; t += 64) reinterpret_cast<u32*>(dst)[t] = 0; }
            else { for (u32 t = lane; t < row_bytes; t += 64) dst[t] = 0; }
          }
        }
      }
      __syncthreads();   // zero-filled overflow rows are in memory before their entries are scattered
      PVPH(5);
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const u32 t = tid + q * PV_TPB;
        if (t < ovn) {
          const OvRec<KW> o = ov[t];
          const u32 e = otab[hs[q]];
          const u32 rec = e >> 16, rk = e & 0xFFFFu;
          u32 outc = 0;
          if ((solidm >> q) & 1u) outc = o.cnt;
          else {
            if (share_min && rec >= share_min) {
              outc = o.cnt;
              atomicAdd(&T.stats[1 * (u64)N + o.list], 1ULL);
              atomicAdd(&T.stats[5 * (u64)N + o.list], (u64)o.cnt);
            }
          }
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
      pv_lds_barrier();
#pragma unroll
      for (int q = 0; q < 2; q++) if ((ownm >> q) & 1u) otab[hs[q]] = 0;   // hash set clean for the next tile
      PVPH(6);
      ppos += rte;
      seq++;
      rt_try = rt_cap;
      split = false;
      // the range ends with its open-ended tile (that one takes everything the lists have left)
      if (open_end) break;
      pv_lds_barrier();
    }

    // ---- range done ----
    if (tid == 0) {
      if (failed) atomicOr(&T.ctrl[2], (u64)ERR_FALLBACK);
      if (ch_used) {
        const u64 sidx = atomicAdd(&T.ctrl[1], 1ULL);
        if (sidx < T.seg_cap) { Seg sg; sg.range = range; sg.seq = ch_seq; sg.row_off = ch_base; sg.nrows = ch_used; sg.pad = 0; T.segs[sidx] = sg; }
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
template __global__ void k_merge_pivot<2, 0>(const TaskDev*, const uint2*, u32, u32*);
template __global__ void k_merge_pivot<2, 1>(const TaskDev*, const uint2*, u32, u32*);

#ifdef KMX_PHASE_PROF
void pivot_phase_prof_dump()
{
  u64 h[16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(kmx_pivot_prof), sizeof(h)) != hipSuccess) return;
  u64 tot = 0; for (int i = 0; i < 9; i++) tot += h[i];
  static const char* nm[9] = {"setup", "tile-init", "scan-rest", "ov-hash", "publish+rank", "rows-out", "ov-scatter", "scan-loadwait", "scan-process"};
  for (int i = 0; i < 9; i++) fprintf(stderr, "[pivot] %-14s %6.2f%%  %llu\n", nm[i], tot ? 100.0 * h[i] / tot : 0.0, h[i]);
  memset(h, 0, sizeof(h));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(kmx_pivot_prof), h, sizeof(h));
}
#endif

int pivot_lds_bytes(int kw)
{ return PV_IMG + PV_OVCAP * (kw * 8 + 8) + PV_OT * 4 + 384 + PV_OVCAP * 4 + 64 + 40960; }   // image + overflow + hash set + list state
u32 pivot_max_lists() { return PV_TPB; }

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
  else if (kw == 2 && mode == 0) KMX_LAUNCH(2, 0);
  else if (kw == 2 && mode == 1) KMX_LAUNCH(2, 1);
  else return hipErrorInvalidValue;
#undef KMX_LAUNCH
  return hipGetLastError();
}

}  // namespace kmx
