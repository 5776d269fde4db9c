// merge_bft.hip -- hash-mode merge straight into the SAMPLE-major Bloom matrix (hash:bft:bin) on gfx950.
// Replaces km::HashMerger::next + write_as_bft (reference include/kmtricks/merge.hpp:441-517, 575-600, 631-644:
// write_as_bf into a temporary file, load it into a BitMatrix, transpose, dump): row s of the result is sample s's
// slice of its final Bloom filter (howde_utils.hpp:133-187), ceil8(W)/8 bytes, bit r = hash lower + r.
//
// k_merge_bf keeps a tile of hash rows x ALL samples in LDS; with thousands of samples a row is hundreds of bytes,
// a tile a hundred rows, and a sample has a record in it every other tile -- the kernel then spends its time
// re-reading the heads of 2500 lists.  Here the matrix is cut the other way:
//   * a tile is 16384 hash rows (a sample hands over ~80 records, 1 KB, per tile at BASELINE configs[3]'s density); a workgroup
//     owns a run of tiles and walks each tile's samples in blocks of 32 (16 adjacent lanes per sample, 8 loads in flight
//     each: one round trip fetches a sample's records of the tile, in 192-byte runs per load instruction);
//   * the block's LDS image is 32 samples x 2 KB, ALREADY TRANSPOSED: a record sets bit (hash - tile start) of
//     its sample's row (one ds_or), and the block leaves as 32 runs of 2 KB contiguous bytes -- no hash-major image
//     in HBM, no transpose pass;
//   * what needs all samples of a row -- the recurrence for recurrence-min > 1 or share-min (merge.hpp:458-467,
//     491-510) -- is counted first, by the same workgroup over the same tile: every sample's records of the tile
//     into a u16 per row in LDS (32 KB), two blocks of samples per memory round trip; the second walk over the tile's
//     records (2.4 MB for 2500 samples) is served by the L2;
//   * the walks are bound by memory latency, not bandwidth (one workgroup per CU: 156 KB of LDS): the bits walk requests the
//     NEXT block's records before it uses this block's, so a block's round trip hides behind the previous block's work.
// Input is read once from HBM, the matrix is written once.
// Round 4: (1) when the thresholds a row's recurrence is compared with are all <= 1 (recurrence-min <= 1 and share-min 1: BASELINE
// configs[3]) the recurrence is ONE BIT per row -- "a sample holds this hash with a solid count" -- set with ds_or: 2-3 KB of LDS
// instead of 32 KB of u16 counters; (2) a sample's cursor is advanced by the bits walk itself (it visits every sample once per
// tile), so the second cursor array and the recurrence walk's cursor arithmetic are gone: 20 bytes per sample of tables instead of
// 24; (3) with that room a tile is 24576 rows (a 96 KB image) when the one-bit recurrence applies: what a tile costs beside its
// bytes -- 2500 samples' cursors and tables, the loads a round requests beyond the tile's end -- is paid a third less often.
// (4) ONE walk instead of two when the recurrence is one bit (the kernel moves 9.9 GB for configs[3]'s 7.2 GB at 4.5 TB/s -- it is
// bound by its traffic, and a quarter of that is the tile's records read a second time, long gone from the L2).  The question the
// first walk answers for a non-solid record is "does ANY sample hold this hash with a solid count?".  The single walk answers it
// from the bit map as it fills: a solid record sets its row's bit; a non-solid record whose row's bit is set is rescued on the
// spot; one whose bit is not set YET is put aside -- (sample, row, count), 8 bytes, in the workgroup's scratch in HBM -- and looked at
// again when the tile's last sample has been walked: bit set by then -> rescued late (a global atomic-or into the sample's row, which
// left the LDS long ago; its statistics), else dropped.  On cohort data the map is complete after the first blocks of samples, so
// what is put aside is those blocks' non-solid records and the hashes nobody holds solid: ~1 % of the records.  A tile that would
// put more aside than the scratch holds flags its task; the driver runs flagged tasks again with two walks (kmx_result_wait).
#include "kmx_dev.hpp"
#include <algorithm>
#include <cstdio>
#include <cstring>

namespace kmx {

// (round 6: 768 threads -- 48 samples a block, twelve waves a CU instead of eight, rounds of 7 x 16 / 5 x 16 records a sample: 2.30 -> 2.04 ms
//  for configs[3] on one box, 1.97 -> 1.88 on another (profiles/r06_bft_variants.txt); 1024 threads with rounds that fit their registers: 1.93-1.95)
#ifndef KMX_BT_TPB
#define KMX_BT_TPB 768
#endif
constexpr int BT_TPB = KMX_BT_TPB;
#ifndef KMX_BT_G
#define KMX_BT_G 16
#endif
constexpr int BT_G = KMX_BT_G;             // lanes per sample (16: a sample's ~80 records of a tile are 1 KB, read by 16 adjacent lanes)
constexpr int BT_NB = BT_TPB / BT_G;       // samples per block (32)
#ifndef KMX_BT_IMG_KB
#define KMX_BT_IMG_KB 66
#endif
constexpr int BT_RT = KMX_BT_IMG_KB * 1024 * 8 / BT_NB;   // hash rows per tile: a 66 KB image (11264 rows of 48 samples) ...
static_assert((KMX_BT_IMG_KB * 1024 * 8) % BT_NB == 0 && BT_RT % 256 == 0, "a tile is a whole number of 256-row steps (bft_fit_rows)");
#ifndef KMX_BT_IMG1_KB
#define KMX_BT_IMG1_KB 96
#endif
constexpr int BT_RT1 = KMX_BT_IMG1_KB * 1024 * 8 / BT_NB; // ... 16384 rows (96 KB) when a row's recurrence is one bit (or not needed at all)
static_assert((KMX_BT_IMG1_KB * 1024 * 8) % BT_NB == 0 && BT_RT1 % 256 == 0, "a tile is a whole number of 256-row steps");
#ifndef KMX_BT_UNR
#define KMX_BT_UNR 5
#endif
#ifndef KMX_BT_UNR1
#define KMX_BT_UNR1 7
#endif
#ifndef KMX_BT_RB
#define KMX_BT_RB 2
#endif
constexpr int BT_RB = KMX_BT_RB;           // blocks of samples per round trip of the recurrence walk
template <int UNR> struct BtRound { u64 hh[UNR]; u32 cc[UNR]; };      // UNR record loads in flight per lane and round (8; 10 for the larger tile: ~117 records of a sample in it)
constexpr int BT_LDS = 160 * 1024 - 128;                  // dynamic LDS a workgroup may ask for (one workgroup per CU)
// LDS of a launch: the image (32 samples x rt / 8 bytes), the recurrences (rt / 8 bytes as bits, 2 rt as u16), the per-sample
// tables (20 B per sample: list address, range end, soft-min, cursor; 4 B when only the cursors fit)
__host__ __device__ inline u32 bt_img_bytes(u32 rt) { return (u32)BT_NB * (rt / 8); }
__host__ __device__ inline u32 bt_rec_bytes(u32 rt, bool bits) { return bits ? rt / 8 : 2 * rt; }

// a workgroup barrier that waits for the LDS only: __syncthreads() also waits until every global store and atomic of the wave is
// acknowledged (s_waitcnt vmcnt(0)) -- here that is the 64 KB of image a block has just sent out, per pass.  What the barriers
// order is LDS (image, recurrences, cursors); a store holds its data once its ds_read has returned.
__device__ __forceinline__ void bt_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef KMX_PHASE_PROF
__device__ u64 kmx_bft_prof[8];
#define BTPH(i) do { const long long n_ = clock64(); bpt[i] += n_ - bpc; bpc = n_; } while (0)
#else
#define BTPH(i) do {} while (0)
#endif
// rt_max: the largest tile of the launch's tasks; flags bit 0: every task's recurrence is one bit (or none) -- the two size the LDS
// regions --, bit 1: the per-sample tables of the launch's largest task fit beside them
template <int UNR>
__global__ __launch_bounds__(BT_TPB)      // (one workgroup per CU: the LDS decides)
void k_merge_bft(const TaskDev* __restrict__ tasks, const uint2* __restrict__ items, u32 n_items, u32* ticket, u32 rt_max, u32 flags,
                 u64* __restrict__ rem_all, u32 rem_cap)
{
  const u32 rec_bits = flags & 1u;
  u64* const rem = rem_all ? rem_all + (u64)blockIdx.x * rem_cap : nullptr;      // this workgroup's room for the records put aside (single walk)
  __shared__ u32 nrem;
#ifdef KMX_PHASE_PROF
  long long bpt[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long bpc = clock64();
#endif
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  u32* const img = lds;                                                 // [sample][row / 32]
  u32* const rec = lds + bt_img_bytes(rt_max) / 4;                      // the tile's recurrences: a bit or a u16 per row, when needed
  u32* const curs = rec + bt_rec_bytes(rt_max, rec_bits != 0) / 4;      // [N]: every sample's cursor (then its tables)
  __shared__ u32 bc;
  typedef BtRound<UNR> Round;
  constexpr u32 S = UNR * BT_G;                                         // records a round asks for, per sample
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (u32 t = tid; t < bt_img_bytes(rt_max) / 16; t += BT_TPB) reinterpret_cast<uint4*>(img)[t] = make_uint4(0, 0, 0, 0);      // (every block leaves it zero behind it)
  for (;;) {
    if (tid == 0) bc = atomicAdd(ticket, 1u);
    bt_barrier();
    const u32 item = bc;
    bt_barrier();
    if (item >= n_items) {
#ifdef KMX_PHASE_PROF
      if (tid == 64) for (int i = 0; i < 8; i++) atomicAdd(&kmx_bft_prof[i], (u64)bpt[i]);
#endif
      return;
    }
    BTPH(0);
    const TaskDev& T = tasks[items[item].x];
    const u32 N = T.N, N8 = (N + 7u) & ~7u, nblk = (N8 + BT_NB - 1) / BT_NB;
    const u32 range = items[item].y;
    const u32 rec_min = T.rec_min, share_min = T.share_min;
    const bool two_pass = rec_min > 1 || share_min > 0;
    const bool rec1 = max(rec_min, share_min) <= 1u;      // the thresholds a row's recurrence meets are <= 1: one bit per row will do
    const bool opt = two_pass && rec1 && rem != nullptr;  // ... and one walk (no scratch: a re-run with two)
    const u64 W = T.upper - T.lower + 1, W8 = (W + 7) & ~7ULL;
    const u32 rt = T.rt;                       // hash rows per tile (<= rt_max, a multiple of 64)
    const u32 rw = rt / 32;                    // image words per sample
    const u64 tiles = (W + rt - 1) / rt;
    const u64 tiles_per = (tiles + T.c - 1) / T.c;
    const u64 tile0 = (u64)range * tiles_per, tile1 = min(tiles, tile0 + tiles_per);
    const float tile_share = (float)rt / (float)W;      // a list's records are spread evenly over the window: this share of them per tile
    u32* const cur = curs;
    // (while they fit -- ~3000 samples -- the lists' addresses, lengths and soft-mins sit in LDS too: a walk over a block then starts
    //  with its record loads instead of a round trip for their addresses)
    const bool meta = (flags & 2u) != 0;
    u64* const mbase = reinterpret_cast<u64*>(curs + (size_t)N + (N & 1u));
    u32* const mend = reinterpret_cast<u32*>(mbase + N);
    u32* const msm = mend + N;
    // Where every list enters the item's first tile is found HERE (round 4; a kernel of its own before: k_range_bounds_bf, 0.25 ms
    // per launch of configs[3] for 1.6 M searches, against 2.1 ms of merge).  An item is a RUN of consecutive tiles (the host cuts a
    // task into about as many items as there are workgroups): one search per list and item, the cursors carry from tile to tile.
    // Window hashes are uniform in [lower, upper], so an interpolation search takes ~5 probes; a thread runs the searches of its
    // samples side by side (their dependent loads overlap).
    {
      const u64 q0 = T.lower + tile0 * rt;
#ifndef KMX_BT_SPT
#define KMX_BT_SPT 5
#endif
      constexpr int SPT = KMX_BT_SPT;
      for (u32 i0 = tid; i0 < N; i0 += SPT * BT_TPB) {
        u32 lo[SPT], hi[SPT], n[SPT]; u64 klo[SPT], khi[SPT]; const u8* base[SPT];
#pragma unroll
        for (int x = 0; x < SPT; x++) {
          const u32 i = i0 + x * BT_TPB;
          n[x] = i < N ? T.len[i] : 0u; base[x] = i < N ? T.recs[i] : nullptr;
          lo[x] = 0; hi[x] = q0 <= T.lower ? 0u : n[x]; klo[x] = T.lower; khi[x] = T.upper + 1;      // (the window's first hash: every record is at or above it)
        }
        for (u32 step = 0; ; step++) {
          bool open = false;
#pragma unroll
          for (int x = 0; x < SPT; x++) open |= lo[x] < hi[x];
          if (!open) break;
          u32 mid[SPT]; u64 k[SPT];
#pragma unroll
          for (int x = 0; x < SPT; x++) {
            mid[x] = lo[x]; k[x] = 0;
            if (lo[x] < hi[x]) {
              // (every third probe a plain bisection: no input costs more than 3 log n)
              if (step % 3 == 2 || khi[x] <= klo[x] || q0 <= klo[x]) mid[x] = lo[x] + ((hi[x] - lo[x]) >> 1);
              else { const float f = (float)(q0 - klo[x]) / (float)(khi[x] - klo[x]); mid[x] = lo[x] + min(hi[x] - lo[x] - 1u, (u32)(f * (float)(hi[x] - lo[x]))); }
              k[x] = load_key<1>(base[x] + (u64)mid[x] * 12).w[0];
            }
          }
#pragma unroll
          for (int x = 0; x < SPT; x++) if (lo[x] < hi[x]) { if (k[x] < q0) { lo[x] = mid[x] + 1; klo[x] = k[x]; } else { hi[x] = mid[x]; khi[x] = k[x]; } }
        }
#pragma unroll
        for (int x = 0; x < SPT; x++) {
          const u32 i = i0 + x * BT_TPB;
          if (i < N) { cur[i] = lo[x]; if (meta) { mbase[i] = (u64)(uintptr_t)base[x]; mend[i] = n[x]; msm[i] = T.soft_min[i]; } }
        }
      }
    }
    BTPH(1);
    const u32 r = (u32)tid & (BT_G - 1);
    u32* const myrow = img + ((u32)tid / BT_G) * rw;
    bt_barrier();
    for (u64 tile = tile0; tile < tile1; tile++) {
      const u64 tlo = T.lower + tile * rt;
      const u64 rows = min((u64)rt, T.upper + 1 - tlo), thi = tlo + rows;
      // a round: the sample's records [b, min(b + S, lim)), lane r's slots b + r + q * BT_G -- the loads issued together
      auto issue = [&](const u8* base, u32 b, u32 lim, Round& d) {
#pragma unroll
        for (int q = 0; q < UNR; q++) {
          const u32 ix = b + r + q * BT_G;
          d.hh[q] = ~0ULL; d.cc[q] = 0;
          if (ix < lim) { gu32* p = (gu32*)(uintptr_t)(base + (u64)ix * 12); d.hh[q] = (u64)p[0] | ((u64)p[1] << 32); d.cc[q] = p[2]; }
        }
      };
      // a sample's list, soft-min and cursor; lim: where its FIRST round of the tile stops asking -- the records a tile is expected to
      // hold (the list's share of the window) plus two standard deviations: what lies beyond is most likely the next tile's, and a
      // load for it is traffic for nothing (a round asks for S = UNR x 16 records, a tile holds ~100 of a sample's).  The few samples
      // with more get a second round (walk below).
      auto sample_of = [&](u32 li, const u8*& base, u32& e, u32& sm, u32& start, u32& lim) {
        base = meta ? (const u8*)(uintptr_t)mbase[li] : T.recs[li];
        e = meta ? mend[li] : T.len[li]; sm = meta ? msm[li] : T.soft_min[li]; start = cur[li];
        const float m = (float)e * tile_share;
        lim = min(e, start + min(S, (u32)(m + 2.0f * __builtin_sqrtf(m)) + 8u));
      };
      // The walk over a sample's records of the tile (written out twice below: a functor that updates the caller's counters ends up
      // in scratch memory).  In-tile records are a prefix of what a round loaded (a list ascends): the walk goes on while a round was
      // in the tile to its last record and the list has more.  `next` = the sample's cursor behind the tile, the same in all its lanes.
#define BT_WALK_NEXT_ROUND(d, base, b, next, lim, e)                                                        \
          {                                                                                                \
            _Pragma("unroll") for (int off = 1; off < BT_G; off <<= 1) next = max(next, (u32)__shfl_xor(next, off)); \
            const u32 got = min(b + S, lim);      /* the round loaded [b, got) */                          \
            if (next < got || got >= e) break;    /* a loaded record lies beyond the tile, or the list is at its end */ \
            b = got; lim = e;                                                                              \
            issue(base, b, lim, d);                                                                        \
          }
      if (opt) {
        for (u32 t = tid; t < rt / 32; t += BT_TPB) rec[t] = 0;
        if (tid == 0) nrem = 0;
        bt_barrier();
      }
      // ---- two walks: the rows' recurrences first -- every sample's records of the tile, BT_RB blocks of samples per memory round trip ----
      if (two_pass && !opt) {
        for (u32 t = tid; t < (rec1 ? rt / 32 : rt / 2); t += BT_TPB) rec[t] = 0;
        bt_barrier();
        for (u32 blk = 0; blk < nblk; blk += BT_RB) {
          Round d[BT_RB]; const u8* base[BT_RB]; u32 e[BT_RB], sm[BT_RB], st[BT_RB], lm[BT_RB], li[BT_RB];
#pragma unroll
          for (int x = 0; x < BT_RB; x++) {
            li[x] = (blk + x) * BT_NB + (u32)tid / BT_G;
            base[x] = nullptr; e[x] = 0; sm[x] = 0; st[x] = 0; lm[x] = 0;
            if (li[x] < N) { sample_of(li[x], base[x], e[x], sm[x], st[x], lm[x]); issue(base[x], st[x], lm[x], d[x]); }
          }
#pragma unroll
          for (int x = 0; x < BT_RB; x++) {
            if (li[x] >= N) continue;
            u32 b = st[x], next = st[x], lim = lm[x];
            for (;;) {
#pragma unroll
              for (int q = 0; q < UNR; q++) {
                if (!(d[x].hh[q] < thi)) continue;
                next = b + r + q * BT_G + 1;
                if (d[x].cc[q] >= sm[x]) {
                  const u32 row = (u32)(d[x].hh[q] - tlo);
                  if (rec1) atomicOr(&rec[row >> 5], 1u << (row & 31u));
                  else atomicAdd(&rec[row >> 1], 1u << ((row & 1u) * 16));
                }
              }
              BT_WALK_NEXT_ROUND(d[x], base[x], b, next, lim, e[x])
            }
          }
        }
        bt_barrier();
      }
      // ---- the bits, a block of samples at a time; the next block's records are requested before this block's are used ----
      Round dn; const u8* nbase = nullptr; u32 ne = 0, nsm = 0, nst = 0, nlm = 0;
      { const u32 li0 = (u32)tid / BT_G; if (li0 < N) { sample_of(li0, nbase, ne, nsm, nst, nlm); issue(nbase, nst, nlm, dn); } }
      const u32 nby = (u32)((min(tlo + (u64)rt, T.lower + W8) - tlo) >> 3);      // bytes of a sample's row the tile holds (the last tile carries the pad bits of ceil8(W))
      const u64 row_bytes = W8 >> 3, tb = (tlo - T.lower) >> 3;
      for (u32 blk = 0; blk < nblk; blk++) {
        const u32 col0 = blk * BT_NB;
        const u32 nrows_out = min((u32)BT_NB, N8 - col0);               // rows of the result this block writes (the padding rows too)
        const u32 li = col0 + (u32)tid / BT_G;
        const bool on = li < N;
        BTPH(2);
        Round d = dn; const u8* const base = nbase; const u32 e = ne, sm = nsm, start = nst, lim = nlm;
        { const u32 ln = li + BT_NB; if (blk + 1 < nblk && ln < N) { sample_of(ln, nbase, ne, nsm, nst, nlm); issue(nbase, nst, nlm, dn); } }
        if (on) {      // (the image is zero: the wave cleared its rows when it sent the previous block out)
          u32 uwo = 0, nresc = 0; u64 two = 0, tresc = 0;
          u32 b = start, next = start, lm = lim;
          for (;;) {
            // (the rows' recurrences are all requested before the first is used)
            u32 rcw[UNR];
#pragma unroll
            for (int q = 0; q < UNR; q++) { const bool in = d.hh[q] < thi; const u32 row = in ? (u32)(d.hh[q] - tlo) : 0u; rcw[q] = two_pass ? rec[rec1 ? row >> 5 : row >> 1] : 0u; }
#pragma unroll
            for (int q = 0; q < UNR; q++) {
              if (!(d.hh[q] < thi)) continue;
              next = b + r + q * BT_G + 1;
              const u32 c = d.cc[q], row = (u32)(d.hh[q] - tlo);
              const bool solid = c >= sm;
              const u32 rc = rec1 ? (rcw[q] >> (row & 31u)) & 1u : (rcw[q] >> ((row & 1u) * 16)) & 0xFFFFu;
              u32 outc = 0;
              if (opt) {
                // one walk: the map says whether a solid record of the row has been seen SO FAR (rcw was read before this round's
                // records set their bits: a stale "no" only puts the record aside, where the complete map decides)
                if (solid) { outc = c; uwo++; two += c; atomicOr(&rec[row >> 5], 1u << (row & 31u)); }
                else if (rc) { outc = c; nresc++; tresc += c; }
                else {
                  const u32 ps = atomicAdd(&nrem, 1u);
                  if (ps < rem_cap) rem[ps] = (u64)c | ((u64)row << 32) | ((u64)li << 48);
                }
                if (outc) atomicOr(&myrow[row >> 5], 1u << (row & 31u));
                continue;
              }
              if (solid) { outc = c; uwo++; two += c; }
              else if (share_min && rc >= share_min) { outc = c; nresc++; tresc += c; }      // rescued (merge.hpp:491-510)
              const bool keep = two_pass ? (rc >= rec_min) : (solid || rec_min == 0);
              if (keep && outc) atomicOr(&myrow[row >> 5], 1u << (row & 31u));
            }
            BT_WALK_NEXT_ROUND(d, base, b, next, lm, e)
          }
#pragma unroll
          for (int off = 1; off < BT_G; off <<= 1) {
            uwo += __shfl_xor(uwo, off); nresc += __shfl_xor(nresc, off);
            two += shfl_xor_u64(two, off); tresc += shfl_xor_u64(tresc, off);
          }
          if (r == 0) {
            cur[li] = next;      // (this walk visits a sample once per tile: nobody reads its cursor again before the next tile)
            if (uwo | two) { atomicAdd(&T.stats[2 * (u64)N + li], (u64)uwo); atomicAdd(&T.stats[4 * (u64)N + li], two); }
            if (nresc) { atomicAdd(&T.stats[1 * (u64)N + li], (u64)nresc); atomicAdd(&T.stats[5 * (u64)N + li], tresc); }
          }
        }
        // (round 3) A sample's image row is written by the 16 lanes that walk the sample -- lanes of ONE wave: the wave's own rows
        // (64 / BT_G samples of the block) need no workgroup barrier between the walk and the way out, only the wave's own LDS
        // operations in order.  The waves drift apart over a tile's blocks instead of meeting twice per block (158 barriers per
        // tile before), and a wave that drew long lists no longer holds the other seven.
        BTPH(3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // out: sample s = col0 + j gets bytes [tb, tb + nby) of its row -- the wave sends its samples' rows out and leaves them zero
        // for its next block (the padding rows of the last block too)
        constexpr u32 SPW = 64 / BT_G;                                                  // samples per wave
#ifdef KMX_BT_OUT8
        if (false) {
#else
        if (((row_bytes | tb | (u64)nby) & 15u) == 0) {
#endif
          // 16 bytes per lane and step, the same word of the wave's SPW rows together: their LDS reads in flight at once
          const u32 w16 = nby / 16;
          for (u32 t = (u32)lane; t < w16; t += 64) {
            static_assert(SPW <= 8, "a wave's rows");
            uint4 v0, v1, v2, v3, v4, v5, v6, v7;      // (named, not an array: an array of them went to scratch memory)
#define BT_TAKE(x, v) if ((x) < SPW) { uint4* const s4 = reinterpret_cast<uint4*>(img + (wave * SPW + (x)) * rw) + t; v = *s4; *s4 = make_uint4(0, 0, 0, 0); }
#define BT_SEND(x, v) if ((x) < SPW && wave * SPW + (x) < nrows_out) reinterpret_cast<uint4*>(T.out + (u64)(col0 + wave * SPW + (x)) * row_bytes + tb)[t] = v;
            BT_TAKE(0, v0) BT_TAKE(1, v1) BT_TAKE(2, v2) BT_TAKE(3, v3) BT_TAKE(4, v4) BT_TAKE(5, v5) BT_TAKE(6, v6) BT_TAKE(7, v7)
            BT_SEND(0, v0) BT_SEND(1, v1) BT_SEND(2, v2) BT_SEND(3, v3) BT_SEND(4, v4) BT_SEND(5, v5) BT_SEND(6, v6) BT_SEND(7, v7)
#undef BT_TAKE
#undef BT_SEND
          }
        } else {
          for (u32 j = wave * SPW; j < min(nrows_out, (wave + 1) * SPW); j++) {
            u8* dst = T.out + (u64)(col0 + j) * row_bytes + tb;
            const u32* src = img + j * rw;
            if ((W8 & 63u) == 0) {
              for (u32 t = (u32)lane; t * 8 < nby; t += 64) reinterpret_cast<u64*>(dst)[t] = reinterpret_cast<const u64*>(src)[t];
            } else {
              for (u32 t = lane; t < nby; t += 64) dst[t] = (u8)(src[t >> 2] >> ((t & 3u) * 8));
            }
          }
          for (u32 j = wave * SPW; j < (wave + 1) * SPW; j++) {
            uint4* const z = reinterpret_cast<uint4*>(img + j * rw);
            for (u32 t = (u32)lane; t < rw / 4; t += 64) z[t] = make_uint4(0, 0, 0, 0);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
      BTPH(4);
      if (opt) {
        // the records put aside, now that the map is complete.  (__syncthreads: the image blocks this workgroup stored are out --
        // the late bits are atomics on the same words)
        __syncthreads();
        const u32 nr = nrem;
        if (nr > rem_cap) { if (tid == 0) atomicOr(&T.ctrl[2], (u64)ERR_FALLBACK); }
        else {
          for (u32 j = tid; j < nr; j += BT_TPB) {
            const u64 v = rem[j];
            const u32 c = (u32)v, row = (u32)(v >> 32) & 0xFFFFu, li = (u32)(v >> 48);
            if (!((rec[row >> 5] >> (row & 31u)) & 1u)) continue;      // nobody holds it solid: dropped
            atomicAdd(&T.stats[1 * (u64)N + li], 1ULL); atomicAdd(&T.stats[5 * (u64)N + li], (u64)c);
            if (c) {
              const u64 byte = (u64)li * row_bytes + tb + (row >> 3);      // T.out is 256-byte aligned: the dword that holds the byte
              atomicOr(reinterpret_cast<u32*>(T.out + (byte & ~3ULL)), 1u << ((u32)(byte & 3ULL) * 8u + (row & 7u)));
            }
          }
        }
      }
      bt_barrier();      // (every wave is through the tile: the recurrences and the cursors of the next tile are the workgroup's)
      BTPH(5);
    }
  }
}

// the most rows a tile of a task may have: the larger tile when its rows' recurrences are one bit each or not needed (thresholds <= 1)
u32 bft_tile_rows(u32 rec_min, u32 share_min) { return std::max(rec_min, share_min) <= 1u ? (u32)BT_RT1 : (u32)BT_RT; }
// the most rows a tile may have when the launch's largest task has max_n samples: image + recurrences + at least the cursors fit the LDS
u32 bft_fit_rows(u32 max_n, bool bits)
{
  for (u32 rt = bits ? (u32)BT_RT1 : (u32)BT_RT; rt >= 1024u; rt -= 256u)
    if ((size_t)bt_img_bytes(rt) + bt_rec_bytes(rt, bits) + (size_t)max_n * 4 + 16 <= (size_t)BT_LDS) return rt;
  return 1024u;
}
u32 bft_round_records(bool wide) { return (u32)(wide ? KMX_BT_UNR1 : KMX_BT_UNR) * BT_G; }
u32 bft_block_lists() { return BT_NB; }
#ifdef KMX_PHASE_PROF
void bft_phase_prof_dump()
{
  u64 h[8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(kmx_bft_prof), sizeof(h)) != hipSuccess) return;
  u64 tot = 0; for (int i = 0; i < 8; i++) tot += h[i];
  static const char* nm[8] = {"ticket", "item prologue (bounds)", "image out + clear (prev block)", "walk a block", "tile end wait", "put-aside records + barrier", "-", "-"};
  for (int i = 0; i < 6; i++) fprintf(stderr, "[bft] %-32s %6.2f%%  %llu\n", nm[i], tot ? 100.0 * h[i] / tot : 0.0, h[i]);
  memset(h, 0, sizeof(h));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(kmx_bft_prof), h, sizeof(h));
}
#endif
u32 bft_max_lists() { return (u32)(BT_LDS - (int)bt_img_bytes(BT_RT) - (int)bt_rec_bytes(BT_RT, false) - 16) / 4; }      // the cursors of every sample must fit the LDS
// rt_max: the largest tile among the launch's tasks; rec_bits: all of them take the one-bit recurrence (or none)
// rem / rem_cap: scratch of grid_x x rem_cap u64 for the single walk (null: two walks)
// wide: a tile holds more of a sample's records than a round of KMX_BT_UNR x 16 covers with margin: rounds of KMX_BT_UNR1 x 16
hipError_t launch_merge_bft(const TaskDev* tasks, const uint2* items, u32 n_items, u32* ticket, u32 grid_x, u32 max_n, u32 rt_max, bool rec_bits, bool wide, u64* rem, u32 rem_cap, hipStream_t st)
{
  const size_t fixed = (size_t)bt_img_bytes(rt_max) + bt_rec_bytes(rt_max, rec_bits);
  const bool meta = fixed + (size_t)max_n * 20 + 16 <= (size_t)BT_LDS;
  const size_t dyn = fixed + (meta ? (size_t)max_n * 20 + 16 : (size_t)max_n * 4 + 16);
  if (dyn > (size_t)BT_LDS) return hipErrorInvalidValue;
  const u32 flags = (rec_bits ? 1u : 0u) | (meta ? 2u : 0u);
  hipError_t e;
  if (wide) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_merge_bft<KMX_BT_UNR1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_merge_bft<KMX_BT_UNR1>, dim3(grid_x), dim3(BT_TPB), dyn, st, tasks, items, n_items, ticket, rt_max, flags, rem, rem_cap);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_merge_bft<KMX_BT_UNR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_merge_bft<KMX_BT_UNR>, dim3(grid_x), dim3(BT_TPB), dyn, st, tasks, items, n_items, ticket, rt_max, flags, rem, rem_cap);
  }
  return hipGetLastError();
}

}  // namespace kmx
