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
#include "kmx_dev.hpp"

namespace kmx {

constexpr int BT_TPB = 512;
#ifndef KMX_BT_G
#define KMX_BT_G 16
#endif
constexpr int BT_G = KMX_BT_G;             // lanes per sample (16: a sample's ~80 records of a tile are 1 KB, read by 16 adjacent lanes)
constexpr int BT_NB = BT_TPB / BT_G;       // samples per block (32)
#ifndef KMX_BT_IMG_KB
#define KMX_BT_IMG_KB 64
#endif
constexpr int BT_RT = KMX_BT_IMG_KB * 1024 * 8 / BT_NB;   // hash rows per tile: a 64 KB image (16384 rows of 32 samples)
constexpr int BT_RW = BT_RT / 32;          // image words per sample
#ifndef KMX_BT_UNR
#define KMX_BT_UNR 8
#endif
constexpr int BT_UNR = KMX_BT_UNR;         // record loads in flight per lane and round
#ifndef KMX_BT_RB
#define KMX_BT_RB 2
#endif
constexpr int BT_RB = KMX_BT_RB;           // blocks of samples per round trip of the recurrence walk
struct BtRound { u64 hh[BT_UNR]; u32 cc[BT_UNR]; };
constexpr int BT_DYN_MAX = 160 * 1024 - KMX_BT_IMG_KB * 1024 - BT_RT * 2 - 64;     // LDS left beside the image and the recurrences
constexpr int BT_META_MAX = BT_DYN_MAX - 16;           // ... for the per-sample tables (24 B per sample; 8 B when only the cursors fit)

// a workgroup barrier that waits for the LDS only: __syncthreads() also waits until every global store and atomic of the wave is
// acknowledged (s_waitcnt vmcnt(0)) -- here that is the 64 KB of image a block has just sent out, per pass.  What the barriers
// order is LDS (image, recurrences, cursors); a store holds its data once its ds_read has returned.
__device__ __forceinline__ void bt_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(BT_TPB)      // (one workgroup per CU: the LDS decides)
void k_merge_bft(const TaskDev* __restrict__ tasks, const uint2* __restrict__ items, u32 n_items, u32* ticket)
{
  __shared__ __attribute__((aligned(16))) u32 img[BT_NB * BT_RW];      // [sample][row / 32]: 64 KB
  __shared__ __attribute__((aligned(16))) u32 rec[BT_RT / 2];           // the tile's recurrences (u16), when needed
  __shared__ u32 bc;
  extern __shared__ u32 curs[];                                         // [2][N]: every sample's cursor, and its cursor behind the tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (u32 t = tid; t < BT_NB * BT_RW / 4; t += BT_TPB) reinterpret_cast<uint4*>(img)[t] = make_uint4(0, 0, 0, 0);      // (every block leaves it zero behind it)
  for (;;) {
    if (tid == 0) bc = atomicAdd(ticket, 1u);
    bt_barrier();
    const u32 item = bc;
    bt_barrier();
    if (item >= n_items) return;
    const TaskDev& T = tasks[items[item].x];
    const u32 N = T.N, N8 = (N + 7u) & ~7u, nblk = (N8 + BT_NB - 1) / BT_NB;
    const u32 range = items[item].y;
    const u32 rec_min = T.rec_min, share_min = T.share_min;
    const bool two_pass = rec_min > 1 || share_min > 0;
    const u64 W = T.upper - T.lower + 1, W8 = (W + 7) & ~7ULL;
    const u32 rt = T.rt;                       // hash rows per tile (BT_RT; the host may pass fewer -- a multiple of 64)
    const u64 tiles = (W + rt - 1) / rt;
    const u64 tiles_per = (tiles + T.c - 1) / T.c;
    const u64 tile0 = (u64)range * tiles_per, tile1 = min(tiles, tile0 + tiles_per);
    u32* const cur = curs;
    u32* const nxt = curs + N;
    // (up to ~3400 samples the lists' addresses, range ends and soft-mins sit in LDS too: a walk over a block then starts with
    //  its record loads instead of a round trip for their addresses)
    const bool meta = (u64)N * 24 <= (u64)BT_META_MAX;
    u64* const mbase = reinterpret_cast<u64*>(curs + 2 * (size_t)N + (N & 1u));
    u32* const mend = reinterpret_cast<u32*>(mbase + N);
    u32* const msm = mend + N;
    for (u32 i = tid; i < N; i += BT_TPB) {
      cur[i] = T.bounds[(u64)range * N + i];
      if (meta) { mbase[i] = (u64)(uintptr_t)T.recs[i]; mend[i] = T.bounds[(u64)(range + 1) * N + i]; msm[i] = T.soft_min[i]; }
    }
    const u32 r = (u32)tid & (BT_G - 1);
    u32* const myrow = img + ((u32)tid / BT_G) * BT_RW;
    bt_barrier();
    for (u64 tile = tile0; tile < tile1; tile++) {
      const u64 tlo = T.lower + tile * rt;
      const u64 rows = min((u64)rt, T.upper + 1 - tlo), thi = tlo + rows;
      // a sample's first BT_UNR records per lane from its cursor on: the loads of one round, issued together
      auto issue = [&](const u8* base, u32 idx0, u32 e, BtRound& d) {
#pragma unroll
        for (int q = 0; q < BT_UNR; q++) {
          const u32 ix = idx0 + q * BT_G;
          d.hh[q] = ~0ULL; d.cc[q] = 0;
          if (ix < e) { gu32* p = (gu32*)(uintptr_t)(base + (u64)ix * 12); d.hh[q] = (u64)p[0] | ((u64)p[1] << 32); d.cc[q] = p[2]; }
        }
      };
      auto sample_of = [&](u32 li, const u8*& base, u32& e, u32& sm, u32& start) {
        base = meta ? (const u8*)(uintptr_t)mbase[li] : T.recs[li];
        e = meta ? mend[li] : T.bounds[(u64)(range + 1) * N + li]; sm = meta ? msm[li] : T.soft_min[li]; start = cur[li];
      };
      // ---- the rows' recurrences: every sample's records of the tile, BT_RB blocks of samples per memory round trip ----
      if (two_pass) {
        for (u32 t = tid; t < BT_RT / 2; t += BT_TPB) rec[t] = 0;
        bt_barrier();
        for (u32 blk = 0; blk < nblk; blk += BT_RB) {
          BtRound d[BT_RB]; const u8* base[BT_RB]; u32 e[BT_RB], sm[BT_RB], st[BT_RB], li[BT_RB];
#pragma unroll
          for (int x = 0; x < BT_RB; x++) {
            li[x] = (blk + x) * BT_NB + (u32)tid / BT_G;
            base[x] = nullptr; e[x] = 0; sm[x] = 0; st[x] = 0;
            if (li[x] < N) { sample_of(li[x], base[x], e[x], sm[x], st[x]); issue(base[x], st[x] + r, e[x], d[x]); }
          }
#pragma unroll
          for (int x = 0; x < BT_RB; x++) {
            if (li[x] >= N) continue;
            u32 next = st[x];
            bool stop = false;
            for (u32 idx0 = st[x] + r; ; ) {
              // (a list ascends and a slot past its end holds ~0: once a record is beyond the tile, so are the lane's later ones)
#pragma unroll
              for (int q = 0; q < BT_UNR; q++) {
                const bool in = d[x].hh[q] < thi;
                if (in) next = idx0 + q * BT_G + 1;
                if (in && d[x].cc[q] >= sm[x]) { const u32 row = (u32)(d[x].hh[q] - tlo); atomicAdd(&rec[row >> 1], 1u << ((row & 1u) * 16)); }
              }
              stop = !(d[x].hh[BT_UNR - 1] < thi);
              idx0 += BT_UNR * BT_G;
              if (stop || idx0 >= e[x]) break;
              issue(base[x], idx0, e[x], d[x]);      // (a sample with more than a round of records in the tile)
            }
#pragma unroll
            for (int off = 1; off < BT_G; off <<= 1) next = max(next, (u32)__shfl_xor(next, off));
            if (r == 0) nxt[li[x]] = next;
          }
        }
        bt_barrier();
      }
      // ---- the bits, a block of samples at a time; the next block's records are requested before this block's are used ----
      BtRound dn; const u8* nbase = nullptr; u32 ne = 0, nsm = 0, nst = 0;
      { const u32 li0 = (u32)tid / BT_G; if (li0 < N) { sample_of(li0, nbase, ne, nsm, nst); issue(nbase, nst + r, ne, dn); } }
      for (u32 blk = 0; blk < nblk; blk++) {
        const u32 col0 = blk * BT_NB;
        const u32 nrows_out = min((u32)BT_NB, N8 - col0);               // rows of the result this block writes (the padding rows too)
        const u32 li = col0 + (u32)tid / BT_G;
        const bool on = li < N;
        BtRound d = dn; const u8* const base = nbase; const u32 e = ne, sm = nsm, start = nst;
        { const u32 ln = li + BT_NB; if (blk + 1 < nblk && ln < N) { sample_of(ln, nbase, ne, nsm, nst); issue(nbase, nst + r, ne, dn); } }
        if (on) {      // (the image is zero: the previous block cleared it before its last barrier)
          u32 next = start, uwo = 0, nresc = 0; u64 two = 0, tresc = 0;
          bool stop = false;
          for (u32 idx0 = start + r; ; ) {
            // (as above: in-tile records are a prefix of the lane's slots; the rows' recurrences are all requested before the first is used)
            u32 rw[BT_UNR];
#pragma unroll
            for (int q = 0; q < BT_UNR; q++) { const bool in = d.hh[q] < thi; const u32 row = in ? (u32)(d.hh[q] - tlo) : 0u; rw[q] = two_pass ? rec[row >> 1] : 0u; }
#pragma unroll
            for (int q = 0; q < BT_UNR; q++) {
              if (!(d.hh[q] < thi)) continue;
              next = idx0 + q * BT_G + 1;
              const u32 c = d.cc[q], row = (u32)(d.hh[q] - tlo);
              const bool solid = c >= sm;
              const u32 rc = (rw[q] >> ((row & 1u) * 16)) & 0xFFFFu;
              u32 outc = 0;
              if (solid) { outc = c; uwo++; two += c; }
              else if (share_min && rc >= share_min) { outc = c; nresc++; tresc += c; }      // rescued (merge.hpp:491-510)
              const bool keep = two_pass ? (rc >= rec_min) : (solid || rec_min == 0);
              if (keep && outc) atomicOr(&myrow[row >> 5], 1u << (row & 31u));
            }
            stop = !(d.hh[BT_UNR - 1] < thi);
            idx0 += BT_UNR * BT_G;
            if (stop || idx0 >= e) break;
            issue(base, idx0, e, d);
          }
#pragma unroll
          for (int off = 1; off < BT_G; off <<= 1) {
            next = max(next, (u32)__shfl_xor(next, off));
            uwo += __shfl_xor(uwo, off); nresc += __shfl_xor(nresc, off);
            two += shfl_xor_u64(two, off); tresc += shfl_xor_u64(tresc, off);
          }
          if (r == 0) {
            if (!two_pass) nxt[li] = next;
            if (uwo | two) { atomicAdd(&T.stats[2 * (u64)N + li], (u64)uwo); atomicAdd(&T.stats[4 * (u64)N + li], two); }
            if (nresc) { atomicAdd(&T.stats[1 * (u64)N + li], (u64)nresc); atomicAdd(&T.stats[5 * (u64)N + li], tresc); }
          }
        }
        // (round 3) A sample's image row is written by the 16 lanes that walk the sample -- lanes of ONE wave: the wave's own rows
        // (64 / BT_G samples of the block) need no workgroup barrier between the walk and the way out, only the wave's own LDS
        // operations in order.  The waves drift apart over a tile's blocks instead of meeting twice per block (158 barriers per
        // tile before), and a wave that drew long lists no longer holds the other seven.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // out: sample s = col0 + j gets bytes [(tlo - lower) / 8, ...) of its row -- the wave writes its samples' rows, 8 bytes per lane
        const u32 nby = (u32)((min(tlo + (u64)rt, T.lower + W8) - tlo) >> 3);      // (the last tile carries the pad bits of ceil8(W))
        constexpr u32 SPW = 64 / BT_G;                                                  // samples per wave
        for (u32 j = wave * SPW; j < min(nrows_out, (wave + 1) * SPW); j++) {
          u8* dst = T.out + (u64)(col0 + j) * (W8 >> 3) + ((tlo - T.lower) >> 3);
          const u32* src = img + j * BT_RW;
          if ((W8 & 63u) == 0) {
            for (u32 t = (u32)lane; t * 8 < nby; t += 64) reinterpret_cast<u64*>(dst)[t] = reinterpret_cast<const u64*>(src)[t];
          } else {
            for (u32 t = lane; t < nby; t += 64) dst[t] = (u8)(src[t >> 2] >> ((t & 3u) * 8));
          }
        }
        // the wave's rows are zero again for its next block (the padding rows of the last block too)
        for (u32 j = wave * SPW; j < (wave + 1) * SPW; j++) {
          uint4* const z = reinterpret_cast<uint4*>(img + j * BT_RW);
          for (u32 t = (u32)lane; t < (u32)BT_RW / 4; t += 64) z[t] = make_uint4(0, 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
      bt_barrier();      // (every wave is through the tile: the cursors below, and the recurrences of the next tile, are the workgroup's)
      for (u32 i = tid; i < N; i += BT_TPB) cur[i] = nxt[i];
      bt_barrier();
    }
  }
}

u32 bft_tile_rows() { return BT_RT; }
u32 bft_block_lists() { return BT_NB; }
u32 bft_max_lists() { return (u32)(BT_DYN_MAX - 16) / 8; }      // the cursors of every sample must fit the LDS
hipError_t launch_merge_bft(const TaskDev* tasks, const uint2* items, u32 n_items, u32* ticket, u32 grid_x, u32 max_n, hipStream_t st)
{
  const size_t dyn = (u64)max_n * 24 <= (u64)BT_META_MAX ? (size_t)max_n * 24 + 16 : (size_t)max_n * 8 + 16;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_merge_bft), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_merge_bft, dim3(grid_x), dim3(BT_TPB), dyn, st, tasks, items, n_items, ticket);
  return hipGetLastError();
}

}  // namespace kmx
