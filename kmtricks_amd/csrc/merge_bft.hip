// merge_bft.hip -- hash-mode merge straight into the SAMPLE-major Bloom matrix (hash:bft:bin) on gfx950.
// Replaces km::HashMerger::next + write_as_bft (reference include/kmtricks/merge.hpp:441-517, 575-600, 631-644:
// write_as_bf into a temporary file, load it into a BitMatrix, transpose, dump): row s of the result is sample s's
// slice of its final Bloom filter (howde_utils.hpp:133-187), ceil8(W)/8 bytes, bit r = hash lower + r.
//
// k_merge_bf keeps a tile of hash rows x ALL samples in LDS; with thousands of samples a row is hundreds of bytes,
// a tile a hundred rows, and a sample has a record in it every other tile -- the kernel then spends its time
// re-reading the heads of 2500 lists.  Here the matrix is cut the other way:
//   * work item = (task, run of row tiles, block of 128 samples); a tile is 4096 hash rows, so a sample hands over
//     ~20 records per tile (4 adjacent lanes per sample, 8 loads in flight each) and its LDS image is
//     128 samples x 512 bytes, ALREADY TRANSPOSED: a record sets bit (hash - tile start) of its sample's row
//     (one ds_or), and the tile leaves as 128 runs of 512 contiguous bytes -- no hash-major image in HBM, no
//     transpose pass;
//   * what needs all samples of a row -- the recurrence for recurrence-min > 1 or share-min (merge.hpp:458-467,
//     491-510) -- comes from k_bf_rowrec, which runs first: same tiles, every sample's records of the tile counted
//     into a u16 per row in LDS, written to a W-entry array the sample blocks then read (2 bytes per row against the
//     ceil(N/8) of the matrix).
// Input is read once (twice when the recurrence pass is needed), the matrix is written once.
#include "kmx_dev.hpp"

namespace kmx {

constexpr int BT_TPB = 512;
#ifndef KMX_BT_G
#define KMX_BT_G 4
#endif
constexpr int BT_G = KMX_BT_G;             // lanes per sample
constexpr int BT_NB = BT_TPB / BT_G;       // samples per block (128)
constexpr int BT_RT = 64 * 1024 * 8 / BT_NB;   // hash rows per tile: a 64 KB image (4096 rows of 128 samples)
constexpr int BT_RW = BT_RT / 32;          // image words per sample
constexpr int BT_UNR = 8;                  // record loads in flight per lane

// ---- recurrence per hash row: number of samples in which the row's hash is solid (count >= soft-min) ----
__global__ __launch_bounds__(BT_TPB)
void k_bf_rowrec(const TaskDev* __restrict__ tasks, const uint2* __restrict__ items, u32 n_items)
{
  __shared__ u32 rec[BT_RT / 2];            // u16 counters, two per word
  extern __shared__ u32 cur[];              // [N] cursors of the range
  const u32 item = blockIdx.x;
  if (item >= n_items) return;
  const TaskDev& T = tasks[items[item].x];
  if (!(T.rec_min > 1 || T.share_min > 0)) return;      // (every row's fate is decided by the record alone)
  const u32 range = items[item].y, N = T.N;
  const int tid = threadIdx.x;
  const u64 W = T.upper - T.lower + 1;
  const u64 tiles = (W + BT_RT - 1) / BT_RT;
  const u64 tiles_per = (tiles + T.c - 1) / T.c;
  const u64 tile0 = (u64)range * tiles_per, tile1 = min(tiles, tile0 + tiles_per);
  u32 g = 1; while (g * 2 * N <= (u32)BT_TPB && g < 64) g <<= 1;
  const u32 lpp = BT_TPB / g, passes = (N + lpp - 1) / lpp, r = tid & (g - 1);
  for (u32 i = tid; i < N; i += BT_TPB) cur[i] = T.bounds[(u64)range * N + i];
  __syncthreads();
  for (u64 tile = tile0; tile < tile1; tile++) {
    const u64 tlo = T.lower + tile * BT_RT;
    const u64 rows = min((u64)BT_RT, T.upper + 1 - tlo), thi = tlo + rows;
    for (u32 t = tid; t < BT_RT / 2; t += BT_TPB) rec[t] = 0;
    __syncthreads();
    for (u32 ps = 0; ps < passes; ps++) {
      const u32 i = ps * lpp + tid / g;
      u32 next = 0;
      if (i < N) {
        const u8* base = T.recs[i];
        const u32 e = T.bounds[(u64)(range + 1) * N + i], sm = T.soft_min[i];
        const u32 start = cur[i];
        next = start;
        bool stop = false;
        for (u32 idx0 = start + r; idx0 < e && !stop; idx0 += BT_UNR * g) {
          u64 hh[BT_UNR]; u32 cc[BT_UNR];
#pragma unroll
          for (int q = 0; q < BT_UNR; q++) {
            const u32 ix = idx0 + q * g;
            hh[q] = ~0ULL; cc[q] = 0;
            if (ix < e) { gu32* p = (gu32*)(uintptr_t)(base + (u64)ix * 12); hh[q] = (u64)p[0] | ((u64)p[1] << 32); cc[q] = p[2]; }
          }
#pragma unroll
          for (int q = 0; q < BT_UNR; q++) {
            if (stop) continue;
            if (hh[q] >= thi) { stop = true; continue; }
            next = idx0 + q * g + 1;
            if (cc[q] >= sm) { const u32 row = (u32)(hh[q] - tlo); atomicAdd(&rec[row >> 1], 1u << ((row & 1u) * 16)); }
          }
        }
      }
      for (u32 off = 1; off < g; off <<= 1) next = max(next, (u32)__shfl_xor(next, (int)off));
      if (i < N && r == 0) cur[i] = next;
    }
    __syncthreads();
    u32* dst = reinterpret_cast<u32*>(T.rowrec + (tlo - T.lower));      // (tiles start at multiples of 4096 rows: aligned)
    for (u32 t = tid; t < (u32)((rows + 1) / 2); t += BT_TPB) dst[t] = rec[t];
    __syncthreads();
  }
}

// ---- the merge: (task, run of row tiles, block of 128 samples) -> the block's rows of the transposed matrix ----
__global__ __launch_bounds__(BT_TPB, 2)
void k_merge_bft(const TaskDev* __restrict__ tasks, const uint2* __restrict__ items, u32 n_items, u32* ticket)
{
  __shared__ __attribute__((aligned(16))) u32 img[BT_NB * BT_RW];      // [sample][row / 32]: 64 KB
  __shared__ __attribute__((aligned(16))) u32 rec[BT_RT / 2];           // the tile's recurrences (u16), when needed
  __shared__ u32 bc;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (;;) {
    if (tid == 0) bc = atomicAdd(ticket, 1u);
    __syncthreads();
    const u32 item = bc;
    __syncthreads();
    if (item >= n_items) return;
    const TaskDev& T = tasks[items[item].x];
    const u32 N = T.N, nblk = (((N + 7u) & ~7u) + BT_NB - 1) / BT_NB;
    const u32 range = items[item].y / nblk, blk = items[item].y - range * nblk;
    const u32 rec_min = T.rec_min, share_min = T.share_min;
    const bool two_pass = rec_min > 1 || share_min > 0;
    const u64 W = T.upper - T.lower + 1, W8 = (W + 7) & ~7ULL;
    const u64 tiles = (W + BT_RT - 1) / BT_RT;
    const u64 tiles_per = (tiles + T.c - 1) / T.c;
    const u64 tile0 = (u64)range * tiles_per, tile1 = min(tiles, tile0 + tiles_per);
    const u32 col0 = blk * BT_NB;
    const u32 nrows_out = min((u32)BT_NB, ((N + 7u) & ~7u) - col0);       // rows of the result this block writes (the padding rows too)
    const u32 li = col0 + (u32)tid / BT_G, r = (u32)tid & (BT_G - 1);
    const bool on = li < N;
    const u8* base = on ? T.recs[li] : nullptr;
    const u32 e = on ? T.bounds[(u64)(range + 1) * N + li] : 0u, sm = on ? T.soft_min[li] : 0u;
    u32 curp = on ? T.bounds[(u64)range * N + li] : 0u;                 // the sample's cursor (all its 4 lanes keep it)
    u32* const myrow = img + ((u32)tid / BT_G) * BT_RW;
    u32 uwo = 0, nresc = 0; u64 two = 0, tresc = 0;
    for (u64 tile = tile0; tile < tile1; tile++) {
      const u64 tlo = T.lower + tile * BT_RT;
      const u64 rows = min((u64)BT_RT, T.upper + 1 - tlo), thi = tlo + rows;
      for (u32 t = tid; t < BT_NB * BT_RW / 4; t += BT_TPB) reinterpret_cast<uint4*>(img)[t] = make_uint4(0, 0, 0, 0);
      if (two_pass) {
        const u32* src = reinterpret_cast<const u32*>(T.rowrec + (tlo - T.lower));
        for (u32 t = tid; t < (u32)((rows + 1) / 2); t += BT_TPB) rec[t] = src[t];
      }
      __syncthreads();
      u32 next = curp;
      bool stop = false;
      for (u32 idx0 = curp + r; idx0 < e && !stop; idx0 += BT_UNR * BT_G) {
        u64 hh[BT_UNR]; u32 cc[BT_UNR];
#pragma unroll
        for (int q = 0; q < BT_UNR; q++) {
          const u32 ix = idx0 + q * BT_G;
          hh[q] = ~0ULL; cc[q] = 0;
          if (ix < e) { gu32* p = (gu32*)(uintptr_t)(base + (u64)ix * 12); hh[q] = (u64)p[0] | ((u64)p[1] << 32); cc[q] = p[2]; }
        }
#pragma unroll
        for (int q = 0; q < BT_UNR; q++) {
          if (stop) continue;
          if (hh[q] >= thi) { stop = true; continue; }
          next = idx0 + q * BT_G + 1;
          const u32 c = cc[q], row = (u32)(hh[q] - tlo);
          const bool solid = c >= sm;
          u32 rc = 0;
          if (two_pass) rc = (rec[row >> 1] >> ((row & 1u) * 16)) & 0xFFFFu;
          u32 outc = 0;
          if (solid) { outc = c; uwo++; two += c; }
          else if (share_min && rc >= share_min) { outc = c; nresc++; tresc += c; }      // rescued (merge.hpp:491-510)
          const bool keep = two_pass ? (rc >= rec_min) : (solid || rec_min == 0);
          if (keep && outc) atomicOr(&myrow[row >> 5], 1u << (row & 31u));
        }
      }
#pragma unroll
      for (int off = 1; off < BT_G; off <<= 1) next = max(next, (u32)__shfl_xor(next, off));
      curp = next;
      __syncthreads();
      // out: sample s = col0 + j gets bytes [(tlo - lower) / 8, + rows8 / 8) of its row -- a wave per sample, 8 bytes per lane
      const u32 nby = (u32)((min(tlo + (u64)BT_RT, T.lower + W8) - tlo) >> 3);      // (the last tile carries the pad bits of ceil8(W))
      for (u32 j = wave; j < nrows_out; j += BT_TPB / 64) {
        u8* dst = T.out + (u64)(col0 + j) * (W8 >> 3) + ((tlo - T.lower) >> 3);
        const u32* src = img + j * BT_RW;
        if ((W8 & 63u) == 0) {
          if ((u32)lane * 8 < nby) reinterpret_cast<u64*>(dst)[lane] = reinterpret_cast<const u64*>(src)[lane];
        } else {
          for (u32 t = lane; t < nby; t += 64) dst[t] = (u8)(src[t >> 2] >> ((t & 3u) * 8));
        }
      }
      __syncthreads();
    }
    // statistics of my sample (its 4 lanes add up)
#pragma unroll
    for (int off = 1; off < BT_G; off <<= 1) {
      uwo += __shfl_xor(uwo, off); nresc += __shfl_xor(nresc, off);
      two += shfl_xor_u64(two, off); tresc += shfl_xor_u64(tresc, off);
    }
    if (on && r == 0) {
      if (uwo | two) { atomicAdd(&T.stats[2 * (u64)N + li], (u64)uwo); atomicAdd(&T.stats[4 * (u64)N + li], two); }
      if (nresc) { atomicAdd(&T.stats[1 * (u64)N + li], (u64)nresc); atomicAdd(&T.stats[5 * (u64)N + li], tresc); }
    }
  }
}

u32 bft_tile_rows() { return BT_RT; }
u32 bft_block_lists() { return BT_NB; }
hipError_t launch_bf_rowrec(const TaskDev* tasks, const uint2* items, u32 n_items, u32 max_n, hipStream_t st)
{
  hipLaunchKernelGGL(k_bf_rowrec, dim3(n_items), dim3(BT_TPB), (size_t)max_n * 4, st, tasks, items, n_items);
  return hipGetLastError();
}
hipError_t launch_merge_bft(const TaskDev* tasks, const uint2* items, u32 n_items, u32* ticket, u32 grid_x, hipStream_t st)
{
  hipLaunchKernelGGL(k_merge_bft, dim3(grid_x), dim3(BT_TPB), 0, st, tasks, items, n_items, ticket);
  return hipGetLastError();
}

}  // namespace kmx
