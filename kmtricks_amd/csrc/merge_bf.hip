// merge_bf.hip -- hash-mode merge straight into the dense hash-major Bloom matrix on gfx950.
// Replaces km::HashMerger::next + write_as_bf / write_as_bfc (reference
// include/kmtricks/merge.hpp:441-517, 575-629; set_bit_vector utils.hpp:104-116; pack_v
// packc.hpp:26-43): one ceil(N/8)-byte (BF) or ceil(N*w/8)-byte (BFC) row for EVERY hash of
// [lower, upper], zero where the hash is absent or the row is not kept.
//
// The row index is the hash itself (row = h - lower), so no ranking is needed: a workgroup owns a
// run of row tiles, keeps a tile's bit image (and, when soft-min / recurrence-min / share-min make
// it necessary, a 16-bit recurrence counter per row) in LDS, lets g adjacent lanes stream each
// sample's sorted records of the tile (coalesced 12-byte records), sets bits with ds_or and writes
// the finished tile with coalesced stores.  Input read once (twice from L2 when a recurrence pass
// is needed), output written once.
#include "kmx_dev.hpp"

namespace kmx {

constexpr int BF_TPB = 512;
constexpr int UNR = 8;             // record loads in flight per lane while scanning a list

// bounds[j*N + i] = first record of list i with hash >= lower + j * rows_per_range
__global__ void k_range_bounds_bf(const TaskDev* __restrict__ tasks, u32 max_c)
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
    const u64 W = T.upper - T.lower + 1;
    const u64 tiles = (W + T.rt - 1) / T.rt;
    const u64 tiles_per = (tiles + T.c - 1) / T.c;
    const u64 q = T.lower + (u64)j * tiles_per * T.rt;
    const u8* base = T.recs[i];
    // window hashes are uniform in [lower, upper]: interpolate between the bracket's ends (log log n probes instead of log n
    // dependent loads: 5 against 14 for configs[3]'s lists), every third probe a plain bisection so that no input costs more than 3 log n
    u32 lo = 0, hi = n;
    u64 klo = T.lower, khi = T.upper + 1;      // keys below position lo are < q (>= klo), the key at hi (if any) is >= q (<= khi)
    for (u32 step = 0; lo < hi; step++) {
      u32 mid;
      if (step % 3 == 2 || khi <= klo || q <= klo) mid = lo + ((hi - lo) >> 1);
      else {
        const double f = (double)(q - klo) / (double)(khi - klo);
        mid = lo + (u32)min((double)(hi - lo - 1), f * (double)(hi - lo));
      }
      const u64 k = load_key<1>(base + (u64)mid * 12).w[0];
      if (k < q) { lo = mid + 1; klo = k; } else { hi = mid; khi = k; }
    }
    res = lo;
  }
  T.bounds[(u64)j * T.N + i] = res;
}

__device__ __forceinline__ u32 to_n_b_dev(u32 c, u32 w)
{ // packc.hpp:26-35
  if (!c) return 0;
  const u32 r = 32 - __clz(c), cap = w >= 32 ? 0xFFFFFFFFu : (1u << w) - 1;   // (bitw 32: a 32-bit shift would wrap to 0)
  return r > cap ? cap : r;
}

template <int BFC>
__global__ __launch_bounds__(BF_TPB, 4)
void k_merge_bf(const TaskDev* __restrict__ tasks, const uint2* __restrict__ items, u32 n_items, u32* ticket)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32* bc32 = reinterpret_cast<u32*>(smem);            // first 16 bytes: broadcast slot (all LDS is dynamic)
  unsigned char* const lds = smem + 16;
  const int tid = threadIdx.x;

  for (;;) {
    if (tid == 0) bc32[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    const u32 item = bc32[0];
    __syncthreads();
    if (item >= n_items) return;
    const TaskDev& T = tasks[items[item].x];
    const u32 range = items[item].y;
    const u32 N = T.N, rt = T.rt, nb = T.row_bytes, bitw = T.bitw;
    const u32 rec_min = T.rec_min, share_min = T.share_min;
    const bool two_pass = rec_min > 1 || share_min > 0;
    const u64 W = T.upper - T.lower + 1;
    const u64 tiles = (W + rt - 1) / rt;
    const u64 tiles_per = (tiles + T.c - 1) / T.c;
    const u64 tile0 = (u64)range * tiles_per;
    const u64 tile1 = min(tiles, tile0 + tiles_per);

    const u32 img_bytes = (rt * nb + 15u) & ~15u;
    const u32 rec_bytes = (rt * 2 + 15u) & ~15u;
    u32* img = reinterpret_cast<u32*>(lds);
    u32* rec = reinterpret_cast<u32*>(lds + img_bytes);           // rt u16 counters, two per word
    u32* cur = reinterpret_cast<u32*>(lds + img_bytes + rec_bytes);

    // g adjacent lanes per list
    u32 g = 1; while (g * 2 * N <= (u32)BF_TPB && g < 64) g <<= 1;
    const u32 lpp = BF_TPB / g;                 // lists per pass
    const u32 passes = (N + lpp - 1) / lpp;
    const u32 r = tid & (g - 1);

    for (u32 i = tid; i < N; i += BF_TPB) cur[i] = T.bounds[(u64)range * N + i];
    // one pass covers all lists (N <= lists per pass): a lane serves the same list in every tile, so its
    // record base, range end and soft-min are read once per range instead of once per tile
    const bool single = passes == 1;
    const u8* base1 = nullptr; u32 e1 = 0, sm1 = 0;
    if (single && tid / g < N) { const u32 i1 = tid / g; base1 = T.recs[i1]; e1 = T.bounds[(u64)(range + 1) * N + i1]; sm1 = T.soft_min[i1]; }
    __syncthreads();

    for (u64 tile = tile0; tile < tile1; tile++) {
      const u64 tlo = T.lower + tile * rt;
      const u64 rows = min((u64)rt, T.upper + 1 - tlo);
      const u64 thi = tlo + rows;                       // exclusive
      const u32 bytes = (u32)rows * nb;
      for (u32 t = tid; t < (img_bytes + rec_bytes) / 16; t += BF_TPB)
        reinterpret_cast<uint4*>(lds)[t] = make_uint4(0, 0, 0, 0);
      __syncthreads();

      if (two_pass) {   // recurrence per row: number of solid samples (merge.hpp:458-467)
        for (u32 ps = 0; ps < passes; ps++) {
          const u32 i = ps * lpp + tid / g;
          if (i < N) {
            const u8* base = single ? base1 : T.recs[i];
            const u32 e = single ? e1 : T.bounds[(u64)(range + 1) * N + i], sm = single ? sm1 : T.soft_min[i];
            for (u32 idx = cur[i] + r; idx < e; idx += UNR * g) {   // UNR records in flight per lane
              u64 hh[UNR]; u32 cc[UNR];
#pragma unroll
              for (int q = 0; q < UNR; q++) {
                const u32 ix = idx + q * g;
                hh[q] = ~0ULL; cc[q] = 0;
                if (ix < e) { gu32* p = (gu32*)(uintptr_t)(base + (u64)ix * 12); hh[q] = (u64)p[0] | ((u64)p[1] << 32); cc[q] = p[2]; }
              }
              bool stop = false;
#pragma unroll
              for (int q = 0; q < UNR; q++) {
                if (hh[q] >= thi) { stop = true; break; }
                if (cc[q] >= sm) { const u32 row = (u32)(hh[q] - tlo); atomicAdd(&rec[row >> 1], 1u << ((row & 1u) * 16)); }
              }
              if (stop) break;
            }
          }
        }
        __syncthreads();
      }

      for (u32 ps = 0; ps < passes; ps++) {
        const u32 i = ps * lpp + tid / g;
        u32 uwo = 0; u64 two = 0; u32 next = 0;
        if (i < N) {
          const u8* base = single ? base1 : T.recs[i];
          const u32 e = single ? e1 : T.bounds[(u64)(range + 1) * N + i], sm = single ? sm1 : T.soft_min[i];
          const u32 start = cur[i];
          next = start;
          bool stop = false;
          for (u32 idx0 = start + r; idx0 < e && !stop; idx0 += UNR * g) {   // UNR records in flight per lane
            u64 hh[UNR]; u32 cc[UNR];
#pragma unroll
            for (int q = 0; q < UNR; q++) {
              const u32 ix = idx0 + q * g;
              hh[q] = ~0ULL; cc[q] = 0;
              if (ix < e) { gu32* p = (gu32*)(uintptr_t)(base + (u64)ix * 12); hh[q] = (u64)p[0] | ((u64)p[1] << 32); cc[q] = p[2]; }
            }
#pragma unroll
            for (int q = 0; q < UNR; q++) {
            if (stop) continue;
            const u64 h = hh[q];
            if (h >= thi) { stop = true; continue; }
            const u32 idx = idx0 + q * g;
            next = idx + 1;
            const u32 c = cc[q];
            const u32 row = (u32)(h - tlo);
            const bool solid = c >= sm;
            u32 rc = 0;
            if (two_pass) rc = (rec[row >> 1] >> ((row & 1u) * 16)) & 0xFFFFu;
            u32 outc = 0;
            if (solid) { outc = c; uwo++; two += c; }
            else if (share_min && rc >= share_min) {
              outc = c;
              atomicAdd(&T.stats[1 * (u64)N + i], 1ULL);
              atomicAdd(&T.stats[5 * (u64)N + i], (u64)c);
            }
            const bool keep = two_pass ? (rc >= rec_min) : (solid || rec_min == 0);
            if (keep && outc) {
              if (!BFC) {
                const u32 ob = row * nb + (i >> 3);
                atomicOr(&img[ob >> 2], 1u << (((ob & 3u) << 3) + (i & 7u)));
              } else {
                const u32 v = to_n_b_dev(outc, bitw);
                for (u32 b = 0; b < bitw; b++) {
                  if ((v >> (bitw - 1 - b)) & 1u) {
                    const u32 P = i * bitw + b;                 // bit index from the MSB of byte 0 (bitpacker)
                    const u32 ob = row * nb + (P >> 3);
                    atomicOr(&img[ob >> 2], 1u << (((ob & 3u) << 3) + (7u - (P & 7u))));
                  }
                }
              }
            }
            }
          }
        }
        // the group's new cursor and statistics
        for (u32 off = 1; off < g; off <<= 1) {
          next = max(next, (u32)__shfl_xor(next, (int)off));
          uwo += __shfl_xor(uwo, (int)off); two += shfl_xor_u64(two, (int)off);
        }
        // (the g lanes of a list share a wave: they all read cur[i] before the shuffles above)
        if (i < N && r == 0) {
          cur[i] = next;
          if (uwo | two) { atomicAdd(&T.stats[2 * (u64)N + i], (u64)uwo); atomicAdd(&T.stats[4 * (u64)N + i], two); }
        }
      }
      __syncthreads();
      // stream the tile out: tile starts at a multiple of rt rows (rt % 4 == 0) -> 4-byte aligned
      u8* dst = T.out + (tlo - T.lower) * nb;
      {
        // 16-byte stores where the destination allows them (round 5: dword stores before): the dwords in front of the first 16-byte
        // boundary one by one, then four image dwords a lane and store, then the tail
        u32* d32 = reinterpret_cast<u32*>(dst);
        const u32 nw = bytes >> 2;
        const u32 head = min(nw, (u32)(((16u - (u32)((uintptr_t)dst & 15u)) & 15u) >> 2)), nq = (nw - head) >> 2;
        if (tid < head) d32[tid] = img[tid];
        for (u32 t = tid; t < nq; t += BF_TPB) {
          const u32 w = head + 4 * t;
          *reinterpret_cast<uint4*>(d32 + w) = make_uint4(img[w], img[w + 1], img[w + 2], img[w + 3]);
        }
        for (u32 t = head + 4 * nq + tid; t < nw; t += BF_TPB) d32[t] = img[t];
        for (u32 t = (nw << 2) + tid; t < bytes; t += BF_TPB) dst[t] = lds[t];
      }
      __syncthreads();
    }
  }
}

template __global__ void k_merge_bf<0>(const TaskDev*, const uint2*, u32, u32*);
template __global__ void k_merge_bf<1>(const TaskDev*, const uint2*, u32, u32*);

}  // namespace kmx

namespace kmx {

int bf_lds_bytes(u32 rt, u32 nb, u32 n_lists)
{ return 16 + (int)((rt * nb + 15u) & ~15u) + (int)((rt * 2 + 15u) & ~15u) + 4 * (int)n_lists; }

hipError_t launch_range_bounds_bf(const TaskDev* tasks, u32 n_tasks, u32 max_n, u32 max_c, hipStream_t st)
{
  dim3 grid((max_n + 255) / 256, max_c + 1, n_tasks), block(256);
  hipLaunchKernelGGL(k_range_bounds_bf, grid, block, 0, st, tasks, max_c);
  return hipGetLastError();
}

hipError_t launch_merge_bf(int bfc, const TaskDev* tasks, const uint2* items, u32 n_items, u32* ticket,
                           u32 grid_x, int lds, hipStream_t st)
{
  dim3 grid(grid_x), block(BF_TPB);
  hipError_t e;
  if (!bfc) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_merge_bf<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_merge_bf<0>, grid, block, lds, st, tasks, items, n_items, ticket);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_merge_bf<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_merge_bf<1>, grid, block, lds, st, tasks, items, n_items, ticket);
  }
  return hipGetLastError();
}

}  // namespace kmx
