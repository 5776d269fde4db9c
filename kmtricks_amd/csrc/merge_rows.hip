// merge_rows.hip -- N-way merge of sorted per-sample count lists into count / presence-absence
// matrix rows on gfx950.  Replaces km::KmerMerger::next + write_as_bin/write_as_pa and
// km::HashMerger::next + write_as_bin/write_as_pa (reference include/kmtricks/merge.hpp:183-286,
// 441-558): for every distinct key, counts[i] = c_i if c_i >= soft_min[i]; non-solid entries are
// rescued when share_min > 0 and recurrence >= share_min; the row is kept iff recurrence >= rec_min.
//
// Decomposition (see DESIGN.md "merge kernel"):
//   * the key space of a partition is cut into c ranges at quantiles of one pivot list
//     (k_range_bounds: one lower_bound per (range, list));
//   * a workgroup owns one range and walks it tile by tile.  A tile gives every list a window of
//     w = 2^wl record slots (w adjacent lanes read w consecutive 12/20-byte records); the tile's
//     key bound b is the smallest "last key of a window that has more records behind it", so every
//     record <= b of every list is inside its window: tiles are disjoint, ascending key intervals
//     that always fit the 4096 register slots, whatever the skew;
//   * inside a tile the distinct keys are found with an LDS hash set (owner index + recurrence
//     packed in one u32, claimed with ds_cmpst), kept keys are ranked, and rows are assembled as
//     a byte-exact file image in LDS, then streamed out with coalesced stores;
//   * a tile's rows go to a segment of the task's row arena claimed with ONE global atomic; the
//     (range, seq) directory restores ascending key order when the body is copied out.  Input is
//     read once, output written once, no inter-workgroup dependency.
#include "kmx_dev.hpp"

namespace kmx {

constexpr int TPB = 512;           // 8 waves; two workgroups per CU at <= 80 KiB LDS
constexpr int M = 8;               // record slots per thread
constexpr int CAP = TPB * M;       // 4096 record slots per tile
constexpr int TS = 2 * CAP;        // hash set entries (load factor <= 0.5)
constexpr int KLBYTES = 10240;     // kept-key list: fast path keys + table slots, or u16 sort array
constexpr int NWAVE = TPB / 64;

__host__ __device__ inline int rows_emit_bytes(int kw) { return CAP * kw * 8 + TS * 4 + KLBYTES; }
__host__ __device__ inline int rows_dkmax(int kw) { return 8192 / (kw * 8); }   // + 2 B table slot each

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
    while (lo < hi) {
      u32 mid = lo + ((hi - lo) >> 1);
      Key<KW> k = load_key<KW>(base + (u64)mid * RB);
      if (key_less<KW>(k, q)) lo = mid + 1; else hi = mid;
    }
    res = lo;
  }
  T.bounds[(u64)j * T.N + i] = res;
}

// ---- the merge kernel ----------------------------------------------------------------------------
template <int KW> __device__ __forceinline__ u32 key_hash(const Key<KW>& k)
{
  u64 x = k.w[0];
  if (KW == 2) x ^= k.w[KW - 1] * 0x9E3779B97F4A7C15ULL;
  x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
  return (u32)x & (TS - 1);
}

// KW = 1: 80 KiB LDS -> two workgroups per CU (4 waves/SIMD, <= 128 VGPRs);
// KW = 2: 112 KiB LDS -> one workgroup per CU, so it may use 256 VGPRs.
template <int KW, int MODE>
__global__ __launch_bounds__(TPB, (KW == 1 ? 4 : 2))
void k_merge_rows(const TaskDev* __restrict__ tasks, const uint2* __restrict__ items, u32 n_items, u32* ticket)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int RB = KW * 8 + 4;
  constexpr int KEYS_BYTES = CAP * KW * 8;
  constexpr int EMIT_BYTES = KEYS_BYTES + TS * 4 + KLBYTES;
  constexpr int DKMAX = 8192 / (KW * 8);

  Key<KW>* keysL = reinterpret_cast<Key<KW>*>(smem);
  u32* tab = reinterpret_cast<u32*>(smem + KEYS_BYTES);
  unsigned char* klist = smem + KEYS_BYTES + TS * 4;
  Key<KW>* dkeys = reinterpret_cast<Key<KW>*>(klist);
  u16* dslot = reinterpret_cast<u16*>(klist + 8192);
  u16* sortv = reinterpret_cast<u16*>(klist);
  // misc (after the emission image): per-wave partials, then the cursors
  unsigned char* misc = smem + EMIT_BYTES;
  Key<KW>* wmin = reinterpret_cast<Key<KW>*>(misc);                 // NWAVE keys (<= 128 B)
  u32* wsum = reinterpret_cast<u32*>(misc + 128);                    // NWAVE + 1
  u64* bc64 = reinterpret_cast<u64*>(misc + 192);                    // broadcast slot
  u32* bc32 = reinterpret_cast<u32*>(misc + 208);                    // broadcast: item / flags
  u32* cur = reinterpret_cast<u32*>(misc + 256);                     // N cursors

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  for (;;) {
    // ---- next work item (dynamic, ascending ids) ----
    if (tid == 0) bc32[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    const u32 item = bc32[0];
    __syncthreads();
    if (item >= n_items) return;
    const TaskDev& T = tasks[items[item].x];
    const u32 range = items[item].y;
    const u32 N = T.N, wl = T.wl, w = 1u << wl;
    const u32 rec_min = T.rec_min, share_min = T.share_min, row_bytes = T.row_bytes;

    // fixed slot -> (list, position in window) mapping of this thread
    const u8* ptr[M]; u32 endv[M], smin[M], uwo[M]; u64 two[M];
    u32 li[M];
#pragma unroll
    for (int m = 0; m < M; m++) {
      const u32 s = tid + m * TPB;
      li[m] = s >> wl;
      uwo[m] = 0; two[m] = 0;
      if (li[m] < N) {
        ptr[m] = T.recs[li[m]];
        endv[m] = T.bounds[(u64)(range + 1) * N + li[m]];
        smin[m] = T.soft_min[li[m]];
      } else { ptr[m] = nullptr; endv[m] = 0; smin[m] = 0; }
    }
    for (u32 i = tid; i < N; i += TPB) cur[i] = T.bounds[(u64)range * N + i];
    __syncthreads();

    u32 seq = 0;
    for (;;) {
      // ---- 1. load the windows ----
      Key<KW> key[M]; u32 cnt[M]; u32 idx[M];
      u32 validm = 0;
      Key<KW> cand = key_inf<KW>();
#pragma unroll
      for (int m = 0; m < M; m++) {
        const u32 s = tid + m * TPB, rr = s & (w - 1);
        key[m] = key_inf<KW>(); cnt[m] = 0; idx[m] = 0;
        if (li[m] < N) {
          idx[m] = cur[li[m]] + rr;
          if (idx[m] < endv[m]) {
            const u8* p = ptr[m] + (u64)idx[m] * RB;
            key[m] = load_key<KW>(p);
            cnt[m] = reinterpret_cast<const u32*>(p)[2 * KW];
            validm |= 1u << m;
            if (rr == w - 1 && idx[m] + 1 < endv[m]) cand = key_min<KW>(cand, key[m]);
          }
        }
      }
      cand = wave_min_key<KW>(cand);
      if (lane == 0) wmin[wave] = cand;
      const int any = __syncthreads_or(validm != 0);
      if (!any) break;
      Key<KW> b = wmin[0];
#pragma unroll
      for (int v = 1; v < NWAVE; v++) b = key_min<KW>(b, wmin[v]);

      // ---- 2. consume keys <= b, advance cursors, stage keys, clear the hash set ----
      u32 consm = 0;
#pragma unroll
      for (int m = 0; m < M; m++) {
        const u32 s = tid + m * TPB, rr = s & (w - 1);
        const int c = ((validm >> m) & 1u) && key_le<KW>(key[m], b);
        const int nxt = __shfl_down(c, 1);
        if (c) {
          consm |= 1u << m;
          keysL[s] = key[m];
          if (rr == w - 1 || lane == 63 || !nxt) atomicMax(&cur[li[m]], idx[m] + 1);
        }
      }
      {
        uint4* t4 = reinterpret_cast<uint4*>(tab);
        for (int t = tid; t < TS / 4; t += TPB) t4[t] = make_uint4(0, 0, 0, 0);
      }
      __syncthreads();

      // ---- 3. hash-set insert: entry = owner slot + 1 (low 16) | recurrence (high 16) ----
      u32 hs[M]; u32 ownm = 0, solidm = 0;
#pragma unroll
      for (int m = 0; m < M; m++) {
        hs[m] = 0;
        if ((consm >> m) & 1u) {
          const u32 s = tid + m * TPB;
          u32 h = key_hash<KW>(key[m]);
          for (;;) {
            const u32 old = atomicCAS(&tab[h], 0u, s + 1);
            if (old == 0) { ownm |= 1u << m; break; }
            const Key<KW> ok = keysL[(old & 0xFFFFu) - 1];
            if (key_eq<KW>(ok, key[m])) break;
            h = (h + 1) & (TS - 1);
          }
          hs[m] = h;
          if (cnt[m] >= smin[m]) { solidm |= 1u << m; atomicAdd(&tab[h], 1u << 16); }
        }
      }
      __syncthreads();

      // ---- 4. kept distinct keys: compact, rank ----
      u32 dk;
      {
        u32 e[16]; u32 nk = 0;
        const uint4* t4 = reinterpret_cast<const uint4*>(tab) + tid * 4;
#pragma unroll
        for (int q = 0; q < 4; q++) { uint4 v = t4[q]; e[4 * q] = v.x; e[4 * q + 1] = v.y; e[4 * q + 2] = v.z; e[4 * q + 3] = v.w; }
#pragma unroll
        for (int q = 0; q < 16; q++) nk += (e[q] != 0 && (e[q] >> 16) >= rec_min) ? 1u : 0u;
        const u32 incl = wave_incl_scan(nk, lane);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        u32 base = 0, total = 0;
#pragma unroll
        for (int v = 0; v < NWAVE; v++) { const u32 x = wsum[v]; if (v < wave) base += x; total += x; }
        dk = total;
        u32 pos = base + incl - nk;
        const bool fast = dk <= (u32)DKMAX;
#pragma unroll
        for (int q = 0; q < 16; q++) {
          if (e[q] == 0) continue;
          const u32 t = tid * 16 + q;
          if ((e[q] >> 16) >= rec_min) {
            if (fast) { dkeys[pos] = keysL[(e[q] & 0xFFFFu) - 1]; dslot[pos] = (u16)t; }
            else sortv[pos] = (u16)t;
            pos++;
          } else tab[t] = (e[q] & 0xFFFF0000u) | 0xFFFFu;   // not kept
        }
        __syncthreads();
        if (fast) {
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
          for (u32 p = dk + tid; p < p2; p += TPB) sortv[p] = 0xFFFFu;   // +inf padding
          __syncthreads();
          for (u32 k2 = 2; k2 <= p2; k2 <<= 1) {
            for (u32 j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
              for (u32 p = tid; p < p2; p += TPB) {
                const u32 q = p ^ j2;
                if (q > p) {
                  const u32 a = sortv[p], bb = sortv[q];
                  bool a_gt_b;
                  if (a == 0xFFFFu) a_gt_b = (bb != 0xFFFFu);
                  else if (bb == 0xFFFFu) a_gt_b = false;
                  else a_gt_b = key_less<KW>(keysL[(tab[bb] & 0xFFFFu) - 1], keysL[(tab[a] & 0xFFFFu) - 1]);
                  const bool up = (p & k2) == 0;
                  if (a_gt_b == up) { sortv[p] = (u16)bb; sortv[q] = (u16)a; }
                }
              }
              __syncthreads();
            }
          }
          for (u32 p = tid; p < dk; p += TPB) { const u32 t = sortv[p]; tab[t] = (tab[t] & 0xFFFF0000u) | p; }
        }
      }
      // ---- 5. claim the row segment ----
      if (tid == 0) {
        u64 off = 0; u32 ok = 1;
        if (dk) {
          off = atomicAdd(&T.ctrl[0], (u64)dk);
          if (off + dk > T.out_cap_rows) { ok = 0; atomicOr(&T.ctrl[2], (u64)ERR_ROWS_OVERFLOW); }
          const u64 sidx = atomicAdd(&T.ctrl[1], 1ULL);
          if (sidx < T.seg_cap) { Seg sg; sg.range = range; sg.seq = seq; sg.row_off = off; sg.nrows = dk; sg.pad = 0; T.segs[sidx] = sg; }
          else atomicOr(&T.ctrl[2], (u64)ERR_SEGS_OVERFLOW);
        }
        bc64[0] = off; bc32[1] = ok;
      }
      __syncthreads();
      const u64 row_off = bc64[0];
      const bool can_write = bc32[1] != 0;

      // ---- 6. per-record decision (merge.hpp:199-247), statistics ----
      u32 kr[M], outc[M];
#pragma unroll
      for (int m = 0; m < M; m++) {
        kr[m] = 0xFFFFu; outc[m] = 0;
        if ((consm >> m) & 1u) {
          const u32 e = tab[hs[m]];
          const u32 rec = e >> 16;
          kr[m] = e & 0xFFFFu;
          if ((solidm >> m) & 1u) { outc[m] = cnt[m]; uwo[m] += 1; two[m] += cnt[m]; }
          else if (share_min && rec >= share_min) {
            outc[m] = cnt[m];
            atomicAdd(&T.stats[1 * (u64)N + li[m]], 1ULL);
            atomicAdd(&T.stats[5 * (u64)N + li[m]], (u64)cnt[m]);
          }
        }
      }
      __syncthreads();   // table fully read: the LDS image may now alias it

      // ---- 7. assemble rows as a file-body image in LDS, stream them out ----
      if (dk && can_write) {
        const u32 rows_per = (u32)EMIT_BYTES / row_bytes;
        u8* const dst0 = T.out + row_off * row_bytes;
        for (u32 b0 = 0; b0 < dk; b0 += rows_per) {
          const u32 nb = min(rows_per, dk - b0);
          const u32 bytes = nb * row_bytes;
          {
            uint4* z = reinterpret_cast<uint4*>(smem);
            for (u32 t = tid; t < (bytes + 15) / 16; t += TPB) z[t] = make_uint4(0, 0, 0, 0);
          }
          __syncthreads();
#pragma unroll
          for (int m = 0; m < M; m++) {
            const u32 r = kr[m] - b0;   // wraps for 0xFFFF / other batches
            if (((consm >> m) & 1u) && kr[m] != 0xFFFFu && r < nb) {
              u8* row = smem + r * row_bytes;
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
              if (outc[m]) {
                if (MODE == 0) reinterpret_cast<u32*>(row + KW * 8)[li[m]] = outc[m];
                else {
                  const u32 ob = r * row_bytes + KW * 8 + (li[m] >> 3);
                  atomicOr(reinterpret_cast<u32*>(smem) + (ob >> 2), 1u << (((ob & 3u) << 3) + (li[m] & 7u)));
                }
              }
            }
          }
          __syncthreads();
          u8* dst = dst0 + (u64)b0 * row_bytes;
          if (MODE == 0) {
            const u32* src = reinterpret_cast<const u32*>(smem);
            u32* d32 = reinterpret_cast<u32*>(dst);
            for (u32 t = tid; t < bytes / 4; t += TPB) d32[t] = src[t];
          } else {
            for (u32 t = tid; t < bytes; t += TPB) dst[t] = smem[t];
          }
          __syncthreads();
        }
      }
      seq++;
    }

    // ---- range done: flush per-list statistics (UNIQUE_WO, TOTAL_WO) ----
#pragma unroll
    for (int m = 0; m < M; m++) {
      u32 a = uwo[m]; u64 t2 = two[m];
      // reduce over the w adjacent lanes of a list (w <= 64 here; wider windows add per lane)
      for (u32 off = 1; off < w && off < 64; off <<= 1) {
        a += __shfl_xor(a, (int)off); t2 += shfl_xor_u64(t2, (int)off);
      }
      const u32 s = tid + m * TPB;
      const bool leader = (w >= 64) ? (lane == 0) : ((s & (w - 1)) == 0);
      if (li[m] < N && leader && (a | t2)) {
        atomicAdd(&T.stats[2 * (u64)N + li[m]], (u64)a);
        atomicAdd(&T.stats[4 * (u64)N + li[m]], t2);
      }
    }
    __syncthreads();
  }
}

// explicit instantiations used by the host side
template __global__ void k_range_bounds<1>(const TaskDev*, u32);
template __global__ void k_range_bounds<2>(const TaskDev*, u32);
template __global__ void k_merge_rows<1, 0>(const TaskDev*, const uint2*, u32, u32*);
template __global__ void k_merge_rows<1, 1>(const TaskDev*, const uint2*, u32, u32*);
template __global__ void k_merge_rows<2, 0>(const TaskDev*, const uint2*, u32, u32*);
template __global__ void k_merge_rows<2, 1>(const TaskDev*, const uint2*, u32, u32*);

}  // namespace kmx

// ---- host-side launchers (plain functions so other translation units need no device code) -----
namespace kmx {

int rows_lds_bytes(int kw, u32 n_lists) { return rows_emit_bytes(kw) + 256 + 4 * (int)n_lists; }
int rows_cap() { return CAP; }

hipError_t launch_range_bounds(int kw, const TaskDev* tasks, u32 n_tasks, u32 max_n, u32 max_c, hipStream_t st)
{
  dim3 grid((max_n + 255) / 256, max_c + 1, n_tasks), block(256);
  if (kw == 1) hipLaunchKernelGGL(k_range_bounds<1>, grid, block, 0, st, tasks, max_c);
  else hipLaunchKernelGGL(k_range_bounds<2>, grid, block, 0, st, tasks, max_c);
  return hipGetLastError();
}

hipError_t launch_merge_rows(int kw, int mode, const TaskDev* tasks, const uint2* items, u32 n_items, u32* ticket,
                             u32 grid_x, u32 max_n, hipStream_t st)
{
  const int lds = rows_lds_bytes(kw, max_n);
  dim3 grid(grid_x), block(TPB);
#define KMX_LAUNCH(KW_, MODE_)                                                                              \
  do {                                                                                                      \
    hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_merge_rows<KW_, MODE_>),           \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);                   \
    if (e_ != hipSuccess) return e_;                                                                        \
    hipLaunchKernelGGL((k_merge_rows<KW_, MODE_>), grid, block, lds, st, tasks, items, n_items, ticket);    \
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
