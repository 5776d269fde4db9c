// kmx_dev.hpp -- device-side descriptors and small wave/workgroup helpers shared by the
// gfx950 kernels of libkmx.  Wave = 64 lanes (CDNA4); no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kmx {

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;
// explicit global address space for record streams (global_load instead of flat_load)
typedef __attribute__((address_space(1))) const u32 gu32;

// ---- merge task as the kernels see it (one per partition, array in HBM) ----------------------
struct Seg {            // one row segment of a COUNT/PA task: rows [row_off, row_off + nrows) of the arena
  u32 range;            // key range that produced it
  u32 seq;              // tile sequence number inside the range (ascending = ascending keys)
  u64 row_off;
  u32 nrows;
  u32 pad;
};

struct TaskDev {
  const u8* const* recs;   // [N] device pointers to packed records (key words + u32 count)
  const u32* len;          // [N] records per list
  const u32* soft_min;     // [N]
  u32* bounds;             // [(c + 1) * N] first record of range j in list i
  u64* stats;              // [6 * N]; kernels fill rows 1 (RESCUED), 2 (UNIQUE_WO), 4 (TOTAL_WO), 5 (rescued total)
  u8* out;                 // COUNT/PA: row arena; BF/BFC: dense window image
  u64* ctrl;               // [0] rows allocated so far, [1] segments produced, [2] error bits
  Seg* segs;
  u16* rowrec;             // BFT: recurrence of every hash row of the window (k_bf_rowrec), when recurrence-min > 1 or share-min
  u64 out_cap_rows;
  u64 lower, upper;        // BF window [lower, upper]
  u32 seg_cap;
  u32 N;                   // lists
  u32 c;                   // key ranges
  u32 rec_min, share_min;
  u32 mode, bitw;
  u32 row_bytes;           // COUNT/PA: key + payload; BF/BFC: payload
  u32 wl;                  // log2(window slots per list) for the rows kernel
  u32 pivot;               // list whose quantiles split the key space
  u32 rt;                  // BF: rows per tile
  u32 item0;               // first work item of this task in the batch's flat item space
};

// ---- column-blocked merge (merge_cols.hip): per-task companion of TaskDev ------------------------------
struct ColsDev {
  u64* skel;               // [out_cap_rows] ascending row keys (the keys kept by a merge of a few of the task's lists)
  u32* nskel;              // -> number of row keys
  u32* rbounds;            // [c + 1] first row key of each key range
  u64* ovkeys;             // [slots_cap][halves][nblk][16][CL_OVW][2] solid records that are not row keys: key, list << 32 | count
  u32* ovcnt;              // [slots_cap][halves][nblk][16][2] how many of them; 0, or 1 + where the slice goes on in ovx once its CL_OVW entries are full
  u64* ovx;                // [xcap] entries: extensions of CL_XS entries, claimed from xcur by the wave that fills its slice (an outlier sample)
  u32* xcur;               // -> entries of ovx claimed so far
  u32 xcap;
  void* spdir;             // [slots_cap][halves][8] where k_cols_sparse put the rows of the keys outside the row keys
  u32 slots_cap;           // tile slots (tile q of range j: (rbounds[j] + q * rt) / rt + j)
  u32 nblk;                // column blocks
  u32 nb;                  // lists per column block (the last one may hold fewer)
  u32 rt;                  // row keys per tile
  // rows at their final place (file order) out of the two kernels: k_merge_cols leaves the row keys' rows -- payload only, the key
  // is in skel -- in `dense` (row r at r * dpitch), k_cols_sparse writes every slice group's rows, those included, at the
  // group's place in the arena (group offsets by a decoupled look-back over `chain`)
  u8* dense;               // null: rows where the kernels leave them (row keys' rows first, the others behind, + directory)
  u32 dpitch;              // bytes between two rows of `dense` (payload rounded up to 8)
  u32 dense_cap;           // rows `dense` holds
  u8* dnarrow;             // count rows (null: not used): the same rows with a BYTE per count, padded to 8, + a flag byte per column block (1: this block's counts of the row are in `dense`)
  u32 npitch;              // ... bytes between two of them
  u32* gbase;              // [c + 1] slice groups of the task in front of range j (k_cols_prep)
  u64* chain;              // [groups] status << 62 | rows: 1 = the group's own rows, 2 = all rows up to and including it
  uint4* gmap;             // [groups] (range, group in the range, the range's first and last row key) of the task's g-th slice group
  u32* gmax;               // -> max over the batch's tasks of their slice groups (k_cols_sparse's tickets end there)
  u32 ngcap;               // entries of chain / gmap
};

// rows are claimed from a task's arena in chunks of this many bytes (one global atomic + one directory entry per chunk)
#ifndef KMX_CHUNK_BYTES
#define KMX_CHUNK_BYTES 262144
#endif

enum { ERR_ROWS_OVERFLOW = 1, ERR_SEGS_OVERFLOW = 2, ERR_FALLBACK = 4,
       ERR_DIVERGENT = 8,    // (with ERR_FALLBACK, from k_cols_prep: the lists share too few keys for the kernels built for cohorts)
       ERR_SLICES = 16,      // (with ERR_FALLBACK, from k_merge_cols: a wave's set-aside slice ran over -- the batches that follow use the build with extensions)
       ERR_DENSE_CAP = 32 }; // (with ERR_FALLBACK, from k_cols_prep: more row keys than the side store of their rows was sized for -- a host estimate;
                             //  the next batches are sized from what this one reported: no back-off)

// ---- keys -------------------------------------------------------------------------------------
template <int KW> struct Key { u64 w[KW]; };

template <int KW> __device__ __forceinline__ bool key_less(const Key<KW>& a, const Key<KW>& b) {
#pragma unroll
  for (int i = KW - 1; i > 0; i--) { if (a.w[i] != b.w[i]) return a.w[i] < b.w[i]; }
  return a.w[0] < b.w[0];
}
template <int KW> __device__ __forceinline__ bool key_eq(const Key<KW>& a, const Key<KW>& b) {
  bool e = a.w[0] == b.w[0];
#pragma unroll
  for (int i = 1; i < KW; i++) e = e && (a.w[i] == b.w[i]);
  return e;
}
template <int KW> __device__ __forceinline__ bool key_le(const Key<KW>& a, const Key<KW>& b) { return !key_less<KW>(b, a); }
template <int KW> __device__ __forceinline__ Key<KW> key_inf() { Key<KW> k; for (int i = 0; i < KW; i++) k.w[i] = ~0ULL; return k; }
template <int KW> __device__ __forceinline__ Key<KW> key_min(const Key<KW>& a, const Key<KW>& b) { return key_less<KW>(b, a) ? b : a; }

// records are 4-byte aligned only (KW*8 + 4 bytes each): load as dwords
template <int KW> __device__ __forceinline__ Key<KW> load_key(const u8* p) {
  const u32* q = reinterpret_cast<const u32*>(p);
  Key<KW> k;
#pragma unroll
  for (int i = 0; i < KW; i++) k.w[i] = (u64)q[2 * i] | ((u64)q[2 * i + 1] << 32);
  return k;
}

// ---- wave helpers -------------------------------------------------------------------------------
__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m) {
  u32 lo = __shfl_xor((u32)v, m), hi = __shfl_xor((u32)(v >> 32), m);
  return (u64)lo | ((u64)hi << 32);
}
template <int KW> __device__ __forceinline__ Key<KW> wave_min_key(Key<KW> k) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Key<KW> o;
#pragma unroll
    for (int i = 0; i < KW; i++) o.w[i] = shfl_xor_u64(k.w[i], off);
    k = key_min<KW>(k, o);
  }
  return k;
}
// the value of lane (l ^ M), M a constant power of two: DPP moves inside a row of 16 lanes, the gfx950 permlane swaps across rows --
// VALU instructions, where __shfl_xor goes through the LDS crossbar (ds_bpermute_b32: an address register and a slot of the CU's
// one LDS pipe per dword).  scripts/dev/xor_lanes.hip checks them on the device.
template <int M> __device__ __forceinline__ u32 xor_lane_u32(u32 v) {
  static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "a power of two below 64");
  if constexpr (M == 1) return (u32)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);            // quad_perm:[1,0,3,2]
  else if constexpr (M == 2) return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);       // quad_perm:[2,3,0,1]
  else if constexpr (M == 4)                                                                            // row_half_mirror (^ 7), then quad_perm:[3,2,1,0] (^ 3)
    return (u32)__builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true), 0x1B, 0xF, 0xF, true);
  else if constexpr (M == 8) return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);       // row_ror:8
  else if constexpr (M == 16) { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (threadIdx.x & 16u) ? r[0] : r[1]; }
  else { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (threadIdx.x & 32u) ? r[0] : r[1]; }
}
template <int M> __device__ __forceinline__ u64 xor_lane_u64(u64 v) { return (u64)xor_lane_u32<M>((u32)v) | ((u64)xor_lane_u32<M>((u32)(v >> 32)) << 32); }
__device__ __forceinline__ u32 wave_incl_scan(u32 v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { u32 t = __shfl_up(v, off); if (lane >= off) v += t; }
  return v;
}

}  // namespace kmx
