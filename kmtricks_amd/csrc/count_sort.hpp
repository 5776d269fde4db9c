// count_sort.hpp -- partition-local sort + run-length count of decoded k-mers (or window hashes) on gfx950: the device side of
// km::KmerSort / HashSort + KmerPartCounter / HashPartCounter::executeDump (reference include/kmtricks/gatb/sorting_count.hpp:
// 488-533 sort, 694-884 and 971-990 run-length dump; count_processor.hpp:61-70, 135-146 hard-min).  Included by count.hip.
//
// The decode kernel leaves the keys grouped by partition.  A partition (tens to hundreds of thousands of keys) does not fit the
// LDS, a bucket of it does -- so it is a SAMPLE SORT per partition, every stage staged in LDS, no key ever compared in HBM:
//   k_cs_splitters  a workgroup per partition sorts an even sample of its keys in LDS (bitonic) and keeps every (sample / buckets)-th
//                   as a splitter: buckets of ~1000 keys whatever the key distribution (canonical k-mers of a minimizer partition
//                   crowd a few prefixes -- a fixed radix digit would not balance);
//   k_cs_count      a workgroup per chunk of 4096 keys: bucket of every key by binary search in the partition's splitters (LDS),
//                   LDS histogram, one global add per bucket and chunk;
//   k_cs_scan       exclusive scan of the bucket sizes (one workgroup);
//   k_cs_scatter    the same walk again: a key goes to its bucket's place (rank inside the chunk from the LDS histogram, the
//                   chunk's base from one global add per bucket);
//   k_cs_sort       a workgroup per bucket: keys into LDS, bitonic sort, run starts by neighbour compare, run lengths = counts,
//                   runs of at least hard-min kept (counts saturate at u32); the kept (key, count) pairs go to the bucket's own
//                   place in a temporary, their number to a table;
//   k_cs_scan + k_cs_compact  the kept pairs of all buckets, packed: partition p's result is one ascending run.
// Equal keys always share a bucket (the splitter compare decides).  A bucket over the LDS capacity (a k-mer repeated thousands of
// times, an unlucky sample) makes the call fall back to the library sort (rocPRIM radix sort + run-length encode) -- same result.
// Traffic per k-mer: 8 B written by the decode, read 3 times and written once here (K + 4 per DISTINCT k-mer out): ~40 B against
// the >= 128 B of an 8-pass LSD radix sort over the whole batch.
#pragma once
#include "kmx_dev.hpp"
#include "skf.hpp"

namespace kmx {

constexpr int CS_TPB = 256;
constexpr int CS_WALK_TPB = 1024;
constexpr int CS_CHUNK = 16 * CS_WALK_TPB;  // (one- and two-word keys; cs_chunk<K>() below)  // keys per workgroup in the count / scatter walks (a chunk holds ~11 keys per bucket of a partition cut into 1500: the pieces the scatter writes)
constexpr int CS_MAXB = 2048;             // buckets per partition (partitions of up to ~1 M keys; beyond: the library sort)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "count_sort.hpp is written for gfx950 (MI355X): 160 KB of LDS per workgroup (k_cs_splitters: 128 KB), v_permlane16_swap / v_permlane32_swap"
#endif
// keys of three and four words (k = 65 ... 127: Kmer<96> / Kmer<128>), low word first, compared most significant word first -- round 5:
// they take the same kernels as the one- and two-word keys (before: the library's radix sort word by word, kw * 8 + 1 passes over the batch)
template <int KW> struct WideKey { u64 w[KW]; };
template <int KW> __host__ __device__ __forceinline__ bool operator<(const WideKey<KW>& a, const WideKey<KW>& b)
{
  bool lt = false, eq = true;
#pragma unroll
  for (int i = KW - 1; i >= 0; i--) { lt = lt || (eq && a.w[i] < b.w[i]); eq = eq && a.w[i] == b.w[i]; }
  return lt;
}
template <int KW> __host__ __device__ __forceinline__ bool operator==(const WideKey<KW>& a, const WideKey<KW>& b)
{
  bool eq = true;
#pragma unroll
  for (int i = 0; i < KW; i++) eq = eq && a.w[i] == b.w[i];
  return eq;
}
template <int KW> __host__ __device__ __forceinline__ bool operator!=(const WideKey<KW>& a, const WideKey<KW>& b) { return !(a == b); }
template <int KW> __host__ __device__ __forceinline__ bool operator>(const WideKey<KW>& a, const WideKey<KW>& b) { return b < a; }
template <int KW> __host__ __device__ __forceinline__ bool operator<=(const WideKey<KW>& a, const WideKey<KW>& b) { return !(b < a); }
// dword w of a key (its record's layout: low dword first)
template <typename K> __device__ __forceinline__ u32 key_dword(const K& k, u32 w) { return (u32)(k >> (32 * w)); }
template <> __device__ __forceinline__ u32 key_dword<WideKey<3>>(const WideKey<3>& k, u32 w) { return (u32)(k.w[w >> 1] >> (32 * (w & 1))); }
template <> __device__ __forceinline__ u32 key_dword<WideKey<4>>(const WideKey<4>& k, u32 w) { return (u32)(k.w[w >> 1] >> (32 * (w & 1))); }

// what the splitters of a key type are: the key itself, or -- wide keys -- its two most significant words (the first 64 bases: a
// partition's k-mers hardly ever share them, equal keys share them by definition, and 8192 samples of 16 bytes fit the LDS)
template <typename K> struct CsSpl { typedef K type; static __device__ __forceinline__ K top(const K& k) { return k; } };
template <int KW> struct CsSpl<WideKey<KW>> {
  typedef __uint128_t type;
  static __device__ __forceinline__ __uint128_t top(const WideKey<KW>& k) { return ((__uint128_t)k.w[KW - 1] << 64) | k.w[KW - 2]; }
};

// walk: keys a thread of the count / scatter walks holds (1024 threads: 128 registers each)
template <typename K> struct CsCap { static constexpr int cap = 4096, sample = 16384, walk = 16; };            // keys of a bucket that fit the sort's LDS; keys sampled per partition (16384 x 8 B or 8192 x 16 B = 128 KB of LDS in k_cs_splitters)
template <> struct CsCap<__uint128_t> { static constexpr int cap = 4096, sample = 8192, walk = 16; };
template <int KW> struct CsCap<WideKey<KW>> { static constexpr int cap = 2048, sample = 8192, walk = 8; };      // (cap: 48 / 64 + 8 KB in k_cs_sort; sample: of CsSpl's 16-byte tops)      // (cap: 64 + 16 KB of LDS in k_cs_sort, which only sees the buckets the wave kernel leaves)
constexpr u32 CS_WAVE_MAX = 1024;         // keys of a bucket that one wave sorts in registers (16 per lane)
// aimed bucket size.  Round 6: 416 with 8-16 samples a bucket (before: 512 with 4-8) -- a bucket's size then scatters by a third of
// its aim: three in four fit the wave kernel's small network (512 keys: 0.7 compare-exchanges a key and stage against 0.86 in the
// large one), and hardly one is beyond its 1024 (before: one in forty, 70 us of LDS kernel a sample)
template <typename K> __host__ __device__ inline u32 cs_target() { return 416u; }

template <typename K> __host__ __device__ constexpr u32 cs_chunk() { return (u32)CsCap<K>::walk * (u32)CS_WALK_TPB; }
struct CsPart { u32 key0, nkeys, bucket0, nb; };      // a partition's keys [key0, key0 + nkeys), its buckets [bucket0, bucket0 + nb)
struct CsChunk { u32 part, key0, nkeys, pad; };

template <typename K> __device__ __forceinline__ K cs_max() { return ~(K)0; }
template <> __device__ __forceinline__ WideKey<3> cs_max<WideKey<3>>() { return WideKey<3>{{~0ULL, ~0ULL, ~0ULL}}; }
template <> __device__ __forceinline__ WideKey<4> cs_max<WideKey<4>>() { return WideKey<4>{{~0ULL, ~0ULL, ~0ULL, ~0ULL}}; }

// ---- bitonic sort of P keys in LDS (P a power of two, pads = cs_max).  Steps between keys less than 128 apart stay inside a
//      wave: it holds a chunk of 128 keys in registers, two per lane, and exchanges them with lane shuffles -- no workgroup
//      barrier, half the LDS traffic.  4096 keys: 15 barriers instead of 78. ----
template <typename K> __device__ __forceinline__ K cs_shfl_xor(K k, int m);
template <> __device__ __forceinline__ u64 cs_shfl_xor<u64>(u64 k, int m) { return (u64)__shfl_xor((unsigned long long)k, m); }
template <> __device__ __forceinline__ __uint128_t cs_shfl_xor<__uint128_t>(__uint128_t k, int m)
{
  const u64 lo = (u64)__shfl_xor((unsigned long long)(u64)k, m), hi = (u64)__shfl_xor((unsigned long long)(u64)(k >> 64), m);
  return ((__uint128_t)hi << 64) | lo;
}
template <int KW> __device__ __forceinline__ WideKey<KW> cs_shfl_xor_wide(const WideKey<KW>& k, int m)
{
  WideKey<KW> r;
#pragma unroll
  for (int i = 0; i < KW; i++) r.w[i] = (u64)__shfl_xor((unsigned long long)k.w[i], m);
  return r;
}
template <> __device__ __forceinline__ WideKey<3> cs_shfl_xor<WideKey<3>>(WideKey<3> k, int m) { return cs_shfl_xor_wide<3>(k, m); }
template <> __device__ __forceinline__ WideKey<4> cs_shfl_xor<WideKey<4>>(WideKey<4> k, int m) { return cs_shfl_xor_wide<4>(k, m); }
// the key of lane ^ M, M a constant (kmx_dev.hpp: DPP moves and permlane swaps instead of ds_bpermute)
template <typename K, int M> __device__ __forceinline__ K cs_xor(K k);
template <typename K, int M> struct CsXor;
template <int M> struct CsXor<u64, M> { static __device__ __forceinline__ u64 get(u64 k) { return xor_lane_u64<M>(k); } };
template <int M> struct CsXor<__uint128_t, M> {
  static __device__ __forceinline__ __uint128_t get(__uint128_t k) { return ((__uint128_t)xor_lane_u64<M>((u64)(k >> 64)) << 64) | xor_lane_u64<M>((u64)k); }
};
template <int KW, int M> struct CsXor<WideKey<KW>, M> {
  static __device__ __forceinline__ WideKey<KW> get(const WideKey<KW>& k)
  {
    WideKey<KW> r;
#pragma unroll
    for (int i = 0; i < KW; i++) r.w[i] = xor_lane_u64<M>(k.w[i]);
    return r;
  }
};
template <typename K, int J>
__device__ __forceinline__ void cs_chunk_step(K (&k)[2], u32 base, u32 lane, u32 k2)      // steps J .. 1
{
#pragma unroll
  for (int x = 0; x < 2; x++) {
    const K o = CsXor<K, J>::get(k[x]);
    const bool asc = ((base + 64u * x + lane) & k2) == 0;
    const bool keep_min = ((lane & (u32)J) == 0) == asc;
    if ((o < k[x]) == keep_min) k[x] = o;      // (an equal partner may be taken: nothing changes)
  }
  if constexpr (J > 1) cs_chunk_step<K, J / 2>(k, base, lane, k2);
}
template <typename K>
__device__ __forceinline__ void cs_chunk_steps(K (&k)[2], u32 base, u32 lane, u32 k2, u32 jtop)      // steps j = jtop .. 1 of stage k2 (jtop <= 64)
{
  if (jtop >= 64u) {
    const bool asc = ((base + lane) & k2) == 0;      // (k2 >= 128: both of my keys sort the same way)
    if ((k[1] < k[0]) == asc) { const K t = k[0]; k[0] = k[1]; k[1] = t; }
    jtop = 32;
  }
  switch (jtop) {      // (uniform over the wave)
    case 32: cs_chunk_step<K, 32>(k, base, lane, k2); break;
    case 16: cs_chunk_step<K, 16>(k, base, lane, k2); break;
    case 8: cs_chunk_step<K, 8>(k, base, lane, k2); break;
    case 4: cs_chunk_step<K, 4>(k, base, lane, k2); break;
    case 2: cs_chunk_step<K, 2>(k, base, lane, k2); break;
    default: cs_chunk_step<K, 1>(k, base, lane, k2); break;
  }
}
template <typename K, int TPB = CS_TPB>
__device__ __forceinline__ void cs_sort_lds(K* s, u32 P, u32 tid)
{
  if (P < 128u) {
    for (u32 k2 = 2; k2 <= P; k2 <<= 1)
      for (u32 j = k2 >> 1; j > 0; j >>= 1) {
        for (u32 t = tid; t < P / 2; t += TPB) {
          const u32 a = ((t & ~(j - 1)) << 1) | (t & (j - 1)), c = a | j;
          const K x = s[a], y = s[c];
          if ((x > y) == ((a & k2) == 0)) { s[a] = y; s[c] = x; }
        }
        __syncthreads();
      }
    return;
  }
  const u32 lane = tid & 63u, wave = tid >> 6;
  for (u32 base = wave * 128u; base < P; base += (TPB / 64) * 128u) {
    K k[2] = {s[base + lane], s[base + 64u + lane]};
    for (u32 k2 = 2; k2 <= 128u; k2 <<= 1) cs_chunk_steps<K>(k, base, lane, k2, k2 >> 1);
    s[base + lane] = k[0]; s[base + 64u + lane] = k[1];
  }
  __syncthreads();
  for (u32 k2 = 256; k2 <= P; k2 <<= 1) {
    for (u32 j = k2 >> 1; j >= 128u; j >>= 1) {
      for (u32 t = tid; t < P / 2; t += TPB) {
        const u32 a = ((t & ~(j - 1)) << 1) | (t & (j - 1)), c = a | j;
        const K x = s[a], y = s[c];
        if ((x > y) == ((a & k2) == 0)) { s[a] = y; s[c] = x; }
      }
      __syncthreads();
    }
    for (u32 base = wave * 128u; base < P; base += (TPB / 64) * 128u) {
      K k[2] = {s[base + lane], s[base + 64u + lane]};
      cs_chunk_steps<K>(k, base, lane, k2, 64u);
      s[base + lane] = k[0]; s[base + 64u + lane] = k[1];
    }
    __syncthreads();
  }
}

constexpr int CS_SPL_TPB = 1024;          // (a workgroup per partition: few of them for few large partitions -- the sort of the samples is the kernel's time)
// ---- round 6: a look-up table in front of the bucket search.  A walk finds a key's bucket with a binary search over the partition's
//      splitters: 8 dependent 8-byte LDS reads a key, half of a walk's time.  The table cuts a partition's key range -- seen through
//      t(key) = 32 bits of the key from bit `tshift` on (its top 32 significant bits) -- into CS_LUT cells of equal width between the
//      first and the last splitter and holds, per cell border, how many splitters lie below it: a key's cell leaves the exact search
//      (the same compares on the same splitters) the splitters INSIDE the cell -- none or one as a rule, all of them only when every
//      splitter shares its 32 bits.  k_cs_splitters builds it, 2 KB a partition. ----
constexpr u32 CS_LUT = 1024;
struct CsLut { u32 tmin, lsh; u16 lb[CS_LUT + 2]; };      // lb[c]: splitters whose t is below cell c's first value; lb[CS_LUT] = all of them
template <typename K> __device__ __forceinline__ u32 cs_t32(const K& k, u32 tshift) { return (u32)(k >> tshift); }
template <typename K>
__global__ __launch_bounds__(CS_SPL_TPB)
void k_cs_splitters(const K* __restrict__ keys, const CsPart* __restrict__ parts, typename CsSpl<K>::type* __restrict__ splitters, const SkfCtl* __restrict__ ctl = nullptr,
                    CsLut* __restrict__ luts = nullptr /* [partitions]; with tshift */, u32 tshift = 0)
{
  typedef typename CsSpl<K>::type S_t;
  constexpr u32 SMAX = (u32)CsCap<K>::sample;
  __shared__ S_t sm[SMAX];
  __shared__ u32 tsp[CS_MAXB];      // (luts) t of the partition's splitters
  if (ctl && ctl->status) return;      // (the sync-free path: the tables may name more than this kernel takes -- the call goes the old way)
  const CsPart P = parts[blockIdx.x];
  if (P.nb <= 1) return;
  const u32 tid = threadIdx.x;
  u32 S = 8 * P.nb; { u32 p2 = 64; while (p2 < S) p2 <<= 1; S = min(p2, SMAX); }      // samples: a power of two, 8-16 per bucket (4-8 for a partition of more than SMAX / 8 buckets; 16-32 cost the samples' sort 60 us a sample more than they saved)
  // (a sample per stratum of nkeys / S keys, at a hashed place inside it: evenly spaced samples of a batch that holds the same reads
  //  twice -- a genome given twice, paired files -- are the same keys twice, half as many samples as it looks)
  for (u32 i = tid; i < S; i += CS_SPL_TPB) {
    const u32 lo = (u32)(((u64)i * P.nkeys) / S), hi = (u32)(((u64)(i + 1) * P.nkeys) / S);
    sm[i] = CsSpl<K>::top(keys[P.key0 + lo + (hi > lo ? (i * 2654435761u >> 7) % (hi - lo) : 0u)]);
  }
  __syncthreads();
  cs_sort_lds<S_t, CS_SPL_TPB>(sm, S, tid);
  // bucket b holds the keys k with splitter[b - 1] <= k < splitter[b]
  for (u32 b = tid; b + 1 < P.nb; b += CS_SPL_TPB) {
    const S_t v = sm[(u32)(((u64)(b + 1) * S) / P.nb)];
    splitters[(u64)P.bucket0 + b] = v;
    if constexpr (sizeof(S_t) == sizeof(K)) { if (luts) tsp[b] = cs_t32<S_t>(v, tshift); }
  }
  if constexpr (sizeof(S_t) == sizeof(K)) {
    if (luts) {
      __syncthreads();
      const u32 ns = P.nb - 1, tmin = tsp[0], tmax = tsp[ns - 1];
      u32 lsh = 0; while (((tmax - tmin) >> lsh) >= CS_LUT) lsh++;
      CsLut& L = luts[blockIdx.x];
      if (tid == 0) { L.tmin = tmin; L.lsh = lsh; }
      for (u32 c = tid; c <= CS_LUT; c += CS_SPL_TPB) {
        u32 lo = 0, hi = ns;
        if (c == CS_LUT) lo = ns;
        else {
          const u64 start = (u64)tmin + ((u64)c << lsh);      // splitters with t < start
          while (lo < hi) { const u32 mid = (lo + hi) >> 1; if ((u64)tsp[mid] < start) lo = mid + 1; else hi = mid; }
        }
        L.lb[c] = (u16)lo;
      }
    }
  }
}
// the bucket of key k (= the number of splitters <= k) with the table in front: lt = the partition's table in LDS
template <typename S_t> __device__ __forceinline__ u32 cs_bucket_lut(const S_t* spl, const CsLut* lt, u32 tshift, S_t k)
{
  const u32 t = cs_t32<S_t>(k, tshift);
  const u32 c = t <= lt->tmin ? 0u : min(CS_LUT - 1u, (t - lt->tmin) >> lt->lsh);
  u32 lo = lt->lb[c], hi = lt->lb[c + 1];
  while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (spl[mid] <= k) lo = mid + 1; else hi = mid; }
  return lo;
}

template <typename K> __device__ __forceinline__ u32 cs_bucket(const K* spl, u32 nb, K k)
{ // number of splitters <= k
  u32 lo = 0, hi = nb - 1;
  while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (spl[mid] <= k) lo = mid + 1; else hi = mid; }
  return lo;
}
// SCATTER = false: bucket sizes.  SCATTER = true: keys to their buckets (cursor[] starts at the buckets' offsets).
// (Round 3 tried ordering the chunk by bucket in LDS first, so that the ~5 keys a chunk holds for a bucket leave as adjacent lanes
//  of one store: 0.32 -> 0.36 ms for the 24 M k-mer sample -- the pieces are as small either way; not kept.)
template <typename K, bool SCATTER>
__global__ __launch_bounds__(CS_WALK_TPB)
void k_cs_walk(const K* __restrict__ keys, const CsPart* __restrict__ parts, const CsChunk* __restrict__ chunks, const typename CsSpl<K>::type* __restrict__ splitters,
               u32* __restrict__ counts_or_cursor, K* __restrict__ out,
               const SkfCtl* __restrict__ ctl = nullptr, const u32* __restrict__ cfirst = nullptr /* [n_parts + 1] first chunk of every partition */, u32 n_parts = 0,
               const CsLut* __restrict__ luts = nullptr, u32 tshift = 0)
{
  typedef typename CsSpl<K>::type S_t;
  constexpr int IPT = CsCap<K>::walk;      // (the chunks are cut to IPT * CS_WALK_TPB keys by the host: cs_chunk<K>())
  __shared__ S_t spl[CS_MAXB];
  __shared__ u32 hist[CS_MAXB];
  __shared__ u32 base[CS_MAXB];
  __shared__ CsLut lt;
  CsChunk C;
  if (ctl) {      // the sync-free path: no chunk table -- chunk blockIdx.x belongs to the last partition whose first chunk is at or before it
    if (ctl->status || blockIdx.x >= ctl->NC) return;
    u32 lo = 0, hi = n_parts;
    while (lo + 1 < hi) { const u32 mid = (lo + hi) >> 1; if (cfirst[mid] <= blockIdx.x) lo = mid; else hi = mid; }
    const CsPart Q = parts[lo];
    const u32 o = (blockIdx.x - cfirst[lo]) * cs_chunk<K>();
    C.part = lo; C.key0 = Q.key0 + o; C.nkeys = min(cs_chunk<K>(), Q.nkeys - o); C.pad = 0;
  } else C = chunks[blockIdx.x];
  const CsPart P = parts[C.part];
  const u32 tid = threadIdx.x;
  for (u32 b = tid; b < P.nb; b += CS_WALK_TPB) { hist[b] = 0; if (b + 1 < P.nb) spl[b] = splitters[(u64)P.bucket0 + b]; }
  const bool use_lut = luts != nullptr && P.nb > 1 && sizeof(S_t) == sizeof(K);
  if (use_lut) { const u32* src = reinterpret_cast<const u32*>(luts + C.part); u32* dst = reinterpret_cast<u32*>(&lt); for (u32 i = tid; i < sizeof(CsLut) / 4; i += CS_WALK_TPB) dst[i] = src[i]; }
  __syncthreads();
  K k[IPT]; u32 bk[IPT], rk[IPT];
  // (round 6 tried the IPT searches in step -- every key the same stride at the same time, IPT LDS reads in flight a thread: 70 -> 81 us
  //  for the count walk of the 24 M k-mer sample, +-0 for the staged scatter: the registers it takes cost more than the latency it hides)
#pragma unroll
  for (int x = 0; x < IPT; x++) {
    const u32 i = tid + x * CS_WALK_TPB;
    bk[x] = 0xFFFFFFFFu;
    if (i < C.nkeys) {
      k[x] = keys[C.key0 + i];
      if constexpr (sizeof(S_t) == sizeof(K)) bk[x] = P.nb > 1 ? (use_lut ? cs_bucket_lut<S_t>(spl, &lt, tshift, k[x]) : cs_bucket<S_t>(spl, P.nb, k[x])) : 0u;
      else bk[x] = P.nb > 1 ? cs_bucket<S_t>(spl, P.nb, CsSpl<K>::top(k[x])) : 0u;
      rk[x] = atomicAdd(&hist[bk[x]], 1u);
    }
  }
  __syncthreads();
  if (!SCATTER) { for (u32 b = tid; b < P.nb; b += CS_WALK_TPB) if (hist[b]) atomicAdd(&counts_or_cursor[P.bucket0 + b], hist[b]); return; }
  for (u32 b = tid; b < P.nb; b += CS_WALK_TPB) base[b] = hist[b] ? atomicAdd(&counts_or_cursor[P.bucket0 + b], hist[b]) : 0u;
  __syncthreads();
#pragma unroll
  for (int x = 0; x < IPT; x++) if (bk[x] != 0xFFFFFFFFu) out[base[bk[x]] + rk[x]] = k[x];
}

// ---- the scatter walk, its pieces put together in LDS first (round 6, the sync-free path).  k_cs_walk<.., true> writes a key where its
//      bucket's cursor says: 64 lanes, 64 cache lines, 8 bytes each -- 24 us of a workgroup's 35 are these stores.  Here a workgroup
//      takes 64 KB of keys (half a count chunk of 8-byte keys), ranks them per bucket as before, lays them out bucket after bucket in
//      LDS (a scan of the chunk's bucket sizes), claims each bucket's room with one global add, and writes the staged keys in order:
//      a bucket's ~36 keys of the chunk leave as adjacent lanes of one store. ----
template <typename K> __host__ __device__ constexpr u32 cs_schunk() { return 65536u / (u32)sizeof(K); }      // keys a workgroup of the staged scatter takes
template <typename K>
__global__ __launch_bounds__(CS_WALK_TPB)
void k_cs_scatter_staged(const K* __restrict__ keys, const CsPart* __restrict__ parts, const typename CsSpl<K>::type* __restrict__ splitters,
                         u32* __restrict__ cursor, K* __restrict__ out, const SkfCtl* __restrict__ ctl, const u32* __restrict__ cfirst, u32 n_parts,
                         const CsLut* __restrict__ luts = nullptr, u32 tshift = 0)
{
  typedef typename CsSpl<K>::type S_t;
  constexpr u32 SC = cs_schunk<K>(), IPT = SC / CS_WALK_TPB, SUB = cs_chunk<K>() / SC;      // (SUB staged chunks per count chunk)
  static_assert(cs_chunk<K>() % SC == 0 && SC % CS_WALK_TPB == 0, "a count chunk is cut into whole staged chunks");
  __shared__ S_t spl[CS_MAXB];
  __shared__ u32 hist[CS_MAXB];       // keys of the chunk per bucket; then: where the bucket starts in `stage`
  __shared__ u32 delta[CS_MAXB];      // the bucket's place in `out` minus its place in `stage`
  __shared__ K stage[SC];
  __shared__ u16 bid[SC];
  __shared__ u32 wsum[CS_WALK_TPB / 64];
  __shared__ CsLut lt;
  if (ctl->status) return;
  const u32 cc = blockIdx.x / SUB, sub = blockIdx.x % SUB;
  if (cc >= ctl->NC) return;
  u32 plo = 0, phi = n_parts;
  while (plo + 1 < phi) { const u32 mid = (plo + phi) >> 1; if (cfirst[mid] <= cc) plo = mid; else phi = mid; }
  const CsPart P = parts[plo];
  const u32 o = (cc - cfirst[plo]) * cs_chunk<K>() + sub * SC;
  if (o >= P.nkeys) return;
  const u32 key0 = P.key0 + o, nkeys = min(SC, P.nkeys - o);
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  for (u32 b = tid; b < P.nb; b += CS_WALK_TPB) { hist[b] = 0; if (b + 1 < P.nb) spl[b] = splitters[(u64)P.bucket0 + b]; }
  const bool use_lut = luts != nullptr && P.nb > 1 && sizeof(S_t) == sizeof(K);
  if (use_lut) { const u32* src = reinterpret_cast<const u32*>(luts + plo); u32* dst = reinterpret_cast<u32*>(&lt); for (u32 i = tid; i < sizeof(CsLut) / 4; i += CS_WALK_TPB) dst[i] = src[i]; }
  __syncthreads();
  K k[IPT]; u32 bk[IPT], rk[IPT];
#pragma unroll
  for (u32 x = 0; x < IPT; x++) {
    const u32 i = tid + x * CS_WALK_TPB;
    bk[x] = 0xFFFFFFFFu;
    if (i < nkeys) {
      k[x] = keys[key0 + i];
      if constexpr (sizeof(S_t) == sizeof(K)) bk[x] = P.nb > 1 ? (use_lut ? cs_bucket_lut<S_t>(spl, &lt, tshift, k[x]) : cs_bucket<S_t>(spl, P.nb, k[x])) : 0u;
      else bk[x] = P.nb > 1 ? cs_bucket<S_t>(spl, P.nb, CsSpl<K>::top(k[x])) : 0u;
      rk[x] = atomicAdd(&hist[bk[x]], 1u);
    }
  }
  __syncthreads();
  // the buckets' places in `stage`: exclusive scan of hist (two entries a thread: CS_MAXB = 2 * CS_WALK_TPB), their room in `out`
  static_assert(CS_MAXB == 2 * CS_WALK_TPB, "two buckets a thread");
  const u32 h0 = 2 * tid < P.nb ? hist[2 * tid] : 0u, h1 = 2 * tid + 1 < P.nb ? hist[2 * tid + 1] : 0u;
  const u32 incl = wave_incl_scan(h0 + h1, (int)lane);
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  u32 a = incl - (h0 + h1);
  for (u32 w = 0; w < wave; w++) a += wsum[w];
  if (2 * tid < P.nb) { hist[2 * tid] = a; delta[2 * tid] = (h0 ? atomicAdd(&cursor[P.bucket0 + 2 * tid], h0) : 0u) - a; }
  if (2 * tid + 1 < P.nb) { hist[2 * tid + 1] = a + h0; delta[2 * tid + 1] = (h1 ? atomicAdd(&cursor[P.bucket0 + 2 * tid + 1], h1) : 0u) - (a + h0); }
  __syncthreads();
#pragma unroll
  for (u32 x = 0; x < IPT; x++) if (bk[x] != 0xFFFFFFFFu) { const u32 at = hist[bk[x]] + rk[x]; stage[at] = k[x]; bid[at] = (u16)bk[x]; }
  __syncthreads();
  for (u32 i = tid; i < nkeys; i += CS_WALK_TPB) out[i + delta[bid[i]]] = stage[i];
}

// exclusive scan of n u32 values (n <= a few 100 k): one workgroup, 1024 threads; out[n] = total.  What the caller would otherwise
// queue as calls of their own rides along: a second copy of out[0, n) (the scatter's cursors), and a word brought next to the
// result (out[n + 1] = *flag: one download for both)
__global__ __launch_bounds__(1024)
void k_cs_scan(const u32* __restrict__ in, u32 n, u32* __restrict__ out, u32* __restrict__ out2, const u32* __restrict__ flag, const u32* __restrict__ n_dev = nullptr)
{
  __shared__ u32 wsum[16];
  __shared__ u32 carry_s;
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (n_dev) n = *n_dev;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  // tiles of 16 consecutive values per thread (round 5; 4 before: the kernel is one workgroup whose tiles follow each other at the
  // latency of their loads -- 12 of them for a sample's 48 k buckets, 33 us; now 3)
  constexpr u32 VPT = 16;
  for (u32 t0 = 0; t0 < n; t0 += 1024u * VPT) {
    const u32 i = t0 + tid * VPT;
    u32 v[VPT];
    if (i + VPT <= n && ((uintptr_t)in & 15u) == 0) {
#pragma unroll
      for (u32 x = 0; x < VPT; x += 4) { const uint4 q = *reinterpret_cast<const uint4*>(in + i + x); v[x] = q.x; v[x + 1] = q.y; v[x + 2] = q.z; v[x + 3] = q.w; }
    } else {
#pragma unroll
      for (u32 x = 0; x < VPT; x++) v[x] = i + x < n ? in[i + x] : 0u;
    }
    u32 s = 0;
#pragma unroll
    for (u32 x = 0; x < VPT; x++) s += v[x];
    const u32 incl = wave_incl_scan(s, (int)lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u32 a = carry_s + incl - s, tot = 0;
#pragma unroll
    for (u32 w = 0; w < 16; w++) { const u32 x = wsum[w]; if (w < wave) a += x; tot += x; }
#pragma unroll
    for (u32 x = 0; x < VPT; x++) { if (i + x < n) { out[i + x] = a; if (out2) out2[i + x] = a; } a += v[x]; }
    __syncthreads();
    if (tid == 0) carry_s += tot;
    __syncthreads();
  }
  if (tid == 0) { out[n] = carry_s; if (flag) out[n + 1] = *flag; }
}

// the same scan over several workgroups (round 6, the sync-free path: n is the device's, the grid covers a bound): workgroup g takes
// values [4096 g, 4096 (g + 1)), publishes their sum, adds up the sums of the workgroups in front of it (at most 255 of them: all
// resident) -- one tile's latency instead of a tile after a tile.  flags[gridDim.x] zeroed by the caller.
__global__ __launch_bounds__(1024)
void k_cs_scan_mw(const u32* __restrict__ in, const u32* __restrict__ n_dev, u32* __restrict__ out, u32* __restrict__ out2, u32* __restrict__ agg, u32* __restrict__ flags)
{
  __shared__ u32 wsum[16];
  __shared__ u32 base_s;
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, g = blockIdx.x, n = *n_dev;
  const u32 i = g * 4096u + tid * 4u;
  u32 v[4] = {0, 0, 0, 0};
  if (i + 4 <= n && ((uintptr_t)in & 15u) == 0) { const uint4 q = *reinterpret_cast<const uint4*>(in + i); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
  else { for (u32 x = 0; x < 4; x++) if (i + x < n) v[x] = in[i + x]; }
  const u32 s = v[0] + v[1] + v[2] + v[3];
  const u32 incl = wave_incl_scan(s, (int)lane);
  if (lane == 63) wsum[wave] = incl;
  if (tid == 0) base_s = 0;
  __syncthreads();
  u32 a = incl - s, tot = 0;
#pragma unroll
  for (u32 w = 0; w < 16; w++) { const u32 x = wsum[w]; if (w < wave) a += x; tot += x; }
  if (tid == 0) { agg[g] = tot; __threadfence(); __hip_atomic_store(&flags[g], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
  if (tid < g) {
    while (__hip_atomic_load(&flags[tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(1);
    const u32 x = __hip_atomic_load(&agg[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (x) atomicAdd(&base_s, x);
  }
  __syncthreads();
  a += base_s;
#pragma unroll
  for (u32 x = 0; x < 4; x++) { if (i + x < n) { out[i + x] = a; if (out2) out2[i + x] = a; } a += v[x]; }
  // out[n] = the total: the thread whose values end at n (or thread 0 of workgroup 0 when there is nothing)
  if (n == 0) { if (g == 0 && tid == 0) out[0] = 0; }
  else if (i < n && i + 4 >= n) out[n] = a;
}

// ---- a bucket by sorting: keys into LDS, bitonic sort, run starts by neighbour compare, run lengths = counts (n <= CsCap<K>::cap).
//      LDS arrays are the caller's: sk[cap], starts[cap], wsum[CS_TPB / 64], hh[258] ----
template <typename K, int CAP = CsCap<K>::cap>
__device__ __forceinline__ void cs_bucket_by_sort(K* sk, u32* starts, u32* wsum, u32* hh, const K* __restrict__ bkeys, u32 o, u32 n, u32 b, u32 hard_min,
                                                  K* __restrict__ tk, u32* __restrict__ tc, u32* __restrict__ nkept, unsigned long long* __restrict__ hist)
{
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (hist) for (u32 i = tid; i < 258; i += CS_TPB) hh[i] = 0;
  u32 Pn = 2; while (Pn < n) Pn <<= 1;
  for (u32 i = tid; i < Pn; i += CS_TPB) sk[i] = i < n ? bkeys[o + i] : cs_max<K>();
  __syncthreads();
  cs_sort_lds<K>(sk, Pn, tid);
  // run starts among the n real keys (pads sort last; a real key may equal the pad value: positions decide, not values)
  constexpr int PT = CAP / CS_TPB;
  u32 mine = 0, m = 0;
#pragma unroll
  for (int x = 0; x < PT; x++) {
    const u32 i = tid * PT + x;
    if (i < n) { const bool st = i == 0 || sk[i - 1] != sk[i]; m |= (st ? 1u : 0u) << x; mine += st ? 1u : 0u; }
  }
  const u32 incl = wave_incl_scan(mine, (int)lane);
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  u32 r = incl - mine, nruns = 0;
  for (u32 w = 0; w < CS_TPB / 64; w++) { if (w < wave) r += wsum[w]; nruns += wsum[w]; }
#pragma unroll
  for (int x = 0; x < PT; x++) if ((m >> x) & 1u) starts[r++] = tid * PT + x;
  __syncthreads();
  // kept runs, in order
  u32 kept = 0;
  const u32 per = (nruns + CS_TPB - 1) / CS_TPB;      // consecutive runs per thread
  for (u32 x = 0; x < per; x++) {
    const u32 j = tid * per + x;
    if (j < nruns) {
      const u32 len = (j + 1 < nruns ? starts[j + 1] : n) - starts[j];
      if (len >= hard_min) kept++;
      if (hist) { if (len <= 255u) atomicAdd(&hh[len], 1u); else { atomicAdd(&hh[256], 1u); atomicAdd(&hh[257], len); } }      // (a bucket holds < 2^32 keys)
    }
  }
  const u32 incl2 = wave_incl_scan(kept, (int)lane);
  __syncthreads();
  if (lane == 63) wsum[wave] = incl2;
  __syncthreads();
  u32 w0 = incl2 - kept, tot = 0;
  for (u32 w = 0; w < CS_TPB / 64; w++) { if (w < wave) w0 += wsum[w]; tot += wsum[w]; }
  for (u32 x = 0; x < per; x++) {
    const u32 j = tid * per + x;
    if (j < nruns) {
      const u32 s = starts[j], len = (j + 1 < nruns ? starts[j + 1] : n) - s;
      if (len >= hard_min) { tk[o + w0] = sk[s]; tc[o + w0] = len; w0++; }      // (a run is at most the bucket: no saturation below 2^32)
    }
  }
  if (tid == 0) nkept[b] = tot;
  if (hist) {      // (the barriers above separate the histogram's atomics from these reads)
    for (u32 i = tid; i < 258; i += CS_TPB) if (hh[i]) atomicAdd(&hist[i], (unsigned long long)hh[i]);
  }
}

// a workgroup per bucket: sort in LDS, run-length count, hard-min.  kept pairs -> tk / tc at the bucket's offset, their number -> nkept.
// A bucket over the LDS capacity raises *overflow (the caller then takes the library sort for the batch).
// (CAP: the LDS is sized by it -- 128-bit keys: a launch for up to 2048 keys at 40 KB, one for up to 4096 at 80 KB behind it, so that the
//  rare bucket does not take the usual one's occupancy; last: this launch reports the buckets beyond its CAP)
template <typename K, int CAP = CsCap<K>::cap>
__global__ __launch_bounds__(CS_TPB)
void k_cs_sort(const K* __restrict__ bkeys, const u32* __restrict__ boff, u32 hard_min, K* __restrict__ tk, u32* __restrict__ tc, u32* __restrict__ nkept,
               unsigned long long* __restrict__ hist, u32* __restrict__ overflow, u32 lo /* buckets of at most lo keys are another kernel's */, u32 last = 1,
               const SkfCtl* __restrict__ ctl = nullptr, const u32* __restrict__ big = nullptr /* set: the buckets to take are listed (the sync-free path), ctl->n_big of them */)
{
  __shared__ K sk[CAP];
  __shared__ u32 starts[CAP];      // positions of the run starts, in order
  __shared__ u32 wsum[CS_TPB / 64];
  __shared__ u32 hh[258];          // abundance histogram of the bucket's runs (hist != nullptr): see kmx_ctx::d_hist
  if (big && ctl->status) return;
  const u32 nlist = big ? min(ctl->n_big, SKF_BIG_CAP) : 0u;
  for (u32 it = blockIdx.x; big ? it < nlist : it == blockIdx.x; it += gridDim.x) {
    __syncthreads();      // (the LDS of the bucket before)
    const u32 b = big ? big[it] : blockIdx.x;
    const u32 o = boff[b], n = boff[b + 1] - o;
    if (lo && n <= lo) continue;
    if (n == 0) { if (threadIdx.x == 0) nkept[b] = 0; continue; }
    if (n > (u32)CAP) { if (last && threadIdx.x == 0) { nkept[b] = 0; atomicOr(overflow, 1u); } continue; }
    cs_bucket_by_sort<K, CAP>(sk, starts, wsum, hh, bkeys, o, n, b, hard_min, tk, tc, nkept, hist);
  }
}

// ---- 64-bit keys: a bucket by HASHING first.  The keys of a bucket are mostly repeats (a k-mer is seen once per read that covers it:
//      6 times at 6x), and sorting repeats is the expensive way to count them.  Every key goes into an LDS hash set (key claimed with
//      one ds_cmpst_b64, count with one ds_add -- the keys are STREAMED from HBM, so a k-mer repeated 100 000 times is no overflow any
//      more), then only the DISTINCT keys are compacted and sorted (a bitonic network over a few hundred keys instead of a few
//      thousand) and each looks its count up.  More than CS_HT_LIMIT distinct keys: the bucket is sorted as before when it fits the
//      LDS, else *overflow is raised. ----
constexpr int CS_HT = 2048;                 // hash set entries (keys 16 KB + counts 8 KB; the distinct keys' dense copy: 16 KB)
constexpr int CS_HT_LIMIT = CS_HT - 256;    // distinct keys it takes (load 0.875)
__device__ __forceinline__ u32 cs_hash(u64 k) { return (u32)((k * 0x9E3779B97F4A7C15ULL) >> (64 - 11)); }
static_assert(CS_HT == (1 << 11), "cs_hash takes the top 11 bits of the product");

__global__ __launch_bounds__(CS_TPB)
void k_cs_count_hash(const u64* __restrict__ bkeys, const u32* __restrict__ boff, u32 hard_min, u64* __restrict__ tk, u32* __restrict__ tc, u32* __restrict__ nkept,
                     unsigned long long* __restrict__ hist, u32* __restrict__ overflow, u32 lo /* buckets of at most lo keys are another kernel's */,
                     const SkfCtl* __restrict__ ctl = nullptr, const u32* __restrict__ big = nullptr /* set: the buckets to take are listed (the sync-free path), ctl->n_big of them */)
{
  constexpr int CAP = CsCap<u64>::cap;
  __shared__ u64 la[CAP];          // hash set keys [0, CS_HT) + the distinct keys' dense copy [CS_HT, 2 * CS_HT)  |  the sort path's keys
  __shared__ u32 lb[CAP];          // hash set counts [0, CS_HT)                                                 |  the sort path's run starts
  __shared__ u32 wsum[CS_TPB / 64];
  __shared__ u32 hh[258];
  __shared__ u32 ndist;
  __shared__ volatile u32 full;
  static_assert(2 * CS_HT <= CAP, "the dense copy lies behind the hash set");
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (big && ctl->status) return;
  const u32 nlist = big ? min(ctl->n_big, SKF_BIG_CAP) : 0u;
  for (u32 it = blockIdx.x; big ? it < nlist : it == blockIdx.x; it += gridDim.x) {
  __syncthreads();      // (the LDS of the bucket before)
  const u32 b = big ? big[it] : blockIdx.x;
  const u32 o = boff[b], n = boff[b + 1] - o;
  if (lo && n <= lo) continue;
  if (n == 0) { if (tid == 0) nkept[b] = 0; continue; }
  const u64 EMPTY = ~0ULL;           // (no canonical k-mer and no window hash is all ones)
  for (u32 i = tid; i < (u32)CS_HT; i += CS_TPB) { la[i] = EMPTY; lb[i] = 0; }
  if (tid == 0) { ndist = 0; full = 0; }
  if (hist) for (u32 i = tid; i < 258; i += CS_TPB) hh[i] = 0;
  __syncthreads();
  for (u32 i0 = 0; i0 < n; i0 += CS_TPB) {
    const u32 i = i0 + tid;
    if (i < n) {
      const u64 k = bkeys[o + i];
      u32 h = cs_hash(k);
      for (u32 probe = 0; probe < (u32)CS_HT; probe++) {
        u64 cur = ((volatile u64*)la)[h];
        if (cur == EMPTY) {
          cur = (u64)atomicCAS((unsigned long long*)&la[h], (unsigned long long)EMPTY, (unsigned long long)k);
          if (cur == EMPTY) { if (atomicAdd(&ndist, 1u) + 1u > (u32)CS_HT_LIMIT) full = 1; cur = k; }
        }
        if (cur == k) { atomicAdd(&lb[h], 1u); break; }
        h = (h + 1) & (CS_HT - 1);
        if (full) break;
      }
    }
    if (full) break;                 // (uniform enough: every thread leaves at its next look)
  }
  __syncthreads();
  if (full) {                        // too many distinct keys for the hash set
    if (n > (u32)CAP) { if (tid == 0) { nkept[b] = 0; atomicOr(overflow, 1u); } continue; }
    __syncthreads();
    cs_bucket_by_sort<u64>(la, lb, wsum, hh, bkeys, o, n, b, hard_min, tk, tc, nkept, hist);
    continue;
  }
  // the distinct keys, dense: every thread's CS_HT / CS_TPB slots
  constexpr int SPT = CS_HT / CS_TPB;
  u64* const dk = la + CS_HT;
  u32 mine = 0;
#pragma unroll
  for (int x = 0; x < SPT; x++) mine += la[tid * SPT + x] != EMPTY ? 1u : 0u;
  const u32 incl = wave_incl_scan(mine, (int)lane);
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  u32 r = incl - mine, nd = 0;
  for (u32 w = 0; w < CS_TPB / 64; w++) { if (w < wave) r += wsum[w]; nd += wsum[w]; }
#pragma unroll
  for (int x = 0; x < SPT; x++) { const u64 k = la[tid * SPT + x]; if (k != EMPTY) dk[r++] = k; }
  u32 Pn = 2; while (Pn < nd) Pn <<= 1;
  __syncthreads();
  for (u32 i = nd + tid; i < Pn; i += CS_TPB) dk[i] = EMPTY;
  __syncthreads();
  cs_sort_lds<u64>(dk, Pn, tid);
  // counts of the sorted keys from the hash set; kept ones out in order
  const u32 per = (nd + CS_TPB - 1) / CS_TPB;          // consecutive keys per thread
  u32 kept = 0;
  for (u32 x = 0; x < per; x++) {
    const u32 j = tid * per + x;
    if (j < nd) {
      const u64 k = dk[j];
      u32 h = cs_hash(k);
      while (la[h] != k) h = (h + 1) & (CS_HT - 1);
      const u32 len = lb[h];
      if (len >= hard_min) kept++;
      if (hist) { if (len <= 255u) atomicAdd(&hh[len], 1u); else { atomicAdd(&hh[256], 1u); atomicAdd(&hh[257], len); } }
    }
  }
  const u32 incl2 = wave_incl_scan(kept, (int)lane);
  __syncthreads();
  if (lane == 63) wsum[wave] = incl2;
  __syncthreads();
  u32 w0 = incl2 - kept, tot = 0;
  for (u32 w = 0; w < CS_TPB / 64; w++) { if (w < wave) w0 += wsum[w]; tot += wsum[w]; }
  for (u32 x = 0; x < per; x++) {
    const u32 j = tid * per + x;
    if (j < nd) {
      const u64 k = dk[j];
      u32 h = cs_hash(k);
      while (la[h] != k) h = (h + 1) & (CS_HT - 1);
      const u32 len = lb[h];
      if (len >= hard_min) { tk[o + w0] = k; tc[o + w0] = len; w0++; }
    }
  }
  if (tid == 0) nkept[b] = tot;
  if (hist) {
    for (u32 i = tid; i < 258; i += CS_TPB) if (hh[i]) atomicAdd(&hist[i], (unsigned long long)hh[i]);
  }
  }
}

// ---- a bucket by ONE WAVE, the keys in registers (round 3, second half).  The kernels above give a bucket to a workgroup and
//      sort or hash it in LDS: ~1000 keys per bucket are 4 per thread, and a dozen workgroup barriers with three workgroups per CU
//      (48 KB of LDS each) is what their time is made of (0.39 ms for a 24 M k-mer sample: 0.5 TB/s).  Here a wave holds NPL keys
//      per lane (64 * NPL >= the bucket; loaded coalesced -- a sorting network takes any initial order), runs a bitonic network over
//      them in the blocked layout (element e = lane * NPL + x: steps closer than NPL are compare-exchanges between a lane's own
//      registers, the others one lane shuffle per register), and reads the runs off the sorted registers: run starts by neighbour
//      compare (one shuffle for the lane's first key), the length of a lane's last run from a ballot of the lanes that hold a start.
//      No LDS, no barrier (the abundance histogram, when asked for, is the exception), occupancy bound by registers only.
//      Two launches cover the sizes (the register file is sized by a kernel's largest NPL): LO < n <= 64 * NPL_B, a wave picking
//      NPL_A (n <= 64 * NPL_A) or NPL_B.  Larger buckets raise *overflow as before. ----
template <typename K> __device__ __forceinline__ K cs_shfl_up1(K k);
template <> __device__ __forceinline__ u64 cs_shfl_up1<u64>(u64 k) { return (u64)__shfl_up((unsigned long long)k, 1); }
template <> __device__ __forceinline__ __uint128_t cs_shfl_up1<__uint128_t>(__uint128_t k)
{
  const u64 lo = (u64)__shfl_up((unsigned long long)(u64)k, 1), hi = (u64)__shfl_up((unsigned long long)(u64)(k >> 64), 1);
  return ((__uint128_t)hi << 64) | lo;
}

template <int KW> __device__ __forceinline__ WideKey<KW> cs_shfl_up1_wide(const WideKey<KW>& k)
{
  WideKey<KW> r;
#pragma unroll
  for (int i = 0; i < KW; i++) r.w[i] = (u64)__shfl_up((unsigned long long)k.w[i], 1);
  return r;
}
template <> __device__ __forceinline__ WideKey<3> cs_shfl_up1<WideKey<3>>(WideKey<3> k) { return cs_shfl_up1_wide<3>(k); }
template <> __device__ __forceinline__ WideKey<4> cs_shfl_up1<WideKey<4>>(WideKey<4> k) { return cs_shfl_up1_wide<4>(k); }

// the network, a step per instantiation (stage K2, distance J): every register index is a constant
template <typename K, int NPL, u32 K2, u32 J>
__device__ __forceinline__ void cs_wave_net(K (&k)[NPL], u32 lane)
{
  constexpr u32 N = 64u * NPL;
  if constexpr (J >= (u32)NPL) {          // partner: the same register of lane ^ (J / NPL)
    constexpr u32 m = J / (u32)NPL;
    const bool asc = K2 >= N ? true : ((lane & (K2 / (u32)NPL)) == 0);      // (K2 > J >= NPL: the direction is a lane bit)
    const bool keep_min = ((lane & m) == 0) == asc;
#pragma unroll
    for (int x = 0; x < NPL; x++) {
      const K p = CsXor<K, (int)m>::get(k[x]);
      if ((p < k[x]) == keep_min) k[x] = p;      // (an equal partner may be taken: nothing changes)
    }
  } else {                                // both keys are mine
#pragma unroll
    for (int x = 0; x < NPL; x++) {
      if ((x & (int)J) == 0) {
        const bool asc = K2 >= (u32)NPL ? (K2 >= N ? true : ((lane & (K2 / (u32)NPL)) == 0)) : (((u32)x & K2) == 0);
        const K a = k[x], c = k[x | (int)J];
        if ((c < a) == asc) { k[x] = c; k[x | (int)J] = a; }
      }
    }
  }
  if constexpr (J > 1) cs_wave_net<K, NPL, K2, J / 2>(k, lane);
  else if constexpr (K2 < N) cs_wave_net<K, NPL, K2 * 2, K2>(k, lane);
}

template <typename K, int NPL>
__device__ __forceinline__ void cs_wave_bucket(const K* __restrict__ bkeys, u32 o, u32 n, u32 b, u32 hard_min, K* __restrict__ tk, u32* __restrict__ tc,
                                               u32* __restrict__ nkept, u32* hh /* LDS histogram or null */)
{
  const u32 lane = threadIdx.x & 63u;
  K k[NPL];
#pragma unroll
  for (int x = 0; x < NPL; x++) { const u32 i = (u32)x * 64u + lane; k[x] = i < n ? bkeys[o + i] : cs_max<K>(); }
  cs_wave_net<K, NPL, 2, 1>(k, lane);      // bitonic network, blocked layout
  // run starts among the n real keys (positions decide: pads sort last, a real key may equal the pad value)
  const u32 e0 = lane * (u32)NPL;
  const K prev0 = cs_shfl_up1<K>(k[NPL - 1]);
  u64 sm = 0;                                   // bit x: a run starts at my key x
#pragma unroll
  for (int x = 0; x < NPL; x++) {
    const bool st = (e0 + (u32)x < n) && (x == 0 ? (lane == 0 || k[0] != prev0) : (k[x] != k[x - 1]));
    sm |= (u64)(st ? 1u : 0u) << x;
  }
  // the run that is open at the end of my keys goes on to the next start: in the first lane behind me that holds one, or to n
  const u64 has = __ballot(sm != 0);
  const u64 above = lane == 63 ? 0ULL : (has >> (lane + 1)) << (lane + 1);
  const u32 nl = above ? (u32)__builtin_ctzll(above) : 64u;
  const u32 fs = sm ? (u32)__builtin_ctzll(sm) : 0u;
  const u32 nfs = (u32)__shfl((int)fs, (int)(nl & 63u));
  const u32 next_start = above ? nl * (u32)NPL + nfs : n;      // element index where the run open at my end stops
  u32 kept = 0;
#pragma unroll
  for (int x = 0; x < NPL; x++) {
    if ((sm >> x) & 1ULL) {
      const u64 later = x == 63 ? 0ULL : (sm >> (x + 1));
      const u32 len = later ? (u32)__builtin_ctzll(later) + 1u : next_start - (e0 + (u32)x);
      if (len >= hard_min) kept++;
      if (hh) { if (len <= 255u) atomicAdd(&hh[len], 1u); else { atomicAdd(&hh[256], 1u); atomicAdd(&hh[257], len); } }
    }
  }
  const u32 incl = wave_incl_scan(kept, (int)lane);
  u32 w0 = o + incl - kept;
#pragma unroll
  for (int x = 0; x < NPL; x++) {
    if ((sm >> x) & 1ULL) {
      const u64 later = x == 63 ? 0ULL : (sm >> (x + 1));
      const u32 len = later ? (u32)__builtin_ctzll(later) + 1u : next_start - (e0 + (u32)x);
      if (len >= hard_min) { tk[w0] = k[x]; tc[w0] = len; w0++; }
    }
  }
  if (lane == 63) nkept[b] = incl;
}

// ---- round 6: a bucket by counting first.  Sequencing data repeats its k-mers (coverage: five occurrences of a distinct k-mer in
// configs[2]'s samples), and the sort above is bound by its vector instructions (SQ_ACTIVE_INST_VALU: 92 % of the SIMDs' cycles): the
// bucket's keys go into a hash table of the wave's own in LDS (512 entries: compare-and-swap a key, add to its count), the entries that
// pass hard-min -- a fifth of the keys, fewer with sequencing errors -- are sorted by the 128-, 256- or 512-key network (28 / 36 / 45 steps on
// 2 / 4 / 8 registers a lane -- the sort above takes 45 / 55 steps on 8 / 16 for the bucket's 512 / 1024 keys), each sorted key looks its count up
// again.  A bucket whose distinct keys do not fit the table (data without repeats) takes the sort above.  Results identical: ascending keys, their counts.
constexpr u32 CS_HW = 512;        // table entries per wave
__device__ __forceinline__ u32 cs_hw_hash(u64 k) { return ((u32)k * 0x9E3779B1u + (u32)(k >> 32) * 0x85EBCA6Bu) >> 23; }
static_assert(CS_HW == (1u << 9), "cs_hw_hash takes the top 9 bits");
__device__ __forceinline__ void cs_wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
// the network above with a payload: (key, count) pairs, ordered by key (the keys are distinct)
template <int NPL, u32 K2, u32 J>
__device__ __forceinline__ void cs_wave_net_kv(u64 (&k)[NPL], u32 (&v)[NPL], u32 lane)
{
  constexpr u32 N = 64u * NPL;
  if constexpr (J >= (u32)NPL) {
    constexpr u32 m = J / (u32)NPL;
    const bool asc = K2 >= N ? true : ((lane & (K2 / (u32)NPL)) == 0);
    const bool keep_min = ((lane & m) == 0) == asc;
#pragma unroll
    for (int x = 0; x < NPL; x++) {
      const u64 p = xor_lane_u64<(int)m>(k[x]); const u32 pv = xor_lane_u32<(int)m>(v[x]);
      if ((p < k[x]) == keep_min) { k[x] = p; v[x] = pv; }
    }
  } else {
#pragma unroll
    for (int x = 0; x < NPL; x++) {
      if ((x & (int)J) == 0) {
        const bool asc = K2 >= (u32)NPL ? (K2 >= N ? true : ((lane & (K2 / (u32)NPL)) == 0)) : (((u32)x & K2) == 0);
        const u64 a = k[x], c = k[x | (int)J]; const u32 av = v[x], cv = v[x | (int)J];
        if ((c < a) == asc) { k[x] = c; k[x | (int)J] = a; v[x] = cv; v[x | (int)J] = av; }
      }
    }
  }
  if constexpr (J > 1) cs_wave_net_kv<NPL, K2, J / 2>(k, v, lane);
  else if constexpr (K2 < N) cs_wave_net_kv<NPL, K2 * 2, K2>(k, v, lane);
}
template <int NPL>
__device__ __forceinline__ void cs_hw_sorted_out(const u64* sk, const u32* sc, u32 d, u32 o, u64* __restrict__ tk, u32* __restrict__ tc)
{
  const u32 lane = threadIdx.x & 63u;
  u64 k[NPL]; u32 v[NPL];
#pragma unroll
  for (int x = 0; x < NPL; x++) { const u32 i = lane * (u32)NPL + (u32)x; k[x] = i < d ? sk[i] : ~0ULL; v[x] = i < d ? sc[i] : 0u; }
  cs_wave_net_kv<NPL, 2, 1>(k, v, lane);
#pragma unroll
  for (int x = 0; x < NPL; x++) { const u32 i = lane * (u32)NPL + (u32)x; if (i < d) { tk[o + i] = k[x]; tc[o + i] = v[x]; } }
}
// -> false: the bucket is not this path's (nothing written)
__device__ __forceinline__ bool cs_wave_bucket_hash(const u64* __restrict__ bkeys, u32 o, u32 n, u32 b, u32 hard_min, u64* __restrict__ tk, u32* __restrict__ tc,
                                                    u32* __restrict__ nkept, u64* hk /* [CS_HW] */, u32* hc /* [CS_HW] */)
{
  const u32 lane = threadIdx.x & 63u;
#pragma unroll
  for (u32 j = 0; j < CS_HW / 64u; j++) { hk[lane + 64u * j] = ~0ULL; hc[lane + 64u * j] = 0u; }
  cs_wave_lds_fence();
  // (the key of all ones is the table's "empty": counted beside it.)  All of a lane's keys are requested, then all their compare-and-swaps:
  // eight round trips side by side instead of one behind the other; the few that met another key in their entry walk on afterwards
  u32 n_top = 0; bool lost = false;
  for (u32 base = 0; base < n; base += 512u) {      // (a bucket of up to 1024 keys: two rounds of eight keys a lane)
    u32 pend = 0;
    u64 key[8]; u32 h[8]; u64 old[8];
#pragma unroll
    for (u32 x = 0; x < 8u; x++) { const u32 i = base + x * 64u + lane; key[x] = i < n ? bkeys[o + i] : ~0ULL; n_top += (i < n && key[x] == ~0ULL) ? 1u : 0u; }
#pragma unroll
    for (u32 x = 0; x < 8u; x++) {
      h[x] = cs_hw_hash(key[x]); old[x] = key[x];
      if (key[x] != ~0ULL) old[x] = (u64)atomicCAS(reinterpret_cast<unsigned long long*>(&hk[h[x]]), ~0ULL, (unsigned long long)key[x]);
    }
#pragma unroll
    for (u32 x = 0; x < 8u; x++) {
      if (key[x] != ~0ULL) { if (old[x] == ~0ULL || old[x] == key[x]) atomicAdd(&hc[h[x]], 1u); else pend |= 1u << x; }
    }
    for (u32 t = 0; t < 64u && __ballot(pend != 0u) != 0ULL; t++) {
#pragma unroll
      for (u32 x = 0; x < 8u; x++) {
        if ((pend >> x) & 1u) {
          h[x] = (h[x] + 1u) & (CS_HW - 1u);
          const u64 o2 = (u64)atomicCAS(reinterpret_cast<unsigned long long*>(&hk[h[x]]), ~0ULL, (unsigned long long)key[x]);
          if (o2 == ~0ULL || o2 == key[x]) { atomicAdd(&hc[h[x]], 1u); pend &= ~(1u << x); }
        }
      }
    }
    lost = lost || pend != 0u;
    if (__ballot(lost) != 0ULL) return false;
  }
  if (__ballot(lost) != 0ULL) return false;
  cs_wave_lds_fence();
  // the entries that pass hard-min, packed to the table's front (over the table itself: every lane holds its entries by then), then
  // sorted as (key, count) pairs
  u64 ek[CS_HW / 64u]; u32 ec[CS_HW / 64u]; u32 km = 0, nk = 0;
#pragma unroll
  for (u32 j = 0; j < CS_HW / 64u; j++) {
    ek[j] = hk[lane + 64u * j]; ec[j] = hc[lane + 64u * j];
    const bool keep = ek[j] != ~0ULL && ec[j] >= hard_min;
    km |= (keep ? 1u : 0u) << j; nk += keep ? 1u : 0u;
  }
  const u32 incl = wave_incl_scan(nk, (int)lane);
  const u32 d = (u32)__shfl((int)incl, 63);
  cs_wave_lds_fence();
  u32 at = incl - nk;
#pragma unroll
  for (u32 j = 0; j < CS_HW / 64u; j++) if ((km >> j) & 1u) { hk[at] = ek[j]; hc[at] = ec[j]; at++; }
  cs_wave_lds_fence();
  if (d <= 128u) cs_hw_sorted_out<2>(hk, hc, d, o, tk, tc);
  else if (d <= 256u) cs_hw_sorted_out<4>(hk, hc, d, o, tk, tc);
  else cs_hw_sorted_out<8>(hk, hc, d, o, tk, tc);
  // the key of all ones, the largest: behind the others
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) n_top += (u32)__shfl_xor((int)n_top, off);
  const bool top = n_top != 0u && n_top >= hard_min;
  if (lane == 0) { if (top) { tk[o + d] = ~0ULL; tc[o + d] = n_top; } nkept[b] = d + (top ? 1u : 0u); }
  return true;
}

#ifndef KMX_CS_WAVES
#define KMX_CS_WAVES 4
#endif
constexpr int CS_WAVES = KMX_CS_WAVES;      // buckets (waves) per workgroup
#ifndef KMX_CS_WC_OCC
#define KMX_CS_WC_OCC 6
#endif
template <typename K, int NPL_A, int NPL_B>
__global__ __launch_bounds__(64 * CS_WAVES)
void k_cs_wave_sort(const K* __restrict__ bkeys, const u32* __restrict__ boff, u32 n_buckets, u32 lo /* this launch takes lo < n <= 64 * NPL_B */, u32 cap,
                    u32 hard_min, K* __restrict__ tk, u32* __restrict__ tc, u32* __restrict__ nkept, unsigned long long* __restrict__ hist, u32* __restrict__ overflow,
                    SkfCtl* __restrict__ ctl = nullptr, u32* __restrict__ big = nullptr /* [SKF_BIG_CAP]: the buckets beyond this kernel, for the LDS kernels behind it */)
{
  __shared__ u32 hh[258];
  const u32 tid = threadIdx.x, b = blockIdx.x * CS_WAVES + (tid >> 6);
  if (ctl) { if (ctl->status) return; n_buckets = ctl->TB; }      // (the sync-free path: the grid covers a bound)
  if (hist) { for (u32 i = tid; i < 258; i += 64 * CS_WAVES) hh[i] = 0; __syncthreads(); }
  if (b < n_buckets) {
    const u32 o = boff[b], n = boff[b + 1] - o;
    // (buckets beyond 64 * NPL_B keys are the LDS kernels': the caller launches them with lo = that; they report the ones beyond them)
    if (n == 0) { if (lo == 0 && (tid & 63u) == 0) nkept[b] = 0; }
    else if (n > lo && n <= 64u * NPL_B) {
      if (n <= 64u * NPL_A) cs_wave_bucket<K, NPL_A>(bkeys, o, n, b, hard_min, tk, tc, nkept, hist ? hh : nullptr);
      else cs_wave_bucket<K, NPL_B>(bkeys, o, n, b, hard_min, tk, tc, nkept, hist ? hh : nullptr);
    } else if (big && n > 64u * NPL_B && (tid & 63u) == 0) {
      const u32 at = atomicAdd(&ctl->n_big, 1u);
      if (at < SKF_BIG_CAP) big[at] = b; else atomicOr(&ctl->status, (u32)SKF_ST_BUCKET);
    }
  }
  if (hist) {
    __syncthreads();
    for (u32 i = tid; i < 258; i += 64 * CS_WAVES) if (hh[i]) atomicAdd(&hist[i], (unsigned long long)hh[i]);
  }
}

// the sync-free path's wave kernel for 64-bit keys (round 6): every bucket of up to 1024 keys by counting first (cs_wave_bucket_hash);
// one whose distinct keys do not fit the table is sorted here when it has up to 512 keys and listed for the LDS kernels beyond -- and
// counted in *n_lost: a sample of data without repeats sends the next call to k_cs_wave_sort (the caller's choice).  6 KB of LDS a
// wave and 80 registers a lane: six waves a SIMD (the walk through LDS round trips wants them: 207 us at four, 179 at five)
__global__ __launch_bounds__(64 * CS_WAVES) __attribute__((amdgpu_waves_per_eu(KMX_CS_WC_OCC, KMX_CS_WC_OCC)))
void k_cs_wave_count(const u64* __restrict__ bkeys, const u32* __restrict__ boff, u32 hard_min, u64* __restrict__ tk, u32* __restrict__ tc, u32* __restrict__ nkept,
                     SkfCtl* __restrict__ ctl, u32* __restrict__ big, u32* __restrict__ n_lost)
{
  __shared__ u64 hw_k[CS_WAVES * CS_HW]; __shared__ u32 hw_c[CS_WAVES * CS_HW];
  const u32 tid = threadIdx.x, w = tid >> 6, b = blockIdx.x * CS_WAVES + w;
  if (ctl->status || b >= ctl->TB) return;
  const u32 o = boff[b], n = boff[b + 1] - o;
  if (n == 0) { if ((tid & 63u) == 0) nkept[b] = 0; return; }
  if (n <= 1024u && cs_wave_bucket_hash(bkeys, o, n, b, hard_min, tk, tc, nkept, hw_k + w * CS_HW, hw_c + w * CS_HW)) return;
  if (n <= 1024u && (tid & 63u) == 0) atomicAdd(n_lost, 1u);
  if (n <= 512u) { cs_wave_bucket<u64, 8>(bkeys, o, n, b, hard_min, tk, tc, nkept, nullptr); return; }
  if ((tid & 63u) == 0) {
    const u32 at = atomicAdd(&ctl->n_big, 1u);
    if (at < SKF_BIG_CAP) big[at] = b; else atomicOr(&ctl->status, (u32)SKF_ST_BUCKET);
  }
}

template <typename K>
__global__ __launch_bounds__(CS_TPB)
void k_cs_compact(const K* __restrict__ tk, const u32* __restrict__ tc, const u32* __restrict__ boff, const u32* __restrict__ koff, K* __restrict__ ok, u32* __restrict__ oc)
{
  const u32 b = blockIdx.x, src = boff[b], dst = koff[b], n = koff[b + 1] - dst;
  for (u32 i = threadIdx.x; i < n; i += CS_TPB) { ok[dst + i] = tk[src + i]; oc[dst + i] = tc[src + i]; }
}

// the same, the pairs written as packed records (key words + u32 count: the body of a .kmer file with 4-byte counts) -- bucket b's
// kept pairs become records [bdst[b], bdst[b] + kept) of `out`
template <typename K>
__global__ __launch_bounds__(CS_TPB)
void k_cs_compact_recs(const K* __restrict__ tk, const u32* __restrict__ tc, const u32* __restrict__ boff, const u32* __restrict__ koff,
                       const u32* __restrict__ bdst, u8* __restrict__ out, const SkfCtl* __restrict__ ctl = nullptr, u32 cap_recs = 0)
{
  constexpr u32 KWD = sizeof(K) / 4;      // key dwords
  u32 b = blockIdx.x, t0 = threadIdx.x, step = CS_TPB;
  if (ctl) {      // the sync-free path: the grid covers a bound, `out` is room for cap_recs records reserved on an estimate; a WAVE per bucket (a bucket keeps ~80 pairs)
    b = blockIdx.x * (CS_TPB / 64) + (threadIdx.x >> 6); t0 = threadIdx.x & 63u; step = 64;
    if (ctl->status || ctl->overflow || b >= ctl->TB || koff[ctl->TB] > cap_recs) return;
  }
  const u32 src = boff[b], n = koff[b + 1] - koff[b];
  const u64 dst = bdst[b];
  for (u32 i = t0; i < n; i += step) {
    u32* o = reinterpret_cast<u32*>(out + (dst + i) * (u64)(sizeof(K) + 4));
    const K k = tk[src + i];
#pragma unroll
    for (u32 w = 0; w < KWD; w++) o[w] = key_dword<K>(k, w);
    o[KWD] = tc[src + i];
  }
}

}  // namespace kmx
