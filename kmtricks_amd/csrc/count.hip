// count.hip -- per (sample, partition) k-mer / window-hash counting on gfx950.
// Replaces km::ReadSuperk / ReadSuperkHash (decode), KmerSort / HashSort (std::sort),
// KmerPartCounter::executeDump / HashPartCounter::executeDump (run-length) and the hard-min /
// saturate step of the count processors (reference include/kmtricks/gatb/sorting_count.hpp:141-312,
// 346-470, 488-533, 694-884, 971-990; include/kmtricks/gatb/count_processor.hpp:61-70, 135-146).
//
// Pipeline: 2-bit super-k-mer records (HBM, read coalesced-ish one thread per record) -> canonical
// k-mers rolled in registers (revcomp by bit tricks, XXH64 in registers for hash mode) -> device
// radix sort -> run-length encode -> hard-min filter.  The sort / RLE / select primitives are
// rocPRIM's (plain library ops); the decode + hash kernel is hand-written.
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <atomic>
#include <thread>
#include <rocprim/iterator/counting_iterator.hpp>
#include <cstdio>
#include <cstdlib>
#include "kmx_host.hpp"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/device/device_scan.hpp>
#include "skf.hpp"
#include "count_sort.hpp"

namespace kmx {
static bool list_copy_trace() { static const bool on = getenv("KMX_TRACE") != nullptr; return on; }


typedef __uint128_t u128;

__device__ __forceinline__ u64 rev_digits64(u64 x)
{ // reverse the 32 2-bit digits of a word
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  return __builtin_bswap64(x);
}
__device__ __forceinline__ u64 revcomp64(u64 x, int k)
{ // A0 C1 T2 G3: complement = digit ^ 2 (gatb kmer/impl/Model.hpp:857-884)
  return (rev_digits64(x) ^ 0xAAAAAAAAAAAAAAAAULL) >> (64 - 2 * k);
}
__device__ __forceinline__ u128 revcomp128(u128 x, int k)
{
  const u64 lo = (u64)x, hi = (u64)(x >> 64);
  const u128 r = ((u128)(rev_digits64(lo) ^ 0xAAAAAAAAAAAAAAAAULL) << 64) | (u128)(rev_digits64(hi) ^ 0xAAAAAAAAAAAAAAAAULL);
  return r >> (128 - 2 * k);
}

// XXH64 of 8 / 16 bytes, seed 0 (Cyan4973/xxHash specification; KmXXHash sorting_count.hpp:346-363)
#define XP1 0x9E3779B185EBCA87ULL
#define XP2 0xC2B2AE3D27D4EB4FULL
#define XP3 0x165667B19E3779F9ULL
#define XP4 0x85EBCA77C2B2AE63ULL
#define XP5 0x27D4EB2F165667C5ULL
__device__ __forceinline__ u64 rotl64d(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ u64 xxh64_round(u64 acc, u64 in) { return rotl64d(acc + in * XP2, 31) * XP1; }
__device__ __forceinline__ u64 xxh64_merge(u64 h, u64 v) { return (h ^ xxh64_round(0, v)) * XP1 + XP4; }
__device__ __forceinline__ u64 xxh64_words(const u64* w, int nw)
{
  if (nw == 4) {      // 32 bytes (Kmer<128>): one stripe through the four accumulators, nothing left over
    const u64 v1 = xxh64_round(XP1 + XP2, w[0]), v2 = xxh64_round(XP2, w[1]), v3 = xxh64_round(0, w[2]), v4 = xxh64_round(0ULL - XP1, w[3]);
    u64 h = rotl64d(v1, 1) + rotl64d(v2, 7) + rotl64d(v3, 12) + rotl64d(v4, 18);
    h = xxh64_merge(h, v1); h = xxh64_merge(h, v2); h = xxh64_merge(h, v3); h = xxh64_merge(h, v4);
    h += 32;
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    return h;
  }
  u64 h = XP5 + (u64)nw * 8;
  for (int i = 0; i < nw; i++) {
    h ^= rotl64d(w[i] * XP2, 31) * XP1;
    h = rotl64d(h, 27) * XP1 + XP4;
  }
  h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
  return h;
}

// ---- decode, one LANE per k-mer (round 3; rounds 1-2 walked a record per thread with byte loads and strided stores) ----------
// Record [u8 n][ceil((k+n-1)/4) bytes]: with S the record bytes as a little-endian integer, seed = S mod 4^k and the j-th following
// nucleotide is (S >> 2(k+j-1)) & 3 (gatb Model.hpp:1388-1433; decoder sorting_count.hpp:153-275).
// Record i of the stream starts at byte lo(prefix[i]) and its first k-mer is number hi(prefix[i]) of the batch (prefix has one more
// entry than there are records: the totals).  K-mer j of a record: the seed's digits shifted up by j, the j nucleotides that follow
// it below them in reverse order (the record appends the following nucleotides at digits k, k + 1, ...; a k-mer's lowest digit is its
// LAST nucleotide) -- two unaligned 8-byte loads (four for 128-bit keys), no loop over the record.  A workgroup takes DK consecutive
// k-mers: the prefix entries of the records that hold them go to LDS (k_decode_block_starts found the first one), a lane finds its
// record with a binary search there, and the keys leave as one coalesced store per wave.
constexpr int DK = 1024;            // k-mers per workgroup (4 per thread)
static_assert(DK == (int)SKF_DK, "k_sk_scatter names the first record of every block of DK k-mers");
__global__ void k_decode_block_starts(const u64* __restrict__ prefix, u32 n_recs, u32* __restrict__ blk_first)
{
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_recs) return;
  const u64 ko = prefix[i] >> 32, ke = prefix[i + 1] >> 32;
  const u64 B = (ko + DK - 1) / DK;
  if (B * DK < ke) blk_first[B] = i;      // (a record holds at most 60 k-mers: at most one block starts inside it)
}

__device__ __forceinline__ u64 load8u(const u8* p) { u64 w; __builtin_memcpy(&w, p, 8); return w; }

// DIRECT (kmx_count_reads_dev without the super-k-mer files): no record stream is written at all.  `recs` is then the batch's
// bases, 2 bits each, 32 to a 64-bit word, the first base in the top bits (k_pack_bases), and sbase[i] the position of record i's
// first base: k-mer j of the record is the 2k bits at base sbase[i] + j -- two (three) aligned word loads and a funnel shift.
template <int KW, int HASH, bool DIRECT>
__global__ __launch_bounds__(256)
void k_superk_decode_kmers(const u8* __restrict__ recs, const u64* __restrict__ prefix, const u32* __restrict__ blk_first, const u16* __restrict__ rec_part,
                           const u64* __restrict__ part_ids, u32 n_recs, u32 total, int k, u64 win, void* __restrict__ out, const u32* __restrict__ sbase,
                           const SkfCtl* __restrict__ ctl = nullptr /* set: the sizes are the device's (the grid covers a bound) */,
                           unsigned long long* __restrict__ strand = nullptr /* set (DIRECT): bit g = k-mer g is its own canonical form (forward < reverse complement) */)
{
  __shared__ u64 pk[DK + 1];
  __shared__ u32 nrec_s;
  const u32 tid = threadIdx.x;
  const u32 g0 = blockIdx.x * DK;
  if (ctl) { if (ctl->status) return; n_recs = ctl->nd; total = ctl->total; if (g0 >= total) return; }
  const u32 r0 = blk_first[blockIdx.x];
  const u32 avail = min((u32)DK + 1u, n_recs + 1u - r0);
  for (u32 t = tid; t < avail; t += 256) {
    const u64 e = prefix[r0 + t];
    pk[t] = DIRECT ? ((e & 0xFFFFFFFF00000000ULL) | (r0 + t < n_recs ? sbase[r0 + t] : 0u)) : e;      // (low word: the record's byte offset, or its first base)
  }
  if (tid == 0) nrec_s = avail;
  __syncthreads();
#pragma unroll
  for (int x = 0; x < DK / 256; x++) {
    const u32 g = g0 + tid + x * 256;
    if (g >= total) break;
    // the last entry t with hi(pk[t]) <= g  (entry 0 qualifies: the block's first k-mer lies in record r0)
    u32 lo = 0, hi = avail - 1;
    while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if ((u32)(pk[mid] >> 32) <= g) lo = mid; else hi = mid - 1; }
    const u64 pe = pk[lo];
    const u32 j = g - (u32)(pe >> 32);
    const u8* p = recs + (u32)pe + 1;                  // behind the record's length byte
    const u32 r = r0 + lo;
    const u64 part = HASH ? (part_ids ? part_ids[rec_part[r]] : (u64)rec_part[r]) : 0;      // (no table: the partition's index is its id)
    const u32 eb = (u32)k >> 2, es = ((u32)k & 3u) * 2;   // the following nucleotides start at digit k: byte eb, bit es
    if (DIRECT) {
      const u64* W = reinterpret_cast<const u64*>(recs);
      const u32 q = (u32)pe + j, w = q >> 5, o = (q & 31u) * 2u;
      if (KW == 1) {
        const u64 h = W[w], l = W[w + 1];
        const u64 fwd = (o ? (h << o) | (l >> (64 - o)) : h) >> (64 - 2 * k);
        const u64 rev = revcomp64(fwd, k);
        const u64 c = fwd < rev ? fwd : rev;
        reinterpret_cast<u64*>(out)[g] = HASH ? (xxh64_words(&c, 1) % win + win * part) : c;
        if (strand) { const u64 m = __ballot(fwd < rev); if ((tid & 63u) == 0) strand[g >> 6] = m; }      // (a wave's 64 k-mers are g .. g + 63, g a multiple of 64; lanes behind the batch's end are not in the ballot)
      } else {
        const u128 a = ((u128)W[w] << 64) | W[w + 1];
        const u64 l = W[w + 2];
        const u128 fwd = (o ? (a << o) | (u128)(l >> (64 - o)) : a) >> (128 - 2 * k);
        const u128 rev = revcomp128(fwd, k);
        const u128 c = fwd < rev ? fwd : rev;
        if (HASH) { u64 x[2] = {(u64)c, (u64)(c >> 64)}; reinterpret_cast<u64*>(out)[g] = xxh64_words(x, 2) % win + win * part; }
        else reinterpret_cast<u128*>(out)[g] = c;
        if (strand) { const u64 m = __ballot(fwd < rev); if ((tid & 63u) == 0) strand[g >> 6] = m; }
      }
    } else if (KW == 1) {
      const u64 mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
      u64 fwd = load8u(p) & mask;
      if (j && k < 32) {      // (super-k-mers of at most 28 k-mers: j <= 27, the 54 bits wanted lie in one shifted word)
        const u64 ex = load8u(p + eb) >> es;                                           // digits k, k + 1, ...
        fwd = ((fwd << (2 * j)) | (rev_digits64(ex) >> (64 - 2 * j))) & mask;
      } else if (j) {         // k = 32: a 64-bit key, but super-k-mers of up to 60 k-mers (the reference runs it as Kmer<64>): es = 0
        const u128 ex = ((u128)load8u(p + eb + 8) << 64) | load8u(p + eb);
        const u128 rv = ((u128)rev_digits64((u64)ex) << 64) | (u128)rev_digits64((u64)(ex >> 64));
        fwd = (u64)((((u128)fwd << (2 * j)) | (rv >> (128 - 2 * j))));                 // (mask is all ones)
      }
      const u64 rev = revcomp64(fwd, k);
      const u64 c = fwd < rev ? fwd : rev;
      reinterpret_cast<u64*>(out)[g] = HASH ? (xxh64_words(&c, 1) % win + win * part) : c;
    } else {
      const u128 mask = (k == 64) ? ~(u128)0 : ((((u128)1) << (2 * k)) - 1);
      u128 fwd = (((u128)load8u(p + 8) << 64) | load8u(p)) & mask;
      if (j) {
        const u64 a = load8u(p + eb), b2 = load8u(p + eb + 8);
        u128 ex = (((u128)b2 << 64) | a) >> es;
        if (es) ex |= (u128)p[eb + 16] << (128 - es);                                  // (j <= 59: 118 bits)
        const u128 rv = ((u128)rev_digits64((u64)ex) << 64) | (u128)rev_digits64((u64)(ex >> 64));   // the 64 digits reversed
        fwd = ((fwd << (2 * j)) | (rv >> (128 - 2 * j))) & mask;
      }
      const u128 rev = revcomp128(fwd, k);
      const u128 c = fwd < rev ? fwd : rev;
      if (HASH) { u64 w[2] = {(u64)c, (u64)(c >> 64)}; reinterpret_cast<u64*>(out)[g] = xxh64_words(w, 2) % win + win * part; }
      else reinterpret_cast<u128*>(out)[g] = c;
    }
  }
  (void)nrec_s;
}

// ---- k = 64 ... 95 (Kmer<96>) and 96 ... 127 (Kmer<128>), the reference's default KMER_LIST "32 64 96 128" (CMakeLists.txt:25-27,
//      kmer.hpp:164-630): keys of ceil(k / 32) words (kmer.hpp:215) -- two at k = 64, three up to 96, four beyond.  Correct first: a lane per k-mer as above, the record's integer S in nine registers'
//      worth of words, multi-word shifts; a super-k-mer holds up to 92 / 124 k-mers there (Sequence2SuperKmer.hpp:146: (bits - 8) / 2).
template <int NW> __device__ __forceinline__ void w_shr(const u64* a, u32 bits, u64* out)      // out = a >> bits (bits < 64 NW)
{
  const u32 ws = bits >> 6, bs = bits & 63u;
#pragma unroll
  for (int i = 0; i < NW; i++) {
    u64 lo = 0, hi = 0;
#pragma unroll
    for (int q = 0; q < NW; q++) { if ((u32)q == (u32)i + ws) lo = a[q]; if ((u32)q == (u32)i + ws + 1u) hi = a[q]; }
    out[i] = bs ? (lo >> bs) | (hi << (64u - bs)) : lo;
  }
}
template <int NW> __device__ __forceinline__ void w_shl(const u64* a, u32 bits, u64* out)      // out = a << bits (bits < 64 NW), NW words kept
{
  const u32 ws = bits >> 6, bs = bits & 63u;
#pragma unroll
  for (int i = 0; i < NW; i++) {
    u64 hi = 0, lo = 0;
#pragma unroll
    for (int q = 0; q < NW; q++) { if ((u32)q + ws == (u32)i) hi = a[q]; if ((u32)q + ws + 1u == (u32)i) lo = a[q]; }
    out[i] = bs ? (hi << bs) | (lo >> (64u - bs)) : hi;
  }
}
template <int KW, int HASH>
__global__ __launch_bounds__(256)
void k_superk_decode_wide(const u8* __restrict__ recs, const u64* __restrict__ prefix, const u32* __restrict__ blk_first, const u16* __restrict__ rec_part,
                          const u64* __restrict__ part_ids, u32 n_recs, u32 total, int k, u64 win, u64* __restrict__ out)
{
  __shared__ u64 pk[DK + 1];
  const u32 tid = threadIdx.x;
  const u32 g0 = blockIdx.x * DK;
  const u32 r0 = blk_first[blockIdx.x];
  const u32 avail = min((u32)DK + 1u, n_recs + 1u - r0);
  for (u32 t = tid; t < avail; t += 256) pk[t] = prefix[r0 + t];
  __syncthreads();
  for (int x = 0; x < DK / 256; x++) {
    const u32 g = g0 + tid + x * 256;
    if (g >= total) break;
    u32 lo = 0, hi = avail - 1;
    while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if ((u32)(pk[mid] >> 32) <= g) lo = mid; else hi = mid - 1; }
    const u64 pe = pk[lo];
    const u32 j = g - (u32)(pe >> 32);                 // my k-mer of the record
    const u8* p = recs + (u32)pe + 1;                  // behind the record's length byte
    u64 S[8];                                          // 2 (k + j) <= 500 bits of the record (the stream is readable 64 bytes past a record's start)
#pragma unroll
    for (int i = 0; i < 8; i++) S[i] = load8u(p + 8 * i);
    // digits j .. k-1: the seed's digits shifted up by j; digits 0 .. j-1: the j nucleotides that follow the seed (S's digits k, k + 1, ...) in reverse order
    u64 A[8], E[8], f[KW];
    w_shl<8>(S, 2u * j, A);
    w_shr<8>(S, 2u * (u32)k, E);
    u64 R[4] = {rev_digits64(E[3]), rev_digits64(E[2]), rev_digits64(E[1]), rev_digits64(E[0])}, B[4] = {0, 0, 0, 0};
    if (j) w_shr<4>(R, 256u - 2u * j, B);
    const u32 topbits = 2u * (u32)k - 64u * (KW - 1);   // 1 .. 64 bits of the top word
    const u64 topmask = topbits >= 64u ? ~0ULL : ((1ULL << topbits) - 1ULL);
#pragma unroll
    for (int i = 0; i < KW; i++) f[i] = A[i] | B[i];
    f[KW - 1] &= topmask;
    // reverse complement: the digits reversed over all KW words, digit ^ 2, shifted down to k digits
    u64 RC[KW], r[KW];
#pragma unroll
    for (int i = 0; i < KW; i++) RC[i] = rev_digits64(f[KW - 1 - i]) ^ 0xAAAAAAAAAAAAAAAAULL;
    w_shr<KW>(RC, 64u * KW - 2u * (u32)k, r);
    bool less = false, decided = false;
#pragma unroll
    for (int i = KW - 1; i >= 0; i--) if (!decided && f[i] != r[i]) { less = f[i] < r[i]; decided = true; }
    u64 c[KW];
#pragma unroll
    for (int i = 0; i < KW; i++) c[i] = less ? f[i] : r[i];
    if (HASH) {
      const u32 rr = r0 + lo;
      const u64 part = part_ids ? part_ids[rec_part[rr]] : (u64)rec_part[rr];
      out[g] = xxh64_words(c, KW) % win + win * part;
    } else {
#pragma unroll
      for (int i = 0; i < KW; i++) out[(u64)g * KW + i] = c[i];
    }
  }
}

// the sort of such keys: least significant word first, a stable 64-bit radix sort of (word, index) pairs per word, the partition last
__global__ void k_wide_iota(u32* __restrict__ perm, u32 n) { const u32 i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) perm[i] = i; }
__global__ void k_wide_gather_word(const u64* __restrict__ keys, const u32* __restrict__ perm, u32 n, int kw, int w, u64* __restrict__ out)
{ const u32 i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = keys[(u64)perm[i] * kw + w]; }
__global__ void k_wide_heads(const u64* __restrict__ keys, const u32* __restrict__ perm, const u16* __restrict__ spart, u32 n, int kw, u32* __restrict__ flag)
{
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 f = 1;
  if (i && spart[i] == spart[i - 1]) {
    const u64* a = keys + (u64)perm[i] * kw, *b = keys + (u64)perm[i - 1] * kw;
    f = 0;
    for (int w = 0; w < kw; w++) f |= a[w] != b[w] ? 1u : 0u;
  }
  flag[i] = f;
}
__global__ void k_wide_run_starts(const u32* __restrict__ flag, const u32* __restrict__ incl, u32 n, u32* __restrict__ start)
{
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flag[i]) start[incl[i] - 1] = i;
  if (i == n - 1) start[incl[i]] = n;
}
__global__ void k_wide_run_keep(const u32* __restrict__ start, u32 runs, u32 hard_min, u32* __restrict__ cnt, u32* __restrict__ keep)
{
  const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= runs) return;
  const u32 c = start[r + 1] - start[r];
  cnt[r] = c; keep[r] = c >= hard_min ? 1u : 0u;
}
__global__ void k_wide_emit(const u64* __restrict__ keys, const u32* __restrict__ perm, const u16* __restrict__ spart, const u32* __restrict__ start,
                            const u32* __restrict__ cnt, const u32* __restrict__ keep, const u32* __restrict__ pos, u32 runs, int kw,
                            u64* __restrict__ okeys, u32* __restrict__ ocnt, u16* __restrict__ opart)
{
  const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= runs || !keep[r]) return;
  const u32 o = pos[r], at = start[r];
  const u64* a = keys + (u64)perm[at] * kw;
  for (int w = 0; w < kw; w++) okeys[(u64)o * kw + w] = a[w];
  ocnt[o] = cnt[r]; opart[o] = spart[at];
}

// ASCII bases -> 2 bits each ((c >> 1) & 3: A 0, C 1, T 2, G 3), 32 to a word, the first one in the top bits; a thread per word.
// `n` bases, the array behind them readable for 16 more bytes; words [0, ceil(n / 32) + 2) are written (the decode reads up to two
// words past a k-mer's first one)
__global__ void k_pack_bases(const char* __restrict__ bases, u64 n, u64* __restrict__ out)
{
  const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x, nw = (n + 31) / 32 + 2;
  if (w >= nw) return;
  u64 v = 0;
  const u64 b0 = w * 32;
  if (b0 + 32 <= n) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      u64 x = load8u(reinterpret_cast<const u8*>(bases) + b0 + 8 * q);
      x = (x >> 1) & 0x0303030303030303ULL;                      // a code per byte, the first base in the low byte
      x = (x | (x >> 6)) & 0x000F000F000F000FULL;                // two codes per 16 bits (first one low) ...
      x = (x | (x >> 12)) & 0x000000FF000000FFULL;               // ... four per 32 ...
      x = (x | (x >> 24)) & 0xFFFFULL;                           // ... eight in 16 bits, first base in bits 0-1
      v |= x << (16 * q);                                        // 32 codes, first base lowest: reversed below
    }
    v = rev_digits64(v);
  } else {
    for (u32 i = 0; i < 32 && b0 + i < n; i++) v |= (u64)(((u32)(u8)bases[b0 + i] >> 1) & 3u) << (62 - 2 * i);
  }
  out[w] = v;
}

// the partition of every k-mer (the library sort's second key): only the fallback asks for it
__global__ void k_fill_kpart(const u32* __restrict__ part_kmer_off, u32 n_parts, u32 total, u16* __restrict__ out)
{
  const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  u32 lo = 0, hi = n_parts;                          // the last partition that starts at or before g
  while (lo + 1 < hi) { const u32 mid = (lo + hi) >> 1; if (part_kmer_off[mid] <= g) lo = mid; else hi = mid; }
  out[g] = (u16)lo;
}

}  // namespace kmx

using namespace kmx;

// abundance histogram of run counts (the library-sort paths; k_cs_sort adds its own runs): see kmx_ctx::d_hist
__global__ __launch_bounds__(256) void k_hist_runs(const u32* __restrict__ cnt, u32 n, unsigned long long* __restrict__ hist)
{
  __shared__ u32 hh[257];
  __shared__ unsigned long long big;
  for (u32 i = threadIdx.x; i < 257; i += 256) hh[i] = 0;
  if (threadIdx.x == 0) big = 0;
  __syncthreads();
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const u32 c = cnt[i];
    if (c <= 255u) atomicAdd(&hh[c], 1u); else { atomicAdd(&hh[256], 1u); atomicAdd(&big, (unsigned long long)c); }
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < 257; i += 256) if (hh[i]) atomicAdd(&hist[i], (unsigned long long)hh[i]);
  if (threadIdx.x == 0 && big) atomicAdd(&hist[257], big);
}
static void hist_runs(kmx_ctx* ctx, const u32* d_cnt, u32 runs)
{
  if (!ctx->hist_on || !runs) return;
  hipLaunchKernelGGL(k_hist_runs, dim3(std::min<u32>((runs + 255) / 256, 1024u)), dim3(256), 0, ctx->stream, d_cnt, runs, ctx->d_hist);
}

extern "C" int kmx_hist_reset(kmx_ctx* ctx)
{
  if (!ctx) return KMX_E_INVAL;
  KMX_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->d_hist) KMX_HIP(ctx, hipMalloc((void**)&ctx->d_hist, 258 * 8));
  KMX_HIP(ctx, hipMemsetAsync(ctx->d_hist, 0, 258 * 8, ctx->stream));
  ctx->hist_on = true;
  return KMX_OK;
}
extern "C" int kmx_hist_off(kmx_ctx* ctx) { if (!ctx) return KMX_E_INVAL; ctx->hist_on = false; return KMX_OK; }
extern "C" int kmx_hist_read(kmx_ctx* ctx, uint32_t lower, uint32_t upper, uint64_t* uniq_bins, uint64_t* total_bins, uint64_t* oob, uint64_t* sums)
{
  if (!ctx || !uniq_bins || !total_bins || !oob || !sums) return KMX_E_INVAL;
  if (!ctx->d_hist) return ctx->fail(KMX_E_INVAL, "kmx_hist_read: no histogram (kmx_hist_reset first)");
  if (upper > 255 || lower > upper) return ctx->fail(KMX_E_INVAL, "kmx_hist_read: bounds must satisfy lower <= upper <= 255");
  KMX_HIP(ctx, hipSetDevice(ctx->device));
  unsigned long long h[258];
  KMX_HIP(ctx, hipMemcpyAsync(h, ctx->d_hist, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (uint32_t i = 0; i <= upper - lower; i++) uniq_bins[i] = total_bins[i] = 0;
  oob[0] = oob[1] = oob[2] = oob[3] = 0; sums[0] = sums[1] = 0;
  for (uint32_t c = 0; c <= 255; c++) {      // KHist::inc (histogram.hpp:48-68) over h[c] keys of count c
    sums[0] += h[c]; sums[1] += (uint64_t)c * h[c];
    if (c < lower) { oob[0] += h[c]; oob[2] += (uint64_t)c * h[c]; }
    else if (c > upper) { oob[1] += h[c]; oob[3] += (uint64_t)c * h[c]; }
    else { uniq_bins[c - lower] = h[c]; total_bins[c - lower] = (uint64_t)c * h[c]; }
  }
  sums[0] += h[256]; sums[1] += h[257]; oob[1] += h[256]; oob[3] += h[257];
  return KMX_OK;
}

// one (sample, partition) stream: the batched path with one partition (rounds 1-2 had a library sort + run-length encode + select
// of their own here)
static int count_impl(kmx_ctx* ctx, const uint8_t* superk, uint64_t len, uint32_t k, int hash, uint64_t win, uint64_t part,
                      uint32_t hard_min, void** keys, uint32_t** counts, uint64_t* n_out)
{
  if (!ctx) return KMX_E_INVAL;
  if (!keys || !counts || !n_out || (len && !superk)) return ctx->fail(KMX_E_INVAL, "count: null argument");
  if (len >= 0xFFFFFF00ULL) return ctx->fail(KMX_E_UNSUPPORTED, "super-k-mer stream of 4 GiB or more: split it");
  *keys = nullptr; *counts = nullptr; *n_out = 0;
  uint64_t* kp = nullptr;
  const int rc = kmx_count_batch(ctx, 1, &superk, &len, k, hash, win, &part, hard_min, &kp, counts, n_out);
  *keys = kp;
  return rc;
}

extern "C" int kmx_count_kmer(kmx_ctx* ctx, const uint8_t* superk, uint64_t len, uint32_t kmer_size, uint32_t hard_min,
                              uint64_t** keys, uint32_t** counts, uint64_t* n_out)
{
  return count_impl(ctx, superk, len, kmer_size, 0, 0, 0, hard_min, (void**)keys, counts, n_out);
}

extern "C" int kmx_count_hash(kmx_ctx* ctx, const uint8_t* superk, uint64_t len, uint32_t kmer_size, uint64_t window,
                              uint64_t partition, uint32_t hard_min, uint64_t** hashes, uint32_t** counts, uint64_t* n_out)
{
  return count_impl(ctx, superk, len, kmer_size, 1, window, partition, hard_min, (void**)hashes, counts, n_out);
}


// ---- batched count: every partition stream of one sample in one call ---------------------------------
__global__ void k_run_part_flags(const u32* __restrict__ run_start, const u32* __restrict__ run_cnt, u32 n_runs,
                                 const u16* __restrict__ sorted_part, u32 hard_min, u16* __restrict__ run_part, u8* __restrict__ flags)
{
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_runs) return;
  run_part[i] = sorted_part[run_start[i]];      // equal keys always come from one partition
  flags[i] = run_cnt[i] >= hard_min ? 1 : 0;
}

template <typename KeyT>
__global__ void k_gather_runs(const u32* __restrict__ idx, u32 n, const KeyT* __restrict__ uniq, const u32* __restrict__ cnt,
                              KeyT* __restrict__ out_k, u32* __restrict__ out_c)
{
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 j = idx[i];
  out_k[i] = uniq[j];
  out_c[i] = cnt[j];
}

__global__ void k_gather_u16(const u32* __restrict__ idx, u32 n, const u16* __restrict__ in, u16* __restrict__ out)
{
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[idx[i]];
}

// first index of every partition in the partition-sorted run list (n_parts + 1 entries)
__global__ void k_part_bounds(const u16* __restrict__ part_sorted, u32 n, u32 n_parts, u32* __restrict__ bounds)
{
  const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > n_parts) return;
  u32 lo = 0, hi = n;                 // first i with part_sorted[i] >= p
  while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (part_sorted[mid] < p) lo = mid + 1; else hi = mid; }
  bounds[p] = lo;
}

// ---- results left on the device (kmx_count_reads_dev): the kept (key, count) pairs -- two arrays, partition after partition --
//      become packed records, the body of a .kmer file with 4-byte counts (io/kmer_file.hpp:102-108), in the stores of the GPUs
//      that will merge them: partition p -> stores[p % G], the partitions of one store back to back (one copy per GPU) ----
struct CountOut {
  uint64_t** keys = nullptr; uint32_t** counts = nullptr; uint64_t* n_out = nullptr;          // host arrays, or
  kmx_store* const* stores = nullptr; u32 n_stores = 0; kmx_list* lists = nullptr;           // device stores
  u32 inner = 0;           // several samples in one call: partition p' = sample * inner + p, the store is p's (0: p' is the partition)
  u32 store_of(u32 p) const { return (inner ? p % inner : p) % n_stores; }
  bool dev() const { return lists != nullptr; }
};

// (WideKey<3|4> -- keys of three and four words, k = 65 ... 127, low word first -- and key_dword: count_sort.hpp)

template <typename KeyT>
__global__ __launch_bounds__(256)
void k_pack_recs(const KeyT* __restrict__ keys, const u32* __restrict__ cnt, u32 n, const u32* __restrict__ bounds, u32 n_parts,
                 const u64* __restrict__ pdst, u8* __restrict__ out)
{
  constexpr u32 KWD = sizeof(KeyT) / 4;      // key dwords
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 lo = 0, hi = n_parts;                  // the last partition that starts at or before i (empty ones in front of it share its start)
  while (lo + 1 < hi) { const u32 mid = (lo + hi) >> 1; if (bounds[mid] <= i) lo = mid; else hi = mid; }
  u32* o = reinterpret_cast<u32*>(out + (pdst[lo] + (i - bounds[lo])) * (u64)(sizeof(KeyT) + 4));
  const KeyT k = keys[i];
#pragma unroll
  for (u32 w = 0; w < KWD; w++) o[w] = key_dword<KeyT>(k, w);
  o[KWD] = cnt[i];
}

template <typename KeyT>
static int pack_to_stores(kmx_ctx* ctx, const KeyT* d_k, const u32* d_c, const std::vector<u32>& bounds /* n_parts + 1 */, u32 n_parts, const CountOut& out)
{
  const u32 kept = bounds[n_parts], G = out.n_stores;
  const size_t RB = sizeof(KeyT) + 4;
  for (u32 p = 0; p < n_parts; p++) { out.lists[p].recs = nullptr; out.lists[p].n = bounds[p + 1] - bounds[p]; }
  if (!kept) return KMX_OK;
  std::vector<u64> pdst(n_parts), doff((size_t)G + 1, 0);
  { u64 at = 0; for (u32 d = 0; d < G; d++) { doff[d] = at; for (u32 p = 0; p < n_parts; p++) if (out.store_of(p) == d) { pdst[p] = at; at += bounds[p + 1] - bounds[p]; } } doff[G] = at; }
  hipStream_t st = ctx->stream;
  const bool direct = G == 1 && out.stores[0]->device == ctx->device;      // one GPU: packed straight into the store
  u8* d_pack = direct ? (u8*)out.stores[0]->alloc((size_t)kept * RB) : (u8*)ctx->dalloc((size_t)kept * RB);
  u32* d_bounds = (u32*)ctx->dalloc(((size_t)n_parts + 1) * 4);
  u64* d_pdst = (u64*)ctx->dalloc((size_t)n_parts * 8);
  auto release = [&]() { if (!direct) ctx->dfree(d_pack); ctx->dfree(d_bounds); ctx->dfree(d_pdst); };
  if (!d_pack || !d_bounds || !d_pdst) { release(); return ctx->fail(KMX_E_NOMEM, direct && !d_pack ? "count store is full" : "count: device allocation failed"); }
  hipError_t e;
  if ((e = hipMemcpyAsync(d_bounds, bounds.data(), ((size_t)n_parts + 1) * 4, hipMemcpyHostToDevice, st)) != hipSuccess ||
      (e = hipMemcpyAsync(d_pdst, pdst.data(), (size_t)n_parts * 8, hipMemcpyHostToDevice, st)) != hipSuccess) { release(); return ctx->fail(KMX_E_HIP, std::string("count pack upload: ") + hipGetErrorString(e)); }
  hipLaunchKernelGGL((k_pack_recs<KeyT>), dim3((kept + 255) / 256), dim3(256), 0, st, d_k, d_c, kept, d_bounds, n_parts, d_pdst, d_pack);
  if ((e = hipGetLastError()) != hipSuccess) { release(); return ctx->fail(KMX_E_HIP, std::string("k_pack_recs: ") + hipGetErrorString(e)); }
  int rc = KMX_OK;
  for (u32 d = 0; d < G && rc == KMX_OK; d++) {
    const size_t nb = (size_t)(doff[d + 1] - doff[d]) * RB;
    if (!nb) continue;
    u8* dst = d_pack;
    if (!direct) {
      kmx_store* S = out.stores[d];
      dst = (u8*)S->alloc(nb);
      if (!dst) { rc = ctx->fail(KMX_E_NOMEM, "count store is full"); break; }
      // the store of another GPU is filled over xGMI: peer access is enabled for the pair the first time it is used
      // (kmx_peer_path; hipMemcpyPeerAsync stages through the host when the two GPUs have none)
      if (S->device != ctx->device) (void)kmx_peer_path(ctx->device, S->device);
      // (KMX_TRACE=1: a line per copy -- one per destination store and count call is the contract: partitions bound for one GPU are contiguous)
      if (list_copy_trace()) fprintf(stderr, "[kmx copy] lists from device %d to store %u on device %d: %zu bytes, %s\n", ctx->device, d, S->device, nb, S->device == ctx->device ? "device copy" : "hipMemcpyPeerAsync");
      e = S->device == ctx->device ? hipMemcpyAsync(dst, d_pack + doff[d] * RB, nb, hipMemcpyDeviceToDevice, st)
                                   : hipMemcpyPeerAsync(dst, S->device, d_pack + doff[d] * RB, ctx->device, nb, st);
      if (e != hipSuccess) { rc = ctx->fail(KMX_E_HIP, std::string("count list copy: ") + hipGetErrorString(e)); break; }
    }
    for (u32 p = 0; p < n_parts; p++) if (out.store_of(p) == d && out.lists[p].n) out.lists[p].recs = dst + (pdst[p] - doff[d]) * RB;
  }
  if ((e = hipStreamSynchronize(st)) != hipSuccess && rc == KMX_OK) rc = ctx->fail(KMX_E_HIP, std::string("count pack: ") + hipGetErrorString(e));
  release();
  return rc;
}

// the same from the partition-local count: the kept pairs of every bucket still lie in the bucket's own place (d_tk / d_tc at
// boff[b]); one kernel writes them as records where they belong -- straight into the store when one GPU holds everything
template <typename KeyT>
static int compact_to_stores(kmx_ctx* ctx, const KeyT* d_tk, const u32* d_tc, const u32* d_boff, const u32* d_koff, const u32* koff /* host, TB + 1 */,
                             const std::vector<CsPart>& parts, u32 TB, const CountOut& out)
{
  const u32 n_parts = (u32)parts.size(), kept = koff[TB], G = out.n_stores;
  const size_t RB = sizeof(KeyT) + 4;
  auto plen = [&](u32 p) { return koff[parts[p].bucket0 + parts[p].nb] - koff[parts[p].bucket0]; };
  for (u32 p = 0; p < n_parts; p++) { out.lists[p].recs = nullptr; out.lists[p].n = plen(p); }
  if (!kept) return KMX_OK;
  std::vector<u64> pdst(n_parts), doff((size_t)G + 1, 0);
  { u64 at = 0; for (u32 d = 0; d < G; d++) { doff[d] = at; for (u32 p = 0; p < n_parts; p++) if (out.store_of(p) == d) { pdst[p] = at; at += plen(p); } } doff[G] = at; }
  hipStream_t st = ctx->stream;
  const bool direct = G == 1 && out.stores[0]->device == ctx->device;      // one GPU: packed straight into the store
  u8* d_pack = direct ? (u8*)out.stores[0]->alloc((size_t)kept * RB) : (u8*)ctx->dalloc((size_t)kept * RB);
  // where bucket b's records go: with one store the partitions lie in their own order (= the kept offsets, already on the device)
  u32* d_bdst = G > 1 ? (u32*)ctx->dalloc((size_t)TB * 4) : nullptr;
  u32* h_bdst = G > 1 ? (u32*)ctx->halloc((size_t)TB * 4) : nullptr;
  auto release = [&]() { if (!direct) ctx->dfree(d_pack); ctx->dfree(d_bdst); ctx->hfree(h_bdst); };
  if (!d_pack || (G > 1 && (!d_bdst || !h_bdst))) { release(); return ctx->fail(KMX_E_NOMEM, direct && !d_pack ? "count store is full" : "count: device allocation failed"); }
  hipError_t e;
  if (G > 1) {
    for (u32 p = 0; p < n_parts; p++) for (u32 b = parts[p].bucket0; b < parts[p].bucket0 + parts[p].nb; b++) h_bdst[b] = (u32)(pdst[p] + (koff[b] - koff[parts[p].bucket0]));
    if ((e = hipMemcpyAsync(d_bdst, h_bdst, (size_t)TB * 4, hipMemcpyHostToDevice, st)) != hipSuccess) { release(); return ctx->fail(KMX_E_HIP, std::string("count pack upload: ") + hipGetErrorString(e)); }
  }
  hipLaunchKernelGGL((k_cs_compact_recs<KeyT>), dim3(TB), dim3(CS_TPB), 0, st, d_tk, d_tc, d_boff, d_koff, G > 1 ? (const u32*)d_bdst : d_koff, d_pack);
  if ((e = hipGetLastError()) != hipSuccess) { release(); return ctx->fail(KMX_E_HIP, std::string("k_cs_compact_recs: ") + hipGetErrorString(e)); }
  int rc = KMX_OK;
  for (u32 d = 0; d < G && rc == KMX_OK; d++) {
    const size_t nb = (size_t)(doff[d + 1] - doff[d]) * RB;
    if (!nb) continue;
    u8* dst = d_pack;
    if (!direct) {
      kmx_store* S = out.stores[d];
      dst = (u8*)S->alloc(nb);
      if (!dst) { rc = ctx->fail(KMX_E_NOMEM, "count store is full"); break; }
      // the store of another GPU is filled over xGMI: peer access is enabled for the pair the first time it is used
      // (kmx_peer_path; hipMemcpyPeerAsync stages through the host when the two GPUs have none)
      if (S->device != ctx->device) (void)kmx_peer_path(ctx->device, S->device);
      // (KMX_TRACE=1: a line per copy -- one per destination store and count call is the contract: partitions bound for one GPU are contiguous)
      if (list_copy_trace()) fprintf(stderr, "[kmx copy] lists from device %d to store %u on device %d: %zu bytes, %s\n", ctx->device, d, S->device, nb, S->device == ctx->device ? "device copy" : "hipMemcpyPeerAsync");
      e = S->device == ctx->device ? hipMemcpyAsync(dst, d_pack + doff[d] * RB, nb, hipMemcpyDeviceToDevice, st)
                                   : hipMemcpyPeerAsync(dst, S->device, d_pack + doff[d] * RB, ctx->device, nb, st);
      if (e != hipSuccess) { rc = ctx->fail(KMX_E_HIP, std::string("count list copy: ") + hipGetErrorString(e)); break; }
    }
    for (u32 p = 0; p < n_parts; p++) if (out.store_of(p) == d && out.lists[p].n) out.lists[p].recs = dst + (pdst[p] - doff[d]) * RB;
  }
  if ((e = hipStreamSynchronize(st)) != hipSuccess && rc == KMX_OK) rc = ctx->fail(KMX_E_HIP, std::string("count pack: ") + hipGetErrorString(e));
  release();
  return rc;
}

// ---- partition-local sample sort + run-length count (count_sort.hpp): keys grouped by partition in d_keys, partition p =
//      keys [kmoff[p], kmoff[p + 1]).  Returns KMX_OK, a negative error, or 1 when a bucket would not fit the LDS (the caller
//      then uses the library sort: d_keys is still untouched at that point). ----
// the bucket kernels: a wave per bucket with the keys in registers (k_cs_wave_sort, two launches for the two size classes); the LDS
// kernels of before -- hash first (64-bit keys) or bitonic sort per workgroup -- stay behind KMX_COUNT_BUCKETS=hash|sort for comparison
template <typename KeyT> static void cs_launch_bucket_count(u32 TB, hipStream_t st, const KeyT* bkeys, const u32* boff, u32 hard_min, KeyT* tk, u32* tc, u32* nkept,
                                                            unsigned long long* hist, u32* overflow);
static int cs_bucket_kernel()      // 0: a wave per bucket, keys in registers (k_cs_wave_sort), the LDS kernels for the buckets beyond it; 1: hash first in LDS (64-bit keys); 2: bitonic sort in LDS
{
  const char* e = getenv("KMX_COUNT_BUCKETS");      // (read per call: the tests switch it)
  return !e ? 0 : !strcmp(e, "hash") ? 1 : !strcmp(e, "sort") ? 2 : 0;
}
template <> void cs_launch_bucket_count<u64>(u32 TB, hipStream_t st, const u64* bkeys, const u32* boff, u32 hard_min, u64* tk, u32* tc, u32* nkept,
                                             unsigned long long* hist, u32* overflow)
{
  const int which = cs_bucket_kernel();
  const u32 lo = which == 0 ? CS_WAVE_MAX : 0u;
  if (which == 0) hipLaunchKernelGGL((k_cs_wave_sort<u64, 8, 16>), dim3((TB + CS_WAVES - 1) / CS_WAVES), dim3(64 * CS_WAVES), 0, st, bkeys, boff, TB, 0u, (u32)CsCap<u64>::cap,
                                     hard_min, tk, tc, nkept, hist, overflow);
  if (which == 2) hipLaunchKernelGGL((k_cs_sort<u64>), dim3(TB), dim3(CS_TPB), 0, st, bkeys, boff, hard_min, tk, tc, nkept, hist, overflow, lo);
  else hipLaunchKernelGGL(k_cs_count_hash, dim3(TB), dim3(CS_TPB), 0, st, bkeys, boff, hard_min, tk, tc, nkept, hist, overflow, lo);
}
template <> void cs_launch_bucket_count<__uint128_t>(u32 TB, hipStream_t st, const __uint128_t* bkeys, const u32* boff, u32 hard_min, __uint128_t* tk, u32* tc, u32* nkept,
                                                     unsigned long long* hist, u32* overflow)
{
  const int which = cs_bucket_kernel();
  const u32 lo = which == 0 ? CS_WAVE_MAX : 0u;
  if (which == 0) hipLaunchKernelGGL((k_cs_wave_sort<__uint128_t, 8, 16>), dim3((TB + CS_WAVES - 1) / CS_WAVES), dim3(64 * CS_WAVES), 0, st, bkeys, boff, TB, 0u,
                                     (u32)CsCap<__uint128_t>::cap, hard_min, tk, tc, nkept, hist, overflow);
  hipLaunchKernelGGL((k_cs_sort<__uint128_t, 2048>), dim3(TB), dim3(CS_TPB), 0, st, bkeys, boff, hard_min, tk, tc, nkept, hist, overflow, lo, 0u);
  hipLaunchKernelGGL((k_cs_sort<__uint128_t, 4096>), dim3(TB), dim3(CS_TPB), 0, st, bkeys, boff, hard_min, tk, tc, nkept, hist, overflow, 2048u, 1u);
}
// (three- and four-word keys: up to 1024 keys a wave as well -- 96 / 128 registers of keys at 16 per lane, one wave per SIMD of a 256-thread
//  workgroup has 512 --, the LDS sort for up to 2048 behind it; a larger bucket sends the call to wide_sort_count)
template <int KW> static void cs_launch_bucket_count_wide(u32 TB, hipStream_t st, const WideKey<KW>* bkeys, const u32* boff, u32 hard_min, WideKey<KW>* tk, u32* tc, u32* nkept,
                                                          unsigned long long* hist, u32* overflow)
{
  typedef WideKey<KW> K;
  hipLaunchKernelGGL((k_cs_wave_sort<K, 8, 16>), dim3((TB + CS_WAVES - 1) / CS_WAVES), dim3(64 * CS_WAVES), 0, st, bkeys, boff, TB, 0u, (u32)CsCap<K>::cap,
                     hard_min, tk, tc, nkept, hist, overflow);
  hipLaunchKernelGGL((k_cs_sort<K, CsCap<K>::cap>), dim3(TB), dim3(CS_TPB), 0, st, bkeys, boff, hard_min, tk, tc, nkept, hist, overflow, (u32)CS_WAVE_MAX, 1u);
}
template <> void cs_launch_bucket_count<WideKey<3>>(u32 TB, hipStream_t st, const WideKey<3>* bkeys, const u32* boff, u32 hard_min, WideKey<3>* tk, u32* tc, u32* nkept,
                                                    unsigned long long* hist, u32* overflow) { cs_launch_bucket_count_wide<3>(TB, st, bkeys, boff, hard_min, tk, tc, nkept, hist, overflow); }
template <> void cs_launch_bucket_count<WideKey<4>>(u32 TB, hipStream_t st, const WideKey<4>* bkeys, const u32* boff, u32 hard_min, WideKey<4>* tk, u32* tc, u32* nkept,
                                                    unsigned long long* hist, u32* overflow) { cs_launch_bucket_count_wide<4>(TB, st, bkeys, boff, hard_min, tk, tc, nkept, hist, overflow); }
__global__ void k_hist_add(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst)
{
  if (threadIdx.x < 258 && src[threadIdx.x]) atomicAdd(&dst[threadIdx.x], src[threadIdx.x]);
}

template <typename KeyT>
static int partition_sort_count(kmx_ctx* ctx, StageClock& clk, KeyT* d_keys, const std::vector<u64>& kmoff, u32 n_parts, u32 hard_min,
                                const CountOut& out)
{
  uint64_t** keys = out.keys; uint32_t** counts = out.counts; uint64_t* n_out = out.n_out;
  const char* force = getenv("KMX_COUNT_SORT");
  if (force && !strcmp(force, "library")) return 1;
  const u64 total = kmoff[n_parts];
  const u32 target = cs_target<KeyT>(), cap = (u32)CsCap<KeyT>::cap;
  std::vector<CsPart> parts(n_parts); std::vector<CsChunk> chunks;
  u32 TB = 0;
  for (u32 p = 0; p < n_parts; p++) {
    const u64 n = kmoff[p + 1] - kmoff[p];
    const u64 nb = std::max<u64>(1, (n + target - 1) / target);
    if (nb > (u64)CS_MAXB || nb * 4 > (u64)CsCap<KeyT>::sample) return 1;      // a partition beyond the bucket / sample limits: the library sort takes the batch
    parts[p] = CsPart{(u32)kmoff[p], (u32)n, TB, (u32)nb};
    TB += (u32)nb;
    for (u64 o = 0; o < n; o += cs_chunk<KeyT>()) chunks.push_back(CsChunk{p, (u32)(kmoff[p] + o), (u32)std::min<u64>(cs_chunk<KeyT>(), n - o), 0});
  }
  hipStream_t st = ctx->stream;
  std::vector<void*> blocks;
  auto dal = [&](size_t b) { void* p = ctx->dalloc(b); blocks.push_back(p); return p; };
  auto release = [&]() { for (void* b : blocks) ctx->dfree(b); };
  // the two tables go up in one copy, out of one page-locked block that also takes the kept sizes on their way back
  static_assert(sizeof(CsPart) == 16 && sizeof(CsChunk) == 16, "the tables share one block");
  const size_t tab_bytes = sizeof(CsPart) * n_parts + sizeof(CsChunk) * chunks.size();
  u8* h_tab = (u8*)ctx->halloc(tab_bytes + 4 * ((size_t)TB + 2));
  struct HRel { kmx_ctx* c; void* p; ~HRel() { c->hfree(p); } } h_tab_rel{ctx, h_tab};
  if (!h_tab) return ctx->fail(KMX_E_NOMEM, "count sort: host staging allocation failed");
  memcpy(h_tab, parts.data(), sizeof(CsPart) * n_parts);
  if (!chunks.empty()) memcpy(h_tab + sizeof(CsPart) * n_parts, chunks.data(), sizeof(CsChunk) * chunks.size());
  u8* d_tab = (u8*)dal(tab_bytes + 16);
  CsPart* d_parts = (CsPart*)d_tab;
  CsChunk* d_chunks = (CsChunk*)(d_tab + sizeof(CsPart) * n_parts);
  typedef typename CsSpl<KeyT>::type SplT;      // (wide keys: their two most significant words)
  SplT* d_spl = (SplT*)dal(sizeof(SplT) * (size_t)TB);
  u32* d_cnt = (u32*)dal(4 * ((size_t)TB + 1)), *d_boff = (u32*)dal(4 * ((size_t)TB + 1)), *d_cur = (u32*)dal(4 * ((size_t)TB + 1));
  u32* d_nkept = (u32*)dal(4 * ((size_t)TB + 1)), *d_koff = (u32*)dal(4 * ((size_t)TB + 2));
  KeyT* d_bkeys = (KeyT*)dal(sizeof(KeyT) * total);
  u32* d_tc = (u32*)dal(4 * total);
  for (void* b : blocks) if (!b) { release(); return ctx->fail(KMX_E_NOMEM, "count sort: device allocation failed"); }
  auto fail = [&](hipError_t e, const char* what) { release(); return ctx->fail(KMX_E_HIP, std::string(what) + ": " + hipGetErrorString(e)); };
  hipError_t e;
  if ((e = hipMemcpyAsync(d_tab, h_tab, tab_bytes, hipMemcpyHostToDevice, st)) != hipSuccess ||
      (e = hipMemsetAsync(d_cnt, 0, 4 * ((size_t)TB + 1), st)) != hipSuccess) return fail(e, "count sort upload");      // (d_cnt[TB]: the overflow word, cleared with the rest)
  (void)cap;
  hipLaunchKernelGGL((k_cs_splitters<KeyT>), dim3(n_parts), dim3(CS_SPL_TPB), 0, st, d_keys, d_parts, d_spl);
  hipLaunchKernelGGL((k_cs_walk<KeyT, false>), dim3((unsigned)chunks.size()), dim3(CS_WALK_TPB), 0, st, d_keys, d_parts, d_chunks, d_spl, d_cnt, (KeyT*)nullptr);
  hipLaunchKernelGGL(k_cs_scan, dim3(1), dim3(1024), 0, st, d_cnt, TB, d_boff, d_cur, (const u32*)nullptr);      // (d_cur: the scatter's cursors)
  hipLaunchKernelGGL((k_cs_walk<KeyT, true>), dim3((unsigned)chunks.size()), dim3(CS_WALK_TPB), 0, st, d_keys, d_parts, d_chunks, d_spl, d_cur, d_bkeys);
  // (a bucket beyond what the count kernel takes is found by the kernel itself and reported with the kept sizes: one round trip to
  //  the host per call, not two.  The grouped keys stay as they are until then -- the library sort needs them -- and the abundance
  //  histogram of this call is kept apart until the call is known to stand.)
  KeyT* d_tk = (KeyT*)dal(sizeof(KeyT) * total);
  u32* d_flag = d_cnt + TB;                // (d_cnt has TB + 1 entries; the last one is free behind the scan: the overflow word)
  unsigned long long* d_htmp = ctx->hist_on ? (unsigned long long*)dal(258 * 8) : nullptr;
  if (!d_tk || (ctx->hist_on && !d_htmp)) { release(); return ctx->fail(KMX_E_NOMEM, "count sort: device allocation failed"); }
  if (d_htmp && (e = hipMemsetAsync(d_htmp, 0, 258 * 8, st)) != hipSuccess) return fail(e, "count sort clear");
  cs_launch_bucket_count<KeyT>(TB, st, d_bkeys, d_boff, hard_min, d_tk, d_tc, d_nkept, d_htmp, d_flag);
  hipLaunchKernelGGL(k_cs_scan, dim3(1), dim3(1024), 0, st, d_nkept, TB, d_koff, (u32*)nullptr, (const u32*)d_flag);      // (d_koff[TB + 1] = the overflow word)
  const u32* koff = reinterpret_cast<const u32*>(h_tab + tab_bytes);
  if ((e = hipMemcpyAsync(h_tab + tab_bytes, d_koff, 4 * ((size_t)TB + 2), hipMemcpyDeviceToHost, st)) != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "count sort kept");
  if (koff[TB + 1]) { release(); return 1; }      // (d_keys is untouched: the caller takes the library sort)
  if (d_htmp) hipLaunchKernelGGL(k_hist_add, dim3(1), dim3(258), 0, st, d_htmp, ctx->d_hist);
  clk.mark("buckets+sort+count");
  const u32 kept = koff[TB];
  if (out.dev()) {      // the pairs stay on the device, packed as records in the stores of the GPUs that merge them
    const int rc = compact_to_stores<KeyT>(ctx, d_tk, d_tc, d_boff, d_koff, koff, parts, TB, out);
    release();
    clk.mark("pack");
    return rc;
  }
  KeyT* d_ok = d_bkeys;                    // (and the buckets are dead behind the sort)
  u32* d_oc = (u32*)dal(4 * (size_t)std::max<u32>(kept, 1));
  KeyT* h_k = kept ? (KeyT*)ctx->halloc((size_t)kept * sizeof(KeyT)) : nullptr;
  u32* h_c = kept ? (u32*)ctx->halloc((size_t)kept * 4) : nullptr;
  auto hrel = [&]() { ctx->hfree(h_k); ctx->hfree(h_c); };
  if (!d_oc || (kept && (!h_k || !h_c))) { hrel(); release(); return ctx->fail(KMX_E_NOMEM, "count sort: allocation failed"); }
  if (kept) {
    hipLaunchKernelGGL((k_cs_compact<KeyT>), dim3(TB), dim3(CS_TPB), 0, st, d_tk, d_tc, d_boff, d_koff, d_ok, d_oc);
    if ((e = hipMemcpyAsync(h_k, d_ok, (size_t)kept * sizeof(KeyT), hipMemcpyDeviceToHost, st)) != hipSuccess ||
        (e = hipMemcpyAsync(h_c, d_oc, (size_t)kept * 4, hipMemcpyDeviceToHost, st)) != hipSuccess ||
        (e = hipStreamSynchronize(st)) != hipSuccess) { hrel(); return fail(e, "count sort download"); }
  }
  std::atomic<u32> next{0}; std::atomic<int> oom{0};
  auto fill = [&]() {
    for (u32 p; (p = next++) < n_parts;) {
      const u32 lo = koff[parts[p].bucket0], hi = koff[parts[p].bucket0 + parts[p].nb];
      const size_t n = hi - lo;
      keys[p] = (uint64_t*)malloc(n ? n * sizeof(KeyT) : 8);
      counts[p] = (uint32_t*)malloc(n ? n * 4 : 4);
      if (!keys[p] || !counts[p]) { oom = 1; continue; }
      n_out[p] = n;
      if (n) { memcpy(keys[p], h_k + lo, n * sizeof(KeyT)); memcpy(counts[p], h_c + lo, n * 4); }
    }
  };
  {
    const unsigned nthr = std::max(1u, std::min({16u, std::thread::hardware_concurrency(), n_parts}));
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nthr; t++) th.emplace_back(fill);
    fill();
    for (auto& x : th) x.join();
  }
  hrel(); release();
  if (oom) return ctx->fail(KMX_E_NOMEM, "count sort: host allocation failed");
  clk.mark("download");
  return KMX_OK;
}

// ---- batched count: every partition stream of one sample in one call ---------------------------------
// One global radix sort of (key, partition) pairs, one run-length pass, then only the kept runs are
// regrouped by partition (stable, so each partition's keys stay ascending).
template <typename KeyT>
static int batch_sort_rle(kmx_ctx* ctx, StageClock& clk, KeyT* d_keys, u16* d_kpart, u32 total, u32 n_parts, unsigned key_bits, u32 hard_min,
                          const CountOut& out)
{
  uint64_t** keys = out.keys; uint32_t** counts = out.counts; uint64_t* n_out = out.n_out;
  hipStream_t st = ctx->stream;
  KeyT* d_sorted = (KeyT*)ctx->dalloc((size_t)total * sizeof(KeyT));
  u16* d_spart = (u16*)ctx->dalloc((size_t)total * 2);
  KeyT* d_uniq = (KeyT*)ctx->dalloc((size_t)total * sizeof(KeyT));
  u32* d_cnt = (u32*)ctx->dalloc((size_t)total * 4);
  u32* d_runs = (u32*)ctx->dalloc(256);
  std::vector<void*> blocks = {d_sorted, d_spart, d_uniq, d_cnt, d_runs};
  auto release = [&]() { for (void* b : blocks) ctx->dfree(b); };
  for (void* b : blocks) if (!b) { release(); return ctx->fail(KMX_E_NOMEM, "count batch: device allocation failed"); }
  auto fail = [&](hipError_t e, const char* what) { release(); return ctx->fail(KMX_E_HIP, std::string(what) + ": " + hipGetErrorString(e)); };
  size_t t1 = 0, t2 = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, t1, d_keys, d_sorted, d_kpart, d_spart, (size_t)total, 0, key_bits, st);
  if (e == hipSuccess) e = rocprim::run_length_encode(nullptr, t2, d_sorted, total, d_uniq, d_cnt, d_runs, st);
  if (e != hipSuccess) return fail(e, "rocPRIM temp sizing");
  size_t tmax = std::max(t1, t2) + 256;
  void* d_tmp = ctx->dalloc(tmax); blocks.push_back(d_tmp);
  if (!d_tmp) { release(); return ctx->fail(KMX_E_NOMEM, "count batch: device allocation failed"); }
  size_t t = tmax;
  if ((e = rocprim::radix_sort_pairs(d_tmp, t, d_keys, d_sorted, d_kpart, d_spart, (size_t)total, 0, key_bits, st)) != hipSuccess) return fail(e, "radix_sort_pairs");
  clk.mark("sort");
  t = tmax;
  if ((e = rocprim::run_length_encode(d_tmp, t, d_sorted, total, d_uniq, d_cnt, d_runs, st)) != hipSuccess) return fail(e, "run_length_encode");
  u32 runs = 0;
  if ((e = hipMemcpyAsync(&runs, d_runs, 4, hipMemcpyDeviceToHost, st)) != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "sync");
  clk.mark("rle");
  hist_runs(ctx, d_cnt, runs);
  // per run: partition + keep flag; kept run indices regrouped by partition
  u32* d_start = (u32*)ctx->dalloc(((size_t)runs + 1) * 4);
  u16* d_rpart = (u16*)ctx->dalloc((size_t)runs * 2 + 2), *d_kp = (u16*)ctx->dalloc((size_t)runs * 2 + 2), *d_kp2 = (u16*)ctx->dalloc((size_t)runs * 2 + 2);
  u8* d_flags = (u8*)ctx->dalloc((size_t)runs + 1);
  u32* d_idx = (u32*)ctx->dalloc(((size_t)runs + 1) * 4), *d_idx2 = (u32*)ctx->dalloc(((size_t)runs + 1) * 4);
  u32* d_bounds = (u32*)ctx->dalloc(((size_t)n_parts + 1) * 4);
  for (void* b : {(void*)d_start, (void*)d_rpart, (void*)d_kp, (void*)d_kp2, (void*)d_flags, (void*)d_idx, (void*)d_idx2, (void*)d_bounds}) blocks.push_back(b);
  for (void* b : blocks) if (!b) { release(); return ctx->fail(KMX_E_NOMEM, "count batch: device allocation failed"); }
  unsigned pbits = 1; while ((1u << pbits) < n_parts) pbits++;
  size_t t3 = 0, t4 = 0, t5 = 0;
  e = rocprim::exclusive_scan(nullptr, t3, d_cnt, d_start, 0u, (size_t)runs, rocprim::plus<u32>(), st);
  if (e == hipSuccess) e = rocprim::select(nullptr, t4, rocprim::counting_iterator<u32>(0), d_flags, d_idx, d_runs, (size_t)runs, st);
  if (e == hipSuccess) e = rocprim::radix_sort_pairs(nullptr, t5, d_kp, d_kp2, d_idx, d_idx2, (size_t)runs, 0, pbits, st);
  if (e != hipSuccess) return fail(e, "rocPRIM temp sizing");
  if (std::max(t3, std::max(t4, t5)) > tmax) {
    tmax = std::max(t3, std::max(t4, t5)) + 256;
    void* nt = ctx->dalloc(tmax); blocks.push_back(nt);
    if (!nt) { release(); return ctx->fail(KMX_E_NOMEM, "count batch: device allocation failed"); }
    d_tmp = nt;
  }
  u32 kept = 0;
  if (runs) {
    t = tmax; if ((e = rocprim::exclusive_scan(d_tmp, t, d_cnt, d_start, 0u, (size_t)runs, rocprim::plus<u32>(), st)) != hipSuccess) return fail(e, "scan");
    hipLaunchKernelGGL(k_run_part_flags, dim3((runs + 255) / 256), dim3(256), 0, st, d_start, d_cnt, runs, d_spart, hard_min, d_rpart, d_flags);
    t = tmax; if ((e = rocprim::select(d_tmp, t, rocprim::counting_iterator<u32>(0), d_flags, d_idx, d_runs, (size_t)runs, st)) != hipSuccess) return fail(e, "select");
    if ((e = hipMemcpyAsync(&kept, d_runs, 4, hipMemcpyDeviceToHost, st)) != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "sync");
  }
  std::vector<u32> bounds(n_parts + 1, 0);
  KeyT* d_ok = d_sorted; u32* d_oc = (u32*)d_keys;      // both arrays are dead by now: reuse them for the output
  if (kept) {
    hipLaunchKernelGGL(k_gather_u16, dim3((kept + 255) / 256), dim3(256), 0, st, d_idx, kept, d_rpart, d_kp);
    t = tmax; if ((e = rocprim::radix_sort_pairs(d_tmp, t, d_kp, d_kp2, d_idx, d_idx2, (size_t)kept, 0, pbits, st)) != hipSuccess) return fail(e, "regroup");
    hipLaunchKernelGGL((k_gather_runs<KeyT>), dim3((kept + 255) / 256), dim3(256), 0, st, d_idx2, kept, d_uniq, d_cnt, d_ok, d_oc);
    hipLaunchKernelGGL(k_part_bounds, dim3((n_parts + 256) / 256), dim3(256), 0, st, d_kp2, kept, n_parts, d_bounds);
    if ((e = hipMemcpyAsync(bounds.data(), d_bounds, ((size_t)n_parts + 1) * 4, hipMemcpyDeviceToHost, st)) != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "sync");
  }
  clk.mark("regroup");
  if (out.dev()) {
    const int rc = pack_to_stores<KeyT>(ctx, d_ok, d_oc, bounds, n_parts, out);
    release();
    return rc;
  }
  // one D2H into pinned staging, then the per-partition output arrays are filled by a few host threads
  // (first-touch page faults of fresh allocations dominate a single-threaded copy)
  KeyT* h_k = kept ? (KeyT*)ctx->halloc((size_t)kept * sizeof(KeyT)) : nullptr;
  u32* h_c = kept ? (u32*)ctx->halloc((size_t)kept * 4) : nullptr;
  auto hrel = [&]() { ctx->hfree(h_k); ctx->hfree(h_c); };
  if (kept && (!h_k || !h_c)) { hrel(); release(); return ctx->fail(KMX_E_NOMEM, "count batch: host allocation failed"); }
  if (kept) {
    if ((e = hipMemcpyAsync(h_k, d_ok, (size_t)kept * sizeof(KeyT), hipMemcpyDeviceToHost, st)) != hipSuccess ||
        (e = hipMemcpyAsync(h_c, d_oc, (size_t)kept * 4, hipMemcpyDeviceToHost, st)) != hipSuccess ||
        (e = hipStreamSynchronize(st)) != hipSuccess) { hrel(); return fail(e, "download"); }
  }
  std::atomic<u32> next{0}; std::atomic<int> oom{0};
  auto fill = [&]() {
    for (u32 p; (p = next++) < n_parts;) {
      const size_t n = bounds[p + 1] - bounds[p];
      keys[p] = (uint64_t*)malloc(n ? n * sizeof(KeyT) : 8);
      counts[p] = (uint32_t*)malloc(n ? n * 4 : 4);
      if (!keys[p] || !counts[p]) { oom = 1; continue; }
      n_out[p] = n;
      if (n) { memcpy(keys[p], h_k + bounds[p], n * sizeof(KeyT)); memcpy(counts[p], h_c + bounds[p], n * 4); }
    }
  };
  {
    const unsigned nthr = std::max(1u, std::min({16u, std::thread::hardware_concurrency(), n_parts}));
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nthr; t++) th.emplace_back(fill);
    fill();
    for (auto& x : th) x.join();
  }
  hrel();
  if (oom) { release(); return ctx->fail(KMX_E_NOMEM, "count batch: host allocation failed"); }
  clk.mark("download");
  release();
  return KMX_OK;
}

// ---- keys of three and four words (k = 64 ... 127): sorted word by word (the library's stable radix sort on 64-bit keys, least
//      significant word first, the partition last), runs counted by head flags + scans.  d_keys: total keys of kw words, partition p =
//      k-mers [kmoff[p], kmoff[p + 1]).  Host output: keys[p] = n_out[p] * kw words, low word first; or packed records in the stores ----
static int wide_sort_count(kmx_ctx* ctx, StageClock& clk, const u64* d_keys, int kw, const std::vector<u64>& kmoff, u32 n_parts, u32 total, u32 hard_min, const CountOut& co)
{
  hipStream_t st = ctx->stream; hipError_t e;
  std::vector<void*> blocks;
  auto release = [&]() { for (void* b : blocks) ctx->dfree(b); };
  auto get = [&](size_t bytes) { void* b = ctx->dalloc(bytes ? bytes : 256); blocks.push_back(b); return b; };
  u64* d_w = (u64*)get((size_t)total * 8), *d_w2 = (u64*)get((size_t)total * 8);
  u32* d_perm = (u32*)get((size_t)total * 4), *d_perm2 = (u32*)get((size_t)total * 4);
  u16* d_kpart = (u16*)get((size_t)total * 2), *d_gp = (u16*)get((size_t)total * 2), *d_sp = (u16*)get((size_t)total * 2);
  u32* d_kmo = (u32*)get(((size_t)n_parts + 1) * 4);
  u32* d_flag = (u32*)get((size_t)total * 4), *d_incl = (u32*)get((size_t)total * 4), *d_start = (u32*)get(((size_t)total + 1) * 4);
  size_t t1 = 0, t2 = 0, t3 = 0;
  unsigned pbits = 1; while ((1u << pbits) < n_parts) pbits++;
  e = rocprim::radix_sort_pairs(nullptr, t1, d_w, d_w2, d_perm, d_perm2, (size_t)total, 0, 64, st);
  if (e == hipSuccess) e = rocprim::radix_sort_pairs(nullptr, t2, d_gp, d_sp, d_perm, d_perm2, (size_t)total, 0, pbits, st);
  if (e == hipSuccess) e = rocprim::inclusive_scan(nullptr, t3, d_flag, d_incl, (size_t)total, rocprim::plus<u32>(), st);
  const size_t tmax = std::max(t1, std::max(t2, t3)) + 256;
  void* d_tmp = get(tmax);
  for (void* b : blocks) if (!b) { release(); return ctx->fail(KMX_E_NOMEM, "count (k >= 64): device allocation failed"); }
  auto fail = [&](hipError_t er, const char* what) { release(); return ctx->fail(KMX_E_HIP, std::string(what) + ": " + hipGetErrorString(er)); };
  if (e != hipSuccess) return fail(e, "rocPRIM temp sizing");
  std::vector<u32> kmo32(kmoff.begin(), kmoff.end());
  if ((e = hipMemcpyAsync(d_kmo, kmo32.data(), ((size_t)n_parts + 1) * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e, "count upload");
  const dim3 gt((total + 255) / 256), b256(256);
  hipLaunchKernelGGL(k_fill_kpart, gt, b256, 0, st, d_kmo, n_parts, total, d_kpart);
  hipLaunchKernelGGL(k_wide_iota, gt, b256, 0, st, d_perm, total);
  for (int w = 0; w < kw; w++) {
    hipLaunchKernelGGL(k_wide_gather_word, gt, b256, 0, st, d_keys, d_perm, total, kw, w, d_w);
    size_t t = tmax;
    if ((e = rocprim::radix_sort_pairs(d_tmp, t, d_w, d_w2, d_perm, d_perm2, (size_t)total, 0, 64, st)) != hipSuccess) return fail(e, "radix_sort_pairs");
    std::swap(d_perm, d_perm2);
  }
  hipLaunchKernelGGL(k_gather_u16, gt, b256, 0, st, d_perm, total, d_kpart, d_gp);
  { size_t t = tmax; if ((e = rocprim::radix_sort_pairs(d_tmp, t, d_gp, d_sp, d_perm, d_perm2, (size_t)total, 0, pbits, st)) != hipSuccess) return fail(e, "radix_sort_pairs"); }
  std::swap(d_perm, d_perm2);
  clk.mark("sort");
  hipLaunchKernelGGL(k_wide_heads, gt, b256, 0, st, d_keys, d_perm, d_sp, total, kw, d_flag);
  { size_t t = tmax; if ((e = rocprim::inclusive_scan(d_tmp, t, d_flag, d_incl, (size_t)total, rocprim::plus<u32>(), st)) != hipSuccess) return fail(e, "scan"); }
  hipLaunchKernelGGL(k_wide_run_starts, gt, b256, 0, st, d_flag, d_incl, total, d_start);
  u32 runs = 0;
  if ((e = hipMemcpyAsync(&runs, d_incl + (total - 1), 4, hipMemcpyDeviceToHost, st)) != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "sync");      // (kmo32 is this frame's, too)
  u32* d_cnt = (u32*)get((size_t)runs * 4), *d_keep = (u32*)get((size_t)runs * 4), *d_pos = (u32*)get(((size_t)runs + 1) * 4);
  u32* d_bounds = (u32*)get(((size_t)n_parts + 1) * 4);
  if (!d_cnt || !d_keep || !d_pos || !d_bounds) { release(); return ctx->fail(KMX_E_NOMEM, "count (k >= 64): device allocation failed"); }
  const dim3 gr((runs + 255) / 256);
  hipLaunchKernelGGL(k_wide_run_keep, gr, b256, 0, st, d_start, runs, hard_min, d_cnt, d_keep);
  hist_runs(ctx, d_cnt, runs);
  { size_t t = 0;
    if ((e = rocprim::exclusive_scan(nullptr, t, d_keep, d_pos, 0u, (size_t)runs, rocprim::plus<u32>(), st)) != hipSuccess) return fail(e, "scan size");
    void* tmp = t <= tmax ? d_tmp : get(t);
    if (!tmp) { release(); return ctx->fail(KMX_E_NOMEM, "count (k >= 64): device allocation failed"); }
    if ((e = rocprim::exclusive_scan(tmp, t, d_keep, d_pos, 0u, (size_t)runs, rocprim::plus<u32>(), st)) != hipSuccess) return fail(e, "scan"); }
  u32 last[2] = {0, 0};
  if ((e = hipMemcpyAsync(&last[0], d_pos + (runs - 1), 4, hipMemcpyDeviceToHost, st)) != hipSuccess ||
      (e = hipMemcpyAsync(&last[1], d_keep + (runs - 1), 4, hipMemcpyDeviceToHost, st)) != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "sync");
  const u32 kept = last[0] + last[1];
  clk.mark("rle");
  std::vector<u32> bounds(n_parts + 1, 0);
  std::vector<u64> h_k(co.dev() ? 0 : (size_t)kept * kw); std::vector<u32> h_c(co.dev() ? 0 : kept);
  if (kept) {
    u64* d_ok = (u64*)get((size_t)kept * kw * 8); u32* d_oc = (u32*)get((size_t)kept * 4); u16* d_op = (u16*)get((size_t)kept * 2);
    if (!d_ok || !d_oc || !d_op) { release(); return ctx->fail(KMX_E_NOMEM, "count (k >= 64): device allocation failed"); }
    hipLaunchKernelGGL(k_wide_emit, gr, b256, 0, st, d_keys, d_perm, d_sp, d_start, d_cnt, d_keep, d_pos, runs, kw, d_ok, d_oc, d_op);
    hipLaunchKernelGGL(k_part_bounds, dim3((n_parts + 256) / 256), b256, 0, st, d_op, kept, n_parts, d_bounds);
    if (co.dev()) {      // (kmx_count_reads_dev: the kept pairs become packed records in the stores of the GPUs that merge them)
      if ((e = hipMemcpyAsync(bounds.data(), d_bounds, ((size_t)n_parts + 1) * 4, hipMemcpyDeviceToHost, st)) != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "sync");
      const int rc = kw == 3 ? pack_to_stores<WideKey<3>>(ctx, (const WideKey<3>*)d_ok, d_oc, bounds, n_parts, co)
                             : pack_to_stores<WideKey<4>>(ctx, (const WideKey<4>*)d_ok, d_oc, bounds, n_parts, co);
      release();
      clk.mark("pack");
      return rc;
    }
    if ((e = hipMemcpyAsync(bounds.data(), d_bounds, ((size_t)n_parts + 1) * 4, hipMemcpyDeviceToHost, st)) != hipSuccess ||
        (e = hipMemcpyAsync(h_k.data(), d_ok, (size_t)kept * kw * 8, hipMemcpyDeviceToHost, st)) != hipSuccess ||
        (e = hipMemcpyAsync(h_c.data(), d_oc, (size_t)kept * 4, hipMemcpyDeviceToHost, st)) != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "download");
  }
  release();
  if (co.dev()) { for (u32 p = 0; p < n_parts; p++) { co.lists[p].recs = nullptr; co.lists[p].n = 0; } return KMX_OK; }      // (nothing kept)
  for (u32 p = 0; p < n_parts; p++) {
    const size_t n = bounds[p + 1] - bounds[p];
    co.keys[p] = (uint64_t*)malloc(n ? n * kw * 8 : 8);
    co.counts[p] = (uint32_t*)malloc(n ? n * 4 : 4);
    if (!co.keys[p] || !co.counts[p]) return ctx->fail(KMX_E_NOMEM, "count (k >= 64): host allocation failed");
    co.n_out[p] = n;
    if (n) { memcpy(co.keys[p], h_k.data() + (size_t)bounds[p] * kw, n * kw * 8); memcpy(co.counts[p], h_c.data() + bounds[p], n * 4); }
  }
  clk.mark("download");
  return KMX_OK;
}

// ---- a batch from its record stream to its counts: decode (one lane per k-mer), partition-local count, the library sort if a
//      bucket is beyond the count kernel.  d_prefix: u64[nr + 1], record i starts at byte lo(d_prefix[i]), its first k-mer is number
//      hi(d_prefix[i]); d_rpart[i]: the record's partition (index into pid / kmoff); kmoff: first k-mer of every partition ----
static int decode_and_count(kmx_ctx* ctx, StageClock& clk, const u8* d_recs, const u64* d_prefix, const u16* d_rpart, u32 nr, u64 total, u32 n_parts,
                            const std::vector<u64>& kmoff, const std::vector<u64>& pid, u32 k, int hash_mode, u64 window, u32 hard_min, const CountOut& co,
                            const u32* d_sbase = nullptr /* set: d_recs are packed bases (k_pack_bases), record i starts at base d_sbase[i] */)
{
  const int kw = (k + 31) / 32;      // (kmer.hpp:215: the words files, hashes and comparisons see -- two at k = 64, three up to 96, four beyond)
  const bool wide_k = k >= 64;       // Kmer<96> / Kmer<128> (loop_executor.hpp:47-63): records of up to 92 / 124 k-mers, k_superk_decode_wide
  if (wide_k && d_sbase) return ctx->fail(KMX_E_UNSUPPORTED, "k >= 64: counted from super-k-mer records (kmx_count_batch)");
  const size_t key_bytes = hash_mode ? 8 : (size_t)kw * 8;
  const u32 NB = (u32)((total + DK - 1) / DK);
  u64* d_pid = (u64*)ctx->dalloc(pid.empty() ? 8 : (size_t)n_parts * 8);      // (pid empty: partition p has id p, nothing to upload)
  u32* d_blk = (u32*)ctx->dalloc((size_t)NB * 4);
  void* d_keys = ctx->dalloc(total * key_bytes);
  std::vector<void*> blocks = {d_pid, d_blk, d_keys};
  auto release = [&]() { for (void* b : blocks) ctx->dfree(b); };
  for (void* b : blocks) if (!b) { release(); return ctx->fail(KMX_E_NOMEM, "count: device allocation failed"); }
  hipStream_t st = ctx->stream; hipError_t e;
  const u64* d_pid_arg = nullptr;
  if (hash_mode && !pid.empty()) {
    if ((e = hipMemcpyAsync(d_pid, pid.data(), (size_t)n_parts * 8, hipMemcpyHostToDevice, st)) != hipSuccess) { release(); return ctx->fail(KMX_E_HIP, std::string("count upload: ") + hipGetErrorString(e)); }
    d_pid_arg = d_pid;
  }
  hipLaunchKernelGGL(k_decode_block_starts, dim3((nr + 255) / 256), dim3(256), 0, st, d_prefix, nr, d_blk);
  const dim3 grid(NB), block(256);
#define KMX_DECODE(KW_, H_, D_) hipLaunchKernelGGL((k_superk_decode_kmers<KW_, H_, D_>), grid, block, 0, st, d_recs, d_prefix, d_blk, d_rpart, d_pid_arg, nr, (u32)total, (int)k, window, d_keys, d_sbase)
  if (wide_k) {
#define KMX_DECODE_W(KW_, H_) hipLaunchKernelGGL((k_superk_decode_wide<KW_, H_>), grid, block, 0, st, d_recs, d_prefix, d_blk, d_rpart, d_pid_arg, nr, (u32)total, (int)k, window, (u64*)d_keys)
    if (kw == 2 && !hash_mode) KMX_DECODE_W(2, 0); else if (kw == 2) KMX_DECODE_W(2, 1);      // (k = 64: two words, the records of a Kmer<96>)
    else if (kw == 3 && !hash_mode) KMX_DECODE_W(3, 0); else if (kw == 3) KMX_DECODE_W(3, 1); else if (!hash_mode) KMX_DECODE_W(4, 0); else KMX_DECODE_W(4, 1);
#undef KMX_DECODE_W
  } else if (d_sbase) {
    if (kw == 1 && !hash_mode) KMX_DECODE(1, 0, true); else if (kw == 1) KMX_DECODE(1, 1, true); else if (!hash_mode) KMX_DECODE(2, 0, true); else KMX_DECODE(2, 1, true);
  } else {
    if (kw == 1 && !hash_mode) KMX_DECODE(1, 0, false); else if (kw == 1) KMX_DECODE(1, 1, false); else if (!hash_mode) KMX_DECODE(2, 0, false); else KMX_DECODE(2, 1, false);
  }
#undef KMX_DECODE
  if ((e = hipGetLastError()) != hipSuccess) { release(); return ctx->fail(KMX_E_HIP, std::string("k_superk_decode_kmers: ") + hipGetErrorString(e)); }
  clk.mark("decode");
  int rc;
  // partition-local sample sort / hash count first (count_sort.hpp); the library sort when a bucket is beyond it
  if (!hash_mode && kw > 2) {
    // (round 5: the same sample sort as below with keys of 24 / 32 bytes; a bucket beyond its LDS sort, KMX_COUNT_SORT=library: word by word)
    rc = kw == 3 ? partition_sort_count<WideKey<3>>(ctx, clk, (WideKey<3>*)d_keys, kmoff, n_parts, hard_min, co)
                 : partition_sort_count<WideKey<4>>(ctx, clk, (WideKey<4>*)d_keys, kmoff, n_parts, hard_min, co);
    if (rc == 1) {
      // (a bucket beyond the LDS sort: wide keys are split on their two most significant words only, so k-mers that share their first 33-64
      //  bases -- low-complexity reads -- crowd one bucket: the whole batch then takes the word-by-word library sort.  KMX_TRACE says so.)
      if (list_copy_trace()) fprintf(stderr, "[kmx count] k = %u: a bucket of the sample sort overflowed -- the batch (%llu k-mers) goes through the library's radix passes\n", k, (unsigned long long)total);
      if (!co.dev()) for (u32 p = 0; p < n_parts; p++) { free(co.keys[p]); free(co.counts[p]); co.keys[p] = nullptr; co.counts[p] = nullptr; co.n_out[p] = 0; }
      rc = wide_sort_count(ctx, clk, (const u64*)d_keys, kw, kmoff, n_parts, (u32)total, hard_min, co);
    }
    release();
    return rc;
  }
  if (hash_mode || kw == 1) rc = partition_sort_count<u64>(ctx, clk, (u64*)d_keys, kmoff, n_parts, hard_min, co);
  else rc = partition_sort_count<__uint128_t>(ctx, clk, (__uint128_t*)d_keys, kmoff, n_parts, hard_min, co);
  if (rc == 1) {
    if (!co.dev()) for (u32 p = 0; p < n_parts; p++) { free(co.keys[p]); free(co.counts[p]); co.keys[p] = nullptr; co.counts[p] = nullptr; co.n_out[p] = 0; }
    unsigned key_bits = 2 * k;
    if (hash_mode) { u64 top = pid.empty() ? (u64)n_parts - 1 : 0; for (u32 p = 0; p < n_parts && !pid.empty(); p++) top = std::max(top, pid[p]); key_bits = 64; const unsigned __int128 span = (unsigned __int128)window * (top + 1);
      if (span < ((unsigned __int128)1 << 63)) { key_bits = 1; while ((((u64)1) << key_bits) < (u64)span) key_bits++; } }
    // the partition of every k-mer: the library sort's second key
    u16* d_kpart = (u16*)ctx->dalloc(total * 2); u32* d_kmo = (u32*)ctx->dalloc(((size_t)n_parts + 1) * 4);
    blocks.push_back(d_kpart); blocks.push_back(d_kmo);
    if (!d_kpart || !d_kmo) { release(); return ctx->fail(KMX_E_NOMEM, "count: device allocation failed"); }
    std::vector<u32> kmo32(kmoff.begin(), kmoff.end());
    if ((e = hipMemcpyAsync(d_kmo, kmo32.data(), ((size_t)n_parts + 1) * 4, hipMemcpyHostToDevice, st)) != hipSuccess) { release(); return ctx->fail(KMX_E_HIP, std::string("count upload: ") + hipGetErrorString(e)); }
    hipLaunchKernelGGL(k_fill_kpart, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_kmo, n_parts, (u32)total, d_kpart);
    if ((e = hipStreamSynchronize(st)) != hipSuccess) { release(); return ctx->fail(KMX_E_HIP, std::string("k_fill_kpart: ") + hipGetErrorString(e)); }      // (kmo32 is this frame's)
    if (hash_mode || kw == 1) rc = batch_sort_rle<u64>(ctx, clk, (u64*)d_keys, d_kpart, (u32)total, n_parts, std::min(key_bits, 64u), hard_min, co);
    else rc = batch_sort_rle<__uint128_t>(ctx, clk, (__uint128_t*)d_keys, d_kpart, (u32)total, n_parts, std::min(key_bits, 128u), hard_min, co);
  }
  release();
  return rc;
}

extern "C" int kmx_count_batch(kmx_ctx* ctx, uint32_t n_parts, const uint8_t* const* superk, const uint64_t* len,
                               uint32_t k, int hash_mode, uint64_t window, const uint64_t* partition_ids, uint32_t hard_min,
                               uint64_t** keys, uint32_t** counts, uint64_t* n_out)
{
  if (!ctx) return KMX_E_INVAL;
  if (!n_parts || !superk || !len || !keys || !counts || !n_out) return ctx->fail(KMX_E_INVAL, "kmx_count_batch: null argument");
  if (k < 8 || k > 127) return ctx->fail(KMX_E_UNSUPPORTED, "k-mer size outside 8..127");
  if (hash_mode && (window == 0 || !partition_ids)) return ctx->fail(KMX_E_INVAL, "hash mode needs window and partition ids");
  if (n_parts > 65535) return ctx->fail(KMX_E_UNSUPPORTED, "more than 65535 partitions in one batch");
  KMX_HIP(ctx, hipSetDevice(ctx->device));
  for (u32 p = 0; p < n_parts; p++) { keys[p] = nullptr; counts[p] = nullptr; n_out[p] = 0; }

  // host: record offsets of every stream (a record's length depends on its first byte only).  Streams are
  // independent, so they are walked by a few threads: once to size the tables, once to fill them.
  StageClock clk(ctx->stream);
  std::vector<u64> n_rec(n_parts, 0), n_km(n_parts, 0);
  std::atomic<u32> next{0}; std::atomic<int> bad{0};
  const unsigned nthr = std::max(1u, std::min({16u, std::thread::hardware_concurrency(), n_parts}));
  auto run_threads = [&](auto&& fn) {
    next = 0;
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nthr; t++) th.emplace_back([&]() { for (u32 p; (p = next++) < n_parts;) fn(p); });
    for (u32 p; (p = next++) < n_parts;) fn(p);
    for (auto& x : th) x.join();
  };
  for (u32 p = 0; p < n_parts; p++) if (len[p] && !superk[p]) return ctx->fail(KMX_E_INVAL, "null stream");
  run_threads([&](u32 p) {
    const uint8_t* s = superk[p]; u64 pos = 0, nr = 0, nk = 0;
    while (pos < len[p]) {
      const u32 n = s[pos];
      const u64 nb = ((u64)k + n - 1 + 3) / 4;
      // (a super-k-mer holds at most 28 k-mers for k < 32, 60 above -- Sequence2SuperKmer.hpp:90-132; the lane-per-k-mer decode cuts a
      //  record's k-mers from one 64- / 128-bit window and relies on it: a longer record is a malformed stream, not silently wrong keys)
      if (n == 0 || n > (k < 32 ? 28u : k < 64 ? 60u : k < 96 ? 92u : 124u) || pos + 1 + nb > len[p]) { bad = 1; return; }      // ((type bits - 8) / 2: Kmer<96> 92, Kmer<128> 124)
      nr++; nk += n; pos += 1 + nb;
    }
    n_rec[p] = nr; n_km[p] = nk;
  });
  if (bad) return ctx->fail(KMX_E_INVAL, "malformed super-k-mer stream");
  std::vector<u64> rec_base(n_parts + 1, 0), km_base(n_parts + 1, 0), byte_base(n_parts + 1, 0);
  for (u32 p = 0; p < n_parts; p++) { rec_base[p + 1] = rec_base[p] + n_rec[p]; km_base[p + 1] = km_base[p] + n_km[p]; byte_base[p + 1] = byte_base[p] + len[p]; }
  const u64 bytes = byte_base[n_parts], total = km_base[n_parts];
  if (total >= 0xFFFFFF00ULL || bytes >= 0xFFFFFF00ULL) return ctx->fail(KMX_E_UNSUPPORTED, "more than 2^32 k-mers or bytes in one batch: split it");
  const u32 nr = (u32)rec_base[n_parts];
  u64* prefix = (u64*)ctx->halloc(((size_t)nr + 1) * 8);
  u16* rec_part = (u16*)ctx->halloc((size_t)nr * 2 + 2);
  auto hrelease = [&]() { ctx->hfree(prefix); ctx->hfree(rec_part); };
  if (!prefix || !rec_part) { hrelease(); return ctx->fail(KMX_E_NOMEM, "count batch: host allocation failed"); }
  run_threads([&](u32 p) {
    const uint8_t* s = superk[p]; u64 pos = 0, r = rec_base[p], o = km_base[p];
    while (pos < len[p]) {
      const u32 n = s[pos];
      prefix[r] = (o << 32) | (byte_base[p] + pos); rec_part[r] = (u16)p;
      r++; o += n; pos += 1 + ((u64)k + n - 1 + 3) / 4;
    }
  });
  prefix[nr] = (total << 32) | bytes;
  clk.mark("parse");
  auto empty_out = [&]() { for (u32 p = 0; p < n_parts; p++) { if (!keys[p]) { keys[p] = (uint64_t*)malloc(8); counts[p] = (uint32_t*)malloc(4); n_out[p] = 0; } } };
  if (total == 0) { hrelease(); empty_out(); return KMX_OK; }
  u8* d_recs = (u8*)ctx->dalloc(bytes + 128);      // (the decode reads whole words past a record: 64 bytes from its start for k >= 64)
  u64* d_prefix = (u64*)ctx->dalloc(((size_t)nr + 1) * 8);
  u16* d_rp = (u16*)ctx->dalloc((size_t)nr * 2);
  std::vector<void*> blocks = {d_recs, d_prefix, d_rp};
  auto release = [&]() { for (void* b : blocks) ctx->dfree(b); hrelease(); };
  for (void* b : blocks) if (!b) { release(); return ctx->fail(KMX_E_NOMEM, "count batch: device allocation failed"); }
  hipStream_t st = ctx->stream; hipError_t e = hipSuccess;
  u64 off = 0;
  for (u32 p = 0; p < n_parts && e == hipSuccess; p++) { if (len[p]) e = hipMemcpyAsync(d_recs + off, superk[p], len[p], hipMemcpyHostToDevice, st); off += len[p]; }
  std::vector<u64> pid(n_parts, 0); if (hash_mode) for (u32 p = 0; p < n_parts; p++) pid[p] = partition_ids[p];
  if (e == hipSuccess) e = hipMemcpyAsync(d_prefix, prefix, ((size_t)nr + 1) * 8, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_rp, rec_part, (size_t)nr * 2, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) { release(); return ctx->fail(KMX_E_HIP, std::string("count batch upload: ") + hipGetErrorString(e)); }
  CountOut co; co.keys = keys; co.counts = counts; co.n_out = n_out;
  const int rc = decode_and_count(ctx, clk, d_recs, d_prefix, d_rp, nr, total, n_parts, km_base, pid, k, hash_mode, window, hard_min, co);
  release();
  if (rc != KMX_OK) for (u32 p = 0; p < n_parts; p++) { free(keys[p]); free(counts[p]); keys[p] = nullptr; counts[p] = nullptr; n_out[p] = 0; }
  return rc;
}


// ---- split -> count without the streams leaving HBM (kmx_count_reads; called from superk.hip) ---------------------------
// record i of the partition-ordered stream: byte offset = low word of d_prefix[i], first k-mer index = high word (d_prefix has one
// more entry than there are records: the totals) -- exactly what the decode takes
int kmx_count_from_device(kmx_ctx* ctx, const u8* d_recs, const u64* d_prefix, const u16* d_part, u32 nr, u64 total, u32 n_parts, const u64* part_kmer_off, const kmx_count_req& rq,
                          const u32* d_sbase)
{
  if (total >= 0xFFFFFF00ULL) return ctx->fail(KMX_E_UNSUPPORTED, "more than 2^32 k-mers in one batch: split it");
  StageClock clk(ctx->stream, "count_reads");
  CountOut co; co.keys = rq.keys; co.counts = rq.counts; co.n_out = rq.n_out; co.stores = rq.stores; co.n_stores = rq.n_stores; co.lists = rq.lists; co.inner = rq.inner_parts;
  if (total == 0) {
    for (u32 p = 0; p < n_parts; p++) { if (co.dev()) { co.lists[p].recs = nullptr; co.lists[p].n = 0; } else { rq.keys[p] = (uint64_t*)malloc(8); rq.counts[p] = (uint32_t*)malloc(4); rq.n_out[p] = 0; } }
    return KMX_OK;
  }
  std::vector<u64> pid;      // (empty: partition p of the stream has id p)
  if (rq.inner_parts && rq.hash_mode) { pid.resize(n_parts); for (u32 p = 0; p < n_parts; p++) pid[p] = p % rq.inner_parts; }      // (the window of p' = sample * inner + p is p's)
  std::vector<u64> kmoff(part_kmer_off, part_kmer_off + n_parts + 1);
  return decode_and_count(ctx, clk, d_recs, d_prefix, d_part, nr, total, n_parts, kmoff, pid, rq.k, rq.hash_mode, rq.window, rq.hard_min, co, d_sbase);
}

// ---- round 6: the count behind the sync-free split (superk_fast.hpp): the same kernels as decode_and_count + partition_sort_count,
//      their sizes and tables read from the device (grids cover bounds, the surplus workgroups leave at once), nothing read back until
//      the kept sizes are known -- then the lists are packed into the stores as before (compact_to_stores) ----
kmx::SkfLayout kmx_fast_layout(int key_words)
{
  if (key_words <= 1) return SkfLayout{cs_target<u64>(), cs_chunk<u64>(), (u32)CS_MAXB, (u32)CsCap<u64>::sample};
  return SkfLayout{cs_target<__uint128_t>(), cs_chunk<__uint128_t>(), (u32)CS_MAXB, (u32)CsCap<__uint128_t>::sample};
}
template <typename KeyT>
static int fast_tail_impl(kmx_ctx* ctx, StageClock& clk, const kmx_fast_split& F, const kmx_count_req& rq, const CountOut& co)
{
  typedef typename CsSpl<KeyT>::type SplT;
  const u64 kb = F.kmer_bound; const u32 TBm = F.tb_max, P = F.n_parts;
  hipStream_t st = ctx->stream; hipError_t e;
  std::vector<void*> blocks;
  auto dal = [&](size_t b) { void* p = ctx->dalloc(b); blocks.push_back(p); return p; };
  auto release = [&]() { for (void* b : blocks) ctx->dfree(b); };
  KeyT* d_keys = (KeyT*)dal(sizeof(KeyT) * kb), *d_bkeys = (KeyT*)dal(sizeof(KeyT) * kb), *d_tk = (KeyT*)dal(sizeof(KeyT) * kb);
  u32* d_tc = (u32*)dal(4 * kb);
  SplT* d_spl = (SplT*)dal(sizeof(SplT) * (size_t)TBm);
  u32* d_boff = (u32*)dal(4 * ((size_t)TBm + 2)), *d_cur = (u32*)dal(4 * ((size_t)TBm + 2)), *d_nkept = (u32*)dal(4 * ((size_t)TBm + 2)), *d_koff = F.d_koff;      // (the offsets of the kept pairs: in the caller's block, read back with the control block)
  u32* d_big = (u32*)dal(4 * (size_t)SKF_BIG_CAP);
  u32* const h_koff = F.h_koff;
  bool ok = h_koff != nullptr && d_koff != nullptr;
  for (void* b : blocks) ok = ok && b;
  if (!ok) { release(); return ctx->fail(KMX_E_NOMEM, "count: device allocation failed"); }
  auto fail = [&](hipError_t er, const char* what) { release(); return ctx->fail(KMX_E_HIP, std::string(what) + ": " + hipGetErrorString(er)); };
  const CsPart* d_parts = reinterpret_cast<const CsPart*>(F.d_parts);
  static_assert(sizeof(CsPart) == sizeof(uint4), "k_sk_scan writes the sample sort's partitions as uint4");
  constexpr int KWD = sizeof(KeyT) == 8 ? 1 : 2;
  const dim3 gd(F.nb_max), bd(256);
#define KMX_DECODE_F(KW_, H_) hipLaunchKernelGGL((k_superk_decode_kmers<KW_, H_, true>), gd, bd, 0, st, (const u8*)F.d_words, F.d_boff, F.d_blk, F.d_part16, (const u64*)nullptr, 0u, 0u, (int)rq.k, rq.window, (void*)d_keys, F.d_sbase, (const SkfCtl*)F.d_ctl, F.d_strand)
  if (rq.hash_mode) { if (rq.k <= 32) KMX_DECODE_F(1, 1); else KMX_DECODE_F(2, 1); }
  else if (KWD == 1) KMX_DECODE_F(1, 0); else KMX_DECODE_F(2, 0);
#undef KMX_DECODE_F
  // the look-up table in front of the walks' bucket search: on the keys' top 32 significant bits (KMX_COUNT_LUT=0: the plain binary search)
  u32 tshift = 0;
  if (rq.hash_mode) { unsigned __int128 span = (unsigned __int128)rq.window * (rq.inner_parts ? rq.inner_parts : P); u32 bits = 0; while (bits < 64 && (span >> bits)) bits++; tshift = bits > 32 ? bits - 32 : 0; }
  else tshift = 2 * rq.k > 32 ? 2 * rq.k - 32 : 0;
  static_assert(sizeof(CsLut) % 4 == 0, "the walks copy the table as dwords");
  CsLut* d_luts = nullptr;
  { const char* le = getenv("KMX_COUNT_LUT"); if (!(le && le[0] == '0')) { d_luts = (CsLut*)dal(sizeof(CsLut) * (size_t)P); if (!d_luts) { release(); return ctx->fail(KMX_E_NOMEM, "count: device allocation failed"); } } }
  hipLaunchKernelGGL((k_cs_splitters<KeyT>), dim3(P), dim3(CS_SPL_TPB), 0, st, d_keys, d_parts, d_spl, (const SkfCtl*)F.d_ctl, d_luts, tshift);
  hipLaunchKernelGGL((k_cs_walk<KeyT, false>), dim3(F.nc_max), dim3(CS_WALK_TPB), 0, st, d_keys, d_parts, (const CsChunk*)nullptr, d_spl, F.d_cnt, (KeyT*)nullptr, (const SkfCtl*)F.d_ctl, F.d_cfirst, P, (const CsLut*)d_luts, tshift);
  const dim3 gs((TBm + 1 + 4095) / 4096);      // (at most 256 workgroups: the caller bounds the batch)
  u32* d_agg = (u32*)dal(4 * 512);
  if (!d_agg || gs.x > 256) { release(); return ctx->fail(KMX_E_NOMEM, "count: device allocation failed"); }
  hipLaunchKernelGGL(k_cs_scan_mw, gs, dim3(1024), 0, st, F.d_cnt, (const u32*)&F.d_ctl->TB, d_boff, d_cur, d_agg, F.d_sflags);
  if (getenv("KMX_COUNT_SCATTER_PLAIN"))      // (the scatter of rounds 3-5, for comparison)
    hipLaunchKernelGGL((k_cs_walk<KeyT, true>), dim3(F.nc_max), dim3(CS_WALK_TPB), 0, st, d_keys, d_parts, (const CsChunk*)nullptr, d_spl, d_cur, d_bkeys, (const SkfCtl*)F.d_ctl, F.d_cfirst, P, (const CsLut*)d_luts, tshift);
  else
    hipLaunchKernelGGL((k_cs_scatter_staged<KeyT>), dim3(F.nc_max * (cs_chunk<KeyT>() / cs_schunk<KeyT>())), dim3(CS_WALK_TPB), 0, st, d_keys, d_parts, d_spl, d_cur, d_bkeys, (const SkfCtl*)F.d_ctl, F.d_cfirst, P, (const CsLut*)d_luts, tshift);
  if (F.behind_scatter) { const int brc = F.behind_scatter(); if (brc != KMX_OK) { (void)hipStreamSynchronize(st); release(); return brc; } }
  // 64-bit keys: the buckets by counting first (k_cs_wave_count) unless the context's last call had a quarter of its buckets' distinct keys
  // overflow the waves' tables -- data without repeats: the full sort then (KMX_COUNT_HASH_FIRST=0: always; =1: never mind the last call)
  u32* const d_lost = reinterpret_cast<u32*>(reinterpret_cast<u8*>(F.d_ctl) + 48);      // (a word of the control block's 64 bytes: cleared and read back with it)
  bool hash_first = false;
  if constexpr (KWD == 1) { const char* he = getenv("KMX_COUNT_HASH_FIRST"); hash_first = he ? he[0] != '0' : ctx->hash_lost_frac < 0.25; }
  if constexpr (KWD == 1) {
    if (hash_first)
      hipLaunchKernelGGL(k_cs_wave_count, dim3((TBm + CS_WAVES - 1) / CS_WAVES), dim3(64 * CS_WAVES), 0, st, (const u64*)d_bkeys, d_boff, rq.hard_min, (u64*)d_tk, d_tc, d_nkept, F.d_ctl, d_big, d_lost);
  }
  if (!hash_first)
  hipLaunchKernelGGL((k_cs_wave_sort<KeyT, 8, 16>), dim3((TBm + CS_WAVES - 1) / CS_WAVES), dim3(64 * CS_WAVES), 0, st, d_bkeys, d_boff, 0u, 0u, (u32)CsCap<KeyT>::cap,
                     rq.hard_min, d_tk, d_tc, d_nkept, (unsigned long long*)nullptr, &F.d_ctl->overflow, F.d_ctl, d_big);
  // the buckets beyond a wave's registers (a k-mer repeated a thousand times, an unlucky sample): listed by the kernel above, a few workgroups take them
  if constexpr (KWD == 1)
    hipLaunchKernelGGL(k_cs_count_hash, dim3(256), dim3(CS_TPB), 0, st, d_bkeys, d_boff, rq.hard_min, d_tk, d_tc, d_nkept, (unsigned long long*)nullptr, &F.d_ctl->overflow, 0u, (const SkfCtl*)F.d_ctl, (const u32*)d_big);
  else
    hipLaunchKernelGGL((k_cs_sort<KeyT, 4096>), dim3(256), dim3(CS_TPB), 0, st, d_bkeys, d_boff, rq.hard_min, d_tk, d_tc, d_nkept, (unsigned long long*)nullptr, &F.d_ctl->overflow, 0u, 1u, (const SkfCtl*)F.d_ctl, (const u32*)d_big);
  hipLaunchKernelGGL(k_cs_scan_mw, gs, dim3(1024), 0, st, d_nkept, (const u32*)&F.d_ctl->TB, d_koff, (u32*)nullptr, d_agg + 256, F.d_sflags + 256);
  // one store on this GPU: room for 1.25 x what the last call kept per k-mer is reserved and the lists are written there before
  // their size is known -- the read-back below is then the call's only wait
  constexpr size_t RB = sizeof(KeyT) + 4;
  kmx_store* const S0 = co.n_stores == 1 && co.stores[0]->device == ctx->device ? co.stores[0] : nullptr;
  u8* d_resv = nullptr; u32 cap_recs = 0;
  if (S0 && ctx->kept_per_kmer > 0.0) {
    cap_recs = (u32)std::min<double>((double)kb, (double)kb * ctx->kept_per_kmer * 1.25 + 65536.0);
    d_resv = (u8*)S0->try_reserve((size_t)cap_recs * RB);
    if (d_resv) hipLaunchKernelGGL((k_cs_compact_recs<KeyT>), dim3((TBm + CS_TPB / 64 - 1) / (CS_TPB / 64)), dim3(CS_TPB), 0, st, d_tk, d_tc, d_boff, d_koff, (const u32*)d_koff, d_resv, (const SkfCtl*)F.d_ctl, cap_recs);
  }
  struct Resv { kmx_store* s; u8* p; ~Resv() { if (s && p) s->commit(p, 0); } } resv{S0, d_resv};      // (left open by an error: given back)
  if ((e = hipGetLastError()) != hipSuccess) return fail(e, "count kernels");
  if ((e = hipMemcpyAsync(F.h_ctl, F.d_ctl, F.back_bytes, hipMemcpyDeviceToHost, st)) != hipSuccess) return fail(e, "count read-back");      // (the call's one copy back: control block, tables, offsets)
  kmx_count_chain_end(ctx);      // (everything of this call is queued: the next call of this GPU may start behind it)
  kmx_phase_mark(3);
  if (F.before_wait) { const int brc = F.before_wait(); if (brc != KMX_OK) { (void)hipStreamSynchronize(st); release(); return brc; } }
  kmx_phase_mark(4);
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "count read-back");
  kmx_phase_mark(5);
  clk.mark("split+decode+sort+count");
  if (clk.on) fprintf(stderr, "[kmx count_reads_fast] records %u k-mers %u buckets %u walk chunks %u listed buckets %u status %u overflow %u; the buckets %s\n", F.h_ctl->nd, F.h_ctl->total, F.h_ctl->TB, F.h_ctl->NC, F.h_ctl->n_big, F.h_ctl->status, F.h_ctl->overflow,
                      hash_first ? (std::string("by counting first (") + std::to_string(reinterpret_cast<const u32*>(F.h_ctl)[12]) + " of them sorted: their distinct keys beyond the table)").c_str() : "by the full sort");
  // (before the status word is looked at: a sample whose lost buckets overflow the list hands the call back -- the next one must not try again)
  if (hash_first && F.h_ctl->TB) ctx->hash_lost_frac = (double)reinterpret_cast<const u32*>(F.h_ctl)[12] / (double)F.h_ctl->TB;
  else if (!hash_first) ctx->hash_lost_frac *= 0.9;      // (a call by the full sort: counting first is tried again a dozen samples on -- a cohort may hold an assembly among its read sets)
  if (F.h_ctl->status || F.h_ctl->overflow) { release(); return 1; }
  const u32 TB = F.h_ctl->TB;
  std::vector<CsPart> parts(P);
  memcpy(parts.data(), F.h_parts, sizeof(CsPart) * P);
  if (F.h_ctl->total) ctx->kept_per_kmer = (double)h_koff[TB] / (double)F.h_ctl->total;
  if (d_resv && h_koff[TB] <= cap_recs) {      // the lists are in the store already
    for (u32 p = 0; p < P; p++) {
      const u32 lo = h_koff[parts[p].bucket0], hi = h_koff[parts[p].bucket0 + parts[p].nb];
      co.lists[p].n = hi - lo; co.lists[p].recs = hi > lo ? d_resv + (size_t)lo * RB : nullptr;
    }
    S0->commit(d_resv, (size_t)h_koff[TB] * RB); resv.p = nullptr;
    release();
    kmx_phase_mark(6);
    return KMX_OK;
  }
  if (d_resv) { S0->commit(d_resv, 0); resv.p = nullptr; }      // (more kept than estimated: the exact way)
  const int rc = compact_to_stores<KeyT>(ctx, d_tk, d_tc, d_boff, d_koff, h_koff, parts, TB, co);
  release();
  clk.mark("pack");
  return rc;
}
int kmx_count_fast_tail(kmx_ctx* ctx, const kmx_fast_split& F, const kmx_count_req& rq)
{
  StageClock clk(ctx->stream, "count_reads_fast");
  CountOut co; co.keys = rq.keys; co.counts = rq.counts; co.n_out = rq.n_out; co.stores = rq.stores; co.n_stores = rq.n_stores; co.lists = rq.lists; co.inner = rq.inner_parts;
  if (rq.hash_mode || rq.k <= 32) return fast_tail_impl<u64>(ctx, clk, F, rq, co);
  return fast_tail_impl<__uint128_t>(ctx, clk, F, rq, co);
}

// the bases of a batch, 2 bits each (see k_pack_bases); `out` holds (n + 31) / 32 + 2 words
void kmx_launch_pack_bases(const char* d_bases, u64 n, u64* out, hipStream_t st)
{
  const u64 nw = (n + 31) / 32 + 2;
  hipLaunchKernelGGL(k_pack_bases, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, d_bases, n, out);
}
