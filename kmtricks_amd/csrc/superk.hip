// superk.hip -- reads -> canonical k-mers -> minimizers -> super-k-mers -> 2-bit records per partition
// on gfx950.  Replaces Model::iterate + ModelMinimizer::next + Sequence2SuperKmer::KmerFunctor +
// KmFillPartitions::processSuperkmer + SuperKmer::save (reference gatb kmer/impl/Model.hpp:725-765,
// 857-884, 1010-1139, 1254-1287, 1388-1433; kmer/impl/Sequence2SuperKmer.hpp:80-158;
// include/kmtricks/gatb/fill_partitions.hpp:59-105).
//
//   pass 1  k_superk_wave<false>: one WAVE per read, one lane per k-mer position; bases become ballot
//           bit planes, the m-mer value (min(m-mer, revcomp), or 4^m-1 when it contains AA except as
//           prefix) is computed instead of looked up, the window minimizer is a doubling min-scan with
//           wave shuffles (the reference's rolling minimizer always equals the window minimum, whatever
//           its tie rules), super-k-mer cuts (minimizer change, invalid k-mer, maxs k-mers) are ballots;
//           counts the read's super-k-mers;
//   scan    exclusive scan of the per-read counts (rocPRIM);
//   pass 2  k_superk_wave<true>: the same walk writes one descriptor per super-k-mer
//           {first base, n, partition, record bytes};
//   order   stable radix sort of the descriptors by partition + exclusive scan of their sizes
//           (rocPRIM) = the byte offset of every record inside its partition's stream, in read
//           order, exactly the order the reference appends them;
//   pack    k_superk_pack: one thread per record writes [u8 n][2-bit nucleotides]: with S the record
//           as a little-endian integer, digit d < k is base[k-1-d] of the first k-mer and digit
//           d >= k is base[d] (Model.hpp:1388-1433).
// Integer/byte work, HBM-bound on the base stream (1 B per base in, ~0.3 B per base out).
#include <cstdlib>
#include <cstring>
#include "kmx_host.hpp"
#include <atomic>
#include <thread>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "skf.hpp"

namespace kmx {

__device__ __forceinline__ bool nt_valid(u8 c)
{ // gatb tools/misc/api/Data.hpp:179-196
  const u8 u = c & 0xDF;
  return u == 'A' || u == 'C' || u == 'G' || u == 'T';
}

// value of an m-mer as the reference's minimizer table gives it: min(x, revcomp_m(x)), or 4^m - 1 when
// that contains AA anywhere but as a prefix (Model.hpp:1040-1064, 1220-1251) -- computed, not looked up
__device__ __forceinline__ u32 mmer_value(u32 x, int m)
{
  const u32 n1 = (1u << (2 * m)) - 1;                                     // m <= 15
  u32 t = __brev(x);
  t = ((t >> 1) & 0x55555555u) | ((t & 0x55555555u) << 1);                 // digits reversed, bits of a digit in order
  const u32 rc = (t >> (32 - 2 * m)) ^ (0xAAAAAAAAu & n1);                 // complement: A0 C1 T2 G3 -> digit ^ 2
  const u32 v = rc < x ? rc : x;
  const u64 mask_ma1 = 0x5555555555555555ULL & ((1ULL << ((m - 2) * 2)) - 1);
  u64 a1 = v; a1 = ~(a1 | (a1 >> 2)); a1 = ((a1 >> 1) & a1) & mask_ma1;
  return a1 ? n1 : v;
}
// bit i of y -> bit 2i (i < 16)
__device__ __forceinline__ u32 spread16(u32 y)
{
  y = (y | (y << 8)) & 0x00FF00FFu; y = (y | (y << 4)) & 0x0F0F0F0Fu;
  y = (y | (y << 2)) & 0x33333333u; y = (y | (y << 1)) & 0x55555555u;
  return y;
}

// One WAVE per read, one lane per k-mer position (a chunk = 64 consecutive base positions):
//   * the chunk's bases become three 64-bit bit planes (two code bits, one "invalid" bit) with ballots;
//     a lane funnel-shifts them to its own position: its m-mer is m bits of each plane, its k-mer is
//     valid iff k bits of the invalid plane are zero -- no per-lane base loop, no table;
//   * the minimizer of the k-mer at lane l is the minimum of the m-mer values of lanes l .. l+k-m
//     (what the reference's rolling minimizer always equals, whatever its tie rules): a doubling
//     min-scan with wave shuffles;
//   * super-k-mers are runs of valid k-mers with one minimizer, cut every `maxs` k-mers
//     (Sequence2SuperKmer.hpp:90-158): break / start / end flags are ballots, a lane that ends a
//     super-k-mer finds its start with a count-leading-zeros on the start mask.  Only the state of the
//     last owned k-mer is carried to the next chunk (64-(k-m)-1 positions further).
//   * STATS (PartiInfo<5>, fill_partitions.hpp:67-102; SampleRepart, RepartitionAlgorithm.cpp:182-215): a lane knows its
//     k-mer's strand (forward < reverse complement: the first digit where the k-mer differs from its reverse
//     complement, found with a bit reversal of the code planes and a count-trailing-zeros); kx-mers are runs of at
//     most 5 k-mers of one strand inside a super-k-mer -- start / end flags are ballots again --, the lane that ends
//     one adds 1 to the (partition, run length, radix) counter, radix = top 4 nucleotides of the run's first
//     canonical k-mer (forward run) or of its last one (reverse run).
struct SkSort {           // what the partition sort of the descriptors takes, written with the descriptors (EMIT); all null or all set
  u16* keys;              // [n] partition of descriptor i
  u32* ids;               // [n] i
  u32* sizes;             // [n] bytes of its record: 1 + ceil((k + n - 1) / 4)
};
struct SkStats {          // device tables, zeroed by the caller; any of them may be null
  u32* pc;                // [nb_parts][5][256] kx-mers per (partition, x, radix)
  u32* ms;                // [4^m] super-k-mers per minimizer
  u32* mk;                // [4^m] k-mers per minimizer
  u32* mx;                // [4^m] kx-mers per minimizer
  // DEFER (k_superk_wave<.., DEFER>): no counter is touched while the reads are walked -- a descriptor takes its minimizer and the
  // strands of its k-mers along, and k_part_stats counts partition by partition once the descriptors are sorted
  // [n] per descriptor, ONE 16-byte record (k_part_stats gathers them through the sorted ids: one sector each):
  //   .x = strands (bit t: k-mer t of the super-k-mer is its own canonical form, t < 60) | (k-mers & 15) << 60
  //   .y = first base | minimizer << 32 (m <= 15: 30 bits) | (k-mers >> 4) << 62
  ulonglong2* sk_rec;
};

// LB (one pass instead of count + scan + emit): a wave keeps its read's descriptors in LDS, the workgroup's four reads get their
// place among all descriptors -- in read order, as the two-pass path gives it -- by a decoupled look-back over the workgroups:
// a workgroup takes a ticket (its number: the reads 4 t .. 4 t + 3), publishes how many descriptors it has (flag 1), adds up its
// predecessors' numbers until one of them has published its inclusive prefix (flag 2), publishes its own.  A workgroup only waits
// for workgroups with smaller tickets, which are running.  The last state word holds the total.
struct SkLook { unsigned long long* state; u32* ticket; u32* over; u32 cap; };
// several samples in one call (kmx_count_reads_dev_multi): the reads of sample s are [first[s], first[s + 1]); its partitions are
// s * parts + repart[minimizer], its per-minimizer tables start at s * nm.  first == nullptr: one sample
struct SkMulti { const u32* first; u32 n; u32 parts; u32 nm; };      // state[workgroup]: flag << 62 | count or prefix; cap: descriptors that fit
constexpr u32 SK_WCAP = 512;      // descriptors of one read the LDS takes (the host sends longer reads the two-pass way)
// CH (round 6, the sync-free count path): ONE walk -- a wave takes `rpw` consecutive reads (a chunk) and writes their descriptors
// back to back from slot offsets[first read of the chunk] on (a read holds fewer super-k-mers than bases: the chunks' slot ranges
// never meet), their number to counts[chunk].  No count pass, no scan, no size read back by the host: k_sk_hist / k_sk_scan /
// k_sk_scatter below put the descriptors in partition order, read order kept inside a partition.
template <bool EMIT, bool STATS, bool LB = false, bool DEFER = false, bool CH = false>
__global__ __launch_bounds__(256)
void k_superk_wave(const char* __restrict__ bases, const u64* __restrict__ offsets, u64 n_seqs,
                   int k, int m, int maxs, const u16* __restrict__ repart,
                   u32* __restrict__ counts, const u32* __restrict__ desc_off, SkDesc* __restrict__ desc, SkStats S, SkSort so, SkLook lk, SkMulti mu, u32 rpw = 1)
{
  __shared__ SkDesc wbuf[LB ? 4 : 1][LB ? SK_WCAP : 1];
  __shared__ u32 wcnt[4];
  __shared__ u32 bid_s, base_s;
  static_assert(!LB || EMIT, "the look-back places descriptors");
  static_assert(!DEFER || (EMIT && STATS && !LB), "deferred statistics travel with the descriptors of the two-pass path");
  static_assert(!CH || (EMIT && !LB), "a chunk's descriptors are written where they are found");
  const int lane = threadIdx.x & 63;
  const u32 wave = threadIdx.x >> 6;
  u64 r = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (LB) {
    if (threadIdx.x == 0) bid_s = atomicAdd(lk.ticket, 1u);
    __syncthreads();
    r = (u64)bid_s * 4u + wave;
  }
  const u64 chunk = r;
  u64 r_end = r + 1;
  if (CH) { r = chunk * rpw; r_end = r + rpw < n_seqs ? r + rpw : n_seqs; }
  if (!LB && r >= n_seqs) return;
  u32 nsk = 0;
  u32 out = CH ? (u32)offsets[r] : 0u;
  const u32 out_first = out;
  for (; r < r_end; r++) {
  const bool act = r < n_seqs;
  const u64 b0 = act ? offsets[r] : 0, len = act ? offsets[r + 1] - b0 : 0;
  u32 smp = 0;
  if (mu.first) while (smp + 1 < mu.n && r >= (u64)mu.first[smp + 1]) smp++;      // (a handful of samples: uniform over the wave)
  const u32 pbase = smp * mu.parts, sbase = smp * mu.nm;
  if (len >= (u64)k) {
    const char* seq = bases + b0;
    const int nbm = k - m + 1;                       // m-mers per k-mer (<= 60)
    // WIDE (k - m >= 32: at k = 63, m = 10 a chunk would own 10 k-mers): the m-mer values of the NEXT 64 positions are computed as well
    // and a window that runs past lane 63 is the minimum of a suffix of this chunk's values and a prefix of the next chunk's (van
    // Herk): every lane has its minimizer, a chunk owns 63 k-mers for ~1.5x the instructions
    const bool wide = nbm > 32;
    const u32 C = wide ? 64u : 64u - (u32)nbm + 1u;  // k-mer positions of a chunk with their whole m-mer window at hand
    const u32 own = C - 1;                           // the last one only serves as look-ahead
    const u64 nk = len - (u64)k + 1;
    const u64 kmask = (1ULL << k) - 1;               // k <= 63
    const u32 mmask = (1u << m) - 1;
    if (!CH) out = (EMIT && !LB) ? desc_off[r] : 0;
    bool pv = false; u32 pmin = 0; u64 run_start = 0, open_start = 0;   // state of the last owned k-mer of the previous chunk
    int pw = 0; u64 t_start = 0, x_start = 0; u32 rf_open = 0;          // (STATS) its strand, strand-run start, kx-mer start + that k-mer's radix
    u64 hist = 0;                                                       // (DEFER) the strands of the 64 positions before p0: bit 63 is p0 - 1
    for (u64 p0 = 0; p0 < nk; p0 += own) {
      const u64 q = p0 + lane;
      const u8 c0 = q < len ? (u8)seq[q] : (u8)'N';
      const u8 c1 = q + 64 < len ? (u8)seq[q + 64] : (u8)'N';
      const u64 I0 = __ballot(!nt_valid(c0)), I1 = __ballot(!nt_valid(c1));
      const u64 A0 = __ballot((c0 >> 1) & 1), A1 = __ballot((c1 >> 1) & 1);
      const u64 B0 = __ballot((c0 >> 2) & 1), B1 = __ballot((c1 >> 2) & 1);
      const u64 fi = lane ? (I0 >> lane) | (I1 << (64 - lane)) : I0;
      const u64 fa = lane ? (A0 >> lane) | (A1 << (64 - lane)) : A0;
      const u64 fb = lane ? (B0 >> lane) | (B1 << (64 - lane)) : B0;
      // m-mer starting at my base: base j is digit m-1-j
      const u32 y0 = __brev((u32)fa & mmask) >> (32 - m), y1 = __brev((u32)fb & mmask) >> (32 - m);
      const u32 v = mmer_value(spread16(y0) | (spread16(y1) << 1), m);
      // minimizer of the k-mer starting at my base: min of v over lanes [lane, lane + nbm)
      u32 s1 = min(v, (u32)__shfl_down(v, 1)), s2 = min(s1, (u32)__shfl_down(s1, 2)), s3 = min(s2, (u32)__shfl_down(s2, 4));
      u32 s4 = min(s3, (u32)__shfl_down(s3, 8)), s5 = min(s4, (u32)__shfl_down(s4, 16));
      u32 mini = 0xFFFFFFFFu; int off = 0;
      if (nbm & 32) { mini = min(mini, (u32)__shfl_down(s5, off)); off += 32; }
      if (nbm & 16) { mini = min(mini, (u32)__shfl_down(s4, off)); off += 16; }
      if (nbm & 8) { mini = min(mini, (u32)__shfl_down(s3, off)); off += 8; }
      if (nbm & 4) { mini = min(mini, (u32)__shfl_down(s2, off)); off += 4; }
      if (nbm & 2) { mini = min(mini, (u32)__shfl_down(s1, off)); off += 2; }
      if (nbm & 1) { mini = min(mini, (u32)__shfl_down(v, off)); }
      if (wide) {      // (uniform) lanes whose window [lane, lane + nbm) runs past lane 63
        const u8 c2 = q + 128 < len ? (u8)seq[q + 128] : (u8)'N';
        const u64 A2 = __ballot((c2 >> 1) & 1), B2 = __ballot((c2 >> 2) & 1);
        const u64 fa2 = lane ? (A1 >> lane) | (A2 << (64 - lane)) : A1;
        const u64 fb2 = lane ? (B1 >> lane) | (B2 << (64 - lane)) : B1;
        const u32 z0 = __brev((u32)fa2 & mmask) >> (32 - m), z1 = __brev((u32)fb2 & mmask) >> (32 - m);
        u32 pfx = mmer_value(spread16(z0) | (spread16(z1) << 1), m);      // the m-mer at position p0 + 64 + lane ... min over lanes [0, lane]
        u32 sfx = v;                                                       // ... and this chunk's values: min over lanes [lane, 63]
#pragma unroll
        for (int o2 = 1; o2 < 64; o2 <<= 1) {
          const u32 t = (u32)__shfl_down(sfx, o2), u = (u32)__shfl_up(pfx, o2);
          if (lane + o2 < 64) sfx = min(sfx, t);
          if (lane >= o2) pfx = min(pfx, u);
        }
        const int idx2 = lane + nbm - 65;                                  // the last value of my window, as a lane of the next chunk
        const u32 pm = (u32)__shfl((int)pfx, idx2 >= 0 ? idx2 : 0);
        if (idx2 >= 0) mini = min(sfx, pm);
      }
      const u64 pk = p0 + lane;                                           // my k-mer
      const bool valid = (u32)lane < C && pk < nk && (fi & kmask) == 0;
      u32 pmin_l = (u32)__shfl_up(mini, 1); int pv_l = __shfl_up((int)valid, 1);
      if (lane == 0) { pmin_l = pmin; pv_l = pv; }
      const bool brk = valid && (!pv_l || mini != pmin_l);                 // first k-mer of a run
      const u64 lowmask = (2ULL << lane) - 1;                              // lanes <= mine (lane 63: all)
      const u64 Bm = __ballot(brk) & lowmask;
      const u64 rs = Bm ? p0 + (63 - __clzll(Bm)) : run_start;            // start of my run
      const bool start = valid && (brk || ((u32)(pk - rs) % (u32)maxs) == 0);      // (32 bits: a run lies inside a read, and a 64-bit remainder is ~130 instructions a lane)
      const u64 Sall = __ballot(start);
      const int nvalid = __shfl_down((int)valid, 1), nstart = __shfl_down((int)start, 1);
      const bool owned = (u32)lane < own && pk < nk;
      const bool endf = valid && owned && (pk + 1 == nk || !nvalid || nstart);   // last k-mer of a super-k-mer
      const u64 Em = __ballot(endf);
      u64 ps = 0;
      if ((EMIT || STATS) && endf) {
        const u64 sb = Sall & lowmask;
        ps = sb ? p0 + (63 - __clzll(sb)) : open_start;
      }
      const u32 di = out + __popcll(Em & ((1ULL << lane) - 1));
      if (EMIT && endf) {
        SkDesc d; d.base = (u32)(b0 + ps); d.part = (u16)(pbase + repart[mini]); d.n = (u8)(pk - ps + 1); d.pad = 0;
        if (LB) { if (di < SK_WCAP) wbuf[wave][di] = d; }
        else {
          desc[di] = d;
          if (so.keys) { so.keys[di] = d.part; so.ids[di] = di; so.sizes[di] = 1u + ((u32)k + d.n - 1u + 3u) / 4u; }
          else if (CH && so.ids) so.ids[di] = mini;      // (round 6, the statistics of the sync-free path: the minimizer rides in the slot's word; the k-mers' strands come from the decode)
        }
      }
      int w = 0; u64 ts = 0, xs = 0; u32 rf_s = 0;
      if (STATS) {
        // strand: digit i of the k-mer is base i, digit i of its reverse complement is comp(base k-1-i) = code ^ 2
        const u64 fa_k = fa & kmask, fb_k = fb & kmask;
        const u64 ra = __brevll(fa_k) >> (64 - k), nrb = ~(__brevll(fb_k) >> (64 - k)) & kmask;
        const u64 D = (fa_k ^ ra) | (fb_k ^ nrb);
        if (D) {
          const int i0 = __builtin_ctzll(D);
          const u32 xb = (u32)(fb_k >> i0) & 1u, yb = (u32)(nrb >> i0) & 1u, xa = (u32)(fa_k >> i0) & 1u, ya = (u32)(ra >> i0) & 1u;
          w = xb != yb ? xb < yb : xa < ya;          // forward is the smaller one (KmerCanonical::which, Model.hpp:294; a palindrome counts as reverse)
        }
        if (DEFER) {
          const u64 Wb = __ballot(w != 0);
          if (endf) {      // the strands of my super-k-mer's k-mers, the first one in bit 0 (at most 60 of them: they reach into the previous chunk(s) at most)
            const u32 n = (u32)(pk - ps + 1);
            u64 bits;
            if (ps >= p0) bits = Wb >> (u32)(ps - p0);
            else { const u32 dd = (u32)(p0 - ps); bits = (hist >> (64u - dd)) | (Wb << dd); }
            S.sk_rec[di] = make_ulonglong2((bits & ((1ULL << n) - 1ULL)) | ((u64)(n & 15u) << 60), (u64)(u32)(b0 + ps) | ((u64)mini << 32) | ((u64)(n >> 4) << 62));
          }
          hist = (hist >> own) | (Wb << (64u - own));      // (own = 33 .. 63)
        } else {
        int w_l = __shfl_up(w, 1); if (lane == 0) w_l = pw;
        const bool T = valid && (start || w != w_l);                       // first k-mer of a run of one strand
        const u64 Tm = __ballot(T) & lowmask;
        ts = Tm ? p0 + (63 - __clzll(Tm)) : t_start;
        const bool X = valid && (T || ((u32)(pk - ts) % 5u) == 0);         // first k-mer of a kx-mer (x <= 4)
        const u64 Xall = __ballot(X);
        const int nX = __shfl_down((int)X, 1);
        const bool xend = valid && owned && (endf || nX);
        const u64 xm = Xall & lowmask;
        xs = xm ? p0 + (63 - __clzll(xm)) : x_start;
        u32 rf = 0, rr = 0;                                                // top 4 nucleotides of my k-mer / of its reverse complement
#pragma unroll
        for (int j = 0; j < 4; j++) {
          rf |= ((((u32)(fb >> j) & 1u) << 1) | ((u32)(fa >> j) & 1u)) << (6 - 2 * j);
          rr |= (((((u32)(fb >> (k - 1 - j)) & 1u) ^ 1u) << 1) | ((u32)(fa >> (k - 1 - j)) & 1u)) << (6 - 2 * j);
        }
        rf_s = (u32)__shfl((int)rf, xs >= p0 ? (int)(xs - p0) : 0);
        if (xs < p0) rf_s = rf_open;
        if (xend) {
          const u32 x = (u32)(pk - xs), radix = w ? rf_s : rr;
          if (S.pc) atomicAdd(&S.pc[((pbase + (u32)repart[mini]) * 5u + x) * 256u + radix], 1u);
          if (S.mx) atomicAdd(&S.mx[mini], 1u);
        }
        if (endf) {
          if (S.ms) atomicAdd(&S.ms[sbase + mini], 1u);
          if (S.mk) atomicAdd(&S.mk[sbase + mini], (u32)(pk - ps + 1));
        }
        }
      }
      const u32 ne = (u32)__popcll(Em);
      nsk += ne; out += ne;
      // carry the state of the last owned k-mer
      const u64 rem = nk - p0;
      const int lo = (int)(rem < (u64)own ? rem : (u64)own) - 1;
      pv = __builtin_amdgcn_readlane((int)valid, lo) != 0;
      pmin = (u32)__builtin_amdgcn_readlane((int)mini, lo);
      if (pv) {
        run_start = (u64)(u32)__builtin_amdgcn_readlane((int)(u32)rs, lo) | ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(rs >> 32), lo) << 32);
        const u64 sb = Sall & ((2ULL << lo) - 1);
        if (sb) open_start = p0 + (63 - __clzll(sb));
      }
      if (STATS && !DEFER) {
        pw = __builtin_amdgcn_readlane(w, lo);
        t_start = (u64)(u32)__builtin_amdgcn_readlane((int)(u32)ts, lo) | ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(ts >> 32), lo) << 32);
        x_start = (u64)(u32)__builtin_amdgcn_readlane((int)(u32)xs, lo) | ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(xs >> 32), lo) << 32);
        rf_open = (u32)__builtin_amdgcn_readlane((int)rf_s, lo);
      }
    }
  }
  if (!CH) break;
  }
  if (CH) { if (lane == 0) counts[chunk] = out - out_first; return; }
  r = chunk;
  if (!EMIT && lane == 0) { counts[r] = nsk; if (r == 0) counts[n_seqs] = 0; }      // (every read's entry is written, and the scan's terminating zero: no clear beforehand)
  if (LB) {
    if (lane == 0) wcnt[wave] = min(nsk, SK_WCAP);
    __syncthreads();
    if (wave == 0) {      // the look-back, 64 predecessors at a time (the workgroups in flight have only their counts out: ~2000 of them)
      const u32 bid = bid_s, B = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
      const unsigned long long MASK = (1ULL << 62) - 1;
      if (lane == 0 && bid > 0) __hip_atomic_store(&lk.state[bid], (1ULL << 62) | B, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned long long excl = 0;
      for (long long top = (long long)bid - 1; top >= 0; top -= 64) {
        const long long idx = top - lane;
        unsigned long long st = idx >= 0 ? __hip_atomic_load(&lk.state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ULL << 62);      // (before the first workgroup: prefix 0)
        u32 first_p;
        for (;;) {
          const u64 pm = __ballot((st >> 62) == 2), em = __ballot((st >> 62) == 0);
          first_p = pm ? (u32)__builtin_ctzll(pm) : 64u;
          const u64 need = first_p >= 63u ? ~0ULL : ((2ULL << first_p) - 1ULL);      // the lanes up to the nearest published prefix
          if (!(em & need)) break;
          if ((st >> 62) == 0) st = __hip_atomic_load(&lk.state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned long long v = (u32)lane <= first_p ? (st & MASK) : 0ULL;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += shfl_xor_u64(v, off);
        excl += v;
        if (first_p < 64u) break;
      }
      if (lane == 0) {
        __hip_atomic_store(&lk.state[bid], (2ULL << 62) | (excl + B), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (excl + B > (unsigned long long)lk.cap) { atomicOr(lk.over, 1u); base_s = 0xFFFFFFFFu; } else base_s = (u32)excl;
      }
    }
    __syncthreads();
    u32 at = base_s;
    if (at != 0xFFFFFFFFu) {
      for (u32 w = 0; w < wave; w++) at += wcnt[w];
      const u32 n = wcnt[wave];
      for (u32 i = (u32)lane; i < n; i += 64) {
        const SkDesc d = wbuf[wave][i];
        const u32 di = at + i;
        desc[di] = d; so.keys[di] = d.part; so.ids[di] = di; so.sizes[di] = 1u + ((u32)k + d.n - 1u + 3u) / 4u;
      }
    }
  }
}

// ---- k = 64 ... 127 (Kmer<96> / Kmer<128>: the reference's default KMER_LIST "32 64 96 128", CMakeLists.txt:25-27).  The same wave per
//      read and lane per k-mer position, written for correctness first: a lane's k-mer is up to 127 bits of each plane (two words, from
//      three ballots), its window holds 50 ... 124 m-mers that start in this chunk's 64 positions or the 128 behind them -- the
//      minimum over a window comes from a sparse table over the 192 values (spans of 32 and 64, shuffles across the three
//      registers); a chunk owns 63 k-mers.  Two passes (count, emit), statistics by atomics: the per-partition pass over the sorted
//      descriptors (k_part_stats) carries the strands of at most 60 k-mers a descriptor.
__device__ __forceinline__ u32 sk_at(u32 a, u32 b, int d, int lane)      // the value at position lane + d of the 128 positions (a: 0 .. 63, b: 64 .. 127), 0 <= d < 64
{
  const int src = (lane + d) & 63;
  const u32 x = (u32)__shfl((int)a, src), y = (u32)__shfl((int)b, src);
  return lane + d < 64 ? x : y;
}
__device__ __forceinline__ u32 sk_bit(u64 lo, u64 hi, int i) { return (u32)((i < 64 ? lo >> i : hi >> (i - 64)) & 1ULL); }
// the k bits (lo, hi) in reverse order (bit i <- bit k - 1 - i), 64 <= k <= 127
__device__ __forceinline__ void sk_rev(u64 lo, u64 hi, int k, u64& rlo, u64& rhi)
{
  const u64 RL = __brevll(hi), RH = __brevll(lo);      // the 128 bits reversed: RH:RL
  const int sft = 128 - k;                             // 1 .. 64
  if (sft == 64) { rlo = RH; rhi = 0; }
  else { rlo = (RL >> sft) | (RH << (64 - sft)); rhi = RH >> sft; }
}
template <bool EMIT, bool STATS>
__global__ __launch_bounds__(256)
void k_superk_wide(const char* __restrict__ bases, const u64* __restrict__ offsets, u64 n_seqs,
                   int k, int m, int maxs, const u16* __restrict__ repart,
                   u32* __restrict__ counts, const u32* __restrict__ desc_off, SkDesc* __restrict__ desc, SkStats S, SkSort so)
{
  const int lane = threadIdx.x & 63;
  const u64 r = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= n_seqs) return;
  const u64 b0 = offsets[r], len = offsets[r + 1] - b0;
  u32 nsk = 0;
  if (len >= (u64)k) {
    const char* seq = bases + b0;
    const int nbm = k - m + 1;                       // m-mers per k-mer: 50 .. 124
    const u32 own = 63;                              // the 64th k-mer of a chunk only serves as look-ahead
    const u64 nk = len - (u64)k + 1;
    const u64 himask = k > 64 ? (1ULL << (k - 64)) - 1ULL : 0ULL;      // the k-mer's bits beyond the first 64
    const u32 mmask = (1u << m) - 1;
    u32 out = EMIT ? desc_off[r] : 0;
    bool pv = false; u32 pmin = 0; u64 run_start = 0, open_start = 0;
    int pw = 0; u64 t_start = 0, x_start = 0; u32 rf_open = 0;
    for (u64 p0 = 0; p0 < nk; p0 += own) {
      const u64 q = p0 + lane;
      u8 c[4];
#pragma unroll
      for (int i = 0; i < 4; i++) c[i] = q + 64u * i < len ? (u8)seq[q + 64u * i] : (u8)'N';
      u64 I[3], A[4], B[4];
#pragma unroll
      for (int i = 0; i < 4; i++) { if (i < 3) I[i] = __ballot(!nt_valid(c[i])); A[i] = __ballot((c[i] >> 1) & 1); B[i] = __ballot((c[i] >> 2) & 1); }
      auto fun = [&](u64 x, u64 y) { return lane ? (x >> lane) | (y << (64 - lane)) : x; };
      const u64 fi_lo = fun(I[0], I[1]), fi_hi = fun(I[1], I[2]);
      const u64 fa_lo = fun(A[0], A[1]), fa_hi = fun(A[1], A[2]), fa_2 = fun(A[2], A[3]);
      const u64 fb_lo = fun(B[0], B[1]), fb_hi = fun(B[1], B[2]), fb_2 = fun(B[2], B[3]);
      auto mval = [&](u64 a, u64 b) {      // the m-mer that starts at bit 0 of (a, b): base j is digit m-1-j
        const u32 y0 = __brev((u32)a & mmask) >> (32 - m), y1 = __brev((u32)b & mmask) >> (32 - m);
        return mmer_value(spread16(y0) | (spread16(y1) << 1), m);
      };
      u32 t0 = mval(fa_lo, fb_lo), t1 = mval(fa_hi, fb_hi), t2 = mval(fa_2, fb_2);      // positions lane, 64 + lane, 128 + lane
#pragma unroll
      for (int d = 1; d <= 16; d <<= 1) {      // spans 2, 4, ..., 32
        const u32 n0 = min(t0, sk_at(t0, t1, d, lane)), n1 = min(t1, sk_at(t1, t2, d, lane)), n2 = min(t2, sk_at(t2, 0xFFFFFFFFu, d, lane));
        t0 = n0; t1 = n1; t2 = n2;
      }
      u32 mini;
      if (nbm >= 64) {                         // (uniform) two spans of 64
        const u32 u0 = min(t0, sk_at(t0, t1, 32, lane)), u1 = min(t1, sk_at(t1, t2, 32, lane));
        const int o = nbm - 64;                // 0 .. 60
        mini = o ? min(u0, sk_at(u0, u1, o, lane)) : u0;
      } else mini = min(t0, sk_at(t0, t1, nbm - 32, lane));      // two spans of 32
      const u64 pk = p0 + lane;
      const bool valid = pk < nk && fi_lo == 0 && (fi_hi & himask) == 0;
      u32 pmin_l = (u32)__shfl_up(mini, 1); int pv_l = __shfl_up((int)valid, 1);
      if (lane == 0) { pmin_l = pmin; pv_l = pv; }
      const bool brk = valid && (!pv_l || mini != pmin_l);
      const u64 lowmask = (2ULL << lane) - 1;
      const u64 Bm = __ballot(brk) & lowmask;
      const u64 rs = Bm ? p0 + (63 - __clzll(Bm)) : run_start;
      const bool start = valid && (brk || ((u32)(pk - rs) % (u32)maxs) == 0);      // (32 bits: a run lies inside a read, and a 64-bit remainder is ~130 instructions a lane)
      const u64 Sall = __ballot(start);
      const int nvalid = __shfl_down((int)valid, 1), nstart = __shfl_down((int)start, 1);
      const bool owned = (u32)lane < own && pk < nk;
      const bool endf = valid && owned && (pk + 1 == nk || !nvalid || nstart);
      const u64 Em = __ballot(endf);
      u64 ps = 0;
      if ((EMIT || STATS) && endf) {
        const u64 sb = Sall & lowmask;
        ps = sb ? p0 + (63 - __clzll(sb)) : open_start;
      }
      const u32 di = out + __popcll(Em & ((1ULL << lane) - 1));
      if (EMIT && endf) {
        SkDesc d; d.base = (u32)(b0 + ps); d.part = (u16)repart[mini]; d.n = (u8)(pk - ps + 1); d.pad = 0;
        desc[di] = d;
        if (so.keys) { so.keys[di] = d.part; so.ids[di] = di; so.sizes[di] = 1u + ((u32)k + d.n - 1u + 3u) / 4u; }
      }
      int w = 0; u64 ts = 0, xs = 0; u32 rf_s = 0;
      if (STATS) {
        const u64 fa_h = fa_hi & himask, fb_h = fb_hi & himask;
        u64 ra_lo, ra_hi, rb_lo, rb_hi;
        sk_rev(fa_lo, fa_h, k, ra_lo, ra_hi); sk_rev(fb_lo, fb_h, k, rb_lo, rb_hi);
        const u64 nrb_lo = ~rb_lo, nrb_hi = ~rb_hi & himask;
        const u64 D_lo = (fa_lo ^ ra_lo) | (fb_lo ^ nrb_lo), D_hi = (fa_h ^ ra_hi) | (fb_h ^ nrb_hi);
        if (D_lo | D_hi) {
          const int i0 = D_lo ? __builtin_ctzll(D_lo) : 64 + __builtin_ctzll(D_hi);
          const u32 xb = sk_bit(fb_lo, fb_h, i0), yb = sk_bit(nrb_lo, nrb_hi, i0), xa = sk_bit(fa_lo, fa_h, i0), ya = sk_bit(ra_lo, ra_hi, i0);
          w = xb != yb ? xb < yb : xa < ya;
        }
        int w_l = __shfl_up(w, 1); if (lane == 0) w_l = pw;
        const bool T = valid && (start || w != w_l);
        const u64 Tm = __ballot(T) & lowmask;
        ts = Tm ? p0 + (63 - __clzll(Tm)) : t_start;
        const bool X = valid && (T || ((u32)(pk - ts) % 5u) == 0);
        const u64 Xall = __ballot(X);
        const int nX = __shfl_down((int)X, 1);
        const bool xend = valid && owned && (endf || nX);
        const u64 xm = Xall & lowmask;
        xs = xm ? p0 + (63 - __clzll(xm)) : x_start;
        u32 rf = 0, rr = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          rf |= ((((u32)(fb_lo >> j) & 1u) << 1) | ((u32)(fa_lo >> j) & 1u)) << (6 - 2 * j);
          rr |= (((sk_bit(fb_lo, fb_h, k - 1 - j) ^ 1u) << 1) | sk_bit(fa_lo, fa_h, k - 1 - j)) << (6 - 2 * j);
        }
        rf_s = (u32)__shfl((int)rf, xs >= p0 ? (int)(xs - p0) : 0);
        if (xs < p0) rf_s = rf_open;
        if (xend) {
          const u32 x = (u32)(pk - xs), radix = w ? rf_s : rr;
          if (S.pc) atomicAdd(&S.pc[((u32)repart[mini] * 5u + x) * 256u + radix], 1u);
          if (S.mx) atomicAdd(&S.mx[mini], 1u);
        }
        if (endf) {
          if (S.ms) atomicAdd(&S.ms[mini], 1u);
          if (S.mk) atomicAdd(&S.mk[mini], (u32)(pk - ps + 1));
        }
      }
      const u32 ne = (u32)__popcll(Em);
      nsk += ne; out += ne;
      const u64 rem = nk - p0;
      const int lo = (int)(rem < (u64)own ? rem : (u64)own) - 1;
      pv = __builtin_amdgcn_readlane((int)valid, lo) != 0;
      pmin = (u32)__builtin_amdgcn_readlane((int)mini, lo);
      if (pv) {
        run_start = (u64)(u32)__builtin_amdgcn_readlane((int)(u32)rs, lo) | ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(rs >> 32), lo) << 32);
        const u64 sb = Sall & ((2ULL << lo) - 1);
        if (sb) open_start = p0 + (63 - __clzll(sb));
      }
      if (STATS) {
        pw = __builtin_amdgcn_readlane(w, lo);
        t_start = (u64)(u32)__builtin_amdgcn_readlane((int)(u32)ts, lo) | ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(ts >> 32), lo) << 32);
        x_start = (u64)(u32)__builtin_amdgcn_readlane((int)(u32)xs, lo) | ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(xs >> 32), lo) << 32);
        rf_open = (u32)__builtin_amdgcn_readlane((int)rf_s, lo);
      }
    }
  }
  if (!EMIT && lane == 0) { counts[r] = nsk; if (r == 0) counts[n_seqs] = 0; }
}

__global__ void k_superk_gather_sizes(const u32* __restrict__ ids, const u32* __restrict__ sizes_unsorted, u32 n, u64* __restrict__ sizes_sorted)
{
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sizes_sorted[i] = sizes_unsorted[ids[i]];
}
// partition-sorted order: (k-mers << 32 | bytes) per record, so one scan yields both prefixes (both totals < 2^32)
__global__ void k_superk_gather_sizes2(const u32* __restrict__ ids, const u32* __restrict__ sizes_unsorted, const SkDesc* __restrict__ desc,
                                       u32 n, u64* __restrict__ sizes_sorted, u32* __restrict__ base_sorted /* or null */)
{
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const u32 j = ids[i]; const SkDesc d = desc[j]; sizes_sorted[i] = ((u64)d.n << 32) | sizes_unsorted[j]; if (base_sorted) base_sorted[i] = d.base; }
  else if (i == n) sizes_sorted[n] = 0;      // (the scan runs over n + 1 entries: launch with more than n threads)
}
// prefix (k-mers << 32 | bytes) at the first record of every partition (nb_parts + 1 entries) + that record's index
__global__ void k_superk_part_bounds(const u16* __restrict__ part_sorted, u32 n, u32 nb_parts, const u64* __restrict__ prefix,
                                     u64* __restrict__ part_prefix, u32* __restrict__ part_first, u64* __restrict__ zeroed, u32 n_zeroed)
{
  const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > nb_parts) return;
  if (p < n_zeroed) zeroed[p] = 0;      // (the counters of k_minim_sparse, one per sample, which runs behind this kernel)
  u32 lo = 0, hi = n;
  while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (part_sorted[mid] < p) lo = mid + 1; else hi = mid; }
  part_prefix[p] = prefix[lo];
  part_first[p] = lo;
}

// What SuperKStorageWriter::SaveInfoFile reports per partition (io/superk_storage.hpp:205-225, 328-340; the file is saved before
// the final flush): k-mers since the last full 32 KB block, bytes of the blocks flushed so far.  A thread per partition walks
// its records' sizes (prefix differences).
__global__ void k_superk_info(const u32* __restrict__ part_first, const u64* __restrict__ prefix, u32 nb_parts, u64* __restrict__ info, const SkfCtl* __restrict__ ctl = nullptr)
{
  const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nb_parts) return;
  if (ctl && ctl->status) return;      // (the sync-free path handed the call back: the prefix array was not written, or not all of it)
  // a block = the longest run of whole records of at most 32768 bytes: its end is a binary search in the byte prefix (a step per
  // block, not per record); the last run is still in the writer's buffer when the info file is saved
  const u32 i1 = part_first[p + 1];
  u32 s = part_first[p];
  u64 flushed = 0, km = 0;
  while (s < i1) {
    const u32 b0 = (u32)prefix[s];
    u32 lo = s + 1, hi = i1;                       // largest e in (s, i1] with bytes[s, e) <= 32768
    while (lo < hi) { const u32 mid = lo + ((hi - lo + 1) >> 1); if ((u32)prefix[mid] - b0 <= 32768u) lo = mid; else hi = mid - 1; }
    const u32 e = lo;
    if (e < i1) flushed += (u64)((u32)prefix[e] - b0) + 4;
    else km = (prefix[e] >> 32) - (prefix[s] >> 32);
    s = e;
  }
  info[2 * p] = km; info[2 * p + 1] = flushed;
}

// four ASCII bases (one unaligned dword) -> four 2-bit codes in one byte, first base in the low bits
__device__ __forceinline__ u32 pack4(u32 w)
{
  w = (w >> 1) & 0x03030303u;
  return (w | (w >> 6) | (w >> 12) | (w >> 18)) & 0xFFu;
}
__device__ __forceinline__ u32 load4(const char* p) { u32 w; __builtin_memcpy(&w, p, 4); return w; }

__global__ void k_superk_pack(const char* __restrict__ bases, const SkDesc* __restrict__ desc, const u32* __restrict__ ids,
                              const u64* __restrict__ byte_off, u32 n, int k, u8* __restrict__ out)
{
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const SkDesc d = desc[ids[i]];
  const char* seq = bases + d.base;
  u8* o = out + (u32)byte_off[i];
  const int ndig = k + d.n - 1;
  o[0] = d.n;
  // Bytes are made of four digits; digit d < k is base[k-1-d] (the first k-mer, last base first), digit d >= k is
  // base[d].  A byte that lies wholly in one of the two parts comes from ONE 4-byte load of the read; only the byte
  // that straddles digit k and the last, partial byte are assembled digit by digit.
  for (int by = 0; by * 4 < ndig; by++) {
    const int d0 = by * 4;
    u32 v;
    if (d0 + 3 < k) v = pack4(__builtin_bswap32(load4(seq + (k - d0 - 4))));     // bases k-d0-4 .. k-d0-1, reversed
    else if (d0 >= k && d0 + 3 < ndig) v = pack4(load4(seq + d0));
    else {
      v = 0;
      for (int q = 0; q < 4; q++) {
        const int dg = d0 + q;
        if (dg >= ndig) break;
        const int bi = dg < k ? (k - 1 - dg) : dg;
        v |= (((u32)(u8)seq[bi] >> 1) & 3u) << (2 * q);
      }
    }
    o[1 + by] = (u8)v;
  }
}

// ---- the PartiInfo<5> statistics of a sample from its SORTED descriptors (DEFER): a workgroup per partition ------------------------
// fill_partitions.hpp:67-102 counts, per super-k-mer, its minimizer's super-k-mers and k-mers and, per kx-mer (a run of at most 5
// k-mers of one strand), the (partition, run length, radix) counter.  Walking the reads (k_superk_wave<.., STATS>) that is three
// device-scope atomics per super-k-mer into 9 MB of tables that must be cleared and compacted again: a third of the count stage.
// Here a partition's descriptors are one run (the sort by partition that the stage makes anyway); its 5 x 256 counters and the
// minimizers that map to it (a few thousand of 4^m: an open-addressing table) live in LDS, the counters leave as plain stores (no
// clear beforehand), the minimizers as the {minimizer, super-k-mers, k-mers} triples kmx_superk_raw::minim_sparse wants.  A
// minimizer that finds no slot within PS_PROBE steps goes the old way (dense tables + k_minim_sparse, which n_out[1] switches on).
constexpr int PS_TPB = 1024;
constexpr u32 PS_H = 4096, PS_PROBE = 48, PS_EMPTY = 0xFFFFFFFFu;
__global__ __launch_bounds__(PS_TPB)
void k_part_stats(const u32* __restrict__ ids, const u32* __restrict__ part_first, u32 part0, const ulonglong2* __restrict__ sk_rec, const char* __restrict__ bases, int k,
                  u32* __restrict__ pc, u32* __restrict__ ms_dense, u32* __restrict__ mk_dense, u32* __restrict__ out, u32 cap, u32* __restrict__ n_out,
                  const u64* __restrict__ boff = nullptr, const u32* __restrict__ sbase = nullptr, const u32* __restrict__ mini_sorted = nullptr, const u64* __restrict__ strand = nullptr,
                  const SkfCtl* __restrict__ ctl = nullptr)
{   // (round 6, sk_rec == null: the sorted records' own arrays -- prefix (k-mers << 32 | bytes), first base, minimizer -- read in order, and
    //  the k-mers' strands as the decode left them, a bit per k-mer of the batch: no 16-byte record per super-k-mer, no gather)
  __shared__ u32 tab[5 * 256];
  __shared__ u32 hkey[PS_H], hms[PS_H], hmk[PS_H];
  __shared__ u32 wtot[PS_TPB / 64], out_base;
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 p = part0 + blockIdx.x;
  // (the sync-free path: a status bit means the sorted arrays were not written, or not all of them -- the call is repeated the old way,
  //  statistics included; found by scripts/fuzz_count.py: k = 12, m = 11, a super-k-mer a k-mer, more records than the arrays were sized for)
  if (ctl && ctl->status) return;
  for (u32 j = tid; j < 5 * 256; j += PS_TPB) tab[j] = 0;
  for (u32 j = tid; j < PS_H; j += PS_TPB) { hkey[j] = PS_EMPTY; hms[j] = 0; hmk[j] = 0; }
  __syncthreads();
  const u32 i1 = part_first[p + 1];
  for (u32 i = part_first[p] + tid; i < i1; i += PS_TPB) {
    ulonglong2 rec;
    if (sk_rec) rec = sk_rec[ids[i]];
    else {
      const u64 p0 = boff[i], p1 = boff[i + 1];
      const u32 ko = (u32)(p0 >> 32), nn = (u32)(p1 >> 32) - ko;
      const u64 w0 = strand[ko >> 6], w1 = strand[(ko >> 6) + 1];
      const u32 sft = ko & 63u;
      const u64 b = sft ? (w0 >> sft) | (w1 << (64u - sft)) : w0;
      rec.x = (b & ((1ULL << nn) - 1ULL) & ((1ULL << 60) - 1ULL)) | ((u64)(nn & 15u) << 60);
      rec.y = (u64)sbase[i] | ((u64)mini_sorted[i] << 32) | ((u64)(nn >> 4) << 62);
    }
    const u32 mini = (u32)(rec.y >> 32) & 0x3FFFFFFFu, n = (u32)(rec.x >> 60) | ((u32)(rec.y >> 62) << 4);
    const u64 bits = rec.x & ((1ULL << 60) - 1ULL);
    {
      u32 h = (mini * 2654435761u) >> 20;      // (PS_H = 2^12)
      bool done = false;
      for (u32 t = 0; t < PS_PROBE; t++) {
        const u32 prev = atomicCAS(&hkey[h], PS_EMPTY, mini);
        if (prev == PS_EMPTY || prev == mini) { atomicAdd(&hms[h], 1u); atomicAdd(&hmk[h], n); done = true; break; }
        h = (h + 1u) & (PS_H - 1u);
      }
      if (!done) { atomicAdd(&ms_dense[mini], 1u); atomicAdd(&mk_dense[mini], n); n_out[1] = 1u; }
    }
    const char* const seq = bases + (u32)rec.y;
    for (u32 t = 0; t < n;) {
      const u32 w = (u32)(bits >> t) & 1u;
      const u64 same = (w ? bits : ~bits) >> t;                       // ones: the k-mers from t on that share its strand
      const u32 run = min((u32)(~same ? __builtin_ctzll(~same) : 64), n - t);
      for (u32 s0 = 0; s0 < run; s0 += 5) {
        const u32 len = min(5u, run - s0), first = t + s0, last = first + len - 1u;
        u32 radix;
        if (w) { const u32 c = (load4(seq + first) >> 1) & 0x03030303u; radix = ((c & 3u) << 6) | (((c >> 8) & 3u) << 4) | (((c >> 16) & 3u) << 2) | (c >> 24); }
        else { const u32 c = ((load4(seq + last + (u32)k - 4u) >> 1) & 0x03030303u) ^ 0x02020202u; radix = ((c >> 24) << 6) | (((c >> 16) & 3u) << 4) | (((c >> 8) & 3u) << 2) | (c & 3u); }
        atomicAdd(&tab[(len - 1u) * 256u + radix], 1u);
      }
      t += run;
    }
  }
  __syncthreads();
  for (u32 j = tid; j < 5 * 256; j += PS_TPB) pc[(u64)p * 1280u + j] = tab[j];
  // the minimizers that occur: slots tid * 8 .. tid * 8 + 7, their places by a scan over the workgroup, ONE global atomic
  u32 c = 0;
#pragma unroll
  for (u32 q = 0; q < PS_H / PS_TPB; q++) c += hkey[tid * (PS_H / PS_TPB) + q] != PS_EMPTY ? 1u : 0u;
  const u32 incl = wave_incl_scan(c, (int)lane);
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  if (tid == 0) { u32 t = 0; for (u32 q = 0; q < PS_TPB / 64; q++) t += wtot[q]; out_base = t ? atomicAdd(n_out, t) : 0u; }
  __syncthreads();
  u32 pos = out_base + incl - c;
  for (u32 q = 0; q < wave; q++) pos += wtot[q];
#pragma unroll
  for (u32 q = 0; q < PS_H / PS_TPB; q++) {
    const u32 j = tid * (PS_H / PS_TPB) + q;
    if (hkey[j] != PS_EMPTY) {
      if (pos < cap) { out[3 * (u64)pos] = hkey[j]; out[3 * (u64)pos + 1] = hms[j]; out[3 * (u64)pos + 2] = hmk[j]; }
      pos++;
    }
  }
}

#include "superk_fast.hpp"

}  // namespace kmx

using namespace kmx;

// the minimizers that occur: {minimizer, super-k-mers, k-mers} triples, in no particular order (kmx_superk_raw::minim_sparse)
// (16 table entries per thread: a wave takes 1024 of them and claims its output with ONE atomic -- with an entry per thread the 16 000
//  waves of a 4^10 table queued up on that one word: 40 us for 4 MB)
__global__ void k_minim_sparse(u32* __restrict__ ms, u32* __restrict__ mk, u64 nm, u32* __restrict__ out, u32 cap, u32* __restrict__ n_out, int clean, int only_if_flag)
{
  if (only_if_flag && n_out[1] == 0) return;      // (behind k_part_stats: the tables were touched only when a partition's minimizers did not fit its LDS)
  const u64 i0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 16u;      // (4^m is a multiple of 16: m >= 4)
  const u32 lane = threadIdx.x & 63u;
  u32 v[16]; u32 c = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint4 x = i0 + 4 * q < nm ? reinterpret_cast<const uint4*>(ms + i0)[q] : make_uint4(0, 0, 0, 0);
    v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
  }
#pragma unroll
  for (int q = 0; q < 16; q++) c += v[q] != 0 ? 1u : 0u;
  const u32 incl = wave_incl_scan(c, (int)lane);
  const u32 total = (u32)__shfl((int)incl, 63);
  if (!total) return;
  u32 base = 0;
  if (lane == 0) base = atomicAdd(n_out, total);
  base = (u32)__shfl((int)base, 0);
  u32 pos = base + incl - c;
#pragma unroll
  for (int q = 0; q < 16; q++)
    if (v[q]) {
      if (pos < cap) { out[3 * (u64)pos] = (u32)(i0 + q); out[3 * (u64)pos + 1] = v[q]; out[3 * (u64)pos + 2] = mk[i0 + q]; }
      pos++;
      if (clean) { ms[i0 + q] = 0; mk[i0 + q] = 0; }      // (the tables are the context's: left as the next call wants them)
    }
}

// stats tables of one call (u32 on the device): added to the caller's u64 arrays (kmx_superk_stats), or copied as they are into
// the caller's u32 buffers (kmx_superk_raw: no host arithmetic, one synchronisation)
struct StatsDev {
  SkStats S{nullptr, nullptr, nullptr, nullptr, nullptr};
  bool want_defer = false, deferred = false;      // DEFER: asked for by the caller (the two-pass path) / granted (kmx_superk_raw in sparse form: the context's tables)
  kmx_superk_stats* dst = nullptr; kmx_superk_raw* raw = nullptr;      // raw: one per sample of the call ([ns])
  u32 nb_parts = 0; u64 nm = 0;             // partitions of the call (samples x partitions), per-minimizer entries of the call (samples x 4^m)
  u32 ns = 1; u64 nm1 = 0; u32 parts1 = 0;  // samples of the call, 4^m, partitions per sample
  u32* d_sp = nullptr; std::vector<u32> sp_cap, sp_off;      // kmx_superk_raw::minim_sparse: the triples on the device, how many fit per sample, where a sample's start
  u32* blk_ = nullptr; size_t words_ = 0;   // the tables' block (cleared again when a pass over the reads is repeated)
  bool persistent = false; kmx_ctx* pctx = nullptr;
  hipError_t clear(hipStream_t s) const { return blk_ ? hipMemsetAsync(blk_, 0, words_ * 4, s) : hipSuccess; }
  int alloc(kmx_ctx* ctx, kmx_superk_stats* st, kmx_superk_raw* rw, u32 P, u64 nminim, std::vector<void*>& blocks, hipStream_t s, u32 n_samples = 1) {
    dst = st; raw = rw; ns = std::max(1u, n_samples); nb_parts = P; parts1 = P / ns; nm1 = nminim; nm = nminim * ns;
    if (!st && !rw) return KMX_OK;
    const bool w_pc = (st && st->part_counters) || (rw && rw->part_radix), w_ms = (st && st->minim_superks) || (rw && (rw->minim_superks || rw->minim_sparse)),
               w_mk = (st && st->minim_kmers) || (rw && (rw->minim_kmers || rw->minim_sparse)), w_mx = st && st->minim_kxmers;
    // the tables lie in one block, cleared with one call
    const size_t n_pc = w_pc ? (size_t)P * 1280 : 0, n_m = (size_t)nm, words = n_pc + ((size_t)w_ms + w_mk + w_mx) * n_m;
    if (!words) return KMX_OK;
    // kmx_superk_raw in sparse form and nothing else: the context's own tables, of which only the partitions' counters need a clear
    persistent = rw && rw->minim_sparse && !st && w_pc;
    u32* blk;
    if (persistent) {
      if (ctx->d_stat && ctx->stat_cap < words) { (void)hipStreamSynchronize(s); (void)hipFree(ctx->d_stat); ctx->d_stat = nullptr; }
      if (!ctx->d_stat) {
        if (hipMalloc((void**)&ctx->d_stat, words * 4) != hipSuccess) { ctx->d_stat = nullptr; return ctx->fail(KMX_E_NOMEM, "superk: statistics allocation failed"); }
        ctx->stat_cap = words; ctx->stat_dirty = true;
      }
      if (ctx->stat_parts != P || ctx->stat_nm != nm) ctx->stat_dirty = true;      // (another layout: the counters of the last call lie elsewhere)
      ctx->stat_parts = P; ctx->stat_nm = nm;
      blk = ctx->d_stat;
      deferred = want_defer && !w_mx;
      // (deferred: k_part_stats writes every counter of every partition and leaves the per-minimizer tables as they are -- zero -- or cleans them)
      const size_t n_clear = ctx->stat_dirty ? ctx->stat_cap : deferred ? 0 : n_pc;
      if (n_clear && hipMemsetAsync(blk, 0, n_clear * 4, s) != hipSuccess) return ctx->fail(KMX_E_HIP, "superk: statistics memset failed");
      ctx->stat_dirty = true;      // (until k_minim_sparse has run behind the kernel that fills the tables)
      pctx = ctx;
    } else {
      blk = (u32*)ctx->dalloc(words * 4); blocks.push_back(blk);
      if (!blk) return ctx->fail(KMX_E_NOMEM, "superk: statistics allocation failed");
      if (hipMemsetAsync(blk, 0, words * 4, s) != hipSuccess) return ctx->fail(KMX_E_HIP, "superk: statistics memset failed");
    }
    blk_ = blk; words_ = words;
    u32* at = blk;
    if (w_pc) { S.pc = at; at += n_pc; }
    if (w_ms) { S.ms = at; at += n_m; }
    if (w_mk) { S.mk = at; at += n_m; }
    if (w_mx) { S.mx = at; at += n_m; }
    if (rw && rw->minim_sparse) {
      size_t tot = 0;
      for (u32 i = 0; i < ns; i++) {
        if (!rw[i].minim_sparse) return ctx->fail(KMX_E_INVAL, "kmx_superk_raw: the samples of a call take their statistics in the same form");
        sp_off.push_back((u32)tot);
        sp_cap.push_back((u32)std::min<u64>(std::min<u64>(rw[i].minim_sparse_cap, 0xFFFFFF0ULL), nminim));
        tot += sp_cap.back();
      }
      d_sp = (u32*)ctx->dalloc(tot * 12 + 16); blocks.push_back(d_sp);
      if (!d_sp) return ctx->fail(KMX_E_NOMEM, "superk: statistics allocation failed");
    }
    return KMX_OK;
  }
  bool any() const { return S.pc || S.ms || S.mk || S.mx; }
  // kmx_superk_raw, behind the kernel that fills the tables: the minimizers that occur are compacted on the device, sample by sample
  // (their numbers land in d_n[sample], which the caller has zeroed on the stream and downloads with its own results) ...
  void launch_sparse(u64* d_n, hipStream_t s) const {
    if (!d_sp) return;
    for (u32 i = 0; i < ns; i++)
      hipLaunchKernelGGL(k_minim_sparse, dim3((unsigned)((nm1 / 16 + 255) / 256)), dim3(256), 0, s, S.ms + (size_t)i * nm1, S.mk + (size_t)i * nm1, nm1,
                         d_sp + (size_t)sp_off[i] * 3, sp_cap[i], reinterpret_cast<u32*>(d_n + i), persistent ? 1 : 0, deferred ? 1 : 0);
  }
  // DEFER: the statistics from the sorted descriptors, a launch per sample (d_n[sample]: low word the triples, high word "the dense tables were used")
  void launch_part_stats(const u32* ids_sorted, const u32* part_first, const char* bases, u32 k, u64* d_n, hipStream_t s,
                         const u64* boff = nullptr, const u32* sbase = nullptr, const u32* mini_sorted = nullptr, const u64* strand = nullptr, const SkfCtl* ctl = nullptr) const {
    if (!deferred) return;
    for (u32 i = 0; i < ns; i++)
      hipLaunchKernelGGL(k_part_stats, dim3(parts1), dim3(PS_TPB), 0, s, ids_sorted, part_first, i * parts1, (const ulonglong2*)S.sk_rec, bases, (int)k,
                         S.pc, S.ms + (size_t)i * nm1, S.mk + (size_t)i * nm1, d_sp + (size_t)sp_off[i] * 3, sp_cap[i], reinterpret_cast<u32*>(d_n + i),
                         boff, sbase, mini_sorted, strand, ctl);
  }
  void compacted() const { if (persistent && pctx && d_sp) pctx->stat_dirty = false; }      // (the caller has waited for k_minim_sparse: the per-minimizer tables are zero again)
  // ... and once those numbers are on the host, the copies into the caller's (page-locked) buffers are queued: no synchronisation
  // here, the caller's next one covers them
  int finish_raw(kmx_ctx* ctx, const u64* n_sparse /* [ns] */, hipStream_t s) {
    if (!raw) return KMX_OK;
    hipError_t e = hipSuccess;
    for (u32 i = 0; i < ns && e == hipSuccess; i++) {
      kmx_superk_raw& R = raw[i];
      if (R.part_radix) e = hipMemcpyAsync(R.part_radix, S.pc + (size_t)i * parts1 * 1280, (size_t)parts1 * 1280 * 4, hipMemcpyDeviceToHost, s);
      if (R.minim_sparse) {
        const u32 n = (u32)n_sparse[i];
        if (n > sp_cap[i]) return ctx->fail(KMX_E_INVAL, "kmx_superk_raw: more minimizers occur than minim_sparse_cap");
        R.minim_sparse_n = n;
        if (n && e == hipSuccess) e = hipMemcpyAsync(R.minim_sparse, d_sp + (size_t)sp_off[i] * 3, (size_t)n * 12, hipMemcpyDeviceToHost, s);
      } else {
        if (e == hipSuccess && R.minim_superks) e = hipMemcpyAsync(R.minim_superks, S.ms + (size_t)i * nm1, nm1 * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && R.minim_kmers) e = hipMemcpyAsync(R.minim_kmers, S.mk + (size_t)i * nm1, nm1 * 4, hipMemcpyDeviceToHost, s);
      }
    }
    return e == hipSuccess ? KMX_OK : ctx->fail(KMX_E_HIP, std::string("superk statistics: ") + hipGetErrorString(e));
  }
  // after the kernel: download and accumulate (PartiInfo::incKmer_and_rad / incSuperKmer_per_minimBin / incKxmer_per_minimBin)
  int collect(kmx_ctx* ctx, hipStream_t s) {
    if (!dst || !any()) return KMX_OK;
    u32* h = (u32*)ctx->halloc(std::max<size_t>((size_t)nb_parts * 1280, nm) * 4);      // pinned staging: the tables come back at PCIe speed
    if (!h) return ctx->fail(KMX_E_NOMEM, "superk statistics: host staging allocation failed");
    struct Rel { kmx_ctx* c; void* p; ~Rel() { c->hfree(p); } } rel{ctx, h};
    auto pull = [&](const u32* d, size_t n) -> int {
      hipError_t e = hipMemcpyAsync(h, d, n * 4, hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      return e == hipSuccess ? KMX_OK : ctx->fail(KMX_E_HIP, std::string("superk statistics: ") + hipGetErrorString(e));
    };
    int rc;
    if (S.pc && dst->part_counters) {
      if ((rc = pull(S.pc, (size_t)nb_parts * 1280)) != KMX_OK) return rc;
      for (u32 p = 0; p < nb_parts; p++) {
        uint64_t* o = dst->part_counters + (size_t)p * KMX_PINFO_STRIDE;
        for (u32 x = 0; x < 5; x++) for (u32 r = 0; r < 256; r++) {
          const u64 c = h[((size_t)p * 5 + x) * 256 + r];
          if (!c) continue;
          o[0] += c * (x + 1); o[1] += c; o[2 + x * 256 + r] += c;
        }
      }
    }
    if (S.ms && dst->minim_superks) { if ((rc = pull(S.ms, nm)) != KMX_OK) return rc; for (u64 i = 0; i < nm; i++) { dst->minim_superks[i] += h[i]; dst->nb_superk += h[i]; } }
    if (S.mk && dst->minim_kmers) { if ((rc = pull(S.mk, nm)) != KMX_OK) return rc; for (u64 i = 0; i < nm; i++) dst->minim_kmers[i] += h[i]; }
    if (S.mx && dst->minim_kxmers) { if ((rc = pull(S.mx, nm)) != KMX_OK) return rc; for (u64 i = 0; i < nm; i++) dst->minim_kxmers[i] += h[i]; }
    return KMX_OK;
  }
};

// several samples in one call: the reads of sample i are [seq_first[i], seq_first[i + 1]) of `offsets` (rebased into one run by the
// caller), its bases bases[i], which the concatenation holds from base_first[i] on; `parts` partitions per sample (nb_parts = n * parts)
struct SkSegs { u32 n; const char* const* bases; const u64* base_first; const u32* seq_first; u32 parts; };

// ---- the chain of count calls per GPU (kmx_host.hpp) ----
#include <mutex>
namespace {
constexpr int CHAIN_DEV = 64;
std::mutex g_chain_mu[CHAIN_DEV];
hipEvent_t g_chain_tail[CHAIN_DEV] = {};
thread_local bool tl_chain_held = false;
// (off since the count path queues a whole call without a host wait inside: two workers' calls side by side then fill each other's gaps --
//  1000 x 5 Mbp count in 1.64-1.68 s against 1.79-1.80 s chained, three interleaved passes on one box; KMX_COUNT_CHAIN=1 chains them)
bool chain_on() { const char* e = getenv("KMX_COUNT_CHAIN"); return e && !strcmp(e, "1"); }      // (read per call)
}
void kmx_count_chain_begin(kmx_ctx* ctx)
{
  if (!chain_on() || ctx->device < 0 || ctx->device >= CHAIN_DEV) return;
  if (!ctx->ev_chain && hipEventCreateWithFlags(&ctx->ev_chain, hipEventDisableTiming) != hipSuccess) { ctx->ev_chain = nullptr; (void)hipGetLastError(); return; }
  g_chain_mu[ctx->device].lock(); tl_chain_held = true;
  hipEvent_t tail = g_chain_tail[ctx->device];
  if (tail && tail != ctx->ev_chain) (void)hipStreamWaitEvent(ctx->stream, tail, 0);      // (its own last call is in front of this one on the stream anyway)
}
void kmx_count_chain_end(kmx_ctx* ctx)
{
  if (!tl_chain_held) return;
  if (hipEventRecord(ctx->ev_chain, ctx->stream) == hipSuccess) g_chain_tail[ctx->device] = ctx->ev_chain; else (void)hipGetLastError();
  tl_chain_held = false; g_chain_mu[ctx->device].unlock();
}
void kmx_count_chain_forget(kmx_ctx* ctx)
{
  if (!ctx->ev_chain || ctx->device < 0 || ctx->device >= CHAIN_DEV) return;
  std::lock_guard<std::mutex> lk(g_chain_mu[ctx->device]);
  if (g_chain_tail[ctx->device] == ctx->ev_chain) g_chain_tail[ctx->device] = nullptr;
  (void)hipEventDestroy(ctx->ev_chain); ctx->ev_chain = nullptr;
}

// KMX_COUNT_PHASES=1: where a kmx_count_reads_dev call spends its HOST time (sums over calls, a line every 200 calls on stderr)
struct PhaseClock {
  static constexpr int NP = 8;
  bool on; std::chrono::steady_clock::time_point t;
  PhaseClock() : on(getenv("KMX_COUNT_PHASES") != nullptr), t(std::chrono::steady_clock::now()) {}
  void mark(int i) {
    if (!on) return;
    static std::atomic<unsigned long long> ns[NP]; static std::atomic<unsigned> calls{0};
    const auto n = std::chrono::steady_clock::now();
    ns[i] += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(n - t).count(); t = n;
    if (i == NP - 1 && ++calls % 200 == 0) {
      const double c = (double)calls.load();
      fprintf(stderr, "[kmx count phases] %u calls, ms a call: setup+digest %.3f | uploads queued %.3f | split queued %.3f | count queued %.3f | statistics waited %.3f | stream waited %.3f | lists packed %.3f | tail %.3f\n",
              calls.load(), ns[0] / c / 1e6, ns[1] / c / 1e6, ns[2] / c / 1e6, ns[3] / c / 1e6, ns[4] / c / 1e6, ns[5] / c / 1e6, ns[6] / c / 1e6, ns[7] / c / 1e6);
    }
  }
};
PhaseClock* g_phase_clock = nullptr;      // (the count's half marks its phases through this: set for the duration of a call, per thread below)
static thread_local PhaseClock* tl_phase = nullptr;
void kmx_phase_mark(int i) { if (tl_phase) tl_phase->mark(i); }

static int superk_impl(kmx_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n_seqs,
                       uint32_t k, uint32_t m, const uint16_t* repart, uint32_t nb_parts,
                       uint8_t** out_bytes, uint64_t* out_len, uint64_t* out_kmers, kmx_superk_stats* stats,
                       bool sampling = false, uint64_t budget = 0, uint64_t* n_used = nullptr, uint64_t* n_superk = nullptr,
                       const kmx_count_req* creq = nullptr, bool streams_to_host = true, uint64_t* superk_info = nullptr, kmx_superk_raw* raw = nullptr,
                       const SkSegs* segs = nullptr)
{
  if (!ctx) return KMX_E_INVAL;
  const bool want_streams = out_bytes != nullptr;
  if (!offsets || (!repart && want_streams) || (want_streams && (!out_len || !out_kmers)) || (!want_streams && !stats) || nb_parts == 0 || nb_parts > 65535)
    return ctx->fail(KMX_E_INVAL, "kmx_superk_partition: bad argument");
  if (k < 8 || k > 127 || m < 4 || m > 15 || m > k) return ctx->fail(KMX_E_UNSUPPORTED, "k outside 8..127 or minimizer size outside 4..15");
  const bool wide_k = k >= 64;      // Kmer<96> / Kmer<128>: k_superk_wide, two passes, statistics by atomics, counts from the record stream
  if (wide_k && segs) return ctx->fail(KMX_E_UNSUPPORTED, "k >= 64: one sample per call");
  if (want_streams) for (u32 p = 0; p < nb_parts; p++) { out_bytes[p] = nullptr; out_len[p] = 0; out_kmers[p] = 0; }
  if (n_used) *n_used = 0;
  if (n_superk) *n_superk = 0;
  if (n_seqs == 0) { if (want_streams) for (u32 p = 0; p < nb_parts; p++) out_bytes[p] = (uint8_t*)malloc(1); return KMX_OK; }
  const u64 total_bases = offsets[n_seqs];
  if (total_bases >= 0xFFFFFF00ULL || n_seqs >= 0x7FFFFFFFULL) return ctx->fail(KMX_E_UNSUPPORTED, "batch of 4 Gbases or more: split it");
  if (total_bases && !bases && !segs) return ctx->fail(KMX_E_INVAL, "null bases");
  const u32 n_smp = segs ? segs->n : 1u;
  KMX_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // Type::getSize() of the k-mer type the reference instantiates: k < 32 -> MAX_K 32 (64 bits), else MAX_K 64 (128 bits)
  // (loop_executor.hpp:47-52: first KMER_LIST entry with k < entry) -- so k = 32 already gets 60, not 28
  const int span_bits = k < 32 ? 64 : k < 64 ? 128 : k < 96 ? 192 : 256;
  int maxs = (span_bits - 8) / 2; if (maxs > 255) maxs = 255;   // Sequence2SuperKmer.hpp:146
  const u64 nm = 1ULL << (2 * m);

  // (bases that kmx_reads_upload has sent ahead: they lie on the device already, or are on their way there)
  kmx_ctx::ReadsAhead* ahead = nullptr;
  if (!segs && bases) for (auto& a : ctx->ahead) if (a.live && a.d == bases) ahead = &a;
  if (ahead && ahead->bytes < total_bases) return ctx->fail(KMX_E_INVAL, "kmx_reads_upload: fewer bytes were uploaded than the offsets name");
  char* d_bases = ahead ? ahead->d : (char*)ctx->dalloc(total_bases + 16);
  u64* d_offs = (u64*)ctx->dalloc((n_seqs + 1) * 8);
  // the repartition table: resident in the context, uploaded when it is another one than the previous call's (address, size, and a
  // digest over all of it: 8 words at a time, ~0.05 ms for 2 MB -- against a 2 MB upload from pageable memory per sample)
  u16* d_rep = nullptr; bool rep_upload = false;
  if (repart) {
    u64 dg = 0x9E3779B97F4A7C15ULL ^ nm;
    { const u64* w = reinterpret_cast<const u64*>(repart); const size_t nw = (size_t)(nm * 2 / 8);
      u64 a0 = 1, a1 = 2, a2 = 3, a3 = 4;
      size_t i = 0;
      for (; i + 4 <= nw; i += 4) { a0 = (a0 ^ w[i]) * 0x9E3779B97F4A7C15ULL; a1 = (a1 ^ w[i + 1]) * 0xC2B2AE3D27D4EB4FULL; a2 = (a2 ^ w[i + 2]) * 0x165667B19E3779F9ULL; a3 = (a3 ^ w[i + 3]) * 0x85EBCA77C2B2AE63ULL; }
      for (; i < nw; i++) a0 = (a0 ^ w[i]) * 0x9E3779B97F4A7C15ULL;
      for (size_t b = nw * 8; b < nm * 2; b++) a1 = (a1 ^ reinterpret_cast<const u8*>(repart)[b]) * 0xC2B2AE3D27D4EB4FULL;
      dg ^= a0 ^ (a1 >> 7) ^ (a2 << 9) ^ (a3 >> 13) ^ (a0 >> 31); }
    if (!ctx->d_rep || ctx->rep_n != nm || ctx->rep_host != repart || ctx->rep_digest != dg) {
      if (ctx->d_rep && ctx->rep_n != nm) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->d_rep); ctx->d_rep = nullptr; }
      if (!ctx->d_rep && hipMalloc((void**)&ctx->d_rep, nm * 2) != hipSuccess) { ctx->d_rep = nullptr; return ctx->fail(KMX_E_NOMEM, "superk: device allocation failed"); }
      ctx->rep_n = nm; ctx->rep_host = repart; ctx->rep_digest = dg; rep_upload = true;
    }
    d_rep = ctx->d_rep;
  } else d_rep = (u16*)ctx->dalloc(nm * 2);      // (the sampling pass: any table -- zeros)
  const bool rep_pooled = repart == nullptr;
  u32* d_cnt = (u32*)ctx->dalloc((n_seqs + 1) * 4);
  u32* d_doff = (u32*)ctx->dalloc((n_seqs + 1) * 4);
  u32* d_first = (u32*)ctx->dalloc(((size_t)n_smp + 1) * 4);
  std::vector<void*> blocks = {d_offs, d_cnt, d_doff, d_first};
  if (!ahead) blocks.push_back(d_bases);
  if (rep_pooled) blocks.push_back(d_rep);
  if (!d_rep) { for (void* b : blocks) ctx->dfree(b); return ctx->fail(KMX_E_NOMEM, "superk: device allocation failed"); }
  bool aux_busy = false;      // (kmx_superk_raw's copies are on the second stream: nothing they read is given back before they are through)
  auto release = [&]() { if (aux_busy && ctx->aux) (void)hipStreamSynchronize(ctx->aux); for (void* b : blocks) ctx->dfree(b); };
  for (void* b : blocks) if (!b) { release(); return ctx->fail(KMX_E_NOMEM, "superk: device allocation failed"); }
  auto fail = [&](hipError_t e, const char* what) { release(); return ctx->fail(KMX_E_HIP, std::string(what) + ": " + hipGetErrorString(e)); };
  StageClock clk(st, "superk_partition");
  PhaseClock ph; struct TlSet { PhaseClock* p; TlSet(PhaseClock* q) : p(q) { tl_phase = q; } ~TlSet() { tl_phase = nullptr; } } tlset(&ph);
  ph.mark(0);
  hipError_t e;
  if (segs) {
    for (u32 i = 0; i < segs->n; i++) {
      const u64 nb = segs->base_first[i + 1] - segs->base_first[i];
      if (nb && (e = hipMemcpyAsync(d_bases + segs->base_first[i], segs->bases[i], nb, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e, "upload bases");
    }
    if ((e = hipMemcpyAsync(d_first, segs->seq_first, ((size_t)segs->n + 1) * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e, "upload sample bounds");
  } else if (ahead) { if ((e = hipStreamWaitEvent(st, ahead->ev, 0)) != hipSuccess) return fail(e, "wait for the uploaded bases"); }
  else if ((e = hipMemcpyAsync(d_bases, bases, total_bases, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e, "upload bases");
  const SkMulti mu{segs ? d_first : nullptr, n_smp, segs ? segs->parts : 0u, segs ? (u32)nm : 0u};
  // (round 6: the reads' offsets go through a page-locked block -- the caller's array is pageable as a rule, and a copy from pageable memory
  //  is staged by the runtime with the calling thread waiting on the stream; 1.6 MB a sample)
  // (... unless they lie in page-locked memory already: `kmx pipeline` puts them behind the bases in the batch's block)
  u64* h_offs = (segs || kmx_is_pinned(offsets, (n_seqs + 1) * 8)) ? nullptr : (u64*)ctx->halloc((n_seqs + 1) * 8);
  struct HOffs { kmx_ctx* c; void* p; ~HOffs() { c->hfree(p); } } h_offs_rel{ctx, h_offs};
  if (h_offs) memcpy(h_offs, offsets, (n_seqs + 1) * 8);
  if ((e = hipMemcpyAsync(d_offs, h_offs ? (const void*)h_offs : (const void*)offsets, (n_seqs + 1) * 8, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e, "upload offsets");
  if (repart) { if (rep_upload && (e = hipMemcpyAsync(d_rep, repart, nm * 2, hipMemcpyHostToDevice, st)) != hipSuccess) { ctx->rep_host = nullptr; return fail(e, "upload repartition"); } }
  else if ((e = hipMemsetAsync(d_rep, 0, nm * 2, st)) != hipSuccess) return fail(e, "memset");
  // (KMX_SUPERK_ONE_PASS: see below; KMX_STATS_ATOMICS=1: the statistics by atomics while the reads are walked, as rounds 1-3 had them)
  const bool two_pass = getenv("KMX_SUPERK_ONE_PASS") == nullptr || segs || wide_k;      // (read per call: the tests switch it)
  StatsDev sd;
  sd.want_defer = two_pass && want_streams && getenv("KMX_STATS_ATOMICS") == nullptr && !wide_k;
  { const int rc = sd.alloc(ctx, stats, raw, nb_parts, nm, blocks, st, n_smp); if (rc != KMX_OK) { release(); return rc; } }
  if (raw) for (u32 i = 0; i < n_smp; i++) { raw[i].nb_superk = 0; raw[i].minim_sparse_n = 0; }
  const dim3 g1((unsigned)((n_seqs + 3) / 4)), b1(256);   // one wave per read
  if (!want_streams) {   // statistics only (the sampling pass of the repartition): one walk, nothing emitted
    u64 use = n_seqs;
    if (sampling) {
      // the shortest prefix of the reads that holds more than `budget` super-k-mers (the reference's iterator is cancelled by the
      // super-k-mer that brings the count past the sample size, and stops before the next read; RepartitionAlgorithm.cpp:205-211)
      if (wide_k) hipLaunchKernelGGL((k_superk_wide<false, false>), g1, b1, 0, st, d_bases, d_offs, (u64)n_seqs, (int)k, (int)m, maxs, d_rep, d_cnt,
                                     (const u32*)nullptr, (SkDesc*)nullptr, sd.S, SkSort{nullptr, nullptr, nullptr});
      else
      hipLaunchKernelGGL((k_superk_wave<false, false>), g1, b1, 0, st, d_bases, d_offs, (u64)n_seqs, (int)k, (int)m, maxs, d_rep, d_cnt,
                         (const u32*)nullptr, (SkDesc*)nullptr, sd.S, SkSort{nullptr, nullptr, nullptr}, SkLook{}, mu);
      std::vector<u32> cnt(n_seqs);
      if ((e = hipMemcpyAsync(cnt.data(), d_cnt, n_seqs * 4, hipMemcpyDeviceToHost, st)) != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "sample counts");
      u64 acc = 0; use = 0;
      while (use < n_seqs) { acc += cnt[use++]; if (acc > budget) break; }
      if (n_superk) *n_superk = acc;
    }
    if (n_used) *n_used = use;
    const dim3 gs((unsigned)((use + 3) / 4));
    if (wide_k) hipLaunchKernelGGL((k_superk_wide<false, true>), gs, b1, 0, st, d_bases, d_offs, (u64)use, (int)k, (int)m, maxs, d_rep, d_cnt,
                                   (const u32*)nullptr, (SkDesc*)nullptr, sd.S, SkSort{nullptr, nullptr, nullptr});
    else
    hipLaunchKernelGGL((k_superk_wave<false, true>), gs, b1, 0, st, d_bases, d_offs, (u64)use, (int)k, (int)m, maxs, d_rep, d_cnt,
                       (const u32*)nullptr, (SkDesc*)nullptr, sd.S, SkSort{nullptr, nullptr, nullptr}, SkLook{}, mu);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e, "k_superk_wave");
    const int rc = sd.collect(ctx, st);
    release();
    return rc;
  }
  // ---- round 6, kmx_count_reads_dev's usual call (one sample, k < 64, at most SKF_MAXP partitions, results into device stores): the
  //      split without a host round trip (superk_fast.hpp) handed to a count whose tables are made on the device (count.hip,
  //      kmx_count_fast_tail).  One read-back, at the count's end.  A status bit raised on the device -- or KMX_COUNT_FAST=0 -- sends
  //      the call down the path below (which reads three sizes back and has the library sort the descriptors). ----
  {
    const char* fe = getenv("KMX_COUNT_FAST");      // (read per call: the tests switch it)
    const bool fast = creq && creq->lists && !streams_to_host && !wide_k && !segs && nb_parts <= SKF_MAXP && total_bases < (500ULL << 20) && two_pass && (!sd.any() || sd.deferred) && !ctx->hist_on &&
                      !(fe && !strcmp(fe, "0")) && !getenv("KMX_COUNT_SORT") && !getenv("KMX_COUNT_BUCKETS") && total_bases >= 1;
    if (fast) {
      ph.mark(1);
      const u32 P = nb_parts, wpg = skf_wpg(P);
      const u32 n_chunks = (u32)((n_seqs + SKF_RPW - 1) / SKF_RPW), R = (n_chunks + wpg - 1) / wpg;
      const u32 rpg = std::max<u32>(32u, (R + 127u) / 128u), Gc = (R + rpg - 1) / rpg;      // (k_sk_scan: at most 128 workgroups; its time goes with their number -- 16 rows each: 32 us, 7: 53 us)
      u32 pbits = 1; while ((1u << pbits) < P) pbits++;
      const u64 kb = total_bases;                                                   // k-mers of the batch: fewer than bases
      const u32 nd_cap = (u32)std::min<u64>(total_bases, total_bases / 4 + n_seqs + 1024);      // records the sorted arrays take (a super-k-mer holds ~9 k-mers; beyond: the old path)
      const int kw = (int)((k + 31) / 32);
      const SkfLayout L = kmx_fast_layout(creq->hash_mode ? 1 : kw);
      const u32 tb_max = (u32)(kb / L.target) + P + 1, nc_max = (u32)(kb / L.chunk) + P + 1, nb_max = (u32)((kb + SKF_DK - 1) / SKF_DK) + 1;
      // ONE block: the part that is cleared -- [k_sk_scan's flags][the bucket scans' flags][the sample sort's bucket counters] ... [control 32 B]
      // [minimizers that occur, u64][lost buckets, u32 at + 48][pad to 64], a multiple of 256 bytes (the runtime clears an unaligned tail with a
      // launch of its own) -- and, right behind the control block, what the host reads at the call's end IN ONE COPY with it: [pp (P + 1) u64]
      // [info 2 P u64][parts P uint4][pf (P + 1) u32][cfirst (P + 1) u32][pad to 16][the kept pairs' offsets per bucket, tb_max + 2 u32]
      const size_t z_flags = 0, z_sfl = z_flags + 128 * 4, z_cnt = z_sfl + 512 * 4, z_bytes = ((z_cnt + ((size_t)tb_max + 2) * 4 + 64 + 255) & ~(size_t)255), z_ctl = z_bytes - 64;
      const size_t P1f = (size_t)P + 1, o_info = P1f * 8, o_parts = o_info + (size_t)P * 16, o_pf = o_parts + (size_t)P * 16, o_cf = o_pf + P1f * 4, sumf = o_cf + P1f * 4;
      const size_t o_koff = (sumf + 15) & ~(size_t)15, back_bytes = 64 + o_koff + ((size_t)tb_max + 2) * 4;
      u8* d_z = (u8*)ctx->dalloc(z_bytes + back_bytes - 64);
      u8* d_sumf = d_z ? d_z + z_bytes : nullptr;
      u8* h_f = (u8*)ctx->halloc(back_bytes + 24);      // (+ the statistics' own read-back word, behind everything the main stream's copy writes)
      struct HRelF { kmx_ctx* c; void* p; ~HRelF() { c->hfree(p); } } h_f_rel{ctx, h_f};
      SkDesc* d_descf = (SkDesc*)ctx->dalloc(((size_t)total_bases + 64) * sizeof(SkDesc));
      u32* d_ccnt = (u32*)ctx->dalloc((size_t)n_chunks * 4);
      ulonglong2* d_T = (ulonglong2*)ctx->dalloc((size_t)R * P * 16), *d_agg = (ulonglong2*)ctx->dalloc((size_t)Gc * P * 16);
      u32* d_sb = (u32*)ctx->dalloc((size_t)nd_cap * 4 + 16);
      u64* d_bo = (u64*)ctx->dalloc(((size_t)nd_cap + 1) * 8);
      u32* d_idf = sd.deferred ? (u32*)ctx->dalloc((size_t)nd_cap * 4 + 16) : nullptr;
      u16* d_p16 = creq->hash_mode ? (u16*)ctx->dalloc((size_t)nd_cap * 2 + 16) : nullptr;
      u32* d_bf = (u32*)ctx->dalloc((size_t)nb_max * 4);
      u64* d_wordsf = (u64*)ctx->dalloc(((size_t)total_bases / 32 + 4) * 8);
      std::vector<void*> fb = {d_z, d_descf, d_ccnt, d_T, d_agg, d_sb, d_bo, d_bf, d_wordsf};
      bool ok = h_f != nullptr;
      for (void* b : fb) ok = ok && b;
      // the statistics' inputs (round 6): the minimizer of every descriptor in its slot's word, gathered into sorted order by the scatter (d_idf), and
      // the k-mers' strands, a bit each, written by the decode -- the walk is the build without statistics (0.28 instead of 0.36 ms)
      u32* d_mslot = sd.deferred ? (u32*)ctx->dalloc(((size_t)total_bases + 64) * 4) : nullptr;
      u64* d_strand = sd.deferred ? (u64*)ctx->dalloc((kb / 64 + 4) * 8) : nullptr;
      if (sd.deferred) { fb.push_back(d_mslot); fb.push_back(d_strand); fb.push_back(d_idf); ok = ok && d_mslot && d_strand && d_idf; sd.S.sk_rec = nullptr; }
      if (creq->hash_mode) { fb.push_back(d_p16); ok = ok && d_p16; }
      auto frel = [&]() { for (void* b : fb) ctx->dfree(b); };
      if (!ok) { frel(); release(); return ctx->fail(KMX_E_NOMEM, "superk: device allocation failed"); }
      SkfCtl* d_ctl = (SkfCtl*)(d_z + z_ctl); u64* d_nspf = (u64*)(d_z + z_ctl + 32);
      u64* d_ppf = (u64*)d_sumf, *d_infof = (u64*)(d_sumf + o_info); uint4* d_partsf = (uint4*)(d_sumf + o_parts); u32* d_pff = (u32*)(d_sumf + o_pf), *d_cff = (u32*)(d_sumf + o_cf);
      auto ffail = [&](hipError_t er, const char* what) { frel(); return fail(er, what); };
      kmx_count_chain_begin(ctx);
      struct ChainEnd { kmx_ctx* c; ~ChainEnd() { kmx_count_chain_end(c); } } chain_end{ctx};      // (an error on the way out: the lock goes back; the count's half ends the chain itself once everything is queued)
      if ((e = hipMemsetAsync(d_z, 0, z_bytes, st)) != hipSuccess) return ffail(e, "memset");
      kmx_launch_pack_bases(d_bases, total_bases, d_wordsf, st);
      const dim3 gw((n_chunks + 3) / 4);
      hipLaunchKernelGGL((k_superk_wave<true, false, false, false, true>), gw, b1, 0, st, d_bases, d_offs, (u64)n_seqs, (int)k, (int)m, maxs, d_rep, d_ccnt,
                         (const u32*)nullptr, d_descf, sd.S, SkSort{nullptr, d_mslot, nullptr}, SkLook{}, mu, (u32)SKF_RPW);
      hipLaunchKernelGGL(k_sk_hist, dim3(R), dim3(64 * wpg), (size_t)P * 12, st, d_descf, d_offs, d_ccnt, n_chunks, wpg, P, (u32)k, d_T, d_ctl);
      hipLaunchKernelGGL(k_sk_scan, dim3(Gc), dim3(256), 0, st, d_T, R, rpg, P, d_agg, (u32*)(d_z + z_flags), nd_cap, L, d_ctl, d_pff, d_ppf, d_bo, d_partsf, d_cff);
      hipLaunchKernelGGL(k_sk_scatter, dim3(R), dim3(64 * wpg), (size_t)wpg * P * 12, st, d_descf, d_offs, d_ccnt, n_chunks, wpg, P, pbits, (u32)k, d_T, d_ctl,
                         d_sb, d_bo, d_idf, d_p16, d_bf, (const u32*)d_mslot);
      if (superk_info) hipLaunchKernelGGL(k_superk_info, dim3((P + 63) / 64), dim3(64), 0, st, d_pff, d_bo, P, d_infof, (const SkfCtl*)d_ctl);
      // the PartiInfo statistics (kmx_superk_raw): from the sorted descriptors, on the context's SECOND stream beside the count kernels --
      // k_part_stats is a workgroup per partition waiting on gathers and LDS atomics (150 us by itself), the count kernels are bound by
      // their instructions.  Their tables travel back on that stream as well, before the count is through (before_wait below).
      hipStream_t sx = ctx->aux ? ctx->aux : st;
      u64* h_nsp = reinterpret_cast<u64*>(h_f + ((back_bytes + 7) & ~(size_t)7));      // (the minimizers that occur, read back on the second stream: its own 8 bytes -- the control block's 64 land at h_f whenever the main stream gets there)
      auto launch_stats = [&]() -> int {      // (behind the scatter walk of the sample sort: up to there the count's kernels want the LDS k_part_stats holds -- started behind the split it kept k_cs_splitters off the CUs: 40 -> 160 us)
        hipError_t er = hipSuccess;
        if (sx != st) {
          if (!ctx->ev_split) er = hipEventCreateWithFlags(&ctx->ev_split, hipEventDisableTiming);
          if (er == hipSuccess) er = hipEventRecord(ctx->ev_split, st);
          if (er == hipSuccess) er = hipStreamWaitEvent(sx, ctx->ev_split, 0);
          aux_busy = true;
        }
        if (er == hipSuccess) {
          sd.launch_part_stats(nullptr, d_pff, d_bases, k, d_nspf, sx, d_bo, d_sb, d_idf, d_strand, d_ctl);
          sd.launch_sparse(d_nspf, sx);
          er = hipMemcpyAsync(h_nsp, d_nspf, 8, hipMemcpyDeviceToHost, sx);
        }
        return er == hipSuccess ? KMX_OK : ctx->fail(KMX_E_HIP, std::string("superk statistics: ") + hipGetErrorString(er));
      };
      if ((e = hipGetLastError()) != hipSuccess) return ffail(e, "split kernels");
      ph.mark(2);
      kmx_fast_split F{d_wordsf, d_sb, d_bo, d_p16, d_bf, d_ctl, d_partsf, d_cff, (u32*)(d_z + z_cnt), (u32*)(d_z + z_sfl), P, kb, tb_max, nc_max, nb_max,
                       reinterpret_cast<SkfCtl*>(h_f), reinterpret_cast<const uint4*>(h_f + 64 + o_parts), d_strand, nullptr, nullptr,
                       reinterpret_cast<u32*>(d_sumf + o_koff), reinterpret_cast<u32*>(h_f + 64 + o_koff), back_bytes};
      if (sd.deferred) F.behind_scatter = launch_stats;
      bool raw_queued = false;
      if (raw && sd.deferred && sx != st)
        F.before_wait = [&]() -> int {      // the statistics are through long before the count: their size is read, their copies queued behind them
          if (hipStreamSynchronize(sx) != hipSuccess) return ctx->fail(KMX_E_HIP, "superk statistics: the second stream failed");
          raw_queued = true;
          return sd.finish_raw(ctx, h_nsp, sx);
        };
      const int frc = kmx_count_fast_tail(ctx, F, *creq);      // (waits for the stream: the copies above are through when it returns)
      if (frc < 0) { if (sx != st) (void)hipStreamSynchronize(sx); frel(); release(); return frc; }
      const SkfCtl* hc = reinterpret_cast<const SkfCtl*>(h_f);
      if (frc == 0 && (hc->status & SKF_ST_PART)) { if (sx != st) (void)hipStreamSynchronize(sx); frel(); release(); return ctx->fail(KMX_E_INVAL, "repartition table names a partition >= nb_parts"); }
      if (frc == 0) {
        const u64* ppf = reinterpret_cast<const u64*>(h_f + 64);
        sd.compacted();
        if (superk_info) memcpy(superk_info, h_f + 64 + o_info, (size_t)P * 16);
        for (u32 p = 0; p < P; p++) out_kmers[p] = (ppf[p + 1] >> 32) - (ppf[p] >> 32);
        if (raw) {
          raw[0].nb_superk = hc->nd;
          aux_busy = true;
          if (!raw_queued) {
            if (sx != st && (e = hipStreamSynchronize(sx)) != hipSuccess) return ffail(e, "sync");
            const int rc = sd.finish_raw(ctx, h_nsp, sx);
            if (rc != KMX_OK) { frel(); release(); return rc; }
          }
          if ((e = hipStreamSynchronize(sx)) != hipSuccess) return ffail(e, "sync");
        }
        frel(); release();
        ph.mark(7);
        return KMX_OK;
      }
      // (a status bit: the old path takes the call from the start -- the statistics of the abandoned pass are cleared)
      if (getenv("KMX_TRACE")) fprintf(stderr, "[kmx count] the sync-free path handed the call back (status %u, overflow %u)\n", hc->status, hc->overflow);
      if (sx != st && (e = hipStreamSynchronize(sx)) != hipSuccess) return ffail(e, "sync");      // (the statistics kernels of the abandoned pass)
      frel();
      if ((e = sd.clear(st)) != hipSuccess) return fail(e, "memset");
    }
  }
  // what the host reads back between the steps lands in one page-locked block (a copy into pageable memory is staged by the
  // runtime and waited for): [super-k-mers u64 + a spare] [pp (P + 1) u64] [info 2P u64] [minimizers that occur u64] [pf (P + 1) u32]
  const size_t P1 = (size_t)nb_parts + 1, sum_bytes = (P1 + 2 * (size_t)nb_parts + n_smp) * 8 + P1 * 4;      // (... [minimizers that occur, per sample] ...)
  u8* h_sum = (u8*)ctx->halloc(16 + sum_bytes);
  struct HRel { kmx_ctx* c; void* p; ~HRel() { c->hfree(p); } } h_sum_rel{ctx, h_sum};
  if (!h_sum) { release(); return ctx->fail(KMX_E_NOMEM, "superk: host staging allocation failed"); }
  u32 nd = 0;
  SkDesc* d_desc = nullptr; u16* d_keys = nullptr, *d_keys2 = nullptr; u32* d_ids = nullptr, *d_ids2 = nullptr, *d_sz = nullptr;
  u64* d_szs = nullptr, *d_boff = nullptr; u8* d_sum = nullptr;
  auto alloc_desc = [&](size_t n) -> bool {
    d_desc = (SkDesc*)ctx->dalloc(n * sizeof(SkDesc));
    d_keys = (u16*)ctx->dalloc(n * 2); d_keys2 = (u16*)ctx->dalloc(n * 2);
    d_ids = (u32*)ctx->dalloc(n * 4); d_ids2 = (u32*)ctx->dalloc(n * 4);
    d_sz = (u32*)ctx->dalloc(n * 4);
    d_szs = (u64*)ctx->dalloc((n + 1) * 8); d_boff = (u64*)ctx->dalloc((n + 1) * 8);
    d_sum = (u8*)ctx->dalloc(sum_bytes);
    bool ok = true;
    for (void* b : {(void*)d_desc, (void*)d_keys, (void*)d_keys2, (void*)d_ids, (void*)d_ids2, (void*)d_sz, (void*)d_szs, (void*)d_boff, (void*)d_sum}) { blocks.push_back(b); ok = ok && b; }
    return ok;
  };
  // ---- KMX_SUPERK_ONE_PASS: one pass over the reads (k_superk_wave<.., LB>): descriptors placed by a look-back over the workgroups.  Needs every read's
  //      descriptors in LDS (reads of at most SK_WCAP k-mers) and room for the descriptors before their number is known: half as
  //      many as there are k-mers (a super-k-mer holds ~9); more than that, or a longer read: the count + scan + emit passes below ----
  bool emitted = false;
  {
    // (measured: 2.74 against 2.53 ms per call on the 24 M k-mer sample, 0.83 against 0.76 on the 1 Mbp one -- the look-back's
    //  barriers, LDS staging and spinning cost more than the second walk over bases that are still in L2: off unless asked for)
    u64 maxlen = 0, nk_total = 0;
    for (u64 r = 0; r < n_seqs && !two_pass; r++) { const u64 l = offsets[r + 1] - offsets[r]; maxlen = std::max(maxlen, l); if (l >= k) nk_total += l - k + 1; }
    const u64 cap = std::min<u64>(nk_total, nk_total / 2 + n_seqs + 1024);
    if (!two_pass && nk_total > 0 && maxlen >= k && maxlen - k + 1 <= (u64)SK_WCAP && cap < 0xFFFFFF00ULL) {
      const size_t nwg = (size_t)((n_seqs + 3) / 4);
      unsigned long long* d_state = (unsigned long long*)ctx->dalloc((nwg + 1) * 8); blocks.push_back(d_state);
      if (!d_state || !alloc_desc((size_t)cap)) { release(); return ctx->fail(KMX_E_NOMEM, "superk: device allocation failed"); }
      if ((e = hipMemsetAsync(d_state, 0, (nwg + 1) * 8, st)) != hipSuccess) return fail(e, "memset");
      u32* const d_tick = reinterpret_cast<u32*>(d_state + nwg);      // the ticket counter and the overflow word share the last entry
      const SkLook lk{d_state, d_tick, d_tick + 1, (u32)cap};
      if (sd.any()) hipLaunchKernelGGL((k_superk_wave<true, true, true>), g1, b1, 0, st, d_bases, d_offs, (u64)n_seqs, (int)k, (int)m, maxs, d_rep, (u32*)nullptr,
                                        (const u32*)nullptr, d_desc, sd.S, SkSort{d_keys, d_ids, d_sz}, lk, mu);
      else hipLaunchKernelGGL((k_superk_wave<true, false, true>), g1, b1, 0, st, d_bases, d_offs, (u64)n_seqs, (int)k, (int)m, maxs, d_rep, (u32*)nullptr,
                              (const u32*)nullptr, d_desc, sd.S, SkSort{d_keys, d_ids, d_sz}, lk, mu);
      if ((e = hipMemcpyAsync(h_sum, d_state + nwg - 1, 16, hipMemcpyDeviceToHost, st)) != hipSuccess) return fail(e, "memcpy");
      if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "sync");
      const unsigned long long last = reinterpret_cast<const unsigned long long*>(h_sum)[0];
      const u32 over = reinterpret_cast<const u32*>(h_sum)[3];
      if (!over) { nd = (u32)(last & ((1ULL << 62) - 1)); emitted = true; }
      else if ((e = sd.clear(st)) != hipSuccess) return fail(e, "memset");      // (the statistics of the abandoned pass)
    }
  }
  if (!emitted) {
    if (wide_k) hipLaunchKernelGGL((k_superk_wide<false, false>), g1, b1, 0, st, d_bases, d_offs, (u64)n_seqs, (int)k, (int)m, maxs, d_rep, d_cnt,
                                   (const u32*)nullptr, (SkDesc*)nullptr, sd.S, SkSort{nullptr, nullptr, nullptr});
    else
    hipLaunchKernelGGL((k_superk_wave<false, false>), g1, b1, 0, st, d_bases, d_offs, (u64)n_seqs, (int)k, (int)m, maxs, d_rep, d_cnt,
                       (const u32*)nullptr, (SkDesc*)nullptr, sd.S, SkSort{nullptr, nullptr, nullptr}, SkLook{}, mu);
    size_t tb = 0;
    if ((e = rocprim::exclusive_scan(nullptr, tb, d_cnt, d_doff, 0u, (size_t)n_seqs + 1, rocprim::plus<u32>(), st)) != hipSuccess) return fail(e, "scan size");
    void* d_tmp = ctx->dalloc(tb ? tb : 256); blocks.push_back(d_tmp);
    if (!d_tmp) { release(); return ctx->fail(KMX_E_NOMEM, "superk: device allocation failed"); }
    if ((e = rocprim::exclusive_scan(d_tmp, tb, d_cnt, d_doff, 0u, (size_t)n_seqs + 1, rocprim::plus<u32>(), st)) != hipSuccess) return fail(e, "scan");
    if ((e = hipMemcpyAsync(h_sum, d_doff + n_seqs, 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return fail(e, "memcpy");
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "sync");
    nd = *reinterpret_cast<const u32*>(h_sum);
  }
  clk.mark("upload+scan");
  if (nd == 0) {
    if (raw) for (u32 i = 0; i < n_smp; i++) {      // nothing counted: the caller's tables are all zeros
      if (raw[i].part_radix) memset(raw[i].part_radix, 0, (size_t)(nb_parts / n_smp) * 1280 * 4);
      if (raw[i].minim_sparse) raw[i].minim_sparse_n = 0;
      else {
        if (raw[i].minim_superks) memset(raw[i].minim_superks, 0, nm * 4);
        if (raw[i].minim_kmers) memset(raw[i].minim_kmers, 0, nm * 4);
      }
    }
    release();
    for (u32 p = 0; p < nb_parts; p++) {
      if (streams_to_host) out_bytes[p] = (uint8_t*)malloc(1);
      if (creq && creq->lists) { creq->lists[p].recs = nullptr; creq->lists[p].n = 0; }
      else if (creq) { creq->keys[p] = (uint64_t*)malloc(8); creq->counts[p] = (uint32_t*)malloc(4); creq->n_out[p] = 0; }
      if (superk_info) { superk_info[2 * p] = 0; superk_info[2 * p + 1] = 0; }
    }
    return KMX_OK;
  }
  if (!emitted) {
    if (!alloc_desc((size_t)nd)) { release(); return ctx->fail(KMX_E_NOMEM, "superk: device allocation failed"); }
    if (sd.deferred) {
      sd.S.sk_rec = (ulonglong2*)ctx->dalloc((size_t)nd * 16);
      blocks.push_back(sd.S.sk_rec);
      if (!sd.S.sk_rec) { release(); return ctx->fail(KMX_E_NOMEM, "superk: device allocation failed"); }
      hipLaunchKernelGGL((k_superk_wave<true, true, false, true>), g1, b1, 0, st, d_bases, d_offs, (u64)n_seqs, (int)k, (int)m, maxs, d_rep, d_cnt,
                         (const u32*)d_doff, d_desc, sd.S, SkSort{d_keys, d_ids, d_sz}, SkLook{}, mu);
    } else if (wide_k) {
      if (sd.any()) hipLaunchKernelGGL((k_superk_wide<true, true>), g1, b1, 0, st, d_bases, d_offs, (u64)n_seqs, (int)k, (int)m, maxs, d_rep, d_cnt,
                                       (const u32*)d_doff, d_desc, sd.S, SkSort{d_keys, d_ids, d_sz});
      else hipLaunchKernelGGL((k_superk_wide<true, false>), g1, b1, 0, st, d_bases, d_offs, (u64)n_seqs, (int)k, (int)m, maxs, d_rep, d_cnt,
                              (const u32*)d_doff, d_desc, sd.S, SkSort{d_keys, d_ids, d_sz});
    } else
    if (sd.any()) hipLaunchKernelGGL((k_superk_wave<true, true>), g1, b1, 0, st, d_bases, d_offs, (u64)n_seqs, (int)k, (int)m, maxs, d_rep, d_cnt,
                                      (const u32*)d_doff, d_desc, sd.S, SkSort{d_keys, d_ids, d_sz}, SkLook{}, mu);
    else hipLaunchKernelGGL((k_superk_wave<true, false>), g1, b1, 0, st, d_bases, d_offs, (u64)n_seqs, (int)k, (int)m, maxs, d_rep, d_cnt,
                            (const u32*)d_doff, d_desc, sd.S, SkSort{d_keys, d_ids, d_sz}, SkLook{}, mu);
  }
  const dim3 g2((nd + 255) / 256), b2(256);
  size_t tb2 = 0, tb3 = 0;
  if ((e = rocprim::radix_sort_pairs(nullptr, tb2, d_keys, d_keys2, d_ids, d_ids2, (size_t)nd, 0, 16, st)) != hipSuccess) return fail(e, "sort size");
  if ((e = rocprim::exclusive_scan(nullptr, tb3, d_szs, d_boff, (u64)0, (size_t)nd + 1, rocprim::plus<u64>(), st)) != hipSuccess) return fail(e, "scan size");
  void* d_tmp2 = ctx->dalloc(std::max(tb2, tb3) + 256); blocks.push_back(d_tmp2);
  if (!d_tmp2) { release(); return ctx->fail(KMX_E_NOMEM, "superk: device allocation failed"); }
  if ((e = rocprim::radix_sort_pairs(d_tmp2, tb2, d_keys, d_keys2, d_ids, d_ids2, (size_t)nd, 0, 16, st)) != hipSuccess) return fail(e, "sort");
  // counting without the super-k-mer files: the record stream is never written -- the k-mers are cut from the bases themselves
  // (2 bits each, k_pack_bases), a record being where its first base lies
  const bool direct = creq && !streams_to_host && !wide_k;
  u32* d_sbase = nullptr; u64* d_words = nullptr;
  if (direct) {
    d_sbase = (u32*)ctx->dalloc((size_t)nd * 4); d_words = (u64*)ctx->dalloc(((size_t)total_bases / 32 + 4) * 8);
    blocks.push_back(d_sbase); blocks.push_back(d_words);
    if (!d_sbase || !d_words) { release(); return ctx->fail(KMX_E_NOMEM, "superk: device allocation failed"); }
    kmx_launch_pack_bases(d_bases, total_bases, d_words, st);
  }
  hipLaunchKernelGGL(k_superk_gather_sizes2, dim3((nd + 256) / 256), b2, 0, st, d_ids2, d_sz, d_desc, nd, d_szs, d_sbase);
  if ((e = rocprim::exclusive_scan(d_tmp2, tb3, d_szs, d_boff, (u64)0, (size_t)nd + 1, rocprim::plus<u64>(), st)) != hipSuccess) return fail(e, "scan");
  u64* d_pp = (u64*)d_sum, *d_info = d_pp + P1, *d_nsp = d_info + 2 * (size_t)nb_parts; u32* d_pf = (u32*)(d_nsp + n_smp);
  hipLaunchKernelGGL(k_superk_part_bounds, dim3((nb_parts + 256) / 256), dim3(256), 0, st, d_keys2, nd, nb_parts, d_boff, d_pp, d_pf, d_nsp, n_smp);
  if (superk_info) hipLaunchKernelGGL(k_superk_info, dim3((nb_parts + 63) / 64), dim3(64), 0, st, d_pf, d_boff, nb_parts, d_info);
  sd.launch_part_stats(d_ids2, d_pf, d_bases, k, d_nsp, st);
  sd.launch_sparse(d_nsp, st);
  if ((e = hipMemcpyAsync(h_sum + 16, d_sum, sum_bytes, hipMemcpyDeviceToHost, st)) != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "sync");
  const u64* pp = reinterpret_cast<const u64*>(h_sum + 16);
  const u32* pf = reinterpret_cast<const u32*>(h_sum + 16 + (P1 + 2 * (size_t)nb_parts + n_smp) * 8);
  sd.compacted();
  const u64 tot = pp[nb_parts];                 // (the prefix at the end of the last partition: all k-mers << 32 | all bytes)
  clk.mark("emit+sort");
  if (superk_info) memcpy(superk_info, pp + P1, (size_t)nb_parts * 16);
  if (raw) {
    for (u32 i = 0; i < n_smp; i++) raw[i].nb_superk = (u64)pf[(size_t)(i + 1) * (nb_parts / n_smp)] - pf[(size_t)i * (nb_parts / n_smp)];      // (one descriptor per super-k-mer)
    // (queued on the context's SECOND stream -- the tables are complete, this stream has just been waited for -- so that the 2.5 MB
    //  per sample cross PCIe beside the count kernels instead of in front of them; waited for before the call returns)
    aux_busy = true;
    const int rc = sd.finish_raw(ctx, pp + P1 + 2 * (size_t)nb_parts, ctx->aux ? ctx->aux : st);
    if (rc != KMX_OK) { release(); return rc; }
  }
  { const int rc = sd.collect(ctx, st); if (rc != KMX_OK) { release(); return rc; } }
  if (pf[nb_parts] != nd) { release(); return ctx->fail(KMX_E_INVAL, "repartition table names a partition >= nb_parts"); }
  const u64 total_bytes = tot & 0xFFFFFFFFULL;
  u8* d_out = direct ? nullptr : (u8*)ctx->dalloc(total_bytes + 16);
  if (!direct) blocks.push_back(d_out);
  u8* h_out = streams_to_host ? (u8*)ctx->halloc(total_bytes + 16) : nullptr;
  if ((!direct && !d_out) || (streams_to_host && !h_out)) { ctx->hfree(h_out); release(); return ctx->fail(KMX_E_NOMEM, "superk: allocation failed"); }
  if (!direct) hipLaunchKernelGGL(k_superk_pack, g2, b2, 0, st, d_bases, d_desc, d_ids2, d_boff, nd, (int)k, d_out);
  if ((e = hipGetLastError()) != hipSuccess) { ctx->hfree(h_out); return fail(e, "k_superk_pack"); }
  clk.mark("pack");
  for (u32 p = 0; p < nb_parts; p++) out_kmers[p] = (pp[p + 1] >> 32) - (pp[p] >> 32);
  if (creq) {   // count straight from the device-resident stream (kmx_count_reads)
    std::vector<u64> pko((size_t)nb_parts + 1); for (u32 p = 0; p <= nb_parts; p++) pko[p] = pp[p] >> 32;
    const int rc = kmx_count_from_device(ctx, direct ? (const u8*)d_words : d_out, d_boff, d_keys2, nd, tot >> 32, nb_parts, pko.data(), *creq, d_sbase);
    if (rc != KMX_OK) { ctx->hfree(h_out); release(); return rc; }
    clk.mark("count");
  }
  if (!streams_to_host) {
    if ((raw || !creq) && (e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "sync");
    if (raw && ctx->aux && (e = hipStreamSynchronize(ctx->aux)) != hipSuccess) return fail(e, "sync");      // (kmx_superk_raw's copies)
    release(); return KMX_OK;
  }
  // the partition-ordered stream comes back in one copy; a few host threads cut it into the per-partition buffers
  if ((e = hipMemcpyAsync(h_out, d_out, total_bytes, hipMemcpyDeviceToHost, st)) != hipSuccess ||
      (e = hipStreamSynchronize(st)) != hipSuccess || (raw && ctx->aux && (e = hipStreamSynchronize(ctx->aux)) != hipSuccess)) { ctx->hfree(h_out); return fail(e, "download"); }
  release();
  std::atomic<u32> next{0}; std::atomic<int> oom{0};
  auto fill = [&]() {
    for (u32 p; (p = next++) < nb_parts;) {
      const u64 lo = pp[p] & 0xFFFFFFFFULL, hi = pp[p + 1] & 0xFFFFFFFFULL;
      out_bytes[p] = (uint8_t*)malloc(hi - lo ? hi - lo : 1);
      if (!out_bytes[p]) { oom = 1; continue; }
      memcpy(out_bytes[p], h_out + lo, hi - lo);
      out_len[p] = hi - lo; out_kmers[p] = (pp[p + 1] >> 32) - (pp[p] >> 32);
    }
  };
  {
    const unsigned nthr = std::max(1u, std::min({16u, std::thread::hardware_concurrency(), nb_parts}));
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nthr; t++) th.emplace_back(fill);
    fill();
    for (auto& x : th) x.join();
  }
  ctx->hfree(h_out);
  clk.mark("download");
  if (oom) return ctx->fail(KMX_E_NOMEM, "superk: host allocation failed");
  return KMX_OK;
}

extern "C" int kmx_reads_upload(kmx_ctx* ctx, const char* bases, uint64_t n_bytes, const char** dev_bases)
{
  if (!ctx) return KMX_E_INVAL;
  if (!dev_bases || (!bases && n_bytes)) return ctx->fail(KMX_E_INVAL, "kmx_reads_upload: null argument");
  *dev_bases = nullptr;
  kmx_ctx::ReadsAhead* slot = nullptr;
  for (auto& a : ctx->ahead) if (!a.live) { slot = &a; break; }
  if (!slot) return ctx->fail(KMX_E_INVAL, "kmx_reads_upload: KMX_READS_AHEAD uploads are alive already");
  KMX_HIP(ctx, hipSetDevice(ctx->device));
  if (!slot->ev) KMX_HIP(ctx, hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming));
  slot->d = (char*)ctx->dalloc(n_bytes + 64);
  if (!slot->d) return ctx->fail(KMX_E_NOMEM, "kmx_reads_upload: device allocation failed");
  hipError_t e = n_bytes ? hipMemcpyAsync(slot->d, bases, n_bytes, hipMemcpyHostToDevice, ctx->up) : hipSuccess;
  if (e == hipSuccess) e = hipEventRecord(slot->ev, ctx->up);
  if (e != hipSuccess) { ctx->dfree(slot->d); slot->d = nullptr; return ctx->fail(KMX_E_HIP, std::string("kmx_reads_upload: ") + hipGetErrorString(e)); }
  slot->bytes = n_bytes; slot->live = true;
  *dev_bases = slot->d;
  return KMX_OK;
}
extern "C" void kmx_reads_release(kmx_ctx* ctx, const char* dev_bases)
{
  if (!ctx || !dev_bases) return;
  for (auto& a : ctx->ahead) if (a.live && a.d == dev_bases) {
    (void)hipEventSynchronize(a.ev);      // (released instead of counted: the copy must be through before the block is used again)
    ctx->dfree(a.d); a.d = nullptr; a.live = false; a.bytes = 0;
  }
}

extern "C" int kmx_superk_partition(kmx_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n_seqs,
                                    uint32_t k, uint32_t m, const uint16_t* repart, uint32_t nb_parts,
                                    uint8_t** out_bytes, uint64_t* out_len, uint64_t* out_kmers)
{
  if (ctx && !out_bytes) return ctx->fail(KMX_E_INVAL, "kmx_superk_partition: bad argument");
  return superk_impl(ctx, bases, offsets, n_seqs, k, m, repart, nb_parts, out_bytes, out_len, out_kmers, nullptr);
}

extern "C" int kmx_superk_partition_stats(kmx_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n_seqs,
                                          uint32_t k, uint32_t m, const uint16_t* repart, uint32_t nb_parts,
                                          uint8_t** out_bytes, uint64_t* out_len, uint64_t* out_kmers, kmx_superk_stats* stats)
{
  return superk_impl(ctx, bases, offsets, n_seqs, k, m, repart, nb_parts, out_bytes, out_len, out_kmers, stats);
}

extern "C" int kmx_superk_sample(kmx_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n_seqs,
                                 uint32_t k, uint32_t m, uint64_t budget, kmx_superk_stats* stats, uint64_t* n_used, uint64_t* n_superk)
{
  if (ctx && (!stats || !n_used || !n_superk)) return ctx->fail(KMX_E_INVAL, "kmx_superk_sample: null argument");
  return superk_impl(ctx, bases, offsets, n_seqs, k, m, nullptr, 1, nullptr, nullptr, nullptr, stats, true, budget, n_used, n_superk);
}

extern "C" int kmx_count_reads(kmx_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n_seqs,
                               uint32_t k, uint32_t m, const uint16_t* repart, uint32_t nb_parts,
                               int hash_mode, uint64_t window, uint32_t hard_min,
                               uint64_t** keys, uint32_t** counts, uint64_t* n_out, uint64_t* out_kmers,
                               uint8_t** superk_bytes, uint64_t* superk_len, uint64_t* superk_info, kmx_superk_stats* stats)
{
  if (!ctx) return KMX_E_INVAL;
  if (!keys || !counts || !n_out || !out_kmers || (superk_bytes && !superk_len)) return ctx->fail(KMX_E_INVAL, "kmx_count_reads: null argument");
  if (hash_mode && window == 0) return ctx->fail(KMX_E_INVAL, "hash window is 0");
  for (u32 p = 0; p < nb_parts; p++) { keys[p] = nullptr; counts[p] = nullptr; n_out[p] = 0; }
  kmx_count_req rq{k, hash_mode, window, hard_min, keys, counts, n_out};
  std::vector<uint8_t*> dummy_b; std::vector<uint64_t> dummy_l;
  uint8_t** ob = superk_bytes; uint64_t* ol = superk_len;
  if (!ob) { dummy_b.assign(nb_parts, nullptr); dummy_l.assign(nb_parts, 0); ob = dummy_b.data(); ol = dummy_l.data(); }
  const int rc = superk_impl(ctx, bases, offsets, n_seqs, k, m, repart, nb_parts, ob, ol, out_kmers, stats, false, 0, nullptr, nullptr, &rq, superk_bytes != nullptr, superk_info);
  if (rc != KMX_OK) for (u32 p = 0; p < nb_parts; p++) { free(keys[p]); free(counts[p]); keys[p] = nullptr; counts[p] = nullptr; n_out[p] = 0; }
  return rc;
}

extern "C" int kmx_count_reads_dev(kmx_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n_seqs,
                                   uint32_t k, uint32_t m, const uint16_t* repart, uint32_t nb_parts,
                                   int hash_mode, uint64_t window, uint32_t hard_min,
                                   kmx_store* const* stores, uint32_t n_stores, kmx_list* lists, uint64_t* out_kmers,
                                   uint8_t** superk_bytes, uint64_t* superk_len, uint64_t* superk_info,
                                   kmx_superk_stats* stats, kmx_superk_raw* raw)
{
  if (!ctx) return KMX_E_INVAL;
  if (!stores || !n_stores || !lists || !out_kmers || (superk_bytes && !superk_len)) return ctx->fail(KMX_E_INVAL, "kmx_count_reads_dev: null argument");
  for (u32 d = 0; d < n_stores; d++) if (!stores[d]) return ctx->fail(KMX_E_INVAL, "kmx_count_reads_dev: null store");
  if (hash_mode && window == 0) return ctx->fail(KMX_E_INVAL, "hash window is 0");
  for (u32 p = 0; p < nb_parts; p++) { lists[p].recs = nullptr; lists[p].n = 0; }
  kmx_count_req rq{k, hash_mode, window, hard_min, nullptr, nullptr, nullptr, stores, n_stores, lists};
  std::vector<uint8_t*> dummy_b; std::vector<uint64_t> dummy_l;
  uint8_t** ob = superk_bytes; uint64_t* ol = superk_len;
  if (!ob) { dummy_b.assign(nb_parts, nullptr); dummy_l.assign(nb_parts, 0); ob = dummy_b.data(); ol = dummy_l.data(); }
  const int rc = superk_impl(ctx, bases, offsets, n_seqs, k, m, repart, nb_parts, ob, ol, out_kmers, stats, false, 0, nullptr, nullptr, &rq, superk_bytes != nullptr, superk_info, raw);
  if (rc != KMX_OK) for (u32 p = 0; p < nb_parts; p++) { lists[p].recs = nullptr; lists[p].n = 0; }
  return rc;
}

extern "C" int kmx_count_reads_dev_multi(kmx_ctx* ctx, uint32_t n_samples, const char* const* bases, const uint64_t* const* offsets, const uint64_t* n_seqs,
                                         uint32_t k, uint32_t m, const uint16_t* repart, uint32_t nb_parts,
                                         int hash_mode, uint64_t window, uint32_t hard_min,
                                         kmx_store* const* stores, uint32_t n_stores, kmx_list* lists, uint64_t* out_kmers,
                                         uint64_t* superk_info, kmx_superk_raw* raw)
{
  if (!ctx) return KMX_E_INVAL;
  if (!n_samples || !bases || !offsets || !n_seqs || !stores || !n_stores || !lists || !out_kmers) return ctx->fail(KMX_E_INVAL, "kmx_count_reads_dev_multi: null argument");
  for (u32 d = 0; d < n_stores; d++) if (!stores[d]) return ctx->fail(KMX_E_INVAL, "kmx_count_reads_dev_multi: null store");
  if (hash_mode && window == 0) return ctx->fail(KMX_E_INVAL, "hash window is 0");
  if ((u64)n_samples * nb_parts > 65535) return ctx->fail(KMX_E_UNSUPPORTED, "kmx_count_reads_dev_multi: samples x partitions above 65535");
  if (ctx->hist_on) return ctx->fail(KMX_E_UNSUPPORTED, "kmx_count_reads_dev_multi: the abundance histogram is per call -- one sample per call while it is on");
  if (n_samples == 1)
    return kmx_count_reads_dev(ctx, bases[0], offsets[0], n_seqs[0], k, m, repart, nb_parts, hash_mode, window, hard_min, stores, n_stores, lists, out_kmers,
                               nullptr, nullptr, superk_info, nullptr, raw);
  const u32 PT = n_samples * nb_parts;
  for (u32 p = 0; p < PT; p++) { lists[p].recs = nullptr; lists[p].n = 0; out_kmers[p] = 0; }
  // one run of reads: the offsets rebased (into page-locked memory: the upload is a DMA), the samples' first reads and first bases
  u64 tot_seqs = 0;
  for (u32 i = 0; i < n_samples; i++) { if (n_seqs[i] && (!offsets[i] || (offsets[i][n_seqs[i]] && !bases[i]))) return ctx->fail(KMX_E_INVAL, "kmx_count_reads_dev_multi: null sample"); tot_seqs += n_seqs[i]; }
  if (tot_seqs >= 0x7FFFFFFFULL) return ctx->fail(KMX_E_UNSUPPORTED, "batch of 2^31 reads or more: split it");
  const size_t hb = (tot_seqs + 1) * 8 + ((size_t)n_samples + 1) * (8 + 4) + 64;
  u8* h = (u8*)ctx->halloc(hb);
  if (!h) return ctx->fail(KMX_E_NOMEM, "kmx_count_reads_dev_multi: host staging allocation failed");
  struct HRel { kmx_ctx* c; void* p; ~HRel() { c->hfree(p); } } hrel{ctx, h};
  u64* offs = reinterpret_cast<u64*>(h); u64* base_first = offs + tot_seqs + 1; u32* seq_first = reinterpret_cast<u32*>(base_first + n_samples + 1);
  u64 at_seq = 0, at_base = 0;
  for (u32 i = 0; i < n_samples; i++) {
    seq_first[i] = (u32)at_seq; base_first[i] = at_base;
    for (u64 r = 0; r < n_seqs[i]; r++) offs[at_seq + r] = at_base + offsets[i][r];
    at_seq += n_seqs[i]; at_base += n_seqs[i] ? offsets[i][n_seqs[i]] : 0;
  }
  seq_first[n_samples] = (u32)at_seq; base_first[n_samples] = at_base; offs[tot_seqs] = at_base;
  const SkSegs segs{n_samples, bases, base_first, seq_first, nb_parts};
  kmx_count_req rq{k, hash_mode, window, hard_min, nullptr, nullptr, nullptr, stores, n_stores, lists, nb_parts};
  std::vector<uint8_t*> dummy_b(PT, nullptr); std::vector<uint64_t> dummy_l(PT, 0);
  const int rc = superk_impl(ctx, nullptr, reinterpret_cast<const uint64_t*>(offs), tot_seqs, k, m, repart, PT, dummy_b.data(), dummy_l.data(), out_kmers, nullptr, false, 0, nullptr, nullptr, &rq, false,
                             superk_info, raw, &segs);
  if (rc != KMX_OK) for (u32 p = 0; p < PT; p++) { lists[p].recs = nullptr; lists[p].n = 0; }
  return rc;
}
