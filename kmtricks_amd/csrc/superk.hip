// superk.hip -- reads -> canonical k-mers -> minimizers -> super-k-mers -> 2-bit records per partition
// on gfx950.  Replaces Model::iterate + ModelMinimizer::next + Sequence2SuperKmer::KmerFunctor +
// KmFillPartitions::processSuperkmer + SuperKmer::save (reference gatb kmer/impl/Model.hpp:725-765,
// 857-884, 1010-1139, 1254-1287, 1388-1433; kmer/impl/Sequence2SuperKmer.hpp:80-158;
// include/kmtricks/gatb/fill_partitions.hpp:59-105).
//
//   pass 1  k_superk_scan<false>: one thread per read rolls the forward k-mer, tracks the window
//           minimizer with the reference's rules (LUT[m-mer] = min(m-mer, revcomp) or 4^m-1 when it
//           contains AA except as prefix; a new m-mer wins only if strictly smaller; rescan from the
//           rightmost m-mer when the minimizer leaves the window) and counts its super-k-mers
//           (cut on minimizer change, invalid k-mer, or maxs k-mers);
//   scan    exclusive scan of the per-read counts (rocPRIM);
//   pass 2  k_superk_scan<true>: the same walk writes one descriptor per super-k-mer
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

namespace kmx {

struct SkDesc { u32 base; u16 part; u8 n; u8 pad; };   // base = index of the record's first base in `bases`

__device__ __forceinline__ bool nt_valid(u8 c)
{ // gatb tools/misc/api/Data.hpp:179-196
  const u8 u = c & 0xDF;
  return u == 'A' || u == 'C' || u == 'G' || u == 'T';
}

// m-mer at digit offset `s` (from the last base) of the forward k-mer ending at base `end` (inclusive)
__device__ __forceinline__ u32 mmer_at(const char* __restrict__ seq, u64 end, int s, int m)
{
  u32 v = 0;
  for (int j = m - 1; j >= 0; j--) v = (v << 2) | (((u8)seq[end - s - j] >> 1) & 3u);
  return v;
}

template <bool EMIT>
__global__ void k_superk_scan(const char* __restrict__ bases, const u64* __restrict__ offsets, u64 n_seqs,
                              int k, int m, int maxs, const u32* __restrict__ lut, const u16* __restrict__ repart,
                              u32* __restrict__ counts, const u32* __restrict__ desc_off, SkDesc* __restrict__ desc)
{
  const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_seqs) return;
  const u64 b0 = offsets[r], len = offsets[r + 1] - b0;
  u32 nsk = 0;
  if (len >= (u64)k) {
    const char* seq = bases + b0;
    const u32 maskm = (1u << (2 * m)) - 1;
    const int nbm = k - m + 1;
    u32 out = EMIT ? desc_off[r] : 0;
    // first k-mer: last bad character, rolling m-mer (Model.hpp:636-657)
    int bad = -1;
    for (int i = 0; i < k; i++) if (!nt_valid((u8)seq[i])) bad = i;
    u32 cur_mm = 0;
    for (int i = k - m; i < k; i++) cur_mm = ((cur_mm << 2) | (((u8)seq[i] >> 1) & 3u)) & maskm;
    // minimizer of the first k-mer: scan from the rightmost m-mer, strict '<' (Model.hpp:1254-1287)
    u32 minim = maskm; int pos = -1;
    for (int idx = nbm - 1, s = 0; idx >= 0; idx--, s++) {
      const u32 cand = lut[mmer_at(seq, (u64)k - 1, s, m)];
      if (cand < minim) { minim = cand; pos = idx; }
    }
    // super-k-mer state (Sequence2SuperKmer.hpp:90-132)
    u32 sk_min = 0; bool sk_valid = false; u32 sk_n = 0; u64 sk_first = 0;
    u64 end = (u64)k - 1;   // index of the last base of the current k-mer
    for (;;) {
      const bool valid = bad < 0;
      if (!valid) {
        if (sk_valid && sk_n) {
          if (EMIT) { SkDesc d; d.base = (u32)(b0 + sk_first); d.part = repart[sk_min]; d.n = (u8)sk_n; d.pad = 0; desc[out++] = d; }
          nsk++;
        }
        sk_n = 0; sk_valid = false;
      } else {
        if (!sk_valid) { sk_min = minim; sk_valid = true; }
        if (minim != sk_min || sk_n >= (u32)maxs) {
          if (sk_n) {
            if (EMIT) { SkDesc d; d.base = (u32)(b0 + sk_first); d.part = repart[sk_min]; d.n = (u8)sk_n; d.pad = 0; desc[out++] = d; }
            nsk++;
          }
          sk_n = 0;
        }
        sk_min = minim;
        if (sk_n == 0) sk_first = end + 1 - (u64)k;
        sk_n++;
      }
      if (end + 1 >= len) break;
      // next k-mer (Model.hpp:740-757, 1106-1139)
      end++;
      const u8 ch = (u8)seq[end];
      if (!nt_valid(ch)) bad = k - 1; else bad--;
      cur_mm = ((cur_mm << 2) | ((ch >> 1) & 3u)) & maskm;
      const u32 mmer = lut[cur_mm];
      pos--;
      if (mmer < minim) { minim = mmer; pos = nbm - 1; }
      else if (pos < 0) {
        minim = maskm; pos = -1;
        for (int idx = nbm - 1, s = 0; idx >= 0; idx--, s++) {
          const u32 cand = lut[mmer_at(seq, end, s, m)];
          if (cand < minim) { minim = cand; pos = idx; }
        }
      }
    }
    if (sk_valid && sk_n) {   // Sequence2SuperKmer.hpp:155
      if (EMIT) { SkDesc d; d.base = (u32)(b0 + sk_first); d.part = repart[sk_min]; d.n = (u8)sk_n; d.pad = 0; desc[out++] = d; }
      nsk++;
    }
  }
  if (!EMIT) counts[r] = nsk;
}

__global__ void k_superk_sizes(const SkDesc* __restrict__ desc, u32 n, int k, u16* __restrict__ keys, u32* __restrict__ ids, u32* __restrict__ sizes_unsorted)
{
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = desc[i].part; ids[i] = i;
  sizes_unsorted[i] = 1u + ((u32)k + desc[i].n - 1u + 3u) / 4u;
}
__global__ void k_superk_gather_sizes(const u32* __restrict__ ids, const u32* __restrict__ sizes_unsorted, u32 n, u64* __restrict__ sizes_sorted)
{
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sizes_sorted[i] = sizes_unsorted[ids[i]];
}
// partition-sorted order: (k-mers << 32 | bytes) per record, so one scan yields both prefixes (both totals < 2^32)
__global__ void k_superk_gather_sizes2(const u32* __restrict__ ids, const u32* __restrict__ sizes_unsorted, const SkDesc* __restrict__ desc,
                                       u32 n, u64* __restrict__ sizes_sorted)
{
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const u32 j = ids[i]; sizes_sorted[i] = ((u64)desc[j].n << 32) | sizes_unsorted[j]; }
}
// prefix (k-mers << 32 | bytes) at the first record of every partition (nb_parts + 1 entries) + that record's index
__global__ void k_superk_part_bounds(const u16* __restrict__ part_sorted, u32 n, u32 nb_parts, const u64* __restrict__ prefix,
                                     u64* __restrict__ part_prefix, u32* __restrict__ part_first)
{
  const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > nb_parts) return;
  u32 lo = 0, hi = n;
  while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (part_sorted[mid] < p) lo = mid + 1; else hi = mid; }
  part_prefix[p] = prefix[lo];
  part_first[p] = lo;
}

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
  for (int by = 0; by * 4 < ndig; by++) {
    u32 v = 0;
    for (int q = 0; q < 4; q++) {
      const int dg = by * 4 + q;
      if (dg >= ndig) break;
      const int bi = dg < k ? (k - 1 - dg) : dg;
      v |= (((u32)(u8)seq[bi] >> 1) & 3u) << (2 * q);
    }
    o[1 + by] = (u8)v;
  }
}

// LUT[x] = min(x, revcomp_m(x)), 4^m-1 if it contains AA except as prefix (Model.hpp:1040-1064, 1220-1251)
__global__ void k_minimizer_lut(int m, u32* __restrict__ lut)
{
  const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 n = 1u << (2 * m);
  if (x >= n) return;
  u32 rc = 0, t = x;
  for (int i = 0; i < m; i++) { rc = (rc << 2) | ((t & 3u) ^ 2u); t >>= 2; }
  u32 v = rc < x ? rc : x;
  const u64 mask_ma1 = 0x5555555555555555ULL & ((1ULL << ((m - 2) * 2)) - 1);
  u64 a1 = v; a1 = ~(a1 | (a1 >> 2)); a1 = ((a1 >> 1) & a1) & mask_ma1;
  lut[x] = a1 ? (n - 1) : v;
}

}  // namespace kmx

using namespace kmx;

extern "C" int kmx_superk_partition(kmx_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n_seqs,
                                    uint32_t k, uint32_t m, const uint16_t* repart, uint32_t nb_parts,
                                    uint8_t** out_bytes, uint64_t* out_len, uint64_t* out_kmers)
{
  if (!ctx) return KMX_E_INVAL;
  if (!offsets || !repart || !out_bytes || !out_len || !out_kmers || nb_parts == 0 || nb_parts > 65535)
    return ctx->fail(KMX_E_INVAL, "kmx_superk_partition: bad argument");
  if (k < 8 || k > 63 || m < 4 || m > 15 || m > k) return ctx->fail(KMX_E_UNSUPPORTED, "k outside 8..63 or minimizer size outside 4..15");
  for (u32 p = 0; p < nb_parts; p++) { out_bytes[p] = nullptr; out_len[p] = 0; out_kmers[p] = 0; }
  if (n_seqs == 0) { for (u32 p = 0; p < nb_parts; p++) out_bytes[p] = (uint8_t*)malloc(1); return KMX_OK; }
  const u64 total_bases = offsets[n_seqs];
  if (total_bases >= 0xFFFFFF00ULL || n_seqs >= 0x7FFFFFFFULL) return ctx->fail(KMX_E_UNSUPPORTED, "batch of 4 Gbases or more: split it");
  if (total_bases && !bases) return ctx->fail(KMX_E_INVAL, "null bases");
  KMX_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int span_bits = (int)((k + 31) / 32) * 64;
  int maxs = (span_bits - 8) / 2; if (maxs > 255) maxs = 255;   // Sequence2SuperKmer.hpp:146
  const u64 nm = 1ULL << (2 * m);

  char* d_bases = (char*)ctx->dalloc(total_bases + 16);
  u64* d_offs = (u64*)ctx->dalloc((n_seqs + 1) * 8);
  u32* d_lut = (u32*)ctx->dalloc(nm * 4);
  u16* d_rep = (u16*)ctx->dalloc(nm * 2);
  u32* d_cnt = (u32*)ctx->dalloc((n_seqs + 1) * 4);
  u32* d_doff = (u32*)ctx->dalloc((n_seqs + 1) * 4);
  std::vector<void*> blocks = {d_bases, d_offs, d_lut, d_rep, d_cnt, d_doff};
  auto release = [&]() { for (void* b : blocks) ctx->dfree(b); };
  for (void* b : blocks) if (!b) { release(); return ctx->fail(KMX_E_NOMEM, "superk: device allocation failed"); }
  auto fail = [&](hipError_t e, const char* what) { release(); return ctx->fail(KMX_E_HIP, std::string(what) + ": " + hipGetErrorString(e)); };
  StageClock clk(st, "superk_partition");
  hipError_t e;
  if ((e = hipMemcpyAsync(d_bases, bases, total_bases, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e, "upload bases");
  if ((e = hipMemcpyAsync(d_offs, offsets, (n_seqs + 1) * 8, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e, "upload offsets");
  if ((e = hipMemcpyAsync(d_rep, repart, nm * 2, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e, "upload repartition");
  hipLaunchKernelGGL(k_minimizer_lut, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, st, (int)m, d_lut);
  if ((e = hipMemsetAsync(d_cnt, 0, (n_seqs + 1) * 4, st)) != hipSuccess) return fail(e, "memset");
  const dim3 g1((unsigned)((n_seqs + 127) / 128)), b1(128);
  hipLaunchKernelGGL((k_superk_scan<false>), g1, b1, 0, st, d_bases, d_offs, (u64)n_seqs, (int)k, (int)m, maxs, d_lut, d_rep, d_cnt,
                     (const u32*)nullptr, (SkDesc*)nullptr);
  size_t tb = 0;
  if ((e = rocprim::exclusive_scan(nullptr, tb, d_cnt, d_doff, 0u, (size_t)n_seqs + 1, rocprim::plus<u32>(), st)) != hipSuccess) return fail(e, "scan size");
  void* d_tmp = ctx->dalloc(tb ? tb : 256); blocks.push_back(d_tmp);
  if (!d_tmp) { release(); return ctx->fail(KMX_E_NOMEM, "superk: device allocation failed"); }
  if ((e = rocprim::exclusive_scan(d_tmp, tb, d_cnt, d_doff, 0u, (size_t)n_seqs + 1, rocprim::plus<u32>(), st)) != hipSuccess) return fail(e, "scan");
  u32 nd = 0;
  if ((e = hipMemcpyAsync(&nd, d_doff + n_seqs, 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return fail(e, "memcpy");
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "sync");
  clk.mark("upload+scan");
  if (nd == 0) { release(); for (u32 p = 0; p < nb_parts; p++) out_bytes[p] = (uint8_t*)malloc(1); return KMX_OK; }

  SkDesc* d_desc = (SkDesc*)ctx->dalloc((size_t)nd * sizeof(SkDesc));
  u16* d_keys = (u16*)ctx->dalloc((size_t)nd * 2), *d_keys2 = (u16*)ctx->dalloc((size_t)nd * 2);
  u32* d_ids = (u32*)ctx->dalloc((size_t)nd * 4), *d_ids2 = (u32*)ctx->dalloc((size_t)nd * 4);
  u32* d_sz = (u32*)ctx->dalloc((size_t)nd * 4);
  u64* d_szs = (u64*)ctx->dalloc(((size_t)nd + 1) * 8), *d_boff = (u64*)ctx->dalloc(((size_t)nd + 1) * 8);
  for (void* b : {(void*)d_desc, (void*)d_keys, (void*)d_keys2, (void*)d_ids, (void*)d_ids2, (void*)d_sz, (void*)d_szs, (void*)d_boff}) blocks.push_back(b);
  for (void* b : blocks) if (!b) { release(); return ctx->fail(KMX_E_NOMEM, "superk: device allocation failed"); }
  hipLaunchKernelGGL((k_superk_scan<true>), g1, b1, 0, st, d_bases, d_offs, (u64)n_seqs, (int)k, (int)m, maxs, d_lut, d_rep, d_cnt,
                     (const u32*)d_doff, d_desc);
  const dim3 g2((nd + 255) / 256), b2(256);
  hipLaunchKernelGGL(k_superk_sizes, g2, b2, 0, st, d_desc, nd, (int)k, d_keys, d_ids, d_sz);
  size_t tb2 = 0, tb3 = 0;
  if ((e = rocprim::radix_sort_pairs(nullptr, tb2, d_keys, d_keys2, d_ids, d_ids2, (size_t)nd, 0, 16, st)) != hipSuccess) return fail(e, "sort size");
  if ((e = rocprim::exclusive_scan(nullptr, tb3, d_szs, d_boff, (u64)0, (size_t)nd + 1, rocprim::plus<u64>(), st)) != hipSuccess) return fail(e, "scan size");
  void* d_tmp2 = ctx->dalloc(std::max(tb2, tb3) + 256); blocks.push_back(d_tmp2);
  if (!d_tmp2) { release(); return ctx->fail(KMX_E_NOMEM, "superk: device allocation failed"); }
  if ((e = rocprim::radix_sort_pairs(d_tmp2, tb2, d_keys, d_keys2, d_ids, d_ids2, (size_t)nd, 0, 16, st)) != hipSuccess) return fail(e, "sort");
  if ((e = hipMemsetAsync(d_szs + nd, 0, 8, st)) != hipSuccess) return fail(e, "memset");
  hipLaunchKernelGGL(k_superk_gather_sizes2, g2, b2, 0, st, d_ids2, d_sz, d_desc, nd, d_szs);
  if ((e = rocprim::exclusive_scan(d_tmp2, tb3, d_szs, d_boff, (u64)0, (size_t)nd + 1, rocprim::plus<u64>(), st)) != hipSuccess) return fail(e, "scan");
  u64* d_pp = (u64*)ctx->dalloc(((size_t)nb_parts + 1) * 8); u32* d_pf = (u32*)ctx->dalloc(((size_t)nb_parts + 1) * 4);
  blocks.push_back(d_pp); blocks.push_back(d_pf);
  if (!d_pp || !d_pf) { release(); return ctx->fail(KMX_E_NOMEM, "superk: device allocation failed"); }
  hipLaunchKernelGGL(k_superk_part_bounds, dim3((nb_parts + 256) / 256), dim3(256), 0, st, d_keys2, nd, nb_parts, d_boff, d_pp, d_pf);
  std::vector<u64> pp((size_t)nb_parts + 1); std::vector<u32> pf((size_t)nb_parts + 1);
  u64 tot = 0;
  if ((e = hipMemcpyAsync(&tot, d_boff + nd, 8, hipMemcpyDeviceToHost, st)) != hipSuccess ||
      (e = hipMemcpyAsync(pp.data(), d_pp, ((size_t)nb_parts + 1) * 8, hipMemcpyDeviceToHost, st)) != hipSuccess ||
      (e = hipMemcpyAsync(pf.data(), d_pf, ((size_t)nb_parts + 1) * 4, hipMemcpyDeviceToHost, st)) != hipSuccess ||
      (e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "sync");
  clk.mark("emit+sort");
  if (pf[nb_parts] != nd) { release(); return ctx->fail(KMX_E_INVAL, "repartition table names a partition >= nb_parts"); }
  const u64 total_bytes = tot & 0xFFFFFFFFULL;
  u8* d_out = (u8*)ctx->dalloc(total_bytes + 16); blocks.push_back(d_out);
  u8* h_out = (u8*)ctx->halloc(total_bytes + 16);
  if (!d_out || !h_out) { ctx->hfree(h_out); release(); return ctx->fail(KMX_E_NOMEM, "superk: allocation failed"); }
  hipLaunchKernelGGL(k_superk_pack, g2, b2, 0, st, d_bases, d_desc, d_ids2, d_boff, nd, (int)k, d_out);
  if ((e = hipGetLastError()) != hipSuccess) { ctx->hfree(h_out); return fail(e, "k_superk_pack"); }
  clk.mark("pack");
  // the partition-ordered stream comes back in one copy; a few host threads cut it into the per-partition buffers
  if ((e = hipMemcpyAsync(h_out, d_out, total_bytes, hipMemcpyDeviceToHost, st)) != hipSuccess ||
      (e = hipStreamSynchronize(st)) != hipSuccess) { ctx->hfree(h_out); return fail(e, "download"); }
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
