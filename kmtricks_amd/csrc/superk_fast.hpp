// superk_fast.hpp -- round 6: the split of kmx_count_reads_dev without a host round trip (included by superk.hip, inside namespace kmx).
//
// The old path walks the reads twice (count, emit), has the library sort the descriptors by partition, gathers their sizes, scans
// them with the library, and reads three sizes back on the way (the host shapes the next launch from each).  Here:
//   k_superk_wave<.., CH>  ONE walk: a wave takes SKF_RPW consecutive reads (a chunk) and writes their descriptors back to back from slot
//                 offsets[first read of the chunk] on -- a read holds fewer super-k-mers than bases, so the chunks' slot ranges never
//                 meet and nothing has to be counted first;
//   k_sk_hist     a workgroup per ROW (wpg chunks, a wave each): (records, k-mers << 32 | record bytes) per partition -> T[row][p];
//   k_sk_scan     T's prefix along the rows and over the partitions: a workgroup sums its rows per partition, publishes the sums,
//                 waits for every workgroup's (at most 128 workgroups of 256 threads: all resident), adds up those in front of it.
//                 Workgroup 0 also leaves the partitions' bounds, the totals and the layout of the partition-local sample sort that
//                 follows (count_sort.hpp: buckets and walk chunks per partition) -- what the host used to compute from a read-back;
//   k_sk_scatter  a workgroup per row again: its waves count their chunks once more, take their places inside the row in wave order,
//                 then walk their chunks 64 records at a time: a record's place among the wave's records of its partition from a match
//                 over the partition's bits (ballots), the sizes of those in front of it by shuffles.  Out: per sorted record its first
//                 base, its prefix (k-mers << 32 | bytes over ALL records in front of it: what a gather + a library scan made), its slot
//                 (for k_part_stats), its partition (hash mode), and the first record of every block of DK k-mers (k_decode_block_starts).
// A stable counting sort, so inside a partition the records keep read order -- the order the reference appends them in
// (fill_partitions.hpp:59-105), which SuperKmerBinInfoFile's numbers depend on (io/superk_storage.hpp:205-225).
// Everything the host wants (sizes, status) sits in one control block it reads once, at the end of the call; a status bit (a table that
// names a partition >= nb_parts, more records than estimated, a partition or bucket beyond the sample sort) sends the call down the old path.
#pragma once

__global__ __launch_bounds__(1024)
void k_sk_hist(const SkDesc* __restrict__ desc, const u64* __restrict__ offsets, const u32* __restrict__ ccnt, u32 n_chunks, u32 wpg, u32 P, u32 k,
               ulonglong2* __restrict__ T, SkfCtl* __restrict__ ctl)
{
  extern __shared__ u64 sk_lds[];
  u64* hsz = sk_lds;                                  // [P]
  u32* hcnt = reinterpret_cast<u32*>(hsz + P);        // [P]
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, row = blockIdx.x;
  for (u32 p = tid; p < P; p += blockDim.x) { hsz[p] = 0; hcnt[p] = 0; }
  __syncthreads();
  const u32 chunk = row * wpg + wave;
  if (wave < wpg && chunk < n_chunks) {
    const u32 base = (u32)offsets[(u64)chunk * SKF_RPW], n = ccnt[chunk];
    bool bad = false;
    for (u32 j = lane; j < n; j += 64) {
      const SkDesc d = desc[base + j];
      if (d.part < P) { atomicAdd(&hcnt[d.part], 1u); atomicAdd((unsigned long long*)&hsz[d.part], ((u64)d.n << 32) | skf_rec_bytes(k, d.n)); }
      else bad = true;
    }
    if (bad) atomicOr(&ctl->status, (u32)SKF_ST_PART);
  }
  __syncthreads();
  for (u32 p = tid; p < P; p += blockDim.x) T[(u64)row * P + p] = make_ulonglong2((u64)hcnt[p], hsz[p]);
}

// exclusive scan, in place, of the P (count, sizes) pairs in c[] / s[] (LDS; a thread takes `per` consecutive entries); the totals
// land in c[P] / s[P].  Ends with a barrier.
__device__ __forceinline__ void skf_scan_pairs(u32* c, u64* s, u32 P, u32* wsum_c, u64* wsum_s)
{
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, nw = blockDim.x >> 6;
  const u32 per = (P + blockDim.x - 1) / blockDim.x, i0 = tid * per;
  u32 mc = 0; u64 ms = 0;
  for (u32 x = 0; x < per; x++) if (i0 + x < P) { mc += c[i0 + x]; ms += s[i0 + x]; }
  u32 ic = mc; u64 is = ms;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const u32 tc = (u32)__shfl_up((int)ic, off); const u64 ts = (u64)__shfl_up((unsigned long long)is, off);
    if ((int)lane >= off) { ic += tc; is += ts; }
  }
  if (lane == 63) { wsum_c[wave] = ic; wsum_s[wave] = is; }
  __syncthreads();
  u32 ac = ic - mc; u64 as = is - ms; u32 allc = 0; u64 alls = 0;
  for (u32 w = 0; w < nw; w++) { if (w < wave) { ac += wsum_c[w]; as += wsum_s[w]; } allc += wsum_c[w]; alls += wsum_s[w]; }
  for (u32 x = 0; x < per; x++) if (i0 + x < P) { const u32 vc = c[i0 + x]; const u64 vs = s[i0 + x]; c[i0 + x] = ac; s[i0 + x] = as; ac += vc; as += vs; }
  if (tid == 0) { c[P] = allc; s[P] = alls; }
  __syncthreads();
}

__global__ __launch_bounds__(256)
void k_sk_scan(ulonglong2* __restrict__ T, u32 R, u32 rpg, u32 P, ulonglong2* __restrict__ agg /* [G][P] */, u32* __restrict__ flags /* [G], zeroed */,
               u32 nd_cap, SkfLayout L, SkfCtl* __restrict__ ctl, u32* __restrict__ pf /* [P + 1] */, u64* __restrict__ pp /* [P + 1] */, u64* __restrict__ boff,
               uint4* __restrict__ parts /* CsPart[P] */, u32* __restrict__ cfirst /* [P + 1] */)
{
  __shared__ u32 tot_c[SKF_MAXP + 1];      // records in front of partition p ([P]: all)
  __shared__ u64 tot_s[SKF_MAXP + 1];      // their sizes
  __shared__ u32 col_c[SKF_MAXP + 1];      // partition p: records of the rows in front of mine (later: the sample sort's first bucket)
  __shared__ u64 col_s[SKF_MAXP + 1];      // ... sizes (later: first walk chunk)
  __shared__ u32 nb_s[SKF_MAXP];
  __shared__ u32 wsum_c[4]; __shared__ u64 wsum_s[4];
  const u32 tid = threadIdx.x, g = blockIdx.x, G = gridDim.x;
  const u32 r0 = g * rpg, r1 = min(R, r0 + rpg);
  for (u32 p = tid; p < P; p += 256) {      // my rows' sums per partition
    u32 c = 0; u64 s = 0;
    for (u32 r = r0; r < r1; r++) { const ulonglong2 v = T[(u64)r * P + p]; c += (u32)v.x; s += v.y; }
    agg[(u64)g * P + p] = make_ulonglong2((u64)c, s);
  }
  __threadfence();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(&flags[g], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  for (u32 q = tid; q < G; q += 256) while (__hip_atomic_load(&flags[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(1);
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  for (u32 p = tid; p < P; p += 256) {
    u32 c = 0, bc = 0; u64 s = 0, bs = 0;
    for (u32 q = 0; q < G; q++) {
      const ulonglong2 v = agg[(u64)q * P + p];
      if (q < g) { bc += (u32)v.x; bs += v.y; }
      c += (u32)v.x; s += v.y;
    }
    tot_c[p] = c; tot_s[p] = s; col_c[p] = bc; col_s[p] = bs;
  }
  __syncthreads();
  skf_scan_pairs(tot_c, tot_s, P, wsum_c, wsum_s);
  const u32 nd = tot_c[P];
  const bool over = nd > nd_cap;
  // my rows' places: the partition's start + the rows in front of mine + my rows in front of the row
  for (u32 p = tid; p < P; p += 256) {
    u32 c = tot_c[p] + col_c[p]; u64 s = tot_s[p] + col_s[p];
    for (u32 r = r0; r < r1; r++) { const ulonglong2 v = T[(u64)r * P + p]; T[(u64)r * P + p] = make_ulonglong2((u64)c, s); c += (u32)v.x; s += v.y; }
  }
  if (g != 0) return;
  // workgroup 0: the partitions' bounds, the totals, the sample sort's layout
  for (u32 p = tid; p <= P; p += 256) { pf[p] = tot_c[p]; pp[p] = tot_s[p]; }
  if (tid == 0 && !over) boff[nd] = tot_s[P];      // (the prefix array's closing entry: the totals)
  __syncthreads();
  bool bad = false;
  for (u32 p = tid; p < P; p += 256) {      // k-mers of partition p -> its buckets (at least one) and walk chunks
    const u32 n = (u32)(tot_s[p + 1] >> 32) - (u32)(tot_s[p] >> 32);
    const u32 nb = max(1u, (n + L.target - 1) / L.target);
    if (nb > L.maxb || nb * 4u > L.sample) bad = true;
    nb_s[p] = nb; col_c[p] = nb; col_s[p] = (u64)((n + L.chunk - 1) / L.chunk);
  }
  __syncthreads();
  skf_scan_pairs(col_c, col_s, P, wsum_c, wsum_s);
  for (u32 p = tid; p < P; p += 256) {
    const u32 k0 = (u32)(tot_s[p] >> 32), n = (u32)(tot_s[p + 1] >> 32) - k0;
    parts[p] = make_uint4(k0, n, col_c[p], nb_s[p]);      // CsPart{key0, nkeys, bucket0, nb}
    cfirst[p] = (u32)col_s[p];
  }
  if (tid == 0) {
    cfirst[P] = (u32)col_s[P];
    ctl->nd = nd; ctl->total = (u32)(tot_s[P] >> 32); ctl->bytes = (u32)tot_s[P]; ctl->TB = col_c[P]; ctl->NC = (u32)col_s[P];
    if (over) atomicOr(&ctl->status, (u32)SKF_ST_CAP);
  }
  if (bad) atomicOr(&ctl->status, (u32)SKF_ST_LAYOUT);
}

__global__ __launch_bounds__(1024)
void k_sk_scatter(const SkDesc* __restrict__ desc, const u64* __restrict__ offsets, const u32* __restrict__ ccnt, u32 n_chunks, u32 wpg, u32 P, u32 pbits, u32 k,
                  const ulonglong2* __restrict__ T, const SkfCtl* __restrict__ ctl,
                  u32* __restrict__ sbase, u64* __restrict__ boff, u32* __restrict__ ids /* or null */, u16* __restrict__ part16 /* or null */, u32* __restrict__ blk_first,
                  const u32* __restrict__ slot_word = nullptr /* set: ids[pos] = slot_word[slot] (the descriptor's minimizer) instead of the slot */)
{
  extern __shared__ u64 sk_lds[];
  u64* wsz = sk_lds;                                           // [wpg][P]
  u32* wpos = reinterpret_cast<u32*>(wsz + (size_t)wpg * P);   // [wpg][P]
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, row = blockIdx.x;
  if (ctl->status & (SKF_ST_PART | SKF_ST_CAP)) return;      // (uniform: raised by kernels that are through)
  for (u32 i = tid; i < wpg * P; i += blockDim.x) { wsz[i] = 0; wpos[i] = 0; }
  __syncthreads();
  const u32 chunk = row * wpg + wave;
  const bool mine = wave < wpg && chunk < n_chunks;
  const u32 base = mine ? (u32)offsets[(u64)chunk * SKF_RPW] : 0u, n = mine ? ccnt[chunk] : 0u;
  u64* const msz = wsz + (size_t)(mine ? wave : 0u) * P; u32* const mpos = wpos + (size_t)(mine ? wave : 0u) * P;
  for (u32 j = lane; j < n; j += 64) {
    const SkDesc d = desc[base + j];
    atomicAdd(&mpos[d.part], 1u); atomicAdd((unsigned long long*)&msz[d.part], ((u64)d.n << 32) | skf_rec_bytes(k, d.n));
  }
  __syncthreads();
  for (u32 p = tid; p < P; p += blockDim.x) {      // the waves' places inside the row, in wave order
    const ulonglong2 b = T[(u64)row * P + p];
    u32 c = (u32)b.x; u64 s = b.y;
    for (u32 w = 0; w < wpg; w++) { const u32 vc = wpos[w * P + p]; const u64 vs = wsz[(size_t)w * P + p]; wpos[w * P + p] = c; wsz[(size_t)w * P + p] = s; c += vc; s += vs; }
  }
  __syncthreads();
  const u64 below = (1ULL << lane) - 1ULL;
  for (u32 j0 = 0; j0 < n; j0 += 64) {      // (n is the wave's: no workgroup barrier from here on)
    const u32 j = j0 + lane;
    const bool v = j < n;
    SkDesc d; d.base = 0; d.part = 0; d.n = 0; d.pad = 0;
    if (v) d = desc[base + j];
    const u32 key = d.part;
    u64 peers = __ballot(v);
    for (u32 b = 0; b < pbits; b++) { const bool bit = (key >> b) & 1u; const u64 bm = __ballot(v && bit); peers &= bit ? bm : ~bm; }
    const u64 sz = ((u64)d.n << 32) | skf_rec_bytes(k, d.n);
    u64 mlow = v ? (peers & below) : 0ULL, acc = 0;
    const u32 rank = (u32)__popcll(mlow);
    while (__ballot(mlow != 0)) {      // (uniform: every lane shuffles)
      const int src = mlow ? __builtin_ctzll(mlow) : (int)lane;
      const u64 t = (u64)__shfl((unsigned long long)sz, src);
      if (mlow) { acc += t; mlow &= mlow - 1ULL; }
    }
    u32 pos = 0; u64 pre = 0;
    if (v) { pos = mpos[key] + rank; pre = msz[key] + acc; }
    if (v && (peers >> lane) == 1ULL) { mpos[key] = pos + 1u; msz[key] = pre + sz; }      // the last of its partition in this step
    if (v) {
      sbase[pos] = d.base; boff[pos] = pre;
      if (ids) ids[pos] = slot_word ? slot_word[base + j] : base + j;
      if (part16) part16[pos] = (u16)key;
      const u32 ko = (u32)(pre >> 32), ke = ko + d.n, B = (ko + SKF_DK - 1u) / SKF_DK;
      if (B * SKF_DK < ke) blk_first[B] = pos;      // (a record holds at most 60 k-mers: at most one block of the decode starts inside it)
    }
  }
}
