// skf.hpp -- round 6: what the sync-free count path (kmx_count_reads_dev) shares between the split (superk.hip, superk_fast.hpp) and
// the count (count.hip): the descriptor of a super-k-mer, the control block on the device, the limits.
#pragma once
#include "kmx_dev.hpp"
namespace kmx {
struct SkDesc { u32 base; u16 part; u8 n; u8 pad; };   // base = index of the record's first base in `bases`
constexpr u32 SKF_MAXP = 1024;           // partitions the path takes (the per-wave tables of k_sk_scatter live in LDS)
constexpr u32 SKF_RPW = 16;              // reads per chunk (per wave of the walk)
constexpr u32 SKF_DK = 1024;             // = DK of count.hip's decode (k-mers per workgroup there)
enum { SKF_ST_PART = 1u,                 // a minimizer's partition is >= nb_parts (an error of the caller's table)
       SKF_ST_CAP = 2u,                  // more records than the sorted arrays were sized for
       SKF_ST_LAYOUT = 4u,               // a partition beyond the sample sort's limits: the library sort takes the batch
       SKF_ST_BUCKET = 8u };             // a bucket beyond the count kernels
struct SkfCtl {                          // device block, zeroed at the start of a call; the host reads it once
  u32 nd, total, bytes, TB, NC, status, n_big, overflow;      // (n_big: buckets beyond a wave's registers, listed for the LDS kernels; overflow: one beyond those too)
};
struct SkfLayout { u32 target, chunk, maxb, sample; };      // the sample sort's limits for the key type of the call
__host__ __device__ __forceinline__ u32 skf_rec_bytes(u32 k, u32 n) { return 1u + (k + n - 1u + 3u) / 4u; }
inline u32 skf_wpg(u32 P) { return P <= 256 ? 16u : P <= 512 ? 8u : 4u; }      // chunks (waves) per row: 12 bytes of LDS per (wave, partition)

constexpr u32 SKF_BIG_CAP = 8192;        // listed buckets (beyond: the old path)
}  // namespace kmx
