// kmx_repart.hpp -- minimizer -> partition tables.
//   * static:  table[m] = XXH64(&m, 4, 0) % P                  (reference include/kmtricks/repartition.hpp:45-56)
//   * sampled: what gatb's RepartitorAlgorithm builds when kmtricks runs without --static-repart
//     (gatb kmer/impl/RepartitionAlgorithm.cpp:395-496 sampling, kmer/impl/PartiInfo.cpp:48-103 computeDistrib):
//     the kx-mers per minimizer of a sample of the reads (counted on the GPU by kmx_superk_partition_stats) are
//     spread over the partitions largest bin first, each into the partition with the least load so far.
// computeDistrib orders the bins with std::sort and keeps the partitions in a std::priority_queue: how bins of equal
// size (nearly all minimizers have none) and partitions of equal load are ordered is whatever those two libstdc++
// algorithms do on this exact input, so the table is only reproduced by running the same two algorithms on the same
// sequence of elements -- which is what this file does (checked against the reference's committed
// tests/data/repart_gatb/repartition.minimRepart).
#pragma once
#include <algorithm>
#include <cstdint>
#include <queue>
#include <utility>
#include <vector>

namespace kmxio {

inline uint64_t xxh64_u32(uint32_t v)
{ // XXH64(&v, 4, seed 0)
  const uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P5 = 0x27D4EB2F165667C5ULL;
  uint64_t h = P5 + 4;
  h ^= (uint64_t)v * P1; h = ((h << 23) | (h >> 41)) * P2 + P3;
  h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
  return h;
}
inline std::vector<uint16_t> repart_static(uint32_t msize, uint32_t nb_parts)
{
  std::vector<uint16_t> t(1ULL << (2 * msize));
  for (uint64_t m = 0; m < t.size(); m++) t[m] = (uint16_t)(xxh64_u32((uint32_t)m) % nb_parts);
  return t;
}

// Repartitor::computeDistrib (gatb PartiInfo.cpp:48-103; element types and comparators PartiInfo.hpp:405-428)
inline std::vector<uint16_t> repart_from_kxmers(const uint64_t* kxmers_per_minim, uint64_t nb_minims, uint32_t nb_parts)
{
  typedef std::pair<uint64_t, uint64_t> ipair;                       // (bin size, minimizer)
  struct itriple { uint64_t first, second, third; };                 // (partition, load, minimizers in it)
  struct by_size { bool operator()(ipair l, ipair r) const { return l.first > r.first; } };
  struct by_load { bool operator()(itriple l, itriple r) const { return l.second > r.second; } };
  std::vector<ipair> bins; bins.reserve(nb_minims);
  for (uint64_t i = 0; i < nb_minims; i++) bins.push_back(ipair(kxmers_per_minim[i], i));
  std::priority_queue<itriple, std::vector<itriple>, by_load> pq;
  for (uint32_t j = 0; j < nb_parts; j++) pq.push(itriple{j, 0, 0});
  std::sort(bins.begin(), bins.end(), by_size());
  std::vector<uint16_t> table(nb_minims, 0);
  for (uint64_t c = 0; c < nb_minims; c++) {
    itriple s = pq.top(); pq.pop();                                  // the emptiest partition takes the largest bin left
    table[bins[c].second] = (uint16_t)s.first;
    s.second += bins[c].first; s.third++;
    pq.push(s);
  }
  return table;
}

}  // namespace kmxio
