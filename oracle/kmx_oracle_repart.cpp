/*
 * kmx_oracle_repart.cpp -- CPU restatement of gatb's Repartitor::computeDistrib.  TEST INFRASTRUCTURE ONLY (see
 * kmx_oracle.h): loaded by tests/ as the checker of `kmx pipeline`'s sampled repartition, never by the product.
 *
 * C++ and not C like the rest of the oracle, for one reason: the reference orders the minimizer bins with std::sort
 * and keeps the partitions in a std::priority_queue (gatb-core-stripped src/gatb/kmer/impl/PartiInfo.cpp:48-103, element
 * types and comparators PartiInfo.hpp:405-428).  Neither is stable, nearly all bins tie at size 0 and the partitions tie
 * at equal load, so WHICH table comes out is defined by what libstdc++'s introsort and heap do with this exact element
 * sequence; a restatement has to run the same two library algorithms.  Parity: PINNED by the reference's committed
 * tests/data/repart_gatb/repartition.minimRepart (4 partitions, m = 10, its two test FASTA files): tests/test_oracle_goldens.py
 * feeds this function the kx-mers per minimizer the C oracle counts on those files and compares all 4^10 entries.
 */
#include <algorithm>
#include <cstdint>
#include <queue>
#include <utility>
#include <vector>

namespace {
typedef std::pair<uint64_t, uint64_t> ipair;                 /* PartiInfo.hpp:405 : bin size, bin number */
struct itriple { uint64_t first, second, third; };           /* PartiInfo.hpp:407-416 : partition, space used, bins in it */
struct comp_bins { bool operator()(ipair l, ipair r) { return l.first > r.first; } };              /* :418-420 */
struct compSpaceTriple { bool operator()(itriple l, itriple r) { return l.second > r.second; } };  /* :426-428 */
}

extern "C" int orc_repart_sampled(const uint64_t* nb_kxmers_per_minim, uint64_t nb_minims, uint32_t nbpart, uint16_t* repart_table)
{ /* PartiInfo.cpp:48-103 */
  if (!nb_kxmers_per_minim || !repart_table || nbpart == 0 || nbpart > 65535) return -1;
  std::vector<ipair> bin_size_vec;
  std::priority_queue<itriple, std::vector<itriple>, compSpaceTriple> pq;
  for (uint64_t ii = 0; ii < nb_minims; ii++) bin_size_vec.push_back(ipair(nb_kxmers_per_minim[ii], ii));   /* :59-65 */
  for (uint32_t jj = 0; jj < nbpart; jj++) { itriple t = {jj, 0, 0}; pq.push(t); }                           /* :70 */
  std::sort(bin_size_vec.begin(), bin_size_vec.end(), comp_bins());                                          /* :73 */
  uint64_t cur_minim = 0;
  while (cur_minim < nb_minims) {                                                                             /* :85-102 */
    itriple smallest_parti = pq.top(); pq.pop();
    repart_table[bin_size_vec[cur_minim].second] = (uint16_t)smallest_parti.first;
    smallest_parti.second += bin_size_vec[cur_minim].first;
    smallest_parti.third++;
    pq.push(smallest_parti);
    cur_minim++;
  }
  return 0;
}
