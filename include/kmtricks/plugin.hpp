// kmtricks/plugin.hpp -- the merge-plugin interface (row operator) of kmtricks, as the kmx driver
// loads it.  The class must be binary compatible with reference include/kmtricks/plugin.hpp:12-30
// (same virtual-function order, same data members) because plugin .so files are built against it:
//   virtuals: ~dtor, set_out_dir (final), set_partition (final), set_kmer_size, configure,
//             process_kmer, process_hash;  members: std::string out dir, size_t k, size_t partition.
// A plugin exports extern "C": std::string plugin_name(); int use_template();
// km::IMergePlugin* create0() or create<MAX_K>(); void destroy(km::IMergePlugin*)
// (reference include/kmtricks/plugin_manager.hpp:38-113, plugins/example/*.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <kmtricks/kmer.hpp>

#define KMTRICKS_PUBLIC
#include <kmtricks/utils.hpp>

#ifndef DMAX_C
#define DMAX_C 4294967295
#endif

namespace km {

class IMergePlugin {
 public:
  using count_type = typename selectC<DMAX_C>::type;
  IMergePlugin() = default;
  virtual ~IMergePlugin() {}
  virtual void set_out_dir(const std::string& s) final { m_output_directory = s; }
  virtual void set_partition(size_t p) final { m_partition = p; }
  virtual void set_kmer_size(const size_t kmer_size) { m_kmer_size = kmer_size; }
  // --plugin-config string
  virtual void configure(const std::string&) {}
  // one call per merged row, ascending keys; may edit the counts; the return value decides
  // whether the row is written (it replaces the recurrence-min test, merge.hpp:252-257)
  virtual bool process_kmer(const uint64_t*, std::vector<count_type>&) { return true; }
  virtual bool process_hash(uint64_t, std::vector<count_type>&) { return true; }

 protected:
  std::string m_output_directory;
  size_t m_kmer_size;
  size_t m_partition;
};

}  // namespace km
