// A merge plugin in the style kmtricks documents (row filter with a threshold from --plugin-config):
// keeps a row only if every sample's abundance is >= threshold, and doubles sample 0's count of kept
// rows (so the tests can see edited counts being written).  Built against include/kmtricks/plugin.hpp.
#include <kmtricks/plugin.hpp>

class ThresholdPlugin : public km::IMergePlugin {
 public:
  void configure(const std::string& s) override { m_threshold = (unsigned)std::stoul(s); }
  bool process_kmer(const uint64_t*, std::vector<count_type>& v) override { return filter(v); }
  bool process_hash(uint64_t, std::vector<count_type>& v) override { return filter(v); }
 private:
  bool filter(std::vector<count_type>& v) {
    m_calls++;
    for (auto& c : v) if (c < m_threshold) return false;
    if (!v.empty()) v[0] *= 2;
    return true;
  }
  unsigned m_threshold {0};
  unsigned long m_calls {0};
};

extern "C" std::string plugin_name() { return "ThresholdPlugin"; }
extern "C" int use_template() { return 0; }
extern "C" km::IMergePlugin* create0() { return new ThresholdPlugin(); }
extern "C" void destroy(km::IMergePlugin* p) { delete p; }
