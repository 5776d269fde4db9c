// tests/helpers/reader_equiv.cpp -- SeqReader::next and SeqReader::next_view (kmtricks_amd/host/kmx_io.hpp) must cut a file into the same records (tests/test_tools_cpu.py)
#include "kmx_io.hpp"
#include <cstdio>
#include <cassert>
using namespace kmxio;
int main(int argc, char** argv) {
  // two passes over the same file: next() and next_view() must give the same records
  std::vector<std::string> a, b;
  { SeqReader r(argv[1]); std::string s; while (r.next(s)) a.push_back(s); }
  { SeqReader r(argv[1]); std::string t; const char* p; size_t n; while (r.next_view(t, p, n)) b.emplace_back(p, n); }
  printf("%zu %zu %s\n", a.size(), b.size(), a == b ? "same" : "DIFFERENT");
  return a == b ? 0 : 1;
}
