// back-to-back device-to-host copies on one stream: what does a copy cost beside its bytes?  page-locked memory from hipHostMalloc
// and from a transparent-huge-page mapping + hipHostRegister (what kmx_alloc_pinned makes)
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipSetDevice(0);
  const size_t TOT = (size_t)4 << 30;
  char* d; hipMalloc(&d, TOT); hipMemset(d, 1, TOT);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (int kind = 0; kind < 2; kind++) {
    char* h = nullptr;
    if (kind == 0) hipHostMalloc((void**)&h, TOT, hipHostMallocDefault);
    else { h = (char*)mmap(nullptr, TOT, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); madvise(h, TOT, MADV_HUGEPAGE); for (size_t o = 0; o < TOT; o += 4096) h[o] = 0; hipHostRegister(h, TOT, hipHostRegisterPortable); }
    for (size_t piece : {(size_t)32 << 20, (size_t)128 << 20, (size_t)512 << 20}) {
      for (int mode = 0; mode < 2; mode++) {      // 0: all queued, one wait; 1: a wait per copy
        hipStreamSynchronize(st);
        const double t0 = now(); double issue = 0;
        for (size_t o = 0; o < TOT; o += piece) { const double a = now(); hipMemcpyAsync(h + o, d + o, piece, hipMemcpyDeviceToHost, st); issue += now() - a; if (mode) hipStreamSynchronize(st); }
        hipStreamSynchronize(st);
        const double t = now() - t0;
        printf("%s pieces of %4zu MB, %s: %.1f GB/s (%.2f ms per piece, %.3f ms of it in the call)\n", kind ? "THP+register" : "hipHostMalloc", piece >> 20, mode ? "a wait per copy" : "queued, one wait ",
               TOT / t / 1e9, t / (TOT / piece) * 1e3, issue / (TOT / piece) * 1e3);
      }
    }
  }
  return 0;
}
