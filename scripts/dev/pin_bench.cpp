// how fast does host memory get page-locked?  hipHostMalloc in blocks / in one slab, hipHostRegister of pages touched beforehand
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipSetDevice(0); void* d; hipMalloc(&d, 1 << 20);
  const size_t B = 40u << 20; const int N = 16;
  { std::vector<void*> p(N); double t0 = now(); for (int i = 0; i < N; i++) hipHostMalloc(&p[i], B, hipHostMallocDefault); double t = now() - t0;
    printf("hipHostMalloc %d x 40 MB: %.1f ms each (%.2f GB/s)\n", N, t / N * 1e3, N * B / t / 1e9); for (auto q : p) hipHostFree(q); }
  { void* p; double t0 = now(); hipHostMalloc(&p, (size_t)N * B, hipHostMallocDefault); double t = now() - t0; printf("hipHostMalloc 1 x %zu MB: %.1f ms (%.2f GB/s)\n", N * B >> 20, t * 1e3, N * B / t / 1e9); hipHostFree(p); }
  for (int huge = 0; huge < 2; huge++) {
    const size_t S = (size_t)N * B;
    double t0 = now();
    char* m = (char*)mmap(nullptr, S, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (huge) madvise(m, S, MADV_HUGEPAGE);
    std::vector<std::thread> th; const int T = 16;
    for (int t = 0; t < T; t++) th.emplace_back([&, t]() { for (size_t o = S / T * t; o < S / T * (t + 1); o += 4096) m[o] = 1; });
    for (auto& x : th) x.join();
    double t1 = now();
    hipError_t e = hipHostRegister(m, S, hipHostRegisterDefault);
    double t2 = now();
    printf("mmap%s + touch by %d threads: %.1f ms; hipHostRegister of %zu MB: %.1f ms (%.2f GB/s) %s\n", huge ? " (MADV_HUGEPAGE)" : "", T, (t1 - t0) * 1e3, S >> 20, (t2 - t1) * 1e3, S / (t2 - t1) / 1e9, hipGetErrorString(e));
    // registered pieces one by one
    hipHostUnregister(m);
    double t3 = now(); for (int i = 0; i < N; i++) hipHostRegister(m + (size_t)i * B, B, hipHostRegisterDefault); double t4 = now();
    printf("  hipHostRegister %d x 40 MB of touched pages: %.1f ms each\n", N, (t4 - t3) / N * 1e3);
    // an upload from it
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); void* dd; hipMalloc(&dd, B);
    hipEventRecord(a); hipMemcpyAsync(dd, m, B, hipMemcpyHostToDevice, 0); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    printf("  upload of 40 MB from it: %.2f ms (%.1f GB/s)\n", ms, B / ms / 1e6);
    for (int i = 0; i < N; i++) hipHostUnregister(m + (size_t)i * B);
    munmap(m, S); hipFree(dd);
  }
  return 0;
}
