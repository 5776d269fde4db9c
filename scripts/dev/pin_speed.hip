// how fast can 32 MB of host memory be page-locked: hipHostMalloc against an anonymous mapping with transparent huge pages + hipHostRegister
// (hipcc --offload-arch=gfx950 pin_speed.hip -o /tmp/pin && /tmp/pin)
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
  const size_t n = 32u << 20;
  (void)hipFree(nullptr);
  for (int rep = 0; rep < 3; rep++) {
    double t0 = now(); void* p = nullptr; (void)hipHostMalloc(&p, n, hipHostMallocDefault); double t1 = now();
    printf("hipHostMalloc 32 MB: %.2f ms\n", (t1 - t0) * 1e3);
    t0 = now(); void* q = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); madvise(q, n, MADV_HUGEPAGE); memset(q, 0, n); t1 = now();
    const hipError_t e = hipHostRegister(q, n, hipHostRegisterPortable); double t2 = now();
    printf("mmap + MADV_HUGEPAGE + touch: %.2f ms, hipHostRegister: %.2f ms (%s)\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, hipGetErrorString(e));
    t0 = now(); void* r = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0); t1 = now();
    const hipError_t e2 = hipHostRegister(r, n, hipHostRegisterDefault); t2 = now();
    printf("mmap MAP_POPULATE (4 KB pages): %.2f ms, hipHostRegister: %.2f ms (%s)\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, hipGetErrorString(e2));
    // does a registered buffer copy as fast?
    void* d = nullptr; (void)hipMalloc(&d, n);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (void* h : {p, q, r}) { hipEventRecord(a); (void)hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, 0); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); printf("  D2H 32 MB: %.3f ms (%.1f GB/s)\n", ms, n / ms / 1e6); }
  }
  FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r"); char buf[128] = {0}; if (f) { fgets(buf, 127, f); fclose(f); } printf("THP: %s\n", buf);
  return 0;
}
