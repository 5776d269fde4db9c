// Read bandwidth with the merge kernels' access pattern (4 adjacent lanes x dwordx3 per list, lists streamed
// front to back, 8 loads in flight per lane) for a working set in HBM (2 GiB) and one that fits the 256 MiB
// Infinity Cache (96 MiB, re-read 20 times): what does a re-fetched line cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x3 __attribute__((ext_vector_type(3), aligned(4)));
__global__ __launch_bounds__(256) void k_bw(const uint8_t* buf, uint64_t bytes_per_list, uint32_t lists, uint32_t* out)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, li = t >> 2, r = t & 3;
  if (li >= lists) return;
  const uint8_t* base = buf + (uint64_t)li * bytes_per_list;
  const uint32_t nrec = (uint32_t)(bytes_per_list / 12);
  uint32_t acc = 0;
  for (uint32_t i = r; i + 28 < nrec; i += 32) {
    u32x3 v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = *(const u32x3*)(base + (uint64_t)(i + 4 * q) * 12);
#pragma unroll
    for (int q = 0; q < 8; q++) acc += v[q].x ^ v[q].y ^ v[q].z;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
static float run(const uint8_t* d, uint64_t bpl, uint32_t lists, uint32_t* o, int reps)
{
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k_bw, dim3(lists * 4 / 256), dim3(256), 0, 0, d, bpl, lists, o);
  hipEventRecord(a);
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_bw, dim3(lists * 4 / 256), dim3(256), 0, 0, d, bpl, lists, o);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main()
{
  uint8_t* d; uint32_t* o; hipMalloc(&d, 2ull << 30); hipMalloc(&o, 4); hipMemset(d, 1, 2ull << 30);
  const uint64_t bpl = 65520;
  { const uint32_t lists = 32768; const float ms = run(d, bpl, lists, o, 5);
    printf("HBM   : %.1f MB in %.3f ms = %.0f GB/s\n", lists * bpl / 1e6, ms, lists * bpl / ms / 1e6); }
  { const uint32_t lists = 1536; const float ms = run(d, bpl, lists, o, 20);
    printf("L3-fit: %.1f MB in %.3f ms = %.0f GB/s (only %u lists = %u waves in flight)\n", lists * bpl / 1e6, ms, lists * bpl / ms / 1e6, lists, lists * 4 / 64); }
  { const uint64_t bpl2 = 3072; const uint32_t lists = 32768; const float ms = run(d, bpl2, lists, o, 20);
    printf("L3-fit, many short lists: %.1f MB in %.3f ms = %.0f GB/s\n", lists * bpl2 / 1e6, ms, lists * bpl2 / ms / 1e6); }
  return 0;
}
