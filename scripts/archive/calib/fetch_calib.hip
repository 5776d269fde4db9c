// Calibration of rocprofv3 FETCH_SIZE for the merge kernels' read pattern: 12-byte records, 4 adjacent
// lanes read 48 contiguous bytes of one list, every list is read exactly once front to back.
// Known traffic: LISTS * BYTES_PER_LIST (2 GiB, far past the 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x3 __attribute__((ext_vector_type(3), aligned(4)));
__global__ void k_calib(const uint8_t* buf, uint64_t bytes_per_list, uint32_t lists, uint32_t* out)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, li = t >> 2, r = t & 3;
  if (li >= lists) return;
  const uint8_t* base = buf + (uint64_t)li * bytes_per_list;
  const uint32_t nrec = (uint32_t)(bytes_per_list / 12);
  uint32_t acc = 0;
  for (uint32_t i = r; i < nrec; i += 4) { const u32x3 v = *(const u32x3*)(base + (uint64_t)i * 12); acc += v.x ^ v.y ^ v.z; }
  if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
  const uint32_t lists = 32768; const uint64_t bpl = 65520;   // multiple of 12
  uint8_t* d; uint32_t* o;
  hipMalloc(&d, (size_t)lists * bpl); hipMalloc(&o, 4);
  hipMemset(d, 1, (size_t)lists * bpl);
  for (int it = 0; it < 3; it++) hipLaunchKernelGGL(k_calib, dim3(lists * 4 / 256), dim3(256), 0, 0, d, bpl, lists, o);
  hipDeviceSynchronize();
  printf("known bytes per launch: %llu\n", (unsigned long long)lists * bpl);
  return 0;
}
