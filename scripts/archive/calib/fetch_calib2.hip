// Does a 48-byte access in the first half of a 128-byte line pull 64 or 128 bytes over the fabric?
// Every 128-byte line of a 2 GiB buffer is touched once, only bytes [0, 48) of it (4 lanes x dwordx3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x3 __attribute__((ext_vector_type(3), aligned(4)));
__global__ void k_calib_half(const uint8_t* buf, uint64_t lines, uint32_t* out)
{
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, g = t >> 2, r = t & 3;
  const uint64_t ngroups = (uint64_t)gridDim.x * blockDim.x / 4;
  uint32_t acc = 0;
  for (uint64_t l = g; l < lines; l += ngroups) { const u32x3 v = *(const u32x3*)(buf + l * 128 + r * 12); acc += v.x ^ v.y ^ v.z; }
  if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
  const uint64_t lines = 16ull << 20;   // 2 GiB
  uint8_t* d; uint32_t* o;
  hipMalloc(&d, lines * 128); hipMalloc(&o, 4);
  hipMemset(d, 1, lines * 128);
  for (int it = 0; it < 3; it++) hipLaunchKernelGGL(k_calib_half, dim3(4096), dim3(256), 0, 0, d, lines, o);
  hipDeviceSynchronize();
  printf("lines touched per launch: %llu (x64 = %llu, x128 = %llu bytes)\n", (unsigned long long)lines, (unsigned long long)lines * 64, (unsigned long long)lines * 128);
  return 0;
}
