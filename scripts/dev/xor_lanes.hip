// check of the lane exchanges kmx_dev.hpp builds from DPP moves and the gfx950 permlane swaps: lane l must read lane l ^ M
// (hipcc --offload-arch=gfx950 xor_lanes.hip -o /tmp/xl && /tmp/xl)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../kmtricks_amd/csrc/kmx_dev.hpp"
using namespace kmx;
__global__ void k(u32* o)
{
  const u32 v = threadIdx.x * 3u + 7u;
  o[threadIdx.x] = xor_lane_u32<1>(v); o[64 + threadIdx.x] = xor_lane_u32<2>(v); o[128 + threadIdx.x] = xor_lane_u32<4>(v);
  o[192 + threadIdx.x] = xor_lane_u32<8>(v); o[256 + threadIdx.x] = xor_lane_u32<16>(v); o[320 + threadIdx.x] = xor_lane_u32<32>(v);
}
int main()
{
  u32* d; hipMalloc((void**)&d, 384 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  u32 h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0; const int ms[6] = {1, 2, 4, 8, 16, 32};
  for (int q = 0; q < 6; q++) for (int l = 0; l < 64; l++) if (h[q * 64 + l] != (u32)((l ^ ms[q]) * 3 + 7)) { if (!bad) printf("xor %d lane %d: got %u\n", ms[q], l, h[q * 64 + l]); bad++; }
  printf(bad ? "xor_lanes: %d WRONG\n" : "xor_lanes: ok\n", bad);
  return bad != 0;
}
