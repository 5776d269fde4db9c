// global atomic adds on random words of a 1.3 MB table: device scope (what atomicAdd is) against workgroup scope (performed in the
// XCD's own L2 -- only right when no other XCD touches the table), and the XCC id a workgroup sees
// (hipcc --offload-arch=gfx950 atomic_scope.hip -o /tmp/as && /tmp/as)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32;
template <int SCOPE> __global__ void k(u32* tab, u32 words, u32 per_thread, u32 copies_stride)
{
  u32 x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  u32 xcc = 0;
  if (copies_stride) { xcc = __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | (3 << 11)) & 7u; }
  u32* t = tab + (size_t)xcc * copies_stride;
  for (u32 i = 0; i < per_thread; i++) {
    x = x * 1664525u + 1013904223u;
    __hip_atomic_fetch_add(&t[(x >> 8) % words], 1u, __ATOMIC_RELAXED, SCOPE);
  }
}
__global__ void k_xcc(u32* out) { if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; }
int main()
{
  const u32 words = 327680, per = 12, blocks = 1000, tpb = 256;      // 3 M atomics
  u32* d; hipMalloc((void**)&d, (size_t)words * 4 * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](const char* name, auto kern, u32 stride) {
    float best = 1e9f;
    for (int r = 0; r < 5; r++) { hipMemset(d, 0, (size_t)words * 4 * 8); hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(blocks), dim3(tpb), 0, 0, d, words, per, stride); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    printf("%-44s %.1f us for %.1f M atomics\n", name, best * 1e3, blocks * tpb * per / 1e6);
  };
  run("device scope, one table", k<__HIP_MEMORY_SCOPE_AGENT>, 0);
  run("workgroup scope, one table (WRONG across XCDs)", k<__HIP_MEMORY_SCOPE_WORKGROUP>, 0);
  run("device scope, table per XCC", k<__HIP_MEMORY_SCOPE_AGENT>, words);
  run("workgroup scope, table per XCC", k<__HIP_MEMORY_SCOPE_WORKGROUP>, words);
  // is the sum right with per-XCC tables and workgroup scope?
  hipMemset(d, 0, (size_t)words * 4 * 8);
  hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(blocks), dim3(tpb), 0, 0, d, words, per, words);
  hipDeviceSynchronize();
  u32* h = (u32*)malloc((size_t)words * 4 * 8); hipMemcpy(h, d, (size_t)words * 4 * 8, hipMemcpyDeviceToHost);
  unsigned long long tot = 0; unsigned long long per_x[8] = {0};
  for (int c = 0; c < 8; c++) for (u32 i = 0; i < words; i++) { tot += h[(size_t)c * words + i]; per_x[c] += h[(size_t)c * words + i]; }
  printf("sum over the 8 tables: %llu (expected %u)  per XCC:", tot, blocks * tpb * per); for (int c = 0; c < 8; c++) printf(" %llu", per_x[c]); printf("\n");
  u32* dx; hipMalloc((void**)&dx, 64 * 4); hipLaunchKernelGGL(k_xcc, dim3(64), dim3(64), 0, 0, dx); u32 hx[64]; hipMemcpy(hx, dx, sizeof(hx), hipMemcpyDeviceToHost);
  printf("XCC id of workgroups 0..23:"); for (int i = 0; i < 24; i++) printf(" %u", hx[i]); printf("\n");
  return 0;
}
