// superk.hip -- reads -> canonical k-mers -> minimizers -> super-k-mers -> 2-bit records per partition.
// (placeholder translation unit: the HIP partitioner kernel is the next row of the scope table;
// until it lands the entry point reports KMX_E_UNSUPPORTED instead of silently using a CPU path.)
#include "kmx_host.hpp"

extern "C" int kmx_superk_partition(kmx_ctx* ctx, const char*, const uint64_t*, uint64_t, uint32_t, uint32_t,
                                    const uint16_t*, uint32_t, uint8_t**, uint64_t*, uint64_t*)
{
  if (!ctx) return KMX_E_INVAL;
  return ctx->fail(KMX_E_UNSUPPORTED, "kmx_superk_partition: HIP partitioner not built yet (no CPU fallback)");
}
