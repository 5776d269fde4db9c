// transpose.hip -- bit-matrix transpose on gfx950.  Replaces km::BitMatrix::transpose / __sse_trans
// (reference include/kmtricks/bitmatrix.hpp:209-214, 238-289): out[c][r] = in[r][c] with bits
// numbered LSB-first inside bytes; nrows, ncols multiples of 8.
//
// One wave per 64 x 64-bit tile: lane l loads the 8 bytes of row r0+l (columns c0..c0+63), then for
// each bit b a __ballot gathers column c0+b over the 64 rows -- exactly the 64 bits of output row
// c0+b -- and lane b keeps it.  Loads and stores are 8 bytes per lane; edge tiles fall back to
// byte accesses.  HBM-bound: read + write W'*N'/8 bytes each.
#include "kmx_host.hpp"

namespace kmx {

__global__ __launch_bounds__(256)
void k_bit_transpose(const u8* __restrict__ in, u8* __restrict__ out, u64 nrows, u64 ncols)
{
  const u64 in_stride = ncols >> 3, out_stride = nrows >> 3;
  const u64 tiles_c = (ncols + 63) >> 6, tiles_r = (nrows + 63) >> 6;
  const int lane = threadIdx.x & 63;
  const u64 wave = (u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (wave >= tiles_c * tiles_r) return;
  const u64 tr = wave / tiles_c, tc = wave % tiles_c;
  const u64 r0 = tr << 6, c0 = tc << 6;
  // load: row r0 + lane, bytes c0/8 .. c0/8+7
  u64 x = 0;
  const u64 r = r0 + lane;
  const u64 cb = c0 >> 3;
  const u64 nbytes_c = min((u64)8, in_stride - cb);
  if (r < nrows) {
    const u8* p = in + r * in_stride + cb;
    if (nbytes_c == 8 && ((reinterpret_cast<uintptr_t>(p) & 7u) == 0)) x = *reinterpret_cast<const u64*>(p);
    else for (u64 b = 0; b < nbytes_c; b++) x |= (u64)p[b] << (8 * b);
  }
  u64 mine = 0;
#pragma unroll
  for (int b = 0; b < 64; b++) {
    const u64 m = __ballot((x >> b) & 1ULL);
    if (lane == b) mine = m;
  }
  // store: output row c0 + lane, bytes r0/8 .. r0/8+7
  const u64 c = c0 + lane;
  if (c < ncols) {
    const u64 rb = r0 >> 3;
    const u64 nbytes_r = min((u64)8, out_stride - rb);
    u8* q = out + c * out_stride + rb;
    if (nbytes_r == 8 && ((reinterpret_cast<uintptr_t>(q) & 7u) == 0)) *reinterpret_cast<u64*>(q) = mine;
    else for (u64 b = 0; b < nbytes_r; b++) q[b] = (u8)(mine >> (8 * b));
  }
}

}  // namespace kmx

using namespace kmx;

extern "C" int kmx_transpose_bits(kmx_ctx* ctx, const uint8_t* in, uint64_t nrows, uint64_t ncols, uint8_t* out)
{
  if (!ctx) return KMX_E_INVAL;
  if (!in || !out) return ctx->fail(KMX_E_INVAL, "transpose: null argument");
  if ((nrows & 7) || (ncols & 7)) return ctx->fail(KMX_E_INVAL, "transpose: nrows and ncols must be multiples of 8 (bitmatrix.hpp:220-226)");
  const u64 bytes = nrows * (ncols >> 3);
  if (bytes == 0) return KMX_OK;
  KMX_HIP(ctx, hipSetDevice(ctx->device));
  u8* d_in = (u8*)ctx->dalloc(bytes);
  u8* d_out = (u8*)ctx->dalloc(bytes);
  if (!d_in || !d_out) { ctx->dfree(d_in); ctx->dfree(d_out); return ctx->fail(KMX_E_NOMEM, "transpose: device allocation failed"); }
  hipError_t e = hipMemcpyAsync(d_in, in, bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    const u64 tiles = ((nrows + 63) >> 6) * ((ncols + 63) >> 6);
    hipLaunchKernelGGL(k_bit_transpose, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, ctx->stream, d_in, d_out, nrows, ncols);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  ctx->dfree(d_in); ctx->dfree(d_out);
  if (e != hipSuccess) return ctx->fail(KMX_E_HIP, std::string("transpose: ") + hipGetErrorString(e));
  return KMX_OK;
}
