// transpose.hip -- bit-matrix transpose on gfx950.  Replaces km::BitMatrix::transpose / __sse_trans
// (reference include/kmtricks/bitmatrix.hpp:209-214, 238-289): out[c][r] = in[r][c] with bits
// numbered LSB-first inside bytes; nrows, ncols multiples of 8.  Also the second half of
// HashMerger::write_as_bft (merge.hpp:631-644): the hash-major Bloom image of a partition becomes the
// sample-major one whose row s is sample s's slice of its final filter.
//
// A workgroup owns a tile of TR_ROWS input rows x TR_CB input bytes (512 x 512 bits):
//   * in:  every input row piece (64 B at row pitch ncols/8, any alignment: a Bloom row of N samples is
//          ceil(N/8) bytes, 313 for N = 2500) is read as aligned dwords and re-aligned with v_alignbyte into LDS;
//   * 64 x 64-bit sub-tiles are transposed by one wave each: lane l holds row l (8 bytes), 64 ballots gather
//     one column each, lane b keeps column b and stores it to the LDS image of the output tile;
//   * out: each of the 512 output rows leaves as one 64-byte piece (coalesced dword stores).
// The blockIdx -> tile map gives every XCD a contiguous run of tiles (column tiles fastest), so the input
// lines two neighbouring column tiles share are fetched once into that XCD's L2.
// HBM-bound: reads and writes nrows*ncols/8 bytes once each.
#include "kmx_host.hpp"

namespace kmx {

constexpr int TR_ROWS = 512;                 // input rows per tile (output bytes per row piece: 64)
constexpr int TR_CB = 64;                    // input bytes per row piece (512 columns)
constexpr int TR_IS = TR_CB + 8;             // LDS pitch of an input row  (72 B: 18 banks apart)
constexpr int TR_OS = TR_ROWS / 8 + 8;       // LDS pitch of an output row (72 B)
constexpr int TR_TPB = 512;

__global__ __launch_bounds__(TR_TPB, 2)
void k_bit_transpose(const u8* __restrict__ in, u8* __restrict__ out, u64 nrows, u64 ncols, u32 n_tiles, u32 tiles_c)
{
  __shared__ __attribute__((aligned(16))) u8 lin[TR_ROWS * TR_IS];
  __shared__ __attribute__((aligned(16))) u8 lout[TR_CB * 8 * TR_OS];
  const u64 in_stride = ncols >> 3, out_stride = nrows >> 3;
  // XCD-aware: workgroup b runs on XCD b % 8; give each XCD a contiguous run of tiles
  const u32 per_xcd = (n_tiles + 7) / 8;
  const u32 tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (tile >= n_tiles) return;
  const u64 tr = tile / tiles_c, tc = tile % tiles_c;
  const u64 r0 = tr * TR_ROWS, cb0 = tc * TR_CB;
  const u32 rows = (u32)min((u64)TR_ROWS, nrows - r0);
  const u32 cbytes = (u32)min((u64)TR_CB, in_stride - cb0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- in: 16 lanes per row, one aligned dword pair each ----
  {
    const uintptr_t last = (reinterpret_cast<uintptr_t>(in) + nrows * in_stride + 3) & ~(uintptr_t)3;   // end of the last dword holding data
    const u32 j = tid & 15;
    for (u32 r = tid >> 4; r < TR_ROWS; r += TR_TPB / 16) {
      u32 w = 0;
      if (r < rows && j * 4 < cbytes) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(in) + (r0 + r) * in_stride + cb0 + j * 4;
        const uintptr_t a4 = a & ~(uintptr_t)3;
        const u32 sh = (u32)(a & 3u);
        const u32 lo = *reinterpret_cast<const u32*>(a4);
        u32 hi = 0;
        if (sh && a4 + 4 < last) hi = *reinterpret_cast<const u32*>(a4 + 4);
        w = sh ? __builtin_amdgcn_alignbyte(hi, lo, sh) : lo;
        const u32 nb = cbytes - j * 4;                       // bytes of this dword that belong to the piece
        if (nb < 4) w &= (1u << (8 * nb)) - 1u;
      }
      *reinterpret_cast<u32*>(&lin[r * TR_IS + j * 4]) = w;
    }
  }
  __syncthreads();
  // ---- 64 x 64-bit sub-tiles: (TR_ROWS / 64) x (TR_CB / 8) of them, one wave each ----
  for (int t = wave; t < (TR_ROWS / 64) * (TR_CB / 8); t += TR_TPB / 64) {
    const int ri = t / (TR_CB / 8), ci = t % (TR_CB / 8);
    const u64 x = *reinterpret_cast<const u64*>(&lin[(ri * 64 + lane) * TR_IS + ci * 8]);
    u64 mine = 0;
#pragma unroll
    for (int b = 0; b < 64; b++) {
      const u64 m = __ballot((x >> b) & 1ULL);
      if (lane == b) mine = m;
    }
    *reinterpret_cast<u64*>(&lout[(ci * 64 + lane) * TR_OS + ri * 8]) = mine;
  }
  __syncthreads();
  // ---- out: output row c0 + c gets bytes [r0/8, r0/8 + rows/8) ----
  {
    const u32 ocols = cbytes * 8;                              // output rows of this tile
    const u32 obytes = rows >> 3;                              // bytes per output row piece (rows is a multiple of 8)
    u8* const obase = out + (cb0 * 8) * out_stride + (r0 >> 3);
    if (((out_stride & 3u) == 0) && ((reinterpret_cast<uintptr_t>(out) & 3u) == 0)) {
      const u32 j = tid & 15;                                  // 16 dwords per piece
      for (u32 c = tid >> 4; c < ocols; c += TR_TPB / 16) {
        const u32 w = *reinterpret_cast<const u32*>(&lout[c * TR_OS + j * 4]);
        u8* q = obase + (u64)c * out_stride + j * 4;
        if (j * 4 + 4 <= obytes) *reinterpret_cast<u32*>(q) = w;
        else for (u32 b = j * 4; b < obytes; b++) q[b - j * 4] = (u8)(w >> (8 * (b - j * 4)));
      }
    } else {
      for (u32 i = tid; i < ocols * obytes; i += TR_TPB) {
        const u32 c = i / obytes, b = i % obytes;
        obase[(u64)c * out_stride + b] = lout[c * TR_OS + b];
      }
    }
  }
}

hipError_t launch_bit_transpose(const u8* in, u8* out, u64 nrows, u64 ncols, hipStream_t st)
{
  if (nrows == 0 || ncols == 0) return hipSuccess;
  const u64 tiles_c = ((ncols >> 3) + TR_CB - 1) / TR_CB, tiles_r = (nrows + TR_ROWS - 1) / TR_ROWS;
  const u64 n_tiles = tiles_c * tiles_r;
  if (n_tiles > 0x7FFFFFF0ULL) return hipErrorInvalidValue;
  const u32 per_xcd = (u32)((n_tiles + 7) / 8);
  hipLaunchKernelGGL(k_bit_transpose, dim3(per_xcd * 8), dim3(TR_TPB), 0, st, in, out, nrows, ncols, (u32)n_tiles, (u32)tiles_c);
  return hipGetLastError();
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
  if (e == hipSuccess) e = launch_bit_transpose(d_in, d_out, nrows, ncols, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  ctx->dfree(d_in); ctx->dfree(d_out);
  if (e != hipSuccess) return ctx->fail(KMX_E_HIP, std::string("transpose: ") + hipGetErrorString(e));
  return KMX_OK;
}
