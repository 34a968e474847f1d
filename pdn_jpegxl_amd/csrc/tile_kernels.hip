// LDS-tiled and register-streaming pixel stages (gfx950).
//
//   recon_tile_kernel     one workgroup per 64x64 tile, all three channels: the sparse entry lists hf_decode_kernel wrote (one 32-bit
//                         entry per non-zero coefficient) are dequantised and scattered into three zeroed LDS tiles, chroma from
//                         luma added from the same luma entries, LF -> LLF, both IDCT passes in place (16x16 sub-blocks of varblocks
//                         >= 16 points on v_mfma_f32_16x16x4_f32), f32 XYB out (12 B/px).  Tiles that hold part of a varblock larger
//                         than the tile, a special 8x8 transform or a progressive frame go to a per-image list for the generic
//                         (unfused, any-size) kernels of kernels.hip.
//   filter_stream_kernel, filter_stream2_kernel
//                         Gaborish + EPF iteration 1 (+ colour and output, or f32 rows for) iteration 2 + colour and output: register
//                         streaming with DPP wave shifts, four pixels per lane, any width and output layout.
//   filter_stream_pairs_kernel, filter_stream2_pairs_kernel
//                         the same for frames of even width with 8-bit output (RGBA, RGB, gray + alpha, gray): two pixels per lane,
//                         buffer addressing, no load in a branch, four / eight wavefronts per SIMD.
//   filter_tile_kernel    one Gaborish / EPF stage on a 64x32 tile + halo staged in LDS (mirrored at the frame edge); the LAST
//                         enabled stage of an image converts XYB -> output samples and merges alpha.  Used for frames without EPF
//                         (Gaborish alone), for Gaborish and iteration 0 of three-iteration frames, and for the stage taps.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "dev_types.h"
#include "dev_util.h"
#include "kernels.h"

namespace jxlhip {

// Wavefront issue priority of the pixel kernels, for tools/ab_prio.sh.  Measured in the pipelined step (round 3): 1 or 3 make
// the streaming filters slower (filters + output 51 -> 82 ms per batch, step 123 -> 148 ms), so the default emits no instruction.
#ifndef JXLHIP_PRIO_PIXEL
#define JXLHIP_PRIO_PIXEL 0
#endif
#if JXLHIP_PRIO_PIXEL
#define JXL_PIXEL_PRIO() __builtin_amdgcn_s_setprio(JXLHIP_PRIO_PIXEL)
#else
#define JXL_PIXEL_PRIO() ((void)0)
#endif

namespace {

constexpr int kTS = 64;       // tile side
constexpr int kLP = 65;       // LDS pitch (conflict-free column walks)

__device__ const uint8_t t_quant_table[kNumStrategies] = {0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16};

__device__ __forceinline__ bool Special(uint32_t s) { return (s >= 1 && s <= 3) || (s >= 12 && s <= 17); }
typedef float __attribute__((ext_vector_type(4))) F4;
// a 16x16 sub-block (2x2 cells at even cell coordinates) goes to the matrix cores, per pass, when the varblocks of its cells are at
// least 16 points long in the pass's direction, of one length and at one offset (MfmaRows / MfmaCols below)
// One 16x16 output sub-block: acc = sum_k A[k] * B[k] with both operand streams fetched up front (N / 4 independent loads
// each, one wait) so that the MFMA chain is not paced by a memory round trip per step.  pa / pb: lane's first operand;
// sa / sb: stride between consecutive k-steps of four.
template <int N>
__device__ __forceinline__ F4 MfmaChain(const float* pa, int sa, const float* pb, int sb) {
  float a[N / 4], b[N / 4];
#pragma unroll
  for (int k = 0; k < N / 4; k++) { a[k] = pa[k * sa]; b[k] = pb[k * sb]; }
  F4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < N / 4; k++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[k], acc, 0, 0, 0);
  return acc;
}
// 32- and 64-point passes: one operand stream is the k-contiguous basis table (basis_mfma: for a lane (n, q) the N / 4 values
// basis[(4 k + q) * N + n], k = 0 .. N/4-1, lie side by side), read from an LDS copy (arrays of 16-byte pieces, so that consecutive
// (n, q) lanes read consecutive 16 bytes): from global memory every 16x16 sub-block paid two to four dependent L2 round trips before
// its first MFMA.
template <bool kTableIsA>
__device__ __forceinline__ F4 MfmaChain32L(const float4* t_lo, const float4* t_hi, int pair, const float* po, int so) {
  const float4 v0 = t_lo[pair], v1 = t_hi[pair];
  const float t[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
  float o[8];
#pragma unroll
  for (int k = 0; k < 8; k++) o[k] = po[k * so];
  F4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 8; k++)
    acc = kTableIsA ? __builtin_amdgcn_mfma_f32_16x16x4f32(t[k], o[k], acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x4f32(o[k], t[k], acc, 0, 0, 0);
  return acc;
}
// The 64-point table likewise (four arrays of 16-byte quarters; staged only by tiles that hold a 64-point varblock): from global
// memory a sub-block paid four dependent L2 round trips before its sixteen MFMAs, and a fifth of this content's area is 64 points long.
template <bool kTableIsA>
__device__ __forceinline__ F4 MfmaChain64L(const float4* t64, int pair, const float* po, int so) {
  float t[16], o[16];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const float4 v = t64[j * 256 + pair];
    t[4 * j] = v.x; t[4 * j + 1] = v.y; t[4 * j + 2] = v.z; t[4 * j + 3] = v.w;
  }
#pragma unroll
  for (int k = 0; k < 16; k++) o[k] = po[k * so];
  F4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 16; k++)
    acc = kTableIsA ? __builtin_amdgcn_mfma_f32_16x16x4f32(t[k], o[k], acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x4f32(o[k], t[k], acc, 0, 0, 0);
  return acc;
}
// Per pass, a 16x16 sub-block only needs ONE transform length and phase over its 16 columns (vertical pass) or rows (horizontal pass):
// the two cells across the pass direction must belong to varblocks of the same length in that direction, at the same offset - they
// need not be the same varblock (16x8 blocks side by side, a 32x16 beside a 32x32 ...).  With the one-varblock test alone, every
// rectangular block of 8 points in the other direction fell back to the vector-ALU loops (a fifth of this content's cells).
__device__ __forceinline__ bool MfmaRows(uint32_t tl, uint32_t tr) {   // vertical pass: rows >= 16, same length and row offset
  return (tl >> 31) && (tr >> 31) && ((tl >> 21) & 7) >= 1 && ((tl >> 21) & 7) == ((tr >> 21) & 7) && ((tl >> 13) & 31) == ((tr >> 13) & 31);
}
__device__ __forceinline__ bool MfmaCols(uint32_t tl, uint32_t bl) {   // horizontal pass: columns >= 16, same length and column offset
  return (tl >> 31) && (bl >> 31) && ((tl >> 18) & 7) >= 1 && ((tl >> 18) & 7) == ((bl >> 18) & 7) && ((tl >> 8) & 31) == ((bl >> 8) & 31);
}
__device__ __forceinline__ int Mirror(int v, int n) {
  while (v < 0 || v >= n) v = v < 0 ? -v - 1 : 2 * n - 1 - v;
  return v;
}

// v^(1/2.4) through the hardware log2 / exp2 (v_log_f32, v_exp_f32: about 1 ulp each, i.e. < 1e-3 of an 8-bit step after the
// * 255) instead of the ~50-instruction powf: the colour conversion was 46 % of the fused filter kernel.
__device__ __forceinline__ float SrgbOetfT(float v) {
  // both branches are computed and selected (v_cndmask): as a conditional the compiler emitted an exec-mask branch per sample
  const float lin = 12.92f * v;
  const float cur = 1.055f * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(v) * (1.0f / 2.4f)) - 0.055f;
  return v <= 0.0031308f ? lin : cur;
}
// the same curve scaled to 0 .. 255 (the 8-bit output path)
__device__ __forceinline__ float SrgbOetf255T(float v) {
  const float lin = (12.92f * 255.0f) * v;
  const float cur = (1.055f * 255.0f) * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(v) * (1.0f / 2.4f)) - (0.055f * 255.0f);
  return v <= 0.0031308f ? lin : cur;
}
__device__ __forceinline__ float PowT(float a, float e) { return __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(a) * e); }   // a > 0
// Encoded value from display-linear, sign-symmetric like the reference's library.  kind: 0 linear, 1 sRGB, 2 BT.709, 3 PQ,
// 5 table (4096 entries over sqrt(linear), clamped to [0, 1]: an evaluated ICC tone curve).
__device__ __forceinline__ float EncodeTransferT(int kind, float v, float pq_scale, const float* lut = nullptr) {
  if (kind == 0) return v;
  if (kind == 5) {
    const float t = __builtin_sqrtf(fminf(fmaxf(v, 0.f), 1.f)) * 4095.0f;
    const int i = min((int)t, 4094);
    return lut[i] + (lut[i + 1] - lut[i]) * (t - (float)i);
  }
  const float a = fabsf(v);
  float r;
  if (kind == 1) r = a <= 0.0031308f ? 12.92f * a : 1.055f * PowT(a, 1.0f / 2.4f) - 0.055f;
  else if (kind == 2) r = a < 0.018f ? 4.5f * a : 1.099f * PowT(a, 0.45f) - 0.099f;
  else {
    const float xp = a > 0.f ? PowT(a * pq_scale, 0.1593017578125f) : 0.f;
    r = a > 0.f ? PowT((0.8359375f + 18.8515625f * xp) / (1.0f + 18.6875f * xp), 78.84375f) : 0.f;
  }
  return copysignf(r, v);
}
__device__ __forceinline__ uint8_t ToU8T(float v) {   // round half up, clamped; NaN -> 0 (v_max_f32 returns the other operand)
  v = fminf(fmaxf(v * 255.0f, 0.f), 255.0f);
  return (uint8_t)(v + 0.5f);
}

// The alpha plane holds samples of the output type (alpha_finish_kernel scales them): one u8 or u16 load.
__device__ __forceinline__ uint32_t LoadAlpha(const DevImage& im, size_t i) {
  return im.out_bits == 8 ? (uint32_t)im.alpha[i] : (im.out_bits == 16 ? (uint32_t)((const uint16_t*)im.alpha)[i] : ((const uint32_t*)im.alpha)[i]);
}
// `a`: the pixel's alpha sample (ignored by layouts without alpha), fetched by the caller ahead of the arithmetic
__device__ __forceinline__ void WritePixelA(const DevImage& im, int x, int y, float X, float Y, float B, uint32_t a) {
  const float gr = Y + X - im.opsin_bias_cbrt[0], gg = Y - X - im.opsin_bias_cbrt[1], gb = B - im.opsin_bias_cbrt[2];
  const float mr = gr * gr * gr + im.opsin_bias[0], mg = gg * gg * gg + im.opsin_bias[1], mb = gb * gb * gb + im.opsin_bias[2];
  float r = im.opsin_inv[0] * mr + im.opsin_inv[1] * mg + im.opsin_inv[2] * mb;
  float g = im.opsin_inv[3] * mr + im.opsin_inv[4] * mg + im.opsin_inv[5] * mb;
  float bl = im.opsin_inv[6] * mr + im.opsin_inv[7] * mg + im.opsin_inv[8] * mb;
  if (im.to_srgb == 1 && !im.out_float) {
    // the common case on its own short path (integer outputs clamp negatives to 0 anyway, so the curve need not be sign-symmetric here)
    r = SrgbOetfT(r); g = SrgbOetfT(g); bl = SrgbOetfT(bl);
  } else if (im.to_srgb) {
    r = EncodeTransferT(im.to_srgb, r, im.pq_scale, im.trc_lut); g = EncodeTransferT(im.to_srgb, g, im.pq_scale, im.trc_lut + 4096);
    bl = EncodeTransferT(im.to_srgb, bl, im.pq_scale, im.trc_lut + 8192);
  }
  if (im.unpremultiply) {   // associated alpha: the encoded samples are divided by max(alpha, 2^-26) (Decoder/JxlDecoder.cpp:233)
    const float m = 1.0f / fmaxf(1.0f / 67108864.0f, (float)a * im.alpha_unit);
    r *= m; g *= m; bl *= m;
  }
  const size_t o = (size_t)(y - im.band_y0) * im.w + x;        // position in the output band
  if (im.out_bits != 8) {   // u16 above 8 bits per sample, f16 / f32 for float samples (Decoder/JxlDecoder.cpp:510-548); `a` is raw bits
    const size_t b = o * im.nch_out;
    if (im.ncolor == 3) {
      StoreOutSample(im.out, b, FloatToOutBits(r, im.out_bits, im.out_float), im.out_bits);
      StoreOutSample(im.out, b + 1, FloatToOutBits(g, im.out_bits, im.out_float), im.out_bits);
      StoreOutSample(im.out, b + 2, FloatToOutBits(bl, im.out_bits, im.out_float), im.out_bits);
      if (im.has_alpha) StoreOutSample(im.out, b + 3, a, im.out_bits);
    } else {
      StoreOutSample(im.out, b, FloatToOutBits(g, im.out_bits, im.out_float), im.out_bits);
      if (im.has_alpha) StoreOutSample(im.out, b + 1, a, im.out_bits);
    }
    return;
  }
  if (im.nch_out == 4) {
    uchar4 px;
    px.x = ToU8T(r); px.y = ToU8T(g); px.z = ToU8T(bl); px.w = (uint8_t)a;
    ((uchar4*)im.out)[o] = px;
  } else {
    uint8_t* out = im.out + o * im.nch_out;
    if (im.ncolor == 3) {
      out[0] = ToU8T(r); out[1] = ToU8T(g); out[2] = ToU8T(bl);
    } else {
      out[0] = ToU8T(g);
      if (im.has_alpha) out[1] = (uint8_t)a;
    }
  }
}
// The common layouts (u8 samples, sRGB or linear transfer) without the branches of the general function: used by the fused filter
// kernel, whose output phase is a large part of its time.
__device__ __forceinline__ bool PlainOutput(const DevImage& im) { return im.out_bits == 8 && im.to_srgb <= 1 && !im.unpremultiply; }
// Out of line on purpose: inlined four times into the fused filter kernel's unrolled output phase, the general function more than
// doubled that kernel's code (2.3 k -> 5.8 k instructions) and cost 8 % of its speed on the plain path that never executes it.
__device__ __noinline__ void WritePixelGeneral(const DevImage& im, int x, int y, float X, float Y, float B, uint32_t a) {
  WritePixelA(im, x, y, X, Y, B, a);
}
__device__ __forceinline__ void WritePixel(const DevImage& im, int x, int y, float X, float Y, float B) {
  WritePixelA(im, x, y, X, Y, B, im.has_alpha ? LoadAlpha(im, (size_t)y * im.w + x) : 0u);
}

}  // namespace

// ------------------------------------------------------------------ fused reconstruction of one 64x64 tile
// one basis row (8 outputs) times two inputs: every basis fetch feeds 16 FMAs
#define JXL_IDCT_STEP(Bp, v0, v1)                                                                  \
  {                                                                                                \
    const float4 b0 = *(const float4*)(Bp);                                                        \
    const float4 b1 = *(const float4*)((Bp) + 4);                                                  \
    a0[0] += v0 * b0.x; a0[1] += v0 * b0.y; a0[2] += v0 * b0.z; a0[3] += v0 * b0.w;                \
    a0[4] += v0 * b1.x; a0[5] += v0 * b1.y; a0[6] += v0 * b1.z; a0[7] += v0 * b1.w;                \
    a1[0] += v1 * b0.x; a1[1] += v1 * b0.y; a1[2] += v1 * b0.z; a1[3] += v1 * b0.w;                \
    a1[4] += v1 * b1.x; a1[5] += v1 * b1.y; a1[6] += v1 * b1.z; a1[7] += v1 * b1.w;                \
  }

// One workgroup per (64x64 tile, channel): dequantisation (+ chroma from luma), LF -> LLF, both IDCT passes in LDS.  Tiles that hold
// part of a varblock larger than the tile, or a special 8x8 transform, are appended to a per-image list for the generic kernels.
// The phases are organised around what the profile of the first version showed (two thirds of the wave time parked at its ten
// barriers / on memory; that version took 118 ms per batch of 384 frames, this one 55):
//   * one 64x65 tile: both IDCT passes run IN PLACE.  The vertical pass gives wavefront w the column stripe [16w, 16w+16)
//     (everything a stripe's outputs depend on lies in the same columns), the horizontal pass gives it the row band
//     [16w, 16w+16); a wavefront fetches all its operands before it stores, so no workgroup barrier is needed inside a
//     pass, and the copy-out of a row band needs only that wavefront.  LDS 19 KB -> 6 workgroups per CU (VGPR-bound).
//   * LLF is computed by wavefront 0 with lane shuffles (no LDS staging, no barriers) while the others dequantise; the
//     dequantisation skips the LLF positions instead of being overwritten after a barrier.
//   * dequantisation: one thread = four consecutive coefficients of a row (one 16-byte load per plane and per weight
//     table), per-cell constants (table pointer with the block offset folded in, scale, index shift) prepared once.
//   * 1 / v through v_rcp_f32 (1 ulp; the term is a bias correction <= 0.15 / |v|).
__device__ __forceinline__ float DequantBias(int32_t v, float qb, float qb3) {
  if (v == 0) return 0.f;
  if (v == 1) return qb;
  if (v == -1) return -qb;
  const float f = (float)v;
  return f - qb3 * __builtin_amdgcn_rcpf(f);
}
typedef int __attribute__((ext_vector_type(4))) I4v;

// The scatter: every entry of the tile's (block, channel) lists is one non-zero quantised coefficient; its scan position k selects
// {stored index, weight} from the per-table scan list, the stored index gives the coefficient's place in the varblock.  The lists of
// the three channels are walked as ONE range [0, T_y + T_x + T_b) by all 768 threads (a 4K frame has about 270 luma and 30 + 30
// chroma entries per tile: one trip).  An entry's cell comes from a binary search over the exclusive prefix of its channel's
// per-cell entry counts (only origin cells have entries).  A luma entry is also what chroma from luma adds to X and B - the IDCT is
// linear and the tile is inside one 64x64 chroma-from-luma tile - so it is added three times (dequantised once) instead of the X
// and B workgroups of the previous version walking and dequantising the luma list again.  Adds into the zeroed tiles with
// ds_add_f32: a position receives at most one own addend and one luma addend, two addends commute, so the sums do not depend on
// the order the lanes arrive in.
constexpr int kTileF = kTS * kLP;   // floats of one channel's tile

// One workgroup per 64x64 tile, all three channels (12 wavefronts: team t = wave / 4 owns channel Y, X, B; inside a team the four
// wavefronts split the tile as the one-channel version did): dequantisation with chroma from luma, LF -> LLF, both IDCT passes in
// LDS.  Tiles that hold part of a varblock larger than the tile, or a special 8x8 transform, are appended to a per-image list for
// the generic kernels.
//   * three 64x65 tiles: both IDCT passes run IN PLACE.  The vertical pass gives a wavefront the column stripe [16w, 16w+16) of
//     its channel (everything a stripe's outputs depend on lies in the same columns), the horizontal pass the row band
//     [16w, 16w+16); a wavefront fetches all its operands before it stores, so no workgroup barrier is needed inside a pass, and
//     the copy-out of a row band needs only that wavefront.
//   * the per-cell set-up (cell info, scale, table pointers, entry-list prefixes of the three channels) is done once per tile by
//     wavefront 0; LLF of channel t is computed by the first wavefront of team t with lane shuffles while the others scatter.
//   * 1 / v through v_rcp_f32 (1 ulp; the term is a bias correction <= 0.15 / |v|).
__global__ __launch_bounds__(768, 6) void recon_tile_kernel(const DevImage* __restrict__ imgs, const float* basis_all, const float* basis_small,
                                                             const float* llf_scale, const float* basis_mfma) {
  JXL_PIXEL_PRIO();
  extern __shared__ __align__(16) uint8_t smem_raw[];
  float* cfc3 = (float*)smem_raw;                      // 3 * kTileF   coefficients -> columns done -> pixels; team order Y, X, B
  float* B816 = cfc3 + 3 * kTileF;                     // 320  IDCT bases of the two common sizes (N = 8 at 0, N = 16 at 64)
  float* cscale = B816 + 320;                          // 64   per cell: inv_global_scale / raw quant of its varblock
  uint32_t* ci = (uint32_t*)(cscale + 64);             // 64   cell info
  uint32_t* cmeta = ci + 64;                           // 64   log2 of the table pitch | transposed << 4
  uint32_t* cnq = cmeta + 64;                          // 64   entries per channel of the cell's dequant table
  uint32_t* cpre = cnq + 64;                           // 3 x 64   exclusive prefix of the entry counts per team ...
  uint32_t* cstart = cpre + 3 * 64;                    // 3 x 64   ... and where each origin cell's entries start in the group's list
  uint32_t* ctot = cstart + 3 * 64;                    // 4    totals per team
  const U32x2** csc = (const U32x2**)(ctot + 4);       // 64   scan list of the cell's quant table
  float4* t32_lo = (float4*)(csc + 64);                // 128  k-contiguous 32-point basis, first / second four k-steps of lane (n, q) = [n * 4 + q]
  float4* t32_hi = t32_lo + 128;                       // 128
  float4* t64 = t32_hi + 128;                          // 4 x 256  the 64-point basis, quarter j of lane (n, q) = [j * 256 + n * 4 + q]
  const DevImage& im = imgs[blockIdx.y];
  const int tile = blockIdx.x;
  if (tile >= im.wt * im.ht) return;
  const int tid = threadIdx.x;
  const int tx = tile % im.wt, ty = tile / im.wt;
  if (ty < im.dec_gy0 * 4 || ty >= im.dec_gy1 * 4) return;   // outside the decoded band (4 tile rows per group row)
  const int wave = tid >> 6, lane = tid & 63, team = wave >> 2, tw = wave & 3;
  const int c = team == 0 ? 1 : (team == 1 ? 0 : 2);          // channel of this wavefront's team
  const int wp = im.wp, hp = im.hp;
  // the tiles start as zeros: only non-zero coefficients exist in the entry lists
  {
    float4* z = (float4*)cfc3;
    for (int i = tid; i < 3 * kTileF / 4; i += 768) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  int bad = 0, has64 = 0;
  if (tid < 64) {
    const int cx = tx * 8 + (tid & 7), cy = ty * 8 + (tid >> 3);
    const bool inside = cx < im.w8 && cy < im.h8;
    const size_t cell_g = (size_t)min(cy, im.h8 - 1) * im.w8 + min(cx, im.w8 - 1);
    const size_t ncells = (size_t)im.w8 * im.h8;
    const uint32_t info_g = im.cellinfo[cell_g];
    const uint32_t rq_g = im.rawq[cell_g];
    U32x2 blk[3];
    blk[0] = im.cblk[ncells + cell_g]; blk[1] = im.cblk[cell_g]; blk[2] = im.cblk[2 * ncells + cell_g];   // team order: Y, X, B
    const uint32_t info = inside ? info_g : 0u;
    const uint32_t rqv = inside ? rq_g : 1u;
    const int ix = (info >> 8) & 31, iy = (info >> 13) & 31, lcx = (info >> 18) & 7, lcy = (info >> 21) & 7;
    if (inside) {
      if (!(info >> 31)) bad = 1;
      else {
        const int ox = (tid & 7) - ix, oy = (tid >> 3) - iy;
        // blocks larger than the tile, the rare special 8x8 transforms, and blocks that do not start on a multiple of their own size
        // (every encoder aligns them, the format's placement rule does not; the matrix-core sub-blocks below rely on it) are left
        // to the generic kernels
        bad = ox < 0 || oy < 0 || ox + (1 << lcx) > 8 || oy + (1 << lcy) > 8 || Special(info & 0xFF) ||
              (ox & ((1 << lcx) - 1)) != 0 || (oy & ((1 << lcy) - 1)) != 0;
      }
    }
    const bool valid = (info >> 31) && !bad;
    has64 = valid && (lcx == 3 || lcy == 3);
    {  // (wavefront 0 is the only writer of the per-cell tables; the flag travels with them through the barrier below)
      const uint64_t any = __ballot(has64);
      if (tid == 0) ctot[3] = any != 0;
    }
    const uint32_t q = t_quant_table[valid ? (info & 0xFF) : 0];
    const uint32_t rq_origin = (uint32_t)__shfl((int)rqv, valid ? tid - iy * 8 - ix : tid);
    const uint32_t lng = 3 + max(lcx, lcy);
    const bool transposed = lcy >= lcx;   // the stored layout of a varblock has its longer side horizontal (squares: transposed too)
    ci[tid] = info;
    cscale[tid] = im.inv_global_scale / (float)rq_origin;
    cmeta[tid] = lng | (transposed ? 16u : 0u);
    cnq[tid] = im.dq_n[q];
    csc[tid] = im.scan[q];
    // entry lists of the varblocks that start in this tile; a list a failed section left unwritten is treated as empty
    const bool origin = valid && ix == 0 && iy == 0;
    uint32_t cnt[3], sum[3];
#pragma unroll
    for (int t = 0; t < 3; t++) {
      cnt[t] = origin ? blk[t].y : 0u;
      if (cnt[t] > 65536u || blk[t].x > kGroupEntriesCap - cnt[t]) cnt[t] = 0;
      sum[t] = cnt[t];   // inclusive scans over the wavefront
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const uint32_t a = (uint32_t)__shfl_up((int)sum[t], d);
        if (tid >= d) sum[t] += a;
      }
    }
#pragma unroll
    for (int t = 0; t < 3; t++) {
      cpre[t * 64 + tid] = sum[t] - cnt[t];
      cstart[t * 64 + tid] = blk[t].x;
      if (tid == 63) ctot[t] = sum[t];
    }
  } else if (tid < 64 + 320) {
    B816[tid - 64] = basis_all[tid - 64];   // basis_all holds N = 8 at offset 0 and N = 16 at offset 64
  } else if (tid < 64 + 320 + 256) {
    const int i = tid - (64 + 320);         // 16-byte piece i of the 32-point table: run (n, q) = i >> 1, half i & 1
    const float4 v = ((const float4*)basis_mfma)[i];
    if (i & 1) t32_hi[i >> 1] = v; else t32_lo[i >> 1] = v;
  }
  if (im.num_passes > 1) bad = 1;   // progressive frames: the passes' entries are summed as integers first (expand kernels, generic path)
  if (__syncthreads_or(bad)) {
    if (tid == 0) im.tile_list[atomicAdd(&im.status[1], 1u)] = (uint32_t)tile;
    return;
  }
  if (ctot[3]) {   // the tile holds a 64-point varblock: stage that table (16 KB; the barrier before the passes covers it)
    for (int i = tid; i < 1024; i += 768) t64[(i & 3) * 256 + (i >> 2)] = ((const float4*)(basis_mfma + 1024))[i];
  }
  const float* const Bl = basis_all;
  float* const cfc = cfc3 + team * kTileF;   // this team's tile
  // ---- LLF (first wavefront of each team): lowest cx*cy coefficients of every varblock = scaled 2-D DCT of its LF samples, by lane shuffles
  if (tw == 0) {
    const int cxg = tx * 8 + (lane & 7), cyg = ty * 8 + (lane >> 3);
    const bool inside = cxg < im.w8 && cyg < im.h8;
    const float lf_g = im.lf_final[c][(size_t)min(cyg, im.h8 - 1) * im.w8 + min(cxg, im.w8 - 1)];
    const float lfv = inside ? lf_g : 0.f;
    const uint32_t info = ci[lane];
    const bool valid = info >> 31;
    const int ix = (info >> 8) & 31, iy = (info >> 13) & 31, lcx = (info >> 18) & 7, lcy = (info >> 21) & 7;
    const int cx = 1 << lcx, cy = 1 << lcy;
    const int ox = (lane & 7) - ix, oy = (lane >> 3) - iy;
    const float* Bx = basis_small + (cx * cx - 1) / 3 + ix * cx;
    const float* By = basis_small + (cy * cy - 1) / 3 + iy * cy;
    float bx[8], byv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { bx[k] = (valid && k < cx) ? Bx[k] : 0.f; byv[k] = (valid && k < cy) ? By[k] : 0.f; }
    const float sc = valid ? llf_scale[lcy * 32 + iy] * llf_scale[lcx * 32 + ix] / (float)(cx * cy) : 0.f;
    float row = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const float v = __shfl(lfv, (lane - ix + k) & 63);
      if (valid && k < cx) row += v * bx[k];
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const float v = __shfl(row, ((oy + k) * 8 + (lane & 7)) & 63);
      if (valid && k < cy) acc += v * byv[k];
    }
    // (an add like the entries': a custom coefficient order may send an entry to an LLF position, and sums do not depend on who comes first)
    if (valid) atomicAdd(&cfc[(oy * 8 + iy) * kLP + ox * 8 + ix], acc * sc);
  }
  // ---- dequantisation (+ chroma from luma) of the non-zero coefficients, scattered to their places
  {
    const size_t tile_cfl = (size_t)ty * im.wt + tx;
    const float cfl_x = im.base_x + (float)im.ytox[tile_cfl] * im.inv_color_factor;
    const float cfl_b = im.base_b + (float)im.ytob[tile_cfl] * im.inv_color_factor;
    // the tile's group (four tiles per group side)
    const uint32_t* entries = im.centries + (size_t)((ty >> 2) * im.xg + (tx >> 2) - im.centries_g0) * kGroupEntriesCap;
    const uint32_t t0 = ctot[0], t01 = t0 + ctot[1], total = t01 + ctot[2];
    const float qb3 = im.qbias[3];
    for (uint32_t e = (uint32_t)tid; e < total; e += 768) {
      const uint32_t t = (e >= t0) + (e >= t01);                     // team of this entry: 0 Y, 1 X, 2 B
      const uint32_t el = e - (t == 0 ? 0u : (t == 1 ? t0 : t01));
      const uint32_t* pre = cpre + t * 64;
      uint32_t j = 0;
#pragma unroll
      for (int step = 32; step; step >>= 1) if (pre[j + step] <= el) j += step;
      const uint32_t ent = entries[cstart[t * 64 + j] + (el - pre[j])];
      const uint32_t k = ent & 0xFFFFu, nq = cnq[j];
      if (k >= nq) continue;   // a list that a failed section left half-written must not index past the table
      const int chan = t == 0 ? 1 : (t == 1 ? 0 : 2);
      const U32x2 se = csc[j][(size_t)chan * nq + k];
      const uint32_t meta = cmeta[j];
      const uint32_t lng = meta & 15, p = se.x;
      const uint32_t r = p >> lng, cc = p & ((1u << lng) - 1);
      const uint32_t ky = (meta & 16) ? cc : r, kx = (meta & 16) ? r : cc;
      if (ky >= 64 || kx >= 64) continue;
      const int32_t v = (int32_t)ent >> 16;
      const float dm = t == 0 ? 1.0f : (t == 1 ? im.x_dm : im.b_dm);
      const float o = DequantBias(v, im.qbias[chan], qb3) * (cscale[j] * dm) * __uint_as_float(se.y);
      const uint32_t at = ((j >> 3) * 8 + ky) * kLP + (j & 7) * 8 + kx;
      atomicAdd(&cfc3[t * kTileF + at], o);
      if (t == 0) {   // chroma from luma: X += cfl_x * Y, B += cfl_b * Y on the dequantised luma coefficient
        if (cfl_x != 0.f) atomicAdd(&cfc3[kTileF + at], o * cfl_x);
        if (cfl_b != 0.f) atomicAdd(&cfc3[2 * kTileF + at], o * cfl_b);
      }
    }
  }
  __syncthreads();
  const int l16 = tid & 15, lq = lane >> 4;
  // ---- vertical pass, in place: wavefront = column stripe.  Matrix cores for every 16x16 sub-block that lies inside a varblock
  // of at least 16 points both ways: Basis_R^T (16 x R) * coefficients (R x 16) by v_mfma_f32_16x16x4_f32 (exact f32, the same
  // k-ordered fma chain as the VALU path).  Lane l feeds A[l & 15][l >> 4] and B[l >> 4][l & 15], owns D[4 * (l >> 4) + r][l & 15].
  {
    F4 acc[4];
    bool m[4];
    const int x0 = tw * 16;
#pragma unroll
    for (int rb = 0; rb < 4; rb++) {
      const uint32_t inf = ci[rb * 16 + tw * 2];
      m[rb] = MfmaRows(inf, ci[rb * 16 + tw * 2 + 1]);   // wave-uniform
      if (m[rb]) {
        const int iy = (inf >> 13) & 31, lcy = (inf >> 21) & 7, R = 8 << lcy;
        const float* cp = cfc + ((rb * 2 - iy) * 8 + lq) * kLP + x0 + l16;
        if (R == 16) acc[rb] = MfmaChain<16>(B816 + 64 + lq * 16 + iy * 8 + l16, 64, cp, 4 * kLP);
        else if (R == 32) acc[rb] = MfmaChain32L<true>(t32_lo, t32_hi, (iy * 8 + l16) * 4 + lq, cp, 4 * kLP);
        else acc[rb] = MfmaChain64L<true>(t64, (iy * 8 + l16) * 4 + lq, cp, 4 * kLP);
      }
    }
    // the rest on the vector ALUs.  One lane = two adjacent columns x one 8-row cell: every basis fetch feeds 16 FMAs.
    const int x = x0 + (lane & 7) * 2, cr = lane >> 3;
    const uint32_t info = ci[cr * 8 + (x >> 3)];
    const bool valu = (info >> 31) && !MfmaRows(ci[(cr >> 1) * 16 + (x >> 4) * 2], ci[(cr >> 1) * 16 + (x >> 4) * 2 + 1]);
    float a0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (valu) {
      const int iy = (info >> 13) & 31, lcy = (info >> 21) & 7;
      const int R = 8 << lcy;
      const float* in = cfc + (cr - iy) * 8 * kLP + x;
      if (R == 8) {            // the common sizes read their basis from LDS: no global load inside the pass
#pragma unroll 2
        for (int k = 0; k < 8; k++) { const float v0 = in[k * kLP], v1 = in[k * kLP + 1]; JXL_IDCT_STEP(B816 + k * 8, v0, v1) }
      } else if (R == 16) {
        const float* B = B816 + 64 + iy * 8;
#pragma unroll 2
        for (int k = 0; k < 16; k++) { const float v0 = in[k * kLP], v1 = in[k * kLP + 1]; JXL_IDCT_STEP(B + k * 16, v0, v1) }
      } else {
        const float* B = Bl + (R * R - 64) / 3 + iy * 8;
        for (int k = 0; k < R; k++) { const float v0 = in[k * kLP], v1 = in[k * kLP + 1]; JXL_IDCT_STEP(B + k * R, v0, v1) }
      }
    }
    // every operand of this stripe is in registers: store
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
      if (m[rb]) {
        float* o = cfc + (rb * 16 + 4 * lq) * kLP + x0 + l16;
        o[0] = acc[rb].x; o[kLP] = acc[rb].y; o[2 * kLP] = acc[rb].z; o[3 * kLP] = acc[rb].w;
      }
    if (valu) {
#pragma unroll
      for (int j = 0; j < 8; j++) { cfc[(cr * 8 + j) * kLP + x] = a0[j]; cfc[(cr * 8 + j) * kLP + x + 1] = a1[j]; }
    }
  }
  __syncthreads();
  // ---- horizontal pass, in place: wavefront = row band: rows (16 x C) * Basis_C (C x 16)
  {
    F4 acc[4];
    bool m[4];
    const int y0 = tw * 16;
#pragma unroll
    for (int cs = 0; cs < 4; cs++) {
      const uint32_t inf = ci[tw * 16 + cs * 2];
      m[cs] = MfmaCols(inf, ci[tw * 16 + 8 + cs * 2]);
      if (m[cs]) {
        const int ix = (inf >> 8) & 31, lcx = (inf >> 18) & 7, C = 8 << lcx;
        const float* ap = cfc + (y0 + l16) * kLP + (cs * 2 - ix) * 8 + lq;
        if (C == 16) acc[cs] = MfmaChain<16>(ap, 4, B816 + 64 + lq * 16 + ix * 8 + l16, 64);
        else if (C == 32) acc[cs] = MfmaChain32L<false>(t32_lo, t32_hi, (ix * 8 + l16) * 4 + lq, ap, 4);
        else acc[cs] = MfmaChain64L<false>(t64, (ix * 8 + l16) * 4 + lq, ap, 4);
      }
    }
    // the rest: two adjacent rows per lane
    const int y = y0 + (lane & 7) * 2, cc = lane >> 3;
    const uint32_t info = ci[(y >> 3) * 8 + cc];
    const bool valu = (info >> 31) && !MfmaCols(ci[(y >> 4) * 16 + (cc >> 1) * 2], ci[(y >> 4) * 16 + 8 + (cc >> 1) * 2]);
    float a0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (valu) {
      const int ix = (info >> 8) & 31, lcx = (info >> 18) & 7;
      const int C = 8 << lcx;
      const float* in = cfc + y * kLP + (cc - ix) * 8;
      if (C == 8) {
#pragma unroll 2
        for (int k = 0; k < 8; k++) { const float v0 = in[k], v1 = in[k + kLP]; JXL_IDCT_STEP(B816 + k * 8, v0, v1) }
      } else if (C == 16) {
        const float* B = B816 + 64 + ix * 8;
#pragma unroll 2
        for (int k = 0; k < 16; k++) { const float v0 = in[k], v1 = in[k + kLP]; JXL_IDCT_STEP(B + k * 16, v0, v1) }
      } else {
        const float* B = Bl + (C * C - 64) / 3 + ix * 8;
        for (int k = 0; k < C; k++) { const float v0 = in[k], v1 = in[k + kLP]; JXL_IDCT_STEP(B + k * C, v0, v1) }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int cs = 0; cs < 4; cs++)
      if (m[cs]) {
        float* o = cfc + (y0 + 4 * lq) * kLP + cs * 16 + l16;
        o[0] = acc[cs].x; o[kLP] = acc[cs].y; o[2 * kLP] = acc[cs].z; o[3 * kLP] = acc[cs].w;
      }
    if (valu) {
#pragma unroll
      for (int j = 0; j < 8; j++) { cfc[y * kLP + cc * 8 + j] = a0[j]; cfc[(y + 1) * kLP + cc * 8 + j] = a1[j]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- copy-out of this wavefront's row band: 16 bytes per lane
    float* dst = im.xyb[c];
    const int x4 = l16 * 4;
    const int gx = tx * kTS + x4;
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int yl = y0 + lq + 4 * it;
      const int gy = ty * kTS + yl;
      const float* r = cfc + yl * kLP + x4;
      const float4 o = make_float4(r[0], r[1], r[2], r[3]);
      if (gx < wp && gy < hp) *(float4*)(dst + (size_t)gy * wp + gx) = o;
    }
  }
}

// ------------------------------------------------------------------ LDS-tiled loop filters
// kStage: 0 Gaborish, 1 EPF pass 0, 2 EPF pass 1, 3 EPF pass 2.  Tile = 64 x 32 output pixels.
template <int kStage>
__global__ __launch_bounds__(256) void filter_tile_kernel(const DevImage* __restrict__ imgs) {
  JXL_PIXEL_PRIO();
  constexpr int TW = 64, TH = 32;
  constexpr int HALO = kStage == 0 ? 1 : (kStage == 1 ? 3 : (kStage == 2 ? 2 : 1));
  constexpr int LW = TW + 2 * HALO, LH = TH + 2 * HALO;
  __shared__ float t[3][LH][LW + 1];
  const DevImage& im = imgs[blockIdx.y];
  if (!im.stage_on[kStage]) return;
  const int w = im.w, h = im.h, wp = im.wp;
  const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + TH - 1) / TH;
  if ((int)blockIdx.x >= tiles_x * tiles_y) return;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int x0 = tx * TW, y0 = ty * TH;
  // Band decode: the last stage writes exactly the band; earlier stages also produce the rows later stages read (<= 8)
  const bool final_stage = im.final_stage == kStage;
  const int row_lo = final_stage ? im.band_y0 : max(0, im.band_y0 - 8), row_hi = final_stage ? im.band_y1 : min(h, im.band_y1 + 8);
  if (y0 + TH <= row_lo || y0 >= row_hi) return;
  const float* in0 = im.stage_in[kStage][0];
  const float* in1 = im.stage_in[kStage][1];
  const float* in2 = im.stage_in[kStage][2];
  // staged with unconditional (clamped-index) loads in unrolled batches so that a thread's fetches overlap
  constexpr int kLoadIters = (LW * LH + 255) / 256;
#pragma unroll
  for (int it = 0; it < kLoadIters; it++) {
    const int e0 = threadIdx.x + it * 256;
    const int e = e0 < LW * LH ? e0 : LW * LH - 1;
    const int ly = e / LW, lx = e % LW;
    const size_t g = (size_t)Mirror(y0 - HALO + ly, h) * wp + Mirror(x0 - HALO + lx, w);
    const float v0 = in0[g], v1 = in1[g], v2 = in2[g];
    t[0][ly][lx] = v0;
    t[1][ly][lx] = v1;
    t[2][ly][lx] = v2;
  }
  __syncthreads();
#pragma unroll 2
  for (int e = threadIdx.x; e < TW * TH; e += 256) {
    const int ly = e / TW, lx = e % TW;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= w || y < row_lo || y >= row_hi) continue;
    const int cy = ly + HALO, cx = lx + HALO;
    float o0, o1, o2;
    if (kStage == 0) {
      float o[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        o[c] = t[c][cy][cx] * im.gab_w[c][0] +
               (t[c][cy - 1][cx] + t[c][cy + 1][cx] + t[c][cy][cx - 1] + t[c][cy][cx + 1]) * im.gab_w[c][1] +
               (t[c][cy - 1][cx - 1] + t[c][cy - 1][cx + 1] + t[c][cy + 1][cx - 1] + t[c][cy + 1][cx + 1]) * im.gab_w[c][2];
      }
      o0 = o[0]; o1 = o[1]; o2 = o[2];
    } else {
      constexpr int kNoff = kStage == 1 ? 12 : 4;
      constexpr int kNplus = kStage == 3 ? 1 : 5;
      const int off0[12][2] = {{-2, 0}, {-1, -1}, {-1, 0}, {-1, 1}, {0, -2}, {0, -1}, {0, 1}, {0, 2}, {1, -1}, {1, 0}, {1, 1}, {2, 0}};
      const int off1[4][2] = {{-1, 0}, {0, -1}, {0, 1}, {1, 0}};
      const int plus[5][2] = {{0, 0}, {-1, 0}, {1, 0}, {0, -1}, {0, 1}};
      const float is = im.inv_sigma[(size_t)(y >> 3) * im.w8 + (x >> 3)];
      o0 = t[0][cy][cx]; o1 = t[1][cy][cx]; o2 = t[2][cy][cx];
      if (!(is < -3.90524291751269967465540850526868f)) {
        const float sm = kStage == 1 ? im.epf_pass0_sigma_scale : (kStage == 2 ? 1.0f : im.epf_pass2_sigma_scale);
        const bool border = ((x & 7) == 0) || ((x & 7) == 7) || ((y & 7) == 0) || ((y & 7) == 7);
        const float inv = is * (border ? sm * im.epf_border_sad_mul : sm);
        float wsum = 1.0f;
#pragma unroll
        for (int k = 0; k < kNoff; k++) {
          const int dy = kStage == 1 ? off0[k][0] : off1[k][0], dx = kStage == 1 ? off0[k][1] : off1[k][1];
          float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int p = 0; p < kNplus; p++) {
            const int ay = cy + plus[p][0], ax = cx + plus[p][1];
            s0 += fabsf(t[0][ay][ax] - t[0][ay + dy][ax + dx]);
            s1 += fabsf(t[1][ay][ax] - t[1][ay + dy][ax + dx]);
            s2 += fabsf(t[2][ay][ax] - t[2][ay + dy][ax + dx]);
          }
          const float sad = s0 * im.epf_channel_scale[0] + s1 * im.epf_channel_scale[1] + s2 * im.epf_channel_scale[2];
          const float wt = fmaxf(0.0f, 1.0f + sad * inv);
          wsum += wt;
          o0 += wt * t[0][cy + dy][cx + dx];
          o1 += wt * t[1][cy + dy][cx + dx];
          o2 += wt * t[2][cy + dy][cx + dx];
        }
        const float iw = 1.0f / wsum;
        o0 *= iw; o1 *= iw; o2 *= iw;
      }
    }
    if (final_stage) {
      WritePixel(im, x, y, o0, o1, o2);
    } else {
      const size_t g = (size_t)y * wp + x;
      im.stage_out[kStage][0][g] = o0;
      im.stage_out[kStage][1][g] = o1;
      im.stage_out[kStage][2][g] = o2;
    }
  }
}

// Gaborish + EPF pass 1 (the only pass of epf_iters == 1) + XYB -> output samples in one kernel, WITHOUT LDS: a wavefront owns a
// strip of 256 columns (64 quads of 4 pixels; the outer quad each side is halo, 248 columns are output) and walks
// down a segment of rows.  Every lane keeps the rolling windows of its quad in registers - two input rows with their horizontal
// neighbour sums, three Gaborish rows, three rows of vertical and of horizontal channel-weighted differences - and gets its
// horizontal neighbours from the lanes beside it by DPP wave shifts (v_mov_dpp wave_shr / wave_shl: one VALU op, no LDS round trip).
// Per input row a lane issues three 16-byte loads (256 contiguous bytes per group and plane), one row ahead of their use; per
// output row one 16-byte store.  The LDS-tiled version read ~75 LDS words per pixel (37 % of its LDS cycles bank conflicts) and
// re-read a 38 x 40 footprint per 32 x 32 tile (1.48x); this one reads 64 / 56 x 70 / 64 = 1.25x and is bound by its VALU work.
constexpr int kStripLanes = 64;                    // lanes of a strip: a whole wavefront (DPP wave shifts cross the rows of 16)
constexpr int kStripOut = 4 * (kStripLanes - 2);   // output columns of a strip: every lane's quad but the two halo quads at its ends (248;
                                                   // with strips of one DPP row - 16 lanes, 56 of 64 columns - a seventh of the work was halo)
constexpr int kGroupsPerWg = 256 / kStripLanes;
constexpr int kSegRows = 128;    // output rows of a segment (+ 3 halo rows each side)
__device__ __forceinline__ float DppFromLeft(float v) {    // value of lane - 1 inside the row of 16 (lane 0: 0)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /* wave_shr:1 */, 0xF, 0xF, true));
}
__device__ __forceinline__ float DppFromRight(float v) {   // value of lane + 1 inside the row of 16 (lane 15: 0)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xF, 0xF, true));
}
__device__ __forceinline__ F4 ShiftFromLeft(F4 v) { return F4{DppFromLeft(v.w), v.x, v.y, v.z}; }      // element x - 1
__device__ __forceinline__ F4 ShiftFromRight(F4 v) { return F4{v.y, v.z, v.w, DppFromRight(v.x)}; }    // element x + 1
__device__ __forceinline__ F4 ShiftFromLeft2(F4 v) { return F4{DppFromLeft(v.z), DppFromLeft(v.w), v.x, v.y}; }   // element x - 2
__device__ __forceinline__ F4 Abs4(F4 v) { return F4{fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w)}; }

struct Row3 { F4 c[3]; };
#define JXL_GLOBAL __attribute__((address_space(1)))
// Quad (X .. X + 3) of row y of the three planes; rows and columns outside the frame are mirrored like the unfused stages do.
__device__ __forceinline__ Row3 LoadRow3(const JXL_GLOBAL float* p0, const JXL_GLOBAL float* p1, const JXL_GLOBAL float* p2, int X, int y, int w, int h,
                                         int wp, bool interior) {
  const int yy = y < 0 ? -y - 1 : (y >= h ? 2 * h - 1 - y : y);
  Row3 r;
  if (interior) {
    const size_t g = (size_t)yy * wp + X;
    r.c[0] = *(const JXL_GLOBAL F4*)(p0 + g); r.c[1] = *(const JXL_GLOBAL F4*)(p1 + g); r.c[2] = *(const JXL_GLOBAL F4*)(p2 + g);
  } else {
    const size_t g = (size_t)yy * wp;
    const int x0 = Mirror(X, w), x1 = Mirror(X + 1, w), x2 = Mirror(X + 2, w), x3 = Mirror(X + 3, w);
    r.c[0] = F4{p0[g + x0], p0[g + x1], p0[g + x2], p0[g + x3]};
    r.c[1] = F4{p1[g + x0], p1[g + x1], p1[g + x2], p1[g + x3]};
    r.c[2] = F4{p2[g + x0], p2[g + x1], p2[g + x2], p2[g + x3]};
  }
  return r;
}
// XYB -> sRGB-encoded u8 (the plain output layouts)
__device__ __forceinline__ uint32_t PixelToRgba8(const DevImage& im, float X, float Y, float B, uint32_t a) {
  const float gr = Y + X - im.opsin_bias_cbrt[0], gg = Y - X - im.opsin_bias_cbrt[1], gb = B - im.opsin_bias_cbrt[2];
  const float mr = gr * gr * gr + im.opsin_bias[0], mg = gg * gg * gg + im.opsin_bias[1], mb = gb * gb * gb + im.opsin_bias[2];
  float r = im.opsin_inv[0] * mr + im.opsin_inv[1] * mg + im.opsin_inv[2] * mb;
  float g = im.opsin_inv[3] * mr + im.opsin_inv[4] * mg + im.opsin_inv[5] * mb;
  float bl = im.opsin_inv[6] * mr + im.opsin_inv[7] * mg + im.opsin_inv[8] * mb;
  // to 8 bits with v_cvt_pk_u8_f32: converts (round to nearest, ties to even - tools/cvt_probe.hip), saturates to 0 .. 255, maps NaN to
  // 0 and inserts the byte, in ONE instruction per channel instead of scale / max / min / add / convert / shift / or; the * 255 is
  // folded into the transfer curve's constants.  (Differs from round-half-up only on exact ties.)
  if (im.to_srgb) { r = SrgbOetf255T(r); g = SrgbOetf255T(g); bl = SrgbOetf255T(bl); }
  else { r *= 255.0f; g *= 255.0f; bl *= 255.0f; }
  uint32_t px = a << 24;
  px = __builtin_amdgcn_cvt_pk_u8_f32(r, 0, px);
  px = __builtin_amdgcn_cvt_pk_u8_f32(g, 1, px);
  px = __builtin_amdgcn_cvt_pk_u8_f32(bl, 2, px);
  return px;
}

// Rolling windows of one lane, as four slots each, indexed by (row & 3) with COMPILE-TIME phases (the row loop is unrolled by four):
// no register moves to shift a window.  in / s: input rows and their horizontal neighbour sums; g: Gaborish rows; dv / dh: vertical /
// horizontal channel-weighted differences of Gaborish rows; hd: dv[x-1] + dv[x+1].
struct StreamState {
  Row3 in[4], s[4], g[4];
  F4 dv[4], dh[4], hd[4];
};
struct StreamConst {
  const JXL_GLOBAL float *in0, *in1, *in2, *inv_sigma;
  const JXL_GLOBAL uint8_t* alpha;
  float gw0[3], gw1[3], gw2[3], cs[3], bsm;
  int X, w, h, wp, w8, y0, y1, r_end, cellx, xa;
  bool interior, stores, full_quad, plain, xb0, xb3;
  JXL_GLOBAL float *f0, *f1, *f2;   // two EPF iterations: the first one's rows leave as f32 planes (filter_stream2_kernel reads them)
  bool to_float;
};

// One input row r (phase P = (r - first row) & 3): Gaborish row r - 1, differences, and the output row r - 3.
template <int P>
__device__ __forceinline__ void StreamRow(const DevImage& im, const StreamConst& k, StreamState& st, Row3& next, int r) {
  constexpr int s0 = P, s1 = (P + 1) & 3, s2 = (P + 2) & 3, s3 = (P + 3) & 3;   // slots of rows r (= r - 4), r - 3, r - 2, r - 1
  const Row3 cur = next;
  next = LoadRow3(k.in0, k.in1, k.in2, k.X, min(r + 1, k.r_end - 1), k.w, k.h, k.wp, k.interior);
  // operands of the output row y = r - 3, requested a whole row of arithmetic ahead of their use
  const int y = r - 3;
  const int yc = min(max(y, 0), k.h - 1);
  const float is = k.inv_sigma[(size_t)(yc >> 3) * k.w8 + k.cellx];
  uint32_t al = 0xFFFFFFFFu;
  if (im.has_alpha && k.plain) {
    const JXL_GLOBAL uint8_t* ap = k.alpha + (size_t)yc * k.w + k.xa;
    al = (uint32_t)ap[0] | (uint32_t)ap[1] << 8 | (uint32_t)ap[2] << 16 | (uint32_t)ap[3] << 24;
  }
  const F4 zero = {0.f, 0.f, 0.f, 0.f};
  F4 dhn = zero, dvn = zero;
  st.in[s0] = cur;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    st.s[s0].c[c] = ShiftFromLeft(cur.c[c]) + ShiftFromRight(cur.c[c]);
    // Gaborish of row r - 1 from input rows r - 2, r - 1, r
    const F4 gn = st.in[s3].c[c] * k.gw0[c] + ((st.in[s2].c[c] + cur.c[c]) + st.s[s3].c[c]) * k.gw1[c] + (st.s[s2].c[c] + st.s[s0].c[c]) * k.gw2[c];
    dhn += Abs4(gn - ShiftFromRight(gn)) * k.cs[c];      // dh[r-1][x] = sum_c scale_c |gab[x] - gab[x + 1]|
    dvn += Abs4(st.g[s2].c[c] - gn) * k.cs[c];            // dv[r-2][x] = sum_c scale_c |gab[r-2] - gab[r-1]|
    st.g[s3].c[c] = gn;
  }
  // now: g[s0] = gab[r-4], g[s1] = gab[r-3], g[s2] = gab[r-2]; dv[s3] = dv[r-5], dv[s0] = dv[r-4], dv[s1] = dv[r-3]; dh[s0..s2] = dh[r-4..r-2]
  const F4 dv0 = st.dv[s3], dv1 = st.dv[s0], dv2 = st.dv[s1];
  const F4 hd2 = ShiftFromLeft(dv2) + ShiftFromRight(dv2);    // dv[y][x-1] + dv[y][x+1]
  const F4 hd1 = st.hd[s0];
  if (y >= k.y0 && y < k.y1) {   // uniform over the wavefront except for its groups in a shorter last segment
    // EPF pass 1 of row y: neighbours up / left / right / down, SAD over the plus-shaped support
    const F4 dh0 = st.dh[s0], dh1 = st.dh[s1], dh2 = st.dh[s2];
    const F4 a = dv1 + dv2;
    const F4 sad_u = a + dv0 + hd1, sad_d = a + dvn + hd2;
    const F4 vh = dh0 + dh1 + dh2;
    const F4 sad_r = vh + (ShiftFromLeft(dh1) + ShiftFromRight(dh1));
    const F4 sad_l = ShiftFromLeft(vh) + (ShiftFromLeft2(dh1) + dh1);
    const bool yb = ((y & 7) == 0) || ((y & 7) == 7);
    const float inv_in = is * (yb ? k.bsm : 1.0f), inv_b = is * k.bsm;
    const F4 inv = {k.xb0 ? inv_b : inv_in, inv_in, inv_in, k.xb3 ? inv_b : inv_in};
    const bool skip = is < -3.90524291751269967465540850526868f;
    F4 wu, wl, wr, wd, iw;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      wu[j] = fmaxf(0.f, 1.0f + sad_u[j] * inv[j]); wl[j] = fmaxf(0.f, 1.0f + sad_l[j] * inv[j]);
      wr[j] = fmaxf(0.f, 1.0f + sad_r[j] * inv[j]); wd[j] = fmaxf(0.f, 1.0f + sad_d[j] * inv[j]);
      iw[j] = __builtin_amdgcn_rcpf(1.0f + wu[j] + wl[j] + wr[j] + wd[j]);
    }
    F4 o[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const F4 g1 = st.g[s1].c[c];
      const F4 f = (g1 + wu * st.g[s0].c[c] + wl * ShiftFromLeft(g1) + wr * ShiftFromRight(g1) + wd * st.g[s2].c[c]) * iw;
      o[c] = skip ? g1 : f;
    }
    if (k.stores && k.to_float) {
      // (the planes are padded to whole 8x8 cells: a quad that starts inside the frame ends inside the plane)
      const size_t g = (size_t)y * k.wp + k.X;
      *(JXL_GLOBAL F4*)(k.f0 + g) = o[0]; *(JXL_GLOBAL F4*)(k.f1 + g) = o[1]; *(JXL_GLOBAL F4*)(k.f2 + g) = o[2];
    } else if (k.stores) {
      if (k.plain && im.nch_out == 4 && k.full_quad) {
        typedef unsigned __attribute__((ext_vector_type(4))) U4v;
        U4v px;
        px.x = PixelToRgba8(im, o[0].x, o[1].x, o[2].x, al & 0xFF);
        px.y = PixelToRgba8(im, o[0].y, o[1].y, o[2].y, (al >> 8) & 0xFF);
        px.z = PixelToRgba8(im, o[0].z, o[1].z, o[2].z, (al >> 16) & 0xFF);
        px.w = PixelToRgba8(im, o[0].w, o[1].w, o[2].w, al >> 24);
        *(JXL_GLOBAL U4v*)((JXL_GLOBAL uint8_t*)im.out + ((size_t)(y - im.band_y0) * k.w + k.X) * 4) = px;
      } else {
#pragma unroll 1
        for (int j = 0; j < 4; j++) {
          if (k.X + j >= k.w) break;
          const uint32_t aj = !im.has_alpha ? 0u : LoadAlpha(im, (size_t)y * k.w + k.X + j);   // (the packed quad `al` is only valid for whole quads)
          const float ox = j == 0 ? o[0].x : (j == 1 ? o[0].y : (j == 2 ? o[0].z : o[0].w));
          const float oy = j == 0 ? o[1].x : (j == 1 ? o[1].y : (j == 2 ? o[1].z : o[1].w));
          const float ob = j == 0 ? o[2].x : (j == 1 ? o[2].y : (j == 2 ? o[2].z : o[2].w));
          WritePixelGeneral(im, k.X + j, y, ox, oy, ob, aj);
        }
      }
    }
  }
  st.dv[s2] = dvn;   // dv[r-2]
  st.dh[s3] = dhn;   // dh[r-1]
  st.hd[s1] = hd2;   // of dv[r-3]: next row's hd1
}

__global__ __launch_bounds__(256, 2) void filter_stream_kernel(const DevImage* __restrict__ imgs) {
  JXL_PIXEL_PRIO();
  const DevImage& im = imgs[blockIdx.y];
  if (!im.fused_gab_epf1 || (im.stream_pairs & 1)) return;   // (the common layouts run in filter_stream_pairs_kernel)
  StreamConst k;
  k.w = im.w; k.h = im.h; k.wp = im.wp; k.w8 = im.w8;
  const int strips = (k.w + kStripOut - 1) / kStripOut;
  // two EPF iterations: this kernel's rows feed the second iteration, which looks one row up and down
  k.to_float = im.fused_gab_epf1 == 2;
  const int band_lo = k.to_float ? max(0, im.band_y0 - 1) : im.band_y0, band_hi = k.to_float ? min(k.h, im.band_y1 + 1) : im.band_y1;
  const int band_rows = band_hi - band_lo;
  const int segs = (band_rows + kSegRows - 1) / kSegRows;
  if ((int)blockIdx.x * kGroupsPerWg >= strips * segs) return;   // whole workgroup past the end
  // every lane stays in the loop (DPP reads neighbouring lanes); groups past the end repeat the last task and store nothing
  const int gidx = blockIdx.x * kGroupsPerWg + (threadIdx.x / kStripLanes);
  const bool task = gidx < strips * segs;
  const int gi = task ? gidx : strips * segs - 1;
  const int strip = gi % strips, seg = gi / strips;
  const int q = threadIdx.x & (kStripLanes - 1);
  k.X = strip * kStripOut - 4 + 4 * q;           // first column of this lane's quad
  k.y0 = band_lo + seg * kSegRows; k.y1 = min(k.y0 + kSegRows, band_hi);
  k.r_end = k.y1 + 3;
  k.interior = k.X >= 0 && k.X + 3 < k.w;
  k.stores = task && q >= 1 && q <= kStripLanes - 2 && k.X < k.w;
  k.full_quad = k.X + 3 < k.w;
  k.in0 = (const JXL_GLOBAL float*)im.stream_in[0];
  k.in1 = (const JXL_GLOBAL float*)im.stream_in[1];
  k.in2 = (const JXL_GLOBAL float*)im.stream_in[2];
  const bool no_gab = im.stream_no_gab != 0;
  k.inv_sigma = (const JXL_GLOBAL float*)im.inv_sigma;
  k.alpha = (const JXL_GLOBAL uint8_t*)im.alpha;
  k.plain = PlainOutput(im);
  k.f0 = (JXL_GLOBAL float*)im.stream_mid[0]; k.f1 = (JXL_GLOBAL float*)im.stream_mid[1]; k.f2 = (JXL_GLOBAL float*)im.stream_mid[2];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    // without Gaborish the kernel's weights are the identity: x * 1 + a * 0 + b * 0 is x exactly for finite planes
    k.gw0[c] = no_gab ? 1.0f : im.gab_w[c][0]; k.gw1[c] = no_gab ? 0.0f : im.gab_w[c][1]; k.gw2[c] = no_gab ? 0.0f : im.gab_w[c][2];
    k.cs[c] = im.epf_channel_scale[c];
  }
  k.bsm = im.epf_border_sad_mul;
  // block-border columns of the quad: strips start on a multiple of 8, so X = 4 (q - 1) mod 8
  k.xb0 = (q & 1) != 0; k.xb3 = (q & 1) == 0;     // component 0 is column 0 of a block / component 3 is column 7
  k.cellx = min(max(k.X, 0), k.w - 1) >> 3;
  k.xa = min(max(k.X, 0), max(k.w - 4, 0));       // alpha quad (clamped: halo lanes and partial quads never use it)
  StreamState st;
  const F4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int c = 0; c < 3; c++) { st.in[i].c[c] = zero; st.s[i].c[c] = zero; st.g[i].c[c] = zero; }
    st.dv[i] = zero; st.dh[i] = zero; st.hd[i] = zero;
  }
  const int r0 = k.y0 - 3;
  Row3 next = LoadRow3(k.in0, k.in1, k.in2, k.X, r0, k.w, k.h, k.wp, k.interior);
  // four rows per trip, unconditionally (rows past the end recompute the last input row and store nothing: y >= y1): with
  // conditional phases the compiler must keep every slot of every window alive across the back edge
  for (int r = r0; r < k.r_end; r += 4) {
    StreamRow<0>(im, k, st, next, r);
    StreamRow<1>(im, k, st, next, r + 1);
    StreamRow<2>(im, k, st, next, r + 2);
    StreamRow<3>(im, k, st, next, r + 3);
  }
}

// ---- The same kernel for the common layouts, TWO pixels per lane --------------------------------------------------------------------
// (even width >= 8; RGBA8 output with an alpha plane, or the f32 rows of a first of two iterations.)  What it changes, and why:
//  * half the window state per lane -> 128 registers instead of 252: four wavefronts per SIMD instead of two, and a wavefront of
//    this kernel still fits on a SIMD that an entropy wavefront of another batch occupies (measured in the pipelined step: with 252
//    registers ONE wavefront fitted beside an entropy wavefront, the kernel ran at half speed whenever the chains overlapped);
//  * buffer addressing (a 128-bit resource in scalar registers + a scalar row offset + the lane's 32-bit byte offset): no per-lane
//    64-bit addresses to compute or keep;
//  * no load sits in a branch: the pair left of column 0 and the pairs right of column w - 1 are pairs inside the frame, mirrored -
//    such a lane loads the pair it mirrors and swaps it.  (With loads in divergent branches the compiler loses count of the loads in
//    flight at the join and waits for ALL of them, the prefetched row included);
//  * the sigma cell and the alpha pair of a row are requested one row ahead, like the input pairs.
// A strip = 128 columns, the outer two pairs each side are halo (the filters reach 3 columns): 120 output columns.
typedef float __attribute__((ext_vector_type(2))) F2;
typedef unsigned __attribute__((ext_vector_type(2))) U2v;
constexpr int kPairOut = 2 * (kStripLanes - 4);
__device__ __forceinline__ F2 ShiftFromLeft(F2 v) { return F2{DppFromLeft(v.y), v.x}; }
__device__ __forceinline__ F2 ShiftFromRight(F2 v) { return F2{v.y, DppFromRight(v.x)}; }
__device__ __forceinline__ F2 ShiftFromLeft2(F2 v) { return F2{DppFromLeft(v.x), DppFromLeft(v.y)}; }
__device__ __forceinline__ F2 Abs2(F2 v) { return F2{fabsf(v.x), fabsf(v.y)}; }
struct PairRow { F2 c[3]; };
struct PairState {
  PairRow in[4], s[4], g[4];
  F2 dv[4], dh[4], hd[4];
};
struct PairConst {
  __amdgpu_buffer_rsrc_t in0, in1, in2, sigma, alpha, out, f0, f1, f2;
  float gw0[3], gw1[3], gw2[3], cs[3], bsm;
  int w, h, wp, w8, y0, y1, r_end, band_y0;
  uint32_t in_bytes, cell_bytes, alpha_bytes, out_bytes;   // the lane's byte offsets inside a row: mirrored input pair, sigma cell, alpha pair, pair
  bool outside, stores, to_float, use_alpha, xb_lo, xb_hi;
  int bpp;   // bytes per output pixel: 4 RGBA, 3 RGB, 2 gray + alpha, 1 gray
};
struct PairAux { float is; uint32_t al; };
// Two output pixels of a lane (p = R | G << 8 | B << 16 | A << 24 each): 8 bytes RGBA; 6 bytes RGB as three 16-bit stores (the pair starts
// on an even column: 2-byte aligned); gray + alpha as one 32-bit store; gray as one 16-bit store (gray frames: R = G = B)
__device__ __forceinline__ void StorePixelPair(const PairConst& k, uint32_t p0, uint32_t p1, int row) {
  const uint32_t x = k.out_bytes >> 2;   // first column of the pair
  if (k.bpp == 4) {
    U2v px; px.x = p0; px.y = p1;
    __builtin_amdgcn_raw_buffer_store_b64(px, k.out, k.out_bytes, (uint32_t)row * (uint32_t)k.w * 4u, 0);
  } else if (k.bpp == 3) {
    const uint32_t soff = (uint32_t)row * (uint32_t)k.w * 3u, v = x * 3u;
    __builtin_amdgcn_raw_buffer_store_b16((short)(p0 & 0xFFFFu), k.out, v, soff, 0);
    __builtin_amdgcn_raw_buffer_store_b16((short)(((p0 >> 16) & 0xFFu) | ((p1 & 0xFFu) << 8)), k.out, v + 2, soff, 0);
    __builtin_amdgcn_raw_buffer_store_b16((short)((p1 >> 8) & 0xFFFFu), k.out, v + 4, soff, 0);
  } else if (k.bpp == 2) {
    const uint32_t ga = ((p0 >> 8) & 0xFFu) | ((p0 >> 24) << 8) | (((p1 >> 8) & 0xFFu) << 16) | ((p1 >> 24) << 24);
    __builtin_amdgcn_raw_buffer_store_b32(ga, k.out, x * 2u, (uint32_t)row * (uint32_t)k.w * 2u, 0);
  } else {
    __builtin_amdgcn_raw_buffer_store_b16((short)(((p0 >> 8) & 0xFFu) | (((p1 >> 8) & 0xFFu) << 8)), k.out, x, (uint32_t)row * (uint32_t)k.w, 0);
  }
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t PlaneResource(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7FFFFFFF, 0x00020000);   // raw buffer, no swizzle; the planes are smaller than 2 GB
}
__device__ __forceinline__ PairRow LoadPairRow(const PairConst& k, int y) {
  const int yy = y < 0 ? -y - 1 : (y >= k.h ? 2 * k.h - 1 - y : y);   // uniform over the wavefront
  const uint32_t row = (uint32_t)yy * (uint32_t)k.wp * 4u;
  PairRow r;
  r.c[0] = __builtin_bit_cast(F2, __builtin_amdgcn_raw_buffer_load_b64(k.in0, k.in_bytes, row, 0));
  r.c[1] = __builtin_bit_cast(F2, __builtin_amdgcn_raw_buffer_load_b64(k.in1, k.in_bytes, row, 0));
  r.c[2] = __builtin_bit_cast(F2, __builtin_amdgcn_raw_buffer_load_b64(k.in2, k.in_bytes, row, 0));
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const F2 v = r.c[c];
    r.c[c] = k.outside ? F2{v.y, v.x} : v;
  }
  return r;
}
__device__ __forceinline__ PairAux LoadPairAux(const PairConst& k, int y) {
  const int yc = min(max(y, 0), k.h - 1);
  PairAux a;
  a.is = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(k.sigma, k.cell_bytes, (uint32_t)(yc >> 3) * (uint32_t)k.w8 * 4u, 0));
  a.al = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(k.alpha, k.alpha_bytes, (uint32_t)yc * (uint32_t)k.w, 0);
  return a;
}
template <int P>
__device__ __forceinline__ void PairStreamRow(const DevImage& im, const PairConst& k, PairState& st, PairRow& next, PairAux& next_aux, int r) {
  constexpr int s0 = P, s1 = (P + 1) & 3, s2 = (P + 2) & 3, s3 = (P + 3) & 3;   // slots of rows r (= r - 4), r - 3, r - 2, r - 1
  const PairRow cur = next;
  const PairAux aux = next_aux;
  // all the next row reads from memory, a whole row of arithmetic ahead of its use
  next = LoadPairRow(k, min(r + 1, k.r_end - 1));
  next_aux = LoadPairAux(k, r - 2);
  const int y = r - 3;   // the output row
  const float is = aux.is;
  const uint32_t al = k.use_alpha ? aux.al : 0xFFFFu;
  const F2 zero = {0.f, 0.f};
  F2 dhn = zero, dvn = zero;
  st.in[s0] = cur;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    st.s[s0].c[c] = ShiftFromLeft(cur.c[c]) + ShiftFromRight(cur.c[c]);
    const F2 gn = st.in[s3].c[c] * k.gw0[c] + ((st.in[s2].c[c] + cur.c[c]) + st.s[s3].c[c]) * k.gw1[c] + (st.s[s2].c[c] + st.s[s0].c[c]) * k.gw2[c];
    dhn += Abs2(gn - ShiftFromRight(gn)) * k.cs[c];
    dvn += Abs2(st.g[s2].c[c] - gn) * k.cs[c];
    st.g[s3].c[c] = gn;
  }
  const F2 dv0 = st.dv[s3], dv1 = st.dv[s0], dv2 = st.dv[s1];
  const F2 hd2 = ShiftFromLeft(dv2) + ShiftFromRight(dv2);
  const F2 hd1 = st.hd[s0];
  if (y >= k.y0 && y < k.y1) {   // uniform over the wavefront
    const F2 dh0 = st.dh[s0], dh1 = st.dh[s1], dh2 = st.dh[s2];
    const F2 a = dv1 + dv2;
    const F2 sad_u = a + dv0 + hd1, sad_d = a + dvn + hd2;
    const F2 vh = dh0 + dh1 + dh2;
    const F2 sad_r = vh + (ShiftFromLeft(dh1) + ShiftFromRight(dh1));
    const F2 sad_l = ShiftFromLeft(vh) + (ShiftFromLeft2(dh1) + dh1);
    const bool yb = ((y & 7) == 0) || ((y & 7) == 7);
    const float inv_in = is * (yb ? k.bsm : 1.0f), inv_b = is * k.bsm;
    const F2 inv = {k.xb_lo ? inv_b : inv_in, k.xb_hi ? inv_b : inv_in};
    const bool skip = is < -3.90524291751269967465540850526868f;
    F2 wu, wl, wr, wd, iw;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      wu[j] = fmaxf(0.f, 1.0f + sad_u[j] * inv[j]); wl[j] = fmaxf(0.f, 1.0f + sad_l[j] * inv[j]);
      wr[j] = fmaxf(0.f, 1.0f + sad_r[j] * inv[j]); wd[j] = fmaxf(0.f, 1.0f + sad_d[j] * inv[j]);
      iw[j] = __builtin_amdgcn_rcpf(1.0f + wu[j] + wl[j] + wr[j] + wd[j]);
    }
    F2 o[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const F2 g1 = st.g[s1].c[c];
      const F2 f = (g1 + wu * st.g[s0].c[c] + wl * ShiftFromLeft(g1) + wr * ShiftFromRight(g1) + wd * st.g[s2].c[c]) * iw;
      o[c] = skip ? g1 : f;
    }
    if (k.stores) {
      if (k.to_float) {
        const uint32_t row = (uint32_t)y * (uint32_t)k.wp * 4u;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(U2v, o[0]), k.f0, k.out_bytes, row, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(U2v, o[1]), k.f1, k.out_bytes, row, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(U2v, o[2]), k.f2, k.out_bytes, row, 0);
      } else {
        StorePixelPair(k, PixelToRgba8(im, o[0].x, o[1].x, o[2].x, al & 0xFF), PixelToRgba8(im, o[0].y, o[1].y, o[2].y, (al >> 8) & 0xFF), y - k.band_y0);
      }
    }
  }
  st.dv[s2] = dvn;
  st.dh[s3] = dhn;
  st.hd[s1] = hd2;
}

__global__ __launch_bounds__(256, 4) void filter_stream_pairs_kernel(const DevImage* __restrict__ imgs) {
  JXL_PIXEL_PRIO();
  const DevImage& im = imgs[blockIdx.y];
  if (!im.fused_gab_epf1 || !(im.stream_pairs & 1)) return;
  PairConst k;
  k.w = im.w; k.h = im.h; k.wp = im.wp; k.w8 = im.w8; k.band_y0 = im.band_y0;
  const int strips = (k.w + kPairOut - 1) / kPairOut;
  k.to_float = im.fused_gab_epf1 == 2;
  const int band_lo = k.to_float ? max(0, im.band_y0 - 1) : im.band_y0, band_hi = k.to_float ? min(k.h, im.band_y1 + 1) : im.band_y1;
  const int segs = (band_hi - band_lo + kSegRows - 1) / kSegRows;
  if ((int)blockIdx.x * kGroupsPerWg >= strips * segs) return;
  // (the wavefront's index, given to the compiler as the uniform value it is: rows, segment bounds and row offsets are scalar)
  const int gidx = blockIdx.x * kGroupsPerWg + __builtin_amdgcn_readfirstlane(threadIdx.x / kStripLanes);
  const bool task = gidx < strips * segs;
  const int gi = task ? gidx : strips * segs - 1;   // (every lane stays in the loop: DPP reads neighbouring lanes)
  const int strip = gi % strips, seg = gi / strips;
  const int q = threadIdx.x & (kStripLanes - 1);
  const int X = strip * kPairOut - 4 + 2 * q;       // first column of the lane's pair; strips start on multiples of 8
  k.y0 = band_lo + seg * kSegRows; k.y1 = min(k.y0 + kSegRows, band_hi);
  k.r_end = k.y1 + 3;
  k.outside = X < 0 || X >= k.w;
  const int xm = X < 0 ? -X - 2 : (X >= k.w ? 2 * k.w - 2 - X : X);   // the pair this lane loads: itself, or the pair it mirrors
  k.in_bytes = (uint32_t)min(max(xm, 0), k.w - 2) * 4u;
  const int xc = min(max(X, 0), k.w - 2);
  k.cell_bytes = (uint32_t)(xc >> 3) * 4u;
  k.alpha_bytes = (uint32_t)xc;
  k.out_bytes = (uint32_t)xc * 4u;
  k.stores = task && q >= 2 && q <= kStripLanes - 3 && X < k.w;
  k.xb_lo = (X & 7) == 0; k.xb_hi = (X & 7) == 6;   // component 0 is column 0 of a block / component 1 is column 7
  k.use_alpha = im.has_alpha && !k.to_float;
  k.bpp = im.nch_out;
  k.in0 = PlaneResource(im.stream_in[0]); k.in1 = PlaneResource(im.stream_in[1]); k.in2 = PlaneResource(im.stream_in[2]);
  const bool no_gab = im.stream_no_gab != 0;
  k.sigma = PlaneResource(im.inv_sigma);
  k.alpha = k.use_alpha ? PlaneResource(im.alpha) : k.in0;   // (ignored without alpha: any readable bytes)
  k.out = PlaneResource(im.out);
  k.f0 = PlaneResource(im.stream_mid[0]); k.f1 = PlaneResource(im.stream_mid[1]); k.f2 = PlaneResource(im.stream_mid[2]);
#pragma unroll
  for (int c = 0; c < 3; c++) {
    // without Gaborish the kernel's weights are the identity: x * 1 + a * 0 + b * 0 is x exactly for finite planes
    k.gw0[c] = no_gab ? 1.0f : im.gab_w[c][0]; k.gw1[c] = no_gab ? 0.0f : im.gab_w[c][1]; k.gw2[c] = no_gab ? 0.0f : im.gab_w[c][2];
    k.cs[c] = im.epf_channel_scale[c];
  }
  k.bsm = im.epf_border_sad_mul;
  PairState st;
  const F2 zero = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int c = 0; c < 3; c++) { st.in[i].c[c] = zero; st.s[i].c[c] = zero; st.g[i].c[c] = zero; }
    st.dv[i] = zero; st.dh[i] = zero; st.hd[i] = zero;
  }
  const int r0 = k.y0 - 3;
  PairRow next = LoadPairRow(k, r0);
  PairAux next_aux = LoadPairAux(k, r0 - 3);
  for (int r = r0; r < k.r_end; r += 4) {   // four rows per trip, unconditionally (see filter_stream_kernel)
    PairStreamRow<0>(im, k, st, next, next_aux, r);
    PairStreamRow<1>(im, k, st, next, next_aux, r + 1);
    PairStreamRow<2>(im, k, st, next, next_aux, r + 2);
    PairStreamRow<3>(im, k, st, next, next_aux, r + 3);
  }
}

// The second EPF iteration (frames with two: distance 1.5 ... 4) + XYB -> output samples, register streaming like the kernel above
// but much lighter: three rows of the first iteration's output (f32 planes), four neighbours, a one-pixel SAD.  Same strip geometry
// (a wavefront = 256 columns, the outer quads are halo), same output paths.  With it a two-iteration frame runs
// reconstruction -> two streaming kernels instead of three LDS-tiled stage kernels (measured, 384 4K frames at distance 2:
// filters + output 104.8 ms before).
__global__ __launch_bounds__(256) void filter_stream2_kernel(const DevImage* __restrict__ imgs) {
  JXL_PIXEL_PRIO();
  const DevImage& im = imgs[blockIdx.y];
  if (im.fused_gab_epf1 != 2 || (im.stream_pairs & 2)) return;   // (RGBA8 frames of even width: filter_stream2_pairs_kernel)
  const int w = im.w, h = im.h, wp = im.wp;
  const int strips = (w + kStripOut - 1) / kStripOut;
  const int band_rows = im.band_y1 - im.band_y0;
  const int segs = (band_rows + kSegRows - 1) / kSegRows;
  if ((int)blockIdx.x * kGroupsPerWg >= strips * segs) return;
  const int gidx = blockIdx.x * kGroupsPerWg + (threadIdx.x / kStripLanes);
  const bool task = gidx < strips * segs;
  const int gi = task ? gidx : strips * segs - 1;
  const int strip = gi % strips, seg = gi / strips;
  const int q = threadIdx.x & (kStripLanes - 1);
  const int X = strip * kStripOut - 4 + 4 * q;
  const int y0 = im.band_y0 + seg * kSegRows, y1 = min(y0 + kSegRows, im.band_y1);
  const bool interior = X >= 0 && X + 3 < w;
  const bool stores = task && q >= 1 && q <= kStripLanes - 2 && X < w;
  const bool full_quad = X + 3 < w;
  const JXL_GLOBAL float* in0 = (const JXL_GLOBAL float*)im.stream_mid[0];
  const JXL_GLOBAL float* in1 = (const JXL_GLOBAL float*)im.stream_mid[1];
  const JXL_GLOBAL float* in2 = (const JXL_GLOBAL float*)im.stream_mid[2];
  const JXL_GLOBAL float* inv_sigma = (const JXL_GLOBAL float*)im.inv_sigma;
  const bool plain = PlainOutput(im);
  const float cs0 = im.epf_channel_scale[0], cs1 = im.epf_channel_scale[1], cs2 = im.epf_channel_scale[2];
  const float sm = im.epf_pass2_sigma_scale, smb = sm * im.epf_border_sad_mul;
  const bool xb0 = (q & 1) != 0, xb3 = (q & 1) == 0;
  const int cellx = min(max(X, 0), w - 1) >> 3;
  const int xa = min(max(X, 0), max(w - 4, 0));
  Row3 prev = LoadRow3(in0, in1, in2, X, y0 - 1, w, h, wp, interior);
  Row3 cur = LoadRow3(in0, in1, in2, X, y0, w, h, wp, interior);
  for (int y = y0; y < y1; y++) {
    const Row3 next = LoadRow3(in0, in1, in2, X, y + 1, w, h, wp, interior);
    const float is = inv_sigma[(size_t)(y >> 3) * im.w8 + cellx];
    uint32_t al = 0xFFFFFFFFu;
    if (im.has_alpha && plain) {
      const JXL_GLOBAL uint8_t* ap = (const JXL_GLOBAL uint8_t*)im.alpha + (size_t)y * w + xa;
      al = (uint32_t)ap[0] | (uint32_t)ap[1] << 8 | (uint32_t)ap[2] << 16 | (uint32_t)ap[3] << 24;
    }
    const F4 zero = {0.f, 0.f, 0.f, 0.f};
    F4 su = zero, sd = zero, sl = zero, sr = zero;
    F4 lft[3], rgt[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float sc = c == 0 ? cs0 : (c == 1 ? cs1 : cs2);
      lft[c] = ShiftFromLeft(cur.c[c]); rgt[c] = ShiftFromRight(cur.c[c]);
      su += Abs4(cur.c[c] - prev.c[c]) * sc; sd += Abs4(cur.c[c] - next.c[c]) * sc;
      sl += Abs4(cur.c[c] - lft[c]) * sc; sr += Abs4(cur.c[c] - rgt[c]) * sc;
    }
    const bool yb = ((y & 7) == 0) || ((y & 7) == 7);
    const float inv_in = is * (yb ? smb : sm), inv_b = is * smb;
    const F4 inv = {xb0 ? inv_b : inv_in, inv_in, inv_in, xb3 ? inv_b : inv_in};
    const bool skip = is < -3.90524291751269967465540850526868f;
    F4 wu, wl, wr, wd, iw;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      wu[j] = fmaxf(0.f, 1.0f + su[j] * inv[j]); wl[j] = fmaxf(0.f, 1.0f + sl[j] * inv[j]);
      wr[j] = fmaxf(0.f, 1.0f + sr[j] * inv[j]); wd[j] = fmaxf(0.f, 1.0f + sd[j] * inv[j]);
      iw[j] = __builtin_amdgcn_rcpf(1.0f + wu[j] + wl[j] + wr[j] + wd[j]);
    }
    F4 o[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const F4 f = (cur.c[c] + wu * prev.c[c] + wl * lft[c] + wr * rgt[c] + wd * next.c[c]) * iw;
      o[c] = skip ? cur.c[c] : f;
    }
    if (stores) {
      if (plain && im.nch_out == 4 && full_quad) {
        typedef unsigned __attribute__((ext_vector_type(4))) U4v;
        U4v px;
        px.x = PixelToRgba8(im, o[0].x, o[1].x, o[2].x, al & 0xFF);
        px.y = PixelToRgba8(im, o[0].y, o[1].y, o[2].y, (al >> 8) & 0xFF);
        px.z = PixelToRgba8(im, o[0].z, o[1].z, o[2].z, (al >> 16) & 0xFF);
        px.w = PixelToRgba8(im, o[0].w, o[1].w, o[2].w, al >> 24);
        *(JXL_GLOBAL U4v*)((JXL_GLOBAL uint8_t*)im.out + ((size_t)(y - im.band_y0) * w + X) * 4) = px;
      } else {
#pragma unroll 1
        for (int j = 0; j < 4; j++) {
          if (X + j >= w) break;
          const uint32_t aj = !im.has_alpha ? 0u : LoadAlpha(im, (size_t)y * w + X + j);
          const float ox = j == 0 ? o[0].x : (j == 1 ? o[0].y : (j == 2 ? o[0].z : o[0].w));
          const float oy = j == 0 ? o[1].x : (j == 1 ? o[1].y : (j == 2 ? o[1].z : o[1].w));
          const float ob = j == 0 ? o[2].x : (j == 1 ? o[2].y : (j == 2 ? o[2].z : o[2].w));
          WritePixelGeneral(im, X + j, y, ox, oy, ob, aj);
        }
      }
    }
    prev = cur;
    cur = next;
  }
}

// The second iteration in the two-pixels-per-lane form (RGBA8 + alpha frames of even width): same addressing and edge handling as
// filter_stream_pairs_kernel, one halo pair each side (the iteration reaches one column): 124 output columns per strip.
constexpr int kPair2Out = 2 * (kStripLanes - 2);
__global__ __launch_bounds__(256, 4) void filter_stream2_pairs_kernel(const DevImage* __restrict__ imgs) {
  JXL_PIXEL_PRIO();
  const DevImage& im = imgs[blockIdx.y];
  if (im.fused_gab_epf1 != 2 || !(im.stream_pairs & 2)) return;
  PairConst k;
  k.w = im.w; k.h = im.h; k.wp = im.wp; k.w8 = im.w8; k.band_y0 = im.band_y0;
  const int strips = (k.w + kPair2Out - 1) / kPair2Out;
  const int segs = (im.band_y1 - im.band_y0 + kSegRows - 1) / kSegRows;
  if ((int)blockIdx.x * kGroupsPerWg >= strips * segs) return;
  const int gidx = blockIdx.x * kGroupsPerWg + __builtin_amdgcn_readfirstlane(threadIdx.x / kStripLanes);
  const bool task = gidx < strips * segs;
  const int gi = task ? gidx : strips * segs - 1;
  const int strip = gi % strips, seg = gi / strips;
  const int q = threadIdx.x & (kStripLanes - 1);
  const int X = strip * kPair2Out - 2 + 2 * q;      // strips start on multiples of 4
  k.y0 = im.band_y0 + seg * kSegRows; k.y1 = min(k.y0 + kSegRows, im.band_y1);
  k.r_end = k.y1 + 1;                               // last input row + 1
  k.outside = X < 0 || X >= k.w;
  const int xm = X < 0 ? -X - 2 : (X >= k.w ? 2 * k.w - 2 - X : X);
  k.in_bytes = (uint32_t)min(max(xm, 0), k.w - 2) * 4u;
  const int xc = min(max(X, 0), k.w - 2);
  k.cell_bytes = (uint32_t)(xc >> 3) * 4u;
  k.alpha_bytes = (uint32_t)xc;
  k.out_bytes = (uint32_t)xc * 4u;
  k.stores = task && q >= 1 && q <= kStripLanes - 2 && X < k.w;
  k.xb_lo = (X & 7) == 0; k.xb_hi = (X & 7) == 6;
  k.use_alpha = im.has_alpha != 0;
  k.bpp = im.nch_out;
  k.in0 = PlaneResource(im.stream_mid[0]); k.in1 = PlaneResource(im.stream_mid[1]); k.in2 = PlaneResource(im.stream_mid[2]);
  k.sigma = PlaneResource(im.inv_sigma);
  k.alpha = k.use_alpha ? PlaneResource(im.alpha) : k.in0;
  k.out = PlaneResource(im.out);
  const float cs0 = im.epf_channel_scale[0], cs1 = im.epf_channel_scale[1], cs2 = im.epf_channel_scale[2];
  const float sm = im.epf_pass2_sigma_scale, smb = sm * im.epf_border_sad_mul;
  PairRow prev = LoadPairRow(k, k.y0 - 1);
  PairRow cur = LoadPairRow(k, k.y0);
  PairRow next = LoadPairRow(k, min(k.y0 + 1, k.r_end - 1));
  PairAux aux = LoadPairAux(k, k.y0);
  for (int y = k.y0; y < k.y1; y++) {
    // two rows ahead for the pairs, one for sigma / alpha: nothing this row uses was requested in it
    const PairRow next2 = LoadPairRow(k, min(y + 2, k.r_end - 1));
    const PairAux aux2 = LoadPairAux(k, y + 1);
    const float is = aux.is;
    const uint32_t al = k.use_alpha ? aux.al : 0xFFFFu;
    const F2 zero = {0.f, 0.f};
    F2 su = zero, sd = zero, sl = zero, sr = zero;
    F2 lft[3], rgt[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float sc = c == 0 ? cs0 : (c == 1 ? cs1 : cs2);
      lft[c] = ShiftFromLeft(cur.c[c]); rgt[c] = ShiftFromRight(cur.c[c]);
      su += Abs2(cur.c[c] - prev.c[c]) * sc; sd += Abs2(cur.c[c] - next.c[c]) * sc;
      sl += Abs2(cur.c[c] - lft[c]) * sc; sr += Abs2(cur.c[c] - rgt[c]) * sc;
    }
    const bool yb = ((y & 7) == 0) || ((y & 7) == 7);
    const float inv_in = is * (yb ? smb : sm), inv_b = is * smb;
    const F2 inv = {k.xb_lo ? inv_b : inv_in, k.xb_hi ? inv_b : inv_in};
    const bool skip = is < -3.90524291751269967465540850526868f;
    F2 wu, wl, wr, wd, iw;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      wu[j] = fmaxf(0.f, 1.0f + su[j] * inv[j]); wl[j] = fmaxf(0.f, 1.0f + sl[j] * inv[j]);
      wr[j] = fmaxf(0.f, 1.0f + sr[j] * inv[j]); wd[j] = fmaxf(0.f, 1.0f + sd[j] * inv[j]);
      iw[j] = __builtin_amdgcn_rcpf(1.0f + wu[j] + wl[j] + wr[j] + wd[j]);
    }
    F2 o[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const F2 f = (cur.c[c] + wu * prev.c[c] + wl * lft[c] + wr * rgt[c] + wd * next.c[c]) * iw;
      o[c] = skip ? cur.c[c] : f;
    }
    if (k.stores) {
      StorePixelPair(k, PixelToRgba8(im, o[0].x, o[1].x, o[2].x, al & 0xFF), PixelToRgba8(im, o[0].y, o[1].y, o[2].y, (al >> 8) & 0xFF), y - k.band_y0);
    }
    prev = cur; cur = next; next = next2; aux = aux2;
  }
}

// no loop filter at all: plain conversion
__global__ void out_only_kernel(const DevImage* __restrict__ imgs) {
  const DevImage& im = imgs[blockIdx.y];
  if (im.final_stage != 4) return;
  const int w = im.w, wp = im.wp;
  const size_t n = (size_t)w * (im.band_y1 - im.band_y0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % w), y = im.band_y0 + (int)(i / w);
    const size_t o = (size_t)y * wp + x;
    WritePixel(im, x, y, im.stage_in[4][0][o], im.stage_in[4][1][o], im.stage_in[4][2][o]);
  }
}

void LaunchReconTiles(const DevImage* imgs, int nimg, int max_tiles, const float* basis_all, const float* basis_small,
                      const float* llf_scale, const float* basis_mfma, hipStream_t s) {
  // three tiles, B816, four per-cell words + 2 x 3 prefix words per cell, totals, per-cell scan-list pointers
  const size_t lds = (size_t)(3 * kTileF + 320 + 64 * 4 + 64 * 6 + 4 + 128 + 1024 + 4096) * 4;   // + the 32- and 64-point matrix-core tables
  static bool raised = false;
  if (!raised) { (void)hipFuncSetAttribute((const void*)recon_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); raised = true; }
  hipLaunchKernelGGL(recon_tile_kernel, dim3(max_tiles, nimg), dim3(768), lds, s, imgs, basis_all, basis_small, llf_scale, basis_mfma);
}

void LaunchFilterTiles(const DevImage* imgs, int nimg, int max_w, int max_h, int stage_mask, bool any_unfiltered,
                       int any_fused, int any_fused2, hipStream_t s) {   // any_fused / any_fused2: 1 = frames of the pair kernels, 2 = others
  const int tiles = ((max_w + 63) / 64) * ((max_h + 31) / 32);
  dim3 g(tiles, nimg);
  // Gaborish / iteration 0 as stage kernels: frames without EPF, frames with three iterations (before their streaming kernels), taps
  if (stage_mask & 1) hipLaunchKernelGGL(filter_tile_kernel<0>, g, dim3(256), 0, s, imgs);
  if (stage_mask & 2) hipLaunchKernelGGL(filter_tile_kernel<1>, g, dim3(256), 0, s, imgs);
  if (any_fused) {
    // one wavefront per (strip, segment); + 2 rows: the first of two streaming kernels also writes the rows its second one reads
    const int groups = ((max_w + kStripOut - 1) / kStripOut) * ((max_h + 2 + kSegRows - 1) / kSegRows);
    const int pair_groups = ((max_w + kPairOut - 1) / kPairOut) * ((max_h + 2 + kSegRows - 1) / kSegRows);
    if (any_fused & 1) hipLaunchKernelGGL(filter_stream_pairs_kernel, dim3((pair_groups + kGroupsPerWg - 1) / kGroupsPerWg, nimg), dim3(256), 0, s, imgs);
    if (any_fused & 2) hipLaunchKernelGGL(filter_stream_kernel, dim3((groups + kGroupsPerWg - 1) / kGroupsPerWg, nimg), dim3(256), 0, s, imgs);
    const int pair2_groups = ((max_w + kPair2Out - 1) / kPair2Out) * ((max_h + kSegRows - 1) / kSegRows);
    if (any_fused2 & 1) hipLaunchKernelGGL(filter_stream2_pairs_kernel, dim3((pair2_groups + kGroupsPerWg - 1) / kGroupsPerWg, nimg), dim3(256), 0, s, imgs);
    if (any_fused2 & 2) hipLaunchKernelGGL(filter_stream2_kernel, dim3((groups + kGroupsPerWg - 1) / kGroupsPerWg, nimg), dim3(256), 0, s, imgs);
  }
  if (stage_mask & 4) hipLaunchKernelGGL(filter_tile_kernel<2>, g, dim3(256), 0, s, imgs);   // (only with the stage taps)
  if (stage_mask & 8) hipLaunchKernelGGL(filter_tile_kernel<3>, g, dim3(256), 0, s, imgs);
  if (any_unfiltered) {
    size_t work = (size_t)max_w * max_h;
    size_t b = std::min<size_t>((work + 255) / 256, 8192);
    hipLaunchKernelGGL(out_only_kernel, dim3((unsigned)b, nimg), dim3(256), 0, s, imgs);
  }
}

}  // namespace jxlhip
