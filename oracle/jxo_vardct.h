// ORACLE — test infrastructure only (see jxo_common.h header).
// VarDCT building blocks of ISO/IEC 18181-1: AC strategies, coefficient orders, dequant
// matrices, (I)DCTs of all block shapes, LF->LLF, chroma-from-luma, Gaborish, EPF,
// XYB->sRGB.  The reference reaches these only inside JxlDecoderProcessInput
// (src/JxlFileTypeIO/Decoder/JxlDecoder.cpp:252).
#pragma once
#include "jxo_common.h"
#include "jxo_headers.h"

namespace jxo {

constexpr int kNumStrategies = 27;
constexpr int kNumOrders = 13;
constexpr int kNumQuantTables = 17;
enum Strategy {
  DCT8 = 0, IDENTITY, DCT2X2, DCT4X4, DCT16X16, DCT32X32, DCT16X8, DCT8X16, DCT32X8, DCT8X32, DCT32X16, DCT16X32,
  DCT4X8, DCT8X4, AFV0, AFV1, AFV2, AFV3, DCT64X64, DCT64X32, DCT32X64, DCT128X128, DCT128X64, DCT64X128, DCT256X256,
  DCT256X128, DCT128X256
};
extern const uint8_t kCoveredX[kNumStrategies];   // width in 8x8 blocks
extern const uint8_t kCoveredY[kNumStrategies];   // height in 8x8 blocks
extern const uint8_t kStrategyOrder[kNumStrategies];
extern const uint8_t kStrategyQuantTable[kNumStrategies];

struct Plane {
  int w = 0, h = 0;
  std::vector<float> d;
  Plane() {}
  Plane(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_, 0.f) {}
  float* Row(int y) { return d.data() + (size_t)y * w; }
  const float* Row(int y) const { return d.data() + (size_t)y * w; }
};

// Natural coefficient order of a strategy: order[k] = position in the stored block.
const std::vector<uint32_t>& NaturalOrder(int strategy);

// Dequantisation matrices (1/weight), per quant table: 3 channels x (rows*cols), stored layout.
struct DequantMatrices {
  std::vector<float> table[kNumQuantTables];  // size 3 * n
  size_t n[kNumQuantTables];
  void SetDefault();
  void Decode(BitReader& br);  // HfGlobal: all_default bit + explicit encodings
  // Encoder, test streams: every table written explicitly in its own encoding mode with the library's parameters scaled a little
  // (per table, by `seed`); the tables are then computed from the F16-rounded parameters, i.e. exactly what a decoder will see.
  void SetCustomAndWrite(uint32_t seed, BitWriter& bw);
  const float* Get(int strategy, int c) const {
    int q = kStrategyQuantTable[strategy];
    return table[q].data() + c * n[q];
  }
};

// Inverse / forward transforms between a stored coefficient block (cx*cy*64 floats, stored
// layout: short x long) and pixels (rows 8*cy, cols 8*cx) at `stride`.
void InverseTransform(int strategy, const float* coeffs, float* pixels, int stride);
void ForwardTransform(int strategy, const float* pixels, int stride, float* coeffs);
// LLF: fill the lowest cx*cy coefficients of `coeffs` from the LF samples (cy rows, cx cols at lf_stride).
void LlfFromLf(int strategy, const float* lf, int lf_stride, float* coeffs);
// Encoder: LF samples (block means) from stored coefficients' LLF.
void LfFromLlf(int strategy, const float* coeffs, float* lf, int lf_stride);

// Generic scaled DCT helpers (stored layout), exposed for tests.
void IdctStored(int R, int C, const float* stored, float* out, int stride);
void DctStored(int R, int C, const float* in, int stride, float* stored);

// Loop filters and colour (operate on three planes X,Y,B of the frame size).
void AdaptiveLfSmoothing(Plane lf[3], const float lf_factors[3]);
void Gaborish(Plane xyb[3], const LoopFilter& lf);
// inv_sigma: one value per 8x8 block (w8 x h8), as stored by the decoder (negative).
void Epf(Plane xyb[3], const LoopFilter& lf, const Plane& inv_sigma);
void XybToLinear(const ImageMetadata& m, Plane xyb[3]);   // in place: X,Y,B -> linear R,G,B
void LinearToXyb(Plane rgb[3]);                            // in place
float LinearToSrgb(float v);
float SrgbToLinear(float v);
// Enumerated colour encodings the reference's host knows by name (Decoder/JxlDecoder.cpp:36-108): D65, primaries sRGB / P3 /
// BT.2100, transfer linear / sRGB / BT.709 / PQ.  kind: 0 linear, 1 sRGB, 2 BT.709, 3 PQ; -1: not one of those.
int TransferKind(const ColorEncoding& c);
// Encoded value from display-linear (sign-symmetric like the reference's library); PQ takes the image's intensity target.
float EncodeTransfer(int kind, float v, float intensity_target, double gamma = 1.0);
float DecodeTransfer(int kind, float e, float intensity_target, double gamma = 1.0);
// 3x3 (row-major) linear sRGB -> linear RGB of the primaries (1 sRGB: identity, 9 BT.2100, 11 P3); false: other primaries.
bool MatrixFromSrgb(uint32_t primaries, double out[9]);
// The same for any enumerated encoding (custom chromaticities, white points D65 / E / DCI / custom): through XYZ adapted to D50 with
// the Bradford transform, as the decoder library behind the reference builds its output matrices.  false: not expressible.
bool MatrixFromSrgbGeneral(const ColorEncoding& c, double out[9]);
// Exponent of pure power-law transfer functions (encoded = linear ^ gamma): explicit gamma and DCI (1 / 2.6); 0 otherwise.
double PowerLawGamma(const ColorEncoding& c);

// Block-context map (HfBlockContext) of LfGlobal.
struct BlockCtxMap {
  std::vector<int32_t> lf_thresholds[3];
  std::vector<uint32_t> qf_thresholds;
  std::vector<uint8_t> ctx_map;
  uint32_t num_ctxs = 15;
  uint32_t num_lf_ctxs = 1;
  void SetDefault();
  void Decode(BitReader& br);
  uint32_t Context(uint32_t lf_idx, uint32_t qf, uint32_t ord, uint32_t c) const {
    uint32_t qf_idx = 0;
    for (uint32_t t : qf_thresholds) if (qf > t) qf_idx++;
    uint32_t idx = c < 2 ? (c ^ 1) : 2;
    idx = idx * kNumOrders + ord;
    idx = idx * (uint32_t)(qf_thresholds.size() + 1) + qf_idx;
    idx = idx * num_lf_ctxs + lf_idx;
    return ctx_map[idx];
  }
  uint32_t NumAcContexts() const { return num_ctxs * (37 + 458); }
  uint32_t NonZeroContext(uint32_t nz, uint32_t block_ctx) const {
    uint32_t ctx;
    if (nz >= 64) nz = 64;
    if (nz < 8) ctx = nz; else ctx = 4 + nz / 2;
    return ctx * num_ctxs + block_ctx;
  }
  uint32_t ZeroDensityContextsOffset(uint32_t block_ctx) const { return num_ctxs * 37 + 458 * block_ctx; }
};
extern const uint16_t kCoeffFreqContext[64];
extern const uint16_t kCoeffNumNonzeroContext[64];
static inline uint32_t ZeroDensityContext(uint32_t nz_left, uint32_t k, uint32_t covered, uint32_t log2_covered, uint32_t prev) {
  nz_left = (nz_left + covered - 1) >> log2_covered;
  k >>= log2_covered;
  return (kCoeffNumNonzeroContext[nz_left] + kCoeffFreqContext[k]) * 2 + prev;
}

}  // namespace jxo
