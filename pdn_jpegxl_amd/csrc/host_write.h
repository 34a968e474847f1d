// Host-side writers of the MI355X encode path: everything that is small and serial in a JPEG XL file
// (signature, image / frame headers, TOC, MA tree, entropy-code headers, container boxes).  The bulk
// data (LF groups, pass groups) is tokenised and ANS-coded by the kernels of encode_kernels.hip.
//
// Product-side counterpart of the libjxl calls the reference makes in Encoder/JxlEncoder.cpp:147-392
// (JxlEncoderSetBasicInfo :247, SetColorEncoding :274, UseBoxes/AddBox :201,284-310, AddImageFrame :128).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#include "dev_types.h"

namespace jxlhip {

// LSB-first bit writer (the codestream's bit order).
class BitWriter {
 public:
  void Write(int nbits, uint64_t value);   // nbits <= 56
  void Bool(bool b) { Write(1, b ? 1 : 0); }
  struct Dist { int bits; uint32_t off; };   // bits < 0: the constant `off`
  void U32(Dist a, Dist b, Dist c, Dist d, uint32_t value);
  void U64(uint64_t value);
  void Enum(uint32_t value);
  void AlignByte();
  void AppendBits(const uint8_t* bytes, uint64_t nbits);   // bit-granular append of another stream
  uint64_t BitCount() const { return (uint64_t)bytes_.size() * 8 + nbits_; }
  std::vector<uint8_t> Finish();   // pads to a byte boundary
 private:
  std::vector<uint8_t> bytes_;
  uint64_t acc_ = 0;
  int nbits_ = 0;
};
inline BitWriter::Dist WV(uint32_t v) { return {-1, v}; }
inline BitWriter::Dist WB(int n, uint32_t o = 0) { return {n, o}; }

struct EncToken { uint32_t ctx, value; };

// Hybrid-uint split used by every stream this encoder writes: split_exponent 4, msb_in_token 2, lsb_in_token 0.
constexpr uint32_t kEncSplitExp = 4, kEncMsb = 2, kEncLsb = 0;
constexpr uint32_t kEncAlphabet = 128;   // tokens of 32-bit values stay below 16 + 28 * 4
inline void HybridEncode(uint32_t v, uint32_t* tok, uint32_t* nbits, uint32_t* bits) {
  if (v < (1u << kEncSplitExp)) { *tok = v; *nbits = 0; *bits = 0; return; }
  const uint32_t n = 31 - (uint32_t)__builtin_clz(v), m = v - (1u << n);
  *tok = (1u << kEncSplitExp) + ((n - kEncSplitExp) << (kEncMsb + kEncLsb)) + ((m >> (n - kEncMsb)) << kEncLsb) + (m & ((1u << kEncLsb) - 1));
  *nbits = n - kEncMsb - kEncLsb;
  *bits = (m >> kEncLsb) & ((1u << *nbits) - 1);
}

// An ANS code ready for encoding: context map + per cluster the normalised frequencies and the map
// (symbol, offset) -> 12-bit slot that inverts the decoder's alias table.
struct EncCode {
  uint32_t log_alpha = 5;
  uint32_t num_clusters = 1;
  std::vector<uint8_t> ctx_map;
  std::vector<uint16_t> freq;    // [cluster * kEncAlphabet + symbol]
  std::vector<uint16_t> start;   // [cluster * kEncAlphabet + symbol]: cumulative frequency below the symbol
  std::vector<uint16_t> rmap;    // [cluster * 4096 + start + offset] -> slot
};

// Clusters the per-context histograms hist[ctx * kEncAlphabet + symbol] into at most max_clusters ANS distributions,
// writes the entropy-code header (context map, log_alpha, uint configs, distributions) and fills `out`.
// pinned_zero: contexts whose tokens are never emitted because every one of them is the symbol 0 (constant channels);
// they share one single-symbol cluster of their own.
void BuildAndWriteCode(const uint32_t* hist, size_t num_ctx, int max_clusters, const std::vector<uint8_t>& pinned_zero, BitWriter& bw,
                       EncCode& out);
// ANS-codes `tokens` (host side: MA tree, context maps).
void WriteTokensHost(const std::vector<EncToken>& tokens, const EncCode& code, BitWriter& bw);

struct EncTreeNode { int property; int32_t splitval; int pred; int32_t offset; uint32_t multiplier; };   // property < 0: leaf
// Serialises an MA tree given in decode (breadth-first) order.
void WriteTree(const std::vector<EncTreeNode>& nodes, BitWriter& bw);

struct EncImageInfo {
  uint32_t xsize = 0, ysize = 0;
  bool gray = false, alpha = false;
  bool xyb = true;       // false: lossless (original colour space, Encoder/JxlEncoder.cpp:214)
  const uint8_t* icc = nullptr;   // embedded ICC profile (JxlEncoderSetICCProfile, Encoder/JxlEncoder.cpp:258-268) instead of the enum encoding
  size_t icc_size = 0;
};
struct EncFrameInfo {
  uint32_t encoding = 0;           // 0 VarDCT, 1 Modular
  uint32_t group_size_shift = 1;
  uint32_t x_qm_scale = 3, b_qm_scale = 2;
  bool gab = true;
  uint32_t epf_iters = 1;
  uint64_t flags = 0;
};
void WriteCodestreamHeaders(const EncImageInfo& im, BitWriter& bw);   // signature, SizeHeader, ImageMetadata; byte-aligned at the end
void WriteFrameHeader(const EncImageInfo& im, const EncFrameInfo& f, BitWriter& bw);
void WriteToc(const std::vector<uint32_t>& sizes, BitWriter& bw);
// ISO BMFF container (18181-2): signature, ftyp, Exif, xml, jxlc.  exif already carries its 4-byte TIFF offset prefix.
std::vector<uint8_t> WriteContainer(const std::vector<uint8_t>& codestream, const uint8_t* exif, size_t exif_size, const uint8_t* xmp,
                                    size_t xmp_size);

// exported by host_parse.cc: the decoder's alias-table construction (the encoder inverts exactly this table)
void BuildAliasTable(const std::vector<int>& counts, uint32_t log_alpha, uint64_t* out);

}  // namespace jxlhip
