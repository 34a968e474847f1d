// ORACLE — test infrastructure only (see jxo_common.h header).
// Codestream headers of ISO/IEC 18181-1: SizeHeader, ImageMetadata, FrameHeader, TOC,
// and the ISO/IEC 18181-2 box container.  The reference consumes these through
// JxlDecoderGetBasicInfo / JxlDecoderGetFrameHeader (Decoder/JxlDecoder.cpp:463,263) and
// produces them through JxlEncoderSetBasicInfo / UseBoxes (Encoder/JxlEncoder.cpp:247,201).
#pragma once
#include "jxo_common.h"

namespace jxo {

struct ExtraChannelInfo {
  uint32_t type = 0;  // 0 alpha, 4 black
  uint32_t bits = 8, exp_bits = 0;
  uint32_t dim_shift = 0;
  std::string name;
  bool alpha_associated = false;
};

struct ColorEncoding {
  bool all_default = true;
  bool want_icc = false;
  uint32_t color_space = 0;   // 0 RGB, 1 Gray, 2 XYB, 3 Unknown
  uint32_t white_point = 1;   // 1 D65, 2 custom, 10 E, 11 DCI
  uint32_t primaries = 1;     // 1 sRGB, 2 custom, 9 2100, 11 P3
  bool have_gamma = false;
  uint32_t gamma = 0;
  uint32_t tf = 13;           // 1 709, 2 unknown, 8 linear, 13 sRGB, 16 PQ, 17 DCI, 18 HLG
  uint32_t rendering_intent = 1;  // 0 perceptual, 1 relative
  int32_t custom_xy[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};  // white, r, g, b
};

struct ImageMetadata {
  uint32_t xsize = 0, ysize = 0;
  uint32_t orientation = 1;
  uint32_t bits = 8, exp_bits = 0;
  bool modular_16bit = true;
  std::vector<ExtraChannelInfo> ec;
  bool xyb_encoded = true;
  ColorEncoding color;
  float intensity_target = 255.f, min_nits = 0.f, linear_below = 0.f;
  bool relative_to_max_display = false;
  bool have_animation = false, have_timecodes = false;
  bool have_preview = false;
  bool default_transform = true;
  float opsin_inverse[9];
  float opsin_bias[3];
  float quant_bias[4];
  ImageMetadata() { SetDefaultTransformData(); }
  void SetDefaultTransformData();   // what a stream with default_transform = true implies (the encoder's quantiser reads quant_bias too)
  int alpha_index() const {
    for (size_t i = 0; i < ec.size(); i++) if (ec[i].type == 0) return (int)i;
    return -1;
  }
  int num_color_channels() const { return color.color_space == 1 ? 1 : 3; }
};

struct LoopFilter {
  bool gab = true;
  float gab_w1[3] = {0.115169525f, 0.115169525f, 0.115169525f};
  float gab_w2[3] = {0.061248592f, 0.061248592f, 0.061248592f};
  uint32_t epf_iters = 2;
  float epf_sharp_lut[8] = {0, 1.f / 7, 2.f / 7, 3.f / 7, 4.f / 7, 5.f / 7, 6.f / 7, 1.f};
  float epf_channel_scale[3] = {40.0f, 5.0f, 3.5f};
  float epf_quant_mul = 0.46f, epf_pass0_sigma_scale = 0.9f, epf_pass2_sigma_scale = 6.5f, epf_border_sad_mul = 2.0f / 3;
  float epf_sigma_for_modular = 1.0f;
};

struct FrameHeader {
  enum { kRegular = 0, kLF = 1, kReferenceOnly = 2, kSkipProgressive = 3 };
  enum { kNoise = 1, kPatches = 2, kSplines = 16, kUseLfFrame = 32, kSkipAdaptiveLfSmoothing = 128 };
  uint32_t frame_type = 0;
  uint32_t encoding = 0;  // 0 VarDCT, 1 Modular
  uint64_t flags = 0;
  bool do_ycbcr = false;
  uint32_t upsampling = 1;
  std::vector<uint32_t> ec_upsampling;
  uint32_t group_size_shift = 1;
  uint32_t x_qm_scale = 3, b_qm_scale = 2;
  uint32_t num_passes = 1;
  uint32_t pass_shift[8] = {0};
  uint32_t lf_level = 0;
  bool have_crop = false;
  int32_t x0 = 0, y0 = 0;
  uint32_t width = 0, height = 0;
  uint32_t blend_mode = 0;
  uint32_t duration = 0;
  bool is_last = true;
  uint32_t save_as_reference = 0;
  bool save_before_ct = false;
  std::string name;
  LoopFilter lf;
  // derived
  uint32_t xsize = 0, ysize = 0;
  uint32_t group_dim = 256;
  uint32_t xsize_blocks = 0, ysize_blocks = 0;
  uint32_t xsize_groups = 0, ysize_groups = 0, num_groups = 0;
  uint32_t xsize_lf_groups = 0, ysize_lf_groups = 0, num_lf_groups = 0;
  void Derive(const ImageMetadata& m);
  size_t NumTocEntries() const {
    return (num_groups == 1 && num_passes == 1) ? 1 : 2 + num_lf_groups + (size_t)num_groups * num_passes;
  }
};

void ReadSizeHeader(BitReader& br, uint32_t* xsize, uint32_t* ysize);
void WriteSizeHeader(BitWriter& bw, uint32_t xsize, uint32_t ysize);
void ReadImageMetadata(BitReader& br, ImageMetadata& m);   // after the size header; includes transform data
void WriteImageMetadata(BitWriter& bw, const ImageMetadata& m);
void ReadFrameHeader(BitReader& br, const ImageMetadata& m, FrameHeader& f);
void WriteFrameHeader(BitWriter& bw, const ImageMetadata& m, const FrameHeader& f);

struct Toc {
  std::vector<uint32_t> sizes;        // physical order
  std::vector<uint64_t> offsets;      // logical section i -> byte offset from end of TOC
  std::vector<uint32_t> logical_size; // logical section i -> size
};
void ReadToc(BitReader& br, size_t num_entries, Toc& toc);
void WriteToc(BitWriter& bw, const std::vector<uint32_t>& sizes);
struct EntropyReader;
void ReadPermutation(BitReader& br, EntropyReader& rd, size_t skip, size_t size, std::vector<uint32_t>& perm);
// tokens of a permutation whose first `skip` entries are fixed (Lehmer code; contexts as ReadPermutation reads them)
struct Token;
void TokenizePermutation(const std::vector<uint32_t>& perm, size_t skip, std::vector<Token>& out);

// ------------------------------------------------------------------ container
struct ContainerInfo {
  bool is_container = false;
  std::vector<uint8_t> codestream;  // concatenated jxlc / jxlp payloads (or the raw input)
  std::vector<uint8_t> exif;        // first Exif box payload (incl. 4-byte offset prefix)
  std::vector<std::vector<uint8_t>> xml;  // every "xml " box
  bool has_brob = false;
};
// 0: not JXL, 1: bare codestream, 2: container
int SignatureCheck(const uint8_t* data, size_t size);
void ParseContainer(const uint8_t* data, size_t size, ContainerInfo& out);
std::vector<uint8_t> WriteContainer(const std::vector<uint8_t>& codestream, const uint8_t* exif, size_t exif_size,
                                    const uint8_t* xmp, size_t xmp_size);

}  // namespace jxo
