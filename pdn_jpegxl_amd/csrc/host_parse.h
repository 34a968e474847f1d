// Host-side parsing of everything that is small and serial in a JPEG XL file: container boxes,
// image/frame headers, TOC, LfGlobal and HfGlobal (entropy-code headers -> alias tables, MA tree,
// dequantisation weights, coefficient orders).  The bulk data (LF groups, pass groups) is never
// touched on the host: it is decoded by the HIP kernels straight from HBM.
//
// This is the product-side counterpart of the libjxl calls the reference makes in
// Decoder/JxlDecoder.cpp:412-793 (ReadImageInfoAndMetadata) and :217-410 (ReadFrameData).
#pragma once
#include <cstdint>
#include <cstddef>
#include <deque>
#include <stdexcept>
#include <string>
#include <vector>
#include "dev_types.h"

namespace jxlhip {

// status codes mirror DecoderStatus in include/jxlfiletypeio.h
struct ParseError : std::runtime_error {
  int status;
  ParseError(int st, const std::string& m) : std::runtime_error(m), status(st) {}
};

struct HybridCfg { uint32_t split = 4, msb = 2, lsb = 0; };

struct HostCode {
  bool lz77 = false;
  uint32_t lz_min_symbol = 0, lz_min_length = 0;
  HybridCfg lz_len;
  std::vector<uint8_t> ctx_map;
  uint32_t num_hist = 1;
  bool use_prefix = false;
  uint32_t log_alpha = 8;
  std::vector<HybridCfg> cfg;
  std::vector<uint64_t> alias;   // [hist << log_alpha | i], packed as DevCode::alias
  // prefix codes (host-side decoding only)
  struct Prefix { uint16_t count[16]; std::vector<uint16_t> sorted; int single; };
  std::vector<Prefix> prefix;
};

struct ColorInfo {
  bool all_default = true, want_icc = false;
  uint32_t color_space = 0, white_point = 1, primaries = 1, tf = 13, rendering_intent = 1;
  bool have_gamma = false;
  uint32_t gamma = 0;
  double white_xy[2] = {0.3127, 0.3290};   // custom white point / primaries as signalled (white_point == 2 / primaries == 2)
  double prim_xy[3][2] = {{0.64, 0.33}, {0.30, 0.60}, {0.15, 0.06}};
};

struct ExtraChannel { uint32_t type = 0, bits = 8, exp_bits = 0, dim_shift = 0; bool alpha_associated = false; };

// What the colour encoding of a stream means for the decode (reference: SetProfileFromColorEncoding, Decoder/JxlDecoder.cpp:36-108).
struct ColorPlan {
  int known_profile = -1;   // KnownColorProfile the host is told, -1: none of the eight named ones
  bool report_icc = false;  // the host is handed the embedded ICC profile (setIccProfile) and the samples are in that profile's space
  int transfer = 1;         // 0 linear, 1 sRGB, 2 BT.709, 3 PQ, 5 the tables of `trc_lut` (an evaluated ICC profile)
  std::vector<float> trc_lut;   // 3 x kIccInvLut: sqrt(linear) -> encoded
  std::vector<uint8_t> icc_out; // report_icc without an embedded profile: the profile synthesised for the enumerated encoding
  float from_srgb[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};   // linear sRGB -> linear RGB of the image's primaries (row-major)
};
struct ParsedFrame;
ColorPlan PlanColor(const ParsedFrame& f);

struct ParsedFrame {
  // ---- container
  bool is_container = false;
  std::vector<uint8_t> cs_copy;   // only when the codestream is split over jxlp boxes
  const uint8_t* cs = nullptr;    // codestream bytes (inside the caller's buffer or cs_copy)
  size_t cs_size = 0;
  size_t cs_file_offset = 0;      // offset of cs inside the file when contiguous (for device-resident files)
  bool cs_contiguous = true;
  const uint8_t* exif = nullptr; size_t exif_size = 0;
  std::vector<std::pair<const uint8_t*, size_t>> xml;
  // the boxes the host is told about, in file order (the first Exif box, every xml box; payloads may be empty)
  struct MetaBox { bool is_exif; const uint8_t* data; size_t size; };
  std::vector<MetaBox> meta_in_order;
  bool have_exif = false;
  std::deque<std::vector<uint8_t>> owned_boxes;   // decompressed `brob` payloads (exif / xml may point into these)
  // ---- image header
  uint32_t xsize = 0, ysize = 0, orientation = 1;
  uint32_t bits = 8, exp_bits = 0;
  std::vector<ExtraChannel> ec;
  bool xyb_encoded = true;
  ColorInfo color;
  std::vector<uint8_t> icc;       // the embedded ICC profile (want_icc), decoded
  float intensity_target = 255.f;
  bool have_animation = false, have_timecodes = false;
  float opsin_inv[9], opsin_bias[3], qbias[4];
  int alpha_index = -1, black_index = -1;
  int ncolor = 3;
  // ---- frame header
  uint32_t frame_type = 0, encoding = 0;
  uint64_t flags = 0;
  uint32_t group_size_shift = 1, x_qm_scale = 3, b_qm_scale = 2, num_passes = 1;
  bool is_last = true;
  std::string name;
  bool gab = true;
  float gab_w1[3], gab_w2[3];
  uint32_t epf_iters = 2;
  float epf_sharp_lut[8], epf_channel_scale[3];
  float epf_quant_mul, epf_pass0_sigma_scale, epf_pass2_sigma_scale, epf_border_sad_mul;
  // derived geometry
  uint32_t w8 = 0, h8 = 0, xg = 0, yg = 0, ng = 0, xlf = 0, ylf = 0, nlf = 0, group_dim = 256;
  // ---- TOC (logical order), offsets relative to cs
  std::vector<uint64_t> sec_off;
  std::vector<uint32_t> sec_size;
  // ---- LfGlobal
  float m_lf[3];
  uint32_t global_scale = 1, quant_lf = 16;
  std::vector<uint8_t> block_ctx_map;
  std::vector<uint32_t> qf_thr;
  std::vector<int32_t> lf_thr[3];   // LF thresholds of the block-context map (X, Y, B)
  uint32_t num_block_ctx = 15;
  uint32_t color_factor = 84;
  float base_x = 0.f, base_b = 1.f;
  int ytox_lf = 0, ytob_lf = 0;
  bool has_global_tree = false;
  std::vector<DevTreeNode> tree;
  bool tree_uses_wp = false, tree_uses_ref = false;
  bool tree_row_static = true;   // every inner node tests channel / stream / row only and every leaf predicts Zero / W / N / Gradient
  HostCode mcode;
  bool global_modular_has_channels = false;
  // Modular frames (encoding 1): the GlobalModular image header
  struct ModSqueeze { bool horizontal = false, in_place = false; uint32_t begin_c = 0, num_c = 0; };
  struct ModTransform { uint32_t id = 0, begin_c = 0, rct_type = 0, num_c = 0, nb_colors = 0, nb_deltas = 0, predictor = 0; std::vector<ModSqueeze> squeezes; };
  std::vector<ModTransform> mod_transforms;   // RCT, Palette (plain: no delta entries) and Squeeze
  uint64_t mod_data_bits = 0;                 // codestream bit position of the GlobalModular channel data (stream 0)
  // Channel layout after the transforms (what the streams code) and the inverse operations in execution order.
  // Every channel instance owns a plane; planes 0 .. nch-1 are the image channels the inverse ends in.
  struct ModPlane { int32_t w = 0, h = 0; };
  struct ModChan { int32_t w = 0, h = 0, hshift = 0, vshift = 0, plane = 0; };
  // kind 0: RCT on planes a,b,c; 1 / 2: horizontal / vertical unsqueeze (avg a, residual b) -> plane c;
  // 3: palette lookup: palette plane a (type colours wide, one row per output), index plane b -> planes out[0 .. nout)
  struct ModOp { int32_t kind = 0; int32_t a = 0, b = 0, c = 0, type = 0; int32_t out[4] = {0, 0, 0, 0}; int32_t nout = 0; };
  std::vector<ModPlane> mod_planes;
  std::vector<ModChan> mod_coded;
  std::vector<ModOp> mod_ops;
  uint32_t mod_first_group_channel = 0;       // coded channels before this one live in the GlobalModular stream
  bool mod_has_squeeze = false;
  bool mod_has_palette = false;
  bool single = false;                 // one TOC entry: all sections share one bit stream (frames that fit one group)
  uint64_t after_lf_global_bits = 0;   // codestream bit position right after the host-parsed part of LfGlobal
  uint64_t hf_start_bits = 0;          // single-section frames: bit position right after HfGlobal (set by the LF pre-pass)
  // ---- HfGlobal
  bool dq_default = true;
  std::vector<float> custom_dq[kNumQuantTables];   // !dq_default: 3 * n multipliers (1 / weight) per table, stored layout
  uint32_t num_presets = 1;
  std::vector<uint16_t> custom_order[kNumOrders][3];  // empty => natural
  HostCode acode;
  // Progressive frames: the coefficients arrive in num_passes instalments, pass p shifted left by pass_shift[p] (the last by 0);
  // pass 0 uses custom_order / acode above, the later passes their own orders and codes
  uint32_t pass_shift[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  struct PassCodes {
    std::vector<uint16_t> custom_order[kNumOrders][3];
    HostCode acode;
  };
  std::vector<PassCodes> extra_passes;
};

// Throws ParseError.  headers_only: stop after the frame header + TOC (jxlhip_peek / pass 1 of LoadImage).
void ParseFile(const uint8_t* data, size_t size, bool headers_only, ParsedFrame& out);
// Single-section frames: parses HfGlobal at `bit_pos` (where the GPU finished the LF group); returns the bit position after it.
uint64_t ParseHfGlobalAt(ParsedFrame& f, uint64_t bit_pos);

// Static tables shared by every image (computed once on the host, uploaded at decoder creation).
struct StaticTables {
  std::vector<uint16_t> natural_order[kNumOrders];
  std::vector<float> dq[kNumQuantTables];     // default library tables, 3*n each
  std::vector<float> basis[6];                // N = 8,16,...,256: B[k*N+n]
  std::vector<float> llf_scale;               // [6][32]: resample scale per (log2 c, k)
};
const StaticTables& GetStaticTables();

extern const uint8_t kCoveredX[kNumStrategies];
extern const uint8_t kCoveredY[kNumStrategies];
extern const uint8_t kStrategyOrderBucket[kNumStrategies];
extern const uint8_t kStrategyQuantTable[kNumStrategies];

}  // namespace jxlhip
