// ORACLE — test infrastructure only (see jxo_common.h header).
// Public C++ surface of the CPU restatement: whole-file decode and encode.
#pragma once
#include "jxo_common.h"
#include "jxo_headers.h"
#include "jxo_vardct.h"
#include <functional>

namespace jxo {

// Intermediate results kept for per-stage parity tests against the HIP path.
// Plane layouts are the ones documented in DESIGN.md ("Data layout in HBM").
struct StageDump {
  int w8 = 0, h8 = 0;                      // frame size in 8x8 blocks
  int wp = 0, hp = 0;                      // frame size padded to blocks (8*w8, 8*h8)
  std::vector<int32_t> lf_quant[3];        // X,Y,B quantised LF, w8*h8
  std::vector<float> lf[3];                // dequantised (+CfL, +smoothing) LF, w8*h8
  std::vector<uint8_t> strategy;           // per 8x8 cell: strategy id | 0x80 if first (top-left) cell
  std::vector<int32_t> raw_quant;          // per cell
  std::vector<uint8_t> sharpness;          // per cell
  std::vector<int8_t> ytox, ytob;          // per 64x64 tile, ceil(w8/8) x ceil(h8/8)
  std::vector<int32_t> qcoef[3];           // quantised HF coefficients, footprint layout, wp*hp
  std::vector<float> xyb_idct[3];          // after IDCT, wp*hp (cropped to frame size: w*h)
  std::vector<float> xyb_filtered[3];      // after Gaborish + EPF, w*h
  std::vector<int32_t> alpha;              // decoded alpha (w*h) if present
};

struct DecodeResult {
  ImageMetadata meta;
  FrameHeader frame;
  ContainerInfo boxes;
  int num_channels = 0;         // 1,2,3,4 (gray[+a], rgb[+a])
  int out_w = 0, out_h = 0;     // size of `pixels`: the frame size with the header's orientation applied (5..8 swap the sides)
  int bits_out = 8;             // 8: `pixels` holds u8 samples; 16: little-endian u16 samples (streams of more than 8 bits per sample)
  bool out_float = false;       // float-sample streams: bits_out 16 = binary16, 32 = binary32 (little-endian bit patterns)
  std::vector<uint8_t> pixels;  // interleaved, tight rows; CMYK (black extra channel): C, M, Y, K [, A] with 0 = no ink (the
                                // reference inverts the stored samples for its host, Decoder/JxlDecoder.cpp:159-215)
  std::vector<uint8_t> icc;     // embedded ICC profile (what the reference hands to setIccProfile for original-profile streams)
  bool cmyk = false;
  StageDump dump;
};

struct DecodeOptions {
  bool want_dump = false;
  int num_threads = 1;
};

void DecodeJxl(const uint8_t* data, size_t size, const DecodeOptions& opt, DecodeResult& out);

struct EncodeParams {
  float distance = 1.0f;
  bool lossless = false;
  int effort = 7;
  // 0: activity heuristic (default), 1: DCT8 only, 2: seeded pseudo-random mix of every supported
  // strategy (parity-test coverage), 3: every block of one strategy (`fixed_strategy`) where it fits
  int strategy_mode = 0;
  int fixed_strategy = 0;
  uint32_t seed = 1;
  int epf_iters = -1;     // -1: by distance
  bool gaborish = true;
  bool container = true;
  bool adaptive_lf_smoothing = true;
  int lossless_predictor = 6;   // leaf predictor for lossless modular (6 = weighted)
  bool lossless_squeeze = false;
  int lossless_tree = 0;        // 0: contexts from the weighted predictor's error (property 15); 1: local-gradient contexts (W-NW, NW-N)
  int num_threads = 1;
  int orientation = 1;          // EXIF orientation written to the header (1..8); the pixels handed in are the STORED image
  int bits = 8;                 // bits per sample signalled in the header (8..16); above 8 the input samples are uint16
  // colour encoding signalled in the header; the pixels handed in are ALREADY in that space (lossy frames are converted to XYB from
  // it).  0: sRGB; 1: Display P3 (sRGB transfer); 2: BT.709 transfer, sRGB primaries; 3: BT.2100 primaries, linear;
  // 4: BT.2100 primaries, PQ (intensity target 10000); 5: linear sRGB; 6: HLG written to the header only (refusal tests);
  // 7: Adobe RGB (custom primaries + gamma); 8: DCI-P3 (DCI white, gamma 2.6); 9: sRGB primaries, D50 white, linear
  int colour = 0;
  int float_samples = 0;        // 0: integer samples; 16 / 32: binary16 / binary32 samples (input arrays of that float type)
  std::vector<uint8_t> icc;     // embedded ICC profile instead of the enumerated colour encoding
  int num_passes = 1;                 // lossy: 2 or 3 = progressive passes (coefficient bits split by the shifts 1 / 2, 1)
  bool palette = false;               // lossless: colour (+ alpha) channels through a Palette transform when the image has at most 1024 colours
  bool lf_contexts = false;           // lossy: a block-context map with LF thresholds (quartiles of the quantised LF) and two quant-field thresholds
  bool custom_orders = false;         // lossy: coefficient orders sorted by how often each position is non-zero (per order bucket and channel)
  bool custom_quant_tables = false;   // lossy: every dequantisation table written explicitly (parameters scaled per table by `seed`)
  int animation_frames = 1;     // > 1: an animation; frame k > 0 shows the picture rotated by 180 degrees / inverted (any decoder
                                // that returns something other than the first frame is caught)
  bool mislabel_afv = false;    // lossy, REFUSAL TESTS ONLY: some 8x8 DCT blocks are written to the block-metadata stream as AFV0..AFV3 (the
                                // coefficients stay those of the 8x8 DCT, so the stream is not a meaningful picture): a decoder without AFV must refuse it
  bool premultiplied_alpha = false;   // the alpha channel is signalled as associated; the colour samples handed in are ALREADY premultiplied
  bool cmyk = false;            // lossless only: nch 4 / 5 = C, M, Y, K [, A] as STORED (0 = full ink); K goes to a black extra channel
};

// rgba: interleaved RGBA8 (or RGB8 / Gray8 / GrayA8 according to nch), tight rows.
// Token counts of the last VarDCT frame this thread encoded: LF coefficients, HF metadata, HF coefficients, alpha.
void SetLastEncodeTokenCounts(const uint64_t n[4]);
void GetLastEncodeTokenCounts(uint64_t n[4]);
std::vector<uint8_t> EncodeJxl(const uint8_t* px, uint32_t w, uint32_t h, int nch, const EncodeParams& p,
                               const uint8_t* exif = nullptr, size_t exif_size = 0, const uint8_t* xmp = nullptr,
                               size_t xmp_size = 0);

void ParallelFor(int n, int num_threads, const std::function<void(int)>& fn);

}  // namespace jxo
