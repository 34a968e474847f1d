// Device-side description of one image being encoded (encode_kernels.hip / encoder.cc).
#pragma once
#include <stdint.h>
#include "dev_types.h"

namespace jxlhip {

struct DevToken { uint32_t ctx, value; };   // after the reverse ANS pass `ctx` holds flushed << 16 | flush bits

// leaf (= context) ids of the encoder's fixed MA tree, in decode order (encoder.cc: MakeEncoderTree)
enum EncLeaf : uint32_t {
  kLeafAlpha = 0, kLeafSharp = 1, kLeafAlphaGlobal = 2, kLeafCfl = 3, kLeafLfB = 4, kLeafQf = 5, kLeafStrategy = 6, kLeafLfX = 7,
  kLeafLfY = 8, kNumEncLeaves = 9
};
constexpr uint32_t kEncSyms = 128;            // token alphabet of the fixed hybrid-uint config (4, 2, 0)
constexpr uint32_t kAcContexts = 495 * 15;    // one preset, default block-context map
constexpr uint32_t kLfTokCap = 3 * 65536;     // tokens per LF group: LF coefficients
constexpr uint32_t kMetaTokCap = 2 * 65536;   // ... and the two rows of the block info (strategies, quant field)
constexpr uint32_t kAcTokCap = 3 * 65 * 1024; // tokens per group: 1 + 64 per (block, channel)
constexpr uint32_t kAlphaTokCap = 65536;
constexpr uint32_t kLlTokCap = 4 * 65536;     // lossless: up to four channels per group

struct EncCodeDev {
  const uint8_t* ctx_map;
  const uint16_t* freq;    // [cluster * kEncSyms + symbol]
  const uint16_t* start;
  const uint16_t* rmap;    // [cluster * 4096 + start + offset]
  uint32_t num_clusters, num_ctx;
};

struct EncImage {
  int32_t w, h, w8, h8, wp, hp;
  int32_t xg, yg, ng, xlf, ylf, nlf;
  int32_t gray, has_alpha, gab, lossless;
  // source (host layout BitmapData: BGRA8, `stride` bytes per row)
  const uint8_t* bgra;
  int32_t stride, pad0;
  uint32_t* flags;          // [0] some pixel is not gray, [1] some pixel has alpha < 255
  // documents with an (evaluated, matrix / TRC) ICC profile: 3 x 256 samples -> linear of the profile, then 3x3 -> linear sRGB
  const float* icc_lin;     // nullptr: the samples are sRGB
  float icc_to_srgb[9];
  // planes
  float* xyb[3];            // w*h
  float* pad[3];            // wp*hp: (inverse-Gaborish sharpened) XYB, edge-replicated to whole 8x8 cells
  int32_t* alpha_px;        // w*h
  int32_t* lfq[3];          // w8*h8 quantised LF (X, Y, B)
  int32_t* rawq;            // w8*h8 raw quant field (1..256)
  // varblocks: `squares` = 0: 8x8 DCTs only (the fast effort); 1: 8x8 / 16x16 / 32x32; 2: also 64x64 and the rectangular
  // 16x8 ... 64x32 shapes (the default effort)
  int32_t squares, pad1;
  float* act;               // w8*h8 activity of Y per cell (standard deviation)
  uint8_t* strat;           // w8*h8: strategy code of the varblock covering the cell | 0x80 on its first cell
  int32_t* qs[3];           // w8*h8*64 quantised coefficients: scan position k of a varblock at its covered cell (k >> 6, row-major) * 64 + (k & 63)
  uint8_t* nz[3];           // per cell: the non-zero context value of the varblock covering it, (count + covered - 1) >> log2(covered)
  uint16_t* nzc[3];         // per first cell: number of non-zero HF coefficients of the varblock
  uint16_t* last[3];        // per first cell: scan position of the last non-zero coefficient (0: none)
  // quantiser
  float inv_mul_lf[3];      // 1 / (m_lf * inv_global_scale / quant_lf)
  float mul_lf_y;           // LF dequant step of Y (chroma-from-luma of the LF uses the dequantised Y)
  float inv_gs;             // 65536 / global_scale
  float x_dm, b_dm;         // 0.8 ^ (x_qm_scale - 2), 0.8 ^ (b_qm_scale - 2)
  float qbias1, qbias3;     // quantisation bias of |q| == 1 (Y) and the 1/q term
  float gab_w[3][3];
  // per order bucket (13): inverse natural order (stored index -> scan position); per quant table (17): 3 * n dequantisation
  // multipliers in the stored layout; per transform length N = 8, 16, 32, 64 (index 0..3): the basis B[k * N + n] and the same divided
  // by N; per c = 1, 2, 4, 8: the c x c basis (LF values of a varblock from its lowest coefficients) and the resample scales
  const uint16_t* scan_of[13];
  const float* dq[17];
  const float* basis[4];
  const float* basis_div[4];
  const float* bsmall[4];
  const float* rs;          // [(lcy * 4 + lcx) * 64 + ky * 8 + kx]: scale of coefficient (ky, kx) of the lowest cy x cx
  // tokens
  DevToken* tok_lf;         // [nlf][kLfTokCap]
  DevToken* tok_meta;       // [nlf][kMetaTokCap]
  DevToken* tok_ac;         // [ng][kAcTokCap]
  DevToken* tok_alpha;      // [ng][kAlphaTokCap]
  uint32_t* n_ac;           // [ng]
  uint32_t* n_meta;         // [nlf] tokens of the block-info stream
  uint32_t* hist_mod;       // [kNumEncLeaves][kEncSyms]
  uint32_t* hist_ac;        // [kAcContexts][kEncSyms]
  // lossless (Modular) frames: whole-image integer channels, tokens per group
  int32_t* ll_plane[4];
  int32_t ll_nch, ll_rct;   // ll_rct: RGB is coded as YCoCg-R (reversible colour transform 6)
  DevToken* tok_ll;         // [ng][4 * 65536]
  // entropy coding
  EncCodeDev mcode, acode;
  uint8_t* sec_bytes;       // section s at s * sec_cap
  uint32_t* stream_state;   // final rANS state per token stream (lossy frames: 2 per LF group, 2 per group, 1 global alpha)
  uint64_t* sec_bits;       // bits written per section
  uint64_t sec_cap;
};

}  // namespace jxlhip
