// Structures shared between the host-side frame parser and the HIP kernels.
// Everything a kernel needs about one image lives in a DevImage record in HBM; batches are
// arrays of DevImage and every kernel takes (images, task table) so that one launch covers
// all images of a batch.
#pragma once
#include <stdint.h>

namespace jxlhip {

constexpr int kNumStrategies = 27;
constexpr int kNumOrders = 13;
constexpr int kNumQuantTables = 17;
constexpr int kGroupDim = 256;
constexpr int kGroupBlocks = 32;        // 8x8 cells per group side
constexpr int kLfGroupBlocks = 256;     // 8x8 cells per LF group side
constexpr int kMaxClustersLds = 128;
// HF coefficients leave the entropy stage as a sparse list per 256x256 group: one 32-bit entry per NON-ZERO coefficient, in decode
// order, entry = scan position k (low 16 bits) | quantised value (int16, high 16 bits).  A group holds at most 3 * 65536 of them.
constexpr uint32_t kGroupEntriesCap = 3u * 65536u;

// error bits written by kernels into DevImage::status[0]
enum DevError : uint32_t {
  kErrNone = 0,
  kErrBitstream = 1u << 0,        // overrun / invalid symbol / final ANS state mismatch
  kErrUnsupportedHeader = 1u << 1,  // modular group header other than {global tree, default wp, no transforms}
  kErrUnsupportedTree = 1u << 2,    // weighted predictor / reference-channel properties on the GPU path
  kErrBlockLayout = 1u << 3,        // invalid varblock placement
  kErrRange = 1u << 4,              // value out of range (sharpness, quant, cfl)
  kErrUnsupportedTransform = 1u << 5,   // AFV0..AFV3 varblocks (strategy ids 14..17): their 16x16 basis is not built; never decoded as anything else
};

struct alignas(8) U32x2 { uint32_t x, y; };   // 8 bytes, 8-byte aligned loads / stores on the device

struct DevTreeNode {    // 16 bytes
  int32_t property;     // -1: leaf
  int32_t splitval;     // leaf: offset
  uint32_t a;           // inner: child for (property > splitval); leaf: predictor | ctx << 8
  uint32_t b;           // inner: other child; leaf: multiplier
};

struct DevCode {
  const uint8_t* ctx_map;
  const uint32_t* cfg;      // per cluster: split_exponent | msb << 4 | lsb << 8 | degenerate << 12 | symbol << 16
  const uint64_t* alias;    // [cluster << log_alpha | i]: low32 = cutoff | right << 8 | freq0 << 16, high32 = offsets1 | (freq1^freq0) << 16
  uint32_t num_ctx;
  uint32_t num_clusters;
  uint32_t log_alpha;
  uint32_t slow;            // bit 0: prefix codes (log_alpha 15, no alias tables), bit 1: LZ77 - both take the general symbol reader
  // prefix codes: per cluster 16 counts of codes per length (canonical code), the symbols sorted by (length, value)
  const uint16_t* pfx_count;   // [cluster * 16 + length]
  const uint16_t* pfx_sorted;  // [pfx_off[cluster] + rank]
  const uint32_t* pfx_off;
  // LZ77: symbols >= lz_min_symbol start a copy; its length uses lz_len_cfg (packed like cfg), its distance the cluster of the last context
  uint32_t lz_min_symbol, lz_min_length, lz_len_cfg, lz_dist_cluster;
  // ANS codes of small launches: per cluster 4096 entries, one per state residue (freq - 1 | offset << 12 | symbol << 24), read by
  // one-section wavefronts through the scalar cache; null when not built
  const uint32_t* direct;
};

// Outcome of phase A (one lane per section) for one Modular channel; phase B (a wavefront per channel) finishes it.
enum ChanKind : int32_t {
  kChanFinal = 0,   // samples are final
  kChanResid = 1,   // residuals stored in place; the row predictors (Zero / W / N / Gradient) are still to be applied
  kChanConst = 2,   // every sample equals `value`; nothing was stored
};
struct ChanDesc {
  int32_t kind;
  int32_t value;
  int32_t pad0, pad1;   // pad0: phase A writes 0; alpha_finish_gradient_kernel sets 1 on the groups it finished
};

// One coded channel of a Modular frame (sizes after Squeeze; shifts tell which sections code it).
struct ModChanDev {
  int32_t w, h, hshift, vshift;
  int32_t* plane;
};

struct DevImage {
  // geometry
  int32_t w, h, w8, h8, wp, hp, wt, ht;
  int32_t xg, yg, ng, xlf, ylf, nlf;
  int32_t ncolor, has_alpha, nch_out, to_srgb;   // to_srgb: transfer function of the output, 0 linear, 1 sRGB, 2 BT.709, 3 PQ, 5 tables
  float pq_scale, pad_color;                     // intensity target / 10000 (PQ)
  const float* trc_lut;                          // to_srgb == 5: 3 x 4096 tables, sqrt(linear) -> encoded (an evaluated ICC profile)
  // sample depths: the colour channels / the alpha channel as coded (integers of 1..16 bits, binary16 / binary32 floats), and the
  // output sample type chosen from the colour depth like the reference does (Decoder/JxlDecoder.cpp:510-556): u8, u16, f16, f32
  int32_t sample_bits, alpha_bits, out_bits, out_float;   // out_float: out_bits 16 / 32 are binary16 / binary32 samples
  int32_t sample_exp, alpha_exp;                          // exponent bits of float-coded channels (0: integer samples)
  // alpha is associated (premultiplied): the encoded colour samples are multiplied by 1 / max(alpha, 2^-26) where they are written,
  // as the reference asks of its library (Decoder/JxlDecoder.cpp:233); alpha_unit = 1 / (2^alpha_bits - 1)
  int32_t unpremultiply; float alpha_unit;
  // codestream
  const uint8_t* cs;
  uint64_t cs_size;
  const uint64_t* sec_off;   // logical section -> byte offset in cs
  const uint32_t* sec_size;
  // modular (global tree + code)
  const DevTreeNode* tree;
  int32_t tree_size;
  int32_t pad0;
  DevCode mcode;
  // HF
  DevCode acode;
  int32_t num_presets, num_block_ctx;
  uint8_t block_ctx_map[39 * 64];   // [((c'*13 + ord) * (nqf+1) + qf_idx) * num_lf_ctx + lf_idx]; (nqf+1) * num_lf_ctx <= 64
  // block contexts may also depend on the quantised LF of the block's first cell: per channel (X, Y, B) the number of thresholds it
  // exceeds; the three indices combine in the order X, B, Y  [spec, recalled; no external vector]
  int32_t lf_thr[3][15];
  int32_t n_lf_thr[3];
  int32_t num_lf_ctx;
  uint32_t qf_thr[15];
  int32_t n_qf;
  int32_t custom_orders;    // any non-natural coefficient order in this frame
  const uint16_t* order[kNumOrders * 3];   // coefficient order per (bucket, channel)
  // quantisation
  float inv_global_scale, quant_scale;
  float mul_lf[3];
  float lf_cfl_x, lf_cfl_b;
  float base_x, base_b, inv_color_factor;
  float x_dm, b_dm;
  float qbias[4];
  const float* dq[kNumQuantTables];   // 3 * n floats each (1/weight), stored layout
  uint32_t dq_n[kNumQuantTables];
  // per quant table (which also fixes the coefficient order bucket), per channel, in SCAN order: .x = order[k] (index in the
  // stored layout), .y = bits of the dequantisation weight at that index; channel c starts at scan[q] + c * dq_n[q]
  const U32x2* scan[kNumQuantTables];
  // loop filters / colour
  int32_t gab, epf_iters, skip_lf_smoothing, pad1;
  float gab_w[3][3];        // [c][0..2] normalised centre, edge, corner
  float epf_sharp_lut[8];
  float epf_channel_scale[3];
  float epf_quant_mul, epf_pass0_sigma_scale, epf_pass2_sigma_scale, epf_border_sad_mul;
  float opsin_inv[9];
  float opsin_bias[3], opsin_bias_cbrt[3];
  // workspace planes (device)
  float* lf[3];             // w8*h8, dequantised
  float* lf_tmp[3];         // smoothing output
  float* lf_final[3];       // what LLF reads (lf_tmp when smoothing is on)
  uint8_t* lf_extra;        // per LF group: extra_precision
  int32_t* lfq[3];          // quantised LF (X,Y,B)
  uint32_t* cellinfo;       // per 8x8 cell: strategy | ix << 8 | iy << 13 | log2cx << 18 | log2cy << 21 | valid << 31
  uint16_t* rawq;           // per cell
  uint8_t* sharp;           // per cell
  int8_t* ytox;             // per 64x64 tile
  int8_t* ytob;
  int32_t* binfo;           // scratch per LF group (kBinfoInts ints): cfl x, cfl b, block info rows, sharpness
  ChanDesc* lf_desc;        // per LF group: 8 entries (3 LF channels, 4 HF-metadata channels, spare)
  uint32_t* lf_count;       // per LF group: number of varblocks (0 = phase A failed)
  ChanDesc* alpha_desc;     // per group
  uint32_t* blk_list;       // per group: 1024 x 2 words, varblocks in decode order (hf_blocklist_kernel)
  uint32_t* blk_count;      // per group
  const uint32_t* hf_order; // slot -> group of this (image, pass): the decoded groups sorted by section size, largest first (hf_decode_kernel)
  uint64_t* grp_bitpos;     // per group: codestream bit position after the HF tokens (~0 = failed)
  // Progressive frames: every pass after the first is a record of its own after the batch's images (same block layout, its own code,
  // coefficient orders, entry lists and sections); next_pass chains them.  The alpha stream follows the HF tokens of the LAST pass.
  int32_t num_passes, pass_shift;          // this record's pass contributes value << pass_shift
  int32_t hf_sec_base, alpha_sec_base;     // TOC entry of group 0 of this pass / of the pass that holds the Modular streams
  const DevImage* next_pass;
  const uint64_t* alpha_bitpos;            // grp_bitpos of the last pass
  // single-section frames (they fit one group): the sections share one bit stream
  int32_t single, alpha_in_global;
  uint64_t lf_start_bits;   // where the GPU starts (global alpha channel, then the LF group)
  uint64_t hf_start_bits;   // after HfGlobal (filled in once the host has parsed it)
  uint64_t* lf_end_bits;    // written by lf_group_kernel: bit position after the LF group
  // Band-restricted decode of one large frame (multi-GPU sharding by group rows): groups of rows [dec_gy0, dec_gy1) are
  // decoded (the band plus one halo row each side for the loop filters), pixel rows [band_y0, band_y1) are written to `out`,
  // whose first row is band_y0.  A whole-frame decode has dec_gy0 = 0, dec_gy1 = yg, band = [0, h).
  int32_t dec_gy0, dec_gy1, band_y0, band_y1;
  // Modular frames (lossless): up to 4 channels of the whole image, decoded per group (or, for a frame that fits one
  // group, from the GlobalModular stream), then inverse colour transforms and interleaving in modular_out_kernel
  int32_t is_modular, mod_nch, group_dim, mod_ntr;
  int32_t mod_tr[4][2];     // reversible colour transforms in stream order: begin channel, type (frames without Squeeze: undone in modular_out)
  int32_t* mod_plane[5];    // the image channels in stream order (colour, then the extra channels), w*h each
  int32_t mod_out_pos[5];   // output position of each of them (CMYK: C M Y K [A] whatever the order of the black and alpha channels)
  int32_t cmyk, black_bits; // a black extra channel: DecoderImageFormat::Cmyk; the host wants 0 = no ink, the stream stores 0 = full ink
  const ModChanDev* mod_chan;   // coded channels (after the transforms), mod_ncoded entries
  int32_t mod_ncoded, mod_first_group;   // channels before mod_first_group are coded in the GlobalModular stream
  ChanDesc* mod_desc;       // [section][coded channel]; sections: 0 global, 1 + g LF group g, 1 + nlf + g pass group g
  uint64_t mod_data_bits;   // bit position of the GlobalModular channel data inside LfGlobal
  // weighted-predictor state (only allocated when the MA tree references it): lane-private scratch per LF group / per group
  int32_t* wp_lf;           // [nlf][kWpLfInts]
  int32_t* wp_grp;          // [ng][wp_grp_ints]
  int64_t wp_grp_ints;
  // LZ77 windows (allocated only for codes that use it): decoded values of one stream, 2^log entries per lane
  uint32_t* lz_lf;          // per LF group, 2^20
  uint32_t* lz_grp;         // per group (alpha stream), 2^16
  uint32_t* lz_hf;          // per group (HF coefficient stream), 2^18
  uint32_t* lz_mod;         // per section of a Modular frame, 2^20
  int32_t* alpha32;         // w*h decoded alpha (aliases tmp[0])
  // HF coefficients as decoded (hf_decode_kernel -> recon_tile_kernel): sparse entry lists per group, see kGroupEntriesCap.
  uint32_t* centries;       // group g at centries + (g - centries_g0) * kGroupEntriesCap
  int32_t centries_g0, pad2;
  U32x2* cblk;             // [3][w8 * h8], valid at the origin cell of every varblock: .x = first entry of (block, channel) in its
                            // group's list, .y = number of entries
  int32_t* coef[3];         // wp*hp, footprint layout (generic path / debug taps only): int32 quantised, then float dequantised in place;
                            // shares the third set of pixel-chunk planes with xyb2
  float* tmp[3];            // wp*hp
  float* xyb[3];            // wp*hp
  float* xyb2[3];           // wp*hp
  float* inv_sigma;         // per cell
  // loop-filter / output stage routing: 0 gaborish, 1 epf0, 2 epf1, 3 epf2, 4 output
  float* stage_in[5][3];
  float* stage_out[5][3];
  int32_t stage_on[5];
  int32_t final_stage;      // last enabled filter stage (0..3) converts to u8; 4 = no filter, out_only_kernel converts; 5 = fused kernel
  int32_t fused_gab_epf1;   // 1: Gaborish + one EPF iteration + output as one kernel (filter_stream_kernel); 2: two EPF iterations - that kernel
                            // leaves f32 rows in stage_out[0], filter_stream2_kernel does the second iteration and the output
  float* stream_in[3];      // input planes of the first streaming kernel (after the stage kernels that precede it, if any) ...
  float* stream_mid[3];     // ... and the f32 rows between the two streaming kernels
  int32_t stream_no_gab;    // the first streaming kernel skips Gaborish (frame without it, or done before iteration 0 by its stage kernel)
  int32_t stream_pairs;     // bit 0 / 1: the two-pixels-per-lane form of the fused kernel / of the second iteration's kernel takes this frame
  uint32_t* tile_list;      // 64x64 tiles left to the generic reconstruction kernels (count in status[1])
  uint8_t* alpha;           // w*h samples of the OUTPUT type (u8, or u16 when out_bits == 16), already scaled from alpha_bits
  uint8_t* out;             // w*h*nch_out interleaved samples of the output type
  uint32_t* status;         // [0] error bits, [1..] debug
};

constexpr int kUniGridCells = 2048;   // LDS grid (16-byte leaf records) of the one-section-per-wavefront per-sample Modular decoder (modular_uniform.h)
constexpr int kBinfoInts = 2 * 1024 + 2 * 65536 + 65536 + 65536 + 64;   // ... + prefix sums of the block widths (placement)
constexpr int kWpLfInts = 10 * (65536 + 2);   // widest channel of an LF group section: the block-info rows

struct SectionTask {   // one workgroup's share of sections of one image
  int32_t image;
  int32_t first;   // first LF group / group index
  int32_t count;   // sections handled by this workgroup
  int32_t pad;
};


}  // namespace jxlhip
