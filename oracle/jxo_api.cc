// ORACLE — test infrastructure only (see jxo_common.h header).
// Plain C entry points for ctypes (tests/, bench.py cpu_baseline leg, smoke()).
#include "jxo_codec.h"
#include "jxo_icc.h"
#include "jxo_entropy.h"
#include <string>

using namespace jxo;

static thread_local std::string g_err;

struct JxoImage { DecodeResult r; };
struct JxoBytes { std::vector<uint8_t> b; };

extern "C" {

const char* jxo_last_error() { return g_err.c_str(); }

JxoImage* jxo_decode(const uint8_t* data, size_t size, int num_threads, int want_dump) {
  try {
    JxoImage* im = new JxoImage();
    DecodeOptions o;
    o.num_threads = num_threads;
    o.want_dump = want_dump != 0;
    DecodeJxl(data, size, o, im->r);
    return im;
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void jxo_image_free(JxoImage* im) { delete im; }
void jxo_image_info(const JxoImage* im, int32_t* w, int32_t* h, int32_t* nch, int32_t* w8, int32_t* h8) {
  *w = im->r.out_w; *h = im->r.out_h; *nch = im->r.num_channels;   // as displayed (orientation applied)
  *w8 = im->r.frame.xsize_blocks; *h8 = im->r.frame.ysize_blocks;
}
const uint8_t* jxo_image_pixels(const JxoImage* im) { return im->r.pixels.data(); }
int32_t jxo_image_bits_out(const JxoImage* im) { return im->r.bits_out; }   // 8: u8 samples, 16: little-endian u16 samples
int32_t jxo_image_out_float(const JxoImage* im) { return im->r.out_float ? 1 : 0; }   // bits_out 16 / 32 are binary16 / binary32
int32_t jxo_image_epf_iters(const JxoImage* im) { return im->r.frame.lf.epf_iters; }
size_t jxo_image_exif(const JxoImage* im, const uint8_t** p) { *p = im->r.boxes.exif.data(); return im->r.boxes.exif.size(); }
size_t jxo_image_xml(const JxoImage* im, const uint8_t** p) {
  if (im->r.boxes.xml.empty()) { *p = nullptr; return 0; }
  *p = im->r.boxes.xml[0].data();
  return im->r.boxes.xml[0].size();
}
// name: lf_quant, lf, qcoef, xyb_idct, xyb_filtered (c = 0..2); strategy, raw_quant, sharpness, ytox, ytob, alpha (c ignored)
// Returns element count, sets *ptr and *elem_size.
size_t jxo_image_plane(const JxoImage* im, const char* name, int c, const void** ptr, int32_t* elem_size) {
  const StageDump& d = im->r.dump;
  std::string n(name);
#define RET(v, es) do { *ptr = (v).data(); *elem_size = (es); return (v).size(); } while (0)
  if (n == "lf_quant") RET(d.lf_quant[c], 4);
  if (n == "lf") RET(d.lf[c], 4);
  if (n == "qcoef") RET(d.qcoef[c], 4);
  if (n == "xyb_idct") RET(d.xyb_idct[c], 4);
  if (n == "xyb_filtered") RET(d.xyb_filtered[c], 4);
  if (n == "strategy") RET(d.strategy, 1);
  if (n == "raw_quant") RET(d.raw_quant, 4);
  if (n == "sharpness") RET(d.sharpness, 1);
  if (n == "ytox") RET(d.ytox, 1);
  if (n == "ytob") RET(d.ytob, 1);
  if (n == "alpha") RET(d.alpha, 4);
#undef RET
  *ptr = nullptr; *elem_size = 0;
  return 0;
}

struct JxoEncodeParams {
  float distance;
  int32_t lossless;
  int32_t effort;
  int32_t strategy_mode;
  int32_t fixed_strategy;
  uint32_t seed;
  int32_t epf_iters;
  int32_t gaborish;
  int32_t container;
  int32_t adaptive_lf_smoothing;
  int32_t lossless_predictor;
  int32_t lossless_squeeze;
  int32_t lossless_tree;
  int32_t num_threads;
  int32_t bits;   // bits per sample (8..16); above 8 `px` holds uint16 samples
  int32_t orientation;   // 0 / 1: none; 2..8: EXIF orientation written to the header
  int32_t float_samples; // 0: integer samples; 16 / 32: `px` holds binary16 (as uint16 bit patterns) / binary32 samples
  int32_t colour;        // EncodeParams::colour
};

// Optional inputs of the NEXT jxo_encode call on this thread (kept out of the parameter struct so that its layout stays put):
// an ICC profile to embed, and whether the 4 / 5 channels are CMYK[A].
void jxo_last_encode_token_counts(uint64_t out[4]) { GetLastEncodeTokenCounts(out); }
static thread_local std::vector<uint8_t> g_next_icc;
static thread_local bool g_next_cmyk = false;
static thread_local int g_next_frames = 1;
static thread_local uint32_t g_next_flags = 0;   // 1: explicit (custom) dequantisation tables; 2: prefix codes; 4: LZ77; 8 / 16: two / three passes; 32: custom coefficient orders; 64: block contexts from LF / quant-field thresholds; 128: palette; 256: some DCT8 blocks labelled AFV (refusal tests); 512: alpha signalled as premultiplied
void jxo_set_next_flags(uint32_t flags) { g_next_flags = flags; }
void jxo_set_next_animation(int frames) { g_next_frames = frames; }
void jxo_set_next_icc(const uint8_t* icc, size_t size, int cmyk) {
  g_next_icc.assign(icc ? icc : nullptr, icc ? icc + size : nullptr);
  g_next_cmyk = cmyk != 0;
}
// ICC predicted stream -> profile with the oracle's own reader (0: the stream is invalid)
size_t jxo_icc_from_stream(const uint8_t* enc, size_t size, uint8_t* dst, size_t capacity) {
  try {
    std::vector<uint8_t> out = IccFromStream(std::vector<uint8_t>(enc, enc + size));
    for (size_t i = 0; i < out.size() && i < capacity; i++) dst[i] = out[i];
    return out.size();
  } catch (const std::exception& e) {
    g_err = e.what();
    return 0;
  }
}
size_t jxo_icc_to_stream(const uint8_t* icc, size_t size, uint8_t* dst, size_t capacity) {
  std::vector<uint8_t> out = IccToStream(std::vector<uint8_t>(icc, icc + size));
  for (size_t i = 0; i < out.size() && i < capacity; i++) dst[i] = out[i];
  return out.size();
}
size_t jxo_image_icc(const JxoImage* im, const uint8_t** ptr) { *ptr = im->r.icc.data(); return im->r.icc.size(); }
int jxo_image_cmyk(const JxoImage* im) { return im->r.cmyk ? 1 : 0; }

JxoBytes* jxo_encode(const uint8_t* px, uint32_t w, uint32_t h, int32_t nch, const JxoEncodeParams* ep, const uint8_t* exif,
                     size_t exif_size, const uint8_t* xmp, size_t xmp_size) {
  try {
    EncodeParams p;
    p.distance = ep->distance; p.lossless = ep->lossless != 0; p.effort = ep->effort;
    p.strategy_mode = ep->strategy_mode; p.fixed_strategy = ep->fixed_strategy; p.seed = ep->seed;
    p.epf_iters = ep->epf_iters; p.gaborish = ep->gaborish != 0; p.container = ep->container != 0;
    p.adaptive_lf_smoothing = ep->adaptive_lf_smoothing != 0;
    p.lossless_predictor = ep->lossless_predictor; p.lossless_squeeze = ep->lossless_squeeze != 0; p.lossless_tree = ep->lossless_tree;
    p.num_threads = ep->num_threads;
    p.bits = ep->bits ? ep->bits : 8;
    p.orientation = ep->orientation ? ep->orientation : 1;
    p.float_samples = ep->float_samples;
    p.colour = ep->colour;
    p.icc.swap(g_next_icc); g_next_icc.clear();
    p.cmyk = g_next_cmyk; g_next_cmyk = false;
    p.animation_frames = g_next_frames; g_next_frames = 1;
    p.custom_quant_tables = (g_next_flags & 1) != 0;
    p.custom_orders = (g_next_flags & 32) != 0;
    p.lf_contexts = (g_next_flags & 64) != 0;
    p.palette = (g_next_flags & 128) != 0;
    p.mislabel_afv = (g_next_flags & 256) != 0;
    p.premultiplied_alpha = (g_next_flags & 512) != 0;
    p.num_passes = (g_next_flags & 16) ? 3 : ((g_next_flags & 8) ? 2 : 1);
    const uint32_t g_next_flags_entropy = g_next_flags;
    g_next_flags = 0;
    JxoBytes* b = new JxoBytes();
    SetEntropyTestMode(((g_next_flags_entropy >> 1) & 3));
    try {
      b->b = EncodeJxl(px, w, h, nch, p, exif, exif_size, xmp, xmp_size);
    } catch (...) { SetEntropyTestMode(0); delete b; throw; }
    SetEntropyTestMode(0);
    return b;
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
const uint8_t* jxo_bytes_data(const JxoBytes* b) { return b->b.data(); }
size_t jxo_bytes_size(const JxoBytes* b) { return b->b.size(); }
void jxo_bytes_free(JxoBytes* b) { delete b; }

}  // extern "C"

// ---- table introspection for host-logic tests
extern "C" {
size_t jxo_natural_order(int strategy, uint32_t* dst, size_t capacity) {
  const std::vector<uint32_t>& o = NaturalOrder(strategy);
  for (size_t i = 0; i < o.size() && i < capacity; i++) dst[i] = o[i];
  return o.size();
}
size_t jxo_dequant_table(int strategy, int c, float* dst, size_t capacity) {
  static DequantMatrices dq;
  static bool init = false;
  if (!init) { dq.SetDefault(); init = true; }
  int q = kStrategyQuantTable[strategy];
  const float* p = dq.Get(strategy, c);
  for (size_t i = 0; i < dq.n[q] && i < capacity; i++) dst[i] = p[i];
  return dq.n[q];
}
}
extern "C" {
void jxo_idct_stored(int R, int C, const float* stored, float* out) { IdctStored(R, C, stored, out, C); }
void jxo_dct_stored(int R, int C, const float* in, float* stored) { DctStored(R, C, in, C, stored); }
}
