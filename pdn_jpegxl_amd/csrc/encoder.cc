// Host orchestration of the HIP encode path behind SaveImage (include/jxlfiletypeio.h).
//
// Mirrors EncoderWriteImage (reference: src/JxlFileTypeIO/Encoder/JxlEncoder.cpp:147-392): parameter checks (:155-158),
// progress / cancellation checkpoints (:79-89,162,169,242,253,312,340-344,362), pixel-format analysis (:33-77),
// channel conversion (PixelFormatConversion.cpp:16-121), container output with Exif / XMP boxes (:201,284-310) through
// the host's Write callback in chunks of at most 64 KiB (OutputProcessor.cpp:16,87,134-151).  The arithmetic the
// reference delegates to libjxl (JxlEncoderAddImageFrame :128, JxlEncoderFlushInput :367) runs in encode_kernels.hip.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/jxlfiletypeio.h"
#include "enc_types.h"
#include "host_parse.h"
#include "host_write.h"
#include "icc.h"
#include "kernels.h"

namespace jxlhip {

void LaunchEncAnalyze(const EncImage& im, hipStream_t s);
void LaunchEncFrontEnd(const EncImage& im, hipStream_t s);
void LaunchEncTokens(const EncImage& im, hipStream_t s);
void LaunchEncReverse(const EncImage& im, int which, hipStream_t s);
void LaunchEncSections(const EncImage& im, hipStream_t s);
void LaunchEncCompact(const EncImage& im, const uint64_t* dst_off, uint8_t* dst, int nsec, hipStream_t s);
void LaunchEncLossless(const EncImage& im, int stage, hipStream_t s);

namespace {

struct EncFail : std::runtime_error {
  EncoderStatus status;
  EncFail(EncoderStatus st, const std::string& m) : std::runtime_error(m), status(st) {}
};
#define ENC_HIP(expr)                                                                                                   \
  do {                                                                                                                  \
    hipError_t e_ = (expr);                                                                                             \
    if (e_ == hipErrorOutOfMemory) throw EncFail(EncoderStatus_OutOfMemory, "out of device memory");                    \
    if (e_ != hipSuccess) throw EncFail(EncoderStatus_EncodeError, std::string(#expr) + ": " + hipGetErrorString(e_));  \
  } while (0)

void SetEncErr(ErrorInfo* e, const char* msg) {
  if (!e || !msg) return;
  size_t n = strlen(msg);
  if (n == 0) return;
  if (n > 255) n = 255;
  memcpy(e->errorMessage, msg, n);
  e->errorMessage[n] = 0;
}

// device allocations of one SaveImage call
struct Arena {
  std::vector<void*> ptrs;
  ~Arena() { for (void* p : ptrs) (void)hipFree(p); }
  template <class T> T* Get(size_t count, bool zero = false) {
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(count * sizeof(T), 256);
    ENC_HIP(hipMalloc(&p, bytes));
    ptrs.push_back(p);
    if (zero) ENC_HIP(hipMemset(p, 0, bytes));
    return (T*)p;
  }
  template <class T> T* Upload(const std::vector<T>& v) {
    T* p = Get<T>(std::max<size_t>(v.size(), 1));
    if (!v.empty()) ENC_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
  }
};

// JXLHIP_ENC_TIMING=1: wall time of the encoder's phases on stderr (host + device, synchronised at each checkpoint)
struct PhaseClock {
#ifdef JXLHIP_EXPERIMENTS
  bool on = getenv("JXLHIP_ENC_TIMING") != nullptr;
#else
  bool on = false;
#endif
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void Lap(const char* what) {
    if (!on) return;
    (void)hipDeviceSynchronize();
    const auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[enc] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};

// Device time of the encoder's kernel groups, always recorded (a pair of HIP events on the group's own stream; the streams overlap,
// so the groups can add up to more than the call): what bench.py's encode workload reads through jxlhip_last_save_stage_times.
struct StageMarks {
  struct M { const char* name; hipEvent_t a, b; };
  std::vector<M> marks;
  ~StageMarks() { for (auto& m : marks) { (void)hipEventDestroy(m.a); (void)hipEventDestroy(m.b); } }
  size_t Begin(const char* name, hipStream_t s) {
    M m{name, nullptr, nullptr};
    if (hipEventCreate(&m.a) != hipSuccess || hipEventCreate(&m.b) != hipSuccess) return (size_t)-1;
    (void)hipEventRecord(m.a, s);
    marks.push_back(m);
    return marks.size() - 1;
  }
  void End(size_t i, hipStream_t s) { if (i < marks.size()) (void)hipEventRecord(marks[i].b, s); }
  void Publish(std::vector<std::pair<const char*, float>>* out) {
    out->clear();
    for (auto& m : marks) {
      float ms = 0.f;
      if (hipEventSynchronize(m.b) == hipSuccess && hipEventElapsedTime(&ms, m.a, m.b) == hipSuccess) out->push_back({m.name, ms});
    }
  }
};
thread_local std::vector<std::pair<const char*, float>> g_last_save_stages;

void Progress(ProgressProc progress, int percent) {
  if (progress && !progress(percent)) throw EncFail(EncoderStatus_UserCanceled, "");   // Encoder/JxlEncoder.cpp:79-89
}

}  // namespace

// The fixed MA tree of this encoder, in decode (breadth-first) order; leaf order = context ids of EncLeaf.  (Not file-local: the
// test library's self tests serialise it, csrc/selftest.cc.)
std::vector<EncTreeNode> MakeEncoderTree(uint32_t nlf) {
  auto split = [](int prop, int32_t v) { return EncTreeNode{prop, v, 0, 0, 1}; };
  auto leaf = [](int pred, int32_t offset = 0) { return EncTreeNode{-1, 0, pred, offset, 1}; };
  std::vector<EncTreeNode> t;
  t.push_back(split(1, (int32_t)(3 * nlf + kNumQuantTables)));   // 0: stream > LF + metadata + quant-table streams ? alpha : 2
  t.push_back(leaf(5));                                          // 1: alpha of a pass group              (kLeafAlpha)
  t.push_back(split(1, (int32_t)(2 * nlf)));                     // 2: HF metadata (3) : 4
  t.push_back(split(0, 2));                                      // 3: channel 3 = sharpness (5) : 6
  t.push_back(split(1, 0));                                      // 4: LF coefficients (7) : global alpha (8)
  t.push_back(leaf(0, 4));                                       // 5: EPF sharpness, constant 4            (kLeafSharp)
  t.push_back(split(0, 1));                                      // 6: channel 2 = block info (9) : chroma-from-luma maps (10)
  t.push_back(split(0, 1));                                      // 7: channel 2 = B (11) : 12
  t.push_back(leaf(5));                                          // 8: alpha coded in the global section   (kLeafAlphaGlobal)
  t.push_back(split(2, 0));                                      // 9: row 1 = quant field (13) : row 0 = strategies (14)
  t.push_back(leaf(0));                                          // 10: chroma-from-luma maps, constant 0   (kLeafCfl)
  t.push_back(leaf(5));                                          // 11: LF of B                            (kLeafLfB)
  t.push_back(split(0, 0));                                      // 12: channel 1 = X (15) : channel 0 = Y (16)
  t.push_back(leaf(1));                                          // 13: quant field row, West predictor    (kLeafQf)
  t.push_back(leaf(0));                                          // 14: strategy row, constant 0 (DCT8)    (kLeafStrategy)
  t.push_back(leaf(5));                                          // 15: LF of X                            (kLeafLfX)
  t.push_back(leaf(5));                                          // 16: LF of Y                            (kLeafLfY)
  return t;
}

namespace {

int32_t HResultToStatus(int32_t hr) {   // OutputProcessor.cpp:134-151
  if (hr >= 0) return EncoderStatus_Ok;
  if ((uint32_t)hr == 0x80004004u) return EncoderStatus_UserCanceled;
  if ((uint32_t)hr == 0x8007000Eu) return EncoderStatus_OutOfMemory;
  return EncoderStatus_WriteError;
}

EncCodeDev UploadCode(Arena& A, const EncCode& c) {
  EncCodeDev d;
  d.ctx_map = A.Upload(c.ctx_map);
  d.freq = A.Upload(c.freq);
  d.start = A.Upload(c.start);
  d.rmap = A.Upload(c.rmap);
  d.num_clusters = c.num_clusters;
  d.num_ctx = (uint32_t)c.ctx_map.size();
  return d;
}

// Container + output through the host callbacks (always boxes, Encoder/JxlEncoder.cpp:201) in chunks of at most 64 KiB.
void EmitFile(const std::vector<uint8_t>& codestream, const EncoderImageMetadata* md, IOCallbacks* io, ProgressProc progress) {
  std::vector<uint8_t> file = WriteContainer(codestream, md->exif, md->exifSize, md->xmp, md->xmpSize);
  const size_t kChunk = 64 * 1024;   // OutputProcessor.cpp:16
  size_t done = 0;
  int last_pct = 30;
  while (done < file.size()) {
    const size_t n = std::min(kChunk, file.size() - done);
    const int32_t st = HResultToStatus(io->Write(file.data() + done, n));
    if (st != EncoderStatus_Ok) throw EncFail(st, st == EncoderStatus_WriteError ? "the output stream rejected a write" : "");
    done += n;
    const int pct = 30 + (int)(60.0 * done / file.size()) / 5 * 5;   // 30 -> 90 in steps of 5 (:340-344)
    if (pct > last_pct) { Progress(progress, pct); last_pct = pct; }
  }
  Progress(progress, 95);
}

// Geometry + upload + pixel-format analysis shared by the lossy and the lossless path (Encoder/JxlEncoder.cpp:33-77).
void BeginImage(const BitmapData* bmp, const EncoderImageMetadata* md, Arena& A, EncImage& im, ProgressProc progress) {
  const uint32_t w = bmp->width, h = bmp->height;
  if (!w || !h || !bmp->scan0 || bmp->stride < (uint64_t)w * 4) throw EncFail(EncoderStatus_EncodeError, "invalid bitmap");
  memset(&im, 0, sizeof(im));
  im.w = (int32_t)w; im.h = (int32_t)h;
  im.w8 = (int32_t)((w + 7) / 8); im.h8 = (int32_t)((h + 7) / 8);
  im.wp = im.w8 * 8; im.hp = im.h8 * 8;
  im.xg = (int32_t)((w + 255) / 256); im.yg = (int32_t)((h + 255) / 256); im.ng = im.xg * im.yg;
  im.xlf = (int32_t)((w + 2047) / 2048); im.ylf = (int32_t)((h + 2047) / 2048); im.nlf = im.xlf * im.ylf;
  uint8_t* d_bgra = A.Get<uint8_t>((size_t)bmp->stride * h);
  ENC_HIP(hipMemcpy(d_bgra, bmp->scan0, (size_t)bmp->stride * h, hipMemcpyHostToDevice));
  im.bgra = d_bgra; im.stride = (int32_t)bmp->stride;
  im.flags = A.Get<uint32_t>(4, true);
  LaunchEncAnalyze(im, nullptr);
  uint32_t flags[2] = {0, 0};
  ENC_HIP(hipMemcpy(flags, im.flags, sizeof(flags), hipMemcpyDeviceToHost));
  im.gray = flags[0] ? 0 : 1;       // every pixel r == g == b and no ICC profile: one colour channel (:67-70)
  if (md->iccProfile && md->iccProfileSize) im.gray = 0;   // an RGB profile must keep describing RGB samples (:71-75)
  im.has_alpha = flags[1] ? 1 : 0;  // some pixel a < 255 (:54-57)
  Progress(progress, 5);
}

// Lossless: a Modular frame in the original (sRGB) colour space (Encoder/JxlEncoder.cpp:214,325); 256x256 groups, YCoCg-R for RGB,
// gradient predictor, one context per channel.
void EncodeLossless(const BitmapData* bmp, const EncoderImageMetadata* md, IOCallbacks* io, ProgressProc progress) {
  Arena A;
  hipStream_t s = nullptr;
  EncImage im;
  BeginImage(bmp, md, A, im, progress);
  const size_t npx = (size_t)im.w * im.h;
  im.lossless = 1;
  im.ll_nch = (im.gray ? 1 : 3) + im.has_alpha;
  im.ll_rct = im.gray ? 0 : 1;
  for (int c = 0; c < im.ll_nch; c++) im.ll_plane[c] = A.Get<int32_t>(npx);
  im.tok_ll = A.Get<DevToken>((size_t)im.ng * kLlTokCap);
  im.hist_mod = A.Get<uint32_t>(kNumEncLeaves * kEncSyms, true);
  Progress(progress, 15);
  LaunchEncLossless(im, 0, s);
  std::vector<uint32_t> hist(4 * kEncSyms);
  ENC_HIP(hipMemcpy(hist.data(), im.hist_mod, hist.size() * 4, hipMemcpyDeviceToHost));
  Progress(progress, 25);
  // LfGlobal: tree on the channel index (leaf ids: channel 3, 2, 1, 0), code, GlobalModular header
  const bool single = im.ng == 1;
  BitWriter lf_global;
  EncCode mcode;
  lf_global.Bool(true);   // default LF dequantisation factors (read for every frame encoding)
  lf_global.Bool(true);   // global MA tree
  {
    std::vector<EncTreeNode> t;
    t.push_back(EncTreeNode{0, 1, 0, 0, 1});     // 0: channel > 1 ? 1 : 2
    t.push_back(EncTreeNode{0, 2, 0, 0, 1});     // 1: channel > 2 ? 3 : 4
    t.push_back(EncTreeNode{0, 0, 0, 0, 1});     // 2: channel > 0 ? 5 : 6
    for (int i = 0; i < 4; i++) t.push_back(EncTreeNode{-1, 0, 5, 0, 1});   // gradient predictor leaves: channel 3, 2, 1, 0
    WriteTree(t, lf_global);
  }
  BuildAndWriteCode(hist.data(), 4, 4, {}, lf_global, mcode);
  lf_global.Bool(true);   // use the global tree
  lf_global.Bool(true);   // default weighted-predictor parameters
  lf_global.U32(WV(0), WV(1), WB(4, 2), WB(8, 18), im.ll_rct ? 1 : 0);
  if (im.ll_rct) {
    lf_global.Write(2, 0);                                            // transform id 0: reversible colour transform
    lf_global.U32(WB(3), WB(6, 8), WB(10, 72), WB(13, 1096), 0);      // first channel
    lf_global.U32(WV(6), WB(2), WB(4, 2), WB(6, 10), 6);              // type 6: YCoCg-R
  }
  im.mcode = UploadCode(A, mcode);
  const int nsec = im.ng;
  im.sec_cap = ((size_t)kLlTokCap * 6 + 256) & ~(size_t)15;
  im.sec_bytes = A.Get<uint8_t>((size_t)nsec * im.sec_cap);
  im.sec_bits = A.Get<uint64_t>(nsec, true);
  im.stream_state = A.Get<uint32_t>((size_t)2 * (im.nlf + im.ng) + 1, true);
  LaunchEncLossless(im, 1, s);
  std::vector<uint64_t> sec_bits(nsec);
  ENC_HIP(hipMemcpy(sec_bits.data(), im.sec_bits, sec_bits.size() * 8, hipMemcpyDeviceToHost));
  std::vector<uint64_t> off(nsec + 1, 0);
  for (int i = 0; i < nsec; i++) off[i + 1] = off[i] + ((sec_bits[i] + 7) >> 3);
  std::vector<uint8_t> packed(std::max<uint64_t>(off[nsec], 1));
  {
    uint64_t* d_off = A.Upload(off);
    uint8_t* d_packed = A.Get<uint8_t>(packed.size());
    LaunchEncCompact(im, d_off, d_packed, nsec, s);
    ENC_HIP(hipMemcpy(packed.data(), d_packed, packed.size(), hipMemcpyDeviceToHost));
  }
  ENC_HIP(hipGetLastError());
  Progress(progress, 30);
  EncImageInfo ii;
  ii.xsize = (uint32_t)im.w; ii.ysize = (uint32_t)im.h; ii.gray = im.gray; ii.alpha = im.has_alpha; ii.xyb = false;
  ii.icc = md->iccProfile; ii.icc_size = md->iccProfile ? md->iccProfileSize : 0;
  EncFrameInfo fi;
  fi.encoding = 1; fi.group_size_shift = 1; fi.gab = false; fi.epf_iters = 0;
  BitWriter cs;
  WriteCodestreamHeaders(ii, cs);
  WriteFrameHeader(ii, fi, cs);
  std::vector<std::vector<uint8_t>> sections;
  if (single) {
    lf_global.AppendBits(packed.data(), sec_bits[0]);   // the channels of a frame that fits one group belong to the global stream
    sections.push_back(lf_global.Finish());
  } else {
    sections.push_back(lf_global.Finish());
    for (int g = 0; g < im.nlf; g++) sections.emplace_back();   // no channel is small enough for the LF groups
    sections.emplace_back();                                    // HfGlobal is empty in a Modular frame
    for (int g = 0; g < im.ng; g++) sections.emplace_back(packed.begin() + off[g], packed.begin() + off[g + 1]);
  }
  std::vector<uint32_t> sizes;
  for (auto& sec : sections) sizes.push_back((uint32_t)sec.size());
  WriteToc(sizes, cs);
  std::vector<uint8_t> codestream = cs.Finish();
  for (auto& sec : sections) codestream.insert(codestream.end(), sec.begin(), sec.end());
  EmitFile(codestream, md, io, progress);
}

void EncodeLossy(const BitmapData* bmp, const EncoderOptions* opt, const EncoderImageMetadata* md, IOCallbacks* io, ProgressProc progress) {
  Arena A;
  hipStream_t s = nullptr;
  EncImage im;
  PhaseClock clk;
  BeginImage(bmp, md, A, im, progress);
  if (md->iccProfile && md->iccProfileSize) {
    // Lossy with a profile: the samples reach XYB through the profile (the reference leaves that to its encoder library's colour
    // management, Encoder/JxlEncoder.cpp:258-268).  Matrix / TRC profiles are evaluated here; a profile that would need a full
    // colour management system cannot be encoded lossily without misrepresenting its colours.
    IccModel model;
    if (!IccBuildModel(md->iccProfile, md->iccProfileSize, &model) || model.gray)
      throw EncFail(EncoderStatus_EncodeError, "this ICC profile is not an RGB matrix/TRC profile: lossy saving needs a full colour management system (save lossless instead)");
    std::vector<float> lin;
    for (int c = 0; c < 3; c++) lin.insert(lin.end(), model.to_linear[c].begin(), model.to_linear[c].end());
    float* d_lin = A.Get<float>(lin.size());
    ENC_HIP(hipMemcpy(d_lin, lin.data(), lin.size() * 4, hipMemcpyHostToDevice));
    im.icc_lin = d_lin;
    for (int k = 0; k < 9; k++) im.icc_to_srgb[k] = (float)model.to_linear_srgb[k];
  }
  clk.Lap("upload + analysis");
  const uint32_t w = bmp->width, h = bmp->height;
  const size_t npx = (size_t)w * h, npad = (size_t)im.wp * im.hp, ncell = (size_t)im.w8 * im.h8;
  // ---- 2. quantiser and loop-filter parameters (distance, :319); the effort picks the transform set below
  const float distance = std::max(0.05f, std::min(25.0f, opt->distance));
  EncFrameInfo fi;
  fi.encoding = 0;
  fi.gab = true;
  fi.epf_iters = 0;
  for (float t : {0.7f, 1.5f, 4.0f}) if (distance >= t) fi.epf_iters++;
  const double qf = 0.85 / distance;
  const uint32_t global_scale = (uint32_t)std::max<long>(1, std::min<long>(8193 + 65535, std::lrint(65536.0 * qf / 16.0)));
  const uint32_t quant_lf = (uint32_t)std::max<long>(1, std::min<long>(65536, std::lrint(1.1 / distance * 65536.0 / global_scale)));
  const float inv_gs = 65536.0f / global_scale;
  const float m_lf[3] = {1.0f / 4096, 1.0f / 512, 1.0f / 256};
  for (int c = 0; c < 3; c++) im.inv_mul_lf[c] = 1.0f / (m_lf[c] * inv_gs / quant_lf);
  im.mul_lf_y = m_lf[1] * inv_gs / quant_lf;
  im.inv_gs = inv_gs;
  im.x_dm = std::pow(0.8f, (float)fi.x_qm_scale - 2.0f);
  im.b_dm = std::pow(0.8f, (float)fi.b_qm_scale - 2.0f);
  im.qbias1 = 1.0f - 0.07005449891748593f;
  im.qbias3 = 0.145f;
  im.gab = fi.gab;
  {
    const float w1 = 0.115169525f, w2 = 0.061248592f, div = 1.0f + 4.0f * (w1 + w2);
    for (int c = 0; c < 3; c++) { im.gab_w[c][0] = 1.0f / div; im.gab_w[c][1] = w1 / div; im.gab_w[c][2] = w2 / div; }
  }
  // effort (JxlEncoderTypes.h:29, passed to the encoder library as its effort setting, Encoder/JxlEncoder.cpp:319-326): the library's
  // fast settings (1..4) keep every block an 8x8 DCT; 5 and 6 add 16x16 / 32x32 DCTs on flat regions; from 7 (the host's default) the
  // 64x64 and the rectangular 16x8 ... 64x32 shapes join
  im.squares = opt->effort >= 7 ? 2 : (opt->effort >= 5 ? 1 : 0);
  const StaticTables& st = GetStaticTables();
  {
    for (auto& q : im.scan_of) q = nullptr;
    for (auto& q : im.dq) q = nullptr;
    for (int o : {0, 2, 3, 4, 6, 7, 8}) {   // order buckets of DCT8, 16x16, 32x32, 16x8 / 8x16, 32x16 / 16x32, 64x64, 64x32 / 32x64
      const std::vector<uint16_t>& order = st.natural_order[o];   // scan position -> stored index
      std::vector<uint16_t> inv(order.size());
      for (size_t k = 0; k < order.size(); k++) inv[order[k]] = (uint16_t)k;
      im.scan_of[o] = A.Upload(inv);
    }
    for (int q : {0, 4, 5, 6, 8, 11, 12}) im.dq[q] = A.Upload(st.dq[q]);
    std::vector<float> rs(16 * 64, 1.0f);
    for (int l = 0; l < 4; l++) {
      const int N = 8 << l, c = 1 << l;
      im.basis[l] = A.Upload(st.basis[l]);
      std::vector<float> div(st.basis[l]);
      for (auto& v : div) v = v / (float)N;
      im.basis_div[l] = A.Upload(div);
      std::vector<float> small((size_t)c * c);
      for (int k = 0; k < c; k++)
        for (int n = 0; n < c; n++) small[(size_t)k * c + n] = (float)((k ? std::sqrt(2.0) : 1.0) * std::cos((2 * n + 1) * k * M_PI / (2.0 * c)));
      im.bsmall[l] = A.Upload(small);
    }
    auto resample = [](int c, int k) {   // coefficient k of an 8c-point DCT from the c-point DCT of the block means
      if (k == 0) return 1.0;
      const double t = k * M_PI / (2.0 * c);
      return 1.0 / (std::cos(t / 2) * std::cos(t / 4) * std::cos(t / 8));
    };
    for (int ly = 0; ly < 4; ly++)
      for (int lx = 0; lx < 4; lx++)
        for (int ky = 0; ky < (1 << ly); ky++)
          for (int kx = 0; kx < (1 << lx); kx++) rs[(ly * 4 + lx) * 64 + ky * 8 + kx] = (float)(resample(1 << ly, ky) * resample(1 << lx, kx));
    im.rs = A.Upload(rs);
  }
  // ---- 3. planes, front end
  for (int c = 0; c < 3; c++) {
    im.xyb[c] = A.Get<float>(npx);
    im.pad[c] = A.Get<float>(npad);
    im.lfq[c] = A.Get<int32_t>(ncell);
    im.qs[c] = A.Get<int32_t>(ncell * 64);
    im.nz[c] = A.Get<uint8_t>(ncell);
    im.nzc[c] = A.Get<uint16_t>(ncell);
    im.last[c] = A.Get<uint16_t>(ncell);
  }
  im.rawq = A.Get<int32_t>(ncell);
  im.act = A.Get<float>(ncell);
  im.strat = A.Get<uint8_t>(ncell);
  if (im.has_alpha) im.alpha_px = A.Get<int32_t>(npx);
  Progress(progress, 15);
  clk.Lap("allocation");
  StageMarks marks;
  { const size_t k = marks.Begin("front_end (xyb, sharpen, activity, strategy, DCT + quantise)", s); LaunchEncFrontEnd(im, s); marks.End(k, s); }
  clk.Lap("xyb + sharpen + dct/quant");
  Progress(progress, 20);
  // ---- 4. tokens + histograms
  im.tok_lf = A.Get<DevToken>((size_t)im.nlf * kLfTokCap);
  im.tok_meta = A.Get<DevToken>((size_t)im.nlf * kMetaTokCap);
  im.tok_ac = A.Get<DevToken>((size_t)im.ng * kAcTokCap);
  if (im.has_alpha) im.tok_alpha = A.Get<DevToken>((size_t)im.ng * kAlphaTokCap);
  im.n_ac = A.Get<uint32_t>(im.ng, true);
  im.n_meta = A.Get<uint32_t>(im.nlf, true);
  im.hist_mod = A.Get<uint32_t>(kNumEncLeaves * kEncSyms, true);
  im.hist_ac = A.Get<uint32_t>((size_t)kAcContexts * kEncSyms, true);
  { const size_t k = marks.Begin("tokens + histograms", s); LaunchEncTokens(im, s); marks.End(k, s); }
  std::vector<uint32_t> hist_mod(kNumEncLeaves * kEncSyms), hist_ac((size_t)kAcContexts * kEncSyms);
  ENC_HIP(hipMemcpy(hist_mod.data(), im.hist_mod, hist_mod.size() * 4, hipMemcpyDeviceToHost));
  ENC_HIP(hipMemcpy(hist_ac.data(), im.hist_ac, hist_ac.size() * 4, hipMemcpyDeviceToHost));
  clk.Lap("tokens + histograms");
  Progress(progress, 25);
  // ---- 5. LfGlobal and HfGlobal (host): quantiser, MA tree, entropy codes
  const bool single = im.ng == 1;
  BitWriter lf_global, hf_global;
  EncCode mcode, acode;
  lf_global.Bool(true);   // default LF dequantisation factors
  lf_global.U32(WB(11, 1), WB(11, 2049), WB(12, 4097), WB(16, 8193), global_scale);
  lf_global.U32(WV(16), WB(5, 1), WB(8, 1), WB(16, 1), quant_lf);
  lf_global.Bool(true);   // default block-context map
  lf_global.Bool(true);   // default LF chroma-from-luma parameters
  lf_global.Bool(true);   // global MA tree
  WriteTree(MakeEncoderTree((uint32_t)im.nlf), lf_global);
  {
    std::vector<uint8_t> pinned(kNumEncLeaves, 0);
    pinned[kLeafSharp] = pinned[kLeafCfl] = 1;   // constant channels: no token is ever written
    pinned[kLeafStrategy] = im.squares ? 0 : 1;  // ... and so is the strategy row while every block is an 8x8 DCT
    BuildAndWriteCode(hist_mod.data(), kNumEncLeaves, 8, pinned, lf_global, mcode);
  }
  clk.Lap("host: tree + modular code");
  // The Modular code is ready: start the recurrences of its streams (the LF coefficients of an LF group are the longest of the frame)
  // on a stream of their own, and build the HF code meanwhile.
  hipStream_t s_ans = nullptr, s_hf = nullptr;
  ENC_HIP(hipStreamCreateWithFlags(&s_ans, hipStreamNonBlocking));
  struct StreamGuard { hipStream_t s; ~StreamGuard() { if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } } } s_ans_guard{s_ans};
  ENC_HIP(hipStreamCreateWithFlags(&s_hf, hipStreamNonBlocking));   // the HF streams' recurrences run beside the Modular ones
  StreamGuard s_hf_guard{s_hf};
  const int nsec = im.nlf + im.ng + 1;   // + the global alpha stream of single-group frames
  im.mcode = UploadCode(A, mcode);
  im.sec_cap = ((size_t)std::max(kLfTokCap + kMetaTokCap, kAcTokCap + kAlphaTokCap) * 6 + 256) & ~(size_t)15;
  im.sec_bytes = A.Get<uint8_t>((size_t)nsec * im.sec_cap);
  im.sec_bits = A.Get<uint64_t>(nsec, true);
  im.stream_state = A.Get<uint32_t>((size_t)2 * (im.nlf + im.ng) + 1, true);
  ENC_HIP(hipDeviceSynchronize());   // tokens, histogram downloads and the clears above are done before the other stream starts
  { const size_t k = marks.Begin("ans recurrences, Modular streams (LF, metadata, alpha)", s_ans); LaunchEncReverse(im, 0, s_ans); marks.End(k, s_ans); }
  if (im.has_alpha) lf_global.Write(4, 3);   // global Modular image header: global tree, default predictor, no transforms
  hf_global.Bool(true);                      // default dequantisation matrices
  hf_global.Write(im.ng <= 1 ? 0 : 32 - __builtin_clz((unsigned)(im.ng - 1)), 0);   // one HF preset
  hf_global.U32(WV(0x5F), WV(0x13), WV(0), WB(kNumOrders), 0);                       // natural coefficient orders
  {
    const auto t0 = std::chrono::steady_clock::now();
    BuildAndWriteCode(hist_ac.data(), kAcContexts, 64, {}, hf_global, acode);
    if (clk.on) fprintf(stderr, "[enc] %-28s %8.2f ms (host only)\n", "HF code construction", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  clk.Lap("host: HF code (overlaps the LF recurrences)");
  // ---- 6. ANS coding of every section on the GPU (the Modular streams' recurrences have been running since their code was built)
  im.acode = UploadCode(A, acode);
  { const size_t k = marks.Begin("ans recurrences, HF streams", s_hf); LaunchEncReverse(im, 1, s_hf); marks.End(k, s_hf); }
  {
    hipEvent_t hf_done;
    ENC_HIP(hipEventCreateWithFlags(&hf_done, hipEventDisableTiming));
    ENC_HIP(hipEventRecord(hf_done, s_hf));
    ENC_HIP(hipStreamWaitEvent(s_ans, hf_done, 0));
    const size_t k = marks.Begin("section bit layout", s_ans);
    LaunchEncSections(im, s_ans);   // bit layout of every section: needs the states of both kinds of stream
    marks.End(k, s_ans);
    ENC_HIP(hipStreamSynchronize(s_ans));
    (void)hipEventDestroy(hf_done);
  }
  clk.Lap("ans sections");
  std::vector<uint64_t> sec_bits(nsec);
  ENC_HIP(hipMemcpy(sec_bits.data(), im.sec_bits, sec_bits.size() * 8, hipMemcpyDeviceToHost));
  std::vector<uint64_t> off(nsec + 1, 0);
  for (int i = 0; i < nsec; i++) off[i + 1] = off[i] + ((sec_bits[i] + 7) >> 3);
  std::vector<uint8_t> packed(std::max<uint64_t>(off[nsec], 1));
  {
    uint64_t* d_off = A.Upload(off);
    uint8_t* d_packed = A.Get<uint8_t>(packed.size());
    const size_t k = marks.Begin("compact", s);
    LaunchEncCompact(im, d_off, d_packed, nsec, s);
    marks.End(k, s);
    ENC_HIP(hipMemcpy(packed.data(), d_packed, packed.size(), hipMemcpyDeviceToHost));
  }
  ENC_HIP(hipGetLastError());
  marks.Publish(&g_last_save_stages);
  clk.Lap("compact + download");
  Progress(progress, 30);
  // ---- 7. codestream assembly
  EncImageInfo ii;
  ii.xsize = w; ii.ysize = h; ii.gray = im.gray; ii.alpha = im.has_alpha; ii.xyb = true;
  ii.icc = md->iccProfile; ii.icc_size = md->iccProfile ? md->iccProfileSize : 0;
  BitWriter cs;
  WriteCodestreamHeaders(ii, cs);
  WriteFrameHeader(ii, fi, cs);
  std::vector<std::vector<uint8_t>> sections;
  if (single) {
    // LfGlobal | LfGroup | HfGlobal | PassGroup share one bit stream
    if (im.has_alpha) lf_global.AppendBits(packed.data() + off[im.nlf + im.ng], sec_bits[im.nlf + im.ng]);
    const uint64_t hg_bits = hf_global.BitCount();
    std::vector<uint8_t> hg = hf_global.Finish();
    lf_global.AppendBits(packed.data() + off[0], sec_bits[0]);
    lf_global.AppendBits(hg.data(), hg_bits);
    lf_global.AppendBits(packed.data() + off[1], sec_bits[1]);
    sections.push_back(lf_global.Finish());
  } else {
    sections.push_back(lf_global.Finish());
    for (int g = 0; g < im.nlf; g++) sections.emplace_back(packed.begin() + off[g], packed.begin() + off[g + 1]);
    sections.push_back(hf_global.Finish());
    for (int g = 0; g < im.ng; g++) sections.emplace_back(packed.begin() + off[im.nlf + g], packed.begin() + off[im.nlf + g + 1]);
  }
  std::vector<uint32_t> sizes;
  for (auto& sec : sections) sizes.push_back((uint32_t)sec.size());
  WriteToc(sizes, cs);
  std::vector<uint8_t> codestream = cs.Finish();
  for (auto& sec : sections) codestream.insert(codestream.end(), sec.begin(), sec.end());
  clk.Lap("assembly");
  EmitFile(codestream, md, io, progress);
  clk.Lap("container + write callbacks");
}

}  // namespace
}  // namespace jxlhip

using namespace jxlhip;

// Device time of the kernel groups of this thread's last lossy SaveImage (see StageMarks)
extern "C" JXLFILETYPEIO_API int32_t jxlhip_last_save_stage_times(const char** names, float* ms, int32_t capacity) {
  int32_t k = 0;
  for (auto& e : g_last_save_stages) {
    if (k >= capacity) break;
    if (names) names[k] = e.first;
    if (ms) ms[k] = e.second;
    k++;
  }
  return k;
}

extern "C" JXLFILETYPEIO_API EncoderStatus SaveImage(const BitmapData* bitmap, const EncoderOptions* options, const EncoderImageMetadata* metadata,
                                   IOCallbacks* callbacks, ErrorInfo* err, ProgressProc progress) {
  if (!bitmap || !options || !callbacks || !metadata) return EncoderStatus_NullParameter;   // Encoder/JxlEncoder.cpp:155-158
  try {
    Progress(progress, 0);   // :162
    if (!callbacks->Write) throw EncFail(EncoderStatus_NullParameter, "");
    if (options->lossless) EncodeLossless(bitmap, metadata, callbacks, progress);   // :214,325
    else EncodeLossy(bitmap, options, metadata, callbacks, progress);
    return EncoderStatus_Ok;
  } catch (const EncFail& e) {
    if (e.status == EncoderStatus_EncodeError) SetEncErr(err, e.what());
    return e.status;
  } catch (const std::bad_alloc&) {
    return EncoderStatus_OutOfMemory;
  } catch (const std::exception& e) {
    SetEncErr(err, e.what());
    return EncoderStatus_EncodeError;
  } catch (...) {
    return EncoderStatus_EncodeError;   // never throw across the ABI (:382-389)
  }
}
