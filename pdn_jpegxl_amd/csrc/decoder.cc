// Host orchestration of the HIP decode path and the C-ABI (include/jxlfiletypeio.h).
//
// LoadImage mirrors DecoderReadImage (reference: src/JxlFileTypeIO/Decoder/JxlDecoder.cpp:796-852):
// pass 1 = headers + metadata callbacks (:412-793), pass 2 = frame -> one interleaved buffer ->
// setLayerData (:217-410).  The arithmetic that the reference delegates to libjxl
// (JxlDecoderProcessInput, :252) runs in the kernels of kernels.hip.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/jxlfiletypeio.h"
#include "host_parse.h"
#include "icc.h"
#include "kernels.h"

namespace jxlhip {

struct HipError : std::runtime_error {
  explicit HipError(const std::string& m) : std::runtime_error(m) {}
};
#define HIP_OK(expr)                                                                                      \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) throw HipError(std::string(#expr) + ": " + hipGetErrorString(e_));              \
  } while (0)

static void SetErr(ErrorInfo* e, const char* fmt, ...) {
  if (!e) return;
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  int n = vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  // reference semantics (Common.cpp:18-53): only messages of 1..255 chars are stored
  if (n > 0 && n <= 255) memcpy(e->errorMessage, buf, (size_t)n + 1);
  else if (n > 255) { memcpy(e->errorMessage, buf, 255); e->errorMessage[255] = 0; }
}

static inline size_t Align(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Bump {
  size_t off = 0;
  size_t Take(size_t bytes, size_t align = 256) {
    off = Align(off, align);
    size_t r = off;
    off += bytes;
    return r;
  }
};

struct StageTimer {
  std::vector<const char*> names;
  std::vector<hipEvent_t> ev;
};

}  // namespace jxlhip

using namespace jxlhip;

// Bytes per output sample: the sample type follows the colour channels' depth (Decoder/JxlDecoder.cpp:510-556): u8, u16, f16, f32.
// Experiment knobs (environment variables read by tools/sweep_*.sh) exist only in a library built with -DJXLHIP_EXPERIMENTS
// (JXLHIP_EXTRA_CFLAGS): a host process's environment must not be able to change what the shipping library decodes.
static inline const char* Knob(const char* name) {
#ifdef JXLHIP_EXPERIMENTS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}
static inline bool PowerOfTwoUpTo64(int v) { return v >= 1 && v <= 64 && (v & (v - 1)) == 0; }
static inline int KnobStride(const char* name, int fallback) {   // lane strides: powers of two in 1..64, anything else is ignored
  const char* e = Knob(name);
  if (!e) return fallback;
  const int v = atoi(e);
  return PowerOfTwoUpTo64(v) ? v : fallback;
}
static inline size_t OutBytesPerSample(const ParsedFrame& f) { return f.exp_bits ? (f.bits <= 16 ? 2 : 4) : (f.bits > 8 ? 2 : 1); }

// Order bucket of a quant table (every strategy of a quant table shares one bucket).
static int OrderBucketOfQuantTable(int q) {
  for (int s = 0; s < kNumStrategies; s++) if (kStrategyQuantTable[s] == q) return kStrategyOrderBucket[s];
  return 0;
}
// Scan list of quant table q: for each channel (X, Y, B) and scan position k, the stored-layout index order[k] and the bits of
// the dequantisation weight at that index.  custom: the frame's own coefficient orders ([bucket][channel], empty = natural).
static void BuildScanList(int q, const std::vector<uint16_t> (*custom)[3], std::vector<U32x2>& out, const std::vector<float>* custom_dq = nullptr) {
  const StaticTables& st = GetStaticTables();
  const int o = OrderBucketOfQuantTable(q);
  const std::vector<float>& dq = (custom_dq && custom_dq->size() == st.dq[q].size()) ? *custom_dq : st.dq[q];
  const size_t n = st.dq[q].size() / 3;
  out.resize(3 * n);
  for (int c = 0; c < 3; c++) {
    const std::vector<uint16_t>& ord = (custom && !custom[o][c].empty()) ? custom[o][c] : st.natural_order[o];
    for (size_t k = 0; k < n; k++) {
      const uint32_t p = k < ord.size() ? ord[k] : 0u;
      uint32_t wb;
      const float w = dq[(size_t)c * n + (p < n ? p : 0)];
      memcpy(&wb, &w, 4);
      out[(size_t)c * n + k] = U32x2{p, wb};
    }
  }
}

struct JxlHipDecoder {
  int device = 0;
  hipStream_t own_stream = nullptr;
  // static tables
  float* d_basis_all = nullptr;
  float* d_basis_small = nullptr;
  float* d_basis_mfma = nullptr;   // the 32- and 64-point bases laid out per matrix-core lane (tile_kernels.hip: MfmaChainT)
  float* d_llf_scale = nullptr;
  uint16_t* d_natural[kNumOrders] = {};
  U32x2* d_scan[kNumQuantTables] = {};   // per quant table: {order[k], weight bits} in scan order, 3 channels (natural orders, library tables)
  float* d_dq[kNumQuantTables] = {};
  uint32_t dq_n[kNumQuantTables] = {};
  // Per-batch state lives in one of three slots so that, when the caller does not synchronise between batches, the LF
  // stage of batch k+2 (stream_lf), the HF-coefficient stage of batch k+1 (stream_hf) and the alpha + pixel stages of
  // batch k (main stream) run concurrently: the serial entropy kernels leave most issue slots of the chip idle.
  struct Tap { std::vector<uint8_t> qcoef[3], xyb_idct[3], xyb_filtered[3]; };
  struct Slot {
    // grow-only buffers
    uint8_t* d_ws = nullptr; size_t ws_cap = 0;        // planes
    uint8_t* d_blob = nullptr; size_t blob_cap = 0;    // tables / descriptors / uploaded bitstreams
    uint8_t* h_blob = nullptr; size_t h_blob_cap = 0;  // pinned mirror of d_blob
    uint32_t* h_status = nullptr; size_t h_status_cap = 0;
    int n = 0;
    std::vector<ParsedFrame> frames;
    std::vector<DevImage> imgs;          // host copies (device pointers inside)
    std::vector<int> parse_status;
    std::vector<std::string> parse_msg;
    std::vector<size_t> status_off;      // offset of each image's status words in the workspace
    DevImage* d_imgs = nullptr;
    bool pending = false;
    hipEvent_t lf_done = nullptr, hf_done = nullptr, done = nullptr;
    std::vector<Tap> taps;
    // timing: two event chains (LF stream, main stream)
    std::vector<std::string> stage_names;
    std::vector<int> stage_chain;
    std::vector<hipEvent_t> events;
    std::vector<float> stage_ms;
  };
  static constexpr int kSlots = 3;
  static constexpr int kPixelChunk = 32;   // frames that share one set of reconstruction / filter planes
  Slot slots[kSlots];
  int cur = 0, last = 0;
  Slot* active = &slots[0];
  hipStream_t stream_lf = nullptr;
  hipStream_t stream_hf = nullptr;
  hipStream_t last_stream = nullptr;
  // cumulative per-stage HIP-event time over every finished batch since the last reset (bench: timed region)
  std::vector<std::string> total_names;
  std::vector<double> total_ms;
  int total_batches = 0;
  std::string sticky_error;             // failure of an older, not yet reported batch
  int sticky_status = 0;
  // options
  // LoadImage's device / pinned result buffers, kept between calls of a thread (a 33 MB hipHostMalloc per call costs milliseconds)
  uint8_t* li_dev = nullptr; uint8_t* li_host = nullptr; size_t li_cap = 0;
  void EnsureLoadImageBuffers(size_t bytes);
  int lane_stride_override = 0;
  int hf_stride_override = 0;   // experiment knob: lane stride of the HF kernel only
  bool debug_taps = false;
  // band-restricted decode (multi-GPU sharding of one frame by group rows): 0 rows = whole frame
  int band_first_row = 0, band_rows = 0;
  bool no_stream_pairs = false;   // every fused frame through the four-pixels-per-lane filter kernel (parity tests: same output either way)
  bool no_direct = false, mod_lanes64 = false;   // launch shapes of the vector loops for small launches too (parity tests: same output either way)
  bool overlap = true;

  explicit JxlHipDecoder(int dev);
  ~JxlHipDecoder();
  void EnsureWs(size_t bytes);
  void EnsureBlob(size_t bytes);
  void Mark(const char* name, hipStream_t s, int chain);
  void WaitSlot(Slot& S);
  Slot& Last() { return slots[last]; }
  void Decode(int32_t n_, const uint8_t* const* host_data, const size_t* sizes, const uint8_t* const* dev_data,
              uint8_t* const* dev_out, hipStream_t stream, bool sync, DecoderStatus* statuses, ErrorInfo* err);
  DecoderStatus Finish(DecoderStatus* statuses, ErrorInfo* err);
  void CopyPlaneTap(int stage);
  void PrepassSingle(ParsedFrame& f, const uint8_t* dev_file);
};

JxlHipDecoder::JxlHipDecoder(int dev) {
  if (dev < 0) HIP_OK(hipGetDevice(&dev));
  device = dev;
  HIP_OK(hipSetDevice(device));
  // (measured on MI355X, batch 384: stream priorities and CU masks that confine the entropy streams to part of the chip change
  // nothing or lose - the three chains already add up to the chip's capacity)
  // JXLHIP_ENTROPY_CUS=N (experiments build): the two entropy streams may only use N of the 256 CUs (the mask's bits go round the
  // XCDs, so N / 8 per XCD); JXLHIP_PIXEL_CUS=M: this object's own stream keeps off the first 256 - M.
  int entropy_cus = 0, pixel_cus = 0;
  if (const char* e = Knob("JXLHIP_ENTROPY_CUS")) entropy_cus = std::min(256, std::max(0, atoi(e)));
  if (const char* e = Knob("JXLHIP_PIXEL_CUS")) pixel_cus = std::min(256, std::max(0, atoi(e)));
  auto masked_stream = [](hipStream_t* s, int first, int count) {
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = first; i < first + count; i++) mask[i >> 5] |= 1u << (i & 31);
    HIP_OK(hipExtStreamCreateWithCUMask(s, 8, mask));
  };
  if (pixel_cus) masked_stream(&own_stream, 256 - pixel_cus, pixel_cus);
  else HIP_OK(hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking));
  // The entropy chains are latency-bound (a kernel lasts as long as its longest section) and need few wavefronts, but ALL of them at
  // once: while their workgroups queue behind a pixel kernel's thousands of short ones, an entropy kernel runs in rounds and takes a
  // multiple of its time.  Their streams get the highest priority, so that their workgroups take the next free slots.
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  const int prio = Knob("JXLHIP_NO_PRIORITY") ? prio_least : prio_greatest;
  if (entropy_cus) {
    masked_stream(&stream_lf, 0, entropy_cus);
    masked_stream(&stream_hf, 0, entropy_cus);
  } else {
    HIP_OK(hipStreamCreateWithPriority(&stream_lf, hipStreamNonBlocking, prio));
    HIP_OK(hipStreamCreateWithPriority(&stream_hf, hipStreamNonBlocking, prio));
  }
  for (auto& S : slots) {
    HIP_OK(hipEventCreateWithFlags(&S.lf_done, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&S.hf_done, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
  }
  if (const char* e = Knob("JXLHIP_NO_OVERLAP")) overlap = atoi(e) == 0;
  const StaticTables& st = GetStaticTables();
  std::vector<float> all;
  for (int i = 0; i < 6; i++) all.insert(all.end(), st.basis[i].begin(), st.basis[i].end());
  HIP_OK(hipMalloc(&d_basis_all, all.size() * 4));
  HIP_OK(hipMemcpy(d_basis_all, all.data(), all.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> small;
  for (int c = 1; c <= 32; c *= 2)
    for (int k = 0; k < c; k++)
      for (int nn = 0; nn < c; nn++) small.push_back((float)((k ? std::sqrt(2.0) : 1.0) * std::cos((2 * nn + 1) * k * M_PI / (2.0 * c))));
  {
    // for lane (n, q) of a 16x16x4 product over an N-point transform: basis[(4 k + q) * N + n], k = 0 .. N/4-1, contiguous
    std::vector<float> t;
    for (int li = 2; li <= 3; li++) {   // N = 32, 64  (st.basis[li]: N = 8 << li)
      const int N = 8 << li;
      const std::vector<float>& b = st.basis[li];
      for (int nn = 0; nn < N; nn++)
        for (int q = 0; q < 4; q++)
          for (int k = 0; k < N / 4; k++) t.push_back(b[(size_t)(4 * k + q) * N + nn]);
    }
    HIP_OK(hipMalloc(&d_basis_mfma, t.size() * 4));
    HIP_OK(hipMemcpy(d_basis_mfma, t.data(), t.size() * 4, hipMemcpyHostToDevice));
  }
  HIP_OK(hipMalloc(&d_basis_small, small.size() * 4));
  HIP_OK(hipMemcpy(d_basis_small, small.data(), small.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMalloc(&d_llf_scale, st.llf_scale.size() * 4));
  HIP_OK(hipMemcpy(d_llf_scale, st.llf_scale.data(), st.llf_scale.size() * 4, hipMemcpyHostToDevice));
  for (int o = 0; o < kNumOrders; o++) {
    HIP_OK(hipMalloc(&d_natural[o], st.natural_order[o].size() * 2));
    HIP_OK(hipMemcpy(d_natural[o], st.natural_order[o].data(), st.natural_order[o].size() * 2, hipMemcpyHostToDevice));
  }
  for (int q = 0; q < kNumQuantTables; q++) {
    HIP_OK(hipMalloc(&d_dq[q], st.dq[q].size() * 4));
    HIP_OK(hipMemcpy(d_dq[q], st.dq[q].data(), st.dq[q].size() * 4, hipMemcpyHostToDevice));
    dq_n[q] = (uint32_t)(st.dq[q].size() / 3);
    std::vector<U32x2> sl;
    BuildScanList(q, nullptr, sl);
    HIP_OK(hipMalloc(&d_scan[q], sl.size() * sizeof(U32x2)));
    HIP_OK(hipMemcpy(d_scan[q], sl.data(), sl.size() * sizeof(U32x2), hipMemcpyHostToDevice));
  }
  lane_stride_override = KnobStride("JXLHIP_LANE_STRIDE", 0);
  hf_stride_override = KnobStride("JXLHIP_HF_STRIDE", 0);
}

JxlHipDecoder::~JxlHipDecoder() {
  (void)hipSetDevice(device);
  (void)hipDeviceSynchronize();
  for (auto& S : slots) {
    for (auto e : S.events) (void)hipEventDestroy(e);
    if (S.lf_done) (void)hipEventDestroy(S.lf_done);
    if (S.hf_done) (void)hipEventDestroy(S.hf_done);
    if (S.done) (void)hipEventDestroy(S.done);
    (void)hipFree(S.d_ws); (void)hipFree(S.d_blob);
    if (S.h_blob) (void)hipHostFree(S.h_blob);
    if (S.h_status) (void)hipHostFree(S.h_status);
  }
  (void)hipFree(d_basis_all); (void)hipFree(d_basis_small); (void)hipFree(d_basis_mfma); (void)hipFree(d_llf_scale);
  for (auto p : d_natural) (void)hipFree(p);
  for (auto p : d_dq) (void)hipFree(p);
  for (auto p : d_scan) (void)hipFree(p);
  if (li_dev) (void)hipFree(li_dev);
  if (li_host) (void)hipHostFree(li_host);
  if (own_stream) (void)hipStreamDestroy(own_stream);
  if (stream_lf) (void)hipStreamDestroy(stream_lf);
  if (stream_hf) (void)hipStreamDestroy(stream_hf);
}

void JxlHipDecoder::EnsureLoadImageBuffers(size_t bytes) {
  if (bytes <= li_cap) return;
  if (li_dev) { (void)hipFree(li_dev); li_dev = nullptr; }
  if (li_host) { (void)hipHostFree(li_host); li_host = nullptr; }
  li_cap = 0;
  HIP_OK(hipMalloc(&li_dev, bytes));
  HIP_OK(hipHostMalloc(&li_host, bytes, hipHostMallocDefault));
  li_cap = bytes;
}

void JxlHipDecoder::EnsureWs(size_t bytes) {
  auto& d_ws = active->d_ws; auto& ws_cap = active->ws_cap;
  if (bytes <= ws_cap) return;
  if (d_ws) HIP_OK(hipFree(d_ws));
  d_ws = nullptr;
  ws_cap = 0;
  size_t cap = bytes + bytes / 8;
  HIP_OK(hipMalloc(&d_ws, cap));
  ws_cap = cap;
}

void JxlHipDecoder::EnsureBlob(size_t bytes) {
  auto& d_blob = active->d_blob; auto& blob_cap = active->blob_cap; auto& h_blob = active->h_blob; auto& h_blob_cap = active->h_blob_cap;
  if (bytes > blob_cap) {
    if (d_blob) HIP_OK(hipFree(d_blob));
    d_blob = nullptr;
    blob_cap = 0;
    size_t cap = bytes + bytes / 4 + 4096;
    HIP_OK(hipMalloc(&d_blob, cap));
    blob_cap = cap;
  }
  if (bytes > h_blob_cap) {
    if (h_blob) HIP_OK(hipHostFree(h_blob));
    h_blob = nullptr;
    h_blob_cap = 0;
    size_t cap = bytes + bytes / 4 + 4096;
    HIP_OK(hipHostMalloc(&h_blob, cap, hipHostMallocDefault));
    h_blob_cap = cap;
  }
}

void JxlHipDecoder::Mark(const char* name, hipStream_t s, int chain) {
  Slot& S = *active;
  size_t i = S.stage_names.size();
  S.stage_names.push_back(name);
  S.stage_chain.push_back(chain);
  if (S.events.size() <= i) {
    hipEvent_t e;
    HIP_OK(hipEventCreate(&e));
    S.events.push_back(e);
  }
  HIP_OK(hipEventRecord(S.events[i], s));
}

// Waits for a submitted batch and folds a failure into the sticky error (reported by the next Finish).
void JxlHipDecoder::WaitSlot(Slot& S) {
  if (!S.pending) return;
  HIP_OK(hipEventSynchronize(S.done));
  S.pending = false;
  // a stage's time = from the previous mark of ITS chain (stream) to its own; marks of different chains may interleave in the list
  S.stage_ms.assign(S.stage_names.size(), -1.f);
  {
    int last_of[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    for (size_t i = 0; i < S.stage_names.size(); i++) {
      const int c = S.stage_chain[i] & 7;
      if (last_of[c] >= 0) { S.stage_ms[i] = 0.f; (void)hipEventElapsedTime(&S.stage_ms[i], S.events[last_of[c]], S.events[i]); }
      last_of[c] = (int)i;
    }
  }
  for (size_t i = 1; i < S.stage_names.size(); i++) {
    if (S.stage_ms[i] < 0.f) continue;   // the first mark of a chain: a start, not a stage
    size_t k = 0;
    while (k < total_names.size() && total_names[k] != S.stage_names[i]) k++;
    if (k == total_names.size()) { total_names.push_back(S.stage_names[i]); total_ms.push_back(0.0); }
    total_ms[k] += S.stage_ms[i];
  }
  total_batches++;
}

void JxlHipDecoder::Decode(int32_t n_, const uint8_t* const* host_data, const size_t* sizes, const uint8_t* const* dev_data,
                           uint8_t* const* dev_out, hipStream_t stream, bool sync, DecoderStatus* statuses, ErrorInfo* err) {
  HIP_OK(hipSetDevice(device));
  if (!stream) stream = own_stream;
  Slot& S = slots[cur];
  active = &S;
  if (S.pending) {
    // the slot is reused: its previous batch must be complete; remember a failure nobody has looked at yet
    WaitSlot(S);
    for (int i = 0; i < S.n; i++) {
      int st = S.parse_status[i];
      if (st == DecoderStatus_Ok && S.h_status[(size_t)i * 16]) st = DecoderStatus_DecodeError;
      if (st != DecoderStatus_Ok && sticky_status == DecoderStatus_Ok) { sticky_status = st; sticky_error = "an earlier asynchronous batch failed: " + S.parse_msg[i]; }
    }
  }
  auto& d_ws = S.d_ws; auto& d_blob = S.d_blob; auto& h_blob = S.h_blob; auto& h_status = S.h_status; auto& h_status_cap = S.h_status_cap;
  auto& n = S.n; auto& frames = S.frames; auto& imgs = S.imgs; auto& parse_status = S.parse_status; auto& parse_msg = S.parse_msg;
  auto& status_off = S.status_off; auto& d_imgs = S.d_imgs; auto& taps = S.taps; auto& stage_names = S.stage_names;
  hipStream_t s_lf = (overlap && !debug_taps) ? stream_lf : stream;
  hipStream_t s_hf = (overlap && !debug_taps) ? stream_hf : stream;
  n = n_;
  frames.assign(n, ParsedFrame());
  parse_status.assign(n, DecoderStatus_Ok);
  parse_msg.assign(n, "");
  // ---- 1. host parse (threaded across images)
  {
    int nt = std::min<int>(n, std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
    std::vector<std::thread> th;
    std::atomic<int> next(0);
    auto work = [&]() {
      for (;;) {
        int i = next.fetch_add(1);
        if (i >= n) return;
        try {
          ParseFile(host_data[i], sizes[i], false, frames[i]);
        } catch (const ParseError& e) {
          parse_status[i] = e.status;
          parse_msg[i] = e.what();
        } catch (const std::bad_alloc&) {
          parse_status[i] = DecoderStatus_OutOfMemory;
        } catch (const std::exception& e) {
          parse_status[i] = DecoderStatus_DecodeError;
          parse_msg[i] = e.what();
        }
      }
    };
    if (nt <= 1) work();
    else {
      for (int t = 0; t < nt; t++) th.emplace_back(work);
      for (auto& t : th) t.join();
    }
  }
  // frames that fit one group: their HfGlobal can only be located once the LF group has been decoded (one bit stream)
  for (int i = 0; i < n; i++) {
    if (parse_status[i] != DecoderStatus_Ok || !frames[i].single || frames[i].encoding != 0) continue;
    try {
      PrepassSingle(frames[i], dev_data ? dev_data[i] : nullptr);
    } catch (const ParseError& e) {
      parse_status[i] = e.status;
      parse_msg[i] = e.what();
    } catch (const std::exception& e) {
      parse_status[i] = DecoderStatus_DecodeError;
      parse_msg[i] = e.what();
    }
  }
  // images that failed to parse are skipped on the device (their DevImage stays zeroed, ng = 0)
  // ---- 2. layout of blob and workspace
  Bump blob, ws_zero, ws;
  // every pass after the first of a progressive frame is an image record of its own, after the batch's n (dev_types.h: next_pass)
  int n_extra = 0;
  std::vector<int> first_extra((size_t)n, 0);
  for (int i = 0; i < n; i++) {
    first_extra[i] = n + n_extra;
    if (parse_status[i] == DecoderStatus_Ok && frames[i].encoding == 0) n_extra += (int)frames[i].extra_passes.size();
  }
  const size_t off_imgs = blob.Take(sizeof(DevImage) * (size_t)(n + n_extra));
  struct PassLayout {   // what a pass owns: its code, scan lists, entry lists, block index, end positions, LZ77 windows
    size_t a_cmap, a_cfg, a_alias, a_pfx[3] = {}, lz_hf = 0, scan[kNumQuantTables] = {}, centries, cblk, bitpos, hf_order;
  };
  struct PerImg {
    std::vector<PassLayout> extra;
    size_t m_direct = 0;   // + 1 when present
    size_t sec_off, sec_size, tree, m_cmap, m_cfg, m_alias, a_cmap, a_cfg, a_alias, order[kNumOrders][3], cs;
    size_t m_pfx[3] = {}, a_pfx[3] = {};           // prefix codes: counts, symbol offsets, sorted symbols
    size_t lz_lf = 0, lz_grp = 0, lz_hf = 0, lz_mod = 0;   // LZ77 windows (+1; 0: none)
    size_t scan[kNumQuantTables] = {};   // frames with their own coefficient orders: scan lists of the affected quant tables
    size_t trc_lut = 0;                  // tone-curve tables of an evaluated ICC profile (+1; 0: none)
    size_t dq[kNumQuantTables] = {};     // frames with their own dequantisation tables (+1; 0: the library table)
    size_t z_cellinfo, z_status, centries, cblk, hf_order;
    std::vector<size_t> mod_planes;
    size_t mod_chan, mod_desc, wp_lf, wp_grp, lf_end, alpha32;
    size_t lf[3], lf_tmp[3], lfq[3], lf_extra, rawq, sharp, ytox, ytob, binfo, lf_desc, lf_count, alpha_desc, blk_list, blk_count, bitpos, tile_list, tmp[3], xyb[3], inv_sigma, alpha;
    size_t orient_tmp = 0;   // frames with an orientation other than 1 are decoded here, then laid out as displayed in the caller's buffer
  };
  std::vector<PerImg> L((size_t)n);
  // the images' status words (64 B each) lie side by side: ONE copy brings them back (a copy per image was 384 five-microsecond copy
  // kernels at the end of the pixel stream - 2 ms of the step - and as many API calls)
  const size_t z_status_base = ws_zero.Take((size_t)std::max(1, n) * 64);
  int total_lf = 0, total_groups = 0, n_mod_tasks = 0;
  // Modular frames whose MA tree looks at decoded neighbours take the generic per-lane path: give it LDS row buffers (groups of up to
  // 256 columns); with the weighted predictor its per-sample state goes to LDS as well, which limits a workgroup to 8 sections
  int mod_lanes = 64, mod_rb = 0, mod_wp_lds = 0;
  size_t total_mod_sections = 0;
  for (int i = 0; i < n; i++) {
    if (parse_status[i] != DecoderStatus_Ok || frames[i].encoding != 1) continue;
    if (!frames[i].tree_row_static && frames[i].group_dim <= 256) mod_rb = 256;
    if (frames[i].tree_uses_wp && frames[i].group_dim <= 256) { mod_wp_lds = 1; mod_lanes = 8; }
    total_mod_sections += frames[i].single ? 1 : 1 + (size_t)frames[i].nlf + frames[i].ng;
  }
  if (!mod_rb) { mod_wp_lds = 0; mod_lanes = 64; }
  // few sections (one frame, a small batch): one section per wavefront - no divergence between sections, and row-static channels
  // decode on the scalar unit from per-residue tables (see the LF launch below)
  if (total_mod_sections <= 512 && !mod_lanes64 && !Knob("JXLHIP_MOD_LANES64")) mod_lanes = 1;
  // Small launches get the Modular code's per-residue tables (the alias tables spelled out for each of the 4096 state residues,
  // 16 KB per cluster, codes of up to 8 clusters): one-section wavefronts read them through the scalar cache (RowScalar).  Measured
  // against a copy in LDS (one 4K frame): lf_ans 20.2 -> 18.8 ms, alpha_ans 5.7 -> 5.1 ms, and no LDS spent on them.
  bool global_direct = false;
  if (!no_direct && !Knob("JXLHIP_NO_DIRECT")) {
    int pre_lf = 0;
    for (int i = 0; i < n; i++) if (parse_status[i] == DecoderStatus_Ok && frames[i].encoding == 0) pre_lf += (int)frames[i].nlf;
    global_direct = n <= 64 && pre_lf <= 1024 && total_mod_sections <= 512;
  }
  size_t chunk_pix = 0;   // padded pixels of the largest VarDCT frame of the batch
  for (int i = 0; i < n; i++) {
    if (parse_status[i] != DecoderStatus_Ok) continue;
    const ParsedFrame& f = frames[i];
    PerImg& l = L[i];
    const size_t cells = (size_t)f.w8 * f.h8, pix = cells * 64, tiles = (size_t)((f.w8 + 7) / 8) * ((f.h8 + 7) / 8);
    l.sec_off = blob.Take(8 * f.sec_off.size());
    l.sec_size = blob.Take(4 * f.sec_size.size());
    l.hf_order = blob.Take(4 * (size_t)std::max<uint32_t>(1, f.ng));
    l.tree = blob.Take(sizeof(DevTreeNode) * f.tree.size());
    l.m_cmap = blob.Take(f.mcode.ctx_map.size());
    l.m_cfg = blob.Take(4 * f.mcode.cfg.size());
    l.m_alias = blob.Take(8 * f.mcode.alias.size());
    if (global_direct && !f.mcode.use_prefix && !f.mcode.lz77 && f.mcode.num_hist <= 8) l.m_direct = blob.Take((size_t)f.mcode.num_hist << 14) + 1;
    auto pfx_layout = [&](const HostCode& hc, size_t* o) {
      if (!hc.use_prefix) return;
      size_t total = 0;
      for (auto& pc : hc.prefix) total += pc.sorted.size();
      o[0] = blob.Take(2 * 16 * hc.prefix.size()); o[1] = blob.Take(4 * hc.prefix.size()); o[2] = blob.Take(2 * std::max<size_t>(1, total));
    };
    pfx_layout(f.mcode, l.m_pfx);
    if (f.encoding == 1) {
      // Modular (lossless) frame: whole-image int32 channel planes, no VarDCT workspace
      const bool resident_m = dev_data && dev_data[i] && f.cs_contiguous;
      l.cs = resident_m ? 0 : blob.Take(f.cs_size + 16);
      l.z_status = z_status_base + (size_t)i * 64;
      for (auto& pl : f.mod_planes) l.mod_planes.push_back(ws.Take(4 * (size_t)std::max(1, pl.w) * std::max(1, pl.h)));
      const size_t nsec = 1 + (size_t)f.nlf + f.ng;
      l.mod_chan = blob.Take(sizeof(ModChanDev) * f.mod_coded.size());
      l.mod_desc = ws.Take(nsec * f.mod_coded.size() * sizeof(ChanDesc));
      if (f.tree_uses_wp) l.wp_grp = ws.Take(nsec * 10 * (f.group_dim + 2) * 4);
      if (f.mcode.lz77) l.lz_mod = ws.Take(nsec * ((size_t)4 << 20)) + 1;
      n_mod_tasks += f.single ? 1 : ((int)nsec + mod_lanes - 1) / mod_lanes;
      continue;
    }
    l.a_cmap = blob.Take(f.acode.ctx_map.size());
    l.a_cfg = blob.Take(4 * f.acode.cfg.size());
    l.a_alias = blob.Take(8 * f.acode.alias.size());
    pfx_layout(f.acode, l.a_pfx);
    if (f.mcode.lz77) { l.lz_lf = ws.Take((size_t)f.nlf * ((size_t)4 << 20)) + 1; l.lz_grp = ws.Take((size_t)f.ng * ((size_t)4 << 16)) + 1; }
    if (f.acode.lz77) l.lz_hf = ws.Take((size_t)f.ng * ((size_t)4 << 18)) + 1;
    for (int o = 0; o < kNumOrders; o++)
      for (int c = 0; c < 3; c++) l.order[o][c] = f.custom_order[o][c].empty() ? 0 : blob.Take(2 * f.custom_order[o][c].size());
    for (int q = 0; q < kNumQuantTables; q++) {
      const int o = OrderBucketOfQuantTable(q);
      const bool own_dq = !f.dq_default && f.custom_dq[q].size() == 3 * (size_t)dq_n[q];
      if (own_dq) l.dq[q] = blob.Take(4 * 3 * (size_t)dq_n[q], 256) + 1;
      if (own_dq || !f.custom_order[o][0].empty() || !f.custom_order[o][1].empty() || !f.custom_order[o][2].empty()) l.scan[q] = blob.Take(8 * 3 * (size_t)dq_n[q], 256) + 1;
    }
    const bool resident = dev_data && dev_data[i] && f.cs_contiguous;
    l.cs = resident ? 0 : blob.Take(f.cs_size + 16);
    if (PlanColor(f).transfer == 5) l.trc_lut = blob.Take(4 * 3 * 4096, 256) + 1;
    l.z_cellinfo = ws_zero.Take(4 * cells);
    l.z_status = z_status_base + (size_t)i * 64;
    {
      // entry lists of the decoded group rows only (a band decode touches a band's worth), block index for the whole cell grid
      int b0 = 0, b1 = (int)f.yg;
      if (band_rows > 0) { b0 = std::min<int>(band_first_row, (int)f.yg); b1 = std::min<int>(b0 + band_rows, (int)f.yg); }
      const int g0 = std::max(0, b0 - 1), g1 = std::min<int>((int)f.yg, b1 + 1);
      l.centries = ws.Take((size_t)std::max(1, g1 - g0) * f.xg * kGroupEntriesCap * 4);
      l.cblk = ws.Take(3 * cells * sizeof(U32x2));
      for (auto& ep : f.extra_passes) {
        PassLayout pl;
        pl.a_cmap = blob.Take(ep.acode.ctx_map.size());
        pl.a_cfg = blob.Take(4 * ep.acode.cfg.size());
        pl.a_alias = blob.Take(8 * ep.acode.alias.size());
        pfx_layout(ep.acode, pl.a_pfx);
        if (ep.acode.lz77) pl.lz_hf = ws.Take((size_t)f.ng * ((size_t)4 << 18)) + 1;
        for (int q = 0; q < kNumQuantTables; q++) {
          const int o = OrderBucketOfQuantTable(q);
          if (l.dq[q] || !ep.custom_order[o][0].empty() || !ep.custom_order[o][1].empty() || !ep.custom_order[o][2].empty()) pl.scan[q] = blob.Take(8 * 3 * (size_t)dq_n[q], 256) + 1;
        }
        pl.centries = ws.Take((size_t)std::max(1, g1 - g0) * f.xg * kGroupEntriesCap * 4);
        pl.cblk = ws.Take(3 * cells * sizeof(U32x2));
        pl.bitpos = ws.Take((size_t)f.ng * 8);
        pl.hf_order = blob.Take(4 * (size_t)std::max<uint32_t>(1, f.ng));
        l.extra.push_back(pl);
      }
    }
    for (int c = 0; c < 3; c++) { l.lf[c] = ws.Take(4 * cells); l.lf_tmp[c] = ws.Take(4 * cells); l.lfq[c] = ws.Take(4 * cells); }
    l.lf_extra = ws.Take(f.nlf);
    l.rawq = ws.Take(2 * cells);
    l.sharp = ws.Take(cells);
    l.ytox = ws.Take(tiles);
    l.ytob = ws.Take(tiles);
    l.binfo = ws.Take((size_t)f.nlf * kBinfoInts * 4);
    l.lf_desc = ws.Take((size_t)f.nlf * 8 * sizeof(ChanDesc));
    l.lf_count = ws.Take((size_t)f.nlf * 4);
    l.alpha_desc = ws.Take((size_t)f.ng * sizeof(ChanDesc));
    l.blk_list = ws.Take((size_t)f.ng * 1024 * 4);
    l.blk_count = ws.Take((size_t)f.ng * 4);
    l.bitpos = ws.Take((size_t)f.ng * 8);
    l.tile_list = ws.Take(tiles * 4);
    l.alpha32 = ws.Take(4 * (size_t)f.xsize * f.ysize);
    chunk_pix = std::max(chunk_pix, pix);
    l.inv_sigma = ws.Take(4 * cells);
    l.alpha = ws.Take((size_t)f.xsize * f.ysize * OutBytesPerSample(f));
    l.lf_end = ws.Take(8);
    if (f.tree_uses_wp) { l.wp_lf = ws.Take((size_t)f.nlf * kWpLfInts * 4); l.wp_grp = ws.Take((size_t)f.ng * 10 * (kGroupDim + 2) * 4); }
    {
      // sections this call decodes (a band: its group rows + one each side, the LF groups they touch): what the launch shapes go by
      int b0 = 0, b1 = (int)f.yg;
      if (band_rows > 0) { b0 = std::min<int>(band_first_row, (int)f.yg); b1 = std::min<int>(b0 + band_rows, (int)f.yg); }
      const int g0 = std::max(0, b0 - 1), g1 = std::min<int>((int)f.yg, b1 + 1);
      total_lf += ((g1 + 7) / 8 - g0 / 8) * (int)f.xlf;
      total_groups += (g1 - g0) * (int)f.xg * (int)f.num_passes;
    }
  }
  // The float planes between reconstruction and the loop filters (24 B/px) are only alive while a frame is in the pixel stages:
  // frames go through those stages in chunks that share kPixelChunk sets of planes, so the batch size is bounded by the
  // entropy-stage state (12 B/px of coefficients), not by 36 B/px.
  const int pixel_chunk = debug_taps ? std::max(1, n) : std::min(std::max(1, n), kPixelChunk);
  // (a third set: the loop-filter ping-pong planes, which double as the dense coefficient planes of the generic path)
  std::vector<size_t> chunk_tmp((size_t)pixel_chunk * 3), chunk_xyb((size_t)pixel_chunk * 3), chunk_coef((size_t)pixel_chunk * 3);
  if (chunk_pix)
    for (int k = 0; k < pixel_chunk * 3; k++) { chunk_tmp[k] = ws.Take(4 * chunk_pix); chunk_xyb[k] = ws.Take(4 * chunk_pix); chunk_coef[k] = ws.Take(4 * chunk_pix); }
  // The reference's decoder library hands out the image as displayed (orientation applied; it is only kept when the caller asks,
  // which Decoder/DecoderContext.cpp never does): such frames are decoded into a scratch buffer and re-laid out at the end.
  // A band is a set of VarDCT group rows.  A Modular frame has no band mode (global Squeeze / whole-image transforms: SURVEY 8e
  // "replicas only"); its output kernel writes the whole frame, so accepting the option would overrun the caller's band buffer.
  for (int i = 0; i < n; i++)
    if (parse_status[i] == DecoderStatus_Ok && band_rows > 0 && (band_first_row < 0 || band_first_row >= (int)frames[i].yg)) {
      parse_status[i] = DecoderStatus_DecodeError;
      parse_msg[i] = "band decode: the first group row lies outside the frame";
    }
  for (int i = 0; i < n; i++)
    if (parse_status[i] == DecoderStatus_Ok && band_rows > 0 && frames[i].encoding == 1) {
      parse_status[i] = DecoderStatus_DecodeError;
      parse_msg[i] = "band decode of a Modular (lossless) frame is not supported";
    }
  for (int i = 0; i < n; i++) {
    if (parse_status[i] != DecoderStatus_Ok || frames[i].orientation == 1) continue;
    const ParsedFrame& f = frames[i];
    if (band_rows > 0) { parse_status[i] = DecoderStatus_DecodeError; parse_msg[i] = "band decode of a frame with an orientation is not supported"; continue; }
    L[i].orient_tmp = ws.Take((size_t)f.xsize * f.ysize * (f.ncolor + (f.black_index >= 0 ? 1 : 0) + (f.alpha_index >= 0 ? 1 : 0)) * OutBytesPerSample(f));
  }
  // Lane mapping of the HF kernel: one wavefront per section while every workgroup of the launch can be resident at once
  // (the kernel is latency-bound, a second round of workgroups doubles its time); otherwise pack more sections per wavefront.
  int lane_stride = 64;
  int hf_wg_capacity = 256 * 8;   // workgroups of the HF kernel that can be resident at once (by the LDS of the widest tables)
  if (lane_stride_override > 0) lane_stride = lane_stride_override;
  else {
    size_t lds_est = 0;
    for (int i = 0; i < n; i++)
      if (parse_status[i] == DecoderStatus_Ok && frames[i].encoding == 0)
        lds_est = std::max(lds_est, 8 + 8 * frames[i].acode.alias.size() + 4 * frames[i].acode.cfg.size() + frames[i].acode.ctx_map.size() + 64 + 4 * (96 + 64 + 128));
    const int wg_per_cu = lds_est ? (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds_est)) : 8;
    const int capacity = 256 * wg_per_cu;   // resident 256-thread workgroups on the chip
    hf_wg_capacity = capacity;
    while (lane_stride > 1 && (total_groups + (256 / lane_stride) - 1) / (256 / lane_stride) > capacity) lane_stride >>= 1;
    // measured (MI355X, 4K frames, batch 384): once a batch holds thousands of sections, 32 sections per wavefront
    // (half-filled wavefronts, five per image instead of three) is the best trade between instruction efficiency and wavefronts
    // in flight: hf_decode 72 ms (stride 1) / 61 ms (stride 2) / 87 ms (stride 4)
    if (total_groups >= 8192) lane_stride = 2;
    if (hf_stride_override > 0) lane_stride = hf_stride_override;
  }
  // Workgroup width of the HF kernel: four wavefronts, or up to eight when the sections are spread thinly over the lanes and one
  // image's sections would otherwise need a second workgroup (each workgroup stages the image's ~50 KB of tables in LDS).
  int hf_waves = 4;
  {
    int max_ng = 0;
    for (int i = 0; i < n; i++)
      if (parse_status[i] == DecoderStatus_Ok && frames[i].encoding == 0) max_ng = std::max<int>(max_ng, (int)frames[i].ng);
    const int per_wave = 64 / lane_stride;
    const int need = (max_ng + per_wave - 1) / per_wave;
    if (need > 4 && lane_stride <= 8) hf_waves = std::min(8, need);
    if (lane_stride == 64 && !Knob("JXLHIP_HF_WAVES8")) {
      // One section per wavefront: its token loop runs on the scalar unit, and a CU has ONE scalar unit - eight such wavefronts in
      // a workgroup share it (measured: hf_decode of one 4K frame 12.5 ms with 17 workgroups of 8 wavefronts).  Spread the sections
      // over as many workgroups as can be resident at once, one wavefront each if they all fit.
      hf_waves = 1;
      while (hf_waves < 8 && (total_groups + hf_waves - 1) / hf_waves > hf_wg_capacity) hf_waves *= 2;
    }
  }
  const int per_wg = hf_waves * (64 / lane_stride);
  // Sections per workgroup, per image: the HF kernel's LDS is the image's code tables (30 .. 60 KB: they double with the alias-table
  // width) plus 288 B per lane, and a launch has ONE LDS size.  Sized by the batch-wide maximum, a single image with wide tables
  // pushed every workgroup from two per CU to one (hf_decode 30 -> 58 ms at batch 384); instead every image gets as many lanes per
  // workgroup as fit beside ITS tables in the budget (whole wavefronts; images with wide tables use more, smaller workgroups).
  // The budget (half a CU's LDS) was re-measured in round 3 against 96 / 112 / 128 KB (one workgroup per 4K frame, one copy of its
  // tables): those won 6 % of the pipelined step while the fused filter kernel held 252 registers, and nothing since it holds 122
  // (profiles/r03_experiment_hf_lds_budget.txt, r03_experiment_pairs_balance.txt); alone, the HF kernel is 36 ms with 80 KB and 65 ms
  // with 112 KB (two rounds of workgroups), so 80 KB it stays.
  auto hf_code_bytes = [](const HostCode& c) { return 8 + 8 * c.alias.size() + 4 * c.cfg.size() + c.ctx_map.size() + 64 + 32; };
  auto hf_table_bytes = [&](const ParsedFrame& f) {   // the widest of the frame's passes
    size_t b = hf_code_bytes(f.acode);
    for (auto& ep : f.extra_passes) b = std::max(b, hf_code_bytes(ep.acode));
    return b;
  };
  size_t kHfLdsTarget = 80 * 1024;
  if (const char* e = Knob("JXLHIP_HF_LDS_KB")) { const int kb = atoi(e); if (kb >= 32 && kb <= 160) kHfLdsTarget = (size_t)kb * 1024; }   // experiment knob
  auto hf_per_wg = [&](const ParsedFrame& f) {
    // the fewest workgroups whose (tables + lanes) fit the budget, the image's sections spread evenly over them: every workgroup
    // carries a copy of the tables, so a small last workgroup (64 + 64 + 7 sections) costs a full LDS slot for a few lanes
    const int per_wave = 64 / lane_stride;
    const size_t tab = hf_table_bytes(f);
    const int ng = std::max(1, (int)f.ng);
    for (int nwg = 1; nwg <= ng; nwg++) {
      const int lanes = (((ng + nwg - 1) / nwg) + per_wave - 1) / per_wave * per_wave;
      if (lanes <= per_wg && tab + HfLaneLdsBytes(32) * (size_t)lanes <= kHfLdsTarget) return lanes;
      if (lanes <= per_wave) break;
    }
    return per_wave;
  };
  int n_pass_wg = 0;
  for (int i = 0; i < n; i++)
    if (parse_status[i] == DecoderStatus_Ok && frames[i].encoding == 0)
      n_pass_wg += (int)frames[i].num_passes * (((int)frames[i].ng + hf_per_wg(frames[i]) - 1) / hf_per_wg(frames[i]));
  const size_t off_lf_tasks = blob.Take(sizeof(SectionTask) * (size_t)std::max(1, total_lf));
  const size_t off_pass_tasks = blob.Take(sizeof(SectionTask) * (size_t)std::max(1, n_pass_wg));
  // Lane mapping of the alpha phase-A kernel (one wavefront per workgroup, sections of one image per wavefront): spread the
  // sections over as many wavefronts as the chip holds in one round, then pack.
  int alpha_stride = 64;
  if (lane_stride_override > 0) alpha_stride = lane_stride_override;
  else {
    int alpha_sections = 0;
    for (int i = 0; i < n; i++)
      if (parse_status[i] == DecoderStatus_Ok && frames[i].encoding == 0 && frames[i].alpha_index >= 0) {
        int b0 = 0, b1 = (int)frames[i].yg;   // a band decodes the alpha of its own group rows only
        if (band_rows > 0) { b0 = std::min<int>(band_first_row, b1); b1 = std::min<int>(b0 + band_rows, b1); }
        alpha_sections += (b1 - b0) * (int)frames[i].xg;
      }
    // Two good shapes and a bad middle (measured, 4K frames): one section per wavefront on the scalar unit (5 ms for 135 sections,
    // degrading gently while a CU holds a dozen such wavefronts) and 32 sections per wavefront on the vector unit (12 ms for 8640
    // sections); 2 ... 16 sections per wavefront pay the vector chain for a few lanes (21 ms for 2160 sections at 2 per wavefront).
    if (global_direct && !Knob("JXLHIP_ALPHA_OLD_SHAPES")) alpha_stride = alpha_sections > 3072 ? 2 : 64;
    else while (alpha_stride > 1 && alpha_sections / (64 / alpha_stride) > 256 * 8) alpha_stride >>= 1;
    if (alpha_sections >= 8192) alpha_stride = 2;   // measured: alpha_ans 27.8 ms (stride 1) / 22.5 (2) / 26.4 (4) at batch 384
    alpha_stride = KnobStride("JXLHIP_ALPHA_STRIDE", alpha_stride);   // experiment knob
  }
  const int per_alpha_wg = 64 / alpha_stride;
  int n_alpha_wg = 0;
  for (int i = 0; i < n; i++)
    if (parse_status[i] == DecoderStatus_Ok && frames[i].encoding == 0) n_alpha_wg += ((int)frames[i].ng + per_alpha_wg - 1) / per_alpha_wg;
  // LF groups per wavefront of the LF phase-A kernel.  Measured on MI355X: the 64 LF groups of a 16384^2 frame in ONE wavefront (every
  // lane walking its own divergent stream) took 68 ms against 33 ms for the four of a 4K frame; one group per wavefront is no faster
  // for a single frame and much slower for a batch (384 frames: 44 -> 67 ms, four times the wavefronts for the same tokens).  So: four
  // per wavefront, more only when a batch brings tens of thousands of LF groups.
  // LF sections per wavefront: one while the batch is small (the wavefront's recurrence then runs on the scalar unit: lower latency),
  // four for large batches (fewer wavefronts and table copies for the same latency-bound time), more only for huge ones
  int lf_per_wave = total_lf <= 256 ? 1 : 4;
  while (lf_per_wave < 64 && total_lf / lf_per_wave > 4096) lf_per_wave *= 2;
  if (const char* e = Knob("JXLHIP_LF_PER_WAVE")) lf_per_wave = std::min(64, std::max(1, atoi(e)));   // experiment knob
  int n_lf_ans = 0;
  for (int i = 0; i < n; i++)
    if (parse_status[i] == DecoderStatus_Ok && frames[i].encoding == 0) n_lf_ans += ((int)frames[i].nlf + lf_per_wave - 1) / lf_per_wave;
  const size_t off_lf_ans_tasks = blob.Take(sizeof(SectionTask) * (size_t)std::max(1, n_lf_ans));
  const size_t off_mod_tasks = blob.Take(sizeof(SectionTask) * (size_t)std::max(1, n_mod_tasks));
  const size_t off_alpha_tasks = blob.Take(sizeof(SectionTask) * (size_t)std::max(1, n_alpha_wg));
  const size_t zero_bytes = Align(ws_zero.off, 256);
  EnsureBlob(blob.off);
  EnsureWs(zero_bytes + ws.off);
  if ((size_t)n * 16 > h_status_cap) {
    if (h_status) HIP_OK(hipHostFree(h_status));
    h_status = nullptr;
    HIP_OK(hipHostMalloc(&h_status, (size_t)n * 16 * 4 + 64, hipHostMallocDefault));
    h_status_cap = (size_t)n * 16;
  }
  // ---- 3. fill the pinned blob
  memset(h_blob, 0, blob.off);
  imgs.assign(n + n_extra, DevImage());
  for (auto& im : imgs) memset(&im, 0, sizeof(DevImage));
  status_off.assign(n, 0);
  size_t lds_hf = 0, lds_hf_lanes = 0, lds_lf = 0, lds_alpha = 0;   // lds_hf: tables + lanes, the largest workgroup; lds_hf_lanes: the most lanes (global-table variant)
  bool any_alpha = false, any_unfiltered = false;
  int stage_mask = 0;   // LDS-tiled loop-filter stage kernels some frame of the batch needs (bit s: filter_tile_kernel<s>)
  int any_fused = 0, any_fused2 = 0;   // 1: fused frames (with a second iteration) of the two-pixels-per-lane kernels, 2: others
  int max_w = 1, max_h = 1, max_tiles = 1;
  auto tiles_of = [](const ParsedFrame& f) { return (size_t)((f.w8 + 7) / 8) * ((f.h8 + 7) / 8); };
  size_t max_cells = 1, max_padded = 8;
  SectionTask* lf_tasks = (SectionTask*)(h_blob + off_lf_tasks);
  SectionTask* pass_tasks = (SectionTask*)(h_blob + off_pass_tasks);
  SectionTask* alpha_tasks = (SectionTask*)(h_blob + off_alpha_tasks);
  SectionTask* lf_ans_tasks = (SectionTask*)(h_blob + off_lf_ans_tasks);
  SectionTask* mod_tasks = (SectionTask*)(h_blob + off_mod_tasks);
  int nlf_t = 0, npass_t = 0, nalpha_t = 0, nlf_ans_t = 0, nmod_t = 0, max_groups = 1, max_mod_groups = 1;
  size_t lds_mod = 0, max_mod_pixels = 1;
  int max_mod_coded = 1;
  struct ModLaunch { int kind; int32_t *a, *b, *c; int aw, ah, rw, rh, type; int32_t* out[4]; int nout; uint32_t* status; };
  std::vector<ModLaunch> mod_ops;
  uint8_t* wz = d_ws;
  uint8_t* wr = d_ws + zero_bytes;
  for (int i = 0; i < n; i++) {
    memset(&imgs[i], 0, sizeof(DevImage));
    if (parse_status[i] != DecoderStatus_Ok) continue;
    const ParsedFrame& f = frames[i];
    const PerImg& l = L[i];
    DevImage& d = imgs[i];
    d.w = f.xsize; d.h = f.ysize; d.w8 = f.w8; d.h8 = f.h8; d.wp = f.w8 * 8; d.hp = f.h8 * 8;
    d.wt = (f.w8 + 7) / 8; d.ht = (f.h8 + 7) / 8;
    d.xg = f.xg; d.yg = f.yg; d.ng = f.ng; d.xlf = f.xlf; d.ylf = f.ylf; d.nlf = f.nlf;
    d.ncolor = f.ncolor; d.has_alpha = f.alpha_index >= 0; d.nch_out = d.ncolor + d.has_alpha;
    d.sample_bits = (int32_t)f.bits; d.sample_exp = (int32_t)f.exp_bits;
    d.alpha_bits = d.has_alpha ? (int32_t)f.ec[f.alpha_index].bits : 8; d.alpha_exp = d.has_alpha ? (int32_t)f.ec[f.alpha_index].exp_bits : 0;
    d.out_bits = 8 * (int32_t)OutBytesPerSample(f); d.out_float = f.exp_bits ? 1 : 0;
    d.unpremultiply = (d.has_alpha && f.ec[f.alpha_index].alpha_associated) ? 1 : 0;
    d.alpha_unit = d.alpha_exp ? 1.0f : 1.0f / (float)((1u << d.alpha_bits) - 1);
    // band: group rows [b0, b1) are output; one more row each side is decoded for the loop-filter halo
    int b0 = 0, b1 = (int)f.yg;
    if (band_rows > 0 && f.encoding == 0) { b0 = std::min<int>(band_first_row, (int)f.yg); b1 = std::min<int>(b0 + band_rows, (int)f.yg); }
    d.dec_gy0 = std::max(0, b0 - 1); d.dec_gy1 = std::min<int>((int)f.yg, b1 + 1);
    d.band_y0 = std::min<int>(b0 * kGroupDim, (int)f.ysize); d.band_y1 = std::min<int>(b1 * kGroupDim, (int)f.ysize);
    const ColorPlan plan = PlanColor(f);
    d.to_srgb = plan.transfer;   // 0 linear, 1 sRGB, 2 BT.709, 3 PQ, 5 tables
    if (plan.transfer == 5 && l.trc_lut) {
      memcpy(h_blob + l.trc_lut - 1, plan.trc_lut.data(), 4 * 3 * 4096);
      d.trc_lut = (const float*)(d_blob + l.trc_lut - 1);
    }
    d.pq_scale = f.intensity_target * 1e-4f;
    auto put = [&](size_t off, const void* src, size_t bytes) { if (bytes) memcpy(h_blob + off, src, bytes); };
    put(l.sec_off, f.sec_off.data(), 8 * f.sec_off.size());
    put(l.sec_size, f.sec_size.data(), 4 * f.sec_size.size());
    put(l.tree, f.tree.data(), sizeof(DevTreeNode) * f.tree.size());
    auto code = [&](const HostCode& hc, size_t cm, size_t cf, size_t al, DevCode& dc, const size_t* pfx = nullptr) {
      put(cm, hc.ctx_map.data(), hc.ctx_map.size());
      std::vector<uint32_t> cfgp(hc.cfg.size());
      for (size_t k = 0; k < hc.cfg.size(); k++) {
        cfgp[k] = hc.cfg[k].split | hc.cfg[k].msb << 4 | hc.cfg[k].lsb << 8;
        if (hc.use_prefix) {   // one-symbol prefix codes read no bits
          if (hc.prefix[k].single >= 0) cfgp[k] |= 1u << 12 | (uint32_t)hc.prefix[k].single << 16;
          continue;
        }
        // single-symbol clusters: decoding never changes the ANS state nor reads bits (alias special form)
        const uint64_t e0 = hc.alias[k << hc.log_alpha];
        const uint32_t x0 = (uint32_t)e0, y0 = (uint32_t)(e0 >> 32);
        if ((x0 >> 16) == 0 && (y0 >> 16) == 4096) cfgp[k] |= 1u << 12 | ((x0 >> 8) & 0xFF) << 16;
      }
      dc.slow = (hc.use_prefix ? 1u : 0u) | (hc.lz77 ? 2u : 0u);
      if (hc.use_prefix && pfx) {
        std::vector<uint16_t> counts(16 * hc.prefix.size(), 0), sorted;
        std::vector<uint32_t> offs(hc.prefix.size(), 0);
        for (size_t k = 0; k < hc.prefix.size(); k++) {
          memcpy(&counts[16 * k], hc.prefix[k].count, 32);
          offs[k] = (uint32_t)sorted.size();
          sorted.insert(sorted.end(), hc.prefix[k].sorted.begin(), hc.prefix[k].sorted.end());
        }
        put(pfx[0], counts.data(), 2 * counts.size()); put(pfx[1], offs.data(), 4 * offs.size()); put(pfx[2], sorted.data(), 2 * sorted.size());
        dc.pfx_count = (const uint16_t*)(d_blob + pfx[0]); dc.pfx_off = (const uint32_t*)(d_blob + pfx[1]); dc.pfx_sorted = (const uint16_t*)(d_blob + pfx[2]);
      }
      if (hc.lz77) {
        dc.lz_min_symbol = hc.lz_min_symbol; dc.lz_min_length = hc.lz_min_length;
        dc.lz_len_cfg = hc.lz_len.split | hc.lz_len.msb << 4 | hc.lz_len.lsb << 8;
        dc.lz_dist_cluster = hc.ctx_map.back();
      }
      put(cf, cfgp.data(), 4 * cfgp.size());
      put(al, hc.alias.data(), 8 * hc.alias.size());
      dc.ctx_map = d_blob + cm;
      dc.cfg = (const uint32_t*)(d_blob + cf);
      dc.alias = (const uint64_t*)(d_blob + al);
      dc.num_ctx = (uint32_t)hc.ctx_map.size();
      dc.num_clusters = hc.num_hist;
      dc.log_alpha = hc.log_alpha;
      dc.direct = nullptr;
    };
    code(f.mcode, l.m_cmap, l.m_cfg, l.m_alias, d.mcode, l.m_pfx);
    if (l.m_direct) {   // the alias tables spelled out per state residue
      uint32_t* dt = (uint32_t*)(h_blob + l.m_direct - 1);
      const uint32_t la = f.mcode.log_alpha, le = 12 - la;
      for (uint32_t r = 0; r < (f.mcode.num_hist << 12); r++) {
        const uint32_t cl = r >> 12, res = r & 0xFFF, i = res >> le, pos = res & ((1u << le) - 1);
        const uint64_t e = f.mcode.alias[(cl << la) + i];
        const uint32_t x = (uint32_t)e, y = (uint32_t)(e >> 32);
        const bool g = pos >= (x & 0xFF);
        const uint32_t sym = g ? ((x >> 8) & 0xFF) : i, o = g ? (y & 0xFFFF) + pos : pos, freq = g ? ((x >> 16) ^ (y >> 16)) : (x >> 16);
        dt[r] = ((freq - 1) & 0xFFF) | ((o & 0xFFF) << 12) | (sym << 24);
      }
      d.mcode.direct = (const uint32_t*)(d_blob + l.m_direct - 1);
    }
    d.sec_off = (const uint64_t*)(d_blob + l.sec_off);
    d.sec_size = (const uint32_t*)(d_blob + l.sec_size);
    d.tree = (const DevTreeNode*)(d_blob + l.tree);
    d.tree_size = (int32_t)f.tree.size();
    const bool resident = dev_data && dev_data[i] && f.cs_contiguous;
    if (resident) d.cs = dev_data[i] + f.cs_file_offset;
    else { put(l.cs, f.cs, f.cs_size); d.cs = d_blob + l.cs; }
    d.cs_size = f.cs_size;
    if (f.encoding == 1) {
      d.is_modular = 1;
      d.w8 = d.h8 = d.wt = d.ht = d.wp = d.hp = 0;   // nothing of the VarDCT pipeline runs for this image
      d.cmyk = f.black_index >= 0 ? 1 : 0;
      d.black_bits = d.cmyk ? (int32_t)f.ec[f.black_index].bits : 8;
      d.nch_out = d.ncolor + d.cmyk + d.has_alpha;
      d.mod_nch = d.nch_out;
      // stream order: colour channels, then the extra channels as listed; output order: colour, black, alpha
      for (int c = 0; c < d.ncolor; c++) d.mod_out_pos[c] = c;
      if (d.cmyk) d.mod_out_pos[d.ncolor + f.black_index] = d.ncolor;
      if (d.has_alpha) d.mod_out_pos[d.ncolor + f.alpha_index] = d.nch_out - 1;
      d.group_dim = (int32_t)f.group_dim;
      d.single = f.single ? 1 : 0;
      d.mod_data_bits = f.mod_data_bits;
      // without Squeeze (at most four colour transforms) the RCTs are undone inside modular_out_kernel; otherwise every inverse
      // operation is its own launch (below) and the output kernel only clamps and interleaves
      const bool inline_rct = !f.mod_has_squeeze && !f.mod_has_palette && f.mod_transforms.size() <= 4;
      d.mod_ntr = inline_rct ? (int32_t)f.mod_transforms.size() : 0;
      for (int t = 0; t < d.mod_ntr; t++) { d.mod_tr[t][0] = (int32_t)f.mod_transforms[t].begin_c; d.mod_tr[t][1] = (int32_t)f.mod_transforms[t].rct_type; }
      for (int c = 0; c < d.mod_nch; c++) d.mod_plane[c] = (int32_t*)(wr + l.mod_planes[c]);
      {
        std::vector<ModChanDev> table;
        for (auto& ch : f.mod_coded) table.push_back(ModChanDev{ch.w, ch.h, ch.hshift, ch.vshift, (int32_t*)(wr + l.mod_planes[ch.plane])});
        put(l.mod_chan, table.data(), sizeof(ModChanDev) * table.size());
      }
      d.mod_chan = (const ModChanDev*)(d_blob + l.mod_chan);
      d.mod_ncoded = (int32_t)f.mod_coded.size();
      d.mod_first_group = (int32_t)f.mod_first_group_channel;
      d.mod_desc = (ChanDesc*)(wr + l.mod_desc);
      if (f.tree_uses_wp) { d.wp_grp = (int32_t*)(wr + l.wp_grp); d.wp_grp_ints = 10 * ((int64_t)f.group_dim + 2); }
      if (l.lz_mod) d.lz_mod = (uint32_t*)(wr + l.lz_mod - 1);
      if (!inline_rct)
        for (auto& op : f.mod_ops) {
          const ParsedFrame::ModPlane &pa = f.mod_planes[op.a], &pb = f.mod_planes[op.b];
          ModLaunch ml{op.kind, (int32_t*)(wr + l.mod_planes[op.a]), (int32_t*)(wr + l.mod_planes[op.b]),
                       (int32_t*)(wr + l.mod_planes[op.c]), pa.w, pa.h, pb.w, pb.h, op.type, {nullptr, nullptr, nullptr, nullptr}, op.nout, (uint32_t*)(wz + l.z_status)};
          for (int k = 0; k < op.nout && k < 4; k++) ml.out[k] = (int32_t*)(wr + l.mod_planes[op.out[k]]);
          mod_ops.push_back(ml);
        }
      d.status = (uint32_t*)(wz + l.z_status);
      status_off[i] = l.z_status;
      d.out = f.orientation == 1 ? dev_out[i] : wr + l.orient_tmp;
      auto code_lds_m = [](const HostCode& hc) { return 8 + 8 * hc.alias.size() + 4 * hc.cfg.size() + hc.ctx_map.size(); };
      // (one section per wavefront with row buffers: three rows and the leaf grid of modular_uniform.h instead of one row per lane)
      const size_t mod_rows = (mod_lanes == 1 && mod_rb) ? (size_t)3 * mod_rb * 4 + (size_t)kUniGridCells * 16 : (size_t)mod_lanes * mod_rb * 4;
      lds_mod = std::max(lds_mod, 64 * 128 + mod_rows + (mod_wp_lds ? (size_t)mod_lanes * 10 * (mod_rb + 2) * 4 : 0) + 16 +
                                      sizeof(DevTreeNode) * f.tree.size() + code_lds_m(f.mcode));
      const uint32_t nsec = 1 + f.nlf + f.ng;
      max_mod_groups = std::max<int>(max_mod_groups, (int)nsec);
      max_mod_coded = std::max<int>(max_mod_coded, (int)f.mod_coded.size());
      max_mod_pixels = std::max(max_mod_pixels, (size_t)f.xsize * f.ysize);
      if (f.single) mod_tasks[nmod_t++] = SectionTask{i, 0, 1, 0};   // one bit stream: one lane walks all three sections
      else for (uint32_t g = 0; g < nsec; g += mod_lanes) mod_tasks[nmod_t++] = SectionTask{i, (int32_t)g, (int32_t)std::min<uint32_t>(mod_lanes, nsec - g), 0};
      continue;
    }
    code(f.acode, l.a_cmap, l.a_cfg, l.a_alias, d.acode, l.a_pfx);
    if (l.lz_lf) d.lz_lf = (uint32_t*)(wr + l.lz_lf - 1);
    if (l.lz_grp) d.lz_grp = (uint32_t*)(wr + l.lz_grp - 1);
    if (l.lz_hf) d.lz_hf = (uint32_t*)(wr + l.lz_hf - 1);
    d.num_presets = f.num_presets;
    d.num_block_ctx = f.num_block_ctx;
    memcpy(d.block_ctx_map, f.block_ctx_map.data(), std::min(sizeof(d.block_ctx_map), f.block_ctx_map.size()));
    d.n_qf = (int32_t)f.qf_thr.size();
    d.num_lf_ctx = 1;
    for (int j = 0; j < 3; j++) {
      d.n_lf_thr[j] = (int32_t)std::min<size_t>(15, f.lf_thr[j].size());
      for (int k = 0; k < d.n_lf_thr[j]; k++) d.lf_thr[j][k] = f.lf_thr[j][k];
      d.num_lf_ctx *= d.n_lf_thr[j] + 1;
    }
    d.custom_orders = 0;
    for (size_t k = 0; k < f.qf_thr.size() && k < 15; k++) d.qf_thr[k] = f.qf_thr[k];
    for (int o = 0; o < kNumOrders; o++)
      for (int c = 0; c < 3; c++) {
        if (f.custom_order[o][c].empty()) d.order[o * 3 + c] = d_natural[o];
        else { put(l.order[o][c], f.custom_order[o][c].data(), 2 * f.custom_order[o][c].size()); d.order[o * 3 + c] = (const uint16_t*)(d_blob + l.order[o][c]); d.custom_orders = 1; }
      }
    d.inv_global_scale = 65536.0f / f.global_scale;
    d.quant_scale = f.global_scale / 65536.0f;
    for (int c = 0; c < 3; c++) d.mul_lf[c] = f.m_lf[c] * (d.inv_global_scale / f.quant_lf);
    d.inv_color_factor = 1.0f / f.color_factor;
    d.lf_cfl_x = f.base_x + f.ytox_lf * d.inv_color_factor;
    d.lf_cfl_b = f.base_b + f.ytob_lf * d.inv_color_factor;
    d.base_x = f.base_x; d.base_b = f.base_b;
    d.x_dm = std::pow(0.8f, (float)f.x_qm_scale - 2.0f);
    d.b_dm = std::pow(0.8f, (float)f.b_qm_scale - 2.0f);
    memcpy(d.qbias, f.qbias, sizeof(d.qbias));
    for (int q = 0; q < kNumQuantTables; q++) {
      d.dq[q] = d_dq[q]; d.dq_n[q] = dq_n[q]; d.scan[q] = d_scan[q];
      if (l.dq[q]) { put(l.dq[q] - 1, f.custom_dq[q].data(), 4 * f.custom_dq[q].size()); d.dq[q] = (const float*)(d_blob + l.dq[q] - 1); }
      if (l.scan[q]) {
        std::vector<U32x2> sl;
        BuildScanList(q, f.custom_order, sl, l.dq[q] ? &f.custom_dq[q] : nullptr);
        put(l.scan[q] - 1, sl.data(), sl.size() * sizeof(U32x2));
        d.scan[q] = (const U32x2*)(d_blob + l.scan[q] - 1);
      }
    }
    d.gab = f.gab; d.epf_iters = f.epf_iters; d.skip_lf_smoothing = (f.flags & 128) ? 1 : 0;
    for (int c = 0; c < 3; c++) {
      float div = 1.0f + 4.0f * (f.gab_w1[c] + f.gab_w2[c]);
      d.gab_w[c][0] = 1.0f / div; d.gab_w[c][1] = f.gab_w1[c] / div; d.gab_w[c][2] = f.gab_w2[c] / div;
    }
    memcpy(d.epf_sharp_lut, f.epf_sharp_lut, sizeof(d.epf_sharp_lut));
    memcpy(d.epf_channel_scale, f.epf_channel_scale, sizeof(d.epf_channel_scale));
    d.epf_quant_mul = f.epf_quant_mul; d.epf_pass0_sigma_scale = f.epf_pass0_sigma_scale;
    d.epf_pass2_sigma_scale = f.epf_pass2_sigma_scale; d.epf_border_sad_mul = f.epf_border_sad_mul;
    // linear RGB of the image's own primaries, relative to its intensity target: change of primaries folded into the inverse opsin matrix
    for (int r = 0; r < 3; r++)
      for (int k = 0; k < 3; k++) {
        double a = 0;
        for (int j = 0; j < 3; j++) a += (double)plan.from_srgb[r * 3 + j] * (double)f.opsin_inv[j * 3 + k];
        d.opsin_inv[r * 3 + k] = (float)a * (255.0f / f.intensity_target);
      }
    for (int k = 0; k < 3; k++) { d.opsin_bias[k] = f.opsin_bias[k]; d.opsin_bias_cbrt[k] = std::cbrt(f.opsin_bias[k]); }
    // planes
    d.cellinfo = (uint32_t*)(wz + l.z_cellinfo);
    d.status = (uint32_t*)(wz + l.z_status);
    status_off[i] = l.z_status;
    for (int c = 0; c < 3; c++) {
      d.coef[c] = (int32_t*)(wr + chunk_coef[(size_t)(i % pixel_chunk) * 3 + c]);
      d.lf[c] = (float*)(wr + l.lf[c]); d.lf_tmp[c] = (float*)(wr + l.lf_tmp[c]); d.lfq[c] = (int32_t*)(wr + l.lfq[c]);
      d.lf_final[c] = d.skip_lf_smoothing ? d.lf[c] : d.lf_tmp[c];
      d.tmp[c] = (float*)(wr + chunk_tmp[(size_t)(i % pixel_chunk) * 3 + c]); d.xyb[c] = (float*)(wr + chunk_xyb[(size_t)(i % pixel_chunk) * 3 + c]);
      d.xyb2[c] = (float*)d.coef[c];   // the dense coefficient planes (generic path only) are dead once the frame is reconstructed
    }
    d.lf_extra = wr + l.lf_extra;
    d.rawq = (uint16_t*)(wr + l.rawq); d.sharp = wr + l.sharp;
    d.ytox = (int8_t*)(wr + l.ytox); d.ytob = (int8_t*)(wr + l.ytob);
    d.binfo = (int32_t*)(wr + l.binfo);
    d.lf_desc = (ChanDesc*)(wr + l.lf_desc); d.lf_count = (uint32_t*)(wr + l.lf_count); d.alpha_desc = (ChanDesc*)(wr + l.alpha_desc);
    d.blk_list = (uint32_t*)(wr + l.blk_list); d.blk_count = (uint32_t*)(wr + l.blk_count);
    d.grp_bitpos = (uint64_t*)(wr + l.bitpos);
    d.tile_list = (uint32_t*)(wr + l.tile_list);
    d.alpha32 = (int32_t*)(wr + l.alpha32);
    d.centries = (uint32_t*)(wr + l.centries); d.centries_g0 = d.dec_gy0 * (int32_t)f.xg;
    d.cblk = (U32x2*)(wr + l.cblk);
    d.inv_sigma = (float*)(wr + l.inv_sigma);
    d.alpha = wr + l.alpha;
    d.lf_end_bits = (uint64_t*)(wr + l.lf_end);
    // a frame of one group has its alpha channel in LfGlobal (channels no larger than a group are coded globally)
    d.alpha_in_global = (d.has_alpha && f.ng == 1) ? 1 : 0;
    d.lf_start_bits = f.after_lf_global_bits;
    if (f.single) {
      d.single = 1;
      d.hf_start_bits = f.hf_start_bits;
    }
    if (f.tree_uses_wp) { d.wp_lf = (int32_t*)(wr + l.wp_lf); d.wp_grp = (int32_t*)(wr + l.wp_grp); d.wp_grp_ints = 10 * (kGroupDim + 2); }
    d.out = f.orientation == 1 ? dev_out[i] : wr + l.orient_tmp;
    // Loop-filter routing.  Frames with EPF iterations run iteration 1 (+ Gaborish when no iteration 0 has to come between them) and
    // iteration 2 in the streaming kernels: fused_gab_epf1 = 1: one kernel -> output; 2: two kernels, f32 rows (stream_mid) between
    // them.  Three iterations (distance >= 4): Gaborish and iteration 0 first, as LDS-tiled stage kernels, then the two streaming
    // kernels without Gaborish (stream_no_gab).  Stage kernels ping-pong between xyb and xyb2 (the dead dense-coefficient planes).
    float** cur = d.xyb;
    float** other = d.xyb2;
    d.fused_gab_epf1 = (!debug_taps && f.epf_iters >= 1) ? (f.epf_iters == 1 ? 1 : 2) : 0;
    d.stream_no_gab = (d.fused_gab_epf1 && (!f.gab || f.epf_iters == 3)) ? 1 : 0;
    d.stage_on[0] = (f.gab && (!d.fused_gab_epf1 || f.epf_iters == 3)) ? 1 : 0;
    d.stage_on[1] = f.epf_iters == 3;
    d.stage_on[2] = f.epf_iters >= 1 && !d.fused_gab_epf1;
    d.stage_on[3] = f.epf_iters >= 2 && !d.fused_gab_epf1;
    d.stage_on[4] = 1;
    for (int s = 0; s < 5; s++) {
      if (s == 2) for (int c = 0; c < 3; c++) { d.stream_in[c] = cur[c]; d.stream_mid[c] = other[c]; }
      for (int c = 0; c < 3; c++) { d.stage_in[s][c] = cur[c]; d.stage_out[s][c] = other[c]; }
      if (s < 4 && d.stage_on[s]) std::swap(cur, other);
    }
    d.final_stage = 4;
    for (int s = 0; s < 4; s++) if (d.stage_on[s]) d.final_stage = s;
    if (debug_taps) d.final_stage = 4;   // keep the filtered float planes for the stage taps; out_only_kernel converts
    // the layouts the two-pixels-per-lane kernels handle: even width; 8-bit RGBA / RGB / gray + alpha / gray output (bit 0: the Gaborish + first
    // iteration kernel, which for a two-iteration frame writes f32 rows whatever the output; bit 1: the second iteration's kernel)
    {
      // (their buffer resources address 2 GB from a plane's base: larger planes take the general kernels)
      const bool even = d.fused_gab_epf1 && (d.w & 1) == 0 && d.w >= 8 && !no_stream_pairs && (uint64_t)d.wp * (uint64_t)d.hp * 4u < (1ull << 31);
      // 8-bit samples, sRGB or linear: RGBA, RGB, gray + alpha, gray (a frame's channel count is ncolor + has_alpha)
      const bool rgba8 = d.out_bits == 8 && d.to_srgb <= 1 && !d.unpremultiply && (d.ncolor == 3 || d.ncolor == 1) &&
                         d.nch_out == d.ncolor + (d.has_alpha ? 1 : 0) && !d.cmyk;
      d.stream_pairs = (even && (d.fused_gab_epf1 == 2 || rgba8) ? 1 : 0) | (even && d.fused_gab_epf1 == 2 && rgba8 ? 2 : 0);
    }
    if (d.fused_gab_epf1) {
      d.final_stage = 5;
      any_fused |= (d.stream_pairs & 1) ? 1 : 2;
      if (d.fused_gab_epf1 == 2) any_fused2 |= (d.stream_pairs & 2) ? 1 : 2;
    }
    any_unfiltered |= d.final_stage == 4;
    max_w = std::max<int>(max_w, f.xsize); max_h = std::max<int>(max_h, f.ysize);
    max_tiles = std::max<int>(max_tiles, (int)tiles_of(f));
    for (int st = 0; st < 4; st++) if (d.stage_on[st]) stage_mask |= 1 << st;
    any_alpha |= d.has_alpha != 0;
    max_cells = std::max(max_cells, (size_t)f.w8 * f.h8);
    max_padded = std::max(max_padded, (size_t)f.w8 * f.h8 * 64);
    // LDS budgets (must mirror the carving in entropy_kernels.hip)
    auto code_lds = [](const HostCode& hc) { return 8 + 8 * hc.alias.size() + 4 * hc.cfg.size() + hc.ctx_map.size(); };
    lds_lf = std::max(lds_lf, (size_t)lf_per_wave * 128 + 16 + sizeof(DevTreeNode) * f.tree.size() + code_lds(f.mcode));
    lds_alpha = std::max(lds_alpha, (size_t)per_alpha_wg * 128 + 16 + sizeof(DevTreeNode) * f.tree.size() + code_lds(f.mcode));
    max_groups = std::max<int>(max_groups, (int)f.ng);
    // LF groups that intersect the decoded group rows (8 group rows per LF group row); HF groups of the decoded rows; alpha of the band
    const uint32_t lfy0 = (uint32_t)d.dec_gy0 / 8, lfy1 = ((uint32_t)d.dec_gy1 + 7) / 8;
    for (uint32_t g = lfy0 * f.xlf; g < std::min<uint32_t>(f.nlf, lfy1 * f.xlf); g++) lf_tasks[nlf_t++] = SectionTask{i, (int32_t)g, 1, 0};
    {
      const uint32_t l0 = lfy0 * f.xlf, l1 = std::min<uint32_t>(f.nlf, lfy1 * f.xlf);
      for (uint32_t g = l0; g < l1; g += lf_per_wave)
        lf_ans_tasks[nlf_ans_t++] = SectionTask{i, (int32_t)g, (int32_t)std::min<uint32_t>((uint32_t)lf_per_wave, l1 - g), 0};
    }
    // progressive frames: one record per further pass, chained from this one
    d.num_passes = (int32_t)f.num_passes;
    d.pass_shift = (int32_t)f.pass_shift[0];
    d.hf_sec_base = 2 + (int32_t)f.nlf;
    d.alpha_sec_base = 2 + (int32_t)f.nlf + (int32_t)((f.num_passes - 1) * f.ng);   // the Modular streams of all shifts below 3 are in the last pass
    d.alpha_bitpos = d.grp_bitpos;
    d.next_pass = nullptr;
    for (size_t p = 0; p < f.extra_passes.size(); p++) {
      const ParsedFrame::PassCodes& ep = f.extra_passes[p];
      const PassLayout& pl = l.extra[p];
      DevImage& sh = imgs[(size_t)first_extra[i] + p];
      sh = d;
      memset(&sh.acode, 0, sizeof(sh.acode));
      code(ep.acode, pl.a_cmap, pl.a_cfg, pl.a_alias, sh.acode, pl.a_pfx);
      sh.lz_hf = pl.lz_hf ? (uint32_t*)(wr + pl.lz_hf - 1) : nullptr;
      for (int q = 0; q < kNumQuantTables; q++) {
        sh.scan[q] = d_scan[q];
        if (pl.scan[q]) {
          std::vector<U32x2> sl;
          BuildScanList(q, ep.custom_order, sl, l.dq[q] ? &f.custom_dq[q] : nullptr);
          put(pl.scan[q] - 1, sl.data(), sl.size() * sizeof(U32x2));
          sh.scan[q] = (const U32x2*)(d_blob + pl.scan[q] - 1);
        }
      }
      sh.centries = (uint32_t*)(wr + pl.centries);
      sh.cblk = (U32x2*)(wr + pl.cblk);
      sh.grp_bitpos = (uint64_t*)(wr + pl.bitpos);
      sh.pass_shift = p + 1 < f.num_passes - 1 ? (int32_t)f.pass_shift[p + 1] : 0;
      sh.hf_sec_base = 2 + (int32_t)f.nlf + (int32_t)((p + 1) * f.ng);
      sh.next_pass = nullptr;
      d.alpha_bitpos = sh.grp_bitpos;
    }
    {
      const DevImage* d_recs = (const DevImage*)(d_blob + off_imgs);
      for (size_t p = 0; p < f.extra_passes.size(); p++) {
        DevImage& prev = p ? imgs[(size_t)first_extra[i] + p - 1] : d;
        prev.next_pass = d_recs + first_extra[i] + p;
        imgs[(size_t)first_extra[i] + p].alpha_bitpos = d.alpha_bitpos;
      }
    }
    const uint32_t hg0 = (uint32_t)d.dec_gy0 * f.xg, hg1 = (uint32_t)d.dec_gy1 * f.xg;
    const uint32_t pw = (uint32_t)hf_per_wg(f);
    for (uint32_t pass = 0; pass < f.num_passes; pass++) {
      // Sections go to lanes in order of their byte size (the TOC has it), largest first: a wavefront runs until its longest
      // section ends, so lanes of similar length finish together (the sum over wavefronts of their longest lane - the
      // wave-instructions of the launch - nearly halves for 4K frames, whose sections spread 1 : 2.2 around the mean), and the
      // longest sections start first.  Slot j of the (image, pass) decodes group hf_order[j]; tasks index slots.
      DevImage& rec = pass ? imgs[(size_t)first_extra[i] + pass - 1] : d;
      const size_t order_off = pass ? l.extra[pass - 1].hf_order : l.hf_order;
      uint32_t* order = (uint32_t*)(h_blob + order_off);
      const size_t sec0 = f.single ? 0 : 2 + (size_t)f.nlf + (size_t)pass * f.ng;
      for (uint32_t g = hg0; g < hg1; g++) order[g - hg0] = g;
      if (!f.single && !Knob("JXLHIP_NO_HF_SORT"))   // (experiment knob: measured effect of the order, profiles/r03_hf_sort_ab.txt)
        std::stable_sort(order, order + (hg1 - hg0), [&](uint32_t a, uint32_t b) { return f.sec_size[sec0 + a] > f.sec_size[sec0 + b]; });
      rec.hf_order = (const uint32_t*)(d_blob + order_off);
      for (uint32_t j = 0; j < hg1 - hg0; j += pw) {
        const uint32_t cnt = std::min<uint32_t>(pw, hg1 - hg0 - j);
        pass_tasks[npass_t++] = SectionTask{pass ? first_extra[i] + (int)pass - 1 : i, (int32_t)j, (int32_t)cnt, 0};
        const size_t lanes = (size_t)((cnt + 3) & ~3u) * HfLaneLdsBytes(32);   // the kernel lays its per-lane arrays out for the task's lanes
        lds_hf = std::max(lds_hf, hf_table_bytes(f) + lanes);
        lds_hf_lanes = std::max(lds_hf_lanes, lanes);
      }
    }
    const uint32_t ag0 = (uint32_t)(d.band_y0 / kGroupDim) * f.xg, ag1 = (uint32_t)((d.band_y1 + kGroupDim - 1) / kGroupDim) * f.xg;
    if (d.has_alpha)
      for (uint32_t g = ag0; g < ag1; g += per_alpha_wg)
        alpha_tasks[nalpha_t++] = SectionTask{i, (int32_t)g, (int32_t)std::min<uint32_t>(per_alpha_wg, ag1 - g), 0};
  }
  if (Knob("JXLHIP_DEBUG_LDS")) {
    for (int i = 0; i < std::min(n, 8); i++)
      if (parse_status[i] == DecoderStatus_Ok && frames[i].encoding == 0)
        fprintf(stderr, "[jxlhip] image %d: hf tables %zu B (clusters %u, log_alpha %u, contexts %zu), lanes/wg %d; modular tables: clusters %u log_alpha %u contexts %zu tree %zu\n", i,
                hf_table_bytes(frames[i]), frames[i].acode.num_hist, frames[i].acode.log_alpha, frames[i].acode.ctx_map.size(), hf_per_wg(frames[i]),
                frames[i].mcode.num_hist, frames[i].mcode.log_alpha, frames[i].mcode.ctx_map.size(), frames[i].tree.size());
    fprintf(stderr, "[jxlhip] launch LDS: hf %zu B (lanes %zu), lf %zu B, alpha %zu B; hf workgroups %d x %d threads, stride %d; alpha workgroups %d; lf_ans workgroups %d\n",
            lds_hf, lds_hf_lanes, lds_lf, lds_alpha, npass_t, hf_waves * 64, lane_stride, nalpha_t, nlf_ans_t);
  }
  const int hf_ring = 32;   // words of the per-lane bit window
  const size_t kLdsMax = 150 * 1024;
  // experiment knobs: code tables of the entropy kernels read from global memory (L1 / L2) instead of LDS copies
  if (Knob("JXLHIP_HF_GLOBAL")) lds_hf = kLdsMax + 1;
  if (Knob("JXLHIP_ALPHA_GLOBAL")) lds_alpha = kLdsMax + 1;
  if (Knob("JXLHIP_LF_GLOBAL")) lds_lf = kLdsMax + 1;
  memcpy(h_blob + off_imgs, imgs.data(), sizeof(DevImage) * imgs.size());
  d_imgs = (DevImage*)(d_blob + off_imgs);
  // ---- 4. enqueue: LF chain on s_lf, everything that needs the block layout on the main stream
  stage_names.clear();
  S.stage_chain.clear();
  Mark("start", s_lf, 0);
  // experiment knob (timing only, the output is stale): bit 0 skips the LF chain, 1 the HF decode, 2 alpha, 3 reconstruction, 4 filters -
  // with the same files resubmitted, a skipped stage's results of the previous batch in this workspace slot are still in place
  static const int skip_stages = Knob("JXLHIP_SKIP_STAGES") ? atoi(Knob("JXLHIP_SKIP_STAGES")) : 0;
  if (!(skip_stages & 1)) HIP_OK(hipMemsetAsync(d_ws, 0, zero_bytes, s_lf));
  HIP_OK(hipMemcpyAsync(d_blob, h_blob, blob.off, hipMemcpyHostToDevice, s_lf));
  Mark("upload+clear", s_lf, 0);
  // wavefronts that decode one section run their row loops on the scalar unit from the per-residue tables built above
  // (measured at batch 384: LF sections as four such wavefronts per workgroup - 135.7 ms per batch against 125.1 with four lanes of
  // one wavefront: large batches keep the lane layout, the scalar path is for small ones)
  const int direct_lf = global_direct && lf_per_wave == 1, direct_alpha = global_direct && per_alpha_wg == 1;
  // the lean forms of the LF / alpha kernels (no per-sample Modular path compiled in: two thirds of the registers) when no lossy frame
  // of the batch can take that path: row-static MA tree, standard predictors, plain ANS codes
  bool lean_mod = true;
  for (int i = 0; i < n; i++)
    if (parse_status[i] == DecoderStatus_Ok && frames[i].encoding == 0 &&
        (!frames[i].tree_row_static || frames[i].tree_uses_wp || frames[i].mcode.use_prefix || frames[i].mcode.lz77)) lean_mod = false;
  if (!(skip_stages & 1)) {
  LaunchLfAns(d_imgs, (const SectionTask*)(d_blob + off_lf_ans_tasks), nlf_ans_t, lf_per_wave, lds_lf <= kLdsMax ? lds_lf : 0, direct_lf, lean_mod, s_lf);
  Mark("lf_ans", s_lf, 0);
  LaunchLfFinish(d_imgs, (const SectionTask*)(d_blob + off_lf_tasks), nlf_t, s_lf);
  LaunchHfBlockList(d_imgs, n, max_groups, s_lf);
  LaunchLfPixelStages(d_imgs, n, max_cells, s_lf);
  }
  Mark("lf_finish+pixels", s_lf, 0);
  // three chains, three streams: LF (batch k+2) | HF coefficients (batch k+1) | alpha + pixels (batch k)
  if (s_lf != s_hf) {
    HIP_OK(hipEventRecord(S.lf_done, s_lf));
    HIP_OK(hipStreamWaitEvent(s_hf, S.lf_done, 0));
  }
#ifdef JXLHIP_EXPERIMENTS
  // JXLHIP_INTERFERE=kind,workgroups-per-CU,LDS-KB,ms: a probe kernel that occupies one resource (kernels.hip) starts with the HF stage on a
  // stream of its own and runs for `ms`: which stage is sensitive to which resource (tools/interfere.sh)
  if (const char* e = Knob("JXLHIP_INTERFERE")) {
    int kind = 0, wgs = 1, lds_kb = 64; float ms = 100.f;
    if (sscanf(e, "%d,%d,%d,%f", &kind, &wgs, &lds_kb, &ms) >= 1 && kind >= 1 && kind <= 6 && wgs >= 1 && wgs <= 16 && lds_kb >= 1 && lds_kb <= 160 && ms > 0 && ms < 2000) {
      static hipStream_t xs = nullptr;
      static float* xbuf = nullptr;
      const size_t xbytes = (size_t)512 << 20;
      if (!xs) { HIP_OK(hipStreamCreateWithFlags(&xs, hipStreamNonBlocking)); HIP_OK(hipMalloc(&xbuf, xbytes)); }
      if (s_lf != s_hf) HIP_OK(hipStreamWaitEvent(xs, S.lf_done, 0));
      LaunchInterference(kind, wgs, lds_kb, ms, xbuf, xbytes, xs);
    }
  }
#endif
  Mark("hf_start", s_hf, 1);
  if (!(skip_stages & 2))
  LaunchHfDecode(d_imgs, (const SectionTask*)(d_blob + off_pass_tasks), npass_t, hf_waves * 64, lane_stride, hf_ring, lds_hf <= kLdsMax ? lds_hf : 0, lds_hf_lanes, s_hf);
  Mark("hf_decode", s_hf, 1);
  // alpha follows the HF tokens in every pass-group section (its first bit is where the HF kernel stopped reading): same chain,
  // necessarily; the main stream carries nothing but the pixel stages
  if (any_alpha && !(skip_stages & 4))
    LaunchAlphaAns(d_imgs, (const SectionTask*)(d_blob + off_alpha_tasks), nalpha_t, alpha_stride, lds_alpha <= kLdsMax ? lds_alpha : 0, direct_alpha, lean_mod, s_hf);
  Mark("alpha_ans", s_hf, 1);
  if (debug_taps) {   // the quantised coefficients as dense planes (every frame has its own planes in this mode)
    taps.assign(n, Tap());
    LaunchExpandCoefficients(d_imgs, n, true, max_tiles, stream);
    HIP_OK(hipStreamSynchronize(stream));
    CopyPlaneTap(0);
  }
  // the alpha planes' prediction pass: behind the token pass on the HF stream (the pixel stream is the longest of the three chains
  // since the reconstruction workgroup grew to 72 KB of LDS and the HF kernel keeps its residency beside it: 107.2 against 108.7 ms;
  // while the HF stream was the longest it was the other way round - knob JXLHIP_ALPHA_FINISH_ON_PIX)
  const bool finish_on_hf = s_hf == stream || debug_taps || !Knob("JXLHIP_ALPHA_FINISH_ON_PIX");
  if (finish_on_hf) {
    if (any_alpha && !(skip_stages & 4)) LaunchAlphaFinish(d_imgs, n, max_groups, s_hf);
    Mark("alpha_finish", s_hf, 1);
  }
  if (s_hf != stream) {
    HIP_OK(hipEventRecord(S.hf_done, s_hf));
    HIP_OK(hipStreamWaitEvent(stream, S.hf_done, 0));
  }
  Mark("main_start", stream, 2);
  if (!finish_on_hf) {
    if (any_alpha && !(skip_stages & 4)) LaunchAlphaFinish(d_imgs, n, max_groups, stream);
    Mark("alpha_finish", stream, 2);
  }
  // experiment knob: the pixel stages behind the HF chain on ITS stream (no overlap between a batch's pixels and the next batch's HF decode)
  hipStream_t s_pix = (Knob("JXLHIP_PIX_ON_HF") && s_hf != stream) ? s_hf : stream;
  for (int c0 = 0; c0 < n; c0 += pixel_chunk) {
    const int cnt = std::min(pixel_chunk, n - c0);
    if (!(skip_stages & 8))
    LaunchReconTiles(d_imgs + c0, cnt, max_tiles, d_basis_all, d_basis_small, d_llf_scale, d_basis_mfma, s_pix);
    Mark("reconstruct", s_pix, 2);   // exactly recon_tile_kernel; one mark per chunk, the per-stage totals add them up
    LaunchExpandCoefficients(d_imgs + c0, cnt, false, max_tiles, s_pix);
    LaunchGenericReconstruct(d_imgs + c0, cnt, d_basis_all, d_basis_small, d_llf_scale, s_pix);
    Mark("reconstruct_generic", s_pix, 2);
    if (debug_taps) { HIP_OK(hipStreamSynchronize(s_pix)); CopyPlaneTap(1); }
    if (!(skip_stages & 16))
    LaunchFilterTiles(d_imgs + c0, cnt, max_w, max_h, stage_mask, any_unfiltered, any_fused, any_fused2, s_pix);
    Mark("filters+output", s_pix, 2);
  }
  if (s_pix != stream) {
    HIP_OK(hipEventRecord(S.hf_done, s_pix));
    HIP_OK(hipStreamWaitEvent(stream, S.hf_done, 0));
  }
  if (nmod_t) {
    // Modular (lossless) frames of the batch; they depend on nothing but the upload
    if (s_lf != stream) { HIP_OK(hipEventRecord(S.lf_done, s_lf)); HIP_OK(hipStreamWaitEvent(stream, S.lf_done, 0)); }
    const int direct_mod = global_direct && mod_lanes == 1;
    LaunchModularAns(d_imgs, n, (const SectionTask*)(d_blob + off_mod_tasks), nmod_t, lds_mod <= kLdsMax ? lds_mod : 0, max_mod_groups,
                     max_mod_coded, mod_lanes, mod_rb, mod_wp_lds, direct_mod, stream);
    for (auto& op : mod_ops) {
      if (op.kind == 3) LaunchModularPalette(op.a, op.b, op.out, op.nout, op.type, op.rw, op.rh, op.status, stream);   // a: palette, b: indices (rw x rh)
      else LaunchModularOp(op.kind, op.a, op.b, op.c, op.aw, op.ah, op.rw, op.rh, op.type, stream);
    }
    LaunchModularOut(d_imgs, n, max_mod_pixels, stream);
    Mark("modular", stream, 2);
  }
  for (int i = 0; i < n; i++)
    if (parse_status[i] == DecoderStatus_Ok && frames[i].orientation != 1) {
      const ParsedFrame& f = frames[i];
      LaunchOrient(imgs[i].out, dev_out[i], (int)f.xsize, (int)f.ysize, (f.ncolor + (f.black_index >= 0 ? 1 : 0) + (f.alpha_index >= 0 ? 1 : 0)) * (int)OutBytesPerSample(f),
                   (int)f.orientation, stream);
    }
  LaunchStatusToHost((const uint32_t*)(d_ws + z_status_base), h_status, n * 16, stream);   // (images that failed to parse: zeros)
  HIP_OK(hipEventRecord(S.done, stream));
  HIP_OK(hipGetLastError());
  last_stream = stream;
  S.pending = true;
  last = cur;
  cur = (cur + 1) % kSlots;
  (void)max_padded;
  if (sync) {
    DecoderStatus st = Finish(statuses, err);
    (void)st;
    if (debug_taps) CopyPlaneTap(2);
  } else if (statuses) {
    for (int i = 0; i < n; i++) statuses[i] = parse_status[i];
  }
}

// LF stage of ONE single-section frame on temporary buffers, synchronously: yields the bit position where HfGlobal starts,
// which the host then parses (ParseHfGlobalAt).  The main pass decodes the (tiny) LF group again with everything in place.
void JxlHipDecoder::PrepassSingle(ParsedFrame& f, const uint8_t* dev_file) {
  Bump b;
  const size_t cells = (size_t)f.w8 * f.h8;
  const size_t o_img = b.Take(sizeof(DevImage)), o_task = b.Take(sizeof(SectionTask));
  const size_t o_secoff = b.Take(8 * f.sec_off.size()), o_secsize = b.Take(4 * f.sec_size.size());
  const size_t o_tree = b.Take(sizeof(DevTreeNode) * f.tree.size());
  const size_t o_cmap = b.Take(f.mcode.ctx_map.size()), o_cfg = b.Take(4 * f.mcode.cfg.size()), o_alias = b.Take(8 * f.mcode.alias.size());
  // prefix codes: counts per length, symbol offsets, symbols sorted by code (as the main pass lays them out)
  std::vector<uint16_t> pfx_counts, pfx_sorted;
  std::vector<uint32_t> pfx_offs;
  if (f.mcode.use_prefix)
    for (auto& pc : f.mcode.prefix) {
      pfx_counts.insert(pfx_counts.end(), pc.count, pc.count + 16);
      pfx_offs.push_back((uint32_t)pfx_sorted.size());
      pfx_sorted.insert(pfx_sorted.end(), pc.sorted.begin(), pc.sorted.end());
    }
  const size_t o_pcount = b.Take(2 * std::max<size_t>(1, pfx_counts.size())), o_poff = b.Take(4 * std::max<size_t>(1, pfx_offs.size())),
               o_psorted = b.Take(2 * std::max<size_t>(1, pfx_sorted.size()));
  const bool resident = dev_file && f.cs_contiguous;
  const size_t o_cs = resident ? 0 : b.Take(f.cs_size + 16);
  const size_t upload = b.off;
  const size_t o_status = b.Take(64), o_end = b.Take(8), o_count = b.Take(4), o_extra = b.Take(4), o_desc = b.Take(8 * sizeof(ChanDesc));
  const size_t o_adesc = b.Take(sizeof(ChanDesc));
  const size_t zero_end = b.off;
  size_t o_lfq[3];
  for (int c = 0; c < 3; c++) o_lfq[c] = b.Take(4 * cells);
  const size_t o_binfo = b.Take((size_t)kBinfoInts * 4);
  const size_t o_alpha = b.Take(4 * (size_t)f.xsize * f.ysize);
  const size_t o_wp = f.tree_uses_wp ? b.Take((size_t)kWpLfInts * 4) : 0;
  const size_t o_lz = f.mcode.lz77 ? b.Take((size_t)4 << 20) : 0;   // the LF group's LZ77 window
  uint8_t* d = nullptr;
  HIP_OK(hipMalloc(&d, b.off));
  std::vector<uint8_t> h(upload, 0);
  DevImage im;
  memset(&im, 0, sizeof(im));
  im.w = f.xsize; im.h = f.ysize; im.w8 = f.w8; im.h8 = f.h8; im.xlf = im.ylf = im.nlf = 1; im.xg = im.yg = im.ng = 1;
  im.has_alpha = f.alpha_index >= 0;
  im.single = 1; im.alpha_in_global = im.has_alpha;
  im.lf_start_bits = f.after_lf_global_bits;
  im.cs = resident ? dev_file + f.cs_file_offset : d + o_cs;
  im.cs_size = f.cs_size;
  im.sec_off = (const uint64_t*)(d + o_secoff); im.sec_size = (const uint32_t*)(d + o_secsize);
  im.tree = (const DevTreeNode*)(d + o_tree); im.tree_size = (int32_t)f.tree.size();
  im.mcode.ctx_map = d + o_cmap; im.mcode.cfg = (const uint32_t*)(d + o_cfg); im.mcode.alias = (const uint64_t*)(d + o_alias);
  im.mcode.num_ctx = (uint32_t)f.mcode.ctx_map.size(); im.mcode.num_clusters = f.mcode.num_hist; im.mcode.log_alpha = f.mcode.log_alpha;
  im.mcode.slow = (f.mcode.use_prefix ? 1u : 0u) | (f.mcode.lz77 ? 2u : 0u);
  if (f.mcode.use_prefix) {
    im.mcode.pfx_count = (const uint16_t*)(d + o_pcount); im.mcode.pfx_off = (const uint32_t*)(d + o_poff); im.mcode.pfx_sorted = (const uint16_t*)(d + o_psorted);
  }
  if (f.mcode.lz77) {
    im.mcode.lz_min_symbol = f.mcode.lz_min_symbol; im.mcode.lz_min_length = f.mcode.lz_min_length;
    im.mcode.lz_len_cfg = f.mcode.lz_len.split | f.mcode.lz_len.msb << 4 | f.mcode.lz_len.lsb << 8;
    im.mcode.lz_dist_cluster = f.mcode.ctx_map.back();
    im.lz_lf = (uint32_t*)(d + o_lz);
  }
  im.status = (uint32_t*)(d + o_status); im.lf_end_bits = (uint64_t*)(d + o_end); im.lf_count = (uint32_t*)(d + o_count);
  im.lf_extra = d + o_extra; im.lf_desc = (ChanDesc*)(d + o_desc); im.alpha_desc = (ChanDesc*)(d + o_adesc);
  for (int c = 0; c < 3; c++) im.lfq[c] = (int32_t*)(d + o_lfq[c]);
  im.binfo = (int32_t*)(d + o_binfo); im.alpha32 = (int32_t*)(d + o_alpha);
  if (f.tree_uses_wp) im.wp_lf = (int32_t*)(d + o_wp);
  memcpy(h.data() + o_img, &im, sizeof(im));
  const SectionTask task{0, 0, 1, 0};
  memcpy(h.data() + o_task, &task, sizeof(task));
  memcpy(h.data() + o_secoff, f.sec_off.data(), 8 * f.sec_off.size());
  memcpy(h.data() + o_secsize, f.sec_size.data(), 4 * f.sec_size.size());
  memcpy(h.data() + o_tree, f.tree.data(), sizeof(DevTreeNode) * f.tree.size());
  memcpy(h.data() + o_cmap, f.mcode.ctx_map.data(), f.mcode.ctx_map.size());
  {
    std::vector<uint32_t> cfgp(f.mcode.cfg.size());
    for (size_t k = 0; k < cfgp.size(); k++) {
      cfgp[k] = f.mcode.cfg[k].split | f.mcode.cfg[k].msb << 4 | f.mcode.cfg[k].lsb << 8;
      if (f.mcode.use_prefix) {   // one-symbol prefix codes read no bits
        if (f.mcode.prefix[k].single >= 0) cfgp[k] |= 1u << 12 | (uint32_t)f.mcode.prefix[k].single << 16;
        continue;
      }
      const uint64_t e0 = f.mcode.alias[k << f.mcode.log_alpha];
      const uint32_t x0 = (uint32_t)e0, y0 = (uint32_t)(e0 >> 32);
      if ((x0 >> 16) == 0 && (y0 >> 16) == 4096) cfgp[k] |= 1u << 12 | ((x0 >> 8) & 0xFF) << 16;
    }
    memcpy(h.data() + o_cfg, cfgp.data(), 4 * cfgp.size());
  }
  if (!pfx_counts.empty()) memcpy(h.data() + o_pcount, pfx_counts.data(), 2 * pfx_counts.size());
  if (!pfx_offs.empty()) memcpy(h.data() + o_poff, pfx_offs.data(), 4 * pfx_offs.size());
  if (!pfx_sorted.empty()) memcpy(h.data() + o_psorted, pfx_sorted.data(), 2 * pfx_sorted.size());
  memcpy(h.data() + o_alias, f.mcode.alias.data(), 8 * f.mcode.alias.size());
  if (!resident) memcpy(h.data() + o_cs, f.cs, f.cs_size);
  hipError_t e = hipMemcpy(d, h.data(), upload, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(d + upload, 0, zero_end - upload);
  uint32_t st_words[16] = {0};
  uint64_t lf_end = 0;
  if (e == hipSuccess) {
    const size_t lds = 64 * 128 + 16 + sizeof(DevTreeNode) * f.tree.size() + 8 + 8 * f.mcode.alias.size() + 4 * f.mcode.cfg.size() + f.mcode.ctx_map.size();
    LaunchLfAns((const DevImage*)(d + o_img), (const SectionTask*)(d + o_task), 1, 64, lds <= 150 * 1024 ? lds : 0, 0, false, own_stream);
    e = hipStreamSynchronize(own_stream);
  }
  if (e == hipSuccess) e = hipMemcpy(st_words, d + o_status, 64, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(&lf_end, d + o_end, 8, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) throw HipError(std::string("single-group LF pre-pass: ") + hipGetErrorString(e));
  if (st_words[0]) throw ParseError(DecoderStatus_DecodeError, "GPU decode failed in the LF group of a single-group frame (corrupt bitstream)");
  f.hf_start_bits = ParseHfGlobalAt(f, lf_end);
}

void JxlHipDecoder::CopyPlaneTap(int stage) {
  Slot& S = *active;
  auto& n = S.n; auto& parse_status = S.parse_status; auto& imgs = S.imgs; auto& taps = S.taps;
  for (int i = 0; i < n; i++) {
    if (parse_status[i] != DecoderStatus_Ok) continue;
    const DevImage& d = imgs[i];
    size_t bytes = (size_t)d.wp * d.hp * 4;
    for (int c = 0; c < 3; c++) {
      std::vector<uint8_t>& dst = stage == 0 ? taps[i].qcoef[c] : (stage == 1 ? taps[i].xyb_idct[c] : taps[i].xyb_filtered[c]);
      const void* src = stage == 0 ? (const void*)d.coef[c] : (stage == 1 ? (const void*)d.xyb[c] : (const void*)d.stage_in[4][c]);
      dst.resize(bytes);
      HIP_OK(hipMemcpy(dst.data(), src, bytes, hipMemcpyDeviceToHost));
    }
  }
}

DecoderStatus JxlHipDecoder::Finish(DecoderStatus* statuses, ErrorInfo* err) {
  HIP_OK(hipSetDevice(device));
  // older batches first, in submission order (a failure there becomes the sticky error), then the most recent one
  for (int k = 1; k < kSlots; k++) {
    Slot& O = slots[(last + k) % kSlots];
    if (!O.pending) continue;
    WaitSlot(O);
    for (int i = 0; i < O.n; i++) {
      int st = O.parse_status[i];
      if (st == DecoderStatus_Ok && O.h_status[(size_t)i * 16]) st = DecoderStatus_DecodeError;
      if (st != DecoderStatus_Ok && sticky_status == DecoderStatus_Ok) { sticky_status = st; sticky_error = "an earlier asynchronous batch failed: " + O.parse_msg[i]; }
    }
  }
  Slot& S = Last();
  WaitSlot(S);
  DecoderStatus worst = DecoderStatus_Ok;
  for (int i = 0; i < S.n; i++) {
    DecoderStatus st = S.parse_status[i];
    if (st != DecoderStatus_Ok) {
      if (worst == DecoderStatus_Ok) SetErr(err, "%s", S.parse_msg[i].c_str());
    } else if (S.h_status[(size_t)i * 16]) {
      uint32_t bits = S.h_status[(size_t)i * 16];
      st = DecoderStatus_DecodeError;
      if (worst == DecoderStatus_Ok) {
        const uint32_t* w = S.h_status + (size_t)i * 16;   // [3..7]: last failing section + 1 of lf_ans / lf_finish / hf_decode / alpha_ans / modular
        if (bits & kErrUnsupportedTransform) SetErr(err, "AFV transforms are not supported (image %d, LF group section %u)", i, w[4]);
        else SetErr(err, "GPU decode failed (flags 0x%x:%s%s%s%s%s; image %d, sections lf %u hf %u alpha %u)", bits, bits & kErrBitstream ? " corrupt-bitstream" : "",
               bits & kErrUnsupportedHeader ? " unsupported-modular-header" : "", bits & kErrUnsupportedTree ? " unsupported-tree" : "",
               bits & kErrBlockLayout ? " invalid-varblock-layout" : "", bits & kErrRange ? " value-out-of-range" : "", i, w[3], w[5], w[6]);
      }
    }
    if (statuses) statuses[i] = st;
    if (st != DecoderStatus_Ok && worst == DecoderStatus_Ok) worst = st;
  }
  if (worst == DecoderStatus_Ok && sticky_status != DecoderStatus_Ok) {
    worst = sticky_status;
    SetErr(err, "%s", sticky_error.c_str());
  }
  sticky_status = DecoderStatus_Ok;
  sticky_error.clear();
  return worst;
}

// ====================================================================== C-ABI
extern "C" {

uint32_t GetLibJxlVersion(void) {
  // This library is not libjxl; it reports the libjxl API level whose behaviour it follows (0.11.1).
  return (0u << 24) | (11u << 16) | (1u << 8);
}

JxlHipDecoder* jxlhip_decoder_create(int32_t device, ErrorInfo* err) {
  try {
    return new JxlHipDecoder(device);
  } catch (const std::exception& e) {
    SetErr(err, "%s", e.what());
    return nullptr;
  }
}

void jxlhip_decoder_destroy(JxlHipDecoder* dec) { delete dec; }

DecoderStatus jxlhip_peek(const uint8_t* data, size_t size, JxlHipImageInfo* info, ErrorInfo* err) {
  if (!data || !info) return DecoderStatus_NullParameter;
  try {
    ParsedFrame f;
    ParseFile(data, size, true, f);
    info->width = f.orientation >= 5 ? f.ysize : f.xsize; info->height = f.orientation >= 5 ? f.xsize : f.ysize;   // as displayed
    info->has_alpha = f.alpha_index >= 0;
    info->num_channels = f.ncolor + (f.black_index >= 0 ? 1 : 0) + info->has_alpha;
    info->bytes_per_sample = (int32_t)OutBytesPerSample(f);
    info->reserved = f.exp_bits ? 1 : 0;   // float samples
    info->xsize_blocks = f.w8; info->ysize_blocks = f.h8;
    info->num_groups = f.ng; info->num_lf_groups = f.nlf;
    info->epf_iters = f.epf_iters; info->gaborish = f.gab;
    info->codestream_bytes = f.cs_size;
    return DecoderStatus_Ok;
  } catch (const ParseError& e) {
    SetErr(err, "%s", e.what());
    return e.status;
  } catch (const std::exception& e) {
    SetErr(err, "%s", e.what());
    return DecoderStatus_DecodeError;
  }
}

DecoderStatus jxlhip_decode_batch(JxlHipDecoder* dec, int32_t n, const uint8_t* const* host_data, const size_t* sizes,
                                  const uint8_t* const* dev_data, uint8_t* const* dev_out, void* stream, int32_t synchronize,
                                  DecoderStatus* statuses, ErrorInfo* err) {
  if (!dec || !host_data || !sizes || !dev_out || n <= 0) return DecoderStatus_NullParameter;
  try {
    dec->Decode(n, host_data, sizes, dev_data, dev_out, (hipStream_t)stream, synchronize != 0, statuses, err);
    if (statuses) for (int i = 0; i < n; i++) if (statuses[i] != DecoderStatus_Ok) return statuses[i];
    return DecoderStatus_Ok;
  } catch (const std::bad_alloc&) {
    return DecoderStatus_OutOfMemory;
  } catch (const std::exception& e) {
    SetErr(err, "%s", e.what());
    return DecoderStatus_DecodeError;
  }
}

DecoderStatus jxlhip_finish(JxlHipDecoder* dec, DecoderStatus* statuses, ErrorInfo* err) {
  if (!dec) return DecoderStatus_NullParameter;
  try {
    return dec->Finish(statuses, err);
  } catch (const std::exception& e) {
    SetErr(err, "%s", e.what());
    return DecoderStatus_DecodeError;
  }
}

int32_t jxlhip_set_option(JxlHipDecoder* dec, const char* name, int32_t value) {
  if (!dec || !name) return 0;
  if (!strcmp(name, "debug_taps")) { dec->debug_taps = value != 0; return 1; }
  if (!strcmp(name, "query_pixel_chunk")) return JxlHipDecoder::kPixelChunk;
  // values a kernel's indexing depends on are validated here: a negative band start would put pixel rows in front of the caller's
  // band buffer, a stride that is not a power of two breaks the section -> lane mapping
  if (!strcmp(name, "lane_stride")) { if (value != 0 && !PowerOfTwoUpTo64(value)) return 0; dec->lane_stride_override = value; return 1; }
  if (!strcmp(name, "band_first_row")) { if (value < 0) return 0; dec->band_first_row = value; return 1; }
  if (!strcmp(name, "band_rows")) { if (value < 0) return 0; dec->band_rows = value; return 1; }
  if (!strcmp(name, "no_direct")) { dec->no_direct = value != 0; return 1; }
  if (!strcmp(name, "no_stream_pairs")) { dec->no_stream_pairs = value != 0; return 1; }
  if (!strcmp(name, "mod_lanes64")) { dec->mod_lanes64 = value != 0; return 1; }
  if (!strcmp(name, "overlap")) { dec->overlap = value != 0; return 1; }
  return 0;
}

size_t jxlhip_read_plane(JxlHipDecoder* dec, int32_t index, const char* name, int32_t channel, void* dst, size_t capacity) {
  if (!dec || !name) return 0;
  JxlHipDecoder::Slot& S = dec->Last();
  if (index < 0 || index >= S.n || S.pending) return 0;
  if (S.parse_status[index] != DecoderStatus_Ok) return 0;
  const DevImage& d = S.imgs[index];
  std::string nm(name);
  const size_t cells = (size_t)d.w8 * d.h8, pix = (size_t)d.wp * d.hp;
  const void* src = nullptr;
  size_t bytes = 0;
  bool host = false;
  const int c = std::min(2, std::max(0, channel));
  if (nm == "lf") { src = d.lf_final[c]; bytes = cells * 4; }
  else if (nm == "lf_quant") { src = d.lfq[c]; bytes = cells * 4; }
  else if (nm == "cellinfo") { src = d.cellinfo; bytes = cells * 4; }
  else if (nm == "raw_quant") { src = d.rawq; bytes = cells * 2; }
  else if (nm == "sharpness") { src = d.sharp; bytes = cells; }
  else if (nm == "ytox") { src = d.ytox; bytes = (size_t)d.wt * d.ht; }
  else if (nm == "ytob") { src = d.ytob; bytes = (size_t)d.wt * d.ht; }
  else if (nm == "alpha") { src = d.alpha; bytes = (size_t)d.w * d.h; }
  else if (nm == "inv_sigma") { src = d.inv_sigma; bytes = cells * 4; }
  else if (dec->debug_taps && (size_t)index < S.taps.size()) {
    host = true;
    if (nm == "qcoef") { src = S.taps[index].qcoef[c].data(); bytes = S.taps[index].qcoef[c].size(); }
    else if (nm == "xyb_idct") { src = S.taps[index].xyb_idct[c].data(); bytes = S.taps[index].xyb_idct[c].size(); }
    else if (nm == "xyb_filtered") { src = S.taps[index].xyb_filtered[c].data(); bytes = S.taps[index].xyb_filtered[c].size(); }
  }
  (void)pix;
  if (!src || !bytes) return 0;
  if (dst && capacity) {
    size_t nb = std::min(bytes, capacity);
    if (host) memcpy(dst, src, nb);
    else if (hipMemcpy(dst, src, nb, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  }
  return bytes;
}

int32_t jxlhip_stage_times(JxlHipDecoder* dec, const char** names, float* ms, int32_t capacity) {
  if (!dec) return 0;
  JxlHipDecoder::Slot& S = dec->Last();
  int32_t k = 0;
  for (size_t i = 1; i < S.stage_names.size() && i < S.stage_ms.size() && k < capacity; i++) {
    if (S.stage_ms[i] < 0.f) continue;
    if (names) names[k] = S.stage_names[i].c_str();
    if (ms) ms[k] = S.stage_ms[i];
    k++;
  }
  return k;
}

// Cumulative stage times over all batches finished since the last reset; returns the number of stages, *batches = batch count.
JXLFILETYPEIO_API int32_t jxlhip_stage_totals(JxlHipDecoder* dec, const char** names, float* ms, int32_t capacity, int32_t* batches,
                                              int32_t reset) {
  if (!dec) return 0;
  int32_t k = 0;
  for (size_t i = 0; i < dec->total_names.size() && k < capacity; i++, k++) {
    if (names) names[k] = dec->total_names[i].c_str();
    if (ms) ms[k] = (float)dec->total_ms[i];
  }
  if (batches) *batches = dec->total_batches;
  // the names stay (the caller reads the returned pointers after this call); only the sums restart
  if (reset) { std::fill(dec->total_ms.begin(), dec->total_ms.end(), 0.0); dec->total_batches = 0; }
  return k;
}

// ---------------------------------------------------------------------- LoadImage
static JxlHipDecoder* ThreadDecoder() {
  static thread_local std::unique_ptr<JxlHipDecoder> dec;
  if (!dec) dec.reset(new JxlHipDecoder(-1));
  return dec.get();
}

// Stage times (HIP events) of this thread's last LoadImage: what bench.py reports beside the wall time of the call
extern "C" JXLFILETYPEIO_API int32_t jxlhip_last_load_stage_times(const char** names, float* ms, int32_t capacity) {
  try { return jxlhip_stage_times(ThreadDecoder(), names, ms, capacity); } catch (...) { return 0; }
}

DecoderStatus LoadImage(DecoderCallbacks* cb, const uint8_t* data, size_t size, ErrorInfo* err) {
  if (!cb || !data) return DecoderStatus_NullParameter;   // Decoder/JxlDecoder.cpp:802-805
  DecoderStatus result = DecoderStatus_Ok;
  try {
    // ---- pass 1: basic info, colour profile, metadata boxes (Decoder/JxlDecoder.cpp:412-793)
    ParsedFrame f;
    ParseFile(data, size, true, f);
    if (f.xsize > 0x7FFFFFFFu || f.ysize > 0x7FFFFFFFu) return DecoderStatus_ImageDimensionExceedsInt32;   // :477-481
    int black = 0, alphas = 0;
    for (auto& e : f.ec) { if (e.type == 4) black++; if (e.type == 0) alphas++; }
    if ((f.ncolor != 1 && f.ncolor != 3) || black > 1 || alphas > 1) return DecoderStatus_UnsupportedChannelFormat;   // :485-490
    const bool has_alpha = f.alpha_index >= 0;
    const bool cmyk = black == 1;   // ExtraChannelsAreSupported + DecoderImageFormat::Cmyk (:110-157, :499-503)
    // sample type by bit depth (Decoder/JxlDecoder.cpp:510-556)
    int rep = ImageChannelRepresentation_Uint8;
    if (f.exp_bits > 0) {
      if (f.bits <= 16) rep = ImageChannelRepresentation_Float16;        // :521-525
      else if (f.bits <= 32) rep = ImageChannelRepresentation_Float32;   // :526-530
      else { SetErr(err, "Unsupported floating point bit depth: %u.", f.bits); return DecoderStatus_DecodeError; }   // :531-535
    } else if (f.bits > 8) {
      if (f.bits > 16) { SetErr(err, "Unsupported integer bit depth: %u.", f.bits); return DecoderStatus_DecodeError; }   // :551
      rep = ImageChannelRepresentation_Uint16;
    }
    const bool swap_sides = f.orientation >= 5;   // the host is told the size as displayed
    cb->setBasicInfo((int32_t)(swap_sides ? f.ysize : f.xsize), (int32_t)(swap_sides ? f.xsize : f.ysize),
                     cmyk ? DecoderImageFormat_Cmyk : (f.ncolor == 1 ? DecoderImageFormat_Gray : DecoderImageFormat_Rgb), (ImageChannelRepresentation)rep,
                     has_alpha);   // :558
    // colour encoding -> KnownColorProfile (Decoder/JxlDecoder.cpp:36-108); anything else would take the reference's ICC route
    {
      const ColorPlan plan = PlanColor(f);
      if (plan.report_icc) {   // the target-data profile is the embedded ICC profile (:652-682)
        const std::vector<uint8_t>& prof = f.icc.empty() ? plan.icc_out : f.icc;   // embedded, or synthesised for the enumerated encoding
        if (!prof.empty() && !cb->setIccProfile(const_cast<uint8_t*>(prof.data()), prof.size())) return DecoderStatus_CreateMetadataError;
      } else if (plan.known_profile < 0) {
        SetErr(err, "This colour encoding needs a synthesised ICC profile, which the GPU path does not build yet.");
        return DecoderStatus_DecodeError;
      } else if (!cb->setKnownColorProfile((KnownColorProfile)plan.known_profile)) {
        return DecoderStatus_CreateMetadataError;   // :648-651
      }
    }
    // metadata boxes in file order, whatever their payload size (the reference reports a box when the library completes it, :756-782)
    static uint8_t empty_payload[1] = {0};
    for (auto& b : f.meta_in_order) {
      uint8_t* p = b.size ? const_cast<uint8_t*>(b.data) : empty_payload;
      if (!(b.is_exif ? cb->setExif(p, b.size) : cb->setXmp(p, b.size))) return DecoderStatus_CreateMetadataError;
    }
    // ---- pass 2: the frame (Decoder/JxlDecoder.cpp:217-410)
    JxlHipDecoder* dec = ThreadDecoder();
    const int nch = f.ncolor + (cmyk ? 1 : 0) + (has_alpha ? 1 : 0);   // CMYK: C M Y K [A], inverted for the host on the device (:159-215)
    const size_t bytes = (size_t)f.xsize * f.ysize * nch * OutBytesPerSample(f);   // tightly packed, :291-313
    dec->EnsureLoadImageBuffers(bytes);
    uint8_t* const d_out = dec->li_dev;
    uint8_t* const h_out = dec->li_host;   // valid for the duration of the setLayerData call, like the reference's buffer (:291-313)
    DecoderStatus st = DecoderStatus_Ok;
    const uint8_t* hd = data;
    uint8_t* od = d_out;
    dec->Decode(1, &hd, &size, nullptr, &od, nullptr, true, &st, err);
    if (st != DecoderStatus_Ok) result = st;
    else {
      HIP_OK(hipMemcpy(h_out, d_out, bytes, hipMemcpyDeviceToHost));
      std::vector<char> name;
      if (!f.name.empty()) { name.assign(f.name.begin(), f.name.end()); name.push_back(0); }   // nameLength includes the NUL (:274,369)
      if (!cb->setLayerData(h_out, name.empty() ? nullptr : name.data(), name.size())) result = DecoderStatus_CreateLayerError;   // :384-395
    }
  } catch (const ParseError& e) {
    SetErr(err, "%s", e.what());
    result = e.status;
  } catch (const std::bad_alloc&) {
    result = DecoderStatus_OutOfMemory;
  } catch (const std::exception& e) {
    SetErr(err, "%s", e.what());
    result = DecoderStatus_DecodeError;
  } catch (...) {
    result = DecoderStatus_DecodeError;
  }
  return result;
}

// SaveImage lives in encoder.cc.

}  // extern "C"

// ---------------------------------------------------------------------- host-only introspection (CPU tests)
extern "C" {

// Metadata boxes as LoadImage would hand them to the host (no GPU): which = 0 Exif, k >= 1 the k-th `xml ` box.  Returns the payload
// size (0: absent) and copies up to `capacity` bytes.
JXLFILETYPEIO_API size_t jxlhip_parse_metadata(const uint8_t* data, size_t size, int32_t which, uint8_t* dst, size_t capacity, DecoderStatus* status,
                                               ErrorInfo* err) {
  if (status) *status = DecoderStatus_Ok;
  if (!data) { if (status) *status = DecoderStatus_NullParameter; return 0; }
  try {
    ParsedFrame f;
    ParseFile(data, size, true, f);
    const uint8_t* p = nullptr;
    size_t n = 0;
    if (which == 0) { p = f.exif; n = f.exif_size; }
    else if (which >= 1 && (size_t)which <= f.xml.size()) { p = f.xml[which - 1].first; n = f.xml[which - 1].second; }
    if (p && dst && capacity) memcpy(dst, p, std::min(n, capacity));
    return p ? n : 0;
  } catch (const ParseError& e) {
    SetErr(err, "%s", e.what());
    if (status) *status = (DecoderStatus)e.status;
  } catch (const std::exception& e) {
    SetErr(err, "%s", e.what());
    if (status) *status = DecoderStatus_DecodeError;
  }
  return 0;
}

// Full host-side parse (no GPU): returns the status and a few facts about the parsed tables.
// facts[0..7] = tree nodes, modular clusters, modular log_alpha, AC clusters, AC log_alpha, AC contexts, presets, sections
JXLFILETYPEIO_API DecoderStatus jxlhip_parse_check(const uint8_t* data, size_t size, int32_t* facts, ErrorInfo* err) {
  if (!data) return DecoderStatus_NullParameter;
  try {
    ParsedFrame f;
    ParseFile(data, size, false, f);
    if (facts) {
      facts[0] = (int32_t)f.tree.size(); facts[1] = f.mcode.num_hist; facts[2] = f.mcode.log_alpha;
      facts[3] = f.acode.num_hist; facts[4] = f.acode.log_alpha; facts[5] = (int32_t)f.acode.ctx_map.size();
      facts[6] = f.num_presets; facts[7] = (int32_t)f.sec_off.size();
    }
    return DecoderStatus_Ok;
  } catch (const ParseError& e) {
    SetErr(err, "%s", e.what());
    return e.status;
  } catch (const std::exception& e) {
    SetErr(err, "%s", e.what());
    return DecoderStatus_DecodeError;
  }
}

// Host-only: the embedded ICC profile as LoadImage would hand it to setIccProfile (0: the stream has none).
JXLFILETYPEIO_API size_t jxlhip_parse_icc(const uint8_t* data, size_t size, uint8_t* dst, size_t capacity, DecoderStatus* status, ErrorInfo* err) {
  if (status) *status = DecoderStatus_Ok;
  if (!data) { if (status) *status = DecoderStatus_NullParameter; return 0; }
  try {
    ParsedFrame f;
    ParseFile(data, size, true, f);
    if (dst && capacity) memcpy(dst, f.icc.data(), std::min(f.icc.size(), capacity));
    return f.icc.size();
  } catch (const ParseError& e) {
    SetErr(err, "%s", e.what());
    if (status) *status = (DecoderStatus)e.status;
  } catch (const std::exception& e) {
    SetErr(err, "%s", e.what());
    if (status) *status = DecoderStatus_DecodeError;
  }
  return 0;
}

// Host-only test hooks of icc.cc: the predicted stream -> profile (returns the size, 0 on failure), and the colour model of a
// matrix / TRC profile (model[0..8] linear sRGB -> profile RGB, model[9..17] the inverse; returns 1 if the profile is of that kind).
JXLFILETYPEIO_API size_t jxlhip_icc_unpredict(const uint8_t* enc, size_t size, uint8_t* dst, size_t capacity, ErrorInfo* err) {
  std::vector<uint8_t> e(enc, enc + size), out;
  std::string why;
  if (!IccUnpredict(e, &out, &why)) { SetErr(err, "%s", why.c_str()); return 0; }
  if (dst && capacity) memcpy(dst, out.data(), std::min(out.size(), capacity));
  return out.size();
}
JXLFILETYPEIO_API int32_t jxlhip_icc_model(const uint8_t* icc, size_t size, double* model, float* to_linear, float* from_linear) {
  IccModel m;
  if (!IccBuildModel(icc, size, &m)) return 0;
  for (int k = 0; k < 9; k++) { model[k] = m.from_linear_srgb[k]; model[9 + k] = m.to_linear_srgb[k]; }
  if (to_linear) for (int c = 0; c < 3; c++) memcpy(to_linear + 256 * c, m.to_linear[c].data(), 256 * 4);
  if (from_linear) for (int c = 0; c < 3; c++) memcpy(from_linear + kIccInvLut * c, m.from_linear[c].data(), kIccInvLut * 4);
  return m.gray ? 2 : 1;
}

// Host-only: byte sizes of the TOC sections in logical order (LfGlobal, LF groups, HfGlobal, pass groups).  Returns their number.
JXLFILETYPEIO_API int32_t jxlhip_section_sizes(const uint8_t* data, size_t size, uint32_t* dst, int32_t capacity) {
  if (!data) return 0;
  try {
    ParsedFrame f;
    ParseFile(data, size, true, f);
    const int32_t n = (int32_t)f.sec_size.size();
    for (int32_t i = 0; i < n && i < capacity && dst; i++) dst[i] = f.sec_size[i];
    return n;
  } catch (...) {
    return 0;
  }
}

// name: "natural_order" (index = order bucket, uint16), "dequant" (index = quant table, float, 3*n),
//       "basis" (index = log2(N/8), float N*N).  Returns the byte size.
JXLFILETYPEIO_API size_t jxlhip_static_table(const char* name, int32_t index, void* dst, size_t capacity) {
  const StaticTables& st = GetStaticTables();
  const void* src = nullptr;
  size_t bytes = 0;
  std::string nm(name ? name : "");
  if (nm == "natural_order" && index >= 0 && index < kNumOrders) { src = st.natural_order[index].data(); bytes = st.natural_order[index].size() * 2; }
  else if (nm == "dequant" && index >= 0 && index < kNumQuantTables) { src = st.dq[index].data(); bytes = st.dq[index].size() * 4; }
  else if (nm == "basis" && index >= 0 && index < 6) { src = st.basis[index].data(); bytes = st.basis[index].size() * 4; }
  if (src && dst) memcpy(dst, src, std::min(bytes, capacity));
  return bytes;
}

}  // extern "C"
