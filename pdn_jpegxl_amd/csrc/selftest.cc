// Host-only self tests of the writers (CPU tests, no GPU): the encoder's code builder, MA-tree writer and header writers against the
// decoder's parsers.  TEST INFRASTRUCTURE: compiled into lib/libjxlhip_selftest.so only (build.py), never into the library the
// reference's host loads - the production library exports GetLibJxlVersion / LoadImage / SaveImage and the jxlhip_* batch API, no hooks.
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/jxlfiletypeio.h"
#include "enc_types.h"
#include "host_parse.h"
#include "host_write.h"
#include "icc.h"

namespace jxlhip {
std::vector<EncTreeNode> MakeEncoderTree(uint32_t nlf);
}
using namespace jxlhip;

static void SetEncErr(ErrorInfo* e, const char* msg) {
  if (!e || !msg) return;
  size_t n = strlen(msg);
  if (n == 0) return;
  if (n > 255) n = 255;
  memcpy(e->errorMessage, msg, n);
  e->errorMessage[n] = 0;
}

namespace jxlhip {
bool ReadBackTokens(const uint8_t* bytes, size_t nbytes, size_t num_ctx, const uint32_t* ctxs, const uint32_t* values, size_t n, std::string* why);
bool ReadBackTree(const uint8_t* bytes, size_t nbytes, std::vector<DevTreeNode>* tree, std::string* why);
}

// Writes n pseudo-random tokens over num_ctx contexts with the encoder's code builder and reads them back with the decoder's
// header parser and symbol reader.  Returns 0 on success (message in err otherwise).
extern "C" JXLFILETYPEIO_API int32_t jxlhip_selftest_entropy(uint32_t seed, uint32_t num_ctx, uint32_t n, int32_t max_clusters,
                                                             uint32_t pinned_ctx_plus1, ErrorInfo* err) {
  try {
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 12345;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 32); };
    std::vector<uint32_t> ctxs(n), vals(n), hist((size_t)num_ctx * kEncAlphabet, 0);
    std::vector<uint8_t> pinned(num_ctx, 0);
    if (pinned_ctx_plus1) pinned[pinned_ctx_plus1 - 1] = 1;
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t c = rnd() % num_ctx;
      // geometric-ish magnitudes whose spread depends on the context; a few large outliers exercise the extra bits
      uint32_t v = 0;
      const uint32_t spread = 1 + c % 7;
      while ((rnd() % (spread + 1)) != 0 && v < 40) v++;
      if (rnd() % 97 == 0) v = rnd() >> (rnd() % 30);
      if (pinned[c]) v = 0;
      ctxs[i] = c; vals[i] = v;
      uint32_t tok, nb, bits;
      HybridEncode(v, &tok, &nb, &bits);
      hist[(size_t)c * kEncAlphabet + tok]++;
    }
    BitWriter bw;
    EncCode code;
    BuildAndWriteCode(hist.data(), num_ctx, max_clusters, pinned, bw, code);
    std::vector<EncToken> toks;
    for (uint32_t i = 0; i < n; i++)
      if (!pinned[ctxs[i]]) toks.push_back(EncToken{ctxs[i], vals[i]});   // tokens of pinned contexts are never written
    WriteTokensHost(toks, code, bw);
    std::vector<uint8_t> bytes = bw.Finish();
    std::string why;
    // the decoder reads every token, the pinned ones included (they cost no bits and leave the state alone)
    if (!ReadBackTokens(bytes.data(), bytes.size(), num_ctx, ctxs.data(), vals.data(), n, &why)) { SetEncErr(err, why.c_str()); return 1; }
    return 0;
  } catch (const std::exception& e) {
    SetEncErr(err, e.what());
    return 2;
  }
}

// Serialises the encoder's MA tree and parses it back; returns 0 when node for node identical.
extern "C" JXLFILETYPEIO_API int32_t jxlhip_selftest_tree(uint32_t nlf, ErrorInfo* err) {
  try {
    const std::vector<EncTreeNode> t = MakeEncoderTree(nlf);
    BitWriter bw;
    WriteTree(t, bw);
    std::vector<uint8_t> bytes = bw.Finish();
    std::vector<DevTreeNode> back;
    std::string why;
    if (!ReadBackTree(bytes.data(), bytes.size(), &back, &why)) { SetEncErr(err, why.c_str()); return 1; }
    if (back.size() != t.size()) { SetEncErr(err, "node count"); return 2; }
    uint32_t leaf = 0;
    for (size_t i = 0; i < t.size(); i++) {
      if (t[i].property >= 0) {
        if (back[i].property != t[i].property || back[i].splitval != t[i].splitval) { SetEncErr(err, "inner node"); return 3; }
      } else {
        if (back[i].property >= 0 || (back[i].a & 0xFF) != (uint32_t)t[i].pred || (back[i].a >> 8) != leaf || back[i].splitval != t[i].offset ||
            back[i].b != t[i].multiplier) { SetEncErr(err, "leaf"); return 4; }
        leaf++;
      }
    }
    return leaf == kNumEncLeaves ? 0 : 5;
  } catch (const std::exception& e) {
    SetEncErr(err, e.what());
    return 6;
  }
}

// Writes the codestream headers + frame header + TOC the encoder would emit for the given geometry (sections of `sec_bytes` bytes each)
// into dst; the CPU tests read them back with jxlhip_peek (ParseFile, headers only).  Returns the byte count (0 on failure).
extern "C" JXLFILETYPEIO_API size_t jxlhip_selftest_headers(uint32_t xsize, uint32_t ysize, int32_t gray, int32_t alpha, int32_t lossless,
                                                            uint32_t epf_iters, uint32_t sec_bytes, uint8_t* dst, size_t capacity) {
  try {
    EncImageInfo ii;
    ii.xsize = xsize; ii.ysize = ysize; ii.gray = gray != 0; ii.alpha = alpha != 0; ii.xyb = !lossless;
    EncFrameInfo fi;
    fi.encoding = lossless ? 1 : 0;
    fi.gab = !lossless; fi.epf_iters = lossless ? 0 : epf_iters;
    BitWriter cs;
    WriteCodestreamHeaders(ii, cs);
    WriteFrameHeader(ii, fi, cs);
    const uint32_t ng = ((xsize + 255) / 256) * ((ysize + 255) / 256), nlf = ((xsize + 2047) / 2048) * ((ysize + 2047) / 2048);
    std::vector<uint32_t> sizes(ng == 1 ? 1 : 2 + nlf + ng, sec_bytes);
    WriteToc(sizes, cs);
    std::vector<uint8_t> bytes = cs.Finish();
    bytes.resize(bytes.size() + (size_t)sec_bytes * sizes.size(), 0);
    std::vector<uint8_t> file = WriteContainer(bytes, nullptr, 0, nullptr, 0);
    if (file.size() > capacity) return 0;
    memcpy(dst, file.data(), file.size());
    return file.size();
  } catch (...) {
    return 0;
  }
}

// The same with an embedded ICC profile (host only: the ICC stream writer against the parser's reader).
extern "C" JXLFILETYPEIO_API size_t jxlhip_selftest_headers_icc(uint32_t xsize, uint32_t ysize, int32_t alpha, int32_t lossless, const uint8_t* icc,
                                                                size_t icc_size, uint8_t* dst, size_t capacity) {
  try {
    EncImageInfo ii;
    ii.xsize = xsize; ii.ysize = ysize; ii.gray = false; ii.alpha = alpha != 0; ii.xyb = !lossless;
    ii.icc = icc; ii.icc_size = icc_size;
    EncFrameInfo fi;
    fi.encoding = lossless ? 1 : 0;
    fi.gab = !lossless; fi.epf_iters = lossless ? 0 : 1;
    BitWriter cs;
    WriteCodestreamHeaders(ii, cs);
    WriteFrameHeader(ii, fi, cs);
    const uint32_t ng = ((xsize + 255) / 256) * ((ysize + 255) / 256), nlf = ((xsize + 2047) / 2048) * ((ysize + 2047) / 2048);
    std::vector<uint32_t> sizes(ng == 1 ? 1 : 2 + nlf + ng, 3);
    WriteToc(sizes, cs);
    std::vector<uint8_t> bytes = cs.Finish();
    bytes.resize(bytes.size() + 3 * sizes.size(), 0);
    std::vector<uint8_t> file = WriteContainer(bytes, nullptr, 0, nullptr, 0);
    if (file.size() > capacity) return 0;
    memcpy(dst, file.data(), file.size());
    return file.size();
  } catch (...) {
    return 0;
  }
}
