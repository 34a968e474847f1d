// Host-side writers of the encode path (see host_write.h).  Restates the header syntax of ISO/IEC 18181-1 in the
// write direction; every field order mirrors the reader in host_parse.cc.
#include "host_write.h"
#include "icc.h"
#include <algorithm>
#include <cmath>
#include <mutex>
#include <cstring>
#include <stdexcept>

namespace jxlhip {

namespace {
inline int FloorLog2W(uint64_t x) { return 63 - __builtin_clzll(x | 1); }
inline int CeilLog2W(uint64_t x) { return x <= 1 ? 0 : FloorLog2W(x - 1) + 1; }
inline uint32_t PackSignedW(int32_t v) { return v >= 0 ? (uint32_t)v << 1 : (((uint32_t)(-(int64_t)v)) << 1) - 1; }
}  // namespace

// ------------------------------------------------------------------ bit writer
void BitWriter::Write(int nbits, uint64_t value) {
  if (nbits <= 0) return;
  if (nbits < 64) value &= ((uint64_t)1 << nbits) - 1;
  acc_ |= value << nbits_;
  nbits_ += nbits;
  while (nbits_ >= 8) {
    bytes_.push_back((uint8_t)acc_);
    acc_ >>= 8;
    nbits_ -= 8;
  }
}

void BitWriter::U32(Dist a, Dist b, Dist c, Dist d, uint32_t value) {
  const Dist t[4] = {a, b, c, d};
  for (int s = 0; s < 4; s++) {
    if (t[s].bits < 0) {
      if (value == t[s].off) { Write(2, s); return; }
    } else if (value >= t[s].off && (uint64_t)(value - t[s].off) < ((uint64_t)1 << t[s].bits)) {
      Write(2, s);
      Write(t[s].bits, value - t[s].off);
      return;
    }
  }
  throw std::runtime_error("U32: value not representable");
}

void BitWriter::U64(uint64_t v) {
  if (v == 0) { Write(2, 0); return; }
  if (v <= 16) { Write(2, 1); Write(4, v - 1); return; }
  if (v <= 272) { Write(2, 2); Write(8, v - 17); return; }
  Write(2, 3);
  Write(12, v & 0xFFF);
  v >>= 12;
  int shift = 12;
  while (v) {
    Write(1, 1);
    if (shift == 60) { Write(4, v & 0xF); return; }
    Write(8, v & 0xFF);
    v >>= 8;
    shift += 8;
  }
  Write(1, 0);
}

void BitWriter::Enum(uint32_t v) { U32(WV(0), WV(1), WB(4, 2), WB(6, 18), v); }

void BitWriter::AlignByte() {
  if (nbits_) Write(8 - nbits_, 0);
}

void BitWriter::AppendBits(const uint8_t* bytes, uint64_t nbits) {
  uint64_t i = 0;
  for (; i + 8 <= nbits; i += 8) Write(8, bytes[i >> 3]);
  if (i < nbits) Write((int)(nbits - i), bytes[i >> 3]);
}

std::vector<uint8_t> BitWriter::Finish() {
  AlignByte();
  return bytes_;
}

// ------------------------------------------------------------------ entropy-code headers
namespace {

void WriteVarLen8(BitWriter& bw, uint32_t n) {
  if (n == 0) { bw.Write(1, 0); return; }
  bw.Write(1, 1);
  const uint32_t nb = (uint32_t)FloorLog2W(n);
  bw.Write(3, nb);
  bw.Write((int)nb, n - (1u << nb));
}

// counts (raw) -> frequencies summing to 4096, every used symbol >= 1
std::vector<int> Normalise(const uint64_t* h, size_t n) {
  std::vector<int> out(n, 0);
  uint64_t total = 0;
  for (size_t i = 0; i < n; i++) total += h[i];
  if (!total) return out;
  int64_t sum = 0;
  for (size_t i = 0; i < n; i++) {
    if (!h[i]) continue;
    int64_t v = (int64_t)((h[i] * 4096 + total / 2) / total);
    if (v < 1) v = 1;
    out[i] = (int)v;
    sum += v;
  }
  while (sum != 4096) {   // push the rounding error onto the most frequent symbols
    size_t best = 0;
    for (size_t i = 1; i < n; i++) if (out[i] > out[best]) best = i;
    int64_t delta = 4096 - sum;
    if (delta < 0 && out[best] + delta < 1) delta = 1 - out[best];
    if (delta == 0) throw std::runtime_error("histogram normalisation failed");
    out[best] += (int)delta;
    sum += delta;
  }
  return out;
}

// log-count prefix code of the general distribution form (inverse of the reader's 7-bit lookup table)
struct LogCountCode {
  uint8_t len[14], bits[14];
  LogCountCode() {
    static const uint8_t kLen[128] = {
        3, 7, 3, 4, 3, 3, 3, 4, 3, 4, 3, 4, 3, 3, 3, 4, 3, 5, 3, 4, 3, 3, 3, 4, 3, 4, 3, 4, 3, 3, 3, 4,
        3, 6, 3, 4, 3, 3, 3, 4, 3, 4, 3, 4, 3, 3, 3, 4, 3, 5, 3, 4, 3, 3, 3, 4, 3, 4, 3, 4, 3, 3, 3, 4,
        3, 7, 3, 4, 3, 3, 3, 4, 3, 4, 3, 4, 3, 3, 3, 4, 3, 5, 3, 4, 3, 3, 3, 4, 3, 4, 3, 4, 3, 3, 3, 4,
        3, 6, 3, 4, 3, 3, 3, 4, 3, 4, 3, 4, 3, 3, 3, 4, 3, 5, 3, 4, 3, 3, 3, 4, 3, 4, 3, 4, 3, 3, 3, 4};
    static const uint8_t kSym[128] = {
        10, 12, 7, 3, 6, 8, 9, 5, 10, 4, 7, 1, 6, 8, 9, 2, 10, 0, 7, 3, 6, 8, 9, 5, 10, 4, 7, 1, 6, 8, 9, 2,
        10, 11, 7, 3, 6, 8, 9, 5, 10, 4, 7, 1, 6, 8, 9, 2, 10, 0, 7, 3, 6, 8, 9, 5, 10, 4, 7, 1, 6, 8, 9, 2,
        10, 13, 7, 3, 6, 8, 9, 5, 10, 4, 7, 1, 6, 8, 9, 2, 10, 0, 7, 3, 6, 8, 9, 5, 10, 4, 7, 1, 6, 8, 9, 2,
        10, 11, 7, 3, 6, 8, 9, 5, 10, 4, 7, 1, 6, 8, 9, 2, 10, 0, 7, 3, 6, 8, 9, 5, 10, 4, 7, 1, 6, 8, 9, 2};
    for (int s = 0; s < 14; s++)
      for (int idx = 0; idx < 128; idx++)
        if (kSym[idx] == s) { len[s] = kLen[idx]; bits[s] = (uint8_t)(idx & ((1 << kLen[idx]) - 1)); break; }
  }
};

void WriteDistribution(BitWriter& bw, const std::vector<int>& counts) {
  std::vector<int> syms;
  for (size_t i = 0; i < counts.size(); i++) if (counts[i]) syms.push_back((int)i);
  if (syms.size() <= 2) {
    bw.Write(1, 1);
    if (syms.empty()) { bw.Write(1, 0); WriteVarLen8(bw, 0); return; }
    bw.Write(1, syms.size() - 1);
    for (int s : syms) WriteVarLen8(bw, (uint32_t)s);
    if (syms.size() == 2) bw.Write(12, (uint64_t)counts[syms[0]]);
    return;
  }
  static const LogCountCode code;
  bw.Write(1, 0);   // not the 1-2 symbol form
  bw.Write(1, 0);   // not flat
  bw.Write(3, 7);   // unary 111: three bits follow for the shift
  bw.Write(3, 6);   // shift = (6 | 8) - 1 = 13: every count at full precision
  const int len = syms.back() + 1;
  WriteVarLen8(bw, (uint32_t)(len - 3));
  std::vector<int> lc(len);
  int omit = -1, omit_lc = -1;
  for (int i = 0; i < len; i++) {
    lc[i] = counts[i] ? FloorLog2W((uint64_t)counts[i]) + 1 : 0;
    if (lc[i] > omit_lc) { omit_lc = lc[i]; omit = i; }
  }
  for (int i = 0; i < len; i++) bw.Write(code.len[lc[i]], code.bits[lc[i]]);
  for (int i = 0; i < len; i++) {
    if (i == omit || lc[i] <= 1) continue;
    bw.Write(lc[i] - 1, (uint64_t)(counts[i] - (1 << (lc[i] - 1))));
  }
}

// x * log2(x): the clustering below evaluates it millions of times (64 seeds x thousands of contexts x the alphabet); counts are
// small integers almost always, so a table replaces the library call (the code construction was 19 ms of a 4K save, 15 ms of a
// 200 x 150 one)
double XLog2X(uint64_t x) {
  static std::vector<double> table;
  static std::once_flag once;
  std::call_once(once, [] {
    table.resize(1 << 16);
    table[0] = 0;
    for (size_t i = 1; i < table.size(); i++) table[i] = (double)i * std::log2((double)i);
  });
  return x < table.size() ? table[x] : (double)x * std::log2((double)x);
}

struct Hist {
  std::vector<uint64_t> c;
  std::vector<uint16_t> nz;   // the symbols that occur (most contexts use a handful of the alphabet)
  uint64_t total = 0;
  double sumx = 0;   // sum c log2(c)
  double bits = 0;   // sum -c log2(c / total) = total log2(total) - sum c log2(c)
  void Finish() {
    total = 0;
    sumx = 0;
    nz.clear();
    for (size_t i = 0; i < c.size(); i++) if (c[i]) { total += c[i]; sumx += XLog2X(c[i]); nz.push_back((uint16_t)i); }
    bits = XLog2X(total) - sumx;
  }
};

// extra bits when both are coded with their merged distribution; nsym: symbols that occur at all (the tail of the alphabet is empty)
// (only the symbols of `a` change b's sum of c log2 c: the cost is linear in a's support, not in the alphabet)
double JoinCost(const Hist& a, const Hist& b) {
  if (!a.total || !b.total) return 0;
  double merged = b.sumx;
  for (uint16_t i : a.nz) merged += XLog2X(a.c[i] + b.c[i]) - XLog2X(b.c[i]);
  return XLog2X(a.total + b.total) - merged - a.bits - b.bits;
}

}  // namespace

void WriteTokensHost(const std::vector<EncToken>& tokens, const EncCode& code, BitWriter& bw) {
  const size_t n = tokens.size();
  std::vector<uint32_t> flush(n, 0);
  uint32_t state = 0x130000u;
  for (size_t r = n; r-- > 0;) {
    const uint32_t cl = code.ctx_map[tokens[r].ctx];
    uint32_t tok, nb, bits;
    HybridEncode(tokens[r].value, &tok, &nb, &bits);
    const uint32_t freq = code.freq[cl * kEncAlphabet + tok];
    if (!freq) throw std::runtime_error("token missing from its distribution");
    if ((state >> 20) >= freq) { flush[r] = 0x10000u | (state & 0xFFFF); state >>= 16; }
    state = ((state / freq) << 12) + code.rmap[(size_t)cl * 4096 + code.start[cl * kEncAlphabet + tok] + state % freq];
  }
  bw.Write(32, state);
  for (size_t i = 0; i < n; i++) {
    uint32_t tok, nb, bits;
    HybridEncode(tokens[i].value, &tok, &nb, &bits);
    if (flush[i]) bw.Write(16, flush[i] & 0xFFFF);
    bw.Write((int)nb, bits);
  }
}

void BuildAndWriteCode(const uint32_t* hist, size_t num_ctx, int max_clusters, const std::vector<uint8_t>& pinned_zero, BitWriter& bw,
                       EncCode& out) {
  out = EncCode();
  const size_t A = kEncAlphabet;
  std::vector<Hist> h(num_ctx);
  std::vector<size_t> used;
  bool any_pinned = false;
  uint32_t max_sym = 0;
  for (size_t i = 0; i < num_ctx; i++) {
    h[i].c.assign(hist + i * A, hist + (i + 1) * A);
    h[i].Finish();
    const bool pinned = i < pinned_zero.size() && pinned_zero[i];
    any_pinned |= pinned;
    if (pinned) continue;
    if (h[i].total) used.push_back(i);
    for (size_t s = 0; s < A; s++) if (h[i].c[s]) max_sym = std::max<uint32_t>(max_sym, (uint32_t)s);
  }
  // ---- clustering: farthest-point seeds in join-cost distance, every context joins its cheapest seed
  size_t budget = (size_t)std::max(1, std::min(max_clusters, 255)) - (any_pinned ? 1 : 0);
  if (budget < 1) budget = 1;
  std::vector<Hist> clusters;
  out.ctx_map.assign(num_ctx, 0);
  if (used.size() <= budget) {
    for (size_t k = 0; k < used.size(); k++) { out.ctx_map[used[k]] = (uint8_t)k; clusters.push_back(h[used[k]]); }
  } else {
    std::vector<size_t> seeds;
    size_t first = used[0];
    for (size_t i : used) if (h[i].total > h[first].total) first = i;
    seeds.push_back(first);
    // dmin / best: every context's cheapest seed so far - the distances of the seed search ARE the assignment (the first cheapest
    // seed wins), so each (context, seed) pair is evaluated once
    std::vector<double> dmin(num_ctx, 1e300);
    std::vector<uint8_t> best(num_ctx, 0);
    dmin[first] = -1.0;
    for (;;) {
      const size_t k = seeds.size() - 1, s = seeds[k];
      size_t far = used[0];
      double fard = -1;
      for (size_t i : used) {
        if (i != s) {
          const double d = JoinCost(h[i], h[s]);
          if (d < dmin[i]) { dmin[i] = d; best[i] = (uint8_t)k; }
        }
        if (dmin[i] > fard) { fard = dmin[i]; far = i; }
      }
      if (seeds.size() >= budget || fard <= 0) break;
      seeds.push_back(far);
      dmin[far] = -1.0;
      best[far] = (uint8_t)(seeds.size() - 1);
    }
    clusters.assign(seeds.size(), Hist());
    for (auto& c : clusters) c.c.assign(A, 0);
    for (size_t i : used) {
      out.ctx_map[i] = best[i];
      for (uint16_t s : h[i].nz) clusters[best[i]].c[s] += h[i].c[s];
    }
  }
  if (any_pinned) {
    Hist z;
    z.c.assign(A, 0);
    z.c[0] = 1;
    for (size_t i = 0; i < num_ctx; i++) if (i < pinned_zero.size() && pinned_zero[i]) out.ctx_map[i] = (uint8_t)clusters.size();
    clusters.push_back(z);
  }
  if (clusters.empty()) { Hist z; z.c.assign(A, 0); clusters.push_back(z); }
  out.num_clusters = (uint32_t)clusters.size();
  out.log_alpha = (uint32_t)std::max(5, CeilLog2W(max_sym + 1));
  if (out.log_alpha > 8) throw std::runtime_error("token alphabet exceeds the ANS table");
  // ---- header
  bw.Write(1, 0);   // no LZ77
  if (num_ctx > 1) {
    const int nb = CeilLog2W(out.num_clusters);
    if (nb <= 3 && (size_t)nb * num_ctx <= 512) {
      bw.Write(1, 1);
      bw.Write(2, nb);
      for (auto m : out.ctx_map) bw.Write(nb, m);
    } else {
      bw.Write(1, 0);   // entropy-coded map
      bw.Write(1, 0);   // no move-to-front
      std::vector<uint32_t> mh(A, 0);
      std::vector<EncToken> mt;
      for (auto m : out.ctx_map) {
        uint32_t tok, nbits, bits;
        HybridEncode(m, &tok, &nbits, &bits);
        mh[tok]++;
        mt.push_back(EncToken{0, m});
      }
      EncCode mc;
      BuildAndWriteCode(mh.data(), 1, 1, {}, bw, mc);
      WriteTokensHost(mt, mc, bw);
    }
  }
  bw.Write(1, 0);   // ANS, not prefix codes
  bw.Write(2, out.log_alpha - 5);
  for (uint32_t k = 0; k < out.num_clusters; k++) {
    bw.Write(CeilLog2W(out.log_alpha + 1), kEncSplitExp);
    if (kEncSplitExp != out.log_alpha) {
      bw.Write(CeilLog2W(kEncSplitExp + 1), kEncMsb);
      bw.Write(CeilLog2W(kEncSplitExp - kEncMsb + 1), kEncLsb);
    }
  }
  out.freq.assign((size_t)out.num_clusters * A, 0);
  out.start.assign((size_t)out.num_clusters * A, 0);
  out.rmap.assign((size_t)out.num_clusters * 4096, 0);
  const uint32_t T = 1u << out.log_alpha, le = 12 - out.log_alpha;
  std::vector<uint64_t> alias(T);
  for (uint32_t k = 0; k < out.num_clusters; k++) {
    std::vector<int> counts = Normalise(clusters[k].c.data(), A);
    while (!counts.empty() && counts.back() == 0) counts.pop_back();
    WriteDistribution(bw, counts);
    if (counts.empty()) counts.assign(1, 4096);   // what the decoder reconstructs for an empty distribution
    uint32_t acc = 0;
    for (size_t s = 0; s < counts.size(); s++) {
      out.freq[k * A + s] = (uint16_t)counts[s];
      out.start[k * A + s] = (uint16_t)acc;
      acc += counts[s];
    }
    BuildAliasTable(counts, out.log_alpha, alias.data());
    for (uint32_t res = 0; res < 4096; res++) {
      const uint32_t i = res >> le, pos = res & ((1u << le) - 1);
      const uint64_t e = alias[i];
      const uint32_t x = (uint32_t)e, y = (uint32_t)(e >> 32);
      const bool g = pos >= (x & 0xFF);
      const uint32_t sym = g ? ((x >> 8) & 0xFF) : i;
      const uint32_t off = g ? (y & 0xFFFF) + pos : pos;
      out.rmap[(size_t)k * 4096 + ((out.start[k * A + sym] + off) & 4095)] = (uint16_t)res;
    }
  }
}

// ------------------------------------------------------------------ MA tree
void WriteTree(const std::vector<EncTreeNode>& nodes, BitWriter& bw) {
  // contexts of the tree stream: 0 split value, 1 property + 1, 2 predictor, 3 offset, 4 multiplier log, 5 multiplier bits
  std::vector<EncToken> t;
  for (const EncTreeNode& n : nodes) {
    if (n.property >= 0) {
      t.push_back(EncToken{1, (uint32_t)n.property + 1});
      t.push_back(EncToken{0, PackSignedW(n.splitval)});
    } else {
      uint32_t m = n.multiplier ? n.multiplier : 1, ml = 0;
      while (!(m & 1)) { m >>= 1; ml++; }
      t.push_back(EncToken{1, 0});
      t.push_back(EncToken{2, (uint32_t)n.pred});
      t.push_back(EncToken{3, PackSignedW(n.offset)});
      t.push_back(EncToken{4, ml});
      t.push_back(EncToken{5, m - 1});
    }
  }
  std::vector<uint32_t> hist(6 * kEncAlphabet, 0);
  for (auto& k : t) {
    uint32_t tok, nb, bits;
    HybridEncode(k.value, &tok, &nb, &bits);
    hist[k.ctx * kEncAlphabet + tok]++;
  }
  EncCode code;
  BuildAndWriteCode(hist.data(), 6, 6, {}, bw, code);
  WriteTokensHost(t, code, bw);
}

// ------------------------------------------------------------------ headers
namespace {
void WriteSize(BitWriter& bw, uint32_t xs, uint32_t ys) {
  static const uint32_t num[8] = {0, 1, 12, 4, 3, 16, 5, 2}, den[8] = {0, 1, 10, 3, 2, 9, 4, 1};
  uint32_t ratio = 0;
  for (uint32_t r = 1; r < 8; r++)
    if ((uint32_t)((uint64_t)ys * num[r] / den[r]) == xs) { ratio = r; break; }
  const bool small = ys <= 256 && ys % 8 == 0 && (ratio != 0 || (xs <= 256 && xs % 8 == 0));
  auto dim = [&](uint32_t v) {
    if (small) bw.Write(5, v / 8 - 1);
    else bw.U32(WB(9, 1), WB(13, 1), WB(18, 1), WB(30, 1), v);
  };
  bw.Bool(small);
  dim(ys);
  bw.Write(3, ratio);
  if (!ratio) dim(xs);
}
}  // namespace

void WriteCodestreamHeaders(const EncImageInfo& im, BitWriter& bw) {
  bw.Write(8, 0xFF);
  bw.Write(8, 0x0A);
  WriteSize(bw, im.xsize, im.ysize);
  // ImageMetadata
  bw.Bool(false);                                   // not all_default
  bw.Bool(false);                                   // no extra fields (orientation 1, no animation, default tone mapping)
  bw.Bool(false);                                   // integer samples ...
  bw.U32(WV(8), WV(10), WV(12), WB(6, 1), 8);       // ... of 8 bits
  bw.Bool(true);                                    // modular_16_bit_buffer_sufficient
  bw.U32(WV(0), WV(1), WB(4, 2), WB(12, 1), im.alpha ? 1 : 0);
  if (im.alpha) bw.Bool(true);                      // the default extra channel: 8-bit unassociated alpha
  bw.Bool(im.xyb);
  // colour encoding: sRGB (or gray with the sRGB transfer curve), D65, perceptual intent (Encoder/JxlEncoder.cpp:269-282)
  const bool want_icc = im.icc && im.icc_size;
  bw.Bool(false);                                   // not all_default
  bw.Bool(want_icc);
  bw.Enum(im.gray ? 1 : 0);
  if (!want_icc) {
    bw.Enum(1);                                     // white point D65
    if (!im.gray) bw.Enum(1);                       // primaries sRGB
    bw.Bool(false);                                 // no gamma
    bw.Enum(13);                                    // transfer function sRGB
    bw.Enum(0);                                     // rendering intent perceptual
  }
  bw.U64(0);                                        // extensions
  bw.Bool(true);                                    // default transform data
  if (want_icc) {
    // the profile: predicted byte stream (icc.cc), 41 contexts on the two previous bytes, ANS
    std::vector<uint8_t> enc;
    IccPredict(im.icc, im.icc_size, &enc);
    bw.U64(enc.size());
    std::vector<EncToken> tokens(enc.size());
    std::vector<uint32_t> hist(kIccContexts * kEncAlphabet, 0);
    for (size_t i = 0; i < enc.size(); i++) {
      tokens[i] = EncToken{IccContext(i, i > 0 ? enc[i - 1] : 0, i > 1 ? enc[i - 2] : 0), enc[i]};
      uint32_t tok, nb, bits;
      HybridEncode(enc[i], &tok, &nb, &bits);
      hist[tokens[i].ctx * kEncAlphabet + tok]++;
    }
    EncCode code;
    BuildAndWriteCode(hist.data(), kIccContexts, 8, {}, bw, code);
    WriteTokensHost(tokens, code, bw);
  }
  bw.AlignByte();
}

void WriteFrameHeader(const EncImageInfo& im, const EncFrameInfo& f, BitWriter& bw) {
  const size_t nec = im.alpha ? 1 : 0;
  bw.Bool(false);   // not all_default
  bw.Write(2, 0);   // regular frame
  bw.Write(1, f.encoding);
  bw.U64(f.flags);
  if (!im.xyb) bw.Bool(false);   // no YCbCr
  bw.U32(WV(1), WV(2), WV(4), WV(8), 1);
  for (size_t i = 0; i < nec; i++) bw.U32(WV(1), WV(2), WV(4), WV(8), 1);
  if (f.encoding == 1) bw.Write(2, f.group_size_shift);
  if (f.encoding == 0 && im.xyb) { bw.Write(3, f.x_qm_scale); bw.Write(3, f.b_qm_scale); }
  bw.U32(WV(1), WV(2), WV(3), WB(3, 4), 1);   // one pass
  bw.Bool(false);                             // no crop
  bw.U32(WV(0), WV(1), WV(2), WB(2, 3), 0);   // blend mode: replace
  for (size_t i = 0; i < nec; i++) bw.U32(WV(0), WV(1), WV(2), WB(2, 3), 0);
  bw.Bool(true);                              // is_last
  bw.U32(WV(0), WB(4), WB(5, 16), WB(10, 48), 0);   // no name
  // loop filter
  const bool lf_default = f.gab && f.epf_iters == 2;
  bw.Bool(lf_default);
  if (!lf_default) {
    bw.Bool(f.gab);
    if (f.gab) bw.Bool(false);   // default Gaborish weights
    bw.Write(2, f.epf_iters);
    if (f.epf_iters) {
      if (f.encoding == 0) bw.Bool(false);   // default sharpness lut
      bw.Bool(false);                        // default channel weights
      bw.Bool(false);                        // default sigma parameters
      if (f.encoding == 1) bw.Write(16, 0x3C00);   // sigma for modular = 1.0 (binary16)
    }
    bw.U64(0);
  }
  bw.U64(0);   // extensions
}

void WriteToc(const std::vector<uint32_t>& sizes, BitWriter& bw) {
  bw.Bool(false);   // not permuted
  bw.AlignByte();
  for (auto s : sizes) bw.U32(WB(10), WB(14, 1024), WB(22, 17408), WB(30, 4211712), s);
  bw.AlignByte();
}

std::vector<uint8_t> WriteContainer(const std::vector<uint8_t>& cs, const uint8_t* exif, size_t exif_size, const uint8_t* xmp,
                                    size_t xmp_size) {
  static const uint8_t kSig[12] = {0, 0, 0, 0xC, 'J', 'X', 'L', ' ', 0xD, 0xA, 0x87, 0xA};
  static const uint8_t kFtyp[20] = {0, 0, 0, 0x14, 'f', 't', 'y', 'p', 'j', 'x', 'l', ' ', 0, 0, 0, 0, 'j', 'x', 'l', ' '};
  std::vector<uint8_t> out(kSig, kSig + 12);
  out.insert(out.end(), kFtyp, kFtyp + 20);
  auto be32 = [&](uint32_t v) { out.push_back(v >> 24); out.push_back(v >> 16); out.push_back(v >> 8); out.push_back(v); };
  auto box = [&](const char* type, const uint8_t* p, size_t n) {
    if (n + 8 > 0xFFFFFFFFull) {
      be32(1);
      out.insert(out.end(), type, type + 4);
      be32((uint32_t)((uint64_t)(n + 16) >> 32));
      be32((uint32_t)(n + 16));
    } else {
      be32((uint32_t)(n + 8));
      out.insert(out.end(), type, type + 4);
    }
    out.insert(out.end(), p, p + n);
  };
  if (exif && exif_size) box("Exif", exif, exif_size);
  if (xmp && xmp_size) box("xml ", xmp, xmp_size);
  box("jxlc", cs.data(), cs.size());
  return out;
}

}  // namespace jxlhip
