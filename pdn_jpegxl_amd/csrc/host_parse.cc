// Host-side frame parser (see host_parse.h).  Restates ISO/IEC 18181-1 header syntax; the
// reference reaches it via libjxl (Decoder/JxlDecoder.cpp:454 JxlDecoderProcessInput).
#include "host_parse.h"
#include "icc.h"
#include "../../include/jxlfiletypeio.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <dlfcn.h>
#include <mutex>

namespace jxlhip {

namespace {
enum { stDecodeError = 10, stInvalidSignature = 12, stUnsupportedChannelFormat = 7 };

[[noreturn]] void Fail(const std::string& m, int st = stDecodeError) { throw ParseError(st, m); }
#define REQUIRE(c, msg) do { if (!(c)) Fail(msg); } while (0)

inline int FloorLog2(uint64_t x) { return 63 - __builtin_clzll(x | 1); }
inline int CeilLog2(uint64_t x) { return x <= 1 ? 0 : FloorLog2(x - 1) + 1; }
inline int64_t Unpack(uint64_t u) { return (int64_t)(u >> 1) ^ -(int64_t)(u & 1); }

// ------------------------------------------------------------------ bit reader
class Bits {
 public:
  Bits(const uint8_t* p, size_t n) : p_(p), n_(n) {}
  uint64_t Peek(int k) const {
    size_t b = pos_ >> 3;
    uint64_t v = 0;
    if (b + 8 <= n_) memcpy(&v, p_ + b, 8);
    else for (size_t i = 0; b + i < n_ && i < 8; i++) v |= (uint64_t)p_[b + i] << (8 * i);
    v >>= pos_ & 7;
    return k >= 57 ? v : v & ((1ull << k) - 1);
  }
  void Skip(size_t k) { pos_ += k; }
  uint32_t u(int k) { if (!k) return 0; uint32_t v = (uint32_t)Peek(k); pos_ += k; return v; }
  bool b() { return u(1) != 0; }
  void Align() { pos_ = (pos_ + 7) & ~(size_t)7; }
  size_t pos() const { return pos_; }
  bool ok() const { return pos_ <= n_ * 8; }
  struct Dist { int bits; uint32_t off; };
  uint32_t U32(Dist a, Dist b_, Dist c, Dist d) {
    Dist t[4] = {a, b_, c, d};
    uint32_t s = u(2);
    return t[s].bits < 0 ? t[s].off : t[s].off + u(t[s].bits);
  }
  uint64_t U64() {
    switch (u(2)) {
      case 0: return 0;
      case 1: return 1 + u(4);
      case 2: return 17 + u(8);
    }
    uint64_t v = u(12);
    for (int shift = 12; u(1);) {
      if (shift == 60) { v |= (uint64_t)u(4) << 60; break; }
      v |= (uint64_t)u(8) << shift;
      shift += 8;
    }
    return v;
  }
  float F16() {
    uint32_t h = u(16), e = (h >> 10) & 31, m = h & 1023;
    REQUIRE(e != 31, "non-finite half float");
    float v = e ? std::ldexp((float)(m | 1024), (int)e - 25) : std::ldexp((float)m, -24);
    return (h >> 15) ? -v : v;
  }
  uint32_t Enum() { return U32({-1, 0}, {-1, 1}, {4, 2}, {6, 18}); }
  void SkipExtensions() {
    uint64_t ext = U64(), total = 0;
    for (int i = 0; i < 64; i++) if (ext >> i & 1) total += U64();
    Skip(total);
  }

 private:
  const uint8_t* p_;
  size_t n_;
  size_t pos_ = 0;
};
typedef Bits::Dist D;
inline D V(uint32_t v) { return {-1, v}; }
inline D B(int n, uint32_t o = 0) { return {n, o}; }

// ------------------------------------------------------------------ entropy code headers
uint32_t VarLen8(Bits& r) { if (!r.u(1)) return 0; uint32_t n = r.u(3); return r.u(n) + (1u << n); }
uint32_t VarLen16(Bits& r) { if (!r.u(1)) return 0; uint32_t n = r.u(4); return r.u(n) + (1u << n); }

HybridCfg ReadCfg(Bits& r, uint32_t log_alpha) {
  HybridCfg c;
  c.split = r.u(CeilLog2(log_alpha + 1));
  REQUIRE(c.split <= log_alpha, "hybrid uint split_exponent");
  c.msb = c.lsb = 0;
  if (c.split != log_alpha) {
    c.msb = r.u(CeilLog2(c.split + 1));
    REQUIRE(c.msb <= c.split, "hybrid uint msb_in_token");
    c.lsb = r.u(CeilLog2(c.split - c.msb + 1));
    REQUIRE(c.msb + c.lsb <= c.split, "hybrid uint lsb_in_token");
  }
  return c;
}

// ANS distribution (12-bit) -> counts
void ReadDistribution(Bits& r, std::vector<int>& d) {
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
  d.clear();
  if (r.u(1)) {
    int n = r.u(1) + 1, s[2] = {0, 0};
    for (int i = 0; i < n; i++) s[i] = VarLen8(r);
    d.assign(std::max(s[0], s[1]) + 1, 0);
    if (n == 1) d[s[0]] = 4096;
    else {
      REQUIRE(s[0] != s[1], "distribution: duplicate symbol");
      d[s[0]] = r.u(12);
      d[s[1]] = 4096 - d[s[0]];
    }
    return;
  }
  if (r.u(1)) {
    int n = VarLen8(r) + 1;
    d.assign(n, 4096 / n);
    for (int i = 0; i < 4096 % n; i++) d[i]++;
    return;
  }
  int log = 0;
  while (log < 3 && r.u(1)) log++;
  int shift = (int)((r.u(log) | (1u << log)) - 1);
  REQUIRE(shift <= 13, "distribution shift");
  int len = VarLen8(r) + 3;
  d.assign(len, 0);
  std::vector<int> lc(len, 0), rle(len, 0);
  int omit = -1, omit_lc = -1;
  for (int i = 0; i < len; i++) {
    uint32_t idx = (uint32_t)r.Peek(7);
    r.Skip(kLen[idx]);
    lc[i] = kSym[idx];
    if (lc[i] == 13) {
      int n = VarLen8(r);
      rle[i] = n + 5;
      i += n + 3;
      continue;
    }
    if (lc[i] > omit_lc) { omit_lc = lc[i]; omit = i; }
  }
  REQUIRE(omit >= 0, "distribution: no omitted symbol");
  REQUIRE(!(omit + 1 < len && lc[omit + 1] == 13), "distribution: RLE after omitted symbol");
  int prev = 0, run = 0, total = 0;
  for (int i = 0; i < len; i++) {
    if (rle[i]) { run = rle[i] - 1; prev = i ? d[i - 1] : 0; }
    if (run > 0) { d[i] = prev; run--; }
    else if (i != omit && lc[i]) {
      if (lc[i] == 1) d[i] = 1;
      else {
        int lg = lc[i] - 1;
        int prec = std::min(lg, shift - ((12 - lg) >> 1));
        if (prec < 0) prec = 0;
        d[i] = (1 << lg) + (r.u(prec) << (lg - prec));
      }
    }
    total += d[i];
  }
  d[omit] = 4096 - total;
  REQUIRE(d[omit] > 0, "distribution does not sum to 4096");
}

// counts -> packed alias entries (table of 1 << log_alpha)
void BuildAlias(std::vector<int> d, uint32_t log_alpha, uint64_t* out) {
  while (!d.empty() && d.back() == 0) d.pop_back();
  if (d.empty()) d.push_back(4096);
  const uint32_t T = 1u << log_alpha, E = 4096 >> log_alpha;
  REQUIRE(d.size() <= T, "alphabet exceeds alias table");
  struct Ent { uint32_t cutoff, right, off1; };
  std::vector<Ent> e(T, Ent{0, 0, 0});
  auto pack = [&](uint32_t i, uint32_t cutoff, uint32_t right, uint32_t off1) {
    uint32_t f0 = i < d.size() ? d[i] : 0, f1 = right < d.size() ? d[right] : 0;
    uint32_t x = cutoff | right << 8 | f0 << 16, y = off1 | (f0 ^ f1) << 16;
    out[i] = (uint64_t)x | (uint64_t)y << 32;
  };
  for (size_t s = 0; s < d.size(); s++)
    if (d[s] == 4096) {
      for (uint32_t i = 0; i < T; i++) {
        // freq0 = 0, freq1 = 4096 for every slot: the state is left unchanged
        uint32_t x = 0 | (uint32_t)s << 8 | 0u << 16, y = (E * i) | 4096u << 16;
        out[i] = (uint64_t)x | (uint64_t)y << 32;
      }
      return;
    }
  std::vector<uint32_t> under, over, cut(T, 0);
  for (uint32_t i = 0; i < d.size(); i++) {
    cut[i] = d[i];
    if (cut[i] > E) over.push_back(i); else if (cut[i] < E) under.push_back(i);
  }
  for (uint32_t i = (uint32_t)d.size(); i < T; i++) under.push_back(i);
  while (!over.empty()) {
    uint32_t o = over.back(); over.pop_back();
    REQUIRE(!under.empty(), "alias construction");
    uint32_t u = under.back(); under.pop_back();
    uint32_t by = E - cut[u];
    cut[o] -= by;
    e[u].right = o;
    e[u].off1 = cut[o];
    if (cut[o] < E) under.push_back(o); else if (cut[o] > E) over.push_back(o);
  }
  for (uint32_t i = 0; i < T; i++) {
    if (cut[i] == E) pack(i, 0, i, 0);
    else pack(i, cut[i], e[i].right, (e[i].off1 - cut[i]) & 0xFFFF);
  }
}

void BuildPrefix(const std::vector<uint8_t>& len, HostCode::Prefix& p) {
  memset(p.count, 0, sizeof(p.count));
  p.sorted.clear();
  p.single = -1;
  int nz = 0, last = 0;
  for (size_t i = 0; i < len.size(); i++) if (len[i]) { p.count[len[i]]++; nz++; last = (int)i; }
  if (nz <= 1) { p.single = nz ? last : 0; return; }
  for (int l = 1; l < 16; l++) for (size_t i = 0; i < len.size(); i++) if (len[i] == l) p.sorted.push_back((uint16_t)i);
}

uint32_t PrefixSymbol(Bits& r, const HostCode::Prefix& p) {
  if (p.single >= 0) return p.single;
  int code = 0, first = 0, index = 0;
  for (int l = 1; l < 16; l++) {
    code |= r.u(1);
    int c = p.count[l];
    if (code - c < first) return p.sorted[index + code - first];
    index += c; first = (first + c) << 1; code <<= 1;
  }
  Fail("invalid prefix code");
}

void ReadPrefix(Bits& r, uint32_t asz, HostCode::Prefix& p) {
  std::vector<uint8_t> len(asz, 0);
  if (asz == 1) { BuildPrefix(len, p); p.single = 0; return; }
  uint32_t hskip = r.u(2);
  if (hskip == 1) {
    int nbits = 0;
    for (uint32_t c = asz - 1; c; c >>= 1) nbits++;
    uint32_t n = r.u(2) + 1, s[4];
    for (uint32_t i = 0; i < n; i++) { s[i] = r.u(nbits); REQUIRE(s[i] < asz, "prefix symbol range"); }
    for (uint32_t i = 0; i < n; i++) for (uint32_t j = i + 1; j < n; j++) REQUIRE(s[i] != s[j], "prefix duplicate symbol");
    if (n == 1) { BuildPrefix(len, p); p.single = s[0]; return; }
    if (n == 2) len[s[0]] = len[s[1]] = 1;
    else if (n == 3) { len[s[0]] = 1; len[s[1]] = len[s[2]] = 2; }
    else if (r.u(1)) { len[s[0]] = 1; len[s[1]] = 2; len[s[2]] = len[s[3]] = 3; }
    else len[s[0]] = len[s[1]] = len[s[2]] = len[s[3]] = 2;
    BuildPrefix(len, p);
    return;
  }
  static const uint8_t kOrd[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
  static const uint8_t kL[16] = {2, 2, 2, 3, 2, 2, 2, 4, 2, 2, 2, 3, 2, 2, 2, 4};
  static const uint8_t kV[16] = {0, 4, 3, 2, 0, 4, 3, 1, 0, 4, 3, 2, 0, 4, 3, 5};
  std::vector<uint8_t> cl(18, 0);
  int space = 32, ncodes = 0;
  for (int i = hskip; i < 18 && space > 0; i++) {
    uint32_t pk = (uint32_t)r.Peek(4);
    r.Skip(kL[pk]);
    cl[kOrd[i]] = kV[pk];
    if (kV[pk]) { space -= 32 >> kV[pk]; ncodes++; }
  }
  REQUIRE(ncodes == 1 || space == 0, "code-length code");
  HostCode::Prefix clp;
  BuildPrefix(cl, clp);
  uint32_t sym = 0;
  int prev = 8, rep = 0, rep_len = 0, sp = 32768;
  while (sym < asz && sp > 0) {
    uint32_t v = PrefixSymbol(r, clp);
    if (v < 16) {
      rep = 0;
      len[sym++] = (uint8_t)v;
      if (v) { prev = v; sp -= 32768 >> v; }
    } else {
      int extra = v == 16 ? 2 : 3, nl = v == 16 ? prev : 0;
      if (rep_len != nl) { rep = 0; rep_len = nl; }
      int old = rep;
      if (rep > 0) rep = (rep - 2) << extra;
      rep += r.u(extra) + 3;
      int delta = rep - old;
      REQUIRE(sym + delta <= asz, "prefix repeat overflow");
      for (int i = 0; i < delta; i++) len[sym++] = (uint8_t)rep_len;
      if (rep_len) sp -= delta << (15 - rep_len);
    }
  }
  REQUIRE(sp == 0, "prefix code incomplete");
  BuildPrefix(len, p);
}

void ReadCode(Bits& r, size_t num_ctx, HostCode& c, bool no_lz77 = false);

// Symbol reader for the small host-decoded streams (context maps, permutations, MA trees).
class SymReader {
 public:
  SymReader(const HostCode& c, Bits& r, uint32_t dist_mult = 0) : c_(c), r_(r), dm_(dist_mult) {
    if (c.lz77) win_.assign(1u << 20, 0);
    state_ = c.use_prefix ? 0x130000u : r.u(32);
  }
  uint32_t Get(uint32_t ctx) {
    if (c_.lz77 && copy_ > 0) return Copy();
    uint32_t h = c_.ctx_map[ctx], tok = Sym(h);
    if (c_.lz77 && tok >= c_.lz_min_symbol) {
      copy_ = Hybrid(c_.lz_len, tok - c_.lz_min_symbol) + c_.lz_min_length;
      uint32_t dh = c_.ctx_map.back(), dist = Hybrid(c_.cfg[dh], Sym(dh));
      static const int8_t kSD[120][2] = {
          {0, 1}, {1, 0}, {1, 1}, {-1, 1}, {0, 2}, {2, 0}, {1, 2}, {-1, 2}, {2, 1}, {-2, 1}, {2, 2}, {-2, 2}, {0, 3}, {3, 0}, {1, 3},
          {-1, 3}, {3, 1}, {-3, 1}, {2, 3}, {-2, 3}, {3, 2}, {-3, 2}, {0, 4}, {4, 0}, {1, 4}, {-1, 4}, {4, 1}, {-4, 1}, {3, 3}, {-3, 3},
          {2, 4}, {-2, 4}, {4, 2}, {-4, 2}, {0, 5}, {3, 4}, {-3, 4}, {4, 3}, {-4, 3}, {5, 0}, {1, 5}, {-1, 5}, {5, 1}, {-5, 1}, {2, 5},
          {-2, 5}, {5, 2}, {-5, 2}, {4, 4}, {-4, 4}, {3, 5}, {-3, 5}, {5, 3}, {-5, 3}, {0, 6}, {6, 0}, {1, 6}, {-1, 6}, {6, 1}, {-6, 1},
          {2, 6}, {-2, 6}, {6, 2}, {-6, 2}, {4, 5}, {-4, 5}, {5, 4}, {-5, 4}, {3, 6}, {-3, 6}, {6, 3}, {-6, 3}, {0, 7}, {7, 0}, {1, 7},
          {-1, 7}, {5, 5}, {-5, 5}, {7, 1}, {-7, 1}, {4, 6}, {-4, 6}, {6, 4}, {-6, 4}, {2, 7}, {-2, 7}, {7, 2}, {-7, 2}, {3, 7}, {-3, 7},
          {7, 3}, {-7, 3}, {5, 6}, {-5, 6}, {6, 5}, {-6, 5}, {8, 0}, {4, 7}, {-4, 7}, {7, 4}, {-7, 4}, {8, 1}, {8, 2}, {6, 6}, {-6, 6},
          {8, 3}, {5, 7}, {-5, 7}, {7, 5}, {-7, 5}, {8, 4}, {6, 7}, {-6, 7}, {7, 6}, {-7, 6}, {8, 5}, {7, 7}, {-7, 7}, {8, 6}, {8, 7}};
      if (dm_ == 0) dist++;
      else if (dist < 120) { int o = kSD[dist][0] + (int)dm_ * kSD[dist][1]; dist = o < 1 ? 1 : o; }
      else dist -= 119;
      dist = std::min(dist, std::min(done_, 1u << 20));
      src_ = done_ - dist;
      return Copy();
    }
    uint32_t v = Hybrid(c_.cfg[h], tok);
    if (c_.lz77) win_[(done_++) & 0xFFFFF] = v;
    return v;
  }
  bool Final() const { return c_.use_prefix || state_ == 0x130000u; }

 private:
  uint32_t Copy() {
    uint32_t v = win_[(src_++) & 0xFFFFF];
    copy_--;
    win_[(done_++) & 0xFFFFF] = v;
    return v;
  }
  uint32_t Sym(uint32_t h) {
    if (c_.use_prefix) return PrefixSymbol(r_, c_.prefix[h]);
    uint32_t le = 12 - c_.log_alpha, res = state_ & 0xFFF, i = res >> le, pos = res & ((1u << le) - 1);
    uint64_t e = c_.alias[((size_t)h << c_.log_alpha) | i];
    uint32_t x = (uint32_t)e, y = (uint32_t)(e >> 32);
    uint32_t cutoff = x & 0xFF, right = (x >> 8) & 0xFF, f0 = x >> 16, off1 = y & 0xFFFF, fx = y >> 16;
    bool g = pos >= cutoff;
    uint32_t sym = g ? right : i, off = g ? off1 + pos : pos, freq = g ? (f0 ^ fx) : f0;
    state_ = freq * (state_ >> 12) + off;
    if (state_ < 65536) state_ = state_ << 16 | r_.u(16);
    return sym;
  }
  uint32_t Hybrid(const HybridCfg& c, uint32_t tok) {
    uint32_t split = 1u << c.split;
    if (tok < split) return tok;
    uint32_t nb = c.split - (c.msb + c.lsb) + ((tok - split) >> (c.msb + c.lsb));
    REQUIRE(nb <= 31, "hybrid uint width");
    uint32_t low = tok & ((1u << c.lsb) - 1);
    tok >>= c.lsb;
    uint32_t bits = r_.u(nb), hi = (1u << c.msb) | (tok & ((1u << c.msb) - 1));
    return (uint32_t)(((((uint64_t)hi << nb) | bits) << c.lsb) | low);
  }
  const HostCode& c_;
  Bits& r_;
  uint32_t dm_, state_ = 0, copy_ = 0, src_ = 0, done_ = 0;
  std::vector<uint32_t> win_;
};

void ReadContextMap(Bits& r, std::vector<uint8_t>& map, uint32_t* num_hist) {
  if (r.b()) {
    int nb = r.u(2);
    for (auto& m : map) m = (uint8_t)r.u(nb);
  } else {
    bool mtf = r.b();
    HostCode c;
    ReadCode(r, 1, c, map.size() <= 2);
    SymReader sr(c, r);
    for (auto& m : map) {
      uint32_t v = sr.Get(0);
      REQUIRE(v < 256, "context map value");
      m = (uint8_t)v;
    }
    REQUIRE(sr.Final(), "context map: ANS final state");
    if (mtf) {
      uint8_t t[256];
      for (int i = 0; i < 256; i++) t[i] = (uint8_t)i;
      for (auto& m : map) {
        uint8_t i = m, v = t[i];
        m = v;
        memmove(t + 1, t, i);
        t[0] = v;
      }
    }
  }
  uint32_t mx = 0;
  for (auto m : map) mx = std::max<uint32_t>(mx, m);
  *num_hist = mx + 1;
}

void ReadCode(Bits& r, size_t num_ctx, HostCode& c, bool no_lz77) {
  c = HostCode();
  c.lz77 = r.b();
  if (c.lz77) {
    REQUIRE(!no_lz77, "lz77 not allowed in this stream");
    c.lz_min_symbol = r.U32(V(224), V(512), V(4096), B(15, 8));
    c.lz_min_length = r.U32(V(3), V(4), B(2, 5), B(8, 9));
    c.lz_len = ReadCfg(r, 8);
    num_ctx++;
  }
  c.ctx_map.assign(num_ctx, 0);
  if (num_ctx > 1) ReadContextMap(r, c.ctx_map, &c.num_hist);
  c.use_prefix = r.b();
  c.log_alpha = c.use_prefix ? 15 : 5 + r.u(2);
  c.cfg.resize(c.num_hist);
  for (auto& g : c.cfg) g = ReadCfg(r, c.log_alpha);
  if (c.use_prefix) {
    std::vector<uint32_t> asz(c.num_hist);
    for (auto& a : asz) { a = VarLen16(r) + 1; REQUIRE(a <= 32768, "prefix alphabet size"); }
    c.prefix.resize(c.num_hist);
    for (uint32_t i = 0; i < c.num_hist; i++) ReadPrefix(r, asz[i], c.prefix[i]);
  } else {
    c.alias.assign((size_t)c.num_hist << c.log_alpha, 0);
    std::vector<int> d;
    for (uint32_t i = 0; i < c.num_hist; i++) {
      ReadDistribution(r, d);
      REQUIRE(d.size() <= (1u << c.log_alpha), "distribution alphabet size");
      BuildAlias(d, c.log_alpha, &c.alias[(size_t)i << c.log_alpha]);
    }
  }
  REQUIRE(r.ok(), "truncated entropy code header");
}

uint32_t PermCtx(uint32_t v) { return v ? std::min(FloorLog2(v) + 1, 7) : 0; }

void ReadPermutation(SymReader& sr, size_t skip, size_t n, std::vector<uint32_t>& perm) {
  std::vector<uint32_t> lehmer(n, 0);
  uint32_t end = sr.Get(PermCtx((uint32_t)n));
  REQUIRE(end <= n - skip, "permutation length");
  uint32_t last = 0;
  for (size_t i = skip; i < skip + end; i++) {
    last = lehmer[i] = sr.Get(PermCtx(last));
    REQUIRE(lehmer[i] < n - i, "lehmer digit");
  }
  std::vector<uint32_t> pool(n);
  for (size_t i = 0; i < n; i++) pool[i] = (uint32_t)i;
  perm.resize(n);
  for (size_t i = 0; i < n; i++) {
    perm[i] = pool[lehmer[i]];
    pool.erase(pool.begin() + lehmer[i]);
  }
}

// ------------------------------------------------------------------ headers
void ReadSize(Bits& r, uint32_t* xs, uint32_t* ys) {
  static const uint32_t num[8] = {0, 1, 12, 4, 3, 16, 5, 2}, den[8] = {0, 1, 10, 3, 2, 9, 4, 1};
  bool small = r.b();
  auto dim = [&]() { return small ? (r.u(5) + 1) * 8 : r.U32(B(9, 1), B(13, 1), B(18, 1), B(30, 1)); };
  *ys = dim();
  uint32_t ratio = r.u(3);
  *xs = ratio ? (uint32_t)((uint64_t)*ys * num[ratio] / den[ratio]) : dim();
}

void ReadBitDepth(Bits& r, uint32_t* bits, uint32_t* exp) {
  if (!r.b()) { *bits = r.U32(V(8), V(10), V(12), B(6, 1)); *exp = 0; }
  else { *bits = r.U32(V(32), V(16), V(24), B(6, 1)); *exp = r.u(4) + 1; }
}

void ReadImageHeader(Bits& r, ParsedFrame& f) {
  ReadSize(r, &f.xsize, &f.ysize);
  bool extra = false;
  if (!r.b()) {
    extra = r.b();
    if (extra) {
      f.orientation = r.u(3) + 1;
      if (r.b()) { uint32_t a, b2; ReadSize(r, &a, &b2); }
      if (r.b()) Fail("preview frames are not supported");
      f.have_animation = r.b();
      if (f.have_animation) {
        r.U32(V(100), V(1000), B(10, 1), B(30, 1));
        r.U32(V(1), V(1001), B(8, 1), B(10, 1));
        r.U32(V(0), B(3), B(16), B(32));
        f.have_timecodes = r.b();
      }
    }
    ReadBitDepth(r, &f.bits, &f.exp_bits);
    r.b();  // modular_16_bit_buffer_sufficient
    uint32_t nec = r.U32(V(0), V(1), B(4, 2), B(12, 1));
    f.ec.resize(nec);
    for (auto& e : f.ec) {
      if (r.b()) continue;  // default: 8-bit alpha
      e.type = r.Enum();
      ReadBitDepth(r, &e.bits, &e.exp_bits);
      e.dim_shift = r.U32(V(0), V(3), V(4), B(3, 1));
      uint32_t nl = r.U32(V(0), B(4), B(5, 16), B(10, 48));
      r.Skip(8 * (size_t)nl);
      if (e.type == 0) e.alpha_associated = r.b();
      if (e.type == 2) r.Skip(64);
      if (e.type == 5) r.U32(V(1), B(2), B(4, 3), B(8, 19));
    }
    f.xyb_encoded = r.b();
    ColorInfo& c = f.color;
    c.all_default = r.b();
    if (!c.all_default) {
      c.want_icc = r.b();
      c.color_space = r.Enum();
      if (!c.want_icc) {
        auto xy = [&]() {
          const uint32_t u = r.U32(B(19), B(19, 524288), B(20, 1048576), B(21, 2097152));
          return (double)((int32_t)(u >> 1) ^ -(int32_t)(u & 1)) * 1e-6;
        };
        if (c.color_space != 2) { c.white_point = r.Enum(); if (c.white_point == 2) { c.white_xy[0] = xy(); c.white_xy[1] = xy(); } }
        if (c.color_space != 2 && c.color_space != 1) {
          c.primaries = r.Enum();
          if (c.primaries == 2) for (int i = 0; i < 3; i++) { c.prim_xy[i][0] = xy(); c.prim_xy[i][1] = xy(); }
        }
        if (c.color_space != 2) { c.have_gamma = r.b(); if (c.have_gamma) c.gamma = r.u(24); else c.tf = r.Enum(); }
        c.rendering_intent = r.Enum();
      }
    }
    if (extra && !r.b()) {
      f.intensity_target = r.F16();
      r.F16(); r.b(); r.F16();
    }
    r.SkipExtensions();
  }
  static const float kInv[9] = {11.031566901960783f, -9.866943921568629f, -0.16462299647058826f, -3.254147380392157f, 4.418770392156863f,
                                -0.16462299647058826f, -3.6588512862745097f, 2.7129230470588235f, 1.9459282392156863f};
  memcpy(f.opsin_inv, kInv, sizeof(kInv));
  for (int i = 0; i < 3; i++) f.opsin_bias[i] = -0.0037930732552754493f;
  f.qbias[0] = 1.0f - 0.05465007330715401f; f.qbias[1] = 1.0f - 0.07005449891748593f;
  f.qbias[2] = 1.0f - 0.049935103337343655f; f.qbias[3] = 0.145f;
  if (!r.b()) {  // custom transform data
    if (f.xyb_encoded && !r.b()) {
      for (auto& v : f.opsin_inv) v = r.F16();
      for (auto& v : f.opsin_bias) v = r.F16();
      for (auto& v : f.qbias) v = r.F16();
    }
    uint32_t mask = r.u(3);
    if (mask & 1) r.Skip(16 * 15);
    if (mask & 2) r.Skip(16 * 55);
    if (mask & 4) r.Skip(16 * 210);
  }
  f.ncolor = f.color.color_space == 1 ? 1 : 3;
  for (size_t i = 0; i < f.ec.size(); i++) {
    if (f.ec[i].type == 0 && f.alpha_index < 0) f.alpha_index = (int)i;
    if (f.ec[i].type == 4 && f.black_index < 0) f.black_index = (int)i;
  }
  REQUIRE(r.ok(), "truncated image header");
}

void ReadBlending(Bits& r, size_t nec, bool partial, uint32_t* mode) {
  *mode = r.U32(V(0), V(1), V(2), B(2, 3));
  if (nec && (*mode == 2 || *mode == 3)) r.U32(V(0), V(1), V(2), B(3, 3));
  if (nec && *mode >= 2 && *mode <= 4) r.b();
  if (*mode != 0 || partial) r.u(2);
}

void ReadFrameHeader(Bits& r, ParsedFrame& f) {
  const float w1 = 0.115169525f, w2 = 0.061248592f;
  for (int c = 0; c < 3; c++) { f.gab_w1[c] = w1; f.gab_w2[c] = w2; }
  for (int i = 0; i < 8; i++) f.epf_sharp_lut[i] = i / 7.0f;
  f.epf_channel_scale[0] = 40.f; f.epf_channel_scale[1] = 5.f; f.epf_channel_scale[2] = 3.5f;
  f.epf_quant_mul = 0.46f; f.epf_pass0_sigma_scale = 0.9f; f.epf_pass2_sigma_scale = 6.5f; f.epf_border_sad_mul = 2.0f / 3;
  bool have_crop = false, do_ycbcr = false;
  uint32_t upsampling = 1, blend = 0, duration = 0, save_ref = 0;
  int32_t x0 = 0, y0 = 0;
  uint32_t cw = 0, ch = 0;
  if (!r.b()) {
    f.frame_type = r.u(2);
    f.encoding = r.u(1);
    f.flags = r.U64();
    if (!f.xyb_encoded) do_ycbcr = r.b();
    bool lf_frame = f.flags & 32;
    if (do_ycbcr && !lf_frame) r.u(6);
    if (!lf_frame) {
      upsampling = r.U32(V(1), V(2), V(4), V(8));
      for (size_t i = 0; i < f.ec.size(); i++) REQUIRE(r.U32(V(1), V(2), V(4), V(8)) == 1, "extra-channel upsampling is not supported");
    }
    if (f.encoding == 1) f.group_size_shift = r.u(2);
    if (f.encoding == 0 && f.xyb_encoded) { f.x_qm_scale = r.u(3); f.b_qm_scale = r.u(3); }
    if (f.frame_type != 2) {
      f.num_passes = r.U32(V(1), V(2), V(3), B(3, 4));
      if (f.num_passes != 1) {
        uint32_t nds = r.U32(V(0), V(1), V(2), B(1, 3));
        REQUIRE(f.num_passes <= 8, "too many passes");
        for (uint32_t i = 0; i + 1 < f.num_passes; i++) f.pass_shift[i] = r.u(2);
        for (uint32_t i = 0; i < nds; i++) r.U32(V(1), V(2), V(4), V(8));
        for (uint32_t i = 0; i < nds; i++) r.U32(V(0), V(1), V(2), B(3));
        // downsampling brackets move squeezed Modular channels into earlier passes; without them everything Modular is in the last pass
        REQUIRE(nds == 0, "progressive frames with downsampling brackets are not supported yet");
      }
    }
    if (f.frame_type == 1) r.U32(V(1), V(2), V(3), V(4));
    else {
      have_crop = r.b();
      if (have_crop) {
        auto dim = [&]() { return r.U32(B(8), B(11, 256), B(14, 2304), B(30, 18688)); };
        if (f.frame_type != 2) { x0 = (int32_t)Unpack(dim()); y0 = (int32_t)Unpack(dim()); }
        cw = dim(); ch = dim();
      }
    }
    bool normal = f.frame_type == 0 || f.frame_type == 3;
    bool full = !have_crop || (x0 <= 0 && y0 <= 0 && x0 + (int64_t)cw >= f.xsize && y0 + (int64_t)ch >= f.ysize);
    if (normal) {
      ReadBlending(r, f.ec.size(), !full, &blend);
      for (size_t i = 0; i < f.ec.size(); i++) { uint32_t m; ReadBlending(r, f.ec.size(), !full, &m); }
      if (f.have_animation) { duration = r.U32(V(0), V(1), B(8), B(32)); if (f.have_timecodes) r.u(32); }
      f.is_last = r.b();
    } else f.is_last = false;
    if (f.frame_type != 1 && !f.is_last) save_ref = r.u(2);
    if (f.frame_type != 1) {
      bool can_ref = !f.is_last && (duration == 0 || save_ref != 0);
      if (f.frame_type == 2 || (full && blend == 0 && can_ref)) r.b();
    }
    uint32_t nl = r.U32(V(0), B(4), B(5, 16), B(10, 48));
    f.name.resize(nl);
    for (auto& c : f.name) c = (char)r.u(8);
    if (!r.b()) {  // loop filter
      f.gab = r.b();
      if (f.gab && r.b()) for (int c = 0; c < 3; c++) { f.gab_w1[c] = r.F16(); f.gab_w2[c] = r.F16(); }
      f.epf_iters = r.u(2);
      if (f.epf_iters) {
        if (f.encoding == 0 && r.b()) for (auto& v : f.epf_sharp_lut) v = r.F16();
        if (r.b()) { for (auto& v : f.epf_channel_scale) v = r.F16(); r.u(32); }
        if (r.b()) {
          if (f.encoding == 0) f.epf_quant_mul = r.F16();
          f.epf_pass0_sigma_scale = r.F16(); f.epf_pass2_sigma_scale = r.F16(); f.epf_border_sad_mul = r.F16();
        }
        if (f.encoding == 1) r.F16();
      }
      REQUIRE(r.U64() == 0, "loop filter extensions");
    }
    r.SkipExtensions();
  }
  REQUIRE(f.frame_type == 0 || f.frame_type == 3, "first frame is not a regular frame (LF / reference frames are not supported yet)");
  REQUIRE(!have_crop && upsampling == 1, "cropped or upsampled frames are not supported yet");
  REQUIRE(!do_ycbcr, "YCbCr frames are not supported yet");
  // The reference keeps the first frame its decoder library hands out (Decoder/JxlDecoder.cpp:398-400), and that library composes
  // layers: the first DISPLAYED frame is this frame alone only if it ends the file or is an animation frame with a duration.
  REQUIRE(f.is_last || (f.have_animation && duration > 0), "images composed of several layers are not supported yet");
  REQUIRE(blend == 0, "frame blend modes other than replace are not supported yet");
  f.group_dim = 128u << f.group_size_shift;
  f.w8 = (f.xsize + 7) / 8; f.h8 = (f.ysize + 7) / 8;
  f.xg = (f.xsize + f.group_dim - 1) / f.group_dim; f.yg = (f.ysize + f.group_dim - 1) / f.group_dim;
  f.ng = f.xg * f.yg;
  f.xlf = (f.xsize + f.group_dim * 8 - 1) / (f.group_dim * 8); f.ylf = (f.ysize + f.group_dim * 8 - 1) / (f.group_dim * 8);
  f.nlf = f.xlf * f.ylf;
  REQUIRE(r.ok(), "truncated frame header");
}

void ReadToc(Bits& r, ParsedFrame& f, size_t frame_base_bits) {
  size_t n = (f.ng == 1 && f.num_passes == 1) ? 1 : 2 + f.nlf + (size_t)f.ng * f.num_passes;
  std::vector<uint32_t> perm;
  bool permuted = r.b();
  if (permuted) {
    HostCode c;
    ReadCode(r, 8, c);
    SymReader sr(c, r);
    ReadPermutation(sr, 0, n, perm);
    REQUIRE(sr.Final(), "TOC permutation: ANS final state");
  }
  r.Align();
  std::vector<uint32_t> sizes(n);
  for (auto& s : sizes) s = r.U32(B(10), B(14, 1024), B(22, 17408), B(30, 4211712));
  r.Align();
  REQUIRE(r.ok(), "truncated TOC");
  (void)frame_base_bits;
  uint64_t base = r.pos() / 8;
  std::vector<uint64_t> phys(n + 1, base);
  for (size_t i = 0; i < n; i++) phys[i + 1] = phys[i] + sizes[i];
  REQUIRE(phys[n] <= f.cs_size, "sections exceed the codestream");
  f.sec_off.resize(n);
  f.sec_size.resize(n);
  for (size_t i = 0; i < n; i++) {
    size_t p = permuted ? perm[i] : i;
    f.sec_off[i] = phys[p];
    f.sec_size[i] = sizes[p];
  }
}

// ------------------------------------------------------------------ LfGlobal / HfGlobal
void ReadTree(Bits& r, ParsedFrame& f, size_t limit) {
  HostCode c;
  ReadCode(r, 6, c);
  SymReader sr(c, r);
  f.tree.clear();
  f.tree_row_static = true;
  size_t pending = 1, leaves = 0;
  while (pending--) {
    REQUIRE(f.tree.size() < limit, "MA tree too large");
    DevTreeNode n;
    int prop = (int)sr.Get(1) - 1;
    REQUIRE(prop < 256, "MA tree property");
    n.property = prop;
    if (prop < 0) {
      uint32_t pred = sr.Get(2);
      REQUIRE(pred < 14, "MA tree predictor");
      int64_t off = Unpack(sr.Get(3));
      uint32_t ml = sr.Get(4);
      REQUIRE(ml < 31, "MA tree multiplier");
      uint32_t mb = sr.Get(5);
      REQUIRE(mb + 1 < (1u << (31 - ml)), "MA tree multiplier bits");
      REQUIRE(leaves < (1u << 20), "too many MA tree leaves");
      n.splitval = (int32_t)off;
      n.a = pred | (uint32_t)leaves << 8;
      n.b = (mb + 1) << ml;
      leaves++;
      if (pred == 6) f.tree_uses_wp = true;
      if (!(pred == 0 || pred == 1 || pred == 2 || pred == 5)) f.tree_row_static = false;
    } else {
      n.splitval = (int32_t)Unpack(sr.Get(0));
      n.a = (uint32_t)(f.tree.size() + pending + 1);
      n.b = (uint32_t)(f.tree.size() + pending + 2);
      pending += 2;
      if (prop == 15) f.tree_uses_wp = true;
      if (prop > 2) f.tree_row_static = false;
      if (prop > 15) f.tree_uses_ref = true;
    }
    f.tree.push_back(n);
    REQUIRE(r.ok(), "truncated MA tree");
  }
  REQUIRE(sr.Final(), "MA tree: ANS final state");
}

void ReadLfGlobal(Bits& r, ParsedFrame& f) {
  REQUIRE(!(f.flags & (1 | 2 | 16)), "noise / patches / splines are not supported yet");
  REQUIRE(!(f.flags & 32), "LF frames are not supported yet");
  f.m_lf[0] = 1.0f / 4096; f.m_lf[1] = 1.0f / 512; f.m_lf[2] = 1.0f / 256;
  if (!r.b()) for (auto& v : f.m_lf) { v = r.F16() / 128; REQUIRE(v >= 1e-8f, "LF dequantisation factor"); }
  static const uint8_t kDefCtx[39] = {0, 1, 2, 2, 3, 3, 4, 5, 6, 6, 6, 6, 6, 7, 8, 9, 9, 10, 11, 12,
                                      13, 14, 14, 14, 14, 14, 7, 8, 9, 9, 10, 11, 12, 13, 14, 14, 14, 14, 14};
  f.block_ctx_map.assign(kDefCtx, kDefCtx + 39);
  f.num_block_ctx = 15;
  if (f.encoding == 0) {
    f.global_scale = r.U32(B(11, 1), B(11, 2049), B(12, 4097), B(16, 8193));
    f.quant_lf = r.U32(V(16), B(5, 1), B(8, 1), B(16, 1));
    if (!r.b()) {
      size_t nlf_ctx = 1;
      for (int j = 0; j < 3; j++) {
        uint32_t n = r.u(4);
        f.lf_thr[j].resize(n);
        for (auto& t : f.lf_thr[j]) { const uint32_t u = r.U32(B(4), B(8, 16), B(16, 272), B(32, 65808)); t = (int32_t)(u >> 1) ^ -(int32_t)(u & 1); }
        nlf_ctx *= n + 1;
      }
      uint32_t nqf = r.u(4);
      f.qf_thr.resize(nqf);
      for (auto& t : f.qf_thr) t = r.U32(B(2), B(3, 4), B(5, 12), B(8, 44)) + 1;
      REQUIRE(nlf_ctx * (nqf + 1) <= 64, "block-context map too large");
      f.block_ctx_map.assign(39 * (nqf + 1) * nlf_ctx, 0);
      ReadContextMap(r, f.block_ctx_map, &f.num_block_ctx);
      REQUIRE(f.num_block_ctx <= 16, "too many block contexts");
    }
    if (!r.b()) {
      f.color_factor = r.U32(V(84), V(256), B(8, 2), B(16, 258));
      f.base_x = r.F16(); f.base_b = r.F16();
      f.ytox_lf = (int)r.u(8) - 128; f.ytob_lf = (int)r.u(8) - 128;
    }
  }
  f.has_global_tree = r.b();
  if (f.has_global_tree) {
    size_t limit = std::min<size_t>((size_t)1 << 22, 1024 + (size_t)f.xsize * f.ysize * (f.ncolor + f.ec.size()) / 16);
    ReadTree(r, f, limit);
    ReadCode(r, (f.tree.size() + 1) / 2, f.mcode);
  }
  // GlobalModular image header.  VarDCT frames: only {global tree, default predictor parameters, no transforms} (the alpha
  // channel); Modular frames additionally allow reversible colour transforms.
  size_t nchan = (f.encoding == 1 ? f.ncolor : 0) + f.ec.size();
  f.global_modular_has_channels = nchan > 0;
  if (nchan > 0) {
    bool use_global = r.b(), wp_default = r.b();
    REQUIRE(use_global && f.has_global_tree, "modular streams with local MA trees are not supported on the GPU path yet");
    REQUIRE(wp_default, "custom weighted-predictor parameters are not supported on the GPU path yet");
    uint32_t ntr = r.U32(V(0), V(1), B(4, 2), B(8, 18));
    REQUIRE(f.encoding == 1 || ntr == 0, "modular transforms on extra channels are not supported on the GPU path yet");
    for (uint32_t i = 0; i < ntr; i++) {
      ParsedFrame::ModTransform t;
      t.id = r.u(2);
      REQUIRE(t.id != 3, "invalid modular transform");
      if (t.id == 1) {
        t.begin_c = r.U32(B(3), B(6, 8), B(10, 72), B(13, 1096));
        t.num_c = r.U32(V(1), V(3), V(4), B(13, 1));
        t.nb_colors = r.U32(B(8), B(10, 256), B(12, 1280), B(16, 5376));
        t.nb_deltas = r.U32(V(0), B(8, 1), B(10, 257), B(16, 1281));
        t.predictor = r.u(4);
        REQUIRE(t.nb_deltas == 0, "delta palettes are not supported yet");
        REQUIRE(t.num_c >= 1 && t.num_c <= 4 && t.nb_colors >= 1, "palettes of more than four channels are not supported yet");
      } else if (t.id == 0) {
        t.begin_c = r.U32(B(3), B(6, 8), B(10, 72), B(13, 1096));
        t.rct_type = r.U32(V(6), B(2), B(4, 2), B(6, 10));
        REQUIRE(t.rct_type < 42, "reversible colour transform type");
      } else {
        uint32_t nsq = r.U32(V(0), B(4, 1), B(6, 9), B(8, 41));
        t.squeezes.resize(nsq);
        for (auto& q : t.squeezes) {
          q.horizontal = r.b();
          q.in_place = r.b();
          q.begin_c = r.U32(B(3), B(6, 8), B(10, 72), B(13, 1096));
          q.num_c = r.U32(V(1), V(2), V(3), B(4, 4));
        }
      }
      f.mod_transforms.push_back(t);
    }
  }
  if (f.encoding == 1) {
    // simulate the transforms on the channel list (sizes only): what is coded, and how to get back
    typedef ParsedFrame::ModChan Chan;
    std::vector<Chan> cur;
    for (size_t c = 0; c < nchan; c++) {
      Chan ch;
      ch.w = (int32_t)f.xsize; ch.h = (int32_t)f.ysize; ch.plane = (int32_t)c;
      cur.push_back(ch);
      f.mod_planes.push_back({ch.w, ch.h});
    }
    std::vector<ParsedFrame::ModOp> fwd;
    size_t nb_meta = 0;
    for (auto& t : f.mod_transforms) {
      if (t.id == 1) {
        // Palette: num_c channels become one channel of indices, their colours travel in a meta channel (nb_colors x num_c) put
        // first in the channel list; meta channels always live in the GlobalModular stream
        REQUIRE(t.begin_c >= nb_meta && (size_t)t.begin_c + t.num_c <= cur.size(), "palette channel range");
        const Chan first = cur[t.begin_c];
        ParsedFrame::ModOp op;
        op.kind = 3; op.type = (int32_t)t.nb_colors; op.nout = (int32_t)t.num_c;
        for (uint32_t k = 0; k < t.num_c; k++) {
          const Chan& ck = cur[t.begin_c + k];
          REQUIRE(ck.w == first.w && ck.h == first.h && ck.hshift == first.hshift && ck.vshift == first.vshift, "palette on channels of different size");
          op.out[k] = ck.plane;
        }
        Chan idx = first, pal;
        idx.plane = (int32_t)f.mod_planes.size(); f.mod_planes.push_back({idx.w, idx.h});
        pal.w = (int32_t)t.nb_colors; pal.h = (int32_t)t.num_c; pal.hshift = -1; pal.vshift = -1;
        pal.plane = (int32_t)f.mod_planes.size(); f.mod_planes.push_back({pal.w, pal.h});
        op.a = pal.plane; op.b = idx.plane;
        fwd.push_back(op);
        cur.erase(cur.begin() + t.begin_c, cur.begin() + t.begin_c + t.num_c);
        cur.insert(cur.begin() + t.begin_c, idx);
        cur.insert(cur.begin(), pal);
        nb_meta++;
        f.mod_has_palette = true;
        continue;
      }
      if (t.id == 0) {
        REQUIRE((size_t)t.begin_c + 3 <= cur.size(), "reversible colour transform out of range");
        const Chan &a = cur[t.begin_c], &b2 = cur[t.begin_c + 1], &c2 = cur[t.begin_c + 2];
        REQUIRE(a.w == b2.w && a.h == b2.h && a.w == c2.w && a.h == c2.h, "reversible colour transform on channels of different size");
        fwd.push_back({0, a.plane, b2.plane, c2.plane, (int32_t)t.rct_type});
        continue;
      }
      f.mod_has_squeeze = true;
      std::vector<ParsedFrame::ModSqueeze> sq = t.squeezes;
      if (sq.empty()) {   // default parameters: chroma first, then alternate until no side exceeds 8
        const int nb = (int)(cur.size() - nb_meta);   // (meta channels - a palette - are left alone)
        REQUIRE(nb > 0, "squeeze without channels");
        const uint32_t m0 = (uint32_t)nb_meta;
        int w = cur[m0].w, h = cur[m0].h;
        if (nb > 2 && cur[m0 + 1].w == w && cur[m0 + 1].h == h) {
          sq.push_back({true, false, m0 + 1, 2});
          sq.push_back({false, false, m0 + 1, 2});
        }
        ParsedFrame::ModSqueeze p{true, true, m0, (uint32_t)nb};
        if (h > w && h > 8) { p.horizontal = false; sq.push_back(p); h = (h + 1) / 2; }
        while (w > 8 || h > 8) {
          if (w > 8) { p.horizontal = true; sq.push_back(p); w = (w + 1) / 2; }
          if (h > 8) { p.horizontal = false; sq.push_back(p); h = (h + 1) / 2; }
        }
      }
      for (auto& q : sq) {
        REQUIRE(q.num_c > 0 && (size_t)q.begin_c + q.num_c <= cur.size(), "squeeze channel range");
        REQUIRE(q.begin_c >= nb_meta, "squeeze of meta channels is not supported yet");
        const uint32_t end_c = q.begin_c + q.num_c - 1;
        const size_t offset = q.in_place ? end_c + 1 : cur.size();
        for (uint32_t c = q.begin_c; c <= end_c; c++) {
          Chan src = cur[c], avg = src, res = src;
          if (q.horizontal) { avg.w = (src.w + 1) / 2; avg.hshift++; res.w = src.w - avg.w; res.hshift = avg.hshift; }
          else { avg.h = (src.h + 1) / 2; avg.vshift++; res.h = src.h - avg.h; res.vshift = avg.vshift; }
          avg.plane = (int32_t)f.mod_planes.size(); f.mod_planes.push_back({avg.w, avg.h});
          res.plane = (int32_t)f.mod_planes.size(); f.mod_planes.push_back({res.w, res.h});
          fwd.push_back({q.horizontal ? 1 : 2, avg.plane, res.plane, src.plane, 0});
          cur[c] = avg;
          cur.insert(cur.begin() + offset + (c - q.begin_c), res);
          REQUIRE(cur.size() <= 4096 && f.mod_planes.size() <= 8192, "too many squeeze steps");
        }
      }
    }
    f.mod_coded = cur;
    f.mod_ops.assign(fwd.rbegin(), fwd.rend());
    uint32_t fg = (uint32_t)nb_meta;
    for (; fg < cur.size(); fg++) if (cur[fg].w > (int32_t)f.group_dim || cur[fg].h > (int32_t)f.group_dim) break;
    f.mod_first_group_channel = fg;
  }
  REQUIRE(r.ok(), "truncated LfGlobal");
}

// ---- dequantisation weights
struct BandParams { int n; float b[3][17]; };
float MultF(float v) { return v > 0 ? 1 + v : 1 / (1 - v); }

void BandWeights(int rows, int cols, const BandParams& p, float* out) {
  for (int c = 0; c < 3; c++) {
    float bands[17];
    bands[0] = p.b[c][0];
    REQUIRE(bands[0] >= 1e-8f, "distance bands");
    for (int i = 1; i < p.n; i++) { bands[i] = bands[i - 1] * MultF(p.b[c][i]); REQUIRE(bands[i] >= 1e-8f, "distance bands"); }
    float scale = (p.n - 1) / ((float)std::sqrt(2.0) + 1e-6f);
    float rc = scale / (cols - 1), rr = scale / (rows - 1);
    for (int y = 0; y < rows; y++)
      for (int x = 0; x < cols; x++) {
        float dx = x * rc, dy = y * rr, dist = std::sqrt(dx * dx + dy * dy), w;
        if (p.n == 1) w = bands[0];
        else {
          int i = (int)dist;
          float fr = dist - i, a = bands[i], b2 = i + 1 < p.n ? bands[i + 1] : a;
          w = a * std::pow(b2 / a, fr);
        }
        out[(size_t)c * rows * cols + (size_t)y * cols + x] = w;
      }
  }
}

BandParams MakeBands(int n, std::initializer_list<std::initializer_list<double>> v) {
  BandParams p;
  p.n = n;
  int c = 0;
  for (auto& ch : v) { int i = 0; for (double x : ch) p.b[c][i++] = (float)x; c++; }
  return p;
}

struct QEnc {
  int mode = 6;
  float idw[3][3], d2w[3][6], d4m[3][2], d48m[3], afv[3][9];
  BandParams dct, afv44;
};

const int kReqS[kNumQuantTables] = {1, 1, 1, 1, 2, 4, 1, 1, 2, 1, 1, 8, 4, 16, 8, 32, 16};   // short side in blocks
const int kReqL[kNumQuantTables] = {1, 1, 1, 1, 2, 4, 2, 4, 4, 1, 1, 8, 8, 16, 16, 32, 32};  // long side

QEnc Library(int q) {
  QEnc e;
  auto large = [](double m, bool rect) {
    double a = rect ? 23629.073922049845 : 26629.073922049845, b = rect ? 8611.3238710010046 : 9311.3238710010046,
           c = rect ? 4492.2486445538634 : 4992.2486445538634;
    return MakeBands(8, {{m * a, -1.025, -0.78, -0.65012, -0.19041574084286472, -0.20819395464, -0.421064, -0.32733845535848671},
                         {m * b, -0.3041958212306401, -0.3633036457487539, -0.35660379990111464, -0.3443074455424403, -0.33699592683512467,
                          -0.30180866526242109, -0.27321683125358037},
                         {m * c, -1.2, -1.2, -0.8, -0.7, -0.7, -0.4, -0.5}});
  };
  BandParams p48 = MakeBands(4, {{2198.050556016380522, -0.96269623020744692, -0.76194253026666783, -0.6551140670773547},
                                 {764.3655248643528689, -0.92630200888366945, -0.9675229603596517, -0.27845290869168118},
                                 {527.107573587542228, -1.4594385811273854, -1.450082094097871593, -1.5843722511996204}});
  BandParams p44 = MakeBands(4, {{2200.0, 0.0, 0.0, 0.0}, {392.0, 0.0, 0.0, 0.0}, {112.0, -0.25, -0.25, -0.5}});
  switch (q) {
    case 0: e.dct = MakeBands(6, {{3150.0, 0.0, -0.4, -0.4, -0.4, -2.0}, {560.0, 0.0, -0.3, -0.3, -0.3, -0.3}, {512.0, -2.0, -1.0, 0.0, -1.0, -2.0}}); break;
    case 1: {
      e.mode = 1;
      const float w[3][3] = {{280.0f, 3160.0f, 3160.0f}, {60.0f, 864.0f, 864.0f}, {18.0f, 200.0f, 200.0f}};
      memcpy(e.idw, w, sizeof(w));
      break;
    }
    case 2: {
      e.mode = 2;
      const float w[3][6] = {{3840.0f, 2560.0f, 1280.0f, 640.0f, 480.0f, 300.0f}, {960.0f, 640.0f, 320.0f, 180.0f, 140.0f, 120.0f},
                             {640.0f, 320.0f, 128.0f, 64.0f, 32.0f, 16.0f}};
      memcpy(e.d2w, w, sizeof(w));
      break;
    }
    case 3: e.mode = 3; e.dct = p44; for (auto& m : e.d4m) m[0] = m[1] = 1.0f; break;
    case 4:
      e.dct = MakeBands(7, {{8996.8725711814115328, -1.3000777393353804, -0.49424529824571225, -0.439093774457103443, -0.6350101832695744,
                             -0.90177264050827612, -1.6162099239887414},
                            {3191.48366296844234752, -0.67424582104194355, -0.80745813428471001, -0.44925837484843441, -0.35865440981033403,
                             -0.31322389111877305, -0.37615025315725483},
                            {1157.50408145487200256, -2.0531423165804414, -1.4, -0.50687130033378396, -0.42708730624733904,
                             -1.4856834539296244, -4.9209142884401604}});
      break;
    case 5:
      e.dct = MakeBands(8, {{15718.40830982518931456, -1.025, -0.98, -0.9012, -0.4, -0.48819395464, -0.421064, -0.27},
                            {7305.7636810695983104, -0.8041958212306401, -0.7633036457487539, -0.55660379990111464, -0.49785304658857626,
                             -0.43699592683512467, -0.40180866526242109, -0.27321683125358037},
                            {3803.53173721215041536, -3.060733579805728, -2.0413270132490346, -2.0235650159727417, -0.5495389509954993, -0.4,
                             -0.4, -0.3}});
      break;
    case 6:
      e.dct = MakeBands(7, {{7240.7734393502, -0.7, -0.7, -0.2, -0.2, -0.2, -0.5}, {1448.15468787004, -0.5, -0.5, -0.5, -0.2, -0.2, -0.2},
                            {506.854140754517, -1.4, -0.2, -0.5, -0.5, -1.5, -3.6}});
      break;
    case 7:
      e.dct = MakeBands(8, {{16283.2494710648897, -1.7812845336559429, -1.6309059012653515, -1.0382179034313539, -0.85, -0.7, -0.9,
                             -1.2360638576849587},
                            {5089.15750884921511936, -0.320049391452786891, -0.35362849922161446, -0.30340000000000003, -0.61, -0.5, -0.5, -0.6},
                            {3397.77603275308720128, -0.321327362693153371, -0.34507619223117997, -0.70340000000000003, -0.9, -1.0, -1.0,
                             -1.1754605576265209}});
      break;
    case 8:
      e.dct = MakeBands(8, {{13844.97076442300573, -0.97113799999999995, -0.658, -0.42026, -0.22712, -0.2206, -0.226, -0.6},
                            {4798.964084220744293, -0.61125308982767057, -0.83770786552491361, -0.79014862079498627, -0.2692727459704829,
                             -0.38272769465388551, -0.22924222653091453, -0.20719098826199578},
                            {1807.236946760964614, -1.2, -1.2, -0.7, -0.7, -0.7, -0.4, -0.5}});
      break;
    case 9: e.mode = 4; e.dct = p48; for (auto& m : e.d48m) m = 1.0f; break;
    case 10: {
      e.mode = 5; e.dct = p48; e.afv44 = p44;
      const float w[3][9] = {{3072.0f, 3072.0f, 256.0f, 256.0f, 256.0f, 414.0f, 0.0f, 0.0f, 0.0f},
                             {1024.0f, 1024.0f, 50.0f, 50.0f, 50.0f, 58.0f, 0.0f, 0.0f, 0.0f},
                             {384.0f, 384.0f, 12.0f, 12.0f, 12.0f, 22.0f, -0.25f, -0.25f, -0.25f}};
      memcpy(e.afv, w, sizeof(w));
      break;
    }
    case 11: e.dct = large(0.9, false); break;
    case 12: e.dct = large(0.65, true); break;
    case 13: e.dct = large(1.8, false); break;
    case 14: e.dct = large(1.3, true); break;
    case 15: e.dct = large(3.6, false); break;
    case 16: e.dct = large(2.6, true); break;
  }
  return e;
}

void WeightsFor(int q, const QEnc& e, std::vector<float>& table) {
  int rows = 8 * kReqS[q], cols = 8 * kReqL[q];
  size_t n = (size_t)rows * cols;
  std::vector<float> w(3 * n, 0.f);
  if (e.mode != 6) REQUIRE(n == 64, "8x8-only quant encoding used for a larger table");
  switch (e.mode) {
    case 6: BandWeights(rows, cols, e.dct, w.data()); break;
    case 1:
      for (int c = 0; c < 3; c++) {
        for (int i = 0; i < 64; i++) w[64 * c + i] = e.idw[c][0];
        w[64 * c + 1] = w[64 * c + 8] = e.idw[c][1];
        w[64 * c + 9] = e.idw[c][2];
      }
      break;
    case 2:
      for (int c = 0; c < 3; c++) {
        float* p = &w[64 * c];
        p[0] = 1.f;
        for (int i = 0, s = 1; i < 3; i++, s *= 2) {
          for (int y = 0; y < s; y++) for (int x = s; x < 2 * s; x++) p[y * 8 + x] = p[x * 8 + y] = e.d2w[c][2 * i];
          for (int y = s; y < 2 * s; y++) for (int x = s; x < 2 * s; x++) p[y * 8 + x] = e.d2w[c][2 * i + 1];
        }
      }
      break;
    case 3: {
      float w4[48];
      BandWeights(4, 4, e.dct, w4);
      for (int c = 0; c < 3; c++) {
        for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) w[64 * c + y * 8 + x] = w4[16 * c + (y / 2) * 4 + x / 2];
        w[64 * c + 1] /= e.d4m[c][0]; w[64 * c + 8] /= e.d4m[c][0]; w[64 * c + 9] /= e.d4m[c][1];
      }
      break;
    }
    case 4: {
      float w48[96];
      BandWeights(4, 8, e.dct, w48);
      for (int c = 0; c < 3; c++) {
        for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) w[64 * c + y * 8 + x] = w48[32 * c + (y / 2) * 8 + x];
        w[64 * c + 8] /= e.d48m[c];
      }
      break;
    }
    case 5: {
      static const float kFreq[16] = {0, 0, 0.8517778890324296f, 5.37778436506804f, 0, 0, 4.734747904497923f, 5.449245381693219f,
                                      1.6598270267479331f, 4, 7.275749096817861f, 10.423227632456525f, 2.662932286148962f,
                                      7.630657783650829f, 8.962388608184032f, 12.97166202570235f};
      float w48[96], w44[48];
      BandWeights(4, 8, e.dct, w48);
      BandWeights(4, 4, e.afv44, w44);
      const float lo = 0.8517778890324296f, hi = 12.97166202570235f - lo + 1e-6f;
      for (int c = 0; c < 3; c++) {
        float bands[4] = {e.afv[c][5], 0, 0, 0};
        for (int i = 1; i < 4; i++) bands[i] = bands[i - 1] * MultF(e.afv[c][i + 5]);
        float* p = &w[64 * c];
        p[0] = 1;
        p[1 * 8 + 0] = e.afv[c][0]; p[0 * 8 + 1] = e.afv[c][1];
        p[2 * 8 + 0] = e.afv[c][2]; p[0 * 8 + 2] = e.afv[c][3]; p[2 * 8 + 2] = e.afv[c][4];
        for (int y = 0; y < 4; y++)
          for (int x = 0; x < 4; x++) {
            if (x < 2 && y < 2) continue;
            float pos = (kFreq[y * 4 + x] - lo) * 3 / hi;
            int i = (int)pos;
            REQUIRE(i + 1 < 4, "afv band index");
            p[2 * y * 8 + 2 * x] = bands[i] * std::pow(bands[i + 1] / bands[i], pos - i);
          }
        for (int y = 0; y < 4; y++) for (int x = 0; x < 8; x++) if (x || y) p[(2 * y + 1) * 8 + x] = w48[32 * c + y * 8 + x];
        for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) if (x || y) p[2 * y * 8 + 2 * x + 1] = w44[16 * c + y * 4 + x];
      }
      break;
    }
    default: Fail("RAW quantisation tables are not supported yet");
  }
  table.resize(3 * n);
  for (size_t i = 0; i < 3 * n; i++) {
    REQUIRE(w[i] > 1e-8f && w[i] < 1e8f, "quantisation weight out of range");
    table[i] = 1.0f / w[i];
  }
}

void ReadBands(Bits& r, BandParams& p) {
  p.n = r.u(4) + 1;
  for (int c = 0; c < 3; c++) {
    for (int i = 0; i < p.n; i++) p.b[c][i] = r.F16();
    p.b[c][0] *= 64;
  }
}

const int kBucketStrategy[kNumOrders] = {0, 1, 4, 5, 6, 8, 10, 18, 19, 21, 22, 24, 25};

void ReadHfGlobal(Bits& r, ParsedFrame& f, std::vector<float> custom_dq[kNumQuantTables]) {
  f.dq_default = r.b();
  if (!f.dq_default) {
    for (int q = 0; q < kNumQuantTables; q++) {
      QEnc e;
      e.mode = r.u(3);
      switch (e.mode) {
        case 0: e = Library(q); break;
        case 1: for (auto& c : e.idw) for (auto& v : c) v = r.F16() * 64; break;
        case 2: for (auto& c : e.d2w) for (auto& v : c) v = r.F16() * 64; break;
        case 3: for (auto& c : e.d4m) for (auto& v : c) v = r.F16(); ReadBands(r, e.dct); break;
        case 4: for (auto& v : e.d48m) v = r.F16(); ReadBands(r, e.dct); break;
        case 5:
          for (auto& c : e.afv) { for (auto& v : c) v = r.F16(); for (int i = 0; i < 6; i++) c[i] *= 64; }
          ReadBands(r, e.dct); ReadBands(r, e.afv44);
          break;
        case 6: ReadBands(r, e.dct); break;
        default: Fail("RAW quantisation tables are not supported yet");
      }
      WeightsFor(q, e, custom_dq[q]);
    }
  }
  f.num_presets = 1 + r.u(CeilLog2(f.ng));
  REQUIRE(f.encoding == 0 || f.num_passes == 1, "multi-pass Modular frames are not supported yet");
  f.extra_passes.assign(f.num_passes - 1, ParsedFrame::PassCodes());
  for (uint32_t pass = 0; pass < f.num_passes; pass++) {
    std::vector<uint16_t> (*orders)[3] = pass ? f.extra_passes[pass - 1].custom_order : f.custom_order;
    HostCode& acode = pass ? f.extra_passes[pass - 1].acode : f.acode;
    uint32_t used = r.U32(V(0x5F), V(0x13), V(0), B(kNumOrders));
    if (used) {
      HostCode c;
      ReadCode(r, 8, c);
      SymReader sr(c, r);
      const StaticTables& st = GetStaticTables();
      for (int o = 0; o < kNumOrders; o++) {
        if (!(used >> o & 1)) continue;
        const std::vector<uint16_t>& nat = st.natural_order[o];
        int s = kBucketStrategy[o];
        size_t llf = (size_t)kCoveredX[s] * kCoveredY[s];
        for (int ch = 0; ch < 3; ch++) {
          std::vector<uint32_t> perm;
          ReadPermutation(sr, llf, nat.size(), perm);
          orders[o][ch].resize(nat.size());
          for (size_t k = 0; k < nat.size(); k++) orders[o][ch][k] = nat[perm[k]];
        }
      }
      REQUIRE(sr.Final(), "coefficient orders: ANS final state");
    }
    ReadCode(r, (size_t)f.num_presets * f.num_block_ctx * 495, acode);
  }
  REQUIRE(r.ok(), "truncated HfGlobal");
}

uint32_t BE32(const uint8_t* p) { return (uint32_t)p[0] << 24 | p[1] << 16 | p[2] << 8 | p[3]; }

// `brob` payloads: Brotli streams.  The decoder library is part of the base image (libbrotlidec.so.1) but its headers are not, and
// the product must not depend on it at link time: it is bound at first use through its stable C ABI.
void BrotliDecompress(const uint8_t* in, size_t in_size, std::vector<uint8_t>& out) {
  typedef void* (*CreateFn)(void*, void*, void*);
  typedef void (*DestroyFn)(void*);
  typedef int (*StreamFn)(void*, size_t*, const uint8_t**, size_t*, uint8_t**, size_t*);
  static std::once_flag once;
  static CreateFn create = nullptr;
  static DestroyFn destroy = nullptr;
  static StreamFn stream = nullptr;
  std::call_once(once, [] {
    void* h = dlopen("libbrotlidec.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libbrotlidec.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    create = (CreateFn)dlsym(h, "BrotliDecoderCreateInstance");
    destroy = (DestroyFn)dlsym(h, "BrotliDecoderDestroyInstance");
    stream = (StreamFn)dlsym(h, "BrotliDecoderDecompressStream");
  });
  if (!create || !destroy || !stream) Fail("the file holds Brotli-compressed metadata (brob box) and libbrotlidec is not available");
  void* st = create(nullptr, nullptr, nullptr);
  if (!st) throw std::bad_alloc();
  out.clear();
  std::vector<uint8_t> chunk(1 << 16);
  size_t avail_in = in_size;
  const uint8_t* next_in = in;
  int r;
  do {
    size_t avail_out = chunk.size();
    uint8_t* next_out = chunk.data();
    r = stream(st, &avail_in, &next_in, &avail_out, &next_out, nullptr);
    out.insert(out.end(), chunk.data(), next_out);
    if (out.size() > ((size_t)1 << 30)) { r = 0; break; }   // metadata of a gigabyte: refuse
  } while (r == 3);   // BROTLI_DECODER_RESULT_NEEDS_MORE_OUTPUT
  destroy(st);
  if (r != 1) Fail("corrupt Brotli stream in a brob box");   // BROTLI_DECODER_RESULT_SUCCESS
}

void SplitContainer(const uint8_t* data, size_t size, ParsedFrame& f) {
  static const uint8_t kSig[12] = {0, 0, 0, 0xC, 'J', 'X', 'L', ' ', 0xD, 0xA, 0x87, 0xA};
  if (size >= 2 && data[0] == 0xFF && data[1] == 0x0A) {
    f.cs = data; f.cs_size = size; f.cs_file_offset = 0;
    return;
  }
  if (!(size >= 12 && !memcmp(data, kSig, 12))) Fail("not a JPEG XL file", stInvalidSignature);
  f.is_container = true;
  size_t pos = 0;
  int parts = 0;
  while (pos + 8 <= size) {
    uint64_t box = BE32(data + pos);
    const uint8_t* type = data + pos + 4;
    size_t hdr = 8;
    if (box == 1) {
      REQUIRE(pos + 16 <= size, "truncated box header");
      box = (uint64_t)BE32(data + pos + 8) << 32 | BE32(data + pos + 12);
      hdr = 16;
    } else if (box == 0) box = size - pos;
    REQUIRE(box >= hdr && box <= size - pos, "box size out of range");
    const uint8_t* pl = data + pos + hdr;
    size_t n = box - hdr;
    bool is_c = !memcmp(type, "jxlc", 4), is_p = !memcmp(type, "jxlp", 4);
    if (is_c || is_p) {
      if (is_p) { REQUIRE(n >= 4, "jxlp box too small"); pl += 4; n -= 4; }
      if (parts == 0) { f.cs = pl; f.cs_size = n; f.cs_file_offset = pl - data; }
      else {
        if (parts == 1) f.cs_copy.assign(f.cs, f.cs + f.cs_size);
        f.cs_copy.insert(f.cs_copy.end(), pl, pl + n);
        f.cs_contiguous = false;
      }
      parts++;
    } else {
      // Brotli-compressed boxes are handed to the host under their inner type with the decompressed payload, exactly as the
      // reference sees them (JxlDecoderSetDecompressBoxes, Decoder/JxlDecoder.cpp:435; JxlDecoderGetBoxType(..., JXL_TRUE) :691).
      uint8_t inner[4];
      if (!memcmp(type, "brob", 4)) {
        REQUIRE(n >= 4, "brob box too small");
        memcpy(inner, pl, 4);
        if (!memcmp(inner, "Exif", 4) || !memcmp(inner, "xml ", 4)) {
          f.owned_boxes.emplace_back();
          BrotliDecompress(pl + 4, n - 4, f.owned_boxes.back());
          pl = f.owned_boxes.back().data();
          n = f.owned_boxes.back().size();
        }
        type = inner;
      }
      if (!memcmp(type, "Exif", 4)) {
        if (!f.have_exif) {   // first Exif box wins, even an empty one (Decoder/JxlDecoder.cpp:697-719)
          f.have_exif = true; f.exif = pl; f.exif_size = n;
          f.meta_in_order.push_back(ParsedFrame::MetaBox{true, pl, n});
        }
      } else if (!memcmp(type, "xml ", 4)) {
        f.xml.emplace_back(pl, n);
        f.meta_in_order.push_back(ParsedFrame::MetaBox{false, pl, n});
      }
    }
    pos += box;
  }
  REQUIRE(parts > 0, "container holds no codestream");
  if (!f.cs_contiguous) { f.cs = f.cs_copy.data(); f.cs_size = f.cs_copy.size(); }
}

}  // namespace

// ------------------------------------------------------------------ colour encodings
namespace {
void Inv3(const double m[9], double o[9]) {
  const double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
  o[0] = (m[4] * m[8] - m[5] * m[7]) / d; o[1] = (m[2] * m[7] - m[1] * m[8]) / d; o[2] = (m[1] * m[5] - m[2] * m[4]) / d;
  o[3] = (m[5] * m[6] - m[3] * m[8]) / d; o[4] = (m[0] * m[8] - m[2] * m[6]) / d; o[5] = (m[2] * m[3] - m[0] * m[5]) / d;
  o[6] = (m[3] * m[7] - m[4] * m[6]) / d; o[7] = (m[1] * m[6] - m[0] * m[7]) / d; o[8] = (m[0] * m[4] - m[1] * m[3]) / d;
}
// RGB -> XYZ of a set of primaries with a D65 white point, from the chromaticities
void RgbToXyz(const double xy[3][2], double m[9]) {
  const double wx = 0.3127, wy = 0.3290;
  double p[9], pi[9];
  for (int c = 0; c < 3; c++) { p[c] = xy[c][0] / xy[c][1]; p[3 + c] = 1.0; p[6 + c] = (1.0 - xy[c][0] - xy[c][1]) / xy[c][1]; }
  Inv3(p, pi);
  const double W[3] = {wx / wy, 1.0, (1.0 - wx - wy) / wy};
  for (int c = 0; c < 3; c++) {
    const double sc = pi[c * 3] * W[0] + pi[c * 3 + 1] * W[1] + pi[c * 3 + 2] * W[2];
    for (int r = 0; r < 3; r++) m[r * 3 + c] = p[r * 3 + c] * sc;
  }
}
}  // namespace

// The enumerated encodings the reference's host knows by name; everything else would take its ICC route.  XYB decodes to linear
// sRGB; other primaries are a 3x3 matrix on the linear values (folded into the inverse opsin matrix by the caller).
ColorPlan PlanColor(const ParsedFrame& f) {
  ColorPlan p;
  const ColorInfo& c = f.color;
  const uint32_t tf = c.all_default ? 13 : c.tf, wp = c.all_default ? 1 : c.white_point, pr = c.all_default ? 1 : c.primaries;
  const uint32_t cs = c.all_default ? 0 : c.color_space;
  if (c.want_icc) {
    // Samples that were never converted to XYB (lossless / original-profile streams; CMYK) ARE in the profile's space: hand the
    // profile over.  An XYB stream would need the profile evaluated (the reference lets the decoder library's colour management do
    // it, Decoder/JxlDecoder.cpp:617-632); without one this path takes the reference's own fallback (:586-601): sRGB out.
    if (!f.xyb_encoded) { p.report_icc = true; p.transfer = 0; return p; }
    IccModel model;
    if (IccBuildModel(f.icc.data(), f.icc.size(), &model) && model.gray == (cs == 1)) {
      // a matrix / TRC profile: decode into its space (what the reference gets from its decoder library + colour management, :617-632)
      p.report_icc = true;
      p.transfer = 5;
      for (int k = 0; k < 9; k++) p.from_srgb[k] = (float)model.from_linear_srgb[k];
      for (int c = 0; c < 3; c++) p.trc_lut.insert(p.trc_lut.end(), model.from_linear[c].begin(), model.from_linear[c].end());
      return p;
    }
    p.transfer = 1;
    p.known_profile = cs == 1 ? KnownColorProfile_GraySrgbTRC : KnownColorProfile_Srgb;
    return p;
  }
  p.transfer = tf == 8 ? 0 : (tf == 13 ? 1 : (tf == 1 ? 2 : (tf == 16 ? 3 : -1)));
  const bool named = !c.have_gamma && wp == 1 &&
                     ((cs == 0 && ((tf == 8 && (pr == 1 || pr == 9)) || (tf == 13 && (pr == 1 || pr == 11)) || (tf == 1 && pr == 1) || (tf == 16 && pr == 9))) ||
                      (cs == 1 && (tf == 8 || tf == 13)));
  if (!named) {
    // Not one of the host's eight named profiles (Decoder/JxlDecoder.cpp:36-108): the reference then asks its decoder library for an ICC
    // profile of the target data (:652-682).  Spaces given by chromaticities + a power / sRGB / BT.709 / linear curve are decoded into
    // and described by a synthesised matrix / TRC profile; PQ and HLG outside BT.2100-PQ, "unknown" curves and XYB-as-output are not.
    p.known_profile = -1;
    if (cs != 0 && cs != 1) return p;
    double white[2] = {0.3127, 0.3290}, prim[3][2] = {{0.639998686, 0.330010138}, {0.300003784, 0.600003357}, {0.150002046, 0.059997204}};
    if (wp == 2) { white[0] = c.white_xy[0]; white[1] = c.white_xy[1]; }
    else if (wp == 10) { white[0] = white[1] = 1.0 / 3; }
    else if (wp == 11) { white[0] = 0.314; white[1] = 0.351; }
    else if (wp != 1) return p;
    if (cs == 0) {
      static const double kP3[3][2] = {{0.680, 0.320}, {0.265, 0.690}, {0.150, 0.060}}, k2100[3][2] = {{0.708, 0.292}, {0.170, 0.797}, {0.131, 0.046}};
      if (pr == 2) memcpy(prim, c.prim_xy, sizeof(prim));
      else if (pr == 11) memcpy(prim, kP3, sizeof(prim));
      else if (pr == 9) memcpy(prim, k2100, sizeof(prim));
      else if (pr != 1) return p;
    }
    if (white[1] < 1e-3 || prim[0][1] < 1e-3 || prim[1][1] < 1e-3 || prim[2][1] < 1e-3) return p;
    IccCurveSpec curve;
    if (c.have_gamma) { curve.kind = 4; curve.gamma = c.gamma * 1e-7; if (!(curve.gamma > 0.01 && curve.gamma <= 1.0)) return p; }
    else if (tf == 8) curve.kind = 0;
    else if (tf == 13) curve.kind = 1;
    else if (tf == 1) curve.kind = 2;
    else if (tf == 17) { curve.kind = 4; curve.gamma = 1 / 2.6; }
    else return p;
    double m[9];
    if (cs == 0) { if (!MatrixFromLinearSrgb(prim, white, m)) return p; for (int k = 0; k < 9; k++) p.from_srgb[k] = (float)m[k]; }
    if (curve.kind == 4) {   // power curve: as a table over sqrt(linear), like the evaluated ICC curves
      p.transfer = 5;
      p.trc_lut.resize(3 * (size_t)kIccInvLut);
      for (int i = 0; i < kIccInvLut; i++) {
        const double t = (double)i / (kIccInvLut - 1);
        const float v = (float)std::pow(t * t, curve.gamma);
        p.trc_lut[i] = p.trc_lut[kIccInvLut + i] = p.trc_lut[2 * kIccInvLut + i] = v;
      }
    } else p.transfer = curve.kind;
    p.report_icc = true;
    p.icc_out = IccSynthesize(cs == 1, prim, white, curve, c.all_default ? 1 : c.rendering_intent);
    return p;
  }
  if (cs == 0) {   // RGB, D65 (Decoder/JxlDecoder.cpp:42-88)
    if (tf == 8) p.known_profile = pr == 1 ? KnownColorProfile_LinearSrgb : (pr == 9 ? KnownColorProfile_Rec2020Linear : -1);
    else if (tf == 13) p.known_profile = pr == 1 ? KnownColorProfile_Srgb : (pr == 11 ? KnownColorProfile_DisplayP3 : -1);
    else if (tf == 1) p.known_profile = pr == 1 ? KnownColorProfile_Rec709 : -1;
    else if (pr == 9 && tf == 16) p.known_profile = KnownColorProfile_Rec2020PQ;
  } else if (cs == 1) {   // gray, D65 (:90-104)
    if (tf == 8) p.known_profile = KnownColorProfile_LinearGray;
    else if (tf == 13) p.known_profile = KnownColorProfile_GraySrgbTRC;
  }
  if (p.known_profile >= 0 && cs == 0 && pr != 1) {
    static const double kSrgb[3][2] = {{0.639998686, 0.330010138}, {0.300003784, 0.600003357}, {0.150002046, 0.059997204}};
    static const double kP3[3][2] = {{0.680, 0.320}, {0.265, 0.690}, {0.150, 0.060}};
    static const double k2100[3][2] = {{0.708, 0.292}, {0.170, 0.797}, {0.131, 0.046}};
    double ms[9], mt[9], mti[9];
    RgbToXyz(kSrgb, ms);
    RgbToXyz(pr == 11 ? kP3 : k2100, mt);
    Inv3(mt, mti);
    for (int r = 0; r < 3; r++)
      for (int k = 0; k < 3; k++) p.from_srgb[r * 3 + k] = (float)(mti[r * 3] * ms[k] + mti[r * 3 + 1] * ms[3 + k] + mti[r * 3 + 2] * ms[6 + k]);
  }
  return p;
}

void BuildAliasTable(const std::vector<int>& counts, uint32_t log_alpha, uint64_t* out) { BuildAlias(counts, log_alpha, out); }

// Test hooks for the encoder's host writers (CPU tests): read back what host_write.cc wrote with the decoder's own readers.
bool ReadBackTokens(const uint8_t* bytes, size_t nbytes, size_t num_ctx, const uint32_t* ctxs, const uint32_t* values, size_t n,
                    std::string* why) {
  try {
    Bits r(bytes, nbytes);
    HostCode c;
    ReadCode(r, num_ctx, c);
    SymReader sr(c, r);
    for (size_t i = 0; i < n; i++) {
      const uint32_t v = sr.Get(ctxs[i]);
      if (v != values[i]) { *why = "token " + std::to_string(i) + ": got " + std::to_string(v) + ", wrote " + std::to_string(values[i]); return false; }
    }
    if (!sr.Final()) { *why = "final ANS state"; return false; }
    return true;
  } catch (const std::exception& e) {
    *why = e.what();
    return false;
  }
}

bool ReadBackTree(const uint8_t* bytes, size_t nbytes, std::vector<DevTreeNode>* tree, std::string* why) {
  try {
    Bits r(bytes, nbytes);
    ParsedFrame f;
    ReadTree(r, f, 1 << 16);
    *tree = f.tree;
    return true;
  } catch (const std::exception& e) {
    *why = e.what();
    return false;
  }
}

const uint8_t kCoveredX[kNumStrategies] = {1, 1, 1, 1, 2, 4, 1, 2, 1, 4, 2, 4, 1, 1, 1, 1, 1, 1, 8, 4, 8, 16, 8, 16, 32, 16, 32};
const uint8_t kCoveredY[kNumStrategies] = {1, 1, 1, 1, 2, 4, 2, 1, 4, 1, 4, 2, 1, 1, 1, 1, 1, 1, 8, 8, 4, 16, 16, 8, 32, 32, 16};
const uint8_t kStrategyOrderBucket[kNumStrategies] = {0, 1, 1, 1, 2, 3, 4, 4, 5, 5, 6, 6, 1, 1, 1, 1, 1, 1, 7, 8, 8, 9, 10, 10, 11, 12, 12};
const uint8_t kStrategyQuantTable[kNumStrategies] = {0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16};

const StaticTables& GetStaticTables() {
  static StaticTables t;
  static std::once_flag once;
  std::call_once(once, [] {
    // natural (zig-zag) orders: LLF first in raster order, then anti-diagonals of the long x long grid
    // restricted to every (long/short)-th row
    for (int o = 0; o < kNumOrders; o++) {
      int s = kBucketStrategy[o];
      size_t L = std::max(kCoveredX[s], kCoveredY[s]), S = std::min(kCoveredX[s], kCoveredY[s]);
      size_t ratio = L / S, rshift = CeilLog2(ratio), N = L * 8;
      std::vector<uint16_t>& ord = t.natural_order[o];
      ord.assign(L * S * 64, 0);
      size_t next = L * S;
      for (size_t d = 0; d < 2 * N - 1; d++) {
        size_t lo = d < N ? 0 : d - (N - 1), hi = d < N ? d : N - 1;
        for (size_t j = lo; j <= hi; j++) {
          // walk direction alternates with the diagonal index
          size_t a = d < N ? j : hi - (j - lo), bq = d - a;
          size_t x, y;
          if (d < N) { x = j; y = d - j; if (d & 1) std::swap(x, y); }
          else {
            size_t i = 2 * N - 2 - d, jj = j - lo;   // mirrored second half
            x = N - 1 - (i - jj); y = N - 1 - jj;
            if (i & 1) std::swap(x, y);
          }
          (void)a; (void)bq;
          if (y & (ratio - 1)) continue;
          y >>= rshift;
          size_t idx = (x < L && y < S) ? y * L + x : next++;
          ord[idx] = (uint16_t)(y * N + x);
        }
      }
    }
    for (int q = 0; q < kNumQuantTables; q++) WeightsFor(q, Library(q), t.dq[q]);
    for (int i = 0; i < 6; i++) {
      int N = 8 << i;
      t.basis[i].resize((size_t)N * N);
      for (int k = 0; k < N; k++)
        for (int n = 0; n < N; n++)
          t.basis[i][(size_t)k * N + n] = (float)((k ? std::sqrt(2.0) : 1.0) * std::cos((2 * n + 1) * k * M_PI / (2.0 * N)));
    }
    t.llf_scale.assign(6 * 32, 1.0f);
    for (int i = 0; i < 6; i++) {
      int c = 1 << i;
      for (int k = 1; k < c; k++) {
        double th = k * M_PI / (2.0 * c);
        t.llf_scale[i * 32 + k] = (float)(1.0 / (std::cos(th / 2) * std::cos(th / 4) * std::cos(th / 8)));
      }
    }
  });
  return t;
}

// The embedded ICC profile: size of the predicted stream, its entropy code (41 contexts on the two previous bytes), the bytes, then
// the inverse of the predictor (icc.cc).
void ReadIcc(Bits& r, ParsedFrame& f) {
  const uint64_t n = r.U64();
  REQUIRE(n > 0 && n <= kIccMaxEncodedSize && n <= (uint64_t)f.cs_size * 64, "ICC profile: encoded size");
  HostCode code;
  ReadCode(r, kIccContexts, code);
  SymReader sr(code, r);
  std::vector<uint8_t> enc((size_t)n);
  for (size_t i = 0; i < enc.size(); i++) {
    const uint32_t v = sr.Get(IccContext(i, i > 0 ? enc[i - 1] : 0, i > 1 ? enc[i - 2] : 0));
    REQUIRE(v < 256, "ICC profile: byte out of range");
    enc[i] = (uint8_t)v;
    REQUIRE(r.ok(), "truncated ICC profile");
  }
  REQUIRE(sr.Final(), "ICC profile: ANS final state");
  std::string why;
  if (!IccUnpredict(enc, &f.icc, &why)) Fail(why);
}

void ParseFile(const uint8_t* data, size_t size, bool headers_only, ParsedFrame& f) {
  f = ParsedFrame();
  SplitContainer(data, size, f);
  if (!(f.cs_size >= 2 && f.cs[0] == 0xFF && f.cs[1] == 0x0A)) Fail("invalid codestream signature", stInvalidSignature);
  Bits r(f.cs, f.cs_size);
  r.Skip(16);
  ReadImageHeader(r, f);
  if (f.color.want_icc) ReadIcc(r, f);
  r.Align();
  size_t frame_base = r.pos();
  ReadFrameHeader(r, f);
  ReadToc(r, f, frame_base);
  if (headers_only) return;
  auto depth_ok = [](uint32_t bits, uint32_t exp) { return exp ? ((bits == 32 && exp == 8) || (bits == 16 && exp == 5)) : (bits >= 1 && bits <= 16); };
  if (!depth_ok(f.bits, f.exp_bits)) Fail("only integer samples of up to 16 bits and binary16 / binary32 float samples are supported yet");
  if (f.alpha_index >= 0 && !depth_ok(f.ec[f.alpha_index].bits, f.ec[f.alpha_index].exp_bits))
    Fail("only integer alpha of up to 16 bits and binary16 / binary32 float alpha are supported yet");
  // the reference asks for un-premultiplied output (Decoder/JxlDecoder.cpp:233): premultiplied streams would need the division
  // - done where the samples are written.  VarDCT frames keep their alpha plane in the OUTPUT sample type, so the division there is
  // exact only when that type is the alpha channel's own integer depth; Modular frames divide by the coded sample (any depth).
  if (f.alpha_index >= 0 && f.ec[f.alpha_index].alpha_associated && f.encoding == 0 &&
      (f.exp_bits || f.ec[f.alpha_index].exp_bits || f.ec[f.alpha_index].bits != (f.bits > 8 ? 16u : 8u)))
    Fail("premultiplied alpha whose depth differs from the output samples' is not supported yet on lossy frames");
  for (auto& e : f.ec) if (e.dim_shift) Fail("subsampled extra channels are not supported yet");
  if (PlanColor(f).known_profile < 0 && !PlanColor(f).report_icc)
    Fail("colour encodings other than D65 sRGB / linear sRGB / Display P3 / BT.709 / BT.2020 linear / BT.2020 PQ / gray need an ICC profile, which is not built yet");
  if (f.encoding == 0 && !f.xyb_encoded) Fail("VarDCT frames without XYB are not supported yet");
  if (f.encoding == 1 && f.xyb_encoded) Fail("lossy Modular (XYB) frames are not decoded on the GPU path yet");
  if (f.encoding == 1 && f.ec.size() > (f.alpha_index >= 0 ? 1u : 0u) + (f.black_index >= 0 ? 1u : 0u))
    Fail("extra channels other than one alpha and one black (CMYK) channel are not supported yet");
  if (f.black_index >= 0) {
    // the reference delivers CMYK as 8-bit samples only (SetCmykImageDataUInt8, Decoder/JxlDecoder.cpp:159-215)
    if (f.encoding != 1 || f.xyb_encoded || f.ncolor != 3 || f.bits != 8 || f.exp_bits || f.ec[f.black_index].bits != 8 || f.ec[f.black_index].exp_bits)
      Fail("only 8-bit lossless (Modular, original colour space) CMYK streams are supported");
  }
  if (f.encoding == 0 && f.ec.size() > (f.alpha_index >= 0 ? 1u : 0u)) Fail("extra channels other than alpha are not supported in VarDCT frames yet");
  f.single = f.sec_off.size() == 1;
  {
    Bits s(f.cs + f.sec_off[0], f.sec_size[0]);
    ReadLfGlobal(s, f);
    // single-section frames: LfGlobal | LfGroup | HfGlobal | PassGroup share one bit stream; the kernels continue from here
    f.after_lf_global_bits = f.sec_off[0] * 8 + s.pos();
    f.mod_data_bits = f.after_lf_global_bits;
  }
  if (f.tree_uses_ref) Fail("MA trees using reference-channel properties (16 and up) are not supported on the GPU path yet");
  if (f.encoding == 1) {
    if (!f.has_global_tree) Fail("Modular frames without a global MA tree are not supported on the GPU path yet");
    return;   // nothing else is global in a Modular frame: every group section is decoded on the GPU
  }
  // HfGlobal of a single-section frame starts where the GPU finishes the LF group: the decoder runs the LF stage of such
  // frames first (JxlHipDecoder::PrepassSingle) and then calls ParseHfGlobalAt().
  if (f.single) return;
  {
    Bits s(f.cs + f.sec_off[1 + f.nlf], f.sec_size[1 + f.nlf]);
    ReadHfGlobal(s, f, f.custom_dq);
  }
}

uint64_t ParseHfGlobalAt(ParsedFrame& f, uint64_t bit_pos) {
  const uint64_t sec_bits = f.sec_off[0] * 8;
  if (bit_pos < sec_bits || bit_pos > sec_bits + (uint64_t)f.sec_size[0] * 8) Fail("LF group ends outside its section");
  Bits s(f.cs + f.sec_off[0], f.sec_size[0]);
  s.Skip(bit_pos - sec_bits);
  ReadHfGlobal(s, f, f.custom_dq);
  return sec_bits + s.pos();
}

}  // namespace jxlhip
