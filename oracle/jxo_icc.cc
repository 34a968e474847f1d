// ORACLE — test infrastructure only (see jxo_common.h header).
#include "jxo_icc.h"
#include <cstring>

namespace jxo {

namespace {

const size_t kHdr = 128;
const char* const kKnownTags[] = {"cprt", "wtpt", "bkpt", "rXYZ", "gXYZ", "bXYZ", "kXYZ", "rTRC", "gTRC", "bTRC", "kTRC", "chad", "desc", "chrm", "dmnd", "dmdd", "lumi"};
const char* const kKnownTypes[] = {"XYZ ", "desc", "text", "mluc", "para", "curv", "sf32", "gbd "};

int ClassA(uint32_t b) {   // of the byte before
  if ((b | 32) >= 'a' && (b | 32) <= 'z') return 0;
  if ((b >= '0' && b <= '9') || b == '.' || b == ',') return 1;
  if (b == 0) return 2;
  if (b == 1) return 3;
  if (b < 16) return 4;
  if (b == 255) return 6;
  if (b > 240) return 5;
  return 7;
}
int ClassB(uint32_t b) {   // of the byte two back
  if ((b | 32) >= 'a' && (b | 32) <= 'z') return 0;
  if ((b >= '0' && b <= '9') || b == '.' || b == ',') return 1;
  if (b < 16) return 2;
  if (b > 240) return 3;
  return 4;
}

struct Out {
  std::vector<uint8_t> v;
  void U8(uint32_t b) { v.push_back((uint8_t)b); }
  void Var(uint64_t x) { for (; x > 127; x >>= 7) U8((x & 127) | 128); U8((uint32_t)x); }
  void BE32(uint32_t x) { U8(x >> 24); U8(x >> 16); U8(x >> 8); U8(x); }
  void Str4(const char* s) { for (int i = 0; i < 4; i++) U8((uint8_t)s[i]); }
};
uint32_t BE32At(const std::vector<uint8_t>& d, size_t p) { return (uint32_t)d[p] << 24 | (uint32_t)d[p + 1] << 16 | (uint32_t)d[p + 2] << 8 | d[p + 3]; }

// predicted header as a function of the bytes already known
void HeaderGuess(const uint8_t* known, size_t n, uint32_t total, uint8_t* g) {
  memset(g, 0, kHdr);
  g[0] = total >> 24; g[1] = total >> 16; g[2] = total >> 8; g[3] = total;
  g[8] = 4;
  memcpy(g + 12, "mntrRGB XYZ ", 12);
  memcpy(g + 36, "acsp", 4);
  g[70] = 0xF6; g[71] = 0xD6; g[73] = 1; g[78] = 0xD3; g[79] = 0x2D;
  if (n >= 8) memcpy(g + 80, known + 4, 4);
  if (n >= 41) {
    if (known[40] == 'A') memcpy(g + 41, "PPL", 3);
    if (known[40] == 'M') memcpy(g + 41, "SFT", 3);
  }
  if (n >= 42) {
    if (known[40] == 'S' && known[41] == 'G') memcpy(g + 42, "I ", 2);
    if (known[40] == 'S' && known[41] == 'U') memcpy(g + 42, "NW", 2);
  }
}

uint64_t Var(const std::vector<uint8_t>& d, size_t& p, size_t end) {
  uint64_t v = 0;
  for (int s = 0;; s += 7) {
    JXO_CHECK(p < end && s < 63, "ICC stream: varint");
    const uint8_t b = d[p++];
    v |= (uint64_t)(b & 127) << s;
    if (b < 128) return v;
  }
}

std::vector<uint8_t> Deinterleave(const std::vector<uint8_t>& planes, size_t width) {
  // planes: byte 0 of every word, then byte 1 of every word, ... -> words
  const size_t n = planes.size(), words = (n + width - 1) / width;
  std::vector<uint8_t> out(n);
  size_t src = 0;
  for (size_t b = 0; b < width; b++)
    for (size_t wd = 0; wd < words; wd++) {
      const size_t dst = wd * width + b;
      if (dst < n) out[dst] = planes[src++];
    }
  return out;
}

}  // namespace

uint32_t IccByteContext(size_t index, uint32_t b1, uint32_t b2) { return index <= kHdr ? 0 : 1 + ClassA(b1) + 8 * ClassB(b2); }

std::vector<uint8_t> IccToStream(const std::vector<uint8_t>& icc) {
  Out cmd, data;
  const size_t n = icc.size();
  uint8_t g[kHdr];
  for (size_t i = 0; i < std::min(n, kHdr); i++) {
    HeaderGuess(icc.data(), i, (uint32_t)n, g);
    data.U8((uint8_t)(icc[i] - g[i]));
  }
  size_t pos = kHdr;
  if (n > kHdr) {
    // tag table, when it is well formed
    bool table = n >= kHdr + 4;
    uint32_t ntags = table ? BE32At(icc, kHdr) : 0;
    if (table && (ntags > 4096 || kHdr + 4 + (uint64_t)ntags * 12 > n)) table = false;
    if (!table) cmd.Var(0);
    else {
      cmd.Var((uint64_t)ntags + 1);
      pos = kHdr + 4;
      uint64_t prev_start = kHdr + (uint64_t)ntags * 12, prev_size = 0;   // [spec, recalled]: without the 4 bytes of the tag count
      for (uint32_t t = 0; t < ntags; t++, pos += 12) {
        const char* name = (const char*)&icc[pos];
        const uint32_t start = BE32At(icc, pos + 4), size = BE32At(icc, pos + 8);
        int code = 1;
        for (int k = 0; k < 17; k++) if (!memcmp(name, kKnownTags[k], 4)) code = 4 + k;
        uint64_t want_start = prev_start + prev_size, want_size = prev_size;
        for (const char* x : {"rXYZ", "gXYZ", "bXYZ", "kXYZ", "wtpt", "bkpt", "lumi"}) if (!memcmp(name, x, 4)) want_size = 20;
        uint32_t c = code;
        if (start != want_start) c |= 64;
        if (size != want_size) c |= 128;
        cmd.U8(c);
        if (code == 1) for (int k = 0; k < 4; k++) data.U8((uint8_t)name[k]);
        if (c & 64) cmd.Var(start);
        if (c & 128) cmd.Var(size);
        prev_start = start; prev_size = size;
      }
      cmd.U8(0);   // end of the tag table
    }
    // tag data: recognised 8-byte type starts as commands, XYZ triplets whole, everything else inserted
    size_t run = pos;   // start of the pending insert
    auto flush = [&](size_t upto) {
      if (upto > run) { cmd.U8(1); cmd.Var(upto - run); for (size_t i = run; i < upto; i++) data.U8(icc[i]); }
      run = upto;
    };
    while (pos < n) {
      bool done = false;
      if ((pos & 3) == 0 && pos + 20 <= n && !memcmp(&icc[pos], "XYZ ", 4) && BE32At(icc, pos + 4) == 0) {
        flush(pos);
        cmd.U8(10);
        for (size_t i = pos + 8; i < pos + 20; i++) data.U8(icc[i]);
        pos += 20; run = pos; done = true;
      } else if ((pos & 3) == 0 && pos + 8 <= n && BE32At(icc, pos + 4) == 0) {
        for (int k = 0; k < 8 && !done; k++)
          if (!memcmp(&icc[pos], kKnownTypes[k], 4)) { flush(pos); cmd.U8(16 + k); pos += 8; run = pos; done = true; }
      }
      if (!done) pos++;
    }
    flush(n);
  }
  Out enc;
  enc.Var(n);
  enc.Var(cmd.v.size());
  enc.v.insert(enc.v.end(), cmd.v.begin(), cmd.v.end());
  enc.v.insert(enc.v.end(), data.v.begin(), data.v.end());
  return enc.v;
}

std::vector<uint8_t> IccFromStream(const std::vector<uint8_t>& enc) {
  size_t p = 0;
  const uint64_t total = Var(enc, p, enc.size()), ncmd = Var(enc, p, enc.size());
  JXO_CHECK(total <= (1u << 28) && ncmd <= enc.size() - p, "ICC stream: sizes");
  size_t c = p;
  const size_t cend = p + ncmd;
  size_t d = cend;
  std::vector<uint8_t> out;
  auto data = [&](size_t k) {
    JXO_CHECK(k <= enc.size() - d && out.size() + k <= total, "ICC stream: data overrun");
    out.insert(out.end(), enc.begin() + d, enc.begin() + d + k);
    d += k;
  };
  auto be32 = [&](uint64_t x) { out.push_back(x >> 24); out.push_back(x >> 16); out.push_back(x >> 8); out.push_back(x); };
  auto str4 = [&](const char* s) { out.insert(out.end(), s, s + 4); };
  uint8_t g[kHdr];
  while (out.size() < std::min<uint64_t>(total, kHdr)) {
    HeaderGuess(out.data(), out.size(), (uint32_t)total, g);
    JXO_CHECK(d < enc.size(), "ICC stream: header");
    out.push_back((uint8_t)(enc[d++] + g[out.size()]));
  }
  if (out.size() == total) { JXO_CHECK(c == cend && d == enc.size(), "ICC stream: trailing bytes"); return out; }
  JXO_CHECK(c < cend, "ICC stream: commands missing");
  uint64_t ntags = Var(enc, c, cend);
  if (ntags) {
    ntags--;
    JXO_CHECK(ntags < (1u << 20), "ICC stream: tag count");
    be32(ntags);
    uint64_t prev_start = kHdr + ntags * 12, prev_size = 0;
    while (c < cend) {
      const uint32_t cm = enc[c++], code = cm & 63;
      if (!code) break;
      char name[4];
      if (code == 1) { JXO_CHECK(4 <= enc.size() - d, "ICC stream: tag name"); memcpy(name, &enc[d], 4); d += 4; }
      else if (code == 2) memcpy(name, "rTRC", 4);
      else if (code == 3) memcpy(name, "rXYZ", 4);
      else { JXO_CHECK(code < 4 + 17, "ICC stream: tag code"); memcpy(name, kKnownTags[code - 4], 4); }
      out.insert(out.end(), name, name + 4);
      uint64_t start = prev_start + prev_size, size = prev_size;
      for (const char* x : {"rXYZ", "gXYZ", "bXYZ", "kXYZ", "wtpt", "bkpt", "lumi"}) if (!memcmp(name, x, 4)) size = 20;
      if (cm & 64) start = Var(enc, c, cend);
      if (cm & 128) size = Var(enc, c, cend);
      JXO_CHECK(start <= 0xFFFFFFFFull && size <= 0xFFFFFFFFull && start + 2 * size <= 0xFFFFFFFFull, "ICC stream: tag offset or size exceeds 32 bits");
      be32(start);
      be32(size);
      prev_start = start; prev_size = size;
      if (code == 2) { str4("gTRC"); be32(start); be32(size); str4("bTRC"); be32(start); be32(size); }
      if (code == 3) { str4("gXYZ"); be32(start + size); be32(size); str4("bXYZ"); be32(start + 2 * size); be32(size); prev_start = start + 2 * size; }
      JXO_CHECK(out.size() <= total, "ICC stream: tag table overrun");
    }
  }
  while (c < cend) {
    const uint32_t cm = enc[c++];
    if (cm == 1) data((size_t)Var(enc, c, cend));
    else if (cm == 2 || cm == 3) {
      const size_t k = (size_t)Var(enc, c, cend), at = out.size();
      data(k);
      std::vector<uint8_t> planes(out.begin() + at, out.end());
      std::vector<uint8_t> words = Deinterleave(planes, cm == 2 ? 2 : 4);
      std::copy(words.begin(), words.end(), out.begin() + at);
    } else if (cm == 4) {
      JXO_CHECK(c < cend, "ICC stream: predictor flags");
      const uint32_t fl = enc[c++];
      const size_t width = (fl & 3) + 1;
      const int order = (fl >> 2) & 3;
      JXO_CHECK(width != 3 && order != 3, "ICC stream: predictor parameters");
      uint64_t stride = width;
      if (fl & 16) stride = Var(enc, c, cend);
      JXO_CHECK(stride >= width && stride < (out.size() + 3) / 4, "ICC stream: predictor stride");   // no multiplication: 63-bit varint
      const size_t k = (size_t)Var(enc, c, cend), at = out.size();
      data(k);
      std::vector<uint8_t> res(out.begin() + at, out.end());
      if (width > 1) res = Deinterleave(res, width);
      for (size_t i = 0; i < k; i++) {
        const size_t word = at + i - i % width;   // start of the word byte i belongs to
        auto value = [&](size_t q) { uint64_t v = 0; for (size_t b = 0; b < width; b++) v = v << 8 | out[q + b]; return v; };
        const uint64_t a1 = value(word - stride), a2 = value(word - 2 * stride), a3 = value(word - 3 * stride);
        const uint64_t pred = order == 0 ? a1 : (order == 1 ? 2 * a1 - a2 : 3 * a1 - 3 * a2 + a3);
        const uint32_t byte = (uint32_t)(pred >> (8 * (width - 1 - i % width))) & 255;
        out[at + i] = (uint8_t)(res[i] + byte);
      }
    } else if (cm == 10) {
      JXO_CHECK(out.size() + 20 <= total, "ICC stream: XYZ overrun");
      str4("XYZ "); be32(0); data(12);
    } else if (cm >= 16 && cm < 24) {
      JXO_CHECK(out.size() + 8 <= total, "ICC stream: type overrun");
      str4(kKnownTypes[cm - 16]); be32(0);
    } else {
      JXO_CHECK(false, "ICC stream: unknown command");
    }
  }
  JXO_CHECK(d == enc.size() && out.size() == total, "ICC stream: size mismatch");
  return out;
}

}  // namespace jxo
