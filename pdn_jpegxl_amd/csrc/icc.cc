// ICC profile coding of the JPEG XL codestream: see icc.h.
#include "icc.h"
#include <cmath>
#include <cstring>

namespace jxlhip {

namespace {

constexpr size_t kHeaderSize = 128;

// Byte classes of the context model.
uint32_t Kind1(uint8_t b) {
  if ((b >= 'a' && b <= 'z') || (b >= 'A' && b <= 'Z')) return 0;
  if ((b >= '0' && b <= '9') || b == '.' || b == ',') return 1;
  if (b <= 1) return 2 + b;
  if (b < 16) return 4;
  if (b > 240 && b < 255) return 5;
  if (b == 255) return 6;
  return 7;
}
uint32_t Kind2(uint8_t b) {
  if ((b >= 'a' && b <= 'z') || (b >= 'A' && b <= 'Z')) return 0;
  if ((b >= '0' && b <= '9') || b == '.' || b == ',') return 1;
  if (b < 16) return 2;
  if (b > 240) return 3;
  return 4;
}

void PutVarint(uint64_t v, std::vector<uint8_t>* out) {
  while (v >= 128) { out->push_back((uint8_t)(v | 128)); v >>= 7; }
  out->push_back((uint8_t)v);
}
bool GetVarint(const std::vector<uint8_t>& in, size_t* pos, size_t end, uint64_t* v) {
  *v = 0;
  for (int shift = 0; shift < 63; shift += 7) {
    if (*pos >= end) return false;
    const uint8_t b = in[(*pos)++];
    *v |= (uint64_t)(b & 127) << shift;
    if (!(b & 128)) return true;
  }
  return false;
}

// The header every profile is predicted from; bytes that depend on the profile itself are filled in as they become known.
void InitialHeader(uint32_t size, uint8_t* h) {
  memset(h, 0, kHeaderSize);
  h[0] = (uint8_t)(size >> 24); h[1] = (uint8_t)(size >> 16); h[2] = (uint8_t)(size >> 8); h[3] = (uint8_t)size;
  h[8] = 4;
  memcpy(h + 12, "mntr", 4);
  memcpy(h + 16, "RGB ", 4);
  memcpy(h + 20, "XYZ ", 4);
  memcpy(h + 36, "acsp", 4);
  h[70] = 246; h[71] = 214;   // PCS illuminant D50 as s15Fixed16: 0.9642, 1.0, 0.8249
  h[73] = 1;
  h[78] = 211; h[79] = 45;
}
void UpdateHeaderPrediction(const uint8_t* icc, size_t known, uint8_t* h, size_t pos) {
  if (pos == 8 && known >= 8) memcpy(h + 80, icc + 4, 4);   // creator signature repeats the preferred CMM
  if (pos == 41 && known >= 41) {
    if (icc[40] == 'A') memcpy(h + 41, "PPL", 3);
    if (icc[40] == 'M') memcpy(h + 41, "SFT", 3);
  }
  if (pos == 42 && known >= 42) {
    if (icc[40] == 'S' && icc[41] == 'G') { h[42] = 'I'; h[43] = ' '; }
    if (icc[40] == 'S' && icc[41] == 'U') { h[42] = 'N'; h[43] = 'W'; }
  }
}

void PutU32(uint32_t v, std::vector<uint8_t>* out) {
  out->push_back((uint8_t)(v >> 24)); out->push_back((uint8_t)(v >> 16)); out->push_back((uint8_t)(v >> 8)); out->push_back((uint8_t)v);
}
void PutTag(const char* t, std::vector<uint8_t>* out) { out->insert(out->end(), t, t + 4); }

// Bytes of `width`-byte big-endian words were stored plane by plane (all first bytes, then all second bytes, ...): undo that.
void Unshuffle(std::vector<uint8_t>* data, size_t width) {
  const size_t n = data->size(), height = (n + width - 1) / width;
  std::vector<uint8_t> out(n);
  size_t s = 0, j = 0;
  for (size_t i = 0; i < n; i++) {
    out[j] = (*data)[i];
    j += width;
    if (j >= n) j = ++s;
  }
  (void)height;
  data->swap(out);
}

uint32_t Extrapolate(uint32_t p1, uint32_t p2, uint32_t p3, int order) {
  if (order == 0) return p1;
  if (order == 1) return 2 * p1 - p2;
  return 3 * p1 - 3 * p2 + p3;
}
// Prediction of byte i of a run that starts at `start`: the words one, two and three strides back, extrapolated.
uint8_t PredictByte(const std::vector<uint8_t>& d, size_t start, size_t i, size_t stride, size_t width, int order) {
  const size_t pos = start + i;
  if (width == 1) return (uint8_t)Extrapolate(d[pos - stride], d[pos - 2 * stride], d[pos - 3 * stride], order);
  if (width == 2) {
    const size_t p = start + (i & ~(size_t)1);
    auto w16 = [&](size_t q) { return (uint32_t)d[q] << 8 | d[q + 1]; };
    const uint32_t v = Extrapolate(w16(p - stride), w16(p - 2 * stride), w16(p - 3 * stride), order) & 0xFFFF;
    return (uint8_t)((i & 1) ? v : v >> 8);
  }
  const size_t p = start + (i & ~(size_t)3);
  auto w32 = [&](size_t q) { return (uint32_t)d[q] << 24 | (uint32_t)d[q + 1] << 16 | (uint32_t)d[q + 2] << 8 | d[q + 3]; };
  const uint32_t v = Extrapolate(w32(p - stride), w32(p - 2 * stride), w32(p - 3 * stride), order);
  return (uint8_t)(v >> (8 * (3 - (i & 3))));
}

const char* const kTagNames[17] = {"cprt", "wtpt", "bkpt", "rXYZ", "gXYZ", "bXYZ", "kXYZ", "rTRC", "gTRC",
                                   "bTRC", "kTRC", "chad", "desc", "chrm", "dmnd", "dmdd", "lumi"};
const char* const kTypeNames[8] = {"XYZ ", "desc", "text", "mluc", "para", "curv", "sf32", "gbd "};

}  // namespace

uint32_t IccContext(size_t i, uint8_t prev1, uint8_t prev2) {
  if (i <= kHeaderSize) return 0;
  return 1 + Kind1(prev1) + 8 * Kind2(prev2);
}

uint32_t IccDataColorSpace(const uint8_t* icc, size_t size) {
  if (size < 20) return 0;
  return (uint32_t)icc[16] << 24 | (uint32_t)icc[17] << 16 | (uint32_t)icc[18] << 8 | icc[19];
}

void IccPredict(const uint8_t* icc, size_t size, std::vector<uint8_t>* enc) {
  std::vector<uint8_t> commands, data;
  uint8_t h[kHeaderSize];
  InitialHeader((uint32_t)size, h);
  const size_t hn = size < kHeaderSize ? size : kHeaderSize;
  for (size_t i = 0; i < hn; i++) {
    UpdateHeaderPrediction(icc, i, h, i);
    data.push_back((uint8_t)(icc[i] - h[i]));
  }
  if (size > kHeaderSize) {
    commands.push_back(0);            // no tag-table commands
    commands.push_back(1);            // insert ...
    PutVarint(size - kHeaderSize, &commands);
    data.insert(data.end(), icc + kHeaderSize, icc + size);
  }
  enc->clear();
  PutVarint(size, enc);
  PutVarint(commands.size(), enc);
  enc->insert(enc->end(), commands.begin(), commands.end());
  enc->insert(enc->end(), data.begin(), data.end());
}

bool IccUnpredict(const std::vector<uint8_t>& enc, std::vector<uint8_t>* icc, std::string* why) {
  auto fail = [&](const char* m) { if (why) *why = m; return false; };
  icc->clear();
  size_t pos = 0;
  uint64_t osize, csize;
  if (!GetVarint(enc, &pos, enc.size(), &osize) || !GetVarint(enc, &pos, enc.size(), &csize)) return fail("ICC stream: truncated sizes");
  if (osize > kIccMaxEncodedSize || csize > enc.size() - pos) return fail("ICC stream: sizes out of range");
  size_t cpos = pos;
  const size_t cend = pos + (size_t)csize;
  pos = cend;   // the data stream follows the commands
  std::vector<uint8_t>& out = *icc;
  out.reserve((size_t)osize);
  // ---- header
  uint8_t h[kHeaderSize];
  InitialHeader((uint32_t)osize, h);
  for (size_t i = 0; i <= kHeaderSize; i++) {
    if (out.size() == osize) {
      if (cpos != cend || pos != enc.size()) return fail("ICC stream: data after the end of the profile");
      return true;
    }
    if (i == kHeaderSize) break;
    UpdateHeaderPrediction(out.data(), out.size(), h, i);
    if (pos >= enc.size()) return fail("ICC stream: truncated header");
    out.push_back((uint8_t)(enc[pos++] + h[i]));
  }
  if (cpos >= cend) return fail("ICC stream: no commands after the header");
  auto take = [&](size_t n, std::vector<uint8_t>* dst) {
    if (n > enc.size() - pos || out.size() + n > osize) return false;
    dst->insert(dst->end(), enc.begin() + pos, enc.begin() + pos + n);
    pos += n;
    return true;
  };
  // ---- tag table
  uint64_t ntags;
  if (!GetVarint(enc, &cpos, cend, &ntags)) return fail("ICC stream: truncated tag count");
  if (ntags != 0) {
    ntags--;
    if (ntags > (1u << 20)) return fail("ICC stream: tag count");
    PutU32((uint32_t)ntags, &out);
    // the predictor's first "previous tag" ends where a tag table WITHOUT its 4-byte count would end  [spec, recalled; unpinned]
    uint64_t prev_start = kHeaderSize + ntags * 12, prev_size = 0;
    for (;;) {
      if (out.size() > osize) return fail("ICC stream: tag table exceeds the profile");
      if (cpos == cend) break;
      const uint8_t command = enc[cpos++];
      const uint8_t code = command & 63;
      if (code == 0) break;
      char tag[4];
      if (code == 1) {
        if (4 > enc.size() - pos) return fail("ICC stream: truncated tag name");
        memcpy(tag, &enc[pos], 4);
        pos += 4;
      } else if (code == 2) memcpy(tag, "rTRC", 4);
      else if (code == 3) memcpy(tag, "rXYZ", 4);
      else if (code - 4 < 17) memcpy(tag, kTagNames[code - 4], 4);
      else return fail("ICC stream: unknown tag code");
      out.insert(out.end(), tag, tag + 4);
      uint64_t start = prev_start + prev_size, size = prev_size;
      static const char* const kXyzLike[7] = {"rXYZ", "gXYZ", "bXYZ", "kXYZ", "wtpt", "bkpt", "lumi"};
      for (auto t : kXyzLike) if (!memcmp(tag, t, 4)) size = 20;
      if (command & 64) { if (!GetVarint(enc, &cpos, cend, &start)) return fail("ICC stream: truncated tag offset"); }
      if (command & 128) { if (!GetVarint(enc, &cpos, cend, &size)) return fail("ICC stream: truncated tag size"); }
      // offsets and sizes are 32-bit fields of the profile: anything wider is a broken stream, not something to truncate
      if (start > 0xFFFFFFFFull || size > 0xFFFFFFFFull || start + 2 * size > 0xFFFFFFFFull) return fail("ICC stream: tag offset or size exceeds 32 bits");
      PutU32((uint32_t)start, &out);
      PutU32((uint32_t)size, &out);
      prev_start = start; prev_size = size;
      if (code == 2) {   // the three tone curves usually share one curve
        PutTag("gTRC", &out); PutU32((uint32_t)start, &out); PutU32((uint32_t)size, &out);
        PutTag("bTRC", &out); PutU32((uint32_t)start, &out); PutU32((uint32_t)size, &out);
      }
      if (code == 3) {   // the three colorants follow each other
        PutTag("gXYZ", &out); PutU32((uint32_t)(start + size), &out); PutU32((uint32_t)size, &out);
        PutTag("bXYZ", &out); PutU32((uint32_t)(start + 2 * size), &out); PutU32((uint32_t)size, &out);
        prev_start = start + 2 * size;
      }
    }
  }
  // ---- tag data
  for (;;) {
    if (out.size() > osize) return fail("ICC stream: data exceeds the profile");
    if (cpos == cend) break;
    const uint8_t command = enc[cpos++];
    if (command == 1) {
      uint64_t n;
      if (!GetVarint(enc, &cpos, cend, &n) || !take((size_t)n, &out)) return fail("ICC stream: insert");
    } else if (command == 2 || command == 3) {
      uint64_t n;
      std::vector<uint8_t> run;
      if (!GetVarint(enc, &cpos, cend, &n) || !take((size_t)n, &run)) return fail("ICC stream: shuffle");
      Unshuffle(&run, command == 2 ? 2 : 4);
      out.insert(out.end(), run.begin(), run.end());
    } else if (command == 4) {
      if (cpos >= cend) return fail("ICC stream: truncated predictor");
      const uint8_t flags = enc[cpos++];
      const size_t width = (flags & 3) + 1;
      const int order = (flags >> 2) & 3;
      if (width == 3 || order == 3) return fail("ICC stream: predictor parameters");
      uint64_t stride = width, n;
      if (flags & 16) { if (!GetVarint(enc, &cpos, cend, &stride) || stride < width) return fail("ICC stream: predictor stride"); }
      // (compared without multiplying: the stride is a varint of up to 63 bits)
      if (stride >= (out.size() + 3) / 4) return fail("ICC stream: predictor stride exceeds the data");
      std::vector<uint8_t> run;
      if (!GetVarint(enc, &cpos, cend, &n) || !take((size_t)n, &run)) return fail("ICC stream: predictor run");
      if (width > 1) Unshuffle(&run, width);
      const size_t start = out.size();
      for (size_t i = 0; i < run.size(); i++) out.push_back((uint8_t)(run[i] + PredictByte(out, start, i, (size_t)stride, width, order)));
    } else if (command == 10) {
      if (out.size() + 20 > osize) return fail("ICC stream: XYZ exceeds the profile");
      PutTag("XYZ ", &out);
      PutU32(0, &out);
      if (!take(12, &out)) return fail("ICC stream: XYZ");
    } else if (command >= 16 && command < 24) {
      if (out.size() + 8 > osize) return fail("ICC stream: type start exceeds the profile");
      PutTag(kTypeNames[command - 16], &out);
      PutU32(0, &out);
    } else {
      return fail("ICC stream: unknown command");
    }
  }
  if (pos != enc.size()) return fail("ICC stream: unused data");
  if (out.size() != osize) return fail("ICC stream: profile size mismatch");
  return true;
}

// ---------------------------------------------------------------------------------------------- matrix / TRC profiles
namespace {

uint32_t Be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
double S15F16(const uint8_t* p) { return (double)(int32_t)Be32(p) / 65536.0; }

bool FindTag(const uint8_t* icc, size_t size, const char* name, const uint8_t** data, size_t* len) {
  if (size < 132) return false;
  const uint32_t n = Be32(icc + 128);
  if (n > 4096 || 132 + (size_t)n * 12 > size) return false;
  for (uint32_t i = 0; i < n; i++) {
    const uint8_t* e = icc + 132 + (size_t)i * 12;
    if (memcmp(e, name, 4)) continue;
    const uint32_t off = Be32(e + 4), sz = Be32(e + 8);
    if (off > size || sz > size - off) return false;
    *data = icc + off; *len = sz;
    return true;
  }
  return false;
}

struct Curve {
  int kind = 0;          // 0 identity, 1 gamma, 2 parametric, 3 table
  int ptype = 0;
  double g = 1, a = 1, b = 0, c = 0, d = 0, e = 0, f = 0;
  std::vector<double> table;
  double Eval(double x) const {   // encoded -> linear
    x = x < 0 ? 0 : (x > 1 ? 1 : x);
    if (kind == 0) return x;
    if (kind == 1) return std::pow(x, g);
    if (kind == 3) {
      const double t = x * (table.size() - 1);
      const size_t i = (size_t)t;
      if (i + 1 >= table.size()) return table.back();
      return table[i] + (table[i + 1] - table[i]) * (t - i);
    }
    switch (ptype) {
      case 0: return std::pow(x, g);
      case 1: return x >= -b / a ? std::pow(a * x + b, g) : 0.0;
      case 2: return x >= -b / a ? std::pow(a * x + b, g) + c : c;
      case 3: return x >= d ? std::pow(a * x + b, g) : c * x;
      default: return x >= d ? std::pow(a * x + b, g) + e : c * x + f;
    }
  }
};

bool ParseCurve(const uint8_t* p, size_t n, Curve* cv) {
  if (n < 12) return false;
  if (!memcmp(p, "curv", 4)) {
    const uint32_t cnt = Be32(p + 8);
    if (cnt == 0) { cv->kind = 0; return true; }
    if (12 + (size_t)cnt * 2 > n) return false;
    if (cnt == 1) { cv->kind = 1; cv->g = ((uint32_t)p[12] << 8 | p[13]) / 256.0; return cv->g > 0; }
    cv->kind = 3;
    cv->table.resize(cnt);
    for (uint32_t i = 0; i < cnt; i++) cv->table[i] = ((uint32_t)p[12 + 2 * i] << 8 | p[13 + 2 * i]) / 65535.0;
    for (uint32_t i = 1; i < cnt; i++) if (cv->table[i] < cv->table[i - 1]) return false;   // must be invertible
    return true;
  }
  if (!memcmp(p, "para", 4)) {
    const uint32_t t = (uint32_t)p[8] << 8 | p[9];
    static const int kParams[5] = {1, 3, 4, 5, 7};
    if (t > 4 || 12 + (size_t)kParams[t] * 4 > n) return false;
    double v[7] = {1, 1, 0, 0, 0, 0, 0};
    for (int i = 0; i < kParams[t]; i++) v[i] = S15F16(p + 12 + 4 * i);
    cv->kind = 2; cv->ptype = (int)t;
    cv->g = v[0]; cv->a = v[1]; cv->b = v[2]; cv->c = v[3]; cv->d = v[4]; cv->e = v[5]; cv->f = v[6];
    if (t >= 1 && cv->a == 0) return false;
    return cv->g > 0;
  }
  return false;
}

bool Invert3(const double* m, double* o) {
  const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
  if (std::fabs(det) < 1e-12) return false;
  const double id = 1.0 / det;
  o[0] = (m[4] * m[8] - m[5] * m[7]) * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = (m[5] * m[6] - m[3] * m[8]) * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = (m[3] * m[7] - m[4] * m[6]) * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  return true;
}
void Mul3(const double* a, const double* b, double* o) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) o[r * 3 + c] = a[r * 3] * b[c] + a[r * 3 + 1] * b[3 + c] + a[r * 3 + 2] * b[6 + c];
}

}  // namespace

bool IccBuildModel(const uint8_t* icc, size_t size, IccModel* m) {
  if (size < 132 || memcmp(icc + 36, "acsp", 4) || memcmp(icc + 20, "XYZ ", 4)) return false;
  const bool gray = !memcmp(icc + 16, "GRAY", 4);
  if (!gray && memcmp(icc + 16, "RGB ", 4)) return false;
  Curve cv[3];
  const uint8_t* t;
  size_t n;
  if (gray) {
    if (!FindTag(icc, size, "kTRC", &t, &n) || !ParseCurve(t, n, &cv[0])) return false;
    cv[1] = cv[2] = cv[0];
  } else {
    const char* const kC[3] = {"rXYZ", "gXYZ", "bXYZ"};
    const char* const kT[3] = {"rTRC", "gTRC", "bTRC"};
    for (int c = 0; c < 3; c++) {
      if (!FindTag(icc, size, kC[c], &t, &n) || n < 20 || memcmp(t, "XYZ ", 4)) return false;
      for (int r = 0; r < 3; r++) m->rgb_to_xyz_d50[r * 3 + c] = S15F16(t + 8 + 4 * r);
      if (!FindTag(icc, size, kT[c], &t, &n) || !ParseCurve(t, n, &cv[c])) return false;
    }
  }
  m->gray = gray;
  for (int c = 0; c < 3; c++) {
    m->to_linear[c].resize(256);
    for (int i = 0; i < 256; i++) m->to_linear[c][i] = (float)cv[c].Eval(i / 255.0);
    // numeric inverse of the (monotonic) curve
    m->from_linear[c].resize(kIccInvLut);
    for (int i = 0; i < kIccInvLut; i++) {
      // indexed by the SQUARE ROOT of the linear value: tone curves are close to a power of ~2, so the table is nearly linear in that
      // variable and interpolation stays exact to a small fraction of an 8-bit step even next to black
      const double t = (double)i / (kIccInvLut - 1), y = t * t;
      double lo = 0, hi = 1;
      for (int it = 0; it < 40; it++) { const double mid = 0.5 * (lo + hi); if (cv[c].Eval(mid) < y) lo = mid; else hi = mid; }
      m->from_linear[c][i] = (float)(0.5 * (lo + hi));
    }
  }
  static const double kSrgbToXyzD65[9] = {0.4123907993, 0.3575843394, 0.1804807884, 0.2126390059, 0.7151686788, 0.0721923154,
                                          0.0193308187, 0.1191947798, 0.9505321522};
  static const double kBradfordD65ToD50[9] = {1.0478112, 0.0228866, -0.0501270, 0.0295424, 0.9904844, -0.0170491, -0.0092345, 0.0150436, 0.7521316};
  double id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (gray) { memcpy(m->from_linear_srgb, id, sizeof(id)); memcpy(m->to_linear_srgb, id, sizeof(id)); memcpy(m->rgb_to_xyz_d50, id, sizeof(id)); return true; }
  double srgb_to_d50[9], inv[9];
  Mul3(kBradfordD65ToD50, kSrgbToXyzD65, srgb_to_d50);
  if (!Invert3(m->rgb_to_xyz_d50, inv)) return false;
  Mul3(inv, srgb_to_d50, m->from_linear_srgb);
  return Invert3(m->from_linear_srgb, m->to_linear_srgb);
}

// ---------------------------------------------------------------------------------------------- spaces from chromaticities
namespace {
const double kBradford[9] = {0.8951, 0.2664, -0.1614, -0.7502, 1.7135, 0.0367, 0.0389, -0.0685, 1.0296};
void XyToXyz(const double xy[2], double out[3]) { out[0] = xy[0] / xy[1]; out[1] = 1.0; out[2] = (1.0 - xy[0] - xy[1]) / xy[1]; }
// adaptation of white `w` (XYZ) to D50 in the Bradford cone space
void AdaptToD50(const double w[3], double out[9]) {
  const double d50[3] = {0.96422, 1.0, 0.82521};
  double lw[3], ld[3], bi[9];
  for (int r = 0; r < 3; r++) {
    lw[r] = kBradford[r * 3] * w[0] + kBradford[r * 3 + 1] * w[1] + kBradford[r * 3 + 2] * w[2];
    ld[r] = kBradford[r * 3] * d50[0] + kBradford[r * 3 + 1] * d50[1] + kBradford[r * 3 + 2] * d50[2];
  }
  Invert3(kBradford, bi);
  double scaled[9];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) scaled[r * 3 + c] = kBradford[r * 3 + c] * (ld[r] / lw[r]);
  Mul3(bi, scaled, out);
}
void PutS15(double v, std::vector<uint8_t>* out) { PutU32((uint32_t)(int32_t)std::lrint(v * 65536.0), out); }
}  // namespace

void PrimariesToXyzD50(const double prim_xy[3][2], const double white_xy[2], double out[9]) {
  double p[9], pi[9], w[3];
  for (int c = 0; c < 3; c++) { double t[3]; XyToXyz(prim_xy[c], t); p[c] = t[0]; p[3 + c] = t[1]; p[6 + c] = t[2]; }
  XyToXyz(white_xy, w);
  Invert3(p, pi);
  double m[9];
  for (int c = 0; c < 3; c++) {
    const double sc = pi[c * 3] * w[0] + pi[c * 3 + 1] * w[1] + pi[c * 3 + 2] * w[2];
    for (int r = 0; r < 3; r++) m[r * 3 + c] = p[r * 3 + c] * sc;
  }
  double ad[9];
  AdaptToD50(w, ad);
  Mul3(ad, m, out);
}

bool MatrixFromLinearSrgb(const double prim_xy[3][2], const double white_xy[2], double out[9]) {
  static const double kSrgb[3][2] = {{0.639998686, 0.330010138}, {0.300003784, 0.600003357}, {0.150002046, 0.059997204}};
  static const double kD65[2] = {0.3127, 0.3290};
  double s[9], t[9], ti[9];
  PrimariesToXyzD50(kSrgb, kD65, s);
  PrimariesToXyzD50(prim_xy, white_xy, t);
  if (!Invert3(t, ti)) return false;
  Mul3(ti, s, out);
  return true;
}

std::vector<uint8_t> IccSynthesize(bool gray, const double prim_xy[3][2], const double white_xy[2], const IccCurveSpec& curve, uint32_t rendering_intent) {
  auto mluc = [](const char* text) {
    std::vector<uint8_t> t;
    PutTag("mluc", &t); PutU32(0, &t); PutU32(1, &t); PutU32(12, &t); PutTag("enUS", &t);
    const size_t n = strlen(text);
    PutU32((uint32_t)(2 * n), &t); PutU32(28, &t);
    for (size_t i = 0; i < n; i++) { t.push_back(0); t.push_back((uint8_t)text[i]); }
    return t;
  };
  auto xyz = [](const double v[3]) {
    std::vector<uint8_t> t;
    PutTag("XYZ ", &t); PutU32(0, &t);
    for (int i = 0; i < 3; i++) PutS15(v[i], &t);
    return t;
  };
  // tone curve as a parametric curve (encoded -> linear)
  std::vector<uint8_t> trc;
  PutTag("para", &trc); PutU32(0, &trc);
  if (curve.kind == 1 || curve.kind == 2) {
    trc.push_back(0); trc.push_back(3); trc.push_back(0); trc.push_back(0);
    const double p1[5] = {2.4, 1 / 1.055, 0.055 / 1.055, 1 / 12.92, 0.04045}, p2[5] = {1 / 0.45, 1 / 1.099, 0.099 / 1.099, 1 / 4.5, 0.081};
    for (int i = 0; i < 5; i++) PutS15(curve.kind == 1 ? p1[i] : p2[i], &trc);
  } else {
    trc.push_back(0); trc.push_back(0); trc.push_back(0); trc.push_back(0);
    PutS15(curve.kind == 0 ? 1.0 : 1.0 / curve.gamma, &trc);
  }
  double w[3], ad[9], m[9];
  XyToXyz(white_xy, w);
  AdaptToD50(w, ad);
  std::vector<uint8_t> chad;
  PutTag("sf32", &chad); PutU32(0, &chad);
  for (int i = 0; i < 9; i++) PutS15(ad[i], &chad);
  const double d50[3] = {0.9642, 1.0, 0.8249};
  std::vector<std::pair<const char*, std::vector<uint8_t>>> tags;
  tags.push_back({"desc", mluc(gray ? "JPEG XL gray (synthesised)" : "JPEG XL RGB (synthesised)")});
  tags.push_back({"cprt", mluc("CC0")});
  tags.push_back({"wtpt", xyz(d50)});
  tags.push_back({"chad", chad});
  if (gray) tags.push_back({"kTRC", trc});
  else {
    PrimariesToXyzD50(prim_xy, white_xy, m);
    const char* const kC[3] = {"rXYZ", "gXYZ", "bXYZ"};
    for (int c = 0; c < 3; c++) { const double col[3] = {m[c], m[3 + c], m[6 + c]}; tags.push_back({kC[c], xyz(col)}); }
    tags.push_back({"rTRC", trc});
  }
  const size_t nshared = gray ? 0 : 2;   // gTRC, bTRC point at rTRC
  const size_t ntags = tags.size() + nshared;
  std::vector<uint8_t> table, data;
  PutU32((uint32_t)ntags, &table);
  size_t off = 128 + 4 + 12 * ntags, trc_off = 0, trc_len = 0;
  for (auto& t : tags) {
    PutTag(t.first, &table); PutU32((uint32_t)(off + data.size()), &table); PutU32((uint32_t)t.second.size(), &table);
    if (!strcmp(t.first, "rTRC")) { trc_off = off + data.size(); trc_len = t.second.size(); }
    data.insert(data.end(), t.second.begin(), t.second.end());
    while (data.size() & 3) data.push_back(0);
  }
  if (!gray) for (const char* n : {"gTRC", "bTRC"}) { PutTag(n, &table); PutU32((uint32_t)trc_off, &table); PutU32((uint32_t)trc_len, &table); }
  std::vector<uint8_t> out;
  PutU32((uint32_t)(128 + table.size() + data.size()), &out);
  PutTag("jxl ", &out); PutU32(0x04400000, &out); PutTag("mntr", &out); PutTag(gray ? "GRAY" : "RGB ", &out); PutTag("XYZ ", &out);
  const uint8_t date[12] = {7, 234, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0};
  out.insert(out.end(), date, date + 12);
  PutTag("acsp", &out); PutTag("APPL", &out); PutU32(0, &out); PutU32(0, &out); PutU32(0, &out); PutU32(0, &out); PutU32(0, &out);
  PutU32(rendering_intent & 3, &out);
  for (int i = 0; i < 3; i++) PutS15(d50[i], &out);
  PutTag("jxl ", &out);
  out.resize(128, 0);
  out.insert(out.end(), table.begin(), table.end());
  out.insert(out.end(), data.begin(), data.end());
  return out;
}

}  // namespace jxlhip
