// ORACLE — test infrastructure only (see jxo_common.h header).
// A simple but complete JPEG XL encoder used to produce the .jxl fixtures/benchmark inputs
// (the reference produces them through JxlEncoderAddImageFrame, Encoder/JxlEncoder.cpp:128).
// Streams are single-frame, single-pass; lossy = VarDCT+XYB with lossless Modular alpha,
// lossless = Modular RGB(A) with RCT.  Not a port of libjxl's heuristics.
#include "jxo_codec.h"
#include "jxo_icc.h"
#include "jxo_entropy.h"
#include "jxo_modular.h"

namespace jxo {
namespace {

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
  uint32_t Next() {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (uint32_t)(s >> 32);
  }
};

static inline int Mirror(int v, int n) {
  while (v < 0 || v >= n) v = v < 0 ? -v - 1 : 2 * n - 1 - v;
  return v;
}

// ------------------------------------------------------------------ global MA tree
struct TreeBuilder {
  Tree t;
  int Leaf(int predictor) {
    TreeNode n;
    n.property = -1; n.predictor = predictor; n.offset = 0; n.multiplier = 1;
    t.push_back(n);
    return (int)t.size() - 1;
  }
  int Split(int prop, int32_t val, int gt, int le) {
    TreeNode n;
    n.property = prop; n.splitval = val; n.lchild = gt; n.rchild = le;
    t.push_back(n);
    return (int)t.size() - 1;
  }
};

Tree MakeVarDctTree(uint32_t nlf) {
  TreeBuilder b;
  // alpha: one gradient-predicted context (extra local-gradient contexts bought < 0.01 % on the synthetic masks and
  // cost a tree walk per sample in every decoder)
  int alpha = b.Leaf(5);
  // LF coefficients
  int lf_b = b.Leaf(5), lf_x = b.Leaf(5), lf_y = b.Leaf(5);
  int lf1 = b.Split(0, 0, lf_x, lf_y);
  int lfn = b.Split(0, 1, lf_b, lf1);
  // HF metadata
  // constant maps (EPF sharpness 4, chroma-from-luma 0): Zero predictor + leaf offset => every residual is 0, the
  // cluster has a single symbol and a decoder can fill the channel without touching the stream
  int sharp = b.Leaf(0), qrow = b.Leaf(1), srow = b.Leaf(0), cfl = b.Leaf(0);
  b.t[sharp].offset = 4;
  int binfo = b.Split(2, 0, qrow, srow);
  int m1 = b.Split(0, 1, binfo, cfl);
  int meta = b.Split(0, 2, sharp, m1);
  int a_global = b.Leaf(5);                       // alpha coded in the GlobalModular section (stream 0)
  int n2 = b.Split(1, 0, lfn, a_global);
  int n1 = b.Split(1, (int32_t)(2 * nlf), meta, n2);
  int root = b.Split(1, (int32_t)(3 * nlf + kNumQuantTables), alpha, n1);
  return MakeBfsTree(b.t, root);
}

Tree MakeLosslessTree(int predictor, int mode) {
  TreeBuilder b;
  if (mode == 1) {
    // no weighted predictor anywhere: horizontal gradient W-NW (property 10) split three ways, then the vertical one NW-N (11)
    auto vert = [&]() {
      int hi = b.Leaf(predictor), mid = b.Leaf(predictor), lo = b.Leaf(predictor);
      int inner = b.Split(11, -4, mid, lo);
      return b.Split(11, 3, hi, inner);
    };
    int hi = vert(), mid = vert(), lo = vert();
    int inner = b.Split(10, -6, mid, lo);
    int root = b.Split(10, 5, hi, inner);
    return MakeBfsTree(b.t, root);
  }
  // contexts from the weighted predictor's max error (property 15), symmetric buckets
  static const int32_t cuts[] = {-80, -24, -8, -3, -1, 0, 2, 7, 23, 79};
  const int ncut = sizeof(cuts) / sizeof(cuts[0]);
  // build a right-leaning chain: prop15 > cuts[i] ? ... (balanced enough for 11 leaves)
  std::function<int(int, int)> build = [&](int lo, int hi) -> int {  // leaves for cut range [lo,hi)
    if (lo == hi) return b.Leaf(predictor);
    int mid = (lo + hi) / 2;
    int gt = build(mid + 1, hi);
    int le = build(lo, mid);
    return b.Split(15, cuts[mid], gt, le);
  };
  int root = build(0, ncut);
  return MakeBfsTree(b.t, root);
}

// ------------------------------------------------------------------ section assembly
struct SectionWriter {
  std::vector<BitWriter> sec;
};

std::vector<uint8_t> AssembleFrame(const ImageMetadata& m, const FrameHeader& f, std::vector<BitWriter>& sections) {
  BitWriter bw;
  WriteFrameHeader(bw, m, f);
  std::vector<uint32_t> sizes;
  std::vector<std::vector<uint8_t>> bytes;
  if (f.NumTocEntries() == 1) {
    BitWriter all;
    for (auto& s : sections) all.Append(s);
    bytes.push_back(all.Finish());
  } else {
    for (auto& s : sections) bytes.push_back(s.Finish());
  }
  for (auto& b : bytes) sizes.push_back((uint32_t)b.size());
  JXO_CHECK(sizes.size() == f.NumTocEntries(), "section count");
  WriteToc(bw, sizes);
  for (auto& b : bytes) bw.AppendBytes(b.data(), b.size());
  return bw.Finish();
}

// ------------------------------------------------------------------ VarDCT
static float HalfBitsToFloat(uint32_t h) {
  const uint32_t sign = (h >> 15) & 1, e = (h >> 10) & 31, mnt = h & 1023;
  uint32_t u;
  if (e == 0) {
    if (mnt == 0) u = sign << 31;
    else {
      int shift = 0;
      uint32_t mm = mnt;
      while (!(mm & 1024)) { mm <<= 1; shift++; }
      u = sign << 31 | (uint32_t)(127 - 15 + 1 - shift) << 23 | (mm & 1023) << 13;
    }
  } else if (e == 31) u = sign << 31 | 0xFFu << 23 | mnt << 13;
  else u = sign << 31 | (e - 15 + 127) << 23 | mnt << 13;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

struct VarDctEncoder {
  const EncodeParams& p;
  ImageMetadata m;
  FrameHeader f;
  int w, h, w8, h8, wp, hp, wt, ht;
  Plane xyb[3];              // padded
  std::vector<uint8_t> strategy;  // per cell, 0x80 = first
  std::vector<int32_t> raw_quant;
  uint32_t global_scale, quant_lf;
  DequantMatrices dq;
  BlockCtxMap bctx;
  std::vector<int32_t> lfq[3];
  std::vector<int32_t> alpha;
  bool has_alpha;

  VarDctEncoder(const EncodeParams& p_) : p(p_) {}

  void ChooseStrategies() {
    strategy.assign((size_t)w8 * h8, 0xFF);
    // per-cell activity from Y
    std::vector<float> act((size_t)w8 * h8);
    for (int by = 0; by < h8; by++)
      for (int bx = 0; bx < w8; bx++) {
        double s = 0, s2 = 0;
        for (int y = 0; y < 8; y++)
          for (int x = 0; x < 8; x++) {
            float v = xyb[1].Row(by * 8 + y)[bx * 8 + x];
            s += v; s2 += (double)v * v;
          }
        double var = s2 / 64 - (s / 64) * (s / 64);
        act[(size_t)by * w8 + bx] = (float)std::sqrt(std::max(0.0, var));
      }
    auto fits = [&](int bx, int by, int s) {
      int cx = kCoveredX[s], cy = kCoveredY[s];
      if (bx % cx || by % cy) return false;
      if (bx + cx > w8 || by + cy > h8) return false;
      if ((bx % 32) + cx > 32 || (by % 32) + cy > 32) return false;
      for (int iy = 0; iy < cy; iy++)
        for (int ix = 0; ix < cx; ix++)
          if (strategy[(size_t)(by + iy) * w8 + bx + ix] != 0xFF) return false;
      return true;
    };
    auto place = [&](int bx, int by, int s) {
      int cx = kCoveredX[s], cy = kCoveredY[s];
      for (int iy = 0; iy < cy; iy++)
        for (int ix = 0; ix < cx; ix++) strategy[(size_t)(by + iy) * w8 + bx + ix] = (uint8_t)s;
      strategy[(size_t)by * w8 + bx] = (uint8_t)(s | 0x80);
    };
    auto max_act = [&](int bx, int by, int cx, int cy) {
      float mx = 0;
      for (int iy = 0; iy < cy; iy++)
        for (int ix = 0; ix < cx; ix++) mx = std::max(mx, act[(size_t)(by + iy) * w8 + bx + ix]);
      return mx;
    };
    static const int kAll[] = {DCT8, IDENTITY, DCT2X2, DCT4X4, DCT16X16, DCT32X32, DCT16X8, DCT8X16, DCT32X8, DCT8X32, DCT32X16,
                               DCT16X32, DCT4X8, DCT8X4, DCT64X64, DCT64X32, DCT32X64, DCT128X128, DCT128X64, DCT64X128,
                               DCT256X256, DCT256X128, DCT128X256};
    const int nall = sizeof(kAll) / sizeof(kAll[0]);
    Rng rng(p.seed);
    for (int by = 0; by < h8; by++)
      for (int bx = 0; bx < w8; bx++) {
        if (strategy[(size_t)by * w8 + bx] != 0xFF) continue;
        int s = DCT8;
        if (p.strategy_mode == 1) {
          s = DCT8;
        } else if (p.strategy_mode == 3) {
          s = fits(bx, by, p.fixed_strategy) ? p.fixed_strategy : DCT8;
        } else if (p.strategy_mode == 2) {
          // bias towards small transforms so large ones do not eat the whole frame
          uint32_t r = rng.Next();
          int cand = kAll[r % nall];
          int area = kCoveredX[cand] * kCoveredY[cand];
          if (area >= 256 && (rng.Next() % 4)) cand = kAll[rng.Next() % 14];
          s = fits(bx, by, cand) ? cand : DCT8;
        } else if (p.strategy_mode == 4) {
          // the square transforms only (what the product's encoder chooses between at its default effort)
          static const struct { int s; float t; } kSquares[] = {{DCT32X32, 0.007f}, {DCT16X16, 0.016f}};
          for (auto& t : kSquares)
            if (fits(bx, by, t.s) && max_act(bx, by, kCoveredX[t.s], kCoveredY[t.s]) < t.t) { s = t.s; break; }
        } else {
          static const struct { int s; float t; } kTry[] = {{DCT64X64, 0.0035f}, {DCT64X32, 0.0045f}, {DCT32X64, 0.0045f},
                                                           {DCT32X32, 0.007f},  {DCT32X16, 0.009f},  {DCT16X32, 0.009f},
                                                           {DCT16X16, 0.016f},  {DCT16X8, 0.024f},   {DCT8X16, 0.024f}};
          for (auto& t : kTry)
            if (fits(bx, by, t.s) && max_act(bx, by, kCoveredX[t.s], kCoveredY[t.s]) < t.t) { s = t.s; break; }
        }
        place(bx, by, s);
      }
    // quant field: finer in flat areas
    raw_quant.assign((size_t)w8 * h8, 16);
    for (int by = 0; by < h8; by++)
      for (int bx = 0; bx < w8; bx++) {
        size_t cell = (size_t)by * w8 + bx;
        if (!(strategy[cell] & 0x80)) continue;
        int s = strategy[cell] & 0x7F, cx = kCoveredX[s], cy = kCoveredY[s];
        double a = 0;
        for (int iy = 0; iy < cy; iy++) for (int ix = 0; ix < cx; ix++) a += act[cell + (size_t)iy * w8 + ix];
        a /= cx * cy;
        double mod = 0.7 + 0.8 / (1.0 + a / 0.012);
        int q = (int)std::lrint(16.0 * mod);
        q = std::max(1, std::min(256, q));
        for (int iy = 0; iy < cy; iy++) for (int ix = 0; ix < cx; ix++) raw_quant[cell + (size_t)iy * w8 + ix] = q;
      }
  }

  std::vector<uint8_t> Encode(const uint8_t* px, int nch) {
    w = m.xsize; h = m.ysize;
    f.Derive(m);
    w8 = f.xsize_blocks; h8 = f.ysize_blocks; wp = w8 * 8; hp = h8 * 8;
    wt = (int)DivCeil(w8, 8); ht = (int)DivCeil(h8, 8);
    const int ncolor = nch >= 3 ? 3 : 1;
    has_alpha = nch == 2 || nch == 4;
    // 1. sRGB8 -> linear -> XYB
    Plane img[3] = {Plane(w, h), Plane(w, h), Plane(w, h)};
    if (has_alpha) alpha.resize((size_t)w * h);
    const int tfk = p.colour == 6 ? 1 : TransferKind(m.color);   // option 6: pixels treated as sRGB, header says HLG
    JXO_CHECK(tfk >= 0, "transfer function");
    if (m.exp_bits) {
      // float samples (binary32 arrays; binary16 arrays as their uint16 bit patterns): colour goes through the transfer function as
      // is, alpha is coded losslessly as the bit pattern of its float type
      for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
          const size_t o = ((size_t)y * w + x) * nch;
          auto F = [&](int c) -> float { return m.bits == 32 ? ((const float*)px)[o + c] : HalfBitsToFloat(((const uint16_t*)px)[o + c]); };
          for (int c = 0; c < 3; c++) img[c].Row(y)[x] = DecodeTransfer(tfk, F(ncolor == 3 ? c : 0), m.intensity_target, PowerLawGamma(m.color));
          if (has_alpha) alpha[(size_t)y * w + x] = m.bits == 32 ? ((const int32_t*)px)[o + ncolor] : (int32_t)((const uint16_t*)px)[o + ncolor];
        }
    } else {
    const uint32_t maxv = (1u << m.bits) - 1;
    std::vector<float> lut((size_t)maxv + 1);
    for (uint32_t i = 0; i <= maxv; i++) lut[i] = DecodeTransfer(tfk, (float)i / (float)maxv, m.intensity_target, PowerLawGamma(m.color));
    const uint16_t* px16 = (const uint16_t*)px;   // samples above 8 bits arrive as uint16
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        const size_t o = ((size_t)y * w + x) * nch;
        auto S = [&](int c) -> uint32_t { return std::min<uint32_t>(maxv, m.bits > 8 ? px16[o + c] : px[o + c]); };
        for (int c = 0; c < 3; c++) img[c].Row(y)[x] = lut[S(ncolor == 3 ? c : 0)];
        if (has_alpha) alpha[(size_t)y * w + x] = (int32_t)S(ncolor);
      }
    }
    {
      // linear RGB of the image's primaries, relative to the intensity target -> linear sRGB relative to 255 nits (what XYB is built on)
      double conv[9], inv[9];
      if (ncolor == 3) JXO_CHECK(MatrixFromSrgbGeneral(m.color, conv), "primaries / white point");
      else MatrixFromSrgb(1, conv);
      const double d = conv[0] * (conv[4] * conv[8] - conv[5] * conv[7]) - conv[1] * (conv[3] * conv[8] - conv[5] * conv[6]) +
                       conv[2] * (conv[3] * conv[7] - conv[4] * conv[6]);
      inv[0] = (conv[4] * conv[8] - conv[5] * conv[7]) / d; inv[1] = (conv[2] * conv[7] - conv[1] * conv[8]) / d; inv[2] = (conv[1] * conv[5] - conv[2] * conv[4]) / d;
      inv[3] = (conv[5] * conv[6] - conv[3] * conv[8]) / d; inv[4] = (conv[0] * conv[8] - conv[2] * conv[6]) / d; inv[5] = (conv[2] * conv[3] - conv[0] * conv[5]) / d;
      inv[6] = (conv[3] * conv[7] - conv[4] * conv[6]) / d; inv[7] = (conv[1] * conv[6] - conv[0] * conv[7]) / d; inv[8] = (conv[0] * conv[4] - conv[1] * conv[3]) / d;
      const double sc = m.intensity_target / 255.0;
      const bool ident = m.color.primaries == 1 && m.color.white_point == 1 && sc == 1.0;
      if (!ident)
        for (size_t i = 0; i < img[0].d.size(); i++) {
          const double r = img[0].d[i], g = img[1].d[i], b = img[2].d[i];
          img[0].d[i] = (float)((inv[0] * r + inv[1] * g + inv[2] * b) * sc);
          img[1].d[i] = (float)((inv[3] * r + inv[4] * g + inv[5] * b) * sc);
          img[2].d[i] = (float)((inv[6] * r + inv[7] * g + inv[8] * b) * sc);
        }
    }
    LinearToXyb(img);
    // 2. approximate inverse of the decoder-side Gaborish: 2*I - K
    if (f.lf.gab) {
      Plane blur[3] = {img[0], img[1], img[2]};
      Gaborish(blur, f.lf);
      for (int c = 0; c < 3; c++)
        for (size_t i = 0; i < img[c].d.size(); i++) img[c].d[i] = 2 * img[c].d[i] - blur[c].d[i];
    }
    for (int c = 0; c < 3; c++) {
      xyb[c] = Plane(wp, hp);
      for (int y = 0; y < hp; y++)
        for (int x = 0; x < wp; x++) xyb[c].Row(y)[x] = img[c].Row(std::min(y, h - 1))[std::min(x, w - 1)];
    }
    // 3. quantiser
    double qf = 0.85 / std::max(0.05f, p.distance);
    global_scale = (uint32_t)std::max<long>(1, std::min<long>(8193 + 65535, std::lrint(65536.0 * qf / 16.0)));
    double lfq_f = 1.1 / std::max(0.05f, p.distance);
    quant_lf = (uint32_t)std::max<long>(1, std::min<long>(65536, std::lrint(lfq_f * 65536.0 / global_scale)));
    BitWriter dq_bits;   // explicit tables: their HfGlobal bits are produced now, the quantiser uses the tables they describe
    if (p.custom_quant_tables) dq.SetCustomAndWrite(p.seed, dq_bits);
    else dq.SetDefault();
    bctx.SetDefault();
    ChooseStrategies();
    const float inv_gs = 65536.0f / global_scale;
    const float m_lf[3] = {1.0f / 4096, 1.0f / 512, 1.0f / 256};
    float mul_lf[3];
    for (int c = 0; c < 3; c++) mul_lf[c] = m_lf[c] * inv_gs / quant_lf;
    // 4. transforms, LF, quantised AC (kept per varblock)
    const size_t ncell = (size_t)w8 * h8;
    for (int c = 0; c < 3; c++) lfq[c].assign(ncell, 0);
    std::vector<std::vector<int32_t>> qac[3];  // per cell (first cells only): quantised block in stored layout
    for (int c = 0; c < 3; c++) qac[c].resize(ncell);
    const float xmul = std::pow(0.8f, (float)f.x_qm_scale - 2.0f), bmul = std::pow(0.8f, (float)f.b_qm_scale - 2.0f);
    const float base_b = 1.0f;
    ParallelFor(h8, p.num_threads, [&](int by) {
      std::vector<float> coef[3], lfv;
      for (int bx = 0; bx < w8; bx++) {
        size_t cell = (size_t)by * w8 + bx;
        if (!(strategy[cell] & 0x80)) continue;
        int s = strategy[cell] & 0x7F, cx = kCoveredX[s], cy = kCoveredY[s];
        size_t size = (size_t)cx * cy * 64;
        float lfs[3][1024];
        for (int c = 0; c < 3; c++) {
          coef[c].resize(size);
          ForwardTransform(s, xyb[c].Row(by * 8) + bx * 8, wp, coef[c].data());
          LfFromLlf(s, coef[c].data(), lfs[c], cx);
        }
        // LF quantisation (Y first; X and B code the residual after LF chroma-from-luma)
        for (int iy = 0; iy < cy; iy++)
          for (int ix = 0; ix < cx; ix++) {
            size_t cc = cell + (size_t)iy * w8 + ix;
            int32_t qy = (int32_t)std::lrint(lfs[1][iy * cx + ix] / mul_lf[1]);
            float fy = qy * mul_lf[1];
            lfq[1][cc] = qy;
            lfq[0][cc] = (int32_t)std::lrint(lfs[0][iy * cx + ix] / mul_lf[0]);
            lfq[2][cc] = (int32_t)std::lrint((lfs[2][iy * cx + ix] - base_b * fy) / mul_lf[2]);
          }
        // AC quantisation
        const float scale = inv_gs / raw_quant[cell];
        const float dqs[3] = {scale * xmul, scale, scale * bmul};
        const float cfl[3] = {0.f, 0.f, base_b};
        std::vector<float> yd(size);
        for (int c : {1, 0, 2}) {
          const float* wq = dq.Get(s, c);
          std::vector<int32_t>& q = qac[c][cell];
          q.assign(size, 0);
          for (size_t k = 0; k < size; k++) {
            float target = coef[c][k];
            if (c != 1) target -= cfl[c] * yd[k];
            float v = target / (dqs[c] * wq[k]);
            float a = std::fabs(v);
            int32_t qi = a < 0.6f ? 0 : (int32_t)std::lrint(v);
            q[k] = qi;
            if (c == 1) {
              float adj = qi == 0 ? 0 : (std::abs(qi) == 1 ? (qi > 0 ? m.quant_bias[1] : -m.quant_bias[1]) : qi - m.quant_bias[3] / qi);
              yd[k] = adj * dqs[1] * wq[k];
            }
          }
        }
      }
    });
    // 5. tokens
    const uint32_t nlf = f.num_lf_groups, ng = f.num_groups;
    Tree tree = MakeVarDctTree(nlf);
    WPHeader wp_default;
    const uint32_t np = (uint32_t)std::max(1, std::min(3, p.num_passes));
    f.num_passes = np;
    if (np == 2) f.pass_shift[0] = 1;
    if (np == 3) { f.pass_shift[0] = 2; f.pass_shift[1] = 1; }
    std::vector<std::vector<Token>> lf_tok(nlf), meta_tok(nlf), alpha_tok(ng), ac_tok((size_t)ng * np);
    std::vector<uint32_t> nb_blocks(nlf);
    ParallelFor((int)nlf, p.num_threads, [&](int g) {
      int gx = g % f.xsize_lf_groups, gy = g / f.xsize_lf_groups;
      int bx0 = gx * (int)f.group_dim, by0 = gy * (int)f.group_dim;
      int bw = std::min((int)f.group_dim, w8 - bx0), bh = std::min((int)f.group_dim, h8 - by0);
      ModularImage img;
      for (int c = 0; c < 3; c++) img.ch.emplace_back(bw, bh, 0, 0);
      const int chan_of[3] = {1, 0, 2};  // modular channel order Y, X, B
      for (int mc = 0; mc < 3; mc++)
        for (int y = 0; y < bh; y++)
          for (int x = 0; x < bw; x++) img.ch[mc].Row(y)[x] = lfq[chan_of[mc]][(size_t)(by0 + y) * w8 + bx0 + x];
      for (int mc = 0; mc < 3; mc++) TokenizeChannel(tree, wp_default, img, mc, 1 + g, lf_tok[g]);
      // HF metadata
      std::vector<int32_t> srow, qrow;
      for (int y = 0; y < bh; y++)
        for (int x = 0; x < bw; x++) {
          size_t cell = (size_t)(by0 + y) * w8 + bx0 + x;
          if (!(strategy[cell] & 0x80)) continue;
          int sw = strategy[cell] & 0x7F;
          if (p.mislabel_afv && sw == DCT8 && (x + 2 * y) % 3 == 0) sw = AFV0 + ((x ^ y) & 3);
          srow.push_back(sw);
          qrow.push_back(raw_quant[cell] - 1);
        }
      nb_blocks[g] = (uint32_t)srow.size();
      int tw = (int)DivCeil(bw, 8), th = (int)DivCeil(bh, 8);
      ModularImage mi;
      mi.ch.emplace_back(tw, th, 0, 0);
      mi.ch.emplace_back(tw, th, 0, 0);
      mi.ch.emplace_back((int)srow.size(), 2, 0, 0);
      mi.ch.emplace_back(bw, bh, 0, 0);
      memcpy(mi.ch[2].Row(0), srow.data(), srow.size() * 4);
      memcpy(mi.ch[2].Row(1), qrow.data(), qrow.size() * 4);
      for (auto& v : mi.ch[3].d) v = 4;  // EPF sharpness
      for (int mc = 0; mc < 4; mc++) TokenizeChannel(tree, wp_default, mi, mc, 1 + 2 * nlf + g, meta_tok[g]);
    });
    // block contexts from thresholds on the quantised LF (quartiles of Y, zero for X and B) and on the quant field
    std::vector<uint8_t> lf_idx;
    if (p.lf_contexts) {
      std::vector<int32_t> ys(lfq[1]);
      std::sort(ys.begin(), ys.end());
      std::vector<int32_t> yt = {ys[ys.size() / 4], ys[ys.size() / 2], ys[3 * ys.size() / 4]};
      yt.erase(std::unique(yt.begin(), yt.end()), yt.end());
      bctx.lf_thresholds[0] = {0};
      bctx.lf_thresholds[1] = yt;
      bctx.lf_thresholds[2] = {0};
      bctx.num_lf_ctxs = 2 * (uint32_t)(yt.size() + 1) * 2;
      bctx.qf_thresholds = {12, 20};
      const uint32_t nq = 3;
      bctx.ctx_map.assign((size_t)3 * kNumOrders * nq * bctx.num_lf_ctxs, 0);
      for (uint32_t cp = 0; cp < 3; cp++)
        for (uint32_t o = 0; o < (uint32_t)kNumOrders; o++)
          for (uint32_t q = 0; q < nq; q++)
            for (uint32_t l = 0; l < bctx.num_lf_ctxs; l++)
              bctx.ctx_map[((cp * kNumOrders + o) * nq + q) * bctx.num_lf_ctxs + l] = (uint8_t)((cp ? 8 : 0) + (l + 3 * q + (o ? 5 : 0)) % 8);
      bctx.num_ctxs = 16;
      lf_idx.assign(ncell, 0);
      for (size_t cell = 0; cell < ncell; cell++) {
        uint32_t ix = 0, iy = 0, ib = 0;
        for (int32_t t : bctx.lf_thresholds[0]) ix += lfq[0][cell] > t;
        for (int32_t t : bctx.lf_thresholds[1]) iy += lfq[1][cell] > t;
        for (int32_t t : bctx.lf_thresholds[2]) ib += lfq[2][cell] > t;
        lf_idx[cell] = (uint8_t)((ix * (bctx.lf_thresholds[2].size() + 1) + ib) * (bctx.lf_thresholds[1].size() + 1) + iy);
      }
    }
    // custom coefficient orders: per order bucket and channel, the positions after the LLF ones sorted by how often they are
    // non-zero in this frame (stable, so unused positions keep their natural order); every pass uses the same orders
    std::vector<uint32_t> custom_order[kNumOrders][3];
    uint32_t used_orders = 0;
    if (p.custom_orders) {
      std::vector<uint32_t> hits[kNumOrders][3];
      for (size_t cell = 0; cell < ncell; cell++) {
        if (!(strategy[cell] & 0x80)) continue;
        const int s = strategy[cell] & 0x7F, o = kStrategyOrder[s];
        const std::vector<uint32_t>& nat = NaturalOrder(s);
        for (int c = 0; c < 3; c++) {
          if (hits[o][c].empty()) hits[o][c].assign(nat.size(), 0);
          if (nat.size() != hits[o][c].size()) continue;   // (strategies that share a bucket share its size)
          for (size_t k = 0; k < nat.size(); k++) hits[o][c][k] += qac[c][cell][nat[k]] != 0;
        }
        used_orders |= 1u << o;
      }
      for (int o = 0; o < kNumOrders; o++) {
        if (!(used_orders >> o & 1)) continue;
        for (int c = 0; c < 3; c++) {
          const size_t n = hits[o][c].size();
          size_t llf = 0;
          for (int s = 0; s < kNumStrategies; s++) if (kStrategyOrder[s] == o) { llf = (size_t)kCoveredX[s] * kCoveredY[s]; break; }
          std::vector<uint32_t> perm(n);
          for (size_t k = 0; k < n; k++) perm[k] = (uint32_t)k;
          std::stable_sort(perm.begin() + llf, perm.end(), [&](uint32_t a, uint32_t b) { return hits[o][c][a] > hits[o][c][b]; });
          custom_order[o][c] = perm;   // scan position k reads natural position perm[k]
        }
      }
    }
    ParallelFor((int)ng, p.num_threads, [&](int g) {
      int gx = g % f.xsize_groups, gy = g / f.xsize_groups;
      // AC tokens
      const int bx0 = gx * 32, by0 = gy * 32;
      const int bw = std::min(32, w8 - bx0), bh = std::min(32, h8 - by0);
      for (uint32_t pass = 0; pass < np; pass++) {
      uint8_t nz[3][32 * 32];
      memset(nz, 0, sizeof(nz));
      std::vector<Token>& out = ac_tok[(size_t)pass * ng + g];
      // this pass's share of a quantised value: what is left after the earlier passes, shifted down by the pass's shift
      auto share = [&](int32_t v) {
        for (uint32_t k = 0; k < pass; k++) v -= (v >> f.pass_shift[k]) << f.pass_shift[k];
        return pass + 1 < np ? v >> f.pass_shift[pass] : v;
      };
      for (int by = 0; by < bh; by++)
        for (int bx = 0; bx < bw; bx++) {
          size_t cell = (size_t)(by0 + by) * w8 + bx0 + bx;
          if (!(strategy[cell] & 0x80)) continue;
          int s = strategy[cell] & 0x7F, cx = kCoveredX[s], cy = kCoveredY[s];
          uint32_t covered = cx * cy, log2c = CeilLog2(covered), size = covered * 64;
          uint32_t ord = kStrategyOrder[s];
          const uint32_t* natp = NaturalOrder(s).data();
          std::vector<uint32_t> ord_c;
          for (int c : {1, 0, 2}) {
            const uint32_t* ordp = natp;
            if (used_orders >> ord & 1) {
              ord_c.resize(size);
              for (uint32_t k = 0; k < size; k++) ord_c[k] = natp[custom_order[ord][c][k]];
              ordp = ord_c.data();
            }
            std::vector<int32_t> q = qac[c][cell];
            if (np > 1) for (auto& v : q) v = share(v);
            uint32_t nzeros = 0;
            for (uint32_t k = covered; k < size; k++) nzeros += q[ordp[k]] != 0;
            uint32_t predicted;
            const uint8_t* row = nz[c] + by * 32;
            if (bx == 0) predicted = by == 0 ? 32 : row[-32 + bx];
            else if (by == 0) predicted = row[bx - 1];
            else predicted = (row[-32 + bx] + row[bx - 1] + 1) / 2;
            uint32_t block_ctx = bctx.Context(lf_idx.empty() ? 0 : lf_idx[cell], raw_quant[cell], ord, c);
            out.emplace_back(bctx.NonZeroContext(predicted, block_ctx), nzeros);
            uint8_t fill = (uint8_t)((nzeros + covered - 1) >> log2c);
            for (int iy = 0; iy < cy; iy++)
              for (int ix = 0; ix < cx; ix++) nz[c][(by + iy) * 32 + bx + ix] = fill;
            const uint32_t histo_offset = bctx.ZeroDensityContextsOffset(block_ctx);
            uint32_t prev = nzeros > size / 16 ? 0 : 1;
            for (uint32_t k = covered; k < size && nzeros != 0; k++) {
              uint32_t ctx = histo_offset + ZeroDensityContext(nzeros, k, covered, log2c, prev);
              uint32_t u = (uint32_t)PackSigned(q[ordp[k]]);
              out.emplace_back(ctx, u);
              prev = u != 0;
              nzeros -= prev;
            }
          }
        }
      }
      // alpha tokens
      if (has_alpha) {
        int x0 = gx * f.group_dim, y0 = gy * f.group_dim;
        int gw = std::min<int>(f.group_dim, w - x0), gh = std::min<int>(f.group_dim, h - y0);
        ModularImage ai;
        ai.ch.emplace_back(gw, gh, 0, 0);
        for (int y = 0; y < gh; y++) memcpy(ai.ch[0].Row(y), &alpha[(size_t)(y0 + y) * w + x0], sizeof(int32_t) * gw);
        const bool global = w <= (int)f.group_dim && h <= (int)f.group_dim;
        uint32_t sid = global ? 0 : 1 + 3 * nlf + kNumQuantTables + g;
        TokenizeChannel(tree, wp_default, ai, 0, sid, alpha_tok[g]);
      }
    });
    // 6. write sections
    const bool single = f.NumTocEntries() == 1;
    const bool alpha_global = has_alpha && w <= (int)f.group_dim && h <= (int)f.group_dim;
    std::vector<BitWriter> sec(single ? 4 : 2 + nlf + (size_t)ng * np);
    // --- LfGlobal
    BitWriter& g0 = sec[0];
    g0.Bool(true);  // LF dequant defaults
    g0.U32(BitsOff(11, 1), BitsOff(11, 2049), BitsOff(12, 4097), BitsOff(16, 8193), global_scale);
    g0.U32(Val(16), BitsOff(5, 1), BitsOff(8, 1), BitsOff(16, 1), quant_lf);
    if (!p.lf_contexts) {
      g0.Bool(true);  // default block context map
    } else {
      g0.Bool(false);
      for (int j = 0; j < 3; j++) {
        g0.Write(4, (uint32_t)bctx.lf_thresholds[j].size());
        for (int32_t t : bctx.lf_thresholds[j]) g0.U32(Bits(4), BitsOff(8, 16), BitsOff(16, 272), BitsOff(32, 65808), (uint32_t)PackSigned(t));
      }
      g0.Write(4, (uint32_t)bctx.qf_thresholds.size());
      for (uint32_t t : bctx.qf_thresholds) g0.U32(Bits(2), BitsOff(3, 4), BitsOff(5, 12), BitsOff(8, 44), t - 1);
      EncodeContextMap(g0, bctx.ctx_map);
    }
    g0.Bool(true);  // default LF chroma-from-luma
    g0.Bool(true);  // has global tree
    WriteTree(g0, tree);
    std::vector<const std::vector<Token>*> sets;
    for (auto& t : lf_tok) sets.push_back(&t);
    for (auto& t : meta_tok) sets.push_back(&t);
    for (auto& t : alpha_tok) sets.push_back(&t);
    // LZ77 distance multipliers: the widest channel of each Modular stream (LF: the three LF planes; HF metadata: chroma-from-luma
    // maps, the two block-info rows, sharpness; alpha: the group)
    for (uint32_t g = 0; g < nlf; g++) {
      const int gx = g % f.xsize_lf_groups;
      const int bwg = std::min((int)f.group_dim, w8 - gx * (int)f.group_dim);
      SetStreamDistMult(&lf_tok[g], (uint32_t)bwg);
      SetStreamDistMult(&meta_tok[g], std::max<uint32_t>((uint32_t)bwg, nb_blocks[g]));
    }
    for (size_t g = 0; g < alpha_tok.size(); g++) {
      const int gx = alpha_global ? 0 : (int)(g % f.xsize_groups);
      SetStreamDistMult(&alpha_tok[g], alpha_global ? (uint32_t)w : (uint32_t)std::min((int)f.group_dim, w - gx * (int)f.group_dim));
    }
    EncOptions mo;
    mo.max_clusters = 32;
    EncCode mcode;
    BuildAndWriteCode(sets, (tree.size() + 1) / 2, mo, g0, mcode);
    GroupHeader gh;
    gh.use_global_tree = true;
    if (has_alpha) {
      // global modular image: header always; channel data only when it fits one group
      WriteGroupHeader(g0, gh);
      if (alpha_global) WriteTokens(alpha_tok[0], mcode, g0);
    }
    // --- LfGroups
    for (uint32_t g = 0; g < nlf; g++) {
      BitWriter& s = sec[1 + g];
      int gx = g % f.xsize_lf_groups, gy = g / f.xsize_lf_groups;
      int bw = std::min((int)f.group_dim, w8 - gx * (int)f.group_dim), bh = std::min((int)f.group_dim, h8 - gy * (int)f.group_dim);
      s.Write(2, 0);  // extra_precision
      WriteGroupHeader(s, gh);
      WriteTokens(lf_tok[g], mcode, s);
      s.Write(CeilLog2((uint64_t)bw * bh), nb_blocks[g] - 1);
      WriteGroupHeader(s, gh);
      WriteTokens(meta_tok[g], mcode, s);
    }
    // --- HfGlobal
    BitWriter& hg = sec[1 + nlf];
    if (p.custom_quant_tables) hg.Append(dq_bits);
    else hg.Bool(true);                     // default dequant matrices
    hg.Write(CeilLog2(ng), 0);              // num_hf_presets - 1
    auto write_orders = [&]() {
      hg.U32(Val(0x5F), Val(0x13), Val(0), Bits(kNumOrders), used_orders);
      if (!used_orders) return;
      std::vector<Token> tok;
      for (int o = 0; o < kNumOrders; o++) {
        if (!(used_orders >> o & 1)) continue;
        size_t llf = 0;
        for (int s = 0; s < kNumStrategies; s++) if (kStrategyOrder[s] == o) { llf = (size_t)kCoveredX[s] * kCoveredY[s]; break; }
        for (int c = 0; c < 3; c++) TokenizePermutation(custom_order[o][c], llf, tok);
      }
      EncCode code;
      EncOptions eo;
      eo.max_clusters = 8;
      BuildAndWriteCode({&tok}, 8, eo, hg, code);
      WriteTokens(tok, code, hg);
    };
    write_orders();
    std::vector<EncCode> acode(np);
    {   // token counts of the frame's four stream families (bench.py prices the GPU entropy stages per token with them)
      uint64_t n[4] = {0, 0, 0, 0};
      for (auto& t : lf_tok) n[0] += t.size();
      for (auto& t : meta_tok) n[1] += t.size();
      for (auto& t : ac_tok) n[2] += t.size();
      for (auto& t : alpha_tok) n[3] += t.size();
      if (getenv("JXO_TOKEN_STATS")) {
        size_t mx = 0, mxlf = 0;
        for (auto& t : ac_tok) mx = std::max(mx, t.size());
        for (size_t g = 0; g < lf_tok.size(); g++) mxlf = std::max(mxlf, lf_tok[g].size() + meta_tok[g].size());
        fprintf(stderr, "[jxo] tokens: largest HF stream %zu of %llu in %zu streams; largest LF-group section %zu\n", mx, (unsigned long long)n[2], ac_tok.size(), mxlf);
      }
      SetLastEncodeTokenCounts(n);
    }
    for (uint32_t pass = 0; pass < np; pass++) {
      if (pass) write_orders();   // every pass: its coefficient orders, then its code
      std::vector<const std::vector<Token>*> acsets;
      for (uint32_t g = 0; g < ng; g++) acsets.push_back(&ac_tok[(size_t)pass * ng + g]);
      EncOptions ao;
      ao.max_clusters = 96;
      BuildAndWriteCode(acsets, bctx.NumAcContexts(), ao, hg, acode[pass]);
    }
    // --- PassGroups
    for (uint32_t pass = 0; pass < np; pass++)
      for (uint32_t g = 0; g < ng; g++) {
        BitWriter& s = sec[single ? 3 : 2 + nlf + (size_t)pass * ng + g];
        WriteTokens(ac_tok[(size_t)pass * ng + g], acode[pass], s);
        if (pass + 1 == np && has_alpha && !alpha_global) {
          WriteGroupHeader(s, gh);
          WriteTokens(alpha_tok[g], mcode, s);
        }
      }
    return AssembleFrame(m, f, sec);
  }
};

// ------------------------------------------------------------------ lossless Modular
std::vector<uint8_t> EncodeLosslessFrame(const ImageMetadata& m, FrameHeader& f, const uint8_t* px, int nch, const EncodeParams& p) {
  f.encoding = 1;
  f.group_size_shift = 1;
  f.lf.gab = false;
  f.lf.epf_iters = 0;
  f.Derive(m);
  const int w = m.xsize, h = m.ysize;
  const int ncolor = nch >= 3 ? 3 : 1;
  ModularImage full;
  for (int c = 0; c < nch; c++) {
    full.ch.emplace_back(w, h, 0, 0);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        const size_t o = ((size_t)y * w + x) * nch + c;
        full.ch[c].Row(y)[x] = m.bits == 32 ? ((const int32_t*)px)[o]   // binary32: the bit pattern
                                            : (m.bits > 8 ? (int32_t)((const uint16_t*)px)[o] : (int32_t)px[o]);
      }
  }
  GroupHeader gh_global;
  gh_global.use_global_tree = true;
  bool paletted = false;
  if (p.palette && !m.exp_bits) paletted = ForwardPalette(full, 0, (uint32_t)nch, 1024);   // colour and alpha together, as one index channel
  if (!paletted && ncolor == 3 && !m.exp_bits) ForwardRCT(full, 0, 6);   // float samples are coded as bit patterns: no colour transform on those
  if (p.lossless_squeeze) {
    std::vector<SqueezeParams> sp;
    DefaultSqueezeParams(full, sp);
    ForwardSqueeze(full, sp);
  }
  gh_global.transforms = full.transforms;
  Tree tree = MakeLosslessTree(p.lossless_predictor, p.lossless_tree);
  WPHeader wph;
  const uint32_t nlf = f.num_lf_groups, ng = f.num_groups;
  const int gd = f.group_dim;
  // which channels are coded globally
  size_t first_group_channel = (size_t)full.nb_meta;   // meta channels (a palette) always travel in the global stream
  for (; first_group_channel < full.ch.size(); first_group_channel++)
    if (full.ch[first_group_channel].w > gd || full.ch[first_group_channel].h > gd) break;
  std::vector<Token> global_tok;
  {
    uint32_t dm = 0;
    for (size_t c = 0; c < first_group_channel; c++) dm = std::max<uint32_t>(dm, (uint32_t)full.ch[c].w);
    SetStreamDistMult(&global_tok, dm);
  }
  for (size_t c = 0; c < first_group_channel; c++) TokenizeChannel(tree, wph, full, (int)c, 0, global_tok);
  auto group_tokens = [&](int x0, int y0, int xs, int ys, int min_shift, int max_shift, uint32_t sid, std::vector<Token>& out) -> bool {
    ModularImage sub;
    for (size_t c = first_group_channel; c < full.ch.size(); c++) {
      const Channel& fc = full.ch[c];
      if (!fc.w || !fc.h) continue;
      int shift = std::min(fc.hshift, fc.vshift);
      if (shift > max_shift || shift < min_shift) continue;
      int rx = x0 >> fc.hshift, ry = y0 >> fc.vshift, rw = xs >> fc.hshift, rh = ys >> fc.vshift;
      rw = std::max(0, std::min(rw, fc.w - rx));
      rh = std::max(0, std::min(rh, fc.h - ry));
      if (rw <= 0 || rh <= 0) continue;
      Channel ch(rw, rh, fc.hshift, fc.vshift);
      for (int y = 0; y < rh; y++) memcpy(ch.Row(y), fc.Row(ry + y) + rx, sizeof(int32_t) * rw);
      sub.ch.push_back(ch);
    }
    if (sub.ch.empty()) return false;
    uint32_t dm = 0;
    for (auto& c : sub.ch) dm = std::max<uint32_t>(dm, (uint32_t)c.w);
    SetStreamDistMult(&out, dm);   // LZ77 distance multiplier of the stream: its widest channel
    for (size_t c = 0; c < sub.ch.size(); c++) TokenizeChannel(tree, wph, sub, (int)c, sid, out);
    return true;
  };
  std::vector<std::vector<Token>> lf_tok(nlf), ac_tok(ng);
  std::vector<uint8_t> lf_has(nlf, 0), ac_has(ng, 0);
  ParallelFor((int)nlf, p.num_threads, [&](int g) {
    int gx = g % f.xsize_lf_groups, gy = g / f.xsize_lf_groups;
    lf_has[g] = group_tokens(gx * gd * 8, gy * gd * 8, gd * 8, gd * 8, 3, 1000, 1 + nlf + g, lf_tok[g]);
  });
  ParallelFor((int)ng, p.num_threads, [&](int g) {
    int gx = g % f.xsize_groups, gy = g / f.xsize_groups;
    ac_has[g] = group_tokens(gx * gd, gy * gd, gd, gd, 0, 2, 1 + 3 * nlf + kNumQuantTables + g, ac_tok[g]);
  });
  const bool single = f.NumTocEntries() == 1;
  std::vector<BitWriter> sec(single ? 4 : 2 + nlf + ng);
  BitWriter& g0 = sec[0];
  g0.Bool(true);  // LF dequant defaults
  g0.Bool(true);  // global tree
  WriteTree(g0, tree);
  std::vector<const std::vector<Token>*> sets = {&global_tok};
  for (auto& t : lf_tok) sets.push_back(&t);
  for (auto& t : ac_tok) sets.push_back(&t);
  EncOptions mo;
  mo.max_clusters = 64;
  EncCode mcode;
  BuildAndWriteCode(sets, (tree.size() + 1) / 2, mo, g0, mcode);
  WriteGroupHeader(g0, gh_global);
  if (first_group_channel > 0) WriteTokens(global_tok, mcode, g0);
  GroupHeader gh;
  gh.use_global_tree = true;
  for (uint32_t g = 0; g < nlf; g++)
    if (lf_has[g]) { WriteGroupHeader(sec[1 + g], gh); WriteTokens(lf_tok[g], mcode, sec[1 + g]); }
  for (uint32_t g = 0; g < ng; g++)
    if (ac_has[g]) { WriteGroupHeader(sec[2 + nlf + g], gh); WriteTokens(ac_tok[g], mcode, sec[2 + nlf + g]); }
  return AssembleFrame(m, f, sec);
}

}  // namespace

static thread_local uint64_t g_token_counts[4] = {0, 0, 0, 0};
void SetLastEncodeTokenCounts(const uint64_t n[4]) { for (int i = 0; i < 4; i++) g_token_counts[i] = n[i]; }
void GetLastEncodeTokenCounts(uint64_t n[4]) { for (int i = 0; i < 4; i++) n[i] = g_token_counts[i]; }

std::vector<uint8_t> EncodeJxl(const uint8_t* px, uint32_t w, uint32_t h, int nch, const EncodeParams& p, const uint8_t* exif,
                               size_t exif_size, const uint8_t* xmp, size_t xmp_size) {
  JXO_CHECK(nch >= 1 && nch <= (p.cmyk ? 5 : 4) && w > 0 && h > 0, "bad image");
  JXO_CHECK(!p.cmyk || (p.lossless && nch >= 4 && !p.icc.empty()), "CMYK: lossless, 4 or 5 channels, with a (CMYK) ICC profile");
  ImageMetadata m;
  m.xsize = w; m.ysize = h;
  m.xyb_encoded = !p.lossless;
  // The reference always signals sRGB with perceptual intent (Encoder/JxlEncoder.cpp:269-282).
  m.color.all_default = false;
  m.color.color_space = nch >= 3 ? 0 : 1;
  m.color.white_point = 1; m.color.primaries = 1; m.color.tf = 13; m.color.rendering_intent = 0;
  switch (p.colour) {
    case 0: break;
    case 1: m.color.primaries = 11; break;
    case 2: m.color.tf = 1; break;
    case 3: m.color.primaries = 9; m.color.tf = 8; break;
    case 4: m.color.primaries = 9; m.color.tf = 16; m.intensity_target = 10000.f; break;
    case 5: m.color.tf = 8; break;
    case 6: m.color.tf = 18; break;   // HLG in the header only (test stream for decoders that must refuse it)
    case 7:   // Adobe RGB (1998): custom primaries, D65, gamma 563 / 256
      m.color.primaries = 2; m.color.have_gamma = true; m.color.gamma = 4547069;
      m.color.custom_xy[1][0] = 640000; m.color.custom_xy[1][1] = 330000; m.color.custom_xy[2][0] = 210000; m.color.custom_xy[2][1] = 710000;
      m.color.custom_xy[3][0] = 150000; m.color.custom_xy[3][1] = 60000;
      break;
    case 8: m.color.white_point = 11; m.color.primaries = 11; m.color.tf = 17; break;   // DCI-P3: DCI white, P3 primaries, gamma 2.6
    case 9: m.color.white_point = 2; m.color.custom_xy[0][0] = 345700; m.color.custom_xy[0][1] = 358500; m.color.tf = 8; break;   // linear, D50 white
    default: JXO_CHECK(false, "unknown colour option");
  }
  if (nch < 3) m.color.primaries = 1;   // gray: no primaries
  if (!p.icc.empty()) m.color.want_icc = true;
  JXO_CHECK(p.bits >= 8 && p.bits <= 16, "bits per sample must be 8..16");
  JXO_CHECK(p.float_samples == 0 || p.float_samples == 16 || p.float_samples == 32, "float samples are binary16 or binary32");
  m.bits = p.float_samples ? (uint32_t)p.float_samples : (uint32_t)p.bits;
  m.exp_bits = p.float_samples == 32 ? 8 : (p.float_samples == 16 ? 5 : 0);
  JXO_CHECK(p.orientation >= 1 && p.orientation <= 8, "orientation must be 1..8");
  m.orientation = (uint32_t)p.orientation;
  if (p.cmyk) {   // "the RGB samples are to be interpreted as CMY" + a black extra channel (+ alpha)
    m.ec.push_back(ExtraChannelInfo()); m.ec.back().type = 4; m.ec.back().bits = m.bits;
    if (nch == 5) { m.ec.push_back(ExtraChannelInfo()); m.ec.back().bits = m.bits; }
  } else if (nch == 2 || nch == 4) { m.ec.push_back(ExtraChannelInfo()); m.ec.back().bits = m.bits; m.ec.back().exp_bits = m.exp_bits; m.ec.back().alpha_associated = p.premultiplied_alpha; }
  m.have_animation = p.animation_frames > 1;
  std::vector<uint8_t> frame;
  const size_t bytes_per_px = (size_t)nch * (m.bits > 16 ? 4 : (m.bits > 8 ? 2 : 1));
  for (int k = 0; k < std::max(1, p.animation_frames); k++) {
    // later frames of an animation: the same picture upside down and mirrored
    std::vector<uint8_t> other;
    const uint8_t* src = px;
    if (k > 0) {
      other.resize((size_t)w * h * bytes_per_px);
      for (size_t i = 0; i < (size_t)w * h; i++) memcpy(&other[i * bytes_per_px], px + ((size_t)w * h - 1 - i) * bytes_per_px, bytes_per_px);
      src = other.data();
    }
    FrameHeader f;
    ClearStreamDistMults();
    f.ec_upsampling.assign(m.ec.size(), 1);
    f.is_last = k + 1 == std::max(1, p.animation_frames);
    f.duration = m.have_animation ? 10 : 0;
    std::vector<uint8_t> one;
    if (p.lossless) {
      one = EncodeLosslessFrame(m, f, src, nch, p);
    } else {
      f.encoding = 0;
      f.lf.gab = p.gaborish;
      int iters = p.epf_iters;
      if (iters < 0) {
        iters = 0;
        for (float t : {0.7f, 1.5f, 4.0f}) if (p.distance >= t) iters++;
      }
      f.lf.epf_iters = iters;
      if (!p.adaptive_lf_smoothing) f.flags |= FrameHeader::kSkipAdaptiveLfSmoothing;
      VarDctEncoder enc(p);
      enc.m = m;
      enc.f = f;
      one = enc.Encode(src, nch);
    }
    frame.insert(frame.end(), one.begin(), one.end());
    ClearStreamDistMults();
  }
  BitWriter bw;
  bw.Write(8, 0xFF);
  bw.Write(8, 0x0A);
  WriteSizeHeader(bw, w, h);
  WriteImageMetadata(bw, m);
  if (m.color.want_icc) {   // the profile follows the metadata: predicted byte stream, 41 contexts on the two bytes before
    const std::vector<uint8_t> enc = IccToStream(p.icc);
    bw.U64(enc.size());
    std::vector<Token> tok;
    for (size_t i = 0; i < enc.size(); i++) tok.emplace_back(IccByteContext(i, i > 0 ? enc[i - 1] : 0, i > 1 ? enc[i - 2] : 0), enc[i]);
    EncCode code;
    EncOptions eo;
    eo.max_clusters = 16;
    BuildAndWriteCode({&tok}, kNumIccContexts, eo, bw, code);
    WriteTokens(tok, code, bw);
  }
  bw.AlignByte();
  std::vector<uint8_t> cs = bw.Finish();
  cs.insert(cs.end(), frame.begin(), frame.end());
  if (!p.container) return cs;
  return WriteContainer(cs, exif, exif_size, xmp, xmp_size);
}

}  // namespace jxo
