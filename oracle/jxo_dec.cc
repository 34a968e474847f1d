// ORACLE — test infrastructure only (see jxo_common.h header).
// Whole-frame decode: the CPU restatement of what JxlDecoderProcessInput does for the
// reference at src/JxlFileTypeIO/Decoder/JxlDecoder.cpp:252 (frame) and :454 (headers).
#include "jxo_codec.h"
#include "jxo_icc.h"
#include "jxo_entropy.h"
#include "jxo_modular.h"
#include <atomic>
#include <mutex>
#include <thread>

namespace jxo {

void ReadPermutation(BitReader& br, EntropyReader& rd, size_t skip, size_t size, std::vector<uint32_t>& perm);

void ParallelFor(int n, int num_threads, const std::function<void(int)>& fn) {
  if (num_threads <= 1 || n <= 1) {
    for (int i = 0; i < n; i++) fn(i);
    return;
  }
  std::atomic<int> next(0);
  std::mutex mu;
  std::string err;
  auto worker = [&]() {
    for (;;) {
      int i = next.fetch_add(1);
      if (i >= n) return;
      try {
        fn(i);
      } catch (const std::exception& e) {
        std::lock_guard<std::mutex> g(mu);
        if (err.empty()) err = e.what();
      }
    }
  };
  std::vector<std::thread> th;
  int nt = std::min(num_threads, n);
  for (int t = 0; t < nt; t++) th.emplace_back(worker);
  for (auto& t : th) t.join();
  if (!err.empty()) throw Error(err);
}

namespace {

struct FrameDecoder {
  const ImageMetadata& m;
  const FrameHeader& f;
  const DecodeOptions& opt;
  StageDump* dump;

  // LfGlobal
  float m_lf[3] = {1.0f / 4096, 1.0f / 512, 1.0f / 256};
  uint32_t global_scale = 1, quant_lf = 16;
  BlockCtxMap bctx;
  uint32_t color_factor = 84;
  float base_x = 0.f, base_b = 1.f;
  int ytox_lf = 0, ytob_lf = 0;
  bool has_global_tree = false;
  Tree gtree;
  EntropyCode gcode;
  ModularImage full;       // frame-level modular image
  size_t first_group_channel = 0;
  // LF
  int w8, h8, wp, hp, wt, ht;  // blocks, padded pixels, 64x64 tiles
  std::vector<int32_t> lfq[3];
  Plane lf[3];
  std::vector<uint8_t> lf_idx;  // block-context lf index per cell
  std::vector<uint8_t> strategy;
  std::vector<int32_t> raw_quant;
  std::vector<uint8_t> sharpness;
  std::vector<int8_t> ytox, ytob;
  // HfGlobal
  DequantMatrices dq;
  uint32_t num_presets = 1;
  std::vector<uint32_t> order[8][kNumOrders][3];  // per pass; empty => natural
  EntropyCode hf_code[8];
  std::vector<std::vector<int32_t>> qacc[3];   // multi-pass frames: per origin cell, the quantised block accumulated so far
  // pixels
  Plane xyb[3];
  std::vector<int32_t> qcoef[3];

  FrameDecoder(const ImageMetadata& m_, const FrameHeader& f_, const DecodeOptions& o, StageDump* d) : m(m_), f(f_), opt(o), dump(d) {
    w8 = f.xsize_blocks; h8 = f.ysize_blocks;
    wp = w8 * 8; hp = h8 * 8;
    wt = (int)DivCeil(w8, 8); ht = (int)DivCeil(h8, 8);
  }

  float InvGlobalScale() const { return 65536.0f / global_scale; }
  float MulLf(int c) const { return m_lf[c] * (InvGlobalScale() / quant_lf); }

  // ---------------------------------------------------------------- LfGlobal
  void ReadLfGlobal(BitReader& br) {
    JXO_CHECK(!(f.flags & FrameHeader::kPatches), "patches are not supported");
    JXO_CHECK(!(f.flags & FrameHeader::kSplines), "splines are not supported");
    JXO_CHECK(!(f.flags & FrameHeader::kNoise), "noise is not supported");
    if (!br.Bool()) {
      for (int c = 0; c < 3; c++) {
        m_lf[c] = br.F16() * (1.0f / 128);
        JXO_CHECK(m_lf[c] >= 1e-8f, "invalid LF dequant");
      }
    }
    if (f.encoding == 0) {
      global_scale = br.U32(BitsOff(11, 1), BitsOff(11, 2049), BitsOff(12, 4097), BitsOff(16, 8193));
      quant_lf = br.U32(Val(16), BitsOff(5, 1), BitsOff(8, 1), BitsOff(16, 1));
      bctx.Decode(br);
      if (!br.Bool()) {
        color_factor = br.U32(Val(84), Val(256), BitsOff(8, 2), BitsOff(16, 258));
        base_x = br.F16();
        base_b = br.F16();
        ytox_lf = (int)br.Read(8) - 128;
        ytob_lf = (int)br.Read(8) - 128;
      }
    }
    // GlobalModular
    has_global_tree = br.Bool();
    size_t nb_color = f.encoding == 1 ? (size_t)(m.num_color_channels()) : 0;
    if (has_global_tree) {
      size_t limit = std::min<size_t>((size_t)1 << 22, 1024 + (size_t)f.xsize * f.ysize * (m.num_color_channels() + m.ec.size()) / 16);
      DecodeTree(br, gtree, limit);
      DecodeHistograms(br, (gtree.size() + 1) / 2, gcode);
    }
    full = ModularImage();
    for (size_t c = 0; c < nb_color; c++) full.ch.emplace_back(f.xsize, f.ysize, 0, 0);
    for (size_t e = 0; e < m.ec.size(); e++) {
      JXO_CHECK(f.ec_upsampling[e] == 1 && m.ec[e].dim_shift == 0, "extra channel upsampling is not supported");
      full.ch.emplace_back(f.xsize, f.ysize, 0, 0);
    }
    GroupHeader gh;
    ModularDecode(br, full, &gh, 0, (int)f.group_dim, has_global_tree ? &gtree : nullptr, has_global_tree ? &gcode : nullptr);
    size_t c = full.nb_meta;
    for (; c < full.ch.size(); c++)
      if (full.ch[c].w > (int)f.group_dim || full.ch[c].h > (int)f.group_dim) break;
    first_group_channel = c;
    JXO_CHECK(!br.overrun, "truncated LfGlobal");
  }

  // Modular channels of the frame-level image that belong to a (LF or pass) group.
  void DecodeModularGroup(BitReader& br, int x0, int y0, int xs, int ys, int min_shift, int max_shift, uint32_t stream_id) {
    std::vector<size_t> idx;
    ModularImage sub;
    struct R { int x, y, w, h; };
    std::vector<R> rects;
    for (size_t c = first_group_channel; c < full.ch.size(); c++) {
      Channel& fc = full.ch[c];
      if (!fc.w || !fc.h) continue;
      int shift = std::min(fc.hshift, fc.vshift);
      if (shift > max_shift || shift < min_shift) continue;
      R r{x0 >> fc.hshift, y0 >> fc.vshift, xs >> fc.hshift, ys >> fc.vshift};
      r.w = std::max(0, std::min(r.w, fc.w - r.x));
      r.h = std::max(0, std::min(r.h, fc.h - r.y));
      if (r.w <= 0 || r.h <= 0) continue;
      idx.push_back(c);
      rects.push_back(r);
      sub.ch.emplace_back(r.w, r.h, fc.hshift, fc.vshift);
    }
    if (sub.ch.empty()) return;
    ModularDecode(br, sub, nullptr, stream_id, 1 << 30, has_global_tree ? &gtree : nullptr, has_global_tree ? &gcode : nullptr);
    UndoTransforms(sub);
    JXO_CHECK(sub.ch.size() == idx.size(), "group-local transforms changed the channel count");
    for (size_t k = 0; k < idx.size(); k++) {
      Channel& fc = full.ch[idx[k]];
      const R& r = rects[k];
      JXO_CHECK(sub.ch[k].w == r.w && sub.ch[k].h == r.h, "group channel size");
      for (int y = 0; y < r.h; y++) memcpy(fc.Row(r.y + y) + r.x, sub.ch[k].Row(y), sizeof(int32_t) * r.w);
    }
  }

  // ---------------------------------------------------------------- LfGroup
  uint32_t NumLfGroups() const { return f.num_lf_groups; }
  void ReadLfGroup(BitReader& br, uint32_t g) {
    const int gdim = f.group_dim;  // in blocks per LF group side
    int gx = g % f.xsize_lf_groups, gy = g / f.xsize_lf_groups;
    int bx0 = gx * gdim, by0 = gy * gdim;
    int bw = std::min(gdim, w8 - bx0), bh = std::min(gdim, h8 - by0);
    if (f.encoding == 0) {
      JXO_CHECK(!(f.flags & FrameHeader::kUseLfFrame), "LF frames are not supported");
      uint32_t extra_precision = br.Read(2);
      ModularImage img;
      for (int c = 0; c < 3; c++) img.ch.emplace_back(bw, bh, 0, 0);
      uint32_t sid = 1 + g;
      ModularDecode(br, img, nullptr, sid, 1 << 30, has_global_tree ? &gtree : nullptr, has_global_tree ? &gcode : nullptr);
      UndoTransforms(img);
      JXO_CHECK(img.ch.size() == 3, "LF image channels");
      float mul = 1.0f / (1 << extra_precision);
      float fac_x = base_x + ytox_lf * (1.0f / color_factor), fac_b = base_b + ytob_lf * (1.0f / color_factor);
      float mx = MulLf(0) * mul, my = MulLf(1) * mul, mb = MulLf(2) * mul;
      for (int y = 0; y < bh; y++) {
        const int32_t* qy = img.ch[0].Row(y);
        const int32_t* qx = img.ch[1].Row(y);
        const int32_t* qb = img.ch[2].Row(y);
        size_t o = (size_t)(by0 + y) * w8 + bx0;
        for (int x = 0; x < bw; x++) {
          lfq[0][o + x] = qx[x]; lfq[1][o + x] = qy[x]; lfq[2][o + x] = qb[x];
          float fy = qy[x] * my;
          lf[1].d[o + x] = fy;
          lf[0].d[o + x] = fy * fac_x + qx[x] * mx;
          lf[2].d[o + x] = fy * fac_b + qb[x] * mb;
          if (bctx.num_lf_ctxs > 1) {
            uint32_t ix = 0, iy = 0, ib = 0;
            for (int32_t t : bctx.lf_thresholds[0]) ix += qx[x] > t;
            for (int32_t t : bctx.lf_thresholds[1]) iy += qy[x] > t;
            for (int32_t t : bctx.lf_thresholds[2]) ib += qb[x] > t;
            // [spec, recalled; no external vector] the three bucket indices combine in the order X, B, Y
            lf_idx[o + x] = (uint8_t)((ix * (bctx.lf_thresholds[2].size() + 1) + ib) * (bctx.lf_thresholds[1].size() + 1) + iy);
          }
        }
      }
    }
    // modular channels with shift >= 3
    DecodeModularGroup(br, bx0 * 8, by0 * 8, gdim * 8, gdim * 8, 3, 1000, 1 + f.num_lf_groups + g);
    if (f.encoding == 0) {
      // HF metadata
      uint32_t nbits = CeilLog2((uint64_t)bw * bh);
      uint32_t count = br.Read(nbits) + 1;
      JXO_CHECK(count <= (uint32_t)(bw * bh), "varblock count");
      int tw = (int)DivCeil(bw, 8), th = (int)DivCeil(bh, 8);
      ModularImage img;
      img.ch.emplace_back(tw, th, 0, 0);
      img.ch.emplace_back(tw, th, 0, 0);
      img.ch.emplace_back((int)count, 2, 0, 0);
      img.ch.emplace_back(bw, bh, 0, 0);
      uint32_t sid = 1 + 2 * f.num_lf_groups + g;
      ModularDecode(br, img, nullptr, sid, 1 << 30, has_global_tree ? &gtree : nullptr, has_global_tree ? &gcode : nullptr);
      UndoTransforms(img);
      JXO_CHECK(img.ch.size() == 4, "HF metadata channels");
      int tx0 = bx0 / 8, ty0 = by0 / 8;
      for (int y = 0; y < th; y++)
        for (int x = 0; x < tw; x++) {
          int vx = img.ch[0].Row(y)[x], vb = img.ch[1].Row(y)[x];
          JXO_CHECK(vx >= -128 && vx <= 127 && vb >= -128 && vb <= 127, "CfL factor out of range");
          ytox[(size_t)(ty0 + y) * wt + tx0 + x] = (int8_t)vx;
          ytob[(size_t)(ty0 + y) * wt + tx0 + x] = (int8_t)vb;
        }
      uint32_t num = 0;
      const int32_t* row_s = img.ch[2].Row(0);
      const int32_t* row_q = img.ch[2].Row(1);
      for (int y = 0; y < bh; y++)
        for (int x = 0; x < bw; x++) {
          size_t cell = (size_t)(by0 + y) * w8 + bx0 + x;
          int sh = img.ch[3].Row(y)[x];
          JXO_CHECK(sh >= 0 && sh < 8, "EPF sharpness out of range");
          sharpness[cell] = (uint8_t)sh;
          if (strategy[cell] != 0xFF) continue;
          JXO_CHECK(num < count, "not enough varblocks");
          int s = row_s[num];
          JXO_CHECK(s >= 0 && s < kNumStrategies, "invalid AC strategy");
          int cx = kCoveredX[s], cy = kCoveredY[s];
          JXO_CHECK(x + cx <= bw && y + cy <= bh, "AC strategy overflows the LF group");
          JXO_CHECK((x % 32) + cx <= 32 && (y % 32) + cy <= 32, "AC strategy crosses a group boundary");
          int q = 1 + row_q[num];
          JXO_CHECK(q >= 1 && q <= 256, "quant field out of range");
          for (int iy = 0; iy < cy; iy++)
            for (int ix = 0; ix < cx; ix++) {
              size_t cc = cell + (size_t)iy * w8 + ix;
              JXO_CHECK(strategy[cc] == 0xFF, "overlapping varblocks");
              strategy[cc] = (uint8_t)s;
              raw_quant[cc] = q;
            }
          strategy[cell] = (uint8_t)(s | 0x80);
          num++;
        }
      JXO_CHECK(num == count, "varblock count mismatch");
    }
    JXO_CHECK(!br.overrun, "truncated LfGroup");
  }

  // ---------------------------------------------------------------- HfGlobal
  void ReadHfGlobal(BitReader& br) {
    dq.Decode(br);
    num_presets = 1 + br.Read(CeilLog2(f.num_groups));
    static const int kOrderStrategy[kNumOrders] = {DCT8, IDENTITY, DCT16X16, DCT32X32, DCT16X8, DCT32X8, DCT32X16,
                                                   DCT64X64, DCT64X32, DCT128X128, DCT128X64, DCT256X256, DCT256X128};
    for (uint32_t p = 0; p < f.num_passes; p++) {
      uint32_t used = br.U32(Val(0x5F), Val(0x13), Val(0), Bits(kNumOrders));
      if (used) {
        EntropyCode code;
        DecodeHistograms(br, 8, code);
        EntropyReader rd;
        rd.Init(code, br);
        for (int o = 0; o < kNumOrders; o++) {
          if (!(used >> o & 1)) continue;
          int s = kOrderStrategy[o];
          const std::vector<uint32_t>& nat = NaturalOrder(s);
          size_t llf = (size_t)kCoveredX[s] * kCoveredY[s];
          for (int c = 0; c < 3; c++) {
            std::vector<uint32_t> perm;
            ReadPermutation(br, rd, llf, nat.size(), perm);
            order[p][o][c].resize(nat.size());
            for (size_t k = 0; k < nat.size(); k++) order[p][o][c][k] = nat[perm[k]];
          }
        }
        JXO_CHECK(rd.CheckFinal(), "coefficient order final state");
      }
      DecodeHistograms(br, (size_t)num_presets * bctx.NumAcContexts(), hf_code[p]);
    }
    JXO_CHECK(!br.overrun, "truncated HfGlobal");
  }

  // ---------------------------------------------------------------- PassGroup
  void ReadPassGroup(BitReader& br, uint32_t g, uint32_t pass) {
    int gx = g % f.xsize_groups, gy = g / f.xsize_groups;
    if (f.encoding == 0) DecodeAcGroup(br, gx, gy, pass);
    int min_shift = 0, max_shift = 2;
    // without downsampling brackets every Modular channel of the group belongs to the last pass (earlier passes hold none)
    if (pass + 1 != f.num_passes) return;
    uint32_t sid = 1 + 3 * f.num_lf_groups + kNumQuantTables + f.num_groups * pass + g;
    DecodeModularGroup(br, gx * f.group_dim, gy * f.group_dim, f.group_dim, f.group_dim, min_shift, max_shift, sid);
    JXO_CHECK(!br.overrun, "truncated PassGroup");
  }

  void DecodeAcGroup(BitReader& br, int gx, int gy, uint32_t pass) {
    const int bx0 = gx * 32, by0 = gy * 32;
    const int bw = std::min(32, w8 - bx0), bh = std::min(32, h8 - by0);
    uint32_t preset = br.Read(CeilLog2(num_presets));
    JXO_CHECK(preset < num_presets, "histogram preset index");
    const uint32_t ctx_offset = preset * bctx.NumAcContexts();
    EntropyReader rd;
    rd.Init(hf_code[pass], br);
    uint8_t nz[3][32 * 32];
    memset(nz, 0, sizeof(nz));
    const uint32_t shift = f.pass_shift[pass];  // 0 for single pass
    std::vector<int32_t> q[3];
    std::vector<float> coef[3];
    std::vector<float> pix;
    size_t stat_coef = 0, stat_nnz = 0, stat_nonzero = 0;   // JXO_TOKEN_STATS: tokens of this section (sizing of decoder measurements)
    for (int by = 0; by < bh; by++) {
      for (int bx = 0; bx < bw; bx++) {
        size_t cell = (size_t)(by0 + by) * w8 + bx0 + bx;
        if (!(strategy[cell] & 0x80)) continue;
        int s = strategy[cell] & 0x7F;
        int cx = kCoveredX[s], cy = kCoveredY[s];
        uint32_t covered = cx * cy, log2c = CeilLog2(covered), size = covered * 64;
        uint32_t ord = kStrategyOrder[s];
        // several passes: the quantised values add up over the passes (each shifted by its pass's shift), kept per varblock
        for (int c = 0; c < 3; c++) {
          if (f.num_passes == 1) q[c].assign(size, 0);
          else {
            std::vector<int32_t>& acc = qacc[c][cell];
            if (acc.empty()) acc.assign(size, 0);
            q[c] = acc;
          }
        }
        for (int c : {1, 0, 2}) {
          uint32_t predicted;
          {
            const uint8_t* row = nz[c] + by * 32;
            if (bx == 0) predicted = by == 0 ? 32 : row[-32 + bx];
            else if (by == 0) predicted = row[bx - 1];
            else predicted = (row[-32 + bx] + row[bx - 1] + 1) / 2;
          }
          uint32_t block_ctx = bctx.Context(lf_idx.empty() ? 0 : lf_idx[cell], raw_quant[cell], ord, c);
          uint32_t nzeros = rd.Read(ctx_offset + bctx.NonZeroContext(predicted, block_ctx));
          JXO_CHECK(nzeros + covered <= size, "too many nonzero coefficients");
          stat_nnz++; stat_nonzero += nzeros;
          uint8_t fill = (uint8_t)((nzeros + covered - 1) >> log2c);
          for (int iy = 0; iy < cy; iy++)
            for (int ix = 0; ix < cx; ix++) nz[c][(by + iy) * 32 + bx + ix] = fill;
          const uint32_t histo_offset = ctx_offset + bctx.ZeroDensityContextsOffset(block_ctx);
          const std::vector<uint32_t>& custom = order[pass][ord][c];
          const uint32_t* ordp = custom.empty() ? NaturalOrder(s).data() : custom.data();
          uint32_t prev = nzeros > size / 16 ? 0 : 1;
          for (uint32_t k = covered; k < size && nzeros != 0; k++) {
            uint32_t ctx = histo_offset + ZeroDensityContext(nzeros, k, covered, log2c, prev);
            uint32_t u = rd.Read(ctx);
            stat_coef++;
            int32_t v = (int32_t)UnpackSigned(u);
            q[c][ordp[k]] += v * (1 << shift);
            prev = u != 0;
            nzeros -= prev;
          }
          JXO_CHECK(nzeros == 0, "nonzero count mismatch");
        }
        if (pass + 1 == f.num_passes) ReconstructBlock(s, bx0 + bx, by0 + by, q, coef, pix);
        else for (int c = 0; c < 3; c++) qacc[c][cell] = q[c];
        JXO_CHECK(!br.overrun, "truncated AC group");
      }
    }
    JXO_CHECK(rd.CheckFinal(), "AC group ANS final state");
    if (getenv("JXO_TOKEN_STATS"))
      fprintf(stderr, "[jxo] pass %u group (%d,%d): %zu block-channels, %zu coefficient tokens, %zu non-zero\n", pass, gx, gy, stat_nnz, stat_coef, stat_nonzero);
  }

  // Dequantise (+CfL), insert LLF, inverse transform, store pixels.
  void ReconstructBlock(int s, int bx, int by, std::vector<int32_t> q[3], std::vector<float> coef[3], std::vector<float>& pix) {
    const int cx = kCoveredX[s], cy = kCoveredY[s];
    const size_t size = (size_t)cx * cy * 64;
    const size_t cell = (size_t)by * w8 + bx;
    const int R = 8 * cy, C = 8 * cx;
    const bool special = s == IDENTITY || s == DCT2X2 || s == DCT4X4 || s == DCT4X8 || s == DCT8X4 || (s >= AFV0 && s <= AFV3);
    const bool transposed = !special && R >= C;
    const int lng = std::max(R, C);
    if (dump) {
      // footprint layout: logical (ky,kx) at (y0+ky, x0+kx); special 8x8 transforms keep their stored index
      for (int c = 0; c < 3; c++)
        for (size_t p = 0; p < size; p++) {
          int r = (int)(p / lng), cc = (int)(p % lng);
          int ky = transposed ? cc : r, kx = transposed ? r : cc;
          dump->qcoef[c][(size_t)(by * 8 + ky) * wp + bx * 8 + kx] = q[c][p];
        }
    }
    const float scale = InvGlobalScale() / raw_quant[cell];
    const float xmul = std::pow(0.8f, (float)f.x_qm_scale - 2.0f), bmul = std::pow(0.8f, (float)f.b_qm_scale - 2.0f);
    const float dq_scale[3] = {scale * xmul, scale, scale * bmul};
    const size_t tile = (size_t)(by / 8) * wt + bx / 8;
    const float cfl[3] = {base_x + ytox[tile] * (1.0f / color_factor), 0.f, base_b + ytob[tile] * (1.0f / color_factor)};
    for (int c = 0; c < 3; c++) coef[c].resize(size);
    for (int c : {1, 0, 2}) {
      const float* w = dq.Get(s, c);
      const float bias = m.quant_bias[c], bias3 = m.quant_bias[3];
      for (size_t k = 0; k < size; k++) {
        int32_t v = q[c][k];
        float a;
        if (v == 0) a = 0;
        else if (v == 1) a = bias;
        else if (v == -1) a = -bias;
        else a = (float)v - bias3 / (float)v;
        float d = a * dq_scale[c] * w[k];
        if (c != 1) d += cfl[c] * coef[1][k];
        coef[c][k] = d;
      }
    }
    pix.resize((size_t)R * C);
    for (int c = 0; c < 3; c++) {
      LlfFromLf(s, lf[c].Row(by) + bx, w8, coef[c].data());
      InverseTransform(s, coef[c].data(), pix.data(), C);
      for (int y = 0; y < R; y++) memcpy(xyb[c].Row(by * 8 + y) + bx * 8, &pix[(size_t)y * C], sizeof(float) * C);
    }
  }
};

static uint8_t ToU8(float v) {
  v = v * 255.0f;
  if (!(v > 0)) return 0;
  if (v >= 255.0f) return 255;
  return (uint8_t)(v + 0.5f);
}
static uint16_t ToU16(float v) {
  v = v * 65535.0f;
  if (!(v > 0)) return 0;
  if (v >= 65535.0f) return 65535;
  return (uint16_t)(v + 0.5f);
}
// Integer sample of `bits` bits -> output type, through [0, 1] floats like the reference's decoder library does for every channel
// whose depth differs from the output type's; equal depths pass through (clamped).
static uint32_t IntToOut(int32_t v, uint32_t bits, int bits_out) {
  const int32_t maxv = (int32_t)((1u << bits) - 1);
  if ((int)bits == bits_out) return (uint32_t)std::min(maxv, std::max(0, v));
  const float f = (float)v * (1.0f / (float)maxv);
  return bits_out == 16 ? ToU16(f) : ToU8(f);
}
// IEEE half <-> float (round to nearest even; subnormals kept), the conversions behind the Float16 representation
static float HalfToFloat(uint32_t h) {
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
static uint32_t FloatToHalf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000;
  const int32_t e = (int32_t)((u >> 23) & 0xFF) - 127 + 15;
  uint32_t mnt = u & 0x7FFFFF;
  if (((u >> 23) & 0xFF) == 0xFF) return sign | 0x7C00 | (mnt ? 0x200 : 0);
  if (e >= 31) return sign | 0x7C00;
  if (e <= 0) {
    if (e < -10) return sign;
    mnt |= 0x800000;
    const int shift = 14 - e;
    uint32_t h = mnt >> shift;
    const uint32_t rem = mnt & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1))) h++;
    return sign | h;
  }
  uint32_t h = (uint32_t)e << 10 | mnt >> 13;
  const uint32_t rem = mnt & 0x1FFF;
  if (rem > 0x1000 || (rem == 0x1000 && (h & 1))) h++;
  return sign | h;
}
// Sample of a float-coded Modular channel: the integer is the bit pattern of the float type (binary32 / binary16 only).
static float BitsToFloat(int32_t v, uint32_t bits, uint32_t exp_bits) {
  if (bits == 32 && exp_bits == 8) { float f; memcpy(&f, &v, 4); return f; }
  JXO_CHECK(bits == 16 && exp_bits == 5, "float samples other than binary32 / binary16 are not supported yet");
  return HalfToFloat((uint32_t)v & 0xFFFF);
}
// One channel sample (integer of `bits` bits, or float bit pattern when exp_bits > 0) -> raw bits of the output sample type
static uint32_t SampleToOut(int32_t v, uint32_t bits, uint32_t exp_bits, int bits_out, bool out_float) {
  if (!out_float && !exp_bits) return IntToOut(v, bits, bits_out);
  const float f = exp_bits ? BitsToFloat(v, bits, exp_bits) : (float)v * (1.0f / (float)((1u << bits) - 1));
  if (!out_float) return bits_out == 16 ? ToU16(f) : ToU8(f);
  if (bits_out == 16) return FloatToHalf(f);
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
static uint32_t FloatToOut(float v, int bits_out, bool out_float) {
  if (!out_float) return bits_out == 16 ? ToU16(v) : ToU8(v);
  if (bits_out == 16) return FloatToHalf(v);
  uint32_t u;
  memcpy(&u, &v, 4);
  return u;
}

}  // namespace

void DecodeJxl(const uint8_t* data, size_t size, const DecodeOptions& opt, DecodeResult& out) {
  out = DecodeResult();
  ParseContainer(data, size, out.boxes);
  const std::vector<uint8_t>& cs = out.boxes.codestream;
  JXO_CHECK(cs.size() >= 2 && cs[0] == 0xFF && cs[1] == 0x0A, "invalid codestream signature");
  BitReader br(cs.data(), cs.size(), 16);
  ImageMetadata& m = out.meta;
  ReadSizeHeader(br, &m.xsize, &m.ysize);
  ReadImageMetadata(br, m);
  if (m.color.want_icc) {   // the embedded profile
    const uint64_t n = br.U64();
    JXO_CHECK(n > 0 && n < (1u << 28), "ICC profile: encoded size");
    EntropyCode code;
    DecodeHistograms(br, kNumIccContexts, code);
    EntropyReader rd;
    rd.Init(code, br);
    std::vector<uint8_t> enc((size_t)n);
    for (size_t i = 0; i < enc.size(); i++) {
      const uint32_t v = rd.Read(IccByteContext(i, i > 0 ? enc[i - 1] : 0, i > 1 ? enc[i - 2] : 0));
      JXO_CHECK(v < 256, "ICC profile: byte range");
      enc[i] = (uint8_t)v;
    }
    JXO_CHECK(rd.CheckFinal(), "ICC profile: final ANS state");
    out.icc = IccFromStream(enc);
    // An XYB stream with a profile would need that profile evaluated (colour management); original-profile streams carry their
    // samples in the profile's space untouched.
    JXO_CHECK(!m.xyb_encoded, "XYB streams with an ICC profile need colour management, which the oracle does not have");
  }
  JXO_CHECK(m.exp_bits ? ((m.bits == 32 && m.exp_bits == 8) || (m.bits == 16 && m.exp_bits == 5)) : (m.bits >= 1 && m.bits <= 16),
            "only integer samples of up to 16 bits and binary16 / binary32 float samples are supported yet");
  br.AlignByte();
  FrameHeader& f = out.frame;
  ReadFrameHeader(br, m, f);
  JXO_CHECK(f.frame_type == FrameHeader::kRegular || f.frame_type == FrameHeader::kSkipProgressive,
            "first frame is not a regular frame (LF / reference frames are not supported)");
  JXO_CHECK(!f.have_crop && f.upsampling == 1, "cropped / upsampled frames are not supported");
  JXO_CHECK(!f.do_ycbcr, "YCbCr frames are not supported");
  JXO_CHECK(f.num_passes <= 8, "too many passes");
  if (f.encoding == 1) JXO_CHECK(f.num_passes == 1, "multi-pass Modular frames are not supported yet");
  if (f.encoding == 0) JXO_CHECK(m.xyb_encoded, "VarDCT without XYB is not supported");
  if (f.encoding == 1) JXO_CHECK(!m.xyb_encoded, "Modular XYB frames are not supported yet");
  Toc toc;
  size_t nsec = f.NumTocEntries();
  ReadToc(br, nsec, toc);
  JXO_CHECK((br.pos & 7) == 0, "TOC alignment");
  const size_t base = br.pos / 8;
  auto section = [&](size_t i) {
    JXO_CHECK(base + toc.offsets[i] + toc.logical_size[i] <= cs.size(), "section out of range");
    return BitReader(cs.data() + base + toc.offsets[i], toc.logical_size[i]);
  };

  StageDump* dump = opt.want_dump ? &out.dump : nullptr;
  FrameDecoder d(m, f, opt, dump);
  const int w = f.xsize, h = f.ysize;
  const size_t ncell = (size_t)d.w8 * d.h8;
  if (f.encoding == 0) {
    for (int c = 0; c < 3; c++) {
      d.lfq[c].assign(ncell, 0);
      d.lf[c] = Plane(d.w8, d.h8);
      d.xyb[c] = Plane(d.wp, d.hp);
    }
    d.strategy.assign(ncell, 0xFF);
    d.raw_quant.assign(ncell, 0);
    d.sharpness.assign(ncell, 0);
    d.ytox.assign((size_t)d.wt * d.ht, 0);
    d.ytob.assign((size_t)d.wt * d.ht, 0);
    if (dump) {
      dump->w8 = d.w8; dump->h8 = d.h8; dump->wp = d.wp; dump->hp = d.hp;
      for (int c = 0; c < 3; c++) dump->qcoef[c].assign((size_t)d.wp * d.hp, 0);
    }
  }

  if (nsec == 1) {
    BitReader sr = section(0);
    d.ReadLfGlobal(sr);
    if (d.bctx.num_lf_ctxs > 1) d.lf_idx.assign(ncell, 0);
    d.ReadLfGroup(sr, 0);
    if (f.encoding == 0) {
      if (!(f.flags & FrameHeader::kSkipAdaptiveLfSmoothing)) {
        float fac[3] = {d.MulLf(0), d.MulLf(1), d.MulLf(2)};
        if (dump) for (int c = 0; c < 3; c++) dump->lf_quant[c] = d.lfq[c];
        AdaptiveLfSmoothing(d.lf, fac);
      }
      d.ReadHfGlobal(sr);
    }
    d.ReadPassGroup(sr, 0, 0);
  } else {
    {
      BitReader sr = section(0);
      d.ReadLfGlobal(sr);
    }
    if (d.bctx.num_lf_ctxs > 1) d.lf_idx.assign(ncell, 0);
    ParallelFor((int)f.num_lf_groups, opt.num_threads, [&](int g) {
      BitReader sr = section(1 + g);
      d.ReadLfGroup(sr, g);
    });
    if (f.encoding == 0) {
      if (!(f.flags & FrameHeader::kSkipAdaptiveLfSmoothing)) {
        float fac[3] = {d.MulLf(0), d.MulLf(1), d.MulLf(2)};
        AdaptiveLfSmoothing(d.lf, fac);
      }
      BitReader sr = section(1 + f.num_lf_groups);
      d.ReadHfGlobal(sr);
    }
    if (f.num_passes > 1) for (int c = 0; c < 3; c++) d.qacc[c].assign(ncell, std::vector<int32_t>());
    for (uint32_t pass = 0; pass < f.num_passes; pass++)
      ParallelFor((int)f.num_groups, opt.num_threads, [&](int g) {
        BitReader sr = section(2 + f.num_lf_groups + pass * f.num_groups + g);
        d.ReadPassGroup(sr, g, pass);
      });
  }
  UndoTransforms(d.full);

  // ------------------------------------------------------------- colour pipeline
  const int ncolor = m.num_color_channels();
  const int alpha_ec = m.alpha_index();
  int black_ec = -1;
  for (size_t i = 0; i < m.ec.size(); i++) if (m.ec[i].type == 4 && black_ec < 0) black_ec = (int)i;
  out.cmyk = black_ec >= 0;   // Decoder/JxlDecoder.cpp:110-157: one black channel + at most one alpha channel
  if (out.cmyk) JXO_CHECK(ncolor == 3 && f.encoding == 1 && !m.exp_bits && m.bits == 8 && m.ec[black_ec].bits == 8, "CMYK: 8-bit lossless streams only");
  const int nch = ncolor + (out.cmyk ? 1 : 0) + (alpha_ec >= 0 ? 1 : 0);
  out.num_channels = nch;
  // output sample type by the colour channels' depth (Decoder/JxlDecoder.cpp:510-556 of the reference)
  out.out_float = m.exp_bits != 0;
  out.bits_out = out.out_float ? (m.bits <= 16 ? 16 : 32) : (m.bits > 8 ? 16 : 8);
  const int bpo = out.bits_out / 8;
  out.pixels.assign((size_t)w * h * nch * bpo, 0);
  auto put = [&](int y, int x, int c, uint32_t v) {
    const size_t i = ((size_t)y * w + x) * nch + c;
    for (int k = 0; k < bpo; k++) out.pixels[(size_t)bpo * i + k] = (uint8_t)(v >> (8 * k));
  };
  // Premultiplied (associated) alpha: the reference asks its library for un-premultiplied output (Decoder/JxlDecoder.cpp:233).  The library
  // divides the ENCODED colour samples (after the transfer function, as floats) by max(alpha, 2^-26) where it writes the output
  // samples  [spec, recalled; no external vector: parity unpinned].  unpremul(y, x): the multiplier, 1 when alpha is not associated.
  const bool unpremultiply = alpha_ec >= 0 && m.ec[alpha_ec].alpha_associated;
  const Channel* alpha_ch = nullptr;
  if (unpremultiply) {
    size_t ci = (f.encoding == 1 ? ncolor : 0) + alpha_ec;
    JXO_CHECK(ci < d.full.ch.size(), "alpha channel missing");
    alpha_ch = &d.full.ch[ci];
    JXO_CHECK(alpha_ch->w == w && alpha_ch->h == h, "alpha channel size");
  }
  auto unpremul = [&](int y, int x) -> float {
    const ExtraChannelInfo& ae = m.ec[alpha_ec];
    const int32_t av = alpha_ch->Row(y)[x];
    const float a = ae.exp_bits ? BitsToFloat(av, ae.bits, ae.exp_bits) : (float)av * (1.0f / (float)((1u << ae.bits) - 1));
    return 1.0f / std::max(1.0f / (float)(1u << 26), a);
  };
  if (f.encoding == 0) {
    Plane img[3];
    for (int c = 0; c < 3; c++) {
      img[c] = Plane(w, h);
      for (int y = 0; y < h; y++) memcpy(img[c].Row(y), d.xyb[c].Row(y), sizeof(float) * w);
    }
    if (dump) {
      for (int c = 0; c < 3; c++) {
        dump->lf_quant[c] = d.lfq[c];
        dump->lf[c] = d.lf[c].d;
        dump->xyb_idct[c] = img[c].d;
      }
      dump->strategy = d.strategy;
      dump->raw_quant = d.raw_quant;
      dump->sharpness = d.sharpness;
      dump->ytox = d.ytox;
      dump->ytob = d.ytob;
    }
    if (f.lf.gab) Gaborish(img, f.lf);
    if (f.lf.epf_iters > 0) {
      Plane inv_sigma(d.w8, d.h8);
      const float kInvSigmaNum = -1.1715728752538099024f;
      const float quant_scale = d.global_scale / 65536.0f;
      for (size_t i = 0; i < ncell; i++) {
        float sigma_quant = f.lf.epf_quant_mul / (quant_scale * d.raw_quant[i] * kInvSigmaNum);
        float sigma = sigma_quant * f.lf.epf_sharp_lut[d.sharpness[i]];
        sigma = std::min(-1e-4f, sigma);
        inv_sigma.d[i] = 1.0f / sigma;
      }
      Epf(img, f.lf, inv_sigma);
    }
    if (dump) for (int c = 0; c < 3; c++) dump->xyb_filtered[c] = img[c].d;
    XybToLinear(m, img);
    const int tfk = TransferKind(m.color);
    JXO_CHECK(tfk >= 0, "only linear / sRGB / BT.709 / PQ / power-law transfer functions are supported");
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        if (ncolor == 3) {
          for (int c = 0; c < 3; c++) {
            float v = img[c].Row(y)[x];
            v = EncodeTransfer(tfk, v, m.intensity_target, PowerLawGamma(m.color));
            if (unpremultiply) v *= unpremul(y, x);
            put(y, x, c, FloatToOut(v, out.bits_out, out.out_float));
          }
        } else {
          float v = img[1].Row(y)[x];
          v = EncodeTransfer(tfk, v, m.intensity_target, PowerLawGamma(m.color));
          if (unpremultiply) v *= unpremul(y, x);
          put(y, x, 0, FloatToOut(v, out.bits_out, out.out_float));
        }
      }
  } else {
    JXO_CHECK((int)d.full.ch.size() >= ncolor, "modular colour channels");
    for (int c = 0; c < ncolor; c++) {
      const Channel& ch = d.full.ch[c];
      JXO_CHECK(ch.w == w && ch.h == h, "modular colour channel size");
      for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
          uint32_t v;
          if (unpremultiply) {
            const int32_t sv = ch.Row(y)[x];
            const float fv = m.exp_bits ? BitsToFloat(sv, m.bits, m.exp_bits) : (float)sv * (1.0f / (float)((1u << m.bits) - 1));
            v = FloatToOut(fv * unpremul(y, x), out.bits_out, out.out_float);
          } else {
            v = SampleToOut(ch.Row(y)[x], m.bits, m.exp_bits, out.bits_out, out.out_float);
          }
          if (out.cmyk) v = 255 - v;   // stored 0 = full ink; the host wants 0 = no ink (Decoder/JxlDecoder.cpp:199-202)
          put(y, x, c, v);
        }
    }
    if (out.cmyk) {
      const Channel& ch = d.full.ch[ncolor + black_ec];
      JXO_CHECK(ch.w == w && ch.h == h, "black channel size");
      for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) put(y, x, 3, 255 - SampleToOut(ch.Row(y)[x], 8, 0, 8, false));
    }
  }
  if (alpha_ec >= 0) {
    size_t ci = (f.encoding == 1 ? ncolor : 0) + alpha_ec;
    JXO_CHECK(ci < d.full.ch.size(), "alpha channel missing");
    const Channel& ch = d.full.ch[ci];
    JXO_CHECK(ch.w == w && ch.h == h, "alpha channel size");
    const ExtraChannelInfo& ae = m.ec[alpha_ec];
    JXO_CHECK(ae.exp_bits ? ((ae.bits == 32 && ae.exp_bits == 8) || (ae.bits == 16 && ae.exp_bits == 5)) : (ae.bits >= 1 && ae.bits <= 16),
              "only integer alpha of up to 16 bits and binary16 / binary32 float alpha are supported yet");
    if (dump) dump->alpha = ch.d;
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) put(y, x, nch - 1, SampleToOut(ch.Row(y)[x], ae.bits, ae.exp_bits, out.bits_out, out.out_float));
  }
  // ---- orientation: the decoder library behind the reference hands out the image as it is meant to be displayed
  // (keep_orientation is off by default), sides swapped for orientations 5..8
  out.out_w = w; out.out_h = h;
  if (m.orientation != 1) {
    const int o = (int)m.orientation;
    const int ow = o >= 5 ? h : w, oh = o >= 5 ? w : h;
    const size_t pb = (size_t)nch * bpo;
    std::vector<uint8_t> t((size_t)ow * oh * pb);
    for (int oy = 0; oy < oh; oy++)
      for (int ox = 0; ox < ow; ox++) {
        int sx, sy;
        switch (o) {
          case 2: sx = w - 1 - ox; sy = oy; break;
          case 3: sx = w - 1 - ox; sy = h - 1 - oy; break;
          case 4: sx = ox; sy = h - 1 - oy; break;
          case 5: sx = oy; sy = ox; break;
          case 6: sx = oy; sy = h - 1 - ox; break;
          case 7: sx = w - 1 - oy; sy = h - 1 - ox; break;
          default: sx = w - 1 - oy; sy = ox; break;   // 8
        }
        memcpy(&t[((size_t)oy * ow + ox) * pb], &out.pixels[((size_t)sy * w + sx) * pb], pb);
      }
    out.pixels.swap(t);
    out.out_w = ow; out.out_h = oh;
  }
}

}  // namespace jxo
