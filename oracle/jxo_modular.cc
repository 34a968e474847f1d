// ORACLE — test infrastructure only (see jxo_common.h header).
#include "jxo_modular.h"
#include <map>
#include <set>
#include <deque>

namespace jxo {

// ------------------------------------------------------------------ headers
static BitReader::D kBeginC[4] = {Bits(3), BitsOff(6, 8), BitsOff(10, 72), BitsOff(13, 1096)};

void ReadGroupHeader(BitReader& br, GroupHeader& h) {
  h = GroupHeader();
  h.use_global_tree = br.Bool();
  h.wp.default_wp = br.Bool();
  if (!h.wp.default_wp) {
    h.wp.p1C = br.Read(5); h.wp.p2C = br.Read(5);
    h.wp.p3Ca = br.Read(5); h.wp.p3Cb = br.Read(5); h.wp.p3Cc = br.Read(5); h.wp.p3Cd = br.Read(5); h.wp.p3Ce = br.Read(5);
    for (int i = 0; i < 4; i++) h.wp.w[i] = br.Read(4);
  }
  uint32_t nb = br.U32(Val(0), Val(1), BitsOff(4, 2), BitsOff(8, 18));
  h.transforms.resize(nb);
  for (auto& t : h.transforms) {
    t.id = br.Read(2);
    JXO_CHECK(t.id < 3, "invalid transform id");
    if (t.id == 0) {
      t.begin_c = br.U32(kBeginC[0], kBeginC[1], kBeginC[2], kBeginC[3]);
      t.rct_type = br.U32(Val(6), Bits(2), BitsOff(4, 2), BitsOff(6, 10));
      JXO_CHECK(t.rct_type < 42, "rct_type");
    } else if (t.id == 1) {
      t.begin_c = br.U32(kBeginC[0], kBeginC[1], kBeginC[2], kBeginC[3]);
      t.num_c = br.U32(Val(1), Val(3), Val(4), BitsOff(13, 1));
      t.nb_colors = br.U32(BitsOff(8, 0), BitsOff(10, 256), BitsOff(12, 1280), BitsOff(16, 5376));
      t.nb_deltas = br.U32(Val(0), BitsOff(8, 1), BitsOff(10, 257), BitsOff(16, 1281));
      t.predictor = br.Read(4);
    } else {
      uint32_t n = br.U32(Val(0), BitsOff(4, 1), BitsOff(6, 9), BitsOff(8, 41));
      t.squeezes.resize(n);
      for (auto& s : t.squeezes) {
        s.horizontal = br.Bool();
        s.in_place = br.Bool();
        s.begin_c = br.U32(kBeginC[0], kBeginC[1], kBeginC[2], kBeginC[3]);
        s.num_c = br.U32(Val(1), Val(2), Val(3), BitsOff(4, 4));
      }
    }
  }
}

void WriteGroupHeader(BitWriter& bw, const GroupHeader& h) {
  bw.Bool(h.use_global_tree);
  bw.Bool(h.wp.default_wp);
  if (!h.wp.default_wp) {
    bw.Write(5, h.wp.p1C); bw.Write(5, h.wp.p2C);
    bw.Write(5, h.wp.p3Ca); bw.Write(5, h.wp.p3Cb); bw.Write(5, h.wp.p3Cc); bw.Write(5, h.wp.p3Cd); bw.Write(5, h.wp.p3Ce);
    for (int i = 0; i < 4; i++) bw.Write(4, h.wp.w[i]);
  }
  bw.U32(Val(0), Val(1), BitsOff(4, 2), BitsOff(8, 18), (uint32_t)h.transforms.size());
  for (auto& t : h.transforms) {
    bw.Write(2, t.id);
    if (t.id == 0) {
      bw.U32(kBeginC[0], kBeginC[1], kBeginC[2], kBeginC[3], t.begin_c);
      bw.U32(Val(6), Bits(2), BitsOff(4, 2), BitsOff(6, 10), t.rct_type);
    } else if (t.id == 1) {
      bw.U32(kBeginC[0], kBeginC[1], kBeginC[2], kBeginC[3], t.begin_c);
      bw.U32(Val(1), Val(3), Val(4), BitsOff(13, 1), t.num_c);
      bw.U32(BitsOff(8, 0), BitsOff(10, 256), BitsOff(12, 1280), BitsOff(16, 5376), t.nb_colors);
      bw.U32(Val(0), BitsOff(8, 1), BitsOff(10, 257), BitsOff(16, 1281), t.nb_deltas);
      bw.Write(4, t.predictor);
    } else {
      bw.U32(Val(0), BitsOff(4, 1), BitsOff(6, 9), BitsOff(8, 41), (uint32_t)t.squeezes.size());
      for (auto& s : t.squeezes) {
        bw.Bool(s.horizontal);
        bw.Bool(s.in_place);
        bw.U32(kBeginC[0], kBeginC[1], kBeginC[2], kBeginC[3], s.begin_c);
        bw.U32(Val(1), Val(2), Val(3), BitsOff(4, 4), s.num_c);
      }
    }
  }
}

// ------------------------------------------------------------------ MA tree
enum { kSplitValCtx = 0, kPropertyCtx = 1, kPredictorCtx = 2, kOffsetCtx = 3, kMulLogCtx = 4, kMulBitsCtx = 5, kNumTreeCtx = 6 };

void DecodeTree(BitReader& br, Tree& tree, size_t size_limit) {
  EntropyCode code;
  DecodeHistograms(br, kNumTreeCtx, code);
  EntropyReader rd;
  rd.Init(code, br);
  tree.clear();
  size_t leaf_id = 0, to_decode = 1;
  while (to_decode > 0) {
    JXO_CHECK(tree.size() < size_limit, "MA tree too large");
    to_decode--;
    uint32_t prop1 = rd.Read(kPropertyCtx);
    JXO_CHECK(prop1 <= 256, "MA tree property");
    TreeNode n;
    n.property = (int)prop1 - 1;
    if (n.property == -1) {
      n.predictor = rd.Read(kPredictorCtx);
      JXO_CHECK(n.predictor < 14, "MA tree predictor");
      n.offset = UnpackSigned(rd.Read(kOffsetCtx));
      uint32_t mul_log = rd.Read(kMulLogCtx);
      JXO_CHECK(mul_log < 31, "MA tree mul_log");
      uint32_t mul_bits = rd.Read(kMulBitsCtx);
      JXO_CHECK(mul_bits + 1 < (1u << (31 - mul_log)), "MA tree mul_bits");
      n.multiplier = (mul_bits + 1) << mul_log;
      n.leaf_id = (int)leaf_id++;
      tree.push_back(n);
      continue;
    }
    n.splitval = (int32_t)UnpackSigned(rd.Read(kSplitValCtx));
    n.lchild = (int)(tree.size() + to_decode + 1);
    n.rchild = (int)(tree.size() + to_decode + 2);
    tree.push_back(n);
    to_decode += 2;
    JXO_CHECK(!br.overrun, "truncated MA tree");
  }
  JXO_CHECK(rd.CheckFinal(), "MA tree ANS final state");
}

void TokenizeTree(const Tree& tree, std::vector<Token>& out) {
  // tree must already be in BFS layout (children indices as DecodeTree would assign).
  for (size_t i = 0; i < tree.size(); i++) {
    const TreeNode& n = tree[i];
    out.emplace_back(kPropertyCtx, (uint32_t)(n.property + 1));
    if (n.property == -1) {
      out.emplace_back(kPredictorCtx, (uint32_t)n.predictor);
      out.emplace_back(kOffsetCtx, (uint32_t)PackSigned(n.offset));
      uint32_t mul_log = __builtin_ctz(n.multiplier);
      uint32_t mul_bits = (n.multiplier >> mul_log) - 1;
      out.emplace_back(kMulLogCtx, mul_log);
      out.emplace_back(kMulBitsCtx, mul_bits);
    } else {
      out.emplace_back(kSplitValCtx, (uint32_t)PackSigned(n.splitval));
    }
  }
}

void WriteTree(BitWriter& bw, const Tree& tree) {
  std::vector<Token> toks;
  TokenizeTree(tree, toks);
  EncOptions o;
  o.max_clusters = 6;
  EncCode ec;
  std::vector<const std::vector<Token>*> sets = {&toks};
  BuildAndWriteCode(sets, kNumTreeCtx, o, bw, ec);
  WriteTokens(toks, ec, bw);
}

Tree MakeBfsTree(const Tree& linked, int root) {
  Tree out;
  std::deque<int> q;
  q.push_back(root);
  int leaf_id = 0;
  std::vector<int> order;
  // first pass: BFS order
  while (!q.empty()) {
    int i = q.front(); q.pop_front();
    order.push_back(i);
    if (linked[i].property >= 0) { q.push_back(linked[i].lchild); q.push_back(linked[i].rchild); }
  }
  // second pass: assign indices exactly like DecodeTree does
  size_t to_decode = 1;
  for (size_t k = 0; k < order.size(); k++) {
    TreeNode n = linked[order[k]];
    to_decode--;
    if (n.property >= 0) {
      n.lchild = (int)(out.size() + to_decode + 1);
      n.rchild = (int)(out.size() + to_decode + 2);
      to_decode += 2;
    } else {
      n.leaf_id = leaf_id++;
    }
    out.push_back(n);
  }
  return out;
}

// ------------------------------------------------------------------ weighted predictor
namespace {
constexpr int kPredExtraBits = 3;
constexpr int64_t kPredictionRound = ((1 << kPredExtraBits) >> 1) - 1;

struct WPState {
  WPHeader h;
  int64_t prediction[4] = {0, 0, 0, 0};
  int64_t pred = 0;
  std::vector<uint32_t> pred_errors[4];
  std::vector<int32_t> error;
  uint32_t divlookup[64];
  void Init(const WPHeader& hh, size_t xsize) {
    h = hh;
    for (int i = 0; i < 4; i++) pred_errors[i].assign((xsize + 2) * 2, 0);
    error.assign((xsize + 2) * 2, 0);
    for (int i = 0; i < 64; i++) divlookup[i] = (1u << 24) / (i + 1);
  }
  uint32_t ErrorWeight(uint64_t x, uint32_t maxweight) const {
    int shift = FloorLog2(x + 1) - 5;
    if (shift < 0) shift = 0;
    return 4 + (uint32_t)((maxweight * (uint64_t)divlookup[x >> shift]) >> shift);
  }
  int64_t Predict(size_t x, size_t y, size_t xsize, int64_t N, int64_t W, int64_t NE, int64_t NW, int64_t NN, int64_t* max_err) {
    size_t cur_row = (y & 1) ? 0 : (xsize + 2);
    size_t prev_row = (y & 1) ? (xsize + 2) : 0;
    size_t pos_N = prev_row + x;
    size_t pos_NE = x < xsize - 1 ? pos_N + 1 : pos_N;
    size_t pos_NW = x > 0 ? pos_N - 1 : pos_N;
    uint32_t weights[4];
    for (int i = 0; i < 4; i++) {
      uint64_t e = (uint64_t)pred_errors[i][pos_N] + pred_errors[i][pos_NE] + pred_errors[i][pos_NW];
      weights[i] = ErrorWeight(e, h.w[i]);
    }
    N <<= kPredExtraBits; W <<= kPredExtraBits; NE <<= kPredExtraBits; NW <<= kPredExtraBits; NN <<= kPredExtraBits;
    int64_t teW = x == 0 ? 0 : error[cur_row + x - 1];
    int64_t teN = error[pos_N];
    int64_t teNW = error[pos_NW];
    int64_t sumWN = teN + teW;
    int64_t teNE = error[pos_NE];
    if (max_err) {
      int64_t p = teW;
      if (std::llabs(teN) > std::llabs(p)) p = teN;
      if (std::llabs(teNW) > std::llabs(p)) p = teNW;
      if (std::llabs(teNE) > std::llabs(p)) p = teNE;
      *max_err = p;
    }
    prediction[0] = W + NE - N;
    prediction[1] = N - (((sumWN + teNE) * h.p1C) >> 5);
    prediction[2] = W - (((sumWN + teNW) * h.p2C) >> 5);
    prediction[3] = N - ((teNW * h.p3Ca + teN * h.p3Cb + teNE * h.p3Cc + (NN - N) * h.p3Cd + (NW - W) * h.p3Ce) >> 5);
    // weighted average
    uint32_t weight_sum = 0;
    for (int i = 0; i < 4; i++) weight_sum += weights[i];
    int log_weight = FloorLog2(weight_sum);
    weight_sum = 0;
    for (int i = 0; i < 4; i++) { weights[i] >>= log_weight - 4; weight_sum += weights[i]; }
    int64_t sum = (weight_sum >> 1) - 1;
    for (int i = 0; i < 4; i++) sum += prediction[i] * weights[i];
    pred = (sum * divlookup[weight_sum - 1]) >> 24;
    if (((teN ^ teW) | (teN ^ teNW)) > 0) return (pred + kPredictionRound) >> kPredExtraBits;
    int64_t mx = std::max(W, std::max(NE, N));
    int64_t mn = std::min(W, std::min(NE, N));
    pred = std::max(mn, std::min(mx, pred));
    return (pred + kPredictionRound) >> kPredExtraBits;
  }
  void UpdateErrors(int64_t val, size_t x, size_t y, size_t xsize) {
    size_t cur_row = (y & 1) ? 0 : (xsize + 2);
    size_t prev_row = (y & 1) ? (xsize + 2) : 0;
    val <<= kPredExtraBits;
    error[cur_row + x] = (int32_t)(pred - val);
    for (int i = 0; i < 4; i++) {
      uint32_t err = (uint32_t)((std::llabs(prediction[i] - val) + kPredictionRound) >> kPredExtraBits);
      pred_errors[i][cur_row + x] = err;
      pred_errors[i][prev_row + x + 1] += err;
    }
  }
};

static inline int64_t ClampedGradient(int64_t l, int64_t t, int64_t tl) {
  int64_t mn = std::min(l, t), mx = std::max(l, t);
  return std::max(mn, std::min(mx, l + t - tl));
}

struct TreeInfo {
  bool uses_wp = false;
  int max_prop = kNumNonrefProps - 1;
};
static TreeInfo AnalyzeTree(const Tree& tree) {
  TreeInfo ti;
  for (auto& n : tree) {
    if (n.property >= 0) {
      ti.max_prop = std::max(ti.max_prop, n.property);
      if (n.property == 15) ti.uses_wp = true;
    } else if (n.predictor == 6) ti.uses_wp = true;
  }
  return ti;
}

// Visits every pixel of channel `chan` in raster order.  `Coder` is called with
// (leaf, guess) and must return the pixel value (decoder: from the stream; encoder: the known value).
template <class Coder>
void VisitChannel(const Tree& tree, const WPHeader& wph, const std::vector<Channel>& chs, int chan, uint32_t stream_id,
                  int32_t* pixels, Coder&& coder) {
  const Channel& ch = chs[chan];
  const int w = ch.w, h = ch.h;
  TreeInfo ti = AnalyzeTree(tree);
  // eligible previous channels for properties >= 16
  std::vector<int> refs;
  for (int j = chan - 1; j >= 0 && (int)(kNumNonrefProps + 4 * refs.size()) <= ti.max_prop; j--) {
    const Channel& r = chs[j];
    if (r.w != w || r.h != h || r.hshift != ch.hshift || r.vshift != ch.vshift) continue;
    refs.push_back(j);
  }
  std::vector<int64_t> props(std::max(ti.max_prop + 1, kNumNonrefProps + 4 * (int)refs.size()), 0);
  WPState wp;
  if (ti.uses_wp) wp.Init(wph, w);
  props[0] = chan;
  props[1] = stream_id;
  for (int y = 0; y < h; y++) {
    int32_t* row = pixels + (size_t)y * w;
    const int32_t* prow = y > 0 ? row - w : nullptr;
    const int32_t* pprow = y > 1 ? row - 2 * w : nullptr;
    props[2] = y;
    props[9] = 0;
    for (int x = 0; x < w; x++) {
      int64_t left = x ? row[x - 1] : (y ? prow[x] : 0);
      int64_t top = y ? prow[x] : left;
      int64_t topleft = (x && y) ? prow[x - 1] : left;
      int64_t topright = (x + 1 < w && y) ? prow[x + 1] : top;
      int64_t leftleft = x > 1 ? row[x - 2] : left;
      int64_t toptop = y > 1 ? pprow[x] : top;
      int64_t toprightright = (x + 2 < w && y) ? prow[x + 2] : topright;
      int64_t wp_pred = 0, wp_err = 0;
      if (ti.uses_wp) wp_pred = wp.Predict(x, y, w, top, left, topright, topleft, toptop, &wp_err);
      props[3] = x;
      props[4] = std::llabs(top);
      props[5] = std::llabs(left);
      props[6] = top;
      props[7] = left;
      props[8] = left - props[9];
      props[9] = left + top - topleft;
      props[10] = left - topleft;
      props[11] = topleft - top;
      props[12] = top - topright;
      props[13] = top - toptop;
      props[14] = left - leftleft;
      props[15] = wp_err;
      for (size_t k = 0; k < refs.size(); k++) {
        const Channel& r = chs[refs[k]];
        const int32_t* rr = r.Row(y);
        int64_t rv = rr[x];
        int64_t vl = x ? rr[x - 1] : 0;
        int64_t vt = y ? r.Row(y - 1)[x] : vl;
        int64_t vtl = (x && y) ? r.Row(y - 1)[x - 1] : vl;
        int64_t vp = ClampedGradient(vl, vt, vtl);
        props[16 + 4 * k] = std::llabs(rv);
        props[17 + 4 * k] = rv;
        props[18 + 4 * k] = std::llabs(rv - vp);
        props[19 + 4 * k] = rv - vp;
      }
      int pos = 0;
      while (tree[pos].property >= 0) {
        const TreeNode& n = tree[pos];
        int64_t pv = n.property < (int)props.size() ? props[n.property] : 0;
        pos = pv > n.splitval ? n.lchild : n.rchild;
      }
      const TreeNode& leaf = tree[pos];
      int64_t guess;
      switch (leaf.predictor) {
        case 0: guess = 0; break;
        case 1: guess = left; break;
        case 2: guess = top; break;
        case 3: guess = (left + top) / 2; break;
        case 4: {
          int64_t p = left + top - topleft;
          guess = std::llabs(p - left) < std::llabs(p - top) ? left : top;
          break;
        }
        case 5: guess = ClampedGradient(left, top, topleft); break;
        case 6: guess = wp_pred; break;
        case 7: guess = topright; break;
        case 8: guess = topleft; break;
        case 9: guess = leftleft; break;
        case 10: guess = (left + topleft) / 2; break;
        case 11: guess = (topleft + top) / 2; break;
        case 12: guess = (top + topright) / 2; break;
        case 13: guess = (6 * top - 2 * toptop + 7 * left + leftleft + toprightright + 3 * topright + 8) / 16; break;
        default: throw Error("bad predictor");
      }
      int32_t v = coder(leaf, guess, row[x]);
      row[x] = v;
      if (ti.uses_wp) wp.UpdateErrors(v, x, y, w);
    }
  }
}
}  // namespace

void DecodeChannel(EntropyReader& rd, const Tree& tree, const WPHeader& wp, ModularImage& img, int chan, uint32_t stream_id) {
  Channel& ch = img.ch[chan];
  VisitChannel(tree, wp, img.ch, chan, stream_id, ch.d.data(), [&](const TreeNode& leaf, int64_t guess, int32_t) -> int32_t {
    uint32_t v = rd.Read(leaf.leaf_id);
    int64_t val = UnpackSigned(v) * (int64_t)leaf.multiplier + leaf.offset + guess;
    return (int32_t)val;
  });
}

void TokenizeChannel(const Tree& tree, const WPHeader& wp, const ModularImage& img, int chan, uint32_t stream_id,
                     std::vector<Token>& out) {
  const Channel& ch = img.ch[chan];
  if (!ch.w || !ch.h) return;
  // VisitChannel rewrites pixels in place with the same values; work on a copy to stay const-correct.
  std::vector<int32_t> px = ch.d;
  out.reserve(out.size() + px.size());
  VisitChannel(tree, wp, img.ch, chan, stream_id, px.data(), [&](const TreeNode& leaf, int64_t guess, int32_t actual) -> int32_t {
    JXO_CHECK(leaf.multiplier == 1, "encoder trees use multiplier 1");
    int64_t res = (int64_t)actual - guess - leaf.offset;
    out.emplace_back((uint32_t)leaf.leaf_id, (uint32_t)PackSigned(res));
    return actual;
  });
}

// ------------------------------------------------------------------ transforms
static void CheckRange(const ModularImage& img, uint32_t begin, uint32_t num) {
  JXO_CHECK(begin + num <= img.ch.size() && num > 0, "transform channel range");
}

void DefaultSqueezeParams(const ModularImage& img, std::vector<SqueezeParams>& out) {
  out.clear();
  int nb = (int)img.ch.size() - img.nb_meta;
  JXO_CHECK(nb > 0, "squeeze on empty image");
  int w = img.ch[img.nb_meta].w, h = img.ch[img.nb_meta].h;
  if (nb > 2 && img.ch[img.nb_meta + 1].w == w && img.ch[img.nb_meta + 1].h == h) {
    SqueezeParams p{true, false, (uint32_t)img.nb_meta + 1, 2};
    out.push_back(p);
    p.horizontal = false;
    out.push_back(p);
  }
  SqueezeParams p{true, true, (uint32_t)img.nb_meta, (uint32_t)nb};
  const int kMaxFirstPreview = 8;
  if (h > w && h > kMaxFirstPreview) {
    p.horizontal = false;
    out.push_back(p);
    h = (h + 1) / 2;
  }
  while (w > kMaxFirstPreview || h > kMaxFirstPreview) {
    if (w > kMaxFirstPreview) { p.horizontal = true; out.push_back(p); w = (w + 1) / 2; }
    if (h > kMaxFirstPreview) { p.horizontal = false; out.push_back(p); h = (h + 1) / 2; }
  }
}

static void MetaSqueeze(ModularImage& img, std::vector<SqueezeParams>& params) {
  if (params.empty()) DefaultSqueezeParams(img, params);
  for (auto& s : params) {
    CheckRange(img, s.begin_c, s.num_c);
    uint32_t end_c = s.begin_c + s.num_c - 1;
    JXO_CHECK(!(s.begin_c < (uint32_t)img.nb_meta && end_c >= (uint32_t)img.nb_meta), "squeeze mixes meta channels");
    uint32_t offset = s.in_place ? end_c + 1 : (uint32_t)img.ch.size();
    if (s.begin_c < (uint32_t)img.nb_meta) {
      JXO_CHECK(s.in_place, "meta squeeze must be in place");
      img.nb_meta += s.num_c;
    }
    for (uint32_t c = s.begin_c; c <= end_c; c++) {
      Channel& ch = img.ch[c];
      Channel res;
      if (s.horizontal) {
        int w = ch.w;
        ch.w = (w + 1) / 2;
        ch.hshift++;
        res = Channel(w - ch.w, ch.h, ch.hshift, ch.vshift);
      } else {
        int h = ch.h;
        ch.h = (h + 1) / 2;
        ch.vshift++;
        res = Channel(ch.w, h - ch.h, ch.hshift, ch.vshift);
      }
      ch.d.assign((size_t)ch.w * ch.h, 0);
      img.ch.insert(img.ch.begin() + offset + (c - s.begin_c), res);
    }
  }
}

void MetaApplyTransforms(ModularImage& img, const GroupHeader& h) {
  img.transforms = h.transforms;
  img.wp = h.wp;
  for (auto& t : img.transforms) {
    if (t.id == 0) {
      CheckRange(img, t.begin_c, 3);
      const Channel& a = img.ch[t.begin_c];
      for (int i = 1; i < 3; i++) {
        const Channel& b = img.ch[t.begin_c + i];
        JXO_CHECK(a.w == b.w && a.h == b.h, "RCT channel size mismatch");
      }
    } else if (t.id == 1) {
      // Palette: num_c channels become one channel of indices; the colours travel in a meta channel (nb_colors x num_c) put first
      CheckRange(img, t.begin_c, t.num_c);
      JXO_CHECK(t.begin_c >= (uint32_t)img.nb_meta, "palette of meta channels is not supported");
      JXO_CHECK(t.nb_deltas == 0, "delta palettes are not supported");
      const Channel& a = img.ch[t.begin_c];
      for (uint32_t i = 1; i < t.num_c; i++) JXO_CHECK(img.ch[t.begin_c + i].w == a.w && img.ch[t.begin_c + i].h == a.h, "palette channel size mismatch");
      img.ch.erase(img.ch.begin() + t.begin_c + 1, img.ch.begin() + t.begin_c + t.num_c);
      img.ch.insert(img.ch.begin(), Channel((int)t.nb_colors, (int)t.num_c, -1, -1));
      img.nb_meta++;
    } else {
      MetaSqueeze(img, t.squeezes);
    }
  }
}

static inline int64_t SmoothTendency(int64_t B, int64_t a, int64_t n) {
  int64_t diff = 0;
  if (B >= a && a >= n) {
    diff = (4 * B - 3 * n - a + 6) / 12;
    if (diff - (diff & 1) > 2 * (B - a)) diff = 2 * (B - a) + 1;
    if (diff + (diff & 1) > 2 * (a - n)) diff = 2 * (a - n);
  } else if (B <= a && a <= n) {
    diff = (4 * B - 3 * n - a - 6) / 12;
    if (diff + (diff & 1) < 2 * (B - a)) diff = 2 * (B - a) - 1;
    if (diff - (diff & 1) < 2 * (a - n)) diff = 2 * (a - n);
  }
  return diff;
}

static void InvHSqueeze(ModularImage& img, uint32_t c, uint32_t rc) {
  const Channel& avg = img.ch[c];
  const Channel& res = img.ch[rc];
  JXO_CHECK(avg.h == res.h && (avg.w == res.w || avg.w == res.w + 1), "hsqueeze dims");
  Channel out(avg.w + res.w, avg.h, avg.hshift - 1, avg.vshift);
  for (int y = 0; y < avg.h; y++) {
    const int32_t* pa = avg.Row(y);
    const int32_t* pr = res.Row(y);
    int32_t* po = out.Row(y);
    for (int x = 0; x < res.w; x++) {
      int64_t dmt = pr[x], a = pa[x];
      int64_t next = x + 1 < avg.w ? pa[x + 1] : a;
      int64_t left = x ? po[2 * x - 1] : a;
      int64_t diff = dmt + SmoothTendency(left, a, next);
      int64_t A = ((a * 2) + diff + (diff > 0 ? -(diff & 1) : (diff & 1))) >> 1;
      po[2 * x] = (int32_t)A;
      po[2 * x + 1] = (int32_t)(A - diff);
    }
    if (avg.w > res.w) po[2 * res.w] = pa[res.w];
  }
  img.ch[c] = out;
}

static void InvVSqueeze(ModularImage& img, uint32_t c, uint32_t rc) {
  const Channel& avg = img.ch[c];
  const Channel& res = img.ch[rc];
  JXO_CHECK(avg.w == res.w && (avg.h == res.h || avg.h == res.h + 1), "vsqueeze dims");
  Channel out(avg.w, avg.h + res.h, avg.hshift, avg.vshift - 1);
  for (int y = 0; y < res.h; y++) {
    const int32_t* pa = avg.Row(y);
    const int32_t* pn = y + 1 < avg.h ? avg.Row(y + 1) : pa;
    const int32_t* pr = res.Row(y);
    int32_t* p0 = out.Row(2 * y);
    int32_t* p1 = out.Row(2 * y + 1);
    const int32_t* pt = y ? out.Row(2 * y - 1) : pa;
    for (int x = 0; x < avg.w; x++) {
      int64_t dmt = pr[x], a = pa[x];
      int64_t next = pn[x];
      int64_t top = pt[x];
      int64_t diff = dmt + SmoothTendency(top, a, next);
      int64_t A = ((a * 2) + diff + (diff > 0 ? -(diff & 1) : (diff & 1))) >> 1;
      p0[x] = (int32_t)A;
      p1[x] = (int32_t)(A - diff);
    }
  }
  if (avg.h > res.h) memcpy(out.Row(2 * res.h), avg.Row(res.h), sizeof(int32_t) * avg.w);
  img.ch[c] = out;
}

static void FwdHSqueeze(ModularImage& img, uint32_t c, uint32_t rc) {
  const Channel in = img.ch[c];
  int wa = (in.w + 1) / 2, wr = in.w - wa;
  Channel avg(wa, in.h, in.hshift + 1, in.vshift), res(wr, in.h, in.hshift + 1, in.vshift);
  for (int y = 0; y < in.h; y++) {
    const int32_t* p = in.Row(y);
    int32_t* pa = avg.Row(y);
    int32_t* pr = res.Row(y);
    for (int x = 0; x < wa; x++) {
      if (2 * x + 1 < in.w) {
        int64_t A = p[2 * x], B = p[2 * x + 1];
        pa[x] = (int32_t)((A + B + (A > B)) >> 1);
      } else pa[x] = p[2 * x];
    }
    for (int x = 0; x < wr; x++) {
      int64_t A = p[2 * x], B = p[2 * x + 1];
      int64_t a = pa[x];
      int64_t next = x + 1 < wa ? pa[x + 1] : a;
      int64_t left = x ? p[2 * x - 1] : a;
      pr[x] = (int32_t)((A - B) - SmoothTendency(left, a, next));
    }
  }
  img.ch[c] = avg;
  img.ch.insert(img.ch.begin() + rc, res);
}

static void FwdVSqueeze(ModularImage& img, uint32_t c, uint32_t rc) {
  const Channel in = img.ch[c];
  int ha = (in.h + 1) / 2, hr = in.h - ha;
  Channel avg(in.w, ha, in.hshift, in.vshift + 1), res(in.w, hr, in.hshift, in.vshift + 1);
  for (int y = 0; y < ha; y++) {
    const int32_t* p0 = in.Row(2 * y);
    int32_t* pa = avg.Row(y);
    if (2 * y + 1 < in.h) {
      const int32_t* p1 = in.Row(2 * y + 1);
      for (int x = 0; x < in.w; x++) {
        int64_t A = p0[x], B = p1[x];
        pa[x] = (int32_t)((A + B + (A > B)) >> 1);
      }
    } else memcpy(pa, p0, sizeof(int32_t) * in.w);
  }
  for (int y = 0; y < hr; y++) {
    const int32_t* p0 = in.Row(2 * y);
    const int32_t* p1 = in.Row(2 * y + 1);
    const int32_t* pa = avg.Row(y);
    const int32_t* pn = y + 1 < ha ? avg.Row(y + 1) : pa;
    const int32_t* pt = y ? in.Row(2 * y - 1) : pa;
    int32_t* pr = res.Row(y);
    for (int x = 0; x < in.w; x++) {
      int64_t A = p0[x], B = p1[x];
      pr[x] = (int32_t)((A - B) - SmoothTendency(pt[x], pa[x], pn[x]));
    }
  }
  img.ch[c] = avg;
  img.ch.insert(img.ch.begin() + rc, res);
}

static void InvRCT(ModularImage& img, uint32_t begin_c, uint32_t rct_type) {
  int permutation = rct_type / 7, custom = rct_type % 7;
  Channel& c0 = img.ch[begin_c];
  Channel& c1 = img.ch[begin_c + 1];
  Channel& c2 = img.ch[begin_c + 2];
  size_t n = c0.d.size();
  int second = custom >> 1, third = custom & 1;
  for (size_t i = 0; i < n; i++) {
    int32_t a = c0.d[i], b = c1.d[i], c = c2.d[i];
    if (custom == 6) {
      int32_t tmp = a - (c >> 1);
      int32_t G = c + tmp;
      int32_t B = tmp - (b >> 1);
      int32_t R = B + b;
      c0.d[i] = R; c1.d[i] = G; c2.d[i] = B;
    } else {
      if (third) c = c + a;
      if (second == 1) b = b + a;
      else if (second == 2) b = b + ((a + c) >> 1);
      c0.d[i] = a; c1.d[i] = b; c2.d[i] = c;
    }
  }
  if (permutation) {
    Channel t0 = c0, t1 = c1, t2 = c2;
    img.ch[begin_c + (permutation % 3)] = t0;
    img.ch[begin_c + ((permutation + 1 + permutation / 3) % 3)] = t1;
    img.ch[begin_c + ((permutation + 2 - permutation / 3) % 3)] = t2;
  }
}

void ForwardRCT(ModularImage& img, uint32_t begin_c, uint32_t rct_type) {
  int permutation = rct_type / 7, custom = rct_type % 7;
  // inverse of the output permutation
  Channel in[3];
  in[0] = img.ch[begin_c + (permutation % 3)];
  in[1] = img.ch[begin_c + ((permutation + 1 + permutation / 3) % 3)];
  in[2] = img.ch[begin_c + ((permutation + 2 - permutation / 3) % 3)];
  size_t n = in[0].d.size();
  int second = custom >> 1, third = custom & 1;
  for (size_t i = 0; i < n; i++) {
    int32_t a = in[0].d[i], b = in[1].d[i], c = in[2].d[i];
    if (custom == 6) {
      int32_t R = a, G = b, B = c;
      int32_t Co = R - B;
      int32_t tmp = B + (Co >> 1);
      int32_t Cg = G - tmp;
      int32_t Y = tmp + (Cg >> 1);
      in[0].d[i] = Y; in[1].d[i] = Co; in[2].d[i] = Cg;
    } else {
      // invert: c' = c + a (if third); b' = b + a | b + ((a + c') >> 1)
      int32_t cc = c;  // reconstructed third
      if (second == 1) b = b - a;
      else if (second == 2) b = b - ((a + cc) >> 1);
      if (third) c = c - a;
      in[0].d[i] = a; in[1].d[i] = b; in[2].d[i] = c;
    }
  }
  for (int i = 0; i < 3; i++) img.ch[begin_c + i] = in[i];
  Transform t;
  t.id = 0; t.begin_c = begin_c; t.rct_type = rct_type;
  img.transforms.push_back(t);
}

bool ForwardPalette(ModularImage& img, uint32_t begin_c, uint32_t num_c, size_t max_colors) {
  CheckRange(img, begin_c, num_c);
  const int w = img.ch[begin_c].w, h = img.ch[begin_c].h;
  std::vector<std::vector<int32_t>> colors;
  {
    std::set<std::vector<int32_t>> seen;
    std::vector<int32_t> px(num_c);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        for (uint32_t c = 0; c < num_c; c++) px[c] = img.ch[begin_c + c].Row(y)[x];
        seen.insert(px);
        if (seen.size() > max_colors) return false;
      }
    colors.assign(seen.begin(), seen.end());   // lexicographic order
  }
  std::map<std::vector<int32_t>, int32_t> index;
  for (size_t i = 0; i < colors.size(); i++) index[colors[i]] = (int32_t)i;
  Channel pal((int)colors.size(), (int)num_c, -1, -1);
  for (size_t i = 0; i < colors.size(); i++)
    for (uint32_t c = 0; c < num_c; c++) pal.Row((int)c)[i] = colors[i][c];
  Channel idx(w, h, img.ch[begin_c].hshift, img.ch[begin_c].vshift);
  std::vector<int32_t> px(num_c);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      for (uint32_t c = 0; c < num_c; c++) px[c] = img.ch[begin_c + c].Row(y)[x];
      idx.Row(y)[x] = index[px];
    }
  img.ch.erase(img.ch.begin() + begin_c, img.ch.begin() + begin_c + num_c);
  img.ch.insert(img.ch.begin() + begin_c, idx);
  img.ch.insert(img.ch.begin(), pal);
  img.nb_meta++;
  Transform t;
  t.id = 1; t.begin_c = begin_c; t.num_c = num_c; t.nb_colors = (uint32_t)colors.size(); t.nb_deltas = 0; t.predictor = 0;
  img.transforms.push_back(t);
  return true;
}

void ForwardSqueeze(ModularImage& img, const std::vector<SqueezeParams>& params_in) {
  std::vector<SqueezeParams> params = params_in;
  if (params.empty()) DefaultSqueezeParams(img, params);
  for (auto& s : params) {
    uint32_t end_c = s.begin_c + s.num_c - 1;
    uint32_t offset = s.in_place ? end_c + 1 : (uint32_t)img.ch.size();
    for (uint32_t c = s.begin_c; c <= end_c; c++) {
      uint32_t rc = offset + (c - s.begin_c);
      if (s.horizontal) FwdHSqueeze(img, c, rc); else FwdVSqueeze(img, c, rc);
    }
  }
  Transform t;
  t.id = 2;
  t.squeezes = params;
  img.transforms.push_back(t);
}

void UndoTransforms(ModularImage& img) {
  for (size_t ti = img.transforms.size(); ti-- > 0;) {
    Transform& t = img.transforms[ti];
    if (t.id == 0) {
      InvRCT(img, t.begin_c, t.rct_type);
    } else if (t.id == 2) {
      for (size_t si = t.squeezes.size(); si-- > 0;) {
        const SqueezeParams& s = t.squeezes[si];
        uint32_t end_c = s.begin_c + s.num_c - 1;
        uint32_t offset = s.in_place ? end_c + 1 : (uint32_t)(img.ch.size() - s.num_c);
        if (s.begin_c < (uint32_t)img.nb_meta) img.nb_meta -= s.num_c;
        for (uint32_t c = s.begin_c; c <= end_c; c++) {
          uint32_t rc = offset + (c - s.begin_c);
          if (s.horizontal) InvHSqueeze(img, c, rc); else InvVSqueeze(img, c, rc);
        }
        img.ch.erase(img.ch.begin() + offset, img.ch.begin() + offset + s.num_c);
      }
    } else {
      // inverse Palette: look every index up (indices outside the stored palette - implicit and delta colours - are not supported)
      const Channel pal = img.ch[0];
      const uint32_t ic = t.begin_c + 1;   // the index channel, after the meta channel in front
      const Channel idx = img.ch[ic];
      std::vector<Channel> out;
      for (uint32_t c = 0; c < t.num_c; c++) out.emplace_back(idx.w, idx.h, idx.hshift, idx.vshift);
      for (int y = 0; y < idx.h; y++)
        for (int x = 0; x < idx.w; x++) {
          const int32_t i = idx.Row(y)[x];
          JXO_CHECK(i >= 0 && i < pal.w, "palette index outside the stored palette");
          for (uint32_t c = 0; c < t.num_c; c++) out[c].Row(y)[x] = pal.Row((int)c)[i];
        }
      img.ch.erase(img.ch.begin() + ic);
      img.ch.insert(img.ch.begin() + ic, out.begin(), out.end());
      img.ch.erase(img.ch.begin());
      img.nb_meta--;
    }
  }
  img.transforms.clear();
}

// ------------------------------------------------------------------ generic sub-stream decode
void ModularDecode(BitReader& br, ModularImage& img, GroupHeader* header_out, uint32_t stream_id, int max_chan_size,
                   const Tree* global_tree, const EntropyCode* global_code) {
  if (img.ch.empty()) return;
  GroupHeader h;
  ReadGroupHeader(br, h);
  if (header_out) *header_out = h;
  MetaApplyTransforms(img, h);
  size_t nb = img.ch.size(), num_chans = 0, dist_mult = 0, pixels = 0;
  for (size_t i = 0; i < nb; i++) {
    Channel& c = img.ch[i];
    if (!c.w || !c.h) continue;
    if ((int)i >= img.nb_meta && (c.w > max_chan_size || c.h > max_chan_size)) break;
    dist_mult = std::max<size_t>(dist_mult, c.w);
    pixels += (size_t)c.w * c.h;
    num_chans++;
  }
  if (!num_chans) return;
  Tree local_tree;
  EntropyCode local_code;
  const Tree* tree = global_tree;
  const EntropyCode* code = global_code;
  if (!h.use_global_tree) {
    size_t limit = std::min<size_t>((size_t)1 << 20, 1024 + pixels);
    DecodeTree(br, local_tree, limit);
    DecodeHistograms(br, (local_tree.size() + 1) / 2, local_code);
    tree = &local_tree;
    code = &local_code;
  } else {
    JXO_CHECK(tree && code && !tree->empty(), "global MA tree requested but absent");
  }
  EntropyReader rd;
  rd.Init(*code, br, (uint32_t)dist_mult);
  for (size_t i = 0; i < nb; i++) {
    Channel& c = img.ch[i];
    if (!c.w || !c.h) continue;
    if ((int)i >= img.nb_meta && (c.w > max_chan_size || c.h > max_chan_size)) break;
    DecodeChannel(rd, *tree, h.wp, img, (int)i, stream_id);
    JXO_CHECK(!br.overrun, "truncated modular stream");
  }
  JXO_CHECK(rd.CheckFinal(), "modular stream ANS final state");
}

}  // namespace jxo
