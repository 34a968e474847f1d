// ORACLE — test infrastructure only (see jxo_common.h header).
// Modular sub-bitstream of ISO/IEC 18181-1 (MA-tree context modelling, predictors incl.
// the self-correcting "weighted" predictor, RCT / Squeeze transforms).  Used for the LF
// image, HF metadata, alpha, and lossless frames (BASELINE.json configs[4]).
#pragma once
#include "jxo_common.h"
#include "jxo_entropy.h"

namespace jxo {

struct Channel {
  int w = 0, h = 0;
  int hshift = 0, vshift = 0;
  std::vector<int32_t> d;
  Channel() {}
  Channel(int w_, int h_, int hs = 0, int vs = 0) : w(w_), h(h_), hshift(hs), vshift(vs), d((size_t)w_ * h_, 0) {}
  int32_t* Row(int y) { return d.data() + (size_t)y * w; }
  const int32_t* Row(int y) const { return d.data() + (size_t)y * w; }
};

struct TreeNode {
  int property = -1;  // -1: leaf
  int32_t splitval = 0;
  int lchild = 0, rchild = 0;  // lchild: property > splitval
  int predictor = 0;
  int64_t offset = 0;
  uint32_t multiplier = 1;
  int leaf_id = 0;  // context
};
typedef std::vector<TreeNode> Tree;

struct WPHeader {
  bool default_wp = true;
  int p1C = 16, p2C = 10, p3Ca = 7, p3Cb = 7, p3Cc = 7, p3Cd = 0, p3Ce = 0;
  int w[4] = {13, 12, 12, 12};
};

struct SqueezeParams { bool horizontal, in_place; uint32_t begin_c, num_c; };
struct Transform {
  int id = 0;  // 0 RCT, 1 Palette, 2 Squeeze
  uint32_t begin_c = 0, rct_type = 6;
  uint32_t num_c = 3, nb_colors = 256, nb_deltas = 0, predictor = 0;
  std::vector<SqueezeParams> squeezes;
};

struct GroupHeader {
  bool use_global_tree = false;
  WPHeader wp;
  std::vector<Transform> transforms;
};

struct ModularImage {
  std::vector<Channel> ch;
  int nb_meta = 0;
  std::vector<Transform> transforms;  // as applied (with default squeeze params expanded)
  WPHeader wp;
};

constexpr int kNumNonrefProps = 16;

void ReadGroupHeader(BitReader& br, GroupHeader& h);
void WriteGroupHeader(BitWriter& bw, const GroupHeader& h);
void DecodeTree(BitReader& br, Tree& tree, size_t size_limit);
// Applies the channel-list effect of the header's transforms to `img`.
void MetaApplyTransforms(ModularImage& img, const GroupHeader& h);
void UndoTransforms(ModularImage& img);
// Forward transforms for the encoder (applies and records in img.transforms).
void ForwardRCT(ModularImage& img, uint32_t begin_c, uint32_t rct_type);
void ForwardSqueeze(ModularImage& img, const std::vector<SqueezeParams>& params);
// channels [begin_c, begin_c + num_c) -> one index channel + a palette meta channel in front; false (nothing changed) above max_colors
bool ForwardPalette(ModularImage& img, uint32_t begin_c, uint32_t num_c, size_t max_colors);
void DefaultSqueezeParams(const ModularImage& img, std::vector<SqueezeParams>& out);

// Decodes one channel with an already-initialised reader.
void DecodeChannel(EntropyReader& rd, const Tree& tree, const WPHeader& wp, ModularImage& img, int chan, uint32_t stream_id);

// Generic modular sub-stream decode (header, optional local tree, channels up to max_chan_size).
// header_out may be null.  Returns after reading channel data; does NOT undo transforms.
void ModularDecode(BitReader& br, ModularImage& img, GroupHeader* header_out, uint32_t stream_id, int max_chan_size,
                   const Tree* global_tree, const EntropyCode* global_code);

// ------------------------------------------------------------------ encoder helpers
// Tokenises channel `chan` of img against `tree` (multiplier 1, offset 0 leaves only).
void TokenizeChannel(const Tree& tree, const WPHeader& wp, const ModularImage& img, int chan, uint32_t stream_id,
                     std::vector<Token>& out);
void TokenizeTree(const Tree& tree, std::vector<Token>& out);
// Writes tree (its own 6-context code) to bw.
void WriteTree(BitWriter& bw, const Tree& tree);
// Builds a tree from a compact description: nodes listed in any order with explicit child links
// are re-laid-out in BFS order and leaf ids assigned in BFS order.
Tree MakeBfsTree(const Tree& linked, int root = 0);

}  // namespace jxo
