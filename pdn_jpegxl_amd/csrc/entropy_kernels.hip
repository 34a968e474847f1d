// HIP kernels (gfx950) for the serial, entropy-coded parts of a JPEG XL VarDCT frame:
//   lf_group_kernel    LF coefficients + HF metadata (Modular: ANS + MA tree) and varblock placement
//   hf_decode_kernel   HF coefficient tokens of every 256x256 group -> quantised coefficient planes
//   alpha_kernel       the Modular alpha stream that follows the HF tokens in each pass-group section
//
// Design (DESIGN.md "Entropy kernels"):
//  * every lane owns one section: a bit reader with one-word lookahead and a 32-bit ANS state;
//  * the code tables (context map, alias tables, hybrid-uint configs, MA tree, small coefficient orders)
//    are staged once per workgroup in LDS; the previous image row needed by the predictors lives in a
//    bank-swizzled per-lane LDS row buffer, so the token loop touches HBM only for the (prefetched)
//    bitstream words and fire-and-forget stores;
//  * the HF decoder is flattened to ONE token per loop iteration (block / channel bookkeeping is folded
//    into the same loop), so lanes of a wavefront that own different sections stay convergent;
//  * `lane_stride` picks the mapping: 64 = one section per wavefront, 1 = one section per lane.
#include <hip/hip_runtime.h>
#include "dev_types.h"
#include "kernels.h"

namespace jxlhip {

namespace {

__device__ const uint8_t d_order_bucket[kNumStrategies] = {0, 1, 1, 1, 2, 3, 4, 4, 5, 5, 6, 6, 1, 1, 1, 1, 1, 1, 7, 8, 8, 9, 10, 10, 11, 12, 12};
__device__ const uint8_t d_log2cx[kNumStrategies] = {0, 0, 0, 0, 1, 2, 0, 1, 0, 2, 1, 2, 0, 0, 0, 0, 0, 0, 3, 2, 3, 4, 3, 4, 5, 4, 5};
__device__ const uint8_t d_log2cy[kNumStrategies] = {0, 0, 0, 0, 1, 2, 1, 0, 2, 0, 2, 1, 0, 0, 0, 0, 0, 0, 3, 3, 2, 4, 4, 3, 5, 5, 4};
__device__ const uint8_t d_nnz_ctx[64] = {0,   0,   31,  62,  62,  93,  93,  93,  93,  123, 123, 123, 123, 152, 152, 152, 152, 152, 152, 152, 152, 180,
                                          180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206,
                                          206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206};
// natural coefficient orders small enough for LDS: buckets 0..8, entry offsets
__device__ const uint16_t d_order_lds_off[10] = {0, 64, 128, 384, 1408, 1536, 1792, 2304, 6400, 8448};

__device__ __forceinline__ bool IsSpecialS(uint32_t s) { return (s >= 1 && s <= 3) || (s >= 12 && s <= 17); }
__device__ __forceinline__ int CeilLog2D(uint32_t x) { return x <= 1 ? 0 : 32 - __clz(x - 1); }
__device__ __forceinline__ int32_t UnpackSigned(uint32_t u) { return (int32_t)(u >> 1) ^ -(int32_t)(u & 1); }
__device__ __forceinline__ void SetError(const DevImage& im, uint32_t bits) { atomicOr(im.status, bits); }

// Table pointers are typed by address space so that the LDS variants compile to ds_read (not flat_load).
#define JXL_LDS __attribute__((address_space(3)))
#define JXL_GLB __attribute__((address_space(1)))
// pointers read out of DevImage are generic; cast the hot ones so they compile to global_load/global_store
template <class T> __device__ __forceinline__ JXL_GLB T* G(T* p) { return (JXL_GLB T*)p; }
typedef int __attribute__((ext_vector_type(4))) I4;   // one MA-tree node (builtin vector: loadable from any address space)

// ------------------------------------------------------------------ per-lane bit reader (one-word lookahead)
struct LaneBits {
  const JXL_GLB uint32_t* w;
  uint32_t idx, nwords;   // idx: index of the word held in `next`
  uint64_t buf;
  int n;
  int skip;
  uint32_t next;
  __device__ void Init(const uint8_t* cs, uint64_t cs_size, uint64_t bit_off) {
    const uintptr_t addr = (uintptr_t)cs + (bit_off >> 3);
    const uintptr_t al = addr & ~(uintptr_t)3;
    w = (const JXL_GLB uint32_t*)al;
    nwords = (uint32_t)(((uintptr_t)(cs + cs_size) + 3 - al) >> 2);
    idx = 0; buf = 0; n = 0;
    skip = (int)(addr - al) * 8 + (int)(bit_off & 7);
    next = nwords ? w[0] : 0u;
    Refill();
    buf >>= skip;
    n -= skip;
  }
  __device__ __forceinline__ void Refill() {
    if (n <= 32) {
      buf |= (uint64_t)next << n;
      n += 32;
      idx++;
      next = idx < nwords ? w[idx] : 0u;
    }
  }
  __device__ __forceinline__ uint32_t Read(int k) {   // k <= 32
    Refill();
    const uint32_t v = (uint32_t)(buf & (((uint64_t)1 << k) - 1));
    buf >>= k;
    n -= k;
    return v;
  }
  // bits consumed since Init (relative to the bit offset given to Init)
  __device__ uint64_t Consumed() const { return (uint64_t)idx * 32 - n - skip; }
};

template <bool kLds> struct AS;
template <> struct AS<true> {
  typedef const JXL_LDS uint8_t* U8;
  typedef const JXL_LDS uint16_t* U16;
  typedef const JXL_LDS uint32_t* U32;
  typedef const JXL_LDS uint64_t* U64;
  typedef const JXL_LDS I4* Tree;
  typedef JXL_LDS int32_t* Row;
};
template <> struct AS<false> {
  typedef const uint8_t* U8;
  typedef const uint16_t* U16;
  typedef const uint32_t* U32;
  typedef const uint64_t* U64;
  typedef const I4* Tree;
  typedef int32_t* Row;
};

template <bool kLds>
struct CodeTab {
  typename AS<kLds>::U8 cmap;
  typename AS<kLds>::U32 cfg;     // split | msb << 4 | lsb << 8 | degenerate << 12 | symbol << 16
  typename AS<kLds>::U64 alias;
  uint32_t log_alpha;
};

__device__ __forceinline__ DevTreeNode NodeOf(I4 v) {
  DevTreeNode n;
  n.property = v.x; n.splitval = v.y; n.a = (uint32_t)v.z; n.b = (uint32_t)v.w;
  return n;
}

__device__ __forceinline__ uint32_t HybridTail(LaneBits& b, uint32_t c, uint32_t sym) {
  const uint32_t se = c & 0xF, split = 1u << se;
  if (sym < split) return sym;
  const uint32_t msb = (c >> 4) & 0xF, lsb = (c >> 8) & 0xF;
  const uint32_t nb = se - (msb + lsb) + ((sym - split) >> (msb + lsb));
  const uint32_t low = sym & ((1u << lsb) - 1);
  const uint32_t t = sym >> lsb;
  const uint32_t bits = b.Read(nb > 32 ? 32 : nb);
  const uint32_t hi = (1u << msb) | (t & ((1u << msb) - 1));
  return (uint32_t)(((((uint64_t)hi << nb) | bits) << lsb) | low);
}

// One ANS symbol from the alias table of a fixed cluster (`abase` = that cluster's table).
template <bool kLds>
__device__ __forceinline__ uint32_t AnsSym(LaneBits& b, uint32_t& state, typename AS<kLds>::U64 abase, uint32_t log_alpha) {
  const uint32_t le = 12 - log_alpha;
  const uint32_t res = state & 0xFFF, i = res >> le, pos = res & ((1u << le) - 1);
  const uint64_t e = abase[i];
  const uint32_t x = (uint32_t)e, y = (uint32_t)(e >> 32);
  const bool g = pos >= (x & 0xFF);
  const uint32_t sym = g ? ((x >> 8) & 0xFF) : i;
  const uint32_t off = g ? (y & 0xFFFF) + pos : pos;
  const uint32_t freq = g ? ((x >> 16) ^ (y >> 16)) : (x >> 16);
  state = freq * (state >> 12) + off;
  if (state < 65536u) state = (state << 16) | b.Read(16);
  return sym;
}

template <bool kLds>
__device__ __forceinline__ uint32_t AnsGet(LaneBits& b, uint32_t& state, const CodeTab<kLds>& t, uint32_t ctx) {
  const uint32_t cl = t.cmap[ctx];
  const uint32_t le = 12 - t.log_alpha;
  const uint32_t res = state & 0xFFF, i = res >> le, pos = res & ((1u << le) - 1);
  const uint64_t e = t.alias[(cl << t.log_alpha) | i];
  const uint32_t c = t.cfg[cl];
  const uint32_t x = (uint32_t)e, y = (uint32_t)(e >> 32);
  const bool g = pos >= (x & 0xFF);
  const uint32_t sym = g ? ((x >> 8) & 0xFF) : i;
  const uint32_t off = g ? (y & 0xFFFF) + pos : pos;
  const uint32_t freq = g ? ((x >> 16) ^ (y >> 16)) : (x >> 16);
  state = freq * (state >> 12) + off;
  if (state < 65536u) state = (state << 16) | b.Read(16);
  return HybridTail(b, c, sym);
}

// Cooperative copy of a code's tables into LDS; returns the carved end offset.
__device__ __forceinline__ size_t StageCode(JXL_LDS uint8_t* smem, size_t off, const DevCode& dc, CodeTab<true>& t, int tid, int nt) {
  const uint32_t na = dc.num_clusters << dc.log_alpha;
  off = (off + 7) & ~(size_t)7;
  JXL_LDS uint64_t* sa = (JXL_LDS uint64_t*)(smem + off); off += (size_t)na * 8;
  JXL_LDS uint32_t* sc = (JXL_LDS uint32_t*)(smem + off); off += (size_t)dc.num_clusters * 4;
  JXL_LDS uint8_t* sm = smem + off; off += dc.num_ctx;
  for (uint32_t i = tid; i < na; i += nt) sa[i] = dc.alias[i];
  for (uint32_t i = tid; i < dc.num_clusters; i += nt) sc[i] = dc.cfg[i];
  for (uint32_t i = tid; i < dc.num_ctx; i += nt) sm[i] = dc.ctx_map[i];
  t.cmap = sm; t.cfg = sc; t.alias = sa; t.log_alpha = dc.log_alpha;
  return off;
}

__device__ __forceinline__ void GlobalCode(const DevCode& dc, CodeTab<false>& t) {
  t.cmap = dc.ctx_map; t.cfg = dc.cfg; t.alias = dc.alias; t.log_alpha = dc.log_alpha;
}

// ------------------------------------------------------------------ Modular channel
// Row buffer: the previous row (and, behind the cursor, the current row) of the channel being decoded,
// element x at rb[x * rb_stride] (rb_stride = sections per workgroup => lanes hit distinct LDS banks).
// Channels wider than rb_width fall back to reading the previous row from the output plane.
template <bool kLds>
struct RowBuf {
  typename AS<kLds>::Row rb;
  int rb_stride;
  int rb_width;
};

// Fast path: a whole row whose MA-tree walk ends in one leaf before any sample-dependent property (the common
// case for LF / metadata / alpha streams).  Predictor and cluster are loop invariants; kPred is a compile-time
// predictor id for the cheap ones (-1: any other, evaluated by the generic path instead).
template <bool kLds, int kPred>
__device__ __forceinline__ void LeafRow(LaneBits& b, uint32_t& state, const CodeTab<kLds>& tab, const DevTreeNode& leaf, int w, int y,
                                        JXL_GLB int32_t* row, int stride, typename AS<kLds>::Row rb, int rs, bool use_rb) {
  const uint32_t cl = tab.cmap[leaf.a >> 8];
  const uint32_t cfg = tab.cfg[cl];
  const typename AS<kLds>::U64 abase = tab.alias + (cl << tab.log_alpha);
  const bool constant_token = (cfg & 0x1000) && ((cfg >> 16) & 0xFF) < (1u << (cfg & 0xF));
  const uint32_t const_res = (uint32_t)UnpackSigned((cfg >> 16) & 0xFF) * leaf.b + (uint32_t)leaf.splitval;
  const JXL_GLB int32_t* prow = row - stride;
  int32_t W = y ? (use_rb ? rb[0] : prow[0]) : 0, N = W, NW = W;
  for (int x = 0; x < w; x++) {
    int32_t NE = N;
    if (kPred != 0 && kPred != 1) {
      if (x + 1 < w && y) NE = use_rb ? rb[(x + 1) * rs] : prow[x + 1];
    }
    uint32_t guess;
    if (kPred == 0) guess = 0;
    else if (kPred == 1) guess = (uint32_t)W;
    else if (kPred == 2) guess = (uint32_t)N;
    else {   // 5: clamped gradient
      const int64_t mn = W < N ? W : N, mx = W < N ? N : W, gr = (int64_t)W + N - NW;
      guess = (uint32_t)(int32_t)(gr < mn ? mn : (gr > mx ? mx : gr));
    }
    uint32_t res;
    if (constant_token) res = const_res;
    else {
      const uint32_t sym = AnsSym<kLds>(b, state, abase, tab.log_alpha);
      res = (uint32_t)UnpackSigned(HybridTail(b, cfg, sym)) * leaf.b + (uint32_t)leaf.splitval;
    }
    const int32_t val = (int32_t)(res + guess);   // low 32 bits of the 64-bit reference arithmetic
    row[x] = val;
    if (use_rb) rb[x * rs] = val;
    W = val;
    if (y) { NW = N; N = NE; } else { NW = val; N = val; }
  }
}

// Decodes one channel (w x h) into `out` (row stride `stride`).  Every property 0..14 and every predictor
// except the weighted one are supported; the host rejects trees that need more.
template <bool kLds>
__device__ void ModularChannel(LaneBits& b, uint32_t& state, const CodeTab<kLds>& tab, typename AS<kLds>::Tree tree, int chan,
                               int stream_id, int w, int h, int32_t* out_generic, int stride, const RowBuf<kLds>& rbuf) {
  JXL_GLB int32_t* const out = G(out_generic);
  int root = 0;
  for (;;) {
    const DevTreeNode nd = NodeOf(tree[root]);
    if (nd.property != 0 && nd.property != 1) break;
    const int v = nd.property == 0 ? chan : stream_id;
    root = v > nd.splitval ? nd.a : nd.b;
  }
  const bool use_rb = w <= rbuf.rb_width;
  const typename AS<kLds>::Row rb = rbuf.rb;
  const int rs = rbuf.rb_stride;
  int x = 0, y = 0;
  int rroot = root;
  bool row_leaf = false;
  DevTreeNode leaf = NodeOf(tree[root]);
  int64_t prev9 = 0;
  int32_t W = 0, N = 0, NW = 0, WW = 0;
  JXL_GLB int32_t* row = out;
  while (y < h) {
    const JXL_GLB int32_t* prow = row - stride;
    if (x == 0) {
      rroot = root;
      for (;;) {
        const DevTreeNode nd = NodeOf(tree[rroot]);
        if (nd.property < 0 || nd.property > 2) { leaf = nd; break; }
        const int v = nd.property == 0 ? chan : (nd.property == 1 ? stream_id : y);
        rroot = v > nd.splitval ? nd.a : nd.b;
      }
      row_leaf = leaf.property < 0;
      if (row_leaf) {
        const uint32_t lp = leaf.a & 0xFF;
        if (lp == 0 || lp == 1 || lp == 2 || lp == 5) {
          if (lp == 0) LeafRow<kLds, 0>(b, state, tab, leaf, w, y, row, stride, rb, rs, use_rb);
          else if (lp == 1) LeafRow<kLds, 1>(b, state, tab, leaf, w, y, row, stride, rb, rs, use_rb);
          else if (lp == 2) LeafRow<kLds, 2>(b, state, tab, leaf, w, y, row, stride, rb, rs, use_rb);
          else LeafRow<kLds, 5>(b, state, tab, leaf, w, y, row, stride, rb, rs, use_rb);
          y++;
          row += stride;
          continue;
        }
      }
      W = y ? (use_rb ? rb[0] : prow[0]) : 0;
      N = W; NW = W; WW = W;
      prev9 = 0;
    }
    const int32_t NE = (x + 1 < w && y) ? (use_rb ? rb[(x + 1) * rs] : prow[x + 1]) : N;
    DevTreeNode nd = leaf;
    if (!row_leaf) {
      int node = rroot;
      nd = NodeOf(tree[node]);
      while (nd.property >= 0) {
        int64_t p;
        switch (nd.property) {
          case 0: p = chan; break;
          case 1: p = stream_id; break;
          case 2: p = y; break;
          case 3: p = x; break;
          case 4: p = N < 0 ? -(int64_t)N : N; break;
          case 5: p = W < 0 ? -(int64_t)W : W; break;
          case 6: p = N; break;
          case 7: p = W; break;
          case 8: p = (int64_t)W - prev9; break;
          case 9: p = (int64_t)W + N - NW; break;
          case 10: p = (int64_t)W - NW; break;
          case 11: p = (int64_t)NW - N; break;
          case 12: p = (int64_t)N - NE; break;
          case 13: { const int32_t NN = y > 1 ? prow[x - stride] : N; p = (int64_t)N - NN; break; }
          case 14: p = (int64_t)W - WW; break;
          default: p = 0; break;
        }
        node = p > nd.splitval ? nd.a : nd.b;
        nd = NodeOf(tree[node]);
      }
    }
    const uint32_t pred = nd.a & 0xFF, ctx = nd.a >> 8;
    int64_t guess;
    switch (pred) {
      case 0: guess = 0; break;
      case 1: guess = W; break;
      case 2: guess = N; break;
      case 3: guess = ((int64_t)W + N) / 2; break;
      case 4: {
        int64_t pp = (int64_t)W + N - NW, pa = pp - W, pb = pp - N;
        if (pa < 0) pa = -pa;
        if (pb < 0) pb = -pb;
        guess = pa < pb ? W : N;
        break;
      }
      case 5: {
        const int64_t mn = W < N ? W : N, mx = W < N ? N : W, gr = (int64_t)W + N - NW;
        guess = gr < mn ? mn : (gr > mx ? mx : gr);
        break;
      }
      case 7: guess = NE; break;
      case 8: guess = NW; break;
      case 9: guess = WW; break;
      case 10: guess = ((int64_t)W + NW) / 2; break;
      case 11: guess = ((int64_t)NW + N) / 2; break;
      case 12: guess = ((int64_t)N + NE) / 2; break;
      case 13: {
        const int32_t NN = y > 1 ? prow[x - stride] : N;
        const int32_t NEE = (x + 2 < w && y) ? (use_rb ? rb[(x + 2) * rs] : prow[x + 2]) : NE;
        guess = (6 * (int64_t)N - 2 * (int64_t)NN + 7 * (int64_t)W + WW + NEE + 3 * (int64_t)NE + 8) / 16;
        break;
      }
      default: guess = 0; break;
    }
    const uint32_t tok = AnsGet(b, state, tab, ctx);
    const int32_t val = (int32_t)((int64_t)UnpackSigned(tok) * (int64_t)nd.b + nd.splitval + guess);
    row[x] = val;
    if (use_rb) rb[x * rs] = val;   // slot x held prev[x], which now lives in N
    prev9 = (int64_t)W + N - NW;
    const int32_t oldW = W;
    W = val;
    WW = x >= 1 ? oldW : val;
    if (y) { NW = N; N = NE; } else { NW = val; N = val; }
    if (++x == w) { x = 0; y++; row += stride; }
  }
}

// Resolves the MA tree of (chan, stream_id) on static properties only.  Returns true when that already ends in a
// leaf, i.e. neither the context nor the predictor of any sample depends on decoded samples or on the position.
template <bool kLds>
__device__ bool StaticLeaf(typename AS<kLds>::Tree tree, int chan, int stream_id, DevTreeNode* leaf) {
  int root = 0;
  DevTreeNode nd = NodeOf(tree[0]);
  while (nd.property == 0 || nd.property == 1) {
    const int v = nd.property == 0 ? chan : stream_id;
    root = v > nd.splitval ? nd.a : nd.b;
    nd = NodeOf(tree[root]);
  }
  *leaf = nd;
  return nd.property < 0;
}

// Split-phase decode of one channel by a whole wavefront (64 lanes, LDS variant only).  Precondition: StaticLeaf()
// holds with predictor Zero / W / N / Gradient and w <= 256.  Per batch of up to 64 rows:
//   phase A (lane 0): the serial part — ANS + hybrid-uint tokens -> residuals in LDS (`resid`, 64 x 256);
//   phase B (all lanes): lane r reconstructs row r, skewed by one sample per row, so that N comes from lane r-1's
//            previous step (one DPP shuffle), NW is the lane's own previous N and W its own previous value.
// `b` and `state` are meaningful in lane 0 only.  rb: LDS row (>= w ints) carrying the last row between batches.
__device__ void ModularChannelWave(LaneBits& b, uint32_t& state, const CodeTab<true>& tab, const DevTreeNode& leaf, int w, int h,
                                   int32_t* out_generic, int stride, JXL_LDS int32_t* rb, JXL_LDS int32_t* resid, int lane) {
  JXL_GLB int32_t* const out = G(out_generic);
  const uint32_t pred = leaf.a & 0xFF;
  const uint32_t cl = tab.cmap[leaf.a >> 8];
  const uint32_t cfg = tab.cfg[cl];
  const AS<true>::U64 abase = tab.alias + (cl << tab.log_alpha);
  const bool constant_token = (cfg & 0x1000) && ((cfg >> 16) & 0xFF) < (1u << (cfg & 0xF));
  const uint32_t const_res = (uint32_t)UnpackSigned((cfg >> 16) & 0xFF) * leaf.b + (uint32_t)leaf.splitval;
  for (int y0 = 0; y0 < h; y0 += 64) {
    const int nrows = min(64, h - y0);
    if (lane == 0 && !constant_token) {
      const int n = nrows * w;
      int x = 0, r = 0;
      for (int i = 0; i < n; i++) {
        const uint32_t sym = AnsSym<true>(b, state, abase, tab.log_alpha);
        resid[r * 256 + x] = (int32_t)((uint32_t)UnpackSigned(HybridTail(b, cfg, sym)) * leaf.b + (uint32_t)leaf.splitval);
        if (++x == w) { x = 0; r++; }
      }
    }
    __syncthreads();
    {
      const int y = y0 + lane;
      const bool row_active = lane < nrows;
      int32_t W = 0, N = 0, NW = 0, val = 0;
      const int steps = w + nrows - 1;
      for (int t = 0; t < steps; t++) {
        const int32_t from_up = __shfl_up(val, 1);   // lane r-1's value of the previous step = sample (x, y-1)
        const int x = t - lane;
        if (row_active && x >= 0 && x < w) {
          int32_t n_in;
          if (lane == 0) n_in = y ? rb[x] : 0;
          else n_in = from_up;
          if (x == 0) { W = y ? n_in : 0; N = W; NW = W; }
          else if (y) { NW = N; N = n_in; }
          else { NW = W; N = W; }
          uint32_t guess;
          if (pred == 0) guess = 0;
          else if (pred == 1) guess = (uint32_t)W;
          else if (pred == 2) guess = (uint32_t)N;
          else {
            const int64_t mn = W < N ? W : N, mx = W < N ? N : W, gr = (int64_t)W + N - NW;
            guess = (uint32_t)(int32_t)(gr < mn ? mn : (gr > mx ? mx : gr));
          }
          const uint32_t res = constant_token ? const_res : (uint32_t)resid[lane * 256 + x];
          val = (int32_t)(res + guess);
          out[(size_t)y * stride + x] = val;
          if (lane == nrows - 1) rb[x] = val;
          W = val;
        }
      }
    }
    __syncthreads();
  }
}

// If channel `chan` of stream `stream_id` resolves (on static properties only) to a leaf with the Zero
// predictor whose cluster is degenerate (one symbol, no extra bits), every sample equals a constant that
// can be written without touching the stream.  Returns true and the constant.
template <bool kLds>
__device__ bool ConstantChannel(const CodeTab<kLds>& tab, typename AS<kLds>::Tree tree, int chan, int stream_id, int32_t* value) {
  int root = 0;
  DevTreeNode nd = NodeOf(tree[0]);
  while (nd.property == 0 || nd.property == 1) {
    const int v = nd.property == 0 ? chan : stream_id;
    root = v > nd.splitval ? nd.a : nd.b;
    nd = NodeOf(tree[root]);
  }
  if (nd.property >= 0 || (nd.a & 0xFF) != 0) return false;
  const uint32_t c = tab.cfg[tab.cmap[nd.a >> 8]];
  if (!(c & 0x1000)) return false;
  const uint32_t sym = (c >> 16) & 0xFF;
  if (sym >= (1u << (c & 0xF))) return false;
  *value = (int32_t)((int64_t)UnpackSigned(sym) * (int64_t)nd.b + nd.splitval);
  return true;
}

}  // namespace

// ------------------------------------------------------------------ LF groups: one workgroup (one wavefront) per LF group
template <bool kLds>
__global__ __launch_bounds__(64) void lf_group_kernel(const DevImage* imgs, const SectionTask* tasks) {
  extern __shared__ __align__(16) uint8_t smem[];
  __shared__ uint32_t s_err;
  __shared__ uint32_t s_count;
  __shared__ uint32_t s_cov[256 * 8];   // coverage bitmap of the LF group's 256 x 256 cells
  const DevImage& im = imgs[tasks[blockIdx.x].image];
  const int g = tasks[blockIdx.x].first;
  const int tid = threadIdx.x;
  CodeTab<kLds> tab;
  typename AS<kLds>::Tree tree;
  RowBuf<kLds> rbuf;
  JXL_LDS int32_t* resid = nullptr;
  if constexpr (kLds) {
    JXL_LDS uint8_t* lds = (JXL_LDS uint8_t*)smem;
    size_t off = 0;
    rbuf.rb = (JXL_LDS int32_t*)lds; off += 256 * 4;
    resid = (JXL_LDS int32_t*)(lds + off); off += 64 * 256 * 4;
    JXL_LDS I4* st = (JXL_LDS I4*)(lds + off); off += (size_t)im.tree_size * sizeof(DevTreeNode);
    for (int i = tid; i < im.tree_size; i += 64) st[i] = ((const I4*)im.tree)[i];
    tree = st;
    StageCode(lds, off, im.mcode, tab, tid, 64);
    rbuf.rb_stride = 1;
    rbuf.rb_width = 256;
  } else {
    GlobalCode(im.mcode, tab);
    tree = (const I4*)im.tree;
    rbuf.rb = nullptr; rbuf.rb_stride = 1; rbuf.rb_width = 0;
  }
  if (tid == 0) s_err = 0;
  __syncthreads();
  const int gx = g % im.xlf, gy = g / im.xlf;
  const int bx0 = gx * kLfGroupBlocks, by0 = gy * kLfGroupBlocks;
  const int bw = min(kLfGroupBlocks, im.w8 - bx0), bh = min(kLfGroupBlocks, im.h8 - by0);
  const int tw = (bw + 7) / 8, th = (bh + 7) / 8;
  int32_t* scratch = im.binfo + (size_t)g * kBinfoInts;
  int32_t* s_x = scratch;
  int32_t* s_b = scratch + 1024;
  int32_t* s_info = scratch + 2048;
  int32_t* s_sharp = scratch + 2048 + 2 * 65536;
  const int sid_meta = 1 + 2 * im.nlf + g;
  LaneBits b;
  uint32_t state = 0;
  // Decodes one channel: wavefront split-phase when the stream allows it, else serially on lane 0.
  auto channel = [&](int chan, int sid, int w, int h, int32_t* out, int stride) {
    DevTreeNode leaf;
    bool wave = false;
    if constexpr (kLds) {
      if (w <= 256 && StaticLeaf<kLds>(tree, chan, sid, &leaf)) {
        const uint32_t p = leaf.a & 0xFF;
        wave = p == 0 || p == 1 || p == 2 || p == 5;
      }
      if (wave) {
        ModularChannelWave(b, state, tab, leaf, w, h, out, stride, rbuf.rb, resid, tid);
        return;
      }
    }
    if (tid == 0) ModularChannel<kLds>(b, state, tab, tree, chan, sid, w, h, out, stride, rbuf);
    __syncthreads();
  };
  const int lf_sec = im.single ? 0 : 1 + g;
  if (tid == 0) {
    b.Init(im.cs, im.cs_size, im.single ? im.lf_start_bits : im.sec_off[lf_sec] * 8);
    if (im.single && im.alpha_in_global) state = b.Read(32);
    s_err = 0;
  }
  __syncthreads();
  if (im.single && im.alpha_in_global) {
    // the alpha channel of a frame that fits one group is coded in the GlobalModular part of LfGlobal (stream 0)
    channel(0, 0, im.w, im.h, im.alpha32, im.w);
    if (tid == 0 && state != 0x130000u) s_err = kErrBitstream;
    __syncthreads();
    if (s_err) { if (tid == 0) SetError(im, s_err); return; }
  }
  if (tid == 0) {
    uint32_t err = 0;
    im.lf_extra[g] = (uint8_t)b.Read(2);
    if (b.Read(4) != 3) err |= kErrUnsupportedHeader;
    state = b.Read(32);
    s_err = err;
  }
  __syncthreads();
  if (s_err) { if (tid == 0) SetError(im, s_err); return; }
  {
    const int chan_of[3] = {1, 0, 2};
    for (int mc = 0; mc < 3; mc++) channel(mc, 1 + g, bw, bh, im.lfq[chan_of[mc]] + (size_t)by0 * im.w8 + bx0, im.w8);
  }
  if (tid == 0) {
    uint32_t err = 0, count = 1;
    if (state != 0x130000u) err |= kErrBitstream;
    if (!err) {
      count = b.Read(CeilLog2D((uint32_t)(bw * bh))) + 1;
      if (count > (uint32_t)(bw * bh)) err |= kErrBlockLayout;
      else if (b.Read(4) != 3) err |= kErrUnsupportedHeader;
      state = b.Read(32);
    }
    s_err = err;
    s_count = count;
  }
  __syncthreads();
  if (s_err) { if (tid == 0) SetError(im, s_err); return; }
  const uint32_t count = s_count;
  {
    // channels whose value is a stream-independent constant are filled by the whole wavefront
    int32_t cval;
    if (ConstantChannel<kLds>(tab, tree, 0, sid_meta, &cval)) { for (int i = tid; i < tw * th; i += 64) s_x[i] = cval; }
    else channel(0, sid_meta, tw, th, s_x, tw);
    if (ConstantChannel<kLds>(tab, tree, 1, sid_meta, &cval)) { for (int i = tid; i < tw * th; i += 64) s_b[i] = cval; }
    else channel(1, sid_meta, tw, th, s_b, tw);
    if (ConstantChannel<kLds>(tab, tree, 2, sid_meta, &cval)) { for (uint32_t i = tid; i < 2 * count; i += 64) s_info[i] = cval; }
    else channel(2, sid_meta, (int)count, 2, s_info, (int)count);
    if (ConstantChannel<kLds>(tab, tree, 3, sid_meta, &cval)) { for (int i = tid; i < bw * bh; i += 64) s_sharp[i] = cval; }
    else channel(3, sid_meta, bw, bh, s_sharp, bw);
  }
  if (tid == 0) {
    uint32_t err = 0;
    const uint64_t start_bits = im.single ? im.lf_start_bits : im.sec_off[lf_sec] * 8;
    if (state != 0x130000u || start_bits + b.Consumed() > (im.sec_off[lf_sec] + im.sec_size[lf_sec]) * 8) err |= kErrBitstream;
    if (im.single) im.lf_end_bits[0] = start_bits + b.Consumed();
    s_err = err;
  }
  __syncthreads();
  if (s_err) { if (tid == 0) SetError(im, s_err); return; }
  // chroma-from-luma maps and sharpness: parallel copies with range checks
  uint32_t err = 0;
  const int tx0 = bx0 / 8, ty0 = by0 / 8;
  for (int i = tid; i < tw * th; i += 64) {
    const int x = i % tw, y = i / tw;
    const int vx = s_x[i], vb = s_b[i];
    if (vx < -128 || vx > 127 || vb < -128 || vb > 127) err |= kErrRange;
    im.ytox[(size_t)(ty0 + y) * im.wt + tx0 + x] = (int8_t)vx;
    im.ytob[(size_t)(ty0 + y) * im.wt + tx0 + x] = (int8_t)vb;
  }
  for (int i = tid; i < bw * bh; i += 64) {
    const int x = i % bw, y = i / bw;
    int sh = s_sharp[i];
    if (sh < 0 || sh > 7) { err |= kErrRange; sh = 0; }
    im.sharp[(size_t)(by0 + y) * im.w8 + bx0 + x] = (uint8_t)sh;
  }
  __syncthreads();   // s_sharp has been consumed: its scratch is reused for the block positions
  // ---- varblock placement.  By definition serial (each entry goes to the first cell, in raster order, not yet covered),
  // but only per BLOCK: lane 0 walks a coverage bitmap in LDS (one 32-bit word per 32 cells) and records positions; the
  // per-cell fan-out (cellinfo / raw quant of every covered cell) is then done by all lanes.
  int32_t* s_pos = s_sharp;
  if (tid == 0) s_count = 0;
  for (int i = tid; i < 256 * 8; i += 64) s_cov[i] = 0;
  __syncthreads();
  if (tid == 0) {
    uint32_t num = 0;
    const int words = (bw + 31) >> 5;
    for (int y = 0; y < bh && !(err & kErrBlockLayout); y++) {
      for (int wi = 0; wi < words; wi++) {
        for (;;) {
          uint32_t freebits = ~s_cov[y * 8 + wi];
          const int lim = min(32, bw - wi * 32);
          if (lim < 32) freebits &= (1u << lim) - 1;
          if (!freebits) break;
          const int xb = __ffs(freebits) - 1;
          const int x = wi * 32 + xb;
          if (num >= count) { err |= kErrBlockLayout; break; }
          const int s = s_info[num];
          const int q = 1 + s_info[count + num];
          if (s < 0 || s >= kNumStrategies || q < 1 || q > 256) { err |= kErrBlockLayout; break; }
          const int lcx = d_log2cx[s], lcy = d_log2cy[s], cx = 1 << lcx, cy = 1 << lcy;
          if (x + cx > bw || y + cy > bh || xb + cx > 32 || (y & 31) + cy > 32) { err |= kErrBlockLayout; break; }
          const uint32_t mask = (cx == 32 ? 0xFFFFFFFFu : ((1u << cx) - 1)) << xb;
          uint32_t clash = 0;
          for (int iy = 0; iy < cy; iy++) { clash |= s_cov[(y + iy) * 8 + wi] & mask; s_cov[(y + iy) * 8 + wi] |= mask; }
          if (clash) { err |= kErrBlockLayout; break; }
          s_pos[num] = x | y << 8;
          num++;
        }
        if (err & kErrBlockLayout) break;
      }
    }
    if (num != count) err |= kErrBlockLayout;
    s_count = num;
  }
  __syncthreads();
  const uint32_t placed = s_count;
  for (uint32_t i = tid; i < placed; i += 64) {
    const int x = s_pos[i] & 0xFF, y = s_pos[i] >> 8;
    const int s = s_info[i], q = 1 + s_info[count + i];
    const int lcx = d_log2cx[s], lcy = d_log2cy[s], cx = 1 << lcx, cy = 1 << lcy;
    const size_t cell = (size_t)(by0 + y) * im.w8 + bx0 + x;
    for (int iy = 0; iy < cy; iy++)
      for (int ix = 0; ix < cx; ix++) {
        const size_t cc = cell + (size_t)iy * im.w8 + ix;
        im.cellinfo[cc] = (uint32_t)s | ix << 8 | iy << 13 | lcx << 18 | lcy << 21 | 1u << 31;
        im.rawq[cc] = (uint16_t)q;
      }
  }
  if (err) SetError(im, err);
}

// ------------------------------------------------------------------ HF coefficients
// One lane per group section (lane_stride spreads sections over wavefronts).  The loop decodes exactly one
// token per iteration: the number-of-nonzeros token of a (block, channel) or one coefficient token.
template <bool kLds>
__global__ __launch_bounds__(256) void hf_decode_kernel(const DevImage* imgs, const SectionTask* tasks, int lane_stride,
                                                        const uint16_t* natural_orders_small) {
  extern __shared__ __align__(16) uint8_t smem[];
  const SectionTask task = tasks[blockIdx.x];
  const DevImage& im = imgs[task.image];
  CodeTab<kLds> tab;
  typename AS<kLds>::U16 lds_orders;
  typename AS<kLds>::U8 nnz_tab;
  if constexpr (kLds) {
    JXL_LDS uint8_t* lds = (JXL_LDS uint8_t*)smem;
    size_t off = StageCode(lds, 0, im.acode, tab, threadIdx.x, blockDim.x);
    off = (off + 1) & ~(size_t)1;
    JXL_LDS uint16_t* so = (JXL_LDS uint16_t*)(lds + off); off += 8448 * 2;
    for (int i = threadIdx.x; i < 8448; i += blockDim.x) so[i] = natural_orders_small[i];
    JXL_LDS uint8_t* sn = lds + off; off += 64;
    if (threadIdx.x < 64) sn[threadIdx.x] = d_nnz_ctx[threadIdx.x];
    lds_orders = so;
    nnz_tab = sn;
    __syncthreads();
  } else {
    GlobalCode(im.acode, tab);
    lds_orders = natural_orders_small;
    nnz_tab = d_nnz_ctx;
  }
  // Active lanes are the FIRST 64/lane_stride lanes of every wavefront: a wave64 whose upper 32 lanes are idle issues
  // each vector instruction in one pass instead of two.
  const int per_wave = 64 / lane_stride;
  if ((int)(threadIdx.x & 63) >= per_wave) return;
  const int si = (threadIdx.x >> 6) * per_wave + (threadIdx.x & 63);
  if (si >= task.count) return;
  const int g = task.first + si;
  const int gx = g % im.xg, gy = g / im.xg;
  const int bx0 = gx * kGroupBlocks, by0 = gy * kGroupBlocks;
  const int bw = min(kGroupBlocks, im.w8 - bx0), bh = min(kGroupBlocks, im.h8 - by0);
  const int sec = im.single ? 0 : 2 + im.nlf + g;
  const uint64_t sec_bits = im.single ? im.hf_start_bits : im.sec_off[sec] * 8;
  LaneBits b;
  b.Init(im.cs, im.cs_size, sec_bits);
  const uint32_t preset = b.Read(CeilLog2D((uint32_t)im.num_presets));
  const uint32_t nbc = im.num_block_ctx;
  const uint32_t ctx_offset = preset * nbc * 495;
  uint32_t err = preset >= (uint32_t)im.num_presets ? (uint32_t)kErrBitstream : 0u;
  uint32_t state = b.Read(32);
  JXL_GLB uint8_t* const nz = G(im.nzmap) + (size_t)g * 3 * 1024;
  const JXL_GLB uint32_t* const cellinfo = G(im.cellinfo);
  const JXL_GLB uint16_t* const rawq = G(im.rawq);
  const int wp = im.wp;
  const int n_qf = im.n_qf;
  const bool use_staged_orders = !im.custom_orders;
  // cursor over 8x8 cells of the group
  int bx = -1, by = 0;
  // current block
  uint32_t lcx = 0, lcy = 0, log2c = 0, covered = 1, size = 64, ord = 0, lng_log2 = 3, qf_idx = 0;
  bool transposed = true;
  size_t px0 = 0;
  // current (block, channel)
  int ci = 3;
  uint32_t nzeros = 0, k = 0, prev = 0, histo = 0, block_ctx = 0;
  const JXL_GLB uint16_t* gorder = nullptr;   // order table in global memory ...
  uint32_t lorder = 0;                // ... or offset of a small natural order staged with the tables
  bool order_staged = false;
  JXL_GLB int32_t* plane = nullptr;
  bool want_nz = true;
  while (!err) {
    if (want_nz && ci >= 3) {
      // advance to the next varblock (top-left cell) of this group
      bool found = false;
      for (;;) {
        if (++bx >= bw) { bx = 0; by++; }
        if (by >= bh) break;
        const uint32_t info = cellinfo[(size_t)(by0 + by) * im.w8 + bx0 + bx];
        if ((info & 0x8003FF00u) != 0x80000000u) continue;
        const uint32_t s = info & 0xFF;
        lcx = (info >> 18) & 7; lcy = (info >> 21) & 7;
        log2c = lcx + lcy; covered = 1u << log2c; size = covered << 6;
        ord = d_order_bucket[s];
        lng_log2 = 3 + max(lcx, lcy);
        transposed = !IsSpecialS(s) && lcy >= lcx;
        const uint32_t rq = rawq[(size_t)(by0 + by) * im.w8 + bx0 + bx];
        qf_idx = 0;
        for (int i = 0; i < n_qf; i++) qf_idx += rq > im.qf_thr[i];
        px0 = (size_t)(by0 + by) * 8 * wp + (size_t)(bx0 + bx) * 8;
        found = true;
        break;
      }
      if (!found) break;
      ci = 0;
    }
    const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
    uint32_t ctx;
    if (want_nz) {
      const JXL_GLB uint8_t* row = nz + c * 1024 + by * 32;
      uint32_t predicted;
      if (bx == 0) predicted = by == 0 ? 32 : row[-32];
      else if (by == 0) predicted = row[bx - 1];
      else predicted = ((uint32_t)row[-32 + bx] + row[bx - 1] + 1) >> 1;
      const uint32_t cprime = c < 2 ? (c ^ 1) : 2;
      block_ctx = im.block_ctx_map[(cprime * kNumOrders + ord) * (n_qf + 1) + qf_idx];
      uint32_t nzc = predicted >= 64 ? 64 : predicted;
      nzc = nzc < 8 ? nzc : 4 + nzc / 2;
      ctx = ctx_offset + nzc * nbc + block_ctx;
    } else {
      const uint32_t nzl = (nzeros + covered - 1) >> log2c;
      const uint32_t ks = k >> log2c;
      const uint32_t fctx = ks < 16 ? ks - 1 : (ks < 32 ? 15 + ((ks - 16) >> 1) : 23 + ((ks - 32) >> 2));
      ctx = histo + ((uint32_t)nnz_tab[nzl] + fctx) * 2 + prev;
    }
    const uint32_t u = AnsGet(b, state, tab, ctx);
    if (want_nz) {
      nzeros = u;
      if (nzeros + covered > size) { err |= kErrBitstream; break; }
      const uint8_t fill = (uint8_t)((nzeros + covered - 1) >> log2c);
      JXL_GLB uint8_t* row = nz + c * 1024 + by * 32;
      for (uint32_t iy = 0; iy < (1u << lcy); iy++)
        for (uint32_t ix = 0; ix < (1u << lcx); ix++) row[iy * 32 + bx + ix] = fill;
      if (nzeros) {
        histo = ctx_offset + nbc * 37 + 458 * block_ctx;
        order_staged = use_staged_orders && ord <= 8;
        lorder = d_order_lds_off[ord <= 8 ? ord : 0];
        gorder = G(im.order[ord * 3 + c]);
        plane = G(im.coef[c]) + px0;
        prev = nzeros > size / 16 ? 0 : 1;
        k = covered;
        want_nz = false;
      } else {
        ci++;
      }
    } else {
      if (u) {
        const uint32_t p = order_staged ? (uint32_t)lds_orders[lorder + k] : (uint32_t)gorder[k];
        const uint32_t r = p >> lng_log2, cc = p & ((1u << lng_log2) - 1);
        const uint32_t ky = transposed ? cc : r, kx = transposed ? r : cc;
        plane[(size_t)ky * wp + kx] = UnpackSigned(u);
        prev = 1;
        if (--nzeros == 0) { want_nz = true; ci++; }
      } else {
        prev = 0;
      }
      if (++k >= size && nzeros != 0) { err |= kErrBitstream; break; }
    }
  }
  if (!err && state != 0x130000u) err |= kErrBitstream;
  const uint64_t used = b.Consumed();
  if (!err && sec_bits + used > (im.sec_off[sec] + im.sec_size[sec]) * 8) err |= kErrBitstream;
  im.grp_bitpos[g] = err ? ~(uint64_t)0 : sec_bits + used;
  if (err) SetError(im, err);
}

// ------------------------------------------------------------------ alpha (Modular stream after the HF tokens)
// Two mappings: lane_stride == 64 -> 64-thread workgroup = one section, split-phase decode by the whole wavefront;
// otherwise 256-thread workgroups (one wavefront per SIMD of a CU), the first 64/lane_stride lanes of every
// wavefront own one section each.
template <bool kLds>
__global__ __launch_bounds__(256) void alpha_kernel(const DevImage* imgs, const SectionTask* tasks, int lane_stride) {
  extern __shared__ __align__(16) uint8_t smem[];
  const SectionTask task = tasks[blockIdx.x];
  const DevImage& im = imgs[task.image];
  if (!im.has_alpha || im.alpha_in_global) return;
  CodeTab<kLds> tab;
  typename AS<kLds>::Tree tree;
  RowBuf<kLds> rbuf;
  JXL_LDS int32_t* resid = nullptr;
  const int per_wave = 64 / lane_stride;
  const int slots = lane_stride == 64 ? 1 : 4 * per_wave;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = wave * per_wave + (lane < per_wave ? lane : 0);
  if constexpr (kLds) {
    JXL_LDS uint8_t* lds = (JXL_LDS uint8_t*)smem;
    size_t off = 0;
    rbuf.rb = (JXL_LDS int32_t*)lds + (lane_stride == 64 ? 0 : slot); off += (size_t)slots * 256 * 4;
    rbuf.rb_stride = slots;
    rbuf.rb_width = 256;
    if (lane_stride == 64) { resid = (JXL_LDS int32_t*)(lds + off); off += 64 * 256 * 4; }
    JXL_LDS I4* st = (JXL_LDS I4*)(lds + off); off += (size_t)im.tree_size * sizeof(DevTreeNode);
    for (int i = threadIdx.x; i < im.tree_size; i += blockDim.x) st[i] = ((const I4*)im.tree)[i];
    tree = st;
    StageCode(lds, off, im.mcode, tab, threadIdx.x, blockDim.x);
    __syncthreads();
  } else {
    GlobalCode(im.mcode, tab);
    tree = (const I4*)im.tree;
    rbuf.rb = nullptr; rbuf.rb_stride = 1; rbuf.rb_width = 0;
  }
  if constexpr (kLds) {
    if (lane_stride == 64) {
      // one section per wavefront (blockDim == 64): split-phase decode by all 64 lanes when the stream allows it
      __shared__ uint32_t s_fail;
      const int g = task.first;
      const uint64_t start = im.grp_bitpos[g];
      if (start == ~(uint64_t)0) return;
      const int gx = g % im.xg, gy = g / im.xg;
      const int sec = 2 + im.nlf + g;
      const int x0 = gx * kGroupDim, y0 = gy * kGroupDim;
      const int gw = min(kGroupDim, im.w - x0), gh = min(kGroupDim, im.h - y0);
      const int sid = 1 + 3 * im.nlf + kNumQuantTables + g;
      DevTreeNode leaf;
      bool wavepath = StaticLeaf<true>(tree, 0, sid, &leaf);
      if (wavepath) { const uint32_t p = leaf.a & 0xFF; wavepath = p == 0 || p == 1 || p == 2 || p == 5; }
      LaneBits b;
      uint32_t state = 0;
      if (threadIdx.x == 0) {
        b.Init(im.cs, im.cs_size, start);
        s_fail = b.Read(4) != 3 ? (uint32_t)kErrUnsupportedHeader : 0u;
        state = b.Read(32);
      }
      __syncthreads();
      if (s_fail) { if (threadIdx.x == 0) SetError(im, s_fail); return; }
      int32_t* out = im.alpha32 + (size_t)y0 * im.w + x0;
      if (wavepath) {
        ModularChannelWave(b, state, tab, leaf, gw, gh, out, im.w, (JXL_LDS int32_t*)smem, resid, threadIdx.x);
      } else if (threadIdx.x == 0) {
        ModularChannel<true>(b, state, tab, tree, 0, sid, gw, gh, out, im.w, rbuf);
      }
      if (threadIdx.x == 0) {
        uint32_t err = 0;
        if (state != 0x130000u) err |= kErrBitstream;
        if (start + b.Consumed() > (im.sec_off[sec] + im.sec_size[sec]) * 8) err |= kErrBitstream;
        if (err) SetError(im, err);
      }
      return;
    }
  }
  if (lane >= per_wave) return;   // the first per_wave lanes of every wavefront own one section each
  const int si = lane_stride == 64 ? 0 : slot;
  if (si >= task.count) return;
  const int g = task.first + si;
  const uint64_t start = im.grp_bitpos[g];
  if (start == ~(uint64_t)0) return;   // the HF decoder already reported the failure
  const int gx = g % im.xg, gy = g / im.xg;
  const int sec = 2 + im.nlf + g;
  LaneBits b;
  b.Init(im.cs, im.cs_size, start);
  uint32_t err = 0;
  if (b.Read(4) != 3) err |= kErrUnsupportedHeader;
  else {
    uint32_t state = b.Read(32);
    const int x0 = gx * kGroupDim, y0 = gy * kGroupDim;
    const int gw = min(kGroupDim, im.w - x0), gh = min(kGroupDim, im.h - y0);
    const int sid = 1 + 3 * im.nlf + kNumQuantTables + g;
    ModularChannel(b, state, tab, tree, 0, sid, gw, gh, im.alpha32 + (size_t)y0 * im.w + x0, im.w, rbuf);
    if (state != 0x130000u) err |= kErrBitstream;
    if (start + b.Consumed() > (im.sec_off[sec] + im.sec_size[sec]) * 8) err |= kErrBitstream;
  }
  if (err) SetError(im, err);
}

// ------------------------------------------------------------------ launch wrappers
static void RaiseLds(const void* fn, size_t bytes) {
  if (bytes > 48 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

void LaunchLfGroups(const DevImage* imgs, const SectionTask* tasks, int ntasks, size_t lds_bytes, hipStream_t s) {
  if (ntasks <= 0) return;
  if (lds_bytes) {
    RaiseLds((const void*)lf_group_kernel<true>, lds_bytes);
    hipLaunchKernelGGL(lf_group_kernel<true>, dim3(ntasks), dim3(64), lds_bytes, s, imgs, tasks);
  } else {
    hipLaunchKernelGGL(lf_group_kernel<false>, dim3(ntasks), dim3(64), 0, s, imgs, tasks);
  }
}

void LaunchHfDecode(const DevImage* imgs, const SectionTask* tasks, int nwg, int lane_stride, size_t lds_bytes,
                    const uint16_t* natural_orders_small, hipStream_t s) {
  if (nwg <= 0) return;
  if (lds_bytes) {
    RaiseLds((const void*)hf_decode_kernel<true>, lds_bytes);
    hipLaunchKernelGGL(hf_decode_kernel<true>, dim3(nwg), dim3(256), lds_bytes, s, imgs, tasks, lane_stride, natural_orders_small);
  } else {
    hipLaunchKernelGGL(hf_decode_kernel<false>, dim3(nwg), dim3(256), 0, s, imgs, tasks, lane_stride, natural_orders_small);
  }
}

void LaunchAlpha(const DevImage* imgs, const SectionTask* tasks, int nwg, int lane_stride, size_t lds_bytes, hipStream_t s) {
  if (nwg <= 0) return;
  if (lds_bytes) {
    RaiseLds((const void*)alpha_kernel<true>, lds_bytes);
    hipLaunchKernelGGL(alpha_kernel<true>, dim3(nwg), dim3(lane_stride == 64 ? 64 : 256), lds_bytes, s, imgs, tasks, lane_stride);
  } else {
    hipLaunchKernelGGL(alpha_kernel<false>, dim3(nwg), dim3(lane_stride == 64 ? 64 : 256), 0, s, imgs, tasks, lane_stride);
  }
}

}  // namespace jxlhip
