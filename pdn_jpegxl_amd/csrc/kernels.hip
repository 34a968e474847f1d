// HIP kernels (gfx950) of the JPEG XL VarDCT decode path.
//
// Stage map (DESIGN.md has the data layout and per-kernel roofline):
//   lf_group_kernel      LF coefficients + HF metadata (Modular, ANS + MA tree) and varblock placement
//   pass_group_kernel    per 256x256 group: HF coefficient entropy decode, then the alpha Modular stream
//   lf_dequant / lf_smooth / cell_sigma / alpha_to_u8 / dequant / llf / idct_v / idct_h / idct_special
//   gaborish / epf<stage> / xyb_to_out
//
// The entropy decoders are written as a per-lane state machine: every lane owns one section
// (bit reader + ANS state).  `lane_stride` selects the mapping: 64 = one section per wavefront
// (latency-optimal for a single image), 1 = one section per lane (throughput-optimal for batches).
#include <hip/hip_runtime.h>
#include "dev_types.h"
#include "kernels.h"

namespace jxlhip {

__constant__ uint8_t c_order_bucket[kNumStrategies] = {0, 1, 1, 1, 2, 3, 4, 4, 5, 5, 6, 6, 1, 1, 1, 1, 1, 1, 7, 8, 8, 9, 10, 10, 11, 12, 12};
__constant__ uint8_t c_quant_table[kNumStrategies] = {0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16};
__constant__ uint8_t c_log2cx[kNumStrategies] = {0, 0, 0, 0, 1, 2, 0, 1, 0, 2, 1, 2, 0, 0, 0, 0, 0, 0, 3, 2, 3, 4, 3, 4, 5, 4, 5};
__constant__ uint8_t c_log2cy[kNumStrategies] = {0, 0, 0, 0, 1, 2, 1, 0, 2, 0, 2, 1, 0, 0, 0, 0, 0, 0, 3, 3, 2, 4, 4, 3, 5, 5, 4};
__constant__ uint8_t c_nnz_ctx[64] = {0,   0,   31,  62,  62,  93,  93,  93,  93,  123, 123, 123, 123, 152, 152, 152, 152, 152, 152, 152, 152, 180,
                                      180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206,
                                      206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206};

__device__ __forceinline__ bool IsSpecial(uint32_t s) { return (s >= 1 && s <= 3) || (s >= 12 && s <= 17); }
__device__ __forceinline__ int DMirror(int v, int n) {
  while (v < 0 || v >= n) v = v < 0 ? -v - 1 : 2 * n - 1 - v;
  return v;
}
__device__ __forceinline__ int DCeilLog2(uint32_t x) { return x <= 1 ? 0 : 32 - __clz(x - 1); }

// ------------------------------------------------------------------ per-lane bit reader
struct LaneBits {
  const uint32_t* w;
  uint32_t idx, nwords;
  uint64_t buf;
  int n;
  int skip;
  __device__ void Init(const uint8_t* cs, uint64_t cs_size, uint64_t byte_off) {
    uintptr_t addr = (uintptr_t)(cs + byte_off);
    uintptr_t al = addr & ~(uintptr_t)3;
    w = (const uint32_t*)al;
    nwords = (uint32_t)(((uintptr_t)(cs + cs_size) + 3 - al) >> 2);
    idx = 0; buf = 0; n = 0;
    skip = (int)(addr - al) * 8;
    Refill();
    buf >>= skip;
    n -= skip;
  }
  __device__ __forceinline__ void Refill() {
    if (n <= 32) {
      uint32_t v = idx < nwords ? w[idx] : 0u;
      idx++;
      buf |= (uint64_t)v << n;
      n += 32;
    }
  }
  __device__ __forceinline__ uint32_t Read(int k) {   // k <= 32
    Refill();
    uint32_t v = (uint32_t)(buf & (((uint64_t)1 << k) - 1));
    buf >>= k;
    n -= k;
    return v;
  }
  __device__ uint64_t Consumed() const { return (uint64_t)idx * 32 - n - skip; }
};

// ------------------------------------------------------------------ entropy-coded symbol source
template <bool kLdsTables>
struct Ans {
  LaneBits* b;
  const uint8_t* cmap;
  const uint32_t* cfg;
  const uint64_t* alias;
  uint32_t log_alpha;
  uint32_t state;
  __device__ __forceinline__ void Start() { state = b->Read(32); }
  __device__ __forceinline__ uint32_t Get(uint32_t ctx) {
    const uint32_t cl = cmap[ctx];
    const uint32_t le = 12 - log_alpha;
    const uint32_t res = state & 0xFFF, i = res >> le, pos = res & ((1u << le) - 1);
    const uint64_t e = alias[(cl << log_alpha) | i];
    const uint32_t x = (uint32_t)e, y = (uint32_t)(e >> 32);
    const bool g = pos >= (x & 0xFF);
    const uint32_t sym = g ? ((x >> 8) & 0xFF) : i;
    const uint32_t off = g ? (y & 0xFFFF) + pos : pos;
    const uint32_t freq = g ? ((x >> 16) ^ (y >> 16)) : (x >> 16);
    state = freq * (state >> 12) + off;
    if (state < 65536u) state = (state << 16) | b->Read(16);
    // hybrid uint
    const uint32_t c = cfg[cl];
    const uint32_t se = c & 0xFF, split = 1u << se;
    if (sym < split) return sym;
    const uint32_t msb = (c >> 8) & 0xFF, lsb = (c >> 16) & 0xFF;
    const uint32_t nb = se - (msb + lsb) + ((sym - split) >> (msb + lsb));
    const uint32_t low = sym & ((1u << lsb) - 1);
    const uint32_t t = sym >> lsb;
    const uint32_t bits = b->Read(nb > 32 ? 32 : nb);
    const uint32_t hi = (1u << msb) | (t & ((1u << msb) - 1));
    return (uint32_t)(((((uint64_t)hi << nb) | bits) << lsb) | low);
  }
  __device__ __forceinline__ bool Final() const { return state == 0x130000u; }
};

__device__ __forceinline__ int32_t UnpackS(uint32_t u) { return (int32_t)(u >> 1) ^ -(int32_t)(u & 1); }

// ------------------------------------------------------------------ Modular channel (MA tree + predictors)
// Decodes one channel (w x h) into `out` (row stride `stride`).  Supports every property 0..14 and
// every predictor except the weighted one; the host rejects trees that need more.
template <bool kLds>
__device__ void ModularChannel(Ans<kLds>& ans, const DevTreeNode* tree, int chan, int stream_id, int w, int h, int32_t* out,
                               int stride) {
  int root = 0;
  for (;;) {
    const DevTreeNode nd = tree[root];
    if (nd.property != 0 && nd.property != 1) break;
    int v = nd.property == 0 ? chan : stream_id;
    root = v > nd.splitval ? nd.a : nd.b;
  }
  for (int y = 0; y < h; y++) {
    int32_t* row = out + (size_t)y * stride;
    const int32_t* prow = row - stride;
    const int32_t* pprow = prow - stride;
    int rroot = root;
    for (;;) {
      const DevTreeNode nd = tree[rroot];
      if (nd.property < 0 || nd.property > 2) break;
      int v = nd.property == 0 ? chan : (nd.property == 1 ? stream_id : y);
      rroot = v > nd.splitval ? nd.a : nd.b;
    }
    int64_t prev9 = 0;
    // neighbours live in registers; only the previous row is (re)loaded from memory
    int32_t W = y ? prow[0] : 0, N = W, NW = W, WW = W, NE;
    for (int x = 0; x < w; x++) {
      NE = (x + 1 < w && y) ? prow[x + 1] : N;
      int node = rroot;
      DevTreeNode nd = tree[node];
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
          case 13: { int32_t NN = y > 1 ? pprow[x] : N; p = (int64_t)N - NN; break; }
          case 14: p = (int64_t)W - WW; break;
          default: p = 0; break;
        }
        node = p > nd.splitval ? nd.a : nd.b;
        nd = tree[node];
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
          int64_t mn = W < N ? W : N, mx = W < N ? N : W, gr = (int64_t)W + N - NW;
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
          int32_t NN = y > 1 ? pprow[x] : N;
          int32_t NEE = (x + 2 < w && y) ? prow[x + 2] : NE;
          guess = (6 * (int64_t)N - 2 * (int64_t)NN + 7 * (int64_t)W + WW + NEE + 3 * (int64_t)NE + 8) / 16;
          break;
        }
        default: guess = 0; break;
      }
      const uint32_t tok = ans.Get(ctx);
      const int32_t val = (int32_t)((int64_t)UnpackS(tok) * (int64_t)nd.b + nd.splitval + guess);
      row[x] = val;
      prev9 = (int64_t)W + N - NW;
      // slide the window to x + 1
      const int32_t oldW = W;
      W = val;
      WW = x >= 1 ? oldW : val;
      if (y) { NW = N; N = NE; } else { NW = val; N = val; }
    }
  }
}

__device__ __forceinline__ void SetError(const DevImage& im, uint32_t bits) { atomicOr(im.status, bits); }

// ------------------------------------------------------------------ LF groups
__global__ __launch_bounds__(64) void lf_group_kernel(const DevImage* imgs, const SectionTask* tasks, int ntasks) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntasks) return;
  const DevImage& im = imgs[tasks[t].image];
  const int g = tasks[t].first;
  const int gx = g % im.xlf, gy = g / im.xlf;
  const int bx0 = gx * kLfGroupBlocks, by0 = gy * kLfGroupBlocks;
  const int bw = min(kLfGroupBlocks, im.w8 - bx0), bh = min(kLfGroupBlocks, im.h8 - by0);
  LaneBits b;
  b.Init(im.cs, im.cs_size, im.sec_off[1 + g]);
  Ans<false> ans;
  ans.b = &b; ans.cmap = im.mcode.ctx_map; ans.cfg = im.mcode.cfg; ans.alias = im.mcode.alias; ans.log_alpha = im.mcode.log_alpha;
  im.lf_extra[g] = (uint8_t)b.Read(2);
  if (b.Read(4) != 3) { SetError(im, kErrUnsupportedHeader); return; }
  ans.Start();
  const int chan_of[3] = {1, 0, 2};
  for (int mc = 0; mc < 3; mc++)
    ModularChannel(ans, im.tree, mc, 1 + g, bw, bh, im.lfq[chan_of[mc]] + (size_t)by0 * im.w8 + bx0, im.w8);
  if (!ans.Final()) { SetError(im, kErrBitstream); return; }
  // HF metadata
  const uint32_t count = b.Read(DCeilLog2((uint32_t)(bw * bh))) + 1;
  if (count > (uint32_t)(bw * bh)) { SetError(im, kErrBlockLayout); return; }
  if (b.Read(4) != 3) { SetError(im, kErrUnsupportedHeader); return; }
  ans.Start();
  const int tw = (bw + 7) / 8, th = (bh + 7) / 8;
  int32_t* scratch = im.binfo + (size_t)g * kBinfoInts;
  int32_t* s_x = scratch;
  int32_t* s_b = scratch + 1024;
  int32_t* s_info = scratch + 2048;
  int32_t* s_sharp = scratch + 2048 + 2 * 65536;
  const int sid = 1 + 2 * im.nlf + g;
  ModularChannel(ans, im.tree, 0, sid, tw, th, s_x, tw);
  ModularChannel(ans, im.tree, 1, sid, tw, th, s_b, tw);
  ModularChannel(ans, im.tree, 2, sid, (int)count, 2, s_info, (int)count);
  ModularChannel(ans, im.tree, 3, sid, bw, bh, s_sharp, bw);
  if (!ans.Final() || b.Consumed() > (uint64_t)im.sec_size[1 + g] * 8) { SetError(im, kErrBitstream); return; }
  const int tx0 = bx0 / 8, ty0 = by0 / 8;
  uint32_t err = 0;
  for (int y = 0; y < th; y++)
    for (int x = 0; x < tw; x++) {
      int vx = s_x[y * tw + x], vb = s_b[y * tw + x];
      if (vx < -128 || vx > 127 || vb < -128 || vb > 127) err |= kErrRange;
      im.ytox[(size_t)(ty0 + y) * im.wt + tx0 + x] = (int8_t)vx;
      im.ytob[(size_t)(ty0 + y) * im.wt + tx0 + x] = (int8_t)vb;
    }
  uint32_t num = 0;
  for (int y = 0; y < bh; y++)
    for (int x = 0; x < bw; x++) {
      const size_t cell = (size_t)(by0 + y) * im.w8 + bx0 + x;
      int sh = s_sharp[y * bw + x];
      if (sh < 0 || sh > 7) { err |= kErrRange; sh = 0; }
      im.sharp[cell] = (uint8_t)sh;
      if (im.cellinfo[cell] >> 31) continue;
      if (num >= count) { err |= kErrBlockLayout; continue; }
      const int s = s_info[num];
      const int q = 1 + s_info[count + num];
      num++;
      if (s < 0 || s >= kNumStrategies || q < 1 || q > 256) { err |= kErrBlockLayout; continue; }
      const int lcx = c_log2cx[s], lcy = c_log2cy[s], cx = 1 << lcx, cy = 1 << lcy;
      if (x + cx > bw || y + cy > bh || (x & 31) + cx > 32 || (y & 31) + cy > 32) { err |= kErrBlockLayout; continue; }
      for (int iy = 0; iy < cy; iy++)
        for (int ix = 0; ix < cx; ix++) {
          const size_t cc = cell + (size_t)iy * im.w8 + ix;
          if (im.cellinfo[cc] >> 31) err |= kErrBlockLayout;
          im.cellinfo[cc] = (uint32_t)s | ix << 8 | iy << 13 | lcx << 18 | lcy << 21 | 1u << 31;
          im.rawq[cc] = (uint16_t)q;
        }
    }
  if (num != count) err |= kErrBlockLayout;
  if (err) SetError(im, err);
}

// ------------------------------------------------------------------ pass groups (HF coefficients + alpha)
template <bool kLds>
__global__ __launch_bounds__(256) void pass_group_kernel(const DevImage* imgs, const SectionTask* tasks, int lane_stride) {
  extern __shared__ __align__(16) uint8_t smem[];
  const SectionTask task = tasks[blockIdx.x];
  const DevImage& im = imgs[task.image];
  const uint8_t* a_cmap = im.acode.ctx_map;
  const uint32_t* a_cfg = im.acode.cfg;
  const uint64_t* a_alias = im.acode.alias;
  const uint8_t* m_cmap = im.mcode.ctx_map;
  const uint32_t* m_cfg = im.mcode.cfg;
  const uint64_t* m_alias = im.mcode.alias;
  const DevTreeNode* tree = im.tree;
  if (kLds) {
    // carve: alias tables (8-byte aligned) first, then trees (16), cfg (4), ctx maps (1)
    size_t off = 0;
    uint64_t* sa = (uint64_t*)(smem + off); off += (size_t)(im.acode.num_clusters << im.acode.log_alpha) * 8;
    uint64_t* sm = (uint64_t*)(smem + off); off += (size_t)(im.mcode.num_clusters << im.mcode.log_alpha) * 8;
    off = (off + 15) & ~(size_t)15;
    DevTreeNode* st = (DevTreeNode*)(smem + off); off += (size_t)im.tree_size * sizeof(DevTreeNode);
    uint32_t* sac = (uint32_t*)(smem + off); off += (size_t)im.acode.num_clusters * 4;
    uint32_t* smc = (uint32_t*)(smem + off); off += (size_t)im.mcode.num_clusters * 4;
    uint8_t* sacm = smem + off; off += im.acode.num_ctx;
    uint8_t* smcm = smem + off; off += im.mcode.num_ctx;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (uint32_t i = tid; i < (im.acode.num_clusters << im.acode.log_alpha); i += nt) sa[i] = a_alias[i];
    for (uint32_t i = tid; i < (im.mcode.num_clusters << im.mcode.log_alpha); i += nt) sm[i] = m_alias[i];
    for (int i = tid; i < im.tree_size; i += nt) st[i] = tree[i];
    for (uint32_t i = tid; i < im.acode.num_clusters; i += nt) sac[i] = a_cfg[i];
    for (uint32_t i = tid; i < im.mcode.num_clusters; i += nt) smc[i] = m_cfg[i];
    for (uint32_t i = tid; i < im.acode.num_ctx; i += nt) sacm[i] = a_cmap[i];
    for (uint32_t i = tid; i < im.mcode.num_ctx; i += nt) smcm[i] = m_cmap[i];
    __syncthreads();
    a_cmap = sacm; a_cfg = sac; a_alias = sa; m_cmap = smcm; m_cfg = smc; m_alias = sm; tree = st;
  }
  if (threadIdx.x % lane_stride) return;
  const int si = threadIdx.x / lane_stride;
  if (si >= task.count) return;
  const int g = task.first + si;
  const int gx = g % im.xg, gy = g / im.xg;
  const int bx0 = gx * kGroupBlocks, by0 = gy * kGroupBlocks;
  const int bw = min(kGroupBlocks, im.w8 - bx0), bh = min(kGroupBlocks, im.h8 - by0);
  const int sec = 2 + im.nlf + g;
  LaneBits b;
  b.Init(im.cs, im.cs_size, im.sec_off[sec]);
  Ans<kLds> ans;
  ans.b = &b; ans.cmap = a_cmap; ans.cfg = a_cfg; ans.alias = a_alias; ans.log_alpha = im.acode.log_alpha;
  const uint32_t preset = b.Read(DCeilLog2((uint32_t)im.num_presets));
  if (preset >= (uint32_t)im.num_presets) { SetError(im, kErrBitstream); return; }
  const uint32_t nbc = im.num_block_ctx;
  const uint32_t ctx_offset = preset * nbc * 495;
  ans.Start();
  uint8_t* nz = im.nzmap + (size_t)g * 3 * 1024;
  const int wp = im.wp;
  uint32_t err = 0;
  for (int by = 0; by < bh && !err; by++) {
    for (int bx = 0; bx < bw && !err; bx++) {
      const size_t cell = (size_t)(by0 + by) * im.w8 + bx0 + bx;
      const uint32_t info = im.cellinfo[cell];
      if ((info & 0x8003FF00u) != 0x80000000u) continue;   // valid and ix == iy == 0
      const uint32_t s = info & 0xFF;
      const uint32_t lcx = (info >> 18) & 7, lcy = (info >> 21) & 7;
      const uint32_t log2c = lcx + lcy, covered = 1u << log2c, size = covered << 6;
      const uint32_t ord = c_order_bucket[s];
      const bool special = IsSpecial(s);
      const uint32_t lng_log2 = 3 + max(lcx, lcy);
      const bool transposed = !special && lcy >= lcx;
      const uint32_t rq = im.rawq[cell];
      uint32_t qf_idx = 0;
      for (int i = 0; i < im.n_qf; i++) qf_idx += rq > im.qf_thr[i];
      const size_t px0 = (size_t)(by0 + by) * 8 * wp + (size_t)(bx0 + bx) * 8;
#pragma unroll 1
      for (int ci = 0; ci < 3; ci++) {
        const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
        uint8_t* row = nz + c * 1024 + by * 32;
        uint32_t predicted;
        if (bx == 0) predicted = by == 0 ? 32 : row[-32];
        else if (by == 0) predicted = row[bx - 1];
        else predicted = (row[-32 + bx] + row[bx - 1] + 1) >> 1;
        const uint32_t cprime = c < 2 ? (c ^ 1) : 2;
        const uint32_t block_ctx = im.block_ctx_map[(cprime * kNumOrders + ord) * (im.n_qf + 1) + qf_idx];
        uint32_t nzc = predicted >= 64 ? 64 : predicted;
        nzc = nzc < 8 ? nzc : 4 + nzc / 2;
        uint32_t nzeros = ans.Get(ctx_offset + nzc * nbc + block_ctx);
        if (nzeros + covered > size) { err |= kErrBitstream; break; }
        const uint8_t fill = (uint8_t)((nzeros + covered - 1) >> log2c);
        for (uint32_t iy = 0; iy < (1u << lcy); iy++)
          for (uint32_t ix = 0; ix < (1u << lcx); ix++) row[iy * 32 + bx + ix] = fill;
        const uint32_t histo = ctx_offset + nbc * 37 + 458 * block_ctx;
        const uint16_t* order = im.order[ord * 3 + c];
        int32_t* plane = im.coef[c] + px0;
        uint32_t prev = nzeros > size / 16 ? 0 : 1;
        for (uint32_t k = covered; k < size && nzeros != 0; k++) {
          const uint32_t nzl = (nzeros + covered - 1) >> log2c;
          const uint32_t ks = k >> log2c;
          const uint32_t fctx = ks < 16 ? ks - 1 : (ks < 32 ? 15 + ((ks - 16) >> 1) : 23 + ((ks - 32) >> 2));
          const uint32_t ctx = histo + ((uint32_t)c_nnz_ctx[nzl] + fctx) * 2 + prev;
          const uint32_t u = ans.Get(ctx);
          if (u) {
            const uint32_t p = order[k];
            const uint32_t r = p >> lng_log2, cc = p & ((1u << lng_log2) - 1);
            const uint32_t ky = transposed ? cc : r, kx = transposed ? r : cc;
            plane[(size_t)ky * wp + kx] = UnpackS(u);
            prev = 1;
            nzeros--;
          } else {
            prev = 0;
          }
        }
        if (nzeros != 0) { err |= kErrBitstream; break; }
      }
    }
  }
  if (!err && !ans.Final()) err |= kErrBitstream;
  if (!err && im.has_alpha) {
    if (b.Read(4) != 3) err |= kErrUnsupportedHeader;
    else {
      Ans<kLds> ma;
      ma.b = &b; ma.cmap = m_cmap; ma.cfg = m_cfg; ma.alias = m_alias; ma.log_alpha = im.mcode.log_alpha;
      ma.Start();
      const int x0 = gx * kGroupDim, y0 = gy * kGroupDim;
      const int gw = min(kGroupDim, im.w - x0), gh = min(kGroupDim, im.h - y0);
      const int sid = 1 + 3 * im.nlf + kNumQuantTables + g;
      ModularChannel(ma, tree, 0, sid, gw, gh, im.alpha32 + (size_t)y0 * im.w + x0, im.w);
      if (!ma.Final()) err |= kErrBitstream;
    }
  }
  if (!err && b.Consumed() > (uint64_t)im.sec_size[sec] * 8) err |= kErrBitstream;
  if (err) SetError(im, err);
}

// ------------------------------------------------------------------ LF pixel stages
__global__ void lf_dequant_kernel(const DevImage* imgs) {
  const DevImage& im = imgs[blockIdx.y];
  const int n = im.w8 * im.h8;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int bx = i % im.w8, by = i / im.w8;
    const int g = (by / kLfGroupBlocks) * im.xlf + bx / kLfGroupBlocks;
    const float mul = 1.0f / (float)(1 << im.lf_extra[g]);
    const float qx = (float)im.lfq[0][i], qy = (float)im.lfq[1][i], qb = (float)im.lfq[2][i];
    const float fy = qy * (im.mul_lf[1] * mul);
    im.lf[1][i] = fy;
    im.lf[0][i] = fy * im.lf_cfl_x + qx * (im.mul_lf[0] * mul);
    im.lf[2][i] = fy * im.lf_cfl_b + qb * (im.mul_lf[2] * mul);
  }
}

__global__ void lf_smooth_kernel(const DevImage* imgs) {
  const DevImage& im = imgs[blockIdx.y];
  if (im.skip_lf_smoothing) return;
  const int w = im.w8, h = im.h8, n = w * h;
  const float kW0 = 0.05226273532324128f, kW1 = 0.20345139757231578f, kW2 = 0.0334829185968739f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int x = i % w, y = i / w;
    if (w <= 2 || h <= 2 || x == 0 || y == 0 || x == w - 1 || y == h - 1) {
      for (int c = 0; c < 3; c++) im.lf_tmp[c][i] = im.lf[c][i];
      continue;
    }
    float sm[3], mc[3], gap = 0.5f;
    for (int c = 0; c < 3; c++) {
      const float* p = im.lf[c] + i;
      const float corner = p[-w - 1] + p[-w + 1] + p[w - 1] + p[w + 1];
      const float edge = p[-w] + p[-1] + p[1] + p[w];
      mc[c] = p[0];
      sm[c] = corner * kW2 + edge * kW1 + mc[c] * kW0;
      gap = fmaxf(gap, fabsf((mc[c] - sm[c]) / im.mul_lf[c]));
    }
    const float factor = fmaxf(0.0f, 3.0f - 4.0f * gap);
    for (int c = 0; c < 3; c++) im.lf_tmp[c][i] = (sm[c] - mc[c]) * factor + mc[c];
  }
}

__global__ void cell_sigma_kernel(const DevImage* imgs) {
  const DevImage& im = imgs[blockIdx.y];
  const int n = im.w8 * im.h8;
  const float kInvSigmaNum = -1.1715728752538099024f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float sigma_quant = im.epf_quant_mul / (im.quant_scale * (float)im.rawq[i] * kInvSigmaNum);
    float sigma = sigma_quant * im.epf_sharp_lut[im.sharp[i]];
    sigma = fminf(-1e-4f, sigma);
    im.inv_sigma[i] = 1.0f / sigma;
  }
}

__global__ void alpha_to_u8_kernel(const DevImage* imgs) {
  const DevImage& im = imgs[blockIdx.y];
  if (!im.has_alpha) return;
  const size_t n = (size_t)im.w * im.h;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int v = im.alpha32[i];
    im.alpha[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

// ------------------------------------------------------------------ dequantisation (+ chroma from luma), in place
__global__ void dequant_kernel(const DevImage* imgs) {
  const DevImage& im = imgs[blockIdx.y];
  const size_t n = (size_t)im.wp * im.hp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % im.wp), y = (int)(i / im.wp);
    const int bx = x >> 3, by = y >> 3;
    const size_t cell = (size_t)by * im.w8 + bx;
    const uint32_t info = im.cellinfo[cell];
    const uint32_t s = info & 0xFF, ix = (info >> 8) & 31, iy = (info >> 13) & 31, lcx = (info >> 18) & 7, lcy = (info >> 21) & 7;
    const int kx = (x & 7) + 8 * ix, ky = (y & 7) + 8 * iy;
    const int obx = bx - (int)ix, oby = by - (int)iy;   // varblock origin cell
    const bool special = IsSpecial(s);
    const uint32_t lng_log2 = 3 + max(lcx, lcy);
    const bool transposed = !special && lcy >= lcx;
    const uint32_t idx = transposed ? ((uint32_t)kx << lng_log2) + ky : ((uint32_t)ky << lng_log2) + kx;
    const uint32_t q = c_quant_table[s];
    const float* wt = im.dq[q];
    const uint32_t nq = im.dq_n[q];
    const size_t ocell = (size_t)oby * im.w8 + obx;
    const float scale = im.inv_global_scale / (float)im.rawq[ocell];
    const size_t tile = (size_t)(oby >> 3) * im.wt + (obx >> 3);
    const float cfx = im.base_x + (float)im.ytox[tile] * im.inv_color_factor;
    const float cfb = im.base_b + (float)im.ytob[tile] * im.inv_color_factor;
    float out[3];
#pragma unroll
    for (int ci = 0; ci < 3; ci++) {
      const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
      const int32_t v = im.coef[c][i];
      float a;
      if (v == 0) a = 0.f;
      else if (v == 1) a = im.qbias[c];
      else if (v == -1) a = -im.qbias[c];
      else a = (float)v - im.qbias[3] / (float)v;
      const float dqs = c == 0 ? scale * im.x_dm : (c == 1 ? scale : scale * im.b_dm);
      float d = a * dqs * wt[(size_t)c * nq + idx];
      if (c == 0) d += cfx * out[1];
      if (c == 2) d += cfb * out[1];
      out[c] = d;
    }
    float* f0 = (float*)im.coef[0];
    float* f1 = (float*)im.coef[1];
    float* f2 = (float*)im.coef[2];
    f0[i] = out[0]; f1[i] = out[1]; f2[i] = out[2];
  }
}

// LLF: the lowest cx*cy coefficients of each varblock from the (smoothed) LF image; one thread per cell.
__global__ void llf_kernel(const DevImage* imgs, const float* basis_small, const float* llf_scale) {
  const DevImage& im = imgs[blockIdx.y];
  const int n = im.w8 * im.h8;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int bx = i % im.w8, by = i / im.w8;
    const uint32_t info = im.cellinfo[i];
    const uint32_t ix = (info >> 8) & 31, iy = (info >> 13) & 31, lcx = (info >> 18) & 7, lcy = (info >> 21) & 7;
    const int cx = 1 << lcx, cy = 1 << lcy;
    const int obx = bx - (int)ix, oby = by - (int)iy;
    // basis_small holds, for c = 1,2,4,...,32 at offset (c*c-1)/3, the c x c scaled DCT basis B[k*c+n]
    const float* Bx = basis_small + (cx * cx - 1) / 3;
    const float* By = basis_small + (cy * cy - 1) / 3;
    const float sc = llf_scale[lcy * 32 + iy] * llf_scale[lcx * 32 + ix] / (float)(cx * cy);
    const size_t dst = (size_t)(oby * 8 + (int)iy) * im.wp + obx * 8 + (int)ix;
    for (int c = 0; c < 3; c++) {
      const float* lf = im.lf_final[c] + (size_t)oby * im.w8 + obx;
      float acc = 0.f;
      for (int y = 0; y < cy; y++) {
        float racc = 0.f;
        for (int x = 0; x < cx; x++) racc += lf[(size_t)y * im.w8 + x] * Bx[ix * cx + x];
        acc += racc * By[iy * cy + y];
      }
      ((float*)im.coef[c])[dst] = acc * sc;
    }
  }
}

// Vertical 1-D IDCT: thread = (column x, 8-row cell), all three channels.
__global__ void idct_v_kernel(const DevImage* imgs, const float* basis_all) {
  const DevImage& im = imgs[blockIdx.y];
  const size_t n = (size_t)im.wp * im.h8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % im.wp), by = (int)(i / im.wp);
    const uint32_t info = im.cellinfo[(size_t)by * im.w8 + (x >> 3)];
    const uint32_t s = info & 0xFF;
    if (IsSpecial(s)) continue;
    const uint32_t iy = (info >> 13) & 31, lcy = (info >> 21) & 7;
    const int R = 8 << lcy;
    // basis_all holds, for N = 8,...,256 at offset (N*N-64)/3, the scaled basis B[k*N+n]
    const float* B = basis_all + ((size_t)R * R - 64) / 3 + iy * 8;
    const size_t src = (size_t)(by - (int)iy) * 8 * im.wp + x;
    const size_t dst = (size_t)by * 8 * im.wp + x;
    for (int c = 0; c < 3; c++) {
      const float* in = (const float*)im.coef[c] + src;
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = 0; k < R; k++) {
        const float v = in[(size_t)k * im.wp];
        const float* b = B + (size_t)k * R;
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] += v * b[j];
      }
      float* o = im.tmp[c] + dst;
#pragma unroll
      for (int j = 0; j < 8; j++) o[(size_t)j * im.wp] = acc[j];
    }
  }
}

// Horizontal 1-D IDCT: thread = (row y, 8-column cell).
__global__ void idct_h_kernel(const DevImage* imgs, const float* basis_all) {
  const DevImage& im = imgs[blockIdx.y];
  const size_t n = (size_t)im.w8 * im.hp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int bx = (int)(i % im.w8), y = (int)(i / im.w8);
    const uint32_t info = im.cellinfo[(size_t)(y >> 3) * im.w8 + bx];
    const uint32_t s = info & 0xFF;
    if (IsSpecial(s)) continue;
    const uint32_t ix = (info >> 8) & 31, lcx = (info >> 18) & 7;
    const int C = 8 << lcx;
    const float* B = basis_all + ((size_t)C * C - 64) / 3 + ix * 8;
    const size_t src = (size_t)y * im.wp + (size_t)(bx - (int)ix) * 8;
    const size_t dst = (size_t)y * im.wp + (size_t)bx * 8;
    for (int c = 0; c < 3; c++) {
      const float* in = im.tmp[c] + src;
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = 0; k < C; k++) {
        const float v = in[k];
        const float* b = B + (size_t)k * C;
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] += v * b[j];
      }
      float* o = im.xyb[c] + dst;
#pragma unroll
      for (int j = 0; j < 8; j++) o[j] = acc[j];
    }
  }
}

// 1-D scaled IDCT of length n (4 or 8) with basis row stride n.
__device__ __forceinline__ void Idct1(const float* B, int n, const float* in, int in_stride, float* out, int out_stride) {
  for (int j = 0; j < n; j++) {
    float a = 0.f;
    for (int k = 0; k < n; k++) a += in[k * in_stride] * B[k * n + j];
    out[j * out_stride] = a;
  }
}

// 8x8 special transforms: IDENTITY, DCT2X2, DCT4X4, DCT4X8, DCT8X4.  One thread per (cell, channel).
__global__ void idct_special_kernel(const DevImage* imgs, const float* basis_all, const float* basis_small) {
  const DevImage& im = imgs[blockIdx.y];
  const int n = im.w8 * im.h8 * 3;
  const float* B8 = basis_all;                 // N = 8
  const float* B4 = basis_small + (16 - 1) / 3;  // c = 4
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int c = i % 3, cell = i / 3;
    const uint32_t s = im.cellinfo[cell] & 0xFF;
    if (!IsSpecial(s)) continue;
    const int bx = cell % im.w8, by = cell / im.w8;
    const size_t base = (size_t)by * 8 * im.wp + (size_t)bx * 8;
    const float* src = (const float*)im.coef[c] + base;
    float* dst = im.xyb[c] + base;
    float co[64], px[64];
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) co[y * 8 + x] = src[(size_t)y * im.wp + x];
    if (s == 2) {          // DCT2X2
      for (int S = 2; S <= 8; S *= 2) {
        const int h = S / 2;
        for (int y = 0; y < h; y++)
          for (int x = 0; x < h; x++) {
            const float c00 = co[y * 8 + x], c01 = co[y * 8 + h + x], c10 = co[(y + h) * 8 + x], c11 = co[(y + h) * 8 + h + x];
            px[y * 2 * 8 + x * 2] = c00 + c01 + c10 + c11;
            px[y * 2 * 8 + x * 2 + 1] = c00 + c01 - c10 - c11;
            px[(y * 2 + 1) * 8 + x * 2] = c00 - c01 + c10 - c11;
            px[(y * 2 + 1) * 8 + x * 2 + 1] = c00 - c01 - c10 + c11;
          }
        for (int y = 0; y < S; y++) for (int x = 0; x < S; x++) co[y * 8 + x] = px[y * 8 + x];
      }
      for (int k = 0; k < 64; k++) px[k] = co[k];
    } else if (s == 1 || s == 3) {   // IDENTITY / DCT4X4
      const float b00 = co[0], b01 = co[1], b10 = co[8], b11 = co[9];
      const float dcs[4] = {b00 + b01 + b10 + b11, b00 + b01 - b10 - b11, b00 - b01 + b10 - b11, b00 - b01 - b10 + b11};
      for (int y = 0; y < 2; y++)
        for (int x = 0; x < 2; x++) {
          if (s == 1) {
            float rs = 0.f;
            for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) if (ix || iy) rs += co[(y + iy * 2) * 8 + x + ix * 2];
            const float ref = dcs[y * 2 + x] - rs * (1.0f / 16);
            for (int iy = 0; iy < 4; iy++)
              for (int ix = 0; ix < 4; ix++) px[(y * 4 + iy) * 8 + x * 4 + ix] = co[(y + iy * 2) * 8 + x + ix * 2] + ref;
            px[(y * 4 + 1) * 8 + x * 4 + 1] = ref;
            px[(y * 4) * 8 + x * 4] = co[(y + 2) * 8 + x + 2] + ref;
          } else {
            // 4x4 block in stored layout (square => transposed): blk[kx*4+ky]
            float blk[16], t[16];
            for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) blk[iy * 4 + ix] = co[(y + iy * 2) * 8 + x + ix * 2];
            blk[0] = dcs[y * 2 + x];
            // horizontal: t[ky][xx] = sum_kx blk[kx*4+ky] * B4[kx][xx]
            for (int ky = 0; ky < 4; ky++) Idct1(B4, 4, blk + ky, 4, t + ky * 4, 1);
            // vertical
            for (int xx = 0; xx < 4; xx++) Idct1(B4, 4, t + xx, 4, px + (y * 4) * 8 + x * 4 + xx, 8);
          }
        }
    } else if (s == 12 || s == 13) {   // DCT4X8 / DCT8X4
      const float b0 = co[0], b1 = co[8];
      const float dcs[2] = {b0 + b1, b0 - b1};
      for (int hlf = 0; hlf < 2; hlf++) {
        float blk[32], t[32];
        for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 8; ix++) blk[iy * 8 + ix] = co[(hlf + iy * 2) * 8 + ix];
        blk[0] = dcs[hlf];
        if (s == 12) {
          // 4 rows x 8 cols, not transposed: blk[ky*8+kx]
          for (int ky = 0; ky < 4; ky++) Idct1(B8, 8, blk + ky * 8, 1, t + ky * 8, 1);          // horizontal (len 8)
          for (int xx = 0; xx < 8; xx++) Idct1(B4, 4, t + xx, 8, px + (hlf * 4) * 8 + xx, 8);    // vertical (len 4)
        } else {
          // 8 rows x 4 cols, transposed: blk[kx*8+ky]
          for (int ky = 0; ky < 8; ky++) Idct1(B4, 4, blk + ky, 8, t + ky * 4, 1);              // horizontal (len 4): t[ky*4+xx]
          for (int xx = 0; xx < 4; xx++) Idct1(B8, 8, t + xx, 4, px + hlf * 4 + xx, 8);          // vertical (len 8)
        }
      }
    } else {
      for (int k = 0; k < 64; k++) px[k] = 0.f;   // AFV: rejected on the host
    }
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) dst[(size_t)y * im.wp + x] = px[y * 8 + x];
  }
}

// ------------------------------------------------------------------ loop filters
__global__ void gaborish_kernel(const DevImage* imgs) {
  const DevImage& im = imgs[blockIdx.y];
  if (!im.stage_on[0]) return;
  const int w = im.w, h = im.h, wp = im.wp;
  const size_t n = (size_t)w * h;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % w), y = (int)(i / w);
    const int xl = DMirror(x - 1, w), xr = DMirror(x + 1, w), yt = DMirror(y - 1, h), yb = DMirror(y + 1, h);
    for (int c = 0; c < 3; c++) {
      const float* in = im.stage_in[0][c];
      const float* t = in + (size_t)yt * wp;
      const float* m = in + (size_t)y * wp;
      const float* b = in + (size_t)yb * wp;
      im.stage_out[0][c][(size_t)y * wp + x] =
          m[x] * im.gab_w[c][0] + (t[x] + b[x] + m[xl] + m[xr]) * im.gab_w[c][1] + (t[xl] + t[xr] + b[xl] + b[xr]) * im.gab_w[c][2];
    }
  }
}

template <int kStage>
__global__ void epf_kernel(const DevImage* imgs) {
  const DevImage& im = imgs[blockIdx.y];
  if (!im.stage_on[1 + kStage]) return;
  const int w = im.w, h = im.h, wp = im.wp;
  const size_t n = (size_t)w * h;
  constexpr int kNoff = kStage == 0 ? 12 : 4;
  constexpr int kNplus = kStage == 2 ? 1 : 5;
  const int off0[12][2] = {{-2, 0}, {-1, -1}, {-1, 0}, {-1, 1}, {0, -2}, {0, -1}, {0, 1}, {0, 2}, {1, -1}, {1, 0}, {1, 1}, {2, 0}};
  const int off1[4][2] = {{-1, 0}, {0, -1}, {0, 1}, {1, 0}};
  const int plus[5][2] = {{0, 0}, {-1, 0}, {1, 0}, {0, -1}, {0, 1}};
  const float sm = kStage == 0 ? im.epf_pass0_sigma_scale : (kStage == 1 ? 1.0f : im.epf_pass2_sigma_scale);
  const float bsm = sm * im.epf_border_sad_mul;
  const float* in0 = im.stage_in[1 + kStage][0];
  const float* in1 = im.stage_in[1 + kStage][1];
  const float* in2 = im.stage_in[1 + kStage][2];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % w), y = (int)(i / w);
    const size_t o = (size_t)y * wp + x;
    const float is = im.inv_sigma[(size_t)(y >> 3) * im.w8 + (x >> 3)];
    if (is < -3.90524291751269967465540850526868f) {
      im.stage_out[1 + kStage][0][o] = in0[o];
      im.stage_out[1 + kStage][1][o] = in1[o];
      im.stage_out[1 + kStage][2][o] = in2[o];
      continue;
    }
    const bool border = ((x & 7) == 0) || ((x & 7) == 7) || ((y & 7) == 0) || ((y & 7) == 7);
    const float inv = is * (border ? bsm : sm);
    float wsum = 1.0f, a0 = in0[o], a1 = in1[o], a2 = in2[o];
#pragma unroll
    for (int k = 0; k < kNoff; k++) {
      const int dy = kStage == 0 ? off0[k][0] : off1[k][0], dx = kStage == 0 ? off0[k][1] : off1[k][1];
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int p = 0; p < kNplus; p++) {
        const size_t pa = (size_t)DMirror(y + plus[p][0], h) * wp + DMirror(x + plus[p][1], w);
        const size_t pb = (size_t)DMirror(y + dy + plus[p][0], h) * wp + DMirror(x + dx + plus[p][1], w);
        s0 += fabsf(in0[pa] - in0[pb]);
        s1 += fabsf(in1[pa] - in1[pb]);
        s2 += fabsf(in2[pa] - in2[pb]);
      }
      const float sad = s0 * im.epf_channel_scale[0] + s1 * im.epf_channel_scale[1] + s2 * im.epf_channel_scale[2];
      const float wt = fmaxf(0.0f, 1.0f + sad * inv);
      const size_t pn = (size_t)DMirror(y + dy, h) * wp + DMirror(x + dx, w);
      wsum += wt;
      a0 += wt * in0[pn]; a1 += wt * in1[pn]; a2 += wt * in2[pn];
    }
    const float iw = 1.0f / wsum;
    im.stage_out[1 + kStage][0][o] = a0 * iw;
    im.stage_out[1 + kStage][1][o] = a1 * iw;
    im.stage_out[1 + kStage][2][o] = a2 * iw;
  }
}

__device__ __forceinline__ float SrgbOetf(float v) { return v <= 0.0031308f ? 12.92f * v : 1.055f * powf(v, 1.0f / 2.4f) - 0.055f; }
__device__ __forceinline__ uint8_t ToU8(float v) {
  v *= 255.0f;
  if (!(v > 0.f)) return 0;
  if (v >= 255.0f) return 255;
  return (uint8_t)(v + 0.5f);
}

__global__ void xyb_to_out_kernel(const DevImage* imgs) {
  const DevImage& im = imgs[blockIdx.y];
  const int w = im.w, wp = im.wp;
  const size_t n = (size_t)w * im.h;
  const float* p0 = im.stage_in[4][0];
  const float* p1 = im.stage_in[4][1];
  const float* p2 = im.stage_in[4][2];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % w), y = (int)(i / w);
    const size_t o = (size_t)y * wp + x;
    const float X = p0[o], Y = p1[o], B = p2[o];
    const float gr = Y + X - im.opsin_bias_cbrt[0], gg = Y - X - im.opsin_bias_cbrt[1], gb = B - im.opsin_bias_cbrt[2];
    const float mr = gr * gr * gr + im.opsin_bias[0], mg = gg * gg * gg + im.opsin_bias[1], mb = gb * gb * gb + im.opsin_bias[2];
    float r = im.opsin_inv[0] * mr + im.opsin_inv[1] * mg + im.opsin_inv[2] * mb;
    float g = im.opsin_inv[3] * mr + im.opsin_inv[4] * mg + im.opsin_inv[5] * mb;
    float bl = im.opsin_inv[6] * mr + im.opsin_inv[7] * mg + im.opsin_inv[8] * mb;
    if (im.to_srgb) { r = SrgbOetf(r); g = SrgbOetf(g); bl = SrgbOetf(bl); }
    uint8_t* out = im.out + i * im.nch_out;
    if (im.ncolor == 3) {
      out[0] = ToU8(r); out[1] = ToU8(g); out[2] = ToU8(bl);
      if (im.has_alpha) out[3] = im.alpha[i];
    } else {
      out[0] = ToU8(g);
      if (im.has_alpha) out[1] = im.alpha[i];
    }
  }
}

// ------------------------------------------------------------------ launch wrappers
static inline dim3 Grid2(size_t work, int nimg, int block = 256, int cap = 4096) {
  size_t b = (work + block - 1) / block;
  if (b > (size_t)cap) b = cap;
  if (b < 1) b = 1;
  return dim3((unsigned)b, (unsigned)nimg);
}

void LaunchLfGroups(const DevImage* imgs, const SectionTask* tasks, int ntasks, hipStream_t s) {
  if (ntasks <= 0) return;
  hipLaunchKernelGGL(lf_group_kernel, dim3((ntasks + 63) / 64), dim3(64), 0, s, imgs, tasks, ntasks);
}

void LaunchPassGroups(const DevImage* imgs, const SectionTask* tasks, int nwg, int lane_stride, size_t lds_bytes, hipStream_t s) {
  if (nwg <= 0) return;
  if (lds_bytes > 0 && lds_bytes <= 64 * 1024) {
    hipLaunchKernelGGL(pass_group_kernel<true>, dim3(nwg), dim3(256), lds_bytes, s, imgs, tasks, lane_stride);
  } else {
    hipLaunchKernelGGL(pass_group_kernel<false>, dim3(nwg), dim3(256), 0, s, imgs, tasks, lane_stride);
  }
}

void LaunchLfPixelStages(const DevImage* imgs, int nimg, size_t max_cells, hipStream_t s) {
  hipLaunchKernelGGL(lf_dequant_kernel, Grid2(max_cells, nimg), dim3(256), 0, s, imgs);
  hipLaunchKernelGGL(lf_smooth_kernel, Grid2(max_cells, nimg), dim3(256), 0, s, imgs);
  hipLaunchKernelGGL(cell_sigma_kernel, Grid2(max_cells, nimg), dim3(256), 0, s, imgs);
}

void LaunchAlphaToU8(const DevImage* imgs, int nimg, size_t max_pixels, hipStream_t s) {
  hipLaunchKernelGGL(alpha_to_u8_kernel, Grid2(max_pixels, nimg), dim3(256), 0, s, imgs);
}

void LaunchReconstruct(const DevImage* imgs, int nimg, size_t max_padded_pixels, size_t max_cells, const float* basis_all,
                       const float* basis_small, const float* llf_scale, hipStream_t s) {
  hipLaunchKernelGGL(dequant_kernel, Grid2(max_padded_pixels, nimg, 256, 8192), dim3(256), 0, s, imgs);
  hipLaunchKernelGGL(llf_kernel, Grid2(max_cells, nimg), dim3(256), 0, s, imgs, basis_small, llf_scale);
  hipLaunchKernelGGL(idct_v_kernel, Grid2(max_padded_pixels / 8, nimg, 256, 8192), dim3(256), 0, s, imgs, basis_all);
  hipLaunchKernelGGL(idct_h_kernel, Grid2(max_padded_pixels / 8, nimg, 256, 8192), dim3(256), 0, s, imgs, basis_all);
  hipLaunchKernelGGL(idct_special_kernel, Grid2(max_cells * 3, nimg), dim3(256), 0, s, imgs, basis_all, basis_small);
}

void LaunchFiltersAndOutput(const DevImage* imgs, int nimg, size_t max_pixels, bool any_gab, int max_epf, hipStream_t s) {
  dim3 g = Grid2(max_pixels, nimg, 256, 8192);
  if (any_gab) hipLaunchKernelGGL(gaborish_kernel, g, dim3(256), 0, s, imgs);
  if (max_epf >= 3) hipLaunchKernelGGL(epf_kernel<0>, g, dim3(256), 0, s, imgs);
  if (max_epf >= 1) hipLaunchKernelGGL(epf_kernel<1>, g, dim3(256), 0, s, imgs);
  if (max_epf >= 2) hipLaunchKernelGGL(epf_kernel<2>, g, dim3(256), 0, s, imgs);
  hipLaunchKernelGGL(xyb_to_out_kernel, g, dim3(256), 0, s, imgs);
}

}  // namespace jxlhip
