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
#include <cstdlib>
#include "dev_types.h"
#include "dev_util.h"
#include "kernels.h"

namespace jxlhip {

namespace {

__device__ const uint8_t d_order_bucket[kNumStrategies] = {0, 1, 1, 1, 2, 3, 4, 4, 5, 5, 6, 6, 1, 1, 1, 1, 1, 1, 7, 8, 8, 9, 10, 10, 11, 12, 12};
__device__ const uint8_t d_log2cx[kNumStrategies] = {0, 0, 0, 0, 1, 2, 0, 1, 0, 2, 1, 2, 0, 0, 0, 0, 0, 0, 3, 2, 3, 4, 3, 4, 5, 4, 5};
__device__ const uint8_t d_log2cy[kNumStrategies] = {0, 0, 0, 0, 1, 2, 1, 0, 2, 0, 2, 1, 0, 0, 0, 0, 0, 0, 3, 3, 2, 4, 4, 3, 5, 5, 4};
__device__ const uint8_t d_nnz_ctx[64] = {0,   0,   31,  62,  62,  93,  93,  93,  93,  123, 123, 123, 123, 152, 152, 152, 152, 152, 152, 152, 152, 180,
                                          180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206,
                                          206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206};

__device__ __forceinline__ int CeilLog2D(uint32_t x) { return x <= 1 ? 0 : 32 - __clz(x - 1); }
__device__ __forceinline__ int32_t UnpackSigned(uint32_t u) { return (int32_t)(u >> 1) ^ -(int32_t)(u & 1); }
// who: 1 lf_ans, 2 lf_finish, 3 hf_decode, 4 alpha_ans, 5 modular; where: section index (diagnostics: status[2 + who] = where + 1)
// Wavefront issue priority (s_setprio) of the serial decoders, for tools/ab_prio.sh.  Measured in the pipelined step (round 3):
// raising it changes nothing (123-125 ms at 0, 1 and 3), so the default emits no instruction.
#ifndef JXLHIP_PRIO_SERIAL
#define JXLHIP_PRIO_SERIAL 0
#endif
#if JXLHIP_PRIO_SERIAL
#define JXL_SERIAL_PRIO() __builtin_amdgcn_s_setprio(JXLHIP_PRIO_SERIAL)
#else
#define JXL_SERIAL_PRIO() ((void)0)
#endif

__device__ __forceinline__ void SetError(const DevImage& im, uint32_t bits, int who = 0, int where = 0) {
  atomicOr(im.status, bits);
  if (who) im.status[2 + who] = (uint32_t)where + 1;
}

// Table pointers are typed by address space so that the LDS variants compile to ds_read (not flat_load).
#define JXL_LDS __attribute__((address_space(3)))
#define JXL_GLB __attribute__((address_space(1)))
// pointers read out of DevImage are generic; cast the hot ones so they compile to global_load/global_store
template <class T> __device__ __forceinline__ JXL_GLB T* G(T* p) { return (JXL_GLB T*)p; }
typedef int __attribute__((ext_vector_type(4))) I4;
typedef unsigned __attribute__((ext_vector_type(2))) U2;   // one varblock descriptor
typedef unsigned __attribute__((ext_vector_type(4))) U4;   // one MA-tree node (builtin vector: loadable from any address space)

// ------------------------------------------------------------------ per-lane bit reader over an LDS ring
// vmcnt counts loads AND stores in issue order, so a reader that fetches its next word from global memory inside the
// token loop makes every refill wait for the coefficient / sample store issued just before it (a full HBM round trip
// per token).  Instead every lane owns a 32-word window of its stream in LDS (word j at ring[(j & 31) * slots + slot]:
// lanes reading the same j hit distinct banks); the token loops call TopUp() every 16 tokens (a token consumes at most
// 48 bits, so 24 words), which is the only place that waits on global memory.
constexpr int kRingWords = 32;
constexpr int kTopUpEvery = 16;
template <int kRing, int kBatch>
struct LaneBitsT {
  const JXL_GLB uint32_t* w;   // 32-byte aligned base of the lane's stream
  JXL_LDS uint32_t* ring;
  uint32_t rs;                 // ring stride (slots)
  uint32_t nwords;             // words readable from w
  uint32_t rd, filled;         // next word to consume / words [0, filled) have been staged
  uint64_t buf;
  int n;
  int skip;
  uint32_t pend[kBatch];       // words [filled, filled + kBatch) as requested by the previous top-up, not yet in the ring
  bool has_pend;
  // state of the general symbol reader (prefix codes / LZ77), per stream like the bit position
  JXL_GLB uint32_t* lz_win;    // window of decoded values (LZ77), lz_mask + 1 entries
  uint32_t lz_mask, lz_dm, lz_copy, lz_src, lz_done, slow_err;
  __device__ __forceinline__ void SetLz(uint32_t* win, uint32_t log_size, uint32_t dist_mult) {
    lz_win = (JXL_GLB uint32_t*)win; lz_mask = (1u << log_size) - 1; lz_dm = dist_mult;
  }
  __device__ __forceinline__ void Fetch(uint32_t* v) {   // words [filled, filled + kBatch) into registers
    static_assert(kBatch == 4 || kBatch == 8, "one or two 16-byte loads");
    if (filled + kBatch <= nwords) {
      const U4 a = *(const JXL_GLB U4*)(w + filled);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
      if constexpr (kBatch == 8) {
        const U4 c = *(const JXL_GLB U4*)(w + filled + 4);
        v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < kBatch; i++) v[i] = filled + i < nwords ? w[filled + i] : 0u;
    }
  }
  __device__ __forceinline__ void Stage(const uint32_t* v) {
#pragma unroll
    for (int i = 0; i < kBatch; i++) ring[__umul24((filled + i) & (kRing - 1), rs)] = v[i];
    filled += kBatch;
  }
  __device__ __forceinline__ void Batch() {   // stages words [filled, filled + kBatch), waiting for them
    uint32_t v[kBatch];
    Fetch(v);
    Stage(v);
  }
  __device__ __forceinline__ void TopUpSync() {
    while (filled + kBatch <= rd + kRing) Batch();
  }
  // Called every kRing / 2 tokens (a token consumes at most 48 bits, so a period consumes at most 3/4 of the window).  The words
  // requested by the PREVIOUS call go into the ring now - their loads are a whole period old, so this wait is normally free,
  // whereas load-and-wait on the spot exposed a full L2 / HBM round trip per period (hf_decode took twice as long once a batch held
  // eight distinct images instead of one cached one) - then the next batch is requested.  Only a burst that drained more than one
  // batch in a period falls back to waiting.
  __device__ __forceinline__ void TopUp() {
    if (has_pend) { Stage(pend); has_pend = false; }
    while (filled + kBatch <= rd + kRing && filled - rd < (uint32_t)(kRing * 3 / 4)) Batch();
    if (filled + kBatch <= rd + kRing) { Fetch(pend); has_pend = true; }
  }
  __device__ __forceinline__ void Init(const uint8_t* cs, uint64_t cs_size, uint64_t bit_off, JXL_LDS uint32_t* ring_slot, uint32_t ring_stride) {
    const uintptr_t addr = (uintptr_t)cs + (bit_off >> 3);
    const uintptr_t al = addr & ~(uintptr_t)31;
    w = (const JXL_GLB uint32_t*)al;
    ring = ring_slot; rs = ring_stride;
    const uintptr_t end = ((uintptr_t)cs + cs_size + 3) & ~(uintptr_t)3;
    nwords = end > al ? (uint32_t)((end - al) >> 2) : 0u;
    rd = 0; filled = 0; buf = 0; n = 0; has_pend = false;
    lz_win = nullptr; lz_mask = 0; lz_dm = 0; lz_copy = 0; lz_src = 0; lz_done = 0; slow_err = 0;
    skip = (int)(addr - al) * 8 + (int)(bit_off & 7);
    TopUpSync();
    rd = (uint32_t)skip >> 5;   // whole words before the start are skipped, the rest bit by bit
    Refill();
    const int s2 = skip & 31;
    buf >>= s2;
    n -= s2;
  }
  __device__ __forceinline__ void Refill() {
    if (n <= 32) {
      buf |= (uint64_t)ring[__umul24(rd & (kRing - 1), rs)] << n;
      n += 32;
      rd++;
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
  __device__ uint64_t Consumed() const { return (uint64_t)rd * 32 - n - skip; }
};
// the default window: 32 words, topped up (two 16-byte loads at a time) every 16 tokens
typedef LaneBitsT<kRingWords, 8> LaneBits;

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
  const uint32_t* direct;   // DevCode::direct when this wavefront decodes one section (per-residue tables, read by scalar loads), else null
  uint32_t log_alpha;
  uint32_t slow;        // DevCode::slow (uniform over the workgroup: one image's code)
  const DevCode* dc;    // the prefix / LZ77 parameters stay in global memory
};

__device__ __forceinline__ DevTreeNode NodeOf(I4 v) {
  DevTreeNode n;
  n.property = v.x; n.splitval = v.y; n.a = (uint32_t)v.z; n.b = (uint32_t)v.w;
  return n;
}

template <class Bits>
__device__ __forceinline__ uint32_t HybridTail(Bits& b, uint32_t c, uint32_t sym) {
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
  state = __umul24(freq, state >> 12) + off;   // 13 x 20 bits: full-rate v_mad_u32_u24 (v_mul_lo_u32 is quarter rate)
  if (state < 65536u) state = (state << 16) | b.Read(16);
  return sym;
}

// The same with the alias entry of the NEXT token requested as soon as the new state is known: its LDS round trip then overlaps the
// hybrid-uint tail, the predictor arithmetic and the store of the current token instead of heading the next iteration.  `e` holds the
// entry for the current state on entry (AnsPrefetch) and for the new state on exit; only valid while the cluster stays the same.
template <bool kLds>
__device__ __forceinline__ uint64_t AnsPrefetch(uint32_t state, typename AS<kLds>::U64 abase, uint32_t log_alpha) {
  return abase[(state & 0xFFF) >> (12 - log_alpha)];
}
template <bool kLds>
__device__ __forceinline__ uint32_t AnsSymPf(LaneBits& b, uint32_t& state, typename AS<kLds>::U64 abase, uint32_t log_alpha, uint64_t& e) {
  const uint32_t le = 12 - log_alpha;
  const uint32_t res = state & 0xFFF, i = res >> le, pos = res & ((1u << le) - 1);
  const uint32_t x = (uint32_t)e, y = (uint32_t)(e >> 32);
  const bool g = pos >= (x & 0xFF);
  const uint32_t sym = g ? ((x >> 8) & 0xFF) : i;
  const uint32_t off = g ? (y & 0xFFFF) + pos : pos;
  const uint32_t freq = g ? ((x >> 16) ^ (y >> 16)) : (x >> 16);
  state = __umul24(freq, state >> 12) + off;
  if (state < 65536u) state = (state << 16) | b.Read(16);
  e = abase[(state & 0xFFF) >> le];
  return sym;
}

// The general symbol reader: prefix codes (canonical code walked one length at a time over the next 15 bits) and LZ77 (a window of
// decoded values per stream in global memory; copies, special two-dimensional distances).  Compatibility path: correct, not tuned -
// streams written with these options are decoded several times slower than ANS streams without LZ77.
__device__ const int8_t d_special_dist[120][2] = {
    {0, 1}, {1, 0}, {1, 1}, {-1, 1}, {0, 2}, {2, 0}, {1, 2}, {-1, 2}, {2, 1}, {-2, 1}, {2, 2}, {-2, 2}, {0, 3}, {3, 0}, {1, 3},
    {-1, 3}, {3, 1}, {-3, 1}, {2, 3}, {-2, 3}, {3, 2}, {-3, 2}, {0, 4}, {4, 0}, {1, 4}, {-1, 4}, {4, 1}, {-4, 1}, {3, 3}, {-3, 3},
    {2, 4}, {-2, 4}, {4, 2}, {-4, 2}, {0, 5}, {3, 4}, {-3, 4}, {4, 3}, {-4, 3}, {5, 0}, {1, 5}, {-1, 5}, {5, 1}, {-5, 1}, {2, 5},
    {-2, 5}, {5, 2}, {-5, 2}, {4, 4}, {-4, 4}, {3, 5}, {-3, 5}, {5, 3}, {-5, 3}, {0, 6}, {6, 0}, {1, 6}, {-1, 6}, {6, 1}, {-6, 1},
    {2, 6}, {-2, 6}, {6, 2}, {-6, 2}, {4, 5}, {-4, 5}, {5, 4}, {-5, 4}, {3, 6}, {-3, 6}, {6, 3}, {-6, 3}, {0, 7}, {7, 0}, {1, 7},
    {-1, 7}, {5, 5}, {-5, 5}, {7, 1}, {-7, 1}, {4, 6}, {-4, 6}, {6, 4}, {-6, 4}, {2, 7}, {-2, 7}, {7, 2}, {-7, 2}, {3, 7}, {-3, 7},
    {7, 3}, {-7, 3}, {5, 6}, {-5, 6}, {6, 5}, {-6, 5}, {8, 0}, {4, 7}, {-4, 7}, {7, 4}, {-7, 4}, {8, 1}, {8, 2}, {6, 6}, {-6, 6},
    {8, 3}, {5, 7}, {-5, 7}, {7, 5}, {-7, 5}, {8, 4}, {6, 7}, {-6, 7}, {7, 6}, {-7, 6}, {8, 5}, {7, 7}, {-7, 7}, {8, 6}, {8, 7}};

template <bool kLds, class Bits>
__device__ __forceinline__ uint32_t SlowSymbol(Bits& b, uint32_t& state, const CodeTab<kLds>& t, uint32_t cl) {
  const DevCode& dc = *t.dc;
  if (dc.slow & 1) {
    const uint32_t c = t.cfg[cl];
    if (c & 0x1000) return c >> 16;   // one-symbol code: no bits
    b.Refill();
    const uint32_t bits = (uint32_t)b.buf;
    const uint16_t* cnt = dc.pfx_count + cl * 16;
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; l++) {
      code |= (bits >> (l - 1)) & 1;
      const int k = cnt[l];
      if (code - k < first) {
        b.buf >>= l; b.n -= l;
        return dc.pfx_sorted[dc.pfx_off[cl] + index + code - first];
      }
      index += k; first = (first + k) << 1; code <<= 1;
    }
    b.slow_err = 1;
    return 0;
  }
  const uint32_t le = 12 - t.log_alpha;
  const uint32_t res = state & 0xFFF, i = res >> le, pos = res & ((1u << le) - 1);
  const uint64_t e = t.alias[(cl << t.log_alpha) | i];
  const uint32_t x = (uint32_t)e, y = (uint32_t)(e >> 32);
  const bool g = pos >= (x & 0xFF);
  const uint32_t sym = g ? ((x >> 8) & 0xFF) : i;
  const uint32_t off = g ? (y & 0xFFFF) + pos : pos;
  const uint32_t freq = g ? ((x >> 16) ^ (y >> 16)) : (x >> 16);
  state = freq * (state >> 12) + off;
  if (state < 65536u) state = (state << 16) | b.Read(16);
  return sym;
}

template <bool kLds, class Bits>
__device__ __forceinline__ uint32_t SlowGet(Bits& b, uint32_t& state, const CodeTab<kLds>& t, uint32_t ctx) {
  const DevCode& dc = *t.dc;
  const bool lz = (dc.slow & 2) != 0 && b.lz_win != nullptr;
  if (lz && b.lz_copy) {
    const uint32_t v = b.lz_win[(b.lz_src++) & b.lz_mask];
    b.lz_copy--;
    b.lz_win[(b.lz_done++) & b.lz_mask] = v;
    return v;
  }
  const uint32_t cl = t.cmap[ctx];
  const uint32_t sym = SlowSymbol(b, state, t, cl);
  if ((dc.slow & 2) && sym >= dc.lz_min_symbol) {
    if (!lz) { b.slow_err = 1; return 0; }   // a copy in a stream that was given no window
    const uint32_t len = HybridTail(b, dc.lz_len_cfg, sym - dc.lz_min_symbol) + dc.lz_min_length;
    const uint32_t dcl = dc.lz_dist_cluster;
    uint32_t dist = HybridTail(b, t.cfg[dcl], SlowSymbol(b, state, t, dcl));
    if (b.lz_dm == 0) dist++;
    else if (dist < 120) { const int o = d_special_dist[dist][0] + (int)b.lz_dm * d_special_dist[dist][1]; dist = o < 1 ? 1u : (uint32_t)o; }
    else dist -= 119;
    dist = min(dist, min(b.lz_done, 1u << 20));
    if (dist == 0 || b.lz_done > b.lz_mask + 1) { b.slow_err = 1; return 0; }   // nothing to copy from / a stream longer than its window
    b.lz_src = b.lz_done - dist;
    b.lz_copy = len - 1;
    const uint32_t v = b.lz_win[(b.lz_src++) & b.lz_mask];
    b.lz_win[(b.lz_done++) & b.lz_mask] = v;
    return v;
  }
  const uint32_t v = HybridTail(b, t.cfg[cl], sym);
  if (lz) b.lz_win[(b.lz_done++) & b.lz_mask] = v;
  return v;
}
// first value of a stream's ANS state: read from the stream, except for prefix codes (no state)
template <bool kLds, class Bits>
__device__ __forceinline__ uint32_t InitAnsState(Bits& b, const CodeTab<kLds>& t) { return (t.slow & 1) ? 0x130000u : b.Read(32); }

template <bool kLds, class Bits>
__device__ __forceinline__ uint32_t AnsGet(Bits& b, uint32_t& state, const CodeTab<kLds>& t, uint32_t ctx) {
  if (t.slow) return SlowGet(b, state, t, ctx);   // uniform over the wavefront: one image, one code
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
  state = __umul24(freq, state >> 12) + off;   // 13 x 20 bits: full-rate v_mad_u32_u24 (v_mul_lo_u32 is quarter rate)
  if (state < 65536u) state = (state << 16) | b.Read(16);
  return HybridTail(b, c, sym);
}

// Cooperative copy of a code's tables into LDS; returns the carved end offset.
__device__ __forceinline__ size_t StageCode(JXL_LDS uint8_t* smem, size_t off, const DevCode& dc, CodeTab<true>& t, int tid, int nt,
                                            bool one_section = false) {
  const uint32_t na = (dc.slow & 1) ? 0u : dc.num_clusters << dc.log_alpha;   // prefix codes have no alias tables
  // One-section wavefronts run their row loops on the scalar unit (RowScalar) and read the code's per-residue tables - the alias
  // tables spelled out for each of the 4096 state residues, built by the host for small launches - through the scalar cache: such a
  // wavefront issues an instruction every four to five cycles whatever unit executes it, so its time per token is its instruction
  // count, and the alias arithmetic (bucket, cutoff compare, three selects) is a third of the token.
  t.direct = (one_section && !dc.slow) ? dc.direct : nullptr;
  off = (off + 7) & ~(size_t)7;
  JXL_LDS uint64_t* sa = (JXL_LDS uint64_t*)(smem + off); off += (size_t)na * 8;
  JXL_LDS uint32_t* sc = (JXL_LDS uint32_t*)(smem + off); off += (size_t)dc.num_clusters * 4;
  JXL_LDS uint8_t* sm = smem + off; off += dc.num_ctx;
  for (uint32_t i = tid; i < na; i += nt) sa[i] = dc.alias[i];
  for (uint32_t i = tid; i < dc.num_clusters; i += nt) sc[i] = dc.cfg[i];
  for (uint32_t i = tid; i < dc.num_ctx; i += nt) sm[i] = dc.ctx_map[i];
  t.cmap = sm; t.cfg = sc; t.alias = sa; t.log_alpha = dc.log_alpha; t.slow = dc.slow; t.dc = &dc;
  return off;
}

__device__ __forceinline__ void GlobalCode(const DevCode& dc, CodeTab<false>& t) {
  t.cmap = dc.ctx_map; t.cfg = dc.cfg; t.alias = dc.alias; t.direct = nullptr; t.log_alpha = dc.log_alpha; t.slow = dc.slow; t.dc = &dc;
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
  uint64_t entry = constant_token ? 0 : AnsPrefetch<kLds>(state, abase, tab.log_alpha);
  for (int x = 0; x < w; x++) {
    if ((x & (kTopUpEvery - 1)) == 0) b.TopUp();
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
      const uint32_t sym = AnsSymPf<kLds>(b, state, abase, tab.log_alpha, entry);
      res = (uint32_t)UnpackSigned(HybridTail(b, cfg, sym)) * leaf.b + (uint32_t)leaf.splitval;
    }
    const int32_t val = (int32_t)(res + guess);   // low 32 bits of the 64-bit reference arithmetic
    row[x] = val;
    if (use_rb) rb[x * rs] = val;
    W = val;
    if (y) { NW = N; N = NE; } else { NW = val; N = val; }
  }
}

// ------------------------------------------------------------------ weighted ("self-correcting") predictor, default parameters
// Per-channel state in a lane-private global scratch: the true error and the four sub-predictor errors of the current
// and the previous row ((w + 2) entries each), exactly as the format defines them.
struct WpState {
  int32_t* err;                // [2][w + 2]   (generic pointers: the scratch is LDS when it fits, global memory otherwise)
  uint32_t* pe;                // [4][2][w + 2]
  int w2;                      // w + 2
  int64_t prediction[4];
  int64_t pred;
  __device__ void Init(int32_t* scratch, int w) {
    w2 = w + 2;
    err = scratch;
    pe = (uint32_t*)(err + 2 * w2);
    for (int i = 0; i < 10 * w2; i++) err[i] = 0;
  }
  static __device__ __forceinline__ uint32_t Div(uint32_t i) { return (1u << 24) / (i + 1); }
  static __device__ __forceinline__ uint32_t ErrorWeight(uint64_t x, uint32_t maxweight) {
    int shift = (63 - __clzll((long long)(x + 1))) - 5;
    if (shift < 0) shift = 0;
    return 4 + (uint32_t)((maxweight * (uint64_t)Div((uint32_t)(x >> shift))) >> shift);
  }
  static __device__ __forceinline__ int64_t Abs(int64_t v) { return v < 0 ? -v : v; }
  __device__ int64_t Predict(int x, int y, int w, int64_t N, int64_t W, int64_t NE, int64_t NW, int64_t NN, int64_t* max_err) {
    const int cur = (y & 1) ? 0 : w2, prev = (y & 1) ? w2 : 0;
    const int pN = prev + x, pNE = x < w - 1 ? pN + 1 : pN, pNW = x > 0 ? pN - 1 : pN;
    const uint32_t kW[4] = {13, 12, 12, 12};
    uint32_t weights[4];
    for (int i = 0; i < 4; i++) {
      const uint32_t* p = pe + (size_t)i * 2 * w2;
      weights[i] = ErrorWeight((uint64_t)p[pN] + p[pNE] + p[pNW], kW[i]);
    }
    N <<= 3; W <<= 3; NE <<= 3; NW <<= 3; NN <<= 3;
    const int64_t teW = x == 0 ? 0 : err[cur + x - 1], teN = err[pN], teNW = err[pNW], teNE = err[pNE];
    const int64_t sumWN = teN + teW;
    int64_t p = teW;
    if (Abs(teN) > Abs(p)) p = teN;
    if (Abs(teNW) > Abs(p)) p = teNW;
    if (Abs(teNE) > Abs(p)) p = teNE;
    *max_err = p;
    prediction[0] = W + NE - N;
    prediction[1] = N - (((sumWN + teNE) * 16) >> 5);
    prediction[2] = W - (((sumWN + teNW) * 10) >> 5);
    prediction[3] = N - ((teNW * 7 + teN * 7 + teNE * 7) >> 5);
    uint32_t weight_sum = weights[0] + weights[1] + weights[2] + weights[3];
    const int log_weight = 31 - __clz(weight_sum);
    weight_sum = 0;
    for (int i = 0; i < 4; i++) { weights[i] >>= log_weight - 4; weight_sum += weights[i]; }
    int64_t sum = (int64_t)(weight_sum >> 1) - 1;
    for (int i = 0; i < 4; i++) sum += prediction[i] * (int64_t)weights[i];
    pred = (sum * (int64_t)Div(weight_sum - 1)) >> 24;
    if (((teN ^ teW) | (teN ^ teNW)) > 0) return (pred + 3) >> 3;
    const int64_t mx = W > NE ? (W > N ? W : N) : (NE > N ? NE : N), mn = W < NE ? (W < N ? W : N) : (NE < N ? NE : N);
    pred = pred < mn ? mn : (pred > mx ? mx : pred);
    return (pred + 3) >> 3;
  }
  __device__ void Update(int64_t val, int x, int y) {
    const int cur = (y & 1) ? 0 : w2, prev = (y & 1) ? w2 : 0;
    val <<= 3;
    err[cur + x] = (int32_t)(pred - val);
    for (int i = 0; i < 4; i++) {
      const uint32_t e = (uint32_t)((Abs(prediction[i] - val) + 3) >> 3);
      uint32_t* p = pe + (size_t)i * 2 * w2;
      p[cur + x] = e;
      p[prev + x + 1] += e;
    }
  }
};

// Decodes one channel (w x h) into `out` (row stride `stride`).  Every property 0..14 and every predictor
// except the weighted one are supported; the host rejects trees that need more.
template <bool kLds>
__device__ __forceinline__ void ModularChannel(LaneBits& b, uint32_t& state, const CodeTab<kLds>& tab, typename AS<kLds>::Tree tree, int chan,
                               int stream_id, int w, int h, int32_t* out_generic, int stride, const RowBuf<kLds>& rbuf,
                               int32_t* wp_scratch = nullptr) {
  JXL_GLB int32_t* const out = G(out_generic);
  const bool use_wp = wp_scratch != nullptr;   // the tree references the weighted predictor: its state follows every sample
  WpState wp;
  if (use_wp) wp.Init(wp_scratch, w);
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
      if (row_leaf && !use_wp && !tab.slow) {
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
    if ((x & (kTopUpEvery - 1)) == 0) b.TopUp();
    const int32_t NE = (x + 1 < w && y) ? (use_rb ? rb[(x + 1) * rs] : prow[x + 1]) : N;
    int64_t wp_pred = 0, wp_err = 0;
    if (use_wp) {
      const int32_t NNw = y > 1 ? prow[x - stride] : N;
      wp_pred = wp.Predict(x, y, w, N, W, NE, NW, NNw, &wp_err);
    }
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
          case 15: p = wp_err; break;
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
      case 6: guess = wp_pred; break;
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
    if (use_wp) wp.Update(val, x, y);
    if (use_rb) rb[x * rs] = val;   // slot x held prev[x], which now lives in N
    prev9 = (int64_t)W + N - NW;
    const int32_t oldW = W;
    W = val;
    WW = x >= 1 ? oldW : val;
    if (y) { NW = N; N = NE; } else { NW = val; N = val; }
    if (++x == w) { x = 0; y++; row += stride; }
  }
}

// ------------------------------------------------------------------ row-static channels (two-phase decode)
// A channel is "row-static" when, for every row, the MA-tree walk ends in a leaf using only the static properties
// (channel, stream, row).  Its tokens then need no decoded neighbour: phase A (one lane per section, *_ans_kernel)
// turns the ANS stream into residuals stored in the output plane; phase B (*_finish_kernel, a whole wavefront per
// channel) applies the predictors.  Everything else takes the generic per-lane ModularChannel path in phase A.
constexpr int kCarryInts = 1024;   // widest channel whose last row can be carried between 64-row batches in LDS

template <class TreeP>
__device__ __forceinline__ DevTreeNode RowNode(TreeP tree, int chan, int sid, int y, bool* used_y) {
  DevTreeNode nd = NodeOf(tree[0]);
  while (nd.property >= 0 && nd.property <= 2) {
    if (nd.property == 2) *used_y = true;
    const int v = nd.property == 0 ? chan : (nd.property == 1 ? sid : y);
    nd = NodeOf(tree[v > nd.splitval ? nd.a : nd.b]);
  }
  return nd;
}

__device__ __forceinline__ bool SkewPred(uint32_t p) { return p == 0 || p == 1 || p == 2 || p == 5; }

// 0: generic path; 1: row-static (needs_n: some row predicts from the row above); 2: constant channel (*cval)
template <bool kLds>
__device__ int ClassifyChannel(const CodeTab<kLds>& tab, typename AS<kLds>::Tree tree, int chan, int sid, int w, int h, bool* needs_n,
                               int32_t* cval) {
  bool all_const = true, first = true, nn = false;
  int32_t cv = 0;
  for (int y = 0; y < h; y++) {
    bool used_y = false;
    const DevTreeNode nd = RowNode(tree, chan, sid, y, &used_y);
    if (nd.property >= 0) return 0;
    const uint32_t p = nd.a & 0xFF;
    if (!SkewPred(p)) return 0;
    if (p == 2 || p == 5) nn = true;
    const uint32_t c = tab.cfg[tab.cmap[nd.a >> 8]];
    const uint32_t sym = (c >> 16) & 0xFF;
    if (p == 0 && (c & 0x1000) && sym < (1u << (c & 0xF))) {
      const int32_t v = (int32_t)((uint32_t)UnpackSigned(sym) * nd.b + (uint32_t)nd.splitval);
      if (first) { cv = v; first = false; }
      else if (v != cv) all_const = false;
    } else {
      all_const = false;
    }
    if (!used_y) break;   // every row resolves to this leaf
  }
  if (nn && h > 64 && w > kCarryInts) return 0;
  *needs_n = nn;
  *cval = cv;
  return all_const ? 2 : 1;
}

// One row of a row-static channel decoded by a wavefront that owns ONE section (single frames, small batches).  The recurrence is
// uniform, so it runs on the scalar unit: ANS state, bit buffer and West sample live in scalar registers, the direct-table entry and
// the window words come back from LDS through v_readfirstlane, only the sample store is a vector instruction.  A lone wavefront
// issues one instruction every four to five cycles whichever unit executes it, so what counts here is the instruction count per
// token: sixteen tokens unrolled between top-ups (no loop or address arithmetic per token), the common leaf (multiplier 1, offset
// 0, predictor applied in phase B) without the multiply / offset / West instructions.
#define JXL_RFL(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
typedef const __attribute__((address_space(4))) uint32_t* ConstU32;   // constant address space: uniform reads become scalar loads
template <bool kPlain, class TabPtr>
__device__ __forceinline__ void RowScalar(LaneBits& b, uint32_t& state, TabPtr dt, uint32_t cfg, uint32_t mul, uint32_t off,
                                          bool add_w, uint32_t W, uint32_t& first_out, JXL_GLB int32_t* row, int w) {
  uint32_t s_state = JXL_RFL(state), s_w = JXL_RFL(W), s_first = 0, s_rd = JXL_RFL(b.rd);
  uint64_t s_buf = ((uint64_t)JXL_RFL((uint32_t)(b.buf >> 32)) << 32) | JXL_RFL((uint32_t)b.buf);
  int s_n = (int)JXL_RFL(b.n);
  const uint32_t s_cfg = JXL_RFL(cfg), s_mul = JXL_RFL(mul), s_off = JXL_RFL(off);
  const uint32_t se = s_cfg & 0xF, split = 1u << se, msb = (s_cfg >> 4) & 0xF, lsb = (s_cfg >> 8) & 0xF;
  const bool s_addw = JXL_RFL(add_w ? 1u : 0u) != 0;
  const int sw = (int)JXL_RFL(w);
  auto word = [&]() {
    if (__builtin_expect(s_n <= 32, 0)) { s_buf |= (uint64_t)JXL_RFL(b.ring[__umul24(s_rd & (kRingWords - 1), b.rs)]) << s_n; s_n += 32; s_rd++; }
  };
  auto token = [&](int x) {
    const uint32_t e = JXL_RFL(dt[s_state & 0xFFF]);
    const uint32_t hi = s_state >> 12;
    // freq * hi + offset with freq - 1 stored: (freq - 1) * hi + (hi + offset), the sum kept apart so that the chain from the table
    // entry to the new state is and -> multiply -> add (left alone, the compiler forms freq first: one more dependent operation)
    uint32_t t = hi + ((e >> 12) & 0xFFF);
    asm volatile("" : "+s"(t));
    s_state = (e & 0xFFF) * hi + t;
    if (__builtin_expect(s_state < 65536u, 0)) {
      word();
      s_state = (s_state << 16) | ((uint32_t)s_buf & 0xFFFFu);
      s_buf >>= 16; s_n -= 16;
    }
    const uint32_t sym = e >> 24;
    uint32_t u = sym;
    if (__builtin_expect(sym >= split, 0)) {   // out of line: a taken scalar branch costs far more than a scalar instruction here
      const uint32_t nb = se - (msb + lsb) + ((sym - split) >> (msb + lsb)), nbr = nb > 32 ? 32 : nb;
      word();
      const uint32_t bits = (uint32_t)(s_buf & (((uint64_t)1 << nbr) - 1));
      s_buf >>= nbr; s_n -= (int)nbr;
      const uint32_t low = sym & ((1u << lsb) - 1), top = (1u << msb) | ((sym >> lsb) & ((1u << msb) - 1));
      u = (uint32_t)(((((uint64_t)top << (nb & 63)) | bits) << lsb) | low);
    }
    if constexpr (kPlain) {
      row[x] = UnpackSigned(u);
    } else {
      const uint32_t val = (uint32_t)UnpackSigned(u) * s_mul + s_off + (s_addw ? s_w : 0u);
      row[x] = (int32_t)val;
      s_w = val;
      s_first = x == 0 ? val : s_first;
    }
  };
  int x0 = 0;
  for (; x0 + kTopUpEvery <= sw; x0 += kTopUpEvery) {
    b.rd = s_rd;
    b.TopUp();
#pragma unroll
    for (int k = 0; k < kTopUpEvery; k++) token(x0 + k);
  }
  if (x0 < sw) {
    b.rd = s_rd;
    b.TopUp();
    for (int x = x0; x < sw; x++) token(x);
  }
  state = s_state; b.buf = s_buf; b.n = s_n; b.rd = s_rd;
  first_out = s_first;
}

#include "modular_uniform.h"

// LDS areas of the one-section-per-wavefront per-sample decoder (modular_uniform.h); unused (null) in every other launch shape
struct UniAreas {
  JXL_LDS int32_t* rows = nullptr;   // 3 * rw
  JXL_LDS int32_t* wp = nullptr;     // 10 * (rw + 2), null when the tree does not use the weighted predictor
  JXL_LDS uint32_t* grid = nullptr;  // kUniGridCells records of 16 bytes
  int rw = 0;
  bool use_wp = false;
};

// Phase A of one channel on one lane.  Writes residuals (kChanResid), final samples (kChanFinal) or nothing (kChanConst).
// kGeneric = false: the per-sample path (ModularChannel: weighted predictor, any tree) is not compiled in.  The LF and alpha kernels
// carry both forms: the generic path alone raises their register allocation by half (lf_ans 156 -> 203 VGPRs, alpha_ans 96 -> 135)
// whether or not a stream ever takes it, and these wavefronts sit on their registers for tens of milliseconds while the pixel
// kernels look for room.  The host launches the lean form when every frame of the batch has a row-static tree and plain ANS codes.
template <bool kLds, bool kUni = false, bool kGeneric = true>
__device__ __forceinline__ void DecodeChannelLane(LaneBits& b, uint32_t& state, const CodeTab<kLds>& tab, typename AS<kLds>::Tree tree, int chan, int sid,
                                  int w, int h, int32_t* out_generic, int stride, ChanDesc* desc, int32_t* wp_scratch = nullptr,
                                  const RowBuf<kLds>* lane_rows = nullptr, const UniAreas* uni = nullptr) {
  ChanDesc d;
  d.kind = kChanFinal; d.value = 0; d.pad0 = 0; d.pad1 = 0;
  if (w <= 0 || h <= 0) { *desc = d; return; }
  bool needs_n = false;
  int32_t cval = 0;
  // a tree that references the weighted predictor anywhere keeps every channel on the generic path (its state is per sample)
  // (prefix-coded / LZ77 streams too: their symbols come from the general reader, one AnsGet per sample)
  const int cls = (wp_scratch || tab.slow) ? 0 : ClassifyChannel<kLds>(tab, tree, chan, sid, w, h, &needs_n, &cval);
  if (cls == 2) { d.kind = kChanConst; d.value = cval; *desc = d; return; }
  if (cls == 0) {
    if constexpr (kLds && kUni) {
      // one section per wavefront, every lane alive: the per-sample chain on the scalar unit, the lanes as its neighbour / threshold /
      // division tables (modular_uniform.h); channels it does not take (wide rows, general symbol reader) fall through
      if (uni && uni->rows && ModularChannelUniform(b, state, tab, tree, chan, sid, w, h, out_generic, stride, uni->rows, uni->rw, uni->use_wp ? uni->wp : nullptr,
                                                    uni->grid, kUniGridCells, uni->use_wp)) {
        *desc = d;
        return;
      }
    }
    if constexpr (kGeneric) {
      RowBuf<kLds> rbuf;
      rbuf.rb = nullptr; rbuf.rb_stride = 1; rbuf.rb_width = 0;
      if (lane_rows) rbuf = *lane_rows;   // previous row in LDS instead of re-reading the plane (store -> load round trips)
      ModularChannel<kLds>(b, state, tab, tree, chan, sid, w, h, out_generic, stride, rbuf, wp_scratch);
    } else {
      b.slow_err = 1;   // the host chose the lean form for a stream that needs the other one: reported as a failed section, never decoded wrong
    }
    *desc = d;
    return;
  }
  JXL_GLB int32_t* const out = G(out_generic);
  uint32_t first_prev = 0;   // sample (0, y - 1): what West means at the start of a row
  for (int y = 0; y < h; y++) {
    // row constants: leaf -> cluster, hybrid-uint config, multiplier / offset, predictor
    bool used_y = false;
    const DevTreeNode nd = RowNode(tree, chan, sid, y, &used_y);
    const uint32_t cl = tab.cmap[nd.a >> 8];
    const uint32_t cfg = tab.cfg[cl];
    const typename AS<kLds>::U64 abase = tab.alias + (cl << tab.log_alpha);
    const uint32_t mul = nd.b, off = (uint32_t)nd.splitval;
    const bool add_w = !needs_n && (nd.a & 0xFF) == 1;   // Zero / West rows need no row above: finished inline
    const uint32_t csym = (cfg >> 16) & 0xFF;
    const bool ctok = (cfg & 0x1000) && csym < (1u << (cfg & 0xF));
    const uint32_t cres = (uint32_t)UnpackSigned(csym) * mul + off;
    JXL_GLB int32_t* const row = out + (size_t)y * stride;
    uint32_t W = y ? first_prev : 0u;
    if (ctok) {   // a row of one-symbol tokens: nothing is read from the stream
      for (int x = 0; x < w; x++) {
        const uint32_t val = cres + (add_w ? W : 0u);
        row[x] = (int32_t)val;
        W = val;
      }
      first_prev = cres + (add_w ? (y ? first_prev : 0u) : 0u);
      continue;
    }
    // sixteen tokens per top-up of the bit window; no per-token bookkeeping beyond the decode itself
    uint32_t first = 0;
    if constexpr (kLds) {
      if (tab.direct) {   // one section per wavefront: the scalar-unit loop, table entries by scalar loads
        const uint64_t ga = (uint64_t)(uintptr_t)(tab.direct + (cl << 12));
        const ConstU32 dt = (ConstU32)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ga >> 32)) << 32) |
                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ga));
        if (mul == 1 && off == 0 && needs_n) RowScalar<true, ConstU32>(b, state, dt, cfg, 1u, 0u, false, W, first, row, w);
        else RowScalar<false, ConstU32>(b, state, dt, cfg, mul, off, add_w, W, first, row, w);
        first_prev = first;
        continue;
      }
    }
    uint64_t entry = AnsPrefetch<kLds>(state, abase, tab.log_alpha);
    for (int x0 = 0; x0 < w; x0 += kTopUpEvery) {
      b.TopUp();
      const int xe = min(w, x0 + kTopUpEvery);
      for (int x = x0; x < xe; x++) {
        const uint32_t sym = AnsSymPf<kLds>(b, state, abase, tab.log_alpha, entry);
        const uint32_t val = (uint32_t)UnpackSigned(HybridTail(b, cfg, sym)) * mul + off + (add_w ? W : 0u);
        row[x] = (int32_t)val;
        W = val;
        first = x == 0 ? val : first;
      }
    }
    first_prev = first;
  }
  d.kind = needs_n ? kChanResid : kChanFinal;
  *desc = d;
}

// Phase B: a wavefront applies the predictors of a row-static channel in place.  Lane r owns row y0 + r of a 64-row
// batch, skewed by one sample per row: N comes from lane r-1's previous step (one DPP shuffle), NW is the lane's own
// previous N, W its own previous value.  carry: LDS row (w ints, only used when h > 64) holding the batch's last row.
// Value of `v` in the lane below (lane - 1), by a DPP wave shift (one VALU move; __shfl_up goes through the LDS crossbar,
// and that round trip sat on the recurrence's critical path).  Lane 0 gets 0.  Call with all lanes active.
__device__ __forceinline__ int32_t FromLaneBelow(int32_t v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
// min(W, N) if NW >= max, max if NW <= min, else W + N - NW: the clamped gradient without 64-bit arithmetic (inside the
// interval the sum cannot overflow).
__device__ __forceinline__ int32_t ClampedGradient32(int32_t W, int32_t N, int32_t NW) {
  const int32_t mn = W < N ? W : N, mx = W < N ? N : W;
  return NW >= mx ? mn : (NW <= mn ? mx : (int32_t)((uint32_t)W + (uint32_t)N - (uint32_t)NW));
}
__device__ __forceinline__ uint32_t RowGuess(uint32_t pred, int32_t W, int32_t N, int32_t NW) {
  const uint32_t g = (uint32_t)ClampedGradient32(W, N, NW);
  return pred == 0 ? 0u : (pred == 1 ? (uint32_t)W : (pred == 2 ? (uint32_t)N : g));
}
template <bool kU8Out>
__device__ void PredictWave(const I4* tree, int chan, int sid, int32_t* plane_generic, int stride, int w, int h, int kind, int32_t cvalue,
                            uint8_t* out8_generic, int out_stride, JXL_LDS int32_t* carry, int lane) {
  JXL_GLB int32_t* const plane = G(plane_generic);
  JXL_GLB uint8_t* const out8 = G(out8_generic);
  for (int y0 = 0; y0 < h; y0 += 64) {
    const int nrows = min(64, h - y0);
    const int y = y0 + lane;
    const bool row_active = lane < nrows;
    uint32_t pred = 0;
    if (row_active) { bool u = false; pred = RowNode(tree, chan, sid, y, &u).a & 0xFF; }
    JXL_GLB int32_t* const prow = plane + (size_t)(row_active ? y : y0) * stride;
    const bool more = y0 + 64 < h;
    int32_t W = 0, N = 0, NW = 0, val = 0;
    int32_t r_next = (row_active && lane == 0 && kind == kChanResid) ? prow[0] : cvalue;
    const int steps = w + nrows - 1;
    for (int t = 0; t < steps; t++) {
      const int32_t from_up = FromLaneBelow(val);   // lane r-1's value of the previous step = sample (x, y-1)
      const int x = t - lane;
      const int32_t r = r_next;
      if (row_active && kind == kChanResid && x + 1 >= 0 && x + 1 < w) r_next = prow[x + 1];   // in flight during this step
      if (row_active && x >= 0 && x < w) {
        const int32_t n_in = lane == 0 ? (y ? carry[x] : 0) : from_up;
        if (x == 0) { W = y ? n_in : 0; N = W; NW = W; }
        else if (y) { NW = N; N = n_in; }
        else { NW = W; N = W; }
        const uint32_t guess = RowGuess(pred, W, N, NW);
        val = (int32_t)((uint32_t)r + guess);
        if (kU8Out) out8[(size_t)y * out_stride + x] = (uint8_t)(val < 0 ? 0 : (val > 255 ? 255 : val));
        else prow[x] = val;
        if (more && lane == nrows - 1) carry[x] = val;
        W = val;
      }
    }
  }
}

// The same recurrence with all global traffic coalesced: the channel rectangle is walked in 64-row batches x 64-column
// blocks; a block is loaded into an LDS tile row by row (lanes along x), the skewed pass runs inside the tile, and the
// results leave it row by row again.  (Reading 64 different rows per step straight from HBM costs a cache line per
// 4-byte sample once thousands of wavefronts run: measured 404 MB of traffic per 4K alpha plane instead of 41 MB.)
// W / N / NW live in registers and simply carry over from block to block.  tile: 64 x 64 ints of LDS, pitch 64 ON PURPOSE: the
// skewed pass has lane r at column t - r, i.e. word r * pitch + t - r; with the usual padded pitch of 65 that is r * 64 + t - every
// lane in the same bank (the SQ counters showed 83 % of this kernel's LDS cycles as bank conflicts); 63 * r spreads over all banks,
// and the row-wise load / store phases are conflict-free with any pitch.
constexpr int kTilePitch = 64;
// The tile belongs to ONE wavefront: its LDS operations execute in program order, so the phases only need the compiler kept in line
// (a workgroup barrier here would also forbid callers whose wavefronts do different things, like lf_finish_kernel).
__device__ __forceinline__ void TileSync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// The step loop of PredictWaveTiled for a band whose rows all use the clamped gradient (what alpha and LF channels are written with):
// the same recurrence without the per-row predictor select, with every lane reading its next residual / the carried row at a clamped
// index instead of under an exec mask, and the carried row copied out of the tile afterwards instead of by one lane per step.
// kFirstTile: the tile starts at column 0 (a row's first sample has no West / North-West); kTopBand: lane 0 is the channel's top row.
template <bool kFirstTile, bool kTopBand>
__device__ __forceinline__ void GradientTileSteps(JXL_LDS int32_t* trow, const JXL_LDS int32_t* carry_x0, int ncols, int nrows, int lane, bool row_active,
                                                  int32_t& W, int32_t& N, int32_t& NW, int32_t& val) {
  const int steps = ncols + nrows - 1;
  const bool top = kTopBand && lane == 0;
  int32_t r_next = trow[(0 - lane) & 63], up0_next = kTopBand ? 0 : carry_x0[(0 - lane) & 63];   // column of step 0 (wrapped: see below)
  for (int t = 0; t < steps; t++) {
    const int32_t from_up = FromLaneBelow(val);   // lane r-1's value of the previous step = sample (x, y-1)
    const int c = t - lane;
    const int32_t r = r_next, up0 = up0_next;
    // lanes outside their row's span read too, at the WRAPPED column: a clamped index would send all of them to one LDS bank
    // (row pitch 64 words: column k of every row shares a bank), which made this loop slower than the masked reads it replaced
    const int cn = (c + 1) & 63;
    r_next = trow[cn];
    if (!kTopBand) up0_next = carry_x0[cn];
    if (row_active && (unsigned)c < (unsigned)ncols) {
      const int32_t n_in = lane == 0 ? up0 : from_up;
      int32_t n = n_in;
      if (kTopBand) n = top ? W : n_in;            // top row: North = North-West = West
      int32_t nw = kTopBand ? (top ? W : N) : N;
      if (kFirstTile && c == 0) {                  // first column: West = North = North-West = the sample above (0 in the top row)
        const int32_t w0 = top ? 0 : n_in;
        W = w0; n = w0; nw = w0;
      }
      val = (int32_t)((uint32_t)r + (uint32_t)ClampedGradient32(W, n, nw));
      trow[c] = val;
      W = val;
      N = n;
    }
  }
  NW = N;
}

template <bool kU8Out>
__device__ void PredictWaveTiled(const I4* tree, int chan, int sid, int32_t* plane_generic, int stride, int w, int h, int kind, int32_t cvalue,
                                 uint8_t* out8_generic, int out_stride, JXL_LDS int32_t* carry, JXL_LDS int32_t* tile, int lane) {
  JXL_GLB int32_t* const plane = G(plane_generic);
  JXL_GLB uint8_t* const out8 = G(out8_generic);
  for (int y0 = 0; y0 < h; y0 += 64) {
    const int nrows = min(64, h - y0);
    const int y = y0 + lane;
    const bool row_active = lane < nrows;
    uint32_t pred = 0;
    if (row_active) { bool u = false; pred = RowNode(tree, chan, sid, y, &u).a & 0xFF; }
    const bool more = y0 + 64 < h;
    int32_t W = 0, N = 0, NW = 0, val = 0;
    for (int x0 = 0; x0 < w; x0 += 64) {
      const int ncols = min(64, w - x0);
      // sixteen row loads in flight, then sixteen LDS stores (a load-store-per-row loop waited for every row: 64 serial memory
      // round trips per block, which was most of this function's time)
      if (kind == kChanResid && lane < ncols)
        for (int r0 = 0; r0 < nrows; r0 += 16) {
          int32_t v[16];
#pragma unroll
          for (int i = 0; i < 16; i++) v[i] = plane[(size_t)(y0 + min(r0 + i, nrows - 1)) * stride + x0 + lane];
#pragma unroll
          for (int i = 0; i < 16; i++) if (r0 + i < nrows) tile[(r0 + i) * kTilePitch + lane] = v[i];
        }
      TileSync();
      const int steps = ncols + nrows - 1;
      // The residual of the step after this one and (lane 0) the sample above it are requested a step early: LDS latency
      // then overlaps the arithmetic instead of adding to the recurrence.
      JXL_LDS int32_t* const trow = tile + lane * kTilePitch;
      const bool grad_band = kind == kChanResid && __ballot(row_active && pred != 5) == 0;   // (uniform)
      if (grad_band) {
        if (x0 == 0) {
          if (y0 == 0) GradientTileSteps<true, true>(trow, carry + x0, ncols, nrows, lane, row_active, W, N, NW, val);
          else GradientTileSteps<true, false>(trow, carry + x0, ncols, nrows, lane, row_active, W, N, NW, val);
        } else {
          if (y0 == 0) GradientTileSteps<false, true>(trow, carry + x0, ncols, nrows, lane, row_active, W, N, NW, val);
          else GradientTileSteps<false, false>(trow, carry + x0, ncols, nrows, lane, row_active, W, N, NW, val);
        }
        TileSync();
        if (more && lane < ncols) carry[x0 + lane] = tile[(nrows - 1) * kTilePitch + lane];   // the band's last row, for the band below
      } else {
      const bool top = y == 0, keeps_carry = more && lane == nrows - 1;
      int32_t r_next = cvalue, up0_next = 0;
      if (lane == 0 && row_active) {
        if (kind == kChanResid) r_next = trow[0];
        if (y) up0_next = carry[x0];
      }
      for (int t = 0; t < steps; t++) {
        const int32_t from_up = FromLaneBelow(val);   // lane r-1's value of the previous step = sample (x, y-1)
        const int c = t - lane;
        const int32_t r = r_next, up0 = up0_next;
        if (row_active && c + 1 >= 0 && c + 1 < ncols) {
          if (kind == kChanResid) r_next = trow[c + 1];
          if (lane == 0 && y) up0_next = carry[x0 + c + 1];
        }
        if (row_active && (unsigned)c < (unsigned)ncols) {
          // neighbours without branches: first column -> W = N = NW = the sample above (0 in the top row); top row -> N = NW = W
          const int32_t n_in = lane == 0 ? up0 : from_up;
          const bool first = (x0 | c) == 0;
          const int32_t w0 = top ? 0 : n_in;
          NW = first ? w0 : (top ? W : N);
          N = first ? w0 : (top ? W : n_in);
          W = first ? w0 : W;
          val = (int32_t)((uint32_t)r + RowGuess(pred, W, N, NW));
          trow[c] = val;
          if (keeps_carry) carry[x0 + c] = val;
          W = val;
        }
      }
      }
      TileSync();
      if (kU8Out && ncols == 64 && ((out_stride | x0) & 3) == 0 && ((uintptr_t)out8_generic & 3) == 0) {
        // four samples per lane: one 16-byte LDS read, one packed 4-byte store; a wavefront instruction covers four rows
        const int l16 = lane & 15, lr = lane >> 4;
#pragma unroll 4
        for (int r0 = 0; r0 < nrows; r0 += 4) {
          const int r = r0 + lr;
          if (r < nrows) {
            const I4 v = *(const JXL_LDS I4*)(tile + r * kTilePitch + 4 * l16);
            const uint32_t px = (uint32_t)min(max(v.x, 0), 255) | (uint32_t)min(max(v.y, 0), 255) << 8 | (uint32_t)min(max(v.z, 0), 255) << 16 |
                                (uint32_t)min(max(v.w, 0), 255) << 24;
            *(JXL_GLB uint32_t*)(out8 + (size_t)(y0 + r) * out_stride + x0 + 4 * l16) = px;
          }
        }
      } else if (lane < ncols) {
        for (int r0 = 0; r0 < nrows; r0 += 16) {
          int32_t v[16];
#pragma unroll
          for (int i = 0; i < 16; i++) v[i] = tile[min(r0 + i, nrows - 1) * kTilePitch + lane];
#pragma unroll
          for (int i = 0; i < 16; i++)
            if (r0 + i < nrows) {
              if (kU8Out) out8[(size_t)(y0 + r0 + i) * out_stride + x0 + lane] = (uint8_t)(v[i] < 0 ? 0 : (v[i] > 255 ? 255 : v[i]));
              else plane[(size_t)(y0 + r0 + i) * stride + x0 + lane] = v[i];
            }
        }
      }
      TileSync();
    }
  }
}

// Applies a channel descriptor to an int32 plane (phase B for planes that stay int32).
__device__ void FinishChannelI32(const ChanDesc d, const I4* tree, int chan, int sid, int32_t* plane, int stride, int w, int h,
                                 JXL_LDS int32_t* carry, int lane, JXL_LDS int32_t* tile = nullptr) {
  if (w <= 0 || h <= 0) return;
  if (d.kind == kChanConst) {
    for (int i = lane; i < w * h; i += 64) plane[(size_t)(i / w) * stride + (i % w)] = d.value;
  } else if (d.kind == kChanResid) {
    if (tile) PredictWaveTiled<false>(tree, chan, sid, plane, stride, w, h, d.kind, d.value, nullptr, 0, carry, tile, lane);
    else PredictWave<false>(tree, chan, sid, plane, stride, w, h, d.kind, d.value, nullptr, 0, carry, lane);
  }
}

template <bool kLds>
struct ModTables {
  CodeTab<kLds> tab;
  typename AS<kLds>::Tree tree;
};

// Stages the MA tree + modular code of `im` (LDS variant) or points at them in global memory.
template <bool kLds>
__device__ __forceinline__ void LoadModTables(const DevImage& im, uint8_t* smem, size_t off, ModTables<kLds>& t, int tid, int nt,
                                              bool one_section = false) {
  if constexpr (kLds) {
    JXL_LDS uint8_t* lds = (JXL_LDS uint8_t*)smem;
    off = (off + 15) & ~(size_t)15;
    JXL_LDS I4* st = (JXL_LDS I4*)(lds + off); off += (size_t)im.tree_size * sizeof(DevTreeNode);
    for (int i = tid; i < im.tree_size; i += nt) st[i] = ((const I4*)im.tree)[i];
    t.tree = st;
    StageCode(lds, off, im.mcode, t.tab, tid, nt, one_section);
    __syncthreads();
  } else {
    GlobalCode(im.mcode, t.tab);
    t.tree = (const I4*)im.tree;
  }
}

}  // namespace

// ------------------------------------------------------------------ LF groups, phase A: one lane per LF group
// (workgroup = one wavefront = up to 64 LF groups of ONE image, tables in LDS).  LF coefficients (3 channels) and the HF
// metadata (chroma-from-luma maps, block info, sharpness) of the group are decoded into lfq / binfo scratch.
template <bool kLds, bool kGeneric = true>
__global__ __launch_bounds__(64) void lf_ans_kernel(const DevImage* imgs, const SectionTask* tasks, int slots, int scalar_rows) {
  JXL_SERIAL_PRIO();
  extern __shared__ __align__(16) uint8_t smem[];
  const SectionTask task = tasks[blockIdx.x];
  const DevImage& im = imgs[task.image];
  ModTables<kLds> mt;
  // `slots` lanes of the wavefront decode (the launch's sections per workgroup): the bit windows take slots * 128 B of LDS, not 8 KB -
  // LDS is what decides whether this kernel can share a CU with the HF decoder of the batch before
  LoadModTables<kLds>(im, smem, (size_t)slots * kRingWords * 4, mt, threadIdx.x, 64, slots == 1 && scalar_rows);
  const int lane = threadIdx.x;
  if (lane >= task.count || lane >= slots) return;
  const int g = task.first + lane;
  const int gx = g % im.xlf, gy = g / im.xlf;
  const int bx0 = gx * kLfGroupBlocks, by0 = gy * kLfGroupBlocks;
  const int bw = min(kLfGroupBlocks, im.w8 - bx0), bh = min(kLfGroupBlocks, im.h8 - by0);
  const int tw = (bw + 7) / 8, th = (bh + 7) / 8;
  int32_t* scratch = im.binfo + (size_t)g * kBinfoInts;
  ChanDesc* desc = im.lf_desc + (size_t)g * 8;
  const int sid_meta = 1 + 2 * im.nlf + g;
  const int lf_sec = im.single ? 0 : 1 + g;
  const uint64_t start_bits = im.single ? im.lf_start_bits : im.sec_off[lf_sec] * 8;
  LaneBits b;
  b.Init(im.cs, im.cs_size, im.alpha_in_global ? im.lf_start_bits : start_bits, (JXL_LDS uint32_t*)smem + lane, (uint32_t)slots);
  uint32_t state = 0, err = 0, count = 1;
  int32_t* const wps = im.wp_lf ? im.wp_lf + (size_t)g * kWpLfInts : nullptr;
  uint32_t* const lzw = im.lz_lf ? im.lz_lf + ((size_t)g << 20) : nullptr;   // LZ77 window of this lane's streams
  // Eight channels, one loop (a single inlined copy of the channel decoder; a second call site makes it a real call, which puts the
  // bit reader in scratch memory): -1 the alpha channel of a frame that fits one group - it is coded in the GlobalModular part of
  // LfGlobal (stream 0), which the LF group follows directly (one-section frames) or in a section of its own (progressive frames:
  // one section per pass); 0-2 LF coefficients; 3-6 HF metadata
#pragma unroll 1
  for (int i = im.alpha_in_global ? -1 : 0; i < 7 && !err; i++) {
    if (i == -1) {
      b.SetLz(lzw, 20, (uint32_t)im.w);
      b.lz_copy = 0; b.lz_done = 0;
      state = InitAnsState(b, mt.tab);
    }
    if (i == 0) {
      if (im.alpha_in_global) {
        if (state != 0x130000u || b.slow_err) { err |= kErrBitstream; break; }
        if (!im.single) {
          if (im.lf_start_bits + b.Consumed() > (im.sec_off[0] + im.sec_size[0]) * 8) { err |= kErrBitstream; break; }
          b.Init(im.cs, im.cs_size, start_bits, (JXL_LDS uint32_t*)smem + lane, (uint32_t)slots);
        }
      }
      im.lf_extra[g] = (uint8_t)b.Read(2);
      if (b.Read(4) != 3) { err |= kErrUnsupportedHeader; break; }
      // the LF stream: three channels bw wide
      b.SetLz(lzw, 20, (uint32_t)bw);
      b.lz_copy = 0; b.lz_done = 0;
      state = InitAnsState(b, mt.tab);
    }
    if (i == 3) {
      if (state != 0x130000u) { err |= kErrBitstream; break; }
      count = b.Read(CeilLog2D((uint32_t)(bw * bh))) + 1;
      if (count > (uint32_t)(bw * bh)) { err |= kErrBlockLayout; break; }
      if (b.Read(4) != 3) { err |= kErrUnsupportedHeader; break; }
      // the HF metadata stream: chroma-from-luma maps (tw wide), block info (count wide), sharpness (bw wide)
      b.SetLz(lzw, 20, max(max((uint32_t)tw, count), (uint32_t)bw));
      b.lz_copy = 0; b.lz_done = 0;
      state = InitAnsState(b, mt.tab);
    }
    int chan, sid, w, h, stride;
    int32_t* out;
    ChanDesc* cd = desc + max(i, 0);
    if (i < 0) {
      chan = 0; sid = 0; w = im.w; h = im.h; stride = im.w; out = im.alpha32; cd = im.alpha_desc;
    } else if (i < 3) {
      const int c = i == 0 ? 1 : (i == 1 ? 0 : 2);   // modular channel order Y, X, B
      chan = i; sid = 1 + g; w = bw; h = bh; stride = im.w8;
      out = im.lfq[c] + (size_t)by0 * im.w8 + bx0;
    } else {
      chan = i - 3; sid = sid_meta;
      if (i < 5) { w = tw; h = th; stride = tw; out = scratch + (i - 3) * 1024; }
      else if (i == 5) { w = (int)count; h = 2; stride = (int)count; out = scratch + 2048; }
      else { w = bw; h = bh; stride = bw; out = scratch + 2048 + 2 * 65536; }
    }
    DecodeChannelLane<kLds, false, kGeneric>(b, state, mt.tab, mt.tree, chan, sid, w, h, out, stride, cd, wps);
  }
  if (!err && (state != 0x130000u || b.slow_err || start_bits + b.Consumed() > (im.sec_off[lf_sec] + im.sec_size[lf_sec]) * 8)) err |= kErrBitstream;
  if (im.single) im.lf_end_bits[0] = start_bits + b.Consumed();
  im.lf_count[g] = err ? 0u : count;
  if (err) SetError(im, err, 1, g);
}

// ------------------------------------------------------------------ LF groups, phase B: one workgroup (4 wavefronts) per LF group
// Predictors of the row-static channels, then the chroma-from-luma / sharpness maps and the varblock placement.
__global__ __launch_bounds__(256) void lf_finish_kernel(const DevImage* __restrict__ imgs, const SectionTask* tasks) {
  __shared__ int32_t s_carry[4][256];
  __shared__ uint32_t s_count, s_flag;
  __shared__ uint32_t s_part[256];
  __shared__ uint32_t s_cov[256 * 8];   // coverage bitmap of the LF group's 256 x 256 cells
  const DevImage& im = imgs[tasks[blockIdx.x].image];
  const int g = tasks[blockIdx.x].first;
  const uint32_t count = im.lf_count[g];
  if (!count) return;   // phase A failed (already reported)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int gx = g % im.xlf, gy = g / im.xlf;
  const int bx0 = gx * kLfGroupBlocks, by0 = gy * kLfGroupBlocks;
  const int bw = min(kLfGroupBlocks, im.w8 - bx0), bh = min(kLfGroupBlocks, im.h8 - by0);
  const int tw = (bw + 7) / 8, th = (bh + 7) / 8;
  int32_t* scratch = im.binfo + (size_t)g * kBinfoInts;
  int32_t* s_x = scratch;
  int32_t* s_b = scratch + 1024;
  int32_t* s_info = scratch + 2048;
  int32_t* s_sharp = scratch + 2048 + 2 * 65536;
  const ChanDesc* desc = im.lf_desc + (size_t)g * 8;
  const I4* tree = (const I4*)im.tree;
  const int sid_meta = 1 + 2 * im.nlf + g;
  JXL_LDS int32_t* carry = (JXL_LDS int32_t*)s_carry[wave];
  if (wave < 3) {
    const int chan_of[3] = {1, 0, 2};
    // (measured: an LDS tile per channel wavefront - 48 KB more per workgroup - doubles this kernel's time in a batch and does not
    // change the single-frame time; the untiled pass stays)
    FinishChannelI32(desc[wave], tree, wave, 1 + g, im.lfq[chan_of[wave]] + (size_t)by0 * im.w8 + bx0, im.w8, bw, bh, carry, lane);
  } else {
    FinishChannelI32(desc[3], tree, 0, sid_meta, s_x, tw, tw, th, carry, lane);
    FinishChannelI32(desc[4], tree, 1, sid_meta, s_b, tw, tw, th, carry, lane);
    FinishChannelI32(desc[5], tree, 2, sid_meta, s_info, (int)count, (int)count, 2, carry, lane);
    FinishChannelI32(desc[6], tree, 3, sid_meta, s_sharp, bw, bw, bh, carry, lane);
  }
  __threadfence_block();
  __syncthreads();
  // chroma-from-luma maps and sharpness: parallel copies with range checks
  uint32_t err = 0;
  const int tx0 = bx0 / 8, ty0 = by0 / 8;
  for (int i = tid; i < tw * th; i += 256) {
    const int x = i % tw, y = i / tw;
    const int vx = s_x[i], vb = s_b[i];
    if (vx < -128 || vx > 127 || vb < -128 || vb > 127) err |= kErrRange;
    im.ytox[(size_t)(ty0 + y) * im.wt + tx0 + x] = (int8_t)vx;
    im.ytob[(size_t)(ty0 + y) * im.wt + tx0 + x] = (int8_t)vb;
  }
  for (int i = tid; i < bw * bh; i += 256) {
    const int x = i % bw, y = i / bw;
    int sh = s_sharp[i];
    if (sh < 0 || sh > 7) { err |= kErrRange; sh = 0; }
    im.sharp[(size_t)(by0 + y) * im.w8 + bx0 + x] = (uint8_t)sh;
  }
  __syncthreads();   // s_sharp has been consumed: its scratch is reused for the block positions
  // ---- varblock placement: every block goes to the first cell, in raster order, that is not covered yet.  Serial by definition -
  // and a one-block-at-a-time walk cost 0.4 us per block, 25 ms for an LF group of 65536 8x8 blocks - but only ACROSS rows: the
  // blocks that start in row y are exactly the next ones of the sequence whose widths add up to the row's free cells, and block j
  // of them starts at the free cell number (sum of the widths before it).  So: prefix sums of the widths once, then per row every
  // thread takes one candidate block, finds its cell by a rank-select in the row's coverage words, checks that its cells are
  // consecutive free ones and marks the rows below it.  One barrier per row.
  int32_t* s_pos = s_sharp;
  uint32_t* const s_pre = (uint32_t*)(scratch + 2048 + 3 * 65536);   // s_pre[i]: cells taken by the widths of blocks [0, i)
  if (tid == 0) { s_count = 0; s_flag = 0; }
  for (int i = tid; i < 256 * 8; i += 256) s_cov[i] = 0;
  {
    const uint32_t per = (count + 255) / 256, i0 = min(count, (uint32_t)tid * per), i1 = min(count, i0 + per);
    uint32_t sum = 0;
    for (uint32_t i = i0; i < i1; i++) {
      const int es = s_info[i], eq = 1 + s_info[count + i];
      const bool ok = es >= 0 && es < kNumStrategies && eq >= 1 && eq <= 256;
      if (!ok) err |= kErrBlockLayout;
      // AFV0..AFV3: the strategy ids are only known here (the block-metadata stream is decoded on the GPU), so this is where the
      // frame is refused; the group keeps no varblock (s_count = 0 below), nothing downstream reconstructs it
      if (es >= 14 && es <= 17) err |= kErrUnsupportedTransform | kErrBlockLayout;
      sum += ok ? 1u << d_log2cx[es] : 1u;
    }
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
      uint32_t acc = 0;
      for (int i = 0; i < 256; i++) { const uint32_t v = s_part[i]; s_part[i] = acc; acc += v; }
      s_pre[count] = acc;
    }
    __syncthreads();
    uint32_t base = s_part[tid];
    for (uint32_t i = i0; i < i1; i++) {
      const int es = s_info[i];
      s_pre[i] = base;
      base += (es >= 0 && es < kNumStrategies) ? 1u << d_log2cx[es] : 1u;
    }
    if (err & kErrBlockLayout) atomicOr(&s_flag, 1u);
  }
  __threadfence_block();
  __syncthreads();
  uint32_t num = 0;
  bool layout_bad = false;
  if (!s_flag) {
    const int words = (bw + 31) >> 5;
    for (int y = 0; y < bh; y++) {
      // the row's free cells: words of free bits and their running counts (every thread computes the same eight)
      uint32_t pre[9];
      pre[0] = 0;
#pragma unroll
      for (int wi = 0; wi < 8; wi++) {
        uint32_t fr = 0;
        if (wi < words) {
          fr = ~s_cov[y * 8 + wi];
          const int lim = min(32, bw - wi * 32);
          if (lim < 32) fr &= (1u << lim) - 1;
        }
        pre[wi + 1] = pre[wi] + (uint32_t)__popc(fr);
      }
      const uint32_t nfree = pre[8];
      bool inrow = false, bad = false;
      if (nfree) {
        const uint32_t j = num + (uint32_t)tid;
        uint32_t off = 0;
        if (j < count) { off = s_pre[j] - s_pre[num]; inrow = off < nfree; }
        if (inrow) {
          const int es = s_info[j];
          const int lcx = d_log2cx[es], lcy = d_log2cy[es], cx = 1 << lcx, cy = 1 << lcy;
          // rank-select: the word that holds free cell number `off`, then the position of that set bit
          int wi = 0;
          uint32_t before = 0;
#pragma unroll
          for (int k = 1; k < 8; k++) { const bool past = pre[k] <= off; wi += past ? 1 : 0; before = past ? pre[k] : before; }
          uint32_t word = ~s_cov[y * 8 + wi];
          const int lim = min(32, bw - wi * 32);
          if (lim < 32) word &= (1u << lim) - 1;
          uint32_t rem = off - before;
          int xb = 0;
#pragma unroll
          for (int sft = 16; sft; sft >>= 1) {
            const uint32_t c = (uint32_t)__popc((word >> xb) & ((1u << sft) - 1));
            if (rem >= c) { rem -= c; xb += sft; }
          }
          const int x = wi * 32 + xb;
          const uint32_t mask = (cx == 32 ? 0xFFFFFFFFu : ((1u << cx) - 1)) << xb;
          if (off + (uint32_t)cx > nfree || x + cx > bw || y + cy > bh || xb + cx > 32 || (y & 31) + cy > 32 || (word & mask) != mask) {
            bad = true;
          } else {
            for (int iy = 1; iy < cy; iy++)   // (row y itself is finished after this step: nobody reads its coverage again)
              if (atomicOr(&s_cov[(y + iy) * 8 + wi], mask) & mask) bad = true;
            s_pos[j] = x | y << 8;
          }
        }
      }
      // both reductions are barriers: the marks of this row are visible to the next, and every thread takes the same exit
      const int n_row = __syncthreads_count(inrow);
      if (__syncthreads_or(bad || (nfree && !n_row))) { layout_bad = true; break; }   // (free cells left but no block left)
      num += (uint32_t)n_row;
    }
  }
  __threadfence_block();
  __syncthreads();
  if (s_flag || layout_bad || num != count) err |= kErrBlockLayout;
  if (tid == 0) s_count = (err & kErrBlockLayout) ? 0u : num;
  __syncthreads();
  const uint32_t placed = s_count;
  for (uint32_t i = tid; i < placed; i += 256) {
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
  if (err) SetError(im, err, 2, g);
}

// ------------------------------------------------------------------ HF coefficients
// Pre-pass (fully parallel, one wavefront per group): the varblocks of the group in decode order, each with the block
// context of its three channels, so that the serial token loop below never waits on cellinfo / raw-quant / context-map
// loads: it streams 8-byte descriptors, fetched one block ahead.
//   one word: bx | by << 5 | log2cx << 10 | log2cy << 13 | block context of Y << 16 | X << 21 | B << 26   (at most 16 block contexts)
__global__ __launch_bounds__(64) void hf_blocklist_kernel(const DevImage* __restrict__ imgs) {
  const DevImage& im = imgs[blockIdx.y];
  const int g = blockIdx.x;
  if (g >= im.ng || im.is_modular) return;
  const int lane = threadIdx.x;
  const int gx = g % im.xg, gy = g / im.xg;
  const int bx0 = gx * kGroupBlocks, by0 = gy * kGroupBlocks;
  const int bw = min(kGroupBlocks, im.w8 - bx0), bh = min(kGroupBlocks, im.h8 - by0);
  uint32_t* list = im.blk_list + (size_t)g * 1024;
  const int n_qf = im.n_qf;
  uint32_t total = 0;
  for (int it = 0; it < 16; it++) {
    const int idx = it * 64 + lane;
    const int bx = idx & 31, by = idx >> 5;
    bool first = false;
    uint32_t info = 0;
    if (bx < bw && by < bh) {
      info = im.cellinfo[(size_t)(by0 + by) * im.w8 + bx0 + bx];
      first = (info & 0x8003FF00u) == 0x80000000u;
    }
    const uint64_t m = __ballot(first);
    if (first) {
      const uint32_t pos = total + (uint32_t)__popcll(m & (((uint64_t)1 << lane) - 1));
      const uint32_t s = info & 0xFF, lcx = (info >> 18) & 7, lcy = (info >> 21) & 7;
      const uint32_t ord = d_order_bucket[s];
      const uint32_t rq = im.rawq[(size_t)(by0 + by) * im.w8 + bx0 + bx];
      uint32_t qf_idx = 0;
      for (int i = 0; i < n_qf; i++) qf_idx += rq > im.qf_thr[i];
      uint32_t lf_idx = 0;
      if (im.num_lf_ctx > 1) {   // thresholds on the quantised LF of the block's first cell; X, B, Y order of combination
        const size_t cell = (size_t)(by0 + by) * im.w8 + bx0 + bx;
        uint32_t ix = 0, iy = 0, ib = 0;
        for (int i = 0; i < im.n_lf_thr[0]; i++) ix += im.lfq[0][cell] > im.lf_thr[0][i];
        for (int i = 0; i < im.n_lf_thr[1]; i++) iy += im.lfq[1][cell] > im.lf_thr[1][i];
        for (int i = 0; i < im.n_lf_thr[2]; i++) ib += im.lfq[2][cell] > im.lf_thr[2][i];
        lf_idx = (ix * (uint32_t)(im.n_lf_thr[2] + 1) + ib) * (uint32_t)(im.n_lf_thr[1] + 1) + iy;
      }
      uint32_t ctxs = 0;
      for (int ci = 0; ci < 3; ci++) {
        const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
        const uint32_t cprime = c < 2 ? (c ^ 1) : 2;
        ctxs |= ((uint32_t)im.block_ctx_map[((cprime * kNumOrders + ord) * (n_qf + 1) + qf_idx) * im.num_lf_ctx + lf_idx] & 31u) << (5 * ci);
      }
      list[pos] = (uint32_t)bx | (uint32_t)by << 5 | lcx << 10 | lcy << 13 | ctxs << 16;
    }
    total += (uint32_t)__popcll(m);
  }
  if (lane == 0) im.blk_count[g] = total;
}

// One lane per group section (lane_stride spreads sections over wavefronts).  The loop decodes exactly one token per
// iteration: the number-of-nonzeros token of a (block, channel) or one coefficient token.  Per-lane state that the
// context model needs (non-zero counts of the row above / the cell to the left) is a 32-entry column buffer per
// channel in LDS: col[x] = value of the last decoded block covering column x, which is both "above" and "left".
// Output: every NON-ZERO coefficient becomes one 32-bit entry (scan position | value << 16) appended to the group's list -
// sequential 4-byte stores, no coefficient order in the loop, nothing to pre-zero; the (first entry, count) pair of each
// (block, channel) goes to cblk at the block's origin cell.  recon_tile_kernel scatters the entries into its LDS tile.
// kRing: words of the per-lane bit window.  32 (top-up every 16 tokens, 8 queued descriptors: 288 B of LDS per lane) is the
// faster loop for one frame; 16 (top-up every 8 tokens, 4 descriptors: 192 B per lane) lets more workgroups share a CU, which is
// what matters when a batch launches more workgroups than the chip has CUs.  `nslots` = lanes the LDS arrays are laid out for
// (the task's section count rounded up to four: every workgroup uses what ITS image's tables leave of the launch's LDS).
// (the four serial decoders do NOT take the DevImage array as __restrict__ like the other kernels do: with it the compiler re-reads
// frame fields through the scalar cache inside the token loops - 101 scalar loads instead of 24 in this kernel - and a lone section's
// chain gets longer: hf_decode of one 4K frame 12.3 -> 12.8 ms)
template <bool kLds, int kRing>
__global__ __launch_bounds__(512) void hf_decode_kernel(const DevImage* imgs, const SectionTask* tasks, int lane_stride) {
  JXL_SERIAL_PRIO();
  constexpr int kTop = kRing / 2;   // tokens between top-ups: a token consumes at most 48 bits and starts at most one block ...
  constexpr int kQ = kRing / 4;     // ... so kTop tokens never outrun kRing - kRing / 4 + 1 words / kTop / 3 + 1 descriptors
  typedef LaneBitsT<kRing, kRing / 4> Bits;
  extern __shared__ __align__(16) uint8_t smem[];
  const SectionTask task = tasks[blockIdx.x];
  const DevImage& im = imgs[task.image];
  const int per_wave = 64 / lane_stride;
  const int nslots = (task.count + 3) & ~3;
  CodeTab<kLds> tab;
  typename AS<kLds>::U8 nnz_tab;
  JXL_LDS uint8_t* nzcol;
  JXL_LDS uint32_t* ring_base;
  JXL_LDS uint32_t* descq;   // per lane: queue of the next 2 * kQ varblock descriptors, entry j at descq[(j & (2 * kQ - 1)) * nslots + slot]
  {
    JXL_LDS uint8_t* lds = (JXL_LDS uint8_t*)smem;
    size_t off = 0;
    ring_base = (JXL_LDS uint32_t*)lds; off += (size_t)nslots * kRing * 4;
    descq = (JXL_LDS uint32_t*)(lds + off); off += (size_t)nslots * 2 * kQ * 4;
    nzcol = lds + off; off += (size_t)nslots * 96;
    if constexpr (kLds) {
      off = StageCode(lds, off, im.acode, tab, threadIdx.x, blockDim.x);
      JXL_LDS uint8_t* sn = lds + off; off += 64;
      if (threadIdx.x < 64) sn[threadIdx.x] = d_nnz_ctx[threadIdx.x];
      nnz_tab = sn;
      __syncthreads();
    } else {
      GlobalCode(im.acode, tab);
      nnz_tab = d_nnz_ctx;
    }
  }
  // Active lanes are the FIRST 64/lane_stride lanes of every wavefront: a wave64 whose upper 32 lanes are idle issues
  // each vector instruction in one pass instead of two.
  if ((int)(threadIdx.x & 63) >= per_wave) return;
  const int si = (threadIdx.x >> 6) * per_wave + (threadIdx.x & 63);
  if (si >= task.count || si >= nslots) return;
  JXL_LDS uint8_t* const col = nzcol + si;   // col[(c * 32 + x) * nslots]
  const int g = im.hf_order[task.first + si];   // lanes in order of section size (see the task table in decoder.cc)
  const int sec = im.single ? 0 : im.hf_sec_base + g;
  const uint64_t sec_bits = im.single ? im.hf_start_bits : im.sec_off[sec] * 8;
  Bits b;
  b.Init(im.cs, im.cs_size, sec_bits, ring_base + si, (uint32_t)nslots);
  const uint32_t preset = b.Read(CeilLog2D((uint32_t)im.num_presets));
  const uint32_t nbc = im.num_block_ctx;
  const uint32_t ctx_offset = preset * nbc * 495;
  uint32_t err = preset >= (uint32_t)im.num_presets ? (uint32_t)kErrBitstream : 0u;
  if (im.lz_hf) b.SetLz(im.lz_hf + ((size_t)g << 18), 18, 0);   // one-dimensional stream: plain distances
  uint32_t state = InitAnsState(b, tab);
  const int gx = g % im.xg, gy = g / im.xg;
  // block descriptors: staged through a small LDS queue that is topped up together with the bit window
  const JXL_GLB uint32_t* const list = G(im.blk_list + (size_t)g * 1024);
  const uint32_t nblk = im.blk_count[g];
  JXL_LDS uint32_t* const dq = descq + si;
  constexpr uint32_t dqmask = 2u * kQ - 1;
  uint32_t bi = 0, dfilled = 0, it = 0, dpend_n = 0;
  uint32_t dpend[kQ];
#pragma unroll
  for (int i = 0; i < kQ; i++) dpend[i] = 0u;
  // output: the group's entry list and the per-(block, channel) index (no global load may sit in the token loop - its wait would
  // also wait for the stores - so every pointer is formed up front)
  JXL_GLB uint32_t* const ent = G(im.centries) + (size_t)(g - im.centries_g0) * kGroupEntriesCap;
  const uint32_t ncells = (uint32_t)im.w8 * (uint32_t)im.h8;
  JXL_GLB U2* const cblk = (JXL_GLB U2*)G(im.cblk) + (size_t)(gy * kGroupBlocks) * im.w8 + (size_t)gx * kGroupBlocks;
  const uint32_t w8 = (uint32_t)im.w8;
  uint32_t epos = 0;
  // current block
  uint32_t bx = 0, by = 0, lcx = 0, log2c = 0, covered = 1, size = 64, ctxs = 0, cell = 0;
  // current (block, channel)
  int ci = 3;
  uint32_t nzeros = 0, k = 0, prev = 0, histo = 0;
  bool want_nz = true;
#ifdef JXLHIP_PROFILE_HF
  const uint64_t pf_start = clock64();
  uint64_t pf_scalar = 0, pf_period = 0, pf_tokens = 0, pf_entries = 0, pf_periods = 0, pf_general = 0, pf_general_n = 0;
#endif
  // one-section wavefronts: do all clusters share one hybrid-integer configuration?  (then the token loop keeps it in a register)
  uint32_t cfg_uni = 0;
  bool cfg_is_uni = false;
  if (per_wave == 1 && !tab.slow) {
    cfg_uni = tab.cfg[0] & 0xFFF;
    cfg_is_uni = true;
    for (uint32_t i = 1; i < tab.dc->num_clusters; i++) cfg_is_uni &= (tab.cfg[i] & 0xFFF) == cfg_uni;
    cfg_uni = (uint32_t)__builtin_amdgcn_readfirstlane((int)cfg_uni);
    cfg_is_uni = __builtin_amdgcn_readfirstlane((int)cfg_is_uni) != 0;
  }
  while (!err) {
    if ((it & (kTop - 1)) == 0) {
#ifdef JXLHIP_PROFILE_HF
      const uint64_t pf_t0 = clock64();
      pf_periods++;
#endif
      b.TopUp();
      // descriptors the same way: a period starts at most kQ blocks, the queue holds 2 * kQ; what the previous period requested is
      // queued now, the next kQ are requested
      if (dpend_n) {
#pragma unroll
        for (int i = 0; i < kQ; i++) if ((uint32_t)i < dpend_n) dq[__umul24((dfilled + i) & dqmask, nslots)] = dpend[i];
        dfilled += dpend_n;
        dpend_n = 0;
      }
      if (dfilled < min(nblk, bi + kQ)) {   // start of the section, or a burst: wait for them
        const uint32_t lim = min(nblk, bi + kQ);
        uint32_t v[kQ];
#pragma unroll
        for (int i = 0; i < kQ; i++) { v[i] = 0u; if (dfilled + i < lim) v[i] = list[dfilled + i]; }
#pragma unroll
        for (int i = 0; i < kQ; i++) if (dfilled + i < lim) dq[__umul24((dfilled + i) & dqmask, nslots)] = v[i];
        dfilled = lim;
      }
      {
        const uint32_t lim = min(nblk, bi + 2 * kQ);
        if (dfilled < lim) {
          dpend_n = min(lim - dfilled, (uint32_t)kQ);
#pragma unroll
          for (int i = 0; i < kQ; i++) { dpend[i] = 0u; if ((uint32_t)i < dpend_n) dpend[i] = list[dfilled + i]; }
        }
      }
#ifdef JXLHIP_PROFILE_HF
      pf_period += clock64() - pf_t0;
#endif
    }
    // Fast path: while every lane of the wavefront that is still decoding sits inside a run of coefficient tokens (always the
    // case with one section per wavefront, i.e. small batches), stay in a loop that holds nothing but the coefficient token:
    // the general iteration below pays for both token kinds and their bookkeeping on every step.
    if (__all(!want_nz)) {
      if (per_wave == 1 && !tab.slow) {
        // One section per wavefront (single frames, small batches): every value of this loop is the same in all lanes, i.e. it is
        // scalar work - and on this machine a dependent scalar operation costs a fraction of a dependent vector one (which issues
        // every ~10 cycles).  The loop-carried state is moved to scalar registers (v_readfirstlane), table entries come back from LDS
        // through the same door, and the recurrence state -> alias entry -> state runs on the scalar unit; only the LDS addresses and
        // the entry store touch vector registers.
#define JXL_RFL(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#ifdef JXLHIP_PROFILE_HF
        const uint64_t pf_t1 = clock64();
        pf_entries++;
#endif
        // A lone wavefront issues one instruction every four cycles or so, and every LDS lookup on the chain adds ~80: the loop is
        // written for few instructions and few dependent lookups - the tokens a period or the block still allows are counted down
        // (no per-token stop conditions), the non-zero-count bucket is looked up again only after a non-zero coefficient, the range
        // check is one OR per coefficient, and a code whose clusters share one hybrid-integer configuration skips that lookup.
        uint32_t s_state = JXL_RFL(state), s_nz = JXL_RFL(nzeros), s_k = JXL_RFL(k), s_prev = JXL_RFL(prev), s_epos = JXL_RFL(epos), s_it = JXL_RFL(it);
        const uint32_t s_cov = JXL_RFL(covered), s_l2 = JXL_RFL(log2c), s_size = JXL_RFL(size), s_histo = JXL_RFL(histo);
        uint64_t s_buf = ((uint64_t)JXL_RFL((uint32_t)(b.buf >> 32)) << 32) | JXL_RFL((uint32_t)b.buf);
        int s_n = (int)JXL_RFL(b.n);
        uint32_t s_rd = JXL_RFL(b.rd), s_rng = 0;
        const uint32_t la = JXL_RFL(tab.log_alpha), le = 12 - la;
        const uint32_t left = kTop - (s_it & (kTop - 1)), room = s_size > s_k ? s_size - s_k : 0u;
        const uint32_t cnt0 = min(left, room);
        uint32_t cnt = cnt0;
        uint32_t s_a = s_histo + 2 * JXL_RFL(nnz_tab[(s_nz + s_cov - 1) >> s_l2]);
        uint32_t s_cidx = 0xFFFFFFFFu, s_cl = 0;
        if (cnt) do {
          const uint32_t ks = s_k >> s_l2;
          const uint32_t fctx = min(ks - 1, min(7 + (ks >> 1), 15 + (ks >> 2)));   // the three-piece position context, without branches
          // inside a run of zeros the context only moves when the position context does (every second / fourth position from 16 / 32
          // on): the cluster of the last lookup is kept while the context-map index stays the same
          const uint32_t cidx = s_a + fctx * 2 + s_prev;
          if (cidx != s_cidx) { s_cl = JXL_RFL(tab.cmap[cidx]); s_cidx = cidx; }
          const uint32_t cl = s_cl;
          const uint32_t res = s_state & 0xFFF, i = res >> le, pos = res & ((1u << le) - 1);
          const uint64_t e = tab.alias[(cl << la) | i];
          uint32_t c = cfg_uni;
          if (__builtin_expect(!cfg_is_uni, 0)) c = JXL_RFL(tab.cfg[cl]);
          const uint32_t x = JXL_RFL((uint32_t)e), y = JXL_RFL((uint32_t)(e >> 32));
          const bool gt = pos >= (x & 0xFF);
          const uint32_t sym = gt ? ((x >> 8) & 0xFF) : i;
          const uint32_t off = gt ? (y & 0xFFFF) + pos : pos;
          const uint32_t freq = gt ? ((x >> 16) ^ (y >> 16)) : (x >> 16);
          s_state = freq * (s_state >> 12) + off;
          if (s_state < 65536u) {
            if (s_n <= 32) { s_buf |= (uint64_t)JXL_RFL(b.ring[__umul24(s_rd & (kRing - 1), b.rs)]) << s_n; s_n += 32; s_rd++; }
            s_state = (s_state << 16) | ((uint32_t)s_buf & 0xFFFFu);
            s_buf >>= 16; s_n -= 16;
          }
          uint32_t u = sym;
          const uint32_t se = c & 0xF, split = 1u << se;
          if (__builtin_expect(sym >= split, 0)) {
            const uint32_t msb = (c >> 4) & 0xF, lsb = (c >> 8) & 0xF;
            const uint32_t nb = se - (msb + lsb) + ((sym - split) >> (msb + lsb)), nbr = nb > 32 ? 32 : nb;
            if (s_n <= 32) { s_buf |= (uint64_t)JXL_RFL(b.ring[__umul24(s_rd & (kRing - 1), b.rs)]) << s_n; s_n += 32; s_rd++; }
            const uint32_t bits = (uint32_t)(s_buf & (((uint64_t)1 << nbr) - 1));
            s_buf >>= nbr; s_n -= (int)nbr;
            const uint32_t low = sym & ((1u << lsb) - 1), hi = (1u << msb) | ((sym >> lsb) & ((1u << msb) - 1));
            u = (uint32_t)(((((uint64_t)hi << (nb & 63)) | bits) << lsb) | low);
          }
          cnt--;
          s_prev = 0;
          if (u) {
            const int32_t v = UnpackSigned(u);
            s_rng |= (uint32_t)(v + 0x8000);   // a value outside int16 leaves a bit above bit 15
            ent[s_epos++] = s_k | (uint32_t)v << 16;
            s_prev = 1;
            s_k++;
            if (--s_nz == 0) break;
            s_a = s_histo + 2 * JXL_RFL(nnz_tab[(s_nz + s_cov - 1) >> s_l2]);
          } else {
            s_k++;
          }
        } while (cnt);
        s_it += cnt0 - cnt;
#ifdef JXLHIP_PROFILE_HF
        pf_scalar += clock64() - pf_t1;
        pf_tokens += cnt0 - cnt;
#endif
        uint32_t s_err = s_rng > 0xFFFFu ? (uint32_t)kErrRange : 0u;
        if (s_nz != 0 && s_k >= s_size) s_err |= kErrBitstream;
#undef JXL_RFL
        state = s_state; nzeros = s_nz; k = s_k; prev = s_prev; epos = s_epos; it = s_it;
        b.buf = s_buf; b.n = s_n; b.rd = s_rd;
        err |= s_err;
        if (s_nz == 0 && !s_err) { want_nz = true; ci++; }
        continue;
      }
      for (;;) {
        const uint32_t nzl = (nzeros + covered - 1) >> log2c;
        const uint32_t ks = k >> log2c;
        const uint32_t fctx = ks < 16 ? ks - 1 : (ks < 32 ? 15 + ((ks - 16) >> 1) : 23 + ((ks - 32) >> 2));
        const uint32_t u = AnsGet(b, state, tab, histo + ((uint32_t)nnz_tab[nzl] + fctx) * 2 + prev);
        it++;
        bool leave = false;
        if (u) {
          const int32_t v = UnpackSigned(u);
          if (v != (int32_t)(int16_t)v) err |= kErrRange;
          ent[epos++] = k | (uint32_t)v << 16;
          prev = 1;
          if (--nzeros == 0) { want_nz = true; ci++; leave = true; }
        } else {
          prev = 0;
        }
        if (++k >= size && nzeros != 0) { err |= kErrBitstream; leave = true; }
        if (__any(leave) || (it & (kTop - 1)) == 0) break;
      }
      continue;
    }
#ifdef JXLHIP_PROFILE_HF
    const uint64_t pf_t2 = clock64();
    pf_general_n++;
#endif
    if (want_nz && ci >= 3) {
      if (bi >= nblk) break;
      const uint32_t d = dq[__umul24(bi & dqmask, nslots)];
      bi++;
      bx = d & 31; by = (d >> 5) & 31;
      lcx = (d >> 10) & 7;
      const uint32_t lcy = (d >> 13) & 7;
      ctxs = d >> 16;
      log2c = lcx + lcy; covered = 1u << log2c; size = covered << 6;
      cell = __umul24(by, w8) + bx;
      ci = 0;
    }
    const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
    uint32_t ctx;
    if (want_nz) {
      JXL_LDS uint8_t* const cc = col + __umul24(c * 32, nslots);
      uint32_t predicted;
      if (bx == 0) predicted = by == 0 ? 32u : (uint32_t)cc[0];
      else if (by == 0) predicted = cc[__umul24(bx - 1, nslots)];
      else predicted = ((uint32_t)cc[__umul24(bx, nslots)] + cc[__umul24(bx - 1, nslots)] + 1) >> 1;
      const uint32_t block_ctx = (ctxs >> (5 * ci)) & 31;
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
    it++;
    if (want_nz) {
      nzeros = u;
      if (nzeros + covered > size) { err |= kErrBitstream; break; }
      const uint8_t fill = (uint8_t)((nzeros + covered - 1) >> log2c);
      JXL_LDS uint8_t* const cc = col + __umul24(c * 32 + bx, nslots);
      for (uint32_t ix = 0; ix < (1u << lcx); ix++) cc[__umul24(ix, nslots)] = fill;
      U2 rec;
      rec.x = epos; rec.y = nzeros;
      cblk[(size_t)c * ncells + cell] = rec;
      if (nzeros) {
        const uint32_t block_ctx = (ctxs >> (5 * ci)) & 31;
        histo = ctx_offset + nbc * 37 + 458 * block_ctx;
        prev = nzeros > size / 16 ? 0 : 1;
        k = covered;
        want_nz = false;
      } else {
        ci++;
      }
    } else {
      if (u) {
        const int32_t v = UnpackSigned(u);
        if (v != (int32_t)(int16_t)v) err |= kErrRange;
        ent[epos++] = k | (uint32_t)v << 16;
        prev = 1;
        if (--nzeros == 0) { want_nz = true; ci++; }
      } else {
        prev = 0;
      }
      if (++k >= size && nzeros != 0) { err |= kErrBitstream; break; }
    }
#ifdef JXLHIP_PROFILE_HF
    pf_general += clock64() - pf_t2;
#endif
  }
  if (!err && (state != 0x130000u || b.slow_err)) err |= kErrBitstream;
  const uint64_t used = b.Consumed();
  if (!err && sec_bits + used > (im.sec_off[sec] + im.sec_size[sec]) * 8) err |= kErrBitstream;
  im.grp_bitpos[g] = err ? ~(uint64_t)0 : sec_bits + used;
#ifdef JXLHIP_PROFILE_HF
  if (per_wave == 1)
    printf("[hfprof] g %d total %llu scalar-loop %llu (%llu entries, %llu tokens) period %llu (%llu periods) blocks %u general %llu (%llu iterations)\n", g,
           (unsigned long long)(clock64() - pf_start), (unsigned long long)pf_scalar, (unsigned long long)pf_entries,
           (unsigned long long)pf_tokens, (unsigned long long)pf_period, (unsigned long long)pf_periods, nblk,
           (unsigned long long)pf_general, (unsigned long long)pf_general_n);
#endif
  if (err) SetError(im, err, 3, g);
}

// ------------------------------------------------------------------ alpha (Modular stream after the HF tokens), phase A
// One lane per pass-group section; a workgroup (one wavefront) holds sections of ONE image.
template <bool kLds, bool kGeneric = true>
__global__ __launch_bounds__(64) void alpha_ans_kernel(const DevImage* imgs, const SectionTask* tasks, int lane_stride, int scalar_rows) {
  JXL_SERIAL_PRIO();
  extern __shared__ __align__(16) uint8_t smem[];
  const SectionTask task = tasks[blockIdx.x];
  const DevImage& im = imgs[task.image];
  if (!im.has_alpha || im.alpha_in_global) return;
  ModTables<kLds> mt;
  const int slots = 64 / lane_stride;
  LoadModTables<kLds>(im, smem, (size_t)slots * kRingWords * 4, mt, threadIdx.x, 64, slots == 1 && scalar_rows);
  const int lane = threadIdx.x;
  if (lane >= slots || lane >= task.count) return;
  const int g = task.first + lane;
  ChanDesc* desc = im.alpha_desc + g;
  const uint64_t start = im.alpha_bitpos[g];
  if (start == ~(uint64_t)0) {   // the HF decoder already reported the failure
    ChanDesc d;
    d.kind = kChanFinal; d.value = 0; d.pad0 = 0; d.pad1 = 0;
    *desc = d;
    return;
  }
  const int gx = g % im.xg, gy = g / im.xg;
  const int sec = im.alpha_sec_base + g;
  LaneBits b;
  b.Init(im.cs, im.cs_size, start, (JXL_LDS uint32_t*)smem + lane, (uint32_t)slots);
  uint32_t err = 0;
  if (b.Read(4) != 3) err |= kErrUnsupportedHeader;
  else {
    const int x0 = gx * kGroupDim, y0 = gy * kGroupDim;
    const int gw = min(kGroupDim, im.w - x0), gh = min(kGroupDim, im.h - y0);
    if (im.lz_grp) b.SetLz(im.lz_grp + ((size_t)g << 16), 16, (uint32_t)gw);
    uint32_t state = InitAnsState(b, mt.tab);
    const int sid = 1 + 3 * im.nlf + kNumQuantTables + g;
    DecodeChannelLane<kLds, false, kGeneric>(b, state, mt.tab, mt.tree, 0, sid, gw, gh, im.alpha32 + (size_t)y0 * im.w + x0, im.w, desc,
                            im.wp_grp ? im.wp_grp + (size_t)g * im.wp_grp_ints : nullptr);
    if (state != 0x130000u || b.slow_err) err |= kErrBitstream;
    if (start + b.Consumed() > (im.sec_off[sec] + im.sec_size[sec]) * 8) err |= kErrBitstream;
  }
  if (err) {
    ChanDesc d;
    d.kind = kChanFinal; d.value = 0; d.pad0 = 0; d.pad1 = 0;
    *desc = d;
    SetError(im, err, 4, g);
  }
}

// Phase B: one wavefront per group: predictors (or plain conversion) -> 8-bit alpha plane.
// ---- alpha planes whose rows all use the clamped gradient (what every encoder writes): the prediction pass as a register pipeline ----
// A group's rows are a recurrence in both directions: a sample needs its West neighbour (same row) and North / North-West (row above).
// Sixteen lanes (one DPP row) own one group.  Lane i takes rows i, 16 + i, 32 + i ...; a row is worked through in steps of 16 columns
// (one 64-byte line of residuals, loaded whole and consumed from registers: no reliance on a cache keeping a line between uses, which
// is what sank the first version of this pass - 404 MB of traffic per plane instead of 41); row r starts at step r, so that the
// sixteen North samples of a step are exactly what the lane below (row r - 1) produced in the previous step: sixteen DPP row rotations.
// A group is at most 16 steps wide, so a lane is never idle between its rows.  Four groups per wavefront.  No LDS, no tile staging:
// ~13 vector instructions per sample instead of 62.  Groups this does not take (other predictors, widths that are not a multiple of 16,
// deeper samples) are left to alpha_finish_kernel, which skips the groups taken here (ChanDesc::pad0).
__device__ __forceinline__ int32_t RowRotateFromBelow(int32_t v) {   // value of lane i - 1 of the same row of 16 lanes (lane 0: lane 15)
  return __builtin_amdgcn_update_dpp(0, v, 0x121 /* row_ror:1 */, 0xF, 0xF, false);
}
// whether rows [row0, row0 + step * k) ... of the group's alpha channel all end in a gradient leaf; called by `nlanes` lanes (lane index
// `li`) that must then combine their answers
__device__ __forceinline__ bool AlphaRowsAreGradient(const I4* tree, int sid, int gh, int li, int nlanes) {
  bool ok = true;
  for (int r = li; r < gh; r += nlanes) { bool u = false; ok = ok && (RowNode(tree, 0, sid, r, &u).a & 0xFF) == 5; }
  return ok;
}
__device__ __forceinline__ bool AlphaGroupStatic(const DevImage& im, int g, int* gw, int* gh, int* sid, ChanDesc* d) {
  // (a Modular frame has no alpha_desc, no group grid of this kind: nothing of it may be touched)
  if (!im.has_alpha || im.is_modular || g < 0 || g >= im.ng || im.out_bits != 8 || im.alpha_bits != 8 || im.alpha_exp || (im.w & 3) != 0) return false;
  const int gx = g % im.xg, gy = g / im.xg;
  if (gy * kGroupDim >= im.band_y1 || (gy + 1) * kGroupDim <= im.band_y0) return false;   // outside the band: never decoded
  *gw = min(kGroupDim, im.w - gx * kGroupDim); *gh = min(kGroupDim, im.h - gy * kGroupDim);
  *sid = im.alpha_in_global ? 0 : 1 + 3 * im.nlf + kNumQuantTables + g;
  *d = im.alpha_desc[g];
  return d->kind == kChanResid && (*gw & 15) == 0;
}
__global__ __launch_bounds__(64) void alpha_finish_gradient_kernel(const DevImage* __restrict__ imgs) {
  const DevImage& im = imgs[blockIdx.y];
  const int lane = threadIdx.x, sub = lane >> 4, li = lane & 15;
  const int g = blockIdx.x * 4 + sub;
  int gw = 0, gh = 0, sid = 0;
  ChanDesc d;
  d.kind = kChanFinal;
  bool mine = AlphaGroupStatic(im, g, &gw, &gh, &sid, &d);
  if (mine) mine = AlphaRowsAreGradient((const I4*)im.tree, sid, gh, li, 16);
  {  // all sixteen lanes of the group must agree
    const uint64_t bal = __ballot(mine);
    mine = ((bal >> (sub * 16)) & 0xFFFFu) == 0xFFFFu;
  }
  if (__ballot(mine) == 0) return;
  const int gx = mine ? g % im.xg : 0, gy = mine ? g / im.xg : 0;
  const int x0 = gx * kGroupDim, y0 = gy * kGroupDim;
  const int nq = mine ? gw >> 4 : 0, rows = mine ? gh : 0;
  const JXL_GLB int32_t* const plane = G(im.alpha32) + (size_t)y0 * im.w + x0;
  JXL_GLB uint8_t* const out = G(im.alpha) + (size_t)y0 * im.w + x0;
  // uniform loop bound: the longest of the wavefront's groups
  int steps = rows + 15;
  steps = max(steps, __shfl_xor(steps, 16)); steps = max(steps, __shfl_xor(steps, 32));
  typedef int32_t __attribute__((ext_vector_type(4))) I4v;
  auto line_of = [&](int T, int* r_out, int* q_out) {   // the (row, 16-column step) this lane works on at time T; false: idle
    const int m = T - li;
    const int r = (m >> 4) * 16 + li, q = m & 15;
    *r_out = r; *q_out = q;
    return m >= 0 && r < rows && q < nq;
  };
  auto load_line = [&](int T, I4v* v) {   // the lane's residual line of time T (any in-range line when it is idle then)
    int r, q;
    const bool act = line_of(T, &r, &q);
    const int rc = act ? r : 0, qc = act ? q : 0;
    const JXL_GLB I4v* p = (const JXL_GLB I4v*)(plane + (size_t)(mine ? rc : 0) * im.w + qc * 16);
    v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3];
  };
  if (mine && li == 0) im.alpha_desc[g].pad0 = 1;   // tells alpha_finish_kernel (the next launch) to leave the group alone
  I4v cur[4], nxt[4];
  load_line(0, nxt);
  int32_t prev_out[16];
#pragma unroll
  for (int j = 0; j < 16; j++) prev_out[j] = 0;
  int32_t W = 0, n_last = 0;
  for (int T = 0; T < steps; T++) {
#pragma unroll
    for (int k = 0; k < 4; k++) cur[k] = nxt[k];
    load_line(T + 1, nxt);   // in flight during this step
    int r, q;
    const bool act = line_of(T, &r, &q);
    int32_t nq16[16];
#pragma unroll
    for (int j = 0; j < 16; j++) nq16[j] = RowRotateFromBelow(prev_out[j]);   // row r - 1, the same 16 columns
    const bool top = r == 0;
    int32_t o[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int32_t res = cur[j >> 2][j & 3];
      int32_t n, nw;
      if (top) { n = W; nw = W; }                       // top row: North = North-West = West
      else { n = nq16[j]; nw = j == 0 ? n_last : nq16[j - 1]; }
      if (j == 0 && q == 0) {                            // first column: West = North = North-West = the sample above (0 in the top row)
        const int32_t w0 = top ? 0 : nq16[0];
        W = w0; n = w0; nw = w0;
      }
      const int32_t val = (int32_t)((uint32_t)res + (uint32_t)ClampedGradient32(W, n, nw));
      o[j] = val;
      W = val;
    }
    n_last = nq16[15];
    if (act) {
#pragma unroll
      for (int j = 0; j < 16; j++) prev_out[j] = o[j];
      uint32_t pk[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t wd = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) { const int32_t v = o[4 * k + b]; wd |= (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)) << (8 * b); }
        pk[k] = wd;
      }
      typedef uint32_t __attribute__((ext_vector_type(4))) U4v;
      *(JXL_GLB U4v*)(out + (size_t)r * im.w + q * 16) = U4v{pk[0], pk[1], pk[2], pk[3]};
    }
  }
}

__global__ __launch_bounds__(64, 3) void alpha_finish_kernel(const DevImage* __restrict__ imgs) {
  __shared__ int32_t s_carry[256];
  __shared__ int32_t s_tile[64 * 65];
  const DevImage& im = imgs[blockIdx.y];
  const int g = blockIdx.x;
  if (!im.has_alpha || g >= im.ng || im.is_modular) return;
  const int lane = threadIdx.x;
  const int gx = g % im.xg, gy = g / im.xg;
  if (gy * kGroupDim >= im.band_y1 || (gy + 1) * kGroupDim <= im.band_y0) return;   // outside the band: never decoded
  const ChanDesc d = im.alpha_desc[g];
  const int x0 = gx * kGroupDim, y0 = gy * kGroupDim;
  const int gw = min(kGroupDim, im.w - x0), gh = min(kGroupDim, im.h - y0);
  int32_t* plane = im.alpha32 + (size_t)y0 * im.w + x0;
  const int sid = im.alpha_in_global ? 0 : 1 + 3 * im.nlf + kNumQuantTables + g;
  if (d.pad0 == 1) return;   // finished by alpha_finish_gradient_kernel (phase A writes pad0 = 0)
  if (im.out_bits == 8 && im.alpha_bits == 8 && !im.alpha_exp) {   // the common case: clamp to u8, packed stores
    uint8_t* out = im.alpha + (size_t)y0 * im.w + x0;
    if (d.kind == kChanResid) {
      PredictWaveTiled<true>((const I4*)im.tree, 0, sid, plane, im.w, gw, gh, d.kind, d.value, out, im.w, (JXL_LDS int32_t*)s_carry,
                             (JXL_LDS int32_t*)s_tile, lane);
    } else {
      // constant (e.g. fully opaque) or already final samples: row by row, no division per sample
      const bool cst = d.kind == kChanConst;
      const uint8_t cv = (uint8_t)(d.value < 0 ? 0 : (d.value > 255 ? 255 : d.value));
      for (int y = 0; y < gh; y++) {
        const size_t ro = (size_t)y * im.w;
        for (int x = lane; x < gw; x += 64) {
          const int v = cst ? 0 : plane[ro + x];
          out[ro + x] = cst ? cv : (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
      }
    }
    return;
  }
  // any other depth: finish the integers in place, then scale them to the output sample type
  if (d.kind == kChanResid)
    PredictWaveTiled<false>((const I4*)im.tree, 0, sid, plane, im.w, gw, gh, d.kind, d.value, nullptr, 0, (JXL_LDS int32_t*)s_carry,
                            (JXL_LDS int32_t*)s_tile, lane);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the wavefront reads back what its lanes stored
  __syncthreads();
  const bool cst = d.kind == kChanConst;
  for (int i = lane; i < gw * gh; i += 64) {
    const size_t o = (size_t)(i / gw) * im.w + (i % gw);
    const uint32_t v = SampleToOutBits(cst ? d.value : plane[o], im.alpha_bits, im.alpha_exp, im.out_bits, im.out_float);
    StoreOutSample(im.alpha, (size_t)y0 * im.w + x0 + o, v, im.out_bits);
  }
}

// ------------------------------------------------------------------ Modular (lossless) frames
// Sections of a Modular frame: 0 = GlobalModular stream (channels before mod_first_group), 1 + g = LF group g (channels
// squeezed by >= 3 in both directions), 1 + nlf + g = pass group g (the rest).  A section codes, for each of its channels, the
// rectangle of that channel covered by the group; the channel index property counts the channels the section holds.
struct ModRect { int x0, y0, w, h; };
__device__ __forceinline__ bool ModSectionRect(const DevImage& im, int kind, int g, const ModChanDev& ch, int c, ModRect* r) {
  if (kind == 0) {
    if (c >= im.mod_first_group) return false;
    r->x0 = 0; r->y0 = 0; r->w = ch.w; r->h = ch.h;
    return ch.w > 0 && ch.h > 0;
  }
  if (c < im.mod_first_group) return false;
  const int shift = min(ch.hshift, ch.vshift);
  if (kind == 1 ? shift < 3 : shift > 2) return false;
  const int dim = kind == 1 ? im.group_dim * 8 : im.group_dim;
  const int gx = kind == 1 ? g % im.xlf : g % im.xg, gy = kind == 1 ? g / im.xlf : g / im.xg;
  const int x0 = (gx * dim) >> ch.hshift, y0 = (gy * dim) >> ch.vshift;
  const int w = max(0, min(dim >> ch.hshift, ch.w - x0)), h = max(0, min(dim >> ch.vshift, ch.h - y0));
  r->x0 = x0; r->y0 = y0; r->w = w; r->h = h;
  return w > 0 && h > 0;
}
__device__ __forceinline__ void ModSectionOf(const DevImage& im, int s, int* kind, int* g, int* sid, int* sec) {
  if (s == 0) { *kind = 0; *g = 0; *sid = 0; *sec = 0; }
  else if (s <= im.nlf) { *kind = 1; *g = s - 1; *sid = 1 + im.nlf + *g; *sec = 1 + *g; }
  else { *kind = 2; *g = s - 1 - im.nlf; *sid = 1 + 3 * im.nlf + kNumQuantTables + *g; *sec = 2 + im.nlf + *g; }
}

// Phase A: one lane per section.  A frame with a single TOC entry is one bit stream: its lane walks global, LF group and pass
// group one after the other.
// kUni (launches with one section per wavefront, tables in LDS): all 64 lanes stay alive and compute the same (uniform) values - the
// row-static loops already keep their chains on the scalar unit - so that per-sample channels can use the lanes (modular_uniform.h).
// LDS of that shape: bit windows | 3 rows | weighted-predictor rows | grid | tree + code.
template <bool kLds, bool kUni = false>
__global__ __launch_bounds__(64) void modular_ans_kernel(const DevImage* imgs, const SectionTask* tasks, int lanes, int rb_width, int wp_lds, int scalar_rows) {
  JXL_SERIAL_PRIO();
  extern __shared__ __align__(16) uint8_t smem[];
  const SectionTask task = tasks[blockIdx.x];
  const DevImage& im = imgs[task.image];
  // LDS: bit windows | previous-row buffers (rb_width ints per lane, interleaved) | weighted-predictor state | tree + code
  const size_t off_rows = (size_t)64 * kRingWords * 4;
  const size_t off_wp = off_rows + (size_t)(kUni ? 3 : lanes) * rb_width * 4;
  const size_t wp_ints = (size_t)10 * (rb_width + 2);
  const size_t off_grid = off_wp + (wp_lds ? (size_t)lanes * wp_ints * 4 : 0);
  const size_t off_tab = off_grid + (kUni ? (size_t)kUniGridCells * 16 : 0);
  ModTables<kLds> mt;
  LoadModTables<kLds>(im, smem, off_tab, mt, threadIdx.x, 64, lanes == 1 && scalar_rows);
  const int lane = kUni ? 0 : (int)threadIdx.x;   // kUni: every lane plays lane 0 (same slot, same section)
  if (lane >= task.count || lane >= lanes) return;
  RowBuf<kLds> rows;
  rows.rb = (typename AS<kLds>::Row)((JXL_LDS int32_t*)((JXL_LDS uint8_t*)smem + off_rows) + lane);
  rows.rb_stride = lanes;
  rows.rb_width = rb_width;
  int32_t* const wp_local = wp_lds ? (int32_t*)(smem + off_wp) + (size_t)lane * wp_ints : nullptr;
  UniAreas uni;
  if constexpr (kUni) {
    uni.rows = (JXL_LDS int32_t*)((JXL_LDS uint8_t*)smem + off_rows);
    uni.wp = wp_lds ? (JXL_LDS int32_t*)((JXL_LDS uint8_t*)smem + off_wp) : nullptr;
    uni.grid = (JXL_LDS uint32_t*)((JXL_LDS uint8_t*)smem + off_grid);
    uni.rw = rb_width;
    uni.use_wp = im.wp_grp != nullptr && wp_lds;
  }
  const int s0 = task.first + lane;
  const int nsub = im.single ? 3 : 1;   // single: this lane continues through sections 0, 1, 2
  LaneBits b;
  b.rs = 0;   // not initialised yet
  uint32_t err = 0;
  int32_t* const wps = im.wp_grp ? im.wp_grp + (size_t)s0 * im.wp_grp_ints : nullptr;
  for (int k = 0; k < nsub && !err; k++) {
    const int s = s0 + k;
    int kind, g, sid, sec;
    ModSectionOf(im, s, &kind, &g, &sid, &sec);
    if (im.single) sec = 0;
    ChanDesc* desc = im.mod_desc + (size_t)s * im.mod_ncoded;
    // does the section hold any channel?  (an empty one has no bits at all, not even a header)
    bool any = false;
    for (int c = 0; c < im.mod_ncoded && !any; c++) { ModRect r; any = ModSectionRect(im, kind, g, im.mod_chan[c], c, &r); }
    if (!any) continue;
    uint64_t start = 0;
    if (kind == 0) { start = im.mod_data_bits; b.Init(im.cs, im.cs_size, start, (JXL_LDS uint32_t*)smem + lane, 64); }
    else if (!im.single) {
      if (im.sec_size[sec] == 0) { err |= kErrBitstream; break; }
      start = im.sec_off[sec] * 8;
      b.Init(im.cs, im.cs_size, start, (JXL_LDS uint32_t*)smem + lane, 64);
    } else if (k > 0 && b.rs == 0) {   // single stream but the global part was empty: start where LfGlobal's header ended
      b.Init(im.cs, im.cs_size, im.mod_data_bits, (JXL_LDS uint32_t*)smem + lane, 64);
    }
    if (kind != 0 && b.Read(4) != 3) { err |= kErrUnsupportedHeader; break; }
    if (im.lz_mod) {   // LZ77: distance multiplier = the widest channel of this section
      uint32_t dm = 0;
      for (int c = 0; c < im.mod_ncoded; c++) { ModRect r; if (ModSectionRect(im, kind, g, im.mod_chan[c], c, &r)) dm = max(dm, (uint32_t)r.w); }
      b.SetLz(im.lz_mod + ((size_t)s << 20), 20, dm);
      b.lz_copy = 0; b.lz_done = 0;
    }
    uint32_t state = InitAnsState(b, mt.tab);
    int sub = 0;
#pragma unroll 1
    for (int c = 0; c < im.mod_ncoded; c++) {
      const ModChanDev ch = im.mod_chan[c];
      ModRect r;
      if (!ModSectionRect(im, kind, g, ch, c, &r)) continue;
      DecodeChannelLane<kLds, kUni>(b, state, mt.tab, mt.tree, sub, sid, r.w, r.h, ch.plane + (size_t)r.y0 * ch.w + r.x0, ch.w, desc + c,
                                    (wps && wp_local && r.w <= rb_width) ? wp_local : wps, rb_width ? &rows : nullptr, kUni ? &uni : nullptr);
      sub++;
    }
    if (state != 0x130000u || b.slow_err) err |= kErrBitstream;
    if (!im.single && start + b.Consumed() > (im.sec_off[sec] + im.sec_size[sec]) * 8) err |= kErrBitstream;
  }
  if (err) {
    ChanDesc d;
    d.kind = kChanFinal; d.value = 0; d.pad0 = 0; d.pad1 = 0;
    for (int k = 0; k < nsub; k++)
      for (int c = 0; c < im.mod_ncoded; c++) im.mod_desc[(size_t)(s0 + k) * im.mod_ncoded + c] = d;
    SetError(im, err);
  }
}

// Phase B: one wavefront per (section, coded channel)
__global__ __launch_bounds__(64) void modular_finish_kernel(const DevImage* __restrict__ imgs, int max_coded) {
  __shared__ int32_t s_carry[kCarryInts];
  __shared__ int32_t s_tile[64 * 65];
  const DevImage& im = imgs[blockIdx.y];
  if (!im.is_modular) return;
  const int s = blockIdx.x / max_coded, c = blockIdx.x % max_coded;
  if (s >= 1 + im.nlf + im.ng || c >= im.mod_ncoded) return;
  int kind, g, sid, sec;
  ModSectionOf(im, s, &kind, &g, &sid, &sec);
  const ModChanDev ch = im.mod_chan[c];
  ModRect r;
  if (!ModSectionRect(im, kind, g, ch, c, &r)) return;
  // the channel-index property of the section: how many of its channels precede this one
  int sub = 0;
  for (int k = 0; k < c; k++) { ModRect rk; sub += ModSectionRect(im, kind, g, im.mod_chan[k], k, &rk) ? 1 : 0; }
  FinishChannelI32(im.mod_desc[(size_t)s * im.mod_ncoded + c], (const I4*)im.tree, sub, sid, ch.plane + (size_t)r.y0 * ch.w + r.x0, ch.w, r.w, r.h,
                   (JXL_LDS int32_t*)s_carry, threadIdx.x, (JXL_LDS int32_t*)s_tile);
}

// Inverse Squeeze steps (between phase B and the output): the average / residual pair of one step -> the unsqueezed channel.
__device__ __forceinline__ int64_t SmoothTendency(int64_t B, int64_t a, int64_t n) {
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
// horizontal: one thread per row (the recurrence runs along x)
__global__ void unsqueeze_h_kernel(const int32_t* avg, const int32_t* res, int32_t* out, int aw, int ah, int rw) {
  const int y = blockIdx.x * blockDim.x + threadIdx.x;
  if (y >= ah) return;
  const int32_t* pa = avg + (size_t)y * aw;
  const int32_t* pr = res + (size_t)y * rw;
  int32_t* po = out + (size_t)y * (aw + rw);
  int64_t left = 0;
  for (int x = 0; x < rw; x++) {
    const int64_t a = pa[x], next = x + 1 < aw ? pa[x + 1] : a;
    if (x == 0) left = a;
    const int64_t diff = (int64_t)pr[x] + SmoothTendency(left, a, next);
    const int64_t A = ((a * 2) + diff + (diff > 0 ? -(diff & 1) : (diff & 1))) >> 1;
    po[2 * x] = (int32_t)A;
    po[2 * x + 1] = (int32_t)(A - diff);
    left = A - diff;
  }
  if (aw > rw) po[2 * rw] = pa[rw];
}
// vertical: one thread per column (the recurrence runs along y; coalesced across threads)
__global__ void unsqueeze_v_kernel(const int32_t* avg, const int32_t* res, int32_t* out, int aw, int ah, int rh) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= aw) return;
  int64_t top = 0;
  for (int y = 0; y < rh; y++) {
    const int64_t a = avg[(size_t)y * aw + x], next = y + 1 < ah ? avg[(size_t)(y + 1) * aw + x] : a;
    if (y == 0) top = a;
    const int64_t diff = (int64_t)res[(size_t)y * aw + x] + SmoothTendency(top, a, next);
    const int64_t A = ((a * 2) + diff + (diff > 0 ? -(diff & 1) : (diff & 1))) >> 1;
    out[(size_t)(2 * y) * aw + x] = (int32_t)A;
    out[(size_t)(2 * y + 1) * aw + x] = (int32_t)(A - diff);
    top = A - diff;
  }
  if (ah > rh) out[(size_t)(2 * rh) * aw + x] = avg[(size_t)rh * aw + x];
}
// inverse reversible colour transform on three planes of n samples, in place
__global__ void rct_inverse_kernel(int32_t* p0, int32_t* p1, int32_t* p2, size_t n, int type) {
  const int perm = type / 7, custom = type % 7;
  int32_t* dst[3] = {p0, p1, p2};
  int32_t* da = dst[perm % 3];
  int32_t* db = dst[(perm + 1 + perm / 3) % 3];
  int32_t* dc = dst[(perm + 2 - perm / 3) % 3];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int32_t a = p0[i], b = p1[i], c = p2[i];
    if (custom == 6) {
      const int32_t tmp = a - (c >> 1), G = c + tmp, B = tmp - (b >> 1), R = B + b;
      a = R; b = G; c = B;
    } else {
      if (custom & 1) c += a;
      if ((custom >> 1) == 1) b += a;
      else if ((custom >> 1) == 2) b += (a + c) >> 1;
    }
    da[i] = a; db[i] = b; dc[i] = c;
  }
}

// Inverse reversible colour transforms (last first), clamp, interleave.  CMYK streams (black extra channel): C, M, Y, K leave as
// 255 - stored sample (Decoder/JxlDecoder.cpp:159-215: the stream stores 0 = full ink, the host wants 0 = no ink), alpha as it is.
__global__ void modular_out_kernel(const DevImage* __restrict__ imgs) {
  const DevImage& im = imgs[blockIdx.y];
  if (!im.is_modular) return;
  const size_t n = (size_t)im.w * im.h;
  const int nch = im.mod_nch;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int32_t v[5] = {0, 0, 0, 0, 0};
    for (int c = 0; c < nch; c++) v[c] = im.mod_plane[c][i];
    for (int t = im.mod_ntr - 1; t >= 0; t--) {
      const int bc = im.mod_tr[t][0], type = im.mod_tr[t][1];
      const int perm = type / 7, custom = type % 7;
      int32_t a = v[bc], b = v[bc + 1], c = v[bc + 2];
      if (custom == 6) {
        const int32_t tmp = a - (c >> 1), G = c + tmp, B = tmp - (b >> 1), R = B + b;
        a = R; b = G; c = B;
      } else {
        if (custom & 1) c += a;
        if ((custom >> 1) == 1) b += a;
        else if ((custom >> 1) == 2) b += (a + c) >> 1;
      }
      int32_t o[3];
      o[perm % 3] = a;
      o[(perm + 1 + perm / 3) % 3] = b;
      o[(perm + 2 - perm / 3) % 3] = c;
      v[bc] = o[0]; v[bc + 1] = o[1]; v[bc + 2] = o[2];
    }
    // colour channels carry sample_bits / sample_exp, the alpha channel alpha_bits / alpha_exp, the black channel black_bits
    const int ncol = im.ncolor;
    float unmul = 1.0f;   // associated alpha: colour samples leave as float(sample) / max(float(alpha), 2^-26)
    if (im.unpremultiply) {
      int32_t av = 0;
      for (int c = ncol; c < nch; c++) if (im.mod_out_pos[c] == nch - 1) av = v[c];
      const float af = im.alpha_exp ? BitsToFloatSample(av, im.alpha_bits) : (float)av * im.alpha_unit;
      unmul = 1.0f / fmaxf(1.0f / 67108864.0f, af);
    }
    for (int c = 0; c < nch; c++) {
      const int pos = im.mod_out_pos[c];
      const bool is_alpha = im.has_alpha && pos == nch - 1;
      const bool is_black = im.cmyk && pos == 3;
      uint32_t s = SampleToOutBits(v[c], c < ncol ? im.sample_bits : (is_alpha ? im.alpha_bits : im.black_bits), c < ncol ? im.sample_exp : (is_alpha ? im.alpha_exp : 0),
                                   im.out_bits, im.out_float);
      if (im.unpremultiply && c < ncol) {
        const float f = im.sample_exp ? BitsToFloatSample(v[c], im.sample_bits) : (float)v[c] * (1.0f / (float)((1u << im.sample_bits) - 1));
        s = FloatToOutBits(f * unmul, im.out_bits, im.out_float);
      }
      if (im.cmyk && !is_alpha) { (void)is_black; s = 255u - s; }   // 8-bit CMYK only (checked on the host)
      StoreOutSample(im.out, i * nch + pos, s, im.out_bits);
    }
  }
}

// ------------------------------------------------------------------ launch wrappers
static void RaiseLds(const void* fn, size_t bytes) {
  if (bytes > 48 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

void LaunchLfAns(const DevImage* imgs, const SectionTask* tasks, int ntasks, int slots, size_t lds_bytes, int scalar_rows, bool lean, hipStream_t s) {
  if (ntasks <= 0) return;   // slots: sections per workgroup (lanes that decode); lds_bytes: their bit windows + the tables (0: tables stay global)
  if (lds_bytes && lean) {   // lean: no frame of the launch needs the per-sample path (see DecodeChannelLane)
    RaiseLds((const void*)lf_ans_kernel<true, false>, lds_bytes);
    hipLaunchKernelGGL((lf_ans_kernel<true, false>), dim3(ntasks), dim3(64), lds_bytes, s, imgs, tasks, slots, scalar_rows);
  } else if (lds_bytes) {
    RaiseLds((const void*)lf_ans_kernel<true>, lds_bytes);
    hipLaunchKernelGGL(lf_ans_kernel<true>, dim3(ntasks), dim3(64), lds_bytes, s, imgs, tasks, slots, scalar_rows);
  } else {
    hipLaunchKernelGGL(lf_ans_kernel<false>, dim3(ntasks), dim3(64), (size_t)slots * kRingWords * 4, s, imgs, tasks, slots, 0);
  }
}

void LaunchLfFinish(const DevImage* imgs, const SectionTask* tasks, int ntasks, hipStream_t s) {
  if (ntasks <= 0) return;
  hipLaunchKernelGGL(lf_finish_kernel, dim3(ntasks), dim3(256), 0, s, imgs, tasks);
}

void LaunchHfBlockList(const DevImage* imgs, int nimg, int max_groups, hipStream_t s) {
  if (nimg <= 0 || max_groups <= 0) return;
  hipLaunchKernelGGL(hf_blocklist_kernel, dim3(max_groups, nimg), dim3(64), 0, s, imgs);
}

size_t HfLaneLdsBytes(int ring_words) { return 96 + (size_t)ring_words * 4 + (size_t)(ring_words / 2) * 4; }

void LaunchHfDecode(const DevImage* imgs, const SectionTask* tasks, int nwg, int threads, int lane_stride, int ring_words,
                    size_t lds_bytes, size_t lane_bytes, hipStream_t s) {
  if (nwg <= 0) return;   // lds_bytes: tables + lanes of the largest workgroup; lane_bytes: the lanes alone (tables in global memory)
  (void)ring_words;   // one window size: 32 words (a 16-word variant paid off while the coefficient orders lived in LDS; not any more)
  if (lds_bytes) {
    RaiseLds((const void*)hf_decode_kernel<true, 32>, lds_bytes);
    hipLaunchKernelGGL((hf_decode_kernel<true, 32>), dim3(nwg), dim3(threads), lds_bytes, s, imgs, tasks, lane_stride);
  } else {
    RaiseLds((const void*)hf_decode_kernel<false, 32>, lane_bytes);
    hipLaunchKernelGGL((hf_decode_kernel<false, 32>), dim3(nwg), dim3(threads), lane_bytes, s, imgs, tasks, lane_stride);
  }
}

void LaunchAlphaAns(const DevImage* imgs, const SectionTask* tasks, int nwg, int lane_stride, size_t lds_bytes, int scalar_rows, bool lean, hipStream_t s) {
  if (nwg <= 0) return;
  if (lds_bytes && lean) {
    RaiseLds((const void*)alpha_ans_kernel<true, false>, lds_bytes);
    hipLaunchKernelGGL((alpha_ans_kernel<true, false>), dim3(nwg), dim3(64), lds_bytes, s, imgs, tasks, lane_stride, scalar_rows);
  } else if (lds_bytes) {
    RaiseLds((const void*)alpha_ans_kernel<true>, lds_bytes);
    hipLaunchKernelGGL(alpha_ans_kernel<true>, dim3(nwg), dim3(64), lds_bytes, s, imgs, tasks, lane_stride, scalar_rows);
  } else {
    hipLaunchKernelGGL(alpha_ans_kernel<false>, dim3(nwg), dim3(64), (size_t)(64 / lane_stride) * kRingWords * 4, s, imgs, tasks, lane_stride, 0);
  }
}

void LaunchModularAns(const DevImage* imgs, int nimg, const SectionTask* tasks, int ntasks, size_t lds_bytes, int max_sections, int max_coded,
                      int lanes, int rb_width, int wp_lds, int scalar_rows, hipStream_t s) {
  if (ntasks <= 0) return;
  if (lds_bytes && lanes == 1 && rb_width > 0) {   // one section per wavefront: the uniform shape (host sizes lds_bytes for its layout)
    RaiseLds((const void*)modular_ans_kernel<true, true>, lds_bytes);
    hipLaunchKernelGGL((modular_ans_kernel<true, true>), dim3(ntasks), dim3(64), lds_bytes, s, imgs, tasks, lanes, rb_width, wp_lds, scalar_rows);
  } else if (lds_bytes) {
    RaiseLds((const void*)modular_ans_kernel<true>, lds_bytes);
    hipLaunchKernelGGL(modular_ans_kernel<true>, dim3(ntasks), dim3(64), lds_bytes, s, imgs, tasks, lanes, rb_width, wp_lds, scalar_rows);
  } else {
    const size_t lds = (size_t)64 * kRingWords * 4 + (size_t)lanes * rb_width * 4 + (wp_lds ? (size_t)lanes * 10 * (rb_width + 2) * 4 : 0);
    RaiseLds((const void*)modular_ans_kernel<false>, lds);
    hipLaunchKernelGGL(modular_ans_kernel<false>, dim3(ntasks), dim3(64), lds, s, imgs, tasks, lanes, rb_width, wp_lds, 0);
  }
  hipLaunchKernelGGL(modular_finish_kernel, dim3(max_sections * max_coded, nimg), dim3(64), 0, s, imgs, max_coded);
}

void LaunchModularOp(int kind, int32_t* a, int32_t* b, int32_t* c, int aw, int ah, int rw, int rh, int type, hipStream_t s) {
  if (kind == 0) {
    const size_t n = (size_t)aw * ah;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (n) hipLaunchKernelGGL(rct_inverse_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, b, c, n, type);
  } else if (kind == 1) {
    if (ah > 0 && aw > 0) hipLaunchKernelGGL(unsqueeze_h_kernel, dim3((ah + 63) / 64), dim3(64), 0, s, a, b, c, aw, ah, rw);
  } else {
    if (ah > 0 && aw > 0) hipLaunchKernelGGL(unsqueeze_v_kernel, dim3((aw + 63) / 64), dim3(64), 0, s, a, b, c, aw, ah, rh);
  }
}

struct PaletteOut { int32_t* p[4]; };
__global__ void palette_inverse_kernel(const int32_t* palette, const int32_t* index, PaletteOut out, int nout, int nb_colors, size_t n, uint32_t* status) {
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int32_t idx = index[i];
    const bool ok = idx >= 0 && idx < nb_colors;
    bad |= !ok;
    for (int k = 0; k < nout; k++) out.p[k][i] = ok ? palette[(size_t)k * nb_colors + idx] : 0;
  }
  if (bad) atomicOr(status, (uint32_t)kErrUnsupportedHeader);
}
void LaunchModularPalette(const int32_t* palette, const int32_t* index, int32_t* const* out, int nout, int nb_colors, int w, int h, uint32_t* status,
                          hipStream_t s) {
  const size_t n = (size_t)w * h;
  if (!n || nout <= 0) return;
  PaletteOut o;
  for (int k = 0; k < 4; k++) o.p[k] = k < nout ? out[k] : nullptr;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(palette_inverse_kernel, dim3((unsigned)blocks), dim3(256), 0, s, palette, index, o, nout, nb_colors, n, status);
}

void LaunchModularOut(const DevImage* imgs, int nimg, size_t max_pixels, hipStream_t s) {
  size_t b = (max_pixels + 255) / 256;
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(modular_out_kernel, dim3((unsigned)b, nimg), dim3(256), 0, s, imgs);
}

void LaunchAlphaFinish(const DevImage* imgs, int nimg, int max_groups, hipStream_t s) {
  if (nimg <= 0 || max_groups <= 0) return;
  hipLaunchKernelGGL(alpha_finish_gradient_kernel, dim3((max_groups + 3) / 4, nimg), dim3(64), 0, s, imgs);
  hipLaunchKernelGGL(alpha_finish_kernel, dim3(max_groups, nimg), dim3(64), 0, s, imgs);
}

}  // namespace jxlhip
