// Per-sample ("generic") Modular channels for launches with ONE section per wavefront (a single lossless frame, small batches):
// the channel's MA tree looks at decoded neighbours, or its leaves use predictors beyond Zero / West / North / Gradient, so every
// token depends on the sample before it - a serial chain per section, and a 4K frame has only ~140 sections.  What the reference
// reaches through JxlDecoderProcessInput on `uses_original_profile` streams (Decoder/JxlDecoder.cpp:252; written at
// Encoder/JxlEncoder.cpp:214,325).  Included by entropy_kernels.hip (it uses that file's bit reader and table types).
//
// The previous version ran this chain on one lane of the vector unit with the other 63 switched off: a dependent vector
// operation every ~10 cycles, a dependent LDS read per tree level, the neighbours of the row above by global loads that queue
// behind the sample stores, five integer divisions per sample in the weighted predictor (measured: 4.5 us per sample).  Here the
// whole wavefront stays active and everything on the chain is uniform, so that the compiler keeps it on the scalar unit, while the
// 64 lanes do what is parallel:
//   * neighbours: the rows y-1 and y-2 live in LDS; per 64 samples one vector load per neighbour kind (N, NE, NN, the weighted
//     predictor's error rows), and the chain picks its operand with v_readlane - no memory latency on the chain;
//   * MA tree: the subtree of the channel is flattened once per channel into a grid over the thresholds of the (at most four)
//     sample-dependent properties it tests (at most 64 thresholds in all).  Every threshold sits in one lane together with the
//     coefficients of ITS property over the neighbourhood (each property of the format is a signed sum of W, N, NW, NE, NN, WW,
//     x, y, the previous W + N - NW and the weighted predictor's error, two of them with an absolute value): per sample the lanes
//     evaluate their property with a few multiply-adds on the uniform neighbourhood - no branch on the property id - one vector
//     compare against the thresholds gives all buckets at once (population counts of the slots' lane ranges), and the grid cell
//     holds the leaf (predictor, cluster, multiplier, offset): in a lane when the grid has at most 64 cells, else one LDS read -
//     instead of one dependent LDS read and a branch cascade per tree level.  Trees that need more walk the tree as before;
//   * weighted predictor: the error sums of the row above come from the per-batch vector loads, the in-row "+=" of the
//     format's error update is carried in registers (its memory form is only ever read back by the next two samples), the
//     divisions are lookups in a 64-entry table held one entry per lane;
//   * samples leave through v_writelane into a vector register that is stored (plane and LDS row) once per 64 samples.
// (no include guard / namespace of its own: textually part of entropy_kernels.hip's kernel namespace)

constexpr int kUniMaxProps = 4;

__device__ __forceinline__ int32_t UReadLane(int32_t v, uint32_t lane) { return __builtin_amdgcn_readlane(v, (int)lane); }
// v_writelane_b32 by hand (this compiler has the readlane builtin only).  On gfx9 a VALU instruction reads one SGPR over the constant
// bus, so the lane select goes through M0 - a reserved register the compiler itself only ever sets immediately before a use.
__device__ __forceinline__ int32_t UWriteLane(int32_t v, uint32_t lane, int32_t old) {
  const uint32_t sv = JXL_RFL(v), sl = JXL_RFL(lane);   // (no-ops for values the compiler already holds in scalar registers)
  asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(sv), "s"(sl));
  return old;
}
__device__ __forceinline__ int64_t UAbs64(int64_t v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int UFloorLog2_64(uint64_t v) { return 63 - __builtin_clzll(v); }   // v > 0

// Sum over the four lanes of a quad, left in all four (two DPP quad permutes).
__device__ __forceinline__ uint32_t UQuadSum(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E /* quad_perm:[2,3,0,1] */, 0xF, 0xF, false);
  return v;
}
// vdiv[idx] with a per-lane index (ds_bpermute: the LDS crossbar, no LDS memory)
__device__ __forceinline__ uint32_t ULaneTable(int32_t table, uint32_t idx) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(idx << 2), table); }

struct UniLeaf {   // one grid cell / one leaf
  uint32_t pred, cl, mul;
  int32_t off;
};

// Reads a tree node with a uniform index: one 16-byte LDS read, the four words moved to scalar registers.
__device__ __forceinline__ DevTreeNode UniNode(const JXL_LDS I4* tree, uint32_t idx) {
  const I4 v = tree[idx];
  DevTreeNode n;
  n.property = (int32_t)JXL_RFL(v.x); n.splitval = (int32_t)JXL_RFL(v.y); n.a = JXL_RFL(v.z); n.b = JXL_RFL(v.w);
  return n;
}

// The sample-dependent properties of the format (ids 2 .. 15) from the uniform neighbourhood (tree-walk fallback only).
struct UniHood {
  int32_t W, N, NW, NE, NN, WW;
  int64_t prev9, wp_err;
  int x, y;
};
__device__ __forceinline__ int64_t UniProperty(int id, const UniHood& h) {
  switch (id) {
    case 2: return h.y;
    case 3: return h.x;
    case 4: return h.N < 0 ? -(int64_t)h.N : h.N;
    case 5: return h.W < 0 ? -(int64_t)h.W : h.W;
    case 6: return h.N;
    case 7: return h.W;
    case 8: return (int64_t)h.W - h.prev9;
    case 9: return (int64_t)h.W + h.N - h.NW;
    case 10: return (int64_t)h.W - h.NW;
    case 11: return (int64_t)h.NW - h.N;
    case 12: return (int64_t)h.N - h.NE;
    case 13: return (int64_t)h.N - h.NN;
    case 14: return (int64_t)h.W - h.WW;
    case 15: return h.wp_err;
    default: return 0;
  }
}

template <class T> __device__ __forceinline__ T* UniPtr64(T* p) {   // a pointer every lane holds alike -> scalar registers
  const uint64_t v = (uint64_t)(uintptr_t)p;
  return (T*)(uintptr_t)(((uint64_t)JXL_RFL((uint32_t)(v >> 32)) << 32) | JXL_RFL((uint32_t)v));
}
template <class T> __device__ __forceinline__ JXL_LDS T* UniPtrLds(JXL_LDS T* p) { return (JXL_LDS T*)(uintptr_t)JXL_RFL((uint32_t)(uintptr_t)p); }

// Everything the row loops need about one channel.  Uniform members live in scalar registers, the `v*` members are per lane.
struct UniCtx {
  // geometry / stream
  int w, h, stride, rw, w2, chan, sid;
  JXL_GLB int32_t* out;
  JXL_LDS int32_t *rows, *werr, *wpe;
  JXL_LDS uint32_t* ring;
  uint32_t ring_rs, la, le;
  const JXL_LDS uint64_t* alias;
  const JXL_LDS uint8_t* cmap;
  const JXL_LDS uint32_t* cfg;
  const JXL_LDS I4* tree;
  uint32_t root;
  // tables held one entry per lane
  int32_t vdiv;        // 2^24 / (lane + 1)
  int32_t vcfg;        // hybrid-integer configuration of cluster `lane`
  bool cfg_in_lanes;
  // the grid
  bool grid_ok, leaf_in_lanes;
  // the thresholds of property slot k sit in FIXED lanes - two slots of 32 lanes (layout32) or four of 16 - so that the bucket of a
  // slot is a population count of a constant bit field of the compare mask (no per-slot masks to keep in registers)
  bool layout32;
  uint32_t stride1, stride2, stride3;   // grid strides of slots 1 .. 3 (slot 0: 1)
  int64_t vthr;                         // threshold of this lane (unused lanes lie outside every slot's range)
  int32_t vthr32;                       // the same as the 32-bit number the stream carries
  int32_t vcW, vcN, vcNW, vcNE, vcNN, vcWW, vcX, vcY, vcE;   // coefficients of this lane's property
  bool vabs, vp8;                       // ... its absolute-value flag; property 8 (subtracts the previous W + N - NW)
  bool use_far, use_xy, use_e, use_p8, use_abs;   // which input groups any lane needs
  int32_t vleaf0, vleaf1, vleaf2;       // grids of at most 64 cells: the leaf record of cell `lane` (packed word, multiplier, offset)
  const JXL_LDS U4* grid;
  // stream state
  uint32_t s_state, s_rd;
  uint64_t s_buf;
  int s_n;
};

template <bool kWp, int kPred>
__device__ __forceinline__ void UniRows(UniCtx& c, LaneBits& b) {
  const int lane = (int)(threadIdx.x & 63);
  const int w = c.w, h = c.h, rw = c.rw, w2 = c.w2;
  uint32_t s_state = c.s_state, s_rd = c.s_rd;
  uint64_t s_buf = c.s_buf;
  int s_n = c.s_n;
  JXL_LDS uint32_t* const ring = c.ring;
  const uint32_t ring_rs = c.ring_rs;
  auto word = [&]() {
    if (__builtin_expect(s_n <= 32, 0)) { s_buf |= (uint64_t)JXL_RFL(ring[__umul24(s_rd & (kRingWords - 1), ring_rs)]) << s_n; s_n += 32; s_rd++; }
  };
  bool big = false;   // a sample (or an error of the weighted predictor) beyond the bound of the 32-bit forms has been seen
  // Weighted predictor, 32-bit form: its four sub-predictors are four lanes (lane & 3; every quad computes the same).  Per lane: the
  // coefficients of ITS prediction over (W, N, NE) and over the true errors (West, N, NW, NE) - the format's four formulas as one -
  // its maximum weight, its two error rows in LDS.
  const int vi = lane & 3;
  const uint32_t vmaxw = vi == 0 ? 13u : 12u;
  const int32_t paW = (vi == 0 || vi == 2) ? 1 : 0, paN = vi == 0 ? -1 : (vi == 2 ? 0 : 1), paNE = vi == 0 ? 1 : 0;
  const int32_t pkW = vi == 1 ? 16 : (vi == 2 ? 10 : 0), pkN = vi == 1 ? 16 : (vi == 2 ? 10 : (vi == 3 ? 7 : 0)),
                pkNW = vi == 2 ? 10 : (vi == 3 ? 7 : 0), pkNE = vi == 1 ? 16 : (vi == 3 ? 7 : 0);
  JXL_LDS int32_t* const pe_lane = kWp ? c.wpe + (size_t)vi * 2 * w2 : nullptr;
  for (int y = 0; y < h; y++) {
    JXL_LDS int32_t* const cur = c.rows + (y % 3) * rw;
    JXL_LDS int32_t* const prv = c.rows + ((y + 2) % 3) * rw;
    JXL_LDS int32_t* const prv2 = c.rows + ((y + 1) % 3) * rw;
    const int cur_o = (y & 1) ? 0 : w2, prev_o = (y & 1) ? w2 : 0;   // the weighted predictor's row parity, as the format has it
    JXL_GLB int32_t* const row = c.out + (size_t)y * c.stride;
    int32_t W = 0, N = 0, NW = 0, WW = 0;
    int64_t prev9 = 0;
    // weighted predictor, carried along the row: wA_prev[i] = error sum slot "N" of the sample before (= slot "NW" of this one),
    // e_prev[i] = the error that sample adds to this one's slot "N"; teW / teN_prev likewise for the true errors
    uint32_t wA_prev[4] = {0, 0, 0, 0}, e_prev[4] = {0, 0, 0, 0};
    int32_t teW = 0, teN_prev = 0;
    // the same per lane (32-bit form): error of the row above at x and x + 1, slot "N" sum and error of the sample before
    int32_t ven = 0, vene = 0;
    uint32_t vA_prev = 0, ve_prev = 0;
    if constexpr (kWp) {
      ven = pe_lane[prev_o];
      vene = pe_lane[prev_o + 1];
    }
    for (int x0 = 0; x0 < w; x0 += 64) {
      const int cnt = min(64, w - x0);
      // ---- per-batch vector loads: the neighbours of 64 samples
      const int xl = x0 + lane;
      const int32_t vNE = (y && xl + 1 < w) ? prv[xl + 1] : 0;
      const int32_t vN0 = (y && xl < w) ? prv[xl] : 0;
      const int32_t vNN = (y > 1 && xl < w) ? prv2[xl] : 0;
      int32_t vE[4] = {0, 0, 0, 0}, vEne[4] = {0, 0, 0, 0}, vTe = 0, vTene = 0;
      if constexpr (kWp) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const JXL_LDS int32_t* p = c.wpe + (size_t)i * 2 * w2 + prev_o;
          vE[i] = xl < w ? p[xl] : 0;
          vEne[i] = xl + 1 < w ? p[xl + 1] : 0;
        }
        vTe = xl < w ? c.werr[prev_o + xl] : 0;
        vTene = xl + 1 < w ? c.werr[prev_o + xl + 1] : 0;
      }
      int32_t vout = 0, vTo = 0;
      if (x0 == 0) {   // start of the row: West, North and North-West are the sample above (0 in the top row)
        W = y ? UReadLane(vN0, 0) : 0;
        N = W; NW = W; WW = W;
        prev9 = 0;
      }
      for (int p0 = 0; p0 < cnt; p0 += kTopUpEvery) {
        b.rd = s_rd;
        b.TopUp();
        const int pe = min(cnt, p0 + kTopUpEvery);
        for (int li = p0; li < pe; li++) {
          const int x = x0 + li;
          const int32_t NE = (y && x + 1 < w) ? UReadLane(vNE, (uint32_t)li) : N;
          const int32_t NN = y > 1 ? UReadLane(vNN, (uint32_t)li) : N;
          // ---- weighted predictor (the format's arithmetic).  While every sample and error seen so far is small (`big` unset:
          // |sample| < 2^18, |error| < 2^22 - any image of up to 16 bits) each intermediate fits 32 bits except the last product;
          // the first larger value switches the channel to the 64-bit form for good.
          int64_t wp_pred = 0, wpred8 = 0;
          int32_t wp_err = 0;
          int64_t prediction[4] = {0, 0, 0, 0};
          int32_t vp = 0;      // (32-bit form) this lane's sub-prediction
          uint32_t vA = 0;     // ... and its slot "N" error sum
          uint32_t A[4] = {0, 0, 0, 0};   // (64-bit form) the same, scalar
          if constexpr (kWp) {
            uint32_t weights[4];
            const int32_t teN = UReadLane(vTe, (uint32_t)li);
            const int32_t teNE = x + 1 < w ? UReadLane(vTene, (uint32_t)li) : teN;
            const int32_t teNW = x > 0 ? teN_prev : teN;
            const int32_t tW = x > 0 ? teW : 0;
            if (__builtin_expect(!big, 1)) {
              // four lanes, one sub-predictor each
              vA = (uint32_t)ven + ve_prev;                                   // ve_prev is 0 at the start of a row
              const uint32_t ene = x + 1 < w ? (uint32_t)vene : vA;
              const uint32_t enw = x > 0 ? vA_prev : vA;
              const uint32_t s3 = vA + ene + enw;
              const int sh = max(26 - (int)__clz(s3 + 1), 0);
              uint32_t wgt = 4 + ((vmaxw * ULaneTable(c.vdiv, s3 >> sh)) >> sh);
              const uint32_t wsum = UQuadSum(wgt);
              wgt >>= (31 - (int)__clz(wsum)) - 4;
              const uint32_t wsum2 = UQuadSum(wgt);
              const int32_t N8 = N << 3, W8 = W << 3, NE8 = NE << 3;
              // (24-bit multiplies: full rate; under the 32-bit form's bound the samples * 8 stay below 2^21 and the errors below 2^22)
              vp = __mul24(paW, W8) + __mul24(paN, N8) + __mul24(paNE, NE8) -
                   ((__mul24(pkW, tW) + __mul24(pkN, teN) + __mul24(pkNW, teNW) + __mul24(pkNE, teNE)) >> 5);
              const int32_t acc = (int32_t)UQuadSum((uint32_t)(vp * (int32_t)wgt)) + (int32_t)(wsum2 >> 1) - 1;
              int32_t wq = (int32_t)JXL_RFL((uint32_t)(int32_t)(((int64_t)acc * (int64_t)ULaneTable(c.vdiv, wsum2 - 1)) >> 24));
              int32_t pm = tW;
              if (abs(teN) > abs(pm)) pm = teN;
              if (abs(teNW) > abs(pm)) pm = teNW;
              if (abs(teNE) > abs(pm)) pm = teNE;
              wp_err = pm;
              if (((teN ^ tW) | (teN ^ teNW)) <= 0) {
                const int32_t mx = max(W8, max(NE8, N8)), mn = min(W8, min(NE8, N8));
                wq = wq < mn ? mn : (wq > mx ? mx : wq);
              }
              wpred8 = wq;
              wp_pred = (wq + 3) >> 3;
            } else {
#pragma unroll
              for (int i = 0; i < 4; i++) {
                const uint32_t en = (uint32_t)UReadLane(vE[i], (uint32_t)li);
                A[i] = en + (x > 0 ? e_prev[i] : 0u);
                const uint32_t ene = x + 1 < w ? (uint32_t)UReadLane(vEne[i], (uint32_t)li) : A[i];
                const uint32_t enw = x > 0 ? wA_prev[i] : A[i];
                const uint64_t s3 = (uint64_t)A[i] + ene + enw;
                int shift = UFloorLog2_64(s3 + 1) - 5;
                if (shift < 0) shift = 0;
                const uint32_t maxw = i == 0 ? 13u : 12u;
                weights[i] = 4 + (uint32_t)(((uint64_t)maxw * (uint32_t)UReadLane(c.vdiv, (uint32_t)(s3 >> shift))) >> shift);
              }
              const int64_t N8 = (int64_t)N << 3, W8 = (int64_t)W << 3, NE8 = (int64_t)NE << 3;
              const int64_t sumWN = (int64_t)teN + tW;
              int32_t pm = tW;
              if (UAbs64(teN) > UAbs64(pm)) pm = teN;
              if (UAbs64(teNW) > UAbs64(pm)) pm = teNW;
              if (UAbs64(teNE) > UAbs64(pm)) pm = teNE;
              wp_err = pm;
              prediction[0] = W8 + NE8 - N8;
              prediction[1] = N8 - (((sumWN + teNE) * 16) >> 5);
              prediction[2] = W8 - (((sumWN + teNW) * 10) >> 5);
              prediction[3] = N8 - (((int64_t)teNW * 7 + (int64_t)teN * 7 + (int64_t)teNE * 7) >> 5);
              uint32_t wsum = weights[0] + weights[1] + weights[2] + weights[3];
              const int lw = 31 - __builtin_clz(wsum);
              wsum = 0;
#pragma unroll
              for (int i = 0; i < 4; i++) { weights[i] >>= lw - 4; wsum += weights[i]; }
              int64_t sum = (int64_t)(wsum >> 1) - 1;
#pragma unroll
              for (int i = 0; i < 4; i++) sum += prediction[i] * (int64_t)weights[i];
              wpred8 = (sum * (int64_t)(uint32_t)UReadLane(c.vdiv, wsum - 1)) >> 24;
              if (((teN ^ tW) | (teN ^ teNW)) <= 0) {
                const int64_t mx = W8 > NE8 ? (W8 > N8 ? W8 : N8) : (NE8 > N8 ? NE8 : N8), mn = W8 < NE8 ? (W8 < N8 ? W8 : N8) : (NE8 < N8 ? NE8 : N8);
                wpred8 = wpred8 < mn ? mn : (wpred8 > mx ? mx : wpred8);
              }
              wp_pred = (wpred8 + 3) >> 3;
            }
            teN_prev = teN;
          }
          // ---- leaf
          uint32_t l_pred, l_cl, l_mul, l_cfg;
          int32_t l_off;
          if (c.grid_ok) {
            // every threshold lane evaluates its property: a signed sum over the neighbourhood, coefficients per lane.  While the
            // samples are small (`big` unset) the sum fits 32 bits with room to spare and is a handful of full-rate multiply-adds
            // with no branch at all; thresholds are 32-bit numbers in the stream, so the compare is a 32-bit one too
            uint64_t gt;
            if (__builtin_expect(!big, 1)) {
              int32_t pv = c.vcW * W + c.vcN * N + c.vcNW * NW + c.vcNE * NE + c.vcNN * NN + c.vcWW * WW + c.vcX * x + c.vcY * y + c.vcE * wp_err -
                           (c.vp8 ? (int32_t)prev9 : 0);
              pv = c.vabs ? abs(pv) : pv;
              gt = __ballot(pv > c.vthr32);
            } else {
              int64_t pv = (int64_t)c.vcW * W + (int64_t)c.vcN * N + (int64_t)c.vcNW * NW;
              if (c.use_far) pv += (int64_t)c.vcNE * NE + (int64_t)c.vcNN * NN + (int64_t)c.vcWW * WW;
              if (c.use_xy) pv += (int64_t)c.vcX * x + (int64_t)c.vcY * y;
              if (c.use_e) pv += (int64_t)c.vcE * wp_err;
              if (c.use_p8) pv -= c.vp8 ? prev9 : 0;
              if (c.use_abs) pv = (c.vabs && pv < 0) ? -pv : pv;
              gt = __ballot(pv > c.vthr);
            }
            uint32_t cell;
            const uint32_t glo = (uint32_t)gt, ghi = (uint32_t)(gt >> 32);
            if (c.layout32) cell = (uint32_t)__builtin_popcount(glo) + (uint32_t)__builtin_popcount(ghi) * c.stride1;
            else cell = (uint32_t)__builtin_popcount(glo & 0xFFFFu) + (uint32_t)__builtin_popcount(glo >> 16) * c.stride1 +
                        (uint32_t)__builtin_popcount(ghi & 0xFFFFu) * c.stride2 + (uint32_t)__builtin_popcount(ghi >> 16) * c.stride3;
            // the leaf as one packed word: predictor | cluster << 4 | hybrid-integer configuration << 12 | (multiplier 1, offset 0) << 24
            uint32_t r0;
            if (c.leaf_in_lanes) r0 = (uint32_t)UReadLane(c.vleaf0, cell);
            else r0 = JXL_RFL(c.grid[cell].x);
            l_pred = r0 & 0xF; l_cl = (r0 >> 4) & 0xFF; l_cfg = (r0 >> 12) & 0xFFF;
            l_mul = 1; l_off = 0;
            if (__builtin_expect(!(r0 >> 24), 0)) {
              if (c.leaf_in_lanes) { l_mul = (uint32_t)UReadLane(c.vleaf1, cell); l_off = UReadLane(c.vleaf2, cell); }
              else { const U4 rec = c.grid[cell]; l_mul = JXL_RFL(rec.y); l_off = (int32_t)JXL_RFL(rec.z); }
            }
          } else {
            UniHood hd;
            hd.W = W; hd.N = N; hd.NW = NW; hd.NE = NE; hd.NN = NN; hd.WW = WW; hd.prev9 = prev9; hd.wp_err = wp_err; hd.x = x; hd.y = y;
            uint32_t node = c.root;
            DevTreeNode nd = UniNode(c.tree, node);
            while (nd.property >= 0) {
              const int64_t pv = nd.property == 0 ? (int64_t)c.chan : (nd.property == 1 ? (int64_t)c.sid : UniProperty(nd.property, hd));
              node = pv > nd.splitval ? nd.a : nd.b;
              nd = UniNode(c.tree, node);
            }
            l_pred = nd.a & 0xFF; l_cl = JXL_RFL(c.cmap[nd.a >> 8]); l_mul = nd.b; l_off = nd.splitval;
            l_cfg = (c.cfg_in_lanes ? (uint32_t)UReadLane(c.vcfg, l_cl) : JXL_RFL(c.cfg[l_cl])) & 0xFFF;
          }
          // ---- predictor (kPred >= 0: every leaf of the channel uses it)
          const uint32_t pred = kPred >= 0 ? (uint32_t)kPred : l_pred;
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
              const int32_t NEE = (y && x + 2 < w) ? (int32_t)JXL_RFL(prv[x + 2]) : NE;
              guess = (6 * (int64_t)N - 2 * (int64_t)NN + 7 * (int64_t)W + WW + NEE + 3 * (int64_t)NE + 8) / 16;
              break;
            }
            default: guess = 0; break;
          }
          // ---- token
          const uint32_t res = s_state & 0xFFF, ai = res >> c.le, pos = res & ((1u << c.le) - 1);
          const uint64_t ae = c.alias[(l_cl << c.la) | ai];
          const uint32_t ax = JXL_RFL((uint32_t)ae), ay = JXL_RFL((uint32_t)(ae >> 32));
          const uint32_t cf = l_cfg;
          const bool gtc = pos >= (ax & 0xFF);
          const uint32_t sym = gtc ? ((ax >> 8) & 0xFF) : ai;
          const uint32_t aoff = gtc ? (ay & 0xFFFF) + pos : pos;
          const uint32_t freq = gtc ? ((ax >> 16) ^ (ay >> 16)) : (ax >> 16);
          s_state = freq * (s_state >> 12) + aoff;
          if (__builtin_expect(s_state < 65536u, 0)) {
            word();
            s_state = (s_state << 16) | ((uint32_t)s_buf & 0xFFFFu);
            s_buf >>= 16; s_n -= 16;
          }
          uint32_t u = sym;
          {
            const uint32_t se = cf & 0xF, split = 1u << se;
            if (__builtin_expect(sym >= split, 0)) {
              const uint32_t msb = (cf >> 4) & 0xF, lsb = (cf >> 8) & 0xF;
              const uint32_t nb = se - (msb + lsb) + ((sym - split) >> (msb + lsb)), nbr = nb > 32 ? 32 : nb;
              word();
              const uint32_t bits = (uint32_t)(s_buf & (((uint64_t)1 << nbr) - 1));
              s_buf >>= nbr; s_n -= (int)nbr;
              const uint32_t low = sym & ((1u << lsb) - 1), top = (1u << msb) | ((sym >> lsb) & ((1u << msb) - 1));
              u = (uint32_t)(((((uint64_t)top << (nb & 63)) | bits) << lsb) | low);
            }
          }
          const int32_t val = (int32_t)((int64_t)UnpackSigned(u) * (int64_t)l_mul + l_off + guess);
          vout = UWriteLane(val, (uint32_t)li, vout);
          if constexpr (kWp) {
            const int64_t v8l = (int64_t)val << 3;
            const int32_t terr = (int32_t)(wpred8 - v8l);
            // the bound the 32-bit form relies on, checked on every value the later samples will read back (this sample's own update
            // already takes the 64-bit form when it is the first large one)
            const bool big_new = big || (uint32_t)(val + (1 << 18)) >= (1u << 19) || (uint32_t)(terr + (1 << 22)) >= (1u << 23);
            if (__builtin_expect(!big_new, 1)) {
              const int32_t v8 = val << 3;
              const uint32_t ve = (uint32_t)(abs(vp - v8) + 3) >> 3;
              if (lane < 4) pe_lane[cur_o + x] = (int32_t)ve;
              vA_prev = vA; ve_prev = ve;
              ven = vene;
              vene = x + 2 < w ? pe_lane[prev_o + x + 2] : 0;     // for the sample after the next: off the chain
            } else {
              if (!big) {   // the switch: this sample was predicted on the lanes; from here on the state lives in scalar registers
#pragma unroll
                for (int i = 0; i < 4; i++) { prediction[i] = UReadLane(vp, (uint32_t)i); A[i] = (uint32_t)UReadLane((int32_t)vA, (uint32_t)i); }
              }
#pragma unroll
              for (int i = 0; i < 4; i++) {
                const uint32_t e = (uint32_t)((UAbs64(prediction[i] - v8l) + 3) >> 3);
                if (lane == i) (c.wpe + (size_t)i * 2 * w2 + cur_o)[x] = (int32_t)e;
                e_prev[i] = e;
                wA_prev[i] = A[i];
              }
            }
            big = big_new;
            vTo = UWriteLane(terr, (uint32_t)li, vTo);
            teW = terr;
          } else {
            big = big || (uint32_t)(val + (1 << 18)) >= (1u << 19);   // the bound of the 32-bit property sums
          }
          prev9 = (int64_t)W + N - NW;
          const int32_t oldW = W;
          W = val;
          WW = x >= 1 ? oldW : val;
          if (y) { NW = N; N = NE; } else { NW = val; N = val; }
        }
      }
      // ---- the batch leaves: plane (coalesced) and the LDS rows
      if (lane < cnt) {
        row[x0 + lane] = vout;
        cur[x0 + lane] = vout;
        if constexpr (kWp) c.werr[cur_o + x0 + lane] = vTo;   // (the sub-predictor errors were stored sample by sample)
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  c.s_state = s_state; c.s_rd = s_rd; c.s_buf = s_buf; c.s_n = s_n;
}

// Returns false when the channel has to take the old path (general symbol reader, rows wider than the LDS rows).
// lds_rows: 3 * rw ints; lds_wp: 10 * (rw + 2) ints (null: the tree does not use the weighted predictor); lds_grid: grid_cells
// records of 16 bytes.
__device__ __forceinline__ bool ModularChannelUniform(LaneBits& b, uint32_t& state_io, const CodeTab<true>& tab_in, const JXL_LDS I4* tree_in, int chan_in, int sid_in,
                                                   int w_in, int h_in, int32_t* out_generic, int stride_in, JXL_LDS int32_t* lds_rows_in, int rw_in,
                                                   JXL_LDS int32_t* lds_wp_in, JXL_LDS uint32_t* lds_grid_in, int grid_cells, bool use_wp_in) {
  // every argument is the same in all lanes, but it reached them through vector loads: say so once, here, so that everything derived
  // from them (loop bounds, addresses, the whole chain) is scalar to the compiler
  UniCtx c;
  c.w = (int)JXL_RFL(w_in); c.h = (int)JXL_RFL(h_in); c.chan = (int)JXL_RFL(chan_in); c.sid = (int)JXL_RFL(sid_in); c.stride = (int)JXL_RFL(stride_in);
  c.rw = (int)JXL_RFL(rw_in); c.w2 = c.rw + 2;
  const bool use_wp = JXL_RFL(use_wp_in ? 1u : 0u) != 0;
  c.rows = UniPtrLds(lds_rows_in);
  JXL_LDS int32_t* const lds_wp = UniPtrLds(lds_wp_in);
  JXL_LDS uint32_t* const lds_grid = UniPtrLds(lds_grid_in);
  c.tree = UniPtrLds((JXL_LDS I4*)tree_in);
  c.cmap = UniPtrLds((JXL_LDS uint8_t*)tab_in.cmap);
  c.cfg = UniPtrLds((JXL_LDS uint32_t*)tab_in.cfg);
  c.alias = UniPtrLds((JXL_LDS uint64_t*)tab_in.alias);
  const DevCode* const dc = UniPtr64(tab_in.dc);
  c.ring = UniPtrLds(b.ring);
  c.ring_rs = JXL_RFL(b.rs);
  const int w = c.w, h = c.h, chan = c.chan, sid = c.sid;
  if (JXL_RFL(tab_in.slow) || w > c.rw || w <= 0 || h <= 0 || (use_wp && !lds_wp)) return false;
  const int lane = (int)(threadIdx.x & 63);
  c.out = G(UniPtr64(out_generic));
  c.la = JXL_RFL(tab_in.log_alpha); c.le = 12 - c.la;
  c.werr = lds_wp;
  c.wpe = lds_wp ? lds_wp + 2 * c.w2 : nullptr;
  c.grid = (const JXL_LDS U4*)lds_grid;
  const JXL_LDS I4* const tree = c.tree;
  // ---- the channel's subtree: resolve the static properties at the top, then collect the thresholds of the others
  uint32_t root = 0;
  for (;;) {
    const DevTreeNode nd = UniNode(tree, root);
    if (nd.property != 0 && nd.property != 1) break;
    const int v = nd.property == 0 ? chan : sid;
    root = v > nd.splitval ? nd.a : nd.b;
  }
  c.root = root;
  int nprops = 0;
  int prop_id[kUniMaxProps] = {0, 0, 0, 0};
  int nthr[kUniMaxProps] = {0, 0, 0, 0};
  int64_t thr[kUniMaxProps];   // per lane: the sorted thresholds of property slot k (lanes >= nthr[k]: +inf)
#pragma unroll
  for (int k = 0; k < kUniMaxProps; k++) thr[k] = INT64_MAX;
  bool grid_ok = true;
  {
    // depth-first walk of the subtree with a stack in the grid area (the grid is built afterwards); static properties follow their
    // one live child
    JXL_LDS uint32_t* const stack = lds_grid;
    const uint32_t cap = (uint32_t)grid_cells * 4;
    uint32_t sp = 0, node = root;
    uint32_t visited = 0;
    for (;;) {
      const DevTreeNode nd = UniNode(tree, node);
      if (++visited > (1u << 20)) { grid_ok = false; break; }   // a malformed (cyclic) tree cannot hang the wavefront
      bool pop = false;
      if (nd.property < 0) pop = true;
      else if (nd.property == 0 || nd.property == 1) {
        const int v = nd.property == 0 ? chan : sid;
        node = v > nd.splitval ? nd.a : nd.b;
      } else {
        if (nd.property > 15) { grid_ok = false; break; }
        int slot = -1;
#pragma unroll
        for (int k = 0; k < kUniMaxProps; k++) if (k < nprops && prop_id[k] == nd.property) slot = k;
        if (slot < 0) {
          if (nprops == kUniMaxProps) { grid_ok = false; break; }
          slot = nprops;
#pragma unroll
          for (int k = 0; k < kUniMaxProps; k++) if (k == nprops) prop_id[k] = nd.property;
          nprops++;
        }
        const int64_t t = nd.splitval;
#pragma unroll
        for (int k = 0; k < kUniMaxProps; k++) {
          if (k != slot) continue;
          const bool present = __ballot(thr[k] == t) != 0;
          if (!present) {
            if (nthr[k] >= 63) { grid_ok = false; break; }
            const int pos = __popcll(__ballot(thr[k] < t));
            const int64_t below = __shfl_up(thr[k], 1);
            thr[k] = lane > pos ? below : (lane == pos ? t : thr[k]);
            nthr[k]++;
          }
        }
        if (!grid_ok) break;
        if (sp >= cap) { grid_ok = false; break; }
        stack[sp++] = nd.b;
        node = nd.a;
      }
      if (pop) {
        if (sp == 0) break;
        node = JXL_RFL(stack[--sp]);
      }
    }
  }
  int strd[kUniMaxProps] = {0, 0, 0, 0}, lane0[kUniMaxProps] = {0, 0, 0, 0};
  uint32_t cells = 1;
  int max_thr = 0;
#pragma unroll
  for (int k = 0; k < kUniMaxProps; k++) max_thr = max(max_thr, nthr[k]);
  const bool layout32 = nprops <= 2;
  if (max_thr > (layout32 ? 32 : 16)) grid_ok = false;
  if (grid_ok) {
#pragma unroll
    for (int k = 0; k < kUniMaxProps; k++)
      if (k < nprops) {
        strd[k] = (int)cells;
        lane0[k] = layout32 ? 32 * k : 16 * k;
        cells *= (uint32_t)(nthr[k] + 1);
        if (cells > (uint32_t)grid_cells) grid_ok = false;
      }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  int all_pred = -1;
  if (grid_ok) {
    // every cell: representative values of its buckets (bucket q of sorted thresholds t: values > t[q-1] and <= t[q]), one walk
    bool mixed = false, unknown_pred = false;
    uint32_t first_pred = 0;
    for (uint32_t c0 = 0; c0 < cells; c0 += 64) {
      const uint32_t cell = c0 + (uint32_t)lane;
      int64_t rep[kUniMaxProps] = {0, 0, 0, 0};
      uint32_t rest = cell < cells ? cell : 0u;
#pragma unroll
      for (int k = kUniMaxProps - 1; k >= 0; k--)
        if (k < nprops) {
          const uint32_t q = rest / (uint32_t)strd[k];
          rest -= q * (uint32_t)strd[k];
          const int64_t lo = __shfl(thr[k], q ? (int)q - 1 : 0);
          rep[k] = q ? lo + 1 : lo;
        }
      uint32_t node = root;
      I4 v = tree[node];
      for (int guard = 0; v.x >= 0 && guard < (1 << 20); guard++) {
        int64_t p = v.x == 0 ? (int64_t)chan : (int64_t)sid;
#pragma unroll
        for (int k = 0; k < kUniMaxProps; k++) if (k < nprops && prop_id[k] == v.x) p = rep[k];
        node = p > (int64_t)v.y ? (uint32_t)v.z : (uint32_t)v.w;
        v = tree[node];
      }
      const uint32_t a = (uint32_t)v.z;
      if (c0 == 0) first_pred = JXL_RFL(a & 0xFF);
      if (__ballot(cell < cells && (a & 0xFF) != first_pred)) mixed = true;
      if (cell < cells) {
        U4 rec;
        const uint32_t cl = c.cmap[a >> 8];
        const bool plain = (uint32_t)v.w == 1u && v.y == 0;
        if ((a & 0xFF) > 13) unknown_pred = true;
        rec.x = (a & 0xF) | cl << 4 | (c.cfg[cl] & 0xFFFu) << 12 | (plain ? 1u << 24 : 0u);
        rec.y = (uint32_t)v.w;      // multiplier
        rec.z = (uint32_t)v.y;      // offset
        rec.w = a >> 8;             // context (diagnostics)
        ((JXL_LDS U4*)lds_grid)[cell] = rec;
      }
    }
    if (!mixed) all_pred = (int)first_pred;
    if (__ballot(unknown_pred)) grid_ok = false;   // a predictor id the format does not have: the tree walk's default case deals with it
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // ---- per-lane tables of the grid: thresholds of all slots in one register, the coefficients of each lane's property
  c.grid_ok = grid_ok;
  c.leaf_in_lanes = grid_ok && cells <= 64;
  c.layout32 = layout32;
  c.stride1 = (uint32_t)strd[1]; c.stride2 = (uint32_t)strd[2]; c.stride3 = (uint32_t)strd[3];
  c.vthr = INT64_MAX;       // unused lanes: no value is greater, they never count
  c.vthr32 = INT32_MAX;
  int vpid = 0;
#pragma unroll
  for (int k = 0; k < kUniMaxProps; k++) {
    if (grid_ok && k < nprops) {
      const int64_t t = __shfl(thr[k], (lane - lane0[k]) & 63);
      if (lane >= lane0[k] && lane < lane0[k] + nthr[k]) { c.vthr = t; c.vthr32 = (int32_t)t; vpid = prop_id[k]; }
    }
  }
  c.vcW = (vpid == 5 || vpid == 7 || vpid == 8 || vpid == 9 || vpid == 10 || vpid == 14) ? 1 : 0;
  c.vcN = (vpid == 4 || vpid == 6 || vpid == 9 || vpid == 12 || vpid == 13) ? 1 : (vpid == 11 ? -1 : 0);
  c.vcNW = vpid == 11 ? 1 : ((vpid == 9 || vpid == 10) ? -1 : 0);
  c.vcNE = vpid == 12 ? -1 : 0;
  c.vcNN = vpid == 13 ? -1 : 0;
  c.vcWW = vpid == 14 ? -1 : 0;
  c.vcX = vpid == 3 ? 1 : 0;
  c.vcY = vpid == 2 ? 1 : 0;
  c.vcE = vpid == 15 ? 1 : 0;
  c.vabs = vpid == 4 || vpid == 5;
  c.vp8 = vpid == 8;
  c.use_far = __ballot(vpid == 12 || vpid == 13 || vpid == 14) != 0;
  c.use_xy = __ballot(vpid == 2 || vpid == 3) != 0;
  c.use_e = __ballot(vpid == 15) != 0;
  c.use_p8 = __ballot(vpid == 8) != 0;
  c.use_abs = __ballot(vpid == 4 || vpid == 5) != 0;
  c.vleaf0 = 0; c.vleaf1 = 0; c.vleaf2 = 0;
  if (c.leaf_in_lanes) {
    const U4 rec = c.grid[(uint32_t)lane < cells ? lane : 0];
    c.vleaf0 = (int32_t)rec.x; c.vleaf1 = (int32_t)rec.y; c.vleaf2 = (int32_t)rec.z;
  }
  // ---- tables held one entry per lane
  c.vdiv = (int32_t)((1u << 24) / (uint32_t)(lane + 1));                       // Div(i) = 2^24 / (i + 1), i < 64
  const uint32_t ncl = JXL_RFL(dc->num_clusters);
  c.vcfg = (int32_t)((uint32_t)lane < ncl ? c.cfg[lane] : 0u);
  c.cfg_in_lanes = ncl <= 64;
  if (use_wp) for (int i = lane; i < 10 * c.w2; i += 64) lds_wp[i] = 0;
  // the stream state in scalar registers
  c.s_state = JXL_RFL(state_io); c.s_rd = JXL_RFL(b.rd);
  c.s_buf = ((uint64_t)JXL_RFL((uint32_t)(b.buf >> 32)) << 32) | JXL_RFL((uint32_t)b.buf);
  c.s_n = (int)JXL_RFL(b.n);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // the row loops, specialised for what a channel's leaves usually share (everything else: the generic instance)
  if (use_wp) {
    if (all_pred == 6) UniRows<true, 6>(c, b);
    else UniRows<true, -1>(c, b);
  } else {
    if (all_pred == 5) UniRows<false, 5>(c, b);
    else UniRows<false, -1>(c, b);
  }
  state_io = c.s_state; b.buf = c.s_buf; b.n = c.s_n; b.rd = c.s_rd;
  return true;
}
