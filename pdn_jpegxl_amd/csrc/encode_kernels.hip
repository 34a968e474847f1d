// HIP kernels (gfx950) of the encode path behind SaveImage (reference: Encoder/JxlEncoder.cpp:33-144, the
// arithmetic it delegates to JxlEncoderAddImageFrame :128).
//
//   enc_analyze_kernel        GetOutputPixelFormat (:33-77): is every pixel gray / opaque?                 (reduction)
//   enc_xyb_kernel            PixelFormatConversion (BGRA -> channels) + sRGB -> linear -> XYB              (elementwise)
//   enc_sharpen_pad_kernel    inverse-Gaborish pre-sharpening, edge replication to whole 8x8 cells          (3x3 stencil)
//   enc_activity / enc_strategy / enc_varblock_kernel
//                             per-cell activity; 8x8 / 16x16 / 32x32 DCT per aligned region from it (effort), adaptive quant
//                             field; one workgroup per 32x32 region: DCTs, LF + HF quantisation with chroma from luma,
//                             coefficients in scan order, non-zero statistics
//   enc_*_tokens_kernel       context modelling: (context, value) tokens + histograms, all data-parallel
//   enc_sections_kernel       ANS coding, one lane per section (reverse pass for the state, forward pass for the bits)
#include <hip/hip_runtime.h>
#include "enc_types.h"
#include "kernels.h"

namespace jxlhip {

namespace {

__device__ __forceinline__ uint32_t PackSignedD(int32_t v) { return v >= 0 ? (uint32_t)v << 1 : (((uint32_t)(-(int64_t)v)) << 1) - 1; }
__device__ __forceinline__ int CeilLog2E(uint32_t x) { return x <= 1 ? 0 : 32 - __clz(x - 1); }
__device__ __forceinline__ int MirrorE(int v, int n) {
  while (v < 0 || v >= n) v = v < 0 ? -v - 1 : 2 * n - 1 - v;
  return v;
}
// hybrid-uint token of `v` under the config (split_exponent 4, msb_in_token 2, lsb_in_token 0)
__device__ __forceinline__ void HybridD(uint32_t v, uint32_t* tok, uint32_t* nbits, uint32_t* bits) {
  if (v < 16) { *tok = v; *nbits = 0; *bits = 0; return; }
  const uint32_t n = 31 - __clz(v), m = v - (1u << n);
  *tok = 16 + ((n - 4) << 2) + (m >> (n - 2));
  *nbits = n - 2;
  *bits = m & ((1u << (n - 2)) - 1);
}
__device__ __forceinline__ int32_t GradientPred(int32_t W, int32_t N, int32_t NW) {
  const int64_t mn = W < N ? W : N, mx = W < N ? N : W, gr = (int64_t)W + N - NW;
  return (int32_t)(gr < mn ? mn : (gr > mx ? mx : gr));
}

__device__ const uint8_t e_nnz_ctx[64] = {0,   0,   31,  62,  62,  93,  93,  93,  93,  123, 123, 123, 123, 152, 152, 152, 152, 152, 152, 152, 152, 180,
                                          180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206,
                                          206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206};

}  // namespace

// ------------------------------------------------------------------ pixel format analysis (Encoder/JxlEncoder.cpp:33-77)
__global__ void enc_analyze_kernel(EncImage im) {
  uint32_t notgray = 0, alpha = 0;
  const size_t n = (size_t)im.w * im.h;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % im.w), y = (int)(i / im.w);
    const uchar4 p = *(const uchar4*)(im.bgra + (size_t)y * im.stride + (size_t)x * 4);   // B, G, R, A
    notgray |= (p.x != p.y) | (p.y != p.z);
    alpha |= p.w < 255;
  }
  if (__any(notgray) && (threadIdx.x & 63) == 0) atomicOr(&im.flags[0], 1u);
  if (__any(alpha) && (threadIdx.x & 63) == 0) atomicOr(&im.flags[1], 1u);
}

// ------------------------------------------------------------------ BGRA8 -> XYB (+ alpha plane)
// Channel selection as PixelFormatConversion.cpp:16-121 (gray takes the B channel); sRGB decoding, opsin absorbance.
__global__ void enc_xyb_kernel(EncImage im) {
  const size_t n = (size_t)im.w * im.h;
  const float kM[9] = {0.30f, 0.622f, 0.078f, 0.23f, 0.692f, 0.078f, 0.24342268924547819f, 0.20476744424496821f, 0.55180986650955360f};
  const float kB = 0.0037930732552754493f;
  const float cb = cbrtf(kB);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % im.w), y = (int)(i / im.w);
    const uchar4 p = *(const uchar4*)(im.bgra + (size_t)y * im.stride + (size_t)x * 4);
    const uint8_t r8 = im.gray ? p.x : p.z, g8 = im.gray ? p.x : p.y, b8 = p.x;
    auto lin = [](uint8_t v8) {
      const float v = (float)v8 * (1.0f / 255.0f);
      return v <= 0.04045f ? v / 12.92f : powf((v + 0.055f) / 1.055f, 2.4f);
    };
    float r = lin(r8), g = lin(g8), b = lin(b8);
    if (im.icc_lin) {   // the document's own tone curves and primaries
      const float pr = im.icc_lin[r8], pg = im.icc_lin[256 + g8], pb = im.icc_lin[512 + b8];
      r = im.icc_to_srgb[0] * pr + im.icc_to_srgb[1] * pg + im.icc_to_srgb[2] * pb;
      g = im.icc_to_srgb[3] * pr + im.icc_to_srgb[4] * pg + im.icc_to_srgb[5] * pb;
      b = im.icc_to_srgb[6] * pr + im.icc_to_srgb[7] * pg + im.icc_to_srgb[8] * pb;
    }
    float mr = kM[0] * r + kM[1] * g + kM[2] * b + kB;
    float mg = kM[3] * r + kM[4] * g + kM[5] * b + kB;
    float mb = kM[6] * r + kM[7] * g + kM[8] * b + kB;
    mr = fmaxf(mr, 0.f); mg = fmaxf(mg, 0.f); mb = fmaxf(mb, 0.f);
    const float gr = cbrtf(mr) - cb, gg = cbrtf(mg) - cb, gb = cbrtf(mb) - cb;
    im.xyb[0][i] = 0.5f * (gr - gg);
    im.xyb[1][i] = 0.5f * (gr + gg);
    im.xyb[2][i] = gb;
    if (im.has_alpha) im.alpha_px[i] = p.w;
  }
}

// ------------------------------------------------------------------ 2*I - Gaborish, padded to whole cells
__global__ void enc_sharpen_pad_kernel(EncImage im) {
  const size_t n = (size_t)im.wp * im.hp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = min((int)(i % im.wp), im.w - 1), y = min((int)(i / im.wp), im.h - 1);
    const int xl = MirrorE(x - 1, im.w), xr = MirrorE(x + 1, im.w), yt = MirrorE(y - 1, im.h), yb = MirrorE(y + 1, im.h);
    for (int c = 0; c < 3; c++) {
      const float* p = im.xyb[c];
      const float v = p[(size_t)y * im.w + x];
      float o = v;
      if (im.gab) {
        const float* t = p + (size_t)yt * im.w;
        const float* m = p + (size_t)y * im.w;
        const float* b = p + (size_t)yb * im.w;
        const float blur = m[x] * im.gab_w[c][0] + (t[x] + b[x] + m[xl] + m[xr]) * im.gab_w[c][1] + (t[xl] + t[xr] + b[xl] + b[xr]) * im.gab_w[c][2];
        o = 2.0f * v - blur;
      }
      im.pad[c][i] = o;
    }
  }
}

// ------------------------------------------------------------------ varblocks: strategy choice, DCT, quantisation
// Activity of a cell: standard deviation of its 64 Y samples (double accumulation: the strategy thresholds below compare it).
__global__ void enc_activity_kernel(EncImage im) {
  const int ncell = im.w8 * im.h8;
  for (int cell = blockIdx.x * blockDim.x + threadIdx.x; cell < ncell; cell += gridDim.x * blockDim.x) {
    const int bx = cell % im.w8, by = cell / im.w8;
    const float* p = im.pad[1] + (size_t)by * 8 * im.wp + bx * 8;
    double s = 0, s2 = 0;
    for (int y = 0; y < 8; y++)
      for (int x = 0; x < 8; x++) {
        const float v = p[(size_t)y * im.wp + x];
        s += v; s2 += (double)v * v;
      }
    const double var = s2 / 64 - (s / 64) * (s / 64);
    im.act[cell] = (float)sqrt(var > 0 ? var : 0.0);
  }
}

// Strategy tables (the shapes this encoder uses): cells covered across / down (log2), order bucket, quantisation table.
__device__ const uint8_t e_lcx[27] = {0, 0, 0, 0, 1, 2, 0, 1, 0, 2, 1, 2, 0, 0, 0, 0, 0, 0, 3, 2, 3, 4, 3, 4, 5, 4, 5};
__device__ const uint8_t e_lcy[27] = {0, 0, 0, 0, 1, 2, 1, 0, 2, 0, 2, 1, 0, 0, 0, 0, 0, 0, 3, 3, 2, 4, 4, 3, 5, 5, 4};
__device__ const uint8_t e_bucket[27] = {0, 1, 1, 1, 2, 3, 4, 4, 5, 5, 6, 6, 1, 1, 1, 1, 1, 1, 7, 8, 8, 9, 10, 10, 11, 12, 12};
__device__ const uint8_t e_qtable[27] = {0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16};

// One thread per aligned 64x64 region (8 x 8 cells).  The choice is the raster-greedy one: cells in raster order, at every free cell the
// largest shape of the list that is aligned to its own size there, lies inside the frame, covers only free cells and whose cells' maximum
// activity stays under the shape's threshold.  Shapes of at most 64 points aligned to their size never leave their aligned 64x64
// region, so the regions decide independently.  The quant field of a varblock follows its mean activity (finer steps in flat areas).
__global__ void enc_strategy_kernel(EncImage im) {
  const int rw = (im.w8 + 7) / 8, rh = (im.h8 + 7) / 8;
  // strategy code, threshold; `squares` == 1 keeps the square shapes of at most 32 points
  const int kShape[9] = {18, 19, 20, 5, 10, 11, 4, 6, 7};   // 64x64, 64x32, 32x64, 32x32, 32x16, 16x32, 16x16, 16x8, 8x16
  const float kThr[9] = {0.0035f, 0.0045f, 0.0045f, 0.007f, 0.009f, 0.009f, 0.016f, 0.024f, 0.024f};
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rw * rh; r += gridDim.x * blockDim.x) {
    const int bx0 = (r % rw) * 8, by0 = (r / rw) * 8;
    uint64_t occ = 0;   // cells of the region already covered (bit iy * 8 + ix)
    for (int iy = 0; iy < 8 && by0 + iy < im.h8; iy++)
      for (int ix = 0; ix < 8 && bx0 + ix < im.w8; ix++) {
        if (occ >> (iy * 8 + ix) & 1) continue;
        int s = 0;
        if (im.squares)
          for (int t = 0; t < 9; t++) {
            const int cand = kShape[t];
            if (im.squares == 1 && cand != 5 && cand != 4) continue;
            const int cx = 1 << e_lcx[cand], cy = 1 << e_lcy[cand];
            if ((ix & (cx - 1)) || (iy & (cy - 1)) || bx0 + ix + cx > im.w8 || by0 + iy + cy > im.h8) continue;
            const uint64_t rowm = ((1ull << cx) - 1) << ix;
            uint64_t m = 0;
            for (int y = 0; y < cy; y++) m |= rowm << ((iy + y) * 8);
            if (m & occ) continue;
            float mx = 0.f;
            for (int y = 0; y < cy; y++)
              for (int x = 0; x < cx; x++) mx = fmaxf(mx, im.act[(size_t)(by0 + iy + y) * im.w8 + bx0 + ix + x]);
            if (mx < kThr[t]) { s = cand; break; }
          }
        const int cx = 1 << e_lcx[s], cy = 1 << e_lcy[s];
        double a = 0;
        for (int y = 0; y < cy; y++)
          for (int x = 0; x < cx; x++) a += im.act[(size_t)(by0 + iy + y) * im.w8 + bx0 + ix + x];
        a /= cx * cy;
        const double mod = 0.7 + 0.8 / (1.0 + a / 0.012);
        int q = (int)rint(16.0 * mod);
        q = max(1, min(256, q));
        for (int y = 0; y < cy; y++)
          for (int x = 0; x < cx; x++) {
            const size_t cell = (size_t)(by0 + iy + y) * im.w8 + bx0 + ix + x;
            im.strat[cell] = (uint8_t)(s | ((x | y) == 0 ? 0x80 : 0));
            im.rawq[cell] = q;
            occ |= 1ull << ((iy + y) * 8 + ix + x);
          }
      }
  }
}

// One workgroup per 64x64 region, one channel at a time (Y first: X needs nothing from it, B the dequantised Y for chroma from luma):
// every sample position of the region belongs to exactly one varblock, so both DCT passes run over all 4096 positions at once (a
// thread's loop length is its varblock's height / width), through LDS.  Then per coefficient: its scan position, quantisation,
// non-zero statistics; and per cell the quantised LF from the varblock's lowest cy x cx coefficients.
__global__ __launch_bounds__(256) void enc_varblock_kernel(EncImage im) {
  constexpr int P = 65;
  __shared__ float s_a[64 * P];    // pixels, later the coefficients of the channel
  __shared__ float s_t[64 * P];    // after the vertical pass
  __shared__ float s_yd[64 * P];   // dequantised Y coefficients (chroma from luma of B)
  __shared__ uint8_t s_strat[64];
  __shared__ uint32_t s_nz[64], s_last[64];
  __shared__ float s_fy[64];       // dequantised LF of Y per cell
  const int rw = (im.w8 + 7) / 8;
  const int bx0 = (blockIdx.x % rw) * 8, by0 = (blockIdx.x / rw) * 8;
  const int tid = threadIdx.x;
  if (tid < 64) {
    const int ix = tid & 7, iy = tid >> 3;
    s_strat[tid] = (bx0 + ix < im.w8 && by0 + iy < im.h8) ? im.strat[(size_t)(by0 + iy) * im.w8 + bx0 + ix] : 0xFF;   // 0xFF: outside the frame
  }
  __syncthreads();
  // geometry of the varblock that holds region position (y, x)
  auto geom = [&](int y, int x, int* s, int* oy, int* ox) -> bool {
    const uint32_t st = s_strat[(y >> 3) * 8 + (x >> 3)];
    if (st == 0xFF) return false;
    const int sc = (int)(st & 0x7F);
    *s = sc;
    *oy = y & ~((8 << e_lcy[sc]) - 1);
    *ox = x & ~((8 << e_lcx[sc]) - 1);
    return true;
  };
  for (int ci = 0; ci < 3; ci++) {
    const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
    if (tid < 64) { s_nz[tid] = 0; s_last[tid] = 0; }
    for (int i = tid; i < 4096; i += 256) {
      const int y = i >> 6, x = i & 63;
      const bool in = bx0 * 8 + x < im.wp && by0 * 8 + y < im.hp;
      s_a[y * P + x] = in ? im.pad[c][(size_t)(by0 * 8 + y) * im.wp + bx0 * 8 + x] : 0.f;
    }
    __syncthreads();
    // vertical: t[oy + ky][x] = sum_y px[oy + y][x] * (B_R[ky][y] / R)
    for (int i = tid; i < 4096; i += 256) {
      const int y = i >> 6, x = i & 63;
      int s, oy, ox;
      if (!geom(y, x, &s, &oy, &ox)) continue;
      const int lr = e_lcy[s], R = 8 << lr, ky = y - oy;
      const float* b = im.basis_div[lr] + ky * R;
      float acc = 0.f;
      for (int k = 0; k < R; k++) acc += b[k] * s_a[(oy + k) * P + x];
      s_t[y * P + x] = acc;
    }
    __syncthreads();
    // horizontal: coef[ky][kx] = (sum_x t[ky][ox + x] * B_C[kx][x]) / C
    for (int i = tid; i < 4096; i += 256) {
      const int y = i >> 6, x = i & 63;
      int s, oy, ox;
      if (!geom(y, x, &s, &oy, &ox)) continue;
      const int lc = e_lcx[s], C = 8 << lc, kx = x - ox;
      const float* b = im.basis[lc] + kx * C;
      float acc = 0.f;
      for (int k = 0; k < C; k++) acc += s_t[y * P + ox + k] * b[k];
      s_a[y * P + x] = acc / (float)C;   // every thread reads s_t and writes its own s_a element: no hazard
    }
    __syncthreads();
    // quantisation of the HF coefficients
    for (int i = tid; i < 4096; i += 256) {
      const int y = i >> 6, x = i & 63;
      int s, oy, ox;
      if (!geom(y, x, &s, &oy, &ox)) continue;
      const int lcx = e_lcx[s], lcy = e_lcy[s], cx = 1 << lcx, cy = 1 << lcy, R = 8 << lcy, C = 8 << lcx;
      const int ky = y - oy, kx = x - ox;
      // stored layout: the longer side runs along a row (squares: transposed)
      const uint32_t p = R >= C ? (uint32_t)(kx * R + ky) : (uint32_t)(ky * C + kx);
      const uint32_t k = im.scan_of[e_bucket[s]][p];
      const int fcell = (oy >> 3) * 8 + (ox >> 3);         // first cell of the varblock, within the region
      const size_t first = (size_t)(by0 + (oy >> 3)) * im.w8 + bx0 + (ox >> 3);
      const float scale = im.inv_gs / (float)im.rawq[first];
      const float dqs = c == 1 ? scale : (c == 0 ? scale * im.x_dm : scale * im.b_dm);
      const float coef = s_a[y * P + x];
      int32_t qi = 0;
      if (!(ky < cy && kx < cx)) {   // the lowest cy x cx coefficients travel with the LF image
        const float w = im.dq[e_qtable[s]][(size_t)c * R * C + p];
        const float target = c == 2 ? coef - s_yd[y * P + x] : coef;
        const float v = target / (dqs * w);
        qi = fabsf(v) < 0.6f ? 0 : (int32_t)rintf(v);
        if (c == 1) {
          const float adj = qi == 0 ? 0.f : (abs(qi) == 1 ? (qi > 0 ? im.qbias1 : -im.qbias1) : (float)qi - im.qbias3 / (float)qi);
          s_yd[y * P + x] = adj * dqs * w;
        }
      } else if (c == 1) {
        s_yd[y * P + x] = 0.f;
      }
      // scan position k lives in the slot of the varblock's covered cell number k >> 6 (row-major over cy rows of cx cells)
      const uint32_t cj = k >> 6;
      const size_t slot = (first + (size_t)(cj >> lcx) * im.w8 + (cj & (cx - 1))) * 64 + (k & 63);
      im.qs[c][slot] = qi;
      if (qi) { atomicAdd(&s_nz[fcell], 1u); atomicMax(&s_last[fcell], k); }
    }
    __syncthreads();
    // per cell: non-zero context value of its varblock, LF from the lowest cy x cx coefficients; per first cell: counts
    if (tid < 64) {
      const int ix = tid & 7, iy = tid >> 3;
      const uint32_t st = s_strat[tid];
      if (st != 0xFF) {
        int s, oy, ox;
        geom(iy * 8, ix * 8, &s, &oy, &ox);
        const int lcx = e_lcx[s], lcy = e_lcy[s], cx = 1 << lcx, cy = 1 << lcy;
        const int fcell = (oy >> 3) * 8 + (ox >> 3), py = iy - (oy >> 3), px = ix - (ox >> 3);
        const size_t cell = (size_t)(by0 + iy) * im.w8 + bx0 + ix;
        // IDCT of the cy x cx lowest coefficients, undoing the resampling scale: horizontal then vertical
        const float* Bx = im.bsmall[lcx];
        const float* By = im.bsmall[lcy];
        const float* rs = im.rs + (lcy * 4 + lcx) * 64;
        float lf = 0.f;
        for (int ky = 0; ky < cy; ky++) {
          float t = 0.f;
          for (int kx = 0; kx < cx; kx++) t += (s_a[(oy + ky) * P + ox + kx] / rs[ky * 8 + kx]) * Bx[kx * cx + px];
          lf += By[ky * cy + py] * t;
        }
        if (c == 1) {
          const int32_t qy = (int32_t)rintf(lf * im.inv_mul_lf[1]);
          s_fy[tid] = (float)qy * im.mul_lf_y;
          im.lfq[1][cell] = qy;
        } else if (c == 0) {
          im.lfq[0][cell] = (int32_t)rintf(lf * im.inv_mul_lf[0]);
        } else {
          im.lfq[2][cell] = (int32_t)rintf((lf - s_fy[tid]) * im.inv_mul_lf[2]);
        }
        const uint32_t n = s_nz[fcell];
        im.nz[c][cell] = (uint8_t)((n + (uint32_t)(cx * cy) - 1) >> (lcx + lcy));
        if (st & 0x80) { im.nzc[c][cell] = (uint16_t)n; im.last[c][cell] = (uint16_t)s_last[fcell]; }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ tokens: LF coefficients (gradient predictor)
__global__ __launch_bounds__(256) void enc_lf_tokens_kernel(EncImage im) {
  __shared__ uint32_t s_h[3 * kEncSyms];
  for (int i = threadIdx.x; i < 3 * (int)kEncSyms; i += 256) s_h[i] = 0;
  __syncthreads();
  const int g = blockIdx.y;
  const int gx = g % im.xlf, gy = g / im.xlf;
  const int bx0 = gx * kLfGroupBlocks, by0 = gy * kLfGroupBlocks;
  const int bw = min(kLfGroupBlocks, im.w8 - bx0), bh = min(kLfGroupBlocks, im.h8 - by0);
  const int n = 3 * bw * bh;
  DevToken* out = im.tok_lf + (size_t)g * kLfTokCap;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int mc = i / (bw * bh), r = i % (bw * bh), y = r / bw, x = r % bw;
    const int c = mc == 0 ? 1 : (mc == 1 ? 0 : 2);   // modular channel order Y, X, B
    const int32_t* pl = im.lfq[c] + (size_t)by0 * im.w8 + bx0;
    const int32_t v = pl[(size_t)y * im.w8 + x];
    int32_t W, N, NW;
    if (x == 0) { W = y ? pl[(size_t)(y - 1) * im.w8] : 0; N = W; NW = W; }
    else {
      W = pl[(size_t)y * im.w8 + x - 1];
      N = y ? pl[(size_t)(y - 1) * im.w8 + x] : W;
      NW = y ? pl[(size_t)(y - 1) * im.w8 + x - 1] : W;
    }
    const uint32_t val = PackSignedD(v - GradientPred(W, N, NW));
    DevToken t;
    t.ctx = mc == 0 ? kLeafLfY : (mc == 1 ? kLeafLfX : kLeafLfB);
    t.value = val;
    out[i] = t;
    uint32_t tok, nb, bits;
    HybridD(val, &tok, &nb, &bits);
    atomicAdd(&s_h[mc * kEncSyms + tok], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * (int)kEncSyms; i += 256) {
    const uint32_t leaf = i / kEncSyms == 0 ? kLeafLfY : (i / kEncSyms == 1 ? kLeafLfX : kLeafLfB);
    if (s_h[i]) atomicAdd(&im.hist_mod[leaf * kEncSyms + (i % kEncSyms)], s_h[i]);
  }
}

// Block info of an LF group: its varblocks in raster order of their first cells, row 0 = strategies (Zero predictor), row 1 = quant
// field - 1 (West predictor; the first entry's West is the sample above it, i.e. the first strategy).  One workgroup per LF group:
// first cells per 64-cell run (one ballot each), a scan over the runs, then both rows are written in parallel.
__global__ __launch_bounds__(1024) void enc_meta_tokens_kernel(EncImage im) {
  __shared__ uint32_t s_hq[kEncSyms], s_hs[kEncSyms];
  __shared__ uint32_t s_run[1024 + 1];   // first cells before run r (65536 cells / 64)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < (int)kEncSyms; i += 1024) { s_hq[i] = 0; s_hs[i] = 0; }
  const int g = blockIdx.x;
  const int gx = g % im.xlf, gy = g / im.xlf;
  const int bx0 = gx * kLfGroupBlocks, by0 = gy * kLfGroupBlocks;
  const int bw = min(kLfGroupBlocks, im.w8 - bx0), bh = min(kLfGroupBlocks, im.h8 - by0);
  const int n = bw * bh, nrun = (n + 63) / 64;
  auto cell_of = [&](int k) { return (size_t)(by0 + k / bw) * im.w8 + bx0 + k % bw; };
  for (int r = wave; r < nrun; r += 16) {
    const int k = r * 64 + lane;
    const uint64_t m = __ballot(k < n && (im.strat[cell_of(k)] & 0x80));
    if (lane == 0) s_run[r + 1] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t acc = 0;
    s_run[0] = 0;
    for (int i = 1; i <= nrun; i++) { acc += s_run[i]; s_run[i] = acc; }
  }
  __syncthreads();
  const uint32_t count = s_run[nrun];
  DevToken* out = im.tok_meta + (size_t)g * kMetaTokCap;
  const bool with_strategies = im.squares != 0;   // 8x8 only: the strategy row is a constant channel, no token is written for it
  const uint32_t qbase = with_strategies ? count : 0;
  for (int r = wave; r < nrun; r += 16) {
    const int k = r * 64 + lane;
    const bool first = k < n && (im.strat[cell_of(k)] & 0x80);
    const uint64_t m = __ballot(first);
    if (!first) continue;
    const uint32_t idx = s_run[r] + (uint32_t)__popcll(m & ((1ull << lane) - 1));
    const size_t cell = cell_of(k);
    const int32_t strategy = im.strat[cell] & 0x7F;
    uint32_t tok, nb, bits;
    DevToken t;
    if (with_strategies) {
      t.ctx = kLeafStrategy;
      t.value = PackSignedD(strategy);
      out[idx] = t;
      HybridD(t.value, &tok, &nb, &bits);
      atomicAdd(&s_hs[tok], 1u);
    }
    // West of the quant row: the previous varblock's value (walk back to the previous first cell); for the first varblock the
    // sample above, which is its strategy
    int32_t W = strategy;
    if (idx) {
      int kk = k - 1;
      while (!(im.strat[cell_of(kk)] & 0x80)) kk--;
      W = im.rawq[cell_of(kk)] - 1;
    }
    t.ctx = kLeafQf;
    t.value = PackSignedD(im.rawq[cell] - 1 - W);
    out[qbase + idx] = t;
    HybridD(t.value, &tok, &nb, &bits);
    atomicAdd(&s_hq[tok], 1u);
  }
  __syncthreads();
  if (tid == 0) im.n_meta[g] = qbase + count;
  for (int i = tid; i < (int)kEncSyms; i += 1024) {
    if (s_hq[i]) atomicAdd(&im.hist_mod[kLeafQf * kEncSyms + i], s_hq[i]);
    if (s_hs[i]) atomicAdd(&im.hist_mod[kLeafStrategy * kEncSyms + i], s_hs[i]);
  }
}

// alpha: one lossless Modular channel per group, gradient predictor
__global__ __launch_bounds__(256) void enc_alpha_tokens_kernel(EncImage im) {
  __shared__ uint32_t s_h[kEncSyms];
  for (int i = threadIdx.x; i < (int)kEncSyms; i += 256) s_h[i] = 0;
  __syncthreads();
  const int g = blockIdx.y;
  const int gx = g % im.xg, gy = g / im.xg;
  const int x0 = gx * kGroupDim, y0 = gy * kGroupDim;
  const int gw = min(kGroupDim, im.w - x0), gh = min(kGroupDim, im.h - y0);
  const int n = gw * gh;
  DevToken* out = im.tok_alpha + (size_t)g * kAlphaTokCap;
  const int32_t* pl = im.alpha_px + (size_t)y0 * im.w + x0;
  const uint32_t leaf = im.ng == 1 ? kLeafAlphaGlobal : kLeafAlpha;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int y = i / gw, x = i % gw;
    const int32_t v = pl[(size_t)y * im.w + x];
    int32_t W, N, NW;
    if (x == 0) { W = y ? pl[(size_t)(y - 1) * im.w] : 0; N = W; NW = W; }
    else {
      W = pl[(size_t)y * im.w + x - 1];
      N = y ? pl[(size_t)(y - 1) * im.w + x] : W;
      NW = y ? pl[(size_t)(y - 1) * im.w + x - 1] : W;
    }
    DevToken t;
    t.ctx = leaf;
    t.value = PackSignedD(v - GradientPred(W, N, NW));
    out[i] = t;
    uint32_t tok, nb, bits;
    HybridD(t.value, &tok, &nb, &bits);
    atomicAdd(&s_h[tok], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (int)kEncSyms; i += 256)
    if (s_h[i]) atomicAdd(&im.hist_mod[leaf * kEncSyms + i], s_h[i]);
}

// ------------------------------------------------------------------ tokens: HF coefficients of one group
// Token counts per (varblock, channel) are known from the block kernel, so an exclusive scan over the group's cells (zero for
// cells that are not the first of a varblock) gives every (varblock, channel) its slot and all of them are tokenised in parallel.
__global__ __launch_bounds__(256) void enc_ac_tokens_kernel(EncImage im) {
  __shared__ uint32_t s_cnt[3072];
  __shared__ uint32_t s_part[256];
  const int g = blockIdx.x;
  const int gx = g % im.xg, gy = g / im.xg;
  const int bx0 = gx * kGroupBlocks, by0 = gy * kGroupBlocks;
  const int bw = min(kGroupBlocks, im.w8 - bx0), bh = min(kGroupBlocks, im.h8 - by0);
  const int nent = bw * bh * 3;
  const int tid = threadIdx.x;
  for (int e = tid; e < 3072; e += 256) {
    uint32_t cnt = 0;
    if (e < nent) {
      const int blk = e / 3, ci = e % 3, c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
      const size_t cell = (size_t)(by0 + blk / bw) * im.w8 + bx0 + blk % bw;
      const uint32_t st = im.strat[cell];
      if (st & 0x80) {
        const uint32_t covered = 1u << (e_lcx[st & 0x7F] + e_lcy[st & 0x7F]);
        cnt = 1u + (im.nzc[c][cell] ? (uint32_t)im.last[c][cell] + 1u - covered : 0u);
      }
    }
    s_cnt[e] = cnt;
  }
  __syncthreads();
  uint32_t mine = 0;
  for (int k = 0; k < 12; k++) mine += s_cnt[tid * 12 + k];
  s_part[tid] = mine;
  __syncthreads();
  if (tid == 0) {
    uint32_t acc = 0;
    for (int i = 0; i < 256; i++) { const uint32_t v = s_part[i]; s_part[i] = acc; acc += v; }
    im.n_ac[g] = acc;
  }
  __syncthreads();
  DevToken* out = im.tok_ac + (size_t)g * kAcTokCap;
  uint32_t pos = s_part[tid];
  const uint32_t nbc = 15;
  for (int k = 0; k < 12; k++) {
    const int e = tid * 12 + k;
    if (e >= nent) break;
    const int blk = e / 3, ci = e % 3, c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
    const int bx = blk % bw, by = blk / bw;
    const size_t cell = (size_t)(by0 + by) * im.w8 + bx0 + bx;
    const uint32_t st = im.strat[cell];
    if (!(st & 0x80)) continue;
    const uint32_t sc = st & 0x7F, lcx = e_lcx[sc], cc = 1u << lcx, log2c = lcx + e_lcy[sc], covered = 1u << log2c, size = covered << 6;
    const uint8_t* nzp = im.nz[c];
    const uint32_t nzeros0 = im.nzc[c][cell];
    uint32_t predicted;
    if (bx == 0) predicted = by == 0 ? 32u : (uint32_t)nzp[cell - im.w8];
    else if (by == 0) predicted = nzp[cell - 1];
    else predicted = ((uint32_t)nzp[cell - im.w8] + nzp[cell - 1] + 1) >> 1;
    // default block-context map by order bucket: one row for Y, one shared by X and B
    const uint8_t kCtxY[13] = {0, 1, 2, 2, 3, 3, 4, 5, 6, 6, 6, 6, 6}, kCtxXB[13] = {7, 8, 9, 9, 10, 11, 12, 13, 14, 14, 14, 14, 14};
    const uint32_t block_ctx = c == 1 ? kCtxY[e_bucket[sc]] : kCtxXB[e_bucket[sc]];
    uint32_t nzc = predicted >= 64 ? 64 : predicted;
    nzc = nzc < 8 ? nzc : 4 + nzc / 2;
    DevToken t;
    t.ctx = nzc * nbc + block_ctx;
    t.value = nzeros0;
    out[pos++] = t;
    uint32_t tok, nb, bits;
    HybridD(t.value, &tok, &nb, &bits);
    atomicAdd(&im.hist_ac[(size_t)t.ctx * kEncSyms + tok], 1u);
    if (!nzeros0) continue;
    const uint32_t histo = nbc * 37 + 458 * block_ctx;
    uint32_t nzeros = nzeros0, prev = nzeros0 > size / 16 ? 0u : 1u;
    const uint32_t lastk = im.last[c][cell];
    for (uint32_t kk = covered; kk <= lastk; kk++) {
      const uint32_t cj = kk >> 6;
      const int32_t q = im.qs[c][(cell + (size_t)(cj >> lcx) * im.w8 + (cj & (cc - 1))) * 64 + (kk & 63)];
      const uint32_t ks = kk >> log2c;
      const uint32_t fctx = ks < 16 ? ks - 1 : (ks < 32 ? 15 + ((ks - 16) >> 1) : 23 + ((ks - 32) >> 2));
      t.ctx = histo + ((uint32_t)e_nnz_ctx[(nzeros + covered - 1) >> log2c] + fctx) * 2 + prev;
      t.value = PackSignedD(q);
      out[pos++] = t;
      HybridD(t.value, &tok, &nb, &bits);
      atomicAdd(&im.hist_ac[(size_t)t.ctx * kEncSyms + tok], 1u);
      prev = t.value != 0;
      nzeros -= prev;
    }
  }
}

// ------------------------------------------------------------------ ANS coding, one lane per section
namespace {

// ---- one WAVEFRONT per section.  rANS (12-bit precision, 16-bit renormalisation) is a recurrence over the tokens, back to front,
// that no amount of lanes can split - but everything around it can: per block of 64 tokens the lanes look up cluster, symbol,
// frequency, slot-map base and 1 / frequency in parallel, lane 0 alone runs the recurrence over those 64 prepared entries (a float
// multiply and a fix-up instead of an integer division, operands in LDS instead of HBM), and the lanes store the flushes coalesced.
// The bits are then laid out front to back by all lanes at once: a wavefront prefix sum of the token lengths gives every lane its
// bit offset, the lanes OR their bits into an LDS staging window, complete words go out coalesced.
struct WaveScratch {
  uint32_t flush[64];      // reverse pass: the flush word of each of the 64 tokens in flight
  uint32_t stage[128];   // bits [32 * wbase, ...) of the section, not yet written out
};

__device__ __forceinline__ void WaveSync() {
  // LDS operations of one wavefront execute in program order; this only keeps the compiler from reordering across the point
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct WaveWriter {
  uint32_t* out;
  uint32_t* stage;
  uint64_t bitpos;    // bits written so far (uniform)
  uint32_t wbase;     // words already in `out` (uniform)
  int lane;
  __device__ void Init(uint8_t* p, WaveScratch* sc, int l) {
    out = (uint32_t*)p; stage = sc->stage; bitpos = 0; wbase = 0; lane = l;
    stage[lane] = 0; stage[lane + 64] = 0;
    WaveSync();
  }
  // complete words leave the staging window (coalesced), the partial word moves to its front
  __device__ void Drain() {
    const uint32_t nfull = (uint32_t)(bitpos >> 5) - wbase;
    if (!nfull) return;
    WaveSync();
    for (uint32_t i = lane; i < nfull; i += 64) out[wbase + i] = stage[i];
    const uint32_t carry = stage[nfull];
    WaveSync();
    for (uint32_t i = lane; i <= nfull + 2 && i < 128; i += 64) stage[i] = 0;
    WaveSync();
    if (lane == 0) stage[0] = carry;
    WaveSync();
    wbase += nfull;
  }
  // the same (nbits <= 32, v) on every lane
  __device__ void PutUniform(int nbits, uint32_t v) {
    if (!nbits) return;
    if (lane == 0) {
      const uint64_t V = (uint64_t)(nbits == 32 ? v : (v & ((1u << nbits) - 1))) << (bitpos & 31);
      const uint32_t wi = (uint32_t)(bitpos >> 5) - wbase;
      stage[wi] |= (uint32_t)V;
      if (V >> 32) stage[wi + 1] |= (uint32_t)(V >> 32);
    }
    bitpos += (uint64_t)nbits;
    Drain();
  }
  // lane l appends `len` (<= 48) bits `V` after the bits of lanes < l
  __device__ void PutParallel(uint64_t V, uint32_t len) {
    uint32_t incl = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
      if (lane >= d) incl += up;
    }
    const uint32_t total = (uint32_t)__shfl((int)incl, 63);
    if (len) {
      const uint64_t o = bitpos + (incl - len);
      const uint32_t wi = (uint32_t)(o >> 5) - wbase, sh = (uint32_t)o & 31;
      const uint32_t w0 = (uint32_t)(V << sh);
      const uint64_t rest = V >> (32 - sh);   // bits 32.. of V << sh (sh == 0: V >> 32)
      if (w0) atomicOr(&stage[wi], w0);
      if ((uint32_t)rest) atomicOr(&stage[wi + 1], (uint32_t)rest);
      if (rest >> 32) atomicOr(&stage[wi + 2], (uint32_t)(rest >> 32));
    }
    bitpos += total;
    Drain();
  }
  __device__ uint64_t Finish() {   // returns the bit count, pads the last word with zeros
    Drain();
    if ((bitpos & 31) && lane == 0) out[wbase] = stage[0];
    return bitpos;
  }
};

// Reverse pass of one token stream (its flushes are recorded in the token array); returns the final state on every lane.
// kLdsRmap: the code's slot map was staged in LDS (the Modular code; the HF code's 512 KB map stays in global memory)
template <bool kLdsRmap>
__device__ uint32_t ReversePass(DevToken* tok, uint32_t n, const EncCodeDev& code, WaveScratch* sc, int lane) {
  typedef const __attribute__((address_space(3))) uint16_t* LdsU16;
  const LdsU16 lds_rmap = kLdsRmap ? (LdsU16)code.rmap : (LdsU16) nullptr;
  const DevToken kNone = {0u, 0u};
  const int64_t nblk = ((int64_t)n + 63) >> 6;
  uint32_t state = 0x130000u;   // lane 0's copy is the real one
  {
    DevToken nxt = kNone;
    if (nblk) { const int64_t i = (nblk - 1) * 64 + lane; nxt = i < (int64_t)n ? tok[i] : kNone; }
    for (int64_t b = nblk - 1; b >= 0; b--) {
      const int64_t idx = b * 64 + lane;
      const bool valid = idx < (int64_t)n;
      const DevToken t = nxt;
      if (b > 0) nxt = tok[idx - 64];   // the block below is in flight while lane 0 works through this one
      const uint32_t cl = code.ctx_map[t.ctx];
      uint32_t sym, nb, bits;
      HybridD(t.value, &sym, &nb, &bits);
      // The recurrence runs on the scalar unit (every value in it is uniform): lane k keeps token k's frequency, reciprocal and slot
      // base in registers, the loop fetches them with v_readlane, divides by multiply-high with one fix-up and leaves the flush
      // in LDS - a dependent scalar operation costs less than half a dependent vector one on this machine, and the only
      // lookup left on the chain is the slot map.
      const uint32_t f = max((uint32_t)code.freq[cl * kEncSyms + sym], 1u);
      // floor(2^32 / f) (quotient estimate q or q - 1): exact in double, the fraction of 2^32 / f is 0 or at least 1 / 4096
      const uint32_t rcp = f == 1 ? 0xFFFFFFFFu : (uint32_t)(4294967296.0 / (double)f);
      const uint32_t rbase = cl * 4096 + code.start[cl * kEncSyms + sym];
      uint32_t fl = 0;
      {
        uint32_t s_state = (uint32_t)__builtin_amdgcn_readfirstlane((int)state);
        const int cnt = __builtin_amdgcn_readfirstlane((int)min<int64_t>(64, (int64_t)n - b * 64));
        for (int k = cnt - 1; k >= 0; k--) {
          const uint32_t fk = (uint32_t)__builtin_amdgcn_readlane((int)f, k);
          const uint32_t rk = (uint32_t)__builtin_amdgcn_readlane((int)rcp, k);
          const uint32_t bk = (uint32_t)__builtin_amdgcn_readlane((int)rbase, k);
          uint32_t flush = 0;
          if ((s_state >> 20) >= fk) { flush = 0x10000u | (s_state & 0xFFFF); s_state >>= 16; }
          uint32_t qd = __umulhi(s_state, rk);
          uint32_t rm = s_state - qd * fk;
          if (rm >= fk) { qd++; rm -= fk; }
          const uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(kLdsRmap ? lds_rmap[bk + rm] : code.rmap[(size_t)bk + rm]));
          s_state = (qd << 12) + slot;
          sc->flush[k] = flush;   // every lane stores the same word: off the chain, read back per lane below
        }
        state = s_state;
      }
      WaveSync();
      fl = sc->flush[lane];
      WaveSync();
      if (valid) tok[idx].ctx = fl;
    }
  }
  return (uint32_t)__shfl((int)state, 0);
}

// Forward pass: the state, then per token its flush and its raw bits, laid out by all lanes at once.
__device__ void ForwardPass(const DevToken* tok, uint32_t n, uint32_t state, WaveWriter& w) {
  const int lane = w.lane;
  const DevToken kNone = {0u, 0u};
  const int64_t nblk = ((int64_t)n + 63) >> 6;
  w.PutUniform(32, state);
  {
    DevToken nxt = kNone;
    if (nblk) nxt = (uint32_t)lane < n ? tok[lane] : kNone;
    for (int64_t b = 0; b < nblk; b++) {
      const int64_t idx = b * 64 + lane;
      const bool valid = idx < (int64_t)n;
      const DevToken t = nxt;
      if (b + 1 < nblk) nxt = idx + 64 < (int64_t)n ? tok[idx + 64] : kNone;
      uint32_t sym, nb, bits;
      HybridD(t.value, &sym, &nb, &bits);
      const uint32_t flush = t.ctx;
      const uint32_t len = valid ? (flush ? 16u : 0u) + nb : 0u;
      const uint64_t V = flush ? ((uint64_t)bits << 16) | (flush & 0xFFFF) : (uint64_t)bits;
      w.PutParallel(V, len);
    }
  }
}


template <bool kLdsRmap>
__device__ void EncodeStream(DevToken* tok, uint32_t n, const EncCodeDev& code, WaveWriter& w, WaveScratch* sc) {
  const uint32_t state = ReversePass<kLdsRmap>(tok, n, code, sc, w.lane);
  ForwardPass(tok, n, state, w);
}

__device__ size_t StageEncCode(uint8_t* smem, size_t off, const EncCodeDev& g, EncCodeDev* l, bool with_rmap, int tid, int nt) {
  off = (off + 15) & ~(size_t)15;
  const size_t nfs = (size_t)g.num_clusters * kEncSyms;
  uint16_t* f = (uint16_t*)(smem + off); off += nfs * 2;
  uint16_t* st = (uint16_t*)(smem + off); off += nfs * 2;
  uint16_t* rm = (uint16_t*)(smem + off);
  if (with_rmap) off += (size_t)g.num_clusters * 4096 * 2;
  uint8_t* m = smem + off; off += g.num_ctx;
  for (size_t i = tid; i < nfs; i += nt) { f[i] = g.freq[i]; st[i] = g.start[i]; }
  if (with_rmap) for (size_t i = tid; i < (size_t)g.num_clusters * 4096; i += nt) rm[i] = g.rmap[i];
  for (size_t i = tid; i < g.num_ctx; i += nt) m[i] = g.ctx_map[i];
  l->ctx_map = m; l->freq = f; l->start = st; l->rmap = with_rmap ? rm : g.rmap;
  l->num_clusters = g.num_clusters; l->num_ctx = g.num_ctx;
  return off;
}

}  // namespace

// sections [0, nlf): LF groups; [nlf, nlf + ng): pass groups.  One wavefront per section, kSectionsPerWg wavefronts per workgroup
// share the entropy-code tables staged in LDS.
constexpr int kSectionsPerWg = 4;
__device__ __forceinline__ WaveScratch* CarveScratch(uint8_t* smem, size_t off, int wave) {
  off = (off + 15) & ~(size_t)15;
  return (WaveScratch*)(smem + off) + wave;
}
// Token streams of a lossy frame, numbered so that the two streams of a section get their own wavefronts in the reverse pass:
// 2g / 2g+1: LF coefficients / block metadata of LF group g; 2 nlf + 2g / + 1: HF coefficients / alpha of group g.
__device__ __forceinline__ bool LossyStream(const EncImage& im, int st, DevToken** tok, uint32_t* n, bool* modular) {
  if (st < 2 * im.nlf) {
    const int g = st >> 1;
    const int gx = g % im.xlf, gy = g / im.xlf;
    const int bw = min(kLfGroupBlocks, im.w8 - gx * kLfGroupBlocks), bh = min(kLfGroupBlocks, im.h8 - gy * kLfGroupBlocks);
    *modular = true;
    if (st & 1) { *tok = im.tok_meta + (size_t)g * kMetaTokCap; *n = im.n_meta[g]; }
    else { *tok = im.tok_lf + (size_t)g * kLfTokCap; *n = (uint32_t)(3 * bw * bh); }
    return true;
  }
  const int r = st - 2 * im.nlf;
  const int g = r >> 1;
  if (g >= im.ng) return false;
  if (r & 1) {
    if (!(im.has_alpha && im.ng > 1)) return false;
    const int gx = g % im.xg, gy = g / im.xg;
    const int gw = min(kGroupDim, im.w - gx * kGroupDim), gh = min(kGroupDim, im.h - gy * kGroupDim);
    *modular = true; *tok = im.tok_alpha + (size_t)g * kAlphaTokCap; *n = (uint32_t)(gw * gh);
  } else {
    *modular = false; *tok = im.tok_ac + (size_t)g * kAcTokCap; *n = im.n_ac[g];
  }
  return true;
}
// Reverse (state) passes: one wavefront per token stream - the only serial part of the entropy coder, so the two streams of an LF
// group (196 k + 65 k tokens) and of a pass group run side by side instead of one after the other.
// `which` = 0: the streams coded with the Modular code (LF coefficients, block metadata, alpha); 1: the HF coefficient streams.  Two
// launches, because the Modular code is ready long before the (much larger) HF code: the host builds the latter while the longest
// recurrence of the frame - the LF coefficients of an LF group - is already running.
__global__ __launch_bounds__(64 * kSectionsPerWg) void enc_reverse_kernel(EncImage im, int which) {
  extern __shared__ __align__(16) uint8_t enc_smem[];
  size_t off;
  {
    EncCodeDev l;
    if (which == 0) { off = StageEncCode(enc_smem, 0, im.mcode, &l, true, threadIdx.x, blockDim.x); im.mcode = l; }
    else { off = StageEncCode(enc_smem, 0, im.acode, &l, false, threadIdx.x, blockDim.x); im.acode = l; }
    __syncthreads();
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  WaveScratch* sc = CarveScratch(enc_smem, off, wave);
  const int st = blockIdx.x * kSectionsPerWg + wave;
  DevToken* tok;
  uint32_t n;
  bool modular;
  if (!LossyStream(im, st, &tok, &n, &modular)) return;
  if (modular != (which == 0)) return;
  const uint32_t state = modular ? ReversePass<true>(tok, n, im.mcode, sc, lane) : ReversePass<false>(tok, n, im.acode, sc, lane);
  if (lane == 0) im.stream_state[st] = state;
}

// Bit layout of every section (after enc_reverse_kernel): header fields and, per stream, state + flushes + raw bits; no tables needed.
__global__ __launch_bounds__(64 * kSectionsPerWg) void enc_sections_kernel(EncImage im) {
  __shared__ WaveScratch s_sc[kSectionsPerWg];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  WaveScratch* sc = &s_sc[wave];
  const int s = blockIdx.x * kSectionsPerWg + wave;
  if (s >= im.nlf + im.ng) return;
  WaveWriter w;
  w.Init(im.sec_bytes + (size_t)s * im.sec_cap, sc, lane);
  DevToken* tok;
  uint32_t n;
  bool modular;
  if (s < im.nlf) {
    const int g = s;
    const int gx = g % im.xlf, gy = g / im.xlf;
    const int bw = min(kLfGroupBlocks, im.w8 - gx * kLfGroupBlocks), bh = min(kLfGroupBlocks, im.h8 - gy * kLfGroupBlocks);
    w.PutUniform(2, 0);   // extra_precision
    w.PutUniform(4, 3);   // modular group header: global tree, default weighted-predictor parameters, no transforms
    LossyStream(im, 2 * g, &tok, &n, &modular);
    ForwardPass(tok, n, im.stream_state[2 * g], w);
    const uint32_t nblocks = im.squares ? im.n_meta[g] / 2 : im.n_meta[g];
    w.PutUniform(CeilLog2E((uint32_t)(bw * bh)), nblocks - 1);   // number of varblocks - 1
    w.PutUniform(4, 3);
    LossyStream(im, 2 * g + 1, &tok, &n, &modular);
    ForwardPass(tok, n, im.stream_state[2 * g + 1], w);
  } else {
    const int g = s - im.nlf;
    const int st = 2 * im.nlf + 2 * g;
    LossyStream(im, st, &tok, &n, &modular);
    ForwardPass(tok, n, im.stream_state[st], w);
    if (LossyStream(im, st + 1, &tok, &n, &modular)) {
      w.PutUniform(4, 3);
      ForwardPass(tok, n, im.stream_state[st + 1], w);
    }
  }
  const uint64_t bits = w.Finish();
  if (lane == 0) im.sec_bits[s] = bits;
}

// single-group frames carry their alpha in the global Modular section: coded on its own (section index nlf + ng)
__global__ __launch_bounds__(64) void enc_global_alpha_kernel(EncImage im) {
  __shared__ WaveScratch sc;
  if (blockIdx.x) return;
  WaveWriter w;
  const int s = im.nlf + im.ng;
  w.Init(im.sec_bytes + (size_t)s * im.sec_cap, &sc, threadIdx.x);
  EncodeStream<false>(im.tok_alpha, (uint32_t)(im.w * im.h), im.mcode, w, &sc);
  const uint64_t bits = w.Finish();
  if (threadIdx.x == 0) im.sec_bits[s] = bits;
}

// gathers the used bytes of every section into one contiguous buffer
__global__ void enc_compact_kernel(EncImage im, const uint64_t* dst_off, uint8_t* dst, int nsec) {
  const int s = blockIdx.y;
  if (s >= nsec) return;
  const uint64_t nbytes = (im.sec_bits[s] + 7) >> 3;
  const uint8_t* src = im.sec_bytes + (size_t)s * im.sec_cap;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += (uint64_t)gridDim.x * blockDim.x) dst[dst_off[s] + i] = src[i];
}

// ------------------------------------------------------------------ lossless (Modular) frames
// BGRA8 -> integer channel planes: Gray(A) takes the B channel (PixelFormatConversion.cpp:34,60); RGB goes through the
// reversible YCoCg-R transform (RCT type 6), alpha is the last channel.
__global__ void enc_ll_planes_kernel(EncImage im) {
  const size_t n = (size_t)im.w * im.h;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % im.w), y = (int)(i / im.w);
    const uchar4 p = *(const uchar4*)(im.bgra + (size_t)y * im.stride + (size_t)x * 4);   // B, G, R, A
    int c = 0;
    if (im.gray) im.ll_plane[c++][i] = p.x;
    else {
      const int32_t R = p.z, G = p.y, B = p.x;
      const int32_t co = R - B, tmp = B + (co >> 1), cg = G - tmp, yy = tmp + (cg >> 1);
      im.ll_plane[0][i] = yy; im.ll_plane[1][i] = co; im.ll_plane[2][i] = cg;
      c = 3;
    }
    if (im.has_alpha) im.ll_plane[c][i] = p.w;
  }
}

// gradient-predicted residuals of every channel of one group; context = channel (leaf id 3 - channel)
__global__ __launch_bounds__(256) void enc_ll_tokens_kernel(EncImage im) {
  __shared__ uint32_t s_h[4 * kEncSyms];
  for (int i = threadIdx.x; i < 4 * (int)kEncSyms; i += 256) s_h[i] = 0;
  __syncthreads();
  const int g = blockIdx.y;
  const int gx = g % im.xg, gy = g / im.xg;
  const int x0 = gx * kGroupDim, y0 = gy * kGroupDim;
  const int gw = min(kGroupDim, im.w - x0), gh = min(kGroupDim, im.h - y0);
  const int per = gw * gh, n = per * im.ll_nch;
  DevToken* out = im.tok_ll + (size_t)g * kLlTokCap;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int c = i / per, r = i % per, y = r / gw, x = r % gw;
    const int32_t* pl = im.ll_plane[c] + (size_t)y0 * im.w + x0;
    const int32_t v = pl[(size_t)y * im.w + x];
    int32_t W, N, NW;
    if (x == 0) { W = y ? pl[(size_t)(y - 1) * im.w] : 0; N = W; NW = W; }
    else {
      W = pl[(size_t)y * im.w + x - 1];
      N = y ? pl[(size_t)(y - 1) * im.w + x] : W;
      NW = y ? pl[(size_t)(y - 1) * im.w + x - 1] : W;
    }
    DevToken t;
    t.ctx = 3 - c;
    t.value = PackSignedD(v - GradientPred(W, N, NW));
    out[i] = t;
    uint32_t tok, nb, bits;
    HybridD(t.value, &tok, &nb, &bits);
    atomicAdd(&s_h[t.ctx * kEncSyms + tok], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * (int)kEncSyms; i += 256)
    if (s_h[i]) atomicAdd(&im.hist_mod[i], s_h[i]);
}

__global__ __launch_bounds__(64 * kSectionsPerWg) void enc_ll_sections_kernel(EncImage im) {
  extern __shared__ __align__(16) uint8_t enc_smem[];
  size_t off;
  {
    EncCodeDev lm;
    off = StageEncCode(enc_smem, 0, im.mcode, &lm, true, threadIdx.x, blockDim.x);
    im.mcode = lm;
    __syncthreads();
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  WaveScratch* sc = CarveScratch(enc_smem, off, wave);
  const int g = blockIdx.x * kSectionsPerWg + wave;
  if (g >= im.ng) return;
  const int gx = g % im.xg, gy = g / im.xg;
  const int gw = min(kGroupDim, im.w - gx * kGroupDim), gh = min(kGroupDim, im.h - gy * kGroupDim);
  WaveWriter w;
  w.Init(im.sec_bytes + (size_t)g * im.sec_cap, sc, lane);
  if (im.ng > 1) w.PutUniform(4, 3);   // group header; a single-group frame continues the GlobalModular stream of LfGlobal
  EncodeStream<true>(im.tok_ll + (size_t)g * kLlTokCap, (uint32_t)(gw * gh * im.ll_nch), im.mcode, w, sc);
  const uint64_t bits = w.Finish();
  if (lane == 0) im.sec_bits[g] = bits;
}

// ------------------------------------------------------------------ launch wrappers
static inline unsigned GridFor(size_t work, unsigned cap = 8192) {
  size_t b = (work + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

void LaunchEncAnalyze(const EncImage& im, hipStream_t s) {
  hipLaunchKernelGGL(enc_analyze_kernel, dim3(GridFor((size_t)im.w * im.h, 2048)), dim3(256), 0, s, im);
}
void LaunchEncFrontEnd(const EncImage& im, hipStream_t s) {
  hipLaunchKernelGGL(enc_xyb_kernel, dim3(GridFor((size_t)im.w * im.h)), dim3(256), 0, s, im);
  hipLaunchKernelGGL(enc_sharpen_pad_kernel, dim3(GridFor((size_t)im.wp * im.hp)), dim3(256), 0, s, im);
  hipLaunchKernelGGL(enc_activity_kernel, dim3(GridFor((size_t)im.w8 * im.h8)), dim3(256), 0, s, im);
  const unsigned regions = (unsigned)(((im.w8 + 7) / 8) * ((im.h8 + 7) / 8));
  hipLaunchKernelGGL(enc_strategy_kernel, dim3(GridFor(regions)), dim3(256), 0, s, im);
  hipLaunchKernelGGL(enc_varblock_kernel, dim3(regions), dim3(256), 0, s, im);
}
void LaunchEncTokens(const EncImage& im, hipStream_t s) {
  hipLaunchKernelGGL(enc_lf_tokens_kernel, dim3(64, im.nlf), dim3(256), 0, s, im);
  hipLaunchKernelGGL(enc_meta_tokens_kernel, dim3(im.nlf), dim3(1024), 0, s, im);
  hipLaunchKernelGGL(enc_ac_tokens_kernel, dim3(im.ng), dim3(256), 0, s, im);
  if (im.has_alpha) hipLaunchKernelGGL(enc_alpha_tokens_kernel, dim3(32, im.ng), dim3(256), 0, s, im);
}
static size_t EncCodeLds(const EncCodeDev& c, bool with_rmap) {   // as StageEncCode carves it
  return 16 + (size_t)c.num_clusters * (kEncSyms * 4 + (with_rmap ? 8192 : 0)) + c.num_ctx;
}
void LaunchEncReverse(const EncImage& im, int which, hipStream_t s) {
  // reverse passes: LDS = the code of this launch (the Modular one with its slot map) + one scratch block per wavefront
  const size_t lds = (which == 0 ? EncCodeLds(im.mcode, true) : EncCodeLds(im.acode, false)) + 32 + kSectionsPerWg * sizeof(WaveScratch);
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)enc_reverse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int nstreams = 2 * (im.nlf + im.ng);
  hipLaunchKernelGGL(enc_reverse_kernel, dim3((unsigned)((nstreams + kSectionsPerWg - 1) / kSectionsPerWg)), dim3(64 * kSectionsPerWg), lds, s, im, which);
}
void LaunchEncSections(const EncImage& im, hipStream_t s) {
  hipLaunchKernelGGL(enc_sections_kernel, dim3((unsigned)((im.nlf + im.ng + kSectionsPerWg - 1) / kSectionsPerWg)), dim3(64 * kSectionsPerWg), 0, s, im);
  if (im.has_alpha && im.ng == 1) hipLaunchKernelGGL(enc_global_alpha_kernel, dim3(1), dim3(64), 0, s, im);
}
void LaunchEncLossless(const EncImage& im, int stage, hipStream_t s) {
  if (stage == 0) {
    hipLaunchKernelGGL(enc_ll_planes_kernel, dim3(GridFor((size_t)im.w * im.h)), dim3(256), 0, s, im);
    hipLaunchKernelGGL(enc_ll_tokens_kernel, dim3(64, im.ng), dim3(256), 0, s, im);
  } else {
    const size_t lds = EncCodeLds(im.mcode, true) + 16 + kSectionsPerWg * sizeof(WaveScratch);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)enc_ll_sections_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(enc_ll_sections_kernel, dim3((unsigned)((im.ng + kSectionsPerWg - 1) / kSectionsPerWg)), dim3(64 * kSectionsPerWg), lds, s, im);
  }
}
void LaunchEncCompact(const EncImage& im, const uint64_t* dst_off, uint8_t* dst, int nsec, hipStream_t s) {
  hipLaunchKernelGGL(enc_compact_kernel, dim3(64, nsec), dim3(256), 0, s, im, dst_off, dst, nsec);
}

}  // namespace jxlhip
