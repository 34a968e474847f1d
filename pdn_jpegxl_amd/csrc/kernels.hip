// HIP kernels (gfx950) of the JPEG XL VarDCT decode path.
//
// Pixel-domain stages (the entropy-coded stages live in entropy_kernels.hip):
//   lf_dequant / lf_smooth / cell_sigma / alpha_to_u8 / dequant / llf / idct_v / idct_h / idct_special
//   gaborish / epf<stage> / xyb_to_out
// DESIGN.md has the data layout and the per-kernel roofline.
#include <hip/hip_runtime.h>
#include "dev_types.h"
#include "kernels.h"

namespace jxlhip {

__constant__ uint8_t c_quant_table[kNumStrategies] = {0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16};
__device__ __forceinline__ bool IsSpecial(uint32_t s) { return (s >= 1 && s <= 3) || (s >= 12 && s <= 17); }
__device__ __forceinline__ int DMirror(int v, int n) {
  while (v < 0 || v >= n) v = v < 0 ? -v - 1 : 2 * n - 1 - v;
  return v;
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
