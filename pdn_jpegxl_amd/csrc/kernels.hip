// HIP kernels (gfx950) of the JPEG XL VarDCT decode path.
//
// Pixel-domain stages (the entropy-coded stages live in entropy_kernels.hip):
//   lf_dequant / lf_smooth / cell_sigma / dequant / llf / idct_v / idct_h / idct_special
//   gaborish / epf<stage> / xyb_to_out
// DESIGN.md has the data layout and the per-kernel roofline.
#include <hip/hip_runtime.h>
#include <algorithm>
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
__global__ void lf_dequant_kernel(const DevImage* __restrict__ imgs) {
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

__global__ void lf_smooth_kernel(const DevImage* __restrict__ imgs) {
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

__global__ void cell_sigma_kernel(const DevImage* __restrict__ imgs) {
  const DevImage& im = imgs[blockIdx.y];
  const int n = im.w8 * im.h8;
  const float kInvSigmaNum = -1.1715728752538099024f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float sigma_quant = im.epf_quant_mul / (im.quant_scale * (float)im.rawq[i] * kInvSigmaNum);
    float sigma = sigma_quant * im.epf_sharp_lut[im.sharp[i] & 7];   // & 7: cells outside a decoded band hold no sharpness yet
    sigma = fminf(-1e-4f, sigma);
    im.inv_sigma[i] = 1.0f / sigma;
  }
}

// ------------------------------------------------------------------ generic (any block size) reconstruction
// These kernels only visit the 64x64 tiles that recon_tile_kernel could not handle (tiles touched by a varblock
// larger than the tile); the list and its length (status[1]) are written by that kernel.
#define FOR_LISTED_TILES(im, tile)                                   \
  const uint32_t n_listed_ = (im).status ? (im).status[1] : 0u;       \
  for (uint32_t li_ = blockIdx.x; li_ < n_listed_; li_ += gridDim.x) \
    for (int tile = (int)(im).tile_list[li_], once_ = 1; once_; once_ = 0)

// The tiles a kernel of the generic path visits: the listed ones, or (all != 0: debug taps, multi-pass frames) every tile of the
// decoded band.
#define FOR_TILES(im, tile, all)                                                                    \
  const uint32_t n_tiles_ = (all) ? (uint32_t)((im).wt * (im).ht) : ((im).status ? (im).status[1] : 0u); \
  for (uint32_t li_ = blockIdx.x; li_ < n_tiles_; li_ += gridDim.x)                                 \
    for (int tile = (all) ? (int)li_ : (int)(im).tile_list[li_], once_ = 1; once_; once_ = 0)

// Dense int32 coefficient planes (footprint layout: coefficient (ky, kx) of a varblock at pixel (y0 + ky, x0 + kx)) for the tiles the
// generic kernels handle, from the sparse entry lists hf_decode_kernel wrote: first zeros, then (a second launch: a varblock
// larger than a tile reaches into other listed tiles) every entry of the varblocks that START in the tile.
__global__ void expand_zero_kernel(const DevImage* __restrict__ imgs, int all) {
  const DevImage& im = imgs[blockIdx.y];
  if (im.is_modular) return;
  FOR_TILES(im, tile, all) {
    const int tx = tile % im.wt, ty = tile / im.wt;
    if (ty < im.dec_gy0 * 4 || ty >= im.dec_gy1 * 4) continue;
    for (int e = threadIdx.x; e < 4096; e += blockDim.x) {
      const int x = tx * 64 + (e & 63), y = ty * 64 + (e >> 6);
      if (x >= im.wp || y >= im.hp) continue;
      const size_t i = (size_t)y * im.wp + x;
      im.coef[0][i] = 0; im.coef[1][i] = 0; im.coef[2][i] = 0;
    }
  }
}
__global__ void expand_scatter_kernel(const DevImage* __restrict__ imgs, int all) {
  const DevImage& im = imgs[blockIdx.y];
  if (im.is_modular) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
  const size_t ncells = (size_t)im.w8 * im.h8;
  FOR_TILES(im, tile, all) {
    const int tx = tile % im.wt, ty = tile / im.wt;
    if (ty < im.dec_gy0 * 4 || ty >= im.dec_gy1 * 4) continue;
    for (int pair = wave; pair < 192; pair += nwaves) {   // (cell, channel) pairs of the tile, one wavefront each
      const int cell = pair & 63, c = pair >> 6;
      const int bx = tx * 8 + (cell & 7), by = ty * 8 + (cell >> 3);
      if (bx >= im.w8 || by >= im.h8) continue;
      const uint32_t info = im.cellinfo[(size_t)by * im.w8 + bx];
      if (!(info >> 31) || (info & 0x3FF00u)) continue;   // not the origin cell of a varblock
      const uint32_t s = info & 0xFF, lcx = (info >> 18) & 7, lcy = (info >> 21) & 7;
      const uint32_t q = c_quant_table[s], nq = im.dq_n[q];
      const uint32_t lng = 3 + max(lcx, lcy);
      const bool transposed = !IsSpecial(s) && lcy >= lcx;
      int32_t* plane = im.coef[c] + (size_t)by * 8 * im.wp + (size_t)bx * 8;
      // every pass of a progressive frame adds its share (value << shift); each has its own lists, orders and scan lists
      for (const DevImage* ps = &im; ps; ps = ps->next_pass) {
        const uint32_t* entries = ps->centries + (size_t)((ty >> 2) * im.xg + (tx >> 2) - ps->centries_g0) * kGroupEntriesCap;
        const U32x2 blk = ps->cblk[(size_t)c * ncells + (size_t)by * im.w8 + bx];
        if (blk.y > 65536u || blk.x > kGroupEntriesCap - blk.y) continue;   // left unwritten by a failed section
        const U32x2* scan = ps->scan[q] + (size_t)c * nq;
        const int shift = ps->pass_shift;
        const bool accumulate = im.num_passes > 1;
        for (uint32_t i = lane; i < blk.y; i += 64) {
          const uint32_t ent = entries[blk.x + i], k = ent & 0xFFFFu;
          if (k >= nq) continue;
          const uint32_t p = scan[k].x;
          const uint32_t r = p >> lng, cc = p & ((1u << lng) - 1);
          const uint32_t ky = transposed ? cc : r, kx = transposed ? r : cc;
          if (ky < (8u << lcy) && kx < (8u << lcx)) {
            const int32_t v = ((int32_t)ent >> 16) * (1 << shift);
            if (accumulate) atomicAdd(&plane[(size_t)ky * im.wp + kx], v);   // lanes of different passes may meet at one position
            else plane[(size_t)ky * im.wp + kx] = v;
          }
        }
      }
    }
  }
}

__global__ void dequant_kernel(const DevImage* __restrict__ imgs) {
  const DevImage& im = imgs[blockIdx.y];
  FOR_LISTED_TILES(im, tile) {
    const int tx = tile % im.wt, ty = tile / im.wt;
    for (int e = threadIdx.x; e < 4096; e += blockDim.x) {
      const int x = tx * 64 + (e & 63), y = ty * 64 + (e >> 6);
      if (x >= im.wp || y >= im.hp) continue;
      const size_t i = (size_t)y * im.wp + x;
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
      const size_t tcfl = (size_t)(oby >> 3) * im.wt + (obx >> 3);
      const float cfx = im.base_x + (float)im.ytox[tcfl] * im.inv_color_factor;
      const float cfb = im.base_b + (float)im.ytob[tcfl] * im.inv_color_factor;
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
      ((float*)im.coef[0])[i] = out[0];
      ((float*)im.coef[1])[i] = out[1];
      ((float*)im.coef[2])[i] = out[2];
    }
  }
}

// LLF: the lowest cx*cy coefficients of each varblock from the (smoothed) LF image; one thread per cell.
__global__ void llf_kernel(const DevImage* __restrict__ imgs, const float* basis_small, const float* llf_scale) {
  const DevImage& im = imgs[blockIdx.y];
  FOR_LISTED_TILES(im, tile) {
    const int tx = tile % im.wt, ty = tile / im.wt;
    for (int e = threadIdx.x; e < 64; e += blockDim.x) {
      const int bx = tx * 8 + (e & 7), by = ty * 8 + (e >> 3);
      if (bx >= im.w8 || by >= im.h8) continue;
      const uint32_t info = im.cellinfo[(size_t)by * im.w8 + bx];
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
}

// A 64x64 tile that lies inside ONE varblock of at least 64 points both ways (varblocks are aligned to their size, so it is enough
// to look at the tile's first cell) takes the matrix-core kernel below; everything else the vector-ALU kernels.
__device__ __forceinline__ bool GemmTile(const DevImage& im, int tx, int ty) {
  const uint32_t info = im.cellinfo[(size_t)ty * 8 * im.w8 + tx * 8];
  return (info & 0xFF) >= 18 && (info & 0xFF) <= 26 && ((info >> 18) & 7) >= 3 && ((info >> 21) & 7) >= 3;
}

// Both 1-D IDCT passes of such a tile as 64 x R x 64 matrix products on the matrix cores (v_mfma_f32_16x16x4_f32: exact f32
// products, k-ordered accumulation), R = 64 / 128 / 256: the 128- and 256-point transforms are the largest matrix products of the
// codec.  Per step of 16 along k, the workgroup stages a 64 x 16 slice of the left operand and a 16 x 64 slice of the right one in
// LDS (one of them is a slice of the basis, the other a slice of the coefficient / intermediate plane); wavefront w owns output
// rows 16 w .. 16 w + 15, four 16 x 16 accumulators.  pass 0: tmp = Basis_R^T * coefficients (columns); pass 1: xyb = tmp * Basis_C.
typedef float __attribute__((ext_vector_type(4))) GF4;
__global__ __launch_bounds__(256) void idct_gemm_kernel(const DevImage* __restrict__ imgs, const float* basis_all, int pass) {
  __shared__ float s_a[64 * 17];   // A[m][k], pitch 17
  __shared__ float s_b[16 * 65];   // B[k][n], pitch 65
  const DevImage& im = imgs[blockIdx.y];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, lq = lane >> 4;
  FOR_LISTED_TILES(im, tile) {
    const int tx = tile % im.wt, ty = tile / im.wt;
    if (!GemmTile(im, tx, ty)) continue;
    const uint32_t info = im.cellinfo[(size_t)ty * 8 * im.w8 + tx * 8];
    const int ix = (info >> 8) & 31, iy = (info >> 13) & 31, lcx = (info >> 18) & 7, lcy = (info >> 21) & 7;
    const int K = pass == 0 ? 8 << lcy : 8 << lcx;           // length of the transform along the contracted dimension
    const int off = pass == 0 ? iy * 8 : ix * 8;              // the tile's first output row / column inside the varblock
    const float* B = basis_all + ((size_t)K * K - 64) / 3;   // B[k * K + n]
    const size_t y0 = (size_t)ty * 64, x0 = (size_t)tx * 64;
    for (int c = 0; c < 3; c++) {
      const float* in = pass == 0 ? (const float*)im.coef[c] : im.tmp[c];
      float* out = pass == 0 ? im.tmp[c] : im.xyb[c];
      GF4 acc[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) acc[s4] = GF4{0.f, 0.f, 0.f, 0.f};
      for (int k0 = 0; k0 < K; k0 += 16) {
        __syncthreads();
        // stage: 1024 elements of each operand, four per thread
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int e = tid + j * 256;
          if (pass == 0) {
            // A[m][k] = Basis[k][off + m] (m fastest in memory);  B[k][n] = coefficient row (block row k) x column n of the tile
            const int m = e & 63, k = e >> 6;
            s_a[m * 17 + k] = B[(size_t)(k0 + k) * K + off + m];
            const int n = e & 63;
            s_b[k * 65 + n] = in[(y0 - (size_t)iy * 8 + k0 + k) * im.wp + x0 + n];
          } else {
            // A[m][k] = intermediate row m of the tile x block column k;  B[k][n] = Basis[k][off + n]
            const int k = e & 15, m = e >> 4;
            s_a[m * 17 + k] = in[(y0 + m) * im.wp + x0 - (size_t)ix * 8 + k0 + k];
            const int n = e & 63, kk = e >> 6;
            s_b[kk * 65 + n] = B[(size_t)(k0 + kk) * K + off + n];
          }
        }
        __syncthreads();
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) {
          const float a = s_a[(wave * 16 + l16) * 17 + k4 * 4 + lq];
#pragma unroll
          for (int s4 = 0; s4 < 4; s4++) acc[s4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, s_b[(k4 * 4 + lq) * 65 + s4 * 16 + l16], acc[s4], 0, 0, 0);
        }
      }
      // lane owns D[4 * lq + r][l16] of each 16 x 16 accumulator
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) {
        float* o = out + (y0 + wave * 16 + 4 * lq) * im.wp + x0 + s4 * 16 + l16;
        o[0] = acc[s4].x; o[im.wp] = acc[s4].y; o[2 * (size_t)im.wp] = acc[s4].z; o[3 * (size_t)im.wp] = acc[s4].w;
      }
    }
  }
}

// Vertical 1-D IDCT: thread = (column x, 8-row cell), all three channels.
__global__ void idct_v_kernel(const DevImage* __restrict__ imgs, const float* basis_all) {
  const DevImage& im = imgs[blockIdx.y];
  FOR_LISTED_TILES(im, tile) {
    const int tx = tile % im.wt, ty = tile / im.wt;
    if (GemmTile(im, tx, ty)) continue;
    for (int e = threadIdx.x; e < 512; e += blockDim.x) {
      const int x = tx * 64 + (e & 63), by = ty * 8 + (e >> 6);
      if (x >= im.wp || by >= im.h8) continue;
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
}

// Horizontal 1-D IDCT: thread = (row y, 8-column cell).
__global__ void idct_h_kernel(const DevImage* __restrict__ imgs, const float* basis_all) {
  const DevImage& im = imgs[blockIdx.y];
  FOR_LISTED_TILES(im, tile) {
    const int tx = tile % im.wt, ty = tile / im.wt;
    if (GemmTile(im, tx, ty)) continue;
    for (int e = threadIdx.x; e < 512; e += blockDim.x) {
      const int bx = tx * 8 + (e >> 6), y = ty * 64 + (e & 63);
      if (bx >= im.w8 || y >= im.hp) continue;
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
__global__ void idct_special_kernel(const DevImage* __restrict__ imgs, const float* basis_all, const float* basis_small) {
  const DevImage& im = imgs[blockIdx.y];
  const float* B8 = basis_all;                 // N = 8
  const float* B4 = basis_small + (16 - 1) / 3;  // c = 4
  FOR_LISTED_TILES(im, tile)
  for (int e = threadIdx.x; e < 192; e += blockDim.x) {
    const int c = e / 64;
    const int bx = (tile % im.wt) * 8 + (e & 7), by = (tile / im.wt) * 8 + ((e >> 3) & 7);
    if (bx >= im.w8 || by >= im.h8) continue;
    const uint32_t s = im.cellinfo[(size_t)by * im.w8 + bx] & 0xFF;
    if (!IsSpecial(s)) continue;
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
      for (int k = 0; k < 64; k++) px[k] = 0.f;   // unreachable: lf_finish_kernel refuses AFV ids (kErrUnsupportedTransform) and places no block of such a group
    }
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) dst[(size_t)y * im.wp + x] = px[y * 8 + x];
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

void LaunchExpandCoefficients(const DevImage* imgs, int nimg, bool all_tiles, int max_tiles, hipStream_t s) {
  const dim3 g(all_tiles ? (unsigned)std::max(1, std::min(max_tiles, 8192)) : 128u, nimg);
  hipLaunchKernelGGL(expand_zero_kernel, g, dim3(256), 0, s, imgs, all_tiles ? 1 : 0);
  hipLaunchKernelGGL(expand_scatter_kernel, g, dim3(256), 0, s, imgs, all_tiles ? 1 : 0);
}

void LaunchGenericReconstruct(const DevImage* imgs, int nimg, const float* basis_all, const float* basis_small,
                              const float* llf_scale, hipStream_t s) {
  const dim3 g(128, nimg);
  hipLaunchKernelGGL(dequant_kernel, g, dim3(256), 0, s, imgs);
  hipLaunchKernelGGL(llf_kernel, g, dim3(64), 0, s, imgs, basis_small, llf_scale);
  hipLaunchKernelGGL(idct_v_kernel, g, dim3(256), 0, s, imgs, basis_all);
  hipLaunchKernelGGL(idct_gemm_kernel, g, dim3(256), 0, s, imgs, basis_all, 0);
  hipLaunchKernelGGL(idct_h_kernel, g, dim3(256), 0, s, imgs, basis_all);
  hipLaunchKernelGGL(idct_gemm_kernel, g, dim3(256), 0, s, imgs, basis_all, 1);
  hipLaunchKernelGGL(idct_special_kernel, g, dim3(192), 0, s, imgs, basis_all, basis_small);
}

// ------------------------------------------------------------------ orientation (rare: one pixel per thread, byte copies)
__global__ void orient_kernel(const uint8_t* src, uint8_t* dst, int w, int h, int pb, int o) {
  const int ow = o >= 5 ? h : w, oh = o >= 5 ? w : h;
  const size_t n = (size_t)ow * oh;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % ow), oy = (int)(i / ow);
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
    const uint8_t* s = src + ((size_t)sy * w + sx) * pb;
    uint8_t* d = dst + i * pb;
    for (int k = 0; k < pb; k++) d[k] = s[k];
  }
}
void LaunchOrient(const uint8_t* src, uint8_t* dst, int w, int h, int px_bytes, int orientation, hipStream_t s) {
  const size_t n = (size_t)w * h;
  const unsigned blocks = (unsigned)std::min<size_t>(8192, (n + 255) / 256);
  hipLaunchKernelGGL(orient_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, src, dst, w, h, px_bytes, orientation);
}

// ------------------------------------------------------------------ status words back to the host
// The batch's status words go to pinned host memory by a kernel's stores, not by hipMemcpyAsync: one copy of n * 64 bytes takes the
// DMA engine, whose single queue then orders this batch's last command before the NEXT batch's upload on another stream (measured:
// 'upload + clear' 0.5 -> 36 ms, the chains no longer overlap), and a copy per image was 384 five-microsecond blit kernels.
__global__ void status_to_host_kernel(const uint32_t* src, uint32_t* dst, int nwords) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nwords) dst[i] = src[i];
}
void LaunchStatusToHost(const uint32_t* src, uint32_t* dst_pinned, int nwords, hipStream_t s) {
  if (nwords <= 0) return;
  hipLaunchKernelGGL(status_to_host_kernel, dim3((nwords + 255) / 256), dim3(256), 0, s, src, dst_pinned, nwords);
}

#ifdef JXLHIP_EXPERIMENTS
// ------------------------------------------------------------------ interference probes (experiments build only)
// A kernel that occupies ONE kind of resource for a given time while a decode stage runs beside it on another stream: which resource a
// stage is sensitive to.  kind 1: dependent vector chain; 2: independent vector instructions (issue throughput); 3: LDS
// traffic (read / write a 4 KB array); 4: LDS capacity (the launch's dynamic LDS, the wavefronts sleep); 5: registers (256 VGPRs per
// wavefront, sleeping); 6: L2 / HBM traffic (streaming copy).  One wavefront per SIMD per workgroup.
template <int kKind>
__global__ __launch_bounds__(256) void interfere_kernel(float* sink, uint64_t ticks, const float4* src, float4* dst, size_t n4) {
  extern __shared__ __align__(16) float lds[];
  const uint64_t t0 = wall_clock64();
  float a = (float)threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f;
  if (kKind == 3) for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = (float)i;
  if (kKind == 5) {
    float r[250];   // kept alive across the sleep loop
#pragma unroll
    for (int i = 0; i < 250; i++) r[i] = a + (float)i;
    while (wall_clock64() - t0 < ticks) {
      __builtin_amdgcn_s_sleep(64);
#pragma unroll
      for (int i = 0; i < 250; i++) asm volatile("" : "+v"(r[i]));
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 250; i++) sum += r[i];
    if (sum == 12345.678f) sink[0] = sum;
    return;
  }
  size_t pos = (size_t)blockIdx.x * 256 + threadIdx.x;
  while (wall_clock64() - t0 < ticks) {
    if (kKind == 1) {
#pragma unroll
      for (int i = 0; i < 64; i++) a = __builtin_fmaf(a, b, c);
    } else if (kKind == 2) {
#pragma unroll
      for (int i = 0; i < 16; i++) { a = __builtin_fmaf(a, b, c); b = __builtin_fmaf(b, 1.0f, 1e-9f); c = __builtin_fmaf(c, 0.999f, d); d = __builtin_fmaf(d, 1.0f, 1e-9f); }
    } else if (kKind == 3) {
#pragma unroll
      for (int i = 0; i < 16; i++) { const int j = (threadIdx.x * 17 + i * 64) & 1023; a += lds[j]; lds[(j + 256) & 1023] = a; }
    } else if (kKind == 4) {
      __builtin_amdgcn_s_sleep(64);
    } else if (kKind == 6) {
#pragma unroll
      for (int i = 0; i < 4; i++) { dst[pos % n4] = src[pos % n4]; pos += (size_t)gridDim.x * 256; }
    }
  }
  if (a + b + c + d == 12345.678f) sink[0] = a;
}
void LaunchInterference(int kind, int wg_per_cu, int lds_kb, float ms, float* scratch, size_t scratch_bytes, hipStream_t s) {
  const uint64_t ticks = (uint64_t)(ms * 1e-3 * 100e6);   // wall_clock64 runs at 100 MHz
  const dim3 g(256 * wg_per_cu), b(256);
  const size_t lds = kind == 4 ? (size_t)lds_kb * 1024 : (kind == 3 ? 4096 : 0);
  const size_t n4 = scratch_bytes / 32;
  const float4* src = (const float4*)scratch;
  float4* dst = (float4*)scratch + n4;
#define JXL_IK(K) case K: if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)interfere_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                          hipLaunchKernelGGL(interfere_kernel<K>, g, b, lds, s, scratch, ticks, src, dst, n4); break;
  switch (kind) { JXL_IK(1) JXL_IK(2) JXL_IK(3) JXL_IK(4) JXL_IK(5) JXL_IK(6) default: break; }
#undef JXL_IK
}
#endif

}  // namespace jxlhip
