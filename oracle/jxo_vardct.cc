// ORACLE — test infrastructure only (see jxo_common.h header).
#include "jxo_vardct.h"
#include "jxo_entropy.h"
#include <map>
#include <mutex>

namespace jxo {

const uint8_t kCoveredX[kNumStrategies] = {1, 1, 1, 1, 2, 4, 1, 2, 1, 4, 2, 4, 1, 1, 1, 1, 1, 1, 8, 4, 8, 16, 8, 16, 32, 16, 32};
const uint8_t kCoveredY[kNumStrategies] = {1, 1, 1, 1, 2, 4, 2, 1, 4, 1, 4, 2, 1, 1, 1, 1, 1, 1, 8, 8, 4, 16, 16, 8, 32, 32, 16};
const uint8_t kStrategyOrder[kNumStrategies] = {0, 1, 1, 1, 2, 3, 4, 4, 5, 5, 6, 6, 1, 1, 1, 1, 1, 1, 7, 8, 8, 9, 10, 10, 11, 12, 12};
const uint8_t kStrategyQuantTable[kNumStrategies] = {0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16};

const uint16_t kCoeffFreqContext[64] = {
    0xBAD, 0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 15, 16, 16, 17, 17, 18, 18, 19, 19, 20, 20, 21, 21, 22, 22,
    23,    23, 23, 23, 24, 24, 24, 24, 25, 25, 25, 25, 26, 26, 26, 26, 27, 27, 27, 27, 28, 28, 28, 28, 29, 29, 29, 29, 30, 30, 30, 30};
const uint16_t kCoeffNumNonzeroContext[64] = {
    0xBAD, 0,   31,  62,  62,  93,  93,  93,  93,  123, 123, 123, 123, 152, 152, 152, 152, 152, 152, 152, 152, 180,
    180,   180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206,
    206,   206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206};

// ------------------------------------------------------------------ natural coefficient order
static std::vector<uint32_t> ComputeNaturalOrder(int strategy) {
  size_t cx = kCoveredX[strategy], cy = kCoveredY[strategy];
  if (cy > cx) std::swap(cx, cy);  // stored layout: cx >= cy
  std::vector<uint32_t> out(cx * cy * 64);
  size_t xs = cx / cy, xsm = xs - 1, xss = CeilLog2(xs);
  size_t cur = cx * cy;
  for (size_t i = 0; i < cx * 8; i++) {
    for (size_t j = 0; j <= i; j++) {
      size_t x = j, y = i - j;
      if (i % 2) std::swap(x, y);
      if ((y & xsm) != 0) continue;
      y >>= xss;
      size_t val = (x < cx && y < cy) ? y * cx + x : cur++;
      out[val] = (uint32_t)(y * cx * 8 + x);
    }
  }
  for (size_t ip = cx * 8 - 1; ip > 0; ip--) {
    size_t i = ip - 1;
    for (size_t j = 0; j <= i; j++) {
      size_t x = cx * 8 - 1 - (i - j), y = cx * 8 - 1 - j;
      if (i % 2) std::swap(x, y);
      if ((y & xsm) != 0) continue;
      y >>= xss;
      out[cur++] = (uint32_t)(y * cx * 8 + x);
    }
  }
  JXO_CHECK(cur == out.size(), "natural order size");
  return out;
}

const std::vector<uint32_t>& NaturalOrder(int strategy) {
  static std::vector<uint32_t> cache[kNumStrategies];
  static std::once_flag once;
  std::call_once(once, [] { for (int s = 0; s < kNumStrategies; s++) cache[s] = ComputeNaturalOrder(s); });
  return cache[strategy];
}

// ------------------------------------------------------------------ DCT
static const std::vector<float>& Basis(int N) {  // B[k*N+n] = s_k cos((2n+1) k pi / 2N)
  static std::map<int, std::vector<float>> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> g(mu);
  auto it = cache.find(N);
  if (it != cache.end()) return it->second;
  std::vector<float> b((size_t)N * N);
  for (int k = 0; k < N; k++)
    for (int n = 0; n < N; n++)
      b[(size_t)k * N + n] = (float)((k ? std::sqrt(2.0) : 1.0) * std::cos((2 * n + 1) * k * M_PI / (2.0 * N)));
  return cache[N] = b;
}

void IdctStored(int R, int C, const float* stored, float* out, int stride) {
  const std::vector<float>& BR = Basis(R);
  const std::vector<float>& BC = Basis(C);
  const bool transposed = R >= C;
  std::vector<float> coef((size_t)R * C), tmp((size_t)R * C);
  // coef[ky*C + kx]
  if (transposed) {
    for (int kx = 0; kx < C; kx++) for (int ky = 0; ky < R; ky++) coef[(size_t)ky * C + kx] = stored[(size_t)kx * R + ky];
  } else {
    memcpy(coef.data(), stored, sizeof(float) * R * C);
  }
  // horizontal: tmp[ky][x] = sum_kx coef[ky][kx] * BC[kx][x]
  for (int ky = 0; ky < R; ky++) {
    float* t = &tmp[(size_t)ky * C];
    for (int x = 0; x < C; x++) t[x] = 0;
    for (int kx = 0; kx < C; kx++) {
      float c = coef[(size_t)ky * C + kx];
      if (c == 0) continue;
      const float* b = &BC[(size_t)kx * C];
      for (int x = 0; x < C; x++) t[x] += c * b[x];
    }
  }
  // vertical: out[y][x] = sum_ky tmp[ky][x] * BR[ky][y]
  for (int y = 0; y < R; y++) {
    float* o = out + (size_t)y * stride;
    for (int x = 0; x < C; x++) o[x] = 0;
    for (int ky = 0; ky < R; ky++) {
      float b = BR[(size_t)ky * R + y];
      const float* t = &tmp[(size_t)ky * C];
      for (int x = 0; x < C; x++) o[x] += b * t[x];
    }
  }
}

void DctStored(int R, int C, const float* in, int stride, float* stored) {
  const std::vector<float>& BR = Basis(R);
  const std::vector<float>& BC = Basis(C);
  const bool transposed = R >= C;
  std::vector<float> tmp((size_t)R * C), coef((size_t)R * C);
  // vertical: tmp[ky][x] = 1/R sum_y in[y][x] BR[ky][y]
  for (int ky = 0; ky < R; ky++) {
    float* t = &tmp[(size_t)ky * C];
    for (int x = 0; x < C; x++) t[x] = 0;
    for (int y = 0; y < R; y++) {
      float b = BR[(size_t)ky * R + y] / R;
      const float* p = in + (size_t)y * stride;
      for (int x = 0; x < C; x++) t[x] += b * p[x];
    }
  }
  for (int ky = 0; ky < R; ky++)
    for (int kx = 0; kx < C; kx++) {
      const float* b = &BC[(size_t)kx * C];
      const float* t = &tmp[(size_t)ky * C];
      float s = 0;
      for (int x = 0; x < C; x++) s += t[x] * b[x];
      coef[(size_t)ky * C + kx] = s / C;
    }
  if (transposed) {
    for (int kx = 0; kx < C; kx++) for (int ky = 0; ky < R; ky++) stored[(size_t)kx * R + ky] = coef[(size_t)ky * C + kx];
  } else {
    memcpy(stored, coef.data(), sizeof(float) * R * C);
  }
}

static double ResampleScale(int c, int k) {  // LLF coefficient k of an 8c-point DCT from the c-point DCT of block means
  if (k == 0) return 1.0;
  double t = k * M_PI / (2.0 * c);
  return 1.0 / (std::cos(t / 2) * std::cos(t / 4) * std::cos(t / 8));
}

void LlfFromLf(int strategy, const float* lf, int lf_stride, float* coeffs) {
  int cx = kCoveredX[strategy], cy = kCoveredY[strategy];
  if (cx == 1 && cy == 1) { coeffs[0] = lf[0]; return; }
  std::vector<float> d((size_t)cx * cy);
  DctStored(cy, cx, lf, lf_stride, d.data());  // stored layout of a cy x cx DCT
  const bool transposed = cy >= cx;
  const int lng = 8 * std::max(cx, cy);
  for (int ky = 0; ky < cy; ky++)
    for (int kx = 0; kx < cx; kx++) {
      float v = transposed ? d[(size_t)kx * cy + ky] : d[(size_t)ky * cx + kx];
      v *= (float)(ResampleScale(cy, ky) * ResampleScale(cx, kx));
      coeffs[transposed ? (size_t)kx * lng + ky : (size_t)ky * lng + kx] = v;
    }
}

void LfFromLlf(int strategy, const float* coeffs, float* lf, int lf_stride) {
  int cx = kCoveredX[strategy], cy = kCoveredY[strategy];
  if (cx == 1 && cy == 1) { lf[0] = coeffs[0]; return; }
  std::vector<float> d((size_t)cx * cy);
  const bool transposed = cy >= cx;
  const int lng = 8 * std::max(cx, cy);
  for (int ky = 0; ky < cy; ky++)
    for (int kx = 0; kx < cx; kx++) {
      float v = coeffs[transposed ? (size_t)kx * lng + ky : (size_t)ky * lng + kx];
      v /= (float)(ResampleScale(cy, ky) * ResampleScale(cx, kx));
      d[transposed ? (size_t)kx * cy + ky : (size_t)ky * cx + kx] = v;
    }
  IdctStored(cy, cx, d.data(), lf, lf_stride);
}

// ------------------------------------------------------------------ special 8x8 transforms
static void Idct2Top(int S, float* block) {  // in place on an 8x8 block (stride 8)
  float temp[64];
  int n = S / 2;
  for (int y = 0; y < n; y++)
    for (int x = 0; x < n; x++) {
      float c00 = block[y * 8 + x], c01 = block[y * 8 + n + x], c10 = block[(y + n) * 8 + x], c11 = block[(y + n) * 8 + n + x];
      temp[y * 2 * 8 + x * 2] = c00 + c01 + c10 + c11;
      temp[y * 2 * 8 + x * 2 + 1] = c00 + c01 - c10 - c11;
      temp[(y * 2 + 1) * 8 + x * 2] = c00 - c01 + c10 - c11;
      temp[(y * 2 + 1) * 8 + x * 2 + 1] = c00 - c01 - c10 + c11;
    }
  for (int y = 0; y < S; y++) for (int x = 0; x < S; x++) block[y * 8 + x] = temp[y * 8 + x];
}
static void Dct2Top(int S, float* block) {
  float temp[64];
  int n = S / 2;
  for (int y = 0; y < n; y++)
    for (int x = 0; x < n; x++) {
      float r00 = block[y * 2 * 8 + x * 2], r01 = block[y * 2 * 8 + x * 2 + 1], r10 = block[(y * 2 + 1) * 8 + x * 2],
            r11 = block[(y * 2 + 1) * 8 + x * 2 + 1];
      temp[y * 8 + x] = (r00 + r01 + r10 + r11) * 0.25f;
      temp[y * 8 + n + x] = (r00 + r01 - r10 - r11) * 0.25f;
      temp[(y + n) * 8 + x] = (r00 - r01 + r10 - r11) * 0.25f;
      temp[(y + n) * 8 + n + x] = (r00 - r01 - r10 + r11) * 0.25f;
    }
  for (int y = 0; y < S; y++) for (int x = 0; x < S; x++) block[y * 8 + x] = temp[y * 8 + x];
}

void InverseTransform(int strategy, const float* coefficients, float* pixels, int stride) {
  switch (strategy) {
    case IDENTITY: {
      float b00 = coefficients[0], b01 = coefficients[1], b10 = coefficients[8], b11 = coefficients[9];
      float dcs[4] = {b00 + b01 + b10 + b11, b00 + b01 - b10 - b11, b00 - b01 + b10 - b11, b00 - b01 - b10 + b11};
      for (int y = 0; y < 2; y++)
        for (int x = 0; x < 2; x++) {
          float block_dc = dcs[y * 2 + x], residual_sum = 0;
          for (int iy = 0; iy < 4; iy++)
            for (int ix = 0; ix < 4; ix++) {
              if (ix == 0 && iy == 0) continue;
              residual_sum += coefficients[(y + iy * 2) * 8 + x + ix * 2];
            }
          float ref = block_dc - residual_sum * (1.0f / 16);
          pixels[(4 * y + 1) * stride + 4 * x + 1] = ref;
          for (int iy = 0; iy < 4; iy++)
            for (int ix = 0; ix < 4; ix++) {
              if (ix == 1 && iy == 1) continue;
              pixels[(y * 4 + iy) * stride + x * 4 + ix] = coefficients[(y + iy * 2) * 8 + x + ix * 2] + ref;
            }
          pixels[y * 4 * stride + x * 4] = coefficients[(y + 2) * 8 + x + 2] + ref;
        }
      return;
    }
    case DCT2X2: {
      float c[64];
      memcpy(c, coefficients, sizeof(c));
      Idct2Top(2, c); Idct2Top(4, c); Idct2Top(8, c);
      for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) pixels[y * stride + x] = c[y * 8 + x];
      return;
    }
    case DCT4X4: {
      float b00 = coefficients[0], b01 = coefficients[1], b10 = coefficients[8], b11 = coefficients[9];
      float dcs[4] = {b00 + b01 + b10 + b11, b00 + b01 - b10 - b11, b00 - b01 + b10 - b11, b00 - b01 - b10 + b11};
      for (int y = 0; y < 2; y++)
        for (int x = 0; x < 2; x++) {
          float block[16];
          block[0] = dcs[y * 2 + x];
          for (int iy = 0; iy < 4; iy++)
            for (int ix = 0; ix < 4; ix++) {
              if (ix == 0 && iy == 0) continue;
              block[iy * 4 + ix] = coefficients[(y + iy * 2) * 8 + x + ix * 2];
            }
          IdctStored(4, 4, block, pixels + y * 4 * stride + x * 4, stride);
        }
      return;
    }
    case DCT4X8:
    case DCT8X4: {
      float b0 = coefficients[0], b1 = coefficients[8];
      float dcs[2] = {b0 + b1, b0 - b1};
      for (int h = 0; h < 2; h++) {
        float block[32];
        block[0] = dcs[h];
        for (int iy = 0; iy < 4; iy++)
          for (int ix = 0; ix < 8; ix++) {
            if (ix == 0 && iy == 0) continue;
            block[iy * 8 + ix] = coefficients[(h + iy * 2) * 8 + ix];
          }
        if (strategy == DCT4X8) IdctStored(4, 8, block, pixels + h * 4 * stride, stride);
        else IdctStored(8, 4, block, pixels + h * 4, stride);
      }
      return;
    }
    case AFV0: case AFV1: case AFV2: case AFV3:
      throw Error("AFV transforms are not supported by the oracle yet");
    default:
      IdctStored(8 * kCoveredY[strategy], 8 * kCoveredX[strategy], coefficients, pixels, stride);
  }
}

void ForwardTransform(int strategy, const float* pixels, int stride, float* coefficients) {
  switch (strategy) {
    case IDENTITY: {
      float dcs[4];
      for (int y = 0; y < 2; y++)
        for (int x = 0; x < 2; x++) {
          const float* p = pixels + y * 4 * stride + x * 4;
          float ref = p[stride + 1], sum = 0;
          for (int iy = 0; iy < 4; iy++)
            for (int ix = 0; ix < 4; ix++) {
              sum += p[iy * stride + ix];
              if ((ix == 0 && iy == 0) || (ix == 1 && iy == 1)) continue;
              coefficients[(y + iy * 2) * 8 + x + ix * 2] = p[iy * stride + ix] - ref;
            }
          coefficients[(y + 2) * 8 + x + 2] = p[0] - ref;
          dcs[y * 2 + x] = sum * (1.0f / 16);
        }
      coefficients[0] = (dcs[0] + dcs[1] + dcs[2] + dcs[3]) * 0.25f;
      coefficients[1] = (dcs[0] + dcs[1] - dcs[2] - dcs[3]) * 0.25f;
      coefficients[8] = (dcs[0] - dcs[1] + dcs[2] - dcs[3]) * 0.25f;
      coefficients[9] = (dcs[0] - dcs[1] - dcs[2] + dcs[3]) * 0.25f;
      return;
    }
    case DCT2X2: {
      float c[64];
      for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) c[y * 8 + x] = pixels[y * stride + x];
      Dct2Top(8, c); Dct2Top(4, c); Dct2Top(2, c);
      memcpy(coefficients, c, sizeof(c));
      return;
    }
    case DCT4X4: {
      float dcs[4];
      for (int y = 0; y < 2; y++)
        for (int x = 0; x < 2; x++) {
          float block[16];
          DctStored(4, 4, pixels + y * 4 * stride + x * 4, stride, block);
          dcs[y * 2 + x] = block[0];
          for (int iy = 0; iy < 4; iy++)
            for (int ix = 0; ix < 4; ix++) {
              if (ix == 0 && iy == 0) continue;
              coefficients[(y + iy * 2) * 8 + x + ix * 2] = block[iy * 4 + ix];
            }
        }
      coefficients[0] = (dcs[0] + dcs[1] + dcs[2] + dcs[3]) * 0.25f;
      coefficients[1] = (dcs[0] + dcs[1] - dcs[2] - dcs[3]) * 0.25f;
      coefficients[8] = (dcs[0] - dcs[1] + dcs[2] - dcs[3]) * 0.25f;
      coefficients[9] = (dcs[0] - dcs[1] - dcs[2] + dcs[3]) * 0.25f;
      return;
    }
    case DCT4X8:
    case DCT8X4: {
      float dcs[2];
      for (int h = 0; h < 2; h++) {
        float block[32];
        if (strategy == DCT4X8) DctStored(4, 8, pixels + h * 4 * stride, stride, block);
        else DctStored(8, 4, pixels + h * 4, stride, block);
        dcs[h] = block[0];
        for (int iy = 0; iy < 4; iy++)
          for (int ix = 0; ix < 8; ix++) {
            if (ix == 0 && iy == 0) continue;
            coefficients[(h + iy * 2) * 8 + ix] = block[iy * 8 + ix];
          }
      }
      coefficients[0] = (dcs[0] + dcs[1]) * 0.5f;
      coefficients[8] = (dcs[0] - dcs[1]) * 0.5f;
      return;
    }
    case AFV0: case AFV1: case AFV2: case AFV3:
      throw Error("AFV transforms are not supported by the oracle yet");
    default:
      DctStored(8 * kCoveredY[strategy], 8 * kCoveredX[strategy], pixels, stride, coefficients);
  }
}

// ------------------------------------------------------------------ dequant matrices
namespace {
struct DctParams { int num_bands; float bands[3][17]; };

float Mult(float v) { return v > 0 ? 1 + v : 1 / (1 - v); }

float Interpolate(float pos, float max, const float* array, int len) {
  float scaled = pos * (len - 1) / max;
  int idx = (int)scaled;
  JXO_CHECK(idx + 1 < len, "interpolate index");
  float a = array[idx], b = array[idx + 1];
  return a * std::pow(b / a, scaled - idx);
}

void GetQuantWeights(int rows, int cols, const DctParams& p, float* out) {
  for (int c = 0; c < 3; c++) {
    float bands[17];
    bands[0] = p.bands[c][0];
    JXO_CHECK(bands[0] >= 1e-8f, "invalid distance bands");
    for (int i = 1; i < p.num_bands; i++) {
      bands[i] = bands[i - 1] * Mult(p.bands[c][i]);
      JXO_CHECK(bands[i] >= 1e-8f, "invalid distance bands");
    }
    float scale = (p.num_bands - 1) / ((float)std::sqrt(2.0) + 1e-6f);
    float rcpcol = scale / (cols - 1), rcprow = scale / (rows - 1);
    for (int y = 0; y < rows; y++) {
      float dy = y * rcprow;
      for (int x = 0; x < cols; x++) {
        float dx = x * rcpcol;
        float d = std::sqrt(dx * dx + dy * dy);
        float w;
        if (p.num_bands == 1) w = bands[0];
        else {
          int idx = (int)d;
          float frac = d - idx;
          JXO_CHECK(idx + 1 < p.num_bands || frac == 0, "band index");
          float a = bands[idx], b = idx + 1 < p.num_bands ? bands[idx + 1] : a;
          w = a * std::pow(b / a, frac);
        }
        out[(size_t)c * rows * cols + (size_t)y * cols + x] = w;
      }
    }
  }
}

struct QuantEncoding {
  int mode = 0;  // 0 library, 1 identity, 2 dct2, 3 dct4, 4 dct4x8, 5 afv, 6 dct
  float idweights[3][3];
  float dct2weights[3][6];
  float dct4multipliers[3][2];
  float dct4x8multipliers[3];
  float afv_weights[3][9];
  DctParams dct, dct_afv4x4;
};

DctParams MakeParams(int n, std::initializer_list<std::initializer_list<double>> v) {
  DctParams p;
  p.num_bands = n;
  int c = 0;
  for (auto& ch : v) {
    int i = 0;
    for (double x : ch) p.bands[c][i++] = (float)x;
    c++;
  }
  return p;
}

const int kReqX[kNumQuantTables] = {1, 1, 1, 1, 2, 4, 1, 1, 2, 1, 1, 8, 4, 16, 8, 32, 16};
const int kReqY[kNumQuantTables] = {1, 1, 1, 1, 2, 4, 2, 4, 4, 1, 1, 8, 8, 16, 16, 32, 32};

QuantEncoding LibraryEncoding(int q) {
  QuantEncoding e;
  auto big = [](double m0, double m1, double m2, bool wide) {
    if (!wide)
      return MakeParams(8, {{m0 * 26629.073922049845, -1.025, -0.78, -0.65012, -0.19041574084286472, -0.20819395464, -0.421064, -0.32733845535848671},
                            {m1 * 9311.3238710010046, -0.3041958212306401, -0.3633036457487539, -0.35660379990111464, -0.3443074455424403,
                             -0.33699592683512467, -0.30180866526242109, -0.27321683125358037},
                            {m2 * 4992.2486445538634, -1.2, -1.2, -0.8, -0.7, -0.7, -0.4, -0.5}});
    return MakeParams(8, {{m0 * 23629.073922049845, -1.025, -0.78, -0.65012, -0.19041574084286472, -0.20819395464, -0.421064, -0.32733845535848671},
                          {m1 * 8611.3238710010046, -0.3041958212306401, -0.3633036457487539, -0.35660379990111464, -0.3443074455424403,
                           -0.33699592683512467, -0.30180866526242109, -0.27321683125358037},
                          {m2 * 4492.2486445538634, -1.2, -1.2, -0.8, -0.7, -0.7, -0.4, -0.5}});
  };
  DctParams p4x8 = MakeParams(4, {{2198.050556016380522, -0.96269623020744692, -0.76194253026666783, -0.6551140670773547},
                                  {764.3655248643528689, -0.92630200888366945, -0.9675229603596517, -0.27845290869168118},
                                  {527.107573587542228, -1.4594385811273854, -1.450082094097871593, -1.5843722511996204}});
  DctParams p4x4 = MakeParams(4, {{2200.0, 0.0, 0.0, 0.0}, {392.0, 0.0, 0.0, 0.0}, {112.0, -0.25, -0.25, -0.5}});
  switch (q) {
    case 0:
      e.mode = 6;
      e.dct = MakeParams(6, {{3150.0, 0.0, -0.4, -0.4, -0.4, -2.0}, {560.0, 0.0, -0.3, -0.3, -0.3, -0.3}, {512.0, -2.0, -1.0, 0.0, -1.0, -2.0}});
      break;
    case 1: {
      e.mode = 1;
      const float w[3][3] = {{280.0f, 3160.0f, 3160.0f}, {60.0f, 864.0f, 864.0f}, {18.0f, 200.0f, 200.0f}};
      memcpy(e.idweights, w, sizeof(w));
      break;
    }
    case 2: {
      e.mode = 2;
      const float w[3][6] = {{3840.0f, 2560.0f, 1280.0f, 640.0f, 480.0f, 300.0f}, {960.0f, 640.0f, 320.0f, 180.0f, 140.0f, 120.0f},
                             {640.0f, 320.0f, 128.0f, 64.0f, 32.0f, 16.0f}};
      memcpy(e.dct2weights, w, sizeof(w));
      break;
    }
    case 3:
      e.mode = 3;
      e.dct = p4x4;
      for (int c = 0; c < 3; c++) e.dct4multipliers[c][0] = e.dct4multipliers[c][1] = 1.0f;
      break;
    case 4:
      e.mode = 6;
      e.dct = MakeParams(7, {{8996.8725711814115328, -1.3000777393353804, -0.49424529824571225, -0.439093774457103443, -0.6350101832695744,
                              -0.90177264050827612, -1.6162099239887414},
                             {3191.48366296844234752, -0.67424582104194355, -0.80745813428471001, -0.44925837484843441, -0.35865440981033403,
                              -0.31322389111877305, -0.37615025315725483},
                             {1157.50408145487200256, -2.0531423165804414, -1.4, -0.50687130033378396, -0.42708730624733904,
                              -1.4856834539296244, -4.9209142884401604}});
      break;
    case 5:
      e.mode = 6;
      e.dct = MakeParams(8, {{15718.40830982518931456, -1.025, -0.98, -0.9012, -0.4, -0.48819395464, -0.421064, -0.27},
                             {7305.7636810695983104, -0.8041958212306401, -0.7633036457487539, -0.55660379990111464, -0.49785304658857626,
                              -0.43699592683512467, -0.40180866526242109, -0.27321683125358037},
                             {3803.53173721215041536, -3.060733579805728, -2.0413270132490346, -2.0235650159727417, -0.5495389509954993, -0.4,
                              -0.4, -0.3}});
      break;
    case 6:
      e.mode = 6;
      e.dct = MakeParams(7, {{7240.7734393502, -0.7, -0.7, -0.2, -0.2, -0.2, -0.5}, {1448.15468787004, -0.5, -0.5, -0.5, -0.2, -0.2, -0.2},
                             {506.854140754517, -1.4, -0.2, -0.5, -0.5, -1.5, -3.6}});
      break;
    case 7:
      e.mode = 6;
      e.dct = MakeParams(8, {{16283.2494710648897, -1.7812845336559429, -1.6309059012653515, -1.0382179034313539, -0.85, -0.7, -0.9,
                              -1.2360638576849587},
                             {5089.15750884921511936, -0.320049391452786891, -0.35362849922161446, -0.30340000000000003, -0.61, -0.5, -0.5, -0.6},
                             {3397.77603275308720128, -0.321327362693153371, -0.34507619223117997, -0.70340000000000003, -0.9, -1.0, -1.0,
                              -1.1754605576265209}});
      break;
    case 8:
      e.mode = 6;
      e.dct = MakeParams(8, {{13844.97076442300573, -0.97113799999999995, -0.658, -0.42026, -0.22712, -0.2206, -0.226, -0.6},
                             {4798.964084220744293, -0.61125308982767057, -0.83770786552491361, -0.79014862079498627, -0.2692727459704829,
                              -0.38272769465388551, -0.22924222653091453, -0.20719098826199578},
                             {1807.236946760964614, -1.2, -1.2, -0.7, -0.7, -0.7, -0.4, -0.5}});
      break;
    case 9:
      e.mode = 4;
      e.dct = p4x8;
      for (int c = 0; c < 3; c++) e.dct4x8multipliers[c] = 1.0f;
      break;
    case 10: {
      e.mode = 5;
      e.dct = p4x8;
      e.dct_afv4x4 = p4x4;
      const float w[3][9] = {{3072.0f, 3072.0f, 256.0f, 256.0f, 256.0f, 414.0f, 0.0f, 0.0f, 0.0f},
                             {1024.0f, 1024.0f, 50.0f, 50.0f, 50.0f, 58.0f, 0.0f, 0.0f, 0.0f},
                             {384.0f, 384.0f, 12.0f, 12.0f, 12.0f, 22.0f, -0.25f, -0.25f, -0.25f}};
      memcpy(e.afv_weights, w, sizeof(w));
      break;
    }
    case 11: e.mode = 6; e.dct = big(0.9, 0.9, 0.9, false); break;
    case 12: e.mode = 6; e.dct = big(0.65, 0.65, 0.65, true); break;
    case 13: e.mode = 6; e.dct = big(1.8, 1.8, 1.8, false); break;
    case 14: e.mode = 6; e.dct = big(1.3, 1.3, 1.3, true); break;
    case 15: e.mode = 6; e.dct = big(3.6, 3.6, 3.6, false); break;
    case 16: e.mode = 6; e.dct = big(2.6, 2.6, 2.6, true); break;
  }
  return e;
}

void ComputeQuantTable(int q, const QuantEncoding& e, std::vector<float>& table, size_t* n_out) {
  int wrows = 8 * kReqX[q], wcols = 8 * kReqY[q];
  size_t n = (size_t)wrows * wcols;
  std::vector<float> weights(3 * n, 0.f);
  switch (e.mode) {
    case 6: GetQuantWeights(wrows, wcols, e.dct, weights.data()); break;
    case 1:
      JXO_CHECK(n == 64, "identity table size");
      for (int c = 0; c < 3; c++) {
        for (int i = 0; i < 64; i++) weights[64 * c + i] = e.idweights[c][0];
        weights[64 * c + 1] = e.idweights[c][1];
        weights[64 * c + 8] = e.idweights[c][1];
        weights[64 * c + 9] = e.idweights[c][2];
      }
      break;
    case 2:
      JXO_CHECK(n == 64, "dct2 table size");
      for (int c = 0; c < 3; c++) {
        float* w = &weights[64 * c];
        int start = 1, end = 2;
        w[0] = 1.0f;
        for (int i = 0; i < 6; i += 2) {
          // bands: [start,end) x [0,start) (and transposed) then [start,end) x [start,end)
          for (int y = 0; y < start; y++)
            for (int x = start; x < end; x++) { w[y * 8 + x] = e.dct2weights[c][i]; w[x * 8 + y] = e.dct2weights[c][i]; }
          for (int y = start; y < end; y++)
            for (int x = start; x < end; x++) w[y * 8 + x] = e.dct2weights[c][i + 1];
          start = end;
          end *= 2;
        }
      }
      break;
    case 3: {
      JXO_CHECK(n == 64, "dct4 table size");
      float w4[3 * 16];
      GetQuantWeights(4, 4, e.dct, w4);
      for (int c = 0; c < 3; c++) {
        for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) weights[64 * c + y * 8 + x] = w4[16 * c + (y / 2) * 4 + x / 2];
        weights[64 * c + 1] /= e.dct4multipliers[c][0];
        weights[64 * c + 8] /= e.dct4multipliers[c][0];
        weights[64 * c + 9] /= e.dct4multipliers[c][1];
      }
      break;
    }
    case 4: {
      JXO_CHECK(n == 64, "dct4x8 table size");
      float w48[3 * 32];
      GetQuantWeights(4, 8, e.dct, w48);
      for (int c = 0; c < 3; c++) {
        for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) weights[64 * c + y * 8 + x] = w48[32 * c + (y / 2) * 8 + x];
        weights[64 * c + 8] /= e.dct4x8multipliers[c];
      }
      break;
    }
    case 5: {
      JXO_CHECK(n == 64, "afv table size");
      static const float kFreqs[16] = {0, 0, 0.8517778890324296f, 5.37778436506804f, 0, 0, 4.734747904497923f, 5.449245381693219f,
                                       1.6598270267479331f, 4, 7.275749096817861f, 10.423227632456525f, 2.662932286148962f,
                                       7.630657783650829f, 8.962388608184032f, 12.97166202570235f};
      float w48[3 * 32], w44[3 * 16];
      GetQuantWeights(4, 8, e.dct, w48);
      GetQuantWeights(4, 4, e.dct_afv4x4, w44);
      const float lo = 0.8517778890324296f, hi = 12.97166202570235f - lo + 1e-6f;
      for (int c = 0; c < 3; c++) {
        float bands[4];
        bands[0] = e.afv_weights[c][5];
        for (int i = 1; i < 4; i++) bands[i] = bands[i - 1] * Mult(e.afv_weights[c][i + 5]);
        float* w = &weights[64 * c];
        auto set = [&](int x, int y, float v) { w[y * 8 + x] = v; };
        w[0] = 1;
        set(0, 1, e.afv_weights[c][0]);
        set(1, 0, e.afv_weights[c][1]);
        set(0, 2, e.afv_weights[c][2]);
        set(2, 0, e.afv_weights[c][3]);
        set(2, 2, e.afv_weights[c][4]);
        for (int y = 0; y < 4; y++)
          for (int x = 0; x < 4; x++) {
            if (x < 2 && y < 2) continue;
            set(2 * x, 2 * y, Interpolate(kFreqs[y * 4 + x] - lo, hi, bands, 4));
          }
        for (int y = 0; y < 4; y++)
          for (int x = 0; x < 8; x++) {
            if (x == 0 && y == 0) continue;
            w[(2 * y + 1) * 8 + x] = w48[32 * c + y * 8 + x];
          }
        for (int y = 0; y < 4; y++)
          for (int x = 0; x < 4; x++) {
            if (x == 0 && y == 0) continue;
            w[(2 * y) * 8 + 2 * x + 1] = w44[16 * c + y * 4 + x];
          }
      }
      break;
    }
    default: throw Error("quant encoding mode");
  }
  table.resize(3 * n);
  for (size_t i = 0; i < 3 * n; i++) {
    JXO_CHECK(weights[i] > 1e-8f && weights[i] < 1e8f, "invalid quant weight");
    table[i] = 1.0f / weights[i];
  }
  *n_out = n;
}

void ReadDctParams(BitReader& br, DctParams& p) {
  p.num_bands = br.Read(4) + 1;
  for (int c = 0; c < 3; c++) {
    for (int i = 0; i < p.num_bands; i++) p.bands[c][i] = br.F16();
    p.bands[c][0] *= 64;
  }
}
}  // namespace

void DequantMatrices::SetDefault() {
  for (int q = 0; q < kNumQuantTables; q++) ComputeQuantTable(q, LibraryEncoding(q), table[q], &n[q]);
}

void DequantMatrices::SetCustomAndWrite(uint32_t seed, BitWriter& bw) {
  bw.Bool(false);   // not all_default
  auto w16 = [&](float& v, float scale, float store_div) {   // parameter -> scaled, rounded to what F16 carries, written
    v = RoundToF16(v * scale / store_div) * store_div;
    bw.F16(v / store_div);
  };
  auto bands = [&](DctParams& pr, float scale) {
    bw.Write(4, pr.num_bands - 1);
    for (int c = 0; c < 3; c++)
      for (int i = 0; i < pr.num_bands; i++) w16(pr.bands[c][i], i == 0 ? scale : 1.0f, i == 0 ? 64.0f : 1.0f);
  };
  for (int q = 0; q < kNumQuantTables; q++) {
    QuantEncoding e = LibraryEncoding(q);
    const float scale = 1.0f + 0.04f * (float)((int)((seed * 7 + q * 13) % 7) - 3);   // 0.88 .. 1.12
    bw.Write(3, e.mode);
    switch (e.mode) {
      case 1: for (int c = 0; c < 3; c++) for (int i = 0; i < 3; i++) w16(e.idweights[c][i], scale, 64.0f); break;
      case 2: for (int c = 0; c < 3; c++) for (int i = 0; i < 6; i++) w16(e.dct2weights[c][i], scale, 64.0f); break;
      case 3: for (int c = 0; c < 3; c++) for (int i = 0; i < 2; i++) w16(e.dct4multipliers[c][i], 1.0f, 1.0f); bands(e.dct, scale); break;
      case 4: for (int c = 0; c < 3; c++) w16(e.dct4x8multipliers[c], 1.0f, 1.0f); bands(e.dct, scale); break;
      case 5:
        for (int c = 0; c < 3; c++) for (int i = 0; i < 9; i++) w16(e.afv_weights[c][i], i < 6 ? scale : 1.0f, i < 6 ? 64.0f : 1.0f);
        bands(e.dct, scale); bands(e.dct_afv4x4, scale);
        break;
      case 6: bands(e.dct, scale); break;
      default: throw Error("library encoding mode");
    }
    ComputeQuantTable(q, e, table[q], &n[q]);
  }
}

void DequantMatrices::Decode(BitReader& br) {
  bool all_default = br.Bool();
  if (all_default) { SetDefault(); return; }
  for (int q = 0; q < kNumQuantTables; q++) {
    QuantEncoding e;
    e.mode = br.Read(3);
    bool small = kReqX[q] == 1 && kReqY[q] == 1;
    switch (e.mode) {
      case 0: e = LibraryEncoding(q); break;
      case 1:
        JXO_CHECK(small, "identity encoding for large table");
        for (int c = 0; c < 3; c++) for (int i = 0; i < 3; i++) e.idweights[c][i] = br.F16() * 64;
        break;
      case 2:
        JXO_CHECK(small, "dct2 encoding for large table");
        for (int c = 0; c < 3; c++) for (int i = 0; i < 6; i++) e.dct2weights[c][i] = br.F16() * 64;
        break;
      case 3:
        JXO_CHECK(small, "dct4 encoding for large table");
        for (int c = 0; c < 3; c++) for (int i = 0; i < 2; i++) e.dct4multipliers[c][i] = br.F16();
        ReadDctParams(br, e.dct);
        break;
      case 4:
        JXO_CHECK(small, "dct4x8 encoding for large table");
        for (int c = 0; c < 3; c++) e.dct4x8multipliers[c] = br.F16();
        ReadDctParams(br, e.dct);
        break;
      case 5:
        JXO_CHECK(small, "afv encoding for large table");
        for (int c = 0; c < 3; c++) {
          for (int i = 0; i < 9; i++) e.afv_weights[c][i] = br.F16();
          for (int i = 0; i < 6; i++) e.afv_weights[c][i] *= 64;
        }
        ReadDctParams(br, e.dct);
        ReadDctParams(br, e.dct_afv4x4);
        break;
      case 6: ReadDctParams(br, e.dct); break;
      default: throw Error("RAW quant tables are not supported by the oracle yet");
    }
    ComputeQuantTable(q, e, table[q], &n[q]);
  }
}

// ------------------------------------------------------------------ block context map
void BlockCtxMap::SetDefault() {
  static const uint8_t kDefault[39] = {0, 1, 2, 2, 3, 3, 4, 5, 6, 6, 6, 6, 6, 7, 8, 9, 9, 10, 11, 12,
                                       13, 14, 14, 14, 14, 14, 7, 8, 9, 9, 10, 11, 12, 13, 14, 14, 14, 14, 14};
  for (auto& t : lf_thresholds) t.clear();
  qf_thresholds.clear();
  ctx_map.assign(kDefault, kDefault + 39);
  num_ctxs = 15;
  num_lf_ctxs = 1;
}

void BlockCtxMap::Decode(BitReader& br) {
  SetDefault();
  if (br.Bool()) return;
  num_lf_ctxs = 1;
  for (int j = 0; j < 3; j++) {
    uint32_t n = br.Read(4);
    lf_thresholds[j].resize(n);
    for (auto& t : lf_thresholds[j]) t = (int32_t)UnpackSigned(br.U32(Bits(4), BitsOff(8, 16), BitsOff(16, 272), BitsOff(32, 65808)));
    num_lf_ctxs *= n + 1;
  }
  uint32_t nqf = br.Read(4);
  qf_thresholds.resize(nqf);
  for (auto& t : qf_thresholds) t = br.U32(Bits(2), BitsOff(3, 4), BitsOff(5, 12), BitsOff(8, 44)) + 1;
  size_t sz = 3 * kNumOrders * (nqf + 1) * num_lf_ctxs;
  JXO_CHECK(sz <= 39 * 64, "block context map too large");
  ctx_map.assign(sz, 0);
  DecodeContextMap(br, ctx_map, &num_ctxs);
  JXO_CHECK(num_ctxs <= 16, "too many block contexts");
}

// ------------------------------------------------------------------ LF smoothing, loop filters
void AdaptiveLfSmoothing(Plane lf[3], const float f[3]) {
  int w = lf[0].w, h = lf[0].h;
  if (w <= 2 || h <= 2) return;
  Plane out[3] = {lf[0], lf[1], lf[2]};
  const float kW0 = 0.05226273532324128f, kW1 = 0.20345139757231578f, kW2 = 0.0334829185968739f;
  for (int y = 1; y + 1 < h; y++) {
    for (int x = 1; x + 1 < w; x++) {
      float sm[3], mc[3], gap = 0.5f;
      for (int c = 0; c < 3; c++) {
        const float* t = lf[c].Row(y - 1);
        const float* m = lf[c].Row(y);
        const float* b = lf[c].Row(y + 1);
        float corner = t[x - 1] + t[x + 1] + b[x - 1] + b[x + 1];
        float edge = t[x] + m[x - 1] + m[x + 1] + b[x];
        mc[c] = m[x];
        sm[c] = corner * kW2 + edge * kW1 + mc[c] * kW0;
        gap = std::max(gap, std::fabs((mc[c] - sm[c]) / f[c]));
      }
      float factor = std::max(0.0f, 3.0f - 4.0f * gap);
      for (int c = 0; c < 3; c++) out[c].Row(y)[x] = (sm[c] - mc[c]) * factor + mc[c];
    }
  }
  for (int c = 0; c < 3; c++) lf[c] = out[c];
}

static inline int Mirror(int v, int n) {
  while (v < 0 || v >= n) {
    if (v < 0) v = -v - 1;
    else v = 2 * n - 1 - v;
  }
  return v;
}

void Gaborish(Plane xyb[3], const LoopFilter& lf) {
  for (int c = 0; c < 3; c++) {
    const Plane& in = xyb[c];
    Plane out(in.w, in.h);
    float div = 1.0f + 4.0f * (lf.gab_w1[c] + lf.gab_w2[c]);
    float w0 = 1.0f / div, w1 = lf.gab_w1[c] / div, w2 = lf.gab_w2[c] / div;
    for (int y = 0; y < in.h; y++) {
      const float* t = in.Row(Mirror(y - 1, in.h));
      const float* m = in.Row(y);
      const float* b = in.Row(Mirror(y + 1, in.h));
      float* o = out.Row(y);
      for (int x = 0; x < in.w; x++) {
        int xl = Mirror(x - 1, in.w), xr = Mirror(x + 1, in.w);
        o[x] = m[x] * w0 + (t[x] + b[x] + m[xl] + m[xr]) * w1 + (t[xl] + t[xr] + b[xl] + b[xr]) * w2;
      }
    }
    xyb[c] = out;
  }
}

static void EpfPass(int stage, const Plane in[3], Plane out[3], const LoopFilter& lf, const Plane& inv_sigma) {
  static const int kOff0[12][2] = {{-2, 0}, {-1, -1}, {-1, 0}, {-1, 1}, {0, -2}, {0, -1}, {0, 1}, {0, 2}, {1, -1}, {1, 0}, {1, 1}, {2, 0}};
  static const int kOff1[4][2] = {{-1, 0}, {0, -1}, {0, 1}, {1, 0}};
  static const int kPlus[5][2] = {{0, 0}, {-1, 0}, {1, 0}, {0, -1}, {0, 1}};
  const float kMinSigma = -3.90524291751269967465540850526868f;
  const int w = in[0].w, h = in[0].h;
  const int noff = stage == 0 ? 12 : 4;
  const int (*off)[2] = stage == 0 ? kOff0 : kOff1;
  const int nplus = stage == 2 ? 1 : 5;
  float sm = stage == 0 ? lf.epf_pass0_sigma_scale : (stage == 1 ? 1.0f : lf.epf_pass2_sigma_scale);
  float bsm = sm * lf.epf_border_sad_mul;
  auto px = [&](int c, int y, int x) { return in[c].Row(Mirror(y, h))[Mirror(x, w)]; };
  for (int y = 0; y < h; y++) {
    bool yborder = (y % 8 == 0) || (y % 8 == 7);
    for (int x = 0; x < w; x++) {
      float is = inv_sigma.Row(y / 8)[x / 8];
      if (is < kMinSigma) {
        for (int c = 0; c < 3; c++) out[c].Row(y)[x] = in[c].Row(y)[x];
        continue;
      }
      bool xborder = (x % 8 == 0) || (x % 8 == 7);
      float inv = is * ((yborder || xborder) ? bsm : sm);
      float wsum = 1.0f, acc[3] = {in[0].Row(y)[x], in[1].Row(y)[x], in[2].Row(y)[x]};
      for (int i = 0; i < noff; i++) {
        float sad = 0;
        for (int c = 0; c < 3; c++) {
          float s = 0;
          for (int p = 0; p < nplus; p++)
            s += std::fabs(px(c, y + kPlus[p][0], x + kPlus[p][1]) - px(c, y + off[i][0] + kPlus[p][0], x + off[i][1] + kPlus[p][1]));
          sad += s * lf.epf_channel_scale[c];
        }
        float wt = std::max(0.0f, 1.0f + sad * inv);
        wsum += wt;
        for (int c = 0; c < 3; c++) acc[c] += wt * px(c, y + off[i][0], x + off[i][1]);
      }
      float iw = 1.0f / wsum;
      for (int c = 0; c < 3; c++) out[c].Row(y)[x] = acc[c] * iw;
    }
  }
}

void Epf(Plane xyb[3], const LoopFilter& lf, const Plane& inv_sigma) {
  if (lf.epf_iters == 0) return;
  Plane tmp[3] = {Plane(xyb[0].w, xyb[0].h), Plane(xyb[0].w, xyb[0].h), Plane(xyb[0].w, xyb[0].h)};
  auto run = [&](int stage) {
    EpfPass(stage, xyb, tmp, lf, inv_sigma);
    for (int c = 0; c < 3; c++) std::swap(xyb[c].d, tmp[c].d);
  };
  if (lf.epf_iters == 3) run(0);
  if (lf.epf_iters >= 1) run(1);
  if (lf.epf_iters >= 2) run(2);
}

// ------------------------------------------------------------------ colour
void XybToLinear(const ImageMetadata& m, Plane xyb[3]) {
  // linear RGB of the image's own primaries: the change of primaries is folded into the inverse opsin matrix
  double conv[9];
  JXO_CHECK(MatrixFromSrgbGeneral(m.color, conv), "this combination of primaries and white point is not supported");
  if (m.color.color_space == 1) MatrixFromSrgb(1, conv);
  float inv[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double a = 0;
      for (int k = 0; k < 3; k++) a += conv[r * 3 + k] * (double)m.opsin_inverse[k * 3 + c];
      inv[r * 3 + c] = (float)a * (255.0f / m.intensity_target);
    }
  float bias[3], cb[3];
  for (int i = 0; i < 3; i++) { bias[i] = m.opsin_bias[i]; cb[i] = std::cbrt(bias[i]); }
  size_t n = xyb[0].d.size();
  for (size_t i = 0; i < n; i++) {
    float X = xyb[0].d[i], Y = xyb[1].d[i], B = xyb[2].d[i];
    float gr = Y + X - cb[0], gg = Y - X - cb[1], gb = B - cb[2];
    float mr = gr * gr * gr + bias[0], mg = gg * gg * gg + bias[1], mb = gb * gb * gb + bias[2];
    xyb[0].d[i] = inv[0] * mr + inv[1] * mg + inv[2] * mb;
    xyb[1].d[i] = inv[3] * mr + inv[4] * mg + inv[5] * mb;
    xyb[2].d[i] = inv[6] * mr + inv[7] * mg + inv[8] * mb;
  }
}

void LinearToXyb(Plane rgb[3]) {
  const float kM[9] = {0.30f, 0.622f, 0.078f, 0.23f, 0.692f, 0.078f, 0.24342268924547819f, 0.20476744424496821f, 0.55180986650955360f};
  const float kB = 0.0037930732552754493f;
  const float cb = std::cbrt(kB);
  size_t n = rgb[0].d.size();
  for (size_t i = 0; i < n; i++) {
    float r = rgb[0].d[i], g = rgb[1].d[i], b = rgb[2].d[i];
    float mr = kM[0] * r + kM[1] * g + kM[2] * b + kB;
    float mg = kM[3] * r + kM[4] * g + kM[5] * b + kB;
    float mb = kM[6] * r + kM[7] * g + kM[8] * b + kB;
    mr = std::max(mr, 0.f); mg = std::max(mg, 0.f); mb = std::max(mb, 0.f);
    float gr = std::cbrt(mr) - cb, gg = std::cbrt(mg) - cb, gb = std::cbrt(mb) - cb;
    rgb[0].d[i] = 0.5f * (gr - gg);
    rgb[1].d[i] = 0.5f * (gr + gg);
    rgb[2].d[i] = gb;
  }
}

float LinearToSrgb(float v) {   // sign-symmetric
  const float a = std::fabs(v);
  const float r = a <= 0.0031308f ? 12.92f * a : 1.055f * std::pow(a, 1.0f / 2.4f) - 0.055f;
  return std::copysign(r, v);
}
float SrgbToLinear(float v) {
  const float a = std::fabs(v);
  const float r = a <= 0.04045f ? a / 12.92f : std::pow((a + 0.055f) / 1.055f, 2.4f);
  return std::copysign(r, v);
}

double PowerLawGamma(const ColorEncoding& c) {
  if (c.all_default) return 0.0;
  if (c.have_gamma) return c.gamma * 1e-7;
  return c.tf == 17 ? 1.0 / 2.6 : 0.0;
}
int TransferKind(const ColorEncoding& c) {   // 4: pure power law (PowerLawGamma)
  if (c.all_default) return 1;
  if (c.have_gamma) return 4;
  switch (c.tf) {
    case 8: return 0;
    case 13: return 1;
    case 1: return 2;
    case 16: return 3;
    case 17: return 4;
    default: return -1;
  }
}
static const float kPqM1 = 0.1593017578125f, kPqM2 = 78.84375f, kPqC1 = 0.8359375f, kPqC2 = 18.8515625f, kPqC3 = 18.6875f;
float EncodeTransfer(int kind, float v, float intensity_target, double gamma) {
  const float a = std::fabs(v);
  float r;
  switch (kind) {
    case 4: r = (float)std::pow((double)a, gamma); break;
    case 1: return LinearToSrgb(v);
    case 2: r = a < 0.018f ? 4.5f * a : 1.099f * std::pow(a, 0.45f) - 0.099f; break;
    case 3: {
      const float xp = std::pow(a * (intensity_target * 1e-4f), kPqM1);
      r = std::pow((kPqC1 + kPqC2 * xp) / (1.0f + kPqC3 * xp), kPqM2);
      if (a == 0.f) r = 0.f;   // the curve's value at 0 is 7.3e-7: keep black black
      break;
    }
    default: return v;
  }
  return std::copysign(r, v);
}
float DecodeTransfer(int kind, float e, float intensity_target, double gamma) {
  const float a = std::fabs(e);
  float r;
  switch (kind) {
    case 4: r = (float)std::pow((double)a, 1.0 / gamma); break;
    case 1: return SrgbToLinear(e);
    case 2: r = a < 0.081f ? a / 4.5f : std::pow((a + 0.099f) / 1.099f, 1.0f / 0.45f); break;
    case 3: {
      const float xp = std::pow(a, 1.0f / kPqM2);
      const float num = std::max(xp - kPqC1, 0.0f), den = kPqC2 - kPqC3 * xp;
      r = std::pow(num / den, 1.0f / kPqM1) * (1e4f / intensity_target);
      break;
    }
    default: return e;
  }
  return std::copysign(r, e);
}

static void Inv3(const double m[9], double o[9]) {
  const double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
  o[0] = (m[4] * m[8] - m[5] * m[7]) / d; o[1] = (m[2] * m[7] - m[1] * m[8]) / d; o[2] = (m[1] * m[5] - m[2] * m[4]) / d;
  o[3] = (m[5] * m[6] - m[3] * m[8]) / d; o[4] = (m[0] * m[8] - m[2] * m[6]) / d; o[5] = (m[2] * m[3] - m[0] * m[5]) / d;
  o[6] = (m[3] * m[7] - m[4] * m[6]) / d; o[7] = (m[1] * m[6] - m[0] * m[7]) / d; o[8] = (m[0] * m[4] - m[1] * m[3]) / d;
}
// RGB -> XYZ of a set of primaries with a D65 white point, from the chromaticities
static void RgbToXyz(const double xy[3][2], double m[9]) {
  const double wx = 0.3127, wy = 0.3290;
  double p[9];
  for (int c = 0; c < 3; c++) { p[c] = xy[c][0] / xy[c][1]; p[3 + c] = 1.0; p[6 + c] = (1.0 - xy[c][0] - xy[c][1]) / xy[c][1]; }
  double pi[9];
  Inv3(p, pi);
  const double W[3] = {wx / wy, 1.0, (1.0 - wx - wy) / wy};
  for (int c = 0; c < 3; c++) {
    const double sc = pi[c * 3] * W[0] + pi[c * 3 + 1] * W[1] + pi[c * 3 + 2] * W[2];
    for (int r = 0; r < 3; r++) m[r * 3 + c] = p[r * 3 + c] * sc;
  }
}
bool MatrixFromSrgb(uint32_t primaries, double out[9]) {
  static const double kSrgb[3][2] = {{0.639998686, 0.330010138}, {0.300003784, 0.600003357}, {0.150002046, 0.059997204}};
  static const double kP3[3][2] = {{0.680, 0.320}, {0.265, 0.690}, {0.150, 0.060}};
  static const double k2100[3][2] = {{0.708, 0.292}, {0.170, 0.797}, {0.131, 0.046}};
  const double (*t)[2] = primaries == 1 ? kSrgb : (primaries == 11 ? kP3 : (primaries == 9 ? k2100 : nullptr));
  if (!t) return false;
  if (primaries == 1) { for (int i = 0; i < 9; i++) out[i] = (i % 4 == 0) ? 1.0 : 0.0; return true; }
  double ms[9], mt[9], mti[9];
  RgbToXyz(kSrgb, ms);
  RgbToXyz(t, mt);
  Inv3(mt, mti);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) out[r * 3 + c] = mti[r * 3] * ms[c] + mti[r * 3 + 1] * ms[3 + c] + mti[r * 3 + 2] * ms[6 + c];
  return true;
}

// RGB (chromaticities + white) -> XYZ relative to D50: colorant matrix scaled to the white, then the von Kries step in Bradford's cone space
static void ToXyzD50(const double prim[3][2], const double white[2], double out[9]) {
  auto XYZ = [](const double* xy, double* v) { v[0] = xy[0] / xy[1]; v[1] = 1.0; v[2] = (1.0 - xy[0] - xy[1]) / xy[1]; };
  double P[9], Pi[9], W[3];
  for (int c = 0; c < 3; c++) { double v[3]; XYZ(prim[c], v); P[c] = v[0]; P[3 + c] = v[1]; P[6 + c] = v[2]; }
  XYZ(white, W);
  Inv3(P, Pi);
  double M[9];
  for (int c = 0; c < 3; c++) {
    const double s = Pi[c * 3] * W[0] + Pi[c * 3 + 1] * W[1] + Pi[c * 3 + 2] * W[2];
    for (int r = 0; r < 3; r++) M[r * 3 + c] = P[r * 3 + c] * s;
  }
  static const double kCone[9] = {0.8951, 0.2664, -0.1614, -0.7502, 1.7135, 0.0367, 0.0389, -0.0685, 1.0296};
  const double D50[3] = {0.96422, 1.0, 0.82521};
  double cw[3], cd[3], ci[9], diag[9], ad[9];
  for (int r = 0; r < 3; r++) {
    cw[r] = kCone[r * 3] * W[0] + kCone[r * 3 + 1] * W[1] + kCone[r * 3 + 2] * W[2];
    cd[r] = kCone[r * 3] * D50[0] + kCone[r * 3 + 1] * D50[1] + kCone[r * 3 + 2] * D50[2];
  }
  Inv3(kCone, ci);
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) diag[r * 3 + c] = kCone[r * 3 + c] * cd[r] / cw[r];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) ad[r * 3 + c] = ci[r * 3] * diag[c] + ci[r * 3 + 1] * diag[3 + c] + ci[r * 3 + 2] * diag[6 + c];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) out[r * 3 + c] = ad[r * 3] * M[c] + ad[r * 3 + 1] * M[3 + c] + ad[r * 3 + 2] * M[6 + c];
}

bool MatrixFromSrgbGeneral(const ColorEncoding& c, double out[9]) {
  if (c.all_default) return MatrixFromSrgb(1, out);
  double white[2] = {0.3127, 0.3290};
  if (c.white_point == 2) { white[0] = c.custom_xy[0][0] * 1e-6; white[1] = c.custom_xy[0][1] * 1e-6; }
  else if (c.white_point == 10) white[0] = white[1] = 1.0 / 3;
  else if (c.white_point == 11) { white[0] = 0.314; white[1] = 0.351; }
  else if (c.white_point != 1) return false;
  static const double kSrgb[3][2] = {{0.639998686, 0.330010138}, {0.300003784, 0.600003357}, {0.150002046, 0.059997204}};
  static const double kP3[3][2] = {{0.680, 0.320}, {0.265, 0.690}, {0.150, 0.060}};
  static const double k2100[3][2] = {{0.708, 0.292}, {0.170, 0.797}, {0.131, 0.046}};
  double prim[3][2];
  if (c.color_space == 1 || c.primaries == 1) memcpy(prim, kSrgb, sizeof(prim));
  else if (c.primaries == 11) memcpy(prim, kP3, sizeof(prim));
  else if (c.primaries == 9) memcpy(prim, k2100, sizeof(prim));
  else if (c.primaries == 2) for (int i = 0; i < 3; i++) { prim[i][0] = c.custom_xy[i + 1][0] * 1e-6; prim[i][1] = c.custom_xy[i + 1][1] * 1e-6; }
  else return false;
  const double d65[2] = {0.3127, 0.3290};
  double s[9], t[9], ti[9];
  ToXyzD50(kSrgb, d65, s);
  ToXyzD50(prim, white, t);
  Inv3(t, ti);
  for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) out[r * 3 + k] = ti[r * 3] * s[k] + ti[r * 3 + 1] * s[3 + k] + ti[r * 3 + 2] * s[6 + k];
  return true;
}

}  // namespace jxo
