// Small device-side helpers shared by the kernel files (included after <hip/hip_runtime.h>).
#pragma once
#include <stdint.h>

namespace jxlhip {

// Integer sample of `bits` bits -> output sample (8 or 16 bits).  Equal depths pass through clamped; otherwise through a [0, 1]
// float, like the decoder library behind the reference does for every channel whose depth differs from the output type's.
__device__ __forceinline__ uint32_t IntToOutSample(int32_t v, int bits, int out_bits) {
  const int32_t maxv = (int32_t)((1u << bits) - 1);
  if (bits == out_bits) return (uint32_t)min(maxv, max(0, v));
  float f = (float)v * (1.0f / (float)maxv);
  f *= out_bits == 16 ? 65535.0f : 255.0f;
  const float top = out_bits == 16 ? 65535.0f : 255.0f;
  if (!(f > 0.f)) return 0;
  if (f >= top) return (uint32_t)top;
  return (uint32_t)(f + 0.5f);
}

}  // namespace jxlhip
