// Small device-side helpers shared by the kernel files (included after <hip/hip_runtime.h>).
#pragma once
#include <stdint.h>
#include <hip/hip_fp16.h>

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

// Bit pattern of a float-coded Modular sample (binary32 as is, binary16 widened) -> float
__device__ __forceinline__ float BitsToFloatSample(int32_t v, int bits) {
  if (bits == 32) return __int_as_float(v);
  return __half2float(__ushort_as_half((unsigned short)(v & 0xFFFF)));
}
__device__ __forceinline__ uint32_t FloatToOutBits(float f, int out_bits, int out_float) {
  if (out_float) return out_bits == 32 ? (uint32_t)__float_as_int(f) : (uint32_t)__half_as_ushort(__float2half_rn(f));
  const float top = out_bits == 16 ? 65535.0f : 255.0f;
  f *= top;
  if (!(f > 0.f)) return 0;
  if (f >= top) return (uint32_t)top;
  return (uint32_t)(f + 0.5f);
}
// One channel sample (integer of `bits` bits, or a float bit pattern when exp_bits > 0) -> raw bits of the output sample type
__device__ __forceinline__ uint32_t SampleToOutBits(int32_t v, int bits, int exp_bits, int out_bits, int out_float) {
  if (!out_float && !exp_bits) return IntToOutSample(v, bits, out_bits);
  const float f = exp_bits ? BitsToFloatSample(v, bits) : (float)v * (1.0f / (float)((1u << bits) - 1));
  return FloatToOutBits(f, out_bits, out_float);
}
__device__ __forceinline__ void StoreOutSample(uint8_t* base, size_t index, uint32_t raw, int out_bits) {
  if (out_bits == 8) base[index] = (uint8_t)raw;
  else if (out_bits == 16) ((uint16_t*)base)[index] = (uint16_t)raw;
  else ((uint32_t*)base)[index] = raw;
}

}  // namespace jxlhip
