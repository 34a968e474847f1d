// ORACLE — test infrastructure only.
//
// CPU restatement of the JPEG XL codestream algorithms that the reference
// (0xC0000054/pdn-jpegxl) reaches through libjxl at
//   src/JxlFileTypeIO/Decoder/JxlDecoder.cpp:252   (JxlDecoderProcessInput)
//   src/JxlFileTypeIO/Encoder/JxlEncoder.cpp:128,367 (JxlEncoderAddImageFrame / FlushInput)
// libjxl is an un-vendored vcpkg dependency (src/JxlFileTypeIO/vcpkg.json:6-9) that is
// absent from the reference tree and from this container, and the reference holds no
// tests, fixtures or .jxl files => PARITY WITH LIBJXL IS UNPINNED.  Everything here
// restates ISO/IEC 18181-1 as published ("Oracle"), and is pinned only by
// self-consistency (encode->decode round trips, float64 closed forms).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this
// directory.  The product (pdn_jpegxl_amd/csrc) never includes or links it.
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>
#include <algorithm>

namespace jxo {

struct Error : std::runtime_error {
  explicit Error(const std::string& m) : std::runtime_error(m) {}
};
#define JXO_CHECK(cond, msg) \
  do { if (!(cond)) throw ::jxo::Error(std::string(msg) + " [" #cond "]"); } while (0)

static inline int FloorLog2(uint64_t x) { return 63 - __builtin_clzll(x | 1); }
static inline int CeilLog2(uint64_t x) { return x <= 1 ? 0 : FloorLog2(x - 1) + 1; }
static inline size_t DivCeil(size_t a, size_t b) { return (a + b - 1) / b; }
static inline int64_t UnpackSigned(uint64_t u) { return (int64_t)(u >> 1) ^ -(int64_t)(u & 1); }
static inline uint64_t PackSigned(int64_t s) { return ((uint64_t)s << 1) ^ (uint64_t)(s >> 63); }

// ---------------------------------------------------------------- bit reader (LSB first)
struct BitReader {
  const uint8_t* data = nullptr;
  size_t size = 0;      // bytes
  size_t pos = 0;       // bit position
  bool overrun = false;
  BitReader() {}
  BitReader(const uint8_t* d, size_t n, size_t bitpos = 0) : data(d), size(n), pos(bitpos) {}
  inline uint64_t Peek(int n) {  // n <= 56
    size_t byte = pos >> 3;
    uint64_t v = 0;
    if (byte + 8 <= size) {
      memcpy(&v, data + byte, 8);
    } else {
      for (size_t i = 0; i < 8 && byte + i < size; i++) v |= (uint64_t)data[byte + i] << (8 * i);
    }
    v >>= (pos & 7);
    return n >= 64 ? v : (v & ((1ull << n) - 1));
  }
  inline void Skip(size_t n) {
    pos += n;
    if (pos > size * 8) overrun = true;
  }
  inline uint32_t Read(int n) {
    if (n == 0) return 0;
    uint64_t v = Peek(n);
    Skip(n);
    return (uint32_t)v;
  }
  inline bool Bool() { return Read(1) != 0; }
  void AlignByte() { Skip((8 - (pos & 7)) & 7); }
  // U32 with 4 distributions: each (bits, offset); bits<0 => constant offset.
  struct D { int bits; uint32_t off; };
  uint32_t U32(D d0, D d1, D d2, D d3) {
    D d[4] = {d0, d1, d2, d3};
    uint32_t s = Read(2);
    return d[s].bits < 0 ? d[s].off : d[s].off + Read(d[s].bits);
  }
  uint64_t U64() {
    uint32_t sel = Read(2);
    if (sel == 0) return 0;
    if (sel == 1) return 1 + Read(4);
    if (sel == 2) return 17 + Read(8);
    uint64_t v = Read(12);
    int shift = 12;
    while (Read(1)) {
      if (shift == 60) { v |= (uint64_t)Read(4) << shift; break; }
      v |= (uint64_t)Read(8) << shift;
      shift += 8;
    }
    return v;
  }
  float F16() {
    uint32_t b = Read(16);
    uint32_t sign = b >> 15, e = (b >> 10) & 31, m = b & 1023;
    float v;
    if (e == 0) v = std::ldexp((float)m, -24);
    else if (e == 31) throw Error("F16 inf/nan");
    else v = std::ldexp((float)(m + 1024), (int)e - 25);
    return sign ? -v : v;
  }
  uint32_t Enum() { return U32({-1, 0}, {-1, 1}, {4, 2}, {6, 18}); }
};
static inline BitReader::D Val(uint32_t v) { return {-1, v}; }
static inline BitReader::D Bits(int n) { return {n, 0}; }
static inline BitReader::D BitsOff(int n, uint32_t o) { return {n, o}; }

// ---------------------------------------------------------------- bit writer (LSB first)
struct BitWriter {
  std::vector<uint8_t> buf;
  size_t bitpos = 0;
  void Write(int n, uint64_t v) {  // n <= 56
    if (n == 0) return;
    JXO_CHECK(n <= 56, "BitWriter width");
    JXO_CHECK(n == 64 || (v >> n) == 0, "BitWriter value too wide");
    size_t need = (bitpos + n + 7) / 8;
    if (buf.size() < need + 8) buf.resize(need + 8 + buf.size() / 2, 0);
    size_t byte = bitpos >> 3;
    uint64_t cur;
    memcpy(&cur, &buf[byte], 8);
    cur |= v << (bitpos & 7);
    memcpy(&buf[byte], &cur, 8);
    if ((bitpos & 7) + n > 64) buf[byte + 8] |= (uint8_t)(v >> (64 - (bitpos & 7)));
    bitpos += n;
  }
  void Bool(bool b) { Write(1, b ? 1 : 0); }
  void AlignByte() { Write((8 - (bitpos & 7)) & 7, 0); }
  size_t Bytes() const { return (bitpos + 7) / 8; }
  std::vector<uint8_t> Finish() {
    AlignByte();
    std::vector<uint8_t> out(buf.begin(), buf.begin() + bitpos / 8);
    return out;
  }
  // Append the bits of another writer.
  void Append(const BitWriter& o) {
    size_t full = o.bitpos / 8;
    for (size_t i = 0; i < full; i++) Write(8, o.buf[i]);
    int rem = (int)(o.bitpos & 7);
    if (rem) Write(rem, o.buf[full] & ((1u << rem) - 1));
  }
  void AppendBytes(const uint8_t* p, size_t n) {
    JXO_CHECK((bitpos & 7) == 0, "AppendBytes unaligned");
    for (size_t i = 0; i < n; i++) Write(8, p[i]);
  }
  // U32 writer: picks the first distribution that can represent v.
  void U32(BitReader::D d0, BitReader::D d1, BitReader::D d2, BitReader::D d3, uint32_t v) {
    BitReader::D d[4] = {d0, d1, d2, d3};
    for (int s = 0; s < 4; s++) {
      if (d[s].bits < 0) {
        if (d[s].off == v) { Write(2, s); return; }
      } else if (v >= d[s].off && (uint64_t)(v - d[s].off) < (1ull << d[s].bits)) {
        Write(2, s);
        Write(d[s].bits, v - d[s].off);
        return;
      }
    }
    throw Error("U32 value not representable");
  }
  void U64(uint64_t v) {
    if (v == 0) { Write(2, 0); return; }
    if (v <= 16) { Write(2, 1); Write(4, v - 1); return; }
    if (v <= 272) { Write(2, 2); Write(8, v - 17); return; }
    Write(2, 3);
    Write(12, v & 4095);
    v >>= 12;
    int shift = 12;
    while (v) {
      Write(1, 1);
      if (shift == 60) { Write(4, v & 15); return; }
      Write(8, v & 255);
      v >>= 8;
      shift += 8;
    }
    Write(1, 0);
  }
  void F16(float f) {
    // exact for values representable in half precision; otherwise round to nearest.
    uint32_t sign = f < 0 ? 1 : 0;
    float a = std::fabs(f);
    uint32_t bits;
    if (a == 0) bits = 0;
    else {
      int e;
      float m = std::frexp(a, &e);  // a = m * 2^e, m in [0.5,1)
      int he = e + 14;              // half exponent field
      if (he <= 0) {
        uint32_t mant = (uint32_t)std::lrint(std::ldexp(a, 24));
        bits = mant;  // subnormal (may round up to the smallest normal, still correct)
      } else {
        uint32_t mant = (uint32_t)std::lrint(std::ldexp(m, 11)) - 1024;
        if (mant == 1024) { mant = 0; he++; }
        JXO_CHECK(he < 31, "F16 overflow");
        bits = ((uint32_t)he << 10) | mant;
      }
    }
    Write(16, bits | (sign << 15));
  }
  void Enum(uint32_t v) { U32(Val(0), Val(1), BitsOff(4, 2), BitsOff(6, 18), v); }
};

static inline float RoundToF16(float f) {
  BitWriter w;
  w.F16(f);
  BitReader r(w.buf.data(), w.buf.size());
  return r.F16();
}

}  // namespace jxo
