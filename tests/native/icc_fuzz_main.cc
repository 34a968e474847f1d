// CPU AddressSanitizer harness for the product's ICC stream reader (csrc/icc.cc alone; no GPU code).  Built and run by
// tests/test_icc.py::test_icc_unpredict_under_asan: reads seed streams from the files named on the command line, feeds the reader
// byte mutations of each (seeded xorshift) plus hand-made hostile varints, and exits 0 when no run crashed; ASan aborts otherwise.
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include "../../pdn_jpegxl_amd/csrc/icc.h"

static uint64_t g_s = 0x9E3779B97F4A7C15ull;
static uint32_t Rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return (uint32_t)(g_s >> 16); }

static void Varint(uint64_t v, std::vector<uint8_t>* o) {
  while (v >= 128) { o->push_back((uint8_t)(v | 128)); v >>= 7; }
  o->push_back((uint8_t)v);
}

int main(int argc, char** argv) {
  size_t runs = 0, accepted = 0;
  std::vector<uint8_t> icc;
  std::string why;
  // hostile predictor strides after a plain 128-byte header: {insert 4 bytes, predictor with a stride of 2^62 / 2^63 - 1 / ...}
  for (uint64_t stride : {1ull << 62, (1ull << 63) - 1, (1ull << 62) + 1, 1ull << 61, 0xFFFFFFFFull, 1ull << 32, 33ull, 34ull}) {
    std::vector<uint8_t> cmd, data(128, 0);
    Varint(0, &cmd);                       // no tag table
    cmd.push_back(1); Varint(4, &cmd);     // insert 4
    data.insert(data.end(), {1, 2, 3, 4});
    cmd.push_back(4); cmd.push_back(0x10); Varint(stride, &cmd); Varint(4, &cmd);
    data.insert(data.end(), {0, 0, 0, 0});
    std::vector<uint8_t> enc;
    Varint(136, &enc); Varint(cmd.size(), &enc);
    enc.insert(enc.end(), cmd.begin(), cmd.end());
    enc.insert(enc.end(), data.begin(), data.end());
    accepted += jxlhip::IccUnpredict(enc, &icc, &why) ? 1 : 0;
    runs++;
  }
  for (int a = 1; a < argc; a++) {
    FILE* f = fopen(argv[a], "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[a]); return 2; }
    std::vector<uint8_t> seed;
    for (int c; (c = fgetc(f)) != EOF;) seed.push_back((uint8_t)c);
    fclose(f);
    if (!jxlhip::IccUnpredict(seed, &icc, &why)) { fprintf(stderr, "seed %s refused: %s\n", argv[a], why.c_str()); return 3; }
    for (int it = 0; it < 4000; it++) {
      std::vector<uint8_t> m = seed;
      const int nmut = 1 + Rnd() % 4;
      for (int k = 0; k < nmut; k++) {
        const size_t lim = (Rnd() & 1) ? (m.size() < 96 ? m.size() : 96) : m.size();   // half of the mutations in the command area
        const size_t at = Rnd() % lim;
        switch (Rnd() % 4) {
          case 0: m[at] = (uint8_t)Rnd(); break;
          case 1: m[at] ^= (uint8_t)(1u << (Rnd() % 8)); break;
          case 2: m[at] = 0xFF; break;                       // long varints
          default: if (m.size() > 2) m.erase(m.begin() + at); break;
        }
      }
      accepted += jxlhip::IccUnpredict(m, &icc, &why) ? 1 : 0;
      runs++;
    }
  }
  printf("icc fuzz: %zu runs, %zu accepted, no crash\n", runs, accepted);
  return 0;
}
