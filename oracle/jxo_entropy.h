// ORACLE — test infrastructure only (see jxo_common.h header).
// Entropy coding of ISO/IEC 18181-1 Annex C: ANS / prefix codes, hybrid-uint, LZ77,
// context maps.  Reached in the reference only through libjxl
// (src/JxlFileTypeIO/Decoder/JxlDecoder.cpp:252); restated from the published format.
#pragma once
#include "jxo_common.h"

namespace jxo {

constexpr int kAnsLogTabSize = 12;
constexpr uint32_t kAnsTabSize = 1u << kAnsLogTabSize;
constexpr uint32_t kAnsSignature = 0x13;

struct HybridUintConfig {
  uint32_t split_exponent = 4, msb_in_token = 2, lsb_in_token = 0;
  HybridUintConfig() {}
  HybridUintConfig(uint32_t s, uint32_t m, uint32_t l) : split_exponent(s), msb_in_token(m), lsb_in_token(l) {}
  void Encode(uint32_t value, uint32_t* token, uint32_t* nbits, uint32_t* bits) const {
    uint32_t split = 1u << split_exponent;
    if (value < split) { *token = value; *nbits = 0; *bits = 0; return; }
    uint32_t n = FloorLog2(value);
    uint32_t m = value - (1u << n);
    *token = split + ((n - split_exponent) << (msb_in_token + lsb_in_token)) +
             ((m >> (n - msb_in_token)) << lsb_in_token) + (m & ((1u << lsb_in_token) - 1));
    *nbits = n - msb_in_token - lsb_in_token;
    *bits = (value >> lsb_in_token) & ((1ull << *nbits) - 1);
  }
};

struct AliasEntry {
  uint8_t cutoff;
  uint8_t right_value;
  uint16_t freq0;
  uint16_t offsets1;
  uint16_t freq1_xor_freq0;
};

struct PrefixCode {  // canonical Huffman, max length 15
  uint16_t count[16] = {0};
  std::vector<uint16_t> sorted;  // symbols sorted by (length, value)
  int single = -1;               // >=0: zero-bit code with that symbol
  std::vector<uint8_t> lengths;  // per symbol (encoder side / debugging)
};

struct EntropyCode {
  bool lz77 = false;
  uint32_t lz_min_symbol = 224, lz_min_length = 3;
  HybridUintConfig lz_len_cfg;
  std::vector<uint8_t> ctx_map;  // num_contexts (+1 when lz77) entries
  uint32_t num_hist = 1;
  bool use_prefix = false;
  uint32_t log_alpha = 8;
  std::vector<HybridUintConfig> cfg;            // per histogram
  std::vector<std::vector<AliasEntry>> alias;   // per histogram (ANS)
  std::vector<std::vector<int32_t>> counts;     // per histogram (ANS), kept for inspection
  std::vector<PrefixCode> prefix;               // per histogram (prefix)
};

uint32_t DecodeVarLenUint8(BitReader& br);
uint32_t DecodeVarLenUint16(BitReader& br);
void ReadHistogram(BitReader& br, std::vector<int32_t>& counts);
void InitAliasTable(std::vector<int32_t> dist, uint32_t log_alpha, std::vector<AliasEntry>& out);
void ReadPrefixCode(BitReader& br, uint32_t alphabet_size, PrefixCode& pc);
void DecodeContextMap(BitReader& br, std::vector<uint8_t>& map, uint32_t* num_hist);
void EncodeContextMap(BitWriter& bw, const std::vector<uint8_t>& map);   // a context map on its own (block-context map of LfGlobal)
void DecodeHistograms(BitReader& br, size_t num_contexts, EntropyCode& code, bool disallow_lz77 = false);

struct EntropyReader {
  const EntropyCode* code = nullptr;
  BitReader* br = nullptr;
  uint32_t state = 0;
  // LZ77
  std::vector<uint32_t> window;
  uint32_t num_to_copy = 0, copy_pos = 0, num_decoded = 0;
  uint32_t dist_multiplier = 0;
  uint32_t lz_ctx = 0;
  static constexpr uint32_t kWindowSize = 1u << 20, kWindowMask = kWindowSize - 1;

  void Init(const EntropyCode& c, BitReader& b, uint32_t dist_mult = 0);
  uint32_t ReadSymbol(uint32_t hist);
  inline uint32_t ReadHybrid(const HybridUintConfig& cfg, uint32_t token) {
    uint32_t split = 1u << cfg.split_exponent;
    if (token < split) return token;
    uint32_t nbits = cfg.split_exponent - (cfg.msb_in_token + cfg.lsb_in_token) +
                     ((token - split) >> (cfg.msb_in_token + cfg.lsb_in_token));
    JXO_CHECK(nbits <= 31, "hybrid uint too wide");
    uint32_t low = token & ((1u << cfg.lsb_in_token) - 1);
    token >>= cfg.lsb_in_token;
    uint32_t bits = br->Read(nbits);
    uint32_t hi = (1u << cfg.msb_in_token) | (token & ((1u << cfg.msb_in_token) - 1));
    return (uint32_t)(((((uint64_t)hi << nbits) | bits) << cfg.lsb_in_token) | low);
  }
  uint32_t Read(uint32_t ctx);
  bool CheckFinal() const { return code->use_prefix || state == (kAnsSignature << 16); }
};

// ---------------------------------------------------------------------------- encoder
struct Token {
  uint32_t ctx;
  uint32_t value;
  Token() {}
  Token(uint32_t c, uint32_t v) : ctx(c), value(v) {}
};

struct EncCode {
  EntropyCode code;                                   // what the decoder will see
  std::vector<std::vector<uint16_t>> reverse_map;     // per histogram: [sym_start + offset] -> slot
  std::vector<std::vector<uint32_t>> sym_start;       // per histogram: start index of symbol in reverse_map
  std::vector<std::vector<uint16_t>> pcode;           // prefix codes: per histogram, canonical code of each symbol (MSB first)
  size_t num_contexts = 0;                            // without the LZ77 distance context
};

struct EncOptions {
  HybridUintConfig cfg = HybridUintConfig(4, 2, 0);
  int max_clusters = 64;   // <= 255
  bool force_single_cluster = false;
  // Test streams (the reference's encoder library chooses these by effort; the oracle writes them on request so that decoders have
  // something to read): prefix codes instead of ANS; LZ77 with runs (distance 1) and copies from `dist_mult` symbols back.
  bool use_prefix = false;
  bool lz77 = false;
  bool top_level = true;   // false for the nested codes of context maps: never switched by SetEntropyTestMode
};
// Process-wide switch for the top-level codes the encoder writes from now on (bit 0: prefix codes, bit 1: LZ77).
void SetEntropyTestMode(uint32_t mode);

// Build a code (clustered ANS histograms) for `num_contexts` from all tokens of a stream
// family, write its header; then write token sections with WriteTokens.
void BuildAndWriteCode(const std::vector<const std::vector<Token>*>& token_sets, size_t num_contexts,
                       const EncOptions& opt, BitWriter& bw, EncCode& out);
void WriteTokens(const std::vector<Token>& tokens, const EncCode& ec, BitWriter& bw);
// The LZ77 distance multiplier the decoder will use for a token stream (0, the default: one-dimensional streams such as the HF
// coefficients; Modular streams: the largest channel width of the sub-image).  Registered per token vector (by address) before the
// code is built, because the choice of copies - hence the histograms - depends on it.
void SetStreamDistMult(const std::vector<Token>* tokens, uint32_t dist_mult);
void ClearStreamDistMults();   // at every frame boundary of the encoder

}  // namespace jxo
