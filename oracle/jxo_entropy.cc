// ORACLE — test infrastructure only (see jxo_common.h header).
#include "jxo_entropy.h"
#include <map>
#include <mutex>
#include <numeric>

namespace jxo {

uint32_t DecodeVarLenUint8(BitReader& br) {
  if (!br.Read(1)) return 0;
  uint32_t n = br.Read(3);
  return br.Read(n) + (1u << n);
}
uint32_t DecodeVarLenUint16(BitReader& br) {
  if (!br.Read(1)) return 0;
  uint32_t n = br.Read(4);
  return br.Read(n) + (1u << n);
}

static int PopulationCountPrecision(int logcount, int shift) {
  int r = std::min(logcount, shift - ((kAnsLogTabSize - logcount) >> 1));
  return r < 0 ? 0 : r;
}

// Fixed prefix code for the log-counts of an ANS distribution: (length, symbol) by 7-bit peek.
static const uint8_t kLogCountLut[128][2] = {
    {3, 10}, {7, 12}, {3, 7}, {4, 3}, {3, 6}, {3, 8}, {3, 9}, {4, 5}, {3, 10}, {4, 4}, {3, 7}, {4, 1}, {3, 6}, {3, 8}, {3, 9}, {4, 2},
    {3, 10}, {5, 0},  {3, 7}, {4, 3}, {3, 6}, {3, 8}, {3, 9}, {4, 5}, {3, 10}, {4, 4}, {3, 7}, {4, 1}, {3, 6}, {3, 8}, {3, 9}, {4, 2},
    {3, 10}, {6, 11}, {3, 7}, {4, 3}, {3, 6}, {3, 8}, {3, 9}, {4, 5}, {3, 10}, {4, 4}, {3, 7}, {4, 1}, {3, 6}, {3, 8}, {3, 9}, {4, 2},
    {3, 10}, {5, 0},  {3, 7}, {4, 3}, {3, 6}, {3, 8}, {3, 9}, {4, 5}, {3, 10}, {4, 4}, {3, 7}, {4, 1}, {3, 6}, {3, 8}, {3, 9}, {4, 2},
    {3, 10}, {7, 13}, {3, 7}, {4, 3}, {3, 6}, {3, 8}, {3, 9}, {4, 5}, {3, 10}, {4, 4}, {3, 7}, {4, 1}, {3, 6}, {3, 8}, {3, 9}, {4, 2},
    {3, 10}, {5, 0},  {3, 7}, {4, 3}, {3, 6}, {3, 8}, {3, 9}, {4, 5}, {3, 10}, {4, 4}, {3, 7}, {4, 1}, {3, 6}, {3, 8}, {3, 9}, {4, 2},
    {3, 10}, {6, 11}, {3, 7}, {4, 3}, {3, 6}, {3, 8}, {3, 9}, {4, 5}, {3, 10}, {4, 4}, {3, 7}, {4, 1}, {3, 6}, {3, 8}, {3, 9}, {4, 2},
    {3, 10}, {5, 0},  {3, 7}, {4, 3}, {3, 6}, {3, 8}, {3, 9}, {4, 5}, {3, 10}, {4, 4}, {3, 7}, {4, 1}, {3, 6}, {3, 8}, {3, 9}, {4, 2},
};

void ReadHistogram(BitReader& br, std::vector<int32_t>& counts) {
  counts.clear();
  if (br.Read(1)) {  // simple
    int num_symbols = br.Read(1) + 1;
    int symbols[2] = {0, 0};
    int max_symbol = 0;
    for (int i = 0; i < num_symbols; i++) {
      symbols[i] = DecodeVarLenUint8(br);
      max_symbol = std::max(max_symbol, symbols[i]);
    }
    counts.assign(max_symbol + 1, 0);
    if (num_symbols == 1) {
      counts[symbols[0]] = kAnsTabSize;
    } else {
      JXO_CHECK(symbols[0] != symbols[1], "simple histogram: duplicate symbols");
      counts[symbols[0]] = br.Read(kAnsLogTabSize);
      counts[symbols[1]] = kAnsTabSize - counts[symbols[0]];
    }
    return;
  }
  if (br.Read(1)) {  // flat
    int alphabet_size = DecodeVarLenUint8(br) + 1;
    counts.assign(alphabet_size, kAnsTabSize / alphabet_size);
    for (int i = 0; i < (int)(kAnsTabSize % alphabet_size); i++) counts[i]++;
    return;
  }
  int upper_bound_log = FloorLog2(kAnsLogTabSize + 1);
  int log = 0;
  for (; log < upper_bound_log; log++)
    if (br.Read(1) == 0) break;
  int shift = (int)((br.Read(log) | (1u << log)) - 1);
  JXO_CHECK(shift <= kAnsLogTabSize + 1, "histogram shift");
  int length = DecodeVarLenUint8(br) + 3;
  counts.assign(length, 0);
  std::vector<int> logcounts(length, 0), same(length, 0);
  int omit_log = -1, omit_pos = -1;
  for (int i = 0; i < length; i++) {
    uint32_t idx = (uint32_t)br.Peek(7);
    br.Skip(kLogCountLut[idx][0]);
    logcounts[i] = kLogCountLut[idx][1];
    if (logcounts[i] == kAnsLogTabSize + 1) {
      int rle_length = DecodeVarLenUint8(br);
      same[i] = rle_length + 5;
      i += rle_length + 3;
      continue;
    }
    if (logcounts[i] > omit_log) { omit_log = logcounts[i]; omit_pos = i; }
  }
  JXO_CHECK(omit_pos >= 0, "invalid histogram (rle)");
  JXO_CHECK(!(omit_pos + 1 < length && logcounts[omit_pos + 1] == kAnsLogTabSize + 1), "invalid histogram (rle after omit)");
  int prev = 0, numsame = 0, total = 0;
  for (int i = 0; i < length; i++) {
    if (same[i]) {
      numsame = same[i] - 1;
      prev = i > 0 ? counts[i - 1] : 0;
    }
    if (numsame > 0) {
      counts[i] = prev;
      numsame--;
    } else {
      int code = logcounts[i];
      if (i == omit_pos || code == 0) continue;
      if (code == 1) counts[i] = 1;
      else {
        int bitcount = PopulationCountPrecision(code - 1, shift);
        counts[i] = (1 << (code - 1)) + (br.Read(bitcount) << (code - 1 - bitcount));
      }
    }
    total += counts[i];
  }
  counts[omit_pos] = (int)kAnsTabSize - total;
  JXO_CHECK(counts[omit_pos] > 0, "invalid histogram count");
}

void InitAliasTable(std::vector<int32_t> dist, uint32_t log_alpha, std::vector<AliasEntry>& a) {
  while (!dist.empty() && dist.back() == 0) dist.pop_back();
  if (dist.empty()) dist.push_back(kAnsTabSize);
  const size_t table_size = (size_t)1 << log_alpha;
  JXO_CHECK(dist.size() <= table_size, "alphabet larger than alias table");
  const uint32_t entry_size = kAnsTabSize >> log_alpha;
  a.assign(table_size, AliasEntry{0, 0, 0, 0, 0});
  for (size_t sym = 0; sym < dist.size(); sym++) {
    if ((uint32_t)dist[sym] == kAnsTabSize) {
      for (size_t i = 0; i < table_size; i++) {
        a[i].right_value = (uint8_t)sym;
        a[i].cutoff = 0;
        a[i].offsets1 = (uint16_t)(entry_size * i);
        a[i].freq0 = 0;
        a[i].freq1_xor_freq0 = (uint16_t)kAnsTabSize;
      }
      return;
    }
  }
  std::vector<uint32_t> underfull, overfull, cutoffs(table_size, 0);
  for (size_t i = 0; i < dist.size(); i++) {
    cutoffs[i] = dist[i];
    if (cutoffs[i] > entry_size) overfull.push_back((uint32_t)i);
    else if (cutoffs[i] < entry_size) underfull.push_back((uint32_t)i);
  }
  for (size_t i = dist.size(); i < table_size; i++) underfull.push_back((uint32_t)i);
  while (!overfull.empty()) {
    uint32_t o = overfull.back(); overfull.pop_back();
    JXO_CHECK(!underfull.empty(), "alias table construction");
    uint32_t u = underfull.back(); underfull.pop_back();
    uint32_t by = entry_size - cutoffs[u];
    cutoffs[o] -= by;
    a[u].right_value = (uint8_t)o;
    a[u].offsets1 = (uint16_t)cutoffs[o];
    if (cutoffs[o] < entry_size) underfull.push_back(o);
    else if (cutoffs[o] > entry_size) overfull.push_back(o);
  }
  for (size_t i = 0; i < table_size; i++) {
    if (cutoffs[i] == entry_size) {
      a[i].right_value = (uint8_t)i;
      a[i].offsets1 = 0;
      a[i].cutoff = 0;
    } else {
      a[i].offsets1 = (uint16_t)(a[i].offsets1 - cutoffs[i]);
      a[i].cutoff = (uint8_t)cutoffs[i];
    }
    uint32_t freq0 = i < dist.size() ? dist[i] : 0;
    uint32_t i1 = a[i].right_value;
    uint32_t freq1 = i1 < dist.size() ? dist[i1] : 0;
    a[i].freq0 = (uint16_t)freq0;
    a[i].freq1_xor_freq0 = (uint16_t)(freq1 ^ freq0);
  }
}

// ------------------------------------------------------------------ prefix codes (RFC 7932 3.4/3.5)
static void BuildPrefix(const std::vector<uint8_t>& lengths, PrefixCode& pc) {
  pc.lengths = lengths;
  memset(pc.count, 0, sizeof(pc.count));
  int nonzero = 0, last = -1;
  for (size_t i = 0; i < lengths.size(); i++)
    if (lengths[i]) { pc.count[lengths[i]]++; nonzero++; last = (int)i; }
  pc.sorted.clear();
  pc.single = -1;
  if (nonzero == 0) { pc.single = 0; return; }
  if (nonzero == 1) { pc.single = last; return; }
  for (int len = 1; len <= 15; len++)
    for (size_t i = 0; i < lengths.size(); i++)
      if (lengths[i] == len) pc.sorted.push_back((uint16_t)i);
}

static inline uint32_t ReadPrefixSymbol(BitReader& br, const PrefixCode& pc) {
  if (pc.single >= 0) return pc.single;
  int code = 0, first = 0, index = 0;
  for (int len = 1; len <= 15; len++) {
    code |= br.Read(1);
    int count = pc.count[len];
    if (code - count < first) return pc.sorted[index + (code - first)];
    index += count;
    first += count;
    first <<= 1;
    code <<= 1;
  }
  throw Error("invalid prefix code");
}

void ReadPrefixCode(BitReader& br, uint32_t alphabet_size, PrefixCode& pc) {
  std::vector<uint8_t> lengths(alphabet_size, 0);
  if (alphabet_size == 1) { BuildPrefix(lengths, pc); pc.single = 0; return; }
  uint32_t hskip = br.Read(2);
  if (hskip == 1) {
    int max_bits = 0;
    for (uint32_t c = alphabet_size - 1; c; c >>= 1) max_bits++;
    uint32_t nsym = br.Read(2) + 1;
    uint32_t syms[4];
    for (uint32_t i = 0; i < nsym; i++) {
      syms[i] = br.Read(max_bits);
      JXO_CHECK(syms[i] < alphabet_size, "simple prefix code symbol out of range");
    }
    for (uint32_t i = 0; i < nsym; i++)
      for (uint32_t j = i + 1; j < nsym; j++) JXO_CHECK(syms[i] != syms[j], "duplicate prefix symbols");
    if (nsym == 1) {
      BuildPrefix(lengths, pc);
      pc.single = syms[0];
      return;
    } else if (nsym == 2) {
      lengths[syms[0]] = lengths[syms[1]] = 1;
    } else if (nsym == 3) {
      lengths[syms[0]] = 1; lengths[syms[1]] = lengths[syms[2]] = 2;
    } else {
      if (br.Read(1)) { lengths[syms[0]] = 1; lengths[syms[1]] = 2; lengths[syms[2]] = lengths[syms[3]] = 3; }
      else lengths[syms[0]] = lengths[syms[1]] = lengths[syms[2]] = lengths[syms[3]] = 2;
    }
    BuildPrefix(lengths, pc);
    return;
  }
  static const uint8_t kOrder[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
  static const uint8_t kLen[16] = {2, 2, 2, 3, 2, 2, 2, 4, 2, 2, 2, 3, 2, 2, 2, 4};
  static const uint8_t kVal[16] = {0, 4, 3, 2, 0, 4, 3, 1, 0, 4, 3, 2, 0, 4, 3, 5};
  std::vector<uint8_t> cl(18, 0);
  int space = 32, num_codes = 0;
  for (int i = hskip; i < 18 && space > 0; i++) {
    uint32_t p = (uint32_t)br.Peek(4);
    br.Skip(kLen[p]);
    uint8_t v = kVal[p];
    cl[kOrder[i]] = v;
    if (v) { space -= 32 >> v; num_codes++; }
  }
  JXO_CHECK(num_codes == 1 || space == 0, "invalid code length code");
  PrefixCode clc;
  BuildPrefix(cl, clc);
  uint32_t symbol = 0;
  int prev_len = 8, repeat = 0, repeat_len = 0;
  int32_t sp = 32768;
  while (symbol < alphabet_size && sp > 0) {
    uint32_t code_len = ReadPrefixSymbol(br, clc);
    if (code_len < 16) {
      repeat = 0;
      lengths[symbol++] = (uint8_t)code_len;
      if (code_len) { prev_len = code_len; sp -= 32768 >> code_len; }
    } else {
      int extra = code_len == 16 ? 2 : 3;
      int new_len = code_len == 16 ? prev_len : 0;
      if (repeat_len != new_len) { repeat = 0; repeat_len = new_len; }
      int old = repeat;
      if (repeat > 0) { repeat -= 2; repeat <<= extra; }
      repeat += br.Read(extra) + 3;
      int delta = repeat - old;
      JXO_CHECK(symbol + delta <= alphabet_size, "prefix code repeat overflow");
      for (int i = 0; i < delta; i++) lengths[symbol++] = (uint8_t)repeat_len;
      if (repeat_len) sp -= delta << (15 - repeat_len);
    }
  }
  JXO_CHECK(sp == 0, "prefix code space");
  BuildPrefix(lengths, pc);
}

// ------------------------------------------------------------------ context map / histograms
static HybridUintConfig ReadUintConfig(BitReader& br, uint32_t log_alpha) {
  HybridUintConfig c;
  c.split_exponent = br.Read(CeilLog2(log_alpha + 1));
  c.msb_in_token = c.lsb_in_token = 0;
  JXO_CHECK(c.split_exponent <= log_alpha, "split_exponent");
  if (c.split_exponent != log_alpha) {
    c.msb_in_token = br.Read(CeilLog2(c.split_exponent + 1));
    JXO_CHECK(c.msb_in_token <= c.split_exponent, "msb_in_token");
    c.lsb_in_token = br.Read(CeilLog2(c.split_exponent - c.msb_in_token + 1));
    JXO_CHECK(c.lsb_in_token + c.msb_in_token <= c.split_exponent, "lsb_in_token");
  }
  return c;
}

void DecodeContextMap(BitReader& br, std::vector<uint8_t>& map, uint32_t* num_hist) {
  bool is_simple = br.Bool();
  if (is_simple) {
    int bits = br.Read(2);
    for (auto& m : map) m = (uint8_t)br.Read(bits);
  } else {
    bool use_mtf = br.Bool();
    EntropyCode code;
    DecodeHistograms(br, 1, code, map.size() <= 2);
    EntropyReader rd;
    rd.Init(code, br);
    for (auto& m : map) {
      uint32_t v = rd.Read(0);
      JXO_CHECK(v < 256, "context map entry");
      m = (uint8_t)v;
    }
    JXO_CHECK(rd.CheckFinal(), "context map ANS final state");
    if (use_mtf) {
      uint8_t mtf[256];
      for (int i = 0; i < 256; i++) mtf[i] = (uint8_t)i;
      for (auto& m : map) {
        uint8_t idx = m, v = mtf[idx];
        m = v;
        for (; idx; idx--) mtf[idx] = mtf[idx - 1];
        mtf[0] = v;
      }
    }
  }
  uint32_t mx = 0;
  for (auto m : map) mx = std::max<uint32_t>(mx, m);
  *num_hist = mx + 1;
}

void EncodeContextMap(BitWriter& bw, const std::vector<uint8_t>& map) {
  uint32_t num_hist = 0;
  for (auto m : map) num_hist = std::max<uint32_t>(num_hist, m + 1u);
  if (num_hist <= 8 && map.size() <= 64) {
    int bits = num_hist <= 1 ? 0 : CeilLog2(num_hist);
    bw.Write(1, 1);
    bw.Write(2, bits);
    for (auto m : map) bw.Write(bits, m);
    return;
  }
  bw.Write(1, 0);  // not simple
  bw.Write(1, 0);  // no move-to-front
  std::vector<Token> mt;
  for (auto m : map) mt.emplace_back(0, m);
  EncOptions mo;
  mo.cfg = HybridUintConfig(4, 2, 0);
  mo.top_level = false;
  EncCode mc;
  std::vector<const std::vector<Token>*> sets = {&mt};
  BuildAndWriteCode(sets, 1, mo, bw, mc);
  WriteTokens(mt, mc, bw);
}

void DecodeHistograms(BitReader& br, size_t num_contexts, EntropyCode& code, bool disallow_lz77) {
  code = EntropyCode();
  code.lz77 = br.Bool();
  if (code.lz77) {
    JXO_CHECK(!disallow_lz77, "lz77 not allowed here");
    code.lz_min_symbol = br.U32(Val(224), Val(512), Val(4096), BitsOff(15, 8));
    code.lz_min_length = br.U32(Val(3), Val(4), BitsOff(2, 5), BitsOff(8, 9));
    code.lz_len_cfg = ReadUintConfig(br, 8);
    num_contexts++;
  }
  code.ctx_map.assign(num_contexts, 0);
  code.num_hist = 1;
  if (num_contexts > 1) DecodeContextMap(br, code.ctx_map, &code.num_hist);
  code.use_prefix = br.Bool();
  code.log_alpha = code.use_prefix ? 15 : br.Read(2) + 5;
  code.cfg.resize(code.num_hist);
  for (auto& c : code.cfg) c = ReadUintConfig(br, code.log_alpha);
  if (code.use_prefix) {
    std::vector<uint32_t> asz(code.num_hist);
    for (auto& a : asz) {
      a = DecodeVarLenUint16(br) + 1;
      JXO_CHECK(a <= (1u << 15), "prefix alphabet size");
    }
    code.prefix.resize(code.num_hist);
    for (uint32_t i = 0; i < code.num_hist; i++) ReadPrefixCode(br, asz[i], code.prefix[i]);
  } else {
    code.alias.resize(code.num_hist);
    code.counts.resize(code.num_hist);
    for (uint32_t i = 0; i < code.num_hist; i++) {
      ReadHistogram(br, code.counts[i]);
      JXO_CHECK(code.counts[i].size() <= (1u << code.log_alpha), "histogram alphabet too large");
      InitAliasTable(code.counts[i], code.log_alpha, code.alias[i]);
    }
  }
  JXO_CHECK(!br.overrun, "truncated entropy code header");
}

// ------------------------------------------------------------------ reader
static const int8_t kSpecialDistances[120][2] = {
    {0, 1},  {1, 0},  {1, 1},  {-1, 1}, {0, 2},  {2, 0},  {1, 2},  {-1, 2}, {2, 1},  {-2, 1}, {2, 2},  {-2, 2}, {0, 3},  {3, 0},  {1, 3},
    {-1, 3}, {3, 1},  {-3, 1}, {2, 3},  {-2, 3}, {3, 2},  {-3, 2}, {0, 4},  {4, 0},  {1, 4},  {-1, 4}, {4, 1},  {-4, 1}, {3, 3},  {-3, 3},
    {2, 4},  {-2, 4}, {4, 2},  {-4, 2}, {0, 5},  {3, 4},  {-3, 4}, {4, 3},  {-4, 3}, {5, 0},  {1, 5},  {-1, 5}, {5, 1},  {-5, 1}, {2, 5},
    {-2, 5}, {5, 2},  {-5, 2}, {4, 4},  {-4, 4}, {3, 5},  {-3, 5}, {5, 3},  {-5, 3}, {0, 6},  {6, 0},  {1, 6},  {-1, 6}, {6, 1},  {-6, 1},
    {2, 6},  {-2, 6}, {6, 2},  {-6, 2}, {4, 5},  {-4, 5}, {5, 4},  {-5, 4}, {3, 6},  {-3, 6}, {6, 3},  {-6, 3}, {0, 7},  {7, 0},  {1, 7},
    {-1, 7}, {5, 5},  {-5, 5}, {7, 1},  {-7, 1}, {4, 6},  {-4, 6}, {6, 4},  {-6, 4}, {2, 7},  {-2, 7}, {7, 2},  {-7, 2}, {3, 7},  {-3, 7},
    {7, 3},  {-7, 3}, {5, 6},  {-5, 6}, {6, 5},  {-6, 5}, {8, 0},  {4, 7},  {-4, 7}, {7, 4},  {-7, 4}, {8, 1},  {8, 2},  {6, 6},  {-6, 6},
    {8, 3},  {5, 7},  {-5, 7}, {7, 5},  {-7, 5}, {8, 4},  {6, 7},  {-6, 7}, {7, 6},  {-7, 6}, {8, 5},  {7, 7},  {-7, 7}, {8, 6},  {8, 7}};

void EntropyReader::Init(const EntropyCode& c, BitReader& b, uint32_t dist_mult) {
  code = &c;
  br = &b;
  dist_multiplier = dist_mult;
  num_to_copy = copy_pos = num_decoded = 0;
  if (c.lz77) {
    window.assign(kWindowSize, 0);
    lz_ctx = (uint32_t)c.ctx_map.size() - 1;
  }
  state = c.use_prefix ? (kAnsSignature << 16) : b.Read(32);
}

uint32_t EntropyReader::ReadSymbol(uint32_t hist) {
  if (code->use_prefix) return ReadPrefixSymbol(*br, code->prefix[hist]);
  const uint32_t log_entry = kAnsLogTabSize - code->log_alpha;
  const uint32_t res = state & (kAnsTabSize - 1);
  const uint32_t i = res >> log_entry;
  const uint32_t pos = res & ((1u << log_entry) - 1);
  const AliasEntry& e = code->alias[hist][i];
  const bool greater = pos >= e.cutoff;
  const uint32_t symbol = greater ? e.right_value : i;
  const uint32_t offset = greater ? e.offsets1 + pos : pos;
  const uint32_t freq = greater ? (e.freq0 ^ e.freq1_xor_freq0) : e.freq0;
  state = freq * (state >> kAnsLogTabSize) + offset;
  if (state < (1u << 16)) state = (state << 16) | br->Read(16);
  return symbol;
}

uint32_t EntropyReader::Read(uint32_t ctx) {
  if (code->lz77) {
    if (num_to_copy > 0) {
      uint32_t v = window[(copy_pos++) & kWindowMask];
      num_to_copy--;
      window[(num_decoded++) & kWindowMask] = v;
      return v;
    }
    uint32_t hist = code->ctx_map[ctx];
    uint32_t token = ReadSymbol(hist);
    if (token >= code->lz_min_symbol) {
      num_to_copy = ReadHybrid(code->lz_len_cfg, token - code->lz_min_symbol) + code->lz_min_length;
      uint32_t dh = code->ctx_map[lz_ctx];
      uint32_t dtoken = ReadSymbol(dh);
      uint32_t distance = ReadHybrid(code->cfg[dh], dtoken);
      if (dist_multiplier == 0) {
        distance++;
      } else if (distance < 120) {
        int off = kSpecialDistances[distance][0] + (int)dist_multiplier * kSpecialDistances[distance][1];
        distance = off < 1 ? 1 : off;
      } else {
        distance -= 119;
      }
      distance = std::min(distance, std::min(num_decoded, kWindowSize));
      copy_pos = num_decoded - distance;
      if (distance == 0) {
        // nothing decoded yet: copies zeros
        JXO_CHECK(num_decoded == 0, "lz77 distance");
        uint32_t n = std::min<uint32_t>(num_to_copy, kWindowSize);
        std::fill(window.begin(), window.begin() + n, 0);
      }
      JXO_CHECK(num_to_copy >= code->lz_min_length, "lz77 copy length");
      return Read(ctx);
    }
    uint32_t v = ReadHybrid(code->cfg[hist], token);
    window[(num_decoded++) & kWindowMask] = v;
    return v;
  }
  uint32_t hist = code->ctx_map[ctx];
  uint32_t token = ReadSymbol(hist);
  return ReadHybrid(code->cfg[hist], token);
}

// ================================================================== encoder
static void WriteVarLenUint8(BitWriter& bw, uint32_t n) {
  if (n == 0) { bw.Write(1, 0); return; }
  bw.Write(1, 1);
  uint32_t nb = FloorLog2(n);
  bw.Write(3, nb);
  bw.Write(nb, n - (1u << nb));
}

static void WriteUintConfig(BitWriter& bw, const HybridUintConfig& c, uint32_t log_alpha) {
  bw.Write(CeilLog2(log_alpha + 1), c.split_exponent);
  if (c.split_exponent == log_alpha) return;
  bw.Write(CeilLog2(c.split_exponent + 1), c.msb_in_token);
  bw.Write(CeilLog2(c.split_exponent - c.msb_in_token + 1), c.lsb_in_token);
}

// Normalise `h` (raw counts) to sum kAnsTabSize with every used symbol >= 1.
static std::vector<int32_t> NormalizeCounts(const std::vector<uint32_t>& h) {
  std::vector<int32_t> out(h.size(), 0);
  uint64_t total = 0;
  for (auto c : h) total += c;
  if (total == 0) return out;
  int64_t sum = 0;
  for (size_t i = 0; i < h.size(); i++) {
    if (!h[i]) continue;
    int64_t v = (int64_t)std::llround((double)h[i] * kAnsTabSize / (double)total);
    if (v < 1) v = 1;
    out[i] = (int32_t)v;
    sum += v;
  }
  // fix up the sum by nudging the largest entries
  while (sum != kAnsTabSize) {
    size_t best = 0;
    for (size_t i = 1; i < out.size(); i++)
      if (out[i] > out[best]) best = i;
    int64_t delta = (int64_t)kAnsTabSize - sum;
    if (delta < 0 && out[best] + delta < 1) delta = 1 - out[best];
    JXO_CHECK(delta != 0, "histogram normalisation");
    out[best] += (int32_t)delta;
    sum += delta;
  }
  while (!out.empty() && out.back() == 0) out.pop_back();
  return out;
}

static void WriteHistogram(BitWriter& bw, const std::vector<int32_t>& counts) {
  std::vector<int> syms;
  for (size_t i = 0; i < counts.size(); i++)
    if (counts[i]) syms.push_back((int)i);
  if (syms.size() <= 2) {
    bw.Write(1, 1);
    if (syms.empty()) { bw.Write(1, 0); WriteVarLenUint8(bw, 0); return; }
    bw.Write(1, syms.size() - 1);
    for (int s : syms) WriteVarLenUint8(bw, s);
    if (syms.size() == 2) bw.Write(kAnsLogTabSize, counts[syms[0]]);
    return;
  }
  bw.Write(1, 0);  // not simple
  bw.Write(1, 0);  // not flat
  const int shift = kAnsLogTabSize + 1;  // full precision
  bw.Write(3, 7);                        // unary "111" => log = 3
  bw.Write(3, (shift + 1) - 8);
  int length = (int)counts.size();
  WriteVarLenUint8(bw, length - 3);
  // encode table for log-count symbols
  static uint8_t enc_len[14], enc_bits[14];
  static bool init = false;
  if (!init) {
    for (int s = 0; s < 14; s++)
      for (int idx = 0; idx < 128; idx++)
        if (kLogCountLut[idx][1] == s) {
          enc_len[s] = kLogCountLut[idx][0];
          enc_bits[s] = idx & ((1 << enc_len[s]) - 1);
          break;
        }
    init = true;
  }
  std::vector<int> lc(length);
  int omit_log = -1, omit_pos = -1;
  for (int i = 0; i < length; i++) {
    lc[i] = counts[i] ? FloorLog2(counts[i]) + 1 : 0;
    if (lc[i] > omit_log) { omit_log = lc[i]; omit_pos = i; }
  }
  for (int i = 0; i < length; i++) bw.Write(enc_len[lc[i]], enc_bits[lc[i]]);
  for (int i = 0; i < length; i++) {
    if (i == omit_pos || lc[i] <= 1) continue;
    int bitcount = PopulationCountPrecision(lc[i] - 1, shift);
    JXO_CHECK(bitcount == lc[i] - 1, "precision");
    bw.Write(bitcount, counts[i] - (1 << (lc[i] - 1)));
  }
}

static double HistEntropyBits(const std::vector<uint32_t>& h, uint64_t total) {
  if (!total) return 0;
  double e = 0;
  for (auto c : h)
    if (c) e -= (double)c * std::log2((double)c / (double)total);
  return e;
}

struct RawHist {
  std::vector<uint32_t> c;
  uint64_t total = 0;
  double entropy = 0;
};

static double MergeCost(const RawHist& a, const RawHist& b) {
  // increase in coded size when the two are coded with their common histogram
  uint64_t tot = a.total + b.total;
  if (!a.total || !b.total) return 0;
  double e = 0;
  size_t n = std::max(a.c.size(), b.c.size());
  for (size_t i = 0; i < n; i++) {
    uint64_t v = (i < a.c.size() ? a.c[i] : 0) + (uint64_t)(i < b.c.size() ? b.c[i] : 0);
    if (v) e -= (double)v * std::log2((double)v / (double)tot);
  }
  return e - a.entropy - b.entropy;
}

// ---- test modes: prefix codes and LZ77
static uint32_t g_entropy_test_mode = 0;
void SetEntropyTestMode(uint32_t mode) { g_entropy_test_mode = mode; }
static std::mutex g_dm_mutex;
static std::map<const void*, uint32_t> g_dist_mult;
void SetStreamDistMult(const std::vector<Token>* tokens, uint32_t dist_mult) {
  std::lock_guard<std::mutex> lock(g_dm_mutex);
  if (dist_mult) g_dist_mult[tokens] = dist_mult; else g_dist_mult.erase(tokens);
}
// The registry is keyed by the address of a token vector: entries must not outlive the frame that made them, or a later vector at
// the same address (say, an HF coefficient stream, which has no multiplier) inherits a stale one and the stream no longer decodes.
void ClearStreamDistMults() {
  std::lock_guard<std::mutex> lock(g_dm_mutex);
  g_dist_mult.clear();
}
static uint32_t StreamDistMult(const void* tokens) {
  std::lock_guard<std::mutex> lock(g_dm_mutex);
  auto it = g_dist_mult.find(tokens);
  return it == g_dist_mult.end() ? 0u : it->second;
}

namespace {
const uint32_t kLzMinSymbol = 224, kLzMinLength = 3, kLzMaxLength = 200;
struct Sym { uint32_t ctx, tok, nb, bits; };

// Token stream -> emitted symbols.  With LZ77: a run of values equal to the value before it, or a stretch equal to the stretch
// `dist_mult` symbols back, of at least kLzMinLength symbols becomes (length symbol in the position's context, distance symbol in
// the extra context).  The window holds decoded VALUES, so matches are on Token::value; contexts of copied positions are never coded.
void Symbolize(const std::vector<Token>& t, const HybridUintConfig& cfg, bool lz77, uint32_t dist_ctx, uint32_t dist_mult, std::vector<Sym>& out) {
  out.clear();
  const size_t n = t.size();
  for (size_t i = 0; i < n;) {
    size_t len = 0;
    uint32_t dist_value = 0;
    if (lz77 && i >= 1) {
      size_t l1 = 0;
      while (i + l1 < n && l1 < kLzMaxLength && t[i + l1].value == t[i - 1].value) l1++;
      size_t l2 = 0;
      if (dist_mult > 1 && i >= dist_mult)
        while (i + l2 < n && l2 < kLzMaxLength && t[i + l2].value == t[i + l2 - dist_mult].value) l2++;
      if (l2 > l1 && l2 >= kLzMinLength) { len = l2; dist_value = 0; }                        // special distance 0: (dx 0, dy 1) = dist_mult back
      else if (l1 >= kLzMinLength) { len = l1; dist_value = dist_mult ? 1 : 0; }             // one back: special distance 1 (dx 1, dy 0), or 0 + 1
    }
    Sym s;
    if (len) {
      HybridUintConfig(4, 2, 0).Encode((uint32_t)(len - kLzMinLength), &s.tok, &s.nb, &s.bits);
      s.tok += kLzMinSymbol;
      s.ctx = t[i].ctx;
      out.push_back(s);
      cfg.Encode(dist_value, &s.tok, &s.nb, &s.bits);
      s.ctx = dist_ctx;
      out.push_back(s);
      i += len;
    } else {
      cfg.Encode(t[i].value, &s.tok, &s.nb, &s.bits);
      JXO_CHECK(!lz77 || s.tok < kLzMinSymbol, "literal token collides with the LZ77 symbols");
      s.ctx = t[i].ctx;
      out.push_back(s);
      i++;
    }
  }
}

// Huffman code lengths (<= limit) for the counts; symbols with count 0 get length 0.  At least two symbols must be used.
std::vector<uint8_t> HuffmanLengths(std::vector<uint64_t> counts, int limit) {
  for (;;) {
    struct Node { uint64_t w; int left, right; };
    std::vector<Node> nodes;
    std::vector<int> alive;
    for (size_t i = 0; i < counts.size(); i++) { nodes.push_back({counts[i], -1, -1}); if (counts[i]) alive.push_back((int)i); }
    JXO_CHECK(alive.size() >= 2, "Huffman needs two symbols");
    while (alive.size() > 1) {
      std::sort(alive.begin(), alive.end(), [&](int a, int b) { return nodes[a].w != nodes[b].w ? nodes[a].w > nodes[b].w : a > b; });
      const int a = alive.back(); alive.pop_back();
      const int b = alive.back(); alive.pop_back();
      nodes.push_back({nodes[a].w + nodes[b].w, a, b});
      alive.push_back((int)nodes.size() - 1);
    }
    std::vector<uint8_t> len(counts.size(), 0);
    int maxlen = 0;
    std::vector<std::pair<int, int>> stack = {{alive[0], 0}};
    while (!stack.empty()) {
      auto [nd, d] = stack.back();
      stack.pop_back();
      if (nodes[nd].left < 0) { len[nd] = (uint8_t)d; maxlen = std::max(maxlen, d); }
      else { stack.push_back({nodes[nd].left, d + 1}); stack.push_back({nodes[nd].right, d + 1}); }
    }
    if (maxlen <= limit) return len;
    for (auto& c : counts) if (c) c = std::max<uint64_t>(1, c >> 1);   // flatten and retry
  }
}
// canonical codes (MSB first) from lengths
std::vector<uint16_t> CanonicalCodes(const std::vector<uint8_t>& len) {
  std::vector<uint16_t> code(len.size(), 0);
  uint32_t next = 0;
  for (int l = 1; l <= 15; l++) {
    for (size_t i = 0; i < len.size(); i++) if (len[i] == l) code[i] = (uint16_t)next++;
    next <<= 1;
  }
  return code;
}
void WriteMsbFirst(BitWriter& bw, uint32_t code, int len) { for (int b = len - 1; b >= 0; b--) bw.Write(1, (code >> b) & 1); }

void WriteVarLenUint16(BitWriter& bw, uint32_t n) {
  if (n == 0) { bw.Write(1, 0); return; }
  bw.Write(1, 1);
  const uint32_t nb = FloorLog2(n);
  bw.Write(4, nb);
  bw.Write(nb, n - (1u << nb));
}

// One prefix code in the complex form of RFC 7932 section 3.5 (no run-length symbols), or the one-symbol simple form.
void WritePrefixCodeHeader(BitWriter& bw, const std::vector<uint8_t>& len, uint32_t alphabet_size) {
  if (alphabet_size == 1) return;   // nothing is coded for a one-symbol alphabet
  int used = 0, last = 0;
  for (size_t i = 0; i < len.size(); i++) if (len[i]) { used++; last = (int)i; }
  if (used == 1) {   // simple form, NSYM = 1
    bw.Write(2, 1);
    bw.Write(2, 0);
    int nbits = 0;
    for (uint32_t c = alphabet_size - 1; c; c >>= 1) nbits++;
    bw.Write(nbits, last);
    return;
  }
  // code-length code over the length values 0..15 that occur
  std::vector<uint64_t> lc(18, 0);
  for (int i = 0; i <= last; i++) lc[len[i]]++;
  int lused = 0, lonly = 0;
  for (int v = 0; v < 18; v++) if (lc[v]) { lused++; lonly = v; }
  std::vector<uint8_t> cl(18, 0);
  if (lused == 1) cl[lonly] = 1;   // (cannot happen with >= 2 used symbols of a complete code and zeros in between, but be safe)
  else cl = HuffmanLengths(lc, 5);
  static const uint8_t kOrder[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
  static const uint8_t kVlcBits[6] = {0x0, 0x7, 0x3, 0x2, 0x1, 0xF}, kVlcLen[6] = {2, 4, 3, 2, 2, 4};   // value -> LSB-first code
  bw.Write(2, 0);   // HSKIP = 0
  int space = 32, ncodes = 0;
  for (int i = 0; i < 18 && space > 0; i++) {
    const int v = cl[kOrder[i]];
    bw.Write(kVlcLen[v], kVlcBits[v]);
    if (v) { space -= 32 >> v; ncodes++; }
  }
  JXO_CHECK(space == 0 || ncodes == 1, "code-length code is not complete");
  const std::vector<uint16_t> clcode = CanonicalCodes(cl);
  for (int i = 0; i <= last; i++) WriteMsbFirst(bw, clcode[len[i]], ncodes == 1 ? 0 : cl[len[i]]);
}
}  // namespace

void BuildAndWriteCode(const std::vector<const std::vector<Token>*>& token_sets, size_t num_contexts,
                       const EncOptions& opt, BitWriter& bw, EncCode& out) {
  out = EncCode();
  EntropyCode& code = out.code;
  const bool use_prefix = opt.use_prefix || (opt.top_level && (g_entropy_test_mode & 1));
  const bool lz77 = opt.lz77 || (opt.top_level && (g_entropy_test_mode & 2));
  out.num_contexts = num_contexts;
  const size_t data_contexts = num_contexts;
  if (lz77) num_contexts++;   // the distance context
  // 1. per-context raw histograms of the emitted symbols
  std::vector<RawHist> hist(num_contexts);
  uint32_t max_token = 0;
  auto count = [&](uint32_t ctx, uint32_t tok) {
    RawHist& h = hist[ctx];
    if (h.c.size() <= tok) h.c.resize(tok + 1, 0);
    h.c[tok]++;
    h.total++;
    max_token = std::max(max_token, tok);
  };
  std::vector<Sym> syms;
  for (auto* ts : token_sets) {
    for (const Token& t : *ts) JXO_CHECK(t.ctx < data_contexts, "token context out of range");
    Symbolize(*ts, opt.cfg, lz77, (uint32_t)data_contexts, StreamDistMult(ts), syms);
    for (const Sym& sy : syms) count(sy.ctx, sy.tok);
  }
  if (lz77) { count((uint32_t)data_contexts, 0); count((uint32_t)data_contexts, 1); }
  JXO_CHECK(use_prefix || max_token < 256, "token alphabet exceeds ANS limit");
  for (auto& h : hist) h.entropy = HistEntropyBits(h.c, h.total);
  // 2. clustering
  code.ctx_map.assign(num_contexts, 0);
  std::vector<RawHist> clusters;
  std::vector<size_t> used;
  for (size_t i = 0; i < num_contexts; i++)
    if (hist[i].total) used.push_back(i);
  size_t maxc = opt.force_single_cluster ? 1 : (size_t)std::max(1, std::min(opt.max_clusters, 255));
  if (used.size() <= maxc) {
    for (size_t k = 0; k < used.size(); k++) {
      code.ctx_map[used[k]] = (uint8_t)k;
      clusters.push_back(hist[used[k]]);
    }
    if (clusters.empty()) clusters.push_back(RawHist());
  } else {
    // farthest-point seeding in merge-cost distance, then nearest-seed assignment
    std::vector<size_t> seeds;
    size_t first = used[0];
    for (size_t i : used)
      if (hist[i].total > hist[first].total) first = i;
    seeds.push_back(first);
    std::vector<double> dmin(num_contexts, 1e300);
    while (seeds.size() < maxc) {
      size_t s = seeds.back();
      size_t far = used[0];
      double fard = -1;
      for (size_t i : used) {
        double d = MergeCost(hist[i], hist[s]);
        dmin[i] = std::min(dmin[i], d);
        if (dmin[i] > fard) { fard = dmin[i]; far = i; }
      }
      if (fard <= 0) break;
      seeds.push_back(far);
    }
    for (size_t s : seeds) clusters.push_back(hist[s]);
    std::vector<RawHist> sums(seeds.size());
    for (size_t i : used) {
      size_t best = 0;
      double bd = 1e300;
      for (size_t k = 0; k < seeds.size(); k++) {
        double d = (i == seeds[k]) ? -1 : MergeCost(hist[i], clusters[k]);
        if (d < bd) { bd = d; best = k; }
      }
      code.ctx_map[i] = (uint8_t)best;
      RawHist& s = sums[best];
      if (s.c.size() < hist[i].c.size()) s.c.resize(hist[i].c.size(), 0);
      for (size_t j = 0; j < hist[i].c.size(); j++) s.c[j] += hist[i].c[j];
      s.total += hist[i].total;
    }
    clusters = sums;
  }
  code.num_hist = (uint32_t)clusters.size();
  code.lz77 = lz77;
  code.lz_min_symbol = kLzMinSymbol; code.lz_min_length = kLzMinLength; code.lz_len_cfg = HybridUintConfig(4, 2, 0);
  code.use_prefix = use_prefix;
  code.log_alpha = use_prefix ? 15 : std::max(5, CeilLog2(max_token + 1));
  JXO_CHECK(use_prefix || code.log_alpha <= 8, "log_alpha");
  code.cfg.assign(code.num_hist, opt.cfg);
  JXO_CHECK(opt.cfg.split_exponent <= code.log_alpha, "uint config vs alphabet");
  // 3. header
  bw.Write(1, lz77 ? 1 : 0);
  if (lz77) {
    bw.U32(Val(224), Val(512), Val(4096), BitsOff(15, 8), kLzMinSymbol);
    bw.U32(Val(3), Val(4), BitsOff(2, 5), BitsOff(8, 9), kLzMinLength);
    WriteUintConfig(bw, code.lz_len_cfg, 8);
  }
  if (num_contexts > 1) {
    if (code.num_hist == 1) {
      bw.Write(1, 1);
      bw.Write(2, 0);
    } else if (code.num_hist <= 8 && num_contexts <= 64) {
      int bits = CeilLog2(code.num_hist);
      bw.Write(1, 1);
      bw.Write(2, bits);
      for (auto m : code.ctx_map) bw.Write(bits, m);
    } else {
      bw.Write(1, 0);  // not simple
      bw.Write(1, 0);  // no MTF
      std::vector<Token> mt;
      for (auto m : code.ctx_map) mt.emplace_back(0, m);
      EncOptions mo;
      mo.cfg = HybridUintConfig(4, 2, 0);
      mo.top_level = false;
      EncCode mc;
      std::vector<const std::vector<Token>*> sets = {&mt};
      BuildAndWriteCode(sets, 1, mo, bw, mc);
      WriteTokens(mt, mc, bw);
    }
  }
  bw.Write(1, use_prefix ? 1 : 0);
  if (!use_prefix) bw.Write(2, code.log_alpha - 5);
  for (auto& c : code.cfg) WriteUintConfig(bw, c, code.log_alpha);
  if (use_prefix) {
    // alphabet sizes, then the codes
    std::vector<uint32_t> asz(code.num_hist);
    for (uint32_t k = 0; k < code.num_hist; k++) {
      asz[k] = std::max<uint32_t>(1, (uint32_t)clusters[k].c.size());
      while (asz[k] > 1 && clusters[k].c[asz[k] - 1] == 0) asz[k]--;
      WriteVarLenUint16(bw, asz[k] - 1);
    }
    code.prefix.resize(code.num_hist);
    out.pcode.resize(code.num_hist);
    for (uint32_t k = 0; k < code.num_hist; k++) {
      std::vector<uint64_t> cnt(asz[k], 0);
      int used = 0;
      for (uint32_t i = 0; i < asz[k] && i < clusters[k].c.size(); i++) { cnt[i] = clusters[k].c[i]; used += cnt[i] != 0; }
      std::vector<uint8_t> len(asz[k], 0);
      if (used >= 2) len = HuffmanLengths(cnt, 15);
      else for (uint32_t i = 0; i < asz[k]; i++) if (cnt[i]) len[i] = 1;   // a single symbol: zero bits, its "length" only marks it
      WritePrefixCodeHeader(bw, len, asz[k]);
      if (used < 2) std::fill(len.begin(), len.end(), 0);
      code.prefix[k].lengths = len;
      out.pcode[k] = CanonicalCodes(len);
    }
    return;
  }
  code.counts.resize(code.num_hist);
  code.alias.resize(code.num_hist);
  out.reverse_map.resize(code.num_hist);
  out.sym_start.resize(code.num_hist);
  for (uint32_t k = 0; k < code.num_hist; k++) {
    std::vector<int32_t> counts = NormalizeCounts(clusters[k].c);
    WriteHistogram(bw, counts);
    if (counts.empty()) counts.assign(1, kAnsTabSize);  // what the decoder reconstructs
    code.counts[k] = counts;
    InitAliasTable(counts, code.log_alpha, code.alias[k]);
    // reverse map: (symbol, offset) -> 12-bit slot
    auto& ss = out.sym_start[k];
    ss.assign(counts.size() + 1, 0);
    for (size_t s = 0; s < counts.size(); s++) ss[s + 1] = ss[s] + counts[s];
    auto& rm = out.reverse_map[k];
    rm.assign(kAnsTabSize, 0);
    const uint32_t log_entry = kAnsLogTabSize - code.log_alpha;
    for (uint32_t res = 0; res < kAnsTabSize; res++) {
      uint32_t i = res >> log_entry, pos = res & ((1u << log_entry) - 1);
      const AliasEntry& e = code.alias[k][i];
      bool greater = pos >= e.cutoff;
      uint32_t symbol = greater ? e.right_value : i;
      uint32_t offset = greater ? e.offsets1 + pos : pos;
      rm[ss[symbol] + offset] = (uint16_t)res;
    }
  }
}

void WriteTokens(const std::vector<Token>& tokens, const EncCode& ec, BitWriter& bw) {
  const EntropyCode& code = ec.code;
  std::vector<Sym> syms;
  Symbolize(tokens, code.cfg[0], code.lz77, (uint32_t)ec.num_contexts, StreamDistMult(&tokens), syms);
  const size_t n = syms.size();
  if (code.use_prefix) {
    for (const Sym& sy : syms) {
      const uint32_t h = code.ctx_map[sy.ctx];
      const std::vector<uint8_t>& len = code.prefix[h].lengths;
      JXO_CHECK(sy.tok < len.size(), "symbol outside the prefix alphabet");
      WriteMsbFirst(bw, ec.pcode[h][sy.tok], len[sy.tok]);
      bw.Write(sy.nb, sy.bits);
    }
    return;
  }
  std::vector<uint16_t> flush_bits(n);
  std::vector<uint8_t> flushed(n, 0);
  uint32_t state = kAnsSignature << 16;
  for (size_t r = n; r-- > 0;) {
    const Sym& t = syms[r];
    uint32_t h = code.ctx_map[t.ctx];
    const uint32_t tok = t.tok;
    JXO_CHECK(tok < code.counts[h].size() && code.counts[h][tok] > 0, "token not in histogram");
    uint32_t freq = code.counts[h][tok];
    if ((state >> (32 - kAnsLogTabSize)) >= freq) {
      flush_bits[r] = (uint16_t)(state & 0xffff);
      flushed[r] = 1;
      state >>= 16;
    }
    state = ((state / freq) << kAnsLogTabSize) + ec.reverse_map[h][ec.sym_start[h][tok] + state % freq];
  }
  bw.Write(32, state);
  for (size_t i = 0; i < n; i++) {
    if (flushed[i]) bw.Write(16, flush_bits[i]);
    bw.Write(syms[i].nb, syms[i].bits);
  }
}

}  // namespace jxo
