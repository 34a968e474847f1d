// ORACLE — test infrastructure only (see jxo_common.h header).
// ICC profile <-> the "predicted" byte stream a JPEG XL codestream carries (18181-1, ICC annex; reached by the reference through
// JxlEncoderSetICCProfile, Encoder/JxlEncoder.cpp:258-268, and JxlDecoderGetColorAsICCProfile, Decoder/JxlDecoder.cpp:596-686).
// Parity unpinned: restated from the published format; the reference holds no vector for it.
#pragma once
#include "jxo_common.h"

namespace jxo {

constexpr size_t kNumIccContexts = 41;
uint32_t IccByteContext(size_t index, uint32_t b1, uint32_t b2);
// Encoder form of the oracle: tag-table commands (known tag names, implied offsets / sizes), XYZ and type-start commands for the tag
// data it recognises, plain inserts for the rest.
std::vector<uint8_t> IccToStream(const std::vector<uint8_t>& icc);
std::vector<uint8_t> IccFromStream(const std::vector<uint8_t>& enc);   // throws Error

}  // namespace jxo
