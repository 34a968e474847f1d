// ICC profiles inside a JPEG XL codestream (ISO/IEC 18181-1 annex on ICC coding; what the reference reaches through
// JxlEncoderSetICCProfile, Encoder/JxlEncoder.cpp:258-268, and JxlDecoderGetColorAsICCProfile, Decoder/JxlDecoder.cpp:596-686).
// The profile travels as a byte stream in which the 128-byte header, the tag table and the tag data are replaced by
// predictions + residuals under a small command language; that byte stream is then entropy coded with 41 contexts.
// [spec: restated from the published format, no fixture in the reference pins it - see DESIGN.md "parity unpinned"]
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace jxlhip {

constexpr size_t kIccContexts = 41;
constexpr size_t kIccMaxEncodedSize = 1u << 28;

// Context of byte i of the predicted stream given the two bytes before it.
uint32_t IccContext(size_t i, uint8_t prev1, uint8_t prev2);

// Profile -> predicted stream.  The form written here is the simplest valid one: header residuals against the standard header
// prediction, no tag-table commands, one "insert" command for everything after the header.
void IccPredict(const uint8_t* icc, size_t size, std::vector<uint8_t>* enc);

// Predicted stream -> profile, full command set (tag table commands, insert, shuffle, n-th order predictors, type starts).
bool IccUnpredict(const std::vector<uint8_t>& enc, std::vector<uint8_t>* icc, std::string* why);

// Minimal colour management: "matrix / TRC" RGB profiles (three colorants + three tone curves: display, working-space and camera-RGB
// profiles - sRGB, Display P3, Adobe RGB, ProPhoto, monitor profiles) and gray TRC profiles are evaluated here, as tables.  Anything
// else (LUT-based A2B profiles, CMYK, Lab, ...) is not: the callers then take the reference's own fallback (Decoder/JxlDecoder.cpp:
// 586-601) or refuse loudly.
struct IccModel {
  bool gray = false;
  double rgb_to_xyz_d50[9];          // linear RGB of the profile -> PCS XYZ (D50), row-major (gray: unused)
  std::vector<float> to_linear[3];   // 256 entries: encoded sample i / 255 -> linear
  std::vector<float> from_linear[3]; // kIccInvLut entries: linear (i / (n - 1))^2 -> encoded
  double from_linear_srgb[9];        // linear sRGB (D65) -> linear RGB of the profile (chromatic adaptation D65 -> D50 by Bradford)
  double to_linear_srgb[9];          // the inverse
};
constexpr int kIccInvLut = 4096;
bool IccBuildModel(const uint8_t* icc, size_t size, IccModel* model);

// Colour spaces given by chromaticities and a simple tone curve (the enumerated encodings of a codestream that are not one of the
// host's eight named profiles): conversion matrices through D50 (Bradford), and a matrix / TRC v4 profile describing the space - the
// counterpart of the profile the reference's decoder library synthesises for JXL_COLOR_PROFILE_TARGET_DATA (Decoder/JxlDecoder.cpp:652-682).
struct IccCurveSpec {
  int kind = 0;        // 0 linear, 1 sRGB, 2 BT.709, 4 pure gamma: encoded = linear ^ gamma
  double gamma = 1.0;
};
// RGB with these primaries and white point -> XYZ adapted to D50 (row-major)
void PrimariesToXyzD50(const double prim_xy[3][2], const double white_xy[2], double out[9]);
// linear sRGB (D65) -> linear RGB of the space
bool MatrixFromLinearSrgb(const double prim_xy[3][2], const double white_xy[2], double out[9]);
std::vector<uint8_t> IccSynthesize(bool gray, const double prim_xy[3][2], const double white_xy[2], const IccCurveSpec& curve, uint32_t rendering_intent);

// What the decode path needs to know about a profile without a colour management system:
//   colour space of the data (header bytes 16..19): 'RGB ', 'GRAY', 'CMYK', ...
uint32_t IccDataColorSpace(const uint8_t* icc, size_t size);

}  // namespace jxlhip
