// ORACLE — test infrastructure only (see jxo_common.h header).
#include "jxo_headers.h"
#include "jxo_entropy.h"

namespace jxo {

// ------------------------------------------------------------------ size header
static const uint32_t kRatioNum[8] = {0, 1, 12, 4, 3, 16, 5, 2};
static const uint32_t kRatioDen[8] = {0, 1, 10, 3, 2, 9, 4, 1};

void ReadSizeHeader(BitReader& br, uint32_t* xsize, uint32_t* ysize) {
  bool small = br.Bool();
  uint32_t y = small ? (br.Read(5) + 1) * 8 : br.U32(BitsOff(9, 1), BitsOff(13, 1), BitsOff(18, 1), BitsOff(30, 1));
  uint32_t ratio = br.Read(3);
  uint32_t x;
  if (ratio == 0) x = small ? (br.Read(5) + 1) * 8 : br.U32(BitsOff(9, 1), BitsOff(13, 1), BitsOff(18, 1), BitsOff(30, 1));
  else x = (uint32_t)((uint64_t)y * kRatioNum[ratio] / kRatioDen[ratio]);
  *xsize = x;
  *ysize = y;
}

void WriteSizeHeader(BitWriter& bw, uint32_t xsize, uint32_t ysize) {
  JXO_CHECK(xsize > 0 && ysize > 0, "empty image");
  uint32_t ratio = 0;
  for (uint32_t r = 1; r < 8; r++)
    if ((uint32_t)((uint64_t)ysize * kRatioNum[r] / kRatioDen[r]) == xsize) { ratio = r; break; }
  bool small = ysize <= 256 && ysize % 8 == 0 && (ratio != 0 || (xsize <= 256 && xsize % 8 == 0));
  bw.Bool(small);
  if (small) bw.Write(5, ysize / 8 - 1);
  else bw.U32(BitsOff(9, 1), BitsOff(13, 1), BitsOff(18, 1), BitsOff(30, 1), ysize);
  bw.Write(3, ratio);
  if (ratio == 0) {
    if (small) bw.Write(5, xsize / 8 - 1);
    else bw.U32(BitsOff(9, 1), BitsOff(13, 1), BitsOff(18, 1), BitsOff(30, 1), xsize);
  }
}

// ------------------------------------------------------------------ image metadata
static void ReadBitDepth(BitReader& br, uint32_t* bits, uint32_t* exp_bits) {
  bool fl = br.Bool();
  if (!fl) {
    *bits = br.U32(Val(8), Val(10), Val(12), BitsOff(6, 1));
    *exp_bits = 0;
  } else {
    *bits = br.U32(Val(32), Val(16), Val(24), BitsOff(6, 1));
    *exp_bits = br.Read(4) + 1;
  }
}
static void WriteBitDepth(BitWriter& bw, uint32_t bits, uint32_t exp_bits) {
  bw.Bool(exp_bits != 0);
  if (!exp_bits) bw.U32(Val(8), Val(10), Val(12), BitsOff(6, 1), bits);
  else { bw.U32(Val(32), Val(16), Val(24), BitsOff(6, 1), bits); bw.Write(4, exp_bits - 1); }
}

static uint32_t ReadNameLen(BitReader& br) { return br.U32(Val(0), Bits(4), BitsOff(5, 16), BitsOff(10, 48)); }

static int32_t ReadCustomXY(BitReader& br) {
  return (int32_t)UnpackSigned(br.U32(Bits(19), BitsOff(19, 524288), BitsOff(20, 1048576), BitsOff(21, 2097152)));
}

static void ReadColorEncoding(BitReader& br, ColorEncoding& c) {
  c = ColorEncoding();
  c.all_default = br.Bool();
  if (c.all_default) return;
  c.want_icc = br.Bool();
  c.color_space = br.Enum();
  if (c.want_icc) return;
  if (c.color_space != 2) {
    c.white_point = br.Enum();
    if (c.white_point == 2) { c.custom_xy[0][0] = ReadCustomXY(br); c.custom_xy[0][1] = ReadCustomXY(br); }
  }
  if (c.color_space != 2 && c.color_space != 1) {
    c.primaries = br.Enum();
    if (c.primaries == 2)
      for (int i = 1; i < 4; i++) { c.custom_xy[i][0] = ReadCustomXY(br); c.custom_xy[i][1] = ReadCustomXY(br); }
  }
  if (c.color_space != 2) {
    c.have_gamma = br.Bool();
    if (c.have_gamma) c.gamma = br.Read(24);
    else c.tf = br.Enum();
  }
  c.rendering_intent = br.Enum();
}

static void WriteColorEncoding(BitWriter& bw, const ColorEncoding& c) {
  bw.Bool(c.all_default);
  if (c.all_default) return;
  bw.Bool(c.want_icc);
  bw.Enum(c.color_space);
  if (c.want_icc) return;
  auto write_xy = [&](int32_t v) {
    const uint32_t u = v >= 0 ? (uint32_t)v << 1 : (((uint32_t)(-(int64_t)v)) << 1) - 1;
    bw.U32(Bits(19), BitsOff(19, 524288), BitsOff(20, 1048576), BitsOff(21, 2097152), u);
  };
  if (c.color_space != 2) {
    bw.Enum(c.white_point);
    if (c.white_point == 2) { write_xy(c.custom_xy[0][0]); write_xy(c.custom_xy[0][1]); }
  }
  if (c.color_space != 2 && c.color_space != 1) {
    bw.Enum(c.primaries);
    if (c.primaries == 2) for (int i = 1; i < 4; i++) { write_xy(c.custom_xy[i][0]); write_xy(c.custom_xy[i][1]); }
  }
  if (c.color_space != 2) {
    bw.Bool(c.have_gamma);
    if (c.have_gamma) bw.Write(24, c.gamma);
    else bw.Enum(c.tf);
  }
  bw.Enum(c.rendering_intent);
}

static const float kDefaultOpsinInverse[9] = {11.031566901960783f, -9.866943921568629f, -0.16462299647058826f,
                                              -3.254147380392157f, 4.418770392156863f,  -0.16462299647058826f,
                                              -3.6588512862745097f, 2.7129230470588235f, 1.9459282392156863f};

void ImageMetadata::SetDefaultTransformData() {
  memcpy(opsin_inverse, kDefaultOpsinInverse, sizeof(kDefaultOpsinInverse));
  for (int i = 0; i < 3; i++) opsin_bias[i] = -0.0037930732552754493f;
  quant_bias[0] = 1.0f - 0.05465007330715401f;
  quant_bias[1] = 1.0f - 0.07005449891748593f;
  quant_bias[2] = 1.0f - 0.049935103337343655f;
  quant_bias[3] = 0.145f;
}

void ReadImageMetadata(BitReader& br, ImageMetadata& m) {
  bool all_default = br.Bool();
  bool extra_fields = false;
  if (!all_default) {
    extra_fields = br.Bool();
    if (extra_fields) {
      m.orientation = br.Read(3) + 1;
      if (br.Bool()) { uint32_t ix, iy; ReadSizeHeader(br, &ix, &iy); }  // intrinsic size
      m.have_preview = br.Bool();
      if (m.have_preview) throw Error("preview frames are not supported");
      m.have_animation = br.Bool();
      if (m.have_animation) {
        br.U32(Val(100), Val(1000), BitsOff(10, 1), BitsOff(30, 1));
        br.U32(Val(1), Val(1001), BitsOff(8, 1), BitsOff(10, 1));
        br.U32(Val(0), Bits(3), Bits(16), Bits(32));
        m.have_timecodes = br.Bool();
      }
    }
    ReadBitDepth(br, &m.bits, &m.exp_bits);
    m.modular_16bit = br.Bool();
    uint32_t num_ec = br.U32(Val(0), Val(1), BitsOff(4, 2), BitsOff(12, 1));
    m.ec.resize(num_ec);
    for (auto& e : m.ec) {
      e = ExtraChannelInfo();
      bool d_alpha = br.Bool();
      if (d_alpha) continue;
      e.type = br.Enum();
      ReadBitDepth(br, &e.bits, &e.exp_bits);
      e.dim_shift = br.U32(Val(0), Val(3), Val(4), BitsOff(3, 1));
      uint32_t nl = ReadNameLen(br);
      e.name.resize(nl);
      for (auto& ch : e.name) ch = (char)br.Read(8);
      if (e.type == 0) e.alpha_associated = br.Bool();
      if (e.type == 2) for (int i = 0; i < 4; i++) br.F16();
      if (e.type == 5) br.U32(Val(1), Bits(2), BitsOff(4, 3), BitsOff(8, 19));
    }
    m.xyb_encoded = br.Bool();
    ReadColorEncoding(br, m.color);
    if (extra_fields) {
      bool tm_default = br.Bool();
      if (!tm_default) {
        m.intensity_target = br.F16();
        m.min_nits = br.F16();
        m.relative_to_max_display = br.Bool();
        m.linear_below = br.F16();
      }
    }
    uint64_t ext = br.U64();
    if (ext) {
      uint64_t total = 0;
      for (int i = 0; i < 64; i++) if (ext >> i & 1) total += br.U64();
      br.Skip(total);
    }
  }
  // custom transform data
  m.SetDefaultTransformData();
  m.default_transform = br.Bool();
  if (!m.default_transform) {
    if (m.xyb_encoded) {
      bool opsin_default = br.Bool();
      if (!opsin_default) {
        for (int i = 0; i < 9; i++) m.opsin_inverse[i] = br.F16();
        for (int i = 0; i < 3; i++) m.opsin_bias[i] = br.F16();
        for (int i = 0; i < 4; i++) m.quant_bias[i] = br.F16();
      }
    }
    uint32_t cw_mask = br.Read(3);
    if (cw_mask & 1) for (int i = 0; i < 15; i++) br.F16();
    if (cw_mask & 2) for (int i = 0; i < 55; i++) br.F16();
    if (cw_mask & 4) for (int i = 0; i < 210; i++) br.F16();
  }
  JXO_CHECK(!br.overrun, "truncated image metadata");
}

void WriteImageMetadata(BitWriter& bw, const ImageMetadata& m) {
  bool ec_default = true;
  for (auto& e : m.ec) if (e.type != 0 || e.bits != 8 || e.exp_bits || e.dim_shift || !e.name.empty() || e.alpha_associated) ec_default = false;
  bool all_default = m.orientation == 1 && m.bits == 8 && m.exp_bits == 0 && m.modular_16bit && m.ec.empty() && m.xyb_encoded &&
                     m.color.all_default && m.intensity_target == 255.f && !m.have_animation;
  bw.Bool(all_default);
  if (!all_default) {
    bool extra = m.orientation != 1 || m.intensity_target != 255.f || m.have_animation;
    bw.Bool(extra);
    if (extra) {
      bw.Write(3, m.orientation - 1);
      bw.Bool(false);  // intrinsic size
      bw.Bool(false);  // preview
      bw.Bool(m.have_animation);
      if (m.have_animation) {
        bw.U32(Val(100), Val(1000), BitsOff(10, 1), BitsOff(30, 1), 100);   // ticks per second: 100 / 1
        bw.U32(Val(1), Val(1001), BitsOff(8, 1), BitsOff(10, 1), 1);
        bw.U32(Val(0), Bits(3), Bits(16), Bits(32), 0);                      // loops forever
        bw.Bool(false);                                                       // no timecodes
      }
    }
    WriteBitDepth(bw, m.bits, m.exp_bits);
    bw.Bool(m.modular_16bit);
    bw.U32(Val(0), Val(1), BitsOff(4, 2), BitsOff(12, 1), (uint32_t)m.ec.size());
    for (auto& e : m.ec) {
      bool d_alpha = e.type == 0 && e.bits == 8 && !e.exp_bits && !e.dim_shift && e.name.empty() && !e.alpha_associated;
      bw.Bool(d_alpha);
      if (d_alpha) continue;
      bw.Enum(e.type);
      WriteBitDepth(bw, e.bits, e.exp_bits);
      bw.U32(Val(0), Val(3), Val(4), BitsOff(3, 1), e.dim_shift);
      bw.U32(Val(0), Bits(4), BitsOff(5, 16), BitsOff(10, 48), (uint32_t)e.name.size());
      for (char ch : e.name) bw.Write(8, (uint8_t)ch);
      if (e.type == 0) bw.Bool(e.alpha_associated);
      JXO_CHECK(e.type != 2 && e.type != 5, "spot/CFA channels not written");
    }
    (void)ec_default;
    bw.Bool(m.xyb_encoded);
    WriteColorEncoding(bw, m.color);
    if (extra) {
      bool tm_default = m.intensity_target == 255.f && m.min_nits == 0.f && !m.relative_to_max_display && m.linear_below == 0.f;
      bw.Bool(tm_default);
      if (!tm_default) {
        bw.F16(m.intensity_target); bw.F16(m.min_nits); bw.Bool(m.relative_to_max_display); bw.F16(m.linear_below);
      }
    }
    bw.U64(0);  // extensions
  }
  bw.Bool(true);  // default transform data
}

// ------------------------------------------------------------------ frame header
void FrameHeader::Derive(const ImageMetadata& m) {
  xsize = have_crop ? width : m.xsize;
  ysize = have_crop ? height : m.ysize;
  if (frame_type == kLF) {
    uint32_t d = 1u << (3 * lf_level);
    xsize = (uint32_t)DivCeil(xsize, d);
    ysize = (uint32_t)DivCeil(ysize, d);
  }
  if (upsampling > 1) {
    xsize = (uint32_t)DivCeil(xsize, upsampling);
    ysize = (uint32_t)DivCeil(ysize, upsampling);
  }
  group_dim = 128u << group_size_shift;
  xsize_blocks = (uint32_t)DivCeil(xsize, 8);
  ysize_blocks = (uint32_t)DivCeil(ysize, 8);
  xsize_groups = (uint32_t)DivCeil(xsize, group_dim);
  ysize_groups = (uint32_t)DivCeil(ysize, group_dim);
  num_groups = xsize_groups * ysize_groups;
  xsize_lf_groups = (uint32_t)DivCeil(xsize, group_dim * 8);
  ysize_lf_groups = (uint32_t)DivCeil(ysize, group_dim * 8);
  num_lf_groups = xsize_lf_groups * ysize_lf_groups;
}

static void ReadBlendingInfo(BitReader& br, size_t num_ec, bool partial, uint32_t* mode) {
  *mode = br.U32(Val(0), Val(1), Val(2), BitsOff(2, 3));
  if (num_ec > 0 && (*mode == 2 || *mode == 3)) br.U32(Val(0), Val(1), Val(2), BitsOff(3, 3));
  if (num_ec > 0 && (*mode == 2 || *mode == 3 || *mode == 4)) br.Bool();
  if (*mode != 0 || partial) br.Read(2);
}

void ReadFrameHeader(BitReader& br, const ImageMetadata& m, FrameHeader& f) {
  f = FrameHeader();
  f.ec_upsampling.assign(m.ec.size(), 1);
  bool all_default = br.Bool();
  if (!all_default) {
    f.frame_type = br.Read(2);
    f.encoding = br.Read(1);
    f.flags = br.U64();
    if (!m.xyb_encoded) f.do_ycbcr = br.Bool();
    if (f.do_ycbcr && !(f.flags & FrameHeader::kUseLfFrame)) br.Read(6);
    if (!(f.flags & FrameHeader::kUseLfFrame)) {
      f.upsampling = br.U32(Val(1), Val(2), Val(4), Val(8));
      for (auto& u : f.ec_upsampling) u = br.U32(Val(1), Val(2), Val(4), Val(8));
    }
    if (f.encoding == 1) f.group_size_shift = br.Read(2);
    if (f.encoding == 0 && m.xyb_encoded) { f.x_qm_scale = br.Read(3); f.b_qm_scale = br.Read(3); }
    if (f.frame_type != FrameHeader::kReferenceOnly) {
      f.num_passes = br.U32(Val(1), Val(2), Val(3), BitsOff(3, 4));
      if (f.num_passes != 1) {
        uint32_t num_ds = br.U32(Val(0), Val(1), Val(2), BitsOff(1, 3));
        for (uint32_t i = 0; i + 1 < f.num_passes; i++) f.pass_shift[i] = br.Read(2);
        for (uint32_t i = 0; i < num_ds; i++) br.U32(Val(1), Val(2), Val(4), Val(8));
        for (uint32_t i = 0; i < num_ds; i++) br.U32(Val(0), Val(1), Val(2), Bits(3));
      }
    }
    if (f.frame_type == FrameHeader::kLF) {
      f.lf_level = br.U32(Val(1), Val(2), Val(3), Val(4));
    } else {
      f.have_crop = br.Bool();
      if (f.have_crop) {
        if (f.frame_type != FrameHeader::kReferenceOnly) {
          f.x0 = (int32_t)UnpackSigned(br.U32(Bits(8), BitsOff(11, 256), BitsOff(14, 2304), BitsOff(30, 18688)));
          f.y0 = (int32_t)UnpackSigned(br.U32(Bits(8), BitsOff(11, 256), BitsOff(14, 2304), BitsOff(30, 18688)));
        }
        f.width = br.U32(Bits(8), BitsOff(11, 256), BitsOff(14, 2304), BitsOff(30, 18688));
        f.height = br.U32(Bits(8), BitsOff(11, 256), BitsOff(14, 2304), BitsOff(30, 18688));
      }
    }
    bool normal = f.frame_type == FrameHeader::kRegular || f.frame_type == FrameHeader::kSkipProgressive;
    bool full_frame = !f.have_crop || (f.x0 <= 0 && f.y0 <= 0 && f.x0 + (int64_t)f.width >= m.xsize && f.y0 + (int64_t)f.height >= m.ysize);
    if (normal) {
      ReadBlendingInfo(br, m.ec.size(), !full_frame, &f.blend_mode);
      for (size_t i = 0; i < m.ec.size(); i++) { uint32_t mm; ReadBlendingInfo(br, m.ec.size(), !full_frame, &mm); }
      if (m.have_animation) {
        f.duration = br.U32(Val(0), Val(1), Bits(8), Bits(32));
        if (m.have_timecodes) br.Read(32);
      }
      f.is_last = br.Bool();
    } else {
      f.is_last = false;
    }
    if (f.frame_type != FrameHeader::kLF && !f.is_last) f.save_as_reference = br.Read(2);
    if (f.frame_type != FrameHeader::kLF) {
      bool can_ref = !f.is_last && (f.duration == 0 || f.save_as_reference != 0);
      if (f.frame_type == FrameHeader::kReferenceOnly || (full_frame && f.blend_mode == 0 && can_ref)) f.save_before_ct = br.Bool();
    }
    uint32_t nl = ReadNameLen(br);
    f.name.resize(nl);
    for (auto& ch : f.name) ch = (char)br.Read(8);
    // loop filter
    bool lf_default = br.Bool();
    if (!lf_default) {
      LoopFilter& lf = f.lf;
      lf.gab = br.Bool();
      if (lf.gab) {
        bool custom = br.Bool();
        if (custom)
          for (int c = 0; c < 3; c++) { lf.gab_w1[c] = br.F16(); lf.gab_w2[c] = br.F16(); }
      }
      lf.epf_iters = br.Read(2);
      if (lf.epf_iters) {
        if (f.encoding == 0) {
          if (br.Bool()) for (int i = 0; i < 8; i++) lf.epf_sharp_lut[i] = br.F16();
        }
        if (br.Bool()) {
          for (int i = 0; i < 3; i++) lf.epf_channel_scale[i] = br.F16();
          br.Read(32);
        }
        if (br.Bool()) {
          if (f.encoding == 0) lf.epf_quant_mul = br.F16();
          lf.epf_pass0_sigma_scale = br.F16();
          lf.epf_pass2_sigma_scale = br.F16();
          lf.epf_border_sad_mul = br.F16();
        }
        if (f.encoding == 1) lf.epf_sigma_for_modular = br.F16();
      }
      uint64_t ext = br.U64();
      JXO_CHECK(ext == 0, "loop filter extensions");
    }
    uint64_t ext = br.U64();
    if (ext) {
      uint64_t total = 0;
      for (int i = 0; i < 64; i++) if (ext >> i & 1) total += br.U64();
      br.Skip(total);
    }
  }
  f.Derive(m);
  JXO_CHECK(!br.overrun, "truncated frame header");
}

void WriteFrameHeader(BitWriter& bw, const ImageMetadata& m, const FrameHeader& f) {
  LoopFilter dlf;
  bool lf_default = f.lf.gab == dlf.gab && f.lf.epf_iters == dlf.epf_iters && !memcmp(f.lf.gab_w1, dlf.gab_w1, sizeof(dlf.gab_w1)) &&
                    !memcmp(f.lf.gab_w2, dlf.gab_w2, sizeof(dlf.gab_w2));
  bool all_default = f.frame_type == 0 && f.encoding == 0 && f.flags == 0 && m.xyb_encoded && f.upsampling == 1 && m.ec.empty() &&
                     f.x_qm_scale == 3 && f.b_qm_scale == 2 && f.num_passes == 1 && !f.have_crop && f.blend_mode == 0 && f.is_last &&
                     f.name.empty() && lf_default && !m.have_animation;
  bw.Bool(all_default);
  if (all_default) return;
  JXO_CHECK(f.frame_type == 0 && !f.have_crop && f.num_passes >= 1 && f.num_passes <= 3 && f.upsampling == 1 && (f.is_last || m.have_animation),
            "oracle encoder writes regular full frames (several only as an animation)");
  bw.Write(2, f.frame_type);
  bw.Write(1, f.encoding);
  bw.U64(f.flags);
  if (!m.xyb_encoded) bw.Bool(f.do_ycbcr);
  JXO_CHECK(!f.do_ycbcr, "YCbCr not written");
  if (!(f.flags & FrameHeader::kUseLfFrame)) {
    bw.U32(Val(1), Val(2), Val(4), Val(8), 1);
    for (size_t i = 0; i < m.ec.size(); i++) bw.U32(Val(1), Val(2), Val(4), Val(8), 1);
  }
  if (f.encoding == 1) bw.Write(2, f.group_size_shift);
  if (f.encoding == 0 && m.xyb_encoded) { bw.Write(3, f.x_qm_scale); bw.Write(3, f.b_qm_scale); }
  bw.U32(Val(1), Val(2), Val(3), BitsOff(3, 4), f.num_passes);
  if (f.num_passes != 1) {
    bw.U32(Val(0), Val(1), Val(2), BitsOff(1, 3), 0);   // no downsampling brackets: everything Modular goes to the last pass
    for (uint32_t i = 0; i + 1 < f.num_passes; i++) bw.Write(2, f.pass_shift[i]);
  }
  bw.Bool(false);                                    // have_crop
  // blending info (replace), full frame => no source field
  bw.U32(Val(0), Val(1), Val(2), BitsOff(2, 3), 0);
  for (size_t i = 0; i < m.ec.size(); i++) bw.U32(Val(0), Val(1), Val(2), BitsOff(2, 3), 0);
  if (m.have_animation) bw.U32(Val(0), Val(1), Bits(8), Bits(32), f.duration);
  bw.Bool(f.is_last);
  if (!f.is_last) {
    bw.Write(2, 0);   // save_as_reference 0; with a duration and no reference slot nothing more is signalled
    JXO_CHECK(f.duration > 0, "a frame that is not the last needs a duration here");
  }
  bw.U32(Val(0), Bits(4), BitsOff(5, 16), BitsOff(10, 48), (uint32_t)f.name.size());
  for (char ch : f.name) bw.Write(8, (uint8_t)ch);
  bw.Bool(lf_default);
  if (!lf_default) {
    bw.Bool(f.lf.gab);
    if (f.lf.gab) {
      bool custom = memcmp(f.lf.gab_w1, dlf.gab_w1, sizeof(dlf.gab_w1)) || memcmp(f.lf.gab_w2, dlf.gab_w2, sizeof(dlf.gab_w2));
      bw.Bool(custom);
      if (custom) for (int c = 0; c < 3; c++) { bw.F16(f.lf.gab_w1[c]); bw.F16(f.lf.gab_w2[c]); }
    }
    bw.Write(2, f.lf.epf_iters);
    if (f.lf.epf_iters) {
      if (f.encoding == 0) bw.Bool(false);  // sharp lut default
      bw.Bool(false);                       // weights default
      bw.Bool(false);                       // sigma default
      if (f.encoding == 1) bw.F16(f.lf.epf_sigma_for_modular);
    }
    bw.U64(0);
  }
  bw.U64(0);  // extensions
}

// ------------------------------------------------------------------ TOC
static uint32_t CoeffOrderContext(uint32_t v) {
  if (v == 0) return 0;
  return std::min<uint32_t>(FloorLog2(v) + 1, 7);
}

void ReadPermutation(BitReader& br, EntropyReader& rd, size_t skip, size_t size, std::vector<uint32_t>& perm) {
  std::vector<uint32_t> lehmer(size, 0);
  uint32_t end = rd.Read(CoeffOrderContext((uint32_t)size));
  JXO_CHECK(end <= size - skip, "permutation end");
  uint32_t last = 0;
  for (size_t i = skip; i < skip + end; i++) {
    lehmer[i] = rd.Read(CoeffOrderContext(last));
    last = lehmer[i];
    JXO_CHECK(lehmer[i] < size - i, "lehmer code");
  }
  (void)br;
  std::vector<uint32_t> temp(size);
  for (size_t i = 0; i < size; i++) temp[i] = (uint32_t)i;
  perm.resize(size);
  for (size_t i = 0; i < size; i++) {
    perm[i] = temp[lehmer[i]];
    temp.erase(temp.begin() + lehmer[i]);
  }
}

void TokenizePermutation(const std::vector<uint32_t>& perm, size_t skip, std::vector<Token>& out) {
  const size_t size = perm.size();
  std::vector<uint32_t> lehmer(size, 0), temp(size);
  for (size_t i = 0; i < size; i++) temp[i] = (uint32_t)i;
  for (size_t i = 0; i < size; i++) {
    size_t pos = 0;
    while (temp[pos] != perm[i]) pos++;
    lehmer[i] = (uint32_t)pos;
    temp.erase(temp.begin() + pos);
  }
  for (size_t i = 0; i < skip; i++) JXO_CHECK(lehmer[i] == 0, "permutation moves a fixed entry");
  size_t end = size;
  while (end > skip && lehmer[end - 1] == 0) end--;
  out.emplace_back(CoeffOrderContext((uint32_t)size), (uint32_t)(end - skip));
  uint32_t last = 0;
  for (size_t i = skip; i < end; i++) {
    out.emplace_back(CoeffOrderContext(last), lehmer[i]);
    last = lehmer[i];
  }
}

void ReadToc(BitReader& br, size_t n, Toc& toc) {
  std::vector<uint32_t> perm;
  bool permuted = br.Bool();
  if (permuted) {
    EntropyCode code;
    DecodeHistograms(br, 8, code);
    EntropyReader rd;
    rd.Init(code, br);
    ReadPermutation(br, rd, 0, n, perm);
    JXO_CHECK(rd.CheckFinal(), "TOC permutation final state");
  }
  br.AlignByte();
  toc.sizes.resize(n);
  for (auto& s : toc.sizes) s = br.U32(Bits(10), BitsOff(14, 1024), BitsOff(22, 17408), BitsOff(30, 4211712));
  br.AlignByte();
  std::vector<uint64_t> phys(n + 1, 0);
  for (size_t i = 0; i < n; i++) phys[i + 1] = phys[i] + toc.sizes[i];
  toc.offsets.resize(n);
  toc.logical_size.resize(n);
  for (size_t i = 0; i < n; i++) {
    size_t p = permuted ? perm[i] : i;
    toc.offsets[i] = phys[p];
    toc.logical_size[i] = toc.sizes[p];
  }
  JXO_CHECK(!br.overrun, "truncated TOC");
}

void WriteToc(BitWriter& bw, const std::vector<uint32_t>& sizes) {
  bw.Bool(false);
  bw.AlignByte();
  for (auto s : sizes) bw.U32(Bits(10), BitsOff(14, 1024), BitsOff(22, 17408), BitsOff(30, 4211712), s);
  bw.AlignByte();
}

// ------------------------------------------------------------------ container
static const uint8_t kContainerSig[12] = {0, 0, 0, 0xC, 'J', 'X', 'L', ' ', 0xD, 0xA, 0x87, 0xA};

int SignatureCheck(const uint8_t* data, size_t size) {
  if (size >= 2 && data[0] == 0xFF && data[1] == 0x0A) return 1;
  if (size >= 12 && !memcmp(data, kContainerSig, 12)) return 2;
  return 0;
}

static uint32_t BE32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }
static void PutBE32(std::vector<uint8_t>& v, uint32_t x) {
  v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}

void ParseContainer(const uint8_t* data, size_t size, ContainerInfo& out) {
  out = ContainerInfo();
  int sig = SignatureCheck(data, size);
  JXO_CHECK(sig != 0, "not a JPEG XL file");
  if (sig == 1) {
    out.codestream.assign(data, data + size);
    return;
  }
  out.is_container = true;
  size_t pos = 0;
  bool have_exif = false;
  while (pos + 8 <= size) {
    uint64_t box = BE32(data + pos);
    const uint8_t* type = data + pos + 4;
    size_t hdr = 8;
    if (box == 1) {
      JXO_CHECK(pos + 16 <= size, "truncated box header");
      box = ((uint64_t)BE32(data + pos + 8) << 32) | BE32(data + pos + 12);
      hdr = 16;
    } else if (box == 0) {
      box = size - pos;
    }
    JXO_CHECK(box >= hdr && pos + box <= size, "box size out of range");
    const uint8_t* payload = data + pos + hdr;
    size_t plen = box - hdr;
    if (!memcmp(type, "jxlc", 4)) {
      out.codestream.insert(out.codestream.end(), payload, payload + plen);
    } else if (!memcmp(type, "jxlp", 4)) {
      JXO_CHECK(plen >= 4, "jxlp box too small");
      out.codestream.insert(out.codestream.end(), payload + 4, payload + plen);
    } else if (!memcmp(type, "Exif", 4)) {
      if (!have_exif) { out.exif.assign(payload, payload + plen); have_exif = true; }
    } else if (!memcmp(type, "xml ", 4)) {
      out.xml.emplace_back(payload, payload + plen);
    } else if (!memcmp(type, "brob", 4)) {
      out.has_brob = true;
    }
    pos += box;
  }
  JXO_CHECK(!out.codestream.empty(), "container without codestream");
}

std::vector<uint8_t> WriteContainer(const std::vector<uint8_t>& cs, const uint8_t* exif, size_t exif_size, const uint8_t* xmp,
                                    size_t xmp_size) {
  std::vector<uint8_t> out(kContainerSig, kContainerSig + 12);
  static const uint8_t ftyp[20] = {0, 0, 0, 0x14, 'f', 't', 'y', 'p', 'j', 'x', 'l', ' ', 0, 0, 0, 0, 'j', 'x', 'l', ' '};
  out.insert(out.end(), ftyp, ftyp + 20);
  auto box = [&](const char* type, const uint8_t* p, size_t n) {
    if (n + 8 > 0xFFFFFFFFull) {
      PutBE32(out, 1);
      out.insert(out.end(), type, type + 4);
      PutBE32(out, (uint32_t)((n + 16) >> 32));
      PutBE32(out, (uint32_t)(n + 16));
    } else {
      PutBE32(out, (uint32_t)(n + 8));
      out.insert(out.end(), type, type + 4);
    }
    out.insert(out.end(), p, p + n);
  };
  if (exif && exif_size) box("Exif", exif, exif_size);
  if (xmp && xmp_size) box("xml ", xmp, xmp_size);
  box("jxlc", cs.data(), cs.size());
  return out;
}

}  // namespace jxo
