"""GPU tests of ICC profiles and CMYK through the C-ABI (LoadImage / SaveImage): rows a4, a6, a8, N2, N4 of SURVEY.md 8.
Reference behaviour: Encoder/JxlEncoder.cpp:67-75 (no gray conversion with a profile), :258-268 (profile embedded);
Decoder/JxlDecoder.cpp:596-686 (target-data profile handed to setIccProfile), :110-215 (CMYK: black channel, inverted samples)."""
import numpy as np
import pytest

import icc_util
from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth

pytestmark = pytest.mark.gpu


def bgra_of(rgba):
    return np.ascontiguousarray(rgba[..., [2, 1, 0, 3]])


def psnr(a, b):
    mse = ((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean()
    return 10 * np.log10(255.0 ** 2 / max(mse, 1e-12))


PROFILES = {"p3-para": icc_util.matrix_profile("p3", "srgb-para"), "adobe-gamma": icc_util.matrix_profile("adobe", "gamma2.2"),
            "srgb-table": icc_util.matrix_profile("srgb", "table1.8"), "cmyk": icc_util.cmyk_profile(), "lut": icc_util.lut_rgb_profile()}


@pytest.mark.parametrize("name", list(PROFILES))
@pytest.mark.parametrize("alpha", [False, True])
def test_load_lossless_stream_with_profile(oracle, name, alpha):
    """Original-profile (lossless) streams: any profile rides along, the samples are exact, the host gets setIccProfile."""
    icc = PROFILES[name]
    img = synth(300, 270, 51)
    src = np.ascontiguousarray(img if alpha else img[..., :3])
    data = oracle.encode(src, lossless=True, icc=icc)
    got = api.load_image(data)
    assert got.icc == icc and got.known_profile is None
    assert got.trace.index("setBasicInfo") < got.trace.index("setIccProfile") < got.trace.index("setLayerData")
    assert (got.pixels == src).all()
    with pytest.raises(api.JxlError) as e:
        api.load_image(data, fail_at="setIccProfile")
    assert e.value.status == "CreateMetadataError"


@pytest.mark.parametrize("name", ["p3-para", "cmyk"])
def test_save_lossless_with_profile(oracle, name):
    icc = PROFILES[name]
    img = synth(280, 300, 52)
    data = api.save_image(bgra_of(img), lossless=True, icc=icc)
    od = oracle.decode(data)                                   # the oracle's reader on the product's ICC stream
    assert od.icc == icc and (od.pixels == img).all()
    got = api.load_image(data)
    assert got.icc == icc and (got.pixels == img).all()


def test_gray_conversion_is_suppressed_by_a_profile(oracle):
    """r == g == b everywhere, but an RGB profile must keep describing RGB samples (Encoder/JxlEncoder.cpp:67-75)."""
    img = synth(120, 90, 53)
    img[..., 0] = img[..., 2] = img[..., 1]
    img[..., 3] = 255
    plain = api.load_image(api.save_image(bgra_of(img), lossless=True))
    assert plain.format == "Gray" and plain.pixels.shape[2] == 1
    tagged = api.load_image(api.save_image(bgra_of(img), lossless=True, icc=PROFILES["p3-para"]))
    assert tagged.format == "Rgb" and tagged.pixels.shape[2] == 3 and (tagged.pixels == img[..., :3]).all()


@pytest.mark.parametrize("prim,curve", [("p3", "srgb-para"), ("adobe", "gamma2.2"), ("srgb", "table1.8")])
def test_lossy_round_trip_in_the_profiles_space(prim, curve):
    """Lossy with a matrix / TRC profile: the encoder goes to XYB THROUGH the profile, the decoder comes back into it.  Checked against
    float64 numpy colour conversion: the same picture saved once as sRGB and once converted to the profile's space must decode to
    the same colours."""
    icc = icc_util.matrix_profile(prim, curve)
    img = synth(400, 300, 54)
    img[..., :3] = (img[..., :3].astype(np.int32) * 3 // 4 + 32).astype(np.uint8)   # stay inside every gamut involved
    in_profile = img.copy()
    in_profile[..., :3] = icc_util.srgb_to_profile(img[..., :3], prim, curve)
    a = api.load_image(api.save_image(bgra_of(in_profile), distance=1.0, icc=icc))
    assert a.icc == icc and a.known_profile is None
    assert psnr(a.pixels[..., :3], in_profile[..., :3]) > 33.0 and (a.pixels[..., 3] == img[..., 3]).all()
    b = api.load_image(api.save_image(bgra_of(img), distance=1.0))          # the sRGB twin
    assert b.known_profile == "Srgb"
    twin = icc_util.srgb_to_profile(b.pixels[..., :3], prim, curve)          # its decoded colours, expressed in the profile's space
    d = np.abs(a.pixels[..., :3].astype(int) - twin.astype(int))
    # two independent lossy encodes of inputs that differ by the u8 rounding of the colour conversion: a wrong matrix or curve would
    # show as a systematic shift (mean of several steps), not as a thin tail at edges
    assert d.mean() < 0.8 and np.percentile(d, 99.9) <= 8, (float(d.mean()), int(d.max()))


def test_lossy_save_with_unevaluable_profile_is_refused():
    bgra = bgra_of(synth(96, 64, 55))
    for name in ("cmyk", "lut"):
        with pytest.raises(api.JxlError) as e:
            api.save_image(bgra, distance=1.0, icc=PROFILES[name])
        assert e.value.status == "EncodeError" and "ICC" in str(e.value)
    assert api.save_image(bgra, lossless=True, icc=PROFILES["lut"])           # lossless carries anything


@pytest.mark.parametrize("alpha", [False, True])
def test_cmyk(oracle, alpha):
    """A black extra channel makes the image CMYK (Decoder/JxlDecoder.cpp:110-157); the host gets C M Y K [A] with the ink samples
    inverted (:159-215) and the CMYK profile."""
    rng = np.random.default_rng(56)
    h, w = 260, 300
    base = synth(w, h, 56)
    k = rng.integers(0, 256, (h, w, 1), dtype=np.uint8)
    stored = np.concatenate([base[..., :3], k] + ([base[..., 3:4]] if alpha else []), axis=2)
    icc = PROFILES["cmyk"]
    data = oracle.encode(np.ascontiguousarray(stored), lossless=True, icc=icc, cmyk=True)
    od = oracle.decode(data)
    assert od.cmyk and od.icc == icc
    want = stored.copy()
    want[..., :4] = 255 - want[..., :4]
    assert (od.pixels == want).all()
    got = api.load_image(data)
    assert got.format == "Cmyk" and got.has_transparency == alpha and got.icc == icc
    assert got.pixels.shape == want.shape and (got.pixels == want).all()


def _model(icc):
    import ctypes as C
    L = api.lib()
    L.jxlhip_icc_model.restype = C.c_int32
    L.jxlhip_icc_model.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    model = np.zeros(18, np.float64)
    to_lin = np.zeros(3 * 256, np.float32)
    kind = L.jxlhip_icc_model(icc, len(icc), model.ctypes.data, to_lin.ctypes.data, None)
    return kind, model[:9].reshape(3, 3), to_lin[:256]


ADOBE = ((0.64, 0.33), (0.21, 0.71), (0.15, 0.06))


@pytest.mark.parametrize("colour,prim,white,gamma", [(7, ADOBE, icc_util.D65, 256 / 563.0), (8, icc_util.PRIMARIES["p3"], (0.314, 0.351), 1 / 2.6),
                                                     (9, icc_util.PRIMARIES["srgb"], (0.3457, 0.3585), 1.0)])
def test_encodings_outside_the_named_profiles_get_a_synthesised_profile(oracle, colour, prim, white, gamma):
    """Custom primaries / white points, explicit gamma, DCI: SetProfileFromColorEncoding (Decoder/JxlDecoder.cpp:36-108) knows none of
    them, so the reference hands the host an ICC profile of the target data (:652-682).  Here: decoded into that space (parity with the
    oracle) and described by a synthesised matrix / TRC profile, which is checked by evaluating it again."""
    img = synth(330, 270, 64)
    data = oracle.encode(img, colour=colour)
    ref = oracle.decode(data).pixels
    got = api.load_image(data)
    assert got.known_profile is None and got.icc is not None and got.trace.count("setIccProfile") == 1
    d = np.abs(got.pixels.astype(int) - ref.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3, (int(d.max()), float((d > 0).mean()))
    kind, from_srgb, to_lin = _model(got.icc)
    assert kind == 1
    # the profile's colorants and curve against float64: Bradford-adapted primaries, power law
    def to_d50(p, w):
        m = icc_util.rgb_to_xyz(p, w)
        cone = np.array([[0.8951, 0.2664, -0.1614], [-0.7502, 1.7135, 0.0367], [0.0389, -0.0685, 1.0296]])
        wxyz = np.array([w[0] / w[1], 1.0, (1 - w[0] - w[1]) / w[1]])
        gain = (cone @ np.array([0.96422, 1.0, 0.82521])) / (cone @ wxyz)
        return np.linalg.inv(cone) @ np.diag(gain) @ cone @ m
    want = np.linalg.inv(to_d50(prim, white)) @ to_d50(icc_util.PRIMARIES["srgb"], icc_util.D65)
    assert np.abs(from_srgb - want).max() < 3e-4
    x = np.arange(256) / 255
    assert np.abs(to_lin - x ** (1 / gamma)).max() < 3e-4
