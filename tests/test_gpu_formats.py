"""GPU parity for streams of more than 8 bits per sample: the host is told Uint16 and receives u16 samples
(reference: src/JxlFileTypeIO/Decoder/JxlDecoder.cpp:510-556, ImageChannelRepresentation Common.h:33-39)."""
import numpy as np
import pytest

from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth, synth16

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nch", [1, 2, 3, 4])
def test_lossless_16_bit_is_exact(oracle, nch):
    img = synth16(300, 270, 7)
    px = {1: img[..., 1], 2: img[..., [1, 3]], 3: img[..., :3], 4: img}[nch]
    data = oracle.encode(px, lossless=True, bits=16, lossless_predictor=5, lossless_tree=1)
    got = api.load_image(data)
    assert got.channel_representation == 1 and got.pixels.dtype == np.uint16      # Uint16
    assert got.format == ("Gray" if nch <= 2 else "Rgb") and got.has_transparency == (nch in (2, 4))
    want = px if px.ndim == 3 else px[..., None]
    assert np.array_equal(got.pixels, want)                                        # ground truth: lossless
    assert np.array_equal(got.pixels, oracle.decode(data).pixels)


@pytest.mark.parametrize("bits", [10, 12])
def test_lossless_intermediate_depths_scale_to_16_bit(oracle, bits):
    px = synth16(260, 200, 3, bits)
    data = oracle.encode(px, lossless=True, bits=bits)
    got, ref = api.load_image(data), oracle.decode(data)
    assert got.channel_representation == 1
    # value * 65535 / (2^bits - 1) through float32 on both sides: identical up to the rounding of one multiply-add
    assert np.abs(got.pixels.astype(np.int32) - ref.pixels.astype(np.int32)).max() <= 1
    exact = np.round(px.astype(np.float64) * 65535 / ((1 << bits) - 1))
    assert np.abs(got.pixels.astype(np.float64) - exact).max() <= 1


@pytest.mark.parametrize("w,h", [(300, 270), (96, 64)])
def test_lossy_16_bit_matches_oracle(oracle, w, h):
    px = synth16(w, h, 5)
    data = oracle.encode(px, distance=1.0, bits=16)
    got, ref = api.load_image(data), oracle.decode(data)
    assert got.channel_representation == 1 and got.pixels.shape == (h, w, 4)
    d = np.abs(got.pixels.astype(np.int32) - ref.pixels.astype(np.int32))
    # float32 summation order / hardware transcendentals: a few 16-bit steps on the colour channels (0.02 of an 8-bit step), alpha exact
    assert d[..., :3].max() <= 48 and (d[..., :3] > 8).mean() < 0.002
    assert np.array_equal(got.pixels[..., 3], px[..., 3])
    # and the decode is close to the source at 16-bit precision
    assert np.abs(got.pixels[..., :3].astype(np.float64) - px[..., :3]).mean() < 6 * 257   # distance 1.0 on noisy content


def test_8_bit_streams_still_decode_to_u8(oracle):
    """8-bit streams still come back as u8 (regression guard for the shared output path)."""
    img = synth(200, 150, 2)
    got = api.load_image(oracle.encode(img, distance=1.0))
    assert got.channel_representation == 0 and got.pixels.dtype == np.uint8


def test_16_bit_batch_entry_point(oracle):
    """jxlhip_decode_batch with a u16 output buffer (bytes_per_sample from jxlhip_peek)."""
    import torch
    px = synth16(280, 260, 9)[..., :3]
    data = oracle.encode(px, lossless=True, bits=16, lossless_predictor=5, lossless_tree=1)
    info = api.peek(data)
    assert info.bytes_per_sample == 2 and info.num_channels == 3
    out = torch.zeros(info.width * info.height * info.num_channels * 2, dtype=torch.uint8, device="cuda")
    dec = api.Decoder(0)
    st = dec.decode_batch([data], [out.data_ptr()], None, synchronize=True)
    assert st[0] == 0
    got = out.cpu().numpy().view(np.uint16).reshape(info.height, info.width, 3)
    assert np.array_equal(got, px)


_ORIENT = {1: lambda a: a, 2: lambda a: a[:, ::-1], 3: lambda a: a[::-1, ::-1], 4: lambda a: a[::-1], 5: lambda a: a.transpose(1, 0, 2),
           6: lambda a: np.rot90(a, -1), 7: lambda a: a[::-1, ::-1].transpose(1, 0, 2), 8: lambda a: np.rot90(a, 1)}


@pytest.mark.parametrize("orientation", [2, 3, 4, 5, 6, 7, 8])
def test_orientation_is_applied(oracle, orientation):
    """The decoder library behind the reference returns the image as displayed (keep_orientation is never requested,
    Decoder/DecoderContext.cpp:98-109): flips / rotations / transpositions per the EXIF numbering, sides swapped for 5..8."""
    img = synth(330, 270, 11)
    data = oracle.encode(img, lossless=True, orientation=orientation, lossless_predictor=5, lossless_tree=1)
    got = api.load_image(data)
    want = _ORIENT[orientation](img)
    assert (got.height, got.width) == want.shape[:2]
    assert np.array_equal(got.pixels, want)                      # ground truth: numpy flips / rot90 of the source
    assert np.array_equal(got.pixels, oracle.decode(data).pixels)
    info = api.peek(data)
    assert (info.height, info.width) == want.shape[:2]


def test_orientation_lossy_and_16_bit(oracle):
    img = synth(300, 260, 12)
    plain = api.load_image(oracle.encode(img, distance=1.0)).pixels
    got = api.load_image(oracle.encode(img, distance=1.0, orientation=6)).pixels
    assert np.array_equal(got, np.rot90(plain, -1))
    px = synth16(130, 90, 4)
    got = api.load_image(oracle.encode(px, lossless=True, bits=16, orientation=8)).pixels
    assert np.array_equal(got, np.rot90(px, 1))


def test_golden_fixtures_of_the_deeper_formats(oracle):
    """The committed 16-bit / 12-bit / oriented fixtures through LoadImage against the oracle's decode."""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name, tol in (("rgba16_96x64_lossless", 0), ("rgba12_96x64_d1", 48), ("rgb_80x60_orient6_d1", 1)):
        data = open(os.path.join(gold, name + ".jxl"), "rb").read()
        got, ref = api.load_image(data).pixels, oracle.decode(data).pixels
        assert got.shape == ref.shape and got.dtype == ref.dtype, name
        assert np.abs(got.astype(np.int32) - ref.astype(np.int32)).max() <= tol, name


def test_mixed_batch_of_formats(oracle):
    """One jxlhip_decode_batch call with 8-bit lossy, 16-bit lossless, 12-bit lossy, an oriented frame and an 8-bit lossless frame:
    every image carries its own sample depths / orientation through the shared kernels."""
    import torch
    a = synth(300, 260, 21)
    b = synth16(200, 180, 22)
    files = [oracle.encode(a, distance=1.0), oracle.encode(b, lossless=True, bits=16, lossless_predictor=5, lossless_tree=1),
             oracle.encode(synth16(260, 140, 23, 12), distance=1.5, bits=12), oracle.encode(a[..., :3], distance=2.0, orientation=5),
             oracle.encode(a, lossless=True)]
    infos = [api.peek(f) for f in files]
    outs = [torch.zeros(i.width * i.height * i.num_channels * i.bytes_per_sample, dtype=torch.uint8, device="cuda") for i in infos]
    dec = api.Decoder(0)
    st = dec.decode_batch(files, [o.data_ptr() for o in outs], None, synchronize=True)
    assert all(s == 0 for s in st), st
    for f, i, o in zip(files, infos, outs):
        ref = oracle.decode(f).pixels
        got = o.cpu().numpy().view(np.uint16 if i.bytes_per_sample == 2 else np.uint8).reshape(i.height, i.width, i.num_channels)
        assert got.shape == ref.shape
        d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        assert d.max() <= (48 if i.bytes_per_sample == 2 else 1), (d.max(), i.bytes_per_sample)
    dec.close()


def _float_image(w, h, seed):
    rng = np.random.default_rng(seed)
    img = synth(w, h, seed).astype(np.float32) / 255
    img = (img + rng.uniform(0, 1e-3, img.shape)).astype(np.float32)
    img[0, 0, :3] = [1.5, 0.0, 0.75]   # out-of-range samples survive float types
    return img


@pytest.mark.parametrize("kind", [32, 16])
def test_float_samples_lossless(oracle, kind):
    """Float-sample streams: Float32 / Float16 representation (Decoder/JxlDecoder.cpp:512-535), samples are the coded bit patterns."""
    img = _float_image(200, 150, 31)
    px = img if kind == 32 else img.astype(np.float16)
    data = oracle.encode(px, lossless=True, float_samples=kind, lossless_predictor=5, lossless_tree=1)
    got = api.load_image(data)
    assert got.channel_representation == (3 if kind == 32 else 2) and got.pixels.dtype == px.dtype
    bits = np.uint32 if kind == 32 else np.uint16
    assert np.array_equal(got.pixels.view(bits), px.view(bits))                      # ground truth: lossless, bit patterns
    assert np.array_equal(got.pixels.view(bits), oracle.decode(data).pixels.view(bits))


@pytest.mark.parametrize("kind", [32, 16])
def test_float_samples_lossy(oracle, kind):
    img = _float_image(300, 260, 32)
    px = img if kind == 32 else img.astype(np.float16)
    data = oracle.encode(px, distance=1.0, float_samples=kind)
    got, ref = api.load_image(data), oracle.decode(data)
    assert got.pixels.dtype == ref.pixels.dtype == px.dtype
    d = np.abs(got.pixels[..., :3].astype(np.float32) - ref.pixels[..., :3].astype(np.float32))
    assert d.max() < (1e-3 if kind == 32 else 2e-3)                                  # float32 summation order / half rounding
    bits = np.uint32 if kind == 32 else np.uint16
    assert np.array_equal(got.pixels[..., 3].view(bits), px[..., 3].view(bits))      # alpha is coded losslessly
    assert np.abs(got.pixels[..., :3].astype(np.float32) - img[..., :3]).mean() < 6 / 255


@pytest.mark.parametrize("colour,profile", [(0, "Srgb"), (1, "DisplayP3"), (2, "Rec709"), (3, "Rec2020Linear"), (4, "Rec2020PQ"), (5, "LinearSrgb")])
def test_enumerated_colour_encodings(oracle, colour, profile):
    """The encodings the reference's host knows by name (SetProfileFromColorEncoding, Decoder/JxlDecoder.cpp:36-108): the pixels come
    back in the image's OWN space (primaries folded into the XYB inverse, its transfer function applied)."""
    img = synth(300, 260, 40 + colour)
    data = oracle.encode(img, distance=1.0, colour=colour)
    got, ref = api.load_image(data), oracle.decode(data)
    assert got.known_profile == profile
    d = np.abs(got.pixels.astype(np.int32) - ref.pixels.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.004
    assert np.abs(got.pixels[..., :3].astype(np.float64) - img[..., :3]).mean() < 6.5      # and close to the source in that space
    # lossless frames carry the samples of their space untouched; only the announced profile differs
    data = oracle.encode(img, lossless=True, colour=colour, lossless_predictor=5, lossless_tree=1)
    got = api.load_image(data)
    assert got.known_profile == profile and np.array_equal(got.pixels, img)


def test_pq_16_bit_and_float(oracle):
    px = synth16(200, 160, 50)
    data = oracle.encode(px, distance=1.0, colour=4, bits=16)
    got, ref = api.load_image(data), oracle.decode(data)
    assert got.known_profile == "Rec2020PQ" and got.pixels.dtype == np.uint16
    d = np.abs(got.pixels[..., :3].astype(np.int32) - ref.pixels[..., :3].astype(np.int32))
    # the PQ curve is very steep near black: float32 summation-order differences of the linear values (1e-5) are amplified there
    assert d.max() <= 1024 and (d > 32).mean() < 0.003 and np.median(d) <= 2
    f = (synth(200, 160, 51).astype(np.float32) / 255)
    data = oracle.encode(f, distance=1.0, colour=3, float_samples=32)
    got, ref = api.load_image(data), oracle.decode(data)
    assert got.known_profile == "Rec2020Linear" and got.pixels.dtype == np.float32
    assert np.abs(got.pixels[..., :3] - ref.pixels[..., :3]).max() < 2e-3


def test_unnamed_colour_encodings_are_refused(oracle):
    """HLG (like DCI, gamma, custom primaries) would go through the reference's ICC route: refused loudly, not decoded as sRGB."""
    for lossless in (False, True):
        data = oracle.encode(synth(64, 48, 3), distance=1.0, lossless=lossless, colour=6)
        st, _, msg = api.parse_check(data)
        assert st == "DecodeError" and "ICC" in msg
        with pytest.raises(api.JxlError) as e:
            api.load_image(data)
        assert e.value.status == "DecodeError"


def _premultiplied(img, maxv):
    """Straight RGBA (integers) -> colour samples multiplied by alpha / maxv, rounded: what a file with associated alpha stores."""
    a = img[..., -1:].astype(np.float64) / maxv
    out = img.copy()
    out[..., :-1] = np.round(img[..., :-1].astype(np.float64) * a).astype(img.dtype)
    return out


@pytest.mark.parametrize("lossless", [True, False], ids=["lossless", "lossy"])
@pytest.mark.parametrize("bits", [8, 16])
def test_premultiplied_alpha_is_undone(oracle, lossless, bits):
    """Associated alpha: the reference asks its library for un-premultiplied samples (JxlDecoderSetUnpremultiplyAlpha(TRUE),
    Decoder/JxlDecoder.cpp:233).  The division is done where the samples are written, on the encoded values, with alpha clamped
    from below at 2^-26 [recalled from the library; unpinned].  Checked against the oracle, and against ground truth: where alpha
    is large the straight colour comes back to within the rounding of the premultiplied samples."""
    img = synth(300, 200, 12) if bits == 8 else synth16(300, 200, 12)
    maxv = (1 << bits) - 1
    img[10:40, 10:60, 3] = 0                 # fully transparent: division by the clamp, samples saturate or stay 0
    img[50:60, :, 3] = maxv
    pm = _premultiplied(img, maxv)
    data = oracle.encode(pm, lossless=lossless, bits=bits, premultiplied_alpha=True, lossless_predictor=5, lossless_tree=1)
    got, ref = api.load_image(data), oracle.decode(data)
    assert got.has_transparency and np.array_equal(got.pixels[..., 3], img[..., 3])
    d = np.abs(got.pixels.astype(np.int64) - ref.pixels.astype(np.int64))
    if lossless:
        assert d.max() <= 1 and (d > 0).mean() < 1e-3      # one float32 multiply on both sides
        # ground truth where alpha >= half: |straight - recovered| <= about maxv / (2 alpha) + 1 from the rounding of the stored sample
        m = img[..., 3] >= maxv // 2
        err = np.abs(got.pixels[..., :3].astype(np.float64) - img[..., :3].astype(np.float64))[m]
        assert err.max() <= 2.0
        assert (got.pixels[10:40, 10:60, :3] == 0).all()   # 0 / clamp = 0
    else:
        tol = 1 if bits == 8 else 48
        # the division amplifies float differences by 1 / alpha: compare where alpha is at least a quarter
        m = img[..., 3] >= maxv // 4
        assert d[..., :3][m].max() <= 4 * tol and (d[..., :3][m] > tol).mean() < 0.002
    # the same file with the flag off keeps the stored (premultiplied) samples
    plain = api.load_image(oracle.encode(pm, lossless=True, bits=bits, lossless_predictor=5, lossless_tree=1))
    assert np.array_equal(plain.pixels, pm)
