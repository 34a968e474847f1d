"""CPU tests of the oracle (the CPU restatement used as the checker).  No GPU.

There are no reference fixtures to pin against (libjxl absent, reference has no tests), so the
oracle is pinned by (1) ground-truth identities (lossless round trips), (2) float64 closed forms (DCT, XYB),
(3) the committed golden vectors (regression), (4) rate/quality sanity of the lossy path.
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np
import pytest

from pdn_jpegxl_amd.synth import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STRATEGIES = ("DCT8 IDENTITY DCT2X2 DCT4X4 DCT16X16 DCT32X32 DCT16X8 DCT8X16 DCT32X8 DCT8X32 DCT32X16 DCT16X32 DCT4X8 DCT8X4 "
              "AFV0 AFV1 AFV2 AFV3 DCT64X64 DCT64X32 DCT32X64 DCT128X128 DCT128X64 DCT64X128 DCT256X256 DCT256X128 DCT128X256").split()


def psnr(a, b):
    mse = ((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean()
    return 10 * np.log10(255.0 ** 2 / max(mse, 1e-12))


def test_golden_vectors(oracle):
    index = json.load(open(os.path.join(GOLD, "index.json")))
    assert len(index) >= 6
    for name, meta in index.items():
        data = open(os.path.join(GOLD, name + ".jxl"), "rb").read()
        assert hashlib.sha256(data).hexdigest() == meta["jxl_sha256"], name
        dec = oracle.decode(data, num_threads=2)
        w, h = meta["size"] if meta["enc"].get("orientation", 1) < 5 else meta["size"][::-1]   # orientations 5..8 swap the sides
        assert dec.pixels.shape == (h, w, meta["nch"]), name
        assert hashlib.sha256(dec.pixels.tobytes()).hexdigest() == meta["pixels_sha256"], name
        if meta["enc"].get("lossless") and meta["enc"].get("orientation", 1) == 1:
            assert meta["pixels_sha256"] == meta["source_sha256"], name  # bit-exact vs the source image


def test_golden_recipe_reproduces_the_committed_streams(oracle):
    """The oracle's encoder is deterministic (it once read its quantiser biases from an uninitialised header struct): running the
    committed recipe again yields the committed bytes, whatever else the process did before and however many threads encode."""
    sys.path.insert(0, GOLD)
    import make_golden
    index = json.load(open(os.path.join(GOLD, "index.json")))
    scratch = np.random.default_rng(1).integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()   # dirty the heap / stack a little
    assert len(scratch) == 1 << 20
    for name, case in make_golden.CASES.items():
        src = np.ascontiguousarray(make_golden.source(case))
        for threads in (1, 5):
            data = oracle.encode(src, num_threads=threads, **case["enc"])
            assert hashlib.sha256(data).hexdigest() == index[name]["jxl_sha256"], (name, threads)


@pytest.mark.parametrize("size", [(1, 1), (8, 8), (17, 9), (255, 257), (256, 256), (300, 520)])
@pytest.mark.parametrize("nch", [1, 2, 3, 4])
def test_lossless_roundtrip_bit_exact(oracle, size, nch):
    img = synth(size[0], size[1], 21)
    src = np.ascontiguousarray({4: img, 3: img[..., :3], 1: img[..., 1:2], 2: img[..., [1, 3]]}[nch])
    for squeeze in (False, True):
        for pred in (6, 5):
            dec = oracle.decode(oracle.encode(src, lossless=True, lossless_squeeze=squeeze, lossless_predictor=pred))
            assert dec.pixels.shape == src.shape
            assert (dec.pixels == src).all()


def test_lossless_extreme_values(oracle):
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (70, 90, 4), dtype=np.uint8)
    src[:10] = 0
    src[10:20] = 255
    assert (oracle.decode(oracle.encode(src, lossless=True)).pixels == src).all()


@pytest.mark.parametrize("s", [i for i, n in enumerate(STRATEGIES) if not n.startswith("AFV")])
def test_lossy_every_strategy(oracle, s):
    img = synth(520, 300, 1)
    data = oracle.encode(img, distance=1.0, strategy_mode=3, fixed_strategy=s)
    d = oracle.decode(data, want_dump=True)
    used = d.planes["strategy"] & 0x7F
    assert (used == s).mean() > 0.5
    assert psnr(d.pixels[..., :3], img[..., :3]) > 33.0
    assert (d.pixels[..., 3] == img[..., 3]).all()  # alpha is lossless


def test_smooth_image_is_near_exact_with_large_transforms(oracle):
    # low-frequency content: LF->LLF reconstruction must be consistent for every block size
    H, W = 256, 512
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.stack([128 + 90 * np.cos(xx / 60) * np.cos(yy / 45), 100 + 80 * np.sin(xx / 35 + yy / 70), 140 + 60 * np.cos(xx / 25 - yy / 50)], -1)
    img = np.clip(img, 0, 255).astype(np.uint8)
    for s in (4, 5, 18, 21, 24, 22, 26):
        d = oracle.decode(oracle.encode(img, distance=0.5, strategy_mode=3, fixed_strategy=s, gaborish=False, epf_iters=0))
        assert psnr(d.pixels, img) > 49.0, STRATEGIES[s]


def test_rate_distortion_monotone(oracle):
    img = synth(400, 300, 3)[..., :3]
    sizes, q = [], []
    for dist in (0.5, 1.0, 2.0, 4.0, 8.0):
        data = oracle.encode(img, distance=dist)
        sizes.append(len(data))
        q.append(psnr(oracle.decode(data).pixels, img))
    assert sizes == sorted(sizes, reverse=True)
    assert q == sorted(q, reverse=True)
    assert q[1] > 36


def test_container_metadata_roundtrip(oracle):
    img = synth(64, 64, 2)
    exif = b"\0\0\0\0II*\0" + bytes(range(40))
    xmp = b"<x:xmpmeta>hello</x:xmpmeta>"
    d = oracle.decode(oracle.encode(img, exif=exif, xmp=xmp))
    assert d.exif == exif and d.xml == xmp
    bare = oracle.encode(img, container=False)
    assert bare[:2] == b"\xff\x0a"
    assert (oracle.decode(bare).pixels[..., 3] == img[..., 3]).all()


def test_corrupt_streams_raise(oracle):
    data = bytearray(oracle.encode(synth(300, 300, 4)))
    with pytest.raises(oracle.OracleError):
        oracle.decode(bytes(data[: len(data) // 2]))
    with pytest.raises(oracle.OracleError):
        oracle.decode(b"not a jxl file at all")


@pytest.mark.parametrize("shape", [(8, 8), (16, 16), (32, 32), (8, 16), (16, 8), (32, 8), (8, 32), (4, 8), (8, 4), (4, 4), (64, 32), (128, 128)])
def test_idct_matches_float64_closed_form(oracle, shape):
    from scipy.fft import idctn
    R, Cc = shape
    L = oracle.lib()
    L.jxo_idct_stored.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.jxo_dct_stored.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(R * 1000 + Cc)
    coef = rng.standard_normal((R, Cc))  # logical (ky, kx)
    # JPEG XL scaling: x = sum_k s_k c_k cos(...), s_0 = 1, s_k = sqrt(2)  == orthonormal IDCT scaled by sqrt(N) per axis
    ref = idctn(coef, type=2, norm="ortho") * np.sqrt(R * Cc)
    stored = np.ascontiguousarray((coef.T if R >= Cc else coef).astype(np.float32))  # stored layout: short x long
    out = np.empty((R, Cc), np.float32)
    L.jxo_idct_stored(R, Cc, stored.ctypes.data, out.ctypes.data)
    assert np.abs(out - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())
    back = np.empty_like(stored)
    L.jxo_dct_stored(R, Cc, out.ctypes.data, back.ctypes.data)
    assert np.abs(back - stored).max() < 1e-4


def test_opsin_matrix_is_inverse_of_forward():
    fwd = np.array([[0.30, 0.622, 0.078], [0.23, 0.692, 0.078], [0.24342268924547819, 0.20476744424496821, 0.5518098665095536]])
    inv = np.array([[11.031566901960783, -9.866943921568629, -0.16462299647058826],
                    [-3.254147380392157, 4.418770392156863, -0.16462299647058826],
                    [-3.6588512862745097, 2.7129230470588235, 1.9459282392156863]])
    assert np.abs(inv @ fwd - np.eye(3)).max() < 1e-6


def test_synth_is_deterministic_and_band_consistent():
    a = synth(700, 500, 9)
    b = synth(700, 500, 9, y0=100, y1=300)
    assert (a[100:300] == b).all()
    assert a[..., 3].min() < 255 and not (a[..., 0] == a[..., 1]).all()


@pytest.mark.parametrize("bits", [16, 12, 10])
def test_lossless_round_trip_above_8_bits(oracle, bits):
    """Ground truth for the deeper sample types: 16-bit lossless is the identity; 10 / 12 bits come back scaled to u16."""
    rng = np.random.default_rng(bits)
    px = rng.integers(0, 1 << bits, (70, 90, 4)).astype(np.uint16)
    d = oracle.decode(oracle.encode(px, lossless=True, bits=bits))
    assert d.pixels.dtype == np.uint16 and d.pixels.shape == px.shape
    if bits == 16:
        assert np.array_equal(d.pixels, px)
    else:
        exact = np.round(px.astype(np.float64) * 65535 / ((1 << bits) - 1))
        assert np.abs(d.pixels.astype(np.float64) - exact).max() <= 1


def test_lossy_16_bit_header(oracle):
    img = synth(128, 96, 4).astype(np.uint16) * 257
    d = oracle.decode(oracle.encode(img, distance=1.0, bits=16))
    assert d.pixels.dtype == np.uint16
    assert np.abs(d.pixels[..., :3].astype(np.float64) - img[..., :3]).mean() < 6 * 257
    assert np.array_equal(d.pixels[..., 3], img[..., 3])


def test_orientation_against_numpy(oracle):
    """Ground truth for the eight EXIF orientations: numpy flips / rot90 / transposes of the un-oriented decode."""
    img = synth(70, 50, 3)
    base = oracle.decode(oracle.encode(img, lossless=True)).pixels
    ops = {1: lambda a: a, 2: lambda a: a[:, ::-1], 3: lambda a: a[::-1, ::-1], 4: lambda a: a[::-1], 5: lambda a: a.transpose(1, 0, 2),
           6: lambda a: np.rot90(a, -1), 7: lambda a: a[::-1, ::-1].transpose(1, 0, 2), 8: lambda a: np.rot90(a, 1)}
    for o, f in ops.items():
        d = oracle.decode(oracle.encode(img, lossless=True, orientation=o)).pixels
        assert np.array_equal(d, f(base)), o


@pytest.mark.parametrize("kind", [32, 16])
def test_float_samples_round_trip(oracle, kind):
    """binary32 / binary16 sample streams: lossless is the identity on the bit patterns (non-negative samples; the integer
    predictor arithmetic of bit patterns that differ by 2^31 and more - sign changes of binary32 - is outside the tested range)."""
    rng = np.random.default_rng(kind)
    img = (synth(90, 70, 5).astype(np.float32) / 255 + rng.uniform(0, 1e-3, (70, 90, 4))).astype(np.float32)
    img[0, 0, :3] = [1.5, 0.0, 0.75]
    px = img if kind == 32 else img.astype(np.float16)
    bits = np.uint32 if kind == 32 else np.uint16
    d = oracle.decode(oracle.encode(px, lossless=True, float_samples=kind, lossless_predictor=5, lossless_tree=1))
    assert d.pixels.dtype == px.dtype and np.array_equal(d.pixels.view(bits), px.view(bits))
    d = oracle.decode(oracle.encode(px, distance=1.0, float_samples=kind))
    assert d.pixels.dtype == px.dtype
    assert np.abs(d.pixels[..., :3].astype(np.float32) - img[..., :3]).mean() < 6 / 255
    assert np.array_equal(d.pixels[..., 3].view(bits), px[..., 3].view(bits))


@pytest.mark.parametrize("colour", [0, 1, 2, 3, 4, 5])
def test_named_colour_encodings_round_trip(oracle, colour):
    """Pixels handed to the encoder in Display P3 / BT.709 / BT.2020 linear / BT.2020 PQ / linear sRGB come back in that same space."""
    img = synth(120, 90, 5)
    d = oracle.decode(oracle.encode(img, distance=1.0, colour=colour)).pixels
    assert np.abs(d[..., :3].astype(np.float64) - img[..., :3]).mean() < 6.0
    d = oracle.decode(oracle.encode(img, lossless=True, colour=colour)).pixels
    assert np.array_equal(d, img)


@pytest.mark.parametrize("opts", [dict(prefix_codes=True), dict(lz77=True), dict(prefix_codes=True, lz77=True), dict(custom_quant_tables=True),
                                  dict(num_passes=2), dict(num_passes=3), dict(num_passes=3, prefix_codes=True, lz77=True),
                                  dict(custom_orders=True), dict(custom_orders=True, num_passes=2), dict(lf_contexts=True),
                                  dict(lf_contexts=True, custom_orders=True, prefix_codes=True)])
def test_stream_coding_variants_decode_to_the_same_pixels(oracle, opts):
    """Prefix codes, LZ77 and progressive passes change how the same quantised data is written, never the data: the decode equals
    the plain frame's.  (Explicit quantisation tables do change the weights: only a round trip is required there.)"""
    img = synth(530, 300, 41)
    a = oracle.decode(oracle.encode(img, strategy_mode=2, seed=3, **opts)).pixels
    b = oracle.decode(oracle.encode(img, strategy_mode=2, seed=3)).pixels
    if "custom_quant_tables" in opts:
        assert np.abs(a.astype(int) - img.astype(int)).mean() < 6 and (a != b).any()
    else:
        assert (a == b).all()


@pytest.mark.parametrize("nch", [1, 2, 3, 4])
def test_palette_round_trip(oracle, nch):
    rng = np.random.default_rng(nch)
    cols = rng.integers(0, 256, (23, nch), dtype=np.uint8)
    yy, xx = np.mgrid[0:190, 0:310]
    img = np.ascontiguousarray(cols[(xx // 9 + yy // 4) % 23])
    for squeeze in (False, True):
        data = oracle.encode(img, lossless=True, palette=True, lossless_squeeze=squeeze)
        assert (oracle.decode(data).pixels == img).all()
        assert len(data) < len(oracle.encode(img, lossless=True, lossless_squeeze=squeeze))


def test_premultiplied_alpha_oracle_identity(oracle):
    O = oracle
    """Oracle alone (CPU): a lossless stream with associated alpha decodes to samples / max(alpha, 2^-26), i.e. the straight colour up
    to the rounding of the stored premultiplied samples; without the flag the stored samples come back untouched."""
    img = synth(120, 90, 12)
    img[5:20, 5:30, 3] = 0
    a = img[..., 3:].astype(np.float64) / 255
    pm = img.copy()
    pm[..., :3] = np.round(img[..., :3] * a).astype(np.uint8)
    dec = O.decode(O.encode(pm, lossless=True, premultiplied_alpha=True))
    assert np.array_equal(dec.pixels[..., 3], img[..., 3])
    m = img[..., 3] >= 128
    assert np.abs(dec.pixels[..., :3].astype(np.float64) - img[..., :3])[m].max() <= 2.0
    want = np.clip(np.floor(pm[..., :3].astype(np.float32) / 255 * (1 / np.maximum(np.float32(2.0 ** -26), (img[..., 3:] / np.float32(255)).astype(np.float32))) * 255 + 0.5), 0, 255)
    assert np.abs(dec.pixels[..., :3].astype(np.float64) - want).max() <= 1
    assert np.array_equal(O.decode(O.encode(pm, lossless=True)).pixels, pm)
