"""GPU tests of Modular (lossless) frames: bit-exact against the source pixels (the ground truth of a lossless codec) and against the
oracle's decoder, for streams written by the oracle's encoder and by the product's own SaveImage(lossless)."""
import numpy as np
import pytest

from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth

pytestmark = pytest.mark.gpu


def bgra_of(rgba):
    return np.ascontiguousarray(rgba[..., [2, 1, 0, 3]])


@pytest.mark.parametrize("size", [(300, 280), (200, 100), (256, 256), (777, 531), (257, 1)])
@pytest.mark.parametrize("layout", ["rgba", "rgb", "gray", "graya"])
def test_oracle_lossless_streams_decode_bit_exact(oracle, size, layout):
    w, h = size
    img = synth(w, h, 21)
    src = {"rgba": img, "rgb": img[..., :3], "gray": img[..., 1:2], "graya": img[..., [1, 3]]}[layout]
    src = np.ascontiguousarray(src)
    # local-gradient context tree (properties 10 / 11: the per-sample tree walk of the generic lane path), gradient predictor
    data = oracle.encode(src, lossless=True, lossless_tree=1, lossless_predictor=5)
    got = api.load_image(data)
    assert got.pixels.shape == src.shape
    assert (got.pixels == src).all()
    assert (got.pixels == oracle.decode(data).pixels).all()


@pytest.mark.parametrize("size", [(300, 280), (200, 100), (777, 531)])
@pytest.mark.parametrize("layout", ["rgba", "gray"])
def test_weighted_predictor_streams_decode_bit_exact(oracle, size, layout):
    """The oracle's default lossless mode: contexts from the weighted predictor's error (property 15) and the weighted predictor
    itself (predictor 6) - the configuration libjxl uses for photographic content."""
    w, h = size
    img = synth(w, h, 29)
    src = np.ascontiguousarray(img if layout == "rgba" else img[..., 1:2])
    for pred in (6, 5):
        data = oracle.encode(src, lossless=True, lossless_predictor=pred)
        got = api.load_image(data)
        assert got.pixels.shape == src.shape and (got.pixels == src).all(), pred


@pytest.mark.parametrize("size", [(300, 280), (200, 100), (777, 531), (64, 9), (2100, 300)])
@pytest.mark.parametrize("layout", ["rgba", "rgb", "gray"])
def test_squeeze_streams_decode_bit_exact(oracle, size, layout):
    """Squeeze (default parameters: chroma first, then alternating horizontal / vertical steps down to 8 pixels) on top of the
    colour transform: dozens of channels of every size, spread over the global stream, the LF groups and the pass groups."""
    w, h = size
    img = synth(w, h, 37)
    src = np.ascontiguousarray({"rgba": img, "rgb": img[..., :3], "gray": img[..., 1:2]}[layout])
    for kw in (dict(lossless_tree=1, lossless_predictor=5), dict()):   # plain gradient tree, and the weighted-predictor default
        data = oracle.encode(src, lossless=True, lossless_squeeze=True, **kw)
        got = api.load_image(data)
        assert got.pixels.shape == src.shape and (got.pixels == src).all(), kw


@pytest.mark.parametrize("predictor", [0, 1, 2, 3, 4, 7, 8, 9, 10, 11, 12, 13])
def test_every_plain_predictor(oracle, predictor):
    img = synth(300, 260, 23)
    data = oracle.encode(img, lossless=True, lossless_tree=1, lossless_predictor=predictor)
    assert (api.load_image(data).pixels == img).all()


@pytest.mark.parametrize("size,seed", [((300, 280), 3), ((130, 90), 4), ((520, 400), 5)])
def test_save_image_lossless_round_trip(oracle, size, seed):
    w, h = size
    img = synth(w, h, seed)
    data = api.save_image(bgra_of(img), lossless=True)
    assert (oracle.decode(data).pixels == img).all()          # the oracle reads the product's stream back exactly
    got = api.load_image(data)
    assert got.pixels.shape == img.shape and (got.pixels == img).all()
    opaque = img.copy()
    opaque[..., 3] = 255
    data = api.save_image(bgra_of(opaque), lossless=True)
    got = api.load_image(data)
    assert got.pixels.shape == (h, w, 3) and (got.pixels == opaque[..., :3]).all()
    gray = opaque.copy()
    gray[..., 0] = gray[..., 2] = gray[..., 1]
    got = api.load_image(api.save_image(bgra_of(gray), lossless=True))
    assert got.pixels.shape == (h, w, 1) and (got.pixels[..., 0] == gray[..., 1]).all()


def test_4k_lossless_bit_exact(oracle):
    """BASELINE.json configs[4] shape: 3840x2160 Modular lossless decode, bit-exact check (Squeeze / weighted predictor: next)."""
    img = synth(3840, 2160, 2)
    data = api.save_image(bgra_of(img), lossless=True)
    got = api.load_image(data)
    assert got.pixels.shape == img.shape and (got.pixels == img).all()
    rgb = np.ascontiguousarray(img[..., :3])
    data = oracle.encode(rgb, lossless=True, lossless_tree=1, lossless_predictor=5)
    assert (api.load_image(data).pixels == rgb).all()
    # Squeeze + MA-tree (weighted) predictor, as BASELINE.json words the config
    data = oracle.encode(rgb, lossless=True, lossless_squeeze=True)
    assert (api.load_image(data).pixels == rgb).all()


def test_corrupt_lossless_stream_is_reported(oracle):
    data = bytearray(oracle.encode(synth(300, 280, 7), lossless=True, lossless_tree=1, lossless_predictor=5))
    data[len(data) // 2] ^= 0x55
    data[len(data) // 2 + 1] ^= 0xAA
    with pytest.raises(api.JxlError):
        api.load_image(bytes(data))


def _palette_image(w, h, ncol, nch, seed):
    rng = np.random.default_rng(seed)
    cols = rng.integers(0, 256, (ncol, nch), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    idx = (xx // 7 + yy // 5 + (xx * yy) // 977) % ncol
    return np.ascontiguousarray(cols[idx])


@pytest.mark.parametrize("case", [(300, 200, 17, 4), (700, 530, 200, 3), (120, 90, 5, 1), (600, 300, 900, 4), (200, 140, 40, 2), (1030, 260, 300, 3)])
@pytest.mark.parametrize("squeeze", [False, True])
def test_palette_streams_decode_bit_exact(oracle, case, squeeze):
    """Palette transform (images of few colours: one channel of indices + the colours in a meta channel of the global stream), alone
    and under Squeeze; one-group and multi-group frames, 1 to 4 channels through one palette."""
    w, h, ncol, nch = case
    img = _palette_image(w, h, ncol, nch, 7 * ncol + nch)
    data = oracle.encode(img, lossless=True, palette=True, lossless_squeeze=squeeze)
    assert len(data) < len(oracle.encode(img, lossless=True, lossless_squeeze=squeeze))   # the transform was really used
    assert (oracle.decode(data).pixels == img).all()
    got = api.load_image(data)
    assert got.pixels.shape == img.shape and (got.pixels == img).all()


def test_palette_with_prefix_codes_and_lz77(oracle):
    img = _palette_image(520, 300, 12, 4, 3)
    data = oracle.encode(img, lossless=True, palette=True, prefix_codes=True, lz77=True)
    assert (api.load_image(data).pixels == img).all()


@pytest.mark.parametrize("kind", ["u16", "u16-extremes", "f32", "f16"])
@pytest.mark.parametrize("squeeze", [False, True], ids=["plain", "squeeze"])
def test_weighted_predictor_with_deep_samples(oracle, kind, squeeze):
    """The weighted predictor on deep samples: 16-bit integers stay inside the 32-bit form of the one-section-per-wavefront decoder
    (csrc/modular_uniform.h), float bit patterns (up to 2^30) leave it - the channel then continues in the 64-bit form.  Both must
    be bit-exact against the source and the oracle, also where the two forms meet inside one channel."""
    from pdn_jpegxl_amd.synth import synth16
    if kind == "u16":
        src = np.ascontiguousarray(synth16(300, 270, 31)[..., :3])
        data = oracle.encode(src, lossless=True, bits=16, lossless_squeeze=squeeze)
    elif kind == "u16-extremes":
        src = np.ascontiguousarray(synth16(200, 120, 32)[..., :3])
        src[20:60, 30:90] = 65535          # hard edges between the extremes: the largest errors a 16-bit image can make
        src[20:60, 90:150] = 0
        src[70:100, ::2] = 65535
        data = oracle.encode(src, lossless=True, bits=16, lossless_squeeze=squeeze)
    else:
        base = synth(200, 150, 33)[..., :3].astype(np.float32) / 255.0
        if kind == "f32":
            base[10:40, 10:60] *= 1e-3      # small magnitudes: bit patterns far below those of the values near one
            src = base.astype(np.float32)
            data = oracle.encode(src, lossless=True, float_samples=32, lossless_squeeze=squeeze)
        else:
            src = base.astype(np.float16)
            data = oracle.encode(src, lossless=True, float_samples=16, lossless_squeeze=squeeze)
    got = api.load_image(data)
    ref = oracle.decode(data).pixels
    assert got.pixels.dtype == src.dtype and got.pixels.shape == src.shape
    assert np.array_equal(got.pixels.view(np.uint8), ref.view(np.uint8))
    assert np.array_equal(got.pixels.view(np.uint8), np.ascontiguousarray(src).view(np.uint8))
