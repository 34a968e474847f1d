"""GPU tests of the drop-in boundary: LoadImage driven exactly like src/Interop/JpegXLNative.cs drives it."""
import numpy as np
import pytest

from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth

pytestmark = pytest.mark.gpu


def test_load_image_callback_order_and_pixels(oracle):
    img = synth(512, 512, 1)  # BASELINE.json configs[0] shape
    exif = b"\0\0\0\0II*\0" + bytes(range(64))
    xmp = b"<x:xmpmeta xmlns:x='adobe:ns:meta/'/>"
    data = oracle.encode(img, exif=exif, xmp=xmp)
    got = api.load_image(data)
    assert got.trace == ["setBasicInfo", "setKnownColorProfile", "setExif", "setXmp", "setLayerData"]
    assert (got.width, got.height, got.format, got.has_transparency) == (512, 512, "Rgb", True)
    assert got.channel_representation == 0 and got.known_profile == "Srgb"
    assert got.exif == exif and got.xmp == xmp and got.layer_name is None
    ref = oracle.decode(data).pixels
    d = np.abs(got.pixels.astype(int) - ref.astype(int))
    assert d.max() <= 1 and (got.pixels[..., 3] == img[..., 3]).all()


def test_load_image_rgb_and_gray(oracle):
    img = synth(300, 300, 2)
    got = api.load_image(oracle.encode(np.ascontiguousarray(img[..., :3])))
    assert got.pixels.shape == (300, 300, 3) and not got.has_transparency  # opaque input => 3 bytes/pixel (SURVEY §8b iii)
    got = api.load_image(oracle.encode(np.ascontiguousarray(img[..., 1:2])))
    assert got.pixels.shape == (300, 300, 1) and got.format == "Gray" and got.known_profile == "GraySrgbTRC"


def test_callback_failures_map_to_reference_statuses(oracle):
    data = oracle.encode(synth(300, 300, 3), exif=b"\0\0\0\0II*\0abcd")
    for cb, status in (("setKnownColorProfile", "CreateMetadataError"), ("setExif", "CreateMetadataError"), ("setLayerData", "CreateLayerError")):
        with pytest.raises(api.JxlError) as e:
            api.load_image(data, fail_at=cb)
        assert e.value.status == status


def test_broken_streams_fail_loudly(oracle):
    data = oracle.encode(synth(300, 300, 4))
    with pytest.raises(api.FormatError) as e:
        api.load_image(data[: len(data) // 2])
    assert e.value.status == "DecodeError" and str(e.value)
    with pytest.raises(api.FormatError) as e:
        api.load_image(b"not a jxl file at all")
    assert e.value.status == "InvalidFileSignature"


@pytest.mark.gpu
def test_load_image_with_brotli_compressed_metadata(oracle):
    """`brob` boxes: the callbacks see the decompressed Exif / XMP in the usual order (Decoder/JxlDecoder.cpp:435,687-784)."""
    import brob_util
    img = synth(320, 200, 5)
    exif = b"\0\0\0\0II*\0" + bytes(range(128)) * 4
    xmp = b"<x:xmpmeta xmlns:x='adobe:ns:meta/'>" + b"0123456789" * 100 + b"</x:xmpmeta>"
    plain = oracle.encode(img, exif=exif, xmp=xmp)
    packed = brob_util.compress_metadata_boxes(plain)
    if packed is None:
        pytest.skip("no Brotli encoder in this image")
    a, b = api.load_image(plain), api.load_image(packed)
    assert b.trace == ["setBasicInfo", "setKnownColorProfile", "setExif", "setXmp", "setLayerData"]
    assert b.exif == exif and b.xmp == xmp
    assert np.array_equal(a.pixels, b.pixels)


@pytest.mark.parametrize("lossless", [False, True])
def test_first_frame_of_an_animation_wins(oracle, lossless):
    """Multi-frame files: the reference stops after the first full image (Decoder/JxlDecoder.cpp:398-400; HasAnimation / HasMultipleFrames
    exist in the status enum and are never returned).  Frames 2 and 3 of the test file hold a different picture."""
    img = synth(300, 280, 62)
    kw = dict(lossless=True) if lossless else dict(distance=1.0)
    one = api.load_image(oracle.encode(img, **kw))
    ani = api.load_image(oracle.encode(img, animation_frames=3, **kw))
    assert ani.trace.count("setLayerData") == 1
    assert ani.pixels.shape == one.pixels.shape and (ani.pixels == one.pixels).all()
    if lossless:
        assert (ani.pixels == img).all()


def test_metadata_callbacks_follow_box_order_and_empty_payloads(oracle):
    """Decoder/JxlDecoder.cpp:686-782 reports a box when the library completes it: file order, the first Exif box only, every `xml `
    box, and a zero-length payload is still reported."""
    import struct
    import brob_util
    plain = oracle.encode(synth(64, 48, 1), exif=b"\0\0\0\0II*\0later", xmp=b"<a/>")
    head, tail = b"", b""
    for typ, payload, raw in brob_util.boxes(plain):
        if typ in (b"Exif", b"xml "):
            continue
        if typ in (b"JXL ", b"ftyp"):
            head += raw
        else:
            tail += raw
    box = lambda t, p: struct.pack(">I4s", 8 + len(p), t) + p
    # xml first, then an EMPTY Exif box (which wins over the later one), then another xml
    data = head + box(b"xml ", b"<first/>") + box(b"Exif", b"") + box(b"Exif", b"\0\0\0\0II*\0later") + box(b"xml ", b"<second/>") + tail
    got = api.load_image(data)
    assert got.trace == ["setBasicInfo", "setKnownColorProfile", "setXmp", "setExif", "setXmp", "setLayerData"]
    assert got.exif == b"" and got.xmp == b"<first/>"
