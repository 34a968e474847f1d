"""CPU tests of the product's host side: the C-ABI surface, the host frame parser and its static tables.
Nothing here launches a kernel."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "jxlfiletypeio.h")).read()
    names = re.findall(r"JXLFILETYPEIO_API\s+[\w\s\*]+?\b(\w+)\s*\(", hdr)
    assert {"GetLibJxlVersion", "LoadImage", "SaveImage"} <= set(names)
    L = api.lib()
    for n in names:
        assert hasattr(L, n), n
    for n in api.EXPORTS:
        assert n in names, n


def test_struct_layouts_match_reference_abi():
    # sizes from SURVEY.md §8b (reference Common.h / JxlDecoderTypes.h / JxlEncoderTypes.h, LP64 == LLP64)
    assert C.sizeof(api.BitmapData) == 24
    assert C.sizeof(api.EncoderOptions) == 12
    assert C.sizeof(api.EncoderImageMetadata) == 48
    assert C.sizeof(api.IOCallbacks) == 16
    assert C.sizeof(api.ErrorInfo) == 256
    assert C.sizeof(api.DecoderCallbacks) == 48
    assert api.DECODER_STATUS.index("InvalidFileSignature") == 12 and api.ENCODER_STATUS.index("WriteError") == 5


def test_version_is_packed_like_libjxl():
    major, minor, patch = api.get_libjxl_version()
    assert (major, minor) >= (0, 10)  # API level the reference's call sites need


def test_load_image_parameter_and_signature_errors_need_no_gpu():
    L = api.lib()
    err = api.ErrorInfo()
    assert L.LoadImage(None, b"x", 1, C.byref(err)) == api.DECODER_STATUS.index("NullParameter")
    with pytest.raises(api.FormatError) as e:
        api.load_image(b"\x89PNG\r\n\x1a\n" + b"\0" * 64)
    assert e.value.status == "InvalidFileSignature"
    with pytest.raises(api.FormatError) as e:
        api.load_image(b"\xff\x0a")  # signature only: truncated headers
    assert e.value.status == "DecodeError"


def test_save_image_null_parameters():
    L = api.lib()
    err = api.ErrorInfo()
    assert L.SaveImage(None, None, None, None, C.byref(err), api.ProgressFn()) == api.ENCODER_STATUS.index("NullParameter")


def test_host_parser_accepts_oracle_streams(oracle):
    img = synth(600, 400, 1)
    for kw in (dict(), dict(strategy_mode=2, seed=3), dict(distance=4.0), dict(container=False)):
        data = oracle.encode(img, **kw)
        info = api.peek(data)
        assert (info.width, info.height, info.num_channels, info.has_alpha) == (600, 400, 4, 1)
        assert info.num_groups == 6 and info.num_lf_groups == 1
        st, facts, msg = api.parse_check(data)
        assert st == "Ok", msg
        assert facts[5] == 7425 and facts[7] == 2 + 1 + 6
    rgb = oracle.encode(img[..., :3])
    assert api.peek(rgb).num_channels == 3


def test_host_parser_reports_unsupported_paths_loudly(oracle):
    img = synth(300, 300, 1)
    assert api.parse_check(oracle.encode(img, lossless=True, lossless_squeeze=True))[0] == "Ok"   # Squeeze is decoded
    assert api.parse_check(oracle.encode(img, lossless=True))[0] == "Ok"   # weighted-predictor trees are decoded
    # a stream feature the GPU path does not have: flip the frame header's "noise" flag of a lossy stream (flags is the U64 after
    # frame_type + encoding; the oracle writes flags = 0 as selector 0)
    data = bytearray(oracle.encode(img, container=False))
    info = api.peek(bytes(data))
    assert info.width == 300
    st, _, msg = api.parse_check(bytes(data[:len(data) // 3]))
    assert st != "Ok"
    assert api.parse_check(oracle.encode(synth(64, 64, 1)))[0] == "Ok"   # single-group frames: HfGlobal is parsed after the GPU LF pre-pass


def test_host_parser_rejects_truncated_files(oracle):
    data = oracle.encode(synth(300, 300, 1))
    for cut in (10, 40, 100, len(data) // 2):
        st, _, _ = api.parse_check(data[:cut])
        assert st != "Ok"


def test_metadata_boxes_are_located(oracle):
    exif = b"\0\0\0\0II*\0" + bytes(range(32))
    data = oracle.encode(synth(300, 300, 1), exif=exif, xmp=b"<xmp/>")
    assert api.peek(data).width == 300


def test_brob_boxes_are_decompressed(oracle):
    """Brotli-compressed metadata (reference: JxlDecoderSetDecompressBoxes, Decoder/JxlDecoder.cpp:435) reaches the host under its inner type."""
    import brob_util
    exif = b"\0\0\0\0II*\0" + bytes(range(200)) * 5
    xmp = b"<x:xmpmeta xmlns:x='adobe:ns:meta/'>" + b"abcdefgh" * 300 + b"</x:xmpmeta>"
    plain = oracle.encode(synth(300, 300, 1), exif=exif, xmp=xmp)
    packed = brob_util.compress_metadata_boxes(plain)
    if packed is None:
        pytest.skip("no Brotli encoder in this image")
    assert len(packed) < len(plain) and b"brob" in packed and b"Exif" in packed
    assert api.parse_metadata(plain, 0)[:2] == ("Ok", exif) and api.parse_metadata(plain, 1)[:2] == ("Ok", xmp)
    assert api.parse_metadata(packed, 0)[:2] == ("Ok", exif)
    assert api.parse_metadata(packed, 1)[:2] == ("Ok", xmp)
    assert api.parse_metadata(packed, 2)[1] is None
    # a damaged Brotli stream is a decode error, not a crash and not silently missing metadata
    k = packed.index(b"brob") + 4 + 4 + 10
    bad = packed[:k] + bytes([packed[k] ^ 0x5A, packed[k + 1] ^ 0xA5]) + packed[k + 2:]
    st, payload, msg = api.parse_metadata(bad, 0)
    assert st == "DecodeError" and "Brotli" in msg or payload != exif


def test_first_exif_wins_and_every_xml_box_is_reported(oracle):
    """Decoder/JxlDecoder.cpp:697-719 keeps the first Exif box only; every `xml ` box reaches setXmp (:720-784)."""
    import struct
    import brob_util
    exif1, exif2 = b"\0\0\0\0II*\0first", b"\0\0\0\0II*\0second-one"
    plain = oracle.encode(synth(64, 48, 1), exif=exif1, xmp=b"<a/>")
    out = b""
    for typ, payload, raw in brob_util.boxes(plain):
        out += raw
        if typ == b"Exif":
            out += struct.pack(">I4s", 8 + len(exif2), b"Exif") + exif2
        if typ == b"xml ":
            out += struct.pack(">I4s", 8 + 4, b"xml ") + b"<b/>"
    assert api.parse_metadata(out, 0)[:2] == ("Ok", exif1)
    assert api.parse_metadata(out, 1)[:2] == ("Ok", b"<a/>") and api.parse_metadata(out, 2)[:2] == ("Ok", b"<b/>")
    assert api.parse_metadata(out, 3)[1] is None


def test_static_tables_match_oracle(oracle):
    L = oracle.lib()
    L.jxo_natural_order.restype = C.c_size_t
    L.jxo_natural_order.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
    L.jxo_dequant_table.restype = C.c_size_t
    L.jxo_dequant_table.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    bucket_strategy = [0, 1, 4, 5, 6, 8, 10, 18, 19, 21, 22, 24, 25]
    for o, s in enumerate(bucket_strategy):
        n = L.jxo_natural_order(s, None, 0)
        a = np.empty(n, np.uint32)
        L.jxo_natural_order(s, a.ctypes.data, n)
        b = api.static_table("natural_order", o, np.uint16)
        assert sorted(b.tolist()) == list(range(n))  # a permutation
        assert (a == b).all(), o
    qt = [0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16]
    for s in range(27):
        b = api.static_table("dequant", qt[s], np.float32)
        n = len(b) // 3
        for c in range(3):
            a = np.empty(n, np.float32)
            L.jxo_dequant_table(s, c, a.ctypes.data, n)
            assert np.allclose(a, b[c * n:(c + 1) * n], rtol=1e-6), (s, c)


def test_golden_files_parse(oracle):
    for name in ("rgba_300x280_mix_d2", "rgb_333x257_d1", "gray_270x300_d3"):
        st, facts, msg = api.parse_check(open(os.path.join(GOLD, name + ".jxl"), "rb").read())
        assert st == "Ok", (name, msg)


# ---------------------------------------------------------------- encoder-side host writers (no GPU)
def _selftests():
    import ctypes as C
    L = api.selftest_lib()
    L.jxlhip_selftest_entropy.restype = C.c_int32
    L.jxlhip_selftest_entropy.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32, C.POINTER(api.ErrorInfo)]
    L.jxlhip_selftest_tree.restype = C.c_int32
    L.jxlhip_selftest_tree.argtypes = [C.c_uint32, C.POINTER(api.ErrorInfo)]
    return L


@pytest.mark.parametrize("seed,num_ctx,n,max_clusters,pinned", [
    (1, 1, 1000, 1, 0), (2, 9, 20000, 8, 4), (3, 9, 5000, 8, 0), (4, 300, 100000, 16, 0), (5, 7425, 300000, 64, 0),
    (6, 2, 10, 8, 0), (7, 40, 3, 8, 0), (8, 1, 0, 1, 0), (9, 600, 100000, 255, 0)])
def test_entropy_code_written_by_the_encoder_reads_back(seed, num_ctx, n, max_clusters, pinned):
    """BuildAndWriteCode + WriteTokensHost (host_write.cc) -> ReadCode + SymReader (host_parse.cc): every token value and the
    final ANS state.  Covers the simple and the entropy-coded context map, 1..255 clusters, empty and pinned contexts."""
    import ctypes as C
    err = api.ErrorInfo()
    assert _selftests().jxlhip_selftest_entropy(seed, num_ctx, n, max_clusters, pinned, C.byref(err)) == 0, err.errorMessage


@pytest.mark.parametrize("nlf", [1, 4, 64])
def test_encoder_tree_reads_back(nlf):
    import ctypes as C
    err = api.ErrorInfo()
    assert _selftests().jxlhip_selftest_tree(nlf, C.byref(err)) == 0, err.errorMessage


@pytest.mark.parametrize("w,h,gray,alpha,lossless,epf", [(300, 280, 0, 1, 0, 1), (3840, 2160, 0, 1, 0, 2), (200, 120, 1, 0, 0, 0), (64, 64, 0, 0, 0, 3),
                                                       (777, 531, 0, 1, 1, 0), (2049, 2049, 1, 1, 0, 1), (8, 8, 0, 0, 1, 0), (16384, 16384, 0, 1, 0, 1)])
def test_encoder_headers_read_back(w, h, gray, alpha, lossless, epf):
    """WriteCodestreamHeaders + WriteFrameHeader + WriteToc + WriteContainer (host_write.cc) -> ParseFile (host_parse.cc): geometry,
    channel layout, loop-filter settings and the section table survive the trip."""
    L = api.selftest_lib()
    L.jxlhip_selftest_headers.restype = C.c_size_t
    L.jxlhip_selftest_headers.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
    buf = (C.c_uint8 * (1 << 22))()
    n = L.jxlhip_selftest_headers(w, h, gray, alpha, lossless, epf, 3, buf, len(buf))
    assert n > 0
    info = api.peek(bytes(buf[:n]))
    assert (info.width, info.height) == (w, h)
    assert info.num_channels == (1 if gray else 3) + alpha and info.has_alpha == alpha
    assert info.num_groups == ((w + 255) // 256) * ((h + 255) // 256)
    assert info.num_lf_groups == ((w + 2047) // 2048) * ((h + 2047) // 2048)
    assert info.gaborish == (0 if lossless else 1) and info.epf_iters == (0 if lossless else epf)


def test_a_stale_native_library_is_refused(monkeypatch):
    """The library is git-ignored but travels to the GPU box: one built from other sources than the tree's must not be used silently."""
    from pdn_jpegxl_amd import build as B
    api.lib()
    monkeypatch.setattr(api, "_lib", None)
    monkeypatch.setattr(B, "source_digest", lambda: "0" * 64)
    with pytest.raises(api.NativeLibraryMissing) as e:
        api.lib()
    assert "stale" in str(e.value)


# ---------------------------------------------------------------- hand-assembled headers (no oracle, no product writer)
class _Bits:
    """LSB-first bit packer, as the codestream defines it."""

    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, nbits, value):
        assert 0 <= value < (1 << nbits)
        self.acc |= value << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 0xFF)
            self.acc >>= 8
            self.n -= 8

    def align(self):
        if self.n:
            self.put(8 - self.n, 0)

    def u32(self, dists, value):
        """U32 with four (nbits, offset) distributions: the first that can hold the value."""
        for sel, (nbits, offset) in enumerate(dists):
            if offset <= value < offset + (1 << nbits):
                self.put(2, sel)
                self.put(nbits, value - offset)
                return
        raise AssertionError(value)


def _hand_built_codestream(xsize, ysize, section_bytes=40):
    """Signature, SizeHeader, all-default ImageMetadata and transform data, all-default FrameHeader (one VarDCT frame), TOC - written
    from the published field tables alone."""
    b = _Bits()
    b.put(8, 0xFF); b.put(8, 0x0A)
    ratios = {1: (1, 1), 2: (12, 10), 3: (4, 3), 4: (3, 2), 5: (16, 9), 6: (5, 4), 7: (2, 1)}
    small = xsize % 8 == 0 and ysize % 8 == 0 and xsize <= 256 and ysize <= 256
    b.put(1, int(small))
    size_dists = [(9, 1), (13, 1), (18, 1), (30, 1)]
    if small:
        b.put(5, ysize // 8 - 1)
    else:
        b.u32(size_dists, ysize)
    ratio = next((r for r, (n, d) in ratios.items() if ysize * n // d == xsize), 0)
    b.put(3, ratio)
    if ratio == 0:
        if small:
            b.put(5, xsize // 8 - 1)
        else:
            b.u32(size_dists, xsize)
    b.put(1, 1)      # ImageMetadata.all_default: 8-bit sRGB, XYB encoded, no extra channels
    b.put(1, 1)      # custom transform data: all default
    b.align()
    b.put(1, 1)      # FrameHeader.all_default: regular VarDCT frame, one pass, the only frame
    groups = -(-xsize // 256) * -(-ysize // 256)
    lf_groups = -(-xsize // 2048) * -(-ysize // 2048)
    entries = 1 if groups == 1 else 2 + lf_groups + groups
    b.put(1, 0)      # TOC not permuted
    b.align()
    toc_dists = [(10, 0), (14, 1024), (22, 17408), (30, 4211712)]
    for _ in range(entries):
        b.u32(toc_dists, section_bytes)
    b.align()
    return bytes(b.out) + bytes(section_bytes * entries), groups, lf_groups


@pytest.mark.parametrize("size", [(64, 48), (256, 256), (8, 8), (320, 240), (1920, 1080), (300, 200), (2560, 2048), (5000, 3000), (70000, 9)])
def test_hand_assembled_headers_and_toc_parse(size):
    """Size header (small / all four U32 ranges / aspect-ratio codes), default metadata, default frame header and the table of contents
    written bit by bit from the format's field tables: the product's parser must read back the same geometry."""
    w, h = size
    data, groups, lf_groups = _hand_built_codestream(w, h)
    info = api.peek(data)
    assert (info.width, info.height) == (w, h)
    assert (info.num_groups, info.num_lf_groups) == (groups, lf_groups)
    assert info.num_channels == 3 and not info.has_alpha and info.bytes_per_sample == 1
    assert (info.xsize_blocks, info.ysize_blocks) == (-(-w // 8), -(-h // 8))
    assert info.codestream_bytes == len(data)
    # the same bytes inside the container
    import struct
    box = lambda t, p: struct.pack(">I4s", 8 + len(p), t) + p
    boxed = box(b"JXL ", b"\r\n\x87\n") + box(b"ftyp", b"jxl \0\0\0\0jxl ") + box(b"jxlc", data)
    assert (api.peek(boxed).width, api.peek(boxed).height) == (w, h)
    # one bit fewer in the table of contents: the sections no longer fit the file
    with pytest.raises(api.FormatError):
        api.peek(data[:-1])
