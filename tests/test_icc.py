"""ICC profiles in the codestream, CPU side: the oracle's and the product's stream codecs against each other, hand-built command
streams (written from the format description, not by either encoder), and the product's matrix / TRC evaluation against float64
ground truth.  Parity with libjxl is unpinned: the reference holds no ICC vector."""
import ctypes as C
import struct

import numpy as np
import pytest

import icc_util
import oracle_lib as O
from pdn_jpegxl_amd import api
from pdn_jpegxl_amd.synth import synth


def product_unpredict(enc):
    L = api.lib()
    L.jxlhip_icc_unpredict.restype = C.c_size_t
    L.jxlhip_icc_unpredict.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(api.ErrorInfo)]
    err = api.ErrorInfo()
    buf = (C.c_uint8 * (1 << 20))()
    n = L.jxlhip_icc_unpredict(bytes(enc), len(enc), buf, len(buf), C.byref(err))
    return bytes(buf[:n]) if n else None


def oracle_unpredict(enc):
    L = O.lib()
    L.jxo_icc_from_stream.restype = C.c_size_t
    L.jxo_icc_from_stream.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    buf = (C.c_uint8 * (1 << 20))()
    n = L.jxo_icc_from_stream(bytes(enc), len(enc), buf, len(buf))
    return bytes(buf[:n]) if n else None


def oracle_predict(icc):
    L = O.lib()
    L.jxo_icc_to_stream.restype = C.c_size_t
    L.jxo_icc_to_stream.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    buf = (C.c_uint8 * (1 << 20))()
    n = L.jxo_icc_to_stream(icc, len(icc), buf, len(buf))
    return bytes(buf[:n])


def parse_icc(data):
    L = api.lib()
    L.jxlhip_parse_icc.restype = C.c_size_t
    L.jxlhip_parse_icc.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(api.ErrorInfo)]
    st, err = C.c_int32(), api.ErrorInfo()
    buf = (C.c_uint8 * (1 << 20))()
    n = L.jxlhip_parse_icc(data, len(data), buf, len(buf), C.byref(st), C.byref(err))
    assert st.value == 0, err.errorMessage
    return bytes(buf[:n])


PROFILES = [icc_util.matrix_profile("p3", "srgb-para"), icc_util.matrix_profile("adobe", "gamma2.2"), icc_util.matrix_profile("srgb", "table1.8"),
            icc_util.gray_profile(), icc_util.cmyk_profile(), icc_util.lut_rgb_profile()]


@pytest.mark.parametrize("k", range(len(PROFILES)))
def test_streams_of_both_implementations_agree(k):
    icc = PROFILES[k]
    enc = oracle_predict(icc)                      # the oracle's form: tag-table, XYZ and type-start commands
    assert len(enc) < len(icc) + 16
    assert oracle_unpredict(enc) == icc
    assert product_unpredict(enc) == icc           # the product's reader on the oracle's commands


@pytest.mark.parametrize("k", range(len(PROFILES)))
def test_oracle_file_with_profile_is_parsed_by_the_product(k):
    icc = PROFILES[k]
    img = synth(70, 50, 41)[..., :3]
    data = O.encode(np.ascontiguousarray(img), lossless=True, icc=icc)
    dec = O.decode(data)
    assert dec.icc == icc and (dec.pixels == img).all()      # original-profile stream: samples untouched
    assert parse_icc(data) == icc                              # entropy-coded stream + predictor, product side


def test_product_header_writer_roundtrip():
    L = api.selftest_lib()
    L.jxlhip_selftest_headers_icc.restype = C.c_size_t
    L.jxlhip_selftest_headers_icc.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    for icc in PROFILES:
        buf = (C.c_uint8 * (1 << 16))()
        n = L.jxlhip_selftest_headers_icc(300, 200, 1, 1, icc, len(icc), buf, len(buf))
        assert n > 0
        assert parse_icc(bytes(buf[:n])) == icc


def varint(v):
    out = bytearray()
    while v >= 128:
        out.append((v & 127) | 128)
        v >>= 7
    out.append(v)
    return bytes(out)


def header_residuals(icc):
    """Residuals of the first 128 bytes against the standard header prediction (format description, written out in Python)."""
    h = bytearray(128)
    h[0:4] = struct.pack(">I", len(icc))
    h[8] = 4
    h[12:24] = b"mntrRGB XYZ "
    h[36:40] = b"acsp"
    h[70], h[71], h[73], h[78], h[79] = 246, 214, 1, 211, 45
    out = bytearray()
    for i in range(128):
        if i == 8:
            h[80:84] = icc[4:8]
        if i == 41:
            if icc[40:41] == b"A":
                h[41:44] = b"PPL"
            if icc[40:41] == b"M":
                h[41:44] = b"SFT"
        if i == 42:
            if icc[40:42] == b"SG":
                h[42:44] = b"I "
            if icc[40:42] == b"SU":
                h[42:44] = b"NW"
        out.append((icc[i] - h[i]) & 255)
    return bytes(out)


def test_hand_built_command_stream():
    """A stream nobody's encoder wrote: tag codes 2 (three tone curves sharing one curve) and 3 (three consecutive colorants),
    an order-1 16-bit predictor over a ramp, a 2-byte shuffle, an XYZ command and a type start."""
    n_ramp = 16
    ramp = b"".join(struct.pack(">H", 1000 + 37 * i) for i in range(n_ramp))           # linear: order 1 predicts it exactly from word 2 on
    # profile layout: header | tag count 6 + 6 entries | curv(12) | 3 x XYZ(20) | ramp | 6 shuffled bytes
    table_end = 128 + 4 + 6 * 12
    curv = b"curv" + b"\0" * 4 + struct.pack(">I", 0)
    xyzs = [b"XYZ " + b"\0" * 4 + struct.pack(">iii", 1000 * k, 2000 * k, 3000 * k) for k in (1, 2, 3)]
    tail = bytes([1, 2, 3, 4, 5, 6])
    body = curv + b"".join(xyzs) + ramp + tail
    entries = [(b"rTRC", table_end, 12), (b"gTRC", table_end, 12), (b"bTRC", table_end, 12),
               (b"rXYZ", table_end + 12, 20), (b"gXYZ", table_end + 32, 20), (b"bXYZ", table_end + 52, 20)]
    size = table_end + len(body)
    hdr = struct.pack(">I", size) + b"abcd" + b"\x04\x30\0\0" + b"mntrRGB XYZ " + b"\0" * 12 + b"acspMSFT" + b"\0" * 88
    hdr = hdr[:128]
    icc = hdr + struct.pack(">I", 6) + b"".join(n + struct.pack(">II", o, s) for n, o, s in entries) + body
    assert len(icc) == size
    commands = varint(6 + 1)
    # rTRC (+ gTRC, bTRC): offset AND size given - the implied first offset is 128 + 12 * ntags, four bytes before the end of a
    # real tag table (the count is not part of the prediction; recalled from the published format, unpinned), see the test below
    commands += bytes([2 | 64 | 128]) + varint(table_end) + varint(12)
    commands += bytes([3])                         # rXYZ (+ gXYZ, bXYZ): offset implied (after the curve), size implied (20)
    commands += bytes([0])                         # end of the table
    data = header_residuals(icc)
    commands += bytes([16 + 5])                    # type start "curv" + 4 zero bytes
    commands += bytes([1]) + varint(4)             # insert: the curve's count
    data += struct.pack(">I", 0)
    for k in (1, 2, 3):
        commands += bytes([10])                    # "XYZ " + 4 zero bytes + 12 data bytes
        data += struct.pack(">iii", 1000 * k, 2000 * k, 3000 * k)
    # the ramp: two words inserted, the rest by the order-1 predictor of width 2 (residuals zero), byte planes shuffled
    commands += bytes([1]) + varint(8)
    data += ramp[:8]
    commands += bytes([4, (2 - 1) | (1 << 2)]) + varint(len(ramp) - 8)
    data += bytes(len(ramp) - 8)
    # the tail through a 2-byte shuffle: planes (1, 3, 5), (2, 4, 6)
    commands += bytes([2]) + varint(6)
    data += bytes([1, 3, 5, 2, 4, 6])
    enc = varint(size) + varint(len(commands)) + commands + data
    assert product_unpredict(enc) == icc
    assert oracle_unpredict(enc) == icc
    # truncated and corrupted streams are refused, not crashed on
    assert product_unpredict(enc[:-3]) is None and oracle_unpredict(enc[:-3]) is None
    bad = bytearray(enc)
    bad[len(varint(size)) + 1 + 5] = 63            # an unknown tag code (in place of the rXYZ command)
    assert enc[len(varint(size)) + 1 + 5] == 3
    assert product_unpredict(bytes(bad)) is None and oracle_unpredict(bytes(bad)) is None


@pytest.mark.parametrize("prim,curve", [("p3", "srgb-para"), ("adobe", "gamma2.2"), ("srgb", "table1.8"), ("adobe", "gamma1.8")])
def test_matrix_trc_model_against_float64(prim, curve):
    icc = icc_util.matrix_profile(prim, curve)
    L = api.lib()
    L.jxlhip_icc_model.restype = C.c_int32
    L.jxlhip_icc_model.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    model = np.zeros(18, np.float64)
    to_lin = np.zeros(3 * 256, np.float32)
    from_lin = np.zeros(3 * 4096, np.float32)
    assert L.jxlhip_icc_model(icc, len(icc), model.ctypes.data, to_lin.ctypes.data, from_lin.ctypes.data) == 1
    want = np.linalg.inv(icc_util.rgb_to_xyz(icc_util.PRIMARIES[prim])) @ icc_util.rgb_to_xyz(icc_util.PRIMARIES["srgb"])
    assert np.abs(model[:9].reshape(3, 3) - want).max() < 2e-4          # s15Fixed16 colorants, Bradford there and back
    assert np.abs(model[:9].reshape(3, 3) @ model[9:].reshape(3, 3) - np.eye(3)).max() < 1e-9
    x = np.arange(256) / 255
    tol = 2e-3 if curve.startswith("table") else 2e-5                     # u16 table entries / u8Fixed8 gamma
    g = curve if not curve.startswith("gamma") else "gamma%.6f" % (round(float(curve[5:]) * 256) / 256)
    assert np.abs(to_lin[:256] - icc_util.decode_curve(g, x)).max() < tol
    # the inverse table (indexed by the square root of the linear value) undoes the forward curve to a fraction of an 8-bit step
    t = np.sqrt(np.clip(to_lin[:256].astype(np.float64), 0, 1)) * 4095
    i = np.minimum(t.astype(int), 4094)
    back = from_lin[i] + (from_lin[i + 1] - from_lin[i]) * (t - i)
    assert np.abs(back - x).max() * 255 < 0.1


def test_profiles_that_need_a_real_cms_are_recognised():
    L = api.lib()
    L.jxlhip_icc_model.restype = C.c_int32
    L.jxlhip_icc_model.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    model = np.zeros(18, np.float64)
    for icc in (icc_util.cmyk_profile(), icc_util.lut_rgb_profile(), b"not a profile" * 20):
        assert L.jxlhip_icc_model(icc, len(icc), model.ctypes.data, None, None) == 0
    g = icc_util.gray_profile()
    assert L.jxlhip_icc_model(g, len(g), model.ctypes.data, None, None) == 2


def test_implied_first_tag_offset_and_wide_fields():
    """The first tag's implied offset is 128 + 12 * ntags (hand-assembled, not written by either encoder); offsets or sizes that
    do not fit the profile's 32-bit fields are refused, not truncated."""
    def stream(first):
        size = 128 + 4 + 12 + 8
        hdr = (struct.pack(">I", size) + b"abcd" + b"\x04\x30\0\0" + b"mntrRGB XYZ " + b"\0" * 12 + b"acspMSFT" + b"\0" * 88)[:128]
        commands = varint(1 + 1) + first + bytes([0]) + bytes([1]) + varint(8)
        data = header_residuals(hdr + b"\0" * 24) + b"desc" + bytes(range(8))
        return hdr, varint(size) + varint(len(commands)) + commands + data
    hdr, enc = stream(bytes([1 | 128]) + varint(8))          # unknown tag name from the data stream, implied offset, size 8
    want = hdr + struct.pack(">I", 1) + b"desc" + struct.pack(">II", 128 + 12, 8) + bytes(range(8))
    assert product_unpredict(enc) == want and oracle_unpredict(enc) == want
    for off, sz in ((1 << 32, 8), (140, 1 << 32), (0xFFFFFFF0, 0x10)):
        _, bad = stream(bytes([1 | 64 | 128]) + varint(off) + varint(sz))
        assert product_unpredict(bad) is None and oracle_unpredict(bad) is None, (off, sz)


@pytest.mark.parametrize("stride", [1 << 62, (1 << 63) - 1, (1 << 62) + 1, 1 << 32, 33, 34])
def test_predictor_stride_is_checked_without_overflow(stride):
    """A 63-bit stride used to pass `stride * 4 >= size` by wrapping and index far outside the profile (advisor, round 2)."""
    commands = varint(0) + bytes([1]) + varint(4) + bytes([4, 0x10]) + varint(stride) + varint(4)
    data = bytes(128) + bytes([1, 2, 3, 4]) + bytes(4)
    enc = varint(136) + varint(len(commands)) + commands + data
    got_p, got_o = product_unpredict(enc), oracle_unpredict(enc)
    if stride * 4 < 132:        # the only legal one of the list: 32 * 4 < 132 fails, 33 * 4 = 132 fails too -> none legal but keep the rule visible
        assert got_p is not None and got_p == got_o
    else:
        assert got_p is None and got_o is None


def test_icc_unpredict_under_asan(tmp_path):
    """Byte-mutation fuzz of the product's ICC stream reader under CPU AddressSanitizer (tests/native/icc_fuzz_main.cc): the code is
    reachable from LoadImage with an untrusted file, and a try/catch does not catch a wild read."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    exe = str(tmp_path / "icc_fuzz")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           os.path.join(here, "native", "icc_fuzz_main.cc"), os.path.join(root, "pdn_jpegxl_amd", "csrc", "icc.cc"), "-o", exe])
    seeds = []
    for k, icc in enumerate(PROFILES[:4]):
        path = str(tmp_path / ("seed%d.bin" % k))
        with open(path, "wb") as f:
            f.write(oracle_predict(icc))
        seeds.append(path)
    r = subprocess.run([exe] + seeds, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
    assert b"no crash" in r.stdout
