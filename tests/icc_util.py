"""ICC profiles for the tests, built from first principles (ICC.1:2010 layout) with numpy: matrix / TRC RGB profiles with parametric,
gamma or table tone curves, a gray profile, and two kinds the minimal colour management must NOT accept (CMYK, LUT-based)."""
import struct

import numpy as np

D65 = (0.3127, 0.3290)
PRIMARIES = {"srgb": ((0.64, 0.33), (0.30, 0.60), (0.15, 0.06)), "p3": ((0.680, 0.320), (0.265, 0.690), (0.150, 0.060)),
             "adobe": ((0.64, 0.33), (0.21, 0.71), (0.15, 0.06))}
BRADFORD_D65_TO_D50 = np.array([[1.0478112, 0.0228866, -0.0501270], [0.0295424, 0.9904844, -0.0170491], [-0.0092345, 0.0150436, 0.7521316]])


def rgb_to_xyz(prim, white=D65):
    xyz = lambda x, y: np.array([x / y, 1.0, (1 - x - y) / y])
    m = np.stack([xyz(*p) for p in prim], axis=1)
    s = np.linalg.solve(m, xyz(*white))
    return m * s


def s15(v):
    return struct.pack(">i", int(round(v * 65536)))


def xyz_tag(v):
    return b"XYZ " + b"\0" * 4 + b"".join(s15(x) for x in v)


def curve_tag(kind):
    if kind == "srgb-para":   # parametric type 3: (a x + b)^g for x >= d, c x below
        return b"para" + b"\0" * 4 + struct.pack(">HH", 3, 0) + b"".join(s15(v) for v in (2.4, 1 / 1.055, 0.055 / 1.055, 1 / 12.92, 0.04045))
    if kind.startswith("gamma"):
        g = float(kind[5:])
        return b"curv" + b"\0" * 4 + struct.pack(">I", 1) + struct.pack(">H", int(round(g * 256))) + b"\0\0"
    if kind.startswith("table"):
        g, n = float(kind[5:]), 1024
        t = np.round((np.arange(n) / (n - 1)) ** g * 65535).astype(">u2")
        return b"curv" + b"\0" * 4 + struct.pack(">I", n) + t.tobytes()
    raise ValueError(kind)


def decode_curve(kind, x):
    """encoded -> linear, the same curves in float64 (ground truth for the tests)."""
    x = np.asarray(x, np.float64)
    if kind == "srgb-para":
        return np.where(x >= 0.04045, ((x + 0.055) / 1.055) ** 2.4, x / 12.92)
    return x ** float(kind[5:])


def assemble(space, tags):
    """tags: list of (name, body) or (name, alias-of-name)."""
    ntags = len(tags)
    off = 128 + 4 + 12 * ntags
    data, where, table = b"", {}, struct.pack(">I", ntags)
    for name, body in tags:
        if isinstance(body, str):
            o, n = where[body]
        else:
            o, n = off + len(data), len(body)
            data += body + b"\0" * ((-len(body)) % 4)
        where[name] = (o, n)
        table += name.encode() + struct.pack(">II", o, n)
    size = 128 + len(table) + len(data)
    hdr = struct.pack(">I", size) + b"jxlt" + struct.pack(">I", 0x04300000) + b"mntr" + space + b"XYZ " + bytes([7, 230, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0]) \
        + b"acsp" + b"APPL" + b"\0" * 12 + b"\0" * 8 + b"\0" * 4 + s15(0.9642) + s15(1.0) + s15(0.8249) + b"jxlt" + b"\0" * 44
    assert len(hdr) == 128
    return hdr + table + data


def matrix_profile(primaries="p3", curve="srgb-para"):
    m = BRADFORD_D65_TO_D50 @ rgb_to_xyz(PRIMARIES[primaries])
    desc = b"mluc" + b"\0" * 4 + struct.pack(">II", 1, 12) + b"enUS" + struct.pack(">II", 14, 28) + "test".encode("utf-16-be") + b"\0" * 6
    return assemble(b"RGB ", [("desc", desc), ("cprt", b"text" + b"\0" * 4 + b"none\0"), ("wtpt", xyz_tag((0.9642, 1.0, 0.8249))),
                              ("rXYZ", xyz_tag(m[:, 0])), ("gXYZ", xyz_tag(m[:, 1])), ("bXYZ", xyz_tag(m[:, 2])),
                              ("rTRC", curve_tag(curve)), ("gTRC", "rTRC"), ("bTRC", "rTRC")])


def gray_profile(curve="gamma2.2"):
    return assemble(b"GRAY", [("wtpt", xyz_tag((0.9642, 1.0, 0.8249))), ("kTRC", curve_tag(curve))])


def cmyk_profile(seed=3):
    lut = bytes(np.random.default_rng(seed).integers(0, 256, 600, dtype=np.uint8))
    return assemble(b"CMYK", [("desc", b"text" + b"\0" * 4 + b"cmyk\0"), ("A2B0", b"mft1" + b"\0" * 4 + lut)])


def lut_rgb_profile():
    return assemble(b"RGB ", [("wtpt", xyz_tag((0.9642, 1.0, 0.8249))), ("A2B0", b"mft2" + b"\0" * 4 + b"\x03\x03\x02\0" + b"\0" * 200)])


def srgb_to_profile(rgb8, primaries, curve):
    """sRGB u8 samples -> the same colours as u8 samples of the given matrix / TRC space, in float64 (ground truth)."""
    v = rgb8.astype(np.float64) / 255
    lin = np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4)
    m = np.linalg.inv(rgb_to_xyz(PRIMARIES[primaries])) @ rgb_to_xyz(PRIMARIES["srgb"])
    plin = np.clip(lin @ m.T, 0, 1)
    if curve == "srgb-para":
        enc = np.where(plin <= 0.0031308, plin * 12.92, 1.055 * plin ** (1 / 2.4) - 0.055)
    else:
        enc = plin ** (1 / float(curve[5:]))
    return np.clip(np.rint(enc * 255), 0, 255).astype(np.uint8)
