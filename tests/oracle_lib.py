"""ctypes binding of the CPU oracle (oracle/_build/libjxo.so) — test infrastructure only."""
import ctypes as C
import os
import subprocess
import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "_build", "libjxo.so")


class EncodeParams(C.Structure):
    _fields_ = [("distance", C.c_float), ("lossless", C.c_int32), ("effort", C.c_int32), ("strategy_mode", C.c_int32),
                ("fixed_strategy", C.c_int32), ("seed", C.c_uint32), ("epf_iters", C.c_int32), ("gaborish", C.c_int32),
                ("container", C.c_int32), ("adaptive_lf_smoothing", C.c_int32), ("lossless_predictor", C.c_int32),
                ("lossless_squeeze", C.c_int32), ("lossless_tree", C.c_int32), ("num_threads", C.c_int32), ("bits", C.c_int32), ("orientation", C.c_int32), ("float_samples", C.c_int32), ("colour", C.c_int32)]


def build():
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(_ROOT, "oracle")])


def build_native():
    """-O3 -march=native build for the machine this runs on (bench.py's cpu_baseline leg; the default build is portable because it
    travels from the build container to the GPU box).  Returns the path of the library."""
    # keyed by the CPU's feature flags: a build made elsewhere (e.g. in the container the repo was snapshotted from) is never reused
    import hashlib
    try:
        flags = [l for l in open("/proc/cpuinfo") if l.startswith("flags")][0]
    except Exception:
        flags = "unknown"
    out = os.path.join("_build", "native-" + hashlib.sha1(flags.encode()).hexdigest()[:10])
    subprocess.check_call(["make", "-s", "-j16", "-C", os.path.join(_ROOT, "oracle"), "BUILD=" + out,
                           "CXXFLAGS=-O3 -march=native -std=c++17 -fPIC -pthread"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(_ROOT, "oracle", out, "libjxo.so")


def use(path):
    """Binds a different build of the oracle library (before or after the first call)."""
    global _lib, _SO
    _SO = path
    _lib = None


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.jxo_last_error.restype = C.c_char_p
        L.jxo_decode.restype = C.c_void_p
        L.jxo_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int]
        L.jxo_image_free.argtypes = [C.c_void_p]
        L.jxo_image_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int32)] * 5
        L.jxo_image_pixels.restype = C.POINTER(C.c_uint8)
        L.jxo_image_pixels.argtypes = [C.c_void_p]
        L.jxo_image_epf_iters.argtypes = [C.c_void_p]
        L.jxo_image_exif.restype = C.c_size_t
        L.jxo_image_exif.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint8))]
        L.jxo_image_xml.restype = C.c_size_t
        L.jxo_image_xml.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint8))]
        L.jxo_image_plane.restype = C.c_size_t
        L.jxo_image_plane.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]
        L.jxo_encode.restype = C.c_void_p
        L.jxo_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(EncodeParams), C.c_char_p, C.c_size_t,
                                 C.c_char_p, C.c_size_t]
        L.jxo_bytes_data.restype = C.POINTER(C.c_uint8)
        L.jxo_bytes_data.argtypes = [C.c_void_p]
        L.jxo_bytes_size.restype = C.c_size_t
        L.jxo_bytes_size.argtypes = [C.c_void_p]
        L.jxo_bytes_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def encode(px, distance=1.0, lossless=False, strategy_mode=0, fixed_strategy=0, seed=1, epf_iters=-1, gaborish=True,
           container=True, adaptive_lf_smoothing=True, lossless_predictor=6, lossless_squeeze=False, num_threads=8, exif=None,
           xmp=None, lossless_tree=0, bits=8, orientation=1, float_samples=0, colour=0, icc=None, cmyk=False, animation_frames=1, custom_quant_tables=False, prefix_codes=False, lz77=False, num_passes=1, custom_orders=False, lf_contexts=False, palette=False, mislabel_afv=False,
           premultiplied_alpha=False):
    """px: uint8 array (h, w, nch) with nch in 1..4 (Gray, GrayA, RGB, RGBA); with bits > 8 (up to 16) a uint16 array whose
    samples use the low `bits` bits; with float_samples = 16 / 32 a float16 / float32 array (nominal range [0, 1]).  Returns bytes."""
    L = lib()
    if float_samples:
        px = np.ascontiguousarray(px, dtype=np.float16 if float_samples == 16 else np.float32)
    else:
        px = np.ascontiguousarray(px, dtype=np.uint8 if bits <= 8 else np.uint16)
    if px.ndim == 2:
        px = px[:, :, None]
    h, w, nch = px.shape
    p = EncodeParams(distance, int(lossless), 7, strategy_mode, fixed_strategy, seed, epf_iters, int(gaborish), int(container),
                     int(adaptive_lf_smoothing), lossless_predictor, int(lossless_squeeze), lossless_tree, num_threads, bits, orientation, float_samples, colour)
    if custom_quant_tables or prefix_codes or lz77 or num_passes > 1 or custom_orders or lf_contexts or palette or mislabel_afv or premultiplied_alpha:
        L.jxo_set_next_flags.argtypes = [C.c_uint32]
        L.jxo_set_next_flags((1 if custom_quant_tables else 0) | (2 if prefix_codes else 0) | (4 if lz77 else 0) | {1: 0, 2: 8, 3: 16}[num_passes] | (32 if custom_orders else 0) | (64 if lf_contexts else 0) | (128 if palette else 0) |
                             (256 if mislabel_afv else 0) | (512 if premultiplied_alpha else 0))
    if animation_frames > 1:
        L.jxo_set_next_animation.argtypes = [C.c_int]
        L.jxo_set_next_animation(animation_frames)
    if icc or cmyk:
        L.jxo_set_next_icc.argtypes = [C.c_char_p, C.c_size_t, C.c_int]
        L.jxo_set_next_icc(icc, len(icc) if icc else 0, int(cmyk))
    hnd = L.jxo_encode(px.ctypes.data, w, h, nch, C.byref(p), exif, len(exif) if exif else 0, xmp, len(xmp) if xmp else 0)
    if not hnd:
        raise OracleError(L.jxo_last_error().decode())
    try:
        n = L.jxo_bytes_size(hnd)
        return bytes(C.string_at(L.jxo_bytes_data(hnd), n))
    finally:
        L.jxo_bytes_free(hnd)


_PLANE_DTYPES = {"lf_quant": np.int32, "lf": np.float32, "qcoef": np.int32, "xyb_idct": np.float32, "xyb_filtered": np.float32,
                 "strategy": np.uint8, "raw_quant": np.int32, "sharpness": np.uint8, "ytox": np.int8, "ytob": np.int8,
                 "alpha": np.int32}


class Decoded:
    def __init__(self, pixels, w8, h8, planes, exif, xml, epf_iters, icc=b"", cmyk=False):
        self.icc, self.cmyk = icc, cmyk
        self.pixels = pixels
        self.w8, self.h8 = w8, h8
        self.planes = planes
        self.exif, self.xml = exif, xml
        self.epf_iters = epf_iters


def decode(data, num_threads=8, want_dump=False):
    L = lib()
    hnd = L.jxo_decode(data, len(data), num_threads, int(want_dump))
    if not hnd:
        raise OracleError(L.jxo_last_error().decode())
    try:
        w, h, nch, w8, h8 = (C.c_int32() for _ in range(5))
        L.jxo_image_info(hnd, w, h, nch, w8, h8)
        n = w.value * h.value * nch.value
        L.jxo_image_bits_out.restype = C.c_int32
        L.jxo_image_bits_out.argtypes = [C.c_void_p]
        L.jxo_image_out_float.restype = C.c_int32
        L.jxo_image_out_float.argtypes = [C.c_void_p]
        bo, fl = L.jxo_image_bits_out(hnd), L.jxo_image_out_float(hnd)
        dt = {(8, 0): np.uint8, (16, 0): np.uint16, (16, 1): np.float16, (32, 1): np.float32}[(bo, fl)]
        px = np.ctypeslib.as_array(L.jxo_image_pixels(hnd), shape=(n * (bo // 8),)).view(dt).reshape(h.value, w.value, nch.value).copy()
        planes = {}
        if want_dump:
            for name, dt in _PLANE_DTYPES.items():
                chans = 3 if name in ("lf_quant", "lf", "qcoef", "xyb_idct", "xyb_filtered") else 1
                out = []
                for c in range(chans):
                    ptr, es = C.c_void_p(), C.c_int32()
                    cnt = L.jxo_image_plane(hnd, name.encode(), c, C.byref(ptr), C.byref(es))
                    if cnt == 0:
                        out.append(np.zeros(0, dt))
                    else:
                        buf = (C.c_uint8 * (cnt * es.value)).from_address(ptr.value)
                        out.append(np.frombuffer(buf, dtype=dt).copy())
                planes[name] = out if chans == 3 else out[0]
        p = C.POINTER(C.c_uint8)()
        n = L.jxo_image_exif(hnd, C.byref(p))
        exif = bytes(C.string_at(p, n)) if n else b""
        n = L.jxo_image_xml(hnd, C.byref(p))
        xml = bytes(C.string_at(p, n)) if n else b""
        L.jxo_image_icc.restype = C.c_size_t
        L.jxo_image_icc.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint8))]
        L.jxo_image_cmyk.argtypes = [C.c_void_p]
        n = L.jxo_image_icc(hnd, C.byref(p))
        icc = bytes(C.string_at(p, n)) if n else b""
        return Decoded(px, w8.value, h8.value, planes, exif, xml, L.jxo_image_epf_iters(hnd), icc, bool(L.jxo_image_cmyk(hnd)))
    finally:
        L.jxo_image_free(hnd)
