"""Host-side mirror of the reference's managed interop (src/Interop/JpegXLNative.cs) over ctypes.

`load_image` / `save_image` / `get_libjxl_version` call the three C-ABI exports exactly like the C# host
does (callback struct of six function pointers, ErrorInfo, status -> exception mapping).  `Decoder` wraps the
device-resident batch entry points used by bench.py and the GPU parity tests.

The native library must exist: there is no CPU fallback.  Build it with `python -m pdn_jpegxl_amd.build`.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

# ---------------------------------------------------------------- enums (include/jxlfiletypeio.h)
DECODER_STATUS = ["Ok", "NullParameter", "InvalidParameter", "OutOfMemory", "HasAnimation", "HasMultipleFrames",
                  "ImageDimensionExceedsInt32", "UnsupportedChannelFormat", "CreateLayerError", "CreateMetadataError",
                  "DecodeError", "MetadataError", "InvalidFileSignature"]
ENCODER_STATUS = ["Ok", "NullParameter", "OutOfMemory", "UserCanceled", "EncodeError", "WriteError"]
IMAGE_FORMAT = ["Gray", "Rgb", "Cmyk"]
KNOWN_PROFILE = ["Srgb", "LinearSrgb", "LinearGray", "GraySrgbTRC", "DisplayP3", "Rec709", "Rec2020Linear", "Rec2020PQ"]
S_OK, E_ABORT, E_OUTOFMEMORY, E_FAIL = 0, -2147467260, -2147024882, -2147467259


class ErrorInfo(C.Structure):
    _fields_ = [("errorMessage", C.c_char * 256)]


class BitmapData(C.Structure):
    _fields_ = [("scan0", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("stride", C.c_uint32)]


class EncoderOptions(C.Structure):
    _fields_ = [("distance", C.c_float), ("effort", C.c_int32), ("lossless", C.c_bool)]


class EncoderImageMetadata(C.Structure):
    _fields_ = [("exif", C.c_void_p), ("exifSize", C.c_size_t), ("iccProfile", C.c_void_p), ("iccProfileSize", C.c_size_t),
                ("xmp", C.c_void_p), ("xmpSize", C.c_size_t)]


SetBasicInfoFn = C.CFUNCTYPE(None, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_bool)
SetMetadataFn = C.CFUNCTYPE(C.c_bool, C.POINTER(C.c_uint8), C.c_size_t)
SetKnownProfileFn = C.CFUNCTYPE(C.c_bool, C.c_int32)
SetLayerDataFn = C.CFUNCTYPE(C.c_bool, C.POINTER(C.c_uint8), C.c_void_p, C.c_size_t)
WriteFn = C.CFUNCTYPE(C.c_int32, C.POINTER(C.c_uint8), C.c_size_t)
SeekFn = C.CFUNCTYPE(C.c_int32, C.c_uint64)
ProgressFn = C.CFUNCTYPE(C.c_bool, C.c_int32)


class DecoderCallbacks(C.Structure):
    _fields_ = [("setBasicInfo", SetBasicInfoFn), ("setIccProfile", SetMetadataFn), ("setKnownColorProfile", SetKnownProfileFn),
                ("setExif", SetMetadataFn), ("setXmp", SetMetadataFn), ("setLayerData", SetLayerDataFn)]


class IOCallbacks(C.Structure):
    _fields_ = [("Write", WriteFn), ("Seek", SeekFn)]


class ImageInfo(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("num_channels", C.c_int32), ("has_alpha", C.c_int32),
                ("xsize_blocks", C.c_int32), ("ysize_blocks", C.c_int32), ("num_groups", C.c_int32), ("num_lf_groups", C.c_int32),
                ("epf_iters", C.c_int32), ("gaborish", C.c_int32), ("codestream_bytes", C.c_uint64), ("bytes_per_sample", C.c_int32),
                ("reserved", C.c_int32)]


EXPORTS = ["GetLibJxlVersion", "LoadImage", "SaveImage", "jxlhip_parse_icc", "jxlhip_decoder_create", "jxlhip_decoder_destroy", "jxlhip_peek",
           "jxlhip_decode_batch", "jxlhip_finish", "jxlhip_read_plane", "jxlhip_set_option", "jxlhip_stage_times", "jxlhip_stage_totals",
           "jxlhip_last_load_stage_times", "jxlhip_last_save_stage_times"]

_lib = None


_selftest_lib = None


def selftest_lib():
    """The TEST library (the product's objects + csrc/selftest.cc): writer self tests for the CPU suite.  The production library
    does not export these hooks."""
    global _selftest_lib
    if _selftest_lib is None:
        lib()   # builds if needed, checks the manifest, loads torch's HIP runtime first
        _selftest_lib = C.CDLL(_build.test_lib_path())
    return _selftest_lib


class NativeLibraryMissing(RuntimeError):
    pass


def lib(build_if_missing=True):
    """Loads the native library; raises NativeLibraryMissing if it cannot be built/loaded (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # When PyTorch is present it must load first: both link libamdhip64.so.7 and have to share ONE HIP runtime
        # (torch bundles its own); loading the system runtime first leaves torch without a visible GPU.
        import torch  # noqa: F401
    except ImportError:
        pass
    path = _build.lib_path()
    if not os.path.exists(path):
        if not build_if_missing:
            raise NativeLibraryMissing(path)
        _build.build()
    # a library built from other sources than the ones in the tree is refused, not used (and not rebuilt here: this process may already
    # hold the GPU, e.g. under a profiler)
    try:
        built_from = open(_build.manifest_path()).read()
    except OSError:
        built_from = None
    if built_from != _build.manifest_text():
        raise NativeLibraryMissing("%s is stale (built from different sources): run python -m pdn_jpegxl_amd.build" % path)
    try:
        L = C.CDLL(path)
    except OSError as e:
        raise NativeLibraryMissing("%s: %s" % (path, e))
    L.GetLibJxlVersion.restype = C.c_uint32
    L.LoadImage.restype = C.c_int32
    L.LoadImage.argtypes = [C.POINTER(DecoderCallbacks), C.c_char_p, C.c_size_t, C.POINTER(ErrorInfo)]
    L.SaveImage.restype = C.c_int32
    L.SaveImage.argtypes = [C.POINTER(BitmapData), C.POINTER(EncoderOptions), C.POINTER(EncoderImageMetadata), C.POINTER(IOCallbacks),
                            C.POINTER(ErrorInfo), ProgressFn]
    L.jxlhip_decoder_create.restype = C.c_void_p
    L.jxlhip_decoder_create.argtypes = [C.c_int32, C.POINTER(ErrorInfo)]
    L.jxlhip_decoder_destroy.argtypes = [C.c_void_p]
    L.jxlhip_peek.restype = C.c_int32
    L.jxlhip_peek.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(ImageInfo), C.POINTER(ErrorInfo)]
    L.jxlhip_decode_batch.restype = C.c_int32
    L.jxlhip_decode_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_void_p), C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(ErrorInfo)]
    L.jxlhip_finish.restype = C.c_int32
    L.jxlhip_finish.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(ErrorInfo)]
    L.jxlhip_read_plane.restype = C.c_size_t
    L.jxlhip_read_plane.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.c_void_p, C.c_size_t]
    L.jxlhip_set_option.restype = C.c_int32
    L.jxlhip_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    L.jxlhip_stage_times.restype = C.c_int32
    L.jxlhip_stage_times.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int32]
    L.jxlhip_stage_totals.restype = C.c_int32
    L.jxlhip_stage_totals.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_int32), C.c_int32]
    L.jxlhip_parse_check.restype = C.c_int32
    L.jxlhip_parse_check.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(ErrorInfo)]
    L.jxlhip_static_table.restype = C.c_size_t
    L.jxlhip_static_table.argtypes = [C.c_char_p, C.c_int32, C.c_void_p, C.c_size_t]
    _lib = L
    return L


# ---------------------------------------------------------------- exceptions (JpegXLNative.cs:128-239)
class JxlError(RuntimeError):
    def __init__(self, status, message=""):
        self.status = status
        super().__init__("%s%s" % (status, (": " + message) if message else ""))


class FormatError(JxlError):
    pass


def get_libjxl_version():
    """(major, minor, patch) as JpegXLNative.GetLibJxlVersion unpacks it (JpegXLNative.cs:40-42)."""
    v = lib().GetLibJxlVersion()
    return (v >> 24) & 0xFF, (v >> 16) & 0xFF, (v >> 8) & 0xFF


class DecodedImage:
    """What the host's DecoderImage sink collects (src/Interop/DecoderImage.cs)."""

    def __init__(self):
        self.width = self.height = 0
        self.format = None
        self.channel_representation = 0
        self.has_transparency = False
        self.known_profile = None
        self.icc = None
        self.exif = None
        self.xmp = None
        self.pixels = None
        self.layer_name = None
        self.trace = []


def load_image(data, fail_at=None):
    """JpegXLNative.LoadImage: data = whole file (bytes).  Returns DecodedImage.  `fail_at` makes the named callback
    return false (host-side failure injection, e.g. 'setLayerData')."""
    L = lib()
    img = DecodedImage()

    def basic(w, h, fmt, rep, alpha):
        img.trace.append("setBasicInfo")
        img.width, img.height, img.format, img.channel_representation, img.has_transparency = w, h, IMAGE_FORMAT[fmt], rep, bool(alpha)

    def icc(p, n):
        img.trace.append("setIccProfile")
        img.icc = bytes(C.string_at(p, n))
        return fail_at != "setIccProfile"

    def known(p):
        img.trace.append("setKnownColorProfile")
        img.known_profile = KNOWN_PROFILE[p]
        return fail_at != "setKnownColorProfile"

    def exif(p, n):
        img.trace.append("setExif")
        img.exif = bytes(C.string_at(p, n))
        return fail_at != "setExif"

    def xmp(p, n):
        img.trace.append("setXmp")
        if img.xmp is None:  # the host keeps the first (DecoderImage.cs:248)
            img.xmp = bytes(C.string_at(p, n))
        return fail_at != "setXmp"

    def layer(p, name, nlen):
        img.trace.append("setLayerData")
        if fail_at == "setLayerData":
            return False
        nch = {"Gray": 1, "Rgb": 3, "Cmyk": 4}[img.format] + (1 if img.has_transparency else 0)
        n = img.width * img.height * nch
        # ImageChannelRepresentation (Common.h:33-39): Uint8, Uint16, Float16, Float32
        dt = (np.uint8, np.uint16, np.float16, np.float32)[img.channel_representation]
        nbytes = n * np.dtype(dt).itemsize
        img.pixels = np.ctypeslib.as_array(p, shape=(nbytes,)).view(dt).reshape(img.height, img.width, nch).copy()
        img.layer_name = C.string_at(name, nlen - 1).decode("utf-8", "replace") if name and nlen else None
        return True

    cbs = DecoderCallbacks(SetBasicInfoFn(basic), SetMetadataFn(icc), SetKnownProfileFn(known), SetMetadataFn(exif), SetMetadataFn(xmp),
                           SetLayerDataFn(layer))
    err = ErrorInfo()
    st = L.LoadImage(C.byref(cbs), data, len(data), C.byref(err))
    if st != 0:
        name = DECODER_STATUS[st] if 0 <= st < len(DECODER_STATUS) else str(st)
        msg = err.errorMessage.decode("ascii", "replace")
        if name in ("InvalidFileSignature", "DecodeError", "MetadataError", "HasAnimation", "HasMultipleFrames", "UnsupportedChannelFormat",
                    "ImageDimensionExceedsInt32"):
            raise FormatError(name, msg)
        raise JxlError(name, msg)
    return img


def save_image(bgra, distance=1.0, effort=7, lossless=False, exif=None, icc=None, xmp=None, progress=None, write_result=None):
    """JpegXLNative.SaveImage: bgra = uint8 (h, w, 4) BGRA surface (rows may be strided).  Returns the encoded bytes.
    write_result: HRESULT the Write callback returns instead of S_OK (failure injection)."""
    L = lib()
    bgra = np.asarray(bgra, dtype=np.uint8)
    if bgra.strides[2] != 1 or bgra.strides[1] != 4:
        bgra = np.ascontiguousarray(bgra)
    h, w, _ = bgra.shape
    out = bytearray()
    pos = [0]

    def write(p, n):
        if write_result is not None:
            return write_result - (1 << 32) if write_result >= (1 << 31) else write_result
        chunk = C.string_at(p, n)
        end = pos[0] + n
        if end > len(out):
            out.extend(b"\0" * (end - len(out)))
        out[pos[0]:end] = chunk
        pos[0] = end
        return S_OK

    def seek(p):
        pos[0] = p
        return S_OK

    bmp = BitmapData(bgra.ctypes.data, w, h, bgra.strides[0])
    opt = EncoderOptions(distance, effort, lossless)
    keep = [np.frombuffer(b, np.uint8) if b else None for b in (exif, icc, xmp)]
    md = EncoderImageMetadata(*sum(([k.ctypes.data if k is not None else None, len(k) if k is not None else 0] for k in keep), []))
    io = IOCallbacks(WriteFn(write), SeekFn(seek))
    err = ErrorInfo()
    st = L.SaveImage(C.byref(bmp), C.byref(opt), C.byref(md), C.byref(io), C.byref(err), ProgressFn(progress) if progress else ProgressFn())
    if st != 0:
        raise JxlError(ENCODER_STATUS[st] if 0 <= st < len(ENCODER_STATUS) else str(st), err.errorMessage.decode("ascii", "replace"))
    return bytes(out)


def peek(data):
    info, err = ImageInfo(), ErrorInfo()
    st = lib().jxlhip_peek(data, len(data), C.byref(info), C.byref(err))
    if st != 0:
        raise FormatError(DECODER_STATUS[st], err.errorMessage.decode("ascii", "replace"))
    return info


def parse_check(data):
    facts = (C.c_int32 * 8)()
    err = ErrorInfo()
    st = lib().jxlhip_parse_check(data, len(data), facts, C.byref(err))
    return DECODER_STATUS[st], list(facts), err.errorMessage.decode("ascii", "replace")


def parse_metadata(data, which):
    """Host-only: the Exif payload (which = 0) or the k-th `xml ` payload (which = k >= 1) as LoadImage would report it
    (Brotli-compressed `brob` boxes already decompressed); None if absent.  Returns (status, payload, message)."""
    L = lib()
    L.jxlhip_parse_metadata.restype = C.c_size_t
    L.jxlhip_parse_metadata.argtypes = [C.c_char_p, C.c_size_t, C.c_int32, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(ErrorInfo)]
    err = ErrorInfo()
    st = C.c_int32(0)
    n = L.jxlhip_parse_metadata(data, len(data), which, None, 0, C.byref(st), C.byref(err))
    if st.value != 0 or n == 0:
        return DECODER_STATUS[st.value], None, err.errorMessage.decode("ascii", "replace")
    buf = (C.c_uint8 * n)()
    L.jxlhip_parse_metadata(data, len(data), which, buf, n, C.byref(st), C.byref(err))
    return DECODER_STATUS[st.value], bytes(buf), ""


def static_table(name, index, dtype):
    L = lib()
    n = L.jxlhip_static_table(name.encode(), index, None, 0)
    buf = np.empty(n // np.dtype(dtype).itemsize, dtype)
    L.jxlhip_static_table(name.encode(), index, buf.ctypes.data, n)
    return buf


_PLANE_DTYPES = {"lf": np.float32, "lf_quant": np.int32, "cellinfo": np.uint32, "raw_quant": np.uint16, "sharpness": np.uint8,
                 "ytox": np.int8, "ytob": np.int8, "alpha": np.uint8, "inv_sigma": np.float32, "qcoef": np.int32,
                 "xyb_idct": np.float32, "xyb_filtered": np.float32}


def _last_stage_times(fn_name):
    L = lib()
    fn = getattr(L, fn_name)
    fn.restype = C.c_int32
    fn.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int32]
    names = (C.c_char_p * 32)()
    ms = (C.c_float * 32)()
    k = fn(names, ms, 32)
    return {names[i].decode(): ms[i] for i in range(k)}


def last_load_stage_times():
    """Device time per stage of this thread's last load_image() call (HIP events)."""
    return _last_stage_times("jxlhip_last_load_stage_times")


def last_save_stage_times():
    """Device time per kernel group of this thread's last lossy save_image() call (HIP events)."""
    return _last_stage_times("jxlhip_last_save_stage_times")


class Decoder:
    """Device-resident batch decoder (jxlhip_* entry points)."""

    def __init__(self, device=-1):
        self._h = None
        self._L = lib()
        err = ErrorInfo()
        self._h = self._L.jxlhip_decoder_create(device, C.byref(err))
        if not self._h:
            raise JxlError("DecoderCreate", err.errorMessage.decode("ascii", "replace"))

    def close(self):
        if self._h:
            self._L.jxlhip_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_option(self, name, value):
        return self._L.jxlhip_set_option(self._h, name.encode(), int(value))

    def decode_batch(self, files, dev_out_ptrs, dev_data_ptrs=None, stream=None, synchronize=True, raise_on_error=True):
        """files: list of bytes; dev_out_ptrs: device pointers (ints); dev_data_ptrs: device pointers of resident file bytes."""
        n = len(files)
        hd = (C.c_char_p * n)(*files)
        sz = (C.c_size_t * n)(*[len(f) for f in files])
        dd = (C.c_void_p * n)(*(dev_data_ptrs or [None] * n))
        do = (C.c_void_p * n)(*dev_out_ptrs)
        st = (C.c_int32 * n)()
        err = ErrorInfo()
        self._keep = (hd, sz, dd, do)
        r = self._L.jxlhip_decode_batch(self._h, n, hd, sz, dd, do, stream, 1 if synchronize else 0, st, C.byref(err))
        self.last_error = err.errorMessage.decode("ascii", "replace")
        if r != 0 and raise_on_error:
            raise FormatError(DECODER_STATUS[r] if 0 <= r < len(DECODER_STATUS) else str(r), self.last_error)
        return list(st)

    def finish(self):
        """Waits for every asynchronous batch; raises if any of them failed."""
        err = ErrorInfo()
        r = self._L.jxlhip_finish(self._h, None, C.byref(err))
        if r != 0:
            raise FormatError(DECODER_STATUS[r], err.errorMessage.decode("ascii", "replace"))

    def read_plane(self, index, name, channel=0):
        n = self._L.jxlhip_read_plane(self._h, index, name.encode(), channel, None, 0)
        if n == 0:
            raise KeyError(name)
        buf = np.empty(n // np.dtype(_PLANE_DTYPES[name]).itemsize, _PLANE_DTYPES[name])
        self._L.jxlhip_read_plane(self._h, index, name.encode(), channel, buf.ctypes.data, n)
        return buf

    def stage_totals(self, reset=False):
        """({stage: cumulative ms}, batches) over every batch finished since the last reset."""
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        nb = C.c_int32()
        k = self._L.jxlhip_stage_totals(self._h, names, ms, 16, C.byref(nb), 1 if reset else 0)
        return {names[i].decode(): ms[i] for i in range(k)}, nb.value

    def stage_times(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        k = self._L.jxlhip_stage_times(self._h, names, ms, 16)
        return {names[i].decode(): ms[i] for i in range(k)}
