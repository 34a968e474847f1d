"""Test helper: rewrites the Exif / `xml ` boxes of a .jxl container as Brotli-compressed `brob` boxes (ISO/IEC 18181-2).  The Brotli
ENCODER comes from the base image's libbrotlienc.so.1 through ctypes; tests that need it skip when it is missing."""
import ctypes as C
import ctypes.util
import struct


def _encoder():
    for name in ("libbrotlienc.so.1", ctypes.util.find_library("brotlienc")):
        if not name:
            continue
        try:
            return C.CDLL(name)
        except OSError:
            pass
    return None


def brotli_compress(data, quality=9):
    lib = _encoder()
    if lib is None:
        return None
    lib.BrotliEncoderMaxCompressedSize.restype = C.c_size_t
    lib.BrotliEncoderMaxCompressedSize.argtypes = [C.c_size_t]
    lib.BrotliEncoderCompress.restype = C.c_int
    lib.BrotliEncoderCompress.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t), C.c_char_p]
    cap = lib.BrotliEncoderMaxCompressedSize(len(data)) or (len(data) + 1024)
    out = C.create_string_buffer(cap)
    n = C.c_size_t(cap)
    ok = lib.BrotliEncoderCompress(quality, 22, 0, len(data), bytes(data), C.byref(n), out)
    assert ok == 1
    return out.raw[: n.value]


def boxes(data):
    pos = 0
    while pos + 8 <= len(data):
        size, typ = struct.unpack(">I4s", data[pos:pos + 8])
        hdr = 8
        if size == 1:
            size = struct.unpack(">Q", data[pos + 8:pos + 16])[0]
            hdr = 16
        elif size == 0:
            size = len(data) - pos
        yield typ, data[pos + hdr:pos + size], data[pos:pos + size]
        pos += size


def compress_metadata_boxes(data):
    """Returns the container with every Exif / xml box replaced by a brob box, or None without a Brotli encoder."""
    out = b""
    for typ, payload, raw in boxes(data):
        if typ in (b"Exif", b"xml "):
            z = brotli_compress(payload)
            if z is None:
                return None
            body = typ + z
            out += struct.pack(">I4s", 8 + len(body), b"brob") + body
        else:
            out += raw
    return out
