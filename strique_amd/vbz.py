"""VBZ (HDF5 filter 32020, Oxford Nanopore's `vbz_compression`): what MinKNOW compresses the raw signal of
multi-read fast5 files with.  The reference reads such files through h5py and the `hdf5plugin` / vbz plugin when the
user has installed it (STRique_lib/fast5Index.py:76-84 just opens the dataset); here the chunk format is decoded
directly, zstd through the system's libzstd (ctypes).

UNPINNED: neither the plugin nor a VBZ-compressed file exists in this image or in the reference tree, so this
module is written from the format as published with the plugin's sources and is checked only against hand-worked
vectors and its own encoder.  Every layer is length-checked (the variable-byte stream must be consumed exactly, the
decoded size must equal the size the chunk header states), so a chunk in another layout raises instead of
yielding samples.

Chunk layout (filter client data = [version, integer size, delta + zig-zag flag, zstd level]):
    u32 little-endian   size of the decoded chunk in bytes
    payload             zstd frame if the zstd level is not 0, else the variable-byte stream itself
Variable-byte stream of n = size / integer_size integers:
    version 0: every integer is widened to 32 bits; if the flag is set it is replaced by the zig-zag code of its
               difference to the previous integer (first: to 0); then classic StreamVByte: (n + 3) // 4 key bytes,
               two bits per integer (number of data bytes - 1, first integer in the low bits), then the data bytes,
               little-endian, back to back.
    version 1, 2-byte integers: the same on 16 bits with one key bit per integer ((n + 7) // 8 key bytes; 0: one data
               byte, 1: two); 4-byte integers as in version 0.
"""
import ctypes
import ctypes.util
import struct

import numpy as np

FILTER_ID = 32020
_zstd = None


def _libzstd():
    global _zstd
    if _zstd is None:
        name = ctypes.util.find_library("zstd") or "libzstd.so.1"
        try:
            lib = ctypes.CDLL(name)
        except OSError as e:
            raise NotImplementedError("VBZ-compressed signal: libzstd is not available (%s)" % e)
        lib.ZSTD_decompress.restype = ctypes.c_size_t
        lib.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        lib.ZSTD_compress.restype = ctypes.c_size_t
        lib.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        lib.ZSTD_compressBound.restype = ctypes.c_size_t
        lib.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
        lib.ZSTD_isError.restype = ctypes.c_uint
        lib.ZSTD_isError.argtypes = [ctypes.c_size_t]
        lib.ZSTD_getFrameContentSize.restype = ctypes.c_ulonglong
        lib.ZSTD_getFrameContentSize.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        _zstd = lib
    return _zstd


def zstd_decompress(data, bound):
    lib = _libzstd()
    src = (ctypes.c_char * len(data)).from_buffer_copy(data)
    size = lib.ZSTD_getFrameContentSize(src, len(data))
    cap = int(size) if size < (1 << 40) else int(bound)          # unknown / error codes are huge values
    dst = ctypes.create_string_buffer(max(cap, 1))
    n = lib.ZSTD_decompress(dst, cap, src, len(data))
    if lib.ZSTD_isError(n):
        raise ValueError("VBZ chunk: zstd frame does not decode")
    return dst.raw[:n]


def zstd_compress(data, level=1):
    lib = _libzstd()
    cap = lib.ZSTD_compressBound(len(data))
    dst = ctypes.create_string_buffer(cap)
    src = (ctypes.c_char * max(len(data), 1)).from_buffer_copy(data or b"\0")
    n = lib.ZSTD_compress(dst, cap, src, len(data), level)
    if lib.ZSTD_isError(n):
        raise ValueError("zstd compression failed")
    return dst.raw[:n]


# ---------------------------------------------------------------------------------------------
# variable-byte layers
# ---------------------------------------------------------------------------------------------
def _gather(data, start, lens, width):
    """little-endian integers of lens[i] bytes each, back to back from data[start:]"""
    off = np.cumsum(lens) - lens + start
    total = int(start + lens.sum())
    if total != len(data):
        raise ValueError("VBZ chunk: variable-byte stream of %d bytes, %d expected" % (len(data), total))
    vals = np.zeros(len(lens), np.uint32)
    for b in range(width):
        m = lens > b
        vals[m] |= data[off[m] + b].astype(np.uint32) << np.uint32(8 * b)
    return vals


def svb32_decode(stream, n):
    data = np.frombuffer(stream, np.uint8)
    nk = (n + 3) // 4
    if len(data) < nk:
        raise ValueError("VBZ chunk: key bytes missing")
    keys = data[:nk]
    codes = ((keys[:, None] >> np.array([0, 2, 4, 6], np.uint8)) & 3).reshape(-1)[:n]
    return _gather(data, nk, codes.astype(np.int64) + 1, 4)


def svb16_decode(stream, n):
    data = np.frombuffer(stream, np.uint8)
    nk = (n + 7) // 8
    if len(data) < nk:
        raise ValueError("VBZ chunk: key bytes missing")
    bits = np.unpackbits(data[:nk], bitorder="little")[:n]
    return _gather(data, nk, bits.astype(np.int64) + 1, 2)


def _scatter(vals, lens, keys):
    out = np.zeros(int(lens.sum()), np.uint8)
    off = np.cumsum(lens) - lens
    for b in range(int(lens.max()) if len(lens) else 0):
        m = lens > b
        out[off[m] + b] = ((vals[m] >> np.uint32(8 * b)) & 0xFF).astype(np.uint8)
    return keys.tobytes() + out.tobytes()


def svb32_encode(vals):
    vals = np.asarray(vals, np.uint32)
    lens = 1 + (vals > 0xFF).astype(np.int64) + (vals > 0xFFFF) + (vals > 0xFFFFFF)
    codes = np.zeros(((len(vals) + 3) // 4) * 4, np.uint8); codes[:len(vals)] = lens - 1
    keys = (codes.reshape(-1, 4) << np.array([0, 2, 4, 6], np.uint8)).sum(axis=1).astype(np.uint8)
    return _scatter(vals, lens, keys)


def svb16_encode(vals):
    vals = np.asarray(vals, np.uint32)
    lens = 1 + (vals > 0xFF).astype(np.int64)
    bits = np.zeros(((len(vals) + 7) // 8) * 8, np.uint8); bits[:len(vals)] = lens - 1
    return _scatter(vals, lens, np.packbits(bits, bitorder="little"))


_native = False


def _native_decode(payload, n, key_bits, zigzag, isize):
    """the variable-byte layer through libstrique_hip's host helper (strq_svb_decode; ctypes releases the GIL, so the
    reader threads of `count` decode in parallel); None when the library is not built -- the numpy path below is the
    same arithmetic"""
    global _native
    if _native is False:
        try:
            from . import ffi
            lib = ffi.load_library()
            lib.strq_svb_decode.restype = ctypes.c_int64
            lib.strq_svb_decode.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                            ctypes.c_int32, ctypes.c_void_p]
            _native = lib
        except (ImportError, OSError, AttributeError):
            _native = None
    if _native is None:
        return None
    out = np.empty(n, "<u%d" % isize)
    used = _native.strq_svb_decode(payload, len(payload), n, key_bits, zigzag, isize, out.ctypes.data)
    if used != len(payload):
        raise ValueError("VBZ chunk: variable-byte stream of %d bytes, %d consumed" % (len(payload), used))
    return out.tobytes()


# ---------------------------------------------------------------------------------------------
# the filter
# ---------------------------------------------------------------------------------------------
def decode(chunk, cd_values):
    """bytes of one dataset chunk, decoded.  cd_values: the filter's client data from the pipeline message."""
    cd = list(cd_values) + [0] * 4
    version, isize, zigzag, level = cd[0], cd[1], cd[2], cd[3]
    if version not in (0, 1) or isize not in (0, 1, 2, 4):
        raise NotImplementedError("VBZ version %d / integer size %d" % (version, isize))
    if len(chunk) < 4:
        raise ValueError("VBZ chunk: header missing")
    size, = struct.unpack_from("<I", chunk, 0)
    payload = bytes(chunk[4:])
    n = size // isize if isize else 0
    if level:
        payload = zstd_decompress(payload, (n + 3) // 4 + 4 * n + size + 64)
    if isize == 0 or (isize == 1 and version == 1):
        out = payload                                    # no variable-byte layer
    else:
        if isize * n != size:
            raise ValueError("VBZ chunk: size is not a multiple of the integer size")
        narrow = version == 1 and isize == 2
        fast = _native_decode(payload, n, 1 if narrow else 2, 1 if zigzag else 0, isize)
        if fast is not None:
            if len(fast) != size:
                raise ValueError("VBZ chunk: %d bytes decoded, %d stated" % (len(fast), size))
            return fast
        u = svb16_decode(payload, n) if narrow else svb32_decode(payload, n)
        if zigzag:
            if narrow:
                d = ((u >> 1) ^ (0 - (u & 1))).astype(np.uint16)
                v = np.cumsum(d.astype(np.uint64)).astype(np.uint16)
            else:
                d = (u >> np.uint32(1)) ^ (np.uint32(0) - (u & np.uint32(1)))
                v = np.cumsum(d.astype(np.uint64)).astype(np.uint32)
        else:
            v = u
        out = v.astype("<u%d" % isize).tobytes()
    if len(out) != size:
        raise ValueError("VBZ chunk: %d bytes decoded, %d stated" % (len(out), size))
    return out


def encode(values, version=0, zigzag=True, level=1):
    """One chunk holding the int16 array `values` (what `decode` reverses); (bytes, cd_values)."""
    a = np.ascontiguousarray(values, "<i2")
    narrow = version == 1
    if zigzag:
        if narrow:
            d = np.diff(a.astype(np.uint16), prepend=np.uint16(0)).astype(np.int16)
            u = ((d.astype(np.int32) << 1) ^ (d.astype(np.int32) >> 15)).astype(np.uint16).astype(np.uint32)
        else:
            w = a.astype(np.int32)
            d = np.diff(w, prepend=np.int32(0)).astype(np.int32)
            u = ((d.astype(np.int64) << 1) ^ (d.astype(np.int64) >> 31)).astype(np.uint32)
    else:
        u = a.astype(np.uint16).astype(np.uint32) if narrow else a.astype(np.int32).astype(np.uint32)
    stream = svb16_encode(u) if narrow else svb32_encode(u)
    if level:
        stream = zstd_compress(stream, level)
    return struct.pack("<I", a.nbytes) + stream, (version, 2, 1 if zigzag else 0, level)
