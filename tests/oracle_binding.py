"""ctypes binding of the CPU oracle (oracle/libfdeflate_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the fdeflate_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ORACLE_DIR, "libfdeflate_oracle.so")

STATUS_NAMES = [
    "Ok", "BadZlibHeader", "InsufficientInput", "InvalidBlockType",
    "InvalidUncompressedBlockLength", "InvalidHlit", "InvalidHdist", "InvalidCodeLengthRepeat",
    "BadCodeLengthHuffmanTree", "BadLiteralLengthHuffmanTree", "BadDistanceHuffmanTree",
    "InvalidLiteralLengthCode", "InvalidDistanceCode", "InputStartsWithRun", "DistanceTooFarBack",
    "WrongChecksum", "ExtraInput", "OutputTooLarge",
]


def build(force=False):
    src = os.path.join(_ORACLE_DIR, "fdeflate_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, u16p, u32p, u64p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64))
        szp = C.POINTER(C.c_size_t)
        L.fdo_adler32.restype = C.c_uint32
        L.fdo_adler32.argtypes = [C.c_void_p, C.c_size_t]
        L.fdo_build_table.restype = C.c_int
        L.fdo_build_table.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, szp, C.c_int,
                                      C.c_int]
        L.fdo_build_decode_tables.restype = C.c_int
        L.fdo_build_decode_tables.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, u16p,
                                              u16p, u8p]
        for name, t in (("fdo_huffman_lengths", u8p), ("fdo_huffman_codes", u16p),
                        ("fdo_litlen_table_entries", u32p), ("fdo_distance_table_entries", u32p),
                        ("fdo_ultrafast_header", u8p)):
            getattr(L, name).restype = t
            getattr(L, name).argtypes = []
        L.fdo_decompressor_new.restype = C.c_void_p
        L.fdo_decompressor_free.argtypes = [C.c_void_p]
        L.fdo_decompressor_ignore_adler32.argtypes = [C.c_void_p]
        L.fdo_decompressor_is_done.restype = C.c_int
        L.fdo_decompressor_is_done.argtypes = [C.c_void_p]
        L.fdo_decompressor_read.restype = C.c_int
        L.fdo_decompressor_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                            C.c_size_t, C.c_size_t, szp, szp]
        L.fdo_decompress_bounded.restype = C.c_int
        L.fdo_decompress_bounded.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, szp,
                                             C.c_int, u32p]
        L.fdo_decompress_by_chunks.restype = C.c_int
        L.fdo_decompress_by_chunks.argtypes = [C.c_void_p, C.c_size_t, C.c_long, C.c_void_p,
                                               C.c_size_t, szp]
        L.fdo_ultrafast_bound.restype = C.c_size_t
        L.fdo_ultrafast_bound.argtypes = [C.c_size_t]
        L.fdo_compress_ultra_fast.restype = C.c_size_t
        L.fdo_compress_ultra_fast.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.fdo_compress_stored.restype = C.c_size_t
        L.fdo_compress_stored.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        for name in ("fdo_compress_level1", "fdo_compress_rle"):
            getattr(L, name).restype = C.c_size_t
            getattr(L, name).argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.fdo_compress_bound.restype = C.c_size_t
        L.fdo_compress_bound.argtypes = [C.c_size_t]
        L.fdo_png_unfilter.restype = C.c_int
        L.fdo_png_unfilter.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
        L.fdo_png_filter.restype = C.c_int
        L.fdo_png_filter.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        L.fdo_inflate_batch.restype = None
        L.fdo_inflate_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]
        L.fdo_deflate_ultrafast_batch.restype = None
        L.fdo_deflate_ultrafast_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_uint64, C.c_int]
        _lib = L
    return _lib


def _buf(b):
    """bytes-like -> (keepalive ndarray, void pointer)"""
    a = np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else b
    a = np.ascontiguousarray(a)
    return a, a.ctypes.data_as(C.c_void_p)


def adler32(data):
    a, p = _buf(data)
    return lib().fdo_adler32(p, a.size)


def decompress_bounded(data, maxlen, ignore_adler32=False):
    """-> (status, output bytes [decoded or partial], adler32 of output)"""
    a, p = _buf(data)
    out = np.zeros(max(maxlen, 1), dtype=np.uint8)
    n = C.c_size_t(0)
    ad = C.c_uint32(0)
    st = lib().fdo_decompress_bounded(p, a.size, out.ctypes.data_as(C.c_void_p), maxlen,
                                      C.byref(n), int(ignore_adler32), C.byref(ad))
    return st, out[:n.value].tobytes(), ad.value


def decompress_by_chunks(data, chunk=0, out_cap=1_000_000):
    a, p = _buf(data)
    out = np.zeros(out_cap, dtype=np.uint8)
    n = C.c_size_t(0)
    st = lib().fdo_decompress_by_chunks(p, a.size, chunk, out.ctypes.data_as(C.c_void_p), out_cap,
                                        C.byref(n))
    return st, out[:n.value].tobytes()


def compress_ultra_fast(data):
    a, p = _buf(data)
    cap = lib().fdo_ultrafast_bound(a.size)
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().fdo_compress_ultra_fast(p, a.size, out.ctypes.data_as(C.c_void_p), cap)
    return out[:n].tobytes()


def _compress_general(fn, data):
    a, p = _buf(data)
    cap = lib().fdo_compress_bound(a.size)
    out = np.zeros(cap, dtype=np.uint8)
    n = fn(p if a.size else None, a.size, out.ctypes.data_as(C.c_void_p), cap)
    assert n > 0
    return out[:n].tobytes()


def compress_level1(data):
    """compress_to_vec (level 1, src/compress/mod.rs:294)"""
    return _compress_general(lib().fdo_compress_level1, data)


def length_limit_events():
    """(trees shortened to 15 bits, to 7 bits) by the general encoder so far -- test instrumentation"""
    a = (C.c_ulong * 2).in_dll(lib(), "fdo_length_limit_events")
    return int(a[0]), int(a[1])


def compress_rle(data):
    """compress_to_vec_rle (src/compress/mod.rs:306)"""
    return _compress_general(lib().fdo_compress_rle, data)


def png_unfilter(filt, row_bytes, bpp):
    """-> (status, pixel bytes)"""
    a, p = _buf(filt)
    rows = a.size // (row_bytes + 1)
    out = np.zeros(max(rows * row_bytes, 1), dtype=np.uint8)
    st = lib().fdo_png_unfilter(p, a.size, row_bytes, bpp, out.ctypes.data_as(C.c_void_p))
    return st, out[:rows * row_bytes].tobytes()


def png_filter(pix, row_bytes, bpp, types):
    a, p = _buf(pix)
    t, tp = _buf(bytes(types))
    rows = a.size // row_bytes
    out = np.zeros(max(rows * (row_bytes + 1), 1), dtype=np.uint8)
    st = lib().fdo_png_filter(p, a.size, row_bytes, bpp, tp, out.ctypes.data_as(C.c_void_p))
    return st, out[:rows * (row_bytes + 1)].tobytes()


def compress_stored(data):
    a, p = _buf(data)
    cap = a.size + 5 * (a.size // 65535 + 2) + 16
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().fdo_compress_stored(p, a.size, out.ctypes.data_as(C.c_void_p), cap)
    return out[:n].tobytes()


class Decompressor:
    """Mirror of fdeflate::Decompressor over the oracle (streaming `read`)."""

    def __init__(self):
        self._d = lib().fdo_decompressor_new()

    def __del__(self):
        if getattr(self, "_d", None):
            lib().fdo_decompressor_free(self._d)
            self._d = None

    def ignore_adler32(self):
        lib().fdo_decompressor_ignore_adler32(self._d)

    def is_done(self):
        return bool(lib().fdo_decompressor_is_done(self._d))

    def read(self, data, output, output_position):
        """output: writable uint8 ndarray.  -> (status, consumed, produced)"""
        a, p = _buf(data)
        c = C.c_size_t(0)
        pr = C.c_size_t(0)
        st = lib().fdo_decompressor_read(self._d, p, a.size, output.ctypes.data_as(C.c_void_p),
                                         output.size, output_position, C.byref(c), C.byref(pr))
        return st, c.value, pr.value


def build_table(lengths, entries, primary_len, is_distance, double_literal):
    """huffman::build_table -> (ok, codes, primary, secondary)"""
    lengths = np.asarray(lengths, dtype=np.uint8)
    ent = np.asarray(entries, dtype=np.uint32)
    codes = np.zeros(max(len(lengths), 1), dtype=np.uint16)
    primary = np.zeros(primary_len, dtype=np.uint32)
    sec = np.zeros(8192, dtype=np.uint16)
    n = C.c_size_t(0)
    ok = lib().fdo_build_table(lengths.ctypes.data_as(C.c_void_p), lengths.size,
                               ent.ctypes.data_as(C.c_void_p) if ent.size else None, ent.size,
                               codes.ctypes.data_as(C.c_void_p), primary.ctypes.data_as(C.c_void_p),
                               primary_len, sec.ctypes.data_as(C.c_void_p), sec.size, C.byref(n),
                               int(is_distance), int(double_literal))
    return bool(ok), codes, primary, sec[:n.value].copy()


def build_decode_tables(hlit, code_lengths):
    cl = np.asarray(code_lengths, dtype=np.uint8)
    assert cl.size == 320
    litlen = np.zeros(4096, dtype=np.uint32)
    dist = np.zeros(512, dtype=np.uint32)
    ec, em, eb = C.c_uint16(0), C.c_uint16(0), C.c_uint8(0)
    st = lib().fdo_build_decode_tables(hlit, cl.ctypes.data_as(C.c_void_p),
                                       litlen.ctypes.data_as(C.c_void_p),
                                       dist.ctypes.data_as(C.c_void_p), C.byref(ec), C.byref(em),
                                       C.byref(eb))
    return st, litlen, dist, (ec.value, em.value, eb.value)


def const_array(name, n, dtype):
    p = getattr(lib(), name)()
    return np.ctypeslib.as_array(p, shape=(n,)).astype(dtype).copy()


def inflate_batch(in_buf, in_off, out_buf, out_off, ignore_adler32=False, nthreads=1):
    n = len(in_off) - 1
    out_len = np.zeros(n, dtype=np.uint32)
    status = np.zeros(n, dtype=np.uint32)
    adler = np.zeros(n, dtype=np.uint32)
    lib().fdo_inflate_batch(in_buf.ctypes.data_as(C.c_void_p), in_off.ctypes.data_as(C.c_void_p),
                            out_buf.ctypes.data_as(C.c_void_p), out_off.ctypes.data_as(C.c_void_p),
                            out_len.ctypes.data_as(C.c_void_p), status.ctypes.data_as(C.c_void_p),
                            adler.ctypes.data_as(C.c_void_p), n, int(ignore_adler32), nthreads)
    return out_len, status, adler


def deflate_ultrafast_batch(in_buf, in_off, out_buf, out_off, nthreads=1):
    n = len(in_off) - 1
    out_len = np.zeros(n, dtype=np.uint32)
    lib().fdo_deflate_ultrafast_batch(in_buf.ctypes.data_as(C.c_void_p),
                                      in_off.ctypes.data_as(C.c_void_p),
                                      out_buf.ctypes.data_as(C.c_void_p),
                                      out_off.ctypes.data_as(C.c_void_p),
                                      out_len.ctypes.data_as(C.c_void_p), n, nthreads)
    return out_len
