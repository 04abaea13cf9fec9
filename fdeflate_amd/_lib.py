"""ctypes loader for libfdeflate_hip.so (the C ABI of include/fdeflate_hip.h).

There is no fallback: if the shared library has not been built, or no GPU is usable, the
calls raise.  Build with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C fdeflate_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FDH_LIB: diagnostics only (tools/segdiag.py loads an instrumented build of the same library)
SO_PATH = os.environ.get("FDH_LIB") or os.path.join(_HERE, "libfdeflate_hip.so")

_lib = None


class FdeflateHipError(RuntimeError):
    """Infrastructure failure reported by the C ABI (not a per-stream decode error)."""


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise FdeflateHipError(
            "%s is missing: the HIP extension has not been built (make -C fdeflate_amd/csrc); "
            "fdeflate_amd has no CPU fallback" % SO_PATH)
    L = C.CDLL(SO_PATH)
    vp, u64, u32, sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_size_t
    L.fdh_version.restype = u32
    L.fdh_status_name.restype = C.c_char_p
    L.fdh_status_name.argtypes = [u32]
    L.fdh_last_error.restype = C.c_char_p
    L.fdh_device_count.restype = C.c_int
    L.fdh_ultrafast_bound.restype = u64
    L.fdh_ultrafast_bound.argtypes = [u64]
    L.fdh_inflate_batch.restype = C.c_int
    L.fdh_inflate_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, u64, u32, vp]
    L.fdh_inflate_batch_resumable.restype = C.c_int
    L.fdh_inflate_batch_resumable.argtypes = [vp, vp, vp, vp, vp, vp, vp, u64, u32, vp, vp]
    pp = C.POINTER(C.c_void_p)
    L.fdh_deflate_ultrafast_batch.restype = C.c_int
    L.fdh_deflate_ultrafast_batch.argtypes = [vp, vp, vp, vp, vp, u64, vp]
    L.fdh_stored_size.restype = u64
    L.fdh_stored_size.argtypes = [u64]
    L.fdh_deflate_stored_batch.restype = C.c_int
    L.fdh_deflate_stored_batch.argtypes = [vp, vp, vp, vp, vp, u64, vp]
    L.fdh_compress_to_vec_stored.restype = C.c_int
    L.fdh_compress_to_vec_stored.argtypes = [vp, sz, pp, C.POINTER(sz)]
    L.fdh_debug_build_tables.restype = C.c_int
    L.fdh_debug_build_tables.argtypes = [vp, u32, vp, vp, vp, vp]
    pp = C.POINTER(C.c_void_p)
    L.fdh_decompress_to_vec.restype = C.c_int
    L.fdh_decompress_to_vec.argtypes = [vp, sz, pp, C.POINTER(sz), C.POINTER(u32)]
    L.fdh_decompress_to_vec_bounded.restype = C.c_int
    L.fdh_decompress_to_vec_bounded.argtypes = [vp, sz, sz, pp, C.POINTER(sz), C.POINTER(u32)]
    L.fdh_compress_to_vec_ultra_fast.restype = C.c_int
    L.fdh_compress_to_vec_ultra_fast.argtypes = [vp, sz, pp, C.POINTER(sz)]
    L.fdh_free.argtypes = [vp]
    L.fdh_png_unfilter_batch.restype = C.c_int
    L.fdh_png_unfilter_batch.argtypes = [vp, vp, vp, vp, vp, u64, u32, u32, vp]
    L.fdh_png_filter_batch.restype = C.c_int
    L.fdh_png_filter_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, u64, u32, u32, vp]
    L.fdh_inflate_png_batch.restype = C.c_int
    L.fdh_inflate_png_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u64, u32, u32, u32, vp]
    L.fdh_png_filter_deflate_ultrafast_batch.restype = C.c_int
    L.fdh_png_filter_deflate_ultrafast_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, u64, u32, u32, vp]
    L.fdh_init.restype = C.c_int
    L.fdh_init.argtypes = [u64]
    L.fdh_shutdown.restype = C.c_int
    L.fdh_multi_device_count.restype = C.c_int
    L.fdh_multi_uses_rccl.restype = C.c_int
    L.fdh_inflate_batch_multi.restype = C.c_int
    L.fdh_inflate_batch_multi.argtypes = [vp, u32, u32, u64]
    L.fdh_compress_bound.restype = u64
    L.fdh_compress_bound.argtypes = [u64]
    L.fdh_deflate_general_batch.restype = C.c_int
    L.fdh_deflate_general_batch.argtypes = [vp, vp, vp, vp, vp, u64, u32, vp]
    L.fdh_compress_to_vec.restype = C.c_int
    L.fdh_compress_to_vec.argtypes = [vp, sz, pp, C.POINTER(sz)]
    L.fdh_compress_to_vec_rle.restype = C.c_int
    L.fdh_compress_to_vec_rle.argtypes = [vp, sz, pp, C.POINTER(sz)]
    L.fdh_decompressor_new.restype = vp
    L.fdh_decompressor_new.argtypes = []
    L.fdh_decompressor_free.argtypes = [vp]
    L.fdh_decompressor_free.restype = None
    L.fdh_decompressor_ignore_adler32.argtypes = [vp]
    L.fdh_decompressor_ignore_adler32.restype = None
    L.fdh_decompressor_is_done.argtypes = [vp]
    L.fdh_decompressor_is_done.restype = C.c_int
    L.fdh_decompressor_attempts.argtypes = [vp]
    L.fdh_decompressor_attempts.restype = C.c_uint64
    L.fdh_decompressor_decoded_bytes.argtypes = [vp]
    L.fdh_decompressor_decoded_bytes.restype = C.c_uint64
    L.fdh_decompressor_device_bytes.argtypes = [vp]
    L.fdh_decompressor_device_bytes.restype = C.c_uint64
    L.fdh_decompressor_read.restype = C.c_int
    L.fdh_decompressor_read.argtypes = [vp, vp, sz, vp, sz, sz, C.POINTER(sz), C.POINTER(sz), C.POINTER(u32)]
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "fdh_version", "fdh_status_name", "fdh_last_error", "fdh_device_count", "fdh_ultrafast_bound",
    "fdh_inflate_batch", "fdh_inflate_batch_resumable", "fdh_deflate_ultrafast_batch", "fdh_debug_build_tables",
    "fdh_decompress_to_vec", "fdh_decompress_to_vec_bounded", "fdh_compress_to_vec_ultra_fast",
    "fdh_free", "fdh_stored_size", "fdh_deflate_stored_batch", "fdh_compress_to_vec_stored",
    "fdh_decompressor_new", "fdh_decompressor_free", "fdh_decompressor_ignore_adler32",
    "fdh_decompressor_is_done", "fdh_decompressor_read", "fdh_decompressor_attempts", "fdh_decompressor_decoded_bytes", "fdh_decompressor_device_bytes",
    "fdh_compress_bound", "fdh_deflate_general_batch", "fdh_compress_to_vec", "fdh_compress_to_vec_rle",
    "fdh_png_unfilter_batch", "fdh_png_filter_batch", "fdh_inflate_png_batch", "fdh_png_filter_deflate_ultrafast_batch",
    "fdh_init", "fdh_shutdown", "fdh_multi_device_count", "fdh_multi_uses_rccl", "fdh_inflate_batch_multi",
]


class Shard(C.Structure):
    """fdh_shard_t"""
    _fields_ = [("in_", C.c_void_p), ("in_off", C.c_void_p), ("out", C.c_void_p), ("out_off", C.c_void_p),
                ("out_len", C.c_void_p), ("status", C.c_void_p), ("adler", C.c_void_p), ("n", C.c_uint64),
                ("meta_all", C.c_void_p)]


def check(rc):
    if rc != 0:
        raise FdeflateHipError("fdeflate_hip error %d: %s" % (rc, lib().fdh_last_error().decode()))
