"""Host-side mirror of fdeflate's public API for the PNG path (reference src/lib.rs:29-36),
implemented over the C ABI (include/fdeflate_hip.h).  Same names, argument meaning and error
behaviour as the Rust crate:

    decompress_to_vec(input) -> bytes                  raises DecompressionError
    decompress_to_vec_bounded(input, maxlen) -> bytes  raises DecompressionError / OutputTooLarge
    compress_to_vec_ultra_fast(input) -> bytes

and the batched entry points the GPU exists for:

    inflate_batch(...), deflate_ultrafast_batch(...)   torch uint8/int64 tensors on the device

torch is used only as the owner of device memory and streams.
"""
import ctypes as C

from . import _lib

STATUS_NAMES = [
    "Ok", "BadZlibHeader", "InsufficientInput", "InvalidBlockType",
    "InvalidUncompressedBlockLength", "InvalidHlit", "InvalidHdist", "InvalidCodeLengthRepeat",
    "BadCodeLengthHuffmanTree", "BadLiteralLengthHuffmanTree", "BadDistanceHuffmanTree",
    "InvalidLiteralLengthCode", "InvalidDistanceCode", "InputStartsWithRun", "DistanceTooFarBack",
    "WrongChecksum", "ExtraInput", "OutputTooLarge",
]
OUTPUT_TOO_LARGE = 17
FLAG_IGNORE_ADLER32 = 1
FLAG_SERIAL_ONLY = 2
FLAG_GENERAL_ONLY = 4
FLAG_NO_RECHECK = 8
FLAG_FORCE_LANES = 16
FLAG_NO_LANES = 32
FLAG_FIRST_ONLY = 64
FLAG_NO_SEGMENTS = 128
FLAG_SPANS = 256
FLAG_NO_FAST_GENERAL = 512
FLAG_NO_INTERVALS = 0x400   # tests / A-B: skip the interval kernel (inflate_seg2.h)
FLAG_INTERVALS_ONLY = 0x800  # debug: run only the interval kernel (what it leaves stays PENDING)
FLAG_NO_LANDING = 0x10000    # tests / A-B: skip the landing decoder (inflate_seg3.h)
FLAG_LANDING_ONLY = 0x20000  # debug: run only the landing decoder (what it leaves stays PENDING)
FLAG_NO_OVERLAP = 0x100000   # tests / A-B: the LZ-window kernel behind the canonical kernels, not beside them
FLAG_NO_LEAN_WRITE = 0x80000  # tests / A-B: the landing decoder always takes the interval decoder's general writing pass
FLAG_TAIL_LONG = 0x200000    # tests / A-B: behind the landing decoder always the five kernels of rounds 3-5
FLAG_TAIL_SHORT = 0x400000   # tests / A-B: behind the landing decoder always the exact kernel alone
FLAG_ORDER_ONCE = 0x800000   # tests / A-B: the streams without the ultra-fast prefix listed in one launch, in no order
FLAG_ORDER_TWICE = 0x1000000  # tests / A-B: ... in two, the long ones first


class DecompressionError(Exception):
    """Mirror of fdeflate::DecompressionError (src/decompress.rs:14-48); `.kind` is the variant."""

    def __init__(self, status):
        self.status = int(status)
        self.kind = STATUS_NAMES[self.status] if self.status < len(STATUS_NAMES) else "Unknown"
        super().__init__(self.kind)


class OutputTooLarge(Exception):
    """Mirror of BoundedDecompressionError::OutputTooLarge (src/decompress.rs:1097-1101)."""

    def __init__(self, partial_output):
        self.partial_output = partial_output
        super().__init__("OutputTooLarge")


class Decompressor:
    """Mirror of fdeflate::Decompressor (src/decompress.rs:96-342) over fdh_decompressor_*:

        d = Decompressor(); d.ignore_adler32()
        consumed, produced = d.read(input, output, output_position)   # raises DecompressionError
        d.is_done()

    `output` is a writable buffer (bytearray / numpy uint8 array); bytes are written at
    output[output_position : output_position + produced]."""

    def __init__(self):
        self._L = _lib.lib()
        self._d = self._L.fdh_decompressor_new()
        if not self._d:
            raise MemoryError("fdh_decompressor_new")

    def __del__(self):
        d, self._d = getattr(self, "_d", None), None
        if d:
            self._L.fdh_decompressor_free(d)

    def ignore_adler32(self):
        self._L.fdh_decompressor_ignore_adler32(self._d)

    def is_done(self):
        return bool(self._L.fdh_decompressor_is_done(self._d))

    def attempts(self):
        """Decode attempts made so far (introspection, fdh_decompressor_attempts)."""
        return int(self._L.fdh_decompressor_attempts(self._d))

    def decoded_bytes(self):
        """Output bytes decoded by all attempts together (introspection, fdh_decompressor_decoded_bytes)."""
        return int(self._L.fdh_decompressor_decoded_bytes(self._d))

    def device_bytes(self):
        """The most device memory the object's buffers have held together (introspection, fdh_decompressor_device_bytes)."""
        return int(self._L.fdh_decompressor_device_bytes(self._d))

    def read(self, data, output, output_position):
        """Decompressor::read (src/decompress.rs:179-337) -> (consumed, produced); raises DecompressionError.  What is
        not consumed (more than 192 KiB, or the room, waiting unread on the device) is to be offered again."""
        data = bytes(data)
        mv = memoryview(output)
        if mv.readonly or mv.itemsize != 1 or not mv.contiguous:
            raise ValueError("output must be a writable contiguous byte buffer")
        n = mv.nbytes
        obuf = (C.c_uint8 * n).from_buffer(mv) if n else None
        c, p, st = C.c_size_t(), C.c_size_t(), C.c_uint32()
        _lib.check(self._L.fdh_decompressor_read(self._d, data, len(data), obuf, n, output_position,
                                                 C.byref(c), C.byref(p), C.byref(st)))
        if st.value != 0:
            raise DecompressionError(st.value)
        return c.value, p.value


def _take(ptr, n):
    try:
        return C.string_at(ptr, n) if n else b""
    finally:
        _lib.lib().fdh_free(ptr)


def decompress_to_vec_bounded(data, maxlen):
    """fdeflate::decompress_to_vec_bounded (src/decompress.rs:1111)."""
    L = _lib.lib()
    data = bytes(data)
    out = C.c_void_p()
    n = C.c_size_t()
    st = C.c_uint32()
    _lib.check(L.fdh_decompress_to_vec_bounded(data, len(data), maxlen, C.byref(out), C.byref(n), C.byref(st)))
    buf = _take(out, n.value)
    if st.value == 0:
        return buf
    if st.value == OUTPUT_TOO_LARGE:
        raise OutputTooLarge(buf)
    raise DecompressionError(st.value)


def decompress_to_vec(data):
    """fdeflate::decompress_to_vec (src/decompress.rs:1079)."""
    L = _lib.lib()
    data = bytes(data)
    out = C.c_void_p()
    n = C.c_size_t()
    st = C.c_uint32()
    _lib.check(L.fdh_decompress_to_vec(data, len(data), C.byref(out), C.byref(n), C.byref(st)))
    buf = _take(out, n.value)
    if st.value != 0:
        raise DecompressionError(st.value)
    return buf


def compress_to_vec_ultra_fast(data):
    """fdeflate::compress_to_vec_ultra_fast (src/compress/mod.rs:313)."""
    L = _lib.lib()
    data = bytes(data)
    out = C.c_void_p()
    n = C.c_size_t()
    _lib.check(L.fdh_compress_to_vec_ultra_fast(data, len(data), C.byref(out), C.byref(n)))
    return _take(out, n.value)


def ultrafast_bound(n):
    return int(_lib.lib().fdh_ultrafast_bound(int(n)))


def compress_to_vec_stored(data):
    """fdeflate::compress_to_vec_with_level(data, 0) (src/compress/mod.rs:299): stored blocks only."""
    L = _lib.lib()
    data = bytes(data)
    out = C.c_void_p()
    n = C.c_size_t()
    _lib.check(L.fdh_compress_to_vec_stored(data, len(data), C.byref(out), C.byref(n)))
    return _take(out, n.value)


def compress_to_vec(data):
    """fdeflate::compress_to_vec (src/compress/mod.rs:294): level 1 in this snapshot."""
    L = _lib.lib()
    data = bytes(data)
    out = C.c_void_p()
    n = C.c_size_t()
    _lib.check(L.fdh_compress_to_vec(data, len(data), C.byref(out), C.byref(n)))
    return _take(out, n.value)


def compress_to_vec_rle(data):
    """fdeflate::compress_to_vec_rle (src/compress/mod.rs:306)."""
    L = _lib.lib()
    data = bytes(data)
    out = C.c_void_p()
    n = C.c_size_t()
    _lib.check(L.fdh_compress_to_vec_rle(data, len(data), C.byref(out), C.byref(n)))
    return _take(out, n.value)


def compress_bound(n):
    return int(_lib.lib().fdh_compress_bound(int(n)))


MODE_LEVEL1 = 1
MODE_RLE = 2


def deflate_general_batch(raw, in_off, out, out_off, mode, out_len=None):
    """Level-1 / RLE encode of n buffers (fdh_deflate_general_batch): a parser kernel (one stream per
    lane) that records the back-references, then a block-writer kernel (one stream per wavefront).
    Returns when the work has finished."""
    import torch
    n = in_off.numel() - 1
    if out_len is None:
        out_len = torch.empty(n, dtype=torch.int32, device=raw.device)
    with _OnDevice(raw, in_off, out, out_off, out_len) as stream:
        _lib.check(_lib.lib().fdh_deflate_general_batch(_ptr(raw), _ptr(in_off), _ptr(out), _ptr(out_off),
                                                       _ptr(out_len), n, mode, C.c_void_p(stream)))
    return out_len


def stored_size(n):
    return int(_lib.lib().fdh_stored_size(int(n)))


# ------------------------------------------------------------------------------------------
# batched device entry points
# ------------------------------------------------------------------------------------------

def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _OnDevice:
    """Checks that every tensor lives on ONE GPU and makes that GPU current for the duration of
    the call: the C ABI launches on the current device (hipGetDevice) and on the stream it is
    handed, so both must belong to the tensors' device even when another one is current."""

    def __init__(self, *ts):
        import torch
        dev = None
        for t in ts:
            if t is None:
                continue
            if not t.is_cuda:
                raise ValueError("batched entry points take device tensors (HBM resident)")
            if not t.is_contiguous():
                raise ValueError("tensors must be contiguous")
            if dev is None:
                dev = t.device
            elif t.device != dev:
                raise ValueError("all tensors of one call must live on the same GPU (%s vs %s)" % (dev, t.device))
        if dev is None:
            raise ValueError("no tensors")
        self.dev = dev
        self._guard = torch.cuda.device(dev)

    def __enter__(self):
        import torch
        self._guard.__enter__()
        return torch.cuda.current_stream(self.dev).cuda_stream

    def __exit__(self, *exc):
        return self._guard.__exit__(*exc)


def inflate_batch(comp, in_off, out, out_off, out_len=None, status=None, adler=None, flags=0):
    """One-shot decode of n zlib streams (fdh_inflate_batch).  All tensors on the device:
    comp/out uint8, in_off/out_off int64 [n+1], out_len/status/adler int32 [n] (allocated when
    None).  Enqueued on torch's current stream; returns (out_len, status, adler)."""
    import torch
    n = in_off.numel() - 1
    dev = comp.device
    if out_len is None:
        out_len = torch.empty(n, dtype=torch.int32, device=dev)
    if status is None:
        status = torch.empty(n, dtype=torch.int32, device=dev)
    if adler is None:
        adler = torch.empty(n, dtype=torch.int32, device=dev)
    with _OnDevice(comp, in_off, out, out_off, out_len, status, adler) as stream:
        _lib.check(_lib.lib().fdh_inflate_batch(_ptr(comp), _ptr(in_off), _ptr(out), _ptr(out_off),
                                               _ptr(out_len), _ptr(status), _ptr(adler), n, flags,
                                               C.c_void_p(stream)))
    return out_len, status, adler


def inflate_batch_resumable(comp, in_off, out, out_off, resume, out_len=None, status=None, adler=None, flags=0, resume_in=False):
    """fdh_inflate_batch_resumable: as inflate_batch; `resume` (int32 [n, 4] on the device) receives, for every
    stream that ended InsufficientInput / OutputTooLarge, the place from which a later call can go on, and with
    resume_in says where each stream is taken up in this call (the slots then hold the output so far)."""
    import torch
    n = in_off.numel() - 1
    dev = comp.device
    if out_len is None:
        out_len = torch.empty(n, dtype=torch.int32, device=dev)
    if status is None:
        status = torch.empty(n, dtype=torch.int32, device=dev)
    if adler is None:
        adler = torch.empty(n, dtype=torch.int32, device=dev)
    assert resume.dtype == torch.int32 and resume.numel() == 4 * n and resume.is_contiguous()
    with _OnDevice(comp, in_off, out, out_off, out_len, status, adler, resume) as stream:
        _lib.check(_lib.lib().fdh_inflate_batch_resumable(_ptr(comp), _ptr(in_off), _ptr(out), _ptr(out_off),
                                                         _ptr(out_len), _ptr(status), _ptr(adler), n,
                                                         flags | (0x8000 if resume_in else 0), _ptr(resume),
                                                         C.c_void_p(stream)))
    return out_len, status, adler


def deflate_ultrafast_batch(raw, in_off, out, out_off, out_len=None):
    """Ultra-fast encode of n buffers (fdh_deflate_ultrafast_batch); returns out_len (int32)."""
    import torch
    n = in_off.numel() - 1
    if out_len is None:
        out_len = torch.empty(n, dtype=torch.int32, device=raw.device)
    with _OnDevice(raw, in_off, out, out_off, out_len) as stream:
        _lib.check(_lib.lib().fdh_deflate_ultrafast_batch(_ptr(raw), _ptr(in_off), _ptr(out), _ptr(out_off),
                                                         _ptr(out_len), n, C.c_void_p(stream)))
    return out_len


def deflate_stored_batch(raw, in_off, out, out_off, out_len=None):
    """Level-0 (stored) encode of n buffers (fdh_deflate_stored_batch); returns out_len (int32)."""
    import torch
    n = in_off.numel() - 1
    if out_len is None:
        out_len = torch.empty(n, dtype=torch.int32, device=raw.device)
    with _OnDevice(raw, in_off, out, out_off, out_len) as stream:
        _lib.check(_lib.lib().fdh_deflate_stored_batch(_ptr(raw), _ptr(in_off), _ptr(out), _ptr(out_off),
                                                      _ptr(out_len), n, C.c_void_p(stream)))
    return out_len


def png_unfilter_batch(filt, filt_off, pix, pix_off, row_bytes, bpp, png_status=None):
    """PNG scanline reconstruction of n images, one image per wavefront (fdh_png_unfilter_batch)."""
    import torch
    n = filt_off.numel() - 1
    if png_status is None:
        png_status = torch.empty(n, dtype=torch.int32, device=filt.device)
    with _OnDevice(filt, filt_off, pix, pix_off, png_status) as stream:
        _lib.check(_lib.lib().fdh_png_unfilter_batch(_ptr(filt), _ptr(filt_off), _ptr(pix), _ptr(pix_off),
                                                    _ptr(png_status), n, row_bytes, bpp, C.c_void_p(stream)))
    return png_status


def png_filter_batch(pix, pix_off, types, types_off, filt, filt_off, row_bytes, bpp, png_status=None):
    """PNG scanline filtering with the given per-row filter types (fdh_png_filter_batch)."""
    import torch
    n = pix_off.numel() - 1
    if png_status is None:
        png_status = torch.empty(n, dtype=torch.int32, device=pix.device)
    with _OnDevice(pix, pix_off, types, types_off, filt, filt_off, png_status) as stream:
        _lib.check(_lib.lib().fdh_png_filter_batch(_ptr(pix), _ptr(pix_off), _ptr(types), _ptr(types_off), _ptr(filt),
                                                  _ptr(filt_off), _ptr(png_status), n, row_bytes, bpp,
                                                  C.c_void_p(stream)))
    return png_status


def png_filter_deflate_ultrafast_batch(pix, pix_off, types, types_off, out, out_off, row_bytes, bpp):
    """Filter n images with the given per-row types and ultra-fast-encode the filtered bytes in one
    kernel, no intermediate buffer (fdh_png_filter_deflate_ultrafast_batch) -> (out_len, png_status)."""
    import torch
    n = pix_off.numel() - 1
    out_len = torch.empty(n, dtype=torch.int32, device=pix.device)
    png_status = torch.empty(n, dtype=torch.int32, device=pix.device)
    with _OnDevice(pix, pix_off, types, types_off, out, out_off, out_len, png_status) as stream:
        _lib.check(_lib.lib().fdh_png_filter_deflate_ultrafast_batch(
            _ptr(pix), _ptr(pix_off), _ptr(types), _ptr(types_off), _ptr(out), _ptr(out_off), _ptr(out_len),
            _ptr(png_status), n, row_bytes, bpp, C.c_void_p(stream)))
    return out_len, png_status


def inflate_png_batch(comp, in_off, filt, filt_off, pix, pix_off, row_bytes, bpp, flags=0):
    """Decode n IDAT-style zlib streams and reconstruct their scanlines in one call
    (fdh_inflate_png_batch) -> (out_len, status, adler, png_status)."""
    import torch
    n = in_off.numel() - 1
    dev = comp.device
    out_len = torch.empty(n, dtype=torch.int32, device=dev)
    status = torch.empty(n, dtype=torch.int32, device=dev)
    adler = torch.empty(n, dtype=torch.int32, device=dev)
    png_status = torch.empty(n, dtype=torch.int32, device=dev)
    with _OnDevice(comp, in_off, filt, filt_off, pix, pix_off) as stream:
        _lib.check(_lib.lib().fdh_inflate_png_batch(_ptr(comp), _ptr(in_off), _ptr(filt), _ptr(filt_off), _ptr(out_len),
                                                   _ptr(status), _ptr(adler), _ptr(pix), _ptr(pix_off),
                                                   _ptr(png_status), n, flags, row_bytes, bpp, C.c_void_p(stream)))
    return out_len, status, adler, png_status


def inflate_batch_multi(shards, flags=0, gather=True):
    """fdh_inflate_batch_multi from one process: `shards` = one tuple (comp, in_off, out, out_off) of
    tensors per GPU selected by init_devices(); returns per shard (out_len, status, adler) and, with
    gather, the all-gathered results [n_shards, 3, stride] on every device."""
    import torch
    L = _lib.lib()
    n_sh = len(shards)
    arr = (_lib.Shard * n_sh)()
    keep, results, metas = [], [], []
    stride = max(s[1].numel() - 1 for s in shards)
    for i, (comp, in_off, out, out_off) in enumerate(shards):
        dev = comp.device
        n = in_off.numel() - 1
        ol = torch.empty(n, dtype=torch.int32, device=dev)
        st = torch.empty(n, dtype=torch.int32, device=dev)
        ad = torch.empty(n, dtype=torch.int32, device=dev)
        meta = torch.empty((n_sh, 3, stride), dtype=torch.int32, device=dev) if gather else None
        keep.append((comp, in_off, out, out_off))
        results.append((ol, st, ad))
        metas.append(meta)
        arr[i] = _lib.Shard(comp.data_ptr(), in_off.data_ptr(), out.data_ptr(), out_off.data_ptr(), ol.data_ptr(),
                            st.data_ptr(), ad.data_ptr(), n, meta.data_ptr() if gather else None)
    for t in keep:
        torch.cuda.synchronize(t[0].device)   # the call runs on the library's own streams
    _lib.check(L.fdh_inflate_batch_multi(C.byref(arr), n_sh, flags, stride))
    return results, metas


def init_devices(mask=0):
    """fdh_init: select the GPUs (bit mask, 0 = all visible) for inflate_batch_multi."""
    _lib.check(_lib.lib().fdh_init(mask))
    return _lib.lib().fdh_multi_device_count()


def multi_uses_rccl():
    """True if inflate_batch_multi gathers through RCCL (more than one device, or FDH_MULTI_FORCE_RCCL=1)."""
    return bool(_lib.lib().fdh_multi_uses_rccl())


def shutdown_devices():
    _lib.check(_lib.lib().fdh_shutdown())


def debug_build_tables(code_lengths, hlit):
    """Device Huffman-table builder on one set of 320 code lengths -> (status, litlen, dist, eof)."""
    import torch
    cl = torch.as_tensor(list(code_lengths), dtype=torch.uint8).cuda()
    lit = torch.empty(4096, dtype=torch.int32, device="cuda")
    dist = torch.empty(512, dtype=torch.int32, device="cuda")
    st = torch.zeros(4, dtype=torch.int32, device="cuda")
    with _OnDevice(cl, lit, dist, st) as stream:
        _lib.check(_lib.lib().fdh_debug_build_tables(_ptr(cl), hlit, _ptr(lit), _ptr(dist), _ptr(st),
                                                    C.c_void_p(stream)))
    torch.cuda.synchronize()
    s = st.cpu().tolist()
    return s[0], lit.cpu().numpy().view("uint32"), dist.cpu().numpy().view("uint32"), tuple(s[1:])
