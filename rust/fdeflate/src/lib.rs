//! fdeflate's public API for the PNG path (reference `src/lib.rs:29-36`) over `libfdeflate_hip.so`.
//!
//! Same items, same signatures, same error values as the reference crate; every byte of decoding
//! and encoding happens on the GPU behind the C ABI of `include/fdeflate_hip.h`.  What the
//! reference has and this shim does not: `Compressor<W>` for levels 2-9 (the LZ77 match search of
//! those levels is outside the hot path, see DESIGN.md "out of scope") and the hidden
//! `compute_code_lengths` helper.
//!
//! Source only: the build container has no Rust toolchain, so this crate has never been compiled
//! there.  The same C entry points are exercised through ctypes by `tests/`.
mod ffi;

use std::io::{self, Write};
use std::os::raw::c_void;

pub use batch::{inflate_batch_device, inflate_batch_multi_device, Shard};

/// An error encountered while decompressing a deflate stream (reference `src/decompress.rs:14-48`;
/// the C ABI's per-stream status is `1 + ordinal`).
#[derive(Debug, PartialEq, Clone)]
pub enum DecompressionError {
    BadZlibHeader,
    InsufficientInput,
    InvalidBlockType,
    InvalidUncompressedBlockLength,
    InvalidHlit,
    InvalidHdist,
    InvalidCodeLengthRepeat,
    BadCodeLengthHuffmanTree,
    BadLiteralLengthHuffmanTree,
    BadDistanceHuffmanTree,
    InvalidLiteralLengthCode,
    InvalidDistanceCode,
    InputStartsWithRun,
    DistanceTooFarBack,
    WrongChecksum,
    ExtraInput,
}

/// Reference `src/decompress.rs:1090-1102`.
#[derive(Debug, PartialEq)]
pub enum BoundedDecompressionError {
    DecompressionError { inner: DecompressionError },
    OutputTooLarge { partial_output: Vec<u8> },
}
impl From<DecompressionError> for BoundedDecompressionError {
    fn from(inner: DecompressionError) -> Self {
        BoundedDecompressionError::DecompressionError { inner }
    }
}

fn to_error(status: u32) -> DecompressionError {
    use DecompressionError::*;
    match status {
        1 => BadZlibHeader,
        2 => InsufficientInput,
        3 => InvalidBlockType,
        4 => InvalidUncompressedBlockLength,
        5 => InvalidHlit,
        6 => InvalidHdist,
        7 => InvalidCodeLengthRepeat,
        8 => BadCodeLengthHuffmanTree,
        9 => BadLiteralLengthHuffmanTree,
        10 => BadDistanceHuffmanTree,
        11 => InvalidLiteralLengthCode,
        12 => InvalidDistanceCode,
        13 => InputStartsWithRun,
        14 => DistanceTooFarBack,
        15 => WrongChecksum,
        16 => ExtraInput,
        other => panic!("fdeflate_hip: unknown stream status {other}"),
    }
}

/// Infrastructure failures (no GPU, HIP error, out of memory) have no counterpart in the
/// reference's signatures: like an allocation failure there, they panic with the library's message.
fn check(rc: std::os::raw::c_int) {
    if rc != ffi::FDH_SUCCESS {
        let msg = unsafe { std::ffi::CStr::from_ptr(ffi::fdh_last_error()) };
        panic!("fdeflate_hip error {rc}: {}", msg.to_string_lossy());
    }
}

/// Copies a malloc'd result of the C side into a Vec and releases it.
unsafe fn take(p: *mut u8, n: usize) -> Vec<u8> {
    let v = if n == 0 { Vec::new() } else { std::slice::from_raw_parts(p, n).to_vec() };
    if !p.is_null() {
        ffi::fdh_free(p as *mut c_void);
    }
    v
}

/// Decompressor for arbitrary zlib streams (reference `src/decompress.rs:96-342`).
pub struct Decompressor {
    raw: *mut ffi::fdh_decompressor,
}
// One instance per stream, no interior sharing (as the reference's, which is Send by construction).
unsafe impl Send for Decompressor {}

impl Default for Decompressor {
    fn default() -> Self {
        Self::new()
    }
}

impl Decompressor {
    /// Create a new decompressor (`src/decompress.rs:123`).
    pub fn new() -> Self {
        let raw = unsafe { ffi::fdh_decompressor_new() };
        assert!(!raw.is_null(), "fdh_decompressor_new: out of memory");
        Self { raw }
    }

    /// Ignore the checksum at the end of the stream (`src/decompress.rs:154`).
    pub fn ignore_adler32(&mut self) {
        unsafe { ffi::fdh_decompressor_ignore_adler32(self.raw) }
    }

    /// Decompresses a chunk of data (`src/decompress.rs:179-337`): returns the number of bytes read
    /// from `input` and written to `output` starting at `output_position`.  When it returns `Ok` the
    /// input is fully consumed, or the output is full but there are more bytes, or the stream is
    /// complete.  Panics if `output_position` is out of bounds, as the reference does.
    pub fn read(
        &mut self,
        input: &[u8],
        output: &mut [u8],
        output_position: usize,
    ) -> Result<(usize, usize), DecompressionError> {
        assert!(output_position <= output.len());
        let (mut consumed, mut produced, mut status) = (0usize, 0usize, 0u32);
        let rc = unsafe {
            ffi::fdh_decompressor_read(
                self.raw,
                input.as_ptr(),
                input.len(),
                output.as_mut_ptr(),
                output.len(),
                output_position,
                &mut consumed,
                &mut produced,
                &mut status,
            )
        };
        check(rc);
        if status != 0 {
            return Err(to_error(status));
        }
        Ok((consumed, produced))
    }

    /// Returns true if the decompressor has finished decompressing the input (`src/decompress.rs:340`).
    pub fn is_done(&self) -> bool {
        unsafe { ffi::fdh_decompressor_is_done(self.raw) != 0 }
    }
}

impl Drop for Decompressor {
    fn drop(&mut self) {
        unsafe { ffi::fdh_decompressor_free(self.raw) }
    }
}

/// Decompress the given data (`src/decompress.rs:1079`).
pub fn decompress_to_vec(input: &[u8]) -> Result<Vec<u8>, DecompressionError> {
    let (mut p, mut n, mut status) = (std::ptr::null_mut(), 0usize, 0u32);
    check(unsafe { ffi::fdh_decompress_to_vec(input.as_ptr(), input.len(), &mut p, &mut n, &mut status) });
    let v = unsafe { take(p, n) };
    match status {
        0 => Ok(v),
        // the C ABI decodes into slots of at most 4 GiB - 1 bytes: an output that does not fit is an
        // infrastructure limit of this drop-in (the reference would go on growing the Vec), reported
        // like an allocation failure there -- a panic with a message, not a bogus `DecompressionError`
        ffi::FDH_OUTPUT_TOO_LARGE => panic!("fdeflate_hip: decompressed output exceeds the 4 GiB slot limit of the C ABI"),
        s => Err(to_error(s)),
    }
}

/// Decompress the given data, returning an error if the output is larger than `maxlen` bytes
/// (`src/decompress.rs:1111`).
pub fn decompress_to_vec_bounded(input: &[u8], maxlen: usize) -> Result<Vec<u8>, BoundedDecompressionError> {
    let (mut p, mut n, mut status) = (std::ptr::null_mut(), 0usize, 0u32);
    check(unsafe {
        ffi::fdh_decompress_to_vec_bounded(input.as_ptr(), input.len(), maxlen, &mut p, &mut n, &mut status)
    });
    let v = unsafe { take(p, n) };
    match status {
        0 => Ok(v),
        ffi::FDH_OUTPUT_TOO_LARGE => Err(BoundedDecompressionError::OutputTooLarge { partial_output: v }),
        s => Err(to_error(s).into()),
    }
}

type CompressFn = unsafe extern "C" fn(*const u8, usize, *mut *mut u8, *mut usize) -> std::os::raw::c_int;

fn compress_with(f: CompressFn, input: &[u8]) -> Vec<u8> {
    let (mut p, mut n) = (std::ptr::null_mut(), 0usize);
    check(unsafe { f(input.as_ptr(), input.len(), &mut p, &mut n) });
    unsafe { take(p, n) }
}

/// Compresses the given data (`src/compress/mod.rs:294`: level 1 in this snapshot of the crate).
pub fn compress_to_vec(input: &[u8]) -> Vec<u8> {
    compress_with(ffi::fdh_compress_to_vec, input)
}

/// Compresses the given data with a specific compression level (`src/compress/mod.rs:299`).
/// Levels 0 and 1 run on the GPU; the LZ77 searches of levels 2-9 are not part of this codec.
pub fn compress_to_vec_with_level(input: &[u8], level: u8) -> Vec<u8> {
    match level {
        0 => compress_with(ffi::fdh_compress_to_vec_stored, input),
        1 => compress_with(ffi::fdh_compress_to_vec, input),
        _ => panic!("fdeflate (hip): compression level {level} is not provided by the GPU codec"),
    }
}

/// Compresses the given data using only RLE matches (`src/compress/mod.rs:306`).
pub fn compress_to_vec_rle(input: &[u8]) -> Vec<u8> {
    compress_with(ffi::fdh_compress_to_vec_rle, input)
}

/// Compresses the given data using the ultra fast compression method (`src/compress/mod.rs:313`).
pub fn compress_to_vec_ultra_fast(input: &[u8]) -> Vec<u8> {
    compress_with(ffi::fdh_compress_to_vec_ultra_fast, input)
}

/// Compressor that uses the ultra-fast mode (`src/compress/ultrafast.rs:9-182`).  The reference
/// streams 8-byte chunks straight to the writer; the GPU encoder works on whole buffers, so the
/// data is collected and encoded when `finish` is called -- the bytes written are the same.
pub struct UltraFastCompressor<W: Write> {
    writer: W,
    data: Vec<u8>,
}

impl<W: Write> UltraFastCompressor<W> {
    /// Create a new Compressor (`src/compress/ultrafast.rs:70`).
    pub fn new(writer: W) -> io::Result<Self> {
        Ok(Self { writer, data: Vec::new() })
    }

    /// Write data to the compressor (`src/compress/ultrafast.rs:94`).
    pub fn write_data(&mut self, data: &[u8]) -> io::Result<()> {
        self.data.extend_from_slice(data);
        Ok(())
    }

    /// Write the remainder of the stream and return the inner writer (`src/compress/ultrafast.rs:170`).
    pub fn finish(mut self) -> io::Result<W> {
        let out = compress_to_vec_ultra_fast(&self.data);
        self.writer.write_all(&out)?;
        Ok(self.writer)
    }
}

/// The batched entry points the GPU exists for: streams already resident in HBM.
pub mod batch {
    use super::{check, ffi};
    use std::os::raw::c_void;

    pub use ffi::fdh_shard_t as Shard;

    /// `n` independent zlib streams, one wavefront each; pointers are device pointers, the call
    /// enqueues on `stream` (a `hipStream_t`) and returns.  Per-stream results: `status[i]` 0 = Ok,
    /// otherwise `1 + ordinal` of `DecompressionError`, 17 = `OutputTooLarge`.
    ///
    /// # Safety
    /// All pointers must be valid device allocations of the sizes `include/fdeflate_hip.h` states.
    #[allow(clippy::too_many_arguments)]
    pub unsafe fn inflate_batch_device(
        input: *const u8,
        in_off: *const u64,
        out: *mut u8,
        out_off: *const u64,
        out_len: *mut u32,
        status: *mut u32,
        adler: *mut u32,
        n: u64,
        ignore_adler32: bool,
        stream: *mut c_void,
    ) {
        let flags = if ignore_adler32 { ffi::FDH_FLAG_IGNORE_ADLER32 } else { 0 };
        check(ffi::fdh_inflate_batch(input, in_off, out, out_off, out_len, status, adler, n, flags, stream));
    }

    /// One shard per GPU selected by `fdh_init`; results all-gathered over RCCL when `meta_all` is set.
    ///
    /// # Safety
    /// Shard `i` must hold device pointers on the `i`-th selected device.
    pub unsafe fn inflate_batch_multi_device(shards: &[Shard], ignore_adler32: bool, meta_stride: u64) {
        let flags = if ignore_adler32 { ffi::FDH_FLAG_IGNORE_ADLER32 } else { 0 };
        check(ffi::fdh_inflate_batch_multi(shards.as_ptr(), shards.len() as u32, flags, meta_stride));
    }
}

#[cfg(test)]
mod tests {
    //! The reference's own unit tests that only need the public API (`src/decompress.rs:1235-1325`).
    use super::*;

    #[test]
    fn round_trip_level1() {
        let data = b"Hello world!";
        assert_eq!(decompress_to_vec(&compress_to_vec(data)).unwrap(), data);
    }

    #[test]
    fn empty_input_kat() {
        assert_eq!(compress_to_vec(b""), [0x78, 0x01, 0x03, 0x00, 0x00, 0x00, 0x00, 0x01]);
    }

    #[test]
    fn checksum_after_eof() {
        let input = b"Hello world!";
        let compressed = compress_to_vec(input);
        let mut d = Decompressor::new();
        let mut out = vec![0; 1024];
        let (c, p) = d.read(&compressed[..compressed.len() - 1], &mut out, 0).unwrap();
        assert_eq!((c, p), (compressed.len() - 1, input.len()));
        let (c2, p2) = d.read(&compressed[c..], &mut out[..p], p).unwrap();
        assert!(d.is_done());
        assert_eq!((c2, p2), (1, 0));
    }

    #[test]
    fn wrong_checksum_and_ignore() {
        let mut compressed = compress_to_vec(b"Hello world!");
        let last = compressed.len() - 1;
        compressed[last] = compressed[last].wrapping_add(1);
        assert_eq!(decompress_to_vec(&compressed), Err(DecompressionError::WrongChecksum));
        let mut d = Decompressor::new();
        d.ignore_adler32();
        let mut out = vec![0; 1024];
        let n = d.read(&compressed, &mut out, 0).unwrap().1;
        assert_eq!(&out[..n], b"Hello world!");
    }
}
