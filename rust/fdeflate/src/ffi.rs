//! `extern "C"` declarations of include/fdeflate_hip.h -- one item per C entry point, each naming
//! the reference item it stands in for.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const FDH_SUCCESS: c_int = 0;
pub const FDH_FLAG_IGNORE_ADLER32: u32 = 0x1;
pub const FDH_OUTPUT_TOO_LARGE: u32 = 17;
pub const FDH_MODE_LEVEL1: u32 = 1;
pub const FDH_MODE_RLE: u32 = 2;

#[repr(C)]
pub struct fdh_decompressor {
    _private: [u8; 0],
}

/// `fdh_resume_point`: where a stream that ran out of input or room can be taken up again.
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct fdh_resume_point {
    pub header_bit: u32,
    pub bit: u32,
    pub out_bytes: u32,
    pub adler32: u32,
}

/// `fdh_shard_t`: the device-resident shard of one GPU for `fdh_inflate_batch_multi`.
#[repr(C)]
pub struct fdh_shard_t {
    pub input: *const u8,
    pub in_off: *const u64,
    pub out: *mut u8,
    pub out_off: *const u64,
    pub out_len: *mut u32,
    pub status: *mut u32,
    pub adler: *mut u32,
    pub n: u64,
    pub meta_all: *mut u32,
}

extern "C" {
    // decompress_to_vec_bounded per stream (src/decompress.rs:1111-1144), device pointers
    pub fn fdh_inflate_batch(input: *const u8, in_off: *const u64, out: *mut u8, out_off: *const u64,
                             out_len: *mut u32, status: *mut u32, adler: *mut u32, n: u64, flags: u32,
                             hip_stream: *mut c_void) -> c_int;
    /// `fdh_inflate_batch` that can stop and go on (16-byte resume points per stream, see the header).
    pub fn fdh_inflate_batch_resumable(input: *const u8, in_off: *const u64, out: *mut u8, out_off: *const u64,
                                       out_len: *mut u32, status: *mut u32, adler: *mut u32, n: u64, flags: u32,
                                       resume: *mut fdh_resume_point, hip_stream: *mut c_void) -> c_int;
    // compress_to_vec_ultra_fast per buffer (src/compress/mod.rs:313-317)
    pub fn fdh_deflate_ultrafast_batch(input: *const u8, in_off: *const u64, out: *mut u8, out_off: *const u64,
                                       out_len: *mut u32, n: u64, hip_stream: *mut c_void) -> c_int;
    pub fn fdh_ultrafast_bound(len: u64) -> u64;
    // compress_to_vec_with_level(.., 0) per buffer (src/compress/mod.rs:299-303)
    pub fn fdh_deflate_stored_batch(input: *const u8, in_off: *const u64, out: *mut u8, out_off: *const u64,
                                    out_len: *mut u32, n: u64, hip_stream: *mut c_void) -> c_int;
    pub fn fdh_stored_size(len: u64) -> u64;
    // compress_to_vec (level 1) / compress_to_vec_rle per buffer (src/compress/mod.rs:294-310)
    pub fn fdh_deflate_general_batch(input: *const u8, in_off: *const u64, out: *mut u8, out_off: *const u64,
                                     out_len: *mut u32, n: u64, mode: u32, hip_stream: *mut c_void) -> c_int;
    pub fn fdh_compress_bound(len: u64) -> u64;

    // Decompressor (src/decompress.rs:96-156, 179-342)
    pub fn fdh_decompressor_new() -> *mut fdh_decompressor;
    pub fn fdh_decompressor_free(d: *mut fdh_decompressor);
    pub fn fdh_decompressor_ignore_adler32(d: *mut fdh_decompressor);
    pub fn fdh_decompressor_is_done(d: *const fdh_decompressor) -> c_int;
    /// Introspection (not part of the reference API): decode attempts made so far.
    pub fn fdh_decompressor_attempts(d: *const fdh_decompressor) -> u64;
    /// Introspection (not part of the reference API): output bytes decoded by all attempts together.
    pub fn fdh_decompressor_decoded_bytes(d: *const fdh_decompressor) -> u64;
    /// Introspection (not part of the reference API): the most device memory the object's buffers have held, in bytes.
    pub fn fdh_decompressor_device_bytes(d: *const fdh_decompressor) -> u64;
    pub fn fdh_decompressor_read(d: *mut fdh_decompressor, input: *const u8, input_len: usize, output: *mut u8,
                                 output_len: usize, output_position: usize, consumed: *mut usize,
                                 produced: *mut usize, stream_status: *mut u32) -> c_int;

    // host-memory conveniences (src/lib.rs:29-36 names)
    pub fn fdh_decompress_to_vec(input: *const u8, len: usize, out: *mut *mut u8, out_len: *mut usize,
                                 status: *mut u32) -> c_int;
    pub fn fdh_decompress_to_vec_bounded(input: *const u8, len: usize, maxlen: usize, out: *mut *mut u8,
                                         out_len: *mut usize, status: *mut u32) -> c_int;
    pub fn fdh_compress_to_vec_ultra_fast(input: *const u8, len: usize, out: *mut *mut u8, out_len: *mut usize) -> c_int;
    pub fn fdh_compress_to_vec_stored(input: *const u8, len: usize, out: *mut *mut u8, out_len: *mut usize) -> c_int;
    pub fn fdh_compress_to_vec(input: *const u8, len: usize, out: *mut *mut u8, out_len: *mut usize) -> c_int;
    pub fn fdh_compress_to_vec_rle(input: *const u8, len: usize, out: *mut *mut u8, out_len: *mut usize) -> c_int;
    pub fn fdh_free(p: *mut c_void);

    // several GPUs of a node from one process
    pub fn fdh_init(device_mask: u64) -> c_int;
    pub fn fdh_shutdown() -> c_int;
    pub fn fdh_multi_device_count() -> c_int;
    pub fn fdh_inflate_batch_multi(shards: *const fdh_shard_t, n_shards: u32, flags: u32, meta_stride: u64) -> c_int;
    pub fn fdh_multi_uses_rccl() -> c_int;

    // the steps either side of the codec in the PNG pipeline (README.md:11 of the reference): scanline filters
    pub fn fdh_png_unfilter_batch(filt: *const u8, filt_off: *const u64, pix: *mut u8, pix_off: *const u64,
                                  png_status: *mut u32, n: u64, row_bytes: u32, bpp: u32, hip_stream: *mut c_void) -> c_int;
    pub fn fdh_png_filter_batch(pix: *const u8, pix_off: *const u64, types: *const u8, types_off: *const u64,
                                filt: *mut u8, filt_off: *const u64, png_status: *mut u32, n: u64, row_bytes: u32,
                                bpp: u32, hip_stream: *mut c_void) -> c_int;
    pub fn fdh_png_filter_deflate_ultrafast_batch(pix: *const u8, pix_off: *const u64, types: *const u8,
                                                  types_off: *const u64, out: *mut u8, out_off: *const u64,
                                                  out_len: *mut u32, png_status: *mut u32, n: u64, row_bytes: u32,
                                                  bpp: u32, hip_stream: *mut c_void) -> c_int;
    pub fn fdh_inflate_png_batch(input: *const u8, in_off: *const u64, filt: *mut u8, filt_off: *const u64,
                                 out_len: *mut u32, status: *mut u32, adler: *mut u32, pix: *mut u8,
                                 pix_off: *const u64, png_status: *mut u32, n: u64, flags: u32, row_bytes: u32,
                                 bpp: u32, hip_stream: *mut c_void) -> c_int;

    pub fn fdh_last_error() -> *const c_char;
    pub fn fdh_device_count() -> c_int;
}
