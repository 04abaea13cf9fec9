"""fdeflate_amd -- MI355X-native batched DEFLATE codec behind fdeflate's PNG-path API.

Public surface mirrors image-rs/fdeflate (reference src/lib.rs:29-36) for the hot path:
decompress_to_vec, decompress_to_vec_bounded, compress_to_vec_ultra_fast, DecompressionError,
plus the batched device entry points.  See DESIGN.md / INTEGRATION.md.
"""
from .api import (Decompressor, DecompressionError, OutputTooLarge, STATUS_NAMES, FLAG_IGNORE_ADLER32,
                  FLAG_SERIAL_ONLY, FLAG_GENERAL_ONLY, FLAG_NO_RECHECK, compress_to_vec_ultra_fast, debug_build_tables,
                  decompress_to_vec, decompress_to_vec_bounded, deflate_ultrafast_batch,
                  inflate_batch, inflate_batch_resumable, ultrafast_bound, compress_to_vec_stored, deflate_stored_batch,
                  stored_size, compress_to_vec, compress_to_vec_rle, compress_bound, deflate_general_batch,
                  MODE_LEVEL1, MODE_RLE, inflate_batch_multi, init_devices, shutdown_devices, multi_uses_rccl,
                  png_unfilter_batch, png_filter_batch, inflate_png_batch, png_filter_deflate_ultrafast_batch)

__all__ = [
    "Decompressor", "DecompressionError", "OutputTooLarge", "STATUS_NAMES", "FLAG_IGNORE_ADLER32",
    "FLAG_SERIAL_ONLY", "FLAG_GENERAL_ONLY", "FLAG_NO_RECHECK", "compress_to_vec_ultra_fast", "debug_build_tables", "decompress_to_vec",
    "decompress_to_vec_bounded", "deflate_ultrafast_batch", "inflate_batch", "inflate_batch_resumable", "ultrafast_bound",
    "compress_to_vec_stored", "deflate_stored_batch", "stored_size", "compress_to_vec", "compress_to_vec_rle",
    "compress_bound", "deflate_general_batch", "MODE_LEVEL1", "MODE_RLE", "inflate_batch_multi", "init_devices",
    "shutdown_devices", "multi_uses_rccl", "png_unfilter_batch", "png_filter_batch", "inflate_png_batch", "png_filter_deflate_ultrafast_batch",
]
