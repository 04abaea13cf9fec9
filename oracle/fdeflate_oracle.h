/*
 * fdeflate_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the image-rs/fdeflate algorithm for the PNG hot path
 * (zlib decode via `Decompressor::read`, ultra-fast encode via
 * `compress_to_vec_ultra_fast`).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may link or call this.  The shipped library
 * (fdeflate_amd/csrc) never includes or links anything from oracle/.
 *
 * Parity pin: the Rust reference cannot be built in this image (no rustc/cargo), so
 * the oracle is pinned against the reference's own golden vectors instead
 * (tests/test_oracle_golden.py): FIXED_LITLEN_TABLE / FIXED_DIST_TABLE
 * (src/tables.rs:142-202 via src/decompress.rs:1218-1233), the RFC-1951 Huffman
 * known-answer tests (src/huffman.rs:335-480), the three tests/NAME.zz regression
 * vectors with their expected length / Adler-32 / error (src/decompress.rs:1344-1384),
 * the 66 fuzz/corpus/inflate streams cross-checked with system zlib, the zero_length
 * and checksum tests (src/decompress.rs:1261-1325) and the ultra-fast HEADER
 * (src/compress/ultrafast.rs:82-86).
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef FDEFLATE_ORACLE_H
#define FDEFLATE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes: 0 = OK, 1 + ordinal of DecompressionError (src/decompress.rs:14-48),
 * 17 = OutputTooLarge (BoundedDecompressionError, src/decompress.rs:1090-1102). */
enum {
    FDO_OK = 0,
    FDO_BAD_ZLIB_HEADER = 1,
    FDO_INSUFFICIENT_INPUT = 2,
    FDO_INVALID_BLOCK_TYPE = 3,
    FDO_INVALID_UNCOMPRESSED_BLOCK_LENGTH = 4,
    FDO_INVALID_HLIT = 5,
    FDO_INVALID_HDIST = 6,
    FDO_INVALID_CODE_LENGTH_REPEAT = 7,
    FDO_BAD_CODE_LENGTH_HUFFMAN_TREE = 8,
    FDO_BAD_LITERAL_LENGTH_HUFFMAN_TREE = 9,
    FDO_BAD_DISTANCE_HUFFMAN_TREE = 10,
    FDO_INVALID_LITERAL_LENGTH_CODE = 11,
    FDO_INVALID_DISTANCE_CODE = 12,
    FDO_INPUT_STARTS_WITH_RUN = 13,
    FDO_DISTANCE_TOO_FAR_BACK = 14,
    FDO_WRONG_CHECKSUM = 15,
    FDO_EXTRA_INPUT = 16,
    FDO_OUTPUT_TOO_LARGE = 17
};

/* ---- Adler-32 (simd-adler32 0.3.x call sites; RFC 1950 arithmetic) ---- */
uint32_t fdo_adler32(const uint8_t *data, size_t len);
uint32_t fdo_adler32_update(uint32_t adler, const uint8_t *data, size_t len);

/* ---- huffman::build_table (src/huffman.rs:18-184) ----
 * `entries` may be NULL (n_entries = 0).  `secondary` must have room for
 * `secondary_cap` u16; *secondary_len receives the used length.  Returns 1 on
 * success, 0 if the lengths are not a complete prefix code. */
int fdo_build_table(const uint8_t *lengths, size_t n_lengths, const uint32_t *entries,
                    size_t n_entries, uint16_t *codes, uint32_t *primary_table,
                    size_t primary_len, uint16_t *secondary, size_t secondary_cap,
                    size_t *secondary_len, int is_distance_table, int double_literal);

/* CompressedBlock::build_tables (src/decompress.rs:561-606) on caller-provided
 * storage: litlen[4096], dist[512].  Returns a status code. */
int fdo_build_decode_tables(size_t hlit, const uint8_t code_lengths[320], uint32_t *litlen,
                            uint32_t *dist, uint16_t *eof_code, uint16_t *eof_mask,
                            uint8_t *eof_bits);

/* Constant tables (src/tables.rs) exposed for golden tests. */
const uint8_t *fdo_huffman_lengths(void);            /* HUFFMAN_LENGTHS[286] */
const uint16_t *fdo_huffman_codes(void);             /* HUFFMAN_CODES[286] via lib.rs:103-127 */
const uint32_t *fdo_litlen_table_entries(void);      /* LITLEN_TABLE_ENTRIES[288] */
const uint32_t *fdo_distance_table_entries(void);    /* DISTANCE_TABLE_ENTRIES[32] */
const uint8_t *fdo_ultrafast_header(void);           /* HEADER[54] */

/* ---- Decompressor (src/decompress.rs:96-556) ---- */
typedef struct fdo_decompressor fdo_decompressor;
fdo_decompressor *fdo_decompressor_new(void);
void fdo_decompressor_free(fdo_decompressor *d);
void fdo_decompressor_ignore_adler32(fdo_decompressor *d);
int fdo_decompressor_is_done(const fdo_decompressor *d);
/* Decompressor::read (src/decompress.rs:179-337).  Returns a status code. */
int fdo_decompressor_read(fdo_decompressor *d, const uint8_t *input, size_t input_len,
                          uint8_t *output, size_t output_len, size_t output_position,
                          size_t *consumed, size_t *produced);

/* decompress_to_vec_bounded (src/decompress.rs:1111-1144) into a caller buffer of
 * `maxlen` bytes (the Vec growth policy is reproduced internally).  On FDO_OK
 * *out_len is the decoded length; on FDO_OUTPUT_TOO_LARGE the buffer holds the
 * partial output (maxlen bytes).  `ignore_adler32` != 0 mirrors
 * Decompressor::ignore_adler32. `adler` (nullable) receives Adler-32 of the output. */
int fdo_decompress_bounded(const uint8_t *input, size_t input_len, uint8_t *out, size_t maxlen,
                           size_t *out_len, int ignore_adler32, uint32_t *adler);

/* test_utils::decompress_by_chunks (src/decompress/tests/test_utils.rs:47-87):
 * checksum ignored, 1 000 000-byte output buffer, <=5000 iterations.  chunk <= 0 means
 * "whole input".  Returns status, or -1 OutputTooLarge(test) / -2 TooManyIterations. */
int fdo_decompress_by_chunks(const uint8_t *input, size_t input_len, long chunk, uint8_t *out,
                             size_t out_cap, size_t *out_len);

/* ---- UltraFastCompressor (src/compress/ultrafast.rs:9-182) ---- */
size_t fdo_ultrafast_bound(size_t len);
/* compress_to_vec_ultra_fast (src/compress/mod.rs:313-317).  Returns bytes written. */
size_t fdo_compress_ultra_fast(const uint8_t *input, size_t len, uint8_t *out, size_t out_cap);

/* Compressor level 0 stored + empty-input level-1 KAT (src/compress/mod.rs:69-71,
 * 194-214, 234-268); used only to generate stored-block test streams. */
size_t fdo_compress_stored(const uint8_t *input, size_t len, uint8_t *out, size_t out_cap);

/* ---- general encoder, level 1 and RLE (src/compress/mod.rs:294-310; parse/, matchfinder/,
 * bitstream.rs): compress_to_vec / compress_to_vec_rle.  Return bytes written, 0 if out_cap is too
 * small (fdo_compress_bound(len) always suffices).  Parity pin: see the section header in the .c. */
size_t fdo_compress_bound(size_t len);
size_t fdo_compress_level1(const uint8_t *input, size_t len, uint8_t *out, size_t out_cap);
size_t fdo_compress_rle(const uint8_t *input, size_t len, uint8_t *out, size_t out_cap);
/* test instrumentation: trees shortened so far, [0] to 15 bits, [1] to 7 bits */
extern unsigned long fdo_length_limit_events[2];

/* ---- PNG scanline filters (PNG specification section 9; the png crate's step either side of the
 * codec).  filt = rows x (1 + row_bytes) with the filter-type byte first, pix = rows x row_bytes.
 * Return 0 ok, 1 filter type > 4, 2 size not a whole number of rows. */
int fdo_png_unfilter(const uint8_t *filt, size_t len, size_t row_bytes, size_t bpp, uint8_t *pix);
int fdo_png_filter(const uint8_t *pix, size_t len, size_t row_bytes, size_t bpp, const uint8_t *types, uint8_t *filt);

/* ---- batch helpers for the CPU baseline leg (one stream per task, pthreads) ---- */
void fdo_inflate_batch(const uint8_t *in, const uint64_t *in_off, uint8_t *out,
                       const uint64_t *out_off, uint32_t *out_len, uint32_t *status,
                       uint32_t *adler, uint64_t n, int ignore_adler32, int nthreads);
void fdo_deflate_ultrafast_batch(const uint8_t *in, const uint64_t *in_off, uint8_t *out,
                                 const uint64_t *out_off, uint32_t *out_len, uint64_t n,
                                 int nthreads);

/* Timed CPU baseline: threads created once, thread t decodes streams t, t+T, ... `passes` times;
 * returns the seconds between start and finish barrier (< 0: failure).  kind 0 = this oracle,
 * kind 1 = system zlib `uncompress` (second comparator). */
double fdo_timed_inflate(const uint8_t *in, const uint64_t *in_off, uint8_t *out,
                         const uint64_t *out_off, uint64_t n, int nthreads, int passes, int kind);

#ifdef __cplusplus
}
#endif
#endif
