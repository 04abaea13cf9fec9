/*
 * fdeflate_hip.h -- C ABI of the MI355X-native batched DEFLATE codec (PNG path).
 *
 * Drop-in boundary for image-rs/fdeflate's public API on the PNG hot path
 * (reference: /root/reference/src/lib.rs:29-36).  The reference has no FFI of its own;
 * each entry point below names the Rust item it replaces.  A Rust shim crate binds these
 * with `extern "C"` (see INTEGRATION.md for the exact stub).
 *
 * Conventions
 *   - plain pointers and sizes, no torch / C++ types;
 *   - `*_batch` entry points take DEVICE pointers (HBM resident) and a hipStream_t passed as
 *     `void*` (NULL = the null stream); they enqueue work and return without synchronising;
 *   - function return value = infrastructure status (0 ok, non-zero = HIP / argument failure,
 *     message via fdh_last_error()); per-stream results are in `status[]`;
 *   - the library never falls back to a CPU path: without a usable GPU every entry point that
 *     does work returns FDH_ERR_NO_DEVICE.
 */
#ifndef FDEFLATE_HIP_H
#define FDEFLATE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDH_VERSION 0x000100u

/* ---- library-level return codes ---------------------------------------------------- */
enum {
    FDH_SUCCESS = 0,
    FDH_ERR_INVALID_ARGUMENT = 1,
    FDH_ERR_NO_DEVICE = 2,
    FDH_ERR_HIP = 3,
    FDH_ERR_OUT_OF_MEMORY = 4
};

/* ---- per-stream status --------------------------------------------------------------
 * 0 = Ok, otherwise 1 + ordinal of `DecompressionError` (src/decompress.rs:14-48), plus one
 * ABI-only code for `BoundedDecompressionError::OutputTooLarge` (src/decompress.rs:1097-1101). */
enum {
    FDH_STREAM_OK = 0,
    FDH_BAD_ZLIB_HEADER = 1,
    FDH_INSUFFICIENT_INPUT = 2,
    FDH_INVALID_BLOCK_TYPE = 3,
    FDH_INVALID_UNCOMPRESSED_BLOCK_LENGTH = 4,
    FDH_INVALID_HLIT = 5,
    FDH_INVALID_HDIST = 6,
    FDH_INVALID_CODE_LENGTH_REPEAT = 7,
    FDH_BAD_CODE_LENGTH_HUFFMAN_TREE = 8,
    FDH_BAD_LITERAL_LENGTH_HUFFMAN_TREE = 9,
    FDH_BAD_DISTANCE_HUFFMAN_TREE = 10,
    FDH_INVALID_LITERAL_LENGTH_CODE = 11,
    FDH_INVALID_DISTANCE_CODE = 12,
    FDH_INPUT_STARTS_WITH_RUN = 13,
    FDH_DISTANCE_TOO_FAR_BACK = 14,
    FDH_WRONG_CHECKSUM = 15,
    FDH_EXTRA_INPUT = 16,
    FDH_OUTPUT_TOO_LARGE = 17
};

/* ---- flags -------------------------------------------------------------------------- */
#define FDH_FLAG_IGNORE_ADLER32 0x1u /* Decompressor::ignore_adler32, src/decompress.rs:154 */
#define FDH_FLAG_SERIAL_ONLY    0x2u /* debug/A-B: force the per-symbol wave-serial decoder */
#define FDH_FLAG_GENERAL_ONLY   0x4u /* debug/A-B: skip the shared-table kernel */
#define FDH_FLAG_NO_RECHECK     0x8u /* tests: do not re-derive non-Ok results serially */
#define FDH_FLAG_FORCE_LANES    0x10u /* tests/A-B: stream-per-lane kernel even for small batches */
#define FDH_FLAG_NO_LANES       0x20u /* tests/A-B: never use the stream-per-lane kernel */
#define FDH_FLAG_FIRST_ONLY     0x40u /* debug: run only the first kernel of the pipeline */
#define FDH_FLAG_NO_SEGMENTS    0x80u /* tests/A-B: skip the segment-parallel kernel */
#define FDH_FLAG_NO_FAST_GENERAL 0x200u /* tests/A-B: skip the small-table general kernel */
#define FDH_FLAG_NO_INTERVALS   0x400u /* tests/A-B: skip the interval kernel (segment kernel first, as in round 2) */
#define FDH_FLAG_INTERVALS_ONLY 0x800u /* debug: run only the interval kernel (what it leaves stays PENDING) */
#define FDH_FLAG_NO_LZ          0x1000u /* tests/A-B: skip the LZ-window kernel (general streams go to the tile decoders) */
#define FDH_FLAG_LZ_ONLY        0x2000u /* debug: nothing behind the LZ-window kernel runs (what it leaves stays PENDING) */
#define FDH_FLAG_RESUME_IN      0x8000u /* fdh_inflate_batch_resumable: `resume` also says where to take each stream up */
#define FDH_FLAG_NO_CHECKPOINTS 0x4000u /* tests/A-B: the exact serial decoder re-derives a doubtful result from the stream's first byte, not from the last check point */
#define FDH_FLAG_NO_LANDING     0x10000u /* tests/A-B: skip the landing decoder (the interval kernel counts for itself, as in rounds 3-4) */
#define FDH_FLAG_LANDING_ONLY   0x20000u /* debug: run only the landing decoder (what it leaves stays PENDING) */
#define FDH_FLAG_NO_LEAN_WRITE   0x80000u /* tests/A-B: the landing decoder always takes the interval decoder's general writing pass */
#define FDH_FLAG_NO_OVERLAP      0x100000u /* tests/A-B: the LZ-window kernel runs behind the canonical kernels, not beside them */
#define FDH_FLAG_TAIL_LONG       0x200000u /* tests/A-B: behind the landing decoder always the five kernels of rounds 3-5 (interval, segment, tile decoders, two exact kernels) */
#define FDH_FLAG_TAIL_SHORT      0x400000u /* tests/A-B: behind the landing decoder always the exact kernel alone (the library chooses by what recent calls left over) */
#define FDH_FLAG_ORDER_ONCE      0x800000u /* tests/A-B: stream_order_kernel lists the streams without the ultra-fast prefix in one launch, in no order */
#define FDH_FLAG_ORDER_TWICE     0x1000000u /* tests/A-B: ... always in two (the long ones first), as in rounds 4-5 (the library chooses by how many recent calls had) */
#define FDH_FLAG_LANDING_COUNT_ONLY 0x40000u /* debug: the landing decoder counts and leaves every stream PENDING */
#define FDH_FLAG_SPANS          0x100u /* experimental: segment-parallel "span" decoder inside the 12-bit general kernel */

/*
 * fdh_inflate_batch -- one-shot decode of `n` independent zlib streams, one wavefront each.
 *
 * Replaces, per stream i: `decompress_to_vec_bounded(&in[in_off[i]..in_off[i+1]],
 * out_off[i+1]-out_off[i])` (src/decompress.rs:1111-1144), i.e. a `Decompressor::new()`
 * (src/decompress.rs:123) driven by `Decompressor::read` (src/decompress.rs:179-337) until
 * `is_done()` (src/decompress.rs:340), with the slot capacity as `maxlen`.
 *
 *   in, in_off[n+1]    packed compressed bytes; stream i = in[in_off[i] .. in_off[i+1])
 *   out, out_off[n+1]  output slots; capacity of stream i = out_off[i+1] - out_off[i]
 *                      (< 4 GiB); bytes outside [out_off[i], out_off[i+1]) are never written
 *   out_len[n]         decoded length; on FDH_OUTPUT_TOO_LARGE the capacity (the slot holds the
 *                      partial output like `partial_output`); on FDH_INSUFFICIENT_INPUT the bytes
 *                      `Decompressor::read` had produced when the input ran out (they are in the
 *                      slot); unspecified for other errors
 *   status[n]          per-stream status (above)
 *   adler[n]           Adler-32 of the decoded bytes (nullable): of all out_len[i] bytes for FDH_OK,
 *                      FDH_WRONG_CHECKSUM and FDH_OUTPUT_TOO_LARGE; for any other status the value is that of a
 *                      prefix of them and not specified further (a decoder that takes back the first literal of a
 *                      cut-off pair -- src/decompress.rs:852 -- keeps the sum it had reached)
 *   flags              FDH_FLAG_*
 * All pointers are device pointers.  Truncated input reports FDH_INSUFFICIENT_INPUT exactly as
 * the one-shot wrapper does (src/decompress.rs:1135-1136); bytes after the Adler-32 trailer are
 * ignored (src/decompress.rs:185-187).
 */
int fdh_inflate_batch(const uint8_t *in, const uint64_t *in_off, uint8_t *out,
                      const uint64_t *out_off, uint32_t *out_len, uint32_t *status,
                      uint32_t *adler, uint64_t n, uint32_t flags, void *hip_stream);

/*
 * fdh_inflate_batch_resumable -- fdh_inflate_batch that can stop and go on: the device-side counterpart of
 * the reference's resumable `Decompressor` (State / BitBuffer / QueuedOutput, src/decompress.rs:84-121), for
 * callers that get a stream's input or its output room in pieces (fdh_decompressor_read is built on it).
 *
 *   resume[n]   (device) per stream, 16 bytes.  OUT: for a stream that ended FDH_INSUFFICIENT_INPUT or
 *               FDH_OUTPUT_TOO_LARGE, a place inside the stream from which decoding can go on later -- a bit
 *               position at the start of one of the reference's decoding steps, the header of the block it
 *               lies in, the number of output bytes in front of it and their Adler-32 -- or all zero (go on
 *               from the first byte).  All zero for every other status.
 *               IN, with FDH_FLAG_RESUME_IN: where to take each stream up in THIS call (all zero: at its first
 *               byte).  The stream's input must start with the same bytes as in the call that produced the
 *               record (more may have arrived behind them), and its output slot must start at the same
 *               place in the caller's data: it holds the `out_bytes` decoded so far (the LZ77 history) and
 *               may have grown.  With that flag the LZ-window kernel goes on from the record and the
 *               12-bit tile / serial decoders do the rest (the segment-parallel kernels for ultra-fast
 *               streams start at a stream's first byte and do not run).
 * Status, length and Adler-32 of a stream that was stopped and taken up again -- any number of times, at any
 * split of input and output -- equal those of one fdh_inflate_batch call on the whole of it.
 */
typedef struct fdh_resume_point {
  uint32_t header_bit; /* stream bit of the block header, | step state << 30; 0: no resume point */
  uint32_t bit;        /* stream bit to go on from */
  uint32_t out_bytes;  /* output bytes in front of it */
  uint32_t adler32;    /* their Adler-32 */
} fdh_resume_point;
int fdh_inflate_batch_resumable(const uint8_t *in, const uint64_t *in_off, uint8_t *out,
                                const uint64_t *out_off, uint32_t *out_len, uint32_t *status,
                                uint32_t *adler, uint64_t n, uint32_t flags,
                                fdh_resume_point *resume, void *hip_stream);

/*
 * fdh_deflate_ultrafast_batch -- `compress_to_vec_ultra_fast` (src/compress/mod.rs:313-317,
 * UltraFastCompressor src/compress/ultrafast.rs:9-182) of `n` buffers, one wavefront each,
 * bit-exact with the reference's byte stream.
 *   in, in_off[n+1]    raw buffers
 *   out, out_off[n+1]  output slots, capacity >= fdh_ultrafast_bound(len_i) each
 *   out_len[n]         compressed length; 0xFFFFFFFF if the slot was too small (nothing valid)
 */
int fdh_deflate_ultrafast_batch(const uint8_t *in, const uint64_t *in_off, uint8_t *out,
                                const uint64_t *out_off, uint32_t *out_len, uint64_t n,
                                void *hip_stream);

/* Worst-case size of an ultra-fast stream: 53 header bytes + ceil((5 + 12*len + 12)/8) + 4. */
uint64_t fdh_ultrafast_bound(uint64_t len);

/*
 * fdh_deflate_stored_batch -- level 0: `compress_to_vec_with_level(input, 0)`
 * (src/compress/mod.rs:299-303; `Compressor::new(.., 0, true)` :69-71, stored blocks :234-268,
 * finish :194-214), bit-exact: header 78 01, stored blocks of <= 65535 bytes, an empty fixed block
 * when the length is a multiple of 65535 (incl. 0), Adler-32.  Same argument convention as
 * fdh_deflate_ultrafast_batch; slots of at least fdh_stored_size(len_i) bytes.
 */
int fdh_deflate_stored_batch(const uint8_t *in, const uint64_t *in_off, uint8_t *out,
                             const uint64_t *out_off, uint32_t *out_len, uint64_t n,
                             void *hip_stream);
/* Exact size of the level-0 stream of a `len`-byte buffer. */
uint64_t fdh_stored_size(uint64_t len);

/*
 * fdh_deflate_general_batch -- the general encoder on `n` buffers, bit-exact (a parser kernel, one
 * stream per lane, records the back-references and block ends; a block-writer kernel, one stream
 * per wavefront, builds the Huffman codes and emits):
 *   FDH_MODE_LEVEL1  `compress_to_vec(input)` = `compress_to_vec_with_level(input, 1)`
 *                    (src/compress/mod.rs:294-303; Compressor::new(.., 1, true) :69-101 =
 *                    GreedyParser src/compress/parse/greedy.rs + HashTableMatchFinder
 *                    src/compress/matchfinder/hashtable.rs, dynamic blocks src/compress/bitstream.rs)
 *   FDH_MODE_RLE     `compress_to_vec_rle(input)` (src/compress/mod.rs:306-310; Compressor::new_rle
 *                    :107-123 = RleParser src/compress/parse/rle.rs)
 * Same argument convention as fdh_deflate_ultrafast_batch; slots of at least fdh_compress_bound(len_i)
 * bytes; out_len[i] = 0xFFFFFFFF if a slot was too small or the buffer exceeds 1 GiB.  The call
 * uses a per-device workspace (one 256 KiB hash table per resident stream at level 1, at most
 * 16 GiB; 8 bytes per 4 input bytes for the back-reference records), reads in_off[0] and in_off[n]
 * back to size it, and returns after the kernels have finished.
 */
#define FDH_MODE_LEVEL1 1u
#define FDH_MODE_RLE 2u
int fdh_deflate_general_batch(const uint8_t *in, const uint64_t *in_off, uint8_t *out,
                              const uint64_t *out_off, uint32_t *out_len, uint64_t n, uint32_t mode,
                              void *hip_stream);
/* Slot size that always suffices for the general encoder: len + len / 2 + 1024. */
uint64_t fdh_compress_bound(uint64_t len);

/* ---- PNG scanline filters: the steps either side of the codec in the PNG pipeline --------
 * Not in the fdeflate crate (its reverse dependency image-rs/image-png does them, reference
 * README.md:11); the algorithm is the PNG specification's (W3C / ISO/IEC 15948, 9.2 and 9.4):
 * filter types 0 None, 1 Sub, 2 Up, 3 Average, 4 Paeth, `bpp` bytes per pixel (1, 2, 3, 4, 6, 8).
 * One image per lane.  A "filtered" image is rows x (1 + row_bytes) bytes, the type byte first (what
 * the zlib stream of an IDAT holds); a "pixel" image is rows x row_bytes.  rows_i = size_i / row size.
 *   fdh_png_unfilter_batch  reconstruction: filt -> pix
 *   fdh_png_filter_batch    filtering with the given per-row types: pix -> filt (types_off[n+1]
 *                           into `types`, one byte per row)
 *   fdh_inflate_png_batch   fdh_inflate_batch into `filt` (the slots must be the exact image sizes)
 *                           followed, on the same stream, by the reconstruction into `pix` of every
 *                           stream that decoded to exactly the bytes of its slot
 * png_status[i]: 0 ok, 1 a filter type > 4, 2 sizes do not fit (fdh_inflate_png_batch: also a stream
 * that ended before its slot was full -- short IDAT data), 3 skipped (stream did not decode). */
int fdh_png_unfilter_batch(const uint8_t *filt, const uint64_t *filt_off, uint8_t *pix,
                           const uint64_t *pix_off, uint32_t *png_status, uint64_t n,
                           uint32_t row_bytes, uint32_t bpp, void *hip_stream);
int fdh_png_filter_batch(const uint8_t *pix, const uint64_t *pix_off, const uint8_t *types,
                         const uint64_t *types_off, uint8_t *filt, const uint64_t *filt_off,
                         uint32_t *png_status, uint64_t n, uint32_t row_bytes, uint32_t bpp,
                         void *hip_stream);
/* Filtering fused into the ultra-fast encoder: pixel rows in (`pix`, rows_i x row_bytes), one filter
 * type per row in `types`, out the zlib stream compress_to_vec_ultra_fast(filtered image) -- what an
 * IDAT holds.  The filtered bytes exist only in registers (no intermediate buffer).  Slots of at
 * least fdh_ultrafast_bound(rows_i * (row_bytes + 1)) bytes; out_len[i] = 0 where png_status[i] != 0. */
int fdh_png_filter_deflate_ultrafast_batch(const uint8_t *pix, const uint64_t *pix_off,
                                           const uint8_t *types, const uint64_t *types_off,
                                           uint8_t *out, const uint64_t *out_off, uint32_t *out_len,
                                           uint32_t *png_status, uint64_t n, uint32_t row_bytes,
                                           uint32_t bpp, void *hip_stream);
int fdh_inflate_png_batch(const uint8_t *in, const uint64_t *in_off, uint8_t *filt,
                          const uint64_t *filt_off, uint32_t *out_len, uint32_t *status,
                          uint32_t *adler, uint8_t *pix, const uint64_t *pix_off,
                          uint32_t *png_status, uint64_t n, uint32_t flags, uint32_t row_bytes,
                          uint32_t bpp, void *hip_stream);

/* ---- streaming decoder: `Decompressor` (src/decompress.rs:96-156, 179-342) ----------------
 * A host-side object with exactly `Decompressor::read`'s contract on HOST buffers; every bit of
 * decoding is done by fdh_inflate_batch_resumable on the device (the object keeps a device-resident
 * copy of the stream so far, a device output slot and the resume point of its last attempt: an attempt
 * decodes what is new; see csrc/stream_decompressor.cpp).
 *
 *   fdh_decompressor_new            Decompressor::new()            src/decompress.rs:123
 *   fdh_decompressor_ignore_adler32 Decompressor::ignore_adler32() src/decompress.rs:154
 *   fdh_decompressor_is_done        Decompressor::is_done()        src/decompress.rs:340
 *   fdh_decompressor_read           Decompressor::read(input, output, output_position)
 *                                   -> Result<(consumed, produced), DecompressionError>
 *                                                                  src/decompress.rs:179-337
 * `read` writes only output[output_position .. output_position + *produced); bytes in front of
 * output_position are never read or written.  When it returns FDH_SUCCESS with *stream_status ==
 * FDH_STREAM_OK at least one of the reference's post-conditions holds (src/decompress.rs:167-170):
 * the input is fully consumed (almost always: see 1. below), the output is full but there are
 * more bytes, or the stream is complete (is_done).  Once done, read returns (0, 0)
 * (src/decompress.rs:185-187).  A `DecompressionError` is reported in *stream_status (1 + ordinal)
 * and is sticky.  An EMPTY input asks for whatever can still be produced from the bytes already
 * handed over (how the reference's own test harness, src/decompress/tests/test_utils.rs:70-74, and
 * the png crate finish a stream).  Function return = infrastructure status as everywhere else.
 * `output_position > output_len` (a panic in the reference, :189) is FDH_ERR_INVALID_ARGUMENT.
 *
 * Where the (consumed, produced) pairs differ from the reference's -- the bytes delivered over a whole
 * stream, their order, the final status and is_done never do (tests/test_gpu_streaming.py):
 *   1. Input is buffered on the device, so a call usually consumes all of it where the reference stops once
 *      the output is full (src/decompress.rs:167-170).  *consumed < input_len only when 192 KiB (or the
 *      caller's room, if that is more) are waiting unread on the device already: the rest is to be offered
 *      again, as with the reference.  A caller written against the contract ("consumed bytes must not be
 *      offered again, the others must") behaves identically.
 *   2. With more than 256 KiB of received input a call with NON-EMPTY input may return (consumed, 0)
 *      without a decode attempt (attempts are then made when the stream has grown by 1/8 or by 64 KiB,
 *      and on every EMPTY input).  A caller must therefore conclude "truncated" (the reference's InsufficientInput,
 *      src/decompress.rs:1135-1136) only after a read with empty input has produced nothing and
 *      is_done is still false -- which is what the reference's own harness and the png crate do at
 *      the end of their input anyway.  (tests/test_gpu_streaming.py,
 *      test_reference_bounded_loop_over_the_streaming_object: the loop of the reference's own
 *      decompress_to_vec_bounded, src/decompress.rs:1111-1144, with that one flush added, over streams of more
 *      than 256 KiB whole and in 40 000-byte pieces.)
 *   3. (round 6) The bound of 1. never holds a stream up: an attempt that moved nothing -- no new resume point, no new
 *      byte -- is followed by a call that takes input again, bound or no bound, because more input is the only
 *      thing that can move it.  A sequence of calls with input left therefore never returns (0, 0) for ever.
 * Device memory (round 5): like the reference, which keeps its tables and needs the last 32 KiB of the caller's
 * buffer (src/decompress.rs:96-113, 1067-1070), the object keeps what its resume point needs and no more -- the
 * unread input, a copy of the current block's header, 32 KiB of history and the window with what has been decoded
 * ahead of it: well under 1 MiB for a 16 KiB window, whatever the stream's length (fdh_decompressor_device_bytes). */
typedef struct fdh_decompressor fdh_decompressor;
fdh_decompressor *fdh_decompressor_new(void);
void fdh_decompressor_free(fdh_decompressor *d);
void fdh_decompressor_ignore_adler32(fdh_decompressor *d);
int fdh_decompressor_is_done(const fdh_decompressor *d);
/* Introspection: decode attempts made so far (a stream drained through a small window needs O(log) of
 * them: every attempt decodes ahead of what the caller can take, see fdh_decompressor_read). */
uint64_t fdh_decompressor_attempts(const fdh_decompressor *d);
/* Introspection: output bytes decoded by all attempts together.  An attempt goes on from where the last one
 * stopped (fdh_inflate_batch_resumable), so for a stream of N decoded bytes this stays close to N however the
 * input and the room arrive (rounds 1-3: every attempt started at the first byte). */
uint64_t fdh_decompressor_decoded_bytes(const fdh_decompressor *d);
/* Introspection: the most device memory the object's buffers (input tail + header copy, output slot, one
 * record of metadata) have held together, in bytes. */
uint64_t fdh_decompressor_device_bytes(const fdh_decompressor *d);
int fdh_decompressor_read(fdh_decompressor *d, const uint8_t *input, size_t input_len,
                          uint8_t *output, size_t output_len, size_t output_position,
                          size_t *consumed, size_t *produced, uint32_t *stream_status);

/* ---- single-buffer conveniences on HOST memory (names mirror src/lib.rs:29-36) ---------
 * Each stages through the device (H2D, batch of one, D2H) and synchronises.  Results are
 * malloc'd; release with fdh_free().  `*stream_status` receives the per-stream status. */
int fdh_decompress_to_vec(const uint8_t *input, size_t input_len, uint8_t **output,
                          size_t *output_len, uint32_t *stream_status); /* decompress.rs:1079 */
int fdh_decompress_to_vec_bounded(const uint8_t *input, size_t input_len, size_t maxlen,
                                  uint8_t **output, size_t *output_len,
                                  uint32_t *stream_status); /* decompress.rs:1111 */
int fdh_compress_to_vec_ultra_fast(const uint8_t *input, size_t input_len, uint8_t **output,
                                   size_t *output_len); /* compress/mod.rs:313 */
int fdh_compress_to_vec_stored(const uint8_t *input, size_t input_len, uint8_t **output,
                               size_t *output_len); /* compress_to_vec_with_level(.., 0), compress/mod.rs:299 */
int fdh_compress_to_vec(const uint8_t *input, size_t input_len, uint8_t **output,
                        size_t *output_len); /* compress_to_vec = level 1, compress/mod.rs:294 */
int fdh_compress_to_vec_rle(const uint8_t *input, size_t input_len, uint8_t **output,
                            size_t *output_len); /* compress_to_vec_rle, compress/mod.rs:306 */
void fdh_free(void *p);

/* ---- several GPUs of one node, one process --------------------------------
 * Streams are independent: a batch is sharded by contiguous stream ranges, one shard per device,
 * decoded with no data-path exchange; the only collective is an all-gather of the per-stream
 * results {status, out_len, adler} over RCCL / xGMI (librccl is loaded on demand, and only when
 * more than one device takes part).  There is no reference counterpart: the crate is
 * single-threaded; this is how a batch API scales it across the node.
 *   fdh_init(device_mask)   devices with bit d set (0 = every visible device): HIP streams, the
 *                           shared decode tables and the RCCL communicator; call again to change
 *   fdh_shutdown()          releases them
 *   fdh_inflate_batch_multi `n_shards` must equal the number of initialised devices; shard i holds
 *                           device pointers ON device i (the i-th selected one) with the meaning of
 *                           fdh_inflate_batch.  If `meta_all` is given (for every shard), device i
 *                           receives the results of ALL shards there: n_shards x 3 x meta_stride
 *                           words, [shard][status | out_len | adler][stream], zero padded.
 *                           Returns when every device has finished. */
typedef struct fdh_shard {
    const uint8_t *in;
    const uint64_t *in_off;
    uint8_t *out;
    const uint64_t *out_off;
    uint32_t *out_len;
    uint32_t *status;
    uint32_t *adler;    /* nullable */
    uint64_t n;
    uint32_t *meta_all; /* nullable (for every shard or for none) */
} fdh_shard_t;
int fdh_init(uint64_t device_mask);
int fdh_shutdown(void);
int fdh_multi_device_count(void); /* devices selected by the last fdh_init, 0 before */
int fdh_multi_uses_rccl(void);    /* 1 if the gather goes through RCCL: more than one device, or
                                     FDH_MULTI_FORCE_RCCL=1 in the environment of fdh_init (a one-rank
                                     communicator: the way to run that path on a one-GPU box) */
int fdh_inflate_batch_multi(const fdh_shard_t *shards, uint32_t n_shards, uint32_t flags,
                            uint64_t meta_stride);

/* ---- introspection ------------------------------------------------------------------- */
uint32_t fdh_version(void);
const char *fdh_status_name(uint32_t stream_status); /* "Ok", "BadZlibHeader", ... */
const char *fdh_last_error(void);                    /* thread-local message of the last failure */
int fdh_device_count(void);                          /* usable gfx950 devices, 0 if none */

/* Debug / parity hook: run the device Huffman-table builder (the restatement of
 * huffman::build_table + CompressedBlock::build_tables, src/huffman.rs:18-184,
 * src/decompress.rs:561-606) on `code_lengths[320]` and return the decode tables in the
 * library's device layout (documented in DESIGN.md): litlen[4096], dist[512] u32 entries.
 * `build_status` = FDH_STREAM_OK or the error build_tables would return.  Device pointers. */
int fdh_debug_build_tables(const uint8_t *code_lengths320, uint32_t hlit, uint32_t *litlen4096,
                           uint32_t *dist512, uint32_t *build_status, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
