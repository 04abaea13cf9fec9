// fdeflate_hip.cpp -- host side of the C ABI declared in include/fdeflate_hip.h.
// Thin: argument checks, kernel launches, and H2D/D2H staging for the single-buffer
// conveniences.  There is deliberately no CPU decode/encode path in this library.
#include "../../include/fdeflate_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

extern "C" {
int fdh_launch_inflate(const uint8_t* in, const uint64_t* in_off, uint8_t* out, const uint64_t* out_off,
                       uint32_t* out_len, uint32_t* status, uint32_t* adler, uint64_t n, uint32_t flags,
                       void* resume_io, hipStream_t stream);
int fdh_launch_canon_build(hipStream_t stream, uint32_t* host_status);
int fdh_launch_build_tables_debug(const uint8_t* code_lengths, uint32_t hlit, uint32_t* litlen, uint32_t* dist,
                                  uint32_t* build_status, hipStream_t stream);
int fdh_launch_deflate_stored(const uint8_t* in, const uint64_t* in_off, uint8_t* out, const uint64_t* out_off,
                              uint32_t* out_len, uint64_t n, hipStream_t stream);
int fdh_launch_png_filter_deflate_ultrafast(const uint8_t* pix, const uint64_t* pix_off, const uint8_t* types,
                                            const uint64_t* types_off, uint8_t* out, const uint64_t* out_off,
                                            uint32_t* out_len, uint32_t* png_status, uint64_t n, uint32_t row_bytes,
                                            uint32_t bpp, hipStream_t stream);
int fdh_launch_deflate_ultrafast(const uint8_t* in, const uint64_t* in_off, uint8_t* out, const uint64_t* out_off,
                                 uint32_t* out_len, uint64_t n, hipStream_t stream);
int fdh_launch_deflate_general(const uint8_t* in, const uint64_t* in_off, uint8_t* out, const uint64_t* out_off,
                               uint32_t* out_len, uint64_t n, int rle, void* hash, void* matches, void* blocks,
                               uint32_t* nblocks, unsigned waves, unsigned lanes, hipStream_t stream);
int fdh_launch_png_unfilter(const uint8_t* filt, const uint64_t* filt_off, uint8_t* pix, const uint64_t* pix_off,
                            uint32_t* status, const uint32_t* gate, const uint32_t* gate_len, uint64_t n,
                            uint32_t row_bytes, uint32_t bpp, hipStream_t stream);
int fdh_launch_png_filter(const uint8_t* pix, const uint64_t* pix_off, const uint8_t* types, const uint64_t* types_off,
                          uint8_t* filt, const uint64_t* filt_off, uint32_t* status, uint64_t n, uint32_t row_bytes,
                          uint32_t bpp, hipStream_t stream);
size_t fdh_deflate_general_hash_bytes(void);
size_t fdh_deflate_general_match_records(uint64_t total_in, uint64_t n);
size_t fdh_deflate_general_block_records(uint64_t total_in, uint64_t n);
size_t fdh_deflate_general_match_record_bytes(void);
size_t fdh_deflate_general_block_record_bytes(void);
}

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

int hip_fail(hipError_t e, const char* what) {
    return fail(e == hipErrorOutOfMemory ? FDH_ERR_OUT_OF_MEMORY : FDH_ERR_HIP,
                std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) return hip_fail(e_, #expr); \
    } while (0)

bool have_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return false;
    return n > 0;
}

// The shared decode tables of the ultra-fast prefix are built on the device once per device.
std::mutex g_canon_mutex;
bool g_canon_ready[64] = {};

int ensure_canon_tables(hipStream_t stream) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(FDH_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    std::lock_guard<std::mutex> lock(g_canon_mutex);
    if (g_canon_ready[dev]) return FDH_SUCCESS;
    uint32_t st = 0xFFFFFFFFu;
    int rc = fdh_launch_canon_build(stream, &st);
    if (rc != 0) return hip_fail(static_cast<hipError_t>(rc), "canonical table build");
    if (st != 0) return fail(FDH_ERR_HIP, "canonical table build returned status " + std::to_string(st));
    g_canon_ready[dev] = true;
    return FDH_SUCCESS;
}

const char* kStatusNames[] = {
    "Ok", "BadZlibHeader", "InsufficientInput", "InvalidBlockType", "InvalidUncompressedBlockLength",
    "InvalidHlit", "InvalidHdist", "InvalidCodeLengthRepeat", "BadCodeLengthHuffmanTree",
    "BadLiteralLengthHuffmanTree", "BadDistanceHuffmanTree", "InvalidLiteralLengthCode",
    "InvalidDistanceCode", "InputStartsWithRun", "DistanceTooFarBack", "WrongChecksum", "ExtraInput",
    "OutputTooLarge"};

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
    template <class T>
    T* as() { return static_cast<T*>(p); }
};

}  // namespace

extern "C" {

uint32_t fdh_version(void) { return FDH_VERSION; }

const char* fdh_status_name(uint32_t s) { return s < 18 ? kStatusNames[s] : "Unknown"; }

const char* fdh_last_error(void) { return g_last_error.c_str(); }

// used by the other translation units of the library (not part of the public header)
void fdh_set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }

int fdh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

uint64_t fdh_ultrafast_bound(uint64_t len) { return 53 + (5 + 12 * len + 12 + 7) / 8 + 4; }

uint64_t fdh_stored_size(uint64_t len) {
    const uint64_t nb = len / 65535, rem = len - nb * 65535;
    return 2 + nb * (5 + 65535) + (rem ? 5 + rem : 2) + 4;
}

int fdh_deflate_stored_batch(const uint8_t* in, const uint64_t* in_off, uint8_t* out, const uint64_t* out_off,
                             uint32_t* out_len, uint64_t n, void* hip_stream) {
    if (n == 0) return FDH_SUCCESS;
    if (!in_off || !out_off || !out_len) return fail(FDH_ERR_INVALID_ARGUMENT, "null metadata pointer");
    if (n > 0x7FFFFFFFull) return fail(FDH_ERR_INVALID_ARGUMENT, "too many buffers in one call (max 2^31-1)");
    if (!have_device()) return fail(FDH_ERR_NO_DEVICE, "no HIP device: fdeflate_hip has no CPU fallback");
    int rc = fdh_launch_deflate_stored(in, in_off, out, out_off, out_len, n, static_cast<hipStream_t>(hip_stream));
    if (rc != 0) return hip_fail(static_cast<hipError_t>(rc), "stored-encoder kernel launch");
    return FDH_SUCCESS;
}

int fdh_inflate_batch(const uint8_t* in, const uint64_t* in_off, uint8_t* out, const uint64_t* out_off,
                      uint32_t* out_len, uint32_t* status, uint32_t* adler, uint64_t n, uint32_t flags,
                      void* hip_stream) {
    if (n == 0) return FDH_SUCCESS;
    if (!in_off || !out_off || !out_len || !status) return fail(FDH_ERR_INVALID_ARGUMENT, "null metadata pointer");
    if (n > 0x7FFFFFFFull) return fail(FDH_ERR_INVALID_ARGUMENT, "too many streams in one call (max 2^31-1)");
    if (!have_device()) return fail(FDH_ERR_NO_DEVICE, "no HIP device: fdeflate_hip has no CPU fallback");
    int rc = ensure_canon_tables(static_cast<hipStream_t>(hip_stream));
    if (rc != FDH_SUCCESS) return rc;
    rc = fdh_launch_inflate(in, in_off, out, out_off, out_len, status, adler, n, flags & ~FDH_FLAG_RESUME_IN, nullptr,
                            static_cast<hipStream_t>(hip_stream));
    if (rc != 0) return hip_fail(static_cast<hipError_t>(rc), "inflate kernel launch");
    return FDH_SUCCESS;
}

int fdh_inflate_batch_resumable(const uint8_t* in, const uint64_t* in_off, uint8_t* out, const uint64_t* out_off,
                                uint32_t* out_len, uint32_t* status, uint32_t* adler, uint64_t n, uint32_t flags,
                                fdh_resume_point* resume, void* hip_stream) {
    if (n == 0) return FDH_SUCCESS;
    if (!in_off || !out_off || !out_len || !status || !resume) return fail(FDH_ERR_INVALID_ARGUMENT, "null metadata pointer");
    if (n > 0x7FFFFFFFull) return fail(FDH_ERR_INVALID_ARGUMENT, "too many streams in one call (max 2^31-1)");
    if (!have_device()) return fail(FDH_ERR_NO_DEVICE, "no HIP device: fdeflate_hip has no CPU fallback");
    int rc = ensure_canon_tables(static_cast<hipStream_t>(hip_stream));
    if (rc != FDH_SUCCESS) return rc;
    rc = fdh_launch_inflate(in, in_off, out, out_off, out_len, status, adler, n, flags, resume, static_cast<hipStream_t>(hip_stream));
    if (rc != 0) return hip_fail(static_cast<hipError_t>(rc), "inflate kernel launch");
    return FDH_SUCCESS;
}

int fdh_deflate_ultrafast_batch(const uint8_t* in, const uint64_t* in_off, uint8_t* out, const uint64_t* out_off,
                                uint32_t* out_len, uint64_t n, void* hip_stream) {
    if (n == 0) return FDH_SUCCESS;
    if (!in_off || !out_off || !out_len) return fail(FDH_ERR_INVALID_ARGUMENT, "null metadata pointer");
    if (n > 0x7FFFFFFFull) return fail(FDH_ERR_INVALID_ARGUMENT, "too many buffers in one call (max 2^31-1)");
    if (!have_device()) return fail(FDH_ERR_NO_DEVICE, "no HIP device: fdeflate_hip has no CPU fallback");
    int rc = fdh_launch_deflate_ultrafast(in, in_off, out, out_off, out_len, n, static_cast<hipStream_t>(hip_stream));
    if (rc != 0) return hip_fail(static_cast<hipError_t>(rc), "deflate kernel launch");
    return FDH_SUCCESS;
}

int fdh_debug_build_tables(const uint8_t* code_lengths320, uint32_t hlit, uint32_t* litlen4096, uint32_t* dist512,
                           uint32_t* build_status, void* hip_stream) {
    if (!have_device()) return fail(FDH_ERR_NO_DEVICE, "no HIP device: fdeflate_hip has no CPU fallback");
    int rc = fdh_launch_build_tables_debug(code_lengths320, hlit, litlen4096, dist512, build_status,
                                           static_cast<hipStream_t>(hip_stream));
    if (rc != 0) return hip_fail(static_cast<hipError_t>(rc), "table-build kernel launch");
    return FDH_SUCCESS;
}

// ---- PNG scanline filters (the steps either side of the codec in the PNG pipeline) ----
static int png_args_ok(const void* a, const void* b, const void* c, const void* d, const void* st, uint32_t row_bytes,
                       uint32_t bpp) {
    if (!a || !b || !c || !d || !st) return fail(FDH_ERR_INVALID_ARGUMENT, "null pointer");
    if (row_bytes == 0) return fail(FDH_ERR_INVALID_ARGUMENT, "row_bytes must be positive");
    if (!(bpp == 1 || bpp == 2 || bpp == 3 || bpp == 4 || bpp == 6 || bpp == 8))
        return fail(FDH_ERR_INVALID_ARGUMENT, "bpp must be 1, 2, 3, 4, 6 or 8 (PNG's whole-byte pixel sizes)");
    if (!have_device()) return fail(FDH_ERR_NO_DEVICE, "no HIP device: fdeflate_hip has no CPU fallback");
    return FDH_SUCCESS;
}

int fdh_png_unfilter_batch(const uint8_t* filt, const uint64_t* filt_off, uint8_t* pix, const uint64_t* pix_off,
                           uint32_t* png_status, uint64_t n, uint32_t row_bytes, uint32_t bpp, void* hip_stream) {
    if (n == 0) return FDH_SUCCESS;
    int rc = png_args_ok(filt, filt_off, pix, pix_off, png_status, row_bytes, bpp);
    if (rc != FDH_SUCCESS) return rc;
    rc = fdh_launch_png_unfilter(filt, filt_off, pix, pix_off, png_status, nullptr, nullptr, n, row_bytes, bpp,
                                 static_cast<hipStream_t>(hip_stream));
    if (rc != 0) return hip_fail(static_cast<hipError_t>(rc), "unfilter kernel launch");
    return FDH_SUCCESS;
}

int fdh_png_filter_batch(const uint8_t* pix, const uint64_t* pix_off, const uint8_t* types, const uint64_t* types_off,
                         uint8_t* filt, const uint64_t* filt_off, uint32_t* png_status, uint64_t n, uint32_t row_bytes,
                         uint32_t bpp, void* hip_stream) {
    if (n == 0) return FDH_SUCCESS;
    int rc = png_args_ok(pix, pix_off, filt, filt_off, png_status, row_bytes, bpp);
    if (rc != FDH_SUCCESS) return rc;
    if (!types || !types_off) return fail(FDH_ERR_INVALID_ARGUMENT, "null pointer");
    rc = fdh_launch_png_filter(pix, pix_off, types, types_off, filt, filt_off, png_status, n, row_bytes, bpp,
                               static_cast<hipStream_t>(hip_stream));
    if (rc != 0) return hip_fail(static_cast<hipError_t>(rc), "filter kernel launch");
    return FDH_SUCCESS;
}

int fdh_png_filter_deflate_ultrafast_batch(const uint8_t* pix, const uint64_t* pix_off, const uint8_t* types,
                                           const uint64_t* types_off, uint8_t* out, const uint64_t* out_off,
                                           uint32_t* out_len, uint32_t* png_status, uint64_t n, uint32_t row_bytes,
                                           uint32_t bpp, void* hip_stream) {
    if (n == 0) return FDH_SUCCESS;
    int rc = png_args_ok(pix, pix_off, out, out_off, png_status, row_bytes, bpp);
    if (rc != FDH_SUCCESS) return rc;
    if (!types || !types_off || !out_len) return fail(FDH_ERR_INVALID_ARGUMENT, "null pointer");
    if (n > 0x7FFFFFFFull) return fail(FDH_ERR_INVALID_ARGUMENT, "too many buffers in one call (max 2^31-1)");
    rc = fdh_launch_png_filter_deflate_ultrafast(pix, pix_off, types, types_off, out, out_off, out_len, png_status, n,
                                                 row_bytes, bpp, static_cast<hipStream_t>(hip_stream));
    if (rc != 0) return hip_fail(static_cast<hipError_t>(rc), "filter + deflate kernel launch");
    return FDH_SUCCESS;
}

int fdh_inflate_png_batch(const uint8_t* in, const uint64_t* in_off, uint8_t* filt, const uint64_t* filt_off,
                          uint32_t* out_len, uint32_t* status, uint32_t* adler, uint8_t* pix, const uint64_t* pix_off,
                          uint32_t* png_status, uint64_t n, uint32_t flags, uint32_t row_bytes, uint32_t bpp,
                          void* hip_stream) {
    if (n == 0) return FDH_SUCCESS;
    int rc = png_args_ok(filt, filt_off, pix, pix_off, png_status, row_bytes, bpp);
    if (rc != FDH_SUCCESS) return rc;
    rc = fdh_inflate_batch(in, in_off, filt, filt_off, out_len, status, adler, n, flags, hip_stream);
    if (rc != FDH_SUCCESS) return rc;
    // same stream: the scanlines are reconstructed as soon as the decode kernels have finished, only
    // for the streams that decoded (status 0) -- the rest gets png_status 3 -- and that decoded to
    // exactly the bytes of their slot: a stream that ends early would leave stale bytes behind it
    // (png_status 2; the png crate treats short IDAT data as an error as well)
    rc = fdh_launch_png_unfilter(filt, filt_off, pix, pix_off, png_status, status, out_len, n, row_bytes, bpp,
                                 static_cast<hipStream_t>(hip_stream));
    if (rc != 0) return hip_fail(static_cast<hipError_t>(rc), "unfilter kernel launch");
    return FDH_SUCCESS;
}

// ---- general encoder (level 1 / RLE): per-device workspace, grown on demand, never shrunk ----
namespace {
struct GenWork {
    std::mutex mutex;         // the workspace is shared by the calls on ITS device only
    void* hash = nullptr;     // one 64 Ki-entry table per resident lane of the parser (level 1)
    void* matches = nullptr;  // what the parser hands to the block writer, sliced per stream
    void* blocks = nullptr;
    void* nblocks = nullptr;
    size_t hash_bytes = 0, match_bytes = 0, block_bytes = 0, nblock_bytes = 0;
};
GenWork g_gen_work[64];

int grow(void** p, size_t* have, size_t want, const char* what, bool headroom = true) {
    if (*have >= want) return FDH_SUCCESS;
    HIP_TRY(hipDeviceSynchronize());  // nobody may still be using the old buffer
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    const size_t sz = headroom ? want + want / 4 : want;  // headroom: batches of similar size do not reallocate
    hipError_t e = hipMalloc(p, sz);
    if (e != hipSuccess) {
        *p = nullptr;
        return hip_fail(e, what);
    }
    *have = sz;
    return FDH_SUCCESS;
}
}  // namespace

uint64_t fdh_compress_bound(uint64_t len) { return len + len / 2 + 1024; }

int fdh_deflate_general_batch(const uint8_t* in, const uint64_t* in_off, uint8_t* out, const uint64_t* out_off,
                              uint32_t* out_len, uint64_t n, uint32_t mode, void* hip_stream) {
    if (n == 0) return FDH_SUCCESS;
    if (!in_off || !out_off || !out_len) return fail(FDH_ERR_INVALID_ARGUMENT, "null metadata pointer");
    if (!out) return fail(FDH_ERR_INVALID_ARGUMENT, "null data pointer");  // (`in` may be null for a batch of empty inputs: below)
    if (mode != FDH_MODE_LEVEL1 && mode != FDH_MODE_RLE) return fail(FDH_ERR_INVALID_ARGUMENT, "unknown encoder mode");
    if (n > 0x7FFFFFFFull) return fail(FDH_ERR_INVALID_ARGUMENT, "too many streams in one call");
    if (!have_device()) return fail(FDH_ERR_NO_DEVICE, "no HIP device: fdeflate_hip has no CPU fallback");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(FDH_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    // the record arrays are sized by the bytes the batch spans
    uint64_t ends[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(&ends[0], in_off, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(&ends[1], in_off + n, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (ends[1] < ends[0]) return fail(FDH_ERR_INVALID_ARGUMENT, "in_off is not ascending");
    const uint64_t total_in = ends[1] - ends[0];
    // an empty input has a defined encoding (78 01 03 00 00 00 00 01), and a zero-element buffer has no address
    if (!in && total_in != 0) return fail(FDH_ERR_INVALID_ARGUMENT, "null data pointer");
    // The parser runs one stream per lane and is bound by the latency of dependent loads: what helps is wavefronts in
    // flight.  A batch that does not fill the device with full wavefronts is given fewer lanes per wavefront and more
    // wavefronts -- 8 per CU at level 1 (201 VGPRs: two per SIMD), 16 per CU for the RLE parser (126 VGPRs, no hash
    // tables) -- and at level 1 the streams in flight are capped so that their hash tables (256 KiB each) stay below
    // 16 GiB.  (Round 5: the cap was 8 GiB and applied to the RLE parser too, which has no tables: 65 536 streams ran as
    // 1 024 wavefronts of 32 lanes in two rounds, one wavefront per SIMD.  All of them in flight: level 1 39.2 -> 31.7 ms,
    // RLE 25.4 -> 21.9 ms.)
    const bool rle = mode == FDH_MODE_RLE;
    unsigned lanes = 64;
    const uint64_t want_waves = (uint64_t)cus * (rle ? 16 : 8);
    while (lanes > 4 && (n + lanes - 1) / lanes < want_waves) lanes /= 2;
    if (const char* e = std::getenv("FDH_GEN_LANES")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 64) lanes = (unsigned)v;
    }
    uint64_t max_resident = rle ? (1ull << 40) : 65536;
    if (const char* e = std::getenv("FDH_GEN_RESIDENT")) {
        const long long v = std::atoll(e);
        if (v >= 64 && v <= (1 << 20)) max_resident = (uint64_t)v;
    }
    unsigned waves = (unsigned)std::min<uint64_t>((n + lanes - 1) / lanes, std::max<uint64_t>(1, max_resident / lanes));
    GenWork& w = g_gen_work[dev];
    std::lock_guard<std::mutex> lock(w.mutex);
    int rc = FDH_SUCCESS;
    if (!rle) {
        // exactly the resident lanes' tables (no headroom: at the cap that is the documented 8 GiB);
        // on a smaller or busy device the batch runs with fewer resident wavefronts instead of failing
        for (;;) {
            rc = grow(&w.hash, &w.hash_bytes, (size_t)waves * lanes * fdh_deflate_general_hash_bytes(), "hipMalloc(hash tables)", false);
            if (rc == FDH_SUCCESS || waves <= 1) break;
            (void)hipGetLastError();
            waves /= 2;
        }
    }
    if (rc == FDH_SUCCESS)
        rc = grow(&w.matches, &w.match_bytes, fdh_deflate_general_match_records(total_in, n) * fdh_deflate_general_match_record_bytes(),
                  "hipMalloc(back-reference records)");
    if (rc == FDH_SUCCESS)
        rc = grow(&w.blocks, &w.block_bytes, fdh_deflate_general_block_records(total_in, n) * fdh_deflate_general_block_record_bytes(),
                  "hipMalloc(block records)");
    if (rc == FDH_SUCCESS) rc = grow(&w.nblocks, &w.nblock_bytes, (size_t)n * 4, "hipMalloc(block counts)");
    if (rc != FDH_SUCCESS) return rc;
    rc = fdh_launch_deflate_general(in, in_off, out, out_off, out_len, n, rle, w.hash, w.matches, w.blocks,
                                    static_cast<uint32_t*>(w.nblocks), waves, lanes, stream);
    if (rc != 0) return hip_fail(static_cast<hipError_t>(rc), "general-encoder kernel launch");
    // the workspace is per device, not per stream: calls are serialised by finishing this one
    HIP_TRY(hipStreamSynchronize(stream));
    return FDH_SUCCESS;
}

// ---- single-buffer conveniences (host memory) --------------------------------------------

// One decode of a host buffer into a device slot of `cap` bytes; the decoded (or partial) bytes
// are returned in a malloc'd buffer.  The device copy of the input is kept by the caller so that
// a retry with a larger slot does not upload it again.
static int inflate_one(DevBuf& d_in, size_t input_len, size_t cap, uint8_t** output, size_t* output_len,
                       uint32_t* stream_status) {
    DevBuf d_out, d_meta;
    HIP_TRY(d_out.alloc(cap));
    HIP_TRY(d_meta.alloc(64));
    uint64_t meta[8] = {0, (uint64_t)input_len, 0, (uint64_t)cap, 0, 0, 0, 0};
    HIP_TRY(hipMemcpy(d_meta.p, meta, sizeof(meta), hipMemcpyHostToDevice));
    uint64_t* m = d_meta.as<uint64_t>();
    uint32_t* res = reinterpret_cast<uint32_t*>(m + 4);
    int rc = fdh_inflate_batch(d_in.as<uint8_t>(), m, d_out.as<uint8_t>(), m + 2, res, res + 1, res + 2, 1, 0, nullptr);
    if (rc != FDH_SUCCESS) return rc;
    uint32_t host_res[4];
    HIP_TRY(hipMemcpy(host_res, res, sizeof(host_res), hipMemcpyDeviceToHost));  // synchronises the null stream
    *stream_status = host_res[1];
    size_t n = host_res[0];
    if (n > cap) n = cap;
    uint8_t* buf = static_cast<uint8_t*>(std::malloc(n ? n : 1));
    if (!buf) return fail(FDH_ERR_OUT_OF_MEMORY, "malloc");
    if (n) {
        hipError_t e = hipMemcpy(buf, d_out.p, n, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            std::free(buf);
            return hip_fail(e, "hipMemcpy(decoded bytes)");
        }
    }
    *output = buf;
    *output_len = n;
    return FDH_SUCCESS;
}

// decompress_to_vec_bounded (src/decompress.rs:1111-1144).  The reference grows its Vec from 1 KiB
// by 32 KiB steps up to `maxlen` (:1117, :1133); the device needs a slot size up front, so the slot
// starts at min(maxlen, 4 x input + 64 KiB) and is quadrupled (up to maxlen) while the stream
// reports OutputTooLarge below maxlen -- a huge `maxlen` never allocates more than the stream
// needs (x4), and the result is the one a slot of `maxlen` bytes would have given.
static int inflate_growing(const uint8_t* input, size_t input_len, size_t maxlen, uint8_t** output,
                           size_t* output_len, uint32_t* stream_status) {
    if (!output || !output_len || !stream_status) return fail(FDH_ERR_INVALID_ARGUMENT, "null result pointer");
    *output = nullptr;
    *output_len = 0;
    if (!have_device()) return fail(FDH_ERR_NO_DEVICE, "no HIP device: fdeflate_hip has no CPU fallback");
    if (input_len >= (1ull << 31)) return fail(FDH_ERR_INVALID_ARGUMENT, "stream too large (>= 2 GiB)");
    if (maxlen > 0xFFFFFFF0ull) maxlen = 0xFFFFFFF0ull;
    DevBuf d_in;
    HIP_TRY(d_in.alloc(input_len));
    if (input_len) HIP_TRY(hipMemcpy(d_in.p, input, input_len, hipMemcpyHostToDevice));
    size_t cap = std::min<size_t>(maxlen, input_len * 4 + 65536);
    for (;;) {
        int rc = inflate_one(d_in, input_len, cap, output, output_len, stream_status);
        if (rc != FDH_SUCCESS) return rc;
        if (*stream_status != FDH_OUTPUT_TOO_LARGE || cap >= maxlen) return FDH_SUCCESS;
        std::free(*output);
        *output = nullptr;
        *output_len = 0;
        cap = cap > maxlen / 4 ? maxlen : cap * 4;
    }
}

int fdh_decompress_to_vec_bounded(const uint8_t* input, size_t input_len, size_t maxlen, uint8_t** output,
                                  size_t* output_len, uint32_t* stream_status) {
    return inflate_growing(input, input_len, maxlen, output, output_len, stream_status);
}

// decompress_to_vec grows its Vec without bound (src/decompress.rs:1079-1087): the same loop with
// the ABI's largest slot as the bound.
int fdh_decompress_to_vec(const uint8_t* input, size_t input_len, uint8_t** output, size_t* output_len,
                          uint32_t* stream_status) {
    return inflate_growing(input, input_len, 0xFFFFFFF0ull, output, output_len, stream_status);
}

static int compress_one(int kind, const uint8_t* input, size_t input_len, uint8_t** output, size_t* output_len) {
    const bool stored = kind == 1;
    if (!output || !output_len) return fail(FDH_ERR_INVALID_ARGUMENT, "null result pointer");
    if (input_len >= 0xFFFFFFFFull) return fail(FDH_ERR_INVALID_ARGUMENT, "buffer too large (>= 4 GiB)");
    if (!have_device()) return fail(FDH_ERR_NO_DEVICE, "no HIP device: fdeflate_hip has no CPU fallback");
    size_t cap = (size_t)(stored ? fdh_stored_size(input_len) : (kind == 0 ? fdh_ultrafast_bound(input_len) : fdh_compress_bound(input_len)));
    DevBuf d_in, d_out, d_meta;
    HIP_TRY(d_in.alloc(input_len));
    HIP_TRY(d_out.alloc(cap));
    HIP_TRY(d_meta.alloc(64));
    uint64_t meta[8] = {0, (uint64_t)input_len, 0, (uint64_t)cap, 0, 0, 0, 0};
    if (input_len) HIP_TRY(hipMemcpy(d_in.p, input, input_len, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_meta.p, meta, sizeof(meta), hipMemcpyHostToDevice));
    uint64_t* m = d_meta.as<uint64_t>();
    uint32_t* res = reinterpret_cast<uint32_t*>(m + 4);
    int rc = stored      ? fdh_deflate_stored_batch(d_in.as<uint8_t>(), m, d_out.as<uint8_t>(), m + 2, res, 1, nullptr)
             : kind == 0 ? fdh_deflate_ultrafast_batch(d_in.as<uint8_t>(), m, d_out.as<uint8_t>(), m + 2, res, 1, nullptr)
                         : fdh_deflate_general_batch(d_in.as<uint8_t>(), m, d_out.as<uint8_t>(), m + 2, res, 1,
                                                     kind == 2 ? FDH_MODE_LEVEL1 : FDH_MODE_RLE, nullptr);
    if (rc != FDH_SUCCESS) return rc;
    HIP_TRY(hipDeviceSynchronize());
    uint32_t n32 = 0;
    HIP_TRY(hipMemcpy(&n32, res, 4, hipMemcpyDeviceToHost));
    if (n32 == 0xFFFFFFFFu) return fail(FDH_ERR_HIP, "internal: encoder bound exceeded");
    uint8_t* buf = static_cast<uint8_t*>(std::malloc(n32 ? n32 : 1));
    if (!buf) return fail(FDH_ERR_OUT_OF_MEMORY, "malloc");
    if (n32) {
        hipError_t e = hipMemcpy(buf, d_out.p, n32, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            std::free(buf);
            return hip_fail(e, "hipMemcpy(compressed bytes)");
        }
    }
    *output = buf;
    *output_len = n32;
    return FDH_SUCCESS;
}

int fdh_compress_to_vec_ultra_fast(const uint8_t* input, size_t input_len, uint8_t** output, size_t* output_len) {
    return compress_one(0, input, input_len, output, output_len);
}

int fdh_compress_to_vec_stored(const uint8_t* input, size_t input_len, uint8_t** output, size_t* output_len) {
    return compress_one(1, input, input_len, output, output_len);
}

int fdh_compress_to_vec(const uint8_t* input, size_t input_len, uint8_t** output, size_t* output_len) {
    return compress_one(2, input, input_len, output, output_len);
}

int fdh_compress_to_vec_rle(const uint8_t* input, size_t input_len, uint8_t** output, size_t* output_len) {
    return compress_one(3, input, input_len, output, output_len);
}

void fdh_free(void* p) { std::free(p); }

}  // extern "C"
