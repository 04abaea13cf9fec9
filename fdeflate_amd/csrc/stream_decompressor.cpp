// stream_decompressor.cpp -- `Decompressor::read` (reference src/decompress.rs:158-337) on top of
// the batch engine.
//
// The reference decoder is resumable at any input / output split (State :84-93, BitBuffer,
// QueuedOutput :1067-1070).  The GPU engine is one-shot, so this object keeps the streaming
// CONTRACT on the host and lets the device do every bit of decoding:
//
//   * input handed to `read` is appended to a device-resident TAIL of the stream: the bytes from 16 in front of
//     the last resume point on.  The reference keeps no input at all (the caller owns it) and stops consuming
//     when the output is full (src/decompress.rs:167-170); this object stops consuming when kInCap bytes (or the
//     caller's room, if that is more) are waiting unread on the device -- *consumed < input_len then, and the
//     caller offers the rest again, as with the reference;
//   * the stream-so-far is decoded by fdh_inflate_batch_resumable (batch of one) into a device slot that holds
//     what the caller can take now (bytes delivered so far + the room left in `output`) and, decoding ahead,
//     up to kAheadMax more (see below); the one-shot classification (src/decompress.rs:1126-1139) of the attempt
//     whose prefix is used up says which post-condition holds: Ok -> done; OutputTooLarge -> "the output is full
//     but there are more bytes"; InsufficientInput -> the engine reports how many bytes the reference had
//     produced when it ran dry, and exactly those are delivered;
//   * only output[output_position ..] is written, never more than the room, and the bytes in
//     front of output_position are not needed (the LZ77 history lives in the device slot);
//   * an attempt goes on where the last one stopped: fdh_inflate_batch_resumable hands back a resume
//     point (bit position at the start of one of the reference's decoding steps, block header, output
//     bytes, Adler-32) for a stream that ran out of input or room, and takes the stream up there in the
//     next call -- the counterpart of the reference's State / BitBuffer / QueuedOutput, kept on the device.
//
// THE REFERENCE'S FOOTPRINT (round 5).  The reference keeps its tables and works in the caller's buffer, of which
// it needs the last 32 KiB (src/decompress.rs:96-113, 1067-1070).  A resume point needs as little: the 32 KiB of
// output in front of it, the input from its bit on -- and the header of its block, which the kernels parse again
// for the tables (inflate_stream.h: run_from; they SEEK from the header to the bit, so nothing in between is
// needed).  The device therefore holds
//     input   [ 16 B: the stream's first bytes (zlib header) | the block header's copy, kHdrBytes | the tail ]
//     output  [ out_base, cap ):  out_base = min(delivered, resume point - 32 KiB), rounded down to a 16-B line
// and every attempt sees a REBASED stream: bit positions and output positions relative to these buffers (the
// resume record is translated on the way in and on the way out; a side effect: no 128 MiB limit on resumable
// streams -- a record's 30-bit positions never grow).  An ultra-fast stream is ONE block whose header is at
// the start: its copy travels with the tail however far the decoder has come.  Buffers are trimmed by moving
// their contents to the front once kTrimStep bytes can go (device copies in pieces that do not overlap their
// destination); they reach their steady size after a few calls and are never reallocated after that
// (fdh_decompressor_device_bytes: 0.6-0.8 MiB for a 16 KiB window whatever the stream's length).
//
// Draining through a small output window: an attempt decodes AHEAD of what the caller can take (slot =
// twice what has been delivered + 64 KiB, at most kAheadMax beyond the window) and the following calls are served
// from that prefix, so the per-attempt cost (a few launches and a round trip) is paid once per kAheadMax bytes.  A
// hard error in the part decoded ahead is not reported early: the attempt is repeated with the exact
// slot, whose classification is the reference's.
// An attempt costs a few kernel launches and a round trip to the device whatever it decodes;
// above kAlwaysBelow received bytes an attempt is made only once the stream has grown by 1/8 (at most kGrow) --
// or when the unread input on the device is about to reach its bound, or when the caller passes an empty `input`,
// which is how both the reference's test harness (src/decompress/tests/test_utils.rs:70-74: chunk size 0 once the
// input is exhausted) and the png crate's finish loop ask for whatever can still be produced.
//
// There is no CPU decode path here: without a GPU every call returns FDH_ERR_NO_DEVICE.
#include "../../include/fdeflate_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

extern "C" void fdh_set_last_error(const char* msg);
extern "C" int fdh_launch_copy_lines(void* dst, const void* src, size_t bytes, hipStream_t stream);  // deflate_stored.hip

namespace {

constexpr size_t kAlwaysBelow = 256 * 1024;
constexpr size_t kHdrAt = 16;        // device input: [0, 16) the stream's first bytes, [16, 16 + kHdrBytes) the block header's copy
constexpr size_t kHdrBytes = 1056;   // a dynamic header is at most 17 + 57 + 316 x 14 bits = 563 bytes, + 15 of alignment; the rest is slack
constexpr size_t kHead = kHdrAt + kHdrBytes + 16;  // ... and the tail from here on (a multiple of 16)
constexpr size_t kInCap = 192 * 1024;     // unread input kept on the device at most (unless the caller's room is larger)
constexpr size_t kGrow = 64 * 1024;       // (large streams) an attempt per this much new input at least
constexpr size_t kAheadMax = 128 * 1024;  // decoded ahead of the caller's window at most
constexpr size_t kHistory = 32768;        // output kept in front of a resume point (the longest distance)
constexpr size_t kTrimStep = 64 * 1024;   // buffers are moved to the front when this much can go
static_assert(kHead % 16 == 0, "the tail keeps the stream's 16-byte phase");

struct DevBuf {
    uint8_t* p = nullptr;
    size_t cap = 0;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    // grows to at least n bytes, keeping the first `keep` bytes.  (The streaming buffers are bounded by the window,
    // the look-ahead and kInCap: they are sized in 64 KiB steps and stop growing after the first few calls.)
    hipError_t reserve(size_t n, size_t keep, hipStream_t st) {
        if (n <= cap) return hipSuccess;
        size_t want = std::max<size_t>((n + 65535) & ~(size_t)65535, 4096);
        if (n <= 4096) want = 4096;
        uint8_t* q = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&q), want);
        if (e != hipSuccess) return e;
        if (keep && p) {
            e = (hipError_t)fdh_launch_copy_lines(q, p, std::min(keep, cap), st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);  // (the old buffer is freed below)
            if (e != hipSuccess) {
                (void)hipFree(q);
                return e;
            }
        }
        if (p) (void)hipFree(p);
        p = q;
        cap = want;
        return hipSuccess;
    }
    // moves [from, from + n) to [to, to + n), to < from (both multiples of 16; whole 16-byte lines are copied: the buffers
    // have 16 bytes of slack): in pieces that do not overlap their destination
    hipError_t move_down(size_t to, size_t from, size_t n, hipStream_t st) {
        const size_t step = from - to;
        for (size_t done = 0; done < n; done += step) {
            const hipError_t e = (hipError_t)fdh_launch_copy_lines(p + to + done, p + from + done, std::min(step, n - done), st);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
};

int fail(int code, const std::string& msg) {
    fdh_set_last_error(msg.c_str());
    return code;
}

}  // namespace

struct fdh_decompressor {
    DevBuf in, out, meta;
    // Everything the object does on the device -- copies, the decode kernels -- is ordered on ONE stream of its own, so
    // that it neither waits for nor holds up the caller's other work on the device.
    hipStream_t stream = nullptr;
    ~fdh_decompressor() {
        if (stream) (void)hipStreamDestroy(stream);
    }
    // ---- input: the device holds [kHead, kHead + in_total - tail_base) = stream bytes [tail_base, in_total) ----
    size_t in_total = 0;      // bytes of the stream received so far
    size_t tail_base = 0;     // stream offset of the first byte kept (a multiple of 16)
    size_t hdr_src = ~(size_t)0;  // stream offset (a multiple of 16) of the bytes copied to [kHdrAt, kHdrAt + kHdrBytes); ~0: none
    bool first_saved = false; // [0, 16) holds the stream's first bytes
    size_t attempted_in = 0;  // in_total at the last decode attempt
    // ---- output: the device slot holds stream positions [out_base, ...) ----
    size_t out_base = 0;      // a multiple of 16
    size_t delivered = 0;     // output bytes handed to the caller so far
    bool ignore_adler = false;
    bool done = false;
    bool output_limited = false;  // the last attempt stopped because the caller's buffer was full
    bool tried = false;
    uint32_t error = 0;       // sticky DecompressionError (status code), 0 = none
    int device = 0;
    // What the last attempt left in the device slot BEYOND what the caller could take: an attempt
    // decodes ahead and the calls that follow are served from that prefix without decoding anything.
    size_t ahead_have = 0;    // valid decoded prefix of the stream (a stream position; the slot holds it from out_base on)
    size_t ahead_in = 0;      // in_total it was decoded from
    uint32_t ahead_st = 0;    // the status of that attempt (for ITS slot)
    uint64_t attempts = 0;    // decode attempts so far (introspection)
    size_t ahead_bad_cap = 0; // an attempt that decoded ahead into a slot this large met a hard error (0: none):
    size_t ahead_bad_in = 0;  // ... with this much input received; later slots stay below it until more input arrives
    // Where the last attempt that ran out of input or room stopped (fdh_inflate_batch_resumable), in STREAM
    // coordinates: the next attempt goes on from there.
    bool res_valid = false;
    uint64_t res_hdr_bit = 0, res_bit = 0;
    size_t res_out = 0;
    uint32_t res_adler = 0, res_step = 0;
    uint32_t lz_unknown = 0;  // attempts in a row whose resume point came without its step state (below)
    bool stalled = false;     // the last attempt moved neither the resume point nor the decoded prefix (the input bound yields then)
    uint64_t decoded = 0;     // output bytes decoded by all attempts together (introspection: N for a stream of N bytes
                              // that is never decoded twice)
    size_t peak_bytes = 0;    // the most device memory the three buffers have held together
    void note_bytes() { peak_bytes = std::max(peak_bytes, in.cap + out.cap + meta.cap); }
};

extern "C" {

fdh_decompressor* fdh_decompressor_new(void) {
    fdh_decompressor* d = new (std::nothrow) fdh_decompressor();
    if (d && hipGetDevice(&d->device) != hipSuccess) d->device = 0;
    return d;
}

void fdh_decompressor_free(fdh_decompressor* d) { delete d; }

void fdh_decompressor_ignore_adler32(fdh_decompressor* d) {
    if (d) d->ignore_adler = true;
}

int fdh_decompressor_is_done(const fdh_decompressor* d) { return d && d->done ? 1 : 0; }

uint64_t fdh_decompressor_attempts(const fdh_decompressor* d) { return d ? d->attempts : 0; }

uint64_t fdh_decompressor_decoded_bytes(const fdh_decompressor* d) { return d ? d->decoded : 0; }

uint64_t fdh_decompressor_device_bytes(const fdh_decompressor* d) { return d ? (uint64_t)d->peak_bytes : 0; }

int fdh_decompressor_read(fdh_decompressor* d, const uint8_t* input, size_t input_len, uint8_t* output,
                          size_t output_len, size_t output_position, size_t* consumed, size_t* produced,
                          uint32_t* stream_status) {
    if (!d || !consumed || !produced || !stream_status) return fail(FDH_ERR_INVALID_ARGUMENT, "null pointer");
    *consumed = 0;
    *produced = 0;
    *stream_status = FDH_STREAM_OK;
    if (d->done) return FDH_SUCCESS;  // src/decompress.rs:185-187: (0, 0) once Done
    if (output_position > output_len)  // the reference panics here (src/decompress.rs:189)
        return fail(FDH_ERR_INVALID_ARGUMENT, "output_position is out of bounds");
    if (d->error) {
        *stream_status = d->error;
        return FDH_SUCCESS;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(FDH_ERR_NO_DEVICE, "no HIP device: fdeflate_hip has no CPU fallback");
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? FDH_ERR_OUT_OF_MEMORY : FDH_ERR_HIP,       \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)
    int prev_dev = 0;
    HIP_TRY(hipGetDevice(&prev_dev));
    if (prev_dev != d->device) HIP_TRY(hipSetDevice(d->device));
    struct Restore {
        int dev, cur;
        ~Restore() {
            if (dev != cur) (void)hipSetDevice(dev);
        }
    } restore{prev_dev, d->device};
    if (!d->stream) HIP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    hipStream_t const sq = d->stream;
    // (FDH_STREAM_NO_RESUME=1: every attempt from the first byte, as in rounds 1-3 -- for A/B measurements; nothing
    //  can be dropped then, and all input is taken)
    static const bool no_resume = std::getenv("FDH_STREAM_NO_RESUME") != nullptr;
    static const bool trace = std::getenv("FDH_STREAM_TRACE") != nullptr;  // one line per attempt on stderr
    static const bool no_lz = std::getenv("FDH_STREAM_NO_LZ") != nullptr;  // (A/B: every attempt without the LZ-window kernel)

    const size_t room = output_len - output_position;
    // take the input: all of it, unless enough is waiting unread on the device already (src/decompress.rs:167-170:
    // "the input is fully consumed" is one of three post-conditions; "the output is full" is what holds otherwise)
    size_t accept = input_len;
    if (!no_resume) {
        const size_t unread = d->in_total - std::min<size_t>(d->in_total, d->res_valid ? (size_t)(d->res_bit >> 3) : 0);
        const size_t bound = std::min<size_t>(std::max(kInCap, room), 32u << 20);
        accept = unread >= bound ? 0 : std::min(input_len, bound - unread);
        // (round 6) The bound is there so that the unread input goes DOWN before more is taken; an attempt that moved
        // nothing -- no point came back, or one without its step state even from the tile decoders, and every decoded
        // byte had been delivered -- would be repeated on the same bytes for ever if the input stayed refused: the call
        // would return (0, 0), not done, no error, again and again.  More input is what can move it: it is taken.
        if (accept == 0 && input_len != 0 && d->stalled) accept = std::min(input_len, bound);
    }
    if (accept) {
        const size_t have = d->in_total - d->tail_base;
        if (have + accept >= (1ull << 31)) return fail(FDH_ERR_INVALID_ARGUMENT, "stream too large (>= 2 GiB on the device)");
        HIP_TRY(d->in.reserve(kHead + have + accept + 16, kHead + have, sq));
        HIP_TRY(hipMemcpyAsync(d->in.p + kHead + have, input, accept, hipMemcpyHostToDevice, sq));
        HIP_TRY(hipStreamSynchronize(sq));  // (the caller's buffer is the caller's again when this returns)
        d->in_total += accept;
        d->note_bytes();
    }
    *consumed = accept;

    // Hands n bytes of the decoded prefix to the caller and says what state that leaves: the prefix
    // used up -> the status of the attempt that made it applies; otherwise the caller's buffer is full
    // "but there are more bytes to output".
    auto deliver = [&](size_t n) -> int {
        if (n) {
            hipError_t e_ = hipMemcpyAsync(output + output_position, d->out.p + (d->delivered - d->out_base), n, hipMemcpyDeviceToHost, sq);
            if (e_ == hipSuccess) e_ = hipStreamSynchronize(sq);
            if (e_ != hipSuccess) return fail(FDH_ERR_HIP, std::string("hipMemcpy(decoded bytes): ") + hipGetErrorString(e_));
            d->delivered += n;
            *produced = n;
        }
        d->output_limited = false;
        if (d->delivered < d->ahead_have) {
            d->output_limited = true;
            return FDH_SUCCESS;
        }
        switch (d->ahead_st) {
            case FDH_STREAM_OK:
                d->done = true;
                break;
            case FDH_OUTPUT_TOO_LARGE:
                d->output_limited = true;  // "the output is full but there are more bytes to output"
                break;
            case FDH_INSUFFICIENT_INPUT:
                // Ok with is_done() == false: wait for more input (input that arrived after the prefix was
                // decoded may hold more bytes for a window that is full now)
                d->output_limited = d->in_total != d->ahead_in && n == room;
                break;
            default:
                d->error = d->ahead_st;    // a DecompressionError: sticky, like the reference's poisoned state
                *stream_status = d->ahead_st;
                break;
        }
        return FDH_SUCCESS;
    };
    const size_t avail = d->ahead_have > d->delivered ? d->ahead_have - d->delivered : 0;
    if (d->tried && avail > 0) {
        // the decoded prefix fills the window, or holds all the reference could have produced from this input
        const bool final_prefix = d->in_total == d->ahead_in && d->ahead_st != FDH_OUTPUT_TOO_LARGE;
        if (avail >= room || final_prefix) return deliver(std::min(room, avail));
    }
    // is a decode attempt worth it?  (nothing new and not output-limited -> no)
    const bool grew = d->in_total != d->attempted_in;
    const bool flush = input_len == 0;
    const bool refused = accept < input_len;  // the unread input has reached its bound: it must go down
    // (an attempt only decodes what is new, but costs a few launches and a round trip whatever it decodes: above
    //  kAlwaysBelow received bytes one is made for every 1/8 the stream has grown, and at least every kGrow)
    bool attempt = !d->tried || d->output_limited || refused ||
                   (grew && (flush || d->in_total < kAlwaysBelow || d->in_total >= d->attempted_in + std::min(d->attempted_in / 8, kGrow)));
    if (d->tried && d->output_limited && room == 0) attempt = false;  // still nowhere to put a byte
    if (!attempt) return FDH_SUCCESS;  // (nothing of the prefix is left over here: that state is output-limited)

    const bool go_on = !no_resume && d->res_valid;
    const uint64_t res_bit_before = d->res_valid ? d->res_bit : 0;
    static const bool no_trim = std::getenv("FDH_STREAM_NO_TRIM") != nullptr;  // (A/B: resume points, but nothing is dropped)
    // ---- what the resume point no longer needs goes: output in front of its history (and of what the caller has not
    //      taken yet), input in front of its bit ----
    if (go_on && !no_trim) {
        const size_t hist = d->res_out > kHistory ? d->res_out - kHistory : 0;
        const size_t ob = std::min(d->delivered, hist) & ~(size_t)15;
        if (ob >= d->out_base + kTrimStep) {
            const size_t keep_end = std::max(d->ahead_have, d->res_out);  // (what has been decoded ahead of the caller stays)
            if (keep_end > ob) HIP_TRY(d->out.move_down(0, ob - d->out_base, std::min(keep_end - ob, d->out.cap - (ob - d->out_base)), sq));
            d->out_base = ob;
        }
        const size_t bit_byte = (size_t)(d->res_bit >> 3);
        const size_t tb = (bit_byte > 16 ? bit_byte - 16 : 0) & ~(size_t)15;
        if (tb >= d->tail_base + kTrimStep && tb <= d->in_total) {
            if (!d->first_saved) {  // (tail_base is 0 here: the stream's first bytes are the tail's)
                HIP_TRY((hipError_t)fdh_launch_copy_lines(d->in.p, d->in.p + kHead, 16, sq));
                d->first_saved = true;
            }
            const size_t drop = tb - d->tail_base;
            HIP_TRY(d->in.move_down(kHead, kHead + drop, d->in_total - tb, sq));
            d->tail_base = tb;
        }
    }
    // ---- the slot: what the caller can take now, and some way ahead ----
    const size_t cap_exact = d->delivered + room;
    size_t cap = std::min(std::max<size_t>(cap_exact, 2 * d->delivered + 65536), cap_exact + kAheadMax);
    // a hard error was met decoding ahead of the caller with this very input: it lies somewhere below that
    // slot's end, so stay with the exact slot until the window gets there (not two decodes per call)
    if (d->ahead_bad_cap != 0 && d->ahead_bad_in == d->in_total && cap > cap_exact) {
        // ... halving the distance to it each time: O(log) attempts up to the damage, as for a good stream
        const size_t mid = cap_exact + (d->ahead_bad_cap > cap_exact ? (d->ahead_bad_cap - cap_exact) / 2 : 0);
        cap = mid >= cap_exact + 65536 ? std::min(cap, mid) : cap_exact;
    }
    if (cap_exact - d->out_base > 0xFFFFFFF0ull) return fail(FDH_ERR_INVALID_ARGUMENT, "more than 4 GiB of room");
    cap = std::min<size_t>(cap, d->out_base + 0xFFFFFFF0ull);
    HIP_TRY(d->meta.reserve(64, 0, sq));
    HIP_TRY(d->in.reserve(kHead + 16, kHead + (d->in_total - d->tail_base), sq));  // (empty only before the first byte arrives)
    uint32_t host_res[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // the resume point in the coordinates of the device buffers
    fdh_resume_point rec = {0, 0, 0, 0};
    if (go_on) {
        const size_t bit_byte = (size_t)(d->res_bit >> 3), hdr_byte = (size_t)(d->res_hdr_bit >> 3);
        const uint64_t bit_dev = 8ull * (kHead + (bit_byte - d->tail_base)) + (d->res_bit & 7);
        uint64_t hdr_dev = bit_dev;  // a point AT a block header: the header is where the tail starts
        if (d->res_bit != d->res_hdr_bit) {
            if (d->tail_base == 0) {
                hdr_dev = 8ull * (kHead + hdr_byte) + (d->res_hdr_bit & 7);  // nothing dropped yet: the header is in its place
            } else {
                hdr_dev = 8ull * (kHdrAt + (hdr_byte - d->hdr_src)) + (d->res_hdr_bit & 7);
            }
        }
        rec.header_bit = (uint32_t)hdr_dev | (d->res_step << 30);
        rec.bit = (uint32_t)bit_dev;
        rec.out_bytes = (uint32_t)(d->res_out - d->out_base);
        rec.adler32 = d->res_adler;
    }
    // (a stream nothing of which has been dropped starts at kHead; one that is taken up again is seen from offset 0:
    //  zlib header, header copy, tail)
    const bool whole = d->tail_base == 0;
    const size_t in_at = whole ? kHead : 0;
    const size_t in_dev_len = (whole ? 0 : kHead) + (d->in_total - d->tail_base);
    if (go_on && whole) {  // (positions were computed for offset 0)
        rec.header_bit -= 8u * (uint32_t)kHead;
        rec.bit -= 8u * (uint32_t)kHead;
    }
    // A resume point must know its place among the reference's table steps (a literal that is the second of a pair is
    // not where a step starts): the serial decoder, asked to take a stream up at a point that does not, may have to go
    // back to a point that does -- in the end to the stream's first byte (inflate.hip, general_one), which is no longer
    // here.  The tile decoders keep track of the steps, the LZ-window kernel does not: when nothing behind it gets far
    // enough to tell -- a stretch of nothing but literals whose codes pair up -- the point comes back with step state
    // 0.  Such an attempt is made again without the LZ-window kernel (3-4 x slower, tools/streamtime.py) -- and, for an
    // attempt from the stream's first byte, without the small-table kernel, whose points are stepless as well --, and
    // a stream that does it twice in a row goes without from then on.
    bool without_lz = no_lz || d->lz_unknown >= 2;
    for (;;) {
        {   // (what lies in front of the resume point, and what has been decoded ahead of the caller, stays)
            const size_t keep_to = std::max(d->ahead_have, d->res_valid ? d->res_out : 0);
            const size_t keep = std::min<size_t>(keep_to > d->out_base ? keep_to - d->out_base : 0, d->out.cap);
            hipError_t re = d->out.reserve(cap - d->out_base + 16, keep, sq);
            if (re == hipErrorOutOfMemory && cap != cap_exact) {  // no room to decode ahead: the exact slot may still fit
                (void)hipGetLastError();
                cap = cap_exact;
                re = d->out.reserve(cap - d->out_base + 16, keep, sq);
            }
            HIP_TRY(re);
            d->note_bytes();
        }
        uint64_t* m = reinterpret_cast<uint64_t*>(d->meta.p);
        uint64_t meta[8] = {0, (uint64_t)in_dev_len, 0, (uint64_t)(cap - d->out_base), 0, 0, 0, 0};
        if (go_on) std::memcpy(&meta[6], &rec, sizeof(rec));
        HIP_TRY(hipMemcpyAsync(m, meta, sizeof(meta), hipMemcpyHostToDevice, sq));
        HIP_TRY(hipStreamSynchronize(sq));  // (`meta` is a stack array)
        uint32_t* res = reinterpret_cast<uint32_t*>(m + 4);
        int rc = fdh_inflate_batch_resumable(d->in.p + in_at, m, d->out.p, m + 2, res, res + 1, res + 2, 1,
                                             (d->ignore_adler ? FDH_FLAG_IGNORE_ADLER32 : 0u) | (go_on ? FDH_FLAG_RESUME_IN : 0u) | (without_lz ? (FDH_FLAG_NO_LZ | FDH_FLAG_NO_FAST_GENERAL) : 0u),
                                             reinterpret_cast<fdh_resume_point*>(m + 6), sq);
        if (rc != FDH_SUCCESS) return rc;
        HIP_TRY(hipMemcpyAsync(host_res, res, sizeof(host_res), hipMemcpyDeviceToHost, sq));
        HIP_TRY(hipStreamSynchronize(sq));
        d->attempts++;
        const uint32_t st1 = host_res[1];
        if (trace)
            std::fprintf(stderr, "attempt %llu: in_total %zu tail_base %zu hdr_src %zd out_base %zu cap %zu go_on %d whole %d rec{h %u s %u b %u o %u} res{h %llu b %llu o %zu} -> st %u len %u got{h %u s %u b %u o %u}\n",
                         (unsigned long long)d->attempts, d->in_total, d->tail_base, (ssize_t)d->hdr_src, d->out_base, cap, (int)go_on, (int)whole,
                         rec.header_bit & 0x3FFFFFFFu, rec.header_bit >> 30, rec.bit, rec.out_bytes, (unsigned long long)d->res_hdr_bit,
                         (unsigned long long)d->res_bit, d->res_out, st1, host_res[0], host_res[4] & 0x3FFFFFFFu, host_res[4] >> 30, host_res[5], host_res[6]);
        const bool classified = st1 == FDH_STREAM_OK || st1 == FDH_WRONG_CHECKSUM || st1 == FDH_OUTPUT_TOO_LARGE ||
                                st1 == FDH_INSUFFICIENT_INPUT;
        if (!no_resume && (st1 == FDH_OUTPUT_TOO_LARGE || st1 == FDH_INSUFFICIENT_INPUT)) {
            const bool stepless = host_res[4] != 0 && (host_res[4] >> 30) == 0 && (host_res[4] & 0x3FFFFFFFu) != host_res[5];
            if (!without_lz) {
                d->lz_unknown = stepless ? d->lz_unknown + 1 : 0;
                if (stepless) {
                    without_lz = true;
                    continue;
                }
            }
        }
        if (classified) {
            const size_t from = go_on ? d->res_out : 0;
            const size_t to = d->out_base + host_res[0];
            if (to > from) d->decoded += to - from;
        }
        if (st1 == FDH_OUTPUT_TOO_LARGE || st1 == FDH_INSUFFICIENT_INPUT) {
            fdh_resume_point got;
            std::memcpy(&got, &host_res[4], sizeof(got));
            // (none: the one this attempt started from still stands.  So it does should a point come back without its
            //  step state even from the tile decoders: the next attempt decodes that stretch again.)
            const bool at_header = (got.header_bit & 0x3FFFFFFFu) == got.bit;
            if (got.header_bit != 0 && ((got.header_bit >> 30) != 0 || at_header || no_resume)) {
                // back to stream coordinates: a position in the tail, or (the header only) in the header's copy
                auto to_stream = [&](uint64_t dev_bit) -> uint64_t {
                    const uint64_t b = dev_bit + 8ull * in_at;  // as seen from offset 0 of the device buffer
                    if (b >= 8ull * kHead) return 8ull * d->tail_base + (b - 8ull * kHead);
                    return 8ull * d->hdr_src + (b - 8ull * kHdrAt);
                };
                d->res_hdr_bit = to_stream(got.header_bit & 0x3FFFFFFFu);
                d->res_step = got.header_bit >> 30;
                d->res_bit = to_stream(got.bit);
                d->res_out = d->out_base + got.out_bytes;
                d->res_adler = got.adler32;
                d->res_valid = true;
                // the header of its block travels with the point from now on (the tail will lose it)
                const size_t hb0 = (size_t)(d->res_hdr_bit >> 3) & ~(size_t)15;
                if (!no_resume && d->res_bit != d->res_hdr_bit && hb0 != d->hdr_src) {
                    const size_t n = std::min(kHdrBytes, d->in_total - hb0);
                    HIP_TRY((hipError_t)fdh_launch_copy_lines(d->in.p + kHdrAt, d->in.p + kHead + (hb0 - d->tail_base), n, sq));
                    d->hdr_src = hb0;
                }
            }
        }
        // a hard error somewhere in the part decoded ahead: the caller must not hear of it before its
        // window gets there -- decode again into exactly what the caller can take
        if (!classified && cap != cap_exact) {
            d->ahead_bad_cap = cap;
            d->ahead_bad_in = d->in_total;
            cap = cap_exact;
            continue;
        }
        break;
    }
    const uint32_t st = host_res[1];
    d->tried = true;
    d->attempted_in = d->in_total;
    d->stalled = false;
    size_t have = 0;  // valid prefix of the decoded stream (a stream position)
    if (st == FDH_STREAM_OK || st == FDH_WRONG_CHECKSUM || st == FDH_OUTPUT_TOO_LARGE || st == FDH_INSUFFICIENT_INPUT) {
        have = std::min<size_t>(d->out_base + host_res[0], cap);
    }
    d->ahead_have = std::max(have, d->delivered);
    d->ahead_in = d->in_total;
    d->ahead_st = st;
    d->stalled = (st == FDH_INSUFFICIENT_INPUT || st == FDH_OUTPUT_TOO_LARGE) && d->ahead_have == d->delivered &&
                 (!d->res_valid || d->res_bit == res_bit_before);
    return deliver(std::min(room, d->ahead_have - d->delivered));
#undef HIP_TRY
}

}  // extern "C"
