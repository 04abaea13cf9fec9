// stream_decompressor.cpp -- `Decompressor::read` (reference src/decompress.rs:158-337) on top of
// the batch engine.
//
// The reference decoder is resumable at any input / output split (State :84-93, BitBuffer,
// QueuedOutput :1067-1070).  The GPU engine is one-shot, so this object keeps the streaming
// CONTRACT on the host and lets the device do every bit of decoding:
//
//   * input handed to `read` is appended to a device-resident copy of the stream (consumed =
//     input_len, always: "the input is fully consumed" is the post-condition we pick);
//   * the stream-so-far is decoded by fdh_inflate_batch (batch of one) into a device slot that holds what
//     the caller can take now (bytes delivered so far + the room left in `output`) and, decoding ahead,
//     up to as much again + 64 KiB (see below); the one-shot classification (src/decompress.rs:1126-1139)
//     of the attempt whose prefix is used up says which post-condition holds: Ok -> done; OutputTooLarge -> "the output is full but there are more
//     bytes"; InsufficientInput -> the engine reports how many bytes the reference had produced when
//     it ran dry, and exactly those are delivered;
//   * only output[output_position ..] is written, never more than the room, and the bytes in
//     front of output_position are not needed (the LZ77 history lives in the device slot);
//   * an attempt goes on where the last one stopped: fdh_inflate_batch_resumable hands back a resume
//     point (bit position at the start of one of the reference's decoding steps, block header, output
//     bytes, Adler-32) for a stream that ran out of input or room, and takes the stream up there in the
//     next call -- the counterpart of the reference's State / BitBuffer / QueuedOutput, kept on the device.
//
// Draining through a small output window: an attempt decodes AHEAD of what the caller can take (slot =
// twice what has been delivered + 64 KiB) and the following calls are served from that prefix, so a
// stream of N bytes costs O(N) decoded bytes however small the window (round 2: O(N^2 / window)).  A
// hard error in the part decoded ahead is not reported early: the attempt is repeated with the exact
// slot, whose classification is the reference's.
// An attempt costs a few kernel launches and a round trip to the device whatever it decodes;
// above kAlwaysBelow buffered bytes an attempt is made only once the stream has grown by 1/8 (or by kAlwaysBelow) --
// or when the caller passes an empty `input`, which is how both the reference's test harness
// (src/decompress/tests/test_utils.rs:70-74: chunk size 0 once the input is exhausted) and the png
// crate's finish loop ask for whatever can still be produced.
//
// There is no CPU decode path here: without a GPU every call returns FDH_ERR_NO_DEVICE.
#include "../../include/fdeflate_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

extern "C" void fdh_set_last_error(const char* msg);

namespace {

constexpr size_t kAlwaysBelow = 256 * 1024;

struct DevBuf {
    uint8_t* p = nullptr;
    size_t cap = 0;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    // grows to at least n bytes, keeping the first `keep` bytes
    hipError_t reserve(size_t n, size_t keep) {
        if (n <= cap) return hipSuccess;
        size_t want = std::max(n, cap * 2);
        want = std::max<size_t>(want, 4096);
        uint8_t* q = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&q), want);
        if (e != hipSuccess) return e;
        if (keep) {
            e = hipMemcpy(q, p, keep, hipMemcpyDeviceToDevice);
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
};

int fail(int code, const std::string& msg) {
    fdh_set_last_error(msg.c_str());
    return code;
}

}  // namespace

struct fdh_decompressor {
    DevBuf in, out, meta;
    size_t in_len = 0;        // bytes of the stream on the device
    size_t attempted_in = 0;  // in_len at the last decode attempt
    size_t delivered = 0;     // output bytes handed to the caller so far
    bool ignore_adler = false;
    bool done = false;
    bool output_limited = false;  // the last attempt stopped because the caller's buffer was full
    bool tried = false;
    uint32_t error = 0;       // sticky DecompressionError (status code), 0 = none
    int device = 0;
    // What the last attempt left in the device slot BEYOND what the caller could take: an attempt
    // decodes ahead (into a slot of twice what has been delivered + 64 KiB), and the calls that follow
    // are served from that prefix without decoding anything -- draining a stream of N bytes through a
    // small window then costs O(N) decoded bytes, not O(N^2 / window).
    size_t ahead_have = 0;    // valid decoded prefix in the device slot
    size_t ahead_in = 0;      // in_len it was decoded from
    uint32_t ahead_st = 0;    // the status of that attempt (for ITS slot)
    uint64_t attempts = 0;    // decode attempts so far (introspection)
    size_t ahead_bad_cap = 0; // an attempt that decoded ahead into a slot this large met a hard error (0: none):
    size_t ahead_bad_in = 0;  // ... with this much input buffered; later slots stay below it until more input arrives
    // Where the last attempt that ran out of input or room stopped (fdh_inflate_batch_resumable): the next attempt
    // goes on from there -- the bytes in front of it stay in the device slot -- instead of at the first byte.
    fdh_resume_point resume = {0, 0, 0, 0};
    uint64_t decoded = 0;     // output bytes decoded by all attempts together (introspection: N for a stream of N bytes
                              // that is never decoded twice)
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

    // take the input (the whole of it: that is the post-condition this implementation offers)
    if (input_len) {
        if (d->in_len + input_len >= (1ull << 31)) return fail(FDH_ERR_INVALID_ARGUMENT, "stream too large (>= 2 GiB)");
        HIP_TRY(d->in.reserve(d->in_len + input_len + 16, d->in_len));
        HIP_TRY(hipMemcpy(d->in.p + d->in_len, input, input_len, hipMemcpyHostToDevice));
        d->in_len += input_len;
    }
    *consumed = input_len;

    const size_t room = output_len - output_position;
    // Hands n bytes of the decoded prefix to the caller and says what state that leaves: the prefix
    // used up -> the status of the attempt that made it applies; otherwise the caller's buffer is full
    // "but there are more bytes to output".
    auto deliver = [&](size_t n) -> int {
        if (n) {
            hipError_t e_ = hipMemcpy(output + output_position, d->out.p + d->delivered, n, hipMemcpyDeviceToHost);
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
                d->output_limited = d->in_len != d->ahead_in && n == room;
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
        const bool final_prefix = d->in_len == d->ahead_in && d->ahead_st != FDH_OUTPUT_TOO_LARGE;
        if (avail >= room || final_prefix) return deliver(std::min(room, avail));
    }
    // is a decode attempt worth it?  (nothing new and not output-limited -> no)
    const bool grew = d->in_len != d->attempted_in;
    const bool flush = input_len == 0;
    // (an attempt only decodes what is new, but costs a few launches and a round trip whatever it decodes: above
    //  kAlwaysBelow buffered bytes one is made for every 1/8 the stream has grown, and at least every kAlwaysBelow)
    bool attempt = !d->tried || d->output_limited || (grew && (flush || d->in_len < kAlwaysBelow ||
                                                               d->in_len >= d->attempted_in + std::min(d->attempted_in / 8, kAlwaysBelow)));
    if (d->tried && d->output_limited && room == 0) attempt = false;  // still nowhere to put a byte
    if (!attempt) return FDH_SUCCESS;  // (nothing of the prefix is left over here: that state is output-limited)

    const size_t cap_exact = std::min<size_t>(d->delivered + room, 0xFFFFFFF0ull);
    size_t cap = std::min<size_t>(std::max<size_t>(cap_exact, 2 * d->delivered + 65536), 0xFFFFFFF0ull);
    // a hard error was met decoding ahead of the caller with this very input: it lies somewhere below that
    // slot's end, so stay with the exact slot until the window gets there (not two decodes per call)
    if (d->ahead_bad_cap != 0 && d->ahead_bad_in == d->in_len && cap > cap_exact) {
        // ... halving the distance to it each time: O(log) attempts up to the damage, as for a good stream
        const size_t mid = cap_exact + (d->ahead_bad_cap > cap_exact ? (d->ahead_bad_cap - cap_exact) / 2 : 0);
        cap = mid >= cap_exact + 65536 ? std::min(cap, mid) : cap_exact;
    }
    HIP_TRY(d->meta.reserve(64, 0));
    // a hipMalloc'd buffer can be empty only before the first byte arrives
    HIP_TRY(d->in.reserve(16, d->in_len));
    uint32_t host_res[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (;;) {
        {   // (what lies in front of the resume point, and what has been decoded ahead of the caller, stays)
            const size_t keep = std::min<size_t>(std::max<size_t>(d->ahead_have, d->resume.out_bytes), d->out.cap);
            hipError_t re = d->out.reserve(cap + 16, keep);
            if (re == hipErrorOutOfMemory && cap != cap_exact) {  // no room to decode ahead: the exact slot may still fit
                (void)hipGetLastError();
                cap = cap_exact;
                re = d->out.reserve(cap + 16, keep);
            }
            HIP_TRY(re);
        }
        // the attempt goes on from where the last one stopped, if that lies inside this slot
        // (FDH_STREAM_NO_RESUME=1: every attempt from the first byte, as in rounds 1-3 -- for A/B measurements)
        static const bool no_resume = std::getenv("FDH_STREAM_NO_RESUME") != nullptr;
        const bool go_on = !no_resume && d->resume.header_bit != 0 && d->resume.out_bytes <= cap;
        uint64_t meta[8] = {0, (uint64_t)d->in_len, 0, (uint64_t)cap, 0, 0, 0, 0};
        if (go_on) std::memcpy(&meta[6], &d->resume, sizeof(d->resume));
        HIP_TRY(hipMemcpy(d->meta.p, meta, sizeof(meta), hipMemcpyHostToDevice));
        uint64_t* m = reinterpret_cast<uint64_t*>(d->meta.p);
        uint32_t* res = reinterpret_cast<uint32_t*>(m + 4);
        int rc = fdh_inflate_batch_resumable(d->in.p, m, d->out.p, m + 2, res, res + 1, res + 2, 1,
                                             (d->ignore_adler ? FDH_FLAG_IGNORE_ADLER32 : 0u) | (go_on ? FDH_FLAG_RESUME_IN : 0u),
                                             reinterpret_cast<fdh_resume_point*>(m + 6), nullptr);
        if (rc != FDH_SUCCESS) return rc;
        HIP_TRY(hipMemcpy(host_res, res, sizeof(host_res), hipMemcpyDeviceToHost));  // synchronises the null stream
        d->attempts++;
        const uint32_t st1 = host_res[1];
        if (st1 == FDH_STREAM_OK || st1 == FDH_WRONG_CHECKSUM || st1 == FDH_OUTPUT_TOO_LARGE || st1 == FDH_INSUFFICIENT_INPUT) {
            const size_t from = go_on ? d->resume.out_bytes : 0;
            if (host_res[0] > from) d->decoded += host_res[0] - from;
        }
        if (st1 == FDH_OUTPUT_TOO_LARGE || st1 == FDH_INSUFFICIENT_INPUT) {
            fdh_resume_point got;
            std::memcpy(&got, &host_res[4], sizeof(got));
            if (got.header_bit != 0) d->resume = got;  // (none: the one this attempt started from still stands)
        }
        const bool classified = st1 == FDH_STREAM_OK || st1 == FDH_WRONG_CHECKSUM || st1 == FDH_OUTPUT_TOO_LARGE ||
                                st1 == FDH_INSUFFICIENT_INPUT;
        // a hard error somewhere in the part decoded ahead: the caller must not hear of it before its
        // window gets there -- decode again into exactly what the caller can take
        if (!classified && cap != cap_exact) {
            d->ahead_bad_cap = cap;
            d->ahead_bad_in = d->in_len;
            cap = cap_exact;
            continue;
        }
        break;
    }
    const uint32_t st = host_res[1];
    d->tried = true;
    d->attempted_in = d->in_len;
    size_t have = 0;  // valid prefix of the decoded stream in the device slot
    if (st == FDH_STREAM_OK || st == FDH_WRONG_CHECKSUM || st == FDH_OUTPUT_TOO_LARGE || st == FDH_INSUFFICIENT_INPUT) {
        have = std::min<size_t>(host_res[0], cap);
    }
    d->ahead_have = std::max(have, d->delivered);
    d->ahead_in = d->in_len;
    d->ahead_st = st;
    return deliver(std::min(room, d->ahead_have - d->delivered));
    return FDH_SUCCESS;
#undef HIP_TRY
}

}  // extern "C"
