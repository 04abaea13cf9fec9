// inflate.hip -- batched zlib decode kernels (gfx950).  One independent stream per wavefront.
//
// fdh_launch_inflate runs, back to back on the caller's stream (each kernel finishes what it can
// and leaves the rest PENDING in the status word):
//
//   inflate_segments_kernel      streams that start with the ultra-fast encoder's fixed 53-byte +
//                                5-bit prefix (reference src/compress/ultrafast.rs:82-88: zlib header
//                                + one final dynamic block header): segment-parallel, one shared
//                                table copy per workgroup (inflate_segments.h).
//   inflate_canon_kernel         canonical streams the segment kernel left over: 8 wavefronts per
//                                workgroup sharing ONE copy of the tables (built once per device from
//                                the prefix bytes by canon_build_kernel), tile + serial decoders.
//   inflate_general_fast_kernel  any zlib stream (stored / fixed / dynamic, multi-block), private
//                                10-bit tables: 8 workgroups per CU; leaves exactness-guard cases.
//   inflate_general_kernel       the same with the reference's 12-bit tables; runs what is left.
//
// Exactness guard: results other than Ok that depend on how literals were paired by the
// double-literal table at the very end of a truncated input, and all hard errors, are re-derived
// by the symbol-serial decoder, whose pairing is the reference's (DESIGN.md "error parity").
#include <algorithm>
#include <mutex>
#include <cstddef>
#include "inflate_stream.h"
#include "inflate_lanes.h"
#include "inflate_segments.h"
#include "inflate_seg2.h"
#include "inflate_lz.h"

namespace fdh {

#include "uf_table_data.inc"

constexpr uint32_t kPending = 0xFFFFFFFFu;         // not decoded yet, tiles allowed
constexpr uint32_t kPendingSerial = 0xFFFFFFFEu;   // not decoded yet, serial decoder only (resume[sid], when there is such an
                                                   // array, was written by whoever set this status: valid or all zero)
constexpr uint32_t kPendingResume = 0xFFFFFFFDu;   // not decoded to its end yet, tiles allowed: the LZ-window kernel left it at resume[sid]
constexpr uint32_t kCanonBits = 53 * 8 + 5;        // ultrafast.rs:87-88
constexpr int kCanonWaves = 8;
constexpr uint64_t kLaneMinStreams = ~0ull;  // opt-in only (FDH_FLAG_FORCE_LANES): 1 wave/SIMD at 64 Ki streams is no win

struct CanonTables {
    uint32_t lit[kLitSize];
    uint32_t dist[kDistSize];
    uint32_t eof[4];
    uint32_t hdr[16];  // the prefix as little-endian dwords (14 used)
    uint32_t len4[32]; // code lengths of the literals 0..255, 4 bits each (segment kernel: first literal of a step)
    // interval kernel: the symbols that are no literals (256 .. 285), decoded without a table in memory.
    // nl[0..16): per code length l: first such code (canonical, MSB first) | their number << 16 | index of
    // the first one in nl[32..] << 22.  nl[16] = shortest, nl[17] = longest such length.
    // nl[32..64): the symbols in (length, symbol) order: length base | extra bits << 9 | 1 << 12 for a
    // length symbol, 1 << 13 for end-of-block (256, and 286 / 287: reference src/tables.rs:100)
    uint32_t nl[64];
    uint32_t lit2[kLitSize];  // the interval kernel's table (seg2_entry_build), built once per device
    uint32_t status;   // build status (ST_OK expected)
};
__device__ CanonTables g_canon;

__device__ const uint8_t g_canon_header[64] = {
    kUfHeaderData[0],  kUfHeaderData[1],  kUfHeaderData[2],  kUfHeaderData[3],  kUfHeaderData[4],
    kUfHeaderData[5],  kUfHeaderData[6],  kUfHeaderData[7],  kUfHeaderData[8],  kUfHeaderData[9],
    kUfHeaderData[10], kUfHeaderData[11], kUfHeaderData[12], kUfHeaderData[13], kUfHeaderData[14],
    kUfHeaderData[15], kUfHeaderData[16], kUfHeaderData[17], kUfHeaderData[18], kUfHeaderData[19],
    kUfHeaderData[20], kUfHeaderData[21], kUfHeaderData[22], kUfHeaderData[23], kUfHeaderData[24],
    kUfHeaderData[25], kUfHeaderData[26], kUfHeaderData[27], kUfHeaderData[28], kUfHeaderData[29],
    kUfHeaderData[30], kUfHeaderData[31], kUfHeaderData[32], kUfHeaderData[33], kUfHeaderData[34],
    kUfHeaderData[35], kUfHeaderData[36], kUfHeaderData[37], kUfHeaderData[38], kUfHeaderData[39],
    kUfHeaderData[40], kUfHeaderData[41], kUfHeaderData[42], kUfHeaderData[43], kUfHeaderData[44],
    kUfHeaderData[45], kUfHeaderData[46], kUfHeaderData[47], kUfHeaderData[48], kUfHeaderData[49],
    kUfHeaderData[50], kUfHeaderData[51], kUfHeaderData[52], kUfHeaderData[53]};

struct InflateBatchArgs {
    const uint8_t* in;
    const uint64_t* in_off;
    uint8_t* out;
    const uint64_t* out_off;
    uint32_t* out_len;
    uint32_t* status;
    uint32_t* adler;
    uint64_t n;
    uint32_t flags;
    uint32_t only_pending;
    uint32_t* list;  // nullable: compacted ids of the PENDING streams ([0] = count, [2] / [3] = hand-out counters of
                     // the fast / the 12-bit general kernel, [4..] = ids)
    uint32_t* span_pool;   // nullable: kSpanSlots busy flags, then kSpanSlots match lists (span decoder scratch)
    uint32_t* lz_counter;  // hand-out counter of the LZ-window kernel (zeroed by the launcher)
    uint2* lz_ck;          // its items: 64 x kLzMaxPhases per wavefront (stream-ordered scratch)
    uint32_t* list_out;    // nullable: where the LZ-window kernel lists what it leaves ([0] = count, [4..] = ids)
    uint4* resume;         // nullable: per stream, where a kernel left it: {bit of the block header (0: nowhere), bit to go on
                           // from, output bytes decoded, their Adler-32} -- a ResumePoint.  Only ever read for a stream
                           // whose status says it was written (kPendingResume, kPendingSerial): no initialisation
    uint4* resume_out;     // nullable (fdh_inflate_batch_resumable): per stream, where a stream that ended InsufficientInput
                           // or OutputTooLarge can be taken up again (all zero: from its first byte); same layout
};
// Statuses and resume records are written by the kernel in front (or by an earlier kernel of the same call) and read
// here with a wavefront-uniform index: plain device-scope loads, whatever the compiler would have picked.
__device__ __forceinline__ uint32_t load_word(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint4 load_record(const uint4* p) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
    return make_uint4(load_word(w), load_word(w + 1), load_word(w + 2), load_word(w + 3));
}
// A status that promises a record (`kPendingResume`) over a record that reads all zero: counted per device
// (fdh_debug_lost_records) -- round 5 met it with scratch from the device's default memory pool (scratch_alloc) --
// and never taken for "start at the stream's first byte": the caller's own record, still in its array, is used.
__device__ unsigned int g_lost_records;
__device__ __forceinline__ ResumePoint unpack_record(const uint4 v) {
    ResumePoint rp;
    rp.hdr_bit = v.x & 0x3FFFFFFFu;
    rp.step = v.x >> 30;
    rp.bit = v.y;
    rp.opos = v.z;
    rp.adler = v.w;
    rp.valid = v.x != 0 ? 1u : 0u;
    if (rp.bit == rp.hdr_bit) rp.step = STEP_START;  // (a block header is where a step starts, whoever left the point)
    return rp;
}
// The resume point a kernel in front left for stream `sid`, whose status is `st` (if any).
__device__ __forceinline__ ResumePoint resume_point(const InflateBatchArgs& a, const uint64_t sid, const uint32_t st) {
    ResumePoint rp;
    rp.valid = 0;
    rp.step = 0;
    if (a.resume && a.only_pending && (st == kPendingResume || st == kPendingSerial) && !(a.flags & 0x4000u)) {
        rp = unpack_record(load_record(&a.resume[sid]));
        if (!rp.valid && st == kPendingResume) {  // the record is lost: not "from the first byte" -- from the caller's record
            if ((threadIdx.x & (kWave - 1)) == 0) atomicAdd(&g_lost_records, 1u);
            if ((a.flags & 0x8000u) && a.resume_out && a.resume_out != a.resume) rp = unpack_record(load_record(&a.resume_out[sid]));
        }
    }
    return rp;
}
// (a record keeps bit positions in 30 bits: streams of 128 MiB and more go without)
__device__ __forceinline__ uint4 resume_record(const ResumePoint& rp) {
    const bool fits = rp.valid && rp.bit < (1ull << 30) && rp.hdr_bit != 0;
    return fits ? make_uint4((uint32_t)rp.hdr_bit | (rp.step << 30), (uint32_t)rp.bit, rp.opos, rp.adler) : make_uint4(0, 0, 0, 0);
}
constexpr uint32_t kSpanSlots = 2048;  // > workgroups of the general kernel resident on one device (256 CUs x 5)

__device__ __forceinline__ StreamArgs stream_args(const InflateBatchArgs& a, uint64_t sid) {
    const uint64_t i0 = a.in_off[sid], i1 = a.in_off[sid + 1];
    const uint64_t o0 = a.out_off[sid], o1 = a.out_off[sid + 1];
    StreamArgs s;
    s.in = a.in + i0;
    s.in_len = i1 - i0;
    s.out = a.out + o0;
    uint64_t capacity = o1 - o0;
    s.cap = capacity > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)capacity;
    s.buf_lo = a.in;
    s.buf_hi = a.in + a.in_off[a.n];
    s.flags = a.flags;
    return s;
}

// A result the tile decoder may have classified differently from the reference: anything but Ok
// and a "robust" OutputTooLarge / InsufficientInput far away from the end of the input.
__device__ __forceinline__ bool needs_serial_recheck(const StreamResult& r, uint32_t flags) {
    if (flags & 8u) return false;  // FDH_FLAG_NO_RECHECK (tests: the tile decoder on its own)
    if (r.status == ST_OK) return false;
    // A decoder that got as far as comparing the checksum read every block to its end and found the four trailer
    // bytes behind the last one: the input never ran short inside a token, which is the only place where the
    // reference's pairing of literals shows (resync_to_step_start, inflate_stream.h), and the reference gets
    // there too -- Ok or WrongChecksum is then the comparison itself (src/decompress.rs:306-326).
    if (r.status == ST_WRONG_CHECKSUM) return false;
    if (r.status == ST_OUTPUT_TOO_LARGE || r.status == ST_INSUFFICIENT_INPUT) return r.ambiguous;
    return true;
}

constexpr int kSpanRingBytes = (kSegInWords + kSegOutWords) * kWave * 4;  // the span decoder's rings overlay `io`
struct GeneralLds {
    TableSet tables;
    WaveIo io;
    uint8_t span_pad[kSpanRingBytes > (int)sizeof(WaveIo) ? kSpanRingBytes - (int)sizeof(WaveIo) : 16];
    HeaderScratch hs;
};

// General kernel: workgroup = one wavefront = one stream, private decode tables in LDS.
__device__ __forceinline__ void general_one(const InflateBatchArgs& a, GeneralLds& lds, const uint64_t sid) {
    const int lane = threadIdx.x;
    if (sid >= a.n) return;
    bool tiles = !(a.flags & 2u);
    uint32_t st = kPending;
    if (a.only_pending) {
        st = load_word(&a.status[sid]);
        if (st != kPending && st != kPendingSerial && st != kPendingResume) return;
    }
    const StreamArgs s = stream_args(a, sid);
    Inflater inf(lds.tables, lds.io, &lds.hs, lane);
    StreamResult r;
    const ResumePoint rec = resume_point(a, sid, st);  // where a kernel in front left the stream
    ResumePoint rp = rec;
#ifdef FDH_DEBUG_STEP
    if (lane == 0) printf("general_one: sid %llu st %x tiles %d rec{valid %u h %llu b %llu o %u step %u} in_len %llu cap %u flags %x\n", (unsigned long long)sid, st, (int)tiles,
                          rec.valid, (unsigned long long)rec.hdr_bit, (unsigned long long)rec.bit, rec.opos, rec.step, (unsigned long long)s.in_len, s.cap, a.flags);
#endif
    // A tile decoder in front has been over this stream and left it for the exact serial decoder: no second
    // pass of tiles, with or without a check point to start from (without one -- no scratch for the records, or
    // FDH_FLAG_NO_CHECKPOINTS -- the serial decoder starts at the stream's first byte).
    if (st == kPendingSerial) tiles = false;
    if (tiles) {
        // the span decoder lists matches in global scratch: take one of the pool's slots
        uint32_t slot = kSpanSlots;
        if (a.span_pool && (a.flags & 0x100u) && s.in_len >= 4096) {
            if (lane == 0) {
                slot = (uint32_t)(sid % kSpanSlots);
                while (atomicCAS(&a.span_pool[slot], 0u, 1u) != 0u) slot = (slot + 1) % kSpanSlots;
            }
            slot = uni(slot);
            inf.span_list = a.span_pool + kSpanSlots + (size_t)slot * (2 * kSpanMaxMatches);
        }
        // Check points are taken at every block header and in front of every tile, and a result the tiles may
        // have classified differently from the reference is re-derived by the exact serial decoder from the last
        // of them -- the tail of the stream, not the stream (rounds 1-3: a damaged stream cost a serial pass over
        // all of it).  A tile starts at a symbol, not necessarily at one of the reference's table steps:
        // resync_to_step_start (inflate_stream.h) deals with that.
        inf.keep_ck = !(a.flags & 0x4000u);
        inf.init(s);
        r = rec.valid ? inf.run_from<true>(rec) : inf.run<true, false>();
        if (slot < kSpanSlots) {
            __threadfence();
            if (lane == 0) atomicExch(&a.span_pool[slot], 0u);
        }
        inf.span_list = nullptr;
        inf.keep_ck = false;
        if (needs_serial_recheck(r, a.flags)) {
            tiles = false;
            if (inf.ck.valid) rp = inf.ck;
        }
    }
    bool retracted = false;
    if (!tiles) {
        inf.keep_ck = a.resume_out != nullptr;  // (block headers, the last steps in front of the slot's end)
        inf.init(s);
        bool whole = true;
        if (rp.valid) {
            r = inf.run_from<false>(rp);
            whole = r.status == RC_REDO;  // the stream ended before the tail fell in step with the reference's table steps
            if (whole) inf.init(s);
            retracted = !whole && r.out_len < rp.opos;  // (the first literal of a pair whose second one is cut off)
        }
        if (whole && (a.flags & 0x8000u) && a.resume_out && a.resume_out != a.resume) {
            // ... then from the point this CALL took the stream up at, if that one knows its place among the steps
            // (the caller's record is still there: a final result overwrites it, and there is none yet)
            const ResumePoint first = unpack_record(load_record(&a.resume_out[sid]));
            if (first.valid && first.step != STEP_UNKNOWN) {
                rp = first;
                r = inf.run_from<false>(rp);
                whole = r.status == RC_REDO;  // (as above: RC_REDO is no result -- the stream goes from its first byte)
                if (whole) inf.init(s);
                retracted = !whole && r.out_len < rp.opos;
            }
        }
        if (whole) r = inf.run<false, false>();
    }
#ifdef FDH_DEBUG_STEP
    if (lane == 0) printf("general_one: sid %llu -> status %u len %u tiles %d rp{valid %u h %llu b %llu o %u} ck{valid %u h %llu b %llu o %u}\n", (unsigned long long)sid, r.status, r.out_len, (int)tiles,
                          rp.valid, (unsigned long long)rp.hdr_bit, (unsigned long long)rp.bit, rp.opos, inf.ck.valid, (unsigned long long)inf.ck.hdr_bit, (unsigned long long)inf.ck.bit, inf.ck.opos);
#endif
    if (lane == 0) {
        a.status[sid] = r.status;
        a.out_len[sid] = r.out_len;
        if (a.adler) a.adler[sid] = r.adler;
        if (a.resume_out) {
            ResumePoint out = inf.stopped_at(r);
            if (!out.valid || retracted) out = rp;  // (the point this run started from: still in front of everything missing)
            const bool again = r.status == ST_INSUFFICIENT_INPUT || r.status == ST_OUTPUT_TOO_LARGE;
            a.resume_out[sid] = again ? resume_record(out) : make_uint4(0, 0, 0, 0);
        }
    }
}
// What the landing decoder left over, told to the host without a synchronisation: a word pair in mapped host memory
// ([0] the count, [1] a counter of reports), written by the first kernel behind the landing decoder.  The host reads
// it at later calls and keeps the chain of launches behind the landing decoder short while nothing is left over
// (fdh_launch_inflate).  Internal flag bits (cleared from the caller's flags): the kernel reports; the exact kernel's
// hand-out counter is word 1 of its list (word 3 of that list was the landing decoder's own).
__device__ uint32_t* g_tail_report = nullptr;
constexpr uint32_t kFlagTailReport = 0x20000000u, kFlagTailCounter1 = 0x40000000u;
__device__ __forceinline__ void tail_report(uint32_t count, int what = 0) {  // what: 0 the landing decoder's leftovers, 2 the other list
    uint32_t* const rep = g_tail_report;
    if (rep) {
        rep[what] = count;
        rep[what + 1] = rep[what + 1] + 1;
        __threadfence_system();
    }
}

__global__ __launch_bounds__(kWave) void inflate_general_kernel(InflateBatchArgs a) {
    __shared__ GeneralLds lds;
    if (a.list) {  // only the streams a kernel in front has listed as left over, handed out by a counter
        const uint32_t cnt = a.list[0];
        if ((a.flags & kFlagTailReport) && blockIdx.x == 0 && threadIdx.x == 0) tail_report(cnt);
        const int cw = (a.flags & kFlagTailCounter1) ? 1 : 3;
        // (the first item of a workgroup is its own index, the following ones come from a counter: an
        // empty list costs no atomic -- four thousand of them on one address are 80 us)
        for (uint32_t i = blockIdx.x; i < cnt;) {
            general_one(a, lds, a.list[4 + i]);
            wave_sync();
            if (threadIdx.x == 0) i = atomicAdd(&a.list[cw], 1u) + gridDim.x;
            i = uni(i);
        }
        return;
    }
    general_one(a, lds, blockIdx.x);
}

// Fast general kernel: the same decoder with an 8-bit literal/length table (1 KiB instead of 16 KiB)
// and at most 128 VGPRs (20 of them spilled): 9.9 KiB of LDS per stream with the 2 KiB output ring,
// 16 workgroups per CU instead of 5 -- the tile decoder is latency-bound, so occupancy is
// throughput (10 bits / 159 VGPRs / 9 per CU: 33 GB/s on zlib-6 streams; 12 per CU: 40; 9 bits and
// 14 per CU: 45; this: 49).  Codes longer than 8 bits are resolved by the canonical walk.  Its
// double-literal pairing is not the reference's, so every result that needs the exact serial
// decoder is left PENDING_SERIAL for inflate_general_kernel.
constexpr int kFastLitBits = 8;
#ifndef FDH_FAST_WAVES_PER_SIMD
#define FDH_FAST_WAVES_PER_SIMD 4
#endif
// (No span decoder in this kernel: it would cost LDS for its rings.  The scratch of the block-header
// parser lies over the tiles' match list: a header is parsed between tiles, never during one.)
struct GeneralFastLds {
    TableSetT<kFastLitBits> tables;
    WaveIo io;
};
static_assert(sizeof(HeaderScratch) <= sizeof(WaveIo::mlist) && offsetof(WaveIo, mlist) % 16 == 0,
              "the header scratch must fit the match list it shares LDS with");
__device__ __forceinline__ void general_fast_one(const InflateBatchArgs& a, GeneralFastLds& lds, const uint64_t sid) {
    const int lane = threadIdx.x;
    if (sid >= a.n) return;
    uint32_t st = kPending;
    if (a.only_pending) {
        st = load_word(&a.status[sid]);
        if (st != kPending && st != kPendingResume) return;
    }
    const StreamArgs s = stream_args(a, sid);
    InflaterT<kFastLitBits, false> inf(lds.tables, lds.io, reinterpret_cast<HeaderScratch*>(lds.io.mlist), lane);
    inf.keep_ck = (a.resume != nullptr || a.resume_out != nullptr) && !(a.flags & 0x4000u);  // (what it cannot classify goes on from its last check point)
    inf.init(s);
    const ResumePoint rec = resume_point(a, sid, st);
    const StreamResult r = rec.valid ? inf.run_from<true>(rec) : inf.run<true, false>();
    if (needs_serial_recheck(r, a.flags)) {
        if (a.resume && lane == 0) a.resume[sid] = resume_record(inf.ck.valid ? inf.ck : rec);
        if (lane == 0) a.status[sid] = kPendingSerial;
    } else if (lane == 0) {
        a.status[sid] = r.status;
        a.out_len[sid] = r.out_len;
        if (a.adler) a.adler[sid] = r.adler;
        if (a.resume_out) {
            const bool again = r.status == ST_INSUFFICIENT_INPUT || r.status == ST_OUTPUT_TOO_LARGE;
            a.resume_out[sid] = again ? resume_record(inf.ck.valid ? inf.ck : rec) : make_uint4(0, 0, 0, 0);
        }
    }
}
__global__ __launch_bounds__(kWave, FDH_FAST_WAVES_PER_SIMD) void inflate_general_fast_kernel(InflateBatchArgs a) {
    __shared__ GeneralFastLds lds;
    if (a.list) {
        const uint32_t cnt = a.list[0];
        for (uint32_t i = blockIdx.x; i < cnt;) {
            general_fast_one(a, lds, a.list[4 + i]);
            wave_sync();
            if (threadIdx.x == 0) i = atomicAdd(&a.list[2], 1u) + gridDim.x;
            i = uni(i);
        }
        return;
    }
    general_fast_one(a, lds, blockIdx.x);
}

// LZ-window kernel (inflate_lz.h): any stream of Huffman blocks, the history in LDS.  One wavefront =
// one workgroup = one stream at a time, four per CU.  Returns true when the stream is finished (Ok).
__device__ __forceinline__ bool lz_one(const InflateBatchArgs& a, LzLds& L, const uint64_t sid, uint4& rec) {
    const int lane = threadIdx.x;
    rec = make_uint4(0, 0, 0, 0);  // (header bit 0: no resume point)
    if (sid >= a.n) return false;
    uint32_t st = kPending;
    if (a.only_pending) {  // finished by a kernel in front: nothing to do; left for the exact serial decoder: not ours
        st = load_word(&a.status[sid]);
        if (st != kPending && st != kPendingResume) return st != kPendingSerial;
    }
    const StreamArgs s = stream_args(a, sid);
    if (s.in_len < 8 || s.in_len >= (1ull << 27) || s.cap >= (1u << 30) || s.cap < 16) return false;  // 32-bit bit positions; far sources are read 16 bytes at a time
    // a stream that was stopped earlier (fdh_inflate_batch_resumable) goes on at its resume point: the block header is
    // parsed again for the tables, the ring gets the history from the slot
#ifdef FDH_LZ_NO_RESUME_IN
    ResumePoint from;
    from.valid = 0; from.step = 0; from.hdr_bit = 0; from.bit = 0; from.opos = 0; from.adler = 0;
#else
    const ResumePoint from = resume_point(a, sid, st);
#endif
    if (st == kPendingResume && !from.valid) return false;
    const uint32_t in_bits = (uint32_t)s.in_len * 8;
    // (the wave-serial reader is used for its bit window only: zlib header, block type, trailer)
    InflaterT<kLitBits, false> inf(*reinterpret_cast<TableSetT<kLitBits>*>(&L), *reinterpret_cast<WaveIo*>(&L.u.hdr), nullptr, lane);
    inf.init(s);
    if (inf.parse_zlib_header() != RC_OK) return false;
    LzIn lin;
    lin.base16 = inf.base16;
    lin.mis = inf.mis;
    lin.win_bytes = inf.win_bytes;
    lin.buf_lo = s.buf_lo;
    lin.buf_hi = s.buf_hi;
    lin.in_bits = in_bits;
    lin.cap = s.cap;
    lin.ck = a.lz_ck + (size_t)blockIdx.x * (kWave * kLzMaxPhases);
    LzOut o;
    o.out_al = inf.out_al;
    o.gmis = inf.gmis;
    o.O = 0;
    o.o_ri = inf.gmis;
    o.flushed = 0;
    o.adler_a = 1;
    o.adler_b = 0;
    uint32_t bitpos = 16, hdr_bit = 16;
    bool fixed_built = false;
    uint32_t start_bit = 0;  // (a resume point in the middle of a block: where its data goes on)
    if (from.valid) {
        if (from.opos > s.cap) return false;
        o.O = from.opos;
        o.flushed = from.opos;
        o.adler_a = from.adler & 0xFFFF;
        o.adler_b = from.adler >> 16;
        o.o_ri = (from.opos + inf.gmis) % kLzRing;
        const uint32_t lo_p = from.opos > kLzRing ? from.opos - kLzRing : 0;
        const uint8_t* const g = inf.out_al + inf.gmis;
        for (uint32_t p = lo_p + (uint32_t)lane; p < from.opos; p += kWave)
            L.ring[(p + inf.gmis) % kLzRing] = __hip_atomic_load(g + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wave_sync();
        bitpos = (uint32_t)from.hdr_bit;
        if (from.bit != from.hdr_bit) start_bit = (uint32_t)from.bit;
        inf.left = in_bits - bitpos;
        inf.loaded = 0x7FFFFFF0u;
        inf.seek(bitpos);
    }
    const uint32_t O_in = o.O;
    // This kernel gives up on the stream at stream bit `leave_bit` of the block whose header starts at hdr_bit
    // (leave_bit == hdr_bit: at the header itself): everything in front of it is decoded, so it all goes to the slot
    // with its checksum and the kernels behind take the stream up there instead of at its first byte.  (ONE place
    // does that, at the end: a dozen inlined copies of the flush cost the kernel 2 % of its speed.)
    uint32_t leave_bit = 0;
#define LZ_LEAVE(bit_)        \
    do {                      \
        leave_bit = (bit_);   \
        goto lz_leave;        \
    } while (0)
#ifdef FDH_LZ_DEBUG
    o.tq = clock64();
#endif
    uint32_t eob_bits = 0;
    for (;;) {  // blocks
        hdr_bit = bitpos;
        inf.refill();
        if (inf.left < 10) LZ_LEAVE(hdr_bit);
        const uint32_t type = ((uint32_t)inf.bb >> 1) & 3;
        if (type == 0 || type == 3) LZ_LEAVE(hdr_bit);  // stored blocks: the kernels behind
        uint32_t rc = RC_OK;
        inf.last_block = ((uint32_t)inf.bb & 1) != 0;
        if (type == 2) {  // dynamic: this kernel's own parser and table builder
            if (inf.left < 17) LZ_LEAVE(hdr_bit);
            const uint32_t hlit = (((uint32_t)inf.bb >> 3) & 31) + 257, hdist = (((uint32_t)inf.bb >> 8) & 31) + 1;
            const uint32_t hclen = (((uint32_t)inf.bb >> 13) & 15) + 4;
            if (hlit > 286 || hdist > 30) LZ_LEAVE(hdr_bit);
            inf.consume(17);
            fixed_built = false;
            LZT(o, 0);
            if (!lz_parse_dynamic(L, inf, hlit, hdist, hclen, lane, o)) LZ_LEAVE(hdr_bit);
            lz_build_sub(L, lane);
            LZT(o, 17);
        } else {  // fixed: the same builder on the lengths of src/tables.rs:207-232 (an empty block is one end-of-block token)
            inf.consume(3);
            if (!fixed_built) {
                uint32_t ll[5];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const uint32_t sy = (uint32_t)lane + 64u * k;
                    ll[k] = sy < 144 ? 8u : sy < 256 ? 9u : sy < 280 ? 7u : sy < 288 ? 8u : 0u;
                }
                if (!lz_build_tables(L, ll, lane < 32 ? 5u : 0u, lane, o)) LZ_LEAVE(hdr_bit);
                lz_build_sub(L, lane);
                fixed_built = true;
            }
        }
        const bool last = inf.last_block;
        bitpos = (uint32_t)inf.consumed_bits();
        if (start_bit != 0) {
            bitpos = start_bit;
            start_bit = 0;
        }
        const LzBounds bd = lz_load_bounds(L.tables);
        LZT(o, 0);
        if (rc == RC_OK) {
            uint32_t qcap = kLzRange;
            for (uint32_t nspans = 0;; nspans++) {  // super-spans
                if (bitpos >= in_bits || nspans > in_bits) LZ_LEAVE(bitpos);
                const uint32_t res = lz_superspan(L, o, bd, lin, bitpos, eob_bits, qcap, lane);
                if (res == LZ_DISTRUST) return false;
                if (res == LZ_BAIL) LZ_LEAVE(bitpos);
                if (res == LZ_SHRINK) {  // an item that does not fit an image: shorter phases from there on
                    if (qcap <= 8) LZ_LEAVE(bitpos);
                    qcap = max(8u, qcap / 4);
                    continue;
                }
                if (res == LZ_EOB) break;
            }
        }
        // back to the wave-serial reader (its window shares LDS with the span's stage: reload)
        if (bitpos > in_bits) LZ_LEAVE(bitpos - eob_bits);
        inf.left = in_bits - bitpos;
        inf.loaded = 0x7FFFFFF0u;
        inf.seek(bitpos);
        if (last) break;
    }
    // (a trailer that is cut short or wrong: the exact kernels take the stream up at the last end-of-block code)
    {
        uint32_t stored = 0;
        if (inf.read_trailer(stored) != RC_OK) LZ_LEAVE(bitpos - eob_bits);
        lz_flush(L, o, true, lane);
        const uint32_t adler = (o.adler_b << 16) | o.adler_a;
        if (!(a.flags & 1u) && stored != adler) LZ_LEAVE(bitpos - eob_bits);  // WrongChecksum is the exact kernels' verdict
        if (lane == 0) {
            a.status[sid] = ST_OK;
            a.out_len[sid] = o.O;
            if (a.adler) a.adler[sid] = adler;
        }
        LZT(o, 8);
#ifdef FDH_LZ_DEBUG
        if (lane == 0) {
            for (int k = 0; k < 32; k++)
                if (o.t[k]) atomicAdd(&g_lzstat[k], o.t[k]);
            atomicAdd(&g_lzstat[31], 1ull);
        }
#endif
    }
    return true;
lz_leave:
    if (from.valid && o.O == O_in) {  // nothing gained: the point this call started from stands (it may know more: its step state)
        rec = resume_record(from);
    } else if (o.O != 0 && !(a.flags & 0x4000u)) {
        lz_flush(L, o, true, lane);
        rec = make_uint4(hdr_bit, leave_bit, o.O, (o.adler_b << 16) | o.adler_a);
    }
    return false;
#undef LZ_LEAVE
}
#ifndef FDH_LZ_WAVES_PER_EU
#define FDH_LZ_WAVES_PER_EU ((FDH_LZ_WAVES_PER_CU + 3) / 4)
#endif
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(FDH_LZ_WAVES_PER_EU, FDH_LZ_WAVES_PER_EU)))
void inflate_lz_kernel(InflateBatchArgs a) {
    __shared__ LzLds lds;
    const uint32_t cnt = a.list[0];
    if ((a.flags & kFlagTailReport) && blockIdx.x == 0 && threadIdx.x == 0) tail_report(cnt, 2);
    for (uint32_t i = blockIdx.x; i < cnt;) {
        const uint32_t sid = a.list[4 + i];
        uint4 rec;
        const bool finished = lz_one(a, lds, sid, rec);
        wave_sync();
        if (threadIdx.x == 0) {
            // what this kernel leaves goes on a list of its own: the kernels behind do not look at the rest
            if (!finished && a.list_out) a.list_out[4 + atomicAdd(&a.list_out[0], 1u)] = sid;
            if (!finished && a.resume && rec.x != 0) {  // where the kernels behind take the stream up
                a.resume[sid] = rec;
                a.status[sid] = kPendingResume;
            }
            // a stream this kernel finished leaves no resume point: "all zero for every other status" (with
            // FDH_FLAG_RESUME_IN the caller's array still holds the record the stream was taken up at)
            if (finished && a.resume_out && a.status[sid] == ST_OK) a.resume_out[sid] = make_uint4(0, 0, 0, 0);
            i = atomicAdd(a.lz_counter, 1u) + gridDim.x;
        }
        i = uni(i);
    }
}

struct CanonLds {
    TableSet tables;
    WaveIo io[kCanonWaves];
};

__global__ __launch_bounds__(kCanonWaves* kWave, 4) void inflate_canon_kernel(InflateBatchArgs a) {
    __shared__ CanonLds lds;
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = threadIdx.x / kWave;
    uint64_t sid = (uint64_t)blockIdx.x * kCanonWaves + wid;
    // when this kernel only mops up after the segment kernel, most workgroups have nothing to do:
    // find that out before staging 18 KiB of tables.  With a compacted list the leftovers are
    // packed 8 to a workgroup.
    bool mine;
    if (a.list) {
        mine = sid < a.list[0];
        if (mine) sid = a.list[4 + sid];
    } else {
        mine = sid < a.n && (!a.only_pending || a.status[sid] == kPending);
    }
    if (!__syncthreads_or(mine ? 1 : 0)) return;
    // stage the shared tables: 16 B per lane, coalesced, served from L2 after the first workgroups
    {
        const uint4* src = reinterpret_cast<const uint4*>(g_canon.lit);
        uint4* dst = reinterpret_cast<uint4*>(lds.tables.lit);
        for (int i = threadIdx.x; i < kLitSize / 4; i += kCanonWaves * kWave) dst[i] = src[i];
        const uint4* srcd = reinterpret_cast<const uint4*>(g_canon.dist);
        uint4* dstd = reinterpret_cast<uint4*>(lds.tables.dist);
        for (int i = threadIdx.x; i < kDistSize / 4; i += kCanonWaves * kWave) dstd[i] = srcd[i];
        if (threadIdx.x < 4) lds.tables.eof[threadIdx.x] = g_canon.eof[threadIdx.x];
    }
    __syncthreads();
    if (!mine) return;
    const StreamArgs s = stream_args(a, sid);
    Inflater inf(lds.tables, lds.io[wid], nullptr, lane);
    inf.init(s);
    // does the stream start with the canonical prefix?  lane k compares stream dword k
    bool mismatch = false;
    if (lane < 14) {
        uint32_t wbit = inf.mis * 8 + 32 * lane;
        uint32_t d = wbit >> 5, sh = wbit & 31;
        uint32_t lo = lds.io[wid].in_ring[d & (kInRingDw - 1)];
        uint32_t hi = lds.io[wid].in_ring[(d + 1) & (kInRingDw - 1)];
        uint32_t v = __builtin_amdgcn_alignbit(hi, lo, sh);
        uint32_t ref = g_canon.hdr[lane];
        if (lane == 13) {
            v &= (1u << (kCanonBits - 13 * 32)) - 1;
        }
        mismatch = v != ref;
    }
    const bool canonical = !__any(mismatch) && s.in_len * 8 >= kCanonBits;
    if (!canonical) {
        if (lane == 0) a.status[sid] = kPending;
        return;
    }
    inf.left -= kCanonBits;
    inf.seek(kCanonBits);
    inf.last_block = true;  // BFINAL = 1 is part of the prefix
    inf.eof_code = lds.tables.eof[0];
    inf.eof_mask = lds.tables.eof[1];
    inf.eof_bits = lds.tables.eof[2];
    inf.hdr_bit = 16;  // (the block header of the prefix, for the check points)
    inf.keep_ck = (a.resume != nullptr || a.resume_out != nullptr) && !(a.flags & 0x4000u);
    StreamResult r = inf.run<true, true>();
    if (lane == 0) {
        if (needs_serial_recheck(r, a.flags)) {
            if (a.resume) a.resume[sid] = resume_record(inf.ck);
            a.status[sid] = kPendingSerial;
        } else {
            a.status[sid] = r.status;
            a.out_len[sid] = r.out_len;
            if (a.adler) a.adler[sid] = r.adler;
            if (a.resume_out) {
                const bool again = r.status == ST_INSUFFICIENT_INPUT || r.status == ST_OUTPUT_TOO_LARGE;
                a.resume_out[sid] = again ? resume_record(inf.ck) : make_uint4(0, 0, 0, 0);
            }
        }
    }
}

// Dense batches of canonical streams: one stream per lane (inflate_lanes.h).
__global__ __launch_bounds__(kLaneBlock) void inflate_lanes_kernel(LaneArgs a) {
    __shared__ LaneLds lds;
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.canon_lit);
        uint4* dst = reinterpret_cast<uint4*>(lds.lit);
        for (int i = threadIdx.x; i < kLitSize / 4; i += kLaneBlock) dst[i] = src[i];
    }
    __syncthreads();
    lanes_decode(a, lds);
}

// Canonical streams, segment-parallel: one stream per wavefront, one segment per lane
// (inflate_segments.h).
__global__ __launch_bounds__(kSegWaves* kWave, 4) void inflate_segments_kernel(SegArgs a) {
    __shared__ SegLds lds;
    if (a.src_list && a.src_list[0] == 0) return;  // the interval kernel left nothing over: do not even stage the table
    // the hand-scheduled loops address the table from LDS offset 0 (`raw & 0x3ffc` IS the address)
    if (lds_offset(lds.lit) != 0) __builtin_trap();
    // stage the table in this kernel's entry layout (up to three literals per entry), built from the
    // canonical table in global memory (L2 resident after the first workgroups)
    for (int i = threadIdx.x; i < kLitSize; i += kSegWaves * kWave) lds.lit[i] = seg_entry_build(a.canon_lit, (uint32_t)i);
    __syncthreads();
    if (a.list) {
        // persistent wavefronts: every wavefront keeps fetching the next stream, so a short stream
        // does not leave its slot of the workgroup idle and the table is staged once per workgroup
        const int lane = threadIdx.x & (kWave - 1);
        for (;;) {
            // The hand-out must not depend on WHICH lanes are active: the compiler gives this loop a
            // per-lane exit mask (it does not know that `next` is uniform), and "lane 0 fetches,
            // readfirstlane broadcasts" silently breaks once lane 0 is not the first active lane --
            // readfirstlane then returns another lane's initialiser and the wavefront decodes that
            // stream for ever (the round-1 "for (;;) hangs" -- reproduced and traced in round 2).  So the
            // first ACTIVE lane fetches, which is also the lane readfirstlane reads.
            uint32_t next = 0;
            const int leader = __ffsll((unsigned long long)__ballot(true)) - 1;
            if (lane == leader) next = atomicAdd(&a.list[1], 1u);
            next = uni(next);
            if (a.src_list) {  // only what the kernel in front left over
                if (next >= a.src_list[0]) break;
                next = uni(a.src_list[4 + next]);
            }
            if (next >= a.n) break;
#ifdef FDH_DEBUG_TILES
            // diagnostics of the stream hand-out: how often was each stream handed out, and was every
            // lane of the wavefront active here?
            if (lane == 0 && next < 65536) atomicAdd(&g_handed[next], 1u);
            if (__ballot(true) != ~0ull) atomicAdd(&g_handed[65536], 1u);
#endif
            segments_decode(a, lds, next);
        }
    } else {
        segments_decode(a, lds, (uint64_t)blockIdx.x * kSegWaves + threadIdx.x / kWave);
    }
}

// Hand-out order of a batch for persistent wavefronts: the streams of at least half the mean length
// first (order[0 ..]), then the shorter ones (order[n ..]), each class roughly in batch order (one atomic
// per wavefront and class).  A kernel ends when its last long stream does, and while the long streams
// drain the wavefronts that have run out of work take the short ones instead of idling (the bench batch:
// every 16th stream 0.5 KB, every 16th half the size).  Streams that do not start like an ultra-fast stream
// (its first eight bytes) never get to the interval kernel: they go straight onto the list of the kernels
// behind it -- handing 65 536 of them out one atomic at a time only to pass them on cost 0.8 ms.
// counters: [0..3] the streams of each class (SegOrder), zeroed.
__global__ __launch_bounds__(1024) void stream_order_kernel(const uint8_t* in, const uint64_t* in_off, uint32_t n, uint32_t* order,
                                                            uint32_t* counters, const uint32_t* canon_hdr, uint32_t* status,
                                                            uint32_t pending, uint32_t* list2, uint32_t second) {
    // (round 6) one atomic per class and workgroup of 16 wavefronts: a wavefront each, 1 024 of them after the
    // one counter most streams of a batch fall into, serialised to ~20 us in front of the first decode kernel
    __shared__ uint32_t s_cnt[16][5], s_base[16][5];
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint64_t mean = (in_off[n] - in_off[0]) / n, thr = mean / 2;
    const bool valid = i < n;
    const uint64_t len = valid ? in_off[i + 1] - in_off[i] : 0;
    bool canon = valid && len * 8 >= kCanonBits;
    if (canon) {
        const uint8_t* p = in + in_off[i];
        uint32_t w0 = 0, w1 = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            w0 |= (uint32_t)p[j] << (8 * j);
            w1 |= (uint32_t)p[4 + j] << (8 * j);
        }
        canon = w0 == canon_hdr[0] && w1 == canon_hdr[1];
    }
    // (the other streams' list is walked from the front by persistent wavefronts too: the long ones are listed by
    //  the first launch of this kernel, the short ones behind them by a second launch, `second`)
    // (second == 2: one launch lists all of them, in no order -- taken while recent calls had next to none, TailHint)
    const bool other = valid && !canon && (second == 2 || ((len >= thr) != (second != 0)));
    // class 0..3: the canonical streams by length (SegOrder); class 4: the other list
    const uint32_t cls = other ? 4u : ((canon && second != 1) ? seg_order_class(len, mean) : 5u);
    const uint64_t below = (1ull << lane) - 1;
    uint32_t rank = 0;
#pragma unroll
    for (uint32_t k = 0; k < 5; k++) {
        const uint64_t m = __ballot(cls == k);
        if (cls == k) rank = (uint32_t)__popcll(m & below);
        if (lane == 0) s_cnt[wave][k] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const uint32_t k = threadIdx.x;
        uint32_t total = 0;
        for (int w = 0; w < 16; w++) {
            s_base[w][k] = total;
            total += s_cnt[w][k];
        }
        uint32_t base = 0;
        if (total) base = atomicAdd(k < 4 ? &counters[k] : &list2[0], total);
        for (int w = 0; w < 16; w++) s_base[w][k] += base;
    }
    __syncthreads();
    if (cls < 4) {
        const uint32_t r = s_base[wave][cls] + rank;
        order[cls == 0 ? r : (cls == 1 ? n - 1 - r : (cls == 2 ? n + r : 2 * n - 1 - r))] = i;
    } else if (cls == 4) {
        status[i] = pending;
        list2[4 + s_base[wave][4] + rank] = i;
    }
}

// Canonical streams, counted by segments and written by intervals (inflate_seg2.h): one stream per
// wavefront, one workgroup of 16 persistent wavefronts per CU with all of its LDS.
__global__ __launch_bounds__(kS2Waves* kWave, 4) void inflate_seg2_kernel(SegArgs a) {
    __shared__ Seg2Lds lds;
    if (a.src_list && (a.flags & kFlagTailReport) && blockIdx.x == 0 && threadIdx.x == 0) tail_report(a.src_list[0]);
    if (a.src_list && a.src_list[0] == 0) return;  // the landing decoder left nothing over: do not even stage the table
    // the hand-scheduled loops address the table from LDS offset 0 (`raw & 0x3ffc` IS the address)
    if (lds_offset(lds.lit) != 0) __builtin_trap();
    {   // the table in this kernel's entry layout, built once per device by canon_build_kernel: a plain copy
        const uint4* src = reinterpret_cast<const uint4*>(a.canon_lit2);
        uint4* dst = reinterpret_cast<uint4*>(lds.lit);
        for (int i = threadIdx.x; i < kLitSize / 4; i += kS2Waves * kWave) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    uint2* const ckpt = a.ckpt + (size_t)(blockIdx.x * kS2Waves + threadIdx.x / kWave) * kS2CkptPerWave;
    // Streams are handed out by a counter, ONE at a time while the wavefront decodes them: the streams
    // in flight are then ~4 096 neighbours, whose input (read twice) and output stay in the Infinity
    // Cache (eight at a time: 3.30 -> 3.38 ms).  An atomic on one address costs ~20 ns when every
    // wavefront is after it -- 1.3 ms for 65 536 streams, all of it exposed in a batch this kernel can
    // only pass on -- so a wavefront that passed on everything it was given takes twice as many next time.
    // (with a hand-out order: its long streams, then its short ones -- what does not look canonical is not in it)
    // (behind the landing decoder: only what that kernel listed)
    SegOrder ord{0, 0, 0, 0};
    if (a.order) ord = SegOrder{uni(a.order_counts[0]), uni(a.order_counts[1]), uni(a.order_counts[2]), uni(a.order_counts[3])};
    const uint32_t n32 = a.src_list ? uni(a.src_list[0]) : (a.order ? ord.total() : (uint32_t)a.n);
    // (the first stream of a wavefront is its own number: 4 096 wavefronts after one counter at once is 80 us)
    const uint32_t n_waves = gridDim.x * kS2Waves;
    uint32_t cur = blockIdx.x * kS2Waves + threadIdx.x / kWave, end = cur + 1, take = 1;
    bool took = true;
    for (;;) {
        if (cur == end) {
            take = took ? 1u : min(16u, 2 * take);
            took = false;
            // the first ACTIVE lane fetches (see inflate_segments_kernel)
            uint32_t next = 0;
            const int leader = __ffsll((unsigned long long)__ballot(true)) - 1;
            if (lane == leader) next = atomicAdd(&a.list[2], take);
            cur = uni(next) + n_waves;
            end = min(n32, cur + take);
        }
        if (cur >= n32) break;
        const uint32_t sid = a.src_list ? uni(a.src_list[4 + cur])
                                        : (a.order ? uni(a.order[ord.at((uint32_t)a.n, cur)]) : cur);
        took = seg2_decode(a, lds, ckpt, sid) || took;
        cur++;
    }
}

// Parses the canonical prefix once per device and keeps the resulting tables in g_canon.
__global__ __launch_bounds__(kWave) void canon_build_kernel() {
    __shared__ GeneralLds lds;
    const int lane = threadIdx.x;
    StreamArgs s;
    s.in = g_canon_header;
    s.in_len = 54;
    s.out = nullptr;
    s.cap = 0;
    s.buf_lo = g_canon_header;
    s.buf_hi = g_canon_header + 64;
    s.flags = 0;
    Inflater inf(lds.tables, lds.io, &lds.hs, lane);
    inf.init(s);
    uint32_t rc = inf.parse_zlib_header();
    if (rc == RC_OK) rc = inf.parse_block_header();
    wave_sync();
    for (int i = lane; i < kLitSize; i += kWave) g_canon.lit[i] = lds.tables.lit[i];
    for (int i = lane; i < kLitSize; i += kWave) g_canon.lit2[i] = seg2_entry_build(lds.tables.lit, (uint32_t)i);
    for (int i = lane; i < kDistSize; i += kWave) g_canon.dist[i] = lds.tables.dist[i];
    if (lane < 14) {
        uint32_t w = (uint32_t)g_canon_header[4 * lane] | ((uint32_t)g_canon_header[4 * lane + 1] << 8) |
                     ((uint32_t)g_canon_header[4 * lane + 2] << 16) | ((uint32_t)g_canon_header[4 * lane + 3] << 24);
        if (lane == 13) w &= (1u << (kCanonBits - 13 * 32)) - 1;
        g_canon.hdr[lane] = w;
    }
    if (lane < 32) {
        uint32_t w = 0;
        for (int j = 0; j < 8; j++) w |= ((uint32_t)lds.hs.lens[8 * lane + j] & 15u) << (4 * j);
        g_canon.len4[lane] = w;
    }
    if (lane == 0) {
        // canonical bookkeeping of the symbols >= 256 (uniform, once per device)
        uint32_t code = 0, prev = 0, off = 0, lmin = 0, lmax = 0;
        for (uint32_t l = 0; l < 64; l++) g_canon.nl[l] = 0;
        for (uint32_t l = 1; l <= 15; l++) {
            uint32_t all = 0, lits = 0;
            for (uint32_t sy = 0; sy < 288; sy++) {
                if (lds.hs.lens[sy] == l) {
                    all++;
                    if (sy < 256) lits++;
                }
            }
            code = (code + prev) << 1;
            prev = all;
            const uint32_t nn = all - lits;
            if (l <= 15) g_canon.nl[l] = (code + lits) | (nn << 16) | (off << 22);
            if (nn) {
                if (!lmin) lmin = l;
                lmax = l;
                for (uint32_t sy = 256; sy < 288; sy++) {
                    if (lds.hs.lens[sy] == l && off < 32) {
                        g_canon.nl[32 + off] = (sy == 256 || sy >= 286)
                                                   ? (1u << 13)
                                                   : ((uint32_t)kLenBase[sy - 257] | ((uint32_t)kLenExtra[sy - 257] << 9) | (1u << 12));
                        off++;
                    }
                }
            }
        }
        g_canon.nl[16] = lmin;
        g_canon.nl[17] = lmax;
        g_canon.eof[0] = inf.eof_code;
        g_canon.eof[1] = inf.eof_mask;
        g_canon.eof[2] = inf.eof_bits;
        g_canon.eof[3] = 0;
        bool ok = rc == RC_OK && inf.last_block && inf.consumed_bits() == kCanonBits;
        // the lane kernel relies on the prefix declaring exactly one distance code: 1 bit, '0' =
        // distance 1 (HDIST = 1, reference src/lib.rs:8 "no distance codes except for RLE of zeros")
        ok = ok && lds.tables.dist[0] == DistTraits::entry(0, 1) && ((lds.tables.dist[1] >> 4) & 15) == D_INVALID;
        g_canon.status = ok ? (uint32_t)ST_OK : (rc == RC_OK ? 0xBADu : rc);
    }
}

// Debug / parity hook behind fdh_debug_build_tables.
__global__ __launch_bounds__(kWave) void build_tables_debug_kernel(const uint8_t* code_lengths, uint32_t hlit,
                                                                   uint32_t* litlen, uint32_t* dist,
                                                                   uint32_t* build_status) {
    __shared__ GeneralLds lds;
    const int lane = threadIdx.x;
    for (int i = lane; i < 320; i += kWave) lds.hs.lens[i] = code_lengths[i];
    for (int i = lane; i < kLitSize; i += kWave) lds.tables.lit[i] = 0xFFFFFFFFu;
    for (int i = lane; i < kDistSize; i += kWave) lds.tables.dist[i] = 0xFFFFFFFFu;
    wave_sync();
    Inflater inf(lds.tables, lds.io, &lds.hs, lane);
    inf.eof_code = inf.eof_mask = inf.eof_bits = 0;
    uint32_t st = inf.build_block_tables(hlit);
    wave_sync();
    for (int i = lane; i < kLitSize; i += kWave) litlen[i] = lds.tables.lit[i];
    for (int i = lane; i < kDistSize; i += kWave) dist[i] = lds.tables.dist[i];
    if (lane == 0) {
        build_status[0] = st;
        build_status[1] = inf.eof_code;
        build_status[2] = inf.eof_mask;
        build_status[3] = inf.eof_bits;
    }
}

}  // namespace fdh

#ifdef FDH_DEBUG_TILES
extern "C" int fdh_debug_read_handed(uint32_t* host, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(host, HIP_SYMBOL(fdh::g_handed), 65537 * 4);
    if (reset) {
        static uint32_t z[65537];
        hipMemcpyToSymbol(HIP_SYMBOL(fdh::g_handed), z, sizeof(z));
    }
    return 0;
}
extern "C" int fdh_debug_read_seg(uint32_t* host) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(host, HIP_SYMBOL(fdh::g_segdbg), 64 * 16 * 4);
    return 0;
}
extern "C" int fdh_debug_read_seg2(uint32_t* host) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(host, HIP_SYMBOL(fdh::g_segdbg2), 16 * 16 * 4);
    return 0;
}
extern "C" int fdh_debug_read_gstat(unsigned long long* host, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(host, HIP_SYMBOL(fdh::g_gstat), 24 * 8);
    if (reset) { unsigned long long z[24] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(fdh::g_gstat), z, 24 * 8); }
    return 0;
}
extern "C" int fdh_debug_read_segtime(uint32_t* host) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(host, HIP_SYMBOL(fdh::g_segtime), 4096 * 8 * 4);
    return 0;
}
extern "C" int fdh_debug_read(uint32_t* host, uint32_t nwords, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(host + 8, HIP_SYMBOL(fdh::g_dbg), (nwords - 8) * 4, 32);
    hipMemcpyFromSymbol(host, HIP_SYMBOL(fdh::g_dbg_n), 4);
    if (reset) { uint32_t z = 0; hipMemcpyToSymbol(HIP_SYMBOL(fdh::g_dbg_n), &z, 4); }
    return 0;
}
#endif

#ifdef FDH_S2_DEBUG
extern "C" int fdh_debug_s2(uint32_t* host, uint32_t sid) {  // returns the records of the last run, then arms for `sid`
    hipDeviceSynchronize();
    uint32_t nrec = 0;
    hipMemcpyFromSymbol(&nrec, HIP_SYMBOL(fdh::g_s2dbg_n), 4);
    hipMemcpyFromSymbol(host, HIP_SYMBOL(fdh::g_s2dbg), 8 * 2048 * 4);
    uint32_t z = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(fdh::g_s2dbg_n), &z, 4);
    hipMemcpyToSymbol(HIP_SYMBOL(fdh::g_s2dbg_sid), &sid, 4);
    return (int)nrec;
}
#endif

#ifdef FDH_LZ_DEBUG
extern "C" int fdh_debug_read_lzstat(unsigned long long* host, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(host, HIP_SYMBOL(fdh::g_lzstat), 32 * 8);
    if (reset) { unsigned long long z[32] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(fdh::g_lzstat), z, 32 * 8); }
    return 0;
}
#endif

#ifdef FDH_S2_DEBUG
extern "C" int fdh_debug_s2time(uint32_t* host) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(host, HIP_SYMBOL(fdh::g_s2time), 4096 * 16 * 4);
    hipMemcpyFromSymbol(host + 4096 * 16, HIP_SYMBOL(fdh::g_s2time2), 4096 * 8 * 4);
    return 0;
}
#endif

// device address of g_canon, looked up once per device by fdh_launch_canon_build (the lookup
// synchronises, so it must stay off the launch path)
static fdh::CanonTables* g_canon_dev[64] = {};
static uint32_t* g_span_pool[64] = {};  // per device: scratch of the span decoder (never freed)
static int g_cu_count[64] = {};         // per device: compute units (0 = not asked yet)
static hipStream_t g_side_stream[64] = {};  // per device: the stream the LZ-window kernel runs on beside the canonical kernels
static std::mutex g_dev_mutex;          // guards the per-device caches above and below
// What the landing decoder leaves over, as the kernels report it (g_tail_report): while the reports say "next to nothing"
// the call launches ONE kernel behind the landing decoder (the exact kernel, which takes any stream) instead of five (the
// interval, segment and tile decoders and the two exact kernels: each launch costs ~10 us of the chain when its list is
// empty -- 60 us of a 2.6 ms call).  The latest report decides: more than kTailFew streams left over, the long chain.
// A hint only: every stream is decoded either way, a wrong guess costs time (the exact kernel is slow).
struct TailHint {
    volatile uint32_t* rep = nullptr;  // mapped host memory
    uint32_t seen = 0;                 // rep[1] at the last look
    uint32_t seen_others = 0;          // rep[3]
    int streak = 0;
    bool short_chain = false;
    bool order_once = false;           // stream_order_kernel in one launch: the other list in no order (it has been next to empty)
    bool tried = false;
};
static TailHint g_tail[64];
constexpr uint32_t kTailFew = 16;

// Scratch of a call (lists, check points, records): stream-ordered allocations from a pool of the library's own that
// KEEPS what is freed (release threshold = everything).  With the device's default pool -- which hands its memory back
// at every synchronisation -- a call that followed a hipStreamSynchronize got fresh pages, and about one such call in
// ten then read ZEROS where the first kernel of the call had just written (seen on the record resume_prepare_kernel
// leaves for the kernels behind it: status "taken up at a resume point", record all zero).  Until round 5 that only
// cost time -- a stream without a record is decoded from its first byte -- and went unnoticed; with the streaming
// object's moved buffers (stream_decompressor.cpp) it decoded garbage.  Memory that stays mapped does not do it
// (tools/streamtime.py, 16 runs of ~130 calls each: 0 failures against 6 in 8), and a call no longer pays for mapping
// and unmapping its scratch.
static hipMemPool_t g_scratch_pool[64];
static hipError_t scratch_alloc(void** p, size_t bytes, hipStream_t stream) {
    int dev = 0;
    hipMemPool_t pool = nullptr;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        std::lock_guard<std::mutex> lock(g_dev_mutex);
        if (!g_scratch_pool[dev]) {
            hipMemPoolProps props = {};
            props.allocType = hipMemAllocationTypePinned;
            props.location.type = hipMemLocationTypeDevice;
            props.location.id = dev;
            hipMemPool_t q = nullptr;
            const hipError_t ce = hipMemPoolCreate(&q, &props);
            if (ce != hipSuccess || !q) {  // (no fall-back to the default pool: that is the pool the zeros came from)
                (void)hipGetLastError();
                return ce != hipSuccess ? ce : hipErrorOutOfMemory;
            }
            uint64_t keep = ~0ull;
            const hipError_t se = hipMemPoolSetAttribute(q, hipMemPoolAttrReleaseThreshold, &keep);
            if (se != hipSuccess) {
                (void)hipGetLastError();
                (void)hipMemPoolDestroy(q);
                return se;
            }
            g_scratch_pool[dev] = q;
        }
        pool = g_scratch_pool[dev];
    }
    if (!pool) return hipErrorInvalidDevice;
    return hipMallocFromPoolAsync(p, bytes, pool, stream);
}

// (introspection, tests / soak: resume records that read zero under a status that promised one, since the library was loaded)
extern "C" int fdh_debug_lost_records(unsigned int* count) {
    if (!count) return 1;
    return hipMemcpyFromSymbol(count, HIP_SYMBOL(fdh::g_lost_records), sizeof(unsigned int)) == hipSuccess ? 0 : 3;
}

extern "C" int fdh_launch_canon_build(hipStream_t stream, uint32_t* host_status) {
    hipLaunchKernelGGL(fdh::canon_build_kernel, dim3(1), dim3(fdh::kWave), 0, stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return (int)e;
    fdh::CanonTables* dev = nullptr;
    e = hipGetSymbolAddress(reinterpret_cast<void**>(&dev), HIP_SYMBOL(fdh::g_canon));
    if (e != hipSuccess) return (int)e;
    e = hipMemcpy(host_status, &dev->status, sizeof(uint32_t), hipMemcpyDeviceToHost);
    int ordinal = 0;
    if (e == hipSuccess) e = hipGetDevice(&ordinal);
    if (e == hipSuccess && ordinal >= 0 && ordinal < 64) {
        std::lock_guard<std::mutex> lock(g_dev_mutex);
        g_canon_dev[ordinal] = dev;
        // the word pair the kernels behind the landing decoder report to (optional: without it every call takes the long chain)
        TailHint& h = g_tail[ordinal];
        if (!h.tried) {
            h.tried = true;
            uint32_t* hp = nullptr;
            if (hipHostMalloc(reinterpret_cast<void**>(&hp), 4 * sizeof(uint32_t), hipHostMallocMapped) == hipSuccess && hp) {
                hp[0] = hp[1] = hp[2] = hp[3] = 0;
                uint32_t* dp = nullptr;
                if (hipHostGetDevicePointer(reinterpret_cast<void**>(&dp), hp, 0) == hipSuccess && dp &&
                    hipMemcpyToSymbol(HIP_SYMBOL(fdh::g_tail_report), &dp, sizeof(dp)) == hipSuccess) {
                    h.rep = hp;
                } else {
                    (void)hipGetLastError();
                    (void)hipHostFree(hp);
                }
            } else {
                (void)hipGetLastError();
            }
        }
    }
    return (int)e;
}

// (introspection: the last report's count, the number of reports, the streak of small ones, the mode)
extern "C" int fdh_debug_tail_state(unsigned int* out4) {
    int dev = 0;
    if (!out4 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    const TailHint& h = g_tail[dev];
    out4[0] = h.rep ? h.rep[0] : 0xFFFFFFFFu;
    out4[1] = h.rep ? h.rep[1] : 0xFFFFFFFFu;
    out4[2] = (unsigned int)h.streak;
    out4[3] = h.short_chain ? 1u : 0u;
    return 0;
}

// (introspection, tests: 1 = the short chain behind the landing decoder is in use on this device, 0 = the long one)
extern "C" int fdh_debug_tail_mode(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    return g_tail[dev].short_chain ? 1 : 0;
}

// Statuses of a batch whose streams are taken up at resume points: PENDING_RESUME where there is one, PENDING elsewhere.
// (... a copy of the records for the kernels to pass on between them, and the list of all streams for the first one)
__global__ __launch_bounds__(256) void resume_prepare_kernel(uint32_t* status, const uint4* resume, uint4* work, uint32_t* scratch, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const uint4 r = resume[i];
        work[i] = r;  // (the caller's record stays until a final result overwrites it: general_one may need it)
        status[i] = r.x != 0 ? fdh::kPendingResume : fdh::kPending;
        scratch[8 + i] = (uint32_t)i;
    }
    if (i == 0) {
        scratch[2] = 0;            // hand-out counter of the LZ-window kernel
        scratch[4] = (uint32_t)n;  // the list's count
        scratch[5] = scratch[6] = scratch[7] = 0;
    }
}

// `resume_io` (nullable, n records of 16 bytes): where a stream that ends InsufficientInput / OutputTooLarge can be
// taken up again; with FDH_FLAG_RESUME_IN (0x8000) also where each stream is to be taken up NOW (all zero: at its
// first byte) -- the slot then holds the output up to that point, and only the two general kernels run.
// inflate_seg3.hip (a translation unit of its own: it builds in a fraction of the time of this one)
int fdh_launch_seg3(const fdh::SegArgs& sa, unsigned blocks, hipStream_t stream);

extern "C" int fdh_launch_inflate(const uint8_t* in, const uint64_t* in_off, uint8_t* out, const uint64_t* out_off,
                                  uint32_t* out_len, uint32_t* status, uint32_t* adler, uint64_t n, uint32_t flags,
                                  void* resume_io, hipStream_t stream) {
    if (n == 0) return 0;
    flags &= ~(fdh::kFlagTailReport | fdh::kFlagTailCounter1);  // (internal)
    fdh::InflateBatchArgs a{in, in_off, out, out_off, out_len, status, adler, n, flags, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                            static_cast<uint4*>(resume_io)};
    if (resume_io && (flags & 0x8000u)) {
        // The LZ-window kernel takes every stream as far as it can (from its resume point), the 12-bit kernel does the
        // rest: its tiles know where the reference's table steps start, so the serial decoder can take over at its
        // check points whatever the data -- the small-table kernel's cannot.  Scratch: a list of all streams for the
        // persistent wavefronts of the first kernel, its items, and the records the two kernels pass between them
        // (the caller's array keeps what came in until the final result of a stream overwrites it).
        int ordinal = 0, cus = 256;
        if (hipGetDevice(&ordinal) == hipSuccess) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ordinal) == hipSuccess && v > 0) cus = v;
        }
        const unsigned lblocks = (unsigned)std::min<uint64_t>(n, (uint64_t)FDH_LZ_WAVES_PER_CU * cus);
        const size_t list_words = ((size_t)n + 8 + 3) & ~(size_t)3;
        const size_t lzck_bytes = (size_t)lblocks * fdh::kWave * fdh::kLzMaxPhases * sizeof(uint2);
        uint32_t* scratch = nullptr;
        hipError_t e = scratch_alloc(reinterpret_cast<void**>(&scratch), list_words * sizeof(uint32_t) + lzck_bytes + (size_t)n * sizeof(uint4), stream);
        if (e != hipSuccess) return (int)e;
        a.lz_counter = scratch + 2;
        a.list = scratch + 4;   // [0] = count, [4..] = ids (its hand-out words [2], [3] are not used by this kernel)
        a.lz_ck = reinterpret_cast<uint2*>(scratch + list_words);
        a.resume = reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(a.lz_ck) + lzck_bytes);
        a.only_pending = 1;
        hipLaunchKernelGGL(resume_prepare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, status, a.resume_out, a.resume,
                           scratch, n);
        if (!(flags & 0x1000u)) hipLaunchKernelGGL(fdh::inflate_lz_kernel, dim3(lblocks), dim3(fdh::kWave), 0, stream, a);
        a.list = nullptr;
        hipLaunchKernelGGL(fdh::inflate_general_kernel, dim3((unsigned)n), dim3(fdh::kWave), 0, stream, a);
        e = hipGetLastError();
        (void)hipFreeAsync(scratch, stream);
        return (int)e;
    }
    if (resume_io) {  // (same promise for a call that starts every stream at its first byte)
        const hipError_t e0 = hipMemsetAsync(resume_io, 0, (size_t)n * sizeof(uint4), stream);
        if (e0 != hipSuccess) return (int)e0;
    }
    if (flags & 0x100u) {  // FDH_FLAG_SPANS: scratch of the span decoder, allocated once per device, zero-initialised
        int ordinal = 0;
        if (hipGetDevice(&ordinal) == hipSuccess && ordinal >= 0 && ordinal < 64) {
            std::lock_guard<std::mutex> lock(g_dev_mutex);
            if (!g_span_pool[ordinal]) {
                const size_t bytes = ((size_t)fdh::kSpanSlots + (size_t)fdh::kSpanSlots * 2 * fdh::kSpanMaxMatches) * sizeof(uint32_t);
                uint32_t* p = nullptr;
                if (hipMalloc(reinterpret_cast<void**>(&p), bytes) == hipSuccess) {
                    if (hipMemset(p, 0, fdh::kSpanSlots * sizeof(uint32_t)) == hipSuccess && hipDeviceSynchronize() == hipSuccess)
                        g_span_pool[ordinal] = p;
                    else (void)hipFree(p);
                } else {
                    (void)hipGetLastError();  // no scratch: the general kernel runs without spans
                }
            }
            a.span_pool = g_span_pool[ordinal];
        }
    }
    if (flags & 6u) {  // FDH_FLAG_SERIAL_ONLY (2) / debug: general kernel only, tiles allowed (4)
        hipLaunchKernelGGL(fdh::inflate_general_kernel, dim3((unsigned)n), dim3(fdh::kWave), 0, stream, a);
        return (int)hipGetLastError();
    }
    hipError_t e;
    // Canonical streams of useful length: segment-parallel kernel first; what it cannot finish
    // stays PENDING for the kernels below.
    if (!(flags & 128u)) {
        int ordinal = 0;
        e = hipGetDevice(&ordinal);
        if (e != hipSuccess) return (int)e;
        fdh::CanonTables* canon = (ordinal >= 0 && ordinal < 64) ? g_canon_dev[ordinal] : nullptr;
        if (!canon) return (int)hipErrorNotInitialized;
        // stream-ordered scratch (no host synchronisation): two compacted lists of leftovers (what the
        // interval kernel leaves to the segment kernel, what that one leaves to the kernels behind) and
        // the interval kernel's checkpoints
        uint32_t* list = nullptr;
        int cus;
        {
            std::lock_guard<std::mutex> lock(g_dev_mutex);
            if (g_cu_count[ordinal & 63] == 0) {
                int v = 0;
                if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ordinal) != hipSuccess || v <= 0) v = 256;
                g_cu_count[ordinal & 63] = v;
            }
            cus = g_cu_count[ordinal & 63];
        }
        const bool seg2 = !(flags & 0x400u);
        // A large batch is split by stream_order_kernel into the streams with the ultra-fast prefix and the others:
        // two independent lists.  The LZ-window kernel takes the others on a stream of its own, beside the landing /
        // interval / segment / tile kernels of the canonical ones, and the two meet again in front of the kernels
        // that take what is left (round 4 ran the lists back to back: the mix of BASELINE config 5 paid the sum).
        hipStream_t side = nullptr;
        if (seg2 && !(flags & (0x1000u | 0x2000u | 0x800u | 0x20000u | 64u | 0x100000u))) {
            std::lock_guard<std::mutex> lock(g_dev_mutex);
            if (!g_side_stream[ordinal & 63]) {
                hipStream_t s2 = nullptr;
                if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) == hipSuccess) g_side_stream[ordinal & 63] = s2;
                else (void)hipGetLastError();
            }
            side = g_side_stream[ordinal & 63];
        }
        const bool seg3 = seg2 && !(flags & 0x10000u);  // the landing decoder in front of the interval decoder
        const unsigned s2blocks = std::min((unsigned)((n + fdh::kS2Waves - 1) / fdh::kS2Waves), (unsigned)cus);
        // (+ the hand-out order of the interval kernel when every wavefront gets several streams);
        // layout: first list | 8 words: counters of stream_order_kernel (4 classes, [4] the LZ-window kernel's hand-out) | second list | order | third list | checkpoints
        const bool ordered = seg2 && n >= 4ull * s2blocks * fdh::kS2Waves && n <= 0x7FFFFFFFull;
        const size_t list2_at = (size_t)(n + 4) + 8;
        const size_t list3_at = list2_at + (size_t)(n + 4) + (ordered ? (size_t)(2 * n) : 0);
        const bool overlap = ordered && side != nullptr;
        const size_t list4_at = list3_at + (seg3 ? (size_t)(n + 4) : 0);  // (overlap: what the canonical kernels leave)
        const size_t list5_at = list4_at + (overlap ? (size_t)(n + 4) : 0);  // (overlap: what the LZ-window kernel leaves)
        const size_t list_words = list5_at + (overlap ? (size_t)(n + 4) : 0);
        const size_t ckpt_bytes = seg2 ? (size_t)s2blocks * fdh::kS2Waves * fdh::kS2CkptPerWave * sizeof(uint2) : 0;
        const unsigned lblocks = (unsigned)std::min<uint64_t>(n, (uint64_t)FDH_LZ_WAVES_PER_CU * cus);  // LZ-window kernel: persistent wavefronts
        const size_t lzck_bytes = (flags & 0x1000u) ? 0 : (size_t)lblocks * fdh::kWave * fdh::kLzMaxPhases * sizeof(uint2);
        const size_t resume_bytes = (size_t)n * sizeof(uint4);  // where a kernel leaves a stream for the kernels behind it
        const size_t words_al = (list_words + 3) & ~(size_t)3;  // (what follows the lists is 16-byte aligned)
        if (scratch_alloc(reinterpret_cast<void**>(&list), words_al * sizeof(uint32_t) + ckpt_bytes + lzck_bytes + resume_bytes, stream) != hipSuccess) {
            (void)hipGetLastError();
            list = nullptr;  // fall back to the status-scan form
        } else {
            // the headers of the lists and the counters between them: one fill over the list words is cheaper than three
            // small ones (a fill is a kernel of its own on the stream)
            if (seg3 || ordered) {
                e = hipMemsetAsync(list, 0, words_al * sizeof(uint32_t), stream);
            } else {
                e = hipMemsetAsync(list, 0, 4 * sizeof(uint32_t), stream);
                if (e == hipSuccess) e = hipMemsetAsync(list + (n + 4), 0, 12 * sizeof(uint32_t), stream);  // counters + second header
            }
            if (e != hipSuccess) {
                (void)hipFreeAsync(list, stream);
                return (int)e;
            }
        }
        fdh::SegArgs sa{in, in_off, out, out_off, out_len, status, adler, n, flags, canon->lit, canon->len4, canon->hdr,
                        fdh::kCanonBits, fdh::kPending, list, nullptr, nullptr, canon->nl, canon->lit2, nullptr, nullptr};
        if (list && seg2) {  // interval kernel first; what it leaves goes through the segment kernel
            sa.ckpt = reinterpret_cast<uint2*>(list + words_al);
            sa.list2 = list + list2_at;
            if (ordered) {
                uint32_t* order = list + list2_at + (n + 4);
                uint32_t* counters = list + (n + 4);
                // (the other list's long streams first takes a launch of its own for the short ones: ~12 us of the chain in
                //  front of the landing decoder, spent in vain while that list is as good as empty -- the LZ-window kernel
                //  reports its count like the kernel behind the landing decoder does, TailHint)
                bool once = false;
                if (side) {
                    std::lock_guard<std::mutex> lock(g_dev_mutex);
                    TailHint& h = g_tail[ordinal & 63];
                    if (h.rep) {
                        const uint32_t seq = h.rep[3], others = h.rep[2];
                        if (seq != h.seen_others) {
                            h.seen_others = seq;
                            h.order_once = others <= kTailFew;
                        }
                        once = h.order_once;
                    }
                }
                if (flags & 0x800000u) once = true;    // FDH_FLAG_ORDER_ONCE
                if (flags & 0x1000000u) once = false;  // FDH_FLAG_ORDER_TWICE
                for (uint32_t second = once ? 2 : 0; second < (once ? 3u : 2u); second++)
                    hipLaunchKernelGGL(fdh::stream_order_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, stream, in, in_off, (uint32_t)n, order,
                                       counters, canon->hdr, status, fdh::kPending, sa.list2, second);
                e = hipGetLastError();
                if (e != hipSuccess) {
                    (void)hipFreeAsync(list, stream);
                    return (int)e;
                }
                sa.order = order;
                sa.order_counts = counters;
            }
            hipEvent_t ev_join = nullptr;
            if (overlap) {  // the other list is complete: the LZ-window kernel starts on it now, on its own stream
                hipEvent_t ev_fork = nullptr;
                bool forked = hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) == hipSuccess &&
                              hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) == hipSuccess &&
                              hipEventRecord(ev_fork, stream) == hipSuccess && hipStreamWaitEvent(side, ev_fork, 0) == hipSuccess;
                if (forked) {
                    fdh::InflateBatchArgs b = a;
                    b.only_pending = 1;
                    b.flags = flags | fdh::kFlagTailReport;  // (its list's count goes to the host's hint, if there is one)
                    b.list = list + list2_at;
                    b.list_out = list + list5_at;
                    b.lz_counter = list + (n + 4) + 4;  // (a spare word of stream_order_kernel's counters, zeroed above)
                    b.lz_ck = reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(list) + words_al * sizeof(uint32_t) + ckpt_bytes);
                    b.resume = reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(list) + words_al * sizeof(uint32_t) + ckpt_bytes + lzck_bytes);
                    hipLaunchKernelGGL(fdh::inflate_lz_kernel, dim3(lblocks), dim3(fdh::kWave), 0, side, b);
                    forked = hipGetLastError() == hipSuccess;
                    // (round 6) what the LZ-window kernel leaves depends on nothing the canonical kernels do: its two
                    // exact kernels follow it on the side stream, off the chain of launches behind the landing decoder
                    // (a launch that finds its list empty still costs ~10 us of that chain)
                    fdh::InflateBatchArgs g = a;
                    g.only_pending = 1;
                    g.resume = b.resume;
                    g.list = list + list5_at;
                    const unsigned gblocks5 = (unsigned)std::min<uint64_t>(n, 4096);
                    if (forked && !(flags & 0x200u)) {
                        hipLaunchKernelGGL(fdh::inflate_general_fast_kernel, dim3(gblocks5), dim3(fdh::kWave), 0, side, g);
                        forked = hipGetLastError() == hipSuccess;
                    }
                    if (forked) {
                        hipLaunchKernelGGL(fdh::inflate_general_kernel, dim3(gblocks5), dim3(fdh::kWave), 0, side, g);
                        forked = hipGetLastError() == hipSuccess;
                    }
                    forked = forked && hipEventRecord(ev_join, side) == hipSuccess;
                }
                if (ev_fork) (void)hipEventDestroy(ev_fork);
                if (!forked) {  // (nothing has been started on the other stream, or it cannot be joined: give up cleanly)
                    if (ev_join) (void)hipEventDestroy(ev_join);
                    (void)hipStreamSynchronize(side);
                    (void)hipFreeAsync(list, stream);
                    return (int)hipErrorUnknown;
                }
                sa.list2 = nullptr;  // (nothing more goes on the other list: what the canonical kernels meet and cannot
                                     //  take stays on their own lists)
            }
            if (seg3) {  // what it does not take (short streams, a chain that did not land) is listed for the interval kernel
                sa.list = list + list3_at;
                e = (hipError_t)fdh_launch_seg3(sa, s2blocks, stream);
                if (e != hipSuccess || (flags & 0x20000u)) {  // (debug: the landing decoder only)
                    if (ev_join) {
                        (void)hipStreamWaitEvent(stream, ev_join, 0);
                        (void)hipEventDestroy(ev_join);
                    }
                    (void)hipFreeAsync(list, stream);
                    return (int)e;
                }
                sa.src_list = list + list3_at;
                sa.list = list;
            }
            if (seg3 && overlap) {
                // how many streams the landing decoder has been leaving over lately (TailHint)
                bool short_chain = false, report = false;
                {
                    std::lock_guard<std::mutex> lock(g_dev_mutex);
                    TailHint& h = g_tail[ordinal & 63];
                    if (h.rep) {
                        report = true;
                        const uint32_t seq = h.rep[1], left = h.rep[0];
                        if (seq != h.seen) {  // (a caller that enqueues calls faster than they run sees few reports: the latest one decides)
                            h.seen = seq;
                            h.streak = left > kTailFew ? 0 : h.streak + 1;
                            h.short_chain = left <= kTailFew;
                        }
                        short_chain = h.short_chain;
                    }
                }
                if (flags & 0x200000u) short_chain = false;  // FDH_FLAG_TAIL_LONG
                if (flags & 0x400000u) short_chain = true;   // FDH_FLAG_TAIL_SHORT
                if (short_chain) {  // the exact kernel alone on what the landing decoder listed, then the join
                    a.only_pending = 1;
                    a.resume = reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(list) + words_al * sizeof(uint32_t) + ckpt_bytes + lzck_bytes);
                    a.list = list + list3_at;
                    a.flags = flags | fdh::kFlagTailCounter1 | (report ? fdh::kFlagTailReport : 0u);
                    hipLaunchKernelGGL(fdh::inflate_general_kernel, dim3((unsigned)std::min<uint64_t>(n, 1024)), dim3(fdh::kWave), 0, stream, a);
                    e = hipGetLastError();
                    a.flags = flags;
                    a.list = nullptr;
                    const hipError_t ej = hipStreamWaitEvent(stream, ev_join, 0);
                    (void)hipEventDestroy(ev_join);
                    if (e == hipSuccess) e = ej;
                    if (e != hipSuccess) (void)hipStreamSynchronize(side);  // (the scratch is about to go)
                    (void)hipFreeAsync(list, stream);
                    return (int)e;
                }
                if (report) sa.flags = flags | fdh::kFlagTailReport;
            }
            hipLaunchKernelGGL(fdh::inflate_seg2_kernel, dim3(s2blocks), dim3(fdh::kS2Waves * fdh::kWave), 0, stream, sa);
            e = hipGetLastError();
            if (e != hipSuccess) {
                if (ev_join) {
                    (void)hipStreamWaitEvent(stream, ev_join, 0);
                    (void)hipEventDestroy(ev_join);
                }
                (void)hipFreeAsync(list, stream);
                return (int)e;
            }
            sa.src_list = list;
            sa.list = overlap ? list + list4_at : list + list2_at;
            sa.list2 = nullptr;
            sa.order = nullptr;
            if (flags & 0x800u) {  // debug: the interval kernel only
                (void)hipFreeAsync(list, stream);
                return 0;
            }
            if (overlap) {
                // the segment kernel and the tile decoder on what the interval kernels left, then -- both lists done
                // with their fast kernels -- the kernels that take whatever is still pending, list by list
                const unsigned sblocks2 = std::min((unsigned)((n + fdh::kSegWaves - 1) / fdh::kSegWaves), (unsigned)(2 * cus));
                hipLaunchKernelGGL(fdh::inflate_segments_kernel, dim3(sblocks2), dim3(fdh::kSegWaves * fdh::kWave), 0, stream, sa);
                e = hipGetLastError();
                a.only_pending = 1;
                a.resume = reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(list) + words_al * sizeof(uint32_t) + ckpt_bytes + lzck_bytes);
                a.list = list + list4_at;
                if (e == hipSuccess) {
                    const unsigned cblocks = (unsigned)((n + fdh::kCanonWaves - 1) / fdh::kCanonWaves);
                    hipLaunchKernelGGL(fdh::inflate_canon_kernel, dim3(cblocks), dim3(fdh::kCanonWaves * fdh::kWave), 0, stream, a);
                    e = hipGetLastError();
                }
                const hipError_t ej = hipStreamWaitEvent(stream, ev_join, 0);
                (void)hipEventDestroy(ev_join);
                if (e == hipSuccess) e = ej;
                const unsigned gblocks = (unsigned)std::min<uint64_t>(n, 4096);
                if (e == hipSuccess) {  // (the other list's exact kernels ran on the side stream, behind the LZ-window kernel)
                    a.list = list + list4_at;
                    if (!(flags & 0x200u)) {
                        hipLaunchKernelGGL(fdh::inflate_general_fast_kernel, dim3(gblocks), dim3(fdh::kWave), 0, stream, a);
                        e = hipGetLastError();
                    }
                    if (e == hipSuccess) {
                        hipLaunchKernelGGL(fdh::inflate_general_kernel, dim3(gblocks), dim3(fdh::kWave), 0, stream, a);
                        e = hipGetLastError();
                    }
                }
                a.list = nullptr;
                if (e != hipSuccess) (void)hipStreamSynchronize(side);  // (the scratch is about to go)
                (void)hipFreeAsync(list, stream);
                return (int)e;
            }
        }
        unsigned sblocks = (unsigned)((n + fdh::kSegWaves - 1) / fdh::kSegWaves);
        if (list) sblocks = std::min(sblocks, (unsigned)(2 * cus));  // persistent wavefronts: two workgroups (80 KiB of LDS each) per CU
        hipLaunchKernelGGL(fdh::inflate_segments_kernel, dim3(sblocks), dim3(fdh::kSegWaves * fdh::kWave), 0, stream, sa);
        e = hipGetLastError();
        if (e != hipSuccess) {
            if (list) (void)hipFreeAsync(list, stream);
            return (int)e;
        }
        a.only_pending = 1;
        if (flags & 64u) {  // debug: first kernel only (PENDING streams stay undecoded)
            if (list) (void)hipFreeAsync(list, stream);
            return 0;
        }
        if (list) {
            a.list = sa.list;
            a.resume = reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(list) + words_al * sizeof(uint32_t) + ckpt_bytes + lzck_bytes);
            unsigned cblocks = (unsigned)((n + fdh::kCanonWaves - 1) / fdh::kCanonWaves);
            hipLaunchKernelGGL(fdh::inflate_canon_kernel, dim3(cblocks), dim3(fdh::kCanonWaves * fdh::kWave), 0, stream, a);
            e = hipGetLastError();
            // the general kernels walk the same list (what the canon kernel finished is no longer PENDING):
            // a grid-stride loop, so a batch that is all canonical costs two near-empty launches
            const unsigned gblocks = (unsigned)std::min<uint64_t>(n, 4096);  // persistent workgroups (16 per CU at most)
            if (e == hipSuccess && !(flags & 0x1000u)) {  // the LZ-window kernel: persistent wavefronts, FDH_LZ_WAVES_PER_CU per CU
                a.lz_counter = list + (n + 4) + 4;  // (a spare word of stream_order_kernel's counters, zeroed above)
                a.lz_ck = reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(list) + words_al * sizeof(uint32_t) + ckpt_bytes);

                // its leftovers: the list region the kernels in front are done with
                a.list_out = (sa.list == list) ? list + list2_at : list;
                e = hipMemsetAsync(a.list_out, 0, 4 * sizeof(uint32_t), stream);
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(fdh::inflate_lz_kernel, dim3(lblocks), dim3(fdh::kWave), 0, stream, a);
                    e = hipGetLastError();
                }
                a.list = a.list_out;
            }
            if (e == hipSuccess && (flags & 0x2000u)) {  // debug: what the LZ-window kernel left stays PENDING
                (void)hipFreeAsync(list, stream);
                return 0;
            }
            if (e == hipSuccess && !(flags & 0x200u)) {
                hipLaunchKernelGGL(fdh::inflate_general_fast_kernel, dim3(gblocks), dim3(fdh::kWave), 0, stream, a);
                e = hipGetLastError();
            }
            if (e == hipSuccess) {
                hipLaunchKernelGGL(fdh::inflate_general_kernel, dim3(gblocks), dim3(fdh::kWave), 0, stream, a);
                e = hipGetLastError();
            }
            a.list = nullptr;
            (void)hipFreeAsync(list, stream);
            return (int)e;
        }
    }
    // Dense batches first go through the stream-per-lane kernel; it finishes the canonical
    // streams that decode cleanly and leaves everything else PENDING.
    const bool lanes = (flags & 16u) || (n >= fdh::kLaneMinStreams && !(flags & 32u));
    if (lanes && a.only_pending == 0) {
        int ordinal = 0;
        e = hipGetDevice(&ordinal);
        if (e != hipSuccess) return (int)e;
        fdh::CanonTables* canon = (ordinal >= 0 && ordinal < 64) ? g_canon_dev[ordinal] : nullptr;
        if (!canon) return (int)hipErrorNotInitialized;
        fdh::LaneArgs la{in, in_off, out, out_off, out_len, status, adler, n, flags,
                         canon->lit, canon->dist, canon->hdr, fdh::kCanonBits, fdh::kPending};
        unsigned lblocks = (unsigned)((n + fdh::kLaneBlock - 1) / fdh::kLaneBlock);
        hipLaunchKernelGGL(fdh::inflate_lanes_kernel, dim3(lblocks), dim3(fdh::kLaneBlock), 0, stream, la);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        a.only_pending = 1;
        if (flags & 64u) return 0;  // debug: lane kernel only (PENDING streams stay undecoded)
    }
    unsigned blocks = (unsigned)((n + fdh::kCanonWaves - 1) / fdh::kCanonWaves);
    hipLaunchKernelGGL(fdh::inflate_canon_kernel, dim3(blocks), dim3(fdh::kCanonWaves * fdh::kWave), 0, stream, a);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    a.only_pending = 1;
    if (!(flags & 0x200u)) {
        hipLaunchKernelGGL(fdh::inflate_general_fast_kernel, dim3((unsigned)n), dim3(fdh::kWave), 0, stream, a);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(fdh::inflate_general_kernel, dim3((unsigned)n), dim3(fdh::kWave), 0, stream, a);
    return (int)hipGetLastError();
}

extern "C" int fdh_launch_build_tables_debug(const uint8_t* code_lengths, uint32_t hlit, uint32_t* litlen,
                                             uint32_t* dist, uint32_t* build_status, hipStream_t stream) {
    hipLaunchKernelGGL(fdh::build_tables_debug_kernel, dim3(1), dim3(fdh::kWave), 0, stream, code_lengths, hlit,
                       litlen, dist, build_status);
    return (int)hipGetLastError();
}
