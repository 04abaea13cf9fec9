// inflate_segments.h -- segment-parallel decode of one ultra-fast-format stream per wavefront.
//
// A stream that starts with the ultra-fast encoder's fixed prefix (reference
// src/compress/ultrafast.rs:82-88) is one final dynamic block with a known table.  Its block
// data is cut into up to 64 equal bit ranges ("segments", >= kSegMinBits each; short streams use
// fewer lanes), one per lane, and every lane runs the reference's inner loop
// (src/decompress.rs:645-830: table look-up, 1-2 literals per step, dist-1 run) sequentially over
// its own segment:
//
//   pass 1  every lane decodes from the FIRST BIT of its segment -- a guess, the real symbol
//           boundary lies up to 17 bits further: the first kSegWindow bits are only walked
//           (impossible tokens slide on by one bit), then output bytes are counted to the end of
//           the segment.  Huffman codes self-synchronise, so by the end of the window the guessed
//           chain has (almost always) joined the real one; x0 = where the chain left the window.
//   check   lane i takes its real start from lane i-1's end, decodes the window from there
//           (counting) and must land exactly on x0.  If it does, by induction from lane 0 every
//           chain from x0 on was the real one.  A lane that lands elsewhere re-counts its segment
//           from the landing point and the check repeats for its successors (rare).
//   scan    a wavefront prefix sum of the byte counts gives every lane its output offset;
//           the byte a leading run repeats comes from the nearest lane to the left that emitted
//           a literal.
//   pass 2  every lane decodes its real chain again and streams the bytes through a 4-byte
//           accumulator and a 64-B LDS ring to global memory (32-B aligned pairs of 16-B stores;
//           long runs line by line), folding them into a per-lane Adler-32 partial; the partials
//           are combined with the block-combine identity.
//
// Input is read per lane through a 64-B LDS ring that is topped up at wavefront-uniform "events"
// (one per kSegSteps steps, two adjacent 16-B loads at a time, committed an event later), so the
// hot loops never wait on memory.  The hot loops themselves are hand-scheduled groups of
// literal-only steps (seg_count_group / seg_write_group); any other entry is handled by a
// select-only C++ step.
// Anything unusual -- not canonical, a bad / truncated token, a full slot, a checksum
// mismatch -- leaves the stream PENDING for the exact wave-per-stream kernels.
#pragma once
#include "inflate_tables.h"

namespace fdh {

#ifdef FDH_DEBUG_TILES
__device__ uint32_t g_segdbg[64 * 16];
#define SEGDBG(slot, val) do { if (sid < 64 && lane == 0) g_segdbg[sid * 16 + (slot)] = (val); } while (0)
#define SEGDBG_ADD(slot, val) do { if (sid < 64 && lane == 0) g_segdbg[sid * 16 + (slot)] += (val); } while (0)
__device__ uint32_t g_segtime[4096 * 8];
#define SEGTIME(k) do { if (sid < 4096 && lane == 0) g_segtime[sid * 8 + (k)] = (uint32_t)clock64(); } while (0)
#else
#define SEGTIME(k) do { } while (0)
#define SEGDBG(slot, val) do { } while (0)
#define SEGDBG_ADD(slot, val) do { } while (0)
#endif

constexpr int kSegWaves = 8;      // wavefronts (= streams) per workgroup: 80 KiB of LDS, two workgroups per CU
constexpr int kSegInWords = 16;   // per-lane input ring, dwords
constexpr int kSegOutWords = 16;  // per-lane output ring, dwords
constexpr int kSegChunk = 4;       // dwords per global load of a lane (16 B)
constexpr int kSegWindow = 256;     // bits of a segment used for self-synchronisation (the default)
constexpr int kSegWindowShort = 192;  // ... for streams of short codes
constexpr int kSegSteps = kSegInWords / 2;  // table look-ups between two global-memory events
constexpr uint32_t kSegNeed = (kSegSteps * 18 + 31) / 32;  // dwords a group of steps can consume (18 bits/token)
constexpr uint32_t kSegMinBits = 768;   // aim: no segment shorter than this (fewer lanes are used instead)
constexpr uint32_t kSegBulkFill = 64;             // runs at least this long are stored line by line
constexpr uint32_t kNoByte = 0x100;  // "no literal seen yet"

// Table entry of this kernel (converted from the device layout of inflate_tables.h while staging).
// The fields the hot loops need are whole bytes, so they are used straight from the entry:
//   byte 0   bits consumed by the whole token [4:0] (a run: code + extra bits + the 1-bit distance
//            code) | SE_RUN | SE_EOB | SE_BAD (cannot occur / invalid)
//   byte 1   8 x literal bytes of the token (0, 8, 16)
//   byte 2-3 literals: first byte, second byte;  runs: length base [24:16], extra-bit count [27:25]
// The common case -- every running lane looks at a literal entry -- has byte 0 = bits consumed.
enum : uint32_t { SE_RUN = 0x20, SE_EOB = 0x40, SE_BAD = 0x80, SE_SPECIAL = 0xE0 };
__device__ __forceinline__ uint32_t seg_entry_from(uint32_t e) {
    const uint32_t nb = e & 15, kind = (e >> 4) & 15;
    if (kind == K_LIT1) return nb | (8u << 8) | (((e >> 8) & 0xFF) << 16);
    if (kind == K_LIT2) return nb | (16u << 8) | (((e >> 8) & 0xFFFF) << 16);
    if (kind == K_LEN) {
        const uint32_t ex = (e >> 8) & 31, base = e >> 16;
        return (nb + ex + 1) | SE_RUN | (base << 16) | (ex << 25);
    }
    if (kind == K_EOB) return nb | SE_EOB;
    return SE_BAD;
}
__device__ __forceinline__ uint32_t seg_used(uint32_t e) { return e & 31; }
__device__ __forceinline__ uint32_t seg_n8(uint32_t e) { return (e >> 8) & 0xFF; }
// last literal byte of a literal entry (one literal: byte 2, two: byte 3)
__device__ __forceinline__ uint32_t seg_lastlit(uint32_t e) { return (e >> (8 + seg_n8(e))) & 0xFF; }

// Rings are [wavefront][word][lane]: any per-lane word index is bank-conflict free, and with
// 16 x 64 dwords per ring the slot of (wavefront, lane) and the word index occupy disjoint bits of
// the element index, so a ring address is one shift and one and-or.
struct SegLds {
    uint32_t lit[kLitSize];
    uint32_t in_ring[kSegWaves * kSegInWords * kWave];
    uint32_t out_ring[kSegWaves * kSegOutWords * kWave];
};
static_assert((kSegInWords == 16 || kSegInWords == 8) && kSegOutWords == kSegInWords && kWave == 64, "ring addressing");
__device__ __forceinline__ uint32_t seg_slot(uint32_t lane_off, uint32_t word) {
    return lane_off | ((word & (uint32_t)(kSegInWords - 1)) << 6);
}

struct SegArgs {
    const uint8_t* in;
    const uint64_t* in_off;
    uint8_t* out;
    const uint64_t* out_off;
    uint32_t* out_len;
    uint32_t* status;
    uint32_t* adler;
    uint64_t n;
    uint32_t flags;
    const uint32_t* canon_lit;  // kLitSize entries (device layout, inflate_tables.h)
    const uint32_t* canon_hdr;  // 14 dwords of prefix (last one masked)
    uint32_t canon_bits;
    uint32_t pending;
    uint32_t* list;  // nullable: [0] = count, [4..] = ids of the streams left PENDING (compacted);
                     // [1] = next stream to hand out (persistent wavefronts fetch their work here)
};

// Leaves stream `sid` to the wave-per-stream kernels (lane 0 only).
__device__ __forceinline__ void seg_leave_pending(const SegArgs& a, uint64_t sid) {
    a.status[sid] = a.pending;
    if (a.list) {
        uint32_t k = atomicAdd(&a.list[0], 1u);
        a.list[4 + k] = (uint32_t)sid;
    }
}

// One chunk (kSegChunk dwords) from a chunk-aligned address, zero where outside [lo, hi).
struct SegChunk {
    uint32_t w[kSegChunk];
};
__device__ __attribute__((noinline)) SegChunk seg_load_edge(const uint8_t* p, const uint8_t* lo, const uint8_t* hi) {
    SegChunk c;
    for (int k = 0; k < kSegChunk; k++) {
        uint32_t v = 0;
        for (int j = 0; j < 4; j++) {
            const uint8_t* q = p + 4 * k + j;
            if (q >= lo && q < hi) v |= (uint32_t)*q << (8 * j);
        }
        c.w[k] = v;
    }
    return c;
}
__device__ __forceinline__ SegChunk seg_load(const uint8_t* p, const uint8_t* lo, const uint8_t* hi) {
    if (p >= lo && p + 4 * kSegChunk <= hi) {
        SegChunk c;
        if (kSegChunk == 4) {
            const uint4 v = *reinterpret_cast<const uint4*>(p);
            c.w[0] = v.x;
            c.w[1] = v.y;
            c.w[kSegChunk - 2] = v.z;
            c.w[kSegChunk - 1] = v.w;
        } else {
            const uint2 v = *reinterpret_cast<const uint2*>(p);
            c.w[0] = v.x;
            c.w[1] = v.y;
        }
        return c;
    }
    return seg_load_edge(p, lo, hi);
}

// Per-lane sequential bit reader over the lane's input ring.  `lo`/`hi` hold the 64 bits at the
// read position; the dword after them is fetched from the ring at the start of every step.
// The reader sits TWO BITS IN FRONT of the next token: bits [13:2] of the raw 32-bit window are the
// 12 table-index bits, so `raw & 0x3ffc` is the byte offset of the table entry (one instruction).
struct SegReader {
    uint32_t* ring;        // in_ring of the workgroup
    uint32_t lane_off;     // wavefront * 1024 + lane: this lane's slot for word 0
    const uint8_t* gp;     // next 16-B chunk to request from global memory
    const uint8_t* buf_lo;
    const uint8_t* buf_hi;
    uint32_t in_wr, in_rd; // dwords written to / read from the ring
    uint32_t lo, hi, boff;
    SegChunk pend_a, pend_b;  // chunks requested two / one events ago
    bool has_a, has_b;

    __device__ __forceinline__ void put(const SegChunk& v) {
        // in_wr is a multiple of the chunk size, so a chunk never wraps: one slot address, constant offsets
        uint32_t* const p = ring + seg_slot(lane_off, in_wr);
#pragma unroll
        for (int k = 0; k < kSegChunk; k++) p[k * kWave] = v.w[k];
        in_wr += kSegChunk;
    }
    // Positions the reader at stream bit `bit` (relative to the stream's first byte `in`) and
    // primes the whole ring synchronously.
    __device__ __forceinline__ void start(const uint8_t* in, uint32_t bit) {
        bit -= 2;  // see above; every start lies behind the stream's 53-byte prefix
        const uint8_t* addr = in + (bit >> 3);
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(addr) & (4 * kSegChunk - 1));
        gp = addr - mis;
        in_wr = in_rd = 0;
        has_a = has_b = false;
        for (int k = 0; k < kSegInWords / kSegChunk; k++) {
            put(seg_load(gp, buf_lo, buf_hi));
            gp += 4 * kSegChunk;
        }
        in_rd = mis >> 2;
        lo = ring[seg_slot(lane_off, in_rd)];
        hi = ring[seg_slot(lane_off, in_rd + 1)];
        in_rd += 2;
        boff = 8 * (mis & 3) + (bit & 7);
    }
    // Synchronous top-up (once, between the window walk and the long counting loop): commits what is
    // in flight, then loads until the ring is full.
    __device__ __forceinline__ void refill_now() {
        if (has_a) {  // a pair is in flight: take it if it fits, otherwise ask again below
            if ((uint32_t)kSegInWords - (in_wr - in_rd) >= (uint32_t)(2 * kSegChunk)) {
                put(pend_a);
                put(pend_b);
            } else {
                gp -= 8 * kSegChunk;
            }
        }
        has_a = has_b = false;
        while ((uint32_t)kSegInWords - (in_wr - in_rd) >= (uint32_t)kSegChunk) {
            put(seg_load(gp, buf_lo, buf_hi));
            gp += 4 * kSegChunk;
        }
    }
    __device__ __forceinline__ uint32_t level() const { return in_wr - in_rd; }  // dwords past lo/hi
    __device__ __forceinline__ bool starved() const { return in_rd > in_wr; }
    __device__ __forceinline__ uint32_t raw_window() const { return __builtin_amdgcn_alignbit(hi, lo, boff); }
    __device__ __forceinline__ uint32_t window() const { return raw_window() >> 2; }  // 30 stream bits
    // the dword that follows lo/hi (read at the start of a step, so its latency hides behind the table look-up)
    __device__ __forceinline__ uint32_t peek() const { return ring[seg_slot(lane_off, in_rd)]; }
    // Branch-free advance by `used` (<= 32) bits; nw = peek() from before.
    __device__ __forceinline__ void advance(uint32_t used, uint32_t nw) {
        boff += used;
        const bool wrap = boff >= 32;
        boff &= 31;
        lo = wrap ? hi : lo;
        hi = wrap ? nw : hi;
        in_rd += wrap ? 1u : 0u;
    }
    // Wavefront-uniform event, pair policy: two adjacent chunks (one 32-B sector of a 16-dword ring's
    // lane) are requested together and committed together once the ring has room for both, so a
    // 128-B line of input is visited 4 times instead of 8.  has_a == has_b at all times.
    __device__ __forceinline__ void event(bool want_more) {
        if (has_a && (uint32_t)kSegInWords - (in_wr - in_rd) >= (uint32_t)(2 * kSegChunk)) {
            put(pend_a);
            put(pend_b);
            has_a = has_b = false;
        }
        if (want_more && !has_a) {
            if (gp >= buf_lo && gp + 8 * kSegChunk <= buf_hi) {  // one range check for the pair
                const uint4 va = *reinterpret_cast<const uint4*>(gp);
                const uint4 vb = *reinterpret_cast<const uint4*>(gp + 4 * kSegChunk);
                pend_a.w[0] = va.x;
                pend_a.w[1] = va.y;
                pend_a.w[2] = va.z;
                pend_a.w[3] = va.w;
                pend_b.w[0] = vb.x;
                pend_b.w[1] = vb.y;
                pend_b.w[2] = vb.z;
                pend_b.w[3] = vb.w;
            } else {
                pend_a = seg_load(gp, buf_lo, buf_hi);
                pend_b = seg_load(gp + 4 * kSegChunk, buf_lo, buf_hi);
            }
            gp += 8 * kSegChunk;
            has_a = has_b = true;
        }
    }
    // One event per kSegSteps steps keeps up with a chunk per group; a group can consume up to
    // kSegSteps * 18 bits = kSegNeed dwords, so denser stretches get extra (waiting) events.
    __device__ __forceinline__ void events(bool running, uint32_t need = kSegNeed) {
        event(running);
        for (int x = 0; x < 2 && __any(running && level() < need); x++) event(running);
    }
};

// A run token, resolved: its length and whether its distance code is not the declared one.
struct SegRun {
    uint32_t length;
    bool bad_dist;
};
__device__ __forceinline__ SegRun seg_run(uint32_t e, uint32_t win) {
    const uint32_t used = seg_used(e);
    const uint32_t ex = (e >> 25) & 7;
    const uint32_t code_bits = used - ex - 1;
    SegRun r;
    r.length = ((e >> 16) & 0x1FF) + ((win >> (code_bits & 31)) & ((1u << ex) - 1));
    // the prefix declares one distance code: '0' = distance 1; it is the token's last bit
    r.bad_dist = ((win >> ((used - 1) & 31)) & 1) != 0;
    return r;
}

// State of one lane's counting scan.
struct SegScan {
    uint32_t pos;      // segment-relative bit position of the next token
    uint32_t count8;   // 8 x output bytes counted so far
    uint32_t last_e;   // entry of the last literal token counted (0 if none)
    uint32_t stop;     // 0 none, 1 end-of-block (pos = its start, eob_bits its length), 2 bad token
    uint32_t eob_bits;
};

// Walks a chain through the synchronisation window: from s.pos until pos >= window (or a stop).
// GUESS: the chain is only a way to find a synchronisation point -- nothing is counted, and an
// impossible token (or a stray end-of-block) just means "not synchronised yet": slide on by one
// bit.  The landing check is what guarantees correctness.  Otherwise (the real chain) every byte
// is counted.  Close to the window's end a literal pair is taken one literal at a time: the
// guessed and the real chain may pair literals differently, but they then still cross the window
// on the same symbol boundary (the length of the first literal alone comes from the canonical
// table in global memory: this happens two or three times per scan).
template <bool GUESS>
__device__ __forceinline__ uint32_t seg_window_scan(const uint32_t* lit, const uint32_t* canon_lit, SegReader& rd,
                                                    uint32_t limit, bool active, uint32_t window, SegScan& s) {
    bool running = active && s.pos < window;
    uint32_t iter = 0;
    while (__any(running)) {
        rd.events(running);
#pragma unroll 1
        for (int k = 0; k < kSegSteps; k++) {
            iter++;
            const uint32_t win = rd.window();
            const uint32_t e = lit[win & (kLitSize - 1)];
            const uint32_t nw = rd.peek();
            uint32_t used = seg_used(e), n8 = seg_n8(e);
            const bool is_run = (e & SE_RUN) != 0;
            bool is_eob = (e & SE_EOB) != 0;
            bool bad = (e & SE_BAD) != 0;
            uint32_t run = 0;
            if (__any(running && is_run)) {
                const SegRun r = seg_run(e, win);
                run = is_run ? r.length : 0u;
                bad = bad || (is_run && r.bad_dist);
            }
            uint32_t e_lit = e;
            const bool single = n8 == 16 && s.pos + 24 >= window;
            if (__any(running && single)) {
                if (running && single) {
                    used = canon_lit[win & (kLitSize - 1)] >> 24;  // K_LIT2: bits of the first symbol (inflate_tables.h)
                    n8 = 8;
                    e_lit = (e & 0x00FF0000u) | (8u << 8) | used;  // the first literal alone
                }
            }
            if (GUESS) {
                const bool slide = (bad || is_eob) && s.pos + 1 <= limit;
                used = slide ? 1u : used;
                bad = slide ? false : bad;
                is_eob = slide ? false : is_eob;
            }
            const bool fault = bad || s.pos + used > limit;
            const bool step = running && !fault && !is_eob;
            const bool halt = running && !step;
            s.stop = halt ? (fault ? 2u : 1u) : s.stop;
            s.eob_bits = halt ? used : s.eob_bits;
            if (!GUESS) {
                s.count8 += step ? n8 + 8 * run : 0u;
                s.last_e = (step && n8 != 0) ? e_lit : s.last_e;
            }
            const uint32_t adv = step ? used : 0u;
            s.pos += adv;
            rd.advance(adv, nw);
            running = step && s.pos < window;
        }
    }
    return iter;
}

// Byte offset of an LDS object inside the workgroup's LDS allocation.
__device__ __forceinline__ uint32_t lds_offset(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// Up to `left` literal-only steps of the counting loop, hand-scheduled.  The choice of
// instructions follows their measured issue cost on gfx950 (two cycles per wavefront for plain
// VOP2, four for VOP3 / compares / scalar instructions): the table entry's fields are whole
// bytes, the entry itself is summed (byte 1 of the sum = 8 x bytes), the ring address is kept as
// an LDS byte address, and the word hand-over at a dword boundary runs under the execution mask.
// Stops in front of a step in which some running lane looks at a special entry and returns the
// number of steps still to do (that step included); nothing of that step has been applied.
// Requires the literal table at LDS offset 0.
__device__ __forceinline__ uint32_t seg_count_group(uint32_t left, uint32_t lim, uint32_t ring_base, SegReader& rd,
                                                    uint32_t& pos, uint32_t& gsum, uint32_t& last_e) {
    static_assert(kSegInWords == 16, "ring word mask 0xf00 below");
    uint32_t ra = ring_base | ((rd.in_rd << 8) & 0xf00u);
    uint32_t w2, e, nw, t;
    uint64_t sv;
    asm volatile(
        "  s_waitcnt lgkmcnt(0)\n"  // nothing of the compiler's may be in flight: the counted waits below assume it
        "Lstep_%=:\n"
        "  v_alignbit_b32 %[w2], %[hi], %[lo], %[b]\n"
        "  v_and_b32 %[t], 0x3ffc, %[w2]\n"
        "  ds_read_b32 %[e], %[t]\n"
        "  ds_read_b32 %[nw], %[ra]\n"
        "  v_cmp_lt_u32 vcc, %[pos], %[lim]\n"
        "  s_and_saveexec_b64 %[sv], vcc\n"
        "  s_waitcnt lgkmcnt(1)\n"
        "  v_and_b32 %[t], 0xe0, %[e]\n"
        "  v_cmp_ne_u32 vcc, 0, %[t]\n"
        "  s_cbranch_vccnz Lspecial_%=\n"
        "  v_and_b32 %[t], 0xff, %[e]\n"
        "  v_add_u32 %[gsum], %[gsum], %[e]\n"
        "  v_mov_b32 %[last], %[e]\n"
        "  v_add_u32 %[pos], %[pos], %[t]\n"
        "  v_add_u32 %[b], %[b], %[t]\n"
        "  v_cmp_lt_u32 vcc, 31, %[b]\n"
        "  v_and_b32 %[b], 31, %[b]\n"
        "  s_and_b64 exec, exec, vcc\n"
        "  s_waitcnt lgkmcnt(0)\n"
        "  v_mov_b32 %[lo], %[hi]\n"
        "  v_mov_b32 %[hi], %[nw]\n"
        "  v_add_u32 %[t], 0x100, %[ra]\n"
        "  v_and_b32 %[t], 0xf00, %[t]\n"
        "  v_or_b32 %[ra], %[rb], %[t]\n"
        "  v_add_u32 %[ird], 1, %[ird]\n"
        "  s_mov_b64 exec, %[sv]\n"
        "  s_sub_u32 %[left], %[left], 1\n"
        "  s_cmp_lg_u32 %[left], 0\n"
        "  s_cbranch_scc1 Lstep_%=\n"
        "  s_branch Ldone_%=\n"
        "Lspecial_%=:\n"
        "  s_mov_b64 exec, %[sv]\n"
        "Ldone_%=:\n"
        "  s_waitcnt lgkmcnt(0)\n"
        : [left] "+s"(left), [pos] "+v"(pos), [gsum] "+v"(gsum), [last] "+v"(last_e), [lo] "+v"(rd.lo),
          [hi] "+v"(rd.hi), [b] "+v"(rd.boff), [ra] "+v"(ra), [ird] "+v"(rd.in_rd), [w2] "=&v"(w2), [e] "=&v"(e),
          [nw] "=&v"(nw), [t] "=&v"(t), [sv] "=&s"(sv)
        : [lim] "v"(lim), [rb] "v"(ring_base)
        : "vcc", "scc", "memory");
    return left;
}

// Up to `left` literal-only steps of the writing loop (pass 2), hand-scheduled like
// seg_count_group: decode one entry, append its 1-2 bytes to the 4-byte accumulator, move the
// accumulator to the output ring when it is full (a partial one stays in its register).
// Entered only while no lane is filling a run.  Returns the steps still to do when some running
// lane meets a special entry (nothing of that step applied).
struct SegWriter {
    uint32_t acc, sh, vposw;
};
__device__ __forceinline__ uint32_t seg_write_group(uint32_t left, uint32_t end2, uint32_t ring_base, uint32_t out_base,
                                                    SegReader& rd, SegWriter& wr, uint32_t& pos, uint32_t& last_e) {
    static_assert(kSegInWords == 16 && kSegOutWords == 16, "ring word mask 0xf00 below");
    uint32_t ra = ring_base | ((rd.in_rd << 8) & 0xf00u);
    uint32_t wa = out_base | ((wr.vposw << 8) & 0xf00u);
    uint32_t w2, e, nw, t, v;
    uint64_t sv, sr;
    asm volatile(
        "  s_waitcnt lgkmcnt(0)\n"  // nothing of the compiler's may be in flight: the counted waits below assume it
        "Lstep_%=:\n"
        "  v_alignbit_b32 %[w2], %[hi], %[lo], %[b]\n"
        "  v_and_b32 %[t], 0x3ffc, %[w2]\n"
        "  ds_read_b32 %[e], %[t]\n"
        "  ds_read_b32 %[nw], %[ra]\n"
        "  v_cmp_lt_u32 vcc, %[pos], %[end2]\n"
        "  s_and_saveexec_b64 %[sv], vcc\n"
        "  s_waitcnt lgkmcnt(1)\n"
        "  v_and_b32 %[t], 0xe0, %[e]\n"
        "  v_cmp_ne_u32 vcc, 0, %[t]\n"
        "  s_cbranch_vccnz Lspecial_%=\n"
        "  v_and_b32 %[t], 0xff, %[e]\n"
        "  v_mov_b32 %[last], %[e]\n"
        "  v_add_u32 %[pos], %[pos], %[t]\n"
        "  v_add_u32 %[b], %[b], %[t]\n"
        // append the literal bytes (entry bits 31:16) at byte sh / 8 of the accumulator
        "  v_lshrrev_b32 %[v], 16, %[e]\n"
        "  v_lshlrev_b32 %[t], %[sh], %[v]\n"
        "  v_or_b32 %[acc], %[acc], %[t]\n"
        "  v_sub_u32 %[t], 32, %[sh]\n"
        "  v_lshrrev_b32 %[v], %[t], %[v]\n"  // bytes that did not fit (only used when the word fills up, sh > 0)
        "  v_lshrrev_b32 %[t], 8, %[e]\n"
        "  v_and_b32 %[t], 0xff, %[t]\n"
        "  v_add_u32 %[sh], %[sh], %[t]\n"
        "  v_cmp_lt_u32 vcc, 31, %[sh]\n"
        "  v_and_b32 %[sh], 31, %[sh]\n"
        "  s_mov_b64 %[sr], exec\n"
        "  s_and_b64 exec, exec, vcc\n"  // lanes whose accumulator is full: it goes to the ring
        "  ds_write_b32 %[wa], %[acc]\n"
        "  v_mov_b32 %[acc], %[v]\n"
        "  v_add_u32 %[t], 0x100, %[wa]\n"
        "  v_and_b32 %[t], 0xf00, %[t]\n"
        "  v_or_b32 %[wa], %[ob], %[t]\n"
        "  v_add_u32 %[vposw], 1, %[vposw]\n"
        "  s_mov_b64 exec, %[sr]\n"
        "  v_cmp_lt_u32 vcc, 31, %[b]\n"
        "  v_and_b32 %[b], 31, %[b]\n"
        "  s_and_b64 exec, exec, vcc\n"  // lanes that crossed a dword of input
        "  s_waitcnt lgkmcnt(0)\n"
        "  v_mov_b32 %[lo], %[hi]\n"
        "  v_mov_b32 %[hi], %[nw]\n"
        "  v_add_u32 %[t], 0x100, %[ra]\n"
        "  v_and_b32 %[t], 0xf00, %[t]\n"
        "  v_or_b32 %[ra], %[rb], %[t]\n"
        "  v_add_u32 %[ird], 1, %[ird]\n"
        "  s_mov_b64 exec, %[sv]\n"
        "  s_sub_u32 %[left], %[left], 1\n"
        "  s_cmp_lg_u32 %[left], 0\n"
        "  s_cbranch_scc1 Lstep_%=\n"
        "  s_branch Ldone_%=\n"
        "Lspecial_%=:\n"
        "  s_mov_b64 exec, %[sv]\n"
        "Ldone_%=:\n"
        "  s_waitcnt lgkmcnt(0)\n"
        : [left] "+s"(left), [pos] "+v"(pos), [last] "+v"(last_e), [lo] "+v"(rd.lo), [hi] "+v"(rd.hi),
          [b] "+v"(rd.boff), [ra] "+v"(ra), [ird] "+v"(rd.in_rd), [acc] "+v"(wr.acc), [sh] "+v"(wr.sh),
          [wa] "+v"(wa), [vposw] "+v"(wr.vposw), [w2] "=&v"(w2), [e] "=&v"(e), [nw] "=&v"(nw), [t] "=&v"(t),
          [v] "=&v"(v), [sv] "=&s"(sv), [sr] "=&s"(sr)
        : [end2] "v"(end2), [rb] "v"(ring_base), [ob] "v"(out_base)
        : "vcc", "scc", "memory");
    return left;
}

// The long loop of pass 1: from s.pos to `stop_at`, counting every byte.
// Outer loop = one global-memory event, inner loop = kSegSteps look-ups that touch neither the
// in-flight load registers nor global memory, so the compiler keeps waits and copies out of it.
// A step is the literal fast path (a dozen byte-field operations under the execution mask) unless
// some running lane looks at a run / end-of-block / impossible entry.  A lane runs while
// pos < lim; halting sets lim = 0.
__device__ __forceinline__ uint32_t seg_count_scan(const uint32_t* lit, SegReader& rd, uint32_t limit, bool active,
                                                   uint32_t stop_at, SegScan& s) {
    uint32_t lim = (active && s.stop == 0) ? stop_at : 0u;
    uint32_t iter = 0;
    const uint32_t ring_base = lds_offset(rd.ring) + 4 * rd.lane_off;
    if (s.pos < lim) rd.refill_now();
    while (__any(s.pos < lim)) {
        // nothing is drained in this pass, so one memory event serves two groups of steps (a pair of
        // chunks = 8 dwords per event covers the <= 2 * kSegNeed - 1 dwords they can consume)
        rd.events(s.pos < lim, 2 * kSegNeed - 1);
      for (int half = 0; half < 2; half++) {
        uint32_t gsum = 0;  // sum of the literal entries of this group: byte 1 = 8 x bytes (<= 8 x 16)
        uint32_t left = kSegSteps;
        while (left) {
            left = seg_count_group(left, lim, ring_base, rd, s.pos, gsum, s.last_e);
            if (left == 0) break;
            // general step (selects only): some running lane looks at a run / end-of-block / impossible entry
            left--;
            iter++;
            const uint32_t win = rd.window();
            const uint32_t e = lit[win & (kLitSize - 1)];
            const uint32_t nw = rd.peek();
            const bool running = s.pos < lim;
            const uint32_t used = seg_used(e);
            const bool is_run = (e & SE_RUN) != 0;
            const SegRun r = seg_run(e, win);
            const bool fault = (e & SE_BAD) != 0 || (is_run && r.bad_dist) || s.pos + used > limit;
            const bool step = running && !fault && (e & SE_EOB) == 0;
            const bool halt = running && !step;
            s.stop = halt ? (fault ? 2u : 1u) : s.stop;
            s.eob_bits = halt ? used : s.eob_bits;
            lim = halt ? 0u : lim;
            s.count8 += step ? (is_run ? 8 * r.length : seg_n8(e)) : 0u;
            s.last_e = (step && seg_n8(e) != 0) ? e : s.last_e;
            const uint32_t adv = step ? used : 0u;
            s.pos += adv;
            rd.advance(adv, nw);
        }
        s.count8 += (gsum >> 8) & 0xFF;
        iter += kSegSteps;
      }
        // the fast path checks neither of these per step; both are monotone within a group
        const bool over = active && s.stop == 0 && (s.pos > limit || rd.starved());
        s.stop = over ? 2u : s.stop;
        lim = over ? 0u : lim;
    }
    return iter;
}

// What pass 2 needs from pass 1 (per lane unless noted).
struct SegPlan {
    uint32_t start;   // real chain start of the lane (segment-relative bit)
    uint32_t end2;    // chain end (>= segment length) or the end-of-block position; 0 = lane has nothing to do
    uint32_t obase;   // output offset of the lane's first byte
    uint32_t count;   // output bytes of the lane
    uint32_t last_e;  // entry of the last literal token in front of the lane (0: none)
    uint32_t total;   // uniform: output bytes of the stream
    uint32_t tb;      // uniform: stream byte position of the Adler-32 trailer
};
constexpr uint32_t kSegPlanWords = 5 * kWave + 4;  // per stream in global scratch: 5 lane arrays + {ok, total, tb, -}

// Passes 1 + check + scan of one stream.  False: the stream was left PENDING (or is out of range).
__device__ __forceinline__ bool segments_plan(const SegArgs& a, const uint32_t* lit, uint32_t* in_ring, const uint32_t lane_off,
                                              const uint64_t sid, SegPlan& plan) {
    const int lane = threadIdx.x & (kWave - 1);
    if (sid >= a.n) return false;

    // ---- stream set-up (uniform) ----
    const uint64_t i0 = a.in_off[sid], i1 = a.in_off[sid + 1];
    const uint64_t o0 = a.out_off[sid], o1 = a.out_off[sid + 1];
    const uint8_t* in = a.in + i0;
    const uint8_t* buf_hi = a.in + a.in_off[a.n];
    const uint64_t ilen = i1 - i0, ocap = o1 - o0;
    bool ours = ilen < (1ull << 28) && ocap < (1ull << 31) &&
                ilen * 8 >= a.canon_bits + 44ull;  // room for an end-of-block symbol and the Adler-32
    const uint32_t in_bits = (uint32_t)(ilen * 8);
    const uint32_t cap = (uint32_t)ocap;
    // canonical prefix: lane k compares stream dword k
    if (ours) {
        bool mismatch = false;
        if (lane < 14) {
            uint32_t v = 0;
            const uint8_t* p = in + 4 * lane;
            for (int k = 0; k < 4; k++) v |= (uint32_t)p[k] << (8 * k);
            if (lane == 13) v &= (1u << (a.canon_bits - 13 * 32)) - 1;
            mismatch = v != a.canon_hdr[lane];
        }
        ours = !__any(mismatch);
    }
    ours = ours && lds_offset(lit) == 0;  // the hand-scheduled loops address the table from LDS offset 0
    if (!ours) {
        if (lane == 0) seg_leave_pending(a, sid);
        return false;
    }
    // ---- segments: equal bit ranges of the block data (the trailer bits ride along), one per lane;
    //      short streams use fewer lanes so that a segment stays several windows long ----
    const uint32_t data_bits = in_bits - a.canon_bits;
    // Synchronisation window: chains of short codes (many tokens per bit) join sooner.  The slot
    // size stands in for the output length: <= 4.25 stream bits per output byte -> the short window.
    // (A wrong guess only costs time: lanes that do not land re-count.)
    const uint32_t window = (uint64_t)in_bits * 4 <= (uint64_t)cap * 17 ? (uint32_t)kSegWindowShort : (uint32_t)kSegWindow;
    const uint32_t nseg = min((uint32_t)kWave, max(1u, (data_bits + kSegMinBits - 1) / kSegMinBits));
    const uint32_t seg = (data_bits + nseg - 1) / nseg;
    const uint32_t seg_bit0 = a.canon_bits + (uint32_t)lane * seg;  // first bit of this lane's segment
    const bool in_range = (uint32_t)lane < nseg && seg_bit0 < in_bits;
    const uint32_t limit = in_range ? in_bits - seg_bit0 : 0;       // tokens must end at or before this

    SegReader rd;
    rd.ring = in_ring;
    rd.lane_off = lane_off;
    rd.buf_lo = a.in;
    rd.buf_hi = buf_hi;
    rd.gp = in;
    rd.in_wr = rd.in_rd = 0;
    rd.lo = rd.hi = rd.boff = 0;
    for (int k = 0; k < kSegChunk; k++) rd.pend_a.w[k] = rd.pend_b.w[k] = 0;
    rd.has_a = rd.has_b = false;

    // ---- pass 1: guessed chain from bit 0 of the segment; count from where it leaves the window ----
    SegScan tail;  // becomes: the chain from the window's end (x0) to the segment's end
    tail.pos = 0;
    tail.count8 = 0;
    tail.last_e = 0;
    tail.stop = 0;
    tail.eob_bits = 0;
    SEGTIME(0);
    if (in_range) rd.start(in, seg_bit0);
    {
        uint32_t itw = seg_window_scan<true>(lit, a.canon_lit, rd, limit, in_range, window, tail);
        (void)itw;
        SEGDBG(6, itw);
        SEGDBG(1, 0);
        SEGDBG(2, 0);
        SEGDBG(3, 0);
        SEGDBG(5, seg);
    }
    SEGTIME(1);
    uint32_t x0 = tail.stop == 0 ? tail.pos : 0;  // where the guessed chain left the window (0: it did not)
    {
        uint32_t it1 = seg_count_scan(lit, rd, limit, in_range, seg, tail);
        (void)it1;
        SEGDBG(0, it1);
    }

    SEGTIME(2);
    // ---- check: real start from the left neighbour, decode the window, must land on x0 ----
    SegScan head;
    head.pos = head.count8 = head.stop = head.eob_bits = head.last_e = 0;
    uint32_t start = 0;        // real chain start of this lane (segment-relative)
    uint32_t cur_start = ~0u;  // start the current `head` was computed for
    bool giveup = false;
    for (int round = 0; round < 6; round++) {
        const uint32_t prev_end = __shfl_up(tail.pos, 1, kWave);
        const uint32_t prev_stop = __shfl_up(tail.stop, 1, kWave);
        start = lane == 0 ? 0 : prev_end - seg;
        const bool have_in = lane == 0 || (prev_stop == 0 && prev_end >= seg);
        const bool need = in_range && have_in && start != cur_start;
        if (!__any(need)) break;
        if (round == 5) giveup = true;
        // head: real chain through the window, every byte counted
        if (need) {
            head.pos = start;
            head.count8 = 0;
            head.last_e = 0;
            head.stop = 0;
            head.eob_bits = 0;
            rd.start(in, seg_bit0 + start);
        }
        {
            uint32_t ith = seg_window_scan<false>(lit, a.canon_lit, rd, limit, need, window, head);
            (void)ith;
            SEGDBG_ADD(1, ith);
            SEGDBG_ADD(3, 1);
        }
        // landed on the guessed chain?  then the counted tail is the real tail
        const bool stopped_in_head = need && head.stop != 0;
        const bool redo = need && head.stop == 0 && (head.pos != x0 || x0 == 0);
        if (stopped_in_head) {  // end-of-block / bad token inside the window: there is no tail
            tail = head;
            tail.count8 = 0;
            tail.last_e = 0;
            x0 = head.pos;
        }
        if (__any(redo)) {  // rare: re-count this segment from the landing point (the reader is there)
            if (redo) {
                tail.pos = head.pos;
                tail.count8 = 0;
                tail.last_e = 0;
                tail.stop = 0;
                tail.eob_bits = 0;
                x0 = head.pos;
            }
            uint32_t itr = seg_count_scan(lit, rd, limit, redo, seg, tail);
            (void)itr;
            SEGDBG_ADD(2, itr);
        }
        if (need) cur_start = start;
    }
    SEGTIME(3);
    // ---- who is live: lanes up to the first stop on a verified chain ----
    const bool verified = in_range && cur_start == start;
    const uint64_t stop_mask = __ballot(verified && tail.stop != 0);
    const uint64_t unver_mask = __ballot(!verified);
    const int stop_lane = stop_mask ? __ffsll((unsigned long long)stop_mask) - 1 : kWave;
    const int first_unver = unver_mask ? __ffsll((unsigned long long)unver_mask) - 1 : kWave;
    const bool live = lane <= stop_lane;
    const uint32_t stop_kind = __shfl(tail.stop, stop_lane & (kWave - 1), kWave);
    // the stream must end with an end-of-block on a verified chain
    bool ok = !giveup && stop_lane < kWave && first_unver > stop_lane && stop_kind == 1;
    // bytes per lane: head (window) + tail; summed in 64 bits (a hostile stream can claim anything)
    const uint32_t count = live ? (head.count8 + tail.count8) >> 3 : 0;
    unsigned long long incl = count;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        unsigned long long y = __shfl_up(incl, o, kWave);
        if (lane >= o) incl += y;
    }
    const unsigned long long total64 = __shfl(incl, kWave - 1, kWave);
    ok = ok && total64 <= cap;
    const uint32_t total = (uint32_t)total64;
    const uint32_t obase = (uint32_t)incl - count;
    // last literal token of every lane's chain -> the byte a leading run of the right neighbour repeats
    uint32_t carry = live ? (tail.last_e != 0 ? tail.last_e : head.last_e) : 0u;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        uint32_t y = __shfl_up(carry, o, kWave);
        if (lane >= o && carry == 0) carry = y;
    }
    uint32_t last_e = __shfl_up(carry, 1, kWave);  // 0: nothing to repeat yet
    if (lane == 0) last_e = 0;
    // trailer position: right after the end-of-block symbol
    const uint32_t eob_end = __shfl(seg_bit0 + tail.pos + tail.eob_bits, stop_lane & (kWave - 1), kWave);
    const uint32_t tb = (eob_end + 7) >> 3;
    ok = ok && (uint64_t)tb * 8 + 32 <= in_bits;
    SEGDBG(8, (uint32_t)stop_lane | ((uint32_t)first_unver << 8) | (stop_kind << 16) | ((giveup ? 1u : 0u) << 24));
    SEGDBG(9, total);
    if (!ok) {
        if (lane == 0) seg_leave_pending(a, sid);
        return false;
    }

    plan.start = start;
    plan.end2 = live ? tail.pos : 0u;
    plan.obase = obase;
    plan.count = count;
    plan.last_e = last_e;
    plan.total = total;
    plan.tb = tb;
    return true;
}

// Pass 2 of one stream (plan from segments_plan): decode the real chains again, this time writing.
__device__ __forceinline__ void segments_write(const SegArgs& a, SegLds& L, const uint64_t sid, const SegPlan& plan) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = threadIdx.x / kWave;
    const uint32_t* lit = L.lit;
    const uint32_t lane_off = (uint32_t)wid * (kSegInWords * kWave) + (uint32_t)lane;  // ring slot of word 0
    const uint64_t i0 = a.in_off[sid], i1 = a.in_off[sid + 1];
    const uint8_t* in = a.in + i0;
    uint8_t* op = a.out + a.out_off[sid];
    const uint32_t in_bits = (uint32_t)((i1 - i0) * 8);
    const uint32_t data_bits = in_bits - a.canon_bits;
    const uint32_t nseg = min((uint32_t)kWave, max(1u, (data_bits + kSegMinBits - 1) / kSegMinBits));
    const uint32_t seg = (data_bits + nseg - 1) / nseg;
    const uint32_t seg_bit0 = a.canon_bits + (uint32_t)lane * seg;
    const uint32_t start = plan.start, obase = plan.obase, count = plan.count, total = plan.total, tb = plan.tb;
    uint32_t last_e = plan.last_e;
    const bool live = plan.end2 != 0 || count != 0;
    SegReader rd;
    rd.ring = L.in_ring;
    rd.lane_off = lane_off;
    rd.buf_lo = a.in;
    rd.buf_hi = a.in + a.in_off[a.n];
    rd.gp = in;
    rd.in_wr = rd.in_rd = 0;
    rd.lo = rd.hi = rd.boff = 0;
    for (int k = 0; k < kSegChunk; k++) rd.pend_a.w[k] = rd.pend_b.w[k] = 0;
    rd.has_a = rd.has_b = false;

    // ---- pass 2: decode the real chain again, this time writing ----
    // Every token on the chain was validated by pass 1 / the check, so nothing is re-checked here
    // except what pass 1 cannot know: a run with nothing before it.
    uint32_t end2 = plan.end2;  // chain end (>= seg) or the end-of-block position; 0 = halted
    uint32_t pos = start;
    // bytes in front of this lane's first byte in its 16-B line of global memory (the slot itself may
    // start anywhere; the first and the last line of a lane are stored byte by byte)
    const uint32_t pad = (uint32_t)(reinterpret_cast<uintptr_t>(op) + obase) & 15;
    uint8_t* const line0 = op + obase - pad;    // 16-B aligned; may lie in front of the slot for lane 0
    uint32_t vposw = pad >> 2;                  // virtual position in dwords: words already in the ring
    uint32_t acc = 0, sh = 8 * (pad & 3);       // 4-byte accumulator holding sh / 8 bytes (the rest is zero)
    uint32_t vstored = 0;                       // virtual bytes stored to global (multiple of 16)
    uint32_t ad_a = 0, ad_b = 0, blocks = 0;    // per-lane Adler partial over its own bytes
    uint32_t fill = 0;
    bool bad2 = false;
    uint32_t* const oring = L.out_ring;
    oring[seg_slot(lane_off, 0)] = 0;           // the words in front of the first byte are pad zeros
    oring[seg_slot(lane_off, 1)] = 0;
    oring[seg_slot(lane_off, 2)] = 0;
    const uint32_t vend = pad + count;          // virtual end

    // 16 virtual bytes at vs: to global memory (unless a neighbour lane has stored them, see drain) and
    // into the checksum
    auto store_piece = [&](uint32_t vs, bool stored_already = false) __attribute__((always_inline)) {
        const uint32_t w = vs >> 2;
        uint4 q;
        q.x = oring[seg_slot(lane_off, w + 0)];
        q.y = oring[seg_slot(lane_off, w + 1)];
        q.z = oring[seg_slot(lane_off, w + 2)];
        q.w = oring[seg_slot(lane_off, w + 3)];
        if (stored_already) {
        } else if (vs >= pad && vs + 16 <= vend) {
            *reinterpret_cast<uint4*>(line0 + vs) = q;
        } else {  // first / last line of this lane: only its own bytes
            for (uint32_t k = 0; k < 16; k++) {
                const uint32_t word = k < 4 ? q.x : (k < 8 ? q.y : (k < 12 ? q.z : q.w));
                if (vs + k >= pad && vs + k < vend) line0[vs + k] = (uint8_t)(word >> (8 * (k & 3)));
            }
        }
        // Adler-32 partial (pad bytes are zero and come first, so they change nothing)
        const uint32_t s = bytesum4(q.x) + bytesum4(q.y) + bytesum4(q.z) + bytesum4(q.w);
        uint32_t u = bytedot4(q.x, 0x0d0e0f10u, 0);
        u = bytedot4(q.y, 0x090a0b0cu, u);
        u = bytedot4(q.z, 0x05060708u, u);
        u = bytedot4(q.w, 0x01020304u, u);
        ad_b += 16 * ad_a + u;
        ad_a += s;
        if (++blocks == 128) {
            ad_a %= kAdlerMod;
            ad_b %= kAdlerMod;
            blocks = 0;
        }
    };
    auto drain_all = [&]() __attribute__((always_inline)) {  // every complete 16-B line
        while (__any(4 * vposw - vstored >= 16)) {
            if (4 * vposw - vstored >= 16) {
                store_piece(vstored);
                vstored += 16;
            }
        }
    };
    // Lines leave in 32-B aligned pairs (two adjacent 16-B stores back to back) so that whole
    // 32-B sectors reach the L2 together; a lane whose first line is the upper half of a sector
    // sends that one alone.  < 32 B stay behind, + <= 32 B per group of steps: fits the 64-B ring.
    const bool odd_first = (reinterpret_cast<uintptr_t>(line0) & 16) != 0;
    auto drain = [&]() __attribute__((always_inline)) {
        for (;;) {
            const uint32_t avail = 4 * vposw - vstored;
            const bool single = odd_first && vstored == 0 && avail >= 16;
            const bool pair = !single && avail >= 32;
            if (!__any(single || pair)) break;
            // An interior pair is stored by two lanes in ONE instruction (lane l and l ^ 1 each write
            // 16 of the 32 bytes: first the pairs of the even lanes, then those of the odd lanes), so
            // the memory pipeline sees 32-B requests instead of twice 16 B.
            const bool coop = pair && vstored >= pad && vstored + 32 <= vend;
            {
                uint8_t* const my_ptr = line0 + vstored;
                const uint32_t my_w = vstored >> 2;
                const uint32_t p_lo = __shfl_xor((uint32_t)reinterpret_cast<uintptr_t>(my_ptr), 1, kWave);
                const uint32_t p_hi = __shfl_xor((uint32_t)(reinterpret_cast<uintptr_t>(my_ptr) >> 32), 1, kWave);
                const uint32_t p_w = __shfl_xor(my_w, 1, kWave);
                const bool p_coop = __shfl_xor((int)coop, 1, kWave) != 0;
                uint8_t* const p_ptr = reinterpret_cast<uint8_t*>(((uintptr_t)p_hi << 32) | p_lo);
#pragma unroll
                for (int par = 0; par < 2; par++) {
                    const bool owner = (lane & 1) == par;
                    if (owner ? coop : p_coop) {
                        const uint32_t w = owner ? my_w : p_w + 4;
                        const uint32_t sl = owner ? lane_off : (lane_off ^ 1u);
                        uint4 q;
                        q.x = oring[seg_slot(sl, w + 0)];
                        q.y = oring[seg_slot(sl, w + 1)];
                        q.z = oring[seg_slot(sl, w + 2)];
                        q.w = oring[seg_slot(sl, w + 3)];
                        *reinterpret_cast<uint4*>(owner ? my_ptr : p_ptr + 16) = q;
                    }
                }
            }
            if (single || pair) {
                store_piece(vstored, coop);
                vstored += 16;
            }
            if (pair) {
                store_piece(vstored, coop);
                vstored += 16;
            }
        }
    };
    // A long dist-1 run (src/decompress.rs:793-801 fills it with one byte): bring the lane to a
    // 16-B line boundary through the ring, then store whole lines of the byte directly; their
    // Adler-32 contribution has a closed form.
    auto bulk_fill = [&](uint32_t c) __attribute__((always_inline)) {
        const uint32_t c4 = c * 0x01010101u;
        // complete the accumulator, then whole words up to the line boundary
        acc |= c4 << sh;
        oring[seg_slot(lane_off, vposw)] = acc;
        vposw++;
        fill -= 4 - (sh >> 3);
        acc = 0;
        sh = 0;
        while (vposw & 3) {
            oring[seg_slot(lane_off, vposw)] = c4;
            vposw++;
            fill -= 4;
        }
        while (4 * vposw != vstored) {  // the complete lines waiting in the ring
            store_piece(vstored);
            vstored += 16;
        }
        uint32_t lines = min(fill, vend - vstored) >> 4;
        const uint32_t m = lines << 4;
        const uint4 q = make_uint4(c4, c4, c4, c4);
        uint8_t* dst = line0 + vstored;
        for (; lines; lines--, dst += 16) *reinterpret_cast<uint4*>(dst) = q;
        // m bytes of value c: a' = a + m c ; b' = b + m a + c m (m + 1) / 2
        ad_a %= kAdlerMod;
        ad_b %= kAdlerMod;
        blocks = 0;
        const uint64_t tri = ((uint64_t)m * (m + 1) / 2) % kAdlerMod;
        ad_b = (uint32_t)((ad_b + (uint64_t)(m % kAdlerMod) * ad_a + tri * c) % kAdlerMod);
        ad_a = (uint32_t)((ad_a + (uint64_t)m * c) % kAdlerMod);
        vstored += m;
        vposw += m >> 2;
        fill -= m;
    };

    SEGTIME(4);
    if (live) rd.start(in, seg_bit0 + pos);
    uint32_t iter = 0;
    const uint32_t ring_base = lds_offset(L.in_ring) + 4 * lane_off;
    const uint32_t out_base = lds_offset(L.out_ring) + 4 * lane_off;
    // A lane runs while pos < end2 or a run is being filled.
    // Outer loop = input event + two x (drain + kSegSteps steps of <= 4 B each).
    while (__any(pos < end2 || fill != 0)) {
        // one memory event per two groups of steps (as in the counting pass); the output side is
        // drained before every group (<= 32 B are produced per group)
        rd.events(pos < end2 || fill != 0, 2 * kSegNeed - 1);
      for (int half = 0; half < 2; half++) {
        drain();
        if (__any(fill >= kSegBulkFill)) {
            if (fill >= kSegBulkFill) bulk_fill(seg_lastlit(last_e));
        }
        uint32_t left = kSegSteps;
        iter += kSegSteps;
        while (left) {
            if (!__any(fill != 0)) {  // nobody is filling a run: literal steps until a special entry turns up
                SegWriter wr{acc, sh, vposw};
                left = seg_write_group(left, end2, ring_base, out_base, rd, wr, pos, last_e);
                acc = wr.acc;
                sh = wr.sh;
                vposw = wr.vposw;
                if (left == 0) break;
            }
            left--;
            {
                // general step (selects only): a token of any kind, or 4 bytes of a run in progress
                const uint32_t win = rd.window();
                const uint32_t e = lit[win & (kLitSize - 1)];
                const uint32_t nw = rd.peek();
                const bool filling = fill != 0;
                const bool dec = !filling && pos < end2;  // this lane decodes a token now
                const bool is_run = (e & SE_RUN) != 0;
                const SegRun r = seg_run(e, win);
                const uint32_t n8_lit = dec ? seg_n8(e) : 0u;
                const bool bad_now = dec && ((e & (SE_BAD | SE_EOB)) != 0 || (is_run && (last_e == 0 || r.bad_dist)));
                bad2 = bad2 || bad_now;
                last_e = n8_lit ? e : last_e;
                const uint32_t nf = min(fill, 4u);
                uint32_t vf = seg_lastlit(last_e) * 0x01010101u;
                vf = nf < 4 ? (vf & ((1u << (8 * nf)) - 1)) : vf;
                const uint32_t n8 = filling ? 8 * nf : n8_lit;
                const uint32_t v = filling ? vf : (n8_lit ? e >> 16 : 0u);
                fill = filling ? fill - nf : ((dec && is_run && !bad_now) ? r.length : 0u);
                const uint32_t used = dec ? seg_used(e) : 0u;
                pos += used;
                rd.advance(used, nw);
                end2 = bad_now ? 0u : end2;
                const uint64_t t = (uint64_t)v << sh;
                acc |= (uint32_t)t;
                // the ring slot at vposw is always free: the (possibly partial) accumulator is
                // written there every time and only counts once it is full
                oring[seg_slot(lane_off, vposw)] = acc;
                const uint32_t tot = sh + n8;
                const bool full = tot >= 32;
                acc = full ? (uint32_t)(t >> 32) : acc;
                vposw += full ? 1u : 0u;
                sh = tot & 31;
            }
        }
      }
        if (live && rd.starved()) {  // cannot happen (events() keeps the ring ahead); stop rather than decode garbage
            bad2 = true;
            end2 = 0;
            fill = 0;
        }
    }
    (void)iter;
    SEGTIME(5);
    SEGDBG(4, iter);
    SEGDBG(7, total);
    // ---- tail of every lane: the last (partial) line ----
    drain_all();
    if (live) {
        // the loose bytes go into the ring as a final word; zero the rest of that 16-B line
        oring[seg_slot(lane_off, vposw)] = acc;
        for (uint32_t w = vposw + 1; (w & 3) != 0; w++) oring[seg_slot(lane_off, w)] = 0;
        if (vstored < vend) {
            store_piece(vstored);
            // the last line was summed as 16 bytes; it holds only 16 - z of ours followed by z zeros
            const uint32_t z = vstored + 16 - vend;
            ad_a %= kAdlerMod;
            ad_b %= kAdlerMod;
            ad_b = (ad_b + kAdlerMod - (uint32_t)(((uint64_t)z * ad_a) % kAdlerMod)) % kAdlerMod;
        }
    }
    const bool wrong_count = live && (4 * vposw + (sh >> 3) != vend);
    {
        const uint64_t m_bad = __ballot(bad2), m_wrong = __ballot(wrong_count);
        (void)m_bad;
        (void)m_wrong;
        SEGDBG(10, (uint32_t)m_bad);
        SEGDBG(11, (uint32_t)(m_bad >> 32));
        SEGDBG(12, (uint32_t)m_wrong);
        SEGDBG(13, (uint32_t)(m_wrong >> 32));
    }
    if (__any(bad2 || wrong_count)) {
        if (lane == 0) seg_leave_pending(a, sid);
        return;
    }
    // ---- combine the Adler-32 partials: A = 1 + sum a_i ; B = total + sum (b_i + rest_i * a_i) ----
    ad_a %= kAdlerMod;
    ad_b %= kAdlerMod;
    const uint32_t rest = total - obase - count;
    uint32_t pa = live ? ad_a : 0;
    uint32_t pb = live ? (uint32_t)(((uint64_t)ad_b + (uint64_t)(rest % kAdlerMod) * ad_a) % kAdlerMod) : 0;
    pa = wave_sum_u32(pa);
    pb = wave_sum_u32(pb);
    const uint32_t A = (1u + pa) % kAdlerMod;
    const uint32_t B = (uint32_t)(((uint64_t)total + pb) % kAdlerMod);
    const uint32_t adler = (B << 16) | A;
    SEGDBG(14, adler);
    SEGTIME(6);
    if (lane == 0) {
        // src/decompress.rs:306-326: byte boundary, then the big-endian Adler-32
        uint32_t stored = ((uint32_t)in[tb] << 24) | ((uint32_t)in[tb + 1] << 16) | ((uint32_t)in[tb + 2] << 8) |
                          (uint32_t)in[tb + 3];
        if (stored == adler || (a.flags & 1u)) {
            a.status[sid] = ST_OK;
            a.out_len[sid] = total;
            if (a.adler) a.adler[sid] = adler;
        } else {
            seg_leave_pending(a, sid);  // the exact kernels report WrongChecksum
        }
    }
}

// Both halves on one stream (the fused kernel).
__device__ __forceinline__ void segments_decode(const SegArgs& a, SegLds& L, const uint64_t sid) {
    const uint32_t lane_off = (threadIdx.x / kWave) * (uint32_t)(kSegInWords * kWave) + (threadIdx.x & (kWave - 1));
    SegPlan plan;
    if (segments_plan(a, L.lit, L.in_ring, lane_off, sid, plan)) segments_write(a, L, sid, plan);
}

}  // namespace fdh
