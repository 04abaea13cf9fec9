// inflate_segments.h -- segment-parallel decode of one ultra-fast-format stream per wavefront.
//
// A stream that starts with the ultra-fast encoder's fixed prefix (reference
// src/compress/ultrafast.rs:82-88) is one final dynamic block with a known table.  Its block
// data is cut into up to 64 equal bit ranges ("segments", >= kSegMinBits each; short streams use
// fewer lanes), one per lane, and every lane runs the reference's inner loop
// (src/decompress.rs:645-830: table look-up, literals, dist-1 run) sequentially over its own
// segment:
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
// (one per two groups of steps, two adjacent 16-B loads at a time, committed an event later), so
// the hot loops never wait on memory.  The hot loops themselves are hand-scheduled groups of
// kSegSteps table look-ups (seg_count_group / seg_write_group) that run, under the execution
// mask of the lanes that are at least a group away from their end, WITHOUT any per-step branch or
// mask change: a table entry retires up to three literals; a run / end-of-block / impossible
// entry is encoded as "0 bits, 0 bytes", so a lane that meets one simply marks time until the
// pair of steps ends, and the select-only C++ step that follows every group decodes that token.
// Anything unusual -- not canonical, a bad / truncated token, a full slot, a checksum
// mismatch -- leaves the stream PENDING for the exact wave-per-stream kernels.
#pragma once
#include "inflate_tables.h"

namespace fdh {

#ifdef FDH_DEBUG_TILES
__device__ uint32_t g_segdbg[64 * 16];
__device__ uint32_t g_segdbg2[16 * 16];
__device__ uint32_t g_handed[65537];  // [i] = times stream i was handed out, [65536] = hand-outs with a partial EXEC
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
constexpr int kSegSteps = kSegInWords / 2;  // table look-ups of a fast group / of a stretch of the window walks
constexpr uint32_t kSegPairs = kSegSteps / 2;               // a fast group = kSegPairs pairs of look-ups
constexpr uint32_t kSegGroupBits = kSegSteps * kLitBits;    // most stream bits a fast group consumes (literal steps only)
constexpr uint32_t kSegTokenBits = 18;                      // longest token: 12-bit code + 5 extra bits + 1 distance bit
constexpr uint32_t kSegNeed = (kSegSteps * kSegTokenBits + 31) / 32;  // dwords a stretch of general steps can consume
// dwords one half of the hot loops can consume (a fast group + one general step), + 1 for the
// word the groups prefetch
constexpr uint32_t kSegHalfNeed = (kSegGroupBits + kSegTokenBits + 31) / 32 + 1;
constexpr uint32_t kSegMinBits = 768;   // aim: no segment shorter than this (fewer lanes are used instead)
constexpr uint32_t kSegBulkFill = 64;             // runs at least this long are stored line by line
constexpr uint32_t kNoByte = 0x100;  // "no literal seen yet"
// What an event can always guarantee before the two halves that follow it: with fewer dwords than
// this in the ring a pair of chunks fits, so waiting for the pair in flight gets the lane there.
constexpr uint32_t kSegEventNeed = kSegInWords - 2 * kSegChunk;
static_assert(kSegHalfNeed <= kSegEventNeed, "one half must be able to run right after an event");
static_assert((2 * kSegGroupBits + kSegTokenBits + 31) / 32 + 1 <= kSegEventNeed, "a double group + a general step fit what an event guarantees");

// Table entry of this kernel (built from the device layout of inflate_tables.h while staging):
//   byte 0    [3:0] stream bits of the whole step, [5:4] literals of the step (1..3), [7:6] kind
//   byte 1-3  kind SK_LIT: the literal bytes in output order (unused ones zero)
// Kind SK_LIT: up to three literals whose codes fit the 12 index bits together.  A step whose
// FIRST symbol is a run length / end-of-block / impossible code has kind != 0 and bits [5:0] = 0
// ("no bits, no literals"): the branch-free groups drop a lane that meets one (v_cmpx) before
// anything of the step is applied, and the general step decodes the token from the entry's upper
// bytes: [11:8] code bits, and for a run [14:12] extra-bit count, [24:16] length base.
enum : uint32_t { SK_LIT = 0x00, SK_RUN = 0x40, SK_EOB = 0x80, SK_BAD = 0xC0, SK_KIND = 0xC0 };
__device__ __forceinline__ uint32_t seg_entry_build(const uint32_t* canon, uint32_t i) {
    uint32_t used = 0, n = 0, lits = 0;
    for (int k = 0; k < 3; k++) {
        // the next symbol at the zero-extended rest of the index: right whenever its code ends
        // inside the index bits (prefix code), which is what `used + len <= kLitBits` checks
        const uint32_t e = canon[i >> used];
        const uint32_t kind = (e >> 4) & 15;
        if (kind != K_LIT1 && kind != K_LIT2) break;
        const uint32_t len = (e >> 24) & 15;  // bits of the first symbol (both kinds, inflate_tables.h)
        if (used + len > (uint32_t)kLitBits) break;
        lits |= ((e >> 8) & 0xFF) << (8 * k);
        used += len;
        n++;
    }
    if (n) return used | (n << 4) | (lits << 8);
    const uint32_t c = canon[i];
    const uint32_t nb = c & 15, kind = (c >> 4) & 15;
    if (kind == K_LEN) return SK_RUN | (nb << 8) | (((c >> 8) & 7) << 12) | ((c >> 16) << 16);  // extra bits <= 5, base <= 258
    if (kind == K_EOB) return SK_EOB | (nb << 8);
    return SK_BAD;
}
__device__ __forceinline__ uint32_t seg_used(uint32_t e) { return e & 15; }
__device__ __forceinline__ uint32_t seg_n8(uint32_t e) { return (e & 0x30) >> 1; }  // 8 x literals
// last literal byte of a literal entry (n literals: byte n)
__device__ __forceinline__ uint32_t seg_lastlit(uint32_t e) { return (e >> seg_n8(e)) & 0xFF; }
__device__ __forceinline__ bool seg_is_lit(uint32_t e) { return (e & SK_KIND) == 0 && (e & 0x30) != 0; }

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
    const uint32_t* canon_len4; // 32 dwords: the code lengths of the literals 0..255, 4 bits each
    const uint32_t* canon_hdr;  // 14 dwords of prefix (last one masked)
    uint32_t canon_bits;
    uint32_t pending;
    uint32_t* list;  // nullable: [0] = count, [4..] = ids of the streams left PENDING (compacted);
                     // [1] = next stream to hand out (persistent wavefronts fetch their work here)
    const uint32_t* src_list;  // nullable: work on these streams only ([0] = count, [4..] = ids)
    uint2* ckpt;     // interval decoder (inflate_seg2.h): checkpoint scratch, kS2CkptPerWave entries per wavefront
    const uint32_t* canon_nl;  // ... and the canonical bookkeeping of the symbols >= 256 (CanonTables::nl)
    const uint32_t* canon_lit2;  // ... and its decode table (CanonTables::lit2)
    uint32_t* list2; // nullable (interval kernel): the list of the kernel BEHIND the segment kernel -- streams without
                     // the ultra-fast prefix go there directly, the segment kernel would only look at them and pass them on
    const uint32_t* order;  // nullable (landing / interval kernel): the order in which the streams are handed out, 2 n words
                            // filled by stream_order_kernel from both ends of each half (SegOrder)
    const uint32_t* order_counts;  // ... and how many streams of each of the four classes
};

// Hand-out order of the landing / interval kernels (stream_order_kernel): four classes by compressed length, longest
// first, so that what the wavefronts take last is short and they finish together.  Class 0 fills order[0 ..) upwards,
// class 1 order[n - 1 ..) downwards, class 2 order[n ..) upwards, class 3 order[2 n - 1 ..) downwards.
struct SegOrder {
    uint32_t n0, n1, n2, n3;
    __device__ __forceinline__ uint32_t total() const { return n0 + n1 + n2 + n3; }
    __device__ __forceinline__ uint32_t at(uint32_t n, uint32_t cur) const {  // index into `order` of the cur-th stream handed out
        if (cur < n0) return cur;
        cur -= n0;
        if (cur < n1) return n - 1 - cur;
        cur -= n1;
        if (cur < n2) return n + cur;
        return 2 * n - 1 - (cur - n2);
    }
};
__device__ __forceinline__ uint32_t seg_order_class(uint64_t len, uint64_t mean) {
    return len >= mean ? 0u : (2 * len >= mean ? 1u : (8 * len >= mean ? 2u : 3u));
}

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
        // all the chunks are requested before the first one is used: one trip to memory, not four
        SegChunk c[kSegInWords / kSegChunk];
#pragma unroll
        for (int k = 0; k < kSegInWords / kSegChunk; k++) c[k] = seg_load(gp + 4 * kSegChunk * k, buf_lo, buf_hi);
#pragma unroll
        for (int k = 0; k < kSegInWords / kSegChunk; k++) put(c[k]);
        gp += 4 * kSegInWords;
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
        // (again: the chunks that fit are requested together, then written to the ring)
        const uint32_t want = ((uint32_t)kSegInWords - (in_wr - in_rd)) / (uint32_t)kSegChunk;
        SegChunk c[kSegInWords / kSegChunk];
#pragma unroll
        for (int k = 0; k < kSegInWords / kSegChunk; k++)
            if ((uint32_t)k < want) c[k] = seg_load(gp + 4 * kSegChunk * k, buf_lo, buf_hi);
#pragma unroll
        for (int k = 0; k < kSegInWords / kSegChunk; k++)
            if ((uint32_t)k < want) put(c[k]);
        gp += 4 * kSegChunk * want;
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
    // One event per stretch keeps up with the average consumption; `need` dwords must be in the
    // ring before the stretch starts, so denser stretches get extra (waiting) events.
    __device__ __forceinline__ void events(bool running, uint32_t need = kSegNeed) {
        event(running);
        for (int x = 0; x < 2 && __any(running && level() < need); x++) event(running);
    }
};

// One table step of any kind, decoded (the general, select-only form of a step).
struct SegTok {
    uint32_t e;       // the table entry (literal steps; a one-literal entry in the single-symbol zone)
    uint32_t used;    // stream bits of the step
    uint32_t n8;      // 8 x literal bytes (0 for a run / end-of-block / impossible token)
    uint32_t run;     // run length (0: not a run)
    bool eob, bad;
};
// `need`: this lane wants the token.  `single`: take the first literal of the step alone (the
// symbol-by-symbol zone at a window's end).  The length of the first literal alone is looked up in
// `len4`, a 256 x 4-bit table kept in ONE VGPR (lane k < 32 holds the lengths of the literals
// 8k .. 8k+7) and read with ds_bpermute -- no LDS memory, no global memory.  Call with all lanes of
// the wavefront active (ds_bpermute reads zero from an inactive lane).
__device__ __forceinline__ SegTok seg_token(const uint32_t* lit, uint32_t len4, uint32_t win, bool need, bool single) {
    const uint32_t idx = win & (kLitSize - 1);
    SegTok t;
    t.e = lit[idx];
    t.used = seg_used(t.e);
    t.n8 = seg_n8(t.e);
    const uint32_t kind = t.e & SK_KIND;
    const uint32_t nb = (t.e >> 8) & 15, ex = (t.e >> 12) & 7, base = (t.e >> 16) & 0x1FF;
    const bool is_run = kind == SK_RUN;
    t.run = is_run ? base + ((win >> nb) & ((1u << ex) - 1)) : 0u;
    t.used = is_run ? nb + ex + 1 : (kind == SK_EOB ? nb : t.used);
    t.eob = kind == SK_EOB;
    // the prefix declares one distance code: '0' = distance 1; it is the run token's last bit
    t.bad = kind == SK_BAD || (is_run && ((win >> (nb + ex)) & 1) != 0);
    const bool first_only = single && t.n8 > 8;
    if (__any(need && first_only)) {
        const uint32_t b1 = (t.e >> 8) & 0xFF;
        const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((b1 >> 3) << 2), (int)len4);
        const uint32_t len1 = (w >> ((b1 & 7) * 4)) & 15;
        t.used = first_only ? len1 : t.used;
        t.n8 = first_only ? 8u : t.n8;
        t.e = first_only ? ((t.e & 0xFF00u) | 0x10u | len1) : t.e;
    }
    return t;
}

// State of one lane's counting scan.
struct SegScan {
    uint32_t pos;      // segment-relative bit position of the next token
    uint32_t count8;   // 8 x output bytes counted so far
    uint32_t last_e;   // entry of the last literal step counted (0 if none)
    uint32_t stop;     // 0 none, 1 end-of-block (pos = its start, eob_bits its length), 2 bad token
    uint32_t eob_bits;
};

// The value of lane ^ 1 (DPP quad_perm [1,0,3,2]: a VALU move, no LDS traffic).  All lanes active.
__device__ __forceinline__ uint32_t swap1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
}

// Byte offset of an LDS object inside the workgroup's LDS allocation.
__device__ __forceinline__ uint32_t lds_offset(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// kSegPairs pairs of look-ups of the counting loop, hand-scheduled, for the lanes the caller has
// enabled (every one of them is >= kSegGroupBits in front of its end, so nothing is checked per
// step).  The 64-bit window {lo, hi} lives in v[120:121]: a pair of steps consumes at most 24 bits,
// so the window is shifted by 64-bit shifts twice and topped up ONCE per pair (selects, no
// execution-mask change; the word after it is prefetched a pair ahead).  Per step: bits += e & 15,
// cnt16 += e & 0x30 (16 x literals).  A step on a run / end-of-block / impossible entry adds
// nothing; a lane that met one leaves the group at the end of that pair (v_cmpx clears its bit of
// the execution mask, which is restored at the end) -- there is no branch but the loop's own.
// eA / eB return the entries of the lane's last pair, `last` the second entry of the pair before
// it (seg_last_after_group picks the lane's last literal entry from the three).  Requires the
// literal table at LDS offset 0.  The choice of instructions follows their measured issue cost on
// gfx950 (tools/ubench: ~2 cycles per wavefront for plain VOP2, ~4 for VOP3 and compares).
#ifndef FDH_SEG_R0   // the four fixed VGPRs of the groups (an aligned pair for the window, one for the shifted window)
#define FDH_SEG_R0 "120"
#define FDH_SEG_R1 "121"
#define FDH_SEG_R2 "122"
#define FDH_SEG_R3 "123"
#endif
#define FDH_SEG_WLO "v" FDH_SEG_R0
#define FDH_SEG_WHI "v" FDH_SEG_R1
#define FDH_SEG_WIN "v[" FDH_SEG_R0 ":" FDH_SEG_R1 "]"
#define FDH_SEG_SHF "v[" FDH_SEG_R2 ":" FDH_SEG_R3 "]"
#define FDH_SEG_SH0 "v" FDH_SEG_R2
#define FDH_SEG_CLOBBER "v" FDH_SEG_R0, "v" FDH_SEG_R1, "v" FDH_SEG_R2, "v" FDH_SEG_R3
// Tops up the window after a pair of steps: b > 31 -> {lo, hi} = {hi, next word}, one more word is
// read from the ring.  (gfx950 needs two instructions between a VALU write of VCC and a VALU read.)
#define FDH_SEG_TOPUP_WINDOW                                                                              \
    "  v_cmp_lt_u32 vcc, 31, %[b]\n"                                                                      \
    "  v_and_b32 %[b], 31, %[b]\n"                                                                        \
    "  v_mov_b32 %[last], %[eB]\n"                                                                        \
    "  v_cndmask_b32 " FDH_SEG_WLO ", " FDH_SEG_WLO ", " FDH_SEG_WHI ", vcc\n"                            \
    "  v_cndmask_b32 " FDH_SEG_WHI ", " FDH_SEG_WHI ", %[nw], vcc\n"                                      \
    "  v_addc_co_u32 %[ird], vcc, 0, %[ird], vcc\n"                                                       \
    "  v_lshlrev_b32 %[t], 8, %[ird]\n"                                                                   \
    "  v_and_or_b32 %[ra], %[t], %[mf00], %[rb]\n"
#define FDH_SEG_LOOP_END                                                                                  \
    FDH_SEG_TOPUP_WINDOW                                                                                  \
    "  ds_read_b32 %[nw], %[ra]\n"                                                                        \
    "  s_sub_u32 %[pairs], %[pairs], 1\n"                                                                 \
    "  s_cmp_lg_u32 %[pairs], 0\n"                                                                        \
    "  s_cbranch_scc1 Lpair_%=\n"                                                                         \
    "  s_mov_b64 exec, %[sv]\n" /* the lanes that left on the way are back: bring THEIR window in order */ \
    "  s_waitcnt lgkmcnt(0)\n"                                                                            \
    "  v_cmp_lt_u32 vcc, 31, %[b]\n"                                                                      \
    "  v_and_b32 %[b], 31, %[b]\n"                                                                        \
    "  s_nop 0\n"                                                                                         \
    "  v_cndmask_b32 %[lo], " FDH_SEG_WLO ", " FDH_SEG_WHI ", vcc\n"                                      \
    "  v_cndmask_b32 %[hi], " FDH_SEG_WHI ", %[nw], vcc\n"                                                \
    "  v_addc_co_u32 %[ird], vcc, 0, %[ird], vcc\n"

// `pairs` pairs of look-ups of the counting loop, hand-scheduled, for the lanes the caller has
// enabled (every one of them is >= 24 * pairs bits in front of its end, so nothing is checked per
// step).  The 64-bit window {lo, hi} lives in an aligned register pair: a pair of steps consumes at
// most 24 bits, so the window is shifted by 64-bit shifts twice and topped up ONCE per pair
// (selects, no execution-mask change; the word after it is prefetched a pair ahead).  Per step:
// bits += e & 15, cnt16 += e & 0x30 (16 x literals).  A lane that meets a run / end-of-block /
// impossible entry (no literals) leaves the group on the spot -- v_cmpx clears its bit of the
// execution mask, nothing of that step is applied, the mask is restored at the end -- so there is
// no branch but the loop's own.  eA / eB return the entries of the lane's last pair, `last` the
// second entry of the pair before it (seg_last_after_group picks the lane's last literal entry from
// the three).  Requires the literal table at LDS offset 0.  The choice of instructions follows
// their measured issue cost on gfx950 (tools/ubench: ~2 cycles per wavefront for plain VOP2, ~4 for
// VOP3 and compares); what bounds the loop is the dependent chain window -> index -> LDS -> bits
// (~160 cycles a step, tools/ubench/seg_group.hip), i.e. the number of wavefronts per SIMD.
#define FDH_SEG_COUNT_STEP(E)                                       \
    "  v_lshrrev_b64 " FDH_SEG_SHF ", %[b], " FDH_SEG_WIN "\n"      \
    "  v_and_b32 %[t], 0x3ffc, " FDH_SEG_SH0 "\n"                   \
    "  ds_read_b32 %[" E "], %[t]\n"                                \
    "  s_waitcnt lgkmcnt(0)\n"                                      \
    "  v_and_b32 %[tn], 0x30, %[" E "]\n"                           \
    "  v_cmpx_ne_u32 vcc, 0, %[tn]\n"                               \
    "  v_and_b32 %[t], 15, %[" E "]\n"                              \
    "  v_add_u32 %[b], %[b], %[t]\n"                                \
    "  v_add_u32 %[cnt], %[cnt], %[tn]\n"
__device__ __forceinline__ void seg_count_group(uint32_t pairs, uint32_t ring_base, SegReader& rd, uint32_t& cnt16,
                                                uint32_t& last, uint32_t& eA, uint32_t& eB) {
    static_assert(kSegInWords == 16, "ring word mask 0xf00 below");
    uint32_t ra = ring_base | ((rd.in_rd << 8) & 0xf00u);
    uint32_t nw, t, tn;
    uint64_t sv;
    uint32_t mf00 = 0xf00u;
    eA = eB = last;
    asm volatile(
        "  s_waitcnt lgkmcnt(0)\n"  // nothing of the compiler's may be in flight: the waits below assume it
        "  s_mov_b64 %[sv], exec\n"
        "  v_mov_b32 " FDH_SEG_WLO ", %[lo]\n"
        "  v_mov_b32 " FDH_SEG_WHI ", %[hi]\n"
        "  ds_read_b32 %[nw], %[ra]\n"
        "Lpair_%=:\n"
        FDH_SEG_COUNT_STEP("eA")
        FDH_SEG_COUNT_STEP("eB")
        FDH_SEG_LOOP_END
        : [pairs] "+s"(pairs), [cnt] "+v"(cnt16), [last] "+v"(last), [lo] "+v"(rd.lo), [hi] "+v"(rd.hi),
          [b] "+v"(rd.boff), [ra] "+v"(ra), [ird] "+v"(rd.in_rd), [eA] "+v"(eA), [eB] "+v"(eB), [nw] "=&v"(nw),
          [t] "=&v"(t), [tn] "=&v"(tn), [sv] "=&s"(sv)
        : [rb] "v"(ring_base), [mf00] "s"(mf00)
        : "vcc", "scc", "memory", FDH_SEG_CLOBBER);
}
#undef FDH_SEG_COUNT_STEP

// The same for the writing loop (pass 2): each step also appends its <= 3 literal bytes to the
// 4-byte accumulator; the accumulator is written to its ring slot every step (the slot at vposw is
// always free) and only counts -- vposw moves on -- once it is full.
struct SegWriter {
    uint32_t acc, sh, vposw;
};
#define FDH_SEG_WRITE_STEP(E)                                                                           \
    "  v_lshrrev_b64 " FDH_SEG_SHF ", %[b], " FDH_SEG_WIN "\n"                                          \
    "  v_and_b32 %[t], 0x3ffc, " FDH_SEG_SH0 "\n"                                                       \
    "  ds_read_b32 %[" E "], %[t]\n"                                                                    \
    "  s_waitcnt lgkmcnt(0)\n"                                                                          \
    "  v_and_b32 %[tn], 0x30, %[" E "]\n"                                                               \
    "  v_cmpx_ne_u32 vcc, 0, %[tn]\n"                                                                   \
    "  v_lshrrev_b32 %[v], 8, %[" E "]\n"        /* the literal bytes */                               \
    "  v_lshlrev_b32 %[t], %[sh], %[v]\n"                                                               \
    "  v_or_b32 %[acc], %[acc], %[t]\n"                                                                 \
    "  v_sub_u32 %[t], 32, %[sh]\n"                                                                     \
    "  v_lshrrev_b32 %[v], %[t], %[v]\n"         /* bytes that did not fit (used when full: sh > 0) */  \
    "  v_lshrrev_b32 %[tn], 1, %[tn]\n"                                                                 \
    "  v_add_u32 %[sh], %[sh], %[tn]\n"                                                                 \
    "  ds_write_b32 %[wa], %[acc]\n"                                                                    \
    "  v_cmp_lt_u32 vcc, 31, %[sh]\n"                                                                   \
    "  v_and_b32 %[sh], 31, %[sh]\n"                                                                    \
    "  v_and_b32 %[t], 15, %[" E "]\n"                                                                  \
    "  v_add_u32 %[b], %[b], %[t]\n"                                                                    \
    "  v_cndmask_b32 %[acc], %[acc], %[v], vcc\n"                                                       \
    "  v_addc_co_u32 %[vposw], vcc, 0, %[vposw], vcc\n"                                                 \
    "  v_lshlrev_b32 %[t], 8, %[vposw]\n"                                                               \
    "  v_and_or_b32 %[wa], %[t], %[mf00], %[ob]\n"
__device__ __forceinline__ void seg_write_group(uint32_t pairs, uint32_t ring_base, uint32_t out_base, SegReader& rd,
                                                SegWriter& wr, uint32_t& last, uint32_t& eA, uint32_t& eB) {
    static_assert(kSegInWords == 16 && kSegOutWords == 16, "ring word mask 0xf00 below");
    uint32_t ra = ring_base | ((rd.in_rd << 8) & 0xf00u);
    uint32_t wa = out_base | ((wr.vposw << 8) & 0xf00u);
    uint32_t nw, t, v, tn;
    uint64_t sv;
    uint32_t mf00 = 0xf00u;
    eA = eB = last;
    asm volatile(
        "  s_waitcnt lgkmcnt(0)\n"
        "  s_mov_b64 %[sv], exec\n"
        "  v_mov_b32 " FDH_SEG_WLO ", %[lo]\n"
        "  v_mov_b32 " FDH_SEG_WHI ", %[hi]\n"
        "  ds_read_b32 %[nw], %[ra]\n"
        "Lpair_%=:\n"
        FDH_SEG_WRITE_STEP("eA")
        FDH_SEG_WRITE_STEP("eB")
        FDH_SEG_LOOP_END
        : [pairs] "+s"(pairs), [last] "+v"(last), [lo] "+v"(rd.lo), [hi] "+v"(rd.hi), [b] "+v"(rd.boff), [ra] "+v"(ra),
          [ird] "+v"(rd.in_rd), [acc] "+v"(wr.acc), [sh] "+v"(wr.sh), [wa] "+v"(wa), [vposw] "+v"(wr.vposw),
          [eA] "+v"(eA), [eB] "+v"(eB), [nw] "=&v"(nw), [t] "=&v"(t), [v] "=&v"(v), [tn] "=&v"(tn), [sv] "=&s"(sv)
        : [rb] "v"(ring_base), [ob] "v"(out_base), [mf00] "s"(mf00)
        : "vcc", "scc", "memory", FDH_SEG_CLOBBER);
}
#undef FDH_SEG_WRITE_STEP
#undef FDH_SEG_LOOP_END
#undef FDH_SEG_TOPUP_WINDOW

// After a fast group: the last literal entry the lane has seen (for a run that follows).  A lane
// leaves the group at the step in which it meets an entry without literals: at the first step of a
// pair (eA is that entry; eB and `last` are the second entry of the pair before) or at the second
// (eB is that entry, eA the literal entry in front of it).
__device__ __forceinline__ uint32_t seg_last_after_group(uint32_t last, uint32_t eA, uint32_t eB) {
    return (eA & SK_KIND) ? last : ((eB & SK_KIND) ? eA : eB);
}

// Walks a chain through the synchronisation window: from s.pos until pos >= window (or a stop).
// GUESS: the chain is only a way to find a synchronisation point -- nothing is counted, and an
// impossible token (or a stray end-of-block) just means "not synchronised yet": slide on by one
// bit.  The landing check is what guarantees correctness.  Otherwise (the real chain) every byte
// is counted.  In the last kLitBits bits of the window a step is taken one literal at a time: the
// guessed and the real chain may group literals differently, but they then still cross the window
// on the same symbol boundary (the length of the first literal alone comes from the canonical
// table in global memory: this happens two or three times per scan).  In front of that zone the
// lanes run the same fast groups as the counting loop (4 or 2 pairs of look-ups).
template <bool GUESS>
__device__ __forceinline__ uint32_t seg_window_scan(const uint32_t* lit, uint32_t len4, SegReader& rd,
                                                    uint32_t limit, bool active, uint32_t window, SegScan& s) {
    bool running = active && s.pos < window;
    uint32_t iter = 0;
    const uint32_t ring_base = lds_offset(rd.ring) + 4 * rd.lane_off;
    const uint32_t zone = window - kLitBits;  // first bit of the symbol-by-symbol zone
    while (__any(running)) {
        rd.events(running, kSegEventNeed);
        for (int half = 0; half < 2; half++) {
            const uint32_t room = min(zone, limit);
            const bool f4 = running && s.pos + kSegGroupBits <= room && rd.level() >= kSegHalfNeed;
            const bool f2 = running && s.pos + kSegGroupBits / 2 <= room && rd.level() >= kSegHalfNeed;
            const uint32_t pairs = __any(f4) ? kSegPairs : kSegPairs / 2;
            const bool fast = pairs == kSegPairs ? f4 : f2;
            bool general = running && !fast && rd.level() >= 2;
            if (__any(fast)) {
                if (fast) {
                    uint32_t cnt16 = 0, eA, eB, last = s.last_e;
                    const uint32_t b0 = rd.boff, r0 = rd.in_rd;
                    seg_count_group(pairs, ring_base, rd, cnt16, last, eA, eB);
                    s.pos += 32 * (rd.in_rd - r0) + rd.boff - b0;
                    if (!GUESS) {
                        s.count8 += cnt16 >> 1;
                        s.last_e = seg_last_after_group(last, eA, eB);
                    }
                    general = ((eA | eB) & SK_KIND) != 0;
                }
                iter += 2 * pairs;
            }
            if (__any(general)) {
                iter++;
                const uint32_t win = rd.window();
                const uint32_t nw = rd.peek();
                SegTok t = seg_token(lit, len4, win, general, s.pos >= zone);
                if (GUESS) {
                    const bool slide = (t.bad || t.eob) && s.pos + 1 <= limit;
                    t.used = slide ? 1u : t.used;
                    t.n8 = slide ? 0u : t.n8;
                    t.run = slide ? 0u : t.run;
                    t.bad = slide ? false : t.bad;
                    t.eob = slide ? false : t.eob;
                }
                const bool fault = t.bad || s.pos + t.used > limit;
                const bool step = general && !fault && !t.eob;
                const bool halt = general && !step;
                s.stop = halt ? (fault ? 2u : 1u) : s.stop;
                s.eob_bits = halt ? t.used : s.eob_bits;
                if (!GUESS) {
                    s.count8 += step ? t.n8 + 8 * t.run : 0u;
                    s.last_e = (step && t.n8 != 0) ? t.e : s.last_e;
                }
                const uint32_t adv = step ? t.used : 0u;
                s.pos += adv;
                rd.advance(adv, nw);
                running = running && !halt;
            }
            running = running && s.pos < window;
        }
    }
    return iter;
}

// The long loop of pass 1: from s.pos to `stop_at`, counting every byte.
// Outer loop = one global-memory event, then two halves of {fast group for the lanes that are at
// least a group away from stop_at, one general step for the lanes that are not or that met a
// run / end-of-block / impossible entry}.  A lane runs while pos < lim; halting sets lim = 0.
__device__ __forceinline__ uint32_t seg_count_scan(const uint32_t* lit, SegReader& rd, uint32_t limit, bool active,
                                                   uint32_t stop_at, SegScan& s) {
    uint32_t lim = (active && s.stop == 0) ? stop_at : 0u;
    uint32_t iter = 0;
    const uint32_t ring_base = lds_offset(rd.ring) + 4 * rd.lane_off;
    if (s.pos < lim) rd.refill_now();
    while (__any(s.pos < lim)) {
        rd.events(s.pos < lim, kSegEventNeed);
        {
            // a fast group must stay inside the lane's range and inside the stream, and needs its
            // input in the ring (a lane that is short of input sits this round out: rare, dense data).
            // Nothing is drained in this pass, so one event serves a group of twice the steps (or,
            // for the lanes close to their end, one of the usual size).
            const uint32_t room = min(lim, limit);
            const bool f2 = s.pos + 2 * kSegGroupBits <= room && rd.level() >= kSegEventNeed;
            const bool f1 = s.pos + kSegGroupBits <= room && rd.level() >= kSegHalfNeed;
            const uint32_t pairs = __any(f2) ? 2 * kSegPairs : kSegPairs;
            const bool fast = pairs == kSegPairs ? f1 : f2;
            bool general = s.pos < lim && !fast && rd.level() >= 2;
            if (__any(fast)) {
                iter += (uint32_t)__popcll(__ballot(fast)) << 16;
                if (fast) {
                    uint32_t cnt16 = 0, eA, eB;
                    const uint32_t b0 = rd.boff, r0 = rd.in_rd;
                    seg_count_group(pairs, ring_base, rd, cnt16, s.last_e, eA, eB);
                    s.pos += 32 * (rd.in_rd - r0) + rd.boff - b0;
                    s.count8 += cnt16 >> 1;
                    s.last_e = seg_last_after_group(s.last_e, eA, eB);
                    general = ((eA | eB) & SK_KIND) != 0;
                }
                iter += 1u << 8;
            }
            // general step (selects only): a token of any kind.  Returns the lanes that took a run of
            // the maximal length and sit on another run token (a flat stretch is a chain of maximal
            // runs): for those the step is repeated at once instead of costing a whole half each.
            auto general_step = [&](bool take) __attribute__((always_inline)) {
                iter++;
                const uint32_t win = rd.window();
                const uint32_t nw = rd.peek();
                const SegTok t = seg_token(lit, 0u, win, take, false);
                const bool fault = t.bad || s.pos + t.used > limit;
                const bool step = take && !fault && !t.eob;
                const bool halt = take && !step;
                s.stop = halt ? (fault ? 2u : 1u) : s.stop;
                s.eob_bits = halt ? t.used : s.eob_bits;
                lim = halt ? 0u : lim;
                s.count8 += step ? t.n8 + 8 * t.run : 0u;
                s.last_e = (step && t.n8 != 0) ? t.e : s.last_e;
                const uint32_t adv = step ? t.used : 0u;
                s.pos += adv;
                rd.advance(adv, nw);
                return step && t.run == 258 && s.pos < lim && rd.level() >= 2;
            };
            if (__any(general)) {
                bool again = general_step(general);
                for (int rep = 0; rep < 16 && __any(again); rep++) {
                    again = again && (lit[rd.window() & (kLitSize - 1)] & SK_KIND) == SK_RUN;
                    if (__any(again)) again = general_step(again);
                }
            }
        }
    }
    return iter;
}

// What pass 2 needs from pass 1 (per lane unless noted).
struct SegPlan {
    uint32_t start;   // real chain start of the lane (segment-relative bit)
    uint32_t end2;    // chain end (>= segment length) or the end-of-block position; 0 = lane has nothing to do
    uint32_t obase;   // output offset of the lane's first byte
    uint32_t count;   // output bytes of the lane
    uint32_t last_e;  // entry of the last literal step in front of the lane (0: none)
    uint32_t total;   // uniform: output bytes of the stream
    uint32_t tb;      // uniform: stream byte position of the Adler-32 trailer
};

// Passes 1 + check + scan of one stream.  False: the stream was left PENDING (or is out of range).
__device__ __forceinline__ bool segments_plan(const SegArgs& a, const uint32_t* lit, uint32_t* in_ring, const uint32_t lane_off,
                                              const uint64_t sid, SegPlan& plan) {
    const int lane = threadIdx.x & (kWave - 1);
    if (sid >= a.n) return false;
    const uint32_t len4 = a.canon_len4[lane & 31];  // lengths of the literal codes, 4 bits each (seg_token)

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
        uint32_t itw = seg_window_scan<true>(lit, len4, rd, limit, in_range, window, tail);
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
            uint32_t ith = seg_window_scan<false>(lit, len4, rd, limit, need, window, head);
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
    // last literal step of every lane's chain -> the byte a leading run of the right neighbour repeats
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
    SegWriter wr;
    wr.vposw = pad >> 2;                        // virtual position in dwords: words already in the ring
    wr.acc = 0;
    wr.sh = 8 * (pad & 3);                      // 4-byte accumulator holding sh / 8 bytes (the rest is zero)
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
    auto account_piece = [&](const uint4 q, uint32_t vs, bool stored_already) __attribute__((always_inline)) {
        if (stored_already) {
        } else if (vs >= pad && vs + 16 <= vend) {
            *reinterpret_cast<uint4*>(line0 + vs) = q;
        } else {  // first / last line of this lane: only its own bytes (unrolled: no indexed access to q)
#pragma unroll
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
    auto ring_line = [&](uint32_t sl, uint32_t w) __attribute__((always_inline)) {
        uint4 q;
        q.x = oring[seg_slot(sl, w + 0)];
        q.y = oring[seg_slot(sl, w + 1)];
        q.z = oring[seg_slot(sl, w + 2)];
        q.w = oring[seg_slot(sl, w + 3)];
        return q;
    };
    auto store_piece = [&](uint32_t vs) __attribute__((always_inline)) { account_piece(ring_line(lane_off, vs >> 2), vs, false); };
    auto drain_all = [&]() __attribute__((always_inline)) {  // every complete 16-B line
        while (__any(4 * wr.vposw - vstored >= 16)) {
            if (4 * wr.vposw - vstored >= 16) {
                store_piece(vstored);
                vstored += 16;
            }
        }
    };
    // Lines leave in 32-B aligned pairs (two adjacent 16-B stores back to back) so that whole
    // 32-B sectors reach the L2 together; a lane whose first line is the upper half of a sector
    // sends that one alone.  < 32 B stay behind, + <= 28 B per half (a group of 3-byte steps and a
    // general step): fits the 64-B ring with the slot of the open accumulator.
    const bool odd_first = (reinterpret_cast<uintptr_t>(line0) & 16) != 0;
    auto drain = [&]() __attribute__((always_inline)) {
        for (;;) {
            const uint32_t avail = 4 * wr.vposw - vstored;
            const bool single = odd_first && vstored == 0 && avail >= 16;
            const bool pair = !single && avail >= 32;
            if (!__any(single || pair)) break;
            // An interior pair is stored by two lanes in ONE instruction (lane l and l ^ 1 each write
            // 16 of the 32 bytes: first the pairs of the even lanes, then those of the odd lanes), so
            // the memory pipeline sees 32-B requests instead of twice 16 B.  Every lane reads its own
            // two lines from the ring once (store + checksum); the neighbour's second line and
            // pointer come over DPP.
            const bool coop = pair && vstored >= pad && vstored + 32 <= vend;
            const uint4 l0 = ring_line(lane_off, vstored >> 2);
            const uint4 l1 = ring_line(lane_off, (vstored >> 2) + 4);
            uint8_t* const my_ptr = line0 + vstored;
            const uint4 p1 = make_uint4(swap1(l1.x), swap1(l1.y), swap1(l1.z), swap1(l1.w));
            const uint32_t p_lo = swap1((uint32_t)reinterpret_cast<uintptr_t>(my_ptr));
            const uint32_t p_hi = swap1((uint32_t)(reinterpret_cast<uintptr_t>(my_ptr) >> 32));
            const bool p_coop = swap1(coop ? 1u : 0u) != 0;
            uint8_t* const p_ptr = reinterpret_cast<uint8_t*>(((uintptr_t)p_hi << 32) | p_lo);
#pragma unroll
            for (int par = 0; par < 2; par++) {
                const bool owner = (lane & 1) == par;
                // (component-wise opaque selects: a select between two uint4 would become a scratch array)
                const uint4 q = make_uint4(vsel(owner, l0.x, p1.x), vsel(owner, l0.y, p1.y), vsel(owner, l0.z, p1.z),
                                           vsel(owner, l0.w, p1.w));
                if (owner ? coop : p_coop) *reinterpret_cast<uint4*>(owner ? my_ptr : p_ptr + 16) = q;
            }
            if (single || pair) {
                account_piece(l0, vstored, coop);
                vstored += 16;
            }
            if (pair) {
                account_piece(l1, vstored, coop);
                vstored += 16;
            }
        }
    };
    // A long dist-1 run (src/decompress.rs:793-801 fills it with one byte): bring the lane to a
    // 16-B line boundary through the ring, then store whole lines of the byte directly; their
    // Adler-32 contribution has a closed form.
    uint32_t bulk_lines = 0, bulk_c4 = 0;
    uint8_t* bulk_dst = nullptr;
    auto bulk_fill = [&](uint32_t c) __attribute__((always_inline)) {
        const uint32_t c4 = c * 0x01010101u;
        // complete the accumulator, then whole words up to the line boundary
        wr.acc |= c4 << wr.sh;
        oring[seg_slot(lane_off, wr.vposw)] = wr.acc;
        wr.vposw++;
        fill -= 4 - (wr.sh >> 3);
        wr.acc = 0;
        wr.sh = 0;
        while (wr.vposw & 3) {
            oring[seg_slot(lane_off, wr.vposw)] = c4;
            wr.vposw++;
            fill -= 4;
        }
        while (4 * wr.vposw != vstored) {  // the complete lines waiting in the ring
            store_piece(vstored);
            vstored += 16;
        }
        bulk_lines = min(fill, vend - vstored) >> 4;
        const uint32_t m = bulk_lines << 4;
        bulk_dst = line0 + vstored;
        bulk_c4 = c4;
        // m bytes of value c: a' = a + m c ; b' = b + m a + c m (m + 1) / 2
        ad_a %= kAdlerMod;
        ad_b %= kAdlerMod;
        blocks = 0;
        const uint64_t tri = ((uint64_t)m * (m + 1) / 2) % kAdlerMod;
        ad_b = (uint32_t)((ad_b + (uint64_t)(m % kAdlerMod) * ad_a + tri * c) % kAdlerMod);
        ad_a = (uint32_t)((ad_a + (uint64_t)m * c) % kAdlerMod);
        vstored += m;
        wr.vposw += m >> 2;
        fill -= m;
    };
    // ... and the lines themselves are stored by the whole wavefront, one lane's run after the
    // other, 64 lines (1 KiB, coalesced) per instruction: a stream of long runs is decoded by a
    // handful of lanes, which would otherwise store their KiBs line by line on their own.
    auto bulk_store = [&]() __attribute__((always_inline)) {
        uint64_t todo = __ballot(bulk_lines != 0);
        while (todo) {
            const int src = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t n_lines = __shfl(bulk_lines, src, kWave);
            const uint32_t c4 = __shfl(bulk_c4, src, kWave);
            const uint32_t d_lo = __shfl((uint32_t)reinterpret_cast<uintptr_t>(bulk_dst), src, kWave);
            const uint32_t d_hi = __shfl((uint32_t)(reinterpret_cast<uintptr_t>(bulk_dst) >> 32), src, kWave);
            uint8_t* const dst = reinterpret_cast<uint8_t*>(((uintptr_t)uni(d_hi) << 32) | uni(d_lo));
            const uint4 q = make_uint4(c4, c4, c4, c4);
            for (uint32_t i = (uint32_t)lane; i < uni(n_lines); i += kWave) *reinterpret_cast<uint4*>(dst + 16 * (size_t)i) = q;
        }
        bulk_lines = 0;
    };

    SEGTIME(4);
#ifdef FDH_DEBUG_TILES
    const long long t_start0 = clock64();
#endif
    if (live) rd.start(in, seg_bit0 + pos);
#ifdef FDH_DEBUG_TILES
    const uint32_t tq_start = (uint32_t)(clock64() - t_start0);
    const long long t_loop0 = clock64();
#endif
    uint32_t iter = 0;
    const uint32_t ring_base = lds_offset(L.in_ring) + 4 * lane_off;
    const uint32_t out_base = lds_offset(L.out_ring) + 4 * lane_off;
    // A lane runs while pos < end2 or a run is being filled.
    // Outer loop = input event + two x (drain + fast group + one general step).
#ifdef FDH_DEBUG_TILES
    uint32_t tq_ev = 0, tq_dr = 0, tq_grp = 0, tq_gen = 0, nq_gen = 0, nq_grp = 0;
#define TQ(var, t0) var += (uint32_t)(clock64() - (t0))
#define TQ0(name) const long long name = clock64()
#else
#define TQ(var, t0) do { } while (0)
#define TQ0(name) do { } while (0)
#endif
    while (__any(pos < end2 || fill != 0)) {
        TQ0(t_ev);
        rd.events(pos < end2 || fill != 0, kSegEventNeed);
        TQ(tq_ev, t_ev);
        for (int half = 0; half < 2; half++) {
            TQ0(t_dr);
            drain();
            if (__any(fill >= kSegBulkFill)) {
                if (fill >= kSegBulkFill) bulk_fill(seg_lastlit(last_e));
                bulk_store();
            }
            TQ(tq_dr, t_dr);
            TQ0(t_grp);
            // fast group: lanes at least a group away from their end that are not filling a run
            const bool fast = fill == 0 && pos + kSegGroupBits <= end2 && rd.level() >= kSegHalfNeed;
            bool general = !fast && (fill != 0 || (pos < end2 && rd.level() >= 2));
            if (__any(fast)) {
                if (fast) {
                    uint32_t eA, eB;
                    const uint32_t b0 = rd.boff, r0 = rd.in_rd;
                    seg_write_group(kSegPairs, ring_base, out_base, rd, wr, last_e, eA, eB);
                    pos += 32 * (rd.in_rd - r0) + rd.boff - b0;
                    last_e = seg_last_after_group(last_e, eA, eB);
                    general = ((eA | eB) & SK_KIND) != 0;
                }
                iter += kSegSteps;
#ifdef FDH_DEBUG_TILES
                nq_grp++;
#endif
            }
            TQ(tq_grp, t_grp);
            TQ0(t_gen);
            // general step (selects only): a token of any kind, or 4 bytes of a run in progress.
            // Returns the lanes that took / merged a run of the maximal length (what a longer run is cut
            // into) and may sit on the next piece: for those the step is repeated at once (nothing is
            // emitted in a repeat: a run right behind a run only adds to `fill`).
            auto general_step = [&](bool take) __attribute__((always_inline)) {
#ifdef FDH_DEBUG_TILES
                nq_gen++;
#endif
                iter++;
                const uint32_t win = rd.window();
                const uint32_t nw = rd.peek();
                const bool filling = fill != 0;
                const bool can_dec = take && pos < end2;
                const SegTok t = seg_token(lit, 0u, win, can_dec, false);
                const bool is_run = t.run != 0;
                // a run right behind a run goes on with the same byte: the lengths add up (a stream of
                // long runs is then one bulk fill for many tokens)
                const bool merge = filling && can_dec && is_run && !t.bad;
                const bool dec = can_dec && (!filling || merge);  // this lane decodes a token now
                const uint32_t n8_lit = (dec && !merge) ? t.n8 : 0u;
                const bool bad_now = dec && (t.bad || t.eob || (is_run && last_e == 0));
                bad2 = bad2 || bad_now;
                last_e = n8_lit ? t.e : last_e;
                const bool emit = take && filling && !merge;
                const uint32_t nf = emit ? min(fill, 4u) : 0u;
                uint32_t vf = seg_lastlit(last_e) * 0x01010101u;
                vf = nf < 4 ? (vf & ((1u << (8 * nf)) - 1)) : vf;
                const uint32_t n8 = emit ? 8 * nf : n8_lit;
                const uint32_t v = emit ? vf : (n8_lit ? t.e >> 8 : 0u);
                fill = merge ? fill + t.run : (filling ? fill - nf : ((dec && is_run && !bad_now) ? t.run : 0u));
                const uint32_t used = dec ? t.used : 0u;
                pos += used;
                rd.advance(used, nw);
                end2 = bad_now ? 0u : end2;
                const uint64_t tt = (uint64_t)v << wr.sh;
                wr.acc |= (uint32_t)tt;
                // the ring slot at vposw is always free: the (possibly partial) accumulator is
                // written there every time and only counts once it is full
                oring[seg_slot(lane_off, wr.vposw)] = wr.acc;
                const uint32_t tot = wr.sh + n8;
                const bool full = tot >= 32;
                wr.acc = full ? (uint32_t)(tt >> 32) : wr.acc;
                wr.vposw += full ? 1u : 0u;
                wr.sh = tot & 31;
                return (merge || (dec && is_run && !bad_now)) && t.run == 258 && pos < end2 && rd.level() >= 2;
            };
            if (__any(general)) {
                bool again = general_step(general);
                for (int rep = 0; rep < 16 && __any(again); rep++) {
                    again = again && (lit[rd.window() & (kLitSize - 1)] & SK_KIND) == SK_RUN;
                    if (__any(again)) again = general_step(again);
                }
            }
            TQ(tq_gen, t_gen);
        }
    }
    (void)iter;
#ifdef FDH_DEBUG_TILES
    if (sid < 16 && lane == 0) {
        g_segdbg2[sid * 8 + 0] = tq_ev;
        g_segdbg2[sid * 8 + 1] = tq_dr;
        g_segdbg2[sid * 8 + 2] = tq_grp;
        g_segdbg2[sid * 8 + 3] = tq_gen;
        g_segdbg2[sid * 8 + 4] = nq_grp;
        g_segdbg2[sid * 8 + 5] = nq_gen;
        g_segdbg2[sid * 8 + 6] = tq_start;
        g_segdbg2[sid * 8 + 7] = (uint32_t)(clock64() - t_loop0);
    }
#endif
    SEGTIME(5);
    SEGDBG(4, iter);
    SEGDBG(7, total);
    // ---- tail of every lane: the last (partial) line ----
    drain_all();
    if (live) {
        // the loose bytes go into the ring as a final word; zero the rest of that 16-B line
        oring[seg_slot(lane_off, wr.vposw)] = wr.acc;
        for (uint32_t w = wr.vposw + 1; (w & 3) != 0; w++) oring[seg_slot(lane_off, w)] = 0;
        if (vstored < vend) {
            store_piece(vstored);
            // the last line was summed as 16 bytes; it holds only 16 - z of ours followed by z zeros
            const uint32_t z = vstored + 16 - vend;
            ad_a %= kAdlerMod;
            ad_b %= kAdlerMod;
            ad_b = (ad_b + kAdlerMod - (uint32_t)(((uint64_t)z * ad_a) % kAdlerMod)) % kAdlerMod;
        }
    }
    const bool wrong_count = live && (4 * wr.vposw + (wr.sh >> 3) != vend);
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
            // Every token of the stream was decoded, its end-of-block code found and the four trailer bytes are
            // there (the plan checked): the reference gets this far too, and Ok / WrongChecksum is the comparison
            // (src/decompress.rs:306-326; see needs_serial_recheck, inflate.hip).  Rounds 1-3 handed such a stream
            // on for two more full decodes.
            a.status[sid] = ST_WRONG_CHECKSUM;
            a.out_len[sid] = total;
            if (a.adler) a.adler[sid] = adler;
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
