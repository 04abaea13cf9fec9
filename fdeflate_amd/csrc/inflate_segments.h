// inflate_segments.h -- segment-parallel decode of one ultra-fast-format stream per wavefront.
//
// A stream that starts with the ultra-fast encoder's fixed prefix (reference
// src/compress/ultrafast.rs:82-88) is one final dynamic block with a known table.  Its block
// data is cut into 64 equal bit ranges ("segments"), one per lane, and every lane runs the
// reference's inner loop (src/decompress.rs:645-830: table look-up, 1-2 literals per step, dist-1
// run) sequentially over its own segment:
//
//   pass 1  every lane decodes from the FIRST BIT of its segment -- a guess, the real symbol
//           boundary lies up to 17 bits further -- skipping the first kSegWindow bits and then
//           counting output bytes to the end of the segment.  Huffman codes self-synchronise, so
//           by the end of the window the guessed chain has (almost always) joined the real one;
//           x0 = where the chain left the window, end = where it left the segment.
//   check   lane i takes its real start from lane i-1's end, decodes the window from there
//           (counting) and must land exactly on x0.  If it does, by induction from lane 0 every
//           chain from x0 on was the real one.  A lane that lands elsewhere re-counts its segment
//           from the landing point and the check repeats for its successors (rare).
//   scan    a wavefront prefix sum of the byte counts gives every lane its output offset;
//           the byte a leading run repeats comes from the nearest lane to the left that emitted
//           a literal.
//   pass 2  every lane decodes its real chain again and streams the bytes through an 8-byte
//           accumulator and a small LDS ring to 16-B aligned global stores, folding them into a
//           per-lane Adler-32 partial; the partials are combined with the block-combine identity.
//
// Input is read per lane through a small LDS ring that is topped up by 16-B global loads at
// wavefront-uniform "events" (two events ahead), so the hot loop never waits on memory.
// Anything unusual -- not canonical, too short, a bad / truncated token, a full slot, a
// checksum mismatch -- leaves the stream PENDING for the exact wave-per-stream kernels.
#pragma once
#include "inflate_tables.h"

namespace fdh {

#ifdef FDH_DEBUG_TILES
__device__ uint32_t g_segdbg[64 * 16];
#define SEGDBG(slot, val) do { if (sid < 64 && lane == 0) g_segdbg[sid * 16 + (slot)] = (val); } while (0)
#define SEGDBG_ADD(slot, val) do { if (sid < 64 && lane == 0) g_segdbg[sid * 16 + (slot)] += (val); } while (0)
#else
#define SEGDBG(slot, val) do { } while (0)
#define SEGDBG_ADD(slot, val) do { } while (0)
#endif

constexpr int kSegWaves = 8;        // wavefronts (= streams) per workgroup: 80 KiB LDS -> 16 wavefronts/CU
constexpr int kSegInWords = 16;     // per-lane input ring: 16 dwords (64 B)
constexpr int kSegOutWords = 8;     // per-lane output ring: 8 qwords (64 B)
constexpr int kSegWindow = 256;     // bits of a segment used for self-synchronisation
constexpr uint32_t kSegMinBits = 4 * kSegWindow;  // shorter segments: not worth it -> PENDING
constexpr uint32_t kNoByte = 0x100;  // "no literal seen yet"

// Rings are [word][lane]: any per-lane word index is bank-conflict free.
struct SegWaveLds {
    uint32_t in_ring[kSegInWords][kWave];
    uint64_t out_ring[kSegOutWords][kWave];
};
struct SegLds {
    uint32_t lit[kLitSize];
    SegWaveLds w[kSegWaves];
#ifdef FDH_SEG_PAD_LDS
    uint32_t pad[FDH_SEG_PAD_LDS / 4];
#endif
};

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
    uint32_t* list;  // nullable: [0] = count, [4..] = ids of the streams left PENDING (compacted)
};

// Leaves stream `sid` to the wave-per-stream kernels (lane 0 only).
__device__ __forceinline__ void seg_leave_pending(const SegArgs& a, uint64_t sid) {
    a.status[sid] = a.pending;
    if (a.list) {
        uint32_t k = atomicAdd(&a.list[0], 1u);
        a.list[4 + k] = (uint32_t)sid;
    }
}

// 16 bytes from a 16-B aligned address, zero where outside [lo, hi).
__device__ __attribute__((noinline)) uint4 seg_load16_edge(const uint8_t* p, const uint8_t* lo, const uint8_t* hi) {
    uint64_t a = 0, b = 0;
    for (int j = 0; j < 8; j++) {
        if (p + j >= lo && p + j < hi) a |= (uint64_t)p[j] << (8 * j);
        if (p + 8 + j >= lo && p + 8 + j < hi) b |= (uint64_t)p[8 + j] << (8 * j);
    }
    return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
}
__device__ __forceinline__ uint4 seg_load16(const uint8_t* p, const uint8_t* lo, const uint8_t* hi) {
    if (p >= lo && p + 16 <= hi) return *reinterpret_cast<const uint4*>(p);
    return seg_load16_edge(p, lo, hi);
}

// Per-lane sequential bit reader over the lane's input ring.
struct SegReader {
    uint32_t* ring;        // &in_ring[0][lane]; word w at ring[w * kWave]
    const uint8_t* gp;     // next 16-B chunk to request from global memory
    const uint8_t* buf_lo;
    const uint8_t* buf_hi;
    uint32_t in_wr, in_rd; // dwords written to / read from the ring
    uint32_t lo, hi, nextw, boff;
    uint4 pend_a, pend_b;  // chunks requested two / one events ago
    bool has_a, has_b;

    __device__ __forceinline__ void put(const uint4& v) {
        ring[((in_wr + 0) & (kSegInWords - 1)) * kWave] = v.x;
        ring[((in_wr + 1) & (kSegInWords - 1)) * kWave] = v.y;
        ring[((in_wr + 2) & (kSegInWords - 1)) * kWave] = v.z;
        ring[((in_wr + 3) & (kSegInWords - 1)) * kWave] = v.w;
        in_wr += 4;
    }
    __device__ __forceinline__ uint32_t get() {
        uint32_t w = ring[(in_rd & (kSegInWords - 1)) * kWave];
        in_rd++;
        return w;
    }
    // Positions the reader at stream bit `bit` (relative to the stream's first byte `in`) and
    // primes the whole ring (64 B) synchronously.
    __device__ __forceinline__ void start(const uint8_t* in, uint32_t bit) {
        const uint8_t* addr = in + (bit >> 3);
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(addr) & 15);
        gp = addr - mis;
        in_wr = in_rd = 0;
        has_a = has_b = false;
        for (int k = 0; k < kSegInWords / 4; k++) {
            put(seg_load16(gp, buf_lo, buf_hi));
            gp += 16;
        }
        in_rd = mis >> 2;
        lo = get();
        hi = get();
        nextw = get();
        boff = 8 * (mis & 3) + (bit & 7);
    }
    __device__ __forceinline__ uint32_t window() const { return __builtin_amdgcn_alignbit(hi, lo, boff); }
    __device__ __forceinline__ void consume(uint32_t used) {  // used <= 32
        boff += used;
        if (boff >= 32) {
            boff -= 32;
            lo = hi;
            hi = nextw;
            nextw = get();
        }
    }
    // Branch-free consume for the hot loops: the ring word is read unconditionally.
    __device__ __forceinline__ void consume_sel(uint32_t used) {  // used <= 32
        const uint32_t nw = ring[(in_rd & (kSegInWords - 1)) * kWave];
        boff += used;
        const bool wrap = boff >= 32;
        lo = wrap ? hi : lo;
        hi = wrap ? nextw : hi;
        nextw = wrap ? nw : nextw;
        boff = wrap ? boff - 32 : boff;
        in_rd += wrap ? 1u : 0u;
    }
    __device__ __forceinline__ bool starved() const { return in_rd > in_wr; }
    // Wavefront-uniform event: commit what was requested two events ago, request the next chunk.
    __device__ __forceinline__ void event(bool want_more) {
        if (has_a) put(pend_a);
        pend_a = pend_b;
        has_a = has_b;
        has_b = false;
        if (want_more && (uint32_t)kSegInWords - (in_wr - in_rd) >= (has_a ? 8u : 4u)) {
            pend_b = seg_load16(gp, buf_lo, buf_hi);
            gp += 16;
            has_b = true;
        }
    }
};

// One table look-up, decoded.  n: literal bytes (0..2) in v; run: length of a dist-1 run started
// by this token; used: stream bits; kind flags.
struct SegToken {
    uint32_t used, used1, n, v, run, lastlit;
    bool is_lit, is_run, is_eob, bad;
};
__device__ __forceinline__ SegToken seg_token(const uint32_t* lit, uint32_t win) {
    SegToken t;
    const uint32_t e = lit[win & (kLitSize - 1)];
    const uint32_t nb = e & 15, kind = (e >> 4) & 15;
    t.is_lit = kind <= K_LIT2;
    t.is_run = kind == K_LEN;
    t.is_eob = kind == K_EOB;
    const uint32_t ex = (e >> 8) & 31;
    const uint32_t length = (e >> 16) + ((win >> nb) & ((1u << ex) - 1));
    const uint32_t dbit = (win >> (nb + ex)) & 1;  // the prefix declares one distance code: '0' = 1
    t.used = nb + (t.is_run ? ex + 1 : 0);
    t.n = t.is_lit ? kind + 1 : 0;
    t.v = t.is_lit ? ((e >> 8) & (kind == K_LIT2 ? 0xFFFFu : 0xFFu)) : 0u;  // bytes only for literals
    t.used1 = e >> 24;  // bits of the first literal alone (literal entries)
    t.run = t.is_run ? length : 0;
    t.lastlit = kind == K_LIT2 ? (e >> 16) & 0xFF : (e >> 8) & 0xFF;
    t.bad = !(t.is_lit || t.is_run || t.is_eob) || (t.is_run && dbit != 0);
    return t;
}

// Counting scan of one lane's chain.  Starts at segment-relative bit `pos`, stops when pos >= stop_at
// (or at end-of-block / a bad token).  Bytes of tokens that start at or after `count_from` are added to
// `count`.  `cross` = first position >= kSegWindow the chain stepped on (recorded when
// RECORD_CROSS).  lastlit = last literal byte seen (kNoByte if none).
struct SegScan {
    uint32_t pos, count, cross, lastlit;
    uint32_t stop;  // 0 running/finished normally, 1 end-of-block (pos = its start, eob_bits set), 2 bad
    uint32_t eob_bits;
};

template <bool EVENTS, bool RECORD_CROSS>
__device__ __forceinline__ uint32_t seg_scan(const uint32_t* lit, SegReader& rd, const uint8_t* in, uint32_t seg_bit0,
                                         uint32_t in_bits, bool active, uint32_t stop_at, uint32_t count_from,
                                         SegScan& s) {
    if (active) rd.start(in, seg_bit0 + s.pos);
    uint32_t iter = 0;
    bool running = active && s.pos < stop_at;
    // Outer loop = one global-memory event, inner loop = 8 table look-ups that touch neither the
    // in-flight load registers nor global memory (so the compiler keeps waits and copies out of it).
    while (__any(running)) {
        if (EVENTS) rd.event(running);
#pragma unroll 1
        for (int k = 0; k < 8; k++) {
            iter++;
            // straight-line step: every lane looks up; only `running` lanes commit (selects only)
            SegToken t = seg_token(lit, rd.window());
            // Close to the window's end a literal pair is taken one literal at a time: the guessed
            // and the real chain may pair literals differently, but they then still cross the
            // window on the same symbol boundary.
            const bool single = t.is_lit && t.n == 2 && s.pos + 24 >= (uint32_t)kSegWindow && s.pos < (uint32_t)kSegWindow;
            t.used = single ? t.used1 : t.used;
            t.n = single ? 1u : t.n;
            t.lastlit = single ? (t.v & 0xFF) : t.lastlit;
            bool fault = t.bad || (seg_bit0 + s.pos + t.used > in_bits) || rd.starved();
            if (RECORD_CROSS) {
                // Inside the window the guessed chain is only a way to find a synchronisation
                // point: an impossible token (or a stray end-of-block) there just means "not
                // synchronised yet", so slide on by one bit.  The landing check is what
                // guarantees correctness.
                const bool slide = s.pos < (uint32_t)kSegWindow && (t.bad || t.is_eob) && !rd.starved() &&
                                   seg_bit0 + s.pos + 1 <= in_bits;
                t.used = slide ? 1u : t.used;
                t.n = slide ? 0u : t.n;
                t.run = slide ? 0u : t.run;
                t.is_lit = slide ? false : t.is_lit;
                t.is_eob = slide ? false : t.is_eob;
                fault = slide ? false : fault;
            }
            const bool step = running && !fault && !t.is_eob;
            const bool halt = running && !step;
            s.stop = halt ? (fault ? 2u : 1u) : s.stop;
            s.eob_bits = halt ? t.used : s.eob_bits;
            const bool counted = step && s.pos >= count_from;
            s.count += counted ? t.n + t.run : 0u;
            s.lastlit = (counted && t.is_lit) ? t.lastlit : s.lastlit;
            const uint32_t used = step ? t.used : 0u;
            s.pos += used;
            rd.consume_sel(used);
            if (RECORD_CROSS) s.cross = (s.cross == 0 && s.pos >= (uint32_t)kSegWindow) ? s.pos : s.cross;
            running = step && s.pos < stop_at;
        }
    }
    return iter;
}

__device__ __forceinline__ void segments_decode(const SegArgs& a, SegLds& L) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = threadIdx.x / kWave;
    const uint64_t sid = (uint64_t)blockIdx.x * kSegWaves + wid;
    if (sid >= a.n) return;
    const uint32_t* lit = L.lit;
    SegWaveLds& W = L.w[wid];

    // ---- stream set-up (uniform) ----
    const uint64_t i0 = a.in_off[sid], i1 = a.in_off[sid + 1];
    const uint64_t o0 = a.out_off[sid], o1 = a.out_off[sid + 1];
    const uint8_t* in = a.in + i0;
    uint8_t* op = a.out + o0;
    const uint8_t* buf_hi = a.in + a.in_off[a.n];
    const uint64_t ilen = i1 - i0, ocap = o1 - o0;
    bool ours = ilen < (1ull << 28) && ocap < (1ull << 31) && (reinterpret_cast<uintptr_t>(op) & 15) == 0 &&
                ilen * 8 >= a.canon_bits + 64ull * kSegMinBits;
    const uint32_t in_bits = (uint32_t)(ilen * 8);
    const uint32_t cap = (uint32_t)ocap;
    // canonical prefix: lane k compares stream dword k (unaligned loads are fine on gfx950)
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
        return;
    }
    // ---- segments: 64 equal bit ranges of the block data (the trailer bits ride along) ----
    const uint32_t data_bits = in_bits - a.canon_bits;
    const uint32_t seg = (data_bits + kWave - 1) / kWave;
    const uint32_t seg_bit0 = a.canon_bits + (uint32_t)lane * seg;  // first bit of this lane's segment

    SegReader rd;
    rd.ring = &W.in_ring[0][lane];
    rd.buf_lo = a.in;
    rd.buf_hi = buf_hi;
    rd.gp = in;
    rd.in_wr = rd.in_rd = 0;
    rd.lo = rd.hi = rd.nextw = rd.boff = 0;
    rd.pend_a = rd.pend_b = make_uint4(0, 0, 0, 0);
    rd.has_a = rd.has_b = false;

    // ---- pass 1: guessed chain from bit 0 of the segment; count from where it leaves the window ----
    SegScan tail;  // the chain from the window's end to the segment's end
    tail.pos = 0;
    tail.count = 0;
    tail.cross = 0;
    tail.lastlit = kNoByte;
    tail.stop = 0;
    tail.eob_bits = 0;
    const bool in_range = seg_bit0 < in_bits;
    {
        uint32_t it1 = seg_scan<true, true>(lit, rd, in, seg_bit0, in_bits, in_range, seg, (uint32_t)kSegWindow, tail);
        (void)it1;
        SEGDBG(0, it1);
        SEGDBG(1, 0);
        SEGDBG(2, 0);
        SEGDBG(3, 0);
        SEGDBG(5, seg);
    }
    // a chain that stopped inside the window never crossed it
    uint32_t x0 = tail.cross;  // 0 = did not cross

    // ---- check: real start from the left neighbour, decode the window, must land on x0 ----
    SegScan head;
    uint32_t start = 0;           // real chain start of this lane (segment-relative)
    uint32_t cur_start = ~0u;     // start the current `head` was computed for
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
            head.count = 0;
            head.cross = 0;
            head.lastlit = kNoByte;
            head.stop = 0;
            head.eob_bits = 0;
        }
        {
            uint32_t ith = seg_scan<false, false>(lit, rd, in, seg_bit0, in_bits, need, (uint32_t)kSegWindow, start, head);
            (void)ith;
            SEGDBG_ADD(1, ith);
            SEGDBG_ADD(3, 1);
        }
        // landed on the guessed chain?  then the counted tail is the real tail
        const bool redo = need && (head.stop == 0 ? (head.pos != x0 || x0 == 0) : false);
        const bool stopped_in_head = need && head.stop != 0;
        if (stopped_in_head) {  // end-of-block / bad token inside the window: no tail
            tail = head;
            tail.count = 0;
            x0 = head.pos;
        }
        if (__any(redo)) {  // rare: re-count this segment from the landing point
            if (redo) {
                tail.pos = head.pos;
                tail.count = 0;
                tail.cross = 0;
                tail.lastlit = kNoByte;
                tail.stop = 0;
                tail.eob_bits = 0;
                x0 = head.pos;
            }
            uint32_t itr = seg_scan<true, false>(lit, rd, in, seg_bit0, in_bits, redo, seg, head.pos, tail);
            (void)itr;
            SEGDBG_ADD(2, itr);
            SEGDBG_ADD(6, (uint32_t)__popcll(__ballot(redo)));
        }
        if (need) cur_start = start;
    }
    // ---- who is live: lanes up to the first stop on a verified chain ----
    const bool verified = in_range && cur_start == start && (lane == 0 || true);
    const uint64_t stop_mask = __ballot(verified && tail.stop != 0);
    const uint64_t unver_mask = __ballot(!verified);
    const int stop_lane = stop_mask ? __ffsll((unsigned long long)stop_mask) - 1 : kWave;
    const int first_unver = unver_mask ? __ffsll((unsigned long long)unver_mask) - 1 : kWave;
    const bool live = lane <= stop_lane;
    const uint32_t stop_kind = __shfl(tail.stop, stop_lane & (kWave - 1), kWave);
    // the stream must end with an end-of-block on a verified chain
    bool ok = !giveup && stop_lane < kWave && first_unver > stop_lane && stop_kind == 1;
    // bytes per lane: head (window) + tail
    const uint32_t head_count = (cur_start == start) ? head.count : 0;
    const uint32_t count = live ? head_count + tail.count : 0;
    // a run at the very start of the stream has nothing to repeat (DistanceTooFarBack upstream)
    uint32_t incl = count;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        uint32_t y = __shfl_up(incl, o, kWave);
        if (lane >= o) incl += y;
    }
    const uint32_t obase = incl - count;
    const uint32_t total = __shfl(incl, kWave - 1, kWave);
    ok = ok && total <= cap;
    // last literal of every lane's chain -> the byte a leading run of the right neighbour repeats
    uint32_t own_last = tail.lastlit != kNoByte ? tail.lastlit : ((cur_start == start) ? head.lastlit : kNoByte);
    if (!live) own_last = kNoByte;
    uint32_t carry = own_last;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        uint32_t y = __shfl_up(carry, o, kWave);
        if (lane >= o && carry == kNoByte) carry = y;
    }
    uint32_t incoming = __shfl_up(carry, 1, kWave);
    if (lane == 0) incoming = kNoByte;
    // trailer position: right after the end-of-block symbol
    const uint32_t eob_end = __shfl(seg_bit0 + tail.pos + tail.eob_bits, stop_lane & (kWave - 1), kWave);
    const uint32_t tb = (eob_end + 7) >> 3;
    ok = ok && (uint64_t)tb * 8 + 32 <= in_bits;
    SEGDBG(8, (uint32_t)stop_lane | ((uint32_t)first_unver << 8) | (stop_kind << 16) | ((giveup ? 1u : 0u) << 24));
    SEGDBG(9, total);
    if (!ok) {
        if (lane == 0) seg_leave_pending(a, sid);
        return;
    }

    // ---- pass 2: decode the real chain again, this time writing ----
    const uint32_t my_end = tail.pos;  // chain end (>= seg) or the end-of-block position
    uint32_t pos = start;
    uint64_t acc = 0;
    const uint32_t pad = obase & 15;          // bytes in front of this lane's first byte in its 16-B line
    uint8_t* const line0 = op + (obase - pad);  // 16-B aligned
    uint32_t vpos = pad & 8;                  // virtual position: multiples of 8 already in the ring
    uint32_t acc_n = pad & 7;
    uint32_t vstored = 0;                     // virtual bytes stored to global (multiple of 16)
    uint32_t ad_a = 0, ad_b = 0, blocks = 0;  // per-lane Adler partial over its own bytes
    uint32_t fill = 0, last = incoming;
    bool bad2 = false;
    uint64_t* const my_out = &W.out_ring[0][lane];
    if (pad & 8) my_out[0] = 0;               // the skipped qword of the first line
    const uint32_t vend = pad + count;        // virtual end

    auto store_piece = [&](uint32_t vs) __attribute__((always_inline)) {  // 16 virtual bytes at vs
        const uint32_t w = vs >> 3;
        const uint64_t x0q = my_out[(w & (kSegOutWords - 1)) * kWave];
        const uint64_t x1q = my_out[((w + 1) & (kSegOutWords - 1)) * kWave];
        const uint4 q = make_uint4((uint32_t)x0q, (uint32_t)(x0q >> 32), (uint32_t)x1q, (uint32_t)(x1q >> 32));
        if (vs >= pad && vs + 16 <= vend) {
            *reinterpret_cast<uint4*>(line0 + vs) = q;
        } else {  // first / last line of this lane: only its own bytes
            for (uint32_t k = 0; k < 16; k++) {
                if (vs + k >= pad && vs + k < vend)
                    line0[vs + k] = (uint8_t)((k < 8 ? x0q : x1q) >> (8 * (k & 7)));
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
    auto drain = [&]() __attribute__((always_inline)) {
        while (__any(vpos - vstored >= 16)) {
            if (vpos - vstored >= 16) {
                store_piece(vstored);
                vstored += 16;
            }
        }
    };

    if (live) rd.start(in, seg_bit0 + pos);
    bool running = live && (pos < my_end || fill);
    uint32_t iter = 0;
    // Outer loop = drain + (every other time) input event; inner loop = 4 straight-line steps.
    while (__any(running)) {
        drain();
        if ((iter & 4) == 0) rd.event(running);
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
            iter++;
            // straight-line step (selects only): a table look-up or 8 bytes of a run
            const SegToken t = seg_token(lit, rd.window());
            const bool filling = fill != 0;
            const bool dec = running && !filling;  // this lane decodes a token now
            bad2 = bad2 || (dec && (t.bad || t.is_eob || rd.starved() || (t.is_run && last == kNoByte)));
            const uint32_t nfill = min(fill, 8u);
            uint64_t vfill = (uint64_t)(last & 0xFF) * 0x0101010101010101ull;
            vfill = nfill < 8 ? (vfill & ((1ull << (8 * nfill)) - 1)) : vfill;
            const uint32_t n = running ? (filling ? nfill : t.n) : 0u;
            const uint64_t v = running ? (filling ? vfill : (uint64_t)t.v) : 0ull;
            fill = running ? (filling ? fill - nfill : t.run) : fill;
            last = (dec && t.is_lit) ? t.lastlit : last;
            const uint32_t used = dec ? t.used : 0u;
            pos += used;
            rd.consume_sel(used);
            // append n bytes; the ring slot at vpos is always free, so the (possibly partial)
            // accumulator is written there unconditionally and only counts once it is full
            const uint32_t tot = acc_n + n;
            acc |= v << (8 * acc_n);
            const bool full = tot >= 8;
            my_out[((vpos >> 3) & (kSegOutWords - 1)) * kWave] = acc;
            vpos += full ? 8u : 0u;
            const uint64_t spill = acc_n ? (v >> (8 * (8 - acc_n))) : 0ull;
            acc = full ? spill : acc;
            acc_n = full ? tot - 8 : tot;
            running = running && !bad2 && (pos < my_end || fill != 0);
        }
    }
    SEGDBG(4, iter);
    SEGDBG(7, total);
    // ---- tail of every lane: the last (partial) line ----
    drain();
    if (live) {
        // put the loose bytes into the ring as a final qword, then store what is left line by line
        if (acc_n) {
            my_out[((vpos >> 3) & (kSegOutWords - 1)) * kWave] = acc;
            if (((vpos >> 3) & 1) == 0) my_out[(((vpos >> 3) + 1) & (kSegOutWords - 1)) * kWave] = 0;
        } else if ((vpos - vstored) == 8) {
            my_out[(((vpos >> 3)) & (kSegOutWords - 1)) * kWave] = 0;
        }
        if (vstored < vend) {
            store_piece(vstored);
            // the last line was summed as 16 bytes; it holds only 16 - z of ours followed by z zeros
            const uint32_t z = vstored + 16 - vend;
            ad_a %= kAdlerMod;
            ad_b %= kAdlerMod;
            ad_b = (ad_b + kAdlerMod - (uint32_t)(((uint64_t)z * ad_a) % kAdlerMod)) % kAdlerMod;
        }
    }
    const bool wrong_count = live && (vpos + acc_n != vend);
    SEGDBG(10, (uint32_t)__ballot(bad2));
    SEGDBG(11, (uint32_t)(__ballot(bad2) >> 32));
    SEGDBG(12, (uint32_t)__ballot(wrong_count));
    SEGDBG(13, (uint32_t)(__ballot(wrong_count) >> 32));
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

}  // namespace fdh
