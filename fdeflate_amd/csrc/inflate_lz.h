// inflate_lz.h -- the LZ-window decoder: any zlib stream of Huffman blocks (dynamic or fixed, real
// distances, several blocks), one stream per wavefront, the sliding window in LDS.
//
// Restates (behaviour, not code) the compressed-block loop of the reference,
// src/decompress.rs:611-1018 (match copy :782-829), with the tables of src/huffman.rs:18-184 as
// inflate_tables.h builds them.  tests/lz_model.py is an executable CPU model of the algorithm.
//
// A block's data is decoded in SUPER-SPANS of 64 x P x Q stream bits (Q = kLzRange, P <= kLzMaxPhases): lane l owns
// the P x Q bits from l P Q on and walks them in P phases of Q bits (lz_superspan, at the end of this file):
//
//   pass 1   a lane walks a guessed chain from up to kLzWarm bits in front of its range (impossible tokens slide
//            on by one bit; Huffman codes self-synchronise: 86 % of the guesses of the bench's zlib-6 streams have
//            merged with the real chain after 256 bits), then its phases; per phase it leaves an ITEM in global
//            scratch: where the phase's first step starts, the output bytes and the matches of the steps that
//            START in the phase.
//   check    a lane's first step must be where its left neighbour's chain left the neighbour's range; by
//            induction from lane 0 (whose start is real) every chain is then the real one.  Lanes that fail walk
//            again from the neighbour's end until they meet their old chain at a phase boundary.
//   images   the items in lane-major order are the stream in order.  64 items at a time (as many as fit an IMAGE
//            of kLzImgCap output bytes / kLzIdxCap matches): output offsets by a wavefront prefix sum, then
//   pass 2   a lane decodes its item again: literals go to the image -- the next piece of the output ring in LDS,
//            which also holds the history in front of it -- a match leaves a 3-byte descriptor (length - 3,
//            distance - 1) in the first three bytes of its place and an entry in the image's match list.
//   resolve  by match, 64 at a time in stream order (lane = match): a match whose source bytes end in front of the
//            batch's first match depends on nothing that is still missing -- nine in ten on the bench data -- and
//            all of those are copied at once, a lane each; the others follow in order against a frontier.  Long or
//            self-overlapping matches are copied by the whole wavefront.  Sources older than the ring come from the
//            output slot in global memory (L2): four unaligned dword loads per match, requested a batch ahead.
//   flush    whole 16-B lines of the image to the slot, the Adler-32 folded in (as flush_ring).
//
// Tables are the kernel's own (lz_build_tables / lz_parse_dynamic): a 9-bit literal/length table whose entries
// carry up to two literals, or a length with its extra bits, and an 8-bit distance table, each with a second level
// (kLzSub / kLzDsub entries) for the longer codes; what fits neither takes the canonical walk of lz_token_slow.
//
// The kernel only ever reports Ok: anything else (stored blocks, errors, truncation, a slot that is
// too small, a wrong checksum) leaves the stream PENDING for the exact kernels behind it.
#pragma once
#include "inflate_stream.h"
#include <type_traits>

namespace fdh {

#ifndef FDH_LZ_LIT_BITS
#define FDH_LZ_LIT_BITS 9
#endif
#ifndef FDH_LZ_DIST_BITS
#define FDH_LZ_DIST_BITS 8
#endif
constexpr int kLzLitBits = FDH_LZ_LIT_BITS;    // index bits of the literal/length table (longer codes: second level)
constexpr int kLzDistBits = FDH_LZ_DIST_BITS;  // index bits of the distance table (longer codes: the slow step)
constexpr uint32_t kLzLitMask = (1u << kLzLitBits) - 1, kLzDistMask = (1u << kLzDistBits) - 1;
constexpr int kLzLitLong = 15 - kLzLitBits, kLzDistLong = 15 - kLzDistBits;  // code lengths beyond the indices
static_assert(kLzLitBits >= 9 && kLzLitBits <= 12 && kLzDistBits >= 7 && kLzDistBits <= 10, "table index bits");
#ifndef FDH_LZ_RING
#define FDH_LZ_RING 3584
#endif
#ifndef FDH_LZ_WAVES_PER_CU
#define FDH_LZ_WAVES_PER_CU 11
#endif
#ifndef FDH_LZ_RANGE
#define FDH_LZ_RANGE 192
#endif
#ifndef FDH_LZ_IMG
#define FDH_LZ_IMG 3328
#endif
#ifndef FDH_LZ_WARM
#define FDH_LZ_WARM 192
#endif
constexpr uint32_t kLzRange = FDH_LZ_RANGE;  // stream bits of a lane's range walked per phase (an item of pass 2)
constexpr uint32_t kLzWarm = FDH_LZ_WARM;    // bits a guessed chain walks in front of its range
constexpr uint32_t kLzImgCap = FDH_LZ_IMG;   // output bytes resolved at a time (an image)
#ifndef FDH_LZ_UNROLL
#define FDH_LZ_UNROLL 4
#endif
constexpr int kLzUnroll = FDH_LZ_UNROLL;       // fast steps per loop trip of a walk
constexpr int kLzMaxPhases = 16;             // phases of a super-span: a lane's range is at most 16 x kLzRange bits
static_assert(kLzWarm <= kLzRange, "the warm-up walk uses a lane's stage slot like a phase");
constexpr uint32_t kLzRing = FDH_LZ_RING;   // history + image, a multiple of 64
constexpr bool kLzRingPow2 = (kLzRing & (kLzRing - 1)) == 0;
// A lane's stage slot: the dwords of one phase -- 31 bits of alignment + the phase + the 96-bit window of the last
// step that starts inside it -- an odd number, so that the lanes' slots start in different LDS banks.
constexpr uint32_t kLzSlotDw = ((31 + kLzRange + 96 + 31) / 32) | 1u;
constexpr uint32_t kLzStageDw = 64 * kLzSlotDw;
static_assert(kLzStageDw >= 256, "the far buffer and the header parser borrow the first KiB of the stage");
constexpr uint32_t kLzIdxCap = kLzImgCap / 4;  // matches of one span (a match is at least three bytes; the bench's zlib-6 streams: one per 5.2)
static_assert(kLzRing % 64 == 0 && kLzRing >= kLzImgCap + 256, "ring = image + history");

struct __attribute__((aligned(16))) LzWork {
    uint32_t stage[kLzStageDw];    // the span's stream bytes (coalesced copy)
    uint16_t idx[kLzIdxCap + 2];   // where the image's matches start, in stream order (+ a spare entry for stores that are not wanted)
};
#ifndef FDH_LZ_SUB
#define FDH_LZ_SUB 256
#endif
#ifndef FDH_LZ_DSUB
#define FDH_LZ_DSUB 64
#endif
constexpr uint32_t kLzSub = FDH_LZ_SUB;   // second-level entries of the literal/length table (codes beyond its index)
constexpr uint32_t kLzDsub = FDH_LZ_DSUB;  // ... and of the distance table
// Decode tables of the current block in the walk's entry layout (below), with the canonical bookkeeping the
// second level and the slow step read.
struct __attribute__((aligned(16))) LzTables {
    uint32_t lit[1 << kLzLitBits];
    uint32_t dist[1 << kLzDistBits];
    CodeBook lit_cb;
    CodeBook dist_cb;
    uint16_t lit_sorted[288];
    uint16_t dist_sorted[32];
};
struct __attribute__((aligned(16))) LzLds {
    LzTables tables;
    uint32_t sub[kLzSub];          // second level of tables.lit (lz_build_sub)
    uint32_t dsub[kLzDsub];        // second level of tables.dist
    union {
        // block headers are parsed by the wave-serial reader (inflate_stream.h) between spans: it only ever
        // touches WaveIo::in_ring (its first member): the tables are this kernel's own business
        struct {
            uint32_t in_ring[kInRingDw];
        } hdr;
        LzWork w;
    } u;
    uint8_t ring[kLzRing + 16];  // output position p lives at (p + gmis) mod kLzRing; 3 guard bytes for descriptors
};
static_assert((sizeof(LzLds) + 1024) * FDH_LZ_WAVES_PER_CU <= 163840, "wavefronts per CU (LDS is handed out in larger pieces than a byte: keep a margin)");
static_assert(offsetof(WaveIo, in_ring) == 0, "the reader's window is the first member of WaveIo");

__device__ __forceinline__ uint32_t lz_wrap(uint32_t i) {  // i < 2 * kLzRing
    if (kLzRingPow2) return i & (kLzRing - 1);
    return i >= kLzRing ? i - kLzRing : i;
}
__device__ __forceinline__ uint32_t lz_back(uint32_t i, uint32_t back) {  // ring index `back` (<= kLzRing) in front of i
    if (kLzRingPow2) return (i - back) & (kLzRing - 1);
    const int32_t r = (int32_t)i - (int32_t)back;
    return (uint32_t)(r < 0 ? r + (int32_t)kLzRing : r);
}

#ifdef FDH_LZ_DEBUG
__device__ unsigned long long g_lzstat[32];
#define LZT(o, k) do { const long long tn_ = clock64(); (o).t[k] += (unsigned long long)(tn_ - (o).tq); (o).tq = tn_; } while (0)
#define LZC(o, k, v) do { (o).t[k] += (unsigned long long)(v); } while (0)
#else
#define LZT(o, k) do { } while (0)
#define LZC(o, k, v) do { } while (0)
#endif

// ---- the walk's table entries (built from the device tables of inflate_tables.h) ----
//   literal/length, index = low 10 stream bits:
//     literals  [4:0] bits of the step, [6:5] literals (1 / 2), [15:8] first, [23:16] second
//     length    bit 31; [4:0] code + extra bits, [7:5] extra bits, [11:8] code bits, [24:16] length base
//     special   bit 30; [4:0] bits, [29:28] 0 end-of-block / 1 code beyond the index (canonical walk) / 2 impossible /
//               3 code beyond the index with a second level: [4:0] its index bits (stream bits 10 ..), [27:8] its
//               first entry in LzLds::sub (entries there are literal / length / end-of-block entries with the full code length)
//   distance, index = low 9 bits:
//     [4:0] code + extra bits, [8:5] code bits, [12:9] extra bits, [31:16] base; bit 13 special ([14]: code beyond the index,
//     canonical walk; [15]: code beyond the index with a second level: [4:0] its index bits, [31:16] its first entry in LzLds::dsub)
constexpr uint8_t kClclOrderHost[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};  // src/tables.rs:63-65
constexpr uint32_t LZW_LEN = 1u << 31, LZW_SPECIAL = 1u << 30, LZD_SPECIAL = 1u << 13, LZD_LONG = 1u << 14, LZD_TWO = 1u << 15;
constexpr uint32_t LZW_TWO = LZW_SPECIAL | (3u << 28);  // (e & LZW_TWO) == LZW_TWO: second-level look-up

__device__ __forceinline__ uint32_t lz_conv_lit(uint32_t e) {
    const uint32_t nb = e & 15, k = (e >> 4) & 15;
    if (k == K_LIT1) return nb | (1u << 5) | (e & 0xFF00u);
    if (k == K_LIT2) return nb | (2u << 5) | (e & 0xFFFF00u);
    if (k == K_LEN) {
        const uint32_t ex = (e >> 8) & 31, base = e >> 16;
        return LZW_LEN | (nb + ex) | (ex << 5) | (nb << 8) | (base << 16);
    }
    if (k == K_EOB) return LZW_SPECIAL | nb;
    if (k == K_LONG) return LZW_SPECIAL | (1u << 28);
    return LZW_SPECIAL | (2u << 28);
}
__device__ __forceinline__ uint32_t lz_conv_dist(uint32_t de) {
    const uint32_t k = (de >> 4) & 15;
    if (k == D_DIST) {
        const uint32_t dcb = de & 15, dex = (de >> 8) & 15;
        return (dcb + dex) | (dcb << 5) | (dex << 9) | (de & 0xFFFF0000u);
    }
    return k == D_LONG ? (LZD_SPECIAL | LZD_LONG) : LZD_SPECIAL;
}
// Second level of the literal/length table (the reference's secondary tables, src/huffman.rs:138-181, in
// this kernel's entry layout): every 10-bit prefix that longer codes share gets 2^(longest - 10) entries
// of LzLds::sub.  Prefixes that do not fit keep the canonical-walk marker.
// Call after lz_convert_tables; CodeBook / sorted symbols as build_table left them.
template <bool DIST>
__device__ __forceinline__ void lz_build_sub_t(LzLds& L, int lane) {
    LzTables& T = L.tables;
    constexpr int BITS = DIST ? kLzDistBits : kLzLitBits;
    constexpr int NL = 15 - BITS;  // lengths BITS + 1 .. 15
    constexpr uint32_t MASK = (1u << BITS) - 1, CAP = DIST ? kLzDsub : kLzSub;
    const CodeBook& cb = DIST ? T.dist_cb : T.lit_cb;
    const uint16_t* const sorted = DIST ? T.dist_sorted : T.lit_sorted;
    uint32_t* const prim = DIST ? T.dist : T.lit;
    uint32_t* const sub = DIST ? L.dsub : L.sub;
    uint32_t nsyms = 0, offs[NL], first[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) {
        offs[i] = uni(cb.offs[BITS + 1 + i]);
        first[i] = uni(cb.first[BITS + 1 + i]);
    }
    nsyms = offs[NL - 1] + uni(cb.hist[15]);
    if (nsyms == offs[0]) return;  // no code beyond the index
    // what the table fill left in the shared prefixes: "code beyond the index: canonical walk"
    constexpr uint32_t kWalk = DIST ? (LZD_SPECIAL | LZD_LONG) : (LZW_SPECIAL | (1u << 28));
    auto code_of = [&](uint32_t j, uint32_t& l, uint32_t& rev) __attribute__((always_inline)) {
        uint32_t li = 0;
#pragma unroll
        for (int i = 1; i < NL; i++) li += j >= offs[i] ? 1u : 0u;
        uint32_t o = offs[0], f = first[0];
#pragma unroll
        for (int i = 1; i < NL; i++) {  // (vsel: hipcc turns a plain chain of selects into a table in scratch memory)
            o = vsel(li == (uint32_t)i, offs[i], o);
            f = vsel(li == (uint32_t)i, first[i], f);
        }
        l = BITS + 1 + li;
        rev = __brev(f + (j - o)) >> (32 - l);
    };
    // the longest code of every prefix, kept in the prefix's own entry: marker | length
    for (uint32_t j = offs[0] + (uint32_t)lane; j < nsyms; j += kWave) {
        uint32_t l, rev;
        code_of(j, l, rev);
        atomicMax(&prim[rev & MASK], kWalk | l);
    }
    wave_sync();
    {   // allocation: lane i owns the prefixes PER i .. PER i + PER - 1
        constexpr int PER = (1 << BITS) / kWave;
        uint32_t sz[PER], tot = 0;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const uint32_t e = prim[PER * lane + i];
            const uint32_t m = (e & 0xFFFFFFE0u) == kWalk ? (e & 31) : 0u;
            sz[i] = m ? 1u << (m - BITS) : 0u;
            tot += sz[i];
        }
        uint32_t off = wave_scan_add(tot) - tot;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            if (sz[i]) {
                const uint32_t m = prim[PER * lane + i] & 31;
                const uint32_t two = DIST ? (LZD_SPECIAL | LZD_TWO | (off << 16) | (m - BITS)) : (LZW_TWO | (off << 8) | (m - BITS));
                prim[PER * lane + i] = off + sz[i] <= CAP ? two : kWalk;
                off += sz[i];
            }
        }
    }
    wave_sync();
    for (uint32_t j = offs[0] + (uint32_t)lane; j < nsyms; j += kWave) {
        uint32_t l, rev;
        code_of(j, l, rev);
        const uint32_t t = prim[rev & MASK];
        uint32_t longest, off;
        if (DIST) {
            if ((t & (LZD_SPECIAL | LZD_TWO)) != (LZD_SPECIAL | LZD_TWO)) continue;  // did not fit: the canonical walk stays
            longest = BITS + (t & 31);
            off = t >> 16;
        } else {
            if ((t & LZW_TWO) != LZW_TWO) continue;
            longest = BITS + (t & 31);
            off = (t >> 8) & 0xFFFFF;
        }
        const uint32_t e = DIST ? lz_conv_dist(DistTraits::entry(sorted[j], l)) : lz_conv_lit(LitlenTraitsT<kLzLitBits>::entry(sorted[j], l));
        const uint32_t step = 1u << (l - BITS), n = 1u << (longest - l);
        uint32_t at = off + (rev >> BITS);
        for (uint32_t i = 0; i < n; i++, at += step) sub[at] = e;
    }
    wave_sync();
}
__device__ __forceinline__ void lz_build_sub(LzLds& L, int lane) {
    lz_build_sub_t<false>(L, lane);
    lz_build_sub_t<true>(L, lane);
}

struct LzTok {
    uint32_t kind;  // 0 literal(s), 1 match, 2 end-of-block, 3 impossible
    uint32_t bits, n, v;  // v: literals b0 | b1 << 8 / match: length | dist << 16
};

// Bounds of the canonical codes beyond the primary tables (CodeBook::run after build_table: the
// left-justified 16-bit bound of the codes of length <= l), uniform per block: kept in scalar registers.
struct LzBounds {
    uint32_t lit[kLzLitLong];    // lengths kLzLitBits + 1 .. 15
    uint32_t dist[kLzDistLong];  // lengths kLzDistBits + 1 .. 15
};
__device__ __forceinline__ LzBounds lz_load_bounds(const LzTables& T) {
    LzBounds b;
#pragma unroll
    for (int i = 0; i < kLzLitLong; i++) b.lit[i] = uni(T.lit_cb.run[kLzLitBits + 1 + i]);
#pragma unroll
    for (int i = 0; i < kLzDistLong; i++) b.dist[i] = uni(T.dist_cb.run[kLzDistBits + 1 + i]);
    return b;
}

// The token in front of (hi:lo), every case: codes beyond the primary tables are resolved against the
// canonical bounds (long_walk of inflate_tables.h with the bounds in registers).
__device__ __forceinline__ LzTok lz_token_slow(const LzTables& T, const uint32_t* sub, const uint32_t* dsub, const LzBounds& bd, uint32_t lo, uint32_t hi) {
    LzTok t;
    t.kind = 3;
    t.bits = t.n = t.v = 0;
    uint32_t e = T.lit[lo & kLzLitMask];
    if ((e & LZW_TWO) == LZW_TWO) e = sub[((e >> 8) & 0xFFFFF) + __builtin_amdgcn_ubfe(lo, kLzLitBits, e & 31)];
    if ((e & LZW_SPECIAL) && ((e >> 28) & 3) == 1) {
        const uint32_t r16 = __brev(lo) >> 16;
        uint32_t len = kLzLitBits + 1;
#pragma unroll
        for (int i = 0; i < kLzLitLong - 1; i++) len += r16 >= bd.lit[i] ? 1u : 0u;
        if (r16 >= bd.lit[kLzLitLong - 1]) return t;
        const uint32_t d = (r16 >> (16 - len)) - T.lit_cb.first[len];
        const uint32_t sym = T.lit_sorted[T.lit_cb.offs[len] + d];
        e = lz_conv_lit(LitlenTraitsT<kLzLitBits>::entry(sym, len));
    }
    if (e & LZW_SPECIAL) {
        if (((e >> 28) & 3) == 0) {
            t.kind = 2;
            t.bits = e & 31;
        }
        return t;
    }
    if (!(e & LZW_LEN)) {
        t.kind = 0;
        t.bits = e & 31;
        t.n = (e >> 5) & 3;
        t.v = (e >> 8) & 0xFFFF;
        return t;
    }
    const uint32_t tb = e & 31, ex = (e >> 5) & 7, nb = (e >> 8) & 15;
    const uint32_t length = ((e >> 16) & 0x1FF) + __builtin_amdgcn_ubfe(lo, nb, ex);
    const uint32_t dv = __builtin_amdgcn_alignbit(hi, lo, tb);
    uint32_t de = T.dist[dv & kLzDistMask];
    if ((de & (LZD_SPECIAL | LZD_TWO)) == (LZD_SPECIAL | LZD_TWO)) de = dsub[(de >> 16) + __builtin_amdgcn_ubfe(dv, kLzDistBits, de & 31)];
    if (de & LZD_SPECIAL) {
        if (!(de & LZD_LONG)) return t;
        const uint32_t r16 = __brev(dv) >> 16;
        uint32_t len = kLzDistBits + 1;
#pragma unroll
        for (int i = 0; i < kLzDistLong - 1; i++) len += r16 >= bd.dist[i] ? 1u : 0u;
        if (r16 >= bd.dist[kLzDistLong - 1]) return t;
        const uint32_t d = (r16 >> (16 - len)) - T.dist_cb.first[len];
        const uint32_t sym = T.dist_sorted[T.dist_cb.offs[len] + d];
        de = lz_conv_dist(DistTraits::entry(sym, len));
        if (de & LZD_SPECIAL) return t;
    }
    const uint32_t dist = (de >> 16) + __builtin_amdgcn_ubfe(dv, (de >> 5) & 15, (de >> 9) & 15);
    t.kind = 1;
    t.bits = tb + (de & 31);
    t.n = length;
    t.v = length | (dist << 16);
    return t;
}

struct LzWalk {
    uint32_t e;     // where the chain left the range (first step at or behind its end), or where it stopped
    uint32_t cnt;   // output bytes of the steps taken
    uint32_t nm;    // matches among them
    uint32_t stop;  // 0 none, 1 end-of-block, 2 impossible token / past the end of the input
    uint32_t stop_bits;
};

// One lane's walk from `pos` to the first step at or behind `end`; positions are window bits (stream bit +
// 8 x the misalignment of the stream's first byte), the lane's stage slot holds the dwords from window
// dword `slot_dw`.  `slide`: a guessed chain in front of its range slides over impossible tokens by one
// bit instead of stopping.  EMIT: pass 2 -- `q_rel` is the lane's offset in the image, `mi` the index of
// its first match in the image's list.  `trouble` collects what must never happen on a real chain.
//
// The fast step is branch-free and the same for every lane: a lane in front of a special token (a code
// beyond the tables, end-of-block, an impossible token) simply does not advance; behind every fourth
// step those lanes take the slow step, which knows every case.
template <bool EMIT>
__device__ __forceinline__ LzWalk lz_walk(LzLds& L, const LzBounds& bd, uint32_t pos, const uint32_t end, const bool slide,
                                          const bool active, const uint32_t limit, const uint32_t slot_dw, uint32_t q_rel,
                                          const uint32_t o_ri, const uint32_t o_abs, uint32_t mi, bool& trouble, uint32_t* iters = nullptr,
                                          uint32_t* slows = nullptr) {
    const LzTables& T = L.tables;
    LzWalk w;
    uint32_t cnt = 0, nm = 0, stop = 0, stop_bits = 0;
    bool run = active && pos < end;
    uint32_t it = 0;
    uint32_t dbad = 0;
    const uint32_t* const slot = &L.u.w.stage[kLzSlotDw * (threadIdx.x & 63)];
    const uint32_t bit0 = slot_dw * 32;
    while (__any(run)) {
        it += kLzUnroll;
        if (it > 8192) {  // cannot happen; never hang
            trouble = true;
            break;
        }
        bool special = false;
        // four fast steps without a branch between them (the scalar unit is shared by the CU's wavefronts: loop
        // control and execution-mask work per step is what eleven of them queue for); a lane that is done or
        // stands in front of a special token does not move
#pragma unroll
        for (int u = 0; u < kLzUnroll; u++) {
            const uint32_t rel = run ? pos - bit0 : 0u, di = rel >> 5, sh = rel & 31;
            const uint32_t r0 = slot[di], r1 = slot[di + 1], r2 = slot[di + 2];
            const uint32_t lo = __builtin_amdgcn_alignbit(r1, r0, sh), hi = __builtin_amdgcn_alignbit(r2, r1, sh);
            uint32_t e = T.lit[lo & kLzLitMask];
            {   // a code beyond the index: its second-level entry (every lane looks one up: no branch)
                const bool two = (e & LZW_TWO) == LZW_TWO;
                const uint32_t e2 = L.sub[two ? ((e >> 8) & 0xFFFFF) + __builtin_amdgcn_ubfe(lo, kLzLitBits, e & 31) : 0u];
                e = two ? e2 : e;
            }
            const uint32_t tb = e & 31;
            const bool is_len = (int32_t)e < 0;
            const uint32_t length = ((e >> 16) & 0x1FF) + __builtin_amdgcn_ubfe(lo, (e >> 8) & 15, (e >> 5) & 7);
            const uint32_t dv = __builtin_amdgcn_alignbit(hi, lo, tb);
            uint32_t de = T.dist[dv & kLzDistMask];
            {   // a distance code beyond the index: its second-level entry (every lane looks one up: no branch)
                const bool two = (de & (LZD_SPECIAL | LZD_TWO)) == (LZD_SPECIAL | LZD_TWO);
                const uint32_t d2 = L.dsub[two ? (de >> 16) + __builtin_amdgcn_ubfe(dv, kLzDistBits, de & 31) : 0u];
                de = two ? d2 : de;
            }
            special = (e & LZW_SPECIAL) != 0 || (is_len && (de & LZD_SPECIAL) != 0);  // (of the last step: what the slow step looks at)
            const uint32_t bits = tb + (is_len ? (de & 31) : 0u);
            const uint32_t inc = is_len ? length : ((e >> 5) & 3);
            const bool go = run && !special;
            if (EMIT) {
                // every lane stores three bytes and a list entry, whatever its token: what is not wanted goes to a
                // spare byte behind the ring / a spare entry behind the list (a select instead of an execution mask)
                const uint32_t qi = lz_wrap(o_ri + q_rel);
                const uint32_t dist = (de >> 16) + __builtin_amdgcn_ubfe(dv, (de >> 5) & 15, (de >> 9) & 15);
                const bool gm = go && is_len;
                dbad |= (gm && dist > o_abs + q_rel) ? 1u : 0u;  // src/decompress.rs:782: the exact kernels report it
                const uint32_t d = (length - 3) | ((dist - 1) << 8);  // a match leaves its descriptor (past the ring's end: guard bytes)
                const uint32_t v3 = is_len ? d : (e >> 8);
                const uint32_t a1 = is_len ? qi + 1 : lz_wrap(qi + 1);
                L.ring[go ? qi : kLzRing + 12] = (uint8_t)v3;
                L.ring[(gm || (go && inc == 2)) ? a1 : kLzRing + 12] = (uint8_t)(v3 >> 8);
                L.ring[gm ? qi + 2 : kLzRing + 12] = (uint8_t)(v3 >> 16);
                L.u.w.idx[gm ? min(mi, kLzIdxCap - 1) : kLzIdxCap] = (uint16_t)q_rel;
                mi += gm ? 1u : 0u;
                q_rel += go ? inc : 0u;
            }
            cnt += go ? inc : 0u;
            nm += (go && is_len) ? 1u : 0u;
            pos += go ? bits : 0u;
            run = run && pos < end;
        }
        {
            const bool act = run && special;
            if (__any(act)) {
                const uint32_t rel = act ? pos - bit0 : 0u, di = rel >> 5, sh = rel & 31;
                const uint32_t r0 = slot[di], r1 = slot[di + 1], r2 = slot[di + 2];
                const uint32_t lo = __builtin_amdgcn_alignbit(r1, r0, sh), hi = __builtin_amdgcn_alignbit(r2, r1, sh);
                const LzTok tk = lz_token_slow(T, L.sub, L.dsub, bd, lo, hi);
                if (slows) *slows += 1;
                if (act) {
                    if (tk.kind >= 2) {
                        if (slide) {
                            pos += 1;  // a guessed chain in front of its range: slide on
                            run = pos < end;
                        } else {
                            stop = tk.kind == 2 ? 1u : 2u;
                            stop_bits = tk.bits;
                            run = false;
                        }
                    } else {
                        cnt += tk.n;
                        if (EMIT) {
                            const uint32_t qi = lz_wrap(o_ri + q_rel);
                            if (tk.kind == 0) {
                                L.ring[qi] = (uint8_t)tk.v;
                                if (tk.n == 2) L.ring[lz_wrap(qi + 1)] = (uint8_t)(tk.v >> 8);
                            } else {
                                const uint32_t length = tk.v & 0xFFFF, dist = tk.v >> 16;
                                dbad |= dist > o_abs + q_rel ? 1u : 0u;
                                const uint32_t d = (length - 3) | ((dist - 1) << 8);
                                L.ring[qi] = (uint8_t)d;
                                L.ring[qi + 1] = (uint8_t)(d >> 8);
                                L.ring[qi + 2] = (uint8_t)(d >> 16);
                                L.u.w.idx[min(mi, kLzIdxCap - 1)] = (uint16_t)q_rel;
                                mi++;
                            }
                            q_rel += tk.n;
                        }
                        nm += tk.kind == 1 ? 1u : 0u;
                        pos += tk.bits;
                        run = pos < end;
                    }
                }
            }
        }
    }
    if (dbad) trouble = true;
    // ran past the end of the input (zeros are staged there) -- an end-of-block code found in those zeros is none
    if (active && (pos > limit || (stop == 1 && pos + stop_bits > limit))) stop = 2;
    w.e = pos;
    w.cnt = cnt;
    w.nm = nm;
    w.stop = stop;
    w.stop_bits = stop_bits;
    if (iters) *iters += it;
    return w;
}

// 16 stream bytes at window offset `w0` (bytes from base16); zero beyond the stream, careful at the
// ends of the packed batch (as Inflater::load_chunk).
__device__ __forceinline__ uint4 lz_load16(const uint8_t* base16, uint64_t w0, uint64_t win_bytes, const uint8_t* buf_lo,
                                           const uint8_t* buf_hi) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (w0 < win_bytes) {
        const uint8_t* p = base16 + w0;
        if (p >= buf_lo && p + 16 <= buf_hi) {
            v = *reinterpret_cast<const uint4*>(p);
        } else {
            uint64_t lo = 0, hi = 0;
            for (int j = 0; j < 8; j++) {
                if (p + j >= buf_lo && p + j < buf_hi) lo |= (uint64_t)p[j] << (j * 8);
                if (p + 8 + j >= buf_lo && p + 8 + j < buf_hi) hi |= (uint64_t)p[8 + j] << (j * 8);
            }
            v = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
        }
        const uint64_t rem = win_bytes - w0;
        if (rem < 16) {
            const uint32_t r = (uint32_t)rem;
            v.x = r >= 4 ? v.x : (r == 0 ? 0 : v.x & ((1u << (r * 8)) - 1));
            v.y = r >= 8 ? v.y : (r <= 4 ? 0 : v.y & ((1u << ((r - 4) * 8)) - 1));
            v.z = r >= 12 ? v.z : (r <= 8 ? 0 : v.z & ((1u << ((r - 8) * 8)) - 1));
            v.w = r <= 12 ? 0 : v.w & ((1u << ((r - 12) * 8)) - 1);
        }
    }
    return v;
}

// The dwords of a lane's next walk into its stage slot: kLzSlotDw dwords from window dword `dw` (zero beyond
// the stream, careful at the ends of the packed batch).  Returns `dw`.
struct __attribute__((packed, aligned(4))) LzU4 {
    uint32_t x, y, z, w;
};
__device__ __forceinline__ uint32_t lz_stage_slot(LzLds& L, const bool active, const uint32_t pos, const uint8_t* base16,
                                                  const uint64_t win_bytes, const uint8_t* buf_lo, const uint8_t* buf_hi, const int lane) {
    const uint32_t dw = pos >> 5;
    uint32_t* const slot = &L.u.w.stage[kLzSlotDw * (uint32_t)lane];
    if (active) {
#pragma unroll
        for (uint32_t k = 0; k < kLzSlotDw; k += 4) {
            const uint64_t w0 = ((uint64_t)dw + k) * 4;
            const uint8_t* p = base16 + w0;
            uint32_t v[4] = {0, 0, 0, 0};
            if (w0 + 16 <= win_bytes && p >= buf_lo && p + 16 <= buf_hi) {
                const LzU4 q = *reinterpret_cast<const LzU4*>(p);
                v[0] = q.x;
                v[1] = q.y;
                v[2] = q.z;
                v[3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    if (w0 + j < win_bytes && p + j >= buf_lo && p + j < buf_hi) v[j >> 2] |= (uint32_t)p[j] << (8 * (j & 3));
                }
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; j++)
                if (k + j < kLzSlotDw) slot[k + j] = v[j];
        }
    }
    return dw;
}

// Builds the walk tables, the canonical bookkeeping (and nothing else) from the code lengths: ll[k] = length
// of literal/length symbol lane + 64 k, dl = length of distance symbol `lane`.  false unless the codes are
// what the reference accepts (src/decompress.rs:561-606: complete, or no / one distance code).
template <class OUT>
__device__ __forceinline__ bool lz_build_tables(LzLds& L, const uint32_t (&ll)[5], const uint32_t dl, const int lane, OUT& o) {
    LzTables& T = L.tables;
    auto rank_in = [&](uint64_t m) __attribute__((always_inline)) -> uint32_t {  // lanes of m below this one
        return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    };
    // ---- canonical bookkeeping of a code: lengths ranked with ballots ----
    // sym_len(k): length of symbol lane + 64 k.  Writes sorted[], the CodeBook; returns false unless complete.
    auto canon = [&](auto&& sym_len, auto rounds_c, CodeBook& cb, uint16_t* sorted, uint32_t (&bound)[16], uint32_t& nsyms,
                     uint32_t& maxlen) __attribute__((always_inline)) -> bool {
        constexpr int rounds = decltype(rounds_c)::value;
        uint32_t hist[16], first[16], offs[16];
        hist[0] = 0;
#pragma unroll
        for (uint32_t l = 1; l <= 15; l++) {
            uint32_t n = 0;
#pragma unroll
            for (int k = 0; k < rounds; k++) n += (uint32_t)__popcll(__ballot(sym_len(k) == l));
            hist[l] = n;
        }
        uint32_t kraft = 0, code = 0, off = 0;
        nsyms = 0;
        maxlen = 0;
#pragma unroll
        for (uint32_t l = 1; l <= 15; l++) {
            kraft += hist[l] << (15 - l);
            code = (code + hist[l - 1]) << 1;
            first[l] = code;
            offs[l] = off;
            off += hist[l];
            bound[l] = (code + hist[l]) << (16 - l);
            if (hist[l]) maxlen = l;
        }
        nsyms = off;
        // (lane l keeps the books the slow path and the second level read)
#pragma unroll
        for (uint32_t l = 1; l <= 15; l++) {
            if (lane == (int)l) {
                cb.hist[l] = hist[l];
                cb.first[l] = first[l];
                cb.offs[l] = offs[l];
                cb.run[l] = bound[l];
            }
        }
#pragma unroll
        for (uint32_t l = 1; l <= 15; l++) {
            uint32_t seen = 0;
#pragma unroll
            for (int k = 0; k < rounds; k++) {
                const uint64_t m = __ballot(sym_len(k) == l);
                if (sym_len(k) == l) sorted[offs[l] + seen + rank_in(m)] = (uint16_t)(lane + 64 * k);
                seen += (uint32_t)__popcll(m);
            }
        }
        return kraft == (1u << 15);
    };
    uint32_t lb[16], db[16], ln, lmax, dn, dmax;
    const bool lit_ok = canon([&](int k) { return ll[k]; }, std::integral_constant<int, 5>{}, T.lit_cb, T.lit_sorted, lb, ln, lmax);
    if (!lit_ok) return false;  // src/decompress.rs:570-580
    const bool dist_ok = canon([&](int) { return dl; }, std::integral_constant<int, 1>{}, T.dist_cb, T.dist_sorted, db, dn, dmax);
    const bool dist_none = dn == 0, dist_one = dn == 1 && dmax == 1;  // src/decompress.rs:588-589, src/huffman.rs:45-58
    if (!dist_ok && !dist_none && !dist_one) return false;
    wave_sync();
    LZT(o, 15);

    // ---- every table index decodes itself ----
    auto decode = [&](const uint32_t r16, const uint32_t (&bound)[16], const int maxbits, const CodeBook& cb, const uint16_t* sorted,
                      uint32_t& len) __attribute__((always_inline)) -> uint32_t {
        len = 1;
#pragma unroll
        for (int l = 1; l < 15; l++)
            if (l < maxbits) len += r16 >= bound[l] ? 1u : 0u;
        if (r16 >= bound[maxbits]) {  // a code longer than the index
            len = 0;
            return 0;
        }
        return sorted[cb.offs[len] + (r16 >> (16 - len)) - cb.first[len]];
    };
    auto lit_entry = [&](const uint32_t sym, const uint32_t len) __attribute__((always_inline)) -> uint32_t {
        if (sym < 256) return len | (1u << 5) | (sym << 8);
        if (sym == 256 || sym >= 286) return LZW_SPECIAL | len;  // 286 / 287: reference src/tables.rs:100
        const uint32_t eb = len_extra_base(sym - 257), ex = eb & 0xFF, base = eb >> 8;
        return LZW_LEN | (len + ex) | (ex << 5) | (len << 8) | (base << 16);
    };
    for (int it = 0; it < (1 << kLzLitBits) / kWave; it++) {
        const uint32_t idx = (uint32_t)lane + 64u * it;
        uint32_t len;
        const uint32_t sym = decode(__brev(idx) >> 16, lb, kLzLitBits, T.lit_cb, T.lit_sorted, len);
        uint32_t e = LZW_SPECIAL | (1u << 28);  // code beyond the index: canonical walk unless the second level takes it
        if (len) {
            e = lit_entry(sym, len);
            if (sym < 256 && len < (uint32_t)kLzLitBits) {  // a second literal whose code fits the index as well
                uint32_t len2;
                const uint32_t sym2 = decode(__brev(idx >> len) >> 16, lb, kLzLitBits, T.lit_cb, T.lit_sorted, len2);
                if (len2 && sym2 < 256 && len + len2 <= (uint32_t)kLzLitBits) e = (len + len2) | (2u << 5) | (sym << 8) | (sym2 << 16);
            }
        }
        T.lit[idx] = e;
    }
    for (int it = 0; it < (1 << kLzDistBits) / kWave; it++) {
        const uint32_t idx = (uint32_t)lane + 64u * it;
        uint32_t e = LZD_SPECIAL;
        if (dist_one) {  // one 1-bit code: '0' is that symbol, '1' is invalid
            const uint32_t sym = T.dist_sorted[0];
            if (!(idx & 1) && sym < 30) {
                const uint32_t eb = dist_extra_base(sym), ex = eb & 0xFF;
                e = (1 + ex) | (1u << 5) | (ex << 9) | ((eb >> 8) << 16);
            }
        } else if (!dist_none) {
            uint32_t len;
            const uint32_t sym = decode(__brev(idx) >> 16, db, kLzDistBits, T.dist_cb, T.dist_sorted, len);
            if (len == 0) {
                e = LZD_SPECIAL | LZD_LONG;
            } else if (sym < 30) {
                const uint32_t eb = dist_extra_base(sym), ex = eb & 0xFF;
                e = (len + ex) | (len << 5) | (ex << 9) | ((eb >> 8) << 16);
            }
        }
        T.dist[idx] = e;
    }
    wave_sync();
    LZT(o, 16);
    return true;
}

// ---- dynamic block header, this kernel's own: code-length code in registers, tables filled by index ----
// Parses a dynamic block header behind its 3 type bits + 14 count bits (hlit / hdist / hclen already read)
// and builds the walk tables, the second level and the canonical bookkeeping.  Restates
// src/decompress.rs:440-555 and huffman::build_table (src/huffman.rs:18-184) for the cases a valid stream
// produces; false = anything else (an incomplete or oversubscribed code, a repeat out of range, the end of
// the input ...): the exact kernels report it.
//
// Unlike the generic builder (inflate_tables.h) nothing here loops over a code's table slots: the code
// lengths are ranked with ballots, and every table INDEX decodes itself canonically (the length of the
// code in front of its bits is the number of left-justified bounds the bits have passed).
template <class INF, class OUT>
__device__ __forceinline__ bool lz_parse_dynamic(LzLds& L, INF& inf, const uint32_t hlit, const uint32_t hdist, const uint32_t hclen,
                                                 const int lane, OUT& o) {
    LzTables& T = L.tables;
    const uint32_t lt_lo = (uint32_t)lanemask_lt(lane), lt_hi = (uint32_t)(lanemask_lt(lane) >> 32);
    auto rank_in = [&](uint64_t m) __attribute__((always_inline)) -> uint32_t {  // lanes of m below this one
        return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    };
    (void)lt_lo;
    (void)lt_hi;
    // ---- the header's bits: 1 KiB from the current position, lane i holding dwords i, i + 64, i + 128, i + 192 ----
    // (a header is at most 57 + 316 x 14 bits; the chain below is a scalar loop, so the bits come out of
    // registers with readlane -- no LDS trip, no execution-mask juggling per token)
    const uint64_t P = inf.consumed_bits();
    const uint32_t wbit = (uint32_t)P + inf.mis * 8, c0 = (wbit >> 5) & ~3u;
    wave_sync();
    *reinterpret_cast<uint4*>(&L.u.w.stage[4 * lane]) = lz_load16(inf.base16, (uint64_t)c0 * 4 + 16u * (uint32_t)lane, inf.win_bytes, inf.buf_lo, inf.buf_hi);
    wave_sync();
    const uint32_t hA = L.u.w.stage[lane], hB = L.u.w.stage[64 + lane], hC = L.u.w.stage[128 + lane], hD = L.u.w.stage[192 + lane];
    wave_sync();
    auto word = [&](uint32_t di) __attribute__((always_inline)) -> uint32_t {
        const uint32_t a = __builtin_amdgcn_readlane(hA, di & 63), b2 = __builtin_amdgcn_readlane(hB, di & 63);
        const uint32_t c = __builtin_amdgcn_readlane(hC, di & 63), d = __builtin_amdgcn_readlane(hD, di & 63);
        return di < 64 ? a : di < 128 ? b2 : di < 192 ? c : d;
    };
    uint32_t di = (wbit - c0 * 32) >> 5, used = 0;
    const uint32_t sh0 = (wbit - c0 * 32) & 31;
    uint64_t buf = (uint64_t)(word(di) >> sh0);
    uint32_t cnt = 32 - sh0;
    di++;
    auto refill = [&]() __attribute__((always_inline)) {  // afterwards cnt >= 33
        if (cnt <= 32) {
            buf |= (uint64_t)word(di) << cnt;
            cnt += 32;
            di++;
        }
    };
    auto consume = [&](uint32_t n) __attribute__((always_inline)) {
        buf >>= n;
        cnt -= n;
        used += n;
    };
    // ---- the code-length code: 19 lengths of 3 bits, lane s holds the length of symbol s ----
    uint32_t cl = 0;
#pragma unroll
    for (int i = 0; i < 19; i++) {
        if ((uint32_t)i < hclen) {
            refill();
            const uint32_t v = (uint32_t)buf & 7;
            consume(3);
            cl = lane == (int)kClclOrderHost[i] ? v : cl;
        }
    }
    uint32_t ccode = 0;
    {
        uint32_t kraft = 0, code = 0, prev = 0;
#pragma unroll
        for (uint32_t l = 1; l <= 7; l++) {
            const uint64_t m = __ballot(cl == l);
            const uint32_t n = (uint32_t)__popcll(m);
            kraft += n << (7 - l);
            code = (code + prev) << 1;
            prev = n;
            ccode = cl == l ? code + rank_in(m) : ccode;
        }
        if (kraft != 128) return false;  // src/huffman.rs:72-75: the code-length code must be complete
    }
    // its 128-entry table in two registers: lane i holds entries i and i + 64 (symbol | bits << 8)
    uint32_t t_lo = 0, t_hi = 0;
    {
        const uint32_t crev = cl ? __brev(ccode) >> (32 - cl) : 0u;
        for (int sy = 0; sy < 19; sy++) {
            const uint32_t l = __builtin_amdgcn_readlane(cl, sy), r = __builtin_amdgcn_readlane(crev, sy);
            if (l == 0) continue;
            const uint32_t mask = (1u << l) - 1, e = (uint32_t)sy | (l << 8);
            t_lo = ((uint32_t)lane & mask) == r ? e : t_lo;
            t_hi = (((uint32_t)lane + 64u) & mask) == r ? e : t_hi;
        }
    }
    LZT(o, 13);
    // ---- the literal/length + distance code lengths: one chain of bits ----
    // 64 bit positions at a time.  Lane p decodes the code-length token that WOULD start at position p of the
    // window (the table look-up is a lane permutation: no memory); a scalar loop follows the real chain from
    // token to token -- a readlane and a handful of scalar instructions each -- and marks its lanes; those lanes
    // then write their lengths to an array in LDS side by side (offsets and the value a "repeat previous"
    // stands for by prefix scans).  The first version of this parser did everything in the scalar loop: ~500
    // clocks per token, a third of them waiting for the scalar unit the CU's wavefronts share.
    const uint32_t total = hlit + hdist;
    uint8_t* const lens = reinterpret_cast<uint8_t*>(&L.u.w.stage[256]);  // 320 code lengths, behind the staged KiB
    for (int i = lane; i < 80; i += kWave) L.u.w.stage[256 + i] = 0;
    wave_sync();
    uint32_t nread = 0, prevlen = 0;
    uint32_t lp = (wbit - c0 * 32) + used;  // the next unread bit, relative to the staged KiB
    bool bad = false;
    while (nread < total) {
        if (lp + 192 > 8192) return false;  // cannot happen: a header is at most 57 + 316 x 14 bits
        const uint32_t d0 = lp >> 5, off = (lp & 31) + (uint32_t)lane;  // this lane's token starts `off` bits into word d0
        const uint32_t x0 = word(d0), x1 = word(d0 + 1), x2 = word(d0 + 2), x3 = word(d0 + 3);
        const uint32_t sel = off >> 5;
        const uint32_t lo = sel == 0 ? x0 : (sel == 1 ? x1 : x2), hi = sel == 0 ? x1 : (sel == 1 ? x2 : x3);
        const uint32_t v = __builtin_amdgcn_alignbit(hi, lo, off & 31);
        const uint32_t idx = v & 127;
        const uint32_t ea = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx & 63) << 2), (int)t_lo);
        const uint32_t eb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx & 63) << 2), (int)t_hi);
        const uint32_t en = idx & 64 ? eb : ea;
        const uint32_t sym = en & 0xFF, nb = en >> 8;
        const uint32_t extra = sym == 16 ? 2u : sym == 17 ? 3u : sym == 18 ? 7u : 0u;
        const uint32_t rep = sym <= 15 ? 1u : (sym == 18 ? 11u : 3u) + __builtin_amdgcn_ubfe(v, nb, extra);
        const uint32_t tokv = nb == 0 ? 0u : ((nb + extra) | (rep << 8));
        // the chain: which lanes' tokens are real, how many lengths they stand for
        uint64_t chain = 0;
        uint32_t pp = 0, nr = nread;
        while (pp < 64 && nr < total) {
            const uint32_t tok = __builtin_amdgcn_readlane(tokv, pp);
            if ((tok & 0xFF) == 0) {
                bad = true;
                break;
            }
            chain |= 1ull << pp;
            nr += tok >> 8;
            pp += tok & 0xFF;
        }
        if (bad || nr > total) return false;  // an impossible code / a repeat beyond the last length
        const bool on = (chain >> lane) & 1;
        // where a lane's lengths go: exclusive prefix sum of the repeat counts along the chain
        const uint32_t incl = wave_scan_add(on ? rep : 0u);
        const uint32_t n0 = nread + incl - (on ? rep : 0u);
        // what "repeat the previous length" repeats: the nearest chain token below that is a length itself
        // (17 / 18 write zeros, and a 16 behind them repeats that zero)
        const uint32_t key = wave_scan_max((on && sym != 16) ? (((uint32_t)lane + 1) << 8) | (sym <= 15 ? sym : 0u) : 0u);
        const uint32_t value = sym <= 15 ? sym : (sym == 16 ? (key ? (key & 0xFF) : prevlen) : 0u);
        if (__any(on && sym == 16 && n0 == 0)) return false;  // nothing to repeat (src/decompress.rs:513-529)
        // short tokens: every lane writes its own (up to four lengths); long repeats: the wavefront, one by one
        if (on) {
#pragma unroll
            for (uint32_t i = 0; i < 4; i++)
                if (i < rep && rep <= 4) lens[n0 + i] = (uint8_t)value;
        }
        uint64_t longm = __ballot(on && rep > 4);
        while (longm) {
            const int f = __ffsll((unsigned long long)longm) - 1;
            longm &= longm - 1;
            const uint32_t fn0 = __builtin_amdgcn_readlane(n0, f), frep = __builtin_amdgcn_readlane(rep, f), fval = __builtin_amdgcn_readlane(value, f);
            for (uint32_t i = (uint32_t)lane; i < frep; i += kWave) lens[fn0 + i] = (uint8_t)fval;
        }
        // the last chain token's value carries over
        if (chain) {
            const int lastl = 63 - __clzll((unsigned long long)chain);
            prevlen = __builtin_amdgcn_readlane(value, lastl);
        }
        nread = nr;
        lp += pp;
    }
    used = lp - (wbit - c0 * 32);
    if ((uint64_t)used > inf.left) return false;  // the header runs past the end of the input
    inf.left -= used;
    wave_sync();
    LZT(o, 14);
    if (lens[256] == 0) return false;  // no end-of-block code (src/decompress.rs:563-566)

    uint32_t ll[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const uint32_t sy = (uint32_t)lane + 64u * k;
        ll[k] = sy < hlit ? (uint32_t)lens[sy] : 0u;
    }
    const uint32_t dl = (uint32_t)lane < hdist ? (uint32_t)lens[hlit + lane] : 0u;
    wave_sync();
    return lz_build_tables(L, ll, dl, lane, o);
}

struct LzOut {
    uint8_t* out_al;  // slot - gmis (16-B aligned)
    uint32_t gmis;
    uint32_t O;       // output bytes decoded so far = start of the next image
    uint32_t o_ri;    // ring index of position O
    uint32_t flushed; // output bytes stored to the slot and folded into the checksum
    uint32_t adler_a, adler_b;
#ifdef FDH_LZ_DEBUG
    unsigned long long t[32] = {};
    long long tq = 0;
#endif
};

// Stores [flushed, O) to the slot -- whole 16-B lines unless `final` -- and folds the bytes into the
// Adler-32 (as flush_ring: order-independent per-line sums, weights from the line's place in its block).
__device__ __forceinline__ void lz_flush(LzLds& L, LzOut& o, const bool final, const int lane) {
    wave_sync();
    const uint32_t q_lo = o.flushed + o.gmis;
    uint32_t q_hi = o.O + o.gmis;
    if (!final) q_hi &= ~15u;
    if (q_hi <= q_lo) return;
    const uint32_t it0 = q_lo & ~15u;
    // ring index of q-space position it0 (q = position + gmis; ring index = q mod kLzRing)
    uint32_t ri = lz_back(o.o_ri, o.O + o.gmis - it0);  // (the distance is < kLzRing: an image + 15)
    uint64_t acc_a = 0, acc_b = 0;
    for (uint32_t it = it0; it < q_hi; it += kWave * 16) {
        const uint32_t lq = it + (uint32_t)lane * 16;
        const uint32_t lri = lz_wrap(ri + (uint32_t)lane * 16);
        const uint32_t blk_hi = min(q_hi, it + kWave * 16), blk_lo = max(q_lo, it);
        uint32_t s = 0, t = 0;
        if (lq < blk_hi && lq + 16 > blk_lo) {
            const uint32_t lo = (blk_lo > lq) ? blk_lo - lq : 0, hi = (blk_hi < lq + 16) ? blk_hi - lq : 16;
            const uint32_t W = blk_hi - lq;  // weight of byte j is W - j
            const uint4 v = *reinterpret_cast<const uint4*>(&L.ring[lri]);
            if (lo == 0 && hi == 16) {
                *reinterpret_cast<uint4*>(o.out_al + lq) = v;
                s = bytesum4(v.x) + bytesum4(v.y) + bytesum4(v.z) + bytesum4(v.w);
                uint32_t u = bytedot4(v.x, 0x03020100u, 0);
                u = bytedot4(v.y, 0x07060504u, u);
                u = bytedot4(v.z, 0x0b0a0908u, u);
                u = bytedot4(v.w, 0x0f0e0d0cu, u);
                t = W * s - u;
            } else {
                for (uint32_t j = lo; j < hi; j++) {
                    const uint32_t word = j < 4 ? v.x : (j < 8 ? v.y : (j < 12 ? v.z : v.w));
                    const uint32_t b = (word >> (8 * (j & 3))) & 0xFFu;
                    o.out_al[lq + j] = (uint8_t)b;
                    s += b;
                    t += (W - j) * b;
                }
            }
        }
        // (per-lane partial sums; the wavefront adds them up and reduces modulo 65521 once per flush: an image
        //  is a few KiB, the sums stay far below 2^64)
        acc_a += s;
        acc_b += (uint64_t)t + (uint64_t)s * (q_hi - blk_hi);  // the bytes of this block weigh (q_hi - blk_hi) more, seen from the flush's end
        ri = lz_wrap(ri + kWave * 16);
    }
    {
        // (lane sums fit 32 bits: at most 64 lines of 16 bytes per lane and flush; the weighted sum in two halves)
        const uint64_t A = __builtin_amdgcn_readlane(wave_scan_add((uint32_t)acc_a), 63);
        const uint64_t Bv = (uint64_t)__builtin_amdgcn_readlane(wave_scan_add((uint32_t)(acc_b & 0xFFFFFu)), 63) +
                            ((uint64_t)__builtin_amdgcn_readlane(wave_scan_add((uint32_t)(acc_b >> 20)), 63) << 20);
        const uint64_t Lf = q_hi - q_lo;  // bytes of this flush
        o.adler_b = (uint32_t)(((uint64_t)o.adler_b + Lf * o.adler_a + Bv) % kAdlerMod);
        o.adler_a = (uint32_t)((o.adler_a + A) % kAdlerMod);
    }
    o.flushed = q_hi - o.gmis;
    wave_sync();
}

// ---- resolution of the image's matches: one match per lane, 64 at a time in stream order ----
struct LzBatch {
    uint32_t qi0;    // ring index of the match's first byte
    uint32_t sidx0;  // ring index of its first source byte
    uint32_t pos;    // its place in the image
    uint32_t len;    // 0: no match in this lane
    uint32_t send;   // where its source bytes end, relative to the image (wraps below zero for history)
    uint32_t dist;
    uint32_t sbase;  // LDS byte address of the first source byte: in the ring, or in the far buffer
    bool simple;     // at most 16 bytes, no overlap with itself, neither end wraps around the ring
    bool far;        // simple, and every source byte is older than the ring
};
// 16 bytes per lane from the slot, for sources older than the ring: requested when a batch is planned,
// parked in LDS (the stage is idle while matches are resolved) at the end of the iteration that planned it.
struct LzFar {
    uint32_t f0, f1, f2, f3;
};

// What does not depend on resolved bytes: the matches of batch `mb`, their descriptors, the request for
// sources older than the ring (one unaligned 16-B load per match: what lies behind a match's source is
// older output of this stream, i.e. readable, as long as the slot is at least 16 bytes long).
__device__ __forceinline__ void lz_batch_plan(LzLds& L, LzBatch& B, LzFar& Fr, const uint32_t mb, const uint32_t nmatch, const uint32_t O,
                                              const uint32_t o_ri, const int32_t ring_lo, const uint8_t* gout, const int lane) {
    const uint32_t m = mb + (uint32_t)lane;
    const bool valid = m < nmatch;
    const uint32_t pos = valid ? (uint32_t)L.u.w.idx[min(m, kLzIdxCap - 1)] : 0u;
    const uint32_t qi0 = lz_wrap(o_ri + pos);
    const uint32_t dsc = (uint32_t)L.ring[qi0] | ((uint32_t)L.ring[qi0 + 1] << 8) | ((uint32_t)L.ring[qi0 + 2] << 16);
    const uint32_t len = (dsc & 0xFF) + 3, dist = (dsc >> 8) + 1;
    const uint32_t sidx0 = lz_back(qi0, dist);
    B.qi0 = qi0;
    B.sidx0 = sidx0;
    B.pos = pos;
    B.len = valid ? len : 0u;
    B.dist = dist;
    B.send = pos - dist + min(len, dist);
    const int32_t src = (int32_t)(O + pos - dist);  // >= 0 (pass 2 checked it)
    // (neither end may wrap around the ring: the BYTES of the match -- what a lane reads beyond them lies in the guard
    //  bytes and is not used.  Round 4 asked for 16 bytes either side: one match in 110 went to the whole wavefront
    //  for that alone, 100 of them per stream at ~1 200 cycles each.)
    const bool small = valid && len <= 16 && dist >= len && qi0 + len <= kLzRing;
    const bool far = small && src + (int32_t)len <= ring_lo;  // (its ring index means nothing: the distance may exceed the ring)
    B.far = far;
    B.simple = far || (small && src >= ring_lo && sidx0 + len <= kLzRing);
    const uint32_t ring_at = (uint32_t)(reinterpret_cast<uintptr_t>(&L.ring[0]) & 0xFFFFu);
    const uint32_t far_at = (uint32_t)(reinterpret_cast<uintptr_t>(&L.u.w.stage[0]) & 0xFFFFu);
    B.sbase = far ? far_at + 16u * (uint32_t)lane : ring_at + sidx0;
    // (always issued, so that the number of loads in flight is known: a lane without a far source reads the slot's start)
    const uint32_t* gp = reinterpret_cast<const uint32_t*>(gout + (far ? (uint32_t)src : 0u));
    Fr.f0 = __hip_atomic_load(gp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Fr.f1 = __hip_atomic_load(gp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Fr.f2 = __hip_atomic_load(gp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Fr.f3 = __hip_atomic_load(gp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The whole wavefront copies one match (any length, any distance, any place in the ring): byte k is
// out[start - dist + k mod dist]; source bytes older than the ring come from the slot.
__device__ __forceinline__ void lz_coop_copy(LzLds& L, const uint32_t qi0, const uint32_t len, const uint32_t dist, const uint32_t src0,
                                             const int32_t ring_lo, const uint8_t* gout, const int lane) {
    const float inv = 1.0f / (float)dist;
    bool waited = false;
    for (uint32_t k0 = 0; k0 < len; k0 += kWave) {
        const uint32_t k = k0 + (uint32_t)lane;
        uint32_t r = k;
        if (dist < len) {  // k mod dist for k < 258 (the quotient estimate is off by at most one)
            const uint32_t qq = (uint32_t)((float)k * inv);
            r = k - qq * dist;
            r = (int32_t)r < 0 ? r + dist : r;
            r = r >= dist ? r - dist : r;
        }
        if (k < len) {
            const int32_t sp = (int32_t)(src0 + r);
            uint32_t v;
            if (sp < ring_lo) {
                if (!waited) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    waited = true;
                }
                v = __hip_atomic_load(gout + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                uint32_t si = qi0 + kLzRing - dist + r;  // ring index of the source byte
                si = si >= kLzRing ? si - kLzRing : si;
                si = si >= kLzRing ? si - kLzRing : si;
                v = L.ring[si];
            }
            uint32_t qi = qi0 + k;
            qi = qi >= kLzRing ? qi - kLzRing : qi;
            L.ring[qi] = (uint8_t)v;
        }
        wave_sync();
    }
}

// Copies the matches of one planned batch.  Matches whose source bytes end in front of the batch's first
// match depend on nothing that is still missing (literals are in place, the batches before are done):
// all of them at once, a lane each.  The others in stream order, against a frontier: a match is ready
// when its source bytes end at or below the start of the first match not copied yet.
__device__ __forceinline__ void lz_batch_fill(LzLds& L, const LzBatch& B, const uint32_t O, const int32_t ring_lo,
                                              const uint8_t* gout, const int lane, uint32_t& rounds) {
    const uint32_t F = __builtin_amdgcn_readfirstlane(B.pos);  // (lane 0 always holds a match)
    const uint8_t __attribute__((address_space(3)))* const sp = (const uint8_t __attribute__((address_space(3)))*)(uintptr_t)B.sbase;
    auto copy = [&](const bool go) __attribute__((always_inline)) {
        // go: simple matches only.  Reads first, then writes (a round's sources are never its destinations);
        // four bytes at a time, as far as the longest match of the round needs
        const uint32_t longest = __any(go && B.len > 12) ? 16u : __any(go && B.len > 8) ? 12u : __any(go && B.len > 4) ? 8u : 4u;
        // (every lane stores every byte: a byte that is not wanted goes to a spare byte behind the ring -- a
        //  select per byte instead of an execution mask per byte)
        const uint32_t n = go ? B.len : 0u;
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += 4) {
            if ((uint32_t)k0 >= longest) break;
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = sp[k0 + k];
#pragma unroll
            for (int k = 0; k < 4; k++) L.ring[(uint32_t)(k0 + k) < n ? B.qi0 + k0 + k : kLzRing + 12] = (uint8_t)v[k];
        }
    };
    const bool indep = B.simple && (B.far || (int32_t)B.send <= (int32_t)F);
    copy(indep);
    uint64_t todo = __ballot(B.len != 0 && !indep);
    while (todo) {
        rounds++;
        wave_sync();
        const int f = __ffsll((unsigned long long)todo) - 1;
        const uint32_t fpos = __builtin_amdgcn_readlane(B.pos, f);
        if (!__builtin_amdgcn_readlane(B.simple ? 1u : 0u, f)) {
            const uint32_t flen = __builtin_amdgcn_readlane(B.len, f), fdist = __builtin_amdgcn_readlane(B.dist, f);
            lz_coop_copy(L, __builtin_amdgcn_readlane(B.qi0, f), flen, fdist, O + fpos - fdist, ring_lo, gout, lane);
            todo &= todo - 1;
            continue;
        }
        const bool waiting = (todo >> lane) & 1;
        const bool ready = waiting && B.simple && (lane == f || (int32_t)B.send <= (int32_t)fpos);
        copy(ready);
        todo &= ~__ballot(ready);
    }
    wave_sync();
}

enum : uint32_t { LZ_MORE = 0, LZ_EOB = 1, LZ_BAIL = 2, LZ_SHRINK = 3, LZ_DISTRUST = 4 };

struct LzIn {
    const uint8_t* base16;  // 16-B aligned address at or below the stream's first byte
    uint32_t mis;           // the stream's first byte = base16 + mis
    uint64_t win_bytes;     // mis + length of the stream
    const uint8_t* buf_lo;  // readable range of the packed batch
    const uint8_t* buf_hi;
    uint32_t in_bits;
    uint32_t cap;           // capacity of the output slot
    uint2* ck;              // this wavefront's checkpoints: 64 x kLzMaxPhases items
};

// One SUPER-SPAN of the current block from stream bit `bitpos`: the next 64 x P x Q stream bits, lane l owning the
// P x Q bits from bitpos + l P Q and walking them in P phases of Q bits (Q = kLzRange unless little input is left).
//
//   pass 1   a lane walks a guessed chain from up to kLzWarm bits in front of its range to its range (sliding over
//            impossible tokens), then phase by phase through it, leaving an ITEM per phase in global scratch:
//            where the phase's first step starts, the output bytes and matches of the steps that start in it.
//   check    a lane's first step must be where its left neighbour's chain left the neighbour's range; lanes that
//            fail walk again from there until they meet their old chain at a phase boundary (the items behind
//            that boundary stand).  Once per super-span -- not once per image as in the first version of this
//            kernel, whose warm-up and re-walks were 46 % of its time.
//   items    lane-major order is stream order; 64 items at a time (as many as fit the image): offsets by prefix
//            sums, pass 2 (every lane walks one item), resolve, flush.
//
// LZ_MORE / LZ_EOB: `bitpos` advanced (LZ_EOB: behind the end-of-block code of `eob_bits` bits), everything decoded
// is resolved, flushed and part of the history.
// LZ_SHRINK: the item at `bitpos` (advanced to it) does not fit an image: again with a smaller Q.
// LZ_BAIL: this stream is for the exact kernels -- from `bitpos` on: what lies in front of it (possibly advanced: the
// lanes in front of a bad token or of the end of the input, the images that fit the slot) is decoded, resolved and
// flushed like the rest, so the kernels behind can take the stream up there (ResumePoint, inflate_stream.h).
// LZ_DISTRUST: the two passes disagree (a bug): nothing of this stream is to be used.
__device__ __forceinline__ uint32_t lz_superspan(LzLds& L, LzOut& o, const LzBounds& bd, const LzIn& in, uint32_t& bitpos,
                                                 uint32_t& eob_bits, const uint32_t qcap, const int lane) {
    const uint32_t mis8 = in.mis * 8;
    const uint32_t w0 = bitpos + mis8, limit = in.in_bits + mis8;  // window bits
    const uint32_t fair = (limit - w0 + kWave - 1) / kWave;
    const uint32_t Q = max(8u, min(min(fair, qcap), kLzRange));
    const uint32_t P = max(1u, min((uint32_t)kLzMaxPhases, (fair + Q - 1) / Q));
    const uint32_t R = Q * P;
    const uint32_t s = w0 + (uint32_t)lane * R;
    const bool live = s < limit;
    uint2* const ck = in.ck + (uint32_t)lane * P;  // this lane's items
    bool trouble = false;
    uint32_t iters = 0, slows = 0;
    auto stage = [&](const bool act, const uint32_t pos) __attribute__((always_inline)) -> uint32_t {
        wave_sync();
        const uint32_t dw = lz_stage_slot(L, act, pos, in.base16, in.win_bytes, in.buf_lo, in.buf_hi, lane);
        wave_sync();
        return dw;
    };
    // ---- pass 1: warm-up, then the phases ----
    uint32_t pos, stop = 0, stop_bits = 0, b;
    {
        const uint32_t ws = (s - w0 > kLzWarm) ? s - kLzWarm : w0;
        const bool act = live && ws < s;
        const uint32_t dw = stage(act, ws);
        const LzWalk w = lz_walk<false>(L, bd, ws, s, ws != w0, act, limit, dw, 0, 0, 0, 0, trouble, &iters, &slows);
        pos = act ? w.e : s;
        if (act && w.stop) {  // a real chain that stops in front of this range: the block ends in a lane to the left
            stop = w.stop;
            stop_bits = w.stop_bits;
        }
        if (!live) stop = 2;
        b = pos;
    }
    LZT(o, 1);
    for (uint32_t p = 0; p < P; p++) {
        const uint32_t end = s + Q * (p + 1);
        const bool act = live && stop == 0 && pos < end;
        const uint32_t dw = stage(act, pos);
        const LzWalk w = lz_walk<false>(L, bd, pos, end, false, act, limit, dw, 0, 0, 0, 0, trouble, &iters, &slows);
        ck[p] = make_uint2(pos, act ? (w.cnt | (w.nm << 16)) : 0u);
        if (act) {
            pos = w.e;
            stop = w.stop;
            stop_bits = w.stop_bits;
        }
    }
    uint32_t e = pos;
    LZT(o, 2);
    // ---- check, fix-up rounds ----
    int first_stop = kWave;
    bool converged = false;
    for (int round = 0; round <= kWave; round++) {
        const uint32_t prev_e = __shfl_up(e, 1, kWave), prev_stop = __shfl_up(stop, 1, kWave);
        const bool ok = lane == 0 || (prev_stop == 0 && b == prev_e);
        const uint64_t bad_mask = __ballot(!ok);
        const int first_bad = bad_mask ? __ffsll((unsigned long long)bad_mask) - 1 : kWave;
        const uint64_t stop_mask = __ballot(stop != 0) & (first_bad < kWave ? lanemask_lt(first_bad) : ~0ull);
        first_stop = stop_mask ? __ffsll((unsigned long long)stop_mask) - 1 : kWave;
        if (first_stop < kWave || first_bad == kWave) {
            converged = true;
            break;
        }
        const bool redo = !ok && prev_stop == 0 && live;
        LZC(o, 10, 1);
        // the lane walks again from where its neighbour's chain arrived, phase by phase, until it stands on the
        // first step of one of its old items (from there on the old chain is the real one) or leaves its range
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the items are read back below
        uint32_t np = redo ? prev_e : 0u, nstop = 0, nstop_bits = 0;
        bool walking = redo;
        if (redo) b = prev_e;
        for (uint32_t p = 0; p < P; p++) {
            const uint32_t end = s + Q * (p + 1);
            if (!__any(walking)) break;
            uint32_t old_pos = 0;
            if (walking && p != 0) old_pos = __hip_atomic_load(&ck[p].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (walking && p != 0 && old_pos == np) walking = false;  // met the old chain: the items from p on stand
            const bool act = walking && np < end;
            const uint32_t dw = stage(act, np);
            const LzWalk w = lz_walk<false>(L, bd, np, end, false, act, limit, dw, 0, 0, 0, 0, trouble, &iters, &slows);
            if (walking) {
                ck[p] = make_uint2(np, act ? (w.cnt | (w.nm << 16)) : 0u);
                if (act) {
                    np = w.e;
                    if (w.stop) {
                        nstop = w.stop;
                        nstop_bits = w.stop_bits;
                        // (the items behind a stop are never looked at; make them empty all the same)
                        for (uint32_t q = p + 1; q < P; q++) ck[q] = make_uint2(np, 0u);
                        walking = false;
                        e = np;
                        stop = nstop;
                        stop_bits = nstop_bits;
                    }
                }
            }
        }
        if (walking) {  // never met the old chain: a new end
            e = np;
            stop = 0;
            stop_bits = 0;
        }
    }
    LZT(o, 3);
    LZC(o, 9, 1);
    LZC(o, 21, iters);
    LZC(o, 23, slows);
    if (!converged || __any(trouble)) return LZ_BAIL;
    const uint32_t fs_kind = first_stop < kWave ? __builtin_amdgcn_readlane(stop, first_stop) : 0u;
    // a bad token on the real chain, or the end of the input: the lanes in front of that one are decoded all the same
    const bool bad_end = fs_kind == 2;
    const int nvalid = first_stop < kWave ? (bad_end ? first_stop : first_stop + 1) : kWave;
    if (nvalid == 0) return LZ_BAIL;
    const uint32_t end_pos = __builtin_amdgcn_readlane(e, nvalid - 1);
    const uint32_t end_bits = fs_kind == 1 ? __builtin_amdgcn_readlane(stop_bits, nvalid - 1) : 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- the items in stream order, an image at a time ----
    const uint32_t T = (uint32_t)nvalid * P;
    const uint32_t recipP = 65536u / P + 1;  // t / P for t < 1024
    uint32_t t0 = 0;
    auto load_item = [&](uint32_t t) __attribute__((always_inline)) -> unsigned long long {
        // (always issued: the next image's items are requested while this image is decoded)
        return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(in.ck + min(t, T - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    unsigned long long raw_next = load_item((uint32_t)lane);
    while (t0 < T) {
        const uint32_t t = t0 + (uint32_t)lane;
        const bool have = t < T;
        const unsigned long long raw = raw_next;
        const uint2 item = have ? make_uint2((uint32_t)raw, (uint32_t)(raw >> 32)) : make_uint2(0, 0);
        const uint32_t cnt = item.y & 0xFFFF, nm = item.y >> 16;
        const uint32_t incl = wave_scan_add(cnt), incl_m = wave_scan_add(nm);
        const int nuse = __popcll(__ballot(have && incl <= kLzImgCap && incl_m <= kLzIdxCap));
        if (nuse == 0) {  // the first item alone is too much for an image: again from there, with shorter phases
            bitpos = __builtin_amdgcn_readfirstlane(item.x) - mis8;
            return LZ_SHRINK;
        }
        const uint32_t N = __builtin_amdgcn_readlane(incl, nuse - 1), nmatch = __builtin_amdgcn_readlane(incl_m, nuse - 1);
        if (N > in.cap - o.O) {  // OutputTooLarge is the exact kernels' business: from this image on
            bitpos = __builtin_amdgcn_readfirstlane(item.x) - mis8;
            return LZ_BAIL;
        }
        raw_next = load_item(t + (uint32_t)nuse);
        LZT(o, 4);
        // ---- pass 2 ----
        const bool mine = lane < nuse && cnt != 0;
        {
            uint32_t owner = (t * recipP) >> 16;
            owner = owner * P > t ? owner - 1 : owner;
            owner = (owner + 1) * P <= t ? owner + 1 : owner;
            const uint32_t phase = t - owner * P;
            const uint32_t end = w0 + owner * R + Q * (phase + 1);
            const uint32_t dw = stage(mine, item.x);
            uint32_t it2 = 0, sl2 = 0;
            const LzWalk w2 = lz_walk<true>(L, bd, item.x, end, false, mine, limit, dw, incl - cnt, o.o_ri, o.O, incl_m - nm, trouble, &it2, &sl2);
            LZC(o, 20, it2);
            LZC(o, 22, sl2);
            if (__any(mine && (w2.cnt != cnt || w2.nm != nm))) return LZ_DISTRUST;  // the passes disagree: a bug, never publish
            if (__any(trouble)) {  // a distance beyond the start of the output: the exact kernels report it, from this image on
                bitpos = __builtin_amdgcn_readfirstlane(item.x) - mis8;
                return LZ_BAIL;
            }
        }
        wave_sync();
        LZT(o, 5);
    // ---- resolve the matches: 64 at a time in stream order, planned a batch ahead ----
    if (nmatch != 0) {
        const uint32_t O = o.O;
        const int32_t ring_lo = (int32_t)(O + N) - (int32_t)kLzRing;  // oldest position still in the ring
        const uint8_t* const gout = o.out_al + o.gmis;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // our own flush stores: far sources are read back
        LzBatch A, B;
        LzFar Fr;
        uint4* const farbuf = reinterpret_cast<uint4*>(&L.u.w.stage[0]);  // (the stage is idle until the next span)
        lz_batch_plan(L, A, Fr, 0, nmatch, O, o.o_ri, ring_lo, gout, lane);
        farbuf[lane] = make_uint4(Fr.f0, Fr.f1, Fr.f2, Fr.f3);
        uint32_t rounds = 0;
        for (uint32_t mb = 0; mb < nmatch; mb += kWave) {
            const bool more = mb + kWave < nmatch;
            if (more) lz_batch_plan(L, B, Fr, mb + kWave, nmatch, O, o.o_ri, ring_lo, gout, lane);
            wave_sync();
            lz_batch_fill(L, A, O, ring_lo, gout, lane, rounds);
            if (more) {  // (the loads of the next batch have had the whole fill to arrive)
                farbuf[lane] = make_uint4(Fr.f0, Fr.f1, Fr.f2, Fr.f3);
                A = B;
            }
            LZC(o, 11, 1);
        }
        LZC(o, 12, rounds);
    }
        wave_sync();
        LZT(o, 6);
        // ---- the image becomes history ----
        o.O += N;
        o.o_ri = lz_wrap(o.o_ri + N);
        lz_flush(L, o, false, lane);
        LZT(o, 7);
        t0 += (uint32_t)nuse;
    }
    bitpos = end_pos + end_bits - mis8;
    eob_bits = end_bits;
    return bad_end ? LZ_BAIL : (fs_kind == 1 ? LZ_EOB : LZ_MORE);
}

}  // namespace fdh
