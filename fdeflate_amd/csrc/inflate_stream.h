// inflate_stream.h -- one zlib stream per wavefront.
//
//   * bit reader over an LDS-staged input window (coalesced 16 B/lane loads of the packed batch)
//   * LDS output ring with coalesced 16-B line flushes and a fused Adler-32
//   * the zlib/deflate state machine with the reference's one-shot semantics
//   * two decoders for compressed-block data:
//       serial_token()  one symbol at a time, all lanes uniform: the exact restatement of the
//                       reference's careful loop (every error / truncation rule)
//       tile_step()     64 lanes x 64 bits speculative parallel decode: every lane decodes the
//                       code chain of its own 8-byte chunk from a guessed start, chains are
//                       re-synchronised lane to lane (Huffman codes self-synchronise), output
//                       offsets come from a wave prefix sum, literals are scattered into the
//                       output ring in parallel and matches are replayed in order.
//
// Restates (behaviour, not code) Decompressor::read and its helpers:
//   reference src/decompress.rs:179-337 (state machine), :344-438 (block header),
//   :440-477 (code-length codes), :479-555 (code lengths), :611-1018 (compressed data),
//   :271-305 (stored data), :306-326 (checksum), :1111-1144 (one-shot wrapper).
//
// One-shot semantics.  The reference's `read` never fails on truncated input or a full output
// buffer; it returns and the wrapper classifies (src/decompress.rs:1126-1139): done -> Ok,
// output full -> OutputTooLarge, otherwise InsufficientInput.  Every place the reference
// "waits for more bits" (`nbits < X`, all X <= 48 < 56 so `nbits` there equals the number of
// unread stream bits) is RC_STUCK here with the same threshold on `left`.
#pragma once
#include "inflate_tables.h"
#include "inflate_segments.h"  // SegReader (per-lane bit reader over an LDS ring)

namespace fdh {

constexpr int kInChunk = 1024;              // bytes per coalesced input load (64 lanes x 16 B)
constexpr int kInRingDw = 2 * kInChunk / 4; // two chunks
constexpr int kOutRing = 2048;              // bytes, power of two (4096: one workgroup fewer per CU, 8 % slower on zlib-6 streams; 1024: too small for a tile)
constexpr int kOutMask = kOutRing - 1;
constexpr int kFlushSlack = 64;
constexpr int kMaxMatches = 192;            // match tokens replayed per tile (8 x WaveIo + tables <= 80 KiB in the canon kernel)
constexpr int kTileBits = 64;               // stream bits owned by one lane of a tile
constexpr uint32_t kSpanSegBits = 1024;     // stream bits owned by one lane of a span
constexpr uint32_t kSpanMinSeg = 512;       // shorter segments: not worth a span, tiles take over
constexpr uint32_t kSpanMaxMatches = 8192;  // match tokens per span (scratch list capacity)

// Decode tables of the current block.
template <int LB>
struct __attribute__((aligned(16))) TableSetT {
    uint32_t lit[1 << LB];
    uint32_t dist[kDistSize];
    CodeBook lit_cb;
    CodeBook dist_cb;
    uint16_t lit_sorted[288];
    uint16_t dist_sorted[32];
    uint32_t eof[4];  // code, mask, bits of the end-of-block symbol (reference eof_code/mask/bits)
};
using TableSet = TableSetT<kLitBits>;

// Per-wavefront staging.
struct __attribute__((aligned(16))) WaveIo {
    uint32_t in_ring[kInRingDw];
    uint8_t out_ring[kOutRing];
    uint32_t mlist[2 * kMaxMatches];
};

// Scratch for dynamic block headers.
struct __attribute__((aligned(16))) HeaderScratch {
    uint32_t cl[kClSize];
    CodeBook cl_cb;
    uint16_t cl_sorted[32];
    uint8_t lens[320 + 16];
};

struct StreamArgs {
    const uint8_t* in;        // stream bytes (global)
    uint64_t in_len;
    uint8_t* out;             // output slot (global)
    uint32_t cap;             // slot capacity
    const uint8_t* buf_lo;    // readable range of the whole packed input buffer
    const uint8_t* buf_hi;
    uint32_t flags;           // FDH_FLAG_*
};

struct StreamResult {
    uint32_t status, out_len, adler;
    bool ambiguous;  // ended "stuck" close to the end of the input (see inflate.hip)
};

// A place a decoder can start from instead of the stream's first byte: a symbol boundary inside a Huffman block
// (`bit` > `hdr_bit`) or the header of a block (`bit` == `hdr_bit`).  Everything in front of it is decoded: the
// bytes [0, opos) are in the output slot in global memory, `adler` is their Adler-32.  The block's tables are
// rebuilt from its header, which is why that is part of the point.
struct ResumePoint {
    uint64_t hdr_bit;  // stream bit of the block header
    uint64_t bit;      // stream bit to go on from
    uint32_t opos;
    uint32_t adler;
    uint32_t valid;
    uint32_t step;     // what `bit` is to the reference's chain of table steps (STEP_*): resync_to_step_start
};
// A symbol boundary inside a block, seen from the reference's chain of table steps (one symbol, or two literals
// whose codes fit the table index together): the start of a step, the second literal of a pair, or not known.
enum : uint32_t { STEP_UNKNOWN = 0, STEP_START = 1, STEP_SECOND = 2 };

enum : uint32_t { RC_OK = 0, RC_EOB = 0x100, RC_STUCK = 0x101, RC_REDO = 0x102 };  // anything else: a StreamStatus

#ifdef FDH_DEBUG_TILES
__device__ uint32_t g_dbg[1 << 16];
__device__ uint32_t g_dbg_n;
__device__ unsigned long long g_gstat[24];
// statistics are collected per stream (member gacc) and added to the global counters once, at its
// end: an atomic per phase and tile, contended by every wavefront, used to cost more than the phases
#define GSTAT(k, v) do { gacc[k] += (unsigned long long)(v); } while (0)
#else
#define GSTAT(k, v) do { } while (0)
#endif

// Stores output [flushed, target) to global memory and folds it into the Adler-32.  target is
// opos rounded down to a 16-B line unless `final`.  Ring index of output position p is
// (p + gmis) & kOutMask, so 16-B lines of the global slot are 16-B lines of the ring.
struct FlushState {
    uint32_t flushed, adler_a, adler_b;
};
__device__ __forceinline__ FlushState flush_ring(WaveIo* iop, uint8_t* out_al, uint32_t gmis,
                                                           uint32_t opos, uint32_t flushed, uint32_t adler_a,
                                                           uint32_t adler_b, bool final, int lane) {
    WaveIo& io = *iop;
    wave_sync();
    uint32_t q_lo = flushed + gmis;
    uint32_t q_hi = opos + gmis;
    if (!final) q_hi &= ~15u;
    if (q_hi <= q_lo) return FlushState{flushed, adler_a, adler_b};
    for (uint32_t it = q_lo & ~15u; it < q_hi; it += kWave * 16) {
        uint32_t lq = it + lane * 16;  // this lane's line, q-space
        uint32_t blk_hi = min(q_hi, it + kWave * 16);
        uint32_t blk_lo = max(q_lo, it);
        uint32_t s = 0, t = 0;
        if (lq < blk_hi && lq + 16 > blk_lo) {
            uint32_t lo = (blk_lo > lq) ? blk_lo - lq : 0;
            uint32_t hi = (blk_hi < lq + 16) ? blk_hi - lq : 16;
            uint32_t W = blk_hi - lq;  // weight of byte j is W - j
            if (lo == 0 && hi == 16) {
                uint4 v = *reinterpret_cast<const uint4*>(&io.out_ring[lq & kOutMask]);
                *reinterpret_cast<uint4*>(out_al + lq) = v;
                s = bytesum4(v.x) + bytesum4(v.y) + bytesum4(v.z) + bytesum4(v.w);
                uint32_t u = bytedot4(v.x, 0x03020100u, 0);
                u = bytedot4(v.y, 0x07060504u, u);
                u = bytedot4(v.z, 0x0b0a0908u, u);
                u = bytedot4(v.w, 0x0f0e0d0cu, u);
                t = W * s - u;
            } else {
                for (uint32_t j = lo; j < hi; j++) {
                    uint32_t b = io.out_ring[(lq + j) & kOutMask];
                    out_al[lq + j] = (uint8_t)b;
                    s += b;
                    t += (W - j) * b;
                }
            }
        }
        uint32_t S = wave_sum_u32(s);
        uint32_t Tt = wave_sum_u32(t);
        uint32_t Lb = blk_hi - blk_lo;
        adler_b = (uint32_t)(((uint64_t)adler_b + (uint64_t)Lb * adler_a + Tt) % kAdlerMod);
        adler_a = (adler_a + S) % kAdlerMod;
    }
    flushed = q_hi - gmis;
    wave_sync();
    return FlushState{flushed, adler_a, adler_b};
}

// ---------------------------------------------------------------------------------------
// LB = index bits of the literal/length table.  12 reproduces the reference's tables (and its
// double-literal pairing, which the exact serial decoder relies on); the fast general kernel uses
// a smaller table for occupancy and resolves longer codes by the canonical walk.
template <int LB, bool SPANS = true>
struct InflaterT {
    static constexpr int kLB = LB;
    static constexpr uint32_t kLSize = 1u << LB;
    using LitTraits = LitlenTraitsT<LB>;
    TableSetT<LB>& T;
    WaveIo& io;
    HeaderScratch* hs;
    const int lane;
    // ---- input window / bit reader (all uniform) ----
    const uint8_t* in;
    const uint8_t* base16;  // 16-B aligned address at or below the first stream byte
    uint32_t mis;           // first stream byte = base16 + mis
    uint64_t win_bytes;     // mis + in_len
    const uint8_t* buf_lo;
    const uint8_t* buf_hi;
    uint32_t loaded;        // chunks [loaded-2, loaded) are resident in the ring
    uint32_t next_dw;       // next window dword to append to `bb`
    uint64_t bb;            // bit buffer, LSB first; bits beyond the stream end are zero
    uint32_t bbn;           // valid bits in bb
    uint64_t left;          // unread stream bits (the reference's "nbits" once clamped)
    // ---- output ----
    uint8_t* out_al;        // out - gmis (16-B aligned)
    uint32_t gmis;          // out & 15
    uint32_t cap, opos, flushed;
    uint32_t adler_a, adler_b;
    // ---- block state ----
    uint32_t eof_code, eof_mask, eof_bits;
    bool fixed_built;
    bool last_block;
    uint32_t flags;
    uint32_t serial_credit;  // tokens to decode serially before the next tile attempt
    uint32_t* span_list;     // scratch of this workgroup: kSpanMaxMatches x {at, length | dist << 16}; null: no spans
    uint32_t span_credit;    // tiles to run before the next span attempt (after a span that did not pay)
    // ---- check points (inflate_general_kernel: the exact serial decoder then only re-derives the tail) ----
    bool keep_ck;            // take a check point at every block header and in front of every tile
    uint64_t hdr_bit;        // where the current block's header starts
    ResumePoint ck;          // the last check point
    uint32_t step_state;     // STEP_*: the current position in the reference's chain of table steps (kept while keep_ck)
    bool last_was_pair;      // serial_token: the table entry it took was a pair of literals
    uint64_t tok_bit;        // where the last serial token started
    bool stuck_at_step;      // the run ended because serial_token, called at the start of a step, could not go on
#ifdef FDH_DEBUG_TILES
    unsigned long long gacc[24] = {};
#endif

    __device__ __forceinline__ InflaterT(TableSetT<LB>& t, WaveIo& w, HeaderScratch* h, int ln)
        : T(t), io(w), hs(h), lane(ln), span_list(nullptr), keep_ck(false) {}

    // ------------------------------------------------------------------ input window
    __device__ __forceinline__ void load_chunk(uint32_t c) {
        uint64_t w0 = (uint64_t)c * kInChunk + (uint64_t)lane * 16;
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
            uint64_t rem = win_bytes - w0;  // bytes of this stream in the lane's 16
            if (rem < 16) {                 // zero what lies past the end of the stream
                uint32_t r = (uint32_t)rem;
                v.x = r >= 4 ? v.x : (r == 0 ? 0 : v.x & ((1u << (r * 8)) - 1));
                v.y = r >= 8 ? v.y : (r <= 4 ? 0 : v.y & ((1u << ((r - 4) * 8)) - 1));
                v.z = r >= 12 ? v.z : (r <= 8 ? 0 : v.z & ((1u << ((r - 8) * 8)) - 1));
                v.w = r <= 12 ? 0 : v.w & ((1u << ((r - 12) * 8)) - 1);
            }
        }
        *reinterpret_cast<uint4*>(&io.in_ring[(c & 1) * (kInChunk / 4) + lane * 4]) = v;
    }

    // Makes the chunk holding window dword `dw` resident.  Loading chunk c evicts chunk c-2; the
    // reader never looks back more than the 64 bits held in `bb`, and a tile names both ends of
    // its window, so nothing that is still needed is ever evicted.
    __device__ __forceinline__ void ensure_dw(uint32_t dw) {
        uint32_t c = dw / (kInChunk / 4);
        if (c >= loaded) {
            wave_sync();
            while (loaded <= c) {
                if (loaded + 1 < c) loaded = c - 1;  // after a seek: skip chunks nobody will read
                load_chunk(loaded);
                loaded++;
            }
            wave_sync();
        }
    }

    __device__ __forceinline__ void seek(uint64_t bitpos) {  // bitpos relative to the first stream byte
        uint64_t wbit = bitpos + (uint64_t)mis * 8;
        next_dw = (uint32_t)(wbit >> 5);
        uint32_t c = next_dw / (kInChunk / 4);
        if (!(c + 2 == loaded || c + 1 == loaded)) loaded = c;  // ring content unusable
        ensure_dw(next_dw);
        uint32_t sh = (uint32_t)wbit & 31;
        bb = (uint64_t)(uni(io.in_ring[next_dw & (kInRingDw - 1)]) >> sh);
        bbn = 32 - sh;
        next_dw++;
    }

    __device__ __forceinline__ void refill() {  // afterwards bbn >= 33
        if (bbn <= 32) {
            ensure_dw(next_dw);
            uint32_t w = uni(io.in_ring[next_dw & (kInRingDw - 1)]);
            bb |= (uint64_t)w << bbn;
            bbn += 32;
            next_dw++;
        }
    }
    __device__ __forceinline__ void consume(uint32_t n) {
        bb >>= n;
        bbn -= n;
        left -= n;
    }
    __device__ __forceinline__ uint64_t consumed_bits() const { return (win_bytes - mis) * 8 - left; }

    // ------------------------------------------------------------------ output ring
    __device__ __forceinline__ void put_byte(uint32_t b) {
        if (lane == 0) io.out_ring[(opos + gmis) & kOutMask] = (uint8_t)b;
        opos++;
    }
    __device__ __forceinline__ void flush(bool final) {
        FlushState f = flush_ring(&io, out_al, gmis, opos, flushed, adler_a, adler_b, final, lane);
        flushed = f.flushed;
        adler_a = f.adler_a;
        adler_b = f.adler_b;
    }
    __device__ __forceinline__ void make_room(uint32_t n) {
        if (opos + n - flushed > (uint32_t)(kOutRing - kFlushSlack)) flush(false);
    }

    // LZ77 copy of n bytes to output position `at` from distance d (src/decompress.rs:792-829):
    // out[at+k] = out[at - d + (k mod d)], every source lies before `at`.  `ring_top` is the
    // highest output position that is (or is about to be) resident in the ring; sources older
    // than ring_top - kOutRing come from global memory (they are < flushed by make_room).
    __device__ __forceinline__ void copy_bytes(uint32_t at, uint32_t n, uint32_t d, uint32_t ring_top) {
        int64_t ring_lo = (int64_t)ring_top - kOutRing;
        bool need_global = (int64_t)at - (int64_t)d < ring_lo;
        if (need_global) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // our own flush stores
        // n and d are uniform: the (slow) remainder is only computed for overlapping copies
        const bool overlap = d < n;
        const float inv_d = overlap ? 1.0f / (float)d : 0.0f;
        for (uint32_t k = lane; k < n; k += kWave) {
            uint32_t r = k;
            if (overlap) {  // k % d for k < 258 (exact: quotient estimate off by at most one)
                uint32_t q = (uint32_t)((float)k * inv_d);
                r = k - q * d;
                if ((int32_t)r < 0) r += d;
                if (r >= d) r -= d;
            }
            uint32_t src = at - d + r;
            uint32_t b;
            if ((int64_t)src >= ring_lo) {
                b = io.out_ring[(src + gmis) & kOutMask];
            } else {
                // L1-bypassing load: the line may have been cached before our later stores
                b = __hip_atomic_load(out_al + gmis + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            io.out_ring[(at + k + gmis) & kOutMask] = (uint8_t)b;
        }
    }
    __device__ __forceinline__ void copy_match(uint32_t n, uint32_t d) {  // caller made room
        wave_sync();
        copy_bytes(opos, n, d, opos + n);
        opos += n;
        wave_sync();
    }

    // Stored-block payload: n bytes from stream byte offset `src_byte` (src/decompress.rs:271-305).
    __device__ __forceinline__ void copy_stored(uint64_t src_byte, uint32_t n) {
        uint32_t done = 0;
        while (done < n) {
            uint32_t m = min(n - done, 1024u);
            make_room(m);
            wave_sync();
            for (uint32_t k = lane; k < m; k += kWave) {
                io.out_ring[(opos + k + gmis) & kOutMask] = in[src_byte + done + k];
            }
            opos += m;
            done += m;
            wave_sync();
        }
    }

    // ------------------------------------------------------------------ tables
    // CompressedBlock::build_tables, src/decompress.rs:561-606.  lens[0..320) in hs->lens.
    __device__ __forceinline__ uint32_t build_block_tables(uint32_t hlit) {
        const uint8_t* lens = hs->lens;
        if (uni(lens[256]) == 0) return ST_BAD_LITERAL_LENGTH_HUFFMAN_TREE;
        if (build_table<LitTraits, false>(T.lit, lens, (int)hlit, T.lit_cb, T.lit_sorted, lane) != BUILD_OK)
            return ST_BAD_CODE_LENGTH_HUFFMAN_TREE;  // sic, src/decompress.rs:579
        add_double_literals<LB>(T.lit, lane);
        // code of the end-of-block symbol, as the reference keeps it (eof_code/mask/bits)
        uint32_t l256 = uni(lens[256]);
        uint32_t rank = 0;
        for (int s = lane; s < 256; s += kWave) rank += (lens[s] == l256) ? 1u : 0u;
        rank = wave_sum_u32(rank);
        uint32_t cw = uni(T.lit_cb.first[l256]) + rank;
        eof_bits = l256;
        eof_mask = (1u << l256) - 1;
        eof_code = __brev(cw) >> (32 - l256);
        if (lane == 0) {
            T.eof[0] = eof_code;
            T.eof[1] = eof_mask;
            T.eof[2] = eof_bits;
        }
        if (build_table<DistTraits, true>(T.dist, lens + 288, 32, T.dist_cb, T.dist_sorted, lane) != BUILD_OK)
            return ST_BAD_DISTANCE_HUFFMAN_TREE;
        return ST_OK;
    }

    __device__ __forceinline__ void fill_fixed_lengths() {  // src/tables.rs:207-232
        for (int i = lane; i < 320; i += kWave) {
            uint8_t v = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5;
            hs->lens[i] = v;
        }
        wave_sync();
    }

    // ------------------------------------------------------------------ set-up
    __device__ __forceinline__ void init(const StreamArgs& a) {
        in = a.in;
        uintptr_t ia = reinterpret_cast<uintptr_t>(a.in);
        mis = (uint32_t)(ia & 15);
        base16 = a.in - mis;
        win_bytes = (uint64_t)mis + a.in_len;
        buf_lo = a.buf_lo;
        buf_hi = a.buf_hi;
        loaded = 0;
        left = a.in_len * 8;
        uintptr_t oa = reinterpret_cast<uintptr_t>(a.out);
        gmis = (uint32_t)(oa & 15);
        out_al = a.out - gmis;
        cap = a.cap;
        opos = 0;
        flushed = 0;
        adler_a = 1;
        adler_b = 0;
        fixed_built = false;
        last_block = false;
        eof_code = eof_mask = eof_bits = 0;
        flags = a.flags;
        serial_credit = 0;
        span_credit = 0;
        hdr_bit = 0;
        ck.valid = 0;
        step_state = STEP_UNKNOWN;
        last_was_pair = false;
        tok_bit = 0;
        stuck_at_step = false;
        seek(0);
    }

    // ------------------------------------------------------------------ check points
    // Everything decoded so far goes to the slot (partial line included: the next flush goes on from an odd
    // position, as the first one of a misaligned slot does) and the place is noted.  Only ever called at a
    // boundary between two table steps of the reference's chain: at a block header, in front of a tile.
    __device__ __forceinline__ void take_ck() {
        flush(true);
        ck.hdr_bit = hdr_bit;
        ck.bit = consumed_bits();
        ck.opos = opos;
        ck.adler = (adler_b << 16) | adler_a;
        ck.valid = 1;
        ck.step = ck.bit == ck.hdr_bit ? (uint32_t)STEP_START : step_state;
    }
    __device__ __forceinline__ void seek_to(uint64_t bit) {
        seek(bit);
        left = (win_bytes - mis) * 8 - bit;
    }
    // The last kOutRing bytes of the output (or all of it) from the slot into the ring: the history of a decoder
    // that starts at a resume point (copy_bytes reads a source from the ring when it is less than kOutRing bytes
    // in front of the END of the copy: all of them have to be there).
    __device__ __forceinline__ void reload_out_ring() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sync();
        const uint32_t lo_p = opos > (uint32_t)kOutRing ? opos - kOutRing : 0;
        const uint8_t* const g = out_al + gmis;
        for (uint32_t p = lo_p + (uint32_t)lane; p < opos; p += kWave)
            io.out_ring[(p + gmis) & kOutMask] = __hip_atomic_load(g + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wave_sync();
    }

    // ------------------------------------------------------------------ zlib header
    __device__ __forceinline__ uint32_t parse_zlib_header() {  // src/decompress.rs:226-244
        refill();
        if (left < 16) return RC_STUCK;
        uint32_t b0 = (uint32_t)bb & 0xFF, b1 = ((uint32_t)bb >> 8) & 0xFF;
        if ((b0 & 0x0F) != 0x08 || (b0 & 0xF0) > 0x70 || (b1 & 0x20) != 0 || ((b0 << 8) | b1) % 31 != 0)
            return ST_BAD_ZLIB_HEADER;
        consume(16);
        return RC_OK;
    }

    // ------------------------------------------------------------------ block header
    // src/decompress.rs:344-438 (+ :440-555 for dynamic blocks, :271-305 for stored payload).
    // RC_OK: a compressed block follows (tables ready).  RC_EOB: the block is already finished
    // (stored or empty fixed block).  RC_STUCK / status otherwise.
    __device__ __forceinline__ uint32_t parse_block_header() {
        refill();
        if (left < 10) return RC_STUCK;
        uint32_t hdr = (uint32_t)bb & 7;
        last_block = (hdr & 1) != 0;
        uint32_t type = hdr >> 1;
        if (type == 0) {
            uint32_t align = (uint32_t)((left - 3) & 7);
            if (left < 35 + align) return RC_STUCK;
            consume(3 + align);
            refill();
            uint32_t len = (uint32_t)bb & 0xFFFF, nlen = ((uint32_t)bb >> 16) & 0xFFFF;
            if (nlen != (~len & 0xFFFF)) return ST_INVALID_UNCOMPRESSED_BLOCK_LENGTH;
            consume(32);
            uint64_t avail = left >> 3;
            uint32_t n = len;
            if ((uint64_t)n > avail) n = (uint32_t)avail;
            if (n > cap - opos) n = cap - opos;
            uint64_t src_byte = consumed_bits() >> 3;
            copy_stored(src_byte, n);
            left -= (uint64_t)n * 8;
            if (n < len) return RC_STUCK;
            seek(consumed_bits());
            return RC_EOB;
        }
        if (type == 1) {
            consume(3);
            if (((uint32_t)bb & 0x7F) == 0) {  // empty fixed block, :377-394
                consume(7);
                return RC_EOB;
            }
            if (!fixed_built) {
                fill_fixed_lengths();
                build_block_tables(288);
                fixed_built = true;
            }
            return RC_OK;
        }
        if (type == 3) return ST_INVALID_BLOCK_TYPE;
        // ---- dynamic block ----
        if (left < 17) return RC_STUCK;
        uint32_t hlit = (((uint32_t)bb >> 3) & 31) + 257;
        uint32_t hdist = (((uint32_t)bb >> 8) & 31) + 1;
        uint32_t hclen = (((uint32_t)bb >> 13) & 15) + 4;
        if (hlit > 286) return ST_INVALID_HLIT;
        if (hdist > 30) return ST_INVALID_HDIST;
        consume(17);
        fixed_built = false;
        // code-length code lengths, src/decompress.rs:440-477 (scratch: lens[301..320))
        refill();
        if (left < 3 * hclen) return RC_STUCK;
        uint8_t* lens = hs->lens;
        if (lane < 19) lens[301 + lane] = 0;
        wave_sync();
        for (uint32_t i = 0; i < hclen; i++) {
            refill();
            if (lane == 0) lens[301 + kClclOrder[i]] = (uint8_t)((uint32_t)bb & 7);
            consume(3);
        }
        wave_sync();
        if (build_table<ClTraits, false>(hs->cl, lens + 301, 19, hs->cl_cb, hs->cl_sorted, lane) != BUILD_OK)
            return ST_BAD_CODE_LENGTH_HUFFMAN_TREE;
        // literal/length + distance code lengths, src/decompress.rs:479-555
        uint32_t total = hlit + hdist, nread = 0;
        while (nread < total) {
            refill();
            if (left < 7) return RC_STUCK;
            uint32_t e = uni(hs->cl[(uint32_t)bb & 127]);
            uint32_t nb = e & 15, sym = (e >> 8) & 0xFF;
            if (sym <= 15) {
                if (lane == 0) lens[nread] = (uint8_t)sym;
                nread++;
                consume(nb);
            } else {
                uint32_t base_rep = sym == 18 ? 11 : 3;
                uint32_t extra = sym == 16 ? 2 : sym == 17 ? 3 : 7;
                if (left < nb + extra) return RC_STUCK;
                uint32_t value = 0;
                if (sym == 16) {
                    if (nread == 0) return ST_INVALID_CODE_LENGTH_REPEAT;
                    wave_sync();
                    value = uni(lens[nread - 1]);
                }
                uint32_t rep = (((uint32_t)bb >> nb) & ((1u << extra) - 1)) + base_rep;
                if (nread + rep > total) return ST_INVALID_CODE_LENGTH_REPEAT;
                for (uint32_t i = lane; i < rep; i += kWave) lens[nread + i] = (uint8_t)value;
                nread += rep;
                consume(nb + extra);
            }
        }
        wave_sync();
        {   // :541-549: distance lengths to [288, 288+hdist), zero the gaps
            uint8_t dl = (lane < (int)hdist) ? lens[hlit + lane] : 0;
            wave_sync();
            for (uint32_t i = hlit + lane; i < 288; i += kWave) lens[i] = 0;
            if (lane < 32) lens[288 + lane] = dl;
            wave_sync();
        }
        return build_block_tables(hlit);
    }

    // ------------------------------------------------------------------ serial decoder
    // One symbol of a compressed block with the reference's careful-loop semantics
    // (src/decompress.rs:836-1015).  RC_OK / RC_EOB / RC_STUCK / status.
    __device__ __forceinline__ uint32_t serial_token() {
        // room for any single token (<= 258 bytes) so nothing below has to flush
        if (opos - flushed > (uint32_t)(kOutRing - kFlushSlack - 264)) flush(false);
        refill();
        if (opos == cap) {  // :838-840 then the trailing end-of-block peek :1009-1015
            if (left >= 15 && ((uint32_t)bb & eof_mask) == eof_code) {
                consume(eof_bits);
                return RC_EOB;
            }
            return RC_STUCK;
        }
        uint32_t e = uni(T.lit[(uint32_t)bb & (kLSize - 1)]);
        uint32_t nb = e & 15, kind = (e >> 4) & 15;
        if (kind == K_LIT1) {
            if (left < nb) return RC_STUCK;
            put_byte((e >> 8) & 0xFF);
            consume(nb);
            return RC_OK;
        }
        if (kind == K_LIT2) {
            if (left < nb) return RC_STUCK;
            last_was_pair = true;
            put_byte((e >> 8) & 0xFF);
            consume(nb);
            if (opos == cap) return RC_STUCK;  // second literal queued, :866-876
            put_byte((e >> 16) & 0xFF);
            return RC_OK;
        }
        uint32_t len_base, len_extra, lcb;
        if (kind == K_LONG) {  // secondary-table symbols, :886-909
            uint32_t sym;
            long_decode(T.lit_cb, T.lit_sorted, bb, sym, lcb);
            if (left < lcb) return RC_STUCK;
            if (sym < 256) {
                consume(lcb);
                put_byte(sym);
                return RC_OK;
            }
            if (sym == 256) {
                consume(lcb);
                return RC_EOB;
            }
            len_base = kLenBase[sym - 257];
            len_extra = kLenExtra[sym - 257];
        } else if (kind == K_EOB) {  // :912-917
            if (left < nb) return RC_STUCK;
            consume(nb);
            return RC_EOB;
        } else {  // K_LEN, :880-885
            lcb = nb;
            len_base = e >> 16;
            len_extra = (e >> 8) & 31;
        }
        // ---- length + distance, :919-965 ----
        const uint64_t left0 = left;
        uint32_t length = len_base + (uint32_t)((bb >> lcb) & ((1u << len_extra) - 1));
        consume(lcb + len_extra);  // (the reference consumes the whole token at once;
        refill();                  //  all of its bit checks are replayed on left0)
        uint32_t de = uni(T.dist[(uint32_t)bb & (kDistSize - 1)]);
        uint32_t dkind = (de >> 4) & 15;
        uint32_t dbase, dextra, dcb;
        if (dkind == D_DIST) {
            dbase = de >> 16;
            dextra = (de >> 8) & 15;
            dcb = de & 15;
        } else if (left0 > lcb + len_extra + kDistBits) {  // :932-933
            if (dkind == D_INVALID) return ST_INVALID_DISTANCE_CODE;
            uint32_t dsym;
            long_decode(T.dist_cb, T.dist_sorted, bb, dsym, dcb);
            if (dsym >= 30) return ST_INVALID_DISTANCE_CODE;
            dbase = kDistBase[dsym];
            dextra = kDistExtra[dsym];
        } else {
            return RC_STUCK;
        }
        uint32_t total_bits = lcb + len_extra + dcb + dextra;
        uint32_t dist = dbase + (uint32_t)((bb >> dcb) & ((1u << dextra) - 1));
        if (left0 < total_bits) return RC_STUCK;
        if (dist > opos) return ST_DISTANCE_TOO_FAR_BACK;
        consume(dcb + dextra);
        uint32_t n = min(length, cap - opos);
        copy_match(n, dist);
        if (n < length) return RC_STUCK;  // remainder queued, output full
        return RC_OK;
    }

    // The exact serial decoder taking over at a SYMBOL boundary in the middle of a block (a check point in front
    // of a tile, a resume point left by another kernel).  The reference walks the block table step by table step
    // (src/decompress.rs:836-1015), and a step is one symbol -- or two literals whose codes fit the table index
    // together (src/huffman.rs:110-130); where it stops when the input runs out depends on which symbols were
    // paired: `if bit_buffer.nbits < litlen_code_bits { break }` (:852) holds back BOTH literals of a pair.
    // So a symbol boundary is not enough to take over: it has to be the start of a step.  What is known:
    //   * a symbol that is no literal of the table (a length, end-of-block, a code beyond the index) is a
    //     step of its own: it starts one, and the next step starts behind it;
    //   * a literal whose table entry is a single (it cannot pair with what follows) ends a step whichever way
    //     it was reached: the next symbol starts a step;
    //   * behind the first literal of a pair entry the position stays in doubt (second half of that pair, or
    //     a step of its own if the step started one symbol earlier).
    // This walks single symbols until the position is the start of a step for certain.  RC_OK: it is (go on
    // with serial_token); RC_EOB: the block ended; RC_REDO: the stream ended, the slot filled up or an error
    // turned up while the position was in doubt -- the caller decodes from the first byte, as rounds 1-3 did
    // for every such stream; anything else: a result met at the start of a step, exact.
    __device__ __forceinline__ uint32_t resync_to_step_start(const uint32_t known) {
        if (known == STEP_START) return RC_OK;
        if (known == STEP_SECOND) {
            // The second literal of a pair.  The reference takes the pair in one step when all its bits are there,
            // and nothing of it otherwise (:852) -- then the first literal, already in the output, was one too
            // many.  (A code found within the bits that are left is the real one: no code is a prefix of another.)
            if (opos - flushed > (uint32_t)(kOutRing - kFlushSlack - 264)) flush(false);
            refill();
            const uint32_t e = uni(T.lit[(uint32_t)bb & (kLSize - 1)]);
            const uint32_t kind = (e >> 4) & 15;
            const uint32_t n2 = kind == K_LIT2 ? (e >> 24) : (e & 15);
            if ((kind != K_LIT1 && kind != K_LIT2) || left < n2) {
                opos -= 1;
                return RC_STUCK;
            }
            if (opos == cap) return RC_STUCK;  // the pair is taken, its second literal queued (:866-876): the slot is full
            put_byte((e >> 8) & 0xFF);
            consume(n2);
            return RC_OK;
        }
        for (;;) {
            if (opos - flushed > (uint32_t)(kOutRing - kFlushSlack - 264)) flush(false);
            refill();
            if (opos == cap) return RC_REDO;
            const uint32_t e = uni(T.lit[(uint32_t)bb & (kLSize - 1)]);
            const uint32_t nb = e & 15, kind = (e >> 4) & 15;
            if (kind == K_LIT2) {  // first literal of a pair entry: the position behind it stays in doubt
                if (left < nb) return RC_REDO;
                put_byte((e >> 8) & 0xFF);
                consume(e >> 24);
                continue;
            }
            const uint32_t rc = serial_token();
            if (kind == K_LIT1) return rc == RC_OK ? (uint32_t)RC_OK : (uint32_t)RC_REDO;
            return rc;  // no literal of the table: this was the start of a step
        }
    }

    // ------------------------------------------------------------------ tile decoder
    // Decodes up to 64 x 64 stream bits of the current block in parallel.  `progress` = stream
    // bits consumed.  Returns RC_OK, RC_EOB (block finished inside the tile) or a status.
    // Everything the tile cannot prove harmless (codes beyond the primary tables, invalid or
    // truncated tokens, a full slot, too many matches) ends the tile in front of the offending
    // symbol; the serial decoder then deals with exactly that symbol.
    __device__ __forceinline__ uint32_t tile_step(uint32_t& progress) {
        progress = 0;
        // keep the un-flushed part of the ring small so a whole tile fits
        if (opos - flushed > 1024) flush(false);
        const uint64_t P = consumed_bits();
        const uint64_t wbit = P + (uint64_t)mis * 8;
        const uint32_t d0 = (uint32_t)(wbit >> 5);
        const uint32_t sh = (uint32_t)wbit & 31;
        ensure_dw(d0);
        ensure_dw(d0 + 2 * kWave + 4);
        // this lane's 64 bits + 64 bits of look-ahead, normalised to start at bit 0 of w0
        uint32_t r0 = io.in_ring[(d0 + 2 * lane + 0) & (kInRingDw - 1)];
        uint32_t r1 = io.in_ring[(d0 + 2 * lane + 1) & (kInRingDw - 1)];
        uint32_t r2 = io.in_ring[(d0 + 2 * lane + 2) & (kInRingDw - 1)];
        uint32_t r3 = io.in_ring[(d0 + 2 * lane + 3) & (kInRingDw - 1)];
        uint32_t r4 = io.in_ring[(d0 + 2 * lane + 4) & (kInRingDw - 1)];
        const uint32_t w0 = __builtin_amdgcn_alignbit(r1, r0, sh);
        const uint32_t w1 = __builtin_amdgcn_alignbit(r2, r1, sh);
        const uint32_t w2 = __builtin_amdgcn_alignbit(r3, r2, sh);
        const uint32_t w3 = __builtin_amdgcn_alignbit(r4, r3, sh);
        // stream bits available from this lane's chunk start (may be <= 0 past the end)
        int64_t la64 = (int64_t)left - (int64_t)kTileBits * lane;
        const int32_t la = la64 > (1 << 20) ? (1 << 20) : (la64 < -1 ? -1 : (int32_t)la64);

        auto bits32 = [&](uint32_t p) __attribute__((always_inline)) -> uint32_t {  // 32 stream bits starting at chunk bit p (< 96)
            const bool a = p < 32, b = p < 64;
            uint32_t lo = vsel(a, w0, vsel(b, w1, w2));
            uint32_t hi = vsel(a, w1, vsel(b, w2, w3));
            return __builtin_amdgcn_alignbit(hi, lo, p & 31);
        };
        // Table entries with codes beyond the primary index resolved by a per-lane canonical walk
        // (the reference's secondary tables, src/huffman.rs:138-181): the result is the entry the
        // symbol would have had in a wide enough primary table.
        auto lit_at = [&](uint32_t w) __attribute__((always_inline)) -> uint32_t {
            uint32_t e = T.lit[w & (kLSize - 1)];
            if (((e >> 4) & 15) == K_LONG) {
                uint32_t sym, len;
                if (long_walk(T.lit_cb, T.lit_sorted, w, LB + 1, sym, len)) e = LitTraits::entry(sym, len);
            }
            return e;
        };
        auto dist_at = [&](uint32_t v) __attribute__((always_inline)) -> uint32_t {
            uint32_t de = T.dist[v & (kDistSize - 1)];
            if (((de >> 4) & 15) == D_LONG) {
                uint32_t sym, len;
                if (long_walk(T.dist_cb, T.dist_sorted, v, kDistBits + 1, sym, len)) de = DistTraits::entry(sym, len);
            }
            return de;
        };
        // Decodes the token at chunk bit p.  kind: 0 literal(s), 1 match, 2 end-of-block, 3 bad.
        // adv1: bits of the first symbol; adv: bits of the whole table entry / token.
        auto token = [&](uint32_t p, uint32_t& kind, uint32_t& adv1, uint32_t& adv) __attribute__((always_inline)) {
            uint32_t e = lit_at(bits32(p));
            uint32_t nb = e & 15, k = (e >> 4) & 15;
            if (k <= K_LIT2) {
                kind = 0;
                adv1 = e >> 24;
                adv = nb;
            } else if (k == K_LEN) {
                uint32_t t = nb + ((e >> 8) & 31);
                uint32_t de = dist_at(bits32(p + t));
                kind = (((de >> 4) & 15) == D_DIST) ? 1 : 3;
                adv = t + (de & 15) + ((de >> 8) & 15);
                adv1 = adv;
            } else if (k == K_EOB) {
                kind = 2;
                adv = adv1 = nb;
            } else {
                kind = 3;
                adv = adv1 = 0;
            }
            if ((int32_t)(p + adv) > la) kind = 3;  // token (or literal pair) not fully inside the input
        };

#ifdef FDH_DEBUG_TILES
        long long tq = clock64();
#ifdef FDH_DEBUG_SPAN
#define TPHASE(k) do { (void)tq; } while (0)
#else
#define TPHASE(k) do { long long tn = clock64(); GSTAT(k, tn - tq); tq = tn; } while (0)
#endif
#else
#define TPHASE(k) do { } while (0)
#endif
        // ---- pass 1: speculative chain from chunk bit 0 (lane 0's start is a real boundary) ----
        uint64_t mask = 0, mmask = 0;   // symbol starts / match starts inside [start, 64)
        uint32_t start = 0, endp = 0;   // chain start, chain end (>= 64 when it left the chunk)
        uint32_t stop = 0, stop_pos = 0, stop_nb = 0;  // 0 none, 1 end-of-block, 2 bad
        {
            uint32_t p = 0;
            while (p < (uint32_t)kTileBits) {
                uint32_t kind, a1, a;
                token(p, kind, a1, a);
                if (kind == 0) {
                    mask |= 1ull << p;
                    if (a != a1 && p + a1 < (uint32_t)kTileBits) {
                        mask |= 1ull << (p + a1);
                        p += a;
                    } else {
                        p += a1;
                    }
                } else if (kind == 1) {
                    mask |= 1ull << p;
                    mmask |= 1ull << p;
                    p += a;
                } else {
                    stop = kind == 2 ? 1 : 2;
                    stop_pos = p;
                    stop_nb = a;
                    break;
                }
            }
            endp = p;
        }
        TPHASE(10);
        // ---- synchronisation: hand every lane the real start of its chain ----
        // A lane whose predecessor's chain currently stops (end-of-block / bad token, possibly on
        // a still-speculative chain) has no start to take over and sits the iteration out.  At
        // exit every lane up to the first stop holds the true chain (induction from lane 0).
        bool converged = false;
        for (int iter = 0; iter < 2 * kWave; iter++) {
#ifdef FDH_DEBUG_TILES
            GSTAT(6, 1);  // synchronisation iterations
#endif
            uint32_t prev_end = __shfl_up(endp, 1, kWave);
            uint32_t prev_stop = __shfl_up(stop, 1, kWave);
            uint32_t in_start = prev_end - kTileBits;
            bool need = lane != 0 && prev_stop == 0 && in_start != start;
            if (!__any(need)) {
                converged = true;
                break;
            }
            if (need) {
                // follow the chain from in_start one symbol at a time until it meets the old
                // chain, leaves the chunk or stops
                uint64_t nm = 0, nmm = 0;
                uint32_t p = in_start;
                for (;;) {
                    if (p >= (uint32_t)kTileBits) {
                        mask = nm;
                        mmask = nmm;
                        stop = 0;
                        endp = p;
                        break;
                    }
                    if ((mask >> p) & 1) {  // merged: the rest of the old chain is right
                        uint64_t keep = ~((1ull << p) - 1);
                        mask = nm | (mask & keep);
                        mmask = nmm | (mmask & keep);
                        break;
                    }
                    uint32_t kind, a1, a;
                    token(p, kind, a1, a);
                    if (kind <= 1) {
                        nm |= 1ull << p;
                        if (kind == 1) nmm |= 1ull << p;
                        p += a1;
                    } else {
                        mask = nm;
                        mmask = nmm;
                        stop = kind == 2 ? 1 : 2;
                        stop_pos = p;
                        stop_nb = a;
                        endp = p;
                        break;
                    }
                }
                start = in_start;
            }
        }
        TPHASE(11);
        if (!converged) return RC_OK;  // pathological input: the serial decoder decides
        const uint64_t stopped = __ballot(stop != 0);
        const int stop_lane = stopped ? __ffsll((unsigned long long)stopped) - 1 : kWave;
        const bool live = lane <= stop_lane;
        const bool dead = !live;
        (void)dead;

        // ---- output bytes per lane: one per symbol + (length - 1) per match ----
        uint32_t count = live ? (uint32_t)__popcll(mask) : 0;
        uint32_t mcount = live ? (uint32_t)__popcll(mmask) : 0;
        if (mcount) {
            uint64_t mm = mmask;
            while (mm) {
                uint32_t p = (uint32_t)__builtin_ctzll(mm);
                mm &= mm - 1;
                uint32_t e = lit_at(bits32(p));
                uint32_t lcb = e & 15, lex = (e >> 8) & 31;
                uint32_t length = (e >> 16) + ((bits32(p + lcb)) & ((1u << lex) - 1));
                count += length - 1;
            }
        }
        // exclusive prefix sums (bytes in the low 20 bits, matches above)
        uint32_t packed = count | (mcount << 20);
        uint32_t incl = packed;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            uint32_t y = __shfl_up(incl, o, kWave);
            if (lane >= o) incl += y;
        }
        uint32_t excl = incl - packed;
        const uint32_t obase = excl & 0xFFFFF, mbase = excl >> 20;
        // lanes that would overflow the slot, the ring or the match list end the tile early
        const uint32_t room = min(cap - opos, (uint32_t)(kOutRing - kFlushSlack - 16) - (opos - flushed));
        const bool over = live && (obase + count > room || mbase + mcount > (uint32_t)kMaxMatches);
        const uint64_t over_mask = __ballot(over);
        const int cut_lane = over_mask ? __ffsll((unsigned long long)over_mask) - 1 : kWave;
        const bool emit = live && lane < cut_lane;
        // last emitting lane decides where the stream continues
        const int last_lane = min(min(stop_lane, cut_lane - 1), kWave - 1);
        if (last_lane < 0) return RC_OK;  // nothing fits: serial decides
        const uint32_t total_packed = __shfl(incl, last_lane, kWave);
        const uint32_t total = total_packed & 0xFFFFF, nmatch = total_packed >> 20;

        TPHASE(12);
        // ---- emit literals, list matches ----
        wave_sync();
        // With check points (and the reference's own tables, LB = 12) the tile also follows the reference's chain
        // of table steps through its symbols: a symbol whose table entry is a pair of literals is the first half
        // of a step when a step starts at it (the next symbol is then a second half, and a step starts behind
        // that one whatever ITS entry says); any other symbol is a step of its own.  Per symbol that is either
        // "swap" (pair entry) or "reset to: a step starts behind it"; a lane composes its symbols, the wavefront
        // its lanes (below).
        const bool track = keep_ck && LB == kLitBits;
        bool lane_reset = false;
        uint32_t lane_flip = 0;
        if (emit) {
            uint64_t m = mask;
            uint32_t opo = opos + obase + gmis;
            uint32_t mi = mbase;
            while (m) {
                uint32_t p = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                uint32_t e = lit_at(bits32(p));
                uint32_t k = (e >> 4) & 15;
                if (k <= K_LIT2) {
                    io.out_ring[opo & kOutMask] = (uint8_t)(e >> 8);
                    opo++;
                    uint32_t n1 = e >> 24;
                    if (k == K_LIT2) {
                        lane_flip ^= 1u;
                    } else {
                        lane_reset = true;
                        lane_flip = 0;
                    }
                    if (k == K_LIT2 && p + n1 < (uint32_t)kTileBits) {
                        io.out_ring[opo & kOutMask] = (uint8_t)(e >> 16);
                        opo++;
                        m &= ~(1ull << (p + n1));
                        // the second literal's own entry: does it pair with what follows?  (A chain that was followed
                        // symbol by symbol may have stopped AT the second literal -- its own pair does not fit the input --
                        // then the byte written above lies behind the tile's output and the symbol is not part of the tile.)
                        if (track && ((mask >> (p + n1)) & 1)) {
                            const uint32_t e2 = T.lit[bits32(p + n1) & (kLSize - 1)];
                            if (((e2 >> 4) & 15) == K_LIT2) {
                                lane_flip ^= 1u;
                            } else {
                                lane_reset = true;
                                lane_flip = 0;
                            }
                        }
                    }
                } else {  // match
                    lane_reset = true;
                    lane_flip = 0;
                    uint32_t lcb = e & 15, lex = (e >> 8) & 31;
                    uint32_t length = (e >> 16) + (bits32(p + lcb) & ((1u << lex) - 1));
                    uint32_t dv = bits32(p + lcb + lex);
                    uint32_t de = dist_at(dv);
                    uint32_t dcb = de & 15, dex = (de >> 8) & 15;
                    uint32_t dist = (de >> 16) + ((dv >> dcb) & ((1u << dex) - 1));
                    io.mlist[2 * mi] = (opo - gmis - opos) | (length << 16);
                    io.mlist[2 * mi + 1] = dist;
                    mi++;
                    opo += length;
                }
            }
        }
        if (track) {
            const uint64_t R = __ballot(emit && lane_reset), F = __ballot(emit && lane_flip != 0);
            if (R) {
                const int lr = 63 - __builtin_clzll((unsigned long long)R);
                step_state = (__popcll(F & ~((1ull << lr) - 1)) & 1) ? (uint32_t)STEP_SECOND : (uint32_t)STEP_START;
            } else if (step_state != STEP_UNKNOWN && (__popcll(F) & 1)) {
                step_state = step_state == STEP_START ? (uint32_t)STEP_SECOND : (uint32_t)STEP_START;
            }
        } else {
            step_state = STEP_UNKNOWN;
        }
        wave_sync();
        TPHASE(13);
        GSTAT(15, nmatch);
        // ---- matches ----
        // 1. Sources older than the ring window are final whatever else is pending: the short matches
        //    that read them are served first, ONE trip to global memory for all of them (an unaligned
        //    16-B load each -- what lies behind a match's end is older output of this stream, i.e.
        //    readable; it bypasses the L1, which may hold the line as it was before our later stores).
        //    On zlib-6 data that is three matches in four (the window is 2 KiB, the distances reach 32).
        // 2. What is left is compacted to the front of the list, usually into one batch of 64.
        // 3. Rounds against a frontier.  F = where the first match not copied yet starts: everything in
        //    front of F is final (literals were scattered above, the matches in front of it are done).
        //    A short match whose DISTINCT source bytes [at - dist, at - dist + min(length, dist)) end at
        //    or below F depends on nothing that is still missing -- the first one always qualifies --
        //    so all of those are copied at once, one lane each, ring to ring; what they write lies at or
        //    above F, where this round reads nothing.  Long matches and sources that straddle the ring
        //    window are taken by the whole wavefront when they come first.  A round that finds little
        //    to do hands the next few matches to the one-by-one path, which is what a chain of matches
        //    feeding each other needs.
        // (Round 2: only the matches whose sources end in front of the tile were copied side by side, a
        // trip to memory per batch, the rest replayed one by one: 42 k of 144 k cycles per tile.)
        {
            // (+ 1: the scatter above may have written ONE byte behind the tile's output -- the second literal of a pair
            //  the tile stopped at -- and in the ring that byte lies on top of position opos + total - kOutRing, which
            //  therefore counts as gone.  It is in the slot: room keeps 80 bytes between the ring's end and `flushed`.
            //  Round 5: found by the streaming soak, seed 1048 -- a match of the last tile in front of a cut read that
            //  position and delivered one wrong byte 2 KiB in front of the end of the input.)
            const uint32_t ring_top = opos + total + 1;
            const int64_t ring_lo = (int64_t)ring_top - kOutRing;
            constexpr int kBatches = (kMaxMatches + kWave - 1) / kWave;
            static_assert(kBatches == 3, "the loads below are written out for three batches");
            uint32_t ntodo = 0;
            {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                u32x4 v[kBatches];
                bool far[kBatches], left[kBatches];
                uint32_t m0s[kBatches], dists[kBatches];
                const uint8_t* src[kBatches];
                uint64_t fm[kBatches];
#pragma unroll
                for (int bi = 0; bi < kBatches; bi++) {
                    const uint32_t j = (uint32_t)bi * kWave + (uint32_t)lane;
                    const bool mine = j < nmatch;
                    const uint32_t m0 = mine ? io.mlist[2 * j] : 0u, dist = mine ? io.mlist[2 * j + 1] : 1u;
                    const uint32_t at = opos + (m0 & 0xFFFF), length = m0 >> 16;
                    if (__any(mine && dist > at)) return ST_DISTANCE_TOO_FAR_BACK;  // src/decompress.rs:782
                    const int64_t src_lo = (int64_t)at - (int64_t)dist;
                    far[bi] = mine && length <= 16 && dist >= length && src_lo + (int64_t)length <= ring_lo;
                    left[bi] = mine && !far[bi];
                    m0s[bi] = m0;
                    dists[bi] = dist;
                    src[bi] = out_al + gmis + (far[bi] ? src_lo : 0);
                    fm[bi] = __ballot(far[bi]);
                }
                if (fm[0] | fm[1] | fm[2]) {
                    // (one statement: the compiler must not touch the registers before the wait)
                    uint64_t saved;
                    asm volatile(
                        "s_waitcnt vmcnt(0)\n\t"  // our own flush stores
                        "s_mov_b64 %[sv], exec\n\t"
                        "s_mov_b64 exec, %[m0]\n\t"
                        "global_load_dwordx4 %[v0], %[a0], off sc1\n\t"
                        "s_mov_b64 exec, %[m1]\n\t"
                        "global_load_dwordx4 %[v1], %[a1], off sc1\n\t"
                        "s_mov_b64 exec, %[m2]\n\t"
                        "global_load_dwordx4 %[v2], %[a2], off sc1\n\t"
                        "s_mov_b64 exec, %[sv]\n\t"
                        "s_waitcnt vmcnt(0)"
                        : [v0] "=&v"(v[0]), [v1] "=&v"(v[1]), [v2] "=&v"(v[2]), [sv] "=&s"(saved)
                        : [a0] "v"(src[0]), [a1] "v"(src[1]), [a2] "v"(src[2]), [m0] "s"(fm[0]), [m1] "s"(fm[1]), [m2] "s"(fm[2])
                        : "memory");
                }
                wave_sync();  // every lane has read its list entries: the compaction below may overwrite them
#pragma unroll
                for (int bi = 0; bi < kBatches; bi++) {
                    if (far[bi]) {
                        const uint32_t at = opos + (m0s[bi] & 0xFFFF), length = m0s[bi] >> 16;
                        const bool more = __any(length > 8);
#pragma unroll
                        for (uint32_t k = 0; k < 16; k++) {
                            if (k >= 8 && !more) break;
                            if (k < length) io.out_ring[(at + k + gmis) & kOutMask] = (uint8_t)((v[bi][k >> 2] >> (8 * (k & 3))) & 0xFFu);
                        }
                    }
                    const uint64_t lm = __ballot(left[bi]);
                    if (left[bi]) {
                        const uint32_t k = ntodo + (uint32_t)__popcll(lm & lanemask_lt(lane));
                        io.mlist[2 * k] = m0s[bi];
                        io.mlist[2 * k + 1] = dists[bi];
                    }
                    ntodo += (uint32_t)__popcll(lm);
                }
                wave_sync();
            }
            TPHASE(16);
            uint32_t first = 0, one_by_one = 0;
            while (first < ntodo) {
                const uint32_t j0 = first & ~(uint32_t)(kWave - 1);
                const uint32_t j = j0 + (uint32_t)lane;
                // this batch of the list lives in registers while its rounds run
                uint32_t m0 = j < ntodo ? io.mlist[2 * j] : 0u;
                const uint32_t dist = j < ntodo ? io.mlist[2 * j + 1] : 1u;
                const uint32_t at = opos + (m0 & 0xFFFF);
                const int64_t src_lo = (int64_t)at - (int64_t)dist;
                const uint32_t batch_end = min(ntodo, j0 + (uint32_t)kWave);
                while (first < batch_end) {
                    const uint32_t length = m0 >> 16;
                    const uint64_t tm = __ballot(j >= first && j < batch_end && length != 0);
                    if (!tm) {
                        first = batch_end;
                        break;
                    }
                    const int fl = __ffsll((unsigned long long)tm) - 1;
                    first = j0 + (uint32_t)fl;
                    const uint32_t m0f = __builtin_amdgcn_readlane(m0, fl), df = __builtin_amdgcn_readlane(dist, fl);
                    const uint32_t atf = opos + (m0f & 0xFFFF), lenf = m0f >> 16;
                    const bool easy = lenf <= 16 && (int64_t)atf - (int64_t)df >= ring_lo;
                    if (!easy || one_by_one) {
                        copy_bytes(atf, lenf, df, ring_top);
                        GSTAT(20, 1);
                        wave_sync();
                        m0 = lane == fl ? m0 & 0xFFFFu : m0;
                        first++;
                        one_by_one -= one_by_one ? 1u : 0u;
                        continue;
                    }
                    // a round of this batch (later batches wait for theirs: their matches start further on anyway)
                    const bool near = j >= first && j < batch_end && length != 0 && length <= 16 && src_lo >= ring_lo &&
                                      src_lo + (int64_t)min(length, dist) <= (int64_t)atf;
                    const uint64_t nmask = __ballot(near);
                    if (near) {
                        uint32_t b[16];
                        uint32_t idx = 0;  // k mod dist (a match may repeat its own first bytes)
                        const bool more = __any(length > 8);
#pragma unroll
                        for (uint32_t k = 0; k < 16; k++) {
                            if (k >= 8 && !more) break;
                            if (k < length) b[k] = io.out_ring[((uint32_t)src_lo + idx + gmis) & kOutMask];
                            idx = idx + 1 == dist ? 0u : idx + 1;
                        }
#pragma unroll
                        for (uint32_t k = 0; k < 16; k++) {
                            if (k >= 8 && !more) break;
                            if (k < length) io.out_ring[(at + k + gmis) & kOutMask] = (uint8_t)b[k];
                        }
                        m0 &= 0xFFFFu;  // done
                    }
                    wave_sync();
                    GSTAT(19, 1);
                    GSTAT(21, __popcll(nmask));
                    if (__popcll(nmask) < 3) one_by_one = 6;
                }
            }
        }
        TPHASE(14);
#ifdef FDH_DEBUG_TILES_TRACE
        {
            uint32_t slot = g_dbg_n;  // single-stream debugging only
            uint32_t* d = g_dbg + 8 + slot * 264;
            if (slot < 200) {
                if (lane == 0) {
                    d[0] = (uint32_t)P; d[1] = total; d[2] = (uint32_t)stop_lane; d[3] = (uint32_t)cut_lane;
                    d[4] = (uint32_t)last_lane; d[5] = opos; d[6] = flushed; d[7] = (uint32_t)left;
                }
                d[8 + lane * 4 + 0] = start; d[8 + lane * 4 + 1] = endp; d[8 + lane * 4 + 2] = count | (stop << 16) | ((dead ? 1u : 0u) << 20);
                d[8 + lane * 4 + 3] = obase;
            }
            wave_sync();
            if (lane == 0) g_dbg_n = slot + 1;
        }
#endif
        opos += total;
        // ---- advance the bit reader ----
        uint32_t rc = RC_OK;
        uint32_t used;  // bits consumed relative to P
        if (last_lane == stop_lane) {
            uint32_t sp = __shfl(stop_pos, last_lane, kWave), sn = __shfl(stop_nb, last_lane, kWave);
            uint32_t sk = __shfl(stop, last_lane, kWave);
            used = (uint32_t)last_lane * kTileBits + sp;
            // With the slot exactly full the reference only accepts the real end-of-block code
            // (src/decompress.rs:1009-1015; fixed symbols 286/287 do not qualify): serial_token
            // owns that rule, so the end-of-block symbol is left to it.
            if (sk == 1 && opos != cap) {
                used += sn;
                rc = RC_EOB;
            }
        } else if (cut_lane < kWave) {
            used = (uint32_t)cut_lane * kTileBits + __shfl(start, cut_lane, kWave);
        } else {
            used = (uint32_t)(kWave - 1) * kTileBits + __shfl(endp, kWave - 1, kWave);
        }
        left -= used;
        progress = used;
        seek(P + used);
        return rc;
    }

    // ------------------------------------------------------------------ span decoder
    // Segment-parallel decode of the next 64 x `seg` stream bits of the current block -- the scheme
    // of inflate_segments.h with this block's own tables: every lane decodes its bit range
    // sequentially (pass 1 from a guessed start: walk a 256-bit window, then count bytes and
    // matches; check: the real start from the left neighbour must land on the same boundary;
    // prefix sums; pass 2 writes the literals to the output slot and lists the matches, leaving
    // zeroed holes), then the matches are resolved from the list, 64 at a time, against a
    // "resolved up to" frontier; the Adler-32 is taken over the finished bytes.  The staging rings
    // of the wavefront are reused as per-lane rings; the input window and the output ring are
    // re-established afterwards.  Like a tile, a span ends in front of anything it cannot prove
    // harmless.  `progress` = stream bits consumed.
    struct SpanTok {
        uint32_t kind;    // 0 literal(s), 1 match, 2 end-of-block, 3 bad
        uint32_t bits_a;  // literal entry / end-of-block: all its bits; match: length code + extra bits
        uint32_t bits_b;  // match: distance code + extra bits
        uint32_t n, lits, n1;  // literals: count, bytes, bits of the first one alone
        uint32_t length, dist;
    };
    __device__ __forceinline__ SpanTok span_token(uint64_t w) {
        SpanTok t;
        t.bits_b = t.n = t.lits = t.n1 = t.length = t.dist = 0;
        uint32_t e = T.lit[(uint32_t)w & (kLSize - 1)];
        if (((e >> 4) & 15) == K_LONG) {
            uint32_t sym, len;
            if (long_walk(T.lit_cb, T.lit_sorted, (uint32_t)w, LB + 1, sym, len)) e = LitTraits::entry(sym, len);
        }
        const uint32_t nb = e & 15, k = (e >> 4) & 15;
        t.bits_a = nb;
        t.kind = 3;
        if (k == K_LIT1) {
            t.kind = 0;
            t.n = 1;
            t.lits = (e >> 8) & 0xFF;
            t.n1 = nb;
        } else if (k == K_LIT2) {
            t.kind = 0;
            t.n = 2;
            t.lits = (e >> 8) & 0xFFFF;
            t.n1 = e >> 24;
        } else if (k == K_LEN) {
            const uint32_t lex = (e >> 8) & 31;
            t.length = (e >> 16) + ((uint32_t)(w >> nb) & ((1u << lex) - 1));
            t.bits_a = nb + lex;
            const uint32_t v = (uint32_t)(w >> t.bits_a);
            uint32_t de = T.dist[v & (kDistSize - 1)];
            if (((de >> 4) & 15) == D_LONG) {
                uint32_t sym, len;
                if (long_walk(T.dist_cb, T.dist_sorted, v, kDistBits + 1, sym, len)) de = DistTraits::entry(sym, len);
            }
            if (((de >> 4) & 15) == D_DIST) {
                const uint32_t dcb = de & 15, dex = (de >> 8) & 15;
                t.dist = (de >> 16) + ((v >> dcb) & ((1u << dex) - 1));
                t.bits_b = dcb + dex;
                t.kind = 1;
            }
        } else if (k == K_EOB) {
            t.kind = 2;
        }
        return t;
    }
    // 62 stream bits at the reader's position (the reader sits two bits in front of the token)
    __device__ __forceinline__ uint64_t span_window(const SegReader& rd, uint32_t nw) {
        const uint32_t w_lo = __builtin_amdgcn_alignbit(rd.hi, rd.lo, rd.boff);
        const uint32_t w_hi = __builtin_amdgcn_alignbit(nw, rd.hi, rd.boff);
        return (((uint64_t)w_hi << 32) | w_lo) >> 2;
    }
    __device__ __forceinline__ void span_events(SegReader& rd, bool running) {
        rd.event(running);
        for (int x = 0; x < 2 && __any(running && rd.level() < 8); x++) rd.event(running);  // tokens up to 48 bits
    }
    struct SpanScan {
        uint32_t pos, count, nmatch, stop, stop_bits;
    };
    // GUESS + WINDOW: walk the synchronisation window without counting, sliding over impossible
    // tokens.  WINDOW alone: the real chain through the window, counted, literal pairs split close
    // to the window's end.  Neither: count to `stop_at`.
    template <bool GUESS, bool WINDOW>
    __device__ __forceinline__ void span_scan(SegReader& rd, uint32_t limit, bool active, uint32_t stop_at, SpanScan& s) {
        bool running = active && s.stop == 0 && s.pos < stop_at;
        if (!WINDOW && running) rd.refill_now();
        while (__any(running)) {
            span_events(rd, running);
#pragma unroll 1
            for (int k = 0; k < 4; k++) {
                const uint32_t nw = rd.peek();
                const SpanTok t = span_token(span_window(rd, nw));
                uint32_t kind = t.kind, bits_a = t.bits_a, bits_b = t.bits_b, n = t.n;
                if (WINDOW) {
                    const bool single = kind == 0 && n == 2 && s.pos + 24 >= (uint32_t)kSegWindow;
                    bits_a = single ? t.n1 : bits_a;
                    n = single ? 1u : n;
                }
                if (GUESS) {
                    const bool slide = kind >= 2 && s.pos + 1 <= limit;
                    bits_a = slide ? 1u : bits_a;
                    bits_b = slide ? 0u : bits_b;
                    n = slide ? 0u : n;
                    kind = slide ? 0u : kind;
                }
                const uint32_t bits = bits_a + bits_b;
                const bool fault = kind == 3 || s.pos + bits > limit || rd.starved();
                const bool step = running && !fault && kind != 2;
                const bool halt = running && !step;
                s.stop = halt ? (fault ? 2u : 1u) : s.stop;
                s.stop_bits = halt ? bits_a : s.stop_bits;
                if (!GUESS) {
                    s.count += step ? (kind == 1 ? t.length : n) : 0u;
                    s.nmatch += (step && kind == 1) ? 1u : 0u;
                }
                rd.advance(step ? bits_a : 0u, nw);
                if (__any(step && kind == 1)) {
                    const uint32_t nw2 = rd.peek();
                    rd.advance((step && kind == 1) ? bits_b : 0u, nw2);
                }
                s.pos += step ? bits : 0u;
                running = step && s.pos < stop_at;
            }
        }
    }
    // The last kOutRing bytes of output back into the output ring (history of later matches), by
    // whole 16-B lines of the slot's address space; what lies outside [.., opos) is never read.
    // Everything up to opos must be in the slot (flushed == opos).
    __device__ __forceinline__ void span_reload_out_ring() { reload_out_ring(); }
    __device__ __forceinline__ uint32_t span_step(uint32_t& progress) {
        progress = 0;
        const uint64_t P64 = consumed_bits();
        const uint64_t total_bits64 = (win_bytes - mis) * 8;
        if (total_bits64 >= (1ull << 31) || cap >= (1u << 31) || P64 < 8) {
            span_credit = ~0u;
            return RC_OK;
        }
        const uint32_t P = (uint32_t)P64, in_bits = (uint32_t)total_bits64;
        const uint32_t seg = min(kSpanSegBits, (in_bits - P + kWave - 1) / kWave);
        if (seg < kSpanMinSeg) {  // the tail of a stream: tiles
            span_credit = ~0u;
            return RC_OK;
        }
        flush(true);  // everything decoded so far is in the slot; the staging rings are free now
        const uint32_t seg_bit0 = P + (uint32_t)lane * seg;
        const bool in_range = seg_bit0 < in_bits;
        const uint32_t limit = in_range ? in_bits - seg_bit0 : 0;

        SegReader rd;
        rd.ring = reinterpret_cast<uint32_t*>(&io);
        rd.lane_off = (uint32_t)lane;
        rd.buf_lo = buf_lo;
        rd.buf_hi = buf_hi;
        rd.gp = in;
        rd.in_wr = rd.in_rd = 0;
        rd.lo = rd.hi = rd.boff = 0;
        for (int k = 0; k < kSegChunk; k++) rd.pend_a.w[k] = rd.pend_b.w[k] = 0;
        rd.has_a = rd.has_b = false;
        uint32_t* const oring = reinterpret_cast<uint32_t*>(&io) + kSegInWords * kWave;
        // (the span rings overlay the staging rings: a kernel that enables spans keeps 8 KiB at &io)
        loaded = 0;  // the input window is gone (seek() below reloads it)

#ifdef FDH_DEBUG_SPAN
        long long sq = clock64();
#define SPHASE(k) do { long long tn = clock64(); GSTAT(k, tn - sq); sq = tn; } while (0)
#else
#define SPHASE(k) do { } while (0)
#endif
        // ---- pass 1 ----
        SpanScan tail;
        tail.pos = tail.count = tail.nmatch = tail.stop = tail.stop_bits = 0;
        if (in_range) rd.start(in, seg_bit0);
        span_scan<true, true>(rd, limit, in_range, (uint32_t)kSegWindow, tail);
        uint32_t x0 = tail.stop == 0 ? tail.pos : 0;
        span_scan<false, false>(rd, limit, in_range, seg, tail);
        SPHASE(10);
        // ---- check ----
        SpanScan head;
        head.pos = head.count = head.nmatch = head.stop = head.stop_bits = 0;
        uint32_t start = 0, cur_start = ~0u;
        bool giveup = false;
        for (int round = 0; round < 6; round++) {
            const uint32_t prev_end = __shfl_up(tail.pos, 1, kWave);
            const uint32_t prev_stop = __shfl_up(tail.stop, 1, kWave);
            start = lane == 0 ? 0 : prev_end - seg;
            const bool have_in = lane == 0 || (prev_stop == 0 && prev_end >= seg);
            const bool need = in_range && have_in && start != cur_start;
            if (!__any(need)) break;
            if (round == 5) giveup = true;
            if (need) {
                head.pos = start;
                head.count = head.nmatch = head.stop = head.stop_bits = 0;
                rd.start(in, seg_bit0 + start);
            }
            span_scan<false, true>(rd, limit, need, (uint32_t)kSegWindow, head);
            const bool stopped_in_head = need && head.stop != 0;
            const bool redo = need && head.stop == 0 && (head.pos != x0 || x0 == 0);
            if (stopped_in_head) {
                tail = head;
                tail.count = tail.nmatch = 0;
                x0 = head.pos;
            }
            if (__any(redo)) {
                if (redo) {
                    tail.pos = head.pos;
                    tail.count = tail.nmatch = tail.stop = tail.stop_bits = 0;
                    x0 = head.pos;
                }
                span_scan<false, false>(rd, limit, redo, seg, tail);
            }
            if (need) cur_start = start;
        }
        const bool verified = in_range && cur_start == start;
        const uint64_t stop_mask = __ballot(verified && tail.stop != 0);
        const uint64_t unver_mask = __ballot(!verified);
        const int stop_lane = stop_mask ? __ffsll((unsigned long long)stop_mask) - 1 : kWave;
        const int first_unver = unver_mask ? __ffsll((unsigned long long)unver_mask) - 1 : kWave;
        // lanes [0, nlive) hold verified chains; the last of them may end at a stop
        const int nlive = min(first_unver, stop_lane + 1);
        if (giveup || nlive == 0) {
            span_credit = 16;
            span_reload_out_ring();
            seek(P64);
            return RC_OK;
        }
        const bool live = lane < nlive;
        const uint32_t count = live ? head.count + tail.count : 0;
        const uint32_t mcount = live ? head.nmatch + tail.nmatch : 0;
        unsigned long long incl = (unsigned long long)count | ((unsigned long long)mcount << 40);
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            unsigned long long y = __shfl_up(incl, o, kWave);
            if (lane >= o) incl += y;
        }
        const unsigned long long tot = __shfl(incl, kWave - 1, kWave);
        const unsigned long long total64 = tot & ((1ull << 40) - 1);
        const uint32_t nmatch = (uint32_t)(tot >> 40);
        if (total64 > (unsigned long long)(cap - opos) || nmatch > kSpanMaxMatches) {  // tiles cope with these
            span_credit = 64;
            span_reload_out_ring();
            seek(P64);
            return RC_OK;
        }
        const uint32_t total = (uint32_t)total64;
        const uint32_t obase = (uint32_t)(incl & ((1ull << 40) - 1)) - count;
        uint32_t mi = (uint32_t)(incl >> 40) - mcount;

        SPHASE(11);
        // ---- pass 2: literals to the slot, matches to the list (their bytes are zero for now) ----
        {
            const uint32_t my_end = live ? tail.pos : 0u;
            uint32_t pos = start;
            uint32_t guard = 0;
            uint8_t* const op = out_al + gmis + opos;
            const uint32_t pad = (uint32_t)(reinterpret_cast<uintptr_t>(op) + obase) & 15;
            uint8_t* const line0 = op + obase - pad;
            uint32_t vposw = pad >> 2, acc = 0, sh = 8 * (pad & 3), vstored = 0, fill = 0, produced = 0;
            const uint32_t vend = pad + count;
            oring[seg_slot((uint32_t)lane, 0)] = 0;
            oring[seg_slot((uint32_t)lane, 1)] = 0;
            oring[seg_slot((uint32_t)lane, 2)] = 0;
            auto store_piece = [&](uint32_t vs) __attribute__((always_inline)) {
                const uint32_t w = vs >> 2;
                uint4 q;
                q.x = oring[seg_slot((uint32_t)lane, w + 0)];
                q.y = oring[seg_slot((uint32_t)lane, w + 1)];
                q.z = oring[seg_slot((uint32_t)lane, w + 2)];
                q.w = oring[seg_slot((uint32_t)lane, w + 3)];
                if (vs >= pad && vs + 16 <= vend) {
                    *reinterpret_cast<uint4*>(line0 + vs) = q;
                } else {
                    for (uint32_t k = 0; k < 16; k++) {
                        const uint32_t word = k < 4 ? q.x : (k < 8 ? q.y : (k < 12 ? q.z : q.w));
                        if (vs + k >= pad && vs + k < vend) line0[vs + k] = (uint8_t)(word >> (8 * (k & 3)));
                    }
                }
            };
            auto drain = [&]() __attribute__((always_inline)) {
                while (__any(4 * vposw - vstored >= 16)) {
                    if (4 * vposw - vstored >= 16) {
                        store_piece(vstored);
                        vstored += 16;
                    }
                }
            };
            if (live) rd.start(in, seg_bit0 + pos);
            while (__any(pos < my_end || fill != 0)) {
                if (++guard > (1u << 20)) return ST_INVALID_LITERAL_LENGTH_CODE;  // cannot happen; never hang
                drain();
                if (__any(fill >= 64)) {  // a long hole: whole zero lines straight to the slot
                    if (fill >= 64) {
                        oring[seg_slot((uint32_t)lane, vposw)] = acc;
                        vposw++;
                        fill -= 4 - (sh >> 3);
                        acc = 0;
                        sh = 0;
                        while (vposw & 3) {
                            oring[seg_slot((uint32_t)lane, vposw)] = 0;
                            vposw++;
                            fill -= 4;
                        }
                        while (4 * vposw != vstored) {
                            store_piece(vstored);
                            vstored += 16;
                        }
                        uint32_t lines = min(fill, vend - vstored) >> 4;
                        const uint32_t m = lines << 4;
                        uint8_t* dst = line0 + vstored;
                        for (; lines; lines--, dst += 16) *reinterpret_cast<uint4*>(dst) = make_uint4(0, 0, 0, 0);
                        vstored += m;
                        vposw += m >> 2;
                        fill -= m;
                    }
                }
                span_events(rd, live && (pos < my_end || fill != 0));
#pragma unroll 1
                for (int k = 0; k < 4; k++) {
                    const uint32_t nw = rd.peek();
                    SpanTok t = span_token(span_window(rd, nw));
                    // same stepping as the counted chains: single literals close to the window's end, so
                    // that this pass crosses the window on x0 and then follows the counted tail exactly
                    const bool single = t.kind == 0 && t.n == 2 && pos < (uint32_t)kSegWindow && pos + 24 >= (uint32_t)kSegWindow;
                    t.bits_a = single ? t.n1 : t.bits_a;
                    t.n = single ? 1u : t.n;
                    t.lits = single ? (t.lits & 0xFF) : t.lits;
                    const bool filling = fill != 0;
                    const bool dec = live && !filling && pos < my_end;
                    const bool is_match = dec && t.kind == 1;
                    if (is_match) {
                        span_list[2 * mi] = opos + obase + produced;
                        span_list[2 * mi + 1] = t.length | (t.dist << 16);
                        mi++;
                    }
                    const uint32_t nf = min(fill, 4u);
                    const uint32_t n = filling ? nf : (dec && t.kind == 0 ? t.n : 0u);
                    const uint32_t v = (!filling && dec && t.kind == 0) ? t.lits : 0u;
                    fill = filling ? fill - nf : (is_match ? t.length : 0u);
                    produced += dec ? (t.kind == 1 ? t.length : (t.kind == 0 ? t.n : 0u)) : 0u;
                    rd.advance(dec ? t.bits_a : 0u, nw);
                    if (__any(is_match)) {
                        const uint32_t nw2 = rd.peek();
                        rd.advance(is_match ? t.bits_b : 0u, nw2);
                    }
                    pos += dec ? t.bits_a + t.bits_b : 0u;
                    const uint64_t tt = (uint64_t)v << sh;
                    acc |= (uint32_t)tt;
                    oring[seg_slot((uint32_t)lane, vposw)] = acc;
                    const uint32_t tot8 = sh + 8 * n;
                    const bool full = tot8 >= 32;
                    acc = full ? (uint32_t)(tt >> 32) : acc;
                    vposw += full ? 1u : 0u;
                    sh = tot8 & 31;
                }
            }
            drain();
            if (live) {
                oring[seg_slot((uint32_t)lane, vposw)] = acc;
                for (uint32_t w = vposw + 1; (w & 3) != 0; w++) oring[seg_slot((uint32_t)lane, w)] = 0;
                if (vstored < vend) store_piece(vstored);
            }
            // a disagreement between the passes would be a bug; never publish such a span
            if (__any(live && (4 * vposw + (sh >> 3) != vend || produced != count))) return ST_INVALID_LITERAL_LENGTH_CODE;
        }
        SPHASE(12);
        // ---- pass 3: matches, in stream order, 64 at a time against the "resolved up to" frontier ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sync();
        {
            uint8_t* const o = out_al + gmis;  // o[p] = output byte p of this stream
            for (uint32_t j0 = 0; j0 < nmatch; j0 += kWave) {
                const uint32_t j = j0 + (uint32_t)lane;
                const bool mine = j < nmatch;
                uint32_t at = 0xFFFFFFFFu, ld = 0;
                if (mine) {
                    at = __hip_atomic_load(span_list + 2 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ld = __hip_atomic_load(span_list + 2 * j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const uint32_t length = ld & 0xFFFF, dist = ld >> 16;
                if (__any(mine && dist > at)) return ST_DISTANCE_TOO_FAR_BACK;  // src/decompress.rs:782
                const uint32_t src_end = at - dist + min(length, dist);
                const float inv_d = 1.0f / (float)max(dist, 1u);
                bool done = !mine;
                for (uint32_t rounds = 0;; rounds++) {
                    if (rounds > 2 * kWave) return ST_INVALID_LITERAL_LENGTH_CODE;  // cannot happen; never hang
                    uint32_t F = done ? 0xFFFFFFFFu : at;  // first unresolved match of the batch
#pragma unroll
                    for (int x = 32; x > 0; x >>= 1) F = min(F, (uint32_t)__shfl_xor(F, x, kWave));
                    if (F == 0xFFFFFFFFu) break;
                    const bool ready = !done && (at == F || src_end <= F);
                    // out[at + k] = out[at - dist + k mod dist]: every source byte lies in front of `at`
                    for (uint32_t c = 0; __any(ready && c < length); c += 16) {
                        uint32_t b[16];
#pragma unroll
                        for (uint32_t k = 0; k < 16; k++) {
                            const uint32_t kk = (ready && c + k < length) ? c + k : 0u;
                            uint32_t r = kk;
                            if (dist < length) {  // kk mod dist for kk < 258 (quotient estimate off by at most one)
                                const uint32_t q = (uint32_t)((float)kk * inv_d);
                                r = kk - q * dist;
                                r = (int32_t)r < 0 ? r + dist : r;
                                r = r >= dist ? r - dist : r;
                            }
                            const uint8_t* sp = (ready && c + k < length) ? o + (at - dist + r) : o;
                            b[k] = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
#pragma unroll
                        for (uint32_t k = 0; k < 16; k++) {
                            if (ready && c + k < length) o[at + c + k] = (uint8_t)b[k];
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next round reads these bytes
                    done = done || ready;
                }
            }
        }
        SPHASE(13);
        GSTAT(15, nmatch);
        // ---- Adler-32 of the new bytes, output ring, input window ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        {
            const uint32_t q_lo = opos + gmis, q_hi = opos + total + gmis;
            for (uint32_t it = q_lo & ~15u; it < q_hi; it += kWave * 16) {
                const uint32_t lq = it + (uint32_t)lane * 16;
                const uint32_t blk_hi = min(q_hi, it + kWave * 16), blk_lo = max(q_lo, it);
                uint32_t sum = 0, wsum = 0;
                if (lq < blk_hi && lq + 16 > blk_lo) {
                    const uint32_t lo = (blk_lo > lq) ? blk_lo - lq : 0, hi = (blk_hi < lq + 16) ? blk_hi - lq : 16;
                    const uint32_t W = blk_hi - lq;  // weight of byte j is W - j
                    if (lo == 0 && hi == 16) {
                        const uint32_t* lp = reinterpret_cast<const uint32_t*>(out_al + lq);
                        const uint32_t vx = __hip_atomic_load(lp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const uint32_t vy = __hip_atomic_load(lp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const uint32_t vz = __hip_atomic_load(lp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const uint32_t vw = __hip_atomic_load(lp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sum = bytesum4(vx) + bytesum4(vy) + bytesum4(vz) + bytesum4(vw);
                        uint32_t u = bytedot4(vx, 0x03020100u, 0);
                        u = bytedot4(vy, 0x07060504u, u);
                        u = bytedot4(vz, 0x0b0a0908u, u);
                        u = bytedot4(vw, 0x0f0e0d0cu, u);
                        wsum = W * sum - u;
                    } else {
                        for (uint32_t b4 = lo; b4 < hi; b4++) {
                            const uint32_t b = __hip_atomic_load(out_al + lq + b4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            sum += b;
                            wsum += (W - b4) * b;
                        }
                    }
                }
                const uint32_t S = wave_sum_u32(sum), Tt = wave_sum_u32(wsum);
                const uint32_t Lb = blk_hi - blk_lo;
                adler_b = (uint32_t)(((uint64_t)adler_b + (uint64_t)Lb * adler_a + Tt) % kAdlerMod);
                adler_a = (adler_a + S) % kAdlerMod;
            }
        }
        opos += total;
        flushed = opos;
        wave_sync();
        span_reload_out_ring();
        // ---- where the stream continues ----
        const int last = nlive - 1;
        const uint32_t last_pos = __shfl(tail.pos, last, kWave);
        const uint32_t last_stop = __shfl(tail.stop, last, kWave), last_bits = __shfl(tail.stop_bits, last, kWave);
        uint32_t used = (uint32_t)last * seg + last_pos;
        uint32_t rc = RC_OK;
        // with the slot exactly full the end-of-block symbol is left to serial_token (see tile_step)
        if (last_stop == 1 && opos != cap) {
            used += last_bits;
            rc = RC_EOB;
        }
        left -= used;
        progress = used;
        seek(P64 + used);
        SPHASE(14);
        return rc;
    }

    // ------------------------------------------------------------------ compressed block data
    template <bool TILES>
    __device__ __forceinline__ uint32_t decode_block_data() {
        for (;;) {
            if (SPANS && TILES && span_list && span_credit == 0 && serial_credit == 0 && opos < cap) {
                uint32_t progress;
#ifdef FDH_DEBUG_TILES
                const long long ts = clock64();
#endif
                uint32_t rc = span_step(progress);
#ifdef FDH_DEBUG_TILES
                GSTAT(3, 1);
                GSTAT(4, progress);
                GSTAT(7, clock64() - ts);
#endif
                step_state = STEP_UNKNOWN;  // (spans do not keep track of the reference's steps)
                if (rc != RC_OK) return rc;
                if (progress) continue;
                if (span_credit == 0) span_credit = 8;
            }
            if (TILES && serial_credit == 0 && opos < cap) {
                uint32_t progress;
                if (span_credit && span_credit != ~0u) span_credit--;
                if (keep_ck) take_ck();
#ifdef FDH_DEBUG_TILES
                const long long t0 = clock64();
#endif
                uint32_t rc = tile_step(progress);
#ifdef FDH_DEBUG_TILES
                GSTAT(0, 1);
                GSTAT(1, progress);
                GSTAT(2, clock64() - t0);
                if (progress < 16 * kTileBits) GSTAT(5, 1);
#endif
                if (rc != RC_OK) return rc;
                // a tile that got stuck early: let the serial decoder clear the obstacle
                if (progress < 16 * kTileBits) serial_credit = progress == 0 ? 4 : 1;
                if (progress) continue;
            }
#ifdef FDH_DEBUG_TILES
            const long long t1 = clock64();
            const uint64_t c0 = consumed_bits();
#endif
            last_was_pair = false;
            tok_bit = consumed_bits();
            // (close to the end of the slot every step may be the last one that fits: the place to take the stream
            //  up again with a larger slot is a check point in front of it)
            if (keep_ck && step_state == STEP_START && cap - opos <= 320u) take_ck();
            const bool at_step = step_state == STEP_START;
            uint32_t rc = serial_token();
            stuck_at_step = rc == RC_STUCK && at_step && opos != cap;
            // (from the start of a step it took a step; from anywhere else a symbol that pairs with nothing puts it in step)
            if (step_state != STEP_START) step_state = last_was_pair ? (uint32_t)STEP_UNKNOWN : (uint32_t)STEP_START;
#ifdef FDH_DEBUG_TILES
            (void)t1;
            (void)c0;
#endif
            if (rc != RC_OK) return rc;
            if (serial_credit) serial_credit--;
        }
    }

    // ------------------------------------------------------------------ checksum / results
    // src/decompress.rs:306-326: skip to the byte boundary, read the big-endian Adler-32.
    __device__ __forceinline__ uint32_t read_trailer(uint32_t& stored) {
        refill();
        uint32_t align = (uint32_t)(left & 7);
        if (left < 32 + align) return RC_STUCK;
        consume(align);
        refill();
        stored = __builtin_bswap32((uint32_t)bb);
        consume(32);
        return RC_OK;
    }

    // Final flush (the only one that may store a partial line) + classification.
    // rc = RC_OK means "trailer read": the checksum comparison happens here, after the flush.
    __device__ __forceinline__ StreamResult finish(uint32_t rc, uint32_t stored) {
        StreamResult r;
        r.ambiguous = false;
        flush(true);
#ifdef FDH_DEBUG_TILES
        if (lane == 0)
            for (int k = 0; k < 24; k++)
                if (gacc[k]) atomicAdd(&g_gstat[k], gacc[k]);
#endif
        uint32_t adler = (adler_b << 16) | adler_a;
        if (rc == RC_OK) {
            r.status = (!(flags & 1u) && stored != adler) ? (uint32_t)ST_WRONG_CHECKSUM : (uint32_t)ST_OK;
        } else if (rc == RC_STUCK) {
            // src/decompress.rs:1126-1139: not done and no error -> OutputTooLarge if the slot is
            // full, InsufficientInput otherwise.
            r.status = (opos == cap) ? ST_OUTPUT_TOO_LARGE : ST_INSUFFICIENT_INPUT;
            r.ambiguous = left < 128;
        } else {
            r.status = rc;
        }
        r.out_len = opos;
        r.adler = adler;
        return r;
    }

    // Where a stream that ran out of input (r from finish()) can be taken up again once more input has arrived:
    // the start of the token that did not fit -- the state of the reference's own decoder at that moment -- or,
    // when it was not a token of block data that stopped the run, the last check point.
    __device__ __forceinline__ ResumePoint stopped_at(const StreamResult& r) const {
        ResumePoint rp = ck;
        if (stuck_at_step && r.status == ST_INSUFFICIENT_INPUT) {
            rp.hdr_bit = hdr_bit;
            rp.bit = tok_bit;
            rp.opos = r.out_len;
            rp.adler = r.adler;
            rp.valid = 1;
            rp.step = STEP_START;
        }
        return rp;
    }

    // Whole stream.  START_IN_BLOCK: the zlib header and the (final, dynamic) block header were
    // recognised by the caller and the tables are in place.
    template <bool TILES, bool START_IN_BLOCK>
    __device__ __forceinline__ StreamResult run() {
        uint32_t rc = RC_OK, stored = 0;
        if (!START_IN_BLOCK) rc = parse_zlib_header();
        while (rc == RC_OK) {
            if (!START_IN_BLOCK) {
#ifdef FDH_DEBUG_TILES
                const long long th = clock64();
#endif
                hdr_bit = consumed_bits();
                if (keep_ck) take_ck();
                rc = parse_block_header();
#ifdef FDH_DEBUG_TILES
                GSTAT(8, clock64() - th);
                GSTAT(9, 1);
#endif
                step_state = STEP_START;
                if (rc == RC_OK) rc = decode_block_data<TILES>();
            } else {
                step_state = STEP_START;
                rc = decode_block_data<TILES>();
            }
            if (rc != RC_EOB) break;  // stuck or error
            if (last_block) {
                rc = read_trailer(stored);
                break;
            }
            rc = RC_OK;
        }
        return finish(rc, stored);
    }

    // The rest of the stream from a resume point (init() has run): the zlib header is looked at all the same,
    // the block header at rp.hdr_bit is parsed again for its tables.
    template <bool TILES>
    __device__ __forceinline__ StreamResult run_from(const ResumePoint rp) {
        uint32_t rc = parse_zlib_header(), stored = 0;
        if (rc == RC_OK) {
            opos = flushed = rp.opos;
            adler_a = rp.adler & 0xFFFF;
            adler_b = rp.adler >> 16;
            reload_out_ring();
            seek_to(rp.hdr_bit);
        }
        bool first = true;
        while (rc == RC_OK) {
            hdr_bit = consumed_bits();
            if (keep_ck && !first) take_ck();
            rc = parse_block_header();
            step_state = STEP_START;
            if (rc == RC_OK) {
                if (first && rp.bit != rp.hdr_bit) {
                    seek_to(rp.bit);
                    // (with tiles the result is only final when it does not depend on the pairing: see needs_serial_recheck)
                    step_state = rp.step;
#ifdef FDH_DEBUG_STEP
                    if (lane == 0) printf("resume bit %llu opos %u step %u left %llu\n", (unsigned long long)rp.bit, rp.opos, rp.step, (unsigned long long)left);
#endif
                    if (!TILES) {
                        rc = resync_to_step_start(rp.step);
                        step_state = STEP_START;
                    }
                    if (rc == RC_REDO) {
                        StreamResult r;
                        r.status = RC_REDO;
                        r.out_len = 0;
                        r.adler = 0;
                        r.ambiguous = true;
                        return r;
                    }
                }
                if (rc == RC_OK) rc = decode_block_data<TILES>();
            }
            first = false;
            if (rc != RC_EOB) break;  // stuck or error
            if (last_block) {
                rc = read_trailer(stored);
                break;
            }
            rc = RC_OK;
        }
        return finish(rc, stored);
    }
};
using Inflater = InflaterT<kLitBits>;

}  // namespace fdh
