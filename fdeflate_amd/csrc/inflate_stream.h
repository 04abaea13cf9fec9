// inflate_stream.h -- one zlib stream per wavefront: bit reader over an LDS-staged input
// window, LDS output ring with coalesced 16-B flushes + fused Adler-32, and the zlib/deflate
// state machine with the reference's one-shot semantics.
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
// unread stream bits) is a `stuck()` here with the same threshold on `left`.
#pragma once
#include "inflate_tables.h"

namespace fdh {

constexpr int kInChunk = 1024;              // bytes per coalesced input load (64 lanes x 16 B)
constexpr int kInRingDw = 2 * kInChunk / 4; // two chunks
constexpr int kOutRing = 4096;              // bytes, power of two
constexpr int kOutMask = kOutRing - 1;
constexpr int kFlushSlack = 64;

struct __attribute__((aligned(16))) WaveLds {
    uint32_t lit[kLitSize];
    uint32_t dist[kDistSize];
    uint32_t in_ring[kInRingDw];
    uint8_t out_ring[kOutRing];
    uint32_t cl[kClSize];
    CodeBook lit_cb;
    CodeBook dist_cb;
    CodeBook cl_cb;
    uint16_t lit_sorted[288];
    uint16_t dist_sorted[32];
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
    uint32_t flags;
};

struct StreamResult {
    uint32_t status, out_len, adler;
};

// ---------------------------------------------------------------------------------------
struct Inflater {
    WaveLds& L;
    const int lane;
    // ---- input window / bit reader (all uniform) ----
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
    bool want_adler;
    // ---- block state ----
    uint32_t eof_code, eof_mask, eof_bits;
    bool fixed_built;

    __device__ Inflater(WaveLds& l, int ln) : L(l), lane(ln) {}

    // ------------------------------------------------------------------ input window
    __device__ void load_chunk(uint32_t c) {
        uint64_t w0 = (uint64_t)c * kInChunk + (uint64_t)lane * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (w0 < win_bytes) {
            const uint8_t* p = base16 + w0;
            if (p >= buf_lo && p + 16 <= buf_hi) {
                v = *reinterpret_cast<const uint4*>(p);
            } else {
                uint32_t w[4] = {0, 0, 0, 0};
                for (int j = 0; j < 16; j++) {
                    if (p + j >= buf_lo && p + j < buf_hi) w[j >> 2] |= (uint32_t)p[j] << ((j & 3) * 8);
                }
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            uint64_t rem = win_bytes - w0;  // bytes of this stream in the lane's 16
            if (rem < 16) {                 // zero what lies past the end of the stream
                uint32_t r = (uint32_t)rem;
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    uint32_t lo = k * 4;
                    if (r <= lo) w[k] = 0;
                    else if (r < lo + 4) w[k] &= (1u << ((r - lo) * 8)) - 1;
                }
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        *reinterpret_cast<uint4*>(&L.in_ring[(c & 1) * (kInChunk / 4) + lane * 4]) = v;
    }

    __device__ void ensure_dw(uint32_t dw) {
        uint32_t c = dw / (kInChunk / 4);
        if (c + 1 >= loaded) {
            wave_sync();
            while (loaded <= c + 1) {
                if (loaded + 2 <= c) loaded = c;  // after a seek: skip chunks nobody will read
                load_chunk(loaded);
                loaded++;
            }
            wave_sync();
        }
    }

    __device__ void seek(uint64_t bitpos) {  // bitpos relative to the first stream byte
        uint64_t wbit = bitpos + (uint64_t)mis * 8;
        next_dw = (uint32_t)(wbit >> 5);
        uint32_t c = next_dw / (kInChunk / 4);
        if (!(c + 2 == loaded || c + 1 == loaded)) loaded = c;  // ring content unusable
        ensure_dw(next_dw);
        uint32_t sh = (uint32_t)wbit & 31;
        bb = (uint64_t)(uni(L.in_ring[next_dw & (kInRingDw - 1)]) >> sh);
        bbn = 32 - sh;
        next_dw++;
    }

    __device__ void refill() {  // afterwards bbn >= 33
        if (bbn <= 32) {
            ensure_dw(next_dw);
            uint32_t w = uni(L.in_ring[next_dw & (kInRingDw - 1)]);
            bb |= (uint64_t)w << bbn;
            bbn += 32;
            next_dw++;
        }
    }
    __device__ void consume(uint32_t n) {
        bb >>= n;
        bbn -= n;
        left -= n;
    }
    __device__ uint64_t consumed_bits() const { return (win_bytes - mis) * 8 - left; }

    // ------------------------------------------------------------------ output ring
    // Ring index of output position p is (p + gmis) & kOutMask, so 16-B lines of the global
    // slot are 16-B lines of the ring.
    __device__ void put_byte(uint32_t b) {
        if (lane == 0) L.out_ring[(opos + gmis) & kOutMask] = (uint8_t)b;
        opos++;
    }

    // Stores output [flushed, target) to global memory and folds it into the Adler-32.
    // target is opos rounded down to a line unless `final`.
    __device__ void flush(bool final) {
        wave_sync();
        uint32_t q_lo = flushed + gmis;
        uint32_t q_hi = opos + gmis;
        if (!final) q_hi &= ~15u;
        if (q_hi <= q_lo) return;
        for (uint32_t it = q_lo & ~15u; it < q_hi; it += kWave * 16) {
            uint32_t lq = it + lane * 16;       // this lane's line, q-space
            uint32_t blk_hi = min(q_hi, it + kWave * 16);
            uint32_t blk_lo = max(q_lo, it);
            uint32_t s = 0, t = 0;
            if (lq < blk_hi && lq + 16 > blk_lo) {
                uint32_t lo = (blk_lo > lq) ? blk_lo - lq : 0;
                uint32_t hi = (blk_hi < lq + 16) ? blk_hi - lq : 16;
                uint32_t W = blk_hi - lq;  // weight of byte j is W - j
                if (lo == 0 && hi == 16) {
                    uint4 v = *reinterpret_cast<const uint4*>(&L.out_ring[lq & kOutMask]);
                    *reinterpret_cast<uint4*>(out_al + lq) = v;
                    if (want_adler) {
                        s = bytesum4(v.x) + bytesum4(v.y) + bytesum4(v.z) + bytesum4(v.w);
                        uint32_t u = bytedot4(v.x, 0x03020100u, 0);
                        u = bytedot4(v.y, 0x07060504u, u);
                        u = bytedot4(v.z, 0x0b0a0908u, u);
                        u = bytedot4(v.w, 0x0f0e0d0cu, u);
                        t = W * s - u;
                    }
                } else {
                    for (uint32_t j = lo; j < hi; j++) {
                        uint32_t b = L.out_ring[(lq + j) & kOutMask];
                        out_al[lq + j] = (uint8_t)b;
                        s += b;
                        t += (W - j) * b;
                    }
                }
            }
            if (want_adler) {
                uint32_t S = wave_sum_u32(s);
                uint32_t T = wave_sum_u32(t);
                uint32_t Lb = blk_hi - blk_lo;
                adler_b = (uint32_t)(((uint64_t)adler_b + (uint64_t)Lb * adler_a + T) % kAdlerMod);
                adler_a = (adler_a + S) % kAdlerMod;
            }
        }
        flushed = q_hi - gmis;
        wave_sync();
    }

    __device__ void make_room(uint32_t n) {
        if (opos + n - flushed > (uint32_t)(kOutRing - kFlushSlack)) flush(false);
    }

    // LZ77 copy of n bytes from distance d (the copy itself, src/decompress.rs:792-829):
    // out[opos+k] = out[opos - d + (k mod d)], every source lies before opos.
    __device__ void copy_match(uint32_t n, uint32_t d) {
        make_room(n);
        wave_sync();
        // Sources older than opos + n - kOutRing may be overwritten during the copy; they are
        // all < flushed (make_room), so they come from global memory.
        int64_t ring_lo = (int64_t)opos + n - kOutRing;
        bool need_global = (int64_t)opos - (int64_t)d < ring_lo;
        if (need_global) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // our own flush stores
        for (uint32_t k = lane; k < n; k += kWave) {
            uint32_t r = (d >= n) ? k : (k % d);
            uint32_t src = opos - d + r;
            uint32_t b;
            if ((int64_t)src >= ring_lo) {
                b = L.out_ring[(src + gmis) & kOutMask];
            } else {
                // L1-bypassing load: the line may have been cached before our later stores
                b = __hip_atomic_load(out_al + gmis + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            L.out_ring[(opos + k + gmis) & kOutMask] = (uint8_t)b;
        }
        opos += n;
        wave_sync();
    }

    // Stored-block payload: n bytes from stream byte offset `src_byte` (src/decompress.rs:271-305).
    __device__ void copy_stored(const uint8_t* in, uint64_t src_byte, uint32_t n) {
        uint32_t done = 0;
        while (done < n) {
            uint32_t m = min(n - done, 1024u);
            make_room(m);
            wave_sync();
            for (uint32_t k = lane; k < m; k += kWave) {
                L.out_ring[(opos + k + gmis) & kOutMask] = in[src_byte + done + k];
            }
            opos += m;
            done += m;
            wave_sync();
        }
    }

    // ------------------------------------------------------------------ tables
    // CompressedBlock::build_tables, src/decompress.rs:561-606.  lens[0..320) in L.lens.
    __device__ uint32_t build_block_tables(uint32_t hlit) {
        if (uni(L.lens[256]) == 0) return ST_BAD_LITERAL_LENGTH_HUFFMAN_TREE;
        if (build_table<LitlenTraits, false>(L.lit, L.lens, (int)hlit, L.lit_cb, L.lit_sorted, lane) != BUILD_OK)
            return ST_BAD_CODE_LENGTH_HUFFMAN_TREE;  // sic, src/decompress.rs:579
        add_double_literals(L.lit, lane);
        // code of the end-of-block symbol, as the reference keeps it (eof_code/mask/bits)
        uint32_t l256 = uni(L.lens[256]);
        uint32_t rank = 0;
        for (int s = lane; s < 256; s += kWave) rank += (L.lens[s] == l256) ? 1u : 0u;
        rank = wave_sum_u32(rank);
        uint32_t cw = uni(L.lit_cb.first[l256]) + rank;
        eof_bits = l256;
        eof_mask = (1u << l256) - 1;
        eof_code = __brev(cw) >> (32 - l256);
        if (build_table<DistTraits, true>(L.dist, L.lens + 288, 32, L.dist_cb, L.dist_sorted, lane) != BUILD_OK)
            return ST_BAD_DISTANCE_HUFFMAN_TREE;
        return ST_OK;
    }

    __device__ void fill_fixed_lengths() {  // src/tables.rs:207-232
        for (int i = lane; i < 320; i += kWave) {
            uint8_t v = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5;
            L.lens[i] = v;
        }
        wave_sync();
    }

    // ------------------------------------------------------------------ the state machine
    __device__ StreamResult run(const StreamArgs& a) {
        // ---- set-up ----
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
        want_adler = true;
        fixed_built = false;
        eof_code = eof_mask = eof_bits = 0;
        seek(0);

        uint32_t status = ST_OK;
        bool done = false;
        bool last_block = false;

        // ---- zlib header, src/decompress.rs:226-244 ----
        {
            refill();
            if (left < 16) goto stuck;
            uint32_t b0 = (uint32_t)bb & 0xFF, b1 = ((uint32_t)bb >> 8) & 0xFF;
            if ((b0 & 0x0F) != 0x08 || (b0 & 0xF0) > 0x70 || (b1 & 0x20) != 0 || ((b0 << 8) | b1) % 31 != 0) {
                status = ST_BAD_ZLIB_HEADER;
                goto finish;
            }
            consume(16);
        }

        for (;;) {  // one iteration per block
            // ---- block header, src/decompress.rs:344-438 ----
            refill();
            if (left < 10) goto stuck;
            {
                uint32_t hdr = (uint32_t)bb & 7;
                last_block = (hdr & 1) != 0;
                uint32_t type = hdr >> 1;
                if (type == 0) {
                    uint32_t align = (uint32_t)((left - 3) & 7);
                    if (left < 35 + align) goto stuck;
                    consume(3 + align);
                    refill();
                    uint32_t len = (uint32_t)bb & 0xFFFF, nlen = ((uint32_t)bb >> 16) & 0xFFFF;
                    if (nlen != (~len & 0xFFFF)) {
                        status = ST_INVALID_UNCOMPRESSED_BLOCK_LENGTH;
                        goto finish;
                    }
                    consume(32);
                    // stored payload, src/decompress.rs:271-305
                    uint64_t avail = left >> 3;
                    uint32_t n = len;
                    if ((uint64_t)n > avail) n = (uint32_t)avail;
                    if (n > cap - opos) n = cap - opos;
                    uint64_t src_byte = consumed_bits() >> 3;
                    copy_stored(a.in, src_byte, n);
                    left -= (uint64_t)n * 8;
                    if (n < len) goto stuck;
                    seek(consumed_bits());
                    if (last_block) break;
                    continue;
                } else if (type == 1) {
                    consume(3);
                    if (((uint32_t)bb & 0x7F) == 0) {  // empty fixed block, :377-394
                        consume(7);
                        if (last_block) break;
                        continue;
                    }
                    if (!fixed_built) {
                        fill_fixed_lengths();
                        build_block_tables(288);
                        fixed_built = true;
                    }
                } else if (type == 2) {
                    if (left < 17) goto stuck;
                    uint32_t hlit = (((uint32_t)bb >> 3) & 31) + 257;
                    uint32_t hdist = (((uint32_t)bb >> 8) & 31) + 1;
                    uint32_t hclen = (((uint32_t)bb >> 13) & 15) + 4;
                    if (hlit > 286) {
                        status = ST_INVALID_HLIT;
                        goto finish;
                    }
                    if (hdist > 30) {
                        status = ST_INVALID_HDIST;
                        goto finish;
                    }
                    consume(17);
                    fixed_built = false;
                    // ---- code-length code lengths, src/decompress.rs:440-477 ----
                    refill();
                    if (left < 3 * hclen) goto stuck;
                    if (lane < 19) L.lens[320 - 19 + lane - 0] = 0;  // scratch: lens[301..320)
                    wave_sync();
                    for (uint32_t i = 0; i < hclen; i++) {
                        refill();
                        if (lane == 0) L.lens[301 + kClclOrder[i]] = (uint8_t)((uint32_t)bb & 7);
                        consume(3);
                    }
                    wave_sync();
                    if (build_table<ClTraits, false>(L.cl, L.lens + 301, 19, L.cl_cb, L.cl_sorted, lane) != BUILD_OK) {
                        status = ST_BAD_CODE_LENGTH_HUFFMAN_TREE;
                        goto finish;
                    }
                    // ---- literal/length + distance code lengths, src/decompress.rs:479-555 ----
                    uint32_t total = hlit + hdist, nread = 0;
                    // staged at lens[0..total), distance lengths moved to lens[288..) afterwards
                    while (nread < total) {
                        refill();
                        if (left < 7) goto stuck;
                        uint32_t e = uni(L.cl[(uint32_t)bb & 127]);
                        uint32_t nb = e & 15, sym = (e >> 8) & 0xFF;
                        if (sym <= 15) {
                            if (lane == 0) L.lens[nread] = (uint8_t)sym;
                            nread++;
                            consume(nb);
                        } else {
                            uint32_t base_rep = sym == 18 ? 11 : 3;
                            uint32_t extra = sym == 16 ? 2 : sym == 17 ? 3 : 7;
                            if (left < nb + extra) goto stuck;
                            uint32_t value = 0;
                            if (sym == 16) {
                                if (nread == 0) {
                                    status = ST_INVALID_CODE_LENGTH_REPEAT;
                                    goto finish;
                                }
                                wave_sync();
                                value = uni(L.lens[nread - 1]);
                            }
                            uint32_t rep = (((uint32_t)bb >> nb) & ((1u << extra) - 1)) + base_rep;
                            if (nread + rep > total) {
                                status = ST_INVALID_CODE_LENGTH_REPEAT;
                                goto finish;
                            }
                            for (uint32_t i = lane; i < rep; i += kWave) L.lens[nread + i] = (uint8_t)value;
                            nread += rep;
                            consume(nb + extra);
                        }
                    }
                    wave_sync();
                    {   // :541-549: distance lengths to [288, 288+hdist), zero the gaps
                        uint8_t dl = (lane < (int)hdist) ? L.lens[hlit + lane] : 0;
                        wave_sync();
                        for (uint32_t i = hlit + lane; i < 288; i += kWave) L.lens[i] = 0;
                        if (lane < 32) L.lens[288 + lane] = dl;
                        wave_sync();
                    }
                    status = build_block_tables(hlit);
                    if (status != ST_OK) goto finish;
                } else {
                    status = ST_INVALID_BLOCK_TYPE;
                    goto finish;
                }
            }

            // ---- compressed data, src/decompress.rs:611-1018 (careful-loop semantics) ----
            for (;;) {
                refill();
                if (opos == cap) {  // :838-840 then the trailing EOB peek :1009-1015
                    if (left >= 15 && ((uint32_t)bb & eof_mask) == eof_code) {
                        consume(eof_bits);
                        break;
                    }
                    goto stuck;
                }
                uint32_t e = uni(L.lit[(uint32_t)bb & (kLitSize - 1)]);
                uint32_t nb = e & 15, kind = (e >> 4) & 15;
                if (kind == K_LIT1) {
                    if (left < nb) goto stuck;
                    put_byte((e >> 8) & 0xFF);
                    consume(nb);
                    if (opos - flushed > (uint32_t)(kOutRing - kFlushSlack)) flush(false);
                    continue;
                }
                if (kind == K_LIT2) {
                    if (left < nb) goto stuck;
                    put_byte((e >> 8) & 0xFF);
                    consume(nb);
                    if (opos == cap) goto stuck;  // second literal queued, :866-876
                    put_byte((e >> 16) & 0xFF);
                    if (opos - flushed > (uint32_t)(kOutRing - kFlushSlack)) flush(false);
                    continue;
                }
                uint32_t len_base, len_extra, lcb;
                if (kind == K_LONG) {  // secondary-table symbols, :886-909
                    uint32_t sym;
                    long_decode(L.lit_cb, L.lit_sorted, bb, sym, lcb);
                    if (left < lcb) goto stuck;
                    if (sym < 256) {
                        consume(lcb);
                        put_byte(sym);
                        if (opos - flushed > (uint32_t)(kOutRing - kFlushSlack)) flush(false);
                        continue;
                    }
                    if (sym == 256) {
                        consume(lcb);
                        break;
                    }
                    len_base = kLenBase[sym - 257];
                    len_extra = kLenExtra[sym - 257];
                } else if (kind == K_EOB) {  // :912-917
                    if (left < nb) goto stuck;
                    consume(nb);
                    break;
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
                uint32_t de = uni(L.dist[(uint32_t)bb & (kDistSize - 1)]);
                uint32_t dkind = (de >> 4) & 15;
                uint32_t dbase, dextra, dcb;
                if (dkind == D_DIST) {
                    dbase = de >> 16;
                    dextra = (de >> 8) & 15;
                    dcb = de & 15;
                } else if (left0 > lcb + len_extra + kDistBits) {  // :932-933
                    if (dkind == D_INVALID) {
                        status = ST_INVALID_DISTANCE_CODE;
                        goto finish;
                    }
                    uint32_t dsym;
                    long_decode(L.dist_cb, L.dist_sorted, bb, dsym, dcb);
                    if (dsym >= 30) {
                        status = ST_INVALID_DISTANCE_CODE;
                        goto finish;
                    }
                    dbase = kDistBase[dsym];
                    dextra = kDistExtra[dsym];
                } else {
                    goto stuck;
                }
                uint32_t total_bits = lcb + len_extra + dcb + dextra;
                uint32_t dist = dbase + (uint32_t)((bb >> dcb) & ((1u << dextra) - 1));
                if (left0 < total_bits) goto stuck;
                if (dist > opos) {
                    status = ST_DISTANCE_TOO_FAR_BACK;
                    goto finish;
                }
                consume(dcb + dextra);
                uint32_t n = min(length, cap - opos);
                copy_match(n, dist);
                if (n < length) goto stuck;  // remainder queued, output full
            }
            if (last_block) break;
        }

        // ---- checksum, src/decompress.rs:306-326 ----
        {
            refill();
            uint32_t align = (uint32_t)(left & 7);
            if (left < 32 + align) goto stuck;
            consume(align);
            refill();
            uint32_t stored = __builtin_bswap32((uint32_t)bb);
            consume(32);
            flush(true);
            uint32_t adler = (adler_b << 16) | adler_a;
            if (!(a.flags & 1u) && stored != adler) {
                status = ST_WRONG_CHECKSUM;
                goto finish;
            }
            done = true;
            goto finish;
        }

    stuck:
        // src/decompress.rs:1126-1139: not done and no error -> OutputTooLarge if the slot is
        // full, InsufficientInput otherwise.
        status = (opos == cap) ? ST_OUTPUT_TOO_LARGE : ST_INSUFFICIENT_INPUT;
    finish:
        (void)done;
        flush(true);
        StreamResult r;
        r.status = status;
        r.out_len = opos;
        r.adler = (adler_b << 16) | adler_a;
        return r;
    }
};

}  // namespace fdh
