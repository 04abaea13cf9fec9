// deflate_general.hip -- the general encoder at level 1 (`compress_to_vec`) and in RLE mode
// (`compress_to_vec_rle`), bit-exact with the reference, in two kernels.
//
// Reference: Compressor src/compress/mod.rs:47-217 (level 1 = GreedyParser + HashTableMatchFinder
// :76, RLE = RleParser :107-123), parsers src/compress/parse/{mod,greedy,rle}.rs, match finders
// src/compress/matchfinder/{mod,hashtable}.rs, block writer src/compress/bitstream.rs:41-325,
// bit writer src/compress/bitwriter.rs.
//
// 1. deflate_parse_kernel -- ONE STREAM PER LANE.  LZ77 parsing with a hash table is order-dependent
//    inside a stream (every decision depends on what the table held, i.e. on all earlier decisions),
//    so the other lanes of a wavefront have nothing to do on the same stream that would not change
//    the bytes; the batch has thousands of independent streams instead.  Every active lane runs the
//    reference's parser on its own stream and records what it decided: the back-references (start,
//    length, distance) in stream order and, whenever the reference would write a block (16384
//    symbols, bitstream.rs / parse/mod.rs:87-93), where that block ends.  The work is bound by the
//    latency of dependent loads, not by arithmetic, so the launch uses FEWER lanes per wavefront and
//    more wavefronts when the batch is small (`lanes`), to have enough wavefronts in flight.
// 2. deflate_write_kernel -- ONE STREAM PER WAVEFRONT.  Everything the block writer does is a
//    function of the positions of the block: a position is the start of a back-reference, covered
//    by one, or a literal.  The wavefront walks the positions 64 at a time (the back-references of
//    the next 1024 positions are dropped into an LDS array of marks first), once to count symbol
//    frequencies with LDS atomics, then -- after the Huffman codes are built -- once more to emit:
//    per-lane bit counts are prefix-summed and every lane ORs its code into the LDS bit ring
//    (bit_ring.h).  The Adler-32 rides on the second walk.
//
// The Huffman construction follows Rust's BinaryHeap exactly as the reference uses it (see the
// oracle's section header for what is pinned and what is not); the order-dependent part (heap
// merges) runs on one lane per tree -- the literal/length and the distance tree side by side on
// lanes 0 and 1 -- the rest (compaction, code assignment, header) on the whole wavefront.
#include "device_common.h"
#include "bit_ring.h"

namespace fdh {

// ---- constant tables (RFC 1951; reference src/tables.rs:28-88, data) -----------------------
__device__ static const uint8_t kGDistLookup[16] = {0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7};

// length - 3 -> (symbol, extra bits): LENGTH_TO_SYMBOL / LENGTH_TO_LEN_EXTRA (tables.rs:28-55)
__device__ __forceinline__ void g_length_symbol(uint32_t length, uint32_t& sym, uint32_t& extra) {
    const uint32_t l = length - 3;
    if (l == 255) {  // 258
        sym = 285;
        extra = 0;
        return;
    }
    if (l < 8) {
        sym = 257 + l;
        extra = 0;
        return;
    }
    const uint32_t e = 29 - (uint32_t)__clz(l);  // floor(log2(l)) - 2
    sym = 257 + 4 * e + 4 + ((l >> e) & 3);
    extra = e;
}

constexpr uint32_t kGHashSize = 1u << 16;
constexpr uint32_t kGBlockSymbols = 16384;

// What the parser hands to the block writer.  Back-references never overlap and are at least 4
// bytes long, so a stream of `len` bytes has at most len / 4 of them; a block that is not the last
// one holds 16384 symbols of at least one byte each.  The per-stream slices of the two record
// arrays are placed with these bounds straight from in_off (no prefix sum needed).
struct GMatchRec {
    uint32_t start;  // position in the stream
    uint32_t info;   // length | dist_sym << 9 | (distance - 1) << 14
};
struct GBlockRec {
    uint32_t end_pos;    // the block covers positions [end of the previous block, end_pos)
    uint32_t match_end;  // and back-references [match_end of the previous block, match_end)
    uint32_t flags;
    uint32_t pad;
};
constexpr uint32_t kGBlockEof = 1;         // BFINAL
constexpr uint32_t kGBlockEmptyFixed = 2;  // the empty fixed block of compress/mod.rs:234-238
__host__ __device__ inline uint64_t g_match_slice(uint64_t rel_off, uint64_t i) { return rel_off / 4 + 2 * i; }
__host__ __device__ inline uint64_t g_block_slice(uint64_t rel_off, uint64_t i) { return rel_off / kGBlockSymbols + 4 * i; }

__device__ __forceinline__ uint64_t g_load64(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ uint32_t g_hash(uint64_t v) { return (uint32_t)((11400714785074694791ull * v) >> 40) & (kGHashSize - 1); }

#ifdef FDH_DEBUG_GEN
__device__ uint64_t g_gen_t[8][32768];
__device__ uint32_t g_gen_w[12][32768];  // block writer: walk 1, prepare, merges (+ depths), limit / codes, header, walk 2, -, whole
#define GT0(k) const uint64_t _t##k = __builtin_readcyclecounter()
#define GT1(k) tacc[k] += __builtin_readcyclecounter() - _t##k
#else
#define GT0(k)
#define GT1(k)
#endif

struct GMatch {
    uint32_t length, distance;
    uint64_t start;
    __device__ uint64_t end() const { return start + length; }
};

// ============================ kernel 1: the parser, one stream per lane ============================

struct GParser {
    uint32_t* hash;     // this lane's table (level 1)
    GMatchRec* mrec;    // this stream's slices
    GBlockRec* brec;
    uint32_t nmatch, nblock;
    uint32_t nsym;      // symbols (literal runs + back-references) of the open block
    uint64_t ip, last_match, last_block_end;
    uint32_t last_index;
    GMatch m;
#ifdef FDH_DEBUG_GEN
    uint64_t tacc[8];
#endif

    // distance_to_dist_sym (bitstream.rs:16-27)
    __device__ static uint32_t dist_sym_of(uint32_t distance) {
        if (distance <= 16) return kGDistLookup[distance - 1];
        const uint32_t d1 = distance - 1, l = 31 - (uint32_t)__clz(d1);  // two symbols per power of two
        return 2 * l + ((d1 >> (l - 1)) & 1);
    }

    // Equal bytes walking backwards, data[a_end - 1 - i] == data[b_end - 1 - i] for i < limit
    // (limit <= min(a_end, b_end)), eight at a time: the reference's byte loops (matchfinder/mod.rs
    // :66-72, parse/mod.rs:72-81) give the same count, a dependent byte load per step is what they
    // would cost here.
    __device__ static uint32_t back_equal(const uint8_t* data, uint64_t a_end, uint64_t b_end, uint32_t limit) {
        uint32_t n = 0;
        while (n + 32 <= limit) {  // four pairs of loads in flight
            const uint8_t *pa = data + a_end - n, *pb = data + b_end - n;
            const uint64_t x0 = g_load64(pa - 8) ^ g_load64(pb - 8), x1 = g_load64(pa - 16) ^ g_load64(pb - 16);
            const uint64_t x2 = g_load64(pa - 24) ^ g_load64(pb - 24), x3 = g_load64(pa - 32) ^ g_load64(pb - 32);
            if (x0 | x1 | x2 | x3) {
                return n + (x0 ? (uint32_t)__builtin_clzll(x0) >> 3
                            : x1 ? 8 + ((uint32_t)__builtin_clzll(x1) >> 3)
                            : x2 ? 16 + ((uint32_t)__builtin_clzll(x2) >> 3) : 24 + ((uint32_t)__builtin_clzll(x3) >> 3));
            }
            n += 32;
        }
        while (n + 8 <= limit) {
            const uint64_t x = g_load64(data + a_end - n - 8) ^ g_load64(data + b_end - n - 8);
            if (x) return n + ((uint32_t)__builtin_clzll(x) >> 3);
            n += 8;
        }
        if (n < limit) {
            if (a_end - n >= 8 && b_end - n >= 8) {
                const uint64_t x = g_load64(data + a_end - n - 8) ^ g_load64(data + b_end - n - 8);
                return n + min(x ? (uint32_t)__builtin_clzll(x) >> 3 : 8u, limit - n);
            }
            while (n < limit && data[a_end - n - 1] == data[b_end - n - 1]) n++;
        }
        return n;
    }
    // Equal bytes walking forwards, a[i] == b[i] for i < limit; `wide_tail`: 8 bytes may be read at
    // any offset below limit.
    __device__ static uint32_t fwd_equal(const uint8_t* a, const uint8_t* b, uint32_t limit, bool wide_tail) {
        uint32_t k = 0;
        for (; k + 32 <= limit; k += 32) {  // four pairs of loads in flight
            const uint64_t x0 = g_load64(a + k) ^ g_load64(b + k), x1 = g_load64(a + k + 8) ^ g_load64(b + k + 8);
            const uint64_t x2 = g_load64(a + k + 16) ^ g_load64(b + k + 16), x3 = g_load64(a + k + 24) ^ g_load64(b + k + 24);
            if (x0 | x1 | x2 | x3) {
                return k + (x0 ? (uint32_t)__builtin_ctzll(x0) >> 3
                            : x1 ? 8 + ((uint32_t)__builtin_ctzll(x1) >> 3)
                            : x2 ? 16 + ((uint32_t)__builtin_ctzll(x2) >> 3) : 24 + ((uint32_t)__builtin_ctzll(x3) >> 3));
            }
        }
        for (; k + 8 <= limit; k += 8) {
            const uint64_t x = g_load64(a + k) ^ g_load64(b + k);
            if (x) return k + ((uint32_t)__builtin_ctzll(x) >> 3);
        }
        if (k < limit) {
            if (wide_tail) {
                const uint64_t x = g_load64(a + k) ^ g_load64(b + k);
                return k + min(x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u, limit - k);
            }
            while (k < limit && a[k] == b[k]) k++;
        }
        return k;
    }

    // match_length::<true> (matchfinder/mod.rs:51-111)
    __device__ static void match_length8(uint64_t value, const uint8_t* data, uint64_t len, uint64_t anchor, uint64_t ip,
                                         uint64_t prev_index, uint32_t& out_len, uint64_t& out_start) {
        if (value != g_load64(data + prev_index)) {
            out_len = 0;
            out_start = ip;
            return;
        }
        uint64_t length = 8;
        {   // backwards while length < 258 && ip > anchor && prev_index > 0 && the bytes in front agree
            const uint32_t n = back_equal(data, ip, prev_index, (uint32_t)min(min((uint64_t)250, ip - anchor), prev_index));
            length += n;
            ip -= n;
            prev_index -= n;
        }
        uint64_t slice = len - ip - length;
        if (slice > 258 - length) slice = 258 - length;
        length += fwd_equal(data + ip + length, data + prev_index + length, (uint32_t)slice, ip + length + slice + 8 <= len);
        out_len = (uint32_t)length;
        out_start = ip;
    }

    // rle_match (matchfinder/mod.rs:113-145)
    __device__ GMatch rle_match(const uint8_t* data, uint64_t len) const {
        const uint8_t value = data[ip];
        GMatch r{4, 1, ip + 1};
        uint64_t min_start = max((uint64_t)1, last_match);
        const uint64_t e = r.end();
        if (e > 258) min_start = max(min_start, e - 258);
        const uint64_t v8 = 0x0101010101010101ull * value;
        {   // while r.start > min_start && data[r.start - 2] == value: one byte further back
            const uint32_t limit = r.start > min_start ? (uint32_t)(r.start - min_start) : 0u;
            const uint64_t end = r.start - 1;  // the bytes tested are data[end - 1], data[end - 2], ...
            uint32_t n = 0;
            bool stop = false;
            while (!stop && n + 32 <= limit) {  // four loads in flight
                const uint8_t* pe = data + end - n;
                const uint64_t x0 = g_load64(pe - 8) ^ v8, x1 = g_load64(pe - 16) ^ v8, x2 = g_load64(pe - 24) ^ v8, x3 = g_load64(pe - 32) ^ v8;
                if (x0 | x1 | x2 | x3) {
                    n += x0 ? (uint32_t)__builtin_clzll(x0) >> 3
                         : x1 ? 8 + ((uint32_t)__builtin_clzll(x1) >> 3)
                         : x2 ? 16 + ((uint32_t)__builtin_clzll(x2) >> 3) : 24 + ((uint32_t)__builtin_clzll(x3) >> 3);
                    stop = true;
                } else {
                    n += 32;
                }
            }
            while (!stop && n + 8 <= limit) {
                const uint64_t x = g_load64(data + end - n - 8) ^ v8;
                if (x) {
                    n += (uint32_t)__builtin_clzll(x) >> 3;
                    stop = true;
                } else {
                    n += 8;
                }
            }
            if (!stop && n < limit) {
                if (end - n >= 8) {
                    const uint64_t x = g_load64(data + end - n - 8) ^ v8;
                    n += min(x ? (uint32_t)__builtin_clzll(x) >> 3 : 8u, limit - n);
                } else {
                    while (n < limit && data[end - n - 1] == value) n++;
                }
            }
            r.start -= n;
            r.length += n;
        }
        uint64_t n = len - r.end();
        if (n > 258 - r.length) n = 258 - r.length;
        const uint8_t* p = data + r.end();
        uint64_t k = 0;
        for (; k + 32 <= n; k += 32) {  // four loads in flight
            const uint64_t c0 = g_load64(p + k) ^ v8, c1 = g_load64(p + k + 8) ^ v8, c2 = g_load64(p + k + 16) ^ v8, c3 = g_load64(p + k + 24) ^ v8;
            if (c0 | c1 | c2 | c3) {
                const uint32_t m = c0 ? (uint32_t)__builtin_ctzll(c0) >> 3
                                   : c1 ? 8 + ((uint32_t)__builtin_ctzll(c1) >> 3)
                                   : c2 ? 16 + ((uint32_t)__builtin_ctzll(c2) >> 3) : 24 + ((uint32_t)__builtin_ctzll(c3) >> 3);
                r.length += m;
                return r;
            }
            r.length += 32;
        }
        for (; k + 8 <= n; k += 8) {
            const uint64_t c = g_load64(p + k);
            if (c != v8) {
                r.length += (uint32_t)(__builtin_ctzll(c ^ v8) / 8);
                return r;
            }
            r.length += 8;
        }
        if (k < n) {
            if (r.end() + 8 <= len) {  // (r.end() has moved with r.length: the next 8 bytes are inside the buffer)
                const uint64_t x = g_load64(data + r.end()) ^ v8;
                r.length += min(x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u, (uint32_t)(n - k));
            } else {
                for (; k < n; k++) {
                    if (p[k] != value) break;
                    r.length++;
                }
            }
        }
        return r;
    }

    // ParserInner::get_match (parse/mod.rs:58-85) with HashTableMatchFinder::get_and_insert
    // (hashtable.rs:16-50) / NullMatchFinder
    template <bool RLE>
    __device__ GMatch get_match(const uint8_t* data, uint64_t len, uint32_t base_index, bool fizzle) {
        const uint64_t current = g_load64(data + ip);
        if ((uint32_t)current == (uint32_t)(current >> 8)) {
            const GMatch r = rle_match(data, len);
            ip = r.end() - 3;
            return r;
        }
        GMatch r{0, 0, 0};
        if (!RLE) {
            const uint64_t anchor = fizzle ? ip : last_match;
            const uint32_t sub = (uint32_t)ip > 32768 ? (uint32_t)ip - 32768 : 0;
            const uint32_t min_offset = max(base_index + sub, 1u);
            const uint32_t h = g_hash(current);
            const uint32_t offset = hash[h];
            hash[h] = (uint32_t)ip + base_index;
            if (offset >= min_offset) {
                uint32_t l;
                uint64_t st;
                match_length8(current, data, len, anchor, ip, (uint64_t)(offset - base_index), l, st);
                if (l >= 8) r = GMatch{l, (uint32_t)(ip - (uint64_t)(offset - base_index)), st};
            }
            if (fizzle && r.length != 0) {  // parse/mod.rs:72-81: take bytes in front of the match while they agree
                const uint64_t room = r.start > (uint64_t)r.distance + 1 ? r.start - r.distance - 1 : 0;  // (never down to byte 0)
                const uint64_t lim = min(min((uint64_t)(258 - r.length), r.start - last_match), room);
                const uint32_t n = back_equal(data, r.start, r.start - r.distance, (uint32_t)lim);
                r.length += n;
                r.start -= n;
            }
        }
        ip++;
        return r;
    }

    // ParserInner::advance_to_match (parse/mod.rs:88-102).  A step that finds nothing costs two
    // dependent memory round trips (the 8 bytes at ip, then the table entry they hash to) and where
    // the next step looks does not depend on either, so the steps are taken kGroup at a time along
    // the path the scan follows while it finds nothing: all the data loads, then all the table loads,
    // then the decisions one by one in order.  The loads of a group are issued before its stores, so
    // an entry that an earlier step of the same group would have replaced is patched from registers;
    // the steps behind a hit are dropped without having stored anything.
    static constexpr int kGroup = 16;
    template <bool RLE>
    __device__ GMatch advance_to_match(const uint8_t* data, uint64_t len, uint32_t base_index, uint64_t max_ip) {
        while (ip < max_ip) {
            uint64_t p[kGroup], cur[kGroup];
            uint32_t h[kGroup], off[kGroup];
            uint64_t q = ip;
#pragma unroll
            for (int k = 0; k < kGroup; k++) {
                p[k] = q;
                cur[k] = g_load64(data + (q < max_ip ? q : ip));
                q++;
                q += (q - last_match) >> 5;  // skip_ahead_shift = 5 (compress/mod.rs:76, :114)
            }
            if (!RLE) {
#pragma unroll
                for (int k = 0; k < kGroup; k++) {
                    h[k] = g_hash(cur[k]);
                    off[k] = hash[h[k]];
                }
            }
            // the unrolled part only finds the first step where something happens (and stores the
            // table entries of the steps in front of it); what happens there is handled once, below
            // -- one copy of the match code instead of kGroup keeps the kernel inside the
            // instruction cache
            int event = 0;  // 1 end of the scan range, 2 a run (RLE pattern), 3 a candidate from the table
            uint64_t ev_p = 0, ev_cur = 0;
            uint32_t ev_off = 0;
#pragma unroll
            for (int k = 0; k < kGroup; k++) {
                if (event == 0) {
                    ev_p = p[k];
                    ev_cur = cur[k];
                    if (p[k] >= max_ip) {
                        event = 1;
                    } else if ((uint32_t)cur[k] == (uint32_t)(cur[k] >> 8)) {
                        event = 2;
                    } else if (!RLE) {  // HashTableMatchFinder::get_and_insert (hashtable.rs:16-50)
                        uint32_t offset = off[k];
#pragma unroll
                        for (int j = 0; j < k; j++)
                            if (h[j] == h[k]) offset = (uint32_t)p[j] + base_index;
                        hash[h[k]] = (uint32_t)p[k] + base_index;
                        const uint32_t sub = (uint32_t)p[k] > 32768 ? (uint32_t)p[k] - 32768 : 0;
                        if (offset >= max(base_index + sub, 1u)) {
                            event = 3;
                            ev_off = offset;
                        }
                    }
                }
            }
            if (event == 0) {
                ip = q;
                continue;
            }
            ip = ev_p;
            if (event == 1) return GMatch{0, 0, 0};
            // ParserInner::get_match (parse/mod.rs:58-85), fizzle = false
            if (event == 2) {
                GT0(1);
                const GMatch r = rle_match(data, len);
                GT1(1);
                ip = r.end() - 3;
                return r;
            }
            uint32_t l;
            uint64_t st;
            GT0(2);
            match_length8(ev_cur, data, len, last_match, ip, (uint64_t)(ev_off - base_index), l, st);
            GT1(2);
            ip++;
            if (l >= 8) return GMatch{l, (uint32_t)(ev_p - (uint64_t)(ev_off - base_index)), st};
            ip += (ip - last_match) >> 5;
        }
        return GMatch{0, 0, 0};
    }

    // ParserInner::advance (parse/mod.rs:105-114): the positions covered by a match go into the
    // table; four at a time (the loads are independent, the stores stay in position order)
    template <bool RLE>
    __device__ void advance(const uint8_t* data, uint64_t len, uint32_t base_index, uint64_t end) {
        if (!RLE) {
            const uint64_t stop = min(end, len - 8);
            uint64_t j = ip;
            for (; j + 4 <= stop; j += 4) {
                const uint64_t v0 = g_load64(data + j), v1 = g_load64(data + j + 1), v2 = g_load64(data + j + 2), v3 = g_load64(data + j + 3);
                hash[g_hash(v0)] = base_index + (uint32_t)j;
                hash[g_hash(v1)] = base_index + (uint32_t)j + 1;
                hash[g_hash(v2)] = base_index + (uint32_t)j + 2;
                hash[g_hash(v3)] = base_index + (uint32_t)j + 3;
            }
            for (; j < stop; j++) hash[g_hash(g_load64(data + j))] = base_index + (uint32_t)j;
        }
        ip = max(ip, end);
    }

    // where write_block (bitstream.rs:41-195) would run: the block ends at `end` (absolute)
    __device__ void record_block(uint32_t end, uint32_t flags) {
        brec[nblock] = GBlockRec{end, nmatch, flags, 0};
        nblock++;
    }

    __device__ void insert_match(uint32_t base_index, const GMatch& r) {
        if (r.start > last_match) nsym++;  // the literal run in front of it
        mrec[nmatch] = GMatchRec{base_index + (uint32_t)r.start, r.length | (dist_sym_of(r.distance) << 9) | ((r.distance - 1) << 14)};
        nmatch++;
        nsym++;
        last_match = r.end();
    }

    __device__ void write_block_if_ready(uint64_t len, uint32_t base_index, bool finish) {
        if (nsym >= kGBlockSymbols) {
            record_block(base_index + (uint32_t)last_match, finish && last_match == len ? kGBlockEof : 0);
            nsym = 0;
            last_block_end = last_match;
        }
    }

    __device__ uint64_t start_compress(uint32_t base_index, uint64_t start) {
        const uint32_t delta = base_index - last_index;
        ip -= delta;
        last_match -= delta;
        last_block_end = start;
        last_index = base_index;
        return delta;
    }

    __device__ uint64_t end_compress(uint64_t len, uint32_t base_index, uint64_t start, bool finish) {
        if (finish && (nsym != 0 || last_match < len)) {
            ip = min(ip, len);
            if (last_match < len) {  // the closing literal run
                nsym++;
                ip = len;
                last_match = len;
            }
            record_block(base_index + (uint32_t)len, kGBlockEof);
            nsym = 0;
            last_block_end = ip;
        }
        return last_block_end - start;
    }

    // CompressorInner::compress (compress/mod.rs:226-290) -> GreedyParser::compress
    // (parse/greedy.rs:27-91) / RleParser::compress (parse/rle.rs:22-47)
    template <bool RLE>
    __device__ uint64_t compress(const uint8_t* data, uint64_t len, uint32_t base_index, uint64_t start, bool finish) {
        if (finish && len == start) {  // :234-238
            record_block(base_index + (uint32_t)len, kGBlockEmptyFixed);
            return 0;
        }
        const uint64_t delta = start_compress(base_index, start);
        if (!RLE && m.length != 0) m.start -= delta;
        const uint64_t lookahead = finish ? 7 : (RLE ? 258 : 258 + 8);
        const uint64_t max_ip = len > lookahead ? len - lookahead : 0;
        if (RLE) {
            for (;;) {
                const GMatch r = advance_to_match<true>(data, len, base_index, max_ip);
                if (r.length == 0) break;
                ip = r.end();
                insert_match(base_index, r);
                write_block_if_ready(len, base_index, finish);
            }
        } else {
            for (;;) {
                if (m.length == 0) {
                    GT0(0);
                    m = advance_to_match<false>(data, len, base_index, max_ip);
                    GT1(0);
                    if (m.length == 0) break;
                }
                GT0(3);
                advance<false>(data, len, base_index, m.end());
                GT1(3);
                GMatch m2{0, 0, 0};
                if (ip < max_ip) {
                    GT0(4);
                    m2 = get_match<false>(data, len, base_index, true);
                    GT1(4);
                } else if (!finish) {
                    break;
                }
                if (m2.length == 0 || m2.start > m.start + 1) {
                    insert_match(base_index, m);
                    write_block_if_ready(len, base_index, finish);
                    if (m2.length != 0 && m2.start < last_match) {
                        m2.length -= (uint32_t)(last_match - m2.start);
                        m2.start = last_match;
                        if (m2.length < 4) m2 = GMatch{0, 0, 0};
                    }
                }
                m = m2;
            }
        }
        return end_compress(len, base_index, start, finish);
    }
};

struct GParseArgs {
    const uint8_t* in;
    const uint64_t* in_off;
    uint64_t n;
    uint32_t* hash;      // kGHashSize entries per resident lane (level 1)
    GMatchRec* matches;  // slices by g_match_slice
    GBlockRec* blocks;   // slices by g_block_slice
    uint32_t* nblocks;   // per stream; 0xFFFFFFFF = not supported (longer than 1 GiB)
    uint32_t lanes;      // streams per wavefront (1..64)
};

template <bool RLE>
__global__ __launch_bounds__(kWave) void deflate_parse_kernel(GParseArgs a) {
    const uint32_t lane = threadIdx.x;
    const uint32_t L = a.lanes;
    const uint64_t in0 = a.in_off[0];
    uint32_t* wave_hash = RLE ? nullptr : a.hash + (uint64_t)blockIdx.x * L * kGHashSize;
    for (uint64_t sid0 = (uint64_t)blockIdx.x * L; sid0 < a.n; sid0 += (uint64_t)gridDim.x * L) {
        if (!RLE) {
            // HashTableMatchFinder::new (hashtable.rs:10-14): the tables of the wavefront's streams
            // are cleared by the whole wavefront (coalesced 16-B stores), not lane by lane
            uint4* t = reinterpret_cast<uint4*>(wave_hash);
            for (uint32_t i = lane; i < L * (kGHashSize / 4); i += kWave) t[i] = make_uint4(0, 0, 0, 0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const uint64_t sid = sid0 + lane;
        if (lane >= L || sid >= a.n) continue;
        const uint64_t off = a.in_off[sid];
        const uint8_t* input = a.in + off;
        const uint64_t len = a.in_off[sid + 1] - off;
        if (len > (1ull << 30)) {  // write_data splits above 1 GiB (compress/mod.rs:130-136): not supported
            a.nblocks[sid] = 0xFFFFFFFFu;
            continue;
        }
        GParser ps;
        ps.hash = RLE ? nullptr : wave_hash + (uint64_t)lane * kGHashSize;
        ps.mrec = a.matches + g_match_slice(off - in0, sid);
        ps.brec = a.blocks + g_block_slice(off - in0, sid);
        ps.nmatch = ps.nblock = ps.nsym = 0;
        ps.ip = ps.last_match = ps.last_block_end = 0;
        ps.last_index = 0;
        ps.m = GMatch{0, 0, 0};
#ifdef FDH_DEBUG_GEN
        for (int k = 0; k < 8; k++) ps.tacc[k] = 0;
        const uint64_t t_all = __builtin_readcyclecounter();
#endif
        const uint64_t window = RLE ? 1 : 32768;
        // Compressor::write_data (compress/mod.rs:126-159, no buffered input), then Compressor::finish
        // (:194-214) over the kept tail input.data = data[start..]; one copy of the parser's code
        uint64_t written = 0, start = 0;
#pragma nounroll
        for (int pass = 0; pass < 2; pass++) {
            if (pass) start = written > window ? written - window : 0;
            const uint64_t r = ps.compress<RLE>(input + start, len - start, (uint32_t)start, pass ? written - start : 0, pass != 0);
            if (!pass) written = r;
        }
        a.nblocks[sid] = ps.nblock;
#ifdef FDH_DEBUG_GEN
        if (sid < 32768) {
            ps.tacc[7] = __builtin_readcyclecounter() - t_all;
            for (int k = 0; k < 8; k++) g_gen_t[k][sid] = ps.tacc[k];
        }
#endif
    }
}

// ============================ kernel 2: the block writer, one stream per wavefront ============================

constexpr uint32_t kGChunk = 1024;  // positions per refill of the marks
constexpr int kGRingDw = 512;       // 2 KiB bit ring: a step of the walk adds at most 64 x 118 bits
using BitRing = BitRingT<kGRingDw>;
constexpr uint32_t kEncTileBudget = BitRing::kEncTileBudget, kEncRingBits = BitRing::kEncRingBits;

// Scratch of one Huffman construction (build_huffman_tree, bitstream.rs:198-325).
template <int CAP>
struct GHuffScratch {
    uint64_t heap[CAP];        // frequency << 32 | node index
    uint16_t in_left[CAP], in_right[CAP];
    uint8_t depth[CAP];        // of the internal nodes
    uint8_t lengths[CAP];
    uint32_t counts[16], first[16];
    uint32_t heap_len, used, max_length, pad;
};

struct GHuffView {  // the same code runs on two lanes over different trees
    uint32_t* freq;
    uint64_t* heap;
    uint16_t *in_left, *in_right;
    uint8_t *depth, *lengths;
    uint32_t *counts, *hdr;  // hdr[0] heap_len, [1] used, [2] max_length
    uint32_t n, limit;
};

template <int CAP>
__device__ __forceinline__ GHuffView g_view(GHuffScratch<CAP>& s, uint32_t* freq, uint32_t n, uint32_t limit) {
    return GHuffView{freq, s.heap, s.in_left, s.in_right, s.depth, s.lengths, s.counts, &s.heap_len, n, limit};
}

struct GWriteLds {
    uint32_t ring[kGRingDw];
    uint32_t freq[288], dfreq[32], clfreq[20];
    uint32_t cl[288], dcl[32], clcl[20];  // code | length << 16
    uint32_t dmeta[32];                   // distance base | extra bits << 16
    union {
        alignas(16) uint32_t marks[kGChunk];  // info of the back-reference that starts at a position of the chunk, else 0
        struct {
            GHuffScratch<288> big;
            GHuffScratch<32> small;
        } h;
    };
};

// Ord of the heap items is `other.0.cmp(&self.0)` on the frequency only: a <= b  <=>  a.f >= b.f;
// ties are decided by the std::collections::BinaryHeap algorithms restated below.
__device__ __forceinline__ uint32_t hf(uint64_t item) { return (uint32_t)(item >> 32); }

__device__ void g_sift_down_range(uint64_t* d, uint32_t pos, uint32_t end) {
    const uint64_t elem = d[pos];
    uint32_t hole = pos, child = 2 * hole + 1;
    const uint32_t lim = end >= 2 ? end - 2 : 0;
    while (child <= lim) {
        const uint64_t c0 = d[child], c1 = d[child + 1];
        const bool right = hf(c0) >= hf(c1);  // d[child] <= d[child + 1]
        const uint64_t c = right ? c1 : c0;
        child += right;
        if (hf(elem) <= hf(c)) {  // elem >= d[child]
            d[hole] = elem;
            return;
        }
        d[hole] = c;
        hole = child;
        child = 2 * hole + 1;
    }
    if (child == end - 1) {
        const uint64_t c = d[child];
        if (hf(elem) > hf(c)) {  // elem < d[child]
            d[hole] = c;
            hole = child;
        }
    }
    d[hole] = elem;
}

// BinaryHeap::pop: the last item goes to the root, sift_down_to_bottom(0), then sift_up
__device__ uint64_t g_heap_pop(uint64_t* d, uint32_t& len) {
    const uint64_t last = d[len - 1];
    len--;
    if (len == 0) return last;
    const uint64_t top = d[0];
    const uint32_t end = len;
    uint32_t hole = 0, child = 1;
    const uint32_t lim = end >= 2 ? end - 2 : 0;
    while (child <= lim) {
        const uint64_t c0 = d[child], c1 = d[child + 1];
        const bool right = hf(c0) >= hf(c1);
        d[hole] = right ? c1 : c0;
        child += right;
        hole = child;
        child = 2 * hole + 1;
    }
    if (child == end - 1) {
        d[hole] = d[child];
        hole = child;
    }
    while (hole > 0) {  // sift_up(0, hole)
        const uint32_t parent = (hole - 1) / 2;
        const uint64_t p = d[parent];
        if (hf(last) >= hf(p)) break;  // elem <= d[parent]
        d[hole] = p;
        hole = parent;
    }
    d[hole] = last;
    return top;
}

// Whole wavefront: clear the lengths, put the used symbols on the heap array in index order
// (bitstream.rs:200-222: `for (i, &f) in frequencies.iter().enumerate()` pushes in that order).
template <bool PACKED>
__device__ __forceinline__ void g_huff_prepare(const GHuffView& v, int lane) {
    uint32_t* const d32 = reinterpret_cast<uint32_t*>(v.heap);
    uint32_t hl = 0;
    for (uint32_t base = 0; base < v.n; base += kWave) {
        const uint32_t i = base + lane;
        const uint32_t f = i < v.n ? v.freq[i] : 0;
        if (i < v.n) v.lengths[i] = 0;
        const uint64_t m = __ballot(f > 0);
        if (f > 0) {
            const uint32_t at = hl + (uint32_t)__popcll(m & lanemask_lt(lane));
            if (PACKED) d32[at] = (f << 10) | i;
            else v.heap[at] = ((uint64_t)f << 32) | i;
        }
        hl += (uint32_t)__popcll(m);
    }
    if (lane == 0) {
        v.hdr[0] = hl;
        v.hdr[1] = hl;
        v.hdr[2] = 0;
    }
}

__device__ void g_huff_depths(const GHuffView& v, uint32_t ni);

// One lane: the order-dependent part -- heapify, merge, depths, length limiting.
__device__ void g_huff_serial(const GHuffView& v) {
    const uint32_t N = v.n;
    uint32_t hl = v.hdr[0];
    if (hl <= 1) {  // :206-213 nothing or a single symbol (length 1)
        if (hl == 1) {
            v.lengths[(uint32_t)v.heap[0] & 0xFFFF] = 1;
            v.hdr[2] = 1;
        }
        return;
    }
    for (uint32_t k = hl / 2; k > 0;) {  // BinaryHeap::from(vec): rebuild
        k--;
        g_sift_down_range(v.heap, k, hl);
    }
    uint32_t ni = 0;
    while (hl > 1) {  // :236-244
        const uint64_t a = g_heap_pop(v.heap, hl);
        const uint64_t b = v.heap[0];
        v.in_left[ni] = (uint16_t)a;
        v.in_right[ni] = (uint16_t)b;
        ni++;
        v.heap[0] = ((uint64_t)(hf(a) + hf(b)) << 32) | (ni + N - 1);
        g_sift_down_range(v.heap, 0, hl);  // PeekMut::drop
    }
    g_huff_depths(v, ni);
}

// One lane: :247-259 depth of every leaf; a node is always created after its children, so one pass from
// the root (the last internal node) down the creation order visits parents first
__device__ void g_huff_depths(const GHuffView& v, uint32_t ni) {
    const uint32_t N = v.n;
    uint32_t max_length = 0;
    v.depth[ni - 1] = 0;
    for (uint32_t k = ni; k > 0;) {
        k--;
        const uint32_t dch = (uint32_t)v.depth[k] + 1;
        const uint32_t l = v.in_left[k], r = v.in_right[k];
        if (l < N) {
            v.lengths[l] = (uint8_t)dch;
            max_length = max(max_length, dch);
        } else {
            v.depth[l - N] = (uint8_t)dch;
        }
        if (r < N) {
            v.lengths[r] = (uint8_t)dch;
            max_length = max(max_length, dch);
        } else {
            v.depth[r - N] = (uint8_t)dch;
        }
    }
    v.hdr[2] = max_length;
}

// ---- The same heap with one dword per item: frequency << 10 | node index. ----
// With 64-bit items on one lane the sifts were 42 % of the block writer's time (measured by running them
// twice).  Node indices stay below 572 and the caller checks that the block's symbol count stays below 2^22
// (else the 64-bit version above runs), so an item fits a dword, both children come with one ds_read2_b32,
// and "a.f >= b.f" is (a | 0x3FF) >= b: the index bits of b never exceed the ten ones or-ed into a.
__device__ __forceinline__ bool p_ge(uint32_t x, uint32_t y) { return (x | 0x3FFu) >= y; }

template <typename P>  // (uint32_t* or its LDS-qualified form)
__device__ __forceinline__ void p_sift_down_range(P d, uint32_t pos, uint32_t end) {
    const uint32_t elem = d[pos];
    uint32_t hole = pos, child = 2 * hole + 1;
    while (child + 1 < end) {  // two children
        const uint32_t c0 = d[child], c1 = d[child + 1];
        const bool right = p_ge(c0, c1);  // d[child] <= d[child + 1]
        const uint32_t c = right ? c1 : c0;
        if (p_ge(c, elem)) {  // elem >= d[child]
            d[hole] = elem;
            return;
        }
        d[hole] = c;
        hole = right ? child + 1 : child;
        child = 2 * hole + 1;
    }
    if (child + 1 == end) {
        const uint32_t c = d[child];
        if (!p_ge(c, elem)) {  // elem < d[child]
            d[hole] = c;
            hole = child;
        }
    }
    d[hole] = elem;
}

__device__ __forceinline__ uint32_t wave_incl_max_u32(uint32_t v) {
    uint32_t x = v;
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false));  // row_shr:1
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false));  // row_bcast:15
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false));  // row_bcast:31
    return x;
}

// Whole wavefront: the depths of the leaves (:247-259) by pointer jumping instead of one pass on one lane.  Every
// internal node starts at distance 1 from its parent (the root at 0 from itself); a round replaces (distance to an
// ancestor, that ancestor) by (distance to the ancestor's ancestor, ...) until every node's ancestor is the root: as
// many rounds as the depth has bits.  The heap's array is free by now and holds the two 16-bit arrays.
// ni = internal nodes (>= 1), CAP = the scratch's capacity.
template <int CAP>
__device__ __noinline__ void g_huff_depths_wave(const GHuffView v, const uint32_t ni, const int lane) {
    using lds_u16 = __attribute__((address_space(3))) uint16_t;
    using lds_u8 = __attribute__((address_space(3))) uint8_t;
    const uint32_t N = v.n, root = ni - 1;
    lds_u16* const anc = (lds_u16*)reinterpret_cast<uint16_t*>(v.heap);
    lds_u16* const dist = anc + CAP;
    const lds_u16* const in_left = (lds_u16*)v.in_left;
    const lds_u16* const in_right = (lds_u16*)v.in_right;
    lds_u8* const lengths = (lds_u8*)v.lengths;
    for (uint32_t k = (uint32_t)lane; k < ni; k += kWave) {
        anc[k] = (uint16_t)root;
        dist[k] = k == root ? 0 : 1;
    }
    wave_sync();
    for (uint32_t k = (uint32_t)lane; k < ni; k += kWave) {  // every internal node but the root is the child of one node
        const uint32_t l = in_left[k], r = in_right[k];
        if (l >= N) anc[l - N] = (uint16_t)k;
        if (r >= N) anc[r - N] = (uint16_t)k;
    }
    wave_sync();
    for (;;) {
        bool moved = false;
        for (uint32_t k = (uint32_t)lane; k < ni; k += kWave) {
            const uint32_t a = anc[k];
            const uint32_t da = dist[a], aa = anc[a];  // (read by every lane before any lane writes)
            wave_sync();
            if (a != root) {
                dist[k] = (uint16_t)(dist[k] + da);
                anc[k] = (uint16_t)aa;
                moved = true;
            }
            wave_sync();
        }
        if (!__any(moved)) break;
    }
    uint32_t max_length = 0;
    for (uint32_t k = (uint32_t)lane; k < ni; k += kWave) {
        const uint32_t dch = (uint32_t)dist[k] + 1;
        const uint32_t l = in_left[k], r = in_right[k];
        if (l < N) {
            lengths[l] = (uint8_t)dch;
            max_length = max(max_length, dch);
        }
        if (r < N) {
            lengths[r] = (uint8_t)dch;
            max_length = max(max_length, dch);
        }
    }
    max_length = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_max_u32(max_length), kWave - 1);
    if (lane == 0) v.hdr[2] = max_length;
    wave_sync();
}

// ---- The merges on the whole wavefront: a sift is a path, and a path is found without touching the items. ----
// sift_down_to_bottom follows the smaller child at every level, and sift_down_range follows the same path until the
// moving item fits; which child is the smaller one is a property of the heap, not of the moving item.  So: every lane
// compares the two children of "its" nodes (positions lane, 64 + lane, 128 + lane) and three ballots hold the
// decision of every node; the path from the root is then scalar bit arithmetic (position p -> 2p + bit(p), positions
// 1-based so that the ancestors of the bottom position P are P >> 1, P >> 2, ...); lane k fetches the item at the
// path's level k, one ballot finds where the moving item stops (pop: the sift_up of the last item from the bottom;
// PeekMut::drop: the first level whose item is not smaller), and the lanes above that level each store their
// neighbour's item one level up.  Two LDS round trips and ~60 instructions per sift instead of ~28 instructions and a
// round trip per LEVEL of it on one lane.  Items behind the heap's end are kept at 0xFFFFFFFF (a frequency no packed
// item reaches), so a node with one child goes left and the path ends where the heap ends.  The comparisons are the
// ones of p_sift_down_range / p_heap_pop, in an order that cannot be observed: same heap after every step.
// LEVELS = 8: the literal/length tree (at most 288 items, 576 dwords of array); 5: the small trees (at most 32 items,
// 64 dwords).  v.hdr[0] = items, in index order as g_huff_prepare<true> left them.  The loop is written for its
// instruction count: no branch in a sift, the path is computed to full depth and cut where it leaves the heap.
template <int LEVELS>
__device__ __noinline__ void g_huff_merges_wave(const GHuffView v, const int lane) {
    constexpr uint32_t kCap = LEVELS == 8 ? 576 : 64;
    const uint32_t N = v.n;
    // (the view's pointers are generic: behind a call the compiler no longer sees that they are LDS addresses)
    using lds_u32 = __attribute__((address_space(3))) uint32_t;
    using lds_u16 = __attribute__((address_space(3))) uint16_t;
    lds_u32* const d = (lds_u32*)reinterpret_cast<uint32_t*>(v.heap);
    lds_u16* const in_left = (lds_u16*)v.in_left;
    lds_u16* const in_right = (lds_u16*)v.in_right;
    uint32_t hl = uni(v.hdr[0]);
    if (hl <= 1) {  // :206-213 nothing or a single symbol (length 1)
        if (hl == 1 && lane == 0) {
            v.lengths[d[0] & 0x3FFu] = 1;
            v.hdr[2] = 1;
        }
        return;
    }
    for (uint32_t i = hl + (uint32_t)lane; i < kCap; i += kWave) d[i] = 0xFFFFFFFFu;
    wave_sync();
    // BinaryHeap::from(vec): sift_down_range(k, len) for k = len / 2 - 1 .. 0.  The nodes of one level have disjoint
    // subtrees, so a level's sifts run side by side, one lane each; levels bottom-up as in the sequential order.
    for (int lev = 31 - __clz((int)(hl / 2)); lev >= 0; lev--) {
        const uint32_t first = 1u << lev, last_node = min(2 * first - 1, hl / 2);
        for (uint32_t node = first + (uint32_t)lane; node <= last_node; node += kWave) p_sift_down_range(d, node - 1, hl);
        wave_sync();
    }
    const lds_u32* const d1 = d - 1;  // position p (1-based) is d1[p]
    // the children of "this lane's" nodes: positions 2q, 2q + 1 for q = lane, 64 + lane, 128 + lane
    const uint32_t q0 = LEVELS == 8 ? (lane ? (uint32_t)lane : 1u) : (lane ? (uint32_t)lane & 31u : 1u);
    const lds_u32* const c0 = d1 + 2 * q0;
    const lds_u32* const c1 = d1 + 2 * (64 + (uint32_t)lane);
    const lds_u32* const c2 = d1 + 2 * (128 + (uint32_t)lane);
    const uint32_t lv = (uint32_t)(LEVELS - lane) & 31u;  // lane k looks at level k of a path
    const uint64_t lanes_on_path = ((uint64_t)2 << LEVELS) - 1;
    // The path from the root along the smaller children, as its position at level LEVELS (it leaves the heap on the
    // way when the heap is shallower); `pk` = this lane's position on it, `vmask` = the lanes whose level exists.
    auto path = [&](uint32_t end, uint32_t& pk, uint64_t& vmask) __attribute__((always_inline)) {
        const uint64_t m0 = __ballot(p_ge(c0[0], c0[1]));
        uint32_t p = 1;
        // p = 2p + bit p of the mask (s_bitcmp1_b64 looks at the low six bits of p; the carry doubles and adds)
#define FDH_PATH_STEP(m) asm("s_bitcmp1_b64 %1, %0\n\ts_addc_u32 %0, %0, %0" : "+s"(p) : "s"(m) : "scc")
#pragma unroll
        for (int l = 0; l < (LEVELS < 6 ? LEVELS : 6); l++) FDH_PATH_STEP(m0);
        if (LEVELS == 8) {
            const uint64_t m1 = __ballot(p_ge(c1[0], c1[1]));
            const uint64_t m2 = __ballot(p_ge(c2[0], c2[1]));
            FDH_PATH_STEP(m1);
            FDH_PATH_STEP(m2);
        }
#undef FDH_PATH_STEP
        const uint32_t mine = p >> lv;
        vmask = __ballot(mine <= end) & lanes_on_path;
        pk = mine <= end ? mine : 1u;
    };
    uint32_t ni = 0;
    while (hl > 1) {  // :236-244
        // ---- pop: the last item goes to the root, sift_down_to_bottom(0), then sift_up ----
        const uint32_t end = hl - 1;
        const uint32_t last_v = d[end];
        if (lane == 0) d[end] = 0xFFFFFFFFu;
        uint32_t pk;
        uint64_t vmask;
        path(end, pk, vmask);
        uint32_t x = d1[pk];
        const uint32_t last = uni(last_v);
        // elem <= d[parent] at level k - 1, whose item came up from level k: the sift_up stops there
        const uint64_t stay = __ballot(p_ge(last, x)) & vmask & ~(uint64_t)1;
        const uint32_t j = stay ? 63 - (uint32_t)__clzll((long long)stay) : 0u;
        uint32_t xn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xF, 0xF, false);  // row_shl:1: lane k + 1's item
        if ((uint32_t)lane <= j) d[pk - 1] = (uint32_t)lane == j ? last : xn;
        const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)x, 0);
        const uint32_t b = j == 0 ? last : (uint32_t)__builtin_amdgcn_readlane((int)x, 1);
        // ---- the root takes the merged node; PeekMut::drop sifts it down ----
        if (lane == 0) {
            in_left[ni] = (uint16_t)(a & 0x3FFu);
            in_right[ni] = (uint16_t)(b & 0x3FFu);
        }
        ni++;
        const uint32_t elem = (((a >> 10) + (b >> 10)) << 10) | (ni + N - 1);
        path(end, pk, vmask);
        x = d1[pk];
        const uint64_t stop = __ballot(p_ge(x, elem)) & vmask & ~(uint64_t)1;  // elem >= d[child]
        const uint32_t sl = stop ? (uint32_t)__ffsll((long long)stop) - 1 : (uint32_t)__popcll(vmask);  // elem goes to level sl - 1
        xn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xF, 0xF, false);
        if ((uint32_t)lane < sl) d[pk - 1] = (uint32_t)lane == sl - 1 ? elem : xn;
        hl = end;
    }
    wave_sync();
}

// Whole wavefront: length limiting (:262-305) when the tree came out deeper than `limit`.
// The reference sorts the symbols by frequency and hands the adjusted counts out from the longest
// length down; here every lane finds the rank of its symbols among the used ones directly (by
// frequency, ties in index order = the order of a stable sort; sort_unstable is an insertion sort
// up to 20 elements and implementation-defined in its tie order beyond that, see the oracle).
__device__ void g_huff_limit(const GHuffView& v, uint32_t* start /* 16 */, int lane) {
    const uint32_t N = v.n, limit = v.limit;
    if (lane < 16) v.counts[lane] = 0;
    wave_sync();
    for (uint32_t i = lane; i < N; i += kWave) atomicAdd(&v.counts[min((uint32_t)v.lengths[i], limit)], 1u);
    wave_sync();
    if (lane == 0) {
        uint32_t total = 0;
        for (uint32_t i = 1; i <= limit; i++) total += v.counts[i] << (limit - i);
        while (total > (1u << limit)) {
            uint32_t i = limit - 1;
            while (v.counts[i] == 0) i--;
            v.counts[i]--;
            v.counts[limit]--;
            v.counts[i + 1] += 2;
            total--;
        }
        uint32_t at = 0;  // the first counts[limit] used symbols (rarest first) get `limit`, and so on down
        for (uint32_t len = limit; len >= 1; len--) {
            start[len] = at;
            at += v.counts[len];
        }
        v.hdr[2] = limit;
    }
    wave_sync();
    for (uint32_t base = 0; base < N; base += kWave) {
        const uint32_t i = base + lane;
        const uint32_t fi = i < N ? v.freq[i] : 0;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < N; j++) {
            const uint32_t fj = v.freq[j];  // same address in every lane: one broadcast read
            rank += (fj > 0 && (fj < fi || (fj == fi && j < i))) ? 1u : 0u;
        }
        if (fi > 0) {
            uint32_t len = limit;
            while (len > 1 && !(rank >= start[len] && rank < start[len] + v.counts[len])) len--;
            v.lengths[i] = (uint8_t)len;
        }
    }
    wave_sync();
}

// Whole wavefront: canonical codes, bit-reversed (:308-320), as code | length << 16.
__device__ void g_huff_codes(const GHuffView& v, uint32_t* cl, uint32_t* first /* 16 */, int lane) {
    if (lane < 16) first[lane] = 0;
    wave_sync();
    for (uint32_t i = lane; i < v.n; i += kWave) {
        const uint32_t l = v.lengths[i];
        if (l) atomicAdd(&first[l], 1u);  // counts per length, turned into first codes below
    }
    wave_sync();
    if (lane == 0) {
        uint32_t code = 0;
        for (uint32_t len = 1; len <= 15; len++) {
            const uint32_t c = first[len];
            first[len] = code;
            code = (code + c) << 1;
        }
    }
    wave_sync();
    for (uint32_t base = 0; base < v.n; base += kWave) {
        const uint32_t i = base + lane;
        const uint32_t l = i < v.n ? v.lengths[i] : 0;
        // the lanes with this lane's length: one ballot per bit of the length instead of one per length
        uint64_t m = ~(uint64_t)0;
#pragma unroll
        for (int bit = 0; bit < 4; bit++) {
            const uint64_t bm = __ballot((l >> bit) & 1);
            m &= ((l >> bit) & 1) ? bm : ~bm;
        }
        const uint32_t rank = (uint32_t)__popcll(m & lanemask_lt(lane)), same = (uint32_t)__popcll(m);
        uint32_t code = 0;
        if (l) code = first[l] + rank;
        wave_sync();
        if (l && rank == 0) first[l] += same;  // one lane per length moves its counter on
        wave_sync();
        if (i < v.n) cl[i] = l ? ((__brev(code) >> (32 - l)) | (l << 16)) : 0u;
    }
}

struct GWriteArgs {
    const uint8_t* in;
    const uint64_t* in_off;
    uint8_t* out;
    const uint64_t* out_off;
    uint32_t* out_len;
    uint64_t n;
    const GMatchRec* matches;
    const GBlockRec* blocks;
    const uint32_t* nblocks;
};


// table[idx] += 1 in the lanes that are `on` (some lane is): the lanes that agree with the first of them in one add.
__device__ __forceinline__ void g_count(uint32_t* table, uint32_t idx, bool on, int lane) {
    const uint64_t act = __ballot(on);
    const int first = __ffsll((unsigned long long)act) - 1;
    const uint32_t v0 = (uint32_t)__builtin_amdgcn_readlane((int)idx, first);
    const uint64_t same = __ballot(on && idx == v0);
    if (lane == first) atomicAdd(&table[v0], (uint32_t)__popcll(same));
    if (on && idx != v0) atomicAdd(&table[idx], 1u);
}

// One walk over the positions [b0, b1) of a block whose back-references are recs[m0, m1).
// EMIT = false: symbol frequencies (bitstream.rs:42-66); EMIT = true: the symbols (:121-186).
// Every lane takes four consecutive positions per step (256 per wavefront): back-references are at
// least three bytes long, so a lane sees at most two of them and its codes stay below the 118 bits
// one ring update takes.
// The loads of the walk (the next step's four bytes, the next chunk's records) are UNCONDITIONAL,
// from clamped addresses: behind a load in a branch the compiler has to wait for every load in
// flight at the join, the prefetch it has just issued included.  (`in` has at least four readable
// bytes: the kernel stands a padded copy in for a shorter stream.)
template <bool EMIT>
__device__ void g_walk(GWriteLds& lds, BitRing& br, const uint8_t* in, const GMatchRec* recs, uint32_t b0, uint32_t b1,
                       uint32_t m0, uint32_t m1, uint64_t len, uint64_t& acc_a, uint64_t& acc_b) {
    const uint32_t lane = (uint32_t)br.lane;
    constexpr uint32_t kStep = 4 * kWave;
    for (uint32_t i = lane; i < kGChunk; i += kWave) lds.marks[i] = 0;  // (shared with the Huffman scratch)
    wave_sync();
    // the four bytes at p (zero behind the block / the stream), never reading past the stream's end
    const uint32_t last4 = len >= 4 ? (uint32_t)len - 4 : 0u;
    auto load4 = [&](uint32_t p) -> uint32_t {
        const uint32_t q = min(p, last4);
        uint32_t w;
        __builtin_memcpy(&w, in + q, 4);
        const uint32_t sh = 8 * (p - q);  // the dword that ends at the stream's end, shifted down to p
        w = sh < 32 ? w >> sh : 0u;
        return p < b1 ? w : 0u;
    };
    const uint32_t mlast = m1 ? m1 - 1 : 0;  // (the slice has at least two records, used or not)
    uint32_t mc = m0, covered = b0;  // next record to mark; end of the last back-reference so far
    uint32_t zeros = 0;              // literal zeros this lane met on the frequency walk
    uint32_t word_next = load4(b0 + 4 * lane);
    GMatchRec rnext = recs[min(mc + lane, mlast)];
    for (uint32_t c0 = b0; c0 < b1; c0 += kGChunk) {
        const uint32_t cend = min(c0 + kGChunk, b1);
        for (;;) {
            const uint32_t idx = mc + lane;
            GMatchRec r = rnext;
            if (idx >= m1) r.start = 0xFFFFFFFFu;
            const bool here = r.start < cend;
            if (here) lds.marks[r.start - c0] = r.info;
            const uint32_t cnt = (uint32_t)__popcll(__ballot(here));
            mc += cnt;
            rnext = recs[min(mc + lane, mlast)];  // for the next round, or for the next chunk (four steps away)
            if (cnt < (uint32_t)kWave) break;
        }
        wave_sync();
        for (uint32_t q = c0; q < cend; q += kStep) {
            const uint32_t p0 = q + 4 * lane;
            const uint32_t nact = p0 < cend ? min(cend - p0, 4u) : 0u;
            uint32_t word = word_next;
            word_next = load4(p0 + kStep);
            if (nact < 4) word &= nact ? (1u << (8 * nact)) - 1 : 0u;  // (the bytes behind belong to the next block)
            uint4 mk4 = *reinterpret_cast<const uint4*>(&lds.marks[p0 - c0]);
            if (mk4.x | mk4.y | mk4.z | mk4.w) *reinterpret_cast<uint4*>(&lds.marks[p0 - c0]) = make_uint4(0, 0, 0, 0);
            // At most two back-references start in a lane's four positions (they are at least three bytes long): the
            // first of them, A, and B at the fourth position behind a three-byte A at the first.  They are looked at
            // once each instead of once per position, and the positions between are told apart without a branch.
            const uint32_t mk0 = mk4.x, mk1 = mk4.y, mk2 = mk4.z, mk3 = mk4.w;
            const uint32_t mA = mk0 ? mk0 : (mk1 ? mk1 : (mk2 ? mk2 : mk3));
            const uint32_t jA = mk0 ? 0u : (mk1 ? 1u : (mk2 ? 2u : 3u));
            const bool hasA = mA != 0, hasB = mk0 != 0 && mk3 != 0;
            const uint32_t lenA = mA & 0x1FF, endA = p0 + jA + lenA;
            const uint32_t lenB = mk3 & 0x1FF, endB = p0 + 3 + lenB;
            const uint32_t e = hasB ? endB : (hasA ? endA : 0u);
            const uint32_t incl = wave_incl_max_u32(e);
            const uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x138, 0xF, 0xF, false);  // wave_shr:1
            const uint32_t cov = max(covered, before);  // end of the last back-reference in front of p0
            covered = max(covered, (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1));
            if (EMIT) {  // Adler-32 partial sums: A = 1 + sum d_i, B = len + sum (len - i) d_i
                const uint32_t sum = bytesum4(word);
                acc_a += sum;
                acc_b += (len - p0) * sum - bytedot4(word, 0x03020100u, 0);
            }
            // A literal: no back-reference starts there, and it lies behind the last one's end (A's, behind A).  As a
            // mask of the four positions: those from `cov` on and in front of A, those from A's end on, without B's.
            uint32_t litmask = 0xFu << min(max(cov, p0) - p0, 4u);
            {
                const uint32_t front = ~(0xFu << jA), back = 0xFu << min(endA - p0, 4u);
                litmask = hasA ? ((litmask & front) | back) : litmask;
                litmask &= hasB ? 0x7u : 0xFu;
                litmask &= (1u << nact) - 1;
            }
            bool lit[4];
#pragma unroll
            for (int j = 0; j < 4; j++) lit[j] = (litmask >> j) & 1;
            uint64_t v0 = 0, v1 = 0;
            uint32_t nb = 0;
            if (!EMIT) {
#pragma unroll
                // An LDS atomic takes one pass per lane that shares its address, and filtered image bytes share a lot:
                // a quarter of the literals are zeros (the zeros too isolated for a run), every run's distance code is
                // the same.  The zeros are counted per lane and added once per walk; of the back-references the lanes
                // that agree with the first one are counted once.
                for (int j = 0; j < 4; j++) {
                    const uint32_t byte = (word >> (8 * j)) & 0xFF;
                    zeros += (lit[j] && byte == 0) ? 1u : 0u;
                    if (lit[j] && byte != 0) atomicAdd(&lds.freq[byte], 1u);
                }
                if (__any(hasA)) {
                    uint32_t sym = 0, extra;
                    if (hasA) g_length_symbol(lenA, sym, extra);
                    g_count(lds.freq, sym, hasA, (int)lane);
                    g_count(lds.dfreq, (mA >> 9) & 31, hasA, (int)lane);
                    if (__any(hasB)) {
                        if (hasB) {
                            g_length_symbol(lenB, sym, extra);
                            atomicAdd(&lds.freq[sym], 1u);
                            atomicAdd(&lds.dfreq[(mk3 >> 9) & 31], 1u);
                        }
                    }
                }
            } else {
                // :163-185 length code, length extra bits, distance code, distance extra bits: at most 48 bits
                auto piece = [&](uint32_t m, uint32_t length, uint64_t& v, uint32_t& n) __attribute__((always_inline)) {
                    uint32_t sym, extra;
                    g_length_symbol(length, sym, extra);
                    const uint32_t ds = (m >> 9) & 31;
                    const uint32_t el = lds.cl[sym], d = lds.dcl[ds], dm = lds.dmeta[ds];
                    const uint32_t distance = ((m >> 14) & 0x7FFF) + 1;
                    uint32_t lo = el & 0xFFFF;
                    n = el >> 16;
                    lo |= ((length - 3) & ((1u << extra) - 1)) << n;  // (15 + 5 bits)
                    n += extra;
                    v = lo | ((uint64_t)(d & 0xFFFF) << n);
                    n += d >> 16;
                    v |= (uint64_t)(distance - (dm & 0xFFFF)) << n;
                    n += dm >> 16;
                };
                uint32_t el[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    el[j] = lds.cl[(word >> (8 * j)) & 0xFF];
                    el[j] = lit[j] ? el[j] : 0u;
                }
                uint64_t vA = 0, vB = 0;
                uint32_t nA = 0, nB = 0;
                if (__any(hasA)) {
                    if (hasA) piece(mA, lenA, vA, nA);
                    if (__any(hasB))
                        if (hasB) piece(mk3, lenB, vB, nB);
                }
                // The literals in front of A (behind it the lane's positions are covered, but for the fourth behind a
                // three-byte A at the first), A, then the fourth position's literal or B.  Every piece lands below bit
                // 64 (at most 45 bits of literals in front of A; A alone, at most 48, in front of the fourth), so only A
                // and the fourth can spill into the high half.
                const uint32_t n0 = el[0] >> 16, n1 = el[1] >> 16, n2 = el[2] >> 16;
                const uint32_t nfront = n0 + n1 + n2;
                v0 = ((el[0] & 0xFFFF) | ((el[1] & 0xFFFF) << n0)) | ((uint64_t)(el[2] & 0xFFFF) << (n0 + n1));
                v0 |= vA << nfront;
                v1 = (vA >> 1) >> (63 - nfront);
                const uint64_t vT = hasB ? vB : (uint64_t)(el[3] & 0xFFFF);
                const uint32_t nT = hasB ? nB : el[3] >> 16;
                const uint32_t at = nfront + nA;  // (64 or more only where the fourth position is covered: vT = 0)
                v0 |= vT << (at & 63);
                v1 |= (vT >> 1) >> ((63 - at) & 63);
                nb = at + nT;
            }
            if (EMIT) {
                uint32_t total;
                const uint32_t off = wave_excl_scan_u32(nb, (int)lane, total);
                if ((uint64_t)total + (br.qbits - br.qflushed) > kEncTileBudget) br.flush(false);
                br.or_bits128(br.qbits + off, v0, v1);
                br.qbits += total;
                if (br.qbits - br.qflushed > kEncRingBits / 2) br.flush(false);
            }
        }
    }
    if (!EMIT) {
        zeros = wave_sum_u32(zeros);
        if (lane == 0 && zeros) atomicAdd(&lds.freq[0], zeros);
    }
    wave_sync();
}

__global__ __launch_bounds__(kWave, 4) void deflate_write_kernel(GWriteArgs a) {
    __shared__ GWriteLds lds;
    const int lane = threadIdx.x;
    const uint64_t sid = blockIdx.x;
    const uint64_t in0 = a.in_off[0], off = a.in_off[sid];
    const uint8_t* in = a.in + off;
    const uint64_t len = a.in_off[sid + 1] - off;
    const uint32_t nblocks = a.nblocks[sid];
    if (nblocks == 0xFFFFFFFFu) {
        if (lane == 0) a.out_len[sid] = 0xFFFFFFFFu;
        return;
    }
    const GMatchRec* recs = a.matches + g_match_slice(off - in0, sid);
    const GBlockRec* blks = a.blocks + g_block_slice(off - in0, sid);
    if (len < 4) {
        // The walks read dwords.  A stream this short has no back-references, so its slice of the
        // record array (two records at least) is free: a zero-padded copy of the bytes goes there.
        uint8_t* pad = reinterpret_cast<uint8_t*>(const_cast<GMatchRec*>(recs));
        if (lane < 8) pad[lane] = (uint64_t)lane < len ? in[lane] : (uint8_t)0;
        __threadfence();
        __builtin_amdgcn_wave_barrier();
        in = pad;
    }
    for (int i = lane; i < kGRingDw; i += kWave) lds.ring[i] = 0;
    if (lane < 30) lds.dmeta[lane] = (uint32_t)kDistBase[lane] | ((uint32_t)kDistExtra[lane] << 16);
    wave_sync();

    uint8_t* out = a.out + a.out_off[sid];
    const uint64_t cap = a.out_off[sid + 1] - a.out_off[sid];
    BitRing br;
    br.ring = lds.ring;
    br.lane = lane;
    br.gmis = (uint32_t)(reinterpret_cast<uintptr_t>(out) & 15);
    br.out_al = out - br.gmis;
    br.cap_bits = ((uint64_t)br.gmis + cap) * 8;
    br.qbits = (uint64_t)br.gmis * 8;
    br.qflushed = 0;
    br.overflow = false;
    br.emit_uniform(0x0178, 16);  // zlib header 78 01 (compress/mod.rs:61-63)

    uint64_t acc_a = 0, acc_b = 0;
    uint32_t b0 = 0, m0 = 0;
#ifdef FDH_DEBUG_GEN
    uint32_t wt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t wt_all = __builtin_readcyclecounter();
    uint64_t wt_last = wt_all;
#define GW_T(k)                                              \
    {                                                        \
        const uint64_t now_ = __builtin_readcyclecounter();  \
        wt[k] += (uint32_t)(now_ - wt_last);                 \
        wt_last = now_;                                      \
    }
#else
#define GW_T(k)
#endif
    for (uint32_t blk = 0; blk < nblocks; blk++) {
        const uint32_t b1 = uni(blks[blk].end_pos), m1 = uni(blks[blk].match_end), flags = uni(blks[blk].flags);
        if (flags & kGBlockEmptyFixed) {  // compress/mod.rs:234-238: BFINAL, fixed codes, end of block; flush
            br.emit_uniform(3, 10);
            continue;
        }
        // ---- frequencies (bitstream.rs:42-66) ----
        for (int i = lane; i < 288; i += kWave) lds.freq[i] = i == 256 ? 1u : 0u;
        if (lane < 32) lds.dfreq[lane] = 0;
        if (lane < 20) lds.clfreq[lane] = 0;
        wave_sync();
        GW_T(6)
        g_walk<false>(lds, br, in, recs, b0, b1, m0, m1, len, acc_a, acc_b);
        GW_T(0)
        // ---- the two code tables (:68-73) ----
        const GHuffView vl = g_view(lds.h.big, lds.freq, 286, 15), vd = g_view(lds.h.small, lds.dfreq, 30, 15);
        // (packed heap items hold 22 bits of frequency: a block's symbols number at most its positions + 1)
        const bool packed = b1 - b0 < (1u << 22) - 2;
        if (packed) {
            g_huff_prepare<true>(vl, lane);
            g_huff_prepare<true>(vd, lane);
            wave_sync();
            GW_T(1)
            g_huff_merges_wave<8>(vl, lane);
            GW_T(8)
            g_huff_merges_wave<5>(vd, lane);
            GW_T(9)
            {
                const uint32_t hl_l = uni(lds.h.big.heap_len), hl_d = uni(lds.h.small.heap_len);
                if (hl_l > 1) g_huff_depths_wave<288>(vl, hl_l - 1, lane);
                if (hl_d > 1) g_huff_depths_wave<32>(vd, hl_d - 1, lane);
            }
        } else {
            g_huff_prepare<false>(vl, lane);
            g_huff_prepare<false>(vd, lane);
            wave_sync();
            if (lane < 2) g_huff_serial(lane == 0 ? vl : vd);
        }
        wave_sync();
        GW_T(2)
        if (uni(lds.h.big.max_length) > 15) g_huff_limit(vl, lds.h.big.first, lane);
        if (uni(lds.h.small.max_length) > 15) g_huff_limit(vd, lds.h.small.first, lane);
        g_huff_codes(vl, lds.cl, lds.h.big.first, lane);
        g_huff_codes(vd, lds.dcl, lds.h.small.first, lane);
        wave_sync();
        GW_T(3)
        // ---- header (:75-119): counts trimmed of trailing zero lengths, the code-length code ----
        uint32_t num_litlen = 286, num_dist = 30;
        if (lane == 0) {
            while (num_litlen > 257 && lds.h.big.lengths[num_litlen - 1] == 0) num_litlen--;
            while (num_dist > 1 && lds.h.small.lengths[num_dist - 1] == 0) num_dist--;
        }
        num_litlen = uni(num_litlen);
        num_dist = uni(num_dist);
        for (uint32_t i = lane; i < num_litlen + num_dist; i += kWave) {
            const uint32_t l = i < num_litlen ? lds.h.big.lengths[i] : lds.h.small.lengths[i - num_litlen];
            atomicAdd(&lds.clfreq[l], 1u);
        }
        wave_sync();
        // the lengths of both tables are needed after the scratch is reused: keep them in cl/dcl (>> 16)
        {
            // the code-length tree reuses the small scratch: move the distance lengths out first
            // (they live on in dcl[] >> 16)
            const GHuffView vc = g_view(lds.h.small, lds.clfreq, 19, 7);
            g_huff_prepare<true>(vc, lane);  // (at most 316 code lengths: always packed)
            wave_sync();
            GW_T(4)
            g_huff_merges_wave<5>(vc, lane);
            GW_T(10)
            {
                const uint32_t hl_c = uni(lds.h.small.heap_len);
                if (hl_c > 1) g_huff_depths_wave<32>(vc, hl_c - 1, lane);
            }
            wave_sync();
            if (uni(lds.h.small.max_length) > 7) g_huff_limit(vc, lds.h.small.first, lane);
            g_huff_codes(vc, lds.clcl, lds.h.small.first, lane);
            wave_sync();
        }
        br.emit_uniform(((flags & kGBlockEof) ? 5u : 4u) | ((num_litlen - 257) << 3) | ((num_dist - 1) << 8) | (15u << 13), 17);
        if (br.qbits + 57 - br.qflushed > kEncTileBudget) br.flush(false);
        if (lane < 19) br.or_bits(br.qbits + 3 * (uint32_t)lane, lds.clcl[kClclOrder[lane]] >> 16);
        br.qbits += 57;
        for (uint32_t base = 0; base < num_litlen + num_dist; base += kWave) {
            const uint32_t i = base + lane;
            uint32_t bits = 0, nb = 0;
            if (i < num_litlen + num_dist) {
                const uint32_t l = (i < num_litlen ? lds.cl[i] : lds.dcl[i - num_litlen]) >> 16;
                const uint32_t e = lds.clcl[l];
                bits = e & 0xFFFF;
                nb = e >> 16;
            }
            uint32_t total;
            const uint32_t o = wave_excl_scan_u32(nb, lane, total);
            if ((uint64_t)total + (br.qbits - br.qflushed) > kEncTileBudget) br.flush(false);
            br.or_bits(br.qbits + o, bits);
            br.qbits += total;
        }
        wave_sync();
        GW_T(4)
        // ---- the symbols, end of block (:121-194) ----
        g_walk<true>(lds, br, in, recs, b0, b1, m0, m1, len, acc_a, acc_b);
        br.emit_uniform(lds.cl[256] & 0xFFFF, lds.cl[256] >> 16);
        GW_T(5)
        b0 = b1;
        m0 = m1;
    }
    // ---- Compressor::finish (compress/mod.rs:194-214): pad to a byte, Adler-32 big-endian ----
    br.emit_uniform(0, (uint32_t)(8 - (br.qbits & 7)) & 7);
    const uint32_t pa = (uint32_t)(acc_a % kAdlerMod), pb = (uint32_t)(acc_b % kAdlerMod);
    const uint32_t A = (1u + wave_sum_u32(pa)) % kAdlerMod;
    const uint32_t B = (uint32_t)(((len % kAdlerMod) + wave_sum_u32(pb)) % kAdlerMod);
    br.emit_uniform(__builtin_bswap32((B << 16) | A), 32);
    br.flush(true);
    if (lane == 0) a.out_len[sid] = br.overflow ? 0xFFFFFFFFu : (uint32_t)((br.qbits >> 3) - br.gmis);
#ifdef FDH_DEBUG_GEN
    wt[7] = (uint32_t)(__builtin_readcyclecounter() - wt_all);
    if (lane == 0 && sid < 32768)
        for (int k = 0; k < 12; k++) g_gen_w[k][sid] = wt[k];
#endif
}

}  // namespace fdh

#ifdef FDH_DEBUG_GEN
extern "C" int fdh_debug_gen_write_timers(uint32_t* host /* 8 x 32768 */) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fdh::g_gen_w), sizeof(uint32_t) * 12 * 32768);
}
extern "C" int fdh_debug_gen_timers(uint64_t* host /* 8 x 32768 */) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fdh::g_gen_t), sizeof(uint64_t) * 8 * 32768);
}
#endif

extern "C" size_t fdh_deflate_general_hash_bytes(void) { return (size_t)fdh::kGHashSize * 4; }
// record slices for a batch whose inputs span `total_in` bytes: element counts
extern "C" size_t fdh_deflate_general_match_records(uint64_t total_in, uint64_t n) { return (size_t)fdh::g_match_slice(total_in, n) + 2; }
extern "C" size_t fdh_deflate_general_block_records(uint64_t total_in, uint64_t n) { return (size_t)fdh::g_block_slice(total_in, n) + 4; }
extern "C" size_t fdh_deflate_general_match_record_bytes(void) { return sizeof(fdh::GMatchRec); }
extern "C" size_t fdh_deflate_general_block_record_bytes(void) { return sizeof(fdh::GBlockRec); }

extern "C" int fdh_launch_deflate_general(const uint8_t* in, const uint64_t* in_off, uint8_t* out, const uint64_t* out_off,
                                          uint32_t* out_len, uint64_t n, int rle, void* hash, void* matches, void* blocks,
                                          uint32_t* nblocks, unsigned waves, unsigned lanes, hipStream_t stream) {
    if (n == 0) return 0;
    fdh::GParseArgs p{in, in_off, n, static_cast<uint32_t*>(hash), static_cast<fdh::GMatchRec*>(matches),
                      static_cast<fdh::GBlockRec*>(blocks), nblocks, lanes};
    if (rle)
        hipLaunchKernelGGL(fdh::deflate_parse_kernel<true>, dim3(waves), dim3(fdh::kWave), 0, stream, p);
    else
        hipLaunchKernelGGL(fdh::deflate_parse_kernel<false>, dim3(waves), dim3(fdh::kWave), 0, stream, p);
    fdh::GWriteArgs w{in, in_off, out, out_off, out_len, n, static_cast<const fdh::GMatchRec*>(matches),
                      static_cast<const fdh::GBlockRec*>(blocks), nblocks};
    hipLaunchKernelGGL(fdh::deflate_write_kernel, dim3((unsigned)n), dim3(fdh::kWave), 0, stream, w);
    return (int)hipGetLastError();
}
