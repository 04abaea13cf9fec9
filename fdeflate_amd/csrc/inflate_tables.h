// inflate_tables.h -- wave-parallel Huffman decode-table construction.
//
// Restates huffman::build_table (reference src/huffman.rs:18-184) and
// CompressedBlock::build_tables (reference src/decompress.rs:561-606) for one wavefront.
// The *decoding function* is identical to the reference's; the entry layout is the device's
// own (DESIGN.md "decode tables"):
//
//   litlen entry (u32), primary index = low kLitBits stream bits
//     [3:0]   nbits   total bits consumed by the entry (1..15)
//     [7:4]   kind    K_LIT1 | K_LIT2 | K_LEN | K_EOB | K_LONG
//     K_LIT1  [15:8] symbol                      [27:24] = nbits
//     K_LIT2  [15:8] first, [23:16] second,      [27:24] = bits of the first symbol
//     K_LEN   [12:8] extra-bit count, [31:16] length base
//     K_LONG  code longer than kLitBits: resolved by a canonical walk (long_decode)
//   dist entry (u32), primary index = low kDistBits bits
//     [3:0] nbits, [7:4] kind D_INVALID | D_DIST | D_LONG, [11:8] extra bits, [31:16] base
//   code-length-code entry (u32): [3:0] nbits, [15:8] symbol
//
// Double-literal entries are synthesised exactly where the reference does
// (src/huffman.rs:110-130): both symbols < 256 and len1 + len2 <= kLitBits.
#pragma once
#include "device_common.h"

namespace fdh {

constexpr int kLitBits = 12;
constexpr int kLitSize = 1 << kLitBits;
constexpr int kDistBits = 9;
constexpr int kDistSize = 1 << kDistBits;
constexpr int kClBits = 7;
constexpr int kClSize = 1 << kClBits;

enum : uint32_t { K_LIT1 = 0, K_LIT2 = 1, K_LEN = 2, K_EOB = 3, K_LONG = 4 };
enum : uint32_t { D_INVALID = 0, D_DIST = 1, D_LONG = 2 };

// Canonical-code bookkeeping kept next to a table for long-code walks.
struct CodeBook {
    uint32_t hist[16];   // symbols per code length (hist[0] unused, kept 0)
    uint32_t first[16];  // first canonical code of each length (MSB-first)
    uint32_t offs[16];   // index of the first symbol of each length in `sorted`
    uint32_t run[16];    // scratch: symbols of each length placed so far
};

// Extra bits / base of a length symbol (i = symbol - 257) and of a distance symbol, computed: the
// RFC-1951 tables (kLenBase / kLenExtra / kDistBase / kDistExtra, reference src/tables.rs:68-88) are
// arithmetic progressions, and a table in memory is a trip to the cache in the middle of a token.
__device__ __forceinline__ uint32_t len_extra_base(uint32_t i) {  // extra | base << 8
    const uint32_t e = (i < 8 || i == 28) ? 0u : (i >> 2) - 1;
    const uint32_t b = i == 28 ? 258u : (i < 8 ? 3 + i : 3 + ((4 + (i & 3)) << e));
    return e | (b << 8);
}
__device__ __forceinline__ uint32_t dist_extra_base(uint32_t sym) {  // extra | base << 8
    const uint32_t e = sym < 4 ? 0u : (sym >> 1) - 1;
    const uint32_t b = sym < 4 ? sym + 1 : 1 + ((2 + (sym & 1)) << e);
    return e | (b << 8);
}

template <int LB>
struct LitlenTraitsT {
    static constexpr int kBits = LB;
    __device__ static uint32_t entry(uint32_t sym, uint32_t nb) {
        if (sym < 256) return nb | (K_LIT1 << 4) | (sym << 8) | (nb << 24);
        if (sym == 256 || sym >= 286) return nb | (K_EOB << 4);  // 286/287: parity trap 1
        return nb | (K_LEN << 4) | (len_extra_base(sym - 257) << 8);  // extra in [12:8], base in [31:16]
    }
    __device__ static uint32_t long_entry() { return K_LONG << 4; }
};
using LitlenTraits = LitlenTraitsT<kLitBits>;
struct DistTraits {
    static constexpr int kBits = kDistBits;
    __device__ static uint32_t entry(uint32_t sym, uint32_t nb) {
        if (sym >= 30) return nb | (D_INVALID << 4);  // tables.rs:130-140: slots 30,31 are 0
        return nb | (D_DIST << 4) | (dist_extra_base(sym) << 8);  // extra in [11:8], base in [31:16]
    }
    __device__ static uint32_t long_entry() { return D_LONG << 4; }
};
struct ClTraits {
    static constexpr int kBits = kClBits;
    __device__ static uint32_t entry(uint32_t sym, uint32_t nb) { return nb | (sym << 8); }
    __device__ static uint32_t long_entry() { return 0; }
};

enum BuildResult : int { BUILD_OK = 0, BUILD_INCOMPLETE = 1 };

// Builds `table` (1 << Traits::kBits entries, LDS) from `lens[0..n)` (LDS).  All 64 lanes call.
// IS_DIST adds the two distance-only cases of src/huffman.rs:40-59.
template <class Traits, bool IS_DIST>
__device__ __forceinline__ int build_table(uint32_t* table, const uint8_t* lens, int n, CodeBook& cb,
                           uint16_t* sorted, int lane) {
    constexpr int PB = Traits::kBits;
    constexpr int TSIZE = 1 << PB;
    if (lane < 16) {
        cb.hist[lane] = 0;
        cb.run[lane] = 0;
    }
    wave_sync();
    for (int s = lane; s < n; s += kWave) {
        uint32_t l = lens[s];
        if (l) atomicAdd(&cb.hist[l], 1u);
    }
    wave_sync();
    // Kraft sum, maximum length, first codes and sorted offsets (uniform, every lane).
    uint32_t kraft = 0, nsyms = 0, max_len = 0, code = 0, off = 0, prev = 0;
    for (int l = 1; l <= 15; l++) {
        uint32_t h = uni(cb.hist[l]);
        kraft += h << (15 - l);
        nsyms += h;
        if (h) max_len = l;
        code = (code + prev) << 1;
        if (lane == 0) {
            cb.first[l] = code;
            cb.offs[l] = off;
        }
        off += h;
        prev = h;
    }
    wave_sync();
    if (IS_DIST) {
        if (nsyms == 0) {  // src/decompress.rs:588-589 (and src/huffman.rs:41-44)
            for (int i = lane; i < TSIZE; i += kWave) table[i] = D_INVALID << 4;
            wave_sync();
            return BUILD_OK;
        }
        if (max_len == 1 && nsyms == 1) {  // src/huffman.rs:45-58: one 1-bit code
            uint32_t sym = 0;
            for (int s = lane; s < n; s += kWave) {
                if (lens[s] == 1) sym = (uint32_t)s;
            }
            sym = wave_sum_u32(sym);
            uint32_t e = Traits::entry(sym, 1);
            for (int i = lane; i < TSIZE; i += kWave) table[i] = (i & 1) ? (D_INVALID << 4) : e;
            if (lane == 0) sorted[0] = (uint16_t)sym;
            wave_sync();
            return BUILD_OK;
        }
    }
    if (kraft != (1u << 15)) return BUILD_INCOMPLETE;  // src/huffman.rs:72-75

    // Counting sort by (length, symbol) + primary-table fill, one symbol per lane per round.
    for (int base = 0; base < n; base += kWave) {
        int s = base + lane;
        uint32_t l = (s < n) ? lens[s] : 0;
        uint32_t rank = 0;
        uint64_t todo = __ballot(l != 0);
        while (todo) {
            int leader = __ffsll((unsigned long long)todo) - 1;
            uint32_t ll = __shfl(l, leader, kWave);
            uint64_t m = __ballot(l == ll);
            uint32_t before = uni(cb.run[ll]);
            if (l == ll) rank = before + __popcll(m & lanemask_lt(lane));
            wave_sync();
            if (lane == leader) cb.run[ll] = before + __popcll(m);
            wave_sync();
            todo &= ~m;
        }
        if (l) {
            uint32_t cw = cb.first[l] + rank;
            sorted[cb.offs[l] + rank] = (uint16_t)s;
            uint32_t rev = __brev(cw) >> (32 - l);
            if ((int)l <= PB) {
                uint32_t e = Traits::entry((uint32_t)s, l);
                for (uint32_t idx = rev; idx < (uint32_t)TSIZE; idx += (1u << l)) table[idx] = e;
            } else {
                table[rev & (TSIZE - 1)] = Traits::long_entry();
            }
        }
    }
    wave_sync();
    // The placement counters have done their work: run[l] now holds the left-justified 16-bit bound of the
    // codes of length <= l (canonical codes of one length are consecutive and the lengths follow each
    // other), which is what long_walk counts against.
    if (lane >= 1 && lane < 16) cb.run[lane] = (cb.first[lane] + cb.hist[lane]) << (16 - lane);
    wave_sync();
    return BUILD_OK;
}

// Second pass for the litlen table: turn single-literal entries into double-literal entries
// wherever the next symbol is also a literal and both codes fit in kLitBits bits.
template <int LB = kLitBits>
__device__ __forceinline__ void add_double_literals(uint32_t* table, int lane) {
    for (int idx = lane; idx < (1 << LB); idx += kWave) {
        uint32_t e1 = table[idx];
        uint32_t k1 = (e1 >> 4) & 15;
        uint32_t n1 = (k1 == K_LIT1) ? (e1 >> 24) : 0;  // [27:24] = bits of the first symbol
        uint32_t e2 = table[(uint32_t)idx >> n1];
        uint32_t k2 = (e2 >> 4) & 15;
        uint32_t n2 = e2 >> 24;
        wave_sync();  // every lane has read before any lane rewrites an entry
        if (k1 == K_LIT1 && (k2 == K_LIT1 || k2 == K_LIT2) && n1 + n2 <= (uint32_t)LB) {
            uint32_t s1 = (e1 >> 8) & 0xFF, s2 = (e2 >> 8) & 0xFF;
            table[idx] = (n1 + n2) | (K_LIT2 << 4) | (s1 << 8) | (s2 << 16) | (n1 << 24);
        }
    }
    wave_sync();
}

// Canonical decode of a code longer than the primary index (the reference's secondary
// tables, src/huffman.rs:138-181).  `bits` holds >= 15 stream bits, LSB first.  Uniform.
__device__ __forceinline__ void long_decode(const CodeBook& cb, const uint16_t* sorted, uint64_t bits,
                                   uint32_t& sym, uint32_t& nbits) {
    uint32_t code = 0, first = 0, index = 0;
    sym = 0;
    nbits = 15;
    for (int len = 1; len <= 15; len++) {
        code |= (uint32_t)(bits >> (len - 1)) & 1u;
        uint32_t count = uni(cb.hist[len]);
        if (code - first < count) {
            sym = uni(sorted[index + (code - first)]);
            nbits = len;
            return;
        }
        index += count;
        first = (first + count) << 1;
        code <<= 1;
    }
}

// Per-lane variant for codes longer than the primary index (lengths min_len..15): `w` holds the
// next >= 15 stream bits of this lane, LSB first.  False if no code matches (cannot happen for a
// complete code).
__device__ __forceinline__ bool long_walk(const CodeBook& cb, const uint16_t* sorted, uint32_t w, int min_len,
                                          uint32_t& sym, uint32_t& nbits) {
    // Canonical codes compared MSB first: the length of the code in front of r16 is the number of
    // bounds (CodeBook::run after build_table) it has passed -- no loop with an exit per lane, the
    // lanes of a wavefront sit on codes of different lengths.
    const uint32_t r16 = __brev(w) >> 16;
    uint32_t len = (uint32_t)min_len;
#pragma unroll
    for (int l = min_len; l < 15; l++) len += r16 >= uni(cb.run[l]) ? 1u : 0u;
    if (r16 >= uni(cb.run[15])) return false;
    const uint32_t d = (r16 >> (16 - len)) - cb.first[len];
    sym = sorted[cb.offs[len] + d];
    nbits = len;
    return true;
}

}  // namespace fdh
