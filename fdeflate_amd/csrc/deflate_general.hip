// deflate_general.hip -- the general encoder at level 1 (`compress_to_vec`) and in RLE mode
// (`compress_to_vec_rle`), bit-exact with the reference, ONE STREAM PER LANE.
//
// Reference: Compressor src/compress/mod.rs:47-217 (level 1 = GreedyParser + HashTableMatchFinder
// :76, RLE = RleParser :107-123), parsers src/compress/parse/{mod,greedy,rle}.rs, match finders
// src/compress/matchfinder/{mod,hashtable}.rs, block writer src/compress/bitstream.rs:41-325,
// bit writer src/compress/bitwriter.rs.
//
// LZ77 parsing with a hash table is order-dependent inside a stream (every decision depends on
// what the table held, i.e. on all earlier decisions), so there is nothing for the other 63 lanes
// of a wavefront to do on the same stream that would not change the bytes.  The batch has tens of
// thousands of independent streams instead: every lane runs the whole sequential algorithm on its
// own stream (64 streams per wavefront, the wavefronts hide one another's memory latency).  State
// that is indexed by data (the 64 Ki-entry hash table, the symbol list of the open block) lives in
// a per-stream slice of a global workspace; the small per-block arrays (histograms, code lengths,
// the Huffman heap) are interleaved [index][lane] in the same workspace so that lanes walking them
// in step touch consecutive addresses.  Nothing is kept in per-lane scratch arrays.
//
// The tie-breaking of the Huffman construction follows Rust's BinaryHeap exactly as the reference
// uses it (see the oracle's section header for what is pinned and what is not).
#include "device_common.h"

namespace fdh {

// ---- constant tables (RFC 1951; reference src/tables.rs:28-88, data) -----------------------
__device__ static const uint8_t kGDistLookup[16] = {0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7};
__device__ static const uint32_t kGBitmask[17] = {0x0000, 0x0001, 0x0003, 0x0007, 0x000F, 0x001F, 0x003F, 0x007F, 0x00FF,
                                                  0x01FF, 0x03FF, 0x07FF, 0x0FFF, 0x1FFF, 0x3FFF, 0x7FFF, 0xFFFF};

// length - 3 -> (symbol - 257, extra bits): LENGTH_TO_SYMBOL / LENGTH_TO_LEN_EXTRA (tables.rs:28-55)
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

constexpr uint32_t kGMaxSymbols = 16384 + 8;
constexpr uint32_t kGHashSize = 1u << 16;

struct GSym {
    uint32_t a;  // literal run: start          | back-reference: 0x80000000 | length
    uint32_t b;  // literal run: end            | back-reference: distance | dist_sym << 16
};

// Per-stream slice of the workspace (data-indexed state).
struct GStreamWork {
    uint32_t hash[kGHashSize];
    GSym symbols[kGMaxSymbols];
};
// Per-wavefront block of small arrays, interleaved [index][lane].
struct GWaveWork {
    uint32_t freq[286][kWave], dfreq[30][kWave], clfreq[19][kWave];
    uint8_t lengths[286][kWave], dlengths[30][kWave], cllengths[19][kWave];
    uint16_t codes[286][kWave], dcodes[30][kWave], clcodes[19][kWave];
    uint32_t heap_f[286][kWave];
    uint16_t heap_i[286][kWave];
    uint16_t in_left[286][kWave], in_right[286][kWave];
    uint16_t stack_node[600][kWave];
    uint8_t stack_depth[600][kWave];
    uint16_t order[286][kWave];
    uint32_t counts[16][kWave];
};

struct GBitWriter {  // bitwriter.rs:3-51 over a bounded slot
    uint64_t buffer;
    uint32_t nbits;
    uint8_t* out;
    uint64_t cap, pos;
    bool overflow;
    __device__ void raw(const void* p, uint32_t n) {
        if (pos + n > cap) {
            overflow = true;
            return;
        }
        const uint8_t* s = static_cast<const uint8_t*>(p);
        for (uint32_t i = 0; i < n; i++) out[pos + i] = s[i];
        pos += n;
    }
    __device__ void write_bits(uint64_t bits, uint32_t n) {
        buffer |= bits << nbits;
        nbits += n;
        if (nbits >= 64) {
            if (pos + 8 > cap) {
                overflow = true;
            } else {
                for (int i = 0; i < 8; i++) out[pos + i] = (uint8_t)(buffer >> (8 * i));
                pos += 8;
            }
            nbits -= 64;
            const uint32_t sh = n - nbits;
            buffer = sh >= 64 ? 0 : bits >> sh;
        }
    }
    __device__ void flush() {
        if (nbits % 8 != 0) write_bits(0, 8 - nbits % 8);
        if (nbits > 0) {
            const uint32_t n = nbits / 8;
            if (pos + n > cap) {
                overflow = true;
            } else {
                for (uint32_t i = 0; i < n; i++) out[pos + i] = (uint8_t)(buffer >> (8 * i));
                pos += n;
            }
            buffer = 0;
            nbits = 0;
        }
    }
};

__device__ __forceinline__ uint64_t g_load64(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ uint32_t g_hash(uint64_t v) { return (uint32_t)((11400714785074694791ull * v) >> 40) & (kGHashSize - 1); }

struct GMatch {
    uint32_t length, distance;
    uint64_t start;
    __device__ uint64_t end() const { return start + length; }
};

// ---- build_huffman_tree (bitstream.rs:198-325) on interleaved arrays -----------------------
// Ord of the heap items: `other.0.cmp(&self.0)`: a <= b  <=>  a.f >= b.f (ties decided by the
// std::collections::BinaryHeap algorithms restated below).
template <int N>
struct GHuff {
    uint32_t (*freq)[kWave];
    uint8_t (*lengths)[kWave];
    uint16_t (*codes)[kWave];
    GWaveWork* w;
    int lane;

    __device__ void sift_down_range(uint32_t pos, uint32_t end) {
        const uint32_t ef = w->heap_f[pos][lane];
        const uint16_t ei = w->heap_i[pos][lane];
        uint32_t hole = pos, child = 2 * hole + 1;
        const uint32_t lim = end >= 2 ? end - 2 : 0;
        while (child <= lim) {
            if (w->heap_f[child][lane] >= w->heap_f[child + 1][lane]) child++;  // d[child] <= d[child + 1]
            if (ef <= w->heap_f[child][lane]) {                                 // elem >= d[child]
                w->heap_f[hole][lane] = ef;
                w->heap_i[hole][lane] = ei;
                return;
            }
            w->heap_f[hole][lane] = w->heap_f[child][lane];
            w->heap_i[hole][lane] = w->heap_i[child][lane];
            hole = child;
            child = 2 * hole + 1;
        }
        if (child == end - 1 && ef > w->heap_f[child][lane]) {  // elem < d[child]
            w->heap_f[hole][lane] = w->heap_f[child][lane];
            w->heap_i[hole][lane] = w->heap_i[child][lane];
            hole = child;
        }
        w->heap_f[hole][lane] = ef;
        w->heap_i[hole][lane] = ei;
    }
    // BinaryHeap::pop: swap the last item into the root, sift_down_to_bottom(0), sift_up
    __device__ void pop(uint32_t& len, uint32_t& f, uint16_t& idx) {
        uint32_t lf = w->heap_f[len - 1][lane];
        uint16_t li = w->heap_i[len - 1][lane];
        len--;
        if (len == 0) {
            f = lf;
            idx = li;
            return;
        }
        f = w->heap_f[0][lane];
        idx = w->heap_i[0][lane];
        const uint32_t end = len;
        uint32_t hole = 0, child = 1;
        const uint32_t lim = end >= 2 ? end - 2 : 0;
        while (child <= lim) {
            if (w->heap_f[child][lane] >= w->heap_f[child + 1][lane]) child++;
            w->heap_f[hole][lane] = w->heap_f[child][lane];
            w->heap_i[hole][lane] = w->heap_i[child][lane];
            hole = child;
            child = 2 * hole + 1;
        }
        if (child == end - 1) {
            w->heap_f[hole][lane] = w->heap_f[child][lane];
            w->heap_i[hole][lane] = w->heap_i[child][lane];
            hole = child;
        }
        while (hole > 0) {  // sift_up(0, hole)
            const uint32_t parent = (hole - 1) / 2;
            if (lf >= w->heap_f[parent][lane]) break;  // elem <= d[parent]
            w->heap_f[hole][lane] = w->heap_f[parent][lane];
            w->heap_i[hole][lane] = w->heap_i[parent][lane];
            hole = parent;
        }
        w->heap_f[hole][lane] = lf;
        w->heap_i[hole][lane] = li;
    }

    __device__ void build(uint32_t length_limit) {
        uint32_t used = 0, first = 0;
        for (int i = 0; i < N; i++) {
            lengths[i][lane] = 0;
            codes[i][lane] = 0;
            if (freq[i][lane] > 0) {
                if (used == 0) first = (uint32_t)i;
                used++;
            }
        }
        if (used <= 1) {  // :206-213
            if (used == 1) lengths[first][lane] = 1;
            return;
        }
        uint32_t hl = 0, ni = 0;
        for (int i = 0; i < N; i++) {
            const uint32_t f = freq[i][lane];
            if (f > 0) {
                w->heap_f[hl][lane] = f;
                w->heap_i[hl][lane] = (uint16_t)i;
                hl++;
            }
        }
        for (uint32_t k = hl / 2; k > 0;) {  // BinaryHeap::from(vec): rebuild
            k--;
            sift_down_range(k, hl);
        }
        while (hl > 1) {  // :236-244
            uint32_t f1;
            uint16_t i1;
            pop(hl, f1, i1);
            w->in_left[ni][lane] = i1;
            w->in_right[ni][lane] = w->heap_i[0][lane];
            ni++;
            w->heap_f[0][lane] = f1 + w->heap_f[0][lane];
            w->heap_i[0][lane] = (uint16_t)(ni + N - 1);
            sift_down_range(0, hl);  // PeekMut::drop
        }
        // :247-259 walk the tree
        uint32_t sp = 0;
        w->stack_node[0][lane] = w->heap_i[0][lane];
        w->stack_depth[0][lane] = 0;
        sp = 1;
        uint32_t max_length = 0;
        while (sp > 0) {
            sp--;
            const uint32_t node = w->stack_node[sp][lane];
            const uint32_t depth = w->stack_depth[sp][lane];
            if (node < (uint32_t)N) {
                lengths[node][lane] = (uint8_t)depth;
                max_length = max(max_length, depth);
            } else {
                w->stack_node[sp][lane] = w->in_left[node - N][lane];
                w->stack_depth[sp][lane] = (uint8_t)(depth + 1);
                sp++;
                w->stack_node[sp][lane] = w->in_right[node - N][lane];
                w->stack_depth[sp][lane] = (uint8_t)(depth + 1);
                sp++;
            }
        }
        if (max_length > length_limit) {  // :262-305
            for (int i = 0; i < 16; i++) w->counts[i][lane] = 0;
            for (int i = 0; i < N; i++) w->counts[min((uint32_t)lengths[i][lane], length_limit)][lane]++;
            uint32_t total = 0;
            for (uint32_t i = 1; i <= length_limit; i++) total += w->counts[i][lane] << (length_limit - i);
            while (total > (1u << length_limit)) {
                uint32_t i = length_limit - 1;
                while (w->counts[i][lane] == 0) i--;
                w->counts[i][lane]--;
                w->counts[length_limit][lane]--;
                w->counts[i + 1][lane] += 2;
                total--;
            }
            // by frequency, ties in index order (insertion sort = what sort_unstable does up to 20
            // elements; beyond that the reference's tie order is implementation-defined)
            for (int i = 0; i < N; i++) w->order[i][lane] = (uint16_t)i;
            for (int i = 1; i < N; i++) {
                const uint16_t v = w->order[i][lane];
                const uint32_t fv = freq[v][lane];
                int j = i;
                while (j > 0 && freq[w->order[j - 1][lane]][lane] > fv) {
                    w->order[j][lane] = w->order[j - 1][lane];
                    j--;
                }
                w->order[j][lane] = v;
            }
            uint32_t len = length_limit;
            for (int k = 0; k < N; k++) {
                const uint32_t i = w->order[k][lane];
                if (freq[i][lane] > 0) {
                    while (w->counts[len][lane] == 0) len--;
                    lengths[i][lane] = (uint8_t)len;
                    w->counts[len][lane]--;
                }
            }
        }
        uint32_t code = 0;  // :308-320 canonical codes, bit-reversed
        for (uint32_t len = 1; len <= length_limit; len++) {
            for (int i = 0; i < N; i++) {
                if (lengths[i][lane] == len) {
                    codes[i][lane] = (uint16_t)(__brev(code) >> (32 - len));
                    code++;
                }
            }
            code <<= 1;
        }
    }
};

struct GParser {
    GStreamWork* sw;
    GWaveWork* ww;
    int lane;
    uint32_t nsym;
    uint64_t ip, last_match, last_block_end;
    uint32_t last_index;
    GMatch m;

    // distance_to_dist_sym (bitstream.rs:16-27)
    __device__ static uint32_t dist_sym_of(uint32_t distance) {
        if (distance <= 16) return kGDistLookup[distance - 1];
        uint32_t s = 29;
        while (s > 0 && distance < kDistBase[s]) s--;
        return s;
    }

    // write_block (bitstream.rs:41-195)
    __device__ void write_block(GBitWriter& bw, const uint8_t* data, uint32_t base_index, bool eof) {
        for (int i = 0; i < 286; i++) ww->freq[i][lane] = 0;
        for (int i = 0; i < 30; i++) ww->dfreq[i][lane] = 0;
        for (int i = 0; i < 19; i++) ww->clfreq[i][lane] = 0;
        ww->freq[256][lane] = 1;
        for (uint32_t k = 0; k < nsym; k++) {
            const GSym s = sw->symbols[k];
            if (s.a & 0x80000000u) {
                uint32_t sym, extra;
                g_length_symbol(s.a & 0xFFFF, sym, extra);
                ww->freq[sym][lane]++;
                ww->dfreq[s.b >> 16][lane]++;
            } else {
                for (uint32_t p = s.a - base_index; p < s.b - base_index; p++) ww->freq[data[p]][lane]++;
            }
        }
        GHuff<286> hl{ww->freq, ww->lengths, ww->codes, ww, lane};
        hl.build(15);
        GHuff<30> hd{ww->dfreq, ww->dlengths, ww->dcodes, ww, lane};
        hd.build(15);
        uint32_t num_litlen = 286, num_dist = 30;
        while (num_litlen > 257 && ww->lengths[num_litlen - 1][lane] == 0) num_litlen--;
        while (num_dist > 1 && ww->dlengths[num_dist - 1][lane] == 0) num_dist--;
        for (uint32_t i = 0; i < num_litlen; i++) ww->clfreq[ww->lengths[i][lane]][lane]++;
        for (uint32_t i = 0; i < num_dist; i++) ww->clfreq[ww->dlengths[i][lane]][lane]++;
        GHuff<19> hc{ww->clfreq, ww->cllengths, ww->clcodes, ww, lane};
        hc.build(7);

        bw.write_bits(eof ? 5 : 4, 3);
        bw.write_bits(num_litlen - 257, 5);
        bw.write_bits(num_dist - 1, 5);
        bw.write_bits(15, 4);
        for (int j = 0; j < 19; j++) bw.write_bits(ww->cllengths[kClclOrder[j]][lane], 3);
        for (uint32_t i = 0; i < num_litlen; i++) {
            const uint32_t l = ww->lengths[i][lane];
            bw.write_bits(ww->clcodes[l][lane], ww->cllengths[l][lane]);
        }
        for (uint32_t i = 0; i < num_dist; i++) {
            const uint32_t l = ww->dlengths[i][lane];
            bw.write_bits(ww->clcodes[l][lane], ww->cllengths[l][lane]);
        }
        for (uint32_t k = 0; k < nsym; k++) {
            const GSym s = sw->symbols[k];
            if (s.a & 0x80000000u) {
                const uint32_t length = s.a & 0xFFFF, distance = s.b & 0xFFFF, ds = s.b >> 16;
                uint32_t sym, extra;
                g_length_symbol(length, sym, extra);
                bw.write_bits(ww->codes[sym][lane], ww->lengths[sym][lane]);
                bw.write_bits((length - 3) & kGBitmask[extra], extra);
                bw.write_bits(ww->dcodes[ds][lane], ww->dlengths[ds][lane]);
                bw.write_bits(distance - kDistBase[ds], kDistExtra[ds]);
            } else {
                // (the reference packs four literals per write_bits, :134-160: the byte stream is the same)
                for (uint32_t p = s.a - base_index; p < s.b - base_index; p++) {
                    const uint32_t c = data[p];
                    bw.write_bits(ww->codes[c][lane], ww->lengths[c][lane]);
                }
            }
        }
        bw.write_bits(ww->codes[256][lane], ww->lengths[256][lane]);
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
        while (length < 258 && ip > anchor && prev_index > 0 && data[ip - 1] == data[prev_index - 1]) {
            length++;
            ip--;
            prev_index--;
        }
        uint64_t slice = len - ip - length;
        if (slice > 258 - length) slice = 258 - length;
        const uint8_t *a = data + ip + length, *b = data + prev_index + length;
        uint64_t k = 0;
        bool done = false;
        for (; k + 8 <= slice; k += 8) {
            const uint64_t x = g_load64(a + k), y = g_load64(b + k);
            if (x == y) {
                length += 8;
            } else {
                length += (uint64_t)__builtin_ctzll(x ^ y) / 8;
                done = true;
                break;
            }
        }
        if (!done) {
            for (; k < slice; k++) {
                if (a[k] != b[k]) break;
                length++;
            }
        }
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
        while (r.start > min_start && data[r.start - 2] == value) {
            r.start--;
            r.length++;
        }
        const uint8_t* p = data + r.end();
        uint64_t n = len - r.end();
        if (n > 258 - r.length) n = 258 - r.length;
        const uint64_t v8 = 0x0101010101010101ull * value;
        uint64_t k = 0;
        for (; k + 8 <= n; k += 8) {
            const uint64_t c = g_load64(p + k);
            if (c != v8) {
                r.length += (uint32_t)(__builtin_ctzll(c ^ v8) / 8);
                return r;
            }
            r.length += 8;
        }
        for (; k < n; k++) {
            if (p[k] != value) break;
            r.length++;
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
            const uint32_t offset = sw->hash[h];
            sw->hash[h] = (uint32_t)ip + base_index;
            if (offset >= min_offset) {
                uint32_t l;
                uint64_t st;
                match_length8(current, data, len, anchor, ip, (uint64_t)(offset - base_index), l, st);
                if (l >= 8) r = GMatch{l, (uint32_t)(ip - (uint64_t)(offset - base_index)), st};
            }
            if (fizzle) {
                while (r.length < 258 && r.start > last_match && r.start > (uint64_t)r.distance + 1 &&
                       data[r.start - 1] == data[r.start - r.distance - 1]) {
                    r.length++;
                    r.start--;
                }
            }
        }
        ip++;
        return r;
    }

    template <bool RLE>
    __device__ GMatch advance_to_match(const uint8_t* data, uint64_t len, uint32_t base_index, uint64_t max_ip) {
        while (ip < max_ip) {
            const GMatch r = get_match<RLE>(data, len, base_index, false);
            if (r.length != 0) return r;
            ip += (ip - last_match) >> 5;  // skip_ahead_shift = 5 (compress/mod.rs:76, :114)
        }
        return GMatch{0, 0, 0};
    }

    template <bool RLE>
    __device__ void advance(const uint8_t* data, uint64_t len, uint32_t base_index, uint64_t end) {
        if (!RLE) {
            const uint64_t stop = min(end, len - 8);
            for (uint64_t j = ip; j < stop; j++) sw->hash[g_hash(g_load64(data + j))] = base_index + (uint32_t)j;
        }
        ip = max(ip, end);
    }

    __device__ void insert_match(uint32_t base_index, const GMatch& r) {
        if (r.start > last_match) {
            sw->symbols[nsym] = GSym{base_index + (uint32_t)last_match, base_index + (uint32_t)r.start};
            nsym++;
        }
        sw->symbols[nsym] = GSym{0x80000000u | r.length, r.distance | (dist_sym_of(r.distance) << 16)};
        nsym++;
        last_match = r.end();
    }

    __device__ void write_block_if_ready(GBitWriter& bw, const uint8_t* data, uint64_t len, uint32_t base_index, bool finish) {
        if (nsym >= 16384) {
            write_block(bw, data, base_index, finish && last_match == len);
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

    __device__ uint64_t end_compress(GBitWriter& bw, const uint8_t* data, uint64_t len, uint32_t base_index, uint64_t start,
                                     bool finish) {
        if (finish && (nsym != 0 || last_match < len)) {
            ip = min(ip, len);
            if (last_match < len) {
                sw->symbols[nsym] = GSym{base_index + (uint32_t)last_match, base_index + (uint32_t)len};
                nsym++;
                ip = len;
                last_match = len;
            }
            write_block(bw, data, base_index, true);
            nsym = 0;
            last_block_end = ip;
        }
        return last_block_end - start;
    }

    // CompressorInner::compress (compress/mod.rs:226-290) -> GreedyParser::compress
    // (parse/greedy.rs:27-91) / RleParser::compress (parse/rle.rs:22-47)
    template <bool RLE>
    __device__ uint64_t compress(GBitWriter& bw, const uint8_t* data, uint64_t len, uint32_t base_index, uint64_t start,
                                 bool finish) {
        if (finish && len == start) {  // :234-238
            bw.write_bits(3, 10);
            bw.flush();
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
                write_block_if_ready(bw, data, len, base_index, finish);
            }
        } else {
            for (;;) {
                if (m.length == 0) {
                    m = advance_to_match<false>(data, len, base_index, max_ip);
                    if (m.length == 0) break;
                }
                advance<false>(data, len, base_index, m.end());
                GMatch m2{0, 0, 0};
                if (ip < max_ip) {
                    m2 = get_match<false>(data, len, base_index, true);
                } else if (!finish) {
                    break;
                }
                if (m2.length == 0 || m2.start > m.start + 1) {
                    insert_match(base_index, m);
                    write_block_if_ready(bw, data, len, base_index, finish);
                    if (m2.length != 0 && m2.start < last_match) {
                        m2.length -= (uint32_t)(last_match - m2.start);
                        m2.start = last_match;
                        if (m2.length < 4) m2 = GMatch{0, 0, 0};
                    }
                }
                m = m2;
            }
        }
        return end_compress(bw, data, len, base_index, start, finish);
    }
};

struct DeflateGeneralArgs {
    const uint8_t* in;
    const uint64_t* in_off;
    uint8_t* out;
    const uint64_t* out_off;
    uint32_t* out_len;
    uint64_t n;
    GStreamWork* stream_work;  // one per resident lane
    GWaveWork* wave_work;      // one per resident wavefront
};

// Adler-32 of a buffer, one lane (RFC 1950; the reference's simd_adler32 call site compress/mod.rs:137-139)
__device__ uint32_t g_adler32(const uint8_t* p, uint64_t n) {
    uint32_t a = 1, b = 0;
    while (n) {
        uint32_t k = n > 5552 ? 5552u : (uint32_t)n;
        n -= k;
        for (; k; k--) {
            a += *p++;
            b += a;
        }
        a %= kAdlerMod;
        b %= kAdlerMod;
    }
    return (b << 16) | a;
}

// One stream per lane; resident lanes walk the batch with a grid stride.
template <bool RLE>
__global__ __launch_bounds__(kWave) void deflate_general_kernel(DeflateGeneralArgs a) {
    const int lane = threadIdx.x;
    GStreamWork* sw = a.stream_work + ((uint64_t)blockIdx.x * kWave + lane);
    GWaveWork* ww = a.wave_work + blockIdx.x;
    for (uint64_t sid0 = (uint64_t)blockIdx.x * kWave; sid0 < a.n; sid0 += (uint64_t)gridDim.x * kWave) {
        if (!RLE) {
            // HashTableMatchFinder::new (hashtable.rs:10-14): all 64 tables of the wavefront are
            // cleared by the whole wavefront (coalesced 16-B stores), not lane by lane
            for (int s = 0; s < kWave; s++) {
                uint4* t = reinterpret_cast<uint4*>(a.stream_work[(uint64_t)blockIdx.x * kWave + s].hash);
                for (uint32_t i = lane; i < kGHashSize / 4; i += kWave) t[i] = make_uint4(0, 0, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const uint64_t sid = sid0 + lane;
        if (sid >= a.n) continue;
        const uint8_t* input = a.in + a.in_off[sid];
        const uint64_t len = a.in_off[sid + 1] - a.in_off[sid];
        if (len > (1ull << 30)) {  // write_data splits above 1 GiB (compress/mod.rs:130-136): not supported
            a.out_len[sid] = 0xFFFFFFFFu;
            continue;
        }
        GBitWriter bw{0, 0, a.out + a.out_off[sid], a.out_off[sid + 1] - a.out_off[sid], 0, false};
        const uint8_t hdr[2] = {0x78, 0x01};
        bw.raw(hdr, 2);
        GParser ps;
        ps.sw = sw;
        ps.ww = ww;
        ps.lane = lane;
        ps.nsym = 0;
        ps.ip = ps.last_match = ps.last_block_end = 0;
        ps.last_index = 0;
        ps.m = GMatch{0, 0, 0};
        const uint64_t window = RLE ? 1 : 32768;
        // Compressor::write_data (compress/mod.rs:126-159, no buffered input) ...
        const uint64_t written = ps.compress<RLE>(bw, input, len, 0, 0, false);
        const uint64_t start = written > window ? written - window : 0;
        // ... and Compressor::finish (:194-214) over the kept tail input.data = data[start..]
        ps.compress<RLE>(bw, input + start, len - start, (uint32_t)start, written - start, true);
        bw.flush();
        const uint32_t ad = g_adler32(input, len);
        const uint8_t tr[4] = {(uint8_t)(ad >> 24), (uint8_t)(ad >> 16), (uint8_t)(ad >> 8), (uint8_t)ad};
        bw.raw(tr, 4);
        a.out_len[sid] = bw.overflow ? 0xFFFFFFFFu : (uint32_t)bw.pos;
    }
}

}  // namespace fdh

extern "C" size_t fdh_deflate_general_stream_work_bytes(void) { return sizeof(fdh::GStreamWork); }
extern "C" size_t fdh_deflate_general_wave_work_bytes(void) { return sizeof(fdh::GWaveWork); }

extern "C" int fdh_launch_deflate_general(const uint8_t* in, const uint64_t* in_off, uint8_t* out, const uint64_t* out_off,
                                          uint32_t* out_len, uint64_t n, int rle, void* stream_work, void* wave_work,
                                          unsigned waves, hipStream_t stream) {
    if (n == 0) return 0;
    fdh::DeflateGeneralArgs a{in, in_off, out, out_off, out_len, n, static_cast<fdh::GStreamWork*>(stream_work),
                              static_cast<fdh::GWaveWork*>(wave_work)};
    if (rle)
        hipLaunchKernelGGL(fdh::deflate_general_kernel<true>, dim3(waves), dim3(fdh::kWave), 0, stream, a);
    else
        hipLaunchKernelGGL(fdh::deflate_general_kernel<false>, dim3(waves), dim3(fdh::kWave), 0, stream, a);
    return (int)hipGetLastError();
}
