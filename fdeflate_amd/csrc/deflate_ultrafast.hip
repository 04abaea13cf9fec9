// deflate_ultrafast.hip -- batched ultra-fast (RLE-of-zeros, fixed Huffman table) zlib encoder.
//
// Restates UltraFastCompressor (reference src/compress/ultrafast.rs:9-182) for one wavefront
// per input buffer.  The reference walks 8-byte chunks with a scalar `run` counter; here one
// lane owns one chunk of a 64-chunk (512 B) tile and the serial state is recovered with
// ballots:
//
//   * hz(c)  = zero bytes at the high end of chunk c (8 for an all-zero chunk)
//   * a run is pending at chunk c iff hz(c-1) > 0                      (ultrafast.rs:102-132)
//   * a non-zero chunk with a pending run closes it with its low zero bytes (`run_extra`,
//     :105-108) and emits literals for bytes [run_extra, 8 - lz) (:110-119); without a pending
//     run it emits bytes [0, 8 - lz) (:122-152) -- low zero bytes are then plain literals;
//   * the run that closes at lane j is  lz(b) + 8 * (j - b - 1) + tz(j)  with b the previous
//     non-zero chunk (or the carry from earlier tiles).
//
// Per-lane bit counts are prefix-summed over the wavefront and every lane ORs its code bits into
// a zeroed LDS bit ring at its own offset; complete 16-B lines are stored coalesced.  The
// emitted byte stream is independent of the reference's 64-bit flush granularity
// (ultrafast.rs:16-29), so bit-exactness only depends on the symbol sequence and the codes.
#include "device_common.h"
#include "bit_ring.h"

namespace fdh {

// ---- constants: reference data (src/tables.rs:7-20, src/compress/ultrafast.rs:82-86) ----
struct UfTables {
    uint32_t sym[286];  // code | len << 16, code bit-reversed as written to the stream
};

#include "uf_table_data.inc"

// compute_codes (reference src/lib.rs:103-127), evaluated at compile time like the reference's
// `const HUFFMAN_CODES` (src/tables.rs:22-25).
constexpr UfTables make_uf_tables() {
    UfTables t{};
    uint32_t code = 0;
    for (unsigned len = 1; len <= 16; len++) {
        for (int i = 0; i < 286; i++) {
            if (kHuffmanLengths[i] == len) {
                uint32_t rev = 0;
                for (unsigned b = 0; b < len; b++) rev |= ((code >> b) & 1u) << (len - 1 - b);
                t.sym[i] = rev | ((uint32_t)len << 16);
                code++;
            }
        }
        code <<= 1;
    }
    return t;
}
__device__ static const UfTables kUfTables = make_uf_tables();
static_assert(make_uf_tables().sym[0] == (0u | (2u << 16)), "HUFFMAN_CODES[0] must be 0 (ultrafast.rs:62)");

__device__ static const uint8_t kUfHeader[56] = {
    kUfHeaderData[0],  kUfHeaderData[1],  kUfHeaderData[2],  kUfHeaderData[3],  kUfHeaderData[4],  kUfHeaderData[5],  kUfHeaderData[6],
    kUfHeaderData[7],  kUfHeaderData[8],  kUfHeaderData[9],  kUfHeaderData[10], kUfHeaderData[11], kUfHeaderData[12], kUfHeaderData[13],
    kUfHeaderData[14], kUfHeaderData[15], kUfHeaderData[16], kUfHeaderData[17], kUfHeaderData[18], kUfHeaderData[19], kUfHeaderData[20],
    kUfHeaderData[21], kUfHeaderData[22], kUfHeaderData[23], kUfHeaderData[24], kUfHeaderData[25], kUfHeaderData[26], kUfHeaderData[27],
    kUfHeaderData[28], kUfHeaderData[29], kUfHeaderData[30], kUfHeaderData[31], kUfHeaderData[32], kUfHeaderData[33], kUfHeaderData[34],
    kUfHeaderData[35], kUfHeaderData[36], kUfHeaderData[37], kUfHeaderData[38], kUfHeaderData[39], kUfHeaderData[40], kUfHeaderData[41],
    kUfHeaderData[42], kUfHeaderData[43], kUfHeaderData[44], kUfHeaderData[45], kUfHeaderData[46], kUfHeaderData[47], kUfHeaderData[48],
    kUfHeaderData[49], kUfHeaderData[50], kUfHeaderData[51], kUfHeaderData[52], kUfHeaderData[53], 0, 0};
constexpr uint32_t kUfHeaderBits = 53 * 8 + 5;  // ultrafast.rs:87-88

constexpr int kEncWaves = 4;             // wavefronts (= streams) per workgroup
constexpr int kEncRingDw = 1024;         // 4 KiB bit ring per wavefront (round 5: 8 KiB kept a CU at 16 wavefronts)
constexpr uint32_t kEncRingBits = kEncRingDw * 32;
constexpr uint32_t kEncTileBudget = kEncRingBits - 1024;
using BitRing = BitRingT<kEncRingDw>;

struct EncLds {
    uint32_t tab[288];
    uint32_t tail[260];  // (round 6) run_tail of every r < 258: bits | nbits << 24 -- a look-up, not ~30 instructions per tile
    uint32_t ring[kEncWaves][kEncRingDw];
};

struct Encoder : BitRing {
    const uint32_t* tab;
    const uint32_t* tail;  // run_tail as a table (EncLds::tail)

    // write_run (ultrafast.rs:45-67) for one run, emitted by the whole wavefront (slow path
    // and the end-of-data run).
    __device__ void emit_run_uniform(uint32_t run) {
        emit_uniform(0, 2);  // literal 0
        run -= 1;
        uint32_t nrep = run / 258, r = run % 258;
        uint32_t e285 = tab[285];
        uint32_t rep_bits = e285 & 0xFFFF, rep_n = (e285 >> 16) + 1;
        while (nrep > 0) {
            uint32_t m = min(nrep, (uint32_t)kWave);
            if (qbits + (uint64_t)m * rep_n - qflushed > kEncTileBudget) flush(false);
            if ((uint32_t)lane < m) or_bits(qbits + (uint64_t)lane * rep_n, rep_bits);
            qbits += (uint64_t)m * rep_n;
            nrep -= m;
        }
        const uint32_t te = tail[r];
        emit_uniform(te & 0xFFFFFFu, te >> 24);
    }

    // tail of a run: r = (run - 1) % 258 more zeros (ultrafast.rs:54-64)
    __device__ void run_tail(uint32_t r, uint32_t& bits, uint32_t& nbits) const {
        if (r > 4) {
            uint32_t lp = r - 3, sym, ebits;
            if (r == 258) {  // unreachable (r < 258), kept for the table's sake
                sym = 285;
                ebits = 0;
            } else if (lp < 8) {
                sym = 257 + lp;
                ebits = 0;
            } else {
                ebits = (31 - __clz(lp)) - 2;
                sym = 257 + 4 * ebits + 4 + ((lp >> ebits) & 3);
            }
            uint32_t e = tab[sym];
            uint32_t clen = e >> 16;
            bits = (e & 0xFFFF) | ((lp & ((1u << ebits) - 1)) << clen);
            nbits = clen + ebits + 1;  // + the 1-bit distance code (value 0), ultrafast.rs:60
        } else {
            bits = 0;
            nbits = r * 2;  // r literal zeros, HUFFMAN_CODES[0] == 0 (ultrafast.rs:62-63)
        }
    }
};

struct DeflateBatchArgs {
    const uint8_t* in;
    const uint64_t* in_off;
    uint8_t* out;
    const uint64_t* out_off;
    uint32_t* out_len;
    uint64_t n;
    // PNG source (fdh_png_filter_deflate_ultrafast_batch): `in` holds pixel rows, the encoder's input
    // is the filtered image -- a type byte + row_bytes filtered bytes per row -- computed on the fly
    const uint8_t* types;       // one filter type per row
    const uint64_t* types_off;
    uint32_t* png_status;       // 0 ok, 1 a filter type > 4, 2 sizes do not fit
    uint32_t row_bytes, bpp;
};

// ---- PNG filtering as the encoder's source (PNG specification 9.2: filtering uses the RAW neighbours,
// so every byte is independent of every other filtered byte) ----
struct PngSource {
    const uint8_t* pix;    // rows x rb pixel bytes
    const uint8_t* types;  // rows filter types
    uint32_t rb, bpp, rows;
    __device__ __forceinline__ uint32_t predict(uint32_t t, uint32_t a, uint32_t b, uint32_t c) const {
        const int p = (int)a + (int)b - (int)c;
        const int pa = abs(p - (int)a), pb = abs(p - (int)b), pc = abs(p - (int)c);
        const uint32_t paeth = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
        return t == 1 ? a : (t == 2 ? b : (t == 3 ? (a + b) >> 1 : (t == 4 ? paeth : 0u)));
    }
    // byte `col` (0 = the type byte) of filtered row `row`
    __device__ __forceinline__ uint32_t byte_at(uint32_t row, uint32_t col) const {
        const uint32_t t = types[row];
        if (col == 0) return t;
        const uint32_t k = col - 1;
        const uint8_t* cur = pix + (uint64_t)row * rb;
        const uint32_t raw = cur[k];
        const uint32_t a = k >= bpp ? cur[k - bpp] : 0u;
        const uint32_t b = row ? cur[(int64_t)k - (int64_t)rb] : 0u;
        const uint32_t c = (row && k >= bpp) ? cur[(int64_t)k - (int64_t)rb - (int64_t)bpp] : 0u;
        return (raw - predict(t, a, b, c)) & 0xFFu;
    }
    // 8 filtered bytes from filtered offset `off` on (all inside the image), in two steps so that the
    // loads of a chunk are all unconditional and one tile ahead of their use (a load inside a branch
    // makes the compiler wait for EVERYTHING in flight at the join):
    //   request  a chunk inside one row takes four 8-byte loads and the row's type: `fast` -- eight data
    //            bytes with their neighbours --, `start` -- the type byte and the row's first seven data
    //            bytes, whose left neighbours are the row's own bytes shifted and zeros (round 6: the rows
    //            of the bench are 128 chunks, and the one at the start sent EVERY lane of every second tile
    //            through the byte-by-byte code below: half of the kernel's filtering instructions) --,
    //            `top` -- the first row, whose upper neighbours are zeros.  Any other chunk (it straddles two
    //            rows, or starts inside a row's first pixel) reads a harmless in-bounds address instead and is
    //            redone byte by byte in `finish`.
    //   finish   byte-wise arithmetic on all eight bytes at once (the type is the same for the chunk)
    struct Req {
        uint64_t rw, lw, uw, cw;
        uint32_t t, off;
        bool fast, start, top;
    };
    uint32_t magic;  // floor(2^32 / (rb + 1)): offsets are divided by the row length once per chunk
    __device__ __forceinline__ void divide(uint32_t off, uint32_t& row, uint32_t& col) const {
        const uint32_t rb1 = rb + 1;
        row = __umulhi(off, magic);  // the quotient or one less (off < 2^32, magic >= 2^32 / rb1 - 1)
        col = off - row * rb1;
        if (col >= rb1) {
            row++;
            col -= rb1;
        }
    }
    __device__ __forceinline__ Req request(uint64_t off) const {
        Req q;
        const uint32_t rb1 = rb + 1;
        const uint32_t o32 = (uint32_t)off;  // (the filtered image is shorter than 2 GiB)
        uint32_t row, col;
        divide(o32, row, col);
        q.off = o32;
        const bool inrow = col + 8 <= rb1 && row < rows;
        q.start = inrow && col == 0;  // (then rb >= 7: the load below reads one byte more, in the row or the next one / the tail)
        q.fast = inrow && col >= 1 + bpp;
        q.top = row == 0;
        const uint32_t trow = row < rows ? row : 0u;
        q.t = types[trow];
        // eight bytes must be readable at the row's start: not in the image's last seven bytes
        if (q.start && (uint64_t)row * rb + 8 > (uint64_t)rows * rb) q.start = false;
        // (neither: any address at which 8 bytes and the three neighbours are readable -- the same
        // place in the second row if there is one, else nothing is loaded)
        const bool direct = q.fast || q.start;
        const bool can = direct || (rows >= 2 && rb >= 8 + bpp);
        const uint8_t* cur = q.fast ? pix + (uint64_t)row * rb + (col - 1) : (q.start ? pix + (uint64_t)row * rb : pix + rb + bpp);
        const bool up_ok = !direct || !q.top;
        q.rw = q.lw = q.uw = q.cw = 0;
        if (can) {
            q.rw = *reinterpret_cast<const uint64_t*>(cur);  // (the hardware handles misalignment)
            q.lw = *reinterpret_cast<const uint64_t*>(q.start ? cur : cur - bpp);
            q.uw = *reinterpret_cast<const uint64_t*>(up_ok ? cur - rb : cur);
            q.cw = *reinterpret_cast<const uint64_t*>((up_ok && !q.start) ? cur - rb - bpp : cur);
        }
        return q;
    }
    __device__ __forceinline__ uint64_t finish(const Req& q, bool wanted) const {
        uint64_t rw = q.rw, lw = q.lw, uw = q.uw, cw = q.cw;
        const bool direct = q.fast || q.start;
        if (q.start) {  // [type, d0 .. d6]: the data bytes one place up, their left neighbours bpp places further
            rw <<= 8;
            uw <<= 8;
            lw = (rw << (4 * bpp)) << (4 * bpp);
            cw = (uw << (4 * bpp)) << (4 * bpp);
        }
        if (q.top) uw = cw = 0;
        const uint64_t H = 0x8080808080808080ull;
        auto sub8 = [&](uint64_t p, uint64_t r) { return ((p | H) - (r & ~H)) ^ ((p ^ ~r) & H); };  // p - r per byte
        uint64_t x = rw;
        const uint32_t t = direct ? q.t : 0u;
        if (t == 1) x = sub8(rw, lw);
        if (t == 2) x = sub8(rw, uw);
        if (t == 3) x = sub8(rw, (lw & uw) + (((lw ^ uw) >> 1) & 0x7F7F7F7F7F7F7F7Full));  // floor((a + b) / 2)
        if (__any(t == 4)) {  // Paeth: byte by byte; |p - a| = |b - c|, |p - b| = |a - c|, |p - c| = |a + b - 2c|
            uint32_t ylo = 0, yhi = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t sh = 8 * (j & 3);
                const uint32_t r4 = j < 4 ? (uint32_t)rw : (uint32_t)(rw >> 32), l4 = j < 4 ? (uint32_t)lw : (uint32_t)(lw >> 32);
                const uint32_t u4 = j < 4 ? (uint32_t)uw : (uint32_t)(uw >> 32), c4 = j < 4 ? (uint32_t)cw : (uint32_t)(cw >> 32);
                const uint32_t raw = (r4 >> sh) & 0xFF, a = (l4 >> sh) & 0xFF, b = (u4 >> sh) & 0xFF, c = (c4 >> sh) & 0xFF;
                const uint32_t pa = __builtin_amdgcn_sad_u8(b, c, 0u), pb = __builtin_amdgcn_sad_u8(a, c, 0u);
                const uint32_t pc = __builtin_amdgcn_sad_u16(a + b, c << 1, 0u);
                const uint32_t bc = pb <= pc ? b : c;
                const uint32_t pr = (pa <= pb && pa <= pc) ? a : bc;
                const uint32_t v = ((raw - pr) & 0xFFu) << sh;
                if (j < 4) ylo |= v;
                else yhi |= v;
            }
            x = t == 4 ? (((uint64_t)yhi << 32) | ylo) : x;
        }
        if (q.start) x = (x & ~(uint64_t)0xFF) | q.t;
        if (__any(wanted && !direct)) {  // a chunk over two rows or inside a row's first pixel: byte by byte (loads: see above)
            if (wanted && !direct) {
                // all the bytes are requested before the first one is used: one trip to memory, not eight
                const uint32_t rb1 = rb + 1;
                uint32_t r = q.off / rb1, cc = q.off - r * rb1;
                uint32_t tt[8], raw[8], pa[8], pb[8], pc[8];
                bool isdata[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t k = cc ? cc - 1 : 0u;
                    const uint8_t* cur = pix + (uint64_t)r * rb;
                    isdata[j] = cc != 0;
                    tt[j] = types[r];
                    const bool l_ok = isdata[j] && k >= bpp, u_ok = isdata[j] && r != 0;
                    raw[j] = isdata[j] ? cur[k] : 0u;
                    pa[j] = l_ok ? cur[k - bpp] : 0u;
                    pb[j] = u_ok ? cur[(int64_t)k - (int64_t)rb] : 0u;
                    pc[j] = (l_ok && u_ok) ? cur[(int64_t)k - (int64_t)rb - (int64_t)bpp] : 0u;
                    if (++cc == rb1) {
                        cc = 0;
                        r++;
                    }
                }
                x = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t v = isdata[j] ? (raw[j] - predict(tt[j], pa[j], pb[j], pc[j])) & 0xFFu : tt[j];
                    x |= (uint64_t)v << (8 * j);
                }
            }
        }
        return x;
    }
};

template <bool PNG>
// Wavefronts per SIMD: five for the plain encoder (96 VGPRs, 12 B of scratch outside the tile loop, 4 KiB rings: 20 wavefronts
// per CU, 2.91 -> 2.71 ms -- the vector ALUs were 86 % busy at four and still had stalls to fill), four for the one that
// filters PNG rows on the way in (114 VGPRs: at 96 it spills 56 B into the loop, 6.45 -> 10.0 ms).
__global__ __launch_bounds__(kEncWaves * kWave) __attribute__((amdgpu_waves_per_eu(PNG ? 4 : 5, PNG ? 4 : 5)))
void deflate_ultrafast_kernel_t(DeflateBatchArgs a) {
    __shared__ EncLds lds;
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = threadIdx.x / kWave;
    for (int i = threadIdx.x; i < 288; i += kEncWaves * kWave) lds.tab[i] = i < 286 ? kUfTables.sym[i] : 0;
    for (int i = lane; i < kEncRingDw; i += kWave) lds.ring[wid][i] = 0;
    __syncthreads();
    {   // the tails of the runs, from the code table just staged
        Encoder e0;
        e0.tab = lds.tab;
        for (int i = threadIdx.x; i < 260; i += kEncWaves * kWave) {
            uint32_t tb = 0, tn = 0;
            if (i < 258) e0.run_tail((uint32_t)i, tb, tn);
            lds.tail[i] = tb | (tn << 24);
        }
    }
    __syncthreads();
    const uint64_t sid = (uint64_t)blockIdx.x * kEncWaves + wid;
    if (sid >= a.n) return;

    const uint8_t* in = a.in + a.in_off[sid];
    uint64_t len = a.in_off[sid + 1] - a.in_off[sid];
    PngSource png{nullptr, nullptr, 0, 0, 0, 0};
    if (PNG) {  // the encoder's input: rows x (1 + row_bytes) filtered bytes
        const uint64_t plen = len, nrows = plen / a.row_bytes;
        const uint64_t tlen = a.types_off[sid + 1] - a.types_off[sid];
        uint32_t st = (nrows * a.row_bytes != plen || tlen != nrows || nrows >= (1ull << 31) / (a.row_bytes + 1ull)) ? 2u : 0u;
        png.pix = in;
        png.types = a.types + a.types_off[sid];
        png.rb = a.row_bytes;
        png.bpp = a.bpp;
        png.magic = (uint32_t)(0x100000000ull / ((uint64_t)a.row_bytes + 1));
        png.rows = st ? 0u : (uint32_t)nrows;
        bool bad_type = false;
        for (uint32_t r = (uint32_t)lane; r < png.rows; r += kWave) bad_type = bad_type || png.types[r] > 4;
        if (__any(bad_type)) st = 1;
        if (lane == 0) a.png_status[sid] = st;
        if (st) {
            if (lane == 0) a.out_len[sid] = 0;
            return;
        }
        len = nrows * (a.row_bytes + 1ull);
    }
    uint8_t* out = a.out + a.out_off[sid];
    const uint64_t cap = a.out_off[sid + 1] - a.out_off[sid];

    Encoder enc;
    enc.ring = lds.ring[wid];
    enc.tab = lds.tab;
    enc.tail = lds.tail;
    enc.lane = lane;
    enc.gmis = (uint32_t)(reinterpret_cast<uintptr_t>(out) & 15);
    enc.out_al = out - enc.gmis;
    enc.cap_bits = ((uint64_t)enc.gmis + cap) * 8;
    enc.qbits = (uint64_t)enc.gmis * 8;
    enc.qflushed = 0;
    enc.overflow = false;

    // ---- header: 53 bytes + 5 bits (ultrafast.rs:81-91) ----
    if (lane < 14) {
        uint32_t w = (uint32_t)kUfHeader[lane * 4] | ((uint32_t)kUfHeader[lane * 4 + 1] << 8) |
                     ((uint32_t)kUfHeader[lane * 4 + 2] << 16) | ((uint32_t)kUfHeader[lane * 4 + 3] << 24);
        if (lane == 13) w &= (1u << (kUfHeaderBits - 13 * 32)) - 1;
        enc.or_bits(enc.qbits + (uint64_t)lane * 32, w);
    }
    enc.qbits += kUfHeaderBits;

    // ---- Adler-32 of the input (ultrafast.rs:95), per-lane partial sums ----
    // A = 1 + sum d_i ; B = len + sum (len - i) d_i  (mod 65521)
    uint64_t acc_a = 0, acc_b = 0;
    // (round 6) per tile only 32-bit sums: the weight of chunk c's bytes is (len - 8 c) = (len - 8 (f0 + lane)) - 512 k for
    // the k-th tile behind the last fold at tile base f0, so  sum (len - 8 c) s_k - u_k  =  (len - 8 (f0 + lane)) S1 -
    // 512 S2 - U  with S1 = sum s_k, S2 = sum k s_k, U = sum u_k -- one 64-bit multiply-add per 256 tiles, not per tile
    uint32_t ad_s1 = 0, ad_s2 = 0, ad_u = 0, ad_k = 0, ad_f0 = 0;
    auto adler_fold = [&]() __attribute__((always_inline)) {
        acc_a += ad_s1;
        acc_b += (uint64_t)((uint32_t)len - 8u * (ad_f0 + (uint32_t)lane)) * ad_s1 - 512ull * ad_s2 - ad_u;
        acc_a %= kAdlerMod;
        acc_b %= kAdlerMod;
        ad_s1 = ad_s2 = ad_u = ad_k = 0;
    };

    // (round 6: chunk numbers are 32-bit -- a buffer of 4 GiB or more cannot be encoded into a slot whose length comes
    // back in 32 bits anyway: out_len = 0xFFFFFFFF, as for a slot that is too small)
    if (len >> 32) {
        if (lane == 0) a.out_len[sid] = 0xFFFFFFFFu;
        return;
    }
    const uint32_t nchunks = (uint32_t)(len / 8);
    uint32_t carry = 0;  // pending run (zero bytes) entering the tile; may exceed 2^32? len < 2^32 assumed below
    // the chunk of the NEXT tile is requested while this one is encoded (the loads are the only
    // global-memory latency of the loop)
    // The load of the next tile's chunk is UNCONDITIONAL (lanes past the end re-read the last chunk
    // and drop it): behind a load in a branch the compiler cannot know how many loads are in flight
    // and waits for all of them -- the prefetch it had just issued included -- at the join, which
    // put a full trip to memory into every tile.
    const uint32_t last_chunk = nchunks ? nchunks - 1 : 0;
    uint64_t x_next = 0;
    PngSource::Req req{0, 0, 0, 0, 0, 0, true, false, false};  // PNG: the chunk of the tile after next, requested
    if (PNG) {
        if (nchunks) x_next = png.finish(png.request((uint64_t)min((uint32_t)lane, last_chunk) * 8), (uint32_t)lane < nchunks);
        if (nchunks) req = png.request((uint64_t)min((uint32_t)lane + kWave, last_chunk) * 8);
    } else if (nchunks) {
        x_next = *reinterpret_cast<const uint64_t*>(in + (uint64_t)min((uint32_t)lane, last_chunk) * 8);  // HW handles misalignment
    }
    if ((uint32_t)lane >= nchunks) x_next = 0;
    for (uint32_t t0 = 0; t0 < nchunks; t0 += kWave) {
        // (the ring's counters are uniform; said so, their 64-bit arithmetic goes to the scalar unit)
        enc.qbits = uni64(enc.qbits);
        enc.qflushed = uni64(enc.qflushed);
        const uint32_t c = t0 + (uint32_t)lane;
        const bool valid = c < nchunks;
        const uint64_t x = x_next;
        if (PNG) {
            x_next = png.finish(req, c + kWave < nchunks);             // tile t + 1: requested a tile ago
            req = png.request((uint64_t)min(c + 2 * kWave, last_chunk) * 8);     // tile t + 2
        } else {
            x_next = *reinterpret_cast<const uint64_t*>(in + (uint64_t)min(c + kWave, last_chunk) * 8);
        }
        if (c + kWave >= nchunks) x_next = 0;
        const uint32_t nvalid = min((uint32_t)kWave, nchunks - t0);
        {   // adler partials: weight of byte j of this chunk is len - (c*8 + j)
            const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
            const uint32_t s = bytesum4(xl) + bytesum4(xh);  // (x = 0 in the lanes past the end: they contribute nothing)
            ad_u = bytedot4(xl, 0x03020100u, ad_u);
            ad_u = bytedot4(xh, 0x07060504u, ad_u);
            ad_s1 += s;
            ad_s2 += ad_k * s;
            ad_k++;
        }
        const bool nz = valid && x != 0;
        const uint32_t tzb = nz ? (uint32_t)__builtin_ctzll(x) >> 3 : 0;
        const uint32_t lzb = nz ? (uint32_t)__builtin_clzll(x) >> 3 : 0;
        const uint64_t nzmask = __ballot(nz);
        const uint64_t below = nzmask & lanemask_lt(lane);
        const int b = below ? 63 - __clzll((long long)below) : -1;
        const uint32_t lz_b = __shfl(lzb, b < 0 ? 0 : b, kWave);
        const uint32_t P = b >= 0 ? lz_b + 8u * (uint32_t)(lane - b - 1) : carry + 8u * (uint32_t)lane;
        const bool pend = nz && P > 0;
        const uint32_t lo = pend ? tzb : 0;  // (and the lzb high zero bytes start the next run)
        // ---- literals of bytes [lo, hi): the codes of all eight bytes are looked up and packed without a
        //      predicate per byte (round 5; rounds 1-4 spent five instructions per byte on `is byte j inside`):
        //      the bytes outside are ZERO bytes, whose code is two zero bits (HUFFMAN_CODES[0] == 0, length 2,
        //      ultrafast.rs:62), so dropping the low ones is a shift of the packed string by 2 x lo bits and
        //      dropping the high ones shortens it by 2 bits each -- they are zeros either way ----
        uint32_t pb[4], pn[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t e0 = lds.tab[(uint32_t)(x >> (16 * k)) & 0xFF];
            const uint32_t e1 = lds.tab[(uint32_t)(x >> (16 * k + 8)) & 0xFF];
            const uint32_t n0 = e0 >> 16;
            pb[k] = (e0 & 0xFFFF) | ((e1 & 0xFFFF) << n0);
            pn[k] = n0 + (e1 >> 16);
        }
        const uint32_t drop_lo = 2 * lo, drop = drop_lo + 2 * lzb;
        uint32_t lit_n = 0;       // bits of the chunk's literals
        uint64_t f0 = 0, f1 = 0;  // the codes of all eight bytes (<= 96 bits); the low drop_lo of them belong to dropped zero bytes
        {
            const uint64_t la = (uint64_t)pb[0] | ((uint64_t)pb[1] << pn[0]);
            const uint64_t lb = (uint64_t)pb[2] | ((uint64_t)pb[3] << pn[2]);
            const uint32_t na = pn[0] + pn[1];  // <= 48, >= 4
            f0 = la | (lb << na);
            f1 = lb >> (64 - na);
            lit_n = nz ? na + pn[2] + pn[3] - drop : 0u;
        }
        // ---- run closed by this chunk (write_run, ultrafast.rs:45-67) ----
        uint32_t nrep = 0, tail_bits = 0, tail_n = 0, run_n = 0;
        const uint32_t e285 = lds.tab[285];
        {   // (all lanes: a table look-up and a handful of selects -- no branch around ~30 instructions)
            const uint32_t run = pend ? P + tzb - 1 : 0u;  // after the leading literal 0
            nrep = run / 258;
            const uint32_t te = lds.tail[run - 258 * nrep];
            tail_bits = pend ? te & 0xFFFFFFu : 0u;
            tail_n = pend ? te >> 24 : 0u;
            run_n = pend ? 2 + nrep * ((e285 >> 16) + 1) + tail_n : 0u;
        }
        const uint32_t lane_bits = run_n + lit_n;
        uint32_t total;
        const uint32_t off = wave_excl_scan_u32(lane_bits, lane, total);
        if ((uint64_t)total + (enc.qbits - enc.qflushed) > kEncTileBudget) {
            enc.flush(false);
        }
        if ((uint64_t)total + (enc.qbits - enc.qflushed) > kEncTileBudget) {
            // ---- slow path: a run too long for the ring; lanes take turns ----
            for (int l = 0; l < (int)nvalid; l++) {
                bool p_l = __shfl((int)pend, l, kWave) != 0;
                uint32_t P_l = __shfl(P, l, kWave), tz_l = __shfl(tzb, l, kWave);
                if (p_l) enc.emit_run_uniform(P_l + tz_l);
                uint32_t left = __shfl(lit_n, l, kWave);
                const uint64_t l0 = drop_lo ? (f0 >> drop_lo) | (f1 << (64 - drop_lo)) : f0, l1 = f1 >> drop_lo;
                const uint32_t w[3] = {(uint32_t)__shfl((int)(uint32_t)l0, l, kWave), (uint32_t)__shfl((int)(uint32_t)(l0 >> 32), l, kWave),
                                       (uint32_t)__shfl((int)(uint32_t)l1, l, kWave)};
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const uint32_t nb = min(left, 32u);
                    enc.emit_uniform(nb < 32 ? w[k] & ((1u << nb) - 1) : w[k], nb);
                    left -= nb;
                }
            }
        } else {
            uint64_t pos = enc.qbits + off;
            // the run's leading literal 0 (two zero bits), its full-length repeats (rare) and its tail ...
            if (pend) {
                pos += 2;
                uint32_t rep_bits = e285 & 0xFFFF;
                const uint32_t rep_n = (e285 >> 16) + 1;  // + the 1-bit distance code, ultrafast.rs:50
                for (uint32_t i = 0; i < nrep; i++) {
                    enc.or_bits(pos, rep_bits);
                    pos += rep_n;
                }
                enc.or_bits(pos, tail_bits);
                pos += tail_n;
            }
            // ... and the chunk's literals.  (round 6) The string of all eight codes goes in as it is, drop_lo bits EARLY:
            // its low drop_lo bits are the codes of the zero bytes the run has swallowed -- zeros, which change nothing
            // where they fall -- so the 96-bit string is shifted once (into its place in the ring), not three times
            // (out of the dropped bits, behind the run's tail, into place).
            enc.or_bits128(pos - drop_lo, f0, f1);
            enc.qbits += total;
        }
        // ---- carry: pending run after this tile ----
        if (nzmask) {
            int last = 63 - __clzll((long long)nzmask);
            carry = __shfl(lzb, last, kWave) + 8u * (nvalid - 1 - (uint32_t)last);
        } else {
            carry += 8u * nvalid;
        }
        if (enc.qbits - enc.qflushed > kEncRingBits / 2) enc.flush(false);
        if (ad_k == 256) {  // (256 tiles x 255 x 2 040 < 2^32: the 32-bit sums cannot overflow)
            adler_fold();
            ad_f0 = t0 + kWave;
        }
    }
    adler_fold();
    // ---- pending run at the end of the chunked part (ultrafast.rs:155-157) ----
    if (carry > 0) enc.emit_run_uniform(carry);
    // ---- remainder bytes as literals (ultrafast.rs:159-164) ----
    {
        const uint32_t rem = (uint32_t)(len & 7);
        uint32_t bits = 0, nb = 0;
        if ((uint32_t)lane < rem) {
            const uint64_t ro = nchunks * 8 + lane;
            uint32_t bv = PNG ? png.byte_at((uint32_t)ro / (png.rb + 1), (uint32_t)ro % (png.rb + 1)) : in[ro];
            uint32_t e = lds.tab[bv];
            bits = e & 0xFFFF;
            nb = e >> 16;
            acc_a += bv;
            acc_b += (uint64_t)(len - (nchunks * 8 + lane)) * bv;
        }
        uint32_t total;
        uint32_t off = wave_excl_scan_u32(nb, lane, total);
        if ((uint64_t)total + (enc.qbits - enc.qflushed) > kEncTileBudget) enc.flush(false);
        enc.or_bits(enc.qbits + off, bits);
        enc.qbits += total;
    }
    // ---- finish: EOB, pad to a byte, Adler-32 big-endian (ultrafast.rs:170-181) ----
    {
        uint32_t e = lds.tab[256];
        enc.emit_uniform(e & 0xFFFF, e >> 16);
        uint32_t pad = (uint32_t)(8 - (enc.qbits & 7)) & 7;
        enc.emit_uniform(0, pad);
        // reduce the per-lane Adler partials
        uint32_t pa = (uint32_t)(acc_a % kAdlerMod), pbm = (uint32_t)(acc_b % kAdlerMod);
        uint32_t A = (1u + wave_sum_u32(pa)) % kAdlerMod;
        uint32_t B = (uint32_t)(((len % kAdlerMod) + wave_sum_u32(pbm)) % kAdlerMod);
        uint32_t adler = (B << 16) | A;
        enc.emit_uniform(__builtin_bswap32(adler), 32);
    }
    enc.flush(true);
    if (lane == 0) {
        uint64_t out_bytes = (enc.qbits >> 3) - enc.gmis;
        a.out_len[sid] = enc.overflow ? 0xFFFFFFFFu : (uint32_t)out_bytes;
    }
}

}  // namespace fdh

extern "C" int fdh_launch_deflate_ultrafast(const uint8_t* in, const uint64_t* in_off, uint8_t* out,
                                            const uint64_t* out_off, uint32_t* out_len, uint64_t n,
                                            hipStream_t stream) {
    fdh::DeflateBatchArgs a{in, in_off, out, out_off, out_len, n, nullptr, nullptr, nullptr, 0, 0};
    if (n == 0) return 0;
    unsigned blocks = (unsigned)((n + fdh::kEncWaves - 1) / fdh::kEncWaves);
    hipLaunchKernelGGL(fdh::deflate_ultrafast_kernel_t<false>, dim3(blocks), dim3(fdh::kEncWaves * fdh::kWave), 0, stream, a);
    return (int)hipGetLastError();
}

// PNG filtering fused into the encoder: pixel rows in, zlib stream of the filtered image out; the
// filtered bytes exist only in registers.
extern "C" int fdh_launch_png_filter_deflate_ultrafast(const uint8_t* pix, const uint64_t* pix_off, const uint8_t* types,
                                                       const uint64_t* types_off, uint8_t* out, const uint64_t* out_off,
                                                       uint32_t* out_len, uint32_t* png_status, uint64_t n,
                                                       uint32_t row_bytes, uint32_t bpp, hipStream_t stream) {
    fdh::DeflateBatchArgs a{pix, pix_off, out, out_off, out_len, n, types, types_off, png_status, row_bytes, bpp};
    if (n == 0) return 0;
    unsigned blocks = (unsigned)((n + fdh::kEncWaves - 1) / fdh::kEncWaves);
    hipLaunchKernelGGL(fdh::deflate_ultrafast_kernel_t<true>, dim3(blocks), dim3(fdh::kEncWaves * fdh::kWave), 0, stream, a);
    return (int)hipGetLastError();
}
