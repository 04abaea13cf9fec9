// inflate_lanes.h -- dense-batch decoder: one ultra-fast-format stream per LANE.
//
// When a batch holds many streams that share the ultra-fast encoder's fixed prefix (reference
// src/compress/ultrafast.rs:82-88) every stream uses the same Huffman tables, and the most
// instruction-efficient mapping is the reference's own loop (src/decompress.rs:645-830: table
// look-up, 1-2 literals per step, dist-1 run fill) run by every lane on its own stream against
// ONE table copy in LDS: no speculation, no cross-lane synchronisation, no output compaction.
// 64 streams per wavefront, 256 per workgroup.
//
// This kernel only ever reports Ok.  Whatever is not the plain happy path -- a stream that is
// not canonical, a token that runs past the input, a full slot, a distance bit that is not the
// single dist-1 code, a checksum mismatch -- is handed to the wave-per-stream kernels
// (PENDING / PENDING_SERIAL), which own the reference's exact error semantics.
#pragma once
#include "inflate_tables.h"

namespace fdh {

constexpr int kLaneBlock = 256;   // streams (threads) per workgroup
constexpr int kLaneInWords = 32;   // per-lane input ring: 32 dwords (128 B)
constexpr int kLaneOutWords = 16;  // per-lane output ring: 16 qwords (128 B)
constexpr int kLaneEvery = 8;      // iterations between global-memory events

// LDS of the lane kernel.  Rings are laid out [word][lane] so that any per-lane word index is
// bank-conflict free (consecutive lanes -> consecutive banks).
struct LaneLds {
    uint32_t lit[kLitSize];
    uint32_t in_ring[kLaneInWords][kLaneBlock];
    uint64_t out_ring[kLaneOutWords][kLaneBlock];
};

struct LaneArgs {
    const uint8_t* in;
    const uint64_t* in_off;
    uint8_t* out;
    const uint64_t* out_off;
    uint32_t* out_len;
    uint32_t* status;
    uint32_t* adler;
    uint64_t n;
    uint32_t flags;
    const uint32_t* canon_lit;   // kLitSize entries (device layout, inflate_tables.h)
    const uint32_t* canon_dist;  // kDistSize entries
    const uint32_t* canon_hdr;   // 14 dwords of prefix (last one masked)
    uint32_t canon_bits;
    uint32_t pending;
};

// 16 bytes from a 16-B aligned address, zero where outside [lo, hi).
__device__ __attribute__((noinline)) uint4 load16_edge(const uint8_t* p, const uint8_t* lo, const uint8_t* hi) {
    uint64_t a = 0, b = 0;
    for (int j = 0; j < 8; j++) {
        if (p + j >= lo && p + j < hi) a |= (uint64_t)p[j] << (8 * j);
        if (p + 8 + j >= lo && p + 8 + j < hi) b |= (uint64_t)p[8 + j] << (8 * j);
    }
    return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
}
__device__ __forceinline__ uint4 load16_guarded(const uint8_t* p, const uint8_t* lo, const uint8_t* hi) {
    if (p >= lo && p + 16 <= hi) return *reinterpret_cast<const uint4*>(p);
    return load16_edge(p, lo, hi);
}

__device__ __forceinline__ void lanes_decode(const LaneArgs& a, LaneLds& L) {
    const int t = threadIdx.x;
    const uint64_t sid = (uint64_t)blockIdx.x * kLaneBlock + t;
    const bool have_stream = sid < a.n;
    // ---- per-lane stream set-up ----
    const uint8_t* in = a.in;
    uint8_t* op = a.out;
    uint32_t in_bits = 0, cap = 0;
    uint32_t state = 2;  // 0 running, 1 finished (end-of-block reached), 2 not ours / gave up
    const uint8_t* buf_hi = a.in;
    if (have_stream) {
        const uint64_t i0 = a.in_off[sid], i1 = a.in_off[sid + 1];
        const uint64_t o0 = a.out_off[sid], o1 = a.out_off[sid + 1];
        in = a.in + i0;
        op = a.out + o0;
        buf_hi = a.in + a.in_off[a.n];
        const uint64_t ilen = i1 - i0, ocap = o1 - o0;
        // the lane decoder wants a 16-B aligned slot and sizes that fit 32-bit counters
        const bool fits = ilen < (1ull << 28) && ocap < (1ull << 31) && (reinterpret_cast<uintptr_t>(op) & 15) == 0;
        in_bits = (uint32_t)(ilen * 8);
        cap = (uint32_t)ocap;
        if (fits && in_bits >= a.canon_bits) state = 0;
    }
    // ---- input ring: 16 B per event, requested two events ahead ----
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(in) & 15);
    const uint8_t* gp = in - mis;      // next 16-B chunk to request
    const uint8_t* gp_end = in + (in_bits >> 3) + 32;
    uint32_t in_wr = 0, in_rd = 0;     // dwords written to / read from the ring
    uint32_t* const my_in = &L.in_ring[0][t];     // word w of this lane: my_in[w * kLaneBlock]
    uint64_t* const my_out = &L.out_ring[0][t];
    auto ring_put = [&](const uint4& v) __attribute__((always_inline)) {
        my_in[((in_wr + 0) & (kLaneInWords - 1)) * kLaneBlock] = v.x;
        my_in[((in_wr + 1) & (kLaneInWords - 1)) * kLaneBlock] = v.y;
        my_in[((in_wr + 2) & (kLaneInWords - 1)) * kLaneBlock] = v.z;
        my_in[((in_wr + 3) & (kLaneInWords - 1)) * kLaneBlock] = v.w;
        in_wr += 4;
    };
    if (state == 0) {  // prime the whole ring (128 B; the prefix alone is 54 B)
        for (int k = 0; k < kLaneInWords / 4; k++) {
            ring_put(load16_guarded(gp, a.in, buf_hi));
            gp += 16;
        }
    }
    // ---- bit reader: (lo, hi) = 64 stream bits, boff = bits of lo already used, nextw = the
    //      following dword (already read from the ring) ----
    uint32_t lo = 0, hi = 0, nextw = 0, boff = 0, cbits = 0;  // cbits: stream bits consumed
    auto ring_get = [&]() __attribute__((always_inline)) -> uint32_t {
        uint32_t w = my_in[(in_rd & (kLaneInWords - 1)) * kLaneBlock];
        in_rd++;
        return w;
    };
    auto skip_bits = [&](uint32_t nbits) __attribute__((always_inline)) {  // nbits <= 32
        boff += nbits;
        cbits += nbits;
        if (boff >= 32) {
            boff -= 32;
            lo = hi;
            hi = nextw;
            nextw = ring_get();
        }
    };
    if (state == 0) {
        for (uint32_t k = 0; k < (mis >> 2); k++) (void)ring_get();  // dwords in front of the stream
        lo = ring_get();
        hi = ring_get();
        nextw = ring_get();
        boff = 8 * (mis & 3);
        // ---- canonical prefix: 14 dwords (the last one masked) ----
        bool same = true;
        for (int k = 0; k < 14; k++) {
            uint32_t nbits = k < 13 ? 32 : a.canon_bits - 13 * 32;
            uint32_t v = __builtin_amdgcn_alignbit(hi, lo, boff);
            if (nbits < 32) v &= (1u << nbits) - 1;
            same = same && v == a.canon_hdr[k];
            skip_bits(nbits);
        }
        if (!same) state = 2;
    }
    // ---- output: 8-byte accumulator -> per-lane ring -> 16-B global stores (+ Adler-32) at events
    uint64_t acc = 0;
    uint32_t acc_n = 0, opos = 0, ostored = 0;  // opos: bytes in ring or stored (multiple of 8)
    uint32_t ad_a = 1, ad_b = 0, blocks = 0;
    uint32_t fill = 0, last = 0;
    uint4 pend_a = make_uint4(0, 0, 0, 0), pend_b = pend_a;  // pend_a: requested two events ago
    bool has_a = false, has_b = false;

    auto drain_out = [&]() __attribute__((always_inline)) {
        while (__any(opos - ostored >= 16)) {
            if (opos - ostored >= 16) {
                const uint32_t w = ostored >> 3;
                const uint64_t x0 = my_out[(w & (kLaneOutWords - 1)) * kLaneBlock];
                const uint64_t x1 = my_out[((w + 1) & (kLaneOutWords - 1)) * kLaneBlock];
                const uint4 q = make_uint4((uint32_t)x0, (uint32_t)(x0 >> 32), (uint32_t)x1, (uint32_t)(x1 >> 32));
                *reinterpret_cast<uint4*>(op + ostored) = q;
                // Adler-32 of the 16 bytes: a += sum, b += 16 a + sum (16 - k) x_k
                const uint32_t s = bytesum4(q.x) + bytesum4(q.y) + bytesum4(q.z) + bytesum4(q.w);
                uint32_t u = bytedot4(q.x, 0x0d0e0f10u, 0);
                u = bytedot4(q.y, 0x090a0b0cu, u);
                u = bytedot4(q.z, 0x05060708u, u);
                u = bytedot4(q.w, 0x01020304u, u);
                ad_b += 16 * ad_a + u;
                ad_a += s;
                if (++blocks == 128) {  // 2 KiB: sums stay far below 2^32
                    ad_a %= kAdlerMod;
                    ad_b %= kAdlerMod;
                    blocks = 0;
                }
                ostored += 16;
            }
        }
    };

    uint32_t iter = 0;
    while (__any(state == 0)) {
        // ---- wave-uniform global-memory event ----
        if ((iter & (kLaneEvery - 1)) == 0) {
            if (has_a) ring_put(pend_a);  // waits for a load issued 2 events ago
            pend_a = pend_b;
            has_a = has_b;
            has_b = false;
            drain_out();
            // free space counts what is already in flight
            if (state == 0 && gp < gp_end && (uint32_t)kLaneInWords - (in_wr - in_rd) >= (has_a ? 8u : 4u)) {
                pend_b = load16_guarded(gp, a.in, buf_hi);
                gp += 16;
                has_b = true;
            }
        }
        iter++;
        // ---- one step of every running lane: a table look-up or 8 bytes of a run ----
        const bool run = state == 0;
        const uint32_t win = __builtin_amdgcn_alignbit(hi, lo, boff);
        const uint32_t e = L.lit[win & (kLitSize - 1)];
        const uint32_t nb = e & 15, kind = (e >> 4) & 15;
        const bool filling = fill != 0;
        const bool is_lit = kind <= K_LIT2, is_len = kind == K_LEN, is_eob = kind == K_EOB;
        // length token: extra bits, then the single distance code ('0' = distance 1)
        const uint32_t ex = (e >> 8) & 31;
        const uint32_t length = (e >> 16) + ((win >> nb) & ((1u << ex) - 1));
        const uint32_t dbit = (win >> (nb + ex)) & 1;
        uint32_t used = nb + (is_len ? ex + 1 : 0);
        uint32_t n = is_lit ? kind + 1 : 0;
        uint64_t v = is_lit ? ((e >> 8) & (kind == K_LIT2 ? 0xFFFFu : 0xFFu)) : 0u;  // bytes only for literals
        bool bad = !(is_lit || is_len || is_eob) || (is_len && (dbit != 0 || opos + acc_n == 0));
        uint32_t new_fill = is_len ? length : 0;
        uint32_t new_last = kind == K_LIT2 ? (e >> 16) & 0xFF : (e >> 8) & 0xFF;
        if (filling) {  // dist-1 run in progress (src/decompress.rs:793-801): no bits consumed
            n = min(fill, 8u);
            new_fill = fill - n;
            v = (uint64_t)last * 0x0101010101010101ull;
            if (n < 8) v &= (1ull << (8 * n)) - 1;
            used = 0;
            bad = false;
            new_last = last;
        }
        if (!is_lit && !filling) new_last = last;
        bad = bad || cbits + used > in_bits      // token runs past the end of the input
                  || opos + acc_n + n > cap      // the exact kernels own OutputTooLarge
                  || in_rd > in_wr;              // starved (the event schedule prevents it; be safe)
        if (run) {
            if (bad) {
                state = 2;
            } else {
                if (is_eob && !filling) state = 1;
                fill = new_fill;
                last = new_last;
                // consume `used` bits
                boff += used;
                cbits += used;
                if (boff >= 32) {
                    boff -= 32;
                    lo = hi;
                    hi = nextw;
                    nextw = ring_get();
                }
                // append n bytes
                const uint32_t tot = acc_n + n;
                acc |= v << (8 * acc_n);
                if (tot >= 8) {
                    my_out[((opos >> 3) & (kLaneOutWords - 1)) * kLaneBlock] = acc;
                    opos += 8;
                    acc = acc_n ? (v >> (8 * (8 - acc_n))) : 0;
                    acc_n = tot - 8;
                } else {
                    acc_n = tot;
                }
            }
        }
    }
    drain_out();
    if (!have_stream) return;
    // ---- tail bytes, trailer, results ----
    uint32_t status = a.pending;  // the wave-per-stream kernels take over (with their own guard)
    if (state == 1) {
        // at most one complete qword and acc_n loose bytes are left
        if (opos - ostored == 8) {
            const uint64_t w = my_out[((ostored >> 3) & (kLaneOutWords - 1)) * kLaneBlock];
            for (uint32_t k = 0; k < 8; k++) {
                uint32_t b = (uint32_t)(w >> (8 * k)) & 0xFF;
                op[ostored + k] = (uint8_t)b;
                ad_a += b;
                ad_b += ad_a;
            }
        }
        for (uint32_t k = 0; k < acc_n; k++) {
            uint32_t b = (uint32_t)(acc >> (8 * k)) & 0xFF;
            op[opos + k] = (uint8_t)b;
            ad_a += b;
            ad_b += ad_a;
        }
        ad_a %= kAdlerMod;
        ad_b %= kAdlerMod;
        const uint32_t adler = (ad_b << 16) | ad_a;
        // src/decompress.rs:306-326: byte boundary, then the big-endian Adler-32
        const uint32_t tb = (cbits + 7) >> 3;
        if ((uint64_t)tb * 8 + 32 <= in_bits) {
            uint32_t stored = ((uint32_t)in[tb] << 24) | ((uint32_t)in[tb + 1] << 16) | ((uint32_t)in[tb + 2] << 8) |
                              (uint32_t)in[tb + 3];
            if (stored == adler || (a.flags & 1u)) {
                status = ST_OK;
                a.out_len[sid] = opos + acc_n;
                if (a.adler) a.adler[sid] = adler;
            }
        }
    }
    a.status[sid] = status;
}

}  // namespace fdh
