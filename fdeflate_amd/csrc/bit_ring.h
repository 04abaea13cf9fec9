// bit_ring.h -- the encoders' output stage: a zeroed LDS ring of bits per wavefront.
//
// Every lane ORs its code bits into the ring at its own bit offset (the offsets come from a prefix
// sum of the per-lane bit counts over the wavefront); complete 16-B lines are stored to the slot
// coalesced and re-zeroed.  "q-space" is the bit position counted from the 16-B aligned address at
// or below the slot, so that the stores are aligned whatever the slot's own alignment.
#pragma once
#include "device_common.h"

namespace fdh {

// RING_DW dwords of LDS per wavefront (a power of two); a wavefront-wide update may add up to
// kTileBudget bits to what has not been flushed yet.
template <int RING_DW>
struct BitRingT {
    static constexpr int kEncRingDw = RING_DW;
    static constexpr uint32_t kEncRingBits = RING_DW * 32;
    static constexpr uint32_t kEncTileBudget = kEncRingBits - 1024;

    uint32_t* ring;
    int lane;
    uint8_t* out_al;   // slot base rounded down to 16 B
    uint32_t gmis;     // slot base & 15
    uint64_t cap_bits; // capacity in q-space bits (gmis*8 + cap*8)
    uint64_t qbits;    // bits emitted so far incl. the gmis*8 offset (uniform)
    uint64_t qflushed; // q-space bit position up to which lines are stored (multiple of 128)
    bool overflow;

    __device__ void or_bits(uint64_t pos, uint32_t bits) {  // bits < 2^32, any lane
        uint32_t sh = (uint32_t)pos & 31;
        uint64_t v = (uint64_t)bits << sh;
        uint32_t d = (uint32_t)(pos >> 5);
        if ((uint32_t)v) atomicOr(&ring[d & (kEncRingDw - 1)], (uint32_t)v);
        if ((uint32_t)(v >> 32)) atomicOr(&ring[(d + 1) & (kEncRingDw - 1)], (uint32_t)(v >> 32));
    }

    // ORs a value of up to 118 bits {v1:v0} into the ring at bit `pos`: the value is shifted into
    // place in registers (five dwords) and only the dwords that hold bits touch the LDS -- a chunk of
    // typical data is one or two atomics instead of one or two per 2-byte piece.
    __device__ void or_bits128(uint64_t pos, uint64_t v0, uint64_t v1) {
        const uint32_t s = (uint32_t)pos & 31;
        const uint32_t d = (uint32_t)(pos >> 5);
        const uint32_t w0 = (uint32_t)v0, w1 = (uint32_t)(v0 >> 32), w2 = (uint32_t)v1, w3 = (uint32_t)(v1 >> 32);
        // out[k] = (w[k] << s) | (w[k-1] >> (32 - s)); v_alignbit takes the shift modulo 32, so s = 0 is selected apart
        const uint32_t r = 32 - s;
        const uint32_t o0 = w0 << s;
        const uint32_t o1 = s ? __builtin_amdgcn_alignbit(w1, w0, r) : w1;
        const uint32_t o2 = s ? __builtin_amdgcn_alignbit(w2, w1, r) : w2;
        const uint32_t o3 = s ? __builtin_amdgcn_alignbit(w3, w2, r) : w3;
        const uint32_t o4 = s ? (w3 >> r) : 0u;
        if (o0) atomicOr(&ring[d & (kEncRingDw - 1)], o0);
        if (o1) atomicOr(&ring[(d + 1) & (kEncRingDw - 1)], o1);
        if (o2) atomicOr(&ring[(d + 2) & (kEncRingDw - 1)], o2);
        if (o3) atomicOr(&ring[(d + 3) & (kEncRingDw - 1)], o3);
        if (o4) atomicOr(&ring[(d + 4) & (kEncRingDw - 1)], o4);
    }

    // Store every complete 16-B line (all lines when final) and re-zero it in the ring.
    __device__ void flush(bool final) {
        wave_sync();
        uint64_t end_line = final ? (qbits + 127) >> 7 : qbits >> 7;
        uint64_t line0 = qflushed >> 7;
        uint64_t end_byte = (qbits + 7) >> 3;  // q-space byte just past the data
        for (uint64_t ln = line0 + lane; ln < end_line; ln += kWave) {
            uint32_t di = (uint32_t)(ln * 4) & (kEncRingDw - 1);
            uint4 v = *reinterpret_cast<uint4*>(&ring[di]);
            *reinterpret_cast<uint4*>(&ring[di]) = make_uint4(0, 0, 0, 0);
            uint64_t b0 = ln * 16;
            uint64_t lo = (b0 < gmis) ? gmis - b0 : 0;
            uint64_t hi = (b0 + 16 > end_byte) ? end_byte - b0 : 16;
            if (overflow || b0 + hi > (cap_bits >> 3)) {
                overflow = true;  // slot too small: never write past it
            } else if (lo == 0 && hi == 16) {
                *reinterpret_cast<uint4*>(out_al + b0) = v;
            } else {
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    if ((uint64_t)j >= lo && (uint64_t)j < hi) out_al[b0 + j] = (uint8_t)(w[j >> 2] >> ((j & 3) * 8));
                }
            }
        }
        overflow = __any(overflow);
        if (end_line > line0) qflushed = end_line << 7;
        wave_sync();
    }

    // uniform emission of up to 32 bits by lane 0 (header, EOB, trailer, slow paths)
    __device__ void emit_uniform(uint32_t bits, uint32_t nbits) {
        if (qbits + nbits - qflushed > kEncTileBudget) flush(false);
        if (lane == 0) or_bits(qbits, bits);
        qbits += nbits;
    }
};

// Exclusive prefix sum over the wavefront with DPP moves only (row shifts inside each row of 16
// lanes, then the row broadcasts): no LDS traffic, unlike ds_bpermute-based shuffles.  All lanes active.
__device__ __forceinline__ uint32_t wave_excl_scan_u32(uint32_t v, int lane, uint32_t& total) {
    uint32_t x = v;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
    total = (uint32_t)__builtin_amdgcn_readlane((int)x, kWave - 1);
    (void)lane;
    return x - v;
}

}  // namespace fdh
