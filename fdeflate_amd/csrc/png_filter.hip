// png_filter.hip -- PNG scanline reconstruction ("unfilter", the step the png crate runs on the
// decoder's output) and filtering (the step in front of the ultra-fast encoder), batched.
//
// Not part of the fdeflate crate itself (its reverse dependency image-rs/image-png does it,
// reference README.md:11); the algorithm is the PNG specification's (W3C / ISO/IEC 15948 section
// 9.2 "Filter types for filter method 0", 9.4 Paeth): types 0 None, 1 Sub, 2 Up, 3 Average,
// 4 Paeth over bytes, `bpp` bytes per pixel, arithmetic modulo 256, zero to the left of the first
// pixel and above the first row.
//
// Reconstruction is serial along a row for Sub / Average / Paeth (byte x needs byte x - bpp of the
// SAME row) and serial down the rows for Up / Average / Paeth, and neighbouring rows may use
// different types: a pixel needs its left, upper and upper-left neighbours, which leaves the
// anti-diagonals of an image as the independent work.  png_wave_kernel (the default) gives an image
// to a wavefront and runs the rows of a 64-row band on the 64 lanes, skewed by one 16-byte chunk per
// row; png_filter_kernel (FDH_PNG_LANE_PER_IMAGE=1) gives an image to a lane.  Both walk rows 16
// bytes at a time in registers with static indexing, the predictor of every type computed
// branch-free and selected by per-row byte masks, so lanes whose rows have different types do not
// diverge.  Filtering has no such dependence (the predictors use raw neighbours).
#include "device_common.h"

#include <hip/hip_cooperative_groups.h>

#include <algorithm>
#include <cstdlib>

namespace fdh {

struct PngArgs {
    const uint8_t* src;       // unfilter: filtered rows (type byte + row_bytes each); filter: pixels
    const uint64_t* src_off;  // n + 1
    uint8_t* dst;             // unfilter: pixels; filter: filtered rows
    const uint64_t* dst_off;  // n + 1
    const uint8_t* types;     // filter only: one type per row, all images back to back
    const uint64_t* types_off;  // filter only: n + 1
    uint32_t* status;         // 0 ok, 1 filter type > 4, 2 sizes do not fit, 3 skipped (gate)
    const uint32_t* gate;     // nullable: image i is processed only if gate[i] == 0 (the decoder's status)
    const uint32_t* gate_len; // nullable (with gate): ... and gate_len[i] (the decoded length) fills the source slot
                              // exactly: a stream that ends early leaves stale bytes behind it -> status 2
    uint64_t n;
    uint32_t row_bytes;
};

// Paeth predictor (PNG specification 9.4): p = a + b - c; the neighbour closest to p, ties in the
// order a, b, c.  |p - a| = |b - c|, |p - b| = |a - c|, |p - c| = |a + b - 2c|: three
// sum-of-absolute-differences instructions on byte values.
__device__ __forceinline__ uint32_t png_paeth(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t pa = __builtin_amdgcn_sad_u8(b, c, 0u), pb = __builtin_amdgcn_sad_u8(a, c, 0u);
    const uint32_t pc = __builtin_amdgcn_sad_u16(a + b, c << 1, 0u);
    const uint32_t bc = pb <= pc ? b : c;
    return (pa <= pb && pa <= pc) ? a : bc;
}

// The predictor of a row's type without branching on the type (lanes hold rows of different types):
// None / Sub / Up / Average are (a * wa + b * wb) >> sh with per-row weights, Paeth is selected over it.
struct PngMasks {
    uint32_t wa, wb, sh;
    uint32_t paeth;  // all ones for a Paeth row: the select is a bit-field insert, not a branch per byte
    __device__ explicit PngMasks(uint32_t t) : wa(t == 1 || t == 3 ? 1u : 0u), wb(t == 2 || t == 3 ? 1u : 0u), sh(t == 3 ? 1u : 0u), paeth(t == 4 ? 0xFFFFFFFFu : 0u) {}
    __device__ __forceinline__ uint32_t pred(uint32_t a, uint32_t b, uint32_t c) const {
        const uint32_t lin = (__umul24(a, wa) + __umul24(b, wb)) >> sh;
        return (png_paeth(a, b, c) & paeth) | (lin & ~paeth);
    }
};

__device__ __forceinline__ uint4 png_load16(const uint8_t* p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);  // rows start at any alignment: unaligned 16-B access (hardware-supported)
    return v;
}
__device__ __forceinline__ void png_store16(uint8_t* p, const uint4& v) { __builtin_memcpy(p, &v, 16); }
__device__ __forceinline__ uint32_t png_byte(const uint4& v, int k) {
    const uint32_t w = k < 4 ? v.x : (k < 8 ? v.y : (k < 12 ? v.z : v.w));
    return (w >> (8 * (k & 3))) & 0xFF;
}

// Sixteen bytes of one row.  UNFILTER: out = filt + pred(reconstructed left, up, up-left); else
// filt = raw - pred(raw left, up, up-left).  la / ua carry the last BPP bytes of this row
// (reconstructed / raw) and of the row above into the next chunk.
template <int BPP, bool UNFILTER>
__device__ __forceinline__ uint4 png_chunk(const uint4& f, const uint4& u, uint32_t (&la)[8], uint32_t (&ua)[8], const PngMasks& m) {
    uint32_t o[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t b = png_byte(u, k);
        // left / up-left neighbour: BPP bytes back, in this chunk or in the carried tail
        const uint32_t a = k >= BPP ? (UNFILTER ? o[k - BPP] : png_byte(f, k - BPP)) : la[8 - BPP + k];
        const uint32_t c = k >= BPP ? png_byte(u, k - BPP) : ua[8 - BPP + k];
        const uint32_t fv = png_byte(f, k);
        const uint32_t pr = m.pred(a, b, c);
        o[k] = (UNFILTER ? fv + pr : fv - pr) & 0xFF;
    }
#pragma unroll
    for (int k = 0; k < BPP; k++) {  // carry the tails (BPP <= 8 <= 16)
        la[8 - BPP + k] = UNFILTER ? o[16 - BPP + k] : png_byte(f, 16 - BPP + k);
        ua[8 - BPP + k] = png_byte(u, 16 - BPP + k);
    }
    uint4 r;
    r.x = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
    r.y = o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24);
    r.z = o[8] | (o[9] << 8) | (o[10] << 16) | (o[11] << 24);
    r.w = o[12] | (o[13] << 8) | (o[14] << 16) | (o[15] << 24);
    return r;
}

// One row of one image, walked by one lane.
template <int BPP, bool UNFILTER>
__device__ __forceinline__ void png_row(const uint8_t* in, const uint8_t* up, uint8_t* out, uint32_t row_bytes, uint32_t type) {
    const PngMasks m(type);
    uint32_t la[8], ua[8];
#pragma unroll
    for (int k = 0; k < 8; k++) la[k] = ua[k] = 0;
    uint32_t x = 0;
    for (; x + 16 <= row_bytes; x += 16) {
        const uint4 f = png_load16(in + x);
        uint4 u = make_uint4(0, 0, 0, 0);
        if (up) u = png_load16(up + x);
        png_store16(out + x, png_chunk<BPP, UNFILTER>(f, u, la, ua, m));
    }
    // the last < 16 bytes, byte by byte, the tails rotating through la / ua
    for (; x < row_bytes; x++) {
        const uint32_t b = up ? up[x] : 0u;
        const uint32_t a = la[8 - BPP], c = ua[8 - BPP];
        const uint32_t fv = in[x];
        const uint32_t pr = m.pred(a, b, c);
        const uint32_t ov = (UNFILTER ? fv + pr : fv - pr) & 0xFF;
        out[x] = (uint8_t)ov;
#pragma unroll
        for (int k = 8 - BPP; k < 7; k++) {
            la[k] = la[k + 1];
            ua[k] = ua[k + 1];
        }
        la[7] = UNFILTER ? ov : fv;
        ua[7] = b;
    }
}

// 0: process image i; 3: its stream did not decode; 2: it decoded to fewer bytes than the slot holds
__device__ __forceinline__ uint32_t png_gate(const PngArgs& a, uint64_t i) {
    if (!a.gate) return 0;
    if (a.gate[i] != 0) return 3;
    if (a.gate_len && (uint64_t)a.gate_len[i] != a.src_off[i + 1] - a.src_off[i]) return 2;
    return 0;
}

template <int BPP, bool UNFILTER>
__global__ __launch_bounds__(kWave) void png_filter_kernel(PngArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * kWave + threadIdx.x;
    if (i >= a.n) return;
    if (const uint32_t g = png_gate(a, i)) {
        a.status[i] = g;
        return;
    }
    const uint64_t s0 = a.src_off[i], s1 = a.src_off[i + 1], d0 = a.dst_off[i], d1 = a.dst_off[i + 1];
    const uint64_t rb = a.row_bytes;
    const uint64_t src_row = UNFILTER ? rb + 1 : rb, dst_row = UNFILTER ? rb : rb + 1;
    const uint64_t rows = (s1 - s0) / src_row;
    uint32_t st = 0;
    if (rows * src_row != s1 - s0 || rows * dst_row > d1 - d0) st = 2;
    if (!UNFILTER && st == 0 && a.types_off[i + 1] - a.types_off[i] < rows) st = 2;
    for (uint64_t r = 0; r < rows && st == 0; r++) {
        const uint8_t* in = a.src + s0 + r * src_row;
        uint8_t* out = a.dst + d0 + r * dst_row;
        if (UNFILTER) {
            const uint32_t t = in[0];
            if (t > 4) {
                st = 1;
                break;
            }
            png_row<BPP, true>(in + 1, r ? out - rb : nullptr, out, a.row_bytes, t);
        } else {
            const uint32_t t = a.types[a.types_off[i] + r];
            if (t > 4) {
                st = 1;
                break;
            }
            out[0] = (uint8_t)t;
            png_row<BPP, false>(in, r ? in - rb : nullptr, out + 1, a.row_bytes, t);
        }
    }
    a.status[i] = st;
}

// `valid` bytes (1..16) at p as a 16-byte chunk, never reading behind them.
// The first `valid` (< 16: else all) of sixteen bytes, the others untouched / zero: an 8-, a 4-, a 2- and a 1-byte access as
// the bits of `valid` say, not a loop over the bytes -- ONE lane of the pipeline has a row's last, partial piece in hand at
// every memory step, and a wavefront issues what one of its lanes executes (the byte loop: ~150 instructions per piece).
__device__ __forceinline__ uint4 png_load_part(const uint8_t* p, uint32_t valid) {
    if (valid >= 16) return png_load16(p);
    uint64_t lo = 0, hi = 0, part = 0;
    uint32_t at = 0, sh = 0;  // bytes read so far of this half; bits filled of `part`
    if (valid & 8) {
        __builtin_memcpy(&lo, p, 8);
        at = 8;
    }
    if (valid & 4) {
        uint32_t w;
        __builtin_memcpy(&w, p + at, 4);
        part = w;
        at += 4;
        sh = 32;
    }
    if (valid & 2) {
        uint16_t h;
        __builtin_memcpy(&h, p + at, 2);
        part |= (uint64_t)h << sh;
        at += 2;
        sh += 16;
    }
    if (valid & 1) part |= (uint64_t)p[at] << sh;
    if (valid & 8) hi = part;
    else lo = part;
    return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
__device__ __forceinline__ void png_store_part(uint8_t* p, const uint4& v, uint32_t valid) {
    if (valid >= 16) {
        png_store16(p, v);
        return;
    }
    const uint64_t lo = ((uint64_t)v.y << 32) | v.x, hi = ((uint64_t)v.w << 32) | v.z;
    uint64_t rest = lo;
    uint32_t at = 0;
    if (valid & 8) {
        __builtin_memcpy(p, &lo, 8);
        rest = hi;
        at = 8;
    }
    if (valid & 4) {
        const uint32_t w = (uint32_t)rest;
        __builtin_memcpy(p + at, &w, 4);
        rest >>= 32;
        at += 4;
    }
    if (valid & 2) {
        const uint16_t h = (uint16_t)rest;
        __builtin_memcpy(p + at, &h, 2);
        rest >>= 16;
        at += 2;
    }
    if (valid & 1) p[at] = (uint8_t)rest;
}
__device__ __forceinline__ uint32_t png_from_lane_below(uint32_t x) {  // lane j gets lane j - 1's value (lane 0: 0)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}

// One image per WAVEFRONT: lane j takes row j of a band of 64 rows.  Reconstruction needs the row
// above, so the lanes run skewed by one 16-byte chunk: at step t lane j works on chunk t - j, and
// the chunk above it is what lane j - 1 produced one step earlier -- it arrives through a DPP lane
// shift, not through memory (row 0 of a later band reads the last row of the band before from the
// output).  Filtering has no such dependence (the predictors use raw neighbours): no skew, the row
// above is the chunk lane j - 1 read in the same step.  Same per-chunk arithmetic as the
// lane-per-image kernel.
//
// Memory: with 16 bytes per lane per step and the rows 1 KiB or more apart, a wavefront touches 64
// different cache lines per access and comes back to each of them eight times; with sixteen
// wavefronts per CU the lines do not survive in the caches in between, and every 16 bytes cost a
// whole line of traffic (measured: the kernel ran at the same speed with the arithmetic removed,
// and with it doubled).  So every lane moves whole 128-byte lines: eight back-to-back loads fill a
// register set one line ahead of the row's position, the set is parked in an LDS line buffer when
// the row gets there, and the output goes through a second line buffer and leaves as eight
// back-to-back stores when a line is complete.
struct PngWaveLds {
    uint4 lin[kWave][8];   // the 128-byte line of the row each lane is reading
    uint4 lout[kWave][8];  // the line each lane is producing
    uint4 lup[8];          // lane 0 of a later band: the line of the row above
};

template <int BPP, bool UNFILTER>
__global__ __launch_bounds__(kWave) void png_wave_kernel(PngArgs a) {
    __shared__ PngWaveLds lds;
    const uint64_t i = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    if (const uint32_t g = png_gate(a, i)) {
        if (lane == 0) a.status[i] = g;
        return;
    }
    const uint64_t s0 = a.src_off[i], s1 = a.src_off[i + 1], d0 = a.dst_off[i], d1 = a.dst_off[i + 1];
    const uint64_t rb = a.row_bytes;
    const uint64_t src_row = UNFILTER ? rb + 1 : rb, dst_row = UNFILTER ? rb : rb + 1;
    const uint64_t rows = (s1 - s0) / src_row;
    uint32_t st = 0;
    if (rows * src_row != s1 - s0 || rows * dst_row > d1 - d0) st = 2;
    if (!UNFILTER && st == 0 && a.types_off[i + 1] - a.types_off[i] < rows) st = 2;
    const uint32_t nchunks = (uint32_t)((rb + 15) / 16);
    const uint8_t* const src_end = a.src + s1;
    const uint8_t* const dst_end = a.dst + d0 + rows * dst_row;
    // the eight 16-byte pieces of line `ln` of the row at `row`; reads stay below `end`
    auto fetch = [&](const uint8_t* row, const uint8_t* end, uint32_t ln, uint4(&regs)[8]) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint64_t o = (uint64_t)ln * 128 + (uint64_t)k * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (o < rb) v = row + o + 16 <= end ? png_load16(row + o) : png_load_part(row + o, (uint32_t)min((uint64_t)16, rb - o));
            regs[k] = v;
        }
    };
    for (uint64_t r0 = 0; r0 < rows && st == 0; r0 += kWave) {
        const uint64_t r = r0 + lane;
        const bool have = r < rows;
        const uint8_t* in = a.src + s0 + r * src_row + (UNFILTER ? 1 : 0);
        uint8_t* out = a.dst + d0 + r * dst_row + (UNFILTER ? 0 : 1);
        uint32_t t = 0;
        if (have) t = UNFILTER ? in[-1] : a.types[a.types_off[i] + r];
        // rows from the first bad filter type on are not produced (the serial walk stops there)
        const uint64_t bad = __ballot(have && t > 4);
        uint32_t band = (uint32_t)min((uint64_t)kWave, rows - r0);
        if (bad) {
            band = (uint32_t)__builtin_ctzll(bad);
            st = 1;
        }
        const bool mine = lane < band;
        if (!UNFILTER && mine) out[-1] = (uint8_t)t;
        const PngMasks m(t);
        uint32_t la[8], ua[8];
#pragma unroll
        for (int k = 0; k < 8; k++) la[k] = ua[k] = 0;
        // the row above lane 0's row, from memory: the output of the band before / the raw input
        const bool from_memory = lane == 0 && r > 0;
        const uint8_t* const uprow = (UNFILTER ? out : in) - rb;
        const uint8_t* const upend = UNFILTER ? dst_end : src_end;
        uint4 pf[8];  // the line behind the one in lds.lin
        if (mine) {
            fetch(in, src_end, 0, pf);
#pragma unroll
            for (int k = 0; k < 8; k++) lds.lin[lane][k] = pf[k];
            fetch(in, src_end, 1, pf);
        }
        uint4 last = make_uint4(0, 0, 0, 0);  // what the lane below needs: this lane's output / input chunk
        const uint32_t steps = band ? nchunks + (UNFILTER ? band - 1 : 0) : 0;
        for (uint32_t step = 0; step < steps; step++) {
            const uint32_t c = step - (UNFILTER ? lane : 0);  // (wraps for the lanes that have not started)
            const bool on = mine && c < nchunks;
            if (on && (c & 7) == 0) {  // entering a line
                if (c) {
#pragma unroll
                    for (int k = 0; k < 8; k++) lds.lin[lane][k] = pf[k];
                    fetch(in, src_end, (c >> 3) + 1, pf);
                }
                if (from_memory) {
                    uint4 up[8];
                    fetch(uprow, upend, c >> 3, up);
#pragma unroll
                    for (int k = 0; k < 8; k++) lds.lup[k] = up[k];
                }
            }
            uint4 f = make_uint4(0, 0, 0, 0);
            if (on) f = lds.lin[lane][c & 7];
            if (!UNFILTER) last = f;
            uint4 u = make_uint4(png_from_lane_below(last.x), png_from_lane_below(last.y), png_from_lane_below(last.z), png_from_lane_below(last.w));
            if (on) {
                if (lane == 0) u = r > 0 ? lds.lup[c & 7] : make_uint4(0, 0, 0, 0);
                const uint4 o = png_chunk<BPP, UNFILTER>(f, u, la, ua, m);
                if (UNFILTER) last = o;
                lds.lout[lane][c & 7] = o;
                if ((c & 7) == 7 || c == nchunks - 1) {  // the line is complete: out it goes
                    const uint64_t base = (uint64_t)(c >> 3) * 128;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const uint64_t o16 = base + (uint64_t)k * 16;
                        if ((uint32_t)k <= (c & 7) && o16 < rb) png_store_part(out + o16, lds.lout[lane][k], (uint32_t)min((uint64_t)16, rb - o16));
                    }
                }
            }
        }
        // the next band's first row reads this band's last row back
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) a.status[i] = st;
}

// ---- reconstruction as one pipeline over many images ------------------------------------------
//
// png_wave_kernel pays for its skew twice on small images: a band of 64 rows of N chunks takes
// N + 63 steps (half the lane-steps idle for the bench's 64-chunk rows), and because every lane
// crosses its 128-byte line boundaries at a different step, every step issues the line loads and
// stores for an eighth of the lanes (the CU's one address unit ends up the busiest unit).  Here
// the rows of `per_wave` consecutive images form ONE sequence R = 0, 1, 2, ...: row R belongs to
// lane R % 64 and starts at step (R % 64) + (R / 64) * P, P = max(N rounded up to 8, 64), so a lane
// goes from one of its rows to the next without waiting and the ramp is paid once per wavefront.
// The row above is still one lane below and one step ahead (lane 63 hands its chunks to lane 0
// through an LDS row buffer, P - 63 steps ahead); the first row of an image has no row above.
// Memory moves at wavefront-uniform steps: every eighth step each lane parks the line it fetched
// eight steps ago in its double-buffered LDS line, fetches the line it will enter next, and writes
// out the line it completed last -- eight full-width loads and stores per eight steps.
constexpr uint32_t kPipeMaxChunks = 256;  // rows up to 4 KiB (the LDS row buffer)
constexpr uint32_t kPipeMaxImages = 8;
#ifndef FDH_PNG_IN_SLOT
#define FDH_PNG_IN_SLOT 4
#endif
// 16-byte chunks per line slot of the INPUT: 64 bytes (four loads in a row to one or two lines), two buffers, 8 KiB of LDS.  With
// the stores transposed this pays (4.99 -> 4.45 ms at 8 wavefronts per CU instead of 10); before, larger input slots only lost
// occupancy, and 128 bytes still do (6.9 ms at 5 per CU).
constexpr int kSlot = FDH_PNG_IN_SLOT;
#ifndef FDH_PNG_OUT_SLOT
#define FDH_PNG_OUT_SLOT 8
#endif
// ... and of the output: 128 bytes, stored TRANSPOSED (the step loop: eight lanes to a row).  The kernel's time follows its
// occupancy and its memory instructions -- compiled without its stores it took 3.0 instead of 6.6 ms, without its loads 3.7,
// without both 2.6 --: a lane storing its own row's bytes makes 64 pieces of 16 bytes in 64 lines per instruction.  A lane
// storing 64 bytes at once (four stores in a row): 6.56 -> 6.16 ms; non-temporal stores: 14 ms; the eight rows whose 128-byte
// slot is complete at a step stored by eight lanes each, from ONE buffer per lane: 6.5 -> 5.35 ms.
constexpr int kOut = FDH_PNG_OUT_SLOT;

struct PngPipeLds {
    uint4 lin[kWave][2][kSlot];
    uint4 lout[kWave][kOut];   // (one buffer: a slot is stored at the step it is complete, before its owner's next chunk)
    uint4 rdesc[kWave][2];     // per lane, row count & 1: where the row's output starts (x, y), whether it is produced (z)
    uint64_t sbase[kPipeMaxImages + 1], dbase[kPipeMaxImages];  // offsets of the images' buffers (sbase[j + 1]: end of j's)
    uint32_t rowsum[kPipeMaxImages + 1];
    uint32_t bad[kPipeMaxImages];  // first row with a bad filter type, per image
};

struct PngPipeRow {  // one row of the sequence, as a lane sees it
    const uint8_t* in;  // first filtered byte (behind the type byte)
    uint8_t* out;
    uint32_t r, img;    // row in its image; image slot in the wavefront
    uint32_t type;
    bool valid;         // the sequence has this row
    bool produced;      // ... and it is reconstructed (no bad filter type at or before it)
};

template <int BPP>
__global__ __launch_bounds__(kWave) void png_pipe_kernel(PngArgs a, uint32_t per_wave) {
    __shared__ PngPipeLds lds;
    extern __shared__ uint4 png_rowbuf[];  // lane 63 -> lane 0: one row of chunks (sized by the launch)
    const uint32_t lane = threadIdx.x;
    const uint64_t img0 = (uint64_t)blockIdx.x * per_wave;
    const uint32_t cnt = (uint32_t)min((uint64_t)per_wave, a.n - img0);
    const uint64_t rb = a.row_bytes;
    const uint32_t N = (uint32_t)((rb + 15) / 16);
    const uint32_t P = max((N + 7) & ~7u, (uint32_t)kWave);
    // ---- the images of this wavefront: row counts, statuses ----
    uint32_t my_rows = 0, my_st = 0;
    if (lane < cnt) {
        const uint64_t i = img0 + lane;
        if (const uint32_t g = png_gate(a, i)) {
            my_st = g;
        } else {
            const uint64_t fl = a.src_off[i + 1] - a.src_off[i], pl = a.dst_off[i + 1] - a.dst_off[i];
            const uint64_t rows = fl / (rb + 1);
            if (rows * (rb + 1) != fl || rows * rb > pl) my_st = 2;
            else my_rows = (uint32_t)rows;
        }
        lds.bad[lane] = 0xFFFFFFFFu;
        lds.sbase[lane] = a.src_off[i];
        lds.dbase[lane] = a.dst_off[i];
        if (lane + 1 == cnt) lds.sbase[cnt] = a.src_off[i + 1];
    }
    {   // prefix sums of the row counts (at most eight images: by hand)
        uint32_t sum = 0;
        for (uint32_t j = 0; j < cnt; j++) {
            if (lane == 0) lds.rowsum[j] = sum;
            sum += (uint32_t)__builtin_amdgcn_readlane((int)my_rows, (int)j);
        }
        if (lane == 0) lds.rowsum[cnt] = sum;
    }
    wave_sync();
    const uint32_t T = lds.rowsum[cnt];
    // A lane's rows come in increasing order: the image is searched on from the one of its last row.
    // (One lane starts a row at every step, so this runs at every step: kept short, and without a
    // global load on the way to the type byte's address.)
    auto describe = [&](uint32_t R, uint32_t j) -> PngPipeRow {
        PngPipeRow d{nullptr, nullptr, 0, 0, 0, false, false};
        if (R >= T) return d;
        while (lds.rowsum[j + 1] <= R) j++;  // (R < T = rowsum[cnt] ends it)
        d.r = R - lds.rowsum[j];
        d.img = j;
        d.in = a.src + lds.sbase[j] + (uint64_t)d.r * (rb + 1) + 1;
        d.out = a.dst + lds.dbase[j] + (uint64_t)d.r * rb;
        d.type = d.in[-1];
        d.valid = true;
        return d;
    };
    // reads of a row's lines stay inside its image's filtered bytes (+ the next row's, harmless)
    auto fetch = [&](const PngPipeRow& d, uint32_t ln, uint4(&regs)[kSlot]) {
        const uint8_t* const end = a.src + lds.sbase[d.img + 1];
        // A line inside the IMAGE's filtered bytes: plain loads.  (Until round 6: a line inside the ROW -- a row's last,
        // partial line went the other way, and as one lane has such a line at every memory step, every step's loads were
        // followed by that path's `s_waitcnt vmcnt(0)`: its zeros go to the registers the loads are in flight to.  What a
        // plain load reads behind the row's end are the next row's bytes: they stay in positions that are never stored.)
        if (d.valid && d.in + (uint64_t)(ln + 1) * (16 * kSlot) <= end) {
#pragma unroll
            for (int k = 0; k < kSlot; k++) regs[k] = png_load16(d.in + (uint64_t)ln * (16 * kSlot) + (uint64_t)k * 16);
            return;
        }
#pragma unroll
        for (int k = 0; k < kSlot; k++) {
            const uint64_t o = (uint64_t)ln * (16 * kSlot) + (uint64_t)k * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (d.valid && o < rb) v = d.in + o + 16 <= end ? png_load16(d.in + o) : png_load_part(d.in + o, (uint32_t)min((uint64_t)16, rb - o));
            regs[k] = v;
        }
    };
    // A lane starts its row k at step 2 * kSlot + k * P + lane and needs the description of row k + 1 there (its lines are
    // fetched ahead).  Until round 6 it was made right there -- an LDS search, 64-bit address arithmetic and the load of
    // the row's type byte, waited for on the spot -- and as ONE lane starts a row at every step, every lane of the
    // wavefront went through those ~90 instructions and that memory round trip at every step.  Now all lanes describe
    // their row k + 2 together, once per P steps (when lane 0 starts its row k), and hand the description down a
    // two-deep queue: a period of at least 64 steps lies between the load of a type byte and its use.
    PngPipeRow prev{nullptr, nullptr, 0, 0, 0, false, false}, cur = prev, next = describe(lane, 0);
    PngPipeRow nn = prev, nn2 = describe(kWave + lane, next.img);
    uint32_t cu = P - 2 * kSlot, ku = 0;  // lane 0's position in its period; the row period it is in
    // position in the period of row k (signed: negative long before the lane's first row); it reaches
    // P (= 0 of the next row) after lane + 16 steps -- two uniform steps ahead of the first chunk, one
    // to fetch its line and one to park it
    int32_t c = (int32_t)P - (int32_t)lane - 2 * kSlot;
    uint32_t k = 0xFFFFFFFFu;  // rows of this lane started so far, minus one
    uint32_t la[8], ua[8];
#pragma unroll
    for (int q = 0; q < 8; q++) la[q] = ua[q] = 0;
    PngMasks m(0);
    uint4 pf[kSlot], last = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < kSlot; q++) pf[q] = make_uint4(0, 0, 0, 0);
    const uint32_t kmax = T ? (T - 1) / kWave : 0;
    // the last slot of the last row ends before step kmax * P + 63 + 16 + (N rounded up to 8); the
    // uniform step behind that writes its line
    const uint32_t steps = T ? ((kmax * P + kWave + 2 * kSlot + ((N + 7) & ~7u) + 7) & ~7u) + 1 : 0;
    for (uint32_t step = 0; step < steps; step++) {
        if (cu == P) {  // lane 0 starts its row ku: every lane's row ku + 1 is due from now on, row ku + 2 is looked up
            cu = 0;
            nn = nn2;
            nn2 = describe((ku + 2) * kWave + lane, nn.img);
            ku++;
        }
        cu++;
        if (c == (int32_t)P) {  // this lane starts its next row
            c = 0;
            k++;
            prev = cur;
            cur = next;
            next = nn;
            if (cur.valid) {
                if (cur.type > 4) atomicMin(&lds.bad[cur.img], cur.r);
                cur.produced = cur.type <= 4 && cur.r < lds.bad[cur.img];
            }
            m = PngMasks(cur.type);
#pragma unroll
            for (int q = 0; q < 8; q++) la[q] = ua[q] = 0;
            const uint64_t op = reinterpret_cast<uint64_t>(cur.out);
            lds.rdesc[lane][k & 1] = make_uint4((uint32_t)op, (uint32_t)(op >> 32), (cur.valid && cur.produced) ? 1u : 0u, 0u);
        }
        {   // ---- write-out, TRANSPOSED.  A lane's slot of kOut chunks (128 bytes) is complete when its position is a multiple of
            //      kOut: at this step that is the lanes l = step - 2 kSlot (mod kOut), 64 / kOut rows.  Each of them is stored by the kOut
            //      lanes of its group, 16 bytes apiece: one store instruction per step, whose 64 pieces are eight runs of 128
            //      bytes (a lane storing its own row's bytes makes 64 pieces of 16 in 64 different lines).  The owner's position
            //      and row count follow from this lane's own -- positions differ by the lane distance --, where its row's output
            //      starts is in its rdesc entry.  The slot is read here, in front of the chunk its owner writes at this very
            //      step: one buffer is enough.
            wave_sync();
            const uint32_t owner = (lane & ~(uint32_t)(kOut - 1)) | ((step - 2 * kSlot) & (kOut - 1)), q = lane & (kOut - 1);  // (a lane's position is step - lane - 2 kSlot)
            int32_t co = c + (int32_t)lane - (int32_t)owner, ko = (int32_t)k;
            if (co >= (int32_t)P) {
                co -= (int32_t)P;
                ko++;
            } else if (co < 0) {
                co += (int32_t)P;
                ko--;
            }
            int32_t cfs = co - kOut;
            const bool back = cfs < 0;  // (that slot belongs to the row before)
            if (back) cfs += (int32_t)P;
            const int32_t kk = back ? ko - 1 : ko;
            const uint32_t cf = (uint32_t)cfs;
            const uint64_t o16 = (uint64_t)(cf + q) * 16;
            if (kk >= 0 && (co & (kOut - 1)) == 0 && cf + q < N && o16 < rb) {
                const uint4 ds = lds.rdesc[owner][kk & 1];
                if (ds.z) {
                    const uint4 v = lds.lout[owner][q];
                    uint8_t* const dst = reinterpret_cast<uint8_t*>(((uint64_t)ds.y << 32) | ds.x) + o16;
                    const uint32_t nb = (uint32_t)min((uint64_t)16, rb - o16);
                    if (nb == 16) png_store16(dst, v);
                    else png_store_part(dst, v, nb);
                }
            }
        }
        if ((step & (kSlot - 1)) == 0) {
            const int32_t ph = c & (kSlot - 1);
            {   // park the line fetched eight steps ago: the slot entered in (step, step + 8]
                const int32_t d1 = ph ? kSlot - ph : kSlot;
                int32_t c1 = c + d1;
                uint32_t k1 = k;
                if (c1 >= (int32_t)P) {
                    c1 -= (int32_t)P;
                    k1++;
                }
                const uint32_t par = (k1 * (P / kSlot) + ((uint32_t)c1 / kSlot)) & 1;
#pragma unroll
                for (int q = 0; q < kSlot; q++) lds.lin[lane][par][q] = pf[q];
            }
            {   // fetch the line of the slot entered in (step + 8, step + 16]
                const int32_t d2 = ph ? 2 * kSlot - ph : 2 * kSlot;
                int32_t c2 = c + d2;
                const bool wrap = c2 >= (int32_t)P;
                if (wrap) c2 -= (int32_t)P;
                const PngPipeRow& d = wrap ? next : cur;
                if ((uint32_t)c2 < N) {
                    fetch(d, (uint32_t)c2 / kSlot, pf);
                } else {
#pragma unroll
                    for (int q = 0; q < kSlot; q++) pf[q] = make_uint4(0, 0, 0, 0);
                }
            }
        }
        const uint32_t cu = (uint32_t)c;
        const bool on = cur.valid && cur.produced && cu < N;  // (cur.valid is false while c is negative)
        const uint32_t par = (k * (P / kSlot) + (cu / kSlot)) & 1;
        uint4 u = make_uint4(png_from_lane_below(last.x), png_from_lane_below(last.y), png_from_lane_below(last.z), png_from_lane_below(last.w));
        if (on) {
            const uint4 f = lds.lin[lane][par][cu & (kSlot - 1)];
            if (lane == 0) u = png_rowbuf[cu];
            if (cur.r == 0) u = make_uint4(0, 0, 0, 0);
            last = png_chunk<BPP, true>(f, u, la, ua, m);
            lds.lout[lane][cu & (kOut - 1)] = last;
            if (lane == kWave - 1) png_rowbuf[cu] = last;
        }
        c++;
    }
    if (lane < cnt) a.status[img0 + lane] = my_st ? my_st : (lds.bad[lane] != 0xFFFFFFFFu ? 1u : 0u);
}

}  // namespace fdh

// One image per wavefront by default; FDH_PNG_LANE_PER_IMAGE=1 selects the one-image-per-lane
// kernel (of use only for batches of very many images of a few rows each).
template <bool UNFILTER>
static int png_launch(const fdh::PngArgs& a, uint32_t bpp, hipStream_t stream) {
    if (a.n == 0) return 0;
    const char* e = getenv("FDH_PNG_LANE_PER_IMAGE");
    const bool per_lane = e && e[0] == '1';
    const char* e2 = getenv("FDH_PNG_NO_PIPELINE");
    // reconstruction of rows up to 4 KiB: the pipeline over several images per wavefront (more of
    // them when the batch is large enough to fill the GPU anyway)
    const bool pipe = UNFILTER && !per_lane && !(e2 && e2[0] == '1') && a.row_bytes <= 16 * fdh::kPipeMaxChunks;
    uint32_t per_wave = (uint32_t)std::min<uint64_t>(fdh::kPipeMaxImages, std::max<uint64_t>(1, a.n / 4096));
    if (const char* e3 = getenv("FDH_PNG_IMAGES_PER_WAVE")) {  // (tests: the multi-image pipeline on small batches)
        const int v = atoi(e3);
        if (v >= 1 && v <= (int)fdh::kPipeMaxImages) per_wave = (uint32_t)v;
    }
    const size_t rowbuf_bytes = (size_t)((a.row_bytes + 15) / 16) * 16;
    const dim3 block(fdh::kWave);
    const dim3 grid(per_lane ? (unsigned)((a.n + fdh::kWave - 1) / fdh::kWave) : pipe ? (unsigned)((a.n + per_wave - 1) / per_wave) : (unsigned)a.n);
#define FDH_PNG_CASE(B)                                                                                  \
    case B:                                                                                              \
        if (per_lane)                                                                                    \
            hipLaunchKernelGGL((fdh::png_filter_kernel<B, UNFILTER>), grid, block, 0, stream, a);        \
        else if (pipe)                                                                                   \
            hipLaunchKernelGGL((fdh::png_pipe_kernel<B>), grid, block, rowbuf_bytes, stream, a, per_wave); \
        else                                                                                             \
            hipLaunchKernelGGL((fdh::png_wave_kernel<B, UNFILTER>), grid, block, 0, stream, a);          \
        break;
    switch (bpp) {
        FDH_PNG_CASE(1)
        FDH_PNG_CASE(2)
        FDH_PNG_CASE(3)
        FDH_PNG_CASE(4)
        FDH_PNG_CASE(6)
        FDH_PNG_CASE(8)
        default: return -1;
    }
#undef FDH_PNG_CASE
    return (int)hipGetLastError();
}

extern "C" int fdh_launch_png_unfilter(const uint8_t* filt, const uint64_t* filt_off, uint8_t* pix, const uint64_t* pix_off,
                                       uint32_t* status, const uint32_t* gate, const uint32_t* gate_len, uint64_t n,
                                       uint32_t row_bytes, uint32_t bpp, hipStream_t stream) {
    fdh::PngArgs a{filt, filt_off, pix, pix_off, nullptr, nullptr, status, gate, gate_len, n, row_bytes};
    return png_launch<true>(a, bpp, stream);
}

extern "C" int fdh_launch_png_filter(const uint8_t* pix, const uint64_t* pix_off, const uint8_t* types,
                                     const uint64_t* types_off, uint8_t* filt, const uint64_t* filt_off, uint32_t* status,
                                     uint64_t n, uint32_t row_bytes, uint32_t bpp, hipStream_t stream) {
    fdh::PngArgs a{pix, pix_off, filt, filt_off, types, types_off, status, nullptr, nullptr, n, row_bytes};
    return png_launch<false>(a, bpp, stream);
}
