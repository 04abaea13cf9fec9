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
    uint64_t n;
    uint32_t row_bytes;
};

__device__ __forceinline__ uint32_t png_paeth(uint32_t a, uint32_t b, uint32_t c) {
    const int p = (int)a + (int)b - (int)c;
    const int pa = abs(p - (int)a), pb = abs(p - (int)b), pc = abs(p - (int)c);
    const uint32_t bc = pb <= pc ? b : c;
    return (pa <= pb && pa <= pc) ? a : bc;
}

struct PngMasks {  // 0xFF for the row's type, 0 otherwise: pred = OR of (candidate & mask)
    uint32_t sub, up, avg, paeth;
    __device__ explicit PngMasks(uint32_t t) : sub(t == 1 ? 0xFFu : 0u), up(t == 2 ? 0xFFu : 0u), avg(t == 3 ? 0xFFu : 0u), paeth(t == 4 ? 0xFFu : 0u) {}
    __device__ __forceinline__ uint32_t pred(uint32_t a, uint32_t b, uint32_t c) const {
        return (a & sub) | (b & up) | (((a + b) >> 1) & avg) | (png_paeth(a, b, c) & paeth);
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

template <int BPP, bool UNFILTER>
__global__ __launch_bounds__(kWave) void png_filter_kernel(PngArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * kWave + threadIdx.x;
    if (i >= a.n) return;
    if (a.gate && a.gate[i] != 0) {
        a.status[i] = 3;
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
__device__ __forceinline__ uint4 png_load_part(const uint8_t* p, uint32_t valid) {
    if (valid >= 16) return png_load16(p);
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < valid; k++) w[k >> 2] |= (uint32_t)p[k] << (8 * (k & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ void png_store_part(uint8_t* p, const uint4& v, uint32_t valid) {
    if (valid >= 16) {
        png_store16(p, v);
        return;
    }
    for (uint32_t k = 0; k < valid; k++) p[k] = (uint8_t)png_byte(v, (int)k);
}
__device__ __forceinline__ uint32_t png_from_lane_below(uint32_t x) {  // lane j gets lane j - 1's value (lane 0: 0)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}

// One image per WAVEFRONT: lane j takes row j of a band of 64 rows.  Reconstruction needs the row
// above, so the lanes run skewed by one 16-byte chunk: at step t lane j works on chunk t - j, and
// the chunk above it is what lane j - 1 produced one step earlier -- it arrives through a DPP lane
// shift, not through memory (row 0 of a later band reads the last row of the band before from the
// output).  Filtering has no such dependence (the predictors use raw neighbours): no skew, the row
// above is read from the input.  Same per-chunk arithmetic as the lane-per-image kernel.
template <int BPP, bool UNFILTER>
__global__ __launch_bounds__(kWave) void png_wave_kernel(PngArgs a) {
    const uint64_t i = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    if (a.gate && a.gate[i] != 0) {
        if (lane == 0) a.status[i] = 3;
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
        uint4 last = make_uint4(0, 0, 0, 0);  // the chunk this lane produced in the step before
        const uint32_t steps = band ? nchunks + (UNFILTER ? band - 1 : 0) : 0;
        for (uint32_t step = 0; step < steps; step++) {
            uint4 u = make_uint4(0, 0, 0, 0);
            if (UNFILTER) u = make_uint4(png_from_lane_below(last.x), png_from_lane_below(last.y), png_from_lane_below(last.z), png_from_lane_below(last.w));
            const uint32_t c = step - (UNFILTER ? lane : 0);  // (wraps for the lanes that have not started)
            if (mine && c < nchunks) {
                const uint32_t valid = (uint32_t)min((uint64_t)16, rb - (uint64_t)c * 16);
                if (UNFILTER ? lane == 0 : true) {
                    // the row above from memory: the output of the band before / the raw input
                    u = make_uint4(0, 0, 0, 0);
                    if (r > 0) u = png_load_part((UNFILTER ? out : in) - rb + (uint64_t)c * 16, valid);
                }
                const uint4 f = png_load_part(in + (uint64_t)c * 16, valid);
                last = png_chunk<BPP, UNFILTER>(f, u, la, ua, m);
                png_store_part(out + (uint64_t)c * 16, last, valid);
            }
        }
        // the next band's first row reads this band's last row back
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) a.status[i] = st;
}

}  // namespace fdh

// One image per wavefront by default; FDH_PNG_LANE_PER_IMAGE=1 selects the one-image-per-lane
// kernel (of use only for batches of very many images of a few rows each).
template <bool UNFILTER>
static int png_launch(const fdh::PngArgs& a, uint32_t bpp, hipStream_t stream) {
    if (a.n == 0) return 0;
    const char* e = getenv("FDH_PNG_LANE_PER_IMAGE");
    const bool per_lane = e && e[0] == '1';
    const dim3 block(fdh::kWave);
    const dim3 grid(per_lane ? (unsigned)((a.n + fdh::kWave - 1) / fdh::kWave) : (unsigned)a.n);
#define FDH_PNG_CASE(B)                                                                                  \
    case B:                                                                                              \
        if (per_lane)                                                                                    \
            hipLaunchKernelGGL((fdh::png_filter_kernel<B, UNFILTER>), grid, block, 0, stream, a);        \
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
                                       uint32_t* status, const uint32_t* gate, uint64_t n, uint32_t row_bytes, uint32_t bpp,
                                       hipStream_t stream) {
    fdh::PngArgs a{filt, filt_off, pix, pix_off, nullptr, nullptr, status, gate, n, row_bytes};
    return png_launch<true>(a, bpp, stream);
}

extern "C" int fdh_launch_png_filter(const uint8_t* pix, const uint64_t* pix_off, const uint8_t* types,
                                     const uint64_t* types_off, uint8_t* filt, const uint64_t* filt_off, uint32_t* status,
                                     uint64_t n, uint32_t row_bytes, uint32_t bpp, hipStream_t stream) {
    fdh::PngArgs a{pix, pix_off, filt, filt_off, types, types_off, status, nullptr, n, row_bytes};
    return png_launch<false>(a, bpp, stream);
}
