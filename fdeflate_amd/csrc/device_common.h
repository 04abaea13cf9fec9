// device_common.h -- shared device helpers for the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fdh {

constexpr int kWave = 64;

// Per-stream status codes (include/fdeflate_hip.h; ordinals of DecompressionError,
// reference src/decompress.rs:14-48).
enum StreamStatus : uint32_t {
    ST_OK = 0,
    ST_BAD_ZLIB_HEADER = 1,
    ST_INSUFFICIENT_INPUT = 2,
    ST_INVALID_BLOCK_TYPE = 3,
    ST_INVALID_UNCOMPRESSED_BLOCK_LENGTH = 4,
    ST_INVALID_HLIT = 5,
    ST_INVALID_HDIST = 6,
    ST_INVALID_CODE_LENGTH_REPEAT = 7,
    ST_BAD_CODE_LENGTH_HUFFMAN_TREE = 8,
    ST_BAD_LITERAL_LENGTH_HUFFMAN_TREE = 9,
    ST_BAD_DISTANCE_HUFFMAN_TREE = 10,
    ST_INVALID_LITERAL_LENGTH_CODE = 11,
    ST_INVALID_DISTANCE_CODE = 12,
    ST_INPUT_STARTS_WITH_RUN = 13,
    ST_DISTANCE_TOO_FAR_BACK = 14,
    ST_WRONG_CHECKSUM = 15,
    ST_EXTRA_INPUT = 16,
    ST_OUTPUT_TOO_LARGE = 17,
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ uint64_t uni64(uint64_t v) {
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// Orders LDS traffic between the lanes of one wavefront: LDS instructions of a wave execute in
// issue order, so all that is needed is to stop the compiler from moving memory operations
// across this point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// c ? a : b as one v_cndmask.  Opaque to the optimiser on purpose: hipcc otherwise rewrites
// chains of selects over a few registers into an indexed load from a scratch-memory table.
__device__ __forceinline__ uint32_t vsel(bool c, uint32_t a, uint32_t b) {
    uint32_t r;
    uint64_t m = __ballot(c);
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}

__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return ((uint64_t)1 << lane) - 1; }

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum of the 4 bytes of x / dot product of the 4 bytes of x with the 4 bytes of w
__device__ __forceinline__ uint32_t bytesum4(uint32_t x) { return __builtin_amdgcn_sad_u8(x, 0u, 0u); }
__device__ __forceinline__ uint32_t bytedot4(uint32_t x, uint32_t w, uint32_t acc) {
    return __builtin_amdgcn_udot4(x, w, acc, false);
}


// Inclusive prefix sum / maximum over the 64 lanes with DPP moves (row shifts inside the rows of 16 lanes, then the
// two row broadcasts of gfx9): no LDS traffic, a dozen instructions.  All 64 lanes must be active.
template <bool MAX>
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    auto op = [](uint32_t a, uint32_t b) __attribute__((always_inline)) { return MAX ? (a > b ? a : b) : a + b; };
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));  // row_shr:1
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));  // row_shr:2
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));  // row_shr:4
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));  // row_shr:8
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));  // row_bcast:15 -> rows 1, 3
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));  // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) { return wave_scan_incl<false>(v); }
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v) { return wave_scan_incl<true>(v); }

constexpr uint32_t kAdlerMod = 65521u;

// RFC-1951 length / distance symbol tables (reference src/tables.rs:68-88, data).
__device__ static const uint16_t kLenBase[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,
                                                 15, 17, 19, 23, 27, 31, 35, 43, 51,  59,
                                                 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                                 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ static const uint16_t kDistBase[30] = {
    1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
    193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2,  3,  3,  4,  4,  5,  5,  6,
                                                  6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
// Order of the code-length-code lengths (reference src/tables.rs:63-65, RFC 1951 3.2.7).
__device__ static const uint8_t kClclOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

}  // namespace fdh
