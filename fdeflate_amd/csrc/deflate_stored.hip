// deflate_stored.hip -- batched level-0 ("stored") zlib encoder.
//
// Restates `Compressor::new(w, 0, true)` + `write_data` + `finish`, i.e.
// `compress_to_vec_with_level(input, 0)` (reference src/compress/mod.rs:299-303): zlib header
// 78 01 (:69-71), stored blocks of at most 65535 bytes (:234-268: while at least a full block is
// buffered it is written with BFINAL = 0 -- three zero header bits padded to a byte, LEN, NLEN, the
// bytes; `finish` writes what is left as the final block, or an empty fixed block `write_bits(3, 10)`
// when nothing is left), Adler-32 big-endian (:208-211).  One buffer per wavefront: the layout is
// known up front (output byte of input byte i = 2 + 5 (i / 65535 + 1) + i), so the block payloads
// are copied 8 bytes per lane (both sides unaligned) and the checksum is accumulated on the way.
#include "device_common.h"

namespace fdh {

constexpr int kStoWaves = 4;
constexpr uint64_t kStoredMax = 65535;  // STORED_BLOCK_MAX_SIZE

struct StoredBatchArgs {
    const uint8_t* in;
    const uint64_t* in_off;
    uint8_t* out;
    const uint64_t* out_off;
    uint32_t* out_len;
    uint64_t n;
};

__device__ __forceinline__ uint64_t stored_size(uint64_t len) {
    const uint64_t nb = len / kStoredMax, rem = len - nb * kStoredMax;
    return 2 + nb * (5 + kStoredMax) + (rem ? 5 + rem : 2) + 4;
}

__global__ __launch_bounds__(kStoWaves* kWave) void deflate_stored_kernel(StoredBatchArgs a) {
    const int lane = threadIdx.x & (kWave - 1);
    const uint64_t sid = (uint64_t)blockIdx.x * kStoWaves + threadIdx.x / kWave;
    if (sid >= a.n) return;
    const uint8_t* in = a.in + a.in_off[sid];
    const uint64_t len = a.in_off[sid + 1] - a.in_off[sid];
    uint8_t* out = a.out + a.out_off[sid];
    const uint64_t cap = a.out_off[sid + 1] - a.out_off[sid];
    const uint64_t need = stored_size(len);
    if (need > cap || need > 0xFFFFFFF0ull) {  // slot too small: nothing valid is written
        if (lane == 0) a.out_len[sid] = 0xFFFFFFFFu;
        return;
    }
    const uint64_t nb = len / kStoredMax, rem = len - nb * kStoredMax;
    if (lane == 0) {
        out[0] = 0x78;
        out[1] = 0x01;
    }
    // A = 1 + sum d_i ; B = len + sum (len - i) d_i  (mod 65521), per-lane partial sums
    uint64_t acc_a = 0, acc_b = 0;
    const uint64_t nblocks = nb + (rem ? 1 : 0);
    for (uint64_t b = 0; b < nblocks; b++) {
        const uint64_t i0 = b * kStoredMax;
        const uint32_t n = (uint32_t)(b < nb ? kStoredMax : rem);
        uint8_t* dst = out + 2 + 5 * (b + 1) + i0;
        if (lane == 0) {  // header: BFINAL only on the block written by finish()
            uint8_t* h = dst - 5;
            h[0] = (b == nb) ? 0x01 : 0x00;
            h[1] = (uint8_t)n;
            h[2] = (uint8_t)(n >> 8);
            h[3] = (uint8_t)~n;
            h[4] = (uint8_t)(~n >> 8);
        }
        const uint8_t* src = in + i0;
        const uint32_t n8 = n & ~7u;
        for (uint32_t k = (uint32_t)lane * 8; k < n8; k += kWave * 8) {
            const uint64_t x = *reinterpret_cast<const uint64_t*>(src + k);  // HW handles misalignment
            *reinterpret_cast<uint64_t*>(dst + k) = x;
            const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
            const uint32_t s = bytesum4(xl) + bytesum4(xh);
            uint32_t u = bytedot4(xl, 0x03020100u, 0);
            u = bytedot4(xh, 0x07060504u, u);
            acc_a += s;
            acc_b += (uint64_t)(len - (i0 + k)) * s - u;
        }
        if ((uint32_t)lane < n - n8) {
            const uint32_t k = n8 + (uint32_t)lane;
            const uint32_t d = src[k];
            dst[k] = (uint8_t)d;
            acc_a += d;
            acc_b += (uint64_t)(len - (i0 + k)) * d;
        }
        acc_a %= kAdlerMod;  // a block adds < 2^16 * 255 * len: reduce once per block
        acc_b %= kAdlerMod;
    }
    const uint32_t A = (1u + wave_sum_u32((uint32_t)acc_a)) % kAdlerMod;
    const uint32_t B = (uint32_t)(((len % kAdlerMod) + wave_sum_u32((uint32_t)acc_b)) % kAdlerMod);
    if (lane == 0) {
        uint8_t* t = out + need - 4;
        if (rem == 0) {  // finish() with nothing buffered: write_bits(3, 10) = 03 00
            t[-2] = 0x03;
            t[-1] = 0x00;
        }
        t[0] = (uint8_t)(B >> 8);
        t[1] = (uint8_t)B;
        t[2] = (uint8_t)(A >> 8);
        t[3] = (uint8_t)A;
        a.out_len[sid] = (uint32_t)need;
    }
}

}  // namespace fdh

extern "C" int fdh_launch_deflate_stored(const uint8_t* in, const uint64_t* in_off, uint8_t* out,
                                         const uint64_t* out_off, uint32_t* out_len, uint64_t n, hipStream_t stream) {
    if (n == 0) return 0;
    fdh::StoredBatchArgs a{in, in_off, out, out_off, out_len, n};
    unsigned blocks = (unsigned)((n + fdh::kStoWaves - 1) / fdh::kStoWaves);
    hipLaunchKernelGGL(fdh::deflate_stored_kernel, dim3(blocks), dim3(fdh::kStoWaves * fdh::kWave), 0, stream, a);
    return (int)hipGetLastError();
}

// ---- 16-byte lines from one place of device memory to another (stream_decompressor.cpp) ----
// The streaming object moves what it keeps of a stream to the front of its buffers with this kernel (a few hundred KiB at
// most, on the object's own stream, between two decode attempts).  src and dst 16-byte aligned, ranges must not overlap.
namespace fdh {
__global__ __launch_bounds__(256) void copy_lines_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t lines) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < lines; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
}  // namespace fdh

extern "C" int fdh_launch_copy_lines(void* dst, const void* src, size_t bytes, hipStream_t stream) {
    const size_t lines = (bytes + 15) / 16;
    if (lines == 0) return 0;
    const unsigned blocks = (unsigned)std::min<size_t>((lines + 255) / 256, 1024);
    hipLaunchKernelGGL(fdh::copy_lines_kernel, dim3(blocks), dim3(256), 0, stream, static_cast<uint4*>(dst), static_cast<const uint4*>(src), lines);
    return (int)hipGetLastError();
}

