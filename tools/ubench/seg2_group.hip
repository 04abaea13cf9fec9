// The look-up groups of the interval decoder (inflate_seg2_groups.h) in isolation:
//   part A  correctness against a host model (random table with ~2 % zero entries, random input)
//   part B  cycles per look-up per wavefront / per SIMD at 16 wavefronts per CU (4 per SIMD), next to
//           the round-2 groups of inflate_segments.h
//   hipcc --offload-arch=gfx950 -O3 -I fdeflate_amd/csrc -o /tmp/s2g tools/ubench/seg2_group.hip && /tmp/s2g
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#define FDH_S2_R0 "40"
#define FDH_S2_R1 "41"
#define FDH_S2_R2 "42"
#define FDH_S2_R3 "43"
#define FDH_S2_R4 "44"
#define FDH_S2_R5 "45"
#define FDH_S2_R6 "46"
#define FDH_S2_R7 "47"
#include "inflate_seg2_groups.h"
using namespace fdh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); } } while (0)

constexpr int kWavesPerBlock = 16;
constexpr int kTableWords = 4096;
constexpr int kAWords = 1024;   // 4 KiB per wavefront: ring (count) / flat input image (write)
constexpr int kBWords = 1280;   // 5 KiB per wavefront: output image
struct Lds {
    uint32_t lit[kTableWords];
    uint32_t a[kWavesPerBlock * kAWords];
    uint32_t b[kWavesPerBlock * kBWords];
};

__device__ __forceinline__ uint32_t lds_off(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

__host__ __device__ inline uint32_t table_entry(uint32_t i) {
    uint32_t h = i * 2654435761u;
    h ^= h >> 15;
    if ((h & 63) == 0) return 0;  // ~1.6 %: run / end-of-block / impossible
    const uint32_t n = 1 + (h >> 28) % 3;
    uint32_t used = 2 * n + ((h >> 20) & 3) * n;
    if (used > 12) used = 12;
    const uint32_t lits = (h >> 4) & (n == 3 ? 0xffffffu : (n == 2 ? 0xffffu : 0xffu));
    return used | (n << 6) | (lits << 8);
}
__host__ __device__ inline uint32_t rnd(uint32_t& x) {
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return x;
}

// MODE 0: count group, ring input; MODE 1: write group, flat input.
// check != 0: one call of `pairs` pairs, state and images copied out.
template <int MODE>
__global__ __launch_bounds__(1024, 4) void k(uint32_t* out, uint32_t* img, int groups, uint32_t pairs, uint64_t* cycles,
                                              int check) {
    __shared__ Lds lds;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < kTableWords; i += blockDim.x) lds.lit[i] = table_entry(i);
    uint32_t x = 0x9E3779B9u * (blockIdx.x + 1) + 77u;
    for (int i = threadIdx.x; i < kWavesPerBlock * kAWords; i += blockDim.x) {
        uint32_t y = x + i * 0x85ebca6bu;
        rnd(y);
        rnd(y);
        lds.a[i] = y;
    }
    for (int i = threadIdx.x; i < kWavesPerBlock * kBWords; i += blockDim.x) lds.b[i] = 0;
    __syncthreads();
    uint32_t* A = lds.a + wid * kAWords;
    uint32_t* B = lds.b + wid * kBWords;
    uint32_t lo, hi, c, ra, acc = 0;
    uint32_t rb = 0;
    const uint32_t boff = (lane * 5 + wid) & 31;
    if (MODE == 0) {
        rb = lds_off(A) + 4 * lane;  // word w at rb + (w & 15) * 256
        lo = A[lane];
        hi = A[64 + lane];
        ra = rb + 2 * 256;
        c = boff | ((uint32_t)lane << 6);
    } else {
        // flat image: lane's words start at word 12 * lane (48 B apart; a group of 8 look-ups reads < 5 words)
        lo = A[12 * lane];
        hi = A[12 * lane + 1];
        ra = lds_off(A) + 4 * (12 * lane + 2);
        c = boff | ((lds_off(B) + 40 * lane + ((lane * 7) & 3)) << 6);
    }
    const uint32_t lo0 = lo, hi0 = hi, c0 = c, ra0 = ra;
    const long long t0 = clock64();
    for (int g = 0; g < groups; g++) {
        if (MODE == 0) {
            seg2_count_group(pairs, rb, lo, hi, c, ra);
        } else {
            if (!check) {  // timing: stay inside the images
                lo = lo0;
                hi = hi0;
                ra = ra0;
                c = (c & 63) | (c0 & ~63u);
            }
            seg2_write_group(pairs, lo, hi, c, ra, acc);
        }
    }
    const long long t1 = clock64();
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    out[tid * 8 + 0] = lo;
    out[tid * 8 + 1] = hi;
    out[tid * 8 + 2] = c;
    out[tid * 8 + 3] = ra - (MODE == 0 ? rb : lds_off(A));
    out[tid * 8 + 4] = acc;
    out[tid * 8 + 5] = c0;
    if (check) {
        __syncthreads();
        for (int i = threadIdx.x; i < kWavesPerBlock * kAWords; i += blockDim.x) img[i] = lds.a[i];
        for (int i = threadIdx.x; i < kWavesPerBlock * kBWords; i += blockDim.x) img[kWavesPerBlock * kAWords + i] = lds.b[i];
    }
    if (threadIdx.x == 0) cycles[blockIdx.x] = (uint64_t)(t1 - t0);
}

static int check_count(uint32_t pairs) {
    uint32_t *out, *img;
    uint64_t* cyc;
    hipMalloc(&out, 1024 * 8 * 4);
    hipMalloc(&img, (kWavesPerBlock * (kAWords + kBWords)) * 4);
    hipMalloc(&cyc, 8);
    printf("launch count check\n");
    k<0><<<1, 1024>>>(out, img, 1, pairs, cyc, 1);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    printf("done\n");
    std::vector<uint32_t> ho(1024 * 8), hi(kWavesPerBlock * (kAWords + kBWords));
    hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hi.data(), img, hi.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 1024; t++) {
        const int lane = t & 63, wid = t >> 6;
        const uint32_t* A = hi.data() + wid * kAWords;
        auto word = [&](uint32_t w) { return A[(w & 15) * 64 + lane]; };
        uint64_t pos = ((lane * 5 + wid) & 31);  // bit position of the window's bit 0 (token at +2)
        uint32_t cnt = lane;
        for (uint32_t s = 0; s < 2 * pairs; s++) {
            const uint32_t w = (uint32_t)(pos >> 5), b = (uint32_t)(pos & 31);
            const uint64_t win = ((uint64_t)word(w + 1) << 32 | word(w)) >> b;
            const uint64_t win2 = b ? ((uint64_t)word(w + 2) << (64 - b)) : 0;
            const uint32_t idx = (uint32_t)((win | win2) >> 2) & 4095;
            const uint32_t e = table_entry(idx);
            pos += e & 15;
            cnt += (e >> 6) & 3;
        }
        const uint32_t w = (uint32_t)(pos >> 5);
        const uint32_t exp_c = (cnt << 6) | (uint32_t)(pos & 31);
        const bool ok = ho[t * 8 + 2] == exp_c && ho[t * 8 + 0] == word(w) && ho[t * 8 + 1] == word(w + 1) &&
                        ho[t * 8 + 3] == ((w + 2) & 15) * 256;
        if (!ok && bad++ < 6)
            printf("count mismatch thread %d: c %08x exp %08x lo %08x exp %08x hi %08x exp %08x ra %x exp %x\n", t, ho[t * 8 + 2],
                   exp_c, ho[t * 8 + 0], word(w), ho[t * 8 + 1], word(w + 1), ho[t * 8 + 3], ((w + 2) & 15) * 256);
    }
    printf("count group, %u pairs: %s (%d threads wrong)\n", pairs, bad ? "BROKEN" : "ok", bad);
    hipFree(out);
    hipFree(img);
    hipFree(cyc);
    return bad;
}

static int check_write(uint32_t pairs) {
    uint32_t *out, *img;
    uint64_t* cyc;
    hipMalloc(&out, 1024 * 8 * 4);
    hipMalloc(&img, (kWavesPerBlock * (kAWords + kBWords)) * 4);
    hipMalloc(&cyc, 8);
    printf("launch write check\n");
    k<1><<<1, 1024>>>(out, img, 1, pairs, cyc, 1);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    printf("done\n");
    std::vector<uint32_t> ho(1024 * 8), hi(kWavesPerBlock * (kAWords + kBWords));
    hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hi.data(), img, hi.size() * 4, hipMemcpyDeviceToHost);
    std::vector<uint8_t> exp(kWavesPerBlock * kBWords * 4, 0);
    int bad = 0;
    const uint32_t b_base = (kTableWords + kWavesPerBlock * kAWords) * 4;  // LDS offset of lds.b (struct layout)
    for (int t = 0; t < 1024; t++) {
        const int lane = t & 63, wid = t >> 6;
        const uint32_t* A = hi.data() + wid * kAWords + 12 * lane;
        uint64_t pos = ((lane * 5 + wid) & 31);
        const uint32_t c0 = ho[t * 8 + 5];
        uint32_t addr = c0 >> 6;
        if (addr != b_base + wid * kBWords * 4 + 40 * lane + ((lane * 7) & 3)) {
            if (bad++ < 6) printf("write: unexpected start address thread %d: %x\n", t, addr);
            continue;
        }
        for (uint32_t s = 0; s < 2 * pairs; s++) {
            const uint32_t w = (uint32_t)(pos >> 5), b = (uint32_t)(pos & 31);
            const uint64_t win = ((uint64_t)A[w + 1] << 32 | A[w]) >> b;
            const uint64_t win2 = b ? ((uint64_t)A[w + 2] << (64 - b)) : 0;
            const uint32_t idx = (uint32_t)((win | win2) >> 2) & 4095;
            const uint32_t e = table_entry(idx);
            pos += e & 15;
            for (uint32_t j = 0; j < ((e >> 6) & 3); j++) exp[addr++ - b_base] |= (uint8_t)(e >> (8 + 8 * j));
        }
        const uint32_t w = (uint32_t)(pos >> 5);
        const uint32_t exp_c = (addr << 6) | (uint32_t)(pos & 31);
        const bool ok = ho[t * 8 + 2] == exp_c && ho[t * 8 + 0] == A[w] && ho[t * 8 + 1] == A[w + 1] &&
                        ho[t * 8 + 3] == 4 * (12 * lane + w + 2);
        if (!ok && bad++ < 6)
            printf("write state mismatch thread %d: c %08x exp %08x lo %08x exp %08x ra %x exp %x\n", t, ho[t * 8 + 2], exp_c,
                   ho[t * 8 + 0], A[w], ho[t * 8 + 3], 4 * (12 * lane + w + 2));
    }
    const uint8_t* got = reinterpret_cast<const uint8_t*>(hi.data() + kWavesPerBlock * kAWords);
    int badb = 0;
    for (size_t i = 0; i < exp.size(); i++)
        if (got[i] != exp[i] && badb++ < 6) printf("write image mismatch at byte %zu: %02x exp %02x\n", i, got[i], exp[i]);
    printf("write group, %u pairs: %s (%d threads, %d bytes wrong)\n", pairs, (bad || badb) ? "BROKEN" : "ok", bad, badb);
    hipFree(out);
    hipFree(img);
    hipFree(cyc);
    return bad + badb;
}

template <int MODE>
void run(const char* name, uint32_t pairs) {
    const int blocks = 256;
    uint32_t *out, *img;
    uint64_t* cyc;
    hipMalloc(&out, (size_t)blocks * 1024 * 8 * 4);
    hipMalloc(&img, (kWavesPerBlock * (kAWords + kBWords)) * 4);
    hipMalloc(&cyc, blocks * 8);
    const int groups = 2000;
    k<MODE><<<blocks, 1024>>>(out, img, 10, pairs, cyc, 0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<blocks, 1024>>>(out, img, groups, pairs, cyc, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < blocks; i++) avg += h[i];
    avg /= blocks;
    const double steps = (double)groups * 2 * pairs;
    printf("%-6s %2u pairs, 4 waves/SIMD: %.3f ms, %.0f cycles per look-up per wavefront (clock64), %.1f per SIMD; %.2f ns per look-up per SIMD\n",
           name, pairs, ms, avg / steps, avg / steps / 4, ms * 1e6 / steps / 4);
    hipFree(out);
    hipFree(img);
    hipFree(cyc);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    int bad = 0;
    for (uint32_t p : {1u, 2u, 4u}) bad += check_count(p);
    for (uint32_t p : {1u, 2u, 4u}) bad += check_write(p);
    for (uint32_t p : {4u, 8u}) run<0>("count", p);
    for (uint32_t p : {4u, 8u, 16u}) run<1>("write", p);
    return bad ? 1 : 0;
}
