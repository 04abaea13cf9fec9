// Microbenchmark of the hand-scheduled decode groups of inflate_segments.h in isolation: a table of
// valid literal entries and per-lane rings of random words in LDS, no global memory in the loop.
// Prints cycles per table look-up ("step") per wavefront and per SIMD for 1 / 2 / 4 wavefronts per
// SIMD: tells whether the loops are bound by instruction issue, by the LDS or by the dependent chain.
//   hipcc --offload-arch=gfx950 -O3 -I fdeflate_amd/csrc -o /tmp/sg tools/ubench/seg_group.hip && /tmp/sg
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define FDH_SEG_R0 "40"
#define FDH_SEG_R1 "41"
#define FDH_SEG_R2 "42"
#define FDH_SEG_R3 "43"
#include "inflate_segments.h"
using namespace fdh;

template <int MODE>  // 0 count group, 1 write group
__global__ __launch_bounds__(1024, 8) void k(uint32_t* out, int groups, uint64_t* cycles) {
    __shared__ SegLds lds;  // 80 KiB: table + 8 waves of rings; waves >= 8 share rings pairwise (timing only)
    const int lane = threadIdx.x & 63, wid = (threadIdx.x >> 6) & 7;
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = threadIdx.x; i < kLitSize; i += blockDim.x) {
        uint32_t h = i * 2654435761u;
        uint32_t n = 1 + (h >> 28) % 3;
        uint32_t used = 2 * n + ((h >> 20) & 3) * n;  // 2..12... keep <= 12
        if (used > 12) used = 12;
        lds.lit[i] = used | (n << 4) | ((h & 0xffffff) << 8 & ~(n < 3 ? (n < 2 ? 0xffff0000u : 0xff000000u) : 0u));
    }
    for (int i = threadIdx.x; i < kSegWaves * kSegInWords * kWave; i += blockDim.x) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        lds.in_ring[i] = x;
        lds.out_ring[i] = 0;
    }
    __syncthreads();
    SegReader rd;
    rd.ring = lds.in_ring;
    rd.lane_off = (uint32_t)wid * (kSegInWords * kWave) + lane;
    rd.in_rd = 2; rd.in_wr = 16; rd.lo = x; rd.hi = x * 7; rd.boff = lane & 31;
    const uint32_t ring_base = lds_offset(lds.in_ring) + 4 * rd.lane_off;
    const uint32_t out_base = lds_offset(lds.out_ring) + 4 * rd.lane_off;
    SegWriter wr{0, 0, 0};
    uint32_t cnt = 0, last = 0, eA = 0, eB = 0;
    const long long t0 = clock64();
    for (int g = 0; g < groups; g++) {
        if (MODE == 0) seg_count_group(kSegPairs, ring_base, rd, cnt, last, eA, eB);
        else seg_write_group(kSegPairs, ring_base, out_base, rd, wr, last, eA, eB);
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = cnt + last + eA + eB + rd.boff + wr.acc + wr.vposw;
    if (threadIdx.x == 0) cycles[blockIdx.x] = (uint64_t)(t1 - t0);
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
    // 1-4 waves per SIMD: one workgroup per CU; 6 / 8: two workgroups of 12 / 16 wavefronts (80 KiB of LDS each)
    const int blocks = waves_per_simd > 4 ? 512 : 256, threads = waves_per_simd > 4 ? 128 * waves_per_simd : 256 * waves_per_simd;
    uint32_t* out; uint64_t* cyc;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    hipMalloc(&cyc, blocks * 8);
    const int groups = 2000;
    k<MODE><<<blocks, threads>>>(out, 10, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, groups, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; i++) avg += h[i]; avg /= 256;
    const double steps = (double)groups * kSegSteps;
    printf("%-6s %d waves/SIMD: %.3f ms, %.0f cycles per step per wavefront (clock64), %.1f cycles per step per SIMD\n", name,
           waves_per_simd, ms, avg / steps, avg / steps / waves_per_simd);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {1, 2, 4, 6, 8}) run<0>("count", w);
    for (int w : {1, 2, 4, 6, 8}) run<1>("write", w);
    return 0;
}
