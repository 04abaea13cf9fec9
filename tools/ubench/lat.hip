// Latency micro-benchmarks (gfx950): dependent LDS reads, dependent VALU, VALU -> SGPR -> VALU, taken branches.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lat.hip -o tools/ubench/bin/lat && tools/ubench/bin/lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(64) void k_lds(uint32_t* out, int n, int mode) {
    __shared__ uint32_t t[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) t[i] = (i * 2654435761u >> 7) & 4095;
    __syncthreads();
    uint32_t x = threadIdx.x * 17, acc = 0;
    long long t0 = clock64();
    if (mode == 0) {  // dependent LDS reads (random addresses)
        for (int i = 0; i < n; i++) x = t[x & 4095];
    } else if (mode == 1) {  // dependent VALU adds / xors
        for (int i = 0; i < n; i++) { x = x * 3 + 1; x ^= x >> 3; x += acc; acc = x & 7; }
    } else if (mode == 2) {  // VALU compare -> SGPR mask -> cndmask, dependent
        for (int i = 0; i < n; i++) { x = (x & 8) ? x + 3 : x ^ 5; x = (x & 16) ? x + 7 : x ^ 9; }
    } else if (mode == 3) {  // uniform branch per iteration on a ballot
        for (int i = 0; i < n; i++) { if (__ballot(x & 1)) x += 3; else x ^= 1; if (__ballot(x & 2)) x += 5; else x ^= 2; }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = (uint32_t)(t1 - t0);
    if (x == 0xdeadbeef) out[0] = acc;
}
int main() {
    uint32_t* d; hipMalloc(&d, 4096 * 4);
    const char* names[] = {"dependent LDS read (random)", "dependent VALU (5 ops/iter)", "cmp->sgpr->cndmask x2 (about 6 ops/iter)", "2 ballots + uniform branches/iter"};
    for (int wpc = 1; wpc <= 8; wpc *= 2 ) {
        for (int mode = 0; mode < 4; mode++) {
            int n = 4096;
            hipLaunchKernelGGL(k_lds, dim3(256 * wpc), dim3(64), 0, 0, d, n, mode);
            hipDeviceSynchronize();
            uint32_t h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("%d blocks/CU: %-45s %.1f cycles/iter\n", wpc, names[mode], (double)h[0] / n);
        }
    }
    return 0;
}
