// Issue-rate micro-benchmark (gfx950): K independent chains of dependent VALU ops per wavefront, W wavefronts per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int K>
__global__ __launch_bounds__(64) void k_valu(uint32_t* out, int n) {
    uint32_t x[K];
    for (int k = 0; k < K; k++) x[k] = threadIdx.x * 17 + k;
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < K; k++) { x[k] = x[k] * 3 + 1; x[k] ^= x[k] >> 3; }
    }
    long long t1 = clock64();
    uint32_t s = 0;
    for (int k = 0; k < K; k++) s += x[k];
    if (threadIdx.x == 0) out[blockIdx.x] = (uint32_t)(t1 - t0);
    if (s == 0xdeadbeef) out[0] = s;
}
template <int K> void run(uint32_t* d, int wpc) {
    int n = 4096;
    hipLaunchKernelGGL(k_valu<K>, dim3(256 * wpc), dim3(64), 0, 0, d, n);
    (void)hipDeviceSynchronize();
    uint32_t h[4]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%2d waves/CU, %d chains: %.1f ticks per iteration (3 VALU ops per chain: mul-add, shift, xor) = %.2f ticks per op\n", wpc, K, (double)h[0] / n, (double)h[0] / n / (3 * K));
}
int main() {
    uint32_t* d; (void)hipMalloc(&d, 65536 * 4);
    for (int wpc : {1, 4, 8, 16, 32}) { run<1>(d, wpc); run<2>(d, wpc); run<4>(d, wpc); run<8>(d, wpc); }
    return 0;
}
