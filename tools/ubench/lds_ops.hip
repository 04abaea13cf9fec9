// LDS microbenchmark (gfx950): what the per-lane decode loops of inflate_segments.h pay for their
// table look-ups and ring accesses.
//   part A  correctness of UNALIGNED ds_write_b32 / ds_read_b32 / ds_read_b64 (byte offsets 1..3)
//   part B  cycles per wavefront-instruction per CU for random look-ups of 1/2/4/8-byte entries and
//           for aligned / unaligned ring writes, every CU busy, 16 wavefronts per CU
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/lds tools/ubench/lds_ops.hip && /tmp/lds
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void unaligned_check(uint32_t* res) {
    __shared__ uint32_t mem[64 * 20];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 20; i += 64) mem[i] = 0;
    __syncthreads();
    // every lane owns 80 bytes; write a dword at byte offset (lane & 3) + 4, then read back around it
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)mem + lane * 80;
    const uint32_t off = base + 4 + (lane & 3);
    const uint32_t val = 0xA1B2C3D4u + lane;
    asm volatile("ds_write_b32 %0, %1\n s_waitcnt lgkmcnt(0)" ::"v"(off), "v"(val) : "memory");
    uint32_t w0, w1, w2;
    asm volatile("ds_read_b32 %0, %3\n ds_read_b32 %1, %3 offset:4\n ds_read_b32 %2, %3 offset:8\n s_waitcnt lgkmcnt(0)"
                 : "=v"(w0), "=v"(w1), "=v"(w2) : "v"(base) : "memory");
    uint32_t u32;
    uint64_t u64;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(u32) : "v"(off) : "memory");
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(u64) : "v"(off) : "memory");
    res[lane * 8 + 0] = w0;
    res[lane * 8 + 1] = w1;
    res[lane * 8 + 2] = w2;
    res[lane * 8 + 3] = u32;
    res[lane * 8 + 4] = (uint32_t)u64;
    res[lane * 8 + 5] = (uint32_t)(u64 >> 32);
    res[lane * 8 + 6] = val;
}

// OP: 0 ds_read_b32 random in 16 KiB | 1 ds_read_b64 random in 32 KiB | 2 ds_read_u8 random in 16 KiB
//     3 ds_read_u16 random in 32 KiB | 4 ds_write_b32 conflict-free ([word][lane]) | 5 ds_write_b32
//     unaligned, lane-major 72-B slots | 6 ds_write_b32 aligned, lane-major 72-B slots, random word
//     7 ds_read_b32 conflict-free | 8 ds_write_b64 unaligned lane-major | 9 ds_read_b32 skewed (PNG-like:
//     half of the lanes hit 4 hot entries) | 10 ds_read_b64 skewed | 11 ds_read_u8 skewed
template <int OP>
__global__ __launch_bounds__(1024) void lds_rate(uint32_t* out, int iters, uint32_t seed) {
    __shared__ uint32_t mem[40960 / 4 * 2];  // 80 KiB
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 40960 / 4 * 2; i += blockDim.x) mem[i] = i * 2654435761u;
    __syncthreads();
    uint32_t a[8];
    uint32_t x = (threadIdx.x + 1) * 2654435761u ^ seed ^ (blockIdx.x * 40503u);
    for (int k = 0; k < 8; k++) {
        x ^= x << 13;
        x ^= x >> 17;
        x ^= x << 5;
        a[k] = x;
    }
    uint32_t acc = 0;
    const uint32_t lm_base = 32768 + ((uint32_t)(wave & 7) * 64 + lane) * 72;  // lane-major slot (72 B), 8 waves' worth
    const uint32_t il_base = 32768 + (uint32_t)(wave & 7) * 4096 + lane * 4;    // [word][lane]
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t r = a[k];
            uint32_t addr, v = 0;
            const bool hot = (r >> 20) & 1;
            if (OP == 0) addr = r & 0x3ffc;
            if (OP == 1) addr = r & 0x7ff8;
            if (OP == 2) addr = r & 0x3fff;
            if (OP == 3) addr = r & 0x7ffe;
            if (OP == 4) addr = il_base + ((r & 15) << 8);
            if (OP == 5) addr = lm_base + (r & 63);
            if (OP == 6) addr = lm_base + (r & 60);
            if (OP == 7) addr = il_base + ((r & 15) << 8);
            if (OP == 8) addr = lm_base + (r & 63);
            if (OP == 9) addr = hot ? (r & 0xc) : (r & 0x3ffc);
            if (OP == 10) addr = hot ? (r & 0x18) : (r & 0x7ff8);
            if (OP == 11) addr = hot ? (r & 0x3) : (r & 0x3fff);
            if (OP == 0 || OP == 7 || OP == 9) asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
            if (OP == 1 || OP == 10) {
                uint64_t q;
                asm volatile("ds_read_b64 %0, %1" : "=v"(q) : "v"(addr) : "memory");
                asm volatile("" ::"v"(q));
            }
            if (OP == 2 || OP == 11) asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(addr) : "memory");
            if (OP == 3) asm volatile("ds_read_u16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
            if (OP == 4 || OP == 5 || OP == 6) asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(r) : "memory");
            if (OP == 8) {
                uint64_t q = r;
                asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(q) : "memory");
            }
            a[k] = r * 5 + 0x9E3779B9u + v * 0;  // next pseudo-random address (does not wait for the read)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (int k = 0; k < 8; k++) acc += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
void run(const char* name) {
    uint32_t* out;
    const int blocks = 256, threads = 1024;  // 16 wavefronts per CU
    hipMalloc(&out, (size_t)blocks * threads * 4);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    lds_rate<OP><<<blocks, threads>>>(out, 50, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    lds_rate<OP><<<blocks, threads>>>(out, iters, 7);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_cu = (double)iters * 8 * 16;  // wave-instructions per CU
    printf("%-44s %.3f ms -> %.2f ns per wave-instr per CU (%.1f cycles @2.1 GHz)\n", name, ms, ms * 1e6 / inst_per_cu,
           ms * 1e6 / inst_per_cu * 2.1);
    hipFree(out);
}

int main() {
    uint32_t* res;
    hipMalloc(&res, 64 * 8 * 4);
    hipMemset(res, 0, 64 * 8 * 4);
    unaligned_check<<<1, 64>>>(res);
    std::vector<uint32_t> h(64 * 8);
    hipMemcpy(h.data(), res, 64 * 8 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; lane++) {
        const uint32_t* r = &h[lane * 8];
        const uint32_t val = r[6];
        const int sh = 8 * (lane & 3);
        // bytes [4+s, 8+s) of the slot hold val
        const uint64_t mid = ((uint64_t)r[2] << 32) | r[1];
        const uint32_t got = (uint32_t)(mid >> sh);
        const bool ok_w = got == val && r[0] == 0 && (sh == 0 ? r[2] == 0 : (r[2] >> sh) == 0);
        const bool ok_r32 = r[3] == val;
        const uint64_t exp64 = mid >> sh;  // upper bytes come from w3 = 0
        const bool ok_r64 = (((uint64_t)r[5] << 32) | r[4]) == exp64;
        if (!ok_w || !ok_r32 || !ok_r64) {
            if (bad < 8) printf("lane %d: w=%08x %08x %08x r32=%08x r64=%08x%08x val=%08x  write_ok=%d r32_ok=%d r64_ok=%d\n", lane, r[0], r[1], r[2], r[3], r[5], r[4], val, ok_w, ok_r32, ok_r64);
            bad++;
        }
    }
    printf("unaligned LDS access: %s (%d lanes wrong)\n", bad ? "BROKEN" : "ok", bad);
    run<7>("ds_read_b32 conflict-free");
    run<0>("ds_read_b32 random 16 KiB");
    run<9>("ds_read_b32 skewed (50% in 4 hot entries)");
    run<1>("ds_read_b64 random 32 KiB");
    run<10>("ds_read_b64 skewed");
    run<2>("ds_read_u8 random 16 KiB");
    run<11>("ds_read_u8 skewed");
    run<3>("ds_read_u16 random 32 KiB");
    run<4>("ds_write_b32 [word][lane] conflict-free");
    run<6>("ds_write_b32 lane-major 72 B, aligned");
    run<5>("ds_write_b32 lane-major 72 B, UNALIGNED");
    run<8>("ds_write_b64 lane-major 72 B, UNALIGNED");
    return 0;
}
