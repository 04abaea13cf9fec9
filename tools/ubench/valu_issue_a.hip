// VALU issue-rate microbenchmark: N waves per SIMD each running independent instruction chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void k(uint32_t* out, int iters) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = blockIdx.x | 1;
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 1) { REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 2) { REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
        if (OP == 3) { REP8(asm volatile("v_alignbit_b32 %0, %0, %8, %8\n v_alignbit_b32 %1, %1, %8, %8\n v_alignbit_b32 %2, %2, %8, %8\n v_alignbit_b32 %3, %3, %8, %8\n v_alignbit_b32 %4, %4, %8, %8\n v_alignbit_b32 %5, %5, %8, %8\n v_alignbit_b32 %6, %6, %8, %8\n v_alignbit_b32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 4) { REP8(asm volatile("v_bfe_u32 %0, %0, 2, 5\n v_bfe_u32 %1, %1, 2, 5\n v_bfe_u32 %2, %2, 2, 5\n v_bfe_u32 %3, %3, 2, 5\n v_bfe_u32 %4, %4, 2, 5\n v_bfe_u32 %5, %5, 2, 5\n v_bfe_u32 %6, %6, 2, 5\n v_bfe_u32 %7, %7, 2, 5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 5) { REP8(asm volatile("v_add_u32_sdwa %0, %8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %8, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %3, %8, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %4, %8, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %5, %8, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %6, %8, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %7, %8, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 6) { REP8(asm volatile("v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3\n v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(b));) }
        if (OP == 7) { REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
        if (OP == 8) { REP8(asm volatile("s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[24:25], s[24:25], s[22:23]\n s_and_b64 s[26:27], s[26:27], s[22:23]\n s_and_b64 s[28:29], s[28:29], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[24:25], s[24:25], s[22:23]\n s_and_b64 s[26:27], s[26:27], s[22:23]\n s_and_b64 s[28:29], s[28:29], s[22:23]" ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "scc");) }
        if (OP == 9) { REP8(asm volatile("v_add_u32 %0, %0, %8\n s_and_b64 s[20:21], s[20:21], s[22:23]\n v_add_u32 %1, %1, %8\n s_and_b64 s[24:25], s[24:25], s[22:23]\n v_add_u32 %2, %2, %8\n s_and_b64 s[26:27], s[26:27], s[22:23]\n v_add_u32 %3, %3, %8\n s_and_b64 s[28:29], s[28:29], s[22:23]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "scc");) }
        if (OP == 10) { REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)(q0 + q1 + q2 + q3);
}
template <int OP>
void run(const char* name, int waves_per_simd, int per_iter) {
    uint32_t* out;
    int blocks = 256 * 4 * waves_per_simd;  // one wave per block: spreads over all SIMDs
    hipMalloc(&out, (size_t)blocks * 64 * 4);
    int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 64>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 64>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst_per_simd = (double)iters * per_iter * waves_per_simd;
    printf("%-22s waves/SIMD %d: %.3f ms  -> %.2f ns per instr per SIMD (%.2f cycles @2.1GHz)\n", name, waves_per_simd, ms, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.1);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_add_u32", w, 64);
        run<1>("v_and_b32", w, 64);
        run<2>("v_cndmask", w, 64);
        run<3>("v_alignbit", w, 64);
        run<4>("v_bfe_u32", w, 64);
        run<5>("v_add_u32_sdwa", w, 64);
        run<6>("v_lshlrev_b64", w, 64);
        run<7>("v_cmp_lt_u32", w, 64);
        run<8>("s_and_b64", w, 64);
        run<9>("v_add+s_and mix", w, 64);
        run<10>("v_add dependent", w, 64);
    }
    return 0;
}
