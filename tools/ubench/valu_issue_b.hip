#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void k(uint32_t* out, int iters, uint32_t sc) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = blockIdx.x | 1;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP8(asm volatile("v_and_b32 %0, 0xfff, %0\n v_and_b32 %1, 0xfff, %1\n v_and_b32 %2, 0xfff, %2\n v_and_b32 %3, 0xfff, %3\n v_and_b32 %4, 0xfff, %4\n v_and_b32 %5, 0xfff, %5\n v_and_b32 %6, 0xfff, %6\n v_and_b32 %7, 0xfff, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc));) }
        if (OP == 1) { REP8(asm volatile("v_and_b32 %0, %9, %0\n v_and_b32 %1, %9, %1\n v_and_b32 %2, %9, %2\n v_and_b32 %3, %9, %3\n v_and_b32 %4, %9, %4\n v_and_b32 %5, %9, %5\n v_and_b32 %6, %9, %6\n v_and_b32 %7, %9, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc));) }
        if (OP == 2) { REP8(asm volatile("v_and_b32 %0, 31, %0\n v_and_b32 %1, 31, %1\n v_and_b32 %2, 31, %2\n v_and_b32 %3, 31, %3\n v_and_b32 %4, 31, %4\n v_and_b32 %5, 31, %5\n v_and_b32 %6, 31, %6\n v_and_b32 %7, 31, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc));) }
        if (OP == 3) { REP8(asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc));) }
        if (OP == 4) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc));) }
        if (OP == 5) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n v_lshl_add_u32 %4, %4, 2, %8\n v_lshl_add_u32 %5, %5, 2, %8\n v_lshl_add_u32 %6, %6, 2, %8\n v_lshl_add_u32 %7, %7, 2, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc));) }
        if (OP == 6) { REP8(asm volatile("v_and_or_b32 %0, %0, %8, %8\n v_and_or_b32 %1, %1, %8, %8\n v_and_or_b32 %2, %2, %8, %8\n v_and_or_b32 %3, %3, %8, %8\n v_and_or_b32 %4, %4, %8, %8\n v_and_or_b32 %5, %5, %8, %8\n v_and_or_b32 %6, %6, %8, %8\n v_and_or_b32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc));) }
        if (OP == 7) { REP8(asm volatile("v_lshlrev_b32_e32 %0, 3, %0\n v_lshlrev_b32_e32 %1, 3, %1\n v_lshlrev_b32_e32 %2, 3, %2\n v_lshlrev_b32_e32 %3, 3, %3\n v_lshlrev_b32_e32 %4, 3, %4\n v_lshlrev_b32_e32 %5, 3, %5\n v_lshlrev_b32_e32 %6, 3, %6\n v_lshlrev_b32_e32 %7, 3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc));) }
        if (OP == 8) { REP8(asm volatile("v_lshrrev_b32_e32 %0, %8, %0\n v_lshrrev_b32_e32 %1, %8, %1\n v_lshrrev_b32_e32 %2, %8, %2\n v_lshrrev_b32_e32 %3, %8, %3\n v_lshrrev_b32_e32 %4, %8, %4\n v_lshrrev_b32_e32 %5, %8, %5\n v_lshrrev_b32_e32 %6, %8, %6\n v_lshrrev_b32_e32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc));) }
        if (OP == 9) { REP8(asm volatile("v_mov_b32_e32 %0, %8\n v_mov_b32_e32 %1, %8\n v_mov_b32_e32 %2, %8\n v_mov_b32_e32 %3, %8\n v_mov_b32_e32 %4, %8\n v_mov_b32_e32 %5, %8\n v_mov_b32_e32 %6, %8\n v_mov_b32_e32 %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc));) }
        if (OP == 10) { REP8(asm volatile("v_or_b32_e32 %0, %8, %0\n v_or_b32_e32 %1, %8, %1\n v_or_b32_e32 %2, %8, %2\n v_or_b32_e32 %3, %8, %3\n v_or_b32_e32 %4, %8, %4\n v_or_b32_e32 %5, %8, %5\n v_or_b32_e32 %6, %8, %6\n v_or_b32_e32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc));) }
        if (OP == 11) { REP8(asm volatile("v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n v_addc_co_u32_e32 %2, vcc, 0, %2, vcc\n v_addc_co_u32_e32 %3, vcc, 0, %3, vcc\n v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n v_addc_co_u32_e32 %5, vcc, 0, %5, vcc\n v_addc_co_u32_e32 %6, vcc, 0, %6, vcc\n v_addc_co_u32_e32 %7, vcc, 0, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc) : "vcc");) }
        if (OP == 12) { REP8(asm volatile("v_cmp_lt_u32_e32 vcc, %0, %8\n v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cmp_lt_u32_e32 vcc, %1, %8\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cmp_lt_u32_e32 vcc, %2, %8\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cmp_lt_u32_e32 vcc, %3, %8\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cmp_lt_u32_e32 vcc, %4, %8\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cmp_lt_u32_e32 vcc, %5, %8\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cmp_lt_u32_e32 vcc, %6, %8\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cmp_lt_u32_e32 vcc, %7, %8\n v_cndmask_b32_e32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc) : "vcc");) }
        if (OP == 13) { REP8(asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %8\n v_cmp_lt_u32_e64 s[20:21], %1, %8\n v_cmp_lt_u32_e64 s[20:21], %2, %8\n v_cmp_lt_u32_e64 s[20:21], %3, %8\n v_cmp_lt_u32_e64 s[20:21], %4, %8\n v_cmp_lt_u32_e64 s[20:21], %5, %8\n v_cmp_lt_u32_e64 s[20:21], %6, %8\n v_cmp_lt_u32_e64 s[20:21], %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc) : "s20","s21");) }
        if (OP == 14) { REP8(asm volatile("v_min_u32_e32 %0, %8, %0\n v_min_u32_e32 %1, %8, %1\n v_min_u32_e32 %2, %8, %2\n v_min_u32_e32 %3, %8, %3\n v_min_u32_e32 %4, %8, %4\n v_min_u32_e32 %5, %8, %5\n v_min_u32_e32 %6, %8, %6\n v_min_u32_e32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc));) }
        if (OP == 15) { REP8(asm volatile("v_subrev_u32_e32 %0, 32, %0\n v_subrev_u32_e32 %1, 32, %1\n v_subrev_u32_e32 %2, 32, %2\n v_subrev_u32_e32 %3, 32, %3\n v_subrev_u32_e32 %4, 32, %4\n v_subrev_u32_e32 %5, 32, %5\n v_subrev_u32_e32 %6, 32, %6\n v_subrev_u32_e32 %7, 32, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc));) }
        if (OP == 16) { REP8(asm volatile("v_perm_b32 %0, %0, %8, %8\n v_perm_b32 %1, %1, %8, %8\n v_perm_b32 %2, %2, %8, %8\n v_perm_b32 %3, %3, %8, %8\n v_perm_b32 %4, %4, %8, %8\n v_perm_b32 %5, %5, %8, %8\n v_perm_b32 %6, %6, %8, %8\n v_perm_b32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc));) }
        if (OP == 17) { REP8(asm volatile("v_add3_u32 %0, %0, %8, %8\n v_add3_u32 %1, %1, %8, %8\n v_add3_u32 %2, %2, %8, %8\n v_add3_u32 %3, %3, %8, %8\n v_add3_u32 %4, %4, %8, %8\n v_add3_u32 %5, %5, %8, %8\n v_add3_u32 %6, %6, %8, %8\n v_add3_u32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sc));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int OP>
void run(const char* name, int waves_per_simd, int per_iter) {
    uint32_t* out;
    int blocks = 256 * 4 * waves_per_simd;
    (void)hipMalloc(&out, (size_t)blocks * 64 * 4);
    int iters = 10000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP><<<blocks, 64>>>(out, 100, 0xfff);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<OP><<<blocks, 64>>>(out, iters, 0xfff);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double inst_per_simd = (double)iters * per_iter * waves_per_simd;
    printf("%-24s w/SIMD %d: %.2f cycles/instr/SIMD @2.1GHz\n", name, waves_per_simd, ms * 1e6 / inst_per_simd * 2.1);
    (void)hipFree(out);
}
int main() {
    for (int w : {1, 4}) {
        run<0>("v_and literal 0xfff", w, 64);
        run<1>("v_and sgpr", w, 64);
        run<2>("v_and inline 31", w, 64);
        run<3>("v_cndmask_e32 vcc", w, 64);
        run<4>("v_cndmask_e64 sgpr", w, 64);
        run<5>("v_lshl_add_u32", w, 64);
        run<6>("v_and_or_b32", w, 64);
        run<7>("v_lshlrev_b32 e32", w, 64);
        run<8>("v_lshrrev_b32 vgpr", w, 64);
        run<9>("v_mov_b32", w, 64);
        run<10>("v_or_b32", w, 64);
        run<11>("v_addc_co_e32", w, 64);
        run<12>("v_cmp_e32+cndmask_e32", w, 128);
        run<13>("v_cmp_e64 sgpr", w, 64);
        run<14>("v_min_u32", w, 64);
        run<15>("v_subrev_u32", w, 64);
        run<16>("v_perm_b32", w, 64);
        run<17>("v_add3_u32", w, 64);
    }
    return 0;
}
