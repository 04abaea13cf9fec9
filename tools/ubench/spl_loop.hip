// Stream-per-lane feasibility (round 5): what one lane-serial decode step costs at 1 / 2 wavefronts
// per SIMD, and what the memory system makes of 65 536 concurrent sequential streams.
//   part A  the writing group of inflate_seg2_groups.h turned onto per-lane LDS rings (input ring and
//           output ring of 256 B per lane, rotated by 16 B x (lane % 8)), 4 or 8 wavefronts per CU
//   part B  transposed line traffic: every wavefront owns 64 streams, reads 128-B lines of them at a
//           ~30 KB stride and writes 128-B lines at a 64 KiB stride, eight lanes per line
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/spl tools/ubench/spl_loop.hip && tools/ubench/bin/spl
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); } } while (0)

__device__ __forceinline__ uint32_t lds_off(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
__host__ __device__ inline uint32_t table_entry(uint32_t i) {
    uint32_t h = i * 2654435761u;
    h ^= h >> 15;
    const uint32_t n = 1 + (h >> 28) % 3;
    uint32_t used = 2 * n + ((h >> 20) & 3) * n;
    if (used > 12) used = 12;
    const uint32_t lits = (h >> 4) & (n == 3 ? 0xffffffu : (n == 2 ? 0xffffu : 0xffu));
    return used | (n << 6) | (lits << 8);
}

#define R0 "v120"
#define R1 "v121"
#define WIN "v[120:121]"
#define SHF "v[122:123]"
#define SH0 "v122"
#define VLO "v124"
#define VHI "v125"
#define V64 "v[124:125]"
#define TLO "v126"
#define THI "v127"
#define T64 "v[126:127]"
#define ADD_BYTE0(C, E) "  v_add_u32_sdwa " C ", " C ", " E " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n"

// c[5:0] bit offset in the window, c[31:6] output bytes so far (+ rotation); r: ring offset (bytes, +
// rotation) of the dword that follows hi; ib / ob: LDS addresses of the lane's rings (256-B aligned).
__device__ __forceinline__ uint32_t spl_group(uint32_t pairs, uint32_t& lo, uint32_t& hi, uint32_t& c, uint32_t& r,
                                              uint32_t& acc, uint32_t ib, uint32_t ob, uint32_t mfc) {
    uint32_t e, t, u, nw, wa, ra, sa, sb, x = acc;
    const uint32_t k4 = 4u;
    asm volatile(
        "  s_waitcnt lgkmcnt(0)\n"
        "  v_mov_b32 " R0 ", %[lo]\n"
        "  v_mov_b32 " R1 ", %[hi]\n"
        "  v_mov_b32 " VHI ", 0\n"
        "  v_lshrrev_b32 %[u], 3, %[c]\n"
        "  v_and_b32 %[sa], 24, %[u]\n"
        "  v_lshrrev_b32 %[u], 6, %[c]\n"
        "  v_and_or_b32 %[wa], %[u], %[mfc], %[ob]\n"
        "  v_and_or_b32 %[ra], %[r], %[mfc], %[ib]\n"
        "  ds_read_b32 %[nw], %[ra]\n"
        "Lpair_%=:\n"
        "  v_lshrrev_b64 " SHF ", %[c], " WIN "\n"
        "  v_and_b32 %[t], 0x3ffc, " SH0 "\n"
        "  ds_read_b32 %[e], %[t]\n"
        "  ds_write_b32 %[wa], %[x]\n"
        "  v_lshrrev_b32 %[u], 6, %[c]\n"
        "  v_and_or_b32 %[wa], %[u], %[mfc], %[ob]\n"
        "  s_waitcnt lgkmcnt(1)\n"
        "  v_lshrrev_b32 " VLO ", 8, %[e]\n"
        "  v_lshlrev_b64 " T64 ", %[sa], " V64 "\n"
        "  v_or_b32 %[x], %[acc], " TLO "\n"
        ADD_BYTE0("%[c]", "%[e]")
        "  v_lshrrev_b32 %[u], 3, %[c]\n"
        "  v_and_b32 %[sb], 24, %[u]\n"
        "  v_cmp_lt_u32 vcc, %[sb], %[sa]\n"
        "  v_lshrrev_b64 " SHF ", %[c], " WIN "\n"
        "  v_and_b32 %[t], 0x3ffc, " SH0 "\n"
        "  v_cndmask_b32 %[acc], %[x], " THI ", vcc\n"
        "  ds_read_b32 %[e], %[t]\n"
        "  ds_write_b32 %[wa], %[x]\n"
        "  v_lshrrev_b32 %[u], 6, %[c]\n"
        "  v_and_or_b32 %[wa], %[u], %[mfc], %[ob]\n"
        "  s_waitcnt lgkmcnt(1)\n"
        "  v_lshrrev_b32 " VLO ", 8, %[e]\n"
        "  v_lshlrev_b64 " T64 ", %[sb], " V64 "\n"
        "  v_or_b32 %[x], %[acc], " TLO "\n"
        ADD_BYTE0("%[c]", "%[e]")
        "  v_lshrrev_b32 %[u], 3, %[c]\n"
        "  v_and_b32 %[sa], 24, %[u]\n"
        "  v_cmp_lt_u32 vcc, %[sa], %[sb]\n"
        "  v_and_b32 %[t], 32, %[c]\n"
        "  v_and_b32 %[c], 0xffffffdf, %[c]\n"
        "  v_cndmask_b32 %[acc], %[x], " THI ", vcc\n"
        "  v_cmp_ne_u32 vcc, 0, %[t]\n"
        "  s_sub_u32 %[pairs], %[pairs], 1\n"
        "  s_cmp_lg_u32 %[pairs], 0\n"
        "  v_cndmask_b32 " R0 ", " R0 ", " R1 ", vcc\n"
        "  v_cndmask_b32 " R1 ", " R1 ", %[nw], vcc\n"
        "  v_cndmask_b32 %[t], 0, %[k4], vcc\n"
        "  v_add_u32 %[r], %[r], %[t]\n"
        "  v_and_or_b32 %[ra], %[r], %[mfc], %[ib]\n"
        "  ds_read_b32 %[nw], %[ra]\n"
        "  s_cbranch_scc1 Lpair_%=\n"
        "  ds_write_b32 %[wa], %[x]\n"
        "  s_waitcnt lgkmcnt(0)\n"
        "  v_mov_b32 %[lo], " R0 "\n"
        "  v_mov_b32 %[hi], " R1 "\n"
        : [pairs] "+s"(pairs), [lo] "+v"(lo), [hi] "+v"(hi), [c] "+v"(c), [r] "+v"(r), [acc] "+v"(acc), [x] "+v"(x),
          [e] "=&v"(e), [t] "=&v"(t), [u] "=&v"(u), [nw] "=&v"(nw), [wa] "=&v"(wa), [ra] "=&v"(ra), [sa] "=&v"(sa),
          [sb] "=&v"(sb)
        : [k4] "v"(k4), [mfc] "s"(mfc), [ib] "v"(ib), [ob] "v"(ob)
        : "vcc", "scc", "memory", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    return e;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_loop(uint32_t* out, int groups, uint32_t pairs, uint64_t* cycles) {
    __shared__ uint32_t lit[4096];
    constexpr int RW = WAVES == 4 ? 64 : 32;  // ring words per lane
    __shared__ uint32_t in[WAVES * 64 * RW];
    __shared__ uint32_t outr[WAVES * 64 * RW];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lit[i] = table_entry(i);
    uint32_t x = 0x9E3779B9u * (blockIdx.x + 1) + 77u;
    for (int i = threadIdx.x; i < WAVES * 64 * RW; i += blockDim.x) {
        uint32_t y = x + i * 0x85ebca6bu;
        y ^= y << 13; y ^= y >> 17; y ^= y << 5;
        y ^= y << 13; y ^= y >> 17; y ^= y << 5;
        in[i] = y;
        outr[i] = 0;
    }
    __syncthreads();
    const uint32_t ib = lds_off(in) + threadIdx.x * RW * 4, ob = lds_off(outr) + threadIdx.x * RW * 4;
    const uint32_t rot = 16 * (lane & 7), mfc = RW * 4 - 4;
    uint32_t lo = in[threadIdx.x * RW + ((rot >> 2) & (RW - 1))], hi = in[threadIdx.x * RW + (((rot >> 2) + 1) & (RW - 1))];
    uint32_t r = rot + 8, c = rot << 6, acc = 0, e = 1;
    const long long t0 = clock64();
    for (int g = 0; g < groups; g++) {
        e = spl_group(pairs, lo, hi, c, r, acc, ib, ob, mfc);
        if (__ballot(e == 0)) {  // the periodic look at the lanes that sit on a special token
            c += 2;
        }
    }
    const long long t1 = clock64();
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    out[tid] = lo ^ hi ^ c ^ r ^ acc ^ outr[threadIdx.x * RW + 5];
    if (threadIdx.x == 0) cycles[blockIdx.x] = (uint64_t)(t1 - t0);
}

// part B: 64 streams per wavefront, lines of 128 B, 8 lanes per line.  ratio_q8: output lines per
// input line x 256.  Data passes through registers only (the LDS side is part A's business); a line
// that was loaded is looked at one iteration later, so the loads of an iteration stay in flight
// across its stores.
__global__ __launch_bounds__(256) void k_mem(const uint8_t* in, uint8_t* out, uint32_t in_stride, uint32_t out_lines,
                                              uint32_t ratio_q8, uint32_t* sink) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t sl = lane >> 3, part = lane & 7;
    uint4 accv = {0, 0, 0, 0};
    uint4 prev[8];
    for (int k = 0; k < 8; k++) prev[k] = accv;
    uint32_t in_line = 0, frac = 0;
    for (uint32_t j = 0; j < out_lines; j++) {
        frac += 256;
        const bool rd = frac >= ratio_q8;
        if (rd) frac -= ratio_q8;
        uint4 cur[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint64_t s = (uint64_t)wave * 64 + 8 * k + sl;
            cur[k] = prev[k];
            if (rd) cur[k] = *(const uint4*)(in + s * in_stride + (uint64_t)in_line * 128 + part * 16);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint64_t s = (uint64_t)wave * 64 + 8 * k + sl;
            uint4 w = {j, (uint32_t)s, part, 7u};
            *(uint4*)(out + s * 65536ull + (uint64_t)j * 128 + part * 16) = w;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            accv.x ^= prev[k].x; accv.y += prev[k].y; accv.z ^= prev[k].z; accv.w += prev[k].w;
            prev[k] = cur[k];
        }
        if (rd) in_line++;
    }
    if ((accv.x ^ accv.y) == 0x12345u) sink[0] = accv.z;
}

// per-lane variant: every lane moves its own stream, 16 B per access (no transposition)
__global__ __launch_bounds__(256) void k_mem_lane(const uint8_t* in, uint8_t* out, uint32_t in_stride, uint32_t out_lines,
                                                   uint32_t ratio_q8, uint32_t* sink) {
    const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint4 accv = {0, 0, 0, 0};
    uint4 prev[8];
    for (int k = 0; k < 8; k++) prev[k] = accv;
    uint32_t in_line = 0, frac = 0;
    for (uint32_t j = 0; j < out_lines; j++) {
        frac += 256;
        const bool rd = frac >= ratio_q8;
        if (rd) frac -= ratio_q8;
        uint4 cur[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            cur[k] = prev[k];
            if (rd) cur[k] = *(const uint4*)(in + s * in_stride + (uint64_t)in_line * 128 + k * 16);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint4 w = {j, (uint32_t)s, (uint32_t)k, 7u};
            *(uint4*)(out + s * 65536ull + (uint64_t)j * 128 + k * 16) = w;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            accv.x ^= prev[k].x; accv.y += prev[k].y; accv.z ^= prev[k].z; accv.w += prev[k].w;
            prev[k] = cur[k];
        }
        if (rd) in_line++;
    }
    if ((accv.x ^ accv.y) == 0x12345u) sink[0] = accv.z;
}

int main() {
    uint32_t* out;
    uint64_t* cyc;
    CK(hipMalloc(&out, 256 * 512 * 4));
    CK(hipMalloc(&cyc, 256 * 8));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const uint32_t pairs = 4;
    const int groups = 27400 / 8;
    for (int rep = 0; rep < 2; rep++) {
        for (int waves = 4; waves <= 8; waves += 4) {
            hipEventRecord(e0);
            if (waves == 4) k_loop<4><<<256, 256>>>(out, groups, pairs, cyc);
            else k_loop<8><<<256, 512>>>(out, groups, pairs, cyc);
            hipEventRecord(e1);
            CK(hipDeviceSynchronize());
            CK(hipGetLastError());
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<uint64_t> h(256);
            hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
            double s = 0;
            for (auto v : h) s += (double)v;
            s /= 256;
            printf("A: %d wavefronts/CU: %.3f ms for %d look-ups per lane; %.1f clock64 ticks per look-up (%.2f ns)\n", waves, ms,
                   groups * 8, s / (groups * 8), ms * 1e6 / (groups * 8));
        }
    }
    // part B
    const uint32_t in_stride = 30016, n = 65536;
    uint8_t *din, *dout;
    uint32_t* sink;
    CK(hipMalloc(&din, (size_t)n * in_stride + 4096));
    CK(hipMalloc(&dout, (size_t)n * 65536));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(din, 1, (size_t)n * in_stride + 4096));
    const uint32_t out_lines = 512, ratio_q8 = (uint32_t)(256.0 * 65536 / 30016);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k_mem<<<256, 256>>>(din, dout, in_stride, out_lines, ratio_q8, sink);
        hipEventRecord(e1);
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)n * (65536.0 + 30016.0);
        printf("B transposed: %.3f ms, %.2f TB/s (in + out)\n", ms, bytes / ms / 1e9);
        hipEventRecord(e0);
        k_mem_lane<<<256, 256>>>(din, dout, in_stride, out_lines, ratio_q8, sink);
        hipEventRecord(e1);
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        hipEventElapsedTime(&ms, e0, e1);
        printf("B per lane:   %.3f ms, %.2f TB/s (in + out)\n", ms, bytes / ms / 1e9);
    }
    return 0;
}
