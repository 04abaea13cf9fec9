// inflate_seg3.hip -- the landing decoder's kernel (inflate_seg3.h) and its launch.
#include "inflate_seg3.h"

namespace fdh {

#ifdef FDH_S3_DEBUG
// per wavefront: [0] = streams taken, [1 + 2k] = stream id, [2 + 2k] = s_memrealtime (100 MHz) when it was done; [63] = start
__device__ uint32_t g_s3seq[4096 * 64];
#endif

// Landing decoder (inflate_seg3.h) in front of the interval decoder: the same hand-out, the same writing pass,
// its own counting pass; what it does not take is listed for the interval kernel.
__global__ __launch_bounds__(kS2Waves* kWave, 4) void inflate_seg3_kernel(SegArgs a) {
    __shared__ Seg3Lds lds;
#ifdef FDH_S3_DEBUG
    if ((threadIdx.x & 63) == 0 && blockIdx.x * kS2Waves + threadIdx.x / kWave < 4096)
        g_s3seq[(blockIdx.x * kS2Waves + threadIdx.x / kWave) * 64 + 63] = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
    if (lds_offset(lds.lit) != 0) __builtin_trap();
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.canon_lit2);
        const uint4* src2 = reinterpret_cast<const uint4*>(a.canon_lit);
        uint4* dst = reinterpret_cast<uint4*>(lds.lit);
        uint4* dst2 = reinterpret_cast<uint4*>(lds.canon);
        for (int i = threadIdx.x; i < kLitSize / 4; i += kS2Waves * kWave) {
            dst[i] = src[i];
            dst2[i] = src2[i];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    uint2* const ckpt = a.ckpt + (size_t)(blockIdx.x * kS2Waves + threadIdx.x / kWave) * kS2CkptPerWave;
    SegOrder ord{0, 0, 0, 0};
    if (a.order) ord = SegOrder{uni(a.order_counts[0]), uni(a.order_counts[1]), uni(a.order_counts[2]), uni(a.order_counts[3])};
    const uint32_t n32 = a.order ? ord.total() : (uint32_t)a.n;
    // (the first stream of a wavefront is its own number: 4 096 wavefronts after one counter at once is 80 us)
    const uint32_t n_waves = gridDim.x * kS2Waves;
    uint32_t cur = blockIdx.x * kS2Waves + threadIdx.x / kWave, end = cur + 1, take = 1;
    bool took = true;
#ifdef FDH_S3_DEBUG
    const uint32_t wv_ = blockIdx.x * kS2Waves + threadIdx.x / kWave;
    uint32_t nseq_ = 0;
    if (lane == 0 && wv_ < 4096) g_s3seq[wv_ * 64 + 62] = (uint32_t)__builtin_amdgcn_s_memrealtime();  // tables staged
#endif
    for (;;) {
        if (cur == end) {
            take = took ? 1u : min(16u, 2 * take);
            took = false;
            uint32_t next = 0;
            const int leader = __ffsll((unsigned long long)__ballot(true)) - 1;
            if (lane == leader) next = atomicAdd(&a.list[3], take);
            cur = uni(next) + n_waves;
            end = min(n32, cur + take);
        }
        if (cur >= n32) break;
#ifdef FDH_S3_DEBUG
        const uint32_t sid_ = a.order ? uni(a.order[ord.at((uint32_t)a.n, cur)]) : cur;
#endif
        took = seg3_decode(a, lds, ckpt, a.order ? uni(a.order[ord.at((uint32_t)a.n, cur)]) : cur) || took;
#ifdef FDH_S3_DEBUG
        if (lane == 0 && wv_ < 4096 && nseq_ < 30) {
            g_s3seq[wv_ * 64 + 1 + 2 * nseq_] = sid_;
            g_s3seq[wv_ * 64 + 2 + 2 * nseq_] = (uint32_t)__builtin_amdgcn_s_memrealtime();
            nseq_++;
            g_s3seq[wv_ * 64] = nseq_;
        }
#endif
        cur++;
    }
}

}  // namespace fdh

int fdh_launch_seg3(const fdh::SegArgs& sa, unsigned blocks, hipStream_t stream) {
    hipLaunchKernelGGL(fdh::inflate_seg3_kernel, dim3(blocks), dim3(fdh::kS2Waves * fdh::kWave), 0, stream, sa);
    return (int)hipGetLastError();
}

#ifdef FDH_S3_DEBUG
extern "C" int fdh_debug_s3time(uint32_t* host) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(fdh::g_s3time), 4096 * 16 * 4) != hipSuccess) return 2;
    if (hipMemcpyFromSymbol(host + 4096 * 16, HIP_SYMBOL(fdh::g_s3stat), 16 * 4) != hipSuccess) return 3;
#ifdef FDH_S2_DEBUG
    if (hipMemcpyFromSymbol(host + 4096 * 16 + 16, HIP_SYMBOL(fdh::g_s2time), 4096 * 16 * 4) != hipSuccess) return 4;
#endif
    if (hipMemcpyFromSymbol(host + 4096 * 16 + 16 + 4096 * 16, HIP_SYMBOL(fdh::g_s3wtime), 4096 * 8 * 4) != hipSuccess) return 5;
    return 0;
}
extern "C" int fdh_debug_s3base(uint32_t base) {
    return hipMemcpyToSymbol(HIP_SYMBOL(fdh::g_s3base), &base, 4) == hipSuccess ? 0 : 1;
}
extern "C" int fdh_debug_s3seq(uint32_t* host) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(fdh::g_s3seq), 4096 * 64 * 4) != hipSuccess) return 2;
    return 0;
}
#endif
