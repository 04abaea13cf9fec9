// inflate.hip -- batched zlib decode kernels (gfx950).  One independent stream per wavefront.
#include "inflate_stream.h"

namespace fdh {

struct InflateBatchArgs {
    const uint8_t* in;
    const uint64_t* in_off;
    uint8_t* out;
    const uint64_t* out_off;
    uint32_t* out_len;
    uint32_t* status;
    uint32_t* adler;
    uint64_t n;
    uint32_t flags;
};

// General kernel: workgroup = one wavefront = one stream, private decode tables in LDS.
__global__ __launch_bounds__(kWave) void inflate_general_kernel(InflateBatchArgs a) {
    __shared__ WaveLds lds;
    const int lane = threadIdx.x;
    const uint64_t sid = blockIdx.x;
    if (sid >= a.n) return;
    const uint64_t i0 = a.in_off[sid], i1 = a.in_off[sid + 1];
    const uint64_t o0 = a.out_off[sid], o1 = a.out_off[sid + 1];
    StreamArgs s;
    s.in = a.in + i0;
    s.in_len = i1 - i0;
    s.out = a.out + o0;
    uint64_t capacity = o1 - o0;
    s.cap = capacity > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)capacity;
    s.buf_lo = a.in;
    s.buf_hi = a.in + a.in_off[a.n];
    s.flags = a.flags;
    Inflater inf(lds, lane);
    StreamResult r = inf.run(s);
    if (lane == 0) {
        a.status[sid] = r.status;
        a.out_len[sid] = r.out_len;
        if (a.adler) a.adler[sid] = r.adler;
    }
}

// Debug / parity hook behind fdh_debug_build_tables.
__global__ __launch_bounds__(kWave) void build_tables_debug_kernel(const uint8_t* code_lengths, uint32_t hlit,
                                                                   uint32_t* litlen, uint32_t* dist,
                                                                   uint32_t* build_status) {
    __shared__ WaveLds lds;
    const int lane = threadIdx.x;
    for (int i = lane; i < 320; i += kWave) lds.lens[i] = code_lengths[i];
    for (int i = lane; i < kLitSize; i += kWave) lds.lit[i] = 0xFFFFFFFFu;
    for (int i = lane; i < kDistSize; i += kWave) lds.dist[i] = 0xFFFFFFFFu;
    wave_sync();
    Inflater inf(lds, lane);
    uint32_t st = inf.build_block_tables(hlit);
    wave_sync();
    for (int i = lane; i < kLitSize; i += kWave) litlen[i] = lds.lit[i];
    for (int i = lane; i < kDistSize; i += kWave) dist[i] = lds.dist[i];
    if (lane == 0) {
        build_status[0] = st;
        build_status[1] = inf.eof_code;
        build_status[2] = inf.eof_mask;
        build_status[3] = inf.eof_bits;
    }
}

}  // namespace fdh

extern "C" int fdh_launch_inflate_general(const uint8_t* in, const uint64_t* in_off, uint8_t* out,
                                          const uint64_t* out_off, uint32_t* out_len, uint32_t* status,
                                          uint32_t* adler, uint64_t n, uint32_t flags, hipStream_t stream) {
    fdh::InflateBatchArgs a{in, in_off, out, out_off, out_len, status, adler, n, flags};
    if (n == 0) return 0;
    hipLaunchKernelGGL(fdh::inflate_general_kernel, dim3((unsigned)n), dim3(fdh::kWave), 0, stream, a);
    return (int)hipGetLastError();
}

extern "C" int fdh_launch_build_tables_debug(const uint8_t* code_lengths, uint32_t hlit, uint32_t* litlen,
                                             uint32_t* dist, uint32_t* build_status, hipStream_t stream) {
    hipLaunchKernelGGL(fdh::build_tables_debug_kernel, dim3(1), dim3(fdh::kWave), 0, stream, code_lengths, hlit,
                       litlen, dist, build_status);
    return (int)hipGetLastError();
}
