// multi_gpu.cpp -- fdh_init / fdh_shutdown / fdh_inflate_batch_multi: one process driving several
// GPUs of a node.  Streams are independent, so a batch is sharded by contiguous
// stream ranges with no data-path exchange: every device decodes its own shard on its own HIP
// stream.  The only collective is the all-gather of the fixed-size per-stream results
// {status, out_len, adler} (12 B per stream) over RCCL (xGMI), so that every device -- and through
// it the caller -- sees the status of the whole batch.  RCCL is loaded lazily (dlopen) and only
// when more than one device takes part; with one device the "gather" is a device-to-device copy
// (FDH_MULTI_FORCE_RCCL=1 makes a single device go through RCCL as well: a one-rank communicator,
// the same ncclGroupStart / ncclAllGather / ncclGroupEnd calls -- the way to exercise that path on
// a one-GPU box).
//
// (The Python layer has the one-process-per-GPU form of the same thing: fdeflate_amd/distributed.py
// over torch.distributed, which is what bench.py uses.)
#include "../../include/fdeflate_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

extern "C" void fdh_set_last_error(const char* msg);

namespace {

// the handful of RCCL entry points used (signatures of rccl.h; ncclUint32 = 3, ncclSuccess = 0)
typedef void* ncclComm_t;
typedef int (*p_ncclCommInitAll)(ncclComm_t*, int, const int*);
typedef int (*p_ncclCommDestroy)(ncclComm_t);
typedef int (*p_ncclAllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t);
typedef int (*p_ncclGroup)(void);
typedef const char* (*p_ncclGetErrorString)(int);
constexpr int kNcclUint32 = 3;

struct Multi {
    std::vector<int> devices;
    std::vector<hipStream_t> streams;
    std::vector<uint32_t*> meta_send;  // per device: 3 x cap words
    size_t cap = 0;                    // streams per shard the staging buffers hold
    void* rccl = nullptr;
    std::vector<ncclComm_t> comms;
    p_ncclCommInitAll CommInitAll = nullptr;
    p_ncclCommDestroy CommDestroy = nullptr;
    p_ncclAllGather AllGather = nullptr;
    p_ncclGroup GroupStart = nullptr, GroupEnd = nullptr;
    p_ncclGetErrorString GetErrorString = nullptr;
    bool use_rccl = false;  // more than one device, or forced
    bool ready = false;
};
Multi g_multi;
std::mutex g_multi_mutex;

int fail(int code, const std::string& msg) {
    fdh_set_last_error(msg.c_str());
    return code;
}
#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return fail(e_ == hipErrorOutOfMemory ? FDH_ERR_OUT_OF_MEMORY : FDH_ERR_HIP, \
                        std::string(#expr) + ": " + hipGetErrorString(e_));              \
    } while (0)

void release_locked() {
    Multi& m = g_multi;
    for (size_t i = 0; i < m.devices.size(); i++) {
        (void)hipSetDevice(m.devices[i]);
        if (i < m.comms.size() && m.comms[i] && m.CommDestroy) (void)m.CommDestroy(m.comms[i]);
        if (i < m.meta_send.size() && m.meta_send[i]) (void)hipFree(m.meta_send[i]);
        if (i < m.streams.size() && m.streams[i]) (void)hipStreamDestroy(m.streams[i]);
    }
    if (m.rccl) dlclose(m.rccl);
    m = Multi();
}

}  // namespace

extern "C" {

int fdh_init(uint64_t device_mask) {
    std::lock_guard<std::mutex> lock(g_multi_mutex);
    // whatever an earlier call left behind -- a complete state or the debris of a failed attempt --
    // goes first; and every error path below releases what this call has built
    release_locked();
    struct Guard {
        bool armed = true;
        ~Guard() {
            if (armed) release_locked();
        }
    } guard;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(FDH_ERR_NO_DEVICE, "no HIP device: fdeflate_hip has no CPU fallback");
    Multi& m = g_multi;
    for (int d = 0; d < ndev && d < 64; d++) {
        if (device_mask == 0 || ((device_mask >> d) & 1)) m.devices.push_back(d);
    }
    if (m.devices.empty()) return fail(FDH_ERR_INVALID_ARGUMENT, "device_mask selects no visible device");
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    m.streams.assign(m.devices.size(), nullptr);
    m.meta_send.assign(m.devices.size(), nullptr);
    for (size_t i = 0; i < m.devices.size(); i++) {
        HIP_TRY(hipSetDevice(m.devices[i]));
        HIP_TRY(hipStreamCreateWithFlags(&m.streams[i], hipStreamNonBlocking));
    }
    const char* force = std::getenv("FDH_MULTI_FORCE_RCCL");
    m.use_rccl = m.devices.size() > 1 || (force && force[0] == '1');
    if (m.use_rccl) {
        m.rccl = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!m.rccl) m.rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!m.rccl) return fail(FDH_ERR_HIP, "librccl.so could not be loaded (needed for more than one device)");
        m.CommInitAll = (p_ncclCommInitAll)dlsym(m.rccl, "ncclCommInitAll");
        m.CommDestroy = (p_ncclCommDestroy)dlsym(m.rccl, "ncclCommDestroy");
        m.AllGather = (p_ncclAllGather)dlsym(m.rccl, "ncclAllGather");
        m.GroupStart = (p_ncclGroup)dlsym(m.rccl, "ncclGroupStart");
        m.GroupEnd = (p_ncclGroup)dlsym(m.rccl, "ncclGroupEnd");
        m.GetErrorString = (p_ncclGetErrorString)dlsym(m.rccl, "ncclGetErrorString");
        if (!m.CommInitAll || !m.CommDestroy || !m.AllGather || !m.GroupStart || !m.GroupEnd)
            return fail(FDH_ERR_HIP, "librccl.so lacks an expected entry point");
        m.comms.assign(m.devices.size(), nullptr);
        int rc = m.CommInitAll(m.comms.data(), (int)m.devices.size(), m.devices.data());
        if (rc != 0) return fail(FDH_ERR_HIP, std::string("ncclCommInitAll: ") + (m.GetErrorString ? m.GetErrorString(rc) : "error"));
    }
    (void)hipSetDevice(prev);
    m.ready = true;
    guard.armed = false;
    return FDH_SUCCESS;
}

int fdh_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_multi_mutex);
    int prev = 0;
    (void)hipGetDevice(&prev);
    release_locked();
    (void)hipSetDevice(prev);
    return FDH_SUCCESS;
}

int fdh_multi_device_count(void) {
    std::lock_guard<std::mutex> lock(g_multi_mutex);
    return g_multi.ready ? (int)g_multi.devices.size() : 0;
}

int fdh_multi_uses_rccl(void) {
    std::lock_guard<std::mutex> lock(g_multi_mutex);
    return (g_multi.ready && g_multi.use_rccl) ? 1 : 0;
}

int fdh_inflate_batch_multi(const fdh_shard_t* shards, uint32_t n_shards, uint32_t flags, uint64_t meta_stride) {
    std::lock_guard<std::mutex> lock(g_multi_mutex);
    Multi& m = g_multi;
    if (!m.ready) return fail(FDH_ERR_INVALID_ARGUMENT, "fdh_init has not been called");
    if (!shards || n_shards != m.devices.size())
        return fail(FDH_ERR_INVALID_ARGUMENT, "one shard per initialised device is required");
    uint64_t n_max = 0;
    for (uint32_t i = 0; i < n_shards; i++) n_max = shards[i].n > n_max ? shards[i].n : n_max;
    bool gather = false;
    for (uint32_t i = 0; i < n_shards; i++) gather = gather || shards[i].meta_all != nullptr;
    if (gather) {
        if (meta_stride < n_max) return fail(FDH_ERR_INVALID_ARGUMENT, "meta_stride is smaller than the largest shard");
        for (uint32_t i = 0; i < n_shards; i++) {
            if (!shards[i].meta_all) return fail(FDH_ERR_INVALID_ARGUMENT, "meta_all must be given for every shard or for none");
        }
    }
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    // on every way out: the shards already launched are waited for (an error must not return while
    // other devices still write the caller's buffers), then the caller's device is restored
    struct Back {
        int dev;
        Multi* m;
        uint32_t launched = 0;
        bool done = false;
        ~Back() {
            if (!done) {
                for (uint32_t i = 0; i < launched; i++) {
                    (void)hipSetDevice(m->devices[i]);
                    (void)hipStreamSynchronize(m->streams[i]);
                }
            }
            (void)hipSetDevice(dev);
        }
    } back{prev, &m};
    // staging for the gather: {status, out_len, adler} of the shard, padded to meta_stride
    if (gather && m.cap < meta_stride) {
        m.cap = 0;  // (a failed allocation below must not leave a stale capacity behind)
        for (size_t i = 0; i < m.devices.size(); i++) {
            HIP_TRY(hipSetDevice(m.devices[i]));
            if (m.meta_send[i]) (void)hipFree(m.meta_send[i]);
            m.meta_send[i] = nullptr;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m.meta_send[i]), 3 * meta_stride * sizeof(uint32_t)));
        }
        m.cap = meta_stride;
    }
    // 1. every device decodes its shard on its own stream: no communication
    for (uint32_t i = 0; i < n_shards; i++) {
        const fdh_shard_t& s = shards[i];
        HIP_TRY(hipSetDevice(m.devices[i]));
        back.launched = i + 1;
        int rc = fdh_inflate_batch(s.in, s.in_off, s.out, s.out_off, s.out_len, s.status, s.adler, s.n, flags, m.streams[i]);
        if (rc != FDH_SUCCESS) return rc;
        if (gather) {
            HIP_TRY(hipMemsetAsync(m.meta_send[i], 0, 3 * meta_stride * sizeof(uint32_t), m.streams[i]));
            if (s.n) {
                HIP_TRY(hipMemcpyAsync(m.meta_send[i], s.status, s.n * 4, hipMemcpyDeviceToDevice, m.streams[i]));
                HIP_TRY(hipMemcpyAsync(m.meta_send[i] + meta_stride, s.out_len, s.n * 4, hipMemcpyDeviceToDevice, m.streams[i]));
                if (s.adler)
                    HIP_TRY(hipMemcpyAsync(m.meta_send[i] + 2 * meta_stride, s.adler, s.n * 4, hipMemcpyDeviceToDevice, m.streams[i]));
            }
        }
    }
    // 2. the one collective: all-gather of the per-stream results (12 B per stream) over RCCL
    if (gather) {
        const size_t count = 3 * meta_stride;
        if (!m.use_rccl) {
            HIP_TRY(hipSetDevice(m.devices[0]));
            HIP_TRY(hipMemcpyAsync(shards[0].meta_all, m.meta_send[0], count * 4, hipMemcpyDeviceToDevice, m.streams[0]));
        } else {
            int rc = m.GroupStart();
            for (uint32_t i = 0; i < n_shards && rc == 0; i++) {
                (void)hipSetDevice(m.devices[i]);
                rc = m.AllGather(m.meta_send[i], shards[i].meta_all, count, kNcclUint32, m.comms[i], m.streams[i]);
            }
            const int rc2 = m.GroupEnd();
            if (rc != 0 || rc2 != 0)
                return fail(FDH_ERR_HIP, std::string("ncclAllGather: ") + (m.GetErrorString ? m.GetErrorString(rc ? rc : rc2) : "error"));
        }
    }
    // 3. the call returns when every device is done
    for (uint32_t i = 0; i < n_shards; i++) {
        HIP_TRY(hipSetDevice(m.devices[i]));
        HIP_TRY(hipStreamSynchronize(m.streams[i]));
    }
    back.done = true;
    return FDH_SUCCESS;
}

}  // extern "C"
