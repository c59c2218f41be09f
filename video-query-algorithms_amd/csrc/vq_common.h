// Shared host-side helpers of libvqamd (error plumbing, HIP call checking).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "vq_amd.h"
#include "host/vq_host.h"      // cdiv, the error string shared with the host-only translation units

namespace vq {

inline int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

#define VQ_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return ::vq::fail(e_ == hipErrorOutOfMemory ? VQ_E_NOMEM : VQ_E_HIP, "%s failed: %s (%s:%d)", #call, \
                              hipGetErrorString(e_), __FILE__, __LINE__);                         \
    } while (0)

#define VQ_CHECK_LAUNCH()                                                                         \
    do {                                                                                          \
        hipError_t e_ = hipGetLastError();                                                        \
        if (e_ != hipSuccess)                                                                     \
            return ::vq::fail(VQ_E_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define VQ_REQUIRE(cond, ...)                                  \
    do {                                                       \
        if (!(cond)) return ::vq::fail(VQ_E_INVALID, __VA_ARGS__); \
    } while (0)

// RAII: make `device` current for the duration of an API call.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) ok = (hipSetDevice(device) == hipSuccess);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the (kernel, DEVICE) pair, and one process may hold handles on several
// GPUs (vq_db_create / vq_tsn_create take a device): remember per device where it has been raised.  Use inside a function that
// returns int, with the handle's device current.
struct PerDeviceOnce {
    std::mutex mu;
    unsigned long long done = 0;          // bit d: set on device d
    bool need(int device) {
        std::lock_guard<std::mutex> lk(mu);
        return device < 0 || device >= 64 || !((done >> device) & 1ull);
    }
    void mark(int device) {
        std::lock_guard<std::mutex> lk(mu);
        if (device >= 0 && device < 64) done |= 1ull << device;
    }
};
#define VQ_DYN_LDS(kern, bytes)                                                                                                  \
    do {                                                                                                                         \
        static ::vq::PerDeviceOnce once_;                                                                                        \
        int dev_ = -1;                                                                                                           \
        (void)hipGetDevice(&dev_);                                                                                               \
        if (once_.need(dev_)) {                                                                                                  \
            VQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
            once_.mark(dev_);                                                                                                    \
        }                                                                                                                        \
    } while (0)


// csrc/vq_boot.hip: closed-form target bootstrapping on rows that already live on the device.  row_off [P][stride]
// element offsets into base_dev (the n_valid[p] validated matches first, then the n_invalid[p] non-matches).
int bootstrap_from_device_rows(const void* base_dev, int dtype, const std::vector<int64_t>& row_off, int stride, const int32_t* n_valid,
                               const int32_t* n_invalid, int P, int D, double mu, hipStream_t stream, double* targets_host,
                               double* targets_dev_copy);

// csrc/vq_sim.hip: copy the n scores of the last scan to dst_dev on `st`, ORDERED BEHIND that scan (which ran on the handle's
// own stream): an event recorded on the handle's stream, waited for by `st`.  VQ_E_STATE without a scan.  Takes the handle's lock.
int db_copy_scores_ordered(vq_db* db, double* dst_dev, int64_t* n_out, hipStream_t st);

// csrc/vq_tsn.hip: closed extractors leave their big device blocks in a process-wide pool for the next extractor of the same shape
// (VQ_DEVICE_POOL_GB); this gives them back to the driver -- called by the other modules when an allocation of theirs runs out of memory.
void device_pool_trim();

// hipMalloc for everything the library allocates itself: when the device is out of memory the blocks that closed extractors left in
// the pool go back to the driver and the allocation is tried once more (the pool must never be the reason another handle cannot grow).
// Allocations made OUTSIDE the library (torch, the host application) cannot see the pool: VQ_DEVICE_POOL_GB bounds what it may hold.
inline hipError_t malloc_trim(void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        device_pool_trim();
        e = hipMalloc(p, bytes);
    }
    return e;
}
template <typename T>
inline hipError_t malloc_trim(T** p, size_t bytes) {
    return malloc_trim(reinterpret_cast<void**>(p), bytes);
}

}  // namespace vq
