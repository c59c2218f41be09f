// The process-wide pool of device blocks closed extractors leave for the next one (csrc/vq_tsn.hip): which block of which size on
// which device is held, under a byte cap.  Pure bookkeeping over opaque pointers -- the hipMalloc / hipFree calls stay with the
// caller -- so that it is a host-only translation unit the sanitizer builds can hammer from many threads.
#pragma once
#include <cstddef>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace vq {
class BlockPool {
public:
    explicit BlockPool(size_t cap_bytes) : cap_(cap_bytes) {}
    static size_t cap_from_env();                       // VQ_DEVICE_POOL_GB (default 40, 0 = keep nothing)
    void* take(int device, size_t bytes);               // a held block of exactly this size on this device, or nullptr
    bool give(int device, void* block, size_t bytes);   // true: the pool keeps it; false: over the cap / too small -- the caller frees it
    std::vector<void*> drain();                         // everything held, for the caller to free (out-of-memory elsewhere)
    size_t held() const;

private:
    mutable std::mutex mu_;
    std::multimap<std::pair<int, size_t>, void*> blocks_;
    size_t held_ = 0, cap_ = 0;
};
}  // namespace vq
