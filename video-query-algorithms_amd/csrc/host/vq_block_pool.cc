// See vq_block_pool.h.  Host-only translation unit: no HIP.
#include "vq_block_pool.h"

#include <cstdlib>

namespace vq {

size_t BlockPool::cap_from_env() {
    const char* e = getenv("VQ_DEVICE_POOL_GB");
    const double gb = e ? atof(e) : 40.0;
    return gb > 0 ? (size_t)(gb * 1073741824.0) : 0;
}

void* BlockPool::take(int device, size_t bytes) {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = blocks_.find({device, bytes});
    if (it == blocks_.end()) return nullptr;
    void* p = it->second;
    blocks_.erase(it);
    held_ -= bytes;
    return p;
}

bool BlockPool::give(int device, void* block, size_t bytes) {
    std::lock_guard<std::mutex> lk(mu_);
    if (bytes < (1u << 20) || held_ + bytes > cap_) return false;
    blocks_.insert({{device, bytes}, block});
    held_ += bytes;
    return true;
}

std::vector<void*> BlockPool::drain() {
    std::lock_guard<std::mutex> lk(mu_);
    std::vector<void*> out;
    out.reserve(blocks_.size());
    for (auto& kv : blocks_) out.push_back(kv.second);
    blocks_.clear();
    held_ = 0;
    return out;
}

size_t BlockPool::held() const {
    std::lock_guard<std::mutex> lk(mu_);
    return held_;
}

}  // namespace vq
