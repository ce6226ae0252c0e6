// Persistent C-ABI handles for the reference-signature shims: one matcher / optimiser handle per calling thread and
// (nnratio, checkOri) pair, created on first use and kept (a handle owns a stream, device arenas and page-locked
// staging buffers: creating one per call cost more than the call).  The reference constructs `ORBmatcher matcher(0.9,
// true)` on the stack inside every tracking function (src/Tracking.cc:862, 981, 1371 ...), so the shim class is a thin
// value object and the heavy state lives here.
#pragma once
#include <map>
#include <stdexcept>
#include <string>
#include <utility>

#include "../../../include/aos2.h"

namespace aos2 {

inline int &default_device()
{
    static int d = 0;
    return d;
}

inline aos2_matcher_t *matcher_handle(float nnratio, bool checkOri)
{
    struct Cache {
        std::map<std::pair<float, bool>, aos2_matcher_t *> m;
        ~Cache()
        {
            for (auto &kv : m) aos2_matcher_destroy(kv.second);
        }
    };
    thread_local Cache cache;
    auto key = std::make_pair(nnratio, checkOri);
    auto it = cache.m.find(key);
    if (it != cache.m.end()) return it->second;
    aos2_matcher_t *h = nullptr;
    if (aos2_matcher_create(nnratio, checkOri ? 1 : 0, default_device(), &h) != AOS2_OK)
        throw std::runtime_error(std::string("ORBmatcher: ") + aos2_last_error());
    cache.m[key] = h;
    return h;
}

inline aos2_lba_t *optimizer_handle()
{
    struct Cache {
        aos2_lba_t *h = nullptr;
        ~Cache() { aos2_lba_destroy(h); }
    };
    thread_local Cache cache;
    if (!cache.h && aos2_lba_create(default_device(), &cache.h) != AOS2_OK)
        throw std::runtime_error(std::string("Optimizer: ") + aos2_last_error());
    return cache.h;
}

// wall-clock split of the last shim call of this thread: gather (pointer graph -> SoA), the C-ABI call, scatter
struct ShimTiming {
    double gather_us = 0, call_us = 0, scatter_us = 0;
};
inline ShimTiming &last_shim_timing()
{
    thread_local ShimTiming t;
    return t;
}

}  // namespace aos2
