// Persistent C-ABI handles for the reference-signature classes of this directory: one matcher / optimiser handle per
// calling thread, device and (nnratio, checkOri) pair, created on first use and kept (a handle owns a stream, device
// arenas and page-locked staging buffers: creating one per call cost more than the call).  The reference constructs
// `ORBmatcher matcher(0.9, true)` on the stack inside every tracking function (src/Tracking.cc:862, 981, 1371 ...), so
// the class is a thin value object and the heavy state lives here.
//
// Device selection: the reference has no notion of a device.  A thread picks the GPU its calls run on with
// aos2::set_thread_device(d) (thread-local, default 0: the Tracking / LocalMapping / LoopClosing threads of one System all
// use the same GPU unless told otherwise); the handle caches are keyed by it, so switching devices mid-thread gets handles
// of the new device instead of silently staying on the first one.
#pragma once
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>

#include "../../include/aos2.h"

namespace aos2 {

// Error convention of the classes in this directory.  The reference's classes have no failure path of their own and the code around
// their call sites has no exception handling; what the reference does when its environment fails it is `cerr << ...; exit(-1)`
// (src/System.cc:78-79, 95-97, 111-112).  A failed C-ABI call here is of that kind (no device, out of device memory, a bad
// argument that the reference's own data cannot produce): by default the message (aos2_last_error) goes to std::cerr and the process
// exits with -1.  Only those: a LocalBundleAdjustment / PoseOptimization call that fails at run time is reported and survived
// (report() below) -- a SLAM process with three running threads is not torn down for one window.  A tree that does handle exceptions compiles with -DAOS2_HOST_EXCEPTIONS and gets std::runtime_error instead
// (tests/cpp/ref_signature_test.cpp does).
[[noreturn]] inline void fail(const char *what)
{
#ifdef AOS2_HOST_EXCEPTIONS
    throw std::runtime_error(std::string(what) + ": " + aos2_last_error());
#else
    std::cerr << what << ": " << aos2_last_error() << std::endl;
    std::exit(-1);
#endif
}

// A per-call failure that the caller can survive (a solve that did not go through): the message, no exit -- the reference's own
// optimiser failures print to cerr and go on.  -DAOS2_HOST_EXCEPTIONS: std::runtime_error like fail().
inline void report(const char *what)
{
#ifdef AOS2_HOST_EXCEPTIONS
    throw std::runtime_error(std::string(what) + ": " + aos2_last_error());
#else
    std::cerr << what << ": " << aos2_last_error() << std::endl;
#endif
}

inline int &thread_device()
{
    thread_local int d = 0;
    return d;
}
inline void set_thread_device(int device) { thread_device() = device; }

inline aos2_matcher_t *matcher_handle(float nnratio, bool checkOri)
{
    struct Cache {
        std::map<std::tuple<int, float, bool>, aos2_matcher_t *> m;
        ~Cache()
        {
            for (auto &kv : m) aos2_matcher_destroy(kv.second);
        }
    };
    thread_local Cache cache;
    const auto key = std::make_tuple(thread_device(), nnratio, checkOri);
    auto it = cache.m.find(key);
    if (it != cache.m.end()) return it->second;
    aos2_matcher_t *h = nullptr;
    if (aos2_matcher_create(nnratio, checkOri ? 1 : 0, thread_device(), &h) != AOS2_OK)
        fail("ORBmatcher");
    cache.m[key] = h;
    return h;
}

inline aos2_lba_t *optimizer_handle()
{
    struct Cache {
        std::map<int, aos2_lba_t *> m;
        ~Cache()
        {
            for (auto &kv : m) aos2_lba_destroy(kv.second);
        }
    };
    thread_local Cache cache;
    auto it = cache.m.find(thread_device());
    if (it != cache.m.end()) return it->second;
    aos2_lba_t *h = nullptr;
    if (aos2_lba_create(thread_device(), &h) != AOS2_OK) fail("Optimizer");
    cache.m[thread_device()] = h;
    return h;
}

// wall-clock split of the last call of this thread through one of the classes: gather (pointer graph -> SoA), the C-ABI
// call, scatter (indices -> pointer graph)
struct ShimTiming {
    double gather_us = 0, call_us = 0, scatter_us = 0;
};
inline ShimTiming &last_shim_timing()
{
    thread_local ShimTiming t;
    return t;
}

struct ShimClock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double lap()
    {
        const auto t1 = std::chrono::steady_clock::now();
        const double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
        t0 = t1;
        return us;
    }
};

// MapPoint::mfMinDistance / mfMaxDistance of an UNCHANGED reference MapPoint.  They are protected (include/MapPoint.h:149-150) and the
// public getters return 0.8f / 1.2f of them (src/MapPoint.cc:413-425) while MapPoint::PredictScale divides the RAW maximum (:432);
// the device applies both itself and dividing the factor out again is not exact in float.  A class derived from MapPoint may name
// its base's protected members, and `&Derived::member` has the type `float MapPoint::*` -- well-formed C++ ([class.protected]
// restricts access through a base OBJECT expression, not forming the member pointer through the derived class): no friend, no new
// accessor, no edit of MapPoint.h.  Read under mMutexPos like the getters.
template <class MP>
struct MapPointDistances : MP {
    static void get(MP *p, float &min_dist, float &max_dist)
    {
        std::unique_lock<std::mutex> lock(p->*(&MapPointDistances::mMutexPos));
        min_dist = p->*(&MapPointDistances::mfMinDistance);
        max_dist = p->*(&MapPointDistances::mfMaxDistance);
    }
};

inline void check(int st, const char *what)
{
    if (st != AOS2_OK) fail(what);
}

}  // namespace aos2
