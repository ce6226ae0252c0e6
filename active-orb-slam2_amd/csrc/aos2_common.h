// Shared host-side helpers of the aos2 HIP library (error handling, device binding).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/aos2.h"

namespace aos2 {

void set_error(const char *fmt, ...);

// Binds `device` for the calling thread; AOS2_ERR_NO_DEVICE if it does not exist.
int bind_device(int device);

// Every stream of the library is created here (csrc/replay.hip): non-blocking, of the highest priority class when asked for.
int stream_create(hipStream_t *q, bool high_priority);
// (measurement switch: AOS2_PRIO_MATCHER / _VOCABULARY / _FRAMES = 1 puts that handle kind's stream into the high priority class)
inline bool stream_priority_env(const char *name)
{
    const char *e = getenv(name);
    return e && e[0] == '1';
}

#define AOS2_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t err__ = (expr);                                                        \
        if (err__ != hipSuccess) {                                                        \
            aos2::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(err__),      \
                            __FILE__, __LINE__);                                          \
            return AOS2_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

// RAII-less tiny device buffer helper (explicit free; handles own their buffers)
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    int alloc(size_t count)
    {
        if (count <= n && p) return AOS2_OK;
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            n = 0;
            set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
            return AOS2_ERR_HIP;
        }
        n = count;
        return AOS2_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

template <typename T>
struct PinnedBuf {
    T *p = nullptr;
    size_t n = 0;
    int alloc(size_t count)
    {
        if (count <= n && p) return AOS2_OK;
        release();
        if (count == 0) count = 1;
        hipError_t e = hipHostMalloc((void **)&p, count * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            n = 0;
            set_error("hipHostMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
            return AOS2_ERR_HIP;
        }
        n = count;
        return AOS2_OK;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = 0;
    }
};

}  // namespace aos2
