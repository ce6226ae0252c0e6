// Test taps: host/device execution of shared primitives so parity tests can pin them in isolation.
#include <vector>

#include "aos2_common.h"
#include "octree.h"
#include "sincos_exact.h"

namespace aos2 {
__global__ void sincos_kernel(const float *a, int n, float *s, float *c)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sincos_exact(a[i], &s[i], &c[i]);
}
}  // namespace aos2

extern "C" {

int aos2_debug_octree_host(const int16_t *xs, const int16_t *ys, const uint8_t *score, int n, int minX, int maxX,
                           int minY, int maxY, int N, int32_t *out_idx, int cap)
{
    using namespace aos2;
    if (n <= 0) return 0;
    const int mn = oct_max_nodes(n, N);
    std::vector<OctNode> nodes(mn);
    std::vector<int32_t> perm(n), tmp(n), pairs((size_t)4 * mn);
    OctScratch S{nodes.data(), perm.data(), tmp.data(), pairs.data(), pairs.data() + 2 * mn, mn, mn};
    return distribute_octree(xs, ys, score, n, minX, maxX, minY, maxY, N, S, out_idx, cap);
}

void aos2_debug_sincos_host(float angle_rad, float *s, float *c) { aos2::sincos_exact(angle_rad, s, c); }

int aos2_debug_sincos_device(const float *angles, int n, float *s, float *c, int device)
{
    using namespace aos2;
    int st;
    if ((st = bind_device(device))) return st;
    DevBuf<float> da, ds, dc;
    if ((st = da.alloc(n)) || (st = ds.alloc(n)) || (st = dc.alloc(n))) return st;
    AOS2_HIP_CHECK(hipMemcpy(da.p, angles, sizeof(float) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sincos_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, da.p, n, ds.p, dc.p);
    AOS2_HIP_CHECK(hipDeviceSynchronize());
    AOS2_HIP_CHECK(hipMemcpy(s, ds.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    AOS2_HIP_CHECK(hipMemcpy(c, dc.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    da.release(); ds.release(); dc.release();
    return AOS2_OK;
}

}  // extern "C"
