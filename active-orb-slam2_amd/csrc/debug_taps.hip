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

// PNG scanline filters undone in place (PNG specification, section 9: None / Sub / Up / Average / Paeth): `rows` = h rows of
// 1 filter byte + stride data bytes as they come out of zlib; bpp = bytes per complete pixel.  Host code for the optional
// real-data loaders (active-orb-slam2_amd/datasets.py): the Average / Paeth recurrences run byte by byte along a row.
int aos2_png_unfilter(uint8_t *rows, int h, int stride, int bpp)
{
    if (!rows || h < 0 || stride <= 0 || bpp <= 0) return AOS2_ERR_ARG;
    const size_t pitch = (size_t)stride + 1;
    for (int y = 0; y < h; ++y) {
        uint8_t *cur = rows + (size_t)y * pitch + 1;
        const uint8_t *up = y ? rows + (size_t)(y - 1) * pitch + 1 : nullptr;
        const int ft = rows[(size_t)y * pitch];
        for (int x = 0; x < stride; ++x) {
            const int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int pred = 0;
            switch (ft) {
            case 0: break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: {
                const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default: return AOS2_ERR_ARG;
            }
            cur[x] = (uint8_t)(cur[x] + pred);
        }
    }
    return AOS2_OK;
}

}  // extern "C"
