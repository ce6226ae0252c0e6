// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against known byte counts, in the access patterns the
// extractor kernels use (MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of a wide coalesced streaming read; other widths
// and WRITE_SIZE are uncalibrated).  Every kernel moves exactly N bytes in and N bytes out of a 1 GiB buffer pair
// (4 x the 256 MiB Infinity Cache), so the counters can be divided by N.
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./hbm_counters      (and a second pass with --pmc WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void copy_b128(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)   // 16 B per lane, aligned
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ void copy_b32(const uint32_t *__restrict__ s, uint32_t *__restrict__ d, size_t n)   // 4 B per lane, aligned
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
// 4 B per lane at byte offset 1 (the extractor stages tiles with unaligned 32-bit loads from the caller's image)
__global__ void copy_b32_unaligned(const uint8_t *__restrict__ s, uint32_t *__restrict__ d, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t v;
        __builtin_memcpy(&v, s + 4 * i + 1, 4);
        d[i] = v;
    }
}
__global__ void copy_b8(const uint8_t *__restrict__ s, uint8_t *__restrict__ d, size_t n)   // 1 B per lane
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
// rows of 640 bytes read as 4-byte words by 64-lane waves, one wave per 43-byte-wide cell tile row: short, strided,
// partially overlapping reads like fast_cells_kernel's halo tiles (reads only; writes 1/16 of the volume)
__global__ void read_tiles(const uint8_t *__restrict__ s, uint32_t *__restrict__ d, int w, int h, int images)
{
    const int img = blockIdx.y, cell = blockIdx.x, lane = threadIdx.x & 63;
    const int cx = cell % 20, cy = cell / 20;
    if (cy >= 15) return;
    const uint8_t *p = s + (size_t)img * w * h;
    uint32_t acc = 0;
    for (int r = 0; r < 38; ++r) {
        const int y = cy * 32 + r;
        if (y >= h) break;
        const int x = cx * 32 + 4 * (lane % 10) - 3;
        if (lane < 10 && x >= 0 && x + 4 <= w) {
            uint32_t v;
            __builtin_memcpy(&v, p + (size_t)y * w + x, 4);
            acc += v;
        }
    }
    if (lane < 10) d[((size_t)img * 300 + cell) * 16 + lane] = acc;
}

int main()
{
    const size_t N = (size_t)1 << 30;
    uint8_t *s, *d;
    hipMalloc(&s, N + 64);
    hipMalloc(&d, N + 64);
    hipMemset(s, 1, N + 64);
    hipMemset(d, 0, N + 64);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(copy_b128, dim3(4096), dim3(256), 0, 0, (const uint4 *)s, (uint4 *)d, N / 16);
        hipLaunchKernelGGL(copy_b32, dim3(4096), dim3(256), 0, 0, (const uint32_t *)s, (uint32_t *)d, N / 4);
        hipLaunchKernelGGL(copy_b32_unaligned, dim3(4096), dim3(256), 0, 0, s, (uint32_t *)d, N / 4);
        hipLaunchKernelGGL(copy_b8, dim3(4096), dim3(256), 0, 0, s, d, N);
        hipLaunchKernelGGL(read_tiles, dim3(300, 3495), dim3(64), 0, 0, s, (uint32_t *)d, 640, 480, 3495);   // 3495 x 307200 B = 1.0 GiB
    }
    hipDeviceSynchronize();
    printf("bytes per kernel: copies read %zu + write %zu; read_tiles reads %zu (interior of every image once, halo columns twice)\n", N, N,
           (size_t)3495 * 640 * 480);
    return 0;
}
