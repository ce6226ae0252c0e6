// How fast can gfx950 launch one-wave workgroups?  Empty kernel with the grid shapes of describe_kernel /
// fast_cells_kernel (one 64-thread workgroup per keypoint / cell) and their LDS footprints.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_kernel(int *out)
{
    extern __shared__ int lds[];
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && out) {
        lds[0] = 1;
        out[0] = lds[0];
    }
}
int main()
{
    int *d;
    hipMalloc(&d, 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int shapes[][4] = {{1064, 256, 64, 5504}, {1064, 256, 64, 0}, {815, 256, 64, 9216}, {815, 256, 64, 5120},
                             {266, 256, 256, 22016}, {1064, 256, 64, 16384}, {1064 * 256, 1, 64, 5504}, {4256, 256, 64, 0}};
    for (auto &s : shapes) {
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(empty_kernel, dim3(s[0], s[1]), dim3(s[2]), s[3], 0, d);
        hipEventRecord(a);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_kernel, dim3(s[0], s[1]), dim3(s[2]), s[3], 0, d);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("grid %6d x %3d  block %3d  lds %5d B : %.4f ms per launch, %.1f M workgroups/s\n", s[0], s[1], s[2], s[3], ms / 20,
               (double)s[0] * s[1] / (ms / 20) / 1e3);
    }
    return 0;
}
