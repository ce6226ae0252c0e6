// Cost of executing straight-line code the FIRST time in a kernel (instruction cache cold at every dispatch) against the second
// pass over the same code: 2048 / 8192 independent 8-byte VALU instructions per pass.
// hipcc --offload-arch=gfx950 -O3 icache.hip -o icache && ./icache
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__global__ __launch_bounds__(512) void k(double *out, long long *cyc, double a)
{
    double z[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) z[u] = a + u + threadIdx.x;
    long long t[3];
    t[0] = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int i = 0; i < N; ++i) z[i & 7] = __builtin_fma(z[i & 7], 1.0000001, 0.5 + (i >> 3));   // distinct literal-free operands: 8-byte VOP3
        t[pass + 1] = __builtin_amdgcn_s_memtime();
    }
    double s = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += z[u];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) {
        cyc[0] = t[1] - t[0];
        cyc[1] = t[2] - t[1];
    }
}
int main()
{
    double *d_out;
    long long *d_c, h[2];
    (void)hipMalloc(&d_out, 512 * 8);
    (void)hipMalloc(&d_c, 16);
#define RUN(N, NT)                                                                                           \
    for (int rep = 0; rep < 2; ++rep) {                                                                      \
        hipLaunchKernelGGL(k<N>, dim3(1), dim3(NT), 0, 0, d_out, d_c, 1.25);                                 \
        (void)hipMemcpy(h, d_c, 16, hipMemcpyDeviceToHost);                                                  \
        printf("%5d instructions, %3d threads, launch %d: first pass %7lld cycles (%.1f per instruction), second pass %7lld (%.1f)\n", N, NT, rep, h[0], (double)h[0] / N, \
               h[1], (double)h[1] / N);                                                                      \
    }
    RUN(2048, 64) RUN(2048, 512) RUN(8192, 64) RUN(8192, 512)
    return 0;
}
