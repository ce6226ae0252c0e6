// Latencies that bound the register-resident reduced-system solve (csrc/ldlt_reg.h): v_mfma_f64_16x16x4 dependent / independent,
// LDS read -> use, workgroup barrier with 8 waves, f64 FMA issue.
// hipcc --offload-arch=gfx950 -O3 lr_latency.hip -o lr_latency && ./lr_latency
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(double *out, long long *cyc, double a, double b, int stride)
{
    __shared__ double sh[4096];
    __shared__ int idx[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += blockDim.x) {
        sh[i] = a + i * 1e-9;
        idx[i] = (i * 17 + 5) & 4095;
    }
    __syncthreads();
    d4 c0 = {a, b, a, b}, c1 = c0, c2 = c0, c3 = c0;
    double x = a + tid * 1e-9, y = b;
    int p = tid & 63;
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
        if (MODE == 0) {   // 8 dependent MFMAs
#pragma unroll
            for (int u = 0; u < 8; ++u) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c0, 0, 0, 0);
        } else if (MODE == 1) {   // 8 MFMAs on 4 accumulators
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c3, 0, 0, 0);
            }
        } else if (MODE == 2) {   // 8 dependent LDS reads (pointer chase)
#pragma unroll
            for (int u = 0; u < 8; ++u) p = idx[p];
        } else if (MODE == 3) {   // 8 barriers
#pragma unroll
            for (int u = 0; u < 8; ++u) __syncthreads();
        } else if (MODE == 4) {   // 16 independent FMAs
            double z[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) z[u] = __builtin_fma(x, y + u, y);
#pragma unroll
            for (int u = 0; u < 16; ++u) z[u] = __builtin_fma(z[u], y, x);
            double s = 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) s += z[u];
            x = s;
        } else if (MODE == 5) {   // LDS write -> barrier -> read by another wave -> barrier (8 waves)
            sh[tid] = x;
            __syncthreads();
            x = sh[(tid + 64) & 511] + y;
            __syncthreads();
        } else if (MODE == 6) {   // MFMA result -> VALU use -> MFMA operand (dependent through the A operand)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c1, 0, 0, 0);
                x = c0[0] * 0.5;
            }
        }
    }
    if (MODE != 3) __syncthreads();   // the LAST wave's finish: the oldest wave is served first
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[tid] = x + c0[0] + c1[1] + c2[2] + c3[3] + p;
    if (tid == 0) cyc[0] = t1 - t0;
}

int main()
{
    double *d_out;
    long long *d_c, h;
    hipMalloc(&d_out, 1024 * 8);
    hipMalloc(&d_c, 8);
    const char *names[] = {"dependent v_mfma_f64_16x16x4", "v_mfma_f64_16x16x4 on 4 accumulators", "dependent LDS read (b32 pointer chase)", "workgroup barrier",
                           "32 independent FMA + 16 add (per group)", "LDS write, barrier, read, barrier", "MFMA -> VALU -> MFMA operand"};
    const int per[] = {8, 8, 8, 8, 1, 1, 8};
#define RUN(M, NT)                                                                  \
    hipLaunchKernelGGL(k<M>, dim3(1), dim3(NT), 0, 0, d_out, d_c, 1.25, 0.75, 17);  \
    hipMemcpy(&h, d_c, 8, hipMemcpyDeviceToHost);                                   \
    printf("%-48s %4d threads %8.1f cycles\n", names[M], NT, (double)h / (256.0 * per[M]));
    RUN(0, 64) RUN(0, 512) RUN(0, 768) RUN(0, 1024) RUN(1, 64) RUN(1, 256) RUN(1, 512) RUN(1, 1024) RUN(2, 64) RUN(2, 512) RUN(3, 512) RUN(3, 64) RUN(4, 64) RUN(4, 512) RUN(5, 512) RUN(6, 64)
    return 0;
}
