// Single-wave latencies that bound the serial pivot chain of the reduced-system LDL^T (csrc/lba.hip: k_ldlt_lds):
// dependent v_fma_f64, IEEE f64 division, v_rcp_f64 + 2 Newton steps, v_readlane -> VALU use, LDS write -> broadcast read.
// hipcc --offload-arch=gfx950 -O3 f64_latency.hip -o f64_latency && ./f64_latency
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double readlane_f64(double v, int src)
{
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int MODE>
__global__ void k(double *out, long long *cyc, double a, double b)
{
    __shared__ double sh[64];
    double x = a + threadIdx.x * 1e-9, y = b;
    sh[threadIdx.x] = x;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
        if (MODE == 0) {   // 8 dependent FMAs
#pragma unroll
            for (int u = 0; u < 8; ++u) x = __builtin_fma(x, y, y);
        } else if (MODE == 1) {   // 8 dependent IEEE divisions
#pragma unroll
            for (int u = 0; u < 8; ++u) x = y / x + 1.0;
        } else if (MODE == 2) {   // 8 dependent rcp + 2 NR
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                double r = __builtin_amdgcn_rcp(x);
                double e = __builtin_fma(-x, r, 1.0);
                r = __builtin_fma(r, e, r);
                e = __builtin_fma(-x, r, 1.0);
                r = __builtin_fma(r, e, r);
                x = r + 1.0;
            }
        } else if (MODE == 3) {   // 8 dependent readlane -> fma
#pragma unroll
            for (int u = 0; u < 8; ++u) x = __builtin_fma(readlane_f64(x, u), y, y);
        } else if (MODE == 4) {   // 8 dependent LDS write -> broadcast read -> fma
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                sh[threadIdx.x] = x;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                x = __builtin_fma(sh[u], y, y);
            }
        } else if (MODE == 5) {   // 8 independent FMAs (issue rate)
            double z[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) z[u] = __builtin_fma(x, y + u, y);
#pragma unroll
            for (int u = 1; u < 8; ++u) z[0] += z[u];
            x = z[0];
        } else if (MODE == 6) {   // 16 independent readlanes of one value, then one fma
            double s = 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) s += readlane_f64(x, u);
            x = s * y;
        } else if (MODE == 7) {   // dependent f32 rcp + f64 refinement (3 NR)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                double r = (double)__builtin_amdgcn_rcpf((float)x);
                double e = __builtin_fma(-x, r, 1.0);
                r = __builtin_fma(r, e, r);
                e = __builtin_fma(-x, r, 1.0);
                r = __builtin_fma(r, e, r);
                e = __builtin_fma(-x, r, 1.0);
                r = __builtin_fma(r, e, r);
                x = r + 1.0;
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    double *d_out;
    long long *d_c, h;
    hipMalloc(&d_out, 64 * 8);
    hipMalloc(&d_c, 8);
    const char *names[] = {"dependent v_fma_f64", "dependent IEEE f64 division (+add)", "dependent v_rcp_f64 + 2 Newton (+add)",
                           "dependent readlane_f64 -> fma", "dependent LDS write -> broadcast read -> fma",
                           "8 independent fma + 7 adds (per group)", "16 readlane_f64 + 16 adds + mul (per group)",
                           "dependent v_rcp_f32 + 3 Newton in f64 (+add)"};
    const int per[] = {8, 8, 8, 8, 8, 1, 1, 8};
#define RUN(M)                                                                  \
    hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, d_out, d_c, 1.25, 0.75);  \
    hipMemcpy(&h, d_c, 8, hipMemcpyDeviceToHost);                               \
    printf("%-52s %8.1f cycles\n", names[M], (double)h / (256.0 * per[M]));
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    return 0;
}
