// The register-resident reduced-system solve (csrc/ldlt_reg.h) alone: random symmetric positive definite systems of the sizes
// LocalBA meets (6 x free keyframes) against a host LDL^T in double, phase cycle counters, time per launch of NM workgroups.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I active-orb-slam2_amd/csrc tools/microbench/ldlt_reg_bench.hip -o tools/microbench/ldlt_reg_bench
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef AOS2_LR_TRACE
#define AOS2_LR_TRACE 0
#endif
#include "ldlt_reg.h"
constexpr int kDbg = 16 + AOS2_LR_TRACE * (aos2::kLrWorkers + 1);

using namespace aos2;

template <bool kTiming>
__global__ __launch_bounds__(kLrThreads) void k_test(const double *Hs, int ld, int n, int npad, const double *bs, double *x, long long *dbg, int *ok)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const size_t m = blockIdx.x;
    double *xs = nullptr;
    const bool good = ldlt_reg_solve<kTiming>(Hs + m * (size_t)ld * ld, n, npad, bs + m * (size_t)npad, sm, xs, dbg + (size_t)kDbg * m);
    if (threadIdx.x == 0) ok[m] = good ? 1 : 0;
    if (good)
        for (int i = threadIdx.x; i < n; i += kLrThreads) x[m * (size_t)npad + i] = xs[i];
}

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

static bool host_ldlt(std::vector<double> A, int n, std::vector<double> b, std::vector<double> &x)
{
    std::vector<double> d(n);
    for (int j = 0; j < n; ++j) {
        double dj = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) dj -= A[(size_t)j * n + k] * A[(size_t)j * n + k] * d[k];
        if (dj == 0.0 || dj != dj) return false;
        d[j] = dj;
        for (int i = j + 1; i < n; ++i) {
            double v = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) v -= A[(size_t)i * n + k] * A[(size_t)j * n + k] * d[k];
            A[(size_t)i * n + j] = v / dj;
        }
    }
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < i; ++k) b[i] -= A[(size_t)i * n + k] * b[k];
    for (int i = 0; i < n; ++i) b[i] /= d[i];
    for (int i = n - 1; i >= 0; --i)
        for (int k = i + 1; k < n; ++k) b[i] -= A[(size_t)k * n + i] * b[k];
    x = b;
    return true;
}

int main(int argc, char **argv)
{
    const int NM = argc > 1 ? atoi(argv[1]) : 38;
    CK(hipFuncSetAttribute((const void *)k_test<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLrMaxDynLds));
    CK(hipFuncSetAttribute((const void *)k_test<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLrMaxDynLds));
    const int nps[] = {1, 2, 3, 5, 8, 11, 16, 20, 21, 22, 27, 28, 33, 38, 40};
    int worst_fail = 0;
    for (int padded = 1; padded < 2; ++padded)   // (the solve takes the padded layout only)
        for (int np : nps) {
            const int n = 6 * np, npad = (n + 15) & ~15, ld = padded ? npad : n;
            std::vector<double> H((size_t)NM * ld * ld, 0.0), B((size_t)NM * npad, 0.0);
            srand(1234 + np);
            auto rnd = [] { return (double)rand() / RAND_MAX - 0.5; };
            for (int m = 0; m < NM; ++m) {
                std::vector<double> G((size_t)n * n);
                for (auto &g : G) g = rnd();
                double *A = H.data() + (size_t)m * ld * ld;
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j <= i; ++j) {
                        double v = 0;
                        for (int k = 0; k < n; ++k) v += G[(size_t)i * n + k] * G[(size_t)j * n + k];
                        v = v / n * 50.0 + (i == j ? 1.0 + 100.0 * fabs(rnd()) : 0.0);
                        A[(size_t)i * ld + j] = A[(size_t)j * ld + i] = v;
                    }
                if (padded)
                    for (int i = n; i < npad; ++i) A[(size_t)i * ld + i] = 1.0;
                for (int i = 0; i < n; ++i) B[(size_t)m * npad + i] = 100.0 * rnd();
            }
            double *dH, *dB, *dX;
            long long *dD;
            int *dOk;
            CK(hipMalloc(&dH, H.size() * 8));
            CK(hipMalloc(&dB, B.size() * 8));
            CK(hipMalloc(&dX, B.size() * 8));
            CK(hipMalloc(&dD, (size_t)kDbg * 8 * NM));
            CK(hipMalloc(&dOk, 4 * NM));
            CK(hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
            CK(hipMemset(dX, 0, B.size() * 8));
            CK(hipMemset(dD, 0, (size_t)kDbg * 8 * NM));
            const size_t lds = ldlt_reg_lds_doubles(npad) * 8;
            for (int rep = 0; rep < 4; ++rep)   // (the counters of the LAST launch: instruction cache warm)
                hipLaunchKernelGGL(k_test<true>, dim3(NM), dim3(kLrThreads), lds, 0, dH, ld, n, npad, dB, dX, dD, dOk);
            CK(hipDeviceSynchronize());
            std::vector<double> X(B.size());
            std::vector<long long> D((size_t)kDbg * NM);
            std::vector<int> ok(NM);
            CK(hipMemcpy(X.data(), dX, X.size() * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ok.data(), dOk, 4 * NM, hipMemcpyDeviceToHost));
            double worst = 0;
            int nfail = 0;
            for (int m = 0; m < NM; ++m) {
                std::vector<double> A((size_t)n * n), b(n), xr;
                for (int i = 0; i < n; ++i) {
                    for (int j = 0; j < n; ++j) A[(size_t)i * n + j] = H[(size_t)m * ld * ld + (size_t)i * ld + j];
                    b[i] = B[(size_t)m * npad + i];
                }
                host_ldlt(A, n, b, xr);
                double num = 0, den = 0;
                for (int i = 0; i < n; ++i) {
                    const double e = X[(size_t)m * npad + i] - xr[i];
                    num += e * e;
                    den += xr[i] * xr[i];
                }
                const double rel = sqrt(num / (den + 1e-300));
                if (!(rel < 1e-9) || !ok[m]) ++nfail;
                if (rel > worst || rel != rel) worst = rel;
            }
            // time per launch without the counters
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_test<false>, dim3(NM), dim3(kLrThreads), lds, 0, dH, ld, n, npad, dB, dX, dD, dOk);
            CK(hipEventRecord(e0));
            const int reps = 50;
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_test<false>, dim3(NM), dim3(kLrThreads), lds, 0, dH, ld, n, npad, dB, dX, dD, dOk);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("np %2d n %3d npad %3d ld %3d: %d systems, worst rel err %.2e, failures %d | %.1f us per launch | cycles: load %lld D0 %lld P %lld U %lld (D inside %lld) wait %lld factor %lld backward %lld\n",
                   np, n, npad, ld, NM, worst, nfail, ms * 1e3 / reps, D[0], D[1], D[2], D[3], D[4], D[7], D[5], D[6]);
            printf("      worker 0: load %lld | P own %lld wait %lld | U own %lld wait %lld | backward own %lld wait %lld\n", D[8], D[9], D[10], D[11], D[12], D[13], D[14]);
            if (AOS2_LR_TRACE && getenv("LR_TRACE_NP") && atoi(getenv("LR_TRACE_NP")) == np) {   // timeline of block 0: per wave, (event, cycles since the first event)
                long long t0 = D[16] >> 12;
                for (int w = 0; w <= kLrWorkers; ++w) {
                    printf("  wave %d:", w);
                    for (int e = 0; e < AOS2_LR_TRACE; ++e) {
                        const long long v = D[16 + (size_t)w * AOS2_LR_TRACE + e];
                        if (!v) break;
                        printf(" %lld.%lld@%lld", (v & 4095) >> 4, v & 15, (v >> 12) - t0);
                    }
                    printf("\n");
                }
            }
            worst_fail += nfail;
            CK(hipFree(dH)); CK(hipFree(dB)); CK(hipFree(dX)); CK(hipFree(dD)); CK(hipFree(dOk));
        }
    // a singular system must be refused
    {
        const int n = 48, npad = 48, ld = 48;
        std::vector<double> H((size_t)ld * ld, 0.0), B(npad, 1.0);
        for (int i = 0; i < n; ++i) H[(size_t)i * ld + i] = i == 20 ? 0.0 : 2.0;
        double *dH, *dB, *dX;
        long long *dD;
        int *dOk;
        CK(hipMalloc(&dH, H.size() * 8)); CK(hipMalloc(&dB, B.size() * 8)); CK(hipMalloc(&dX, B.size() * 8)); CK(hipMalloc(&dD, (size_t)kDbg * 8)); CK(hipMalloc(&dOk, 4));
        CK(hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_test<false>, dim3(1), dim3(kLrThreads), ldlt_reg_lds_doubles(npad) * 8, 0, dH, ld, n, npad, dB, dX, dD, dOk);
        int ok = 1;
        CK(hipMemcpy(&ok, dOk, 4, hipMemcpyDeviceToHost));
        printf("singular system refused: %s\n", ok ? "NO" : "yes");
        if (ok) ++worst_fail;
    }
    printf(worst_fail ? "FAILED\n" : "ALL OK\n");
    return worst_fail ? 1 : 0;
}
