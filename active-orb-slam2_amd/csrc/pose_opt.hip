// Optimizer::PoseOptimization on gfx950 (reference src/Optimizer.cc:239-452; g2o: types_six_dof_expmap.cpp:266-364,
// optimization_algorithm_levenberg.cpp:61-164, linear_solver_dense.h:64-110).  SURVEY.md section 8(f) rank 1.
#include <cstdio>
#include <cstdlib>

#include "lba_math.h"

namespace aos2 {

// ---------------------------------------------------------------------------------------------
// Optimizer::PoseOptimization (src/Optimizer.cc:239-452): one workgroup per frame, the complete
// procedure on the device (no host round trips): residuals + Huber, EdgeSE3ProjectXYZOnlyPose /
// EdgeStereoSE3ProjectXYZOnlyPose Jacobians (types_six_dof_expmap.cpp:266-364), 6x6 normal
// equations by a fixed-order workgroup reduction, Cholesky, exp-map update, the Levenberg
// accept/reject logic (levenberg.cpp:61-164) and the outlier reclassification of :371-430.
// ---------------------------------------------------------------------------------------------
struct PoseProbDev {
    int n;
    const double *Xw, *obs;       // n x 3
    const double *w;              // n
    const uint8_t *stereo;        // n
    double *err;                  // n x 3 scratch
    uint8_t *level1, *robust;     // n scratch
    uint8_t *outlier;             // n out
    double fx, fy, cx, cy, bf;
    double pose_in[7];
    double *pose_out;             // 7
    int32_t *counts;              // [0] n_bad, [1] n_inliers
};

__device__ __forceinline__ void po_edge_error(const double *qt, const double *X, const double *obs, int stereo,
                                              const PoseProbDev &P, double er[3])
{
    double p[3];
    se3_map(qt, X, p);
    if (!stereo) {
        const double u = p[0] / p[2], v = p[1] / p[2];
        er[0] = obs[0] - (u * P.fx + P.cx);
        er[1] = obs[1] - (v * P.fy + P.cy);
        er[2] = 0;
    } else {
        const float invz = (float)(1.0 / p[2]);
        const double r0 = p[0] * invz * P.fx + P.cx;
        const double r1 = p[1] * invz * P.fy + P.cy;
        const double r2 = r0 - P.bf * invz;
        er[0] = obs[0] - r0;
        er[1] = obs[1] - r1;
        er[2] = obs[2] - r2;
    }
}

// fixed-order workgroup sum of K doubles per thread -> out[K] valid in every thread after return.
// The 256 partials of component k are added in thread order (8 slices of 32, then the 8 slice sums), instead of a
// log-depth tree with a barrier per level.
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double *sh /* 256 x (K+1) + 9 x K */, double *out)
{
    // fixed summation order (bit-reproducible): 8 slices of 32 threads, each summed in thread order, then the 8 slice
    // sums in slice order.  The 32 operands of a slice are fetched together before the dependent adds, and the 8-term
    // final sums are formed once (K threads) and broadcast, instead of every thread re-adding 8 x K partials.
    const int tid = threadIdx.x;
    for (int i = 0; i < K; ++i) sh[tid * (K + 1) + i] = v[i];
    __syncthreads();
    double *part = sh + 256 * (K + 1), *fin = part + 8 * K;
    if (tid < 8 * K) {
        const int k = tid % K, slice = tid / K;
        double x[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) x[t] = sh[(32 * slice + t) * (K + 1) + k];
        double acc = 0;
#pragma unroll
        for (int t = 0; t < 32; ++t) acc += x[t];
        part[slice * K + k] = acc;
    }
    __syncthreads();
    if (tid < K) {
        double acc = part[tid];
#pragma unroll
        for (int sl = 1; sl < 8; ++sl) acc += part[sl * K + tid];
        fin[tid] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < K; ++i) out[i] = fin[i];
    __syncthreads();
}

// kEpt > 0: every thread keeps its (<= kEpt) edges -- map point, observation, weight, residual, flags -- in
// registers for the whole procedure (n <= 256 * kEpt); the 40 LM iterations then touch no global memory.
// kEpt == 0: edges stay in global memory (any n).  Same arithmetic, same per-thread edge order either way.
template <int kEpt>
__global__ __launch_bounds__(256) void pose_optimization_kernel(const PoseProbDev *__restrict__ probs)
{
    constexpr int EPT = kEpt > 0 ? kEpt : 1;
    constexpr bool kReg = kEpt > 0;
    double Xr[EPT][3], Or[EPT][3], Wr[EPT], Er[EPT][3];
    uint8_t Sr[EPT], L1r[EPT], Rbr[EPT], Outr[EPT];
    extern __shared__ __attribute__((aligned(16))) double sh[];  // 256 x 28 + 9 x 27
    __shared__ double qt[7], bk[7], xs[6];
    __shared__ double s_lambda, s_ni, s_rho, s_currentChi;
    __shared__ int s_flag;
    const PoseProbDev P = probs[blockIdx.x];
    const int tid = threadIdx.x, n = P.n;
    // edge loop: thread tid owns edges tid, tid + 256, ... (slot j); accessors pick registers or global memory
#define PO_FOR_EDGES(j, e) for (int j = 0, e = tid; e < n && (!kReg || j < EPT); ++j, e += 256)
    auto Xp = [&](int j, int e) -> const double * { return kReg ? Xr[j] : P.Xw + 3 * e; };
    auto Op = [&](int j, int e) -> const double * { return kReg ? Or[j] : P.obs + 3 * e; };
    auto Ep = [&](int j, int e) -> double * { return kReg ? Er[j] : P.err + 3 * e; };
    auto Wv = [&](int j, int e) -> double { return kReg ? Wr[j] : P.w[e]; };
    auto Sv = [&](int j, int e) -> int { return kReg ? Sr[j] : P.stereo[e]; };
    auto L1 = [&](int j, int e) -> uint8_t & { return kReg ? L1r[j] : P.level1[e]; };
    auto Rb = [&](int j, int e) -> uint8_t & { return kReg ? Rbr[j] : P.robust[e]; };
    auto Ou = [&](int j, int e) -> uint8_t & { return kReg ? Outr[j] : P.outlier[e]; };
#pragma unroll EPT
    PO_FOR_EDGES(j, e) {
        if (kReg) {
            for (int k = 0; k < 3; ++k) {
                Xr[j][k] = P.Xw[3 * e + k];
                Or[j][k] = P.obs[3 * e + k];
            }
            Wr[j] = P.w[e];
            Sr[j] = P.stereo[e];
        }
        L1(j, e) = 0;
        Rb(j, e) = 1;
        Ou(j, e) = 0;
        double *er0 = Ep(j, e);
        er0[0] = er0[1] = er0[2] = 0;
    }
    if (tid < 7) qt[tid] = P.pose_in[tid];
    __syncthreads();
    if (n < 3) {  // nInitialCorrespondences < 3 (:355-356): the pose stays, mvbOutlier was already reset (:283, :320)
        for (int e = tid; e < n; e += 256) P.outlier[e] = 0;   // (the register copies above never reach memory here)
        if (tid < 7) P.pose_out[tid] = P.pose_in[tid];
        if (tid == 0) { P.counts[0] = 0; P.counts[1] = 0; }
        return;
    }
    const double delta_m = (double)(float)sqrt(5.991), delta_s = (double)(float)sqrt(7.815);
    int nBad = 0;
    // residuals of the active edges + robust chi2 (computeActiveErrors + activeRobustChi2)
    auto errors_chi2 = [&](double &chi_out) {
        double acc[1] = {0};
#pragma unroll EPT
        PO_FOR_EDGES(j, e) {
            if (L1(j, e)) continue;
            double er[3];
            const int st = Sv(j, e);
            po_edge_error(qt, Xp(j, e), Op(j, e), st, P, er);
            double *ee = Ep(j, e);
            ee[0] = er[0]; ee[1] = er[1]; ee[2] = er[2];
            double c = edge_chi2(er, Wv(j, e), st ? 3 : 2);
            if (Rb(j, e)) {
                double rho[2];
                robustify(c, st ? delta_s : delta_m, rho);
                c = rho[0];
            }
            acc[0] += c;
        }
        double out[1];
        block_sum<1>(acc, sh, out);
        chi_out = out[0];
    };
    for (int it = 0; it < 4; ++it) {
        if (tid < 7) qt[tid] = P.pose_in[tid];  // every round restarts from pFrame->mTcw (:368)
        __syncthreads();
        int n_active = 0;
        {
            double cnt[1] = {0}, out[1];
#pragma unroll EPT
            PO_FOR_EDGES(j, e) cnt[0] += L1(j, e) ? 0.0 : 1.0;
            block_sum<1>(cnt, sh, out);
            n_active = (int)out[0];
        }
        if (n_active > 0) {
            int nBadLM = 0;
            bool ok = true;
            bool fresh = false;   // the residuals and s_currentChi belong to the current pose (left by an accepted trial)
            for (int i = 0; i < 10 && ok; ++i) {
                // computeActiveErrors at the top of solve() (levenberg.cpp:75): recomputing at the pose of the accepted trial
                // reproduces its residuals and chi2 bit for bit, so it is skipped then
                double currentChi;
                if (fresh)
                    currentChi = s_currentChi;
                else
                    errors_chi2(currentChi);
                const double iniChi = currentChi;
                // buildSystem: H (upper triangle, 21) + b (6)
                double acc[27];
#pragma unroll
                for (int k = 0; k < 27; ++k) acc[k] = 0;
#pragma unroll EPT
                PO_FOR_EDGES(j, e) {
                    if (L1(j, e)) continue;
                    const int st = Sv(j, e), D = st ? 3 : 2;
                    double p[3];
                    se3_map(qt, Xp(j, e), p);
                    const double x = p[0], y = p[1], invz = 1.0 / p[2], invz_2 = invz * invz;
                    double J[18];
                    J[0] = x * y * invz_2 * P.fx;
                    J[1] = -(1 + (x * x * invz_2)) * P.fx;
                    J[2] = y * invz * P.fx;
                    J[3] = -invz * P.fx;
                    J[4] = 0;
                    J[5] = x * invz_2 * P.fx;
                    J[6] = (1 + y * y * invz_2) * P.fy;
                    J[7] = -x * y * invz_2 * P.fy;
                    J[8] = -x * invz * P.fy;
                    J[9] = 0;
                    J[10] = -invz * P.fy;
                    J[11] = y * invz_2 * P.fy;
                    J[12] = J[0] - P.bf * y * invz_2;
                    J[13] = J[1] + P.bf * x * invz_2;
                    J[14] = J[2];
                    J[15] = J[3];
                    J[16] = 0;
                    J[17] = J[5] - P.bf * invz_2;
                    const double *er = Ep(j, e);
                    const double w = Wv(j, e);
                    double wo = w, r1 = 1.0;
                    if (Rb(j, e)) {
                        double rho[2];
                        robustify(edge_chi2(er, w, D), st ? delta_s : delta_m, rho);
                        r1 = rho[1];
                        wo = rho[1] * w;
                    }
                    // static indices only (registers): the third row joins for stereo edges; 0 + a == a, so the
                    // sums equal the d-loops of the reference order
                    const bool st3 = D == 3;
                    int k = 0;
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        double sacc = J[r] * (w * er[0]);
                        sacc += J[6 + r] * (w * er[1]);
                        if (st3) sacc += J[12 + r] * (w * er[2]);
                        acc[21 + r] -= r1 * sacc;
#pragma unroll
                        for (int c = r; c < 6; ++c, ++k) {
                            double t = J[r] * wo * J[c];
                            t += J[6 + r] * wo * J[6 + c];
                            if (st3) t += J[12 + r] * wo * J[12 + c];
                            acc[k] += t;
                        }
                    }
                }
                double Hb[27];
                block_sum<27>(acc, sh, Hb);
                if (tid == 0) {
                    if (i == 0) {
                        double maxDiagonal = 0.;
                        constexpr int di[6] = {0, 6, 11, 15, 18, 20};
#pragma unroll
                        for (int d = 0; d < 6; ++d) maxDiagonal = fmax(fabs(Hb[di[d]]), maxDiagonal);
                        s_lambda = 1e-5 * maxDiagonal;
                        s_ni = 2;
                    }
                    s_currentChi = currentChi;
                }
                if (i == 0) nBadLM = 0;
                __syncthreads();
                double rho = 0;
                int qmax = 0;
                do {
                    if (tid == 0) {
                        for (int k = 0; k < 7; ++k) bk[k] = qt[k];
                        // (H + lambda I) x = b by Cholesky; "not positive" -> the step is rejected.  All loops have
                        // constant bounds and are unrolled so that L, y stay in registers (dynamic indexing would put
                        // them in scratch memory, on the serial path of every LM step); after a non-positive pivot
                        // the remaining arithmetic runs on but its result is discarded (pos = false).
                        double L[36];
                        {
                            int k = 0;
#pragma unroll
                            for (int r = 0; r < 6; ++r)
#pragma unroll
                                for (int c = r; c < 6; ++c, ++k) L[c * 6 + r] = L[r * 6 + c] = Hb[k];
                        }
#pragma unroll
                        for (int d = 0; d < 6; ++d) L[d * 7] += s_lambda;
                        bool pos = true;
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            double dd = L[j * 6 + j];
#pragma unroll
                            for (int m = 0; m < j; ++m) dd -= L[j * 6 + m] * L[j * 6 + m];
                            if (!(dd > 0)) pos = false;
                            dd = sqrt(dd);
                            L[j * 6 + j] = dd;
#pragma unroll
                            for (int r = j + 1; r < 6; ++r) {
                                double sacc = L[r * 6 + j];
#pragma unroll
                                for (int m = 0; m < j; ++m) sacc -= L[r * 6 + m] * L[j * 6 + m];
                                L[r * 6 + j] = sacc / dd;
                            }
                        }
                        if (pos) {
                            double yv[6], xv[6];
#pragma unroll
                            for (int r = 0; r < 6; ++r) {
                                double sacc = Hb[21 + r];
#pragma unroll
                                for (int m = 0; m < r; ++m) sacc -= L[r * 6 + m] * yv[m];
                                yv[r] = sacc / L[r * 6 + r];
                            }
#pragma unroll
                            for (int r = 5; r >= 0; --r) {
                                double sacc = yv[r];
#pragma unroll
                                for (int m = r + 1; m < 6; ++m) sacc -= L[m * 6 + r] * xv[m];
                                xv[r] = sacc / L[r * 6 + r];
                            }
#pragma unroll
                            for (int r = 0; r < 6; ++r) xs[r] = xv[r];
                        }
                        s_flag = pos ? 1 : 0;
                        double upd[6], T[7];
                        for (int k2 = 0; k2 < 6; ++k2) upd[k2] = xs[k2];
                        for (int k2 = 0; k2 < 7; ++k2) T[k2] = qt[k2];
                        se3_oplus(upd, T);
                        for (int k2 = 0; k2 < 7; ++k2) qt[k2] = T[k2];
                    }
                    __syncthreads();
                    double tempChi;
                    errors_chi2(tempChi);
                    if (tid == 0) {
                        if (!s_flag) tempChi = 1.7976931348623157e308;
                        double r = s_currentChi - tempChi;
                        double scale = 0.;
#pragma unroll
                        for (int j = 0; j < 6; ++j) scale += xs[j] * (s_lambda * xs[j] + Hb[21 + j]);
                        scale += 1e-3;
                        r /= scale;
                        if (r > 0 && isfinite(tempChi)) {
                            double alpha = 1. - pow((2 * r - 1), 3.0);
                            alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
                            const double scaleFactor = 1. / 3. > alpha ? 1. / 3. : alpha;
                            s_lambda *= scaleFactor;
                            s_ni = 2;
                            s_currentChi = tempChi;
                        } else {
                            s_lambda *= s_ni;
                            s_ni *= 2;
                            for (int k = 0; k < 7; ++k) qt[k] = bk[k];
                        }
                        s_rho = r;
                    }
                    __syncthreads();
                    rho = s_rho;
                    qmax++;
                } while (rho < 0 && qmax < 10);
                fresh = rho > 0;   // the trial was accepted (r > 0 && isfinite(tempChi): chi2 is never negative, so r > 0 implies it)
                const double curChi = s_currentChi;
                if (qmax == 10 || rho == 0) {
                    ok = false;
                } else {
                    if ((iniChi - curChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
                    if (nBadLM >= 3) ok = false;
                }
                __syncthreads();
            }
        }
        // outlier reclassification (:371-430)
        double bad[1] = {0}, outb[1];
#pragma unroll EPT
        PO_FOR_EDGES(j, e) {
            const int st = Sv(j, e);
            if (Ou(j, e)) {
                double er[3];
                po_edge_error(qt, Xp(j, e), Op(j, e), st, P, er);
                double *ee = Ep(j, e);
                ee[0] = er[0]; ee[1] = er[1]; ee[2] = er[2];
            }
            const float chi2 = (float)edge_chi2(Ep(j, e), Wv(j, e), st ? 3 : 2);
            if (chi2 > (st ? 7.815f : 5.991f)) {
                Ou(j, e) = 1;
                L1(j, e) = 1;
                bad[0] += 1.0;
            } else {
                Ou(j, e) = 0;
                L1(j, e) = 0;
            }
            if (it == 2) Rb(j, e) = 0;
        }
        block_sum<1>(bad, sh, outb);
        nBad = (int)outb[0];
        if (n < 10) break;  // optimizer.edges().size() < 10
    }
    if (kReg) {
#pragma unroll EPT
        PO_FOR_EDGES(j, e) P.outlier[e] = Outr[j];
    }
    if (tid < 7) P.pose_out[tid] = qt[tid];
    if (tid == 0) {
        P.counts[0] = nBad;
        P.counts[1] = n - nBad;
    }
#undef PO_FOR_EDGES
}

}  // namespace aos2

using namespace aos2;

extern "C" {

int aos2_pose_optimization(aos2_lba_t *s, const aos2_pose_problem_t *problems, aos2_pose_result_t *results, int n_problems)
{
    if (!s || !problems || !results || n_problems <= 0) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    for (int i = 0; i < n_problems; ++i)
        if (problems[i].n < 0 || (problems[i].n > 0 && (!problems[i].Xw || !problems[i].obs || !problems[i].stereo ||
                                                        !problems[i].inv_sigma2 || !results[i].outlier))) {
            set_error("bad pose problem %d", i);
            return AOS2_ERR_ARG;
        }
    int st = lba_handle_init(s);
    if (st) return st;
    // Inputs are converted in place into the handle's page-locked staging buffer (one asynchronous upload that
    // also carries the problem descriptors); the results of all problems -- pose, counts, outlier flags -- are
    // neighbours in the arena and come back as ONE copy (the former three small pageable copies per problem cost
    // 2 ms of host time for 64 frames, against 0.55 ms of kernel).
    HostArena H;
    struct Off { size_t xw, obs, w, st, err, l1, rb, out, pose, cnt; };
    std::vector<Off> offs(n_problems);
    size_t in_cap = sizeof(PoseProbDev) * (size_t)n_problems + 512;
    for (int i = 0; i < n_problems; ++i) in_cap += (size_t)problems[i].n * (24 + 24 + 8 + 1) + 4 * 256 + 32;
    if ((st = s->h_in.alloc(in_cap))) return st;
    H.host = s->h_in.p;
    H.host_cap = in_cap;
    for (int i = 0; i < n_problems; ++i) {
        const aos2_pose_problem_t &p = problems[i];
        const size_t n = (size_t)p.n;
        double *xw = H.push_fill<double>(3 * n + 1, offs[i].xw), *ob = H.push_fill<double>(3 * n + 1, offs[i].obs);
        double *w = H.push_fill<double>(n + 1, offs[i].w);
        uint8_t *sv = H.push_fill<uint8_t>(n + 1, offs[i].st);
        if (!xw || !ob || !w || !sv) {
            set_error("internal: pose optimisation input staging");
            return AOS2_ERR_ARG;
        }
        for (size_t k = 0; k < 3 * n; ++k) xw[k] = (double)p.Xw[k];
        for (size_t k = 0; k < 3 * n; ++k) ob[k] = (double)p.obs[k];
        for (size_t k = 0; k < n; ++k) w[k] = (double)p.inv_sigma2[k];
        xw[3 * n] = ob[3 * n] = w[n] = 0.0;
        if (n) memcpy(sv, p.stereo, n);
        sv[n] = 0;
    }
    size_t o_probs;
    PoseProbDev *dev = H.push_fill<PoseProbDev>((size_t)n_problems, o_probs);   // filled below (needs the device base)
    if (!dev) {
        set_error("internal: pose optimisation input staging");
        return AOS2_ERR_ARG;
    }
    const size_t in_bytes = H.host_size;
    for (int i = 0; i < n_problems; ++i) {
        const size_t n = (size_t)problems[i].n;
        offs[i].err = H.push(nullptr, (3 * n + 1) * 8);
        offs[i].l1 = H.push(nullptr, n + 1);
        offs[i].rb = H.push(nullptr, n + 1);
    }
    const size_t o_res = (H.size + 255) & ~(size_t)255;   // results of all problems from here on
    for (int i = 0; i < n_problems; ++i) {
        offs[i].pose = H.push(nullptr, 7 * 8);
        offs[i].cnt = H.push(nullptr, 8);
        offs[i].out = H.push(nullptr, (size_t)problems[i].n + 1);
    }
    const size_t res_bytes = H.size - o_res;
    if ((st = s->arena.alloc(H.size + 256))) return st;
    if ((st = s->h_stage.alloc(res_bytes + 64))) return st;
    uint8_t *base = s->arena.p;
    for (int i = 0; i < n_problems; ++i) {
        const aos2_pose_problem_t &p = problems[i];
        PoseProbDev &D = dev[i];
        D.n = p.n;
        D.Xw = (const double *)(base + offs[i].xw); D.obs = (const double *)(base + offs[i].obs);
        D.w = (const double *)(base + offs[i].w); D.stereo = base + offs[i].st;
        D.err = (double *)(base + offs[i].err); D.level1 = base + offs[i].l1; D.robust = base + offs[i].rb;
        D.outlier = base + offs[i].out; D.pose_out = (double *)(base + offs[i].pose); D.counts = (int32_t *)(base + offs[i].cnt);
        D.fx = (double)p.fx; D.fy = (double)p.fy; D.cx = (double)p.cx; D.cy = (double)p.cy; D.bf = (double)p.bf;
        pose_from_Tcw(p.Tcw, D.pose_in);
    }
    hipStream_t q = s->stream;
    AOS2_HIP_CHECK(hipMemcpyAsync(base, H.data(), in_bytes, hipMemcpyHostToDevice, q));
    AOS2_HIP_CHECK(hipEventRecord(s->ev[0], q));
    int max_n = 0;
    for (int i = 0; i < n_problems; ++i) max_n = std::max(max_n, problems[i].n);
    const size_t po_lds = (256 * 28 + 9 * 27) * sizeof(double);
    if (max_n <= 256 * 4)   // the usual case (a frame has <= ~1000 map-point matches): edges live in registers
        hipLaunchKernelGGL(pose_optimization_kernel<4>, dim3(n_problems), dim3(256), po_lds, q, (const PoseProbDev *)(base + o_probs));
    else
        hipLaunchKernelGGL(pose_optimization_kernel<0>, dim3(n_problems), dim3(256), po_lds, q, (const PoseProbDev *)(base + o_probs));
    AOS2_HIP_CHECK(hipEventRecord(s->ev[1], q));
    const uint8_t *res = s->h_stage.p;
    AOS2_HIP_CHECK(hipMemcpyAsync(s->h_stage.p, base + o_res, res_bytes, hipMemcpyDeviceToHost, q));
    AOS2_HIP_CHECK(hipStreamSynchronize(q));
    AOS2_HIP_CHECK(hipGetLastError());
    (void)hipEventElapsedTime(&s->last_pose_ms, s->ev[0], s->ev[1]);
    for (int i = 0; i < n_problems; ++i) {
        const double *pose = reinterpret_cast<const double *>(res + (offs[i].pose - o_res));
        const int32_t *cnt = reinterpret_cast<const int32_t *>(res + (offs[i].cnt - o_res));
        if (problems[i].n > 0) memcpy(results[i].outlier, res + (offs[i].out - o_res), (size_t)problems[i].n);
        if (problems[i].n < 3)   // the reference returns before touching mTcw (:355-356): keep the caller's matrix bit for bit
            memcpy(results[i].Tcw, problems[i].Tcw, sizeof(float) * 16);
        else
            pose_to_Tcw(pose, results[i].Tcw);
        results[i].n_bad = cnt[0];
        results[i].n_inliers = cnt[1];
    }
    return AOS2_OK;
}


float aos2_pose_optimization_last_device_ms(const aos2_lba_t *s) { return s ? s->last_pose_ms : 0.f; }

}  // extern "C"

#include "frames_pose.inc"
