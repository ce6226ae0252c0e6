// Optimizer::PoseOptimization on gfx950 (reference src/Optimizer.cc:239-452; g2o: types_six_dof_expmap.cpp:266-364,
// optimization_algorithm_levenberg.cpp:61-164, linear_solver_dense.h:64-110).  SURVEY.md section 8(f) rank 1.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

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
    const float *Xw, *obs;        // n x 3 (the caller's float32 values; widened on use)
    const float *w;               // n
    const uint8_t *stereo;        // n
    double *err;                  // n scratch: chi2 of the edge's stored residual (e->chi2())
    uint8_t *level1, *robust;     // n scratch
    uint8_t *outlier;             // n out
    double fx, fy, cx, cy, bf;
    double pose_in[7];
    double *pose_out;             // 7
    int32_t *counts;              // [0] n_bad, [1] n_inliers
};

__device__ __forceinline__ void po_edge_error(const double *qt, const float *Xf, const float *obf, int stereo,
                                              const PoseProbDev &P, double er[3])
{
    const double X[3] = {(double)Xf[0], (double)Xf[1], (double)Xf[2]};
    const double obs[3] = {(double)obf[0], (double)obf[1], (double)obf[2]};
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

// 1 / d within an ulp: v_rcp_f64 + two Newton steps (34 cycles of latency; the IEEE division sequence takes 67)
__device__ __forceinline__ double rcp_newton(double d)
{
    double rd = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, rd, 1.0);
    rd = __builtin_fma(rd, e, rd);
    e = __builtin_fma(-d, rd, 1.0);
    return __builtin_fma(rd, e, rd);
}

// (H + lambda I) x = b for the 6x6 system (linear_solver_dense.h:64-110: a Cholesky factorisation), H = upper triangle
// packed row by row in Hb[0..20], b = Hb[21..26].  Returns false on a non-positive pivot ("not positive definite": the LM
// step is rejected) and leaves x alone then, like the solver's x vector in g2o.  On the serial path of every LM trial, so
// it is shaped for latency: the square-root-free form L D L^T (same pivots d_j as the squares of Cholesky's diagonal, so
// the same positivity test; x equal up to rounding), right-looking so that the updates of a column are independent
// operations, one reciprocal per pivot (rcp_newton) instead of 27 divisions + 6 square roots; everything in registers.
__device__ __forceinline__ bool solve6(const double *Hb, double lambda, double (&x)[6])
{
    double a[6][6], rd[6];
    {
        int k = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = r; c < 6; ++c, ++k) a[c][r] = Hb[k];   // lower triangle a[row][col], row >= col
    }
#pragma unroll
    for (int d = 0; d < 6; ++d) a[d][d] += lambda;
    bool pos = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const double d = a[j][j];
        if (!(d > 0)) pos = false;
        rd[j] = rcp_newton(d);
        double l[6];
#pragma unroll
        for (int r = j + 1; r < 6; ++r) l[r] = a[r][j] * rd[j];
#pragma unroll
        for (int r = j + 1; r < 6; ++r)
#pragma unroll
            for (int c = j + 1; c <= r; ++c) a[r][c] = __builtin_fma(-l[r], a[c][j], a[r][c]);
#pragma unroll
        for (int r = j + 1; r < 6; ++r) a[r][j] = l[r];
    }
    double y[6], xv[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        double sacc = Hb[21 + r];
#pragma unroll
        for (int m = 0; m < r; ++m) sacc = __builtin_fma(-a[r][m], y[m], sacc);
        y[r] = sacc;
    }
#pragma unroll
    for (int r = 5; r >= 0; --r) {
        double sacc = y[r] * rd[r];
#pragma unroll
        for (int m = r + 1; m < 6; ++m) sacc = __builtin_fma(-a[m][r], xv[m], sacc);
        xv[r] = sacc;
    }
    if (pos) {
#pragma unroll
        for (int r = 0; r < 6; ++r) x[r] = xv[r];
    }
    return pos;
}

// cycle counters of the phases (s_memtime), printed by workgroup 0: compile with -DAOS2_PO_TIMING (tools only)
#ifdef AOS2_PO_TIMING
#define PO_T(...) __VA_ARGS__
#else
#define PO_T(...)
#endif

constexpr int kPoSum = 28;   // 21 (H, upper triangle) + 6 (b) + 1 (robust chi2)
constexpr size_t kPoLds = (4 * kPoSum + 2 * kPoSum) * sizeof(double);   // 4 wave sums, two result buffers (256 threads)
constexpr size_t po_lds_bytes(int nt) { return ((size_t)(nt / 64) * kPoSum + 2 * kPoSum) * sizeof(double); }

__device__ __forceinline__ double pair_f64(unsigned lo, unsigned hi) { return __longlong_as_double(((long long)hi << 32) | lo); }

// sum of x over the lane pairs {l, l + 32} (kSwap32) or {l, l + 16} of a wave, for two values at once: the lower half
// (even 16-lane rows) ends up with the pair sums of x, the upper half (odd rows) with those of y.  One
// v_permlane{32,16}_swap per dword (gfx950) moves x up and y down in the same instruction.
template <bool kSwap32>
__device__ __forceinline__ double swap_add(double x, double y)
{
    const long long xb = __double_as_longlong(x), yb = __double_as_longlong(y);
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    u2 lo, hi;
    if (kSwap32) {
        lo = __builtin_amdgcn_permlane32_swap((unsigned)xb, (unsigned)yb, false, false);
        hi = __builtin_amdgcn_permlane32_swap((unsigned)(xb >> 32), (unsigned)(yb >> 32), false, false);
    } else {
        lo = __builtin_amdgcn_permlane16_swap((unsigned)xb, (unsigned)yb, false, false);
        hi = __builtin_amdgcn_permlane16_swap((unsigned)(xb >> 32), (unsigned)(yb >> 32), false, false);
    }
    return pair_f64(lo[0], hi[0]) + pair_f64(lo[1], hi[1]);
}

template <int kCtrl>
__device__ __forceinline__ double po_dpp_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)b, (int)b, kCtrl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(b >> 32), (int)(b >> 32), kCtrl, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// fixed-order workgroup sum of the 28 doubles of every thread -> fin[28] in LDS, readable by every thread after return
// (bit-reproducible: the order is a function of the lane number only).  Per wave a reduce-scatter: the two halves of the
// wave exchange halves of the values (28 -> 14 per lane, v_permlane32_swap), the 16-lane rows again (-> 7 per lane,
// v_permlane16_swap), then 7 DPP row sums; 147 instructions instead of the 504 of 28 full butterflies, and 112 doubles of
// LDS traffic per workgroup instead of 57 KB.  The 4 wave sums of each value are added by 28 threads.  Two barriers are
// enough: a wave rewrites part[wave] only after it left the previous call's second barrier, which the readers of the
// previous `part` reach after reading; `fin` is the caller's and alternates between two buffers.
template <int NT = 256>
__device__ __forceinline__ void block_sum28(double (&v)[kPoSum], double *part /* NT / 64 x 28 */, double *fin)
{
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double h[14], g[7];
#pragma unroll
    for (int i = 0; i < 14; ++i) h[i] = swap_add<true>(v[i], v[14 + i]);   // lanes < 32: values 0..13, lanes >= 32: 14..27
#pragma unroll
    for (int i = 0; i < 7; ++i) g[i] = swap_add<false>(h[i], h[7 + i]);    // row r of the wave: values 7 r .. 7 r + 6
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        double x = g[i];
        x += po_dpp_f64<0xB1>(x);    // quad_perm [1, 0, 3, 2]
        x += po_dpp_f64<0x4E>(x);    // quad_perm [2, 3, 0, 1]
        x += po_dpp_f64<0x141>(x);   // row_half_mirror
        x += po_dpp_f64<0x140>(x);   // row_mirror
        g[i] = x;
    }
    if ((lane & 15) == 0) {
        double *dst = part + wave * kPoSum + 7 * (lane >> 4);
#pragma unroll
        for (int i = 0; i < 7; ++i) dst[i] = g[i];
    }
    __syncthreads();
    if (tid < kPoSum) {   // the wave sums in wave order (for 4 waves: ((p0 + p1) + p2) + p3)
        double sacc = part[tid];
#pragma unroll
        for (int w = 1; w < NW; ++w) sacc += part[w * kPoSum + tid];
        fin[tid] = sacc;
    }
    __syncthreads();
}

// The whole procedure for one frame by one workgroup of 256 threads.
// kEpt > 0: every thread keeps its (<= kEpt) edges -- map point, observation, weight, residual, flags -- in
// registers for the whole procedure (n <= 256 * kEpt); the LM iterations then touch no global memory.
// kEpt == 0: edges stay in global memory (any n).  Same arithmetic, same per-thread edge order either way.
//
// Every thread carries the pose and the LM state (lambda, ni, chi2) and runs the 6x6 solve, the exp-map update and the
// accept / reject decision REDUNDANTLY from the broadcast sums: no thread-0 sections, no barriers besides the three of
// the sum.  One pass over the edges at a pose yields residuals, robust chi2 AND the normal equations there (one sum of
// 28): at the trial pose that is the chi2 the decision needs and -- when the trial is accepted, the usual case -- the
// system of the next iteration (g2o recomputes the same numbers in computeActiveErrors + buildSystem at the top of the
// next solve(), levenberg.cpp:75-88); after a rejected trial the previous system is still in registers.
// NT: threads of the workgroup (256, or 512 / 1024 for the latency-bound small batches: fewer edges per thread = a shorter
// dependent chain per pass; the per-thread edge order and the order of the wave sums are functions of NT, so results of
// different NT agree to rounding, not bit for bit -- a batch always runs ONE variant).
template <int kEpt, int NT = 256>
__device__ __forceinline__ void pose_optimization_body(const PoseProbDev &P, double *sh, int *s_cnt)
{
    constexpr int EPT = kEpt > 0 ? kEpt : 1;
    constexpr bool kReg = kEpt > 0;
    float Xr[EPT][3], Or[EPT][3], Wr[EPT];
    double Cr[EPT];   // chi2 of the edge's stored residual (what e->chi2() returns between two computeError calls)
    uint8_t Sr[EPT], L1r[EPT], Rbr[EPT], Outr[EPT];
    const int tid = threadIdx.x, n = P.n;
    // edge loop: thread tid owns edges tid, tid + 256, ... (slot j); accessors pick registers or global memory
#define PO_FOR_EDGES(j, e) for (int j = 0, e = tid; e < n && (!kReg || j < EPT); ++j, e += NT)
    auto Xp = [&](int j, int e) -> const float * { return kReg ? Xr[j] : P.Xw + 3 * e; };
    auto Op = [&](int j, int e) -> const float * { return kReg ? Or[j] : P.obs + 3 * e; };
    auto Cp = [&](int j, int e) -> double & { return kReg ? Cr[j] : P.err[e]; };
    auto Wv = [&](int j, int e) -> double { return (double)(kReg ? Wr[j] : P.w[e]); };
    auto Sv = [&](int j, int e) -> int { return kReg ? Sr[j] : P.stereo[e]; };
    auto L1 = [&](int j, int e) -> uint8_t & { return kReg ? L1r[j] : P.level1[e]; };
    auto Rb = [&](int j, int e) -> uint8_t & { return kReg ? Rbr[j] : P.robust[e]; };
    auto Ou = [&](int j, int e) -> uint8_t & { return kReg ? Outr[j] : P.outlier[e]; };
    if constexpr (kReg) {
#pragma unroll
        for (int j = 0; j < EPT; ++j) {   // (an empty slot holds zeros: the K-at-a-time edge block evaluates it with weight 0)
            const int e = tid + j * NT;
            const bool in = e < n;
            for (int k = 0; k < 3; ++k) {
                Xr[j][k] = in ? P.Xw[3 * e + k] : 0.f;
                Or[j][k] = in ? P.obs[3 * e + k] : 0.f;
            }
            Wr[j] = in ? P.w[e] : 0.f;
            Sr[j] = in ? P.stereo[e] : 0;
            L1r[j] = 0;
            Rbr[j] = 1;
            Outr[j] = 0;
            Cr[j] = 0;
        }
    } else {
        PO_FOR_EDGES(j, e) {
            L1(j, e) = 0;
            Rb(j, e) = 1;
            Ou(j, e) = 0;
            Cp(j, e) = 0;
        }
    }
    double qt[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) qt[k] = P.pose_in[k];
    if (n < 3) {  // nInitialCorrespondences < 3 (:355-356): the pose stays, mvbOutlier was already reset (:283, :320)
        for (int e = tid; e < n; e += NT) P.outlier[e] = 0;   // (the register copies above never reach memory here)
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < 7; ++k) P.pose_out[k] = qt[k];
            P.counts[0] = 0;
            P.counts[1] = 0;
        }
        return;
    }
    const double delta_m = (double)(float)sqrt(5.991), delta_s = (double)(float)sqrt(7.815);
    // one pass over the active edges at pose q: residuals (stored), robust chi2, H (upper triangle) and b
    // (computeActiveErrors + activeRobustChi2 + buildSystem; types_six_dof_expmap.cpp:266-364, base_unary_edge.hpp)
    PO_T(long long t_edges = 0, t_sum = 0, t_solve = 0, t_dec = 0, t_all = __builtin_amdgcn_s_memtime(); int n_pass = 0, n_trial = 0;)
    // kReg: the edges of a wave's threads are processed K at a time as ONE straight-line block (no branch: the mono / stereo forms,
    // the Huber case and "edge is at level 1 / slot is empty" are selects; an inactive slot is the point (0, 0, 1) with weight 0, whose
    // terms are exact zeros), so that the K dependent chains -- ~200 f64 instructions of 9 cycles latency each -- interleave.  K is
    // the number of slots the WAVE uses (a scalar branch); the sums take the slots' terms in slot order, as the one-by-one loop did.
    const int nsl_wave = kReg ? __builtin_amdgcn_readfirstlane(min(EPT, max(0, (n - (tid & ~63) + NT - 1) / NT))) : 0;
    auto edges = [&](auto kc, auto jc, double (&acc)[kPoSum], const double (&q)[7], const double (&Rm)[9]) {
        constexpr int K = decltype(kc)::value, J0 = decltype(jc)::value;
        const double fx = P.fx, fy = P.fy, cx = P.cx, cy = P.cy, bf = P.bf;
        double x[K], y[K], z[K], invz[K], a[K], b[K], w[K], er[K][3], c[K], wo[K], r1[K], cr[K];
        bool st3[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int j = J0 + k;
            const bool act = tid + j * NT < n && !L1r[j];
            st3[k] = Sr[j] != 0;
            const double X0 = (double)Xr[j][0], X1 = (double)Xr[j][1], X2 = (double)Xr[j][2];
            const double xx = __builtin_fma(Rm[2], X2, __builtin_fma(Rm[1], X1, __builtin_fma(Rm[0], X0, q[4])));
            const double yy = __builtin_fma(Rm[5], X2, __builtin_fma(Rm[4], X1, __builtin_fma(Rm[3], X0, q[5])));
            const double zz = __builtin_fma(Rm[8], X2, __builtin_fma(Rm[7], X1, __builtin_fma(Rm[6], X0, q[6])));
            x[k] = act ? xx : 0.0;
            y[k] = act ? yy : 0.0;
            z[k] = act ? zz : 1.0;
            w[k] = act ? (double)Wr[j] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) invz[k] = rcp_newton(z[k]);   // (1 / z within an ulp)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int j = J0 + k;
            a[k] = x[k] * invz[k];
            b[k] = y[k] * invz[k];
            const double ob0 = (double)Or[j][0], ob1 = (double)Or[j][1], ob2 = (double)Or[j][2];
            // EdgeStereoSE3ProjectXYZOnlyPose::cam_project: invz is a float there (types_six_dof_expmap.cpp:338)
            const float invzf = (float)invz[k];
            const double s0 = x[k] * invzf * fx + cx;
            const double s1 = y[k] * invzf * fy + cy;
            const double s2 = s0 - bf * invzf;
            er[k][0] = ob0 - (st3[k] ? s0 : __builtin_fma(a[k], fx, cx));
            er[k][1] = ob1 - (st3[k] ? s1 : __builtin_fma(b[k], fy, cy));
            er[k][2] = st3[k] ? ob2 - s2 : 0.0;
            c[k] = edge_chi2(er[k], w[k], 3);   // (a mono edge's third term is an exact zero)
            Cr[j] = c[k];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {   // RobustKernelHuber::robustify (robust_kernel_impl.cpp:78-91)
            const int j = J0 + k;
            const double delta = st3[k] ? delta_s : delta_m, dsqr = delta * delta;
            double sq, rs;
            sqrt_rsqrt(c[k], sq, rs);
            const bool hub = Rbr[j] && c[k] > dsqr;
            cr[k] = hub ? 2 * sq * delta - dsqr : c[k];
            r1[k] = hub ? delta * rs : 1.0;
            wo[k] = hub ? r1[k] * w[k] : w[k];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            acc[27] += cr[k];
            double J[18];
            const double fxz = fx * invz[k], fyz = fy * invz[k], ab = a[k] * b[k];
            J[0] = ab * fx;
            J[1] = -(__builtin_fma(a[k], a[k], 1.0) * fx);
            J[2] = b[k] * fx;
            J[3] = -fxz;
            J[4] = 0;
            J[5] = a[k] * fxz;
            J[6] = __builtin_fma(b[k], b[k], 1.0) * fy;
            J[7] = -(ab * fy);
            J[8] = -(a[k] * fy);
            J[9] = 0;
            J[10] = -fyz;
            J[11] = b[k] * fyz;
            const double bfz2 = bf * (invz[k] * invz[k]);
            J[12] = __builtin_fma(-bfz2, y[k], J[0]);
            J[13] = __builtin_fma(bfz2, x[k], J[1]);
            J[14] = J[2];
            J[15] = J[3];
            J[16] = 0;
            J[17] = J[5] - bfz2;
            const double wo2 = st3[k] ? wo[k] : 0.0;
            const double we0 = r1[k] * (w[k] * er[k][0]), we1 = r1[k] * (w[k] * er[k][1]), we2 = st3[k] ? r1[k] * (w[k] * er[k][2]) : 0.0;
            int m = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const double j0 = J[r] * wo[k], j1 = J[6 + r] * wo[k], j2 = J[12 + r] * wo2;
                double bb = acc[21 + r];
                if (r != 4) bb = __builtin_fma(-J[r], we0, bb);
                if (r != 3) bb = __builtin_fma(-J[6 + r], we1, bb);
                if (r != 4) bb = __builtin_fma(-J[12 + r], we2, bb);
                acc[21 + r] = bb;
#pragma unroll
                for (int c2 = r; c2 < 6; ++c2, ++m) {
                    double t = acc[m];
                    if (r != 4 && c2 != 4) t = __builtin_fma(j0, J[c2], t);
                    if (r != 3 && c2 != 3) t = __builtin_fma(j1, J[6 + c2], t);
                    if (r != 4 && c2 != 4) t = __builtin_fma(j2, J[12 + c2], t);
                    acc[m] = t;
                }
            }
        }
    };
    auto pass = [&](const double (&q)[7], double *fin) {
        PO_T(const long long p0 = __builtin_amdgcn_s_memtime();)
        double acc[kPoSum];
#pragma unroll
        for (int k = 0; k < kPoSum; ++k) acc[k] = 0;
        // The rotation as a matrix, once per pass (9 fused multiply-adds per point instead of the ~33 operations of Eigen's
        // quaternion * vector), and the Jacobian from a = x / z, b = y / z: the same quantities as
        // types_six_dof_expmap.cpp:266-364 in fewer operations -- results equal to rounding (1e-16 relative).
        double Rm[9];
        rot_from_quat(q, Rm);
        if constexpr (kReg) {
            using std::integral_constant;
            if constexpr (EPT == 4) {
                if (nsl_wave == 1) edges(integral_constant<int, 1>(), integral_constant<int, 0>(), acc, q, Rm);
                else if (nsl_wave == 2) edges(integral_constant<int, 2>(), integral_constant<int, 0>(), acc, q, Rm);
                else if (nsl_wave == 3) edges(integral_constant<int, 3>(), integral_constant<int, 0>(), acc, q, Rm);
                else if (nsl_wave == 4) edges(integral_constant<int, 4>(), integral_constant<int, 0>(), acc, q, Rm);
            } else {
                auto chunks = [&](auto self, auto jc) {
                    constexpr int jj = decltype(jc)::value;
                    if constexpr (jj < EPT) {
                        if constexpr (jj + 1 < EPT) {
                            if (jj + 1 < nsl_wave) edges(integral_constant<int, 2>(), jc, acc, q, Rm);
                            else if (jj < nsl_wave) edges(integral_constant<int, 1>(), jc, acc, q, Rm);
                        } else {
                            if (jj < nsl_wave) edges(integral_constant<int, 1>(), jc, acc, q, Rm);
                        }
                        self(self, integral_constant<int, jj + 2>());
                    }
                };
                chunks(chunks, integral_constant<int, 0>());
            }
        } else {
        const double fx = P.fx, fy = P.fy, cx = P.cx, cy = P.cy, bf = P.bf;
#pragma unroll EPT
        PO_FOR_EDGES(j, e) {
            if (L1(j, e)) continue;
            const int st = Sv(j, e);
            const bool st3 = st != 0;
            const float *Xf = Xp(j, e), *obf = Op(j, e);
            const double X0 = (double)Xf[0], X1 = (double)Xf[1], X2 = (double)Xf[2];
            const double ob[3] = {(double)obf[0], (double)obf[1], (double)obf[2]};
            const double x = __builtin_fma(Rm[2], X2, __builtin_fma(Rm[1], X1, __builtin_fma(Rm[0], X0, q[4])));
            const double y = __builtin_fma(Rm[5], X2, __builtin_fma(Rm[4], X1, __builtin_fma(Rm[3], X0, q[5])));
            const double z = __builtin_fma(Rm[8], X2, __builtin_fma(Rm[7], X1, __builtin_fma(Rm[6], X0, q[6])));
            const double invz = rcp_newton(z);   // (1 / z within an ulp)
            const double a = x * invz, b = y * invz;
            double er[3];
            if (!st3) {
                er[0] = ob[0] - __builtin_fma(a, fx, cx);
                er[1] = ob[1] - __builtin_fma(b, fy, cy);
                er[2] = 0;
            } else {   // EdgeStereoSE3ProjectXYZOnlyPose::cam_project: invz is a float there (types_six_dof_expmap.cpp:338)
                const float invzf = (float)invz;
                const double r0 = x * invzf * fx + cx;
                const double r1 = y * invzf * fy + cy;
                const double r2 = r0 - bf * invzf;
                er[0] = ob[0] - r0;
                er[1] = ob[1] - r1;
                er[2] = ob[2] - r2;
            }
            const double w = Wv(j, e);
            const double c = edge_chi2(er, w, st3 ? 3 : 2);
            Cp(j, e) = c;
            double wo = w, r1 = 1.0, cr = c;
            if (Rb(j, e)) {   // RobustKernelHuber::robustify (robust_kernel_impl.cpp:78-91)
                const double delta = st3 ? delta_s : delta_m, dsqr = delta * delta;
                if (c > dsqr) {
                    double sq, rs;
                    sqrt_rsqrt(c, sq, rs);
                    cr = 2 * sq * delta - dsqr;
                    r1 = delta * rs;
                    wo = r1 * w;
                }
            }
            acc[27] += cr;
            double J[18];
            const double fxz = fx * invz, fyz = fy * invz, ab = a * b;
            J[0] = ab * fx;
            J[1] = -(__builtin_fma(a, a, 1.0) * fx);
            J[2] = b * fx;
            J[3] = -fxz;
            J[4] = 0;
            J[5] = a * fxz;
            J[6] = __builtin_fma(b, b, 1.0) * fy;
            J[7] = -(ab * fy);
            J[8] = -(a * fy);
            J[9] = 0;
            J[10] = -fyz;
            J[11] = b * fyz;
            const double bfz2 = bf * (invz * invz);
            J[12] = __builtin_fma(-bfz2, y, J[0]);
            J[13] = __builtin_fma(bfz2, x, J[1]);
            J[14] = J[2];
            J[15] = J[3];
            J[16] = 0;
            J[17] = J[5] - bfz2;
            // H += J^T (wo I) J, b -= r1 J^T (w e) as fused multiply-adds into the per-thread sums (the association differs
            // from the reference's "t = sum over rows, H += t" by rounding only: same terms, and the workgroup sum reorders
            // them anyway).  A mono edge has no third row: its weight is 0 there.
            // J[4] = J[6 + 3] = J[12 + 4] = 0: those products are left out (45 + 15 fused multiply-adds instead of 63 + 18)
            const double wo2 = st3 ? wo : 0.0;
            const double we0 = r1 * (w * er[0]), we1 = r1 * (w * er[1]), we2 = st3 ? r1 * (w * er[2]) : 0.0;
            int k = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const double j0 = J[r] * wo, j1 = J[6 + r] * wo, j2 = J[12 + r] * wo2;
                double bb = acc[21 + r];
                if (r != 4) bb = __builtin_fma(-J[r], we0, bb);
                if (r != 3) bb = __builtin_fma(-J[6 + r], we1, bb);
                if (r != 4) bb = __builtin_fma(-J[12 + r], we2, bb);
                acc[21 + r] = bb;
#pragma unroll
                for (int c2 = r; c2 < 6; ++c2, ++k) {
                    double t = acc[k];
                    if (r != 4 && c2 != 4) t = __builtin_fma(j0, J[c2], t);
                    if (r != 3 && c2 != 3) t = __builtin_fma(j1, J[6 + c2], t);
                    if (r != 4 && c2 != 4) t = __builtin_fma(j2, J[12 + c2], t);
                    acc[k] = t;
                }
            }
        }
        }
        PO_T(const long long p1 = __builtin_amdgcn_s_memtime();)
        block_sum28<NT>(acc, sh, fin);
        PO_T(const long long p2 = __builtin_amdgcn_s_memtime(); t_edges += p1 - p0; t_sum += p2 - p1; ++n_pass;)
    };
    int nBad = 0, cur = 0;
    double *fin0 = sh + (NT / 64) * kPoSum;   // two result buffers of the sums
    double xs[6] = {0, 0, 0, 0, 0, 0};   // the solver's x: persists over trials and rounds
    for (int it = 0; it < 4; ++it) {
#pragma unroll
        for (int k = 0; k < 7; ++k) qt[k] = P.pose_in[k];  // every round restarts from pFrame->mTcw (:368)
        const int n_active = n - nBad;   // the edges not at level 1
        if (n_active > 0) {
            int nBadLM = 0;
            bool ok = true, have = false;   // have: Hb / currentChi / the stored chi2 belong to the current pose
            double lambda = 0, ni = 2, currentChi = 0;
            for (int i = 0; i < 10 && ok; ++i) {
                // ONE call site of the edge pass: the pass at the top of solve() (computeActiveErrors + buildSystem, levenberg.cpp:75-88
                // -- needed when the sums of the current pose are not there: the first iteration of a round, or after a rejected last
                // trial) runs as a preliminary turn of the trial loop
                bool init = !have;
                double iniChi = currentChi;
                auto begin_iter = [&]() {
                    iniChi = currentChi;
                    if (i == 0) {
                        const double *Hb = fin0 + cur * kPoSum;
                        double maxDiagonal = 0.;
                        constexpr int di[6] = {0, 6, 11, 15, 18, 20};
#pragma unroll
                        for (int d = 0; d < 6; ++d) maxDiagonal = fmax(fabs(Hb[di[d]]), maxDiagonal);
                        lambda = 1e-5 * maxDiagonal;
                        ni = 2;
                        nBadLM = 0;
                    }
                };
                if (!init) begin_iter();
                double rho = 0, scale = 0;
                int qmax = 0;
                bool pos = true;
                double bk[7];
                do {
                    double *dst = fin0 + cur * kPoSum;
                    PO_T(const long long s0 = __builtin_amdgcn_s_memtime();)
                    if (!init) {
#pragma unroll
                        for (int k = 0; k < 7; ++k) bk[k] = qt[k];
                        const double *Hb = fin0 + cur * kPoSum;
                        pos = solve6(Hb, lambda, xs);
                        se3_oplus_fast(xs, qt);   // (after a failed solve: the previous x once more; the trial is rejected below)
                        scale = 0.;
#pragma unroll
                        for (int j = 0; j < 6; ++j) scale += xs[j] * (lambda * xs[j] + Hb[21 + j]);
                        scale += 1e-3;
                        dst = fin0 + (cur ^ 1) * kPoSum;   // the trial's sums go to the other buffer: accepting = switching
                        PO_T(++n_trial;)
                    }
                    PO_T(const long long s1 = __builtin_amdgcn_s_memtime(); t_solve += s1 - s0;)
                    pass(qt, dst);
                    if (init) {
                        currentChi = dst[27];
                        begin_iter();
                        init = false;
                        rho = -1;   // (stay in the loop: the first trial follows)
                        continue;
                    }
                    PO_T(const long long s2 = __builtin_amdgcn_s_memtime();)
                    const double tempChi = pos ? dst[27] : 1.7976931348623157e308;
                    double r = currentChi - tempChi;
                    r /= scale;
                    have = false;
                    if (r > 0 && isfinite(tempChi)) {
                        const double t = 2 * r - 1;
                        double alpha = 1. - t * t * t;
                        alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
                        const double scaleFactor = 1. / 3. > alpha ? 1. / 3. : alpha;
                        lambda *= scaleFactor;
                        ni = 2;
                        currentChi = tempChi;
                        cur ^= 1;
                        have = true;
                    } else {
                        lambda *= ni;
                        ni *= 2;
#pragma unroll
                        for (int k = 0; k < 7; ++k) qt[k] = bk[k];
                    }
                    rho = r;
                    qmax++;
                    PO_T(t_dec += __builtin_amdgcn_s_memtime() - s2;)
                } while (rho < 0 && qmax < 10);
                if (qmax == 10 || rho == 0) {
                    ok = false;
                } else {
                    if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
                    if (nBadLM >= 3) ok = false;
                }
            }
        }
        // outlier reclassification (:371-430)
        if (tid == 0) *s_cnt = 0;
        __syncthreads();
        int bad = 0;
#pragma unroll EPT
        PO_FOR_EDGES(j, e) {
            const int st = Sv(j, e);
            if (Ou(j, e)) {   // e->computeError() for the edges the optimiser skipped (:379, :406)
                double er[3];
                po_edge_error(qt, Xp(j, e), Op(j, e), st, P, er);
                Cp(j, e) = edge_chi2(er, Wv(j, e), st ? 3 : 2);
            }
            const float chi2 = (float)Cp(j, e);
            if (chi2 > (st ? 7.815f : 5.991f)) {
                Ou(j, e) = 1;
                L1(j, e) = 1;
                bad++;
            } else {
                Ou(j, e) = 0;
                L1(j, e) = 0;
            }
            if (it == 2) Rb(j, e) = 0;
        }
        if (bad) atomicAdd(s_cnt, bad);
        __syncthreads();
        nBad = *s_cnt;
        __syncthreads();
        if (n < 10) break;  // optimizer.edges().size() < 10
    }
    if (kReg) {
#pragma unroll EPT
        PO_FOR_EDGES(j, e) P.outlier[e] = Outr[j];
    }
    PO_T(if (tid == 0 && blockIdx.x == 0) printf("PO n %d passes %d trials %d cycles: edges %lld sum %lld solve+oplus %lld decide %lld total %lld\n", n, n_pass, n_trial, t_edges, t_sum, t_solve, t_dec, __builtin_amdgcn_s_memtime() - t_all);)
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) P.pose_out[k] = qt[k];
        P.counts[0] = nBad;
        P.counts[1] = n - nBad;
    }
#undef PO_FOR_EDGES
}

template <int kEpt>
__global__ __launch_bounds__(256) void pose_optimization_kernel(const PoseProbDev *__restrict__ probs)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];  // kPoLds
    __shared__ int s_cnt;
    const PoseProbDev P = probs[blockIdx.x];
    pose_optimization_body<kEpt>(P, sh, &s_cnt);
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
        float *xw = H.push_fill<float>(3 * n + 1, offs[i].xw), *ob = H.push_fill<float>(3 * n + 1, offs[i].obs);
        float *w = H.push_fill<float>(n + 1, offs[i].w);
        uint8_t *sv = H.push_fill<uint8_t>(n + 1, offs[i].st);
        if (!xw || !ob || !w || !sv) {
            set_error("internal: pose optimisation input staging");
            return AOS2_ERR_ARG;
        }
        if (n) {
            memcpy(xw, p.Xw, 12 * n);
            memcpy(ob, p.obs, 12 * n);
            memcpy(w, p.inv_sigma2, 4 * n);
            memcpy(sv, p.stereo, n);
        }
        xw[3 * n] = ob[3 * n] = w[n] = 0.f;
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
        offs[i].err = H.push(nullptr, (n + 1) * 8);
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
        D.Xw = (const float *)(base + offs[i].xw); D.obs = (const float *)(base + offs[i].obs);
        D.w = (const float *)(base + offs[i].w); D.stereo = base + offs[i].st;
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
    const size_t po_lds = kPoLds;
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


__global__ void debug_pose_blocks_kernel(const double *upd, const double *T, double *T_out, const double *Hb, const double *lambda,
                                        double *x, uint8_t *ok, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double u[6], t[7], xv[6], hb[27];
    for (int k = 0; k < 6; ++k) u[k] = upd[6 * i + k];
    for (int k = 0; k < 7; ++k) t[k] = T[7 * i + k];
    se3_oplus_fast(u, t);
    for (int k = 0; k < 7; ++k) T_out[7 * i + k] = t[k];
    for (int k = 0; k < 27; ++k) hb[k] = Hb[27 * i + k];
    for (int k = 0; k < 6; ++k) xv[k] = x[6 * i + k];
    ok[i] = solve6(hb, lambda[i], xv) ? 1 : 0;
    for (int k = 0; k < 6; ++k) x[6 * i + k] = xv[k];
}

int aos2_debug_pose_blocks_device(const double *upd, const double *T, double *T_out, const double *Hb, const double *lambda,
                                  double *x, uint8_t *ok, int n, int device)
{
    int st;
    if ((st = bind_device(device))) return st;
    DevBuf<double> d;
    DevBuf<uint8_t> dk;
    const size_t N = (size_t)n;
    if ((st = d.alloc(N * (6 + 7 + 7 + 27 + 1 + 6))) || (st = dk.alloc(N))) return st;
    double *du = d.p, *dT = du + 6 * N, *dTo = dT + 7 * N, *dH = dTo + 7 * N, *dl = dH + 27 * N, *dx = dl + N;
    AOS2_HIP_CHECK(hipMemcpy(du, upd, 48 * N, hipMemcpyHostToDevice));
    AOS2_HIP_CHECK(hipMemcpy(dT, T, 56 * N, hipMemcpyHostToDevice));
    AOS2_HIP_CHECK(hipMemcpy(dH, Hb, 216 * N, hipMemcpyHostToDevice));
    AOS2_HIP_CHECK(hipMemcpy(dl, lambda, 8 * N, hipMemcpyHostToDevice));
    AOS2_HIP_CHECK(hipMemcpy(dx, x, 48 * N, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(debug_pose_blocks_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, du, dT, dTo, dH, dl, dx, dk.p, n);
    AOS2_HIP_CHECK(hipDeviceSynchronize());
    AOS2_HIP_CHECK(hipMemcpy(T_out, dTo, 56 * N, hipMemcpyDeviceToHost));
    AOS2_HIP_CHECK(hipMemcpy(x, dx, 48 * N, hipMemcpyDeviceToHost));
    AOS2_HIP_CHECK(hipMemcpy(ok, dk.p, N, hipMemcpyDeviceToHost));
    d.release();
    dk.release();
    return AOS2_OK;
}

float aos2_pose_optimization_last_device_ms(const aos2_lba_t *s) { return s ? s->last_pose_ms : 0.f; }

}  // extern "C"

#include "frames_pose.inc"
