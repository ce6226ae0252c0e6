// Shared pieces of the two Optimizer translation units (lba.hip: LocalBundleAdjustment, pose_opt.hip:
// PoseOptimization): SE3 / quaternion arithmetic exactly as g2o + Eigen perform it (se3quat.h, Eigen Quaternion),
// the Huber kernel, the float32 <-> SE3Quat converters of src/Converter.cc, and the handle both entry points share.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <vector>

#include "aos2_common.h"

namespace aos2 {

// ------------------------------------------------------------------------------------------ math
__host__ __device__ inline void quat_from_rot(const double m[9], double q[4])
{
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t;
        q[1] = (m[2] - m[6]) * t;
        q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
}

__host__ __device__ inline void rot_from_quat(const double q[4], double R[9])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

__host__ __device__ inline void quat_normalize_rot(double q[4])
{
    if (q[3] < 0)
        for (int i = 0; i < 4; ++i) q[i] *= -1;
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
}

__host__ __device__ inline void quat_rotate(const double q[4], const double v[3], double out[3])
{
    double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    const double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
    for (int i = 0; i < 3; ++i) out[i] = v[i] + q[3] * uv[i] + c[i];
}

__host__ __device__ inline void se3_map(const double qt[7], const double X[3], double out[3])
{
    double r[3];
    quat_rotate(qt, X, r);
    for (int i = 0; i < 3; ++i) out[i] = r[i] + qt[4 + i];
}

// T <- exp(upd) * T  (VertexSE3Expmap::oplusImpl, SE3Quat::exp se3quat.h:223-257, operator* :104-110)
__device__ inline void se3_oplus(const double upd[6], double T[7])
{
    const double *omega = upd, *ups = upd + 3;
    const double theta = sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
    const double Om[9] = {0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0};
    double Om2[9], R[9], V[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Om2[i * 3 + j] = Om[i * 3] * Om[j] + Om[i * 3 + 1] * Om[3 + j] + Om[i * 3 + 2] * Om[6 + j];
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) {
            R[i] = I[i] + Om[i] + Om2[i];
            V[i] = R[i];
        }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta);
        const double c = (theta - sin(theta)) / (theta * theta * theta);
        for (int i = 0; i < 9; ++i) {
            R[i] = I[i] + a * Om[i] + b * Om2[i];
            V[i] = I[i] + b * Om[i] + c * Om2[i];
        }
    }
    double e[7];
    quat_from_rot(R, e);
    quat_normalize_rot(e);
    for (int i = 0; i < 3; ++i) e[4 + i] = V[i * 3] * ups[0] + V[i * 3 + 1] * ups[1] + V[i * 3 + 2] * ups[2];
    // e * T
    double rt[3], q[4];
    quat_rotate(e, T + 4, rt);
    q[3] = e[3] * T[3] - e[0] * T[0] - e[1] * T[1] - e[2] * T[2];
    q[0] = e[3] * T[0] + e[0] * T[3] + e[1] * T[2] - e[2] * T[1];
    q[1] = e[3] * T[1] + e[1] * T[3] + e[2] * T[0] - e[0] * T[2];
    q[2] = e[3] * T[2] + e[2] * T[3] + e[0] * T[1] - e[1] * T[0];
    quat_normalize_rot(q);
    for (int i = 0; i < 4; ++i) T[i] = q[i];
    for (int i = 0; i < 3; ++i) T[4 + i] = e[4 + i] + rt[i];
}

// sqrt(d) and 1 / sqrt(d) for d > 0: v_rsq_f64 + two coupled Newton (Goldschmidt) steps + one correction of the root;
// ~10 dependent operations instead of the ~30-instruction IEEE sqrt followed by a ~30-instruction IEEE division.
__device__ __forceinline__ void sqrt_rsqrt(double d, double &s, double &r)
{
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double e = __builtin_fma(-g, h, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    e = __builtin_fma(-g, h, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    const double c = __builtin_fma(-g, g, d);
    s = __builtin_fma(c, h, g);
    r = h + h;
}

// T <- exp(upd) * T: VertexSE3Expmap::oplusImpl like se3_oplus (lba_math.h: SE3Quat::exp se3quat.h:223-257, operator*
// :104-110), shaped for the serial path of the pose solver.  An LM step is a small rotation (|omega|^2 < 0.6), and for
// those nothing needs a square root, a division or a sin / cos call: with z = |omega|^2 and fdlibm's minimax polynomials
// sin t = t + t z ps(z), cos t = 1 - z / 2 + z^2 pc(z) on |t| <= pi / 4,
//     sin t / t = 1 + z ps,   (1 - cos t) / t^2 = 1/2 - z pc,   (t - sin t) / t^3 = -ps       (the coefficients of R, V)
// without the cancellation the quotients of se3quat.h:236-240 suffer, and the quaternion of R(omega) is
// (omega sin(t/2) / t, cos(t/2)) = (omega (1 + zh ps(zh)) / 2, 1 - zh / 2 + zh^2 pc(zh)), zh = z / 4 -- what Eigen's
// Quaterniond(R) + normalize() extracts from the matrix, up to rounding.  Below theta = 1e-5 the reference switches to
// R = V = I + Omega + Omega^2 (:231-234); V keeps that, the normalised quaternion of that R equals the exact one to
// 2e-11 relative.  The quaternion product is renormalised by a Newton step from 1 (|q|^2 = 1 + O(1e-16)).  Larger
// rotations take the formulas as written.  Results agree with se3_oplus to a few ulp (pose tolerance: 1e-5).
__device__ __forceinline__ void se3_oplus_fast(const double upd[6], double T[7])
{
    const double *omega = upd, *ups = upd + 3;
    const double th2 = omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2];
    // a x b with fused multiply-adds
    auto cross = [](const double *a, const double *b, double *o) {
        o[0] = __builtin_fma(a[1], b[2], -(a[2] * b[1]));
        o[1] = __builtin_fma(a[2], b[0], -(a[0] * b[2]));
        o[2] = __builtin_fma(a[0], b[1], -(a[1] * b[0]));
    };
    auto poly_s = [](double z) {
        double ps = 1.58969099521155010221e-10;
        ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
        ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
        ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
        ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
        return __builtin_fma(ps, z, -1.66666666666666324348e-01);
    };
    auto poly_c = [](double z) {
        double pc = -1.13596475577881948265e-11;
        pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
        pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
        pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
        pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
        return __builtin_fma(pc, z, 4.16666666666666019037e-02);
    };
    double e[7], bV, cV;
    if (th2 < 0.6) {
        const double z = th2, zh = 0.25 * th2;
        const double ps = poly_s(z), pc = poly_c(z), psh = poly_s(zh), pch = poly_c(zh);
        const bool tiny = th2 < 0.00001 * 0.00001;
        bV = tiny ? 1.0 : __builtin_fma(-z, pc, 0.5);
        cV = tiny ? 1.0 : -ps;
        const double sv = 0.5 * __builtin_fma(zh, psh, 1.0);
        e[0] = omega[0] * sv;
        e[1] = omega[1] * sv;
        e[2] = omega[2] * sv;
        e[3] = __builtin_fma(zh * zh, pch, __builtin_fma(-0.5, zh, 1.0));
    } else {
        double theta, ith, sn, cs, R[9];
        sqrt_rsqrt(th2, theta, ith);
        sincos(theta, &sn, &cs);
        const double ith2 = ith * ith;
        const double a = sn * ith;
        bV = (1 - cs) * ith2;
        cV = (theta - sn) * (ith2 * ith);
        const double Om[9] = {0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0};
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double om2 = Om[i * 3] * Om[j] + Om[i * 3 + 1] * Om[3 + j] + Om[i * 3 + 2] * Om[6 + j];
                R[i * 3 + j] = I[i * 3 + j] + a * Om[i * 3 + j] + bV * om2;
            }
        const double t = R[0] + R[4] + R[8];
        if (t > 0) {   // Eigen's Quaterniond(R), trace branch
            double sq, rs;
            sqrt_rsqrt(t + 1.0, sq, rs);
            e[3] = 0.5 * sq;
            const double tt = 0.5 * rs;
            e[0] = (R[7] - R[5]) * tt;
            e[1] = (R[2] - R[6]) * tt;
            e[2] = (R[3] - R[1]) * tt;
        } else {   // the other three branches, with static indices (a dynamic one would put R and e into scratch memory)
            auto branch = [&](auto ic) {
                constexpr int i = decltype(ic)::value, j = (i + 1) % 3, k = (j + 1) % 3;
                double sq, rs;
                sqrt_rsqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0, sq, rs);
                e[i] = 0.5 * sq;
                const double tt = 0.5 * rs;
                e[3] = (R[k * 3 + j] - R[j * 3 + k]) * tt;
                e[j] = (R[j * 3 + i] + R[i * 3 + j]) * tt;
                e[k] = (R[k * 3 + i] + R[i * 3 + k]) * tt;
            };
            if (R[8] > (R[4] > R[0] ? R[4] : R[0]))
                branch(std::integral_constant<int, 2>());
            else if (R[4] > R[0])
                branch(std::integral_constant<int, 1>());
            else
                branch(std::integral_constant<int, 0>());
        }
        const double sg = e[3] < 0 ? -1.0 : 1.0;
        double sq, rs;
        sqrt_rsqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3], sq, rs);
        rs *= sg;
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] *= rs;
    }
    // V upsilon = upsilon + bV (omega x upsilon) + cV (omega x (omega x upsilon))      (V = I + bV Omega + cV Omega^2)
    double wu[3], wwu[3];
    cross(omega, ups, wu);
    cross(omega, wu, wwu);
#pragma unroll
    for (int i = 0; i < 3; ++i) e[4 + i] = __builtin_fma(cV, wwu[i], __builtin_fma(bV, wu[i], ups[i]));
    // e * T: rotation of T's translation (v + w uv + e_v x uv, uv = 2 e_v x v) and the quaternion product
    double rt[3], q[4], uv[3], cc[3];
    cross(e, T + 4, uv);
#pragma unroll
    for (int i = 0; i < 3; ++i) uv[i] += uv[i];
    cross(e, uv, cc);
#pragma unroll
    for (int i = 0; i < 3; ++i) rt[i] = __builtin_fma(e[3], uv[i], T[4 + i]) + cc[i];
    q[3] = __builtin_fma(-e[2], T[2], __builtin_fma(-e[1], T[1], __builtin_fma(-e[0], T[0], e[3] * T[3])));
    q[0] = __builtin_fma(-e[2], T[1], __builtin_fma(e[1], T[2], __builtin_fma(e[0], T[3], e[3] * T[0])));
    q[1] = __builtin_fma(-e[0], T[2], __builtin_fma(e[2], T[0], __builtin_fma(e[1], T[3], e[3] * T[1])));
    q[2] = __builtin_fma(-e[1], T[0], __builtin_fma(e[0], T[1], __builtin_fma(e[2], T[3], e[3] * T[2])));
    {
        const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
        double rs;
        if (fabs(n2 - 1.0) < 1e-6) {   // 1 / sqrt(1 + eps): two Newton steps from 1 (error O(eps^4))
            const double r0 = __builtin_fma(-0.5, n2, 1.5);
            rs = r0 * __builtin_fma(-0.5 * n2, r0 * r0, 1.5);
        } else {
            double sq;
            sqrt_rsqrt(n2, sq, rs);
        }
        if (q[3] < 0) rs = -rs;
#pragma unroll
        for (int i = 0; i < 4; ++i) T[i] = q[i] * rs;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) T[4 + i] = e[4 + i] + rt[i];
}

__device__ inline void mat3_inverse(const double m[9], double inv[9])
{
    const double c00 = m[4] * m[8] - m[5] * m[7];
    const double c10 = m[5] * m[6] - m[3] * m[8];
    const double c20 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c10 + m[2] * c20;
    const double id = 1.0 / det;
    inv[0] = c00 * id;
    inv[1] = (m[2] * m[7] - m[1] * m[8]) * id;
    inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    inv[3] = c10 * id;
    inv[4] = (m[0] * m[8] - m[2] * m[6]) * id;
    inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    inv[6] = c20 * id;
    inv[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

__device__ inline double edge_chi2(const double *er, double w, int D)
{
    double s = 0;
    for (int i = 0; i < D; ++i) s += er[i] * (w * er[i]);
    return s;
}

__device__ inline void robustify(double e, double delta, double rho[2])
{
    const double dsqr = delta * delta;
    if (e <= dsqr) {
        rho[0] = e;
        rho[1] = 1.;
    } else {
        const double sqrte = sqrt(e);
        rho[0] = 2 * sqrte * delta - dsqr;
        rho[1] = delta / sqrte;
    }
}

// Converter::toSE3Quat (src/Converter.cc:37-47): float32 4x4 Tcw -> SE3Quat (Eigen Quaterniond(R), normalised)
__host__ __device__ inline void pose_from_Tcw(const float *T, double qt[7])
{
    double R[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i * 3 + j] = (double)T[i * 4 + j];
    quat_from_rot(R, qt);
    quat_normalize_rot(qt);
    for (int i = 0; i < 3; ++i) qt[4 + i] = (double)T[i * 4 + 3];
}

// Converter::toCvMat(SE3Quat) (src/Converter.cc:49-71)
__host__ __device__ inline void pose_to_Tcw(const double qt[7], float *T)
{
    double R[9];
    rot_from_quat(qt, R);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[i * 4 + j] = (float)R[i * 3 + j];
        T[i * 4 + 3] = (float)qt[4 + i];
    }
    T[12] = T[13] = T[14] = 0.f;
    T[15] = 1.f;
}

// staged inputs of a call: a prefix of the device arena assembled in page-locked host memory
struct HostArena {
    uint8_t *host = nullptr;
    size_t host_cap = 0, host_size = 0;
    std::vector<uint8_t> own;
    size_t size = 0;            // total arena size including device-only scratch
    const uint8_t *data() const { return host ? host : own.data(); }
    size_t push(const void *src, size_t bytes)
    {
        const size_t off = (size + 255) & ~(size_t)255;
        size = off + bytes;
        if (src && bytes) {  // inputs are pushed before any scratch, so the staged part stays a prefix
            if (!host) {
                own.resize(size);
                memcpy(own.data() + off, src, bytes);
            } else if (size <= host_cap)
                memcpy(host + off, src, bytes);
            host_size = size;
        }
        return off;
    }
    // input produced in place (conversions): returns where to write it
    template <class T>
    T *push_fill(size_t count, size_t &off)
    {
        off = (size + 255) & ~(size_t)255;
        size = off + count * sizeof(T);
        host_size = size;
        return size <= host_cap ? reinterpret_cast<T *>(host + off) : nullptr;
    }
};

}  // namespace aos2

// Handle shared by aos2_lba_solve* (lba.hip) and aos2_pose_optimization (pose_opt.hip): device, stream, arenas.
struct aos2_lba {
    int device;
    bool dev_ready = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev[3] = {};   // [0], [1]: device time of a call; [2]: spare
    hipStream_t stream2 = nullptr;      // LocalBA: the LDS form of the reduced-system kernel runs here beside the device-memory form
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // LocalBA: a batch of many windows runs as TWO groups whose programs are staggered on their own streams (lba.hip, solve_batch)
    hipStream_t stream_b = nullptr, stream2_b = nullptr;
    hipEvent_t ev_fork_b = nullptr, ev_join_b = nullptr, ev_up = nullptr, ev_stag = nullptr, ev_done_b = nullptr;
    int last_groups = 1;
    int window_groups = 0;              // LocalBA: 0 = default (two groups for batches of >= 16 windows), 1 / 2 (aos2_lba_set_window_groups)
    aos2::DevBuf<uint8_t> arena;
    aos2::PinnedBuf<uint8_t> h_stage;   // results on their way back
    aos2::PinnedBuf<uint8_t> h_in;      // staged inputs (the arena's prefix)
    aos2::PinnedBuf<int32_t> h_abort;   // LocalBA: per-window abort words the kernels poll (mapped host memory)
    float last_pose_ms = 0;
    int debug_stop_at_poll = 0;         // test hook: treat pbStopFlag as set from this poll on (0 = off)
    void *lba_cache = nullptr;          // LocalBA: host-side structure buffers kept between calls (lba.hip: LbaCache)
    int host_threads = 0;               // LocalBA: worker threads of the per-window host work (0 = default, aos2_lba_set_host_threads)
    int last_trial_slots = 0, last_host_rounds = 0;   // the device program of the last solve (aos2_lba_last_program)
    long long last_window_slots = 0;                  // ... and its trial slots summed over the windows each round covered
};

namespace aos2 {
// binds the device, creates the stream / events on first use (lba.hip)
int lba_handle_init(aos2_lba *s);
}  // namespace aos2
