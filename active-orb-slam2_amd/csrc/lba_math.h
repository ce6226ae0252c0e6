// Shared pieces of the two Optimizer translation units (lba.hip: LocalBundleAdjustment, pose_opt.hip:
// PoseOptimization): SE3 / quaternion arithmetic exactly as g2o + Eigen perform it (se3quat.h, Eigen Quaternion),
// the Huber kernel, the float32 <-> SE3Quat converters of src/Converter.cc, and the handle both entry points share.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
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
    aos2::DevBuf<uint8_t> arena;
    aos2::PinnedBuf<uint8_t> h_stage;   // results on their way back
    aos2::PinnedBuf<uint8_t> h_in;      // staged inputs (the arena's prefix)
    aos2::PinnedBuf<int32_t> h_abort;   // LocalBA: per-window abort words the kernels poll (mapped host memory)
    float last_pose_ms = 0;
    int debug_stop_at_poll = 0;         // test hook: treat pbStopFlag as set from this poll on (0 = off)
};

namespace aos2 {
// binds the device, creates the stream / events on first use (lba.hip)
int lba_handle_init(aos2_lba *s);
}  // namespace aos2
