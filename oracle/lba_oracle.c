/*
 * ORACLE (test infrastructure only; PARITY UNPINNED, see orb_oracle.h).
 *
 * CPU restatement of the numerical core of Optimizer::LocalBundleAdjustment
 * (reference src/Optimizer.cc:454-779) as executed by the vendored g2o:
 *   SparseOptimizer::{initializeOptimization,optimize}   Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:206-267,354-419
 *   OptimizationAlgorithmLevenberg::solve                core/optimization_algorithm_levenberg.cpp:61-189
 *   BlockSolver<6,3>::{buildSystem,setLambda,solve,restoreDiagonal}  core/block_solver.hpp:367-604
 *   BaseBinaryEdge::constructQuadraticForm               core/base_binary_edge.hpp:55-120
 *   RobustKernelHuber::robustify                         core/robust_kernel_impl.cpp:78-91
 *   Edge{,Stereo}SE3ProjectXYZ                           types/types_six_dof_expmap.{h,cpp}
 *   SE3Quat                                              types/se3quat.h
 * Eigen (un-vendored, >=3.1) fixed-size kernels (Quaternion<->matrix, 3x3 inverse, LDLT) are
 * restated from their published algorithms.  The reduced system is solved by a dense LDL^T
 * instead of Eigen::SimplicialLDLT with AMD ordering (same factorisation up to rounding).
 * Edge order = order of the edge array (DESIGN.md parity convention 2).
 */
#include "orb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Measurement switches (tools/convention_effects.py; never set by a parity test's reference run): how far do the results move
 * when the arithmetic is re-associated at rounding level?  bit 0: the reduced system is eliminated in the opposite order (last
 * unknown first: another admissible ordering of linear_solver_eigen.h:94-124's factorisation); bit 1: every accumulation across
 * edges / landmarks / pivots (H blocks and b in buildSystem, the Schur products, the factorisation's dot products) is carried in
 * long double and rounded once.  0 = the oracle as tested. */
static _Thread_local int g_lba_variant = 0;   /* per THREAD: the checkers run oracle solves of different variants side by side (thread pools of
                                                 * bench.py / tools), and a setter call and the solve it configures happen on one thread */
void orc_set_lba_variant(int bits) { g_lba_variant = bits; }
#define LBA_REVERSE_ELIMINATION 1
#define LBA_EXTENDED_SUMS 2

/* ------------------------------------------------------------------ small linear algebra */
void orc_quat_from_rot(const double m[9], double q[4]) /* q = x y z w ; Eigen Quaternion(Matrix3) */
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
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
}

void orc_rot_from_quat(const double q[4], double R[9])
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

static void quat_normalize_rot(double q[4]) /* SE3Quat::normalizeRotation se3quat.h:280-285 */
{
    if (q[3] < 0)
        for (int i = 0; i < 4; ++i) q[i] *= -1;
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
}

static void quat_rotate(const double q[4], const double v[3], double out[3])
{
    /* Eigen QuaternionBase::_transformVector */
    double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
    for (int i = 0; i < 3; ++i) out[i] = v[i] + q[3] * uv[i] + c[i];
}

static void quat_mul(const double a[4], const double b[4], double o[4])
{
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}

/* SE3Quat::operator* se3quat.h:104-110 ; qt = qx qy qz qw tx ty tz */
void orc_se3_mul(const double a[7], const double b[7], double out[7])
{
    double rt[3], q[4];
    quat_rotate(a, b + 4, rt);
    quat_mul(a, b, q);
    quat_normalize_rot(q);
    for (int i = 0; i < 4; ++i) out[i] = q[i];
    for (int i = 0; i < 3; ++i) out[4 + i] = a[4 + i] + rt[i];
}

static void mat3_mul(const double A[9], const double B[9], double C[9])
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

/* SE3Quat::exp se3quat.h:223-257 ; update = (omega, upsilon) */
void orc_se3_exp(const double upd[6], double qt[7])
{
    const double *omega = upd, *upsilon = upd + 3;
    double theta = sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
    double Om[9] = {0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0};
    double Om2[9], R[9], V[9];
    mat3_mul(Om, Om, Om2);
    static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) R[i] = I[i] + Om[i] + Om2[i];
        memcpy(V, R, sizeof(R));
    } else {
        double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta);
        double c = (theta - sin(theta)) / pow(theta, 3);
        for (int i = 0; i < 9; ++i) {
            R[i] = I[i] + a * Om[i] + b * Om2[i];
            V[i] = I[i] + b * Om[i] + c * Om2[i];
        }
    }
    double q[4];
    orc_quat_from_rot(R, q);
    quat_normalize_rot(q);
    for (int i = 0; i < 4; ++i) qt[i] = q[i];
    for (int i = 0; i < 3; ++i) qt[4 + i] = V[i * 3] * upsilon[0] + V[i * 3 + 1] * upsilon[1] + V[i * 3 + 2] * upsilon[2];
}

static void se3_map(const double qt[7], const double X[3], double out[3])
{
    double r[3];
    quat_rotate(qt, X, r);
    for (int i = 0; i < 3; ++i) out[i] = r[i] + qt[4 + i];
}

/* Eigen 3x3 inverse (cofactors / determinant) */
static void mat3_inverse(const double m[9], double inv[9])
{
    double c00 = m[4] * m[8] - m[5] * m[7];
    double c10 = m[5] * m[6] - m[3] * m[8]; /* cofactor(1,0) wrt column 0 expansion */
    double c20 = m[3] * m[7] - m[4] * m[6];
    double det = m[0] * c00 + m[1] * c10 + m[2] * c20;
    double id = 1.0 / det;
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

/* ------------------------------------------------------------------ edge model */
static void edge_error(const double qt[7], const double X[3], const double obs[3], int stereo,
                       double fx, double fy, double cx, double cy, double bf, double err[3])
{
    double p[3];
    se3_map(qt, X, p);
    if (!stereo) { /* EdgeSE3ProjectXYZ::computeError / cam_project .cpp:141-147 */
        double u = p[0] / p[2], v = p[1] / p[2];
        err[0] = obs[0] - (u * fx + cx);
        err[1] = obs[1] - (v * fy + cy);
        err[2] = 0;
    } else { /* EdgeStereoSE3ProjectXYZ::cam_project .cpp:150-157 (float invz!) */
        const float invz = (float)(1.0f / p[2]);
        double r0 = p[0] * invz * fx + cx;
        double r1 = p[1] * invz * fy + cy;
        /* bf is passed as `const float&` (.cpp:150): float * float product */
        double r2 = r0 - (double)((float)bf * invz);
        err[0] = obs[0] - r0;
        err[1] = obs[1] - r1;
        err[2] = obs[2] - r2;
    }
}

static void edge_jacobians(const double qt[7], const double X[3], int stereo, double fx, double fy,
                           double bf, double Ji[9], double Jj[18])
{
    double p[3], R[9];
    se3_map(qt, X, p);
    orc_rot_from_quat(qt, R);
    double x = p[0], y = p[1], z = p[2], z_2 = z * z;
    memset(Ji, 0, sizeof(double) * 9);
    memset(Jj, 0, sizeof(double) * 18);
    if (!stereo) { /* .cpp:103-139 */
        double tmp[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
        double s = -1. / z;
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 3; ++c) {
                /* (-1./z * tmp) * R, evaluated left to right like the Eigen expression */
                double a0 = s * tmp[r * 3], a1 = s * tmp[r * 3 + 1], a2 = s * tmp[r * 3 + 2];
                Ji[r * 3 + c] = a0 * R[c] + a1 * R[3 + c] + a2 * R[6 + c];
            }
    } else { /* .cpp:188-234 */
        for (int c = 0; c < 3; ++c) {
            Ji[0 * 3 + c] = -fx * R[0 * 3 + c] / z + fx * x * R[2 * 3 + c] / z_2;
            Ji[1 * 3 + c] = -fy * R[1 * 3 + c] / z + fy * y * R[2 * 3 + c] / z_2;
            Ji[2 * 3 + c] = Ji[0 * 3 + c] - bf * R[2 * 3 + c] / z_2;
        }
    }
    Jj[0] = x * y / z_2 * fx;
    Jj[1] = -(1 + (x * x / z_2)) * fx;
    Jj[2] = y / z * fx;
    Jj[3] = -1. / z * fx;
    Jj[4] = 0;
    Jj[5] = x / z_2 * fx;
    Jj[6] = (1 + y * y / z_2) * fy;
    Jj[7] = -x * y / z_2 * fy;
    Jj[8] = -x / z * fy;
    Jj[9] = 0;
    Jj[10] = -1. / z * fy;
    Jj[11] = y / z_2 * fy;
    if (stereo) {
        Jj[12] = Jj[0] - bf * y / z_2;
        Jj[13] = Jj[1] + bf * x / z_2;
        Jj[14] = Jj[2];
        Jj[15] = Jj[3];
        Jj[16] = 0;
        Jj[17] = Jj[5] - bf / z_2;
    }
}

void orc_edge_linearize(const double pose_qt[7], const double xyz[3], const double obs[3],
                        int stereo, double fx, double fy, double cx, double cy, double bf,
                        double err[3], double Ji[9], double Jj[18])
{
    edge_error(pose_qt, xyz, obs, stereo, fx, fy, cx, cy, bf, err);
    edge_jacobians(pose_qt, xyz, stereo, fx, fy, bf, Ji, Jj);
}

/* ------------------------------------------------------------------ solver state */
typedef struct {
    orc_lba_problem_t *p;
    /* per edge persistent state */
    double *err;        /* n_edges x 3, last computed _error */
    uint8_t *level1;    /* setLevel(1) */
    uint8_t *robust;    /* robust kernel present */
    /* active structure */
    int n_act_e;
    int *act_e;         /* active edge ids (insertion order) */
    int np, nl;         /* free active poses, active points */
    int *pose_hidx;     /* per pose: hessian index or -1 */
    int *point_hidx;    /* per point: hessian index (0..nl-1) or -1 */
    int *hpose, *hpoint; /* inverse maps */
    /* system */
    double *Hpp;        /* np x 36 (diagonal blocks) */
    double *Hll;        /* nl x 9 */
    double *Hpl;        /* n_act_e x 18 (6x3 row major), zero if pose fixed */
    double *b;          /* 6np + 3nl */
    double *x;          /* 6np + 3nl */
    double *Hs;         /* (6np)^2 dense */
    double *bs, *coeff;
    double *Dinv;       /* nl x 9 */
    /* point -> active edges with free pose (CSR, sorted by pose hidx) */
    int *pl_off, *pl_edge;
    /* backup */
    double *bk_pose, *bk_point, *bk_dpp, *bk_dll;
    double lambda, ni;
    int nBad;
    int polls, stop_poll, trials; /* evaluations of terminate(); the one that first saw the flag; LM trial steps */
} lba_t;

static double edge_chi2(const lba_t *S, int e)
{
    const double *er = S->err + 3 * (size_t)e;
    double w = (double)S->p->edge_inv_sigma2[e];
    int D = S->p->edge_stereo[e] ? 3 : 2;
    /* _error.dot(information()*_error), information = I*invSigma2 */
    double s = 0;
    for (int i = 0; i < D; ++i) s += er[i] * (w * er[i]);
    return s;
}

static double huber_delta(int stereo)
{
    /* const float thHuberMono = sqrt(5.991); setDelta(float->double)  Optimizer.cc:570-571 */
    return stereo ? (double)(float)sqrt(7.815) : (double)(float)sqrt(5.991);
}

static void robustify(double e, double delta, double rho[3])
{
    double dsqr = delta * delta;
    if (e <= dsqr) {
        rho[0] = e; rho[1] = 1.; rho[2] = 0.;
    } else {
        double sqrte = sqrt(e);
        rho[0] = 2 * sqrte * delta - dsqr;
        rho[1] = delta / sqrte;
        rho[2] = -0.5 * rho[1] / e;
    }
}

static void compute_active_errors(lba_t *S)
{
    orc_lba_problem_t *p = S->p;
    for (int k = 0; k < S->n_act_e; ++k) {
        int e = S->act_e[k];
        edge_error(p->pose_qt + 7 * (size_t)p->edge_pose[e], p->point_xyz + 3 * (size_t)p->edge_point[e],
                   p->edge_obs + 3 * (size_t)e, p->edge_stereo[e], p->fx, p->fy, p->cx, p->cy, p->bf,
                   S->err + 3 * (size_t)e);
    }
}

static double active_robust_chi2(const lba_t *S)
{
    double chi = 0.0;
    for (int k = 0; k < S->n_act_e; ++k) {
        int e = S->act_e[k];
        double c = edge_chi2(S, e);
        if (S->robust[e]) {
            double rho[3];
            robustify(c, huber_delta(S->p->edge_stereo[e]), rho);
            chi += rho[0];
        } else
            chi += c;
    }
    return chi;
}

typedef struct { int64_t id; int idx; } idpair_t;
static int idpair_cmp(const void *a, const void *b)
{
    const idpair_t *x = (const idpair_t *)a, *y = (const idpair_t *)b;
    return x->id < y->id ? -1 : x->id > y->id ? 1 : 0;
}
typedef struct { int hidx, edge; } pe_t;
static int pe_cmp(const void *a, const void *b)
{
    const pe_t *x = (const pe_t *)a, *y = (const pe_t *)b;
    if (x->hidx != y->hidx) return x->hidx < y->hidx ? -1 : 1;
    return x->edge < y->edge ? -1 : x->edge > y->edge ? 1 : 0;
}

static void free_structure(lba_t *S)
{
    free(S->act_e); free(S->hpose); free(S->hpoint);
    free(S->Hpp); free(S->Hll); free(S->Hpl); free(S->b); free(S->x); free(S->Hs);
    free(S->bs); free(S->coeff); free(S->Dinv); free(S->pl_off); free(S->pl_edge);
    free(S->bk_pose); free(S->bk_point); free(S->bk_dpp); free(S->bk_dll);
    S->act_e = S->hpose = S->hpoint = S->pl_off = S->pl_edge = NULL;
    S->Hpp = S->Hll = S->Hpl = S->b = S->x = S->Hs = S->bs = S->coeff = S->Dinv = NULL;
    S->bk_pose = S->bk_point = S->bk_dpp = S->bk_dll = NULL;
}

/* initializeOptimization(level 0) + buildIndexMapping + buildStructure */
static int initialize_optimization(lba_t *S)
{
    orc_lba_problem_t *p = S->p;
    free_structure(S);
    S->act_e = (int *)malloc(sizeof(int) * (p->n_edges ? p->n_edges : 1));
    S->n_act_e = 0;
    uint8_t *pose_act = (uint8_t *)calloc(p->n_poses ? p->n_poses : 1, 1);
    uint8_t *point_act = (uint8_t *)calloc(p->n_points ? p->n_points : 1, 1);
    for (int e = 0; e < p->n_edges; ++e) {
        if (S->level1[e]) continue;
        S->act_e[S->n_act_e++] = e;
        pose_act[p->edge_pose[e]] = 1;
        point_act[p->edge_point[e]] = 1;
    }
    if (S->n_act_e == 0) {
        free(pose_act); free(point_act);
        return -1;
    }
    /* index mapping: free poses by id, then points by id */
    idpair_t *ids = (idpair_t *)malloc(sizeof(idpair_t) * ((p->n_poses > p->n_points ? p->n_poses : p->n_points) + 1));
    int n = 0;
    for (int i = 0; i < p->n_poses; ++i) {
        S->pose_hidx[i] = -1;
        if (pose_act[i] && !p->pose_fixed[i]) { ids[n].id = p->pose_id[i]; ids[n].idx = i; n++; }
    }
    qsort(ids, n, sizeof(idpair_t), idpair_cmp);
    S->np = n;
    S->hpose = (int *)malloc(sizeof(int) * (n ? n : 1));
    for (int i = 0; i < n; ++i) { S->pose_hidx[ids[i].idx] = i; S->hpose[i] = ids[i].idx; }
    n = 0;
    for (int i = 0; i < p->n_points; ++i) {
        S->point_hidx[i] = -1;
        if (point_act[i]) { ids[n].id = p->point_id[i]; ids[n].idx = i; n++; }
    }
    qsort(ids, n, sizeof(idpair_t), idpair_cmp);
    S->nl = n;
    S->hpoint = (int *)malloc(sizeof(int) * (n ? n : 1));
    for (int i = 0; i < n; ++i) { S->point_hidx[ids[i].idx] = i; S->hpoint[i] = ids[i].idx; }
    free(ids); free(pose_act); free(point_act);

    int np = S->np, nl = S->nl;
    size_t dim = (size_t)6 * np + (size_t)3 * nl;
    S->Hpp = (double *)calloc((size_t)np * 36 + 1, sizeof(double));
    S->Hll = (double *)calloc((size_t)nl * 9 + 1, sizeof(double));
    S->Hpl = (double *)calloc((size_t)S->n_act_e * 18 + 1, sizeof(double));
    S->b = (double *)calloc(dim + 1, sizeof(double));
    S->x = (double *)calloc(dim + 1, sizeof(double));
    S->Hs = (double *)calloc((size_t)36 * np * np + 1, sizeof(double));
    S->bs = (double *)calloc((size_t)6 * np + 1, sizeof(double));
    S->coeff = (double *)calloc(dim + 1, sizeof(double));
    S->Dinv = (double *)calloc((size_t)nl * 9 + 1, sizeof(double));
    S->bk_pose = (double *)malloc(sizeof(double) * 7 * (np ? np : 1));
    S->bk_point = (double *)malloc(sizeof(double) * 3 * (nl ? nl : 1));
    S->bk_dpp = (double *)malloc(sizeof(double) * 6 * (np ? np : 1));
    S->bk_dll = (double *)malloc(sizeof(double) * 3 * (nl ? nl : 1));
    /* landmark columns of Hpl: active edges whose pose is free, ascending pose index */
    S->pl_off = (int *)calloc(nl + 1, sizeof(int));
    int cnt = 0;
    for (int k = 0; k < S->n_act_e; ++k) {
        int e = S->act_e[k];
        if (S->pose_hidx[p->edge_pose[e]] >= 0) { S->pl_off[S->point_hidx[p->edge_point[e]] + 1]++; cnt++; }
    }
    for (int i = 0; i < nl; ++i) S->pl_off[i + 1] += S->pl_off[i];
    S->pl_edge = (int *)malloc(sizeof(int) * (cnt ? cnt : 1));
    int *fill = (int *)calloc(nl ? nl : 1, sizeof(int));
    pe_t *tmp = (pe_t *)malloc(sizeof(pe_t) * (cnt ? cnt : 1));
    for (int k = 0; k < S->n_act_e; ++k) {
        int e = S->act_e[k];
        int ph = S->pose_hidx[p->edge_pose[e]];
        if (ph < 0) continue;
        int l = S->point_hidx[p->edge_point[e]];
        int pos = S->pl_off[l] + fill[l]++;
        tmp[pos].hidx = ph;
        tmp[pos].edge = k; /* index into active list (Hpl slot) */
    }
    for (int l = 0; l < nl; ++l) qsort(tmp + S->pl_off[l], S->pl_off[l + 1] - S->pl_off[l], sizeof(pe_t), pe_cmp);
    for (int i = 0; i < cnt; ++i) S->pl_edge[i] = tmp[i].edge;
    free(tmp); free(fill);
    return 0;
}

/* BlockSolver::buildSystem block_solver.hpp:502-560 */
static void build_system(lba_t *S)
{
    orc_lba_problem_t *p = S->p;
    int np = S->np, nl = S->nl;
    memset(S->Hpp, 0, sizeof(double) * 36 * np);
    memset(S->Hll, 0, sizeof(double) * 9 * nl);
    memset(S->Hpl, 0, sizeof(double) * 18 * S->n_act_e);
    memset(S->b, 0, sizeof(double) * (6 * (size_t)np + 3 * (size_t)nl));
    const int ext = (g_lba_variant & LBA_EXTENDED_SUMS) != 0;
    long double *xHpp = NULL, *xHll = NULL, *xb = NULL;
    if (ext) {
        xHpp = (long double *)calloc(36 * (size_t)np + 1, sizeof(long double));
        xHll = (long double *)calloc(9 * (size_t)nl + 1, sizeof(long double));
        xb = (long double *)calloc(6 * (size_t)np + 3 * (size_t)nl + 1, sizeof(long double));
    }
    for (int k = 0; k < S->n_act_e; ++k) {
        int e = S->act_e[k];
        int stereo = p->edge_stereo[e];
        int D = stereo ? 3 : 2;
        int pi = p->edge_pose[e], li = p->edge_point[e];
        double A[9], B[18];
        edge_jacobians(p->pose_qt + 7 * (size_t)pi, p->point_xyz + 3 * (size_t)li, stereo, p->fx, p->fy, p->bf, A, B);
        const double *er = S->err + 3 * (size_t)e;
        double w = (double)p->edge_inv_sigma2[e];
        double omega_r[3];
        for (int i = 0; i < D; ++i) omega_r[i] = -(w * er[i]);
        double wo = w;
        if (S->robust[e]) {
            double rho[3];
            robustify(edge_chi2(S, e), huber_delta(stereo), rho);
            wo = rho[1] * w;
            for (int i = 0; i < D; ++i) omega_r[i] *= rho[1];
        }
        int hl = S->point_hidx[li], hp = S->pose_hidx[pi];
        /* from = point (always free) */
        double *bl = S->b + 6 * (size_t)np + 3 * (size_t)hl;
        double *Hll = S->Hll + 9 * (size_t)hl;
        for (int r = 0; r < 3; ++r) {
            double s = 0;
            for (int d = 0; d < D; ++d) s += A[d * 3 + r] * omega_r[d];
            if (ext) xb[6 * (size_t)np + 3 * (size_t)hl + r] += s; else bl[r] += s;
            for (int c = 0; c < 3; ++c) {
                double t = 0;
                for (int d = 0; d < D; ++d) t += A[d * 3 + r] * wo * A[d * 3 + c];
                if (ext) xHll[9 * (size_t)hl + r * 3 + c] += t; else Hll[r * 3 + c] += t;
            }
        }
        if (hp >= 0) {
            double *Hpl = S->Hpl + 18 * (size_t)k;
            double *bp = S->b + 6 * (size_t)hp;
            double *Hpp = S->Hpp + 36 * (size_t)hp;
            for (int r = 0; r < 6; ++r) {
                double s = 0;
                for (int d = 0; d < D; ++d) s += B[d * 6 + r] * omega_r[d];
                if (ext) xb[6 * (size_t)hp + r] += s; else bp[r] += s;
                for (int c = 0; c < 6; ++c) {
                    double t = 0;
                    for (int d = 0; d < D; ++d) t += B[d * 6 + r] * wo * B[d * 6 + c];
                    if (ext) xHpp[36 * (size_t)hp + r * 6 + c] += t; else Hpp[r * 6 + c] += t;
                }
                for (int c = 0; c < 3; ++c) {
                    double t = 0;
                    for (int d = 0; d < D; ++d) t += B[d * 6 + r] * wo * A[d * 3 + c];
                    Hpl[r * 3 + c] += t;
                }
            }
        }
    }
    if (ext) {
        for (size_t i = 0; i < 36 * (size_t)np; ++i) S->Hpp[i] = (double)xHpp[i];
        for (size_t i = 0; i < 9 * (size_t)nl; ++i) S->Hll[i] = (double)xHll[i];
        for (size_t i = 0; i < 6 * (size_t)np + 3 * (size_t)nl; ++i) S->b[i] = (double)xb[i];
        free(xHpp); free(xHll); free(xb);
    }
}

/* dense LDL^T (no pivoting) of the symmetric n x n matrix A (full storage, row-major), solve
 * A x = rhs.  Fails on an exactly-zero pivot like Eigen::SimplicialLDLT. */
static int ldlt_solve_ext(double *A, int n, const double *rhs, double *x) /* LBA_EXTENDED_SUMS: the same recurrences, sums in long double */
{
    double *d = (double *)malloc(sizeof(double) * (n ? n : 1));
    for (int j = 0; j < n; ++j) {
        long double dj = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) dj -= (long double)A[(size_t)j * n + k] * A[(size_t)j * n + k] * d[k];
        if ((double)dj == 0.0 || dj != dj) { free(d); return 0; }
        d[j] = (double)dj;
        for (int i = j + 1; i < n; ++i) {
            long double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= (long double)A[(size_t)i * n + k] * A[(size_t)j * n + k] * d[k];
            A[(size_t)i * n + j] = (double)(s / d[j]);
        }
    }
    for (int i = 0; i < n; ++i) {
        long double s = rhs[i];
        for (int k = 0; k < i; ++k) s -= (long double)A[(size_t)i * n + k] * x[k];
        x[i] = (double)s;
    }
    for (int i = 0; i < n; ++i) x[i] /= d[i];
    for (int i = n - 1; i >= 0; --i) {
        long double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= (long double)A[(size_t)k * n + i] * x[k];
        x[i] = (double)s;
    }
    free(d);
    return 1;
}

static int ldlt_solve_plain(double *A, int n, const double *rhs, double *x);
static int ldlt_solve(double *A, int n, const double *rhs, double *x)
{
    int (*f)(double *, int, const double *, double *) = (g_lba_variant & LBA_EXTENDED_SUMS) ? ldlt_solve_ext : ldlt_solve_plain;
    if (!(g_lba_variant & LBA_REVERSE_ELIMINATION)) return f(A, n, rhs, x);
    /* P A P^T (P y) = P rhs with P the reversal: the last unknown is eliminated first */
    double *Ar = (double *)malloc(sizeof(double) * ((size_t)n * n + 1)), *br = (double *)malloc(sizeof(double) * (n + 1)),
           *xr = (double *)malloc(sizeof(double) * (n + 1));
    for (int i = 0; i < n; ++i) {
        br[i] = rhs[n - 1 - i];
        for (int j = 0; j < n; ++j) Ar[(size_t)i * n + j] = A[(size_t)(n - 1 - i) * n + (n - 1 - j)];
    }
    int ok = f(Ar, n, br, xr);
    for (int i = 0; i < n; ++i) x[i] = xr[n - 1 - i];
    free(Ar); free(br); free(xr);
    return ok;
}

static int ldlt_solve_plain(double *A, int n, const double *rhs, double *x)
{
    double *d = (double *)malloc(sizeof(double) * (n ? n : 1));
    /* in place: lower triangle becomes L (unit diag) */
    for (int j = 0; j < n; ++j) {
        double dj = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) dj -= A[(size_t)j * n + k] * A[(size_t)j * n + k] * d[k];
        if (dj == 0.0 || dj != dj) { free(d); return 0; }
        d[j] = dj;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k] * d[k];
            A[(size_t)i * n + j] = s / dj;
        }
    }
    for (int i = 0; i < n; ++i) {
        double s = rhs[i];
        for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * x[k];
        x[i] = s;
    }
    for (int i = 0; i < n; ++i) x[i] /= d[i];
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * x[k];
        x[i] = s;
    }
    free(d);
    return 1;
}

/* BlockSolver::solve (Schur) block_solver.hpp:367-486 */
static int solve_schur(lba_t *S)
{
    orc_lba_problem_t *p = S->p;
    (void)p;
    int np = S->np, nl = S->nl, n6 = 6 * np;
    memset(S->Hs, 0, sizeof(double) * (size_t)n6 * n6);
    for (int i = 0; i < np; ++i)
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) S->Hs[(size_t)(6 * i + r) * n6 + 6 * i + c] = S->Hpp[36 * (size_t)i + r * 6 + c];
    memset(S->coeff, 0, sizeof(double) * n6);
    const int ext = (g_lba_variant & LBA_EXTENDED_SUMS) != 0;
    long double *xHs = NULL, *xco = NULL;
    if (ext) {
        xHs = (long double *)malloc(sizeof(long double) * ((size_t)n6 * n6 + 1));
        xco = (long double *)calloc((size_t)n6 + 1, sizeof(long double));
        for (size_t i = 0; i < (size_t)n6 * n6; ++i) xHs[i] = S->Hs[i];
    }
    for (int l = 0; l < nl; ++l) {
        double *Dinv = S->Dinv + 9 * (size_t)l;
        mat3_inverse(S->Hll + 9 * (size_t)l, Dinv);
        const double *bl = S->b + n6 + 3 * (size_t)l;
        double db[3];
        for (int r = 0; r < 3; ++r) db[r] = Dinv[r * 3] * bl[0] + Dinv[r * 3 + 1] * bl[1] + Dinv[r * 3 + 2] * bl[2];
        for (int a = S->pl_off[l]; a < S->pl_off[l + 1]; ++a) {
            int ka = S->pl_edge[a];
            int i1 = S->pose_hidx[S->p->edge_pose[S->act_e[ka]]];
            const double *Bi = S->Hpl + 18 * (size_t)ka;
            double BDinv[18];
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 3; ++c)
                    BDinv[r * 3 + c] = Bi[r * 3] * Dinv[c] + Bi[r * 3 + 1] * Dinv[3 + c] + Bi[r * 3 + 2] * Dinv[6 + c];
            for (int r = 0; r < 6; ++r) {
                double t = Bi[r * 3] * db[0] + Bi[r * 3 + 1] * db[1] + Bi[r * 3 + 2] * db[2];
                if (ext) xco[6 * i1 + r] += t; else S->coeff[6 * i1 + r] += t;
            }
            for (int bq = a; bq < S->pl_off[l + 1]; ++bq) {
                int kb = S->pl_edge[bq];
                int i2 = S->pose_hidx[S->p->edge_pose[S->act_e[kb]]];
                const double *Bj = S->Hpl + 18 * (size_t)kb;
                for (int r = 0; r < 6; ++r)
                    for (int c = 0; c < 6; ++c) {
                        double t = BDinv[r * 3] * Bj[c * 3] + BDinv[r * 3 + 1] * Bj[c * 3 + 1] + BDinv[r * 3 + 2] * Bj[c * 3 + 2];
                        if (ext) xHs[(size_t)(6 * i1 + r) * n6 + 6 * i2 + c] -= t; else S->Hs[(size_t)(6 * i1 + r) * n6 + 6 * i2 + c] -= t;
                    }
            }
        }
    }
    if (ext) {
        for (size_t i = 0; i < (size_t)n6 * n6; ++i) S->Hs[i] = (double)xHs[i];
        for (int i = 0; i < n6; ++i) S->coeff[i] = (double)xco[i];
        free(xHs); free(xco);
    }
    for (int i = 0; i < n6; ++i) S->bs[i] = S->b[i] - S->coeff[i];
    /* symmetrise from the upper block triangle (the solver only reads the upper part) */
    for (int r = 0; r < n6; ++r)
        for (int c = r + 1; c < n6; ++c) S->Hs[(size_t)c * n6 + r] = S->Hs[(size_t)r * n6 + c];
    if (n6 > 0 && !ldlt_solve(S->Hs, n6, S->bs, S->x)) return 0;
    /* landmarks: xl = Dinv * (bl - B^T xp) */
    for (int l = 0; l < nl; ++l) {
        double cl[3] = {S->b[n6 + 3 * l], S->b[n6 + 3 * l + 1], S->b[n6 + 3 * l + 2]};
        for (int a = S->pl_off[l]; a < S->pl_off[l + 1]; ++a) {
            int ka = S->pl_edge[a];
            int i1 = S->pose_hidx[S->p->edge_pose[S->act_e[ka]]];
            const double *Bi = S->Hpl + 18 * (size_t)ka;
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < 6; ++r) cl[c] += Bi[r * 3 + c] * (-S->x[6 * i1 + r]);
        }
        const double *Dinv = S->Dinv + 9 * (size_t)l;
        for (int r = 0; r < 3; ++r) S->x[n6 + 3 * l + r] = Dinv[r * 3] * cl[0] + Dinv[r * 3 + 1] * cl[1] + Dinv[r * 3 + 2] * cl[2];
    }
    return 1;
}

static void set_lambda(lba_t *S, double lambda)
{
    for (int i = 0; i < S->np; ++i)
        for (int d = 0; d < 6; ++d) {
            S->bk_dpp[6 * i + d] = S->Hpp[36 * (size_t)i + d * 7];
            S->Hpp[36 * (size_t)i + d * 7] += lambda;
        }
    for (int i = 0; i < S->nl; ++i)
        for (int d = 0; d < 3; ++d) {
            S->bk_dll[3 * i + d] = S->Hll[9 * (size_t)i + d * 4];
            S->Hll[9 * (size_t)i + d * 4] += lambda;
        }
}
static void restore_diagonal(lba_t *S)
{
    for (int i = 0; i < S->np; ++i)
        for (int d = 0; d < 6; ++d) S->Hpp[36 * (size_t)i + d * 7] = S->bk_dpp[6 * i + d];
    for (int i = 0; i < S->nl; ++i)
        for (int d = 0; d < 3; ++d) S->Hll[9 * (size_t)i + d * 4] = S->bk_dll[3 * i + d];
}
static void push_state(lba_t *S)
{
    for (int i = 0; i < S->np; ++i) memcpy(S->bk_pose + 7 * i, S->p->pose_qt + 7 * (size_t)S->hpose[i], sizeof(double) * 7);
    for (int i = 0; i < S->nl; ++i) memcpy(S->bk_point + 3 * i, S->p->point_xyz + 3 * (size_t)S->hpoint[i], sizeof(double) * 3);
}
static void pop_state(lba_t *S)
{
    for (int i = 0; i < S->np; ++i) memcpy(S->p->pose_qt + 7 * (size_t)S->hpose[i], S->bk_pose + 7 * i, sizeof(double) * 7);
    for (int i = 0; i < S->nl; ++i) memcpy(S->p->point_xyz + 3 * (size_t)S->hpoint[i], S->bk_point + 3 * i, sizeof(double) * 3);
}
/* SparseOptimizer::update :422-435 ; VertexSE3Expmap::oplusImpl ; VertexSBAPointXYZ::oplusImpl */
static void apply_update(lba_t *S)
{
    for (int i = 0; i < S->np; ++i) {
        double ex[7], out[7];
        double *T = S->p->pose_qt + 7 * (size_t)S->hpose[i];
        orc_se3_exp(S->x + 6 * i, ex);
        orc_se3_mul(ex, T, out);
        memcpy(T, out, sizeof(out));
    }
    for (int i = 0; i < S->nl; ++i) {
        double *X = S->p->point_xyz + 3 * (size_t)S->hpoint[i];
        for (int d = 0; d < 3; ++d) X[d] += S->x[6 * S->np + 3 * i + d];
    }
}

/* bool SparseOptimizer::terminate() (reads *_forceStopFlag).  Test hook: stop_at_poll > 0 makes the flag count as set
 * from that evaluation on, as if another thread (LocalMapping::InterruptBA, src/LocalMapping.cc:118-123) had set it
 * at that moment; the evaluations are counted so that an asynchronous abort of the device path can be replayed. */
static int terminate_flag(lba_t *S)
{
    S->polls++;
    int set = S->stop_poll > 0;
    if (!set && ((S->p->stop_flag && *S->p->stop_flag != 0) || (S->p->stop_at_poll > 0 && S->polls >= S->p->stop_at_poll))) {
        S->stop_poll = S->polls;
        set = 1;
    }
    return set;
}

enum { LM_OK = 0, LM_TERMINATE = 1, LM_FAIL = -1 };

/* OptimizationAlgorithmLevenberg::solve :61-164 */
static int lm_solve(lba_t *S, int iteration, orc_lba_result_t *res)
{
    compute_active_errors(S);
    double currentChi = active_robust_chi2(S);
    double tempChi = currentChi;
    double iniChi = currentChi;
    build_system(S);
    if (iteration == 0) {
        double maxDiagonal = 0.;
        for (int i = 0; i < S->np; ++i)
            for (int d = 0; d < 6; ++d) maxDiagonal = fmax(fabs(S->Hpp[36 * (size_t)i + d * 7]), maxDiagonal);
        for (int i = 0; i < S->nl; ++i)
            for (int d = 0; d < 3; ++d) maxDiagonal = fmax(fabs(S->Hll[9 * (size_t)i + d * 4]), maxDiagonal);
        S->lambda = 1e-5 * maxDiagonal;
        S->ni = 2;
        S->nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    const int maxTrials = 10;
    size_t dim = 6 * (size_t)S->np + 3 * (size_t)S->nl;
    do {
        S->trials++;
        push_state(S);
        set_lambda(S, S->lambda);
        int ok2 = solve_schur(S);
        apply_update(S);
        restore_diagonal(S);
        compute_active_errors(S);
        tempChi = active_robust_chi2(S);
        if (!ok2) tempChi = 1.7976931348623157e308;
        rho = (currentChi - tempChi);
        double scale = 0.;
        for (size_t j = 0; j < dim; ++j) scale += S->x[j] * (S->lambda * S->x[j] + S->b[j]);
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && isfinite(tempChi)) {
            double alpha = 1. - pow((2 * rho - 1), 3);
            alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
            double scaleFactor = 1. / 3. > alpha ? 1. / 3. : alpha;
            S->lambda *= scaleFactor;
            S->ni = 2;
            currentChi = tempChi;
        } else {
            S->lambda *= S->ni;
            S->ni *= 2;
            pop_state(S);
        }
        qmax++;
    } while (rho < 0 && qmax < maxTrials && !terminate_flag(S));
    if (res && res->n_trace < 64) {
        res->lambda_trace[res->n_trace] = S->lambda;
        res->chi2_trace[res->n_trace] = currentChi;
        res->n_trace++;
    }
    if (qmax == maxTrials || rho == 0) return LM_TERMINATE;
    if ((iniChi - currentChi) * 1e3 < iniChi)
        S->nBad++;
    else
        S->nBad = 0;
    if (S->nBad >= 3) return LM_TERMINATE;
    return LM_OK;
}

/* SparseOptimizer::optimize :354-419 */
static int optimize(lba_t *S, int iterations, orc_lba_result_t *res)
{
    int cj = 0, ok = 1;
    for (int i = 0; i < iterations && !terminate_flag(S) && ok; i++) {
        int r = lm_solve(S, i, res);
        ok = (r == LM_OK);
        ++cj;
    }
    return cj;
}

static int depth_positive(const orc_lba_problem_t *p, int e)
{
    double q[3];
    se3_map(p->pose_qt + 7 * (size_t)p->edge_pose[e], p->point_xyz + 3 * (size_t)p->edge_point[e], q);
    return q[2] > 0.0;
}

/* Optimizer::LocalBundleAdjustment numerical part, Optimizer.cc:656-744 */
int orc_lba_solve(orc_lba_problem_t *p, orc_lba_result_t *r)
{
    lba_t S;
    memset(&S, 0, sizeof(S));
    S.p = p;
    S.err = (double *)calloc((size_t)p->n_edges * 3 + 1, sizeof(double));
    S.level1 = (uint8_t *)calloc(p->n_edges + 1, 1);
    S.robust = (uint8_t *)malloc(p->n_edges + 1);
    memset(S.robust, 1, p->n_edges + 1);
    S.pose_hidx = (int *)malloc(sizeof(int) * (p->n_poses + 1));
    S.point_hidx = (int *)malloc(sizeof(int) * (p->n_points + 1));
    if (r) { r->n_trace = 0; r->iters_done1 = r->iters_done2 = 0; r->polls = r->stop_poll = r->trials = 0; }
    int status = 0;
    if (terminate_flag(&S)) { status = 1; goto done; } /* :656-658 early return */
    if (initialize_optimization(&S) == 0) {
        int it = optimize(&S, p->iters1, r);
        if (r) r->iters_done1 = it;
    }
    if (!terminate_flag(&S)) { /* bDoMore :663-710 */
        for (int e = 0; e < p->n_edges; ++e) {
            double th = p->edge_stereo[e] ? 7.815 : 5.991;
            if (edge_chi2(&S, e) > th || !depth_positive(p, e)) S.level1[e] = 1;
            S.robust[e] = 0;
        }
        if (initialize_optimization(&S) == 0) {
            int it = optimize(&S, p->iters2, r);
            if (r) r->iters_done2 = it;
        }
    }
    if (r) {
        for (int e = 0; e < p->n_edges; ++e) {
            double th = p->edge_stereo[e] ? 7.815 : 5.991;
            double c = edge_chi2(&S, e);
            int dp = depth_positive(p, e);
            if (r->edge_chi2) r->edge_chi2[e] = c;
            if (r->edge_depth_pos) r->edge_depth_pos[e] = (uint8_t)dp;
            if (r->edge_outlier) r->edge_outlier[e] = (uint8_t)(c > th || !dp);
            if (r->edge_level1) r->edge_level1[e] = S.level1[e];
        }
    }
done:
    if (r) { r->polls = S.polls; r->stop_poll = S.stop_poll; r->trials = S.trials; }
    free_structure(&S);
    free(S.err); free(S.level1); free(S.robust); free(S.pose_hidx); free(S.point_hidx);
    return status;
}

/* Converter::toSE3Quat src/Converter.cc:37-47 : float32 4x4 Tcw -> SE3Quat */
void orc_pose_from_Tcw_f32(const float T[16], double qt[7])
{
    double R[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i * 3 + j] = (double)T[i * 4 + j];
    double q[4];
    orc_quat_from_rot(R, q);
    quat_normalize_rot(q);
    for (int i = 0; i < 4; ++i) qt[i] = q[i];
    for (int i = 0; i < 3; ++i) qt[4 + i] = (double)T[i * 4 + 3];
}

/* Converter::toCvMat(SE3Quat) src/Converter.cc:49-53,63-71 : to_homogeneous_matrix -> float32 */
void orc_pose_to_Tcw_f32(const double qt[7], float T[16])
{
    double R[9];
    orc_rot_from_quat(qt, R);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[i * 4 + j] = (float)R[i * 3 + j];
        T[i * 4 + 3] = (float)qt[4 + i];
    }
    T[12] = T[13] = T[14] = 0.f;
    T[15] = 1.f;
}

/* ------------------------------------------------------------------ Optimizer::PoseOptimization
 * src/Optimizer.cc:239-452 with EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose
 * (types_six_dof_expmap.h:143-205, .cpp:266-364), BaseUnaryEdge::constructQuadraticForm
 * (core/base_unary_edge.hpp:43-72), BlockSolver without Schur + LinearSolverDense
 * (solvers/linear_solver_dense.h:64-110; Eigen's pivoted LDLT restated as an unpivoted Cholesky
 * with the same "positive" acceptance test) and the same Levenberg loop as LocalBA. */
static void pose_edge_error(const double qt[7], const double Xw[3], const double obs[3], int stereo, double fx,
                            double fy, double cx, double cy, double bf, double err[3])
{
    double p[3];
    se3_map(qt, Xw, p);
    if (!stereo) {
        double u = p[0] / p[2], v = p[1] / p[2];
        err[0] = obs[0] - (u * fx + cx);
        err[1] = obs[1] - (v * fy + cy);
        err[2] = 0;
    } else {
        const float invz = (float)(1.0f / p[2]);
        double r0 = p[0] * invz * fx + cx;
        double r1 = p[1] * invz * fy + cy;
        double r2 = r0 - bf * invz; /* member bf is double here (.cpp:304) */
        err[0] = obs[0] - r0;
        err[1] = obs[1] - r1;
        err[2] = obs[2] - r2;
    }
}

static void pose_edge_jacobian(const double qt[7], const double Xw[3], int stereo, double fx, double fy, double bf,
                               double J[18])
{
    double p[3];
    se3_map(qt, Xw, p);
    double x = p[0], y = p[1], invz = 1.0 / p[2], invz_2 = invz * invz;
    memset(J, 0, sizeof(double) * 18);
    J[0] = x * y * invz_2 * fx;
    J[1] = -(1 + (x * x * invz_2)) * fx;
    J[2] = y * invz * fx;
    J[3] = -invz * fx;
    J[4] = 0;
    J[5] = x * invz_2 * fx;
    J[6] = (1 + y * y * invz_2) * fy;
    J[7] = -x * y * invz_2 * fy;
    J[8] = -x * invz * fy;
    J[9] = 0;
    J[10] = -invz * fy;
    J[11] = y * invz_2 * fy;
    if (stereo) {
        J[12] = J[0] - bf * y * invz_2;
        J[13] = J[1] + bf * x * invz_2;
        J[14] = J[2];
        J[15] = J[3];
        J[16] = 0;
        J[17] = J[5] - bf * invz_2;
    }
}

typedef struct {
    const orc_pose_problem_t *p;
    double qt[7];
    double *err;      /* n x 3 */
    uint8_t *level1;  /* n */
    uint8_t *robust;  /* n */
    double H[36], b[6], x[6];
} po_t;

static double po_chi2(const po_t *S, int e)
{
    const double *er = S->err + 3 * (size_t)e;
    double w = (double)S->p->inv_sigma2[e];
    int D = S->p->stereo[e] ? 3 : 2;
    double s = 0;
    for (int i = 0; i < D; ++i) s += er[i] * (w * er[i]);
    return s;
}
static void po_errors(po_t *S)
{
    const orc_pose_problem_t *p = S->p;
    for (int e = 0; e < p->n; ++e)
        if (!S->level1[e])
            pose_edge_error(S->qt, p->Xw + 3 * (size_t)e, p->obs + 3 * (size_t)e, p->stereo[e], p->fx, p->fy, p->cx, p->cy, p->bf,
                            S->err + 3 * (size_t)e);
}
static double po_robust_chi2(const po_t *S)
{
    double chi = 0;
    for (int e = 0; e < S->p->n; ++e) {
        if (S->level1[e]) continue;
        double c = po_chi2(S, e);
        if (S->robust[e]) {
            double rho[3];
            robustify(c, huber_delta(S->p->stereo[e]), rho);
            chi += rho[0];
        } else
            chi += c;
    }
    return chi;
}
static void po_build(po_t *S)
{
    const orc_pose_problem_t *p = S->p;
    memset(S->H, 0, sizeof(S->H));
    memset(S->b, 0, sizeof(S->b));
    for (int e = 0; e < p->n; ++e) {
        if (S->level1[e]) continue;
        int D = p->stereo[e] ? 3 : 2;
        double J[18];
        pose_edge_jacobian(S->qt, p->Xw + 3 * (size_t)e, p->stereo[e], p->fx, p->fy, p->bf, J);
        const double *er = S->err + 3 * (size_t)e;
        double w = (double)p->inv_sigma2[e], wo = w, r1 = 1.0;
        if (S->robust[e]) {
            double rho[3];
            robustify(po_chi2(S, e), huber_delta(p->stereo[e]), rho);
            r1 = rho[1];
            wo = rho[1] * w;
        }
        for (int r = 0; r < 6; ++r) {
            double s = 0;
            for (int d = 0; d < D; ++d) s += J[d * 6 + r] * (w * er[d]);
            S->b[r] -= r1 * s;
            for (int c = 0; c < 6; ++c) {
                double t = 0;
                for (int d = 0; d < D; ++d) t += J[d * 6 + r] * wo * J[d * 6 + c];
                S->H[r * 6 + c] += t;
            }
        }
    }
}
/* Cholesky solve of (H + lambda I) x = b; returns 0 if not positive definite */
static int po_solve(const po_t *S, double lambda, double x[6])
{
    double L[36];
    for (int i = 0; i < 36; ++i) L[i] = S->H[i];
    for (int i = 0; i < 6; ++i) L[i * 7] += lambda;
    for (int j = 0; j < 6; ++j) {
        double d = L[j * 6 + j];
        for (int k = 0; k < j; ++k) d -= L[j * 6 + k] * L[j * 6 + k];
        if (!(d > 0)) return 0;
        d = sqrt(d);
        L[j * 6 + j] = d;
        for (int i = j + 1; i < 6; ++i) {
            double s = L[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
            L[i * 6 + j] = s / d;
        }
    }
    double y[6];
    for (int i = 0; i < 6; ++i) {
        double s = S->b[i];
        for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * y[k];
        y[i] = s / L[i * 6 + i];
    }
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k];
        x[i] = s / L[i * 6 + i];
    }
    return 1;
}

int orc_pose_optimization(const orc_pose_problem_t *p, const double pose_in[7], double pose_out[7], uint8_t *outlier,
                          int *n_bad_out)
{
    po_t S;
    memset(&S, 0, sizeof(S));
    S.p = p;
    S.err = (double *)calloc((size_t)p->n * 3 + 1, sizeof(double));
    S.level1 = (uint8_t *)calloc(p->n + 1, 1);
    S.robust = (uint8_t *)malloc(p->n + 1);
    memset(S.robust, 1, p->n + 1);
    memcpy(S.qt, pose_in, sizeof(S.qt));
    for (int e = 0; e < p->n; ++e) outlier[e] = 0;
    int nBad = 0;
    if (p->n < 3) { /* nInitialCorrespondences < 3 -> return 0 (:355-356) */
        memcpy(pose_out, pose_in, sizeof(double) * 7);
        *n_bad_out = 0;
        free(S.err); free(S.level1); free(S.robust);
        return 0;
    }
    const float chi2Mono = 5.991f, chi2Stereo = 7.815f;
    for (int it = 0; it < 4; ++it) {
        memcpy(S.qt, pose_in, sizeof(S.qt)); /* vSE3->setEstimate(toSE3Quat(pFrame->mTcw)) :368 */
        int n_active = 0;
        for (int e = 0; e < p->n; ++e) n_active += !S.level1[e];
        if (n_active > 0) {
            double lambda = 0, ni = 2;
            int nBadLM = 0, ok = 1;
            for (int i = 0; i < 10 && ok; ++i) { /* optimize(10) */
                po_errors(&S);
                double currentChi = po_robust_chi2(&S), tempChi = currentChi;
                const double iniChi = currentChi;
                po_build(&S);
                if (i == 0) {
                    double maxDiagonal = 0.;
                    for (int d = 0; d < 6; ++d) maxDiagonal = fmax(fabs(S.H[d * 7]), maxDiagonal);
                    lambda = 1e-5 * maxDiagonal;
                    ni = 2;
                    nBadLM = 0;
                }
                double rho = 0;
                int qmax = 0;
                do {
                    double bk[7];
                    memcpy(bk, S.qt, sizeof(bk));
                    int ok2 = po_solve(&S, lambda, S.x);
                    double ex[7], out[7];
                    orc_se3_exp(S.x, ex);
                    orc_se3_mul(ex, S.qt, out);
                    memcpy(S.qt, out, sizeof(out));
                    po_errors(&S);
                    tempChi = po_robust_chi2(&S);
                    if (!ok2) tempChi = 1.7976931348623157e308;
                    rho = (currentChi - tempChi);
                    double scale = 0.;
                    for (int j = 0; j < 6; ++j) scale += S.x[j] * (lambda * S.x[j] + S.b[j]);
                    scale += 1e-3;
                    rho /= scale;
                    if (rho > 0 && isfinite(tempChi)) {
                        double alpha = 1. - pow((2 * rho - 1), 3);
                        alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
                        double scaleFactor = 1. / 3. > alpha ? 1. / 3. : alpha;
                        lambda *= scaleFactor;
                        ni = 2;
                        currentChi = tempChi;
                    } else {
                        lambda *= ni;
                        ni *= 2;
                        memcpy(S.qt, bk, sizeof(bk));
                    }
                    qmax++;
                } while (rho < 0 && qmax < 10);
                if (qmax == 10 || rho == 0) {
                    ok = 0;
                } else {
                    if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
                    if (nBadLM >= 3) ok = 0;
                }
            }
        }
        nBad = 0;
        for (int e = 0; e < p->n; ++e) { /* :371-430 (mono and stereo loops have the same body) */
            if (outlier[e])
                pose_edge_error(S.qt, p->Xw + 3 * (size_t)e, p->obs + 3 * (size_t)e, p->stereo[e], p->fx, p->fy, p->cx, p->cy,
                                p->bf, S.err + 3 * (size_t)e);
            const float chi2 = (float)po_chi2(&S, e);
            if (chi2 > (p->stereo[e] ? chi2Stereo : chi2Mono)) {
                outlier[e] = 1;
                S.level1[e] = 1;
                nBad++;
            } else {
                outlier[e] = 0;
                S.level1[e] = 0;
            }
            if (it == 2) S.robust[e] = 0;
        }
        if (p->n < 10) break; /* optimizer.edges().size() < 10 */
    }
    memcpy(pose_out, S.qt, sizeof(double) * 7);
    *n_bad_out = nBad;
    free(S.err); free(S.level1); free(S.robust);
    return p->n - nBad;
}
