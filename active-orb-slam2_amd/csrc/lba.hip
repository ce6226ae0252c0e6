// Optimizer::LocalBundleAdjustment numerical core on gfx950 (reference src/Optimizer.cc:507-744 and
// the vendored g2o it drives: optimization_algorithm_levenberg.cpp:61-164, block_solver.hpp:367-604,
// base_binary_edge.hpp:55-120, robust_kernel_impl.cpp:78-91, types_six_dof_expmap.{h,cpp},
// se3quat.h).  All arithmetic is IEEE double like g2o (one float reciprocal in the stereo
// projection, types_six_dof_expmap.cpp:151).
//
// One call solves a BATCH of independent windows (aos2_lba_solve_batch; aos2_lba_solve is a batch of one).
// Every kernel takes the array of window descriptors and uses blockIdx.y as the window, so the latency-bound
// Levenberg-Marquardt chains of all windows advance side by side on different compute units.
//
// The whole procedure of a window -- optimize(5), the outlier pass, optimize(10), the final inlier check -- is
// enqueued as ONE program; the Levenberg-Marquardt control flow (rho, lambda, accept / restore, the three
// termination rules, the iteration counters, the polls of pbStopFlag) lives in a small per-window state block on the
// device (LmState), updated after every trial by the landmark kernel's last workgroup (lm_decide); every kernel is gated by that state.  The host only
// looks at the states when the program has run: a window that needed more trials than were enqueued (steps rejected
// by the gain ratio) gets another short program.  No host round trip per trial.
//
// Device layout: SoA doubles for poses (qx qy qz qw tx ty tz), points, edges.  The edge set of a window has three CSR
// views (by point, by free pose, by point restricted to free poses sorted by pose = the Hpl column of
// block_solver.hpp:398) built once per call.  The second optimisation (level-0 edges only, Optimizer.cc:672-708)
// reuses them: an excluded edge is MASKED -- its Jacobians, weights and Hpl block are written as zeros, its residual
// is left alone like g2o leaves the _error of an inactive edge -- so every sum it took part in receives +0.0, which
// is the same as leaving it out; vertices that lose all their edges keep a lambda-only diagonal block, decoupled from
// the rest, and receive a zero update (g2o drops them from the index mapping instead: same result for the others).
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <numeric>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <type_traits>

#include "lba_math.h"
#include "ldlt_reg.h"

namespace aos2 {

struct Cam {
    double fx, fy, cx, cy, bf;
    float bf_f;  // cam_project takes bf as `const float&` (types_six_dof_expmap.cpp:150)
    double delta_mono, delta_stereo;  // Huber deltas (float sqrt -> double, Optimizer.cc:570-571)
};

// Levenberg-Marquardt / SparseOptimizer::optimize state of one window (levenberg.cpp:61-164, sparse_optimizer.cpp:354-419)
struct LmState {
    double lambda, ni, currentChi, iniChi;
    double final_chi2, final_lambda;
    int32_t phase;      // 0: first optimize() running, 1: first finished, 2: second running, 3: finished
    int32_t run;        // the trial kernels execute
    int32_t lin;        // the system is (re)linearised at the current estimates (after an accepted step)
    int32_t initp;      // start of an optimize() call: residuals, system, lambda init pending
    int32_t xmark;      // transition: the outlier pass runs
    int32_t it, qmax, nBad;
    int32_t iters_max[2], iters_done[2], trials[2];
    int32_t polls;      // evaluations of terminate() so far (the host's entry check is the first)
    int32_t stop_poll;  // the poll at which the stop flag was first seen set (0 = never)
    int32_t stop_at_poll;  // test hook: the flag counts as set from this poll on (0 = off)
    int32_t n_active;   // level-0 edges of the second optimisation
    int32_t solver_failed;  // trials whose reduced system hit a zero pivot
    int32_t blocks_done;    // workgroups of the current k_points launch that delivered their partial sums
    int32_t ntr;            // trials recorded below (AOS2_LBA_TRACE=1 prints them)
    // Where the edges' _error of the LAST residual evaluation can be re-formed from (the trials do not store 24 bytes per edge;
    // the two passes that read _error -- the outlier pass and the final check -- re-form it): 0 = nowhere, the stored values
    // stand (no evaluation since they were written); 1 = the current estimates (evaluation at the start of an optimisation, or
    // an accepted trial); 2 = the backup, which after a rejected trial holds the trial's estimates (pop() swaps).
    int32_t err_at;
    int32_t pad_;
    double tr_rho[48], tr_temp[48], tr_cur[48], tr_lambda[48];
    long long dbg[16];      // cycle counters of the last reduced-system kernel (AOS2_LBA_TRACE=1)
};

// device-side view of one window
struct LbaWin {
    int n_poses, n_points, n_edges, np, nl, n_items;
    int iters1, iters2;              // optimize(5), optimize(10) (Optimizer.cc:661, 708)
    // raw float32 inputs as the reference holds them (staged), converted by k_prepare (Converter.cc:37-47, 83-90)
    const float *in_Tcw, *in_xyz, *in_obs, *in_w;   // (observations and information weights stay float32: the kernels widen them on
                                                     // load -- exact -- instead of reading converted double copies, 16 bytes per edge and use)
    double *pose, *point;            // estimates, contiguous [7 n_poses | 3 n_points]
    double *bk;                      // SparseOptimizer::push backup of the same span
    int est_n;
    const int32_t *e_pose, *e_point;
    const uint8_t *e_stereo;
    uint8_t *e_robust, *e_level1;
    double *err;                     // n_edges x 3, last computed _error
    Cam cam;
    const int32_t *pl_pos;           // edge -> its position in the landmark's free-keyframe edge list (-1: fixed keyframe)
    const int32_t *hpose, *hpoint;   // hidx -> pose / point index
    const int32_t *pt_off, *pt_k;    // edges by point (insertion order)
    const int32_t *ps_off, *ps_k;    // edges by free pose
    const int32_t *pl_off, *pl_ph;   // free-pose edges by point, ascending pose hidx: the pose hidx of each position
    // Schur complement by items: item = (landmark, free-pose edges ka <= kb of it), ranked by (pose, pose) block in
    // upper-triangular order, landmark order inside a block (build_schur_items)
    const int32_t *it_ka, *it_kb, *it_l, *blk_off;
    // k_schur's packed units (build_schur_units): rows of <= 16 items of ONE off-diagonal block, 16 rows per workgroup
    const int32_t *sr_o0, *sr_info, *sr_ij;   // per row: first item; items | row-in-block << 8 | rows-of-block << 16; i1 | i2 << 16
    int n_srows;
    // Linearisation record of a free-keyframe edge, DENSE by its position in the pl list (a landmark's records are
    // neighbours): {a = x / z, b = y / z, iz = 1 / z of the camera-frame point, robustified information w} -- 32 bytes in
    // place of the edge's 144-byte Hpl block J_pose^T (w Omega) J_point, which factors as -E^T C R (EdgeLin below; R: the
    // keyframe's rotation at the linearisation point, Rl) and is applied in that form.  w < 0 marks a stereo edge (|w| is
    // the weight); a masked edge holds zeros: every product it takes part in is +-0.
    double *lrec;
    double *lomr;                    // 3 per free-keyframe edge (position like lrec): omr = -rho' Omega e of the linearisation (k_schur forms b_p from it)
    double *Rl;                      // 9 per free keyframe (hidx): rotation the system was linearised at (k_lin)
    double *Hpp, *Hll, *b, *x, *Hs, *bs;
    double *tmp;                     // scale terms of the poses (6 np)
    double *scal;                    // [3] solve ok
    double *part;                    // per-landmark sums of k_points: chi2 terms [0, nl), scale terms [nl, 2 nl)
    int n_part;                      // workgroups of a k_points launch for this window
    double *ldlt;                    // factorisation scratch of the global-memory variant
    int npad, ldlt_lds;
    int hs_ld;                       // leading dimension of Hs: 6 np, or npad when the reduced system is factorised in place (beyond LDS)
    LmState *st;
    const int32_t *abort_word;       // mapped host memory: the forwarded pbStopFlag
    float *out_Tcw, *out_xyz;        // Converter::toCvMat / toCvMat(Vector3d) write-back (Optimizer.cc:763-778)
    double *out_chi2;
    uint8_t *out_outlier;
};

// One workgroup's share of a launch: (window, what).  The host lays the work of ALL windows of a call out as task lists (one per
// kernel family) instead of grids padded to the largest window -- the windows of a batch differ (10-40 keyframes, 2-6 k points),
// and a workgroup that only finds out it has nothing to do still holds a slot for two dependent loads.
struct SchurTask {
    int32_t w;      // window (-1: padding)
    int32_t code;   // k_schur: kind << 28 | argument; k_lin: kind << 28 | block; k_points: block
};

// bool SparseOptimizer::terminate(): counts the evaluation, latches the flag
// `seen` >= 0: the value of the flag read by the caller shortly before (the word lives in host memory: a read is a PCIe
// round trip, which the decision starts ahead of its other loads)
// (measurement switch, off: a raised wave priority for LocalBA's kernels -- chains of dependent memory and f64 operations at low
// occupancy -- so that they issue ahead of another stream's VALU-bound waves on the same SIMD, like octree_kernel's.  The composite
// did not move: 72.5 / 72.5 k frames/s without, 70.8 / 72.2 k with priority 2, 71.9 k with 3; the batch alone 4.10 ms either way
// (profiles/r06_composite_decomposition.txt).  What LocalBA costs the step is not issue order.)
#ifndef AOS2_LBA_WAVE_PRIO
#define AOS2_LBA_WAVE_PRIO 0
#endif
__device__ __forceinline__ void lba_wave_prio()
{
#if AOS2_LBA_WAVE_PRIO > 0
    __builtin_amdgcn_s_setprio(AOS2_LBA_WAVE_PRIO);
#endif
}

__device__ inline bool lm_poll(LmState *st, const int32_t *abort_word, int seen = -1)
{
    st->polls++;
    if (st->stop_poll > 0) return true;
    const int a = seen >= 0 ? seen : abort_word ? __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
    if (a || (st->stop_at_poll > 0 && st->polls >= st->stop_at_poll)) {
        st->stop_poll = st->polls;
        return true;
    }
    return false;
}

// Optimizer.cc:663-666 right after optimizer.optimize(5) returned: bDoMore = !*pbStopFlag.  Called where the first
// optimisation ends (the LM decision of its last trial, or k_prepare when it does not run at all); xmark = 1 lets the
// transition kernel run the outlier pass.
__device__ inline void lm_first_done(LmState *st, const int32_t *abort_word, int seen = -1)
{
    st->phase = 1;
    st->xmark = 0;
    if (lm_poll(st, abort_word, seen)) {
        st->phase = 3;
        return;
    }
    st->xmark = 1;
    st->n_active = 0;
}

// Sums of two values per thread of an N-thread workgroup (binary trees in LDS, fixed order) -> out0[blockIdx.x], out1[blockIdx.x]
template <int N>
__device__ __forceinline__ void workgroup_sum2(double v0, double v1, double *out0, double *out1)
{
    __shared__ double sh[2][N];
    sh[0][threadIdx.x] = v0;
    sh[1][threadIdx.x] = v1;
    __syncthreads();
#pragma unroll
    for (int s = N / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sh[0][threadIdx.x] = sh[0][threadIdx.x] + sh[0][threadIdx.x + s];
            sh[1][threadIdx.x] = sh[1][threadIdx.x] + sh[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out0[blockIdx.x] = sh[0][0];
        out1[blockIdx.x] = sh[1][0];
    }
}

// Converter::toSE3Quat / toVector3d and the float -> double copies of Optimizer.cc:523-525, 552-553, 597-606
__global__ __launch_bounds__(256) void k_prepare(const LbaWin *__restrict__ wins, int stop_at_poll)
{
    lba_wave_prio();
    const LbaWin &W = wins[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < W.n_poses) pose_from_Tcw(W.in_Tcw + 16 * (size_t)i, W.pose + 7 * (size_t)i);
    if (i < W.n_points)
        for (int d = 0; d < 3; ++d) W.point[3 * (size_t)i + d] = (double)W.in_xyz[3 * (size_t)i + d];
    if (i < W.n_edges) {
        for (int d = 0; d < 3; ++d) W.err[3 * (size_t)i + d] = 0.0;
        W.e_robust[i] = 1;
        W.e_level1[i] = 0;
    }
    if (blockIdx.x == 0 && W.hs_ld == W.npad) {   // reduced system stored padded (k_ldlt_reg, k_ldlt_dev): the identity tail of the matrix, once
        const int n = 6 * W.np, npad = W.npad;
        for (int q = threadIdx.x; q < (npad - n) * npad; q += 256) {
            const int r = n + q / npad, c = q % npad;
            W.Hs[(size_t)r * npad + c] = r == c ? 1.0 : 0.0;
        }
        // ... and the columns beyond n of the rows above it (k_ldlt_reg loads its tiles from the upper block triangle)
        for (int q = threadIdx.x; q < n * (npad - n); q += 256) {
            const int r = q / (npad - n), c = n + q % (npad - n);
            W.Hs[(size_t)r * npad + c] = 0.0;
        }
    }
    if (blockIdx.x == 0) {   // the state record (1.8 KB): zeroed by the workgroup, not by one thread
        LmState *st = W.st;
        static_assert(sizeof(LmState) % 8 == 0, "LmState is cleared as 64-bit words");
        for (int k = threadIdx.x; k < (int)(sizeof(LmState) / 8); k += 256) reinterpret_cast<long long *>(st)[k] = 0;
        __syncthreads();
        if (threadIdx.x == 0) {
            st->iters_max[0] = W.iters1;
            st->iters_max[1] = W.iters2;
            st->polls = 1;   // the entry check of Optimizer.cc:656-658 was made on the host
            st->stop_at_poll = stop_at_poll;
            st->n_active = W.n_edges;
            // SparseOptimizer::optimize(iterations) entry, first call (sparse_optimizer.cpp:354-372): `i < iterations &&
            // !terminate() && ok` before the first iteration
            st->phase = 0;
            st->it = 0;
            if (st->iters_max[0] <= 0 || lm_poll(st, W.abort_word)) {
                st->iters_done[0] = 0;
                lm_first_done(st, W.abort_word);
            } else
                st->initp = 1;
        }
    }
}

// EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ::computeError (types_six_dof_expmap.h:80-141) with the camera point given
__device__ __forceinline__ void edge_error(const Cam &cam, const double p[3], const double *obs, int stereo, double er[3])
{
    if (!stereo) {
        const double u = p[0] / p[2], v = p[1] / p[2];
        er[0] = obs[0] - (u * cam.fx + cam.cx);
        er[1] = obs[1] - (v * cam.fy + cam.cy);
        er[2] = 0;
    } else {
        const float invz = (float)(1.0 / p[2]);
        const double r0 = p[0] * invz * cam.fx + cam.cx;
        const double r1 = p[1] * invz * cam.fy + cam.cy;
        const double r2 = r0 - (double)__fmul_rn(cam.bf_f, invz);
        er[0] = obs[0] - r0;
        er[1] = obs[1] - r1;
        er[2] = obs[2] - r2;
    }
}

// The decision of one Levenberg-Marquardt trial and everything that hangs on it (levenberg.cpp:99-164,
// sparse_optimizer.cpp:372-414): gain ratio, lambda update or pop(), the `while (rho < 0 && qmax < maxTrials &&
// !terminate())` condition, the three ways an iteration can end the optimisation, and the `for (i < iterations &&
// !terminate() && ok)` condition of the next iteration.  Thread 0 decides, all threads
// restore the estimates after a rejected step.  Runs at the end of k_points(solve = 1) in the workgroup that finished
// LAST (no launch of its own): what the other workgroups of this launch wrote -- the per-landmark sums, and on the restore
// path their backups -- is read with device-scope loads.
__device__ __forceinline__ double load_dev(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}
// ... and the matching store: write-through at device scope (what a workgroup hands to the deciding one)
__device__ __forceinline__ void store_dev(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
// chi2 (and the scale terms x_j (lambda x_j + b_j)) of a trial from the per-landmark sums k_points left: value i goes to
// virtual lane i % 128, a lane adds its values in ascending order, a tree adds the 128 lanes -- an order that depends on
// the problem only, not on how the landmark kernels were launched (both landmark-kernel layouts and every batch size give
// the same bits).  Called by all threads of a workgroup of >= 128 threads.
__device__ __forceinline__ void canonical_sums(const LbaWin &W, bool with_scale, double &chi_out, double &scale_out)
{
    __shared__ double s_red[2][128];
    const int tid = threadIdx.x;
    if (tid < 128) {
        double c = 0, s2 = 0;
        for (int i0 = tid; i0 < W.nl; i0 += 128 * 16) {   // 16 values of the lane in flight, added in ascending order
            double v[16], w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = i0 + 128 * u;
                v[u] = i < W.nl ? load_dev(W.part + i) : 0.0;
                w[u] = with_scale && i < W.nl ? load_dev(W.part + W.nl + i) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                c += v[u];     // (x + 0.0 == x past the end)
                s2 += w[u];
            }
        }
        if (with_scale)
            for (int i = tid; i < 6 * W.np; i += 128) s2 += W.tmp[i];
        s_red[0][tid] = c;
        s_red[1][tid] = s2;
    }
    __syncthreads();
    for (int h = 64; h > 0; h >>= 1) {
        if (tid < h) {
            s_red[0][tid] += s_red[0][tid + h];
            s_red[1][tid] += s_red[1][tid + h];
        }
        __syncthreads();
    }
    chi_out = s_red[0][0];
    scale_out = s_red[1][0];
}
template <int NT>
__device__ __forceinline__ void lm_decide(const LbaWin &W)
{
    __shared__ int s_restore;
    LmState *st = W.st;
    // pbStopFlag lives in host memory: its read (a PCIe round trip) travels together with the loads of the sums
    const int flag_seen = threadIdx.x == 0 && W.abort_word ? __hip_atomic_load(W.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
    double tempChi, scale;
#if defined(AOS2_TAIL_TIMING)
    const long long tt0 = wall_clock64();
#endif
    canonical_sums(W, true, tempChi, scale);
#if defined(AOS2_TAIL_TIMING)
    const long long tt1 = wall_clock64();
#endif
    if (threadIdx.x == 0) {
        const int pass = st->phase == 0 ? 0 : 1;
        const bool ok2 = W.np == 0 || W.scal[3] != 0.0;
        if (!ok2) {
            tempChi = 1.7976931348623157e308;
            st->solver_failed++;
        }
        double rho = st->currentChi - tempChi;
        scale += 1e-3;
        rho /= scale;
        if (!ok2) rho = -1.0;   // (currentChi - DBL_MAX) / scale: negative for the positive scale of an LM step
        const bool accepted = rho > 0 && isfinite(tempChi);
        if (st->ntr < 48) {
            st->tr_rho[st->ntr] = rho; st->tr_temp[st->ntr] = tempChi; st->tr_cur[st->ntr] = st->currentChi;
            st->tr_lambda[st->ntr] = st->lambda;
            st->ntr++;
        }
        if (accepted) {
            const double t3 = 2 * rho - 1;
            double alpha = 1. - t3 * t3 * t3;
            alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
            const double scaleFactor = 1. / 3. > alpha ? 1. / 3. : alpha;
            st->lambda *= scaleFactor;
            st->ni = 2;
            st->currentChi = tempChi;
            st->err_at = 1;
        } else {
            st->lambda *= st->ni;
            st->ni *= 2;
            st->err_at = 2;
        }
        s_restore = accepted ? 0 : 1;
        st->qmax++;
        st->trials[pass]++;
        int lin = 0;
        const bool again = rho < 0 && st->qmax < 10 && !lm_poll(st, W.abort_word, flag_seen);
        if (!again) {
            bool term = st->qmax == 10 || rho == 0;
            if (!term) {
                if ((st->iniChi - st->currentChi) * 1e3 < st->iniChi)
                    st->nBad++;
                else
                    st->nBad = 0;
                term = st->nBad >= 3;
            }
            st->it++;
            bool more = st->it < st->iters_max[pass];
            if (more) more = !lm_poll(st, W.abort_word, flag_seen);   // evaluated before `ok`
            if (more) more = !term;
            if (more) {
                st->qmax = 0;
                st->iniChi = st->currentChi;
                lin = accepted ? 1 : 0;   // (a step that was neither accepted nor repeated leaves the system as it is)
            } else {
                st->run = 0;
                st->iters_done[pass] = st->it;
                st->final_chi2 = st->currentChi;
                st->final_lambda = st->lambda;
                if (pass == 0)
                    lm_first_done(st, W.abort_word, flag_seen);
                else
                    st->phase = 3;
            }
        }
        st->lin = lin;
#if defined(AOS2_TAIL_TIMING)
        st->dbg[12] += tt1 - tt0;
        st->dbg[13] += wall_clock64() - tt1 + (flag_seen & 0);
        st->dbg[14] += 1;
#endif
    }
    __syncthreads();
    // pop() after a rejected step (SparseOptimizer::push / pop, sparse_optimizer.cpp:600-610).  The backup holds the
    // estimates a trial starts from: the kernels that move an estimate (the pose update of the reduced-system kernel, the
    // landmark update of k_points) save the old value first -- push() costs no pass of its own.
    // The trial's estimates are kept in the backup's place (a swap): the edges' _error after a rejected trial is the
    // one computed AT the trial (computeActiveErrors ran before pop()), and the passes that read it re-form it from there.
    if (s_restore)
        for (int i = threadIdx.x; i < W.est_n; i += NT) {
            const double trial = load_dev(W.pose + i);
            W.pose[i] = load_dev(W.bk + i);
            W.bk[i] = trial;
        }
}

// end of a k_points launch (solve = 1): the workgroup that finishes last takes the LM decision
template <int NT>
__device__ __forceinline__ void points_tail(const LbaWin &W)
{
    __shared__ int s_last;
    // What this workgroup leaves for the deciding one -- the per-landmark sums and the landmark backups -- was stored
    // write-through at device scope (store_dev), so its visibility needs the stores' completion only (every thread waits
    // for its own, then the barrier), not a write-back of the whole L2: __threadfence() here (buffer_wbl2 by each of the
    // ~1000 workgroups of a 32-window launch) cost 38 of the kernel's 79 us.
    // (Round 4: the workgroup-scope release fence alone does NOT wait for them -- outside tgsplit mode the compiler emits only
    // `s_waitcnt lgkmcnt(0)` for it, and the counter's atomic could reach its L2 channel before a sum's store reached its own:
    // a deciding workgroup then read a stale sum.  Two window groups running side by side made it show -- 4 wrong windows in
    // ~250 batches of 17 on two of three boxes, none in 3.6 M hand-overs of the round-2 stress on a quiet device.  Every thread
    // now waits for the completion of its own write-through stores explicitly.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(&W.st->blocks_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == W.n_part - 1;
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // (the reads below are device-scope loads: load_dev)
    if (threadIdx.x == 0) W.st->blocks_done = 0;
    lm_decide<NT>(W);
}

// J_pose of an edge (linearizeOplus, types_six_dof_expmap.cpp:103-157, 188-234): rows 0-1 (and 2 for stereo) x 6.
// One division per edge: with iz = 1 / z, a = x iz, b = y iz the entries x y / z^2 fx, (1 + x^2 / z^2) fx, y / z fx, ... are
// products (the reference divides ~13 times per edge here and ~9 more in the landmark Jacobian; an f64 division is ~10
// dependent instructions on this part).  Same quantities, equal to rounding (1e-16 relative) -- like the pose-only kernel.
__device__ __forceinline__ void jac_pose(const Cam &cam, const double p[3], int stereo, double Jb[18])
{
    const double fx = cam.fx, fy = cam.fy, bf = cam.bf;
    const double iz = 1.0 / p[2], a = p[0] * iz, b = p[1] * iz, ab = a * b;
    Jb[0] = ab * fx;
    Jb[1] = -(1 + a * a) * fx;
    Jb[2] = b * fx;
    Jb[3] = -(iz * fx);
    Jb[4] = 0;
    Jb[5] = a * iz * fx;
    Jb[6] = (1 + b * b) * fy;
    Jb[7] = -(ab * fy);
    Jb[8] = -(a * fy);
    Jb[9] = 0;
    Jb[10] = -(iz * fy);
    Jb[11] = b * iz * fy;
#pragma unroll
    for (int i = 12; i < 18; ++i) Jb[i] = 0;
    if (stereo) {
        const double bfz2 = bf * (iz * iz);
        Jb[12] = Jb[0] - bfz2 * p[1];
        Jb[13] = Jb[1] + bfz2 * p[0];
        Jb[14] = Jb[2];
        Jb[15] = Jb[3];
        Jb[16] = 0;
        Jb[17] = Jb[5] - bfz2;
    }
}

// J_point of an edge (the same linearizeOplus): -1 / z [fx 0 -x / z fx; 0 fy -y / z fy] R for the two pixel rows (mono and
// stereo edges alike), the stereo row = row 0 - bf / z^2 R_2.  R = the keyframe's rotation (row-major), one division.
__device__ __forceinline__ void jac_point(const Cam &cam, const double R[9], const double p[3], int stereo, double Ja[9])
{
    const double iz = 1.0 / p[2], a = p[0] * iz, b = p[1] * iz;
    const double fxz = cam.fx * iz, fyz = cam.fy * iz, afxz = a * fxz, bfyz = b * fyz;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Ja[c] = afxz * R[6 + c] - fxz * R[c];
        Ja[3 + c] = bfyz * R[6 + c] - fyz * R[3 + c];
        Ja[6 + c] = 0;
    }
    if (stereo) {
        const double bfz2 = cam.bf * (iz * iz);
#pragma unroll
        for (int c = 0; c < 3; ++c) Ja[6 + c] = Ja[c] - bfz2 * R[6 + c];
    }
}

// ---- linearisation records (LbaWin::lrec).  With a = x / z, b = y / z, iz = 1 / z the Jacobians of an edge
// (linearizeOplus, jac_pose / jac_point above) are  J_pose = Pt E,  J_point = -iz Pt R  with
//     Pt = [fx 0 -a fx; 0 fy -b fy; (stereo) fx 0 bf iz - a fx],   E = [ [a b 1]x | -iz I ]  (3 x 6),
// so the edge's Hpl block is  J_pose^T w J_point = -E^T C R,  C = w iz Pt^T Pt  (symmetric 3 x 3, C01 = 0): six numbers
// formed from the record in ~20 operations, and the products with E are cross products with (a, b, 1).
typedef double dbl4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lrec_store(const LbaWin &W, int pos, const double p[3], double wo, int stereo, const double omr[3])
{
    const double iz = 1.0 / p[2];
    const dbl4_t v = {p[0] * iz, p[1] * iz, iz, stereo ? -wo : wo};
    *reinterpret_cast<dbl4_t *>(W.lrec + 4 * (size_t)pos) = v;
#pragma unroll
    for (int i = 0; i < 3; ++i) W.lomr[3 * (size_t)pos + i] = omr[i];
}
__device__ __forceinline__ void lrec_mask(const LbaWin &W, int pos)
{
    const dbl4_t v = {0.0, 0.0, 0.0, 0.0};
    *reinterpret_cast<dbl4_t *>(W.lrec + 4 * (size_t)pos) = v;
#pragma unroll
    for (int i = 0; i < 3; ++i) W.lomr[3 * (size_t)pos + i] = 0.0;
}
__device__ __forceinline__ dbl4_t lrec_load(const LbaWin &W, int pos) { return *reinterpret_cast<const dbl4_t *>(W.lrec + 4 * (size_t)pos); }
struct EdgeLin {
    double a, b, iz;
    double C00, C02, C11, C12, C22;
};
__device__ __forceinline__ EdgeLin lrec_form(const Cam &cam, const dbl4_t rec)
{
    EdgeLin L;
    L.a = rec.x; L.b = rec.y; L.iz = rec.z;
    const bool stereo = rec.w < 0.0;
    const double s = fabs(rec.w) * L.iz;
    const double fx2 = cam.fx * cam.fx, fy2 = cam.fy * cam.fy;
    // (a mono edge adds exact zeros for the third row)
    const double fs = stereo ? cam.fx : 0.0, c = stereo ? cam.bf * L.iz - L.a * cam.fx : 0.0;
    L.C00 = (fx2 + fs * fs) * s;
    L.C02 = (fs * c - L.a * fx2) * s;
    L.C11 = fy2 * s;
    L.C12 = -(L.b * fy2) * s;
    L.C22 = (L.a * L.a * fx2 + L.b * L.b * fy2 + c * c) * s;
    return L;
}
// C t
__device__ __forceinline__ void lrec_C(const EdgeLin &L, const double t[3], double h[3])
{
    h[0] = L.C00 * t[0] + L.C02 * t[2];
    h[1] = L.C11 * t[1] + L.C12 * t[2];
    h[2] = L.C02 * t[0] + L.C12 * t[1] + L.C22 * t[2];
}
// J_pose^T omr = E^T (Pt^T omr) of a free-keyframe edge: its share of b_p (constructQuadraticForm, base_binary_edge.hpp:95-110), from the record
// and the omr the landmark side left (lomr): E^T g = (-(a, b, 1) x g, -iz g)
__device__ __forceinline__ void lrec_bp(const Cam &cam, const dbl4_t rec, const double o[3], double (&acc)[6])
{
    const double a = rec.x, b = rec.y, iz = rec.z;
    const bool stereo = rec.w < 0.0;
    const double fs = stereo ? cam.fx : 0.0, c = stereo ? cam.bf * iz - a * cam.fx : 0.0;
    const double g0 = cam.fx * o[0] + fs * o[2], g1 = cam.fy * o[1], g2 = c * o[2] - a * cam.fx * o[0] - b * cam.fy * o[1];
    acc[0] += g1 - b * g2;
    acc[1] += a * g2 - g0;
    acc[2] += b * g0 - a * g1;
    acc[3] += -(iz * g0);
    acc[4] += -(iz * g1);
    acc[5] += -(iz * g2);
}
// B^T xp of a free-keyframe edge (block_solver.hpp:455-480 multiplies by the stored Hpl block B):
// B^T xp = -R^T C E xp,  E xp = (a, b, 1) x omega - iz upsilon  for xp = (omega, upsilon)
__device__ __forceinline__ void lrec_backsub(const Cam &cam, const dbl4_t rec, const double R[9], const double xp[6], double v[3])
{
    const EdgeLin L = lrec_form(cam, rec);
    const double t[3] = {L.b * xp[2] - xp[1] - L.iz * xp[3], xp[0] - L.a * xp[2] - L.iz * xp[4], L.a * xp[1] - L.b * xp[0] - L.iz * xp[5]};
    double h[3];
    lrec_C(L, t, h);
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = -(R[c] * h[0] + R[3 + c] * h[1] + R[6 + c] * h[2]);
}

// Landmark kernels: a 256-thread workgroup takes kLmBlock landmarks, kLmSlots threads each -- thread (landmark, slot j)
// works on the landmark's edges j, j + kLmSlots, ... (one edge for the usual <= 8 observations) and the landmark's first
// thread adds the per-edge terms in edge order, so the sums are the ones a thread walking the edges one after the other
// would form (the order g2o adds them in), while the chain of dependent gathers per thread is one edge long instead of
// the whole observation list (a single window has only ~2000 landmarks: the walk was pure latency).
// Timing-only ablation builds (tools/build_lba_abl_libs.sh; results are wrong by construction): one phase switched off at a
// time under rocprofv3 -- 1: k_points_walk without the deciding tail, 2: without the residual loop, 3: without the
// back-substitution loop, 5: k_lin landmark blocks only, 6: k_lin keyframe blocks only.  (DESIGN.md 5.3's table)
#ifndef AOS2_LBA_ABL
#define AOS2_LBA_ABL 0
#endif
constexpr int kLmBlock = 32, kLmSlots = 8;
// edges of a landmark fetched together by the one-thread-per-landmark kernels (measured on 32 windows of 24 k edges, 6
// observations per landmark: k_points_walk 43.4 us with 4 / 4, 38.6 with 6 / 6; the linearisation spills beyond 4)
#ifndef AOS2_WALK_CHUNK
#define AOS2_WALK_CHUNK 4
#endif
constexpr int kWalkChunk = AOS2_WALK_CHUNK;   // free-keyframe edges (back-substitution)
constexpr int kWalkChunkE = 6;     // all edges (residuals)
constexpr int kWalkChunkLin = 4;   // all edges (linearisation)

// solve = 1 (a Levenberg-Marquardt trial): the landmark's part of
// BlockSolver::solve -- xl = (Hll + lambda I)^-1 (bl - B^T xp), block_solver.hpp:455-480 -- and of
// SparseOptimizer::update (X += xl), its scale terms x_j (lambda x_j + b_j), then computeActiveErrors +
// activeRobustChi2 (sparse_optimizer.cpp:61-114) of the landmark's edges at the new estimates (the poses were
// updated by the kernel before), and -- in the workgroup that finishes last -- the LM decision (lm_decide).
// solve = 0 (top of solve() in iteration 0): the residuals only.
// Landmark l leaves its chi2 terms in part[l] and its scale terms in part[nl + l] (canonical_sums adds them).
__global__ __launch_bounds__(256) void k_points(const LbaWin *__restrict__ wins, const SchurTask *__restrict__ tasks, int solve)
{
    lba_wave_prio();
    __shared__ double s_v[kLmBlock][kLmSlots][3];
    __shared__ double s_X[kLmBlock][3];
    const SchurTask tk = tasks[blockIdx.x];
    const LbaWin &W = wins[tk.w];
    if (!(solve ? W.st->run : W.st->initp)) return;
    const int tid = threadIdx.x, ll = tid / kLmSlots, j = tid % kLmSlots;
    const int l = tk.code * kLmBlock + ll;
    const bool has = l < W.nl, leader = j == 0;
    const int n6 = 6 * W.np;
    double sc = 0, chi = 0;
    double *X = has ? W.point + 3 * (size_t)W.hpoint[l] : nullptr;
    double Xv[3] = {0, 0, 0};
    if (has)
        for (int r = 0; r < 3; ++r) Xv[r] = X[r];
    if (solve) {
        const double lambda = W.st->lambda;
        const int a0 = has ? W.pl_off[l] : 0, a1 = has ? W.pl_off[l + 1] : 0;
        double cl[3] = {0, 0, 0};
        if (has && leader)
            for (int r = 0; r < 3; ++r) cl[r] = W.b[n6 + 3 * l + r];
        for (int rd = 0; __syncthreads_or(a0 + kLmSlots * rd < a1); ++rd) {
            const int a = a0 + kLmSlots * rd + j;
            double v[3] = {0, 0, 0};
            if (a < a1) {   // B_i^T (-x_p) of one free-keyframe edge
                const int i1 = W.pl_ph[a];
                const dbl4_t rec = lrec_load(W, a);
                double xp[6], R[9];
#pragma unroll
                for (int r = 0; r < 6; ++r) xp[r] = -W.x[6 * i1 + r];
#pragma unroll
                for (int i = 0; i < 9; ++i) R[i] = W.Rl[9 * (size_t)i1 + i];
                lrec_backsub(W.cam, rec, R, xp, v);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) s_v[ll][j][c] = v[c];
            __syncthreads();
            if (leader) {
                const int m = min(kLmSlots, a1 - (a0 + kLmSlots * rd));
                for (int jj = 0; jj < m; ++jj)
                    for (int c = 0; c < 3; ++c) cl[c] += s_v[ll][jj][c];
            }
        }
        if (has && leader) {
            // (Hll + lambda I)^-1: the same operations as in the Schur kernel, so the same bits
            double Dm[9], Dinv[9];
            for (int i = 0; i < 9; ++i) Dm[i] = W.Hll[9 * (size_t)l + i];
            Dm[0] += lambda; Dm[4] += lambda; Dm[8] += lambda;
            mat3_inverse(Dm, Dinv);
            double *Xb = W.bk + 7 * (size_t)W.n_poses + 3 * (size_t)W.hpoint[l];
            for (int r = 0; r < 3; ++r) {
                const double xl = Dinv[r * 3] * cl[0] + Dinv[r * 3 + 1] * cl[1] + Dinv[r * 3 + 2] * cl[2];
                W.x[n6 + 3 * l + r] = xl;
                store_dev(Xb + r, Xv[r]);   // push() (read back by the deciding workgroup when the step is rejected)
                Xv[r] += xl;
                store_dev(X + r, Xv[r]);   // (read back by the deciding workgroup when the step is rejected: the swap)
                s_X[ll][r] = Xv[r];
                sc += xl * (lambda * xl + W.b[n6 + 3 * l + r]);
            }
        }
        __syncthreads();
        if (has)
            for (int r = 0; r < 3; ++r) Xv[r] = s_X[ll][r];
    }
    {
        const int e0 = has ? W.pt_off[l] : 0, e1 = has ? W.pt_off[l + 1] : 0;
        for (int rd = 0; __syncthreads_or(e0 + kLmSlots * rd < e1); ++rd) {
            const int a = e0 + kLmSlots * rd + j;
            double c = 0;
            if (a < e1) {
                const int e = W.pt_k[a];
                if (!W.e_level1[e]) {   // an inactive edge keeps its _error (and adds nothing: chi2 >= 0, so + 0.0 is exact)
                    double p[3], er[3];
                    se3_map(W.pose + 7 * (size_t)W.e_pose[e], Xv, p);
                    const int stereo = W.e_stereo[e];
                    const double ob[3] = {(double)W.in_obs[3 * (size_t)e], (double)W.in_obs[3 * (size_t)e + 1], (double)W.in_obs[3 * (size_t)e + 2]};
                    edge_error(W.cam, p, ob, stereo, er);   // (_error is not stored: LmState::err_at)
                    c = edge_chi2(er, (double)W.in_w[e], stereo ? 3 : 2);
                    if (W.e_robust[e]) {
                        double rho[2];
                        robustify(c, stereo ? W.cam.delta_stereo : W.cam.delta_mono, rho);
                        c = rho[0];
                    }
                }
            }
            s_v[ll][j][0] = c;
            __syncthreads();
            if (leader) {
                const int m = min(kLmSlots, e1 - (e0 + kLmSlots * rd));
                for (int jj = 0; jj < m; ++jj) chi += s_v[ll][jj][0];
            }
        }
    }
    if (has && leader) {
        store_dev(W.part + l, chi);
        store_dev(W.part + W.nl + l, sc);
    }
    if (solve) points_tail<256>(W);
}

// The same work with one thread per landmark walking its edges (128-thread workgroups): the layout for many windows per
// launch, where the landmarks alone fill the device and idle slots, barriers and LDS round trips only cost (32 windows of
// 24 k edges: 4.4 ms against 6.4 ms).  Per landmark the operations and their order are those of k_points, so both layouts
// leave the same bits.
__global__ __launch_bounds__(128) void k_points_walk(const LbaWin *__restrict__ wins, const SchurTask *__restrict__ tasks, int solve)
{
    lba_wave_prio();
    const SchurTask tk = tasks[blockIdx.x];
    const LbaWin &W = wins[tk.w];
    if (!(solve ? W.st->run : W.st->initp)) return;
    const int l = tk.code * blockDim.x + threadIdx.x;
    const int n6 = 6 * W.np;
    if (l < W.nl) {
        double sc = 0, chi = 0;
        double *X = W.point + 3 * (size_t)W.hpoint[l];
        double Xv[3] = {X[0], X[1], X[2]};
        if (solve) {
            const double lambda = W.st->lambda;
            double cl[3] = {W.b[n6 + 3 * l], W.b[n6 + 3 * l + 1], W.b[n6 + 3 * l + 2]};
            // The walk is a chain of dependent gathers (edge list -> edge -> keyframe); kWalkChunk edges are fetched level by
            // level together, then their terms are added in edge order (the same sums, a quarter of the round trips).
#if AOS2_LBA_ABL == 3
            const int a0 = 0, a1 = 0;
#else
            const int a0 = W.pl_off[l], a1 = W.pl_off[l + 1];
#endif
            for (int a = a0; a < a1; a += kWalkChunk) {
                int ka[kWalkChunk], i1[kWalkChunk];   // (positions in the list = the indices of the dense Hpl array)
#pragma unroll
                for (int u = 0; u < kWalkChunk; ++u) ka[u] = min(a + u, a1 - 1);
#pragma unroll
                for (int u = 0; u < kWalkChunk; ++u) i1[u] = W.pl_ph[ka[u]];
                dbl4_t rec[kWalkChunk];
                double xp[kWalkChunk][6], Rr[kWalkChunk][9];
#pragma unroll
                for (int u = 0; u < kWalkChunk; ++u) rec[u] = lrec_load(W, ka[u]);
#pragma unroll
                for (int u = 0; u < kWalkChunk; ++u) {
#pragma unroll
                    for (int r = 0; r < 6; ++r) xp[u][r] = -W.x[6 * i1[u] + r];
#pragma unroll
                    for (int i = 0; i < 9; ++i) Rr[u][i] = W.Rl[9 * (size_t)i1[u] + i];
                }
#pragma unroll
                for (int u = 0; u < kWalkChunk; ++u) {
                    if (a + u >= a1) break;
                    double v[3];
                    lrec_backsub(W.cam, rec[u], Rr[u], xp[u], v);
                    for (int c = 0; c < 3; ++c) cl[c] += v[c];
                }
            }
            double Dm[9], Dinv[9];
            for (int i = 0; i < 9; ++i) Dm[i] = W.Hll[9 * (size_t)l + i];
            Dm[0] += lambda; Dm[4] += lambda; Dm[8] += lambda;
            mat3_inverse(Dm, Dinv);
            double *Xb = W.bk + 7 * (size_t)W.n_poses + 3 * (size_t)W.hpoint[l];
            for (int r = 0; r < 3; ++r) {
                const double xl = Dinv[r * 3] * cl[0] + Dinv[r * 3 + 1] * cl[1] + Dinv[r * 3 + 2] * cl[2];
                W.x[n6 + 3 * l + r] = xl;
                store_dev(Xb + r, Xv[r]);   // push()
                Xv[r] += xl;
                store_dev(X + r, Xv[r]);
                sc += xl * (lambda * xl + W.b[n6 + 3 * l + r]);
            }
        }
#if AOS2_LBA_ABL == 2
        const int e0 = 0, e1 = 0;
#else
        const int e0 = W.pt_off[l], e1 = W.pt_off[l + 1];
#endif
        for (int a = e0; a < e1; a += kWalkChunkE) {
            int e[kWalkChunkE], ep[kWalkChunkE];
            uint8_t lv1[kWalkChunkE], ste[kWalkChunkE], rob[kWalkChunkE];
            double T[kWalkChunkE][7], ob[kWalkChunkE][3], ew[kWalkChunkE];
#pragma unroll
            for (int u = 0; u < kWalkChunkE; ++u) e[u] = W.pt_k[min(a + u, e1 - 1)];
#pragma unroll
            for (int u = 0; u < kWalkChunkE; ++u) {
                ep[u] = W.e_pose[e[u]];
                lv1[u] = W.e_level1[e[u]];
                ste[u] = W.e_stereo[e[u]];
                rob[u] = W.e_robust[e[u]];
                ew[u] = (double)W.in_w[e[u]];
#pragma unroll
                for (int i = 0; i < 3; ++i) ob[u][i] = (double)W.in_obs[3 * (size_t)e[u] + i];
            }
#pragma unroll
            for (int u = 0; u < kWalkChunkE; ++u)
#pragma unroll
                for (int i = 0; i < 7; ++i) T[u][i] = W.pose[7 * (size_t)ep[u] + i];
#pragma unroll
            for (int u = 0; u < kWalkChunkE; ++u) {
                if (a + u >= e1) break;
                if (lv1[u]) continue;   // an inactive edge keeps its _error
                double p[3], er[3];
                se3_map(T[u], Xv, p);
                const int stereo = ste[u];
                edge_error(W.cam, p, ob[u], stereo, er);   // (_error is not stored: LmState::err_at)
                double c = edge_chi2(er, ew[u], stereo ? 3 : 2);
                if (rob[u]) {
                    double rho[2];
                    robustify(c, stereo ? W.cam.delta_stereo : W.cam.delta_mono, rho);
                    c = rho[0];
                }
                chi += c;
            }
        }
        store_dev(W.part + l, chi);
        store_dev(W.part + W.nl + l, sc);
    }
#if AOS2_LBA_ABL != 1
    if (solve) points_tail<128>(W);
#endif
}

// robustified information of an edge (BaseBinaryEdge::constructQuadraticForm, base_binary_edge.hpp:55-120):
// omr = -rho' Omega e, wo = rho' * w
__device__ __forceinline__ void edge_weights_of(const Cam &cam, const double er[3], double w, int robust, int stereo, double omr[3], double &wo)
{
    // static indices only (a runtime-length loop over D would put the arrays in scratch memory)
    omr[0] = -(w * er[0]);
    omr[1] = -(w * er[1]);
    omr[2] = stereo ? -(w * er[2]) : 0.0;
    wo = w;
    if (robust) {
        double rho[2];
        robustify(edge_chi2(er, w, stereo ? 3 : 2), stereo ? cam.delta_stereo : cam.delta_mono, rho);
        wo = rho[1] * w;
        omr[0] *= rho[1];
        omr[1] *= rho[1];
        if (stereo) omr[2] *= rho[1];
    }
}
// ... of edge k whose landmark maps to p in its keyframe: the residual is re-formed from p and the observation (the same
// operations on the same values as the residual pass that stored _error: the same bits, for 12 bytes of observation instead
// of 24 of stored residual)
__device__ __forceinline__ void edge_weights(const LbaWin &W, int k, const double p[3], int stereo, double omr[3], double &wo)
{
    const double ob[3] = {(double)W.in_obs[3 * (size_t)k], (double)W.in_obs[3 * (size_t)k + 1], (double)W.in_obs[3 * (size_t)k + 2]};
    double er[3];
    edge_error(W.cam, p, ob, stereo, er);
    edge_weights_of(W.cam, er, (double)W.in_w[k], W.e_robust[k], stereo, omr, wo);
}

// buildSystem, the landmarks' side (block_solver.hpp:502-560), kLmBlock landmarks per workgroup (see k_points): thread
// (landmark, slot) linearises one edge at a time -- linearizeOplus, Ji^T Omega Ji, Ji^T omr, and the edge's Hpl block
// Jj^T Omega Ji -- and the landmark's first thread adds the terms in insertion order like g2o: Hll, b_l.
// A masked edge adds nothing (its Hpl block was zeroed when it was masked).
__device__ __forceinline__ void lin_points_body(const LbaWin &W, int blk)
{
    __shared__ double s_c[kLmBlock][kLmSlots][12 + 1];
    const int tid = threadIdx.x, ll = tid / kLmSlots, j = tid % kLmSlots;
    const int l = blk * kLmBlock + ll;
    const bool has = l < W.nl, leader = j == 0;
    double Xv[3] = {0, 0, 0};
    if (has) {
        const double *X = W.point + 3 * (size_t)W.hpoint[l];
        for (int r = 0; r < 3; ++r) Xv[r] = X[r];
    }
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
    const int a0 = has ? W.pt_off[l] : 0, a1 = has ? W.pt_off[l + 1] : 0;
    for (int rd = 0; __syncthreads_or(a0 + kLmSlots * rd < a1); ++rd) {
        const int a = a0 + kLmSlots * rd + j;
        double cH[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, cb[3] = {0, 0, 0};
        const int k = a < a1 ? W.pt_k[a] : -1;
        if (k >= 0 && !W.e_level1[k]) {
            const double *T = W.pose + 7 * (size_t)W.e_pose[k];
            const int stereo = W.e_stereo[k];
            double p[3], R[9];
            se3_map(T, Xv, p);
            rot_from_quat(T, R);
            double Ja[9];
            jac_point(W.cam, R, p, stereo, Ja);
            double omr[3], wo;
            edge_weights(W, k, p, stereo, omr, wo);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                cb[r] = Ja[r] * omr[0] + Ja[3 + r] * omr[1] + Ja[6 + r] * omr[2];
#pragma unroll
                for (int c = 0; c < 3; ++c) cH[r * 3 + c] = Ja[r] * wo * Ja[c] + Ja[3 + r] * wo * Ja[3 + c] + Ja[6 + r] * wo * Ja[6 + c];
            }
            const int pp = W.pl_pos[k];
            if (pp >= 0) lrec_store(W, pp, p, wo, stereo, omr);
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) s_c[ll][j][i] = cH[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) s_c[ll][j][9 + i] = cb[i];
        __syncthreads();
        if (leader) {
            const int m = min(kLmSlots, a1 - (a0 + kLmSlots * rd));
            for (int jj = 0; jj < m; ++jj) {
#pragma unroll
                for (int i = 0; i < 9; ++i) H[i] += s_c[ll][jj][i];
#pragma unroll
                for (int i = 0; i < 3; ++i) bl[i] += s_c[ll][jj][9 + i];
            }
        }
    }
    if (has && leader) {
        for (int i = 0; i < 9; ++i) W.Hll[9 * (size_t)l + i] = H[i];
        for (int i = 0; i < 3; ++i) W.b[6 * (size_t)W.np + 3 * (size_t)l + i] = bl[i];
    }
}

// the same with one thread per landmark walking its edges (see k_points_walk); the per-edge terms are formed and added in
// the same order, so Hll, b_l and Hpl carry the same bits
__device__ __forceinline__ void lin_points_walk(const LbaWin &W, int l)
{
    if (l >= W.nl) return;
    const double *X = W.point + 3 * (size_t)W.hpoint[l];
    const double Xv[3] = {X[0], X[1], X[2]};
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
    const int e0 = W.pt_off[l], e1 = W.pt_off[l + 1];
    for (int a = e0; a < e1; a += kWalkChunkLin) {
        // kWalkChunkLin edges fetched level by level together (see k_points_walk), linearised and added in edge order
        int kk[kWalkChunkLin], ep[kWalkChunkLin], ph[kWalkChunkLin];
        uint8_t lv1[kWalkChunkLin], ste[kWalkChunkLin], rob[kWalkChunkLin];
        double T[kWalkChunkLin][7], er[kWalkChunkLin][3], ew[kWalkChunkLin];
#pragma unroll
        for (int u = 0; u < kWalkChunkLin; ++u) kk[u] = W.pt_k[min(a + u, e1 - 1)];
#pragma unroll
        for (int u = 0; u < kWalkChunkLin; ++u) {
            ep[u] = W.e_pose[kk[u]];
            ph[u] = W.pl_pos[kk[u]];
            lv1[u] = W.e_level1[kk[u]];
            ste[u] = W.e_stereo[kk[u]];
            rob[u] = W.e_robust[kk[u]];
            ew[u] = (double)W.in_w[kk[u]];
#pragma unroll
            for (int i = 0; i < 3; ++i) er[u][i] = (double)W.in_obs[3 * (size_t)kk[u] + i];   // (the observation; the residual is re-formed below)
        }
#pragma unroll
        for (int u = 0; u < kWalkChunkLin; ++u)
#pragma unroll
            for (int i = 0; i < 7; ++i) T[u][i] = W.pose[7 * (size_t)ep[u] + i];
#pragma unroll
        for (int u = 0; u < kWalkChunkLin; ++u) {
            if (a + u >= e1) break;
            if (lv1[u]) continue;
            const int stereo = ste[u];
            double p[3], R[9];
            se3_map(T[u], Xv, p);
            rot_from_quat(T[u], R);
            double Ja[9];
            jac_point(W.cam, R, p, stereo, Ja);
            double omr[3], wo;
            double res[3];
            edge_error(W.cam, p, er[u], stereo, res);
            edge_weights_of(W.cam, res, ew[u], rob[u], stereo, omr, wo);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                bl[r] += Ja[r] * omr[0] + Ja[3 + r] * omr[1] + Ja[6 + r] * omr[2];
#pragma unroll
                for (int c = 0; c < 3; ++c) H[r * 3 + c] += Ja[r] * wo * Ja[c] + Ja[3 + r] * wo * Ja[3 + c] + Ja[6 + r] * wo * Ja[6 + c];
            }
            if (ph[u] >= 0) lrec_store(W, ph[u], p, wo, stereo, omr);
        }
    }
    for (int i = 0; i < 9; ++i) W.Hll[9 * (size_t)l + i] = H[i];
    for (int i = 0; i < 3; ++i) W.b[6 * (size_t)W.np + 3 * (size_t)l + i] = bl[i];
}

// Sum of `v` over the 16 lanes of a DPP row (xor butterfly: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror):
// VALU speed, no LDS.  Every lane of the row ends with the row's total (lanes may differ in the last bit: the butterfly
// adds in a lane-dependent order; callers read one fixed lane per row).
template <int kCtrl>
__device__ __forceinline__ double dpp_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)b, (int)b, kCtrl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(b >> 32), (int)(b >> 32), kCtrl, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double row_sum_f64(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return v;
}

// acc[K] of every thread of an NT-thread workgroup -> their sums (returned in threads e < K).  16-lane DPP row sums, then
// the row totals of each value through LDS in row order: a fixed order (bit-reproducible), NT / 16 x K doubles of LDS.
// n_items = work items of the workgroup (thread t had one iff t < n_items): waves without any skip their share, and rows
// without any are left out of the final sums (they would add +0.0).
template <int K, int NT = 256>
__device__ __forceinline__ double workgroup_sum_k256(double (&acc)[K], double *red /* (NT / 16) x (K + 1) */, int n_items)
{
    const int lane = threadIdx.x & 63, row = threadIdx.x >> 4, wave = threadIdx.x >> 6;
    const int rows_used = min(NT / 16, (n_items + 15) >> 4);
    if (wave * 64 < n_items) {   // wave-uniform
#pragma unroll
        for (int i = 0; i < K; ++i) acc[i] = row_sum_f64(acc[i]);
        if ((lane & 15) == 0) {
#pragma unroll
            for (int i = 0; i < K; ++i) red[row * (K + 1) + i] = acc[i];
        }
    }
    __syncthreads();
    double sum = 0;
    if ((int)threadIdx.x < K)
        for (int r = 0; r < rows_used; ++r) sum += red[r * (K + 1) + threadIdx.x];
    return sum;
}

// buildSystem, the keyframes' side: Hpp += Jj^T Omega Jj, b_p += Jj^T omr.  One 256-thread workgroup per free pose:
// thread j takes the pose's edges j, j + 256, ... (the Jacobian is recomputed from the estimates: no per-edge arrays;
// few edges per thread keep the chain of dependent gathers short), then workgroup_sum_k256 (fixed order).
// (The former 256 x 43 tree in LDS held 88 KB per pose -- one workgroup per compute unit.)
__device__ __forceinline__ void lin_poses_body(const LbaWin &W, int ph, int init)
{
    __shared__ double red[16 * 43];
    if (ph >= W.np) return;
    double T[7];
    {
        const double *Tp = W.pose + 7 * (size_t)W.hpose[ph];
#pragma unroll
        for (int i = 0; i < 7; ++i) T[i] = Tp[i];
    }
    if (threadIdx.x == 0) {   // the rotation the landmark side linearises at (the same function of the same values)
        double R[9];
        rot_from_quat(T, R);
#pragma unroll
        for (int i = 0; i < 9; ++i) W.Rl[9 * (size_t)ph + i] = R[i];
    }
    // Hpp and b_p themselves are formed by k_schur's diagonal units from the landmark side's records (schur_item<true>, lrec_bp); only the
    // first linearisation of an optimisation needs them here: k_lm_init takes lambda from the diagonal of Hpp before any Schur launch
    if (!init) return;
    double acc[42];
#pragma unroll
    for (int i = 0; i < 42; ++i) acc[i] = 0;
    for (int a = W.ps_off[ph] + threadIdx.x; a < W.ps_off[ph + 1]; a += 256) {
        const int k = W.ps_k[a];
        if (W.e_level1[k]) continue;
        const int stereo = W.e_stereo[k];
        double p[3], Jb[18], omr[3], wo;
        se3_map(T, W.point + 3 * (size_t)W.e_point[k], p);
        jac_pose(W.cam, p, stereo, Jb);
        edge_weights(W, k, p, stereo, omr, wo);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            acc[36 + r] += Jb[r] * omr[0] + Jb[6 + r] * omr[1] + Jb[12 + r] * omr[2];
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[r * 6 + c] += Jb[r] * wo * Jb[c] + Jb[6 + r] * wo * Jb[6 + c] + Jb[12 + r] * wo * Jb[12 + c];
        }
    }
    const double sum = workgroup_sum_k256<42>(acc, red, W.ps_off[ph + 1] - W.ps_off[ph]);
    if (threadIdx.x < 36)
        W.Hpp[36 * (size_t)ph + threadIdx.x] = sum;
    else if (threadIdx.x < 42)
        W.b[6 * (size_t)ph + (threadIdx.x - 36)] = sum;
}

// buildSystem as ONE launch: the landmark side and the keyframe side are independent of each other.  The task list holds the
// landmark blocks of all windows (block by block across the windows), then one task per free keyframe, so the landmark
// workgroups of ALL windows start before any keyframe workgroup: their walks are the long
// dependent chains of the launch (33 of its 54 us on 32 windows when they queued behind the keyframe workgroups of the
// windows before them), the keyframe workgroups fill in behind.
// (three waves per SIMD: 174 -> 168 registers for 12 bytes of scratch per lane; the walk moves scattered bytes and gains from the third
// wave: 36.2 -> 34.4 us on 32 windows.  k_points_walk, 172 registers, does not: 34.2 -> 35.0 us, left at two.)
template <bool kWalk>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_lin(const LbaWin *__restrict__ wins, const SchurTask *__restrict__ tasks, int init)
{
    lba_wave_prio();
    const SchurTask tk = tasks[blockIdx.x];
    const LbaWin &W = wins[tk.w];
    if (!(init ? W.st->initp : W.st->lin)) return;
    const int blk = tk.code & 0x0fffffff;
    const bool keyframe = (tk.code >> 28) != 0;
#if AOS2_LBA_ABL == 5
    if (keyframe) return;
#elif AOS2_LBA_ABL == 6
    if (!keyframe) return;
#endif
    if (keyframe)
        lin_poses_body(W, blk, init);
    else if (kWalk)
        lin_points_walk(W, blk * 256 + threadIdx.x);
    else
        lin_points_body(W, blk);
}

// top of solve() in iteration 0 (levenberg.cpp:75-97): currentChi, lambda = 1e-5 * max |H_jj| over all free vertices
// (computeLambdaInit :166-180), ni = 2; the first trial's push() (the backup of the estimates)
__global__ __launch_bounds__(1024) void k_lm_init(const LbaWin *__restrict__ wins)
{
    lba_wave_prio();
    __shared__ double sh[1024];
    const LbaWin &W = wins[blockIdx.x];
    LmState *st = W.st;
    if (!st->initp) return;
    const int n6 = 6 * W.np, n = n6 + 3 * W.nl;
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        double v;
        if (i < n6)
            v = W.Hpp[36 * (size_t)(i / 6) + (i % 6) * 7];
        else {
            const int j = i - n6;
            v = W.Hll[9 * (size_t)(j / 3) + (j % 3) * 4];
        }
        acc = fmax(acc, fabs(v));
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < W.est_n; i += 1024) W.bk[i] = W.pose[i];
    double chi, unused;
    canonical_sums(W, false, chi, unused);
    if (threadIdx.x == 0) {
        st->currentChi = st->iniChi = chi;
        st->lambda = 1e-5 * sh[0];
        st->ni = 2;
        st->nBad = 0;
        st->it = 0;
        st->qmax = 0;
        st->initp = 0;
        st->lin = 0;
        st->run = 1;
        st->err_at = 1;   // (the residual pass before this launch)
    }
}

// ---- Schur complement (block_solver.hpp:379-432), one 256-thread workgroup per (pose, pose) block of the upper block
// triangle.  The host ranks the items -- (landmark, free-pose edges ka <= kb of it) -- by their block, landmark order
// inside a block (build_schur_items).  Thread j of the block takes the items j, j + 256, ...: (Hll + lambda I)^-1 of the
// landmark, B_a Dinv B_b^T (and, on the diagonal blocks, the coefficient term B_a Dinv b_l), accumulated in registers;
// the partial 6x6 sums are added by workgroup_sum_k256: no atomics, a fixed order (bit-reproducible), and no per-item
// array in memory (a materialised item list cost 288 B written + read per
// item: 0.31 ms per trial for 32 windows of 24 k edges).  Hschur = Hpp + lambda I - sum, bschur = b_p - sum.
// threads per block: measured (one 12 k-edge window / one 24 k-edge window / 32 windows of 24 k edges, whole solve):
// 64: 1.88 / 2.27 / 4.42 ms, 128: 1.74 / 1.92 / 4.35 ms, 256: 1.71 / 1.87 / 4.78 ms (most off-diagonal blocks hold < 128 items)
constexpr int kSchurThreads = 256;
// Work units of k_schur, one 256-thread workgroup each (build_schur_units ranks them, the host lays the units of all windows of
// the call out as ONE task list -- no grid padded to the largest window, no workgroup that only finds out it has nothing to do):
//   DIAG  a diagonal (pose, pose) block: every observation of a keyframe, hundreds to thousands of items, strided over the
//         256 threads, sums through workgroup_sum_k256;
//   BIG   an off-diagonal block of more than 256 items, the same way;
//   PACK  16 ROWS of 16 lanes; a row holds up to 16 consecutive items of ONE off-diagonal block (a block of n items takes
//         ceil(n / 16) consecutive rows of one unit), one item per thread: the covisible landmarks of most keyframe pairs are a
//         few dozen, often fewer than 16 -- one wave per block left 40-85 % of the lanes idle and a window of 40 free keyframes
//         cost 235 workgroups.  Row sums by DPP, then a block's rows are added in row order: for a block of up to 64 items
//         exactly the sums (and bits) of the one-wave-per-block form.
constexpr int kSchurDiag = 0, kSchurBig = 1, kSchurPack = 2;
// (SchurTask.code = kind << 28 | argument -- DIAG: i; BIG: block rank; PACK: first row; w = -1: padding of the XCD interleave)
// one item: (Hll + lambda I)^-1 of the landmark, and B_a D^-1 B_b^T in factored form.  With B = -E^T C R (lrec_form)
//     B_a D^-1 B_b^T = E_a^T Q E_b,   Q = C_a (R_a D^-1 R_b^T) C_b   (3 x 3),
// and the products with E = [ [a b 1]x | -iz I ] are cross products: the two 144-byte blocks are neither stored nor fetched --
// each side is its 32-byte record (LbaWin::lrec) and the keyframe's rotation (uniform over the block: a broadcast load) --
// and an item takes ~360 operations instead of the ~350 of the stored-block form plus 23 divergent 16-byte requests (9 now).
#ifndef AOS2_SCHUR_FMA
#define AOS2_SCHUR_FMA 0
#endif
__device__ __forceinline__ double sf(double a, double b, double c)
{
#if AOS2_SCHUR_FMA
    return __builtin_fma(a, b, c);
#else
    return a * b + c;
#endif
}
template <bool kDiag>
__device__ __forceinline__ void schur_item(const LbaWin &W, int j, double lambda, int n6, const double *__restrict__ Ra_g,
                                           const double *__restrict__ Rb_g, double (&acc)[42])
{
    const int ka = W.it_ka[j], kb = W.it_kb[j], l = W.it_l[j];   // three independent loads, then one level of gathers
    const dbl4_t rb = lrec_load(W, kb);
    const dbl4_t ra = kDiag ? rb : lrec_load(W, ka);             // (diagonal block: the two edges of an item are one)
    double D[9], Dinv[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) D[i] = W.Hll[9 * (size_t)l + i];
    D[0] += lambda; D[4] += lambda; D[8] += lambda;
    mat3_inverse(D, Dinv);
    double Ra[9], G[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Ra[i] = Ra_g[i];
    {
        double G0[9];   // R_a D^-1
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) G0[r * 3 + c] = sf(Ra[r * 3 + 2], Dinv[6 + c], sf(Ra[r * 3 + 1], Dinv[3 + c], Ra[r * 3] * Dinv[c]));
        if (kDiag) {   // the coefficient term B_a D^-1 b_l = -E_a^T C_a (R_a D^-1 b_l)
            const double *bl = W.b + n6 + 3 * (size_t)l;
            const double b0 = bl[0], b1 = bl[1], b2 = bl[2];
            const EdgeLin La = lrec_form(W.cam, ra);
            const double t[3] = {G0[0] * b0 + G0[1] * b1 + G0[2] * b2, G0[3] * b0 + G0[4] * b1 + G0[5] * b2, G0[6] * b0 + G0[7] * b1 + G0[8] * b2};
            double h[3];
            lrec_C(La, t, h);
            acc[36] += La.b * h[2] - h[1];
            acc[37] += h[0] - La.a * h[2];
            acc[38] += La.a * h[1] - La.b * h[0];
            acc[39] += La.iz * h[0];
            acc[40] += La.iz * h[1];
            acc[41] += La.iz * h[2];
        }
        double Rb[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rb[i] = kDiag ? Ra[i] : Rb_g[i];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) G[r * 3 + c] = sf(G0[r * 3 + 2], Rb[c * 3 + 2], sf(G0[r * 3 + 1], Rb[c * 3 + 1], G0[r * 3] * Rb[c * 3]));
    }
    const EdgeLin La = lrec_form(W.cam, ra), Lb = kDiag ? La : lrec_form(W.cam, rb);
    double Q[9];
    {
        double CG[9];   // C_a G
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            CG[c] = sf(La.C02, G[6 + c], La.C00 * G[c]);
            CG[3 + c] = sf(La.C12, G[6 + c], La.C11 * G[3 + c]);
            CG[6 + c] = sf(La.C22, G[6 + c], sf(La.C12, G[3 + c], La.C02 * G[c]));
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            Q[r * 3] = sf(CG[r * 3 + 2], Lb.C02, CG[r * 3] * Lb.C00);
            Q[r * 3 + 1] = sf(CG[r * 3 + 2], Lb.C12, CG[r * 3 + 1] * Lb.C11);
            Q[r * 3 + 2] = sf(CG[r * 3 + 2], Lb.C22, sf(CG[r * 3 + 1], Lb.C12, CG[r * 3] * Lb.C02));
        }
    }
    if (kDiag) {   // the edge's own Hpp term J_pose^T w J_pose = E^T (w Pt^T Pt) E = E^T (C / iz) E enters with the other sign: the
                   // diagonal block is Hpp - sum B D^-1 B^T = -sum E^T (Q - w Pt^T Pt) E, so the keyframe side of the linearisation
                   // (a second pass over every free-keyframe edge: map, Jacobian, 126 products) is not needed after the first one
        const double w = fabs(ra.w), fx2 = W.cam.fx * W.cam.fx, fy2 = W.cam.fy * W.cam.fy;
        const bool st = ra.w < 0.0;
        const double fs = st ? W.cam.fx : 0.0, c = st ? W.cam.bf * La.iz - La.a * W.cam.fx : 0.0;
        const double P00 = (fx2 + fs * fs) * w, P02 = (fs * c - La.a * fx2) * w, P11 = fy2 * w, P12 = -(La.b * fy2) * w;
        const double P22 = (La.a * La.a * fx2 + La.b * La.b * fy2 + c * c) * w;
        Q[0] -= P00; Q[2] -= P02; Q[4] -= P11; Q[5] -= P12; Q[6] -= P02; Q[7] -= P12; Q[8] -= P22;
    }
    // U = Q E_b (3 x 6), then E_a^T U: rows 0-2 = [a b 1]x^T U, rows 3-5 = -iz_a U
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        double u[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double q0 = Q[r * 3], q1 = Q[r * 3 + 1], q2 = Q[r * 3 + 2];
            u[r] = c == 0 ? q1 - Lb.b * q2 : c == 1 ? Lb.a * q2 - q0 : c == 2 ? Lb.b * q0 - Lb.a * q1 : -(Lb.iz * Q[r * 3 + (c - 3)]);
        }
        // (a diagonal block is symmetric -- E^T (Q - w Pt^T Pt) E with a symmetric middle --: its unit forms the upper triangle only and
        // stores both; the six slots that frees, (1,0) (2,0) (2,1) (3,0) (3,1) (3,2), carry b_p below)
        acc[c] += u[1] - La.b * u[2];
        if (!kDiag || c >= 1) acc[6 + c] += La.a * u[2] - u[0];
        if (!kDiag || c >= 2) acc[12 + c] += La.b * u[0] - La.a * u[1];
        if (!kDiag || c >= 3) acc[18 + c] += -(La.iz * u[0]);
        if (!kDiag || c >= 4) acc[24 + c] += -(La.iz * u[1]);
        if (!kDiag || c >= 5) acc[30 + c] += -(La.iz * u[2]);
    }
    if (kDiag) {   // b_p += J_pose^T omr = E^T (Pt^T omr) (lrec_bp), omr as the landmark side left it
        const double o[3] = {W.lomr[3 * (size_t)kb], W.lomr[3 * (size_t)kb + 1], W.lomr[3 * (size_t)kb + 2]};
        double bp[6] = {0, 0, 0, 0, 0, 0};
        lrec_bp(W.cam, ra, o, bp);
        acc[6] += bp[0]; acc[12] += bp[1]; acc[13] += bp[2]; acc[18] += bp[3]; acc[19] += bp[4]; acc[20] += bp[5];
    }
}

// Hschur = Hpp + lambda I - sum (diagonal), -sum (off-diagonal, both triangles); bschur = b_p - sum
__device__ __forceinline__ void schur_store(const LbaWin &W, int i1, int i2, int e, double sum, double lambda)
{
    const bool diag = i1 == i2;
    if (e < 36) {
        double v = -sum;
        const int r = e / 6, c = e - 6 * r;
        if (diag && r == c) v += lambda;   // (Hpp is inside the diagonal unit's sum: schur_item<true>)
        W.Hs[(size_t)(6 * i1 + r) * W.hs_ld + 6 * i2 + c] = v;
        if (!diag) W.Hs[(size_t)(6 * i2 + c) * W.hs_ld + 6 * i1 + r] = v;
    }
    // (bschur of a diagonal unit: k_schur, after its b_p pass)
}

#ifndef AOS2_SCHUR_WPE
#define AOS2_SCHUR_WPE 3   // workgroups per compute unit (one wave of a workgroup per SIMD)
#endif
__global__ __launch_bounds__(kSchurThreads) __attribute__((amdgpu_waves_per_eu(AOS2_SCHUR_WPE, AOS2_SCHUR_WPE)))
void k_schur(const LbaWin *__restrict__ wins, const SchurTask *__restrict__ tasks)
{
    lba_wave_prio();
    constexpr int NT = kSchurThreads;
    __shared__ double red[(NT / 16) * 43];
    __shared__ int32_t s_info[16], s_ij[16];
    const SchurTask tk = tasks[blockIdx.x];
    if (tk.w < 0) return;
    const LbaWin &W = wins[tk.w];
    // (every dependent load is a round trip of its own on the workgroup's critical path: the state words and the unit's
    // item range are requested together, before the branch on the first of them)
    const int run = W.st->run;
    const double lambda = W.st->lambda;
    const int np = W.np, n6 = 6 * np;
    const int kind = tk.code >> 28, arg = tk.code & 0x0fffffff;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double acc[42];
#pragma unroll
    for (int i = 0; i < 42; ++i) acc[i] = 0;
    if (kind == kSchurPack) {
        const int row = tid >> 4, li = tid & 15, r = arg + row;
        int o0 = 0, info = 0, ij = 0;
        if (r < W.n_srows) {
            o0 = W.sr_o0[r];
            info = W.sr_info[r];
            ij = W.sr_ij[r];
        }
        if (!run) return;
        if (li == 0) {
            s_info[row] = info;
            s_ij[row] = ij;
        }
        if (li < (info & 255)) schur_item<false>(W, o0 + li, lambda, n6, W.Rl + 9 * (size_t)(ij & 0xffff), W.Rl + 9 * (size_t)(ij >> 16), acc);
#pragma unroll
        for (int i = 0; i < 36; ++i) acc[i] = row_sum_f64(acc[i]);
        if (li == 0) {
#pragma unroll
            for (int i = 0; i < 36; ++i) red[row * 43 + i] = acc[i];
        }
        __syncthreads();
        // a block's rows are neighbours inside the unit: its first row's slot adds them in row order
        for (int rr = wave; rr < 16; rr += NT / 64) {
            const int inf = s_info[rr], nrb = (inf >> 16) & 255;
            if (((inf >> 8) & 255) != 0 || nrb == 0 || lane >= 36) continue;
            double sum = 0;
            for (int k = 0; k < nrb; ++k) sum += red[(rr + k) * 43 + lane];
            schur_store(W, s_ij[rr] & 0xffff, s_ij[rr] >> 16, lane, sum, lambda);
        }
        return;
    }
    // DIAG / BIG: one block, items strided over the workgroup
    int i1, i2;
    if (kind == kSchurDiag)
        i1 = i2 = arg;
    else {   // block rank -> (i1 < i2), row-major upper triangle (the order of blk_off)
        int rem = arg;
        i1 = 0;
        while (rem >= np - i1) {
            rem -= np - i1;
            ++i1;
        }
        i2 = i1 + rem;
    }
    const int blk = i1 * np - i1 * (i1 - 1) / 2 + (i2 - i1);
    const int o0 = W.blk_off[blk], n = W.blk_off[blk + 1] - o0;
    if (!run) return;
    if (kind == kSchurDiag)
        for (int j = tid; j < n; j += NT) schur_item<true>(W, o0 + j, lambda, n6, W.Rl + 9 * (size_t)i1, W.Rl + 9 * (size_t)i1, acc);
    else
        for (int j = tid; j < n; j += NT) schur_item<false>(W, o0 + j, lambda, n6, W.Rl + 9 * (size_t)i1, W.Rl + 9 * (size_t)i2, acc);
    const double sum = workgroup_sum_k256<42, NT>(acc, red, n);
    if (kind != kSchurDiag) {
        schur_store(W, i1, i2, tid, sum, lambda);
        return;
    }
    // diagonal unit: Hschur_ii = lambda I - (upper-triangle sums, mirrored); b_p from its six slots; bschur = b_p - sum B D^-1 b_l
    __shared__ double s_bsum[6];
    if (tid >= 36 && tid < 42) s_bsum[tid - 36] = sum;
    __syncthreads();
    if (tid < 36) {
        const int r = tid / 6, c = tid - 6 * r;
        if (r <= c) {
            double v = -sum;
            if (r == c) v += lambda;
            W.Hs[(size_t)(6 * i1 + r) * W.hs_ld + 6 * i1 + c] = v;
            if (r != c) W.Hs[(size_t)(6 * i1 + c) * W.hs_ld + 6 * i1 + r] = v;
        } else {
            const int j = tid == 6 ? 0 : tid == 12 ? 1 : tid == 13 ? 2 : tid == 18 ? 3 : tid == 19 ? 4 : tid == 20 ? 5 : -1;
            if (j >= 0) {
                W.b[6 * i1 + j] = sum;
                W.bs[6 * i1 + j] = sum - s_bsum[j];
            }
        }
    }
}

// Dense LDL^T (no pivoting; fails on a zero pivot like Eigen::SimplicialLDLT) + solve of the reduced
// camera system, one workgroup per window.  Blocked right-looking factorisation, block 16:
//   (1) 16x16 diagonal block: unblocked LDL^T by wave 0 (wave-synchronous, no workgroup barrier)
//   (2) panel: one thread per row below the block, 16-column forward substitution, W = L * D kept
//   (3) trailing update A[I][J] -= W[I] * L[J]^T on 16x16 tiles with v_mfma_f64_16x16x4_f64
//       (the only GEMM-shaped piece of the path; tiles round-robin over the waves)
// then forward / diagonal / backward substitution.
// The matrix is padded to a multiple of 16 with an identity tail.  kLds: it lives in LDS (npad <= 128, i.e. <= 21 free
// keyframes; 4 waves; substitution by wave 0 with the vector in registers, which then applies the update to the
// poses).  Otherwise (any size): in a global scratch, 16 waves, the vector in LDS, k_update_poses follows.
typedef double double4_t __attribute__((ext_vector_type(4)));

// ---- LDS variant (npad <= 128): 8 waves, per 16-column panel k
//   D_k  the 16x16 diagonal block by wave 0: unblocked LDL^T with row i of the block in the registers of lane i (pivot
//        and column entries travel through v_readlane; no predication: the upper halves of the rows are scratch), the
//        reciprocal pivots, then T_k = L_kk^-1 (lane c owns column c), stored in the unused UPPER triangle of the
//        block, and the block's part of the forward substitution y_k = T_k r_k
//   P_k  panel below the block as a GEMM: W_I = A_Ik T_k^T on 16x16 tiles with v_mfma_f64_16x16x4_f64, L_Ik = W_I D^-1;
//        the rows' share of the forward substitution r_i -= L_ik y_k (so L y = b is solved while the matrix is
//        factorised)
//   U_k  trailing update A[I][J] -= W[I] L[J]^T on 16x16 tiles (MFMA); wave 0 takes the tile (k+1, k+1) and goes
//        straight on to D_{k+1} (look-ahead) while the other seven waves update the rest
// Two workgroup barriers per panel.  The backward substitution x = L^-T D^-1 y walks the panels the other way round
// with the same look-ahead (x_k = T_k^T s_k, one barrier per panel); finally the first np threads apply the update to
// the poses.  (The former version -- 4 waves, panel by per-row substitution with 16 divisions per row, predicated
// pivot updates, separate forward / backward passes by one wave -- took 76 us at 20 free keyframes.)
// cycle counters of the phases (s_memtime) for tools: compile with -DAOS2_LDLT_TIMING; off by default (every timestamp waits
// for the wave's outstanding LDS traffic)
#ifdef AOS2_LDLT_TIMING
#define LDLT_T(...) __VA_ARGS__
#else
#define LDLT_T(...)
#endif
// kGlob: the same algorithm for a reduced system beyond LDS (more than 21 free keyframes: npad > 128) -- the matrix lives in the
// device memory (in place in Wn.Hs, which k_schur wrote with leading dimension npad: 240 x 240 doubles at 40 free keyframes,
// L2-resident), everything else (the panel's L D,
// the vectors, T) stays in LDS.  What one wave stores and another wave of the workgroup loads afterwards is ordered by
// __syncthreads() (the compute unit's vector cache is write-through and shared by the workgroup's waves); inside wave 0 by
// wave_sync() = completion of the wave's outstanding stores.  The pivot chain of the diagonal blocks never touches memory, so
// what the variant pays is one device-memory round trip per phase instead of an LDS one (measured: DESIGN.md 5.3).
typedef double __attribute__((address_space(1))) gdouble_t;
#ifndef AOS2_LDLT_TILE_BATCH
#define AOS2_LDLT_TILE_BATCH 4
#endif
constexpr int kTileBatch = AOS2_LDLT_TILE_BATCH;
template <bool kGlob>
__device__ __forceinline__ void ldlt_body(const LbaWin &Wn, double *sm)
{
    constexpr int NT = 512, NW = 8;
    __shared__ int s_fail, s_tile;
    const int n = 6 * Wn.np, npad = Wn.npad;
    const double lambda = Wn.st->lambda;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // odd leading dimensions: column-direction accesses (MFMA operands, substitution) fall on distinct banks
    const int ld = kGlob ? npad : npad + 1, lw = 17;   // (device scratch: rows on 128-byte boundaries)
    using mptr_t = std::conditional_t<kGlob, gdouble_t *, double *>;
    // npad x ld: lower triangle = the matrix, then L.  kGlob: k_schur wrote the matrix straight into this layout (Wn.hs_ld = npad,
    // identity tail set once by k_prepare; the tail stays an identity under the factorisation) -- factorised in place
    mptr_t M = kGlob ? (mptr_t)Wn.Hs : (mptr_t)sm;
    double *W = kGlob ? sm : sm + (size_t)npad * ld;   // npad x lw: L * D of the current panel
    double *dvec = W + (size_t)npad * lw;        // npad: D
    double *rdv = dvec + npad;                   // npad: 1 / D
    double *rv = rdv + npad;                     // npad: residual of the forward substitution, then y, then D^-1 y, then s
    double *xs = rv + npad;                      // npad: solution
    double *Tb = xs + npad;                      // 2 x 16 x 17: T_k^T of the current / next panel
    double *Dst = Tb + 2 * 16 * 17;              // kGlob: 16 x 17, the next diagonal block on its way from the trailing update to wave 0's rows
    LDLT_T(long long tD = 0, tP = 0, tU = 0, t_a = 0; const long long t_begin = __builtin_amdgcn_s_memtime();)
    if (tid == 0) s_fail = 0;
    // load (identity-padded), lower BLOCK triangle only -- row r needs its columns up to the end of its diagonal block,
    // nothing reads the blocks above the diagonal -- as element pairs (n is even, rows are 16-byte aligned): block row R
    // holds 16 x 8 (R + 1) pairs, 64 R (R + 1) pairs lie before it.  9 independent 16-byte global loads in flight per
    // thread before the stores: a 128 x 128 system is one round.
    if (!kGlob) {
        const int nbr = npad >> 4, npairs = 64 * nbr * (nbr + 1);
        auto where = [&](int q, int &r, int &c) {
            int R = (int)((__fsqrt_rn(1.0f + (float)q * 0.0625f) - 1.0f) * 0.5f);
            while (64 * (R + 1) * (R + 2) <= q) ++R;   // (the float estimate can be one short)
            while (64 * R * (R + 1) > q) --R;
            const int q1 = q - 64 * R * (R + 1), w = 8 * (R + 1);
            const int rr = q1 / w;
            r = 16 * R + rr;
            c = (q1 - rr * w) << 1;
        };
        for (int p0 = tid; p0 < npairs; p0 += 9 * NT) {
            double2 v[9];
            int rr[9], cc[9];
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                const int q = p0 + u * NT;
                rr[u] = cc[u] = 0;
                if (q < npairs) where(q, rr[u], cc[u]);
                const int r = rr[u], c = cc[u];
                if (q < npairs && r < n && c < n)
                    v[u] = *reinterpret_cast<const double2 *>(Wn.Hs + (size_t)r * n + c);
                else
                    v[u] = double2{r == c ? 1.0 : 0.0, r == c + 1 ? 1.0 : 0.0};
            }
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                const int q = p0 + u * NT;
                if (q < npairs) {
                    M[(size_t)rr[u] * ld + cc[u]] = v[u].x;
                    M[(size_t)rr[u] * ld + cc[u] + 1] = v[u].y;
                }
            }
        }
    }
    for (int i = tid; i < npad; i += NT) rv[i] = i < n ? Wn.bs[i] : 0.0;
    __syncthreads();
    LDLT_T(const long long t_loaded = __builtin_amdgcn_s_memtime();)
    auto wave_sync = [] {   // LDS writes of this wave visible to its other lanes (one wave: LDS operations execute in order)
        if (kGlob) {   // ... and its stores to the device scratch complete before its other lanes load them
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
    // D_k (wave 0 only; lanes >= 16 mirror lanes 0..15, their results are not stored).  Measured single-wave latencies
    // (tools/microbench/f64_latency.hip): dependent f64 FMA 9 cycles, IEEE division 67, v_rcp_f64 + 2 Newton steps 34,
    // a v_readlane pair 13, LDS write -> broadcast read 77.  So per pivot: the pivot itself comes by v_readlane and its
    // reciprocal by rcp + Newton (within 1 ulp), the column's other entries by ONE LDS write + broadcast reads that fly
    // while the reciprocal is refined (30 v_readlane would cost 200 cycles of issue), and nothing is predicated: the
    // upper halves of the rows are scratch.  T = L_kk^-1 is built on the way -- lane c owns column c,
    // T <- (I - l_j e_j^T) T needs only column j of L, which every lane has just read -- and so is y_k = L_kk^-1 r_k.
    LDLT_T(long long dt1 = 0, dt2 = 0, dt3 = 0, dt4 = 0;)
    auto rcp_newton = [](double d) {   // 1 / d within 1 ulp: v_rcp_f64 + two Newton steps (34 cycles; the IEEE division takes 67)
        double rd = __builtin_amdgcn_rcp(d);
        double e = __builtin_fma(-d, rd, 1.0);
        rd = __builtin_fma(rd, e, rd);
        e = __builtin_fma(-d, rd, 1.0);
        return __builtin_fma(rd, e, rd);
    };
    auto diag_block = [&](int k0, bool staged) {
        LDLT_T(const long long q0 = __builtin_amdgcn_s_memtime();)
        const int li = lane & 15;
        double *colbuf = xs;   // 16 doubles of scratch (xs is unused until the backward pass)
        double *Tk = Tb + ((k0 >> 4) & 1) * (16 * 17);
        double row[16];
        if (kGlob && staged) {
#pragma unroll
            for (int c = 0; c < 16; ++c) row[c] = Dst[li * 17 + c];
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) row[c] = M[(size_t)(k0 + li) * ld + k0 + c];
        }
        double dsave = 1.0;
        double cur = rv[k0 + li];
        bool bad = false;
        // T = L_kk^-1 (unit lower triangular), lane c owns column c, built alongside the factorisation: eliminating column j
        // gives every lane the whole column (ck[k] = a_kj, the broadcast it needs for its own row anyway) and the reciprocal
        // pivot, i.e. L[k][j] = ck[k] rd for all k -- so t[k] -= L[k][j] t[j] costs one product and 15 - j fused
        // multiply-adds here, instead of a second pass that re-read L through 120 LDS broadcasts per lane (1840 cycles per
        // block).  (Rows in lanes 0..15 and T in lanes 16..31 as ONE instruction stream -- both updates are
        // x[k] -= (x[j] rd) ck[k] -- was measured slower: 36.5 k against 34.7 k cycles for the 8 blocks, the lane-group
        // selects and predicated stores cost more than the 7.5 multiply-adds per pivot they save.)
        double t[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) t[c] = c == li ? 1.0 : 0.0;
        __builtin_amdgcn_sched_barrier(0);
        LDLT_T(const long long q1 = __builtin_amdgcn_s_memtime(); dt1 += q1 - q0;)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double ci = row[j];
            if (kGlob) dsave = li == j ? ci : dsave;   // the pivot of this lane's row (what is stored at M[k0 + li][k0 + li] below)
            if (j < 15) colbuf[li] = ci;
            const double dj = readlane_f64(ci, j);
            if (dj == 0.0 || dj != dj) bad = true;
            double ck[16];
#pragma unroll
            for (int k = j + 1; k < 16; ++k) ck[k] = colbuf[k];
            const double rd = rcp_newton(dj);
            const double lij = ci * rd;
            const double yj = readlane_f64(cur, j);
            const double st = rd * t[j];
#pragma unroll
            for (int k = j + 1; k < 16; ++k) {
                row[k] = __builtin_fma(-lij, ck[k], row[k]);
                t[k] = __builtin_fma(-ck[k], st, t[k]);
            }
            // row i > j: its entry of column j becomes L[i][j]; row j keeps the pivot; rows above hold scratch there.  Both
            // finished values leave for LDS at once (the rows of the block without predication: the entries right of the
            // diagonal are scratch that nothing reads; lanes >= 16 store the same values to the same places; column c of T
            // = row c of the panel's T^T buffer, dense: zeros above the diagonal, ones on it), so the live registers shrink
            // with j instead of holding 16 finished entries of each array to the end
            M[(size_t)(k0 + li) * ld + k0 + j] = li == j ? ci : lij;   // (lanes >= 16 repeat the stores of lanes 0..15: a predicate here costs 85 registers and spills)
            Tk[li * 17 + j] = t[j];
            cur = li > j ? __builtin_fma(-lij, yj, cur) : cur;   // forward substitution inside the block
            __builtin_amdgcn_sched_barrier(0);   // keep the pivots apart (hoisting the later pivots' reads only costs spills)
        }
        LDLT_T(const long long q2 = __builtin_amdgcn_s_memtime(); dt2 += q2 - q1;)
        if (bad) {
            if (lane == 0) s_fail = 1;
            return;
        }
        rv[k0 + li] = cur;   // y of this block
        wave_sync();
        {
            const double d = kGlob ? dsave : (double)M[(size_t)(k0 + li) * ld + k0 + li];
            dvec[k0 + li] = d;
            rdv[k0 + li] = rcp_newton(d);   // (the same operations as in the loop: the same bits)
        }
        LDLT_T(const long long q3 = __builtin_amdgcn_s_memtime(); dt3 += q3 - q2;)
        LDLT_T(dt4 += __builtin_amdgcn_s_memtime() - q3;)
    };
    if (wave == 0) diag_block(0, false);
    __syncthreads();
    LDLT_T(const long long t_d0 = __builtin_amdgcn_s_memtime();)
    const int nb = npad >> 4;
    const int col = lane & 15, rq = lane >> 4;
    for (int kb = 0; kb < nb && !s_fail; ++kb) {
        const int k0 = kb << 4;
        const int m = nb - kb - 1;
        LDLT_T(t_a = __builtin_amdgcn_s_memtime();)
        // ---- P_k: W_I = A_Ik T^T (one 16-row tile per wave turn), L_Ik = W_I D^-1
        // (kGlob: wave 0 requests the entries of the NEXT diagonal block -- tile 0 of the trailing update, final since the last
        // panel's update -- before the panel: its round trip to device memory is off the critical path of the look-ahead)
        double4_t acc0 = {0, 0, 0, 0};
        if (kGlob && wave == 0 && m > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc0[r] = M[(size_t)(k0 + 16 + rq + 4 * r) * ld + k0 + 16 + col];
        }
        for (int ti = wave; ti < m; ti += NW) {
            const int I0 = (kb + 1 + ti) << 4;
            double4_t acc = {0, 0, 0, 0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int c = 4 * kk + rq;
                const double av = M[(size_t)(I0 + col) * ld + k0 + c];                                  // A[i = col][c]
                const double tv = Tb[(kb & 1) * (16 * 17) + c * 17 + col];                                // B[c][j = col] = T[j][c]
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, tv, acc, 0, 0, 0);
            }
            const double rd = rdv[k0 + col];
            // the tile's A entries were all read by the MFMAs above (this wave only): overwrite them with L
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                W[(size_t)(I0 + rq + 4 * r) * lw + col] = acc[r];
                M[(size_t)(I0 + rq + 4 * r) * ld + k0 + col] = acc[r] * rd;
            }
        }
        if (kGlob && tid == 0) s_tile = 1;   // (tile 0 is wave 0's)
        __syncthreads();
        // forward substitution of the rows below: r_i -= sum_c L[i][k0 + c] y_c (c ascending)
        for (int i = k0 + 16 + tid; i < npad; i += NT) {
            double l[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) l[c] = kGlob ? W[(size_t)i * lw + c] * rdv[k0 + c] : (double)M[(size_t)i * ld + k0 + c];   // (L = W D^-1: the product the panel stored, the same bits)
            double ri = rv[i];
#pragma unroll
            for (int c = 0; c < 16; ++c) ri -= l[c] * rv[k0 + c];
            rv[i] = ri;
        }
        LDLT_T(tP += __builtin_amdgcn_s_memtime() - t_a; t_a = __builtin_amdgcn_s_memtime();)
        // ---- U_k: trailing update with f64 MFMA on the lower-triangle tiles (I >= J > kb); tile 0 = (kb+1, kb+1) goes
        // to wave 0, which then factorises that block while the other waves finish the update
        const int ntiles = m * (m + 1) / 2;
        auto tile_at = [&](int t, int &I0, int &J0) {
            int ii = 0, rem = t;
            while (rem > ii) {  // row-major lower triangle: row ii holds ii+1 tiles
                rem -= ii + 1;
                ++ii;
            }
            I0 = (kb + 1 + ii) << 4;
            J0 = (kb + 1 + rem) << 4;
        };
        // B[k][j] = L[J0 + j][k0 + k]: from the matrix, or (kGlob) as the product W D^-1 the panel stored there -- the same bits, from LDS
        auto lval = [&](int J0, int kk) -> double {
            return kGlob ? W[(size_t)(J0 + col) * lw + 4 * kk + rq] * rdv[k0 + 4 * kk + rq] : (double)M[(size_t)(J0 + col) * ld + k0 + 4 * kk + rq];
        };
        auto tile = [&](int t) {
            int I0, J0;
            tile_at(t, I0, J0);
            double4_t acc;
            if (kGlob && t == 0)
                acc = acc0;
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = M[(size_t)(I0 + rq + 4 * r) * ld + J0 + col];
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const double av = -W[(size_t)(I0 + col) * lw + 4 * kk + rq];          // A[i = lane&15][k = lane>>4]
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, lval(J0, kk), acc, 0, 0, 0);
            }
            if (kGlob && t == 0) {   // the next diagonal block: to wave 0's rows through LDS, not through the device scratch
#pragma unroll
                for (int r = 0; r < 4; ++r) Dst[(rq + 4 * r) * 17 + col] = acc[r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) M[(size_t)(I0 + rq + 4 * r) * ld + J0 + col] = acc[r];
            }
        };
        // kGlob: the tiles are handed out kTileBatch at a time from a counter in LDS (a tile's result does not depend on the wave
        // that forms it), the next batch's entries are requested before the current batch's products are formed (a batch is a
        // round trip to device memory), and wave 0 joins when its diagonal block is done: at 30-40 free keyframes the trailing
        // update, not the pivot chain, was the longer side of the look-ahead (measured: DESIGN.md 5.3)
        auto run_tiles = [&]() {
            auto grab = [&]() {
                int t = 0;
                if (lane == 0) t = atomicAdd(&s_tile, kTileBatch);
                return __builtin_amdgcn_readfirstlane(t);
            };
            int I0[kTileBatch], J0[kTileBatch];
            double4_t acc[kTileBatch];
            auto fetch = [&](int t, int (&I)[kTileBatch], int (&J)[kTileBatch], double4_t (&a)[kTileBatch]) {
#pragma unroll
                for (int u = 0; u < kTileBatch; ++u) tile_at(min(t + u, ntiles - 1), I[u], J[u]);
#pragma unroll
                for (int u = 0; u < kTileBatch; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[u][r] = M[(size_t)(I[u] + rq + 4 * r) * ld + J[u] + col];
            };
            int t = grab();
            if (t < ntiles) fetch(t, I0, J0, acc);
            while (t < ntiles) {
                const int tn = grab();
                int In[kTileBatch], Jn[kTileBatch];
                double4_t an[kTileBatch];
                if (tn < ntiles) fetch(tn, In, Jn, an);
#pragma unroll
                for (int u = 0; u < kTileBatch; ++u)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const double av = -W[(size_t)(I0[u] + col) * lw + 4 * kk + rq];
                        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, lval(J0[u], kk), acc[u], 0, 0, 0);
                    }
#pragma unroll
                for (int u = 0; u < kTileBatch; ++u)
                    if (t + u < ntiles)
#pragma unroll
                        for (int r = 0; r < 4; ++r) M[(size_t)(I0[u] + rq + 4 * r) * ld + J0[u] + col] = acc[u][r];
                t = tn;
#pragma unroll
                for (int u = 0; u < kTileBatch; ++u) {
                    I0[u] = In[u]; J0[u] = Jn[u]; acc[u] = an[u];
                }
            }
        };
        if (wave == 0) {
            if (ntiles > 0) {
                // (the rows k0+16 .. k0+31 of the forward substitution above belong to threads 0..15 = this wave)
                wave_sync();
                tile(0);
                wave_sync();
                LDLT_T(const long long t_d = __builtin_amdgcn_s_memtime();)
                diag_block(k0 + 16, true);
                LDLT_T(tD += __builtin_amdgcn_s_memtime() - t_d;)
                if (kGlob) run_tiles();
            }
        } else if (kGlob) {
            run_tiles();
        } else {
            for (int t = wave; t < ntiles; t += NW - 1) tile(t);
        }
        __syncthreads();
        LDLT_T(tU += __builtin_amdgcn_s_memtime() - t_a;)
    }
    LDLT_T(const long long t_fact = __builtin_amdgcn_s_memtime();)
    if (s_fail) {
        if (tid == 0) Wn.scal[3] = 0.0;
        if (tid < Wn.np) {   // the trial is rejected (lm_decide): its pop() must find the estimates it started from
            const double *Tp = Wn.pose + 7 * (size_t)Wn.hpose[tid];
            double *Tbk = Wn.bk + 7 * (size_t)Wn.hpose[tid];
            for (int i = 0; i < 7; ++i) Tbk[i] = Tp[i];
        }
        return;
    }
    // ---- backward substitution: x = L^-T D^-1 y; per block x_k = T_k^T s_k, then the rows above take L_k^T x_k
    for (int i = tid; i < npad; i += NT) rv[i] = rv[i] * rdv[i];
    __syncthreads();
    auto back_load = [&](int k0, double (&lcol)[16]) {   // L[j][i] for j > i of the diagonal block (the rest is unused)
        const int li = lane & 15;
#pragma unroll
        for (int j = 0; j < 16; ++j) lcol[j] = M[(size_t)(k0 + j) * ld + k0 + li];
    };
    auto back_block = [&](int k0, const double (&lcol)[16]) {   // wave 0: L_kk^T x_k = s_k, k descending inside the block
        const int li = lane & 15;
        double cur = rv[k0 + li];
#pragma unroll
        for (int j = 15; j >= 1; --j) {
            const double xj = readlane_f64(cur, j);
            cur = li < j ? __builtin_fma(-lcol[j], xj, cur) : cur;
        }
        xs[k0 + li] = cur;
    };
    if (wave == 0) {
        double lc[16];
        back_load((nb - 1) << 4, lc);
        back_block((nb - 1) << 4, lc);
    }
    __syncthreads();
    for (int kb = nb - 1; kb > 0; --kb) {
        const int k0 = kb << 4;
        // rows above the block take its contribution; wave 0 does the rows of the next block first and solves it
        auto update_row = [&](int i) {
            double l[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) l[j] = M[(size_t)(k0 + j) * ld + i];
            double acc = rv[i];
#pragma unroll
            for (int j = 15; j >= 0; --j) acc -= l[j] * xs[k0 + j];
            rv[i] = acc;
        };
        if (wave == 0) {
            double lc[16];
            back_load(k0 - 16, lc);   // (independent of x: requested together with the rows' entries -- one round trip, not two, when the matrix is in device memory)
            if (lane < 16) update_row(k0 - 16 + lane);
            wave_sync();
            back_block(k0 - 16, lc);
        } else {
            for (int i = tid - 64; i < k0 - 16; i += NT - 64) update_row(i);
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += NT) Wn.x[i] = xs[i];
    if (tid == 0) {
        Wn.scal[3] = 1.0;
        LDLT_T(long long *g = Wn.st->dbg; g[0] = t_loaded - t_begin; g[1] = t_d0 - t_loaded; g[2] = tP; g[3] = tU; g[4] = tD;
               g[5] = t_fact - t_d0; g[6] = __builtin_amdgcn_s_memtime() - t_fact; g[8] = dt1; g[9] = dt2; g[10] = dt3; g[11] = dt4;)
    }
    if (tid < Wn.np) {   // VertexSE3Expmap::oplusImpl + the poses' scale terms
        double upd[6];
        for (int i = 0; i < 6; ++i) {
            upd[i] = xs[6 * tid + i];
            Wn.tmp[6 * tid + i] = upd[i] * (lambda * upd[i] + Wn.b[6 * tid + i]);
        }
        double *Tp = Wn.pose + 7 * (size_t)Wn.hpose[tid], *Tbk = Wn.bk + 7 * (size_t)Wn.hpose[tid];
        for (int i = 0; i < 7; ++i) Tbk[i] = Tp[i];   // push()
        se3_oplus_fast(upd, Tp);   // (polynomial small-angle form, lba_math.h; agrees with se3_oplus to a few ulp)
    }
}

__global__ __launch_bounds__(512) void k_ldlt_lds(const LbaWin *__restrict__ wins)
{
    lba_wave_prio();
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const LbaWin &Wn = wins[blockIdx.x];
    if (!Wn.st->run || Wn.np == 0 || Wn.ldlt_lds != 1) return;
    ldlt_body<false>(Wn, sm);
}
// (a kernel of its own: both forms in one kernel cost the LDS form 25 spilled registers)
__global__ __launch_bounds__(512) void k_ldlt_dev(const LbaWin *__restrict__ wins)
{
    lba_wave_prio();
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const LbaWin &Wn = wins[blockIdx.x];
    if (!Wn.st->run || Wn.np == 0 || Wn.ldlt_lds != 0) return;
    ldlt_body<true>(Wn, sm);
}

// The form every window of <= 40 free keyframes takes (ldlt_reg.h): the trailing matrix as MFMA accumulator tiles in the registers of
// seven waves, the diagonal tiles / panel / T_k / vectors in LDS, no device-memory traffic between the load and the solution; the
// epilogue is ldlt_body's (solution, pose update with the push() backup, the poses' scale terms).
__global__ __launch_bounds__(kLrThreads) void k_ldlt_reg(const LbaWin *__restrict__ wins)
{
    lba_wave_prio();
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const LbaWin &Wn = wins[blockIdx.x];
    if (!Wn.st->run || Wn.np == 0 || Wn.ldlt_lds != 2) return;
    const int tid = threadIdx.x, n = 6 * Wn.np;
    const double lambda = Wn.st->lambda;
    double *xs = nullptr;
#ifdef AOS2_LDLT_TIMING
    const bool ok = ldlt_reg_solve<true>(Wn.Hs, n, Wn.npad, Wn.bs, sm, xs, Wn.st->dbg);
#else
    const bool ok = ldlt_reg_solve<false>(Wn.Hs, n, Wn.npad, Wn.bs, sm, xs, nullptr);
#endif
    if (!ok) {
        if (tid == 0) Wn.scal[3] = 0.0;
        if (tid < Wn.np) {   // the trial is rejected (lm_decide): its pop() must find the estimates it started from
            const double *Tp = Wn.pose + 7 * (size_t)Wn.hpose[tid];
            double *Tbk = Wn.bk + 7 * (size_t)Wn.hpose[tid];
            for (int i = 0; i < 7; ++i) Tbk[i] = Tp[i];
        }
        return;
    }
    for (int i = tid; i < n; i += kLrThreads) Wn.x[i] = xs[i];
    if (tid == 0) Wn.scal[3] = 1.0;
    if (tid < Wn.np) {   // VertexSE3Expmap::oplusImpl + the poses' scale terms
        double upd[6];
        for (int i = 0; i < 6; ++i) {
            upd[i] = xs[6 * tid + i];
            Wn.tmp[6 * tid + i] = upd[i] * (lambda * upd[i] + Wn.b[6 * tid + i]);
        }
        double *Tp = Wn.pose + 7 * (size_t)Wn.hpose[tid], *Tbk = Wn.bk + 7 * (size_t)Wn.hpose[tid];
        for (int i = 0; i < 7; ++i) Tbk[i] = Tp[i];   // push()
        se3_oplus_fast(upd, Tp);
    }
}

// Between the two optimisations (Optimizer.cc:667-710), one launch (the bDoMore check of :663-666 was made where the first
// optimisation ended: lm_first_done).  Edges with chi2 above the threshold or non-positive depth leave the optimisation
// (setLevel(1)), all edges drop their robust kernel (:672-703); the workgroup that finishes last does
// initializeOptimization(0) (fails without level-0 edges) and the entry of optimize(10).
// _error of edge e as the last residual evaluation left it (LmState::err_at): re-formed with the operations of the residual pass
// from the estimates it ran on (p_cur = the edge's camera point at the CURRENT estimates, which the caller needs anyway), and stored
// when `keep` -- the values an edge keeps while it is inactive
__device__ __forceinline__ void edge_last_error(const LbaWin &W, int e, int at, const double p_cur[3], int stereo, bool keep, double er[3])
{
    double *dst = W.err + 3 * (size_t)e;
    if (at == 0) {
        er[0] = dst[0]; er[1] = dst[1]; er[2] = dst[2];
        return;
    }
    double p[3] = {p_cur[0], p_cur[1], p_cur[2]};
    if (at == 2) se3_map(W.bk + 7 * (size_t)W.e_pose[e], W.bk + 7 * (size_t)W.n_poses + 3 * (size_t)W.e_point[e], p);
    const double ob[3] = {(double)W.in_obs[3 * (size_t)e], (double)W.in_obs[3 * (size_t)e + 1], (double)W.in_obs[3 * (size_t)e + 2]};
    edge_error(W.cam, p, ob, stereo, er);
    if (keep) {
        dst[0] = er[0]; dst[1] = er[1]; dst[2] = er[2];
    }
}

__global__ __launch_bounds__(256) void k_transition(const LbaWin *__restrict__ wins)
{
    lba_wave_prio();
    __shared__ int s_last;
    const LbaWin &W = wins[blockIdx.y];
    LmState *st = W.st;
    if (!st->xmark) return;
    const int err_at = st->err_at;   // (reset by the workgroup that finishes last: after every workgroup's read)
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    int keep = 0;
    if (e < W.n_edges) {
        const int stereo = W.e_stereo[e];
        double p[3], er[3];
        se3_map(W.pose + 7 * (size_t)W.e_pose[e], W.point + 3 * (size_t)W.e_point[e], p);
        edge_last_error(W, e, err_at, p, stereo, true, er);
        const double c = edge_chi2(er, (double)W.in_w[e], stereo ? 3 : 2);
        const bool bad = c > (stereo ? 7.815 : 5.991) || !(p[2] > 0.0);
        if (bad) {
            W.e_level1[e] = 1;
            const int pp = W.pl_pos[e];
            if (pp >= 0) lrec_mask(W, pp);
        }
        W.e_robust[e] = 0;
        keep = bad ? 0 : 1;
    }
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&st->n_active, __popcll(m));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the waves' additions are performed before the workgroup reports itself done: see points_tail)
    __syncthreads();
    // (the last workgroup reads nothing of the others but n_active, an atomic: no device-scope fence)
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&st->blocks_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (!s_last || threadIdx.x != 0) return;
    st->blocks_done = 0;
    st->xmark = 0;
    st->it = 0;
    st->err_at = 0;   // (the stored values stand until the second optimisation evaluates)
    const int n_active = __hip_atomic_load(&st->n_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n_active == 0 || st->iters_max[1] <= 0 || lm_poll(st, W.abort_word)) {
        st->phase = 3;
        st->iters_done[1] = 0;
    } else {
        st->phase = 2;
        st->initp = 1;
    }
}

// final inlier check (:712-744) and the write-back conversions (:763-778)
__global__ __launch_bounds__(256) void k_final(const LbaWin *__restrict__ wins)
{
    lba_wave_prio();
    const LbaWin &W = wins[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W.n_edges) {
        const int stereo = W.e_stereo[i];
        double p[3], er[3];
        se3_map(W.pose + 7 * (size_t)W.e_pose[i], W.point + 3 * (size_t)W.e_point[i], p);
        edge_last_error(W, i, W.e_level1[i] ? 0 : W.st->err_at, p, stereo, false, er);   // (an inactive edge kept its _error)
        const double c = edge_chi2(er, (double)W.in_w[i], stereo ? 3 : 2);
        const bool bad = c > (stereo ? 7.815 : 5.991) || !(p[2] > 0.0);
        if (W.out_chi2) W.out_chi2[i] = c;
        W.out_outlier[i] = bad ? 1 : 0;
    }
    if (i < W.n_poses) pose_to_Tcw(W.pose + 7 * (size_t)i, W.out_Tcw + 16 * (size_t)i);
    if (i < W.n_points)
        for (int d = 0; d < 3; ++d) W.out_xyz[3 * (size_t)i + d] = (float)W.point[3 * (size_t)i + d];
}

// ---------------------------------------------------------------------------------------------------------- host
int lba_handle_init(aos2_lba *s)
{
    int st = bind_device(s->device);
    if (st) return st;
    if (s->dev_ready) return AOS2_OK;
    // The optimiser's kernels are few workgroups on a latency-bound chain; the tracking kernels that share the device in a
    // running system (extraction, searches) are wide and throughput-bound.  On a high-priority stream the optimiser's
    // workgroups are dispatched ahead of the waiting ones of those kernels (AOS2_LBA_STREAM_PRIORITY=normal switches it off).
    {
        const char *e = getenv("AOS2_LBA_STREAM_PRIORITY");
        if ((st = stream_create(&s->stream, !e || strcmp(e, "normal")))) return st;
    }
    for (auto &e : s->ev) AOS2_HIP_CHECK(hipEventCreate(&e));
    {
        if ((st = stream_create(&s->stream2, true))) return st;
        AOS2_HIP_CHECK(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
        AOS2_HIP_CHECK(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
    }
    // the reduced-system factorisation keeps up to 128x128 doubles + panel in LDS (<= 150 KB)
    AOS2_HIP_CHECK(hipFuncSetAttribute((const void *)k_ldlt_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
    AOS2_HIP_CHECK(hipFuncSetAttribute((const void *)k_ldlt_dev, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    AOS2_HIP_CHECK(hipFuncSetAttribute((const void *)k_ldlt_reg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLrMaxDynLds));
    s->dev_ready = true;
    return AOS2_OK;
}

// The second window group's streams and events, created when a call first runs two groups: a handle that never does keeps two
// streams.  (The runtime maps the streams of a priority class onto GPU_MAX_HW_QUEUES = 4 hardware queues by creation order and
// serialises those that share one: two handles solving side by side, each with four streams, had their main streams on the same
// queue in some processes and not in others -- 50 k against 60 k frames/s of the composite from run to run.)
static int lba_group_streams(aos2_lba *s)
{
    if (s->stream_b) return AOS2_OK;
    const char *e = getenv("AOS2_LBA_STREAM_PRIORITY");
    const bool prio = !e || strcmp(e, "normal");
    if (int st_ = stream_create(&s->stream_b, prio)) return st_;
    if (int st_ = stream_create(&s->stream2_b, prio)) return st_;
    for (hipEvent_t *ev : {&s->ev_fork_b, &s->ev_join_b, &s->ev_up, &s->ev_stag, &s->ev_done_b}) AOS2_HIP_CHECK(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    return AOS2_OK;
}

static bool stop_requested(const aos2_lba_problem_t *p) { return p->stop_flag && *p->stop_flag != 0; }

// index mapping and edge lists of a window: initializeOptimization(level 0) + buildIndexMapping + the symbolic part of
// buildStructure for ALL edges (the second optimisation masks edges instead of rebuilding)
struct Pass {
    std::vector<int32_t> k_ph, k_lh, hpose, hpoint, pt_off, pt_k, ps_off, ps_k, pl_off, pl_k;
    std::vector<int32_t> pl_pos, pl_ph;   // edge -> position in pl_k (-1: fixed keyframe); position -> pose hidx
    std::vector<int32_t> it_ka, it_kb, it_l, blk_off;   // Schur items (k_schur_items / k_schur_blocks)
    std::vector<int32_t> sr_o0, sr_info, sr_ij, units;    // k_schur's row table and unit codes (build_schur_units)
    int np = 0, nl = 0;
};
// worker threads of a handle for the per-window host work of a batch (structure build, staging): created once -- 32
// std::thread per phase cost ~1 ms per call
struct WindowPool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    std::function<void(int)> fn;
    int n_items = 0, generation = 0, running = 0;
    std::atomic<int> next{0};
    bool quit = false;
    void start(int n)
    {
        int born;
        {
            std::lock_guard<std::mutex> lk(m);
            born = generation;   // a thread added later starts with the jobs after its creation, not with a finished one
        }
        active = n;
        for (int t = (int)th.size(); t < n; ++t)
            th.emplace_back([this, born, idx = t] {
                int seen = born;
                for (;;) {
                    {
                        std::unique_lock<std::mutex> lk(m);
                        cv_go.wait(lk, [&] { return quit || generation != seen; });
                        if (quit) return;
                        seen = generation;
                    }
                    if (idx < active)
                        for (int i; (i = next.fetch_add(1)) < n_items;) fn(i);
                    std::lock_guard<std::mutex> lk(m);
                    if (--running == 0) cv_done.notify_one();
                }
            });
    }
    int active = 0;   // workers that take items of the current job (a lowered host_threads setting: the others just check in)
    void run(int n, std::function<void(int)> f)
    {
        if (n <= 1 || th.empty()) {
            for (int i = 0; i < n; ++i) f(i);
            return;
        }
        std::unique_lock<std::mutex> lk(m);
        fn = std::move(f);
        n_items = n;
        next = 0;
        running = (int)th.size();
        ++generation;
        cv_go.notify_all();
        cv_done.wait(lk, [&] { return running == 0; });
    }
    ~WindowPool()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv_go.notify_all();
        for (auto &t : th) t.join();
    }
};
struct LbaCache {
    std::vector<Pass> passes;
    WindowPool pool;
};

// worker threads of a batch's per-window host work (structure build, staging): the handle's setting, else
// AOS2_LBA_HOST_THREADS, else the host's cores shared among the ranks of the node (LOCAL_WORLD_SIZE / WORLD_SIZE as the
// launcher exports them: 8 ranks x 2 handles x 32 threads would be 512 threads on 256 cores), at most 32, at most one per window
static int lba_host_threads(const aos2_lba *s, int nw)
{
    int n = s->host_threads;
    if (n <= 0)
        if (const char *e = getenv("AOS2_LBA_HOST_THREADS")) n = atoi(e);
    if (n <= 0) {
        int ranks = 1;
        const char *e = getenv("LOCAL_WORLD_SIZE");
        if (!e) e = getenv("WORLD_SIZE");
        if (e && atoi(e) > 1) ranks = atoi(e);
        n = std::min(32, std::max(1, (int)std::thread::hardware_concurrency() / ranks));
    }
    return std::max(1, std::min(n, nw));
}

static bool build_pass(const aos2_lba_problem_t *p, Pass &S)
{
    // (the vectors keep their storage from call to call: a 24 k-edge window holds ~1.5 MB of index lists, and fresh
    // allocations of that size are mmap / page faults / munmap every time -- 2 ms per 32-window call, half of it at scope exit)
    S.hpose.clear(); S.hpoint.clear(); S.it_ka.clear(); S.it_kb.clear(); S.it_l.clear(); S.blk_off.clear();
    S.np = S.nl = 0;
    const int E = p->n_edges;
    std::vector<uint8_t> pose_act(p->n_poses, 0), point_act(p->n_points, 0);
    for (int e = 0; e < E; ++e) {
        pose_act[p->edge_pose[e]] = 1;
        point_act[p->edge_point[e]] = 1;
    }
    std::vector<int32_t> pose_h(p->n_poses, -1), point_h(p->n_points, -1);
    for (int i = 0; i < p->n_poses; ++i)
        if (pose_act[i] && !p->pose_fixed[i]) S.hpose.push_back(i);
    auto by_pose_id = [&](int a, int b) { return p->pose_id[a] < p->pose_id[b]; };
    if (!std::is_sorted(S.hpose.begin(), S.hpose.end(), by_pose_id)) std::stable_sort(S.hpose.begin(), S.hpose.end(), by_pose_id);
    for (int i = 0; i < p->n_points; ++i)
        if (point_act[i]) S.hpoint.push_back(i);
    auto by_point_id = [&](int a, int b) { return p->point_id[a] < p->point_id[b]; };
    if (!std::is_sorted(S.hpoint.begin(), S.hpoint.end(), by_point_id)) std::stable_sort(S.hpoint.begin(), S.hpoint.end(), by_point_id);
    S.np = (int)S.hpose.size();
    S.nl = (int)S.hpoint.size();
    for (int i = 0; i < S.np; ++i) pose_h[S.hpose[i]] = i;
    for (int i = 0; i < S.nl; ++i) point_h[S.hpoint[i]] = i;
    S.k_ph.resize(E);
    S.k_lh.resize(E);
    S.pt_off.assign(S.nl + 1, 0);
    S.ps_off.assign(S.np + 1, 0);
    S.pl_off.assign(S.nl + 1, 0);
    for (int k = 0; k < E; ++k) {
        S.k_ph[k] = pose_h[p->edge_pose[k]];
        S.k_lh[k] = point_h[p->edge_point[k]];
        S.pt_off[S.k_lh[k] + 1]++;
        if (S.k_ph[k] >= 0) {
            S.ps_off[S.k_ph[k] + 1]++;
            S.pl_off[S.k_lh[k] + 1]++;
        }
    }
    for (int i = 0; i < S.nl; ++i) {
        S.pt_off[i + 1] += S.pt_off[i];
        S.pl_off[i + 1] += S.pl_off[i];
    }
    for (int i = 0; i < S.np; ++i) S.ps_off[i + 1] += S.ps_off[i];
    S.pt_k.resize(S.pt_off[S.nl]);
    S.ps_k.resize(S.ps_off[S.np]);
    S.pl_k.resize(S.pl_off[S.nl]);
    std::vector<int> f1(S.nl, 0), f2(S.np, 0), f3(S.nl, 0);
    for (int k = 0; k < E; ++k) {
        const int l = S.k_lh[k], ph = S.k_ph[k];
        S.pt_k[S.pt_off[l] + f1[l]++] = k;
        if (ph >= 0) {
            S.ps_k[S.ps_off[ph] + f2[ph]++] = k;
            S.pl_k[S.pl_off[l] + f3[l]++] = k;
        }
    }
    // per landmark: ascending pose index, ties in edge order (a stable insertion sort: the runs are a handful of
    // edges long, and std::stable_sort would allocate a buffer for each of the thousands of runs)
    for (int l = 0; l < S.nl; ++l) {
        int32_t *q = S.pl_k.data() + S.pl_off[l];
        const int m = S.pl_off[l + 1] - S.pl_off[l];
        for (int i = 1; i < m; ++i) {
            const int32_t v = q[i], key = S.k_ph[v];
            int j = i - 1;
            for (; j >= 0 && S.k_ph[q[j]] > key; --j) q[j + 1] = q[j];
            q[j + 1] = v;
        }
        for (int i = 1; i < m; ++i)
            if (S.k_ph[q[i]] == S.k_ph[q[i - 1]]) return false;   // two edges between one keyframe and one landmark
    }
    S.pl_pos.assign(E, -1);
    S.pl_ph.resize(S.pl_k.size());
    for (size_t a = 0; a < S.pl_k.size(); ++a) {
        S.pl_pos[S.pl_k[a]] = (int32_t)a;
        S.pl_ph[a] = S.k_ph[S.pl_k[a]];
    }
    return true;
}

// items ranked by (pose, pose) block -- upper block triangle, row-major -- and by landmark inside a block (counting
// sort; this order is the summation order of k_schur_blocks)
static void build_schur_items(Pass &S)
{
    size_t n = 0;
    for (int l = 0; l < S.nl; ++l) {
        const size_t m = (size_t)(S.pl_off[l + 1] - S.pl_off[l]);
        n += m * (m + 1) / 2;
    }
    const int np = S.np;
    const size_t nblk = (size_t)np * (np + 1) / 2;
    S.it_ka.resize(n); S.it_kb.resize(n); S.it_l.resize(n);
    S.blk_off.assign(nblk + 1, 0);
    // block of (i1 <= i2) = row_base[i1] + i2 (pl_k is sorted by pose, so a <= b gives i1 <= i2)
    std::vector<int32_t> row_base(np > 0 ? np : 1);
    const std::vector<int32_t> &ph = S.pl_ph;
    for (int i = 0; i < np; ++i) row_base[i] = i * np - i * (i - 1) / 2 - i;
    for (int l = 0; l < S.nl; ++l) {
        const int32_t *q = ph.data() + S.pl_off[l];
        const int m = S.pl_off[l + 1] - S.pl_off[l];
        for (int a = 0; a < m; ++a) {
            int32_t *cnt = S.blk_off.data() + row_base[q[a]] + 1;
            for (int b = a; b < m; ++b) cnt[q[b]]++;
        }
    }
    for (size_t i = 0; i < nblk; ++i) S.blk_off[i + 1] += S.blk_off[i];
    std::vector<int32_t> fill(S.blk_off.begin(), S.blk_off.end() - 1);
    for (int l = 0; l < S.nl; ++l) {
        const int c0 = S.pl_off[l], m = S.pl_off[l + 1] - c0;
        const int32_t *q = ph.data() + c0;
        for (int a = 0; a < m; ++a) {   // (items name the two edges by their positions in the pl list = the indices of the dense Hpl array)
            int32_t *f = fill.data() + row_base[q[a]];
            const int32_t ka = c0 + a;
            for (int b = a; b < m; ++b) {
                const int slot = f[q[b]]++;
                S.it_ka[slot] = ka;
                S.it_kb[slot] = c0 + b;
                S.it_l[slot] = l;
            }
        }
    }
}

// k_schur's units of a window (see the kernel): DIAG units first (the long ones must not be dispatched last), then the BIG
// off-diagonal blocks, then the PACK units -- off-diagonal blocks in rank order, ceil(n / 16) rows each (an empty block keeps one
// row: its zeros are written like any other sum), a block never split across two units.
static void build_schur_units(Pass &S)
{
    S.sr_o0.clear(); S.sr_info.clear(); S.sr_ij.clear(); S.units.clear();
    const int np = S.np;
    for (int i = 0; i < np; ++i) S.units.push_back((kSchurDiag << 28) | i);
    std::vector<int32_t> pack_first;
    int in_unit = 0;
    for (int i1 = 0, blk = 0; i1 < np; ++i1)
        for (int i2 = i1; i2 < np; ++i2, ++blk) {
            if (i2 == i1) continue;
            const int o0 = S.blk_off[blk], n = S.blk_off[blk + 1] - o0;
            if (n > 256) {
                S.units.push_back((kSchurBig << 28) | blk);
                continue;
            }
            const int nrows = std::max(1, (n + 15) / 16);
            if (in_unit + nrows > 16) {   // pad the unit: a block's rows stay together
                for (; in_unit < 16; ++in_unit) {
                    S.sr_o0.push_back(0); S.sr_info.push_back(0); S.sr_ij.push_back(0);
                }
                in_unit = 0;
            }
            if (in_unit == 0) pack_first.push_back((int32_t)S.sr_o0.size());
            for (int r = 0; r < nrows; ++r) {
                S.sr_o0.push_back(o0 + 16 * r);
                S.sr_info.push_back(std::max(0, std::min(16, n - 16 * r)) | (r << 8) | (nrows << 16));
                S.sr_ij.push_back(i1 | (i2 << 16));
            }
            in_unit = (in_unit + nrows) % 16;
        }
    for (int32_t f : pack_first) S.units.push_back((kSchurPack << 28) | f);
}

// byte offsets of one window's regions in the arena
struct WinLayout {
    // staged (uploaded)
    size_t in_Tcw, in_xyz, in_obs, in_w, e_pose, e_point, e_stereo, pl_pos, hpose, hpoint, pt_off, pt_k, ps_off, ps_k,
        pl_off, pl_k, it_ka, it_kb, it_l, blk_off, sr_o0, sr_info, sr_ij;
    // device only
    size_t est, bk, robust, level1, err, lrec, lomr, Rl, Hpp, Hll, b, x, Hs, bs, tmp, scal, part, ldlt;
    // results (downloaded)
    size_t out_Tcw, out_xyz, out_outlier, out_chi2, st;
    int n_part, npad, ldlt_lds;
    size_t n_items;
};


// ROCTx ranges around the host-side phases of a solve (AOS2_ROCTX=1; rocprofv3 --marker-trace shows them next to the
// kernels).  The library is looked up at run time: no link dependency.  g2o's statistics buckets
// (Thirdparty/g2o/g2o/core/batch_stats.h, filled in block_solver.hpp:441-453 and sparse_optimizer.cpp:376-414) map to
// the kernels of the device program: timeResiduals -> k_points, timeLinearize + timeQuadraticForm -> k_lin,
// timeSchurComplement -> k_schur, timeLinearSolver -> k_ldlt_lds / k_ldlt_solve, timeUpdate -> the pose
// update inside the LDL^T kernel and the landmark update inside k_points.
struct RoctxRange {
    typedef int (*push_t)(const char *);
    typedef int (*pop_t)();
    static void resolve(push_t &push, pop_t &pop)
    {
        static push_t s_push = nullptr;
        static pop_t s_pop = nullptr;
        static bool tried = false;
        if (!tried) {
            tried = true;
            const char *v = getenv("AOS2_ROCTX");
            if (v && atoi(v) != 0) {
                void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
                if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
                if (h) {
                    s_push = (push_t)dlsym(h, "roctxRangePushA");
                    s_pop = (pop_t)dlsym(h, "roctxRangePop");
                }
            }
        }
        push = s_push;
        pop = s_pop;
    }
    pop_t pop_ = nullptr;
    explicit RoctxRange(const char *name)
    {
        push_t push;
        resolve(push, pop_);
        if (push && pop_) push(name); else pop_ = nullptr;
    }
    ~RoctxRange()
    {
        if (pop_) pop_();
    }
};

struct Bump {
    size_t size = 0;
    size_t take(size_t bytes, size_t align = 256)
    {
        const size_t off = (size + align - 1) & ~(align - 1);
        size = off + bytes;
        return off;
    }
};

}  // namespace aos2

using namespace aos2;

extern "C" {

int aos2_lba_create(int device, aos2_lba_t **out)
{
    if (!out) return AOS2_ERR_ARG;
    aos2_lba *s = new aos2_lba();
    s->device = device;
    *out = s;
    return AOS2_OK;
}

void aos2_lba_destroy(aos2_lba_t *s)
{
    if (!s) return;
    delete static_cast<LbaCache *>(s->lba_cache);
    if (s->dev_ready) {
        (void)hipSetDevice(s->device);
        (void)hipStreamSynchronize(s->stream);
        s->arena.release();
        s->h_stage.release();
        s->h_in.release();
        s->h_abort.release();
        for (auto &e : s->ev) (void)hipEventDestroy(e);
        (void)hipEventDestroy(s->ev_fork);
        (void)hipEventDestroy(s->ev_join);
        if (s->stream_b) {
            for (hipEvent_t e : {s->ev_fork_b, s->ev_join_b, s->ev_up, s->ev_stag, s->ev_done_b}) (void)hipEventDestroy(e);
            (void)hipStreamSynchronize(s->stream_b);
            (void)hipStreamDestroy(s->stream2_b);
            (void)hipStreamDestroy(s->stream_b);
        }
        (void)hipStreamDestroy(s->stream2);
        (void)hipStreamDestroy(s->stream);
    }
    delete s;
}

int aos2_lba_set_host_threads(aos2_lba_t *s, int n)
{
    if (!s || n < 0) {
        set_error("aos2_lba_set_host_threads: bad argument");
        return AOS2_ERR_ARG;
    }
    s->host_threads = n;
    return AOS2_OK;
}

int aos2_lba_set_window_groups(aos2_lba_t *s, int n)
{
    if (!s || n < 0 || n > 2) {
        set_error("aos2_lba_set_window_groups: 0 (default), 1 or 2");
        return AOS2_ERR_ARG;
    }
    s->window_groups = n;
    return AOS2_OK;
}

int aos2_lba_last_program(const aos2_lba_t *s, int32_t *trial_slots, int32_t *host_rounds)
{
    if (!s) {
        set_error("aos2_lba_last_program: no handle");
        return AOS2_ERR_ARG;
    }
    if (trial_slots) *trial_slots = s->last_trial_slots;
    if (host_rounds) *host_rounds = s->last_host_rounds;
    return AOS2_OK;
}

// The host part of aos2_lba_solve_batch without a device: the per-window index structures (build_pass, Schur items and units)
// and the staging copies (into pageable memory here), on `threads` worker threads.  For measuring / testing how the host phases of
// several ranks share a node's cores (tests/test_sharding_cpu.py); returns the wall time of the two phases.
int aos2_lba_debug_host_phase(const aos2_lba_problem_t *problems, int n_problems, int threads, double *build_ms, double *stage_ms)
{
    if (!problems || n_problems <= 0 || threads <= 0) {
        set_error("aos2_lba_debug_host_phase: bad argument");
        return AOS2_ERR_ARG;
    }
    std::vector<Pass> passes(n_problems);
    WindowPool pool;
    if (n_problems > 1 && threads > 1) pool.start(std::min(threads, n_problems));
    std::vector<uint8_t> ok(n_problems, 1);
    const auto t0 = std::chrono::steady_clock::now();
    pool.run(n_problems, [&](int i) {
        const aos2_lba_problem_t *p = problems + i;
        for (int e = 0; e < p->n_edges; ++e)
            if (p->edge_pose[e] < 0 || p->edge_pose[e] >= p->n_poses || p->edge_point[e] < 0 || p->edge_point[e] >= p->n_points) {
                ok[i] = 0;
                return;
            }
        ok[i] = build_pass(p, passes[i]) ? 1 : 0;
        if (ok[i] && passes[i].np > 0) {
            build_schur_items(passes[i]);
            build_schur_units(passes[i]);
        }
    });
    const auto t1 = std::chrono::steady_clock::now();
    for (int i = 0; i < n_problems; ++i)
        if (!ok[i]) {
            set_error("problem %d: bad edge", i);
            return AOS2_ERR_ARG;
        }
    std::vector<std::vector<uint8_t>> stage(n_problems);
    pool.run(n_problems, [&](int i) {
        const aos2_lba_problem_t *p = problems + i;
        const Pass &S = passes[i];
        const size_t NP = p->n_poses, NL = p->n_points, E = p->n_edges;
        const std::vector<int32_t> *lists[] = {&S.pl_pos, &S.hpose, &S.hpoint, &S.pt_off, &S.pt_k, &S.ps_off, &S.ps_k, &S.pl_off, &S.pl_ph,
                                               &S.it_ka, &S.it_kb, &S.it_l, &S.blk_off, &S.sr_o0, &S.sr_info, &S.sr_ij};
        size_t bytes = 64 * NP + 12 * NL + 12 * E + 4 * E + 4 * E + 4 * E + E;
        for (auto *v : lists) bytes += 4 * v->size();
        std::vector<uint8_t> &buf = stage[i];
        buf.resize(bytes);
        uint8_t *d = buf.data();
        auto put = [&](const void *src, size_t n) {
            memcpy(d, src, n);
            d += n;
        };
        put(p->pose_Tcw, 64 * NP); put(p->point_xyz, 12 * NL); put(p->edge_obs, 12 * E); put(p->edge_inv_sigma2, 4 * E);
        put(p->edge_pose, 4 * E); put(p->edge_point, 4 * E); put(p->edge_stereo, E);
        for (auto *v : lists)
            if (!v->empty()) put(v->data(), 4 * v->size());
    });
    const auto t2 = std::chrono::steady_clock::now();
    if (build_ms) *build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (stage_ms) *stage_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    return AOS2_OK;
}

int aos2_lba_last_window_slots(const aos2_lba_t *s, int64_t *window_slots)
{
    if (!s || !window_slots) {
        set_error("aos2_lba_last_window_slots: bad argument");
        return AOS2_ERR_ARG;
    }
    *window_slots = s->last_window_slots;
    return AOS2_OK;
}

int aos2_lba_debug_stop_at_poll(aos2_lba_t *s, int poll)
{
    if (!s || poll < 0) return AOS2_ERR_ARG;
    s->debug_stop_at_poll = poll;
    return AOS2_OK;
}

int aos2_lba_solve_batch(aos2_lba_t *s, const aos2_lba_problem_t *problems, aos2_lba_result_t *results, int n_problems)
{
    if (!s || !problems || !results || n_problems <= 0) {
        set_error("bad LocalBA batch");
        return AOS2_ERR_ARG;
    }
    bool want_chi2 = false;
    for (int w = 0; w < n_problems; ++w) {
        const aos2_lba_problem_t *p = problems + w;
        aos2_lba_result_t *r = results + w;
        if (p->n_poses <= 0 || p->n_points <= 0 || p->n_edges <= 0 || !p->pose_Tcw || !p->pose_fixed || !p->pose_id ||
            !p->point_xyz || !p->point_id || !p->edge_pose || !p->edge_point || !p->edge_obs || !p->edge_stereo ||
            !p->edge_inv_sigma2 || !r->pose_Tcw || !r->point_xyz) {
            set_error("bad LocalBA problem %d", w);
            return AOS2_ERR_ARG;
        }
        want_chi2 |= r->edge_chi2 != nullptr;   // (the edges' vertex indices are checked with the structure build, a thread per window)
    }
    // Optimizer.cc:656-658: return before optimising when the flag is already set; nothing is written back
    std::vector<int> act;
    for (int w = 0; w < n_problems; ++w) {
        const aos2_lba_problem_t *p = problems + w;
        aos2_lba_result_t *r = results + w;
        r->iters_done_first = r->iters_done_second = 0;
        r->trials_first = r->trials_second = 0;
        r->final_chi2 = r->final_lambda = 0;
        r->ms_device = 0;
        r->polls = 1;
        r->stop_poll = 0;
        r->status = AOS2_OK;
        if (stop_requested(p) || s->debug_stop_at_poll == 1) {
            memcpy(r->pose_Tcw, p->pose_Tcw, sizeof(float) * 16 * p->n_poses);
            memcpy(r->point_xyz, p->point_xyz, sizeof(float) * 3 * p->n_points);
            if (r->edge_outlier) memset(r->edge_outlier, 0, p->n_edges);
            r->status = AOS2_ERR_STOPPED;
            r->stop_poll = 1;
        } else
            act.push_back(w);
    }
    const int nw = (int)act.size();
    if (nw == 0) return AOS2_OK;
    int st = lba_handle_init(s);
    if (st) return st;
    // Window groups.  A trial is three wide launches (Schur, back-substitution, linearisation: every window's landmarks) and one
    // narrow one (the reduced systems: ONE workgroup per window, the longest launch of the mixed batch, during which most of the device
    // idles).  A batch of many windows runs as TWO groups with the same program each, on their own streams, the second started
    // behind the first group's first Schur launch: one group's reduced systems are factorised while the other group's landmark
    // kernels fill the device.  Windows are dealt to the groups by size (edges), largest first, so both get the same mix; the
    // windows of a group are neighbours in the descriptor array.  (aos2_lba_set_window_groups; AOS2_LBA_GROUPS overrides.)
    int G = s->window_groups ? s->window_groups : nw >= 16 ? 2 : 1;
    if (const char *e = getenv("AOS2_LBA_GROUPS")) G = std::max(1, std::min(2, atoi(e)));
    if (G > nw) G = 1;
    if (G == 2 && (st = lba_group_streams(s))) return st;
    int goff[3] = {0, nw, nw};
    if (G == 2) {
        std::vector<int> order(act);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return problems[a].n_edges > problems[b].n_edges; });
        std::vector<int> g0, g1;
        for (size_t k = 0; k < order.size(); ++k) (((k & 3) == 0 || (k & 3) == 3) ? g0 : g1).push_back(order[k]);   // a b b a | a b b a ...
        act = g0;
        act.insert(act.end(), g1.begin(), g1.end());
        goff[1] = (int)g0.size();
    }
    s->last_groups = G;
    // layout of the landmark kernels: kLmSlots threads per landmark while the landmarks of the call cannot fill the device
    // (one or a few windows: latency), one thread per landmark beyond (throughput); same results either way.
    // AOS2_LBA_LAYOUT=slots|walk forces one (tests).
    size_t total_points = 0;
    for (int i = 0; i < nw; ++i) total_points += (size_t)problems[act[i]].n_points;
    bool walk = total_points > 16000;
    if (const char *e = getenv("AOS2_LBA_LAYOUT")) walk = !strcmp(e, "walk");
    const int lm_per_block = walk ? 128 : kLmBlock;
    const bool prof = getenv("AOS2_LBA_PROF") != nullptr;
    const bool ldlt_old = getenv("AOS2_LDLT") && !strcmp(getenv("AOS2_LDLT"), "old");
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!prof) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[lba] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t - t_prev).count());
        t_prev = t;
    };

    // ---- per-window structure (host; windows in parallel when there are several)
    auto rg = std::make_unique<RoctxRange>("LocalBA::buildStructure (index mapping, edge lists, Schur items)");
    if (!s->lba_cache) s->lba_cache = new LbaCache();
    std::vector<Pass> &passes = static_cast<LbaCache *>(s->lba_cache)->passes;
    if ((int)passes.size() < nw) passes.resize(nw);
    WindowPool &pool = static_cast<LbaCache *>(s->lba_cache)->pool;
    if (nw > 1) pool.start(lba_host_threads(s, nw));
    auto for_windows = [&](auto &&fn) { pool.run(nw, fn); };
    std::vector<uint8_t> pass_ok(nw, 1);
    std::vector<int> bad_edge(nw, -1);
    for_windows([&](int i) {
        const aos2_lba_problem_t *p = problems + act[i];
        for (int e = 0; e < p->n_edges; ++e)
            if (p->edge_pose[e] < 0 || p->edge_pose[e] >= p->n_poses || p->edge_point[e] < 0 || p->edge_point[e] >= p->n_points) {
                bad_edge[i] = e;
                return;
            }
        pass_ok[i] = build_pass(p, passes[i]) ? 1 : 0;
        if (pass_ok[i] && passes[i].np > 0) {
            build_schur_items(passes[i]);
            build_schur_units(passes[i]);
        } else
            passes[i].units.clear();
    });
    for (int i = 0; i < nw; ++i)
        if (bad_edge[i] >= 0) {
            set_error("problem %d: edge %d references a vertex out of range", act[i], bad_edge[i]);
            return AOS2_ERR_ARG;
        }
    for (int i = 0; i < nw; ++i)
        if (!pass_ok[i]) {   // (cannot come from the reference: KeyFrame observations are a map MapPoint -> index)
            set_error("problem %d: two edges connect the same keyframe and map point", act[i]);
            return AOS2_ERR_ARG;
        }
    lap("build_pass + items");
    rg = std::make_unique<RoctxRange>("LocalBA::stage + upload");

    // ---- arena layout: [staged inputs of all windows | descriptors][device-only scratch][results of all windows]
    std::vector<WinLayout> L(nw);
    Bump B;
    for (int i = 0; i < nw; ++i) {
        const aos2_lba_problem_t *p = problems + act[i];
        const Pass &S = passes[i];
        WinLayout &l = L[i];
        const size_t NP = p->n_poses, NL = p->n_points, E = p->n_edges;
        l.in_Tcw = B.take(64 * NP); l.in_xyz = B.take(12 * NL); l.in_obs = B.take(12 * E); l.in_w = B.take(4 * E);
        l.e_pose = B.take(4 * E); l.e_point = B.take(4 * E); l.e_stereo = B.take(E);
        l.pl_pos = B.take(4 * E);
        l.hpose = B.take(4 * (size_t)S.np + 4); l.hpoint = B.take(4 * (size_t)S.nl + 4);
        l.pt_off = B.take(4 * ((size_t)S.nl + 1)); l.pt_k = B.take(4 * S.pt_k.size() + 4);
        l.ps_off = B.take(4 * ((size_t)S.np + 1)); l.ps_k = B.take(4 * S.ps_k.size() + 4);
        l.pl_off = B.take(4 * ((size_t)S.nl + 1)); l.pl_k = B.take(4 * S.pl_k.size() + 4);
        l.n_items = S.it_ka.size();
        l.it_ka = B.take(4 * l.n_items + 4); l.it_kb = B.take(4 * l.n_items + 4); l.it_l = B.take(4 * l.n_items + 4);
        l.blk_off = B.take(4 * S.blk_off.size() + 4);
        l.sr_o0 = B.take(4 * S.sr_o0.size() + 4); l.sr_info = B.take(4 * S.sr_o0.size() + 4); l.sr_ij = B.take(4 * S.sr_o0.size() + 4);
    }
    // k_schur's task list: the units of all windows, window w on XCD w' (workgroups go round-robin to the 8 XCDs in linear-id order,
    // each with an L2 of its own -- 4 MB, about one window's working set: the Hpl blocks a window's items share are then served by
    // one L2), the windows dealt to the XCDs largest first (every XCD gets about the same number of units), two windows of an XCD
    // at a time, unit by unit -- their DIAG units first.  Padding entries (w = -1) keep the 8 queues in step.
    // The task lists of a set of windows `ids` (indices into passes / L), the windows addressed as wmap(position): the Schur units
    // (see above), then the landmark kernels' lists: block b of every window before block b + 1 of any (the windows advance side by
    // side); k_lin: every landmark block first (the long dependent chains of the launch), then one task per free keyframe
    struct TaskLists {
        std::vector<SchurTask> schur, pts, lin;
    };
    const int lin_block = walk ? 256 : kLmBlock;
    auto build_tasks = [&](const std::vector<int> &ids, auto &&wmap, TaskLists &T, bool one_queue = false) {
        std::vector<SchurTask> &tasks = T.schur;
        const int nwg = (int)ids.size();
        std::vector<int> order(nwg);
        std::iota(order.begin(), order.end(), 0);
        // (dealt by unit count.  By estimated cost instead -- a DIAG / BIG unit = 2 + ceil(items / 256) workgroup rounds, a PACK unit one --
        // the mixed 64-window batch took 4.15 / 4.17 / 4.14 ms against 4.14 / 4.13 / 4.12: the XCD queues are not what k_schur's tail waits for)
        auto weight = [&](int k) { return passes[ids[k]].units.size(); };
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return weight(a) > weight(b); });
        const int NX = nwg >= 8 && !one_queue ? 8 : 1;
        std::vector<std::vector<int>> xw(NX);
        std::vector<size_t> load(NX, 0);
        for (int k : order) {
            const int x = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            xw[x].push_back(k);
            load[x] += weight(k);
        }
        std::vector<std::vector<SchurTask>> qx(NX);
        for (int x = 0; x < NX; ++x)
            for (size_t k = 0; k < xw[x].size(); k += 2) {
                const int a = xw[x][k], b = k + 1 < xw[x].size() ? xw[x][k + 1] : -1;
                const std::vector<int32_t> &ua = passes[ids[a]].units;
                static const std::vector<int32_t> none;
                const std::vector<int32_t> &ub = b >= 0 ? passes[ids[b]].units : none;
                for (size_t u = 0; u < std::max(ua.size(), ub.size()); ++u) {
                    if (u < ua.size()) qx[x].push_back(SchurTask{wmap(a), ua[u]});
                    if (u < ub.size()) qx[x].push_back(SchurTask{wmap(b), ub[u]});
                }
            }
        size_t mxq = 0;
        for (auto &q_ : qx) mxq = std::max(mxq, q_.size());
        tasks.assign(mxq * NX, SchurTask{-1, 0});
        for (int x = 0; x < NX; ++x)
            for (size_t k = 0; k < qx[x].size(); ++k) tasks[k * NX + x] = qx[x][k];
        int mx_pb = 0, mx_lb = 0, mx_k = 0;
        std::vector<int> npb(nwg), nlb(nwg);
        for (int k = 0; k < nwg; ++k) {
            const Pass &S = passes[ids[k]];
            npb[k] = std::max(1, (S.nl + lm_per_block - 1) / lm_per_block);   // = WinLayout::n_part
            nlb[k] = (S.nl + lin_block - 1) / lin_block;
            mx_pb = std::max(mx_pb, npb[k]); mx_lb = std::max(mx_lb, nlb[k]); mx_k = std::max(mx_k, S.np);
        }
        T.pts.clear();
        T.lin.clear();
        for (int b = 0; b < mx_pb; ++b)
            for (int k = 0; k < nwg; ++k)
                if (b < npb[k]) T.pts.push_back(SchurTask{wmap(k), b});
        for (int b = 0; b < mx_lb; ++b)
            for (int k = 0; k < nwg; ++k)
                if (b < nlb[k]) T.lin.push_back(SchurTask{wmap(k), b});
        for (int kf = 0; kf < mx_k; ++kf)
            for (int k = 0; k < nwg; ++k)
                if (kf < passes[ids[k]].np) T.lin.push_back(SchurTask{wmap(k), (1 << 28) | kf});
    };
    TaskLists TL[2];
    std::vector<SchurTask> *tasks_g[2] = {&TL[0].schur, &TL[1].schur}, *pts_tasks_g[2] = {&TL[0].pts, &TL[1].pts}, *lin_tasks_g[2] = {&TL[0].lin, &TL[1].lin};
    size_t n_all_tasks = 0;
    for (int g = 0; g < G; ++g) {
        std::vector<int> ids(goff[g + 1] - goff[g]);
        std::iota(ids.begin(), ids.end(), goff[g]);
        build_tasks(ids, [&](int k) { return ids[k]; }, TL[g]);   // (the groups' kernels index the whole descriptor array)
        n_all_tasks += TL[g].schur.size() + TL[g].pts.size() + TL[g].lin.size();
    }
    const size_t o_tasks = B.take(sizeof(SchurTask) * n_all_tasks + 8);
    const size_t o_wins = B.take(sizeof(LbaWin) * (size_t)nw);
    const size_t staged_bytes = B.size;
    // (a continuation round runs on the windows that are not finished, compacted: their descriptors and task lists go here)
    // Sized for ANY subset of the windows, not for the first round's lists: a subset is dealt to the 8 queues anew (largest first; with
    // queues of about equal length the padded list holds about sum + 8 x max units -- a first round of fewer than 8 windows per group
    // is not padded at all, and the dealing is not monotone on subsets); the landmark / linearisation lists of a subset are parts of
    // the full ones.  A subset whose padded list is longer still falls back to one unpadded queue (below), which always fits.
    size_t cont_tasks = 0, cont_max_units = 0;
    for (int i = 0; i < nw; ++i) {
        const Pass &S = passes[i];
        cont_max_units = std::max(cont_max_units, S.units.size());
        cont_tasks += S.units.size() + (size_t)std::max(1, (S.nl + lm_per_block - 1) / lm_per_block) + (size_t)(S.nl + lin_block - 1) / lin_block + (size_t)S.np;
    }
    cont_tasks += 8 * cont_max_units;
    const size_t cont_bytes = sizeof(LbaWin) * (size_t)nw + sizeof(SchurTask) * std::max(cont_tasks, n_all_tasks) + 64;
    const size_t o_cont = B.take(cont_bytes);
    for (int i = 0; i < nw; ++i) {
        const aos2_lba_problem_t *p = problems + act[i];
        const Pass &S = passes[i];
        WinLayout &l = L[i];
        const size_t NP = p->n_poses, NL = p->n_points, E = p->n_edges;
        const size_t n6 = 6 * (size_t)S.np, dim = n6 + 3 * (size_t)S.nl;
        l.est = B.take(8 * (7 * NP + 3 * NL)); l.bk = B.take(8 * (7 * NP + 3 * NL));
        l.robust = B.take(E); l.level1 = B.take(E);
        l.err = B.take(24 * E);
        l.lrec = B.take(32 * (S.pl_k.size() + 1)); l.lomr = B.take(24 * (S.pl_k.size() + 1)); l.Rl = B.take(72 * (size_t)S.np + 8);
        l.Hpp = B.take(288 * (size_t)S.np + 8); l.Hll = B.take(72 * (size_t)S.nl + 8);
        l.b = B.take(8 * dim + 8); l.x = B.take(8 * dim + 8);
        l.npad = (int)((n6 + 15) & ~(size_t)15);
        {
            const size_t ldlt_bytes = ((size_t)l.npad * (l.npad + 1) + (size_t)l.npad * 17 + 4 * (size_t)l.npad + 2 * 16 * 17 + 16) * 8;
            l.ldlt_lds = ldlt_bytes <= 159 * 1024 ? 1 : 0;
            // the register-resident form (k_ldlt_reg) serves every window of up to 40 free keyframes; AOS2_LDLT=old keeps the two
            // earlier forms (LDS-resident up to 21 free keyframes, in place in device memory beyond) for comparison
            if (l.npad <= 16 * kLrMaxNb && !ldlt_old) l.ldlt_lds = 2;
        }
        // (beyond LDS the reduced system is factorised in place: k_schur writes it with leading dimension npad)
        l.Hs = B.take(l.ldlt_lds == 1 ? 8 * n6 * n6 + 8 : 8 * (size_t)l.npad * l.npad + 8, 256); l.bs = B.take(8 * n6 + 8);
        l.tmp = B.take(8 * n6 + 8);
        l.n_part = std::max(1, (int)((S.nl + lm_per_block - 1) / lm_per_block));
        l.scal = B.take(64); l.part = B.take(16 * (size_t)std::max(S.nl, 1) + 8);
        l.ldlt = B.take(8);
    }
    const size_t o_res = B.take(0);
    for (int i = 0; i < nw; ++i) {
        const aos2_lba_problem_t *p = problems + act[i];
        WinLayout &l = L[i];
        const size_t NP = p->n_poses, NL = p->n_points, E = p->n_edges;
        l.st = B.take(sizeof(LmState), 64);
        l.out_Tcw = B.take(64 * NP, 64); l.out_xyz = B.take(12 * NL, 64); l.out_outlier = B.take(E, 64);
        l.out_chi2 = want_chi2 ? B.take(8 * E, 64) : 0;
    }
    const size_t res_bytes = B.size - o_res;
    if ((st = s->arena.alloc(B.size + 256))) return st;
    if ((st = s->h_in.alloc(o_cont + cont_bytes + 256))) return st;
    if ((st = s->h_stage.alloc(res_bytes + 256))) return st;
    if ((st = s->h_abort.alloc((size_t)nw + 1))) return st;
    uint8_t *base = s->arena.p, *hin = s->h_in.p;
    int32_t *d_abort = nullptr;
    AOS2_HIP_CHECK(hipHostGetDevicePointer((void **)&d_abort, s->h_abort.p, 0));
    bool any_flag = false;
    for (int i = 0; i < nw; ++i) {
        s->h_abort.p[i] = 0;
        any_flag |= problems[act[i]].stop_flag != nullptr;
    }
    lap("layout + arena");

    // ---- staging (parallel) + descriptors
    LbaWin *hw = reinterpret_cast<LbaWin *>(hin + o_wins);
    struct GroupDims {
        bool any_lds = false, any_glob = false, any_reg = false;
        int mx_E = 0, mx_pts = 0, mx_npad_glob = 0, mx_npad_lds = 0, mx_npad_reg = 0;
    } gd[2];
    std::vector<uint8_t> up_fail(nw, 0);
    for_windows([&](int i) {
        const aos2_lba_problem_t *p = problems + act[i];
        const Pass &S = passes[i];
        const WinLayout &l = L[i];
        const size_t NP = p->n_poses, NL = p->n_points, E = p->n_edges;
        memcpy(hin + l.in_Tcw, p->pose_Tcw, 64 * NP);
        memcpy(hin + l.in_xyz, p->point_xyz, 12 * NL);
        memcpy(hin + l.in_obs, p->edge_obs, 12 * E);
        memcpy(hin + l.in_w, p->edge_inv_sigma2, 4 * E);
        memcpy(hin + l.e_pose, p->edge_pose, 4 * E);
        memcpy(hin + l.e_point, p->edge_point, 4 * E);
        memcpy(hin + l.e_stereo, p->edge_stereo, E);
        auto put = [&](size_t off, const std::vector<int32_t> &v) {
            if (!v.empty()) memcpy(hin + off, v.data(), v.size() * 4);
        };
        put(l.pl_pos, S.pl_pos); put(l.hpose, S.hpose); put(l.hpoint, S.hpoint);
        put(l.pt_off, S.pt_off); put(l.pt_k, S.pt_k); put(l.ps_off, S.ps_off); put(l.ps_k, S.ps_k);
        put(l.pl_off, S.pl_off); put(l.pl_k, S.pl_ph);
        put(l.it_ka, S.it_ka); put(l.it_kb, S.it_kb); put(l.it_l, S.it_l); put(l.blk_off, S.blk_off);
        put(l.sr_o0, S.sr_o0); put(l.sr_info, S.sr_info); put(l.sr_ij, S.sr_ij);
        // the window's staged region goes to the device as soon as it is assembled: its upload overlaps the staging of the
        // other windows (one copy of everything after the staging cost 0.4 ms more per 32-window call)
        const size_t r0 = l.in_Tcw, r1 = i + 1 < nw ? L[i + 1].in_Tcw : o_tasks;
        if (hipSetDevice(s->device) != hipSuccess || hipMemcpyAsync(base + r0, hin + r0, r1 - r0, hipMemcpyHostToDevice, s->stream) != hipSuccess)
            up_fail[i] = 1;
    });
    for (int i = 0; i < nw; ++i)
        if (up_fail[i]) {
            set_error("LocalBA: upload of window %d failed", act[i]);
            return AOS2_ERR_HIP;
        }
    for (int i = 0; i < nw; ++i) {
        const aos2_lba_problem_t *p = problems + act[i];
        const Pass &S = passes[i];
        const WinLayout &l = L[i];
        LbaWin &W = hw[i];
        memset(&W, 0, sizeof(W));
        W.n_poses = p->n_poses; W.n_points = p->n_points; W.n_edges = p->n_edges;
        W.np = S.np; W.nl = S.nl; W.n_items = (int)l.n_items;
        W.iters1 = p->iters_first; W.iters2 = p->iters_second;
        W.in_Tcw = (const float *)(base + l.in_Tcw); W.in_xyz = (const float *)(base + l.in_xyz);
        W.in_obs = (const float *)(base + l.in_obs); W.in_w = (const float *)(base + l.in_w);
        W.pose = (double *)(base + l.est); W.point = W.pose + 7 * (size_t)p->n_poses;
        W.bk = (double *)(base + l.bk);
        W.est_n = 7 * p->n_poses + 3 * p->n_points;
        W.e_pose = (const int32_t *)(base + l.e_pose); W.e_point = (const int32_t *)(base + l.e_point);

        W.e_stereo = base + l.e_stereo; W.e_robust = base + l.robust; W.e_level1 = base + l.level1;
        W.err = (double *)(base + l.err);
        W.cam.fx = (double)p->fx; W.cam.fy = (double)p->fy; W.cam.cx = (double)p->cx; W.cam.cy = (double)p->cy;
        W.cam.bf = (double)p->bf; W.cam.bf_f = p->bf;
        W.cam.delta_mono = (double)(float)std::sqrt(5.991);
        W.cam.delta_stereo = (double)(float)std::sqrt(7.815);
        W.pl_pos = (const int32_t *)(base + l.pl_pos);
        W.hpose = (const int32_t *)(base + l.hpose); W.hpoint = (const int32_t *)(base + l.hpoint);
        W.pt_off = (const int32_t *)(base + l.pt_off); W.pt_k = (const int32_t *)(base + l.pt_k);
        W.ps_off = (const int32_t *)(base + l.ps_off); W.ps_k = (const int32_t *)(base + l.ps_k);
        W.pl_off = (const int32_t *)(base + l.pl_off); W.pl_ph = (const int32_t *)(base + l.pl_k);
        W.it_ka = (const int32_t *)(base + l.it_ka); W.it_kb = (const int32_t *)(base + l.it_kb);
        W.it_l = (const int32_t *)(base + l.it_l); W.blk_off = (const int32_t *)(base + l.blk_off);
        W.sr_o0 = (const int32_t *)(base + l.sr_o0); W.sr_info = (const int32_t *)(base + l.sr_info); W.sr_ij = (const int32_t *)(base + l.sr_ij);
        W.n_srows = (int)S.sr_o0.size();
        W.lrec = (double *)(base + l.lrec); W.lomr = (double *)(base + l.lomr); W.Rl = (double *)(base + l.Rl); W.Hpp = (double *)(base + l.Hpp);
        W.Hll = (double *)(base + l.Hll); W.b = (double *)(base + l.b); W.x = (double *)(base + l.x);
        W.Hs = (double *)(base + l.Hs); W.bs = (double *)(base + l.bs);
        W.tmp = (double *)(base + l.tmp); W.scal = (double *)(base + l.scal); W.part = (double *)(base + l.part);
        W.n_part = l.n_part;
        W.ldlt = (double *)(base + l.ldlt); W.npad = l.npad; W.ldlt_lds = l.ldlt_lds;
        W.hs_ld = l.ldlt_lds == 1 ? 6 * S.np : l.npad;
        W.st = (LmState *)(base + l.st);
        W.abort_word = d_abort + i;
        W.out_Tcw = (float *)(base + l.out_Tcw); W.out_xyz = (float *)(base + l.out_xyz);
        W.out_outlier = base + l.out_outlier;
        W.out_chi2 = want_chi2 ? (double *)(base + l.out_chi2) : nullptr;
        GroupDims &D = gd[i >= goff[1] ? 1 : 0];
        if (S.np > 0) {
            if (l.ldlt_lds == 2) {
                D.any_reg = true;
                D.mx_npad_reg = std::max(D.mx_npad_reg, l.npad);
            } else if (l.ldlt_lds) {
                D.any_lds = true;
                D.mx_npad_lds = std::max(D.mx_npad_lds, l.npad);
            } else {
                D.any_glob = true;
                D.mx_npad_glob = std::max(D.mx_npad_glob, l.npad);
            }
        }
        D.mx_E = std::max(D.mx_E, p->n_edges);
        D.mx_pts = std::max(D.mx_pts, std::max(p->n_points, p->n_poses));
    }
    lap("staging");
    hipStream_t q = s->stream;
    // the task lists of the groups, one after the other: [schur | points | lin] per group
    const SchurTask *d_schur_tasks[2], *d_pts_tasks[2], *d_lin_tasks[2];
    {
        size_t o = o_tasks;
        for (int g = 0; g < G; ++g)
            for (int k = 0; k < 3; ++k) {
                const std::vector<SchurTask> &v = k == 0 ? *tasks_g[g] : k == 1 ? *pts_tasks_g[g] : *lin_tasks_g[g];
                (k == 0 ? d_schur_tasks : k == 1 ? d_pts_tasks : d_lin_tasks)[g] = (const SchurTask *)(base + o);
                if (!v.empty()) memcpy(hin + o, v.data(), sizeof(SchurTask) * v.size());
                o += sizeof(SchurTask) * v.size();
            }
    }
    AOS2_HIP_CHECK(hipMemcpyAsync(base + o_tasks, hin + o_tasks, staged_bytes - o_tasks, hipMemcpyHostToDevice, q));   // the task lists, the descriptors
    AOS2_HIP_CHECK(hipEventRecord(s->ev[0], q));
    const LbaWin *dw = (const LbaWin *)(base + o_wins);
    auto blocks = [](size_t n, int t) { return (unsigned)((n + t - 1) / t); };
    hipStream_t gq[2] = {s->stream, s->stream_b}, gq2[2] = {s->stream2, s->stream2_b};
    if (getenv("AOS2_LBA_GROUPS_SERIAL")) gq[1] = gq[0], gq2[1] = gq2[0];   // (debugging: the groups one after the other)
    hipEvent_t gfork[2] = {s->ev_fork, s->ev_fork_b}, gjoin[2] = {s->ev_join, s->ev_join_b};
    if (G == 2) {   // the second group's stream starts behind the upload
        AOS2_HIP_CHECK(hipEventRecord(s->ev_up, q));
        AOS2_HIP_CHECK(hipStreamWaitEvent(gq[1], s->ev_up, 0));
    }
    // What a launch sequence runs on: the descriptor array its task lists index (`wins`), the first descriptor and the number of
    // windows of the kernels that take one workgroup row per window (`blk`, nw), the lists, the largest dimensions, the streams
    struct Prog {
        const LbaWin *wins, *blk;
        int nw;
        const SchurTask *schur, *pts, *lin;
        size_t n_schur, n_pts, n_lin;
        GroupDims D;
        hipStream_t q, q2;
        hipEvent_t fork, join;
    };
    auto enqueue_points = [&](const Prog &P, int solve) {
        if (walk)
            hipLaunchKernelGGL(k_points_walk, dim3((unsigned)P.n_pts), dim3(128), 0, P.q, P.wins, P.pts, solve);
        else
            hipLaunchKernelGGL(k_points, dim3((unsigned)P.n_pts), dim3(256), 0, P.q, P.wins, P.pts, solve);
    };
    auto enqueue_lin = [&](const Prog &P, int init) {
        if (!P.n_lin) return;
        if (walk)
            hipLaunchKernelGGL(k_lin<true>, dim3((unsigned)P.n_lin), dim3(256), 0, P.q, P.wins, P.lin, init);
        else
            hipLaunchKernelGGL(k_lin<false>, dim3((unsigned)P.n_lin), dim3(256), 0, P.q, P.wins, P.lin, init);
    };
    auto enqueue_init = [&](const Prog &P) {
        enqueue_points(P, 0);
        enqueue_lin(P, 1);
        hipLaunchKernelGGL(k_lm_init, dim3(P.nw), dim3(1024), 0, P.q, P.blk);
    };
    // one Levenberg-Marquardt trial: 4 launches (5 with reduced systems of two kinds)
    bool stagger_pending = G == 2 && !getenv("AOS2_LBA_NO_STAGGER");
    auto enqueue_trial = [&](const Prog &P, bool first_group) {
        const GroupDims &D = P.D;
        if (P.n_schur) hipLaunchKernelGGL(k_schur, dim3((unsigned)P.n_schur), dim3(kSchurThreads), 0, P.q, P.wins, P.schur);
        if (first_group && stagger_pending) {   // the other group starts here: half a trial behind
            (void)hipEventRecord(s->ev_stag, gq[0]);
            (void)hipStreamWaitEvent(gq[1], s->ev_stag, 0);
            stagger_pending = false;
        }
        // the forms of the reduced-system kernel work on different windows: side by side (one of them on a stream of its own)
        // (k_ldlt_reg and k_ldlt_lds never meet in one call: AOS2_LDLT chooses for the whole call)
        const bool both = D.any_glob && (D.any_lds || D.any_reg);
        if (both) {
            (void)hipEventRecord(P.fork, P.q);
            (void)hipStreamWaitEvent(P.q2, P.fork, 0);
        }
        if (D.any_lds) {
            const size_t need = ((size_t)D.mx_npad_lds * (D.mx_npad_lds + 1) + (size_t)D.mx_npad_lds * 17 + 4 * (size_t)D.mx_npad_lds + 2 * 16 * 17 + 16) * sizeof(double);
            hipLaunchKernelGGL(k_ldlt_lds, dim3(P.nw), dim3(512), need, both ? P.q2 : P.q, P.blk);
        }
        if (D.any_reg)
            hipLaunchKernelGGL(k_ldlt_reg, dim3(P.nw), dim3(kLrThreads), ldlt_reg_lds_doubles(D.mx_npad_reg) * sizeof(double), both ? P.q2 : P.q, P.blk);
        if (D.any_glob)
            hipLaunchKernelGGL(k_ldlt_dev, dim3(P.nw), dim3(512), ((size_t)D.mx_npad_glob * 17 + 4 * (size_t)D.mx_npad_glob + 3 * 16 * 17 + 16) * sizeof(double), P.q,
                               P.blk);
        if (both) {
            (void)hipEventRecord(P.join, P.q2);
            (void)hipStreamWaitEvent(P.q, P.join, 0);
        }
        enqueue_points(P, 1);   // + the LM decision in its last workgroup
        enqueue_lin(P, 0);
    };
    auto enqueue_transition = [&](const Prog &P) {
        hipLaunchKernelGGL(k_transition, dim3(blocks(P.D.mx_E, 256), P.nw), dim3(256), 0, P.q, P.blk);
    };
    Prog PG[2];
    for (int g = 0; g < G; ++g)
        PG[g] = Prog{dw, dw + goff[g], goff[g + 1] - goff[g], d_schur_tasks[g], d_pts_tasks[g], d_lin_tasks[g], tasks_g[g]->size(), pts_tasks_g[g]->size(),
                     lin_tasks_g[g]->size(), gd[g], gq[g], gq2[g], gfork[g], gjoin[g]};
    // results (and states) come back as one copy; the host forwards pbStopFlag into the mapped abort words meanwhile
    auto finish = [&](const Prog *progs, int n_progs) -> int {
        for (int g = 0; g < n_progs; ++g) {
            const Prog &P = progs[g];
            hipLaunchKernelGGL(k_final, dim3(blocks(std::max(P.D.mx_E, P.D.mx_pts), 256), P.nw), dim3(256), 0, P.q, P.blk);
        }
        if (n_progs == 2) {
            AOS2_HIP_CHECK(hipEventRecord(s->ev_done_b, gq[1]));
            AOS2_HIP_CHECK(hipStreamWaitEvent(q, s->ev_done_b, 0));
        }
        AOS2_HIP_CHECK(hipEventRecord(s->ev[1], q));
        AOS2_HIP_CHECK(hipMemcpyAsync(s->h_stage.p, base + o_res, res_bytes, hipMemcpyDeviceToHost, q));
        if (any_flag) {
            AOS2_HIP_CHECK(hipEventRecord(s->ev[2], q));
            while (hipEventQuery(s->ev[2]) == hipErrorNotReady)
                for (int i = 0; i < nw; ++i)
                    if (stop_requested(problems + act[i])) __atomic_store_n(&s->h_abort.p[i], 1, __ATOMIC_RELAXED);
        }
        AOS2_HIP_CHECK(hipStreamSynchronize(q));
        AOS2_HIP_CHECK(hipGetLastError());
        return AOS2_OK;
    };
    int max_i1 = 0, max_i2 = 0;
    for (int i = 0; i < nw; ++i) {
        max_i1 = std::max(max_i1, problems[act[i]].iters_first);
        max_i2 = std::max(max_i2, problems[act[i]].iters_second);
    }
    rg = std::make_unique<RoctxRange>("LocalBA::optimize(5) + outlier pass + optimize(10) + inlier check (one device program)");
    // The program: as many trials as iterations per optimisation -- what every window needs whose steps are all accepted.  A window
    // with rejected steps is not finished when the program ends: it leaves with the others' results and gets a continuation round
    // sized for what it still needs, together with the (few) windows like it, compacted (below).  Rounds 2-4 enqueued one spare
    // trial per optimisation for EVERY window instead (AOS2_LBA_SPARE_SLOTS=1): 13 % of the launches of a batch whose windows need none.
    const int spare = getenv("AOS2_LBA_SPARE_SLOTS") ? atoi(getenv("AOS2_LBA_SPARE_SLOTS")) : 0;
    const int slots1 = max_i1 > 0 ? max_i1 + spare : 0, slots2 = max_i2 > 0 ? max_i2 + spare : 0;
    s->last_trial_slots = slots1 + slots2;
    s->last_window_slots = (long long)(slots1 + slots2) * nw;
    s->last_host_rounds = 1;
    for (int g = 0; g < G; ++g) {
        const Prog &P = PG[g];
        hipLaunchKernelGGL(k_prepare, dim3(blocks(std::max(P.D.mx_E, P.D.mx_pts), 256), P.nw), dim3(256), 0, P.q, P.blk, s->debug_stop_at_poll);
        if (max_i1 > 0) {
            enqueue_init(P);
            for (int t = 0; t < slots1; ++t) enqueue_trial(P, g == 0);
        }
        enqueue_transition(P);
        if (max_i2 > 0) {
            enqueue_init(P);
            for (int t = 0; t < slots2; ++t) enqueue_trial(P, g == 0);
        }
    }
    if ((st = finish(PG, G))) return st;
    lap("program");
    auto state_of = [&](int i) { return reinterpret_cast<const LmState *>(s->h_stage.p + (L[i].st - o_res)); };
    for (int round = 0;; ++round) {
        // the windows that are not finished, what they still need at most if no further step is rejected
        std::vector<int> todo;
        int need1 = 0, need2 = 0;
        bool before_second = false;
        for (int i = 0; i < nw; ++i) {
            const LmState *ls = state_of(i);
            if (ls->phase == 3) continue;
            todo.push_back(i);
            if (ls->phase == 0) need1 = std::max(need1, std::max(1, ls->iters_max[0] - ls->it));
            if (ls->phase <= 1) before_second = true;
            if (ls->phase == 2) need2 = std::max(need2, std::max(1, ls->iters_max[1] - ls->it));
        }
        if (todo.empty()) break;
        if (round > 64) {   // 2 x 10 iterations x 10 trials at most: cannot happen
            set_error("internal: LocalBA program did not finish");
            return AOS2_ERR_ARG;
        }
        if (before_second) need2 = std::max(need2, max_i2);
        // their descriptors, compacted, and task lists that address them by position
        TaskLists TC;
        build_tasks(todo, [](int k) { return k; }, TC);
        size_t tc_bytes = sizeof(SchurTask) * (TC.schur.size() + TC.pts.size() + TC.lin.size());
        static const bool force_one_queue = getenv("AOS2_LBA_CONT_ONE_QUEUE") != nullptr;   // (test hook: the fallback below on every continuation)
        if (force_one_queue || sizeof(LbaWin) * todo.size() + tc_bytes > cont_bytes) {   // (the dealing is not monotone on subsets: should one pad past the bound,
            build_tasks(todo, [](int k) { return k; }, TC, true);      //  one unpadded queue always fits -- its length is the subset's sum)
            tc_bytes = sizeof(SchurTask) * (TC.schur.size() + TC.pts.size() + TC.lin.size());
        }
        if (sizeof(LbaWin) * todo.size() + tc_bytes > cont_bytes) {
            set_error("internal: continuation region");
            return AOS2_ERR_ARG;
        }
        uint8_t *hc = hin + o_cont;
        Prog PC;
        PC.D = GroupDims();
        for (size_t k = 0; k < todo.size(); ++k) {
            const int i = todo[k];
            reinterpret_cast<LbaWin *>(hc)[k] = hw[i];
            const WinLayout &l = L[i];
            const Pass &S = passes[i];
            if (S.np > 0) {
                if (l.ldlt_lds == 2) {
                    PC.D.any_reg = true;
                    PC.D.mx_npad_reg = std::max(PC.D.mx_npad_reg, l.npad);
                } else if (l.ldlt_lds) {
                    PC.D.any_lds = true;
                    PC.D.mx_npad_lds = std::max(PC.D.mx_npad_lds, l.npad);
                } else {
                    PC.D.any_glob = true;
                    PC.D.mx_npad_glob = std::max(PC.D.mx_npad_glob, l.npad);
                }
            }
            PC.D.mx_E = std::max(PC.D.mx_E, problems[act[i]].n_edges);
            PC.D.mx_pts = std::max(PC.D.mx_pts, std::max(problems[act[i]].n_points, problems[act[i]].n_poses));
        }
        size_t o = sizeof(LbaWin) * todo.size();
        const SchurTask *dt[3];
        const std::vector<SchurTask> *tv[3] = {&TC.schur, &TC.pts, &TC.lin};
        for (int k = 0; k < 3; ++k) {
            dt[k] = (const SchurTask *)(base + o_cont + o);
            if (!tv[k]->empty()) memcpy(hc + o, tv[k]->data(), sizeof(SchurTask) * tv[k]->size());
            o += sizeof(SchurTask) * tv[k]->size();
        }
        AOS2_HIP_CHECK(hipMemcpyAsync(base + o_cont, hc, o, hipMemcpyHostToDevice, q));
        PC.wins = PC.blk = (const LbaWin *)(base + o_cont);
        PC.nw = (int)todo.size();
        PC.schur = dt[0]; PC.pts = dt[1]; PC.lin = dt[2];
        PC.n_schur = TC.schur.size(); PC.n_pts = TC.pts.size(); PC.n_lin = TC.lin.size();
        PC.q = gq[0]; PC.q2 = gq2[0]; PC.fork = gfork[0]; PC.join = gjoin[0];
        for (int t = 0; t < need1; ++t) enqueue_trial(PC, false);
        if (before_second) {
            enqueue_transition(PC);
            enqueue_init(PC);
        }
        for (int t = 0; t < need2; ++t) enqueue_trial(PC, false);
        s->last_trial_slots += need1 + need2;
        s->last_window_slots += (long long)(need1 + need2) * (long long)todo.size();
        s->last_host_rounds++;
        if ((st = finish(&PC, 1))) return st;
    }
    lap("continuation");
    rg = std::make_unique<RoctxRange>("LocalBA::write-back");
    if (getenv("AOS2_LBA_TRACE"))
        for (int i = 0; i < nw; ++i) {
            const LmState *ls = state_of(i);
#ifdef AOS2_TAIL_TIMING
            fprintf(stderr, "[lba]   lm_decide (AOS2_TAIL_TIMING builds): sums %lld, decision %lld ticks of 10 ns over %lld calls\n", ls->dbg[12], ls->dbg[13], ls->dbg[14]);
#endif
#ifdef AOS2_LDLT_TIMING
            fprintf(stderr, "[lba] win %d reduced-system kernel cycles: load %lld, D0 %lld, P %lld, U(+lookahead D) %lld, D inside U %lld, factor %lld, backward %lld\n", i,
                    ls->dbg[0], ls->dbg[1], ls->dbg[2], ls->dbg[3], ls->dbg[4], ls->dbg[5], ls->dbg[6]);
            fprintf(stderr, "[lba]   diagonal blocks (8 of them): load %lld, pivots %lld, store %lld, T %lld\n", ls->dbg[8], ls->dbg[9], ls->dbg[10], ls->dbg[11]);
#endif
            for (int t = 0; t < ls->ntr; ++t)
                fprintf(stderr, "[lba] win %d trial %2d lambda %.6e chi %.9e -> %.9e rho %.6e\n", i, t, ls->tr_lambda[t], ls->tr_cur[t], ls->tr_temp[t], ls->tr_rho[t]);
        }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, s->ev[0], s->ev[1]);
    for (int i = 0; i < nw; ++i) {
        const aos2_lba_problem_t *p = problems + act[i];
        aos2_lba_result_t *r = results + act[i];
        const WinLayout &l = L[i];
        const uint8_t *hb = s->h_stage.p;
        const LmState *ls = state_of(i);
        memcpy(r->pose_Tcw, hb + (l.out_Tcw - o_res), 64 * (size_t)p->n_poses);
        memcpy(r->point_xyz, hb + (l.out_xyz - o_res), 12 * (size_t)p->n_points);
        if (r->edge_outlier) memcpy(r->edge_outlier, hb + (l.out_outlier - o_res), (size_t)p->n_edges);
        if (r->edge_chi2) memcpy(r->edge_chi2, hb + (l.out_chi2 - o_res), 8 * (size_t)p->n_edges);
        r->iters_done_first = ls->iters_done[0];
        r->iters_done_second = ls->iters_done[1];
        r->trials_first = ls->trials[0];
        r->trials_second = ls->trials[1];
        r->final_chi2 = ls->final_chi2;
        r->final_lambda = ls->final_lambda;
        r->polls = ls->polls;
        r->stop_poll = ls->stop_poll;
        r->ms_device = ms;
    }
    lap("write-back");
    return AOS2_OK;
}

int aos2_lba_solve(aos2_lba_t *s, const aos2_lba_problem_t *p, aos2_lba_result_t *r)
{
    if (!s || !p || !r) {
        set_error("bad LocalBA problem");
        return AOS2_ERR_ARG;
    }
    const int st = aos2_lba_solve_batch(s, p, r, 1);
    return st ? st : r->status;
}

}  // extern "C"
