// Reduced camera system of one LocalBA window (the linear solve of g2o's BlockSolver::solve, block_solver.hpp:434-486 ->
// LinearSolverEigen, solvers/linear_solver_eigen.h:94-124; a dense LDL^T stands in for the sparse Cholesky):
// blocked right-looking LDL^T + both substitutions by ONE workgroup with the trailing matrix RESIDENT IN REGISTERS.
//
//   * the strictly lower 16x16 tiles (105 of them at npad = 240) live in the vector registers of kLrWorkers waves as MFMA
//     accumulator tiles for the whole factorisation (tile t -> worker t % kLrWorkers, slot t / kLrWorkers; tiles are ranked from the
//     bottom-right corner so that the tiles a panel still touches are always a PREFIX of the ranking: every worker has the same
//     share of every panel's update, and a slot is a compile-time register index);
//   * the diagonal tiles, the panel's L D (W), every panel's T_k = L_kk^-1 and the vectors live in LDS; nothing but the initial
//     load and the solution touches device memory (the in-place global-memory form this replaces paid a device-memory round
//     trip per panel phase and per backward block: 113 us at 40 free keyframes);
//   * wave 0 factorises the diagonal blocks: row i of the block in lane i, column c of T in lane 16 + c -- the SAME
//     instruction stream serves both (x[k] -= (x[j] / d_j) u_k) --, the next pivot's column broadcast (LDS) and reciprocal
//     (v_rcp_f64 + 2 Newton steps) in flight behind the current pivot's updates;
//   * a tile is stored TRANSPOSED while it belongs to the trailing matrix (so it is directly the A operand of the panel product
//     W_I = A_Ik T_k^T) and holds L_Ik itself afterwards (so it is directly the A operand of the backward product L_Ik^T x_I).
//
// Included by lba.hip (k_ldlt_reg) and by tools/microbench/ldlt_reg_bench.hip (the kernel alone against a host LDL^T).
#pragma once
#include <hip/hip_runtime.h>

namespace aos2 {

typedef double lr_double4_t __attribute__((ext_vector_type(4)));

#ifndef AOS2_LR_WORKERS
#define AOS2_LR_WORKERS 7
#endif
constexpr int kLrWorkers = AOS2_LR_WORKERS;        // worker waves (7 or 15); wave 0 factorises the diagonal blocks
constexpr int kLrThreads = 64 * (kLrWorkers + 1);
constexpr int kLrMaxNb = 15;                       // npad <= 240: 40 free keyframes
constexpr int kLrSlots = (kLrMaxNb * (kLrMaxNb - 1) / 2 + kLrWorkers - 1) / kLrWorkers;

// doubles of dynamic LDS the solve needs
__host__ __device__ inline size_t ldlt_reg_lds_doubles(int npad)
{
    const size_t nb = (size_t)(npad >> 4);
    return nb * 272 * 2 + (size_t)npad * 17 + 4 * (size_t)npad + 64 + 32;
}

// value of `v` in lane `src` (wave-uniform index), uniform result
__device__ __forceinline__ double readlane_f64(double v, int src)
{
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// 1 / d within 1 ulp: v_rcp_f64 + two Newton steps (34 cycles; the IEEE division takes 67)
__device__ __forceinline__ double lr_rcp(double d)
{
    double rd = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, rd, 1.0);
    rd = __builtin_fma(rd, e, rd);
    e = __builtin_fma(-d, rd, 1.0);
    return __builtin_fma(rd, e, rd);
}

// LDS writes of this wave visible to its other lanes (one wave: LDS operations execute in order; the fences only pin the compiler)
__device__ __forceinline__ void lr_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Solves Hs x = bs (Hs symmetric, both triangles stored, leading dimension ld; rows / columns >= n are an identity tail whether
// stored or not).  Called by all kLrThreads threads of the workgroup.  Returns false on a zero / NaN pivot; otherwise the solution
// is left in LDS at `xs_out` (npad doubles, the tail zero).  dbg (tid 0, kTiming): cycle counters of the phases.
template <bool kTiming>
__device__ __forceinline__ bool ldlt_reg_solve(const double *__restrict__ Hs, int ld, int n, int npad, const double *__restrict__ bs, double *sm,
                                               double *&xs_out, long long *dbg)
{
    constexpr int KW = kLrWorkers, NS = kLrSlots, NT = kLrThreads;
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, rq = lane >> 4;
    const int nb = npad >> 4;
    double *Dg = sm;                        // nb x (16 x 17): the diagonal tiles
    double *Tt = Dg + nb * 272;             // nb x (16 x 17): T_k^T (Tt_k[c * 17 + i] = T_k[i][c])
    double *Wb = Tt + nb * 272;             // npad x 17: L D of the current panel
    double *rv = Wb + (size_t)npad * 17;    // right-hand side under the forward substitution
    double *zv = rv + npad;                 // D^-1 L^-1 b
    double *rdv = zv + npad;                // 1 / D
    double *xs = rdv + npad;                // solution
    double *colbuf = xs + npad;             // 2 x 32: the pivot column on its way to all lanes
    double *sbuf = colbuf + 64;             // 16: a block's right-hand side in the backward pass
    double *part = Wb;                      // backward pass: KW x npad, worker w's share of sum_I L_IJ^T x_I (W is dead by then)
    const int ntot = nb * (nb - 1) / 2;
    const bool worker = wave > 0;
    const int widx = wave - 1;
    long long t_begin = 0, t_mark = 0, c_p = 0, c_u = 0, c_w = 0, c_d = 0;
    if (kTiming) t_begin = __builtin_amdgcn_s_memtime();

    // ---- load: the diagonal tiles and the right-hand side by everybody; the workers' tiles below (worker role)
    for (int idx = tid; idx < nb * 256; idx += NT) {   // (clamped addresses + a select: no branch, every load in flight at once)
        const int I = idx >> 8, a = (idx >> 4) & 15, b = idx & 15;
        const int r = 16 * I + a, c = 16 * I + b;
        const double v = Hs[(size_t)min(r, n - 1) * ld + min(c, n - 1)];
        Dg[I * 272 + a * 17 + b] = (r < n && c < n) ? v : (r == c ? 1.0 : 0.0);
    }
    for (int i = tid; i < npad; i += NT) rv[i] = i < n ? bs[i] : 0.0;
    if (tid == 0) s_fail = 0;
    long long t_loaded = 0, t_d0 = 0, t_fact = 0;

    // The two roles run the same sequence of workgroup barriers: [load] B [D_0] B { [P_k] B [U_k] B } ... B { [x_k] B [L^T x_k] B } B
    if (!worker) {
        // ================= wave 0: the diagonal blocks =================
        // D_k: unblocked LDL^T of the diagonal block with row i in lane i (lanes 32..47 mirror them), T_k = L_kk^-1 with column c in
        // lane 16 + c (48 + c mirrors) in the SAME instructions: eliminating column j is x[k] -= (x[j] / d_j) u_k for the rows
        // (u = column j = row j, by symmetry) and for the columns of T alike.  u travels through LDS (one write, broadcast
        // reads), the pivot by v_readlane; the next pivot's write, v_readlane and reciprocal are issued as soon as its column
        // entry is updated, ahead of the rest of this pivot's updates.  `cur` = the lane's entry of the right-hand side:
        // y_k = L_kk^-1 r_k on the way.  Leaves T_k^T in Tt, 1 / D in rdv, D^-1 y in zv.
        auto diag_block = [&](int kb, double cur) {
            const int k0 = kb << 4;
            const int li = col, isT = rq & 1;
            double *Dk = Dg + kb * 272, *Tk = Tt + kb * 272;
            double x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = Dk[li * 17 + c];
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = isT ? (c == li ? 1.0 : 0.0) : x[c];
            double *sp = (isT ? Tk : Dk) + li * 17;   // where the finished x[j] of this lane goes (the rows': scratch)
            bool bad = false;
            double xj = x[0];
            colbuf[isT * 16 + li] = xj;
            double dj = readlane_f64(xj, 0);
            double rd = lr_rcp(dj);
            double ck[2][16];
#pragma unroll
            for (int k = 1; k < 16; ++k) ck[0][k] = colbuf[k];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                bad |= (dj == 0.0) | (dj != dj);
                const double yj = readlane_f64(cur, j);   // = y_j: lane j's entry is final
                const double mult = xj * rd;
                sp[j] = xj;
                rdv[k0 + j] = rd;          // (uniform values: every lane stores the same)
                zv[k0 + j] = yj * rd;
                double xn = 0.0, djn = 1.0, rdn = 1.0;
                if (j < 15) {
                    x[j + 1] = __builtin_fma(-mult, ck[j & 1][j + 1], x[j + 1]);
                    xn = x[j + 1];
                    double *cbn = colbuf + ((j + 1) & 1) * 32;
                    cbn[isT * 16 + li] = xn;
#pragma unroll
                    for (int k = j + 2; k < 16; ++k) ck[(j + 1) & 1][k] = cbn[k];   // the next pivot's column: in flight behind this pivot's updates
                    djn = readlane_f64(xn, j + 1);
                    __builtin_amdgcn_sched_barrier(0);   // (the column's write and reads go out BEFORE this pivot's remaining updates)
                    rdn = lr_rcp(djn);
                }
#pragma unroll
                for (int k = j + 2; k < 16; ++k) x[k] = __builtin_fma(-mult, ck[j & 1][k], x[k]);
                cur = __builtin_fma(-mult, yj, cur);   // (unpredicated: the entries of lanes <= j are never read again)
                xj = xn;
                dj = djn;
                rd = rdn;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (bad && lane == 0) s_fail = 1;
        };
        __syncthreads();
        if (kTiming) t_loaded = __builtin_amdgcn_s_memtime();
        diag_block(0, rv[col]);
        __syncthreads();
        if (kTiming) t_d0 = t_mark = __builtin_amdgcn_s_memtime();
        for (int kb = 0; kb < nb - 1; ++kb) {
            if (s_fail) break;
            const int k0 = kb << 4;
            __syncthreads();   // (P_k: the workers)
            if (kTiming) {
                const long long tn = __builtin_amdgcn_s_memtime();
                c_p += tn - t_mark;
                t_mark = tn;
            }
            // U_k, this wave's share: the forward substitution of the next block's rows, the next diagonal tile's update,
            // then straight on to its factorisation (look-ahead) while the workers update the rest
            const int i = k0 + 16 + col;
            double ri = rv[i];
#pragma unroll
            for (int c = 0; c < 16; ++c) ri -= Wb[(size_t)i * 17 + c] * zv[k0 + c];
            double *Dn = Dg + (kb + 1) * 272;
            lr_double4_t acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = Dn[(rq + 4 * r) * 17 + col];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const double w = Wb[(size_t)i * 17 + rq + 4 * kk];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-(w * rdv[k0 + rq + 4 * kk]), w, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Dn[(rq + 4 * r) * 17 + col] = acc[r];
            lr_wave_sync();
            long long tq = 0;
            if (kTiming) tq = __builtin_amdgcn_s_memtime();
            diag_block(kb + 1, ri);
            if (kTiming) {
                const long long tn = __builtin_amdgcn_s_memtime();
                c_d += tn - tq;
                c_u += tn - t_mark;
                t_mark = tn;
            }
            __syncthreads();
            if (kTiming) {
                const long long tn = __builtin_amdgcn_s_memtime();
                c_w += tn - t_mark;
                t_mark = tn;
            }
        }
        if (kTiming) t_fact = __builtin_amdgcn_s_memtime();
        if (s_fail) return false;
        __syncthreads();   // (the workers clear their partial sums)
        // backward substitution L^T x = z, blocks from the bottom: x_k = T_k^T (z_k - sum_{I > k} L_Ik^T x_I)
        for (int kb = nb - 1; kb >= 0; --kb) {
            const int k0 = kb << 4;
            double sv = zv[k0 + col];
#pragma unroll
            for (int w = 0; w < KW; ++w) sv -= part[(size_t)w * npad + k0 + col];
            sbuf[col] = sv;
            lr_wave_sync();
            const double *Tk = Tt + kb * 272 + col * 17;
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                a0 = __builtin_fma(Tk[i], sbuf[i], a0);
                a1 = __builtin_fma(Tk[i + 1], sbuf[i + 1], a1);
                a2 = __builtin_fma(Tk[i + 2], sbuf[i + 2], a2);
                a3 = __builtin_fma(Tk[i + 3], sbuf[i + 3], a3);
            }
            xs[k0 + col] = (a0 + a1) + (a2 + a3);
            if (kb == 0) break;
            __syncthreads();
            __syncthreads();   // (L_kJ^T x_k: the workers)
        }
    } else {
        // ================= waves 1..KW: the tiles =================
        // rank t = a (a - 1) / 2 + b over 0 <= b < a < nb stands for the tile (I, J) = (nb - 1 - b, nb - 1 - a); slot s of this
        // worker holds rank s KW + widx.  Loaded transposed: register r of lane (col, rq) = A[I0 + col][J0 + rq + 4 r]
        // = Hs[J0 + rq + 4 r][I0 + col] (the matrix is symmetric and both triangles are stored: coalesced rows)
        lr_double4_t tile[NS];
        int sIJ[NS];   // I0 | J0 << 8 (uniform)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int t = s * KW + widx;
            int a = 1;
            while (a * (a + 1) / 2 <= t) ++a;
            const int b = t - a * (a - 1) / 2;
            sIJ[s] = t < ntot ? ((nb - 1 - b) << 4) | ((nb - 1 - a) << 12) : 0;
            tile[s] = lr_double4_t{0, 0, 0, 0};
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {   // (clamped addresses + a select: no branch, every load in flight at once)
            const int I0 = sIJ[s] & 255, J0 = sIJ[s] >> 8;
            const bool have = s * KW + widx < ntot;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = J0 + rq + 4 * r, c = I0 + col;
                const double v = Hs[(size_t)min(row, n - 1) * ld + min(c, n - 1)];
                tile[s][r] = (have && row < n && c < n) ? v : 0.0;
            }
        }
        long long w_t = 0, w_load = 0, w_p = 0, w_pb = 0, w_u = 0, w_ub = 0, w_bk = 0, w_bkb = 0;
        auto probe = [&](long long &acc) {
            if (kTiming) {
                const long long tn = __builtin_amdgcn_s_memtime();
                acc += tn - w_t;
                w_t = tn;
            }
        };
        if (kTiming) {
            w_t = t_begin;
            probe(w_load);
        }
        __syncthreads();
        __syncthreads();   // (D_0: wave 0)
        if (kTiming) w_t = __builtin_amdgcn_s_memtime();
        for (int kb = 0; kb < nb - 1; ++kb) {
            if (s_fail) break;
            const int k0 = kb << 4, m = nb - 1 - kb;
            const int lo = m * (m - 1) / 2, hi = lo + m;   // ranks of the trailing tiles: [0, lo); of this panel's tiles: [lo, hi)
            // ---- P_k: W_I = A_Ik T_k^T -- the transposed tile is the A operand as it stands --, L_Ik = W_I D^-1 stays in the tile
            {
                const double *Tk = Tt + kb * 272;
                double tb[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) tb[kk] = Tk[(rq + 4 * kk) * 17 + col];
                const double rdc = rdv[k0 + col];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int t = s * KW + widx;
                    if (t >= lo && t < hi) {
                        lr_double4_t acc = {0, 0, 0, 0};
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tile[s][kk], tb[kk], acc, 0, 0, 0);
                        const int I0 = sIJ[s] & 255;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            Wb[(size_t)(I0 + rq + 4 * r) * 17 + col] = acc[r];
                            tile[s][r] = acc[r] * rdc;
                        }
                    }
                }
            }
            probe(w_p);
            __syncthreads();
            probe(w_pb);
            // ---- U_k: A_IJ -= W_I L_J^T on the tiles still in the trailing matrix (register tiles, then this worker's diagonal tiles)
            double nr[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) nr[kk] = -rdv[k0 + rq + 4 * kk];
            // (a tile is four dependent MFMAs = 256 cycles of the wave's MFMA issue: the next tile's operands are requested before
            // them, so the LDS round trip is in their shadow)
            {
                const int cnt = lo > widx ? (lo - widx + KW - 1) / KW : 0;   // this worker's trailing tiles are its slots [0, cnt)
                double wi[4], wj[4];
                if (cnt > 0) {
                    const int I0 = sIJ[0] & 255, J0 = sIJ[0] >> 8;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        wi[kk] = Wb[(size_t)(I0 + col) * 17 + rq + 4 * kk];
                        wj[kk] = Wb[(size_t)(J0 + col) * 17 + rq + 4 * kk];
                    }
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (s < cnt) {
                        double ai[4], aj[4];
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            ai[kk] = wi[kk];
                            aj[kk] = wj[kk] * nr[kk];
                        }
                        if (s + 1 < NS && s + 1 < cnt) {
                            const int I0 = sIJ[s + 1 < NS ? s + 1 : s] & 255, J0 = sIJ[s + 1 < NS ? s + 1 : s] >> 8;
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk) {
                                wi[kk] = Wb[(size_t)(I0 + col) * 17 + rq + 4 * kk];
                                wj[kk] = Wb[(size_t)(J0 + col) * 17 + rq + 4 * kk];
                            }
                        }
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) tile[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(aj[kk], ai[kk], tile[s], 0, 0, 0);
                    }
                }
            }
            for (int I = kb + 2; I < nb; ++I) {
                if (I % KW != widx) continue;
                double *Dn = Dg + I * 272;
                lr_double4_t acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = Dn[(rq + 4 * r) * 17 + col];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double w = Wb[(size_t)((I << 4) + col) * 17 + rq + 4 * kk];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(w * nr[kk], w, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) Dn[(rq + 4 * r) * 17 + col] = acc[r];
            }
            // forward substitution of the rows below the next block: r_i -= sum_c W[i][c] z_c
            for (int i = k0 + 32 + tid - 64; i < npad; i += NT - 64) {
                double ri = rv[i];
#pragma unroll
                for (int c = 0; c < 16; ++c) ri -= Wb[(size_t)i * 17 + c] * zv[k0 + c];
                rv[i] = ri;
            }
            probe(w_u);
            __syncthreads();
            probe(w_ub);
        }
        if (s_fail) return false;
        // ---- backward substitution: the tile (I, J) holds L_IJ as the A operand of L_IJ^T x_I; every worker keeps its own share
        // of the sums (wave 0 adds them in worker order: fixed)
        for (int i = tid - 64; i < KW * npad; i += NT - 64) part[i] = 0.0;
        __syncthreads();
        if (kTiming) w_t = __builtin_amdgcn_s_memtime();
        for (int kb = nb - 1; kb > 0; --kb) {
            const int k0 = kb << 4;
            __syncthreads();   // (x_k: wave 0)
            probe(w_bkb);
            double xb[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) xb[kk] = xs[k0 + rq + 4 * kk];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s * KW + widx < ntot && (sIJ[s] & 255) == k0) {
                    double *pw = part + (size_t)widx * npad + (sIJ[s] >> 8);
                    lr_double4_t acc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = pw[rq + 4 * r];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tile[s][kk], xb[kk], acc, 0, 0, 0);
                    if (col == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) pw[rq + 4 * r] = acc[r];
                    }
                }
            }
            probe(w_bk);
            __syncthreads();
            probe(w_bkb);
        }
        if (kTiming && tid == 64) {
            dbg[8] = w_load; dbg[9] = w_p; dbg[10] = w_pb; dbg[11] = w_u; dbg[12] = w_ub; dbg[13] = w_bk; dbg[14] = w_bkb;
        }
    }
    __syncthreads();
    if (kTiming && tid == 0) {
        dbg[0] = t_loaded - t_begin;
        dbg[1] = t_d0 - t_loaded;
        dbg[2] = c_p;
        dbg[3] = c_u;
        dbg[4] = c_d;
        dbg[5] = t_fact - t_d0;
        dbg[6] = __builtin_amdgcn_s_memtime() - t_fact;
        dbg[7] = c_w;
    }
    xs_out = xs;
    return true;
}

}  // namespace aos2
