// Reduced camera system of one LocalBA window (the linear solve of g2o's BlockSolver::solve, block_solver.hpp:434-486 ->
// LinearSolverEigen, solvers/linear_solver_eigen.h:94-124; a dense LDL^T stands in for the sparse Cholesky):
// blocked right-looking LDL^T + both substitutions by ONE workgroup with the trailing matrix RESIDENT IN REGISTERS.
//
//   * the strictly lower 16x16 tiles (105 of them at npad = 240) live in the vector registers of kLrWorkers waves as MFMA
//     accumulator tiles for the whole factorisation (tile t -> worker t % kLrWorkers, slot t / kLrWorkers; tiles are ranked from the
//     bottom-right corner so that the tiles a panel still touches are always a PREFIX of the ranking: every worker has the same
//     share of every panel's update, and a slot is a compile-time register index);
//   * the diagonal tiles, the panel's L D (W), every panel's T_k = L_kk^-1, the tiles right below the diagonal (as L) and the
//     vectors live in LDS; nothing but the initial load and the solution touches device memory (the in-place global-memory form this
//     replaces paid a device-memory round trip per panel phase and per backward block: 113 us at 40 free keyframes);
//   * wave 0 factorises the diagonal blocks, the block spread over all 64 lanes (row i, columns 4q..4q+3 in lane i + 16 q; T the
//     same way by columns), so a pivot is 4 + 4 multiply-adds per lane and three LDS reads;
//   * a tile is stored TRANSPOSED while it belongs to the trailing matrix (so it is directly the A operand of the panel product
//     W_I = A_Ik T_k^T) and holds L_Ik itself afterwards (what the backward product L_Ik^T x_I reads).
//
// What bounds it (tools/microbench/lr_latency.hip, icache.hip, the event trace of ldlt_reg_bench): a wave issues one instruction
// every ~5 cycles whatever it is (f64 VALU 5.7, scalar / branch / LDS 4-5), a SIMD one v_mfma_f64_16x16x4 every 64 cycles, an LDS
// round trip is 64, a workgroup barrier 17.  So the code below counts INSTRUCTIONS on the chain D_k -> first panel tile -> next
// diagonal tile -> D_k+1: short compare chains over the slots, 2D lane layouts instead of 16-deep per-lane loops, no
// failure polls (a zero pivot turns the rest into NaNs; wave 0 reports it at the end).
//
// Included by lba.hip (k_ldlt_reg) and by tools/microbench/ldlt_reg_bench.hip (the kernel alone against a host LDL^T).
#pragma once
#include <hip/hip_runtime.h>

namespace aos2 {

typedef double lr_double4_t __attribute__((ext_vector_type(4)));

// Waves of the workgroup: wave 0 factorises the diagonal blocks, the others hold the tiles.  Waves w and w + 4 share a SIMD (measured:
// HW_ID of a workgroup's waves), and a worker's MFMA operand traffic on wave 0's SIMD stretched a diagonal block from 3.7 k to 5.0-5.7 k
// cycles: with AOS2_LR_SPARE0 the waves 4, 8, 12 only take part in the barriers, and wave 0 has its SIMD to itself -- measured
// (ldlt_reg_bench, 40 / 21 free keyframes): 16 waves 63.0 / 27.4 us, 16 waves with the spare ones 79.6 / 33.1 (three SIMDs' MFMA rate
// is the bound at 40, and 128 registers do not hold 9 tiles + the operands without spills), 8 waves 69.6 / 27.8: 16 waves, no spares.
#ifndef AOS2_LR_WAVES
#define AOS2_LR_WAVES 16
#endif
#ifndef AOS2_LR_SPARE0
#define AOS2_LR_SPARE0 0
#endif
constexpr int kLrWaves = AOS2_LR_WAVES;
constexpr bool kLrSpare0 = AOS2_LR_SPARE0 != 0;
constexpr int kLrWorkers = kLrSpare0 ? kLrWaves - kLrWaves / 4 : kLrWaves - 1;
constexpr int kLrThreads = 64 * kLrWaves;
constexpr int kLrMaxNb = 15;                       // npad <= 240: 40 free keyframes
constexpr int kLrSlots = (kLrMaxNb * (kLrMaxNb - 1) / 2 + kLrWorkers - 1) / kLrWorkers;
static_assert(kLrSlots <= 16, "the jump tables below list 16 slots");

// doubles of dynamic LDS the solve needs
__host__ __device__ constexpr size_t ldlt_reg_lds_doubles(int npad)
{
    return (size_t)(npad >> 4) * 272 * 3 + (size_t)npad * 17 + 4 * (size_t)npad + 256 + 160 + (size_t)kLrWorkers * 64;
}
constexpr size_t kLrMaxDynLds = ldlt_reg_lds_doubles(16 * kLrMaxNb) * sizeof(double);   // 149 248 B at 40 free keyframes
static_assert(kLrMaxDynLds + 64 <= 160 * 1024, "k_ldlt_reg: dynamic + static LDS beyond a CU's 160 KB");

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

// Workgroup barrier for waves that talk through LDS only: the wave's LDS traffic is waited for, its outstanding GLOBAL loads are not
// (__syncthreads() waits for vmcnt(0): the tiles' loads would have to land before wave 0 may start on the first diagonal block)
__device__ __forceinline__ void lr_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// AOS2_LR_TRACE (tools/microbench only): every wave logs (event, cycle) pairs behind the 16 counters of dbg -- a timeline of the barriers
#ifdef AOS2_LR_TRACE
#define LR_EV(id)                                                                                                   \
    do {                                                                                                            \
        if (lane == 0 && tr_n < AOS2_LR_TRACE) {                                                                    \
            dbg[16 + (size_t)wave * AOS2_LR_TRACE + tr_n] = (__builtin_amdgcn_s_memtime() << 12) | (unsigned)(id);  \
            ++tr_n;                                                                                                 \
        }                                                                                                           \
    } while (0)
#else
#define LR_EV(id)
#endif

// one branch per slot: `body(S)` with S a compile-time slot index (the register tile of slot S).  (An else-if chain, not a
// switch: with jump tables the register allocator spilled the tiles -- 298 spilled registers against none.)
#define LR_SLOT_CASE(sel, body, K) else if (NS > K && (sel) == K) { body((NS > K ? K : 0)); }
#define LR_SLOT_SWITCH(sel, body)                                                                                                   \
    if ((sel) == 0) { body(0); }                                                                                                    \
    LR_SLOT_CASE(sel, body, 1) LR_SLOT_CASE(sel, body, 2) LR_SLOT_CASE(sel, body, 3) LR_SLOT_CASE(sel, body, 4) LR_SLOT_CASE(sel, body, 5)   \
    LR_SLOT_CASE(sel, body, 6) LR_SLOT_CASE(sel, body, 7) LR_SLOT_CASE(sel, body, 8) LR_SLOT_CASE(sel, body, 9) LR_SLOT_CASE(sel, body, 10)  \
    LR_SLOT_CASE(sel, body, 11) LR_SLOT_CASE(sel, body, 12) LR_SLOT_CASE(sel, body, 13) LR_SLOT_CASE(sel, body, 14) LR_SLOT_CASE(sel, body, 15)

// Solves Hs x = bs.  Hs: symmetric, BOTH triangles stored, npad x npad with leading dimension npad, rows / columns >= n an identity
// tail (k_schur / k_prepare write exactly this).  Called by all kLrThreads threads of the workgroup.  Returns false on a zero / NaN
// pivot; otherwise the solution is left in LDS at `xs_out` (npad doubles, the tail zero).  dbg (kTiming): phase cycle counters.
//
// Barriers (every wave runs the same sequence): [wave 0: D_0 | workers: load] B { [first panel tile of every worker, the tile below the
// diagonal block before any other] B [wave 0: forward substitution of the next block's rows | workers: their other panel tiles] B
// [wave 0: next diagonal tile's update, D_k+1 | workers: U_k] B } [clear] B { [wave 0: x_k | workers: L_IJ^T x_I of the row above] B } B
template <bool kTiming>
__device__ __forceinline__ bool ldlt_reg_solve(const double *__restrict__ Hs, int n, int npad, const double *__restrict__ bs, double *sm, double *&xs_out,
                                               long long *dbg)
{
    constexpr int KW = kLrWorkers, NS = kLrSlots;
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, rq = lane >> 4;
    const int nb = npad >> 4, ld = npad;
    double *Dg = sm;                        // nb x (16 x 17): the diagonal tiles
    double *Tt = Dg + nb * 272;             // nb x (16 x 17): T_k^T (Tt_k[c * 17 + i] = T_k[i][c])
    double *Ls = Tt + nb * 272;             // nb x (16 x 17): L of the tile below diagonal block k (the backward pass's critical term)
    double *Wb = Ls + nb * 272;             // npad x 17: L D of the current panel
    double *rv = Wb + (size_t)npad * 17;    // right-hand side under the forward substitution
    double *rdv = rv + npad;                // 1 / D
    double *zv = rdv + npad;                // D^-1 L^-1 b
    double *xs = zv + npad;                 // solution
    double *cbuf = xs + npad;               // 2 x (64 + 64): the pivot column of A and the pivot row of T on their way to all lanes
    double *sbuf = cbuf + 256;              // 64 + 16 + 64: wave 0's cross-lane sums
    double *wred = sbuf + 160;              // KW x 64: a worker's cross-lane sums (backward pass)
    double *part = Wb;                      // backward pass: KW x npad, worker w's share of sum_I L_IJ^T x_I (W is dead by then)
    const int ntot = nb * (nb - 1) / 2;
    const bool worker = wave > 0 && !(kLrSpare0 && (wave & 3) == 0);
    const int widx = kLrSpare0 ? (wave >> 2) * 3 + (wave & 3) - 1 : wave - 1;   // dense index of a worker
    const int wt = widx * 64 + lane;                                            // ... and of its threads
    constexpr int WT = kLrWorkers * 64;
    long long t_begin = 0, t_mark = 0, c_p = 0, c_u = 0, c_w = 0, c_d = 0;
    if (kTiming) t_begin = __builtin_amdgcn_s_memtime();
    int tr_n = 0;
    (void)tr_n;
    LR_EV(1);
    long long t_loaded = 0, t_d0 = 0, t_fact = 0;

    if (wave == 0) {
        // ================= wave 0: the diagonal blocks =================
        __builtin_amdgcn_s_setprio(3);   // (its instructions are the critical path: ahead of the worker wave that shares the SIMD)
        // its own share of the load: the first diagonal tile and the first block of the right-hand side
        {
            double v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = Hs[(rq + 4 * r) * ld + col];
            const double b0 = col < n ? bs[col] : 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) Dg[(rq + 4 * r) * 17 + col] = v[r];
            rv[col] = b0;
            if (tid == 0) s_fail = 0;
            lr_wave_sync();
        }
        if (kTiming) t_loaded = t_mark = __builtin_amdgcn_s_memtime();
        // D_k: unblocked LDL^T of the diagonal block over all 64 lanes -- lane (i, q) = i + 16 q holds xa[r] = A[i][4q + r] and
        // tt[r] = T[4q + r][i] (T = L_kk^-1, by columns) -- so eliminating column j is four multiply-adds for the rows,
        // xa[r] -= (u_i / d_j) u_{4q+r}, and four for T, tt[r] -= u_{4q+r} (T[j][i] / d_j).  u = column j (= row j, by symmetry)
        // and row j of T travel through LDS: the lanes that hold them are the group q = j / 4 (every group writes, the readers
        // take that group's copy: no predicate); the pivot by v_readlane.  Nothing is predicated: entries of rows <= j are
        // scratch afterwards -- T[j][.] is final when pivot j publishes it, and every lane stores the copy it read to Tt.
        // Leaves T_k^T in Tt and 1 / D in rdv (the right-hand side: rhs_block).  The next pivot's entries are updated first, its write / reads / v_readlane / reciprocal go out ahead of
        // the rest of this pivot's updates.  (One copy of this code: the loop is entered with kb = -1 for D_0.)
        bool bad = false;
        const int i17 = col * 17;
        // y_k = T_k r_k and z_k = D^-1 y_k of a factorised block (while wave 0 would otherwise wait for the first panel tile): lane (i, q)
        // sums its four columns of T, the groups are added through LDS
        auto rhs_block = [&](int kd) {
            const int kd0 = kd << 4;
            const double *Tk = Tt + kd * 272 + col;
            double f = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) f = __builtin_fma(Tk[(4 * rq + r) * 17], rv[kd0 + 4 * rq + r], f);
            sbuf[lane] = f;
            const double rdi = rdv[kd0 + col];
            lr_wave_sync();
            zv[kd0 + col] = ((sbuf[col] + sbuf[16 + col]) + (sbuf[32 + col] + sbuf[48 + col])) * rdi;
            lr_wave_sync();
        };
        double *pub = cbuf + lane;   // this lane's slot of the published column / row
        const double *ui_p = cbuf + col, *u4_p = cbuf + 4 * rq;
        for (int kb = -1; kb < nb - 1; ++kb) {
            const int k0 = kb << 4;
            if (kb >= 0) {
                rhs_block(kb);
                LR_EV(16 * (kb + 1) + 4);
                lr_barrier();   // (every worker's first panel tile: W of the next block's rows is there)
                LR_EV(16 * (kb + 1) + 5);
                if (kTiming) {
                    const long long tn = __builtin_amdgcn_s_memtime();
                    c_p += tn - t_mark;
                    t_mark = tn;
                }
                // forward substitution of the next block's rows: lane (i, q) sums its four columns, the groups are added through LDS
                const double *wr = Wb + (size_t)(k0 + 16 + col) * 17;
                {
                    double f = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) f = __builtin_fma(wr[4 * rq + r], zv[k0 + 4 * rq + r], f);
                    sbuf[lane] = f;
                }
                double wv[4], rdk[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    wv[kk] = wr[rq + 4 * kk];
                    rdk[kk] = rdv[k0 + rq + 4 * kk];
                }
                double *Dn = Dg + (kb + 1) * 272;
                lr_double4_t acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = Dn[(rq + 4 * r) * 17 + col];
                lr_wave_sync();
                rv[k0 + 16 + col] = rv[k0 + 16 + col] - ((sbuf[col] + sbuf[16 + col]) + (sbuf[32 + col] + sbuf[48 + col]));
                LR_EV(16 * (kb + 1) + 6);
                lr_barrier();   // (the workers' other panel tiles)
                LR_EV(16 * (kb + 1) + 7);
                // the next diagonal tile's update, then straight on to its factorisation while the workers update the rest
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-(wv[kk] * rdk[kk]), wv[kk], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) Dn[(rq + 4 * r) * 17 + col] = acc[r];
                lr_wave_sync();
            }
            long long tq = 0;
            if (kTiming) tq = __builtin_amdgcn_s_memtime();
            LR_EV(16 * (kb + 1) + 8);
            {
                const int kd = kb + 1, kd0 = kd << 4;
                const double *Dk = Dg + kd * 272;
                double *Tk = Tt + kd * 272 + i17;
                double xa[4], tt[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    xa[r] = Dk[i17 + 4 * rq + r];
                    tt[r] = 4 * rq + r == col ? 1.0 : 0.0;
                }
                // pivot 0's column
                pub[0] = xa[0];
                pub[64] = tt[0];
                double dj = readlane_f64(xa[0], 0);
                double ui = ui_p[0], tj = ui_p[64];
                double u4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) u4[r] = u4_p[r];
                double rd = lr_rcp(dj);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    bad |= (dj == 0.0) | (dj != dj);
                    const double mult = ui * rd, st = tj * rd;
                    Tk[j] = tj;                               // T[j][i] (the four groups store the same)
                    rdv[kd0 + j] = rd;                        // (uniform value: every lane stores the same)
                    double djn = 1.0, rdn = 1.0, uin = 0.0, tjn = 0.0, u4n[4] = {0, 0, 0, 0};
                    if (j < 15) {
                        const int nq = (j + 1) >> 2, nr_ = (j + 1) & 3, parn = ((j + 1) & 1) * 128;
                        // the entries the next pivot publishes first ...
                        xa[nr_] = __builtin_fma(-mult, u4[nr_], xa[nr_]);
                        tt[nr_] = __builtin_fma(-u4[nr_], st, tt[nr_]);
                        pub[parn] = xa[nr_];
                        pub[parn + 64] = tt[nr_];
                        djn = readlane_f64(xa[nr_], (j + 1) + 16 * nq);
                        uin = ui_p[parn + 16 * nq];
                        tjn = ui_p[parn + 64 + 16 * nq];
#pragma unroll
                        for (int r = 0; r < 4; ++r) u4n[r] = u4_p[parn + 16 * nq + r];
                        __builtin_amdgcn_sched_barrier(0);   // (... its write and reads go out BEFORE this pivot's other updates)
                        rdn = lr_rcp(djn);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (r != nr_) {
                                xa[r] = __builtin_fma(-mult, u4[r], xa[r]);
                                tt[r] = __builtin_fma(-u4[r], st, tt[r]);
                            }
                    }
                    dj = djn;
                    rd = rdn;
                    ui = uin;
                    tj = tjn;
#pragma unroll
                    for (int r = 0; r < 4; ++r) u4[r] = u4n[r];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (kTiming) {
                const long long tn = __builtin_amdgcn_s_memtime();
                c_d += tn - tq;
                c_u += tn - t_mark;
                t_mark = tn;
            }
            LR_EV(16 * (kb + 1) + 9);
            lr_barrier();   // (D_k+1 is there; the workers have finished U_k -- or, the first time, the LDS-resident parts of the load)
            LR_EV(16 * (kb + 1) + 10);
            if (kTiming) {
                const long long tn = __builtin_amdgcn_s_memtime();
                c_w += tn - t_mark;
                t_mark = tn;
                if (kb < 0) t_d0 = tn;
            }
        }
        rhs_block(nb - 1);
        if (kTiming) t_fact = __builtin_amdgcn_s_memtime();
        if (bad && lane == 0) s_fail = 1;
        lr_barrier();   // (the workers clear their partial sums)
        // backward substitution L^T x = z, blocks from the bottom: x_k = T_k^T (z_k - sum_{I > k} L_Ik^T x_I): the term of the tile right
        // below the diagonal block here, from LDS (the one that depends on the block just solved); the others arrive as the workers'
        // sums, formed a step behind.  Both 16 x 16 products with lane (c, q) over rows 4q..4q+3, the groups added through LDS
        double *sx = sbuf + 64, *sb2 = sbuf + 80;
        for (int kb = nb - 1; kb >= 0; --kb) {
            const int k0 = kb << 4;
            double sv = zv[k0 + col];
#pragma unroll
            for (int w = 0; w < KW; ++w) sv -= part[(size_t)w * npad + k0 + col];
            if (kb < nb - 1) {
                const double *Lk = Ls + kb * 272 + col;
                double f = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) f = __builtin_fma(Lk[(4 * rq + r) * 17], xs[k0 + 16 + 4 * rq + r], f);
                sbuf[lane] = f;
                lr_wave_sync();
                sv -= (sbuf[col] + sbuf[16 + col]) + (sbuf[32 + col] + sbuf[48 + col]);
            }
            sx[col] = sv;   // (the four groups hold the same value)
            lr_wave_sync();
            const double *Tk = Tt + kb * 272 + i17;
            double g = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) g = __builtin_fma(Tk[4 * rq + r], sx[4 * rq + r], g);
            sb2[lane] = g;
            lr_wave_sync();
            xs[k0 + col] = (sb2[col] + sb2[16 + col]) + (sb2[32 + col] + sb2[48 + col]);
            lr_barrier();
        }
    } else if (!worker) {
        // ================= the spare waves on wave 0's SIMD: the barriers only =================
        lr_barrier();
        for (int kb = 0; kb < nb - 1; ++kb) {
            lr_barrier();
            lr_barrier();
            lr_barrier();
        }
        lr_barrier();
        for (int kb = nb - 1; kb >= 0; --kb) lr_barrier();
    } else {
        // ================= the workers: the tiles =================
        // rank t = a (a - 1) / 2 + b over 0 <= b < a < nb stands for the tile (I, J) = (nb - 1 - b, nb - 1 - a); slot s of this
        // worker holds rank s KW + widx.  Loaded transposed: register r of lane (col, rq) = A[I0 + col][J0 + rq + 4 r]
        // = Hs[J0 + rq + 4 r][I0 + col] (the matrix is symmetric and both triangles are stored: coalesced rows).
        // tab: lane s holds I0 | J0 << 8 of slot s (read with v_readlane: the table costs one register, not 15 scalar ones)
        lr_double4_t tile[NS];
        int tab;
        {
            const int t = lane * KW + widx;
            int a = (int)((1.0f + __fsqrt_rn(1.0f + 8.0f * (float)t)) * 0.5f);
            if (a * (a - 1) / 2 > t) --a;
            if (a * (a + 1) / 2 <= t) ++a;
            const int b = t - a * (a - 1) / 2;
            tab = (lane < NS && t < ntot) ? ((nb - 1 - b) << 4) | ((nb - 1 - a) << 12) : 0;
        }
        auto slot_ij = [&](int s) { return __builtin_amdgcn_readlane(tab, s); };
        {
            // the LDS-resident parts first (their loads are the first to land): diagonal tiles 1.., the rest of the right-hand side
            constexpr int ND = ((kLrMaxNb - 1) * 256 + WT - 1) / WT;
            double dv[ND], bv = 0.0;
#pragma unroll
            for (int u = 0; u < ND; ++u) {
                const int idx = min(wt + u * WT, (nb - 1) * 256 - 1 + (nb == 1)), I = 1 + (idx >> 8), a = (idx >> 4) & 15, b = idx & 15;
                dv[u] = Hs[(16 * min(I, nb - 1) + a) * ld + 16 * min(I, nb - 1) + b];
            }
            if (16 + wt < npad) bv = 16 + wt < n ? bs[16 + wt] : 0.0;
            const double *hl = Hs + rq * ld + col;
#pragma unroll
            for (int s = NS - 1; s >= 0; --s) {   // (no branch: every load in flight at once; a slot beyond the last tile reads tile (0, 0), unused)
                const int ij = slot_ij(s), off = (ij >> 8) * ld + (ij & 255);
#pragma unroll
                for (int r = 0; r < 4; ++r) tile[s][r] = hl[off + 4 * r * ld];
            }
#pragma unroll
            for (int u = 0; u < ND; ++u) {
                const int idx = wt + u * WT, I = 1 + (idx >> 8), a = (idx >> 4) & 15, b = idx & 15;
                if (idx < (nb - 1) * 256) Dg[I * 272 + a * 17 + b] = dv[u];
            }
            if (16 + wt < npad) rv[16 + wt] = bv;
        }
        LR_EV(9);
        lr_barrier();   // (D_0: wave 0)
        LR_EV(10);
        for (int kb = 0; kb < nb - 1; ++kb) {
            const int k0 = kb << 4, m = nb - 1 - kb;
            const int lo = m * (m - 1) / 2, hi = lo + m;   // ranks of the trailing tiles: [0, lo); of this panel's tiles: [lo, hi)
            // ---- P_k: W_I = A_Ik T_k^T -- the transposed tile is the A operand as it stands --, L_Ik = W_I D^-1 stays in the tile.
            // Every worker's HIGHEST panel tile first (rank hi - 1 = the tile right below the diagonal block: wave 0 waits for it alone)
            const double *Tk = Tt + kb * 272;
            double tb[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) tb[kk] = Tk[(rq + 4 * kk) * 17 + col];
            const double rdc = rdv[k0 + col];
            int sp = hi - 1 >= widx ? (hi - 1 - widx) / KW : -1;   // the worker's highest slot with a rank < hi (it may still be below lo)
            auto panel_tile = [&](lr_double4_t &tl, int I0, bool below_diag) {
                lr_double4_t acc = {0, 0, 0, 0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tl[kk], tb[kk], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Wb[(size_t)(I0 + rq + 4 * r) * 17 + col] = acc[r];
                    tl[r] = acc[r] * rdc;
                }
                if (below_diag) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Ls[kb * 272 + (rq + 4 * r) * 17 + col] = tl[r];
                }
            };
#define LR_PANEL(S) panel_tile(tile[S], slot_ij(S) & 255, (S) * KW + widx == hi - 1)
            if (sp >= 0 && sp * KW + widx >= lo) {
                LR_SLOT_SWITCH(sp, LR_PANEL)
                --sp;
            } else
                sp = -1;
            LR_EV(16 * (kb + 1) + 4);
            lr_barrier();
            LR_EV(16 * (kb + 1) + 5);
            for (; sp >= 0 && sp * KW + widx >= lo; --sp) { LR_SLOT_SWITCH(sp, LR_PANEL) }
#undef LR_PANEL
            LR_EV(16 * (kb + 1) + 6);
            lr_barrier();
            LR_EV(16 * (kb + 1) + 7);
            // ---- U_k: A_IJ -= W_I L_J^T on the tiles still in the trailing matrix (register tiles, then this worker's diagonal tiles).
            // A tile is four dependent MFMAs = 256 cycles of the wave's MFMA issue: the next tile's operands are requested before them,
            // so the LDS round trip is in their shadow; the slots [0, cnt) from the top down
            double nr[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) nr[kk] = -rdv[k0 + rq + 4 * kk];
            {
                const int cnt = lo > widx ? (lo - widx + KW - 1) / KW : 0;
                const double *wl = Wb + col * 17 + rq;
                double wi[4] = {0, 0, 0, 0}, wj[4] = {0, 0, 0, 0};
                if (cnt > 0) {
                    const int ij = slot_ij(cnt - 1);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        wi[kk] = wl[(ij & 255) * 17 + 4 * kk];
                        wj[kk] = wl[(ij >> 8) * 17 + 4 * kk];
                    }
                }
#define LR_UPD(S)                                                                                                                          \
    {                                                                                                                                      \
        double ai[4], aj[4];                                                                                                               \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                                                   \
        {                                                                                                                                  \
            ai[kk] = wi[kk];                                                                                                               \
            aj[kk] = wj[kk] * nr[kk];                                                                                                      \
        }                                                                                                                                  \
        if ((S) > 0) {                                                                                                                     \
            const int ij = slot_ij((S) > 0 ? (S)-1 : 0);                                                                                   \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                                               \
            {                                                                                                                              \
                wi[kk] = wl[(ij & 255) * 17 + 4 * kk];                                                                                     \
                wj[kk] = wl[(ij >> 8) * 17 + 4 * kk];                                                                                      \
            }                                                                                                                              \
        }                                                                                                                                  \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) tile[S] = __builtin_amdgcn_mfma_f64_16x16x4f64(aj[kk], ai[kk], tile[S], 0, 0, 0); \
    }
                // (slot cnt - 1 down to slot 0; a compare and a branch per slot)
#define LR_UPD_IF(K) if (NS > K && cnt > K) LR_UPD((NS > K ? K : 0))
                LR_UPD_IF(15) LR_UPD_IF(14) LR_UPD_IF(13) LR_UPD_IF(12) LR_UPD_IF(11) LR_UPD_IF(10) LR_UPD_IF(9) LR_UPD_IF(8)
                LR_UPD_IF(7) LR_UPD_IF(6) LR_UPD_IF(5) LR_UPD_IF(4) LR_UPD_IF(3) LR_UPD_IF(2) LR_UPD_IF(1) LR_UPD_IF(0)
#undef LR_UPD_IF
#undef LR_UPD
            }
            {   // this worker's diagonal tiles: I = kb + 2 + ((widx - kb - 2) mod KW), then every KW-th
                int I = kb + 2 + (widx + KW * 4 - (kb + 2) % KW) % KW;
                for (; I < nb; I += KW) {
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
            }
            {   // forward substitution of the rows below the next block: r_i -= sum_c W[i][c] z_c (four partial sums)
                const int i = k0 + 32 + wt;
                if (i < npad) {
                    const double *wr = Wb + (size_t)i * 17;
                    double f0 = rv[i], f1 = 0, f2 = 0, f3 = 0;
#pragma unroll
                    for (int c = 0; c < 16; c += 4) {
                        f0 = __builtin_fma(-wr[c], zv[k0 + c], f0);
                        f1 = __builtin_fma(-wr[c + 1], zv[k0 + c + 1], f1);
                        f2 = __builtin_fma(-wr[c + 2], zv[k0 + c + 2], f2);
                        f3 = __builtin_fma(-wr[c + 3], zv[k0 + c + 3], f3);
                    }
                    rv[i] = (f0 + f1) + (f2 + f3);
                }
            }
            LR_EV(16 * (kb + 1) + 9);
            lr_barrier();
            LR_EV(16 * (kb + 1) + 10);
        }
        // ---- backward substitution: the tile (I, J) holds L_IJ; every worker keeps its own share of the sums sum_I L_IJ^T x_I (wave 0
        // adds them in worker order: fixed).  In step kb the workers form the terms of block row kb + 1 (x of that row was published a
        // step ago) except its tile right below the diagonal, which wave 0 applies itself from LDS.  A term: lane (c, q) multiplies
        // its four rows, the groups are added through LDS
        for (int i = wt; i < KW * npad; i += WT) part[i] = 0.0;
        lr_barrier();
        double *myred = wred + widx * 64, *mypart = part + (size_t)widx * npad + col;
        for (int kb = nb - 1; kb >= 0; --kb) {
            const int k0 = kb << 4;
            if (kb < nb - 1 && kb > 0) {
                const int ent = tab;
                unsigned long long todo = __ballot(lane < NS && lane * KW + widx < ntot && (ent & 255) == k0 + 16 && (ent >> 8) != k0);
                if (todo) {
                    double xb[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) xb[r] = xs[k0 + 16 + rq + 4 * r];
                    while (todo) {
                        const int s = __builtin_ctzll(todo);
                        todo &= todo - 1;
                        double p = 0;
#define LR_BACK(S) p = __builtin_fma(tile[S][3], xb[3], __builtin_fma(tile[S][2], xb[2], __builtin_fma(tile[S][1], xb[1], tile[S][0] * xb[0])))
                        LR_SLOT_SWITCH(s, LR_BACK)
#undef LR_BACK
                        myred[lane] = p;
                        lr_wave_sync();
                        double *pw = mypart + (slot_ij(s) >> 8);
                        const double sum = (myred[col] + myred[16 + col]) + (myred[32 + col] + myred[48 + col]);
                        *pw = *pw + sum;   // (the four groups store the same value)
                        lr_wave_sync();
                    }
                }
            }
            lr_barrier();
        }
    }
    lr_barrier();
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
    return s_fail == 0;
}

}  // namespace aos2
