// HIP kernels of the ORB extractor for gfx950 (wave64).  Integer/byte work, HBM/LDS bound:
// no MFMA here by design.  Reference semantics: src/ORBextractor.cc (see each kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cmath>

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "written for gfx950 (MI355X) only: global_load_lds_dwordx4, v_pk_maximum3_f16, v_permlane{16,32}_swap (Makefile: ARCH = gfx950)"
#endif

#include "extractor_kernels.h"
#include "octree.h"
#include "sincos_exact.h"
#include "wave_ops.h"

namespace aos2 {

__constant__ __attribute__((aligned(16))) int8_t c_pattern[1024];   // rBRIEF test locations (src/ORBextractor.cc:150-408)
__constant__ int c_umax[16];           // circular patch row extents (:454-470)
__constant__ int c_gauss[8];           // 7-tap Gaussian, 8 fractional bits {18,34,49,55,49,34,18}
// IC_Angle: byte masks of the circular patch, [31 rows v = -15..15][9 dwords covering patch columns 4..39]:
// byte k of dword dj (column 4*dj + k, u = column - 21) is inside iff |u| <= umax[|v|] (:88-101)
__constant__ uint32_t c_icmask[31 * 9 + 1];

// describe_kernel's horizontal blur works on items (row pair m, output quad j) of the 43 x 37 tile; only the items a
// rotated pattern point can reach (disc of radius max|pattern| + rounding, +-3 rows of vertical taps) are listed:
// 189 of 220 for the ORB pattern, i.e. 3 rounds of 64 lanes instead of 4.  Entry = m << 8 | j, 0xffff = none.
__constant__ uint16_t c_hitems[256];
__constant__ int c_hrounds;

int upload_constants(const int8_t *pattern, const int *umax, const int *gauss7, hipStream_t st)
{
    {
        double rmax = 0;
        for (int i = 0; i < 512; ++i) rmax = std::max(rmax, std::hypot((double)pattern[2 * i], (double)pattern[2 * i + 1]));
        // a sample is (round(x'), round(y')) of a point at distance <= rmax * |(cos, sin)|_float from the centre
        const double R = rmax * (1.0 + 1e-6) + 0.70711, R2 = R * R;
        bool need[22][10] = {};
        for (int ix = -18; ix <= 18; ++ix)
            for (int iy = -18; iy <= 18; ++iy) {
                if ((double)(ix * ix + iy * iy) > R2) continue;
                for (int dy = iy - 3; dy <= iy + 3; ++dy) need[(dy + 21) / 2][(ix + 18) / 4] = true;
            }
        uint16_t items[256];
        int n = 0;
        for (int m = 0; m < 22; ++m)
            for (int j = 0; j < 10; ++j)
                if (need[m][j]) items[n++] = (uint16_t)(m << 8 | j);
        const int rounds = (n + 63) / 64;
        for (; n < 256; ++n) items[n] = 0xffff;
        hipError_t eh = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_hitems), items, sizeof(items), 0, hipMemcpyHostToDevice, st);
        if (eh != hipSuccess) return (int)eh;
        eh = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_hrounds), &rounds, sizeof(int), 0, hipMemcpyHostToDevice, st);
        if (eh != hipSuccess) return (int)eh;
    }
    hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_pattern), pattern, 1024, 0, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_umax), umax, 16 * sizeof(int), 0, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return (int)e;
    int g[8] = {gauss7[0], gauss7[1], gauss7[2], gauss7[3], gauss7[4], gauss7[5], gauss7[6], 0};
    e = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_gauss), g, sizeof(g), 0, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return (int)e;
    uint32_t icm[31 * 9 + 1] = {};
    for (int vr = 0; vr < 31; ++vr) {
        const int v = vr - 15, um = umax[v < 0 ? -v : v];
        for (int dj = 1; dj <= 9; ++dj) {
            uint32_t m = 0;
            for (int k = 0; k < 4; ++k) {
                const int u = 4 * dj + k - 21;
                if ((u < 0 ? -u : u) <= um) m |= 0xffu << (8 * k);
            }
            icm[vr * 9 + dj - 1] = m;
        }
    }
    e = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_icmask), icm, sizeof(icm), 0, hipMemcpyHostToDevice, st);
    return (int)e;
}

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

__device__ __forceinline__ uint32_t udot2_u16(uint32_t a, uint32_t b, uint32_t c)
{
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b), c, false);
}

// ---------------------------------------------------------------------------------------------
// Level 0 of the pyramid (ComputePyramid, src/ORBextractor.cc:1107-1132) is the caller's image
// itself: nothing is copied.  The 19-px REFLECT_101 frame the reference keeps around each level is
// never read by the extractor and is synthesised on demand for mvImagePyramid readers
// (aos2_extractor_pyramid_level).
// ---------------------------------------------------------------------------------------------
// cv::resize(INTER_LINEAR) 8UC1, level L-1 -> L (src/ORBextractor.cc:1120).  Fixed point,
// 11-bit coefficients; the coefficient tables are built on the host exactly like OpenCV's
// resize() does (float fx, two separately rounded shorts), so host and device cannot disagree.
//
// The kernel is integer-VALU bound (4 cycles per wave64 instruction for everything but add / logic / shift,
// tools/microbench/valu_rate.hip), so it is organised around the instruction count per output pixel:
//   * one lane = 4 horizontally adjacent output pixels (one aligned 32-bit store) x RS_BAND output rows;
//   * the <= 8 source bytes the 4 outputs of a row need (scale <= 2) are ONE unaligned 64-bit load straight
//     from L2 -- no LDS staging; all source rows of the band are requested before the first is used;
//   * horizontal pass per source row: v_perm_b32 (pair of neighbours -> u16x2) + v_dot2_u32_u16 with the packed
//     (a0, a1) table entry, 2 instructions per value, and each source row is evaluated once although two output
//     rows use it; vertical pass: two v_mul_hi_u32 per value ((b << 16) * (r >> 4) >> 32 == (b * (r >> 4)) >> 16);
//   * every lane of a wave works on the SAME band of rows (items are flattened over (image, column quad)), so
//     the row bookkeeping is scalar and a level's ragged right edge costs no idle lanes.
// ---------------------------------------------------------------------------------------------
constexpr int RS_BAND = 8;    // output rows per lane (4 when the level's vertical ratio exceeds 2: see launch_resize)
constexpr int RS_NR = 16;     // source rows of a band kept in registers: ceil((band - 1) * ratio) + 2 <= 16

// WIDE = false: one 8-byte window holds sx_0 .. sx_3 + 1 (horizontal ratio <= 2.0); WIDE = true: one window per output
// PAIR (any ratio <= 6).  A level's ratio src / dst can exceed the extractor's scale factor slightly because the level
// sizes are rounded (cvRound(165 / 2.0) = 82 -> 2.012), so scaleFactor = 2.0 needs the wide form and 4-row bands.
template <bool WIDE>
__global__ __launch_bounds__(256) void resize_level_kernel(const uint8_t *__restrict__ src_base,
                                                           size_t src_img_stride, int src_pitch,
                                                           uint8_t *__restrict__ pyr, size_t pyr_stride,
                                                           LevelDev src, LevelDev dst,
                                                           const int *__restrict__ xofs,
                                                           const int *__restrict__ xab,
                                                           const int *__restrict__ yofs,
                                                           const int *__restrict__ yab,
                                                           int batch, int nquads, uint32_t inv_nquads, int band)
{
    const int lane = threadIdx.x & 63;
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    const int dy0 = blockIdx.y * band;
    const int nout = min(band, dst.h - dy0);   // >= 1
    // lane l < nout keeps the y-table entries of output row dy0 + l (read back with v_readlane)
    int ty = 0, tc = 0;
    if (lane < nout) {
        ty = yofs[dst.tab_y + dy0 + lane];
        tc = yab[dst.tab_y + dy0 + lane];
    }
    if (id >= (uint32_t)nquads * (uint32_t)batch) return;
    const uint32_t img = __umulhi(id, inv_nquads);   // id / nquads, exact for id < 2^32 / nquads
    const int dx0 = 4 * (int)(id - img * (uint32_t)nquads);
    // x tables are padded to a multiple of 4 entries per level
    const int4 xo = *reinterpret_cast<const int4 *>(xofs + dst.tab_x + dx0);
    const int4 xa = *reinterpret_cast<const int4 *>(xab + dst.tab_x + dx0);
    const int sxs[4] = {xo.x, xo.y, xo.z, xo.w};
    const uint32_t aas[4] = {(uint32_t)xa.x, (uint32_t)xa.y, (uint32_t)xa.z, (uint32_t)xa.w};
    // 8-byte source window [start, start + 8) holds sx_k and sx_k + 1 of its outputs; at the right edge the window is
    // pulled inside the row and the neighbour selector repeats sx (its weight a1 is 0 there)
    constexpr int NW = WIDE ? 2 : 1;
    int start[NW];
#pragma unroll
    for (int v = 0; v < NW; ++v) start[v] = max(min(sxs[2 * v], src.w - 8), 0);
    uint32_t sel[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t lo = (uint32_t)(sxs[k] - start[WIDE ? k >> 1 : 0]), hi = sxs[k] + 1 < src.w ? lo + 1 : lo;
        sel[k] = lo | (hi << 16) | 0x0c000c00u;   // v_perm_b32: byte 0 <- window[lo], byte 2 <- window[hi], bytes 1,3 <- 0
    }
    const uint8_t *sp = src_base + (size_t)img * src_img_stride;
    uint8_t *dp = pyr + (size_t)img * pyr_stride + dst.off + dx0;
    const uint32_t keep = dx0 + 4 <= dst.w ? 0xffffffffu : (1u << (8 * (dst.w - dx0))) - 1u;   // padding columns stay 0
    // source rows of the band: every row of [r_first, r_last] is used
    const int hmax = src.h - 1;
    const int t_first = __builtin_amdgcn_readlane(ty, 0), t_last = __builtin_amdgcn_readlane(ty, nout - 1);
    const int r_first = min(max(t_first, 0), hmax);
    const int nrows = min(min(max(t_last + 1, 0), hmax) - r_first + 1, RS_NR);
    uint2 win[NW][RS_NR];
#pragma unroll
    for (int c = 0; c < RS_NR / 4; ++c) {
        if (4 * c < nrows) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int r = min(r_first + 4 * c + t, hmax);
#pragma unroll
                for (int v = 0; v < NW; ++v)
                    __builtin_memcpy(&win[v][4 * c + t], sp + (size_t)((uint32_t)r * (uint32_t)src_pitch) + start[v], 8);
            }
        }
    }
    uint32_t prev[4] = {0u, 0u, 0u, 0u}, cur[4] = {0u, 0u, 0u, 0u};
    int j = 0;
    int sy0 = 0, sy1 = min(max(t_first + 1, 0), hmax) - r_first, coef = __builtin_amdgcn_readlane(tc, 0);
#pragma unroll
    for (int i = 0; i < RS_NR; ++i) {
        if (i >= nrows) continue;   // (uniform) not `break`: the loop must unroll completely, win[] is a register array
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint2 wk = win[WIDE ? k >> 1 : 0][i];
            prev[k] = cur[k];
            cur[k] = udot2_u16(__builtin_amdgcn_perm(wk.y, wk.x, sel[k]), aas[k], 0u) >> 4;
        }
        while (j < nout && sy1 == i) {
            const uint32_t b0s = (uint32_t)coef << 16, b1s = (uint32_t)coef & 0xffff0000u;
            const bool same = sy0 == i;   // bottom edge: both rows clamp to the last one
            uint32_t out = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t v = (__umulhi(b0s, same ? cur[k] : prev[k]) + __umulhi(b1s, cur[k]) + 2u) >> 2;   // <= 255
                out |= v << (8 * k);
            }
            *reinterpret_cast<uint32_t *>(dp + (uint32_t)(dy0 + j) * (uint32_t)dst.pitch) = out & keep;
            ++j;
            if (j < nout) {
                const int t = __builtin_amdgcn_readlane(ty, j);
                sy0 = min(max(t, 0), hmax) - r_first;
                sy1 = min(max(t + 1, 0), hmax) - r_first;
                coef = __builtin_amdgcn_readlane(tc, j);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// FAST-9/16 score + 3x3 non-max suppression + threshold fallback, one workgroup per
// (grid cell, image).  cv::FAST(cell, kps, iniThFAST, true) with the minThFAST retry of
// src/ORBextractor.cc:809-816, cell geometry of :775-806.
//
// Closed form used here (equivalent to OpenCV's cornerScore<16>, DESIGN.md "FAST score"):
//   with ring pixels x[0..15] around centre v,
//     A = v - min_s max(x[s..s+8]),  B' = max_s min(x[s..s+8]) - v,   S = max(A, B') - 1
//   pixel is a corner at threshold t  <=>  S >= t,  and its score is S for every t <= S.
// NMS only looks at neighbours inside the same cell's evaluated region (scores outside are 0),
// so no halo beyond the 3-px ring is needed and cells are independent.
// The tile (cell + 3-px ring halo) is staged in LDS with coalesced 32-bit loads.
// ---------------------------------------------------------------------------------------------
// gfx950's packed 3-input IEEE maximum / minimum on f16 pairs, used on integers: a byte v is carried as the f16
// bit pattern 0x0400 | v (exponent 1, mantissa v: a positive normal number, monotonic in v), so maximum3 / minimum3
// of such patterns are the patterns of the integer max / min (checked exhaustively by tools/microbench/valu_rate.hip).
__device__ __forceinline__ uint32_t pk_max3_h(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pk_min3_h(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ int fast_score_full(const uint8_t *t, int TP)
{
    // t points at the centre pixel inside the LDS tile
    int x[16];
    x[0] = t[3 * TP];      x[1] = t[3 * TP + 1];   x[2] = t[2 * TP + 2];   x[3] = t[TP + 3];
    x[4] = t[3];           x[5] = t[-TP + 3];      x[6] = t[-2 * TP + 2];  x[7] = t[-3 * TP + 1];
    x[8] = t[-3 * TP];     x[9] = t[-3 * TP - 1];  x[10] = t[-2 * TP - 2]; x[11] = t[-TP - 3];
    x[12] = t[-3];         x[13] = t[TP - 3];      x[14] = t[2 * TP - 2];  x[15] = t[3 * TP - 1];
    // Both polarities in one chain: low half carries x, high half 255 - x, so a packed max over an arc gives
    // (max x, 255 - min x).  One v_mad_i32_i24 per ring pixel builds the pair: x * (1 - 65536) + 0x04FF0400.
    uint32_t P[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) P[i] = (uint32_t)(__mul24(x[i], -65535) + 0x04FF0400);
    uint32_t M3[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) M3[i] = pk_max3_h(P[i], P[(i + 1) & 15], P[(i + 2) & 15]);
    uint32_t M9[16];   // arc i .. i+8
#pragma unroll
    for (int i = 0; i < 16; ++i) M9[i] = pk_max3_h(M3[i], M3[(i + 3) & 15], M3[(i + 6) & 15]);
    const uint32_t a0 = pk_min3_h(M9[0], M9[1], M9[2]), a1 = pk_min3_h(M9[3], M9[4], M9[5]), a2 = pk_min3_h(M9[6], M9[7], M9[8]);
    const uint32_t a3 = pk_min3_h(M9[9], M9[10], M9[11]), a4 = pk_min3_h(M9[12], M9[13], M9[14]);
    const uint32_t b0 = pk_min3_h(a0, a1, a2), b1 = pk_min3_h(a3, a4, M9[15]);
    const uint32_t R = pk_min3_h(b0, b1, b1);
    const int minmax = (int)(R & 0xffu);                  // min over arcs of the arc's max
    const int maxmin = 255 - (int)((R >> 16) & 0xffu);    // max over arcs of the arc's min
    const int v = t[0];
    return max(v - minmax, maxmin - v) - 1;
}

// number of set bits of a ballot below this lane: v_mbcnt_lo + v_mbcnt_hi (instead of masking with (1 << lane) - 1 and
// two v_bcnt)
__device__ __forceinline__ int lanes_below(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2_t as_us2(uint32_t x) { return __builtin_bit_cast(us2_t, x); }
__device__ __forceinline__ uint32_t as_u32(us2_t x) { return __builtin_bit_cast(uint32_t, x); }

// compass pre-test on two pixels at once (packed u16 lanes): non-zero lane <=> one pixel of EACH antipodal compass pair -- (a, b) =
// ring pixels 0 / 8, (c, d) = ring pixels 4 / 12 -- is brighter than v+t, or one of each pair is darker than v-t.  A contiguous arc of 9 of
// the 16 ring pixels holds ring pixel i or i + 8 for every i, so every FAST-9 corner passes; pixels with two bright compass pixels of
// the SAME pair (no 9-arc can hold both without one of the other pair) do not -- tighter than "two of the four" and two instructions
// shorter (round 3; the survivors are scored exactly either way, so the result is the same).
__device__ __forceinline__ uint32_t compass2(us2_t v, us2_t a, us2_t b, us2_t c, us2_t d, us2_t T)
{
    const us2_t h1 = __builtin_elementwise_max(a, b), l1 = __builtin_elementwise_min(a, b);
    const us2_t h2 = __builtin_elementwise_max(c, d), l2 = __builtin_elementwise_min(c, d);
    const us2_t hi = __builtin_elementwise_min(h1, h2);   // the smaller of the pairs' maxima: both pairs have a pixel above it or equal
    const us2_t lo = __builtin_elementwise_max(l1, l2);
    // hi > v + T  <=>  (hi -sat T) > v: no sum that could leave 16 bits, so the same code serves operands that are
    // scaled by 256 (the odd bytes of a dword taken with one AND instead of shift + AND; T scaled alike)
    const us2_t bright = __builtin_elementwise_sub_sat(__builtin_elementwise_sub_sat(hi, T), v);
    const us2_t dark = __builtin_elementwise_sub_sat(__builtin_elementwise_sub_sat(v, T), lo);
    return as_u32(bright) | as_u32(dark);
}

// The same test as a MARGIN: max(hi -sat v, v -sat lo) per 16-bit lane -- three packed operations (two saturating differences and their
// maximum) instead of four saturating subtractions and an OR; a lane of the result is > T exactly where compass2's lane is non-zero
// (hi - T > v <=> hi - v > T in integers; a difference that saturates to 0 is false on both sides), also for operands scaled by 256
// with a fraction below them (T scaled alike).  Round 6: 0.489 -> 0.478 ms per 512 frames of the profiling batch with the even pixels
// alone in this form, same candidates; -DAOS2_FAST_NO_MARGIN keeps compass2.
__device__ __forceinline__ uint32_t compass2_margin(us2_t v, us2_t a, us2_t b, us2_t c, us2_t d)
{
    const us2_t h1 = __builtin_elementwise_max(a, b), l1 = __builtin_elementwise_min(a, b);
    const us2_t h2 = __builtin_elementwise_max(c, d), l2 = __builtin_elementwise_min(c, d);
    const us2_t hi = __builtin_elementwise_min(h1, h2), lo = __builtin_elementwise_max(l1, l2);
    return as_u32(__builtin_elementwise_max(__builtin_elementwise_sub_sat(hi, v), __builtin_elementwise_sub_sat(v, lo)));
}

// (AOS2_FAST_ABL = 1..4, AOS2_DESC_ABL = 1..4: timing-only ablation builds of tools/build_abl_libs.sh -- the kernel stops after /
// skips one phase, results are wrong by construction; DESIGN.md section 0, item 6 has the phase shares they gave.)
// Phases per wave (one grid cell of one image, 64-thread workgroup = one wave, so list counters are
// wave-uniform registers and no LDS atomics or multi-wave barriers are needed):
//   0. stage the cell + ring halo in LDS (32-bit loads), evaluated column 0 on a dword boundary
//   1. compass pre-test (one pixel of each antipodal compass pair) on every pixel, 4 horizontally adjacent pixels per lane from 5 LDS dwords,
//      two pixels per packed-u16 op; survivors are ballot-compacted into an LDS list
//   2. exact score for the survivors (dense lanes); corners (S >= th) go to the score map
//   3. NMS of the corners against the score map; kept ones to the cell's slots.  Every list is built in row-major
//      order (= cv::FAST's emission order) and every compaction is stable, so nothing is sorted
//   4. empty after NMS and th == iniThFAST -> repeat 1-3 with minThFAST (:812-816)
__global__ __launch_bounds__(64) void fast_cells_kernel(const uint8_t *__restrict__ img0,
                                                        size_t img0_stride, int pitch0,
                                                        const uint8_t *__restrict__ pyr,
                                                        size_t pyr_stride,
                                                        const CellDev *__restrict__ cells,
                                                        int n_cells, int ini_th, int min_th,
                                                        int TP, int TH, int SP,
                                                        uint32_t *__restrict__ slots,
                                                        size_t slot_stride,
                                                        int32_t *__restrict__ cell_cnt,
                                                        int list_cap, int keep_cap, int batch)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tile_bytes = (TP * TH + 15) & ~15;
    const int smap_bytes = (SP * (TH - 4) + 15) & ~15;  // (ch+2) rows
    uint8_t *tile = smem;                                      // TH x TP
    uint8_t *smap = smem + tile_bytes;                         // (ch+2) x SP, 1-px zero border
    // survivors (py<<6 | px): list_cap entries; a cell that produces more is scored in instalments (below)
    uint16_t *list1 = reinterpret_cast<uint16_t *>(smap + smap_bytes);
    (void)keep_cap;

    // Image = blockIdx.x (fastest in dispatch order, padded to a multiple of 8), cell = blockIdx.y: consecutive
    // workgroups go round-robin to the 8 XCDs, so image b always lands on XCD b % 8 and the halo columns / rows that
    // neighbouring cells of an image share are served by that XCD's L2 instead of being fetched by up to 8 L2s.
    // (An XCD-banded CELL order was measured 40 % slower earlier: it unbalances the XCDs because corner density
    // differs between pyramid levels; an image-to-XCD affinity has no such effect.)
    const int b = blockIdx.x;
    if (b >= batch) return;
    const int cell_id = (int)blockIdx.y;
    // the cell record as eight scalar dwords (one load; the 16-bit fields unpacked by scalar shifts): everything the kernel needs of
    // the level -- plane offset and pitch -- is in it, so no second, dependent load stands before the tile's first request
    CellDev cell;
    {
        const uint32_t *cw32 = reinterpret_cast<const uint32_t *>(cells + cell_id);
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = cw32[i];
        cell.level = (int16_t)(w[0] & 0xffffu); cell.vx0 = (int16_t)(w[0] >> 16);
        cell.vy0 = (int16_t)(w[1] & 0xffffu); cell.cw = (int16_t)(w[1] >> 16);
        cell.ch = (int16_t)(w[2] & 0xffffu); cell.pitch = (uint16_t)(w[2] >> 16);
        cell.slot_off = (int32_t)w[3]; cell.inv_ndw = w[4]; cell.inv_nq = w[5]; cell.plane_off = w[6]; cell.inv_n16 = w[7];
    }
    struct { int pitch; } lv{(int)cell.pitch};
    const uint8_t *plane = pyr + (size_t)b * pyr_stride + cell.plane_off;
    if (cell.level == 0) {  // level 0 is the caller's image itself (no copy)
        plane = img0 + (size_t)b * img0_stride;
        lv.pitch = pitch0;
    }
    const int cw = cell.cw, ch = cell.ch;
    const int lane = threadIdx.x;
    const int nq = (cw + 3) >> 2;           // 4-pixel groups per row
    const int ndw = nq + 2;                 // dwords staged per row: [halo | nq quads | halo]

    // ---- 0. stage: LDS column c <-> level column vx0 - 4 + c.  Lane grid (rows x ndw dwords) fixed once per wave.
    // (24-bit multiplies throughout: v_mul_u32_u24 is full rate, v_mul_lo_u32 a quarter of it; every index here is far
    // below 2^24.  The uniform parts of the addresses stay in scalar registers.)
    // The tile's pitch is the cell's own 4 * ndw bytes (<= the TP the LDS region was sized for): lane (rs, c) of a row step
    // then owns LDS dword rs * ndw + c = its lane number, which is the one layout gfx950's LDS-DMA can write
    // (global_load_lds_dword: per-lane global address, LDS destination = M0 base + 4 * lane) -- the tile goes from HBM / L2
    // to LDS without passing through VGPRs and without a ds_write per row step.
#if !defined(AOS2_FAST_STAGE_DWORD) && !defined(AOS2_FAST_STAGE_VGPR)
    // (round 3, second step) 16 bytes per lane: global_load_lds_dwordx4 -- a row is n16 = ceil(ndw / 4) lanes, the pitch 16 * n16, a
    // 41-row tile two instructions instead of eight; the bytes past the right halo come from the same image row (the 16-pixel border)
    const int n16 = (ndw + 3) >> 2;
    const int tp = 16 * n16;
    {
        const uint32_t inv = cell.inv_n16;
        const int nrs = (int)((64u * inv) >> 16);      // rows per step = 64 / n16 (n16 <= 5)
        const int rs = (int)(__umul24((uint32_t)lane, inv) >> 16), c = lane - (int)__umul24((uint32_t)rs, (uint32_t)n16);
        const int nrows = ch + 6;
        if (rs < nrs) {
            const uint8_t *sbase = plane + (size_t)(cell.vy0 - 3) * lv.pitch + (cell.vx0 - 4);   // uniform
            const uint32_t soff = __umul24((uint32_t)rs, (uint32_t)lv.pitch) + 16u * (uint32_t)c;
            const uint32_t sstep = (uint32_t)(nrs * lv.pitch);
            const uint32_t dstep = (uint32_t)(nrs * tp);   // the bytes one step's lanes cover
            uint32_t dbase = 0;
            for (int r0 = 0; r0 < nrows; r0 += nrs, sbase += sstep, dbase += dstep)
                if (rs < nrows - r0)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sbase + soff),
                                                     (__attribute__((address_space(3))) void *)(tile + dbase), 16, 0, 0);
        }
    }
#else
    const int tp = 4 * ndw;
    {
        const int nrs = max(1, (int)((64u * cell.inv_ndw) >> 16));      // rows per step = 64 / ndw (cells are < 64 px wide: ndw <= 18)
        const int rs = (int)(__umul24((uint32_t)lane, cell.inv_ndw) >> 16), c = lane - (int)__umul24((uint32_t)rs, (uint32_t)ndw);
        const int nrows = ch + 6;
        if (rs < nrs) {
            const uint8_t *sbase = plane + (size_t)(cell.vy0 - 3) * lv.pitch + (cell.vx0 - 4);   // uniform
            const uint32_t soff = __umul24((uint32_t)rs, (uint32_t)lv.pitch) + 4u * (uint32_t)c;
            const uint32_t sstep = (uint32_t)(nrs * lv.pitch);
#if !defined(AOS2_FAST_STAGE_VGPR)
            const uint32_t dstep = (uint32_t)(nrs * tp);   // = 4 * nrs * ndw: the dwords one step's lanes cover
            uint32_t dbase = 0;
            for (int r0 = 0; r0 < nrows; r0 += nrs, sbase += sstep, dbase += dstep)
                if (rs < nrows - r0)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sbase + soff),
                                                     (__attribute__((address_space(3))) void *)(tile + dbase), 4, 0, 0);
#else
            // (the round-2 form: up to 8 row steps requested together into registers, then written to LDS)
            uint32_t doff = __umul24((uint32_t)rs, (uint32_t)tp) + 4u * (uint32_t)c;
            const uint32_t dstep = (uint32_t)(nrs * tp);
            for (int r0 = 0; r0 < nrows; r0 += 8 * nrs, sbase += 8 * (size_t)sstep, doff += 8 * dstep) {
                uint32_t t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (rs < nrows - r0 - u * nrs) t[u] = load_u32_unaligned(sbase + (size_t)u * sstep + soff);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (rs < nrows - r0 - u * nrs) *reinterpret_cast<uint32_t *>(tile + doff + (uint32_t)u * dstep) = t[u];
            }
#endif
        }
    }
#endif
    const int SH = ch + 2;
#if defined(AOS2_FAST_ABL) && AOS2_FAST_ABL == 1
    __syncthreads();
    if (lane == 0) cell_cnt[(size_t)b * n_cells + cell_id] = tile[5] == 0x7ffffff;
    return;
#endif
    uint32_t *my_slots = slots + (size_t)b * slot_stride + cell.slot_off;
    int th = ini_th;
    int nkept = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = lane; i < (SH * SP + 15) >> 4; i += 64) reinterpret_cast<uint4 *>(smap)[i] = make_uint4(0, 0, 0, 0);   // (16-byte padded region)
        __syncthreads();
        // ---- 2. exact scores of the survivors (called once per pass, or per instalment on overflow).  Corners (S >= th)
        // go to the score map and, compacted IN PLACE (the write index never passes the read index; one wave, LDS
        // operations in program order), to the head of the list: the list stays in row-major order.
        int nc = 0;
        auto score_survivors = [&](int n) {
            nc = 0;   // (instalments: the list is refilled from its start, and NMS walks the score map instead)
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + lane;
                bool corner = false;
                int pos = 0;
                if (i < n) {
                    pos = list1[i];
                    const int py = pos >> 6, px = pos & 63;
                    if (px < cw) {
                        const int S = fast_score_full(tile + __umul24((uint32_t)(py + 3), (uint32_t)tp) + 4 + px, tp);
                        corner = S >= th;
                        if (corner) smap[__umul24((uint32_t)(py + 1), (uint32_t)SP) + px + 1] = (uint8_t)S;
                    }
                }
                const unsigned long long bc = __ballot(corner);
                if (corner) list1[nc + lanes_below(bc)] = (uint16_t)pos;
                nc += __popcll(bc);
            }
        };
        // ---- 1. compass pre-test (a 9-arc contains one pixel of each antipodal compass pair)
        int n1 = 0;
        bool overflowed = false;   // the survivor list was emptied at least once: NMS walks the score map instead
        const us2_t T = {(unsigned short)th, (unsigned short)th};
        (void)T; (void)TH;   // (the margin form compares with th directly)
        const us2_t TH = {(unsigned short)(th << 8), (unsigned short)(th << 8)};   // for operands scaled by 256 (th <= 255)
        (void)T; (void)TH;   // (the margin form compares with th / th << 8 directly)
        const int nitems = nq * ch;
        for (int g0 = 0; g0 < nitems; g0 += 64) {
            if (n1 > 0 && n1 + 256 > list_cap) {  // one iteration appends <= 256 entries (wave-uniform test)
                __syncthreads();
                score_survivors(n1);
                __syncthreads();
                n1 = 0;
                overflowed = true;
            }
            const int g = g0 + lane;
            uint32_t f_lo = 0, f_hi = 0;
            int py = 0, qd = 0;
            if (g < nitems) {
                py = (int)(__umul24((uint32_t)g, cell.inv_nq) >> 16);
                qd = g - (int)__umul24((uint32_t)py, (uint32_t)nq);
                const uint8_t *row = tile + __umul24((uint32_t)(py + 3), (uint32_t)tp) + 4 + 4 * qd;
                const uint32_t C = *reinterpret_cast<const uint32_t *>(row);
                const uint32_t Wd = *reinterpret_cast<const uint32_t *>(row - 4);
                const uint32_t Ed = *reinterpret_cast<const uint32_t *>(row + 4);
                const uint32_t N = *reinterpret_cast<const uint32_t *>(row - 3 * tp);   // ring pixel 8 (dy=-3)
                const uint32_t S = *reinterpret_cast<const uint32_t *>(row + 3 * tp);   // ring pixel 0 (dy=+3)
                const uint32_t Wq = __builtin_amdgcn_alignbyte(C, Wd, 1);               // columns x-3
                const uint32_t Eq = __builtin_amdgcn_alignbyte(Ed, C, 3);               // columns x+3
                const uint32_t M = 0x00ff00ffu, MH = 0xff00ff00u;
#ifndef AOS2_FAST_NO_MARGIN
                f_lo = compass2_margin(as_us2(C & M), as_us2(S & M), as_us2(N & M), as_us2(Eq & M), as_us2(Wq & M));
#else
                f_lo = compass2(as_us2(C & M), as_us2(S & M), as_us2(N & M), as_us2(Eq & M), as_us2(Wq & M), T);
#endif
                // the odd pixels in the high byte of each 16-bit lane, the even pixels' bytes left below them as "fraction": a maximum / minimum of
                // such lanes has the exact high byte, and with H, L, V the high bytes and t the threshold `(H - t) * 256 + g > V * 256 + g'` can
                // differ from `H - t > V` only for H - t == V (the fractions g, g' < 256): a pixel exactly AT the threshold may survive to the
                // exact score, none above it is lost -- five mask instructions less
                (void)MH;
#ifndef AOS2_FAST_NO_MARGIN
                f_hi = compass2_margin(as_us2(C), as_us2(S), as_us2(N), as_us2(Eq), as_us2(Wq));
#else
                f_hi = compass2(as_us2(C), as_us2(S), as_us2(N), as_us2(Eq), as_us2(Wq), TH);
#endif
            }
            // (columns >= cw of the last quad are dropped in phase 2)
            const int x0 = 4 * qd;
#ifndef AOS2_FAST_NO_MARGIN
            // (a lane of the margin is > t -- t << 8 for the odd pixels, whose operands are scaled by 256 -- where the pixel survives; a lane
            // without an item holds 0.  High lane: m.hi > t <=> m > (t << 16 | 0xffff) as unsigned 32-bit numbers)
            const uint32_t th8 = (uint32_t)th << 8;
            const bool p0 = (f_lo & 0xffffu) > (uint32_t)th, p2 = f_lo > (((uint32_t)th << 16) | 0xffffu);
            const bool p1 = (f_hi & 0xffffu) > th8, p3 = f_hi > ((th8 << 16) | 0xffffu);
#else
            const bool p0 = (f_lo & 0xffffu) != 0, p1 = (f_hi & 0xffffu) != 0;
            const bool p2 = (f_lo >> 16) != 0, p3 = (f_hi >> 16) != 0;
#endif
            const unsigned long long b0 = __ballot(p0), b1 = __ballot(p1), b2 = __ballot(p2), b3 = __ballot(p3);
            if ((b0 | b1 | b2 | b3) == 0ull) continue;
            // row-major list order (= cv::FAST's emission order, kept through scoring and NMS, so nothing is sorted
            // later): a lane's entries follow those of all lower lanes
            const int pos0 = (py << 6) | x0;
            int at = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, (uint32_t)n1));
            at = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, (uint32_t)at));
            at = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, (uint32_t)at));
            at = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b3, (uint32_t)at));
            // (`at += p` through v_addc_co_u32 with the ballot as per-lane carry-in -- one instruction instead of v_cndmask + add -- was
            // measured in round 6: -4 static instructions, 0.481 against 0.489 ms alone and nothing on top of the margin form: not kept)
            if (p0) list1[at] = (uint16_t)pos0;
            at += p0;
            if (p1) list1[at] = (uint16_t)(pos0 + 1);
            at += p1;
            if (p2) list1[at] = (uint16_t)(pos0 + 2);
            at += p2;
            if (p3) list1[at] = (uint16_t)(pos0 + 3);
            n1 += __popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3);
        }
        __syncthreads();
#if defined(AOS2_FAST_ABL) && AOS2_FAST_ABL == 2
        if (lane == 0) cell_cnt[(size_t)b * n_cells + cell_id] = (n1 + list1[n1 >> 1]) == 0x7ffffff;
        return;
#endif
        score_survivors(n1);
        __syncthreads();
#if defined(AOS2_FAST_ABL) && AOS2_FAST_ABL == 3
        if (lane == 0) cell_cnt[(size_t)b * n_cells + cell_id] = (n1 + smap[SP + 5]) == 0x7ffffff;
        return;
#endif
        // ---- 3. NMS (strictly greater than the 8 neighbours inside the cell) over the corners, in row-major order: the
        // kept ones go straight to the cell's slots (a pass that keeps nothing writes nothing)
        nkept = 0;
        const int n3 = overflowed ? cw * ch : nc;   // overflow: every pixel of the cell is a candidate position
        for (int i0 = 0; i0 < n3; i0 += 64) {
            const int i = i0 + lane;
            bool keep = false;
            int pos = 0, sc = 0;
            if (i < n3) {
                if (!overflowed)
                    pos = list1[i];
                else {
                    const int yy = i / cw;
                    pos = (yy << 6) | (i - yy * cw);
                }
                const int py = pos >> 6, px = pos & 63;
                const uint8_t *m = smap + __umul24((uint32_t)(py + 1), (uint32_t)SP) + px + 1;
                sc = m[0];
                if (sc > 0)
                    keep = sc > m[-1] && sc > m[1] && sc > m[-SP - 1] && sc > m[-SP] && sc > m[-SP + 1] &&
                           sc > m[SP - 1] && sc > m[SP] && sc > m[SP + 1];
            }
            const unsigned long long bk = __ballot(keep);
            if (keep) {
                const uint32_t xr = (uint32_t)(cell.vx0 - 16) + ((uint32_t)pos & 63), yr = (uint32_t)(cell.vy0 - 16) + ((uint32_t)pos >> 6);
                my_slots[nkept + lanes_below(bk)] = xr | (yr << 12) | ((uint32_t)sc << 24);
            }
            nkept += __popcll(bk);
        }
#if defined(AOS2_FAST_ABL) && AOS2_FAST_ABL == 4
        break;
#endif
        if (nkept > 0 || th == min_th) break;
        th = min_th;  // vKeysCell.empty() -> retry with minThFAST (:812-816)
        __syncthreads();
    }
    if (lane == 0) cell_cnt[(size_t)b * n_cells + cell_id] = nkept;
}

// ---------------------------------------------------------------------------------------------
// Per image: exclusive scan of the cell counts in the reference's emission order (levels, then
// cells row-major) and gather of the per-cell slots into one dense list per image.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void compact_candidates_kernel(const CellDev *__restrict__ cells,
                                                                 int n_cells, int n_levels,
                                                                 const int *__restrict__ level_cell_begin,
                                                                 const uint32_t *__restrict__ slots,
                                                                 size_t slot_stride,
                                                                 const int32_t *__restrict__ cell_cnt,
                                                                 uint32_t *__restrict__ dense,
                                                                 size_t dense_stride,
                                                                 int32_t *__restrict__ level_off)
{
    __shared__ int wsum[4];
    __shared__ int carry;
    __shared__ int cell_off_sh[256];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t *cnt = cell_cnt + (size_t)b * n_cells;
    const uint32_t *sl = slots + (size_t)b * slot_stride;
    uint32_t *out = dense + (size_t)b * dense_stride;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n_cells; c0 += 256) {
        const int c = c0 + tid;
        const int v = c < n_cells ? cnt[c] : 0;
        // wave inclusive scan
        int x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const int excl = carry + woff + x - v;
        cell_off_sh[tid] = excl;
        // level boundaries
        if (c < n_cells) {
            for (int l = 0; l < n_levels; ++l)
                if (level_cell_begin[l] == c) level_off[(size_t)b * (n_levels + 1) + l] = excl;
            const uint32_t *src = sl + cells[c].slot_off;
            for (int i = 0; i < v; ++i) out[excl + i] = src[i];
        }
        __syncthreads();
        if (tid == 255) carry = excl + v;
        __syncthreads();
    }
    if (tid == 0) level_off[(size_t)b * (n_levels + 1) + n_levels] = carry;
}

// ---------------------------------------------------------------------------------------------
// DistributeOctTree on the device: one lane per (image, level) running the shared serial
// routine of octree.h over global scratch.  Latency-bound by construction (pointer-chasing
// control flow of src/ORBextractor.cc:539-763); it exists to keep the candidates on the device.
// ---------------------------------------------------------------------------------------------
#if defined(AOS2_OCT_PROF)
}  // namespace aos2
__device__ long long g_oct_prof[16];
extern "C" int aos2_debug_oct_prof(long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_oct_prof), sizeof(long long) * 16) != hipSuccess) return -4;
    if (reset) {
        long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_oct_prof), z, sizeof(z)) != hipSuccess) return -4;
    }
    return 0;
}
namespace aos2 {
#endif
static_assert(sizeof(OctNode16) == 16, "oct_lds_bytes() assumes 16-byte compact nodes");

// one (image, level) job, executed by ONE wave (lane = threadIdx.x & 63) over the LDS slice [lds, lds + lds_bytes)
template <bool kGroup = false>
__device__ __forceinline__ void octree_job(int b, int l, uint32_t *__restrict__ dense, size_t dense_stride,
                                           const OctGather &G, const LevelDev *__restrict__ levels,
                                           int n_levels, const OctDevScratch &scr, uint32_t *__restrict__ sel,
                                           size_t sel_stride, int32_t *__restrict__ sel_level_cnt, int cap_level,
                                           uint32_t *oct_lds, int lds_bytes)
{
    const int lane = threadIdx.x & 63;
    // ---- candidates of this (image, level) in the reference's order (:775-846: cells row-major, cv::FAST's
    // emission order inside a cell): exclusive scan of the level's cell counts, each lane copies its cells' slots.
    // The dense list of a level starts at the level's first slot, so no job depends on another level's counts.
    const int cb = G.level_cell_begin[l], ce = l + 1 < n_levels ? G.level_cell_begin[l + 1] : G.n_cells;
    uint32_t *cand = dense + (size_t)b * dense_stride + G.cells[cb].slot_off;
    int n = 0;
    {
        const int32_t *cc = G.cell_cnt + (size_t)b * G.n_cells;
        const uint32_t *sl = G.slots + (size_t)b * G.slot_stride;
        // This runs on the job's critical path, so the memory round trips are batched: the counts (and slot
        // offsets) of 8 rounds of 64 cells are requested together, and a cell's slots are moved 16 at a time
        // (16 predicated loads in flight, then 16 stores) instead of one dependent load -> store per element.
        for (int c0 = cb; c0 < ce; c0 += 8 * 64) {
            int v[8], so[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int c = c0 + 64 * r + lane;
                v[r] = c < ce ? cc[c] : 0;
                so[r] = c < ce ? G.cells[c].slot_off : 0;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (c0 + 64 * r >= ce) continue;   // uniform
                const int excl = octdetail::coop_excl_scan(v[r]);
                const uint32_t *src = sl + so[r];
                uint32_t *dst = cand + n + excl;
                const int vmax = octdetail::coop_max(v[r]);
                for (int i0 = 0; i0 < vmax; i0 += 16) {
                    uint32_t t[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (i0 + u < v[r]) t[u] = src[i0 + u];
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (i0 + u < v[r]) dst[i0 + u] = t[u];
                }
                n += octdetail::coop_shfl(excl + v[r], 63);
            }
        }
        if (lane == 0) G.level_cnt[(size_t)b * n_levels + l] = n;
        octdetail::coop_sync();
    }
    const LevelDev lv = levels[l];
    uint32_t *out = sel + (size_t)b * sel_stride + (size_t)l * cap_level;
    int32_t *idx = scr.out_idx + ((size_t)b * n_levels + l) * cap_level;
    if (n > lv.oct_cand_cap) {  // cannot happen: the capacity is the geometric maximum of NMS survivors
        if (lane == 0) sel_level_cnt[(size_t)b * n_levels + l] = -4;
        return;
    }
    int nk = -2;
    if (n > 0 && n < 32768 && oct_lds_bytes(n, lv.nfeat) <= (size_t)lds_bytes) {
        // working set in LDS: packed candidates | perm | tmp | node arena | pairs
        const int ncand = (n + 3) & ~3;
        uint32_t *c_l = oct_lds;
        int16_t *perm_l = reinterpret_cast<int16_t *>(c_l + ncand);
        int16_t *tmp_l = perm_l + ncand;
        OctNode16 *nodes_l = reinterpret_cast<OctNode16 *>(tmp_l + ncand);
        const int mn = oct_lds_nodes(lv.nfeat), mp = oct_lds_pairs(lv.nfeat);
        int32_t *pairs_l = reinterpret_cast<int32_t *>(nodes_l + mn);
        for (int i = lane; i < n; i += 64) c_l[i] = cand[i];
        octdetail::coop_sync();
        const OctCandsPacked C{c_l};
        const OctScratchT<OctCompact> S{nodes_l, perm_l, tmp_l, pairs_l, pairs_l + 2 * mp, mn, mp};
        if (kGroup) {   // the big partitions of this job are shared with the workgroup's other waves (octree.h: GroupCoop)
            octdetail::GroupMail &gm = octdetail::group_mail();
            if (lane == 0) {
                gm.on = 1; gm.cands = c_l; gm.perm = perm_l; gm.tmp = tmp_l;
            }
            octdetail::coop_sync();
            nk = distribute_octree<GroupCoop, OctCompact>(C, n, 16, lv.w - 16, 16, lv.h - 16, lv.nfeat, S, idx, cap_level);
            if (lane == 0) gm.on = 0;
            octdetail::coop_sync();
        } else
            nk = distribute_octree<WaveCoop, OctCompact>(C, n, 16, lv.w - 16, 16, lv.h - 16, lv.nfeat, S, idx, cap_level);
    }
    if (n > 0 && nk == -2) {
        // general path over global scratch (jobs that do not fit the LDS budget, or exhausted its arena)
        const size_t co = (size_t)b * scr.cand_stride + lv.oct_cand_off, no = (size_t)b * scr.node_stride + lv.oct_node_off;
        int16_t *xs = scr.xs + co;
        int16_t *ys = scr.ys + co;
        uint8_t *sc = scr.sc + co;
        for (int i = lane; i < n; i += 64) {
            const uint32_t c = cand[i];
            xs[i] = (int16_t)(c & 0xfff);
            ys[i] = (int16_t)((c >> 12) & 0xfff);
            sc[i] = (uint8_t)(c >> 24);
        }
        octdetail::coop_sync();
        OctScratch S;
        S.nodes = scr.nodes + no;
        S.perm = scr.perm + co;
        S.tmp = scr.tmp + co;
        S.pairs_a = scr.pairs + 4 * no;
        S.pairs_b = S.pairs_a + 2 * (size_t)lv.oct_node_cap;
        S.max_nodes = lv.oct_node_cap;
        S.max_pairs = lv.oct_node_cap;
        nk = distribute_octree<WaveCoop>(xs, ys, sc, n, 16, lv.w - 16, 16, lv.h - 16, lv.nfeat, S, idx, cap_level);
    }
    if (n <= 0) nk = 0;
    octdetail::coop_sync();
    for (int k = lane; k < nk; k += 64) out[k] = cand[idx[k]];
    if (lane == 0) sel_level_cnt[(size_t)b * n_levels + l] = nk;
}

// one job per workgroup; jobs are level-major (all level-0 jobs first): the long jobs start first, the short ones
// fill in.  Every workgroup reserves the level-0 working set.  The jobs of the first `group_levels` levels (thousands of
// candidates: three partition passes over all of them were half of a level-0 job) keep 4 waves: wave 0 runs the job, the
// others serve its big partitions (GroupCoop); in the jobs of the higher levels they leave at once.
__global__ __launch_bounds__(256) void octree_kernel(uint32_t *__restrict__ dense, size_t dense_stride,
                              OctGather gather, const LevelDev *__restrict__ levels,
                              int n_levels, int batch, OctDevScratch scr, uint32_t *__restrict__ sel,
                              size_t sel_stride, int32_t *__restrict__ sel_level_cnt, int cap_level, int lds_bytes,
                              int group_levels)
{
    extern __shared__ uint32_t oct_lds[];
    // the jobs are few, long and serial: let them issue ahead of the VALU-bound waves of other streams' kernels that
    // share the SIMD (without this the kernel stretches from 0.19 to 0.27 ms when it overlaps FAST / describe)
    __builtin_amdgcn_s_setprio(3);
    const int job = blockIdx.x;
    const int l = job / batch, b = job - l * batch;
    const int wave = threadIdx.x >> 6;
    if (l >= group_levels) {   // (uniform per workgroup)
        if (wave == 0)
            octree_job<false>(b, l, dense, dense_stride, gather, levels, n_levels, scr, sel, sel_stride, sel_level_cnt, cap_level,
                              oct_lds, lds_bytes);
        return;
    }
    if (wave == 0) {
        octdetail::GroupMail &gm = octdetail::group_mail();
        if ((threadIdx.x & 63) == 0) gm.on = 0;
        octree_job<true>(b, l, dense, dense_stride, gather, levels, n_levels, scr, sel, sel_stride, sel_level_cnt, cap_level, oct_lds,
                         lds_bytes);
        if ((threadIdx.x & 63) == 0) gm.cmd = 0;   // the job is over: release the helpers
        __syncthreads();
    } else
        octdetail::group_helper_loop<OctCompact>(wave);
}

// one workgroup per image, one wave per level, each with its own LDS slice sized for that level: the LDS
// reservation matches the jobs (the per-job kernel reserves the level-0 size for every level, which halves the
// number of resident jobs), so all (image, level) jobs of a 256-image batch are resident at once.  The waves are
// independent (wave-local fences only).
__global__ __launch_bounds__(1024) void octree_image_kernel(uint32_t *__restrict__ dense, size_t dense_stride,
                              OctGather gather, const LevelDev *__restrict__ levels,
                              int n_levels, OctDevScratch scr, uint32_t *__restrict__ sel, size_t sel_stride,
                              int32_t *__restrict__ sel_level_cnt, int cap_level, OctImageLayout lay)
{
    extern __shared__ uint32_t oct_lds[];
    const int l = threadIdx.x >> 6;
    if (l >= n_levels) return;
    octree_job<false>(blockIdx.x, l, dense, dense_stride, gather, levels, n_levels, scr, sel, sel_stride, sel_level_cnt, cap_level,
                      oct_lds + (lay.off[l] >> 2), lay.bytes[l]);
}

// two levels per workgroup: wave 0 = level g, wave 1 = level n_levels - 1 - g (extractor_kernels.h); jobs stay independent
__global__ __launch_bounds__(128) void octree_pair_kernel(uint32_t *__restrict__ dense, size_t dense_stride,
                              OctGather gather, const LevelDev *__restrict__ levels,
                              int n_levels, int batch, OctDevScratch scr, uint32_t *__restrict__ sel, size_t sel_stride,
                              int32_t *__restrict__ sel_level_cnt, int cap_level, OctImageLayout lay)
{
    extern __shared__ uint32_t oct_lds[];
    __builtin_amdgcn_s_setprio(3);   // (like octree_kernel: few long serial jobs beside other streams' VALU-bound waves)
    const int g = blockIdx.x / batch, b = blockIdx.x - g * batch;   // group-major: the pairs with the long level-0 jobs start first
    const int wave = threadIdx.x >> 6;
    const int l = wave == 0 ? g : n_levels - 1 - g;
    if (wave == 1 && l == g) return;   // (odd level count: the middle level is alone)
    octree_job<false>(b, l, dense, dense_stride, gather, levels, n_levels, scr, sel, sel_stride, sel_level_cnt, cap_level,
                      oct_lds + (lay.off[l] >> 2), lay.bytes[l]);
}

// ---------------------------------------------------------------------------------------------
// Orientation (IC_Angle, :77-104) + 7x7 Gaussian blur (cv::GaussianBlur sigma 2, :1085-1086)
// + steered rBRIEF (computeOrbDescriptor, :108-147), one wave per DK consecutive keypoints.
//
// The 43x43 unblurred patch around the keypoint is staged in LDS once (BORDER_REFLECT_101 at the
// level's edges, exactly what blurring the cloned interior sees).  From it:
//   * integer moments over the circular r=15 patch -> fastAtan2 polynomial (float, no FMA);
//   * horizontal 7-tap pass (exact 16-bit sums, max 255*257) over the (row pair, quad) items a rotated
//     pattern point can reach, stored transposed with two rows per dword;
//   * vertical 7-tap pass over the 37x37 tile, (sum + 2^15) >> 16 -> bytes; the 512 rotated sample
//     positions then read one LDS byte each.
// This fuses the reference's full-level blur into the consumer: only the <= 37x37 footprint a
// descriptor can touch (pattern reach +-13 rotated => +-18) is ever blurred, and the blurred
// pyramid never goes to HBM.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int n)
{
    if (p < 0) p = -p;
    if (p >= n) p = 2 * (n - 1) - p;
    return p;
}

__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

constexpr int PR = 21;            // patch radius staged (18 sample reach + 3 blur taps)
constexpr int PW = 2 * PR + 1;    // 43 rows / columns
constexpr int PD = 12;            // LDS dwords per patch row (48 bytes, 44 used)
constexpr int HR = 18;            // blurred-tile radius (reach of the rotated pattern: |(13,13)| = 18.4)
constexpr int BW = 2 * HR + 1;    // 37 blurred rows / columns
constexpr int HTP = 23;           // row-PAIR pitch (dwords) of the transposed h-blur tile; odd: column walks hit 32 banks
constexpr int HTC = 40;           // columns of the h-blur tile (10 quads, 37 used)
constexpr int VBP = 44;           // byte pitch of the blurred tile, stored transposed ([x][y]); 11 dwords (odd)

#ifndef AOS2_DESC_KPW
#define AOS2_DESC_KPW 8
#endif
constexpr int DK_BATCH = AOS2_DESC_KPW;   // keypoints per wave of describe_kernel: large batches (the per-wave work is shared by 8)
constexpr int DK_FEW = 2;                 // ... a few images (the launch is as long as one wave: short waves)
constexpr int DKB = 4;              // keypoints per wave (describe_blur_kernel)

// LDS traffic of ONE wave is processed in issue order, so a write by one lane is visible to a later read by
// another lane of the same wave; this only stops the compiler from moving LDS accesses across the phase boundary
// (and, unlike __syncthreads(), leaves the global prefetch of the next keypoint in flight).
__device__ __forceinline__ void wave_lds_phase() { asm volatile("" ::: "memory"); }

// One wave (= one 64-thread workgroup) per DK consecutive output slots of one image.  The 43x43 patch of slot
// i+1 is fetched into registers while slot i is processed out of LDS, so the L2/HBM latency of the staging
// phase is covered by the wave's own arithmetic instead of by occupancy.
#ifdef AOS2_DESC_WPE
#define DESC_ATTR __attribute__((amdgpu_waves_per_eu(AOS2_DESC_WPE, 8)))
#else
#define DESC_ATTR
#endif
template <int DK>
__global__ __launch_bounds__(64) DESC_ATTR void describe_kernel(const uint8_t *__restrict__ img0,
                                                      size_t img0_stride, int pitch0,
                                                      const uint8_t *__restrict__ pyr,
                                                      size_t pyr_stride,
                                                      const LevelDev *__restrict__ levels,
                                                      int n_levels, const uint32_t *__restrict__ sel,
                                                      size_t sel_stride, int cap_level,
                                                      const int32_t *__restrict__ sel_level_cnt,
                                                      aos2_keypoint_t *__restrict__ kps,
                                                      uint8_t *__restrict__ desc, int cap,
                                                      int32_t *__restrict__ n_out,
                                                      unsigned long long umax_nibbles,
                                                      int32_t *__restrict__ status, int batch, int groups)
{
    // raw patch, 43 rows (+1 so that the last row pair can be read); once the h-pass is done the same bytes hold the
    // blurred tile (37 x VBP = 1628 B)
    __shared__ __attribute__((aligned(16))) uint32_t patch32[(PW + 1) * PD];
    __shared__ __attribute__((aligned(16))) uint32_t hbT[HTC * HTP];   // h-pass sums, [column][row pair] u16x2
    // Workgroups go to the XCDs round robin in the order of their linear index, and they start in that order.  Index
    // L = 8 (i * groups + y) + x is wave y of image b = 8 i + x: an image's waves all run on XCD b % 8 and right after each
    // other, so that XCD's L2 (4 MB) holds the pyramids of the ~3 images it is working on (1.3 MB each at 640x480) and
    // the overlapping patches of an image's keypoints are fetched from HBM once.  (With the image as the fast index all
    // the batch's images were in flight at once, 64 per L2: 2.8 x the pyramids' bytes came from HBM.)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, bi = slot / groups;
    const int b = 8 * bi + xcd;
    if (b >= batch) return;
    const int lane = threadIdx.x;
    // lane l < n_levels keeps level l's geometry; lane i < DK locates slot k0 + i (levels are concatenated
    // level-major, :1060-1104).  Both loads are independent of each other.
    int lv_w = 0, lv_h = 0, lv_pitch = 0, lv_sp = 0;
    uint32_t lv_off = 0;
    float lv_scale = 0.f;
    if (lane < n_levels) {
        const LevelDev &L = levels[lane];
        lv_w = L.w; lv_h = L.h; lv_pitch = lane == 0 ? pitch0 : L.pitch; lv_off = (uint32_t)L.off;
        lv_sp = L.scaled_patch; lv_scale = L.scale;
    }
    const int32_t *cnt = sel_level_cnt + (size_t)b * n_levels;
    int total = 0, worst = 0;
    for (int l = 0; l < n_levels; ++l) {
        total += cnt[l] > 0 ? cnt[l] : 0;
        worst = min(worst, cnt[l]);   // negative = the octree stage's failure code for this (image, level)
    }
    const int y = slot - bi * groups, k0 = y * DK;   // first output slot of this wave
    const int nk = min(DK, min(total, cap) - k0);
    if (y == 0 && lane == 0) {
        n_out[b] = total;
        // sticky per-handle status, read by aos2_extractor_wait(): [0] = lowest failure code, [1] = largest n_out
        if (worst < 0) atomicMin(status, worst);
        if (total > cap) atomicMax(status + 1, total);
    }
    if (nk <= 0) return;  // wave-uniform
    int my_level = -1, my_kin = k0 + lane;
    for (int l = 0; l < n_levels; ++l) {
        const int c = cnt[l] > 0 ? cnt[l] : 0;
        if (my_level < 0 && my_kin < c) my_level = l;
        if (my_level < 0) my_kin -= c;
    }
    uint32_t my_sel = 0;
    if (lane < nk) my_sel = sel[(size_t)b * sel_stride + (size_t)my_level * cap_level + my_kin];
    const uint8_t *img_pyr = pyr + (size_t)b * pyr_stride;
    const uint8_t *img_l0 = img0 + (size_t)b * img0_stride;
    uint8_t *patch = reinterpret_cast<uint8_t *>(patch32);
    // per-lane constants of the lane grids (each split done once)
    const int st_rs = (lane * 373) >> 12, st_c = lane - 11 * st_rs;   // staging: 5 rows x 11 dwords
    const int ic_rs = (lane * 57) >> 9, ic_dj = lane - 9 * ic_rs + 1; // IC_Angle: 7 rows x 9 dwords (lane 63: none)
    // Gaussian weights (8 fractional bits, sum 257): bytes for the h-pass dot4, u16 pairs for the v-pass dot2
    const uint32_t g0 = c_gauss[0], g1 = c_gauss[1], g2 = c_gauss[2], g3 = c_gauss[3];
    // h-pass: the 4 outputs of a quad read the same 3 dwords; the taps are shifted in the WEIGHTS (10 dot4, no
    // byte realignment of the data).  Byte k of a weight dword multiplies pixel k of the data dword.
    const uint32_t g[7] = {g0, g1, g2, g3, g2, g1, g0};
    auto wq = [&](int first) {   // weights of taps first .. first+3 (taps outside 0..6 are 0)
        uint32_t w = 0;
        for (int k = 0; k < 4; ++k)
            if (first + k >= 0 && first + k < 7) w |= g[first + k] << (8 * k);
        return w;
    };
    const uint32_t WA0 = wq(0), WA1 = wq(4), WB0 = wq(-1), WB1 = wq(3), WC0 = wq(-2), WC1 = wq(2), WC2 = wq(6);
    const uint32_t WD0 = wq(-3), WD1 = wq(1), WD2 = wq(5);
    uint32_t hit2[2];   // this lane's item of rounds (0,1) and (2,3), two u16 per register
#pragma unroll
    for (int t = 0; t < 2; ++t) hit2[t] = (uint32_t)c_hitems[lane + 128 * t] | ((uint32_t)c_hitems[lane + 128 * t + 64] << 16);
    const int hrounds = c_hrounds;
    const uint32_t WE0 = g0 | (g1 << 16), WE1 = g2 | (g3 << 16), WE2 = g2 | (g1 << 16), WE3 = g0;   // even output row
    const uint32_t WO0 = g0 << 16, WO1 = g1 | (g2 << 16), WO2 = g3 | (g2 << 16), WO3 = g1 | (g0 << 16);   // odd
    // v-pass: item id = lane + 64 t = (column x, 8 output rows seg); its source (h-pass tile, bytes) and destination
    // (blurred tile, bytes) offsets, fixed for the whole wave: low / high half of one register per round
    uint32_t vitem[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int id = lane + 64 * t, seg = (int)(__umul24((uint32_t)id, 1772u) >> 16), x = id - BW * seg;   // id / 37
        vitem[t] = id < BW * 5 ? (uint32_t)(4 * (x * HTP + 4 * seg)) | ((uint32_t)(x * VBP + 8 * seg) << 16) : 0xffffffffu;
    }
    // rBRIEF: this lane's test pair of each of the four 64-bit words, the two points of a pair side by side for the
    // packed float instructions (x of both, y of both)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 patx[4], paty[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t pat = *reinterpret_cast<const uint32_t *>(&c_pattern[4 * (r * 64 + lane)]);
        patx[r] = f32x2{(float)(int8_t)pat, (float)(int8_t)(pat >> 16)};
        paty[r] = f32x2{(float)(int8_t)(pat >> 8), (float)(int8_t)(pat >> 24)};
    }

    struct Slot {
        int level, kx, ky, score, w, h, pitch;
        const uint8_t *plane;
        bool interior;
    };
    auto locate = [&](int i) {
        Slot s;
        s.level = __builtin_amdgcn_readlane(my_level, i);
        const uint32_t csel = (uint32_t)__builtin_amdgcn_readlane((int)my_sel, i);
        s.kx = (int)(csel & 0xfff) + 16;           // + minBorderX (:842)
        s.ky = (int)((csel >> 12) & 0xfff) + 16;
        s.score = (int)(csel >> 24);
        s.w = __builtin_amdgcn_readlane(lv_w, s.level);
        s.h = __builtin_amdgcn_readlane(lv_h, s.level);
        s.pitch = __builtin_amdgcn_readlane(lv_pitch, s.level);
        s.plane = s.level == 0 ? img_l0 : img_pyr + (uint32_t)__builtin_amdgcn_readlane((int)lv_off, s.level);
        s.interior = s.kx - PR >= 0 && s.kx + PR + 1 < s.w && s.ky - PR >= 0 && s.ky + PR < s.h;
        return s;
    };
    uint32_t pre[9];
    auto prefetch = [&](const Slot &s) {
        if (s.interior && st_rs < 5) {
            // uniform base in scalar registers + one 32-bit lane offset (24-bit multiply): no 64-bit vector arithmetic
            const uint8_t *base = s.plane + (size_t)(s.ky - PR) * s.pitch + (s.kx - PR);
            const uint32_t loff = __umul24((uint32_t)st_rs, (uint32_t)s.pitch) + 4u * (uint32_t)st_c;
            const uint32_t step = 5u * (uint32_t)s.pitch;
#pragma unroll
            for (int t = 0; t < 9; ++t)
                if (t < 8 || st_rs < 3) pre[t] = load_u32_unaligned(base + (loff + (uint32_t)t * step));
        }
    };
    Slot cur = locate(0);
    prefetch(cur);

    // ---- IC_Angle (:75-105) of the wave's DK keypoints together, then their orientations and steering pairs in DK
    // lane groups at once: the arctangent and the double-precision sine / cosine are the same instructions for every
    // lane, so one pass serves all the keypoints of the wave instead of one pass each.
    // The moments come straight from the level (a keypoint is >= 19 px from every edge, :45, so the 31-row disc and
    // the dwords around it are inside the plane; the patch staged below is only for the blur): lane (rs, dj) reads
    // rows rs, rs + 7, .. of the disc, dword dj = columns u = 4 dj - 21 .. 4 dj - 18.  With the precomputed byte mask,
    // S = sum of the kept bytes, T = sum of k * byte_k:   m10 = sum((4 dj - 21) S + T),   m01 = sum(v S).
    static_assert(DK <= 8, "the moments of at most 8 keypoints are reduced together");
    int mom[16];
    {
        uint32_t icm[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int vr = ic_rs + 7 * t;
            icm[t] = ic_rs < 7 && vr < 31 ? c_icmask[vr * 9 + ic_dj - 1] : 0u;
        }
        const int vr4 = min(ic_rs + 28, 30);   // (rows of lanes without a fifth row: any row of the disc, mask 0)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            mom[2 * i] = mom[2 * i + 1] = 0;
            if (i < DK && i < nk) {   // wave-uniform
                const Slot s = i == 0 ? cur : locate(i);
                const uint8_t *base = s.plane + (size_t)(s.ky - 15) * s.pitch + (s.kx - PR);
                const uint32_t loff = __umul24((uint32_t)ic_rs, (uint32_t)s.pitch) + 4u * (uint32_t)ic_dj;
                const uint32_t step = 7u * (uint32_t)s.pitch;
                uint32_t d[5];
#pragma unroll
                for (int t = 0; t < 4; ++t) d[t] = load_u32_unaligned(base + (loff + (uint32_t)t * step));
                d[4] = load_u32_unaligned(base + (__umul24((uint32_t)vr4, (uint32_t)s.pitch) + 4u * (uint32_t)ic_dj));
                uint32_t S = 0, T = 0, TS = 0;   // TS = sum of t * S_t: the row of S_t is v = rs - 15 + 7 t
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    const uint32_t dm = d[t] & icm[t];
                    S = __builtin_amdgcn_udot4(dm, 0x01010101u, S, false);
                    T = __builtin_amdgcn_udot4(dm, 0x03020100u, T, false);
                    if (t) TS = __builtin_amdgcn_udot4(dm, 0x01010101u * (uint32_t)t, TS, false);
                }
                mom[2 * i] = __mul24(4 * ic_dj - PR, (int)S) + (int)T;
                mom[2 * i + 1] = __mul24(ic_rs - 15, (int)S) + 7 * (int)TS;
            }
        }
    }
    // sums over the wave, each landing in its own lanes (no scalar round trips): a pair of registers exchanges halves
    // (v_permlane32_swap), then rows (v_permlane16_swap), and the two are added; bits 3 and 2 of the lane then select
    // which register a lane keeps while the other one travels (row_ror:8, row_shl / shr:4).  After that lanes 0-31 hold
    // m10, lanes 32-63 m01, of keypoint (lane bit 4) + 2 (lane bit 3) + 4 (lane bit 2).
    float kp_angle, kp_sin, kp_cos;
    {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        auto swap_add32 = [](int a, int b) {   // lanes < 32: a over both halves; lanes >= 32: b
            const u32x2 r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
            return (int)(r.x + r.y);
        };
        auto swap_add16 = [](int a, int b) {   // even rows: a over the row pair; odd rows: b
            const u32x2 r = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
            return (int)(r.x + r.y);
        };
        int w[8], x[4], y[2];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = swap_add32(mom[2 * i], mom[2 * i + 1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = swap_add16(w[2 * j], w[2 * j + 1]);
        const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int keep = b3 ? x[2 * m + 1] : x[2 * m], give = b3 ? x[2 * m] : x[2 * m + 1];
            y[m] = keep + __builtin_amdgcn_update_dpp(0, give, 0x128, 0xf, 0xf, false);   // row_ror:8 = lane ^ 8
        }
        const int keep = b2 ? y[1] : y[0], give = b2 ? y[0] : y[1];
        int z = keep + __builtin_amdgcn_update_dpp(0, give, 0x104, 0xf, 0x5, false)    // row_shl:4 into lanes with bit 2 = 0
                     + __builtin_amdgcn_update_dpp(0, give, 0x114, 0xf, 0xa, false);   // row_shr:4 into lanes with bit 2 = 1
        z += __builtin_amdgcn_update_dpp(0, z, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
        z += __builtin_amdgcn_update_dpp(0, z, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
        const u32x2 mm = __builtin_amdgcn_permlane32_swap((unsigned)z, (unsigned)z, false, false);   // x: m10, y: m01, in all lanes
        kp_angle = fast_atan2_deg((float)(int)mm.y, (float)(int)mm.x);
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        sincos_exact(__fmul_rn(kp_angle, factorPI), &kp_sin, &kp_cos);
    }
    auto angle_lane = [](int i) { return 16 * (i & 1) + 8 * ((i >> 1) & 1) + 4 * (i >> 2); };   // where keypoint i's orientation is

    for (int i = 0; i < nk; ++i) {
        // ---- stage the 43x43 patch (BORDER_REFLECT_101 at the level's edges)
        if (cur.interior) {
            if (st_rs < 5) {
                uint32_t *dst = patch32 + st_rs * PD + st_c;
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    if (t < 8 || st_rs < 3) dst[t * 5 * PD] = pre[t];
            }
        } else {
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)   // rare path: keep it out of the register budget
            for (int q = lane; q < PW * PW; q += 64) {
                const int r = q / PW, cc = q - r * PW;
                const int yy = reflect101(cur.ky - PR + r, cur.h), xx = reflect101(cur.kx - PR + cc, cur.w);
                patch[r * (4 * PD) + cc] = cur.plane[(size_t)yy * cur.pitch + xx];
            }
        }
        Slot nxt = cur;
        if (i + 1 < nk) {
            nxt = locate(i + 1);
            prefetch(nxt);
        }
        wave_lds_phase();
        // ---- horizontal 7-tap pass (exact 16-bit sums): item = (row pair m, output quad j) from c_hitems; output
        // column cc (0..36) <-> patch column cc+3.  Stored transposed, rows 2m / 2m+1 packed in one dword, for the
        // v-pass dot2.
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t >= hrounds) continue;   // uniform
            // (the per-round addresses are loop-invariant and stay in registers: 76 VGPRs = 6 waves/SIMD measured
            // faster, 0.313 ms, than recomputing them per slot at 7 waves, 0.328 ms -- the kernel is VALU-bound)
            const uint32_t it = (hit2[t >> 1] >> (16 * (t & 1))) & 0xffffu;
            if (it != 0xffffu) {
                const int m = it >> 8, j = it & 255;
                const uint32_t *src = patch32 + 2 * m * PD + j;
                uint32_t o[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t D0 = src[h * PD], D1 = src[h * PD + 1], D2 = src[h * PD + 2];
                    o[h][0] = __builtin_amdgcn_udot4(D0, WA0, __builtin_amdgcn_udot4(D1, WA1, 0u, false), false);
                    o[h][1] = __builtin_amdgcn_udot4(D0, WB0, __builtin_amdgcn_udot4(D1, WB1, 0u, false), false);
                    o[h][2] = __builtin_amdgcn_udot4(D0, WC0, __builtin_amdgcn_udot4(D1, WC1, __builtin_amdgcn_udot4(D2, WC2, 0u, false), false), false);
                    o[h][3] = __builtin_amdgcn_udot4(D0, WD0, __builtin_amdgcn_udot4(D1, WD1, __builtin_amdgcn_udot4(D2, WD2, 0u, false), false), false);
                }
                uint32_t *dst = hbT + 4 * j * HTP + m;
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q * HTP] = o[0][q] | (o[1][q] << 16);   // each <= 255*257 = 65535
            }
        }
        wave_lds_phase();
        // ---- vertical 7-tap pass over the whole 37x37 tile: item = (column x, 8 output rows); two taps per dot2
        {
            uint8_t *vb = patch;   // the raw patch is dead
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (vitem[t] == 0xffffffffu) continue;
                const uint32_t *src = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(hbT) + (vitem[t] & 0xffffu));
                uint32_t P[7];
#pragma unroll
                for (int u = 0; u < 7; ++u) P[u] = src[u];
                uint32_t w[2] = {0u, 0u};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    uint32_t e = udot2_u16(P[u], WE0, 1u << 15), od = udot2_u16(P[u], WO0, 1u << 15);
                    e = udot2_u16(P[u + 1], WE1, e); od = udot2_u16(P[u + 1], WO1, od);
                    e = udot2_u16(P[u + 2], WE2, e); od = udot2_u16(P[u + 2], WO2, od);
                    e = udot2_u16(P[u + 3], WE3, e); od = udot2_u16(P[u + 3], WO3, od);
                    e = min(e >> 16, 255u); od = min(od >> 16, 255u);
                    w[u >> 1] |= (e | (od << 8)) << (16 * (u & 1));
                }
                uint32_t *dst = reinterpret_cast<uint32_t *>(vb + (vitem[t] >> 16));
                dst[0] = w[0]; dst[1] = w[1];
            }
        }
        wave_lds_phase();
        // ---- steered rBRIEF on the blurred tile: one LDS byte per sample
        const int src_lane = angle_lane(i);
        const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, kp_cos), src_lane));
        const float bb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, kp_sin), src_lane));
        // cvRound by the 1.5 * 2^23 trick: the low bits of (f + MAGIC) are 0x4B400000 + rint(f) for |f| < 2^22, with the
        // FPU's round-to-nearest-even = cvRound's rounding; the constant parts of both coordinates fold into K.
        // Products and sums are separately rounded (:118-121 in float, no contraction), two points per instruction.
        const float MAGIC = 12582912.f;
        const uint32_t K = 0x400000u * (uint32_t)VBP + 0x4B400000u - (uint32_t)(HR * VBP + HR);
        const f32x2 va = f32x2{a, a}, vb2 = f32x2{bb, bb}, vmagic = f32x2{MAGIC, MAGIC};
        uint32_t dsc = 0;   // lanes 0-7: the eight dwords of the descriptor
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x2 fy, fx;
            {
#pragma clang fp contract(off)
                const f32x2 t0 = patx[r] * vb2, t1 = paty[r] * va, t2 = patx[r] * va, t3 = paty[r] * vb2;
                const f32x2 sy = t0 + t1, sx = t2 - t3;
                fy = sy + vmagic;
                fx = sx + vmagic;
            }
            // (__builtin_bit_cast of a vector element folds to nothing with this compiler: __float_as_int)
            const uint32_t off0 = (uint32_t)(__mul24(__float_as_int(fx.x), VBP) + __float_as_int(fy.x)) - K;
            const uint32_t off1 = (uint32_t)(__mul24(__float_as_int(fx.y), VBP) + __float_as_int(fy.y)) - K;
            const int v0 = patch[off0], v1 = patch[off1];
            const unsigned long long word = __ballot(v0 < v1);
            // lanes 2r, 2r + 1 of one register collect the word: v_writelane_b32 with the lane as a constant (no builtin
            // for it here; the s_nop covers the wait states between the compare that writes the scalar pair and its
            // use by v_writelane, which the compiler cannot see inside the asm)
            asm("s_nop 3\n\tv_writelane_b32 %0, %1, %3\n\tv_writelane_b32 %0, %2, %4"
                : "+v"(dsc)
                : "s"((uint32_t)word), "s"((uint32_t)(word >> 32)), "n"(2 * r), "n"(2 * r + 1));
        }
        wave_lds_phase();
        if (lane < 8) reinterpret_cast<uint32_t *>(desc + ((size_t)b * cap + (k0 + i)) * 32)[lane] = dsc;
        cur = nxt;
    }
    // ---- lane i writes the record of keypoint i (:1060-1104: coordinates scaled to level 0)
    {
        const int level = max(my_level, 0);
        const float scale = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * level, __builtin_bit_cast(int, lv_scale)));
        const int sp = __builtin_amdgcn_ds_bpermute(4 * level, lv_sp);
        const float angle = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * angle_lane(lane & 7), __builtin_bit_cast(int, kp_angle)));
        if (lane < nk) {
            const int k = k0 + lane;
            const int kx = (int)(my_sel & 0xfff) + 16, ky = (int)((my_sel >> 12) & 0xfff) + 16;
            aos2_keypoint_t kp;
            kp.x = level != 0 ? __fmul_rn((float)kx, scale) : (float)kx;
            kp.y = level != 0 ? __fmul_rn((float)ky, scale) : (float)ky;
            kp.size = (float)sp;
            kp.angle = angle;
            kp.response = (float)(int)(my_sel >> 24);
            kp.octave = level;
            kp.class_id = -1;
            kps[(size_t)b * cap + k] = kp;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The reference's own form of the blur (src/ORBextractor.cc:1085-1086: GaussianBlur(workingMat, 7x7, sigma 2, BORDER_REFLECT_101)
// of every whole level before its descriptors) as a streaming pass: AOS2_DESC_BLUR=level.  A/B against the per-keypoint blur
// of describe_kernel above (profiles/README.md, round 4): the whole pyramid is blurred once (0.95 M pixels per 640x480 frame
// instead of ~1.5 M pixel-blurs over the overlapping 43x37 patches of 1000 keypoints) but goes to HBM and back.
//
// blur_levels_kernel: one launch for all levels.  A lane owns 4 adjacent output pixels x BL_ROWS rows: per source row one
// horizontal pass (the three dwords around the quad, 10 v_dot4 with the taps shifted in the weights -- describe_kernel's h-pass),
// the last 7 rows' sums stay in registers, per output row the symmetric vertical pass (3 adds + 4 multiply-adds per pixel),
// (sum + 2^15) >> 16 saturated, one aligned 32-bit store.  Same integer arithmetic as the per-keypoint form: same bytes.
// ---------------------------------------------------------------------------------------------
constexpr int BL_ROWS = 16;
struct BlurPlan {
    int first[9];            // first item of each level (item = (band, quad)); first[n_levels] = total
    int nq[8];               // quads per row
    int nq_in[8];            // ... of which interior (quads 1 .. nq_in: the 12 bytes around them lie inside the row); a level's interior
                             // items come first, the edge quads (0 and the last two or three of a row, REFLECT_101 byte by byte) behind
                             // them in waves of their own: mixed into every wave they doubled the kernel's time
    uint32_t dst_off[8];     // byte offset of the level's blurred plane inside one image's block
    int dst_pitch[8];
};

__global__ __launch_bounds__(256) void blur_levels_kernel(const uint8_t *__restrict__ img0, size_t img0_stride, int pitch0,
                                                          const uint8_t *__restrict__ pyr, size_t pyr_stride,
                                                          const LevelDev *__restrict__ levels, int n_levels, BlurPlan plan,
                                                          uint8_t *__restrict__ blur, size_t blur_stride)
{
    const int b = blockIdx.y;
    const int it = blockIdx.x * 256 + threadIdx.x;
    if (it >= plan.first[n_levels]) return;
    int l = 0;
    while (l + 1 < n_levels && it >= plan.first[l + 1]) ++l;
    const LevelDev &L = levels[l];
    const int w = L.w, h = L.h, pitch = l == 0 ? pitch0 : L.pitch;
    const uint8_t *plane = l == 0 ? img0 + (size_t)b * img0_stride : pyr + (size_t)b * pyr_stride + L.off;
    uint8_t *dst = blur + (size_t)b * blur_stride + plan.dst_off[l];
    const int dpitch = plan.dst_pitch[l];
    const int id = it - plan.first[l], nq = plan.nq[l], nqi = plan.nq_in[l], nbands = (h + BL_ROWS - 1) / BL_ROWS;
    int band, q;
    if (id < nbands * nqi) {
        band = id / nqi;
        q = 1 + (id - band * nqi);
    } else {
        const int ne = nq - nqi, ie = id - nbands * nqi;
        band = ie / ne;
        q = ie - band * ne;
        q = q == 0 ? 0 : nqi + q;   // quad 0, then the quads behind the interior ones
    }
    const int x0 = 4 * q, y0 = band * BL_ROWS;
    const uint32_t g0 = c_gauss[0], g1 = c_gauss[1], g2 = c_gauss[2], g3 = c_gauss[3];
    const uint32_t g[7] = {g0, g1, g2, g3, g2, g1, g0};
    auto wq = [&](int first) {   // weights of taps first .. first+3 (taps outside 0..6 are 0)
        uint32_t wv = 0;
        for (int k = 0; k < 4; ++k)
            if (first + k >= 0 && first + k < 7) wv |= g[first + k] << (8 * k);
        return wv;
    };
    const uint32_t WA0 = wq(0), WA1 = wq(4), WB0 = wq(-1), WB1 = wq(3), WC0 = wq(-2), WC1 = wq(2), WC2 = wq(6);
    const uint32_t WD0 = wq(-3), WD1 = wq(1), WD2 = wq(5);
    const bool inner = x0 >= 3 && x0 + 8 < w;   // the 12 bytes x0 - 3 .. x0 + 8 lie inside the row
    uint32_t hs[7][4];
#pragma unroll
    for (int r = 0; r < BL_ROWS + 6; ++r) {
        // source row y0 + r - 3 (REFLECT_101 at the level's edges); rows past the band's last needed one are skipped
        const int ys = y0 + r - 3;
        uint32_t D0, D1, D2;
        if (ys - 3 < h) {   // (row ys feeds output rows ys - 3 .. ys + 3: needed iff ys - 3 < h)
            const uint8_t *row = plane + (size_t)reflect101(ys, h) * pitch;
            if (inner) {
                D0 = load_u32_unaligned(row + x0 - 3);
                D1 = load_u32_unaligned(row + x0 + 1);
                D2 = load_u32_unaligned(row + x0 + 5);
            } else {
                uint32_t d[3] = {0, 0, 0};
#pragma unroll
                for (int k = 0; k < 10; ++k) d[k >> 2] |= (uint32_t)row[reflect101(x0 - 3 + k, w)] << (8 * (k & 3));
                D0 = d[0]; D1 = d[1]; D2 = d[2];
            }
        } else
            D0 = D1 = D2 = 0;
        uint32_t *o = hs[r % 7];
        o[0] = __builtin_amdgcn_udot4(D0, WA0, __builtin_amdgcn_udot4(D1, WA1, 0u, false), false);
        o[1] = __builtin_amdgcn_udot4(D0, WB0, __builtin_amdgcn_udot4(D1, WB1, 0u, false), false);
        o[2] = __builtin_amdgcn_udot4(D0, WC0, __builtin_amdgcn_udot4(D1, WC1, __builtin_amdgcn_udot4(D2, WC2, 0u, false), false), false);
        o[3] = __builtin_amdgcn_udot4(D0, WD0, __builtin_amdgcn_udot4(D1, WD1, __builtin_amdgcn_udot4(D2, WD2, 0u, false), false), false);
        if (r >= 6) {
            const int y = y0 + r - 6;
            if (y < h) {
                uint32_t out = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t a0 = hs[(r - 6) % 7][k] + hs[r % 7][k], a1 = hs[(r - 5) % 7][k] + hs[(r - 1) % 7][k];
                    const uint32_t a2 = hs[(r - 4) % 7][k] + hs[(r - 2) % 7][k], a3 = hs[(r - 3) % 7][k];
                    uint32_t v = __umul24(a0, g0) + (1u << 15);
                    v += __umul24(a1, g1);
                    v += __umul24(a2, g2);
                    v += __umul24(a3, g3);
                    out |= min(v >> 16, 255u) << (8 * k);
                }
                *reinterpret_cast<uint32_t *>(dst + (size_t)y * dpitch + x0) = out;
            }
        }
    }
}

// describe_kernel on a blurred pyramid: per keypoint the raw 31-row disc for IC_Angle and the 37x37 blurred tile are staged (no
// reflection: a keypoint lies >= 19 pixels inside its level, the pattern reaches 18), then the steered comparisons.
constexpr int RP = 9;      // dwords per raw row staged (36 bytes: columns -17 .. +18; IC_Angle reads -15 .. +15)
constexpr int BP = 40;     // byte pitch of the blurred tile in LDS ([y][x], 37 used)
__global__ __launch_bounds__(64) void describe_blur_kernel(const uint8_t *__restrict__ img0, size_t img0_stride, int pitch0,
                                                           const uint8_t *__restrict__ pyr, size_t pyr_stride,
                                                           const uint8_t *__restrict__ blur, size_t blur_stride, BlurPlan plan,
                                                           const LevelDev *__restrict__ levels, int n_levels,
                                                           const uint32_t *__restrict__ sel, size_t sel_stride, int cap_level,
                                                           const int32_t *__restrict__ sel_level_cnt,
                                                           aos2_keypoint_t *__restrict__ kps, uint8_t *__restrict__ desc, int cap,
                                                           int32_t *__restrict__ n_out, int32_t *__restrict__ status, int batch)
{
    __shared__ __attribute__((aligned(16))) uint32_t raw32[31 * RP + 1];
    __shared__ __attribute__((aligned(16))) uint32_t bl32[BW * (BP / 4)];
    const int b = blockIdx.x;
    if (b >= batch) return;
    const int lane = threadIdx.x;
    const int k0 = blockIdx.y * DKB;
    int lv_pitch = 0, lv_sp = 0, lv_bpitch = 0;
    uint32_t lv_off = 0, lv_boff = 0;
    float lv_scale = 0.f;
    if (lane < n_levels) {
        const LevelDev &L = levels[lane];
        lv_pitch = lane == 0 ? pitch0 : L.pitch; lv_off = (uint32_t)L.off;
        lv_sp = L.scaled_patch; lv_scale = L.scale;
        lv_boff = plan.dst_off[lane]; lv_bpitch = plan.dst_pitch[lane];
    }
    const int32_t *cnt = sel_level_cnt + (size_t)b * n_levels;
    int my_level = -1, my_kin = k0 + lane, total = 0, worst = 0;
    for (int l = 0; l < n_levels; ++l) {
        const int c = cnt[l] > 0 ? cnt[l] : 0;
        worst = min(worst, cnt[l]);
        if (my_level < 0 && my_kin < c) my_level = l;
        if (my_level < 0) my_kin -= c;
        total += c;
    }
    if (k0 == 0 && lane == 0) {
        n_out[b] = total;
        if (worst < 0) atomicMin(status, worst);
        if (total > cap) atomicMax(status + 1, total);
    }
    const int nk = min(DKB, min(total, cap) - k0);
    if (nk <= 0) return;
    uint32_t my_sel = 0;
    if (lane < nk) my_sel = sel[(size_t)b * sel_stride + (size_t)my_level * cap_level + my_kin];
    const uint8_t *img_pyr = pyr + (size_t)b * pyr_stride, *img_l0 = img0 + (size_t)b * img0_stride;
    const uint8_t *img_bl = blur + (size_t)b * blur_stride;
    const uint8_t *bl8 = reinterpret_cast<const uint8_t *>(bl32);
    uint32_t pats[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pats[r] = *reinterpret_cast<const uint32_t *>(&c_pattern[4 * (r * 64 + lane)]);
    const int ic_rs = (lane * 57) >> 9, ic_dj = lane - 9 * ic_rs + 1;   // IC_Angle: 7 rows x 9 dwords
    // staging grids: raw 31 rows x 9 dwords = 279 items (5 rounds), blurred 37 rows x 10 dwords = 370 items (6 rounds); the loads
    // of keypoint i + 1 are in flight while keypoint i is processed out of LDS (like describe_kernel's prefetch)
    struct Slot {
        int level, kx, ky, score;
    };
    auto locate = [&](int i) {
        Slot sl;
        sl.level = __builtin_amdgcn_readlane(my_level, i);
        const uint32_t csel = (uint32_t)__builtin_amdgcn_readlane((int)my_sel, i);
        sl.kx = (int)(csel & 0xfff) + 16;
        sl.ky = (int)((csel >> 12) & 0xfff) + 16;
        sl.score = (int)(csel >> 24);
        return sl;
    };
    uint32_t rw[5], bw[6];
    auto prefetch = [&](const Slot &sl) {
        const int pitch = __builtin_amdgcn_readlane(lv_pitch, sl.level), bpitch = __builtin_amdgcn_readlane(lv_bpitch, sl.level);
        const uint8_t *plane = sl.level == 0 ? img_l0 : img_pyr + (uint32_t)__builtin_amdgcn_readlane((int)lv_off, sl.level);
        const uint8_t *bplane = img_bl + (uint32_t)__builtin_amdgcn_readlane((int)lv_boff, sl.level);
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int id = lane + 64 * t, rr = id / RP, cc = id - rr * RP;
            rw[t] = id < 31 * RP ? load_u32_unaligned(plane + (size_t)(sl.ky - 15 + rr) * pitch + (sl.kx - 17 + 4 * cc)) : 0u;
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int id = lane + 64 * t, rr = id / 10, cc = id - rr * 10;
            bw[t] = id < BW * 10 ? load_u32_unaligned(bplane + (size_t)(sl.ky - HR + rr) * bpitch + (sl.kx - HR + 4 * cc)) : 0u;
        }
    };
    Slot cur = locate(0);
    prefetch(cur);
    for (int i = 0; i < nk; ++i) {
        const int k = k0 + i;
        const int level = cur.level, kx = cur.kx, ky = cur.ky, score = cur.score;
        wave_lds_phase();   // (the previous keypoint's reads are done)
#pragma unroll
        for (int t = 0; t < 5; ++t)
            if (lane + 64 * t < 31 * RP) raw32[lane + 64 * t] = rw[t];
#pragma unroll
        for (int t = 0; t < 6; ++t)
            if (lane + 64 * t < BW * 10) bl32[lane + 64 * t] = bw[t];
        if (i + 1 < nk) {
            cur = locate(i + 1);
            prefetch(cur);
        }
        wave_lds_phase();
        // ---- IC_Angle: integer moments over the circular patch, 4 pixels per LDS dword.  Raw dword (row vr, dj - 1) holds the
        // columns u = 4 dj - 21 .. 4 dj - 18 -- the alignment of describe_kernel's patch dwords 1..9, so its byte masks apply
        int m10 = 0, m01 = 0;
        if (ic_rs < 7) {
            const int c0 = 4 * ic_dj - PR;
            for (int vr = ic_rs; vr < 31; vr += 7) {
                const uint32_t d = raw32[vr * RP + ic_dj - 1] & c_icmask[vr * 9 + ic_dj - 1];
                const int S = (int)__builtin_amdgcn_udot4(d, 0x01010101u, 0u, false);
                const int T = (int)__builtin_amdgcn_udot4(d, 0x03020100u, 0u, false);
                m10 += c0 * S + T;
                m01 += (vr - 15) * S;
            }
        }
        m10 = wave_sum_i32(m10);
        m01 = wave_sum_i32(m01);
        const float angle = fast_atan2_deg((float)m01, (float)m10);
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        const float ang = __fmul_rn(angle, factorPI);
        float a, bb;
        sincos_exact(ang, &bb, &a);
        unsigned long long words[4];
        const float MAGIC = 12582912.f;
        const uint32_t K = 0x400000u * (uint32_t)BP + 0x4B400000u - (uint32_t)(HR * BP + HR);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t pat = pats[r];
            int val[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float px = (float)(int8_t)(pat >> (16 * q)), py = (float)(int8_t)(pat >> (16 * q + 8));
                const float fy = __fadd_rn(__fadd_rn(__fmul_rn(px, bb), __fmul_rn(py, a)), MAGIC);
                const float fx = __fadd_rn(__fsub_rn(__fmul_rn(px, a), __fmul_rn(py, bb)), MAGIC);
                const uint32_t off = (uint32_t)(__mul24(__builtin_bit_cast(int, fy), BP) + __builtin_bit_cast(int, fx)) - K;
                val[q] = bl8[off];
            }
            words[r] = __ballot(val[0] < val[1]);
        }
        if (lane == 0) {
            unsigned long long *d = reinterpret_cast<unsigned long long *>(desc + ((size_t)b * cap + k) * 32);
            d[0] = words[0]; d[1] = words[1]; d[2] = words[2]; d[3] = words[3];
            aos2_keypoint_t kp;
            const float scale = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lv_scale), level));
            kp.x = level != 0 ? __fmul_rn((float)kx, scale) : (float)kx;
            kp.y = level != 0 ? __fmul_rn((float)ky, scale) : (float)ky;
            kp.size = (float)__builtin_amdgcn_readlane(lv_sp, level);
            kp.angle = angle;
            kp.response = (float)score;
            kp.octave = level;
            kp.class_id = -1;
            kps[(size_t)b * cap + k] = kp;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The whole pyramid of an image in ONE launch (ComputePyramid, src/ORBextractor.cc:1107-1132: level k is cv::resize of
// level k - 1).  A workgroup owns one tile of every level (the tiles of a workgroup sit on top of each other) and walks
// the levels through two LDS buffers: level k - 1's region -> level k's region, of which it stores the part it owns.
// A region holds what the workgroup owns at that level plus what its deeper levels read (a few pixels of halo, computed
// redundantly by the neighbours; the host derives the regions from the resize tables, top level down: tile_x / tile_y,
// int4 {own0, own1, need0, need1} per (level, tile column / row)).  Per pixel the arithmetic is resize_level_kernel's:
// h = (a0 p[sx] + a1 p[sx + 1]) >> 4 on the two source rows, v = ((b0 h0 >> 16) + (b1 h1 >> 16) + 2) >> 2.
// Against the 7 dependent launches of the per-level kernel: the levels are read from HBM / L2 zero times instead of once
// each, and a single frame pays one launch.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pyramid_fused_kernel(const uint8_t *__restrict__ img0, size_t img0_stride, int pitch0,
                                                            uint8_t *__restrict__ pyr, size_t pyr_stride,
                                                            const LevelDev *__restrict__ levels, int nlevels,
                                                            const int4 *__restrict__ tile_x, const int4 *__restrict__ tile_y,
                                                            int ntx, int nty, const int *__restrict__ xofs,
                                                            const int *__restrict__ xab, const int *__restrict__ yofs,
                                                            const int *__restrict__ yab, int buf_pitch, int buf_rows, int batch)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t psm[];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    if (b >= batch) return;
    const int txi = (int)blockIdx.x % ntx, tyi = (int)blockIdx.x / ntx;
    uint8_t *bufA = psm, *bufB = psm + (size_t)buf_pitch * buf_rows;
    int *t_xo = reinterpret_cast<int *>(bufB + (size_t)buf_pitch * buf_rows);   // buf_pitch entries each
    int *t_xa = t_xo + buf_pitch, *t_yo = t_xa + buf_pitch, *t_ya = t_yo + buf_rows;
    // level 0: the region of the caller's image the workgroup's level-1 region reads
    int4 px = tile_x[txi], py = tile_y[tyi];   // level 0 entries
    {
        const int w0 = px.w - px.z, h0 = py.w - py.z;
        const uint8_t *src = img0 + (size_t)b * img0_stride;
        const float inv_w = w0 > 0 ? 1.0f / (float)w0 : 0.f;
        for (int i = tid; i < w0 * h0; i += 256) {
            const int ry = (int)(((float)i + 0.5f) * inv_w), rx = i - ry * w0;
            bufA[ry * buf_pitch + rx] = src[(size_t)(py.z + ry) * pitch0 + px.z + rx];
        }
    }
    uint8_t *prev = bufA, *cur = bufB;
    for (int k = 1; k < nlevels; ++k) {
        const LevelDev lv = levels[k];
        const int4 cx = tile_x[k * ntx + txi], cy = tile_y[k * nty + tyi];
        const int w = cx.w - cx.z, h = cy.w - cy.z;   // this level's region
        const int wprev = levels[k - 1].w, hmax = levels[k - 1].h - 1;
        for (int i = tid; i < w; i += 256) {
            t_xo[i] = xofs[lv.tab_x + cx.z + i];
            t_xa[i] = xab[lv.tab_x + cx.z + i];
        }
        for (int i = tid; i < h; i += 256) {
            t_yo[i] = yofs[lv.tab_y + cy.z + i];
            t_ya[i] = yab[lv.tab_y + cy.z + i];
        }
        __syncthreads();   // tables + the previous level's region
        uint8_t *dst = pyr + (size_t)b * pyr_stride + lv.off;
        const float inv_w = w > 0 ? 1.0f / (float)w : 0.f;
        for (int i = tid; i < w * h; i += 256) {
            const int ry = (int)(((float)i + 0.5f) * inv_w), rx = i - ry * w;
            const int sx = t_xo[rx], sy = t_yo[ry];
            const uint32_t a = (uint32_t)t_xa[rx], bb = (uint32_t)t_ya[ry];
            const int sx1 = sx + 1 < wprev ? sx + 1 : sx;
            const int r0 = min(max(sy, 0), hmax), r1 = min(max(sy + 1, 0), hmax);
            const uint8_t *p0 = prev + (r0 - py.z) * buf_pitch - px.z, *p1 = prev + (r1 - py.z) * buf_pitch - px.z;
            const uint32_t a0 = a & 0xffffu, a1 = a >> 16;
            const uint32_t h0 = (a0 * p0[sx] + a1 * p0[sx1]) >> 4, h1 = (a0 * p1[sx] + a1 * p1[sx1]) >> 4;
            const uint32_t v = (__umulhi(bb << 16, h0) + __umulhi(bb & 0xffff0000u, h1) + 2u) >> 2;   // <= 255
            cur[ry * buf_pitch + rx] = (uint8_t)v;
            const int x = cx.z + rx, y = cy.z + ry;
            if (x >= cx.x && x < cx.y && y >= cy.x && y < cy.y) dst[(size_t)y * lv.pitch + x] = (uint8_t)v;
        }
        // the per-level kernel stores whole quads: the columns between w and the next multiple of 4 are zero
        if (cx.y == lv.w && (lv.w & 3)) {
            const int padw = 4 - (lv.w & 3), oh = cy.y - cy.x;
            for (int i = tid; i < padw * oh; i += 256) dst[(size_t)(cy.x + i / padw) * lv.pitch + lv.w + i % padw] = 0;
        }
        __syncthreads();   // this level's region complete (and the tables free) before the next level
        uint8_t *t = prev; prev = cur; cur = t;
        px = cx; py = cy;
    }
}

// host launchers -------------------------------------------------------------------------------
void launch_resize(const uint8_t *src_base, size_t src_img_stride, int src_pitch, uint8_t *pyr, size_t pyr_stride,
                   const LevelDev &src, const LevelDev &dst, const int *xofs, const int *xab, const int *yofs,
                   const int *yab, int batch, hipStream_t st)
{
    const int nquads = (dst.w + 3) / 4;
    const uint32_t inv_nquads = 0xFFFFFFFFu / (uint32_t)nquads + 1u;
    const uint32_t items = (uint32_t)nquads * (uint32_t)batch;   // < 2^32 / nquads (checked by the caller's plan)
    // actual ratios of this level pair (level sizes are rounded, so they can exceed the scale factor a little)
    const double rx = (double)src.w / dst.w, ry = (double)src.h / dst.h;
    const int band = (int)std::ceil((RS_BAND - 1) * ry) + 2 <= RS_NR ? RS_BAND : 4;   // 4 rows: ratios up to 4.6
    const bool wide = (int)std::ceil(3.0 * rx) + 1 > 7;
    dim3 blk(256), grd((items + 255) / 256, (dst.h + band - 1) / band);
    if (wide)
        hipLaunchKernelGGL(resize_level_kernel<true>, grd, blk, 0, st, src_base, src_img_stride, src_pitch, pyr, pyr_stride, src,
                           dst, xofs, xab, yofs, yab, batch, nquads, inv_nquads, band);
    else
        hipLaunchKernelGGL(resize_level_kernel<false>, grd, blk, 0, st, src_base, src_img_stride, src_pitch, pyr, pyr_stride, src,
                           dst, xofs, xab, yofs, yab, batch, nquads, inv_nquads, band);
}

void launch_pyramid_fused(const uint8_t *img0, size_t img0_stride, int pitch0, uint8_t *pyr, size_t pyr_stride,
                          const LevelDev *levels, int nlevels, const int4 *tile_x, const int4 *tile_y, int ntx, int nty,
                          const int *xofs, const int *xab, const int *yofs, const int *yab, int buf_pitch, int buf_rows,
                          size_t lds_bytes, int batch, hipStream_t st)
{
    hipLaunchKernelGGL(pyramid_fused_kernel, dim3(ntx * nty, batch), dim3(256), lds_bytes, st, img0, img0_stride, pitch0, pyr,
                       pyr_stride, levels, nlevels, tile_x, tile_y, ntx, nty, xofs, xab, yofs, yab, buf_pitch, buf_rows, batch);
}

void launch_fast(const uint8_t *img0, size_t img0_stride, int pitch0, const uint8_t *pyr, size_t pyr_stride,
                 const LevelDev * /*levels: the cell records carry their level's plane offset and pitch*/, const CellDev *cells, int n_cells,
                 int ini_th, int min_th, int TP, int TH, int SP, size_t lds_bytes, int list_cap, int keep_cap,
                 uint32_t *slots, size_t slot_stride, int32_t *cell_cnt, int batch, hipStream_t st)
{
    dim3 blk(64), grd((batch + 7) & ~7, n_cells);
    hipLaunchKernelGGL(fast_cells_kernel, grd, blk, lds_bytes, st, img0, img0_stride, pitch0, pyr, pyr_stride, cells, n_cells, ini_th,
                       min_th, TP, TH, SP, slots, slot_stride, cell_cnt, list_cap, keep_cap, batch);
}

void launch_compact(const CellDev *cells, int n_cells, int n_levels, const int *level_cell_begin,
                    const uint32_t *slots, size_t slot_stride, const int32_t *cell_cnt, uint32_t *dense,
                    size_t dense_stride, int32_t *level_off, int batch, hipStream_t st)
{
    hipLaunchKernelGGL(compact_candidates_kernel, dim3(batch), dim3(256), 0, st, cells, n_cells, n_levels,
                       level_cell_begin, slots, slot_stride, cell_cnt, dense, dense_stride, level_off);
}

void launch_octree(uint32_t *dense, size_t dense_stride, const OctGather &gather, const LevelDev *levels,
                   int n_levels, int batch, const OctDevScratch &scr, uint32_t *sel, size_t sel_stride,
                   int32_t *sel_level_cnt, int cap_level, int lds_bytes, hipStream_t st)
{
    const int jobs = batch * n_levels;
    // levels whose jobs keep 4 waves: the two largest for a few frames per call (a level-0 job 116 -> 101 us: the three
    // partition passes 58 -> 44 us); for batches the jobs of a launch fill the device anyway and idle helper waves only take
    // slots from the kernels of the other streams (measured: -0.012 ms on the stage, nothing on the step).
    // AOS2_OCT_GROUP_LEVELS overrides (tests).
    const char *gv = getenv("AOS2_OCT_GROUP_LEVELS");
    const int group_env = gv ? atoi(gv) : -1;
    const int group_levels = group_env >= 0 ? group_env : (batch < 8 ? 2 : 0);
    hipLaunchKernelGGL(octree_kernel, dim3(jobs), dim3(256), (size_t)lds_bytes, st, dense, dense_stride, gather,
                       levels, n_levels, batch, scr, sel, sel_stride, sel_level_cnt, cap_level, lds_bytes, group_levels);
}

int prepare_octree_image_kernel(int total_lds)
{
    return (int)hipFuncSetAttribute((const void *)octree_image_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, total_lds);
}

void launch_octree_image(uint32_t *dense, size_t dense_stride, const OctGather &gather, const LevelDev *levels,
                         int n_levels, int batch, const OctDevScratch &scr, uint32_t *sel, size_t sel_stride,
                         int32_t *sel_level_cnt, int cap_level, const OctImageLayout &lay, hipStream_t st)
{
    hipLaunchKernelGGL(octree_image_kernel, dim3(batch), dim3(64 * n_levels), (size_t)lay.total, st, dense, dense_stride,
                       gather, levels, n_levels, scr, sel, sel_stride, sel_level_cnt, cap_level, lay);
}

int prepare_octree_pair_kernel(int total_lds)
{
    return (int)hipFuncSetAttribute((const void *)octree_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, total_lds);
}

void launch_octree_pairs(uint32_t *dense, size_t dense_stride, const OctGather &gather, const LevelDev *levels,
                         int n_levels, int batch, const OctDevScratch &scr, uint32_t *sel, size_t sel_stride,
                         int32_t *sel_level_cnt, int cap_level, const OctImageLayout &lay, hipStream_t st)
{
    hipLaunchKernelGGL(octree_pair_kernel, dim3(batch * ((n_levels + 1) / 2)), dim3(128), (size_t)lay.total, st, dense, dense_stride,
                       gather, levels, n_levels, batch, scr, sel, sel_stride, sel_level_cnt, cap_level, lay);
}

// blurred planes of one image: level 0 first (pitch = width rounded up to 16), then the levels >= 1 at their pyramid offsets
size_t blur_plan(const LevelDev *h_levels, int n_levels, size_t pyr_bytes, BlurPlanHost *out)
{
    const size_t l0 = (size_t)((h_levels[0].w + 15) & ~15) * h_levels[0].h;
    const size_t l0_bytes = (l0 + 255) & ~(size_t)255;
    int first = 0;
    for (int l = 0; l < n_levels; ++l) {
        out->first[l] = first;
        out->nq[l] = (h_levels[l].w + 3) / 4;
        out->nq_in[l] = std::max(0, std::min(out->nq[l] - 1, (h_levels[l].w - 9) / 4));   // quads q >= 1 with 4 q + 8 < w
        first += out->nq[l] * ((h_levels[l].h + BL_ROWS - 1) / BL_ROWS);
        out->dst_off[l] = l == 0 ? 0u : (uint32_t)(l0_bytes + h_levels[l].off);
        out->dst_pitch[l] = l == 0 ? (h_levels[0].w + 15) & ~15 : h_levels[l].pitch;
    }
    out->first[n_levels] = first;
    return l0_bytes + pyr_bytes;
}

static BlurPlan to_dev(const BlurPlanHost &h)
{
    BlurPlan p;
    for (int i = 0; i < 9; ++i) p.first[i] = h.first[i];
    for (int i = 0; i < 8; ++i) {
        p.nq[i] = h.nq[i]; p.nq_in[i] = h.nq_in[i]; p.dst_off[i] = h.dst_off[i]; p.dst_pitch[i] = h.dst_pitch[i];
    }
    return p;
}

void launch_blur_levels(const uint8_t *img0, size_t img0_stride, int pitch0, const uint8_t *pyr, size_t pyr_stride,
                        const LevelDev *levels, int n_levels, const BlurPlanHost &plan, uint8_t *blur, size_t blur_stride, int batch,
                        hipStream_t st)
{
    dim3 grd((plan.first[n_levels] + 255) / 256, batch);
    hipLaunchKernelGGL(blur_levels_kernel, grd, dim3(256), 0, st, img0, img0_stride, pitch0, pyr, pyr_stride, levels, n_levels, to_dev(plan),
                       blur, blur_stride);
}

void launch_describe_blur(const uint8_t *img0, size_t img0_stride, int pitch0, const uint8_t *pyr, size_t pyr_stride,
                          const uint8_t *blur, size_t blur_stride, const BlurPlanHost &plan, const LevelDev *levels, int n_levels,
                          const uint32_t *sel, size_t sel_stride, int cap_level, const int32_t *sel_level_cnt,
                          aos2_keypoint_t *kps, uint8_t *desc, int cap, int32_t *n_out, int batch, int32_t *status, hipStream_t st)
{
    dim3 blk(64), grd((batch + 7) & ~7, (cap + DKB - 1) / DKB);
    hipLaunchKernelGGL(describe_blur_kernel, grd, blk, 0, st, img0, img0_stride, pitch0, pyr, pyr_stride, blur, blur_stride, to_dev(plan),
                       levels, n_levels, sel, sel_stride, cap_level, sel_level_cnt, kps, desc, cap, n_out, status, batch);
}

void launch_describe(const uint8_t *img0, size_t img0_stride, int pitch0, const uint8_t *pyr, size_t pyr_stride,
                     const LevelDev *levels, int n_levels,
                     const uint32_t *sel, size_t sel_stride, int cap_level, const int32_t *sel_level_cnt,
                     aos2_keypoint_t *kps, uint8_t *desc, int cap, int32_t *n_out, int batch,
                     unsigned long long umax_nibbles, int32_t *status, hipStream_t st)
{
    // waves of DK_FEW keypoints while they all fit on the chip at once (256 CUs x 28 waves), of DK_BATCH beyond
    const bool few = (long long)batch * ((cap + DK_FEW - 1) / DK_FEW) <= 2 * 7168;
    const int dk = few ? DK_FEW : DK_BATCH, groups = (cap + dk - 1) / dk;
    dim3 blk(64), grd((unsigned)(((batch + 7) & ~7) * groups));
    if (few)
        hipLaunchKernelGGL(describe_kernel<DK_FEW>, grd, blk, 0, st, img0, img0_stride, pitch0, pyr, pyr_stride, levels, n_levels, sel,
                           sel_stride, cap_level, sel_level_cnt, kps, desc, cap, n_out, umax_nibbles, status, batch, groups);
    else
        hipLaunchKernelGGL(describe_kernel<DK_BATCH>, grd, blk, 0, st, img0, img0_stride, pitch0, pyr, pyr_stride, levels, n_levels, sel,
                           sel_stride, cap_level, sel_level_cnt, kps, desc, cap, n_out, umax_nibbles, status, batch, groups);
}

}  // namespace aos2
