// Frame::ComputeStereoMatches (reference src/Frame.cc:495-669) on gfx950.
//
// Every left keypoint is independent until the final median cull, so the kernel is one wave per left
// keypoint: the 64 lanes sweep the right keypoints (row band :505-523, octave gate :557, disparity
// window :562, 256-bit Hamming :565), a DPP min picks the best (dist, iR) pair, and the same wave
// does the 11x11 SAD slide over 11 offsets (:586-620) from a 11x21 right window staged in LDS,
// the parabola fit (:625-633) and the depth (:636-648).  A second kernel (one workgroup per image)
// finds the median SAD by rank counting and applies thDist = 1.5*1.4*median (:654-668).
// The row table vRowIndices of the reference is replaced by evaluating its membership predicate
// (floor(y-r) <= row <= ceil(y+r)) directly: candidates are visited in ascending iR either way,
// and the key dist<<20|iR keeps the reference's first-strictly-smaller tie rule.
#include "stereo.h"

#include "aos2_common.h"

namespace aos2 {
namespace {

constexpr int TH_HIGH = 100, TH_LOW = 50;  // src/ORBmatcher.cc:37-38
constexpr uint32_t KEY_NONE = 0xFFFFFFFFu;

template <int kCtrl>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, kCtrl, 0xf, 0xf, false);
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t k)
{
    k = min(k, dpp_u32<0xB1>(k));
    k = min(k, dpp_u32<0x4E>(k));
    k = min(k, dpp_u32<0x141>(k));
    k = min(k, dpp_u32<0x140>(k));
    uint32_t a = __builtin_amdgcn_readlane(k, 0);
    a = min(a, (uint32_t)__builtin_amdgcn_readlane(k, 16));
    a = min(a, (uint32_t)__builtin_amdgcn_readlane(k, 32));
    a = min(a, (uint32_t)__builtin_amdgcn_readlane(k, 48));
    return a;
}

__device__ __forceinline__ int wave_sum_i32(int v)
{
    v += (int)dpp_u32<0xB1>((uint32_t)v);
    v += (int)dpp_u32<0x4E>((uint32_t)v);
    v += (int)dpp_u32<0x141>((uint32_t)v);
    v += (int)dpp_u32<0x140>((uint32_t)v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

__device__ __forceinline__ const uint8_t *level_plane(const PyrView &P, int image, int level, int &pitch)
{
    if (level == 0) {
        pitch = P.pitch0;
        return P.img0 + (size_t)image * P.img0_stride;
    }
    pitch = P.pitch[level];
    return P.pyr + (size_t)image * P.pyr_bytes + P.off[level];
}

constexpr int kW = 5, kL = 5;            // :587-588
constexpr int kWin = 2 * (kW + kL) + 1;  // 21 columns of the right window
constexpr int kWinPitch = 24;

// vRowIndices (:503-522), one workgroup per image: rows -> right keypoints whose vertical band covers the row.
// (the reference appends in iR order; the order inside a row is irrelevant here because the best candidate is the
// minimum of dist << 20 | iR)
__global__ __launch_bounds__(256) void stereo_rows_kernel(const StereoArgs A)
{
    extern __shared__ int rcnt[];   // rows counters, then cursors
    __shared__ int part[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nR = A.n_r[b], rows = A.rows;
    const size_t base = (size_t)b * A.cap;
    for (int y = tid; y < rows; y += 256) rcnt[y] = 0;
    __syncthreads();
    auto band = [&](int i, int &minr, int &maxr) {
        const aos2_keypoint_t *kr = A.kp_r + base + i;
        const float kpY = kr->y;
        const float r = __fmul_rn(2.0f, A.L.scale[kr->octave]);  // :512 (mvScaleFactors = the left extractor's)
        maxr = min((int)ceilf(__fadd_rn(kpY, r)), rows - 1);
        minr = max((int)floorf(__fsub_rn(kpY, r)), 0);
    };
    for (int i = tid; i < nR; i += 256) {
        int minr, maxr;
        band(i, minr, maxr);
        for (int y = minr; y <= maxr; ++y) atomicAdd(&rcnt[y], 1);
    }
    __syncthreads();
    // exclusive scan over the rows: per-thread segment sums, block scan of the 256 partial sums
    const int seg = (rows + 255) / 256, y0 = tid * seg, y1 = min(y0 + seg, rows);
    int ssum = 0;
    for (int y = y0; y < y1; ++y) ssum += rcnt[y];
    part[tid] = ssum;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int t = 0; t < 256; ++t) {
            const int v = part[t];
            part[t] = acc;
            acc += v;
        }
    }
    __syncthreads();
    int32_t *ro = A.row_off + (size_t)b * (rows + 1);
    int acc = part[tid];
    for (int y = y0; y < y1; ++y) {
        const int v = rcnt[y];
        ro[y] = acc;
        rcnt[y] = acc;   // cursor
        acc += v;
    }
    if (y1 == rows && y0 < rows) ro[rows] = acc;
    if (rows == 0 && tid == 0) ro[0] = 0;
    __syncthreads();
    int32_t *ri = A.row_idx + (size_t)b * A.row_cap;
    for (int i = tid; i < nR; i += 256) {
        int minr, maxr;
        band(i, minr, maxr);
        for (int y = minr; y <= maxr; ++y) {
            const int pos = atomicAdd(&rcnt[y], 1);
            if (pos < A.row_cap) ri[pos] = i;
        }
    }
}

__global__ __launch_bounds__(256) void stereo_match_kernel(const StereoArgs A)
{
    __shared__ uint8_t win[4][(2 * kW + 1) * kWinPitch];
    // image pair = blockIdx.x (padded to a multiple of 8): pair b runs on XCD b % 8, whose L2 keeps both pyramids
    const int b = blockIdx.x;
    if (b >= A.batch) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int iL = blockIdx.y * 4 + wave;
    const int nL = A.n_l[b];
    if (iL >= nL) return;
    const size_t base = (size_t)b * A.cap;
    float out_u = -1.0f, out_d = -1.0f;
    int out_sad = -1;

    const aos2_keypoint_t kpL = A.kp_l[base + iL];
    const int levelL = kpL.octave;
    const float vL = kpL.y, uL = kpL.x;
    const int row = (int)vL;
    const float minZ = A.mb, minD = 0.0f, maxD = __fdiv_rn(A.mbf, minZ);  // :526-528
    const float minU = __fsub_rn(uL, maxD), maxU = __fsub_rn(uL, minD);
    uint32_t key = KEY_NONE;
    if (!(maxU < 0.0f)) {
        const uint4 *dl = reinterpret_cast<const uint4 *>(A.desc_l + (base + iL) * 32);
        const uint4 a0 = dl[0], a1 = dl[1];
        // candidates = vRowIndices[(int)vL] (:546-548); the row test itself is the table's construction
        const int32_t *ro = A.row_off + (size_t)b * (A.rows + 1);
        const int rowc = min(max(row, 0), A.rows - 1);
        const int cbeg = ro[rowc], cend = min(ro[rowc + 1], A.row_cap);
        const int32_t *ri = A.row_idx + (size_t)b * A.row_cap;
        for (int c = cbeg + lane; c < cend; c += 64) {
            const int i = ri[c];
            const aos2_keypoint_t *kr = A.kp_r + base + i;
            const int oct = kr->octave;
            if (oct < levelL - 1 || oct > levelL + 1) continue;  // :557
            const float uR = kr->x;
            if (!(uR >= minU && uR <= maxU)) continue;  // :562
            const uint4 *dr = reinterpret_cast<const uint4 *>(A.desc_r + (base + i) * 32);
            const uint4 b0 = dr[0], b1 = dr[1];
            const int d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                          __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
            const uint32_t k = ((uint32_t)d << 20) | (uint32_t)i;
            key = min(key, k);
        }
    }
    key = wave_min_u32(key);
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;  // :499
    const int bestDist = key == KEY_NONE ? TH_HIGH : (int)(key >> 20);
    if (bestDist < TH_HIGH && bestDist < thOrbDist) {  // :575
        const int bestIdxR = (int)(key & 0xFFFFFu);
        const float uR0 = A.kp_r[base + bestIdxR].x;
        const float scaleFactor = A.L.inv_scale[levelL];
        const float scaleduL = roundf(__fmul_rn(kpL.x, scaleFactor));
        const float scaledvL = roundf(__fmul_rn(kpL.y, scaleFactor));
        const float scaleduR0 = roundf(__fmul_rn(uR0, scaleFactor));
        const float iniu = __fsub_rn(__fadd_rn(scaleduR0, (float)kL), (float)kW);
        const float endu = __fadd_rn(__fadd_rn(__fadd_rn(scaleduR0, (float)kL), (float)kW), 1.0f);
        const int cols = A.R.w[levelL];  // :606
        if (!(iniu < 0.0f || endu >= (float)cols)) {
            int pitchL, pitchR;
            const uint8_t *PL = level_plane(A.L, A.first_image_l + b, levelL, pitchL);
            const uint8_t *PR = level_plane(A.R, A.first_image_r + b, levelL, pitchR);
            const int cuL = (int)scaleduL, cvL = (int)scaledvL, cuR0 = (int)scaleduR0;
            // right window rows cvL-5..cvL+5, columns cuR0-10..cuR0+10 -> LDS
            uint8_t *wv = win[wave];
            for (int p = lane; p < (2 * kW + 1) * kWin; p += 64) {
                const int ry = p / kWin, rx = p - ry * kWin;
                wv[ry * kWinPitch + rx] = PR[(size_t)(cvL - kW + ry) * pitchR + cuR0 - kW - kL + rx];
            }
            // this lane's pixels of the left patch (centre subtracted, :592-594)
            const int cL = PL[(size_t)cvL * pitchL + cuL];
            const int p0 = lane, p1 = lane + 64;
            const int y0 = p0 / 11, x0 = p0 - y0 * 11;
            const int y1 = p1 / 11, x1 = p1 - y1 * 11;
            const bool has1 = p1 < 121;
            const int il0 = (int)PL[(size_t)(cvL - kW + y0) * pitchL + cuL - kW + x0] - cL;
            const int il1 = has1 ? (int)PL[(size_t)(cvL - kW + y1) * pitchL + cuL - kW + x1] - cL : 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wave: LDS writes are in order
            int dist[2 * kL + 1];
#pragma unroll
            for (int inc = 0; inc <= 2 * kL; ++inc) {  // incR = inc - L
                const int cR = wv[kW * kWinPitch + kW + inc];
                int s = abs(il0 - ((int)wv[y0 * kWinPitch + x0 + inc] - cR));
                if (has1) s += abs(il1 - ((int)wv[y1 * kWinPitch + x1 + inc] - cR));
                dist[inc] = wave_sum_i32(s);
            }
            int bestS = dist[0], bestInc = 0;  // first strict minimum (:611-615)
#pragma unroll
            for (int inc = 1; inc <= 2 * kL; ++inc)
                if (dist[inc] < bestS) {
                    bestS = dist[inc];
                    bestInc = inc;
                }
            if (bestInc != 0 && bestInc != 2 * kL) {  // :620
                float d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
                for (int inc = 1; inc < 2 * kL; ++inc)
                    if (inc == bestInc) {
                        d1 = (float)dist[inc - 1];
                        d2 = (float)dist[inc];
                        d3 = (float)dist[inc + 1];
                    }
                const float den = __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2)));
                const float deltaR = __fdiv_rn(__fsub_rn(d1, d3), den);  // :629
                if (!(deltaR < -1.0f || deltaR > 1.0f)) {
                    float bestuR = __fmul_rn(A.L.scale[levelL],
                                             __fadd_rn(__fadd_rn(scaleduR0, (float)(bestInc - kL)), deltaR));
                    float disparity = __fsub_rn(uL, bestuR);
                    if (disparity >= minD && disparity < maxD) {
                        if (disparity <= 0.0f) {
                            disparity = 0.01f;
                            bestuR = __fsub_rn(uL, 0.01f);
                        }
                        out_d = __fdiv_rn(A.mbf, disparity);
                        out_u = bestuR;
                        out_sad = bestS;
                    }
                }
            }
        }
    }
    if (lane == 0) {
        A.u_right[base + iL] = out_u;
        A.depth[base + iL] = out_d;
        A.sad[base + iL] = out_sad;
    }
}

// median cull (:654-668): one workgroup per image
__global__ __launch_bounds__(256) void stereo_cull_kernel(const StereoArgs A)
{
    extern __shared__ int s_sad[];
    __shared__ int s_median, s_nv;
    const int b = blockIdx.x;
    const int n = A.n_l[b];
    const size_t base = (size_t)b * A.cap;
    if (threadIdx.x == 0) {
        s_nv = 0;
        s_median = -1;
    }
    __syncthreads();
    int cnt = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int v = A.sad[base + i];
        s_sad[i] = v;
        cnt += v >= 0;
    }
    atomicAdd(&s_nv, cnt);
    __syncthreads();
    const int nv = s_nv;
    if (nv == 0) return;
    const int k = nv / 2;  // vDistIdx[size/2] after the sort
    for (int i = threadIdx.x; i < n; i += 256) {
        const int v = s_sad[i];
        if (v < 0) continue;
        int less = 0, eq = 0;
        for (int j = 0; j < n; ++j) {
            const int u = s_sad[j];
            less += (u >= 0) & (u < v);
            eq += u == v;
        }
        if (less <= k && k < less + eq) s_median = v;  // every writer stores the same value
    }
    __syncthreads();
    const float median = (float)s_median;
    const float thDist = __fmul_rn(__fmul_rn(1.5f, 1.4f), median);
    for (int i = threadIdx.x; i < n; i += 256) {
        const int v = s_sad[i];
        if (v >= 0 && !((float)v < thDist)) {
            A.u_right[base + i] = -1.0f;
            A.depth[base + i] = -1.0f;
        }
    }
}

}  // namespace

int stereo_row_span(const PyrView &v)
{
    float smax = 1.0f;
    for (int l = 0; l < v.nlevels; ++l) smax = v.scale[l] > smax ? v.scale[l] : smax;
    return 2 * (int)ceilf(2.0f * smax) + 3;
}

int launch_stereo(const StereoArgs &a, int max_n_left, hipStream_t stream)
{
    if (a.rows > 12000) {  // row counters live in LDS
        set_error("ComputeStereoMatches: images with more than 12000 rows");
        return AOS2_ERR_CAPACITY;
    }
    if (max_n_left <= 0 || a.batch <= 0) return AOS2_OK;
    if (max_n_left > 15360) {  // cull kernel keeps one int per left keypoint in LDS
        set_error("ComputeStereoMatches: more than 15360 left keypoints per image");
        return AOS2_ERR_CAPACITY;
    }
    hipLaunchKernelGGL(stereo_rows_kernel, dim3(a.batch), dim3(256), sizeof(int) * (size_t)(a.rows + 1), stream, a);
    AOS2_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(stereo_match_kernel, dim3((a.batch + 7) & ~7, (max_n_left + 3) / 4), dim3(256), 0, stream, a);
    AOS2_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(stereo_cull_kernel, dim3(a.batch), dim3(256), sizeof(int) * (size_t)max_n_left, stream, a);
    AOS2_HIP_CHECK(hipGetLastError());
    return AOS2_OK;
}

}  // namespace aos2
