// Host side of the ORB extractor: plan construction (level sizes, grid cells, resize tables),
// device buffers, stage orchestration on one HIP stream, and the C ABI of include/aos2.h.
// Reference: src/ORBextractor.cc (ctor :410-470, operator() :1043-1105, ComputePyramid :1107-1132,
// ComputeKeyPointsOctTree :765-853).
#include <atomic>
#include <chrono>
#include <cmath>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "aos2_common.h"
#include "extractor_kernels.h"
#include "octree.h"
#include "stereo.h"

namespace aos2 {

static thread_local std::string g_err;
void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

int bind_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device available (%s); this library has no CPU fallback",
                  e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return AOS2_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("HIP device %d out of range (0..%d)", device, n - 1);
        return AOS2_ERR_NO_DEVICE;
    }
    AOS2_HIP_CHECK(hipSetDevice(device));
    return AOS2_OK;
}

static const int8_t k_pattern[1024] = {
#include "orb_pattern.inc"
};

constexpr int kEdge = 19;       // EDGE_THRESHOLD src/ORBextractor.cc:74
constexpr int kPatch = 31;      // PATCH_SIZE :72
constexpr int kHalfPatch = 15;  // HALF_PATCH_SIZE :73
constexpr int kMaxLevels = 16;
constexpr int kMaxStreams = 8;

struct Plan {
    int w = 0, h = 0;
    std::vector<LevelDev> levels;
    std::vector<CellDev> cells;
    std::vector<int> level_cell_begin;
    std::vector<int> xofs, xab, yofs, yab;
    size_t pyr_bytes = 0;      // one image's pyramid block
    size_t oct_cand_total = 0, oct_node_total = 0;  // octree scratch per image
    size_t slot_total = 0;     // candidate slots per image
    int max_cw = 0, max_ch = 0;
    int TP = 0, TH = 0, SP = 0;
    int list_cap = 0, keep_cap = 0;
    size_t fast_lds = 0;
    // one-launch pyramid (pyramid_fused_kernel): per (level, tile column / row) {own0, own1, need0, need1}
    std::vector<int4> tile_x, tile_y;
    int ntx = 0, nty = 0, pyr_buf_pitch = 0, pyr_buf_rows = 0;
    size_t pyr_lds = 0;
    bool pyr_fused = false;
    DevBuf<int4> d_tile_x, d_tile_y;
    // device copies
    DevBuf<LevelDev> d_levels;
    DevBuf<CellDev> d_cells;
    DevBuf<int> d_level_cell_begin, d_xofs, d_xab, d_yofs, d_yab;
    void release_device()
    {
        d_levels.release(); d_cells.release(); d_level_cell_begin.release();
        d_xofs.release(); d_xab.release(); d_yofs.release(); d_yab.release();
        d_tile_x.release(); d_tile_y.release();
    }
};

}  // namespace aos2

using namespace aos2;

struct aos2_extractor {
    int nfeatures, nlevels, iniTh, minTh, device;
    float scaleFactor;
    float mvScaleFactor[kMaxLevels], mvInvScaleFactor[kMaxLevels];
    float mvLevelSigma2[kMaxLevels], mvInvLevelSigma2[kMaxLevels];
    int mnFeaturesPerLevel[kMaxLevels];
    int umax[16];
    int gauss7[7];
    int cap_level = 0, max_kp = 0;
    unsigned long long umax_nibbles = 0;
    bool host_octree = false;
    int oct_lds = 0;                     // LDS bytes per octree job (0 = global-scratch path only)
    OctImageLayout oct_image = {};       // total > 0: one workgroup per image with per-level LDS slices
    OctImageLayout oct_pair = {};        // total > 0: two levels per workgroup for batches of >= 8 images (extractor_kernels.h)
    int host_threads = 8;

    bool dev_ready = false;
    hipStream_t stream = nullptr;       // = streams[0]
    hipStream_t streams[kMaxStreams] = {};
    int chunks = 0;                      // 0 = automatic
    int n_streams = 0;
    hipEvent_t ev[8] = {};
    hipEvent_t order_ev[kMaxStreams] = {};   // aos2_extractor_stream_wait
    // ComputeStereoMatches reads BOTH extractors' pyramid blocks on the left one's first stream: each extractor keeps an event behind
    // those kernels, and its next batch waits for it on every chunk stream before it rewrites the pyramids (stereo_guard_armed)
    hipEvent_t stereo_guard = nullptr, stereo_t0 = nullptr, stereo_t1 = nullptr;
    hipEvent_t input_ev = nullptr, input_fan_ev = nullptr;   // aos2_extractor_wait_for_stream
    int input_waited = 0;   // streams that wait for the inputs announced since the last batch (0 = none announced)
    int last_chunks = 1;    // chunk streams of the last batch
    bool stereo_guard_armed = false;
    bool stereo_guard_captured = false;       // the guard was recorded while its stream was being captured (aos2_capture_begin)
    hipStream_t stereo_guard_stream = nullptr;   // ... on this stream (the LEFT extractor's: see stereo_peer)
    aos2_extractor *stereo_peer = nullptr;       // the other eye of the last ComputeStereoMatches: told when this handle goes away
    int streams_used = 0;                    // streams the batches since the last wait ran on (<= chunks)
    Plan plan;
    int batch_cap = 0;
    int last_batch = 0;
    // asynchronous batches (aos2_extractor_extract_batch_device_async): enqueued, not yet waited for
    int in_flight = 0, flight_cap = 0;
    const int32_t *flight_nout = nullptr;   // d_n_out of the last enqueued batch
    std::chrono::steady_clock::time_point t_enqueue;
    DevBuf<int32_t> d_status;   // sticky [lowest octree failure code, largest n_out] of the batches in flight
    const uint8_t *img0 = nullptr;  // level 0 of the last batch = the caller's (device) images
    size_t img0_stride = 0;
    // AOS2_DESC_BLUR=level: the reference's whole-level GaussianBlur as a streaming pass, describe on the blurred planes
    bool blur_level = false;
    BlurPlanHost blur_plan_h = {};
    size_t blur_bytes = 0;           // per image
    DevBuf<uint8_t> d_blur;
    int pitch0 = 0;
    DevBuf<uint8_t> d_pyr, d_in, d_desc;
    DevBuf<uint32_t> d_slots, d_dense, d_sel;
    DevBuf<int32_t> d_cell_cnt, d_level_off, d_level_cnt, d_sel_cnt, d_nout;
    DevBuf<aos2_keypoint_t> d_kps;
    int out_cap = 0;
    // ComputeStereoMatches scratch (this handle = the left eye)
    DevBuf<int32_t> st_sad, st_rows;
    DevBuf<uint8_t> st_io;
    PinnedBuf<uint8_t> st_host;
    float stereo_ms = 0;
    // device octree scratch
    DevBuf<int16_t> o_xs, o_ys;
    DevBuf<uint8_t> o_sc;
    DevBuf<int32_t> o_perm, o_tmp, o_pairs, o_idx;
    DevBuf<OctNode> o_nodes;
    // host mirrors
    PinnedBuf<int32_t> h_level_off, h_sel_cnt, h_nout, h_status;
    PinnedBuf<uint32_t> h_dense, h_sel;
    float timing[8] = {};
};

namespace aos2 {

static int cv_round_f(float v) { return (int)lrintf(v); }

static void build_host_tables(aos2_extractor *e)
{
    // scale tables :415-432
    e->mvScaleFactor[0] = 1.0f;
    e->mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < e->nlevels; i++) {
        e->mvScaleFactor[i] = e->mvScaleFactor[i - 1] * e->scaleFactor;
        e->mvLevelSigma2[i] = e->mvScaleFactor[i] * e->mvScaleFactor[i];
    }
    for (int i = 0; i < e->nlevels; i++) {
        e->mvInvScaleFactor[i] = 1.0f / e->mvScaleFactor[i];
        e->mvInvLevelSigma2[i] = 1.0f / e->mvLevelSigma2[i];
    }
    // per-level quotas :436-448
    const float factor = 1.0f / e->scaleFactor;
    float nDesired = e->nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)e->nlevels));
    int sum = 0;
    for (int level = 0; level < e->nlevels - 1; level++) {
        e->mnFeaturesPerLevel[level] = cv_round_f(nDesired);
        sum += e->mnFeaturesPerLevel[level];
        nDesired *= factor;
    }
    e->mnFeaturesPerLevel[e->nlevels - 1] = std::max(e->nfeatures - sum, 0);
    // circular patch extents :454-470
    const int vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    const int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (int v = 0; v <= vmax; ++v) e->umax[v] = (int)lrint(std::sqrt(hp2 - v * v));
    for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }
    // cv::getGaussianKernel(7, 2, CV_32F) -> 8 fractional bits (createSeparableLinearFilter 8U path)
    {
        float cf[7];
        double s = 0;
        for (int i = 0; i < 7; ++i) {
            const double x = i - 3.0;
            cf[i] = (float)std::exp(-0.5 / 4.0 * x * x);
            s += cf[i];
        }
        s = 1. / s;
        for (int i = 0; i < 7; ++i) {
            cf[i] = (float)(cf[i] * s);
            e->gauss7[i] = (int)lrint((double)cf[i] * 256.0);
        }
    }
    // capacities for a "normal" aspect ratio; build_plan() raises them for the actual image size (keypoint_bounds)
    {
        int cl = 0, tot = 0;
        for (int l = 0; l < e->nlevels; ++l) {
            cl = std::max(cl, e->mnFeaturesPerLevel[l] + 4);
            tot += e->mnFeaturesPerLevel[l] + 3;
        }
        e->cap_level = cl;
        e->max_kp = tot;
    }
    e->umax_nibbles = 0;
    for (int v = 0; v < 16; ++v) e->umax_nibbles |= (unsigned long long)(e->umax[v] & 15) << (4 * v);
}

// Upper bound of DistributeOctTree's output per level for a w x h image (:539-763): the loop stops at >= N leaves with at
// most 3 extra from the last divide, EXCEPT that its first pass divides all nIni = round(W / H) root nodes unconditionally
// (:549-590), which alone can leave 4 * nIni leaves -- more than N + 3 for wide images with few features
// (found by tools/gpu_fuzz_extractor.py: 838 x 118, nfeatures 100 -> 123 keypoints).
static void keypoint_bounds(const aos2_extractor *e, int w, int h, int *cap_level, int *max_kp)
{
    int cl = 0, tot = 0;
    for (int l = 0; l < e->nlevels; ++l) {
        int b = e->mnFeaturesPerLevel[l] + 3;
        if (w > 0 && h > 0) {
            const float s = e->mvInvScaleFactor[l];
            const int lw = (int)lrintf((float)w * s), lh = (int)lrintf((float)h * s);
            if (lh - 32 > 0 && lw - 32 > 0) {
                const int nIni = (int)roundf((float)(lw - 32) / (float)(lh - 32));   // round(): half away from zero (:545)
                b = std::max(b, 4 * nIni);
            }
        }
        cl = std::max(cl, b + 1);
        tot += b;
    }
    *cap_level = cl;
    *max_kp = tot;
}

static short sat_short(float v)
{
    int i = (int)lrintf(v);
    return (short)std::min(32767, std::max(-32768, i));
}

// cv::resize() coefficient tables for src -> dst (INTER_LINEAR, 8U fixed point)
static void resize_tables(int sw, int sh, int dw, int dh, std::vector<int> &xofs, std::vector<int> &xab,
                          std::vector<int> &yofs, std::vector<int> &yab)
{
    const double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
    const double scale_x = 1. / inv_sx, scale_y = 1. / inv_sy;
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)std::floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        const short a0 = sat_short((1.f - fx) * 2048), a1 = sat_short(fx * 2048);
        xofs.push_back(sx);
        xab.push_back((int)((uint32_t)(uint16_t)a0 | ((uint32_t)(uint16_t)a1 << 16)));
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)std::floor(fy);
        fy -= sy;
        const short b0 = sat_short((1.f - fy) * 2048), b1 = sat_short(fy * 2048);
        yofs.push_back(sy);
        yab.push_back((int)((uint32_t)(uint16_t)b0 | ((uint32_t)(uint16_t)b1 << 16)));
    }
    // the kernel reads x tables as int4: pad every level's table to a multiple of 4 entries
    while (xofs.size() % 4) {
        xofs.push_back(xofs.back());
        xab.push_back(xab.back());
    }
    while (yofs.size() % 4) {
        yofs.push_back(yofs.back());
        yab.push_back(yab.back());
    }
}

static int build_plan(aos2_extractor *e, int w, int h)
{
    Plan &P = e->plan;
    if (P.w == w && P.h == h) return AOS2_OK;
    // smallest level must admit at least one 30-px cell in both directions (:783-786)
    {
        const float s = e->mvInvScaleFactor[e->nlevels - 1];
        const int lw = cv_round_f((float)w * s), lh = cv_round_f((float)h * s);
        if (lw - 32 < 30 || lh - 32 < 30) {
            set_error("image %dx%d too small for %d pyramid levels", w, h, e->nlevels);
            return AOS2_ERR_TOO_SMALL;
        }
        if (w > 4000 || h > 4000) {
            set_error("image %dx%d exceeds the 12-bit candidate packing", w, h);
            return AOS2_ERR_ARG;
        }
    }
    P.release_device();
    P = Plan();
    P.w = w;
    P.h = h;
    keypoint_bounds(e, w, h, &e->cap_level, &e->max_kp);
    size_t off = 0, slot = 0;
    for (int l = 0; l < e->nlevels; ++l) {
        LevelDev L{};
        const float s = e->mvInvScaleFactor[l];
        L.w = cv_round_f((float)w * s);   // :1112
        L.h = cv_round_f((float)h * s);
        L.pitch = (L.w + 4 + 15) & ~15;   // >= w+4 so 32-bit tile loads may overrun a row end
        L.off = off;
        off += ((size_t)L.pitch * (L.h + 1) + 255) & ~(size_t)255;
        L.nfeat = e->mnFeaturesPerLevel[l];
        L.scaled_patch = (int)(kPatch * e->mvScaleFactor[l]);
        L.scale = e->mvScaleFactor[l];
        L.tab_x = (int)P.xofs.size();
        L.tab_y = (int)P.yofs.size();
        if (l > 0)
            resize_tables(P.levels[l - 1].w, P.levels[l - 1].h, L.w, L.h, P.xofs, P.xab, P.yofs, P.yab);
        P.levels.push_back(L);
        // grid cells :768-806
        const int minBorderX = kEdge - 3, minBorderY = minBorderX;
        const int maxBorderX = L.w - kEdge + 3, maxBorderY = L.h - kEdge + 3;
        const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / 30.f), nRows = (int)(height / 30.f);
        const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
        P.level_cell_begin.push_back((int)P.cells.size());
        for (int i = 0; i < nRows; i++) {
            const int iniY = minBorderY + i * hCell;
            int maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = maxBorderY;
            for (int j = 0; j < nCols; j++) {
                const int iniX = minBorderX + j * wCell;
                int maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = maxBorderX;
                CellDev c{};
                c.level = (int16_t)l;
                c.vx0 = (int16_t)(iniX + 3);
                c.vy0 = (int16_t)(iniY + 3);
                c.cw = (int16_t)(maxX - iniX - 6);
                c.ch = (int16_t)(maxY - iniY - 6);
                if (c.cw <= 0 || c.ch <= 0) continue;  // sub-image < 7 px: cv::FAST evaluates nothing
                if (c.cw > 64) {
                    set_error("cell width %d > 64 unsupported", (int)c.cw);
                    return AOS2_ERR_ARG;
                }
                c.slot_off = (int32_t)slot;
                {
                    const uint32_t nq = (uint32_t)(c.cw + 3) / 4;
                    c.inv_nq = 65536u / nq + 1;
                    c.inv_ndw = 65536u / (nq + 2) + 1;
                    c.inv_n16 = 65536u / ((nq + 2 + 3) / 4) + 1;
                }
                if (L.off > 0xffffffffull || L.pitch > 0xffff) {
                    set_error("pyramid of a %dx%d image exceeds the 32-bit plane offsets", w, h);
                    return AOS2_ERR_ARG;
                }
                c.pitch = (uint16_t)L.pitch;
                c.plane_off = (uint32_t)L.off;
                slot += (size_t)((c.cw + 1) / 2) * ((c.ch + 1) / 2);
                P.max_cw = std::max<int>(P.max_cw, c.cw);
                P.max_ch = std::max<int>(P.max_ch, c.ch);
                P.cells.push_back(c);
            }
        }
    }
    P.pyr_bytes = off;
    P.slot_total = (slot + 63) & ~(size_t)63;
    for (int l = 0; l < e->nlevels; ++l) {
        LevelDev &L = P.levels[l];
        size_t cap = 0;
        const int c1 = l + 1 < e->nlevels ? P.level_cell_begin[l + 1] : (int)P.cells.size();
        for (int c = P.level_cell_begin[l]; c < c1; ++c) cap += (size_t)((P.cells[c].cw + 1) / 2) * ((P.cells[c].ch + 1) / 2);
        L.oct_cand_off = (int)P.oct_cand_total;
        L.oct_cand_cap = (int)cap;
        P.oct_cand_total += (cap + 63) & ~(size_t)63;
        L.oct_node_off = (int)P.oct_node_total;
        L.oct_node_cap = oct_max_nodes((int)cap, L.nfeat);
        P.oct_node_total += (size_t)L.oct_node_cap;
    }
    // LDS tile: [4-byte left halo | nq quads | 4-byte right halo] per row, evaluated column 0 at byte 4
    P.TP = (4 * ((P.max_cw + 3) / 4) + 8 + 15) & ~15;   // (a multiple of 16: the tile is staged 16 bytes per lane)
    P.TH = P.max_ch + 6;
    P.SP = (P.max_cw + 2 + 3) & ~3;
    // every pixel of every 4-px group may survive the pre-test (columns >= cw of the last group are
    // only dropped in phase 2), so size the list for whole groups
    P.list_cap = (4 * ((P.max_cw + 3) / 4) * P.max_ch + 7) & ~7;
    P.keep_cap = ((P.max_cw + 1) / 2) * ((P.max_ch + 1) / 2);          // NMS survivors are >= 2 px apart
    // The survivor list is bounded (typical cells produce 100-200 survivors); a cell that produces more is
    // scored in instalments.  The smaller LDS footprint doubles the waves per SIMD.  AOS2_FAST_LIST overrides
    // the bound (tests force the instalment path with a small value).
    int l1 = 768;
    if (const char *v = getenv("AOS2_FAST_LIST")) l1 = std::max(264, atoi(v));
    P.list_cap = std::min(P.list_cap, (l1 + 7) & ~7);
    P.fast_lds = (((size_t)P.TP * P.TH + 15) & ~(size_t)15) + (((size_t)P.SP * (P.TH - 4) + 15) & ~(size_t)15) +
                 (size_t)P.list_cap * 2 + 16;
    // upload
    int st;
    // ---- tiles of the one-launch pyramid.  A workgroup owns [B_k(i), B_k(i + 1)) of level k along each axis, B_k(i) =
    // the level-0 boundary 64 i divided by the level's scale (any monotone choice works); what it must COMPUTE at level k
    // is that plus the sources of what it computes at level k + 1 (read off the resize tables), from the top level down.
    {
        const int L = e->nlevels, TS = 64;
        P.ntx = (w + TS - 1) / TS;
        P.nty = (h + TS - 1) / TS;
        int maxw = 0, maxh = 0;
        auto axis = [&](bool is_x, int nt, std::vector<int4> &out, int &maxn) {
            out.assign((size_t)L * nt, int4{0, 0, 0, 0});
            for (int i = 0; i < nt; ++i) {
                int n0 = 0, n1 = 0;   // need of level k + 1
                for (int k = L - 1; k >= 0; --k) {
                    const int dim = is_x ? P.levels[k].w : P.levels[k].h;
                    auto bound = [&](int j) {
                        if (j >= nt) return dim;
                        return std::min(dim, (int)std::lround((double)(TS * j) / (double)e->mvScaleFactor[k]));
                    };
                    int o0 = bound(i), o1 = bound(i + 1);
                    if (k == 0) o0 = o1 = 0;   // level 0 is the caller's image: nothing to own
                    int c0 = o0, c1 = o1;
                    if (k + 1 < L && n1 > n0) {   // sources of the region of level k + 1
                        const std::vector<int> &tab = is_x ? P.xofs : P.yofs;
                        const int base = is_x ? P.levels[k + 1].tab_x : P.levels[k + 1].tab_y;
                        const int lo = std::min(std::max(tab[base + n0], 0), dim - 1);
                        const int hi = std::min(std::max(tab[base + n1 - 1] + 1, 0), dim - 1);
                        if (c1 > c0) {
                            c0 = std::min(c0, lo);
                            c1 = std::max(c1, hi + 1);
                        } else {
                            c0 = lo;
                            c1 = hi + 1;
                        }
                    }
                    out[(size_t)k * nt + i] = int4{o0, o1, c0, c1};
                    maxn = std::max(maxn, c1 - c0);
                    n0 = c0;
                    n1 = c1;
                }
            }
        };
        axis(true, P.ntx, P.tile_x, maxw);
        axis(false, P.nty, P.tile_y, maxh);
        P.pyr_buf_pitch = (maxw + 3) & ~3;
        P.pyr_buf_rows = maxh;
        P.pyr_lds = 2 * (size_t)P.pyr_buf_pitch * P.pyr_buf_rows + (size_t)(2 * P.pyr_buf_pitch + 2 * P.pyr_buf_rows) * sizeof(int) + 16;
        // steep pyramids (scale factor towards 2) need hundreds of pixels of halo at level 0: those keep one launch per level
        P.pyr_fused = L > 1 && P.pyr_lds <= 60 * 1024;
        if (const char *v = getenv("AOS2_PYRAMID")) P.pyr_fused = P.pyr_fused && strcmp(v, "levels") != 0;
    }
    if ((st = P.d_levels.alloc(P.levels.size()))) return st;
    if ((st = P.d_cells.alloc(P.cells.size()))) return st;
    if ((st = P.d_level_cell_begin.alloc(P.level_cell_begin.size()))) return st;
    if ((st = P.d_xofs.alloc(P.xofs.size()))) return st;
    if ((st = P.d_xab.alloc(P.xab.size()))) return st;
    if ((st = P.d_yofs.alloc(P.yofs.size()))) return st;
    if ((st = P.d_yab.alloc(P.yab.size()))) return st;
    AOS2_HIP_CHECK(hipMemcpy(P.d_levels.p, P.levels.data(), P.levels.size() * sizeof(LevelDev), hipMemcpyHostToDevice));
    AOS2_HIP_CHECK(hipMemcpy(P.d_cells.p, P.cells.data(), P.cells.size() * sizeof(CellDev), hipMemcpyHostToDevice));
    AOS2_HIP_CHECK(hipMemcpy(P.d_level_cell_begin.p, P.level_cell_begin.data(), P.level_cell_begin.size() * sizeof(int), hipMemcpyHostToDevice));
    if (!P.xofs.empty()) {
        AOS2_HIP_CHECK(hipMemcpy(P.d_xofs.p, P.xofs.data(), P.xofs.size() * sizeof(int), hipMemcpyHostToDevice));
        AOS2_HIP_CHECK(hipMemcpy(P.d_xab.p, P.xab.data(), P.xab.size() * sizeof(int), hipMemcpyHostToDevice));
        AOS2_HIP_CHECK(hipMemcpy(P.d_yofs.p, P.yofs.data(), P.yofs.size() * sizeof(int), hipMemcpyHostToDevice));
        AOS2_HIP_CHECK(hipMemcpy(P.d_yab.p, P.yab.data(), P.yab.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    if ((st = P.d_tile_x.alloc(P.tile_x.size()))) return st;
    if ((st = P.d_tile_y.alloc(P.tile_y.size()))) return st;
    AOS2_HIP_CHECK(hipMemcpy(P.d_tile_x.p, P.tile_x.data(), P.tile_x.size() * sizeof(int4), hipMemcpyHostToDevice));
    AOS2_HIP_CHECK(hipMemcpy(P.d_tile_y.p, P.tile_y.data(), P.tile_y.size() * sizeof(int4), hipMemcpyHostToDevice));
    e->batch_cap = 0;  // buffers depend on the plan
    return AOS2_OK;
}

static int init_device(aos2_extractor *e)
{
    int st = bind_device(e->device);
    if (st) return st;
    if (e->dev_ready) return AOS2_OK;
    {
        const char *v = getenv("AOS2_NSTREAMS");
        const int ns = v ? std::max(1, std::min(kMaxStreams, atoi(v))) : kMaxStreams;
        for (int i = 0; i < kMaxStreams; ++i) {
            if (i < ns) {
                if ((st = stream_create(&e->streams[i], false))) return st;
            } else
                e->streams[i] = e->streams[i % ns];
        }
        e->n_streams = ns;
    }
    e->stream = e->streams[0];
    for (auto &ev : e->ev) AOS2_HIP_CHECK(hipEventCreate(&ev));
    if (e->oct_image.total > 0 && prepare_octree_image_kernel(e->oct_image.total) != 0) {
        (void)hipGetLastError();
        e->oct_image.total = 0;  // the runtime refuses that much LDS: keep the per-job kernel
    }
    if (e->oct_pair.total > 0 && prepare_octree_pair_kernel(e->oct_pair.total) != 0) {
        (void)hipGetLastError();
        e->oct_pair.total = 0;
    }
    int r = upload_constants(k_pattern, e->umax, e->gauss7, e->stream);
    if (r != 0) {
        set_error("constant upload failed: %s", hipGetErrorString((hipError_t)r));
        return AOS2_ERR_HIP;
    }
    if ((st = e->d_status.alloc(2))) return st;
    if ((st = e->h_status.alloc(2))) return st;
    AOS2_HIP_CHECK(hipMemsetAsync(e->d_status.p, 0, 2 * sizeof(int32_t), e->stream));
    AOS2_HIP_CHECK(hipStreamSynchronize(e->stream));
    e->dev_ready = true;
    return AOS2_OK;
}

static int ensure_batch(aos2_extractor *e, int batch)
{
    if (batch <= e->batch_cap) return AOS2_OK;
    Plan &P = e->plan;
    const int L = e->nlevels;
    const size_t nc = P.cells.size();
    int st;
    if ((st = e->d_pyr.alloc(P.pyr_bytes * batch + 256))) return st;
    if ((st = e->d_slots.alloc(P.slot_total * batch))) return st;
    if ((st = e->d_dense.alloc(P.slot_total * batch))) return st;
    if ((st = e->d_cell_cnt.alloc(nc * batch))) return st;
    if ((st = e->d_level_off.alloc((size_t)(L + 1) * batch))) return st;
    if ((st = e->d_level_cnt.alloc((size_t)L * batch))) return st;
    if ((st = e->d_sel.alloc((size_t)L * e->cap_level * batch))) return st;
    if ((st = e->d_sel_cnt.alloc((size_t)L * batch))) return st;
    if ((st = e->h_level_off.alloc((size_t)(L + 1) * batch))) return st;
    if ((st = e->h_sel_cnt.alloc((size_t)L * batch))) return st;
    if ((st = e->h_nout.alloc(batch))) return st;
    if (!e->host_octree) {
        const size_t jobs = (size_t)L * batch;
        if ((st = e->o_xs.alloc(P.oct_cand_total * batch))) return st;
        if ((st = e->o_ys.alloc(P.oct_cand_total * batch))) return st;
        if ((st = e->o_sc.alloc(P.oct_cand_total * batch))) return st;
        if ((st = e->o_perm.alloc(P.oct_cand_total * batch))) return st;
        if ((st = e->o_tmp.alloc(P.oct_cand_total * batch))) return st;
        if ((st = e->o_pairs.alloc(P.oct_node_total * 4 * batch))) return st;
        if ((st = e->o_idx.alloc(jobs * e->cap_level))) return st;
        if ((st = e->o_nodes.alloc(P.oct_node_total * batch))) return st;
    } else {
        if ((st = e->h_sel.alloc((size_t)L * e->cap_level * batch))) return st;
    }
    if (e->blur_level) {
        e->blur_bytes = blur_plan(P.levels.data(), L, P.pyr_bytes, &e->blur_plan_h);
        if ((st = e->d_blur.alloc(e->blur_bytes * batch + 256))) return st;
    }
    // row h of every plane (1 guard row) and pitch padding are read by 32-bit tile loads: keep
    // them defined
    AOS2_HIP_CHECK(hipMemsetAsync(e->d_pyr.p, 0, P.pyr_bytes * batch + 256, e->stream));
    // The chunks of the batch that follows run on streams of their own: they must not start while this memset is still running on
    // the first stream (round 4: a fresh handle given 1920 images -- 2 GB of pyramid, a memset of ~1 ms -- had the pyramids of its
    // second and third chunk zeroed under them; 720 images lost a few frames at the end, 256 none).  Once per capacity growth.
    AOS2_HIP_CHECK(hipStreamSynchronize(e->stream));
    e->batch_cap = batch;
    return AOS2_OK;
}

static int ensure_out(aos2_extractor *e, int batch, int cap)
{
    int st;
    if ((st = e->d_kps.alloc((size_t)batch * cap))) return st;
    if ((st = e->d_desc.alloc((size_t)batch * cap * 32))) return st;
    if ((st = e->d_nout.alloc(batch))) return st;
    return AOS2_OK;
}

// host octree stage (optional): D2H candidates, std::thread pool, H2D selection
static int octree_on_host(aos2_extractor *e, int batch)
{
    Plan &P = e->plan;
    const int L = e->nlevels;
    AOS2_HIP_CHECK(hipMemcpyAsync(e->h_level_off.p, e->d_level_off.p, sizeof(int32_t) * (L + 1) * batch,
                                  hipMemcpyDeviceToHost, e->stream));
    AOS2_HIP_CHECK(hipStreamSynchronize(e->stream));
    int maxn = 0;
    for (int b = 0; b < batch; ++b) maxn = std::max(maxn, e->h_level_off.p[(size_t)b * (L + 1) + L]);
    if (maxn == 0) maxn = 1;
    int st;
    if ((st = e->h_dense.alloc((size_t)maxn * batch))) return st;
    AOS2_HIP_CHECK(hipMemcpy2DAsync(e->h_dense.p, (size_t)maxn * 4, e->d_dense.p, P.slot_total * 4, (size_t)maxn * 4,
                                    batch, hipMemcpyDeviceToHost, e->stream));
    AOS2_HIP_CHECK(hipStreamSynchronize(e->stream));
    std::atomic<int> next{0}, fail{0};
    const int jobs = batch * L;
    auto worker = [&]() {
        std::vector<int16_t> xs, ys;
        std::vector<uint8_t> sc;
        std::vector<int32_t> perm, tmp, pairs, idx(e->cap_level);
        std::vector<OctNode> nodes;
        for (;;) {
            const int job = next.fetch_add(1);
            if (job >= jobs) break;
            const int b = job / L, l = job % L;
            const int32_t *lo = e->h_level_off.p + (size_t)b * (L + 1);
            const int beg = lo[l], n = lo[l + 1] - lo[l];
            const uint32_t *cand = e->h_dense.p + (size_t)b * maxn + beg;
            int nk = 0;
            if (n > 0) {
                xs.resize(n); ys.resize(n); sc.resize(n); perm.resize(n); tmp.resize(n);
                for (int i = 0; i < n; ++i) {
                    xs[i] = (int16_t)(cand[i] & 0xfff);
                    ys[i] = (int16_t)((cand[i] >> 12) & 0xfff);
                    sc[i] = (uint8_t)(cand[i] >> 24);
                }
                const int mn = oct_max_nodes(n, P.levels[l].nfeat);
                nodes.resize(mn);
                pairs.resize((size_t)4 * mn);
                OctScratch S{nodes.data(), perm.data(), tmp.data(), pairs.data(), pairs.data() + 2 * mn, mn, mn};
                nk = distribute_octree(xs.data(), ys.data(), sc.data(), n, 16, P.levels[l].w - 16, 16,
                                       P.levels[l].h - 16, P.levels[l].nfeat, S, idx.data(), e->cap_level);
                if (nk < 0) fail.store(nk);
                uint32_t *out = e->h_sel.p + ((size_t)b * L + l) * e->cap_level;
                for (int k = 0; k < nk; ++k) out[k] = cand[idx[k]];
            }
            e->h_sel_cnt.p[(size_t)b * L + l] = nk;
        }
    };
    const int nt = std::max(1, std::min(e->host_threads, jobs));
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(worker);
    worker();
    for (auto &t : th) t.join();
    if (fail.load() < 0) {
        set_error("host octree scratch exhausted (%d)", fail.load());
        return AOS2_ERR_CAPACITY;
    }
    AOS2_HIP_CHECK(hipMemcpyAsync(e->d_sel.p, e->h_sel.p, sizeof(uint32_t) * L * e->cap_level * batch,
                                  hipMemcpyHostToDevice, e->stream));
    AOS2_HIP_CHECK(hipMemcpyAsync(e->d_sel_cnt.p, e->h_sel_cnt.p, sizeof(int32_t) * L * batch, hipMemcpyHostToDevice, e->stream));
    return AOS2_OK;
}

static int finish_device(aos2_extractor *e);

// Enqueues one batch on the handle's streams and returns; finish_device() completes it.  Chunk c of every batch
// uses stream c and the scratch of its own image range, so consecutive batches are ordered per stream and may be
// in flight together: a chunk's latency-bound octree then overlaps the next batch's kernels on the other streams.
// Host buffers of the host-pointer entry point: each chunk's images are uploaded on the chunk's stream in front of
// its kernels and its results downloaded behind them, so the PCIe copies of one chunk overlap the kernels of the
// others (DMA needs page-locked caller memory -- aos2_host_alloc; pageable memory is staged by the runtime).
struct HostIO {
    const uint8_t *imgs;
    int stride;
    size_t image_stride;
    aos2_keypoint_t *kps;
    uint8_t *desc;
};

// true while `s` records for aos2_capture_begin / _end (csrc/replay.hip)
static bool stream_is_capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive;
}

static int enqueue_device(aos2_extractor *e, const uint8_t *d_imgs, int batch, int w, int h, int stride,
                          size_t image_stride, aos2_keypoint_t *d_kps, uint8_t *d_desc, int cap, int32_t *d_nout,
                          const HostIO *io = nullptr)
{
    e->t_enqueue = std::chrono::steady_clock::now();
    int st;
    if ((st = init_device(e))) return st;
    // Batches of one flight share the scratch by absolute image index, chunk c of every batch on stream c.  That is
    // only race-free while the chunk partition stays the same: another batch size cuts the images differently (256 ->
    // [0,85) [85,170) [170,256); 100 -> [0,33) [33,66) [66,100)), and chunk 1 of the new batch (stream 1) would overwrite
    // the scratch of images 33..66 that chunk 0 of the old one (stream 0) may still be reading.  So a change of batch
    // size, like a change of geometry, waits for the flight first.
    if (e->in_flight > 0 && (w != e->plan.w || h != e->plan.h || batch > e->batch_cap || cap != e->flight_cap ||
                             batch != e->last_batch)) {
        if ((st = finish_device(e))) return st;   // scratch is rebuilt / re-partitioned: nothing may be in flight
    }
    if ((st = build_plan(e, w, h))) return st;
    if ((st = ensure_batch(e, batch))) return st;
    Plan &P = e->plan;
    const int L = e->nlevels;
    const int NC = (int)P.cells.size();
    // level 0 is the caller's image (no copy); levels >= 1 live in the pyramid block
    e->img0 = d_imgs;
    e->img0_stride = image_stride;
    e->pitch0 = stride;
    // The batch is cut into chunks that run on separate streams: the octree kernel is
    // latency-bound (one wave per (image, level), ~0.15 ms whatever the batch size) and leaves
    // the CUs idle, so a chunk's octree overlaps the VALU-bound kernels of the other chunks (and, with the
    // asynchronous call, of the next batch).  Measured at B=256, K steps in flight / synchronous call:
    // 2 chunks 0.933 / 0.964 ms, 3 chunks 0.855 / 0.967, 4 chunks 0.852 / 0.971 (needs GPU_MAX_HW_QUEUES=8: the
    // runtime's default of 4 hardware queues puts two of the 4 streams on one queue, 1.16 ms), 6: 1.08, 8: 1.28
    // (smaller chunks lose to kernel tails and queue sharing; replaying each chunk as one hipGraph changed nothing).
    int chunks = e->chunks > 0 ? e->chunks : (batch >= 96 ? 3 : batch >= 64 ? 2 : 1);
    if (io && e->chunks <= 0 && batch >= 32) chunks = 4;   // copy / compute pipeline of the host-pointer call
    if (e->host_octree) chunks = 1;
    chunks = std::min(chunks, std::min(batch, kMaxStreams));
    e->streams_used = std::max(e->streams_used, chunks);
    if (e->input_waited > 0 && chunks > e->input_waited) {   // (aos2_extractor_wait_for_stream: stream 0 waits for the inputs already)
        if (!e->input_fan_ev) AOS2_HIP_CHECK(hipEventCreateWithFlags(&e->input_fan_ev, hipEventDisableTiming));
        AOS2_HIP_CHECK(hipEventRecord(e->input_fan_ev, e->streams[0]));
        for (int c = e->input_waited; c < chunks; ++c) AOS2_HIP_CHECK(hipStreamWaitEvent(e->streams[c], e->input_fan_ev, 0));
    }
    e->input_waited = 0;
    e->last_chunks = chunks;
    if (e->stereo_guard_armed) {   // stereo kernels enqueued since the last batch still read this extractor's pyramids
        // An event of a recording (aos2_capture_begin) and one of executed work cannot wait for each other.  Executed batch behind a
        // replayed recording: the guard is recorded again, now, on the stream the recording ran on (behind every launched replay).
        // Recording behind executed work: no wait (a sequence is run, and waited for, before it is recorded: include/aos2.h).
        const bool rec = stream_is_capturing(e->streams[0]);
        if (!rec && e->stereo_guard_captured) {
            AOS2_HIP_CHECK(hipEventRecord(e->stereo_guard, e->stereo_guard_stream));
            e->stereo_guard_captured = false;
        }
        if (rec == e->stereo_guard_captured)
            for (int c = 0; c < chunks; ++c) AOS2_HIP_CHECK(hipStreamWaitEvent(e->streams[c], e->stereo_guard, 0));
        e->stereo_guard_armed = false;
    }
    auto enqueue = [&](int b0, int nb, hipStream_t s, bool timed) -> int {
        const uint8_t *img = d_imgs + (size_t)b0 * image_stride;
        uint8_t *pyr = e->d_pyr.p + (size_t)b0 * P.pyr_bytes;
        uint32_t *slots = e->d_slots.p + (size_t)b0 * P.slot_total, *dense = e->d_dense.p + (size_t)b0 * P.slot_total;
        int32_t *cell_cnt = e->d_cell_cnt.p + (size_t)b0 * NC, *level_off = e->d_level_off.p + (size_t)b0 * (L + 1);
        uint32_t *sel = e->d_sel.p + (size_t)b0 * L * e->cap_level;
        int32_t *sel_cnt = e->d_sel_cnt.p + (size_t)b0 * L;
        if (io) {
            uint8_t *dst = const_cast<uint8_t *>(img);
            const uint8_t *src = io->imgs + (size_t)b0 * io->image_stride;
            if (io->stride == w && io->image_stride == (size_t)w * h)
                AOS2_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)nb * w * h, hipMemcpyHostToDevice, s));
            else
                for (int b = 0; b < nb; ++b)
                    AOS2_HIP_CHECK(hipMemcpy2DAsync(dst + (size_t)b * image_stride, (size_t)stride, src + (size_t)b * io->image_stride,
                                                    (size_t)io->stride, (size_t)w, (size_t)h, hipMemcpyHostToDevice, s));
        }
        if (timed) AOS2_HIP_CHECK(hipEventRecord(e->ev[0], s));
        // one launch for a few frames (latency: 36 -> ~15 us per frame); the per-level kernel is the throughput form (4 pixels per
        // lane, packed arithmetic: 0.135 ms per 256 frames against 0.38 ms for the fused one, which works pixel by pixel)
        if (P.pyr_fused && nb < 8)
            launch_pyramid_fused(img, image_stride, stride, pyr, P.pyr_bytes, P.d_levels.p, L, P.d_tile_x.p, P.d_tile_y.p, P.ntx, P.nty,
                                 P.d_xofs.p, P.d_xab.p, P.d_yofs.p, P.d_yab.p, P.pyr_buf_pitch, P.pyr_buf_rows, P.pyr_lds, nb, s);
        else
            for (int l = 1; l < L; ++l) {
                const bool from0 = (l == 1);
                launch_resize(from0 ? img : pyr + P.levels[l - 1].off, from0 ? image_stride : P.pyr_bytes,
                              from0 ? stride : P.levels[l - 1].pitch, pyr, P.pyr_bytes, P.levels[l - 1], P.levels[l],
                              P.d_xofs.p, P.d_xab.p, P.d_yofs.p, P.d_yab.p, nb, s);
            }
        if (timed) AOS2_HIP_CHECK(hipEventRecord(e->ev[1], s));
        launch_fast(img, image_stride, stride, pyr, P.pyr_bytes, P.d_levels.p, P.d_cells.p, NC, e->iniTh, e->minTh, P.TP,
                    P.TH, P.SP, P.fast_lds, P.list_cap, P.keep_cap, slots, P.slot_total, cell_cnt, nb, s);
        if (timed) AOS2_HIP_CHECK(hipEventRecord(e->ev[2], s));
        // device octree: every (image, level) job gathers its own candidates from the cell slots; the separate
        // compaction kernel (one latency-bound workgroup per image on the critical path of the chunk, 31 us at
        // B=256) only feeds the host octree path
        if (e->host_octree)
            launch_compact(P.d_cells.p, NC, L, P.d_level_cell_begin.p, slots, P.slot_total, cell_cnt, dense, P.slot_total,
                           level_off, nb, s);
        if (timed) AOS2_HIP_CHECK(hipEventRecord(e->ev[3], s));
        if (e->host_octree) {
            int st2 = octree_on_host(e, nb);
            if (st2) return st2;
        } else {
            const size_t j0 = (size_t)b0 * L, c0 = (size_t)b0 * P.oct_cand_total, n0 = (size_t)b0 * P.oct_node_total;
            OctDevScratch scr{e->o_xs.p + c0, e->o_ys.p + c0, e->o_sc.p + c0, e->o_perm.p + c0, e->o_tmp.p + c0,
                              e->o_pairs.p + 4 * n0, e->o_idx.p + j0 * e->cap_level, e->o_nodes.p + n0,
                              P.oct_cand_total, P.oct_node_total};
            const OctGather gather{P.d_cells.p, P.d_level_cell_begin.p, slots, P.slot_total, cell_cnt, NC,
                                   e->d_level_cnt.p + (size_t)b0 * L};
            if (e->oct_image.total > 0)
                launch_octree_image(dense, P.slot_total, gather, P.d_levels.p, L, nb, scr, sel, (size_t)L * e->cap_level,
                                    sel_cnt, e->cap_level, e->oct_image, s);
            else if (e->oct_pair.total > 0 && nb >= 8)   // (fewer images: the helper-wave form of the per-job kernel)
                launch_octree_pairs(dense, P.slot_total, gather, P.d_levels.p, L, nb, scr, sel, (size_t)L * e->cap_level,
                                    sel_cnt, e->cap_level, e->oct_pair, s);
            else
                launch_octree(dense, P.slot_total, gather, P.d_levels.p, L, nb, scr, sel, (size_t)L * e->cap_level, sel_cnt,
                              e->cap_level, e->oct_lds, s);
        }
        if (timed) AOS2_HIP_CHECK(hipEventRecord(e->ev[4], s));
        if (e->blur_level) {   // (inside the describe stage's events: the A/B compares the stage as a whole)
            uint8_t *bl = e->d_blur.p + (size_t)b0 * e->blur_bytes;
            launch_blur_levels(img, image_stride, stride, pyr, P.pyr_bytes, P.d_levels.p, L, e->blur_plan_h, bl, e->blur_bytes, nb, s);
            launch_describe_blur(img, image_stride, stride, pyr, P.pyr_bytes, bl, e->blur_bytes, e->blur_plan_h, P.d_levels.p, L, sel,
                                 (size_t)L * e->cap_level, e->cap_level, sel_cnt, d_kps + (size_t)b0 * cap, d_desc + (size_t)b0 * cap * 32, cap,
                                 d_nout + b0, nb, e->d_status.p, s);
        } else
        launch_describe(img, image_stride, stride, pyr, P.pyr_bytes, P.d_levels.p, L, sel, (size_t)L * e->cap_level,
                        e->cap_level, sel_cnt, d_kps + (size_t)b0 * cap, d_desc + (size_t)b0 * cap * 32, cap, d_nout + b0, nb,
                        e->umax_nibbles, e->d_status.p, s);
        if (timed) AOS2_HIP_CHECK(hipEventRecord(e->ev[5], s));
        if (io) {
            AOS2_HIP_CHECK(hipMemcpyAsync(io->kps + (size_t)b0 * cap, d_kps + (size_t)b0 * cap, sizeof(aos2_keypoint_t) * (size_t)nb * cap,
                                          hipMemcpyDeviceToHost, s));
            AOS2_HIP_CHECK(hipMemcpyAsync(io->desc + (size_t)b0 * cap * 32, d_desc + (size_t)b0 * cap * 32, (size_t)nb * cap * 32,
                                          hipMemcpyDeviceToHost, s));
        }
        // (the per-level and per-image counts of the LAST batch of a flight are fetched once by finish_device(); errors of
        // earlier batches travel in the sticky status words -- no blit kernels between the chunks' launches)
        return AOS2_OK;
    };
    for (int c = 0; c < chunks; ++c) {
        const int b0 = (int)((long long)batch * c / chunks), b1 = (int)((long long)batch * (c + 1) / chunks);
        if (b1 > b0 && (st = enqueue(b0, b1 - b0, e->streams[c], c == 0))) return st;
    }
    e->timing[6] = (float)chunks;
    e->last_batch = batch;
    e->flight_cap = cap;
    e->flight_nout = d_nout;
    ++e->in_flight;
    return AOS2_OK;
}

// Waits for every batch in flight; reports the sticky device status of all of them and, in detail, the last one.
static int finish_device(aos2_extractor *e)
{
    if (e->in_flight == 0) return AOS2_OK;
    if (stream_is_capturing(e->streams[0])) {   // (refused before the runtime sees the wait: it would invalidate the recording)
        set_error("the extractor's stream is recording (aos2_capture_begin): a host wait -- aos2_extractor_wait, a synchronous call, a "
                  "change of batch size or geometry -- cannot be recorded");
        return AOS2_ERR_ARG;
    }
    const int L = e->nlevels, batch = e->last_batch, cap = e->flight_cap;
    int32_t status[2] = {0, 0};
    for (int c = 1; c < kMaxStreams; ++c) AOS2_HIP_CHECK(hipStreamSynchronize(e->streams[c]));
    // status words + the counts of the last batch: three small copies behind the last chunk of stream 0, one wait
    AOS2_HIP_CHECK(hipMemcpyAsync(e->h_status.p, e->d_status.p, sizeof(status), hipMemcpyDeviceToHost, e->streams[0]));
    if (!e->host_octree)
        AOS2_HIP_CHECK(hipMemcpyAsync(e->h_sel_cnt.p, e->d_sel_cnt.p, sizeof(int32_t) * (size_t)L * batch, hipMemcpyDeviceToHost, e->streams[0]));
    AOS2_HIP_CHECK(hipMemcpyAsync(e->h_nout.p, e->flight_nout, sizeof(int32_t) * (size_t)batch, hipMemcpyDeviceToHost, e->streams[0]));
    AOS2_HIP_CHECK(hipStreamSynchronize(e->streams[0]));
    status[0] = e->h_status.p[0];
    status[1] = e->h_status.p[1];
    e->in_flight = 0;
    e->streams_used = 0;
    AOS2_HIP_CHECK(hipGetLastError());
    for (int i = 0; i < 5; ++i) (void)hipEventElapsedTime(&e->timing[i], e->ev[i], e->ev[i + 1]);
    e->timing[5] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - e->t_enqueue).count();
    if (status[0] != 0 || status[1] != 0) AOS2_HIP_CHECK(hipMemset(e->d_status.p, 0, sizeof(status)));
    for (int i = 0; i < L * batch; ++i) {
        if (e->h_sel_cnt.p[i] < 0) {
            set_error("octree stage failed for image %d level %d (code %d: %s)", i / L, i % L, e->h_sel_cnt.p[i],
                      e->h_sel_cnt.p[i] == -4   ? "candidate capacity exceeded"
                      : e->h_sel_cnt.p[i] == -1 ? "level more than twice as tall as wide: round(width / height) == 0, the reference's "
                                                  "DistributeOctTree divides by zero (src/ORBextractor.cc:545)"
                                                : "node arena exhausted");
            return AOS2_ERR_CAPACITY;
        }
    }
    for (int b = 0; b < batch; ++b)
        if (e->h_nout.p[b] > cap) {
            set_error("image %d has %d keypoints, capacity %d", b, e->h_nout.p[b], cap);
            return AOS2_ERR_CAPACITY;
        }
    // earlier batches of the same flight (their per-image counts are gone; the sticky words are not)
    if (status[0] < 0) {
        set_error("octree stage failed in an earlier batch of this flight (code %d: %s)", status[0],
                  status[0] == -4 ? "candidate capacity exceeded" : status[0] == -1 ? "level more than twice as tall as wide" : "node arena exhausted");
        return AOS2_ERR_CAPACITY;
    }
    if (status[1] > cap) {
        set_error("an earlier batch of this flight produced %d keypoints for one image, capacity %d", status[1], cap);
        return AOS2_ERR_CAPACITY;
    }
    return AOS2_OK;
}

static int run_device(aos2_extractor *e, const uint8_t *d_imgs, int batch, int w, int h, int stride,
                      size_t image_stride, aos2_keypoint_t *d_kps, uint8_t *d_desc, int cap, int32_t *d_nout,
                      const HostIO *io = nullptr)
{
    int st;
    if (e->in_flight > 0 && (st = finish_device(e))) return st;
    if ((st = enqueue_device(e, d_imgs, batch, w, h, stride, image_stride, d_kps, d_desc, cap, d_nout, io))) return st;
    return finish_device(e);
}

}  // namespace aos2

// ------------------------------------------------------------------------------------------ C ABI
extern "C" {

const char *aos2_last_error(void) { return g_err.c_str(); }
const char *aos2_version(void) { return "aos2 0.1 (gfx950, HIP)"; }

int aos2_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int aos2_device_local_cpus(int device, char *buf, int cap)
{
    if (!buf || cap < 1) return AOS2_ERR_ARG;
    buf[0] = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
        (void)hipGetLastError();
        return AOS2_ERR_NO_DEVICE;
    }
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) {
        (void)hipGetLastError();
        return AOS2_OK;   // unknown: the empty list
    }
    for (char *c = bus; *c; ++c) *c = (char)tolower(*c);   // sysfs names are lower case
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist";
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return AOS2_OK;
    char line[4096] = {0};
    const bool got = fgets(line, sizeof(line), f) != nullptr;
    fclose(f);
    if (!got) return AOS2_OK;
    size_t len = strlen(line);
    while (len && (line[len - 1] == '\n' || line[len - 1] == ' ')) line[--len] = 0;
    if ((int)len + 1 > cap) return AOS2_ERR_CAPACITY;
    memcpy(buf, line, len + 1);
    return AOS2_OK;
}

int aos2_extractor_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast,
                          int device, aos2_extractor_t **out)
{
    if (!out) return AOS2_ERR_ARG;
    *out = nullptr;
    if (nfeatures <= 0 || nlevels < 1 || nlevels > kMaxLevels || !(scale_factor > 1.0f) || ini_th_fast < 1 ||
        min_th_fast < 1 || min_th_fast > ini_th_fast || ini_th_fast > 255) {
        set_error("bad extractor parameters");
        return AOS2_ERR_ARG;
    }
    if (scale_factor > 2.0f) {  // the resize kernel stages <= 2.5x source windows in LDS
        set_error("scale factor %.3f > 2.0 is not supported", scale_factor);
        return AOS2_ERR_ARG;
    }
    aos2_extractor *e = new aos2_extractor();
    e->nfeatures = nfeatures;
    e->nlevels = nlevels;
    e->iniTh = ini_th_fast;
    e->minTh = min_th_fast;
    e->scaleFactor = scale_factor;
    e->device = device;
    build_host_tables(e);
    if (const char *v = getenv("AOS2_OCTREE")) e->host_octree = (strcmp(v, "host") == 0);
    {
        // LDS budget of an octree job: sized for ~8 candidates per requested feature on level 0 (the
        // busiest level), capped at the 64 KB a workgroup may take without opt-in; jobs that need more
        // run over global scratch.  AOS2_OCT_LDS=0 forces the global path (tests).
        const int n0 = e->mnFeaturesPerLevel[0];
        size_t want = oct_lds_bytes(8 * n0 + 256, n0);
        if (want > 65536) want = 65536;
        e->oct_lds = (int)want;
        if (const char *v = getenv("AOS2_OCT_LDS")) e->oct_lds = std::max(0, std::min(65536, atoi(v)));
        // Optional (AOS2_OCT_IMAGE=1): if the working sets of all levels of one image fit the 160 KB of a CU together,
        // the octree runs as one workgroup per image (one wave per level, per-level LDS slices), so that every job of
        // the batch is resident at once.  Measured: 0.161 vs 0.166 ms un-chunked, but 1.166 vs 1.151 ms per step with
        // the default two-stream chunking (a 137 KB workgroup starves the other chunk's kernels) -- hence opt-in.
        const char *vi = getenv("AOS2_OCT_IMAGE");
        if (e->oct_lds > 0 && nlevels <= 16 && vi && atoi(vi) != 0) {
            int off = 0;
            for (int l = 0; l < nlevels; ++l) {
                const int nl = e->mnFeaturesPerLevel[l];
                const int bytes = (int)((oct_lds_bytes(8 * nl + 128, nl) + 255) & ~(size_t)255);
                e->oct_image.off[l] = off;
                e->oct_image.bytes[l] = bytes;
                off += bytes;
            }
            e->oct_image.total = off <= 160 * 1024 ? off : 0;
        }
        // Default for batches: level g paired with level nlevels - 1 - g in one workgroup, each job in a slice of its own size
        // (AOS2_OCT_PAIR=0: one job per workgroup with the level-0 reservation, the form of rounds 1-5 and of calls of < 8 images).
        const char *vp = getenv("AOS2_OCT_PAIR");
        if (e->oct_lds > 0 && nlevels >= 2 && nlevels <= 16 && !(vp && atoi(vp) == 0) && !getenv("AOS2_OCT_GROUP_LEVELS")) {   // (that switch tests the per-job kernel's helper waves)
            int bytes[16], total = 0;
            for (int l = 0; l < nlevels; ++l) {
                const int nl = e->mnFeaturesPerLevel[l];
                bytes[l] = (int)((oct_lds_bytes(8 * nl + (l == 0 ? 256 : 128), nl) + 255) & ~(size_t)255);
                if (bytes[l] > e->oct_lds) bytes[l] = (e->oct_lds + 255) & ~255;   // (never more than the per-job form takes: larger jobs use the global scratch either way)
            }
            for (int g2 = 0; g2 < (nlevels + 1) / 2; ++g2) {
                const int la = g2, lb = nlevels - 1 - g2;
                e->oct_pair.off[la] = 0; e->oct_pair.bytes[la] = bytes[la];
                int sum = bytes[la];
                if (lb != la) {
                    e->oct_pair.off[lb] = bytes[la]; e->oct_pair.bytes[lb] = bytes[lb];
                    sum += bytes[lb];
                }
                total = std::max(total, sum);
            }
            e->oct_pair.total = total <= 160 * 1024 ? total : 0;
        }
    }
    const unsigned hc = std::thread::hardware_concurrency();
    e->host_threads = (int)std::min(32u, std::max(1u, hc));
    if (const char *v = getenv("AOS2_HOST_THREADS")) e->host_threads = std::max(1, atoi(v));
    if (const char *v = getenv("AOS2_DESC_BLUR")) e->blur_level = strcmp(v, "level") == 0 && e->nlevels <= 8;   // (the blur plan holds 8 levels; beyond, the per-keypoint form)
    if (const char *v = getenv("AOS2_CHUNKS")) e->chunks = std::max(0, std::min(kMaxStreams, atoi(v)));
    *out = e;
    return AOS2_OK;
}

void aos2_extractor_destroy(aos2_extractor_t *e)
{
    if (!e) return;
    if (e->dev_ready) {
        (void)hipSetDevice(e->device);
        for (auto &sx : e->streams) (void)hipStreamSynchronize(sx);
        if (aos2_extractor *peer = e->stereo_peer) {   // the other eye's guard may name a stream of this handle: drained above, so the
            if (peer->stereo_peer == e) peer->stereo_peer = nullptr;   // guard has nothing left to order -- forget it before the stream dies
            for (int i = 0; i < e->n_streams; ++i)
                if (peer->stereo_guard_stream == e->streams[i]) {
                    peer->stereo_guard_armed = peer->stereo_guard_captured = false;
                    peer->stereo_guard_stream = nullptr;
                }
        }
        e->plan.release_device();
        e->d_pyr.release(); e->d_blur.release(); e->d_in.release(); e->d_desc.release(); e->d_slots.release(); e->d_dense.release();
        e->d_sel.release(); e->d_cell_cnt.release(); e->d_level_off.release(); e->d_level_cnt.release(); e->d_sel_cnt.release();
        e->d_nout.release(); e->d_kps.release();
        e->st_sad.release(); e->st_rows.release(); e->st_io.release(); e->st_host.release();
        e->o_xs.release(); e->o_ys.release(); e->o_sc.release(); e->o_perm.release(); e->o_tmp.release();
        e->o_pairs.release(); e->o_idx.release(); e->o_nodes.release();
        e->h_level_off.release(); e->h_sel_cnt.release(); e->h_nout.release(); e->h_status.release(); e->h_dense.release(); e->h_sel.release();
        for (auto &ev : e->ev) (void)hipEventDestroy(ev);
        for (int i = 0; i < e->n_streams; ++i) (void)hipStreamDestroy(e->streams[i]);
        for (auto &oe : e->order_ev)
            if (oe) (void)hipEventDestroy(oe);
        for (hipEvent_t x : {e->stereo_guard, e->stereo_t0, e->stereo_t1, e->input_ev, e->input_fan_ev})
            if (x) (void)hipEventDestroy(x);
    }
    delete e;
}

int aos2_extractor_levels(const aos2_extractor_t *e) { return e->nlevels; }
float aos2_extractor_scale_factor(const aos2_extractor_t *e) { return e->scaleFactor; }
const float *aos2_extractor_scale_factors(const aos2_extractor_t *e) { return e->mvScaleFactor; }
const float *aos2_extractor_inv_scale_factors(const aos2_extractor_t *e) { return e->mvInvScaleFactor; }
const float *aos2_extractor_sigma2(const aos2_extractor_t *e) { return e->mvLevelSigma2; }
const float *aos2_extractor_inv_sigma2(const aos2_extractor_t *e) { return e->mvInvLevelSigma2; }
const int *aos2_extractor_features_per_level(const aos2_extractor_t *e) { return e->mnFeaturesPerLevel; }
const int *aos2_extractor_umax(const aos2_extractor_t *e) { return e->umax; }
int aos2_extractor_max_keypoints(const aos2_extractor_t *e) { return e->max_kp; }
int aos2_extractor_max_keypoints_for(const aos2_extractor_t *e, int w, int h)
{
    if (!e) return 0;
    int cl = 0, tot = 0;
    keypoint_bounds(e, w, h, &cl, &tot);
    return tot;
}

int aos2_extractor_extract_batch_device(aos2_extractor_t *e, const uint8_t *d_imgs, int batch, int w, int h,
                                        int stride, size_t image_stride, aos2_keypoint_t *d_kps, uint8_t *d_desc,
                                        int cap, int32_t *d_n_out)
{
    if (!e || !d_imgs || !d_kps || !d_desc || !d_n_out || batch <= 0 || w <= 0 || h <= 0 || stride < w || cap <= 0) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    return run_device(e, d_imgs, batch, w, h, stride, image_stride, d_kps, d_desc, cap, d_n_out);
}

int aos2_extractor_extract_batch_device_async(aos2_extractor_t *e, const uint8_t *d_imgs, int batch, int w, int h,
                                              int stride, size_t image_stride, aos2_keypoint_t *d_kps,
                                              uint8_t *d_desc, int cap, int32_t *d_n_out)
{
    if (!e || !d_imgs || !d_kps || !d_desc || !d_n_out || batch <= 0 || w <= 0 || h <= 0 || stride < w || cap <= 0) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    if (e->host_octree) {   // the host octree path synchronises inside the batch: run it synchronously
        return run_device(e, d_imgs, batch, w, h, stride, image_stride, d_kps, d_desc, cap, d_n_out);
    }
    return enqueue_device(e, d_imgs, batch, w, h, stride, image_stride, d_kps, d_desc, cap, d_n_out);
}

// fixed-size per-frame slots for the one exchange step of the sharded path:
// {int32 n; int32 pad[3]; KeyPoint[cap]; uint8 desc[cap][32]} rounded up to 16 B -- sharding.py's layout
__global__ __launch_bounds__(256) void pack_slots_kernel(const aos2_keypoint_t *__restrict__ kps, const uint8_t *__restrict__ desc,
                                                         const int32_t *__restrict__ n_out, int cap, uint8_t *__restrict__ slots,
                                                         size_t slot_bytes)
{
    const int b = blockIdx.y;
    uint4 *dst = reinterpret_cast<uint4 *>(slots + (size_t)b * slot_bytes);
    const int n = min(n_out[b], cap);
    const size_t words = slot_bytes / 16;
    const size_t kp_words = ((size_t)cap * 28) / 16;   // cap * 28 is a multiple of 16 for the capacities in use (checked by the host)
    const uint4 *ksrc = reinterpret_cast<const uint4 *>(kps + (size_t)b * cap);
    const uint4 *dsrc = reinterpret_cast<const uint4 *>(desc + (size_t)b * cap * 32);
    const size_t kp_used = ((size_t)n * 28 + 15) / 16, d_used = (size_t)n * 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (i == 0)
            v.x = (uint32_t)n;
        else if (i - 1 < kp_words) {
            if (i - 1 < kp_used) v = ksrc[i - 1];
        } else if (i - 1 - kp_words < d_used)
            v = dsrc[i - 1 - kp_words];
        dst[i] = v;
    }
}

int aos2_extractor_pack_slots(aos2_extractor_t *e, int batch, const aos2_keypoint_t *d_kps, const uint8_t *d_desc,
                              const int32_t *d_n, int cap, uint8_t *d_slots, size_t slot_bytes, void *hip_stream)
{
    if (!e || batch <= 0 || !d_kps || !d_desc || !d_n || !d_slots || cap <= 0 || ((size_t)cap * 28) % 16 != 0 ||
        slot_bytes % 16 != 0 || slot_bytes < 16 + (size_t)cap * 60) {
        set_error("bad argument (cap must be a multiple of 4, slot_bytes >= 16 + 60 * cap and a multiple of 16)");
        return AOS2_ERR_ARG;
    }
    int st = aos2_extractor_stream_wait(e, hip_stream);
    if (st) return st;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    hipLaunchKernelGGL(pack_slots_kernel, dim3(16, batch), dim3(256), 0, s, d_kps, d_desc, d_n, cap, d_slots, slot_bytes);
    AOS2_HIP_CHECK(hipGetLastError());
    return AOS2_OK;
}

int aos2_extractor_wait_for_stream(aos2_extractor_t *e, void *hip_stream)
{
    if (!e) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    int st = init_device(e);
    if (st) return st;
    hipStream_t src = static_cast<hipStream_t>(hip_stream);
    if (!e->input_ev) AOS2_HIP_CHECK(hipEventCreateWithFlags(&e->input_ev, hipEventDisableTiming));
    AOS2_HIP_CHECK(hipEventRecord(e->input_ev, src));
    // the streams the last batch's chunks ran on wait now; a next batch cut into more chunks orders the others behind stream 0
    // (enqueue_device).  Streams that get no work are left alone: a recording (aos2_capture_begin) must not be joined by
    // streams that never return to it.
    // While `src` is being recorded only stream 0 joins: the batch behind this call may be cut into fewer chunks than the last one,
    // and a stream that joined a recording without getting work never returns to it (hipStreamEndCapture would fail).
    const int k = stream_is_capturing(src) ? 1 : std::max(1, std::min(e->n_streams, e->last_chunks));
    for (int i = 0; i < k; ++i) AOS2_HIP_CHECK(hipStreamWaitEvent(e->streams[i], e->input_ev, 0));
    e->input_waited = std::max(e->input_waited, k);
    return AOS2_OK;
}

int aos2_extractor_stream_wait(aos2_extractor_t *e, void *hip_stream)
{
    if (!e) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    if (!e->dev_ready) return AOS2_OK;   // nothing was ever enqueued
    int st = bind_device(e->device);
    if (st) return st;
    hipStream_t waiter = static_cast<hipStream_t>(hip_stream);
    const int used = std::max(1, std::min(e->n_streams, e->streams_used));
    for (int i = 0; i < used; ++i) {
        if (!e->order_ev[i]) AOS2_HIP_CHECK(hipEventCreateWithFlags(&e->order_ev[i], hipEventDisableTiming));
        if (e->streams[i] == waiter) continue;   // (in order behind its own work already)
        AOS2_HIP_CHECK(hipEventRecord(e->order_ev[i], e->streams[i]));
        AOS2_HIP_CHECK(hipStreamWaitEvent(waiter, e->order_ev[i], 0));
    }
    return AOS2_OK;
}

int aos2_extractor_wait(aos2_extractor_t *e)
{
    if (!e) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    if (!e->dev_ready) return AOS2_OK;
    int st = bind_device(e->device);
    if (st) return st;
    return finish_device(e);
}

int aos2_extractor_extract_batch(aos2_extractor_t *e, const uint8_t *imgs, int batch, int w, int h, int stride,
                                 size_t image_stride, aos2_keypoint_t *kps, uint8_t *desc, int cap, int32_t *n_out)
{
    if (!e || batch <= 0 || !n_out) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    if (!imgs || w <= 0 || h <= 0) {  // empty image: silent return (:1046)
        for (int b = 0; b < batch; ++b) n_out[b] = 0;
        return AOS2_OK;
    }
    if (stride < w || cap <= 0 || !kps || !desc) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    int st;
    if ((st = init_device(e))) return st;
    if ((st = e->d_in.alloc((size_t)batch * w * h))) return st;
    if ((st = ensure_out(e, batch, cap))) return st;
    e->out_cap = cap;
    // uploads, kernels and downloads are pipelined per chunk on the chunk's stream (HostIO)
    const HostIO io{imgs, stride, image_stride, kps, desc};
    st = run_device(e, e->d_in.p, batch, w, h, w, (size_t)w * h, e->d_kps.p, e->d_desc.p, cap, e->d_nout.p, &io);
    if (st == AOS2_OK || st == AOS2_ERR_CAPACITY) {
        for (int b = 0; b < batch; ++b) n_out[b] = e->h_nout.p[b];
    }
    return st;
}

int aos2_host_alloc(void **p, size_t bytes)
{
    if (!p || bytes == 0) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    *p = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device: page-locked host memory needs the GPU runtime");
        return AOS2_ERR_NO_DEVICE;
    }
    AOS2_HIP_CHECK(hipHostMalloc(p, bytes, hipHostMallocDefault));
    return AOS2_OK;
}

int aos2_host_free(void *p)
{
    if (!p) return AOS2_OK;
    AOS2_HIP_CHECK(hipHostFree(p));
    return AOS2_OK;
}

int aos2_extractor_extract(aos2_extractor_t *e, const uint8_t *img, int w, int h, int stride, aos2_keypoint_t *kps,
                           uint8_t *desc, int cap, int *n_out)
{
    int32_t n = 0;
    const int st = aos2_extractor_extract_batch(e, img, 1, w, h, stride, (size_t)stride * (h > 0 ? h : 0), kps, desc,
                                                cap, &n);
    if (n_out) *n_out = n;
    return st;
}

// ---------------------------------------------------------------------------------------------
// Frame::ComputeStereoMatches (src/Frame.cc:495-669): the two extractors keep mvImagePyramid of the
// last call in HBM; kernels are in stereo.hip.
static void fill_view(const aos2_extractor *e, PyrView &v)
{
    v.img0 = e->img0;
    v.img0_stride = e->img0_stride;
    v.pitch0 = e->pitch0;
    v.pyr = e->d_pyr.p;
    v.pyr_bytes = e->plan.pyr_bytes;
    v.nlevels = e->nlevels;
    for (int l = 0; l < e->nlevels; ++l) {
        const LevelDev &L = e->plan.levels[l];
        v.w[l] = L.w; v.h[l] = L.h; v.pitch[l] = L.pitch; v.off[l] = L.off;
        v.scale[l] = e->mvScaleFactor[l];
        v.inv_scale[l] = e->mvInvScaleFactor[l];
    }
}

static int stereo_check(const aos2_extractor *l, const aos2_extractor *r)
{
    if (!l || !r || l->plan.levels.empty() || r->plan.levels.empty() || l->last_batch <= 0 || r->last_batch <= 0) {
        set_error("ComputeStereoMatches: both extractors must hold the pyramids of an extract call");
        return AOS2_ERR_ARG;
    }
    if (l->device != r->device || l->nlevels != r->nlevels || l->nlevels > kStereoMaxLevels ||
        l->scaleFactor != r->scaleFactor || l->plan.w != r->plan.w || l->plan.h != r->plan.h) {
        set_error("ComputeStereoMatches: left/right extractors differ (device, levels, scale or image size)");
        return AOS2_ERR_ARG;
    }
    return AOS2_OK;
}

static int stereo_run(aos2_extractor *l, aos2_extractor *r, int first_image, int batch, const aos2_keypoint_t *d_kpl,
                      const uint8_t *d_dl, const int32_t *d_nl, const aos2_keypoint_t *d_kpr, const uint8_t *d_dr,
                      const int32_t *d_nr, int cap, int max_n_left, float mb, float mbf, float *d_ur, float *d_depth, bool sync = true)
{
    int st;
    if ((st = l->st_sad.alloc((size_t)batch * cap))) return st;
    StereoArgs a;
    fill_view(l, a.L);
    fill_view(r, a.R);
    a.first_image_l = a.first_image_r = first_image;
    a.kp_l = d_kpl; a.kp_r = d_kpr; a.desc_l = d_dl; a.desc_r = d_dr; a.n_l = d_nl; a.n_r = d_nr;
    a.cap = cap; a.batch = batch; a.mb = mb; a.mbf = mbf;
    a.u_right = d_ur; a.depth = d_depth; a.sad = l->st_sad.p;
    a.rows = a.L.h[0];
    a.row_cap = cap * stereo_row_span(a.L);
    if ((st = l->st_rows.alloc((size_t)batch * ((size_t)a.rows + 1 + (size_t)a.row_cap)))) return st;
    a.row_off = l->st_rows.p;
    a.row_idx = l->st_rows.p + (size_t)batch * ((size_t)a.rows + 1);
    // (events of its own: ev[0..5] are the stage timings of the extraction, which finish_device() reads)
    if (!l->stereo_t0) {
        AOS2_HIP_CHECK(hipEventCreate(&l->stereo_t0));
        AOS2_HIP_CHECK(hipEventCreate(&l->stereo_t1));
    }
    for (aos2_extractor *x : {l, r})
        if (!x->stereo_guard) AOS2_HIP_CHECK(hipEventCreateWithFlags(&x->stereo_guard, hipEventDisableTiming));
    AOS2_HIP_CHECK(hipEventRecord(l->stereo_t0, l->stream));
    if ((st = launch_stereo(a, max_n_left, l->stream))) return st;
    AOS2_HIP_CHECK(hipEventRecord(l->stereo_t1, l->stream));
    for (aos2_extractor *x : {l, r}) {   // the next extraction of either eye is ordered behind these kernels on all its streams
        AOS2_HIP_CHECK(hipEventRecord(x->stereo_guard, l->stream));
        x->stereo_guard_armed = true;
        x->stereo_guard_captured = stream_is_capturing(l->stream);
        x->stereo_guard_stream = l->stream;
    }
    if (l != r) {
        l->stereo_peer = r;
        r->stereo_peer = l;
    }
    if (!sync) return AOS2_OK;
    AOS2_HIP_CHECK(hipStreamSynchronize(l->stream));
    float ms = 0;
    AOS2_HIP_CHECK(hipEventElapsedTime(&ms, l->stereo_t0, l->stereo_t1));
    l->stereo_ms = ms;
    return AOS2_OK;
}

// The stereo Frame constructor's pipeline without a host wait (src/Frame.cc:57-113: the two ExtractORB threads are joined, then
// ComputeStereoMatches runs): left's first stream waits -- on the device -- for every stream of both extractors' batches in
// flight, the stereo kernels are enqueued behind, and whatever orders itself behind `left` afterwards (aos2_extractor_stream_wait,
// aos2_frames_build_stereo, aos2_extractor_wait) is ordered behind them.
int aos2_compute_stereo_matches_device_async(aos2_extractor_t *left, aos2_extractor_t *right, int batch,
                                             const aos2_keypoint_t *d_kp_left, const uint8_t *d_desc_left,
                                             const int32_t *d_n_left, const aos2_keypoint_t *d_kp_right,
                                             const uint8_t *d_desc_right, const int32_t *d_n_right, int cap, float mb,
                                             float mbf, float *d_u_right, float *d_depth)
{
    int st;
    if ((st = stereo_check(left, right))) return st;
    if (!d_kp_left || !d_desc_left || !d_n_left || !d_kp_right || !d_desc_right || !d_n_right || !d_u_right ||
        !d_depth || cap <= 0 || batch <= 0 || batch > left->last_batch || batch > right->last_batch || !(mb > 0) || left == right) {
        set_error("ComputeStereoMatches: bad argument");
        return AOS2_ERR_ARG;
    }
    if ((st = bind_device(left->device))) return st;
    if ((st = aos2_extractor_stream_wait(left, left->stream)) || (st = aos2_extractor_stream_wait(right, left->stream))) return st;
    return stereo_run(left, right, 0, batch, d_kp_left, d_desc_left, d_n_left, d_kp_right, d_desc_right, d_n_right, cap,
                      cap, mb, mbf, d_u_right, d_depth, false);
}

int aos2_compute_stereo_matches_device(aos2_extractor_t *left, aos2_extractor_t *right, int batch,
                                       const aos2_keypoint_t *d_kp_left, const uint8_t *d_desc_left,
                                       const int32_t *d_n_left, const aos2_keypoint_t *d_kp_right,
                                       const uint8_t *d_desc_right, const int32_t *d_n_right, int cap, float mb,
                                       float mbf, float *d_u_right, float *d_depth)
{
    int st;
    if ((st = stereo_check(left, right))) return st;
    if (!d_kp_left || !d_desc_left || !d_n_left || !d_kp_right || !d_desc_right || !d_n_right || !d_u_right ||
        !d_depth || cap <= 0 || batch <= 0 || batch > left->last_batch || batch > right->last_batch || !(mb > 0)) {
        set_error("ComputeStereoMatches: bad argument");
        return AOS2_ERR_ARG;
    }
    if ((st = bind_device(left->device))) return st;
    if ((st = finish_device(left)) || (st = finish_device(right))) return st;
    return stereo_run(left, right, 0, batch, d_kp_left, d_desc_left, d_n_left, d_kp_right, d_desc_right, d_n_right, cap,
                      cap, mb, mbf, d_u_right, d_depth);
}

int aos2_compute_stereo_matches(aos2_extractor_t *left, aos2_extractor_t *right, int image,
                                const aos2_keypoint_t *kp_left, const uint8_t *desc_left, int n_left,
                                const aos2_keypoint_t *kp_right, const uint8_t *desc_right, int n_right, float mb,
                                float mbf, float *u_right, float *depth)
{
    int st;
    if ((st = stereo_check(left, right))) return st;
    if (n_left < 0 || n_right < 0 || image < 0 || image >= left->last_batch || image >= right->last_batch ||
        !(mb > 0) || (n_left > 0 && (!kp_left || !desc_left || !u_right || !depth)) ||
        (n_right > 0 && (!kp_right || !desc_right))) {
        set_error("ComputeStereoMatches: bad argument");
        return AOS2_ERR_ARG;
    }
    if (n_left == 0) return AOS2_OK;
    if ((st = bind_device(left->device))) return st;
    if ((st = finish_device(left)) || (st = finish_device(right))) return st;
    aos2_extractor *e = left;
    const int cap = ((n_left > n_right ? n_left : n_right) + 3) & ~3;  // keeps the descriptor blocks 16-byte aligned
    const size_t kb = sizeof(aos2_keypoint_t) * (size_t)cap, db = (size_t)cap * 32;
    // one upload block: kpL | kpR | descL | descR | nL nR ; one download block: uRight | depth
    const size_t o_kr = kb, o_dl = 2 * kb, o_dr = 2 * kb + db, o_n = 2 * kb + 2 * db, o_out = o_n + 16;
    const size_t total = o_out + 2 * sizeof(float) * (size_t)cap;
    if ((st = e->st_io.alloc(total))) return st;
    if ((st = e->st_host.alloc(total))) return st;
    uint8_t *hp = e->st_host.p;
    memcpy(hp, kp_left, sizeof(aos2_keypoint_t) * (size_t)n_left);
    if (n_right) memcpy(hp + o_kr, kp_right, sizeof(aos2_keypoint_t) * (size_t)n_right);
    memcpy(hp + o_dl, desc_left, (size_t)n_left * 32);
    if (n_right) memcpy(hp + o_dr, desc_right, (size_t)n_right * 32);
    int32_t nn[2] = {n_left, n_right};
    memcpy(hp + o_n, nn, sizeof(nn));
    AOS2_HIP_CHECK(hipMemcpyAsync(e->st_io.p, hp, o_out, hipMemcpyHostToDevice, e->stream));
    uint8_t *dp = e->st_io.p;
    st = stereo_run(left, right, image, 1, (const aos2_keypoint_t *)dp, dp + o_dl, (const int32_t *)(dp + o_n),
                    (const aos2_keypoint_t *)(dp + o_kr), dp + o_dr, (const int32_t *)(dp + o_n) + 1, cap, n_left, mb,
                    mbf, (float *)(dp + o_out), (float *)(dp + o_out) + cap);
    if (st) return st;
    AOS2_HIP_CHECK(hipMemcpyAsync(hp + o_out, dp + o_out, 2 * sizeof(float) * (size_t)cap, hipMemcpyDeviceToHost, e->stream));
    AOS2_HIP_CHECK(hipStreamSynchronize(e->stream));
    memcpy(u_right, hp + o_out, sizeof(float) * (size_t)n_left);
    memcpy(depth, hp + o_out + sizeof(float) * (size_t)cap, sizeof(float) * (size_t)n_left);
    return AOS2_OK;
}

float aos2_compute_stereo_matches_last_device_ms(const aos2_extractor_t *left) { return left ? left->stereo_ms : 0.0f; }

int aos2_extractor_pyramid_level_size(const aos2_extractor_t *e, int level, int *w, int *h)
{
    if (!e || level < 0 || level >= e->nlevels || e->plan.levels.empty()) {
        set_error("no pyramid available");
        return AOS2_ERR_ARG;
    }
    if (w) *w = e->plan.levels[level].w;
    if (h) *h = e->plan.levels[level].h;
    return AOS2_OK;
}

int aos2_extractor_pyramid_level(aos2_extractor_t *e, int image, int level, int border, uint8_t *dst, int dst_stride)
{
    if (!e || !dst || level < 0 || level >= e->nlevels || e->plan.levels.empty() || image < 0 ||
        image >= e->last_batch || border < 0 || border > 64) {
        set_error("bad argument / no pyramid available");
        return AOS2_ERR_ARG;
    }
    int st;
    if ((st = bind_device(e->device))) return st;
    if ((st = finish_device(e))) return st;   // batches enqueued asynchronously
    const LevelDev &L = e->plan.levels[level];
    if (dst_stride < L.w + 2 * border) return AOS2_ERR_ARG;
    uint8_t *interior = dst + (size_t)border * dst_stride + border;
    const uint8_t *srcp = level == 0 ? e->img0 + (size_t)image * e->img0_stride
                                     : e->d_pyr.p + (size_t)image * e->plan.pyr_bytes + L.off;
    const size_t srcpitch = level == 0 ? (size_t)e->pitch0 : (size_t)L.pitch;
    AOS2_HIP_CHECK(hipMemcpy2DAsync(interior, (size_t)dst_stride, srcp, srcpitch, (size_t)L.w, (size_t)L.h,
                                    hipMemcpyDeviceToHost, e->stream));
    AOS2_HIP_CHECK(hipStreamSynchronize(e->stream));
    if (border > 0) {  // cv::copyMakeBorder(BORDER_REFLECT_101), :1122-1128
        auto refl = [](int p, int n) {
            while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
            return p;
        };
        for (int y = -border; y < L.h + border; ++y) {
            const uint8_t *srow = interior + (ptrdiff_t)refl(y, L.h) * dst_stride;
            uint8_t *drow = interior + (ptrdiff_t)y * dst_stride;
            for (int x = -border; x < L.w + border; ++x) {
                if (y >= 0 && y < L.h && x >= 0 && x < L.w) continue;
                drow[x] = srow[refl(x, L.w)];
            }
        }
    }
    return AOS2_OK;
}

int aos2_extractor_debug_candidates(aos2_extractor_t *e, int image, int level, int16_t *xs, int16_t *ys,
                                    uint8_t *score, int cap, int *n)
{
    if (!e || !n || level < 0 || level >= e->nlevels || image < 0 || image >= e->last_batch) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    int st;
    if ((st = bind_device(e->device))) return st;
    if ((st = finish_device(e))) return st;   // batches enqueued asynchronously
    const int L = e->nlevels;
    int32_t lo[2];
    if (e->host_octree) {   // dense list of the whole image, written by the compaction kernel
        AOS2_HIP_CHECK(hipMemcpy(lo, e->d_level_off.p + (size_t)image * (L + 1) + level, sizeof(lo), hipMemcpyDeviceToHost));
    } else {                // per-level lists, written by the octree jobs at the level's first slot
        int32_t n_l = 0;
        AOS2_HIP_CHECK(hipMemcpy(&n_l, e->d_level_cnt.p + (size_t)image * L + level, sizeof(n_l), hipMemcpyDeviceToHost));
        lo[0] = e->plan.cells[e->plan.level_cell_begin[level]].slot_off;
        lo[1] = lo[0] + n_l;
    }
    const int cnt = lo[1] - lo[0];
    *n = cnt;
    if (!xs || !ys || !score) return AOS2_OK;
    if (cnt > cap) return AOS2_ERR_CAPACITY;
    std::vector<uint32_t> tmp(cnt > 0 ? cnt : 1);
    if (cnt > 0)
        AOS2_HIP_CHECK(hipMemcpy(tmp.data(), e->d_dense.p + (size_t)image * e->plan.slot_total + lo[0],
                                 sizeof(uint32_t) * cnt, hipMemcpyDeviceToHost));
    for (int i = 0; i < cnt; ++i) {
        xs[i] = (int16_t)(tmp[i] & 0xfff);
        ys[i] = (int16_t)((tmp[i] >> 12) & 0xfff);
        score[i] = (uint8_t)(tmp[i] >> 24);
    }
    return AOS2_OK;
}

int aos2_extractor_set_chunks(aos2_extractor_t *e, int chunks)
{
    if (!e || chunks < 0 || chunks > kMaxStreams) return AOS2_ERR_ARG;
    e->chunks = chunks;
    return AOS2_OK;
}

int aos2_extractor_last_timing(const aos2_extractor_t *e, float *ms, int n)
{
    if (!e || !ms) return AOS2_ERR_ARG;
    for (int i = 0; i < n && i < 8; ++i) ms[i] = e->timing[i];
    return AOS2_OK;
}

int aos2_extractor_bench_fast(aos2_extractor_t *e, int iters, float *avg_ms)
{
    if (!e || !avg_ms || iters <= 0 || e->last_batch <= 0) {
        set_error("bench_fast needs a previous batch");
        return AOS2_ERR_ARG;
    }
    int st;
    if ((st = bind_device(e->device))) return st;
    if ((st = finish_device(e))) return st;   // batches enqueued asynchronously
    Plan &P = e->plan;
    hipStream_t s = e->stream;
    AOS2_HIP_CHECK(hipEventRecord(e->ev[6], s));
    for (int i = 0; i < iters; ++i)
        launch_fast(e->img0, e->img0_stride, e->pitch0, e->d_pyr.p, P.pyr_bytes, P.d_levels.p, P.d_cells.p,
                    (int)P.cells.size(), e->iniTh, e->minTh, P.TP, P.TH, P.SP, P.fast_lds, P.list_cap, P.keep_cap, e->d_slots.p, P.slot_total, e->d_cell_cnt.p, e->last_batch, s);
    AOS2_HIP_CHECK(hipEventRecord(e->ev[7], s));
    AOS2_HIP_CHECK(hipStreamSynchronize(s));
    float ms = 0;
    AOS2_HIP_CHECK(hipEventElapsedTime(&ms, e->ev[6], e->ev[7]));
    *avg_ms = ms / iters;
    return AOS2_OK;
}

int aos2_extractor_bench_describe(aos2_extractor_t *e, int iters, float *avg_ms)
{
    if (!e || !avg_ms || iters <= 0 || e->last_batch <= 0 || !e->d_kps.p || e->out_cap <= 0) {
        set_error("bench_describe needs a previous host-API batch");
        return AOS2_ERR_ARG;
    }
    int st;
    if ((st = bind_device(e->device))) return st;
    if ((st = finish_device(e))) return st;   // batches enqueued asynchronously
    Plan &P = e->plan;
    const int L = e->nlevels;
    hipStream_t s = e->stream;
    const int cap = e->out_cap;
    AOS2_HIP_CHECK(hipEventRecord(e->ev[6], s));
    for (int i = 0; i < iters; ++i)
        launch_describe(e->img0, e->img0_stride, e->pitch0, e->d_pyr.p, P.pyr_bytes, P.d_levels.p, L, e->d_sel.p, (size_t)L * e->cap_level, e->cap_level,
                        e->d_sel_cnt.p, e->d_kps.p, e->d_desc.p, cap, e->d_nout.p, e->last_batch, e->umax_nibbles, e->d_status.p, s);
    AOS2_HIP_CHECK(hipEventRecord(e->ev[7], s));
    AOS2_HIP_CHECK(hipStreamSynchronize(s));
    float ms = 0;
    AOS2_HIP_CHECK(hipEventElapsedTime(&ms, e->ev[6], e->ev[7]));
    *avg_ms = ms / iters;
    return AOS2_OK;
}

}  // extern "C"
