// Device-resident batches of Frames (src/Frame.cc) and the MapPoint table they reference: the members the per-frame
// tracking path reads and writes, kept in HBM between the calls of the chain
//   ORBextractor::operator() -> Frame::Frame (stereo from RGB-D, grid) -> SearchByProjection(Current, Last)
//   -> PoseOptimization -> SearchLocalPoints (isInFrustum + SearchByProjection(F, vpMapPoints)) -> PoseOptimization
// so that no keypoint, descriptor, match or pose crosses PCIe between them.  Internal to the library; the public
// entry points are the aos2_frames_* functions of include/aos2.h.
#pragma once
#include "aos2_common.h"

namespace aos2 {

constexpr int kFrGridCols = AOS2_GRID_COLS, kFrGridRows = AOS2_GRID_ROWS, kFrGridCells = kFrGridCols * kFrGridRows;

// [B][cap] arrays unless noted
struct FramesDev {
    int batch, cap, n_levels;
    const int32_t *n;              // [B] Frame::N
    const aos2_keypoint_t *kps;    // mvKeys (the extractor's output, borrowed)
    const uint8_t *desc;           // [B][cap][32] mDescriptors (borrowed)
    float *kp_x, *kp_y, *kp_angle; // mvKeysUn[i].pt / .angle
    int32_t *kp_octave;            // mvKeysUn[i].octave
    float *u_right, *depth;        // mvuRight, mvDepth
    int32_t *grid_off, *grid_idx;  // mGrid[64][48] as CSR: [B][64 * 48 + 1], [B][cap]
    int32_t *mp;                   // mvpMapPoints as indices into the map-point table, -1 = NULL
    int32_t *mp_seen;              // map point taken away as an outlier in this frame (mnLastFrameSeen = this frame), -1 = none
    uint8_t *mp_state;             // 0 = NULL, 1 = map point without observations, 2 = Observations() > 0
    uint8_t *outlier;              // mvbOutlier
    float *Tcw;                    // [B][16] mTcw, row-major
    const float *scale_factors, *inv_level_sigma2;   // [n_levels] mvScaleFactors, mvInvLevelSigma2
    float min_x, min_y, max_x, max_y, grid_w_inv, grid_h_inv;   // mnMinX .. mfGridElementHeightInv
    float fx, fy, cx, cy, mb, mbf, log_scale_factor;
    float dist[5];                 // mDistCoef: k1 k2 p1 p2 k3 (all 0: mvKeysUn = mvKeys)
    int has_dist;                  // mDistCoef.at<float>(0) != 0 (the reference's test, src/Frame.cc:435)
};

// the members of MapPoint the per-frame path reads (include/MapPoint.h): GetWorldPos, GetDescriptor, Observations() > 0,
// GetNormal, mfMinDistance / mfMaxDistance (raw)
// cv::undistortPoints(p, p, K, distCoef, noArray(), K) on one CV_32FC2 point, as Frame::UndistortKeyPoints /
// Frame::ComputeImageBounds use it (src/Frame.cc:433-493).  OpenCV 3.2 imgproc/undistort.cpp, cvUndistortPoints: camera
// matrix and coefficients widened to double, 5 fixed-point iterations of the inverse of the radial (k1 k2 k3) +
// tangential (p1 p2) model, then P R = K applied, the result rounded to float.  The operation sequence is the library's
// (the terms of the unused higher-order coefficients are kept as products with 0: they decide the sign of a zero); host
// and device run this same function (-ffp-contract=off).
__host__ __device__ inline void undistort_point(float fx_f, float fy_f, float cx_f, float cy_f, const float *k_f, float xin, float yin,
                                                float &xout, float &yout)
{
    const double fx = (double)fx_f, fy = (double)fy_f, cx = (double)cx_f, cy = (double)cy_f;
    const double ifx = 1. / fx, ify = 1. / fy;
    const double k0 = (double)k_f[0], k1 = (double)k_f[1], k2 = (double)k_f[2], k3 = (double)k_f[3], k4 = (double)k_f[4];
    double x = ((double)xin - cx) * ifx, y = ((double)yin - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + 0. * r2 + 0. * r2 * r2;
        const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + 0. * r2 + 0. * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0. * y + cx;
    const double yy = 0. * x + fy * y + cy;
    const double ww = 1. / (0. * x + 0. * y + 1.);
    xout = (float)(xx * ww);
    yout = (float)(yy * ww);
}

struct MapPointsDev {
    int n;
    const float *pos;        // n x 3
    const uint8_t *desc;     // n x 32
    const uint8_t *has_obs;  // n
    const float *normal;     // n x 3
    const float *min_dist, *max_dist;   // n
};

}  // namespace aos2

struct aos2_frames {
    int device = 0;
    bool dev_ready = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {};
    hipEvent_t order_ev = nullptr;   // ordering only (no timing): recorded on this batch's stream for a batch that reads its members
    hipEvent_t ext_ev = nullptr;     // recorded on a caller's stream by aos2_frames_wait_for_stream
    aos2::FramesDev D = {};
    int32_t *d_overflow = nullptr;   // sticky: largest per-frame window population that exceeded the entry pool's share
    aos2::DevBuf<uint8_t> mem;       // the per-frame member arrays
    aos2::DevBuf<float> tables;      // scale factors | inverse level sigma2
    // scratch of the searches that run on this batch (sized at the first use)
    aos2::DevBuf<uint8_t> scratch;
    aos2::DevBuf<uint64_t> pool;     // candidate entries (8 B each)
    aos2::DevBuf<uint8_t> pose_mem;  // PoseOptimization problem arrays
    aos2::PinnedBuf<uint8_t> h_io;   // small page-locked staging (poses, counts)
    aos2::PinnedBuf<int32_t> h_overflow;   // d_overflow: written by the kernels, read by aos2_frames_wait
    aos2::PinnedBuf<uint8_t> kf_host;   // keyframe work (triangulation pairs, fuse targets): staging ...
    aos2::DevBuf<uint8_t> kf_dev;       // ... and its device copy
    aos2::PinnedBuf<uint8_t> kf_host2;  // the same for aos2_frames_fuse (SearchForTriangulation and Fuse of one handle may both be in flight)
    aos2::DevBuf<uint8_t> kf_dev2;
    bool kf_async = false;              // aos2_frames_set_async_keyframe_calls: the keyframe entry points return after enqueueing
    hipEvent_t kf_ev_tri = nullptr, kf_ev_fuse = nullptr;   // behind the upload of the staging buffer of the last call of each kind
    float dist[5] = {0, 0, 0, 0, 0};   // mDistCoef for the next aos2_frames_build (aos2_frames_set_distortion)
    float last_ms[4] = {};
};
