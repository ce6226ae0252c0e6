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
};

// the members of MapPoint the per-frame path reads (include/MapPoint.h): GetWorldPos, GetDescriptor, Observations() > 0,
// GetNormal, GetMin/MaxDistanceInvariance
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
    aos2::FramesDev D = {};
    int32_t *d_overflow = nullptr;   // sticky: largest per-frame window population that exceeded the entry pool's share
    aos2::DevBuf<uint8_t> mem;       // the per-frame member arrays
    aos2::DevBuf<float> tables;      // scale factors | inverse level sigma2
    // scratch of the searches that run on this batch (sized at the first use)
    aos2::DevBuf<uint8_t> scratch;
    aos2::DevBuf<uint64_t> pool;     // candidate entries (8 B each)
    aos2::DevBuf<uint8_t> pose_mem;  // PoseOptimization problem arrays
    aos2::PinnedBuf<uint8_t> h_io;   // small page-locked staging (poses, counts)
    float last_ms[4] = {};
};
