// ORB_SLAM2::Optimizer with the REFERENCE's signatures for the two hot methods (include/Optimizer.h:45-47):
//   void static LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap);
//   int  static PoseOptimization(Frame *pFrame);
// The bodies are what a maintainer puts into src/Optimizer.cc: aos2::LbaWindow (LbaWindow.h) turns the covisibility neighbourhood
// into the POD problem of include/aos2.h (what :457-654 build as std::lists and a g2o graph), ONE C-ABI call replaces :656-744,
// and the write-back of :746-778 consumes its result.  LocalMapping.cc:81 and Tracking.cc:870,
// 994, 1039 call them unchanged.
// Include AFTER the headers that declare Frame, KeyFrame, MapPoint, Map.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <list>
#include <map>
#include <mutex>
#include <vector>

#include "aos2_handles.h"
#include "LbaWindow.h"

namespace aos2 {
// Optional record of the order in which the last LocalBundleAdjustment of this thread emitted its vertices and edges
// (KeyFrame::mnId / MapPoint::mnId): lets a test hand the SAME problem, in the same order, to another solver.
struct LbaRecord {
    bool enabled = false;
    std::vector<int64_t> pose_id, point_id, edge_pose_id, edge_point_id;
};
inline LbaRecord &lba_record()
{
    thread_local LbaRecord r;
    return r;
}
}  // namespace aos2

namespace ORB_SLAM2 {

class Optimizer {
public:
    void static LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap);
    int static PoseOptimization(Frame *pFrame);
};

// src/Optimizer.cc:454-779.  The window comes from aos2::LbaWindow (LbaWindow.h: one walk, hash membership, rows = the C ABI's arrays),
// ONE C-ABI call replaces the g2o graph and its two optimisations (:507-744), the result is scattered back under the map mutex.
inline void Optimizer::LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap)
{
    aos2::ShimClock clk;
    aos2::LbaWindow<KeyFrame, MapPoint> W;
    W.optimise(pKF);
    for (KeyFrame *neighbour : pKF->GetVectorCovisibleKeyFrames()) W.optimise(neighbour);
    W.collect_points();
    W.emit_edges();
    aos2::ShimTiming &T = aos2::last_shim_timing();
    T.gather_us = clk.lap();
    if (pbStopFlag && *pbStopFlag) return;   // :656-658
    if (W.empty()) return;
    const aos2_lba_problem_t P = W.problem(pKF, reinterpret_cast<const volatile uint8_t *>(pbStopFlag));
    std::vector<float> out_T((size_t)P.n_poses * 16), out_X((size_t)P.n_points * 3);
    std::vector<uint8_t> outlier((size_t)P.n_edges);
    aos2_lba_result_t R;
    memset(&R, 0, sizeof(R));
    R.pose_Tcw = out_T.data(); R.point_xyz = out_X.data(); R.edge_outlier = outlier.data(); R.edge_chi2 = nullptr;
    aos2::LbaRecord &rec = aos2::lba_record();
    if (rec.enabled) {
        rec.pose_id = W.pose_ids();
        rec.point_id = W.point_ids();
        rec.edge_pose_id.clear();
        rec.edge_point_id.clear();
        for (const auto &e : W.edge_rows) {
            rec.edge_pose_id.push_back(rec.pose_id[e.first]);
            rec.edge_point_id.push_back(rec.point_id[e.second]);
        }
    }
    clk.lap();
    const int st = aos2_lba_solve(aos2::optimizer_handle(), &P, &R);
    T.call_us = clk.lap();
    if (st == AOS2_ERR_STOPPED) return;
    if (st != AOS2_OK) {   // a failed solve leaves the map as it was (g2o's failures print and go on: SURVEY.md 8(b) "Error conventions")
        aos2::report("LocalBundleAdjustment");
        return;
    }
    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);   // :746
    for (size_t e = 0; e < outlier.size(); ++e) {               // observations that ended as outliers leave the map (:712-757)
        KeyFrame *kf = W.keyframes[W.edge_rows[e].first];
        MapPoint *mp = W.points[W.edge_rows[e].second];
        if (!outlier[e] || mp->isBad()) continue;
        kf->EraseMapPointMatch(mp);
        mp->EraseObservation(kf);
    }
    for (size_t k = 0; k < W.n_optimised; ++k) {                // :761-778
        cv::Mat Tcw(4, 4, CV_32F);
        memcpy(Tcw.ptr<float>(0), &out_T[k * 16], 64);
        W.keyframes[k]->SetPose(Tcw);
    }
    for (size_t j = 0; j < W.points.size(); ++j) {
        cv::Mat X(3, 1, CV_32F);
        memcpy(X.ptr<float>(0), &out_X[j * 3], 12);
        W.points[j]->SetWorldPos(X);
        W.points[j]->UpdateNormalAndDepth();
    }
    T.scatter_us = clk.lap();
}

// src/Optimizer.cc:239-452
inline int Optimizer::PoseOptimization(Frame *pFrame)
{
    aos2::ShimClock clk;
    const int N = pFrame->N;
    std::vector<float> Xw, obs, is2;
    std::vector<uint8_t> stereo;
    std::vector<int> vnIndexEdge;
    for (int i = 0; i < N; i++) {   // :275-350
        MapPoint *pMP = pFrame->mvpMapPoints[i];
        if (!pMP) continue;
        pFrame->mvbOutlier[i] = false;   // :283, :320
        const cv::KeyPoint &kpUn = pFrame->mvKeysUn[i];
        const float kp_ur = pFrame->mvuRight[i];
        const cv::Mat X = pMP->GetWorldPos();
        for (int k = 0; k < 3; ++k) Xw.push_back(X.at<float>(k));
        obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y); obs.push_back(kp_ur);
        stereo.push_back(kp_ur < 0 ? 0 : 1);   // :278
        is2.push_back(pFrame->mvInvLevelSigma2[kpUn.octave]);
        vnIndexEdge.push_back(i);
    }
    const int n = (int)vnIndexEdge.size();
    aos2_pose_problem_t P;
    P.n = n;
    P.Xw = Xw.data(); P.obs = obs.data(); P.stereo = stereo.data(); P.inv_sigma2 = is2.data();
    P.fx = Frame::fx; P.fy = Frame::fy; P.cx = Frame::cx; P.cy = Frame::cy; P.bf = pFrame->mbf;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) P.Tcw[r * 4 + c] = pFrame->mTcw.at<float>(r, c);
    std::vector<uint8_t> outlier(n + 1);
    aos2_pose_result_t R;
    memset(&R, 0, sizeof(R));
    R.outlier = outlier.data();
    aos2::ShimTiming &T = aos2::last_shim_timing();
    T.gather_us = clk.lap();
    if (aos2_pose_optimization(aos2::optimizer_handle(), &P, &R, 1) != AOS2_OK) {
        aos2::report("PoseOptimization");   // the frame keeps its pose guess and counts no inliers: Tracking treats it as a lost frame
        return 0;
    }
    T.call_us = clk.lap();
    if (n < 3) {   // :355-356: return 0 before the pose is touched
        T.scatter_us = clk.lap();
        return 0;
    }
    for (int k = 0; k < n; ++k) pFrame->mvbOutlier[vnIndexEdge[k]] = outlier[k] != 0;
    cv::Mat pose(4, 4, CV_32F);   // Converter::toCvMat(SE3quat_recov); pFrame->SetPose(pose) :446-447
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) pose.at<float>(r, c) = R.Tcw[r * 4 + c];
    pFrame->SetPose(pose);
    pFrame->nBadPoseOpt = R.n_bad;   // :449
    T.scatter_us = clk.lap();
    return R.n_inliers;
}

}  // namespace ORB_SLAM2
