// ORB_SLAM2::Optimizer with the REFERENCE's signatures for the two hot methods (include/Optimizer.h:45-47):
//   void static LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap);
//   int  static PoseOptimization(Frame *pFrame);
// The bodies are what a maintainer puts into src/Optimizer.cc: the window gathering of :457-505 and the vertex / edge
// emission of :520-654 fill the POD problem of include/aos2.h (instead of allocating a g2o graph), ONE C-ABI call
// replaces :656-744, and the write-back of :746-778 consumes its result.  LocalMapping.cc:81 and Tracking.cc:870,
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

// src/Optimizer.cc:454-779
inline void Optimizer::LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap)
{
    aos2::ShimClock clk;
    // the window: the current keyframe and its covisible keyframes, marked with the current keyframe's id (:457-469)
    std::list<KeyFrame *> lLocalKeyFrames;
    lLocalKeyFrames.push_back(pKF);
    pKF->mnBALocalForKF = pKF->mnId;
    const std::vector<KeyFrame *> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
    for (int i = 0, iend = (int)vNeighKFs.size(); i < iend; i++) {
        KeyFrame *pKFi = vNeighKFs[i];
        pKFi->mnBALocalForKF = pKF->mnId;
        if (!pKFi->isBad()) lLocalKeyFrames.push_back(pKFi);
    }
    // the map points those keyframes observe, each once, in the order their feature lists name them (:471-488)
    std::list<MapPoint *> lLocalMapPoints;
    for (KeyFrame *pKFi : lLocalKeyFrames) {
        std::vector<MapPoint *> vpMPs = pKFi->GetMapPointMatches();
        for (MapPoint *pMP : vpMPs)
            if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->mnId) {
                lLocalMapPoints.push_back(pMP);
                pMP->mnBALocalForKF = pKF->mnId;
            }
    }
    // keyframes outside the window that observe a window point: they enter with a constant pose (:490-505)
    std::list<KeyFrame *> lFixedCameras;
    for (MapPoint *pMP : lLocalMapPoints) {
        std::map<KeyFrame *, size_t> observations = pMP->GetObservations();
        for (auto &mit : observations) {
            KeyFrame *pKFi = mit.first;
            if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
                pKFi->mnBAFixedForKF = pKF->mnId;
                if (!pKFi->isBad()) lFixedCameras.push_back(pKFi);
            }
        }
    }
    // vertices (:520-546): local keyframes (fixed iff mnId == 0), then the fixed cameras
    std::vector<KeyFrame *> kfs;
    std::vector<float> pose_Tcw;
    std::vector<uint8_t> pose_fixed;
    std::vector<int64_t> pose_id;
    std::map<KeyFrame *, int32_t> kf_index;
    auto add_kf = [&](KeyFrame *pKFi, bool fixed) {
        kf_index[pKFi] = (int32_t)kfs.size();
        kfs.push_back(pKFi);
        const cv::Mat T = pKFi->GetPose();
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) pose_Tcw.push_back(T.at<float>(r, c));
        pose_fixed.push_back(fixed ? 1 : 0);
        pose_id.push_back((int64_t)pKFi->mnId);
    };
    for (KeyFrame *pKFi : lLocalKeyFrames) add_kf(pKFi, pKFi->mnId == 0);
    for (KeyFrame *pKFi : lFixedCameras) add_kf(pKFi, true);
    // map point vertices and edges (:573-654).  Observations are visited in ascending KeyFrame::mnId instead of the
    // pointer order of std::map<KeyFrame*, size_t> (DESIGN.md convention 2: pointer order is not reproducible).
    std::vector<MapPoint *> mps(lLocalMapPoints.begin(), lLocalMapPoints.end());
    std::vector<float> point_xyz, edge_obs, edge_is2;
    std::vector<int64_t> point_id;
    std::vector<int32_t> edge_pose, edge_point;
    std::vector<uint8_t> edge_stereo;
    std::vector<KeyFrame *> vpEdgeKF;
    std::vector<MapPoint *> vpMapPointEdge;
    point_xyz.reserve(mps.size() * 3);
    for (size_t j = 0; j < mps.size(); ++j) {
        MapPoint *pMP = mps[j];
        const cv::Mat X = pMP->GetWorldPos();
        for (int k = 0; k < 3; ++k) point_xyz.push_back(X.at<float>(k));
        point_id.push_back((int64_t)pMP->mnId);
        const std::map<KeyFrame *, size_t> observations = pMP->GetObservations();
        std::vector<std::pair<KeyFrame *, size_t>> obs(observations.begin(), observations.end());
        std::sort(obs.begin(), obs.end(), [](const std::pair<KeyFrame *, size_t> &a, const std::pair<KeyFrame *, size_t> &b) {
            return a.first->mnId < b.first->mnId;
        });
        for (auto &mit : obs) {
            KeyFrame *pKFi = mit.first;
            if (pKFi->isBad()) continue;
            auto where = kf_index.find(pKFi);
            if (where == kf_index.end()) continue;   // (a bad fixed camera was not given a vertex: g2o drops the edge)
            const cv::KeyPoint &kpUn = pKFi->mvKeysUn[mit.second];
            const float kp_ur = pKFi->mvuRight[mit.second];
            edge_pose.push_back(where->second);
            edge_point.push_back((int32_t)j);
            edge_obs.push_back(kpUn.pt.x); edge_obs.push_back(kpUn.pt.y); edge_obs.push_back(kp_ur);
            edge_stereo.push_back(kp_ur < 0 ? 0 : 1);                 // :595
            edge_is2.push_back(pKFi->mvInvLevelSigma2[kpUn.octave]);  // :606, :632
            vpEdgeKF.push_back(pKFi);
            vpMapPointEdge.push_back(pMP);
        }
    }
    aos2::ShimTiming &T = aos2::last_shim_timing();
    T.gather_us = clk.lap();
    if (pbStopFlag && *pbStopFlag) return;   // :656-658
    if (kfs.empty() || mps.empty() || edge_pose.empty()) return;
    aos2_lba_problem_t P;
    P.n_poses = (int32_t)kfs.size(); P.n_points = (int32_t)mps.size(); P.n_edges = (int32_t)edge_pose.size();
    P.pose_Tcw = pose_Tcw.data(); P.pose_fixed = pose_fixed.data(); P.pose_id = pose_id.data();
    P.point_xyz = point_xyz.data(); P.point_id = point_id.data();
    P.edge_pose = edge_pose.data(); P.edge_point = edge_point.data(); P.edge_obs = edge_obs.data();
    P.edge_stereo = edge_stereo.data(); P.edge_inv_sigma2 = edge_is2.data();
    P.fx = pKF->fx; P.fy = pKF->fy; P.cx = pKF->cx; P.cy = pKF->cy; P.bf = pKF->mbf;   // (one camera: pKFi->fx .. mbf)
    P.stop_flag = reinterpret_cast<const volatile uint8_t *>(pbStopFlag);
    P.iters_first = 5; P.iters_second = 10;   // :661, :708
    std::vector<float> out_T(pose_Tcw.size()), out_X(point_xyz.size());
    std::vector<uint8_t> outlier(edge_pose.size());
    aos2_lba_result_t R;
    memset(&R, 0, sizeof(R));
    R.pose_Tcw = out_T.data(); R.point_xyz = out_X.data(); R.edge_outlier = outlier.data(); R.edge_chi2 = nullptr;
    aos2::LbaRecord &rec = aos2::lba_record();
    if (rec.enabled) {
        rec.pose_id = pose_id;
        rec.point_id = point_id;
        rec.edge_pose_id.clear();
        rec.edge_point_id.clear();
        for (size_t e = 0; e < edge_pose.size(); ++e) {
            rec.edge_pose_id.push_back(pose_id[edge_pose[e]]);
            rec.edge_point_id.push_back(point_id[edge_point[e]]);
        }
    }
    clk.lap();
    const int st = aos2_lba_solve(aos2::optimizer_handle(), &P, &R);
    T.call_us = clk.lap();
    if (st == AOS2_ERR_STOPPED) return;
    if (st != AOS2_OK) aos2::fail("LocalBundleAdjustment");
    // Check inlier observations (:712-744), erase under the map mutex (:746-757)
    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
    for (size_t e = 0; e < outlier.size(); ++e) {
        MapPoint *pMP = vpMapPointEdge[e];
        if (pMP->isBad() || !outlier[e]) continue;
        vpEdgeKF[e]->EraseMapPointMatch(pMP);
        pMP->EraseObservation(vpEdgeKF[e]);
    }
    // Recover optimized data (:761-778)
    size_t k = 0;
    for (KeyFrame *pKFi : lLocalKeyFrames) {
        cv::Mat Tcw(4, 4, CV_32F);
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) Tcw.at<float>(r, c) = out_T[k * 16 + r * 4 + c];
        pKFi->SetPose(Tcw);
        ++k;
    }
    for (size_t j = 0; j < mps.size(); ++j) {
        cv::Mat X(3, 1, CV_32F);
        for (int c = 0; c < 3; ++c) X.at<float>(c) = out_X[j * 3 + c];
        mps[j]->SetWorldPos(X);
        mps[j]->UpdateNormalAndDepth();
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
    if (aos2_pose_optimization(aos2::optimizer_handle(), &P, &R, 1) != AOS2_OK)
        aos2::fail("PoseOptimization");
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
