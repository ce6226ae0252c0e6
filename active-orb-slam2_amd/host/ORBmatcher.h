// ORB_SLAM2::ORBmatcher with the REFERENCE's class surface (include/ORBmatcher.h:37-102): every public method at its
// reference signature, so that Tracking.cc, LocalMapping.cc and LoopClosing.cc call it unchanged.
// Each body is what a maintainer puts into src/ORBmatcher.cc in place of the reference loop: it snapshots the members
// the loop reads into the SoA views of include/aos2.h, makes ONE C-ABI call (the search: projection,
// windows, Hamming distances, ratio / orientation tests run on the GPU), and writes the returned indices back into the
// pointer graph.  The map bookkeeping at the end of Fuse (Replace / AddObservation / AddMapPoint) stays host code on the
// pointer graph, as in the reference.
// Include AFTER the headers that declare Frame, KeyFrame, MapPoint (the reference's, or tests/cpp/refstub/slam_stub.h),
// i.e. where the reference's own ORBmatcher.h is included.  NOTHING of the reference's data model changes: the raw
// mfMinDistance / mfMaxDistance (protected; the searches gate on 0.8f / 1.2f of them and MapPoint::PredictScale divides the raw
// maximum, src/MapPoint.cc:413-459, so both factors are applied on the device from the raw values) are read through
// aos2::MapPointDistances (aos2_handles.h: a member pointer formed through a derived class, under mMutexPos).
#pragma once
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <set>
#include <utility>
#include <vector>

#include "aos2_handles.h"

namespace ORB_SLAM2 {

class ORBmatcher {
public:
    static const int TH_LOW = AOS2_TH_LOW;              // src/ORBmatcher.cc:38
    static const int TH_HIGH = AOS2_TH_HIGH;            // :37
    static const int HISTO_LENGTH = AOS2_HISTO_LENGTH;  // :39

    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    // Computes the Hamming distance between two ORB descriptors (:1647-1663)
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return aos2_descriptor_distance(a.ptr<uint8_t>(), b.ptr<uint8_t>()); }

    // Tracking::SearchLocalPoints (src/Tracking.cc:1371)
    int SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th = 3);
    // Tracking::TrackWithMotionModel (:981, :987)
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono);
    // Tracking::Relocalization (:1632, :1646)
    int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th, const int ORBdist);
    // LoopClosing::ComputeSim3 (src/LoopClosing.cc:377)
    int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched, int th);
    // Tracking::TrackReferenceKeyFrame (:862), Relocalization (:1557)
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches);
    // LoopClosing::ComputeSim3 (src/LoopClosing.cc:267)
    int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12);
    // Tracking::MonocularInitialization (:695)
    int SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize = 10);
    // LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:272)
    int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t>> &vMatchedPairs,
                               const bool bOnlyStereo);
    // LoopClosing::ComputeSim3 (src/LoopClosing.cc:325)
    int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12,
                     const cv::Mat &t12, const float th);
    // LocalMapping::SearchInNeighbors (src/LocalMapping.cc:493, :518)
    int Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th = 3.0);
    // LoopClosing::SearchAndFuse (src/LoopClosing.cc:601)
    int Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint);

protected:
    float mfNNratio;
    bool mbCheckOrientation;

    aos2_matcher_t *handle() const { return aos2::matcher_handle(mfNNratio, mbCheckOrientation); }

    // ---- SoA snapshots.  A view owns the arrays its aos2_frame_view_t points into.
    struct View {
        std::vector<float> kp_x, kp_y, kp_angle, bounds;
        std::vector<int32_t> kp_octave, grid_off, grid_idx;
        std::vector<uint8_t> desc, state;
        aos2_frame_view_t view;
    };
    static const uint8_t *rows32(const cv::Mat &D, int n, std::vector<uint8_t> &packed)
    {
        if (n <= 0 || D.step == 32) return D.ptr<uint8_t>();
        packed.resize((size_t)n * 32);   // non-continuous rows: pack
        for (int i = 0; i < n; ++i) memcpy(&packed[(size_t)i * 32], D.ptr<uint8_t>(i), 32);
        return packed.data();
    }
    // keys / descriptors / grid of a Frame or a KeyFrame (same member names); `grid(ix, iy)` returns the cell's index list
    template <class Keys, class GridFn>
    static void fill_view(View &S, int N, const Keys &keysUn, const cv::Mat &descriptors, const std::vector<float> &uRight,
                          const std::vector<float> &scaleFactors, float minX, float minY, float maxX, float maxY, float gwInv,
                          float ghInv, GridFn grid)
    {
        S.kp_x.resize(N); S.kp_y.resize(N); S.kp_angle.resize(N); S.kp_octave.resize(N);
        S.state.assign(N > 0 ? N : 1, 0);
        for (int i = 0; i < N; ++i) {
            const cv::KeyPoint &kp = keysUn[i];
            S.kp_x[i] = kp.pt.x; S.kp_y[i] = kp.pt.y; S.kp_angle[i] = kp.angle; S.kp_octave[i] = kp.octave;
        }
        S.grid_off.assign(FRAME_GRID_COLS * FRAME_GRID_ROWS + 1, 0);
        S.grid_idx.clear();
        S.grid_idx.reserve(N);
        for (int ix = 0; ix < FRAME_GRID_COLS; ++ix)
            for (int iy = 0; iy < FRAME_GRID_ROWS; ++iy) {
                for (std::size_t k : grid(ix, iy)) S.grid_idx.push_back((int32_t)k);
                S.grid_off[ix * FRAME_GRID_ROWS + iy + 1] = (int32_t)S.grid_idx.size();
            }
        if (S.grid_idx.empty()) S.grid_idx.push_back(0);
        aos2_frame_view_t &v = S.view;
        v.n_f = N;
        v.desc_f = rows32(descriptors, N, S.desc);
        v.kp_x = S.kp_x.data(); v.kp_y = S.kp_y.data(); v.kp_octave = S.kp_octave.data(); v.kp_angle = S.kp_angle.data();
        v.u_right = uRight.data();
        v.scale_factors = scaleFactors.data();
        v.n_levels = (int32_t)scaleFactors.size();
        v.min_x = minX; v.min_y = minY; v.max_x = maxX; v.max_y = maxY;
        v.grid_w_inv = gwInv; v.grid_h_inv = ghInv;
        v.grid_off = S.grid_off.data(); v.grid_idx = S.grid_idx.data();
        v.f_mp_state = S.state.data();
    }
    // Frame: f_mp_state = the skip test of :87-89 / :1403-1405 / :1541 on entry
    static void snapshot(Frame &F, View &S)
    {
        fill_view(S, F.N, F.mvKeysUn, F.mDescriptors, F.mvuRight, F.mvScaleFactors, Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX,
                  Frame::mnMaxY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv,
                  [&F](int ix, int iy) -> const std::vector<std::size_t> & { return F.mGrid[ix][iy]; });
        for (int i = 0; i < F.N; ++i) {
            MapPoint *pMP = F.mvpMapPoints[i];
            S.state[i] = !pMP ? 0 : (pMP->Observations() > 0 ? 2 : 1);
        }
    }
    // KeyFrame: KeyFrame::mGrid is PROTECTED in the reference (include/KeyFrame.h:206,223; the reference's matcher goes through
    // KeyFrame::GetFeaturesInArea) and ORBmatcher is no friend, so the cell lists are rebuilt from public members with the rule
    // that made them -- Frame::AssignFeaturesToGrid / Frame::PosInGrid (src/Frame.cc:262-274, 411-421) on mvKeysUn with Frame's
    // static bounds (KeyFrame::KeyFrame copies F.mGrid, src/KeyFrame.cc:48-54; the keyframe's own mnMinX / mnMinY are the same
    // bounds truncated to int, used below only where KeyFrame::GetFeaturesInArea uses them).  Ascending feature index inside a
    // cell = the push_back order of AssignFeaturesToGrid.
    static void snapshot(KeyFrame *pKF, View &S)
    {
        static thread_local std::vector<std::vector<std::size_t>> cells;
        cells.resize((std::size_t)FRAME_GRID_COLS * FRAME_GRID_ROWS);
        for (auto &c : cells) c.clear();
        for (int i = 0; i < pKF->N; ++i) {
            const cv::KeyPoint &kp = pKF->mvKeysUn[i];
            const int posX = (int)std::round((kp.pt.x - Frame::mnMinX) * Frame::mfGridElementWidthInv);
            const int posY = (int)std::round((kp.pt.y - Frame::mnMinY) * Frame::mfGridElementHeightInv);
            if (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) continue;
            cells[(std::size_t)posX * FRAME_GRID_ROWS + posY].push_back((std::size_t)i);
        }
        fill_view(S, pKF->N, pKF->mvKeysUn, pKF->mDescriptors, pKF->mvuRight, pKF->mvScaleFactors, (float)pKF->mnMinX, (float)pKF->mnMinY,
                  (float)pKF->mnMaxX, (float)pKF->mnMaxY, pKF->mfGridElementWidthInv, pKF->mfGridElementHeightInv,
                  [](int ix, int iy) -> const std::vector<std::size_t> & { return cells[(std::size_t)ix * FRAME_GRID_ROWS + iy]; });
    }

    // a set of map points to project: the per-point members every projection method reads
    struct Points {
        std::vector<uint8_t> valid, desc;
        std::vector<float> pos, max_dist, min_dist, normal, q_angle;
        aos2_proj_points_t P;
        void resize(size_t n)
        {
            valid.assign(n + 1, 0); desc.assign((n + 1) * 32, 0); pos.assign((n + 1) * 3, 0.0f); max_dist.assign(n + 1, 0.0f);
            min_dist.assign(n + 1, 0.0f); normal.assign((n + 1) * 3, 0.0f); q_angle.assign(n + 1, 0.0f);
            memset(&P, 0, sizeof(P));
            P.n_pts = (int32_t)n;
            P.valid = valid.data(); P.desc = desc.data(); P.pos = pos.data(); P.max_dist = max_dist.data(); P.min_dist = min_dist.data();
            P.normal = normal.data(); P.q_angle = q_angle.data();
        }
        void set(size_t i, MapPoint *pMP, bool with_normal)
        {
            valid[i] = 1;
            const cv::Mat X = pMP->GetWorldPos();
            for (int k = 0; k < 3; ++k) pos[i * 3 + k] = X.at<float>(k);
            aos2::MapPointDistances<MapPoint>::get(pMP, min_dist[i], max_dist[i]);   // raw mfMinDistance / mfMaxDistance (header comment)
            if (with_normal) {
                const cv::Mat n = pMP->GetNormal();
                for (int k = 0; k < 3; ++k) normal[i * 3 + k] = n.at<float>(k);
            }
            const cv::Mat d = pMP->GetDescriptor();
            memcpy(&desc[i * 32], d.ptr<uint8_t>(), 32);
        }
    };
    static void put3x3(const cv::Mat &R, float *dst)
    {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) dst[r * 3 + c] = R.at<float>(r, c);
    }
    static void put3(const cv::Mat &t, float *dst)
    {
        for (int k = 0; k < 3; ++k) dst[k] = t.at<float>(k);
    }
    // DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) -> CSR, node ids ascending (the map's order)
    struct FeatCsr {
        std::vector<int32_t> id, off, idx;
        void fill(const DBoW2::FeatureVector &fv)
        {
            id.clear(); off.assign(1, 0); idx.clear();
            for (const auto &kv : fv) {
                id.push_back((int32_t)kv.first);
                for (unsigned int v : kv.second) idx.push_back((int32_t)v);
                off.push_back((int32_t)idx.size());
            }
            if (id.empty()) id.push_back(0);
            if (idx.empty()) idx.push_back(0);
        }
    };
};

// src/ORBmatcher.cc:45-129
inline int ORBmatcher::SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th)
{
    aos2::ShimClock clk;
    const size_t n = vpMapPoints.size();
    View S;
    snapshot(F, S);
    std::vector<uint8_t> in_view(n + 1), has_obs(n + 1), desc((n + 1) * 32);
    std::vector<int32_t> level(n + 1);
    std::vector<float> vcos(n + 1), px(n + 1), py(n + 1), pxr(n + 1);
    for (size_t i = 0; i < n; ++i) {
        MapPoint *pMP = vpMapPoints[i];
        in_view[i] = (pMP->mbTrackInView && !pMP->isBad()) ? 1 : 0;   // :52-56
        if (!in_view[i]) continue;
        level[i] = pMP->mnTrackScaleLevel;
        vcos[i] = pMP->mTrackViewCos;
        px[i] = pMP->mTrackProjX; py[i] = pMP->mTrackProjY; pxr[i] = pMP->mTrackProjXR;
        const cv::Mat d = pMP->GetDescriptor();
        memcpy(&desc[i * 32], d.ptr<uint8_t>(), 32);
        has_obs[i] = pMP->Observations() > 0 ? 1 : 0;
    }
    aos2_proj_mp_t P;
    P.n_mp = (int32_t)n;
    P.track_in_view = in_view.data(); P.pred_level = level.data(); P.view_cos = vcos.data();
    P.proj_x = px.data(); P.proj_y = py.data(); P.proj_xr = pxr.data(); P.desc = desc.data(); P.has_obs = has_obs.data();
    std::vector<int32_t> match(F.N > 0 ? F.N : 1, -1);
    int32_t nmatches = 0;
    aos2::ShimTiming &T = aos2::last_shim_timing();
    T.gather_us = clk.lap();
    aos2::check(aos2_matcher_search_by_projection(handle(), &S.view, &P, th, match.data(), &nmatches), "SearchByProjection");
    T.call_us = clk.lap();
    for (int j = 0; j < F.N; ++j)
        if (match[j] >= 0) F.mvpMapPoints[j] = vpMapPoints[match[j]];   // :94
    T.scatter_us = clk.lap();
    return nmatches;
}

// src/ORBmatcher.cc:1328-1470
inline int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
{
    aos2::ShimClock clk;
    View S;
    snapshot(CurrentFrame, S);
    const int n = LastFrame.N;
    std::vector<uint8_t> valid(n + 1), has_obs(n + 1), desc((size_t)(n + 1) * 32);
    std::vector<float> pos((size_t)(n + 1) * 3), angle(n + 1);
    std::vector<int32_t> octave(n + 1);
    for (int i = 0; i < n; ++i) {
        MapPoint *pMP = LastFrame.mvpMapPoints[i];
        valid[i] = (pMP && !LastFrame.mvbOutlier[i]) ? 1 : 0;   // :1355-1358
        octave[i] = LastFrame.mvKeys[i].octave;                  // :1376
        angle[i] = LastFrame.mvKeysUn[i].angle;                  // :1436
        if (!valid[i]) continue;
        const cv::Mat x3Dw = pMP->GetWorldPos();
        for (int k = 0; k < 3; ++k) pos[(size_t)i * 3 + k] = x3Dw.at<float>(k);
        const cv::Mat d = pMP->GetDescriptor();
        memcpy(&desc[(size_t)i * 32], d.ptr<uint8_t>(), 32);
        has_obs[i] = pMP->Observations() > 0 ? 1 : 0;
    }
    aos2_proj_last_t P;
    P.n_last = n;
    P.last_valid = valid.data(); P.world_pos = pos.data(); P.desc = desc.data(); P.last_octave = octave.data();
    P.last_angle = angle.data(); P.has_obs = has_obs.data();
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            P.Tcw[r * 4 + c] = CurrentFrame.mTcw.at<float>(r, c);
            P.Tlw[r * 4 + c] = LastFrame.mTcw.at<float>(r, c);
        }
    P.fx = Frame::fx; P.fy = Frame::fy; P.cx = Frame::cx; P.cy = Frame::cy; P.mb = CurrentFrame.mb; P.mbf = CurrentFrame.mbf;
    std::vector<int32_t> match(CurrentFrame.N > 0 ? CurrentFrame.N : 1, -1);
    int32_t nmatches = 0;
    aos2::ShimTiming &T = aos2::last_shim_timing();
    T.gather_us = clk.lap();
    aos2::check(aos2_matcher_search_by_projection_last(handle(), &S.view, &P, th, bMono ? 1 : 0, match.data(), &nmatches), "SearchByProjection");
    T.call_us = clk.lap();
    for (int j = 0; j < CurrentFrame.N; ++j) {
        if (match[j] >= 0)
            CurrentFrame.mvpMapPoints[j] = LastFrame.mvpMapPoints[match[j]];   // :1432
        else if (match[j] == -2)
            CurrentFrame.mvpMapPoints[j] = static_cast<MapPoint *>(NULL);      // rotation check, :1459
    }
    T.scatter_us = clk.lap();
    return nmatches;
}

// src/ORBmatcher.cc:1472-1599 (relocalisation: the keyframe's map points projected into the frame with its PnP pose)
inline int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th,
                                          const int ORBdist)
{
    aos2::ShimClock clk;
    // the frame's pose as rotation, translation and camera centre (:1476-1478; these three cv::Mat lines are the caller-side
    // preparation the C ABI leaves on the host, include/aos2.h "Projection family")
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat Ow = -Rcw.t() * tcw;
    View S;
    snapshot(CurrentFrame, S);
    for (int i = 0; i < CurrentFrame.N; ++i) S.state[i] = CurrentFrame.mvpMapPoints[i] ? 1 : 0;   // the skip test of :1541
    const std::vector<MapPoint *> vpMPs = pKF->GetMapPointMatches();
    Points Q;
    Q.resize(vpMPs.size());
    for (size_t i = 0; i < vpMPs.size(); ++i) {
        MapPoint *pMP = vpMPs[i];
        Q.q_angle[i] = pKF->mvKeysUn[i].angle;   // :1557
        if (pMP && !pMP->isBad() && !sAlreadyFound.count(pMP)) Q.set(i, pMP, false);   // :1492-1496
    }
    put3x3(Rcw, Q.P.R); put3(tcw, Q.P.t); put3(Ow, Q.P.Ow);
    Q.P.fx = Frame::fx; Q.P.fy = Frame::fy; Q.P.cx = Frame::cx; Q.P.cy = Frame::cy; Q.P.bf = CurrentFrame.mbf;
    Q.P.log_scale_factor = CurrentFrame.mfLogScaleFactor;
    Q.P.inv_level_sigma2 = CurrentFrame.mvInvLevelSigma2.data();
    Q.P.th = th;
    std::vector<int32_t> match(CurrentFrame.N > 0 ? CurrentFrame.N : 1, -1);
    int32_t nmatches = 0;
    aos2::ShimTiming &T = aos2::last_shim_timing();
    T.gather_us = clk.lap();
    aos2::check(aos2_matcher_search_by_projection_reloc(handle(), &S.view, &Q.P, ORBdist, match.data(), &nmatches), "SearchByProjection");
    T.call_us = clk.lap();
    for (int j = 0; j < CurrentFrame.N; ++j) {
        if (match[j] >= 0)
            CurrentFrame.mvpMapPoints[j] = vpMPs[match[j]];                  // :1553
        else if (match[j] == -2)
            CurrentFrame.mvpMapPoints[j] = static_cast<MapPoint *>(NULL);    // rotation check, :1590
    }
    T.scatter_us = clk.lap();
    return nmatches;
}

namespace shim_detail {
// Scw = [s R | s t] -> R, t, camera centre (:299-304, :986-991): the caller-side cv::Mat lines
struct Sim3Pose {
    cv::Mat Rcw, tcw, Ow;
    explicit Sim3Pose(const cv::Mat &Scw)
    {
        const cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
        const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
        Rcw = sRcw / scw;
        tcw = Scw.rowRange(0, 3).col(3) / scw;
        Ow = -Rcw.t() * tcw;
    }
};
}  // namespace shim_detail

// src/ORBmatcher.cc:290-403
inline int ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched,
                                          int th)
{
    aos2::ShimClock clk;
    const shim_detail::Sim3Pose pose(Scw);
    std::set<MapPoint *> spAlreadyFound(vpMatched.begin(), vpMatched.end());   // :307-308
    spAlreadyFound.erase(static_cast<MapPoint *>(NULL));
    View S;
    snapshot(pKF, S);
    for (int i = 0; i < pKF->N && i < (int)vpMatched.size(); ++i) S.state[i] = vpMatched[i] ? 1 : 0;   // :375-376
    Points Q;
    Q.resize(vpPoints.size());
    for (size_t i = 0; i < vpPoints.size(); ++i) {
        MapPoint *pMP = vpPoints[i];
        if (!pMP->isBad() && !spAlreadyFound.count(pMP)) Q.set(i, pMP, true);   // :318-319
    }
    put3x3(pose.Rcw, Q.P.R); put3(pose.tcw, Q.P.t); put3(pose.Ow, Q.P.Ow);
    Q.P.fx = pKF->fx; Q.P.fy = pKF->fy; Q.P.cx = pKF->cx; Q.P.cy = pKF->cy; Q.P.bf = pKF->mbf;
    Q.P.log_scale_factor = pKF->mfLogScaleFactor;
    Q.P.inv_level_sigma2 = pKF->mvInvLevelSigma2.data();
    Q.P.th = (float)th;
    std::vector<int32_t> match(pKF->N > 0 ? pKF->N : 1, -1);
    int32_t nmatches = 0;
    aos2::ShimTiming &T = aos2::last_shim_timing();
    T.gather_us = clk.lap();
    aos2::check(aos2_matcher_search_by_projection_kf(handle(), &S.view, &Q.P, match.data(), &nmatches), "SearchByProjection");
    T.call_us = clk.lap();
    for (int j = 0; j < pKF->N; ++j)
        if (match[j] >= 0) vpMatched[j] = vpPoints[match[j]];   // :396
    T.scatter_us = clk.lap();
    return nmatches;
}

// src/ORBmatcher.cc:159-288
inline int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches)
{
    aos2::ShimClock clk;
    const std::vector<MapPoint *> vpMapPointsKF = pKF->GetMapPointMatches();   // :161
    vpMapPointMatches = std::vector<MapPoint *>(F.N, static_cast<MapPoint *>(NULL));   // :163
    const int nkf = (int)vpMapPointsKF.size();
    std::vector<uint8_t> has_mp(nkf + 1), dk, df;
    std::vector<float> angle_kf(nkf + 1), angle_f(F.N + 1);
    for (int i = 0; i < nkf; ++i) {
        MapPoint *pMP = vpMapPointsKF[i];
        has_mp[i] = (pMP && !pMP->isBad()) ? 1 : 0;   // :194-198
        angle_kf[i] = pKF->mvKeysUn[i].angle;          // :245
    }
    for (int j = 0; j < F.N; ++j) angle_f[j] = F.mvKeys[j].angle;
    FeatCsr ck, cf;
    ck.fill(pKF->mFeatVec);
    cf.fill(F.mFeatVec);
    aos2_bow_pair_t pair;
    pair.n_kf = nkf; pair.n_f = F.N;
    pair.desc_kf = rows32(pKF->mDescriptors, nkf, dk); pair.desc_f = rows32(F.mDescriptors, F.N, df);
    pair.kf_has_mp = has_mp.data(); pair.angle_kf = angle_kf.data(); pair.angle_f = angle_f.data();
    pair.n_nodes_kf = (int32_t)pKF->mFeatVec.size(); pair.n_nodes_f = (int32_t)F.mFeatVec.size();
    pair.node_id_kf = ck.id.data(); pair.node_off_kf = ck.off.data(); pair.node_idx_kf = ck.idx.data();
    pair.node_id_f = cf.id.data(); pair.node_off_f = cf.off.data(); pair.node_idx_f = cf.idx.data();
    std::vector<int32_t> match(F.N > 0 ? F.N : 1, -1);
    int32_t *mptr = match.data();
    int32_t nmatches = 0;
    aos2::ShimTiming &T = aos2::last_shim_timing();
    T.gather_us = clk.lap();
    aos2::check(aos2_matcher_search_by_bow(handle(), &pair, 1, &mptr, &nmatches), "SearchByBoW");
    T.call_us = clk.lap();
    for (int j = 0; j < F.N; ++j)
        if (match[j] >= 0) vpMapPointMatches[j] = vpMapPointsKF[match[j]];   // :239
    T.scatter_us = clk.lap();
    return nmatches;
}

// src/ORBmatcher.cc:522-655
inline int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12)
{
    aos2::ShimClock clk;
    const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    const int n1 = (int)vpMapPoints1.size(), n2 = (int)vpMapPoints2.size();
    vpMatches12 = std::vector<MapPoint *>(n1, static_cast<MapPoint *>(NULL));   // :534
    std::vector<uint8_t> has1(n1 + 1), has2(n2 + 1), d1, d2;
    std::vector<float> a1(n1 + 1), a2(n2 + 1);
    for (int i = 0; i < n1; ++i) {
        has1[i] = (vpMapPoints1[i] && !vpMapPoints1[i]->isBad()) ? 1 : 0;   // :560-564
        a1[i] = pKF1->mvKeysUn[i].angle;
    }
    for (int i = 0; i < n2; ++i) {
        has2[i] = (vpMapPoints2[i] && !vpMapPoints2[i]->isBad()) ? 1 : 0;   // :576-584
        a2[i] = pKF2->mvKeysUn[i].angle;
    }
    FeatCsr c1, c2;
    c1.fill(pKF1->mFeatVec);
    c2.fill(pKF2->mFeatVec);
    aos2_bow_kf_pair_t pair;
    pair.n1 = n1; pair.n2 = n2;
    pair.desc1 = rows32(pKF1->mDescriptors, n1, d1); pair.desc2 = rows32(pKF2->mDescriptors, n2, d2);
    pair.has_mp1 = has1.data(); pair.has_mp2 = has2.data(); pair.angle1 = a1.data(); pair.angle2 = a2.data();
    pair.n_nodes1 = (int32_t)pKF1->mFeatVec.size(); pair.n_nodes2 = (int32_t)pKF2->mFeatVec.size();
    pair.node_id1 = c1.id.data(); pair.node_off1 = c1.off.data(); pair.node_idx1 = c1.idx.data();
    pair.node_id2 = c2.id.data(); pair.node_off2 = c2.off.data(); pair.node_idx2 = c2.idx.data();
    std::vector<int32_t> match(n1 > 0 ? n1 : 1, -1);
    int32_t *mptr = match.data();
    int32_t nmatches = 0;
    aos2::ShimTiming &T = aos2::last_shim_timing();
    T.gather_us = clk.lap();
    aos2::check(aos2_matcher_search_by_bow_kf(handle(), &pair, 1, &mptr, &nmatches), "SearchByBoW");
    T.call_us = clk.lap();
    for (int i = 0; i < n1; ++i)
        if (match[i] >= 0) vpMatches12[i] = vpMapPoints2[match[i]];   // :604
    T.scatter_us = clk.lap();
    return nmatches;
}

// src/ORBmatcher.cc:405-520
inline int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12,
                                               int windowSize)
{
    aos2::ShimClock clk;
    const int n1 = (int)F1.mvKeysUn.size();
    vnMatches12 = std::vector<int>(n1, -1);   // :408
    View S;
    snapshot(F2, S);
    std::vector<int32_t> oct1(n1 + 1);
    std::vector<float> ang1(n1 + 1), prev((size_t)(n1 + 1) * 2);
    std::vector<uint8_t> d1;
    for (int i = 0; i < n1; ++i) {
        oct1[i] = F1.mvKeysUn[i].octave;
        ang1[i] = F1.mvKeysUn[i].angle;
        prev[(size_t)i * 2] = vbPrevMatched[i].x;
        prev[(size_t)i * 2 + 1] = vbPrevMatched[i].y;
    }
    std::vector<int32_t> match(n1 > 0 ? n1 : 1, -1);
    int32_t nmatches = 0;
    aos2::ShimTiming &T = aos2::last_shim_timing();
    T.gather_us = clk.lap();
    aos2::check(aos2_matcher_search_for_initialization(handle(), &S.view, n1, rows32(F1.mDescriptors, n1, d1), oct1.data(), ang1.data(),
                                                       prev.data(), windowSize, match.data(), &nmatches),
                "SearchForInitialization");
    T.call_us = clk.lap();
    for (int i = 0; i < n1; ++i) {
        vnMatches12[i] = match[i];
        if (match[i] >= 0) vbPrevMatched[i] = F2.mvKeysUn[match[i]].pt;   // :512-515
    }
    T.scatter_us = clk.lap();
    return nmatches;
}

// src/ORBmatcher.cc:657-823
inline int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t>> &vMatchedPairs,
                                              const bool bOnlyStereo)
{
    aos2::ShimClock clk;
    // the epipole of camera 1 in image 2 (:663-670; caller-side cv::Mat lines)
    const cv::Mat Cw = pKF1->GetCameraCenter();
    const cv::Mat C2 = pKF2->GetRotation() * Cw + pKF2->GetTranslation();
    const float invz = 1.0f / C2.at<float>(2);
    const int n1 = pKF1->N, n2 = pKF2->N;
    std::vector<uint8_t> has1(n1 + 1), has2(n2 + 1), d1, d2;
    std::vector<float> x1(n1 + 1), y1(n1 + 1), a1(n1 + 1), x2(n2 + 1), y2(n2 + 1), a2(n2 + 1);
    std::vector<int32_t> oct2(n2 + 1);
    for (int i = 0; i < n1; ++i) {
        has1[i] = pKF1->GetMapPoint(i) ? 1 : 0;   // :702-706
        const cv::KeyPoint &kp = pKF1->mvKeysUn[i];
        x1[i] = kp.pt.x; y1[i] = kp.pt.y; a1[i] = kp.angle;
    }
    for (int i = 0; i < n2; ++i) {
        has2[i] = pKF2->GetMapPoint(i) ? 1 : 0;   // :726-730
        const cv::KeyPoint &kp = pKF2->mvKeysUn[i];
        x2[i] = kp.pt.x; y2[i] = kp.pt.y; a2[i] = kp.angle; oct2[i] = kp.octave;
    }
    FeatCsr c1, c2;
    c1.fill(pKF1->mFeatVec);
    c2.fill(pKF2->mFeatVec);
    aos2_triang_pair_t pair;
    pair.n1 = n1; pair.n2 = n2;
    pair.desc1 = rows32(pKF1->mDescriptors, n1, d1); pair.desc2 = rows32(pKF2->mDescriptors, n2, d2);
    pair.has_mp1 = has1.data(); pair.has_mp2 = has2.data();
    pair.x1 = x1.data(); pair.y1 = y1.data(); pair.angle1 = a1.data(); pair.u_right1 = pKF1->mvuRight.data();
    pair.x2 = x2.data(); pair.y2 = y2.data(); pair.angle2 = a2.data(); pair.u_right2 = pKF2->mvuRight.data();
    pair.octave2 = oct2.data();
    pair.scale_factors2 = pKF2->mvScaleFactors.data(); pair.level_sigma2_2 = pKF2->mvLevelSigma2.data();
    pair.n_levels2 = (int32_t)pKF2->mvScaleFactors.size();
    put3x3(F12, pair.F12);
    pair.ex = pKF2->fx * C2.at<float>(0) * invz + pKF2->cx;
    pair.ey = pKF2->fy * C2.at<float>(1) * invz + pKF2->cy;
    pair.n_nodes1 = (int32_t)pKF1->mFeatVec.size(); pair.n_nodes2 = (int32_t)pKF2->mFeatVec.size();
    pair.node_id1 = c1.id.data(); pair.node_off1 = c1.off.data(); pair.node_idx1 = c1.idx.data();
    pair.node_id2 = c2.id.data(); pair.node_off2 = c2.off.data(); pair.node_idx2 = c2.idx.data();
    std::vector<int32_t> match(n1 > 0 ? n1 : 1, -1);
    int32_t *mptr = match.data();
    int32_t nmatches = 0;
    aos2::ShimTiming &T = aos2::last_shim_timing();
    T.gather_us = clk.lap();
    aos2::check(aos2_matcher_search_for_triangulation(handle(), &pair, 1, bOnlyStereo ? 1 : 0, &mptr, &nmatches), "SearchForTriangulation");
    T.call_us = clk.lap();
    vMatchedPairs.clear();   // :808-820
    vMatchedPairs.reserve(nmatches > 0 ? nmatches : 0);
    for (int i = 0; i < n1; ++i)
        if (match[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)match[i]));
    T.scatter_us = clk.lap();
    return nmatches;
}

// src/ORBmatcher.cc:1102-1326
inline int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12,
                                    const cv::Mat &t12, const float th)
{
    aos2::ShimClock clk;
    // cameras from the world and the similarity between them (:1110-1120; caller-side cv::Mat lines)
    const cv::Mat R1w = pKF1->GetRotation(), t1w = pKF1->GetTranslation();
    const cv::Mat R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
    const cv::Mat sR12 = s12 * R12;
    const cv::Mat sR21 = (1.0 / s12) * R12.t();
    const cv::Mat t21 = -sR21 * t12;
    const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);   // :1128-1142
    for (int i = 0; i < N1; ++i) {
        MapPoint *pMP = vpMatches12[i];
        if (pMP) {
            vbAlreadyMatched1[i] = true;
            const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
    }
    View S1, S2;
    snapshot(pKF1, S1);
    snapshot(pKF2, S2);
    Points Q12, Q21;
    Q12.resize(N1);
    Q21.resize(N2);
    for (int i = 0; i < N1; ++i) {
        MapPoint *pMP = vpMapPoints1[i];
        if (pMP && !vbAlreadyMatched1[i] && !pMP->isBad()) Q12.set(i, pMP, false);   // :1152-1156
    }
    for (int i = 0; i < N2; ++i) {
        MapPoint *pMP = vpMapPoints2[i];
        if (pMP && !vbAlreadyMatched2[i] && !pMP->isBad()) Q21.set(i, pMP, false);   // :1232-1236
    }
    auto camera = [&](Points &Q, const cv::Mat &Rw, const cv::Mat &tw, const cv::Mat &sR, const cv::Mat &tt, KeyFrame *target) {
        put3x3(Rw, Q.P.R); put3(tw, Q.P.t); put3x3(sR, Q.P.R2); put3(tt, Q.P.t2);
        Q.P.fx = pKF1->fx; Q.P.fy = pKF1->fy; Q.P.cx = pKF1->cx; Q.P.cy = pKF1->cy;   // (the reference projects with pKF1's intrinsics both ways, :1105-1108)
        Q.P.bf = pKF1->mbf;
        Q.P.log_scale_factor = target->mfLogScaleFactor;
        Q.P.inv_level_sigma2 = target->mvInvLevelSigma2.data();
        Q.P.th = th;
    };
    camera(Q12, R1w, t1w, sR21, t21, pKF2);
    camera(Q21, R2w, t2w, sR12, t12, pKF1);
    std::vector<int32_t> match(N1 > 0 ? N1 : 1, -1);
    int32_t nFound = 0;
    aos2::ShimTiming &T = aos2::last_shim_timing();
    T.gather_us = clk.lap();
    aos2::check(aos2_matcher_search_by_sim3(handle(), &S1.view, &S2.view, &Q12.P, &Q21.P, match.data(), &nFound), "SearchBySim3");
    T.call_us = clk.lap();
    for (int i1 = 0; i1 < N1; ++i1)
        if (match[i1] >= 0) vpMatches12[i1] = vpMapPoints2[match[i1]];   // :1318
    T.scatter_us = clk.lap();
    return nFound;
}

// src/ORBmatcher.cc:825-975
inline int ORBmatcher::Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th)
{
    aos2::ShimClock clk;
    const cv::Mat Rcw = pKF->GetRotation(), tcw = pKF->GetTranslation(), Ow = pKF->GetCameraCenter();
    View S;
    snapshot(pKF, S);
    const int nMPs = (int)vpMapPoints.size();
    Points Q;
    Q.resize(nMPs);
    for (int i = 0; i < nMPs; ++i) {
        MapPoint *pMP = vpMapPoints[i];
        if (pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKF)) Q.set(i, pMP, true);   // :844-850
    }
    put3x3(Rcw, Q.P.R); put3(tcw, Q.P.t); put3(Ow, Q.P.Ow);
    Q.P.fx = pKF->fx; Q.P.fy = pKF->fy; Q.P.cx = pKF->cx; Q.P.cy = pKF->cy; Q.P.bf = pKF->mbf;
    Q.P.log_scale_factor = pKF->mfLogScaleFactor;
    Q.P.inv_level_sigma2 = pKF->mvInvLevelSigma2.data();
    Q.P.th = th;
    std::vector<int32_t> best_idx(nMPs + 1, -1), best_dist(nMPs + 1, 256);
    int32_t n_search = 0;
    aos2::ShimTiming &T = aos2::last_shim_timing();
    T.gather_us = clk.lap();
    aos2::check(aos2_matcher_fuse(handle(), &S.view, &Q.P, 0, best_idx.data(), best_dist.data(), &n_search), "Fuse");
    T.call_us = clk.lap();
    // the map bookkeeping of :948-969, on the pointer graph and in the loop's order.  The search result of a point does not
    // depend on the points before it, but its gate does: a point that an earlier iteration made bad or put into this
    // keyframe (Replace / AddMapPoint) is skipped by the reference's loop head, so the gate is evaluated again here.
    int nFused = 0;
    for (int i = 0; i < nMPs; ++i) {
        MapPoint *pMP = vpMapPoints[i];
        if (best_idx[i] < 0 || !Q.valid[i]) continue;
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        const int bestIdx = best_idx[i];
        MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) {
                if (pMPinKF->Observations() > pMP->Observations())
                    pMP->Replace(pMPinKF);
                else
                    pMPinKF->Replace(pMP);
            }
        } else {
            pMP->AddObservation(pKF, bestIdx);
            pKF->AddMapPoint(pMP, bestIdx);
        }
        nFused++;
    }
    T.scatter_us = clk.lap();
    return nFused;
}

// src/ORBmatcher.cc:977-1100
inline int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint)
{
    aos2::ShimClock clk;
    const shim_detail::Sim3Pose pose(Scw);
    const std::set<MapPoint *> spAlreadyFound = pKF->GetMapPoints();   // :994
    View S;
    snapshot(pKF, S);
    const int nPoints = (int)vpPoints.size();
    Points Q;
    Q.resize(nPoints);
    for (int i = 0; i < nPoints; ++i) {
        MapPoint *pMP = vpPoints[i];
        if (!pMP->isBad() && !spAlreadyFound.count(pMP)) Q.set(i, pMP, true);   // :1006-1007
    }
    put3x3(pose.Rcw, Q.P.R); put3(pose.tcw, Q.P.t); put3(pose.Ow, Q.P.Ow);
    Q.P.fx = pKF->fx; Q.P.fy = pKF->fy; Q.P.cx = pKF->cx; Q.P.cy = pKF->cy; Q.P.bf = pKF->mbf;
    Q.P.log_scale_factor = pKF->mfLogScaleFactor;
    Q.P.inv_level_sigma2 = pKF->mvInvLevelSigma2.data();
    Q.P.th = th;
    std::vector<int32_t> best_idx(nPoints + 1, -1), best_dist(nPoints + 1, 256);
    int32_t n_search = 0;
    aos2::ShimTiming &T = aos2::last_shim_timing();
    T.gather_us = clk.lap();
    aos2::check(aos2_matcher_fuse(handle(), &S.view, &Q.P, 1, best_idx.data(), best_dist.data(), &n_search), "Fuse");
    T.call_us = clk.lap();
    int nFused = 0;
    for (int iMP = 0; iMP < nPoints; ++iMP) {   // :1076-1095 (spAlreadyFound is the set taken on entry: the gate does not move)
        if (best_idx[iMP] < 0 || !Q.valid[iMP]) continue;
        MapPoint *pMP = vpPoints[iMP];
        if (pMP->isBad()) continue;
        const int bestIdx = best_idx[iMP];
        MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
        } else {
            pMP->AddObservation(pKF, bestIdx);
            pKF->AddMapPoint(pMP, bestIdx);
        }
        nFused++;
    }
    T.scatter_us = clk.lap();
    return nFused;
}

}  // namespace ORB_SLAM2
