// ORB_SLAM2::ORBmatcher with the REFERENCE's signatures for the hot methods (include/ORBmatcher.h:37-102):
//   int SearchByProjection(Frame &F, const std::vector<MapPoint*> &vpMapPoints, const float th = 3);
//   int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono);
//   int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint*> &vpMapPointMatches);
// Each body is what a maintainer puts into src/ORBmatcher.cc in place of the reference loop: it snapshots the members
// the loop reads into the SoA views of include/aos2.h, makes ONE C-ABI call, and writes the
// returned indices back as MapPoint* -- Tracking.cc / LocalMapping.cc call it unchanged.
// Include AFTER the headers that declare Frame, KeyFrame, MapPoint (the reference's, or tests/cpp/refstub/slam_stub.h).
#pragma once
#include <chrono>
#include <cstdint>
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

    int SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th = 3);
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono);
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches);

protected:
    float mfNNratio;
    bool mbCheckOrientation;

    // the members of a Frame every projection search reads, as the SoA view of include/aos2.h;
    // mGrid[64][48] becomes the CSR the kernels walk (cell = ix * 48 + iy, push_back order kept)
    struct FrameSnapshot {
        std::vector<float> kp_x, kp_y, kp_angle;
        std::vector<int32_t> kp_octave, grid_off, grid_idx;
        std::vector<uint8_t> desc, state;
        aos2_frame_view_t view;
    };
    static void snapshot(Frame &F, FrameSnapshot &S)
    {
        const int N = F.N;
        S.kp_x.resize(N); S.kp_y.resize(N); S.kp_angle.resize(N); S.kp_octave.resize(N);
        S.state.resize(N > 0 ? N : 1);
        for (int i = 0; i < N; ++i) {
            const cv::KeyPoint &kp = F.mvKeysUn[i];
            S.kp_x[i] = kp.pt.x; S.kp_y[i] = kp.pt.y; S.kp_angle[i] = kp.angle; S.kp_octave[i] = kp.octave;
            MapPoint *pMP = F.mvpMapPoints[i];
            S.state[i] = !pMP ? 0 : (pMP->Observations() > 0 ? 2 : 1);   // the skip test of :87-89 / :1403-1405
        }
        S.grid_off.assign(FRAME_GRID_COLS * FRAME_GRID_ROWS + 1, 0);
        S.grid_idx.clear();
        S.grid_idx.reserve(N);
        for (int ix = 0; ix < FRAME_GRID_COLS; ++ix)
            for (int iy = 0; iy < FRAME_GRID_ROWS; ++iy) {
                const std::vector<std::size_t> &cell = F.mGrid[ix][iy];
                for (std::size_t k : cell) S.grid_idx.push_back((int32_t)k);
                S.grid_off[ix * FRAME_GRID_ROWS + iy + 1] = (int32_t)S.grid_idx.size();
            }
        if (S.grid_idx.empty()) S.grid_idx.push_back(0);
        const uint8_t *d = F.mDescriptors.ptr<uint8_t>();
        if (N > 0 && F.mDescriptors.step != 32) {   // non-continuous rows: pack
            S.desc.resize((size_t)N * 32);
            for (int i = 0; i < N; ++i) memcpy(&S.desc[(size_t)i * 32], F.mDescriptors.ptr<uint8_t>(i), 32);
            d = S.desc.data();
        }
        aos2_frame_view_t &v = S.view;
        v.n_f = N;
        v.desc_f = d;
        v.kp_x = S.kp_x.data(); v.kp_y = S.kp_y.data(); v.kp_octave = S.kp_octave.data(); v.kp_angle = S.kp_angle.data();
        v.u_right = F.mvuRight.data();
        v.scale_factors = F.mvScaleFactors.data();
        v.n_levels = (int32_t)F.mvScaleFactors.size();
        v.min_x = Frame::mnMinX; v.min_y = Frame::mnMinY; v.max_x = Frame::mnMaxX; v.max_y = Frame::mnMaxY;
        v.grid_w_inv = Frame::mfGridElementWidthInv; v.grid_h_inv = Frame::mfGridElementHeightInv;
        v.grid_off = S.grid_off.data(); v.grid_idx = S.grid_idx.data();
        v.f_mp_state = S.state.data();
    }
    static void check(int st, const char *what)
    {
        if (st != AOS2_OK) throw std::runtime_error(std::string(what) + ": " + aos2_last_error());
    }
    static double us_since(std::chrono::steady_clock::time_point t0)
    {
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
};

// src/ORBmatcher.cc:45-129
inline int ORBmatcher::SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th)
{
    auto t0 = std::chrono::steady_clock::now();
    const size_t n = vpMapPoints.size();
    FrameSnapshot S;
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
    T.gather_us = us_since(t0);
    t0 = std::chrono::steady_clock::now();
    check(aos2_matcher_search_by_projection(aos2::matcher_handle(mfNNratio, mbCheckOrientation), &S.view, &P, th, match.data(), &nmatches),
          "SearchByProjection");
    T.call_us = us_since(t0);
    t0 = std::chrono::steady_clock::now();
    for (int j = 0; j < F.N; ++j)
        if (match[j] >= 0) F.mvpMapPoints[j] = vpMapPoints[match[j]];   // :94
    T.scatter_us = us_since(t0);
    return nmatches;
}

// src/ORBmatcher.cc:1328-1470
inline int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
{
    auto t0 = std::chrono::steady_clock::now();
    FrameSnapshot S;
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
    T.gather_us = us_since(t0);
    t0 = std::chrono::steady_clock::now();
    check(aos2_matcher_search_by_projection_last(aos2::matcher_handle(mfNNratio, mbCheckOrientation), &S.view, &P, th, bMono ? 1 : 0,
                                                 match.data(), &nmatches),
          "SearchByProjection");
    T.call_us = us_since(t0);
    t0 = std::chrono::steady_clock::now();
    for (int j = 0; j < CurrentFrame.N; ++j) {
        if (match[j] >= 0)
            CurrentFrame.mvpMapPoints[j] = LastFrame.mvpMapPoints[match[j]];   // :1432
        else if (match[j] == -2)
            CurrentFrame.mvpMapPoints[j] = static_cast<MapPoint *>(NULL);      // rotation check, :1459
    }
    T.scatter_us = us_since(t0);
    return nmatches;
}

// src/ORBmatcher.cc:159-288
inline int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches)
{
    auto t0 = std::chrono::steady_clock::now();
    const std::vector<MapPoint *> vpMapPointsKF = pKF->GetMapPointMatches();   // :161
    vpMapPointMatches = std::vector<MapPoint *>(F.N, static_cast<MapPoint *>(NULL));   // :163
    const int nkf = (int)vpMapPointsKF.size();
    std::vector<uint8_t> has_mp(nkf + 1);
    std::vector<float> angle_kf(nkf + 1), angle_f(F.N + 1);
    for (int i = 0; i < nkf; ++i) {
        MapPoint *pMP = vpMapPointsKF[i];
        has_mp[i] = (pMP && !pMP->isBad()) ? 1 : 0;   // :194-198
        angle_kf[i] = pKF->mvKeysUn[i].angle;          // :245
    }
    for (int j = 0; j < F.N; ++j) angle_f[j] = F.mvKeys[j].angle;
    // DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) -> CSR, node ids ascending (the map's order)
    auto flatten = [](const DBoW2::FeatureVector &fv, std::vector<int32_t> &id, std::vector<int32_t> &off, std::vector<int32_t> &idx) {
        id.clear(); off.assign(1, 0); idx.clear();
        for (const auto &kv : fv) {
            id.push_back((int32_t)kv.first);
            for (unsigned int v : kv.second) idx.push_back((int32_t)v);
            off.push_back((int32_t)idx.size());
        }
        if (id.empty()) id.push_back(0);
        if (idx.empty()) idx.push_back(0);
    };
    std::vector<int32_t> idk, offk, idxk, idf, offf, idxf;
    flatten(pKF->mFeatVec, idk, offk, idxk);
    flatten(F.mFeatVec, idf, offf, idxf);
    aos2_bow_pair_t pair;
    pair.n_kf = nkf; pair.n_f = F.N;
    pair.desc_kf = pKF->mDescriptors.ptr<uint8_t>(); pair.desc_f = F.mDescriptors.ptr<uint8_t>();
    pair.kf_has_mp = has_mp.data(); pair.angle_kf = angle_kf.data(); pair.angle_f = angle_f.data();
    pair.n_nodes_kf = (int32_t)pKF->mFeatVec.size(); pair.n_nodes_f = (int32_t)F.mFeatVec.size();
    pair.node_id_kf = idk.data(); pair.node_off_kf = offk.data(); pair.node_idx_kf = idxk.data();
    pair.node_id_f = idf.data(); pair.node_off_f = offf.data(); pair.node_idx_f = idxf.data();
    std::vector<int32_t> match(F.N > 0 ? F.N : 1, -1);
    int32_t *mptr = match.data();
    int32_t nmatches = 0;
    aos2::ShimTiming &T = aos2::last_shim_timing();
    T.gather_us = us_since(t0);
    t0 = std::chrono::steady_clock::now();
    check(aos2_matcher_search_by_bow(aos2::matcher_handle(mfNNratio, mbCheckOrientation), &pair, 1, &mptr, &nmatches), "SearchByBoW");
    T.call_us = us_since(t0);
    t0 = std::chrono::steady_clock::now();
    for (int j = 0; j < F.N; ++j)
        if (match[j] >= 0) vpMapPointMatches[j] = vpMapPointsKF[match[j]];   // :239
    T.scatter_us = us_since(t0);
    return nmatches;
}

}  // namespace ORB_SLAM2
