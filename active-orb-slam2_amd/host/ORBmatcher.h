// ORB_SLAM2::ORBmatcher hot-path surface (include/ORBmatcher.h:37-102) over SoA snapshots.
// Frame / KeyFrame / MapPoint are out of scope (host pointer graph); the real ORBmatcher.cc methods
// keep their signatures and bodies of ~20 lines each that snapshot the fields listed in SURVEY.md
// App. E into these views, call the method below, and map the returned indices back to MapPoint*
// (INTEGRATION.md shows them).
#pragma once
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "aos2_types.h"

namespace ORB_SLAM2 {

class ORBmatcher {
public:
    static const int TH_LOW = AOS2_TH_LOW;            // src/ORBmatcher.cc:38
    static const int TH_HIGH = AOS2_TH_HIGH;          // :37
    static const int HISTO_LENGTH = AOS2_HISTO_LENGTH;  // :39

    ORBmatcher(float nnratio = 0.6, bool checkOri = true, int device = 0)
    {
        if (aos2_matcher_create(nnratio, checkOri ? 1 : 0, device, &h_) != AOS2_OK)
            throw std::invalid_argument(std::string("ORBmatcher: ") + aos2_last_error());
    }
    ~ORBmatcher() { aos2_matcher_destroy(h_); }
    ORBmatcher(const ORBmatcher &) = delete;
    ORBmatcher &operator=(const ORBmatcher &) = delete;

    // Computes the Hamming distance between two ORB descriptors (32-byte rows)
    static int DescriptorDistance(const aos2::Mat8 &a, const aos2::Mat8 &b) { return aos2_descriptor_distance(a.data, b.data); }

    // SearchByProjection(Frame &F, const std::vector<MapPoint*> &vpMapPoints, const float th=3)
    // match[j] = index into vpMapPoints newly assigned to F.mvpMapPoints[j], or -1
    int SearchByProjection(const aos2_frame_view_t &F, const aos2_proj_mp_t &vpMapPoints, std::vector<int32_t> &match,
                           const float th = 3)
    {
        match.assign(F.n_f > 0 ? F.n_f : 1, -1);
        int32_t n = 0;
        check(aos2_matcher_search_by_projection(h_, &F, &vpMapPoints, th, match.data(), &n));
        match.resize(F.n_f);
        return n;
    }

    // SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
    int SearchByProjection(const aos2_frame_view_t &CurrentFrame, const aos2_proj_last_t &LastFrame,
                           std::vector<int32_t> &match, const float th, const bool bMono)
    {
        match.assign(CurrentFrame.n_f > 0 ? CurrentFrame.n_f : 1, -1);
        int32_t n = 0;
        check(aos2_matcher_search_by_projection_last(h_, &CurrentFrame, &LastFrame, th, bMono ? 1 : 0, match.data(), &n));
        match.resize(CurrentFrame.n_f);
        return n;
    }

    // SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint*> &vpMapPointMatches)
    // vpMapPointMatches[j] = index of the KF feature whose MapPoint is matched to F feature j, or -1
    int SearchByBoW(const aos2_bow_pair_t &pair, std::vector<int32_t> &vpMapPointMatches)
    {
        vpMapPointMatches.assign(pair.n_f > 0 ? pair.n_f : 1, -1);
        int32_t n = 0;
        int32_t *out = vpMapPointMatches.data();
        check(aos2_matcher_search_by_bow(h_, &pair, 1, &out, &n));
        vpMapPointMatches.resize(pair.n_f);
        return n;
    }

    // SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint*> &vpMatches12)  (:522-655)
    // vpMatches12[i] = index of the KF2 feature whose MapPoint is matched to KF1 feature i, or -1
    int SearchByBoW(const aos2_bow_kf_pair_t &pair, std::vector<int32_t> &vpMatches12)
    {
        vpMatches12.assign(pair.n1 > 0 ? pair.n1 : 1, -1);
        int32_t n = 0;
        int32_t *out = vpMatches12.data();
        check(aos2_matcher_search_by_bow_kf(h_, &pair, 1, &out, &n));
        vpMatches12.resize(pair.n1);
        return n;
    }

    // SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo)  (:657-823); F12 and the epipole
    // travel inside `pair`
    int SearchForTriangulation(const aos2_triang_pair_t &pair, std::vector<std::pair<size_t, size_t>> &vMatchedPairs,
                               const bool bOnlyStereo)
    {
        std::vector<int32_t> m12(pair.n1 > 0 ? pair.n1 : 1, -1);
        int32_t n = 0;
        int32_t *out = m12.data();
        check(aos2_matcher_search_for_triangulation(h_, &pair, 1, bOnlyStereo ? 1 : 0, &out, &n));
        vMatchedPairs.clear();
        vMatchedPairs.reserve(n > 0 ? n : 0);
        for (int i = 0; i < pair.n1; ++i)
            if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));  // :811-820
        return n;
    }

    // search part of Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, th) (:825-975, sim3 = false) and of
    // Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (:977-1100, sim3 = true): bestIdx[i] = feature to fuse with or -1
    int Fuse(const aos2_frame_view_t &pKF, const aos2_proj_points_t &vpMapPoints, std::vector<int32_t> &bestIdx,
             std::vector<int32_t> &bestDist, bool sim3 = false)
    {
        bestIdx.assign(vpMapPoints.n_pts > 0 ? vpMapPoints.n_pts : 1, -1);
        bestDist.assign(bestIdx.size(), 256);
        int32_t n = 0;
        check(aos2_matcher_fuse(h_, &pKF, &vpMapPoints, sim3 ? 1 : 0, bestIdx.data(), bestDist.data(), &n));
        bestIdx.resize(vpMapPoints.n_pts);
        bestDist.resize(vpMapPoints.n_pts);
        return n;
    }

    // SearchByProjection(KeyFrame *pKF, cv::Mat Scw, vpPoints, vpMatched, th)  (:290-403)
    int SearchByProjection(const aos2_frame_view_t &pKF, const aos2_proj_points_t &vpPoints, std::vector<int32_t> &vpMatched)
    {
        vpMatched.assign(pKF.n_f > 0 ? pKF.n_f : 1, -1);
        int32_t n = 0;
        check(aos2_matcher_search_by_projection_kf(h_, &pKF, &vpPoints, vpMatched.data(), &n));
        vpMatched.resize(pKF.n_f);
        return n;
    }

    // SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist)  (:1472-1599)
    int SearchByProjection(const aos2_frame_view_t &CurrentFrame, const aos2_proj_points_t &pKFpoints,
                           std::vector<int32_t> &match, const int ORBdist)
    {
        match.assign(CurrentFrame.n_f > 0 ? CurrentFrame.n_f : 1, -1);
        int32_t n = 0;
        check(aos2_matcher_search_by_projection_reloc(h_, &CurrentFrame, &pKFpoints, ORBdist, match.data(), &n));
        match.resize(CurrentFrame.n_f);
        return n;
    }

    // SearchForInitialization(Frame &F1, Frame &F2, vbPrevMatched, vnMatches12, windowSize=10)  (:405-520)
    // F1 enters through its descriptor rows, octaves and angles; vbPrevMatched (2 floats per F1 feature) is updated
    // like :512-515
    int SearchForInitialization(int n1, const uint8_t *desc1, const int32_t *octave1, const float *angle1,
                                const aos2_frame_view_t &F2, std::vector<float> &vbPrevMatched, std::vector<int> &vnMatches12,
                                int windowSize = 10)
    {
        vnMatches12.assign(n1 > 0 ? n1 : 1, -1);
        int32_t n = 0;
        check(aos2_matcher_search_for_initialization(h_, &F2, n1, desc1, octave1, angle1, vbPrevMatched.data(), windowSize,
                                                     vnMatches12.data(), &n));
        vnMatches12.resize(n1);
        for (int i1 = 0; i1 < n1; ++i1)
            if (vnMatches12[i1] >= 0) {
                vbPrevMatched[2 * i1] = F2.kp_x[vnMatches12[i1]];
                vbPrevMatched[2 * i1 + 1] = F2.kp_y[vnMatches12[i1]];
            }
        return n;
    }

    // SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)  (:1102-1326)
    int SearchBySim3(const aos2_frame_view_t &pKF1, const aos2_frame_view_t &pKF2, const aos2_proj_points_t &p12,
                     const aos2_proj_points_t &p21, std::vector<int32_t> &vpMatches12)
    {
        vpMatches12.assign(p12.n_pts > 0 ? p12.n_pts : 1, -1);
        int32_t n = 0;
        check(aos2_matcher_search_by_sim3(h_, &pKF1, &pKF2, &p12, &p21, vpMatches12.data(), &n));
        vpMatches12.resize(p12.n_pts);
        return n;
    }

    // MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:275-340) for a batch of map points (CSR)
    void ComputeDistinctiveDescriptors(const std::vector<int32_t> &off, const uint8_t *desc, std::vector<int32_t> &best)
    {
        const int n = (int)off.size() - 1;
        best.assign(n > 0 ? n : 1, -1);
        check(aos2_compute_distinctive_descriptors(h_, n > 0 ? n : 0, off.data(), desc, best.data()));
        best.resize(n > 0 ? n : 0);
    }

    aos2_matcher_t *handle() { return h_; }

private:
    static void check(int st)
    {
        if (st != AOS2_OK) throw std::runtime_error(std::string("ORBmatcher: ") + aos2_last_error());
    }
    aos2_matcher_t *h_ = nullptr;
};

}  // namespace ORB_SLAM2
