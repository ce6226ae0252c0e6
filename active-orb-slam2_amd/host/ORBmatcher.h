// ORB_SLAM2::ORBmatcher hot-path surface (include/ORBmatcher.h:37-102) over SoA snapshots.
// Frame / KeyFrame / MapPoint are out of scope (host pointer graph); the real ORBmatcher.cc methods
// keep their signatures and bodies of ~20 lines each that snapshot the fields listed in SURVEY.md
// App. E into these views, call the method below, and map the returned indices back to MapPoint*
// (INTEGRATION.md shows them).
#pragma once
#include <stdexcept>
#include <string>
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

    aos2_matcher_t *handle() { return h_; }

private:
    static void check(int st)
    {
        if (st != AOS2_OK) throw std::runtime_error(std::string("ORBmatcher: ") + aos2_last_error());
    }
    aos2_matcher_t *h_ = nullptr;
};

}  // namespace ORB_SLAM2
