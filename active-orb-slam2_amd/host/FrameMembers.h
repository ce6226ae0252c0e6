// The two members of ORB_SLAM2::Frame that sit on the hot path, at their reference signatures (include/Frame.h:64, :93):
//     void Frame::ComputeStereoMatches();     src/Frame.cc:495-669   (stereo Frame constructor, :109)
//     void Frame::ComputeBoW();               src/Frame.cc:424-431   (Tracking.cc:858, 1540; KeyFrame::ComputeBoW is the same two lines)
// These are the bodies a maintainer puts into src/Frame.cc in place of the reference's; the class declaration
// (include/Frame.h) does not change.  Include AFTER Frame.h / ORBextractor.h / ORBVocabulary.h of this directory (or the
// stand-ins of tests/cpp/refstub).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "ORBVocabulary.h"
#include "ORBextractor.h"

namespace ORB_SLAM2 {

// Reads mvKeys, mvKeysRight, mDescriptors, mDescriptorsRight, mb, mbf and the two extractors' pyramids, fills mvuRight /
// mvDepth.  The pyramids never leave the device: the row-band Hamming search and the 11x11 SAD refinement run on the
// levels the two ORBextractor handles hold from the operator() calls of this frame (no mvImagePyramid copy).
inline void Frame::ComputeStereoMatches()
{
    mvuRight = std::vector<float>(N, -1.0f);  // src/Frame.cc:497-498
    mvDepth = std::vector<float>(N, -1.0f);
    // (this body is the reader of mvImagePyramid that the reference's was: from here on the extractors keep the levels on the device only)
    mpORBextractorLeft->SetExposePyramid(false);
    mpORBextractorRight->SetExposePyramid(false);
    if (N == 0) return;
    auto rows32 = [](const cv::Mat &m, int n, std::vector<uint8_t> &tmp) -> const uint8_t * {
        if (n == 0 || m.step == 32) return m.ptr<uint8_t>();
        tmp.resize((size_t)n * 32);
        for (int r = 0; r < n; ++r) memcpy(tmp.data() + (size_t)r * 32, m.ptr<uint8_t>(r), 32);
        return tmp.data();
    };
    std::vector<uint8_t> tl, tr;
    const int Nr = (int)mvKeysRight.size();
    const int st = aos2_compute_stereo_matches(
        mpORBextractorLeft->handle(), mpORBextractorRight->handle(), 0, reinterpret_cast<const aos2_keypoint_t *>(mvKeys.data()),
        rows32(mDescriptors, N, tl), N, reinterpret_cast<const aos2_keypoint_t *>(mvKeysRight.data()), rows32(mDescriptorsRight, Nr, tr), Nr,
        mb, mbf, mvuRight.data(), mvDepth.data());
    if (st != AOS2_OK) aos2::fail("ComputeStereoMatches");
}

// mBowVec / mFeatVec from the descriptor matrix (levelsup = 4).  The reference first splits mDescriptors into one cv::Mat
// per row (Converter::toDescriptorVector); the rows go to the device as the matrix they already are.
inline void Frame::ComputeBoW()
{
    if (!mBowVec.empty()) return;
    if (mDescriptors.isContinuous() || N == 0)
        mpORBvocabulary->transform(mDescriptors.ptr<uint8_t>(), N, mBowVec, mFeatVec, 4);
    else {
        std::vector<cv::Mat> vCurrentDesc;
        vCurrentDesc.reserve(N);
        for (int j = 0; j < N; ++j) vCurrentDesc.push_back(mDescriptors.row(j));
        mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4);
    }
}

}  // namespace ORB_SLAM2
