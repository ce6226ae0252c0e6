// The part of ORB_SLAM2::Frame that sits on the hot path: Frame::ComputeStereoMatches()
// (include/Frame.h:81, src/Frame.cc:495-669).  In the reference it is a member that reads mvKeys,
// mvKeysRight, mDescriptors, mDescriptorsRight, mb, mbf and the two extractors' mvImagePyramid and
// fills mvuRight / mvDepth; here the same data are arguments and the work runs on the pyramids the
// two ORBextractor handles already hold in HBM (no pyramid copy).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "ORBextractor.h"

namespace ORB_SLAM2 {

inline void ComputeStereoMatches(ORBextractor *mpORBextractorLeft, ORBextractor *mpORBextractorRight,
                                 const std::vector<aos2::KeyPoint> &mvKeys, const std::vector<aos2::KeyPoint> &mvKeysRight,
                                 const aos2::Mat8 &mDescriptors, const aos2::Mat8 &mDescriptorsRight, float mb, float mbf,
                                 std::vector<float> &mvuRight, std::vector<float> &mvDepth)
{
    const int N = (int)mvKeys.size();
    mvuRight.assign(N, -1.0f);  // src/Frame.cc:497-498
    mvDepth.assign(N, -1.0f);
    if (N == 0) return;
    auto rows32 = [](const aos2::Mat8 &m, std::vector<uint8_t> &tmp) -> const uint8_t * {
        if (m.rows == 0 || m.step == 32) return m.data;
        tmp.resize((size_t)m.rows * 32);
        for (int r = 0; r < m.rows; ++r) std::copy(m.data + r * m.step, m.data + r * m.step + 32, tmp.data() + (size_t)r * 32);
        return tmp.data();
    };
    std::vector<uint8_t> tl, tr;
    const int st = aos2_compute_stereo_matches(
        mpORBextractorLeft->handle(), mpORBextractorRight->handle(), 0,
        reinterpret_cast<const aos2_keypoint_t *>(mvKeys.data()), rows32(mDescriptors, tl), N,
        reinterpret_cast<const aos2_keypoint_t *>(mvKeysRight.data()), rows32(mDescriptorsRight, tr), (int)mvKeysRight.size(),
        mb, mbf, mvuRight.data(), mvDepth.data());
    if (st != AOS2_OK) throw std::runtime_error(std::string("ComputeStereoMatches: ") + aos2_last_error());
}

}  // namespace ORB_SLAM2
