// ORB_SLAM2::ORBextractor with the REFERENCE's class surface (include/ORBextractor.h:45-111), forwarding to the C ABI
// (include/aos2.h) -> HIP kernels.  Frame.cc / Tracking.cc call it as they call the reference class: ctor
// (Tracking.cc:120-126), operator()(cv::InputArray, cv::InputArray, std::vector<cv::KeyPoint>&, cv::OutputArray)
// (Frame.cc:276-282), getters (Frame.cc:37-43,94-100), public mvImagePyramid (Frame.cc:502,592,609).
// Include AFTER <opencv2/core/core.hpp> (or tests/cpp/refstub/opencv_stub.h in this image, which has no OpenCV).
#pragma once
#include <cassert>
#include <cstring>
#include <string>
#include <vector>

#include "aos2_handles.h"

namespace ORB_SLAM2 {

static_assert(sizeof(cv::KeyPoint) == sizeof(aos2_keypoint_t) && sizeof(cv::KeyPoint) == 28, "cv::KeyPoint layout");

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };  // include/ORBextractor.h:49 (unused by the reference too)

    // The reference's five arguments; the GPU is the calling thread's (aos2::set_thread_device, default 0).
    // mvImagePyramid is filled by every operator() call, like the reference's (src/ORBextractor.cc:1107-1132): an unchanged
    // Frame::ComputeStereoMatches (src/Frame.cc:502, 592, 609) reads it.  The copy is 8 levels, ~1.4 MB for 640x480, per frame;
    // a tree whose only reader is the ComputeStereoMatches of FrameMembers.h (which runs on the pyramids the two extractors hold on
    // the device) switches it off with SetExposePyramid(false) -- FrameMembers.h does so itself on its first call -- and a caller
    // that wants one frame's levels afterwards calls FillImagePyramid().
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST) : nlevels_(nlevels)
    {
        if (aos2_extractor_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, aos2::thread_device(), &h_) != AOS2_OK)
            aos2::fail("ORBextractor");
        const float *a = aos2_extractor_scale_factors(h_), *b = aos2_extractor_inv_scale_factors(h_);
        const float *c = aos2_extractor_sigma2(h_), *d = aos2_extractor_inv_sigma2(h_);
        mvScaleFactor.assign(a, a + nlevels);
        mvInvScaleFactor.assign(b, b + nlevels);
        mvLevelSigma2.assign(c, c + nlevels);
        mvInvLevelSigma2.assign(d, d + nlevels);
        mvImagePyramid.resize(nlevels);
    }
    ~ORBextractor() { aos2_extractor_destroy(h_); }
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // Compute the ORB features and descriptors on an image.  Mask is ignored (as in the reference).
    void operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint> &_keypoints, cv::OutputArray _descriptors)
    {
        if (_image.empty()) return;  // src/ORBextractor.cc:1046
        const cv::Mat image = _image.getMat();
        assert(image.type() == CV_8UC1);  // :1050
        const int cap = aos2_extractor_max_keypoints_for(h_, image.cols, image.rows);
        _keypoints.resize(cap);
        scratch_.resize((size_t)cap * 32);
        int n = 0;
        const int st = aos2_extractor_extract(h_, image.data, image.cols, image.rows, (int)image.step,
                                              reinterpret_cast<aos2_keypoint_t *>(_keypoints.data()), scratch_.data(), cap, &n);
        if (st != AOS2_OK) aos2::fail("ORBextractor");
        _keypoints.resize(n);
        if (n == 0)
            _descriptors.release();  // :1065
        else {
            _descriptors.create(n, 32, CV_8U);  // :1068
            cv::Mat descriptors = _descriptors.getMat();
            for (int i = 0; i < n; ++i) memcpy(descriptors.ptr<uint8_t>(i), &scratch_[(size_t)i * 32], 32);
        }
        if (exposePyramid_) FillImagePyramid();
    }

    void SetExposePyramid(bool on) { exposePyramid_ = on; }

    // mvImagePyramid[level] of the last operator() call: ROI (interior) of a buffer that carries the 19-px
    // REFLECT_101 frame (src/ORBextractor.cc:1113-1128)
    void FillImagePyramid()
    {
        for (int l = 0; l < nlevels_; ++l) {
            int w = 0, h = 0;
            aos2_extractor_pyramid_level_size(h_, l, &w, &h);
            cv::Mat full(h + 38, w + 38, CV_8UC1);
            if (aos2_extractor_pyramid_level(h_, 0, l, 19, full.data, (int)full.step) != AOS2_OK)
                aos2::fail("ORBextractor");
            mvImagePyramid[l] = full(cv::Rect(19, 19, w, h));
        }
    }

    int inline GetLevels() { return nlevels_; }
    float inline GetScaleFactor() { return aos2_extractor_scale_factor(h_); }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    std::vector<cv::Mat> mvImagePyramid;

    aos2_extractor_t *handle() { return h_; }

protected:
    aos2_extractor_t *h_ = nullptr;
    int nlevels_;
    bool exposePyramid_ = true;
    std::vector<uint8_t> scratch_;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

}  // namespace ORB_SLAM2
