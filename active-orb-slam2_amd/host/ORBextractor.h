// ORB_SLAM2::ORBextractor with the reference's class surface (include/ORBextractor.h:45-111),
// forwarding to the C ABI (include/aos2.h) -> HIP kernels.  Frame.cc / Tracking.cc call it as they
// call the reference class: ctor (Tracking.cc:120-126), operator() (Frame.cc:276-282), getters
// (Frame.cc:37-43,94-100), public mvImagePyramid (Frame.cc:502,592,609).
#pragma once
#include <cassert>
#include <stdexcept>
#include <string>
#include <vector>

#include "aos2_types.h"

namespace ORB_SLAM2 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };  // include/ORBextractor.h:49 (unused by the reference too)

    // exposePyramid: copy all bordered levels back after every operator() (the reference's mvImagePyramid is always
    // there).  Off by default: its only reader, Frame::ComputeStereoMatches, runs on the device pyramids
    // (host/Frame.h), so the copy (8 levels, ~1.4 MB for 640x480) would be paid by every frame for nothing; a caller
    // that does read mvImagePyramid calls FillImagePyramid() after operator().
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0,
                 bool exposePyramid = false)
        : nlevels_(nlevels), exposePyramid_(exposePyramid)
    {
        if (aos2_extractor_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device, &h_) != AOS2_OK)
            throw std::invalid_argument(std::string("ORBextractor: ") + aos2_last_error());
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
    void operator()(const aos2::Mat8 &image, const aos2::Mat8 & /*mask*/, std::vector<aos2::KeyPoint> &keypoints,
                    aos2::Mat8 &descriptors)
    {
        if (image.empty()) return;  // src/ORBextractor.cc:1046
        const int cap = aos2_extractor_max_keypoints_for(h_, image.cols, image.rows);
        keypoints.resize(cap);
        scratch_.resize((size_t)cap * 32);
        int n = 0;
        const int st = aos2_extractor_extract(h_, image.data, image.cols, image.rows, (int)image.step,
                                              reinterpret_cast<aos2_keypoint_t *>(keypoints.data()), scratch_.data(), cap, &n);
        if (st != AOS2_OK) throw std::runtime_error(std::string("ORBextractor: ") + aos2_last_error());
        keypoints.resize(n);
        if (n == 0)
            descriptors.release();  // :1065
        else {
            descriptors.create(n, 32);  // :1068
            std::copy(scratch_.begin(), scratch_.begin() + (size_t)n * 32, descriptors.data);
        }
        if (exposePyramid_) FillImagePyramid();
    }

    // mvImagePyramid[level] of the last operator() call: ROI (interior) of a buffer that carries the 19-px
    // REFLECT_101 frame (src/ORBextractor.cc:1113-1128)
    void FillImagePyramid()
    {
        for (int l = 0; l < nlevels_; ++l) {
            int w = 0, h = 0;
            aos2_extractor_pyramid_level_size(h_, l, &w, &h);
            aos2::Mat8 full;
            full.create(h + 38, w + 38);
            if (aos2_extractor_pyramid_level(h_, 0, l, 19, full.data, (int)full.step) != AOS2_OK)
                throw std::runtime_error(std::string("ORBextractor: ") + aos2_last_error());
            mvImagePyramid[l] = full.roi(19, 19, w, h);
        }
    }

    int inline GetLevels() { return nlevels_; }
    float inline GetScaleFactor() { return aos2_extractor_scale_factor(h_); }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    std::vector<aos2::Mat8> mvImagePyramid;

    aos2_extractor_t *handle() { return h_; }

protected:
    aos2_extractor_t *h_ = nullptr;
    int nlevels_;
    bool exposePyramid_;
    std::vector<uint8_t> scratch_;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

}  // namespace ORB_SLAM2
