// ORB_SLAM2::ORBVocabulary with the part of DBoW2::TemplatedVocabulary's surface ORB-SLAM2 uses
// (include/ORBVocabulary.h:31-32; loaders called from src/System.cc:89-92, transform from
// Frame::ComputeBoW src/Frame.cc:424-431 / KeyFrame::ComputeBoW, score from KeyFrameDatabase.cc:133,249
// and LoopClosing.cc:136), forwarding to the C ABI (include/aos2.h) -> HIP kernels.
// DBoW2::BowVector / FeatureVector keep their std::map types (Thirdparty/DBoW2/DBoW2/BowVector.h, FeatureVector.h), filled
// from the key-ascending arrays.  Include AFTER the DBoW2 headers and OpenCV (or tests/cpp/refstub/slam_stub.h).
#pragma once
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "aos2_handles.h"

namespace ORB_SLAM2 {

class ORBVocabulary {
public:
    ORBVocabulary()   // (the GPU is the calling thread's: aos2::set_thread_device, default 0)
    {
        if (aos2_vocabulary_create(aos2::thread_device(), &h_) != AOS2_OK) aos2::fail("ORBVocabulary");
    }
    ~ORBVocabulary() { aos2_vocabulary_destroy(h_); }
    ORBVocabulary(const ORBVocabulary &) = delete;
    ORBVocabulary &operator=(const ORBVocabulary &) = delete;

    bool loadFromTextFile(const std::string &filename) { return aos2_vocabulary_load_text(h_, filename.c_str()) == AOS2_OK; }
    bool loadFromBinaryFile(const std::string &filename) { return aos2_vocabulary_load_binary(h_, filename.c_str()) == AOS2_OK; }
    void saveToBinaryFile(const std::string &filename) const
    {
        if (aos2_vocabulary_save_binary(h_, filename.c_str()) != AOS2_OK) aos2::fail("ORBVocabulary");
    }
    unsigned int size() const { return aos2_vocabulary_size(h_); }
    bool empty() const { return aos2_vocabulary_empty(h_) != 0; }
    int getBranchingFactor() const { return aos2_vocabulary_k(h_); }
    int getDepthLevels() const { return aos2_vocabulary_levels(h_); }

    // features: one 1x32 CV_8U row per descriptor (Converter::toDescriptorVector, src/Converter.cc:27-35)
    void transform(const std::vector<cv::Mat> &features, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup)
    {
        v.clear();
        fv.clear();
        const int n = (int)features.size();
        std::vector<uint8_t> d((size_t)n * 32);
        for (int i = 0; i < n; ++i) std::copy(features[i].data, features[i].data + 32, d.begin() + (size_t)i * 32);
        transform(d.data(), n, v, fv, levelsup);
    }
    // same on the descriptor matrix itself (n x 32 bytes, row-major), without the per-row Mat vector
    void transform(const uint8_t *desc, int n, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup)
    {
        v.clear();
        fv.clear();
        std::vector<uint32_t> bw((size_t)n + 1);
        std::vector<double> bv((size_t)n + 1);
        std::vector<int32_t> fn((size_t)n + 1), fo((size_t)n + 2), fi((size_t)n + 1);
        int nb = 0, nf = 0;
        if (aos2_vocabulary_transform(h_, desc, n, levelsup, bw.data(), bv.data(), &nb, fn.data(), fo.data(), fi.data(), &nf,
                                      nullptr, nullptr) != AOS2_OK)
            aos2::fail("ORBVocabulary::transform");
        for (int j = 0; j < nb; ++j) v.insert(v.end(), DBoW2::BowVector::value_type(bw[j], bv[j]));
        for (int s = 0; s < nf; ++s) {
            std::vector<unsigned int> idx(fi.begin() + fo[s], fi.begin() + fo[s + 1]);
            fv.insert(fv.end(), DBoW2::FeatureVector::value_type((DBoW2::NodeId)fn[s], std::move(idx)));
        }
    }

    double score(const DBoW2::BowVector &a, const DBoW2::BowVector &b) const
    {
        std::vector<uint32_t> w1, w2;
        std::vector<double> v1, v2;
        for (const auto &e : a) { w1.push_back(e.first); v1.push_back(e.second); }
        for (const auto &e : b) { w2.push_back(e.first); v2.push_back(e.second); }
        double s = 0;
        if (aos2_vocabulary_score(h_, w1.data(), v1.data(), (int)w1.size(), w2.data(), v2.data(), (int)w2.size(), &s) != AOS2_OK)
            aos2::fail("ORBVocabulary::score");
        return s;
    }

    aos2_vocabulary_t *handle() { return h_; }

private:
    aos2_vocabulary_t *h_ = nullptr;
};

}  // namespace ORB_SLAM2
