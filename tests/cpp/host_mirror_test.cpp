// Exercises the C++ mirror classes (host/*.h) exactly like Frame.cc / Tracking.cc use the reference
// classes.  usage: host_mirror_test <raw_u8_image> <w> <h> <nfeatures> [out.bin]
// Without a GPU only the constructor/getter part runs (exit code 3 = no device).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "../../active-orb-slam2_amd/host/Frame.h"
#include "../../active-orb-slam2_amd/host/ORBextractor.h"
#include "../../active-orb-slam2_amd/host/ORBmatcher.h"
#include "../../active-orb-slam2_amd/host/ORBVocabulary.h"
#include "../../active-orb-slam2_amd/host/Optimizer.h"

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    const int w = atoi(argv[2]), h = atoi(argv[3]), nf = atoi(argv[4]);
    ORB_SLAM2::ORBextractor ex(nf, 1.2f, 8, 20, 7);
    std::vector<float> sf = ex.GetScaleFactors(), isig = ex.GetInverseScaleSigmaSquares();
    printf("levels %d scale %.3f sf7 %.6f isig7 %.6f\n", ex.GetLevels(), ex.GetScaleFactor(), sf[7], isig[7]);
    if (ORB_SLAM2::ORBmatcher::TH_LOW != 50 || ORB_SLAM2::ORBmatcher::TH_HIGH != 100 || ORB_SLAM2::ORBmatcher::HISTO_LENGTH != 30) return 4;
    {   // vocabulary host side: 2 words under the root, save / load (eof quirk: one more node and word)
        ORB_SLAM2::ORBVocabulary voc;
        if (!voc.empty()) return 10;
        const int32_t parent[2] = {0, 0};
        uint8_t vd[64] = {};
        for (int i = 32; i < 64; ++i) vd[i] = 255;
        const double vw[2] = {1.0, 3.0};
        const uint8_t leaf[2] = {1, 1};
        if (aos2_vocabulary_set_nodes(voc.handle(), 2, 1, 0, 0, 2, parent, vd, vw, leaf) != AOS2_OK || voc.size() != 2) return 11;
        if (argc > 5) {
            const std::string vp = std::string(argv[5]) + ".voc";
            voc.saveToBinaryFile(vp);
            ORB_SLAM2::ORBVocabulary v2;
            if (!v2.loadFromBinaryFile(vp) || v2.size() != 3 || v2.getBranchingFactor() != 2 || v2.getDepthLevels() != 1) return 12;
            if (aos2_device_count() >= 1) {
                std::vector<uint8_t> q(3 * 32, 0);
                for (int i = 64; i < 96; ++i) q[i] = 255;  // features: word 0, word 0, word 1
                DBoW2::BowVector bv;
                DBoW2::FeatureVector fv;
                v2.transform(q.data(), 3, bv, fv, 1);
                if (bv.size() != 2 || bv[0] != 2.0 / 5.0 || bv[1] != 3.0 / 5.0 || fv.size() != 1 || fv[0].size() != 3) return 13;
                if (v2.score(bv, bv) != 1.0) return 14;
            }
        }
    }
    if (aos2_device_count() < 1) {
        printf("no device\n");
        return 3;
    }
    std::vector<uint8_t> buf((size_t)w * h);
    std::ifstream f(argv[1], std::ios::binary);
    f.read(reinterpret_cast<char *>(buf.data()), buf.size());
    aos2::Mat8 image(h, w, buf.data(), (size_t)w), mask, desc;
    std::vector<aos2::KeyPoint> kps;
    ex(image, mask, kps, desc);
    ex.FillImagePyramid();   // (lazy: only a caller that reads mvImagePyramid pays for the copy)
    printf("n %zu desc %dx%d pyr0 %dx%d pyr7 %dx%d\n", kps.size(), desc.rows, desc.cols, ex.mvImagePyramid[0].cols,
           ex.mvImagePyramid[0].rows, ex.mvImagePyramid[7].cols, ex.mvImagePyramid[7].rows);
    // the ROI exposes the border like the reference: pixel (-1,-1) of level 0 is REFLECT_101 = (1,1)
    const aos2::Mat8 &p0 = ex.mvImagePyramid[0];
    if (*(p0.data - p0.step - 1) != p0.data[p0.step + 1]) return 5;
    if (p0.data[5 * p0.step + 7] != buf[5 * (size_t)w + 7]) return 6;
    if (!kps.empty() && ORB_SLAM2::ORBmatcher::DescriptorDistance(desc.roi(0, 0, 32, 1), desc.roi(0, 0, 32, 1)) != 0) return 7;
    {   // stereo Frame with two identical eyes: matches exist but the median cull (thDist = 0) drops them all
        ORB_SLAM2::ORBextractor exR(nf, 1.2f, 8, 20, 7, 0, false);
        std::vector<aos2::KeyPoint> kr;
        aos2::Mat8 dr;
        exR(image, mask, kr, dr);
        std::vector<float> uR, depth;
        ORB_SLAM2::ComputeStereoMatches(&ex, &exR, kps, kr, desc, dr, 0.0773f, 40.0f, uR, depth);
        if (uR.size() != kps.size() || depth.size() != kps.size()) return 8;
        for (float v : uR)
            if (v != -1.0f) return 9;
    }
    if (argc > 5) {
        std::ofstream o(argv[5], std::ios::binary);
        o.write(reinterpret_cast<const char *>(kps.data()), kps.size() * sizeof(aos2::KeyPoint));
        o.write(reinterpret_cast<const char *>(desc.data), (size_t)desc.rows * 32);
    }
    return 0;
}
