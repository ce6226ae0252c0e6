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
#include "../../active-orb-slam2_amd/host/Optimizer.h"

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    const int w = atoi(argv[2]), h = atoi(argv[3]), nf = atoi(argv[4]);
    ORB_SLAM2::ORBextractor ex(nf, 1.2f, 8, 20, 7);
    std::vector<float> sf = ex.GetScaleFactors(), isig = ex.GetInverseScaleSigmaSquares();
    printf("levels %d scale %.3f sf7 %.6f isig7 %.6f\n", ex.GetLevels(), ex.GetScaleFactor(), sf[7], isig[7]);
    if (ORB_SLAM2::ORBmatcher::TH_LOW != 50 || ORB_SLAM2::ORBmatcher::TH_HIGH != 100 || ORB_SLAM2::ORBmatcher::HISTO_LENGTH != 30) return 4;
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
