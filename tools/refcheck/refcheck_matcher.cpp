// Optional (needs OpenCV + Eigen3 + a checkout of the reference): ORB_SLAM2::ORBmatcher of the reference's own src/ORBmatcher.cc,
// driven through the reference's own Frame / KeyFrame / MapPoint / ORBVocabulary objects (compiled untouched by
// tools/refcheck/CMakeLists.txt), against the ORACLE's results stored in the cases of tools/refcheck/dump_cases.py:
//   dist.bundle     ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1632-1650) on 4096 descriptor pairs
//   bow_<i>.bundle  ORBmatcher(0.7, true).SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) (src/ORBmatcher.cc:159-288; the rotation
//                   histogram + ComputeThreeMaxima :1601-1630 run inside): the two frames are built from the case's images by the
//                   reference's RGB-D Frame constructor, the keyframe's map points are those the case flags
//   refcheck_matcher /tmp/refcases      exit code 0 = every count and every match index equal
// (If the reference's extractor does not reproduce the oracle's keypoints on a case -- refcheck_extractor shows why -- the BoW
// comparison of that case is reported as skipped, not as a matcher difference.)
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <opencv2/core/core.hpp>

#include "Frame.h"
#include "KeyFrame.h"
#include "KeyFrameDatabase.h"
#include "Map.h"
#include "MapPoint.h"
#include "ORBVocabulary.h"
#include "ORBextractor.h"
#include "ORBmatcher.h"

#include "../../tests/cpp/bundle_io.h"

using namespace ORB_SLAM2;

static int check_distance(const std::string &dir)
{
    const Bundle B = Bundle::load(dir + "/dist.bundle");
    const int n = (int)B["dist"].count();
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        cv::Mat a(1, 32, CV_8U, const_cast<uint8_t *>(B["a"].as<uint8_t>() + 32 * i)), b(1, 32, CV_8U, const_cast<uint8_t *>(B["b"].as<uint8_t>() + 32 * i));
        bad += ORBmatcher::DescriptorDistance(a, b) != B["dist"].as<int32_t>()[i];
    }
    printf("DescriptorDistance: %d of %d pairs differ\n", bad, n);
    return bad != 0;
}

static int check_bow(const std::string &dir, int idx, ORBVocabulary &voc)
{
    const std::string path = dir + "/bow_" + std::to_string(idx) + ".bundle";
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return -1;
    fclose(f);
    const Bundle B = Bundle::load(path);
    const int h = (int)B["img_kf"].dims[0], w = (int)B["img_kf"].dims[1];
    cv::Mat imKF(h, w, CV_8UC1, const_cast<uint8_t *>(B["img_kf"].as<uint8_t>())), imF(h, w, CV_8UC1, const_cast<uint8_t *>(B["img_f"].as<uint8_t>()));
    cv::Mat depth(h, w, CV_32F, const_cast<float *>(B["depth"].as<float>()));
    const float *cam = B["cam"].as<float>();
    cv::Mat K = cv::Mat::eye(3, 3, CV_32F);
    K.at<float>(0, 0) = cam[0]; K.at<float>(1, 1) = cam[1]; K.at<float>(0, 2) = cam[2]; K.at<float>(1, 2) = cam[3];
    cv::Mat distCoef = cv::Mat::zeros(4, 1, CV_32F);
    ORBextractor ex(1000, 1.2f, 8, 20, 7);
    Frame::mbInitialComputations = true;   // (the image bounds / grid constants are those of this case's camera)
    Frame fkf(imKF, depth, 0.0, &ex, &voc, K, distCoef, cam[4], 40.0f);
    Frame ff(imF, depth, 1.0, &ex, &voc, K, distCoef, cam[4], 40.0f);
    // the oracle's keypoints of both images (positions): the precondition of comparing match indices
    const int nkf = (int)B["out_kp_kf"].dims[0], nf = (int)B["out_kp_f"].dims[0];
    bool same_kp = fkf.N == nkf && ff.N == nf;
    for (int i = 0; same_kp && i < nkf; ++i) same_kp = fkf.mvKeys[i].pt.x == B["out_kp_kf"].as<float>()[2 * i] && fkf.mvKeys[i].pt.y == B["out_kp_kf"].as<float>()[2 * i + 1];
    for (int i = 0; same_kp && i < nf; ++i) same_kp = ff.mvKeys[i].pt.x == B["out_kp_f"].as<float>()[2 * i] && ff.mvKeys[i].pt.y == B["out_kp_f"].as<float>()[2 * i + 1];
    if (!same_kp) {
        printf("bow_%d: the reference's extractor gives %d / %d keypoints, the oracle %d / %d (or other positions): SearchByBoW not compared\n", idx, fkf.N, ff.N, nkf, nf);
        return 2;
    }
    Map map;
    KeyFrameDatabase db(voc);
    KeyFrame *kf = new KeyFrame(fkf, &map, &db);
    kf->ComputeBoW();
    ff.ComputeBoW();
    std::map<MapPoint *, int> index_of;
    const uint8_t *has = B["kf_has_mp"].as<uint8_t>();
    for (int i = 0; i < nkf; ++i)
        if (has[i]) {
            MapPoint *mp = new MapPoint(cv::Mat::zeros(3, 1, CV_32F), kf, &map);
            kf->AddMapPoint(mp, i);
            index_of[mp] = i;
        }
    ORBmatcher matcher(0.7f, true);
    std::vector<MapPoint *> vm;
    const int n = matcher.SearchByBoW(kf, ff, vm);
    const int32_t *want = B["out_match"].as<int32_t>();
    int bad = n != B["out_n"].as<int32_t>()[0];
    for (int j = 0; j < nf; ++j) {
        const int got = vm[j] ? index_of[vm[j]] : -1;
        bad += got != want[j];
    }
    printf("bow_%d: SearchByBoW %d matches (oracle %d), %d entries differ\n", idx, n, B["out_n"].as<int32_t>()[0], bad);
    return bad != 0;
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp/refcases";
    int bad = check_distance(dir);
    ORBVocabulary voc;
    if (!voc.loadFromTextFile(dir + "/voc.txt")) {
        fprintf(stderr, "cannot load %s/voc.txt\n", dir.c_str());
        return 2;
    }
    for (int i = 0;; ++i) {
        const int r = check_bow(dir, i, voc);
        if (r < 0) break;
        bad += r == 1;
    }
    return bad ? 1 : 0;
}
