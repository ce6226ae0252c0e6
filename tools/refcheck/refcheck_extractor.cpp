// Optional (needs OpenCV + a checkout of the reference): ORB_SLAM2::ORBextractor::operator() of the reference's own
// src/ORBextractor.cc (compiled untouched by tools/refcheck/CMakeLists.txt) against the oracle on seeded frames.  Reports
// keypoint / descriptor differences; the known convention differences (octree tie-break by creation
// order instead of heap address, correctly rounded sin / cos instead of libm's) show up here and nowhere else.
#include <cstdio>
#include <cstring>
#include <vector>

#include <opencv2/core/core.hpp>

#include "ORBextractor.h"   // the reference's header

extern "C" {
#include "orb_oracle.h"
}

int main()
{
    int bad = 0;
    for (unsigned seed = 0; seed < 4; ++seed) {
        const int w = 640, h = 480;
        cv::Mat img(h, w, CV_8UC1);
        unsigned long long s = 0xD1B54A32D192ED03ull ^ seed;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                const int v = ((x / 23) * 61 + (y / 19) * 97 + (int)seed * 17) % 256 + (int)(s % 9) - 4;
                img.at<uint8_t>(y, x) = (uint8_t)std::min(255, std::max(0, v));
            }
        ORB_SLAM2::ORBextractor ref(1000, 1.2f, 8, 20, 7);
        std::vector<cv::KeyPoint> kps;
        cv::Mat desc;
        ref(img, cv::Mat(), kps, desc);
        // the oracle under each of its extractor conventions (orc_set_tiebreak_mode / orc_set_trig_mode): which one,
        // if any, reproduces THIS build of the reference?  Expected (profiles/r05_convention_effects.txt): the octree tie-break moves
        // ~2 % of the keypoints (~20 per frame) whichever way the reference's heap orders equal-size nodes; libm sin / cos none.
        // Keypoints are compared as SETS as well: an order difference alone is the list order of equal nodes, not a different selection.
        int best_dk = -1;
        for (int mode = 0; mode < 4; ++mode) {
            orc_set_tiebreak_mode(mode & 1);
            orc_set_trig_mode(mode >> 1);
            orc_extractor_t *o = orc_extractor_create(1000, 1.2f, 8, 20, 7);
            std::vector<orc_keypoint_t> ok(4096);
            std::vector<uint8_t> od(4096 * 32);
            int n = 0;
            orc_extractor_extract(o, img.data, w, h, (int)img.step, ok.data(), od.data(), 4096, &n);
            int dk = n != (int)kps.size(), dbits = 0, not_in_ref = 0;
            for (int i = 0; i < n && i < (int)kps.size(); ++i) {
                dk += kps[i].pt.x != ok[i].x || kps[i].pt.y != ok[i].y || kps[i].octave != ok[i].octave || kps[i].angle != ok[i].angle;
                for (int b = 0; b < 32; ++b) dbits += __builtin_popcount(desc.at<uint8_t>(i, b) ^ od[(size_t)i * 32 + b]);
            }
            for (int i = 0; i < n; ++i) {
                bool found = false;
                for (size_t j = 0; j < kps.size() && !found; ++j)
                    found = kps[j].pt.x == ok[i].x && kps[j].pt.y == ok[i].y && kps[j].octave == ok[i].octave;
                not_in_ref += !found;
            }
            printf("seed %u, oracle with tie-break %s, sin / cos %s: reference %zu keypoints, oracle %d; %d positions differ in list order, %d oracle "
                   "keypoints are not in the reference's set, %d descriptor bits differ (same list position)\n", seed,
                   (mode & 1) ? "REVERSED" : "as tested", (mode >> 1) ? "libm" : "correctly rounded", kps.size(), n, dk, not_in_ref, dbits);
            if (best_dk < 0 || dk + dbits < best_dk) best_dk = dk + dbits;
            orc_extractor_destroy(o);
        }
        orc_set_tiebreak_mode(0);
        orc_set_trig_mode(0);
        bad += best_dk != 0;
    }
    if (bad) printf("%d of 4 frames match under NO combination of the conventions: a real disagreement (or a third heap order)\n", bad);
    return bad ? 1 : 0;
}
