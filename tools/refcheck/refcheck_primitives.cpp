// Optional (needs OpenCV): the oracle's restatements of the OpenCV calls the extractor makes against
// the real ones on seeded images.  Prints the number of differing pixels / keypoints per primitive; exit code 0 = all equal.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <opencv2/imgproc/imgproc.hpp>

extern "C" {
#include "orb_oracle.h"
}

static cv::Mat seeded(int w, int h, unsigned seed)
{
    cv::Mat m(h, w, CV_8UC1);
    unsigned long long s = 0x9E3779B97F4A7C15ull ^ seed;
    auto next = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    // blocks + noise: corners at both thresholds, flat zones, saturation
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int bx = x / 17, by = y / 13;
            int v = (int)((bx * 73 + by * 151 + seed * 31) % 256);
            v += (int)(next() % 13) - 6;
            m.at<uint8_t>(y, x) = (uint8_t)std::min(255, std::max(0, v));
        }
    return m;
}

int main()
{
    int bad = 0;
    for (unsigned seed = 0; seed < 8; ++seed) {
        const int w = 96 + 37 * (int)seed, h = 80 + 29 * (int)seed;
        cv::Mat img = seeded(w, h, seed);
        // cv::FAST 9/16 with NMS at both thresholds
        for (int th : {20, 7}) {
            std::vector<cv::KeyPoint> kps;
            cv::FAST(img, kps, th, true);
            std::vector<int16_t> xs(w * h), ys(w * h);
            std::vector<uint8_t> sc(w * h);
            const int n = orc_fast9_16(img.data, w, h, (int)img.step, th, xs.data(), ys.data(), sc.data(), w * h);
            int diff = n != (int)kps.size();
            for (int i = 0; i < n && !diff; ++i)
                diff |= (int)kps[i].pt.x != xs[i] || (int)kps[i].pt.y != ys[i] || (int)kps[i].response != sc[i];
            printf("seed %u FAST th %2d: cv %zu oracle %d %s\n", seed, th, kps.size(), n, diff ? "DIFFER" : "equal");
            bad += diff;
        }
        // cv::resize INTER_LINEAR
        {
            const int dw = (int)std::lround(w / 1.2f), dh = (int)std::lround(h / 1.2f);
            cv::Mat ref, got(dh, dw, CV_8UC1);
            cv::resize(img, ref, cv::Size(dw, dh), 0, 0, cv::INTER_LINEAR);
            orc_resize_linear_u8(img.data, w, h, (int)img.step, got.data, dw, dh, (int)got.step);
            const int d = cv::countNonZero(ref != got);
            printf("seed %u resize: %d differing pixels\n", seed, d);
            bad += d != 0;
        }
        // cv::GaussianBlur 7x7 sigma 2 REFLECT_101
        {
            cv::Mat ref, got(h, w, CV_8UC1);
            cv::GaussianBlur(img, ref, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
            orc_gaussian_blur7_u8(img.data, w, h, (int)img.step, got.data, (int)got.step);
            const int d = cv::countNonZero(ref != got);
            printf("seed %u GaussianBlur: %d differing pixels\n", seed, d);
            bad += d != 0;
        }
        // cv::copyMakeBorder REFLECT_101
        {
            cv::Mat ref, got(h + 38, w + 38, CV_8UC1);
            cv::copyMakeBorder(img, ref, 19, 19, 19, 19, cv::BORDER_REFLECT_101);
            orc_copy_make_border_reflect101(img.data, w, h, (int)img.step, got.data, 19, (int)got.step);
            const int d = cv::countNonZero(ref != got);
            printf("seed %u copyMakeBorder: %d differing pixels\n", seed, d);
            bad += d != 0;
        }
    }
    // cv::fastAtan2 and cvRound on a sweep
    int d = 0;
    for (int i = -2000; i <= 2000; ++i)
        for (int j = -2000; j <= 2000; j += 37) {
            const float y = i * 0.37f, x = j * 0.91f;
            d += cv::fastAtan2(y, x) != orc_fast_atan2(y, x);
        }
    for (int i = -100000; i <= 100000; ++i) d += cvRound(i * 0.005f) != orc_cv_round_f(i * 0.005f);
    printf("fastAtan2 / cvRound: %d differing values\n", d);
    bad += d != 0;
    printf(bad ? "REFCHECK: %d primitive checks differ\n" : "REFCHECK: all primitives equal (%d)\n", bad);
    return bad ? 1 : 0;
}
