// Minimal POD stand-ins for the two OpenCV types that appear in the hot-path class surfaces.
// OpenCV is not available in this image; a build against the real reference replaces them with
// cv::Mat / cv::KeyPoint (same memory layout: KeyPoint is the 28-byte cv::KeyPoint, Mat8 is a
// CV_8UC1 view), see INTEGRATION.md.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#include "../../include/aos2.h"

namespace aos2 {

struct Point2f {
    float x, y;
};

// bit-compatible with cv::KeyPoint and aos2_keypoint_t
struct KeyPoint {
    Point2f pt;
    float size;
    float angle;
    float response;
    int octave;
    int class_id;
};
static_assert(sizeof(KeyPoint) == sizeof(aos2_keypoint_t) && sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

// 8-bit single channel matrix (cv::Mat of type CV_8UC1): owning or a view
struct Mat8 {
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t *data = nullptr;
    std::shared_ptr<std::vector<uint8_t>> owner;

    Mat8() = default;
    Mat8(int r, int c, uint8_t *d, size_t s) : rows(r), cols(c), step(s), data(d) {}
    void create(int r, int c)
    {
        owner = std::make_shared<std::vector<uint8_t>>((size_t)r * c);
        rows = r;
        cols = c;
        step = (size_t)c;
        data = owner->data();
    }
    void release()
    {
        owner.reset();
        rows = cols = 0;
        step = 0;
        data = nullptr;
    }
    bool empty() const { return data == nullptr || rows <= 0 || cols <= 0; }
    uint8_t *ptr(int r) { return data + (size_t)r * step; }
    const uint8_t *ptr(int r) const { return data + (size_t)r * step; }
    // ROI view (cv::Mat::operator()(Rect))
    Mat8 roi(int x, int y, int w, int h) const
    {
        Mat8 m(h, w, data + (size_t)y * step + x, step);
        m.owner = owner;
        return m;
    }
};

}  // namespace aos2
