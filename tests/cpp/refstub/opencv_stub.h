// TEST INFRASTRUCTURE: the handful of OpenCV names the reference-signature shims (active-orb-slam2_amd/host/ref/*.h)
// touch, so that they compile in an image without OpenCV.  Not a re-implementation of OpenCV: a cv::Mat here is a
// dense row-major 2-D array of float or uint8_t with the members the shims use (rows, cols, data, step, ptr<T>(),
// at<T>(), create(), clone(), empty(), row()).  In a real build these come from <opencv2/core/core.hpp>.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32F 5

namespace cv {

struct Point2f {
    float x = 0, y = 0;
};

struct KeyPoint {   // 28 bytes, same layout as OpenCV's
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};

class Mat {
public:
    int rows = 0, cols = 0;
    uint8_t *data = nullptr;
    size_t step = 0;   // bytes per row
    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    void create(int r, int c, int type)
    {
        type_ = type;
        rows = r;
        cols = c;
        step = (size_t)c * elem();
        buf_ = std::shared_ptr<uint8_t>(new uint8_t[(size_t)r * step + 1](), std::default_delete<uint8_t[]>());
        data = buf_.get();
    }
    bool empty() const { return rows == 0 || cols == 0 || !data; }
    int type() const { return type_; }
    Mat clone() const
    {
        Mat m;
        if (empty()) return m;
        m.create(rows, cols, type_);
        for (int r = 0; r < rows; ++r) memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols * elem());
        return m;
    }
    Mat row(int r) const   // a view (shares the buffer)
    {
        Mat m = *this;
        m.rows = 1;
        m.data = data + (size_t)r * step;
        return m;
    }
    template <typename T> T *ptr(int r = 0) { return reinterpret_cast<T *>(data + (size_t)r * step); }
    template <typename T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data + (size_t)r * step); }
    template <typename T> T &at(int r, int c = 0) { return ptr<T>(r)[c]; }
    template <typename T> const T &at(int r, int c = 0) const { return ptr<T>(r)[c]; }

private:
    size_t elem() const { return type_ == CV_32F ? 4 : 1; }
    int type_ = CV_8U;
    std::shared_ptr<uint8_t> buf_;
};

}  // namespace cv
