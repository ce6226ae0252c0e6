// TEST INFRASTRUCTURE: the handful of OpenCV names the reference-signature classes (active-orb-slam2_amd/host/*.h)
// touch, so that they compile in an image without OpenCV.  Not a re-implementation of OpenCV: a cv::Mat here is a
// dense row-major 2-D array of float or uint8_t with the members the classes use (rows, cols, data, step, ptr<T>(),
// at<T>(), create(), clone(), empty(), row(), col(), rowRange(), colRange(), t(), dot(), and the small-matrix
// arithmetic of the few pose lines the matcher methods keep from their reference bodies: product, sum, difference,
// negation, scaling).  The arithmetic is plain: products accumulate in double and round to float once -- a stand-in, not
// OpenCV's gemm (the tests take the matrices these lines produce from the driver itself, so no OpenCV rounding behaviour
// is asserted through it).  In a real build all of this comes from <opencv2/core/core.hpp>.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5

namespace cv {

struct Point2f {
    float x = 0, y = 0;
    Point2f() = default;
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};

struct Rect {
    int x = 0, y = 0, width = 0, height = 0;
    Rect() = default;
    Rect(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {}
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
    // views (share the buffer)
    Mat rowRange(int r0, int r1) const
    {
        Mat m = *this;
        m.rows = r1 - r0;
        m.data = data + (size_t)r0 * step;
        return m;
    }
    Mat colRange(int c0, int c1) const
    {
        Mat m = *this;
        m.cols = c1 - c0;
        m.data = data + (size_t)c0 * elem();
        return m;
    }
    Mat col(int c) const { return colRange(c, c + 1); }
    Mat operator()(const Rect &r) const { return rowRange(r.y, r.y + r.height).colRange(r.x, r.x + r.width); }
    bool isContinuous() const { return step == (size_t)cols * elem(); }
    void release() { *this = Mat(); }
    Mat t() const
    {
        Mat m(cols, rows, CV_32F);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) m.at<float>(c, r) = at<float>(r, c);
        return m;
    }
    double dot(const Mat &o) const
    {
        double s = 0;
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) s += (double)at<float>(r, c) * (double)o.at<float>(r, c);
        return s;
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

// small float matrix arithmetic (see the header comment)
inline Mat operator*(const Mat &a, const Mat &b)
{
    Mat m(a.rows, b.cols, CV_32F);
    for (int r = 0; r < a.rows; ++r)
        for (int c = 0; c < b.cols; ++c) {
            double s = 0;
            for (int k = 0; k < a.cols; ++k) s += (double)a.at<float>(r, k) * (double)b.at<float>(k, c);
            m.at<float>(r, c) = (float)s;
        }
    return m;
}
inline Mat elementwise(const Mat &a, const Mat &b, float sb)
{
    Mat m(a.rows, a.cols, CV_32F);
    for (int r = 0; r < a.rows; ++r)
        for (int c = 0; c < a.cols; ++c) m.at<float>(r, c) = a.at<float>(r, c) + sb * b.at<float>(r, c);
    return m;
}
inline Mat operator+(const Mat &a, const Mat &b) { return elementwise(a, b, 1.0f); }
inline Mat operator-(const Mat &a, const Mat &b) { return elementwise(a, b, -1.0f); }
inline Mat operator*(double s, const Mat &a)
{
    Mat m(a.rows, a.cols, CV_32F);
    for (int r = 0; r < a.rows; ++r)
        for (int c = 0; c < a.cols; ++c) m.at<float>(r, c) = (float)(s * (double)a.at<float>(r, c));
    return m;
}
inline Mat operator-(const Mat &a) { return -1.0 * a; }
inline Mat operator/(const Mat &a, double s) { return (1.0 / s) * a; }

// cv::InputArray / cv::OutputArray as ORBextractor::operator() uses them (getMat, empty, create, release)
class _InputArray {
public:
    _InputArray() = default;
    _InputArray(const Mat &m) : m_(&m) {}
    Mat getMat() const { return m_ ? *m_ : Mat(); }
    bool empty() const { return !m_ || m_->empty(); }

private:
    const Mat *m_ = nullptr;
};
class _OutputArray {
public:
    _OutputArray(Mat &m) : m_(&m) {}
    void create(int rows, int cols, int type) const { m_->create(rows, cols, type); }
    void release() const { m_->release(); }
    Mat getMat() const { return *m_; }

private:
    Mat *m_;
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
inline InputArray noArray()
{
    static const _InputArray none;
    return none;
}

}  // namespace cv
