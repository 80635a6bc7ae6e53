// tests/compat_stub/cvstub.h -- TEST-ONLY declaration stand-in for the handful of OpenCV core types that
// orbslam2_amd/compat/ORBmatcher.cc touches (cv::Mat::at / ptr / rows / cols, cv::KeyPoint, cv::Point2f).  OpenCV is absent from
// this image, so this is what lets the reference-signature shim be compiled and run at all; it is never shipped and is not
// an OpenCV re-implementation (no arithmetic: a typed, shared, row-major buffer).
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

typedef unsigned char uchar;
#define CV_8U 0
#define CV_32F 5

namespace cv
{
struct Point2f { float x, y; Point2f() : x(0), y(0) {} Point2f(float a, float b) : x(a), y(b) {} };
struct KeyPoint { // field order of cv::KeyPoint (28 bytes)
    Point2f pt; float size, angle, response; int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
};
class Mat
{
public:
    int rows, cols;
    Mat() : rows(0), cols(0), esz_(0) {}
    Mat(int r, int c, int type) : rows(r), cols(c), esz_(type == CV_32F ? 4 : 1), buf_(std::make_shared<std::vector<uchar> >((size_t)r * c * (type == CV_32F ? 4 : 1), (uchar)0)) {}
    bool empty() const { return !buf_ || rows * cols == 0; }
    template <class T> T &at(int r, int c) { return *reinterpret_cast<T *>(buf_->data() + ((size_t)r * cols + c) * esz_); }
    template <class T> const T &at(int r, int c) const { return *reinterpret_cast<const T *>(buf_->data() + ((size_t)r * cols + c) * esz_); }
    template <class T> T &at(int i) { return *reinterpret_cast<T *>(buf_->data() + (size_t)i * esz_); }             // vectors (n x 1 or 1 x n)
    template <class T> const T &at(int i) const { return *reinterpret_cast<const T *>(buf_->data() + (size_t)i * esz_); }
    template <class T> T *ptr(int r = 0) { return reinterpret_cast<T *>(buf_->data() + (size_t)r * cols * esz_); }
    template <class T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(buf_->data() + (size_t)r * cols * esz_); }
    Mat clone() const { Mat m; m.rows = rows; m.cols = cols; m.esz_ = esz_; if (buf_) m.buf_ = std::make_shared<std::vector<uchar> >(*buf_); return m; }
private:
    int esz_;
    std::shared_ptr<std::vector<uchar> > buf_;
};
} // namespace cv
