// tests/compat_stub/cvstub.h -- TEST-ONLY declaration stand-in for the handful of OpenCV core types that
// orbslam2_amd/compat/*.cc and the ORBFE_WITH_OPENCV overloads of orbslam2_amd/host/ORBextractor.h touch (cv::Mat::at / ptr /
// rows / cols / data / step / type, cv::KeyPoint, cv::Point2f, cv::InputArray / OutputArray).  OpenCV is absent from this
// image, so this is what lets the reference-signature shims be compiled and run at all; it is never shipped and is not an
// OpenCV re-implementation (no arithmetic: a typed, shared, row-major buffer).
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

typedef unsigned char uchar;
#define CV_8U 0
#define CV_8UC1 0
#define CV_16U 2
#define CV_32F 5
#define CV_32FC1 5

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
    uchar *data;
    size_t step; // bytes per row (cv::Mat::step converts to size_t)
    Mat() : rows(0), cols(0), data(NULL), step(0), type_(0) {}
    Mat(int r, int c, int type) : rows(0), cols(0), data(NULL), step(0), type_(0) { create(r, c, type); }
    Mat(int r, int c, int type, void *ext, size_t ext_step = 0) // header over caller memory (not owned)
        : rows(r), cols(c), data((uchar *)ext), step(ext_step ? ext_step : (size_t)c * esz(type)), type_(type) {}
    void create(int r, int c, int type)
    {
        if (buf_ && rows == r && cols == c && type_ == type) return;
        rows = r; cols = c; type_ = type; step = (size_t)c * esz(type);
        buf_ = std::make_shared<std::vector<uchar> >((size_t)r * step, (uchar)0);
        data = buf_->data();
    }
    void release() { buf_.reset(); rows = cols = 0; data = NULL; step = 0; }
    int type() const { return type_; }
    size_t total() const { return (size_t)rows * cols; }
    bool empty() const { return !data || rows * cols == 0; }
    template <class T> T &at(int r, int c) { return *reinterpret_cast<T *>(data + (size_t)r * step + (size_t)c * esz(type_)); }
    template <class T> const T &at(int r, int c) const { return *reinterpret_cast<const T *>(data + (size_t)r * step + (size_t)c * esz(type_)); }
    template <class T> T &at(int i) { return cols == 1 ? at<T>(i, 0) : at<T>(0, i); } // vectors (n x 1 or 1 x n)
    template <class T> const T &at(int i) const { return cols == 1 ? at<T>(i, 0) : at<T>(0, i); }
    template <class T> T *ptr(int r = 0) { return reinterpret_cast<T *>(data + (size_t)r * step); }
    template <class T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data + (size_t)r * step); }
    Mat clone() const
    {
        Mat m;
        if (empty()) return m;
        m.create(rows, cols, type_);
        for (int r = 0; r < rows; r++) std::memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, m.step);
        return m;
    }
private:
    static size_t esz(int type) { return type == CV_32F ? 4 : type == CV_16U ? 2 : 1; }
    int type_;
    std::shared_ptr<std::vector<uchar> > buf_;
};

// cv::InputArray / cv::OutputArray as the reference's ORBextractor::operator() declares them (include/ORBextractor.h:58-60):
// proxies over a Mat, nothing more
class _InputArray
{
public:
    _InputArray(const Mat &m) : m_(const_cast<Mat *>(&m)) {}
    Mat getMat() const { return *m_; }
    bool empty() const { return m_->empty(); }
protected:
    Mat *m_;
};
class _OutputArray : public _InputArray
{
public:
    _OutputArray(Mat &m) : _InputArray(m) {}
    void create(int r, int c, int type) const { m_->create(r, c, type); }
    void release() const { m_->release(); }
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
} // namespace cv
