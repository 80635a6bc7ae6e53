// SYNTAX-CHECK ONLY: declarations (no bodies, no arithmetic) of the OpenCV names tools/opencv_pin/pin.cpp uses, so that
// `g++ -fsyntax-only` can vouch for pin.cpp in an image without OpenCV (tests/test_reference_pins.py).  Never linked, never shipped,
// never used to build anything of the reference; the real headers replace it wherever the kit actually runs.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
typedef unsigned char uchar;
#define CV_VERSION "syntax-stub"
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_8UC4 24
#define CV_32F 5
#define CV_32FC1 5
int cvRound(float v);
int cvRound(double v);
namespace cv
{
struct Size { int width, height; Size(); Size(int w, int h); };
struct Point2f { float x, y; };
struct Point { int x, y; };
typedef Point Point2i;
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
template <class T, int N> struct Vec { T v[N]; T &operator[](int i); };
typedef Vec<uchar, 3> Vec3b;
typedef Vec<uchar, 4> Vec4b;
class MatExpr;
class Mat
{
public:
    int rows, cols;
    uchar *data;
    Mat();
    Mat(int r, int c, int type);
    Mat(const MatExpr &e);
    Mat &operator=(const MatExpr &e);
    template <class T> T &at(int r, int c);
    template <class T> const T &at(int r, int c) const;
    template <class T> T &at(int i);
    template <class T> T *ptr(int r = 0);
    template <class T> const T *ptr(int r = 0) const;
    Mat row(int r) const;
    Mat rowRange(int a, int b) const;
    Mat colRange(int a, int b) const;
    Mat reshape(int cn) const;
    Mat clone() const;
    void convertTo(Mat &dst, int type) const;
    static MatExpr ones(int r, int c, int type);
    static MatExpr eye(int r, int c, int type);
};
class MatExpr { public: operator Mat() const; };
MatExpr operator-(const Mat &a, const MatExpr &b);
MatExpr operator*(float s, const MatExpr &m);
typedef const Mat &InputArray;
typedef Mat &OutputArray;
enum { NORM_L1 = 2 };
enum { BORDER_REFLECT_101 = 4 };
double norm(const Mat &a, const Mat &b, int type);
float fastAtan2(float y, float x);
} // namespace cv
