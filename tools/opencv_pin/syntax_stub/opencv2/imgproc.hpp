// SYNTAX-CHECK ONLY (see core.hpp)
#pragma once
#include "core.hpp"
namespace cv
{
enum { INTER_LINEAR = 1 };
enum { COLOR_BGR2GRAY = 6, COLOR_RGB2GRAY = 7, COLOR_BGRA2GRAY = 10, COLOR_RGBA2GRAY = 11 };
void resize(const Mat &src, Mat &dst, Size sz, double fx, double fy, int interpolation);
void GaussianBlur(const Mat &src, Mat &dst, Size k, double sx, double sy, int border);
void cvtColor(const Mat &src, Mat &dst, int code);
void remap(const Mat &src, Mat &dst, const Mat &m1, const Mat &m2, int interpolation);
} // namespace cv
