// SYNTAX-CHECK ONLY (see core.hpp)
#pragma once
#include "core.hpp"
namespace cv { void undistortPoints(const Mat &src, Mat &dst, const Mat &K, const Mat &D, const Mat &R, const Mat &P); }
