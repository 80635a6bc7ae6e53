// SYNTAX-CHECK ONLY (see core.hpp)
#pragma once
#include "core.hpp"
namespace cv { void FAST(const Mat &img, std::vector<KeyPoint> &kps, int threshold, bool nonmax); }
