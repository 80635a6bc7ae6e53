// the extractor a Frame points to is the mirror over the C ABI (an integration replaces include/ORBextractor.h by it), with
// the cv::InputArray / cv::OutputArray call operator of the reference (include/ORBextractor.h:58-60) switched on
#pragma once
#ifndef ORBFE_WITH_OPENCV
#define ORBFE_WITH_OPENCV 1
#endif
#include "../../orbslam2_amd/host/ORBextractor.h"
