// tests/compat_stub/Optimizer.h -- TEST-ONLY stand-in for the reference's include/Optimizer.h: the one declaration
// orbslam2_amd/compat/Optimizer.cc defines (include/Optimizer.h:46).
#pragma once
#include "Frame.h"
namespace ORB_SLAM2
{
class Optimizer
{
public:
    int static PoseOptimization(Frame* pFrame);
};
} // namespace ORB_SLAM2
