// tests/compat_stub/Optimizer.h -- TEST-ONLY stand-in for the reference's include/Optimizer.h: in this fork Optimizer is an OBJECT
// (constructed from the settings file, src/Optimizer.cc:40-86; Tracking holds mpOptimizer) and PoseOptimization a plain member
// (include/Optimizer.h:48) that uses none of the object's parameters (src/Optimizer.cc:283-495: constants).  Only the two
// declarations the shim and the driver touch; the constructor is defined by the driver, orbslam2_amd/compat/Optimizer.cc defines
// PoseOptimization.
#pragma once
#include <string>
#include "Frame.h"
namespace ORB_SLAM2
{
using std::string;
class Optimizer
{
public:
  Optimizer(const string &strSettingPath);
  int PoseOptimization(Frame* pFrame);
};
} // namespace ORB_SLAM2
