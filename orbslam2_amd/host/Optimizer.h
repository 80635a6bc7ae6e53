// Optimizer.h -- host-side mirror of ORB_SLAM2::Optimizer::PoseOptimization (reference include/Optimizer.h:48,
// src/Optimizer.cc:283-495) over the C ABI (include/orbfe.h).  The reference takes a Frame*; this mirror takes the
// fields of the Frame it reads and writes, flattened:
//   reads   mTcw, N, mvKeysUn, mvuRight, mvpMapPoints[i] != NULL and GetWorldPos() of those, mvInvLevelSigma2 / fx, fy,
//           cx, cy, mbf (taken from the context the extractor of this camera was built with)
//   writes  mvbOutlier (entries that hold a map point), the pose handed to SetPose, and returns the inlier count
// A drop-in Optimizer::PoseOptimization(Frame *pFrame) gathers those fields, calls this, and finishes with
// pFrame->SetPose(cv::Mat(4, 4, CV_32F, Tcw).clone()) when at least 3 correspondences existed.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/orbfe.h"

namespace ORB_SLAM2
{

class Optimizer
{
public:
    // returns nInitialCorrespondences - nBad; Tcw (16 floats, row major) and vbOutlier are updated in place
    static int PoseOptimization(orbfe_context *ctx, float *Tcw, const std::vector<orbfe_keypoint> &vKeysUn,
                                const std::vector<float> &vuRight, const std::vector<uint8_t> &vbHasMapPoint,
                                const std::vector<float> &vWorldPos /* 3 per keypoint */, std::vector<uint8_t> &vbOutlier)
    {
        const size_t N = vKeysUn.size();
        if (vuRight.size() != N || vbHasMapPoint.size() != N || vWorldPos.size() != 3 * N)
            throw std::invalid_argument("Optimizer::PoseOptimization: per-keypoint arrays differ in length");
        vbOutlier.resize(N, 0);
        int nInliers = 0;
        const int rc = orbfe_pose_optimization(ctx, Tcw, (int)N, vKeysUn.data(), vuRight.data(), vbHasMapPoint.data(), vWorldPos.data(),
                                               vbOutlier.data(), &nInliers);
        if (rc != ORBFE_OK) throw std::runtime_error(std::string("orbfe: ") + orbfe_last_error(ctx));
        return nInliers;
    }
};

} // namespace ORB_SLAM2
