// compat/Optimizer.cc -- int Optimizer::PoseOptimization(Frame *pFrame) with the reference's signature (include/Optimizer.h:48: a
// member of the Optimizer OBJECT Tracking holds as mpOptimizer in this fork; it reads none of the object's parameters,
// src/Optimizer.cc:283-495) over the C ABI, so that src/Tracking.cc:875,998,1040,1475,1555,1580 compile and link unchanged.
// Build: remove that one function from src/Optimizer.cc (the bundle adjustments, the essential graph and the Sim3 optimiser stay
// where they are, on g2o) and add this file; "Optimizer.h" is the reference's own header.
//
// It gathers what the reference's function reads from the Frame (mTcw, N, mvKeysUn, mvuRight, mvpMapPoints[i] and their
// GetWorldPos() under MapPoint::mGlobalMutex, :323), calls orbfe_pose_optimization -- g2o's Levenberg solver for this one-vertex
// problem, FP64 on the device (orbfe_pose.hip) -- and writes back what the reference writes: mvbOutlier for the entries that
// hold a point (:339,365,428-460), the pose through SetPose (:490), the inlier count as return value.
#include "Optimizer.h"

#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/orbfe.h"
#include "compat_util.h"

namespace ORB_SLAM2
{
using namespace orbfe_compat;

int Optimizer::PoseOptimization(Frame *pFrame)
{
    orbfe_context *ctx = context_of(*pFrame);
    float cam[5];
    check(ctx, orbfe_get_camera(ctx, cam)); // the solver projects with the context's fx, fy, cx, cy, bf (the reference: pFrame->fx ... pFrame->mbf)
    if (cam[0] != Frame::fx || cam[1] != Frame::fy || cam[2] != Frame::cx || cam[3] != Frame::cy || cam[4] != pFrame->mbf)
        throw std::runtime_error("Optimizer::PoseOptimization: the frame's extractor context was created for another camera");
    const int N = pFrame->N;
    std::vector<uint8_t> has(N > 0 ? N : 1, 0), outlier(N > 0 ? N : 1, 0);
    std::vector<float> Xw(3 * (size_t)(N > 0 ? N : 1), 0.f);
    int nInitialCorrespondences = 0;
    {
        std::unique_lock<std::mutex> lock(MapPoint::mGlobalMutex); // :323
        for (int i = 0; i < N; i++) {
            MapPoint *pMP = pFrame->mvpMapPoints[i];
            if (!pMP) continue;
            has[i] = 1;
            nInitialCorrespondences++;
            const cv::Mat X = pMP->GetWorldPos();
            for (int k = 0; k < 3; k++) Xw[3 * i + k] = X.at<float>(k);
        }
    }
    for (int i = 0; i < N; i++) outlier[i] = pFrame->mvbOutlier[i];
    float Tcw[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) Tcw[4 * r + c] = pFrame->mTcw.at<float>(r, c);
    int nInliers = 0;
    check(ctx, orbfe_pose_optimization(ctx, Tcw, N, keys_of(pFrame->mvKeysUn), pFrame->mvuRight.data(), has.data(), Xw.data(), outlier.data(), &nInliers));
    for (int i = 0; i < N; i++)
        if (has[i]) pFrame->mvbOutlier[i] = outlier[i] != 0;
    if (nInitialCorrespondences < 3) return 0; // :404-405: no SetPose
    cv::Mat pose(4, 4, CV_32F);
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) pose.at<float>(r, c) = Tcw[4 * r + c];
    pFrame->SetPose(pose);
    return nInliers;
}

} // namespace ORB_SLAM2
