// compat/compat_util.h -- helpers shared by the reference-signature shims (compat/ORBmatcher.cc, Frame.cc, Optimizer.cc,
// KeyFrameDatabase.cc): type plumbing between the reference's objects and the C ABI, no arithmetic of the path itself.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "Frame.h"
#include "../../include/orbfe.h"

namespace ORB_SLAM2
{
namespace orbfe_compat
{

static_assert(sizeof(cv::KeyPoint) == sizeof(orbfe_keypoint), "cv::KeyPoint must be layout-identical to orbfe_keypoint");
static_assert(sizeof(cv::Point2f) == 2 * sizeof(float), "cv::Point2f must be two floats");

inline void check(orbfe_context *ctx, int rc)
{
    if (rc != ORBFE_OK) throw std::runtime_error(std::string("orbfe: ") + orbfe_last_error(ctx));
}

inline orbfe_context *context_of(const Frame &F)
{
    orbfe_context *ctx = F.mpORBextractorLeft ? F.mpORBextractorLeft->Context() : ORBextractor::DefaultContext();
    if (!ctx) throw std::runtime_error("ORBmatcher: the frame's ORBextractor has no device context yet (no image extracted)");
    return ctx;
}

inline const orbfe_keypoint *keys_of(const std::vector<cv::KeyPoint> &v) { return reinterpret_cast<const orbfe_keypoint *>(v.data()); }

// what the matchers read from a Frame (include/Frame.h:131-185)
inline orbfe_frame_view view_of(const Frame &F)
{
    orbfe_frame_view v = orbfe_frame_view(); // upload path; see device_view_of() for the resident one
    v.n = F.N;
    v.keys_un = keys_of(F.mvKeysUn);
    v.u_right = F.mvuRight.empty() ? nullptr : F.mvuRight.data();
    v.descriptors = F.mDescriptors.ptr<uchar>(0);
    v.min_x = Frame::mnMinX; v.max_x = Frame::mnMaxX; v.min_y = Frame::mnMinY; v.max_y = Frame::mnMaxY;
    return v;
}

// The frame every Tracking matcher searches IN is the current one, i.e. the latest extraction of its extractor: if F is that
// frame (ORBextractor::IsResidentFrame: same count, same descriptors) the matchers read it where the extraction left
// it in HBM and build its grid once; any other frame takes the upload path.
inline orbfe_frame_view device_view_of(const Frame &F)
{
    orbfe_frame_view v = view_of(F);
    if (F.mpORBextractorLeft && F.mpORBextractorLeft->IsResidentFrame(F.N, F.mDescriptors.ptr<uchar>(0))) v.device_slot_plus1 = 1;
    return v;
}

// top three rows of a 4 x 4 (or the whole of a 3 x 4) CV_32F pose, row major
inline void pose_3x4(const cv::Mat &T, float *out)
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) out[4 * r + c] = T.at<float>(r, c);
}
inline void pose_from_Rt(const cv::Mat &R, const cv::Mat &t, float *out)
{
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) out[4 * r + c] = R.at<float>(r, c);
        out[4 * r + 3] = t.at<float>(r);
    }
}

// MapPoint exposes only the scaled distances (GetMaxDistanceInvariance() = 1.2f * mfMaxDistance, GetMinDistanceInvariance() =
// 0.8f * mfMinDistance, src/MapPoint.cc:390-400); the C ABI takes the raw members (it forms the same products for the range
// gate and needs mfMaxDistance itself for PredictScale, src/MapPoint.cc:402-417).  Recover a raw value whose product is
// EXACTLY the scaled one, so the range gates are bit-identical; where two neighbouring floats share that product (the product
// crosses a binade in about one case in six) the one nearest to scaled / k is taken, which can move PredictScale's ratio by one
// ulp -- its ceil() changes only if log(ratio) / log(scaleFactor) is an integer to within 1e-7.
inline float raw_from_scaled(float scaled, float k)
{
    const float r0 = scaled / k;
    float best = r0;
    bool found = false;
    double best_err = 0.0;
    const float cand[5] = {r0, std::nextafter(r0, 0.0f), std::nextafter(r0, 3.0e38f), std::nextafter(std::nextafter(r0, 0.0f), 0.0f),
                           std::nextafter(std::nextafter(r0, 3.0e38f), 3.0e38f)};
    for (int i = 0; i < 5; i++) {
        if (k * cand[i] != scaled) continue;
        const double err = std::fabs((double)cand[i] - (double)scaled / (double)k);
        if (!found || err < best_err) { best = cand[i]; best_err = err; found = true; }
    }
    return best;
}

} // namespace orbfe_compat
} // namespace ORB_SLAM2
