// tests/compat_stub/Frame.h -- TEST-ONLY declaration stand-in for the reference's include/Frame.h (members the shim reads / writes).
#pragma once
#include <map>
#include <vector>
#include "cvstub.h"
#include "MapPoint.h"
#include "ORBextractor.h"

namespace fbow { typedef std::map<uint32_t, std::vector<uint32_t> > fBow2; } // Thirdparty/fbow: node id -> feature indices

namespace ORB_SLAM2
{
class Frame
{
public:
    Frame() : mpORBextractorLeft(NULL), mpORBextractorRight(NULL), mb(0), N(0), mnId(0) {}
    ORBextractor *mpORBextractorLeft, *mpORBextractorRight;
    static float fx, fy, cx, cy, invfx, invfy;
    float mb;
    int N;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    fbow::fBow2 mFbowFeatVec;
    cv::Mat mDescriptors, mDescriptorsRight;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    static float mfGridElementWidthInv, mfGridElementHeightInv;
    cv::Mat mTcw;
    long unsigned int mnId;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
};
} // namespace ORB_SLAM2
