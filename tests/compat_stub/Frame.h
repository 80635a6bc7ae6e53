// tests/compat_stub/Frame.h -- TEST-ONLY stand-in for the reference's include/Frame.h: the class DECLARATION (every member of
// include/Frame.h:42-214, same names, types and signatures; no code) so that orbslam2_amd/compat/Frame.cc -- which DEFINES these
// members over the C ABI, replacing src/Frame.cc -- and the other shims can be compiled where OpenCV and the reference tree
// are absent.  With the real tree one compiles compat/Frame.cc with -I<reference>/include instead and this file is not used.
#pragma once
#include <cstddef>
#include <vector>
#include "cvstub.h"
#include "fbow_stub.h"
#include "MapPoint.h"
#include "ORBextractor.h"

namespace ORB_SLAM2
{
#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64

class MapPoint;
class KeyFrame;

class Frame
{
public:
    Frame();
    Frame(const Frame &frame);
    // stereo, RGB-D, monocular (src/Tracking.cc:296,326,354-358)
    Frame(const cv::Mat &imLeft, const cv::Mat &imRight, const double &timeStamp, ORBextractor* extractorLeft, ORBextractor* extractorRight, fbow::Vocabulary* voc, cv::Mat &K, cv::Mat &distCoef, const float &bf, const float &thDepth);
    Frame(const cv::Mat &imGray, const cv::Mat &imDepth, const double &timeStamp, ORBextractor* extractor, fbow::Vocabulary* voc, cv::Mat &K, cv::Mat &distCoef, const float &bf, const float &thDepth);
    Frame(const cv::Mat &imGray, const double &timeStamp, ORBextractor* extractor, fbow::Vocabulary* voc, cv::Mat &K, cv::Mat &distCoef, const float &bf, const float &thDepth);

    void ExtractORB(int flag, const cv::Mat &im);
    void ComputeFboW();
    void SetPose(cv::Mat Tcw);
    void UpdatePoseMatrices();
    inline cv::Mat GetCameraCenter() { return mOw.clone(); }
    inline cv::Mat GetRotationInverse() { return mRwc.clone(); }
    bool isInFrustum(MapPoint* pMP, float viewingCosLimit);
    bool PosInGrid(const cv::KeyPoint &kp, int &posX, int &posY);
    std::vector<size_t> GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel=-1, const int maxLevel=-1) const;
    void ComputeStereoMatches();
    void ComputeStereoFromRGBD(const cv::Mat &imDepth);
    cv::Mat UnprojectStereo(const int &i);

    fbow::Vocabulary* mpFBOWvocabulary;
    ORBextractor* mpORBextractorLeft, *mpORBextractorRight;
    double mTimeStamp;
    cv::Mat mK;
    static float fx, fy, cx, cy, invfx, invfy;
    cv::Mat mDistCoef;
    float mbf, mb, mThDepth;
    int N;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    fbow::fBow mFbowVec;
    fbow::fBow2 mFbowFeatVec;
    cv::Mat mDescriptors, mDescriptorsRight;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    static float mfGridElementWidthInv, mfGridElementHeightInv;
    std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    cv::Mat mTcw;
    static long unsigned int nNextId;
    long unsigned int mnId;
    KeyFrame* mpReferenceKF;
    int mnScaleLevels;
    float mfScaleFactor, mfLogScaleFactor;
    std::vector<float> mvScaleFactors, mvInvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
    static bool mbInitialComputations;

private:
    void UndistortKeyPoints();
    void ComputeImageBounds(const cv::Mat &imLeft);
    void AssignFeaturesToGrid();
    cv::Mat mRcw, mtcw, mRwc, mOw;
};
} // namespace ORB_SLAM2
