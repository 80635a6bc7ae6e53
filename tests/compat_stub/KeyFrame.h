// tests/compat_stub/KeyFrame.h -- TEST-ONLY declaration stand-in for the reference's include/KeyFrame.h (members the shim uses).
#pragma once
#include <set>
#include <vector>
#include "cvstub.h"
#include "Frame.h"
#include "MapPoint.h"

namespace ORB_SLAM2
{
class KeyFrame
{
public:
    explicit KeyFrame(Frame &F)
        : fx(Frame::fx), fy(Frame::fy), cx(Frame::cx), cy(Frame::cy), invfx(Frame::invfx), invfy(Frame::invfy), mbf(F.mbf), mb(F.mb), mThDepth(F.mThDepth), N(F.N), mvKeys(F.mvKeys), mvKeysUn(F.mvKeysUn), mvuRight(F.mvuRight),
          mvDepth(F.mvDepth), mDescriptors(F.mDescriptors.clone()), mFbowFeatVec(F.mFbowFeatVec), mnMinX((int)Frame::mnMinX), mnMinY((int)Frame::mnMinY),
          mnMaxX((int)Frame::mnMaxX), mnMaxY((int)Frame::mnMaxY), mvpMapPoints(F.mvpMapPoints), Tcw(F.mTcw.clone())
    {
    }
    cv::Mat GetPose() { return Tcw.clone(); }
    cv::Mat GetRotation() { cv::Mat R(3, 3, CV_32F); for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R.at<float>(r, c) = Tcw.at<float>(r, c); return R; }
    cv::Mat GetTranslation() { cv::Mat t(3, 1, CV_32F); for (int r = 0; r < 3; r++) t.at<float>(r) = Tcw.at<float>(r, 3); return t; }
    cv::Mat GetCameraCenter()
    { // Ow = -Rcw.t() * tcw
        cv::Mat o(3, 1, CV_32F);
        for (int i = 0; i < 3; i++) o.at<float>(i) = ((-Tcw.at<float>(0, i)) * Tcw.at<float>(0, 3) + (-Tcw.at<float>(1, i)) * Tcw.at<float>(1, 3)) + (-Tcw.at<float>(2, i)) * Tcw.at<float>(2, 3);
        return o;
    }
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    std::set<MapPoint *> GetMapPoints() { std::set<MapPoint *> s; for (size_t i = 0; i < mvpMapPoints.size(); i++) if (mvpMapPoints[i] && !mvpMapPoints[i]->isBad()) s.insert(mvpMapPoints[i]); return s; }
    MapPoint *GetMapPoint(const size_t &idx) { return mvpMapPoints[idx]; }
    void AddMapPoint(MapPoint *pMP, const size_t &idx) { mvpMapPoints[idx] = pMP; }

    const float fx, fy, cx, cy, invfx, invfy, mbf, mb, mThDepth;
    // what KeyFrameDatabase reads and writes (include/KeyFrame.h:133,153-158; covisibility graph :62-72)
    long unsigned int mnId = 0;
    fbow::fBow mFbowVec;
    long unsigned int mnLoopQuery = 0; int mnLoopWords = 0; float mLoopScore = 0.f;
    long unsigned int mnRelocQuery = 0; int mnRelocWords = 0; float mRelocScore = 0.f;
    std::vector<KeyFrame *> GetBestCovisibilityKeyFrames(const int &N) { return std::vector<KeyFrame *>(mvpOrderedConnectedKeyFrames.begin(), mvpOrderedConnectedKeyFrames.begin() + ((int)mvpOrderedConnectedKeyFrames.size() < N ? (int)mvpOrderedConnectedKeyFrames.size() : N)); }
    std::set<KeyFrame *> GetConnectedKeyFrames() { return std::set<KeyFrame *>(mConnected.begin(), mConnected.end()); }
    std::vector<KeyFrame *> mvpOrderedConnectedKeyFrames; // protected in the reference
    std::vector<KeyFrame *> mConnected;                   // test fixture: the keys of mConnectedKeyFrameWeights
    const int N;
    const std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    const std::vector<float> mvuRight, mvDepth;
    const cv::Mat mDescriptors;
    fbow::fBow2 mFbowFeatVec;
    const int mnMinX, mnMinY, mnMaxX, mnMaxY;

    std::vector<MapPoint *> mvpMapPoints; // protected in the reference
    cv::Mat Tcw;
};
} // namespace ORB_SLAM2
