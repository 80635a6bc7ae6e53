// compat/ORBmatcher.h -- ORB_SLAM2::ORBmatcher with the REFERENCE'S OWN SIGNATURES (include/ORBmatcher.h:38-102 of
// fabrizioromanelli/ORBSLAM2), implemented over the C ABI of liborbfe.so (include/orbfe.h).
//
// Replace the reference's include/ORBmatcher.h + src/ORBmatcher.cc by this header + compat/ORBmatcher.cc and
// Tracking / LocalMapping / LoopClosing compile unchanged:
//     ORBmatcher matcher(0.9,true);                                                    // src/Tracking.cc:698 (and :862,969,1283,1455,1496 with their ratios)
//     int nmatches = matcher.SearchForInitialization(mInitialFrame,mCurrentFrame,...);  // :699
//     int nmatches = matcher.SearchByFboW(mpReferenceKF,mCurrentFrame,vpMapPointMatches);// :867,1476
//     int nmatches = matcher.SearchByProjection(mCurrentFrame,mLastFrame,th,...);       // :985-992
//     matcher.SearchByProjection(mCurrentFrame,mvpLocalMapPoints,th);                    // :1290
//     matcher2.SearchByProjection(mCurrentFrame,vpCandidateKFs[i],sFound,10,100);        // Relocalization
// It needs the reference's Frame.h / KeyFrame.h / MapPoint.h (and therefore OpenCV) on the include path, with
// ORBextractor.h being the mirror in orbslam2_amd/host/ (Frame::mpORBextractorLeft->Context() is where the device context
// comes from; keyframe-only overloads use ORBextractor::DefaultContext()).  This repository cannot compile it against the
// real headers (OpenCV is absent from the image): tests/compat_stub/ holds a declaration-only stand-in of exactly the
// members used here, and tests/test_compat_matcher.py builds and runs this file against it on the GPU box.
//
// All arithmetic (window queries, Hamming distances, sequential accept rules, rotation histogram) is behind the C ABI;
// this file only flattens the pointer-rich arguments and writes the MapPoint* results back the way the reference does.
#ifndef ORBFE_COMPAT_ORBMATCHER_H
#define ORBFE_COMPAT_ORBMATCHER_H

#include <set>
#include <utility>
#include <vector>

#include "MapPoint.h"
#include "KeyFrame.h"
#include "Frame.h"

namespace ORB_SLAM2
{

class ORBmatcher
{
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);

    // Computes the Hamming distance between two ORB descriptors (src/ORBmatcher.cc:1643-1659)
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);

    // Tracking: local map points into the frame (src/ORBmatcher.cc:43-127)
    int SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th = 3);
    // Tracking: last frame's points into the current frame (:1324-1466)
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono);
    // Relocalisation: keyframe points into the frame (:1468-1595)
    int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th, const int ORBdist);
    // Loop closing: points through a Sim3 (:285-398)
    int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched, int th);

    // Vocabulary-node constrained matching (:157-283, :517-650)
    int SearchByFboW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches);
    int SearchByFboW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12);

    // Monocular initialisation (:400-515)
    int SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize = 10);

    // Local mapping: triangulation candidates under the epipolar constraint (:652-819)
    int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo);

    // Loop closing: mutual search through [s12*R12|t12] (:1098-1322)
    int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12, const float th);

    // Duplicate fusion (:821-971, :973-1096)
    int Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th = 3.0);
    int Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint);

public:
    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

protected:
    float RadiusByViewingCos(const float &viewCos);
    void ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3);

    float mfNNratio;
    bool mbCheckOrientation;
};

} // namespace ORB_SLAM2

#endif // ORBFE_COMPAT_ORBMATCHER_H
