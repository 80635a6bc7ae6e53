// ORBmatcher.h -- host-side mirror of ORB_SLAM2::ORBmatcher (reference include/ORBmatcher.h:38-102) over
// the C ABI (include/orbfe.h).  Same class name, constructor (nnratio, checkOri), constants and method
// names for the Tracking-thread overloads; the pointer-rich arguments (Frame&, KeyFrame*, MapPoint*) are
// replaced by the flattened views a maintainer fills from those objects (INTEGRATION.md shows how):
//   FrameView   <- Frame::N, mvKeysUn, mvuRight, mDescriptors, mnMin/MaxX/Y          (include/Frame.h)
//   PointArrays <- per map point: GetWorldPos(), GetDescriptor(), Observations(), ... (include/MapPoint.h)
// Results come back as index arrays: "keypoint k of the frame received point i" instead of MapPoint*.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/orbfe.h"

namespace ORB_SLAM2
{

struct FrameView {
    std::vector<orbfe_keypoint> mvKeysUn;
    std::vector<float> mvuRight;          // empty for monocular
    std::vector<uint8_t> mDescriptors;    // N x 32
    float mnMinX = 0.f, mnMaxX = 0.f, mnMinY = 0.f, mnMaxY = 0.f;
    int deviceSlot = -1; // >= 0: this frame is image slot deviceSlot of the extractor's latest call (matchers read it in HBM)
    orbfe_frame_view c_view() const
    {
        orbfe_frame_view v = orbfe_frame_view();
        v.device_slot_plus1 = deviceSlot + 1;
        v.n = (int32_t)mvKeysUn.size();
        v.keys_un = mvKeysUn.data();
        v.u_right = mvuRight.empty() ? nullptr : mvuRight.data();
        v.descriptors = mDescriptors.data();
        v.min_x = mnMinX; v.max_x = mnMaxX; v.min_y = mnMinY; v.max_y = mnMaxY;
        return v;
    }
};

// Map points seen from a frame / keyframe, one entry per keypoint slot of that frame.
struct PointArrays {
    std::vector<float> pos;        // GetWorldPos(), n x 3
    std::vector<uint8_t> desc;     // GetDescriptor(), n x 32
    std::vector<int32_t> valid;    // pMP != NULL && usable (see each method)
    std::vector<int32_t> obs;      // Observations()
    std::vector<int32_t> octave;   // mvKeys[i].octave of the owning frame
    std::vector<float> angle;      // mvKeysUn[i].angle of the owning frame
    std::vector<float> maxDistance, minDistance; // mfMaxDistance, mfMinDistance
    int size() const { return (int)valid.size(); }
};

class ORBmatcher
{
public:
    // include/ORBmatcher.h:42
    ORBmatcher(orbfe_context *ctx, float nnratio = 0.6, bool checkOri = true) : mCtx(ctx), mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    static const int TH_LOW = 50;   // src/ORBmatcher.cc:35-37
    static const int TH_HIGH = 100;
    static const int HISTO_LENGTH = 30;

    // ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1643-1659) on two 32-byte rows.
    static int DescriptorDistance(const uint8_t *a, const uint8_t *b)
    {
        int dist = 0;
        for (int i = 0; i < 32; i++) dist += __builtin_popcount((unsigned)(a[i] ^ b[i]));
        return dist;
    }

    // SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono), src/ORBmatcher.cc:1324.
    // Tcw are the top 3x4 rows of Frame::mTcw.  curMatch[k] = index into `last`, or -1.
    int SearchByProjection(const FrameView &CurrentFrame, const float *TcwCur, const float *TcwLast, const PointArrays &last,
                           const std::vector<uint8_t> &curHasObs, float th, bool bMono, std::vector<int32_t> &curMatch)
    {
        const orbfe_frame_view v = CurrentFrame.c_view();
        curMatch.assign(v.n > 0 ? v.n : 1, -1);
        int n = 0;
        Check(orbfe_search_by_projection_last(mCtx, &v, TcwCur, TcwLast, last.size(), last.pos.data(), last.desc.data(), last.valid.data(),
                                              last.obs.data(), last.octave.data(), last.angle.data(),
                                              curHasObs.empty() ? nullptr : curHasObs.data(), th, bMono ? 1 : 0, mbCheckOrientation ? 1 : 0,
                                              curMatch.data(), &n));
        curMatch.resize(v.n);
        return n;
    }

    // SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th), src/ORBmatcher.cc:43; the
    // track points are the outputs of Frame::isInFrustum (orbfe_is_in_frustum).
    int SearchByProjection(const FrameView &F, const std::vector<orbfe_track_point> &pts, const std::vector<uint8_t> &desc,
                           const std::vector<int32_t> &obs, const std::vector<uint8_t> &hasObs, float th, std::vector<int32_t> &match)
    {
        const orbfe_frame_view v = F.c_view();
        match.assign(v.n > 0 ? v.n : 1, -1);
        int n = 0;
        Check(orbfe_search_by_projection_points(mCtx, &v, (int)pts.size(), pts.data(), desc.data(), obs.data(),
                                                hasObs.empty() ? nullptr : hasObs.data(), th, mfNNratio, match.data(), &n));
        match.resize(v.n);
        return n;
    }

    // SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist), src/ORBmatcher.cc:1468
    int SearchByProjection(const FrameView &CurrentFrame, const float *TcwCur, const PointArrays &kf, const std::vector<uint8_t> &curHasPoint,
                           float th, int ORBdist, std::vector<int32_t> &curMatch)
    {
        const orbfe_frame_view v = CurrentFrame.c_view();
        curMatch.assign(v.n > 0 ? v.n : 1, -1);
        int n = 0;
        Check(orbfe_search_by_projection_kf(mCtx, &v, TcwCur, kf.size(), kf.pos.data(), kf.desc.data(), kf.valid.data(), kf.angle.data(),
                                            kf.maxDistance.data(), kf.minDistance.data(), curHasPoint.empty() ? nullptr : curHasPoint.data(),
                                            th, ORBdist, mbCheckOrientation ? 1 : 0, curMatch.data(), &n));
        curMatch.resize(v.n);
        return n;
    }

    // SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize), src/ORBmatcher.cc:400
    int SearchForInitialization(const FrameView &F1, const FrameView &F2, std::vector<float> &vbPrevMatched /* n1 x 2 */,
                                std::vector<int32_t> &vnMatches12, int windowSize = 10)
    {
        const orbfe_frame_view v1 = F1.c_view(), v2 = F2.c_view();
        vnMatches12.assign(v1.n > 0 ? v1.n : 1, -1);
        int n = 0;
        Check(orbfe_search_for_initialization(mCtx, &v1, &v2, vbPrevMatched.data(), windowSize, mfNNratio, mbCheckOrientation ? 1 : 0,
                                              vnMatches12.data(), &n));
        vnMatches12.resize(v1.n);
        return n;
    }

    // SearchByFboW(KeyFrame *pKF, Frame &F, vpMapPointMatches), src/ORBmatcher.cc:157.  FeatVec = the fBow2 map of
    // Frame::ComputeFboW / KeyFrame::mFbowFeatVec flattened by orbfe_bow_maps; fMatch[j] = KF keypoint or -1.
    struct FeatVec {
        std::vector<uint32_t> nodes;
        std::vector<int32_t> off, feat; // node k -> feat[off[k] .. off[k+1])
    };
    int SearchByFboW(const FeatVec &kfFeat, const std::vector<int32_t> &kfValid, const std::vector<uint8_t> &kfDesc,
                     const std::vector<float> &kfAngle, const FeatVec &fFeat, const std::vector<uint8_t> &fDesc,
                     const std::vector<float> &fAngle, std::vector<int32_t> &fMatch)
    {
        const int nk = (int)kfValid.size(), nf = (int)fAngle.size();
        fMatch.assign(nf > 0 ? nf : 1, -1);
        int n = 0;
        Check(orbfe_search_by_bow(mCtx, kfFeat.nodes.data(), kfFeat.off.data(), kfFeat.feat.data(), (int)kfFeat.nodes.size(),
                                  kfValid.data(), kfDesc.data(), kfAngle.data(), nk, fFeat.nodes.data(), fFeat.off.data(),
                                  fFeat.feat.data(), (int)fFeat.nodes.size(), fDesc.data(), fAngle.data(), nf, mfNNratio,
                                  mbCheckOrientation ? 1 : 0, fMatch.data(), &n));
        fMatch.resize(nf);
        return n;
    }

    // SearchByFboW(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12), src/ORBmatcher.cc:517; vnMatches12[i1] = KF2 keypoint or -1.
    int SearchByFboW(const FeatVec &feat1, const std::vector<int32_t> &valid1, const std::vector<uint8_t> &desc1,
                     const std::vector<float> &angle1, const FeatVec &feat2, const std::vector<int32_t> &valid2,
                     const std::vector<uint8_t> &desc2, const std::vector<float> &angle2, std::vector<int32_t> &vnMatches12)
    {
        const int n1 = (int)valid1.size(), n2 = (int)valid2.size();
        vnMatches12.assign(n1 > 0 ? n1 : 1, -1);
        int n = 0;
        Check(orbfe_search_by_bow_kf(mCtx, feat1.nodes.data(), feat1.off.data(), feat1.feat.data(), (int)feat1.nodes.size(),
                                     valid1.data(), desc1.data(), angle1.data(), n1, feat2.nodes.data(), feat2.off.data(),
                                     feat2.feat.data(), (int)feat2.nodes.size(), valid2.data(), desc2.data(), angle2.data(), n2,
                                     mfNNratio, mbCheckOrientation ? 1 : 0, vnMatches12.data(), &n));
        vnMatches12.resize(n1);
        return n;
    }

    // SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo), src/ORBmatcher.cc:652 (LocalMapping::CreateNewMapPoints).
    // hasMapPoint = the keypoint already has a map point; Cw1 = pKF1->GetCameraCenter(), T2w = top 3x4 rows of pKF2's pose.
    int SearchForTriangulation(const FeatVec &feat1, const std::vector<orbfe_keypoint> &keysUn1, const std::vector<float> &uRight1,
                               const std::vector<uint8_t> &hasMapPoint1, const std::vector<uint8_t> &desc1,
                               const FeatVec &feat2, const std::vector<orbfe_keypoint> &keysUn2, const std::vector<float> &uRight2,
                               const std::vector<uint8_t> &hasMapPoint2, const std::vector<uint8_t> &desc2,
                               const float F12[9], const float Cw1[3], const float T2w[12], float fx2, float fy2, float cx2, float cy2,
                               std::vector<std::pair<size_t, size_t> > &vMatchedPairs, bool bOnlyStereo)
    {
        const int n1 = (int)keysUn1.size(), n2 = (int)keysUn2.size();
        std::vector<int32_t> m12(n1 > 0 ? n1 : 1, -1);
        int n = 0;
        Check(orbfe_search_for_triangulation(mCtx, feat1.nodes.data(), feat1.off.data(), feat1.feat.data(), (int)feat1.nodes.size(),
                                             keysUn1.data(), uRight1.data(), hasMapPoint1.data(), desc1.data(), n1,
                                             feat2.nodes.data(), feat2.off.data(), feat2.feat.data(), (int)feat2.nodes.size(),
                                             keysUn2.data(), uRight2.data(), hasMapPoint2.data(), desc2.data(), n2,
                                             F12, Cw1, T2w, fx2, fy2, cx2, cy2, bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0, m12.data(), &n));
        vMatchedPairs.clear();
        vMatchedPairs.reserve(n);
        for (int i = 0; i < n1; i++)
            if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
        return n;
    }

    // Map points offered to a keyframe (Fuse, Sim3 matchers): one entry per candidate point
    struct CandidatePoints {
        std::vector<float> pos, normal;            // GetWorldPos(), GetNormal(), n x 3
        std::vector<float> maxDistance, minDistance; // mfMaxDistance, mfMinDistance
        std::vector<uint8_t> desc;                 // GetDescriptor(), n x 32
        std::vector<int32_t> valid;                // pMP && !isBad() && not already in the keyframe / matched set
        int size() const { return (int)valid.size(); }
    };

    // Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, th), src/ORBmatcher.cc:821: SEARCH PART.  bestIdx[i] = keypoint of
    // pKF to fuse point i with, or -1; the caller then runs the reference's :943-964 (Replace / AddObservation) on it.
    int Fuse(const FrameView &pKF, const float *Tcw, const CandidatePoints &pts, float th, std::vector<int32_t> &bestIdx)
    {
        const orbfe_frame_view v = pKF.c_view();
        bestIdx.assign(pts.size() > 0 ? pts.size() : 1, -1);
        int n = 0;
        Check(orbfe_fuse(mCtx, &v, Tcw, pts.size(), pts.pos.data(), pts.normal.data(), pts.maxDistance.data(), pts.minDistance.data(),
                         pts.desc.data(), pts.valid.data(), th, bestIdx.data(), &n));
        bestIdx.resize(pts.size());
        return n;
    }
    // Fuse(KeyFrame *pKF, cv::Mat Scw, vpPoints, th, vpReplacePoint), src/ORBmatcher.cc:973: search part, Scw = top 3x4 rows
    int Fuse(const FrameView &pKF, const float *Scw, const CandidatePoints &pts, float th, std::vector<int32_t> &bestIdx, bool /*sim3*/)
    {
        const orbfe_frame_view v = pKF.c_view();
        bestIdx.assign(pts.size() > 0 ? pts.size() : 1, -1);
        int n = 0;
        Check(orbfe_fuse_sim3(mCtx, &v, Scw, pts.size(), pts.pos.data(), pts.normal.data(), pts.maxDistance.data(), pts.minDistance.data(),
                              pts.desc.data(), pts.valid.data(), th, bestIdx.data(), &n));
        bestIdx.resize(pts.size());
        return n;
    }
    // SearchByProjection(KeyFrame *pKF, cv::Mat Scw, vpPoints, vpMatched, th), src/ORBmatcher.cc:285; kfMatched[k] = vpMatched[k] != NULL
    int SearchByProjection(const FrameView &pKF, const float *Scw, const CandidatePoints &pts, const std::vector<uint8_t> &kfMatched, int th,
                           std::vector<int32_t> &ptMatch)
    {
        const orbfe_frame_view v = pKF.c_view();
        ptMatch.assign(pts.size() > 0 ? pts.size() : 1, -1);
        int n = 0;
        Check(orbfe_search_by_projection_sim3(mCtx, &v, Scw, pts.size(), pts.pos.data(), pts.normal.data(), pts.maxDistance.data(),
                                              pts.minDistance.data(), pts.desc.data(), pts.valid.data(),
                                              kfMatched.empty() ? nullptr : kfMatched.data(), (float)th, ptMatch.data(), &n));
        ptMatch.resize(pts.size());
        return n;
    }
    // SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th), src/ORBmatcher.cc:1098; pts1 / pts2 have one entry per keypoint slot
    int SearchBySim3(const FrameView &pKF1, const float *T1w, const CandidatePoints &pts1, const FrameView &pKF2, const float *T2w,
                     const CandidatePoints &pts2, float s12, const float R12[9], const float t12[3], float th, std::vector<int32_t> &vnMatches12)
    {
        const orbfe_frame_view v1 = pKF1.c_view(), v2 = pKF2.c_view();
        vnMatches12.assign(v1.n > 0 ? v1.n : 1, -1);
        int n = 0;
        Check(orbfe_search_by_sim3(mCtx, &v1, T1w, pts1.pos.data(), pts1.maxDistance.data(), pts1.minDistance.data(), pts1.desc.data(),
                                   pts1.valid.data(), &v2, T2w, pts2.pos.data(), pts2.maxDistance.data(), pts2.minDistance.data(),
                                   pts2.desc.data(), pts2.valid.data(), s12, R12, t12, th, vnMatches12.data(), &n));
        vnMatches12.resize(v1.n);
        return n;
    }

    // ComputeThreeMaxima(histo, L, ind1, ind2, ind3), src/ORBmatcher.cc:1597
    void ComputeThreeMaxima(const std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3)
    {
        std::vector<int32_t> sizes(L);
        for (int i = 0; i < L; i++) sizes[i] = (int32_t)histo[i].size();
        orbfe_three_maxima(sizes.data(), L, &ind1, &ind2, &ind3);
    }

protected:
    void Check(int rc)
    {
        if (rc != ORBFE_OK) throw std::runtime_error(std::string("orbfe: ") + orbfe_last_error(mCtx));
    }
    orbfe_context *mCtx;
    float mfNNratio;
    bool mbCheckOrientation;
};

} // namespace ORB_SLAM2
