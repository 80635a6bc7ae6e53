// compat/ORBmatcher.cc -- reference-signature ORBmatcher over the C ABI (see compat/ORBmatcher.h).
//
// Every method does three things and nothing else:
//   1. flatten: MapPoint* -> rows of plain arrays (GetWorldPos, GetDescriptor, Observations, ...), Frame / KeyFrame ->
//      orbfe_frame_view (zero copy: cv::KeyPoint is layout-identical to orbfe_keypoint, mDescriptors is a continuous
//      N x 32 CV_8U matrix, mTcw a continuous 4 x 4 CV_32F matrix);
//   2. call the orbfe_* entry point that restates the reference method (file:line cited at each call);
//   3. write the result back exactly where the reference writes it (mvpMapPoints, vpMatches12, vpMatched, ...).
// No matching arithmetic lives here.
#include "ORBmatcher.h"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../include/orbfe.h"
#include "compat_util.h"

namespace ORB_SLAM2
{
using namespace orbfe_compat;

const int ORBmatcher::TH_HIGH = 100;     // src/ORBmatcher.cc:35-37
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;


namespace
{

// The keyframe-only overloads have no Frame to reach an extractor through: they use the process-wide default context
// (ORBextractor::DefaultContext()).  orbfe_fuse / _sim3 / search_by_sim3 project with the CONTEXT's fx, fy, cx, cy, bf and scale
// tables where the reference reads pKF->fx ... (src/ORBmatcher.cc:831-835), so a default context of another camera would give
// wrong matches silently: refuse it.
orbfe_context *kf_context(const KeyFrame *pKF)
{
    orbfe_context *ctx = ORBextractor::DefaultContext();
    if (!ctx) throw std::runtime_error("ORBmatcher: no ORBextractor device context exists in this process");
    float cam[5];
    check(ctx, orbfe_get_camera(ctx, cam));
    if (cam[0] != pKF->fx || cam[1] != pKF->fy || cam[2] != pKF->cx || cam[3] != pKF->cy || cam[4] != pKF->mbf)
        throw std::runtime_error("ORBmatcher: the default device context was created for another camera than this KeyFrame's "
                                 "(call SetCamera() on the left ORBextractor before its first frame, or SetAsDefault() on the right one)");
    return ctx;
}

// ... and from a KeyFrame (include/KeyFrame.h:160-199).  The keyframe's grid is the frame's (filled with the frame's FLOAT bounds and
// cell size, src/KeyFrame.cc:32-50) while its own bounds are ints; Frame's bounds are static and identical for every frame of a
// run, so the view carries those floats plus the keyframe flag, and the library truncates them where KeyFrame.cc uses the ints.
orbfe_frame_view view_of(const KeyFrame *pKF)
{
    orbfe_frame_view v = orbfe_frame_view();
    v.n = pKF->N;
    v.keys_un = keys_of(pKF->mvKeysUn);
    v.u_right = pKF->mvuRight.empty() ? nullptr : pKF->mvuRight.data();
    v.descriptors = pKF->mDescriptors.ptr<uchar>(0);
    v.min_x = Frame::mnMinX; v.max_x = Frame::mnMaxX; v.min_y = Frame::mnMinY; v.max_y = Frame::mnMaxY;
    v.keyframe = 1;
    return v;
}

// rows of per-point arrays, one entry per element of a vector<MapPoint*>
struct PointRows {
    std::vector<float> pos, normal, maxd, mind;
    std::vector<uint8_t> desc;
    std::vector<int32_t> valid, obs;
    explicit PointRows(size_t n) : pos(3 * n, 0.f), normal(3 * n, 0.f), maxd(n, 0.f), mind(n, 0.f), desc(32 * (n ? n : 1), 0), valid(n, 0), obs(n, 0) {}
    void set(size_t i, MapPoint *pMP, bool with_normal, bool with_dist)
    {
        const cv::Mat p = pMP->GetWorldPos();
        for (int k = 0; k < 3; k++) pos[3 * i + k] = p.at<float>(k);
        const cv::Mat d = pMP->GetDescriptor();
        std::memcpy(&desc[32 * i], d.ptr<uchar>(0), 32);
        obs[i] = pMP->Observations();
        if (with_normal) {
            const cv::Mat nv = pMP->GetNormal();
            for (int k = 0; k < 3; k++) normal[3 * i + k] = nv.at<float>(k);
        }
        if (with_dist) {
            maxd[i] = raw_from_scaled(pMP->GetMaxDistanceInvariance(), 1.2f);
            mind[i] = raw_from_scaled(pMP->GetMinDistanceInvariance(), 0.8f);
        }
        valid[i] = 1;
    }
};

// fbow::fBow2 (std::map<uint32_t, std::vector<uint32_t>>) as the CSR the C ABI takes (what orbfe_bow_maps produces)
struct FeatCSR {
    std::vector<uint32_t> nodes;
    std::vector<int32_t> off, feat;
    template <class FeatVec> explicit FeatCSR(const FeatVec &fv)
    {
        off.push_back(0);
        for (typename FeatVec::const_iterator it = fv.begin(); it != fv.end(); ++it) {
            nodes.push_back((uint32_t)it->first);
            for (size_t k = 0; k < it->second.size(); k++) feat.push_back((int32_t)it->second[k]);
            off.push_back((int32_t)feat.size());
        }
        if (feat.empty()) feat.push_back(0);
    }
};

std::vector<float> angles_of(const std::vector<cv::KeyPoint> &keys)
{
    std::vector<float> a(keys.size() ? keys.size() : 1, 0.f);
    for (size_t i = 0; i < keys.size(); i++) a[i] = keys[i].angle;
    return a;
}

} // namespace

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b)
{
    const uint32_t *pa = a.ptr<uint32_t>(0), *pb = b.ptr<uint32_t>(0);
    int dist = 0;
    for (int i = 0; i < 8; i++) dist += __builtin_popcount(pa[i] ^ pb[i]);
    return dist;
}

float ORBmatcher::RadiusByViewingCos(const float &viewCos) { return viewCos > 0.998 ? 2.5f : 4.0f; } // :129-135

void ORBmatcher::ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3)
{
    std::vector<int32_t> sizes(L > 0 ? L : 1);
    for (int i = 0; i < L; i++) sizes[i] = (int32_t)histo[i].size();
    orbfe_three_maxima(sizes.data(), L, &ind1, &ind2, &ind3); // :1597-1638
}

// src/ORBmatcher.cc:43-127 -> orbfe_search_by_projection_points.  Reads what Frame::isInFrustum left in each MapPoint.
int ORBmatcher::SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th)
{
    orbfe_context *ctx = context_of(F);
    const size_t n = vpMapPoints.size();
    std::vector<orbfe_track_point> pts(n ? n : 1);
    std::vector<uint8_t> desc(32 * (n ? n : 1), 0);
    std::vector<int32_t> obs(n ? n : 1, 0);
    for (size_t i = 0; i < n; i++) {
        MapPoint *pMP = vpMapPoints[i];
        orbfe_track_point &t = pts[i];
        std::memset(&t, 0, sizeof(t));
        if (!pMP || !pMP->mbTrackInView || pMP->isBad()) continue; // :52-56
        t.in_view = 1;
        t.proj_x = pMP->mTrackProjX; t.proj_y = pMP->mTrackProjY; t.proj_xr = pMP->mTrackProjXR;
        t.level = pMP->mnTrackScaleLevel; t.view_cos = pMP->mTrackViewCos;
        const cv::Mat d = pMP->GetDescriptor();
        std::memcpy(&desc[32 * i], d.ptr<uchar>(0), 32);
        obs[i] = pMP->Observations();
    }
    std::vector<uint8_t> has_obs(F.N > 0 ? F.N : 1, 0); // :85-87
    for (int k = 0; k < F.N; k++) has_obs[k] = F.mvpMapPoints[k] && F.mvpMapPoints[k]->Observations() > 0;
    const orbfe_frame_view v = device_view_of(F);
    std::vector<int32_t> match(F.N > 0 ? F.N : 1, -1);
    int nmatches = 0;
    check(ctx, orbfe_search_by_projection_points(ctx, &v, (int)n, pts.data(), desc.data(), obs.data(), has_obs.data(), th, mfNNratio, match.data(), &nmatches));
    for (int k = 0; k < F.N; k++)
        if (match[k] >= 0) F.mvpMapPoints[k] = vpMapPoints[match[k]]; // :121
    return nmatches;
}

// src/ORBmatcher.cc:1324-1466 -> orbfe_search_by_projection_last
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
{
    orbfe_context *ctx = context_of(CurrentFrame);
    const int nl = LastFrame.N;
    PointRows last((size_t)nl);
    std::vector<int32_t> octave(nl > 0 ? nl : 1, 0);
    std::vector<float> angle(nl > 0 ? nl : 1, 0.f);
    for (int i = 0; i < nl; i++) {
        MapPoint *pMP = LastFrame.mvpMapPoints[i];
        octave[i] = LastFrame.mvKeys[i].octave;   // :1376
        angle[i] = LastFrame.mvKeysUn[i].angle;   // :1436
        if (pMP && !LastFrame.mvbOutlier[i]) last.set((size_t)i, pMP, false, false); // :1357-1361
    }
    std::vector<uint8_t> has_obs(CurrentFrame.N > 0 ? CurrentFrame.N : 1, 0); // :1399-1401
    for (int k = 0; k < CurrentFrame.N; k++) has_obs[k] = CurrentFrame.mvpMapPoints[k] && CurrentFrame.mvpMapPoints[k]->Observations() > 0;
    float Tc[12], Tl[12];
    pose_3x4(CurrentFrame.mTcw, Tc);
    pose_3x4(LastFrame.mTcw, Tl);
    const orbfe_frame_view v = device_view_of(CurrentFrame);
    std::vector<int32_t> match(CurrentFrame.N > 0 ? CurrentFrame.N : 1, -1);
    int nmatches = 0;
    check(ctx, orbfe_search_by_projection_last(ctx, &v, Tc, Tl, nl, last.pos.data(), last.desc.data(), last.valid.data(), last.obs.data(),
                                                octave.data(), angle.data(), has_obs.data(), th, bMono ? 1 : 0, mbCheckOrientation ? 1 : 0,
                                                match.data(), &nmatches));
    // :1430 (assignment) and :1455-1459 (a match dropped by the orientation histogram leaves NULL).  Tracking clears
    // mvpMapPoints before this call (src/Tracking.cc:980), so "no match" == NULL.
    for (int k = 0; k < CurrentFrame.N; k++)
        if (match[k] >= 0) CurrentFrame.mvpMapPoints[k] = LastFrame.mvpMapPoints[match[k]];
    return nmatches;
}

// src/ORBmatcher.cc:1468-1595 -> orbfe_search_by_projection_kf
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th, const int ORBdist)
{
    orbfe_context *ctx = context_of(CurrentFrame);
    const std::vector<MapPoint *> vpMPs = pKF->GetMapPointMatches();
    const size_t n = vpMPs.size();
    PointRows kf(n);
    const std::vector<float> angle = angles_of(pKF->mvKeysUn); // :1565
    for (size_t i = 0; i < n; i++) {
        MapPoint *pMP = vpMPs[i];
        if (pMP && !pMP->isBad() && !sAlreadyFound.count(pMP)) kf.set(i, pMP, false, true); // :1486-1490
    }
    std::vector<uint8_t> has_pt(CurrentFrame.N > 0 ? CurrentFrame.N : 1, 0); // :1537-1538
    for (int k = 0; k < CurrentFrame.N; k++) has_pt[k] = CurrentFrame.mvpMapPoints[k] != NULL;
    float Tc[12];
    pose_3x4(CurrentFrame.mTcw, Tc);
    const orbfe_frame_view v = device_view_of(CurrentFrame);
    std::vector<int32_t> match(CurrentFrame.N > 0 ? CurrentFrame.N : 1, -1);
    int nmatches = 0;
    check(ctx, orbfe_search_by_projection_kf(ctx, &v, Tc, (int)n, kf.pos.data(), kf.desc.data(), kf.valid.data(), angle.data(), kf.maxd.data(),
                                              kf.mind.data(), has_pt.data(), th, ORBdist, mbCheckOrientation ? 1 : 0, match.data(), &nmatches));
    for (int k = 0; k < CurrentFrame.N; k++)
        if (match[k] >= 0) CurrentFrame.mvpMapPoints[k] = vpMPs[match[k]]; // :1559
    return nmatches;
}

// src/ORBmatcher.cc:285-398 -> orbfe_search_by_projection_sim3
int ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched, int th)
{
    orbfe_context *ctx = kf_context(pKF);
    std::set<MapPoint *> spAlreadyFound(vpMatched.begin(), vpMatched.end()); // :302-303
    spAlreadyFound.erase(static_cast<MapPoint *>(NULL));
    const size_t n = vpPoints.size();
    PointRows pts(n);
    for (size_t i = 0; i < n; i++) {
        MapPoint *pMP = vpPoints[i];
        if (pMP && !pMP->isBad() && !spAlreadyFound.count(pMP)) pts.set(i, pMP, true, true); // :312-314
    }
    std::vector<uint8_t> kf_matched(pKF->N > 0 ? pKF->N : 1, 0); // :369-370
    for (int k = 0; k < pKF->N && k < (int)vpMatched.size(); k++) kf_matched[k] = vpMatched[k] != NULL;
    float S[12];
    pose_3x4(Scw, S);
    const orbfe_frame_view v = view_of(pKF);
    std::vector<int32_t> pt_match(n ? n : 1, -1);
    int nmatches = 0;
    check(ctx, orbfe_search_by_projection_sim3(ctx, &v, S, (int)n, pts.pos.data(), pts.normal.data(), pts.maxd.data(), pts.mind.data(), pts.desc.data(),
                                                pts.valid.data(), kf_matched.data(), (float)th, pt_match.data(), &nmatches));
    for (size_t i = 0; i < n; i++)
        if (pt_match[i] >= 0) vpMatched[pt_match[i]] = vpPoints[i]; // :390
    return nmatches;
}

// src/ORBmatcher.cc:157-283 -> orbfe_search_by_bow
int ORBmatcher::SearchByFboW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches)
{
    orbfe_context *ctx = context_of(F);
    const std::vector<MapPoint *> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPoint *>(F.N, static_cast<MapPoint *>(NULL));
    const FeatCSR kf_fv(pKF->mFbowFeatVec), f_fv(F.mFbowFeatVec);
    const int nk = pKF->N;
    std::vector<int32_t> kf_valid(nk > 0 ? nk : 1, 0);
    for (int i = 0; i < nk && i < (int)vpMapPointsKF.size(); i++) kf_valid[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad(); // :190-196
    const std::vector<float> kf_angle = angles_of(pKF->mvKeysUn), f_angle = angles_of(F.mvKeys); // :237,241
    std::vector<int32_t> f_match(F.N > 0 ? F.N : 1, -1);
    int nmatches = 0;
    check(ctx, orbfe_search_by_bow(ctx, kf_fv.nodes.data(), kf_fv.off.data(), kf_fv.feat.data(), (int)kf_fv.nodes.size(), kf_valid.data(),
                                    pKF->mDescriptors.ptr<uchar>(0), kf_angle.data(), nk, f_fv.nodes.data(), f_fv.off.data(), f_fv.feat.data(),
                                    (int)f_fv.nodes.size(), F.mDescriptors.ptr<uchar>(0), f_angle.data(), F.N, mfNNratio, mbCheckOrientation ? 1 : 0,
                                    f_match.data(), &nmatches));
    for (int j = 0; j < F.N; j++)
        if (f_match[j] >= 0) vpMapPointMatches[j] = vpMapPointsKF[f_match[j]]; // :235
    return nmatches;
}

// src/ORBmatcher.cc:517-650 -> orbfe_search_by_bow_kf
int ORBmatcher::SearchByFboW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12)
{
    orbfe_context *ctx = kf_context(pKF1);
    const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = std::vector<MapPoint *>(vpMapPoints1.size(), static_cast<MapPoint *>(NULL));
    const FeatCSR fv1(pKF1->mFbowFeatVec), fv2(pKF2->mFbowFeatVec);
    const int n1 = (int)vpMapPoints1.size(), n2 = (int)vpMapPoints2.size();
    std::vector<int32_t> valid1(n1 > 0 ? n1 : 1, 0), valid2(n2 > 0 ? n2 : 1, 0);
    for (int i = 0; i < n1; i++) valid1[i] = vpMapPoints1[i] && !vpMapPoints1[i]->isBad(); // :551-555
    for (int i = 0; i < n2; i++) valid2[i] = vpMapPoints2[i] && !vpMapPoints2[i]->isBad(); // :571-575
    const std::vector<float> angle1 = angles_of(pKF1->mvKeysUn), angle2 = angles_of(pKF2->mvKeysUn);
    std::vector<int32_t> m12(n1 > 0 ? n1 : 1, -1);
    int nmatches = 0;
    check(ctx, orbfe_search_by_bow_kf(ctx, fv1.nodes.data(), fv1.off.data(), fv1.feat.data(), (int)fv1.nodes.size(), valid1.data(),
                                       pKF1->mDescriptors.ptr<uchar>(0), angle1.data(), n1, fv2.nodes.data(), fv2.off.data(), fv2.feat.data(),
                                       (int)fv2.nodes.size(), valid2.data(), pKF2->mDescriptors.ptr<uchar>(0), angle2.data(), n2, mfNNratio,
                                       mbCheckOrientation ? 1 : 0, m12.data(), &nmatches));
    for (int i = 0; i < n1; i++)
        if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]]; // :598
    return nmatches;
}

// src/ORBmatcher.cc:400-515 -> orbfe_search_for_initialization
int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize)
{
    orbfe_context *ctx = context_of(F2);
    vnMatches12 = std::vector<int>(F1.mvKeysUn.size(), -1); // :403
    if (vbPrevMatched.size() < F1.mvKeysUn.size()) throw std::invalid_argument("SearchForInitialization: vbPrevMatched is shorter than F1.mvKeysUn");
    const orbfe_frame_view v1 = view_of(F1), v2 = device_view_of(F2);
    std::vector<int32_t> m12(F1.mvKeysUn.size() ? F1.mvKeysUn.size() : 1, -1);
    int nmatches = 0;
    check(ctx, orbfe_search_for_initialization(ctx, &v1, &v2, reinterpret_cast<float *>(vbPrevMatched.data()), windowSize, mfNNratio,
                                                mbCheckOrientation ? 1 : 0, m12.data(), &nmatches));
    for (size_t i = 0; i < vnMatches12.size(); i++) vnMatches12[i] = m12[i];
    return nmatches;
}

// src/ORBmatcher.cc:652-819 -> orbfe_search_for_triangulation
int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo)
{
    orbfe_context *ctx = kf_context(pKF1);
    const FeatCSR fv1(pKF1->mFbowFeatVec), fv2(pKF2->mFbowFeatVec);
    const int n1 = pKF1->N, n2 = pKF2->N;
    std::vector<uint8_t> has1(n1 > 0 ? n1 : 1, 0), has2(n2 > 0 ? n2 : 1, 0);
    for (int i = 0; i < n1; i++) has1[i] = pKF1->GetMapPoint(i) != NULL; // :692-696
    for (int i = 0; i < n2; i++) has2[i] = pKF2->GetMapPoint(i) != NULL; // :716-720
    float F[9], Cw[3], T2w[12];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) F[3 * r + c] = F12.at<float>(r, c);
    const cv::Mat cw = pKF1->GetCameraCenter(); // :658
    for (int k = 0; k < 3; k++) Cw[k] = cw.at<float>(k);
    pose_from_Rt(pKF2->GetRotation(), pKF2->GetTranslation(), T2w); // :659-660
    std::vector<int32_t> m12(n1 > 0 ? n1 : 1, -1);
    int nmatches = 0;
    check(ctx, orbfe_search_for_triangulation(ctx, fv1.nodes.data(), fv1.off.data(), fv1.feat.data(), (int)fv1.nodes.size(), keys_of(pKF1->mvKeysUn),
                                               pKF1->mvuRight.data(), has1.data(), pKF1->mDescriptors.ptr<uchar>(0), n1,
                                               fv2.nodes.data(), fv2.off.data(), fv2.feat.data(), (int)fv2.nodes.size(), keys_of(pKF2->mvKeysUn),
                                               pKF2->mvuRight.data(), has2.data(), pKF2->mDescriptors.ptr<uchar>(0), n2,
                                               F, Cw, T2w, pKF2->fx, pKF2->fy, pKF2->cx, pKF2->cy, bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0,
                                               m12.data(), &nmatches));
    vMatchedPairs.clear(); // :806-816
    vMatchedPairs.reserve(nmatches);
    for (int i = 0; i < n1; i++)
        if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
    return nmatches;
}

// src/ORBmatcher.cc:1098-1322 -> orbfe_search_by_sim3
int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12,
                             const cv::Mat &t12, const float th)
{
    orbfe_context *ctx = kf_context(pKF1);
    const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false); // :1126-1139
    for (int i = 0; i < N1; i++) {
        MapPoint *pMP = vpMatches12[i];
        if (pMP) {
            vbAlreadyMatched1[i] = true;
            const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
    }
    PointRows p1((size_t)N1), p2((size_t)N2);
    for (int i = 0; i < N1; i++) {
        MapPoint *pMP = vpMapPoints1[i];
        if (pMP && !vbAlreadyMatched1[i] && !pMP->isBad()) p1.set((size_t)i, pMP, false, true); // :1147-1153
    }
    for (int i = 0; i < N2; i++) {
        MapPoint *pMP = vpMapPoints2[i];
        if (pMP && !vbAlreadyMatched2[i] && !pMP->isBad()) p2.set((size_t)i, pMP, false, true); // :1222-1228
    }
    float T1w[12], T2w[12], R[9], t[3];
    pose_from_Rt(pKF1->GetRotation(), pKF1->GetTranslation(), T1w);
    pose_from_Rt(pKF2->GetRotation(), pKF2->GetTranslation(), T2w);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) R[3 * r + c] = R12.at<float>(r, c);
        t[r] = t12.at<float>(r);
    }
    const orbfe_frame_view v1 = view_of(pKF1), v2 = view_of(pKF2);
    std::vector<int32_t> m12(N1 > 0 ? N1 : 1, -1);
    int nFound = 0;
    check(ctx, orbfe_search_by_sim3(ctx, &v1, T1w, p1.pos.data(), p1.maxd.data(), p1.mind.data(), p1.desc.data(), p1.valid.data(),
                                     &v2, T2w, p2.pos.data(), p2.maxd.data(), p2.mind.data(), p2.desc.data(), p2.valid.data(),
                                     s12, R, t, th, m12.data(), &nFound));
    for (int i1 = 0; i1 < N1; i1++)
        if (m12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[m12[i1]]; // :1305
    return nFound;
}

// src/ORBmatcher.cc:821-971: the search is orbfe_fuse, the map mutation (:943-964) stays here, literally
int ORBmatcher::Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th)
{
    orbfe_context *ctx = kf_context(pKF);
    const size_t n = vpMapPoints.size();
    PointRows pts(n);
    for (size_t i = 0; i < n; i++) {
        MapPoint *pMP = vpMapPoints[i];
        if (pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKF)) pts.set(i, pMP, true, true); // :843-847
    }
    float Tcw[12];
    pose_from_Rt(pKF->GetRotation(), pKF->GetTranslation(), Tcw);
    const orbfe_frame_view v = view_of(pKF);
    std::vector<int32_t> best(n ? n : 1, -1);
    int nSearch = 0;
    check(ctx, orbfe_fuse(ctx, &v, Tcw, (int)n, pts.pos.data(), pts.normal.data(), pts.maxd.data(), pts.mind.data(), pts.desc.data(), pts.valid.data(), th,
                           best.data(), &nSearch));
    int nFused = 0;
    for (size_t i = 0; i < n; i++) {
        const int bestIdx = best[i];
        if (bestIdx < 0) continue;
        MapPoint *pMP = vpMapPoints[i];
        // The reference searches and mutates point by point (:843-964); here every search ran before the first mutation.  The
        // search itself reads only the keyframe's keypoints, so the two orders differ exactly where an EARLIER mutation changes
        // this point's :843-847 filter: a point replaced meanwhile (it appears in vpMapPoints and was the keyframe's own point
        // at an earlier bestIdx) or already added to pKF (listed twice).  Re-applying the filter here restores the sequence.
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) {
                if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                else pMPinKF->Replace(pMP);
            }
        } else {
            pMP->AddObservation(pKF, bestIdx);
            pKF->AddMapPoint(pMP, bestIdx);
        }
        nFused++;
    }
    return nFused;
}

// src/ORBmatcher.cc:973-1096: search = orbfe_fuse_sim3, bookkeeping (:1073-1090) here
int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint)
{
    orbfe_context *ctx = kf_context(pKF);
    const std::set<MapPoint *> spAlreadyFound = pKF->GetMapPoints(); // :989
    const size_t n = vpPoints.size();
    PointRows pts(n);
    for (size_t i = 0; i < n; i++) {
        MapPoint *pMP = vpPoints[i];
        if (pMP && !pMP->isBad() && !spAlreadyFound.count(pMP)) pts.set(i, pMP, true, true); // :1000-1002
    }
    float S[12];
    pose_3x4(Scw, S);
    const orbfe_frame_view v = view_of(pKF);
    std::vector<int32_t> best(n ? n : 1, -1);
    int nSearch = 0;
    check(ctx, orbfe_fuse_sim3(ctx, &v, S, (int)n, pts.pos.data(), pts.normal.data(), pts.maxd.data(), pts.mind.data(), pts.desc.data(), pts.valid.data(), th,
                                best.data(), &nSearch));
    int nFused = 0;
    for (size_t i = 0; i < n; i++) {
        const int bestIdx = best[i];
        if (bestIdx < 0) continue;
        MapPoint *pMP = vpPoints[i];
        MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) vpReplacePoint[i] = pMPinKF;
        } else {
            pMP->AddObservation(pKF, bestIdx);
            pKF->AddMapPoint(pMP, bestIdx);
        }
        nFused++;
    }
    return nFused;
}

} // namespace ORB_SLAM2
