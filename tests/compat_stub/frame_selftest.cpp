// tests/compat_stub/frame_selftest.cpp -- TEST-ONLY driver: builds ORB_SLAM2::Frame objects through orbslam2_amd/compat/Frame.cc
// exactly the way src/Tracking.cc does (:296 stereo, :326 RGB-D, :354-358 monocular: the reference's constructor signatures,
// extractors new-ed as at :125-131) and dumps every member the constructors fill; tests/test_compat_frame.py compares the
// dumps with the CPU oracle.  Frame / MapPoint / cv::Mat / fbow are the declaration stand-ins of this directory.
//   usage: frame_selftest <dir> stereo|rgbd|mono|threads|backend
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "Frame.h"
#include "KeyFrame.h"
#include "KeyFrameDatabase.h" // orbslam2_amd/compat
#include "Optimizer.h"
#include "ORBmatcher.h"

using namespace ORB_SLAM2;
std::mutex MapPoint::mGlobalMutex;
Optimizer::Optimizer(const string &) {} // src/Optimizer.cc:40-86 reads the settings file; PoseOptimization uses none of it

static std::string g_dir;
template <class T> static std::vector<T> rd(const std::string &name)
{
    std::ifstream f(g_dir + "/" + name, std::ios::binary | std::ios::ate);
    if (!f) { std::cerr << "missing " << name << "\n"; std::exit(2); }
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<T> v((size_t)n / sizeof(T));
    f.read((char *)v.data(), n);
    return v;
}
template <class T> static void wr(const std::string &name, const std::vector<T> &v)
{
    std::ofstream f(g_dir + "/" + name, std::ios::binary);
    f.write((const char *)v.data(), (std::streamsize)(v.size() * sizeof(T)));
}
static void wr_mat(const std::string &name, const cv::Mat &m)
{
    std::vector<uchar> v;
    for (int r = 0; r < m.rows; r++) v.insert(v.end(), m.ptr<uchar>(r), m.ptr<uchar>(r) + (size_t)m.cols * (m.type() == CV_32F ? 4 : 1));
    wr(name, v);
}

// every member the reference's constructors fill (src/Frame.cc:61-229)
static void dump_frame(const std::string &tag, Frame &F)
{
    wr(tag + "_keys.bin", F.mvKeys); wr(tag + "_keys_right.bin", F.mvKeysRight); wr(tag + "_keys_un.bin", F.mvKeysUn);
    wr_mat(tag + "_desc.bin", F.mDescriptors); wr_mat(tag + "_desc_right.bin", F.mDescriptorsRight);
    wr(tag + "_uright.bin", F.mvuRight); wr(tag + "_depth.bin", F.mvDepth);
    std::vector<int32_t> off(1, 0), idx;
    for (int i = 0; i < FRAME_GRID_COLS; i++)
        for (int j = 0; j < FRAME_GRID_ROWS; j++) {
            for (size_t k = 0; k < F.mGrid[i][j].size(); k++) idx.push_back((int32_t)F.mGrid[i][j][k]);
            off.push_back((int32_t)idx.size());
        }
    wr(tag + "_grid_off.bin", off); wr(tag + "_grid_idx.bin", idx);
    std::vector<float> st = {Frame::fx, Frame::fy, Frame::cx, Frame::cy, Frame::invfx, Frame::invfy, Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY,
                             Frame::mnMaxY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv, F.mbf, F.mb, F.mThDepth, F.mfScaleFactor,
                             F.mfLogScaleFactor};
    st.insert(st.end(), F.mvScaleFactors.begin(), F.mvScaleFactors.end());
    st.insert(st.end(), F.mvInvLevelSigma2.begin(), F.mvInvLevelSigma2.end());
    wr(tag + "_statics.bin", st);
    std::vector<int32_t> meta = {F.N, (int32_t)F.mnId, F.mnScaleLevels, (int32_t)F.mvpMapPoints.size(), (int32_t)F.mvbOutlier.size(), (int32_t)Frame::mbInitialComputations};
    wr(tag + "_meta.bin", meta);
}

static void dump_bow(const std::string &tag, Frame &F)
{
    std::vector<uint32_t> words, nodes;
    std::vector<float> ww;
    std::vector<int32_t> off(1, 0), feat;
    for (fbow::fBow::const_iterator it = F.mFbowVec.begin(); it != F.mFbowVec.end(); ++it) { words.push_back(it->first); ww.push_back((float)it->second); }
    for (fbow::fBow2::const_iterator it = F.mFbowFeatVec.begin(); it != F.mFbowFeatVec.end(); ++it) {
        nodes.push_back(it->first);
        for (size_t k = 0; k < it->second.size(); k++) feat.push_back((int32_t)it->second[k]);
        off.push_back((int32_t)feat.size());
    }
    wr(tag + "_bow_words.bin", words); wr(tag + "_bow_w.bin", ww); wr(tag + "_bow_nodes.bin", nodes); wr(tag + "_bow_off.bin", off); wr(tag + "_bow_feat.bin", feat);
}

static bool same_frame(const Frame &a, const Frame &b)
{
    if (a.N != b.N || a.mvKeys.size() != b.mvKeys.size() || a.mvuRight != b.mvuRight || a.mvDepth != b.mvDepth) return false;
    if (a.N && std::memcmp(a.mvKeys.data(), b.mvKeys.data(), a.mvKeys.size() * sizeof(cv::KeyPoint))) return false;
    if (a.N && std::memcmp(a.mvKeysUn.data(), b.mvKeysUn.data(), a.mvKeysUn.size() * sizeof(cv::KeyPoint))) return false;
    if (a.N && std::memcmp(a.mDescriptors.ptr<uchar>(0), b.mDescriptors.ptr<uchar>(0), (size_t)a.N * 32)) return false;
    for (int i = 0; i < FRAME_GRID_COLS; i++)
        for (int j = 0; j < FRAME_GRID_ROWS; j++)
            if (a.mGrid[i][j] != b.mGrid[i][j]) return false;
    return true;
}

int main(int argc, char **argv)
{
    if (argc != 3) { std::cerr << "usage: frame_selftest <dir> stereo|rgbd|mono|threads\n"; return 2; }
    g_dir = argv[1];
    const std::string mode = argv[2];
    try {
        const std::vector<float> cam = rd<float>("cam.f32"); // fx fy cx cy bf W H nfeatures thDepth
        const std::vector<float> dist = rd<float>("dist.f32");
        const int W = (int)cam[5], H = (int)cam[6], nFeatures = (int)cam[7];
        // src/Tracking.cc:52-78: mK, mDistCoef (4 x 1, or 5 x 1 with k3)
        cv::Mat mK(3, 3, CV_32F);
        mK.at<float>(0, 0) = cam[0]; mK.at<float>(1, 1) = cam[1]; mK.at<float>(0, 2) = cam[2]; mK.at<float>(1, 2) = cam[3]; mK.at<float>(2, 2) = 1.f;
        cv::Mat mDistCoef((int)dist.size() > 4 ? 5 : 4, 1, CV_32F);
        for (size_t i = 0; i < dist.size(); i++) mDistCoef.at<float>((int)i) = dist[i];
        const float mbf = cam[4], mThDepth = cam[8];
        fbow::Vocabulary voc; // src/System.cc:71-72
        voc.readFromFile(g_dir + "/vocab.fbow");
        fbow::Vocabulary *mpFBOWVocabulary = &voc;
        // src/Tracking.cc:125-131
        ORBextractor *mpORBextractorLeft = new ORBextractor(nFeatures, 1.2f, 8, 20, 7, 31, 15, 19);
        ORBextractor *mpORBextractorRight = new ORBextractor(nFeatures, 1.2f, 8, 20, 7, 31, 15, 19);
        ORBextractor *mpIniORBextractor = new ORBextractor(2 * nFeatures, 1.2f, 8, 20, 7, 31, 15, 19);

        if (mode == "stereo") {
            for (int t = 0; t < 2; t++) { // frame 0 runs the "initial computations" (:95-114), frame 1 does not
                std::vector<uchar> l = rd<uchar>("left" + std::to_string(t) + ".bin"), r = rd<uchar>("right" + std::to_string(t) + ".bin");
                cv::Mat mImGray(H, W, CV_8UC1, l.data()), imGrayRight(H, W, CV_8UC1, r.data());
                Frame mCurrentFrame = Frame(mImGray, imGrayRight, 0.1 * t, mpORBextractorLeft, mpORBextractorRight, mpFBOWVocabulary, mK, mDistCoef, mbf, mThDepth); // src/Tracking.cc:296
                dump_frame("f" + std::to_string(t), mCurrentFrame);
                mCurrentFrame.ComputeFboW(); // src/Tracking.cc:860 (TrackReferenceKeyFrame)
                dump_bow("f" + std::to_string(t), mCurrentFrame);
                Frame copy(mCurrentFrame); // mLastFrame = Frame(mCurrentFrame), src/Tracking.cc:560
                if (!same_frame(copy, mCurrentFrame)) { std::cerr << "copy constructor lost members\n"; return 3; }
                const std::vector<float> ur = mCurrentFrame.mvuRight, dp = mCurrentFrame.mvDepth;
                mCurrentFrame.ComputeStereoMatches(); // the member by itself: re-read from the resident frame
                if (ur != mCurrentFrame.mvuRight || dp != mCurrentFrame.mvDepth) { std::cerr << "ComputeStereoMatches() differs from the constructor's result\n"; return 4; }
                // Frame::GetFeaturesInArea on the resident frame and PosInGrid against the grid the device built
                const cv::KeyPoint &kp = mCurrentFrame.mvKeysUn[mCurrentFrame.N / 2];
                const std::vector<size_t> area = mCurrentFrame.GetFeaturesInArea(kp.pt.x, kp.pt.y, 25.f, 0, 3);
                wr("f" + std::to_string(t) + "_area.bin", std::vector<int32_t>(area.begin(), area.end()));
                wr("f" + std::to_string(t) + "_area_q.bin", std::vector<float>{kp.pt.x, kp.pt.y, 25.f});
                for (int i = 0; i < mCurrentFrame.N; i += 37) {
                    int px, py;
                    if (!mCurrentFrame.PosInGrid(mCurrentFrame.mvKeysUn[i], px, py)) continue;
                    const std::vector<size_t> &cell = mCurrentFrame.mGrid[px][py];
                    bool in = false;
                    for (size_t k = 0; k < cell.size(); k++) in |= cell[k] == (size_t)i;
                    if (!in) { std::cerr << "PosInGrid disagrees with mGrid for keypoint " << i << "\n"; return 5; }
                }
                // isInFrustum + UnprojectStereo: a point unprojected from the frame projects back into it
                mCurrentFrame.SetPose(([&] { cv::Mat T(4, 4, CV_32F); for (int i = 0; i < 4; i++) T.at<float>(i, i) = 1.f; T.at<float>(0, 3) = 0.05f; return T; })());
                std::vector<float> frustum;
                for (int i = 0; i < mCurrentFrame.N; i += 11) {
                    cv::Mat x3D = mCurrentFrame.UnprojectStereo(i);
                    if (x3D.empty()) continue;
                    MapPoint mp;
                    mp.mWorldPos = x3D;
                    cv::Mat nrm(3, 1, CV_32F);
                    const cv::Mat Ow = mCurrentFrame.GetCameraCenter();
                    float len = 0.f;
                    for (int k = 0; k < 3; k++) { nrm.at<float>(k) = x3D.at<float>(k) - Ow.at<float>(k); len += nrm.at<float>(k) * nrm.at<float>(k); }
                    len = std::sqrt(len);
                    for (int k = 0; k < 3; k++) nrm.at<float>(k) /= len;
                    mp.mNormalVector = nrm;
                    mp.mfMaxDistance = len * mCurrentFrame.mvScaleFactors[mCurrentFrame.mvKeysUn[i].octave] * 1.1f; // off PredictScale's ceil() boundary
                    mp.mfMinDistance = mp.mfMaxDistance / mCurrentFrame.mvScaleFactors[mCurrentFrame.mnScaleLevels - 1];
                    const bool in = mCurrentFrame.isInFrustum(&mp, 0.5f);
                    const float rec[14] = {(float)i, x3D.at<float>(0), x3D.at<float>(1), x3D.at<float>(2), nrm.at<float>(0), nrm.at<float>(1), nrm.at<float>(2),
                                           mp.mfMaxDistance, mp.mfMinDistance, in ? 1.f : 0.f, mp.mTrackProjX, mp.mTrackProjY, mp.mTrackProjXR, (float)mp.mnTrackScaleLevel};
                    frustum.insert(frustum.end(), rec, rec + 14);
                }
                wr("f" + std::to_string(t) + "_frustum.bin", frustum);
                // and the frame is what the Tracking matchers then search in: SearchByProjection(F, points) on the resident frame
                if (t == 1) {
                    std::vector<MapPoint> own((size_t)(mCurrentFrame.N + 1) / 2);
                    std::vector<MapPoint *> pts;
                    for (int i = 0; i < mCurrentFrame.N; i += 2) {
                        MapPoint &p = own[i / 2];
                        p.mDescriptor = cv::Mat(1, 32, CV_8U);
                        std::memcpy(p.mDescriptor.ptr<uchar>(0), mCurrentFrame.mDescriptors.ptr<uchar>(i), 32);
                        p.nObs = 1; p.mbTrackInView = true;
                        p.mTrackProjX = mCurrentFrame.mvKeysUn[i].pt.x + 1.5f; p.mTrackProjY = mCurrentFrame.mvKeysUn[i].pt.y - 1.0f; p.mTrackProjXR = -1.f;
                        p.mnTrackScaleLevel = mCurrentFrame.mvKeysUn[i].octave; p.mTrackViewCos = 0.9f;
                        pts.push_back(&p);
                    }
                    ORBmatcher matcher(0.8);
                    const int nm = matcher.SearchByProjection(mCurrentFrame, pts, 3); // src/Tracking.cc:1290
                    std::vector<int32_t> out(mCurrentFrame.N, -1);
                    for (int k = 0; k < mCurrentFrame.N; k++)
                        if (mCurrentFrame.mvpMapPoints[k]) out[k] = 2 * (int32_t)(mCurrentFrame.mvpMapPoints[k] - own.data());
                    wr("f1_matches.bin", out);
                    wr("f1_nmatches.bin", std::vector<int32_t>(1, nm));
                }
            }
            // empty image: N = 0 and nothing else touched (src/ORBextractor.cc:861-862, src/Frame.cc:83-86)
            cv::Mat none;
            Frame e = Frame(none, none, 0.3, mpORBextractorLeft, mpORBextractorRight, mpFBOWVocabulary, mK, mDistCoef, mbf, mThDepth);
            if (e.N != 0 || !e.mvKeys.empty() || e.mnId != 2) { std::cerr << "empty stereo frame: N = " << e.N << ", id " << e.mnId << "\n"; return 6; }
        } else if (mode == "rgbd") {
            for (int t = 0; t < 2; t++) {
                std::vector<uchar> g = rd<uchar>("left" + std::to_string(t) + ".bin");
                std::vector<float> d = rd<float>("depth" + std::to_string(t) + ".f32");
                cv::Mat mImGray(H, W, CV_8UC1, g.data()), imDepth(H, W, CV_32F, d.data());
                Frame mCurrentFrame = Frame(mImGray, imDepth, 0.1 * t, mpORBextractorLeft, mpFBOWVocabulary, mK, mDistCoef, mbf, mThDepth); // src/Tracking.cc:326
                dump_frame("f" + std::to_string(t), mCurrentFrame);
                const std::vector<float> ur = mCurrentFrame.mvuRight, dp = mCurrentFrame.mvDepth;
                mCurrentFrame.ComputeStereoFromRGBD(imDepth);
                if (ur != mCurrentFrame.mvuRight || dp != mCurrentFrame.mvDepth) { std::cerr << "ComputeStereoFromRGBD() differs from the constructor's result\n"; return 4; }
            }
        } else if (mode == "mono") {
            // src/Tracking.cc:353-358: the initialisation extractor (2 x nFeatures) until the map exists, then the left one
            std::vector<uchar> g0 = rd<uchar>("left0.bin"), g1 = rd<uchar>("left1.bin");
            cv::Mat im0(H, W, CV_8UC1, g0.data()), im1(H, W, CV_8UC1, g1.data());
            Frame ini = Frame(im0, 0.0, mpIniORBextractor, mpFBOWVocabulary, mK, mDistCoef, mbf, mThDepth);
            dump_frame("f0", ini);
            Frame cur = Frame(im1, 0.1, mpORBextractorLeft, mpFBOWVocabulary, mK, mDistCoef, mbf, mThDepth);
            dump_frame("f1", cur);
            std::vector<cv::Point2f> prev(ini.mvKeysUn.size());
            for (size_t i = 0; i < prev.size(); i++) prev[i] = ini.mvKeysUn[i].pt;
            Frame ini2 = Frame(im1, 0.2, mpIniORBextractor, mpFBOWVocabulary, mK, mDistCoef, mbf, mThDepth);
            dump_frame("f2", ini2);
            std::vector<int> vnMatches12;
            ORBmatcher matcher(0.9, true);
            const int nm = matcher.SearchForInitialization(ini, ini2, prev, vnMatches12, 100); // src/Tracking.cc:698-699
            wr("init_matches.bin", std::vector<int32_t>(vnMatches12.begin(), vnMatches12.end()));
            wr("init_n.bin", std::vector<int32_t>(1, nm));
        } else if (mode == "threads") {
            // The reference's threading contract (src/Frame.cc:78-81): (*mpORBextractorLeft)(imLeft, ...) and (*mpORBextractorRight)(imRight, ...)
            // on two fresh std::threads, every frame.  200 frames, each compared with the serial result of the same two objects.
            std::vector<uchar> l = rd<uchar>("left0.bin"), r = rd<uchar>("right0.bin"), l1 = rd<uchar>("left1.bin"), r1 = rd<uchar>("right1.bin");
            cv::Mat imL[2] = {cv::Mat(H, W, CV_8UC1, l.data()), cv::Mat(H, W, CV_8UC1, l1.data())};
            cv::Mat imR[2] = {cv::Mat(H, W, CV_8UC1, r.data()), cv::Mat(H, W, CV_8UC1, r1.data())};
            struct Out { std::vector<cv::KeyPoint> k; cv::Mat d; };
            Out serial[2][2]; // the serial answer comes from two OTHER extractor objects, afterwards: the threaded loop below also
                              // creates the two device contexts concurrently, on its first frame
            int bad = 0;
            struct Got { std::vector<cv::KeyPoint> kl, kr; cv::Mat dl, dr; };
            std::vector<Got> got(200);
            for (int it = 0; it < 200; it++) {
                const int t = it & 1;
                Frame F; // ExtractORB is the member the reference's threads run
                F.mpORBextractorLeft = mpORBextractorLeft; F.mpORBextractorRight = mpORBextractorRight;
                std::thread threadLeft(&Frame::ExtractORB, &F, 0, imL[t]);
                std::thread threadRight(&Frame::ExtractORB, &F, 1, imR[t]);
                threadLeft.join();
                threadRight.join();
                got[it].kl = F.mvKeys; got[it].kr = F.mvKeysRight; got[it].dl = F.mDescriptors.clone(); got[it].dr = F.mDescriptorsRight.clone();
            }
            ORBextractor refL(nFeatures, 1.2f, 8, 20, 7, 31, 15, 19), refR(nFeatures, 1.2f, 8, 20, 7, 31, 15, 19);
            for (int t = 0; t < 2; t++) {
                refL(imL[t], cv::Mat(), serial[t][0].k, serial[t][0].d);
                refR(imR[t], cv::Mat(), serial[t][1].k, serial[t][1].d);
                serial[t][0].d = serial[t][0].d.clone(); serial[t][1].d = serial[t][1].d.clone();
            }
            wr("thr_keys_l.bin", serial[0][0].k); wr("thr_keys_r.bin", serial[0][1].k);
            wr_mat("thr_desc_l.bin", serial[0][0].d); wr_mat("thr_desc_r.bin", serial[0][1].d);
            for (int it = 0; it < 200; it++) {
                const int t = it & 1;
                const Got &g = got[it];
                const bool ok = g.kl.size() == serial[t][0].k.size() && g.kr.size() == serial[t][1].k.size() &&
                                !std::memcmp(g.kl.data(), serial[t][0].k.data(), g.kl.size() * sizeof(cv::KeyPoint)) &&
                                !std::memcmp(g.kr.data(), serial[t][1].k.data(), g.kr.size() * sizeof(cv::KeyPoint)) &&
                                !std::memcmp(g.dl.ptr<uchar>(0), serial[t][0].d.ptr<uchar>(0), g.kl.size() * 32) &&
                                !std::memcmp(g.dr.ptr<uchar>(0), serial[t][1].d.ptr<uchar>(0), g.kr.size() * 32);
                bad += !ok;
            }
            wr("thr_bad.bin", std::vector<int32_t>{bad, 200});
            if (bad) { std::cerr << bad << " of 200 threaded frames differ from the serial result\n"; return 7; }
        } else if (mode == "backend") {
            // The two section-8(f) call sites that take the reference's objects: Optimizer::PoseOptimization(&mCurrentFrame)
            // (src/Tracking.cc:875,998,1040) and mpKeyFrameDB->add / erase / DetectRelocalizationCandidates(&mCurrentFrame) /
            // DetectLoopCandidates(mpCurrentKF, minScore) (src/Tracking.cc:1496, src/LoopClosing.cc:131).
            mpORBextractorLeft->SetCamera(([&] { CameraParams c; c.fx = cam[0]; c.fy = cam[1]; c.cx = cam[2]; c.cy = cam[3]; c.bf = mbf; return c; })());
            mpORBextractorLeft->BindImageSize(W, H);
            Frame::fx = cam[0]; Frame::fy = cam[1]; Frame::cx = cam[2]; Frame::cy = cam[3]; Frame::invfx = 1.f / cam[0]; Frame::invfy = 1.f / cam[1];
            { // ---- PoseOptimization
                const std::vector<cv::KeyPoint> keys = rd<cv::KeyPoint>("pose_keys.bin");
                const std::vector<float> ur = rd<float>("pose_ur.bin"), Xw = rd<float>("pose_xw.bin"), T0 = rd<float>("pose_T0.bin");
                const std::vector<uchar> has = rd<uchar>("pose_has.bin");
                const int n = (int)keys.size();
                Frame F;
                F.mpORBextractorLeft = mpORBextractorLeft; F.mpORBextractorRight = NULL;
                F.N = n; F.mvKeysUn = keys; F.mvKeys = keys; F.mvuRight = ur; F.mbf = mbf; F.mb = mbf / cam[0];
                std::vector<MapPoint> pts(n);
                F.mvpMapPoints.assign(n, static_cast<MapPoint *>(NULL));
                F.mvbOutlier.assign(n, false);
                for (int i = 0; i < n; i++)
                    if (has[i]) { pts[i].mWorldPos = cv::Mat(3, 1, CV_32F); for (int k = 0; k < 3; k++) pts[i].mWorldPos.at<float>(k) = Xw[3 * i + k]; F.mvpMapPoints[i] = &pts[i]; }
                cv::Mat T(4, 4, CV_32F);
                for (int i = 0; i < 16; i++) T.ptr<float>(0)[i] = T0[i];
                F.SetPose(T);
                Optimizer *mpOptimizer = new Optimizer(std::string()); // src/Tracking.cc holds one, built from the settings file
                const int nInl = mpOptimizer->PoseOptimization(&F); // src/Tracking.cc:875
                std::vector<float> Tout(16);
                for (int i = 0; i < 16; i++) Tout[i] = F.mTcw.at<float>(i / 4, i % 4);
                std::vector<uchar> outl(n);
                for (int i = 0; i < n; i++) outl[i] = F.mvbOutlier[i];
                wr("pose_T.bin", Tout); wr("pose_outlier.bin", outl); wr("pose_ninl.bin", std::vector<int32_t>(1, nInl));
                const cv::Mat Ow = F.GetCameraCenter(); // SetPose ran: the camera centre follows the new pose
                wr("pose_Ow.bin", std::vector<float>{Ow.at<float>(0), Ow.at<float>(1), Ow.at<float>(2)});
                // fewer than 3 correspondences: returns 0, pose and flags untouched (src/Optimizer.cc:404-405)
                Frame G(F);
                G.mvpMapPoints.assign(n, static_cast<MapPoint *>(NULL));
                int kept = 0;
                for (int i = 0; i < n && kept < 2; i++) if (has[i]) { G.mvpMapPoints[i] = &pts[i]; kept++; }
                G.SetPose(T);
                if (mpOptimizer->PoseOptimization(&G) != 0 || std::memcmp(G.mTcw.ptr<float>(0), T.ptr<float>(0), 64)) { std::cerr << "PoseOptimization with 2 points touched the pose\n"; return 8; }
            }
            { // ---- KeyFrameDatabase
                const std::vector<int32_t> kf_off = rd<int32_t>("db_kf_off.bin"), cv_off = rd<int32_t>("db_covis_off.bin"), cv_idx = rd<int32_t>("db_covis_idx.bin");
                const std::vector<uint32_t> kf_words = rd<uint32_t>("db_kf_words.bin");
                const std::vector<float> kf_w = rd<float>("db_kf_w.bin");
                const std::vector<int32_t> q_off = rd<int32_t>("db_q_off.bin"), erase_after = rd<int32_t>("db_erase_after.bin");
                const std::vector<uint32_t> q_words = rd<uint32_t>("db_q_words.bin");
                const std::vector<float> q_w = rd<float>("db_q_w.bin"), min_score = rd<float>("db_min_score.bin");
                const std::vector<uchar> connected = rd<uchar>("db_connected.bin");
                const int nkf = (int)kf_off.size() - 1, nq = (int)q_off.size() - 1;
                Frame proto; // a keyframe is made from a frame (src/KeyFrame.cc:29); the database only reads its BoW vector and graph
                proto.mpORBextractorLeft = mpORBextractorLeft; proto.N = 0; proto.mbf = mbf; proto.mb = mbf / cam[0]; proto.mThDepth = mThDepth;
                std::vector<std::unique_ptr<KeyFrame> > kfs;
                for (int k = 0; k < nkf; k++) {
                    kfs.emplace_back(new KeyFrame(proto));
                    kfs[k]->mnId = 100 + k;
                    for (int j = kf_off[k]; j < kf_off[k + 1]; j++) { float w = kf_w[j]; kfs[k]->mFbowVec[kf_words[j]] = w; }
                }
                for (int k = 0; k < nkf; k++)
                    for (int j = cv_off[k]; j < cv_off[k + 1]; j++) kfs[k]->mvpOrderedConnectedKeyFrames.push_back(kfs[cv_idx[j]].get());
                KeyFrameDatabase *mpKeyFrameDB = new KeyFrameDatabase(mpFBOWVocabulary); // src/System.cc:88
                for (int k = 0; k < nkf; k++) mpKeyFrameDB->add(kfs[k].get());
                std::vector<int32_t> reloc_out, loop_out, counts;
                std::vector<float> state;
                for (int q = 0; q < nq; q++) {
                    Frame F;
                    F.mpORBextractorLeft = mpORBextractorLeft; F.N = 0; F.mnId = 1000 + q;
                    for (int j = q_off[q]; j < q_off[q + 1]; j++) { float w = q_w[j]; F.mFbowVec[q_words[j]] = w; }
                    const std::vector<KeyFrame *> vpCandidateKFs = mpKeyFrameDB->DetectRelocalizationCandidates(&F); // src/Tracking.cc:1496
                    for (size_t i = 0; i < vpCandidateKFs.size(); i++) reloc_out.push_back((int32_t)vpCandidateKFs[i]->mnId - 100);
                    for (int k = 0; k < nkf; k++) state.push_back(kfs[k]->mRelocScore);
                    KeyFrame cur(proto); // the current keyframe is not in the database yet (src/LoopClosing.cc:131 runs before :143's add)
                    cur.mnId = 5000 + q; cur.mFbowVec = F.mFbowVec;
                    for (int k = 0; k < nkf; k++) if (connected[(size_t)q * nkf + k]) cur.mConnected.push_back(kfs[k].get());
                    const std::vector<KeyFrame *> vpLoop = mpKeyFrameDB->DetectLoopCandidates(&cur, min_score[q]);
                    for (size_t i = 0; i < vpLoop.size(); i++) loop_out.push_back((int32_t)vpLoop[i]->mnId - 100);
                    counts.push_back((int32_t)vpCandidateKFs.size()); counts.push_back((int32_t)vpLoop.size());
                    if (erase_after[q] >= 0) mpKeyFrameDB->erase(kfs[erase_after[q]].get()); // KeyFrame::SetBadFlag, src/KeyFrame.cc:520
                }
                wr("db_reloc.bin", reloc_out); wr("db_loop.bin", loop_out); wr("db_counts.bin", counts); wr("db_state.bin", state);
                mpKeyFrameDB->clear(); // src/Tracking.cc:1607 (Reset)
                Frame F; F.mpORBextractorLeft = mpORBextractorLeft; F.N = 0; F.mnId = 1;
                { float w = 1.f; F.mFbowVec[q_words[0]] = w; }
                if (!mpKeyFrameDB->DetectRelocalizationCandidates(&F).empty()) { std::cerr << "cleared database returned candidates\n"; return 9; }
            }
        } else { std::cerr << "unknown mode " << mode << "\n"; return 2; }
        std::printf("frame selftest ok (%s)\n", mode.c_str());
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
