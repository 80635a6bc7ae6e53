// compat/Frame.cc -- the reference's Frame (include/Frame.h) DEFINED over the C ABI: replaces src/Frame.cc in a build of
// fabrizioromanelli/ORBSLAM2.  Compile with -I<reference>/include so that "Frame.h" is the
// reference's own header; this repository, which has neither OpenCV nor the reference tree, compiles it against the
// declaration stand-ins of tests/compat_stub (same member names, types and signatures) and runs it on the GPU box
// (tests/test_compat_frame.py).
//
// What changes against src/Frame.cc:
//   * the stereo constructor (src/Frame.cc:61-117) makes ONE device call, orbfe_stereo_frame -- both extractions AND
//     ComputeStereoMatches (:464-642); no second thread (:78-81), no host pyramid;
//   * the RGB-D constructor (:120-172) makes one call, orbfe_rgbd_frame (extraction + ComputeStereoFromRGBD :645-666), plus
//     the device-side UndistortKeyPoints (:402-432) when the camera is distorted;
//   * AssignFeaturesToGrid (:231-246), ComputeFboW (:395-400), ComputeImageBounds (:434-462), isInFrustum (:270-326) and
//     GetFeaturesInArea (:328-381) forward to their C-ABI restatements; mGrid is still filled (KeyFrame copies it,
//     src/KeyFrame.cc:45-50);
//   * the camera of the extractor's device context is taken from K / distCoef / bf of the FIRST constructor call (the
//     reference fixes Frame::fx ... the same way, :104-114): nothing to configure in Tracking.
// No arithmetic of the path lives here; the only float expressions are the pose bookkeeping of SetPose / UnprojectStereo
// (cv::Mat algebra in the reference) and PosInGrid's two lines.
#include "Frame.h"

#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/orbfe.h"
#include "compat_util.h"

namespace ORB_SLAM2
{
using namespace orbfe_compat;

long unsigned int Frame::nNextId = 0;
bool Frame::mbInitialComputations = true;
float Frame::cx, Frame::cy, Frame::fx, Frame::fy, Frame::invfx, Frame::invfy;
float Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY;
float Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv;

namespace
{

std::vector<float> dist_coefs(const cv::Mat &distCoef)
{
    std::vector<float> d;
    const int n = (int)distCoef.total(); // 4 x 1, or 5 x 1 with k3 (src/Tracking.cc:67-78)
    for (int i = 0; i < n && i < 5; i++) d.push_back(distCoef.at<float>(i));
    return d;
}

// The extractor's device context: created by the first frame with the frame's own camera, and checked against every later
// frame's (one camera model per extractor, as Frame's statics assume: src/Frame.cc:29-33).
orbfe_context *bind_context(ORBextractor *ex, const cv::Mat &im, const cv::Mat &K, const cv::Mat &distCoef, float bf)
{
    if (!ex) throw std::invalid_argument("Frame: null ORBextractor");
    const float kfx = K.at<float>(0, 0), kfy = K.at<float>(1, 1), kcx = K.at<float>(0, 2), kcy = K.at<float>(1, 2);
    if (!ex->Context()) {
        CameraParams cam = ex->Camera(); // keeps a colour order chosen by the integrator
        cam.fx = kfx; cam.fy = kfy; cam.cx = kcx; cam.cy = kcy; cam.bf = bf;
        cam.distCoef = dist_coefs(distCoef);
        ex->SetCamera(cam);
        ex->BindImageSize(im.cols, im.rows, 2);
    } else {
        float c[5];
        check(ex->Context(), orbfe_get_camera(ex->Context(), c));
        if (c[0] != kfx || c[1] != kfy || c[2] != kcx || c[3] != kcy || c[4] != bf)
            throw std::invalid_argument("Frame: K / bf differ from the camera this ORBextractor's device context was created for");
        ex->EnsureContext(im.cols, im.rows, 1); // throws on another image size
    }
    return ex->Context();
}

void require_gray(const cv::Mat &im)
{
    if (im.type() != CV_8UC1) throw std::invalid_argument("Frame: image must be CV_8UC1"); // assert at src/ORBextractor.cc:865
}

// hands the tree the reference loaded (src/System.cc:71-72) to the context once: Vocabulary::toStream (fbow.h:95) writes the
// file format orbfe_vocab_load reads
void ensure_vocabulary(orbfe_context *ctx, const fbow::Vocabulary *voc)
{
    static std::mutex mu;
    static std::map<orbfe_context *, const fbow::Vocabulary *> loaded;
    std::lock_guard<std::mutex> lk(mu);
    std::map<orbfe_context *, const fbow::Vocabulary *>::iterator it = loaded.find(ctx);
    if (it != loaded.end() && it->second == voc && orbfe_vocab_bytes(ctx) > 0) return; // (a NEW context at a recycled address holds none)
    std::ostringstream os(std::ios::binary);
    voc->toStream(os);
    const std::string blob = os.str();
    check(ctx, orbfe_vocab_load(ctx, reinterpret_cast<const uint8_t *>(blob.data()), blob.size()));
    loaded[ctx] = voc;
}

} // namespace

Frame::Frame() {}

// src/Frame.cc:38-59
Frame::Frame(const Frame &frame)
    : mpFBOWvocabulary(frame.mpFBOWvocabulary), mpORBextractorLeft(frame.mpORBextractorLeft), mpORBextractorRight(frame.mpORBextractorRight),
      mTimeStamp(frame.mTimeStamp), mK(frame.mK.clone()), mDistCoef(frame.mDistCoef.clone()), mbf(frame.mbf), mb(frame.mb),
      mThDepth(frame.mThDepth), N(frame.N), mvKeys(frame.mvKeys), mvKeysRight(frame.mvKeysRight), mvKeysUn(frame.mvKeysUn),
      mvuRight(frame.mvuRight), mvDepth(frame.mvDepth), mFbowVec(frame.mFbowVec), mFbowFeatVec(frame.mFbowFeatVec),
      mDescriptors(frame.mDescriptors.clone()), mDescriptorsRight(frame.mDescriptorsRight.clone()), mvpMapPoints(frame.mvpMapPoints),
      mvbOutlier(frame.mvbOutlier), mnId(frame.mnId), mpReferenceKF(frame.mpReferenceKF), mnScaleLevels(frame.mnScaleLevels),
      mfScaleFactor(frame.mfScaleFactor), mfLogScaleFactor(frame.mfLogScaleFactor), mvScaleFactors(frame.mvScaleFactors),
      mvInvScaleFactors(frame.mvInvScaleFactors), mvLevelSigma2(frame.mvLevelSigma2), mvInvLevelSigma2(frame.mvInvLevelSigma2)
{
    for (int i = 0; i < FRAME_GRID_COLS; i++)
        for (int j = 0; j < FRAME_GRID_ROWS; j++) mGrid[i][j] = frame.mGrid[i][j];
    if (!frame.mTcw.empty()) SetPose(frame.mTcw);
}

// the part the three constructors share before the extraction (src/Frame.cc:66-75) ...
#define ORBFE_FRAME_SCALE_INFO()                                              \
    mnId = nNextId++;                                                         \
    mnScaleLevels = mpORBextractorLeft->GetLevels();                          \
    mfScaleFactor = mpORBextractorLeft->GetScaleFactor();                     \
    mfLogScaleFactor = log(mfScaleFactor);                                    \
    mvScaleFactors = mpORBextractorLeft->GetScaleFactors();                   \
    mvInvScaleFactors = mpORBextractorLeft->GetInverseScaleFactors();         \
    mvLevelSigma2 = mpORBextractorLeft->GetScaleSigmaSquares();               \
    mvInvLevelSigma2 = mpORBextractorLeft->GetInverseScaleSigmaSquares()

// ... and after it (:92-116)
#define ORBFE_FRAME_FINISH(im)                                                                            \
    mvpMapPoints = std::vector<MapPoint *>(N, static_cast<MapPoint *>(NULL));                             \
    mvbOutlier = std::vector<bool>(N, false);                                                             \
    if (mbInitialComputations) {                                                                          \
        ComputeImageBounds(im);                                                                           \
        mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / static_cast<float>(mnMaxX - mnMinX); \
        mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / static_cast<float>(mnMaxY - mnMinY); \
        fx = K.at<float>(0, 0); fy = K.at<float>(1, 1); cx = K.at<float>(0, 2); cy = K.at<float>(1, 2);   \
        invfx = 1.0f / fx; invfy = 1.0f / fy;                                                             \
        mbInitialComputations = false;                                                                    \
    }                                                                                                     \
    mb = mbf / fx;                                                                                        \
    AssignFeaturesToGrid()

// Stereo (src/Frame.cc:61-117; constructed at src/Tracking.cc:296).
Frame::Frame(const cv::Mat &imLeft, const cv::Mat &imRight, const double &timeStamp, ORBextractor *extractorLeft, ORBextractor *extractorRight,
             fbow::Vocabulary *voc, cv::Mat &K, cv::Mat &distCoef, const float &bf, const float &thDepth)
    : mpFBOWvocabulary(voc), mpORBextractorLeft(extractorLeft), mpORBextractorRight(extractorRight), mTimeStamp(timeStamp), mK(K.clone()),
      mDistCoef(distCoef.clone()), mbf(bf), mThDepth(thDepth), mpReferenceKF(static_cast<KeyFrame *>(NULL))
{
    ORBFE_FRAME_SCALE_INFO();
    N = 0;
    if (imLeft.empty() || imRight.empty()) return; // both operator() calls return silently on an empty image (src/ORBextractor.cc:861-862): N = 0 (:83-86)
    require_gray(imLeft); require_gray(imRight);
    if (imLeft.cols != imRight.cols || imLeft.rows != imRight.rows) throw std::invalid_argument("Frame: left / right image sizes differ");
    // one device context serves both images, so the two extractors must have been built alike (src/Tracking.cc:125-128 does)
    if (extractorRight) {
        const orbfe_params &a = extractorLeft->Params(), &b = extractorRight->Params();
        if (a.nfeatures != b.nfeatures || a.scale_factor != b.scale_factor || a.nlevels != b.nlevels || a.ini_th_fast != b.ini_th_fast ||
            a.min_th_fast != b.min_th_fast || a.patch_size != b.patch_size || a.half_patch_size != b.half_patch_size || a.edge_threshold != b.edge_threshold)
            throw std::invalid_argument("Frame: the left and right ORBextractor differ; the fused stereo call runs both images with the left one's parameters");
    }
    orbfe_context *ctx = bind_context(extractorLeft, imLeft, K, distCoef, bf);

    // ExtractORB(0, imLeft) || ExtractORB(1, imRight) (:78-81) + ComputeStereoMatches (:90) in one call
    const int cap = orbfe_keypoint_capacity(ctx);
    std::vector<cv::KeyPoint> kl(cap), kr(cap);
    std::vector<uchar> dl((size_t)cap * 32), dr((size_t)cap * 32), right_rows;
    std::vector<float> ur(cap, -1.0f), dp(cap, -1.0f);
    const uchar *pr = imRight.data;
    size_t stride = (size_t)imLeft.step;
    if ((size_t)imRight.step != stride) { // the ABI takes one stride for the pair: repack the right image
        right_rows.resize(stride * imRight.rows);
        for (int y = 0; y < imRight.rows; y++) std::memcpy(&right_rows[(size_t)y * stride], imRight.data + (size_t)y * (size_t)imRight.step, (size_t)imRight.cols);
        pr = right_rows.data();
    }
    int nl = 0, nr = 0;
    check(ctx, orbfe_stereo_frame(ctx, imLeft.data, pr, imLeft.cols, imLeft.rows, stride, reinterpret_cast<orbfe_keypoint *>(kl.data()), dl.data(), &nl,
                                  reinterpret_cast<orbfe_keypoint *>(kr.data()), dr.data(), &nr, ur.data(), dp.data(), cap));
    extractorLeft->NoteResidentFrame(nl, dl.data(), 1);
    kl.resize(nl); kr.resize(nr);
    mvKeys.swap(kl); mvKeysRight.swap(kr);
    mDescriptors = cv::Mat(nl, 32, CV_8U);
    if (nl > 0) std::memcpy(mDescriptors.data, dl.data(), (size_t)nl * 32);
    mDescriptorsRight = cv::Mat(nr, 32, CV_8U);
    if (nr > 0) std::memcpy(mDescriptorsRight.data, dr.data(), (size_t)nr * 32);

    N = mvKeys.size();
    if (mvKeys.empty()) return;
    UndistortKeyPoints();
    mvuRight.assign(ur.begin(), ur.begin() + N); // what ComputeStereoMatches leaves (:466-467,620-622)
    mvDepth.assign(dp.begin(), dp.begin() + N);
    ORBFE_FRAME_FINISH(imLeft);
}

// RGB-D (src/Frame.cc:120-172; src/Tracking.cc:326).  imDepth is the CV_32F map Tracking::GrabImageRGBD converted (:323-324).
Frame::Frame(const cv::Mat &imGray, const cv::Mat &imDepth, const double &timeStamp, ORBextractor *extractor, fbow::Vocabulary *voc, cv::Mat &K,
             cv::Mat &distCoef, const float &bf, const float &thDepth)
    : mpFBOWvocabulary(voc), mpORBextractorLeft(extractor), mpORBextractorRight(static_cast<ORBextractor *>(NULL)), mTimeStamp(timeStamp),
      mK(K.clone()), mDistCoef(distCoef.clone()), mbf(bf), mThDepth(thDepth)
{
    ORBFE_FRAME_SCALE_INFO();
    N = 0;
    if (imGray.empty()) return;
    require_gray(imGray);
    if (imDepth.type() != CV_32F || imDepth.cols != imGray.cols || imDepth.rows != imGray.rows)
        throw std::invalid_argument("Frame: imDepth must be CV_32F of the image's size (src/Tracking.cc:323-324 converts it)");
    orbfe_context *ctx = bind_context(extractor, imGray, K, distCoef, bf);

    // ExtractORB(0, imGray) (:136) + ComputeStereoFromRGBD (:146) in one call
    const int cap = orbfe_keypoint_capacity(ctx);
    std::vector<cv::KeyPoint> k(cap);
    std::vector<uchar> d((size_t)cap * 32);
    std::vector<float> ur(cap, -1.0f), dp(cap, -1.0f);
    int n = 0;
    check(ctx, orbfe_rgbd_frame(ctx, imGray.data, imDepth.ptr<float>(0), imGray.cols, imGray.rows, (size_t)imGray.step, (size_t)imDepth.step,
                                reinterpret_cast<orbfe_keypoint *>(k.data()), d.data(), &n, ur.data(), dp.data(), cap));
    extractor->NoteResidentFrame(n, d.data(), 2);
    k.resize(n);
    mvKeys.swap(k);
    mDescriptors = cv::Mat(n, 32, CV_8U);
    if (n > 0) std::memcpy(mDescriptors.data, d.data(), (size_t)n * 32);

    N = mvKeys.size();
    if (mvKeys.empty()) return;
    UndistortKeyPoints();
    mvuRight.assign(ur.begin(), ur.begin() + N);
    mvDepth.assign(dp.begin(), dp.begin() + N);
    ORBFE_FRAME_FINISH(imGray);
}

// Monocular (src/Frame.cc:175-229; src/Tracking.cc:354-358).
Frame::Frame(const cv::Mat &imGray, const double &timeStamp, ORBextractor *extractor, fbow::Vocabulary *voc, cv::Mat &K, cv::Mat &distCoef,
             const float &bf, const float &thDepth)
    : mpFBOWvocabulary(voc), mpORBextractorLeft(extractor), mpORBextractorRight(static_cast<ORBextractor *>(NULL)), mTimeStamp(timeStamp),
      mK(K.clone()), mDistCoef(distCoef.clone()), mbf(bf), mThDepth(thDepth)
{
    ORBFE_FRAME_SCALE_INFO();
    N = 0;
    if (imGray.empty()) return;
    bind_context(extractor, imGray, K, distCoef, bf);
    ExtractORB(0, imGray);
    N = mvKeys.size();
    if (mvKeys.empty()) return;
    UndistortKeyPoints();
    mvuRight = std::vector<float>(N, -1); // :199-200
    mvDepth = std::vector<float>(N, -1);
    ORBFE_FRAME_FINISH(imGray);
}

// src/Frame.cc:231-246 -> orbfe_assign_features_to_grid (for the resident frame the device grid stays cached for the matcher
// calls of this frame)
void Frame::AssignFeaturesToGrid()
{
    for (unsigned int i = 0; i < FRAME_GRID_COLS; i++)
        for (unsigned int j = 0; j < FRAME_GRID_ROWS; j++) mGrid[i][j].clear();
    if (N <= 0) return;
    orbfe_context *ctx = context_of(*this);
    const orbfe_frame_view v = device_view_of(*this);
    std::vector<int32_t> off(FRAME_GRID_COLS * FRAME_GRID_ROWS + 1), idx(N);
    check(ctx, orbfe_assign_features_to_grid(ctx, &v, off.data(), idx.data()));
    for (int i = 0; i < FRAME_GRID_COLS; i++)
        for (int j = 0; j < FRAME_GRID_ROWS; j++) {
            const int c = i * FRAME_GRID_ROWS + j;
            mGrid[i][j].assign(idx.begin() + off[c], idx.begin() + off[c + 1]);
        }
}

// src/Frame.cc:248-254: the call operator of include/ORBextractor.h:58-60 (the mirror's cv:: overload)
void Frame::ExtractORB(int flag, const cv::Mat &im)
{
    if (flag == 0) (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors);
    else (*mpORBextractorRight)(im, cv::Mat(), mvKeysRight, mDescriptorsRight);
}

// src/Frame.cc:256-268.  The reference writes these with cv::Mat algebra; cv::gemm accumulates CV_32F products in double.
void Frame::SetPose(cv::Mat Tcw)
{
    mTcw = Tcw.clone();
    UpdatePoseMatrices();
}

void Frame::UpdatePoseMatrices()
{
    mRcw = cv::Mat(3, 3, CV_32F); mRwc = cv::Mat(3, 3, CV_32F); mtcw = cv::Mat(3, 1, CV_32F); mOw = cv::Mat(3, 1, CV_32F);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) { mRcw.at<float>(r, c) = mTcw.at<float>(r, c); mRwc.at<float>(c, r) = mTcw.at<float>(r, c); }
        mtcw.at<float>(r) = mTcw.at<float>(r, 3);
    }
    for (int i = 0; i < 3; i++) { // mOw = -mRcw.t() * mtcw
        double s = 0.0;
        for (int k = 0; k < 3; k++) s += (double)(-mRcw.at<float>(k, i)) * (double)mtcw.at<float>(k);
        mOw.at<float>(i) = (float)s;
    }
}

// src/Frame.cc:270-326 -> orbfe_is_in_frustum (one point; Tracking::SearchLocalPoints calls it per local map point)
bool Frame::isInFrustum(MapPoint *pMP, float viewingCosLimit)
{
    pMP->mbTrackInView = false;
    orbfe_context *ctx = context_of(*this);
    float T[12], pos[3], nrm[3];
    pose_3x4(mTcw, T);
    const cv::Mat P = pMP->GetWorldPos(), Pn = pMP->GetNormal();
    for (int k = 0; k < 3; k++) { pos[k] = P.at<float>(k); nrm[k] = Pn.at<float>(k); }
    const float maxd = raw_from_scaled(pMP->GetMaxDistanceInvariance(), 1.2f), mind = raw_from_scaled(pMP->GetMinDistanceInvariance(), 0.8f);
    orbfe_track_point tp;
    check(ctx, orbfe_is_in_frustum(ctx, T, mnMinX, mnMaxX, mnMinY, mnMaxY, 1, pos, nrm, &maxd, &mind, viewingCosLimit, &tp));
    if (!tp.in_view) return false;
    pMP->mbTrackInView = true; // :317-323
    pMP->mTrackProjX = tp.proj_x;
    pMP->mTrackProjXR = tp.proj_xr;
    pMP->mTrackProjY = tp.proj_y;
    pMP->mnTrackScaleLevel = tp.level;
    pMP->mTrackViewCos = tp.view_cos;
    return true;
}

// src/Frame.cc:328-381 -> orbfe_features_in_area (the matchers never come here: their window queries are fused on the device)
std::vector<size_t> Frame::GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel, const int maxLevel) const
{
    std::vector<size_t> vIndices;
    if (N <= 0) return vIndices;
    orbfe_context *ctx = context_of(*this);
    const orbfe_frame_view v = device_view_of(*this);
    std::vector<int32_t> out(N);
    int n = 0;
    check(ctx, orbfe_features_in_area(ctx, &v, x, y, r, minLevel, maxLevel, out.data(), N, &n));
    vIndices.assign(out.begin(), out.begin() + n);
    return vIndices;
}

// src/Frame.cc:383-393 (two lines of the reference, as they are; the grid itself comes from the device)
bool Frame::PosInGrid(const cv::KeyPoint &kp, int &posX, int &posY)
{
    posX = round((kp.pt.x - mnMinX) * mfGridElementWidthInv);
    posY = round((kp.pt.y - mnMinY) * mfGridElementHeightInv);
    return !(posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS);
}

// src/Frame.cc:395-400: mpFBOWvocabulary->transform(mDescriptors, 4, mFbowVec, mFbowFeatVec) -> orbfe_bow_transform + orbfe_bow_maps
void Frame::ComputeFboW()
{
    if (!(mFbowVec.empty() && mDescriptors.rows != 0)) return;
    orbfe_context *ctx = context_of(*this);
    ensure_vocabulary(ctx, mpFBOWvocabulary);
    const int n = mDescriptors.rows;
    std::vector<uint32_t> word(n), node(n), words(n), nodes(n);
    std::vector<float> weight(n), word_w(n);
    std::vector<int32_t> node_off(n + 1), node_feat(n);
    int n_words = 0, n_nodes = 0;
    check(ctx, orbfe_bow_transform(ctx, mDescriptors.ptr<uchar>(0), n, 4, word.data(), weight.data(), node.data()));
    check(ctx, orbfe_bow_maps(word.data(), weight.data(), node.data(), n, words.data(), word_w.data(), &n_words, nodes.data(), node_off.data(),
                              node_feat.data(), &n_nodes));
    for (int i = 0; i < n_words; i++) { float w = word_w[i]; mFbowVec[words[i]] = w; }
    for (int k = 0; k < n_nodes; k++) {
        std::vector<uint32_t> &f = mFbowFeatVec[nodes[k]];
        f.assign(node_feat.begin() + node_off[k], node_feat.begin() + node_off[k + 1]);
    }
}

// src/Frame.cc:402-432 -> orbfe_fetch_keys_un on the resident frame (undistorted on the device), orbfe_undistort_keypoints otherwise
void Frame::UndistortKeyPoints()
{
    if (mDistCoef.at<float>(0) == 0.0) {
        mvKeysUn = mvKeys;
        return;
    }
    orbfe_context *ctx = context_of(*this);
    mvKeysUn.resize(N);
    if (mpORBextractorLeft->IsResidentFrame(N, mDescriptors.ptr<uchar>(0))) {
        int n = 0;
        check(ctx, orbfe_fetch_keys_un(ctx, 0, reinterpret_cast<orbfe_keypoint *>(mvKeysUn.data()), N, &n));
        if (n != N) throw std::runtime_error("Frame::UndistortKeyPoints: the device slot holds another frame");
    } else
        check(ctx, orbfe_undistort_keypoints(ctx, keys_of(mvKeys), N, reinterpret_cast<orbfe_keypoint *>(mvKeysUn.data())));
}

// src/Frame.cc:434-462 -> orbfe_image_bounds (the context is bound to this image size)
void Frame::ComputeImageBounds(const cv::Mat &imLeft)
{
    orbfe_context *ctx = context_of(*this);
    if (imLeft.cols != mpORBextractorLeft->Params().width || imLeft.rows != mpORBextractorLeft->Params().height)
        throw std::invalid_argument("Frame::ComputeImageBounds: image size differs from the extractor's");
    float b[4];
    check(ctx, orbfe_image_bounds(ctx, b));
    mnMinX = b[0]; mnMaxX = b[1]; mnMinY = b[2]; mnMaxY = b[3];
}

// src/Frame.cc:464-642.  The stereo constructor already has the result (the fused call); a caller that invokes the member by
// itself gets it re-read from the device, which is only possible while this frame is the extractor's latest stereo call.
void Frame::ComputeStereoMatches()
{
    if (!mpORBextractorLeft->IsResidentFrame(N, mDescriptors.ptr<uchar>(0)) || mpORBextractorLeft->ResidentKind() != 1)
        throw std::logic_error("Frame::ComputeStereoMatches: stereo matching runs inside the stereo constructor's device call; this frame is not "
                               "the latest stereo frame of its extractor");
    orbfe_context *ctx = context_of(*this);
    mvuRight.assign(N, -1.0f); mvDepth.assign(N, -1.0f);
    int n = 0;
    check(ctx, orbfe_fetch_image(ctx, 0, NULL, NULL, mvuRight.data(), mvDepth.data(), N, &n));
}

// src/Frame.cc:645-666, same rule: the RGB-D constructor's call produced it.
void Frame::ComputeStereoFromRGBD(const cv::Mat & /*imDepth*/)
{
    if (!mpORBextractorLeft->IsResidentFrame(N, mDescriptors.ptr<uchar>(0)) || mpORBextractorLeft->ResidentKind() != 2)
        throw std::logic_error("Frame::ComputeStereoFromRGBD: depth sampling runs inside the RGB-D constructor's device call; this frame is not "
                               "the latest RGB-D frame of its extractor");
    orbfe_context *ctx = context_of(*this);
    mvuRight.assign(N, -1.0f); mvDepth.assign(N, -1.0f);
    int n = 0;
    check(ctx, orbfe_fetch_image(ctx, 0, NULL, NULL, mvuRight.data(), mvDepth.data(), N, &n));
}

// src/Frame.cc:668-683
cv::Mat Frame::UnprojectStereo(const int &i)
{
    const float z = mvDepth[i];
    if (z > 0) {
        const float u = mvKeysUn[i].pt.x, v = mvKeysUn[i].pt.y;
        const float c[3] = {(u - cx) * z * invfx, (v - cy) * z * invfy, z};
        cv::Mat x3D(3, 1, CV_32F);
        for (int r = 0; r < 3; r++) { // mRwc * x3Dc + mOw
            double s = 0.0;
            for (int k = 0; k < 3; k++) s += (double)mRwc.at<float>(r, k) * (double)c[k];
            x3D.at<float>(r) = (float)s + mOw.at<float>(r);
        }
        return x3D;
    }
    return cv::Mat();
}

} // namespace ORB_SLAM2
