// ORBextractor.h -- host-side mirror of ORB_SLAM2::ORBextractor over the C ABI.
//
// Same class name, constructor arguments, getters and call operator meaning as the
// reference's include/ORBextractor.h:45-112, so Tracking (src/Tracking.cc:125-131) and
// Frame::ExtractORB (src/Frame.cc:248-254) can keep calling it.  The arithmetic runs in
// liborbfe.so (HIP, gfx950); this header only converts types.  OpenCV is optional: the
// core overloads use plain views / orbfe_keypoint (layout-identical to cv::KeyPoint);
// define ORBFE_WITH_OPENCV to get the cv::InputArray / cv::OutputArray overloads the
// reference declares (include/ORBextractor.h:58-60).
//
// Differences a maintainer must know (also in INTEGRATION.md):
//  * the image size is bound when the first frame arrives, or by BindImageSize() (one camera
//    model per extractor, as the reference assumes through Frame's statics, src/Frame.cc:29-33);
//    the device context is created once and never re-created (see Context());
//  * mvImagePyramid is materialised on the host only if KeepHostPyramid(true) is set
//    (the reference's own ComputeStereoMatches needs it; orbfe_stereo_frame does not).
#pragma once

#include <atomic>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/orbfe.h"

#ifdef ORBFE_WITH_OPENCV
#include <opencv2/core/core.hpp>
#endif

namespace ORB_SLAM2
{

// Non-owning 8UC1 image view (what cv::Mat gives the reference: data, cols, rows, step).
struct ImageView {
    const uint8_t *data = nullptr;
    int cols = 0, rows = 0;
    size_t step = 0;
    bool empty() const { return !data || cols <= 0 || rows <= 0; }
};

struct CameraParams {
    float fx = 0.f, fy = 0.f, cx = 0.f, cy = 0.f, bf = 0.f;
    std::vector<float> distCoef; // Tracking's mDistCoef: k1 k2 p1 p2 [k3] (src/Tracking.cc:67-78); empty = none
    bool rgb = true;             // Tracking's mbRGB (colour input order)
};

class ORBextractor
{
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    // include/ORBextractor.h:51 -- same eight arguments, same order.
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                 int patchSize, int halfPatchSize, int edgeThreshold)
    {
        std::memset(&mParams, 0, sizeof(mParams));
        mParams.nfeatures = nfeatures;
        mParams.scale_factor = scaleFactor;
        mParams.nlevels = nlevels;
        mParams.ini_th_fast = iniThFAST;
        mParams.min_th_fast = minThFAST;
        mParams.patch_size = patchSize;
        mParams.half_patch_size = halfPatchSize;
        mParams.edge_threshold = edgeThreshold;
        mParams.max_images = 1;
        // scale tables exactly as src/ORBextractor.cc:409-425 (scaleFactor member is double)
        const double sf = (double)scaleFactor;
        mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels);
        mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;
        for (int i = 1; i < nlevels; i++) {
            mvScaleFactor[i] = (float)((double)mvScaleFactor[i - 1] * sf);
            mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
        }
        for (int i = 0; i < nlevels; i++) {
            mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
            mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
        }
    }

    ~ORBextractor()
    {
        if (mCtx) {
            orbfe_context *mine = mCtx;
            DefaultSlot().compare_exchange_strong(mine, nullptr); // only if the default is still this extractor's
            orbfe_destroy(mCtx);
        }
    }
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // camera and device must be set before the first frame (they are baked into the device context)
    void SetCamera(const CameraParams &c)
    {
        if (mCtx) throw std::logic_error("ORBextractor::SetCamera after the device context was created");
        mCam = c;
    }
    void SetDevice(int device)
    {
        if (mCtx) throw std::logic_error("ORBextractor::SetDevice after the device context was created");
        mParams.device = device;
    }
    void KeepHostPyramid(bool keep) { mKeepPyramid = keep; }

    // ORBextractor::operator() (src/ORBextractor.cc:858-919).  mask is ignored, as in the
    // reference.  Empty image: silent return, outputs untouched (:861-862).
    void operator()(const ImageView &image, std::vector<orbfe_keypoint> &keypoints, std::vector<uint8_t> &descriptors)
    {
        if (image.empty()) return;
        EnsureContext(image.cols, image.rows, 1);
        const int cap = orbfe_keypoint_capacity(mCtx);
        keypoints.resize(cap);
        descriptors.resize((size_t)cap * 32);
        int n = 0;
        Check(orbfe_extract(mCtx, image.data, image.cols, image.rows, image.step, keypoints.data(), descriptors.data(), cap, &n));
        keypoints.resize(n);
        descriptors.resize((size_t)n * 32);
        NoteResidentFrame(n, descriptors.data());
        if (mKeepPyramid) FetchPyramid();
    }

    // Identity of the frame that currently sits in image slot 0 of the device context (the latest extraction): keypoint count +
    // a hash of ALL its descriptors (8 bytes per step: a few microseconds for 2000 keypoints).  The reference-signature shims
    // (compat/ORBmatcher.cc, compat/Frame.cc) compare a Frame against it to decide whether the matchers may read that frame in
    // HBM (orbfe_frame_view.device_slot_plus1) instead of uploading it; the library additionally refuses a view whose count is
    // not the slot's.  `kind` says what else the slot holds: 1 = mvuRight / mvDepth of orbfe_stereo_frame, 2 = of orbfe_rgbd_frame.
    static uint64_t FrameFingerprint(int n, const uint8_t *descriptors)
    {
        uint64_t h = 1469598103934665603ull ^ (uint64_t)(uint32_t)n;
        const size_t words = (size_t)n * 4;
        for (size_t i = 0; i < words; i++) { uint64_t w; std::memcpy(&w, descriptors + 8 * i, 8); h = (h ^ w) * 1099511628211ull; h ^= h >> 29; }
        return h;
    }
    bool IsResidentFrame(int n, const uint8_t *descriptors) const { return mCtx && n > 0 && mResidentN == n && mResidentFp == FrameFingerprint(n, descriptors); }
    void NoteResidentFrame(int n, const uint8_t *descriptors, int kind = 0) { mResidentN = n; mResidentFp = n > 0 ? FrameFingerprint(n, descriptors) : 0; mResidentKind = kind; }
    int ResidentKind() const { return mResidentKind; }
    const orbfe_params &Params() const { return mParams; } // the eight constructor arguments (+ what the context was bound to)
    const CameraParams &Camera() const { return mCam; }

    int inline GetLevels() { return mParams.nlevels; }
    float inline GetScaleFactor() { return (float)(double)mParams.scale_factor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // Host copy of the latest pyramid (only with KeepHostPyramid(true)): level l is
    // mvPyramidData[l] with mvPyramidCols[l] x mvPyramidRows[l], step == cols.
    std::vector<std::vector<uint8_t>> mvPyramidData;
    std::vector<int> mvPyramidCols, mvPyramidRows;

#ifdef ORBFE_WITH_OPENCV
    // include/ORBextractor.h:58-60
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint> &keypoints, cv::OutputArray descriptors)
    {
        (void)mask; // ignored, as in the reference (src/ORBextractor.cc:858-866 never reads it)
        if (image.empty()) return;
        const cv::Mat img = image.getMat();
        if (img.type() != CV_8UC1) throw std::invalid_argument("ORBextractor: image must be CV_8UC1"); // assert at :865
        static_assert(sizeof(cv::KeyPoint) == sizeof(orbfe_keypoint), "cv::KeyPoint layout");
        std::vector<orbfe_keypoint> kps;
        std::vector<uint8_t> desc;
        (*this)(ImageView{img.data, img.cols, img.rows, (size_t)img.step}, kps, desc);
        keypoints.resize(kps.size());
        if (!kps.empty()) std::memcpy((void *)keypoints.data(), kps.data(), kps.size() * sizeof(orbfe_keypoint));
        if (kps.empty()) { descriptors.release(); }
        else {
            descriptors.create((int)kps.size(), 32, CV_8U);
            std::memcpy(descriptors.getMat().data, desc.data(), desc.size());
        }
        if (mKeepPyramid) {
            mvImagePyramid.resize(mParams.nlevels);
            for (int l = 0; l < mParams.nlevels; l++)
                mvImagePyramid[l] = cv::Mat(mvPyramidRows[l], mvPyramidCols[l], CV_8UC1, mvPyramidData[l].data());
        }
    }
    std::vector<cv::Mat> mvImagePyramid; // include/ORBextractor.h:84
#endif

    // The device context of this extractor.  LIFETIME RULE: it is created ONCE -- by the first frame (whose size it takes),
    // or earlier by BindImageSize() -- and lives until the extractor is destroyed.  ORBmatcher, KeyFrameDatabase and the
    // vocabulary keep this pointer (and state inside it), so it is never re-created behind their back: a later frame of
    // another size, or a batch beyond the capacity fixed at creation, throws instead.
    orbfe_context *Context() { return mCtx; }

    // Fix the image size (and, optionally, the number of images per batched call) before the first frame, e.g. right
    // after construction in Tracking::Tracking, so that Context() can be handed to the matchers / database at once.
    void BindImageSize(int width, int height, int maxImages = 2) { EnsureContext(width, height, maxImages); }

    // Creates the device context on first use: always room for a stereo pair (max_images >= 2), so operator() followed by
    // ComputeStereoFrame never needs a second context.
    void EnsureContext(int width, int height, int maxImages)
    {
        if (mCtx) {
            if (mParams.width != width || mParams.height != height)
                throw std::invalid_argument("ORBextractor: image is " + std::to_string(width) + "x" + std::to_string(height) + ", this extractor is bound to " +
                                            std::to_string(mParams.width) + "x" + std::to_string(mParams.height) + " (one camera model per extractor)");
            if (mParams.max_images < maxImages)
                throw std::invalid_argument("ORBextractor: " + std::to_string(maxImages) + " images per call exceed the capacity fixed at creation (" +
                                            std::to_string(mParams.max_images) + "); call BindImageSize(w, h, maxImages) before the first frame");
            return;
        }
        mParams.width = width; mParams.height = height;
        mParams.max_images = maxImages > 2 ? maxImages : 2;
        mParams.fx = mCam.fx; mParams.fy = mCam.fy; mParams.cx = mCam.cx; mParams.cy = mCam.cy; mParams.bf = mCam.bf;
        const int rc = orbfe_create(&mParams, &mCtx);
        if (rc != ORBFE_OK) throw std::runtime_error(std::string("orbfe_create: ") + orbfe_last_error(nullptr));
        if (!mCam.distCoef.empty()) Check(orbfe_set_distortion(mCtx, mCam.distCoef.data(), (int)mCam.distCoef.size()));
        orbfe_context *none = nullptr;
        DefaultSlot().compare_exchange_strong(none, mCtx); // first context of the process, unless SetAsDefault() chose one
    }

    // The process-wide default context: what the keyframe-only ORBmatcher overloads (no Frame at hand to reach
    // mpORBextractorLeft), KeyFrameDatabase and Optimizer use.  It is the FIRST device context created in the process (Tracking
    // news the left extractor first, src/Tracking.cc:125, and Frame's stereo constructor touches the left one first) and never
    // changes behind a caller's back; an integration that wants to be explicit calls mpORBextractorLeft->SetAsDefault() after
    // BindImageSize().  The keyframe-side entry points project with the context's camera, so the shims check it against
    // pKF->fx ... (compat/ORBmatcher.cc: kf_context) instead of trusting this choice.  Atomic: in mode A of INTEGRATION.md the
    // two extractors create their contexts on two threads (src/Frame.cc:78-81).
    static orbfe_context *DefaultContext() { return DefaultSlot().load(); }
    void SetAsDefault()
    {
        if (!mCtx) throw std::logic_error("ORBextractor::SetAsDefault before the device context exists (call BindImageSize first)");
        DefaultSlot().store(mCtx);
    }

    // Frame::UndistortKeyPoints (src/Frame.cc:402-432) and ComputeImageBounds (:434-462) for this camera
    void UndistortKeyPoints(const std::vector<orbfe_keypoint> &vKeys, std::vector<orbfe_keypoint> &vKeysUn)
    {
        vKeysUn.resize(vKeys.size());
        Check(orbfe_undistort_keypoints(mCtx, vKeys.data(), (int)vKeys.size(), vKeysUn.data()));
    }
    void ComputeImageBounds(float &mnMinX, float &mnMaxX, float &mnMinY, float &mnMaxY)
    {
        float b[4];
        Check(orbfe_image_bounds(mCtx, b));
        mnMinX = b[0]; mnMaxX = b[1]; mnMinY = b[2]; mnMaxY = b[3];
    }
    // Tracking::GrabImage*'s cvtColor (src/Tracking.cc:269-294): feed 3- / 4-channel frames as they are
    void SetInputChannels(int channels) { Check(orbfe_set_input_format(mCtx, channels, mCam.rgb ? 1 : 0, 0)); }

protected:
    static std::atomic<orbfe_context *> &DefaultSlot()
    {
        static std::atomic<orbfe_context *> slot(nullptr);
        return slot;
    }
    void Check(int rc)
    {
        if (rc != ORBFE_OK) throw std::runtime_error(std::string("orbfe: ") + orbfe_last_error(mCtx));
    }
    void FetchPyramid()
    {
        const int nl = mParams.nlevels;
        mvPyramidData.resize(nl); mvPyramidCols.resize(nl); mvPyramidRows.resize(nl);
        for (int l = 0; l < nl; l++) {
            int w = 0, h = 0;
            Check(orbfe_level_size(mCtx, l, &w, &h));
            mvPyramidCols[l] = w; mvPyramidRows[l] = h;
            mvPyramidData[l].resize((size_t)w * h);
            Check(orbfe_fetch_pyramid(mCtx, 0, l, 0, mvPyramidData[l].data(), (size_t)w));
        }
    }

    orbfe_params mParams;
    CameraParams mCam;
    orbfe_context *mCtx = nullptr;
    bool mKeepPyramid = false;
    int mResidentN = -1;
    int mResidentKind = 0;
    uint64_t mResidentFp = 0;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

// Frame::Frame(stereo) body (src/Frame.cc:61-117): both extractions + ComputeStereoMatches in one
// device round trip.  Outputs are the Frame members the reference fills: mvKeys / mDescriptors,
// mvKeysRight / mDescriptorsRight, mvuRight, mvDepth.
struct StereoFrameOutput {
    std::vector<orbfe_keypoint> mvKeys, mvKeysRight;
    std::vector<uint8_t> mDescriptors, mDescriptorsRight; // N x 32
    std::vector<float> mvuRight, mvDepth;
    int N = 0;
};

inline void ComputeStereoFrame(ORBextractor &extractorLeft, const ImageView &imLeft, const ImageView &imRight, StereoFrameOutput &out)
{
    if (imLeft.empty() || imRight.empty()) { out = StereoFrameOutput(); return; }
    if (imLeft.cols != imRight.cols || imLeft.rows != imRight.rows || imLeft.step != imRight.step)
        throw std::invalid_argument("ComputeStereoFrame: left/right geometry differs");
    extractorLeft.EnsureContext(imLeft.cols, imLeft.rows, 2);
    orbfe_context *ctx = extractorLeft.Context();
    const int cap = orbfe_keypoint_capacity(ctx);
    out.mvKeys.resize(cap); out.mvKeysRight.resize(cap);
    out.mDescriptors.resize((size_t)cap * 32); out.mDescriptorsRight.resize((size_t)cap * 32);
    out.mvuRight.assign(cap, -1.0f); out.mvDepth.assign(cap, -1.0f);
    int nl = 0, nr = 0;
    const int rc = orbfe_stereo_frame(ctx, imLeft.data, imRight.data, imLeft.cols, imLeft.rows, imLeft.step,
                                      out.mvKeys.data(), out.mDescriptors.data(), &nl,
                                      out.mvKeysRight.data(), out.mDescriptorsRight.data(), &nr,
                                      out.mvuRight.data(), out.mvDepth.data(), cap);
    if (rc != ORBFE_OK) throw std::runtime_error(std::string("orbfe_stereo_frame: ") + orbfe_last_error(ctx));
    out.N = nl;
    extractorLeft.NoteResidentFrame(nl, out.mDescriptors.data(), 1);
    out.mvKeys.resize(nl); out.mDescriptors.resize((size_t)nl * 32);
    out.mvKeysRight.resize(nr); out.mDescriptorsRight.resize((size_t)nr * 32);
    out.mvuRight.resize(nl); out.mvDepth.resize(nl);
}

// Frame::Frame(rgbd) (src/Frame.cc:120-172): ExtractORB, UndistortKeyPoints, ComputeStereoFromRGBD.  imDepth is the CV_32F
// map the constructor receives; the u16 overload takes the sensor's raw map and Tracking's mDepthMapFactor
// (src/Tracking.cc:323-324).
struct RGBDFrameOutput {
    std::vector<orbfe_keypoint> mvKeys, mvKeysUn;
    std::vector<uint8_t> mDescriptors;
    std::vector<float> mvuRight, mvDepth;
    int N = 0;
};

inline void ComputeRGBDFrame(ORBextractor &extractor, const ImageView &imGray, const void *imDepth, size_t depthStep, bool depthIsU16,
                             float depthMapFactor, RGBDFrameOutput &out)
{
    if (imGray.empty() || !imDepth) { out = RGBDFrameOutput(); return; }
    extractor.EnsureContext(imGray.cols, imGray.rows, 1);
    orbfe_context *ctx = extractor.Context();
    const int cap = orbfe_keypoint_capacity(ctx);
    out.mvKeys.resize(cap); out.mvKeysUn.resize(cap); out.mDescriptors.resize((size_t)cap * 32);
    out.mvuRight.assign(cap, -1.0f); out.mvDepth.assign(cap, -1.0f);
    int n = 0, n2 = 0;
    const int rc = depthIsU16
        ? orbfe_rgbd_frame_u16(ctx, imGray.data, (const uint16_t *)imDepth, depthMapFactor, imGray.cols, imGray.rows, imGray.step, depthStep,
                               out.mvKeys.data(), out.mDescriptors.data(), &n, out.mvuRight.data(), out.mvDepth.data(), cap)
        : orbfe_rgbd_frame(ctx, imGray.data, (const float *)imDepth, imGray.cols, imGray.rows, imGray.step, depthStep,
                           out.mvKeys.data(), out.mDescriptors.data(), &n, out.mvuRight.data(), out.mvDepth.data(), cap);
    if (rc != ORBFE_OK) throw std::runtime_error(std::string("orbfe_rgbd_frame: ") + orbfe_last_error(ctx));
    if (orbfe_fetch_keys_un(ctx, 0, out.mvKeysUn.data(), cap, &n2) != ORBFE_OK || n2 != n)
        throw std::runtime_error(std::string("orbfe_fetch_keys_un: ") + orbfe_last_error(ctx));
    out.N = n;
    extractor.NoteResidentFrame(n, out.mDescriptors.data(), 2);
    out.mvKeys.resize(n); out.mvKeysUn.resize(n); out.mDescriptors.resize((size_t)n * 32);
    out.mvuRight.resize(n); out.mvDepth.resize(n);
}

} // namespace ORB_SLAM2
