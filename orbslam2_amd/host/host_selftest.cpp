// host_selftest.cpp -- exercises the C++ mirror (ORBextractor.h) the way Frame::Frame(stereo) would.
// usage: host_selftest <raw_left_u8> <raw_right_u8> <width> <height> <nfeatures> <fx> <bf> <out_prefix>
// Writes <out_prefix>.kl / .dl / .kr / .dr / .ur / .dp (raw binary) for the pytest comparison.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>

#include "ORBextractor.h"
#include "ORBmatcher.h"
#include <cstring>
#include "KeyFrameDatabase.h"
#include "Optimizer.h"

static std::vector<uint8_t> slurp(const char *path, size_t n)
{
    std::vector<uint8_t> v(n);
    std::ifstream f(path, std::ios::binary);
    if (!f.read((char *)v.data(), (std::streamsize)n)) { std::cerr << "cannot read " << path << "\n"; std::exit(2); }
    return v;
}
template <typename T>
static void dump(const std::string &path, const std::vector<T> &v)
{
    std::ofstream f(path, std::ios::binary);
    f.write((const char *)v.data(), (std::streamsize)(v.size() * sizeof(T)));
}

int main(int argc, char **argv)
{
    if (argc != 9) { std::cerr << "usage: host_selftest left right w h nfeatures fx bf out_prefix\n"; return 2; }
    const int w = std::atoi(argv[3]), h = std::atoi(argv[4]), nf = std::atoi(argv[5]);
    const float fx = (float)std::atof(argv[6]), bf = (float)std::atof(argv[7]);
    const std::string prefix = argv[8];
    std::vector<uint8_t> left = slurp(argv[1], (size_t)w * h), right = slurp(argv[2], (size_t)w * h);
    try {
        // same construction as src/Tracking.cc:125-131 with the defaults of :118-123
        ORB_SLAM2::ORBextractor extractorLeft(nf, 1.2f, 8, 20, 7, 31, 15, 19);
        ORB_SLAM2::CameraParams cam; cam.fx = fx; cam.fy = fx; cam.cx = w * 0.5f; cam.cy = h * 0.5f; cam.bf = bf;
        extractorLeft.SetCamera(cam);
        ORB_SLAM2::StereoFrameOutput out;
        ORB_SLAM2::ComputeStereoFrame(extractorLeft, ORB_SLAM2::ImageView{left.data(), w, h, (size_t)w},
                                      ORB_SLAM2::ImageView{right.data(), w, h, (size_t)w}, out);
        // mono call through operator(), pyramid kept on the host like mvImagePyramid
        ORB_SLAM2::ORBextractor mono(nf, 1.2f, 8, 20, 7, 31, 15, 19);
        mono.KeepHostPyramid(true);
        std::vector<orbfe_keypoint> k; std::vector<uint8_t> d;
        mono(ORB_SLAM2::ImageView{left.data(), w, h, (size_t)w}, k, d);
        if (k.size() != out.mvKeys.size() || d != out.mDescriptors) { std::cerr << "mono/stereo mismatch\n"; return 3; }
        if ((int)mono.mvPyramidData.size() != mono.GetLevels() || mono.mvPyramidCols[0] != w) { std::cerr << "pyramid missing\n"; return 4; }
        // empty image: silent return, outputs untouched
        std::vector<orbfe_keypoint> k2(3); std::vector<uint8_t> d2(96);
        mono(ORB_SLAM2::ImageView{}, k2, d2);
        if (k2.size() != 3 || d2.size() != 96) { std::cerr << "empty-image contract broken\n"; return 5; }
        // ORBmatcher mirror: mono initialisation matching of the left frame against the right frame
        ORB_SLAM2::FrameView F1, F2;
        F1.mvKeysUn = out.mvKeys; F1.mDescriptors = out.mDescriptors;
        F2.mvKeysUn = out.mvKeysRight; F2.mDescriptors = out.mDescriptorsRight;
        F1.mnMinX = F2.mnMinX = 0.f; F1.mnMaxX = F2.mnMaxX = (float)w; F1.mnMinY = F2.mnMinY = 0.f; F1.mnMaxY = F2.mnMaxY = (float)h;
        std::vector<float> prev(2 * out.mvKeys.size());
        for (size_t i = 0; i < out.mvKeys.size(); i++) { prev[2 * i] = out.mvKeys[i].x; prev[2 * i + 1] = out.mvKeys[i].y; }
        ORB_SLAM2::ORBmatcher matcher(extractorLeft.Context(), 0.9f, true); // src/Tracking.cc:698
        std::vector<int32_t> m12;
        const int nInit = matcher.SearchForInitialization(F1, F2, prev, m12, 100);
        if (ORB_SLAM2::ORBmatcher::DescriptorDistance(out.mDescriptors.data(), out.mDescriptors.data()) != 0) { std::cerr << "distance\n"; return 6; }
        dump(prefix + ".m12", m12); dump(prefix + ".pm", prev);
        std::printf("init matches=%d\n", nInit);
        dump(prefix + ".kl", out.mvKeys); dump(prefix + ".dl", out.mDescriptors);
        dump(prefix + ".kr", out.mvKeysRight); dump(prefix + ".dr", out.mDescriptorsRight);
        dump(prefix + ".ur", out.mvuRight); dump(prefix + ".dp", out.mvDepth);
        // Optimizer mirror: map points = the stereo keypoints back-projected at the identity pose (Frame::UnprojectStereo,
        // src/Frame.cc:668-682); start 5 cm off; the optimiser has to come back to the identity
        {
            const size_t N = out.mvKeys.size();
            std::vector<uint8_t> has(N, 0), outlier;
            std::vector<float> Xw(3 * N, 0.f);
            for (size_t i = 0; i < N; i++) {
                const float z = out.mvDepth[i];
                if (z <= 0) continue;
                has[i] = 1;
                Xw[3 * i] = (out.mvKeys[i].x - cam.cx) * z / cam.fx;
                Xw[3 * i + 1] = (out.mvKeys[i].y - cam.cy) * z / cam.fy;
                Xw[3 * i + 2] = z;
            }
            float Tcw[16] = {1, 0, 0, 0.05f, 0, 1, 0, -0.02f, 0, 0, 1, 0.03f, 0, 0, 0, 1};
            const int nInl = ORB_SLAM2::Optimizer::PoseOptimization(extractorLeft.Context(), Tcw, out.mvKeys, out.mvuRight, has, Xw, outlier);
            dump(prefix + ".Tcw", std::vector<float>(Tcw, Tcw + 16)); dump(prefix + ".outl", outlier);
            dump(prefix + ".has", has); dump(prefix + ".Xw", Xw);
            std::printf("pose inliers=%d\n", nInl);
        }
        // RGB-D frame mirror on a distorted camera (TUM1 coefficients): constant 2 m depth map
        {
            ORB_SLAM2::ORBextractor rgbd(nf, 1.2f, 8, 20, 7, 31, 15, 19);
            ORB_SLAM2::CameraParams c2 = cam;
            c2.distCoef = {0.262383f, -0.953104f, -0.005358f, 0.002628f, 1.163314f};
            rgbd.SetCamera(c2);
            std::vector<float> depth((size_t)w * h, 2.0f);
            ORB_SLAM2::RGBDFrameOutput ro;
            ORB_SLAM2::ComputeRGBDFrame(rgbd, ORB_SLAM2::ImageView{left.data(), w, h, (size_t)w}, depth.data(), sizeof(float) * w, false, 1.0f, ro);
            float b[4];
            rgbd.ComputeImageBounds(b[0], b[1], b[2], b[3]);
            dump(prefix + ".kun", ro.mvKeysUn); dump(prefix + ".urd", ro.mvuRight); dump(prefix + ".bounds", std::vector<float>(b, b + 4));
            if (ro.mvKeys.size() != out.mvKeys.size() || std::memcmp(ro.mvKeys.data(), out.mvKeys.data(), sizeof(orbfe_keypoint) * ro.mvKeys.size()) != 0) { std::cerr << "rgbd keypoints differ from the stereo left image\n"; return 7; }
        }
        std::printf("N=%d NR=%zu levels=%d scale1=%.9g\n", out.N, out.mvKeysRight.size(), extractorLeft.GetLevels(),
                    (double)extractorLeft.GetScaleFactors()[1]);
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
