// pin.cpp -- pins this repository's CPU oracle (oracle/orb_oracle*.c) to the REAL reference on a machine that has OpenCV >= 4.5.1.
//
// Why: the reference (fabrizioromanelli/ORBSLAM2) has no golden vectors for its front-end and its arithmetic lives in OpenCV
// (cv::FAST, cv::resize, cv::GaussianBlur, cv::fastAtan2, cvRound: src/ORBextractor.cc:76,98,110,803,808,900,934; cv::norm,
// Mat::convertTo: src/Frame.cc:565-586).  The build image of this repository has no OpenCV, so every "bit-exact" claim in it means
// HIP == oracle, and the oracle's OpenCV half (tagged OPENCV-4.5.5-SEMANTICS in oracle/orb_oracle.c) is restated from the published
// algorithms.  This program produces what is missing: outputs of the reference's OWN src/ORBextractor.cc (compiled where it
// lies, never copied: CMakeLists.txt takes -DORBSLAM2_ROOT=<reference checkout>) linked against a real OpenCV, on the inputs this
// repository's tests use, plus per-primitive dumps keyed to the oracle's tags.  tools/opencv_pin/import_pins.py turns the output
// into tests/golden/reference_pinned/*.npz, which tests/test_reference_pins.py then holds the oracle (CPU) and the HIP path (GPU) to.
//
//   pin <inputs dir (tools/opencv_pin/export_inputs.py)> <output dir>
//
// Output: one container file per case / primitive, format "ORBPIN01": records of {name, dtype code, ndim, dims, raw little-endian
// data} (read by import_pins.py).  Nothing here is product code; nothing of the reference's text is reproduced -- the stereo
// matcher below is a restatement of Frame::ComputeStereoMatches (src/Frame.cc:464-642; that translation unit needs the whole SLAM
// system and cannot be compiled alone) that keeps every OpenCV call of the original, so OpenCV's arithmetic is the real one.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <utility>
#include <vector>

#include <opencv2/calib3d.hpp>
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#include <opencv2/imgproc.hpp>

#include "ORBextractor.h" // the reference's own header (include/ORBextractor.h), found through ORBSLAM2_ROOT

namespace
{
// ---- container ----
enum Dtype : uint32_t { U8 = 0, I32 = 1, F32 = 2, F64 = 3, U16 = 4 };
struct Writer {
    std::ofstream f;
    explicit Writer(const std::string &path) : f(path, std::ios::binary) { f.write("ORBPIN01", 8); }
    void put(const std::string &name, Dtype t, const std::vector<uint32_t> &dims, const void *data)
    {
        static const size_t esz[] = {1, 4, 4, 8, 2};
        const uint32_t nl = (uint32_t)name.size(), nd = (uint32_t)dims.size(), tt = (uint32_t)t;
        f.write((const char *)&nl, 4); f.write(name.data(), nl);
        f.write((const char *)&tt, 4); f.write((const char *)&nd, 4);
        size_t n = 1;
        for (uint32_t d : dims) { f.write((const char *)&d, 4); n *= d; }
        f.write((const char *)data, (std::streamsize)(n * esz[t]));
    }
    void put_mat_u8(const std::string &name, const cv::Mat &m)
    {
        std::vector<uint8_t> buf((size_t)m.rows * m.cols);
        for (int r = 0; r < m.rows; r++) std::memcpy(buf.data() + (size_t)r * m.cols, m.ptr<uint8_t>(r), (size_t)m.cols);
        put(name, U8, {(uint32_t)m.rows, (uint32_t)m.cols}, buf.data());
    }
    void put_f32(const std::string &name, const std::vector<float> &v) { put(name, F32, {(uint32_t)v.size()}, v.data()); }
    void put_i32(const std::string &name, const std::vector<int32_t> &v) { put(name, I32, {(uint32_t)v.size()}, v.data()); }
    // cv::KeyPoint is 28 bytes: pt.x pt.y size angle response octave class_id = this repository's orbfe_keypoint / oracle record
    void put_keys(const std::string &name, const std::vector<cv::KeyPoint> &k)
    {
        std::vector<uint8_t> buf(k.size() * 28);
        for (size_t i = 0; i < k.size(); i++) {
            const float f5[5] = {k[i].pt.x, k[i].pt.y, k[i].size, k[i].angle, k[i].response};
            const int32_t i2[2] = {k[i].octave, k[i].class_id};
            std::memcpy(&buf[i * 28], f5, 20); std::memcpy(&buf[i * 28 + 20], i2, 8);
        }
        put(name, U8, {(uint32_t)k.size(), 28u}, buf.data());
    }
};

// binary PGM (P5, maxval 255) as tools/opencv_pin/export_inputs.py writes it: no dependency on imgcodecs
cv::Mat read_pgm(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    std::string magic; int w = 0, h = 0, maxv = 0;
    f >> magic >> w >> h >> maxv;
    if (!f || magic != "P5" || maxv != 255 || w <= 0 || h <= 0) { std::fprintf(stderr, "pin: cannot read %s\n", path.c_str()); std::exit(2); }
    f.get(); // the single whitespace after maxval
    cv::Mat m(h, w, CV_8UC1);
    f.read((char *)m.data, (std::streamsize)((size_t)w * h));
    return m;
}

// deterministic filler for the primitive dumps (the inputs are dumped next to the outputs, so the generator need not be reproduced)
struct Lcg { uint64_t s; explicit Lcg(uint64_t seed) : s(seed * 6364136223846793005ull + 1442695040888963407ull) {}
             uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); } };

// ---- Frame::ComputeStereoMatches restated (src/Frame.cc:464-642): control flow re-expressed, every OpenCV call kept ----
int descriptor_distance(const cv::Mat &a, const cv::Mat &b) // ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1643-1659) = 256-bit Hamming distance
{
    const uint8_t *pa = a.ptr<uint8_t>(), *pb = b.ptr<uint8_t>();
    int d = 0;
    for (int i = 0; i < 32; i++) d += __builtin_popcount((unsigned)(pa[i] ^ pb[i]));
    return d;
}

void stereo_matches(ORB_SLAM2::ORBextractor &exL, ORB_SLAM2::ORBextractor &exR, const std::vector<cv::KeyPoint> &keysL, const cv::Mat &descL,
                    const std::vector<cv::KeyPoint> &keysR, const cv::Mat &descR, float mbf, float fx, std::vector<float> &uRight,
                    std::vector<float> &depth, std::vector<int32_t> &sad)
{
    const int N = (int)keysL.size();
    uRight.assign(N, -1.0f); depth.assign(N, -1.0f); sad.assign(N, -1);
    const std::vector<float> scales = exL.GetScaleFactors(), inv_scales = exL.GetInverseScaleFactors();
    const int TH_HIGH = 100, TH_LOW = 50, thOrbDist = (TH_HIGH + TH_LOW) / 2; // include/ORBmatcher.h:90-91
    const int nRows = exL.mvImagePyramid[0].rows;
    std::vector<std::vector<size_t> > rows(nRows); // vRowIndices (:474-491)
    for (size_t iR = 0; iR < keysR.size(); iR++) {
        const float y = keysR[iR].pt.y, r = 2.0f * scales[keysR[iR].octave];
        const int hi = (int)std::ceil(y + r), lo = (int)std::floor(y - r);
        for (int yi = lo; yi <= hi; yi++) rows[yi].push_back(iR);
    }
    const float mb = mbf / fx; // Frame::mb := mbf / fx (src/Frame.cc:112)
    if (mb == 0) return;
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    std::vector<std::pair<int, int> > dist_idx;
    for (int iL = 0; iL < N; iL++) {
        const cv::KeyPoint &kpL = keysL[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.pt.y, uL = kpL.pt.x;
        const std::vector<size_t> &cand = rows[(size_t)vL];
        if (cand.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH; size_t bestIdxR = 0;
        const cv::Mat dL = descL.row(iL);
        for (size_t c = 0; c < cand.size(); c++) {
            const size_t iR = cand[c];
            const cv::KeyPoint &kpR = keysR[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.pt.x;
            if (uR >= minU && uR <= maxU) {
                const int d = descriptor_distance(dL, descR.row((int)iR));
                if (d < bestDist) { bestDist = d; bestIdxR = iR; }
            }
        }
        if (bestDist >= thOrbDist) continue;
        // sub-pixel refinement by correlation (:556-626), OpenCV calls as in the original
        const float uR0 = keysR[bestIdxR].pt.x, sf = inv_scales[kpL.octave];
        const float scaleduL = std::round(kpL.pt.x * sf), scaledvL = std::round(kpL.pt.y * sf), scaleduR0 = std::round(uR0 * sf);
        const int w = 5, L = 5;
        cv::Mat IL = exL.mvImagePyramid[kpL.octave].rowRange((int)(scaledvL - w), (int)(scaledvL + w + 1)).colRange((int)(scaleduL - w), (int)(scaleduL + w + 1));
        IL.convertTo(IL, CV_32F);
        IL = IL - IL.at<float>(w, w) * cv::Mat::ones(IL.rows, IL.cols, CV_32F);
        int bestSad = INT_MAX, bestinc = 0;
        std::vector<float> dists(2 * L + 1);
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
        if (iniu < 0 || endu >= exR.mvImagePyramid[kpL.octave].cols) continue;
        if (scaleduR0 - L - w < 0) continue; // the reference's guard misses this (colRange would throw); contract: unmatched
        for (int inc = -L; inc <= L; inc++) {
            cv::Mat IR = exR.mvImagePyramid[kpL.octave].rowRange((int)(scaledvL - w), (int)(scaledvL + w + 1)).colRange((int)(scaleduR0 + inc - w), (int)(scaleduR0 + inc + w + 1));
            IR.convertTo(IR, CV_32F);
            IR = IR - IR.at<float>(w, w) * cv::Mat::ones(IR.rows, IR.cols, CV_32F);
            const float d = (float)cv::norm(IL, IR, cv::NORM_L1);
            if (d < bestSad) { bestSad = (int)d; bestinc = inc; }
            dists[L + inc] = d;
        }
        if (bestinc == -L || bestinc == L) continue;
        const float d1 = dists[L + bestinc - 1], d2 = dists[L + bestinc], d3 = dists[L + bestinc + 1];
        const float deltaR = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2));
        if (deltaR < -1 || deltaR > 1) continue;
        float bestuR = scales[kpL.octave] * ((float)scaleduR0 + (float)bestinc + deltaR);
        float disparity = uL - bestuR;
        if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
            depth[iL] = mbf / disparity; uRight[iL] = bestuR; sad[iL] = bestSad;
            dist_idx.push_back(std::make_pair(bestSad, iL));
        }
    }
    if (dist_idx.empty()) return; // the reference reads element size / 2 of an empty vector; contract: no cut
    std::sort(dist_idx.begin(), dist_idx.end());
    const float median = (float)dist_idx[dist_idx.size() / 2].first, thDist = 1.5f * 1.4f * median;
    for (int i = (int)dist_idx.size() - 1; i >= 0; i--) {
        if (dist_idx[i].first < thDist) break;
        uRight[dist_idx[i].second] = -1; depth[dist_idx[i].second] = -1;
    }
}

struct Case { std::string name, left, right; int nf; float fx, bf, scale; int levels, ini, min; };

void run_case(const Case &c, const std::string &in_dir, const std::string &out_dir)
{
    const cv::Mat left = read_pgm(in_dir + "/" + c.left), right = read_pgm(in_dir + "/" + c.right);
    // Tracking's two extractor objects (src/Tracking.cc:125-128): patch 31, half patch 15, edge threshold 19
    ORB_SLAM2::ORBextractor exL(c.nf, c.scale, c.levels, c.ini, c.min, 31, 15, 19), exR(c.nf, c.scale, c.levels, c.ini, c.min, 31, 15, 19);
    std::vector<cv::KeyPoint> kl, kr;
    cv::Mat dl, dr;
    exL(left, cv::Mat(), kl, dl);
    exR(right, cv::Mat(), kr, dr);
    std::vector<float> ur, dp; std::vector<int32_t> sad;
    stereo_matches(exL, exR, kl, dl, kr, dr, c.bf, c.fx, ur, dp, sad);
    Writer w(out_dir + "/" + c.name + ".pin");
    const double params[9] = {(double)left.cols, (double)left.rows, (double)c.nf, c.fx, c.bf, c.scale, (double)c.levels, (double)c.ini, (double)c.min};
    w.put("params", F64, {9}, params);
    w.put_keys("kl", kl); w.put_keys("kr", kr);
    w.put_mat_u8("dl", dl); w.put_mat_u8("dr", dr);
    w.put_f32("u_right", ur); w.put_f32("depth", dp); w.put_i32("sad", sad);
    for (int l = 0; l < c.levels; l++) { // mvImagePyramid of the left extractor: cv::resize chain (src/ORBextractor.cc:921-946)
        const cv::Mat &pl = exL.mvImagePyramid[l];
        w.put_mat_u8("pyr" + std::to_string(l), pl);
        cv::Mat b; // the blur operator() applies before the descriptors (:899-900)
        cv::GaussianBlur(pl, b, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
        w.put_mat_u8("blur" + std::to_string(l), b);
    }
    const std::vector<float> sc = exL.GetScaleFactors(), isc = exL.GetInverseScaleFactors(), s2 = exL.GetScaleSigmaSquares(), is2 = exL.GetInverseScaleSigmaSquares();
    w.put_f32("scale", sc); w.put_f32("inv_scale", isc); w.put_f32("sigma2", s2); w.put_f32("inv_sigma2", is2);
    std::printf("%s: %zu / %zu keypoints, %d matched\n", c.name.c_str(), kl.size(), kr.size(), (int)std::count_if(ur.begin(), ur.end(), [](float v) { return v >= 0; }));
}

// ---- primitives, one record set per OPENCV-4.5.5-SEMANTICS tag of oracle/orb_oracle.c ----
cv::Mat noise_image(int w, int h, uint64_t seed, int kind)
{
    cv::Mat m(h, w, CV_8UC1);
    Lcg g(seed);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int v;
            if (kind == 0) v = (int)(g.next() & 255u);                                             // white noise: every FAST / NMS tie rule gets hit
            else if (kind == 1) v = (((x / 7) + (y / 5)) & 1) ? 200 + (int)(g.next() % 9u) : 40 + (int)(g.next() % 9u); // blocks: real corners
            else v = (x * 3 + y * 5 + (int)(g.next() % 3u)) & 255;                                 // ramps with wrap-around edges
            m.at<uint8_t>(y, x) = (uint8_t)v;
        }
    return m;
}

void dump_primitives(const std::string &out_dir)
{
    Writer w(out_dir + "/primitives.pin");
    // cvRound (A.1): ties and near-ties, float and double overloads
    {
        std::vector<float> xs; std::vector<double> xd; std::vector<int32_t> rf, rd;
        for (int i = -40; i <= 40; i++) for (float d : {-0.5f, -0.25f, 0.0f, 0.25f, 0.5f}) xs.push_back((float)i + d);
        Lcg g(11);
        for (int i = 0; i < 4000; i++) xs.push_back(((float)(g.next() % 2000001u) - 1000000.f) / 977.f);
        for (float v : xs) { rf.push_back(cvRound(v)); xd.push_back((double)v * 1.000000119); }
        for (double v : xd) rd.push_back(cvRound(v));
        w.put_f32("cvround_f_in", xs); w.put_i32("cvround_f_out", rf);
        w.put("cvround_d_in", F64, {(uint32_t)xd.size()}, xd.data()); w.put_i32("cvround_d_out", rd);
    }
    // cv::fastAtan2 (A.6)
    {
        std::vector<float> ys, xs, out;
        for (int y = -60; y <= 60; y += 3) for (int x = -60; x <= 60; x += 3) { ys.push_back((float)y); xs.push_back((float)x); }
        Lcg g(12);
        for (int i = 0; i < 20000; i++) { ys.push_back((float)((int)(g.next() % 4000001u) - 2000000)); xs.push_back((float)((int)(g.next() % 4000001u) - 2000000)); } // IC_Angle moments are integers
        for (size_t i = 0; i < ys.size(); i++) out.push_back(cv::fastAtan2(ys[i], xs[i]));
        w.put_f32("atan2_y", ys); w.put_f32("atan2_x", xs); w.put_f32("atan2_out", out);
    }
    // cv::resize INTER_LINEAR 8UC1 (A.3) with ComputePyramid's own size rule (src/ORBextractor.cc:925-934), and cv::GaussianBlur 7x7 sigma 2 (A.5)
    {
        const int sizes[][2] = {{1241, 376}, {640, 480}, {752, 480}, {1280, 720}, {97, 61}, {33, 200}};
        int idx = 0;
        for (const auto &s : sizes)
            for (int kind = 0; kind < 3; kind++, idx++) {
                cv::Mat cur = noise_image(s[0], s[1], 100 + idx, kind);
                w.put_mat_u8("resize" + std::to_string(idx) + "_l0", cur);
                float scale = 1.0f;
                for (int l = 1; l < 4; l++) {
                    scale = (float)((double)scale * (double)1.2f);
                    const float inv = 1.0f / scale;
                    const cv::Size sz(cvRound((float)s[0] * inv), cvRound((float)s[1] * inv));
                    cv::Mat nxt;
                    cv::resize(cur, nxt, sz, 0, 0, cv::INTER_LINEAR);
                    w.put_mat_u8("resize" + std::to_string(idx) + "_l" + std::to_string(l), nxt);
                    cur = nxt;
                }
                cv::Mat b;
                cv::GaussianBlur(cur, b, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
                w.put_mat_u8("blur" + std::to_string(idx), b); // of the last level above
            }
        // arbitrary (non-pyramid) ratios, as scale factors other than 1.2 produce
        cv::Mat src = noise_image(401, 203, 7, 0);
        w.put_mat_u8("resize_any_src", src);
        const int dst[][2] = {{334, 169}, {200, 101}, {381, 193}, {175, 88}, {100, 51}};
        for (int i = 0; i < 5; i++) { cv::Mat d; cv::resize(src, d, cv::Size(dst[i][0], dst[i][1]), 0, 0, cv::INTER_LINEAR); w.put_mat_u8("resize_any_" + std::to_string(i), d); }
    }
    // cv::FAST(img, kps, t, true) (A.4): x, y, response in emission order, on cell-sized and larger images, both thresholds of the reference
    {
        const int sizes[][2] = {{36, 36}, {37, 41}, {7, 7}, {8, 30}, {120, 90}, {346, 105}};
        int idx = 0;
        for (const auto &s : sizes)
            for (int kind = 0; kind < 3; kind++, idx++) {
                const cv::Mat img = noise_image(s[0], s[1], 500 + idx, kind);
                w.put_mat_u8("fast" + std::to_string(idx) + "_img", img);
                for (int t : {20, 7}) {
                    std::vector<cv::KeyPoint> k;
                    cv::FAST(img, k, t, true);
                    std::vector<int32_t> rec;
                    for (const cv::KeyPoint &p : k) { rec.push_back((int32_t)p.pt.x); rec.push_back((int32_t)p.pt.y); rec.push_back((int32_t)p.response); }
                    w.put("fast" + std::to_string(idx) + "_t" + std::to_string(t), I32, {(uint32_t)k.size(), 3u}, rec.data());
                }
            }
    }
    // cv::cvtColor -> GRAY, the four codes Tracking::GrabImage* use (src/Tracking.cc:269-351)
    {
        cv::Mat rgb(48, 64, CV_8UC3), rgba(48, 64, CV_8UC4);
        Lcg g(21);
        for (int y = 0; y < 48; y++) for (int x = 0; x < 64; x++) {
            for (int c = 0; c < 3; c++) rgb.at<cv::Vec3b>(y, x)[c] = (uint8_t)(g.next() & 255u);
            for (int c = 0; c < 4; c++) rgba.at<cv::Vec4b>(y, x)[c] = (uint8_t)(g.next() & 255u);
        }
        w.put("cvt_rgb_in", U8, {48, 64, 3}, rgb.data); w.put("cvt_rgba_in", U8, {48, 64, 4}, rgba.data);
        cv::Mat o;
        cv::cvtColor(rgb, o, cv::COLOR_RGB2GRAY); w.put_mat_u8("cvt_rgb2gray", o);
        cv::cvtColor(rgb, o, cv::COLOR_BGR2GRAY); w.put_mat_u8("cvt_bgr2gray", o);
        cv::cvtColor(rgba, o, cv::COLOR_RGBA2GRAY); w.put_mat_u8("cvt_rgba2gray", o);
        cv::cvtColor(rgba, o, cv::COLOR_BGRA2GRAY); w.put_mat_u8("cvt_bgra2gray", o);
    }
    // cv::remap INTER_LINEAR with CV_32FC1 maps (Test/Replay/Stereo/stereo_euroc.cc:136-137)
    {
        const cv::Mat src = noise_image(64, 48, 31, 1);
        cv::Mat mx(48, 64, CV_32FC1), my(48, 64, CV_32FC1);
        for (int y = 0; y < 48; y++) for (int x = 0; x < 64; x++) { mx.at<float>(y, x) = (float)x * 0.97f + 0.013f * (float)y + 1.37f; my.at<float>(y, x) = (float)y * 1.02f - 0.011f * (float)x - 0.81f; }
        cv::Mat o;
        cv::remap(src, o, mx, my, cv::INTER_LINEAR);
        w.put_mat_u8("remap_src", src); w.put("remap_mx", F32, {48, 64}, mx.data); w.put("remap_my", F32, {48, 64}, my.data); w.put_mat_u8("remap_out", o);
    }
    // cv::undistortPoints(src, dst, K, D, noArray(), K) (src/Frame.cc:420,446)
    {
        const float fx = 517.3f, fy = 516.5f, cx = 318.6f, cy = 255.3f;
        cv::Mat K = cv::Mat::eye(3, 3, CV_32F);
        K.at<float>(0, 0) = fx; K.at<float>(1, 1) = fy; K.at<float>(0, 2) = cx; K.at<float>(1, 2) = cy;
        const float d5[5] = {0.262383f, -0.953104f, -0.005358f, 0.002628f, 1.163314f};
        for (int nd : {4, 5}) {
            cv::Mat D(nd, 1, CV_32F);
            for (int i = 0; i < nd; i++) D.at<float>(i) = d5[i];
            cv::Mat pts(200, 2, CV_32F);
            Lcg g(41);
            for (int i = 0; i < 200; i++) { pts.at<float>(i, 0) = (float)(g.next() % 660001u) / 1000.f - 10.f; pts.at<float>(i, 1) = (float)(g.next() % 500001u) / 1000.f - 10.f; }
            w.put("undist" + std::to_string(nd) + "_in", F32, {200, 2}, pts.data);
            cv::Mat m = pts.reshape(2);
            cv::undistortPoints(m, m, K, D, cv::Mat(), K);
            m = m.reshape(1);
            w.put("undist" + std::to_string(nd) + "_out", F32, {200, 2}, m.data);
        }
    }
    std::printf("primitives written (OpenCV %s)\n", CV_VERSION);
}
} // namespace

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: pin <inputs dir> <output dir>\n"); return 1; }
    const std::string in_dir = argv[1], out_dir = argv[2];
    std::ifstream man(in_dir + "/cases.txt"); // name left.pgm right.pgm nfeatures fx bf scale levels iniTh minTh
    if (!man) { std::fprintf(stderr, "pin: %s/cases.txt missing (run tools/opencv_pin/export_inputs.py)\n", in_dir.c_str()); return 2; }
    Case c;
    int n = 0;
    while (man >> c.name >> c.left >> c.right >> c.nf >> c.fx >> c.bf >> c.scale >> c.levels >> c.ini >> c.min) { run_case(c, in_dir, out_dir); n++; }
    dump_primitives(out_dir);
    std::ofstream ver(out_dir + "/opencv_version.txt");
    ver << CV_VERSION << "\n";
    std::printf("%d stereo cases pinned\n", n);
    return 0;
}
