// tests/compat_stub/compat_selftest.cpp -- TEST-ONLY driver: calls orbslam2_amd/compat/ORBmatcher.{h,cc} exactly the way
// src/Tracking.cc / src/LocalMapping.cc do (reference signatures: Frame&, KeyFrame*, vector<MapPoint*>&), on a scene that
// tests/test_compat_matcher.py wrote as raw arrays, and dumps what the calls left in the Frame / KeyFrame / MapPoint objects.
// The pytest compares that with the CPU oracle.  Frame / KeyFrame / MapPoint / cv::Mat are the declaration stand-ins of this
// directory (OpenCV and the reference tree are absent on the GPU box).
//   usage: compat_selftest <scene_dir>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "ORBmatcher.h" // orbslam2_amd/compat

using namespace ORB_SLAM2;

// Frame's statics and members are defined by orbslam2_amd/compat/Frame.cc (linked in); this driver fills Frame objects by hand
// because its scenes are keypoint-level (tests/compat_stub/frame_selftest.cpp is the one that constructs Frames from images)
static float g_bf = 0.f;
std::mutex MapPoint::mGlobalMutex; // include/MapPoint.h:90; src/MapPoint.cc defines it in the reference

static std::string g_dir;
template <class T> static std::vector<T> rd(const char *name)
{
    std::ifstream f(g_dir + "/" + name, std::ios::binary | std::ios::ate);
    if (!f) { std::cerr << "missing " << name << "\n"; std::exit(2); }
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<T> v((size_t)n / sizeof(T));
    f.read((char *)v.data(), n);
    return v;
}
template <class T> static void wr(const char *name, const std::vector<T> &v)
{
    std::ofstream f(g_dir + "/" + name, std::ios::binary);
    f.write((const char *)v.data(), (std::streamsize)(v.size() * sizeof(T)));
}

static cv::Mat mat_f32(int r, int c, const float *src) { cv::Mat m(r, c, CV_32F); for (int i = 0; i < r * c; i++) m.ptr<float>(0)[i] = src[i]; return m; }
static cv::Mat pose4x4(const float *T34)
{
    cv::Mat m(4, 4, CV_32F);
    for (int i = 0; i < 12; i++) m.ptr<float>(0)[i] = T34[i];
    m.at<float>(3, 3) = 1.f;
    return m;
}

static void fill_frame(Frame &F, ORBextractor *ex, const std::vector<cv::KeyPoint> &k, const std::vector<uchar> &d, const std::vector<float> &ur, const float *T34)
{
    F.mpORBextractorLeft = ex; F.mpORBextractorRight = NULL;
    F.mbf = g_bf; F.mb = g_bf / Frame::fx; F.mThDepth = 35.f * F.mb;
    F.N = (int)k.size();
    F.mvKeys = k; F.mvKeysUn = k;
    F.mvuRight = ur.empty() ? std::vector<float>(k.size(), -1.f) : ur;
    F.mvDepth.assign(k.size(), -1.f);
    F.mDescriptors = cv::Mat((int)k.size(), 32, CV_8U);
    if (!d.empty()) std::memcpy(F.mDescriptors.ptr<uchar>(0), d.data(), d.size());
    F.mvpMapPoints.assign(k.size(), static_cast<MapPoint *>(NULL));
    F.mvbOutlier.assign(k.size(), false);
    F.mTcw = pose4x4(T34);
}

static fbow::fBow2 featvec(const char *nodes, const char *off, const char *feat)
{
    const std::vector<uint32_t> n = rd<uint32_t>(nodes);
    const std::vector<int32_t> o = rd<int32_t>(off), f = rd<int32_t>(feat);
    fbow::fBow2 m;
    for (size_t k = 0; k < n.size(); k++)
        for (int j = o[k]; j < o[k + 1]; j++) m[n[k]].push_back((uint32_t)f[j]);
    return m;
}

int main(int argc, char **argv)
{
    if (argc != 2 && argc != 3) { std::cerr << "usage: compat_selftest <scene_dir> [stress]\n"; return 2; }
    g_dir = argv[1];
    try {
        const std::vector<float> cam = rd<float>("cam.f32"), bounds = rd<float>("bounds.f32");
        Frame::fx = cam[0]; Frame::fy = cam[1]; Frame::cx = cam[2]; Frame::cy = cam[3]; Frame::invfx = 1.f / cam[0]; Frame::invfy = 1.f / cam[1];
        g_bf = cam[4];
        Frame::mnMinX = bounds[0]; Frame::mnMaxX = bounds[1]; Frame::mnMinY = bounds[2]; Frame::mnMaxY = bounds[3];
        const int W = (int)cam[5], H = (int)cam[6];
        // same construction as src/Tracking.cc:125; the context exists before the first frame (BindImageSize)
        ORBextractor extractor(2000, 1.2f, 8, 20, 7, 31, 15, 19);
        CameraParams cp; cp.fx = cam[0]; cp.fy = cam[1]; cp.cx = cam[2]; cp.cy = cam[3]; cp.bf = cam[4];
        extractor.SetCamera(cp);
        extractor.BindImageSize(W, H);

        const std::vector<cv::KeyPoint> cur_k = rd<cv::KeyPoint>("cur_k.bin");
        const std::vector<uchar> cur_d = rd<uchar>("cur_d.bin"), cur_has = rd<uchar>("cur_has_obs.bin");
        const std::vector<float> cur_ur = rd<float>("cur_ur.bin");
        const std::vector<float> pos = rd<float>("last_pos.bin"), ang = rd<float>("last_ang.bin"), T_last = rd<float>("T_last.bin"), T_cur = rd<float>("T_cur.bin");
        const std::vector<float> normal = rd<float>("normal.bin"), max_d = rd<float>("max_d.bin"), min_d = rd<float>("min_d.bin");
        const std::vector<int32_t> oct = rd<int32_t>("last_oct.bin"), valid = rd<int32_t>("last_valid.bin"), obs = rd<int32_t>("last_obs.bin");
        const std::vector<uchar> last_desc = rd<uchar>("last_desc.bin");
        const int M = (int)oct.size(), N = (int)cur_k.size();

        // the map: one MapPoint per last-frame keypoint
        std::vector<std::unique_ptr<MapPoint> > store;
        std::map<MapPoint *, int> index_of;
        auto make_points = [&]() {
            std::vector<MapPoint *> v(M);
            for (int i = 0; i < M; i++) {
                store.emplace_back(new MapPoint());
                MapPoint *p = store.back().get();
                p->mWorldPos = mat_f32(3, 1, &pos[3 * i]);
                p->mNormalVector = mat_f32(3, 1, &normal[3 * i]);
                p->mDescriptor = cv::Mat(1, 32, CV_8U);
                std::memcpy(p->mDescriptor.ptr<uchar>(0), &last_desc[32 * i], 32);
                p->nObs = obs[i];
                p->mfMaxDistance = max_d[i]; p->mfMinDistance = min_d[i];
                index_of[p] = i;
                v[i] = p;
            }
            return v;
        };
        MapPoint dummy_obs; dummy_obs.nObs = 1; // "keypoint already holds a point with observations"
        auto dump_matches = [&](const Frame &F, const char *name) {
            std::vector<int32_t> out(F.N, -1);
            for (int k = 0; k < F.N; k++) {
                MapPoint *p = F.mvpMapPoints[k];
                if (p && p != &dummy_obs) out[k] = index_of.at(p);
            }
            wr(name, out);
        };

        // ---- 1. SearchByProjection(CurrentFrame, LastFrame, th, bMono): src/Tracking.cc:969-992
        {
            std::vector<MapPoint *> pts = make_points();
            Frame Last, Cur;
            std::vector<cv::KeyPoint> lk(M);
            for (int i = 0; i < M; i++) { lk[i].octave = oct[i]; lk[i].angle = ang[i]; }
            fill_frame(Last, &extractor, lk, std::vector<uchar>(), std::vector<float>(), T_last.data());
            for (int i = 0; i < M; i++) {
                Last.mvpMapPoints[i] = valid[i] ? pts[i] : NULL;
                if (valid[i] == 2) { Last.mvpMapPoints[i] = pts[i]; Last.mvbOutlier[i] = true; } // a point flagged as outlier is skipped too
            }
            fill_frame(Cur, &extractor, cur_k, cur_d, cur_ur, T_cur.data());
            for (int k = 0; k < N; k++) if (cur_has[k]) Cur.mvpMapPoints[k] = &dummy_obs;
            ORBmatcher matcher(0.9, true);
            const int th = 7;
            const int nmatches = matcher.SearchByProjection(Cur, Last, th, false);
            dump_matches(Cur, "out_last.bin");
            wr("n_last.bin", std::vector<int32_t>(1, nmatches));
        }
        // ---- 2. SearchByProjection(F, vpMapPoints, th): src/Tracking.cc:1283-1290 (SearchLocalPoints)
        {
            std::vector<MapPoint *> pts = make_points();
            const std::vector<orbfe_track_point> tp = rd<orbfe_track_point>("tp.bin"); // what Frame::isInFrustum left in the points
            for (int i = 0; i < M; i++) {
                pts[i]->mbTrackInView = tp[i].in_view != 0;
                pts[i]->mTrackProjX = tp[i].proj_x; pts[i]->mTrackProjY = tp[i].proj_y; pts[i]->mTrackProjXR = tp[i].proj_xr;
                pts[i]->mnTrackScaleLevel = tp[i].level; pts[i]->mTrackViewCos = tp[i].view_cos;
            }
            Frame Cur;
            fill_frame(Cur, &extractor, cur_k, cur_d, cur_ur, T_cur.data());
            for (int k = 0; k < N; k++) if (cur_has[k]) Cur.mvpMapPoints[k] = &dummy_obs;
            ORBmatcher matcher(0.8);
            const int nmatches = matcher.SearchByProjection(Cur, pts, 3);
            dump_matches(Cur, "out_pts.bin");
            wr("n_pts.bin", std::vector<int32_t>(1, nmatches));
        }
        // ---- 3. SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist): src/Tracking.cc:1552 (Relocalization)
        {
            std::vector<MapPoint *> pts = make_points();
            const std::vector<uchar> found = rd<uchar>("already_found.bin");
            Frame KFsrc;
            std::vector<cv::KeyPoint> lk(M);
            for (int i = 0; i < M; i++) { lk[i].octave = oct[i]; lk[i].angle = ang[i]; }
            fill_frame(KFsrc, &extractor, lk, last_desc, std::vector<float>(), T_last.data());
            std::set<MapPoint *> sAlreadyFound;
            for (int i = 0; i < M; i++) {
                KFsrc.mvpMapPoints[i] = valid[i] ? pts[i] : NULL;
                if (valid[i] == 2) pts[i]->mbBad = true;
                if (found[i]) sAlreadyFound.insert(pts[i]);
            }
            KeyFrame KF(KFsrc);
            Frame Cur;
            fill_frame(Cur, &extractor, cur_k, cur_d, std::vector<float>(), T_cur.data());
            for (int k = 0; k < N; k++) if (cur_has[k]) Cur.mvpMapPoints[k] = &dummy_obs;
            ORBmatcher matcher2(0.9, true);
            const int nmatches = matcher2.SearchByProjection(Cur, &KF, sAlreadyFound, 10, 100);
            dump_matches(Cur, "out_kf.bin");
            wr("n_kf.bin", std::vector<int32_t>(1, nmatches));
        }
        // ---- 4. SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, 100): src/Tracking.cc:698-699
        {
            const std::vector<cv::KeyPoint> k1 = rd<cv::KeyPoint>("init_k1.bin"), k2 = rd<cv::KeyPoint>("init_k2.bin");
            const std::vector<uchar> d1 = rd<uchar>("init_d1.bin"), d2 = rd<uchar>("init_d2.bin");
            Frame F1, F2;
            fill_frame(F1, &extractor, k1, d1, std::vector<float>(), T_last.data());
            fill_frame(F2, &extractor, k2, d2, std::vector<float>(), T_cur.data());
            std::vector<cv::Point2f> vbPrevMatched(k1.size());
            for (size_t i = 0; i < k1.size(); i++) vbPrevMatched[i] = k1[i].pt; // src/Tracking.cc:660-662
            std::vector<int> vnMatches12;
            ORBmatcher matcher(0.9, true);
            const int nmatches = matcher.SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, 100);
            wr("out_init.bin", std::vector<int32_t>(vnMatches12.begin(), vnMatches12.end()));
            std::vector<float> pm;
            for (size_t i = 0; i < vbPrevMatched.size(); i++) { pm.push_back(vbPrevMatched[i].x); pm.push_back(vbPrevMatched[i].y); }
            wr("out_prev.bin", pm);
            wr("n_init.bin", std::vector<int32_t>(1, nmatches));
            // ---- 5. SearchByFboW(pKF, F, vpMapPointMatches): src/Tracking.cc:862-867 (TrackReferenceKeyFrame), same two frames
            std::vector<MapPoint *> pts = make_points();
            const std::vector<int32_t> kf_valid = rd<int32_t>("bow_kf_valid.bin");
            for (size_t i = 0; i < k1.size(); i++) {
                if (kf_valid[i] == 0) continue;
                F1.mvpMapPoints[i] = pts[i % M];
                if (kf_valid[i] == 2) { store.emplace_back(new MapPoint()); store.back()->mbBad = true; F1.mvpMapPoints[i] = store.back().get(); index_of[store.back().get()] = -7; }
            }
            F1.mFbowFeatVec = featvec("bow_kf_nodes.bin", "bow_kf_off.bin", "bow_kf_feat.bin");
            F2.mFbowFeatVec = featvec("bow_f_nodes.bin", "bow_f_off.bin", "bow_f_feat.bin");
            KeyFrame KF(F1);
            std::vector<MapPoint *> vpMapPointMatches;
            ORBmatcher matcher3(0.7, true);
            const int nbow = matcher3.SearchByFboW(&KF, F2, vpMapPointMatches);
            std::vector<int32_t> out(F2.N, -1); // KF keypoint whose point frame keypoint j received
            for (int j = 0; j < F2.N; j++) {
                if (!vpMapPointMatches[j]) continue;
                for (size_t i = 0; i < k1.size(); i++)
                    if (F1.mvpMapPoints[i] == vpMapPointMatches[j]) { out[j] = (int32_t)i; break; }
            }
            wr("out_bow.bin", out);
            wr("n_bow.bin", std::vector<int32_t>(1, nbow));
        }
        // ---- 6. Fuse(pKF, vpMapPoints, th): src/LocalMapping.cc:482-512 (SearchInNeighbors), with the map mutation
        {
            std::vector<MapPoint *> pts = make_points();
            for (int i = 0; i < M; i++) if (valid[i] == 0) pts[i] = NULL; else if (valid[i] == 2) pts[i]->mbBad = true;
            Frame src;
            fill_frame(src, &extractor, cur_k, cur_d, cur_ur, T_cur.data());
            const std::vector<int32_t> kf_obs = rd<int32_t>("fuse_kf_obs.bin"); // -1: keypoint holds no point, else Observations() of the point it holds
            std::vector<MapPoint *> held(N, static_cast<MapPoint *>(NULL));
            for (int k = 0; k < N; k++)
                if (kf_obs[k] >= 0) { store.emplace_back(new MapPoint()); held[k] = store.back().get(); held[k]->nObs = kf_obs[k]; src.mvpMapPoints[k] = held[k]; }
            KeyFrame KF(src);
            ORBmatcher matcher;
            const int nFused = matcher.Fuse(&KF, pts, 3.0);
            std::vector<int32_t> added(M, -1);
            std::vector<uchar> pt_bad(M, 0), held_bad(N, 0);
            std::vector<int32_t> pt_replaced_by_held(M, -1);
            for (int i = 0; i < M; i++) {
                if (!pts[i]) continue;
                added[i] = pts[i]->GetIndexInKeyFrame(&KF);
                pt_bad[i] = pts[i]->isBad() && valid[i] != 2;
                if (pts[i]->GetReplaced())
                    for (int k = 0; k < N; k++) if (held[k] == pts[i]->GetReplaced()) pt_replaced_by_held[i] = k;
            }
            std::vector<int32_t> held_replaced_by_pt(N, -1);
            for (int k = 0; k < N; k++)
                if (held[k]) { held_bad[k] = held[k]->isBad(); if (held[k]->GetReplaced()) held_replaced_by_pt[k] = index_of.at(held[k]->GetReplaced()); }
            wr("out_fuse_added.bin", added); wr("out_fuse_pt_replaced.bin", pt_replaced_by_held); wr("out_fuse_held_replaced.bin", held_replaced_by_pt);
            wr("n_fuse.bin", std::vector<int32_t>(1, nFused));
        }
        // ---- 7. the RESIDENT path: a real extraction (ORBextractor::operator() of the mirror), a Frame built from its output, and
        //         SearchByProjection(F, vpMapPoints, th) on it: the shim must recognise the frame as the extractor's latest one
        //         (keypoints / descriptors read in HBM, grid built once) and a second call must reuse the grid
        {
            const std::vector<uchar> img = rd<uchar>("rendered_image.bin");
            std::vector<orbfe_keypoint> rk;
            std::vector<uint8_t> rd_;
            extractor(ImageView{img.data(), W, H, (size_t)W}, rk, rd_);
            std::vector<cv::KeyPoint> ck(rk.size());
            std::memcpy((void *)ck.data(), rk.data(), rk.size() * sizeof(orbfe_keypoint));
            Frame F;
            fill_frame(F, &extractor, ck, rd_, std::vector<float>(), T_cur.data());
            if (!extractor.IsResidentFrame(F.N, F.mDescriptors.ptr<uchar>(0))) { std::cerr << "extracted frame not recognised as resident\n"; return 3; }
            Frame other; // a frame of the scene is NOT the resident one
            fill_frame(other, &extractor, cur_k, cur_d, cur_ur, T_cur.data());
            if (extractor.IsResidentFrame(other.N, other.mDescriptors.ptr<uchar>(0))) { std::cerr << "foreign frame taken for the resident one\n"; return 4; }
            // map points = the frame's own keypoints seen again (descriptor of the keypoint, predicted level = its octave)
            std::vector<std::unique_ptr<MapPoint> > own;
            std::vector<MapPoint *> pts;
            for (int i = 0; i < F.N; i += 2) {
                own.emplace_back(new MapPoint());
                MapPoint *p = own.back().get();
                p->mDescriptor = cv::Mat(1, 32, CV_8U);
                std::memcpy(p->mDescriptor.ptr<uchar>(0), &rd_[(size_t)32 * i], 32);
                p->nObs = 1; p->mbTrackInView = true;
                p->mTrackProjX = ck[i].pt.x + 1.5f; p->mTrackProjY = ck[i].pt.y - 1.0f; p->mTrackProjXR = -1.f;
                p->mnTrackScaleLevel = ck[i].octave; p->mTrackViewCos = 0.9f;
                index_of[p] = i;
                pts.push_back(p);
            }
            ORBmatcher matcher(0.8);
            const int n1 = matcher.SearchByProjection(F, pts, 3);
            dump_matches(F, "out_resident.bin");
            F.mvpMapPoints.assign(F.N, static_cast<MapPoint *>(NULL));
            const int n2 = matcher.SearchByProjection(F, pts, 3); // same frame again: cached grid
            dump_matches(F, "out_resident2.bin");
            wr("n_resident.bin", std::vector<int32_t>{n1, n2, F.N});
            wr("resident_k.bin", rk); wr("resident_d.bin", rd_);
        }
        // ---- 8. (argv[2] == "stress") two threads on ONE device context, as the reference's threads are: Tracking's
        //         SearchByProjection(F, vpMapPoints, th) (src/Tracking.cc:1290) against LocalMapping's Fuse(pKF, vpMapPoints)
        //         (src/LocalMapping.cc:512), each constructing its own ORBmatcher; every iteration must equal the serial result
        if (argc > 2 && std::string(argv[2]) == "stress") {
            const std::vector<orbfe_track_point> tp = rd<orbfe_track_point>("tp.bin");
            const std::vector<int32_t> kf_obs = rd<int32_t>("fuse_kf_obs.bin");
            struct Local { // a private map per call: the shim mutates MapPoints
                std::vector<std::unique_ptr<MapPoint> > own;
                std::vector<MapPoint *> pts;
            };
            auto local_points = [&](Local &L) {
                L.pts.assign(M, NULL);
                for (int i = 0; i < M; i++) {
                    L.own.emplace_back(new MapPoint());
                    MapPoint *p = L.own.back().get();
                    p->mWorldPos = mat_f32(3, 1, &pos[3 * i]); p->mNormalVector = mat_f32(3, 1, &normal[3 * i]);
                    p->mDescriptor = cv::Mat(1, 32, CV_8U);
                    std::memcpy(p->mDescriptor.ptr<uchar>(0), &last_desc[32 * i], 32);
                    p->nObs = obs[i]; p->mfMaxDistance = max_d[i]; p->mfMinDistance = min_d[i];
                    L.pts[i] = p;
                }
            };
            auto track_once = [&]() {
                Local L; local_points(L);
                for (int i = 0; i < M; i++) {
                    L.pts[i]->mbTrackInView = tp[i].in_view != 0;
                    L.pts[i]->mTrackProjX = tp[i].proj_x; L.pts[i]->mTrackProjY = tp[i].proj_y; L.pts[i]->mTrackProjXR = tp[i].proj_xr;
                    L.pts[i]->mnTrackScaleLevel = tp[i].level; L.pts[i]->mTrackViewCos = tp[i].view_cos;
                }
                Frame Cur;
                fill_frame(Cur, &extractor, cur_k, cur_d, cur_ur, T_cur.data());
                ORBmatcher matcher(0.8);
                matcher.SearchByProjection(Cur, L.pts, 3);
                std::vector<int32_t> out(N, -1);
                for (int k = 0; k < N; k++)
                    if (Cur.mvpMapPoints[k]) for (int i = 0; i < M; i++) if (L.pts[i] == Cur.mvpMapPoints[k]) { out[k] = i; break; }
                return out;
            };
            auto fuse_once = [&]() {
                Local L; local_points(L);
                for (int i = 0; i < M; i++) if (valid[i] == 0) L.pts[i] = NULL; else if (valid[i] == 2) L.pts[i]->mbBad = true;
                Frame src;
                fill_frame(src, &extractor, cur_k, cur_d, cur_ur, T_cur.data());
                std::vector<std::unique_ptr<MapPoint> > held;
                for (int k = 0; k < N; k++)
                    if (kf_obs[k] >= 0) { held.emplace_back(new MapPoint()); held.back()->nObs = kf_obs[k]; src.mvpMapPoints[k] = held.back().get(); }
                KeyFrame KF(src);
                ORBmatcher matcher;
                matcher.Fuse(&KF, L.pts, 3.0);
                std::vector<int32_t> out(M, -2);
                for (int i = 0; i < M; i++) if (L.pts[i]) out[i] = L.pts[i]->isBad() ? -3 : L.pts[i]->GetIndexInKeyFrame(&KF);
                return out;
            };
            const std::vector<int32_t> track_ref = track_once(), fuse_ref = fuse_once();
            const int IT = 60;
            int bad_track = 0, bad_fuse = 0;
            std::string err_a, err_b;
            std::thread a([&]() { try { for (int it = 0; it < IT; it++) bad_track += track_once() != track_ref; } catch (const std::exception &e) { err_a = e.what(); } });
            std::thread b([&]() { try { for (int it = 0; it < IT; it++) bad_fuse += fuse_once() != fuse_ref; } catch (const std::exception &e) { err_b = e.what(); } });
            a.join(); b.join();
            wr("stress.bin", std::vector<int32_t>{bad_track, bad_fuse, IT});
            if (!err_a.empty() || !err_b.empty()) { std::cerr << "stress: " << err_a << " | " << err_b << "\n"; return 6; }
            if (bad_track || bad_fuse) { std::cerr << "stress: " << bad_track << " tracking and " << bad_fuse << " fuse results differ from the serial ones\n"; return 5; }
        }
        std::printf("compat selftest ok\n");
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
