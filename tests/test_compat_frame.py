"""Reference-signature Frame (orbslam2_amd/compat/Frame.cc = src/Frame.cc over the C ABI) and the reference's threading contract.

CPU: the stand-in tests/compat_stub/Frame.h declares include/Frame.h's members with the reference's signatures, and compat/Frame.cc
(which DEFINES them) compiles against it with -Werror -- including, for the first time, the cv::InputArray / OutputArray call
operator of orbslam2_amd/host/ORBextractor.h (ORBFE_WITH_OPENCV) that Frame::ExtractORB goes through.
GPU: tests/compat_stub/frame_selftest constructs Frames exactly as src/Tracking.cc:296 / :326 / :354-358 do and every member the
reference's constructors fill must equal the CPU oracle; `threads` runs the two extractor objects on two fresh std::threads per
frame as src/Frame.cc:78-81 does, 200 frames, against the serial result."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import literal_matchers as LM
from oracle import oracle as O
from orbslam2_amd import bow as B
from orbslam2_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "compat_stub")
EXE = os.path.join(STUB, "frame_selftest")
SRC = os.path.join(ROOT, "orbslam2_amd", "compat", "Frame.cc")
REF_DECLS = [  # include/Frame.h:45-110,197-205 of the reference, whitespace-normalised
    "Frame();",
    "Frame(const Frame &frame);",
    "Frame(const cv::Mat &imLeft, const cv::Mat &imRight, const double &timeStamp, ORBextractor* extractorLeft, ORBextractor* extractorRight, fbow::Vocabulary* voc, cv::Mat &K, cv::Mat &distCoef, const float &bf, const float &thDepth);",
    "Frame(const cv::Mat &imGray, const cv::Mat &imDepth, const double &timeStamp, ORBextractor* extractor, fbow::Vocabulary* voc, cv::Mat &K, cv::Mat &distCoef, const float &bf, const float &thDepth);",
    "Frame(const cv::Mat &imGray, const double &timeStamp, ORBextractor* extractor, fbow::Vocabulary* voc, cv::Mat &K, cv::Mat &distCoef, const float &bf, const float &thDepth);",
    "void ExtractORB(int flag, const cv::Mat &im);",
    "void ComputeFboW();",
    "void SetPose(cv::Mat Tcw);",
    "void UpdatePoseMatrices();",
    "bool isInFrustum(MapPoint* pMP, float viewingCosLimit);",
    "bool PosInGrid(const cv::KeyPoint &kp, int &posX, int &posY);",
    "std::vector<size_t> GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel=-1, const int maxLevel=-1) const;",
    "void ComputeStereoMatches();",
    "void ComputeStereoFromRGBD(const cv::Mat &imDepth);",
    "cv::Mat UnprojectStereo(const int &i);",
    "void UndistortKeyPoints();",
    "void ComputeImageBounds(const cv::Mat &imLeft);",
    "void AssignFeaturesToGrid();",
    "std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];",
]
TUM1_DIST = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]  # Config/RGB-D-TUM1.yaml


def _norm(decl):
    return re.sub(r"\s+", "", decl.replace("std::", ""))


def test_stub_declares_the_reference_frame():
    body = _norm(re.sub(r"//[^\n]*", "", open(os.path.join(STUB, "Frame.h")).read()))
    for d in REF_DECLS:
        assert _norm(d) in body, d


def test_frame_shim_compiles_and_only_forwards():
    r = subprocess.run(["g++", "-std=c++14", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", STUB, "-I", os.path.dirname(SRC), SRC],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = open(SRC).read()
    for call in ("orbfe_stereo_frame(", "orbfe_rgbd_frame(", "orbfe_assign_features_to_grid(", "orbfe_bow_transform(", "orbfe_bow_maps(", "orbfe_vocab_load(",
                 "orbfe_fetch_keys_un(", "orbfe_undistort_keypoints(", "orbfe_image_bounds(", "orbfe_is_in_frustum(", "orbfe_features_in_area(", "orbfe_get_camera("):
        assert call in text, call
    assert "std::thread" not in text and "oracle" not in text  # one fused device call instead of src/Frame.cc:78-81's two threads
    # the cv::InputArray / cv::OutputArray call operator is compiled by this translation unit (Frame::ExtractORB uses it)
    assert "ORBFE_WITH_OPENCV" in open(os.path.join(STUB, "ORBextractor.h")).read()
    assert "(*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors)" in text


# ---------------------------------------------------------------------------------------------------------------- GPU
def _run(d, mode, files):
    assert os.path.exists(EXE), "frame_selftest not built (make -C tests/compat_stub)"
    for name, a in files.items():
        if isinstance(a, bytes):
            (d / name).write_bytes(a)
        else:
            np.ascontiguousarray(a).tofile(d / name)
    r = subprocess.run([EXE, str(d), mode], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "frame selftest ok" in r.stdout, r.stdout + r.stderr


def _vocab():
    rng = np.random.default_rng(5)
    protos = rng.integers(0, 256, (40, 32)).astype(np.uint8)
    train = protos[rng.integers(0, 40, 6000)] ^ np.packbits(rng.random((6000, 256)) < 0.12, axis=1, bitorder="little")
    return B.build_vocabulary(train, k=10, levels=5, seed=7)


def _kps(d, name):
    return np.fromfile(d / name, O.KP_DTYPE)


def _same_kps(got, ref, what):
    assert len(got) == len(ref), what
    assert got.tobytes() == np.ascontiguousarray(ref).tobytes(), what


def _grid_of(d, tag):
    off = np.fromfile(d / (tag + "_grid_off.bin"), np.int32)
    idx = np.fromfile(d / (tag + "_grid_idx.bin"), np.int32)
    return [[idx[off[i * 48 + j]: off[i * 48 + j + 1]].tolist() for j in range(48)] for i in range(64)]


def _check_common(d, tag, cfg, k_ref, kun_ref, bounds, frame_id, sf, inv_s2):
    """statics, scale tables, grid, bookkeeping of one dumped frame"""
    n = len(k_ref)
    meta = np.fromfile(d / (tag + "_meta.bin"), np.int32)
    assert meta.tolist() == [n, frame_id, 8, n, n, 0]  # N, mnId, mnScaleLevels, |mvpMapPoints|, |mvbOutlier|, mbInitialComputations
    st = np.fromfile(d / (tag + "_statics.bin"), np.float32)
    f32 = np.float32
    fx, fy, cx, cy, bf = (f32(cfg[k]) for k in ("fx", "fy", "cx", "cy", "bf"))
    assert st[:4].tolist() == [fx, fy, cx, cy] and st[4] == f32(1) / fx and st[5] == f32(1) / fy
    assert np.array_equal(st[6:10], np.asarray(bounds, np.float32))
    assert st[10] == f32(64) / (f32(bounds[1]) - f32(bounds[0])) and st[11] == f32(48) / (f32(bounds[3]) - f32(bounds[2]))
    assert st[12] == bf and st[13] == bf / fx and st[14] == f32(cfg["th_depth"])
    assert st[15] == f32(1.2) and st[16] == f32(np.log(np.float64(f32(1.2))))
    assert np.array_equal(st[17:25], sf) and np.array_equal(st[25:33], inv_s2)
    lit = LM.Frame(kun_ref, None, None, bounds, (fx, fy, cx, cy, bf, bf / fx), sf)
    assert _grid_of(d, tag) == lit.mGrid  # Frame::AssignFeaturesToGrid, literal restatement
    return lit


@pytest.mark.gpu
def test_stereo_frames_built_like_tracking_match_the_oracle(tmp_path):
    """Frame(imLeft, imRight, ts, exL, exR, voc, K, distCoef, bf, thDepth), src/Tracking.cc:296 -- EuRoC-sized, 1200 features"""
    cfg = dict(width=752, height=480, nfeatures=1200, fx=458.654, fy=457.296, cx=367.215, cy=248.375, bf=47.9, th_depth=35.0 * 47.9 / 458.654)
    w, h, nf = cfg["width"], cfg["height"], cfg["nfeatures"]
    pairs = [synth.stereo_pair(w, h, seed=211 + t) for t in range(2)]
    blob = _vocab()
    files = {"cam.f32": np.array([cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], cfg["bf"], w, h, nf, cfg["th_depth"]], np.float32),
             "dist.f32": np.zeros(4, np.float32), "vocab.fbow": blob}
    for t, (l, r) in enumerate(pairs):
        files["left%d.bin" % t] = l; files["right%d.bin" % t] = r
    _run(tmp_path, "stereo", files)
    d = tmp_path
    Lb, vb = None, None
    from tests.test_bow import _oracle_transform, _oracle_voc
    Lb, vb = _oracle_voc(blob)
    cam = O.Camera(cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], cfg["bf"], np.float32(cfg["bf"]) / np.float32(cfg["fx"]))
    bounds = (0.0, float(w), 0.0, float(h))
    for t, (l, r) in enumerate(pairs):
        tag = "f%d" % t
        exl, exr = O.Extractor(nfeatures=nf), O.Extractor(nfeatures=nf)
        kl, dl = exl.extract(l)
        kr, dr = exr.extract(r)
        ur, dp, m = O.stereo_matches(exl, exr, kl, dl, kr, dr, cfg["bf"], cfg["fx"])
        assert m > 200
        _same_kps(_kps(d, tag + "_keys.bin"), kl, "mvKeys"); _same_kps(_kps(d, tag + "_keys_right.bin"), kr, "mvKeysRight")
        _same_kps(_kps(d, tag + "_keys_un.bin"), kl, "mvKeysUn")  # rectified stereo: mDistCoef(0) == 0 (src/Frame.cc:404-408)
        assert np.array_equal(np.fromfile(d / (tag + "_desc.bin"), np.uint8).reshape(-1, 32), dl)
        assert np.array_equal(np.fromfile(d / (tag + "_desc_right.bin"), np.uint8).reshape(-1, 32), dr)
        assert np.array_equal(np.fromfile(d / (tag + "_uright.bin"), np.float32), ur)
        assert np.array_equal(np.fromfile(d / (tag + "_depth.bin"), np.float32), dp)
        _check_common(d, tag, cfg, kl, kl, bounds, t, exl.scale_factors(), exl.inv_sigma2())
        # Frame::ComputeFboW == fbow transform(mDescriptors, 4, mFbowVec, mFbowFeatVec)
        _, (words, ww), (nodes, off, feat) = _oracle_transform(Lb, vb, dl)
        assert np.array_equal(np.fromfile(d / (tag + "_bow_words.bin"), np.uint32), words)
        assert np.array_equal(np.fromfile(d / (tag + "_bow_w.bin"), np.float32), ww)
        assert np.array_equal(np.fromfile(d / (tag + "_bow_nodes.bin"), np.uint32), nodes)
        assert np.array_equal(np.fromfile(d / (tag + "_bow_off.bin"), np.int32), off)
        assert np.array_equal(np.fromfile(d / (tag + "_bow_feat.bin"), np.int32), feat[: off[-1]])
        # Frame::GetFeaturesInArea
        g = O.Grid(kl, *bounds)
        q = np.fromfile(d / (tag + "_area_q.bin"), np.float32)
        ref_area = g.features_in_area(float(q[0]), float(q[1]), float(q[2]), 0, 3)
        assert np.array_equal(np.fromfile(d / (tag + "_area.bin"), np.int32), ref_area) and len(ref_area) > 3
        # Frame::isInFrustum on points unprojected from the frame itself (pose = 5 cm sideways)
        fr = np.fromfile(d / (tag + "_frustum.bin"), np.float32).reshape(-1, 14)
        assert len(fr) > 20
        T = np.eye(4, dtype=np.float32)[:3].copy(); T[0, 3] = 0.05
        tp = O.is_in_frustum(T, cam, bounds, fr[:, 1:4], fr[:, 4:7], fr[:, 7], fr[:, 8], 0.5, float(np.float32(np.log(np.float64(np.float32(1.2))))), 8)
        assert np.array_equal(fr[:, 9] != 0, tp["in_view"] != 0) and (tp["in_view"] != 0).sum() > 10
        v = tp["in_view"] != 0
        assert np.array_equal(fr[v, 10], tp["proj_x"][v]) and np.array_equal(fr[v, 11], tp["proj_y"][v]) and np.array_equal(fr[v, 12], tp["proj_xr"][v])
        assert np.array_equal(fr[v, 13].astype(np.int32), tp["level"][v])
        if t == 1:  # the constructed frame is then searched by Tracking::SearchLocalPoints' matcher, on the device-resident copy
            sel = np.arange(0, len(kl), 2)
            pts = np.zeros(len(sel), O.TP_DTYPE)
            pts["in_view"] = 1; pts["proj_x"] = kl["x"][sel] + np.float32(1.5); pts["proj_y"] = kl["y"][sel] - np.float32(1.0)
            pts["proj_xr"] = -1; pts["level"] = kl["octave"][sel]; pts["view_cos"] = 0.9
            ref, nref = O.search_by_projection_points(g, ur, dl, exl.scale_factors(), pts, dl[sel], np.ones(len(sel), np.int32), None, 3.0, 0.8)
            got = np.fromfile(d / "f1_matches.bin", np.int32)
            assert np.fromfile(d / "f1_nmatches.bin", np.int32)[0] == nref and nref > 300
            assert np.array_equal(got, np.where(ref >= 0, 2 * ref, -1))


@pytest.mark.gpu
def test_rgbd_frames_of_a_distorted_camera_match_the_oracle(tmp_path):
    """Frame(imGray, imDepth, ts, ex, voc, K, distCoef, bf, thDepth), src/Tracking.cc:326 -- TUM1: 640 x 480, five distortion
    coefficients, so mvKeysUn, the image bounds and the grid are those of the undistorted image"""
    cfg = dict(width=640, height=480, nfeatures=1000, fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989, bf=40.0, th_depth=40.0 * 40.0 / 517.306408)
    w, h, nf = cfg["width"], cfg["height"], cfg["nfeatures"]
    files = {"cam.f32": np.array([cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], cfg["bf"], w, h, nf, cfg["th_depth"]], np.float32),
             "dist.f32": np.array(TUM1_DIST, np.float32), "vocab.fbow": _vocab()}
    frames = [synth.stereo_pair(w, h, seed=97 + t, with_depth=True, bf=cfg["bf"]) for t in range(2)]
    for t, (l, _, depth) in enumerate(frames):
        files["left%d.bin" % t] = l; files["depth%d.f32" % t] = depth.astype(np.float32)
    _run(tmp_path, "rgbd", files)
    d = tmp_path
    f32 = np.float32
    bounds = O.image_bounds(w, h, f32(cfg["fx"]), f32(cfg["fy"]), f32(cfg["cx"]), f32(cfg["cy"]), TUM1_DIST)
    assert bounds[0] != 0 and bounds[1] != w
    for t, (l, _, depth) in enumerate(frames):
        tag = "f%d" % t
        ex = O.Extractor(nfeatures=nf)
        k, ds = ex.extract(l)
        und = O.undistort_points(np.stack([k["x"], k["y"]], 1), f32(cfg["fx"]), f32(cfg["fy"]), f32(cfg["cx"]), f32(cfg["cy"]), TUM1_DIST)
        kun = k.copy(); kun["x"], kun["y"] = und[:, 0], und[:, 1]
        ur, dp = O.stereo_from_rgbd(k, kun, depth, cfg["bf"])
        _same_kps(_kps(d, tag + "_keys.bin"), k, "mvKeys"); _same_kps(_kps(d, tag + "_keys_un.bin"), kun, "mvKeysUn")
        assert np.array_equal(np.fromfile(d / (tag + "_desc.bin"), np.uint8).reshape(-1, 32), ds)
        assert np.array_equal(np.fromfile(d / (tag + "_uright.bin"), np.float32), ur)
        assert np.array_equal(np.fromfile(d / (tag + "_depth.bin"), np.float32), dp) and (dp > 0).mean() > 0.8
        assert len(_kps(d, tag + "_keys_right.bin")) == 0
        _check_common(d, tag, cfg, k, kun, tuple(float(b) for b in bounds), t, ex.scale_factors(), ex.inv_sigma2())


@pytest.mark.gpu
def test_monocular_frames_and_the_initialisation_extractor(tmp_path):
    """Frame(imGray, ts, mpIniORBextractor / mpORBextractorLeft, ...), src/Tracking.cc:354-358: two extractors of different budgets
    (2 x nFeatures for initialisation), each with its own device context; then SearchForInitialization on the two Frames"""
    cfg = dict(width=640, height=480, nfeatures=1000, fx=500.0, fy=500.0, cx=320.0, cy=240.0, bf=0.0, th_depth=0.0)
    w, h, nf = cfg["width"], cfg["height"], cfg["nfeatures"]
    im0 = synth.stereo_pair(w, h, seed=303)[0]
    im1 = np.roll(im0, (2, -5), axis=(0, 1))  # the camera moved a little
    files = {"cam.f32": np.array([cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], cfg["bf"], w, h, nf, cfg["th_depth"]], np.float32),
             "dist.f32": np.zeros(4, np.float32), "vocab.fbow": _vocab(), "left0.bin": im0, "left1.bin": im1}
    _run(tmp_path, "mono", files)
    d = tmp_path
    bounds = (0.0, float(w), 0.0, float(h))
    got = {}
    for tag, im, n_feat, fid in (("f0", im0, 2 * nf, 0), ("f1", im1, nf, 1), ("f2", im1, 2 * nf, 2)):
        ex = O.Extractor(nfeatures=n_feat)
        k, ds = ex.extract(im)
        _same_kps(_kps(d, tag + "_keys.bin"), k, tag); _same_kps(_kps(d, tag + "_keys_un.bin"), k, tag)
        assert np.array_equal(np.fromfile(d / (tag + "_desc.bin"), np.uint8).reshape(-1, 32), ds)
        ur = np.fromfile(d / (tag + "_uright.bin"), np.float32)
        assert len(ur) == len(k) and (ur == -1).all() and (np.fromfile(d / (tag + "_depth.bin"), np.float32) == -1).all()
        st = np.fromfile(d / (tag + "_statics.bin"), np.float32)
        assert np.array_equal(st[6:10], np.asarray(bounds, np.float32))
        lit = LM.Frame(k, None, None, bounds, (500.0, 500.0, 320.0, 240.0, 0.0, 0.0), ex.scale_factors())
        assert _grid_of(d, tag) == lit.mGrid
        assert np.fromfile(d / (tag + "_meta.bin"), np.int32)[:2].tolist() == [len(k), fid]
        got[tag] = (k, ds)
    k0, d0 = got["f0"]; k2, d2 = got["f2"]
    prev = np.stack([k0["x"], k0["y"]], axis=1)
    ref, _, nref = O.search_for_initialization(k0, d0, O.Grid(k2, *bounds), d2, prev, 100, 0.9, True)
    assert np.fromfile(d / "init_n.bin", np.int32)[0] == nref and nref > 100
    assert np.array_equal(np.fromfile(d / "init_matches.bin", np.int32), ref)


@pytest.mark.gpu
def test_left_and_right_extractor_on_two_threads_like_the_reference(tmp_path):
    """src/Frame.cc:78-81: thread threadLeft(&Frame::ExtractORB, this, 0, imLeft); thread threadRight(&Frame::ExtractORB, this, 1, imRight);
    join; join -- per frame, on two extractor OBJECTS (two device contexts, created concurrently by the first frame).  200 frames,
    every one equal to what two other extractor objects return serially, which in turn equals the oracle."""
    w, h, nf = 640, 480, 1000
    files = {"cam.f32": np.array([500, 500, 320, 240, 50, w, h, nf, 3.5], np.float32), "dist.f32": np.zeros(4, np.float32), "vocab.fbow": _vocab()}
    pairs = [synth.stereo_pair(w, h, seed=401 + t) for t in range(2)]
    for t, (l, r) in enumerate(pairs):
        files["left%d.bin" % t] = l; files["right%d.bin" % t] = r
    _run(tmp_path, "threads", files)
    assert np.fromfile(tmp_path / "thr_bad.bin", np.int32).tolist() == [0, 200]
    kl, dl = O.Extractor(nfeatures=nf).extract(pairs[0][0])
    kr, dr = O.Extractor(nfeatures=nf).extract(pairs[0][1])
    _same_kps(_kps(tmp_path, "thr_keys_l.bin"), kl, "left"); _same_kps(_kps(tmp_path, "thr_keys_r.bin"), kr, "right")
    assert np.array_equal(np.fromfile(tmp_path / "thr_desc_l.bin", np.uint8).reshape(-1, 32), dl)
    assert np.array_equal(np.fromfile(tmp_path / "thr_desc_r.bin", np.uint8).reshape(-1, 32), dr)
