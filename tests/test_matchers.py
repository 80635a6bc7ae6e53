"""Tracking-thread matchers.

CPU part: first-principles checks of the oracle restatements (orb_oracle_match.c).
GPU part: liborbfe's matchers (GPU window query + Hamming, host greedy resolve) == oracle, exactly, on a
synthetic keypoint-level scene: frame t's stereo points are back-projected, the camera moves by a fixed
SE(3), and frame t+1's keypoints are the re-projections (+ jitter, bit flips, distractors).
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O

FX, FY, CX, CY, BF = 500.0, 500.0, 320.0, 240.0, 50.0
W, H = 640, 480
NL = 8
libm = C.CDLL("libm.so.6")
libm.logf.restype = C.c_float; libm.logf.argtypes = [C.c_float]
LOG_SF = float(libm.logf(np.float32(1.2)))


def _se3(yaw_deg, t):
    a = np.deg2rad(yaw_deg)
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float64)
    return np.concatenate([R, np.array(t, np.float64)[:, None]], axis=1).astype(np.float32)  # 3x4 [R|t]


def _scene(seed=0, n_last=900, n_distract=500):
    rng = np.random.default_rng(seed)
    ex = O.Extractor()
    sf = ex.scale_factors()
    # frame t (pose = identity): keypoints with depth
    u = rng.uniform(20, W - 20, n_last); v = rng.uniform(20, H - 20, n_last); z = rng.uniform(2.0, 40.0, n_last)
    pos = np.stack([(u - CX) * z / FX, (v - CY) * z / FY, z], axis=1).astype(np.float32)
    octave = rng.integers(0, NL, n_last).astype(np.int32)
    angle = rng.uniform(0, 360, n_last).astype(np.float32)
    desc_last = rng.integers(0, 256, (n_last, 32)).astype(np.uint8)
    valid = (rng.random(n_last) < 0.85).astype(np.int32)
    obs = rng.integers(0, 3, n_last).astype(np.int32)
    T_last = _se3(0.0, [0, 0, 0])
    T_cur = _se3(2.0, [0.02, -0.01, -0.3])  # camera moved 0.3 m forward, 2 deg yaw
    # frame t+1 keypoints: re-projections with jitter + distractors, shuffled
    pc = (T_cur[:, :3].astype(np.float64) @ pos.T.astype(np.float64)).T + T_cur[:, 3]
    uu = FX * pc[:, 0] / pc[:, 2] + CX; vv = FY * pc[:, 1] / pc[:, 2] + CY
    keep = (rng.random(n_last) < 0.8) & (pc[:, 2] > 0.5)
    k = np.zeros(int(keep.sum()) + n_distract, O.KP_DTYPE)
    d = np.zeros((len(k), 32), np.uint8); ur = np.full(len(k), -1.0, np.float32)
    m = int(keep.sum())
    k["x"][:m] = uu[keep] + rng.normal(0, 1.2, m); k["y"][:m] = vv[keep] + rng.normal(0, 1.2, m)
    k["octave"][:m] = np.clip(octave[keep] + rng.integers(-1, 2, m), 0, NL - 1)
    k["angle"][:m] = (angle[keep] + rng.normal(0, 6, m)) % 360
    flips = rng.random((m, 256)) < 0.07
    d[:m] = desc_last[keep] ^ np.packbits(flips, axis=1, bitorder="little")
    has_r = rng.random(m) < 0.7
    ur[:m] = np.where(has_r, k["x"][:m] - BF / pc[keep, 2] + rng.normal(0, 0.8, m), -1.0)
    k["x"][m:] = rng.uniform(0, W, n_distract); k["y"][m:] = rng.uniform(0, H, n_distract)
    k["octave"][m:] = rng.integers(0, NL, n_distract); k["angle"][m:] = rng.uniform(0, 360, n_distract)
    d[m:] = rng.integers(0, 256, (n_distract, 32))
    k["size"] = 31; k["class_id"] = -1
    perm = rng.permutation(len(k))
    k, d, ur = k[perm], d[perm], ur[perm]
    # a few keypoints slightly outside the image bounds (undistortion can do that): PosInGrid drops them
    k["x"][:3] = [-2.0, W + 1.5, 5.0]; k["y"][:3] = [10.0, 20.0, H + 3.0]
    bounds = (0.0, float(W), 0.0, float(H))
    cur_has_obs = (rng.random(len(k)) < 0.05).astype(np.uint8)
    return dict(ex=ex, sf=sf, pos=pos, octave=octave, angle=angle, desc_last=desc_last, valid=valid, obs=obs, T_last=T_last, T_cur=T_cur,
                k=k, d=d, ur=ur, bounds=bounds, cur_has_obs=cur_has_obs, rng=rng)


CAM = O.Camera(FX, FY, CX, CY, BF, BF / FX)


# ------------------------------------------------------------------ CPU: oracle from first principles
def test_three_maxima_known():
    assert O.three_maxima([0] * 30) == (-1, -1, -1)
    h = [0] * 30; h[3] = 50; h[7] = 20; h[9] = 4
    assert O.three_maxima(h) == (3, 7, -1)  # third < 10 % of the first
    h[9] = 6
    assert O.three_maxima(h) == (3, 7, 9)
    h[7] = 4
    assert O.three_maxima(h) == (3, 9, -1)
    h[9] = 4
    assert O.three_maxima(h) == (3, -1, -1)  # second < 10 % -> second and third dropped
    assert O.three_maxima([5, 5, 5] + [0] * 27) == (0, 1, 2)  # ties: earlier bin first


def test_grid_and_window_query_brute_force():
    s = _scene(1)
    g = O.Grid(s["k"], *s["bounds"])
    k = s["k"]
    inv_w = np.float32(64) / np.float32(W); inv_h = np.float32(48) / np.float32(H)
    px = np.floor(np.abs((k["x"] - np.float32(0)) * inv_w) + np.float32(0.5)) * np.sign(k["x"])  # round half away
    py = np.floor(np.abs(k["y"] * inv_h) + np.float32(0.5)) * np.sign(k["y"])
    in_grid = (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
    assert in_grid[0] and not in_grid[1] and not in_grid[2]  # x = -2 rounds into column 0; column 64 / row 48 are dropped (Q6)
    rng = np.random.default_rng(2)
    for _ in range(200):
        x, y = rng.uniform(-30, W + 30), rng.uniform(-30, H + 30)
        r = float(rng.uniform(1, 90))
        lo, hi = int(rng.integers(-1, 5)), int(rng.integers(-1, 8))
        got = g.features_in_area(x, y, r, lo, hi)
        xf, yf, rf = np.float32(x), np.float32(y), np.float32(r)
        ok = in_grid & (np.abs(k["x"] - xf) < rf) & (np.abs(k["y"] - yf) < rf)
        if lo > 0 or hi >= 0:  # Q5: literal level rule
            ok &= k["octave"] >= lo
            if hi >= 0:
                ok &= k["octave"] <= hi
        # cells are clamped to the window's cell range: a keypoint rounded into a cell outside that range is missed
        assert set(got.tolist()) <= set(np.nonzero(ok)[0].tolist())
        assert len(got) >= ok.sum() - 3
        order = [(int(px[i]), int(py[i]), int(i)) for i in got]
        assert order == sorted(order)  # ix-major, iy, insertion order


def test_log_det_and_predict_scale():
    for x in (0.3, 0.9999, 1.0, 1.2, 7.5, 123.456, 1e-3):
        assert O.lib().orc_log_det(np.float32(x)) == np.float32(np.log(np.float64(np.float32(x))))
    assert O.lib().orc_predict_scale(10.0, 10.0, LOG_SF, 8) == 0
    assert O.lib().orc_predict_scale(10.0, 10.0 / 1.2 ** 2.5, LOG_SF, 8) == 3
    assert O.lib().orc_predict_scale(10.0, 0.01, LOG_SF, 8) == 7 and O.lib().orc_predict_scale(10.0, 100.0, LOG_SF, 8) == 0


def test_search_by_projection_last_recovers_motion():
    s = _scene(3)
    g = O.Grid(s["k"], *s["bounds"])
    match, n = O.search_by_projection_last(g, s["ur"], s["d"], s["sf"], CAM, s["T_cur"], s["T_last"], s["pos"], s["desc_last"], s["valid"],
                                           s["obs"], s["octave"], s["angle"], s["cur_has_obs"], 7.0, False, True)
    assert n == (match >= 0).sum() or n <= (match >= 0).sum() + 5  # duplicates may be counted twice (reference quirk)
    assert n > 300
    ok = match >= 0
    # matched pairs are the planted ones: descriptors within 100 bits
    dist = np.unpackbits(s["d"][ok] ^ s["desc_last"][match[ok]], axis=1).sum(axis=1)
    assert (dist <= 100).all() and np.median(dist) < 30
    assert (s["valid"][match[ok]] == 1).all() and not (s["cur_has_obs"][ok] == 1).any()


def test_search_for_initialization_one_to_one():
    s = _scene(4)
    k1 = s["k"].copy(); k1["octave"] = np.where(np.arange(len(k1)) % 3 == 0, 1, 0)
    rng = np.random.default_rng(5)
    k2 = k1.copy(); k2["x"] += rng.normal(3, 1, len(k1)).astype(np.float32); k2["y"] += rng.normal(-2, 1, len(k1)).astype(np.float32)
    k2["octave"] = 0
    d2 = s["d"] ^ np.packbits(rng.random((len(k1), 256)) < 0.05, axis=1, bitorder="little")
    g2 = O.Grid(k2, *s["bounds"])
    prev = np.stack([k1["x"], k1["y"]], axis=1)
    m12, pm, n = O.search_for_initialization(k1, s["d"], g2, d2, prev, 100, 0.9, True)
    ok = m12 >= 0
    assert n == ok.sum() and n > 200
    assert (k1["octave"][ok] == 0).all()
    assert len(set(m12[ok].tolist())) == ok.sum()  # one-to-one
    assert np.mean(m12[ok] == np.nonzero(ok)[0]) > 0.95  # planted correspondences
    assert np.array_equal(pm[ok], np.stack([k2["x"][m12[ok]], k2["y"][m12[ok]]], axis=1)) and np.array_equal(pm[~ok], prev[~ok])


# ------------------------------------------------------------------ GPU: product == oracle
@pytest.fixture(scope="module")
def gpu():
    from orbslam2_amd import api
    ctx = api.Context(width=W, height=H, fx=FX, fy=FY, cx=CX, cy=CY, bf=BF)
    yield api, ctx
    ctx.close()


@pytest.mark.gpu
def test_gpu_features_in_area(gpu):
    api, ctx = gpu
    s = _scene(6)
    g = O.Grid(s["k"], *s["bounds"])
    view = ctx._view(s["k"], s["ur"], s["d"], s["bounds"])
    rng = np.random.default_rng(7)
    for _ in range(40):
        x, y, r = float(rng.uniform(-30, W + 30)), float(rng.uniform(-30, H + 30)), float(rng.uniform(1, 120))
        lo, hi = int(rng.integers(-1, 5)), int(rng.integers(-1, 8))
        assert ctx.features_in_area(view, x, y, r, lo, hi).tolist() == g.features_in_area(x, y, r, lo, hi).tolist()
    # the batch form: one upload, many windows, same lists
    qs = [(float(rng.uniform(-30, W + 30)), float(rng.uniform(-30, H + 30)), float(rng.uniform(1, 120)), int(rng.integers(-1, 5)), int(rng.integers(-1, 8)))
          for _ in range(100)]
    got = ctx.features_in_area_batch(view, [q[0] for q in qs], [q[1] for q in qs], [q[2] for q in qs], [q[3] for q in qs], [q[4] for q in qs])
    for q, lst in zip(qs, got):
        assert lst.tolist() == g.features_in_area(*q).tolist()
    assert sum(len(x) for x in got) > 1000
    assert ctx.features_in_area_batch(view, [], [], []) == []
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    h = np.array([0, 9, 2, 9, 1] + [0] * 25, np.int32)
    assert ctx.L.orbfe_three_maxima(h.ctypes.data_as(C.c_void_p), 30, C.byref(a), C.byref(b), C.byref(c)) == 0
    assert (a.value, b.value, c.value) == O.three_maxima(h)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,th,mono,ori", [(10, 7.0, False, True), (11, 15.0, True, True), (12, 14.0, False, False), (13, 3.0, False, True)])
def test_gpu_search_by_projection_last(gpu, seed, th, mono, ori):
    api, ctx = gpu
    s = _scene(seed)
    if seed == 12:  # backward motion: exercises the [0, octave] level window
        s["T_cur"] = _se3(-1.0, [0.0, 0.0, 0.4])
    g = O.Grid(s["k"], *s["bounds"])
    ur = None if mono else s["ur"]
    ref, nref = O.search_by_projection_last(g, ur, s["d"], s["sf"], CAM, s["T_cur"], s["T_last"], s["pos"], s["desc_last"], s["valid"],
                                            s["obs"], s["octave"], s["angle"], s["cur_has_obs"], th, mono, ori)
    view = ctx._view(s["k"], ur, s["d"], s["bounds"])
    got, ngot = ctx.search_by_projection_last(view, s["T_cur"], s["T_last"], s["pos"], s["desc_last"], s["valid"], s["obs"], s["octave"],
                                              s["angle"], s["cur_has_obs"], th, mono, ori)
    assert ngot == nref and np.array_equal(got, ref)
    assert nref > 50


@pytest.mark.gpu
def test_gpu_frustum_and_search_by_projection_points(gpu):
    api, ctx = gpu
    s = _scene(20, n_last=1500)
    rng = s["rng"]
    n = len(s["pos"])
    # MapPoint normal = mean viewing direction (camera -> point)
    normal = s["pos"] / np.linalg.norm(s["pos"], axis=1, keepdims=True) + rng.normal(0, 0.35, (n, 3))
    normal = (normal / np.linalg.norm(normal, axis=1, keepdims=True)).astype(np.float32)
    dist0 = np.linalg.norm(s["pos"], axis=1).astype(np.float32)
    max_d = (dist0 * rng.uniform(0.9, 3.0, n)).astype(np.float32); min_d = (max_d / np.float32(1.2 ** 7)).astype(np.float32)
    ref_tp = O.is_in_frustum(s["T_cur"], CAM, s["bounds"], s["pos"], normal, max_d, min_d, 0.5, LOG_SF, NL)
    got_tp = ctx.is_in_frustum(s["T_cur"], s["bounds"], s["pos"], normal, max_d, min_d, 0.5)
    assert np.array_equal(got_tp["in_view"], ref_tp["in_view"]) and ref_tp["in_view"].sum() > 300
    v = ref_tp["in_view"] == 1
    for f in ("proj_x", "proj_y", "proj_xr", "level", "view_cos"):
        assert np.array_equal(got_tp[f][v], ref_tp[f][v]), f
    g = O.Grid(s["k"], *s["bounds"])
    view = ctx._view(s["k"], s["ur"], s["d"], s["bounds"])
    for th, ratio in ((1.0, 0.8), (3.0, 0.8), (5.0, 0.6)):
        ref, nref = O.search_by_projection_points(g, s["ur"], s["d"], s["sf"], ref_tp, s["desc_last"], s["obs"], s["cur_has_obs"], th, ratio)
        got, ngot = ctx.search_by_projection_points(view, ref_tp, s["desc_last"], s["obs"], s["cur_has_obs"], th, ratio)
        assert ngot == nref and np.array_equal(got, ref)
    assert nref > 100


@pytest.mark.gpu
def test_gpu_search_by_projection_kf(gpu):
    api, ctx = gpu
    s = _scene(30)
    n = len(s["pos"])
    dist0 = np.linalg.norm(s["pos"], axis=1).astype(np.float32)
    max_d = (dist0 * s["rng"].uniform(0.9, 3.0, n)).astype(np.float32); min_d = (max_d / np.float32(1.2 ** 7)).astype(np.float32)
    g = O.Grid(s["k"], *s["bounds"])
    view = ctx._view(s["k"], None, s["d"], s["bounds"])
    for th, od in ((10.0, 100), (3.0, 64)):
        ref, nref = O.search_by_projection_kf(g, s["d"], s["sf"], CAM, s["T_cur"], LOG_SF, NL, s["pos"], s["desc_last"], s["valid"], s["angle"],
                                              max_d, min_d, s["cur_has_obs"], th, od, True)
        got, ngot = ctx.search_by_projection_kf(view, s["T_cur"], s["pos"], s["desc_last"], s["valid"], s["angle"], max_d, min_d,
                                                s["cur_has_obs"], th, od, True)
        assert ngot == nref and np.array_equal(got, ref)
    assert nref > 30


@pytest.mark.gpu
def test_gpu_search_for_initialization(gpu):
    api, ctx = gpu
    s = _scene(40, n_last=1800, n_distract=300)
    k1 = s["k"].copy(); k1["octave"] = np.where(np.arange(len(k1)) % 4 == 0, 1, 0)
    rng = np.random.default_rng(41)
    k2 = k1.copy(); k2["x"] += rng.normal(5, 2, len(k1)).astype(np.float32); k2["y"] += rng.normal(-3, 2, len(k1)).astype(np.float32)
    k2["octave"] = np.where(np.arange(len(k1)) % 7 == 0, 2, 0)
    d2 = s["d"] ^ np.packbits(rng.random((len(k1), 256)) < 0.05, axis=1, bitorder="little")
    g2 = O.Grid(k2, *s["bounds"])
    prev = np.stack([k1["x"], k1["y"]], axis=1)
    v1 = ctx._view(k1, None, s["d"], s["bounds"]); v2 = ctx._view(k2, None, d2, s["bounds"])
    for win, ratio, ori in ((100, 0.9, True), (10, 0.6, False)):
        ref, pm_ref, nref = O.search_for_initialization(k1, s["d"], g2, d2, prev, win, ratio, ori)
        got, pm_got, ngot = ctx.search_for_initialization(v1, v2, prev, win, ratio, ori)
        assert ngot == nref and np.array_equal(got, ref) and np.array_equal(pm_got, pm_ref)
    assert nref > 100


@pytest.mark.gpu
def test_gpu_matchers_edge_cases(gpu):
    api, ctx = gpu
    s = _scene(50)
    view = ctx._view(s["k"], s["ur"], s["d"], s["bounds"])
    z = np.zeros
    got, n = ctx.search_by_projection_last(view, s["T_cur"], s["T_last"], z((0, 3), np.float32), z((0, 32), np.uint8), z(0, np.int32),
                                           z(0, np.int32), z(0, np.int32), z(0, np.float32), None, 7.0, False, True)
    assert n == 0 and (got == -1).all()
    empty = ctx._view(s["k"][:0], None, s["d"][:0], s["bounds"])
    got, n = ctx.search_by_projection_last(empty, s["T_cur"], s["T_last"], s["pos"], s["desc_last"], s["valid"], s["obs"], s["octave"],
                                           s["angle"], None, 7.0, False, True)
    assert n == 0 and len(got) == 0
    bad = ctx._view(s["k"], None, s["d"], (0.0, 0.0, 0.0, float(H)))  # degenerate bounds
    with pytest.raises(api.OrbfeError):
        ctx.features_in_area(bad, 1.0, 1.0, 5.0)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,th,stereo", [(40, 3.0, True), (41, 2.5, False), (42, 6.0, True)])
def test_gpu_fuse(gpu, seed, th, stereo):
    """Search part of ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) (src/ORBmatcher.cc:821-971) == oracle: projection and range /
    viewing-angle gates, predicted level window, chi-square reprojection gates (stereo and mono keypoints), best Hamming <= TH_LOW."""
    api, ctx = gpu
    s = _scene(seed, n_last=1500, n_distract=400)
    rng = np.random.default_rng(seed)
    n = len(s["pos"])
    ow = np.zeros(3)  # map points were created from the last frame at the origin
    dist0 = np.linalg.norm(s["pos"] - ow, axis=1).astype(np.float32)
    lvl0 = s["octave"]
    max_d = (dist0 * s["sf"][lvl0]).astype(np.float32)                 # mfMaxDistance = dist * levelScaleFactor (src/MapPoint.cc:356-361)
    min_d = (max_d / s["sf"][NL - 1]).astype(np.float32)
    normal = (s["pos"] / dist0[:, None] + rng.normal(0, 0.45, (n, 3))).astype(np.float32)  # PO . Pn >= 0.5 |PO| for most, not all
    normal = (normal / np.linalg.norm(normal, axis=1, keepdims=True)).astype(np.float32)
    valid = s["valid"]
    ur = s["ur"] if stereo else None
    g = O.Grid(s["k"], *s["bounds"])
    ex = s["ex"]
    ref, nref = O.fuse(g, ur, s["d"], s["sf"], ex.inv_sigma2(), CAM, s["T_cur"], LOG_SF, NL, s["pos"], normal, max_d, min_d, s["desc_last"], valid, th)
    view = ctx._view(s["k"], ur, s["d"], s["bounds"])
    got, ngot = ctx.fuse(view, s["T_cur"], s["pos"], normal, max_d, min_d, s["desc_last"], valid, th)
    assert ngot == nref and np.array_equal(got, ref)
    assert nref > 100 and (ref[valid == 0] == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("mode,seed,th", [(0, 50, 10.0), (0, 51, 4.0), (1, 52, 4.0), (1, 53, 10.0)])
def test_gpu_sim3_projection_matchers(gpu, mode, seed, th):
    """LoopClosing matchers on a Sim3 pose: SearchByProjection(KeyFrame*, Scw, ...) (src/ORBmatcher.cc:285-398, greedy over the
    points) and the search part of Fuse(KeyFrame*, Scw, ...) (:973-1096) == oracle."""
    api, ctx = gpu
    s = _scene(seed, n_last=1500, n_distract=400)
    rng = np.random.default_rng(seed)
    n = len(s["pos"])
    dist0 = np.linalg.norm(s["pos"], axis=1).astype(np.float32)
    max_d = (dist0 * s["sf"][s["octave"]]).astype(np.float32)
    min_d = (max_d / s["sf"][NL - 1]).astype(np.float32)
    normal = (s["pos"] / dist0[:, None] + rng.normal(0, 0.45, (n, 3))).astype(np.float32)
    normal = (normal / np.linalg.norm(normal, axis=1, keepdims=True)).astype(np.float32)
    scale = np.float32(1.07)
    Scw = s["T_cur"].copy(); Scw *= scale  # [sR | s t]: the same camera pose, map scaled by 1.07
    kf_matched = (rng.random(len(s["k"])) < 0.1).astype(np.uint8)
    g = O.Grid(s["k"], *s["bounds"])
    ref, nref = O.sim3_projection(mode, g, s["d"], s["sf"], CAM, Scw, LOG_SF, NL, s["pos"], normal, max_d, min_d, s["desc_last"], s["valid"], kf_matched, th)
    view = ctx._view(s["k"], s["ur"], s["d"], s["bounds"])
    got, ngot = ctx.sim3_projection(mode, view, Scw, s["pos"], normal, max_d, min_d, s["desc_last"], s["valid"], kf_matched, th)
    assert ngot == nref and np.array_equal(got, ref)
    assert nref > 100
    m = ref[ref >= 0]
    if mode == 0:
        assert len(np.unique(m)) == len(m) and not kf_matched[m].any()  # one keypoint per point, none of the pre-matched ones


@pytest.mark.gpu
@pytest.mark.parametrize("seed,th,kfb", [(60, 7.5, False), (61, 3.0, False), (62, 7.5, True)])
def test_gpu_search_by_sim3(gpu, seed, th, kfb):
    """ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1098-1322) == oracle: two keyframes of the same 3-D points whose maps differ by a
    similarity; both projection directions and the mutual-consistency check."""
    api, ctx = gpu
    rng = np.random.default_rng(seed)
    ex = O.Extractor()
    sf = ex.scale_factors()
    n = 1300
    Pw = np.stack([rng.uniform(-8, 8, n), rng.uniform(-5, 5, n), rng.uniform(4, 35, n)], axis=1)
    T1 = _se3(0.0, [0, 0, 0]).astype(np.float64)
    T2 = _se3(4.0, [-0.5, 0.03, 0.1]).astype(np.float64)
    base_d = rng.integers(0, 256, (n, 32)).astype(np.uint8)
    bounds = KF_BOUNDS if kfb else (0.0, float(W), 0.0, float(H))  # kfb: a distorted camera's bounds, kept as ints by the keyframes

    def keyframe(T, seed2, extra):
        r = np.random.default_rng(seed2)
        pc = (T[:, :3] @ Pw.T).T + T[:, 3]
        k = np.zeros(n + extra, O.KP_DTYPE)
        k["x"][:n] = FX * pc[:, 0] / pc[:, 2] + CX + r.normal(0, 0.8, n); k["y"][:n] = FY * pc[:, 1] / pc[:, 2] + CY + r.normal(0, 0.8, n)
        k["x"][n:] = r.uniform(0, W, extra); k["y"][n:] = r.uniform(0, H, extra)
        k["octave"] = r.integers(0, NL, n + extra); k["angle"] = r.uniform(0, 360, n + extra); k["size"] = 31; k["class_id"] = -1
        d = np.concatenate([base_d ^ np.packbits(r.random((n, 256)) < 0.05, axis=1, bitorder="little"), r.integers(0, 256, (extra, 32)).astype(np.uint8)])
        dist = np.linalg.norm(pc, axis=1)
        pos = np.zeros((n + extra, 3), np.float32); pos[:n] = Pw
        mx = np.ones(n + extra, np.float32); mx[:n] = dist * sf[k["octave"][:n]]
        mn = (mx / sf[NL - 1]).astype(np.float32)
        valid = np.zeros(n + extra, np.int32); valid[:n] = r.random(n) < 0.85
        perm = r.permutation(n + extra)
        return k[perm], d[perm], pos[perm], mx[perm], mn[perm], valid[perm], perm

    k1, d1, pos1, mx1, mn1, v1, perm1 = keyframe(T1, seed * 7 + 1, 300)
    k2, d2, pos2, mx2, mn2, v2, perm2 = keyframe(T2, seed * 7 + 2, 200)
    # Sim3 from camera 2 to camera 1 (s12 = 1 here up to a small scale drift): X1 = s12 R12 X2 + t12
    R12 = (T1[:, :3] @ T2[:, :3].T)
    t12 = T1[:, 3] - R12 @ T2[:, 3]
    s12 = np.float32(1.02)
    g1, g2 = O.Grid(k1, *bounds, keyframe=kfb), O.Grid(k2, *bounds, keyframe=kfb)
    pts1 = (pos1, mx1, mn1, d1, v1); pts2 = (pos2, mx2, mn2, d2, v2)  # GetDescriptor() = the observing keypoint's descriptor here
    ref, nref = O.search_by_sim3(g1, d1, T1.astype(np.float32), pts1, g2, d2, T2.astype(np.float32), pts2, sf, CAM, LOG_SF, NL,
                                 s12, R12.astype(np.float32), t12.astype(np.float32), th)
    view1 = ctx._view(k1, None, d1, bounds, keyframe=kfb); view2 = ctx._view(k2, None, d2, bounds, keyframe=kfb)
    got, ngot = ctx.search_by_sim3(view1, T1.astype(np.float32), pts1, view2, T2.astype(np.float32), pts2, s12, R12.astype(np.float32),
                                   t12.astype(np.float32), th)
    assert ngot == nref and np.array_equal(got, ref)
    assert nref > 150
    ok = np.nonzero(ref >= 0)[0]
    assert (perm1[ok] == perm2[ref[ok]]).mean() > 0.9  # same 3-D point on both sides

# Image bounds of a distorted camera (Frame::ComputeImageBounds undistorts the four corners): not whole numbers.  A KeyFrame keeps
# them as ints (include/KeyFrame.h:194-197) while its grid was filled by the Frame with the floats.
KF_BOUNDS = (-13.62, W + 10.71, -9.37, H + 7.48)


@pytest.mark.gpu
def test_gpu_keyframe_views_use_the_keyframes_integer_bounds(gpu):
    """KeyFrame-side matchers with the bounds of a DISTORTED camera: cells assigned with the frame's float bounds, windows
    (KeyFrame::GetFeaturesInArea, src/KeyFrame.cc:568-580) and IsInImage (:604-607) with their integer truncation -- Fuse, the
    two Sim3 matchers and SearchBySim3 against the oracle's keyframe grid; and the flag is not a no-op on this data."""
    api, ctx = gpu
    s = _scene(71, n_last=1500, n_distract=400)
    rng = np.random.default_rng(71)
    n = len(s["pos"])
    dist0 = np.linalg.norm(s["pos"], axis=1).astype(np.float32)
    max_d = (dist0 * s["sf"][s["octave"]]).astype(np.float32)
    min_d = (max_d / s["sf"][NL - 1]).astype(np.float32)
    normal = (s["pos"] / dist0[:, None] + rng.normal(0, 0.45, (n, 3))).astype(np.float32)
    normal = (normal / np.linalg.norm(normal, axis=1, keepdims=True)).astype(np.float32)
    ex = s["ex"]
    g = O.Grid(s["k"], *KF_BOUNDS, keyframe=True)
    view = ctx._view(s["k"], s["ur"], s["d"], KF_BOUNDS, keyframe=True)
    # the raw window query: integer bounds move cell borders by up to 0.62 px, so some windows gain or lose a cell column / row
    g_frame = O.Grid(s["k"], *KF_BOUNDS)
    qx = rng.uniform(0, W, 600).astype(np.float32); qy = rng.uniform(0, H, 600).astype(np.float32); qr = rng.uniform(2, 30, 600).astype(np.float32)
    got_q = ctx.features_in_area_batch(view, qx, qy, qr)
    for i in range(600):
        assert np.array_equal(got_q[i], g.features_in_area(qx[i], qy[i], qr[i]))
    # Fuse
    ref, nref = O.fuse(g, s["ur"], s["d"], s["sf"], ex.inv_sigma2(), CAM, s["T_cur"], LOG_SF, NL, s["pos"], normal, max_d, min_d, s["desc_last"], s["valid"], 3.0)
    got, ngot = ctx.fuse(view, s["T_cur"], s["pos"], normal, max_d, min_d, s["desc_last"], s["valid"], 3.0)
    assert ngot == nref and np.array_equal(got, ref) and nref > 100
    # Sim3 SearchByProjection / Fuse
    Scw = s["T_cur"].copy(); Scw *= np.float32(1.07)
    kf_matched = (rng.random(len(s["k"])) < 0.1).astype(np.uint8)
    for mode in (0, 1):
        ref, nref = O.sim3_projection(mode, g, s["d"], s["sf"], CAM, Scw, LOG_SF, NL, s["pos"], normal, max_d, min_d, s["desc_last"], s["valid"], kf_matched, 6.0)
        got, ngot = ctx.sim3_projection(mode, view, Scw, s["pos"], normal, max_d, min_d, s["desc_last"], s["valid"], kf_matched, 6.0)
        assert ngot == nref and np.array_equal(got, ref) and nref > 100
    # the cell ranges of the keyframe grid differ from the frame grid's for some windows of this scene (the flag matters)
    differ = 0
    cells = lambda grid, x, y, r: (int(np.floor((x - grid[0] - r) * grid[2])), int(np.ceil((x - grid[0] + r) * grid[2])))
    inv_w = np.float32(64.0) / (np.float32(KF_BOUNDS[1]) - np.float32(KF_BOUNDS[0]))
    for i in range(600):
        a = cells((np.float32(KF_BOUNDS[0]), 0, inv_w), qx[i], 0, qr[i])
        b = cells((np.float32(int(KF_BOUNDS[0])), 0, inv_w), qx[i], 0, qr[i])
        differ += a != b
    assert differ > 10


def _frame_scene(k, d, ur, seed, all_points=False):
    """Map points for a REAL extracted frame (k, d, ur): the stereo keypoints back-projected at the identity pose, seen again
    from a camera 0.25 m further on; a `last frame` whose rows are those points.  all_points: monocular keypoints get a
    random depth too (bench: one map point per keypoint)."""
    rng = np.random.default_rng(seed)
    good = np.nonzero(ur > 0)[0] if not all_points else np.arange(len(k))
    z = np.where(ur[good] > 0, BF / np.maximum(k["x"][good] - ur[good], 1e-3), rng.uniform(3.0, 30.0, len(good))).astype(np.float32)
    pos = np.stack([(k["x"][good] - CX) * z / FX, (k["y"][good] - CY) * z / FY, z], axis=1).astype(np.float32)
    m = len(good)
    desc = d[good] ^ np.packbits(rng.random((m, 256)) < 0.05, axis=1, bitorder="little")
    return dict(pos=pos, desc=desc, valid=(rng.random(m) < 0.9).astype(np.int32), obs=rng.integers(0, 3, m).astype(np.int32),
                octave=k["octave"][good].astype(np.int32), angle=((k["angle"][good] + rng.normal(0, 5, m)) % 360).astype(np.float32),
                T_last=_se3(0.0, [0, 0, 0]), T_cur=_se3(0.6, [0.01, -0.005, -0.25]), has=(rng.random(len(k)) < 0.05).astype(np.uint8), rng=rng)


@pytest.mark.gpu
@pytest.mark.parametrize("distorted", [False, True])
def test_gpu_matchers_on_the_device_resident_frame(distorted):
    """orbfe_frame_view.device_slot_plus1: the Tracking matchers read the CURRENT frame where the extraction left it in HBM
    (keypoints undistorted on the device, grid built once per frame) -- same results as the upload path and as the oracle,
    on the first call, on a cached-grid call, and after the next extraction replaced the frame."""
    from orbslam2_amd import api, synth
    ctx = api.Context(width=W, height=H, nfeatures=1500, fx=FX, fy=FY, cx=CX, cy=CY, bf=BF)
    dist = [-0.28, 0.07, 2e-4, 1e-5, 0.0]
    if distorted:
        ctx.set_distortion(dist)
    for rep, seed in enumerate((501, 502)):
        left, right = synth.stereo_pair(W, H, seed=seed)
        out = ctx.stereo_frame(left, right)
        k, d, ur = out["kps_left"], out["desc_left"], out["u_right"]
        kun = ctx.fetch_keys_un(0) if distorted else k
        bounds = tuple(float(b) for b in ctx.image_bounds()) if distorted else (0.0, float(W), 0.0, float(H))
        if distorted:
            und = O.undistort_points(np.stack([k["x"], k["y"]], 1), FX, FY, CX, CY, dist)
            ref_un = k.copy(); ref_un["x"], ref_un["y"] = und[:, 0], und[:, 1]
            assert np.array_equal(kun, ref_un)
        s = _frame_scene(kun, d, ur, seed)
        sf = O.Extractor().scale_factors()
        g = O.Grid(kun, *bounds)
        up = ctx._view(kun, ur, d, bounds); dev = ctx._view(kun, ur, d, bounds, device_slot=0)
        for th, mono, ori in ((7.0, False, True), (15.0, True, False)):
            u = None if mono else ur
            vu = ctx._view(kun, u, d, bounds); vd = ctx._view(kun, u, d, bounds, device_slot=0)
            ref, nref = O.search_by_projection_last(g, u, d, sf, CAM, s["T_cur"], s["T_last"], s["pos"], s["desc"], s["valid"], s["obs"],
                                                    s["octave"], s["angle"], s["has"], th, mono, ori)
            for v in (vu, vd, vd):  # the second resident call reuses the grid
                got, ngot = ctx.search_by_projection_last(v, s["T_cur"], s["T_last"], s["pos"], s["desc"], s["valid"], s["obs"], s["octave"],
                                                          s["angle"], s["has"], th, mono, ori)
                assert ngot == nref and np.array_equal(got, ref), (rep, th)
            assert nref > 40
        n = len(s["pos"])
        dist0 = np.linalg.norm(s["pos"], axis=1).astype(np.float32)
        max_d = (dist0 * sf[s["octave"]]).astype(np.float32); min_d = (max_d / sf[NL - 1]).astype(np.float32)
        normal = (s["pos"] / dist0[:, None]).astype(np.float32)
        tp = O.is_in_frustum(s["T_cur"], CAM, bounds, s["pos"], normal, max_d, min_d, 0.5, LOG_SF, NL)
        ref, nref = O.search_by_projection_points(g, ur, d, sf, tp, s["desc"], s["obs"], s["has"], 3.0, 0.8)
        for v in (up, dev):
            got, ngot = ctx.search_by_projection_points(v, tp, s["desc"], s["obs"], s["has"], 3.0, 0.8)
            assert ngot == nref and np.array_equal(got, ref)
        assert nref > 40
        ref, nref = O.search_by_projection_kf(g, d, sf, CAM, s["T_cur"], LOG_SF, NL, s["pos"], s["desc"], s["valid"], s["angle"], max_d, min_d, s["has"], 10.0, 100, True)
        for v in (ctx._view(kun, None, d, bounds), ctx._view(kun, None, d, bounds, device_slot=0)):
            got, ngot = ctx.search_by_projection_kf(v, s["T_cur"], s["pos"], s["desc"], s["valid"], s["angle"], max_d, min_d, s["has"], 10.0, 100, True)
            assert ngot == nref and np.array_equal(got, ref)
        assert nref > 40
        # SearchForInitialization: frame 1 = a jittered copy (host arrays), frame 2 = the resident frame
        rng = s["rng"]
        k1 = kun.copy(); k1["x"] += rng.normal(3, 2, len(k1)).astype(np.float32); k1["y"] += rng.normal(-2, 2, len(k1)).astype(np.float32)
        d1 = d ^ np.packbits(rng.random((len(k1), 256)) < 0.04, axis=1, bitorder="little")
        prev = np.stack([k1["x"], k1["y"]], axis=1)
        ref, pm_ref, nref = O.search_for_initialization(k1, d1, g, d, prev, 100, 0.9, True)
        v1 = ctx._view(k1, None, d1, bounds)
        for v2 in (ctx._view(kun, None, d, bounds), ctx._view(kun, None, d, bounds, device_slot=0)):
            got, pm, ngot = ctx.search_for_initialization(v1, v2, prev, 100, 0.9, True)
            assert ngot == nref and np.array_equal(got, ref) and np.array_equal(pm, pm_ref)
        assert nref > 40
    with pytest.raises(api.OrbfeError):
        ctx.search_by_projection_points(ctx._view(kun, ur, d, bounds, device_slot=5), tp, s["desc"], s["obs"], s["has"], 3.0, 0.8)  # no such slot
    ctx.close()


@pytest.mark.gpu
def test_gpu_topk_prefix_runs_out_and_the_full_list_takes_over(gpu):
    """The device hands the host the 4 best statically admissible keys per query; when earlier queries have taken all of them
    the replay must continue on the full list (orbfe_match_resolve.h: first_two).  Forced here: every map point appears 10 times
    and the frame holds a cluster of 8 near-copies of its keypoint (Hamming distance 0 .. 7), so copy j takes the j-th best
    keypoint and copies 5 .. 8 need candidates beyond the prefix; and with windows of more than 256 candidates the LDS stage of
    the top-K selection gives up and the whole query goes to the full list."""
    api, ctx = gpu
    s = _scene(90, n_last=150, n_distract=700)
    rng = np.random.default_rng(91)
    T = s["T_cur"].astype(np.float64)
    pc = (T[:, :3] @ s["pos"].T.astype(np.float64)).T + T[:, 3]
    uu = FX * pc[:, 0] / pc[:, 2] + CX; vv = FY * pc[:, 1] / pc[:, 2] + CY
    ok = np.nonzero((s["valid"] == 1) & (pc[:, 2] > 0.5) & (uu > 30) & (uu < W - 30) & (vv > 30) & (vv < H - 30))[0][:60]
    extra_k = np.zeros(len(ok) * 8, O.KP_DTYPE); extra_d = np.zeros((len(ok) * 8, 32), np.uint8)
    for a_, i in enumerate(ok):
        for j in range(8):
            e = a_ * 8 + j
            extra_k["x"][e] = uu[i] + rng.uniform(-1, 1); extra_k["y"][e] = vv[i] + rng.uniform(-1, 1)
            extra_k["octave"][e] = s["octave"][i]; extra_k["angle"][e] = s["angle"][i]
            bits = np.zeros(256, bool); bits[rng.permutation(256)[:j]] = True  # j bits away from the map point's descriptor
            extra_d[e] = s["desc_last"][i] ^ np.packbits(bits, bitorder="little")
    extra_k["size"] = 31; extra_k["class_id"] = -1
    k = np.concatenate([s["k"], extra_k]); d = np.concatenate([s["d"], extra_d]); ur = np.concatenate([s["ur"], np.full(len(extra_k), -1.0, np.float32)])
    has = np.concatenate([s["cur_has_obs"], np.zeros(len(extra_k), np.uint8)])
    rep = 10
    pos = np.repeat(s["pos"], rep, axis=0); desc = np.repeat(s["desc_last"], rep, axis=0)
    valid = np.repeat(s["valid"], rep); obs = np.ones(len(pos), np.int32)  # Observations() > 0: every match blocks its keypoint
    octave = np.repeat(s["octave"], rep); angle = np.repeat(s["angle"], rep)
    g = O.Grid(k, *s["bounds"])
    view = ctx._view(k, ur, d, s["bounds"])
    for th in (7.0, 120.0):  # 120 px x scale: windows with hundreds of candidates
        ref, nref = O.search_by_projection_last(g, ur, d, s["sf"], CAM, s["T_cur"], s["T_last"], pos, desc, valid, obs, octave, angle, has, th, False, False)
        got, ngot = ctx.search_by_projection_last(view, s["T_cur"], s["T_last"], pos, desc, valid, obs, octave, angle, has, th, False, False)
        assert ngot == nref and np.array_equal(got, ref), th
        assert nref > 8 * len(ok) - 20  # the clusters were used up: copies 5 .. 8 found their keypoint beyond the prefix
    tp = np.zeros(len(pos), O.TP_DTYPE)  # the same through SearchByProjection(F, points): best AND second beyond the prefix
    tp["in_view"] = valid; tp["proj_x"] = np.repeat(uu, rep); tp["proj_y"] = np.repeat(vv, rep); tp["proj_xr"] = -1; tp["level"] = octave; tp["view_cos"] = 0.9
    tp["in_view"][(tp["proj_x"] < 0) | (tp["proj_x"] > W) | (tp["proj_y"] < 0) | (tp["proj_y"] > H)] = 0
    ref, nref = O.search_by_projection_points(g, ur, d, s["sf"], tp, desc, obs, has, 3.0, 0.99)
    got, ngot = ctx.search_by_projection_points(view, tp, desc, obs, has, 3.0, 0.99)
    assert ngot == nref and np.array_equal(got, ref) and nref > 200
    # SearchForInitialization: many frame-1 keypoints compete for the same frame-2 keypoints (stealing rule, :437-438, :460-464)
    k1 = np.repeat(k[-len(extra_k):], 3); d1 = np.repeat(d[-len(extra_k):], 3, axis=0) ^ np.packbits(rng.random((3 * len(extra_k), 256)) < 0.01, axis=1, bitorder="little")
    k1["octave"] = 0
    k2 = k.copy(); k2["octave"] = 0
    g2 = O.Grid(k2, *s["bounds"])
    prev = np.stack([k1["x"], k1["y"]], axis=1)
    ref, pm_ref, nref = O.search_for_initialization(k1, d1, g2, d, prev, 100, 0.99, False)
    got, pm, ngot = ctx.search_for_initialization(ctx._view(k1, None, d1, s["bounds"]), ctx._view(k2, None, d, s["bounds"]), prev, 100, 0.99, False)
    assert ngot == nref and np.array_equal(got, ref) and np.array_equal(pm, pm_ref) and nref > 50
