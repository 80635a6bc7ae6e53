"""C oracle == literal Python transcription of the KeyFrame-side searches (oracle/literal_kf_matchers.py): KeyFrame grid taken
from the frame with INTEGER bounds for GetFeaturesInArea / IsInImage, Fuse(KeyFrame*, points), SearchByProjection(KeyFrame*, Scw)
and Fuse(KeyFrame*, Scw) -- with whole-number bounds and with the fractional bounds of a distorted camera.  CPU only."""
import numpy as np
import pytest

from oracle import literal_kf_matchers as LK
from oracle import literal_matchers as LM
from oracle import oracle as O
from tests.test_matchers import BF, CAM, CX, CY, FX, FY, H, KF_BOUNDS, LOG_SF, NL, W, _scene

CAMT = (FX, FY, CX, CY, BF, BF / FX)


def _setup(seed, bounds):
    s = _scene(seed, n_last=900, n_distract=300)
    rng = np.random.default_rng(seed)
    n = len(s["pos"])
    dist0 = np.linalg.norm(s["pos"], axis=1).astype(np.float32)
    max_d = (dist0 * s["sf"][s["octave"]]).astype(np.float32)
    min_d = (max_d / s["sf"][NL - 1]).astype(np.float32)
    normal = (s["pos"] / dist0[:, None] + rng.normal(0, 0.45, (n, 3))).astype(np.float32)
    normal = (normal / np.linalg.norm(normal, axis=1, keepdims=True)).astype(np.float32)
    g = O.Grid(s["k"], *bounds, keyframe=True)
    F = LM.Frame(s["k"], s["d"], s["ur"], bounds, CAMT, s["sf"], s["T_cur"])
    kf = LK.KeyFrame(F, s["ex"].inv_sigma2(), LOG_SF)
    points = dict(pos=s["pos"], normal=normal, max_distance=max_d, min_distance=min_d, desc=s["desc_last"], valid=s["valid"])
    return s, rng, g, kf, points, normal, max_d, min_d


@pytest.mark.parametrize("seed,bounds", [(81, (0.0, float(W), 0.0, float(H))), (82, KF_BOUNDS)])
def test_keyframe_grid_queries_literal_vs_oracle(seed, bounds):
    s, rng, g, kf, points, *_ = _setup(seed, bounds)
    assert (kf.mnMinX, kf.mnMaxX, kf.mnMinY, kf.mnMaxY) == tuple(int(np.float32(b)) for b in (bounds[0], bounds[1], bounds[2], bounds[3]))
    for _ in range(300):
        x, y, r = float(rng.uniform(-10, W + 10)), float(rng.uniform(-10, H + 10)), float(rng.uniform(1, 40))
        assert kf.GetFeaturesInArea(x, y, r) == g.features_in_area(x, y, r).tolist()


@pytest.mark.parametrize("seed,bounds,th", [(83, (0.0, float(W), 0.0, float(H)), 3.0), (84, KF_BOUNDS, 3.0), (85, KF_BOUNDS, 6.0)])
def test_fuse_literal_vs_oracle(seed, bounds, th):
    s, rng, g, kf, points, normal, max_d, min_d = _setup(seed, bounds)
    ref, nref = O.fuse(g, s["ur"], s["d"], s["sf"], s["ex"].inv_sigma2(), CAM, s["T_cur"], LOG_SF, NL, s["pos"], normal, max_d, min_d,
                       s["desc_last"], s["valid"], th)
    got, ngot = LK.fuse(kf, s["T_cur"], points, th)
    assert ngot == nref and np.array_equal(got, ref) and nref > 60


@pytest.mark.parametrize("seed,bounds,th", [(86, (0.0, float(W), 0.0, float(H)), 4.0), (87, KF_BOUNDS, 4.0), (88, KF_BOUNDS, 10.0)])
def test_sim3_projection_and_fuse_literal_vs_oracle(seed, bounds, th):
    s, rng, g, kf, points, normal, max_d, min_d = _setup(seed, bounds)
    Scw = s["T_cur"].copy(); Scw *= np.float32(1.07)
    kf_matched = (rng.random(len(s["k"])) < 0.1).astype(np.uint8)
    ref0, n0 = O.sim3_projection(0, g, s["d"], s["sf"], CAM, Scw, LOG_SF, NL, s["pos"], normal, max_d, min_d, s["desc_last"], s["valid"], kf_matched, th)
    got0, m0 = LK.search_by_projection_sim3(kf, Scw, points, kf_matched, th)
    assert m0 == n0 and np.array_equal(got0, ref0) and n0 > 60
    ref1, n1 = O.sim3_projection(1, g, s["d"], s["sf"], CAM, Scw, LOG_SF, NL, s["pos"], normal, max_d, min_d, s["desc_last"], s["valid"], None, th)
    got1, m1 = LK.fuse_sim3(kf, Scw, points, th)
    assert m1 == n1 and np.array_equal(got1, ref1) and n1 > 60


@pytest.mark.parametrize("seed,th,kfb", [(90, 7.5, False), (91, 3.0, True)])
def test_search_by_sim3_literal_vs_oracle(seed, th, kfb):
    from tests.test_matchers import _se3
    rng = np.random.default_rng(seed)
    ex = O.Extractor()
    sf = ex.scale_factors()
    n = 500
    Pw = np.stack([rng.uniform(-8, 8, n), rng.uniform(-5, 5, n), rng.uniform(4, 35, n)], axis=1)
    T1 = _se3(0.0, [0, 0, 0]).astype(np.float64)
    T2 = _se3(4.0, [-0.5, 0.03, 0.1]).astype(np.float64)
    base_d = rng.integers(0, 256, (n, 32)).astype(np.uint8)
    bounds = KF_BOUNDS if kfb else (0.0, float(W), 0.0, float(H))

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
        return k[perm], d[perm], pos[perm], mx[perm], mn[perm], valid[perm]

    k1, d1, pos1, mx1, mn1, v1 = keyframe(T1, seed * 7 + 1, 120)
    k2, d2, pos2, mx2, mn2, v2 = keyframe(T2, seed * 7 + 2, 90)
    R12 = (T1[:, :3] @ T2[:, :3].T)
    t12 = T1[:, 3] - R12 @ T2[:, 3]
    s12 = np.float32(1.02)
    g1, g2 = O.Grid(k1, *bounds, keyframe=True), O.Grid(k2, *bounds, keyframe=True)
    pts1 = (pos1, mx1, mn1, d1, v1); pts2 = (pos2, mx2, mn2, d2, v2)
    ref, nref = O.search_by_sim3(g1, d1, T1.astype(np.float32), pts1, g2, d2, T2.astype(np.float32), pts2, sf, CAM, LOG_SF, NL,
                                 s12, R12.astype(np.float32), t12.astype(np.float32), th)
    kf1 = LK.KeyFrame(LM.Frame(k1, d1, None, bounds, CAMT, sf), ex.inv_sigma2(), LOG_SF)
    kf2 = LK.KeyFrame(LM.Frame(k2, d2, None, bounds, CAMT, sf), ex.inv_sigma2(), LOG_SF)
    as_dict = lambda p: dict(pos=p[0], max_distance=p[1], min_distance=p[2], desc=p[3], valid=p[4])
    got, ngot = LK.search_by_sim3(kf1, T1.astype(np.float32), as_dict(pts1), kf2, T2.astype(np.float32), as_dict(pts2), s12,
                                  R12.astype(np.float32), t12.astype(np.float32), th)
    assert ngot == nref and np.array_equal(got, ref) and nref > (60 if th > 5 else 15)
