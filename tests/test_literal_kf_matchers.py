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
