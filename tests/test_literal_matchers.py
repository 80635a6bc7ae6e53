"""The C oracle's matcher restatements (oracle/orb_oracle_match.c: candidate keys + replayed accept rules, the same scheme the
HIP host side uses) against a SECOND, independent statement: oracle/literal_matchers.py, the reference's loops transcribed
literally (mGrid as vector of vectors, GetFeaturesInArea as a triple loop, best / second-best tracked inside the loop).
Breaks the "resolve logic is compared with its own twin" problem of round 1: nothing here shares code with orbfe_match.hip.
CPU only."""
import numpy as np
import pytest

from oracle import literal_matchers as LM
from oracle import oracle as O
from tests.test_matchers import BF, CAM, CX, CY, FX, FY, H, LOG_SF, NL, W, _scene, _se3

CAMT = (FX, FY, CX, CY, BF, BF / FX)


def _obs_list(cur_has_obs):
    return [1 if h else None for h in cur_has_obs]  # keypoint holds a point with Observations() == 1, or NULL


def test_grid_and_features_in_area_literal_vs_oracle():
    s = _scene(61)
    g = O.Grid(s["k"], *s["bounds"])
    F = LM.Frame(s["k"], s["d"], s["ur"], s["bounds"], CAMT, s["sf"])
    rng = np.random.default_rng(62)
    for _ in range(300):
        x, y, r = rng.uniform(-40, W + 40), rng.uniform(-40, H + 40), float(rng.uniform(0.5, 120))
        lo, hi = int(rng.integers(-1, 6)), int(rng.integers(-1, 8))
        assert F.get_features_in_area(x, y, r, lo, hi) == g.features_in_area(x, y, r, lo, hi).tolist()


@pytest.mark.parametrize("seed,th,mono,ori,T", [(63, 7.0, False, True, None), (64, 15.0, True, True, None), (65, 14.0, False, False, (-1.0, [0.0, 0.0, 0.4])),
                                                (66, 3.0, False, True, (1.0, [0.0, 0.0, -0.5]))])
def test_search_by_projection_last_literal_vs_oracle(seed, th, mono, ori, T):
    s = _scene(seed, n_last=700, n_distract=350)
    if T is not None:
        s["T_cur"] = _se3(*T)  # backward / forward motion beyond the baseline: the one-sided level windows
    ur = None if mono else s["ur"]
    g = O.Grid(s["k"], *s["bounds"])
    ref, nref = O.search_by_projection_last(g, ur, s["d"], s["sf"], CAM, s["T_cur"], s["T_last"], s["pos"], s["desc_last"], s["valid"],
                                            s["obs"], s["octave"], s["angle"], s["cur_has_obs"], th, mono, ori)
    F = LM.Frame(s["k"], s["d"], ur, s["bounds"], CAMT, s["sf"], s["T_cur"])
    last = dict(pos=s["pos"], desc=s["desc_last"], valid=s["valid"], obs=s["obs"], octave=s["octave"], angle=s["angle"])
    got, ngot = LM.search_by_projection_last(F, s["T_last"], last, _obs_list(s["cur_has_obs"]), th, mono, ori)
    assert ngot == nref and np.array_equal(got, ref)
    assert nref > 30


def test_frustum_and_search_by_projection_points_literal_vs_oracle():
    s = _scene(67, n_last=900, n_distract=300)
    rng = s["rng"]
    n = len(s["pos"])
    normal = s["pos"] / np.linalg.norm(s["pos"], axis=1, keepdims=True) + rng.normal(0, 0.35, (n, 3))
    normal = (normal / np.linalg.norm(normal, axis=1, keepdims=True)).astype(np.float32)
    dist0 = np.linalg.norm(s["pos"], axis=1).astype(np.float32)
    max_d = (dist0 * rng.uniform(0.9, 3.0, n)).astype(np.float32); min_d = (max_d / np.float32(1.2 ** 7)).astype(np.float32)
    tp = O.is_in_frustum(s["T_cur"], CAM, s["bounds"], s["pos"], normal, max_d, min_d, 0.5, LOG_SF, NL)
    F = LM.Frame(s["k"], s["d"], s["ur"], s["bounds"], CAMT, s["sf"], s["T_cur"])
    points = []
    for i in range(n):
        t = LM.is_in_frustum(F, s["pos"][i], normal[i], max_d[i], min_d[i], 0.5, LOG_SF)
        assert (t is not None) == bool(tp["in_view"][i]), i
        if t is None:
            points.append(None)
            continue
        for f in ("proj_x", "proj_y", "proj_xr", "level", "view_cos"):
            assert t[f] == tp[f][i], (i, f, t[f], tp[f][i])
        t["desc"] = s["desc_last"][i]; t["obs"] = int(s["obs"][i])
        points.append(t)
    assert tp["in_view"].sum() > 200
    g = O.Grid(s["k"], *s["bounds"])
    for th, ratio in ((1.0, 0.8), (3.0, 0.8), (5.0, 0.6)):
        ref, nref = O.search_by_projection_points(g, s["ur"], s["d"], s["sf"], tp, s["desc_last"], s["obs"], s["cur_has_obs"], th, ratio)
        got, ngot = LM.search_by_projection_points(F, points, _obs_list(s["cur_has_obs"]), th, ratio)
        assert ngot == nref and np.array_equal(got, ref), th
    assert nref > 60


def test_search_by_projection_kf_literal_vs_oracle():
    s = _scene(68, n_last=800, n_distract=300)
    n = len(s["pos"])
    dist0 = np.linalg.norm(s["pos"], axis=1).astype(np.float32)
    max_d = (dist0 * s["rng"].uniform(0.9, 3.0, n)).astype(np.float32); min_d = (max_d / np.float32(1.2 ** 7)).astype(np.float32)
    g = O.Grid(s["k"], *s["bounds"])
    F = LM.Frame(s["k"], s["d"], None, s["bounds"], CAMT, s["sf"], s["T_cur"])
    kf = dict(pos=s["pos"], desc=s["desc_last"], valid=s["valid"], angle=s["angle"], max_distance=max_d, min_distance=min_d)
    for th, od, ori in ((10.0, 100, True), (3.0, 64, True), (8.0, 80, False)):
        ref, nref = O.search_by_projection_kf(g, s["d"], s["sf"], CAM, s["T_cur"], LOG_SF, NL, s["pos"], s["desc_last"], s["valid"], s["angle"],
                                              max_d, min_d, s["cur_has_obs"], th, od, ori)
        got, ngot = LM.search_by_projection_kf(F, kf, s["cur_has_obs"], th, od, ori, LOG_SF)
        assert ngot == nref and np.array_equal(got, ref), (th, od)
    assert nref > 20


def test_search_for_initialization_literal_vs_oracle():
    s = _scene(69, n_last=1000, n_distract=250)
    n = len(s["k"])
    k1 = s["k"].copy(); k1["octave"] = np.where(np.arange(n) % 4 == 0, 1, 0)
    rng = np.random.default_rng(70)
    k2 = k1.copy(); k2["x"] += rng.normal(5, 2, n).astype(np.float32); k2["y"] += rng.normal(-3, 2, n).astype(np.float32)
    k2["octave"] = np.where(np.arange(n) % 7 == 0, 2, 0)
    d2 = s["d"] ^ np.packbits(rng.random((n, 256)) < 0.05, axis=1, bitorder="little")
    # duplicates in frame 2 so that the vMatchedDistance stealing rule (src/ORBmatcher.cc:437-438,460-464) fires
    k2[50:90] = k2[10:50]; d2[50:90] = d2[10:50] ^ np.packbits(rng.random((40, 256)) < 0.01, axis=1, bitorder="little")
    g2 = O.Grid(k2, *s["bounds"])
    prev = np.stack([k1["x"], k1["y"]], axis=1)
    F1 = LM.Frame(k1, s["d"], None, s["bounds"], CAMT, s["sf"])
    F2 = LM.Frame(k2, d2, None, s["bounds"], CAMT, s["sf"])
    for win, ratio, ori in ((100, 0.9, True), (10, 0.6, False), (30, 0.9, True)):
        ref, pm_ref, nref = O.search_for_initialization(k1, s["d"], g2, d2, prev, win, ratio, ori)
        got, pm_got, ngot = LM.search_for_initialization(F1, F2, prev, win, ratio, ori)
        assert ngot == nref and np.array_equal(got, ref) and np.array_equal(pm_got, pm_ref), win
    assert nref > 50


def test_three_maxima_and_descriptor_distance_literal_vs_oracle():
    rng = np.random.default_rng(71)
    for _ in range(300):
        sizes = rng.integers(0, rng.integers(1, 40), 30)
        if rng.random() < 0.3:
            sizes[rng.integers(0, 30, 3)] = sizes.max()  # ties
        assert LM.compute_three_maxima([[0] * int(v) for v in sizes], 30) == O.three_maxima(sizes.tolist())
    d = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    for i in range(0, 64, 2):
        assert LM.descriptor_distance(d[i], d[i + 1]) == O.hamming256(d[i], d[i + 1]) == int(np.unpackbits(d[i] ^ d[i + 1]).sum())
