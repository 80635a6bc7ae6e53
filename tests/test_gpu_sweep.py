"""Wider GPU parity sweep: odd geometries, feature budgets that stress every quadtree path (few / many nodes, the
"largest first" phase, levels without cells), several scenes.  Everything is compared bit-exactly with the oracle."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from orbslam2_amd import synth

pytestmark = pytest.mark.gpu

CASES = [
    # (width, height, nfeatures, seed)
    (777, 333, 1500, 1), (1241, 376, 500, 2), (1241, 376, 4000, 3), (640, 480, 3000, 4), (200, 160, 300, 5),
    (131, 97, 200, 6), (1920, 1080, 2000, 7), (1241, 376, 37, 8), (410, 1000, 800, 9), (1241, 376, 2000, 10),
    (1241, 376, 2000, 11), (96, 64, 100, 12), (64, 62, 50, 13),
    (3840, 2160, 8000, 14), (2560, 1440, 15000, 15),  # 4K frame; feature budget far above the reference's configurations
]


@pytest.mark.parametrize("w,h,nf,seed", CASES)
def test_stereo_frame_bit_exact(w, h, nf, seed):
    from orbslam2_amd import api
    left, right = synth.stereo_pair(w, h, seed=seed)
    fx, bf = 0.6 * w, 0.25 * w
    try:
        ctx = api.Context(width=w, height=h, nfeatures=nf, fx=fx, fy=fx, cx=w / 2, cy=h / 2, bf=bf)
    except api.OrbfeError as e:
        # the forced fallback quadtree kernel (tools/r05_fullsuite.sh: ORBFE_OCTREE=1) keeps its node tables in LDS and refuses
        # quotas beyond that budget (15000 features) loudly; by default such a geometry runs on the bucket-pyramid kernel
        if os.environ.get("ORBFE_OCTREE") == "1" and e.code == api.ERR_UNSUPPORTED:
            pytest.skip(str(e))
        raise
    out = ctx.stereo_frame(left, right)
    exl, exr = O.Extractor(nfeatures=nf), O.Extractor(nfeatures=nf)
    kl, dl = exl.extract(left); kr, dr = exr.extract(right)
    ur, dp, _ = O.stereo_matches(exl, exr, kl, dl, kr, dr, bf, fx)
    for l in range(8):
        for a, b in zip(ctx.fetch_candidates(0, l), exl.level_candidates(l)):
            assert np.array_equal(a, b), "candidates level %d" % l
    assert np.array_equal(out["kps_left"], kl.astype(api.KP_DTYPE)), "left keypoints"
    assert np.array_equal(out["kps_right"], kr.astype(api.KP_DTYPE)), "right keypoints"
    assert np.array_equal(out["desc_left"], dl) and np.array_equal(out["desc_right"], dr)
    assert np.array_equal(out["u_right"], ur) and np.array_equal(out["depth"], dp)
    ctx.close()


def test_noise_image_many_candidates():
    """Pure noise: ~20 % of the pixels are FAST corners; exercises big candidate sets (HBM quadtree path for level 0)."""
    from orbslam2_amd import api
    rng = np.random.default_rng(99)
    img = rng.integers(0, 256, (376, 1241)).astype(np.uint8)
    ctx = api.Context(width=1241, height=376, nfeatures=2000)
    k, d = ctx.extract(img)
    ex = O.Extractor(nfeatures=2000)
    kr, dr = ex.extract(img)
    ncand = [len(ex.level_candidates(l)[0]) for l in range(8)]
    assert ncand[0] > 8192  # above the LDS-resident limit
    assert np.array_equal(k, kr.astype(api.KP_DTYPE)) and np.array_equal(d, dr)
    ctx.close()


@pytest.mark.parametrize("ini,mn", [(10, 7), (30, 7)])
def test_ramp_image_every_pixel_passes_both_polarities(ini, mn):
    """A sawtooth ramp v = (16 x + 5 y) mod 256 (+ a little noise): opposite ring pixels are v +- d with d = 15 .. 48 on all four
    pairs the device's quick test uses, so away from the wrap lines EVERY pixel passes it as dark and as bright.  That fills the
    both-polarity side list of fast_cell_kernel far beyond its 256 entries per cell (scored in batches while the quick test is
    still running) and makes the ordered queue as long as the cell has pixels.  ini = 30: most cells go to the second attempt."""
    from orbslam2_amd import api
    w, h = 640, 360
    yy, xx = np.mgrid[0:h, 0:w]
    rng = np.random.default_rng(5)
    img = ((16 * xx + 5 * yy + rng.integers(0, 3, (h, w))) % 256).astype(np.uint8)
    # the fraction of pixels that pass the four-pair test in both polarities at the test's thresholds (numpy restatement)
    v = img.astype(np.int32)[3:-3, 3:-3]
    ring = lambda dx, dy: img.astype(np.int32)[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx]
    pairs = [(ring(0, 3), ring(0, -3)), (ring(2, 2), ring(-2, -2)), (ring(3, 0), ring(-3, 0)), (ring(2, -2), ring(-2, 2))]
    dark = np.ones_like(v, bool); bright = np.ones_like(v, bool)
    for a, b in pairs:
        dark &= np.minimum(a, b) < v - mn
        bright &= np.maximum(a, b) > v + mn
    assert (dark & bright).mean() > 0.5
    kw = dict(nfeatures=800, ini_th_fast=ini, min_th_fast=mn)
    ctx = api.Context(width=w, height=h, **kw)
    k, d = ctx.extract(img)
    ex = O.Extractor(**kw)
    kr, dr = ex.extract(img)
    assert len(kr) > 100
    assert np.array_equal(k, kr.astype(api.KP_DTYPE)) and np.array_equal(d, dr)
    ctx.close()


@pytest.mark.parametrize("square,blur", [(24, False), (37, True)])
def test_checkerboard_many_tied_scores(square, blur):
    """A calibration checkerboard (the reference ships one for its camera calibration tool): flat squares, corners in a regular
    lattice, and -- without blur -- thousands of candidates with IDENTICAL FAST scores, so the NMS strictness, the quadtree's
    first-maximum rule and the order of equal responses decide which keypoints survive.  Stereo frame bit-exact."""
    from orbslam2_amd import api
    w, h = 752, 480
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.where(((xx // square) + (yy // square)) % 2 == 0, 225, 30).astype(np.float32)
    if blur:  # soft edges and a brightness ramp: scores differ, many cells need the minTh attempt
        k = np.array([1, 4, 6, 4, 1], np.float32) / 16
        img = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, img)
        img = np.apply_along_axis(lambda c: np.convolve(c, k, mode="same"), 0, img)
        img = img * (0.55 + 0.45 * xx / w)
    left = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    right = np.roll(left, -9, axis=1)
    fx, bf = 460.0, 50.0
    ctx = api.Context(width=w, height=h, nfeatures=1200, fx=fx, fy=fx, cx=w / 2, cy=h / 2, bf=bf)
    out = ctx.stereo_frame(left, right)
    exl, exr = O.Extractor(nfeatures=1200), O.Extractor(nfeatures=1200)
    kl, dl = exl.extract(left); kr, dr = exr.extract(right)
    ur, dp, _ = O.stereo_matches(exl, exr, kl, dl, kr, dr, bf, fx)
    assert len(kl) > 300
    if not blur:
        assert len(np.unique(kl["response"])) < len(kl) // 8  # heavy ties
    assert np.array_equal(out["kps_left"], kl.astype(api.KP_DTYPE)) and np.array_equal(out["kps_right"], kr.astype(api.KP_DTYPE))
    assert np.array_equal(out["desc_left"], dl) and np.array_equal(out["desc_right"], dr)
    assert np.array_equal(out["u_right"], ur) and np.array_equal(out["depth"], dp)
    ctx.close()


def test_reuse_context_many_frames():
    from orbslam2_amd import api
    ctx = api.Context(width=480, height=320, nfeatures=700, fx=400.0, fy=400.0, cx=240.0, cy=160.0, bf=100.0)
    ex = O.Extractor(nfeatures=700)
    for seed in range(20, 26):
        left = synth.mono_image(480, 320, seed=seed)
        k, d = ctx.extract(left)
        kr, dr = ex.extract(left)
        assert np.array_equal(k, kr.astype(api.KP_DTYPE)) and np.array_equal(d, dr), seed
    ctx.close()


@pytest.mark.parametrize("patch,nf", [(40, 2000), (24, 1000), (90, 3000), (160, 4000)])
def test_clustered_candidates_deep_quadtree(patch, nf):
    """All corners inside one small textured square and a generous quota: the quadtree must split far below the
    bucket depth of the pyramid kernel (its slow path: counting sort by bucket + path replay)."""
    from orbslam2_amd import api
    rng = np.random.default_rng(patch)
    img = np.full((376, 1241), 128, np.uint8)
    img[150:150 + patch, 600:600 + patch] = rng.integers(0, 256, (patch, patch)).astype(np.uint8)
    img[40:40 + patch // 2, 100:100 + patch // 2] = rng.integers(0, 256, (patch // 2, patch // 2)).astype(np.uint8)
    ctx = api.Context(width=1241, height=376, nfeatures=nf)
    k, d = ctx.extract(img)
    ex = O.Extractor(nfeatures=nf)
    kr, dr = ex.extract(img)
    assert len(kr) > 20
    assert np.array_equal(k, kr.astype(api.KP_DTYPE)) and np.array_equal(d, dr)
    ctx.close()


def test_generic_quadtree_kernel_still_matches(monkeypatch):
    """ORBFE_OCTREE=1 routes the quadtree to the generic node-parallel kernel (the fallback beyond the bucket-pyramid kernel's limits)."""
    from orbslam2_amd import api
    left = synth.mono_image(640, 480, seed=5)
    monkeypatch.delenv("ORBFE_OCTREE", raising=False)  # tools/r05_fullsuite.sh may have forced it for the whole run
    ctx = api.Context(width=640, height=480, nfeatures=1200)
    assert ctx.quadtree_kernel() == 3  # the default
    k, d = ctx.extract(left)
    ctx.close()
    monkeypatch.setenv("ORBFE_OCTREE", "1")
    ctx = api.Context(width=640, height=480, nfeatures=1200)
    assert ctx.quadtree_kernel() == 1
    k1, d1 = ctx.extract(left)
    ctx.close()
    ex = O.Extractor(nfeatures=1200)
    kr, dr = ex.extract(left)
    assert np.array_equal(k, kr.astype(api.KP_DTYPE)) and np.array_equal(d, dr)
    assert np.array_equal(k1, kr.astype(api.KP_DTYPE)) and np.array_equal(d1, dr)


def test_lds_resize_kernel_still_matches(monkeypatch):
    """ORBFE_PYR_LDS=1 keeps the pyramid on pyr_resize_kernel (LDS-staged; the path for levels whose 4-column words reach
    further than 8 source bytes) instead of pyr_resize_direct_kernel: same pyramid, same keypoints."""
    from orbslam2_amd import api
    left = synth.mono_image(1241, 376, seed=11)
    ex = O.Extractor(nfeatures=2000)
    kr, dr = ex.extract(left)
    for lds in ("1", "0"):
        monkeypatch.setenv("ORBFE_PYR_LDS", lds)
        ctx = api.Context(width=1241, height=376, nfeatures=2000)
        k, d = ctx.extract(left)
        for l in range(8):
            assert np.array_equal(ctx.fetch_pyramid(0, l), ex.pyramid_level(l)), (lds, l)
        assert np.array_equal(k, kr.astype(api.KP_DTYPE)) and np.array_equal(d, dr), lds
        ctx.close()


@pytest.mark.parametrize("w,h,nlevels", [(48, 45, 2), (65, 49, 3), (257, 52, 4), (60, 300, 5), (1029, 70, 6)])
def test_pyramid_of_narrow_and_short_images(w, h, nlevels):
    """Level sizes around the resize kernel's granules: fewer than 64 four-column words per row (one partly filled strip), 64 n + 1
    words, fewer extended rows than one workgroup's band -- raw and blurred pyramid, keypoints and descriptors against the oracle."""
    from orbslam2_amd import api
    kw = dict(nfeatures=300, nlevels=nlevels)
    try:
        ex = O.Extractor(**kw)
    except ValueError:
        pytest.skip("parameter set rejected by the oracle")
    img = synth.mono_image(w, h, seed=w + h)
    try:
        ctx = api.Context(width=w, height=h, **kw)
    except api.OrbfeError as e:
        if e.code == api.ERR_UNSUPPORTED:
            pytest.skip("geometry refused by orbfe_create: %s" % e)
        raise
    k, d = ctx.extract(img)
    kr, dr = ex.extract(img)
    for l in range(nlevels):
        assert np.array_equal(ctx.fetch_pyramid(0, l), ex.pyramid_level(l)), l
        assert np.array_equal(ctx.fetch_pyramid(0, l, blurred=True), O.gaussian7(ex.pyramid_level(l))), l
    assert np.array_equal(k, kr.astype(api.KP_DTYPE)) and np.array_equal(d, dr)
    ctx.close()


def test_blur_fused_and_separate_launches_match(monkeypatch):
    """By default level l - 1 is blurred inside the launch that resizes it into level l; ORBFE_NO_FUSE=1 blurs every level in one
    launch after the pyramid.  Both must give the oracle's blurred pyramid (and ORBFE_PYR_LDS=1, which leaves nothing to fuse;
    ORBFE_NO_TAIL=1: the last three levels as single launches, which batches of 64 images and more do by themselves)."""
    from orbslam2_amd import api
    left = synth.mono_image(752, 480, seed=3)
    ex = O.Extractor(nfeatures=1200)
    kr, dr = ex.extract(left)
    for env in ({"ORBFE_NO_FUSE": "1"}, {"ORBFE_NO_FUSE": "0"}, {"ORBFE_PYR_LDS": "1"}, {"ORBFE_NO_TAIL": "1"}, {"ORBFE_NO_TAIL": "1", "ORBFE_NO_FUSE": "1"}):
        for k in ("ORBFE_NO_FUSE", "ORBFE_PYR_LDS", "ORBFE_NO_TAIL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = api.Context(width=752, height=480, nfeatures=1200)
        k, d = ctx.extract(left)
        for l in range(8):
            assert np.array_equal(ctx.fetch_pyramid(0, l, blurred=True), O.gaussian7(ex.pyramid_level(l))), (env, l)
        assert np.array_equal(k, kr.astype(api.KP_DTYPE)) and np.array_equal(d, dr), env
        ctx.close()


@pytest.mark.parametrize("sf,nlevels", [(2.6, 3), (2.0, 4), (1.95, 4), (1.05, 6)])
def test_resize_scale_factors_around_the_direct_kernel_limit(sf, nlevels):
    """Scale factors whose words need more than 8 source bytes (2.6; some levels of 2.0, where level sizes round to a ratio
    just above 2) take the LDS kernel, the others the direct one: orbfe_create decides per level from the column table."""
    from orbslam2_amd import api
    left = synth.mono_image(1000, 700, seed=int(sf * 100))
    ex = O.Extractor(nfeatures=1000, scale_factor=sf, nlevels=nlevels)
    kr, dr = ex.extract(left)
    ctx = api.Context(width=1000, height=700, nfeatures=1000, scale_factor=sf, nlevels=nlevels)
    k, d = ctx.extract(left)
    for l in range(nlevels):
        assert np.array_equal(ctx.fetch_pyramid(0, l), ex.pyramid_level(l)), l
    assert np.array_equal(k, kr.astype(api.KP_DTYPE)) and np.array_equal(d, dr)
    ctx.close()


def test_stereo_row_list_overflow():
    """All keypoints inside a thin horizontal band: the per-row candidate lists exceed their fixed capacity and the
    stereo search must fall back to scanning every right keypoint -- same matches as the oracle."""
    from orbslam2_amd import api
    w, h, nf = 1241, 376, 2000
    rng = np.random.default_rng(7)
    tex = rng.integers(0, 256, (44, w + 40)).astype(np.uint8)
    left = np.full((h, w), 120, np.uint8); right = left.copy()
    left[160:204, :] = tex[:, 40:40 + w]
    right[160:204, :] = tex[:, 28:28 + w]  # 12 px disparity
    fx, bf = 718.856, 386.1448
    ctx = api.Context(width=w, height=h, nfeatures=nf, fx=fx, fy=fx, cx=607.0, cy=185.0, bf=bf)
    out = ctx.stereo_frame(left, right)
    exl, exr = O.Extractor(nfeatures=nf), O.Extractor(nfeatures=nf)
    kl, dl = exl.extract(left); kr, dr = exr.extract(right)
    ur, dp, m = O.stereo_matches(exl, exr, kl, dl, kr, dr, bf, fx)
    rows = np.zeros(h, int)
    for k in kr:
        r = 2.0 * 1.2 ** int(k["octave"])
        rows[max(0, int(np.floor(k["y"] - r))):min(h - 1, int(np.ceil(k["y"] + r))) + 1] += 1
    assert rows.max() > 4 * len(kr) * 10 // h  # the capacity formula of orbfe_create
    assert m > 100
    assert np.array_equal(out["kps_right"], kr.astype(api.KP_DTYPE))
    assert np.array_equal(out["u_right"], ur) and np.array_equal(out["depth"], dp)
    ctx.close()


def test_sad_window_left_of_the_level_image_q12():
    """Scale factor 2.22: a right keypoint one octave finer than its left partner can lie 19 / 2.22 < 10 px from the left border of
    the left keypoint's level, so the SAD band (columns cr - 10 ..) starts outside the image.  The reference's guard misses that
    (iniu = scaleduR0 + L - w is never negative) and cv::Mat::colRange throws; contract Q12: the keypoint stays unmatched, in the
    oracle and in the product alike (found by tools/soak.py, SOAK_GEOM=1 SOAK_SEED=131000, case 329)."""
    from orbslam2_amd import api
    w, h = 451, 278
    kw = dict(nfeatures=729, ini_th_fast=33, min_th_fast=19, scale_factor=float(np.float32(2.218409776687622)), nlevels=6)
    left, right = synth.stereo_pair(w, h, seed=131000 + 2000 + 329)
    fx, bf = 0.7 * w, 0.2 * w
    ctx = api.Context(width=w, height=h, fx=fx, fy=fx, cx=w / 2, cy=h / 2, bf=bf, **kw)
    out = ctx.stereo_frame(left, right)
    exl, exr = O.Extractor(**kw), O.Extractor(**kw)
    kl, dl = exl.extract(left); kr, dr = exr.extract(right)
    ur, dp, m = O.stereo_matches(exl, exr, kl, dl, kr, dr, bf, fx)
    assert m > 100
    assert np.array_equal(out["kps_left"], kl.astype(api.KP_DTYPE)) and np.array_equal(out["kps_right"], kr.astype(api.KP_DTYPE))
    assert np.array_equal(out["u_right"], ur) and np.array_equal(out["depth"], dp)
    ctx.close()


def test_stereo_row_lists_longer_than_their_lds_staging():
    """Keypoints inside a 150-row band: the rows' candidate lists hold 100-200 entries -- more than the 64 the row-list waves stage
    in LDS (the rest go straight to the global list), fewer than the lists' capacity -- same matches as the oracle."""
    from orbslam2_amd import api
    w, h, nf = 1241, 376, 2000
    rng = np.random.default_rng(11)
    tex = rng.integers(0, 256, (150, w + 40)).astype(np.uint8)
    left = np.full((h, w), 120, np.uint8); right = left.copy()
    left[110:260, :] = tex[:, 40:40 + w]
    right[110:260, :] = tex[:, 31:31 + w]  # 9 px disparity
    fx, bf = 718.856, 386.1448
    ctx = api.Context(width=w, height=h, nfeatures=nf, fx=fx, fy=fx, cx=607.0, cy=185.0, bf=bf)
    out = ctx.stereo_frame(left, right)
    exl, exr = O.Extractor(nfeatures=nf), O.Extractor(nfeatures=nf)
    kl, dl = exl.extract(left); kr, dr = exr.extract(right)
    ur, dp, m = O.stereo_matches(exl, exr, kl, dl, kr, dr, bf, fx)
    rows = np.zeros(h, int)
    for k in kr:
        r = 2.0 * 1.2 ** int(k["octave"])
        rows[max(0, int(np.floor(k["y"] - r))):min(h - 1, int(np.ceil(k["y"] + r))) + 1] += 1
    cap = 4 * len(kr) * 10 // h  # the capacity formula of orbfe_create
    assert 64 < rows.max() <= cap and (rows > 64).sum() > 50
    assert m > 50
    assert np.array_equal(out["kps_right"], kr.astype(api.KP_DTYPE))
    assert np.array_equal(out["u_right"], ur) and np.array_equal(out["depth"], dp)
    ctx.close()


@pytest.mark.parametrize("kw", [
    dict(nlevels=5, scale_factor=1.5, nfeatures=1500),
    dict(nlevels=12, scale_factor=1.1, nfeatures=3000),
    dict(nlevels=3, scale_factor=2.0, nfeatures=800, ini_th_fast=30, min_th_fast=12),
    dict(nlevels=8, scale_factor=1.2, nfeatures=1000, ini_th_fast=12, min_th_fast=5),
    dict(nlevels=1, scale_factor=1.2, nfeatures=500),
])
def test_extractor_parameter_variants(kw):
    """Pyramid depth, scale factor and FAST thresholds other than the ORB-SLAM2 defaults (stereo frame, bit-exact)."""
    from orbslam2_amd import api
    w, h = 800, 450
    left, right = synth.stereo_pair(w, h, seed=77)
    fx, bf = 520.0, 180.0
    ctx = api.Context(width=w, height=h, fx=fx, fy=fx, cx=w / 2, cy=h / 2, bf=bf, **kw)
    out = ctx.stereo_frame(left, right)
    exl, exr = O.Extractor(**kw), O.Extractor(**kw)
    kl, dl = exl.extract(left); kr, dr = exr.extract(right)
    ur, dp, _ = O.stereo_matches(exl, exr, kl, dl, kr, dr, bf, fx)
    assert np.array_equal(out["kps_left"], kl.astype(api.KP_DTYPE)) and np.array_equal(out["kps_right"], kr.astype(api.KP_DTYPE))
    assert np.array_equal(out["desc_left"], dl) and np.array_equal(out["desc_right"], dr)
    assert np.array_equal(out["u_right"], ur) and np.array_equal(out["depth"], dp)
    assert len(kl) > 100
    ctx.close()


def _random_case(i):
    rng = np.random.default_rng(1000 + i)
    w = int(rng.integers(70, 1500)); h = int(rng.integers(70, 900))
    nlevels = int(rng.integers(1, 11))
    sf = float(np.float32(rng.choice([1.1, 1.15, 1.2, 1.25, 1.3, 1.5, 1.8, 2.0])))
    nf = int(rng.integers(30, 4500))
    ini = int(rng.integers(8, 40)); mn = int(rng.integers(3, ini + 1))
    return dict(w=w, h=h, kw=dict(nlevels=nlevels, scale_factor=sf, nfeatures=nf, ini_th_fast=ini, min_th_fast=mn))


@pytest.mark.parametrize("i", range(32))
def test_random_geometry_and_parameters(i):
    """Randomised (fixed seeds) image sizes and extractor parameters: stereo frame bit-exact against the oracle, whichever quadtree
    kernel the geometry selects.  Catches rounding / indexing cases no hand-picked configuration covers."""
    from orbslam2_amd import api
    c = _random_case(i)
    w, h, kw = c["w"], c["h"], c["kw"]
    try:
        exl, exr = O.Extractor(**kw), O.Extractor(**kw)
    except ValueError:
        pytest.skip("parameter set rejected by the oracle")
    left, right = synth.stereo_pair(w, h, seed=2000 + i)
    fx, bf = 0.7 * w, 0.2 * w
    try:
        ctx = api.Context(width=w, height=h, fx=fx, fy=fx, cx=w / 2, cy=h / 2, bf=bf, **kw)
    except api.OrbfeError as e:
        if e.code == api.ERR_UNSUPPORTED:
            pytest.skip(str(e))
        raise
    out = ctx.stereo_frame(left, right)
    kl, dl = exl.extract(left); kr, dr = exr.extract(right)
    ur, dp, _ = O.stereo_matches(exl, exr, kl, dl, kr, dr, bf, fx)
    assert np.array_equal(out["kps_left"], kl.astype(api.KP_DTYPE)), (c, ctx.quadtree_kernel())
    assert np.array_equal(out["kps_right"], kr.astype(api.KP_DTYPE)), (c, ctx.quadtree_kernel())
    assert np.array_equal(out["desc_left"], dl) and np.array_equal(out["desc_right"], dr), c
    assert np.array_equal(out["u_right"], ur) and np.array_equal(out["depth"], dp), c
    ctx.close()


@pytest.mark.parametrize("hp,edge", [(15, 19), (12, 19), (17, 21), (9, 24)])
def test_other_patch_sizes(hp, edge):
    """HALF_PATCH_SIZE other than the reference's 15 (and a wider edge threshold): describe_kernel then stages the IC_Angle patch
    by aligned words and runs the generic moment loop instead of the 128-bit / byte-dot-product path; the blurred patch and the
    descriptor are the same code.  Stereo frame bit-exact against the oracle."""
    from orbslam2_amd import api
    w, h = 512, 300
    kw = dict(nfeatures=900, half_patch_size=hp, patch_size=2 * hp + 1, edge_threshold=edge)
    exl, exr = O.Extractor(**kw), O.Extractor(**kw)
    left, right = synth.stereo_pair(w, h, seed=4100 + hp)
    fx, bf = 0.7 * w, 0.2 * w
    ctx = api.Context(width=w, height=h, fx=fx, fy=fx, cx=w / 2, cy=h / 2, bf=bf, **kw)
    out = ctx.stereo_frame(left, right)
    kl, dl = exl.extract(left); kr, dr = exr.extract(right)
    ur, dp, _ = O.stereo_matches(exl, exr, kl, dl, kr, dr, bf, fx)
    assert len(kl) > 300
    assert np.array_equal(out["kps_left"], kl.astype(api.KP_DTYPE)) and np.array_equal(out["kps_right"], kr.astype(api.KP_DTYPE))
    assert np.array_equal(out["desc_left"], dl) and np.array_equal(out["desc_right"], dr)
    assert np.array_equal(out["u_right"], ur) and np.array_equal(out["depth"], dp)
    ctx.close()


@pytest.mark.parametrize("kind", ["low_contrast", "checker4", "checker3_mixed", "stripes"])
def test_fast_threshold_fallback_cells(kind):
    """ComputeKeyPointsOctTree runs FAST at iniThFAST and, for a cell without keypoints, again at minThFAST
    (src/ORBextractor.cc:803-810).  The kernel does the same per cell (first attempt at iniTh, tile re-staged for the second):
    low-contrast scenes send most cells through the fallback; periodic patterns give hundreds of equal scores per cell
    (no strict local maximum at iniTh although many pixels pass it), the case where the first attempt's flags have
    overwritten the tile."""
    from orbslam2_amd import api
    w, h, nf = 640, 360, 1500
    base_l, base_r = synth.stereo_pair(w, h, seed=91)
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "low_contrast":
        f = lambda a: np.clip((a.astype(np.float32) - 128.0) * 0.12 + 128.0, 0, 255).astype(np.uint8)
        left, right = f(base_l), f(base_r)
    elif kind == "checker4":
        left = (((xx // 4 + yy // 4) & 1) * 60 + 90).astype(np.uint8)
        right = np.roll(left, -8, axis=1)
    elif kind == "checker3_mixed":
        c = (((xx // 3 + yy // 3) & 1) * 40 + 100).astype(np.uint8)
        left = np.where(xx < w // 2, c, base_l).astype(np.uint8)
        right = np.where(xx < w // 2, np.roll(c, -6, axis=1), base_r).astype(np.uint8)
    else:
        s = (((xx // 5) & 1) * 50 + ((yy // 7) & 1) * 25 + 80).astype(np.uint8)
        left, right = s, np.roll(s, -10, axis=1)
    fx, bf = 0.6 * w, 0.25 * w
    try:
        ctx = api.Context(width=w, height=h, nfeatures=nf, fx=fx, fy=fx, cx=w / 2, cy=h / 2, bf=bf)
    except api.OrbfeError as e:
        # the forced fallback quadtree kernel (tools/r05_fullsuite.sh: ORBFE_OCTREE=1) keeps its node tables in LDS and refuses
        # quotas beyond that budget (15000 features) loudly; by default such a geometry runs on the bucket-pyramid kernel
        if os.environ.get("ORBFE_OCTREE") == "1" and e.code == api.ERR_UNSUPPORTED:
            pytest.skip(str(e))
        raise
    out = ctx.stereo_frame(left, right)
    exl, exr = O.Extractor(nfeatures=nf), O.Extractor(nfeatures=nf)
    kl, dl = exl.extract(left); kr, dr = exr.extract(right)
    ur, dp, _ = O.stereo_matches(exl, exr, kl, dl, kr, dr, bf, fx)
    for l in range(8):
        for a, b in zip(ctx.fetch_candidates(0, l), exl.level_candidates(l)):
            assert np.array_equal(a, b), "candidates level %d" % l
    assert np.array_equal(out["kps_left"], kl.astype(api.KP_DTYPE)) and np.array_equal(out["kps_right"], kr.astype(api.KP_DTYPE))
    assert np.array_equal(out["desc_left"], dl) and np.array_equal(out["desc_right"], dr)
    assert np.array_equal(out["u_right"], ur) and np.array_equal(out["depth"], dp)
    if kind == "low_contrast":
        sc = np.concatenate([exl.level_candidates(l)[2] for l in range(8)])
        assert (sc < 20).mean() > 0.3 and (sc >= 20).any() and len(kl) > 200  # both kinds of cells occur
    ctx.close()
