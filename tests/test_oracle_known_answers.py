"""Known-answer and property tests that pin the CPU oracle from first principles.

The reference ships no tests or golden vectors for this path and OpenCV is
not installed, so parity is UNPINNED by the reference; these tests pin the oracle to the
written contract instead.
"""
import ctypes as C
import hashlib
import math
import struct

import numpy as np
import pytest

from oracle import oracle as O

L = O.lib()
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------- scalars ----------------
def test_cv_round_half_even():
    assert [L.orc_cv_round_f(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]
    assert L.orc_cv_round_d(3.5) == 4 and L.orc_cv_round_d(4.5) == 4


def test_border_reflect101():
    assert [L.orc_border_reflect101(p, 5) for p in (-3, -1, 0, 4, 5, 6, 7)] == [3, 1, 0, 4, 3, 2, 1]
    assert L.orc_border_reflect101(-2, 1) == 0


def test_fast_atan2_cardinals():
    for y, x, deg in ((0, 1, 0), (1, 1, 45), (1, 0, 90), (1, -1, 135), (0, -1, 180), (-1, -1, 225), (-1, 0, 270), (-1, 1, 315)):
        a = L.orc_fast_atan2(float(y), float(x))
        assert abs(a - deg) < 0.3, (y, x, a)
    assert L.orc_fast_atan2(0.0, 0.0) == 0.0
    rng = np.random.default_rng(0)
    for _ in range(2000):
        y, x = rng.integers(-10**6, 10**6, 2)
        a = L.orc_fast_atan2(float(y), float(x))
        ref = math.degrees(math.atan2(y, x)) % 360.0
        d = abs(a - ref)
        assert min(d, 360 - d) < 0.3 and 0.0 <= a <= 360.0


def test_sincos_det_matches_libm_cosf():
    # contract routine == correctly rounded value; libm cosf/sinf (what the reference calls) agrees
    # except on a few near-tie inputs; pin both the accuracy and the mismatch rate
    s, c = C.c_float(), C.c_float()
    libm = C.CDLL("libm.so.6")
    libm.cosf.restype = C.c_float; libm.cosf.argtypes = [C.c_float]
    libm.sinf.restype = C.c_float; libm.sinf.argtypes = [C.c_float]
    factor = np.float32(np.pi / np.float64(np.float32(180.0)))
    deg = np.random.default_rng(1).uniform(0, 360, 20000).astype(np.float32)
    mism = 0
    for a in deg:
        rad = np.float32(a * factor)
        L.orc_sincos_det(float(rad), C.byref(s), C.byref(c))
        rs, rc = np.float32(np.sin(np.float64(rad))), np.float32(np.cos(np.float64(rad)))
        assert s.value == rs and c.value == rc  # == double libm result rounded once
        mism += (libm.sinf(float(rad)) != rs) + (libm.cosf(float(rad)) != rc)  # what the reference calls
    # glibc's sinf/cosf (max error 0.56 ulp, FMA ifunc variants) are NOT correctly rounded: measured here
    # ~1.3 % of angles differ by one ulp.  That is why contract Q4 fixes one deterministic routine; the effect
    # of a 1-ulp change on cvRound(x*b + y*a) is ~1e-5 flipped samples per descriptor.
    assert mism <= 0.03 * 2 * len(deg)


def test_hamming_known():
    a = np.zeros(32, np.uint8); b = np.zeros(32, np.uint8)
    assert O.hamming256(a, b) == 0
    b[:] = 0xFF
    assert O.hamming256(a, b) == 256
    b[:] = 0; b[0] = 0b1011; b[31] = 0x80
    assert O.hamming256(a, b) == 4
    rng = np.random.default_rng(2)
    for _ in range(50):
        a = rng.integers(0, 256, 32).astype(np.uint8); b = rng.integers(0, 256, 32).astype(np.uint8)
        assert O.hamming256(a, b) == int(np.unpackbits(a ^ b).sum())


def test_pattern_checksum():
    p = L.orc_bit_pattern()
    vals = [p[i] for i in range(1024)]
    assert vals[:8] == [8, -3, 9, 5, 4, 2, 7, -12] and sum(vals) == -406
    assert hashlib.sha256(struct.pack("<1024i", *vals)).hexdigest() == \
        "7e645581387b82784797e8adddb9b6f0c12611859fda09ca8a9bec96d767a05f"
    assert max(math.hypot(vals[i], vals[i + 1]) for i in range(0, 1024, 2)) < 18.5  # reach < edgeThreshold 19


def test_gaussian_taps():
    taps = (C.c_int32 * 7)()
    L.orc_gaussian_taps_q8(7, 2.0, taps)
    assert list(taps) == [18, 34, 48, 56, 48, 34, 18] and sum(taps) == 256


# ---------------- extractor tables ----------------
@pytest.mark.parametrize("w,h,nf,sizes,quotas", [
    (1241, 376, 2000, [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)],
     [434, 362, 302, 251, 209, 175, 145, 122]),
    (640, 480, 1000, [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)],
     [217, 181, 151, 126, 105, 87, 73, 60]),
    (752, 480, 1200, [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)],
     [261, 217, 181, 151, 126, 105, 87, 72]),
    (1280, 720, 2500, [(1280, 720), (1067, 600), (889, 500), (741, 417), (617, 347), (514, 289), (429, 241), (357, 201)],
     [543, 452, 377, 314, 262, 218, 182, 152]),
])
def test_level_and_quota_tables(w, h, nf, sizes, quotas):
    ex = O.Extractor(nfeatures=nf)
    assert [ex.level_size(w, h, l) for l in range(8)] == sizes
    assert ex.features_per_level().tolist() == quotas
    assert ex.umax().tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    sf = ex.scale_factors()
    assert sf[1] == np.float32(1.2000000477) and abs(sf[7] - 3.5831816196) < 1e-6
    assert np.array_equal(ex.sigma2(), sf * sf) and np.array_equal(ex.inv_scale_factors(), np.float32(1.0) / sf)


# ---------------- FAST ----------------
def _ring_patch(center, ring_vals):
    img = np.full((7, 7), center, np.uint8)
    for (dx, dy), v in zip(RING, ring_vals):
        img[3 + dy, 3 + dx] = v
    return img


def test_fast_score_hand_made_ring():
    img = _ring_patch(100, [60] * 9 + [100] * 7)
    ptr = img.ctypes.data + 3 * 7 + 3
    assert L.orc_fast_corner_score(ptr, 7, 7) == 39
    assert L.orc_fast_score_closed_form(ptr, 7, 7) == 39
    for t, is_corner in ((20, True), (39, True), (40, False)):
        xs, _, sc = O.fast9_16(img, t)
        assert (len(xs) == 1) == is_corner
        if is_corner:
            assert sc[0] == 39
    # only 8 contiguous: not a corner at any threshold >= 1
    img8 = _ring_patch(100, [60] * 8 + [100] * 8)
    assert len(O.fast9_16(img8, 7)[0]) == 0
    # bright arc
    imgb = _ring_patch(100, [100] * 4 + [150] * 10 + [100] * 2)
    assert L.orc_fast_corner_score(imgb.ctypes.data + 24, 7, 7) == 49


def test_corner_score_equals_closed_form_random():
    rng = np.random.default_rng(3)
    for _ in range(5000):
        amp = int(rng.integers(2, 120))
        img = np.clip(128 + rng.integers(-amp, amp + 1, (7, 7)), 0, 255).astype(np.uint8)
        ptr = img.ctypes.data + 24
        t = int(rng.integers(1, 60))
        assert L.orc_fast_corner_score(ptr, 7, t) == L.orc_fast_score_closed_form(ptr, 7, t)


def test_fast_detector_equals_score_map_formulation():
    """cv::FAST(t, nms) == {p : S(p) >= t and S(p) > S(q) for the 8 neighbours q with S(q) >= t} where S is the
    threshold-independent closed-form score -- the formulation the HIP cell kernel uses."""
    rng = np.random.default_rng(4)
    base = rng.integers(0, 256, (12, 14)).astype(np.uint8)
    img = np.kron(base, np.ones((4, 4), np.uint8))
    img = np.clip(img.astype(np.int32) + rng.integers(-6, 7, img.shape), 0, 255).astype(np.uint8)
    h, w = img.shape
    S = np.zeros((h, w), np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            S[y, x] = L.orc_fast_score_closed_form(img.ctypes.data + y * img.strides[0] + x, img.strides[0], 1)
    for t in (7, 20, 35):
        xs, ys, sc = O.fast9_16(img, t)
        St = np.where(S >= t, S, 0)
        exp = []
        for y in range(3, h - 3):
            for x in range(3, w - 3):
                s = St[y, x]
                if s > 0:
                    nb = St[y - 1:y + 2, x - 1:x + 2].copy(); nb[1, 1] = -1
                    if (s > nb).all():
                        exp.append((x, y, s))
        assert list(zip(xs.tolist(), ys.tolist(), sc.tolist())) == exp, t
        assert len(exp) > 0


# ---------------- resize / blur ----------------
def test_resize_constant_and_ramp():
    img = np.full((40, 60), 77, np.uint8)
    assert (O.resize_linear(img, 50, 33) == 77).all()
    ramp = np.tile(np.arange(0, 240, 2, dtype=np.uint8), (30, 1))  # slope 2 / px, width 120
    out = O.resize_linear(ramp, 100, 25)
    # bilinear of a linear ramp is the ramp sampled at (dx+0.5)*1.2-0.5, up to fixed-point rounding
    exp = ((np.arange(100) + 0.5) * 1.2 - 0.5) * 2
    assert np.abs(out[10].astype(np.float64) - np.clip(exp, 0, 238)).max() <= 1.0
    # the two vertical products are floored separately (A.3), so rows may differ by one grey level
    assert np.abs(out.astype(np.int32) - out[0].astype(np.int32)).max() <= 1


def test_gaussian_constant_and_impulse():
    img = np.full((20, 30), 200, np.uint8)
    assert (O.gaussian7(img) == 200).all()
    imp = np.zeros((21, 21), np.uint8); imp[10, 10] = 255
    out = O.gaussian7(imp)
    k = np.array([18, 34, 48, 56, 48, 34, 18])
    exp = (np.outer(k, k) * 255 + 32768) >> 16
    assert np.array_equal(out[7:14, 7:14], exp) and out.sum() == exp.sum()
    # reflect-101 at the border: column 0 sees x=-1 -> 1
    edge = np.zeros((9, 9), np.uint8); edge[4, 1] = 255
    o = O.gaussian7(edge)
    assert o[4, 0] == ((k[3] * ((k[2] + k[4]) * 255)) + 32768) >> 16


# ---------------- octree ----------------
def test_octree_basic_properties():
    rng = np.random.default_rng(5)
    cells = rng.choice(300 * 100, 3000, replace=False)
    xs, ys = (cells % 300).astype(np.int32), (cells // 300).astype(np.int32)
    sc = rng.integers(7, 200, 3000).astype(np.int32)
    for n in (1, 10, 100, 434, 5000):
        idx = O.distribute_octtree(xs, ys, sc, 16, 316, 16, 116, n)
        assert len(set(idx.tolist())) == len(idx)
        assert len(idx) <= max(n + 2, 12) or n >= 3000
        if n >= 3000:
            assert len(idx) == 3000  # every point isolated
    assert len(O.distribute_octtree(xs[:0], ys[:0], sc[:0], 16, 316, 16, 116, 10)) == 0
    one = O.distribute_octtree(xs[:1], ys[:1], sc[:1], 16, 316, 16, 116, 10)
    assert one.tolist() == [0]


def test_octree_keeps_max_response_first_wins():
    xs = np.array([10, 11, 12, 200], np.int32); ys = np.array([10, 10, 10, 50], np.int32)
    sc = np.array([30, 50, 50, 9], np.int32)
    idx = O.distribute_octtree(xs, ys, sc, 0, 300, 0, 100, 2)
    assert sorted(idx.tolist()) == [1, 3]  # ties: first (index 1) wins over index 2


# ---------------- end to end on synthetic images ----------------
@pytest.fixture(scope="module")
def small_pair():
    from orbslam2_amd import synth
    return synth.stereo_pair(320, 240, seed=1234)


def test_extract_properties(small_pair):
    left, _ = small_pair
    ex = O.Extractor(nfeatures=500)
    k1, d1 = ex.extract(left)
    k2, d2 = O.Extractor(nfeatures=500).extract(left.copy())
    assert np.array_equal(k1, k2) and np.array_equal(d1, d2)  # deterministic
    assert 400 <= len(k1) <= 500 + 8 * 2
    assert (np.diff(k1["octave"]) >= 0).all()  # level-major order
    sf = ex.scale_factors()
    for l in range(8):
        m = k1["octave"] == l
        w, h = ex.level_size(320, 240, l)
        x, y = k1["x"][m] / sf[l], k1["y"][m] / sf[l]
        assert ((x > 18.99) & (x < w - 19 + 0.01) & (y > 18.99) & (y < h - 19 + 0.01)).all()
        assert (k1["size"][m] == int(np.float32(31) * sf[l])).all()
        assert m.sum() <= ex.features_per_level()[l] + 2
    assert ((k1["angle"] >= 0) & (k1["angle"] <= 360)).all() and (k1["response"] >= 7).all() and (k1["class_id"] == -1).all()
    assert ex.extract(np.zeros((240, 320), np.uint8))[0].shape == (0,)  # flat image: no corners


def test_stereo_on_synthetic_layers(small_pair):
    from orbslam2_amd import synth
    left, right = small_pair
    exl, exr = O.Extractor(nfeatures=500), O.Extractor(nfeatures=500)
    kl, dl = exl.extract(left); kr, dr = exr.extract(right)
    ur, dp, m = O.stereo_matches(exl, exr, kl, dl, kr, dr, 120.0, 300.0)
    ok = ur >= 0
    assert m == ok.sum() and m > 60
    disp = kl["x"][ok] - ur[ok]
    assert (disp >= 0).all() and (disp < 300.0).all()
    assert np.allclose(dp[ok], np.float32(120.0) / np.where(disp <= 0, np.float32(0.01), disp.astype(np.float32)), rtol=1e-6)
    near = np.min(np.abs(disp[:, None] - np.array(synth.LAYER_DISPARITY)[None, :]), axis=1)
    assert (near < 1.0).mean() > 0.8  # most disparities sit on the rendered layers
    # no match possible against an unrelated image
    ur2, _, m2 = O.stereo_matches(exl, exr, kl, dl, kr[:0], dr[:0], 120.0, 300.0)
    assert m2 == 0 and (ur2 == -1).all()


def test_cvt_gray_known_answers():
    """cv::cvtColor 8-bit RGB->GRAY: (R*9798 + G*19235 + B*3735 + 2^14) >> 15 (OpenCV 4.x), weights sum to 2^15."""
    a = np.zeros((1, 6, 3), np.uint8)
    a[0, 0] = 255; a[0, 1] = (255, 0, 0); a[0, 2] = (0, 255, 0); a[0, 3] = (0, 0, 255); a[0, 4] = (10, 200, 97); a[0, 5] = (1, 1, 1)
    g = O.cvt_gray(a, rgb=True)
    assert g.tolist() == [[255, 76, 150, 29, (10 * 9798 + 200 * 19235 + 97 * 3735 + 16384) >> 15, 1]]
    assert O.cvt_gray(a, rgb=False).tolist()[0][1:4] == [29, 150, 76]          # BGR: channel 0 is blue
    assert 9798 + 19235 + 3735 == 1 << 15 and 4899 + 9617 + 1868 == 1 << 14
    grey = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)  # R = G = B = v  ->  v exactly
    assert np.array_equal(O.cvt_gray(grey)[0], np.arange(256)) and np.array_equal(O.cvt_gray(grey, legacy14=True)[0], np.arange(256))
    rgba = np.concatenate([a, np.full((1, 6, 1), 77, np.uint8)], axis=2)        # alpha ignored
    assert np.array_equal(O.cvt_gray(rgba), g)
    rng = np.random.default_rng(0)
    r = rng.integers(0, 256, (7, 13, 3), dtype=np.uint8)
    f = 0.299 * r[..., 0] + 0.587 * r[..., 1] + 0.114 * r[..., 2]
    assert np.abs(O.cvt_gray(r).astype(float) - f).max() <= 0.51 + 1e-9         # within half a grey level + weight rounding


def test_undistort_points_known_answers():
    """cv::undistortPoints(pts, K, D, Mat(), K): inverting the Brown-Conrady forward model (TUM1 coefficients,
    Config/RGB-D-TUM1.yaml) recovers the ideal pixel; zero coefficients are the identity up to the float round trip."""
    fx, fy, cx, cy = 517.3, 516.5, 318.6, 255.3
    D = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
    xs = np.linspace(-0.5, 0.5, 9)
    X, Y = np.meshgrid(xs, xs * 0.75)
    x, y = X.ravel(), Y.ravel()
    r2 = x * x + y * y
    k1, k2, p1, p2, k3 = D
    cd = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    xd = x * cd + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * cd + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    pix = np.stack([fx * xd + cx, fy * yd + cy], 1).astype(np.float32)
    ideal = np.stack([fx * x + cx, fy * y + cy], 1)
    und = O.undistort_points(pix, fx, fy, cx, cy, D)
    assert np.abs(pix - ideal).max() > 5 and np.abs(und - ideal).max() < 2e-3   # five iterations: converged to ~1e-4 px here
    same = O.undistort_points(pix, fx, fy, cx, cy, [0, 0, 0, 0])
    assert np.abs(same - pix).max() < 1e-4
    b = O.image_bounds(640, 480, fx, fy, cx, cy, D)
    assert b[0] > 0 and b[1] < 640 and b[2] > 0 and b[3] < 480                  # pincushion-free TUM1: the corners move inwards
    assert O.image_bounds(640, 480, fx, fy, cx, cy, [0, 0, 0, 0]).tolist() == [0, 640, 0, 480]
    assert O.image_bounds(640, 480, fx, fy, cx, cy, [0, 0.5, 0, 0]).tolist() == [0, 640, 0, 480]  # k1 == 0: the reference skips undistortion


def test_remap_bilinear_known_answers():
    """cv::remap INTER_LINEAR 8UC1, zero constant border: identity, integer shift, half-pixel average, 1/32 quantisation."""
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (20, 30), dtype=np.uint8)
    X, Y = np.meshgrid(np.arange(30, dtype=np.float32), np.arange(20, dtype=np.float32))
    assert np.array_equal(O.remap_bilinear(img, X, Y), img)
    s = O.remap_bilinear(img, X + 3, Y - 2)
    assert np.array_equal(s[2:, :27], img[:-2, 3:]) and (s[:2] == 0).all() and (s[:, 27:] == 0).all()
    h = O.remap_bilinear(img, X + 0.5, Y)
    assert np.array_equal(h[:, :-1], (img[:, :-1].astype(int) + img[:, 1:] + 1) >> 1)
    assert np.array_equal(h[:, -1], (img[:, -1].astype(int) + 1) >> 1)             # right tap outside: reads 0
    q = O.remap_bilinear(img, X + 0.01, Y)                                           # 0.01 * 32 rounds to 0: no interpolation
    assert np.array_equal(q, img)
    f = O.remap_bilinear(img, X + 0.25, Y + 0.75)[:-1, :-1].astype(float)
    a = img.astype(float)
    ref = 0.75 * 0.25 * a[:-1, :-1] + 0.25 * 0.25 * a[:-1, 1:] + 0.75 * 0.75 * a[1:, :-1] + 0.25 * 0.75 * a[1:, 1:]
    assert np.abs(f - ref).max() <= 0.5 + 1e-9
    big = O.remap_bilinear(img, X + 1e6, Y)                                          # coordinates saturate: all outside
    assert (big == 0).all()


def test_blur_resize_and_angle_against_float_reimplementations(small_pair):
    """Independent float re-implementations (scipy / numpy, written from the definitions, not from the oracle's code) bound
    the three fixed-point stages on a real image: Gaussian 7x7 sigma 2 with mirror borders, bilinear resize with OpenCV's
    pixel-centre convention, and the intensity-centroid angle over the circular 31-px patch."""
    from scipy import ndimage
    left, _ = small_pair
    # --- GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101): exact normalised taps, 'mirror' = reflect-101
    x = np.arange(-3, 4, dtype=np.float64)
    g = np.exp(-x * x / 8.0); g /= g.sum()
    ref = ndimage.correlate1d(ndimage.correlate1d(left.astype(np.float64), g, axis=1, mode="mirror"), g, axis=0, mode="mirror")
    got = O.gaussian7(left).astype(np.float64)
    assert np.abs(got - ref).max() <= 1.5 and np.abs(got - ref).mean() < 0.45   # 8.8 fixed-point taps + two roundings
    # --- resize INTER_LINEAR: source coordinate (d + 0.5) * scale - 0.5, clamped at the edges
    h, w = left.shape
    dw, dh = int(round(w / 1.2)), int(round(h / 1.2))
    sx = np.clip((np.arange(dw) + 0.5) * (w / dw) - 0.5, 0, w - 1); sy = np.clip((np.arange(dh) + 0.5) * (h / dh) - 0.5, 0, h - 1)
    x0 = np.floor(sx).astype(int); y0 = np.floor(sy).astype(int)
    x1 = np.minimum(x0 + 1, w - 1); y1 = np.minimum(y0 + 1, h - 1)
    fx = (sx - x0)[None, :]; fy = (sy - y0)[:, None]
    a = left.astype(np.float64)
    ref = (a[y0][:, x0] * (1 - fx) + a[y0][:, x1] * fx) * (1 - fy) + (a[y1][:, x0] * (1 - fx) + a[y1][:, x1] * fx) * fy
    got = O.resize_linear(left, dw, dh).astype(np.float64)
    assert np.abs(got - ref).max() <= 1.01 and np.abs(got - ref).mean() < 0.5   # 11-bit weights, floor / round of resize.cpp
    # --- IC_Angle: atan2(m01, m10) over |v| <= 15, |u| <= umax[|v|], umax from the circle of radius 15 (rounded as the reference)
    ex = O.Extractor(nfeatures=300)
    k, _ = ex.extract(left)
    um = ex.umax()
    assert um[0] == 15 and um[15] == 3 and sum(2 * int(u) + 1 for u in um[1:16]) * 2 + 31 == 749
    worst = 0.0
    for kp in k[k["octave"] == 0][:60]:
        cx, cy = int(round(float(kp["x"]))), int(round(float(kp["y"])))
        m10 = m01 = 0.0
        for v in range(-15, 16):
            d = int(um[abs(v)])
            row = left[cy + v, cx - d:cx + d + 1].astype(np.float64)
            m10 += (np.arange(-d, d + 1) * row).sum(); m01 += v * row.sum()
        ang = np.degrees(np.arctan2(m01, m10)) % 360.0
        diff = abs(ang - float(kp["angle"])); diff = min(diff, 360 - diff)
        worst = max(worst, diff)
    assert worst < 0.3                                                             # cv::fastAtan2's documented accuracy


def test_descriptor_against_numpy_reimplementation(small_pair):
    """computeOrbDescriptor (src/ORBextractor.cc:103-142) written again in numpy float32 from the reference text: rotate the
    256 point pairs of bit_pattern_31_ by the keypoint angle (x*b + y*a, x*a - y*b, each product and sum rounded to float,
    cvRound half-even), sample the blurred level, bit = t0 < t1, LSB first.  Octave-0 keypoints (integer coordinates)."""
    left, _ = small_pair
    ex = O.Extractor(nfeatures=400)
    k, d = ex.extract(left)
    blur = O.gaussian7(left).astype(np.int32)
    p = L.orc_bit_pattern()
    pat = np.array([p[i] for i in range(1024)], np.float32).reshape(256, 4)     # x0 y0 x1 y1
    factor = np.float32(np.pi / np.float64(np.float32(180.0)))                   # (float)(CV_PI / 180.f)
    sel = np.nonzero(k["octave"] == 0)[0][:120]
    assert len(sel) >= 60
    for i in sel:
        cx, cy = int(k["x"][i]), int(k["y"][i])
        rad = np.float32(k["angle"][i] * factor)
        a = np.float32(np.cos(np.float64(rad))); b = np.float32(np.sin(np.float64(rad)))   # Q4: correctly rounded
        def samp(x, y):
            r = np.rint(np.float32(x * b) + np.float32(y * a)).astype(np.int32)
            c = np.rint(np.float32(x * a) - np.float32(y * b)).astype(np.int32)
            return blur[cy + r, cx + c]
        bits = (samp(pat[:, 0], pat[:, 1]) < samp(pat[:, 2], pat[:, 3])).astype(np.uint8)
        ref = np.packbits(bits.reshape(32, 8)[:, ::-1], axis=1).ravel()                      # bit k of byte j = test 8j + k
        assert np.array_equal(ref, d[i]), i


def test_descriptor_follows_the_extractors_own_pattern_copy():
    """ORBextractor keeps its own copy of the 512 test points (src/ORBextractor.cc:442-444) and computeOrbDescriptor reads bit i
    from points 2 i, 2 i + 1 (:116-136): permuting the 256 tests permutes the descriptor's bits and changes nothing else."""
    from orbslam2_amd import dist as D, synth
    img = synth.stereo_pair(240, 180, seed=5)[0]
    pat = D.compiled_pattern()
    perm = np.random.default_rng(2).permutation(256)
    ex0, ex1 = O.Extractor(nfeatures=300), O.Extractor(nfeatures=300)
    assert np.array_equal(ex0.pattern(), pat)
    ex1.set_pattern(pat[perm])
    (k0, d0), (k1, d1) = ex0.extract(img), ex1.extract(img)
    assert len(k0) > 100 and np.array_equal(k0, k1)
    b0 = np.unpackbits(d0, axis=1, bitorder="little")
    b1 = np.unpackbits(d1, axis=1, bitorder="little")
    assert np.array_equal(b1, b0[:, perm])
