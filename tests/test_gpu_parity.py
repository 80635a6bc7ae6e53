"""GPU parity: HIP path (through the C ABI) == CPU oracle, bit for bit.

Integer / byte / index work is compared exactly; the sub-pixel stereo outputs
are floats and BASELINE.json's north_star allows 1e-4 -- the kernels follow the
oracle's rounding contract (Q4) so they are compared exactly as well, with the
1e-4 tolerance as the documented fallback bound.
"""
import numpy as np
import pytest

from oracle import oracle as O
from orbslam2_amd import synth

pytestmark = pytest.mark.gpu

KITTI = dict(width=1241, height=376, nfeatures=2000, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448)
SMALL = dict(width=320, height=240, nfeatures=500, fx=300.0, fy=300.0, cx=160.0, cy=120.0, bf=120.0)
TUM1 = dict(width=640, height=480, nfeatures=1000, fx=517.3, fy=516.5, cx=318.6, cy=255.3, bf=40.0)
EUROC = dict(width=752, height=480, nfeatures=1200, fx=458.654, fy=457.296, cx=367.215, cy=248.375, bf=47.9)
D435I = dict(width=1280, height=720, nfeatures=2500, fx=911.0, fy=911.0, cx=640.0, cy=360.0, bf=45.5)
CONFIGS = {"small": SMALL, "kitti": KITTI, "tum1": TUM1, "euroc": EUROC, "d435i": D435I}


def _ctx(cfg, max_images=2):
    from orbslam2_amd import api
    return api.Context(max_images=max_images, **cfg)


@pytest.fixture(scope="module", params=["small", "kitti", "tum1", "euroc", "d435i"])
def case(request):
    cfg = CONFIGS[request.param]
    left, right = synth.stereo_pair(cfg["width"], cfg["height"], seed=1234)
    ctx = _ctx(cfg)
    exl = O.Extractor(nfeatures=cfg["nfeatures"])
    exr = O.Extractor(nfeatures=cfg["nfeatures"])
    kl, dl = exl.extract(left)
    kr, dr = exr.extract(right)
    out = ctx.stereo_frame(left, right)
    yield dict(cfg=cfg, left=left, right=right, ctx=ctx, exl=exl, exr=exr, kl=kl, dl=dl, kr=kr, dr=dr, out=out)
    ctx.close()


def test_tables(case):
    t = case["ctx"].tables()
    ex = case["exl"]
    assert np.array_equal(t["scale"], ex.scale_factors())
    assert np.array_equal(t["inv_scale"], ex.inv_scale_factors())
    assert np.array_equal(t["sigma2"], ex.sigma2())
    assert np.array_equal(t["inv_sigma2"], ex.inv_sigma2())
    assert np.array_equal(t["features_per_level"], ex.features_per_level())
    assert np.array_equal(t["umax"], ex.umax())
    cfg = case["cfg"]
    for l in range(8):
        assert case["ctx"].level_size(l) == ex.level_size(cfg["width"], cfg["height"], l)


def test_pyramid_bit_exact(case):
    for img, ex in ((0, case["exl"]), (1, case["exr"])):
        for l in range(8):
            got = case["ctx"].fetch_pyramid(img, l)
            ref = ex.pyramid_level(l)
            assert got.shape == ref.shape
            bad = np.argwhere(got != ref)
            assert bad.size == 0, "image %d level %d: %d pixels differ, first %s" % (img, l, len(bad), bad[:3].tolist())


def test_blur_bit_exact(case):
    ex = case["exl"]
    for l in range(8):
        got = case["ctx"].fetch_pyramid(0, l, blurred=True)
        ref = O.gaussian7(ex.pyramid_level(l))
        bad = np.argwhere(got != ref)
        assert bad.size == 0, "level %d: %d pixels differ, first %s" % (l, len(bad), bad[:3].tolist())


def test_fast_candidates_exact_order(case):
    for img, ex in ((0, case["exl"]), (1, case["exr"])):
        for l in range(8):
            gx, gy, gs = case["ctx"].fetch_candidates(img, l)
            rx, ry, rs = ex.level_candidates(l)
            assert len(gx) == len(rx), "image %d level %d: %d vs %d candidates" % (img, l, len(gx), len(rx))
            assert np.array_equal(gx, rx) and np.array_equal(gy, ry) and np.array_equal(gs, rs), "image %d level %d" % (img, l)


def _assert_kps_equal(got, ref, what):
    assert len(got) == len(ref), "%s: %d vs %d keypoints" % (what, len(got), len(ref))
    for f in ("octave", "x", "y", "response", "size", "angle", "class_id"):
        bad = np.nonzero(got[f] != ref[f])[0]
        assert bad.size == 0, "%s: field %s differs at %s (got %s ref %s)" % (
            what, f, bad[:5].tolist(), got[f][bad[:5]].tolist(), ref[f][bad[:5]].tolist())


def test_keypoints_bit_exact(case):
    _assert_kps_equal(case["out"]["kps_left"], case["kl"], "left")
    _assert_kps_equal(case["out"]["kps_right"], case["kr"], "right")


def test_descriptors_bit_exact(case):
    for got, ref, what in ((case["out"]["desc_left"], case["dl"], "left"), (case["out"]["desc_right"], case["dr"], "right")):
        assert got.shape == ref.shape
        bad = np.nonzero((got != ref).any(axis=1))[0]
        assert bad.size == 0, "%s: %d descriptors differ, first rows %s" % (what, len(bad), bad[:5].tolist())


def test_stereo_matches(case):
    cfg = case["cfg"]
    ur, dp, m = O.stereo_matches(case["exl"], case["exr"], case["kl"], case["dl"], case["kr"], case["dr"], cfg["bf"], cfg["fx"])
    got_u, got_d = case["out"]["u_right"], case["out"]["depth"]
    assert len(got_u) == len(ur)
    assert np.array_equal(got_u < 0, ur < 0), "matched sets differ at %s" % np.nonzero((got_u < 0) != (ur < 0))[0][:8].tolist()
    assert m == int((ur >= 0).sum()) and m > len(ur) // 10
    # north_star tolerance for the sub-pixel disparity: 1e-4; the kernels are expected to be exact
    assert np.allclose(got_u, ur, rtol=0, atol=1e-4)
    assert np.array_equal(got_u, ur), "u_right not bit-exact: max diff %g" % np.abs(got_u - ur).max()
    assert np.array_equal(got_d, dp), "depth not bit-exact: max rel diff %g" % np.abs((got_d - dp) / np.maximum(dp, 1e-9)).max()


def test_extract_entry_point_matches_stereo_slot(case):
    k, d = case["ctx"].extract(case["left"])
    _assert_kps_equal(k, case["kl"], "extract()")
    assert np.array_equal(d, case["dl"])


def test_hamming_matrix(case):
    a, b = case["dl"][:100], case["dr"][:77]
    got = case["ctx"].hamming_matrix(a, b)
    ref = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(axis=2)
    assert np.array_equal(got, ref)
    assert got[3, 5] == O.hamming256(a[3], b[5])


def test_rgbd_frame_matches_oracle():
    """Frame::Frame(rgbd) body: extract + ComputeStereoFromRGBD (src/Frame.cc:645-666), BASELINE config 5 geometry."""
    cfg = D435I
    left, _, depth = synth.stereo_pair(cfg["width"], cfg["height"], seed=77, with_depth=True, bf=cfg["bf"])
    ctx = _ctx(cfg, max_images=1)
    out = ctx.rgbd_frame(left, depth)
    ex = O.Extractor(nfeatures=cfg["nfeatures"])
    k, d = ex.extract(left)
    ur, dp = O.stereo_from_rgbd(k, k, depth, cfg["bf"])
    _assert_kps_equal(out["kps"], k, "rgbd")
    assert np.array_equal(out["desc"], d)
    assert np.array_equal(out["u_right"], ur) and np.array_equal(out["depth"], dp)
    assert (dp > 0).mean() > 0.8 and (dp == -1).sum() > 0  # 5 % holes in the synthetic depth map
    ctx.close()


def test_empty_and_flat_images():
    from orbslam2_amd import api
    ctx = _ctx(SMALL)
    k, d = ctx.extract(np.zeros((SMALL["height"], SMALL["width"]), np.uint8))  # no corners at all
    assert len(k) == 0 and d.shape == (0, 32)
    out = ctx.stereo_frame(np.full((SMALL["height"], SMALL["width"]), 200, np.uint8), np.zeros((SMALL["height"], SMALL["width"]), np.uint8))
    assert len(out["kps_left"]) == 0 and len(out["u_right"]) == 0
    with pytest.raises(api.OrbfeError):
        ctx.extract(np.zeros((100, 100), np.uint8))  # size other than the context's
    ctx.close()


def test_row_strided_inputs_match_contiguous(case):
    """The reference hands cv::Mat headers that may be ROIs of a wider buffer (step > cols); the ABI takes the row
    stride.  Results must not depend on it."""
    L, R = case["left"], case["right"]
    h, w = L.shape
    big = np.full((2, h, w + 37), 255, np.uint8)
    big[0, :, 5:5 + w] = L
    big[1, :, 5:5 + w] = R
    lv, rv = big[0, :, 5:5 + w], big[1, :, 5:5 + w]
    assert lv.strides[0] == w + 37 and not lv.flags["C_CONTIGUOUS"]
    out = case["ctx"].stereo_frame(lv, rv)
    ref = case["out"]
    for k in ("kps_left", "desc_left", "kps_right", "desc_right", "u_right", "depth"):
        assert np.array_equal(out[k], ref[k]), k
    k1, d1 = case["ctx"].extract(lv)
    assert np.array_equal(k1, ref["kps_left"]) and np.array_equal(d1, ref["desc_left"])


def test_rgbd_row_strided_depth():
    cfg = SMALL
    left, _, depth = synth.stereo_pair(cfg["width"], cfg["height"], seed=5, with_depth=True, bf=cfg["bf"])
    ctx = _ctx(cfg, max_images=1)
    ref = ctx.rgbd_frame(left, depth)
    h, w = depth.shape
    bigd = np.zeros((h, w + 11), np.float32)
    bigd[:, 3:3 + w] = depth
    bigg = np.zeros((h, w + 64), np.uint8)
    bigg[:, 64:] = left
    out = ctx.rgbd_frame(bigg[:, 64:], bigd[:, 3:3 + w])
    for k in ref:
        assert np.array_equal(out[k], ref[k]), k
    assert len(ref["kps"]) > 50 and (ref["depth"] > 0).any()
    ctx.close()


def test_rgbd_raw_u16_depth_matches_converted_map():
    """Tracking::GrabImageRGBD (src/Tracking.cc:323-324) converts the sensor's CV_16U map with
    convertTo(CV_32F, mDepthMapFactor) -- one rounded float multiply per pixel -- before Frame::Frame; the u16 entry
    point folds that into the sampling.  TUM factor 5000 (Examples/RGB-D/TUM1.yaml)."""
    cfg = TUM1
    left, _, depth = synth.stereo_pair(cfg["width"], cfg["height"], seed=21, with_depth=True, bf=cfg["bf"])
    raw = np.clip(np.rint(depth * 5000.0), 0, 65535).astype(np.uint16)
    factor = np.float32(1.0) / np.float32(5000.0)
    conv = raw.astype(np.float32) * factor  # the oracle of convertTo for this type pair
    ctx = _ctx(cfg, max_images=1)
    got = ctx.rgbd_frame(left, raw, depth_map_factor=float(factor))
    ex = O.Extractor(nfeatures=cfg["nfeatures"])
    k, d = ex.extract(left)
    ur, dp = O.stereo_from_rgbd(k, k, conv, cfg["bf"])
    _assert_kps_equal(got["kps"], k, "rgbd u16")
    assert np.array_equal(got["desc"], d)
    assert np.array_equal(got["u_right"], ur) and np.array_equal(got["depth"], dp)
    ref = ctx.rgbd_frame(left, conv)
    for key in ref:
        assert np.array_equal(got[key], ref[key]), key
    wide = np.zeros((raw.shape[0], raw.shape[1] + 6), np.uint16)
    wide[:, 2:2 + raw.shape[1]] = raw
    got2 = ctx.rgbd_frame(left, wide[:, 2:2 + raw.shape[1]], depth_map_factor=float(factor))
    for key in ref:
        assert np.array_equal(got2[key], ref[key]), key
    assert (dp > 0).sum() > 100
    ctx.close()


@pytest.mark.parametrize("cn,rgb,legacy", [(3, True, False), (3, False, False), (4, True, False), (4, False, True)])
def test_colour_input_is_converted_like_cvtcolor(cn, rgb, legacy):
    """Tracking::GrabImageStereo converts 3- / 4-channel input with cv::cvtColor(COLOR_{RGB,BGR}[A]2GRAY) before the Frame is
    built (src/Tracking.cc:269-294); orbfe_set_input_format folds that into ingest.  The result must equal the grey path fed
    with the oracle's conversion, on packed and on row-strided colour buffers, and switching back to grey must work."""
    cfg = SMALL
    w, h = cfg["width"], cfg["height"]
    planes = [synth.stereo_pair(w, h, seed=300 + k) for k in range(3)]
    rng = np.random.default_rng(9)
    def colour(side):
        c = np.stack([planes[k][side] for k in range(3)], axis=2)
        if cn == 4:
            c = np.concatenate([c, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], axis=2)  # alpha: ignored
        return np.ascontiguousarray(c)
    cl, cr = colour(0), colour(1)
    gl, gr = O.cvt_gray(cl, rgb, legacy), O.cvt_gray(cr, rgb, legacy)
    assert gl.std() > 10 and not np.array_equal(gl, O.cvt_gray(cl, not rgb, legacy))
    ctx = _ctx(cfg)
    ref = ctx.stereo_frame(gl, gr)
    assert len(ref["kps_left"]) > 100
    ctx.set_input_format(cn, rgb, legacy)
    out = ctx.stereo_frame(cl, cr)
    for k in ref:
        assert np.array_equal(out[k], ref[k]), k
    assert np.array_equal(ctx.fetch_pyramid(0, 0), gl) and np.array_equal(ctx.fetch_pyramid(1, 0), gr)
    wide = np.zeros((2, h, w + 9, cn), np.uint8)
    wide[0, :, 4:4 + w], wide[1, :, 4:4 + w] = cl, cr
    out2 = ctx.stereo_frame(wide[0, :, 4:4 + w], wide[1, :, 4:4 + w])
    for k in ref:
        assert np.array_equal(out2[k], ref[k]), k
    k1, d1 = ctx.extract(cr)
    assert np.array_equal(k1, ref["kps_right"]) and np.array_equal(d1, ref["desc_right"])
    ctx.set_input_format(1)
    out3 = ctx.stereo_frame(gl, gr)
    for k in ref:
        assert np.array_equal(out3[k], ref[k]), k
    ctx.close()


TUM1_DIST = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]     # Config/RGB-D-TUM1.yaml
D435I_DIST = [9.8300891329190163e-02, 2.9249092034088525e-01, 0.0, 0.0, -1.6417657631343097e+00]  # Config/RealSense-D435i-RGBD.yaml


@pytest.mark.parametrize("name,dist", [("tum1", TUM1_DIST), ("d435i", D435I_DIST), ("tum1-4", TUM1_DIST[:4])])
def test_undistortion_matches_oracle(name, dist):
    """Frame::UndistortKeyPoints / ComputeImageBounds (src/Frame.cc:402-462): cv::undistortPoints in double on the device,
    bit for bit; RGB-D uRight from the undistorted x (:652-664)."""
    cfg = TUM1 if name.startswith("tum1") else D435I
    ctx = _ctx(cfg, max_images=1)
    ctx.set_distortion(dist)
    rng = np.random.default_rng(4)
    k = np.zeros(5000, O.KP_DTYPE)
    k["x"] = rng.uniform(-5, cfg["width"] + 5, 5000); k["y"] = rng.uniform(-5, cfg["height"] + 5, 5000)
    k["x"][:4] = [0, cfg["width"], 0, cfg["width"]]; k["y"][:4] = [0, 0, cfg["height"], cfg["height"]]
    k["octave"] = rng.integers(0, 8, 5000); k["angle"] = rng.uniform(0, 360, 5000); k["size"] = 31; k["response"] = 40; k["class_id"] = -1
    ref = O.undistort_points(np.stack([k["x"], k["y"]], 1), cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], dist)
    got = ctx.undistort_keypoints(k)
    assert np.array_equal(got["x"], ref[:, 0]) and np.array_equal(got["y"], ref[:, 1])
    for f in ("size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(got[f], k[f])
    assert np.abs(got["x"] - k["x"]).max() > 1.0
    assert np.array_equal(ctx.image_bounds(), O.image_bounds(cfg["width"], cfg["height"], cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], dist))
    # RGB-D frame: depth sampled at the distorted keypoint, uRight from the undistorted x
    left, _, depth = synth.stereo_pair(cfg["width"], cfg["height"], seed=31, with_depth=True, bf=cfg["bf"])
    out = ctx.rgbd_frame(left, depth)
    ex = O.Extractor(nfeatures=cfg["nfeatures"])
    kk, dd = ex.extract(left)
    und = O.undistort_points(np.stack([kk["x"], kk["y"]], 1), cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], dist)
    kun = kk.copy(); kun["x"], kun["y"] = und[:, 0], und[:, 1]
    ur, dp = O.stereo_from_rgbd(kk, kun, depth, cfg["bf"])
    _assert_kps_equal(out["kps"], kk, "rgbd distorted")
    assert np.array_equal(out["u_right"], ur) and np.array_equal(out["depth"], dp)
    ur0, _ = O.stereo_from_rgbd(kk, kk, depth, cfg["bf"])
    assert not np.array_equal(ur, ur0)
    got_un = ctx.fetch_keys_un(0)
    assert np.array_equal(got_un, kun)
    # k1 == 0 => identity, bounds = image
    ctx.set_distortion([0.0, 0.3, 0.0, 0.0])
    assert np.array_equal(ctx.undistort_keypoints(k), k) and ctx.image_bounds().tolist() == [0, cfg["width"], 0, cfg["height"]]
    ctx.set_distortion([])
    assert np.array_equal(ctx.rgbd_frame(left, depth)["u_right"], ur0)
    ctx.close()


def _rectify_maps(w, h, sw, sh, seed):
    """Smooth synthetic stand-ins for initUndistortRectifyMap's output: a small rotation + radial term, partly leaving
    the source so the constant border is exercised."""
    X, Y = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    rng = np.random.default_rng(seed)
    th, k = rng.uniform(-0.02, 0.02), rng.uniform(-2e-7, 2e-7)
    xc, yc = X - w / 2, Y - h / 2
    r2 = xc * xc + yc * yc
    mx = (np.cos(th) * xc - np.sin(th) * yc) * (1 + k * r2) + sw / 2 + rng.uniform(-3, 3)
    my = (np.sin(th) * xc + np.cos(th) * yc) * (1 + k * r2) + sh / 2 + rng.uniform(-3, 3)
    return mx.astype(np.float32), my.astype(np.float32)


def test_rectified_ingest_matches_remap_then_extract():
    """EuRoC flow (Test/Replay/Stereo/stereo_euroc.cc:136-137): remap both raw images, then the stereo Frame.  Folded into
    ingest, the result must equal the oracle's remap followed by the plain pipeline, with a raw size that differs from the
    rectified one."""
    cfg = SMALL
    w, h = cfg["width"], cfg["height"]
    sw, sh = w + 16, h + 8
    rawl, rawr = synth.stereo_pair(sw, sh, seed=71)
    mxl, myl = _rectify_maps(w, h, sw, sh, 1)
    mxr, myr = _rectify_maps(w, h, sw, sh, 2)
    rl, rr = O.remap_bilinear(rawl, mxl, myl), O.remap_bilinear(rawr, mxr, myr)
    assert (rl == 0).sum() > 0 and rl.std() > 20
    ctx = _ctx(cfg)
    ref = ctx.stereo_frame(rl, rr)
    ctx.set_rectification(0, mxl, myl, (sw, sh))
    mono = ctx.extract(rawl)                                  # without a right map every slot uses the left one
    assert np.array_equal(mono[0], ref["kps_left"]) and np.array_equal(mono[1], ref["desc_left"])
    ctx.set_rectification(1, mxr, myr, (sw, sh))
    out = ctx.stereo_frame(rawl, rawr)
    assert np.array_equal(ctx.fetch_pyramid(0, 0), rl) and np.array_equal(ctx.fetch_pyramid(1, 0), rr)
    for k in ref:
        assert np.array_equal(out[k], ref[k]), k
    assert len(ref["kps_left"]) > 100
    from orbslam2_amd import api
    with pytest.raises(api.OrbfeError):
        ctx.stereo_frame(rl, rr)                              # rectified-size input is refused while rectification is on
    with pytest.raises(api.OrbfeError):
        ctx.set_input_format(3)
    ctx.set_rectification(0)                                   # off again
    out2 = ctx.stereo_frame(rl, rr)
    for k in ref:
        assert np.array_equal(out2[k], ref[k]), k
    ctx.close()


def test_hamming_matrix_leaves_rectification_and_undistortion_state_alone():
    """Regression (round-1 advisor finding): orbfe_hamming_matrix used to free the rectification maps and the undistortion
    scratch when it grew its own scratch, so the next rectified frame read freed memory.  Sequence from the finding:
    set_rectification + fetch_keys_un, then hamming_matrix (first call always grows), then the same frame again."""
    cfg = SMALL
    w, h = cfg["width"], cfg["height"]
    sw, sh = w + 16, h + 8
    rawl, rawr = synth.stereo_pair(sw, sh, seed=72)
    mxl, myl = _rectify_maps(w, h, sw, sh, 3)
    mxr, myr = _rectify_maps(w, h, sw, sh, 4)
    ctx = _ctx(cfg)
    ctx.set_distortion([-0.28, 0.07, 1e-4, 2e-5, 0.0])
    ctx.set_rectification(0, mxl, myl, (sw, sh))
    ctx.set_rectification(1, mxr, myr, (sw, sh))
    ref = ctx.stereo_frame(rawl, rawr)
    un0, b0 = ctx.fetch_keys_un(0), ctx.image_bounds()
    rng = np.random.default_rng(5)
    for na, nb in ((40, 70), (300, 500), (16, 16)):  # grows twice, then reuses
        a = rng.integers(0, 256, (na, 32), dtype=np.uint8); b = rng.integers(0, 256, (nb, 32), dtype=np.uint8)
        d = ctx.hamming_matrix(a, b)
        assert np.array_equal(d, np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(axis=2))
        out = ctx.stereo_frame(rawl, rawr)
        for k in ref:
            assert np.array_equal(out[k], ref[k]), k
        assert np.array_equal(ctx.fetch_keys_un(0), un0) and np.array_equal(ctx.image_bounds(), b0)
    ctx.set_rectification(0)  # clearing and destroying must not double-free
    ctx.close()


def test_new_entry_points_reject_bad_arguments():
    """Error behaviour of the round's new entry points: invalid arguments come back as error codes with a message, never as
    a crash or a silent default."""
    from orbslam2_amd import api
    import ctypes as C
    ctx = _ctx(SMALL, max_images=2)
    L = ctx.L
    assert L.orbfe_set_distortion(ctx.h, None, 3) == api.ERR_INVALID
    assert L.orbfe_set_distortion(ctx.h, None, 0) == api.OK
    assert L.orbfe_set_input_format(ctx.h, 2, 1, 0) == api.ERR_INVALID
    mx = np.zeros((SMALL["height"], SMALL["width"]), np.float32)
    assert L.orbfe_set_rectification(ctx.h, 1, mx.ctypes.data_as(C.c_void_p), mx.ctypes.data_as(C.c_void_p), 100, 100) == api.ERR_INVALID  # right before left
    assert L.orbfe_set_rectification(ctx.h, 2, None, None, 0, 0) == api.ERR_INVALID
    assert L.orbfe_set_rectification(ctx.h, 0, mx.ctypes.data_as(C.c_void_p), mx.ctypes.data_as(C.c_void_p), 0, 10) == api.ERR_INVALID
    assert L.orbfe_fetch_batch_async(ctx.h, 3, None, None, None, None, None, None) == api.ERR_INVALID                                 # more than max_images
    assert L.orbfe_fetch_batch_async(ctx.h, 2, None, None, None, None, None, None) == api.OK                                          # all outputs optional
    assert b"" != L.orbfe_last_error(ctx.h)
    n = C.c_int()
    assert L.orbfe_pose_optimization(ctx.h, None, 5, None, None, None, None, None, C.byref(n)) == api.ERR_INVALID
    off = np.array([0, 5, 3], np.int32)  # decreasing offsets
    T = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1))
    k = np.zeros(5, api.KP_DTYPE); f = np.zeros(5, np.float32); h = np.zeros(5, np.uint8); X = np.zeros(15, np.float32); ninl = np.zeros(2, np.int32)
    assert L.orbfe_pose_optimization_batch(ctx.h, 2, off.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p),
                                           f.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p), X.ctypes.data_as(C.c_void_p),
                                           h.ctypes.data_as(C.c_void_p), ninl.ctypes.data_as(C.c_void_p)) == api.ERR_INVALID
    # the context still works afterwards
    left, right = synth.stereo_pair(SMALL["width"], SMALL["height"], seed=3)
    assert len(ctx.stereo_frame(left, right)["kps_left"]) > 50
    ctx.close()
