"""The route from "parity unpinned" to "pinned": tools/opencv_pin/ (a kit for the first machine that has OpenCV >= 4.5.1).

tools/opencv_pin/pin.cpp compiles the reference's OWN src/ORBextractor.cc against a real OpenCV and dumps its outputs (and
per-primitive outputs of cv::FAST, cv::resize, cv::GaussianBlur, cv::fastAtan2, cvRound, cv::cvtColor, cv::remap,
cv::undistortPoints) on this repository's test inputs; tools/opencv_pin/import_pins.py turns them into
tests/golden/reference_pinned/*.npz.  While that directory is empty (this image has no OpenCV) the pin tests below SKIP and only
the kit itself is checked: pin.cpp must parse against declaration stand-ins, the stand-in of ORBextractor.h must only name
members the reference's header has, and the container reader must round-trip.  The day the directory is filled, the same tests
hold the CPU oracle (and, under -m gpu, the HIP path) to the reference's own bits."""
import glob
import os
import re
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIT = os.path.join(ROOT, "tools", "opencv_pin")
PIN_DIR = os.path.join(ROOT, "tests", "golden", "reference_pinned")
CASES = sorted(glob.glob(os.path.join(PIN_DIR, "case_*.npz")))
PRIMS = os.path.join(PIN_DIR, "primitives.npz")
REF = "/root/reference"
sys.path.insert(0, KIT)


# ---- the kit itself (always runs) ----
def test_pin_cpp_parses_against_declaration_stand_ins():
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wextra", "-I", os.path.join(KIT, "syntax_stub"), os.path.join(KIT, "pin.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0 and not r.stderr.strip(), r.stderr[-3000:]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "include", "ORBextractor.h")), reason="reference tree not present")
def test_stand_in_names_only_members_the_reference_header_has():
    ref = open(os.path.join(REF, "include", "ORBextractor.h")).read()
    stub = open(os.path.join(KIT, "syntax_stub", "ORBextractor.h")).read()
    names = re.findall(r"\b(Get\w+|mvImagePyramid|operator\(\))", stub.split("class ORBextractor")[1])
    assert len(names) >= 6
    for n in names:
        assert n in ref, n
    # the constructor pin.cpp calls: eight arguments in this order (include/ORBextractor.h:51)
    m = re.search(r"ORBextractor\(([^)]*)\);", ref)
    args = [a.strip().split()[-1] for a in m.group(1).split(",")]
    assert args == ["nfeatures", "scaleFactor", "nlevels", "iniThFAST", "minThFAST", "patchSize", "halfPatchSize", "edgeThreshold"], args
    assert os.path.exists(os.path.join(REF, "src", "ORBextractor.cc"))  # the one translation unit the kit compiles, where it lies


def test_container_reader_round_trips(tmp_path):
    import import_pins as IP
    recs = {"a": np.arange(12, dtype=np.int32).reshape(3, 4), "b": np.linspace(0, 1, 5).astype(np.float32), "c": np.arange(6, dtype=np.uint8).reshape(1, 2, 3),
            "d": np.array([1.5, -2.25], np.float64)}
    code = {np.dtype(np.uint8): 0, np.dtype(np.int32): 1, np.dtype(np.float32): 2, np.dtype(np.float64): 3}
    p = tmp_path / "x.pin"
    with open(p, "wb") as f:
        f.write(b"ORBPIN01")
        for k, v in recs.items():
            f.write(struct.pack("<I", len(k)) + k.encode() + struct.pack("<II", code[v.dtype], v.ndim) + struct.pack("<%dI" % v.ndim, *v.shape) + v.tobytes())
    got = IP.read_pin(str(p))
    assert list(got) == list(recs) and all(np.array_equal(got[k], recs[k]) and got[k].dtype == recs[k].dtype for k in recs)


def test_export_writes_every_case_the_manifest_names(tmp_path):
    import export_inputs as EX
    import import_pins as IP
    EX.main(str(tmp_path))
    rows = [ln.split() for ln in open(tmp_path / "cases.txt").read().splitlines()]
    assert len(rows) >= 10 and {"kitti", "tum1", "euroc", "d435i", "stereo_320x240_f500"} <= {r[0] for r in rows}
    for r in rows:
        assert len(r) == 10
        l = IP.read_pgm(str(tmp_path / r[1])); rr = IP.read_pgm(str(tmp_path / r[2]))
        assert l.shape == rr.shape and l.dtype == np.uint8 and l.std() > 5
    from orbslam2_amd import synth
    k = next(r for r in rows if r[0] == "kitti")
    assert np.array_equal(IP.read_pgm(str(tmp_path / k[1])), synth.stereo_pair(1241, 376, seed=1234)[0])


# ---- the pins (skipped until tests/golden/reference_pinned/ is filled by the kit) ----
needs_pins = pytest.mark.skipif(not CASES, reason="tests/golden/reference_pinned/ is empty: no machine with OpenCV has run tools/opencv_pin yet (parity unpinned)")


def _params(g):
    w, h, nf, fx, bf, scale, levels, ini, mn = g["params"]
    return dict(w=int(w), h=int(h), nf=int(nf), fx=float(fx), bf=float(bf), scale=float(np.float32(scale)), levels=int(levels), ini=int(ini), mn=int(mn))


def _diff_report(name, a, b):
    if a.shape != b.shape:
        return "%s: shape %s vs %s" % (name, a.shape, b.shape)
    bad = np.argwhere(np.asarray(a) != np.asarray(b))
    return "" if bad.size == 0 else "%s: %d of %d differ, first at %s" % (name, len(bad), a.size, bad[0].tolist())


def _check_case(g):
    from oracle import oracle as O
    p = _params(g)
    mk = lambda: O.Extractor(nfeatures=p["nf"], scale_factor=p["scale"], nlevels=p["levels"], ini_th_fast=p["ini"], min_th_fast=p["mn"])
    exl, exr = mk(), mk()
    kl, dl = exl.extract(g["left"]); kr, dr = exr.extract(g["right"])
    ur, dp, _ = O.stereo_matches(exl, exr, kl, dl, kr, dr, p["bf"], p["fx"])
    rep = []
    for l in range(p["levels"]):  # cv::resize chain, then cv::GaussianBlur of every level
        rep.append(_diff_report("pyr%d" % l, exl.pyramid_level(l), g["pyr%d" % l]))
        rep.append(_diff_report("blur%d" % l, O.gaussian7(exl.pyramid_level(l)), g["blur%d" % l]))
    for f, got in (("scale", exl.scale_factors()), ("inv_scale", exl.inv_scale_factors()), ("sigma2", exl.sigma2()), ("inv_sigma2", exl.inv_sigma2())):
        rep.append(_diff_report(f, got.view(np.uint32), g[f].view(np.uint32)))
    for side, k, d, gk, gd in (("left", kl, dl, g["kl"], g["dl"]), ("right", kr, dr, g["kr"], g["dr"])):
        if len(k) != len(gk):
            rep.append("%s: %d keypoints vs %d" % (side, len(k), len(gk)))
            continue
        for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
            rep.append(_diff_report(side + "." + f, k[f].view(np.uint32) if k[f].dtype == np.float32 else k[f], gk[f].view(np.uint32) if gk[f].dtype == np.float32 else gk[f]))
        rep.append(_diff_report(side + ".desc", d, gd))
    if len(kl) == len(g["kl"]):
        rep.append(_diff_report("u_right", ur.view(np.uint32), g["u_right"].view(np.uint32)))
        rep.append(_diff_report("depth", dp.view(np.uint32), g["depth"].view(np.uint32)))
    rep = [r for r in rep if r]
    assert not rep, "oracle != reference (OpenCV %s):\n  " % bytes(g["opencv_version"]).decode() + "\n  ".join(rep)


@needs_pins
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_oracle_equals_the_reference_built_against_opencv(path):
    _check_case(np.load(path))


@needs_pins
@pytest.mark.gpu
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_hip_equals_the_reference_built_against_opencv(path):
    from orbslam2_amd import api
    g = np.load(path)
    p = _params(g)
    ctx = api.Context(width=p["w"], height=p["h"], nfeatures=p["nf"], scale_factor=p["scale"], nlevels=p["levels"], ini_th_fast=p["ini"], min_th_fast=p["mn"],
                      fx=p["fx"], fy=p["fx"], cx=p["w"] / 2, cy=p["h"] / 2, bf=p["bf"])
    out = ctx.stereo_frame(g["left"], g["right"])
    assert out["kps_left"].tobytes() == g["kl"].tobytes() and out["kps_right"].tobytes() == g["kr"].tobytes()
    assert np.array_equal(out["desc_left"], g["dl"]) and np.array_equal(out["desc_right"], g["dr"])
    assert out["u_right"].tobytes() == g["u_right"].tobytes() and out["depth"].tobytes() == g["depth"].tobytes()
    for l in range(p["levels"]):
        assert np.array_equal(ctx.fetch_pyramid(0, l), g["pyr%d" % l]) and np.array_equal(ctx.fetch_pyramid(0, l, blurred=True), g["blur%d" % l])
    ctx.close()


def _check_primitives(g):
    """One block per OPENCV-4.5.5-SEMANTICS tag of oracle/orb_oracle.c."""
    import ctypes as C
    from oracle import oracle as O
    L = O.lib()
    rep = []
    got = np.array([L.orc_cv_round_f(C.c_float(float(v))) for v in g["cvround_f_in"]], np.int32)
    rep.append(_diff_report("cvRound(float)", got, g["cvround_f_out"]))
    got = np.array([L.orc_cv_round_d(C.c_double(float(v))) for v in g["cvround_d_in"]], np.int32)
    rep.append(_diff_report("cvRound(double)", got, g["cvround_d_out"]))
    got = np.array([L.orc_fast_atan2(C.c_float(float(y)), C.c_float(float(x))) for y, x in zip(g["atan2_y"], g["atan2_x"])], np.float32)
    rep.append(_diff_report("fastAtan2", got.view(np.uint32), g["atan2_out"].view(np.uint32)))
    i = 0
    while "resize%d_l0" % i in g:
        cur = g["resize%d_l0" % i]
        for l in range(1, 4):
            ref = g["resize%d_l%d" % (i, l)]
            rep.append(_diff_report("resize%d level %d" % (i, l), O.resize_linear(cur, ref.shape[1], ref.shape[0]), ref))
            cur = ref
        rep.append(_diff_report("GaussianBlur %d" % i, O.gaussian7(cur), g["blur%d" % i]))
        i += 1
    for j in range(5):
        ref = g["resize_any_%d" % j]
        rep.append(_diff_report("resize_any_%d" % j, O.resize_linear(g["resize_any_src"], ref.shape[1], ref.shape[0]), ref))
    i = 0
    while "fast%d_img" % i in g:
        for t in (20, 7):
            xs, ys, ss = O.fast9_16(g["fast%d_img" % i], t, True)
            rep.append(_diff_report("FAST %d t=%d" % (i, t), np.stack([xs, ys, ss], 1).astype(np.int32).reshape(-1, 3), g["fast%d_t%d" % (i, t)].reshape(-1, 3)))
        i += 1
    rep.append(_diff_report("RGB2GRAY", O.cvt_gray(g["cvt_rgb_in"], True), g["cvt_rgb2gray"]))
    rep.append(_diff_report("BGR2GRAY", O.cvt_gray(g["cvt_rgb_in"], False), g["cvt_bgr2gray"]))
    rep.append(_diff_report("RGBA2GRAY", O.cvt_gray(g["cvt_rgba_in"], True), g["cvt_rgba2gray"]))
    rep.append(_diff_report("BGRA2GRAY", O.cvt_gray(g["cvt_rgba_in"], False), g["cvt_bgra2gray"]))
    rep.append(_diff_report("remap", O.remap_bilinear(g["remap_src"], g["remap_mx"], g["remap_my"]), g["remap_out"]))
    d5 = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)
    for nd in (4, 5):
        got = O.undistort_points(g["undist%d_in" % nd], 517.3, 516.5, 318.6, 255.3, d5[:nd])
        rep.append(_diff_report("undistortPoints (%d coefficients)" % nd, got.view(np.uint32), g["undist%d_out" % nd].view(np.uint32)))
    rep = [r for r in rep if r]
    assert not rep, "oracle primitive != OpenCV %s:\n  " % bytes(g["opencv_version"]).decode() + "\n  ".join(rep)


@pytest.mark.skipif(not os.path.exists(PRIMS), reason="tests/golden/reference_pinned/primitives.npz missing (parity unpinned)")
def test_oracle_primitives_equal_opencv():
    _check_primitives(np.load(PRIMS))


def test_the_checkers_themselves_on_oracle_made_stand_in_pins():
    """The pin checkers must be runnable the day real pins arrive: feed them files of the SAME layout whose expected values come
    from the oracle (so they must pass), then corrupt one byte of each kind (so they must fail).  Proves nothing about OpenCV."""
    import ctypes as C
    from oracle import oracle as O
    from orbslam2_amd import synth
    left, right = synth.stereo_pair(240, 180, seed=3)
    exl, exr = O.Extractor(nfeatures=300), O.Extractor(nfeatures=300)
    kl, dl = exl.extract(left); kr, dr = exr.extract(right)
    ur, dp, _ = O.stereo_matches(exl, exr, kl, dl, kr, dr, 90.0, 200.0)
    g = dict(params=np.array([240, 180, 300, 200.0, 90.0, np.float32(1.2), 8, 20, 7], np.float64), left=left, right=right, kl=kl, dl=dl, kr=kr, dr=dr, u_right=ur, depth=dp,
             scale=exl.scale_factors(), inv_scale=exl.inv_scale_factors(), sigma2=exl.sigma2(), inv_sigma2=exl.inv_sigma2(), opencv_version=np.frombuffer(b"stand-in", np.uint8))
    for l in range(8):
        g["pyr%d" % l] = exl.pyramid_level(l); g["blur%d" % l] = O.gaussian7(exl.pyramid_level(l))
    _check_case(g)
    bad = dict(g); bad["dl"] = dl.copy(); bad["dl"][5, 7] ^= 1
    with pytest.raises(AssertionError, match="left.desc"):
        _check_case(bad)
    bad = dict(g); bad["blur3"] = g["blur3"].copy(); bad["blur3"][2, 2] ^= 1
    with pytest.raises(AssertionError, match="blur3"):
        _check_case(bad)
    # primitives: same keys as pin.cpp's dump_primitives, values from the oracle
    L = O.lib()
    rng = np.random.default_rng(1)
    pr = {"opencv_version": np.frombuffer(b"stand-in", np.uint8)}
    xf = rng.uniform(-50, 50, 200).astype(np.float32); xd = xf.astype(np.float64) * 1.000000119
    pr["cvround_f_in"], pr["cvround_f_out"] = xf, np.array([L.orc_cv_round_f(C.c_float(float(v))) for v in xf], np.int32)
    pr["cvround_d_in"], pr["cvround_d_out"] = xd, np.array([L.orc_cv_round_d(C.c_double(float(v))) for v in xd], np.int32)
    ay, ax = rng.integers(-999, 999, 300).astype(np.float32), rng.integers(-999, 999, 300).astype(np.float32)
    pr["atan2_y"], pr["atan2_x"] = ay, ax
    pr["atan2_out"] = np.array([L.orc_fast_atan2(C.c_float(float(y)), C.c_float(float(x))) for y, x in zip(ay, ax)], np.float32)
    img = rng.integers(0, 256, (61, 97), dtype=np.uint8)
    pr["resize0_l0"] = img
    cur = img
    for l, (w, h) in enumerate(((81, 51), (67, 42), (56, 35)), 1):
        cur = O.resize_linear(cur, w, h); pr["resize0_l%d" % l] = cur
    pr["blur0"] = O.gaussian7(cur)
    pr["resize_any_src"] = img
    for j, (w, h) in enumerate(((80, 50), (48, 30), (90, 57), (40, 25), (24, 15))):
        pr["resize_any_%d" % j] = O.resize_linear(img, w, h)
    fi = rng.integers(0, 256, (36, 36), dtype=np.uint8)
    pr["fast0_img"] = fi
    for t in (20, 7):
        pr["fast0_t%d" % t] = np.stack(O.fast9_16(fi, t, True), 1).astype(np.int32)
    rgb = rng.integers(0, 256, (8, 9, 3), dtype=np.uint8); rgba = rng.integers(0, 256, (8, 9, 4), dtype=np.uint8)
    pr.update(cvt_rgb_in=rgb, cvt_rgba_in=rgba, cvt_rgb2gray=O.cvt_gray(rgb, True), cvt_bgr2gray=O.cvt_gray(rgb, False), cvt_rgba2gray=O.cvt_gray(rgba, True), cvt_bgra2gray=O.cvt_gray(rgba, False))
    X, Y = np.meshgrid(np.arange(97, dtype=np.float32), np.arange(61, dtype=np.float32))
    mx, my = (X * 0.9 + 1.3).astype(np.float32), (Y * 1.1 - 0.4).astype(np.float32)
    pr.update(remap_src=img, remap_mx=mx, remap_my=my, remap_out=O.remap_bilinear(img, mx, my))
    d5 = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)
    pts = rng.uniform(0, 600, (50, 2)).astype(np.float32)
    for nd in (4, 5):
        pr["undist%d_in" % nd] = pts; pr["undist%d_out" % nd] = O.undistort_points(pts, 517.3, 516.5, 318.6, 255.3, d5[:nd])
    _check_primitives(pr)
    bad = dict(pr); bad["fast0_t20"] = pr["fast0_t20"].copy()
    if len(bad["fast0_t20"]):
        bad["fast0_t20"][0, 2] += 1
        with pytest.raises(AssertionError, match="FAST 0 t=20"):
            _check_primitives(bad)
    bad = dict(pr); bad["atan2_out"] = pr["atan2_out"].copy(); bad["atan2_out"][3] = np.nextafter(bad["atan2_out"][3], np.float32(400))
    with pytest.raises(AssertionError, match="fastAtan2"):
        _check_primitives(bad)


def test_kit_end_to_end_with_oracle_made_containers(tmp_path):
    """export_inputs -> (a Python stand-in for the `pin` binary that writes the SAME record names from the oracle) -> import_pins ->
    the pin checker: proves the plumbing (file names, record names, dtypes, the 28-byte keypoint records, the params vector), so the first
    real run of the kit does not die on it.  Also: every record name the checkers read is a name pin.cpp writes."""
    import export_inputs as EX
    import import_pins as IP
    from oracle import oracle as O
    src = open(os.path.join(KIT, "pin.cpp")).read()
    written = set(re.findall(r'w\.put(?:_mat_u8|_f32|_i32|_keys)?\("([a-z0-9_]+)"', src))
    for prefix in ("pyr", "blur", "resize", "fast", "undist", "resize_any_"):  # names completed with std::to_string at run time
        assert ('"%s" + std::to_string' % prefix) in src or ('"%s' % prefix) in src, prefix
    need_case = {"params", "kl", "kr", "dl", "dr", "u_right", "depth", "scale", "inv_scale", "sigma2", "inv_sigma2"}
    need_prim = {"cvround_f_in", "cvround_f_out", "cvround_d_in", "cvround_d_out", "atan2_y", "atan2_x", "atan2_out", "resize_any_src", "cvt_rgb_in", "cvt_rgba_in",
                 "cvt_rgb2gray", "cvt_bgr2gray", "cvt_rgba2gray", "cvt_bgra2gray", "remap_src", "remap_mx", "remap_my", "remap_out"}
    assert need_case <= written and need_prim <= written, (need_case - written, need_prim - written)

    in_dir, pin_dir, out_dir = tmp_path / "in", tmp_path / "pins", tmp_path / "npz"
    EX.main(str(in_dir)); os.makedirs(pin_dir)
    row = next(r.split() for r in open(in_dir / "cases.txt").read().splitlines() if r.startswith("stereo_400x160_f300 "))
    name, lf, rf, nf, fx, bf, scale, levels, ini, mn = row[0], row[1], row[2], int(row[3]), float(row[4]), float(row[5]), float(row[6]), int(row[7]), int(row[8]), int(row[9])
    left, right = IP.read_pgm(str(in_dir / lf)), IP.read_pgm(str(in_dir / rf))
    exl, exr = O.Extractor(nfeatures=nf), O.Extractor(nfeatures=nf)
    kl, dl = exl.extract(left); kr, dr = exr.extract(right)
    ur, dp, _ = O.stereo_matches(exl, exr, kl, dl, kr, dr, bf, fx)
    code = {np.dtype(np.uint8): 0, np.dtype(np.int32): 1, np.dtype(np.float32): 2, np.dtype(np.float64): 3}

    def rec(f, k, v):
        v = np.ascontiguousarray(v)
        f.write(struct.pack("<I", len(k)) + k.encode() + struct.pack("<II", code[v.dtype], v.ndim) + struct.pack("<%dI" % v.ndim, *v.shape) + v.tobytes())
    with open(pin_dir / (name + ".pin"), "wb") as f:
        f.write(b"ORBPIN01")
        rec(f, "params", np.array([left.shape[1], left.shape[0], nf, fx, bf, scale, levels, ini, mn], np.float64))
        rec(f, "kl", np.frombuffer(kl.tobytes(), np.uint8).reshape(len(kl), 28)); rec(f, "kr", np.frombuffer(kr.tobytes(), np.uint8).reshape(len(kr), 28))
        rec(f, "dl", dl); rec(f, "dr", dr); rec(f, "u_right", ur); rec(f, "depth", dp); rec(f, "sad", np.zeros(len(kl), np.int32))
        for l in range(levels):
            rec(f, "pyr%d" % l, exl.pyramid_level(l)); rec(f, "blur%d" % l, O.gaussian7(exl.pyramid_level(l)))
        for k, v in (("scale", exl.scale_factors()), ("inv_scale", exl.inv_scale_factors()), ("sigma2", exl.sigma2()), ("inv_sigma2", exl.inv_sigma2())):
            rec(f, k, v)
    open(pin_dir / "opencv_version.txt", "w").write("stand-in\n")
    IP.main(str(in_dir), str(pin_dir), str(out_dir))
    g = np.load(out_dir / ("case_" + name + ".npz"))
    assert g["kl"].dtype == IP.KP_DTYPE and np.array_equal(g["left"], left)
    _check_case(g)
