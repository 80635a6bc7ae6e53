"""Host half of liborbfe.so's Tracking matchers on the CPU, under AddressSanitizer + UBSan (tests/asan/resolve_harness.cpp compiles
orbslam2_amd/csrc/orbfe_match_resolve.h -- the very text orbfe_match.hip includes -- with a brute-force stand-in for the device's
window query) and compared with the C oracle: query builders, Frame::isInFrustum, the greedy replays on the top-K prefixes,
and the full-list fallback (forced on every third query in a second round).  No GPU needed; also the CPU-side sanitizer run
of the oracle itself (liborb_oracle_asan.so reproducing the golden fixture)."""
import os
import subprocess
import sys

import numpy as np

from oracle import oracle as O
from tests.scene_files import write_scene
from tests.test_matchers import CAM, LOG_SF, NL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_DIR = os.path.join(ROOT, "tests", "asan")


def _build():
    r = subprocess.run(["make", "-C", ASAN_DIR], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return os.path.join(ASAN_DIR, "resolve_harness")


def test_product_host_matcher_logic_under_asan_matches_the_oracle(tmp_path):
    exe = _build()
    sc = write_scene(tmp_path, seed=83)
    s, m, usable, max_d, min_d, normal, found = sc["s"], sc["m"], sc["usable"], sc["max_d"], sc["min_d"], sc["normal"], sc["found"]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "resolve harness ok" in r.stdout, r.stdout + r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    n_fallback = int(r.stdout.strip().split()[-1])
    assert n_fallback > 100  # the forced full-list replays really ran

    def out(name, dt=np.int32):
        return np.fromfile(tmp_path / name, dt)

    g = O.Grid(s["k"], *s["bounds"])
    tp = O.is_in_frustum(s["T_cur"], CAM, s["bounds"], s["pos"], normal, max_d, min_d, 0.5, LOG_SF, NL)
    got_tp = np.fromfile(tmp_path / "h_tp.bin", O.TP_DTYPE)
    assert np.array_equal(got_tp["in_view"], tp["in_view"]) and tp["in_view"].sum() > 200
    v = tp["in_view"] == 1
    for f in ("proj_x", "proj_y", "proj_xr", "level", "view_cos"):
        assert np.array_equal(got_tp[f][v], tp[f][v]), f
    ref_last = O.search_by_projection_last(g, s["ur"], s["d"], s["sf"], CAM, s["T_cur"], s["T_last"], s["pos"], s["desc_last"], usable,
                                           s["obs"], s["octave"], s["angle"], s["cur_has_obs"], 7.0, False, True)
    ref_pts = O.search_by_projection_points(g, s["ur"], s["d"], s["sf"], tp, s["desc_last"], s["obs"], s["cur_has_obs"], 3.0, 0.8)
    kf_ok = ((usable == 1) & (found == 0)).astype(np.int32)
    ref_kf = O.search_by_projection_kf(g, s["d"], s["sf"], CAM, s["T_cur"], LOG_SF, NL, s["pos"], s["desc_last"], kf_ok, s["angle"], max_d, min_d,
                                       s["cur_has_obs"], 10.0, 100, True)
    g2 = O.Grid(sc["k2"], *s["bounds"])
    prev = np.stack([sc["k1"]["x"], sc["k1"]["y"]], axis=1)
    ref_init = O.search_for_initialization(sc["k1"], s["d"], g2, sc["d2"], prev, 100, 0.9, True)
    for tag in ("", "_starved"):
        assert out("hn_last%s.bin" % tag)[0] == ref_last[1] and np.array_equal(out("h_last%s.bin" % tag), ref_last[0]), tag
        assert out("hn_pts%s.bin" % tag)[0] == ref_pts[1] and np.array_equal(out("h_pts%s.bin" % tag), ref_pts[0]), tag
        assert out("hn_kf%s.bin" % tag)[0] == ref_kf[1] and np.array_equal(out("h_kf%s.bin" % tag), ref_kf[0]), tag
        assert out("hn_init%s.bin" % tag)[0] == ref_init[2] and np.array_equal(out("h_init%s.bin" % tag), ref_init[0]), tag
        assert np.array_equal(out("h_prev%s.bin" % tag, np.float32).reshape(-1, 2), ref_init[1]), tag
    assert ref_last[1] > 100 and ref_pts[1] > 100 and ref_kf[1] > 30 and ref_init[2] > 100
    assert out("h_three.bin").tolist() == [-1, -1, -1, 0, 1, 2, 3, 7, 9]


def test_oracle_reproduces_the_golden_fixture_under_asan():
    """SURVEY.md section 5 (sanitizer run): the oracle built with -fsanitize=address,undefined reproduces the committed golden
    stereo fixture in a subprocess (libasan must be preloaded into the Python interpreter)."""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liborb_oracle_asan.so"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    libubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.exists(libasan), libasan
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from oracle import oracle as O
from orbslam2_amd import synth
g = np.load(%r)
w, h, nf, fx, bf, seed = g["params"]
left, right = synth.stereo_pair(int(w), int(h), seed=int(seed))
t = "liborb_oracle_asan.so"
exl, exr = O.Extractor(nfeatures=int(nf), target=t), O.Extractor(nfeatures=int(nf), target=t)
kl, dl = exl.extract(left); kr, dr = exr.extract(right)
ur, dp, m = O.stereo_matches(exl, exr, kl, dl, kr, dr, float(bf), float(fx))
assert np.array_equal(kl, g["kl"]) and np.array_equal(dl, g["dl"]) and np.array_equal(kr, g["kr"]) and np.array_equal(dr, g["dr"])
assert np.array_equal(ur, g["u_right"]) and np.array_equal(dp, g["depth"])
print("asan golden ok", len(kl), m)
""" % (ROOT, os.path.join(ROOT, "tests", "golden", "stereo_320x240_f500.npz"))
    env = dict(os.environ, LD_PRELOAD=libasan + ":" + libubsan, ASAN_OPTIONS="detect_leaks=0", PYTHONMALLOC="malloc")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "asan golden ok" in r.stdout, r.stdout + r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]


def test_png_decoder_under_asan_on_fixtures_truncations_and_corruptions():
    """orbfe_png.cpp (the translation unit liborbfe.so links) with -fsanitize=address,undefined: all 29 fixtures decode, and
    every truncation / seeded corruption of them (half with repaired CRCs so the damage reaches inflate, the filters and the
    pixel expansion) is either decoded or refused without a sanitizer report."""
    import glob
    _build()
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "png", "*.png")))
    assert len(files) == 29
    r = subprocess.run([os.path.join(ASAN_DIR, "png_harness")] + files, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0 and "png harness ok" in r.stdout, r.stdout + r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    decoded, refused = int(r.stdout.split()[3]), int(r.stdout.split()[5])
    assert decoded >= 29 and refused > 5000
