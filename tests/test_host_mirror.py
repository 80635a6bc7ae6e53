"""The C++ host mirror (orbslam2_amd/host: ORBextractor / ORBmatcher classes with the reference's names and
call patterns) drives the same C ABI and reproduces the oracle."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from orbslam2_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "orbslam2_amd", "host", "host_selftest")


def test_mirror_headers_keep_the_reference_names():
    ex = open(os.path.join(ROOT, "orbslam2_amd", "host", "ORBextractor.h")).read()
    for name in ("class ORBextractor", "GetLevels", "GetScaleFactor", "GetScaleFactors", "GetInverseScaleFactors", "GetScaleSigmaSquares",
                 "GetInverseScaleSigmaSquares", "mvImagePyramid", "int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST"):
        assert name in ex, name
    m = open(os.path.join(ROOT, "orbslam2_amd", "host", "ORBmatcher.h")).read()
    assert "PoseOptimization" in open(os.path.join(ROOT, "orbslam2_amd", "host", "Optimizer.h")).read()
    for name in ("class ORBmatcher", "TH_LOW = 50", "TH_HIGH = 100", "HISTO_LENGTH = 30", "DescriptorDistance", "SearchByProjection",
                 "SearchForInitialization", "ComputeThreeMaxima"):
        assert name in m, name


@pytest.mark.gpu
def test_cpp_selftest_matches_oracle(tmp_path):
    assert os.path.exists(EXE), "host_selftest not built (make -C orbslam2_amd/host)"
    w, h, nf, fx, bf = 480, 300, 500, 420.0, 150.0
    left, right = synth.stereo_pair(w, h, seed=55)
    lp, rp, pre = tmp_path / "l.raw", tmp_path / "r.raw", str(tmp_path / "out")
    left.tofile(lp); right.tofile(rp)
    r = subprocess.run([EXE, str(lp), str(rp), str(w), str(h), str(nf), str(fx), str(bf), pre], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    exl, exr = O.Extractor(nfeatures=nf), O.Extractor(nfeatures=nf)
    kl, dl = exl.extract(left); kr, dr = exr.extract(right)
    ur, dp, _ = O.stereo_matches(exl, exr, kl, dl, kr, dr, bf, fx)
    assert np.array_equal(np.fromfile(pre + ".kl", O.KP_DTYPE), kl) and np.array_equal(np.fromfile(pre + ".kr", O.KP_DTYPE), kr)
    assert np.array_equal(np.fromfile(pre + ".dl", np.uint8).reshape(-1, 32), dl)
    assert np.array_equal(np.fromfile(pre + ".dr", np.uint8).reshape(-1, 32), dr)
    assert np.array_equal(np.fromfile(pre + ".ur", np.float32), ur) and np.array_equal(np.fromfile(pre + ".dp", np.float32), dp)
    g2 = O.Grid(kr, 0.0, float(w), 0.0, float(h))
    prev = np.stack([kl["x"], kl["y"]], axis=1)
    m12, pm, n = O.search_for_initialization(kl, dl, g2, dr, prev, 100, 0.9, True)
    assert np.array_equal(np.fromfile(pre + ".m12", np.int32), m12) and np.array_equal(np.fromfile(pre + ".pm", np.float32).reshape(-1, 2), pm)
    assert ("init matches=%d" % n) in r.stdout and n > 20
    # Optimizer mirror: same inputs through the oracle
    has = np.fromfile(pre + ".has", np.uint8); Xw = np.fromfile(pre + ".Xw", np.float32).reshape(-1, 3)
    T0 = np.eye(4, dtype=np.float32); T0[:3, 3] = [0.05, -0.02, 0.03]
    Tr, outr, nr = O.pose_optimization(T0, kl, ur, has, Xw, exl.inv_sigma2(), fx, fx, w * 0.5, h * 0.5, bf)
    Tg = np.fromfile(pre + ".Tcw", np.float32).reshape(4, 4)
    assert ("pose inliers=%d" % nr) in r.stdout and nr > 50
    assert np.array_equal(np.fromfile(pre + ".outl", np.uint8), outr)
    assert np.abs(Tg - Tr).max() <= 2e-6 and np.abs(Tg - np.eye(4)).max() < 2e-3
    # RGB-D mirror on a distorted camera
    dist = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
    und = O.undistort_points(np.stack([kl["x"], kl["y"]], 1), fx, fx, w * 0.5, h * 0.5, dist)
    kun = kl.copy(); kun["x"], kun["y"] = und[:, 0], und[:, 1]
    urd, _ = O.stereo_from_rgbd(kl, kun, np.full((h, w), 2.0, np.float32), bf)
    assert np.array_equal(np.fromfile(pre + ".kun", O.KP_DTYPE), kun) and np.array_equal(np.fromfile(pre + ".urd", np.float32), urd)
    assert np.array_equal(np.fromfile(pre + ".bounds", np.float32), O.image_bounds(w, h, fx, fx, w * 0.5, h * 0.5, dist))


def test_multi_device_host_shards_like_dist():
    """ShardPairs (orbslam2_amd/host/multi_device.h) deals pairs exactly as orbslam2_amd/dist.py: shard_pairs does (one rule for the C++
    host and for bench.py's one-process-per-GPU path)."""
    from orbslam2_amd import dist as D
    text = open(os.path.join(ROOT, "orbslam2_amd", "host", "multi_device.h")).read()
    assert "for (int g = rank; g < n_pairs; g += world) mine.push_back(g);" in text
    for n, world in ((64, 8), (7, 3), (2, 4)):
        for r in range(world):
            assert D.shard_pairs(n, r, world) == list(range(r, n, world))
    assert "orbfe_stereo_batch_packed(" in text and "orbfe_expand_packed(" in text and "#include <hip" not in text  # the C ABI only (packed result block, expanded on the host): compiles with plain g++


@pytest.mark.gpu
def test_single_process_multi_context_host_matches_single_context(tmp_path):
    """One C++ process, three device contexts with a feeder thread each (on this one-GPU box all on device 0; context i on device
    i % device_count in general), 10 pairs per step dealt round-robin: twenty steps, every pair equal to the single-context batch, in
    frame order; pairs 0 and 9 against the oracle."""
    exe = os.path.join(ROOT, "orbslam2_amd", "host", "multi_device_selftest")
    assert os.path.exists(exe), "multi_device_selftest not built (make -C orbslam2_amd/host)"
    w, h, nf, fx, bf, P = 480, 300, 500, 420.0, 150.0, 10
    pairs = [synth.stereo_pair(w, h, seed=700 + i) for i in range(P)]
    buf = np.stack([np.stack(p) for p in pairs])  # [P][2][h][w]
    path, pre = tmp_path / "pairs.raw", str(tmp_path / "md")
    buf.tofile(path)
    r = subprocess.run([exe, str(path), str(P), str(w), str(h), str(nf), str(fx), str(bf), "3", pre], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "multi-device selftest ok" in r.stdout, r.stdout + r.stderr
    assert "contexts 3" in r.stdout and "mismatches 0" in r.stdout
    for g in (0, P - 1):
        exl, exr = O.Extractor(nfeatures=nf), O.Extractor(nfeatures=nf)
        kl, dl = exl.extract(pairs[g][0]); kr, dr = exr.extract(pairs[g][1])
        ur, dp, _ = O.stereo_matches(exl, exr, kl, dl, kr, dr, bf, fx)
        t = "%s_p%d" % (pre, g)
        assert np.array_equal(np.fromfile(t + ".kl", O.KP_DTYPE), kl) and np.array_equal(np.fromfile(t + ".kr", O.KP_DTYPE), kr)
        assert np.array_equal(np.fromfile(t + ".dl", np.uint8).reshape(-1, 32), dl) and np.array_equal(np.fromfile(t + ".dr", np.uint8).reshape(-1, 32), dr)
        assert np.array_equal(np.fromfile(t + ".ur", np.float32), ur) and np.array_equal(np.fromfile(t + ".dp", np.float32), dp)
