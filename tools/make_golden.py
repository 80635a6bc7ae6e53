#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (committed generator of the fixtures).

PARITY UNPINNED: the reference has no fixtures for this path and cannot be built here
(no OpenCV), so the goldens pin "HIP == oracle" and "oracle == itself over time", not
"oracle == OpenCV".  Inputs are regenerated from seeds; outputs are stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from orbslam2_amd import synth  # noqa: E402

CASES = {
    # name: (width, height, nfeatures, fx, bf, seed)
    "stereo_320x240_f500": (320, 240, 500, 300.0, 120.0, 1234),
    "stereo_400x160_f300": (400, 160, 300, 350.0, 140.0, 77),
}


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name, (w, h, nf, fx, bf, seed) in CASES.items():
        left, right = synth.stereo_pair(w, h, seed=seed)
        exl, exr = O.Extractor(nfeatures=nf), O.Extractor(nfeatures=nf)
        kl, dl = exl.extract(left)
        kr, dr = exr.extract(right)
        ur, dp, m = O.stereo_matches(exl, exr, kl, dl, kr, dr, bf, fx)
        cand = [np.stack(exl.level_candidates(l), axis=1) for l in range(8)]
        np.savez_compressed(os.path.join(out_dir, name + ".npz"),
                            params=np.array([w, h, nf, fx, bf, seed], np.float64),
                            left_sha=np.frombuffer(__import__("hashlib").sha256(left.tobytes()).digest(), np.uint8),
                            kl=kl, dl=dl, kr=kr, dr=dr, u_right=ur, depth=dp,
                            cand_counts=np.array([len(c) for c in cand], np.int32),
                            cand_l3=cand[3].astype(np.int32),
                            pyr7=exl.pyramid_level(7), blur7=O.gaussian7(exl.pyramid_level(7)))
        print(name, len(kl), len(kr), m)


if __name__ == "__main__" and "--widenings" not in sys.argv:
    main()


def make_widenings(path):
    """Golden vectors of the section 8(f) widenings: PoseOptimization, undistortPoints, cvtColor->GRAY, remap.  Inputs are
    regenerated from seeds by tests/test_golden.py; only the expected outputs (and input digests) are stored."""
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_pose as tp
    s = tp.scene(4242, n=600)
    T, outl, ninl = tp._oracle(s, np.eye(4, dtype=np.float32))
    rng = np.random.default_rng(77)
    pts = rng.uniform(-10, 650, (200, 2)).astype(np.float32)
    dist = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)
    und = O.undistort_points(pts, 517.3, 516.5, 318.6, 255.3, dist)
    bounds = O.image_bounds(640, 480, 517.3, 516.5, 318.6, 255.3, dist)
    rgb = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    gray = O.cvt_gray(rgb, True)
    gray_bgr14 = O.cvt_gray(rgb, False, True)
    X, Y = np.meshgrid(np.arange(64, dtype=np.float32), np.arange(48, dtype=np.float32))
    mx = (X * 0.97 + 0.013 * Y + 1.37).astype(np.float32); my = (Y * 1.02 - 0.011 * X - 0.81).astype(np.float32)
    rem = O.remap_bilinear(gray, mx, my)
    np.savez_compressed(path, pose_T=T, pose_outlier=outl, pose_ninl=np.int32(ninl),
                        pose_in_sha=np.frombuffer(hashlib.sha256(s["keys"].tobytes() + s["Xw"].tobytes()).digest(), np.uint8),
                        und=und, bounds=bounds, gray=gray, gray_bgr14=gray_bgr14, remap=rem)


if __name__ == "__main__" and "--widenings" in __import__("sys").argv:
    make_widenings(os.path.join(ROOT, "tests", "golden", "widenings_r01.npz"))
