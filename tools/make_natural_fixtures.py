#!/usr/bin/env python3
"""Writes tests/golden/natural/{china,flower,hopper}.png (8-bit grey conversions of photographs that ship with scikit-learn and
matplotlib in the build container) and the oracle's outputs on the stereo pairs tests/natural.py cuts from them
(tests/golden/natural/<pair>.npz).  PARITY UNPINNED, as every golden here: the .npz pin HIP == oracle on natural images, not
oracle == OpenCV.   python3 tools/make_natural_fixtures.py [--images] [--golden]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_images(out_dir):
    import matplotlib
    import sklearn
    from PIL import Image
    src = {"china": os.path.join(os.path.dirname(sklearn.__file__), "datasets", "images", "china.jpg"),
           "flower": os.path.join(os.path.dirname(sklearn.__file__), "datasets", "images", "flower.jpg"),
           "hopper": os.path.join(os.path.dirname(matplotlib.__file__), "mpl-data", "sample_data", "grace_hopper.jpg")}
    for name, path in src.items():
        g = Image.open(path).convert("L")  # ITU-R 601 luma, 8 bit
        g.save(os.path.join(out_dir, name + ".png"), optimize=True)
        print(name, g.size, os.path.getsize(os.path.join(out_dir, name + ".png")), "bytes")


def make_golden(out_dir):
    from oracle import oracle as O
    from tests import natural as N
    for name in N.PAIRS:
        left, right, d, nf = N.pair(name)
        h, w = left.shape
        fx, fy, cx, cy, bf = N.camera(w, h)
        exl, exr = O.Extractor(nfeatures=nf), O.Extractor(nfeatures=nf)
        kl, dl = exl.extract(left)
        kr, dr = exr.extract(right)
        ur, dp, m = O.stereo_matches(exl, exr, kl, dl, kr, dr, bf, fx)
        cl = [np.stack(exl.level_candidates(l), axis=1).astype(np.int32) for l in range(8)]
        cr = [np.stack(exr.level_candidates(l), axis=1).astype(np.int32) for l in range(8)]
        good = ur >= 0
        disp = kl["x"][good] - ur[good]
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), in_sha=N.digest(left, right), kl=kl, dl=dl, kr=kr, dr=dr, u_right=ur, depth=dp,
                            cand_counts_l=np.array([len(c) for c in cl], np.int32), cand_counts_r=np.array([len(c) for c in cr], np.int32),
                            cand_l0=cl[0], cand_l2=cl[2], cand_l5=cl[5])
        px = sum(exl.level_size(w, h, l)[0] * exl.level_size(w, h, l)[1] for l in range(8))
        print("%-14s %4dx%-4d kps %4d / %4d  stereo %4d (median disparity %.2f, cut %d)  FAST candidates %d = %.2f %% of the pyramid's pixels"
              % (name, w, h, len(kl), len(kr), m, float(np.median(disp)) if len(disp) else -1, d, sum(len(c) for c in cl),
                 100.0 * sum(len(c) for c in cl) / px))


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden", "natural")
    os.makedirs(out, exist_ok=True)
    todo = sys.argv[1:] or ["--images", "--golden"]
    if "--images" in todo:
        make_images(out)
    if "--golden" in todo:
        make_golden(out)
