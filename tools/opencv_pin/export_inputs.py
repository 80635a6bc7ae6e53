#!/usr/bin/env python3
"""Writes the inputs of tools/opencv_pin/pin as binary PGM files + cases.txt (numpy only; run from the repository root).

    python3 tools/opencv_pin/export_inputs.py /tmp/pin_in

The cases are the ones this repository's tests already use, regenerated from their seeds (orbslam2_amd/synth.py) or read from
the committed photographs (tests/golden/natural/*.png): the two golden stereo pairs of tools/make_golden.py, BASELINE.json's
five geometries at full size, the four natural pairs, and three parameter variants (scale factor 1.5, 4 levels, thresholds 30 / 10).
The KITTI-sized photograph is enlarged by the ORACLE's cv::resize restatement, i.e. by code under test -- its PGM is an input like any
other (whatever bytes it holds, the reference and the oracle are both run on those bytes)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from orbslam2_amd import synth  # noqa: E402


def pgm(path, img):
    img = np.ascontiguousarray(img, np.uint8)
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


def main(out):
    os.makedirs(out, exist_ok=True)
    rows = []

    def add(name, left, right, nf, fx, bf, scale=1.2, levels=8, ini=20, mn=7):
        pgm(os.path.join(out, name + "_L.pgm"), left); pgm(os.path.join(out, name + "_R.pgm"), right)
        rows.append("%s %s_L.pgm %s_R.pgm %d %.9g %.9g %.9g %d %d %d" % (name, name, name, nf, fx, bf, scale, levels, ini, mn))

    # name: (width, height, nfeatures, fx, bf, seed) -- tools/make_golden.py's CASES first, then BASELINE.json's geometries (seed 1234 as tests/test_gpu_parity.py)
    synth_cases = {
        "stereo_320x240_f500": (320, 240, 500, 300.0, 120.0, 1234), "stereo_400x160_f300": (400, 160, 300, 350.0, 140.0, 77),
        "small": (320, 240, 500, 300.0, 120.0, 1234), "kitti": (1241, 376, 2000, 718.856, 386.1448, 1234), "tum1": (640, 480, 1000, 517.3, 40.0, 1234),
        "euroc": (752, 480, 1200, 458.654, 47.9, 1234), "d435i": (1280, 720, 2500, 911.0, 45.5, 1234),
    }
    for name, (w, h, nf, fx, bf, seed) in synth_cases.items():
        l, r = synth.stereo_pair(w, h, seed=seed)
        add(name, l, r, nf, fx, bf)
    l, r = synth.stereo_pair(480, 320, seed=7)
    add("var_scale15", l, r, 600, 400.0, 160.0, scale=1.5)
    add("var_levels4", l, r, 600, 400.0, 160.0, levels=4)
    add("var_th30_10", l, r, 600, 400.0, 160.0, ini=30, mn=10)
    try:
        from tests import natural as N
        for name in N.PAIRS:
            l, r, d, nf = N.pair(name)
            fx, fy, cx, cy, bf = N.camera(l.shape[1], l.shape[0])
            add("natural_" + name, l, r, nf, fx, bf)
    except Exception as e:  # PIL missing: the synthetic cases alone still pin every primitive
        print("natural pairs skipped:", e)
    with open(os.path.join(out, "cases.txt"), "w") as f:
        f.write("\n".join(rows) + "\n")
    print("%d cases written to %s" % (len(rows), out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "pin_in")
