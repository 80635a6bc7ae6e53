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


if __name__ == "__main__":
    main()
