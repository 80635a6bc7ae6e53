#!/usr/bin/env python3
"""One-off soak of PoseOptimization: random scenes, HIP vs oracle (flags / counts equal, pose within POSE_ATOL)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from orbslam2_amd import api  # noqa: E402
import test_pose as tp  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = api.Context(width=1241, height=376, nfeatures=2000, max_images=1, **tp.CAM)
bad_flags = bad_pose = 0
worst = 0.0
for i in range(n):
    rng = np.random.default_rng(100 + i)
    s = tp.scene(20000 + i, n=int(rng.integers(20, 3000)), mono_frac=float(rng.uniform(0, 1)), bad_frac=float(rng.uniform(0, 0.5)),
                 noise_px=float(rng.uniform(0, 1.5)), rv=tuple(rng.uniform(-0.08, 0.08, 3)), t=tuple(rng.uniform(-0.6, 0.6, 3)))
    T0 = np.eye(4, dtype=np.float32)
    Tr, outr, nr = tp._oracle(s, T0)
    Tg, outg, ng = ctx.pose_optimization(T0, s["keys"], s["ur"], s["has"], s["Xw"])
    d = float(np.abs(Tg - Tr).max())
    worst = max(worst, d)
    if not np.array_equal(outg, outr) or ng != nr:
        bad_flags += 1
        print("flags differ: case", i, int((outg != outr).sum()), "of", len(outr))
    elif d > tp.POSE_ATOL:
        bad_pose += 1
        print("pose differs: case", i, d)
ctx.close()
print("soak_pose: %d cases, %d flag mismatches, %d pose mismatches, worst pose diff %.3g" % (n, bad_flags, bad_pose, worst))
