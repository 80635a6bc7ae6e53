#!/usr/bin/env python3
"""Cycle breakdown of one PoseOptimization problem (instrumented build, -DORBFE_POSE_TIMING; see tools/pose_phases.sh)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from orbslam2_amd import api  # noqa: E402
import test_pose as tp  # noqa: E402

ctx = api.Context(width=1241, height=376, nfeatures=2000, max_images=1, **tp.CAM)
s = tp.scene(1000, n=2000)
I4 = np.eye(4, dtype=np.float32)
ctx.pose_optimization(I4, s["keys"], s["ur"], s["has"], s["Xw"])
cyc = (C.c_longlong * 8)()
ctx.L.orbfe_pose_debug_cycles(cyc, 1)
ctx.pose_optimization(I4, s["keys"], s["ur"], s["has"], s["Xw"])
ctx.L.orbfe_pose_debug_cycles(cyc, 1)
names = ["solve", "exp+mul", "edge loop", "reduction", "passes"]
print({n: int(cyc[i]) for i, n in enumerate(names)})
