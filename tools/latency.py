#!/usr/bin/env python3
"""Single-frame latency of the host entry points (host images in, host keypoints / descriptors / depths out): what one
Tracking-thread Frame construction costs through the drop-in boundary, H2D and D2H copies included.
    python3 tools/latency.py > gpurun_out/latency.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orbslam2_amd import api, synth  # noqa: E402


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


out = {"unit": "ms per call, median of 50 (host buffers in and out)"}
left, right = synth.stereo_pair(1241, 376, seed=1234)
ctx = api.Context(width=1241, height=376, nfeatures=2000)
out["stereo KITTI 1241x376, 2000 features: orbfe_stereo_frame"] = round(timeit(lambda: ctx.stereo_frame(left, right)), 3)
out["mono KITTI: orbfe_extract"] = round(timeit(lambda: ctx.extract(left)), 3)
import torch  # noqa: E402  (device memory for the resident leg only)
d = torch.from_numpy(np.stack([left, right])).cuda()
torch.cuda.synchronize()


def resident():
    ctx.enqueue_stereo(d.data_ptr(), 1)
    ctx.synchronize()


out["stereo KITTI, pair already in HBM, results left in HBM: orbfe_enqueue_stereo + sync"] = round(timeit(resident), 3)
ctx.close()
l2, _, depth = synth.stereo_pair(1280, 720, seed=77, with_depth=True, bf=45.5)
ctx = api.Context(width=1280, height=720, nfeatures=2500, fx=911.0, fy=911.0, cx=640.0, cy=360.0, bf=45.5, max_images=1)
out["RGB-D D435i 1280x720, 2500 features: orbfe_rgbd_frame"] = round(timeit(lambda: ctx.rgbd_frame(l2, depth)), 3)
raw = np.clip(np.rint(depth * 1000.0), 0, 65535).astype(np.uint16)
out["RGB-D D435i, raw CV_16U depth: orbfe_rgbd_frame_u16"] = round(timeit(lambda: ctx.rgbd_frame(l2, raw, 0.001)), 3)
out["RGB-D D435i quadtree kernel"] = ctx.quadtree_kernel()
ctx.set_profiling(1)
for _ in range(20):
    ctx.rgbd_frame(l2, raw, 0.001)
st, calls = ctx.stage_times()
out["RGB-D D435i stage ms per frame (events at every boundary)"] = {k: round(v / max(calls, 1), 4) for k, v in st.items()}
ctx.set_profiling(0)
ctx.close()
print(json.dumps(out, indent=1))
