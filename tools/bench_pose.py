#!/usr/bin/env python3
"""Throughput / latency of Optimizer::PoseOptimization on the device (orbfe_pose.hip) beside the CPU oracle.
    python3 tools/bench_pose.py > gpurun_out/pose.json
Workload: KITTI-like frames, N keypoints of which 80 % hold a map point, 15 % gross outliers, start pose = identity
(0.55 m / 2 deg away from the optimum: the constant-velocity prediction is normally much closer)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from orbslam2_amd import api  # noqa: E402
from oracle import oracle as O  # noqa: E402
import test_pose as tp  # noqa: E402

P, N = int(os.environ.get("POSE_PROBLEMS", 512)), 2000
scenes = [tp.scene(1000 + k, n=N) for k in range(min(P, 32))]
idx = [k % len(scenes) for k in range(P)]
off = (np.arange(P + 1) * N).astype(np.int32)
cat = {k: np.concatenate([scenes[i][k] for i in idx]) for k in ("keys", "ur", "has", "Xw")}
T0 = np.tile(np.eye(4, dtype=np.float32), (P, 1, 1))

ctx = api.Context(width=1241, height=376, nfeatures=2000, max_images=1, **tp.CAM)
dev = torch.device("cuda:0")
d = {k: torch.from_numpy(v.view(np.uint8).reshape(-1) if v.dtype.fields else v).to(dev) for k, v in cat.items()}
d_off = torch.from_numpy(off).to(dev)
d_T0 = torch.from_numpy(T0).to(dev)
d_T = d_T0.clone()
d_out = torch.zeros(P * N, dtype=torch.uint8, device=dev)
d_n = torch.zeros(P, dtype=torch.int32, device=dev)
tstream = torch.cuda.Stream()  # a real stream: 0 would select the context's own stream, which torch events do not see
torch.cuda.set_stream(tstream)
stream = tstream.cuda_stream


def launch(np_):
    d_T.copy_(d_T0)
    ctx._check(ctx.L.orbfe_enqueue_pose_optimization(ctx.h, np_, d_off.data_ptr(), d["keys"].data_ptr(), d["ur"].data_ptr(), d["has"].data_ptr(),
                                                     d["Xw"].data_ptr(), d_T.data_ptr(), d_out.data_ptr(), d_n.data_ptr(), N, stream))


out = {"workload": "N=%d keypoints/frame, 80%% with a map point, 15%% gross outliers, identity start" % N}
for np_ in (1, 64, P):
    for _ in range(3):
        launch(np_)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        launch(np_)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out["device, %d problems resident" % np_] = {"ms": round(ms, 4), "problems_per_s": round(np_ / ms * 1e3, 1)}

# host entry point, one frame (Tracking thread view)
s0 = scenes[0]
I4 = np.eye(4, dtype=np.float32)
for _ in range(5):
    ctx.pose_optimization(I4, s0["keys"], s0["ur"], s0["has"], s0["Xw"])
ts = []
for _ in range(50):
    t0 = time.perf_counter()
    Tg, og, ng = ctx.pose_optimization(I4, s0["keys"], s0["ur"], s0["has"], s0["Xw"])
    ts.append((time.perf_counter() - t0) * 1e3)
out["host entry orbfe_pose_optimization, one frame, median ms"] = round(float(np.median(ts)), 4)

ts = []
for k in range(20):
    s = scenes[k % len(scenes)]
    t0 = time.perf_counter()
    Tr, orr, nr = tp._oracle(s, I4)
    ts.append((time.perf_counter() - t0) * 1e3)
out["cpu oracle (1 core), one frame, median ms"] = round(float(np.median(ts)), 4)
Tr, orr, nr = tp._oracle(s0, I4)
out["check"] = {"n_inliers_gpu": int(ng), "n_inliers_cpu": int(nr), "flags_equal": bool(np.array_equal(og, orr)),
                "max_abs_pose_diff": float(np.abs(Tg - Tr).max())}
ctx.close()
print(json.dumps(out, indent=1))
