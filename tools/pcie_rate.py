#!/usr/bin/env python3
"""PCIe-inclusive throughput of the batched path (never bench.py's `value`): 64 KITTI pairs per step start in PINNED host
memory, results (keypoints, descriptors, uRight, depth) end in pinned host memory.
  serial : upload -> chain -> download on one stream
  overlap: two contexts / two streams, step k+1's upload overlaps step k's chain and download
    python3 tools/pcie_rate.py > gpurun_out/pcie.json"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orbslam2_amd import api, synth  # noqa: E402

W, H, P = 1241, 376, 64
kw = dict(width=W, height=H, nfeatures=2000, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448, max_images=2 * P)
pairs = [synth.stereo_pair(W, H, seed=1234 + i) for i in range(4)]
host = np.empty((2 * P, H, W), np.uint8)
for i in range(P):
    host[2 * i], host[2 * i + 1] = pairs[i % 4]
h_in = torch.from_numpy(host).pin_memory()
dev = torch.device("cuda:0")


class Lane:
    def __init__(self):
        self.ctx = api.Context(**kw)
        self.stream = torch.cuda.Stream()
        self.d_in = torch.empty_like(h_in, device=dev)
        cap = self.ctx.capacity
        self.sizes = [2 * P * cap * 28, 2 * P * cap * 32, 2 * P * 4, 2 * P * cap * 4, 2 * P * cap * 4]
        self.h_out = [torch.empty(n, dtype=torch.uint8).pin_memory() for n in self.sizes]

    def step(self):
        with torch.cuda.stream(self.stream):
            self.d_in.copy_(h_in, non_blocking=True)
            self.ctx.enqueue_stereo(self.d_in.data_ptr(), P, self.stream.cuda_stream)
            self.ctx.fetch_batch_async(2 * P, *[h.data_ptr() for h in self.h_out], self.stream.cuda_stream)


def run(lanes, steps=20):
    best = None
    for rep in range(3):  # best of three passes: the first pass after a context is created sometimes runs at half speed
        for l in lanes:
            l.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            lanes[k % len(lanes)].step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best = dt
    return P * steps / best, best / steps * 1e3


a, b = Lane(), Lane()
out = {"workload": "64 KITTI stereo pairs per step, pinned host memory in and out (59.7 MB up, %.1f MB down per step)" % (sum(a.sizes) / 1e6)}
v, ms = run([a])
out["serial (one stream)"] = {"pairs_per_s": round(v), "ms_per_step": round(ms, 3)}
v, ms = run([a, b])
out["overlapped (two contexts, two streams)"] = {"pairs_per_s": round(v), "ms_per_step": round(ms, 3)}
cnt = np.frombuffer(a.h_out[2].numpy().tobytes(), np.int32)
out["keypoints_pair0"] = [int(cnt[0]), int(cnt[1])]
print(json.dumps(out, indent=1))
