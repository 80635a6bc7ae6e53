#!/usr/bin/env python3
"""PCIe-inclusive throughput of the batched path (never bench.py's `value`): 64 KITTI pairs per step start in PINNED host
memory, results (keypoints, descriptors, uRight, depth at device capacity) end in pinned host memory.
  link    : the raw copies alone (up, down, both at once on two streams)
  serial  : upload -> chain -> download on one stream
  lanes L : L contexts, step k on lane k % L; `split`: a lane has separate upload / kernel / download streams chained by events
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
    def __init__(self, split):
        self.ctx = api.Context(**kw)
        self.s_k = torch.cuda.Stream()
        self.s_up = torch.cuda.Stream() if split else self.s_k
        self.s_dn = torch.cuda.Stream() if split else self.s_k
        self.split = split
        self.d_in = torch.empty_like(h_in, device=dev)
        cap = self.ctx.capacity
        self.sizes = [2 * P * cap * 28, 2 * P * cap * 32, 2 * P * 4, 2 * P * cap * 4, 2 * P * cap * 4]
        self.h_out = [torch.empty(n, dtype=torch.uint8).pin_memory() for n in self.sizes]
        self.e_up, self.e_k, self.e_dn = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
        self.first = True

    def step(self):
        if self.split and not self.first:
            self.s_up.wait_event(self.e_k)
        with torch.cuda.stream(self.s_up):
            self.d_in.copy_(h_in, non_blocking=True)
        if self.split:
            self.e_up.record(self.s_up)
            self.s_k.wait_event(self.e_up)
            if not self.first:
                self.s_k.wait_event(self.e_dn)
        self.ctx.enqueue_stereo(self.d_in.data_ptr(), P, self.s_k.cuda_stream)
        if self.split:
            self.e_k.record(self.s_k)
            self.s_dn.wait_event(self.e_k)
        self.ctx.fetch_batch_async(2 * P, *[h.data_ptr() for h in self.h_out], self.s_dn.cuda_stream)
        if self.split:
            self.e_dn.record(self.s_dn)
        self.first = False


def run(lanes, steps=24):
    vals = []
    for rep in range(4):
        for l in lanes:
            l.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            lanes[k % len(lanes)].step()
        torch.cuda.synchronize()
        vals.append(P * steps / (time.perf_counter() - t0))
    return {"pairs_per_s_median": round(sorted(vals)[len(vals) // 2]), "passes": [round(v) for v in vals]}


def link():
    d = torch.empty_like(h_in, device=dev)
    big = torch.empty(17_687_040, dtype=torch.uint8).pin_memory()
    dsrc = torch.empty(17_687_040, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def t(f, n=10):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    def up():
        with torch.cuda.stream(s1):
            d.copy_(h_in, non_blocking=True)

    def down():
        with torch.cuda.stream(s2):
            big.copy_(dsrc, non_blocking=True)

    def both():
        up(); down()
    tu, td, tb = t(up), t(down), t(both)
    return {"up_GBps": round(h_in.numel() / tu / 1e9, 1), "down_GBps": round(big.numel() / td / 1e9, 1),
            "both_ms": round(tb * 1e3, 3), "up_ms": round(tu * 1e3, 3), "down_ms": round(td * 1e3, 3),
            "bound_pairs_per_s_up_only": round(P / tu), "bound_pairs_per_s_duplex": round(P / tb)}


out = {"workload": "64 KITTI stereo pairs per step, pinned host memory in and out (59.7 MB up, 17.7 MB down per step)", "link": link()}
a = Lane(False)
out["serial (one stream)"] = run([a])
for n, split in ((2, False), (3, False), (4, False), (2, True), (3, True)):
    lanes = [Lane(split) for _ in range(n)]
    out["%d lanes%s" % (n, ", split streams" if split else "")] = run(lanes)
    for l in lanes:
        l.ctx.close()
cnt = np.frombuffer(a.h_out[2].numpy().tobytes(), np.int32)
out["keypoints_pair0"] = [int(cnt[0]), int(cnt[1])]
print(json.dumps(out, indent=1))
