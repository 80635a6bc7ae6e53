#!/usr/bin/env python3
"""Round 4: which transfer mechanisms overlap on this link.  64 KITTI pairs per step, pinned host memory in and out, L lanes
(fresh contexts, one stream each).
  download: `unpacked` = round 3's five hipMemcpyAsync (17.7 MB), `packed` = gather kernel + ONE copy (11.7 MB), `direct` = the gather
            kernel stores into the pinned block itself (ORBFE_PACK_DIRECT, no copy engine), `left` = direct, left images only
  upload  : `copy` = hipMemcpyAsync into a device buffer that is then read in place as level 0; `zero` = the chain's first kernel
            (ingest16_kernel, ORBFE_NO_INPLACE=1 contexts) reads the pinned host images across the link itself
    python3 tools/pcie_rate2.py > gpurun_out/pcie2.json"""
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
    def __init__(self, down, up):
        if up == "zero":
            os.environ["ORBFE_NO_INPLACE"] = "1"
        self.ctx = api.Context(**kw)
        os.environ.pop("ORBFE_NO_INPLACE", None)
        self.s = torch.cuda.Stream()
        self.down, self.up = down, up
        self.d_in = torch.empty_like(h_in, device=dev) if up == "copy" else None
        cap = self.ctx.capacity
        if down == "unpacked":
            self.sizes = [2 * P * cap * 28, 2 * P * cap * 32, 2 * P * 4, 2 * P * cap * 4, 2 * P * cap * 4]
            self.h_out = [torch.empty(n, dtype=torch.uint8).pin_memory() for n in self.sizes]
        else:
            self.flags = api.PACK_STEREO | (api.PACK_LEFT_ONLY if down == "left" else 0) | (api.PACK_DIRECT if down in ("direct", "left") else 0)
            self.lay = self.ctx.packed_layout(2 * P, self.flags)
            self.sizes = [int(self.lay.bytes)]
            self.h_out = [torch.empty(self.lay.bytes, dtype=torch.uint8).pin_memory()]

    def step(self):
        with torch.cuda.stream(self.s):
            if self.up == "copy":
                self.d_in.copy_(h_in, non_blocking=True)
                src = self.d_in.data_ptr()
            else:
                src = h_in.data_ptr()
            self.ctx.enqueue_stereo(src, P, self.s.cuda_stream)
            if self.down == "unpacked":
                self.ctx.fetch_batch_async(2 * P, *[h.data_ptr() for h in self.h_out], self.s.cuda_stream)
            else:
                self.ctx.fetch_batch_packed(2 * P, self.flags, self.h_out[0].data_ptr(), self.sizes[0], self.s.cuda_stream)


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


out = {"workload": "64 KITTI stereo pairs per step, pinned host memory in and out (59.7 MB up per step)"}
ref = None
for down, up, n in (("unpacked", "copy", 4), ("packed", "copy", 4), ("direct", "copy", 4), ("direct", "copy", 3), ("direct", "copy", 6), ("left", "copy", 4),
                    ("direct", "zero", 4), ("direct", "zero", 2), ("packed", "zero", 4)):
    lanes = [Lane(down, up) for _ in range(n)]
    r = run(lanes)
    r["bytes_down_per_step"] = int(sum(lanes[0].sizes))
    if down != "unpacked":  # the block of the last step expands to the context's own results
        l0 = lanes[0]
        a, b = l0.ctx.expand_packed(l0.h_out[0].numpy(), l0.lay, 0), l0.ctx.fetch_image(0, stereo=True)
        r["equals_unpacked"] = bool(a["kps"].tobytes() == b["kps"].tobytes() and np.array_equal(a["desc"], b["desc"]) and a["u_right"].tobytes() == b["u_right"].tobytes())
    out["down=%s up=%s lanes=%d" % (down, up, n)] = r
    for l in lanes:
        l.ctx.close()
print(json.dumps(out, indent=1))
