#!/usr/bin/env python3
"""Profiling aid (needs the `make cuts` build copied over liborbfe.so): phase stamps of pyr_tail_kernel's workgroups of image 0
(staging of the source strip, then one entry per fused level) in one 64-pair KITTI step."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orbslam2_amd import api, synth
W, H, P = 1241, 376, 64
pairs = [synth.stereo_pair(W, H, seed=1234 + i) for i in range(16)]
host = np.empty((2 * P, H, W), np.uint8)
for i in range(P):
    host[2 * i], host[2 * i + 1] = pairs[i % 16]
d = torch.from_numpy(host).cuda()
ctx = api.Context(width=W, height=H, max_images=2 * P)
for _ in range(3):
    ctx.enqueue_stereo(d.data_ptr(), P, 0)
ctx.synchronize()
ts = np.zeros(4096, np.int64)
ctx.L.orbfe_debug_timestamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert ctx.L.orbfe_debug_timestamps(ctx.h, ts.ctypes.data_as(C.c_void_p), 4096) == 0
ph = ts[3072:3072 + 128].reshape(8, 16).astype(np.float64) / 100.0
for s in range(8):
    v = ph[s][ph[s] > 0]
    if len(v) > 1:
        print("strip %d (us): tables + staging %.1f  levels: %s  total %.1f" % (s, v[1] - v[0], " ".join("%.1f" % x for x in np.diff(v)[1:]), v[-1] - v[0]))
n = 128 * 6
w = ts[:2 * n].reshape(n, 2).astype(np.float64) / 100.0
w = w[w[:, 0] > 0]
t0 = w[:, 0].min()
print("workgroups %d: launch span %.1f us; starts: median %.1f, 90%% %.1f, max %.1f; duration mean %.1f max %.1f" % (len(w), w[:, 1].max() - t0, np.median(w[:, 0] - t0), np.percentile(w[:, 0] - t0, 90), (w[:, 0] - t0).max(), (w[:, 1] - w[:, 0]).mean(), (w[:, 1] - w[:, 0]).max()))
