#!/usr/bin/env python3
"""Profiling aid (needs the `make cuts` build copied over liborbfe.so): start / end of every octree3_kernel workgroup of one
64-pair KITTI step, per level: when its workgroups start, how long they run, when the launch ends."""
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
t = ts[:2048].reshape(128, 8, 2).astype(np.float64) / 100.0  # us
t0 = t[:, :, 0].min()
print("launch span %.1f us (first start to last end)" % (t[:, :, 1].max() - t0))
for l in range(8):
    s, e = t[:, l, 0] - t0, t[:, l, 1] - t0
    print("level %d: start %5.1f .. %5.1f  duration mean %5.1f max %5.1f  end max %5.1f" % (l, s.min(), s.max(), (e - s).mean(), (e - s).max(), e.max()))
bl = ts[2560:4096].reshape(768, 2).astype(np.float64) / 100.0
bl = bl[bl[:, 0] > 0]
if len(bl):
    print("blur workgroups riding in the launch: %d, start %5.1f .. %5.1f, duration mean %4.1f max %4.1f, end max %5.1f" % (len(bl), bl[:, 0].min() - t0, bl[:, 0].max() - t0, (bl[:, 1] - bl[:, 0]).mean(), (bl[:, 1] - bl[:, 0]).max(), bl[:, 1].max() - t0))
ph = ts[2048:2048 + 512].reshape(8, 64).astype(np.float64) / 100.0
for l in range(8):
    v = ph[l][ph[l] > 0]
    if len(v) > 1:
        d = np.diff(v)
        # after the roots: one entry per full pass, then the largest-first pass as child counts + scan | ranking | k scan | nproc | flag scan | build
        print("level %d phases (us): partials %.1f  points %.1f  pyramid %.1f  roots, passes: %s  final %.1f" % (l, d[0], d[1], d[2], " ".join("%.1f" % x for x in d[3:-1]), d[-1]))
