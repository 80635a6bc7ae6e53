#!/usr/bin/env python3
"""Bring-up aid: per-phase shader-clock timeline of the quadtree kernel for (image 0, level L)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
os.environ["ORBFE_OT2_STOP"] = str(100 + lvl)
import torch
from orbslam2_amd import api, synth
W, H, P = 1241, 376, 32
pairs = [synth.stereo_pair(W, H, seed=1234 + i) for i in range(4)]
host = np.empty((2 * P, H, W), np.uint8)
for i in range(P):
    host[2 * i], host[2 * i + 1] = pairs[i % 4]
d = torch.from_numpy(host).cuda()
ctx = api.Context(width=W, height=H, max_images=2 * P)
for _ in range(3):
    ctx.enqueue_stereo(d.data_ptr(), P, 0)
ctx.synchronize()
ts = np.zeros(4096, np.int64)
ctx.L.orbfe_debug_timestamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert ctx.L.orbfe_debug_timestamps(ctx.h, ts.ctypes.data_as(C.c_void_p), 4096) == 0
t0 = ts[0]
us = lambda a, b: (ts[b] - ts[a]) / 100.0  # clock64 ticks at 100 MHz? printed raw too
print("level", lvl, "raw", ts[:6] - t0)
print("scan %.0f  gather %.0f  roots %.0f  passes %.0f  final %.0f  total %.0f (ticks)" % (ts[1]-ts[0], ts[2]-ts[1], ts[3]-ts[2], ts[4]-ts[3], ts[5]-ts[4], ts[5]-ts[0]))
for it in range(20):
    b = 8 + 8 * it
    if ts[b] == 0: break
    seg = [ts[b + k + 1] - ts[b + k] for k in range(5)]
    nxt = ts[b + 8] if ts[b + 8] else ts[4]
    print("pass %2d: init+walk1 %6d  scan4+order %6d  kscan %6d  decide %6d  build %6d  walk2 %6d  | total %6d" % (it, seg[0], seg[1], seg[2], seg[3], seg[4], nxt - ts[b + 5], nxt - ts[b]))
