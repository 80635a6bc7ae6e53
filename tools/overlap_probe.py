#!/usr/bin/env python3
"""Probe: throughput vs stream groups (orbfe_set_streams), with and without stage-event profiling."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orbslam2_amd import api, synth
W, H = 1241, 376
P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pairs = [synth.stereo_pair(W, H, seed=1234 + i) for i in range(4)]
host = np.empty((2 * P, H, W), np.uint8)
for i in range(P):
    host[2 * i], host[2 * i + 1] = pairs[i % 4]
d = torch.from_numpy(host).cuda()
s = torch.cuda.current_stream().cuda_stream
for prof in (False, True):
    for G in (1, 2, 4, 8):
        ctx = api.Context(width=W, height=H, max_images=2 * P)
        ctx.set_streams(G)
        for _ in range(3): ctx.enqueue_stereo(d.data_ptr(), P, s)
        torch.cuda.synchronize()
        ctx.set_profiling(prof)
        t0 = time.perf_counter()
        K = 20
        for _ in range(K): ctx.enqueue_stereo(d.data_ptr(), P, s)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("prof=%d G=%d  %.0f pairs/s  %.3f ms/step" % (prof, G, P * K / dt, dt / K * 1e3))
        ctx.close()
