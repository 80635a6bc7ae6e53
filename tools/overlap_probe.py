#!/usr/bin/env python3
"""Probe: does running G independent contexts on G streams overlap stages and raise throughput?"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orbslam2_amd import api, synth
W, H = 1241, 376
P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pairs = [synth.stereo_pair(W, H, seed=1234 + i) for i in range(4)]
for G in (1, 2, 4, 8):
    per = P // G
    ctxs = [api.Context(width=W, height=H, max_images=2 * per) for _ in range(G)]
    host = np.empty((2 * per, H, W), np.uint8)
    for i in range(per):
        host[2 * i], host[2 * i + 1] = pairs[i % 4]
    d = torch.from_numpy(host).cuda()
    streams = [torch.cuda.Stream() for _ in range(G)]
    def step():
        for c, s in zip(ctxs, streams):
            c.enqueue_stereo(d.data_ptr(), per, s.cuda_stream)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K): step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("G=%d per=%d  %.0f pairs/s  %.3f ms/step" % (G, per, per * G * K / dt, dt / K * 1e3))
    for c in ctxs: c.close()
