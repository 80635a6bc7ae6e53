#!/usr/bin/env python3
"""BASELINE.json config 4 as worded: 64 pairs in flight over 8 GPUs = 8 pairs per GPU per step.  One chain of 15 dependent
launches over 8 pairs is latency-bound (round 2: 0.190 ms per step, 42 k pairs/s); consecutive steps are independent, so keep
several step chains in flight: C contexts, step k on context k % C and its own stream.
    python3 tools/small_batch.py [--pairs 8] [--chains 1,2,3,4,6] [--groups 1] [--steps 200]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(api, torch, d_images, P, chains, groups, steps, warmup=20, repeats=5):
    W, H = 1241, 376
    ctxs = [api.Context(width=W, height=H, nfeatures=2000, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448, max_images=2 * P) for _ in range(chains)]
    for c in ctxs:
        c.set_streams(groups)
    streams = [torch.cuda.Stream() for _ in range(chains)]
    ptr = d_images.data_ptr()

    def run(n):
        for k in range(n):
            ctxs[k % chains].enqueue_stereo(ptr, P, streams[k % chains].cuda_stream)
    run(warmup)
    torch.cuda.synchronize()
    vals, host_us = [], []
    for _ in range(repeats):
        t0 = time.perf_counter()
        run(steps)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        vals.append(P * steps / (time.perf_counter() - t0))
        host_us.append((t1 - t0) / steps * 1e6)
    counts = ctxs[0].fetch_counts(2 * P)
    for c in ctxs:
        c.close()
    vals.sort()
    return {"pairs": P, "chains_in_flight": chains, "stream_groups": groups, "value": vals[len(vals) // 2], "min": vals[0], "max": vals[-1],
            "ms_per_step": P / vals[len(vals) // 2] * 1e3, "host_enqueue_us_per_step": sorted(host_us)[len(host_us) // 2], "keypoints_pair0": [int(counts[0]), int(counts[1])]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--chains", default="1,2,3,4,6")
    ap.add_argument("--groups", default="1")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import torch
    from orbslam2_amd import api, synth
    P = a.pairs
    host = np.empty((2 * P, 376, 1241), np.uint8)
    for i in range(P):
        host[2 * i], host[2 * i + 1] = synth.stereo_pair(1241, 376, seed=1234 + i)
    d_images = torch.from_numpy(host).cuda()
    rows = []
    for g in [int(x) for x in a.groups.split(",")]:
        for c in [int(x) for x in a.chains.split(",")]:
            r = measure(api, torch, d_images, P, c, g, a.steps)
            rows.append(r)
            print(json.dumps(r), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
