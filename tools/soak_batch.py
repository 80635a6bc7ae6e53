#!/usr/bin/env python3
"""Soak of the LARGE-BATCH launch plan (batches of 64 images and more: one pyramid level per launch, the blur of every level riding in
FAST's launch, resize word bases from the table): random geometries and parameters, 32-40 stereo pairs per orbfe_enqueue_stereo call
built from 6 distinct pairs, every slot compared with the oracle bit for bit.  python3 tools/soak_batch.py [n cases]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from orbslam2_amd import api, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
base = int(os.environ.get("SOAK_SEED", "900000"))
bad = skipped = 0
for i in range(n):
    rng = np.random.default_rng(base + i)
    w, h = int(rng.integers(160, 640)), int(rng.integers(120, 400))
    kw = dict(nfeatures=int(rng.integers(100, 1500)), ini_th_fast=int(rng.integers(10, 40)))
    kw["min_th_fast"] = int(rng.integers(3, kw["ini_th_fast"] + 1))
    if i % 3 == 0:
        kw.update(scale_factor=float(np.float32(rng.uniform(1.1, 1.6))), nlevels=int(rng.integers(3, 10)))
    P, ND = int(rng.integers(32, 41)), 6
    fx, bf = 0.7 * w, 0.2 * w
    try:
        O.Extractor(**kw)
        ctx = api.Context(width=w, height=h, fx=fx, fy=fx, cx=w / 2, cy=h / 2, bf=bf, max_images=2 * P, **kw)
    except (ValueError, api.OrbfeError):
        skipped += 1
        continue
    assert ctx.blur_ride_from(2 * P) == 0  # the plan under test: every level's blur in FAST's launch
    distinct = [synth.stereo_pair(w, h, seed=base + 1000 * i + k) for k in range(ND)]
    refs = []
    for l, r in distinct:
        exl, exr = O.Extractor(**kw), O.Extractor(**kw)
        kl, dl = exl.extract(l); kr, dr = exr.extract(r)
        ur, dp, _ = O.stereo_matches(exl, exr, kl, dl, kr, dr, bf, fx)
        refs.append((kl, dl, kr, dr, ur, dp))
    order = [(5 * j + 1) % ND for j in range(P)]
    host = np.empty((2 * P, h, w), np.uint8)
    for j, k in enumerate(order):
        host[2 * j], host[2 * j + 1] = distinct[k]
    dev = torch.from_numpy(host).cuda()
    ctx.enqueue_stereo(dev.data_ptr(), P, 0)
    ctx.synchronize()
    ok = True
    for j, k in enumerate(order):
        kl, dl, kr, dr, ur, dp = refs[k]
        a, b = ctx.fetch_image(2 * j, stereo=True), ctx.fetch_image(2 * j + 1)
        ok = ok and (a["kps"].tobytes() == kl.tobytes() and np.array_equal(a["desc"], dl) and a["u_right"].tobytes() == ur.tobytes() and a["depth"].tobytes() == dp.tobytes()
                     and b["kps"].tobytes() == kr.tobytes() and np.array_equal(b["desc"], dr))
    if not ok:
        bad += 1
        print("MISMATCH case", i, w, h, P, kw, flush=True)
    if i % 10 == 9:
        print("soak_batch: %d cases done, %d mismatches" % (i + 1, bad), flush=True)
    ctx.close()
print("soak_batch: %d cases (%d refused), %d mismatches" % (n, skipped, bad))
sys.exit(1 if bad else 0)
