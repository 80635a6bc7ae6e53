#!/usr/bin/env python3
"""One-off soak: many random scenes / sizes / thresholds, stereo frame HIP vs oracle, bit for bit.  python3 tools/soak.py [n]
SOAK_SEED=<int> shifts the case list; SOAK_GEOM=1 also draws the scale factor (1.04-2.3) and the number of levels (1-10); SOAK_PATCH=1 draws half_patch_size 8-18 and the edge
threshold for every second case."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orbslam2_amd import api, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = 0
for i in range(n):
    rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "7000")) + i)
    w, h = int(rng.integers(120, 700)), int(rng.integers(100, 420))
    nf = int(rng.integers(50, 2500))
    ini = int(rng.integers(8, 45)); mn = int(rng.integers(3, ini + 1))
    left, right = synth.stereo_pair(w, h, seed=int(os.environ.get("SOAK_SEED", "7000")) + 2000 + i)
    if i % 3 == 1:  # low contrast: many cells fall back to minTh
        f = float(rng.uniform(0.05, 0.4))
        left = np.clip((left.astype(np.float32) - 128) * f + 128, 0, 255).astype(np.uint8)
        right = np.clip((right.astype(np.float32) - 128) * f + 128, 0, 255).astype(np.uint8)
    if i % 7 == 3:  # pure noise: nearly every pixel passes FAST's quick test, many in both polarities (side list, its overflow)
        left = rng.integers(0, 256, (h, w)).astype(np.uint8)
        right = np.roll(left, -5, axis=1)
    if i % 7 == 5:  # period-6 / period-4 stripes + noise: ring pairs straddle the centre almost everywhere
        yy, xx = np.mgrid[0:h, 0:w]
        pat = (((xx // 3) + (yy // 2)) % 2) * 120 + 60 + rng.integers(-25, 26, (h, w))
        left = np.clip(pat, 0, 255).astype(np.uint8)
        right = np.roll(left, -4, axis=1)
    if i % 11 in (2, 6, 9):  # a photograph (tests/golden/natural) cut to this case's size, degraded the way real footage is
        from tests import natural as N
        ph = N.load(("china", "flower", "hopper")[i % 3])
        deg = (N.saturate, N.block_quantise, N.flatten_contrast)[(i // 11) % 3](ph)
        reps = (-(-h // deg.shape[0]), -(-(w + 16) // deg.shape[1]))
        big = np.tile(deg, reps)
        y0 = int(rng.integers(0, big.shape[0] - h + 1)); x0 = int(rng.integers(0, big.shape[1] - w - 15))
        dsp = int(rng.integers(2, 16))
        left = np.ascontiguousarray(big[y0:y0 + h, x0:x0 + w]); right = np.ascontiguousarray(big[y0:y0 + h, x0 + dsp:x0 + dsp + w])
    fx, bf = 0.7 * w, 0.2 * w
    kw = dict(nfeatures=nf, ini_th_fast=ini, min_th_fast=mn)
    if os.environ.get("SOAK_GEOM"):  # any scale factor / pyramid depth (the resize kernel choice is per level, from the column table)
        kw.update(scale_factor=float(np.float32(rng.uniform(1.04, 2.3))), nlevels=int(rng.integers(1, 11)))
    if os.environ.get("SOAK_PATCH") and i % 2:  # other patch geometries: describe_generic_kernel + the row-list launch of its own
        hp = int(rng.integers(8, 19)); edge = max(19, hp + 4) + int(rng.integers(0, 4))
        kw.update(half_patch_size=hp, patch_size=2 * hp + 1, edge_threshold=edge)
    try:
        O.Extractor(**kw)
        ctx = api.Context(width=w, height=h, fx=fx, fy=fx, cx=w / 2, cy=h / 2, bf=bf, **kw)
    except (ValueError, api.OrbfeError):
        skipped = globals().get("skipped", 0) + 1
        continue
    exl, exr = O.Extractor(**kw), O.Extractor(**kw)
    if i % 5 == 4:  # round 5: the context's own copy of the rBRIEF table (orbfe_set_pattern), here a random one inside the supported reach
        pat = rng.integers(-13, 14, (256, 4)).astype(np.int32)
        ctx.set_pattern(pat); exl.set_pattern(pat); exr.set_pattern(pat)
    out = ctx.stereo_frame(left, right)
    kl, dl = exl.extract(left); kr, dr = exr.extract(right)
    ur, dp, _ = O.stereo_matches(exl, exr, kl, dl, kr, dr, bf, fx)
    ok = (np.array_equal(out["kps_left"], kl.astype(api.KP_DTYPE)) and np.array_equal(out["kps_right"], kr.astype(api.KP_DTYPE)) and
          np.array_equal(out["desc_left"], dl) and np.array_equal(out["desc_right"], dr) and np.array_equal(out["u_right"], ur) and
          np.array_equal(out["depth"], dp))
    # round 4: the packed result block of the same (still resident) frame must expand to the very same records
    blk, lay = ctx.fetch_packed(2, api.PACK_STEREO)
    pl, pr = ctx.expand_packed(blk, lay, 0), ctx.expand_packed(blk, lay, 1)
    ok = ok and (pl["kps"].tobytes() == out["kps_left"].tobytes() and pr["kps"].tobytes() == out["kps_right"].tobytes() and
                 np.array_equal(pl["desc"], out["desc_left"]) and np.array_equal(pr["desc"], out["desc_right"]) and
                 pl["u_right"].tobytes() == out["u_right"].tobytes() and pl["depth"].tobytes() == out["depth"].tobytes())
    if i % 100 == 99:
        print("soak: %d cases done, %d mismatches" % (i + 1, bad), flush=True)
    if not ok:
        bad += 1
        print("MISMATCH case", i, w, h, kw)
    ctx.close()
print("soak: %d cases (%d refused by the oracle or orbfe_create), %d mismatches" % (n, globals().get("skipped", 0), bad))
sys.exit(1 if bad else 0)
