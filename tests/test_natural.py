"""Natural-image parity (tests/natural.py, tests/golden/natural): three photographs, four stereo pairs (native sizes + the KITTI
geometry of BASELINE.json), goldens written by tools/make_natural_fixtures.py from the oracle.

CPU: the oracle still reproduces the committed goldens on the committed PNGs (and the PNGs decode to what was hashed).
GPU: the HIP path reproduces them through the C ABI -- every keypoint field, descriptors, mvuRight, mvDepth, and the FAST + NMS
candidates of every level (counts for all, coordinates and scores for levels 0, 2 and 5) -- plus degraded versions of the same
photographs (saturated plateaus, 8 x 8 blocking, contrast flattened to sigma < 3) against the oracle run live."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import natural as N

NAMES = sorted(N.PAIRS)


def _golden(name):
    g = np.load(os.path.join(N.DIR, name + ".npz"))
    left, right, d, nf = N.pair(name)
    assert np.array_equal(N.digest(left, right), g["in_sha"]), "the committed PNGs / pair construction no longer give the hashed inputs"
    return g, left, right, d, nf


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_natural_golden(name):
    g, left, right, d, nf = _golden(name)
    h, w = left.shape
    fx, fy, cx, cy, bf = N.camera(w, h)
    exl, exr = O.Extractor(nfeatures=nf), O.Extractor(nfeatures=nf)
    kl, dl = exl.extract(left); kr, dr = exr.extract(right)
    ur, dp, m = O.stereo_matches(exl, exr, kl, dl, kr, dr, bf, fx)
    assert np.array_equal(kl, g["kl"]) and np.array_equal(dl, g["dl"]) and np.array_equal(kr, g["kr"]) and np.array_equal(dr, g["dr"])
    assert np.array_equal(ur, g["u_right"]) and np.array_equal(dp, g["depth"])
    assert [len(exl.level_candidates(l)[0]) for l in range(8)] == g["cand_counts_l"].tolist()
    # the pairs are cut d pixels apart: the matcher must find that disparity (a sanity check of the scene, not of parity)
    good = ur >= 0
    assert good.sum() > 0.4 * len(kl) and abs(float(np.median(kl["x"][good] - ur[good])) - d) < 0.25


def test_natural_images_are_not_the_synthetic_workload():
    """what the photographs add: a FAST + NMS candidate density of 0.5-2 % of the pyramid's pixels (the generator: 2.1 %, all over the frame), large flat regions,
    cells that only yield corners at minTh"""
    g, left, right, d, nf = _golden("flower_native")
    ex = O.Extractor(nfeatures=nf)
    ex.extract(left)
    sc = np.concatenate([ex.level_candidates(l)[2] for l in range(8)])
    px = sum(int(np.prod(ex.level_size(left.shape[1], left.shape[0], l))) for l in range(8))
    assert len(sc) / px < 0.01 and (sc < 20).any() and (sc >= 20).any()
    for f, chk in ((N.saturate, lambda a: (a == 0).mean() + (a == 255).mean() > 0.15), (N.block_quantise, lambda a: True),
                   (N.flatten_contrast, lambda a: a.std() < 3.0)):
        a = f(N.load("china"))
        assert a.dtype == np.uint8 and a.shape == (427, 640) and chk(a), f.__name__


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_reproduces_natural_golden(name):
    from orbslam2_amd import api
    g, left, right, d, nf = _golden(name)
    h, w = left.shape
    fx, fy, cx, cy, bf = N.camera(w, h)
    ctx = api.Context(width=w, height=h, nfeatures=nf, fx=fx, fy=fy, cx=cx, cy=cy, bf=bf)
    out = ctx.stereo_frame(left, right)
    for side, kk, dd in (("left", "kl", "dl"), ("right", "kr", "dr")):
        got = out["kps_" + side]
        assert len(got) == len(g[kk]), side
        for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
            assert np.array_equal(got[f].view(np.uint32), g[kk][f].view(np.uint32)), (side, f)  # by bits
        assert np.array_equal(out["desc_" + side], g[dd]), side
    assert np.array_equal(out["u_right"].view(np.uint32), g["u_right"].view(np.uint32))
    assert np.array_equal(out["depth"].view(np.uint32), g["depth"].view(np.uint32))
    for img, key in ((0, "cand_counts_l"), (1, "cand_counts_r")):
        assert [len(ctx.fetch_candidates(img, l)[0]) for l in range(8)] == g[key].tolist()
    for l in (0, 2, 5):
        assert np.array_equal(np.stack(ctx.fetch_candidates(0, l), axis=1), g["cand_l%d" % l]), "candidates of level %d" % l
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("photo", ["china", "flower", "hopper"])
@pytest.mark.parametrize("kind", ["saturated", "blocks8", "flat_sigma2.5", "saturated_blocks"])
def test_degraded_photographs_match_the_oracle(photo, kind):
    from orbslam2_amd import api
    base = N.load(photo)
    f = {"saturated": N.saturate, "blocks8": N.block_quantise, "flat_sigma2.5": N.flatten_contrast,
         "saturated_blocks": lambda a: N.block_quantise(N.saturate(a, 2.2), 40)}[kind]
    img = f(base)
    d, nf = 11, 900
    w = img.shape[1] - d
    left = np.ascontiguousarray(img[:, :w]); right = np.ascontiguousarray(img[:, d:d + w])
    right = np.clip(right.astype(np.int16) + ((np.arange(right.size).reshape(right.shape) * 7919 % 5) - 2), 0, 255).astype(np.uint8)  # +-2 grey levels
    h = left.shape[0]
    fx, fy, cx, cy, bf = N.camera(w, h)
    ctx = api.Context(width=w, height=h, nfeatures=nf, fx=fx, fy=fy, cx=cx, cy=cy, bf=bf)
    out = ctx.stereo_frame(left, right)
    exl, exr = O.Extractor(nfeatures=nf), O.Extractor(nfeatures=nf)
    kl, dl = exl.extract(left); kr, dr = exr.extract(right)
    ur, dp, _ = O.stereo_matches(exl, exr, kl, dl, kr, dr, bf, fx)
    for l in range(8):
        for a, b in zip(ctx.fetch_candidates(0, l), exl.level_candidates(l)):
            assert np.array_equal(a, b), "candidates level %d" % l
    assert np.array_equal(out["kps_left"], kl.astype(api.KP_DTYPE)) and np.array_equal(out["kps_right"], kr.astype(api.KP_DTYPE))
    assert np.array_equal(out["desc_left"], dl) and np.array_equal(out["desc_right"], dr)
    assert np.array_equal(out["u_right"], ur) and np.array_equal(out["depth"], dp)
    ctx.close()
