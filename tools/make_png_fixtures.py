#!/usr/bin/env python3
"""Generates tests/golden/png/*.png and tests/golden/png/expected.npz (needs Pillow; run once, results are committed).

Every fixture's expected pixels are what cv::imread(..., IMREAD_UNCHANGED) returns by OpenCV's conventions (BGR order, alpha
kept, low-depth grey scaled, 16 bit kept).  Where Pillow can decode the file to those pixels it is the REFERENCE (independent
third-party decoder: expected = Pillow's output with the channels reordered); the remaining cases (16-bit colour, which Pillow
truncates to 8 bit) take the samples the file was written from.  `pinned_by` in expected.npz records which."""
import io
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import png_oracle as P  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "png")
os.makedirs(OUT, exist_ok=True)
rng = np.random.default_rng(2024)
expected, pinned = {}, {}


def smooth(h, w, nch, full):
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([(x * (3 + c) + y * (5 - c)) % (full + 1) for c in range(nch)], axis=2)
    return ((base + rng.integers(0, max(2, full // 8), (h, w, nch))) % (full + 1)).astype(np.int64)


def pil_expect(data):
    """Pillow's decode mapped to imread's conventions, or None where Pillow cannot express the file's depth."""
    im = Image.open(io.BytesIO(data)); im.load()
    if im.mode in ("1",):
        return (np.array(im.convert("L"))).astype(np.uint8)
    if im.mode == "L":
        return np.array(im)
    if im.mode in ("I;16", "I;16B", "I"):
        return np.array(im).astype(np.uint16)
    if im.mode == "RGB":
        return np.array(im)[:, :, ::-1].copy()
    if im.mode == "RGBA":
        return np.array(im)[:, :, [2, 1, 0, 3]].copy()
    if im.mode == "LA":
        a = np.array(im); return np.stack([a[:, :, 0]] * 3 + [a[:, :, 1]], axis=2)
    if im.mode == "P":
        if "transparency" in im.info:
            return np.array(im.convert("RGBA"))[:, :, [2, 1, 0, 3]].copy()
        return np.array(im.convert("RGB"))[:, :, ::-1].copy()
    return None


def add(name, data, fallback=None, force_fallback=False):
    open(os.path.join(OUT, name + ".png"), "wb").write(data)
    e = None if force_fallback else pil_expect(data)
    pinned[name] = "pillow" if e is not None else "samples"
    expected[name] = e if e is not None else fallback
    assert expected[name] is not None, name


def pil_png(arr, mode=None, **kw):
    b = io.BytesIO(); Image.fromarray(arr, mode).save(b, format="PNG", **kw); return b.getvalue()


# --- files written by Pillow's encoder
g8 = smooth(37, 53, 1, 255)[:, :, 0].astype(np.uint8)
add("pil_gray8", pil_png(g8))
add("pil_gray16", pil_png(smooth(19, 31, 1, 65535)[:, :, 0].astype(np.uint16)))
add("pil_rgb8", pil_png(smooth(23, 17, 3, 255).astype(np.uint8)))
add("pil_rgba8", pil_png(smooth(16, 21, 4, 255).astype(np.uint8)))
add("pil_la8", pil_png(smooth(12, 14, 2, 255).astype(np.uint8), "LA"))
add("pil_bilevel", pil_png((smooth(13, 29, 1, 1)[:, :, 0] > 0), None))
pim = Image.fromarray(smooth(20, 20, 3, 255).astype(np.uint8)).quantize(colors=37)
b = io.BytesIO(); pim.save(b, format="PNG"); add("pil_palette", b.getvalue())
b = io.BytesIO(); pim.save(b, format="PNG", transparency=bytes([0, 128, 255] + [255] * 34)); add("pil_palette_trns", b.getvalue())
# --- files written by the fixture encoder: every filter type, mixed filters, Adam7, low depths, 16-bit colour, split IDAT
s8 = smooth(21, 34, 1, 255)
for ft in range(5):
    add("filter%d_gray8" % ft, P.encode(s8, 8, 0, filters=(ft,)))
add("mixed_filters_rgb8", P.encode(smooth(18, 25, 3, 255), 8, 2, filters=(4, 1, 3, 2, 0)))
add("mixed_filters_rgba8_split_idat", P.encode(smooth(27, 19, 4, 255), 8, 6, filters=(3, 4, 1), idat_split=97))
for hh, ww in ((1, 1), (3, 5), (9, 9), (17, 30)):
    add("adam7_gray8_%dx%d" % (ww, hh), P.encode(smooth(hh, ww, 1, 255), 8, 0, filters=(4, 2, 1, 3), interlace=True))
add("adam7_rgb8", P.encode(smooth(14, 11, 3, 255), 8, 2, filters=(1, 4), interlace=True))
add("adam7_bilevel", P.encode(smooth(10, 13, 1, 1), 1, 0, filters=(0, 2), interlace=True))
add("gray2", P.encode(smooth(11, 15, 1, 3), 2, 0, filters=(0, 1, 2)))
add("gray4", P.encode(smooth(11, 15, 1, 15), 4, 0, filters=(2, 3, 4)))
s = smooth(9, 12, 1, 65535); add("gray16_paeth", P.encode(s, 16, 0, filters=(4,)))
s = smooth(8, 10, 3, 65535); add("rgb16", P.encode(s, 16, 2, filters=(1, 4)), s[:, :, ::-1].astype(np.uint16), force_fallback=True)
s = smooth(8, 10, 4, 65535); add("rgba16", P.encode(s, 16, 6, filters=(3,)), s[:, :, [2, 1, 0, 3]].astype(np.uint16), force_fallback=True)
s = smooth(7, 9, 2, 65535); add("ga16", P.encode(s, 16, 4, filters=(2, 4)), np.stack([s[:, :, 0]] * 3 + [s[:, :, 1]], axis=2).astype(np.uint16), force_fallback=True)
pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
add("palette4", P.encode(smooth(10, 10, 1, 15), 4, 3, filters=(0,), palette=pal))
s = smooth(6, 8, 3, 255); s[2, 3] = (10, 20, 30); s[4, 1] = (10, 20, 30)
e = np.concatenate([s[:, :, ::-1], np.where((s == (10, 20, 30)).all(axis=2), 0, 255)[:, :, None]], axis=2).astype(np.uint8)
add("rgb8_trns_key", P.encode(s, 8, 2, filters=(1,), trns=bytes([0, 10, 0, 20, 0, 30])), e, force_fallback=True)
np.savez_compressed(os.path.join(OUT, "expected.npz"), names=np.array(sorted(expected)), pinned_by=np.array([pinned[n] for n in sorted(expected)]),
                    **{n: expected[n] for n in expected})
print("wrote %d fixtures, %d pinned by Pillow" % (len(expected), sum(v == "pillow" for v in pinned.values())))
for n in sorted(expected):  # the oracle must agree with the expectation before anything is committed
    got = P.decode(open(os.path.join(OUT, n + ".png"), "rb").read())
    assert got.dtype == expected[n].dtype and np.array_equal(got, expected[n]), (n, got.shape, expected[n].shape)
print("oracle == expected on all fixtures")
