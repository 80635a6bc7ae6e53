#!/usr/bin/env python3
"""tools/opencv_pin/pin's output directory -> tests/golden/reference_pinned/*.npz (+ the inputs, so the tests need no regeneration).

    python3 tools/opencv_pin/import_pins.py /tmp/pin_in /tmp/pin_out [tests/golden/reference_pinned]

Container format "ORBPIN01" (pin.cpp: Writer): records of name, dtype code (0 u8, 1 i32, 2 f32, 3 f64, 4 u16), ndim, dims, data."""
import glob
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DTYPES = {0: np.uint8, 1: np.int32, 2: np.float32, 3: np.float64, 4: np.uint16}
KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def read_pin(path):
    b = open(path, "rb").read()
    assert b[:8] == b"ORBPIN01", path
    o, out = 8, {}
    while o < len(b):
        (nl,) = struct.unpack_from("<I", b, o); o += 4
        name = b[o:o + nl].decode(); o += nl
        t, nd = struct.unpack_from("<II", b, o); o += 8
        dims = struct.unpack_from("<%dI" % nd, b, o); o += 4 * nd
        dt = np.dtype(DTYPES[t])
        n = int(np.prod(dims)) if nd else 1
        out[name] = np.frombuffer(b, dt, n, o).reshape(dims).copy(); o += n * dt.itemsize
    return out


def read_pgm(path):
    b = open(path, "rb").read()
    parts = b.split(None, 4)
    assert parts[0] == b"P5" and parts[3] == b"255"
    w, h = int(parts[1]), int(parts[2])
    return np.frombuffer(b[len(b) - w * h:], np.uint8).reshape(h, w).copy()


def main(in_dir, pin_dir, out_dir):
    os.makedirs(out_dir, exist_ok=True)
    version = open(os.path.join(pin_dir, "opencv_version.txt")).read().strip()
    n = 0
    for path in sorted(glob.glob(os.path.join(pin_dir, "*.pin"))):
        name = os.path.splitext(os.path.basename(path))[0]
        d = read_pin(path)
        if name != "primitives":
            for k in ("kl", "kr"):
                d[k] = d[k].reshape(-1).view(KP_DTYPE) if d[k].size else np.zeros(0, KP_DTYPE)
            d["left"] = read_pgm(os.path.join(in_dir, name + "_L.pgm")); d["right"] = read_pgm(os.path.join(in_dir, name + "_R.pgm"))
        d["opencv_version"] = np.frombuffer(version.encode(), np.uint8)
        np.savez_compressed(os.path.join(out_dir, ("case_" if name != "primitives" else "") + name + ".npz"), **d)
        n += 1
    print("%d files -> %s (OpenCV %s)" % (n, out_dir, version))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "tests", "golden", "reference_pinned"))
