"""Input side, SURVEY.md section 8f-4: cv::imread(path, IMREAD_UNCHANGED) for PNG files (orbfe_png_* in liborbfe.so, host code).

tests/golden/png holds 29 small PNG files (all colour types and depths, every scanline filter, Adam7, split IDAT, palette /
tRNS) with the pixels imread returns for them; 25 expectations are Pillow's decode of the file (an independent third-party
decoder; generator tools/make_png_fixtures.py), so for this row the parity is PINNED, not merely oracle == kernel.
No GPU needed (the decoder is host code by design, see orbfe_png.cpp)."""
import glob
import os
import struct
import zlib

import numpy as np
import pytest

from oracle import png_oracle as P
from orbslam2_amd import api

PNG_DIR = os.path.join(os.path.dirname(__file__), "golden", "png")
EXP = np.load(os.path.join(PNG_DIR, "expected.npz"))
NAMES = [str(n) for n in EXP["names"]]


def _file(name):
    return open(os.path.join(PNG_DIR, name + ".png"), "rb").read()


def test_fixture_inventory():
    assert len(NAMES) == 29 and sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(PNG_DIR, "*.png"))) == NAMES
    assert (EXP["pinned_by"] == "pillow").sum() == 25


@pytest.mark.parametrize("name", NAMES)
def test_oracle_decodes_fixture(name):
    got = P.decode(_file(name))
    assert got.dtype == EXP[name].dtype and np.array_equal(got, EXP[name])


@pytest.mark.parametrize("name", NAMES)
def test_native_decoder_decodes_fixture(name):
    data = _file(name)
    e = EXP[name]
    w, h, ch, bd = api.png_info(data)
    assert (h, w) == e.shape[:2] and ch == (1 if e.ndim == 2 else e.shape[2]) and bd == 8 * e.dtype.itemsize
    got = api.png_decode(data)
    assert got.dtype == e.dtype and np.array_equal(got, e)


def test_native_decoder_random_images_vs_oracle():
    """Seeded random images through the fixture encoder (random colour type / depth / filters / interlace / IDAT split):
    native decoder == Python oracle, bit for bit, including 1-pixel-wide and 1-row images."""
    rng = np.random.default_rng(77)
    kinds = [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 2), (3, 4), (3, 8), (4, 8), (4, 16), (6, 8), (6, 16)]
    for it in range(120):
        color, depth = kinds[it % len(kinds)]
        nch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
        h, w = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        s = rng.integers(0, 1 << depth, (h, w, nch))
        pal = rng.integers(0, 256, (1 << depth, 3), dtype=np.uint8) if color == 3 else None
        trns = bytes(rng.integers(0, 256, int(rng.integers(1, (1 << depth) + 1))).astype(np.uint8)) if color == 3 and it % 2 else None
        filters = tuple(int(f) for f in rng.integers(0, 5, int(rng.integers(1, 6))))
        data = P.encode(s, depth, color, filters=filters, interlace=bool(it % 3 == 0), palette=pal, trns=trns,
                        idat_split=int(rng.integers(0, 60)), level=int(rng.integers(0, 10)))
        ref = P.decode(data)
        got = api.png_decode(data)
        assert got.dtype == ref.dtype and np.array_equal(got, ref), (it, color, depth, h, w, filters)


def test_kitti_size_batch_decode_into_caller_buffer():
    """The driver's case: a batch of KITTI-size 8-bit grey frames decoded by several host threads straight into one caller buffer
    (in production: the pinned staging block of the upload), equal to the images the files were written from."""
    from orbslam2_amd import synth
    imgs = [synth.stereo_pair(1241, 376, seed=40 + i)[0] for i in range(3)]
    files = []
    for i, im in enumerate(imgs):
        rows = b"".join(bytes([i % 5 if i < 2 else 2]) + (P._filter_row(i % 5 if i < 2 else 2, im[r], im[r - 1] if r else np.zeros_like(im[0]), 1).tobytes()) for r in range(376))
        z = zlib.compress(rows, 3)
        ch = lambda t, b: struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)
        files.append(b"\x89PNG\r\n\x1a\n" + ch(b"IHDR", struct.pack(">IIBBBBB", 1241, 376, 8, 0, 0, 0, 0)) + ch(b"IDAT", z) + ch(b"IEND", b""))
    out = np.full((6, 376, 1241), 7, np.uint8)
    got = api.png_decode_batch(files * 2, 1241, 376, threads=4, out=out)
    for i in range(6):
        assert np.array_equal(got[i], imgs[i % 3]), i
    with pytest.raises(api.OrbfeError):  # a frame of another geometry in the stream is an error, not a silent resize
        api.png_decode_batch(files + [_file("pil_gray8")], 1241, 376, threads=2)


def test_corrupt_files_are_refused_like_imread_returning_empty():
    good = _file("pil_gray8")
    bad_sig = b"\x89PNX" + good[4:]
    flipped = bytearray(good); flipped[len(good) // 2] ^= 0x40  # inside IDAT: CRC mismatch
    for data in (b"", good[:20], bad_sig, bytes(flipped), good[:-12]):  # empty, truncated, bad signature, bad CRC, no IEND
        with pytest.raises(api.OrbfeError):
            api.png_decode(data)
        with pytest.raises(ValueError):
            P.decode(data)
    # a stream that inflates to the wrong size (IHDR claims one more row)
    hdr = bytearray(good[:33]); struct.pack_into(">I", hdr, 20, struct.unpack_from(">I", hdr, 20)[0] + 1)
    struct.pack_into(">I", hdr, 29, zlib.crc32(bytes(hdr[12:29])) & 0xFFFFFFFF)
    with pytest.raises(api.OrbfeError):
        api.png_decode(bytes(hdr) + good[33:])
    # destination too small is a capacity error, not a write past the buffer
    import ctypes as C
    L = api.load()
    buf = np.frombuffer(good, np.uint8); small = np.zeros(100, np.uint8)
    L.orbfe_png_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t] + [C.POINTER(C.c_int)] * 4
    assert L.orbfe_png_decode(buf.ctypes.data, len(buf), small.ctypes.data, small.nbytes, 0, None, None, None, None) == api.ERR_CAPACITY
    assert not small.any()
