"""oracle/png_oracle.py -- TEST INFRASTRUCTURE: cv::imread(path, cv::IMREAD_UNCHANGED) for PNG files restated in Python
(zlib.decompress + numpy), independent of orbslam2_amd/csrc/orbfe_png.cpp.  Follows the PNG specification (ISO/IEC 15948:
chunk layout, filter types 0-4, Adam7) and the output conventions of OpenCV 4.5.5's PngDecoder (modules/imgcodecs/src/
grfmt_png.cpp; the reference calls it at Test/Replay/Stereo/stereo_kitti.cc:69-70): low-depth grey scaled to 8 bit, RGB ->
BGR, alpha kept (grey+alpha -> BGRA), palette expanded (tRNS -> BGRA), 16 bit kept in host byte order.
Pinned by Pillow: tests/golden/png/*.png decode to the same pixels with PIL.Image.open (tools/make_png_fixtures.py)."""
from __future__ import annotations

import struct
import zlib

import numpy as np

_X0, _Y0, _DX, _DY = (0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1), (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)


def _chunks(data: bytes):
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    p = 8
    while p + 12 <= len(data):
        (n,) = struct.unpack(">I", data[p:p + 4])
        typ = data[p + 4:p + 8]
        body = data[p + 8:p + 8 + n]
        if len(body) != n or p + 12 + n > len(data):
            raise ValueError("truncated chunk")
        (crc,) = struct.unpack(">I", data[p + 8 + n:p + 12 + n])
        if zlib.crc32(typ + body) & 0xFFFFFFFF != crc:
            raise ValueError("chunk CRC mismatch")
        yield typ, body
        p += 12 + n
        if typ == b"IEND":
            return
    raise ValueError("missing IEND")


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _unfilter(raw: memoryview, off: int, rows: int, rowbytes: int, bpp: int):
    out = np.zeros((rows, rowbytes), np.uint8)
    prev = np.zeros(rowbytes, np.int64)
    for r in range(rows):
        ft = raw[off]
        line = np.frombuffer(raw[off + 1:off + 1 + rowbytes], np.uint8).astype(np.int64)
        cur = np.zeros(rowbytes, np.int64)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft in (1, 3, 4):
            for i in range(rowbytes):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                pred = a if ft == 1 else ((a + b) >> 1 if ft == 3 else _paeth(a, b, c))
                cur[i] = (line[i] + pred) & 255
        else:
            raise ValueError("invalid filter type")
        out[r] = cur
        prev = cur
        off += rowbytes + 1
    return out, off


def _samples(rows: np.ndarray, depth: int, n: int) -> np.ndarray:
    """[rows, n] integer samples from unfiltered scanlines."""
    if depth == 8:
        return rows[:, :n].astype(np.uint16)
    if depth == 16:
        return (rows[:, 0:2 * n:2].astype(np.uint16) << 8) | rows[:, 1:2 * n:2].astype(np.uint16)
    bits = np.unpackbits(rows, axis=1)  # MSB first
    per = bits[:, : n * depth].reshape(rows.shape[0], n, depth)
    w = (1 << np.arange(depth - 1, -1, -1)).astype(np.uint16)
    return (per * w).sum(axis=2).astype(np.uint16)


def decode(data: bytes) -> np.ndarray:
    hdr = None
    idat = bytearray()
    palette = trns = None
    for typ, body in _chunks(data):
        if hdr is None:
            if typ != b"IHDR":
                raise ValueError("IHDR missing")
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"PLTE":
            palette = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif typ == b"tRNS":
            trns = body
        elif typ == b"IDAT":
            idat += body
        elif typ != b"IEND" and not (typ[0] & 0x20):
            raise ValueError("unknown critical chunk")
    w, h, depth, color, comp, flt, interlace = hdr
    nch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
    bits_pp = depth * nch
    bpp = max(1, bits_pp // 8)
    raw = memoryview(zlib.decompress(bytes(idat)))
    samp = np.zeros((h, w * nch), np.uint16)
    off = 0
    if interlace:
        for p in range(7):
            pw = (w + _DX[p] - 1 - _X0[p]) // _DX[p] if w > _X0[p] else 0
            ph = (h + _DY[p] - 1 - _Y0[p]) // _DY[p] if h > _Y0[p] else 0
            if pw == 0 or ph == 0:
                continue
            rows, off = _unfilter(raw, off, ph, (pw * bits_pp + 7) // 8, bpp)
            s = _samples(rows, depth, pw * nch).reshape(ph, pw, nch)
            samp.reshape(h, w, nch)[_Y0[p]::_DY[p], _X0[p]::_DX[p]] = s
    else:
        rows, off = _unfilter(raw, off, h, (w * bits_pp + 7) // 8, bpp)
        samp = _samples(rows, depth, w * nch)
    if off != len(raw):
        raise ValueError("IDAT stream does not decode to the image size")
    px = samp.reshape(h, w, nch)
    odt = np.uint16 if depth == 16 else np.uint8
    full = 65535 if depth == 16 else 255
    if color == 0:
        scale = {1: 255, 2: 85, 4: 17, 8: 1, 16: 1}[depth]
        return (px[:, :, 0] * scale).astype(odt)
    if color == 2:
        out = px[:, :, ::-1].astype(odt)
        if trns is not None and len(trns) == 6:
            key = np.array(struct.unpack(">HHH", trns), np.uint16)
            alpha = np.where((px == key).all(axis=2), 0, full).astype(odt)
            out = np.concatenate([out, alpha[:, :, None]], axis=2)
        return out
    if color == 3:
        idx = px[:, :, 0].astype(np.int64)
        idx[idx >= len(palette)] = 0
        out = palette[idx][:, :, ::-1]
        if trns is not None:
            a = np.full(256, 255, np.uint8); a[: len(trns)] = np.frombuffer(trns, np.uint8)[:256]
            out = np.concatenate([out, a[idx][:, :, None]], axis=2)
        return out.astype(np.uint8)
    if color == 4:
        return np.stack([px[:, :, 0]] * 3 + [px[:, :, 1]], axis=2).astype(odt)
    return px[:, :, [2, 1, 0, 3]].astype(odt)


# ---- a small PNG writer for fixtures the Pillow encoder cannot produce (chosen filter types, Adam7, 16-bit colour, split IDAT) ----
def _filter_row(ft, line, prev, bpp):
    line = line.astype(np.int64); prev = prev.astype(np.int64)
    out = np.zeros_like(line)
    for i in range(len(line)):
        a = line[i - bpp] if i >= bpp else 0
        b = prev[i]
        c = prev[i - bpp] if i >= bpp else 0
        pred = 0 if ft == 0 else a if ft == 1 else b if ft == 2 else (a + b) >> 1 if ft == 3 else _paeth(a, b, c)
        out[i] = (line[i] - pred) & 255
    return out.astype(np.uint8)


def _pack(samples: np.ndarray, depth: int) -> np.ndarray:
    """[rows, n] samples -> scanline bytes."""
    if depth == 8:
        return samples.astype(np.uint8)
    if depth == 16:
        s = samples.astype(np.uint16)
        return np.stack([(s >> 8).astype(np.uint8), (s & 255).astype(np.uint8)], axis=2).reshape(s.shape[0], -1)
    bits = ((samples[:, :, None].astype(np.uint16) >> np.arange(depth - 1, -1, -1)) & 1).astype(np.uint8).reshape(samples.shape[0], -1)
    return np.packbits(bits, axis=1)


def encode(samples: np.ndarray, depth: int, color: int, filters=(0,), interlace=False, palette=None, trns=None, idat_split=0, level=6) -> bytes:
    """samples: [h, w, nch] integers of `depth` bits (file order: R, G, B, A).  filters: filter type per row (cycled)."""
    h, w, nch = samples.shape
    bits_pp = depth * nch
    bpp = max(1, bits_pp // 8)
    stream = bytearray()
    passes = [(0, 0, 1, 1)] if not interlace else list(zip(_X0, _Y0, _DX, _DY))
    k = 0
    for x0, y0, dx, dy in passes:
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        rows = _pack(sub.reshape(sub.shape[0], -1), depth)
        prev = np.zeros(rows.shape[1], np.uint8)
        for r in range(rows.shape[0]):
            ft = filters[k % len(filters)]; k += 1
            stream.append(ft)
            stream += _filter_row(ft, rows[r], prev, bpp).tobytes()
            prev = rows[r]
    z = zlib.compress(bytes(stream), level)

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    out += chunk(b"tEXt", b"Comment\x00orbslam2_amd fixture")  # an ancillary chunk decoders must skip
    if idat_split and len(z) > idat_split:
        for i in range(0, len(z), idat_split):
            out += chunk(b"IDAT", z[i:i + idat_split])
    else:
        out += chunk(b"IDAT", z)
    return out + chunk(b"IEND", b"")
