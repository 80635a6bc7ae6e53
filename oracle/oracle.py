"""ctypes loader for the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
PARITY UNPINNED: see oracle/orb_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class Keypoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32), ("class_id", C.c_int32)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28 and C.sizeof(Keypoint) == 28


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("patch_size", C.c_int32),
                ("half_patch_size", C.c_int32), ("edge_threshold", C.c_int32)]


def build(target: str = "liborb_oracle.so") -> str:
    path = os.path.join(HERE, target)
    subprocess.run(["make", "-C", HERE, target], check=True, stdout=subprocess.DEVNULL)
    return path


_libs = {}


def lib(target: str = "liborb_oracle.so"):
    if target in _libs:
        return _libs[target]
    path = os.path.join(HERE, target)
    srcs = [os.path.join(HERE, f) for f in ("orb_oracle.c", "orb_oracle_match.c", "orb_oracle.h")]
    if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs if os.path.exists(s)):
        build(target)
    L = C.CDLL(path)
    u8p, i32p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_float)
    L.orc_cv_round_f.restype = C.c_int; L.orc_cv_round_f.argtypes = [C.c_float]
    L.orc_cv_round_d.restype = C.c_int; L.orc_cv_round_d.argtypes = [C.c_double]
    L.orc_fast_atan2.restype = C.c_float; L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
    L.orc_sincos_det.restype = None; L.orc_sincos_det.argtypes = [C.c_float, f32p, f32p]
    L.orc_border_reflect101.restype = C.c_int; L.orc_border_reflect101.argtypes = [C.c_int, C.c_int]
    L.orc_hamming256.restype = C.c_int; L.orc_hamming256.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_bit_pattern.restype = i32p
    L.orc_gaussian_taps_q8.restype = None; L.orc_gaussian_taps_q8.argtypes = [C.c_int, C.c_double, i32p]
    L.orc_resize_linear_u8.restype = None
    L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_size_t]
    L.orc_gaussian7_u8.restype = None
    L.orc_gaussian7_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]
    L.orc_fast9_16.restype = C.c_int
    L.orc_fast9_16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_fast_corner_score.restype = C.c_int; L.orc_fast_corner_score.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    L.orc_fast_score_closed_form.restype = C.c_int; L.orc_fast_score_closed_form.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    L.orc_extractor_create.restype = C.c_void_p; L.orc_extractor_create.argtypes = [C.POINTER(Params)]
    L.orc_extractor_destroy.restype = None; L.orc_extractor_destroy.argtypes = [C.c_void_p]
    L.orc_extractor_nlevels.restype = C.c_int; L.orc_extractor_nlevels.argtypes = [C.c_void_p]
    for name in ("scale_factors", "inv_scale_factors", "sigma2", "inv_sigma2"):
        f = getattr(L, "orc_extractor_" + name); f.restype = f32p; f.argtypes = [C.c_void_p]
    for name in ("features_per_level", "umax"):
        f = getattr(L, "orc_extractor_" + name); f.restype = i32p; f.argtypes = [C.c_void_p]
    L.orc_level_size.restype = None
    L.orc_level_size.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_extract.restype = C.c_int
    L.orc_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_pyramid_level.restype = C.c_void_p
    L.orc_pyramid_level.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.orc_level_candidates.restype = C.c_int
    L.orc_level_candidates.argtypes = [C.c_void_p, C.c_int, C.POINTER(i32p), C.POINTER(i32p), C.POINTER(i32p)]
    L.orc_distribute_octtree.restype = C.c_int
    L.orc_distribute_octtree.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.orc_ic_angle.restype = C.c_float
    L.orc_ic_angle.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.orc_orb_descriptor.restype = None
    L.orc_orb_descriptor.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_void_p]
    L.orc_stereo_matches.restype = C.c_int
    L.orc_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_stereo_from_rgbd.restype = None
    L.orc_stereo_from_rgbd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_float, C.c_void_p, C.c_void_p]
    _bind_match(L)
    _libs[target] = L
    return L


def _bind_match(L):
    """Matcher entry points (orb_oracle_match.c); bound lazily as they are added."""
    pass


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Extractor:
    """Mirror of ORB_SLAM2::ORBextractor (include/ORBextractor.h:45-112) over the C oracle."""

    def __init__(self, nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th_fast=20, min_th_fast=7,
                 patch_size=31, half_patch_size=15, edge_threshold=19, target="liborb_oracle.so"):
        self.L = lib(target)
        self.params = Params(nfeatures, scale_factor, nlevels, ini_th_fast, min_th_fast, patch_size,
                             half_patch_size, edge_threshold)
        self.h = self.L.orc_extractor_create(C.byref(self.params))
        if not self.h:
            raise ValueError("bad extractor parameters")
        self.nlevels = nlevels
        self.nfeatures = nfeatures

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_extractor_destroy(self.h)
            self.h = None

    def _farr(self, fn, n):
        p = fn(self.h)
        return np.array([p[i] for i in range(n)], dtype=np.float32)

    def scale_factors(self):
        return self._farr(self.L.orc_extractor_scale_factors, self.nlevels)

    def inv_scale_factors(self):
        return self._farr(self.L.orc_extractor_inv_scale_factors, self.nlevels)

    def sigma2(self):
        return self._farr(self.L.orc_extractor_sigma2, self.nlevels)

    def inv_sigma2(self):
        return self._farr(self.L.orc_extractor_inv_sigma2, self.nlevels)

    def features_per_level(self):
        p = self.L.orc_extractor_features_per_level(self.h)
        return np.array([p[i] for i in range(self.nlevels)], dtype=np.int32)

    def umax(self):
        p = self.L.orc_extractor_umax(self.h)
        return np.array([p[i] for i in range(self.params.half_patch_size + 1)], dtype=np.int32)

    def level_size(self, w, h, level):
        lw, lh = C.c_int(), C.c_int()
        self.L.orc_level_size(self.h, w, h, level, C.byref(lw), C.byref(lh))
        return lw.value, lh.value

    def extract(self, img: np.ndarray):
        assert img.dtype == np.uint8 and img.ndim == 2
        img = np.ascontiguousarray(img)
        cap = self.nfeatures + 64 * self.nlevels
        kps = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = self.L.orc_extract(self.h, _ptr(img), img.shape[1], img.shape[0], img.strides[0], _ptr(kps), _ptr(desc), cap)
        if n < 0:
            raise RuntimeError("orc_extract failed")
        assert n <= cap
        return kps[:n].copy(), desc[:n].copy()

    def pyramid_level(self, level):
        w, h, s = C.c_int(), C.c_int(), C.c_size_t()
        p = self.L.orc_pyramid_level(self.h, level, C.byref(w), C.byref(h), C.byref(s))
        if not p:
            return None
        buf = (C.c_uint8 * (s.value * h.value)).from_address(p)
        return np.frombuffer(buf, dtype=np.uint8).reshape(h.value, s.value)[:, : w.value].copy()

    def level_candidates(self, level):
        i32p = C.POINTER(C.c_int32)
        xs, ys, ss = i32p(), i32p(), i32p()
        n = self.L.orc_level_candidates(self.h, level, C.byref(xs), C.byref(ys), C.byref(ss))
        if n <= 0:
            z = np.zeros(0, np.int32)
            return z, z.copy(), z.copy()
        f = lambda p: np.ctypeslib.as_array(p, shape=(n,)).copy()
        return f(xs), f(ys), f(ss)


def resize_linear(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_ptr(src), src.shape[1], src.shape[0], src.strides[0], _ptr(dst), dw, dh, dst.strides[0])
    return dst


def gaussian7(src: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(src)
    dst = np.zeros_like(src)
    lib().orc_gaussian7_u8(_ptr(src), src.shape[1], src.shape[0], src.strides[0], _ptr(dst), dst.strides[0])
    return dst


def fast9_16(img: np.ndarray, threshold: int, nonmax: bool = True):
    img = np.ascontiguousarray(img)
    cap = img.size
    xs, ys, ss = (np.zeros(cap, np.int32) for _ in range(3))
    n = lib().orc_fast9_16(_ptr(img), img.shape[1], img.shape[0], img.strides[0], threshold, int(nonmax), _ptr(xs), _ptr(ys), _ptr(ss), cap)
    return xs[:n].copy(), ys[:n].copy(), ss[:n].copy()


def distribute_octtree(xs, ys, scores, min_x, max_x, min_y, max_y, n_features):
    xs = np.ascontiguousarray(xs, np.int32); ys = np.ascontiguousarray(ys, np.int32); scores = np.ascontiguousarray(scores, np.int32)
    cap = max(len(xs), 1) + 16
    out = np.zeros(cap, np.int32)
    n = lib().orc_distribute_octtree(_ptr(xs), _ptr(ys), _ptr(scores), len(xs), min_x, max_x, min_y, max_y, n_features, _ptr(out), cap)
    return out[:n].copy()


def hamming256(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_hamming256(_ptr(a), _ptr(b))


def stereo_matches(exL: Extractor, exR: Extractor, kL, dL, kR, dR, bf: float, fx: float, debug=False):
    kL = np.ascontiguousarray(kL); kR = np.ascontiguousarray(kR)
    dL = np.ascontiguousarray(dL); dR = np.ascontiguousarray(dR)
    n = len(kL)
    ur = np.zeros(n, np.float32); dp = np.zeros(n, np.float32)
    bi = np.zeros(n, np.int32); bs = np.zeros(n, np.int32)
    m = exL.L.orc_stereo_matches(exL.h, exR.h, _ptr(kL), _ptr(dL), n, _ptr(kR), _ptr(dR), len(kR), bf, fx,
                                 _ptr(ur), _ptr(dp), _ptr(bi), _ptr(bs))
    if debug:
        return ur, dp, m, bi, bs
    return ur, dp, m


def stereo_from_rgbd(k, k_un, depth: np.ndarray, bf: float):
    k = np.ascontiguousarray(k); k_un = np.ascontiguousarray(k_un)
    depth = np.ascontiguousarray(depth, np.float32)
    n = len(k)
    ur = np.zeros(n, np.float32); dp = np.zeros(n, np.float32)
    lib().orc_stereo_from_rgbd(_ptr(k), _ptr(k_un), n, _ptr(depth), depth.shape[1], depth.shape[0], depth.strides[0] // 4, bf, _ptr(ur), _ptr(dp))
    return ur, dp
