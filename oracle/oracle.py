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
    srcs = [os.path.join(HERE, f) for f in ("orb_oracle.c", "orb_oracle_match.c", "orb_oracle_bow.c", "orb_oracle_pose.c", "orb_oracle.h")]
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
    L.orc_extractor_set_pattern.restype = None; L.orc_extractor_set_pattern.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_extractor_pattern.restype = i32p; L.orc_extractor_pattern.argtypes = [C.c_void_p]
    L.orc_stereo_matches.restype = C.c_int
    L.orc_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_stereo_from_rgbd.restype = None
    L.orc_stereo_from_rgbd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_float, C.c_void_p, C.c_void_p]
    _bind_match(L)
    _libs[target] = L
    return L


class Camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float), ("mb", C.c_float)]


class TrackPoint(C.Structure):
    _fields_ = [("in_view", C.c_int32), ("proj_x", C.c_float), ("proj_y", C.c_float), ("proj_xr", C.c_float),
                ("level", C.c_int32), ("view_cos", C.c_float)]


TP_DTYPE = np.dtype([("in_view", "<i4"), ("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"), ("level", "<i4"), ("view_cos", "<f4")])


def _bind_match(L):
    """Matcher entry points (orb_oracle_match.c)."""
    vp = C.c_void_p
    L.orc_grid_create.restype = vp; L.orc_grid_create.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
    L.orc_grid_destroy.restype = None; L.orc_grid_destroy.argtypes = [vp]
    L.orc_grid_as_keyframe.restype = None; L.orc_grid_as_keyframe.argtypes = [vp]
    L.orc_features_in_area.restype = C.c_int
    L.orc_features_in_area.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, vp, C.c_int]
    L.orc_three_maxima.restype = None
    L.orc_three_maxima.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_log_det.restype = C.c_float; L.orc_log_det.argtypes = [C.c_float]
    L.orc_predict_scale.restype = C.c_int; L.orc_predict_scale.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int]
    L.orc_search_by_projection_last.restype = C.c_int
    L.orc_search_by_projection_last.argtypes = [vp, vp, vp, vp, C.POINTER(Camera), vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp,
                                                C.c_float, C.c_int, C.c_int, vp]
    L.orc_is_in_frustum.restype = C.c_int
    L.orc_is_in_frustum.argtypes = [vp, C.POINTER(Camera), C.c_float, C.c_float, C.c_float, C.c_float, vp, vp, C.c_float, C.c_float,
                                    C.c_float, C.c_float, C.c_float, C.c_int, C.POINTER(TrackPoint)]
    L.orc_search_by_projection_points.restype = C.c_int
    L.orc_search_by_projection_points.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_float, C.c_float, vp]
    L.orc_search_by_projection_kf.restype = C.c_int
    L.orc_search_by_projection_kf.argtypes = [vp, vp, vp, C.POINTER(Camera), vp, C.c_float, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp,
                                              vp, C.c_float, C.c_int, C.c_int, vp]
    L.orc_search_for_initialization.restype = C.c_int
    L.orc_search_for_initialization.argtypes = [vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_float, C.c_int, vp]


class Grid:
    """Frame grid (AssignFeaturesToGrid) over mvKeysUn; keeps the arrays alive."""

    def __init__(self, keys_un, min_x, max_x, min_y, max_y, keyframe=False):
        self.L = lib()
        self.keys = np.ascontiguousarray(keys_un)
        self.bounds = (float(min_x), float(max_x), float(min_y), float(max_y))
        self.h = self.L.orc_grid_create(_ptr(self.keys), len(self.keys), *self.bounds)
        if keyframe:  # a KeyFrame's copy of the frame's grid: integer bounds for windows / IsInImage (src/KeyFrame.cc:563-607)
            self.L.orc_grid_as_keyframe(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_grid_destroy(self.h)
            self.h = None

    def features_in_area(self, x, y, r, min_level=-1, max_level=-1):
        out = np.zeros(max(len(self.keys), 1), np.int32)
        n = self.L.orc_features_in_area(self.h, x, y, r, min_level, max_level, _ptr(out), len(out))
        return out[:n].copy()


def three_maxima(sizes):
    s = np.ascontiguousarray(sizes, np.int32)
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    lib().orc_three_maxima(_ptr(s), len(s), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def _opt(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype)


def search_by_projection_last(grid, u_right, desc_cur, scale_factors, cam, Tcw_cur, Tcw_last, last_pos, last_desc, last_valid,
                              last_obs, last_octave, last_angle, cur_has_obs, th, mono, check_ori):
    ur = _opt(u_right, np.float32); dc = np.ascontiguousarray(desc_cur, np.uint8); sf = np.ascontiguousarray(scale_factors, np.float32)
    tc = np.ascontiguousarray(Tcw_cur, np.float32); tl = np.ascontiguousarray(Tcw_last, np.float32)
    lp = np.ascontiguousarray(last_pos, np.float32); ld = np.ascontiguousarray(last_desc, np.uint8)
    lv = np.ascontiguousarray(last_valid, np.int32); lo = np.ascontiguousarray(last_obs, np.int32)
    loc = np.ascontiguousarray(last_octave, np.int32); la = np.ascontiguousarray(last_angle, np.float32)
    ho = _opt(cur_has_obs, np.uint8)
    out = np.zeros(max(len(grid.keys), 1), np.int32)
    n = lib().orc_search_by_projection_last(grid.h, None if ur is None else _ptr(ur), _ptr(dc), _ptr(sf), C.byref(cam), _ptr(tc), _ptr(tl),
                                            len(lv), _ptr(lp), _ptr(ld), _ptr(lv), _ptr(lo), _ptr(loc), _ptr(la),
                                            None if ho is None else _ptr(ho), th, int(mono), int(check_ori), _ptr(out))
    return out[: len(grid.keys)].copy(), n


def is_in_frustum(Tcw, cam, bounds, pos, normal, max_distance, min_distance, viewing_cos_limit, log_scale_factor, n_levels):
    tc = np.ascontiguousarray(Tcw, np.float32)
    pos = np.ascontiguousarray(pos, np.float32); normal = np.ascontiguousarray(normal, np.float32)
    out = np.zeros(len(pos), TP_DTYPE)
    tp = TrackPoint()
    for i in range(len(pos)):
        lib().orc_is_in_frustum(_ptr(tc), C.byref(cam), bounds[0], bounds[1], bounds[2], bounds[3], _ptr(pos[i]), _ptr(normal[i]),
                                np.float32(1.2) * np.float32(max_distance[i]), np.float32(0.8) * np.float32(min_distance[i]),
                                float(max_distance[i]), viewing_cos_limit, log_scale_factor, n_levels, C.byref(tp))
        out[i] = (tp.in_view, tp.proj_x, tp.proj_y, tp.proj_xr, tp.level, tp.view_cos)
    return out


def search_by_projection_points(grid, u_right, desc_cur, scale_factors, pts, pt_desc, pt_obs, cur_has_obs, th, nnratio):
    ur = _opt(u_right, np.float32); dc = np.ascontiguousarray(desc_cur, np.uint8); sf = np.ascontiguousarray(scale_factors, np.float32)
    pts = np.ascontiguousarray(pts, TP_DTYPE); pd = np.ascontiguousarray(pt_desc, np.uint8); po = np.ascontiguousarray(pt_obs, np.int32)
    ho = _opt(cur_has_obs, np.uint8)
    out = np.zeros(max(len(grid.keys), 1), np.int32)
    n = lib().orc_search_by_projection_points(grid.h, None if ur is None else _ptr(ur), _ptr(dc), _ptr(sf), len(pts), _ptr(pts), _ptr(pd),
                                              _ptr(po), None if ho is None else _ptr(ho), th, nnratio, _ptr(out))
    return out[: len(grid.keys)].copy(), n


def search_by_projection_kf(grid, desc_cur, scale_factors, cam, Tcw_cur, log_scale_factor, n_levels, kf_pos, kf_desc, kf_valid, kf_angle,
                            kf_max_distance, kf_min_distance, cur_has_point, th, orb_dist, check_ori):
    dc = np.ascontiguousarray(desc_cur, np.uint8); sf = np.ascontiguousarray(scale_factors, np.float32)
    tc = np.ascontiguousarray(Tcw_cur, np.float32); kp = np.ascontiguousarray(kf_pos, np.float32)
    kd = np.ascontiguousarray(kf_desc, np.uint8); kv = np.ascontiguousarray(kf_valid, np.int32); ka = np.ascontiguousarray(kf_angle, np.float32)
    kmx = np.ascontiguousarray(kf_max_distance, np.float32); kmn = np.ascontiguousarray(kf_min_distance, np.float32)
    hp = _opt(cur_has_point, np.uint8)
    out = np.zeros(max(len(grid.keys), 1), np.int32)
    n = lib().orc_search_by_projection_kf(grid.h, _ptr(dc), _ptr(sf), C.byref(cam), _ptr(tc), log_scale_factor, n_levels, len(kv), _ptr(kp),
                                          _ptr(kd), _ptr(kv), _ptr(ka), _ptr(kmx), _ptr(kmn), None if hp is None else _ptr(hp),
                                          th, orb_dist, int(check_ori), _ptr(out))
    return out[: len(grid.keys)].copy(), n


def fuse(grid, u_right_kf, desc_kf, scale_factors, inv_level_sigma2, cam, Tcw, log_scale_factor, n_levels, pos, normal,
         max_distance, min_distance, pt_desc, pt_valid, th):
    """Search part of ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th): (best keypoint per map point or -1, count)."""
    L = lib()
    L.orc_fuse.restype = C.c_int
    L.orc_fuse.argtypes = [C.c_void_p] * 5 + [C.POINTER(Camera), C.c_void_p, C.c_float, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_void_p]
    ur = _opt(u_right_kf, np.float32); d = np.ascontiguousarray(desc_kf, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32); inv = np.ascontiguousarray(inv_level_sigma2, np.float32)
    t = np.ascontiguousarray(Tcw, np.float32); p = np.ascontiguousarray(pos, np.float32); nrm = np.ascontiguousarray(normal, np.float32)
    mx = np.ascontiguousarray(max_distance, np.float32); mn = np.ascontiguousarray(min_distance, np.float32)
    pd = np.ascontiguousarray(pt_desc, np.uint8); ok = np.ascontiguousarray(pt_valid, np.int32)
    out = np.zeros(max(len(ok), 1), np.int32)
    n = L.orc_fuse(grid.h, None if ur is None else _ptr(ur), _ptr(d), _ptr(sf), _ptr(inv), C.byref(cam), _ptr(t), log_scale_factor, n_levels,
                   len(ok), _ptr(p), _ptr(nrm), _ptr(mx), _ptr(mn), _ptr(pd), _ptr(ok), th, _ptr(out))
    return out[: len(ok)].copy(), n


def sim3_projection(mode, grid, desc_kf, scale_factors, cam, Scw, log_scale_factor, n_levels, pos, normal, max_distance, min_distance,
                    pt_desc, pt_valid, kf_matched, th):
    """mode 0: SearchByProjection(KeyFrame*, Scw, ...); mode 1: search part of Fuse(KeyFrame*, Scw, ...)."""
    L = lib()
    L.orc_search_by_sim3_projection.restype = C.c_int
    L.orc_search_by_sim3_projection.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.POINTER(Camera), C.c_void_p, C.c_float, C.c_int, C.c_int] + \
        [C.c_void_p] * 7 + [C.c_float, C.c_void_p]
    d = np.ascontiguousarray(desc_kf, np.uint8); sf = np.ascontiguousarray(scale_factors, np.float32)
    t = np.ascontiguousarray(Scw, np.float32); p = np.ascontiguousarray(pos, np.float32); nrm = np.ascontiguousarray(normal, np.float32)
    mx = np.ascontiguousarray(max_distance, np.float32); mn = np.ascontiguousarray(min_distance, np.float32)
    pd = np.ascontiguousarray(pt_desc, np.uint8); ok = np.ascontiguousarray(pt_valid, np.int32)
    km = _opt(kf_matched, np.uint8)
    out = np.zeros(max(len(ok), 1), np.int32)
    n = L.orc_search_by_sim3_projection(mode, grid.h, _ptr(d), _ptr(sf), C.byref(cam), _ptr(t), log_scale_factor, n_levels, len(ok), _ptr(p), _ptr(nrm),
                                        _ptr(mx), _ptr(mn), _ptr(pd), _ptr(ok), None if km is None else _ptr(km), th, _ptr(out))
    return out[: len(ok)].copy(), n


def search_by_sim3(grid1, desc_kf1, T1w, pts1, grid2, desc_kf2, T2w, pts2, scale_factors, cam, log_scale_factor, n_levels, s12, R12, t12, th):
    """ORBmatcher::SearchBySim3; pts = (pos, max_distance, min_distance, desc, valid) per keypoint slot.  (match12, count)."""
    L = lib()
    L.orc_search_by_sim3.restype = C.c_int
    L.orc_search_by_sim3.argtypes = [C.c_void_p] * 16 + [C.c_void_p, C.POINTER(Camera), C.c_float, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]

    def prep(dkf, T, pts):
        pos, mx, mn, d, ok = pts
        return [np.ascontiguousarray(dkf, np.uint8), np.ascontiguousarray(T, np.float32), np.ascontiguousarray(pos, np.float32),
                np.ascontiguousarray(mx, np.float32), np.ascontiguousarray(mn, np.float32), np.ascontiguousarray(d, np.uint8),
                np.ascontiguousarray(ok, np.int32)]
    a, b = prep(desc_kf1, T1w, pts1), prep(desc_kf2, T2w, pts2)
    sf = np.ascontiguousarray(scale_factors, np.float32); R = np.ascontiguousarray(R12, np.float32); t = np.ascontiguousarray(t12, np.float32)
    out = np.zeros(max(len(a[6]), 1), np.int32)
    n = L.orc_search_by_sim3(grid1.h, *[_ptr(x) for x in a], grid2.h, *[_ptr(x) for x in b], _ptr(sf), C.byref(cam), log_scale_factor, n_levels,
                             float(s12), _ptr(R), _ptr(t), th, _ptr(out))
    return out[: len(a[6])].copy(), n


def search_for_initialization(keys1, desc1, grid2, desc2, prev_matched, window_size, nnratio, check_ori):
    k1 = np.ascontiguousarray(keys1); d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
    pm = np.ascontiguousarray(prev_matched, np.float32).copy()
    out = np.zeros(max(len(k1), 1), np.int32)
    n = lib().orc_search_for_initialization(_ptr(k1), _ptr(d1), len(k1), grid2.h, _ptr(d2), _ptr(pm), window_size, nnratio, int(check_ori), _ptr(out))
    return out[: len(k1)].copy(), pm, n


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

    def set_pattern(self, pattern):
        """Replace the extractor's copy of the 256 x (x0, y0, x1, y1) rBRIEF tests (src/ORBextractor.cc:442-444)."""
        pat = np.ascontiguousarray(np.asarray(pattern, dtype=np.int32).reshape(1024))
        self.L.orc_extractor_set_pattern(self.h, _ptr(pat))

    def pattern(self):
        p = self.L.orc_extractor_pattern(self.h)
        return np.array([p[i] for i in range(1024)], dtype=np.int32).reshape(256, 4)

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


def pose_optimization(Tcw, keys_un, u_right, has_point, Xw, inv_level_sigma2, fx, fy, cx, cy, bf, outlier=None):
    """Optimizer::PoseOptimization (src/Optimizer.cc:283-495).  Returns (Tcw_out 4x4 float32, outlier uint8[N], n_inliers)."""
    L = lib()
    L.orc_pose_optimization.restype = C.c_int
    L.orc_pose_optimization.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_float] * 5 + [C.c_void_p]
    T = np.ascontiguousarray(Tcw, np.float32).reshape(4, 4).copy()
    k = np.ascontiguousarray(keys_un); ur = np.ascontiguousarray(u_right, np.float32)
    hp = np.ascontiguousarray(has_point, np.uint8); X = np.ascontiguousarray(Xw, np.float32)
    s2 = np.ascontiguousarray(inv_level_sigma2, np.float32)
    out = np.zeros(max(len(k), 1), np.uint8) if outlier is None else np.ascontiguousarray(outlier, np.uint8).copy()
    n = L.orc_pose_optimization(_ptr(T), len(k), _ptr(k), _ptr(ur), _ptr(hp), _ptr(X), _ptr(s2), fx, fy, cx, cy, bf, _ptr(out))
    return T, out[: len(k)].copy(), n


def cvt_gray(img, rgb=True, legacy14=False):
    """cv::cvtColor(img, COLOR_{RGB,BGR}[A]2GRAY) for uint8 HxWx{3,4} (src/Tracking.cc:269-294)."""
    a = np.ascontiguousarray(img, np.uint8)
    h, w, cn = a.shape
    out = np.zeros((h, w), np.uint8)
    L = lib()
    L.orc_cvt_gray.restype = None
    L.orc_cvt_gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    L.orc_cvt_gray(_ptr(a), w, h, a.strides[0], cn, int(rgb), int(legacy14), _ptr(out), out.strides[0])
    return out


def undistort_points(xy, fx, fy, cx, cy, dist):
    """cv::undistortPoints(xy, K, D, Mat(), K) (src/Frame.cc:420): float32 [n,2] in and out."""
    a = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    d = np.ascontiguousarray(dist, np.float32)
    out = np.zeros_like(a)
    L = lib()
    L.orc_undistort_points.restype = None
    L.orc_undistort_points.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_void_p]
    L.orc_undistort_points(_ptr(a), len(a), fx, fy, cx, cy, _ptr(d), len(d), _ptr(out))
    return out


def image_bounds(cols, rows, fx, fy, cx, cy, dist):
    """Frame::ComputeImageBounds: (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    d = np.ascontiguousarray(dist, np.float32)
    out = np.zeros(4, np.float32)
    L = lib()
    L.orc_image_bounds.restype = None
    L.orc_image_bounds.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_void_p]
    L.orc_image_bounds(cols, rows, fx, fy, cx, cy, _ptr(d), len(d), _ptr(out))
    return out


def remap_bilinear(src, mapx, mapy):
    """cv::remap(src, map1, map2, INTER_LINEAR) for uint8 HxW, float32 maps (EuRoC rectification)."""
    a = np.ascontiguousarray(src, np.uint8); mx = np.ascontiguousarray(mapx, np.float32); my = np.ascontiguousarray(mapy, np.float32)
    assert mx.shape == my.shape
    dh, dw = mx.shape
    out = np.zeros((dh, dw), np.uint8)
    L = lib()
    L.orc_remap_bilinear.restype = None
    L.orc_remap_bilinear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    L.orc_remap_bilinear(_ptr(a), a.shape[1], a.shape[0], a.strides[0], _ptr(mx), _ptr(my), dw, dh, _ptr(out), out.strides[0])
    return out
