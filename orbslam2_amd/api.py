"""ctypes binding of the C ABI (include/orbfe.h) -- plumbing for tests and bench.py.

The product is liborbfe.so (HIP, gfx950).  This module never computes anything
itself and has no CPU fallback: if the library is missing or no GPU is present
the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liborbfe.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_CAPACITY, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5

EXPORTS = [
    "orbfe_build_id", "orbfe_set_pattern", "orbfe_get_pattern", "orbfe_blur_ride_from", "orbfe_set_input_retained",
    "orbfe_abi_version", "orbfe_last_error", "orbfe_create", "orbfe_destroy", "orbfe_levels",
    "orbfe_keypoint_capacity", "orbfe_get_tables", "orbfe_level_size", "orbfe_extract",
    "orbfe_stereo_frame", "orbfe_rgbd_frame", "orbfe_rgbd_frame_u16", "orbfe_fetch_pyramid", "orbfe_enqueue_extract",
    "orbfe_enqueue_stereo", "orbfe_synchronize", "orbfe_fetch_image", "orbfe_fetch_counts",
    "orbfe_device_buffers", "orbfe_fetch_candidates", "orbfe_hamming_matrix",
    "orbfe_set_profiling", "orbfe_stage_name", "orbfe_stage_times", "orbfe_set_streams", "orbfe_quadtree_kernel",
    "orbfe_features_in_area", "orbfe_features_in_area_batch", "orbfe_three_maxima", "orbfe_search_by_projection_last", "orbfe_is_in_frustum",
    "orbfe_search_by_projection_points", "orbfe_search_by_projection_kf", "orbfe_search_for_initialization",
    "orbfe_vocab_load", "orbfe_bow_transform", "orbfe_bow_maps", "orbfe_search_by_bow", "orbfe_search_by_bow_kf",  # bound in orbslam2_amd/bow.py
    "orbfe_search_for_triangulation", "orbfe_fuse", "orbfe_search_by_projection_sim3", "orbfe_fuse_sim3", "orbfe_search_by_sim3", "orbfe_kfdb_clear", "orbfe_kfdb_add", "orbfe_kfdb_erase", "orbfe_kfdb_size", "orbfe_kfdb_score", "orbfe_detect_reloc_candidates", "orbfe_detect_loop_candidates",
    "orbfe_pose_optimization", "orbfe_pose_optimization_batch", "orbfe_enqueue_pose_optimization", "orbfe_set_input_format", "orbfe_fetch_batch_async", "orbfe_set_rectification", "orbfe_set_distortion", "orbfe_undistort_keypoints", "orbfe_fetch_keys_un", "orbfe_image_bounds",
    "orbfe_png_last_error", "orbfe_png_info", "orbfe_png_decode", "orbfe_png_decode_batch",
    "orbfe_get_camera", "orbfe_assign_features_to_grid", "orbfe_set_profiling_interval", "orbfe_stereo_batch", "orbfe_device_count", "orbfe_vocab_bytes",
    "orbfe_get_packed_layout", "orbfe_fetch_batch_packed", "orbfe_expand_packed", "orbfe_enqueue_rgbd", "orbfe_stereo_batch_packed",
]
NUM_STAGES = 8
STAGE_NAMES = ["ingest", "pyramid", "blur", "fast", "octree", "describe", "stereo_match", "stereo_median"]  # orbfe_stage_name()


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("patch_size", C.c_int32),
                ("half_patch_size", C.c_int32), ("edge_threshold", C.c_int32),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("bf", C.c_float), ("device", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("max_images", C.c_int32)]


class FrameView(C.Structure):
    _fields_ = [("n", C.c_int32), ("keys_un", C.c_void_p), ("u_right", C.c_void_p), ("descriptors", C.c_void_p),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
                ("device_slot_plus1", C.c_int32), ("keyframe", C.c_int32)]


class PackedLayout(C.Structure):
    """orbfe_packed_layout (include/orbfe.h): byte offsets of the arrays inside a packed result block."""
    _fields_ = [("n_images_out", C.c_int32), ("capacity", C.c_int32), ("nlevels", C.c_int32), ("n_pairs", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32),
                ("counts_off", C.c_size_t), ("level_counts_off", C.c_size_t), ("xy_off", C.c_size_t), ("angle_off", C.c_size_t),
                ("response_off", C.c_size_t), ("desc_off", C.c_size_t), ("u_right_off", C.c_size_t), ("depth_off", C.c_size_t), ("bytes", C.c_size_t)]


PACK_STEREO, PACK_LEFT_ONLY, PACK_DIRECT = 1, 2, 4

TP_DTYPE = np.dtype([("in_view", "<i4"), ("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"), ("level", "<i4"), ("view_cos", "<f4")])


class OrbfeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("orbfe error %d: %s" % (code, msg))
        self.code = code


_lib = None


def build_id() -> str:
    """orbfe_build_id(): sha256 over the sources and flags liborbfe.so was built from."""
    return load().orbfe_build_id().decode()


def load():
    """Load liborbfe.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("liborbfe.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "or `make -C orbslam2_amd/csrc`")
    # PyTorch-ROCm bundles its own libamdhip64.so.7; two HIP runtimes in one process make the second one
    # see no GPU.  Importing torch first makes its copy the process-wide runtime that liborbfe.so binds to.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.orbfe_abi_version.restype = C.c_int
    L.orbfe_build_id.restype = C.c_char_p
    L.orbfe_set_pattern.restype = C.c_int; L.orbfe_set_pattern.argtypes = [vp, vp]
    L.orbfe_get_pattern.restype = C.c_int; L.orbfe_get_pattern.argtypes = [vp, vp]
    L.orbfe_blur_ride_from.restype = C.c_int; L.orbfe_blur_ride_from.argtypes = [vp, C.c_int]
    L.orbfe_set_input_retained.restype = C.c_int; L.orbfe_set_input_retained.argtypes = [vp, C.c_int]
    L.orbfe_last_error.restype = C.c_char_p; L.orbfe_last_error.argtypes = [vp]
    L.orbfe_create.restype = C.c_int; L.orbfe_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.orbfe_destroy.restype = None; L.orbfe_destroy.argtypes = [vp]
    L.orbfe_levels.restype = C.c_int; L.orbfe_levels.argtypes = [vp]
    L.orbfe_keypoint_capacity.restype = C.c_int; L.orbfe_keypoint_capacity.argtypes = [vp]
    L.orbfe_get_tables.restype = C.c_int; L.orbfe_get_tables.argtypes = [vp] + [vp] * 6
    L.orbfe_level_size.restype = C.c_int; L.orbfe_level_size.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orbfe_extract.restype = C.c_int
    L.orbfe_extract.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.orbfe_stereo_frame.restype = C.c_int
    L.orbfe_stereo_frame.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_size_t, vp, vp, C.POINTER(C.c_int),
                                     vp, vp, C.POINTER(C.c_int), vp, vp, C.c_int]
    L.orbfe_rgbd_frame.restype = C.c_int
    L.orbfe_rgbd_frame.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, vp, vp, C.POINTER(C.c_int), vp, vp, C.c_int]
    L.orbfe_rgbd_frame_u16.restype = C.c_int
    L.orbfe_rgbd_frame_u16.argtypes = [vp, vp, vp, C.c_float, C.c_int, C.c_int, C.c_size_t, C.c_size_t, vp, vp, C.POINTER(C.c_int), vp, vp, C.c_int]
    L.orbfe_fetch_pyramid.restype = C.c_int
    L.orbfe_fetch_pyramid.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_size_t]
    L.orbfe_enqueue_extract.restype = C.c_int; L.orbfe_enqueue_extract.argtypes = [vp, vp, C.c_int, vp]
    L.orbfe_enqueue_stereo.restype = C.c_int; L.orbfe_enqueue_stereo.argtypes = [vp, vp, C.c_int, vp]
    L.orbfe_synchronize.restype = C.c_int; L.orbfe_synchronize.argtypes = [vp, vp]
    L.orbfe_fetch_image.restype = C.c_int
    L.orbfe_fetch_image.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.orbfe_fetch_counts.restype = C.c_int; L.orbfe_fetch_counts.argtypes = [vp, vp, C.c_int]
    L.orbfe_device_buffers.restype = C.c_int; L.orbfe_device_buffers.argtypes = [vp] + [C.POINTER(vp)] * 5
    L.orbfe_fetch_candidates.restype = C.c_int
    L.orbfe_fetch_candidates.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.orbfe_hamming_matrix.restype = C.c_int
    L.orbfe_hamming_matrix.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp]
    L.orbfe_set_profiling.restype = C.c_int; L.orbfe_set_profiling.argtypes = [vp, C.c_int]
    L.orbfe_stage_name.restype = C.c_char_p; L.orbfe_stage_name.argtypes = [C.c_int]
    L.orbfe_stage_times.restype = C.c_int; L.orbfe_stage_times.argtypes = [vp, vp, C.POINTER(C.c_int), C.c_int]
    L.orbfe_set_streams.restype = C.c_int; L.orbfe_set_streams.argtypes = [vp, C.c_int]
    L.orbfe_quadtree_kernel.restype = C.c_int; L.orbfe_quadtree_kernel.argtypes = [vp]
    fvp, ip = C.POINTER(FrameView), C.POINTER(C.c_int)
    L.orbfe_features_in_area.restype = C.c_int
    L.orbfe_features_in_area.argtypes = [vp, fvp, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, vp, C.c_int, ip]
    L.orbfe_features_in_area_batch.restype = C.c_int
    L.orbfe_features_in_area_batch.argtypes = [vp, fvp, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int]
    L.orbfe_three_maxima.restype = C.c_int; L.orbfe_three_maxima.argtypes = [vp, C.c_int, ip, ip, ip]
    L.orbfe_search_by_projection_last.restype = C.c_int
    L.orbfe_search_by_projection_last.argtypes = [vp, fvp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, ip]
    L.orbfe_is_in_frustum.restype = C.c_int
    L.orbfe_is_in_frustum.argtypes = [vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, vp, vp, vp, vp, C.c_float, vp]
    L.orbfe_search_by_projection_points.restype = C.c_int
    L.orbfe_search_by_projection_points.argtypes = [vp, fvp, C.c_int, vp, vp, vp, vp, C.c_float, C.c_float, vp, ip]
    L.orbfe_search_by_projection_kf.restype = C.c_int
    L.orbfe_search_by_projection_kf.argtypes = [vp, fvp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, ip]
    L.orbfe_fuse.restype = C.c_int
    L.orbfe_fuse.argtypes = [vp, fvp, vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_float, vp, ip]
    L.orbfe_search_by_projection_sim3.restype = C.c_int
    L.orbfe_search_by_projection_sim3.argtypes = [vp, fvp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp, ip]
    L.orbfe_fuse_sim3.restype = C.c_int
    L.orbfe_fuse_sim3.argtypes = [vp, fvp, vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_float, vp, ip]
    L.orbfe_search_by_sim3.restype = C.c_int
    L.orbfe_search_by_sim3.argtypes = [vp] + [fvp, vp, vp, vp, vp, vp, vp] * 2 + [C.c_float, vp, vp, C.c_float, vp, ip]
    L.orbfe_set_distortion.restype = C.c_int
    L.orbfe_set_distortion.argtypes = [vp, vp, C.c_int]
    L.orbfe_undistort_keypoints.restype = C.c_int
    L.orbfe_undistort_keypoints.argtypes = [vp, vp, C.c_int, vp]
    L.orbfe_fetch_keys_un.restype = C.c_int
    L.orbfe_fetch_keys_un.argtypes = [vp, C.c_int, vp, C.c_int, ip]
    L.orbfe_image_bounds.restype = C.c_int
    L.orbfe_image_bounds.argtypes = [vp, vp]
    L.orbfe_fetch_batch_async.restype = C.c_int
    L.orbfe_fetch_batch_async.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.orbfe_get_packed_layout.restype = C.c_int
    L.orbfe_get_packed_layout.argtypes = [vp, C.c_int, C.c_int, C.POINTER(PackedLayout)]
    L.orbfe_fetch_batch_packed.restype = C.c_int
    L.orbfe_fetch_batch_packed.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t, vp]
    L.orbfe_expand_packed.restype = C.c_int
    L.orbfe_expand_packed.argtypes = [vp, vp, C.POINTER(PackedLayout), C.c_int, vp, C.c_int, ip]
    L.orbfe_enqueue_rgbd.restype = C.c_int
    L.orbfe_enqueue_rgbd.argtypes = [vp, vp, vp, C.c_int, C.c_float, C.c_int, vp]
    L.orbfe_set_rectification.restype = C.c_int
    L.orbfe_set_rectification.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int]
    L.orbfe_set_input_format.restype = C.c_int
    L.orbfe_set_input_format.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.orbfe_pose_optimization.restype = C.c_int
    L.orbfe_pose_optimization.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, ip]
    L.orbfe_pose_optimization_batch.restype = C.c_int
    L.orbfe_pose_optimization_batch.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    L.orbfe_enqueue_pose_optimization.restype = C.c_int
    L.orbfe_enqueue_pose_optimization.argtypes = [vp, C.c_int] + [vp] * 8 + [C.c_int, vp]
    L.orbfe_search_for_initialization.restype = C.c_int
    L.orbfe_search_for_initialization.argtypes = [vp, fvp, fvp, vp, C.c_int, C.c_float, C.c_int, vp, ip]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One device context = one ORBextractor pair / one camera model (include/orbfe.h)."""

    def __init__(self, width, height, nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th_fast=20,
                 min_th_fast=7, patch_size=31, half_patch_size=15, edge_threshold=19,
                 fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448, device=0, max_images=2):
        self.L = load()
        self.params = Params(nfeatures, scale_factor, nlevels, ini_th_fast, min_th_fast, patch_size,
                             half_patch_size, edge_threshold, fx, fy, cx, cy, bf, device, width, height, max_images)
        h = C.c_void_p()
        rc = self.L.orbfe_create(C.byref(self.params), C.byref(h))
        if rc != OK:
            raise OrbfeError(rc, self.L.orbfe_last_error(None).decode())
        self.h = h
        self.width, self.height = width, height
        self.nlevels = nlevels
        self.capacity = self.L.orbfe_keypoint_capacity(self.h)
        self.max_images = max_images

    def close(self):
        if getattr(self, "h", None):
            self.L.orbfe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            raise OrbfeError(rc, self.L.orbfe_last_error(self.h).decode())

    # ---- tables (ORBextractor getters) ----
    def tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        feats = np.zeros(n, np.int32)
        umax = np.zeros(self.params.half_patch_size + 1, np.int32)
        self._check(self.L.orbfe_get_tables(self.h, _p(sc), _p(isc), _p(s2), _p(is2), _p(feats), _p(umax)))
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, features_per_level=feats, umax=umax)

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        self._check(self.L.orbfe_level_size(self.h, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def blur_ride_from(self, n_images):
        """First pyramid level whose blur rides in FAST's launch for a batch of n_images (nlevels: none)."""
        r = self.L.orbfe_blur_ride_from(self.h, n_images)
        if r < 0:
            raise OrbfeError(r, "orbfe_blur_ride_from")
        return r

    def set_input_retained(self, retained=True):
        """Promise that the images of an enqueue_* call stay valid until the next one (orbfe_fetch_pyramid's level 0 then reads them in place)."""
        self._check(self.L.orbfe_set_input_retained(self.h, int(bool(retained))))

    def set_pattern(self, pattern):
        """Replace the context's copy of the 256 x (x0, y0, x1, y1) rBRIEF tests (ORBextractor::pattern, src/ORBextractor.cc:442-444)."""
        pat = np.ascontiguousarray(np.asarray(pattern, dtype=np.int32).reshape(1024))
        self._check(self.L.orbfe_set_pattern(self.h, _p(pat)))

    def pattern(self):
        pat = np.zeros(1024, np.int32)
        self._check(self.L.orbfe_get_pattern(self.h, _p(pat)))
        return pat.reshape(256, 4)

    # ---- host-image entry points ----
    @staticmethod
    def _rows(a, dtype):
        """Row-strided views (a cv::Mat ROI) go through as they are; anything else is made contiguous.  HxWxC colour
        images (orbfe_set_input_format) must have packed pixels."""
        a = np.asarray(a, dtype)
        if a.ndim == 2 and a.strides[1] == a.itemsize and a.strides[0] >= a.shape[1] * a.itemsize:
            return a
        if a.ndim == 3 and a.strides[2] == a.itemsize and a.strides[1] == a.shape[2] * a.itemsize and a.strides[0] >= a.shape[1] * a.strides[1]:
            return a
        return np.ascontiguousarray(a)

    def set_distortion(self, dist):
        """mDistCoef = k1 k2 p1 p2 [k3] (src/Tracking.cc:67-78); empty: no distortion."""
        d = np.ascontiguousarray(dist, np.float32)
        self._check(self.L.orbfe_set_distortion(self.h, _p(d) if len(d) else None, len(d)))

    def undistort_keypoints(self, kps):
        k = np.ascontiguousarray(kps, KP_DTYPE); out = np.zeros(max(len(k), 1), KP_DTYPE)
        self._check(self.L.orbfe_undistort_keypoints(self.h, _p(k), len(k), _p(out)))
        return out[: len(k)].copy()

    def fetch_keys_un(self, image=0):
        cap = self.capacity
        out = np.zeros(cap, KP_DTYPE); n = C.c_int()
        self._check(self.L.orbfe_fetch_keys_un(self.h, image, _p(out), cap, C.byref(n)))
        return out[: n.value].copy()

    def image_bounds(self):
        b = np.zeros(4, np.float32)
        self._check(self.L.orbfe_image_bounds(self.h, _p(b)))
        return b

    def set_rectification(self, side, map_x=None, map_y=None, src_size=None):
        """cv::remap(raw, map1, map2, INTER_LINEAR) in front of the pipeline (stereo_euroc.cc:136-137); None clears."""
        if map_x is None:
            self._check(self.L.orbfe_set_rectification(self.h, side, None, None, 0, 0))
            return
        mx = np.ascontiguousarray(map_x, np.float32); my = np.ascontiguousarray(map_y, np.float32)
        assert mx.shape == my.shape
        sw, sh = src_size if src_size is not None else (mx.shape[1], mx.shape[0])
        self._check(self.L.orbfe_set_rectification(self.h, side, _p(mx), _p(my), sw, sh))

    def set_input_format(self, channels=1, rgb=True, legacy_weights=False):
        """cv::cvtColor(..., COLOR_{RGB,BGR}[A]2GRAY) of Tracking::GrabImage* folded into ingest (src/Tracking.cc:269-294)."""
        self._check(self.L.orbfe_set_input_format(self.h, channels, int(rgb), int(legacy_weights)))

    def extract(self, img: np.ndarray):
        img = self._rows(img, np.uint8)
        cap = self.capacity
        kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int()
        self._check(self.L.orbfe_extract(self.h, _p(img), img.shape[1], img.shape[0], img.strides[0], _p(kps), _p(desc), cap, C.byref(n)))
        return kps[: n.value].copy(), desc[: n.value].copy()

    def stereo_frame(self, left: np.ndarray, right: np.ndarray):
        left = self._rows(left, np.uint8); right = self._rows(right, np.uint8)
        assert left.shape == right.shape
        if left.strides[0] != right.strides[0]:
            left = np.ascontiguousarray(left); right = np.ascontiguousarray(right)
        cap = self.capacity
        kl = np.zeros(cap, KP_DTYPE); dl = np.zeros((cap, 32), np.uint8)
        kr = np.zeros(cap, KP_DTYPE); dr = np.zeros((cap, 32), np.uint8)
        ur = np.zeros(cap, np.float32); dp = np.zeros(cap, np.float32)
        nl, nr = C.c_int(), C.c_int()
        self._check(self.L.orbfe_stereo_frame(self.h, _p(left), _p(right), left.shape[1], left.shape[0], left.strides[0],
                                              _p(kl), _p(dl), C.byref(nl), _p(kr), _p(dr), C.byref(nr), _p(ur), _p(dp), cap))
        a, b = nl.value, nr.value
        return dict(kps_left=kl[:a].copy(), desc_left=dl[:a].copy(), kps_right=kr[:b].copy(), desc_right=dr[:b].copy(),
                    u_right=ur[:a].copy(), depth=dp[:a].copy())

    def rgbd_frame(self, gray: np.ndarray, depth_img: np.ndarray, depth_map_factor: float = 1.0):
        """depth_img float32: the CV_32F map of Frame::Frame(rgbd).  depth_img uint16: the raw sensor map, converted
        with Tracking's mDepthMapFactor (src/Tracking.cc:323-324) on the device."""
        raw = np.asarray(depth_img).dtype == np.uint16
        gray = self._rows(gray, np.uint8); depth_img = self._rows(depth_img, np.uint16 if raw else np.float32)
        cap = self.capacity
        k = np.zeros(cap, KP_DTYPE); d = np.zeros((cap, 32), np.uint8)
        ur = np.zeros(cap, np.float32); dp = np.zeros(cap, np.float32)
        n = C.c_int()
        if raw:
            self._check(self.L.orbfe_rgbd_frame_u16(self.h, _p(gray), _p(depth_img), float(depth_map_factor), gray.shape[1], gray.shape[0],
                                                    gray.strides[0], depth_img.strides[0], _p(k), _p(d), C.byref(n), _p(ur), _p(dp), cap))
        else:
            self._check(self.L.orbfe_rgbd_frame(self.h, _p(gray), _p(depth_img), gray.shape[1], gray.shape[0], gray.strides[0],
                                                depth_img.strides[0], _p(k), _p(d), C.byref(n), _p(ur), _p(dp), cap))
        m = n.value
        return dict(kps=k[:m].copy(), desc=d[:m].copy(), u_right=ur[:m].copy(), depth=dp[:m].copy())

    # ---- taps ----
    def fetch_pyramid(self, image, level, blurred=False):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        self._check(self.L.orbfe_fetch_pyramid(self.h, image, level, int(blurred), _p(out), out.strides[0]))
        return out

    def fetch_candidates(self, image, level):
        w, h = self.level_size(level)
        cap = max(16, (w * h) // 4 + 16)
        xs, ys, sc = (np.zeros(cap, np.int32) for _ in range(3))
        n = C.c_int()
        self._check(self.L.orbfe_fetch_candidates(self.h, image, level, _p(xs), _p(ys), _p(sc), cap, C.byref(n)))
        return xs[: n.value].copy(), ys[: n.value].copy(), sc[: n.value].copy()

    # ---- device-resident batched path ----
    def enqueue_extract(self, d_ptr: int, n_images: int, stream: int = 0):
        self._check(self.L.orbfe_enqueue_extract(self.h, C.c_void_p(d_ptr), n_images, C.c_void_p(stream)))

    def enqueue_stereo(self, d_ptr: int, n_pairs: int, stream: int = 0):
        self._check(self.L.orbfe_enqueue_stereo(self.h, C.c_void_p(d_ptr), n_pairs, C.c_void_p(stream)))

    def synchronize(self, stream: int = 0):
        self._check(self.L.orbfe_synchronize(self.h, C.c_void_p(stream)))

    def fetch_batch_async(self, n_images, kps_ptr, desc_ptr, counts_ptr, u_right_ptr, depth_ptr, stream=0):
        """Raw-pointer form (pinned host buffers owned by the caller); see include/orbfe.h."""
        self._check(self.L.orbfe_fetch_batch_async(self.h, n_images, C.c_void_p(kps_ptr), C.c_void_p(desc_ptr), C.c_void_p(counts_ptr),
                                                   C.c_void_p(u_right_ptr), C.c_void_p(depth_ptr), C.c_void_p(stream)))

    def packed_layout(self, n_images, flags=0):
        lay = PackedLayout()
        self._check(self.L.orbfe_get_packed_layout(self.h, n_images, flags, C.byref(lay)))
        return lay

    def fetch_batch_packed(self, n_images, flags, host_ptr, host_bytes, stream=0):
        """Raw-pointer form: one small gather kernel + ONE device-to-host copy of the packed block (include/orbfe.h)."""
        self._check(self.L.orbfe_fetch_batch_packed(self.h, n_images, flags, C.c_void_p(host_ptr), host_bytes, C.c_void_p(stream)))

    def expand_packed(self, block, lay, out_image):
        """cv::KeyPoint records (+ views of descriptors / uRight / depth) of one out image of a fetched block (numpy uint8 array)."""
        k = np.zeros(lay.capacity, KP_DTYPE)
        n = C.c_int()
        self._check(self.L.orbfe_expand_packed(self.h, _p(block), C.byref(lay), out_image, _p(k), lay.capacity, C.byref(n)))
        m, cap = n.value, lay.capacity
        out = dict(kps=k[:m].copy(), desc=block[lay.desc_off + out_image * cap * 32: lay.desc_off + (out_image * cap + m) * 32].reshape(m, 32).copy())
        slot = out_image * (2 if lay.flags & PACK_LEFT_ONLY else 1)
        if (lay.flags & PACK_STEREO) and slot % 2 == 0:
            pr = slot // 2
            out["u_right"] = block[lay.u_right_off + pr * cap * 4: lay.u_right_off + (pr * cap + m) * 4].view(np.float32).copy()
            out["depth"] = block[lay.depth_off + pr * cap * 4: lay.depth_off + (pr * cap + m) * 4].view(np.float32).copy()
        return out

    def fetch_packed(self, n_images, flags=0, stream=0):
        """Blocking convenience for tests: (block, layout) of the latest batched call's results."""
        lay = self.packed_layout(n_images, flags)
        block = np.zeros(lay.bytes, np.uint8)
        self.fetch_batch_packed(n_images, flags, block.ctypes.data, lay.bytes, stream)
        self.synchronize(stream)
        return block, lay

    def enqueue_rgbd(self, gray_ptr, depth_ptr, n_images, depth_is_u16=False, depth_map_factor=1.0, stream=0):
        self._check(self.L.orbfe_enqueue_rgbd(self.h, C.c_void_p(gray_ptr), C.c_void_p(depth_ptr), 1 if depth_is_u16 else 0, depth_map_factor, n_images, C.c_void_p(stream)))

    def fetch_counts(self, n_images):
        c = np.zeros(n_images, np.int32)
        self._check(self.L.orbfe_fetch_counts(self.h, _p(c), n_images))
        return c

    def fetch_image(self, image, stereo=False):
        cap = self.capacity
        k = np.zeros(cap, KP_DTYPE); d = np.zeros((cap, 32), np.uint8)
        ur = np.zeros(cap, np.float32); dp = np.zeros(cap, np.float32)
        n = C.c_int()
        self._check(self.L.orbfe_fetch_image(self.h, image, _p(k), _p(d), _p(ur) if stereo else None,
                                             _p(dp) if stereo else None, cap, C.byref(n)))
        m = n.value
        out = dict(kps=k[:m].copy(), desc=d[:m].copy())
        if stereo:
            out.update(u_right=ur[:m].copy(), depth=dp[:m].copy())
        return out

    def quadtree_kernel(self) -> int:
        return int(self.L.orbfe_quadtree_kernel(self.h))

    def set_streams(self, groups: int):
        self._check(self.L.orbfe_set_streams(self.h, groups))

    def set_profiling(self, mode):
        """0/False = off, 1/True = events at every stage boundary, 2 + k = only around stage k."""
        self._check(self.L.orbfe_set_profiling(self.h, int(mode)))

    def set_profiling_interval(self, every: int):
        self.L.orbfe_set_profiling_interval.restype = C.c_int
        self.L.orbfe_set_profiling_interval.argtypes = [C.c_void_p, C.c_int]
        self._check(self.L.orbfe_set_profiling_interval(self.h, int(every)))

    def stage_times(self, reset=True):
        """{stage: total ms} over the recorded enqueue calls, and the number of calls."""
        ms = np.zeros(NUM_STAGES, np.float32)
        calls = C.c_int()
        self._check(self.L.orbfe_stage_times(self.h, _p(ms), C.byref(calls), int(reset)))
        names = [self.L.orbfe_stage_name(i).decode() for i in range(NUM_STAGES)]
        return dict(zip(names, ms.tolist())), calls.value

    # ---- Tracking-thread matchers (flattened inputs; see include/orbfe.h) ----
    def _view(self, keys_un, u_right, desc, bounds, device_slot=None, keyframe=False):
        """orbfe_frame_view over host arrays; device_slot = k makes the matchers read image slot k of the latest extraction call
        in HBM instead of uploading the arrays (the host copies are still needed by the host-side accept rules); keyframe: the
        view is a KeyFrame's (windows and the in-image test use the integer-truncated bounds, include/orbfe.h)."""
        k = np.ascontiguousarray(keys_un, KP_DTYPE); d = np.ascontiguousarray(desc, np.uint8)
        ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        fv = FrameView(len(k), _p(k), None if ur is None else _p(ur), _p(d), *[float(b) for b in bounds],
                       0 if device_slot is None else device_slot + 1, 1 if keyframe else 0)
        fv._keep = (k, d, ur)
        return fv

    def features_in_area_batch(self, view, x, y, r, min_level=None, max_level=None):
        """List of index arrays, one per query (Frame::GetFeaturesInArea order)."""
        x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32); r = np.ascontiguousarray(r, np.float32)
        nq = len(x)
        lo = None if min_level is None else np.ascontiguousarray(min_level, np.int32)
        hi = None if max_level is None else np.ascontiguousarray(max_level, np.int32)
        off = np.zeros(nq + 1, np.int32); cap = max(view.n * max(nq, 1), 1)
        out = np.zeros(cap, np.int32)
        self._check(self.L.orbfe_features_in_area_batch(self.h, C.byref(view), nq, _p(x), _p(y), _p(r), None if lo is None else _p(lo),
                                                        None if hi is None else _p(hi), _p(off), _p(out), cap))
        return [out[off[i]:off[i + 1]].copy() for i in range(nq)]

    def features_in_area(self, view, x, y, r, min_level=-1, max_level=-1):
        out = np.zeros(max(view.n, 1), np.int32); n = C.c_int()
        self._check(self.L.orbfe_features_in_area(self.h, C.byref(view), x, y, r, min_level, max_level, _p(out), len(out), C.byref(n)))
        return out[: n.value].copy()

    def search_by_projection_last(self, view, Tcw_cur, Tcw_last, last_pos, last_desc, last_valid, last_obs, last_octave, last_angle,
                                  cur_has_obs, th, mono, check_ori):
        tc = np.ascontiguousarray(Tcw_cur, np.float32); tl = np.ascontiguousarray(Tcw_last, np.float32)
        lp = np.ascontiguousarray(last_pos, np.float32); ld = np.ascontiguousarray(last_desc, np.uint8)
        lv = np.ascontiguousarray(last_valid, np.int32); lo = np.ascontiguousarray(last_obs, np.int32)
        loc = np.ascontiguousarray(last_octave, np.int32); la = np.ascontiguousarray(last_angle, np.float32)
        ho = None if cur_has_obs is None else np.ascontiguousarray(cur_has_obs, np.uint8)
        out = np.zeros(max(view.n, 1), np.int32); nm = C.c_int()
        self._check(self.L.orbfe_search_by_projection_last(self.h, C.byref(view), _p(tc), _p(tl), len(lv), _p(lp), _p(ld), _p(lv), _p(lo),
                                                           _p(loc), _p(la), None if ho is None else _p(ho), th, int(mono), int(check_ori),
                                                           _p(out), C.byref(nm)))
        return out[: view.n].copy(), nm.value

    def is_in_frustum(self, Tcw, bounds, pos, normal, max_distance, min_distance, viewing_cos_limit):
        tc = np.ascontiguousarray(Tcw, np.float32); pos = np.ascontiguousarray(pos, np.float32); nr = np.ascontiguousarray(normal, np.float32)
        mx = np.ascontiguousarray(max_distance, np.float32); mn = np.ascontiguousarray(min_distance, np.float32)
        out = np.zeros(len(pos), TP_DTYPE)
        self._check(self.L.orbfe_is_in_frustum(self.h, _p(tc), bounds[0], bounds[1], bounds[2], bounds[3], len(pos), _p(pos), _p(nr), _p(mx),
                                               _p(mn), viewing_cos_limit, _p(out)))
        return out

    def search_by_projection_points(self, view, pts, pt_desc, pt_obs, cur_has_obs, th, nnratio):
        pts = np.ascontiguousarray(pts, TP_DTYPE); pd = np.ascontiguousarray(pt_desc, np.uint8); po = np.ascontiguousarray(pt_obs, np.int32)
        ho = None if cur_has_obs is None else np.ascontiguousarray(cur_has_obs, np.uint8)
        out = np.zeros(max(view.n, 1), np.int32); nm = C.c_int()
        self._check(self.L.orbfe_search_by_projection_points(self.h, C.byref(view), len(pts), _p(pts), _p(pd), _p(po),
                                                             None if ho is None else _p(ho), th, nnratio, _p(out), C.byref(nm)))
        return out[: view.n].copy(), nm.value

    def search_by_projection_kf(self, view, Tcw_cur, kf_pos, kf_desc, kf_valid, kf_angle, kf_max_distance, kf_min_distance, cur_has_point,
                                th, orb_dist, check_ori):
        tc = np.ascontiguousarray(Tcw_cur, np.float32); kp = np.ascontiguousarray(kf_pos, np.float32)
        kd = np.ascontiguousarray(kf_desc, np.uint8); kv = np.ascontiguousarray(kf_valid, np.int32); ka = np.ascontiguousarray(kf_angle, np.float32)
        kmx = np.ascontiguousarray(kf_max_distance, np.float32); kmn = np.ascontiguousarray(kf_min_distance, np.float32)
        hp = None if cur_has_point is None else np.ascontiguousarray(cur_has_point, np.uint8)
        out = np.zeros(max(view.n, 1), np.int32); nm = C.c_int()
        self._check(self.L.orbfe_search_by_projection_kf(self.h, C.byref(view), _p(tc), len(kv), _p(kp), _p(kd), _p(kv), _p(ka), _p(kmx), _p(kmn),
                                                         None if hp is None else _p(hp), th, orb_dist, int(check_ori), _p(out), C.byref(nm)))
        return out[: view.n].copy(), nm.value

    def fuse(self, view, Tcw, pos, normal, max_distance, min_distance, pt_desc, pt_valid, th):
        """Search part of ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th): best keypoint per map point (-1: none), and the count."""
        t = np.ascontiguousarray(Tcw, np.float32); p = np.ascontiguousarray(pos, np.float32); nrm = np.ascontiguousarray(normal, np.float32)
        mx = np.ascontiguousarray(max_distance, np.float32); mn = np.ascontiguousarray(min_distance, np.float32)
        d = np.ascontiguousarray(pt_desc, np.uint8); ok = np.ascontiguousarray(pt_valid, np.int32)
        out = np.zeros(max(len(ok), 1), np.int32); nf = C.c_int()
        self._check(self.L.orbfe_fuse(self.h, C.byref(view), _p(t), len(ok), _p(p), _p(nrm), _p(mx), _p(mn), _p(d), _p(ok), th, _p(out), C.byref(nf)))
        return out[: len(ok)].copy(), nf.value

    def sim3_projection(self, mode, view, Scw, pos, normal, max_distance, min_distance, pt_desc, pt_valid, kf_matched, th):
        """mode 0: SearchByProjection(KeyFrame*, Scw, ...); mode 1: search part of Fuse(KeyFrame*, Scw, ...): (match per point, count)."""
        t = np.ascontiguousarray(Scw, np.float32); p = np.ascontiguousarray(pos, np.float32); nrm = np.ascontiguousarray(normal, np.float32)
        mx = np.ascontiguousarray(max_distance, np.float32); mn = np.ascontiguousarray(min_distance, np.float32)
        d = np.ascontiguousarray(pt_desc, np.uint8); ok = np.ascontiguousarray(pt_valid, np.int32)
        out = np.zeros(max(len(ok), 1), np.int32); nf = C.c_int()
        if mode == 0:
            km = None if kf_matched is None else np.ascontiguousarray(kf_matched, np.uint8)
            self._check(self.L.orbfe_search_by_projection_sim3(self.h, C.byref(view), _p(t), len(ok), _p(p), _p(nrm), _p(mx), _p(mn), _p(d), _p(ok),
                                                               None if km is None else _p(km), th, _p(out), C.byref(nf)))
        else:
            self._check(self.L.orbfe_fuse_sim3(self.h, C.byref(view), _p(t), len(ok), _p(p), _p(nrm), _p(mx), _p(mn), _p(d), _p(ok), th, _p(out), C.byref(nf)))
        return out[: len(ok)].copy(), nf.value

    def search_by_sim3(self, view1, T1w, pts1, view2, T2w, pts2, s12, R12, t12, th):
        """ORBmatcher::SearchBySim3; pts = (pos, max_distance, min_distance, desc, valid) per keypoint slot.  (match12, count)."""
        def prep(T, pts):
            pos, mx, mn, d, ok = pts
            return (np.ascontiguousarray(T, np.float32), np.ascontiguousarray(pos, np.float32), np.ascontiguousarray(mx, np.float32),
                    np.ascontiguousarray(mn, np.float32), np.ascontiguousarray(d, np.uint8), np.ascontiguousarray(ok, np.int32))
        a, b = prep(T1w, pts1), prep(T2w, pts2)
        R = np.ascontiguousarray(R12, np.float32); t = np.ascontiguousarray(t12, np.float32)
        out = np.zeros(max(view1.n, 1), np.int32); nf = C.c_int()
        self._check(self.L.orbfe_search_by_sim3(self.h, C.byref(view1), *[_p(x) for x in a], C.byref(view2), *[_p(x) for x in b],
                                                float(s12), _p(R), _p(t), th, _p(out), C.byref(nf)))
        return out[: view1.n].copy(), nf.value

    def pose_optimization(self, Tcw, keys_un, u_right, has_point, Xw, outlier=None):
        """Optimizer::PoseOptimization (src/Optimizer.cc:283-495).  Returns (Tcw 4x4 float32, outlier uint8[N], n_inliers)."""
        T = np.ascontiguousarray(Tcw, np.float32).reshape(4, 4).copy()
        k = np.ascontiguousarray(keys_un, KP_DTYPE); ur = np.ascontiguousarray(u_right, np.float32)
        hp = np.ascontiguousarray(has_point, np.uint8); X = np.ascontiguousarray(Xw, np.float32)
        out = np.zeros(max(len(k), 1), np.uint8) if outlier is None else np.ascontiguousarray(outlier, np.uint8).copy()
        n = C.c_int()
        self._check(self.L.orbfe_pose_optimization(self.h, _p(T), len(k), _p(k), _p(ur), _p(hp), _p(X), _p(out), C.byref(n)))
        return T, out[: len(k)].copy(), n.value

    def pose_optimization_batch(self, Tcw, offsets, keys_un, u_right, has_point, Xw, outlier=None):
        """One problem per slice offsets[k]:offsets[k+1].  Returns (Tcw [P,4,4], outlier, n_inliers int32[P])."""
        off = np.ascontiguousarray(offsets, np.int32); P = len(off) - 1
        T = np.ascontiguousarray(Tcw, np.float32).reshape(P, 4, 4).copy()
        k = np.ascontiguousarray(keys_un, KP_DTYPE); ur = np.ascontiguousarray(u_right, np.float32)
        hp = np.ascontiguousarray(has_point, np.uint8); X = np.ascontiguousarray(Xw, np.float32)
        out = np.zeros(max(len(k), 1), np.uint8) if outlier is None else np.ascontiguousarray(outlier, np.uint8).copy()
        n = np.zeros(max(P, 1), np.int32)
        self._check(self.L.orbfe_pose_optimization_batch(self.h, P, _p(off), _p(T), _p(k), _p(ur), _p(hp), _p(X), _p(out), _p(n)))
        return T, out[: len(k)].copy(), n[:P].copy()

    def search_for_initialization(self, view1, view2, prev_matched, window_size, nnratio, check_ori):
        pm = np.ascontiguousarray(prev_matched, np.float32).copy()
        out = np.zeros(max(view1.n, 1), np.int32); nm = C.c_int()
        self._check(self.L.orbfe_search_for_initialization(self.h, C.byref(view1), C.byref(view2), _p(pm), window_size, nnratio,
                                                           int(check_ori), _p(out), C.byref(nm)))
        return out[: view1.n].copy(), pm, nm.value

    def hamming_matrix(self, a: np.ndarray, b: np.ndarray):
        a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
        out = np.zeros((len(a), len(b)), np.int32)
        self._check(self.L.orbfe_hamming_matrix(self.h, _p(a), len(a), _p(b), len(b), _p(out)))
        return out


# ---- PNG input (orbfe_png_*): no context, no GPU ----
def png_info(data: bytes):
    """(width, height, channels, bit_depth) cv::imread(..., IMREAD_UNCHANGED) would return for this PNG file."""
    L = load()
    buf = np.frombuffer(data, np.uint8)
    w, h, ch, bd = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = L.orbfe_png_info(_p(buf), len(buf), C.byref(w), C.byref(h), C.byref(ch), C.byref(bd))
    if rc != OK:
        L.orbfe_png_last_error.restype = C.c_char_p
        raise OrbfeError(rc, L.orbfe_png_last_error().decode())
    return w.value, h.value, ch.value, bd.value


def png_decode(data: bytes) -> np.ndarray:
    """cv::imread(path, IMREAD_UNCHANGED) of a PNG file held in memory: HxW or HxWxC array, uint8 or uint16."""
    L = load()
    w, h, ch, bd = png_info(data)
    out = np.zeros((h, w, ch), np.uint16 if bd == 16 else np.uint8)
    buf = np.frombuffer(data, np.uint8)
    L.orbfe_png_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t] + [C.POINTER(C.c_int)] * 4
    rc = L.orbfe_png_decode(_p(buf), len(buf), _p(out), out.nbytes, 0, None, None, None, None)
    if rc != OK:
        L.orbfe_png_last_error.restype = C.c_char_p
        raise OrbfeError(rc, L.orbfe_png_last_error().decode())
    return out[:, :, 0] if ch == 1 else out


def png_decode_batch(files, width, height, channels=1, bit_depth=8, threads=0, out=None) -> np.ndarray:
    """n PNG files of one geometry -> [n, H, W(, C)] array (optionally a caller buffer, e.g. pinned memory)."""
    L = load()
    n = len(files)
    shape = (n, height, width) if channels == 1 else (n, height, width, channels)
    if out is None:
        out = np.zeros(shape, np.uint16 if bit_depth == 16 else np.uint8)
    bufs = [np.frombuffer(f, np.uint8) for f in files]
    ptrs = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in bufs])
    sizes = (C.c_size_t * max(n, 1))(*[len(b) for b in bufs])
    L.orbfe_png_decode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t] + [C.c_int] * 5
    rc = L.orbfe_png_decode_batch(ptrs, sizes, n, _p(out), out.nbytes // max(n, 1), width, height, channels, bit_depth, threads)
    if rc != OK:
        L.orbfe_png_last_error.restype = C.c_char_p
        raise OrbfeError(rc, L.orbfe_png_last_error().decode())
    return out
