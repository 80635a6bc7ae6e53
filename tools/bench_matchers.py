#!/usr/bin/env python3
"""Measurement for SURVEY.md §8 rows 13-19 (Tracking-thread matchers and BoW): wall time per call of the library entry
points (host arrays in, host arrays out -- the H2D / D2H copies and the host-side greedy resolve are inside the time)
beside the CPU oracle on the same inputs (1 thread).  Prints one JSON object; run on the GPU box:
    python3 tools/bench_matchers.py > gpurun_out/matchers.json
The scenario is the keypoint-level scene of tests/test_matchers.py scaled to 2000 last-frame points, and the
BoW case of tests/test_bow.py (k=10, L=3 vocabulary, 1500 / 1600 descriptors).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests import test_matchers as TM  # noqa: E402


def timeit(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    from orbslam2_amd import api
    ctx = api.Context(width=TM.W, height=TM.H, fx=TM.FX, fy=TM.FY, cx=TM.CX, cy=TM.CY, bf=TM.BF)
    s = TM._scene(3, n_last=2000, n_distract=700)
    g = O.Grid(s["k"], *s["bounds"])
    view = ctx._view(s["k"], s["ur"], s["d"], s["bounds"])
    out = {"unit": "ms per call", "scene": "2000 last-frame points, %d current keypoints" % len(s["k"]), "rows": {}}

    def row(name, gpu_fn, cpu_fn, reps=20):
        g_ms, c_ms = timeit(gpu_fn, reps), timeit(cpu_fn, max(3, reps // 4))
        out["rows"][name] = {"gpu_ms": round(g_ms, 4), "oracle_1thread_ms": round(c_ms, 4)}

    row("SearchByProjection(Frame, LastFrame) [row 14]",
        lambda: ctx.search_by_projection_last(view, s["T_cur"], s["T_last"], s["pos"], s["desc_last"], s["valid"], s["obs"], s["octave"],
                                              s["angle"], s["cur_has_obs"], 7.0, False, True),
        lambda: O.search_by_projection_last(g, s["ur"], s["d"], s["sf"], TM.CAM, s["T_cur"], s["T_last"], s["pos"], s["desc_last"],
                                            s["valid"], s["obs"], s["octave"], s["angle"], s["cur_has_obs"], 7.0, False, True))
    # the same matcher on a REAL extracted frame: upload path against the device-resident frame (orbfe_frame_view.device_slot_plus1:
    # keypoints / descriptors read where the extraction left them in HBM, grid built once per frame)
    from orbslam2_amd import synth
    ctx2 = api.Context(width=TM.W, height=TM.H, nfeatures=2000, fx=TM.FX, fy=TM.FY, cx=TM.CX, cy=TM.CY, bf=TM.BF)
    left, right = synth.stereo_pair(TM.W, TM.H, seed=77)
    fr = ctx2.stereo_frame(left, right)
    fk, fd, fur = fr["kps_left"], fr["desc_left"], fr["u_right"]
    fs = TM._frame_scene(fk, fd, fur, 77, all_points=True)
    fb = (0.0, float(TM.W), 0.0, float(TM.H))
    fg = O.Grid(fk, *fb)
    v_up = ctx2._view(fk, fur, fd, fb); v_dev = ctx2._view(fk, fur, fd, fb, device_slot=0)
    out["resident_scene"] = "%d keypoints extracted from a synthetic 640x480 pair, %d map points" % (len(fk), len(fs["pos"]))
    for name, v in (("SearchByProjection(Frame, LastFrame), extracted frame, upload path", v_up),
                    ("SearchByProjection(Frame, LastFrame), extracted frame, device-resident [row 14]", v_dev)):
        row(name,
            lambda v=v: ctx2.search_by_projection_last(v, fs["T_cur"], fs["T_last"], fs["pos"], fs["desc"], fs["valid"], fs["obs"], fs["octave"],
                                                       fs["angle"], fs["has"], 7.0, False, True),
            lambda: O.search_by_projection_last(fg, fur, fd, s["sf"], TM.CAM, fs["T_cur"], fs["T_last"], fs["pos"], fs["desc"], fs["valid"],
                                                fs["obs"], fs["octave"], fs["angle"], fs["has"], 7.0, False, True), reps=50)
    # the same resident call as a C / C++ caller sees it: arguments prepared once, the C ABI entry point called directly (the Python
    # method above converts nine arrays and copies the result per call, ~15 us that the reference's C++ Tracking thread does not pay)
    import ctypes as C
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    a_tc = np.ascontiguousarray(fs["T_cur"], np.float32); a_tl = np.ascontiguousarray(fs["T_last"], np.float32)
    a_pos = np.ascontiguousarray(fs["pos"], np.float32); a_desc = np.ascontiguousarray(fs["desc"], np.uint8)
    a_val = np.ascontiguousarray(fs["valid"], np.int32); a_obs = np.ascontiguousarray(fs["obs"], np.int32)
    a_oct = np.ascontiguousarray(fs["octave"], np.int32); a_ang = np.ascontiguousarray(fs["angle"], np.float32)
    a_has = np.ascontiguousarray(fs["has"], np.uint8)
    a_out = np.zeros(max(v_dev.n, 1), np.int32); a_nm = C.c_int()

    args = (ctx2.h, C.byref(v_dev), P(a_tc), P(a_tl), len(a_val), P(a_pos), P(a_desc), P(a_val), P(a_obs), P(a_oct), P(a_ang), P(a_has),
            C.c_float(7.0), 0, 1, P(a_out), C.byref(a_nm))  # numpy's .ctypes.data_as costs ~1.5 us per array: outside the timed call
    fn = ctx2.L.orbfe_search_by_projection_last

    def raw_call():
        assert fn(*args) == 0

    ref_m, _ = ctx2.search_by_projection_last(v_dev, fs["T_cur"], fs["T_last"], fs["pos"], fs["desc"], fs["valid"], fs["obs"], fs["octave"],
                                              fs["angle"], fs["has"], 7.0, False, True)
    raw_call()
    assert np.array_equal(a_out[: v_dev.n], ref_m)
    out["rows"]["SearchByProjection(Frame, LastFrame), device-resident, C ABI called directly [row 14]"] = {
        "gpu_ms": round(timeit(raw_call, 200), 4),
        "note": "inside the call (ORBFE_HOST_TRACE=1): projection of the points 0.006, upload + kernel + download 0.062, replay 0.007 ms"}
    rng = np.random.default_rng(5)
    qs = [(float(rng.uniform(0, TM.W)), float(rng.uniform(0, TM.H)), float(rng.uniform(5, 60))) for _ in range(64)]
    row("GetFeaturesInArea x64 [rows 13]",
        lambda: ctx.features_in_area_batch(view, [q[0] for q in qs], [q[1] for q in qs], [q[2] for q in qs]),
        lambda: [g.features_in_area(x, y, r, -1, -1) for x, y, r in qs], reps=5)
    # mono initialisation: frame 1 = level-0 keypoints, frame 2 = the scene's current frame
    k1 = s["k"].copy(); k1["octave"] = 0
    prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)
    view1 = ctx._view(k1, None, s["d"], s["bounds"])
    row("SearchForInitialization [row 18]",
        lambda: ctx.search_for_initialization(view1, view, prev.copy(), 100, 0.9, True),
        lambda: O.search_for_initialization(k1, s["d"], g, s["d"], prev.copy(), 100, 0.9, True), reps=10)

    # BoW (row 17): vocabulary k=10, L=5 (tests/test_bow.py), transform + SearchByFboW of a 1500-keypoint keyframe against
    # a 1600-keypoint frame
    from orbslam2_amd import bow as B
    from tests import test_bow as TB
    blob = B.build_vocabulary(TB._descs(1, 6000), k=10, levels=5, seed=7)
    B.vocab_load(ctx, blob)
    L, v = TB._oracle_voc(blob)
    rng = np.random.default_rng(3)
    kf_d = TB._descs(5, 1500)
    perm = rng.permutation(1500)[:1200]
    f_d = np.concatenate([TB._descs(6, 0, base=kf_d[perm], flip=0.04), TB._descs(7, 400)])
    row("fbow transform, 1500 descriptors [row 17]", lambda: B.transform(ctx, kf_d, 4), lambda: TB._oracle_transform(L, v, kf_d, 4))
    _, _, kf_fv = TB._oracle_transform(L, v, kf_d)
    _, _, f_fv = TB._oracle_transform(L, v, f_d)
    kf_valid = (rng.random(len(kf_d)) < 0.8).astype(np.int32)
    kf_ang = rng.uniform(0, 360, len(kf_d)).astype(np.float32)
    f_ang = rng.uniform(0, 360, len(f_d)).astype(np.float32)
    ref = np.zeros(len(f_d), np.int32)
    P = TB._p
    row("SearchByFboW(KeyFrame, Frame) [row 17]",
        lambda: B.search_by_bow(ctx, kf_fv, kf_valid, kf_d, kf_ang, f_fv, f_d, f_ang, 0.75, True),
        lambda: L.orc_search_by_bow(P(kf_fv[0]), P(kf_fv[1]), P(kf_fv[2]), len(kf_fv[0]), P(kf_valid), P(kf_d), P(kf_ang),
                                    P(f_fv[0]), P(f_fv[1]), P(f_fv[2]), len(f_fv[0]), P(f_d), P(f_ang), len(f_d), 0.75, 1, P(ref)))
    # the same transform against a vocabulary of the size ORB-SLAM2 ships (k = 10, six levels, 10^6 words, 45 MB; synthetic complete tree)
    ctx3 = api.Context(width=TM.W, height=TM.H)
    big = B.build_full_vocabulary()
    B.vocab_load(ctx3, big)
    L3, v3 = TB._oracle_voc(big)
    data = np.frombuffer(big, np.uint8, offset=8 + 120).reshape(-1, 408)
    leaves = data[rng.integers(11111, 111111, 1500), 8:8 + 320].reshape(1500, 10, 32)[np.arange(1500), rng.integers(0, 10, 1500)]
    big_d = leaves ^ np.packbits(rng.random((1500, 256)) < 0.02, axis=1, bitorder="little")
    row("fbow transform, 1500 descriptors, full-size vocabulary (10^6 words, 45 MB) [row 17]",
        lambda: B.transform(ctx3, big_d, 4), lambda: TB._oracle_transform(L3, v3, big_d, 4))
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
