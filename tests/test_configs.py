"""BASELINE.json configs 3 and 5 as end-to-end parity cases: the GPU stages feed each other (image -> keypoints ->
BoW / depth -> database / matcher) and the final result equals the oracle chain run on the same images.

  config 3: Mono-EuRoC 752x480, 1200 features: extract + BoW transform + keyframe database query + SearchByFboW
  config 5: RealSense-D435i RGB-D 1280x720, 2500 features: extract + ComputeStereoFromRGBD + SearchByProjection(last frame)
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from orbslam2_amd import synth

pytestmark = pytest.mark.gpu


def test_config3_euroc_bow_relocalisation():
    from orbslam2_amd import api, bow as B
    from tests import test_bow as TB
    W, H, NF = 752, 480, 1200
    ctx = api.Context(width=W, height=H, nfeatures=NF, fx=458.654, fy=457.296, cx=367.215, cy=248.375, bf=47.9)
    ex = O.Extractor(nfeatures=NF)
    n_kf = 24
    imgs = [synth.mono_image(W, H, seed=900 + i) for i in range(n_kf)]
    # keyframes: GPU extraction (bit-equal to the oracle's, tests/test_gpu_parity.py) -> descriptors
    kf_kd = [ctx.extract(im) for im in imgs]
    train = np.concatenate([d for _, d in kf_kd[:8]])
    blob = B.build_vocabulary(train, k=10, levels=4, seed=3)
    B.vocab_load(ctx, blob)
    L, v = TB._oracle_voc(blob)
    L.orc_detect_reloc_candidates.restype = C.c_int
    L.orc_detect_reloc_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_int]
    db = B.KeyFrameDB(ctx)
    kf_bow, kf_fv = [], []
    for _, d in kf_kd:
        w, wt, nd = B.transform(ctx, d, 4)
        words, ww, nodes, off, feat = B.maps(w, wt, nd)
        db.add(words, ww)
        kf_bow.append((words, ww)); kf_fv.append((nodes, off, feat))
    # query = keyframe 5's view with fresh sensor noise
    rng = np.random.default_rng(1)
    q_img = np.clip(imgs[5].astype(np.int16) + rng.normal(0, 2.0, imgs[5].shape).round().astype(np.int16), 0, 255).astype(np.uint8)
    qk, qd = ctx.extract(q_img)
    qk_ref, qd_ref = ex.extract(q_img)
    assert np.array_equal(qk, qk_ref.astype(api.KP_DTYPE)) and np.array_equal(qd, qd_ref)
    gw, gwt, gnd = B.transform(ctx, qd, 4)
    (w, wt, nd), (q_words, q_ww), q_fv = TB._oracle_transform(L, v, qd_ref)
    assert np.array_equal(gw, w) and np.array_equal(gwt, wt) and np.array_equal(gnd, nd)
    # database query
    covis_off = np.arange(n_kf + 1, dtype=np.int32) * 2
    covis_idx = np.stack([(np.arange(n_kf) + 1) % n_kf, (np.arange(n_kf) - 1) % n_kf], axis=1).astype(np.int32).ravel()
    st_gpu = np.zeros(n_kf, np.float32); st_ref = np.zeros(n_kf, np.float32)
    got = db.detect_reloc_candidates(q_words, q_ww, covis_off, covis_idx, st_gpu)
    kf_off = np.zeros(n_kf + 1, np.int32); kf_off[1:] = np.cumsum([len(a) for a, _ in kf_bow])
    dbw = np.concatenate([a for a, _ in kf_bow]); dbv = np.concatenate([b for _, b in kf_bow])
    cand = np.zeros(n_kf, np.int32)
    n = L.orc_detect_reloc_candidates(TB._p(q_words), TB._p(q_ww), len(q_words), n_kf, TB._p(kf_off), TB._p(dbw), TB._p(dbv),
                                      TB._p(covis_off), TB._p(covis_idx), TB._p(st_ref), TB._p(cand), n_kf)
    assert got.tolist() == cand[:n].tolist() and 5 in got.tolist()
    # SearchByFboW(candidate keyframe, query frame)
    kfi = 5
    kfk, kfd = kf_kd[kfi]
    kf_valid = np.ones(len(kfd), np.int32)
    ref = np.zeros(len(qd), np.int32)
    fv = kf_fv[kfi]
    nref = L.orc_search_by_bow(TB._p(fv[0]), TB._p(fv[1]), TB._p(fv[2]), len(fv[0]), TB._p(kf_valid), TB._p(kfd), TB._p(kfk["angle"].copy()),
                               TB._p(q_fv[0]), TB._p(q_fv[1]), TB._p(q_fv[2]), len(q_fv[0]), TB._p(qd_ref), TB._p(qk_ref["angle"].copy()),
                               len(qd_ref), 0.75, 1, TB._p(ref))
    m, nm = B.search_by_bow(ctx, fv, kf_valid, kfd, kfk["angle"].copy(), q_fv, qd, qk["angle"].copy(), 0.75, True)
    assert nm == nref and np.array_equal(m, ref) and nref > 300  # most of the view is re-found
    L.orc_vocab_destroy(v)
    ctx.close()


def test_config5_d435i_rgbd_tracking():
    from orbslam2_amd import api
    W, H, NF = 1280, 720, 2500
    fx = fy = 911.0; cx, cy, bf = 640.0, 360.0, 45.5
    ctx = api.Context(width=W, height=H, nfeatures=NF, fx=fx, fy=fy, cx=cx, cy=cy, bf=bf)
    ex = O.Extractor(nfeatures=NF)
    img1, img2, depth = synth.stereo_pair(W, H, seed=321, with_depth=True, bf=bf)  # img2: the same scene from 5 cm to the right
    f1 = ctx.rgbd_frame(img1, depth)
    k1, d1 = ex.extract(img1)
    ur1, dp1 = O.stereo_from_rgbd(k1, k1, depth, bf)
    assert np.array_equal(f1["kps"], k1.astype(api.KP_DTYPE)) and np.array_equal(f1["depth"], dp1) and np.array_equal(f1["u_right"], ur1)
    # last frame's map points = back-projections of its keypoints with depth (Frame::UnprojectStereo), pose = identity
    z = f1["depth"]
    valid = (z > 0).astype(np.int32)
    pos = np.stack([(f1["kps"]["x"] - cx) * z / fx, (f1["kps"]["y"] - cy) * z / fy, z], axis=1).astype(np.float32)
    obs = np.ones(len(z), np.int32)
    T_last = np.concatenate([np.eye(3), np.zeros((3, 1))], axis=1).astype(np.float32)
    T_cur = T_last.copy(); T_cur[0, 3] = -bf / fx  # camera moved one baseline to the right
    k2, d2 = ctx.extract(img2)
    k2r, d2r = ex.extract(img2)
    assert np.array_equal(k2, k2r.astype(api.KP_DTYPE)) and np.array_equal(d2, d2r)
    bounds = (0.0, float(W), 0.0, float(H))
    cam = O.Camera(fx, fy, cx, cy, bf, bf / fx)
    g = O.Grid(k2r, *bounds)
    for th, mono in ((7.0, True), (15.0, True)):
        ref, nref = O.search_by_projection_last(g, None, d2r, ex.scale_factors(), cam, T_cur, T_last, pos, d1, valid, obs,
                                                k1["octave"].copy(), k1["angle"].copy(), None, th, mono, True)
        view = ctx._view(k2, None, d2, bounds)
        got, ngot = ctx.search_by_projection_last(view, T_cur, T_last, pos, f1["desc"], valid, obs, f1["kps"]["octave"].copy(),
                                                  f1["kps"]["angle"].copy(), None, th, mono, True)
        assert ngot == nref and np.array_equal(got, ref)
    assert nref > 200
    ctx.close()
