"""Reference-signature shims of the SURVEY section-8(f) rows: orbslam2_amd/compat/Optimizer.cc (int Optimizer::PoseOptimization(Frame*),
src/Optimizer.cc:283) and orbslam2_amd/compat/KeyFrameDatabase.{h,cc} (add / erase / clear / DetectRelocalizationCandidates(Frame*) /
DetectLoopCandidates(KeyFrame*, float), src/KeyFrameDatabase.cc:38-70,73,196), so that src/Tracking.cc:875,998,1040,1496 compile unchanged.

CPU: the declarations are the reference's, the shims compile against the stand-ins with -Werror and only forward.
GPU: tests/compat_stub/frame_selftest `backend` calls them on Frame / KeyFrame objects; results against the CPU oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "compat_stub")
COMPAT = os.path.join(ROOT, "orbslam2_amd", "compat")
EXE = os.path.join(STUB, "frame_selftest")
REF_DB_DECLS = [  # include/KeyFrameDatabase.h:42-62 of the reference
    "KeyFrameDatabase(fbow::Vocabulary *voc);", "void add(KeyFrame *pKF);", "void erase(KeyFrame* pKF);", "void clear();",
    "std::vector<KeyFrame *> DetectLoopCandidates(KeyFrame* pKF, float minScore);", "std::vector<KeyFrame*> DetectRelocalizationCandidates(Frame* F);",
    "void SetFBOWvocabulary(fbow::Vocabulary *pfbowv)",
]


def _norm(decl):
    return re.sub(r"\s+", "", decl.replace("std::", ""))


def test_backend_shims_declare_the_reference_signatures_and_compile():
    body = _norm(re.sub(r"//[^\n]*", "", open(os.path.join(COMPAT, "KeyFrameDatabase.h")).read()))
    for d in REF_DB_DECLS:
        assert _norm(d) in body, d
    assert _norm("int PoseOptimization(Frame* pFrame);") in _norm(open(os.path.join(STUB, "Optimizer.h")).read())  # include/Optimizer.h:48 (a member, not static, in this fork)
    assert "int Optimizer::PoseOptimization(Frame *pFrame)" in open(os.path.join(COMPAT, "Optimizer.cc")).read()
    for f, calls in (("Optimizer.cc", ["orbfe_pose_optimization(", "MapPoint::mGlobalMutex", "pFrame->SetPose("]),
                     ("KeyFrameDatabase.cc", ["orbfe_kfdb_add(", "orbfe_kfdb_erase(", "orbfe_kfdb_clear(", "orbfe_detect_reloc_candidates(", "orbfe_detect_loop_candidates(",
                                              "GetBestCovisibilityKeyFrames(10)", "GetConnectedKeyFrames()"])):
        src = os.path.join(COMPAT, f)
        r = subprocess.run(["g++", "-std=c++14", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", STUB, "-I", COMPAT, src], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        text = open(src).read()
        for c in calls:
            assert c in text, (f, c)
        assert "oracle" not in text and "sqrt" not in text  # no scoring / solving of its own


@pytest.mark.gpu
def test_pose_optimization_and_keyframe_database_through_the_reference_signatures(tmp_path):
    from tests import test_pose as TP
    from tests.test_bow import _descs, _oracle_transform, _oracle_voc, _p
    from orbslam2_amd import bow as B
    assert os.path.exists(EXE), "frame_selftest not built (make -C tests/compat_stub)"
    d = tmp_path
    cam = TP.CAM
    W, H = 1241, 376
    # ---- pose scene
    s = TP.scene(909, n=1400)
    T0 = np.eye(4, dtype=np.float32); T0[2, 3] = -0.3
    files = {"cam.f32": np.array([cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["bf"], W, H, 2000, 35.0], np.float32), "dist.f32": np.zeros(4, np.float32),
             "pose_keys.bin": s["keys"], "pose_ur.bin": s["ur"], "pose_xw.bin": s["Xw"], "pose_has.bin": s["has"], "pose_T0.bin": T0}
    # ---- database scene: 120 keyframes of 12 places, covisibility lists of <= 10, six queries, one erase
    vocab = B.build_vocabulary(_descs(1, 6000), k=10, levels=5, seed=7)
    files["vocab.fbow"] = vocab
    L, v = _oracle_voc(vocab)
    L.orc_detect_reloc_candidates.restype = C.c_int
    L.orc_detect_reloc_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_int]
    L.orc_detect_loop_candidates.restype = C.c_int
    L.orc_detect_loop_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 3 + [C.c_int]
    L.orc_bow_score.restype = C.c_double
    L.orc_bow_score.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(23)
    n_kf, n_pl, nq = 120, 12, 6
    places = [_descs(300 + p, 800) for p in range(n_pl)]
    kf_words, kf_w = [], []
    for k in range(n_kf):
        dd = np.concatenate([places[k % n_pl][rng.permutation(800)[:600]], _descs(7000 + k, 200)])
        _, (words, ww), _ = _oracle_transform(L, v, dd)
        kf_words.append(words.copy()); kf_w.append(ww.copy())
    covis = [[(k + n_pl * j) % n_kf for j in (1, -1, 2, -2)] + [int(x) for x in rng.integers(0, n_kf, 5)] for k in range(n_kf)]  # 9 <= 10
    covis_off = np.zeros(n_kf + 1, np.int32); covis_off[1:] = np.cumsum([len(c) for c in covis])
    covis_idx = np.concatenate(covis).astype(np.int32)
    q_words, q_w, connected, min_score, erase_after = [], [], np.zeros((nq, n_kf), np.uint8), np.zeros(nq, np.float32), np.full(nq, -1, np.int32)
    q_place = []
    for q in range(nq):
        place = int(rng.integers(0, n_pl)); q_place.append(place)
        qd = np.concatenate([places[place][rng.permutation(800)[:560]], _descs(9000 + q, 240)])
        _, (qw, qv), _ = _oracle_transform(L, v, qd)
        q_words.append(qw.copy()); q_w.append(qv.copy())
        connected[q, [k for k in range(n_kf) if k % n_pl == place and rng.random() < 0.33]] = 1
    # expectations, query by query (the erase changes the database for the later ones)
    state_ref = np.zeros(n_kf, np.float32)
    reloc_ref, loop_ref, counts_ref, state_all = [], [], [], []
    words_now, w_now = [w.copy() for w in kf_words], [w.copy() for w in kf_w]
    for q in range(nq):
        kf_off = np.zeros(n_kf + 1, np.int32); kf_off[1:] = np.cumsum([len(w) for w in words_now])
        db_words = np.concatenate(words_now); db_w = np.concatenate(w_now)
        qw, qv = q_words[q], q_w[q]
        cand = np.zeros(n_kf, np.int32)
        n1 = L.orc_detect_reloc_candidates(_p(qw), _p(qv), len(qw), n_kf, _p(kf_off), _p(db_words), _p(db_w), _p(covis_off), _p(covis_idx), _p(state_ref), _p(cand), n_kf)
        reloc_ref += cand[:n1].tolist(); state_all.append(state_ref.copy())
        sc = np.array([L.orc_bow_score(_p(qw), _p(qv), len(qw), _p(words_now[k]), _p(w_now[k]), len(words_now[k])) if len(words_now[k]) else -1.0 for k in range(n_kf)], np.float32)
        min_score[q] = np.sort(sc[sc >= 0])[int(0.85 * (sc >= 0).sum())]
        cl = np.zeros(n_kf, np.int32)
        n2 = L.orc_detect_loop_candidates(_p(qw), _p(qv), len(qw), n_kf, _p(kf_off), _p(db_words), _p(db_w), _p(connected[q]), float(min_score[q]), _p(covis_off), _p(covis_idx),
                                          _p(cl), n_kf)
        loop_ref += cl[:n2].tolist(); counts_ref += [n1, n2]
        assert n1 >= 1 and n2 >= 1
        if q == 2:
            erase_after[q] = int(cand[0])
            words_now[cand[0]] = words_now[cand[0]][:0]; w_now[cand[0]] = w_now[cand[0]][:0]
    kf_off0 = np.zeros(n_kf + 1, np.int32); kf_off0[1:] = np.cumsum([len(w) for w in kf_words])
    q_off = np.zeros(nq + 1, np.int32); q_off[1:] = np.cumsum([len(w) for w in q_words])
    files.update({"db_kf_off.bin": kf_off0, "db_kf_words.bin": np.concatenate(kf_words), "db_kf_w.bin": np.concatenate(kf_w), "db_covis_off.bin": covis_off,
                  "db_covis_idx.bin": covis_idx, "db_q_off.bin": q_off, "db_q_words.bin": np.concatenate(q_words), "db_q_w.bin": np.concatenate(q_w),
                  "db_connected.bin": connected, "db_min_score.bin": min_score, "db_erase_after.bin": erase_after})
    for name, a in files.items():
        if isinstance(a, bytes):
            (d / name).write_bytes(a)
        else:
            np.ascontiguousarray(a).tofile(d / name)
    r = subprocess.run([EXE, str(d), "backend"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "frame selftest ok" in r.stdout, r.stdout + r.stderr
    # ---- PoseOptimization(Frame*) == oracle (tolerance and reasons: tests/test_pose.py)
    Tref, out_ref, n_ref = O.pose_optimization(T0, s["keys"], s["ur"], s["has"], s["Xw"], TP.INV_SIGMA2, cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["bf"])
    T = np.fromfile(d / "pose_T.bin", np.float32).reshape(4, 4)
    assert np.abs(T - Tref).max() <= TP.POSE_ATOL and np.abs(T - s["T"]).max() < 5e-3
    has = s["has"] > 0
    assert np.array_equal(np.fromfile(d / "pose_outlier.bin", np.uint8)[has], out_ref[has]) and not np.fromfile(d / "pose_outlier.bin", np.uint8)[~has].any()
    assert np.fromfile(d / "pose_ninl.bin", np.int32)[0] == n_ref and n_ref > 500
    Ow = np.fromfile(d / "pose_Ow.bin", np.float32)
    assert np.abs(Ow - (-T[:3, :3].T.astype(np.float64) @ T[:3, 3].astype(np.float64))).max() < 1e-6  # SetPose -> UpdatePoseMatrices
    # ---- KeyFrameDatabase
    assert np.fromfile(d / "db_counts.bin", np.int32).tolist() == counts_ref
    assert np.fromfile(d / "db_reloc.bin", np.int32).tolist() == reloc_ref
    assert np.fromfile(d / "db_loop.bin", np.int32).tolist() == loop_ref
    assert np.array_equal(np.fromfile(d / "db_state.bin", np.float32).reshape(nq, n_kf), np.stack(state_all))  # KeyFrame::mRelocScore after every query
    L.orc_vocab_destroy(v)
