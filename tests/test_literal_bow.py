"""C oracle == literal Python transcription of fbow's transform / score and of SearchByFboW (oracle/literal_bow.py): the vocabulary
is walked through fbow's own block offsets, fBow / fBow2 are dicts in key order, the matcher is the reference's two-pointer merge.
CPU only, small cases."""
import ctypes as C

import numpy as np
import pytest

from oracle import literal_bow as LB
from tests.test_bow import _descs, _oracle_transform, _oracle_voc, _p
from orbslam2_amd import bow as B


@pytest.fixture(scope="module")
def vocab():
    return B.build_vocabulary(_descs(1, 3000), k=10, levels=5, seed=7)


def _maps_of(words, ww, fv):
    nodes, off, feat = fv
    r1 = {int(w): np.float32(x) for w, x in zip(words, ww)}
    r2 = {int(n): feat[off[i]:off[i + 1]].tolist() for i, n in enumerate(nodes)}
    return r1, r2


@pytest.mark.parametrize("level", [4, 2, 9])
def test_transform_literal_vs_oracle(vocab, level):
    L, v = _oracle_voc(vocab)
    voc = LB.Vocabulary(vocab)
    assert voc.m_k == L.orc_vocab_k(v) and voc.desc_size == 32
    d = _descs(2, 500)
    _, (words, ww), fv = _oracle_transform(L, v, d, level)
    r1o, r2o = _maps_of(words, ww, fv)
    r1, r2 = LB.transform(voc, d, level)
    assert sorted(r1) == sorted(r1o) and sorted(r2) == sorted(r2o)
    assert all(np.float32(r1[k]).view(np.uint32) == np.float32(r1o[k]).view(np.uint32) for k in r1)  # float accumulation order
    assert all(r2[k] == r2o[k] for k in r2)
    L.orc_vocab_destroy(v)


def test_score_literal_vs_oracle(vocab):
    L, v = _oracle_voc(vocab)
    L.orc_bow_score.restype = C.c_double
    L.orc_bow_score.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    voc = LB.Vocabulary(vocab)
    a, _ = LB.transform(voc, _descs(3, 400), 4)
    for seed in (4, 5, 6):
        b, _ = LB.transform(voc, _descs(seed, 300), 4)
        wa = np.array(sorted(a), np.uint32); va = np.array([a[k] for k in sorted(a)], np.float32)
        wb = np.array(sorted(b), np.uint32); vb = np.array([b[k] for k in sorted(b)], np.float32)
        ref = L.orc_bow_score(_p(wa), _p(va), len(wa), _p(wb), _p(vb), len(wb))
        assert LB.score(a, b) == ref == 1.0  # un-normalised vectors of the four-argument transform saturate the score
        # the same vectors scaled to unit-ish length: the 1 - sqrt(1 - s) branch
        fa = np.float32(1.0 / np.sqrt(float((va.astype(np.float64) ** 2).sum()))); fb = np.float32(1.0 / np.sqrt(float((vb.astype(np.float64) ** 2).sum())))
        va2, vb2 = (va * fa).astype(np.float32), (vb * fb).astype(np.float32)
        a2 = {int(k): x for k, x in zip(wa, va2)}; b2 = {int(k): x for k, x in zip(wb, vb2)}
        ref2 = L.orc_bow_score(_p(wa), _p(va2), len(wa), _p(wb), _p(vb2), len(wb))
        assert LB.score(a2, b2) == ref2 and 0.0 < ref2 < 1.0
    L.orc_vocab_destroy(v)


@pytest.mark.parametrize("ratio,ori", [(0.7, True), (0.75, False), (0.95, True)])
def test_search_by_fbow_literal_vs_oracle(vocab, ratio, ori):
    L, v = _oracle_voc(vocab)
    voc = LB.Vocabulary(vocab)
    rng = np.random.default_rng(5)
    kf_d = _descs(4, 600)
    perm = rng.permutation(600)[:450]
    f_d = np.concatenate([_descs(6, 0, base=kf_d[perm], flip=0.04), _descs(7, 200)])
    _, _, kf_fv = _oracle_transform(L, v, kf_d)
    _, _, f_fv = _oracle_transform(L, v, f_d)
    kf_valid = (rng.random(len(kf_d)) < 0.8).astype(np.int32)
    kf_ang = rng.uniform(0, 360, len(kf_d)).astype(np.float32)
    f_ang = np.concatenate([(kf_ang[perm] + rng.normal(0, 5, 450)) % 360, rng.uniform(0, 360, 200)]).astype(np.float32)
    ref = np.zeros(len(f_d), np.int32)
    nref = L.orc_search_by_bow(_p(kf_fv[0]), _p(kf_fv[1]), _p(kf_fv[2]), len(kf_fv[0]), _p(kf_valid), _p(kf_d), _p(kf_ang),
                               _p(f_fv[0]), _p(f_fv[1]), _p(f_fv[2]), len(f_fv[0]), _p(f_d), _p(f_ang), len(f_d), ratio, int(ori), _p(ref))
    _, kf_r2 = LB.transform(voc, kf_d, 4)
    _, f_r2 = LB.transform(voc, f_d, 4)
    got, ngot = LB.search_by_fbow_kf_frame(kf_r2, kf_valid, kf_d, kf_ang, f_r2, f_d, f_ang, len(f_d), ratio, ori)
    assert ngot == nref and np.array_equal(got, ref) and nref > 100
    L.orc_vocab_destroy(v)


@pytest.mark.parametrize("ratio,ori", [(0.75, True), (0.8, False), (0.95, True)])
def test_search_by_fbow_kf_kf_literal_vs_oracle(vocab, ratio, ori):
    L, v = _oracle_voc(vocab)
    voc = LB.Vocabulary(vocab)
    rng = np.random.default_rng(9)
    d1 = _descs(14, 600)
    perm = rng.permutation(600)[:450]
    d2 = np.concatenate([_descs(16, 0, base=d1[perm], flip=0.04), _descs(17, 200)])
    _, _, fv1 = _oracle_transform(L, v, d1)
    _, _, fv2 = _oracle_transform(L, v, d2)
    good1 = (rng.random(len(d1)) < 0.8).astype(np.int32); good2 = (rng.random(len(d2)) < 0.85).astype(np.int32)
    a1 = rng.uniform(0, 360, len(d1)).astype(np.float32)
    a2 = np.concatenate([(a1[perm] + rng.normal(0, 5, 450)) % 360, rng.uniform(0, 360, 200)]).astype(np.float32)
    ref = np.zeros(len(d1), np.int32)
    nref = L.orc_search_by_bow_kf(_p(fv1[0]), _p(fv1[1]), _p(fv1[2]), len(fv1[0]), _p(good1), _p(d1), _p(a1), len(d1),
                                  _p(fv2[0]), _p(fv2[1]), _p(fv2[2]), len(fv2[0]), _p(good2), _p(d2), _p(a2), len(d2), ratio, int(ori), _p(ref))
    _, r2a = LB.transform(voc, d1, 4)
    _, r2b = LB.transform(voc, d2, 4)
    got, ngot = LB.search_by_fbow_kf_kf(r2a, good1, d1, a1, len(d1), r2b, good2, d2, a2, ratio, ori)
    assert ngot == nref and np.array_equal(got, ref) and nref > 80
    L.orc_vocab_destroy(v)


@pytest.mark.parametrize("only_stereo,ori", [(False, True), (True, True), (False, False)])
def test_search_for_triangulation_literal_vs_oracle(vocab, only_stereo, ori):
    from oracle import oracle as O
    L, v = _oracle_voc(vocab)
    L.orc_search_for_triangulation.restype = C.c_int
    L.orc_search_for_triangulation.argtypes = ([C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int]) * 2 + [C.c_void_p] * 3 + \
        [C.c_float] * 4 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    voc = LB.Vocabulary(vocab)
    fx = fy = 500.0; cx, cy = 320.0, 240.0
    ex = O.Extractor()
    sf = np.ascontiguousarray(ex.scale_factors(), np.float32); s2 = np.ascontiguousarray(ex.sigma2(), np.float32)
    rng = np.random.default_rng(21)
    n = 500
    P = np.stack([rng.uniform(-6, 6, n), rng.uniform(-4, 4, n), rng.uniform(4, 30, n)], axis=1)
    T1 = np.concatenate([np.eye(3), np.zeros((3, 1))], axis=1)
    a = np.deg2rad(3.0)
    R2 = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    T2 = np.concatenate([R2, np.array([[-0.4], [0.02], [0.05]])], axis=1)

    def view(T, seed, desc_base):
        r = np.random.default_rng(seed)
        pc = (T[:, :3] @ P.T).T + T[:, 3]
        k = np.zeros(n, O.KP_DTYPE)
        k["x"] = fx * pc[:, 0] / pc[:, 2] + cx + r.normal(0, 0.4, n); k["y"] = fy * pc[:, 1] / pc[:, 2] + cy + r.normal(0, 0.4, n)
        k["octave"] = r.integers(0, 8, n); k["angle"] = (desc_base[1] + r.normal(0, 4, n)) % 360; k["size"] = 31; k["class_id"] = -1
        d = desc_base[0] ^ np.packbits(r.random((n, 256)) < 0.03, axis=1, bitorder="little")
        ur = np.where(r.random(n) < 0.5, k["x"] - 40.0 / pc[:, 2], -1.0).astype(np.float32)
        has_mp = (r.random(n) < 0.3).astype(np.uint8)
        return k, d, ur, has_mp

    base = (_descs(9, n), rng.uniform(0, 360, n))
    k1, d1, ur1, mp1 = view(T1, 1, base)
    k2, d2, ur2, mp2 = view(T2, 2, base)
    _, _, fv1 = _oracle_transform(L, v, d1)
    _, _, fv2 = _oracle_transform(L, v, d2)
    R12 = T1[:, :3] @ T2[:, :3].T
    t12 = -R12 @ T2[:, 3] + T1[:, 3]
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    F12 = (np.linalg.inv(K).T @ tx @ R12 @ np.linalg.inv(K)).astype(np.float32)
    Cw1 = (-T1[:, :3].T @ T1[:, 3]).astype(np.float32)
    T2f = T2.astype(np.float32)
    ref = np.zeros(n, np.int32)
    nref = L.orc_search_for_triangulation(_p(fv1[0]), _p(fv1[1]), _p(fv1[2]), len(fv1[0]), _p(k1), _p(ur1), _p(mp1), _p(d1), n,
                                          _p(fv2[0]), _p(fv2[1]), _p(fv2[2]), len(fv2[0]), _p(k2), _p(ur2), _p(mp2), _p(d2), n,
                                          _p(F12), _p(Cw1), _p(T2f), fx, fy, cx, cy, _p(sf), _p(s2), int(only_stereo), int(ori), _p(ref))
    _, r2a = LB.transform(voc, d1, 4)
    _, r2b = LB.transform(voc, d2, 4)
    got, ngot = LB.search_for_triangulation(r2a, k1, ur1, mp1, d1, r2b, k2, ur2, mp2, d2, F12, Cw1, T2f, fx, fy, cx, cy, sf, s2, only_stereo, ori)
    assert ngot == nref and np.array_equal(got, ref) and nref > (20 if only_stereo else 60)
    L.orc_vocab_destroy(v)


def _random_db(seed, n_kf=60, n_words=400):
    """Keyframes with sparse unit-length BoW vectors over a small word set (so scores spread over (0, 1)), covisibility lists."""
    rng = np.random.default_rng(seed)
    protos = []
    for _ in range(8):  # places: keyframes of one place share most of their words
        nw = int(rng.integers(30, 60))
        protos.append((np.sort(rng.choice(n_words, nw, replace=False)), rng.random(nw) + 0.05))
    kfs = []
    for k in range(n_kf):
        pw, pv = protos[int(rng.integers(0, len(protos)))]
        keep = rng.random(len(pw)) < 0.8
        extra = np.setdiff1d(rng.choice(n_words, 8, replace=False), pw[keep])
        words = np.concatenate([pw[keep], extra]); w = np.concatenate([pv[keep] * rng.uniform(0.7, 1.3, int(keep.sum())), rng.random(len(extra)) * 0.3 + 0.02])
        order = np.argsort(words)
        words = words[order].astype(np.uint32); w = w[order].astype(np.float32)
        w = (w / np.float32(np.sqrt(float((w.astype(np.float64) ** 2).sum())))).astype(np.float32)
        kfs.append((words, w))
    covis = [rng.choice(n_kf, int(rng.integers(0, 11)), replace=False).astype(np.int32) for _ in range(n_kf)]
    covis = [c[c != k] for k, c in enumerate(covis)]
    return rng, kfs, covis


def _csr(kfs, covis):
    kf_off = np.zeros(len(kfs) + 1, np.int32); kf_off[1:] = np.cumsum([len(a) for a, _ in kfs])
    dbw = np.concatenate([a for a, _ in kfs]).astype(np.uint32); dbv = np.concatenate([b for _, b in kfs]).astype(np.float32)
    c_off = np.zeros(len(kfs) + 1, np.int32); c_off[1:] = np.cumsum([len(c) for c in covis])
    c_idx = np.concatenate(covis + [np.zeros(0, np.int32)]).astype(np.int32)
    if len(c_idx) == 0:
        c_idx = np.zeros(1, np.int32)
    return kf_off, dbw, dbv, c_off, c_idx


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_keyframe_database_literal_vs_oracle(seed):
    from oracle import oracle as O
    L = O.lib()
    L.orc_detect_reloc_candidates.restype = C.c_int
    L.orc_detect_reloc_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_int]
    L.orc_detect_loop_candidates.restype = C.c_int
    L.orc_detect_loop_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 3 + [C.c_int]
    rng, kfs, covis = _random_db(seed)
    n = len(kfs)
    kf_off, dbw, dbv, c_off, c_idx = _csr(kfs, covis)
    db = LB.KeyFrameDatabase()
    lk = [LB.LKeyFrame(k, {int(w): np.float32(x) for w, x in zip(*kfs[k])}) for k in range(n)]
    for k in range(n):
        lk[k].best_covisibles = [lk[int(j)] for j in covis[k]]
        db.add(lk[k])
    state = np.zeros(n, np.float32)  # mRelocScore of every keyframe: persists across queries in both statements
    nonempty = 0
    for q in range(6):  # relocalisation queries (frames): noisy copies of a keyframe's vector
        src = int(rng.integers(0, n))
        words, w = kfs[src]
        keep = rng.random(len(words)) < 0.8
        qw = words[keep]; qv = (w[keep] * np.float32(rng.uniform(0.8, 1.1))).astype(np.float32)
        cand = np.zeros(n, np.int32)
        m = L.orc_detect_reloc_candidates(_p(qw), _p(qv), len(qw), n, _p(kf_off), _p(dbw), _p(dbv), _p(c_off), _p(c_idx), _p(state), _p(cand), n)
        got = db.DetectRelocalizationCandidates(1000 + q, {int(a): np.float32(b) for a, b in zip(qw, qv)})
        assert [kf.mnId for kf in got] == cand[:m].tolist()
        assert all(lk[k].mRelocScore == state[k] for k in range(n))
        nonempty += m > 0
    assert nonempty >= 4
    nonempty = 0
    for q in range(6):  # loop queries (keyframes of the database, with their connected sets)
        src = int(rng.integers(0, n))
        connected = (rng.random(n) < 0.15).astype(np.uint8); connected[src] = 1
        lk[src].connected = {lk[k] for k in range(n) if connected[k]}
        qw, qv = kfs[src]
        min_score = float(rng.uniform(0.0, 0.15))
        cand = np.zeros(n, np.int32)
        m = L.orc_detect_loop_candidates(_p(qw), _p(qv), len(qw), n, _p(kf_off), _p(dbw), _p(dbv), _p(connected), min_score, _p(c_off), _p(c_idx), _p(cand), n)
        lk[src].mnId = 5000 + q + seed * 10  # a fresh query id (mnLoopQuery compares ids)
        got = db.DetectLoopCandidates(lk[src], min_score)
        got_idx = [lk.index(kf) for kf in got]
        assert got_idx == cand[:m].tolist()
        nonempty += m > 0
    assert nonempty >= 3
