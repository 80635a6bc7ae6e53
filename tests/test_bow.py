"""Bag of words: fbow transform + SearchByFboW.  The vocabulary file is missing from the
reference tree (.MISSING_LARGE_BLOBS), so a small one is synthesised (k-majority tree) in the fbow format."""
import ctypes as C
import struct

import numpy as np
import pytest

from oracle import oracle as O
from orbslam2_amd import bow as B


def _descs(seed, n, base=None, flip=0.0):
    rng = np.random.default_rng(seed)
    if base is None:
        protos = rng.integers(0, 256, (40, 32)).astype(np.uint8)
        d = protos[rng.integers(0, 40, n)]
        d = d ^ np.packbits(rng.random((n, 256)) < 0.12, axis=1, bitorder="little")
        return d
    return base ^ np.packbits(rng.random((len(base), 256)) < flip, axis=1, bitorder="little")


@pytest.fixture(scope="module")
def vocab():
    train = _descs(1, 6000)
    return B.build_vocabulary(train, k=10, levels=5, seed=7)


def _oracle_voc(blob):
    L = O.lib()
    L.orc_vocab_from_blob.restype = C.c_void_p; L.orc_vocab_from_blob.argtypes = [C.c_void_p, C.c_size_t]
    L.orc_vocab_destroy.argtypes = [C.c_void_p]; L.orc_vocab_destroy.restype = None
    L.orc_vocab_k.argtypes = [C.c_void_p]; L.orc_vocab_k.restype = C.c_int
    L.orc_bow_descend.restype = None
    L.orc_bow_descend.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_bow_maps.restype = C.c_int
    L.orc_bow_maps.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 5 + [C.POINTER(C.c_int)]
    L.orc_search_by_bow_kf.restype = C.c_int
    L.orc_search_by_bow_kf.argtypes = ([C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int]) * 2 + [C.c_float, C.c_int, C.c_void_p]
    L.orc_search_by_bow.restype = C.c_int
    L.orc_search_by_bow.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 2 + \
        [C.c_int, C.c_float, C.c_int, C.c_void_p]
    buf = np.frombuffer(blob, np.uint8)
    v = L.orc_vocab_from_blob(buf.ctypes.data_as(C.c_void_p), len(buf))
    assert v
    return L, v


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _oracle_transform(L, v, d, level=4):
    n = len(d)
    d = np.ascontiguousarray(d)
    w = np.zeros(n, np.uint32); wt = np.zeros(n, np.float32); nd = np.zeros(n, np.uint32)
    L.orc_bow_descend(v, _p(d), n, level, _p(w), _p(wt), _p(nd))
    words = np.zeros(n, np.uint32); ww = np.zeros(n, np.float32); nodes = np.zeros(n, np.uint32)
    off = np.zeros(n + 1, np.int32); feat = np.zeros(n, np.int32); nn = C.c_int()
    nw = L.orc_bow_maps(_p(w), _p(wt), _p(nd), n, _p(words), _p(ww), _p(nodes), _p(off), _p(feat), C.byref(nn))
    return (w, wt, nd), (words[:nw], ww[:nw]), (nodes[: nn.value], off[: nn.value + 1], feat)


def test_vocabulary_format_and_oracle_descend(vocab):
    sig, = struct.unpack_from("<Q", vocab, 0)
    assert sig == 55824124 and len(vocab) > 128
    L, v = _oracle_voc(vocab)
    assert L.orc_vocab_k(v) == 10
    d = _descs(2, 800)
    (w, wt, nd), (words, ww), (nodes, off, feat) = _oracle_transform(L, v, d)
    assert (wt > 0).all() and len(words) > 50
    # weights of a word: n additions of the same leaf weight in float
    for i in (0, len(words) // 2, len(words) - 1):
        m = w == words[i]
        acc = np.float32(0)
        for x in wt[m]:
            acc = np.float32(acc + x)
        assert ww[i] == acc and len(set(wt[m].tolist())) == 1
    assert off[-1] == len(d) and sorted(feat.tolist()) == list(range(len(d)))
    for kk in range(len(nodes)):
        seg = feat[off[kk]:off[kk + 1]]
        assert (np.diff(seg) > 0).all() and (nd[seg] == nodes[kk]).all()
    # identical descriptors land in the same word / node; node ids are 4 bits per level
    d2 = np.concatenate([d[:5], d[:5]])
    (w2, _, nd2), _, _ = _oracle_transform(L, v, d2)
    assert np.array_equal(w2[:5], w2[5:]) and np.array_equal(nd2[:5], nd2[5:]) and (nd2 < 16 ** 4).all()
    L.orc_vocab_destroy(v)


@pytest.mark.gpu
def test_gpu_bow_transform_and_search(vocab):
    from orbslam2_amd import api
    ctx = api.Context(width=640, height=480)
    with pytest.raises(api.OrbfeError):
        B.transform(ctx, _descs(3, 10))  # no vocabulary yet
    with pytest.raises(api.OrbfeError):
        B.vocab_load(ctx, b"\x00" * 200)  # bad signature
    B.vocab_load(ctx, vocab)
    L, v = _oracle_voc(vocab)
    kf_d = _descs(4, 1500)
    rng = np.random.default_rng(5)
    perm = rng.permutation(1500)[:1200]
    f_d = np.concatenate([_descs(6, 0, base=kf_d[perm], flip=0.04), _descs(7, 400)])
    for level in (4, 2, 9):
        (w, wt, nd), (words, ww), fv = _oracle_transform(L, v, kf_d, level)
        gw, gwt, gnd = B.transform(ctx, kf_d, level)
        assert np.array_equal(gw, w) and np.array_equal(gwt, wt) and np.array_equal(gnd, nd)
        mw, mww, mn, mo, mf = B.maps(gw, gwt, gnd)
        assert np.array_equal(mw, words) and np.array_equal(mww, ww)
        assert np.array_equal(mn, fv[0]) and np.array_equal(mo, fv[1]) and np.array_equal(mf, fv[2])
    with pytest.raises(api.OrbfeError):
        B.transform(ctx, kf_d[:0])  # fbow: "No input data"
    _, _, kf_fv = _oracle_transform(L, v, kf_d)
    _, _, f_fv = _oracle_transform(L, v, f_d)
    kf_valid = (rng.random(len(kf_d)) < 0.8).astype(np.int32)
    kf_ang = rng.uniform(0, 360, len(kf_d)).astype(np.float32)
    f_ang = np.concatenate([(kf_ang[perm] + rng.normal(0, 5, 1200)) % 360, rng.uniform(0, 360, 400)]).astype(np.float32)
    for ratio, ori in ((0.7, True), (0.75, False), (0.95, True)):
        ref = np.zeros(len(f_d), np.int32)
        nref = L.orc_search_by_bow(_p(kf_fv[0]), _p(kf_fv[1]), _p(kf_fv[2]), len(kf_fv[0]), _p(kf_valid), _p(kf_d), _p(kf_ang),
                                   _p(f_fv[0]), _p(f_fv[1]), _p(f_fv[2]), len(f_fv[0]), _p(f_d), _p(f_ang), len(f_d), ratio, int(ori), _p(ref))
        got, ngot = B.search_by_bow(ctx, kf_fv, kf_valid, kf_d, kf_ang, f_fv, f_d, f_ang, ratio, ori)
        assert ngot == nref and np.array_equal(got, ref)
    assert nref > 300
    # SearchByFboW(KeyFrame*, KeyFrame*, ...) (src/ORBmatcher.cc:517-650): both sides need a good map point, result by KF1 keypoint
    f_valid = (rng.random(len(f_d)) < 0.85).astype(np.int32)
    for ratio, ori in ((0.75, True), (0.8, False), (0.95, True)):
        ref12 = np.zeros(len(kf_d), np.int32)
        n12 = L.orc_search_by_bow_kf(_p(kf_fv[0]), _p(kf_fv[1]), _p(kf_fv[2]), len(kf_fv[0]), _p(kf_valid), _p(kf_d), _p(kf_ang), len(kf_d),
                                     _p(f_fv[0]), _p(f_fv[1]), _p(f_fv[2]), len(f_fv[0]), _p(f_valid), _p(f_d), _p(f_ang), len(f_d),
                                     ratio, int(ori), _p(ref12))
        got12, ng12 = B.search_by_bow_kf(ctx, kf_fv, kf_valid, kf_d, kf_ang, f_fv, f_valid, f_d, f_ang, ratio, ori)
        assert ng12 == n12 and np.array_equal(got12, ref12)
        assert n12 == int((ref12 >= 0).sum()) and n12 > 200
        m = ref12[ref12 >= 0]
        assert len(np.unique(m)) == len(m) and f_valid[m].all() and kf_valid[ref12 >= 0].all()  # one-to-one, valid on both sides
    ok = ref >= 0
    assert np.mean(perm[np.nonzero(ok)[0][np.nonzero(ok)[0] < 1200]] == ref[ok][np.nonzero(ok)[0] < 1200]) > 0.9
    L.orc_vocab_destroy(v)
    ctx.close()


@pytest.mark.gpu
def test_gpu_full_size_vocabulary():
    """The size ORB-SLAM2 ships: k = 10, six levels, 10^6 words, 45 MB (orbslam2_amd.bow.build_full_vocabulary; the real file is not in
    the reference checkout).  transform at the reference's level 4 and SearchByFboW on it == oracle."""
    from orbslam2_amd import api
    blob = B.build_full_vocabulary()
    assert 44e6 < len(blob) < 47e6
    ctx = api.Context(width=640, height=480)
    B.vocab_load(ctx, blob)
    L, v = _oracle_voc(blob)
    assert L.orc_vocab_k(v) == 10
    rng = np.random.default_rng(3)
    # descriptors near random leaves of the tree (so that the descents go deep along real branches), plus pure noise
    data = np.frombuffer(blob, np.uint8, offset=8 + 120).reshape(-1, 408)
    leaves = data[rng.integers(11111, 111111, 1300), 8:8 + 320].reshape(1300, 10, 32)[np.arange(1300), rng.integers(0, 10, 1300)]
    kf_d = np.concatenate([leaves ^ np.packbits(rng.random((1300, 256)) < 0.02, axis=1, bitorder="little"),
                           rng.integers(0, 256, (200, 32)).astype(np.uint8)])
    f_d = np.concatenate([kf_d[:1100] ^ np.packbits(rng.random((1100, 256)) < 0.03, axis=1, bitorder="little"),
                          rng.integers(0, 256, (400, 32)).astype(np.uint8)])
    (w, wt, nd), (words, ww), kf_fv = _oracle_transform(L, v, kf_d, 4)
    gw, gwt, gnd = B.transform(ctx, kf_d, 4)
    assert np.array_equal(gw, w) and np.array_equal(gwt, wt) and np.array_equal(gnd, nd)
    assert len(np.unique(w)) > 1200 and w.max() < 10 ** 6 and (nd < 16 ** 4).all()
    _, _, f_fv = _oracle_transform(L, v, f_d, 4)
    kf_valid = (rng.random(len(kf_d)) < 0.85).astype(np.int32)
    kf_ang = rng.uniform(0, 360, len(kf_d)).astype(np.float32)
    f_ang = np.concatenate([(kf_ang[:1100] + rng.normal(0, 5, 1100)) % 360, rng.uniform(0, 360, 400)]).astype(np.float32)
    ref = np.zeros(len(f_d), np.int32)
    nref = L.orc_search_by_bow(_p(kf_fv[0]), _p(kf_fv[1]), _p(kf_fv[2]), len(kf_fv[0]), _p(kf_valid), _p(kf_d), _p(kf_ang),
                               _p(f_fv[0]), _p(f_fv[1]), _p(f_fv[2]), len(f_fv[0]), _p(f_d), _p(f_ang), len(f_d), 0.75, 1, _p(ref))
    got, ngot = B.search_by_bow(ctx, kf_fv, kf_valid, kf_d, kf_ang, f_fv, f_d, f_ang, 0.75, True)
    assert ngot == nref and np.array_equal(got, ref) and nref > 300
    L.orc_vocab_destroy(v)
    ctx.close()


@pytest.mark.gpu
def test_gpu_keyframe_database_relocalisation(vocab):
    """BASELINE config 3's database side: 500 keyframes' BoW vectors in HBM, fBow::score + DetectRelocalizationCandidates
    (src/KeyFrameDatabase.cc:196-307) == oracle; also erase() and the persistent mRelocScore state (contract Q10)."""
    from orbslam2_amd import api
    L, v = _oracle_voc(vocab)
    L.orc_bow_score.restype = C.c_double
    L.orc_bow_score.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_detect_reloc_candidates.restype = C.c_int
    L.orc_detect_reloc_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_int]
    L.orc_detect_loop_candidates.restype = C.c_int
    L.orc_detect_loop_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 3 + [C.c_int]
    ctx = api.Context(width=752, height=480, nfeatures=1200)
    B.vocab_load(ctx, vocab)
    rng = np.random.default_rng(11)
    n_kf = 500
    places = [_descs(100 + p, 1200) for p in range(25)]  # 25 places, 20 keyframes each share most descriptors
    db = B.KeyFrameDB(ctx)
    kf_words, kf_w = [], []
    for k in range(n_kf):
        base = places[k % 25]
        d = np.concatenate([base[rng.permutation(1200)[:900]], _descs(2000 + k, 300)])  # a revisit re-detects the same corners
        _, (words, ww), _ = _oracle_transform(L, v, d)
        kf_words.append(words.copy()); kf_w.append(ww.copy())
        assert db.add(words, ww) == k
    assert len(db) == n_kf
    kf_off = np.zeros(n_kf + 1, np.int32); kf_off[1:] = np.cumsum([len(w) for w in kf_words])
    db_words = np.concatenate(kf_words); db_w = np.concatenate(kf_w)
    covis = [[(k + 25 * j) % n_kf for j in (1, -1, 2, -2, 3)] + [int(x) for x in rng.integers(0, n_kf, 8)] for k in range(n_kf)]  # 13 > 10 listed
    covis_off = np.zeros(n_kf + 1, np.int32); covis_off[1:] = np.cumsum([len(c) for c in covis])
    covis_idx = np.concatenate(covis).astype(np.int32)
    state_ref = np.zeros(n_kf, np.float32); state_gpu = np.zeros(n_kf, np.float32)
    for q in range(6):
        place = int(rng.integers(0, 25))
        qd = np.concatenate([places[place][rng.permutation(1200)[:850]], _descs(4000 + q, 350)])
        _, (qw, qv), _ = _oracle_transform(L, v, qd)
        common, score = db.score(qw, qv)
        for k in rng.integers(0, n_kf, 25):  # fBow::score, bit for bit (float of a double sum in word order)
            if len(kf_words[k]) == 0:
                continue
            ref = np.float32(L.orc_bow_score(_p(qw), _p(qv), len(qw), _p(kf_words[k]), _p(kf_w[k]), len(kf_words[k])))
            assert score[k] == ref, (q, k, score[k], ref)
            assert common[k] == len(np.intersect1d(qw, kf_words[k]))
        cand_ref = np.zeros(n_kf, np.int32)
        n_ref = L.orc_detect_reloc_candidates(_p(qw), _p(qv), len(qw), n_kf, _p(kf_off), _p(db_words), _p(db_w), _p(covis_off), _p(covis_idx),
                                              _p(state_ref), _p(cand_ref), n_kf)
        got = db.detect_reloc_candidates(qw, qv, covis_off, covis_idx, state_gpu)
        assert got.tolist() == cand_ref[:n_ref].tolist(), q
        assert np.array_equal(state_gpu, state_ref)
        assert n_ref >= 1 and (cand_ref[:n_ref] % 25 == place).mean() > 0.8  # the query's place wins
        # DetectLoopCandidates(pKF, minScore) on the same database: a random third of the keyframes of the query's place are
        # "connected" (excluded), minScore = the score of a mid-ranked keyframe so that the filter bites
        connected = np.zeros(n_kf, np.uint8)
        connected[[k for k in range(n_kf) if k % 25 == place and rng.random() < 0.33]] = 1
        live = np.array([len(w) > 0 for w in kf_words])
        min_score = float(np.sort(score[live])[int(0.9 * live.sum())])
        cl_ref = np.zeros(n_kf, np.int32)
        nl_ref = L.orc_detect_loop_candidates(_p(qw), _p(qv), len(qw), n_kf, _p(kf_off), _p(db_words), _p(db_w), _p(connected), min_score,
                                              _p(covis_off), _p(covis_idx), _p(cl_ref), n_kf)
        gl = db.detect_loop_candidates(qw, qv, connected, min_score, covis_off, covis_idx)
        assert gl.tolist() == cl_ref[:nl_ref].tolist(), q
        assert nl_ref >= 1 and not connected[cl_ref[:nl_ref]].any() and (score[cl_ref[:nl_ref]] >= 0).all()
        none = db.detect_loop_candidates(qw, qv, np.ones(n_kf, np.uint8), 0.0, covis_off, covis_idx)   # everything connected
        assert len(none) == 0
        if q == 2:  # KeyFrameDatabase::erase: the best candidate leaves the database (both sides)
            gone = int(cand_ref[0])
            db.erase(gone)
            kf_words[gone] = kf_words[gone][:0]; kf_w[gone] = kf_w[gone][:0]
            kf_off[1:] = np.cumsum([len(w) for w in kf_words]); db_words = np.concatenate(kf_words); db_w = np.concatenate(kf_w)
    L.orc_vocab_destroy(v)
    ctx.close()


def test_oracle_bow_score_and_reloc_known_answers():
    """fbow::fBow::score and DetectRelocalizationCandidates restatements from first principles (CPU only)."""
    L = O.lib()
    L.orc_bow_score.restype = C.c_double
    L.orc_bow_score.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_detect_reloc_candidates.restype = C.c_int
    L.orc_detect_reloc_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_int]
    w = np.array([3, 7, 9, 20], np.uint32); v = np.array([0.5, 0.5, 0.5, 0.5], np.float32)  # unit L2 norm
    assert L.orc_bow_score(_p(w), _p(v), 4, _p(w), _p(v), 4) == 1.0
    w2 = np.array([1, 8, 10], np.uint32); v2 = np.array([0.6, 0.0, 0.8], np.float32)
    assert L.orc_bow_score(_p(w), _p(v), 4, _p(w2), _p(v2), 3) == 0.0  # no shared word
    w3 = np.array([7, 20, 30], np.uint32); v3 = np.array([0.6, 0.64, 0.48], np.float32)
    dot = float(np.float32(0.5) * np.float32(0.6)) + float(np.float32(0.5) * np.float32(0.64))
    assert L.orc_bow_score(_p(w), _p(v), 4, _p(w3), _p(v3), 3) == 1.0 - np.sqrt(1.0 - dot)
    # database of 4 keyframes; query shares 3 words with kf0, 3 with kf2, 1 with kf1 (filtered: 1 <= int(3 * 0.8) = 2), 0 with kf3
    kfs = [([3, 7, 9], [0.6, 0.6, 0.5]), ([20, 21], [0.9, 0.4]), ([3, 7, 20], [0.5, 0.5, 0.7]), ([50], [1.0])]
    kf_off = np.zeros(5, np.int32); kf_off[1:] = np.cumsum([len(a) for a, _ in kfs])
    dbw = np.concatenate([np.array(a, np.uint32) for a, _ in kfs]); dbv = np.concatenate([np.array(b, np.float32) for _, b in kfs])
    covis_off = np.array([0, 1, 1, 2, 2], np.int32); covis_idx = np.array([1, 0], np.int32)  # kf0 -> [kf1], kf2 -> [kf0]
    state = np.array([0.0, 0.25, 0.0, 0.0], np.float32)  # kf1's stale mRelocScore counts for kf0 (Q10)
    cand = np.zeros(4, np.int32)
    n = L.orc_detect_reloc_candidates(_p(w), _p(v), 4, 4, _p(kf_off), _p(dbw), _p(dbv), _p(covis_off), _p(covis_idx), _p(state), _p(cand), 4)
    s0 = np.float32(L.orc_bow_score(_p(w), _p(v), 4, _p(dbw[0:3]), _p(dbv[0:3]), 3))
    s2 = np.float32(L.orc_bow_score(_p(w), _p(v), 4, _p(dbw[5:8]), _p(dbv[5:8]), 3))
    assert state[0] == s0 and state[2] == s2 and state[1] == np.float32(0.25) and state[3] == 0
    acc0 = np.float32(s0 + np.float32(0.25)); acc2 = np.float32(s2 + s0)
    best0 = 1 if np.float32(0.25) > s0 else 0          # best keyframe of kf0's group
    best2 = 0 if s0 > s2 else 2
    expect = []
    for a, b in ((acc0, best0), (acc2, best2)):
        if a > np.float32(0.75) * max(acc0, acc2) and b not in expect:
            expect.append(b)
    assert cand[:n].tolist() == expect and n >= 1


def test_oracle_loop_candidates_known_answers():
    """DetectLoopCandidates (src/KeyFrameDatabase.cc:73-194) on a hand-made database: connected keyframes never appear, the
    minScore filter and the 0.75 * bestAccScore cut follow the reference, the accumulated score starts from minScore."""
    L = O.lib()
    L.orc_bow_score.restype = C.c_double
    L.orc_bow_score.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_detect_loop_candidates.restype = C.c_int
    L.orc_detect_loop_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 3 + [C.c_int]
    w = np.array([3, 7, 9, 20], np.uint32); v = np.array([0.5, 0.5, 0.5, 0.5], np.float32)
    kfs = [([3, 7, 9], [0.6, 0.6, 0.5]), ([3, 7, 9, 20], [0.5, 0.5, 0.5, 0.5]), ([3, 7, 20], [0.5, 0.5, 0.7]), ([50], [1.0]), ([3, 9, 20], [0.1, 0.1, 0.99])]
    n = len(kfs)
    kf_off = np.zeros(n + 1, np.int32); kf_off[1:] = np.cumsum([len(a) for a, _ in kfs])
    dbw = np.concatenate([np.array(a, np.uint32) for a, _ in kfs]); dbv = np.concatenate([np.array(b, np.float32) for _, b in kfs])
    sc = [np.float32(L.orc_bow_score(_p(w), _p(v), 4, _p(dbw[kf_off[k]:kf_off[k + 1]].copy()), _p(dbv[kf_off[k]:kf_off[k + 1]].copy()),
                                     int(kf_off[k + 1] - kf_off[k]))) for k in range(n)]
    covis_off = np.array([0, 1, 1, 2, 2, 2], np.int32); covis_idx = np.array([2, 0], np.int32)  # kf0 -> [kf2], kf2 -> [kf0]
    cand = np.zeros(n, np.int32)

    def run(connected, min_score):
        c = np.asarray(connected, np.uint8)
        m = L.orc_detect_loop_candidates(_p(w), _p(v), 4, n, _p(kf_off), _p(dbw), _p(dbv), _p(c), float(min_score), _p(covis_off), _p(covis_idx), _p(cand), n)
        return cand[:m].tolist()

    assert sc[1] == 1.0 and sc[3] == 0.0
    # kf1 (identical vector, 4 common words) is connected: max common among the rest = 3 -> minCommon = int(2.4) = 2 -> kf0, kf2, kf4 scored
    got = run([0, 1, 0, 0, 0], 0.0)
    assert 1 not in got and 3 not in got and len(got) >= 1
    acc = {0: np.float32(sc[0] + sc[2]), 2: np.float32(sc[2] + sc[0]), 4: sc[4]}
    best = {0: 2 if sc[2] > sc[0] else 0, 2: 0 if sc[0] > sc[2] else 2, 4: 4}
    top = max(acc.values())
    expect = []
    for k in (0, 2, 4):                      # first-encounter order: word 3 lists kf0, kf1, kf2, kf4
        if acc[k] > np.float32(0.75) * top and best[k] not in expect:
            expect.append(best[k])
    assert got == expect
    # nothing connected: kf1 shares 4 words -> minCommon = 3 -> only kf1 is scored and returned
    assert run([0, 0, 0, 0, 0], 0.0) == [1]
    # minScore above every score: no candidate; everything connected: no candidate
    assert run([0, 1, 0, 0, 0], 2.0) == [] and run([1, 1, 1, 1, 1], 0.0) == []
    # minScore between: a keyframe below it is dropped from the list but still feeds its neighbour's accumulated score
    lo, hi = sorted([float(sc[0]), float(sc[2])])
    if lo < hi:
        got = run([0, 1, 0, 0, 1], (lo + hi) / 2)
        assert got == [0 if sc[0] > sc[2] else 2]


@pytest.mark.gpu
def test_gpu_search_for_triangulation(vocab):
    """ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:652-819) == oracle on two keyframes that see the same 3-D
    points from poses 0.4 m apart: unmatched keypoints, epipole exclusion, epipolar-line gate, rotation pruning."""
    from orbslam2_amd import api
    L, v = _oracle_voc(vocab)
    L.orc_search_for_triangulation.restype = C.c_int
    L.orc_search_for_triangulation.argtypes = ([C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int]) * 2 + [C.c_void_p] * 3 + \
        [C.c_float] * 4 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    fx = fy = 500.0; cx, cy = 320.0, 240.0
    ctx = api.Context(width=640, height=480, nfeatures=1000, fx=fx, fy=fy, cx=cx, cy=cy, bf=40.0)
    B.vocab_load(ctx, vocab)
    t = ctx.tables()
    rng = np.random.default_rng(21)
    n = 1400
    P = np.stack([rng.uniform(-6, 6, n), rng.uniform(-4, 4, n), rng.uniform(4, 30, n)], axis=1)
    T1 = np.concatenate([np.eye(3), np.zeros((3, 1))], axis=1)
    a = np.deg2rad(3.0)
    R2 = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    T2 = np.concatenate([R2, np.array([[-0.4], [0.02], [0.05]])], axis=1)

    def view(T, seed, desc_base):
        r = np.random.default_rng(seed)
        pc = (T[:, :3] @ P.T).T + T[:, 3]
        k = np.zeros(n, api.KP_DTYPE)
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
    # F12 = K^-T [t12]x R12 K^-1 (LocalMapping::ComputeF12), computed in double and handed over as float
    R12 = T1[:, :3] @ T2[:, :3].T
    t12 = -R12 @ T2[:, 3] + T1[:, 3]
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    F12 = (np.linalg.inv(K).T @ tx @ R12 @ np.linalg.inv(K)).astype(np.float32)
    Cw1 = (-T1[:, :3].T @ T1[:, 3]).astype(np.float32)
    T2f = T2.astype(np.float32)
    sf = np.ascontiguousarray(t["scale"], np.float32); s2 = np.ascontiguousarray(t["sigma2"], np.float32)
    total = 0
    for only_stereo, ori in ((False, True), (True, True), (False, False)):
        ref = np.zeros(n, np.int32)
        nref = L.orc_search_for_triangulation(_p(fv1[0]), _p(fv1[1]), _p(fv1[2]), len(fv1[0]), _p(k1), _p(ur1), _p(mp1), _p(d1), n,
                                              _p(fv2[0]), _p(fv2[1]), _p(fv2[2]), len(fv2[0]), _p(k2), _p(ur2), _p(mp2), _p(d2), n,
                                              _p(F12), _p(Cw1), _p(T2f), fx, fy, cx, cy, _p(sf), _p(s2), int(only_stereo), int(ori), _p(ref))
        got, ngot = B.search_for_triangulation(ctx, fv1, k1, ur1, mp1, d1, fv2, k2, ur2, mp2, d2, F12, Cw1, T2f, fx, fy, cx, cy, only_stereo, ori)
        assert ngot == nref and np.array_equal(got, ref)
        ok = ref >= 0
        assert nref == int(ok.sum()) and not mp1[ok].any() and not mp2[ref[ok]].any()
        if only_stereo:
            assert (ur1[ok] >= 0).all() and (ur2[ref[ok]] >= 0).all()
        assert (ref[ok] == np.nonzero(ok)[0]).mean() > 0.9  # same 3-D point index on both sides
        total += nref
    assert total > 300
    L.orc_vocab_destroy(v)
    ctx.close()


@pytest.mark.gpu
def test_gpu_keyframe_database_culling_compacts_and_keeps_scores(vocab):
    """Keyframe culling in a long session (round-1 advisor finding): erased keyframes must leave the launch list and their
    words must be reclaimed.  300 keyframes, 220 erased in random order (crosses the compaction threshold several times),
    more added in between; after every phase the scores of the survivors equal a fresh database holding only them."""
    from orbslam2_amd import api
    L, v = _oracle_voc(vocab)
    ctx = api.Context(width=752, height=480, nfeatures=1200)
    B.vocab_load(ctx, vocab)
    rng = np.random.default_rng(21)
    db = B.KeyFrameDB(ctx)
    vecs = {}

    def add(seed):
        _, (w, ww), _ = _oracle_transform(L, v, _descs(seed, 600))
        k = db.add(w, ww)
        vecs[k] = (w.copy(), ww.copy())
        return k

    for k in range(300):
        assert add(9000 + k) == k
    _, (qw, qv), _ = _oracle_transform(L, v, _descs(9000 + 17, 600))
    alive = set(range(300))
    order = rng.permutation(300)[:220]
    for step, gone in enumerate(order):
        db.erase(int(gone)); alive.discard(int(gone))
        db.erase(int(gone))  # idempotent
        if step in (60, 150):
            alive.add(add(20000 + step))  # indices keep growing: never reused
        if step % 55 == 54 or step == len(order) - 1:
            common, score = db.score(qw, qv)
            fresh = B.KeyFrameDB(api.Context(width=752, height=480, nfeatures=1200)); B.vocab_load(fresh.ctx, vocab)
            ids = sorted(alive)
            for k in ids:
                fresh.add(*vecs[k])
            c2, s2 = fresh.score(qw, qv)
            assert np.array_equal(common[ids], c2) and np.array_equal(score[ids], s2)
            dead = sorted(set(range(len(db))) - alive)
            assert not common[dead].any() and not score[dead].any()
            fresh.ctx.close()
    assert len(db) == 302
    L.orc_vocab_destroy(v)
    ctx.close()


@pytest.mark.gpu
def test_gpu_vocab_load_rejects_cyclic_and_shared_links(vocab):
    """A corrupt blob whose child links form a cycle (block 1 -> 2 -> 1) or reach a block twice must be refused at load time:
    the descent kernel would otherwise never terminate (round-1 advisor finding)."""
    from orbslam2_amd import api
    ctx = api.Context(width=320, height=240, nfeatures=300)
    blob = bytearray(vocab)
    hdr = 8 + 120
    name, al, nblocks, desc_wp, block_size, feat_off, child_off, total, dtype, dsize, k = struct.unpack("<50s2xII4x5QiiI4x", bytes(blob[8:hdr]))

    def link(b, c):
        return struct.unpack_from("<I", blob, hdr + b * block_size + child_off + 8 * c)[0]

    # find an inner block b (not the root) with a non-leaf link to block ch; point one of ch's non-leaf links back at b
    inner = [(b, c, link(b, c)) for b in range(1, nblocks) for c in range(struct.unpack_from("<H", blob, hdr + b * block_size)[0])
             if not (link(b, c) & 0x80000000)]
    b, c, ch = inner[0]
    back = [cc for cc in range(struct.unpack_from("<H", blob, hdr + ch * block_size)[0])]
    bad = bytearray(blob)
    struct.pack_into("<I", bad, hdr + ch * block_size + child_off + 8 * back[0], b)  # ch -> b -> ch ...
    with pytest.raises(api.OrbfeError):
        B.vocab_load(ctx, bytes(bad))
    bad2 = bytearray(blob)  # two parents share one child block
    b2, c2, ch2 = inner[1]
    struct.pack_into("<I", bad2, hdr + b2 * block_size + child_off + 8 * c2, ch)
    with pytest.raises(api.OrbfeError):
        B.vocab_load(ctx, bytes(bad2))
    B.vocab_load(ctx, vocab)  # the intact blob still loads
    w, wt, nd = B.transform(ctx, _descs(5, 50))
    assert len(w) == 50
    ctx.close()
