"""ctypes plumbing for the bag-of-words entry points of include/orbfe.h (tests only) and a writer for the
fbow vocabulary file format (Thirdparty/fbow/src/fbow.cpp:10-49,172-191) used to synthesise test vocabularies."""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

from . import api

_bound = False


def _bind():
    global _bound
    L = api.load()
    if _bound:
        return L
    vp, ip = C.c_void_p, C.POINTER(C.c_int)
    L.orbfe_vocab_load.restype = C.c_int; L.orbfe_vocab_load.argtypes = [vp, vp, C.c_size_t]
    L.orbfe_bow_transform.restype = C.c_int; L.orbfe_bow_transform.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp]
    L.orbfe_bow_maps.restype = C.c_int; L.orbfe_bow_maps.argtypes = [vp, vp, vp, C.c_int, vp, vp, ip, vp, vp, vp, ip]
    L.orbfe_search_by_bow.restype = C.c_int
    L.orbfe_search_by_bow.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, C.c_int,
                                      C.c_float, C.c_int, vp, ip]
    L.orbfe_search_by_bow_kf.restype = C.c_int
    L.orbfe_search_by_bow_kf.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int,
                                         C.c_float, C.c_int, vp, ip]
    L.orbfe_search_for_triangulation.restype = C.c_int
    L.orbfe_search_for_triangulation.argtypes = [vp] + ([vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int]) * 2 + [vp, vp, vp] + [C.c_float] * 4 + \
        [C.c_int, C.c_int, vp, ip]
    L.orbfe_kfdb_clear.restype = C.c_int; L.orbfe_kfdb_clear.argtypes = [vp]
    L.orbfe_kfdb_add.restype = C.c_int; L.orbfe_kfdb_add.argtypes = [vp, vp, vp, C.c_int, ip]
    L.orbfe_kfdb_erase.restype = C.c_int; L.orbfe_kfdb_erase.argtypes = [vp, C.c_int]
    L.orbfe_kfdb_size.restype = C.c_int; L.orbfe_kfdb_size.argtypes = [vp]
    L.orbfe_kfdb_score.restype = C.c_int; L.orbfe_kfdb_score.argtypes = [vp, vp, vp, C.c_int, vp, vp]
    L.orbfe_detect_reloc_candidates.restype = C.c_int
    L.orbfe_detect_reloc_candidates.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, ip]
    _bound = True
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def vocab_load(ctx, blob: bytes):
    L = _bind()
    buf = np.frombuffer(blob, np.uint8)
    ctx._check(L.orbfe_vocab_load(ctx.h, _p(buf), len(buf)))


def transform(ctx, desc, level=4):
    L = _bind()
    d = np.ascontiguousarray(desc, np.uint8); n = len(d)
    w = np.zeros(max(n, 1), np.uint32); wt = np.zeros(max(n, 1), np.float32); nd = np.zeros(max(n, 1), np.uint32)
    ctx._check(L.orbfe_bow_transform(ctx.h, _p(d), n, level, _p(w), _p(wt), _p(nd)))
    return w[:n], wt[:n], nd[:n]


def maps(word_id, weight, node_id):
    L = _bind()
    n = len(word_id)
    wi = np.ascontiguousarray(word_id, np.uint32); we = np.ascontiguousarray(weight, np.float32); ni = np.ascontiguousarray(node_id, np.uint32)
    words = np.zeros(max(n, 1), np.uint32); ww = np.zeros(max(n, 1), np.float32); nodes = np.zeros(max(n, 1), np.uint32)
    off = np.zeros(n + 1, np.int32); feat = np.zeros(max(n, 1), np.int32); nw, nn = C.c_int(), C.c_int()
    rc = L.orbfe_bow_maps(_p(wi), _p(we), _p(ni), n, _p(words), _p(ww), C.byref(nw), _p(nodes), _p(off), _p(feat), C.byref(nn))
    assert rc == 0
    return words[: nw.value], ww[: nw.value], nodes[: nn.value], off[: nn.value + 1], feat[:n]


def search_by_bow(ctx, kf_fv, kf_valid, kf_desc, kf_angle, f_fv, f_desc, f_angle, nnratio, check_ori):
    L = _bind()
    kn, ko, kf = (np.ascontiguousarray(a) for a in kf_fv); fn, fo, ff = (np.ascontiguousarray(a) for a in f_fv)
    kv = np.ascontiguousarray(kf_valid, np.int32); kd = np.ascontiguousarray(kf_desc, np.uint8); ka = np.ascontiguousarray(kf_angle, np.float32)
    fd = np.ascontiguousarray(f_desc, np.uint8); fa = np.ascontiguousarray(f_angle, np.float32)
    out = np.zeros(max(len(fd), 1), np.int32); nm = C.c_int()
    ctx._check(L.orbfe_search_by_bow(ctx.h, _p(kn), _p(ko), _p(kf), len(kn), _p(kv), _p(kd), _p(ka), len(kd),
                                     _p(fn), _p(fo), _p(ff), len(fn), _p(fd), _p(fa), len(fd), nnratio, int(check_ori), _p(out), C.byref(nm)))
    return out[: len(fd)].copy(), nm.value


def search_by_bow_kf(ctx, fv1, valid1, desc1, angle1, fv2, valid2, desc2, angle2, nnratio, check_ori):
    """ORBmatcher::SearchByFboW(KeyFrame*, KeyFrame*, vpMatches12): match12[i1] = KF2 keypoint or -1, and the count."""
    L = _bind()
    a_n, a_o, a_f = (np.ascontiguousarray(a) for a in fv1); b_n, b_o, b_f = (np.ascontiguousarray(a) for a in fv2)
    v1 = np.ascontiguousarray(valid1, np.int32); d1 = np.ascontiguousarray(desc1, np.uint8); g1 = np.ascontiguousarray(angle1, np.float32)
    v2 = np.ascontiguousarray(valid2, np.int32); d2 = np.ascontiguousarray(desc2, np.uint8); g2 = np.ascontiguousarray(angle2, np.float32)
    out = np.zeros(max(len(d1), 1), np.int32); nm = C.c_int()
    ctx._check(L.orbfe_search_by_bow_kf(ctx.h, _p(a_n), _p(a_o), _p(a_f), len(a_n), _p(v1), _p(d1), _p(g1), len(d1),
                                        _p(b_n), _p(b_o), _p(b_f), len(b_n), _p(v2), _p(d2), _p(g2), len(d2),
                                        nnratio, int(check_ori), _p(out), C.byref(nm)))
    return out[: len(d1)].copy(), nm.value


def search_for_triangulation(ctx, fv1, keys1, ur1, has_mp1, desc1, fv2, keys2, ur2, has_mp2, desc2, F12, Cw1, T2w, fx2, fy2, cx2, cy2,
                             only_stereo, check_ori):
    """ORBmatcher::SearchForTriangulation: match12[i1] = KF2 keypoint or -1, and the count."""
    from .api import KP_DTYPE
    L = _bind()
    a_n, a_o, a_f = (np.ascontiguousarray(a) for a in fv1); b_n, b_o, b_f = (np.ascontiguousarray(a) for a in fv2)
    k1 = np.ascontiguousarray(keys1, KP_DTYPE); k2 = np.ascontiguousarray(keys2, KP_DTYPE)
    u1 = np.ascontiguousarray(ur1, np.float32); u2 = np.ascontiguousarray(ur2, np.float32)
    m1 = np.ascontiguousarray(has_mp1, np.uint8); m2 = np.ascontiguousarray(has_mp2, np.uint8)
    d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
    F = np.ascontiguousarray(F12, np.float32); cw = np.ascontiguousarray(Cw1, np.float32); T = np.ascontiguousarray(T2w, np.float32)
    out = np.zeros(max(len(k1), 1), np.int32); nm = C.c_int()
    ctx._check(L.orbfe_search_for_triangulation(ctx.h, _p(a_n), _p(a_o), _p(a_f), len(a_n), _p(k1), _p(u1), _p(m1), _p(d1), len(k1),
                                                _p(b_n), _p(b_o), _p(b_f), len(b_n), _p(k2), _p(u2), _p(m2), _p(d2), len(k2),
                                                _p(F), _p(cw), _p(T), fx2, fy2, cx2, cy2, int(only_stereo), int(check_ori), _p(out), C.byref(nm)))
    return out[: len(k1)].copy(), nm.value


class KeyFrameDB:
    """KeyFrameDatabase mirror: BoW vectors resident in HBM (orbfe_kfdb_*)."""

    def __init__(self, ctx):
        self.ctx, self.L = ctx, _bind()
        ctx._check(self.L.orbfe_kfdb_clear(ctx.h))

    def add(self, words, weights) -> int:
        w = np.ascontiguousarray(words, np.uint32); v = np.ascontiguousarray(weights, np.float32)
        idx = C.c_int()
        self.ctx._check(self.L.orbfe_kfdb_add(self.ctx.h, _p(w), _p(v), len(w), C.byref(idx)))
        return idx.value

    def erase(self, kf: int):
        self.ctx._check(self.L.orbfe_kfdb_erase(self.ctx.h, kf))

    def __len__(self):
        return self.L.orbfe_kfdb_size(self.ctx.h)

    def score(self, q_words, q_w):
        w = np.ascontiguousarray(q_words, np.uint32); v = np.ascontiguousarray(q_w, np.float32)
        n = max(len(self), 1)
        common = np.zeros(n, np.int32); score = np.zeros(n, np.float32)
        self.ctx._check(self.L.orbfe_kfdb_score(self.ctx.h, _p(w), _p(v), len(w), _p(common), _p(score)))
        return common[: len(self)], score[: len(self)]

    def detect_reloc_candidates(self, q_words, q_w, covis_off, covis_idx, reloc_score):
        w = np.ascontiguousarray(q_words, np.uint32); v = np.ascontiguousarray(q_w, np.float32)
        co = np.ascontiguousarray(covis_off, np.int32); ci = np.ascontiguousarray(covis_idx, np.int32)
        assert reloc_score.dtype == np.float32 and reloc_score.flags.c_contiguous and len(reloc_score) >= len(self)
        cand = np.zeros(max(len(self), 1), np.int32); n = C.c_int()
        self.ctx._check(self.L.orbfe_detect_reloc_candidates(self.ctx.h, _p(w), _p(v), len(w), _p(co), _p(ci), _p(reloc_score),
                                                             _p(cand), len(cand), C.byref(n)))
        return cand[: n.value].copy()


def _kfdb_detect_loop_candidates(self, q_words, q_w, connected, min_score, covis_off, covis_idx):
    """KeyFrameDatabase::DetectLoopCandidates(pKF, minScore): connected = uint8 flag per database keyframe (or None)."""
    w = np.ascontiguousarray(q_words, np.uint32); v = np.ascontiguousarray(q_w, np.float32)
    co = np.ascontiguousarray(covis_off, np.int32); ci = np.ascontiguousarray(covis_idx, np.int32)
    cn = None if connected is None else np.ascontiguousarray(connected, np.uint8)
    cand = np.zeros(max(len(self), 1), np.int32); n = C.c_int()
    self.L.orbfe_detect_loop_candidates.restype = C.c_int
    self.L.orbfe_detect_loop_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    self.ctx._check(self.L.orbfe_detect_loop_candidates(self.ctx.h, _p(w), _p(v), len(w), None if cn is None else _p(cn), float(min_score),
                                                        _p(co), _p(ci), _p(cand), len(cand), C.byref(n)))
    return cand[: n.value].copy()


KeyFrameDB.detect_loop_candidates = _kfdb_detect_loop_candidates


def build_vocabulary(desc: np.ndarray, k: int = 10, levels: int = 3, seed: int = 7) -> bytes:
    """Hierarchical k-majority clustering of binary descriptors, serialised in the fbow file format
    (alignment 8; block = u16 N, u16 isLeaf, u32 parent, k x 32 B descriptors, k x {u32 id|leafbit, f32 weight})."""
    rng = np.random.default_rng(seed)
    bits = np.unpackbits(desc, axis=1, bitorder="little")
    blocks = []  # dict(parent, feats[list of 32B], infos[list of (id_or_child, weight)], leaf)
    n_words = [0]
    total = len(desc)

    def split(idx, parent, depth):
        bid = len(blocks)
        blk = dict(parent=parent, feats=[], infos=[], leaf=depth == levels - 1)
        blocks.append(blk)
        kk = min(k, len(idx))
        centers = bits[rng.choice(idx, kk, replace=False)].copy()
        assign = None
        for _ in range(4):
            dist = (bits[idx][:, None, :] != centers[None, :, :]).sum(axis=2)
            assign = dist.argmin(axis=1)
            for c in range(kk):
                m = assign == c
                if m.any():
                    centers[c] = (bits[idx][m].mean(axis=0) >= 0.5)
        dist = (bits[idx][:, None, :] != centers[None, :, :]).sum(axis=2)
        assign = dist.argmin(axis=1)
        for c in range(kk):
            sub = idx[assign == c]
            blk["feats"].append(np.packbits(centers[c], bitorder="little").tobytes())
            if depth == levels - 1 or len(sub) < 2:
                w = float(np.log(total / max(len(sub), 1)))
                blk["infos"].append((0x80000000 | n_words[0], np.float32(w)))
                n_words[0] += 1
            else:
                blk["infos"].append(None)  # patched below
                child = split(sub, bid, depth + 1)
                blk["infos"][c] = (child, np.float32(0.0))
        return bid

    split(np.arange(total), 0, 0)
    desc_wp, feat_off = 32, 8
    child_off = feat_off + k * desc_wp
    block_size = feat_off + k * (desc_wp + 8)
    data = bytearray(block_size * len(blocks))
    for b, blk in enumerate(blocks):
        o = b * block_size
        all_leaf = all(i[0] & 0x80000000 for i in blk["infos"])
        struct.pack_into("<HHI", data, o, len(blk["feats"]), 1 if all_leaf else 0, blk["parent"])
        for c, f in enumerate(blk["feats"]):
            data[o + feat_off + c * desc_wp: o + feat_off + c * desc_wp + 32] = f
            struct.pack_into("<If", data, o + child_off + c * 8, blk["infos"][c][0], float(blk["infos"][c][1]))
    params = struct.pack("<50s2xII4x5QiiI4x", b"orb", 8, len(blocks), desc_wp, block_size, feat_off, child_off, len(data), 0, 32, k)
    assert len(params) == 120
    return struct.pack("<Q", 55824124) + params + bytes(data)


def build_full_vocabulary(k: int = 10, levels: int = 6, seed: int = 11) -> bytes:
    """A COMPLETE k-ary tree of `levels` levels in the fbow file format -- (k^levels - 1) / (k - 1) blocks, k^levels words: 45 MB
    for k = 10, levels = 6, the size of the ORB vocabulary ORB-SLAM2 ships (which is not in the reference checkout).  No
    clustering: a child's descriptor is its parent's with a level-dependent share of random bit flips, so descents behave like
    those of a trained tree (siblings are distinct, descendants stay near their ancestors).  Block b's children are blocks
    k * b + 1 .. k * b + k (breadth-first numbering); leaf weights are drawn from [1, 10).  Vectorised: a few seconds."""
    rng = np.random.default_rng(seed)
    n_blocks = (k ** levels - 1) // (k - 1)
    n_internal = (k ** (levels - 1) - 1) // (k - 1)   # blocks whose nodes have children
    desc_wp, feat_off = 32, 8
    child_off = feat_off + k * desc_wp
    block_size = feat_off + k * (desc_wp + 8)
    data = np.zeros((n_blocks, block_size), np.uint8)
    feats = np.zeros((n_blocks, k, 32), np.uint8)
    flip_p = [0.5, 0.22, 0.14, 0.09, 0.06, 0.04, 0.03, 0.02]
    parent_feat = np.zeros((1, 32), np.uint8)          # the root block's "parent"
    first = 0
    for d in range(levels):
        nb = k ** d
        base = np.repeat(parent_feat, k, axis=0).reshape(nb, k, 32)
        flips = np.packbits(rng.random((nb, k, 256)) < flip_p[min(d, len(flip_p) - 1)], axis=2, bitorder="little")
        feats[first:first + nb] = base ^ flips
        parent_feat = feats[first:first + nb].reshape(nb * k, 32)
        first += nb
    hdr = np.zeros((n_blocks, 2), np.uint32)
    leaf_block = np.arange(n_blocks) >= n_internal
    hdr[:, 0] = k | (leaf_block.astype(np.uint32) << 16)                    # u16 N, u16 isLeaf
    hdr[1:, 1] = ((np.arange(1, n_blocks) - 1) // k).astype(np.uint32)      # parent block
    data[:, 0:8] = hdr.view(np.uint8).reshape(n_blocks, 8)
    data[:, feat_off:feat_off + k * desc_wp] = feats.reshape(n_blocks, k * 32)
    info = np.zeros((n_blocks, k, 2), np.uint32)
    child = (np.arange(n_blocks)[:, None] * k + 1 + np.arange(k)[None, :]).astype(np.uint32)
    info[:n_internal, :, 0] = child[:n_internal]
    word = (np.arange(n_blocks - n_internal)[:, None] * k + np.arange(k)[None, :]).astype(np.uint32)
    info[n_internal:, :, 0] = word | np.uint32(0x80000000)
    info[n_internal:, :, 1] = rng.uniform(1.0, 10.0, (n_blocks - n_internal, k)).astype(np.float32).view(np.uint32)
    data[:, child_off:child_off + 8 * k] = info.view(np.uint8).reshape(n_blocks, 8 * k)
    blob = data.tobytes()
    params = struct.pack("<50s2xII4x5QiiI4x", b"orb", 8, n_blocks, desc_wp, block_size, feat_off, child_off, len(blob), 0, 32, k)
    return struct.pack("<Q", 55824124) + params + blob
