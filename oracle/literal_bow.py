"""oracle/literal_bow.py -- TEST INFRASTRUCTURE (never imported by the product, bench.py's timed region or smoke()).

A second, independent statement of the bag-of-words row: the loops of the vendored fbow library and of the reference's
BoW matcher transcribed LITERALLY into Python -- the vocabulary as the byte blob fbow reads (`Vocabulary::fromStream`), blocks
addressed through the same offsets, `std::map` as a dict walked in key order, the two-pointer merges with `lower_bound` as
written.  It shares no code with oracle/orb_oracle_bow.c (sorted arrays, binary searches) or orbslam2_amd/csrc/orbfe_bow.hip
(thread per descriptor, pair kernels), so `C oracle == this file` checks their array formulations against the reference's own
containers.  Pure-Python loops: small cases only.

Transcribed (reference file:line):
  fbow::Vocabulary::fromStream / params / Block          Thirdparty/fbow/src/fbow.cpp:181-191, fbow.h:117-178
  fbow::Vocabulary::transform(features, level, r1, r2)   Thirdparty/fbow/src/fbow.cpp:50-86 -> _transform2<L1_32bytes>, fbow.h:400-444
  fbow::fBow::score                                      Thirdparty/fbow/src/fbow.cpp:206-254
  ORBmatcher::SearchByFboW(KeyFrame*, Frame&, matches)   src/ORBmatcher.cc:157-283
  ORBmatcher::ComputeThreeMaxima / DescriptorDistance    src/ORBmatcher.cc:1597-1659 (from oracle/literal_matchers.py)
"""
from __future__ import annotations

import math
import struct

import numpy as np

from .literal_matchers import HISTO_LENGTH, TH_LOW, c_round, compute_three_maxima, descriptor_distance

F32 = np.float32


class Vocabulary:
    """fbow::Vocabulary over the stream bytes: signature, `params` struct (with the compiler's padding), block data."""

    def __init__(self, blob: bytes):
        sig, = struct.unpack_from("<Q", blob, 0)
        if sig != 55824124:
            raise ValueError("Vocabulary::fromStream invalid signature")
        # struct params { char _desc_name_[50]; uint32_t _aligment, _nblocks; uint64_t _desc_size_bytes_wp, _block_size_bytes_wp,
        #                 _feature_off_start, _child_off_start, _total_size; int32_t _desc_type, _desc_size; uint32_t _m_k; }
        (self.desc_name, self.aligment, self.nblocks, self.desc_size_bytes_wp, self.block_size_bytes_wp, self.feature_off_start,
         self.child_off_start, self.total_size, self.desc_type, self.desc_size, self.m_k) = struct.unpack_from("<50s2xII4xQQQQQiiI", blob, 8)
        # the compiler's layout on LP64: 2 bytes after the name (uint32_t alignment), 4 before the first uint64_t, 4 at the end
        params_size = (struct.calcsize("<50s2xII4xQQQQQiiI") + 7) & ~7  # sizeof(params) = 120
        self.data = blob[8 + params_size: 8 + params_size + self.total_size]
        assert len(self.data) == self.total_size

    # Block accessors, fbow.h:165-178
    def block_n(self, b):
        return struct.unpack_from("<H", self.data, b * self.block_size_bytes_wp)[0]

    def block_feature(self, b, i):
        o = b * self.block_size_bytes_wp + self.feature_off_start + i * self.desc_size_bytes_wp
        return np.frombuffer(self.data, np.uint8, self.desc_size, o)

    def block_node_info(self, b, i):
        o = b * self.block_size_bytes_wp + self.child_off_start + i * 8
        return struct.unpack_from("<If", self.data, o)  # id_or_childblock, weight


def transform(voc: Vocabulary, features: np.ndarray, store_level: int):
    """Vocabulary::transform(features, level, r1, r2) -> (r1: {word: float32 weight sum}, r2: {node: [feature indices]})."""
    r1, r2 = {}, {}
    nbits = int(math.ceil(math.log2(voc.m_k)))
    for cur_feature in range(len(features)):
        feat = features[cur_feature]
        c_block = 0
        level = 0
        cur_node_id = 0
        while True:
            best_first, best_second = 0xFFFFFFFF, 0  # std::pair<DType, uint32_t>(numeric_limits<uint32_t>::max(), 0)
            for cur_node in range(voc.block_n(c_block)):
                d = descriptor_distance(voc.block_feature(c_block, cur_node), feat)  # L1_32bytes: popcount of the xor
                if d < best_first:
                    best_first, best_second = d, cur_node
            if level == store_level:
                r2.setdefault(cur_node_id, []).append(cur_feature)
            id_or_child, weight = voc.block_node_info(c_block, best_second)
            isleaf = bool(id_or_child & 0x80000000)
            node_id = id_or_child & 0x7FFFFFFF
            if isleaf:
                r1[node_id] = F32(r1.get(node_id, F32(0))) + F32(weight)  # fBow is std::map<uint32_t, float>: float accumulation
                if level < store_level:
                    r2.setdefault(cur_node_id, []).append(cur_feature)
                break
            c_block = node_id
            cur_node_id = ((cur_node_id << nbits) | best_second) & 0xFFFFFFFF
            level += 1
            if not (not isleaf and node_id != 0):  # while( !bn_info->isleaf() && bn_info->getId()!=0 )
                break
    return r1, r2


def score(v1: dict, v2: dict) -> float:
    """fBow::score: sum over the common words of vi * wi -- a FLOAT product (`const auto &vi = it->second`) added to a double --
    walking both maps in key order."""
    k1, k2 = sorted(v1), sorted(v2)
    i = j = 0
    s = 0.0
    while i < len(k1) and j < len(k2):
        if k1[i] == k2[j]:
            s += float(F32(v1[k1[i]]) * F32(v2[k2[j]]))
            i += 1
            j += 1
        elif k1[i] < k2[j]:
            while i < len(k1) and k1[i] < k2[j]:
                i += 1
        else:
            while j < len(k2) and k2[j] < k1[i]:
                j += 1
    # ||v - w||_L2 = sqrt(2 - 2 sum v_i w_i) for unit vectors (Nister 2006); the vectors of transform(features, level, ...) are
    # NOT normalised (only the one-argument transform normalises), so with real vocabularies this saturates at 1
    if s >= 1:
        s = 1.0
    else:
        s = 1.0 - math.sqrt(1.0 - s)
    return s


def _lower_bound(keys, k):
    lo = 0
    while lo < len(keys) and keys[lo] < k:
        lo += 1
    return lo


def search_by_fbow_kf_frame(kf_featvec: dict, kf_has_good_point, kf_desc, kf_angle, f_featvec: dict, f_desc, f_angle, n_f,
                            nnratio, check_orientation):
    """ORBmatcher::SearchByFboW(KeyFrame*, Frame&, vpMapPointMatches): returns (match[n_f] = KF keypoint index or -1, nmatches).
    kf_has_good_point[i]: vpMapPointsKF[i] is non-null and not bad."""
    matches = [-1] * n_f
    nmatches = 0
    rot_hist = [[] for _ in range(HISTO_LENGTH)]
    factor = F32(1.0) / F32(HISTO_LENGTH)
    kf_keys, f_keys = sorted(kf_featvec), sorted(f_featvec)
    ki = fi = 0
    while ki < len(kf_keys) and fi < len(f_keys):
        if kf_keys[ki] == f_keys[fi]:
            v_kf, v_f = kf_featvec[kf_keys[ki]], f_featvec[f_keys[fi]]
            for real_kf in v_kf:
                if not kf_has_good_point[real_kf]:
                    continue
                d_kf = kf_desc[real_kf]
                best1, best_idx, best2 = 256, -1, 256
                for real_f in v_f:
                    if matches[real_f] >= 0:
                        continue
                    dist = descriptor_distance(d_kf, f_desc[real_f])
                    if dist < best1:
                        best2 = best1
                        best1 = dist
                        best_idx = real_f
                    elif dist < best2:
                        best2 = dist
                if best1 <= TH_LOW:
                    if F32(best1) < F32(nnratio) * F32(best2):
                        matches[best_idx] = real_kf
                        if check_orientation:
                            rot = F32(kf_angle[real_kf]) - F32(f_angle[best_idx])
                            if rot < 0.0:
                                rot = rot + F32(360.0)
                            b = c_round(rot * factor)
                            if b == HISTO_LENGTH:
                                b = 0
                            rot_hist[b].append(best_idx)
                        nmatches += 1
            ki += 1
            fi += 1
        elif kf_keys[ki] < f_keys[fi]:
            ki = _lower_bound(kf_keys, f_keys[fi])
        else:
            fi = _lower_bound(f_keys, kf_keys[ki])
    if check_orientation:
        ind1, ind2, ind3 = compute_three_maxima(rot_hist, HISTO_LENGTH)
        for i in range(HISTO_LENGTH):
            if i in (ind1, ind2, ind3):
                continue
            for j in rot_hist[i]:
                matches[j] = -1
                nmatches -= 1
    return np.array(matches, np.int32), nmatches


def search_by_fbow_kf_kf(fv1: dict, good1, desc1, angle1, n1, fv2: dict, good2, desc2, angle2, nnratio, check_orientation):
    """ORBmatcher::SearchByFboW(KeyFrame*, KeyFrame*, vpMatches12), src/ORBmatcher.cc:517-650: returns (match12[n1] = keypoint of
    keyframe 2 or -1, nmatches).  good*[i]: the keypoint's map point is non-null and not bad.  (Here the threshold is `< TH_LOW`
    and the histogram holds idx1; a matched keypoint of keyframe 2 stays matched even if the rotation filter drops the pair.)"""
    match12 = [-1] * n1
    matched2 = [False] * len(good2)
    rot_hist = [[] for _ in range(HISTO_LENGTH)]
    factor = F32(1.0) / F32(HISTO_LENGTH)
    nmatches = 0
    k1, k2 = sorted(fv1), sorted(fv2)
    i = j = 0
    while i < len(k1) and j < len(k2):
        if k1[i] == k2[j]:
            for idx1 in fv1[k1[i]]:
                if not good1[idx1]:
                    continue
                d1 = desc1[idx1]
                best1, best_idx2, best2 = 256, -1, 256
                for idx2 in fv2[k2[j]]:
                    if matched2[idx2] or not good2[idx2]:
                        continue
                    dist = descriptor_distance(d1, desc2[idx2])
                    if dist < best1:
                        best2 = best1
                        best1 = dist
                        best_idx2 = idx2
                    elif dist < best2:
                        best2 = dist
                if best1 < TH_LOW:
                    if F32(best1) < F32(nnratio) * F32(best2):
                        match12[idx1] = best_idx2
                        matched2[best_idx2] = True
                        if check_orientation:
                            rot = F32(angle1[idx1]) - F32(angle2[best_idx2])
                            if rot < 0.0:
                                rot = rot + F32(360.0)
                            b = c_round(rot * factor)
                            if b == HISTO_LENGTH:
                                b = 0
                            rot_hist[b].append(idx1)
                        nmatches += 1
            i += 1
            j += 1
        elif k1[i] < k2[j]:
            i = _lower_bound(k1, k2[j])
        else:
            j = _lower_bound(k2, k1[i])
    if check_orientation:
        ind1, ind2, ind3 = compute_three_maxima(rot_hist, HISTO_LENGTH)
        for b in range(HISTO_LENGTH):
            if b in (ind1, ind2, ind3):
                continue
            for idx1 in rot_hist[b]:
                match12[idx1] = -1
                nmatches -= 1
    return np.array(match12, np.int32), nmatches


def check_dist_epipolar_line(kp1, kp2, F12, level_sigma2):
    """ORBmatcher::CheckDistEpipolarLine, src/ORBmatcher.cc:138-155 (float arithmetic, left to right)."""
    x1, y1 = F32(kp1["x"]), F32(kp1["y"])
    a = (x1 * F32(F12[0, 0]) + y1 * F32(F12[1, 0])) + F32(F12[2, 0])
    b = (x1 * F32(F12[0, 1]) + y1 * F32(F12[1, 1])) + F32(F12[2, 1])
    c = (x1 * F32(F12[0, 2]) + y1 * F32(F12[1, 2])) + F32(F12[2, 2])
    num = (a * F32(kp2["x"]) + b * F32(kp2["y"])) + c
    den = a * a + b * b
    if den == 0:
        return False
    dsqr = num * num / den
    return float(dsqr) < 3.84 * float(F32(level_sigma2[int(kp2["octave"])]))


def search_for_triangulation(fv1: dict, keys1, ur1, has_mp1, desc1, fv2: dict, keys2, ur2, has_mp2, desc2, F12, Cw1, T2w,
                             fx2, fy2, cx2, cy2, scale_factors, level_sigma2, only_stereo, check_orientation):
    """ORBmatcher::SearchForTriangulation, src/ORBmatcher.cc:652-819 -> (vMatches12, nmatches)."""
    F12 = np.asarray(F12, np.float32).reshape(3, 3)
    T2w = np.asarray(T2w, np.float32)
    # epipole in the second image: C2 = R2w * Cw + t2w
    C2 = [((F32(T2w[i, 0]) * F32(Cw1[0]) + F32(T2w[i, 1]) * F32(Cw1[1])) + F32(T2w[i, 2]) * F32(Cw1[2])) + F32(T2w[i, 3]) for i in range(3)]
    invz = F32(1.0) / C2[2]
    ex = F32(fx2) * C2[0] * invz + F32(cx2)
    ey = F32(fy2) * C2[1] * invz + F32(cy2)
    n1, n2 = len(keys1), len(keys2)
    nmatches = 0
    matched2 = [False] * n2
    m12 = [-1] * n1
    rot_hist = [[] for _ in range(HISTO_LENGTH)]
    factor = F32(1.0) / F32(HISTO_LENGTH)
    k1, k2 = sorted(fv1), sorted(fv2)
    i = j = 0
    while i < len(k1) and j < len(k2):
        if k1[i] == k2[j]:
            for idx1 in fv1[k1[i]]:
                if has_mp1[idx1]:
                    continue
                stereo1 = ur1[idx1] >= 0
                if only_stereo and not stereo1:
                    continue
                kp1 = keys1[idx1]
                d1 = desc1[idx1]
                best_dist, best_idx2 = TH_LOW, -1
                for idx2 in fv2[k2[j]]:
                    if matched2[idx2] or has_mp2[idx2]:
                        continue
                    stereo2 = ur2[idx2] >= 0
                    if only_stereo and not stereo2:
                        continue
                    dist = descriptor_distance(d1, desc2[idx2])
                    if dist > TH_LOW or dist > best_dist:
                        continue
                    kp2 = keys2[idx2]
                    if not stereo1 and not stereo2:
                        distex = ex - F32(kp2["x"])
                        distey = ey - F32(kp2["y"])
                        if distex * distex + distey * distey < F32(100) * F32(scale_factors[int(kp2["octave"])]):
                            continue
                    if check_dist_epipolar_line(kp1, kp2, F12, level_sigma2):
                        best_idx2 = idx2
                        best_dist = dist
                if best_idx2 >= 0:
                    kp2 = keys2[best_idx2]
                    m12[idx1] = best_idx2
                    matched2[best_idx2] = True
                    nmatches += 1
                    if check_orientation:
                        rot = F32(kp1["angle"]) - F32(kp2["angle"])
                        if rot < 0.0:
                            rot = rot + F32(360.0)
                        b = c_round(rot * factor)
                        if b == HISTO_LENGTH:
                            b = 0
                        rot_hist[b].append(idx1)
            i += 1
            j += 1
        elif k1[i] < k2[j]:
            i = _lower_bound(k1, k2[j])
        else:
            j = _lower_bound(k2, k1[i])
    if check_orientation:
        ind1, ind2, ind3 = compute_three_maxima(rot_hist, HISTO_LENGTH)
        for b in range(HISTO_LENGTH):
            if b in (ind1, ind2, ind3):
                continue
            for idx1 in rot_hist[b]:
                matched2[m12[idx1]] = False
                m12[idx1] = -1
                nmatches -= 1
    return np.array(m12, np.int32), nmatches


class LKeyFrame:
    """The members of KeyFrame that KeyFrameDatabase touches."""

    def __init__(self, mnId, bow: dict):
        self.mnId = mnId
        self.mFbowVec = bow                    # {word: float32 weight}
        self.mnRelocQuery = -1; self.mnRelocWords = 0; self.mRelocScore = F32(0)
        self.mnLoopQuery = -1; self.mnLoopWords = 0; self.mLoopScore = F32(0)
        self.best_covisibles = []              # GetBestCovisibilityKeyFrames(10), in its order
        self.connected = set()                 # GetConnectedKeyFrames()


class KeyFrameDatabase:
    """src/KeyFrameDatabase.cc:38-307 transcribed: mvInvertedFile as a dict of lists (push_back order), list / set / pair logic as
    written.  Query ids must be fresh per call (mnId of the frame / keyframe), as in the reference."""

    def __init__(self):
        self.mvInvertedFile = {}

    def add(self, pKF):
        for w in sorted(pKF.mFbowVec):
            self.mvInvertedFile.setdefault(w, []).append(pKF)

    def erase(self, pKF):
        for w in sorted(pKF.mFbowVec):
            lKFs = self.mvInvertedFile.get(w, [])
            for k, other in enumerate(lKFs):
                if other is pKF:
                    del lKFs[k]
                    break

    def DetectRelocalizationCandidates(self, F_id, F_bow):
        lKFsSharingWords = []
        for w in sorted(F_bow):
            for pKFi in self.mvInvertedFile.get(w, []):
                if pKFi.mnRelocQuery != F_id:
                    pKFi.mnRelocWords = 0
                    pKFi.mnRelocQuery = F_id
                    lKFsSharingWords.append(pKFi)
                pKFi.mnRelocWords += 1
        if not lKFsSharingWords:
            return []
        maxCommonWords = 0
        for kf in lKFsSharingWords:
            if kf.mnRelocWords > maxCommonWords:
                maxCommonWords = kf.mnRelocWords
        minCommonWords = int(F32(maxCommonWords) * F32(0.8))
        lScoreAndMatch = []
        for pKFi in lKFsSharingWords:
            if pKFi.mnRelocWords > minCommonWords:
                si = F32(score(F_bow, pKFi.mFbowVec))
                pKFi.mRelocScore = si
                lScoreAndMatch.append((si, pKFi))
        if not lScoreAndMatch:
            return []
        lAccScoreAndMatch = []
        bestAccScore = F32(0)
        for first, pKFi in lScoreAndMatch:
            bestScore = first
            accScore = bestScore
            pBestKF = pKFi
            for pKF2 in pKFi.best_covisibles:
                if pKF2.mnRelocQuery != F_id:
                    continue
                accScore = accScore + pKF2.mRelocScore
                if pKF2.mRelocScore > bestScore:
                    pBestKF = pKF2
                    bestScore = pKF2.mRelocScore
            lAccScoreAndMatch.append((accScore, pBestKF))
            if accScore > bestAccScore:
                bestAccScore = accScore
        minScoreToRetain = F32(0.75) * bestAccScore
        added, out = set(), []
        for si, pKFi in lAccScoreAndMatch:
            if si > minScoreToRetain:
                if id(pKFi) not in added:
                    out.append(pKFi)
                    added.add(id(pKFi))
        return out

    def DetectLoopCandidates(self, pKF, minScore):
        minScore = F32(minScore)
        spConnectedKeyFrames = pKF.connected
        lKFsSharingWords = []
        for w in sorted(pKF.mFbowVec):
            for pKFi in self.mvInvertedFile.get(w, []):
                if pKFi.mnLoopQuery != pKF.mnId:
                    pKFi.mnLoopWords = 0
                    if pKFi not in spConnectedKeyFrames:
                        pKFi.mnLoopQuery = pKF.mnId
                        lKFsSharingWords.append(pKFi)
                pKFi.mnLoopWords += 1
        if not lKFsSharingWords:
            return []
        lScoreAndMatch = []
        maxCommonWords = 0
        for kf in lKFsSharingWords:
            if kf.mnLoopWords > maxCommonWords:
                maxCommonWords = kf.mnLoopWords
        minCommonWords = int(F32(maxCommonWords) * F32(0.8))
        for pKFi in lKFsSharingWords:
            if pKFi.mnLoopWords > minCommonWords:
                si = F32(score(pKF.mFbowVec, pKFi.mFbowVec))
                pKFi.mLoopScore = si
                if si >= minScore:
                    lScoreAndMatch.append((si, pKFi))
        if not lScoreAndMatch:
            return []
        lAccScoreAndMatch = []
        bestAccScore = minScore
        for first, pKFi in lScoreAndMatch:
            bestScore = first
            accScore = first
            pBestKF = pKFi
            for pKF2 in pKFi.best_covisibles:
                if pKF2.mnLoopQuery == pKF.mnId and pKF2.mnLoopWords > minCommonWords:
                    accScore = accScore + pKF2.mLoopScore
                    if pKF2.mLoopScore > bestScore:
                        pBestKF = pKF2
                        bestScore = pKF2.mLoopScore
            lAccScoreAndMatch.append((accScore, pBestKF))
            if accScore > bestAccScore:
                bestAccScore = accScore
        minScoreToRetain = F32(0.75) * bestAccScore
        added, out = set(), []
        for first, pKFi in lAccScoreAndMatch:
            if first > minScoreToRetain:
                if id(pKFi) not in added:
                    out.append(pKFi)
                    added.add(id(pKFi))
        return out
