"""oracle/literal_matchers.py -- TEST INFRASTRUCTURE (never imported by the product, bench.py's timed region or smoke()).

A second, independent statement of the Tracking-thread matchers: the reference's loops transcribed LITERALLY into Python
(numpy float32 scalars, one IEEE rounding per operation as the C++ float expressions have under -ffp-contract=off), with
Frame::mGrid as the reference builds it (vector of vectors, insertion order) and GetFeaturesInArea as a plain triple loop.
It shares no code, helper or data structure with oracle/orb_oracle_match.c or orbslam2_amd/csrc/orbfe_match.hip (which sort
64-bit candidate keys instead of replaying the loops), so `C oracle == this file` checks the sequential accept / overwrite /
ratio / orientation rules of both by something other than their own twin (round-1 verdict, "matcher resolve logic is compared
with its own twin").  Pure-Python loops: small cases only (a frame of ~1500 keypoints takes a second or two).

Transcribed functions (reference file:line):
  Frame::AssignFeaturesToGrid / PosInGrid       src/Frame.cc:231-246, 388-401
  Frame::GetFeaturesInArea                      src/Frame.cc:336-386
  Frame::isInFrustum                            src/Frame.cc:256-315
  MapPoint::PredictScale                        src/MapPoint.cc:402-417
  ORBmatcher::SearchByProjection(F, points)     src/ORBmatcher.cc:43-127 (+ RadiusByViewingCos :129-135)
  ORBmatcher::SearchByProjection(Cur, Last)     src/ORBmatcher.cc:1324-1466
  ORBmatcher::SearchByProjection(Cur, KF, ...)  src/ORBmatcher.cc:1468-1595
  ORBmatcher::SearchForInitialization           src/ORBmatcher.cc:400-515
  ORBmatcher::ComputeThreeMaxima                src/ORBmatcher.cc:1597-1638
  ORBmatcher::DescriptorDistance                src/ORBmatcher.cc:1643-1659
Library semantics assumed (same contract as DESIGN.md section 2, stated here on its own): cv::Mat 3x3 * 3x1 + 3x1 in float32 as
((r0*x0 + r1*x1) + r2*x2) + t; cv::norm / Mat::dot of CV_32F accumulate in double; log() of MapPoint::PredictScale is the
double-precision logarithm rounded to float (contract Q4).
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32
FRAME_GRID_COLS, FRAME_GRID_ROWS = 64, 48  # include/Frame.h:36-37
TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30  # src/ORBmatcher.cc:35-37
INT_MAX = 2 ** 31 - 1


def c_round(x) -> int:
    """C round(): half away from zero, on the double value of x."""
    x = float(x)
    return int(math.floor(x + 0.5)) if x >= 0 else -int(math.floor(-x + 0.5))


def descriptor_distance(a: np.ndarray, b: np.ndarray) -> int:
    """The bit-twiddling popcount of src/ORBmatcher.cc:1643-1659, word by word."""
    pa = np.frombuffer(np.ascontiguousarray(a, np.uint8).tobytes(), "<u4")
    pb = np.frombuffer(np.ascontiguousarray(b, np.uint8).tobytes(), "<u4")
    dist = 0
    for i in range(8):
        v = int(pa[i]) ^ int(pb[i])
        v = v - ((v >> 1) & 0x55555555)
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333)
        dist += ((((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) & 0xFFFFFFFF) >> 24
    return dist


def compute_three_maxima(histo, L):
    max1 = max2 = max3 = 0
    ind1 = ind2 = ind3 = -1
    for i in range(L):
        s = len(histo[i])
        if s > max1:
            max3 = max2; max2 = max1; max1 = s
            ind3 = ind2; ind2 = ind1; ind1 = i
        elif s > max2:
            max3 = max2; max2 = s
            ind3 = ind2; ind2 = i
        elif s > max3:
            max3 = s; ind3 = i
    if F32(max2) < F32(0.1) * F32(max1):
        ind2 = -1; ind3 = -1
    elif F32(max3) < F32(0.1) * F32(max1):
        ind3 = -1
    return ind1, ind2, ind3


def _rot_bin(a1, a2):
    factor = F32(1.0) / F32(HISTO_LENGTH)
    rot = F32(a1) - F32(a2)
    if rot < 0.0:
        rot = rot + F32(360.0)
    b = c_round(rot * factor)
    if b == HISTO_LENGTH:
        b = 0
    assert 0 <= b < HISTO_LENGTH
    return b


class Frame:
    """The members of ORB_SLAM2::Frame the matchers read, with mGrid built as AssignFeaturesToGrid does."""

    def __init__(self, keys_un, descriptors, u_right, bounds, cam, scale_factors, Tcw=None):
        self.mvKeysUn = keys_un
        self.N = len(keys_un)
        self.mDescriptors = descriptors
        self.mvuRight = u_right if u_right is not None else np.full(self.N, -1.0, np.float32)
        self.mnMinX, self.mnMaxX, self.mnMinY, self.mnMaxY = (F32(b) for b in bounds)
        self.fx, self.fy, self.cx, self.cy, self.mbf, self.mb = (F32(c) for c in cam)
        self.mvScaleFactors = np.asarray(scale_factors, np.float32)
        self.mnScaleLevels = len(self.mvScaleFactors)
        self.mTcw = None if Tcw is None else np.asarray(Tcw, np.float32)
        self.mfGridElementWidthInv = F32(FRAME_GRID_COLS) / (self.mnMaxX - self.mnMinX)  # src/Frame.cc:99-100
        self.mfGridElementHeightInv = F32(FRAME_GRID_ROWS) / (self.mnMaxY - self.mnMinY)
        self.mGrid = [[[] for _ in range(FRAME_GRID_ROWS)] for _ in range(FRAME_GRID_COLS)]
        for i in range(self.N):  # AssignFeaturesToGrid
            ok, px, py = self.pos_in_grid(self.mvKeysUn[i])
            if ok:
                self.mGrid[px][py].append(i)

    def pos_in_grid(self, kp):
        px = c_round((F32(kp["x"]) - self.mnMinX) * self.mfGridElementWidthInv)
        py = c_round((F32(kp["y"]) - self.mnMinY) * self.mfGridElementHeightInv)
        if px < 0 or px >= FRAME_GRID_COLS or py < 0 or py >= FRAME_GRID_ROWS:
            return False, px, py
        return True, px, py

    def get_features_in_area(self, x, y, r, min_level=-1, max_level=-1):
        x, y, r = F32(x), F32(y), F32(r)
        out = []
        n_min_cx = max(0, int(math.floor((x - self.mnMinX - r) * self.mfGridElementWidthInv)))
        if n_min_cx >= FRAME_GRID_COLS:
            return out
        n_max_cx = min(FRAME_GRID_COLS - 1, int(math.ceil((x - self.mnMinX + r) * self.mfGridElementWidthInv)))
        if n_max_cx < 0:
            return out
        n_min_cy = max(0, int(math.floor((y - self.mnMinY - r) * self.mfGridElementHeightInv)))
        if n_min_cy >= FRAME_GRID_ROWS:
            return out
        n_max_cy = min(FRAME_GRID_ROWS - 1, int(math.ceil((y - self.mnMinY + r) * self.mfGridElementHeightInv)))
        if n_max_cy < 0:
            return out
        check_levels = (min_level > 0) or (max_level >= 0)
        for ix in range(n_min_cx, n_max_cx + 1):
            for iy in range(n_min_cy, n_max_cy + 1):
                for j in self.mGrid[ix][iy]:
                    kp = self.mvKeysUn[j]
                    if check_levels:
                        if kp["octave"] < min_level:
                            continue
                        if max_level >= 0 and kp["octave"] > max_level:
                            continue
                    distx = F32(kp["x"]) - x
                    disty = F32(kp["y"]) - y
                    if abs(distx) < r and abs(disty) < r:
                        out.append(j)
        return out


def _rx_plus_t(T, x):
    """cv::Mat Rcw * x3Dw + tcw on CV_32F (T = 3x4 [R|t])."""
    out = []
    for i in range(3):
        acc = (F32(T[i, 0]) * F32(x[0]) + F32(T[i, 1]) * F32(x[1])) + F32(T[i, 2]) * F32(x[2])
        out.append(acc + F32(T[i, 3]))
    return out


def _camera_center(T):
    """-Rcw.t() * tcw."""
    return [((-F32(T[0, i])) * F32(T[0, 3]) + (-F32(T[1, i])) * F32(T[1, 3])) + (-F32(T[2, i])) * F32(T[2, 3]) for i in range(3)]


def _norm3(v):
    return F32(math.sqrt(float(v[0]) * float(v[0]) + float(v[1]) * float(v[1]) + float(v[2]) * float(v[2])))


def predict_scale(mf_max_distance, current_dist, log_scale_factor, n_levels):
    ratio = F32(mf_max_distance) / F32(current_dist)
    n_scale = int(math.ceil(F32(F32(math.log(float(ratio))) / F32(log_scale_factor))))
    if n_scale < 0:
        n_scale = 0
    elif n_scale >= n_levels:
        n_scale = n_levels - 1
    return n_scale


def is_in_frustum(F: Frame, pos, normal, mf_max_distance, mf_min_distance, viewing_cos_limit, log_scale_factor):
    """Frame::isInFrustum for one point: None, or the dict of MapPoint track fields it sets."""
    Pc = _rx_plus_t(F.mTcw, pos)
    if Pc[2] < F32(0.0):
        return None
    invz = F32(1.0) / Pc[2]
    u = F.fx * Pc[0] * invz + F.cx
    v = F.fy * Pc[1] * invz + F.cy
    if u < F.mnMinX or u > F.mnMaxX:
        return None
    if v < F.mnMinY or v > F.mnMaxY:
        return None
    max_distance = F32(1.2) * F32(mf_max_distance)  # GetMaxDistanceInvariance
    min_distance = F32(0.8) * F32(mf_min_distance)
    ow = _camera_center(F.mTcw)
    PO = [F32(pos[k]) - ow[k] for k in range(3)]
    dist = _norm3(PO)
    if dist < min_distance or dist > max_distance:
        return None
    dot = float(PO[0]) * float(normal[0]) + float(PO[1]) * float(normal[1]) + float(PO[2]) * float(normal[2])
    view_cos = F32(dot / float(dist))
    if view_cos < F32(viewing_cos_limit):
        return None
    level = predict_scale(mf_max_distance, dist, log_scale_factor, F.mnScaleLevels)
    return dict(proj_x=u, proj_xr=u - F.mbf * invz, proj_y=v, level=level, view_cos=view_cos)


def search_by_projection_points(F: Frame, points, frame_point_obs, th, nnratio):
    """points[i] = None or dict(track fields + desc + obs); frame_point_obs[k] = None or Observations() of the point keypoint
    k holds.  Returns (mvpMapPoints as point indices / -1 where untouched, nmatches)."""
    held = list(frame_point_obs)   # Observations() of F.mvpMapPoints[k], None if NULL
    assigned = [-1] * F.N
    nmatches = 0
    b_factor = F32(th) != 1.0
    for i_mp, p in enumerate(points):
        if p is None:  # !mbTrackInView or isBad()
            continue
        level = p["level"]
        r = F32(2.5) if p["view_cos"] > 0.998 else F32(4.0)
        if b_factor:
            r = r * F32(th)
        idxs = F.get_features_in_area(p["proj_x"], p["proj_y"], r * F.mvScaleFactors[level], level - 1, level)
        if not idxs:
            continue
        best_dist, best_level, best_dist2, best_level2, best_idx = 256, -1, 256, -1, -1
        for idx in idxs:
            if held[idx] is not None and held[idx] > 0:
                continue
            if F.mvuRight[idx] > 0:
                er = abs(F32(p["proj_xr"]) - F32(F.mvuRight[idx]))
                if er > r * F.mvScaleFactors[level]:
                    continue
            dist = descriptor_distance(p["desc"], F.mDescriptors[idx])
            if dist < best_dist:
                best_dist2 = best_dist; best_dist = dist
                best_level2 = best_level; best_level = int(F.mvKeysUn[idx]["octave"]); best_idx = idx
            elif dist < best_dist2:
                best_level2 = int(F.mvKeysUn[idx]["octave"]); best_dist2 = dist
        if best_dist <= TH_HIGH:
            if best_level == best_level2 and F32(best_dist) > F32(nnratio) * F32(best_dist2):
                continue
            held[best_idx] = p["obs"]
            assigned[best_idx] = i_mp
            nmatches += 1
    return np.array(assigned, np.int32), nmatches


def search_by_projection_last(Cur: Frame, T_last, last, frame_point_obs, th, mono, check_ori):
    """last = dict(pos, desc, valid, obs, octave, angle) with one row per LastFrame keypoint."""
    held = list(frame_point_obs)
    assigned = [-1] * Cur.N
    nmatches = 0
    rot_hist = [[] for _ in range(HISTO_LENGTH)]
    twc = _camera_center(Cur.mTcw)
    tlc = _rx_plus_t(np.asarray(T_last, np.float32), twc)
    forward = (tlc[2] > Cur.mb) and not mono
    backward = (-tlc[2] > Cur.mb) and not mono
    for i in range(len(last["valid"])):
        if not last["valid"][i]:  # pMP && !mvbOutlier[i]
            continue
        x3Dc = _rx_plus_t(Cur.mTcw, last["pos"][i])
        xc, yc = x3Dc[0], x3Dc[1]
        invzc = F32(1.0 / float(x3Dc[2]))
        if invzc < 0:
            continue
        u = Cur.fx * xc * invzc + Cur.cx
        v = Cur.fy * yc * invzc + Cur.cy
        if u < Cur.mnMinX or u > Cur.mnMaxX:
            continue
        if v < Cur.mnMinY or v > Cur.mnMaxY:
            continue
        n_last_octave = int(last["octave"][i])
        radius = F32(th) * Cur.mvScaleFactors[n_last_octave]
        if forward:
            idxs = Cur.get_features_in_area(u, v, radius, n_last_octave)
        elif backward:
            idxs = Cur.get_features_in_area(u, v, radius, 0, n_last_octave)
        else:
            idxs = Cur.get_features_in_area(u, v, radius, n_last_octave - 1, n_last_octave + 1)
        if not idxs:
            continue
        best_dist, best_idx2 = 256, -1
        for i2 in idxs:
            if held[i2] is not None and held[i2] > 0:
                continue
            if Cur.mvuRight[i2] > 0:
                ur = u - Cur.mbf * invzc
                er = abs(ur - F32(Cur.mvuRight[i2]))
                if er > radius:
                    continue
            dist = descriptor_distance(last["desc"][i], Cur.mDescriptors[i2])
            if dist < best_dist:
                best_dist = dist; best_idx2 = i2
        if best_dist <= TH_HIGH:
            held[best_idx2] = int(last["obs"][i])
            assigned[best_idx2] = i
            nmatches += 1
            if check_ori:
                rot_hist[_rot_bin(last["angle"][i], Cur.mvKeysUn[best_idx2]["angle"])].append(best_idx2)
    if check_ori:
        i1, i2_, i3 = compute_three_maxima(rot_hist, HISTO_LENGTH)
        for b in range(HISTO_LENGTH):
            if b != i1 and b != i2_ and b != i3:
                for idx in rot_hist[b]:
                    assigned[idx] = -1  # mvpMapPoints[idx] = NULL
                    nmatches -= 1
    return np.array(assigned, np.int32), nmatches


def search_by_projection_kf(Cur: Frame, kf, frame_has_point, th, orb_dist, check_ori, log_scale_factor):
    """kf = dict(pos, desc, valid, angle, max_distance, min_distance) (raw mfMax/MinDistance), one row per keyframe keypoint."""
    has = [bool(h) for h in frame_has_point]
    assigned = [-1] * Cur.N
    nmatches = 0
    rot_hist = [[] for _ in range(HISTO_LENGTH)]
    ow = _camera_center(Cur.mTcw)
    for i in range(len(kf["valid"])):
        if not kf["valid"][i]:
            continue
        x3Dc = _rx_plus_t(Cur.mTcw, kf["pos"][i])
        invzc = F32(1.0 / float(x3Dc[2]))
        u = Cur.fx * x3Dc[0] * invzc + Cur.cx
        v = Cur.fy * x3Dc[1] * invzc + Cur.cy
        if u < Cur.mnMinX or u > Cur.mnMaxX:
            continue
        if v < Cur.mnMinY or v > Cur.mnMaxY:
            continue
        PO = [F32(kf["pos"][i][k]) - ow[k] for k in range(3)]
        dist3d = _norm3(PO)
        max_distance = F32(1.2) * F32(kf["max_distance"][i])
        min_distance = F32(0.8) * F32(kf["min_distance"][i])
        if dist3d < min_distance or dist3d > max_distance:
            continue
        level = predict_scale(kf["max_distance"][i], dist3d, log_scale_factor, Cur.mnScaleLevels)
        radius = F32(th) * Cur.mvScaleFactors[level]
        idxs = Cur.get_features_in_area(u, v, radius, level - 1, level + 1)
        if not idxs:
            continue
        best_dist, best_idx2 = 256, -1
        for i2 in idxs:
            if has[i2]:
                continue
            dist = descriptor_distance(kf["desc"][i], Cur.mDescriptors[i2])
            if dist < best_dist:
                best_dist = dist; best_idx2 = i2
        if best_dist <= orb_dist:
            has[best_idx2] = True
            assigned[best_idx2] = i
            nmatches += 1
            if check_ori:
                rot_hist[_rot_bin(kf["angle"][i], Cur.mvKeysUn[best_idx2]["angle"])].append(best_idx2)
    if check_ori:
        i1, i2_, i3 = compute_three_maxima(rot_hist, HISTO_LENGTH)
        for b in range(HISTO_LENGTH):
            if b != i1 and b != i2_ and b != i3:
                for idx in rot_hist[b]:
                    assigned[idx] = -1
                    nmatches -= 1
    return np.array(assigned, np.int32), nmatches


def search_for_initialization(F1: Frame, F2: Frame, vb_prev_matched, window_size, nnratio, check_ori):
    """Returns (vnMatches12, vbPrevMatched after the update loop, nmatches)."""
    prev = np.array(vb_prev_matched, np.float32).reshape(-1, 2).copy()
    nmatches = 0
    m12 = [-1] * F1.N
    rot_hist = [[] for _ in range(HISTO_LENGTH)]
    matched_distance = [INT_MAX] * F2.N
    m21 = [-1] * F2.N
    for i1 in range(F1.N):
        level1 = int(F1.mvKeysUn[i1]["octave"])
        if level1 > 0:
            continue
        idxs = F2.get_features_in_area(prev[i1, 0], prev[i1, 1], window_size, level1, level1)
        if not idxs:
            continue
        best_dist, best_dist2, best_idx2 = INT_MAX, INT_MAX, -1
        for i2 in idxs:
            dist = descriptor_distance(F1.mDescriptors[i1], F2.mDescriptors[i2])
            if matched_distance[i2] <= dist:
                continue
            if dist < best_dist:
                best_dist2 = best_dist; best_dist = dist; best_idx2 = i2
            elif dist < best_dist2:
                best_dist2 = dist
        if best_dist <= TH_LOW:
            if F32(best_dist) < F32(best_dist2) * F32(nnratio):
                if m21[best_idx2] >= 0:
                    m12[m21[best_idx2]] = -1
                    nmatches -= 1
                m12[i1] = best_idx2
                m21[best_idx2] = i1
                matched_distance[best_idx2] = best_dist
                nmatches += 1
                if check_ori:
                    rot_hist[_rot_bin(F1.mvKeysUn[i1]["angle"], F2.mvKeysUn[best_idx2]["angle"])].append(i1)
    if check_ori:
        i1_, i2_, i3 = compute_three_maxima(rot_hist, HISTO_LENGTH)
        for b in range(HISTO_LENGTH):
            if b == i1_ or b == i2_ or b == i3:
                continue
            for idx1 in rot_hist[b]:
                if m12[idx1] >= 0:
                    m12[idx1] = -1
                    nmatches -= 1
    for i1 in range(F1.N):  # update prev matched (src/ORBmatcher.cc:508-511)
        if m12[i1] >= 0:
            prev[i1, 0] = F2.mvKeysUn[m12[i1]]["x"]; prev[i1, 1] = F2.mvKeysUn[m12[i1]]["y"]
    return np.array(m12, np.int32), prev, nmatches
