"""oracle/literal_kf_matchers.py -- TEST INFRASTRUCTURE (never imported by the product, bench.py's timed region or smoke()).

Literal Python transcription of the KeyFrame-side searches of LocalMapping / LoopClosing, on top of oracle/literal_matchers.py's
Frame (mGrid as the frame fills it): a KeyFrame takes the frame's grid and cell size as they are and keeps the image bounds as
INTS (src/KeyFrame.cc:32-50, include/KeyFrame.h:194-197), and its GetFeaturesInArea / IsInImage compute with those ints.
Independent of oracle/orb_oracle_match.c and orbslam2_amd/csrc/orbfe_match.hip (sorted candidate keys).  Small cases only.

Transcribed (reference file:line):
  KeyFrame::KeyFrame (grid, bounds)                       src/KeyFrame.cc:32-50
  KeyFrame::GetFeaturesInArea / IsInImage                 src/KeyFrame.cc:563-607
  MapPoint::PredictScale(dist, KeyFrame*)                 src/MapPoint.cc:385-400; Get{Min,Max}DistanceInvariance :373-383
  ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th)            src/ORBmatcher.cc:821-971 (search part: which keypoint each point gets)
  ORBmatcher::SearchByProjection(KeyFrame*, Scw, ...)     src/ORBmatcher.cc:285-398
  ORBmatcher::Fuse(KeyFrame*, Scw, ...)                   src/ORBmatcher.cc:973-1096 (search part)
Library semantics assumed (the contract of DESIGN.md section 2): CV_32F 3x3 * 3x1 + 3x1 as ((r0 x0 + r1 x1) + r2 x2) + t; cv::norm
and Mat::dot accumulate in double; Mat / scalar is a multiplication by (float)(1 / scalar); log() in double rounded to float.
"""
from __future__ import annotations

import math

import numpy as np

from .literal_matchers import F32, TH_LOW, Frame, _camera_center, _norm3, _rx_plus_t, descriptor_distance, predict_scale

INT_MAX = 2 ** 31 - 1


class KeyFrame:
    def __init__(self, F: Frame, inv_level_sigma2, log_scale_factor):
        self.mnGridCols, self.mnGridRows = len(F.mGrid), len(F.mGrid[0])
        self.mfGridElementWidthInv, self.mfGridElementHeightInv = F.mfGridElementWidthInv, F.mfGridElementHeightInv
        self.mGrid = F.mGrid                                   # mGrid[i][j] = F.mGrid[i][j]
        self.mnMinX, self.mnMinY = int(F.mnMinX), int(F.mnMinY)  # const int initialised from the frame's floats
        self.mnMaxX, self.mnMaxY = int(F.mnMaxX), int(F.mnMaxY)
        self.mvKeysUn, self.mvuRight, self.mDescriptors, self.N = F.mvKeysUn, F.mvuRight, F.mDescriptors, F.N
        self.fx, self.fy, self.cx, self.cy, self.mbf = F.fx, F.fy, F.cx, F.cy, F.mbf
        self.mvScaleFactors = F.mvScaleFactors
        self.mnScaleLevels = F.mnScaleLevels
        self.mvInvLevelSigma2 = np.asarray(inv_level_sigma2, np.float32)
        self.mfLogScaleFactor = F32(log_scale_factor)

    def GetFeaturesInArea(self, x, y, r):
        x, y, r = F32(x), F32(y), F32(r)
        vIndices = []
        nMinCellX = max(0, int(math.floor((x - F32(self.mnMinX) - r) * self.mfGridElementWidthInv)))
        if nMinCellX >= self.mnGridCols:
            return vIndices
        nMaxCellX = min(self.mnGridCols - 1, int(math.ceil((x - F32(self.mnMinX) + r) * self.mfGridElementWidthInv)))
        if nMaxCellX < 0:
            return vIndices
        nMinCellY = max(0, int(math.floor((y - F32(self.mnMinY) - r) * self.mfGridElementHeightInv)))
        if nMinCellY >= self.mnGridRows:
            return vIndices
        nMaxCellY = min(self.mnGridRows - 1, int(math.ceil((y - F32(self.mnMinY) + r) * self.mfGridElementHeightInv)))
        if nMaxCellY < 0:
            return vIndices
        for ix in range(nMinCellX, nMaxCellX + 1):
            for iy in range(nMinCellY, nMaxCellY + 1):
                for j in self.mGrid[ix][iy]:
                    kpUn = self.mvKeysUn[j]
                    distx = F32(kpUn["x"]) - x
                    disty = F32(kpUn["y"]) - y
                    if abs(distx) < r and abs(disty) < r:
                        vIndices.append(j)
        return vIndices

    def IsInImage(self, x, y):
        return x >= self.mnMinX and x < self.mnMaxX and y >= self.mnMinY and y < self.mnMaxY


def _predict_scale_kf(mfMaxDistance, currentDist, pKF):
    return predict_scale(mfMaxDistance, currentDist, pKF.mfLogScaleFactor, pKF.mnScaleLevels)


def fuse(pKF: KeyFrame, Tcw, points, th):
    """Search part of Fuse(KeyFrame*, vpMapPoints, th).  points: dict arrays pos, normal, max_distance (mfMaxDistance),
    min_distance (mfMinDistance), desc, valid (non-null, not bad, not yet in the keyframe).  -> (keypoint per point or -1, nFused)."""
    Tcw = np.asarray(Tcw, np.float32)
    Ow = _camera_center(Tcw)
    fx, fy, cx, cy, bf = pKF.fx, pKF.fy, pKF.cx, pKF.cy, pKF.mbf
    out = np.full(len(points["valid"]), -1, np.int32)
    nFused = 0
    for i in range(len(out)):
        if not points["valid"][i]:
            continue
        p3Dw = [F32(c) for c in points["pos"][i]]
        p3Dc = _rx_plus_t(Tcw, p3Dw)
        if p3Dc[2] < F32(0.0):
            continue
        invz = F32(1) / p3Dc[2]
        x = p3Dc[0] * invz
        y = p3Dc[1] * invz
        u = fx * x + cx
        v = fy * y + cy
        if not pKF.IsInImage(u, v):
            continue
        ur = u - bf * invz
        maxDistance = F32(1.2) * F32(points["max_distance"][i])
        minDistance = F32(0.8) * F32(points["min_distance"][i])
        PO = [p3Dw[k] - Ow[k] for k in range(3)]
        dist3D = _norm3(PO)
        if dist3D < minDistance or dist3D > maxDistance:
            continue
        Pn = points["normal"][i]
        if float(PO[0]) * float(F32(Pn[0])) + float(PO[1]) * float(F32(Pn[1])) + float(PO[2]) * float(F32(Pn[2])) < 0.5 * float(dist3D):
            continue
        nPredictedLevel = _predict_scale_kf(points["max_distance"][i], dist3D, pKF)
        radius = F32(th) * pKF.mvScaleFactors[nPredictedLevel]
        vIndices = pKF.GetFeaturesInArea(u, v, radius)
        if not vIndices:
            continue
        dMP = points["desc"][i]
        bestDist, bestIdx = 256, -1
        for idx in vIndices:
            kp = pKF.mvKeysUn[idx]
            kpLevel = int(kp["octave"])
            if kpLevel < nPredictedLevel - 1 or kpLevel > nPredictedLevel:
                continue
            if pKF.mvuRight[idx] >= 0:
                ex = u - F32(kp["x"]); ey = v - F32(kp["y"]); er = ur - F32(pKF.mvuRight[idx])
                e2 = ex * ex + ey * ey + er * er
                if float(e2 * pKF.mvInvLevelSigma2[kpLevel]) > 7.8:
                    continue
            else:
                ex = u - F32(kp["x"]); ey = v - F32(kp["y"])
                e2 = ex * ex + ey * ey
                if float(e2 * pKF.mvInvLevelSigma2[kpLevel]) > 5.99:
                    continue
            dist = descriptor_distance(dMP, pKF.mDescriptors[idx])
            if dist < bestDist:
                bestDist, bestIdx = dist, idx
        if bestDist <= TH_LOW:
            out[i] = bestIdx
            nFused += 1
    return out, nFused


def _decompose_sim3(Scw):
    Scw = np.asarray(Scw, np.float32)
    scw = F32(math.sqrt(float(Scw[0, 0]) * float(Scw[0, 0]) + float(Scw[0, 1]) * float(Scw[0, 1]) + float(Scw[0, 2]) * float(Scw[0, 2])))
    alpha = F32(1.0 / float(scw))
    T = (Scw * alpha).astype(np.float32)  # Rcw = sRcw / scw, tcw = Scw.col(3) / scw
    return T, _camera_center(T)


def _project(pKF, T, Ow, points, i, invz_in_double):
    p3Dw = [F32(c) for c in points["pos"][i]]
    p3Dc = _rx_plus_t(T, p3Dw)
    if p3Dc[2] < 0.0:
        return None
    invz = F32(1.0 / float(p3Dc[2])) if invz_in_double else F32(1) / p3Dc[2]
    x = p3Dc[0] * invz
    y = p3Dc[1] * invz
    u = pKF.fx * x + pKF.cx
    v = pKF.fy * y + pKF.cy
    if not pKF.IsInImage(u, v):
        return None
    maxDistance = F32(1.2) * F32(points["max_distance"][i])
    minDistance = F32(0.8) * F32(points["min_distance"][i])
    PO = [p3Dw[k] - Ow[k] for k in range(3)]
    dist = _norm3(PO)
    if dist < minDistance or dist > maxDistance:
        return None
    Pn = points["normal"][i]
    if float(PO[0]) * float(F32(Pn[0])) + float(PO[1]) * float(F32(Pn[1])) + float(PO[2]) * float(F32(Pn[2])) < 0.5 * float(dist):
        return None
    return u, v, dist


def search_by_projection_sim3(pKF: KeyFrame, Scw, points, kf_matched, th):
    """SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th): points greedy in order, a matched keypoint is skipped."""
    T, Ow = _decompose_sim3(Scw)
    vpMatched = [bool(m) for m in kf_matched]
    out = np.full(len(points["valid"]), -1, np.int32)
    nmatches = 0
    for iMP in range(len(out)):
        if not points["valid"][iMP]:  # isBad() || spAlreadyFound.count(pMP)
            continue
        pr = _project(pKF, T, Ow, points, iMP, invz_in_double=False)
        if pr is None:
            continue
        u, v, dist = pr
        nPredictedLevel = _predict_scale_kf(points["max_distance"][iMP], dist, pKF)
        radius = F32(th) * pKF.mvScaleFactors[nPredictedLevel]
        vIndices = pKF.GetFeaturesInArea(u, v, radius)
        if not vIndices:
            continue
        dMP = points["desc"][iMP]
        bestDist, bestIdx = 256, -1
        for idx in vIndices:
            if vpMatched[idx]:
                continue
            kpLevel = int(pKF.mvKeysUn[idx]["octave"])
            if kpLevel < nPredictedLevel - 1 or kpLevel > nPredictedLevel:
                continue
            d = descriptor_distance(dMP, pKF.mDescriptors[idx])
            if d < bestDist:
                bestDist, bestIdx = d, idx
        if bestDist <= TH_LOW:
            vpMatched[bestIdx] = True
            out[iMP] = bestIdx
            nmatches += 1
    return out, nmatches


def fuse_sim3(pKF: KeyFrame, Scw, points, th):
    """Search part of Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint): points independent."""
    T, Ow = _decompose_sim3(Scw)
    out = np.full(len(points["valid"]), -1, np.int32)
    nFused = 0
    for iMP in range(len(out)):
        if not points["valid"][iMP]:
            continue
        pr = _project(pKF, T, Ow, points, iMP, invz_in_double=True)
        if pr is None:
            continue
        u, v, dist3D = pr
        nPredictedLevel = _predict_scale_kf(points["max_distance"][iMP], dist3D, pKF)
        radius = F32(th) * pKF.mvScaleFactors[nPredictedLevel]
        vIndices = pKF.GetFeaturesInArea(u, v, radius)
        if not vIndices:
            continue
        dMP = points["desc"][iMP]
        bestDist, bestIdx = INT_MAX, -1
        for idx in vIndices:
            kpLevel = int(pKF.mvKeysUn[idx]["octave"])
            if kpLevel < nPredictedLevel - 1 or kpLevel > nPredictedLevel:
                continue
            d = descriptor_distance(dMP, pKF.mDescriptors[idx])
            if d < bestDist:
                bestDist, bestIdx = d, idx
        if bestDist <= TH_LOW:
            out[iMP] = bestIdx
            nFused += 1
    return out, nFused


TH_HIGH = 100  # src/ORBmatcher.cc:35


def search_by_sim3(pKF1: KeyFrame, T1w, pts1, pKF2: KeyFrame, T2w, pts2, s12, R12, t12, th):
    """ORBmatcher::SearchBySim3, src/ORBmatcher.cc:1098-1322.  pts*: dict arrays with one entry per keypoint slot of the keyframe:
    pos, max_distance (mfMaxDistance), min_distance, desc, valid (map point non-null, not bad, not already matched).
    -> (match12[N1] = keypoint of keyframe 2 or -1, nFound).  Both keyframes share the camera of pKF1 (as the reference reads it)."""
    T1w = np.asarray(T1w, np.float32); T2w = np.asarray(T2w, np.float32)
    R12 = np.asarray(R12, np.float32).reshape(3, 3); t12 = np.asarray(t12, np.float32).reshape(3)
    s12 = F32(s12)
    # sR12 = s12 * R12;  sR21 = (1.0 / s12) * R12.t();  t21 = -sR21 * t12  (a scaled Mat is a multiplication by (float)alpha)
    sR12 = (R12 * s12).astype(np.float32)
    inv_s = F32(1.0 / float(s12))
    sR21 = (R12.T * inv_s).astype(np.float32)
    t21 = [-((sR21[i, 0] * t12[0] + sR21[i, 1] * t12[1]) + sR21[i, 2] * t12[2]) for i in range(3)]
    A21 = np.concatenate([sR21, np.array(t21, np.float32).reshape(3, 1)], axis=1)
    A12 = np.concatenate([sR12, t12.reshape(3, 1)], axis=1)
    fx, fy, cx, cy = pKF1.fx, pKF1.fy, pKF1.cx, pKF1.cy

    def one_way(Taw, A, pts, pKFb):
        n = len(pts["valid"])
        vnMatch = [-1] * n
        for i in range(n):
            if not pts["valid"][i]:
                continue
            p3Dw = [F32(c) for c in pts["pos"][i]]
            p3Dca = _rx_plus_t(Taw, p3Dw)
            p3Dcb = _rx_plus_t(A, p3Dca)
            if p3Dcb[2] < 0.0:
                continue
            invz = F32(1.0 / float(p3Dcb[2]))
            x = p3Dcb[0] * invz
            y = p3Dcb[1] * invz
            u = fx * x + cx
            v = fy * y + cy
            if not pKFb.IsInImage(u, v):
                continue
            maxDistance = F32(1.2) * F32(pts["max_distance"][i])
            minDistance = F32(0.8) * F32(pts["min_distance"][i])
            dist3D = _norm3(p3Dcb)
            if dist3D < minDistance or dist3D > maxDistance:
                continue
            nPredictedLevel = _predict_scale_kf(pts["max_distance"][i], dist3D, pKFb)
            radius = F32(th) * pKFb.mvScaleFactors[nPredictedLevel]
            vIndices = pKFb.GetFeaturesInArea(u, v, radius)
            if not vIndices:
                continue
            dMP = pts["desc"][i]
            bestDist, bestIdx = INT_MAX, -1
            for idx in vIndices:
                octave = int(pKFb.mvKeysUn[idx]["octave"])
                if octave < nPredictedLevel - 1 or octave > nPredictedLevel:
                    continue
                d = descriptor_distance(dMP, pKFb.mDescriptors[idx])
                if d < bestDist:
                    bestDist, bestIdx = d, idx
            if bestDist <= TH_HIGH:
                vnMatch[i] = bestIdx
        return vnMatch

    vnMatch1 = one_way(T1w, A21, pts1, pKF2)
    vnMatch2 = one_way(T2w, A12, pts2, pKF1)
    match12 = np.full(len(vnMatch1), -1, np.int32)
    nFound = 0
    for i1 in range(len(vnMatch1)):
        idx2 = vnMatch1[i1]
        if idx2 >= 0:
            if vnMatch2[idx2] == i1:
                match12[i1] = idx2
                nFound += 1
    return match12, nFound
