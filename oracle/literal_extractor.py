"""oracle/literal_extractor.py -- TEST INFRASTRUCTURE (never imported by the product, bench.py's timed region or smoke()).

A second, independent statement of the reference's OWN extractor and stereo logic: the loops of ORBextractor.cc and
Frame::ComputeStereoMatches transcribed LITERALLY into Python (numpy float32 scalars, one IEEE rounding per operation as the
C++ float expressions have under -ffp-contract=off; std::list of nodes with push_front / erase exactly as the reference
manipulates it; a vector of candidate rows per image row; the 11x11 SAD as a plain double loop).  The OpenCV LIBRARY calls the
reference makes -- cv::resize, cv::GaussianBlur, cv::FAST, cv::fastAtan2, cvRound -- are not restated here: they are taken as
primitives from the C oracle (oracle/orb_oracle.c, pinned or unpinned as its header says), because this file is about the
reference's control flow, not OpenCV's arithmetic.  What it shares with the C oracle is therefore exactly those primitives
and the pattern table (data); the cell loop, the quadtree, the orientation / descriptor loops, the output assembly and the
whole stereo search are written a second time, in another language, from the reference text.  `C oracle == this file` on
whole images then says the C restatement's array formulation of those loops (sorted candidate arrays, index lists instead of
std::list, precomputed row tables) did not change their meaning.

Transcribed functions (reference file:line):
  ORBextractor::ORBextractor                    src/ORBextractor.cc:405-464
  IC_Angle / computeOrientation                 src/ORBextractor.cc:72-100, 466-473
  computeOrbDescriptor / computeDescriptors     src/ORBextractor.cc:103-142, 848-856
  ExtractorNode::DivideNode                     src/ORBextractor.cc:475-531
  ORBextractor::DistributeOctTree               src/ORBextractor.cc:533-757
  ORBextractor::ComputeKeyPointsOctTree         src/ORBextractor.cc:759-846
  ORBextractor::operator()                      src/ORBextractor.cc:858-919
  ORBextractor::ComputePyramid                  src/ORBextractor.cc:921-946
  Frame::ComputeStereoMatches                   src/Frame.cc:464-642
Two places where the reference's result depends on something outside its text, resolved as the C oracle's header states them:
std::sort of pair<int, ExtractorNode*> orders equal sizes by heap ADDRESS (contract Q3: creation order is used instead), and
cos / sin of the keypoint angle come from libm (contract Q4: orc_sincos_det).  Pure-Python loops: small images only.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import oracle as O

F32 = np.float32
INT_MAX = 2 ** 31 - 1
TH_HIGH, TH_LOW = 100, 50  # src/ORBmatcher.cc:35-36


def _c_round(x) -> int:
    """C round(): half away from zero."""
    x = float(x)
    return int(math.floor(x + 0.5)) if x >= 0 else -int(math.floor(-x + 0.5))


def _cv_round(x) -> int:
    return int(O.lib().orc_cv_round_d(float(x)))


def _trunc(x) -> int:
    """float -> int conversion of C (toward zero)."""
    return int(float(x))


class KeyPoint:
    """cv::KeyPoint as the reference uses it."""
    __slots__ = ("x", "y", "size", "angle", "response", "octave")

    def __init__(self, x, y, size=7.0, angle=-1.0, response=0.0, octave=0):
        self.x, self.y = F32(x), F32(y)
        self.size, self.angle, self.response, self.octave = F32(size), F32(angle), F32(response), int(octave)

    def copy(self):
        return KeyPoint(self.x, self.y, self.size, self.angle, self.response, self.octave)


def cv_fast(img: np.ndarray, threshold: int):
    """cv::FAST(image, keypoints, threshold, true): KeyPoint(x, y, 7.f, -1, score), raster order (library primitive)."""
    xs, ys, ss = O.fast9_16(img, threshold, True)
    return [KeyPoint(float(x), float(y), 7.0, -1.0, float(s), 0) for x, y, s in zip(xs, ys, ss)]


class ExtractorNode:
    """include/ORBextractor.h:32-43."""

    def __init__(self):
        self.vKeys = []
        self.UL = self.UR = self.BL = self.BR = (0, 0)
        self.bNoMore = False
        self.seq = -1  # creation order: stands in for the node's address in the (size, pointer) sort

    def divide_node(self):
        """src/ORBextractor.cc:475-531."""
        n1, n2, n3, n4 = ExtractorNode(), ExtractorNode(), ExtractorNode(), ExtractorNode()
        halfX = int(math.ceil(F32(self.UR[0] - self.UL[0]) / F32(2)))
        halfY = int(math.ceil(F32(self.BR[1] - self.UL[1]) / F32(2)))
        n1.UL = self.UL
        n1.UR = (self.UL[0] + halfX, self.UL[1])
        n1.BL = (self.UL[0], self.UL[1] + halfY)
        n1.BR = (self.UL[0] + halfX, self.UL[1] + halfY)
        n2.UL = n1.UR
        n2.UR = self.UR
        n2.BL = n1.BR
        n2.BR = (self.UR[0], self.UL[1] + halfY)
        n3.UL = n1.BL
        n3.UR = n1.BR
        n3.BL = self.BL
        n3.BR = (n1.BR[0], self.BL[1])
        n4.UL = n3.UR
        n4.UR = n2.BR
        n4.BL = n3.BR
        n4.BR = self.BR
        for kp in self.vKeys:
            if kp.x < F32(n1.UR[0]):
                if kp.y < F32(n1.BR[1]):
                    n1.vKeys.append(kp)
                else:
                    n3.vKeys.append(kp)
            elif kp.y < F32(n1.BR[1]):
                n2.vKeys.append(kp)
            else:
                n4.vKeys.append(kp)
        for n in (n1, n2, n3, n4):
            if len(n.vKeys) == 1:
                n.bNoMore = True
        return n1, n2, n3, n4


class _List:
    """std::list<ExtractorNode> reduced to what DistributeOctTree does with it: push_back, push_front, erase by node,
    forward iteration that tolerates both.  A Python list with front = index 0."""

    def __init__(self):
        self.items = []

    def __len__(self):
        return len(self.items)

    def push_back(self, n):
        self.items.append(n)

    def push_front(self, n):
        self.items.insert(0, n)

    def erase(self, n):
        for i, m in enumerate(self.items):
            if m is n:
                del self.items[i]
                return i
        raise AssertionError("node not in list")


class LiteralExtractor:
    def __init__(self, nfeatures=2000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, patchSize=31, halfPatchSize=15,
                 edgeThreshold=19):
        """src/ORBextractor.cc:405-464."""
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self.iniThFAST, self.minThFAST = iniThFAST, minThFAST
        self.patchSize, self.halfPatchSize, self.edgeThreshold = patchSize, halfPatchSize, edgeThreshold
        # `double scaleFactor` member initialised from the float argument (include/ORBextractor.h:96)
        self.scaleFactor = float(F32(scaleFactor))
        self.mvScaleFactor = [F32(0)] * nlevels
        self.mvLevelSigma2 = [F32(0)] * nlevels
        self.mvScaleFactor[0] = F32(1.0)
        self.mvLevelSigma2[0] = F32(1.0)
        for i in range(1, nlevels):
            # float * double -> double -> float
            self.mvScaleFactor[i] = F32(float(self.mvScaleFactor[i - 1]) * self.scaleFactor)
            self.mvLevelSigma2[i] = self.mvScaleFactor[i] * self.mvScaleFactor[i]
        self.mvInvScaleFactor = [F32(1.0) / s for s in self.mvScaleFactor]
        self.mvInvLevelSigma2 = [F32(1.0) / s for s in self.mvLevelSigma2]
        self.mvImagePyramid = [None] * nlevels

        self.mnFeaturesPerLevel = [0] * nlevels
        factor = F32(1.0 / self.scaleFactor)  # float factor = 1.0f / scaleFactor (double division, stored to float)
        # nfeatures*(1 - factor)/(1 - (float)pow((double)factor, (double)nlevels)), all float
        nDesired = F32(nfeatures) * (F32(1) - factor) / (F32(1) - F32(math.pow(float(factor), float(nlevels))))
        sumFeatures = 0
        for level in range(nlevels - 1):
            self.mnFeaturesPerLevel[level] = _cv_round(nDesired)
            sumFeatures += self.mnFeaturesPerLevel[level]
            nDesired = nDesired * factor
        self.mnFeaturesPerLevel[nlevels - 1] = max(nfeatures - sumFeatures, 0)

        p = O.lib().orc_bit_pattern()  # OpenCV's bit_pattern_31_ (data)
        self.pattern = [(p[2 * i], p[2 * i + 1]) for i in range(512)]

        self.umax = [0] * (halfPatchSize + 1)
        sqrt2f = F32(math.sqrt(2.0))  # sqrt(2.f)
        vmax = int(math.floor(float(F32(halfPatchSize) * sqrt2f / F32(2) + F32(1))))
        vmin = int(math.ceil(float(F32(halfPatchSize) * sqrt2f / F32(2))))
        hp2 = float(halfPatchSize * halfPatchSize)
        for v in range(vmax + 1):
            self.umax[v] = _cv_round(math.sqrt(hp2 - v * v))
        v, v0 = halfPatchSize, 0
        while v >= vmin:
            while self.umax[v0] == self.umax[v0 + 1]:
                v0 += 1
            self.umax[v] = v0
            v0 += 1
            v -= 1

    # ------------------------------------------------------------------ pyramid
    def ComputePyramid(self, image: np.ndarray):
        """src/ORBextractor.cc:921-946.  The reflected border the reference writes around every level is never read by the
        functions below (cells start 16 px inside, patches reach 15 px), so the levels are kept without it."""
        for level in range(self.nlevels):
            scale = self.mvInvScaleFactor[level]
            w = _cv_round(F32(image.shape[1]) * scale)
            h = _cv_round(F32(image.shape[0]) * scale)
            if level != 0:
                self.mvImagePyramid[level] = O.resize_linear(self.mvImagePyramid[level - 1], w, h)
            else:
                self.mvImagePyramid[level] = np.ascontiguousarray(image).copy()

    # ------------------------------------------------------------------ quadtree
    def DistributeOctTree(self, vToDistributeKeys, minX, maxX, minY, maxY, N, level):
        """src/ORBextractor.cc:533-757."""
        nIni = _c_round(F32(maxX - minX) / F32(maxY - minY))
        hX = F32(maxX - minX) / F32(nIni)
        lNodes = _List()
        vpIniNodes = []
        seq = 0
        for i in range(nIni):
            ni = ExtractorNode()
            ni.UL = (_trunc(hX * F32(i)), 0)
            ni.UR = (_trunc(hX * F32(i + 1)), 0)
            ni.BL = (ni.UL[0], maxY - minY)
            ni.BR = (ni.UR[0], maxY - minY)
            ni.seq = seq
            seq += 1
            lNodes.push_back(ni)
            vpIniNodes.append(ni)
        for kp in vToDistributeKeys:
            vpIniNodes[_trunc(kp.x / hX)].vKeys.append(kp)

        i = 0
        while i < len(lNodes.items):
            n = lNodes.items[i]
            if len(n.vKeys) == 1:
                n.bNoMore = True
                i += 1
            elif not n.vKeys:
                del lNodes.items[i]
            else:
                i += 1

        bFinish = False
        vSizeAndPointerToNode = []

        def add_children(children, count_expand):
            nonlocal seq
            nexp = 0
            for c in children:
                if len(c.vKeys) > 0:
                    c.seq = seq
                    seq += 1
                    lNodes.push_front(c)
                    if len(c.vKeys) > 1:
                        nexp += 1
                        vSizeAndPointerToNode.append((len(c.vKeys), c))
            return nexp

        while not bFinish:
            prevSize = len(lNodes)
            nToExpand = 0
            vSizeAndPointerToNode.clear()
            # iterate the nodes that were in the list when the pass started, front to back; children go to the front and are
            # therefore not visited in this pass
            for n in list(lNodes.items):
                if n.bNoMore:
                    continue
                nToExpand += add_children(n.divide_node(), True)
                lNodes.erase(n)

            if len(lNodes) >= N or len(lNodes) == prevSize:
                bFinish = True
            elif len(lNodes) + nToExpand * 3 > N:
                while not bFinish:
                    prevSize = len(lNodes)
                    vPrev = list(vSizeAndPointerToNode)
                    vSizeAndPointerToNode.clear()
                    vPrev.sort(key=lambda t: (t[0], t[1].seq))  # contract Q3
                    for j in range(len(vPrev) - 1, -1, -1):
                        node = vPrev[j][1]
                        add_children(node.divide_node(), False)
                        lNodes.erase(node)
                        if len(lNodes) >= N:
                            break
                    if len(lNodes) >= N or len(lNodes) == prevSize:
                        bFinish = True

        vResultKeys = []
        for n in lNodes.items:
            pKP = n.vKeys[0]
            maxResponse = pKP.response
            for k in range(1, len(n.vKeys)):
                if n.vKeys[k].response > maxResponse:
                    pKP = n.vKeys[k]
                    maxResponse = n.vKeys[k].response
            vResultKeys.append(pKP.copy())
        return vResultKeys

    # ------------------------------------------------------------------ detection
    def ComputeKeyPointsOctTree(self):
        """src/ORBextractor.cc:759-846."""
        allKeypoints = [[] for _ in range(self.nlevels)]
        W = F32(30)
        for level in range(self.nlevels):
            im = self.mvImagePyramid[level]
            minBorderX = self.edgeThreshold - 3
            minBorderY = minBorderX
            maxBorderX = im.shape[1] - self.edgeThreshold + 3
            maxBorderY = im.shape[0] - self.edgeThreshold + 3
            vToDistributeKeys = []
            width = F32(maxBorderX - minBorderX)
            height = F32(maxBorderY - minBorderY)
            nCols = _trunc(width / W)
            nRows = _trunc(height / W)
            wCell = int(math.ceil(width / F32(nCols)))
            hCell = int(math.ceil(height / F32(nRows)))
            for i in range(nRows):
                iniY = F32(minBorderY + i * hCell)
                maxY = iniY + F32(hCell) + F32(6)
                if iniY >= F32(maxBorderY - 3):
                    continue
                if maxY > F32(maxBorderY):
                    maxY = F32(maxBorderY)
                for j in range(nCols):
                    iniX = F32(minBorderX + j * wCell)
                    maxX = iniX + F32(wCell) + F32(6)
                    if iniX >= F32(maxBorderX - 6):
                        continue
                    if maxX > F32(maxBorderX):
                        maxX = F32(maxBorderX)
                    cell = im[_trunc(iniY):_trunc(maxY), _trunc(iniX):_trunc(maxX)]
                    vKeysCell = cv_fast(cell, self.iniThFAST)
                    if not vKeysCell:
                        vKeysCell = cv_fast(cell, self.minThFAST)
                    for kp in vKeysCell:
                        kp.x = kp.x + F32(j * wCell)
                        kp.y = kp.y + F32(i * hCell)
                        vToDistributeKeys.append(kp)
            keypoints = self.DistributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                               self.mnFeaturesPerLevel[level], level)
            scaledPatchSize = _trunc(F32(self.patchSize) * self.mvScaleFactor[level])
            for kp in keypoints:
                kp.x = kp.x + F32(minBorderX)
                kp.y = kp.y + F32(minBorderY)
                kp.octave = level
                kp.size = F32(scaledPatchSize)
            allKeypoints[level] = keypoints
        for level in range(self.nlevels):
            for kp in allKeypoints[level]:  # computeOrientation, :466-473
                kp.angle = self.IC_Angle(self.mvImagePyramid[level], kp)
        return allKeypoints

    def IC_Angle(self, image, kp):
        """src/ORBextractor.cc:72-100."""
        m_01 = m_10 = 0
        cy, cx = _cv_round(kp.y), _cv_round(kp.x)
        for u in range(-self.halfPatchSize, self.halfPatchSize + 1):
            m_10 += u * int(image[cy, cx + u])
        for v in range(1, self.halfPatchSize + 1):
            v_sum = 0
            d = self.umax[v]
            for u in range(-d, d + 1):
                val_plus, val_minus = int(image[cy + v, cx + u]), int(image[cy - v, cx + u])
                v_sum += val_plus - val_minus
                m_10 += u * (val_plus + val_minus)
            m_01 += v * v_sum
        return F32(O.lib().orc_fast_atan2(float(F32(m_01)), float(F32(m_10))))

    def computeOrbDescriptor(self, kp, img):
        """src/ORBextractor.cc:103-142."""
        factorPI = F32(math.pi / float(F32(180.0)))
        angle = kp.angle * factorPI
        s, c = C.c_float(), C.c_float()
        O.lib().orc_sincos_det(float(angle), C.byref(s), C.byref(c))
        a, b = F32(c.value), F32(s.value)
        cy, cx = _cv_round(kp.y), _cv_round(kp.x)

        def get_value(px, py):
            r = _cv_round(F32(px) * b + F32(py) * a)
            q = _cv_round(F32(px) * a - F32(py) * b)
            return int(img[cy + r, cx + q])

        desc = np.zeros(32, np.uint8)
        pat = self.pattern
        for i in range(32):
            val = 0
            for bit in range(8):
                t0 = get_value(*pat[16 * i + 2 * bit])
                t1 = get_value(*pat[16 * i + 2 * bit + 1])
                val |= int(t0 < t1) << bit
            desc[i] = val
        return desc

    # ------------------------------------------------------------------ operator()
    def __call__(self, image: np.ndarray):
        """src/ORBextractor.cc:858-919 -> (list of KeyPoint, (n, 32) uint8)."""
        self.ComputePyramid(image)
        allKeypoints = self.ComputeKeyPointsOctTree()
        out_keys, out_desc = [], []
        for level in range(self.nlevels):
            keypoints = allKeypoints[level]
            if not keypoints:
                continue
            workingMat = O.gaussian7(self.mvImagePyramid[level])
            for kp in keypoints:
                out_desc.append(self.computeOrbDescriptor(kp, workingMat))
            if level != 0:
                scale = self.mvScaleFactor[level]
                for kp in keypoints:
                    kp.x = kp.x * scale
                    kp.y = kp.y * scale
            out_keys.extend(keypoints)
        desc = np.stack(out_desc) if out_desc else np.zeros((0, 32), np.uint8)
        return out_keys, desc


def descriptor_distance(a: np.ndarray, b: np.ndarray) -> int:
    """src/ORBmatcher.cc:1643-1659."""
    pa = np.frombuffer(np.ascontiguousarray(a, np.uint8).tobytes(), "<u4")
    pb = np.frombuffer(np.ascontiguousarray(b, np.uint8).tobytes(), "<u4")
    dist = 0
    for i in range(8):
        v = int(pa[i]) ^ int(pb[i])
        v = v - ((v >> 1) & 0x55555555)
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333)
        dist += ((((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) & 0xFFFFFFFF) >> 24
    return dist


def compute_stereo_matches(exL: LiteralExtractor, exR: LiteralExtractor, mvKeys, mDescriptors, mvKeysRight, mDescriptorsRight,
                           mbf, fx):
    """Frame::ComputeStereoMatches, src/Frame.cc:464-642 -> (mvuRight, mvDepth) float32 arrays."""
    N = len(mvKeys)
    mbf = F32(mbf)
    mb = mbf / F32(fx)  # src/Frame.cc:112
    mvuRight = np.full(N, -1.0, np.float32)
    mvDepth = np.full(N, -1.0, np.float32)
    thOrbDist = (TH_HIGH + TH_LOW) // 2
    nRows = exL.mvImagePyramid[0].shape[0]
    vRowIndices = [[] for _ in range(nRows)]
    mvScaleFactors, mvInvScaleFactors = exL.mvScaleFactor, exL.mvInvScaleFactor
    for iR, kp in enumerate(mvKeysRight):
        kpY = kp.y
        r = F32(2.0) * mvScaleFactors[kp.octave]
        maxr = int(math.ceil(kpY + r))
        minr = int(math.floor(kpY - r))
        for yi in range(minr, maxr + 1):
            vRowIndices[yi].append(iR)
    if mb == 0:
        return mvuRight, mvDepth
    minZ = mb
    minD = F32(0)
    maxD = mbf / minZ
    vDistIdx = []
    with np.errstate(all="ignore"):
        for iL in range(N):
            kpL = mvKeys[iL]
            levelL, vL, uL = kpL.octave, kpL.y, kpL.x
            vCandidates = vRowIndices[_trunc(vL)]
            if not vCandidates:
                continue
            minU = uL - maxD
            maxU = uL - minD
            if maxU < 0:
                continue
            bestDist = TH_HIGH
            bestIdxR = 0
            dL = mDescriptors[iL]
            for iR in vCandidates:
                kpR = mvKeysRight[iR]
                if kpR.octave < levelL - 1 or kpR.octave > levelL + 1:
                    continue
                uR = kpR.x
                if uR >= minU and uR <= maxU:
                    dist = descriptor_distance(dL, mDescriptorsRight[iR])
                    if dist < bestDist:
                        bestDist = dist
                        bestIdxR = iR
            if bestDist < thOrbDist:
                uR0 = mvKeysRight[bestIdxR].x
                scaleFactor = mvInvScaleFactors[kpL.octave]
                scaleduL = F32(_c_round(kpL.x * scaleFactor))
                scaledvL = F32(_c_round(kpL.y * scaleFactor))
                scaleduR0 = F32(_c_round(uR0 * scaleFactor))
                w = 5
                pyrL = exL.mvImagePyramid[kpL.octave]
                pyrR = exR.mvImagePyramid[kpL.octave]
                r0, c0 = _trunc(scaledvL - w), _trunc(scaleduL - w)
                IL = [[F32(pyrL[r0 + y, c0 + x]) for x in range(2 * w + 1)] for y in range(2 * w + 1)]
                centre = IL[w][w]
                IL = [[v - centre * F32(1) for v in row] for row in IL]
                bestDistS = INT_MAX
                bestincR = 0
                L = 5
                vDists = [F32(0)] * (2 * L + 1)
                iniu = scaleduR0 + F32(L) - F32(w)
                endu = scaleduR0 + F32(L) + F32(w) + F32(1)
                if iniu < 0 or endu >= pyrR.shape[1]:
                    continue
                # Q12: where the reference's colRange / rowRange would throw (a window outside the level image; only with
                # scaleFactor > 1.9) the keypoint stays unmatched
                cu_, cv_, cr_ = _trunc(scaleduL), _trunc(scaledvL), _trunc(scaleduR0)
                if cu_ - 5 < 0 or cu_ + 5 >= pyrL.shape[1] or cv_ - 5 < 0 or cv_ + 5 >= pyrL.shape[0] or cr_ - 10 < 0 or cr_ + 10 >= pyrR.shape[1]:
                    continue
                for incR in range(-L, L + 1):
                    cr = _trunc(scaleduR0 + F32(incR) - F32(w))
                    IR = [[F32(pyrR[r0 + y, cr + x]) for x in range(2 * w + 1)] for y in range(2 * w + 1)]
                    centreR = IR[w][w]
                    acc = 0.0  # cv::norm(IL, IR, NORM_L1) of CV_32F accumulates in double
                    for y in range(2 * w + 1):
                        for x in range(2 * w + 1):
                            acc += float(abs(IL[y][x] - (IR[y][x] - centreR * F32(1))))
                    dist = F32(acc)
                    if dist < F32(bestDistS):
                        bestDistS = _trunc(dist)
                        bestincR = incR
                    vDists[L + incR] = dist
                if bestincR == -L or bestincR == L:
                    continue
                dist1, dist2, dist3 = vDists[L + bestincR - 1], vDists[L + bestincR], vDists[L + bestincR + 1]
                deltaR = (dist1 - dist3) / (F32(2.0) * (dist1 + dist3 - F32(2.0) * dist2))
                if deltaR < -1 or deltaR > 1:
                    continue
                bestuR = mvScaleFactors[kpL.octave] * (scaleduR0 + F32(bestincR) + deltaR)
                disparity = uL - bestuR
                if disparity >= minD and disparity < maxD:
                    if disparity <= 0:
                        disparity = F32(0.01)
                        bestuR = F32(float(uL) - 0.01)
                    mvDepth[iL] = mbf / disparity
                    mvuRight[iL] = bestuR
                    vDistIdx.append((bestDistS, iL))
    if not vDistIdx:  # the reference indexes an empty vector here; nothing to cull
        return mvuRight, mvDepth
    vDistIdx.sort()
    median = F32(vDistIdx[len(vDistIdx) // 2][0])
    thDist = F32(1.5) * F32(1.4) * median
    for i in range(len(vDistIdx) - 1, -1, -1):
        if F32(vDistIdx[i][0]) < thDist:
            break
        mvuRight[vDistIdx[i][1]] = -1
        mvDepth[vDistIdx[i][1]] = -1
    return mvuRight, mvDepth


def compute_stereo_from_rgbd(mvKeys, mvKeysUn_x, imDepth: np.ndarray, mbf):
    """Frame::ComputeStereoFromRGBD, src/Frame.cc:645-666 -> (mvuRight, mvDepth).  mvKeys: KeyPoint list (raw), mvKeysUn_x: the
    undistorted x of each keypoint; imDepth: CV_32F.  `imDepth.at<float>(v, u)` truncates the float coordinates."""
    N = len(mvKeys)
    mvuRight = np.full(N, -1.0, np.float32)
    mvDepth = np.full(N, -1.0, np.float32)
    mbf = F32(mbf)
    for i in range(N):
        kp = mvKeys[i]
        v, u = kp.y, kp.x
        d = F32(imDepth[_trunc(v), _trunc(u)])
        if d > 0:
            mvDepth[i] = d
            mvuRight[i] = F32(mvKeysUn_x[i]) - mbf / d
    return mvuRight, mvDepth
