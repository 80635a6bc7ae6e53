// orbfe_match_resolve.h -- the HOST half of the Tracking-thread matchers: the projection of every map point into one window
// query (what each reference loop does before it calls GetFeaturesInArea), Frame::isInFrustum, and the sequentially greedy
// accept rules of ORBmatcher::SearchByProjection (x3) and SearchForInitialization replayed over per-query candidate lists.
//
// Pure C++ (no HIP, no device types) on purpose: liborbfe.so's orbfe_match.hip includes it for the product path, and
// tests/asan/resolve_harness.cpp compiles the very same text for the CPU with -fsanitize=address,undefined (GPU ASan does not
// exist on this pool; the host logic is where the index arithmetic of the matchers lives).
//
// A candidate is one 64-bit key
//      dist << 36 | ix << 30 | iy << 24 | idx << 8 | octave
// so ascending key order == (Hamming distance, Frame::GetFeaturesInArea order: cell x, cell y, insertion = keypoint index):
// "the first minimum found by the reference's loop" is the smallest key, "the second best" the next one.  The device hands
// over, per query, the K smallest keys that survive the STATIC filters (keypoint blocked before the call; mvuRight gate), plus
// how many survived in all; the rules below add the DYNAMIC ones (a keypoint taken by an earlier query) and ask for the full
// list only when the K keys run out -- rare, and exact when it happens.
#pragma once

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/orbfe.h"

namespace orbfe_resolve
{

// One window query of the device kernel (window_candidates_kernel, orbfe_match.hip): what a map point asks of the frame grid
struct MatchQuery { // 32 bytes
    float u, v, r;
    int min_level, max_level;
    float ur, ur_rad; // right-image check (mvuRight), used when flags & 2
    int flags;        // bit0: valid query, bit1: apply the u_right check
};

struct Camera { float fx, fy, cx, cy, bf, mb; };
static inline Camera camera_of(const orbfe_params *P)
{
    Camera c = {P->fx, P->fy, P->cx, P->cy, P->bf, P->fx != 0.f ? P->bf / P->fx : 0.f}; // mb := mbf / fx
    return c;
}

// ---- projection arithmetic of the matchers: the reference's float expressions in its evaluation order (contract Q4: no FMA
// contraction -- this header must be compiled with -ffp-contract=off) ----
// OPENCV-4.5.5-SEMANTICS: cv::Mat R*x+t for 3x3 * 3x1 CV_32F (small-matrix gemm path)
static inline void rt_apply(const float *T, const float *x, float *out)
{
    for (int i = 0; i < 3; i++) {
        const float t = (T[4 * i] * x[0] + T[4 * i + 1] * x[1]) + T[4 * i + 2] * x[2];
        out[i] = t + T[4 * i + 3];
    }
}
static inline void camera_center(const float *T, float *ow) // -Rcw.t()*tcw
{
    for (int i = 0; i < 3; i++) ow[i] = ((-T[i]) * T[3] + (-T[4 + i]) * T[7]) + (-T[8 + i]) * T[11];
}
// deterministic log for MapPoint::PredictScale (contract Q4)
static inline float log_det(float xf)
{
    double x = (double)xf;
    int e;
    double m = frexp(x, &e);
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 27.0;
    for (int k = 25; k >= 3; k -= 2) p = p * z + 1.0 / (double)k;
    p = p * z + 1.0;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    return (float)((double)e * LN2_HI + (2.0 * s * p + (double)e * LN2_LO));
}
static inline int predict_scale(float max_distance, float current_dist, float log_sf, int n_levels) // src/MapPoint.cc:402-417
{
    const float ratio = max_distance / current_dist;
    int n_scale = (int)ceilf(log_det(ratio) / log_sf);
    if (n_scale < 0) n_scale = 0;
    else if (n_scale >= n_levels) n_scale = n_levels - 1;
    return n_scale;
}
static inline float norm3(const float *po) { return (float)sqrt((double)po[0] * po[0] + (double)po[1] * po[1] + (double)po[2] * po[2]); }

// ---- one window query per map point: the part of each matcher's loop body that precedes GetFeaturesInArea ----
static const MatchQuery NO_QUERY = {0, 0, 0, 0, -1, 0, 0, 0};

// ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono), src/ORBmatcher.cc:1335-1390.  Returns -1 on a bad octave.
static inline int build_queries_last(const Camera &C, const float *sf, int nlevels, float min_x, float max_x, float min_y, float max_y,
                                     const float *Tcw_cur, const float *Tcw_last, int n_last, const float *last_pos, const uint8_t *last_desc,
                                     const int32_t *last_valid, const int32_t *last_octave, float th, int mono, std::vector<MatchQuery> &q,
                                     std::vector<uint8_t> &qd)
{
    float twc[3], tlc[3];
    camera_center(Tcw_cur, twc);
    rt_apply(Tcw_last, twc, tlc);
    const bool forward = tlc[2] > C.mb && !mono, backward = -tlc[2] > C.mb && !mono;
    q.assign(n_last > 0 ? n_last : 0, NO_QUERY);
    qd.assign((size_t)32 * (n_last > 0 ? n_last : 1), 0);
    for (int i = 0; i < n_last; i++) {
        MatchQuery &Q = q[i];
        if (!last_valid[i]) continue;
        if (last_octave[i] < 0 || last_octave[i] >= nlevels) return -1;
        float xc[3];
        rt_apply(Tcw_cur, last_pos + 3 * i, xc);
        const float invzc = (float)(1.0 / (double)xc[2]);
        if (invzc < 0) continue;
        const float u = C.fx * xc[0] * invzc + C.cx;
        const float v = C.fy * xc[1] * invzc + C.cy;
        if (u < min_x || u > max_x) continue;
        if (v < min_y || v > max_y) continue;
        const int oct = last_octave[i];
        const float radius = th * sf[oct];
        Q.u = u; Q.v = v; Q.r = radius; Q.flags = 1 | 2;
        Q.ur = u - C.bf * invzc; Q.ur_rad = radius;
        if (forward) { Q.min_level = oct; Q.max_level = -1; }
        else if (backward) { Q.min_level = 0; Q.max_level = oct; }
        else { Q.min_level = oct - 1; Q.max_level = oct + 1; }
        memcpy(&qd[(size_t)32 * i], last_desc + (size_t)32 * i, 32);
    }
    return 0;
}

// ORBmatcher::SearchByProjection(F, vpMapPoints, th), src/ORBmatcher.cc:49-70.  Returns -1 on a bad predicted level.
static inline int build_queries_points(const float *sf, int nlevels, int n_pts, const orbfe_track_point *pts, const uint8_t *pt_desc, float th,
                                       std::vector<MatchQuery> &q, std::vector<uint8_t> &qd)
{
    const bool b_factor = th != 1.0;
    q.assign(n_pts > 0 ? n_pts : 0, NO_QUERY);
    qd.assign((size_t)32 * (n_pts > 0 ? n_pts : 1), 0);
    for (int i = 0; i < n_pts; i++) {
        MatchQuery &Q = q[i];
        if (!pts[i].in_view) continue;
        const int lvl = pts[i].level;
        if (lvl < 0 || lvl >= nlevels) return -1;
        float r = pts[i].view_cos > 0.998 ? 2.5f : 4.0f; // RadiusByViewingCos, :129-135
        if (b_factor) r *= th;
        Q.u = pts[i].proj_x; Q.v = pts[i].proj_y; Q.r = r * sf[lvl];
        Q.min_level = lvl - 1; Q.max_level = lvl; Q.flags = 1 | 2;
        Q.ur = pts[i].proj_xr; Q.ur_rad = r * sf[lvl];
        memcpy(&qd[(size_t)32 * i], pt_desc + (size_t)32 * i, 32);
    }
    return 0;
}

// ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist), src/ORBmatcher.cc:1484-1527
static inline void build_queries_kf(const Camera &C, const float *sf, int nlevels, float log_sf, float min_x, float max_x, float min_y, float max_y,
                                    const float *Tcw_cur, int n_kf, const float *kf_pos, const uint8_t *kf_desc, const int32_t *kf_valid,
                                    const float *kf_max_distance, const float *kf_min_distance, float th, std::vector<MatchQuery> &q,
                                    std::vector<uint8_t> &qd)
{
    float ow[3];
    camera_center(Tcw_cur, ow);
    q.assign(n_kf > 0 ? n_kf : 0, NO_QUERY);
    qd.assign((size_t)32 * (n_kf > 0 ? n_kf : 1), 0);
    for (int i = 0; i < n_kf; i++) {
        MatchQuery &Q = q[i];
        if (!kf_valid[i]) continue;
        float xc[3];
        rt_apply(Tcw_cur, kf_pos + 3 * i, xc);
        const float invzc = (float)(1.0 / (double)xc[2]);
        const float u = C.fx * xc[0] * invzc + C.cx;
        const float v = C.fy * xc[1] * invzc + C.cy;
        if (u < min_x || u > max_x) continue;
        if (v < min_y || v > max_y) continue;
        float po[3];
        for (int k = 0; k < 3; k++) po[k] = kf_pos[3 * i + k] - ow[k];
        const float dist3d = norm3(po);
        if (dist3d < 0.8f * kf_min_distance[i] || dist3d > 1.2f * kf_max_distance[i]) continue;
        const int lvl = predict_scale(kf_max_distance[i], dist3d, log_sf, nlevels);
        Q.u = u; Q.v = v; Q.r = th * sf[lvl]; Q.min_level = lvl - 1; Q.max_level = lvl + 1; Q.flags = 1;
        memcpy(&qd[(size_t)32 * i], kf_desc + (size_t)32 * i, 32);
    }
}

// ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:414-421
static inline void build_queries_initialization(int n1, const orbfe_keypoint *keys1, const uint8_t *desc1, const float *prev_matched, int window_size,
                                                std::vector<MatchQuery> &q, std::vector<uint8_t> &qd)
{
    q.assign(n1 > 0 ? n1 : 0, NO_QUERY);
    qd.assign((size_t)32 * (n1 > 0 ? n1 : 1), 0);
    for (int i = 0; i < n1; i++) {
        MatchQuery &Q = q[i];
        const int level1 = keys1[i].octave;
        if (level1 > 0) continue;
        Q.u = prev_matched[2 * i]; Q.v = prev_matched[2 * i + 1]; Q.r = (float)window_size;
        Q.min_level = level1; Q.max_level = level1; Q.flags = 1;
        memcpy(&qd[(size_t)32 * i], desc1 + (size_t)32 * i, 32);
    }
}

// Frame::isInFrustum, src/Frame.cc:256-315, for n map points
static inline void is_in_frustum(const Camera &C, int nlevels, float log_sf, const float *Tcw, float min_x, float max_x, float min_y, float max_y, int n,
                                 const float *pos, const float *normal, const float *max_distance, const float *min_distance,
                                 float viewing_cos_limit, orbfe_track_point *out)
{
    float ow[3];
    camera_center(Tcw, ow);
    for (int i = 0; i < n; i++) {
        orbfe_track_point &o = out[i];
        o.in_view = 0; o.proj_x = o.proj_y = o.proj_xr = 0.f; o.level = 0; o.view_cos = 0.f;
        float pc[3];
        rt_apply(Tcw, pos + 3 * i, pc);
        if (pc[2] < 0.0f) continue;
        const float invz = 1.0f / pc[2];
        const float u = C.fx * pc[0] * invz + C.cx;
        const float v = C.fy * pc[1] * invz + C.cy;
        if (u < min_x || u > max_x) continue;
        if (v < min_y || v > max_y) continue;
        float po[3];
        for (int k = 0; k < 3; k++) po[k] = pos[3 * i + k] - ow[k];
        const float dist = norm3(po);
        if (dist < 0.8f * min_distance[i] || dist > 1.2f * max_distance[i]) continue;
        const double dot = (double)po[0] * normal[3 * i] + (double)po[1] * normal[3 * i + 1] + (double)po[2] * normal[3 * i + 2];
        const float view_cos = (float)(dot / (double)dist);
        if (view_cos < viewing_cos_limit) continue;
        o.in_view = 1;
        o.proj_x = u; o.proj_xr = u - C.bf * invz; o.proj_y = v;
        o.level = predict_scale(max_distance[i], dist, log_sf, nlevels);
        o.view_cos = view_cos;
    }
}

enum { HISTO_LENGTH = 30, TH_LOW = 50, TH_HIGH = 100, TOPK = 4 };
typedef unsigned long long ckey_t;
static const ckey_t NO_KEY = ~0ull;

static inline int key_dist(ckey_t k) { return (int)(k >> 36); }
static inline int key_idx(ckey_t k) { return (int)((k >> 8) & 0xffffu); }
static inline int key_level(ckey_t k) { return (int)(k & 0xffu); }

// ORBmatcher::ComputeThreeMaxima, src/ORBmatcher.cc:1597-1638, on the bin sizes.  The reference keeps a running top three with
// strict comparisons, i.e. the three largest NON-EMPTY bins in the order (size descending, index ascending); written here as
// three selections of the largest remaining (size, -index) key instead of the reference's shifting if-chain (the oracle and
// oracle/literal_matchers.py keep that form, so the two formulations check each other).
static inline void three_maxima(const int32_t *sizes, int L, int *ind1, int *ind2, int *ind3)
{
    int ind[3] = {-1, -1, -1};
    int32_t val[3] = {0, 0, 0};
    for (int r = 0; r < 3; r++) {
        long long best = -1;
        for (int i = 0; i < L; i++) {
            if (sizes[i] <= 0 || i == ind[0] || i == ind[1]) continue;
            const long long key = ((long long)sizes[i] << 32) | (long long)(0x7fffffff - i);
            if (key > best) { best = key; ind[r] = i; val[r] = sizes[i]; }
        }
        if (best < 0) break;
    }
    if ((float)val[1] < 0.1f * (float)val[0]) { ind[1] = -1; ind[2] = -1; } // :1628-1637
    else if ((float)val[2] < 0.1f * (float)val[0]) ind[2] = -1;
    *ind1 = ind[0]; *ind2 = ind[1]; *ind3 = ind[2];
}

static inline int rot_bin(float a1, float a2) // Q8: 30 slots, bin = round(rot / 30): the reference's three float statements (:1438-1443), the same in any restatement
{
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

struct RotHist {
    std::vector<int> v[HISTO_LENGTH];
    void three(int &i1, int &i2, int &i3) const
    {
        int32_t sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int32_t)v[i].size();
        three_maxima(sizes, HISTO_LENGTH, &i1, &i2, &i3);
    }
};

// Where the replay gets a query's candidates from.  topk[q * TOPK ..]: ascending, NO_KEY padded; n_static[q]: how many keys
// passed the static filters (> TOPK: the top-K block is a prefix).  full(q, out): every key of the query, unfiltered and
// unsorted (the device's list); called at most for the few queries whose top-K block ran out.
struct CandidateSource {
    const ckey_t *topk = nullptr;
    const int32_t *n_static = nullptr;
    const uint8_t *blocked0 = nullptr; // static filter: keypoints blocked before the call (may be null)
    bool drop_gated = false;           // static filter: keys with dist >= 256 (failed the mvuRight gate)
    void *user = nullptr;
    int (*full)(void *user, int q, std::vector<ckey_t> &out) = nullptr; // returns 0 on success
    int error = 0;
    std::vector<ckey_t> scratch;

    // candidates of query q in ascending order after the static filters; `complete` = nothing beyond them exists
    const ckey_t *begin(int q, int &n, bool &complete)
    {
        const ckey_t *p = topk + (size_t)q * TOPK;
        n = 0;
        while (n < TOPK && p[n] != NO_KEY) n++;
        complete = n_static[q] <= TOPK;
        return p;
    }
    // the full, statically filtered, ascending list (fallback)
    const ckey_t *all(int q, int &n)
    {
        scratch.clear();
        if (!full || full(user, q, scratch) != 0) { error = 1; n = 0; return scratch.data(); }
        size_t w = 0;
        for (size_t i = 0; i < scratch.size(); i++) {
            const ckey_t k = scratch[i];
            if (drop_gated && key_dist(k) >= 256) continue;
            if (blocked0 && blocked0[key_idx(k)]) continue;
            scratch[w++] = k;
        }
        scratch.resize(w);
        std::sort(scratch.begin(), scratch.end());
        n = (int)scratch.size();
        return scratch.data();
    }
};

// first (and second) candidate of query q for which skip(key) is false, in ascending key order
template <class Skip>
static inline void first_two(CandidateSource &src, int q, Skip skip, ckey_t &best, ckey_t &second, bool want_second)
{
    best = second = NO_KEY;
    int n;
    bool complete;
    const ckey_t *c = src.begin(q, n, complete);
    for (int pass = 0; pass < 2; pass++) {
        for (int k = 0; k < n; k++) {
            if (skip(c[k])) continue;
            if (best == NO_KEY) { best = c[k]; if (!want_second) return; }
            else { second = c[k]; return; }
        }
        if (complete || pass == 1) return;
        best = second = NO_KEY; // the prefix ran out before both were found: replay on the full list
        c = src.all(q, n);
    }
}

// ---- ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono), src/ORBmatcher.cc:1324-1466: accept rules ----
// has_obs[k] in/out scratch (initially: keypoint k holds a point with Observations() > 0); cur_angle = mvKeysUn[k].angle
static inline int resolve_last(CandidateSource &src, int n_last, const int32_t *last_obs, const float *last_angle, int n_cur,
                               const float *cur_angle, size_t cur_angle_stride, std::vector<uint8_t> &has_obs, int check_ori, int32_t *cur_match)
{
    for (int k = 0; k < n_cur; k++) cur_match[k] = -1;
    RotHist rh;
    int nm = 0;
    for (int i = 0; i < n_last; i++) {
        ckey_t best, second;
        first_two(src, i, [&](ckey_t k) { return has_obs[key_idx(k)] != 0; }, best, second, false);
        if (best != NO_KEY && key_dist(best) <= TH_HIGH) {
            const int bi = key_idx(best);
            cur_match[bi] = i;
            has_obs[bi] = last_obs[i] > 0;
            nm++;
            if (check_ori) rh.v[rot_bin(last_angle[i], *(const float *)((const char *)cur_angle + cur_angle_stride * bi))].push_back(bi);
        }
    }
    if (check_ori) {
        int i1, i2, i3;
        rh.three(i1, i2, i3);
        for (int b = 0; b < HISTO_LENGTH; b++)
            if (b != i1 && b != i2 && b != i3)
                for (int idx : rh.v[b]) { cur_match[idx] = -1; nm--; }
    }
    return nm;
}

// ---- ORBmatcher::SearchByProjection(F, vpMapPoints, th), src/ORBmatcher.cc:43-127 ----
static inline int resolve_points(CandidateSource &src, int n_pts, const int32_t *pt_obs, int n_cur, std::vector<uint8_t> &has_obs, float nnratio,
                                 int32_t *cur_match)
{
    for (int k = 0; k < n_cur; k++) cur_match[k] = -1;
    int nm = 0;
    for (int i = 0; i < n_pts; i++) {
        ckey_t best, second; // two smallest (dist, order) keys == best / second of the reference's loop
        first_two(src, i, [&](ckey_t k) { return has_obs[key_idx(k)] != 0; }, best, second, true);
        if (best == NO_KEY) continue;
        const int best_dist = key_dist(best), best_level = key_level(best);
        const int best_dist2 = second != NO_KEY ? key_dist(second) : 256, best_level2 = second != NO_KEY ? key_level(second) : -1;
        if (best_dist <= TH_HIGH) {
            if (best_level == best_level2 && (float)best_dist > nnratio * (float)best_dist2) continue;
            cur_match[key_idx(best)] = i;
            has_obs[key_idx(best)] = pt_obs[i] > 0;
            nm++;
        }
    }
    return nm;
}

// ---- ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist), src/ORBmatcher.cc:1468-1595 ----
static inline int resolve_kf(CandidateSource &src, int n_kf, const float *kf_angle, int n_cur, const float *cur_angle, size_t cur_angle_stride,
                             std::vector<uint8_t> &has_pt, int orb_dist, int check_ori, int32_t *cur_match)
{
    for (int k = 0; k < n_cur; k++) cur_match[k] = -1;
    RotHist rh;
    int nm = 0;
    for (int i = 0; i < n_kf; i++) {
        ckey_t best, second;
        first_two(src, i, [&](ckey_t k) { return has_pt[key_idx(k)] != 0; }, best, second, false);
        if (best != NO_KEY && key_dist(best) <= orb_dist) {
            const int bi = key_idx(best);
            cur_match[bi] = i;
            has_pt[bi] = 1;
            nm++;
            if (check_ori) rh.v[rot_bin(kf_angle[i], *(const float *)((const char *)cur_angle + cur_angle_stride * bi))].push_back(bi);
        }
    }
    if (check_ori) {
        int i1, i2, i3;
        rh.three(i1, i2, i3);
        for (int b = 0; b < HISTO_LENGTH; b++)
            if (b != i1 && b != i2 && b != i3)
                for (int idx : rh.v[b]) { cur_match[idx] = -1; nm--; }
    }
    return nm;
}

// ---- ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:400-515 ----
// angle1 / angle2 = mvKeysUn[].angle of the two frames (strided), xy2 = mvKeysUn[].pt of frame 2 (x at xy2, y 4 bytes later)
static inline int resolve_initialization(CandidateSource &src, int n1, int n2, const float *angle1, size_t stride1, const float *angle2, size_t stride2,
                                         const float *xy2, float nnratio, int check_ori, float *prev_matched, int32_t *matches12)
{
    std::vector<int> matched_dist(n2 > 0 ? n2 : 1, INT_MAX), matches21(n2 > 0 ? n2 : 1, -1);
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    RotHist rh;
    int nm = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        ckey_t best, second;
        first_two(src, i1, [&](ckey_t k) { return matched_dist[key_idx(k)] <= key_dist(k); }, best, second, true); // :437-438
        if (best == NO_KEY) continue;
        const int best_dist = key_dist(best), best_dist2 = second != NO_KEY ? key_dist(second) : INT_MAX, bi = key_idx(best);
        if (best_dist <= TH_LOW && (float)best_dist < (float)best_dist2 * nnratio) {
            if (matches21[bi] >= 0) { matches12[matches21[bi]] = -1; nm--; }
            matches12[i1] = bi;
            matches21[bi] = i1;
            matched_dist[bi] = best_dist;
            nm++;
            if (check_ori)
                rh.v[rot_bin(*(const float *)((const char *)angle1 + stride1 * i1), *(const float *)((const char *)angle2 + stride2 * bi))].push_back(i1);
        }
    }
    if (check_ori) {
        int i1, i2, i3;
        rh.three(i1, i2, i3);
        for (int b = 0; b < HISTO_LENGTH; b++) {
            if (b == i1 || b == i2 || b == i3) continue;
            for (int idx1 : rh.v[b])
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; nm--; }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (matches12[i1] >= 0) {
            const float *p = (const float *)((const char *)xy2 + stride2 * matches12[i1]);
            prev_matched[2 * i1] = p[0];
            prev_matched[2 * i1 + 1] = p[1];
        }
    return nm;
}

} // namespace orbfe_resolve
