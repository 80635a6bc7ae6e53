/*
 * orb_oracle_match.c -- CPU ORACLE (test infrastructure, NOT product code): the Tracking-thread
 * matchers of the reference restated on flattened inputs.
 * PARITY UNPINNED (see orb_oracle.h).  Pointer-rich reference state is passed as arrays:
 *   MapPoint*            -> index into the caller's point arrays (or -1)
 *   pMP->Observations()  -> obs[] ; pMP->GetDescriptor() -> desc[][32] ; GetWorldPos() -> pos[][3]
 * cv::Mat algebra follows OpenCV 4.5.5's small-matrix gemm path (float products summed left to
 * right in float, then + c), tagged OPENCV-4.5.5-SEMANTICS; no FMA contraction (contract Q4).
 */
#include "orb_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define GRID_COLS 64 /* FRAME_GRID_COLS, include/Frame.h:36 */
#define GRID_ROWS 48 /* FRAME_GRID_ROWS, include/Frame.h:37 */
#define TH_LOW 50
#define TH_HIGH 100
#define HISTO_LENGTH 30

struct orc_grid {
    int n;
    const orc_keypoint *keys_un;
    float min_x, max_x, min_y, max_y, inv_w, inv_h;
    int *cell_cnt;  /* [GRID_COLS*GRID_ROWS], index ix*GRID_ROWS+iy */
    int *cell_off;
    int *cell_idx;
};

/* KeyFrame::KeyFrame (src/KeyFrame.cc:32-50): the keyframe takes the frame's grid and cell size as they are and keeps the image
 * bounds as INTS initialised from the frame's floats (include/KeyFrame.h:194-197); KeyFrame::GetFeaturesInArea / IsInImage
 * (src/KeyFrame.cc:563-607) then compute with those ints.  Call after orc_grid_create with the frame's float bounds. */
void orc_grid_as_keyframe(orc_grid *g)
{
    g->min_x = (float)(int)g->min_x; g->max_x = (float)(int)g->max_x;
    g->min_y = (float)(int)g->min_y; g->max_y = (float)(int)g->max_y;
}

/* Frame::AssignFeaturesToGrid + PosInGrid, reference src/Frame.cc:231-246,383-393 (Q6: round()) */
orc_grid *orc_grid_create(const orc_keypoint *keys_un, int n, float min_x, float max_x, float min_y, float max_y)
{
    orc_grid *g = (orc_grid *)calloc(1, sizeof(*g));
    g->n = n; g->keys_un = keys_un;
    g->min_x = min_x; g->max_x = max_x; g->min_y = min_y; g->max_y = max_y;
    g->inv_w = (float)GRID_COLS / (max_x - min_x);
    g->inv_h = (float)GRID_ROWS / (max_y - min_y);
    g->cell_cnt = (int *)calloc(GRID_COLS * GRID_ROWS, sizeof(int));
    g->cell_off = (int *)calloc(GRID_COLS * GRID_ROWS + 1, sizeof(int));
    g->cell_idx = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int *cell_of = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        const int px = (int)roundf((keys_un[i].x - min_x) * g->inv_w);
        const int py = (int)roundf((keys_un[i].y - min_y) * g->inv_h);
        cell_of[i] = -1;
        if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) continue;
        cell_of[i] = px * GRID_ROWS + py;
        g->cell_cnt[cell_of[i]]++;
    }
    for (int c = 0; c < GRID_COLS * GRID_ROWS; c++) g->cell_off[c + 1] = g->cell_off[c] + g->cell_cnt[c];
    int *cur = (int *)malloc(sizeof(int) * GRID_COLS * GRID_ROWS);
    memcpy(cur, g->cell_off, sizeof(int) * GRID_COLS * GRID_ROWS);
    for (int i = 0; i < n; i++) if (cell_of[i] >= 0) g->cell_idx[cur[cell_of[i]]++] = i; /* push_back order = ascending i */
    free(cur); free(cell_of);
    return g;
}

void orc_grid_destroy(orc_grid *g)
{
    if (!g) return;
    free(g->cell_cnt); free(g->cell_off); free(g->cell_idx); free(g);
}

/* Frame::GetFeaturesInArea, reference src/Frame.cc:328-381 (Q5: level check rule reproduced literally) */
int orc_features_in_area(const orc_grid *g, float x, float y, float r, int min_level, int max_level, int32_t *out, int cap)
{
    int n = 0;
    int v;
    v = (int)floorf((x - g->min_x - r) * g->inv_w);
    const int min_cx = v > 0 ? v : 0;
    if (min_cx >= GRID_COLS) return 0;
    v = (int)ceilf((x - g->min_x + r) * g->inv_w);
    const int max_cx = v < GRID_COLS - 1 ? v : GRID_COLS - 1;
    if (max_cx < 0) return 0;
    v = (int)floorf((y - g->min_y - r) * g->inv_h);
    const int min_cy = v > 0 ? v : 0;
    if (min_cy >= GRID_ROWS) return 0;
    v = (int)ceilf((y - g->min_y + r) * g->inv_h);
    const int max_cy = v < GRID_ROWS - 1 ? v : GRID_ROWS - 1;
    if (max_cy < 0) return 0;
    const int check_levels = (min_level > 0) || (max_level >= 0);
    for (int ix = min_cx; ix <= max_cx; ix++)
        for (int iy = min_cy; iy <= max_cy; iy++) {
            const int c = ix * GRID_ROWS + iy;
            for (int j = g->cell_off[c]; j < g->cell_off[c + 1]; j++) {
                const int idx = g->cell_idx[j];
                const orc_keypoint *kp = &g->keys_un[idx];
                if (check_levels) {
                    if (kp->octave < min_level) continue;
                    if (max_level >= 0 && kp->octave > max_level) continue;
                }
                const float dx = kp->x - x, dy = kp->y - y;
                if (fabsf(dx) < r && fabsf(dy) < r) {
                    if (n < cap) out[n] = idx;
                    n++;
                }
            }
        }
    return n;
}

/* ORBmatcher::ComputeThreeMaxima, reference src/ORBmatcher.cc:1597-1638 */
void orc_three_maxima(const int32_t *histo_sizes, int L, int *ind1, int *ind2, int *ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    *ind1 = *ind2 = *ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = histo_sizes[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > max3) { max3 = s; *ind3 = i; }
    }
    if ((float)max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

typedef struct rot_hist { int *v[HISTO_LENGTH]; int n[HISTO_LENGTH]; int cap[HISTO_LENGTH]; } rot_hist;
static void rh_push(rot_hist *h, int bin, int val)
{
    if (h->n[bin] == h->cap[bin]) { h->cap[bin] = h->cap[bin] ? 2 * h->cap[bin] : 64; h->v[bin] = (int *)realloc(h->v[bin], sizeof(int) * (size_t)h->cap[bin]); }
    h->v[bin][h->n[bin]++] = val;
}
static void rh_free(rot_hist *h) { for (int i = 0; i < HISTO_LENGTH; i++) free(h->v[i]); }
static int rot_bin(float a1, float a2)
{
    const float factor = 1.0f / HISTO_LENGTH; /* Q8: 30 slots but bin = round(rot/30) */
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

/* OPENCV-4.5.5-SEMANTICS: cv::Mat Rcw*x+tcw for 3x3 * 3x1 CV_32F (small-matrix gemm path) */
static void rt_apply(const float *T /*3x4 row major*/, const float *x, float *out)
{
    for (int i = 0; i < 3; i++) {
        const float t = (T[4 * i] * x[0] + T[4 * i + 1] * x[1]) + T[4 * i + 2] * x[2];
        out[i] = t + T[4 * i + 3];
    }
}
/* twc = -Rcw.t()*tcw */
static void camera_center(const float *T, float *ow)
{
    for (int i = 0; i < 3; i++) {
        const float t = ((-T[i]) * T[3] + (-T[4 + i]) * T[7]) + (-T[8 + i]) * T[11];
        ow[i] = t;
    }
}

/* deterministic log (contract Q4 applied to PredictScale's log(): double evaluation, one rounding).
 * x > 0.  log(x) = e*ln2 + 2*atanh((m-1)/(m+1)), m in [sqrt(1/2), sqrt(2)). */
float orc_log_det(float xf)
{
    double x = (double)xf;
    int e;
    double m = frexp(x, &e); /* m in [0.5, 1) */
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 27.0;
    for (int k = 25; k >= 3; k -= 2) p = p * z + 1.0 / (double)k;
    p = p * z + 1.0;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double r = (double)e * LN2_HI + (2.0 * s * p + (double)e * LN2_LO);
    return (float)r;
}

/* MapPoint::PredictScale, reference src/MapPoint.cc:402-417 (log via the contract routine) */
int orc_predict_scale(float max_distance, float current_dist, float log_scale_factor, int n_levels)
{
    const float ratio = max_distance / current_dist;
    int n_scale = (int)ceilf(orc_log_det(ratio) / log_scale_factor);
    if (n_scale < 0) n_scale = 0;
    else if (n_scale >= n_levels) n_scale = n_levels - 1;
    return n_scale;
}

/* ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono), reference src/ORBmatcher.cc:1324-1466 */
int orc_search_by_projection_last(const orc_grid *g, const float *u_right_cur, const uint8_t *desc_cur,
                                  const float *scale_factors, const orc_camera *cam,
                                  const float *Tcw_cur, const float *Tcw_last,
                                  int n_last, const float *last_pos, const uint8_t *last_desc,
                                  const int32_t *last_valid, const int32_t *last_obs, const int32_t *last_octave,
                                  const float *last_angle, const uint8_t *cur_has_obs_in,
                                  float th, int mono, int check_ori, int32_t *cur_match)
{
    const int N = g->n;
    int nmatches = 0;
    rot_hist rh; memset(&rh, 0, sizeof(rh));
    uint8_t *has_obs = (uint8_t *)malloc((size_t)(N > 0 ? N : 1));
    for (int i = 0; i < N; i++) { has_obs[i] = cur_has_obs_in ? cur_has_obs_in[i] : 0; cur_match[i] = -1; }
    float twc[3], tlc[3];
    camera_center(Tcw_cur, twc);
    rt_apply(Tcw_last, twc, tlc);
    const int forward = tlc[2] > cam->mb && !mono;
    const int backward = -tlc[2] > cam->mb && !mono;
    int32_t *ind = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    for (int i = 0; i < n_last; i++) {
        if (!last_valid[i]) continue; /* pMP && !mvbOutlier[i] */
        float xc[3];
        rt_apply(Tcw_cur, last_pos + 3 * i, xc);
        const float invzc = (float)(1.0 / (double)xc[2]);
        if (invzc < 0) continue;
        const float u = cam->fx * xc[0] * invzc + cam->cx;
        const float v = cam->fy * xc[1] * invzc + cam->cy;
        if (u < g->min_x || u > g->max_x) continue;
        if (v < g->min_y || v > g->max_y) continue;
        const int oct = last_octave[i];
        const float radius = th * scale_factors[oct];
        int nc;
        if (forward) nc = orc_features_in_area(g, u, v, radius, oct, -1, ind, N);
        else if (backward) nc = orc_features_in_area(g, u, v, radius, 0, oct, ind, N);
        else nc = orc_features_in_area(g, u, v, radius, oct - 1, oct + 1, ind, N);
        if (nc == 0) continue;
        int best_dist = 256, best_idx = -1;
        for (int k = 0; k < nc; k++) {
            const int i2 = ind[k];
            if (has_obs[i2]) continue; /* mvpMapPoints[i2] && Observations()>0 */
            if (u_right_cur && u_right_cur[i2] > 0) {
                const float ur = u - cam->bf * invzc;
                const float er = fabsf(ur - u_right_cur[i2]);
                if (er > radius) continue;
            }
            const int dist = orc_hamming256(last_desc + 32 * (size_t)i, desc_cur + 32 * (size_t)i2);
            if (dist < best_dist) { best_dist = dist; best_idx = i2; }
        }
        if (best_dist <= TH_HIGH) {
            cur_match[best_idx] = i;
            has_obs[best_idx] = last_obs[i] > 0;
            nmatches++;
            if (check_ori) rh_push(&rh, rot_bin(last_angle[i], g->keys_un[best_idx].angle), best_idx);
        }
    }
    if (check_ori) {
        int i1, i2, i3;
        orc_three_maxima(rh.n, HISTO_LENGTH, &i1, &i2, &i3);
        for (int b = 0; b < HISTO_LENGTH; b++)
            if (b != i1 && b != i2 && b != i3)
                for (int j = 0; j < rh.n[b]; j++) { cur_match[rh.v[b][j]] = -1; nmatches--; }
    }
    rh_free(&rh); free(has_obs); free(ind);
    return nmatches;
}

/* Frame::isInFrustum, reference src/Frame.cc:270-326 */
int orc_is_in_frustum(const float *Tcw, const orc_camera *cam, float min_x, float max_x, float min_y, float max_y,
                      const float *pos, const float *normal, float max_dist_inv, float min_dist_inv, float max_distance,
                      float viewing_cos_limit, float log_scale_factor, int n_levels, orc_track_point *out)
{
    out->in_view = 0;
    float pc[3];
    rt_apply(Tcw, pos, pc);
    if (pc[2] < 0.0f) return 0;
    const float invz = 1.0f / pc[2];
    const float u = cam->fx * pc[0] * invz + cam->cx;
    const float v = cam->fy * pc[1] * invz + cam->cy;
    if (u < min_x || u > max_x) return 0;
    if (v < min_y || v > max_y) return 0;
    float ow[3], po[3];
    camera_center(Tcw, ow);
    for (int i = 0; i < 3; i++) po[i] = pos[i] - ow[i];
    const float dist = (float)sqrt((double)po[0] * po[0] + (double)po[1] * po[1] + (double)po[2] * po[2]); /* cv::norm: double acc */
    if (dist < min_dist_inv || dist > max_dist_inv) return 0;
    const double dot = (double)po[0] * normal[0] + (double)po[1] * normal[1] + (double)po[2] * normal[2];   /* Mat::dot: double */
    const float view_cos = (float)(dot / (double)dist);
    if (view_cos < viewing_cos_limit) return 0;
    out->in_view = 1;
    out->proj_x = u;
    out->proj_xr = u - cam->bf * invz;
    out->proj_y = v;
    out->level = orc_predict_scale(max_distance, dist, log_scale_factor, n_levels);
    out->view_cos = view_cos;
    return 1;
}

/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th), reference src/ORBmatcher.cc:43-135 */
int orc_search_by_projection_points(const orc_grid *g, const float *u_right_cur, const uint8_t *desc_cur,
                                    const float *scale_factors, int n_pts, const orc_track_point *pts,
                                    const uint8_t *pt_desc, const int32_t *pt_obs, const uint8_t *cur_has_obs_in,
                                    float th, float nnratio, int32_t *cur_match)
{
    const int N = g->n;
    int nmatches = 0;
    const int b_factor = th != 1.0;
    uint8_t *has_obs = (uint8_t *)malloc((size_t)(N > 0 ? N : 1));
    for (int i = 0; i < N; i++) { has_obs[i] = cur_has_obs_in ? cur_has_obs_in[i] : 0; cur_match[i] = -1; }
    int32_t *ind = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    for (int ip = 0; ip < n_pts; ip++) {
        if (!pts[ip].in_view) continue; /* mbTrackInView && !isBad() */
        const int lvl = pts[ip].level;
        float r = pts[ip].view_cos > 0.998 ? 2.5f : 4.0f; /* RadiusByViewingCos (double literal compare) */
        if (b_factor) r *= th;
        const int nc = orc_features_in_area(g, pts[ip].proj_x, pts[ip].proj_y, r * scale_factors[lvl], lvl - 1, lvl, ind, N);
        if (nc == 0) continue;
        int best_dist = 256, best_level = -1, best_dist2 = 256, best_level2 = -1, best_idx = -1;
        for (int k = 0; k < nc; k++) {
            const int idx = ind[k];
            if (has_obs[idx]) continue;
            if (u_right_cur && u_right_cur[idx] > 0) {
                const float er = fabsf(pts[ip].proj_xr - u_right_cur[idx]);
                if (er > r * scale_factors[lvl]) continue;
            }
            const int dist = orc_hamming256(pt_desc + 32 * (size_t)ip, desc_cur + 32 * (size_t)idx);
            if (dist < best_dist) {
                best_dist2 = best_dist; best_dist = dist;
                best_level2 = best_level; best_level = g->keys_un[idx].octave;
                best_idx = idx;
            } else if (dist < best_dist2) {
                best_level2 = g->keys_un[idx].octave;
                best_dist2 = dist;
            }
        }
        if (best_dist <= TH_HIGH) {
            if (best_level == best_level2 && (float)best_dist > nnratio * (float)best_dist2) continue;
            cur_match[best_idx] = ip;
            has_obs[best_idx] = pt_obs[ip] > 0;
            nmatches++;
        }
    }
    free(has_obs); free(ind);
    return nmatches;
}

/* ORBmatcher::SearchByProjection(Frame&, KeyFrame*, set, th, ORBdist), reference src/ORBmatcher.cc:1468-1595 */
int orc_search_by_projection_kf(const orc_grid *g, const uint8_t *desc_cur, const float *scale_factors,
                                const orc_camera *cam, const float *Tcw_cur, float log_scale_factor, int n_levels,
                                int n_kf, const float *kf_pos, const uint8_t *kf_desc, const int32_t *kf_valid,
                                const float *kf_angle, const float *kf_max_distance, const float *kf_min_distance,
                                const uint8_t *cur_has_point_in, float th, int orb_dist, int check_ori, int32_t *cur_match)
{
    const int N = g->n;
    int nmatches = 0;
    rot_hist rh; memset(&rh, 0, sizeof(rh));
    uint8_t *has_pt = (uint8_t *)malloc((size_t)(N > 0 ? N : 1));
    for (int i = 0; i < N; i++) { has_pt[i] = cur_has_point_in ? cur_has_point_in[i] : 0; cur_match[i] = -1; }
    int32_t *ind = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    float ow[3];
    camera_center(Tcw_cur, ow);
    for (int i = 0; i < n_kf; i++) {
        if (!kf_valid[i]) continue; /* pMP && !isBad() && !sAlreadyFound.count(pMP) */
        float xc[3];
        rt_apply(Tcw_cur, kf_pos + 3 * i, xc);
        const float invzc = (float)(1.0 / (double)xc[2]);
        const float u = cam->fx * xc[0] * invzc + cam->cx;
        const float v = cam->fy * xc[1] * invzc + cam->cy;
        if (u < g->min_x || u > g->max_x) continue;
        if (v < g->min_y || v > g->max_y) continue;
        float po[3];
        for (int k = 0; k < 3; k++) po[k] = kf_pos[3 * i + k] - ow[k];
        const float dist3d = (float)sqrt((double)po[0] * po[0] + (double)po[1] * po[1] + (double)po[2] * po[2]);
        const float max_distance = 1.2f * kf_max_distance[i]; /* GetMaxDistanceInvariance, src/MapPoint.cc:379-383 */
        const float min_distance = 0.8f * kf_min_distance[i]; /* GetMinDistanceInvariance, src/MapPoint.cc:373-377 */
        if (dist3d < min_distance || dist3d > max_distance) continue;
        const int lvl = orc_predict_scale(kf_max_distance[i], dist3d, log_scale_factor, n_levels);
        const float radius = th * scale_factors[lvl];
        const int nc = orc_features_in_area(g, u, v, radius, lvl - 1, lvl + 1, ind, N);
        if (nc == 0) continue;
        int best_dist = 256, best_idx = -1;
        for (int k = 0; k < nc; k++) {
            const int i2 = ind[k];
            if (has_pt[i2]) continue;
            const int dist = orc_hamming256(kf_desc + 32 * (size_t)i, desc_cur + 32 * (size_t)i2);
            if (dist < best_dist) { best_dist = dist; best_idx = i2; }
        }
        if (best_dist <= orb_dist) {
            cur_match[best_idx] = i;
            has_pt[best_idx] = 1;
            nmatches++;
            if (check_ori) rh_push(&rh, rot_bin(kf_angle[i], g->keys_un[best_idx].angle), best_idx);
        }
    }
    if (check_ori) {
        int i1, i2, i3;
        orc_three_maxima(rh.n, HISTO_LENGTH, &i1, &i2, &i3);
        for (int b = 0; b < HISTO_LENGTH; b++)
            if (b != i1 && b != i2 && b != i3)
                for (int j = 0; j < rh.n[b]; j++) { cur_match[rh.v[b][j]] = -1; nmatches--; }
    }
    rh_free(&rh); free(has_pt); free(ind);
    return nmatches;
}

/* ORBmatcher::SearchForInitialization, reference src/ORBmatcher.cc:400-515 */
int orc_search_for_initialization(const orc_keypoint *keys1, const uint8_t *desc1, int n1,
                                  const orc_grid *g2, const uint8_t *desc2,
                                  float *prev_matched /* [n1][2] in/out */, int window_size, float nnratio, int check_ori,
                                  int32_t *matches12)
{
    const int n2 = g2->n;
    int nmatches = 0;
    rot_hist rh; memset(&rh, 0, sizeof(rh));
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    int *matched_dist = (int *)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
    int *matches21 = (int *)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
    for (int i = 0; i < n2; i++) { matched_dist[i] = INT_MAX; matches21[i] = -1; }
    int32_t *ind = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 > 0 ? n2 : 1));
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = keys1[i1].octave;
        if (level1 > 0) continue;
        const int nc = orc_features_in_area(g2, prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)window_size, level1, level1, ind, n2);
        if (nc == 0) continue;
        int best_dist = INT_MAX, best_dist2 = INT_MAX, best_idx2 = -1;
        for (int k = 0; k < nc; k++) {
            const int i2 = ind[k];
            const int dist = orc_hamming256(desc1 + 32 * (size_t)i1, desc2 + 32 * (size_t)i2);
            if (matched_dist[i2] <= dist) continue;
            if (dist < best_dist) { best_dist2 = best_dist; best_dist = dist; best_idx2 = i2; }
            else if (dist < best_dist2) best_dist2 = dist;
        }
        if (best_dist <= TH_LOW) {
            if ((float)best_dist < (float)best_dist2 * nnratio) {
                if (matches21[best_idx2] >= 0) { matches12[matches21[best_idx2]] = -1; nmatches--; }
                matches12[i1] = best_idx2;
                matches21[best_idx2] = i1;
                matched_dist[best_idx2] = best_dist;
                nmatches++;
                if (check_ori) rh_push(&rh, rot_bin(keys1[i1].angle, g2->keys_un[best_idx2].angle), i1);
            }
        }
    }
    if (check_ori) {
        int i1, i2, i3;
        orc_three_maxima(rh.n, HISTO_LENGTH, &i1, &i2, &i3);
        for (int b = 0; b < HISTO_LENGTH; b++) {
            if (b == i1 || b == i2 || b == i3) continue;
            for (int j = 0; j < rh.n[b]; j++) {
                const int idx1 = rh.v[b][j];
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
            }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (matches12[i1] >= 0) {
            prev_matched[2 * i1] = g2->keys_un[matches12[i1]].x;
            prev_matched[2 * i1 + 1] = g2->keys_un[matches12[i1]].y;
        }
    rh_free(&rh); free(matched_dist); free(matches21); free(ind);
    return nmatches;
}

/* Search part of ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:821-971): for every
 * candidate map point the keyframe keypoint it would be fused with (best_idx, -1 if none).  The map mutation that
 * follows (Replace / AddObservation, :943-964) stays with the caller; it never feeds back into the search.
 * pt_valid = pMP && !isBad() && !IsInKeyFrame(pKF).  max_distance / min_distance are mfMaxDistance / mfMinDistance
 * (the 1.2 / 0.8 invariance factors are applied here, src/MapPoint.cc:373-383). */
int orc_fuse(const orc_grid *g, const float *u_right_kf, const uint8_t *desc_kf, const float *scale_factors, const float *inv_level_sigma2,
             const orc_camera *cam, const float *Tcw, float log_scale_factor, int n_levels,
             int n_pts, const float *pos, const float *normal, const float *max_distance, const float *min_distance,
             const uint8_t *pt_desc, const int32_t *pt_valid, float th, int32_t *best_idx_out)
{
    const int N = g->n;
    int n_fused = 0;
    int32_t *ind = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    float ow[3];
    camera_center(Tcw, ow);
    for (int i = 0; i < n_pts; i++) {
        best_idx_out[i] = -1;
        if (!pt_valid[i]) continue;
        float pc[3];
        rt_apply(Tcw, pos + 3 * i, pc);
        if (pc[2] < 0.0f) continue;
        const float invz = 1 / pc[2];
        const float x = pc[0] * invz, y = pc[1] * invz;
        const float u = cam->fx * x + cam->cx;
        const float v = cam->fy * y + cam->cy;
        if (!(u >= g->min_x && u < g->max_x && v >= g->min_y && v < g->max_y)) continue; /* KeyFrame::IsInImage */
        const float ur = u - cam->bf * invz;
        const float max_d = 1.2f * max_distance[i], min_d = 0.8f * min_distance[i];
        float po[3];
        for (int k = 0; k < 3; k++) po[k] = pos[3 * i + k] - ow[k];
        const float dist3d = (float)sqrt((double)po[0] * po[0] + (double)po[1] * po[1] + (double)po[2] * po[2]);
        if (dist3d < min_d || dist3d > max_d) continue;
        const double dot = (double)po[0] * normal[3 * i] + (double)po[1] * normal[3 * i + 1] + (double)po[2] * normal[3 * i + 2];
        if (dot < 0.5 * (double)dist3d) continue; /* viewing angle below 60 degrees */
        const int lvl = orc_predict_scale(max_distance[i], dist3d, log_scale_factor, n_levels);
        const float radius = th * scale_factors[lvl];
        const int nc = orc_features_in_area(g, u, v, radius, -1, -1, ind, N); /* KeyFrame::GetFeaturesInArea: no level filter */
        if (nc == 0) continue;
        int best_dist = 256, best_idx = -1;
        for (int k = 0; k < nc; k++) {
            const int idx = ind[k];
            const orc_keypoint *kp = &g->keys_un[idx];
            const int kp_level = kp->octave;
            if (kp_level < lvl - 1 || kp_level > lvl) continue;
            if (u_right_kf && u_right_kf[idx] >= 0) { /* reprojection error in stereo */
                const float ex = u - kp->x, ey = v - kp->y, er = ur - u_right_kf[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if ((double)(e2 * inv_level_sigma2[kp_level]) > 7.8) continue;
            } else {
                const float ex = u - kp->x, ey = v - kp->y;
                const float e2 = ex * ex + ey * ey;
                if ((double)(e2 * inv_level_sigma2[kp_level]) > 5.99) continue;
            }
            const int dist = orc_hamming256(pt_desc + 32 * (size_t)i, desc_kf + 32 * (size_t)idx);
            if (dist < best_dist) { best_dist = dist; best_idx = idx; }
        }
        if (best_dist <= TH_LOW) { best_idx_out[i] = best_idx; n_fused++; }
    }
    free(ind);
    return n_fused;
}

/* Sim3 decomposition shared by the LoopClosing matchers (src/ORBmatcher.cc:293-298, 981-986): scw = sqrt(row0 . row0)
 * (Mat::dot accumulates in double), Rcw = sRcw / scw and tcw = t / scw (OPENCV-4.5.5-SEMANTICS: Mat / scalar is a
 * convertTo with alpha = 1 / scw, applied as a float multiply), packed as a 3x4 [R|t]. */
static void sim3_to_rt(const float *Scw /*3x4 row major: [sR|t]*/, float *T)
{
    const double d = (double)Scw[0] * Scw[0] + (double)Scw[1] * Scw[1] + (double)Scw[2] * Scw[2];
    const float scw = (float)sqrt(d);
    const float alpha = (float)(1.0 / (double)scw);
    for (int i = 0; i < 12; i++) T[i] = Scw[i] * alpha;
}

/* mode 0: ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (src/ORBmatcher.cc:285-398): greedy over
 *         the points (a keypoint that already has a match -- kf_matched on entry or taken earlier in the loop -- is skipped);
 * mode 1: search part of ORBmatcher::Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint) (:973-1096): points independent.
 * pt_valid = !isBad() && !already found in the keyframe.  pt_match[i] = keypoint or -1. */
int orc_search_by_sim3_projection(int mode, const orc_grid *g, const uint8_t *desc_kf, const float *scale_factors, const orc_camera *cam,
                                  const float *Scw, float log_scale_factor, int n_levels,
                                  int n_pts, const float *pos, const float *normal, const float *max_distance, const float *min_distance,
                                  const uint8_t *pt_desc, const int32_t *pt_valid, const uint8_t *kf_matched_in, float th, int32_t *pt_match)
{
    const int N = g->n;
    int nmatches = 0;
    float T[12], ow[3];
    sim3_to_rt(Scw, T);
    camera_center(T, ow);
    uint8_t *matched = (uint8_t *)malloc((size_t)(N > 0 ? N : 1));
    for (int i = 0; i < N; i++) matched[i] = (mode == 0 && kf_matched_in) ? kf_matched_in[i] : 0;
    int32_t *ind = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    for (int i = 0; i < n_pts; i++) {
        pt_match[i] = -1;
        if (!pt_valid[i]) continue;
        float pc[3];
        rt_apply(T, pos + 3 * i, pc);
        if (mode == 0 ? ((double)pc[2] < 0.0) : (pc[2] < 0.0f)) continue;
        const float invz = mode == 0 ? 1 / pc[2] : (float)(1.0 / (double)pc[2]);
        const float x = pc[0] * invz, y = pc[1] * invz;
        const float u = cam->fx * x + cam->cx;
        const float v = cam->fy * y + cam->cy;
        if (!(u >= g->min_x && u < g->max_x && v >= g->min_y && v < g->max_y)) continue;
        float po[3];
        for (int k = 0; k < 3; k++) po[k] = pos[3 * i + k] - ow[k];
        const float dist = (float)sqrt((double)po[0] * po[0] + (double)po[1] * po[1] + (double)po[2] * po[2]);
        if (dist < 0.8f * min_distance[i] || dist > 1.2f * max_distance[i]) continue;
        const double dot = (double)po[0] * normal[3 * i] + (double)po[1] * normal[3 * i + 1] + (double)po[2] * normal[3 * i + 2];
        if (dot < 0.5 * (double)dist) continue;
        const int lvl = orc_predict_scale(max_distance[i], dist, log_scale_factor, n_levels);
        const float radius = th * scale_factors[lvl];
        const int nc = orc_features_in_area(g, u, v, radius, -1, -1, ind, N);
        if (nc == 0) continue;
        int best_dist = 256, best_idx = -1;
        for (int k = 0; k < nc; k++) {
            const int idx = ind[k];
            if (matched[idx]) continue;
            const int kp_level = g->keys_un[idx].octave;
            if (kp_level < lvl - 1 || kp_level > lvl) continue;
            const int d = orc_hamming256(pt_desc + 32 * (size_t)i, desc_kf + 32 * (size_t)idx);
            if (d < best_dist) { best_dist = d; best_idx = idx; }
        }
        if (best_dist <= TH_LOW) {
            pt_match[i] = best_idx;
            if (mode == 0) matched[best_idx] = 1;
            nmatches++;
        }
    }
    free(matched); free(ind);
    return nmatches;
}

/* one direction of ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1143-1216 / 1218-1291): map points of keyframe A (camera pose
 * Taw) moved into camera B by [sR|t] and searched in B's grid; match[i] = keypoint of B or -1. */
static void sim3_one_way(const float *Taw, const float *sRt /*3x4*/, const orc_grid *gb, const uint8_t *desc_b,
                         const float *scale_factors, const orc_camera *cam, float log_scale_factor, int n_levels,
                         int n, const float *pos, const float *max_distance, const float *min_distance, const uint8_t *desc,
                         const int32_t *valid, float th, int32_t *match, int32_t *ind)
{
    for (int i = 0; i < n; i++) {
        match[i] = -1;
        if (!valid[i]) continue;
        float pa[3], pb[3];
        rt_apply(Taw, pos + 3 * i, pa);
        rt_apply(sRt, pa, pb);
        if ((double)pb[2] < 0.0) continue;
        const float invz = (float)(1.0 / (double)pb[2]);
        const float x = pb[0] * invz, y = pb[1] * invz;
        const float u = cam->fx * x + cam->cx, v = cam->fy * y + cam->cy;
        if (!(u >= gb->min_x && u < gb->max_x && v >= gb->min_y && v < gb->max_y)) continue;
        const float dist = (float)sqrt((double)pb[0] * pb[0] + (double)pb[1] * pb[1] + (double)pb[2] * pb[2]);
        if (dist < 0.8f * min_distance[i] || dist > 1.2f * max_distance[i]) continue;
        const int lvl = orc_predict_scale(max_distance[i], dist, log_scale_factor, n_levels);
        const int nc = orc_features_in_area(gb, u, v, th * scale_factors[lvl], -1, -1, ind, gb->n);
        int best_dist = INT_MAX, best_idx = -1;
        for (int k = 0; k < nc; k++) {
            const int idx = ind[k];
            const int oct = gb->keys_un[idx].octave;
            if (oct < lvl - 1 || oct > lvl) continue;
            const int d = orc_hamming256(desc + 32 * (size_t)i, desc_b + 32 * (size_t)idx);
            if (d < best_dist) { best_dist = d; best_idx = idx; }
        }
        if (best_dist <= TH_HIGH) match[i] = best_idx;
    }
}

/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (src/ORBmatcher.cc:1098-1322).  Point arrays have one
 * entry per keypoint slot of their keyframe; valid1[i] = pMP && !isBad() && !vbAlreadyMatched1[i] (same for 2).  Both
 * keyframes use `cam`.  match12[i1] = i2 for mutually consistent pairs, else -1; returns their number. */
int orc_search_by_sim3(const orc_grid *g1, const uint8_t *desc_kf1, const float *T1w, const float *pos1, const float *maxd1, const float *mind1,
                       const uint8_t *pdesc1, const int32_t *valid1,
                       const orc_grid *g2, const uint8_t *desc_kf2, const float *T2w, const float *pos2, const float *maxd2, const float *mind2,
                       const uint8_t *pdesc2, const int32_t *valid2,
                       const float *scale_factors, const orc_camera *cam, float log_scale_factor, int n_levels,
                       float s12, const float *R12, const float *t12, float th, int32_t *match12)
{
    const int N1 = g1->n, N2 = g2->n;
    /* sR12 = s12 * R12; sR21 = (1.0 / s12) * R12.t(); t21 = -sR21 * t12  (:1116-1119; MatExpr scale = float multiply by (float)alpha) */
    float A12[12], A21[12];
    const float inv_s = (float)(1.0 / (double)s12);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { A12[4 * i + j] = R12[3 * i + j] * s12; A21[4 * i + j] = R12[3 * j + i] * inv_s; }
    for (int i = 0; i < 3; i++) {
        A12[4 * i + 3] = t12[i];
        const float t0 = (A21[4 * i] * t12[0] + A21[4 * i + 1] * t12[1]) + A21[4 * i + 2] * t12[2];
        A21[4 * i + 3] = -t0;
    }
    int32_t *m1 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N1 > 0 ? N1 : 1));
    int32_t *m2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N2 > 0 ? N2 : 1));
    int32_t *ind = (int32_t *)malloc(sizeof(int32_t) * (size_t)((N1 > N2 ? N1 : N2) > 0 ? (N1 > N2 ? N1 : N2) : 1));
    sim3_one_way(T1w, A21, g2, desc_kf2, scale_factors, cam, log_scale_factor, n_levels, N1, pos1, maxd1, mind1, pdesc1, valid1, th, m1, ind);
    sim3_one_way(T2w, A12, g1, desc_kf1, scale_factors, cam, log_scale_factor, n_levels, N2, pos2, maxd2, mind2, pdesc2, valid2, th, m2, ind);
    int n_found = 0;
    for (int i1 = 0; i1 < N1; i1++) {
        match12[i1] = -1;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && m2[idx2] == i1) { match12[i1] = idx2; n_found++; }
    }
    free(m1); free(m2); free(ind);
    return n_found;
}
