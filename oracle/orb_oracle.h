/*
 * orb_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's per-frame ORB front-end
 * (fabrizioromanelli/ORBSLAM2: src/ORBextractor.cc, src/Frame.cc:464-666,
 * src/ORBmatcher.cc) plus the OpenCV 4.5.5 routines that path calls
 * (cv::FAST, cv::resize INTER_LINEAR 8U, cv::GaussianBlur 8U fixed point,
 * cv::fastAtan2, cvRound).  Every function cites the reference file:line it
 * follows.
 *
 * PARITY UNPINNED: the reference ships no tests / golden vectors for this path
 * and OpenCV is not available in the authoring container, so this oracle is
 * pinned only by first-principles known-answer tests (tests/test_oracle_*.py)
 * and the committed fixtures generated from it (tests/golden/).  Routines that
 * restate OpenCV behaviour from knowledge of the 4.5.5 sources carry the tag
 * OPENCV-4.5.5-SEMANTICS.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (orbslam2_amd/csrc) never links it.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Layout-identical to cv::KeyPoint (28 bytes). */
typedef struct orc_keypoint {
    float x, y;      /* pt */
    float size;      /* patch diameter at level scale */
    float angle;     /* degrees [0,360) */
    float response;  /* FAST score */
    int32_t octave;
    int32_t class_id;
} orc_keypoint;

typedef struct orc_params {
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
    int32_t patch_size;
    int32_t half_patch_size;
    int32_t edge_threshold;
} orc_params;

typedef struct orc_extractor orc_extractor;

/* cv::cvtColor(..., COLOR_{RGB,BGR,RGBA,BGRA}2GRAY), 8-bit (src/Tracking.cc:269-294); see orb_oracle.c */
void orc_cvt_gray(const uint8_t *src, int w, int h, size_t stride, int cn, int rgb, int legacy14, uint8_t *dst, size_t dst_stride);

/* cv::undistortPoints(src, dst, K, D, Mat(), K) of Frame::UndistortKeyPoints / ComputeImageBounds (src/Frame.cc:402-462) */
void orc_undistort_points(const float *src_xy, int n, float fx, float fy, float cx, float cy, const float *dist, int ndist, float *dst_xy);
void orc_image_bounds(int cols, int rows, float fx, float fy, float cx, float cy, const float *dist, int ndist, float *bounds);

/* cv::remap(..., INTER_LINEAR), 8UC1, float maps, constant zero border (Test/Replay/Stereo/stereo_euroc.cc:136-137) */
void orc_remap_fixed(const float *mapx, const float *mapy, int n, int16_t *sx, int16_t *sy, uint16_t *alpha);
void orc_remap_bilinear(const uint8_t *src, int sw, int sh, size_t sstride, const float *mapx, const float *mapy,
                        int dw, int dh, uint8_t *dst, size_t dstride);

/* ---- scalar helpers (OpenCV semantics) ---- */
int orc_cv_round_f(float v);            /* cvRound(float): round-half-even */
int orc_cv_round_d(double v);           /* cvRound(double) */
float orc_fast_atan2(float y, float x); /* cv::fastAtan2, degrees */
void orc_sincos_det(float rad, float *s, float *c); /* contract Q4 routine */
int orc_border_reflect101(int p, int len);
int orc_hamming256(const uint8_t *a, const uint8_t *b); /* ORBmatcher::DescriptorDistance */
const int32_t *orc_bit_pattern(void);   /* 1024 ints */
void orc_gaussian_taps_q8(int ksize, double sigma, int32_t *taps); /* fixed-point 8.8 taps */

/* ---- image primitives ---- */
void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, size_t sstride,
                          uint8_t *dst, int dw, int dh, size_t dstride);
void orc_gaussian7_u8(const uint8_t *src, int w, int h, size_t sstride,
                      uint8_t *dst, size_t dstride);
/* cv::FAST(img, kps, threshold, nonmax=true), type 9_16.  Returns count; xs/ys/scores hold up to cap. */
int orc_fast9_16(const uint8_t *img, int w, int h, size_t stride, int threshold,
                 int nonmax, int32_t *xs, int32_t *ys, int32_t *scores, int cap);
/* cornerScore<16> at one pixel (needs 3-px margin). */
int orc_fast_corner_score(const uint8_t *ptr, size_t stride, int threshold);
/* closed-form score: max(t, max_arc min(v-p), max_arc min(p-v)) - 1 */
int orc_fast_score_closed_form(const uint8_t *ptr, size_t stride, int threshold);

/* ---- extractor ---- */
orc_extractor *orc_extractor_create(const orc_params *p);
void orc_extractor_destroy(orc_extractor *ex);
int orc_extractor_nlevels(const orc_extractor *ex);
const float *orc_extractor_scale_factors(const orc_extractor *ex);
const float *orc_extractor_inv_scale_factors(const orc_extractor *ex);
const float *orc_extractor_sigma2(const orc_extractor *ex);
const float *orc_extractor_inv_sigma2(const orc_extractor *ex);
const int32_t *orc_extractor_features_per_level(const orc_extractor *ex);
const int32_t *orc_extractor_umax(const orc_extractor *ex);
void orc_level_size(const orc_extractor *ex, int w, int h, int level, int *lw, int *lh);

/* ORBextractor::operator().  Returns number of keypoints (may exceed cap: then
 * only cap entries were written), or -1 on bad args.  Empty image -> 0. */
int orc_extract(orc_extractor *ex, const uint8_t *img, int w, int h, size_t stride,
                orc_keypoint *kps, uint8_t *desc, int cap);
/* pyramid of the latest orc_extract call (unblurred), as mvImagePyramid */
const uint8_t *orc_pyramid_level(const orc_extractor *ex, int level, int *w, int *h, size_t *stride);
/* candidates (pre-quadtree) of the latest call at one level: returns count; coords are region-relative */
int orc_level_candidates(const orc_extractor *ex, int level, const int32_t **xs, const int32_t **ys,
                         const int32_t **scores);

/* DistributeOctTree on integer candidates (region-relative coords).  Writes the
 * indices of the retained candidates in output order; returns their count. */
int orc_distribute_octtree(const int32_t *xs, const int32_t *ys, const int32_t *scores, int n,
                           int min_x, int max_x, int min_y, int max_y, int n_features,
                           int32_t *out_idx, int cap);

float orc_ic_angle(const uint8_t *img, size_t stride, int cx, int cy, const int32_t *umax, int half_patch);
void orc_orb_descriptor(const uint8_t *img, size_t stride, int cx, int cy, float angle_deg, uint8_t *desc32);
/* the same with an explicit test table (256 x (x0, y0, x1, y1)); NULL = the compiled bit_pattern_31_ */
void orc_orb_descriptor_pat(const uint8_t *img, size_t stride, int cx, int cy, float angle_deg, const int32_t *pat, uint8_t *desc32);
/* the extractor's own copy of the pattern (src/ORBextractor.cc:442-444) */
void orc_extractor_set_pattern(orc_extractor *ex, const int32_t *pat1024);
const int32_t *orc_extractor_pattern(const orc_extractor *ex);

/* ---- Frame::ComputeStereoMatches ---- */
/* Uses the pyramids held by exL/exR (latest orc_extract calls).  mb := bf/fx (Q1). */
int orc_stereo_matches(const orc_extractor *exL, const orc_extractor *exR,
                       const orc_keypoint *kL, const uint8_t *dL, int nL,
                       const orc_keypoint *kR, const uint8_t *dR, int nR,
                       float bf, float fx, float *u_right, float *depth,
                       int32_t *best_idx_r /* optional, -1 when no coarse match */,
                       int32_t *best_sad /* optional */);

/* Frame::ComputeStereoFromRGBD */
void orc_stereo_from_rgbd(const orc_keypoint *k, const orc_keypoint *k_un, int n,
                          const float *depth, int w, int h, size_t stride_floats,
                          float bf, float *u_right, float *out_depth);

/* ---- Tracking-thread matchers on flattened inputs (orb_oracle_match.c) ---- */
typedef struct orc_grid orc_grid;
typedef struct orc_camera { float fx, fy, cx, cy, bf, mb; } orc_camera;
/* what Frame::isInFrustum leaves in a MapPoint (mbTrackInView, mTrackProjX/Y/XR, mnTrackScaleLevel, mTrackViewCos) */
typedef struct orc_track_point { int32_t in_view; float proj_x, proj_y, proj_xr; int32_t level; float view_cos; } orc_track_point;

orc_grid *orc_grid_create(const orc_keypoint *keys_un, int n, float min_x, float max_x, float min_y, float max_y);
void orc_grid_destroy(orc_grid *g);
void orc_grid_as_keyframe(orc_grid *g); /* bounds -> the ints a KeyFrame keeps (queries and IsInImage only; cells stay) */
int orc_features_in_area(const orc_grid *g, float x, float y, float r, int min_level, int max_level, int32_t *out, int cap);
void orc_three_maxima(const int32_t *histo_sizes, int L, int *ind1, int *ind2, int *ind3);
float orc_log_det(float x);
int orc_predict_scale(float max_distance, float current_dist, float log_scale_factor, int n_levels);
int orc_search_by_projection_last(const orc_grid *g, const float *u_right_cur, const uint8_t *desc_cur,
                                  const float *scale_factors, const orc_camera *cam,
                                  const float *Tcw_cur, const float *Tcw_last,
                                  int n_last, const float *last_pos, const uint8_t *last_desc,
                                  const int32_t *last_valid, const int32_t *last_obs, const int32_t *last_octave,
                                  const float *last_angle, const uint8_t *cur_has_obs_in,
                                  float th, int mono, int check_ori, int32_t *cur_match);
int orc_is_in_frustum(const float *Tcw, const orc_camera *cam, float min_x, float max_x, float min_y, float max_y,
                      const float *pos, const float *normal, float max_dist_inv, float min_dist_inv, float max_distance,
                      float viewing_cos_limit, float log_scale_factor, int n_levels, orc_track_point *out);
int orc_search_by_projection_points(const orc_grid *g, const float *u_right_cur, const uint8_t *desc_cur,
                                    const float *scale_factors, int n_pts, const orc_track_point *pts,
                                    const uint8_t *pt_desc, const int32_t *pt_obs, const uint8_t *cur_has_obs_in,
                                    float th, float nnratio, int32_t *cur_match);
int orc_search_by_projection_kf(const orc_grid *g, const uint8_t *desc_cur, const float *scale_factors,
                                const orc_camera *cam, const float *Tcw_cur, float log_scale_factor, int n_levels,
                                int n_kf, const float *kf_pos, const uint8_t *kf_desc, const int32_t *kf_valid,
                                const float *kf_angle, const float *kf_max_distance, const float *kf_min_distance,
                                const uint8_t *cur_has_point_in, float th, int orb_dist, int check_ori, int32_t *cur_match);
int orc_search_for_initialization(const orc_keypoint *keys1, const uint8_t *desc1, int n1,
                                  const orc_grid *g2, const uint8_t *desc2,
                                  float *prev_matched, int window_size, float nnratio, int check_ori, int32_t *matches12);

/* ---- fbow vocabulary transform + SearchByFboW (orb_oracle_bow.c) ---- */
typedef struct orc_vocab orc_vocab;
orc_vocab *orc_vocab_from_blob(const uint8_t *blob, size_t size);
void orc_vocab_destroy(orc_vocab *v);
int orc_vocab_k(const orc_vocab *v);
int orc_vocab_nblocks(const orc_vocab *v);
void orc_bow_descend(const orc_vocab *v, const uint8_t *desc, int n, int store_level,
                     uint32_t *word_id, float *weight, uint32_t *node_id);
int orc_bow_maps(const uint32_t *word_id, const float *weight, const uint32_t *node_id, int n,
                 uint32_t *words, float *word_w, uint32_t *nodes, int32_t *node_off, int32_t *node_feat, int *n_nodes);
int orc_search_by_bow(const uint32_t *kf_nodes, const int32_t *kf_off, const int32_t *kf_feat, int kf_nnodes,
                      const int32_t *kf_valid, const uint8_t *kf_desc, const float *kf_angle,
                      const uint32_t *f_nodes, const int32_t *f_off, const int32_t *f_feat, int f_nnodes,
                      const uint8_t *f_desc, const float *f_angle, int n_f,
                      float nnratio, int check_ori, int32_t *f_match);
int orc_search_by_bow_kf(const uint32_t *n1, const int32_t *off1, const int32_t *feat1, int nn1,
                         const int32_t *valid1, const uint8_t *desc1, const float *angle1, int nk1,
                         const uint32_t *n2, const int32_t *off2, const int32_t *feat2, int nn2,
                         const int32_t *valid2, const uint8_t *desc2, const float *angle2, int nk2,
                         float nnratio, int check_ori, int32_t *match12);
int orc_fuse(const orc_grid *g, const float *u_right_kf, const uint8_t *desc_kf, const float *scale_factors, const float *inv_level_sigma2,
             const orc_camera *cam, const float *Tcw, float log_scale_factor, int n_levels,
             int n_pts, const float *pos, const float *normal, const float *max_distance, const float *min_distance,
             const uint8_t *pt_desc, const int32_t *pt_valid, float th, int32_t *best_idx_out);
int orc_search_by_sim3_projection(int mode, const orc_grid *g, const uint8_t *desc_kf, const float *scale_factors, const orc_camera *cam,
                                  const float *Scw, float log_scale_factor, int n_levels,
                                  int n_pts, const float *pos, const float *normal, const float *max_distance, const float *min_distance,
                                  const uint8_t *pt_desc, const int32_t *pt_valid, const uint8_t *kf_matched_in, float th, int32_t *pt_match);
int orc_search_by_sim3(const orc_grid *g1, const uint8_t *desc_kf1, const float *T1w, const float *pos1, const float *maxd1, const float *mind1,
                       const uint8_t *pdesc1, const int32_t *valid1,
                       const orc_grid *g2, const uint8_t *desc_kf2, const float *T2w, const float *pos2, const float *maxd2, const float *mind2,
                       const uint8_t *pdesc2, const int32_t *valid2,
                       const float *scale_factors, const orc_camera *cam, float log_scale_factor, int n_levels,
                       float s12, const float *R12, const float *t12, float th, int32_t *match12);
/* Optimizer::PoseOptimization (src/Optimizer.cc:283-495) with the vendored g2o Levenberg solver; see orb_oracle_pose.c */
int orc_pose_optimization(float *Tcw, int N, const orc_keypoint *keys_un, const float *u_right, const uint8_t *has_point,
                          const float *Xw, const float *inv_level_sigma2, float fx, float fy, float cx, float cy, float bf,
                          uint8_t *outlier);
double orc_bow_score(const uint32_t *w1, const float *v1, int n1, const uint32_t *w2, const float *v2, int n2);
int orc_detect_reloc_candidates(const uint32_t *q_words, const float *q_w, int nq,
                                int n_kf, const int32_t *kf_off, const uint32_t *db_words, const float *db_w,
                                const int32_t *covis_off, const int32_t *covis_idx, float *reloc_score, int32_t *cand, int cap);
int orc_detect_loop_candidates(const uint32_t *q_words, const float *q_w, int nq,
                               int n_kf, const int32_t *kf_off, const uint32_t *db_words, const float *db_w,
                               const uint8_t *connected, float min_score,
                               const int32_t *covis_off, const int32_t *covis_idx, int32_t *cand, int cap);
int orc_search_for_triangulation(const uint32_t *n1, const int32_t *off1, const int32_t *feat1, int nn1,
                                 const orc_keypoint *k1, const float *ur1, const uint8_t *has_mp1, const uint8_t *desc1, int nk1,
                                 const uint32_t *n2, const int32_t *off2, const int32_t *feat2, int nn2,
                                 const orc_keypoint *k2, const float *ur2, const uint8_t *has_mp2, const uint8_t *desc2, int nk2,
                                 const float *F12, const float *Cw1, const float *T2w, float fx2, float fy2, float cx2, float cy2,
                                 const float *scale_factors, const float *level_sigma2, int only_stereo, int check_ori, int32_t *match12);

#ifdef __cplusplus
}
#endif
#endif
