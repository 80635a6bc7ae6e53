/*
 * orb_oracle_bow.c -- CPU ORACLE (test infrastructure, NOT product code): fbow vocabulary transform
 * and ORBmatcher::SearchByFboW(KeyFrame*, Frame&) on flattened inputs.
 * PARITY UNPINNED (see orb_oracle.h).  fbow is vendored in the reference, so this restates source
 * that is present: Thirdparty/fbow/src/fbow.h:134-198,342-351,400-444, fbow.cpp:10-49,172-191.
 */
#include "orb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define TH_LOW 50
#define HISTO_LENGTH 30

/* fbow::Vocabulary::params, Thirdparty/fbow/src/fbow.h:118-129 (120 bytes on LP64) */
typedef struct fbow_params {
    char desc_name[50];
    uint32_t aligment, nblocks;
    uint64_t desc_size_bytes_wp, block_size_bytes_wp, feature_off_start, child_off_start, total_size;
    int32_t desc_type, desc_size;
    uint32_t m_k;
} fbow_params;

struct orc_vocab {
    fbow_params p;
    uint8_t *data;
};

/* fbow::Vocabulary::fromStream, Thirdparty/fbow/src/fbow.cpp:181-191 */
orc_vocab *orc_vocab_from_blob(const uint8_t *blob, size_t size)
{
    if (!blob || size < 8 + sizeof(fbow_params)) return NULL;
    uint64_t sig;
    memcpy(&sig, blob, 8);
    if (sig != 55824124ull) return NULL;
    orc_vocab *v = (orc_vocab *)calloc(1, sizeof(*v));
    memcpy(&v->p, blob + 8, sizeof(fbow_params));
    if (sizeof(fbow_params) != 120 || size < 8 + 120 + v->p.total_size || v->p.desc_size != 32 || v->p.m_k == 0 ||
        v->p.total_size != v->p.block_size_bytes_wp * (uint64_t)v->p.nblocks) {
        free(v);
        return NULL;
    }
    v->data = (uint8_t *)malloc(v->p.total_size);
    memcpy(v->data, blob + 8 + 120, v->p.total_size);
    return v;
}

void orc_vocab_destroy(orc_vocab *v)
{
    if (!v) return;
    free(v->data);
    free(v);
}

int orc_vocab_k(const orc_vocab *v) { return (int)v->p.m_k; }
int orc_vocab_nblocks(const orc_vocab *v) { return (int)v->p.nblocks; }

static uint64_t pop64(uint64_t x) { return (uint64_t)__builtin_popcountll(x); }

/* Vocabulary::_transform2<L1_32bytes>, Thirdparty/fbow/src/fbow.h:400-444: per feature the leaf word id,
 * its weight and the node id reached at store_level (node_id[i] = 0xffffffff never happens: every feature
 * records exactly one node).  The maps fBow / fBow2 are rebuilt from these by orc_bow_maps. */
void orc_bow_descend(const orc_vocab *v, const uint8_t *desc, int n, int store_level,
                     uint32_t *word_id, float *weight, uint32_t *node_id)
{
    const fbow_params *p = &v->p;
    const int nbits = (int)ceil(log2((double)p->m_k));
    for (int f = 0; f < n; f++) {
        uint64_t feat[4];
        memcpy(feat, desc + (size_t)32 * f, 32);
        const uint8_t *blk = v->data; /* block 0 */
        uint32_t level = 0, cur_node = 0;
        node_id[f] = 0; word_id[f] = 0; weight[f] = 0.f;
        for (;;) {
            const int N = *(const uint16_t *)blk;
            uint64_t best_d = 0xffffffffull; /* numeric_limits<uint32_t>::max() */
            uint32_t best_i = 0;
            for (int c = 0; c < N; c++) {
                uint64_t nf[4];
                memcpy(nf, blk + p->feature_off_start + (size_t)c * p->desc_size_bytes_wp, 32);
                const uint64_t d = pop64(nf[0] ^ feat[0]) + pop64(nf[1] ^ feat[1]) + pop64(nf[2] ^ feat[2]) + pop64(nf[3] ^ feat[3]);
                if (d < best_d) { best_d = d; best_i = (uint32_t)c; }
            }
            if (level == (uint32_t)store_level) node_id[f] = cur_node;
            uint32_t id_or_child;
            float w;
            memcpy(&id_or_child, blk + p->child_off_start + (size_t)best_i * 8, 4);
            memcpy(&w, blk + p->child_off_start + (size_t)best_i * 8 + 4, 4);
            if (id_or_child & 0x80000000u) {
                word_id[f] = id_or_child & 0x7fffffffu;
                weight[f] = w;
                if (level < (uint32_t)store_level) node_id[f] = cur_node;
                break;
            }
            const uint32_t child = id_or_child & 0x7fffffffu;
            blk = v->data + (size_t)child * p->block_size_bytes_wp;
            cur_node = (cur_node << nbits) | best_i;
            level++;
            if (child == 0) break; /* while(!isleaf && getId()!=0) */
        }
    }
}

static int cmp_u32(const void *a, const void *b)
{
    const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* fBow (word -> summed weight, in feature order) and fBow2 (node -> feature indices, ascending) as sorted arrays.
 * Returns the number of distinct words; *n_nodes the number of distinct nodes.  node_off has n_nodes+1 entries. */
int orc_bow_maps(const uint32_t *word_id, const float *weight, const uint32_t *node_id, int n,
                 uint32_t *words, float *word_w, uint32_t *nodes, int32_t *node_off, int32_t *node_feat, int *n_nodes)
{
    uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    memcpy(tmp, word_id, sizeof(uint32_t) * (size_t)n);
    qsort(tmp, (size_t)n, sizeof(uint32_t), cmp_u32);
    int nw = 0;
    for (int i = 0; i < n; i++) if (i == 0 || tmp[i] != tmp[i - 1]) words[nw++] = tmp[i];
    for (int w = 0; w < nw; w++) word_w[w] = 0.f;
    for (int f = 0; f < n; f++) { /* r1[id] += weight, feature order */
        const uint32_t *pos = (const uint32_t *)bsearch(&word_id[f], words, (size_t)nw, sizeof(uint32_t), cmp_u32);
        word_w[pos - words] += weight[f];
    }
    memcpy(tmp, node_id, sizeof(uint32_t) * (size_t)n);
    qsort(tmp, (size_t)n, sizeof(uint32_t), cmp_u32);
    int nn = 0;
    for (int i = 0; i < n; i++) if (i == 0 || tmp[i] != tmp[i - 1]) nodes[nn++] = tmp[i];
    for (int k = 0; k <= nn; k++) node_off[k] = 0;
    for (int f = 0; f < n; f++) {
        const uint32_t *pos = (const uint32_t *)bsearch(&node_id[f], nodes, (size_t)nn, sizeof(uint32_t), cmp_u32);
        node_off[(pos - nodes) + 1]++;
    }
    for (int k = 0; k < nn; k++) node_off[k + 1] += node_off[k];
    int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nn > 0 ? nn : 1));
    memcpy(cur, node_off, sizeof(int32_t) * (size_t)nn);
    for (int f = 0; f < n; f++) {
        const uint32_t *pos = (const uint32_t *)bsearch(&node_id[f], nodes, (size_t)nn, sizeof(uint32_t), cmp_u32);
        node_feat[cur[pos - nodes]++] = f;
    }
    free(cur); free(tmp);
    *n_nodes = nn;
    return nw;
}

static int rot_bin(float a1, float a2)
{
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

/* ORBmatcher::SearchByFboW(KeyFrame*, Frame&, vpMapPointMatches), reference src/ORBmatcher.cc:157-283.
 * Feature vectors come as the sorted arrays of orc_bow_maps.  kf_valid[i] = KF keypoint i has a good map point.
 * f_match[j] receives the KF keypoint index whose map point was given to frame keypoint j, or -1. */
int orc_search_by_bow(const uint32_t *kf_nodes, const int32_t *kf_off, const int32_t *kf_feat, int kf_nnodes,
                      const int32_t *kf_valid, const uint8_t *kf_desc, const float *kf_angle,
                      const uint32_t *f_nodes, const int32_t *f_off, const int32_t *f_feat, int f_nnodes,
                      const uint8_t *f_desc, const float *f_angle, int n_f,
                      float nnratio, int check_ori, int32_t *f_match)
{
    int nmatches = 0;
    for (int j = 0; j < n_f; j++) f_match[j] = -1;
    int *hist[HISTO_LENGTH], hn[HISTO_LENGTH], hc[HISTO_LENGTH];
    for (int b = 0; b < HISTO_LENGTH; b++) { hist[b] = NULL; hn[b] = 0; hc[b] = 0; }
    int a = 0, b = 0;
    while (a < kf_nnodes && b < f_nnodes) {
        if (kf_nodes[a] == f_nodes[b]) {
            for (int ik = kf_off[a]; ik < kf_off[a + 1]; ik++) {
                const int real_kf = kf_feat[ik];
                if (!kf_valid[real_kf]) continue;
                int best1 = 256, best_f = -1, best2 = 256;
                for (int jf = f_off[b]; jf < f_off[b + 1]; jf++) {
                    const int real_f = f_feat[jf];
                    if (f_match[real_f] >= 0) continue;
                    const int dist = orc_hamming256(kf_desc + (size_t)32 * real_kf, f_desc + (size_t)32 * real_f);
                    if (dist < best1) { best2 = best1; best1 = dist; best_f = real_f; }
                    else if (dist < best2) best2 = dist;
                }
                if (best1 <= TH_LOW && (float)best1 < nnratio * (float)best2) {
                    f_match[best_f] = real_kf;
                    if (check_ori) {
                        const int bin = rot_bin(kf_angle[real_kf], f_angle[best_f]);
                        if (hn[bin] == hc[bin]) { hc[bin] = hc[bin] ? 2 * hc[bin] : 64; hist[bin] = (int *)realloc(hist[bin], sizeof(int) * (size_t)hc[bin]); }
                        hist[bin][hn[bin]++] = best_f;
                    }
                    nmatches++;
                }
            }
            a++; b++;
        } else if (kf_nodes[a] < f_nodes[b]) {
            while (a < kf_nnodes && kf_nodes[a] < f_nodes[b]) a++; /* lower_bound */
        } else {
            while (b < f_nnodes && f_nodes[b] < kf_nodes[a]) b++;
        }
    }
    if (check_ori) {
        int i1, i2, i3;
        orc_three_maxima(hn, HISTO_LENGTH, &i1, &i2, &i3);
        for (int k = 0; k < HISTO_LENGTH; k++) {
            if (k == i1 || k == i2 || k == i3) continue;
            for (int j = 0; j < hn[k]; j++) { f_match[hist[k][j]] = -1; nmatches--; }
        }
    }
    for (int k = 0; k < HISTO_LENGTH; k++) free(hist[k]);
    return nmatches;
}

/* ORBmatcher::SearchByFboW(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12) (src/ORBmatcher.cc:517-650): like the
 * (KeyFrame, Frame) overload, but both sides need a good map point (valid1 / valid2), a KF2 feature is taken at most
 * once (vbMatched2), the distance bound is strict (bestDist1 < TH_LOW, :590) and the result is indexed by the KF1
 * feature: match12[idx1] = idx2 or -1. */
int orc_search_by_bow_kf(const uint32_t *n1, const int32_t *off1, const int32_t *feat1, int nn1,
                         const int32_t *valid1, const uint8_t *desc1, const float *angle1, int nk1,
                         const uint32_t *n2, const int32_t *off2, const int32_t *feat2, int nn2,
                         const int32_t *valid2, const uint8_t *desc2, const float *angle2, int nk2,
                         float nnratio, int check_ori, int32_t *match12)
{
    int nmatches = 0;
    for (int i = 0; i < nk1; i++) match12[i] = -1;
    uint8_t *matched2 = (uint8_t *)calloc((size_t)(nk2 > 0 ? nk2 : 1), 1);
    int *hist[HISTO_LENGTH], hn[HISTO_LENGTH], hc[HISTO_LENGTH];
    for (int b = 0; b < HISTO_LENGTH; b++) { hist[b] = NULL; hn[b] = 0; hc[b] = 0; }
    int a = 0, b = 0;
    while (a < nn1 && b < nn2) {
        if (n1[a] == n2[b]) {
            for (int i1 = off1[a]; i1 < off1[a + 1]; i1++) {
                const int idx1 = feat1[i1];
                if (!valid1[idx1]) continue;
                int best1 = 256, best_idx2 = -1, best2 = 256;
                for (int i2 = off2[b]; i2 < off2[b + 1]; i2++) {
                    const int idx2 = feat2[i2];
                    if (matched2[idx2] || !valid2[idx2]) continue;
                    const int dist = orc_hamming256(desc1 + (size_t)32 * idx1, desc2 + (size_t)32 * idx2);
                    if (dist < best1) { best2 = best1; best1 = dist; best_idx2 = idx2; }
                    else if (dist < best2) best2 = dist;
                }
                if (best1 < TH_LOW && (float)best1 < nnratio * (float)best2) {
                    match12[idx1] = best_idx2;
                    matched2[best_idx2] = 1;
                    if (check_ori) {
                        const int bin = rot_bin(angle1[idx1], angle2[best_idx2]);
                        if (hn[bin] == hc[bin]) { hc[bin] = hc[bin] ? 2 * hc[bin] : 64; hist[bin] = (int *)realloc(hist[bin], sizeof(int) * (size_t)hc[bin]); }
                        hist[bin][hn[bin]++] = idx1;
                    }
                    nmatches++;
                }
            }
            a++; b++;
        } else if (n1[a] < n2[b]) {
            while (a < nn1 && n1[a] < n2[b]) a++; /* lower_bound */
        } else {
            while (b < nn2 && n2[b] < n1[a]) b++;
        }
    }
    if (check_ori) {
        int i1, i2, i3;
        orc_three_maxima(hn, HISTO_LENGTH, &i1, &i2, &i3);
        for (int k = 0; k < HISTO_LENGTH; k++) {
            if (k == i1 || k == i2 || k == i3) continue;
            for (int j = 0; j < hn[k]; j++) { match12[hist[k][j]] = -1; nmatches--; }
        }
    }
    for (int k = 0; k < HISTO_LENGTH; k++) free(hist[k]);
    free(matched2);
    (void)nk2;
    return nmatches;
}

/* fbow::fBow::score (Thirdparty/fbow/src/fbow.cpp:206-256): both vectors sorted by word id; the products are floats
 * (the map's value type converts to float), accumulated in double in ascending word order. */
double orc_bow_score(const uint32_t *w1, const float *v1, int n1, const uint32_t *w2, const float *v2, int n2)
{
    double score = 0;
    int a = 0, b = 0;
    while (a < n1 && b < n2) {
        if (w1[a] == w2[b]) { const float p = v1[a] * v2[b]; score += p; a++; b++; }
        else if (w1[a] < w2[b]) { while (a < n1 && w1[a] < w2[b]) a++; }
        else { while (b < n2 && w2[b] < w1[a]) b++; }
    }
    if (score >= 1) score = 1.0;
    else score = 1.0 - sqrt(1.0 - score);
    return score;
}

/* KeyFrameDatabase::DetectRelocalizationCandidates (src/KeyFrameDatabase.cc:196-307).  The database is n_kf BoW
 * vectors in CSR form (kf_off, db_words ascending per keyframe, db_w); the inverted file lists keyframes in index order
 * (KeyFrameDatabase::add appends, :38-44).  covis = each keyframe's GetBestCovisibilityKeyFrames(10) list in its order.
 * reloc_score[] is the keyframes' persistent mRelocScore (in/out): the reference never initialises it and only updates it
 * for keyframes that pass the common-word filter, so a neighbour that merely shares a word contributes its OLD value
 * (contract Q10: the caller owns that state; start from zeros).  Returns the number of candidates written to cand[]. */
int orc_detect_reloc_candidates(const uint32_t *q_words, const float *q_w, int nq,
                                int n_kf, const int32_t *kf_off, const uint32_t *db_words, const float *db_w,
                                const int32_t *covis_off, const int32_t *covis_idx, float *reloc_score, int32_t *cand, int cap)
{
    /* inverted file: word -> keyframes (index order) */
    uint32_t max_word = 0;
    for (int i = 0; i < kf_off[n_kf]; i++) if (db_words[i] > max_word) max_word = db_words[i];
    for (int i = 0; i < nq; i++) if (q_words[i] > max_word) max_word = q_words[i];
    const size_t nw = (size_t)max_word + 2;
    int *inv_off = (int *)calloc(nw + 1, sizeof(int));
    for (int i = 0; i < kf_off[n_kf]; i++) inv_off[db_words[i] + 1]++;
    for (size_t w = 0; w < nw; w++) inv_off[w + 1] += inv_off[w];
    int *inv = (int *)malloc(sizeof(int) * (size_t)(kf_off[n_kf] > 0 ? kf_off[n_kf] : 1));
    int *cur = (int *)malloc(sizeof(int) * (nw + 1));
    memcpy(cur, inv_off, sizeof(int) * (nw + 1));
    for (int k = 0; k < n_kf; k++)
        for (int i = kf_off[k]; i < kf_off[k + 1]; i++) inv[cur[db_words[i]]++] = k;
    /* keyframes sharing a word, in first-encounter order (:205-221) */
    int *words = (int *)calloc((size_t)(n_kf > 0 ? n_kf : 1), sizeof(int));
    uint8_t *seen = (uint8_t *)calloc((size_t)(n_kf > 0 ? n_kf : 1), 1);
    int *sharing = (int *)malloc(sizeof(int) * (size_t)(n_kf > 0 ? n_kf : 1));
    int n_sh = 0;
    for (int i = 0; i < nq; i++)
        for (int j = inv_off[q_words[i]]; j < inv_off[q_words[i] + 1]; j++) {
            const int k = inv[j];
            if (!seen[k]) { seen[k] = 1; words[k] = 0; sharing[n_sh++] = k; }
            words[k]++;
        }
    int n_out = 0;
    if (n_sh > 0) {
        int max_common = 0;
        for (int i = 0; i < n_sh; i++) if (words[sharing[i]] > max_common) max_common = words[sharing[i]];
        const int min_common = (int)((float)max_common * 0.8f);
        int *sm_kf = (int *)malloc(sizeof(int) * (size_t)n_sh);
        float *sm_s = (float *)malloc(sizeof(float) * (size_t)n_sh);
        int n_sm = 0;
        for (int i = 0; i < n_sh; i++) {
            const int k = sharing[i];
            if (words[k] > min_common) {
                const float si = (float)orc_bow_score(q_words, q_w, nq, db_words + kf_off[k], db_w + kf_off[k], kf_off[k + 1] - kf_off[k]);
                reloc_score[k] = si;
                sm_kf[n_sm] = k; sm_s[n_sm] = si; n_sm++;
            }
        }
        if (n_sm > 0) {
            float *acc = (float *)malloc(sizeof(float) * (size_t)n_sm);
            int *best_kf = (int *)malloc(sizeof(int) * (size_t)n_sm);
            float best_acc = 0;
            for (int i = 0; i < n_sm; i++) { /* accumulate score by covisibility (:254-277) */
                const int k = sm_kf[i];
                float best = sm_s[i], a = best;
                int bk = k;
                int nn = covis_off[k + 1] - covis_off[k];
                if (nn > 10) nn = 10;
                for (int j = 0; j < nn; j++) {
                    const int k2 = covis_idx[covis_off[k] + j];
                    if (!seen[k2]) continue;
                    a += reloc_score[k2];
                    if (reloc_score[k2] > best) { bk = k2; best = reloc_score[k2]; }
                }
                acc[i] = a; best_kf[i] = bk;
                if (a > best_acc) best_acc = a;
            }
            const float min_retain = 0.75f * best_acc;
            uint8_t *added = (uint8_t *)calloc((size_t)n_kf, 1);
            for (int i = 0; i < n_sm; i++)
                if (acc[i] > min_retain && !added[best_kf[i]]) {
                    added[best_kf[i]] = 1;
                    if (n_out < cap) cand[n_out] = best_kf[i];
                    n_out++;
                }
            free(added); free(acc); free(best_kf);
        }
        free(sm_kf); free(sm_s);
    }
    free(inv_off); free(inv); free(cur); free(words); free(seen); free(sharing);
    return n_out;
}

/* ORBmatcher::CheckDistEpipolarLine (src/ORBmatcher.cc:138-155): float arithmetic left to right, the final comparison
 * in double (3.84 is a double literal). */
static int check_dist_epipolar_line(float x1, float y1, float x2, float y2, const float *F12 /*3x3 row major*/, float sigma2_kp2)
{
    const float a = x1 * F12[0] + y1 * F12[3] + F12[6];
    const float b = x1 * F12[1] + y1 * F12[4] + F12[7];
    const float c = x1 * F12[2] + y1 * F12[5] + F12[8];
    const float num = a * x2 + b * y2 + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return (double)dsqr < 3.84 * (double)sigma2_kp2;
}

/* ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:652-819): keypoints WITHOUT a map point in both keyframes,
 * paired inside shared vocabulary nodes, Hamming <= TH_LOW, away from the epipole (monocular pairs) and close to the
 * epipolar line of F12.  has_mp = the keypoint already has a map point; u_right < 0 = monocular keypoint.  Cw1 = camera
 * centre of KF1, T2w = [R|t] of KF2 (3x4).  match12[idx1] = idx2 or -1; the reference's pair list is its non-negative
 * entries in index order (:808-814). */
int orc_search_for_triangulation(const uint32_t *n1, const int32_t *off1, const int32_t *feat1, int nn1,
                                 const orc_keypoint *k1, const float *ur1, const uint8_t *has_mp1, const uint8_t *desc1, int nk1,
                                 const uint32_t *n2, const int32_t *off2, const int32_t *feat2, int nn2,
                                 const orc_keypoint *k2, const float *ur2, const uint8_t *has_mp2, const uint8_t *desc2, int nk2,
                                 const float *F12, const float *Cw1, const float *T2w, float fx2, float fy2, float cx2, float cy2,
                                 const float *scale_factors, const float *level_sigma2, int only_stereo, int check_ori, int32_t *match12)
{
    /* epipole in the second image (:658-664); cv::Mat R*x+t in float, small-matrix order */
    float C2[3];
    for (int i = 0; i < 3; i++) {
        const float t = (T2w[4 * i] * Cw1[0] + T2w[4 * i + 1] * Cw1[1]) + T2w[4 * i + 2] * Cw1[2];
        C2[i] = t + T2w[4 * i + 3];
    }
    const float invz = 1.0f / C2[2];
    const float ex = fx2 * C2[0] * invz + cx2;
    const float ey = fy2 * C2[1] * invz + cy2;
    int nmatches = 0;
    for (int i = 0; i < nk1; i++) match12[i] = -1;
    uint8_t *matched2 = (uint8_t *)calloc((size_t)(nk2 > 0 ? nk2 : 1), 1);
    int *hist[HISTO_LENGTH], hn[HISTO_LENGTH], hc[HISTO_LENGTH];
    for (int b = 0; b < HISTO_LENGTH; b++) { hist[b] = NULL; hn[b] = 0; hc[b] = 0; }
    int a = 0, b = 0;
    while (a < nn1 && b < nn2) {
        if (n1[a] == n2[b]) {
            for (int i1 = off1[a]; i1 < off1[a + 1]; i1++) {
                const int idx1 = feat1[i1];
                if (has_mp1[idx1]) continue;
                const int stereo1 = ur1[idx1] >= 0;
                if (only_stereo && !stereo1) continue;
                int best_dist = TH_LOW, best_idx2 = -1;
                for (int i2 = off2[b]; i2 < off2[b + 1]; i2++) {
                    const int idx2 = feat2[i2];
                    if (matched2[idx2] || has_mp2[idx2]) continue;
                    const int stereo2 = ur2[idx2] >= 0;
                    if (only_stereo && !stereo2) continue;
                    const int dist = orc_hamming256(desc1 + (size_t)32 * idx1, desc2 + (size_t)32 * idx2);
                    if (dist > TH_LOW || dist > best_dist) continue;
                    if (!stereo1 && !stereo2) {
                        const float distex = ex - k2[idx2].x, distey = ey - k2[idx2].y;
                        if (distex * distex + distey * distey < 100 * scale_factors[k2[idx2].octave]) continue;
                    }
                    if (check_dist_epipolar_line(k1[idx1].x, k1[idx1].y, k2[idx2].x, k2[idx2].y, F12, level_sigma2[k2[idx2].octave])) {
                        best_idx2 = idx2;
                        best_dist = dist;
                    }
                }
                if (best_idx2 >= 0) {
                    match12[idx1] = best_idx2;
                    matched2[best_idx2] = 1;
                    nmatches++;
                    if (check_ori) {
                        const int bin = rot_bin(k1[idx1].angle, k2[best_idx2].angle);
                        if (hn[bin] == hc[bin]) { hc[bin] = hc[bin] ? 2 * hc[bin] : 64; hist[bin] = (int *)realloc(hist[bin], sizeof(int) * (size_t)hc[bin]); }
                        hist[bin][hn[bin]++] = idx1;
                    }
                }
            }
            a++; b++;
        } else if (n1[a] < n2[b]) {
            while (a < nn1 && n1[a] < n2[b]) a++;
        } else {
            while (b < nn2 && n2[b] < n1[a]) b++;
        }
    }
    if (check_ori) {
        int i1, i2, i3;
        orc_three_maxima(hn, HISTO_LENGTH, &i1, &i2, &i3);
        for (int k = 0; k < HISTO_LENGTH; k++) {
            if (k == i1 || k == i2 || k == i3) continue;
            for (int j = 0; j < hn[k]; j++) { match12[hist[k][j]] = -1; nmatches--; }
        }
    }
    for (int k = 0; k < HISTO_LENGTH; k++) free(hist[k]);
    free(matched2);
    return nmatches;
}


/* KeyFrameDatabase::DetectLoopCandidates(KeyFrame *pKF, float minScore) (src/KeyFrameDatabase.cc:73-194).  connected[k] != 0:
 * keyframe k is in pKF->GetConnectedKeyFrames() (never a candidate, never counted in the covisibility accumulation);
 * covis lists = every keyframe's GetBestCovisibilityKeyFrames(10).  Unlike the relocalisation query no state survives the
 * call: mLoopScore is only read for keyframes scored in this query (mnLoopQuery == id && mnLoopWords > minCommonWords). */
int orc_detect_loop_candidates(const uint32_t *q_words, const float *q_w, int nq,
                               int n_kf, const int32_t *kf_off, const uint32_t *db_words, const float *db_w,
                               const uint8_t *connected, float min_score,
                               const int32_t *covis_off, const int32_t *covis_idx, int32_t *cand, int cap)
{
    uint32_t max_word = 0;
    for (int i = 0; i < kf_off[n_kf]; i++) if (db_words[i] > max_word) max_word = db_words[i];
    for (int i = 0; i < nq; i++) if (q_words[i] > max_word) max_word = q_words[i];
    const size_t nw = (size_t)max_word + 2;
    int *inv_off = (int *)calloc(nw + 1, sizeof(int));
    for (int i = 0; i < kf_off[n_kf]; i++) inv_off[db_words[i] + 1]++;
    for (size_t w = 0; w < nw; w++) inv_off[w + 1] += inv_off[w];
    int *inv = (int *)malloc(sizeof(int) * (size_t)(kf_off[n_kf] > 0 ? kf_off[n_kf] : 1));
    int *cur = (int *)malloc(sizeof(int) * (nw + 1));
    memcpy(cur, inv_off, sizeof(int) * (nw + 1));
    for (int k = 0; k < n_kf; k++)
        for (int i = kf_off[k]; i < kf_off[k + 1]; i++) inv[cur[db_words[i]]++] = k;
    const size_t nk = (size_t)(n_kf > 0 ? n_kf : 1);
    int *words = (int *)calloc(nk, sizeof(int));
    uint8_t *listed = (uint8_t *)calloc(nk, 1);      /* mnLoopQuery == pKF->mnId */
    uint8_t *scored = (uint8_t *)calloc(nk, 1);
    float *loop_score = (float *)calloc(nk, sizeof(float));
    int *sharing = (int *)malloc(sizeof(int) * nk);
    int n_sh = 0, n_out = 0;
    for (int i = 0; i < nq; i++)
        for (int j = inv_off[q_words[i]]; j < inv_off[q_words[i] + 1]; j++) { /* :82-102 */
            const int k = inv[j];
            if (connected && connected[k]) continue;
            if (!listed[k]) { listed[k] = 1; words[k] = 0; sharing[n_sh++] = k; }
            words[k]++;
        }
    if (n_sh > 0) {
        int max_common = 0;
        for (int i = 0; i < n_sh; i++) if (words[sharing[i]] > max_common) max_common = words[sharing[i]];
        const int min_common = (int)((float)max_common * 0.8f);
        int *sm_kf = (int *)malloc(sizeof(int) * (size_t)n_sh);
        float *sm_s = (float *)malloc(sizeof(float) * (size_t)n_sh);
        int n_sm = 0;
        for (int i = 0; i < n_sh; i++) { /* :121-137 */
            const int k = sharing[i];
            if (words[k] > min_common) {
                const float si = (float)orc_bow_score(q_words, q_w, nq, db_words + kf_off[k], db_w + kf_off[k], kf_off[k + 1] - kf_off[k]);
                loop_score[k] = si; scored[k] = 1;
                if (si >= min_score) { sm_kf[n_sm] = k; sm_s[n_sm] = si; n_sm++; }
            }
        }
        if (n_sm > 0) {
            float *acc = (float *)malloc(sizeof(float) * (size_t)n_sm);
            int *best_kf = (int *)malloc(sizeof(int) * (size_t)n_sm);
            float best_acc = min_score;
            for (int i = 0; i < n_sm; i++) { /* :146-171 */
                const int k = sm_kf[i];
                float best = sm_s[i], a = sm_s[i];
                int bk = k;
                int nn = covis_off[k + 1] - covis_off[k];
                if (nn > 10) nn = 10;
                for (int j = 0; j < nn; j++) {
                    const int k2 = covis_idx[covis_off[k] + j];
                    if (!scored[k2]) continue;
                    a += loop_score[k2];
                    if (loop_score[k2] > best) { bk = k2; best = loop_score[k2]; }
                }
                acc[i] = a; best_kf[i] = bk;
                if (a > best_acc) best_acc = a;
            }
            const float min_retain = 0.75f * best_acc;
            uint8_t *added = (uint8_t *)calloc(nk, 1);
            for (int i = 0; i < n_sm; i++)
                if (acc[i] > min_retain && !added[best_kf[i]]) {
                    added[best_kf[i]] = 1;
                    if (n_out < cap) cand[n_out] = best_kf[i];
                    n_out++;
                }
            free(added); free(acc); free(best_kf);
        }
        free(sm_kf); free(sm_s);
    }
    free(inv_off); free(inv); free(cur); free(words); free(listed); free(scored); free(loop_score); free(sharing);
    return n_out;
}
