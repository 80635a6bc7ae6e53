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
