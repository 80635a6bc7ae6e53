// orbfe_bow.hip -- fbow vocabulary transform (Frame::ComputeFboW, src/Frame.cc:395-400 ->
// fbow::Vocabulary::_transform2<L1_32bytes>, Thirdparty/fbow/src/fbow.h:400-444) and
// ORBmatcher::SearchByFboW(KeyFrame*, Frame&) (src/ORBmatcher.cc:157-283), SURVEY.md §8a row 17.
//
// The vocabulary blob (fbow file format, fbow.cpp:172-191) is uploaded once and stays resident in HBM
// (k=10, L=6: ~44 MB; L2 / Infinity-Cache resident).  bow_descend_kernel: one thread per descriptor walks
// the k-ary tree by minimum Hamming distance (first minimum wins) and returns (word id, weight, node id
// at the store level); the two std::map results are rebuilt on the host, weights summed in feature order
// so the float sums equal the reference's.  SearchByFboW: the Hamming distances of all (KF feature, frame
// feature) pairs that share a vocabulary node are computed on the GPU (one thread per pair); the greedy
// per-node resolution runs in order on the host, like the other matchers (orbfe_match.hip).
#include "../../include/orbfe.h"
#include "orbfe_device.h"
#include "orbfe_host.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <new>
#include <vector>

#define HISTO_LENGTH 30
#define TH_LOW 50
#define BOW_MAX_DEPTH 32 // node ids are 32-bit paths of ceil(log2 k) bits per level: deeper trees cannot be addressed anyway

struct FbowParams { // fbow::Vocabulary::params, Thirdparty/fbow/src/fbow.h:118-129
    char desc_name[50];
    uint32_t aligment, nblocks;
    uint64_t desc_size_bytes_wp, block_size_bytes_wp, feature_off_start, child_off_start, total_size;
    int32_t desc_type, desc_size;
    uint32_t m_k;
};
static_assert(sizeof(FbowParams) == 120, "fbow params layout");

struct orbfe_bow_state {
    FbowParams p;
    uint8_t *d_data = nullptr;
    bool loaded = false;
    void *d_scratch = nullptr;
    size_t scratch_bytes = 0;
    // keyframe database (KeyFrameDatabase): BoW vectors of the keyframes, resident in HBM, CSR by keyframe
    uint32_t *d_db_words = nullptr;
    float *d_db_w = nullptr;
    size_t db_cap = 0;                 // entries allocated
    std::vector<int> db_off, db_len;   // per keyframe: first entry, number of words (0 after erase)
    std::vector<uint8_t> db_dead;      // erased keyframes: never scored again, their words are reclaimed by kfdb_compact
    size_t db_used = 0;
    size_t db_dead_words = 0;          // words of erased keyframes still occupying the CSR
    ~orbfe_bow_state()
    {
        if (d_data) hipFree(d_data);
        if (d_scratch) hipFree(d_scratch);
        if (d_db_words) hipFree(d_db_words);
        if (d_db_w) hipFree(d_db_w);
    }
};

orbfe_bow_state *orbfe_bow_state_create() { return new (std::nothrow) orbfe_bow_state(); }
void orbfe_bow_state_destroy(orbfe_bow_state *s) { delete s; }

#define BTRY(ctx, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return orbfe_fail(ctx, ORBFE_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

static int ensure_scratch(orbfe_context *ctx, orbfe_bow_state *st, size_t need)
{
    if (need <= st->scratch_bytes) return ORBFE_OK;
    if (st->d_scratch) hipFree(st->d_scratch);
    st->d_scratch = nullptr; st->scratch_bytes = 0;
    BTRY(ctx, hipMalloc(&st->d_scratch, need));
    st->scratch_bytes = need;
    return ORBFE_OK;
}

// one thread per descriptor: descend by minimum Hamming distance
__global__ __launch_bounds__(256) void bow_descend_kernel(const uint8_t *__restrict__ data, unsigned block_size, unsigned feat_off,
                                                          unsigned child_off, unsigned desc_wp, int nbits, int store_level,
                                                          const uint8_t *__restrict__ desc, int n,
                                                          uint32_t *__restrict__ word_id, float *__restrict__ weight, uint32_t *__restrict__ node_id)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= n) return;
    unsigned long long feat[4];
    {
        const unsigned long long *p = (const unsigned long long *)(desc + (size_t)32 * f);
#pragma unroll
        for (int i = 0; i < 4; i++) feat[i] = p[i];
    }
    const uint8_t *blk = data;
    uint32_t level = 0, cur_node = 0, nid = 0, wid = 0;
    float w = 0.f;
    for (;;) {
        const int N = *(const uint16_t *)blk;
        unsigned best_d = 0xffffffffu, best_i = 0;
        for (int c = 0; c < N; c++) {
            const unsigned long long *nf = (const unsigned long long *)(blk + feat_off + (size_t)c * desc_wp);
            const unsigned d = __popcll(nf[0] ^ feat[0]) + __popcll(nf[1] ^ feat[1]) + __popcll(nf[2] ^ feat[2]) + __popcll(nf[3] ^ feat[3]);
            if (d < best_d) { best_d = d; best_i = (unsigned)c; }
        }
        if (level == (uint32_t)store_level) nid = cur_node;
        const uint32_t id_or_child = *(const uint32_t *)(blk + child_off + (size_t)best_i * 8);
        if (id_or_child & 0x80000000u) {
            wid = id_or_child & 0x7fffffffu;
            w = *(const float *)(blk + child_off + (size_t)best_i * 8 + 4);
            if (level < (uint32_t)store_level) nid = cur_node;
            break;
        }
        const uint32_t child = id_or_child & 0x7fffffffu;
        blk = data + (size_t)child * block_size;
        cur_node = (cur_node << nbits) | best_i;
        level++;
        if (child == 0 || level > BOW_MAX_DEPTH) break; // orbfe_vocab_load rejects deeper / cyclic trees; never spin on a bad blob
    }
    word_id[f] = wid; weight[f] = w; node_id[f] = nid;
}

__global__ __launch_bounds__(256) void pair_hamming_kernel(const uint8_t *__restrict__ da, const uint8_t *__restrict__ db,
                                                           const uint32_t *__restrict__ pa, const uint32_t *__restrict__ pb, int npairs,
                                                           uint16_t *__restrict__ dist)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    const uint32_t *a = (const uint32_t *)(da + (size_t)32 * pa[i]);
    const uint32_t *b = (const uint32_t *)(db + (size_t)32 * pb[i]);
    int d = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) d += __popc(a[k] ^ b[k]);
    dist[i] = (uint16_t)d;
}

// fbow::Vocabulary::fromStream (fbow.cpp:181-191) from a memory blob; the tree goes to HBM.
extern "C" int orbfe_vocab_load(orbfe_context *ctx, const uint8_t *blob, size_t size)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !blob) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    orbfe_bow_state *st = orbfe_ctx_bow_state(ctx);
    if (!st) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "out of host memory");
    if (size < 8 + sizeof(FbowParams)) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "vocabulary blob too small");
    uint64_t sig;
    memcpy(&sig, blob, 8);
    if (sig != 55824124ull) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "Vocabulary::fromStream invalid signature");
    FbowParams p;
    memcpy(&p, blob + 8, sizeof(p));
    if (p.desc_size != 32) return orbfe_fail(ctx, ORBFE_ERR_UNSUPPORTED, "only 32-byte (ORB) vocabularies are supported");
    if (p.m_k == 0 || p.nblocks == 0 || p.total_size != p.block_size_bytes_wp * (uint64_t)p.nblocks || size < 8 + sizeof(p) + p.total_size ||
        p.child_off_start + (uint64_t)p.m_k * 8 > p.block_size_bytes_wp || p.feature_off_start + (uint64_t)p.m_k * p.desc_size_bytes_wp > p.child_off_start ||
        (p.desc_size_bytes_wp % 8) || (p.feature_off_start % 8) || (p.block_size_bytes_wp % 8))
        return orbfe_fail(ctx, ORBFE_ERR_INVALID, "inconsistent vocabulary header");
    // validate child links once so the kernel cannot leave the blob
    const uint8_t *data = blob + 8 + sizeof(p);
    for (uint32_t b = 0; b < p.nblocks; b++) {
        const uint8_t *blk = data + (size_t)b * p.block_size_bytes_wp;
        const unsigned N = *(const uint16_t *)blk;
        if (N > p.m_k) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "block %u holds %u > k nodes", b, N);
        for (unsigned c = 0; c < N; c++) {
            uint32_t ic;
            memcpy(&ic, blk + p.child_off_start + (size_t)c * 8, 4);
            if (!(ic & 0x80000000u) && ic >= p.nblocks) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "block %u links to block %u", b, ic);
        }
    }
    {   // the links must form a tree of bounded depth below block 0: a cycle (or a block reached twice) would make the
        // descent kernel loop, so walk every link once and refuse anything else
        std::vector<uint8_t> depth(p.nblocks, 0); // 0 = not reached
        std::vector<uint32_t> stack(1, 0u);
        depth[0] = 1;
        while (!stack.empty()) {
            const uint32_t b = stack.back();
            stack.pop_back();
            const uint8_t *blk = data + (size_t)b * p.block_size_bytes_wp;
            const unsigned N = *(const uint16_t *)blk;
            for (unsigned c = 0; c < N; c++) {
                uint32_t ic;
                memcpy(&ic, blk + p.child_off_start + (size_t)c * 8, 4);
                if (ic & 0x80000000u) continue; // leaf: word id
                if (ic == 0) continue;          // the descent stops at a zero link
                if (depth[ic]) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "block %u is linked twice (cycle or shared subtree)", ic);
                if (depth[b] >= BOW_MAX_DEPTH) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "vocabulary deeper than %d levels", BOW_MAX_DEPTH);
                depth[ic] = (uint8_t)(depth[b] + 1);
                stack.push_back(ic);
            }
        }
    }
    BTRY(ctx, hipSetDevice(orbfe_ctx_device(ctx)));
    if (st->d_data) { hipFree(st->d_data); st->d_data = nullptr; }
    st->loaded = false;
    BTRY(ctx, hipMalloc((void **)&st->d_data, p.total_size));
    BTRY(ctx, hipMemcpy(st->d_data, data, p.total_size, hipMemcpyHostToDevice));
    st->p = p;
    st->loaded = true;
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// Vocabulary::transform(features, level, fBow, fBow2): per-feature results (word, weight, node at `level`).
// size in bytes of the vocabulary image this context holds (0: none): lets a shim that caches "already loaded" per context pointer
// notice a NEW context at a recycled address
extern "C" long long orbfe_vocab_bytes(orbfe_context *ctx)
{
    ORBFE_ENTRY(ctx);
    if (!ctx) return 0;
    orbfe_bow_state *st = orbfe_ctx_bow_state(ctx);
    return st && st->loaded ? (long long)st->p.total_size : 0;
}

extern "C" int orbfe_bow_transform(orbfe_context *ctx, const uint8_t *desc, int n, int level,
                                   uint32_t *word_id, float *weight, uint32_t *node_id)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || n < 0 || level < 0 || (n > 0 && (!desc || !word_id || !weight || !node_id))) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "bad argument");
    orbfe_bow_state *st = orbfe_ctx_bow_state(ctx);
    if (!st || !st->loaded) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "no vocabulary loaded (orbfe_vocab_load)");
    if (n == 0) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "Vocabulary::transform No input data"); // fbow.cpp:52
    hipStream_t s = orbfe_ctx_stream(ctx);
    BTRY(ctx, hipSetDevice(orbfe_ctx_device(ctx)));
    int rc = ensure_scratch(ctx, st, (size_t)n * (32 + 12));
    if (rc != ORBFE_OK) return rc;
    uint8_t *d_desc = (uint8_t *)st->d_scratch;
    uint32_t *d_word = (uint32_t *)(d_desc + (size_t)32 * n);
    float *d_w = (float *)(d_word + n);
    uint32_t *d_node = (uint32_t *)(d_w + n);
    BTRY(ctx, hipMemcpyAsync(d_desc, desc, (size_t)32 * n, hipMemcpyHostToDevice, s));
    const int nbits = (int)ceil(log2((double)st->p.m_k));
    hipLaunchKernelGGL(bow_descend_kernel, dim3((n + 255) / 256), dim3(256), 0, s, st->d_data, (unsigned)st->p.block_size_bytes_wp,
                       (unsigned)st->p.feature_off_start, (unsigned)st->p.child_off_start, (unsigned)st->p.desc_size_bytes_wp, nbits, level,
                       d_desc, n, d_word, d_w, d_node);
    BTRY(ctx, hipMemcpyAsync(word_id, d_word, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, s));
    BTRY(ctx, hipMemcpyAsync(weight, d_w, sizeof(float) * n, hipMemcpyDeviceToHost, s));
    BTRY(ctx, hipMemcpyAsync(node_id, d_node, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, s));
    BTRY(ctx, hipStreamSynchronize(s));
    BTRY(ctx, hipGetLastError());
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// fBow / fBow2 as sorted arrays from the per-feature results (host; weights summed in feature order).
extern "C" int orbfe_bow_maps(const uint32_t *word_id, const float *weight, const uint32_t *node_id, int n,
                              uint32_t *words, float *word_w, int *n_words,
                              uint32_t *nodes, int32_t *node_off, int32_t *node_feat, int *n_nodes)
try {
    if (n < 0 || !n_words || !n_nodes || (n > 0 && (!word_id || !weight || !node_id || !words || !word_w || !nodes || !node_off || !node_feat)))
        return ORBFE_ERR_INVALID;
    std::map<uint32_t, float> r1;
    std::map<uint32_t, std::vector<int32_t>> r2;
    for (int f = 0; f < n; f++) {
        r1[word_id[f]] += weight[f]; // fbow.h:431, _float default 0
        r2[node_id[f]].push_back(f);
    }
    int i = 0;
    for (auto &kv : r1) { words[i] = kv.first; word_w[i] = kv.second; i++; }
    *n_words = i;
    int k = 0, o = 0;
    for (auto &kv : r2) {
        nodes[k] = kv.first; node_off[k] = o;
        for (int32_t f : kv.second) node_feat[o++] = f;
        k++;
    }
    if (n > 0) node_off[k] = o;
    *n_nodes = k;
    return ORBFE_OK;
} ORBFE_CATCH(nullptr)

static int rot_bin(float a1, float a2)
{
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

// ORBmatcher::SearchByFboW(KeyFrame*, Frame&, vpMapPointMatches), src/ORBmatcher.cc:157-283
// Shared body of the two SearchByFboW overloads.  kf_kf = false: (KeyFrame, Frame), src/ORBmatcher.cc:157-283, result
// indexed by the frame keypoint; kf_kf = true: (KeyFrame, KeyFrame), :517-650, the second side needs f_valid, takes each of
// its keypoints at most once, the distance bound is strict and the result is indexed by the first keyframe's keypoint.
static int search_by_bow_impl(orbfe_context *ctx, bool kf_kf,
                              const uint32_t *kf_nodes, const int32_t *kf_off, const int32_t *kf_feat, int kf_nnodes,
                              const int32_t *kf_valid, const uint8_t *kf_desc, const float *kf_angle, int n_kf,
                              const uint32_t *f_nodes, const int32_t *f_off, const int32_t *f_feat, int f_nnodes,
                              const int32_t *f_valid, const uint8_t *f_desc, const float *f_angle, int n_f,
                              float nnratio, int check_ori, int32_t *match, int *nmatches)
{
    const int n_out = kf_kf ? n_kf : n_f;
    if (!ctx || !nmatches || n_kf < 0 || n_f < 0 || kf_nnodes < 0 || f_nnodes < 0 || (n_out > 0 && !match) || (kf_kf && n_f > 0 && !f_valid) ||
        (kf_nnodes > 0 && (!kf_nodes || !kf_off || !kf_feat || !kf_valid || !kf_desc || !kf_angle)) ||
        (f_nnodes > 0 && (!f_nodes || !f_off || !f_feat || !f_desc || !f_angle)))
        return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    orbfe_bow_state *st = orbfe_ctx_bow_state(ctx);
    if (!st) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "out of host memory");
    for (int j = 0; j < n_out; j++) match[j] = -1;
    std::vector<uint8_t> taken(kf_kf ? (size_t)(n_f > 0 ? n_f : 1) : 1, 0); // vbMatched2
    *nmatches = 0;
    // merge-join of the two feature vectors: enumerate every (KF feature, frame feature) pair of a shared node
    struct Seg { int a, b, pair0; };
    std::vector<Seg> segs;
    std::vector<uint32_t> pa, pb;
    for (int a = 0, b = 0; a < kf_nnodes && b < f_nnodes;) {
        if (kf_nodes[a] == f_nodes[b]) {
            segs.push_back(Seg{a, b, (int)pa.size()});
            for (int ik = kf_off[a]; ik < kf_off[a + 1]; ik++) {
                if (kf_feat[ik] < 0 || kf_feat[ik] >= n_kf) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "KF feature index out of range");
                for (int jf = f_off[b]; jf < f_off[b + 1]; jf++) {
                    if (f_feat[jf] < 0 || f_feat[jf] >= n_f) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "frame feature index out of range");
                    pa.push_back((uint32_t)kf_feat[ik]); pb.push_back((uint32_t)f_feat[jf]);
                }
            }
            a++; b++;
        } else if (kf_nodes[a] < f_nodes[b]) a++;
        else b++;
    }
    const int np = (int)pa.size();
    std::vector<uint16_t> dist(np > 0 ? np : 1);
    if (np > 0) {
        hipStream_t s = orbfe_ctx_stream(ctx);
        BTRY(ctx, hipSetDevice(orbfe_ctx_device(ctx)));
        const size_t need = (size_t)32 * n_kf + (size_t)32 * n_f + (size_t)np * (4 + 4 + 2) + 64;
        int rc = ensure_scratch(ctx, st, need);
        if (rc != ORBFE_OK) return rc;
        uint8_t *d_a = (uint8_t *)st->d_scratch, *d_b = d_a + (size_t)32 * n_kf;
        uint32_t *d_pa = (uint32_t *)(d_b + (size_t)32 * n_f), *d_pb = d_pa + np;
        uint16_t *d_dist = (uint16_t *)(d_pb + np);
        BTRY(ctx, hipMemcpyAsync(d_a, kf_desc, (size_t)32 * n_kf, hipMemcpyHostToDevice, s));
        BTRY(ctx, hipMemcpyAsync(d_b, f_desc, (size_t)32 * n_f, hipMemcpyHostToDevice, s));
        BTRY(ctx, hipMemcpyAsync(d_pa, pa.data(), sizeof(uint32_t) * np, hipMemcpyHostToDevice, s));
        BTRY(ctx, hipMemcpyAsync(d_pb, pb.data(), sizeof(uint32_t) * np, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(pair_hamming_kernel, dim3((np + 255) / 256), dim3(256), 0, s, d_a, d_b, d_pa, d_pb, np, d_dist);
        BTRY(ctx, hipMemcpyAsync(dist.data(), d_dist, sizeof(uint16_t) * np, hipMemcpyDeviceToHost, s));
        BTRY(ctx, hipStreamSynchronize(s));
        BTRY(ctx, hipGetLastError());
    }
    // greedy resolve in the reference's order
    std::vector<int> hist[HISTO_LENGTH];
    int nm = 0;
    for (const Seg &sg : segs) {
        const int nf = f_off[sg.b + 1] - f_off[sg.b];
        int pi = sg.pair0;
        for (int ik = kf_off[sg.a]; ik < kf_off[sg.a + 1]; ik++, pi += nf) {
            const int real_kf = kf_feat[ik];
            if (!kf_valid[real_kf]) continue;
            int best1 = 256, best_f = -1, best2 = 256;
            for (int j = 0; j < nf; j++) {
                const int real_f = f_feat[f_off[sg.b] + j];
                if (kf_kf ? (taken[real_f] || !f_valid[real_f]) : (match[real_f] >= 0)) continue;
                const int d = dist[pi + j];
                if (d < best1) { best2 = best1; best1 = d; best_f = real_f; }
                else if (d < best2) best2 = d;
            }
            if ((kf_kf ? best1 < TH_LOW : best1 <= TH_LOW) && (float)best1 < nnratio * (float)best2) {
                if (kf_kf) { match[real_kf] = best_f; taken[best_f] = 1; }
                else match[best_f] = real_kf;
                if (check_ori) hist[rot_bin(kf_angle[real_kf], f_angle[best_f])].push_back(kf_kf ? real_kf : best_f);
                nm++;
            }
        }
    }
    if (check_ori) {
        int32_t sizes[HISTO_LENGTH];
        for (int b = 0; b < HISTO_LENGTH; b++) sizes[b] = (int32_t)hist[b].size();
        int i1, i2, i3;
        orbfe_three_maxima(sizes, HISTO_LENGTH, &i1, &i2, &i3);
        for (int b = 0; b < HISTO_LENGTH; b++) {
            if (b == i1 || b == i2 || b == i3) continue;
            for (int idx : hist[b]) { match[idx] = -1; nm--; }
        }
    }
    *nmatches = nm;
    return ORBFE_OK;
}

extern "C" int orbfe_search_by_bow(orbfe_context *ctx,
                                   const uint32_t *kf_nodes, const int32_t *kf_off, const int32_t *kf_feat, int kf_nnodes,
                                   const int32_t *kf_valid, const uint8_t *kf_desc, const float *kf_angle, int n_kf,
                                   const uint32_t *f_nodes, const int32_t *f_off, const int32_t *f_feat, int f_nnodes,
                                   const uint8_t *f_desc, const float *f_angle, int n_f,
                                   float nnratio, int check_ori, int32_t *f_match, int *nmatches)
try {
    ORBFE_ENTRY(ctx);
    return search_by_bow_impl(ctx, false, kf_nodes, kf_off, kf_feat, kf_nnodes, kf_valid, kf_desc, kf_angle, n_kf,
                              f_nodes, f_off, f_feat, f_nnodes, nullptr, f_desc, f_angle, n_f, nnratio, check_ori, f_match, nmatches);
} ORBFE_CATCH(ctx)

extern "C" int orbfe_search_by_bow_kf(orbfe_context *ctx,
                                      const uint32_t *nodes1, const int32_t *off1, const int32_t *feat1, int nnodes1,
                                      const int32_t *valid1, const uint8_t *desc1, const float *angle1, int n1,
                                      const uint32_t *nodes2, const int32_t *off2, const int32_t *feat2, int nnodes2,
                                      const int32_t *valid2, const uint8_t *desc2, const float *angle2, int n2,
                                      float nnratio, int check_ori, int32_t *match12, int *nmatches)
try {
    ORBFE_ENTRY(ctx);
    return search_by_bow_impl(ctx, true, nodes1, off1, feat1, nnodes1, valid1, desc1, angle1, n1,
                              nodes2, off2, feat2, nnodes2, valid2, desc2, angle2, n2, nnratio, check_ori, match12, nmatches);
} ORBFE_CATCH(ctx)

// ---------------------------------------------------------------------------------------------
// Keyframe database: KeyFrameDatabase::add / erase / clear / DetectRelocalizationCandidates
// (src/KeyFrameDatabase.cc:38-70,196-307) with fbow::fBow::score (Thirdparty/fbow/src/fbow.cpp:206-256)
// ---------------------------------------------------------------------------------------------
#define KFDB_Q_LDS 3072 // query words staged in LDS (longer queries are searched in HBM)

// One wave per keyframe: its words are looked up in the query by binary search, 64 at a time; the matching products are
// compacted in word order and added by lane 0 SEQUENTIALLY in double -- the association order of fBow::score, which its
// float result depends on.  Outputs per keyframe: shared-word count, smallest shared word (the inverted file's
// first-encounter order = ascending (first shared word, keyframe index)), score.
__global__ __launch_bounds__(256) void kfdb_score_kernel(const uint32_t *__restrict__ q_words, const float *__restrict__ q_w, int nq,
                                                         const int *__restrict__ kf_off, const int *__restrict__ kf_len,
                                                         const uint32_t *__restrict__ db_words, const float *__restrict__ db_w, int n_kf, // n_kf = LIVE keyframes

                                                         int *__restrict__ common, uint32_t *__restrict__ first_word, float *__restrict__ score)
{
    __shared__ uint32_t s_qw[KFDB_Q_LDS];
    __shared__ float s_qv[KFDB_Q_LDS];
    __shared__ float s_prod[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool in_lds = nq <= KFDB_Q_LDS;
    if (in_lds)
        for (int i = threadIdx.x; i < nq; i += 256) { s_qw[i] = q_words[i]; s_qv[i] = q_w[i]; }
    __syncthreads();
    const int kf = blockIdx.x * 4 + wave;
    if (kf >= n_kf) return;
    const uint32_t *qw = in_lds ? s_qw : q_words;
    const float *qv = in_lds ? s_qv : q_w;
    const int o = kf_off[kf], len = kf_len[kf];
    double sum = 0.0;
    int ncommon = 0;
    uint32_t first = 0xffffffffu;
    for (int i0 = 0; i0 < len; i0 += 64) {
        const int i = i0 + lane;
        bool hit = false;
        float prod = 0.f;
        uint32_t w = 0;
        if (i < len) {
            w = db_words[o + i];
            int lo = 0, hi = nq; // first query word >= w
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (qw[mid] < w) lo = mid + 1; else hi = mid;
            }
            if (lo < nq && qw[lo] == w) { hit = true; prod = __fmul_rn(qv[lo], db_w[o + i]); }
        }
        const unsigned long long m = __ballot(hit);
        if (m) {
            if (hit) s_prod[wave][__popcll(m & ((1ull << lane) - 1ull))] = prod;
            if (first == 0xffffffffu) first = (uint32_t)__shfl((int)w, __ffsll((long long)m) - 1, 64);
            const int c = __popcll(m);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0)
                for (int j = 0; j < c; j++) sum = __dadd_rn(sum, (double)s_prod[wave][j]);
            __builtin_amdgcn_wave_barrier();
            ncommon += c;
        }
    }
    if (lane == 0) {
        common[kf] = ncommon;
        first_word[kf] = first;
        const double sc = sum >= 1.0 ? 1.0 : 1.0 - sqrt(1.0 - sum);
        score[kf] = (float)sc;
    }
}

extern "C" int orbfe_kfdb_clear(orbfe_context *ctx)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx) return ORBFE_ERR_INVALID;
    orbfe_bow_state *st = orbfe_ctx_bow_state(ctx);
    if (!st) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "out of host memory");
    st->db_off.clear(); st->db_len.clear(); st->db_dead.clear(); st->db_used = 0; st->db_dead_words = 0;
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

extern "C" int orbfe_kfdb_add(orbfe_context *ctx, const uint32_t *words, const float *weights, int n, int *kf_index)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || n < 0 || (n > 0 && (!words || !weights))) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    orbfe_bow_state *st = orbfe_ctx_bow_state(ctx);
    if (!st) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "out of host memory");
    for (int i = 1; i < n; i++)
        if (words[i] <= words[i - 1]) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "BoW words must be strictly ascending (std::map order)");
    BTRY(ctx, hipSetDevice(orbfe_ctx_device(ctx)));
    hipStream_t s = orbfe_ctx_stream(ctx);
    if (st->db_used + (size_t)n > st->db_cap) { // grow: twice the need, contents moved device to device
        const size_t cap = 2 * (st->db_used + (size_t)n) + 1024;
        uint32_t *nw = nullptr; float *nv = nullptr;
        BTRY(ctx, hipMalloc((void **)&nw, cap * sizeof(uint32_t)));
        if (hipMalloc((void **)&nv, cap * sizeof(float)) != hipSuccess) { hipFree(nw); return orbfe_fail(ctx, ORBFE_ERR_HIP, "hipMalloc failed"); }
        if (st->db_used) {
            hipMemcpyAsync(nw, st->d_db_words, st->db_used * sizeof(uint32_t), hipMemcpyDeviceToDevice, s);
            hipMemcpyAsync(nv, st->d_db_w, st->db_used * sizeof(float), hipMemcpyDeviceToDevice, s);
        }
        BTRY(ctx, hipStreamSynchronize(s));
        if (st->d_db_words) hipFree(st->d_db_words);
        if (st->d_db_w) hipFree(st->d_db_w);
        st->d_db_words = nw; st->d_db_w = nv; st->db_cap = cap;
    }
    if (n > 0) {
        BTRY(ctx, hipMemcpyAsync(st->d_db_words + st->db_used, words, sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
        BTRY(ctx, hipMemcpyAsync(st->d_db_w + st->db_used, weights, sizeof(float) * n, hipMemcpyHostToDevice, s));
        BTRY(ctx, hipStreamSynchronize(s));
    }
    if (kf_index) *kf_index = (int)st->db_off.size();
    st->db_off.push_back((int)st->db_used); st->db_len.push_back(n); st->db_dead.push_back(0);
    st->db_used += (size_t)n;
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// Reclaims the words of erased keyframes: the live segments are moved to the front of the CSR (keyframe indices stay).
// Rare (once the dead words outnumber the live ones), so it goes through the host.
static int kfdb_compact(orbfe_context *ctx, orbfe_bow_state *st)
{
    BTRY(ctx, hipSetDevice(orbfe_ctx_device(ctx)));
    hipStream_t s = orbfe_ctx_stream(ctx);
    std::vector<uint32_t> w(st->db_used ? st->db_used : 1);
    std::vector<float> v(st->db_used ? st->db_used : 1);
    if (st->db_used) {
        BTRY(ctx, hipMemcpyAsync(w.data(), st->d_db_words, st->db_used * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        BTRY(ctx, hipMemcpyAsync(v.data(), st->d_db_w, st->db_used * sizeof(float), hipMemcpyDeviceToHost, s));
        BTRY(ctx, hipStreamSynchronize(s));
    }
    size_t o = 0;
    for (size_t k = 0; k < st->db_off.size(); k++) {
        if (st->db_dead[k]) { st->db_off[k] = 0; continue; }
        const size_t from = (size_t)st->db_off[k], len = (size_t)st->db_len[k];
        if (from != o) { memmove(&w[o], &w[from], len * sizeof(uint32_t)); memmove(&v[o], &v[from], len * sizeof(float)); }
        st->db_off[k] = (int)o;
        o += len;
    }
    if (o) {
        BTRY(ctx, hipMemcpyAsync(st->d_db_words, w.data(), o * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        BTRY(ctx, hipMemcpyAsync(st->d_db_w, v.data(), o * sizeof(float), hipMemcpyHostToDevice, s));
        BTRY(ctx, hipStreamSynchronize(s));
    }
    st->db_used = o;
    st->db_dead_words = 0;
    return ORBFE_OK;
}

extern "C" int orbfe_kfdb_erase(orbfe_context *ctx, int kf_index)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx) return ORBFE_ERR_INVALID;
    orbfe_bow_state *st = orbfe_ctx_bow_state(ctx);
    if (!st || kf_index < 0 || kf_index >= (int)st->db_len.size()) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "keyframe index out of range");
    if (st->db_dead[kf_index]) return ORBFE_OK;
    // KeyFrameDatabase::erase: the keyframe leaves every inverted-file list.  It is dropped from the launch list of every
    // later query, and its words are reclaimed once the dead ones outnumber the live ones (culling in a long session must
    // not grow HBM use or query cost with the number of keyframes ever added).
    st->db_dead[kf_index] = 1;
    st->db_dead_words += (size_t)st->db_len[kf_index];
    st->db_len[kf_index] = 0;
    if (st->db_dead_words > 1024 && 2 * st->db_dead_words > st->db_used) return kfdb_compact(ctx, st);
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

extern "C" int orbfe_kfdb_size(orbfe_context *ctx)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx) return 0;
    orbfe_bow_state *st = orbfe_ctx_bow_state(ctx);
    return st ? (int)st->db_off.size() : 0;
} ORBFE_CATCH(ctx)

// scores of every keyframe against the query; host vectors out
static int kfdb_scores(orbfe_context *ctx, orbfe_bow_state *st, const uint32_t *q_words, const float *q_w, int nq,
                       std::vector<int> &common, std::vector<uint32_t> &first, std::vector<float> &score)
{
    const int n_kf = (int)st->db_off.size();
    common.assign(n_kf > 0 ? n_kf : 1, 0); first.assign(n_kf > 0 ? n_kf : 1, 0xffffffffu); score.assign(n_kf > 0 ? n_kf : 1, 0.f);
    if (n_kf == 0 || nq == 0) return ORBFE_OK;
    // launch list: live keyframes only (an erased one costs neither a wave nor a result slot)
    std::vector<int> live, l_off, l_len;
    for (int k = 0; k < n_kf; k++)
        if (!st->db_dead[k]) { live.push_back(k); l_off.push_back(st->db_off[k]); l_len.push_back(st->db_len[k]); }
    const int n_live = (int)live.size();
    if (n_live == 0) return ORBFE_OK;
    BTRY(ctx, hipSetDevice(orbfe_ctx_device(ctx)));
    hipStream_t s = orbfe_ctx_stream(ctx);
    const size_t need = (size_t)nq * 8 + (size_t)n_live * 20 + 64;
    int rc = ensure_scratch(ctx, st, need);
    if (rc != ORBFE_OK) return rc;
    uint32_t *d_qw = (uint32_t *)st->d_scratch;
    float *d_qv = (float *)(d_qw + nq);
    int *d_off = (int *)(d_qv + nq), *d_len = d_off + n_live, *d_common = d_len + n_live;
    uint32_t *d_first = (uint32_t *)(d_common + n_live);
    float *d_score = (float *)(d_first + n_live);
    std::vector<int> l_common(n_live);
    std::vector<uint32_t> l_first(n_live);
    std::vector<float> l_score(n_live);
    BTRY(ctx, hipMemcpyAsync(d_qw, q_words, sizeof(uint32_t) * nq, hipMemcpyHostToDevice, s));
    BTRY(ctx, hipMemcpyAsync(d_qv, q_w, sizeof(float) * nq, hipMemcpyHostToDevice, s));
    BTRY(ctx, hipMemcpyAsync(d_off, l_off.data(), sizeof(int) * n_live, hipMemcpyHostToDevice, s));
    BTRY(ctx, hipMemcpyAsync(d_len, l_len.data(), sizeof(int) * n_live, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(kfdb_score_kernel, dim3((n_live + 3) / 4), dim3(256), 0, s, d_qw, d_qv, nq, d_off, d_len, st->d_db_words, st->d_db_w, n_live,
                       d_common, d_first, d_score);
    BTRY(ctx, hipMemcpyAsync(l_common.data(), d_common, sizeof(int) * n_live, hipMemcpyDeviceToHost, s));
    BTRY(ctx, hipMemcpyAsync(l_first.data(), d_first, sizeof(uint32_t) * n_live, hipMemcpyDeviceToHost, s));
    BTRY(ctx, hipMemcpyAsync(l_score.data(), d_score, sizeof(float) * n_live, hipMemcpyDeviceToHost, s));
    BTRY(ctx, hipStreamSynchronize(s));
    BTRY(ctx, hipGetLastError());
    for (int i = 0; i < n_live; i++) { common[live[i]] = l_common[i]; first[live[i]] = l_first[i]; score[live[i]] = l_score[i]; }
    return ORBFE_OK;
}

extern "C" int orbfe_kfdb_score(orbfe_context *ctx, const uint32_t *q_words, const float *q_w, int nq, int32_t *common, float *score)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || nq < 0 || (nq > 0 && (!q_words || !q_w))) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    orbfe_bow_state *st = orbfe_ctx_bow_state(ctx);
    if (!st) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "out of host memory");
    std::vector<int> c; std::vector<uint32_t> f; std::vector<float> sc;
    int rc = kfdb_scores(ctx, st, q_words, q_w, nq, c, f, sc);
    if (rc != ORBFE_OK) return rc;
    const int n_kf = (int)st->db_off.size();
    for (int k = 0; k < n_kf; k++) { if (common) common[k] = c[k]; if (score) score[k] = sc[k]; }
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

extern "C" int orbfe_detect_reloc_candidates(orbfe_context *ctx, const uint32_t *q_words, const float *q_w, int nq,
                                             const int32_t *covis_off, const int32_t *covis_idx, float *reloc_score,
                                             int32_t *cand, int cap, int *n_cand)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !n_cand || nq < 0 || (nq > 0 && (!q_words || !q_w)) || !covis_off || !reloc_score || cap < 0 || (cap > 0 && !cand))
        return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    orbfe_bow_state *st = orbfe_ctx_bow_state(ctx);
    if (!st) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "out of host memory");
    *n_cand = 0;
    const int n_kf = (int)st->db_off.size();
    std::vector<int> common; std::vector<uint32_t> first; std::vector<float> score;
    int rc = kfdb_scores(ctx, st, q_words, q_w, nq, common, first, score);
    if (rc != ORBFE_OK) return rc;
    // keyframes sharing a word, in the inverted file's first-encounter order (src/KeyFrameDatabase.cc:205-221)
    std::vector<int> sharing;
    for (int k = 0; k < n_kf; k++)
        if (common[k] > 0) sharing.push_back(k);
    if (sharing.empty()) return ORBFE_OK;
    std::stable_sort(sharing.begin(), sharing.end(), [&](int a, int b) { return first[a] < first[b]; });
    int max_common = 0;
    for (int k : sharing) max_common = common[k] > max_common ? common[k] : max_common;
    const int min_common = (int)((float)max_common * 0.8f);
    std::vector<int> sm_kf;
    for (int k : sharing)
        if (common[k] > min_common) { reloc_score[k] = score[k]; sm_kf.push_back(k); }
    if (sm_kf.empty()) return ORBFE_OK;
    std::vector<float> acc(sm_kf.size());
    std::vector<int> best_kf(sm_kf.size());
    float best_acc = 0.f;
    for (size_t i = 0; i < sm_kf.size(); i++) { // accumulate score by covisibility (:254-277)
        const int k = sm_kf[i];
        float best = score[k], a = best;
        int bk = k;
        int nn = covis_off[k + 1] - covis_off[k];
        if (nn > 10) nn = 10;
        for (int j = 0; j < nn; j++) {
            const int k2 = covis_idx[covis_off[k] + j];
            if (k2 < 0 || k2 >= n_kf) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "covisibility index out of range");
            if (common[k2] <= 0) continue;
            a += reloc_score[k2];
            if (reloc_score[k2] > best) { bk = k2; best = reloc_score[k2]; }
        }
        acc[i] = a; best_kf[i] = bk;
        if (a > best_acc) best_acc = a;
    }
    const float min_retain = 0.75f * best_acc;
    std::vector<uint8_t> added(n_kf, 0);
    int n = 0;
    for (size_t i = 0; i < sm_kf.size(); i++)
        if (acc[i] > min_retain && !added[best_kf[i]]) {
            added[best_kf[i]] = 1;
            if (n < cap) cand[n] = best_kf[i];
            n++;
        }
    *n_cand = n;
    if (n > cap) return orbfe_fail(ctx, ORBFE_ERR_CAPACITY, "caller buffer holds %d candidates, %d found", cap, n);
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// KeyFrameDatabase::DetectLoopCandidates(KeyFrame *pKF, float minScore) (src/KeyFrameDatabase.cc:73-194): same device scoring
// pass as the relocalisation query; the selection differs (connected keyframes excluded, minScore filter, accumulated score
// starts from minScore) and keeps no state between calls.
extern "C" int orbfe_detect_loop_candidates(orbfe_context *ctx, const uint32_t *q_words, const float *q_w, int nq,
                                            const uint8_t *connected, float min_score,
                                            const int32_t *covis_off, const int32_t *covis_idx,
                                            int32_t *cand, int cap, int *n_cand)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !n_cand || nq < 0 || (nq > 0 && (!q_words || !q_w)) || !covis_off || cap < 0 || (cap > 0 && !cand))
        return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    orbfe_bow_state *st = orbfe_ctx_bow_state(ctx);
    if (!st) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "out of host memory");
    *n_cand = 0;
    const int n_kf = (int)st->db_off.size();
    std::vector<int> common; std::vector<uint32_t> first; std::vector<float> score;
    int rc = kfdb_scores(ctx, st, q_words, q_w, nq, common, first, score);
    if (rc != ORBFE_OK) return rc;
    std::vector<int> sharing; // lKFsSharingWords: shares a word, not connected, first-encounter order (:82-102)
    for (int k = 0; k < n_kf; k++)
        if (common[k] > 0 && !(connected && connected[k])) sharing.push_back(k);
    if (sharing.empty()) return ORBFE_OK;
    std::stable_sort(sharing.begin(), sharing.end(), [&](int a, int b) { return first[a] < first[b]; });
    int max_common = 0;
    for (int k : sharing) max_common = common[k] > max_common ? common[k] : max_common;
    const int min_common = (int)((float)max_common * 0.8f);
    std::vector<uint8_t> scored(n_kf, 0);
    std::vector<int> sm_kf;
    for (int k : sharing)
        if (common[k] > min_common) {
            scored[k] = 1;                                  // mLoopScore = si (:131)
            if (score[k] >= min_score) sm_kf.push_back(k);  // :132-133
        }
    if (sm_kf.empty()) return ORBFE_OK;
    std::vector<float> acc(sm_kf.size());
    std::vector<int> best_kf(sm_kf.size());
    float best_acc = min_score;
    for (size_t i = 0; i < sm_kf.size(); i++) { // :146-171
        const int k = sm_kf[i];
        float best = score[k], a = score[k];
        int bk = k;
        int nn = covis_off[k + 1] - covis_off[k];
        if (nn > 10) nn = 10;
        for (int j = 0; j < nn; j++) {
            const int k2 = covis_idx[covis_off[k] + j];
            if (k2 < 0 || k2 >= n_kf) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "covisibility index out of range");
            if (!scored[k2]) continue;
            a += score[k2];
            if (score[k2] > best) { bk = k2; best = score[k2]; }
        }
        acc[i] = a; best_kf[i] = bk;
        if (a > best_acc) best_acc = a;
    }
    const float min_retain = 0.75f * best_acc;
    std::vector<uint8_t> added(n_kf, 0);
    int n = 0;
    for (size_t i = 0; i < sm_kf.size(); i++)
        if (acc[i] > min_retain && !added[best_kf[i]]) {
            added[best_kf[i]] = 1;
            if (n < cap) cand[n] = best_kf[i];
            n++;
        }
    *n_cand = n;
    if (n > cap) return orbfe_fail(ctx, ORBFE_ERR_CAPACITY, "caller buffer holds %d candidates, %d found", cap, n);
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// ---------------------------------------------------------------------------------------------
// ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:652-819; LocalMapping::CreateNewMapPoints)
// ---------------------------------------------------------------------------------------------
// ORBmatcher::CheckDistEpipolarLine (:138-155): float arithmetic left to right, the last comparison in double
static bool check_dist_epipolar_line(float x1, float y1, float x2, float y2, const float *F12, float sigma2_kp2)
{
    const float a = x1 * F12[0] + y1 * F12[3] + F12[6];
    const float b = x1 * F12[1] + y1 * F12[4] + F12[7];
    const float c = x1 * F12[2] + y1 * F12[5] + F12[8];
    const float num = a * x2 + b * y2 + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return (double)dsqr < 3.84 * (double)sigma2_kp2;
}

extern "C" int orbfe_search_for_triangulation(orbfe_context *ctx,
                                              const uint32_t *nodes1, const int32_t *off1, const int32_t *feat1, int nnodes1,
                                              const orbfe_keypoint *keys1, const float *u_right1, const uint8_t *has_mp1, const uint8_t *desc1, int n1,
                                              const uint32_t *nodes2, const int32_t *off2, const int32_t *feat2, int nnodes2,
                                              const orbfe_keypoint *keys2, const float *u_right2, const uint8_t *has_mp2, const uint8_t *desc2, int n2,
                                              const float *F12, const float *Cw1, const float *T2w, float fx2, float fy2, float cx2, float cy2,
                                              int only_stereo, int check_ori, int32_t *match12, int *nmatches)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !nmatches || n1 < 0 || n2 < 0 || nnodes1 < 0 || nnodes2 < 0 || (n1 > 0 && !match12) || !F12 || !Cw1 || !T2w ||
        (nnodes1 > 0 && (!nodes1 || !off1 || !feat1 || !keys1 || !u_right1 || !has_mp1 || !desc1)) ||
        (nnodes2 > 0 && (!nodes2 || !off2 || !feat2 || !keys2 || !u_right2 || !has_mp2 || !desc2)))
        return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    orbfe_bow_state *st = orbfe_ctx_bow_state(ctx);
    if (!st) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "out of host memory");
    const int nlevels = orbfe_ctx_params(ctx)->nlevels;
    const float *scale = orbfe_ctx_scale_factors(ctx);
    for (int j = 0; j < n1; j++) match12[j] = -1;
    *nmatches = 0;
    // epipole in the second image (:658-664): cv::Mat R*x+t in float, small-matrix evaluation order
    float C2[3];
    for (int i = 0; i < 3; i++) {
        const float t = (T2w[4 * i] * Cw1[0] + T2w[4 * i + 1] * Cw1[1]) + T2w[4 * i + 2] * Cw1[2];
        C2[i] = t + T2w[4 * i + 3];
    }
    const float invz = 1.0f / C2[2];
    const float ex = fx2 * C2[0] * invz + cx2;
    const float ey = fy2 * C2[1] * invz + cy2;
    // pairs of the shared nodes whose endpoints pass the per-keypoint filters; Hamming distances on the device
    struct Seg { int a, b, pair0; };
    std::vector<Seg> segs;
    std::vector<uint32_t> pa, pb;
    auto usable1 = [&](int i) { return !has_mp1[i] && !(only_stereo && !(u_right1[i] >= 0)); };
    auto usable2 = [&](int i) { return !has_mp2[i] && !(only_stereo && !(u_right2[i] >= 0)); };
    for (int a = 0, b = 0; a < nnodes1 && b < nnodes2;) {
        if (nodes1[a] == nodes2[b]) {
            segs.push_back(Seg{a, b, (int)pa.size()});
            for (int i1 = off1[a]; i1 < off1[a + 1]; i1++) {
                if (feat1[i1] < 0 || feat1[i1] >= n1) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "KF1 feature index out of range");
                if (!usable1(feat1[i1])) continue;
                for (int i2 = off2[b]; i2 < off2[b + 1]; i2++) {
                    if (feat2[i2] < 0 || feat2[i2] >= n2) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "KF2 feature index out of range");
                    if (keys2[feat2[i2]].octave < 0 || keys2[feat2[i2]].octave >= nlevels) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "octave out of range");
                    if (!usable2(feat2[i2])) continue;
                    pa.push_back((uint32_t)feat1[i1]); pb.push_back((uint32_t)feat2[i2]);
                }
            }
            a++; b++;
        } else if (nodes1[a] < nodes2[b]) a++;
        else b++;
    }
    const int np = (int)pa.size();
    std::vector<uint16_t> dist(np > 0 ? np : 1);
    if (np > 0) {
        hipStream_t s = orbfe_ctx_stream(ctx);
        BTRY(ctx, hipSetDevice(orbfe_ctx_device(ctx)));
        const size_t need = (size_t)32 * n1 + (size_t)32 * n2 + (size_t)np * (4 + 4 + 2) + 64;
        int rc = ensure_scratch(ctx, st, need);
        if (rc != ORBFE_OK) return rc;
        uint8_t *d_a = (uint8_t *)st->d_scratch, *d_b = d_a + (size_t)32 * n1;
        uint32_t *d_pa = (uint32_t *)(d_b + (size_t)32 * n2), *d_pb = d_pa + np;
        uint16_t *d_dist = (uint16_t *)(d_pb + np);
        BTRY(ctx, hipMemcpyAsync(d_a, desc1, (size_t)32 * n1, hipMemcpyHostToDevice, s));
        BTRY(ctx, hipMemcpyAsync(d_b, desc2, (size_t)32 * n2, hipMemcpyHostToDevice, s));
        BTRY(ctx, hipMemcpyAsync(d_pa, pa.data(), sizeof(uint32_t) * np, hipMemcpyHostToDevice, s));
        BTRY(ctx, hipMemcpyAsync(d_pb, pb.data(), sizeof(uint32_t) * np, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(pair_hamming_kernel, dim3((np + 255) / 256), dim3(256), 0, s, d_a, d_b, d_pa, d_pb, np, d_dist);
        BTRY(ctx, hipMemcpyAsync(dist.data(), d_dist, sizeof(uint16_t) * np, hipMemcpyDeviceToHost, s));
        BTRY(ctx, hipStreamSynchronize(s));
        BTRY(ctx, hipGetLastError());
    }
    // sequential resolve in the reference's order (vbMatched2 makes it order dependent)
    std::vector<uint8_t> matched2(n2 > 0 ? n2 : 1, 0);
    std::vector<int> hist[HISTO_LENGTH];
    std::vector<float> sigma2(nlevels);
    for (int l = 0; l < nlevels; l++) sigma2[l] = scale[l] * scale[l]; // mvLevelSigma2 (src/ORBextractor.cc:419-423)
    int nm = 0;
    for (const Seg &sg : segs) {
        int pi = sg.pair0;
        for (int i1 = off1[sg.a]; i1 < off1[sg.a + 1]; i1++) {
            const int idx1 = feat1[i1];
            if (!usable1(idx1)) continue;
            const bool stereo1 = u_right1[idx1] >= 0;
            int best_dist = TH_LOW, best_idx2 = -1;
            for (int i2 = off2[sg.b]; i2 < off2[sg.b + 1]; i2++) {
                const int idx2 = feat2[i2];
                if (!usable2(idx2)) continue;
                const int d = dist[pi++];
                if (matched2[idx2]) continue;
                if (d > TH_LOW || d > best_dist) continue;
                const bool stereo2 = u_right2[idx2] >= 0;
                if (!stereo1 && !stereo2) {
                    const float distex = ex - keys2[idx2].x, distey = ey - keys2[idx2].y;
                    if (distex * distex + distey * distey < 100 * scale[keys2[idx2].octave]) continue;
                }
                if (check_dist_epipolar_line(keys1[idx1].x, keys1[idx1].y, keys2[idx2].x, keys2[idx2].y, F12, sigma2[keys2[idx2].octave])) {
                    best_idx2 = idx2;
                    best_dist = d;
                }
            }
            if (best_idx2 >= 0) {
                match12[idx1] = best_idx2;
                matched2[best_idx2] = 1;
                nm++;
                if (check_ori) hist[rot_bin(keys1[idx1].angle, keys2[best_idx2].angle)].push_back(idx1);
            }
        }
    }
    if (check_ori) {
        int32_t sizes[HISTO_LENGTH];
        for (int b = 0; b < HISTO_LENGTH; b++) sizes[b] = (int32_t)hist[b].size();
        int i1, i2, i3;
        orbfe_three_maxima(sizes, HISTO_LENGTH, &i1, &i2, &i3);
        for (int b = 0; b < HISTO_LENGTH; b++) {
            if (b == i1 || b == i2 || b == i3) continue;
            for (int idx : hist[b]) { match12[idx] = -1; nm--; }
        }
    }
    *nmatches = nm;
    return ORBFE_OK;
} ORBFE_CATCH(ctx)
