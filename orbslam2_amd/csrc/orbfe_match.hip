// orbfe_match.hip -- the Tracking-thread matchers.
//
// Split the way SURVEY.md §7.1 prescribes: the data-parallel part -- Frame grid, window query
// (Frame::GetFeaturesInArea, src/Frame.cc:328-381) and the Hamming distance of every (query, candidate)
// pair -- runs on the GPU, one wave per query; the sequentially greedy resolution (a keypoint taken by an
// earlier map point changes what a later one may take: src/ORBmatcher.cc:85-87,1399-1401,1537-1538,
// 439-440) runs in order on the host over the returned candidate lists.  A candidate is one 64-bit key
//      dist << 36 | ix << 30 | iy << 24 | idx << 8 | octave
// so "first minimum in GetFeaturesInArea order" (cell-x major, cell-y, insertion = keypoint index) is
// simply the smallest key, independent of the order in which the GPU emitted the list.
// Projection / frustum arithmetic is host code in the reference's evaluation order (contract Q4: no FMA
// contraction; cv::Mat 3x3*3x1+t as OpenCV's small-matrix gemm path; PredictScale's log through one
// deterministic routine).  The candidate machinery (64-bit keys, top-4 prefixes, replays) has no counterpart in the oracle, which
// walks vectors as the reference does; the float statements of the projections and rot_bin are the reference's own and therefore
// read the same on both sides (oracle/literal_matchers.py is the independent second reading of the control flow).
#include "../../include/orbfe.h"
#include "orbfe_device.h"
#include "orbfe_host.h"
#include "orbfe_match_resolve.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

// KeyFrame::IsInImage (src/KeyFrame.cc:604-607): the keyframe's bounds are ints initialised from the frame's floats
static inline bool kf_is_in_image(const orbfe_frame_view *kf, float u, float v)
{
    if (kf->keyframe) return u >= (float)(int)kf->min_x && u < (float)(int)kf->max_x && v >= (float)(int)kf->min_y && v < (float)(int)kf->max_y;
    return u >= kf->min_x && u < kf->max_x && v >= kf->min_y && v < kf->max_y;
}
#define GRID_COLS 64 // FRAME_GRID_COLS include/Frame.h:36
#define GRID_ROWS 48 // FRAME_GRID_ROWS include/Frame.h:37
using orbfe_resolve::HISTO_LENGTH;
using orbfe_resolve::TH_HIGH;
using orbfe_resolve::TH_LOW;
#define MATCH_TOPK 4       // orbfe_resolve::TOPK
#define MATCH_TOPK_LDS 256 // candidates per query the top-K selection stages in LDS; longer lists fall back to the full list

using orbfe_resolve::MatchQuery;

struct MatchFrame {
    const KeyPointPOD *keys; // mvKeysUn
    const uint8_t *desc;
    const float *u_right;    // may be null
    int n;
    float min_x, min_y, inv_w, inv_h;    // Frame::mnMinX / mnMinY, mfGridElementWidthInv / HeightInv: the grid ASSIGNMENT
    float q_min_x, q_min_y;              // bounds of the window QUERY: the same, or (float)(int) of them for a KeyFrame (src/KeyFrame.cc:568-580)
    int *cell_cnt, *cell_off, *cell_idx; // CSR over ix * GRID_ROWS + iy
};

// ---------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------
// Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:231-246,383-393; Q6: round(), column 64 dropped)
__global__ __launch_bounds__(256) void grid_count_kernel(MatchFrame f, int *cell_of)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= f.n) return;
    const int px = (int)roundf(__fmul_rn(__fsub_rn(f.keys[i].x, f.min_x), f.inv_w));
    const int py = (int)roundf(__fmul_rn(__fsub_rn(f.keys[i].y, f.min_y), f.inv_h));
    int c = -1;
    if (px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS) {
        c = px * GRID_ROWS + py;
        atomicAdd(&f.cell_cnt[c], 1);
    }
    cell_of[i] = c;
}

__global__ __launch_bounds__(1024) void grid_scan_kernel(MatchFrame f)
{
    __shared__ int s[1024];
    const int tid = threadIdx.x;
    const int per = (GRID_COLS * GRID_ROWS + 1023) / 1024; // 3
    int sum = 0;
    for (int k = 0; k < per; k++) {
        const int c = tid * per + k;
        if (c < GRID_COLS * GRID_ROWS) sum += f.cell_cnt[c];
    }
    s[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = tid >= o ? s[tid - o] : 0;
        __syncthreads();
        s[tid] += v;
        __syncthreads();
    }
    int run = s[tid] - sum;
    for (int k = 0; k < per; k++) {
        const int c = tid * per + k;
        if (c < GRID_COLS * GRID_ROWS) {
            const int v = f.cell_cnt[c];
            f.cell_off[c] = run;
            f.cell_cnt[c] = run; // becomes the fill cursor
            run += v;
        }
    }
    if (tid == 1023) f.cell_off[GRID_COLS * GRID_ROWS] = s[1023];
}

__global__ __launch_bounds__(256) void grid_fill_kernel(MatchFrame f, const int *cell_of)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= f.n) return;
    const int c = cell_of[i];
    if (c >= 0) f.cell_idx[atomicAdd(&f.cell_cnt[c], 1)] = i; // order inside a cell is irrelevant (keys carry it)
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_excl_scan(int v, int lane)
{
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    return inc - v;
}

// One wave per query: Frame::GetFeaturesInArea + DescriptorDistance of every hit.
// With topk != null the wave also selects the MATCH_TOPK smallest keys that pass the call's STATIC filters (blocked0[idx] == 0;
// gate_drop: dist < 256) and reports how many passed: the host replays the greedy rules on that prefix (orbfe_match_resolve.h).
__global__ __launch_bounds__(256) void window_candidates_kernel(MatchFrame f, const MatchQuery *__restrict__ q, const uint8_t *__restrict__ qdesc,
                                                                int nq, int *__restrict__ q_off, int *__restrict__ q_cnt,
                                                                unsigned long long *__restrict__ list, int *__restrict__ cursor, int list_cap,
                                                                unsigned long long *__restrict__ topk, int *__restrict__ n_static,
                                                                const uint8_t *__restrict__ blocked0, int gate_drop)
{
    __shared__ unsigned long long s_keys[4][MATCH_TOPK_LDS];
    const int iq = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (iq >= nq) return;
    const MatchQuery Q = q[iq];
    int total = 0, ncx = 0, ncy = 0, min_cx = 0, min_cy = 0;
    if (Q.flags & 1) {
        int v = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.u, f.q_min_x), Q.r), f.inv_w));
        min_cx = v > 0 ? v : 0;
        v = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.u, f.q_min_x), Q.r), f.inv_w));
        const int max_cx = v < GRID_COLS - 1 ? v : GRID_COLS - 1;
        v = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.v, f.q_min_y), Q.r), f.inv_h));
        min_cy = v > 0 ? v : 0;
        v = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.v, f.q_min_y), Q.r), f.inv_h));
        const int max_cy = v < GRID_ROWS - 1 ? v : GRID_ROWS - 1;
        if (min_cx < GRID_COLS && max_cx >= 0 && min_cy < GRID_ROWS && max_cy >= 0) {
            ncx = max_cx - min_cx + 1;
            ncy = max_cy - min_cy + 1;
        }
    }
    const int ncells = ncx > 0 && ncy > 0 ? ncx * ncy : 0;
    const bool check_levels = (Q.min_level > 0) || (Q.max_level >= 0); // Q5, literally
    // pass 1: count hits per lane (lane owns cells lane, lane + 64, ...)
    int mine = 0;
    for (int c = lane; c < ncells; c += 64) {
        const int cell = (min_cx + c / ncy) * GRID_ROWS + (min_cy + c % ncy);
        for (int j = f.cell_off[cell]; j < f.cell_off[cell + 1]; j++) {
            const KeyPointPOD kp = f.keys[f.cell_idx[j]];
            if (check_levels && (kp.octave < Q.min_level || (Q.max_level >= 0 && kp.octave > Q.max_level))) continue;
            if (fabsf(__fsub_rn(kp.x, Q.u)) < Q.r && fabsf(__fsub_rn(kp.y, Q.v)) < Q.r) mine++;
        }
    }
    total = wave_sum(mine);
    int base = 0;
    if (lane == 0) {
        base = total > 0 ? atomicAdd(cursor, total) : 0;
        q_off[iq] = base;
        q_cnt[iq] = (base + total <= list_cap) ? total : -total; // negative: list capacity exceeded
    }
    base = __shfl(base, 0, 64);
    if (topk && (total == 0 || base + total > list_cap)) { // nothing (or nothing usable): an empty prefix, the host re-runs on overflow
        if (lane < MATCH_TOPK) topk[(size_t)iq * MATCH_TOPK + lane] = ~0ull;
        if (lane == 0) n_static[iq] = 0;
    }
    if (total == 0 || base + total > list_cap) return;
    const int rel0 = wave_excl_scan(mine, lane);
    int pos = base + rel0;
    unsigned long long *sk = s_keys[threadIdx.x >> 6];
    const bool stage = topk && total <= MATCH_TOPK_LDS;
    int rel = rel0;
    uint32_t qd[8];
    {
        const uint32_t *p = (const uint32_t *)(qdesc + (size_t)iq * 32);
#pragma unroll
        for (int k = 0; k < 8; k++) qd[k] = p[k];
    }
    for (int c = lane; c < ncells; c += 64) {
        const int ix = min_cx + c / ncy, iy = min_cy + c % ncy;
        const int cell = ix * GRID_ROWS + iy;
        for (int j = f.cell_off[cell]; j < f.cell_off[cell + 1]; j++) {
            const int idx = f.cell_idx[j];
            const KeyPointPOD kp = f.keys[idx];
            if (check_levels && (kp.octave < Q.min_level || (Q.max_level >= 0 && kp.octave > Q.max_level))) continue;
            if (!(fabsf(__fsub_rn(kp.x, Q.u)) < Q.r && fabsf(__fsub_rn(kp.y, Q.v)) < Q.r)) continue;
            unsigned dist = 0;
            const uint32_t *p = (const uint32_t *)(f.desc + (size_t)idx * 32);
#pragma unroll
            for (int k = 0; k < 8; k++) dist += __popc(qd[k] ^ p[k]);
            // the mvuRight gate (src/ORBmatcher.cc:93-98,1403-1409) is a pure function of the pair: mark it
            if ((Q.flags & 2) && f.u_right && f.u_right[idx] > 0 && fabsf(__fsub_rn(Q.ur, f.u_right[idx])) > Q.ur_rad) dist = 511;
            const unsigned long long key = ((unsigned long long)dist << 36) | ((unsigned long long)ix << 30) | ((unsigned long long)iy << 24) |
                                           ((unsigned long long)idx << 8) | (unsigned long long)(kp.octave & 255);
            list[pos++] = key;
            if (stage) sk[rel++] = ((gate_drop && dist >= 256) || (blocked0 && blocked0[idx])) ? ~0ull : key; // static filters
        }
    }
    if (!topk) return;
    if (!stage) { // too long for the LDS stage: the host takes this query from the full list
        if (lane < MATCH_TOPK) topk[(size_t)iq * MATCH_TOPK + lane] = ~0ull;
        if (lane == 0) n_static[iq] = INT_MAX;
        return;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // each lane owns keys lane, lane + 64, ...; MATCH_TOPK rounds of (lane minimum, wave minimum, owner retires its key)
    unsigned long long mykeys[MATCH_TOPK_LDS / 64];
    int passed = 0;
#pragma unroll
    for (int j = 0; j < MATCH_TOPK_LDS / 64; j++) {
        mykeys[j] = (lane + 64 * j < total) ? sk[lane + 64 * j] : ~0ull;
        passed += mykeys[j] != ~0ull;
    }
    passed = wave_sum(passed);
    for (int r = 0; r < MATCH_TOPK; r++) {
        unsigned long long m = mykeys[0];
#pragma unroll
        for (int j = 1; j < MATCH_TOPK_LDS / 64; j++) m = mykeys[j] < m ? mykeys[j] : m;
        unsigned long long w = m;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)w, o, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(w >> 32), o, 64);
            const unsigned long long t = ((unsigned long long)hi << 32) | lo;
            w = t < w ? t : w;
        }
        if (lane == 0) topk[(size_t)iq * MATCH_TOPK + r] = w;
        if (w != ~0ull) { // keys are unique within a query (they carry the keypoint index): exactly one lane retires it
#pragma unroll
            for (int j = 0; j < MATCH_TOPK_LDS / 64; j++)
                if (mykeys[j] == w) mykeys[j] = ~0ull;
        }
    }
    if (lane == 0) n_static[iq] = passed;
}

// ---------------------------------------------------------------------------------------------
// host: device session for one frame + candidate query
// ---------------------------------------------------------------------------------------------
struct orbfe_match_state {
    DevBuf in_blk, out_blk, cells, cell_of, list, keys_un; // inputs of a call in one block (one pinned copy up), results in another
    uint8_t *h_in = nullptr, *h_out = nullptr;               // pinned images of in_blk / out_blk
    size_t h_in_bytes = 0, h_out_bytes = 0;
    std::vector<int> h_off, h_cnt, h_nstatic;
    std::vector<unsigned long long> h_list, h_topk;
    bool list_on_host = false; // h_list / h_off / h_cnt hold the device list of the latest query
    int list_total = 0;
    size_t lazy_off = 0, lazy_cnt = 0; // where the offsets / counts of a top-K query sit in out_blk (downloaded on demand)
    int lazy_nq = 0;
    // grid of a device-resident frame (image slot of the latest extraction call): built once per frame
    unsigned grid_epoch = 0;
    int grid_slot = -1, grid_n = -1;
    float grid_bounds[4] = {0, 0, 0, 0};
    ~orbfe_match_state() { if (h_in) (void)hipHostFree(h_in); if (h_out) (void)hipHostFree(h_out); }
};

static orbfe_match_state *match_state(orbfe_context *ctx) { return orbfe_ctx_match_state(ctx); }

#define MTRY(ctx, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return orbfe_fail(ctx, ORBFE_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

static int ensure_pinned(orbfe_context *ctx, uint8_t *&h, size_t &have, size_t need)
{
    if (have >= need) return ORBFE_OK;
    if (h) (void)hipHostFree(h);
    h = nullptr; have = 0;
    MTRY(ctx, hipHostMalloc((void **)&h, need, hipHostMallocDefault));
    have = need;
    return ORBFE_OK;
}

// Static filters of a Tracking matcher call, applied on the device before the top-K selection (orbfe_match_resolve.h)
struct TopkRequest {
    const uint8_t *blocked0 = nullptr; // [frame n] keypoints blocked before the call, or null
    bool gate_drop = false;            // drop keys that failed the mvuRight gate (dist marked 511)
};

// Runs the window query for `nq` queries against the frame and leaves the per-query results on the host:
//   topk == null : every candidate key (st->h_list / h_off / h_cnt), as the keyframe-side matchers want them;
//   topk != null : the MATCH_TOPK best statically admissible keys per query (st->h_topk / h_nstatic); the full list stays in
//                  HBM and is fetched by fetch_full_list() only if a replay runs out of its prefix.
// Frame source: fv->device_slot_plus1 == 0: the host arrays of the view are uploaded and bucketed (any frame, any keyframe);
// > 0: image slot (value - 1) of the LATEST extraction call of this context -- keypoints and descriptors are read where the
// extraction left them in HBM, the grid is built once per frame and reused by every later call on that frame.
static int run_window_queries(orbfe_context *ctx, const orbfe_frame_view *fv, const std::vector<MatchQuery> &queries,
                              const std::vector<uint8_t> &qdesc, const TopkRequest *topk = nullptr)
{
    orbfe_match_state *st = match_state(ctx);
    hipStream_t s = orbfe_ctx_stream(ctx);
    const int n = fv->n, nq = (int)queries.size();
    st->h_off.assign(nq, 0); st->h_cnt.assign(nq, 0); st->h_list.clear();
    st->h_topk.assign((size_t)nq * MATCH_TOPK, ~0ull); st->h_nstatic.assign(nq, 0);
    st->list_on_host = true; st->list_total = 0;
    if (n <= 0 || nq == 0) return ORBFE_OK;
    if (n > 65535) return orbfe_fail(ctx, ORBFE_ERR_UNSUPPORTED, "frames with more than 65535 keypoints are not supported by the matchers");
    MTRY(ctx, hipSetDevice(orbfe_ctx_device(ctx)));
    const size_t ncell = GRID_COLS * GRID_ROWS;
    const bool resident = fv->device_slot_plus1 > 0;
    const int slot = fv->device_slot_plus1 - 1;
    const DeviceConfig *cfg = orbfe_ctx_config(ctx);
    const DeviceBuffers *buf = orbfe_ctx_buffers(ctx);
    if (resident) {
        if (slot >= orbfe_ctx_params(ctx)->max_images) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "device slot %d out of range", slot);
        int rc = orbfe_ctx_wait_foreign_stream(ctx); // the extraction may have been enqueued on a caller stream
        if (rc != ORBFE_OK) return rc;
        // a stale or mismatched view would silently match against another frame's leftovers: the view's count must be the slot's
        int slot_n = 0;
        rc = orbfe_ctx_slot_count(ctx, slot, &slot_n);
        if (rc != ORBFE_OK) return rc;
        if (n != slot_n) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "frame view holds %d keypoints, device slot %d holds %d (not the latest extraction?)", n, slot, slot_n);
    }
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t fn = resident ? 0 : (size_t)n; // frame rows in the upload block
    const size_t i_keys = 0, i_desc = up16(i_keys + sizeof(KeyPointPOD) * fn), i_ur = up16(i_desc + (size_t)32 * fn);
    const size_t i_q = up16(i_ur + (fv->u_right ? sizeof(float) * (size_t)n : 0)), i_qd = up16(i_q + sizeof(MatchQuery) * nq), i_blk = up16(i_qd + (size_t)32 * nq);
    const size_t in_bytes = up16(i_blk + (topk && topk->blocked0 ? (size_t)n : 0));
    // results: [cursor | n_static | top-K keys] are what a top-K replay needs; [offsets | counts] follow and are only downloaded
    // with the full list
    const size_t o_cur = 0, o_ns = 16, o_tk = up16(o_ns + sizeof(int) * nq), o_off = up16(o_tk + sizeof(unsigned long long) * MATCH_TOPK * nq);
    const size_t o_cnt = up16(o_off + sizeof(int) * nq), out_bytes = up16(o_cnt + sizeof(int) * nq);
    const size_t topk_bytes = o_off; // prefix of the result block a top-K replay downloads
    if (st->in_blk.ensure(in_bytes) || st->out_blk.ensure(out_bytes) || st->cells.ensure(sizeof(int) * (2 * ncell + 2 + cfg->sel_total + n)) ||
        st->cell_of.ensure(sizeof(int) * (size_t)(n > cfg->sel_total ? n : cfg->sel_total)))
        return orbfe_fail(ctx, ORBFE_ERR_HIP, "matcher scratch allocation failed");
    int rc = ensure_pinned(ctx, st->h_in, st->h_in_bytes, in_bytes);
    if (rc != ORBFE_OK) return rc;
    rc = ensure_pinned(ctx, st->h_out, st->h_out_bytes, out_bytes);
    if (rc != ORBFE_OK) return rc;
    uint8_t *din = (uint8_t *)st->in_blk.p, *dout = (uint8_t *)st->out_blk.p;
    MatchFrame f;
    f.n = n; f.min_x = fv->min_x; f.min_y = fv->min_y;
    f.q_min_x = fv->keyframe ? (float)(int)fv->min_x : fv->min_x; // KeyFrame::mnMinX is an int initialised from the frame's float
    f.q_min_y = fv->keyframe ? (float)(int)fv->min_y : fv->min_y;
    f.inv_w = (float)GRID_COLS / (fv->max_x - fv->min_x); // mfGridElementWidthInv, src/Frame.cc:99
    f.inv_h = (float)GRID_ROWS / (fv->max_y - fv->min_y);
    f.cell_cnt = (int *)st->cells.p; f.cell_off = f.cell_cnt + ncell; f.cell_idx = f.cell_off + ncell + 1;
    bool build_grid = true;
    if (resident) {
        const size_t so = (size_t)slot * cfg->sel_total;
        const KeyPointPOD *raw = (const KeyPointPOD *)buf->kps + so;
        const bool distorted = cfg->n_dist > 0 && cfg->dist[0] != 0.0f;
        const unsigned epoch = orbfe_ctx_epoch(ctx);
        build_grid = !(st->grid_epoch == epoch && st->grid_slot == slot && st->grid_n == n && st->grid_bounds[0] == fv->min_x &&
                       st->grid_bounds[1] == fv->max_x && st->grid_bounds[2] == fv->min_y && st->grid_bounds[3] == fv->max_y);
        if (distorted) { // mvKeysUn: undistorted on the device once per frame (Frame::UndistortKeyPoints)
            if (st->keys_un.ensure(sizeof(KeyPointPOD) * (size_t)cfg->sel_total)) return orbfe_fail(ctx, ORBFE_ERR_HIP, "matcher scratch allocation failed");
            if (build_grid) orbfe_launch_undistort(*cfg, raw, st->keys_un.p, n, s);
            f.keys = (const KeyPointPOD *)st->keys_un.p;
        } else f.keys = raw;
        f.desc = buf->desc + so * 32;
        // mvuRight travels with the call (n floats): Frame::ComputeStereoMatches may have run on the host in an integration
        // that keeps the reference's Frame.cc, and then the device copy of this slot is not the frame's mvuRight
        f.u_right = fv->u_right ? (const float *)(din + i_ur) : nullptr;
        st->grid_epoch = epoch; st->grid_slot = slot; st->grid_n = n;
        st->grid_bounds[0] = fv->min_x; st->grid_bounds[1] = fv->max_x; st->grid_bounds[2] = fv->min_y; st->grid_bounds[3] = fv->max_y;
    } else {
        f.keys = (const KeyPointPOD *)(din + i_keys); f.desc = din + i_desc;
        f.u_right = fv->u_right ? (const float *)(din + i_ur) : nullptr;
        memcpy(st->h_in + i_keys, fv->keys_un, sizeof(KeyPointPOD) * n);
        memcpy(st->h_in + i_desc, fv->descriptors, (size_t)32 * n);
        st->grid_slot = -1; // the grid buffers now describe a host frame
    }
    if (fv->u_right) memcpy(st->h_in + i_ur, fv->u_right, sizeof(float) * n);
    memcpy(st->h_in + i_q, queries.data(), sizeof(MatchQuery) * nq);
    memcpy(st->h_in + i_qd, qdesc.data(), (size_t)32 * nq);
    if (topk && topk->blocked0) memcpy(st->h_in + i_blk, topk->blocked0, (size_t)n);
    MTRY(ctx, hipMemcpyAsync(din, st->h_in, in_bytes, hipMemcpyHostToDevice, s));
    if (build_grid) {
        MTRY(ctx, hipMemsetAsync(f.cell_cnt, 0, sizeof(int) * ncell, s));
        hipLaunchKernelGGL(grid_count_kernel, dim3((n + 255) / 256), dim3(256), 0, s, f, (int *)st->cell_of.p);
        hipLaunchKernelGGL(grid_scan_kernel, dim3(1), dim3(1024), 0, s, f);
        hipLaunchKernelGGL(grid_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, s, f, (const int *)st->cell_of.p);
    }
    // candidate list capacity: grown and the query re-run if a frame overflows it
    size_t cap = st->list.bytes / 8;
    if (cap < (size_t)nq * 64) cap = (size_t)nq * 64;
    for (int attempt = 0; attempt < 2; attempt++) {
        if (st->list.ensure(cap * 8)) return orbfe_fail(ctx, ORBFE_ERR_HIP, "candidate list allocation failed");
        MTRY(ctx, hipMemsetAsync(dout + o_cur, 0, sizeof(int), s));
        hipLaunchKernelGGL(window_candidates_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, f, (const MatchQuery *)(din + i_q),
                           (const uint8_t *)(din + i_qd), nq, (int *)(dout + o_off), (int *)(dout + o_cnt),
                           (unsigned long long *)st->list.p, (int *)(dout + o_cur), (int)std::min<size_t>(cap, INT_MAX),
                           topk ? (unsigned long long *)(dout + o_tk) : nullptr, (int *)(dout + o_ns),
                           topk && topk->blocked0 ? (const uint8_t *)(din + i_blk) : nullptr, topk && topk->gate_drop ? 1 : 0);
        MTRY(ctx, hipMemcpyAsync(st->h_out, dout, topk ? topk_bytes : out_bytes, hipMemcpyDeviceToHost, s));
        MTRY(ctx, hipStreamSynchronize(s));
        MTRY(ctx, hipGetLastError());
        const int total = *(const int *)(st->h_out + o_cur);
        if ((size_t)total <= cap) {
            st->list_total = total;
            if (topk) {
                memcpy(st->h_nstatic.data(), st->h_out + o_ns, sizeof(int) * nq);
                memcpy(st->h_topk.data(), st->h_out + o_tk, sizeof(unsigned long long) * MATCH_TOPK * nq);
                st->list_on_host = total == 0;
                st->lazy_off = o_off; st->lazy_cnt = o_cnt; st->lazy_nq = nq;
                return ORBFE_OK;
            }
            memcpy(st->h_off.data(), st->h_out + o_off, sizeof(int) * nq);
            memcpy(st->h_cnt.data(), st->h_out + o_cnt, sizeof(int) * nq);
            st->h_list.resize(total);
            if (total > 0) MTRY(ctx, hipMemcpy(st->h_list.data(), st->list.p, sizeof(unsigned long long) * total, hipMemcpyDeviceToHost));
            return ORBFE_OK;
        }
        cap = (size_t)total + 1024;
    }
    return orbfe_fail(ctx, ORBFE_ERR_CAPACITY, "candidate list overflow");
}

// CandidateSource::full of the top-K replays: the device list of the latest query, downloaded once on first use
struct FullListCtx { orbfe_context *ctx; orbfe_match_state *st; };
static int fetch_full_list(void *user, int q, std::vector<orbfe_resolve::ckey_t> &out)
{
    FullListCtx *c = (FullListCtx *)user;
    orbfe_match_state *st = c->st;
    if (!st->list_on_host) {
        st->h_list.resize(st->list_total);
        const uint8_t *dout = (const uint8_t *)st->out_blk.p;
        if (hipMemcpy(st->h_off.data(), dout + st->lazy_off, sizeof(int) * st->lazy_nq, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(st->h_cnt.data(), dout + st->lazy_cnt, sizeof(int) * st->lazy_nq, hipMemcpyDeviceToHost) != hipSuccess)
            return -1;
        if (st->list_total > 0 &&
            hipMemcpy(st->h_list.data(), st->list.p, sizeof(unsigned long long) * st->list_total, hipMemcpyDeviceToHost) != hipSuccess)
            return -1;
        st->list_on_host = true;
    }
    out.assign(st->h_list.begin() + st->h_off[q], st->h_list.begin() + st->h_off[q] + st->h_cnt[q]);
    return 0;
}

static orbfe_resolve::CandidateSource topk_source(orbfe_match_state *st, FullListCtx *fl, const TopkRequest &req)
{
    orbfe_resolve::CandidateSource src;
    src.topk = st->h_topk.data();
    src.n_static = st->h_nstatic.data();
    src.blocked0 = req.blocked0;
    src.drop_gated = req.gate_drop;
    src.user = fl;
    src.full = fetch_full_list;
    return src;
}

using orbfe_resolve::key_dist;
using orbfe_resolve::key_idx;
using orbfe_resolve::key_level;

using orbfe_resolve::camera_center;
using orbfe_resolve::predict_scale;
using orbfe_resolve::rt_apply;
using orbfe_resolve::RotHist;
using orbfe_resolve::rot_bin;

extern "C" int orbfe_three_maxima(const int32_t *histo_sizes, int L, int *ind1, int *ind2, int *ind3)
try {
    if (!histo_sizes || !ind1 || !ind2 || !ind3 || L < 0) return ORBFE_ERR_INVALID;
    orbfe_resolve::three_maxima(histo_sizes, L, ind1, ind2, ind3); // ORBmatcher::ComputeThreeMaxima, src/ORBmatcher.cc:1597-1638
    return ORBFE_OK;
} ORBFE_CATCH(nullptr)

static int check_view(orbfe_context *ctx, const orbfe_frame_view *fv)
{
    if (!ctx || !fv || fv->n < 0 || (fv->n > 0 && (!fv->keys_un || (!fv->descriptors && fv->device_slot_plus1 <= 0))) || fv->device_slot_plus1 < 0 ||
        !(fv->max_x > fv->min_x) || !(fv->max_y > fv->min_y))
        return orbfe_fail(ctx, ORBFE_ERR_INVALID, "bad frame view");
    return ORBFE_OK;
}

// Frame::GetFeaturesInArea (src/Frame.cc:328-381) for one window, in the reference's result order.
extern "C" int orbfe_features_in_area(orbfe_context *ctx, const orbfe_frame_view *fv, float x, float y, float r,
                                      int min_level, int max_level, int32_t *out, int cap, int *n)
try {
    ORBFE_ENTRY(ctx);
    int rc = check_view(ctx, fv);
    if (rc != ORBFE_OK) return rc;
    if (!n) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    std::vector<MatchQuery> q(1);
    q[0] = MatchQuery{x, y, r, min_level, max_level, 0.f, 0.f, 1};
    std::vector<uint8_t> qd(32, 0);
    rc = run_window_queries(ctx, fv, q, qd);
    if (rc != ORBFE_OK) return rc;
    orbfe_match_state *st = match_state(ctx);
    std::vector<unsigned long long> keys(st->h_list.begin() + st->h_off[0], st->h_list.begin() + st->h_off[0] + st->h_cnt[0]);
    for (auto &k : keys) k &= ((1ull << 36) - 1); // drop the distance: order = (ix, iy, idx)
    std::sort(keys.begin(), keys.end());
    *n = (int)keys.size();
    if ((int)keys.size() > cap) return orbfe_fail(ctx, ORBFE_ERR_CAPACITY, "window holds %d keypoints, caller buffer %d", (int)keys.size(), cap);
    for (size_t i = 0; i < keys.size(); i++) out[i] = key_idx(keys[i]);
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// Frame::AssignFeaturesToGrid (src/Frame.cc:231-246): mGrid as CSR.  The grid is the one every matcher call on this frame uses
// (built by the same kernels; for a device-resident frame it stays cached for the calls that follow).
extern "C" int orbfe_assign_features_to_grid(orbfe_context *ctx, const orbfe_frame_view *fv, int32_t *cell_off, int32_t *cell_idx)
try {
    ORBFE_ENTRY(ctx);
    int rc = check_view(ctx, fv);
    if (rc != ORBFE_OK) return rc;
    if (!cell_off || (fv->n > 0 && !cell_idx)) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    const int ncell = GRID_COLS * GRID_ROWS;
    if (fv->n == 0) { memset(cell_off, 0, sizeof(int32_t) * (ncell + 1)); return ORBFE_OK; }
    std::vector<MatchQuery> q(1);
    q[0] = MatchQuery{0.f, 0.f, 0.f, -1, -1, 0.f, 0.f, 0}; // flags 0: no window is searched, the call only builds the grid
    std::vector<uint8_t> qd(32, 0);
    rc = run_window_queries(ctx, fv, q, qd);
    if (rc != ORBFE_OK) return rc;
    orbfe_match_state *st = match_state(ctx);
    const int *d_off = (const int *)st->cells.p + ncell;
    MTRY(ctx, hipMemcpy(cell_off, d_off, sizeof(int) * (ncell + 1), hipMemcpyDeviceToHost));
    const int total = cell_off[ncell]; // keypoints that fell inside the grid (PosInGrid drops the others)
    if (total < 0 || total > fv->n) return orbfe_fail(ctx, ORBFE_ERR_HIP, "grid holds %d entries for %d keypoints", total, fv->n);
    if (total > 0) MTRY(ctx, hipMemcpy(cell_idx, d_off + ncell + 1, sizeof(int) * total, hipMemcpyDeviceToHost));
    // the fill kernel claims a cell's slots through an atomic cursor (the matchers order candidates by their keys, not by list
    // position); mGrid[ix][iy] is in push_back order = ascending keypoint index
    for (int c = 0; c < ncell; c++)
        if (cell_off[c + 1] - cell_off[c] > 1) std::sort(cell_idx + cell_off[c], cell_idx + cell_off[c + 1]);
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// Many GetFeaturesInArea queries against one frame: the frame is uploaded and its grid built once.
extern "C" int orbfe_features_in_area_batch(orbfe_context *ctx, const orbfe_frame_view *fv, int nq, const float *x, const float *y,
                                            const float *r, const int32_t *min_level, const int32_t *max_level,
                                            int32_t *out_off, int32_t *out, int cap)
try {
    ORBFE_ENTRY(ctx);
    int rc = check_view(ctx, fv);
    if (rc != ORBFE_OK) return rc;
    if (nq < 0 || !out_off || (nq > 0 && (!x || !y || !r))) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    std::vector<MatchQuery> q(nq);
    for (int i = 0; i < nq; i++) q[i] = MatchQuery{x[i], y[i], r[i], min_level ? min_level[i] : -1, max_level ? max_level[i] : -1, 0.f, 0.f, 1};
    std::vector<uint8_t> qd((size_t)32 * std::max(nq, 1), 0);
    rc = run_window_queries(ctx, fv, q, qd);
    if (rc != ORBFE_OK) return rc;
    orbfe_match_state *st = match_state(ctx);
    int total = 0;
    out_off[0] = 0;
    std::vector<unsigned long long> keys;
    for (int i = 0; i < nq; i++) {
        keys.assign(st->h_list.begin() + st->h_off[i], st->h_list.begin() + st->h_off[i] + st->h_cnt[i]);
        for (auto &k : keys) k &= ((1ull << 36) - 1); // drop the distance: order = (ix, iy, idx)
        std::sort(keys.begin(), keys.end());
        if (total + (int)keys.size() <= cap && out)
            for (size_t j = 0; j < keys.size(); j++) out[total + j] = key_idx(keys[j]);
        total += (int)keys.size();
        out_off[i + 1] = total;
    }
    if (total > cap) return orbfe_fail(ctx, ORBFE_ERR_CAPACITY, "the windows hold %d keypoints, caller buffer %d", total, cap);
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono), src/ORBmatcher.cc:1324-1466
extern "C" int orbfe_search_by_projection_last(orbfe_context *ctx, const orbfe_frame_view *cur,
                                               const float *Tcw_cur, const float *Tcw_last, int n_last,
                                               const float *last_pos, const uint8_t *last_desc, const int32_t *last_valid,
                                               const int32_t *last_obs, const int32_t *last_octave, const float *last_angle,
                                               const uint8_t *cur_has_obs, float th, int mono, int check_ori,
                                               int32_t *cur_match, int *nmatches)
try {
    ORBFE_ENTRY(ctx);
    int rc = check_view(ctx, cur);
    if (rc != ORBFE_OK) return rc;
    if (!Tcw_cur || !Tcw_last || !cur_match || !nmatches || n_last < 0 ||
        (n_last > 0 && (!last_pos || !last_desc || !last_valid || !last_obs || !last_octave || !last_angle)))
        return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    const orbfe_params *P = orbfe_ctx_params(ctx);
    const int N = cur->n;
    static const bool trace = getenv("ORBFE_HOST_TRACE") != nullptr; // where a call's time goes: projection | device round trip | replay
    auto now_ms = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = trace ? now_ms() : 0.0;
    std::vector<MatchQuery> q;
    std::vector<uint8_t> qd;
    if (orbfe_resolve::build_queries_last(orbfe_resolve::camera_of(P), orbfe_ctx_scale_factors(ctx), P->nlevels, cur->min_x, cur->max_x, cur->min_y, cur->max_y,
                                          Tcw_cur, Tcw_last, n_last, last_pos, last_desc, last_valid, last_octave, th, mono, q, qd) != 0)
        return orbfe_fail(ctx, ORBFE_ERR_INVALID, "octave out of range");
    std::vector<uint8_t> has_obs(N > 0 ? N : 1, 0);
    for (int i = 0; i < N; i++) has_obs[i] = cur_has_obs ? (cur_has_obs[i] != 0) : 0;
    std::vector<uint8_t> blocked0(has_obs); // the static filter is the state BEFORE the replay mutates has_obs
    TopkRequest req;
    req.blocked0 = cur_has_obs ? blocked0.data() : nullptr;
    req.gate_drop = true; // 511 = failed the mvuRight gate
    const double t1 = trace ? now_ms() : 0.0;
    rc = run_window_queries(ctx, cur, q, qd, &req);
    if (rc != ORBFE_OK) return rc;
    const double t2 = trace ? now_ms() : 0.0;
    orbfe_match_state *st = match_state(ctx);
    FullListCtx fl{ctx, st};
    orbfe_resolve::CandidateSource src = topk_source(st, &fl, req);
    *nmatches = orbfe_resolve::resolve_last(src, n_last, last_obs, last_angle, N, &cur->keys_un[0].angle, sizeof(orbfe_keypoint), has_obs, check_ori, cur_match);
    if (src.error) return orbfe_fail(ctx, ORBFE_ERR_HIP, "candidate list download failed");
    if (trace)
        fprintf(stderr, "[orbfe] SearchByProjection(last): queries %.3f  device round trip %.3f  replay %.3f ms (%d points, %d keypoints)\n",
                t1 - t0, t2 - t1, now_ms() - t2, n_last, N);
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// Frame::isInFrustum, src/Frame.cc:270-326, for n map points
extern "C" int orbfe_is_in_frustum(orbfe_context *ctx, const float *Tcw, float min_x, float max_x, float min_y, float max_y,
                                   int n, const float *pos, const float *normal, const float *max_distance,
                                   const float *min_distance, float viewing_cos_limit, orbfe_track_point *out)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !Tcw || n < 0 || (n > 0 && (!pos || !normal || !max_distance || !min_distance || !out)))
        return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    const orbfe_params *P = orbfe_ctx_params(ctx);
    const float log_sf = logf((float)(double)P->scale_factor); // mfLogScaleFactor = log(mfScaleFactor), src/Frame.cc:71
    orbfe_resolve::is_in_frustum(orbfe_resolve::camera_of(P), P->nlevels, log_sf, Tcw, min_x, max_x, min_y, max_y, n, pos, normal, max_distance, min_distance,
                                 viewing_cos_limit, out);
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th), src/ORBmatcher.cc:43-135
extern "C" int orbfe_search_by_projection_points(orbfe_context *ctx, const orbfe_frame_view *cur, int n_pts,
                                                 const orbfe_track_point *pts, const uint8_t *pt_desc, const int32_t *pt_obs,
                                                 const uint8_t *cur_has_obs, float th, float nnratio,
                                                 int32_t *cur_match, int *nmatches)
try {
    ORBFE_ENTRY(ctx);
    int rc = check_view(ctx, cur);
    if (rc != ORBFE_OK) return rc;
    if (!cur_match || !nmatches || n_pts < 0 || (n_pts > 0 && (!pts || !pt_desc || !pt_obs))) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    const orbfe_params *P = orbfe_ctx_params(ctx);
    const int N = cur->n;
    std::vector<MatchQuery> q;
    std::vector<uint8_t> qd;
    if (orbfe_resolve::build_queries_points(orbfe_ctx_scale_factors(ctx), P->nlevels, n_pts, pts, pt_desc, th, q, qd) != 0)
        return orbfe_fail(ctx, ORBFE_ERR_INVALID, "predicted level out of range");
    std::vector<uint8_t> has_obs(N > 0 ? N : 1, 0);
    for (int i = 0; i < N; i++) has_obs[i] = cur_has_obs ? (cur_has_obs[i] != 0) : 0;
    std::vector<uint8_t> blocked0(has_obs);
    TopkRequest req;
    req.blocked0 = cur_has_obs ? blocked0.data() : nullptr;
    req.gate_drop = true;
    rc = run_window_queries(ctx, cur, q, qd, &req);
    if (rc != ORBFE_OK) return rc;
    orbfe_match_state *st = match_state(ctx);
    FullListCtx fl{ctx, st};
    orbfe_resolve::CandidateSource src = topk_source(st, &fl, req);
    *nmatches = orbfe_resolve::resolve_points(src, n_pts, pt_obs, N, has_obs, nnratio, cur_match);
    if (src.error) return orbfe_fail(ctx, ORBFE_ERR_HIP, "candidate list download failed");
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// ORBmatcher::SearchByProjection(Frame&, KeyFrame*, set, th, ORBdist), src/ORBmatcher.cc:1468-1595
extern "C" int orbfe_search_by_projection_kf(orbfe_context *ctx, const orbfe_frame_view *cur, const float *Tcw_cur, int n_kf,
                                             const float *kf_pos, const uint8_t *kf_desc, const int32_t *kf_valid,
                                             const float *kf_angle, const float *kf_max_distance, const float *kf_min_distance,
                                             const uint8_t *cur_has_point, float th, int orb_dist, int check_ori,
                                             int32_t *cur_match, int *nmatches)
try {
    ORBFE_ENTRY(ctx);
    int rc = check_view(ctx, cur);
    if (rc != ORBFE_OK) return rc;
    if (!Tcw_cur || !cur_match || !nmatches || n_kf < 0 ||
        (n_kf > 0 && (!kf_pos || !kf_desc || !kf_valid || !kf_angle || !kf_max_distance || !kf_min_distance)))
        return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    const orbfe_params *P = orbfe_ctx_params(ctx);
    const int N = cur->n;
    std::vector<MatchQuery> q;
    std::vector<uint8_t> qd;
    orbfe_resolve::build_queries_kf(orbfe_resolve::camera_of(P), orbfe_ctx_scale_factors(ctx), P->nlevels, logf((float)(double)P->scale_factor), cur->min_x, cur->max_x,
                                    cur->min_y, cur->max_y, Tcw_cur, n_kf, kf_pos, kf_desc, kf_valid, kf_max_distance, kf_min_distance, th, q, qd);
    std::vector<uint8_t> has_pt(N > 0 ? N : 1, 0);
    for (int i = 0; i < N; i++) has_pt[i] = cur_has_point ? (cur_has_point[i] != 0) : 0;
    std::vector<uint8_t> blocked0(has_pt);
    TopkRequest req;
    req.blocked0 = cur_has_point ? blocked0.data() : nullptr;
    req.gate_drop = false; // this overload has no mvuRight gate
    rc = run_window_queries(ctx, cur, q, qd, &req);
    if (rc != ORBFE_OK) return rc;
    orbfe_match_state *st = match_state(ctx);
    FullListCtx fl{ctx, st};
    orbfe_resolve::CandidateSource src = topk_source(st, &fl, req);
    *nmatches = orbfe_resolve::resolve_kf(src, n_kf, kf_angle, N, &cur->keys_un[0].angle, sizeof(orbfe_keypoint), has_pt, orb_dist, check_ori, cur_match);
    if (src.error) return orbfe_fail(ctx, ORBFE_ERR_HIP, "candidate list download failed");
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// Search part of ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th), src/ORBmatcher.cc:821-971: per candidate map
// point the keyframe keypoint to fuse with.  The points do not interact (the map mutation that follows stays with the
// caller), so this is one window query per point; the chi-square gates of :905-930 are applied to its candidates here.
extern "C" int orbfe_fuse(orbfe_context *ctx, const orbfe_frame_view *kf, const float *Tcw, int n_pts,
                          const float *pos, const float *normal, const float *max_distance, const float *min_distance,
                          const uint8_t *pt_desc, const int32_t *pt_valid, float th, int32_t *best_idx, int *n_fused)
try {
    ORBFE_ENTRY(ctx);
    int rc = check_view(ctx, kf);
    if (rc != ORBFE_OK) return rc;
    if (!Tcw || !n_fused || n_pts < 0 || (n_pts > 0 && (!pos || !normal || !max_distance || !min_distance || !pt_desc || !pt_valid || !best_idx)))
        return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    const orbfe_params *P = orbfe_ctx_params(ctx);
    const float *sf = orbfe_ctx_scale_factors(ctx);
    const float log_sf = logf((float)(double)P->scale_factor);
    float ow[3];
    camera_center(Tcw, ow);
    std::vector<MatchQuery> q(n_pts);
    std::vector<uint8_t> qd((size_t)32 * (n_pts > 0 ? n_pts : 1));
    std::vector<float> pu(n_pts > 0 ? n_pts : 1), pv(n_pts > 0 ? n_pts : 1), pur(n_pts > 0 ? n_pts : 1);
    for (int i = 0; i < n_pts; i++) {
        MatchQuery &Q = q[i];
        Q = MatchQuery{0, 0, 0, 0, -1, 0, 0, 0};
        best_idx[i] = -1;
        if (!pt_valid[i]) continue;
        float pc[3];
        rt_apply(Tcw, pos + 3 * i, pc);
        if (pc[2] < 0.0f) continue;
        const float invz = 1 / pc[2];
        const float x = pc[0] * invz, y = pc[1] * invz;
        const float u = P->fx * x + P->cx;
        const float v = P->fy * y + P->cy;
        if (!kf_is_in_image(kf, u, v)) continue; // KeyFrame::IsInImage
        float po[3];
        for (int k = 0; k < 3; k++) po[k] = pos[3 * i + k] - ow[k];
        const float dist3d = (float)sqrt((double)po[0] * po[0] + (double)po[1] * po[1] + (double)po[2] * po[2]);
        if (dist3d < 0.8f * min_distance[i] || dist3d > 1.2f * max_distance[i]) continue;
        const double dot = (double)po[0] * normal[3 * i] + (double)po[1] * normal[3 * i + 1] + (double)po[2] * normal[3 * i + 2];
        if (dot < 0.5 * (double)dist3d) continue;
        const int lvl = predict_scale(max_distance[i], dist3d, log_sf, P->nlevels);
        Q.u = u; Q.v = v; Q.r = th * sf[lvl]; Q.min_level = lvl - 1; Q.max_level = lvl; Q.flags = 1;
        pu[i] = u; pv[i] = v; pur[i] = u - P->bf * invz;
        memcpy(&qd[(size_t)32 * i], pt_desc + (size_t)32 * i, 32);
    }
    rc = run_window_queries(ctx, kf, q, qd);
    if (rc != ORBFE_OK) return rc;
    orbfe_match_state *st = match_state(ctx);
    int nf = 0;
    for (int i = 0; i < n_pts; i++) {
        unsigned long long best = ~0ull;
        for (int k = 0; k < st->h_cnt[i]; k++) {
            const unsigned long long key = st->h_list[st->h_off[i] + k];
            const int idx = key_idx(key), lv = key_level(key);
            const orbfe_keypoint &kp = kf->keys_un[idx];
            const float inv_sigma2 = 1.0f / (sf[lv] * sf[lv]); // mvInvLevelSigma2 (src/ORBextractor.cc:419-425)
            if (kf->u_right && kf->u_right[idx] >= 0) { // reprojection error in stereo (:905-918)
                const float ex = pu[i] - kp.x, ey = pv[i] - kp.y, er = pur[i] - kf->u_right[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if ((double)(e2 * inv_sigma2) > 7.8) continue;
            } else {
                const float ex = pu[i] - kp.x, ey = pv[i] - kp.y;
                const float e2 = ex * ex + ey * ey;
                if ((double)(e2 * inv_sigma2) > 5.99) continue;
            }
            if (key < best) best = key; // smallest (distance, GetFeaturesInArea order) = the loop's first minimum
        }
        if (best != ~0ull && key_dist(best) <= TH_LOW) { best_idx[i] = key_idx(best); nf++; }
    }
    *n_fused = nf;
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// Sim3 decomposition of the LoopClosing matchers (src/ORBmatcher.cc:293-298, 981-986); see oracle/orb_oracle_match.c
static void sim3_to_rt(const float *Scw, float *T)
{
    const double d = (double)Scw[0] * Scw[0] + (double)Scw[1] * Scw[1] + (double)Scw[2] * Scw[2];
    const float scw = (float)sqrt(d);
    const float alpha = (float)(1.0 / (double)scw);
    for (int i = 0; i < 12; i++) T[i] = Scw[i] * alpha;
}

// mode 0: ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th), src/ORBmatcher.cc:285-398 (greedy over points);
// mode 1: search part of ORBmatcher::Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint), :973-1096 (points independent).
static int sim3_projection_impl(orbfe_context *ctx, int mode, const orbfe_frame_view *kf, const float *Scw, int n_pts,
                                const float *pos, const float *normal, const float *max_distance, const float *min_distance,
                                const uint8_t *pt_desc, const int32_t *pt_valid, const uint8_t *kf_matched, float th,
                                int32_t *pt_match, int *nmatches)
{
    int rc = check_view(ctx, kf);
    if (rc != ORBFE_OK) return rc;
    if (!Scw || !nmatches || n_pts < 0 || (n_pts > 0 && (!pos || !normal || !max_distance || !min_distance || !pt_desc || !pt_valid || !pt_match)))
        return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    const orbfe_params *P = orbfe_ctx_params(ctx);
    const float *sf = orbfe_ctx_scale_factors(ctx);
    const float log_sf = logf((float)(double)P->scale_factor);
    const int N = kf->n;
    float T[12], ow[3];
    sim3_to_rt(Scw, T);
    camera_center(T, ow);
    std::vector<MatchQuery> q(n_pts);
    std::vector<uint8_t> qd((size_t)32 * (n_pts > 0 ? n_pts : 1));
    for (int i = 0; i < n_pts; i++) {
        MatchQuery &Q = q[i];
        Q = MatchQuery{0, 0, 0, 0, -1, 0, 0, 0};
        pt_match[i] = -1;
        if (!pt_valid[i]) continue;
        float pc[3];
        rt_apply(T, pos + 3 * i, pc);
        if (mode == 0 ? ((double)pc[2] < 0.0) : (pc[2] < 0.0f)) continue;
        const float invz = mode == 0 ? 1 / pc[2] : (float)(1.0 / (double)pc[2]);
        const float x = pc[0] * invz, y = pc[1] * invz;
        const float u = P->fx * x + P->cx;
        const float v = P->fy * y + P->cy;
        if (!kf_is_in_image(kf, u, v)) continue;
        float po[3];
        for (int k = 0; k < 3; k++) po[k] = pos[3 * i + k] - ow[k];
        const float dist = (float)sqrt((double)po[0] * po[0] + (double)po[1] * po[1] + (double)po[2] * po[2]);
        if (dist < 0.8f * min_distance[i] || dist > 1.2f * max_distance[i]) continue;
        const double dot = (double)po[0] * normal[3 * i] + (double)po[1] * normal[3 * i + 1] + (double)po[2] * normal[3 * i + 2];
        if (dot < 0.5 * (double)dist) continue;
        const int lvl = predict_scale(max_distance[i], dist, log_sf, P->nlevels);
        Q.u = u; Q.v = v; Q.r = th * sf[lvl]; Q.min_level = lvl - 1; Q.max_level = lvl; Q.flags = 1;
        memcpy(&qd[(size_t)32 * i], pt_desc + (size_t)32 * i, 32);
    }
    rc = run_window_queries(ctx, kf, q, qd);
    if (rc != ORBFE_OK) return rc;
    orbfe_match_state *st = match_state(ctx);
    std::vector<uint8_t> matched(N > 0 ? N : 1, 0);
    if (mode == 0 && kf_matched)
        for (int i = 0; i < N; i++) matched[i] = kf_matched[i];
    int nm = 0;
    for (int i = 0; i < n_pts; i++) {
        unsigned long long best = ~0ull;
        for (int k = 0; k < st->h_cnt[i]; k++) {
            const unsigned long long key = st->h_list[st->h_off[i] + k];
            if (matched[key_idx(key)]) continue;
            if (key < best) best = key;
        }
        if (best != ~0ull && key_dist(best) <= TH_LOW) {
            pt_match[i] = key_idx(best);
            if (mode == 0) matched[key_idx(best)] = 1;
            nm++;
        }
    }
    *nmatches = nm;
    return ORBFE_OK;
}

extern "C" int orbfe_search_by_projection_sim3(orbfe_context *ctx, const orbfe_frame_view *kf, const float *Scw, int n_pts,
                                               const float *pos, const float *normal, const float *max_distance, const float *min_distance,
                                               const uint8_t *pt_desc, const int32_t *pt_valid, const uint8_t *kf_matched, float th,
                                               int32_t *pt_match, int *nmatches)
try {
    ORBFE_ENTRY(ctx);
    return sim3_projection_impl(ctx, 0, kf, Scw, n_pts, pos, normal, max_distance, min_distance, pt_desc, pt_valid, kf_matched, th, pt_match, nmatches);
} ORBFE_CATCH(ctx)

extern "C" int orbfe_fuse_sim3(orbfe_context *ctx, const orbfe_frame_view *kf, const float *Scw, int n_pts,
                               const float *pos, const float *normal, const float *max_distance, const float *min_distance,
                               const uint8_t *pt_desc, const int32_t *pt_valid, float th, int32_t *best_idx, int *n_fused)
try {
    ORBFE_ENTRY(ctx);
    return sim3_projection_impl(ctx, 1, kf, Scw, n_pts, pos, normal, max_distance, min_distance, pt_desc, pt_valid, nullptr, th, best_idx, n_fused);
} ORBFE_CATCH(ctx)

// One direction of ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1143-1216 / 1218-1291): keyframe A's map points moved into
// camera B by [sR|t], one window query each against B; match[i] = keypoint of B or -1 (the points do not interact).
static int sim3_one_way(orbfe_context *ctx, const float *Taw, const float *sRt, const orbfe_frame_view *kfb, int n,
                        const float *pos, const float *max_distance, const float *min_distance, const uint8_t *desc,
                        const int32_t *valid, float th, std::vector<int32_t> &match)
{
    const orbfe_params *P = orbfe_ctx_params(ctx);
    const float *sf = orbfe_ctx_scale_factors(ctx);
    const float log_sf = logf((float)(double)P->scale_factor);
    std::vector<MatchQuery> q(n);
    std::vector<uint8_t> qd((size_t)32 * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        MatchQuery &Q = q[i];
        Q = MatchQuery{0, 0, 0, 0, -1, 0, 0, 0};
        if (!valid[i]) continue;
        float pa[3], pb[3];
        rt_apply(Taw, pos + 3 * i, pa);
        rt_apply(sRt, pa, pb);
        if ((double)pb[2] < 0.0) continue;
        const float invz = (float)(1.0 / (double)pb[2]);
        const float x = pb[0] * invz, y = pb[1] * invz;
        const float u = P->fx * x + P->cx, v = P->fy * y + P->cy;
        if (!kf_is_in_image(kfb, u, v)) continue;
        const float dist = (float)sqrt((double)pb[0] * pb[0] + (double)pb[1] * pb[1] + (double)pb[2] * pb[2]);
        if (dist < 0.8f * min_distance[i] || dist > 1.2f * max_distance[i]) continue;
        const int lvl = predict_scale(max_distance[i], dist, log_sf, P->nlevels);
        Q.u = u; Q.v = v; Q.r = th * sf[lvl]; Q.min_level = lvl - 1; Q.max_level = lvl; Q.flags = 1;
        memcpy(&qd[(size_t)32 * i], desc + (size_t)32 * i, 32);
    }
    int rc = run_window_queries(ctx, kfb, q, qd);
    if (rc != ORBFE_OK) return rc;
    orbfe_match_state *st = match_state(ctx);
    match.assign(n > 0 ? n : 1, -1);
    for (int i = 0; i < n; i++) {
        unsigned long long best = ~0ull;
        for (int k = 0; k < st->h_cnt[i]; k++) {
            const unsigned long long key = st->h_list[st->h_off[i] + k];
            if (key < best) best = key;
        }
        if (best != ~0ull && key_dist(best) <= TH_HIGH) match[i] = key_idx(best);
    }
    return ORBFE_OK;
}

// ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th), src/ORBmatcher.cc:1098-1322
extern "C" int orbfe_search_by_sim3(orbfe_context *ctx,
                                    const orbfe_frame_view *kf1, const float *T1w, const float *pos1, const float *max_distance1,
                                    const float *min_distance1, const uint8_t *pt_desc1, const int32_t *valid1,
                                    const orbfe_frame_view *kf2, const float *T2w, const float *pos2, const float *max_distance2,
                                    const float *min_distance2, const uint8_t *pt_desc2, const int32_t *valid2,
                                    float s12, const float *R12, const float *t12, float th, int32_t *match12, int *n_found)
try {
    ORBFE_ENTRY(ctx);
    int rc = check_view(ctx, kf1);
    if (rc != ORBFE_OK) return rc;
    rc = check_view(ctx, kf2);
    if (rc != ORBFE_OK) return rc;
    const int N1 = kf1->n, N2 = kf2->n;
    if (!T1w || !T2w || !R12 || !t12 || !n_found || (N1 > 0 && (!pos1 || !max_distance1 || !min_distance1 || !pt_desc1 || !valid1 || !match12)) ||
        (N2 > 0 && (!pos2 || !max_distance2 || !min_distance2 || !pt_desc2 || !valid2)))
        return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    // sR12 = s12 * R12; sR21 = (1.0 / s12) * R12.t(); t21 = -sR21 * t12 (:1116-1119)
    float A12[12], A21[12];
    const float inv_s = (float)(1.0 / (double)s12);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { A12[4 * i + j] = R12[3 * i + j] * s12; A21[4 * i + j] = R12[3 * j + i] * inv_s; }
    for (int i = 0; i < 3; i++) {
        A12[4 * i + 3] = t12[i];
        const float t0 = (A21[4 * i] * t12[0] + A21[4 * i + 1] * t12[1]) + A21[4 * i + 2] * t12[2];
        A21[4 * i + 3] = -t0;
    }
    std::vector<int32_t> m1, m2;
    rc = sim3_one_way(ctx, T1w, A21, kf2, N1, pos1, max_distance1, min_distance1, pt_desc1, valid1, th, m1);
    if (rc != ORBFE_OK) return rc;
    rc = sim3_one_way(ctx, T2w, A12, kf1, N2, pos2, max_distance2, min_distance2, pt_desc2, valid2, th, m2);
    if (rc != ORBFE_OK) return rc;
    int nf = 0;
    for (int i1 = 0; i1 < N1; i1++) { // check agreement (:1293-1308)
        match12[i1] = -1;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && m2[idx2] == i1) { match12[i1] = idx2; nf++; }
    }
    *n_found = nf;
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:400-515
extern "C" int orbfe_search_for_initialization(orbfe_context *ctx, const orbfe_frame_view *f1, const orbfe_frame_view *f2,
                                               float *prev_matched, int window_size, float nnratio, int check_ori,
                                               int32_t *matches12, int *nmatches)
try {
    ORBFE_ENTRY(ctx);
    int rc = check_view(ctx, f2);
    if (rc != ORBFE_OK) return rc;
    if (!f1 || f1->n < 0 || (f1->n > 0 && (!f1->keys_un || !f1->descriptors || !prev_matched)) || !matches12 || !nmatches)
        return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    const int n1 = f1->n, n2 = f2->n;
    std::vector<MatchQuery> q;
    std::vector<uint8_t> qd;
    orbfe_resolve::build_queries_initialization(n1, f1->keys_un, f1->descriptors, prev_matched, window_size, q, qd);
    TopkRequest req; // no static filter: the stealing rule (vMatchedDistance) is dynamic
    rc = run_window_queries(ctx, f2, q, qd, &req);
    if (rc != ORBFE_OK) return rc;
    orbfe_match_state *st = match_state(ctx);
    FullListCtx fl{ctx, st};
    orbfe_resolve::CandidateSource src = topk_source(st, &fl, req);
    static const orbfe_keypoint none = {};
    *nmatches = orbfe_resolve::resolve_initialization(src, n1, n2, n1 > 0 ? &f1->keys_un[0].angle : &none.angle, sizeof(orbfe_keypoint),
                                                      n2 > 0 ? &f2->keys_un[0].angle : &none.angle, sizeof(orbfe_keypoint),
                                                      n2 > 0 ? &f2->keys_un[0].x : &none.x, nnratio, check_ori, prev_matched, matches12);
    if (src.error) return orbfe_fail(ctx, ORBFE_ERR_HIP, "candidate list download failed");
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

orbfe_match_state *orbfe_match_state_create() { return new (std::nothrow) orbfe_match_state(); }
void orbfe_match_state_destroy(orbfe_match_state *s) { delete s; }
