// orbfe_api.hip -- host side of the C ABI declared in include/orbfe.h.
//
// Builds the level / cell / quota tables exactly as ORBextractor::ORBextractor and
// ComputePyramid do (reference src/ORBextractor.cc:405-464,921-946), owns the HBM
// buffers of one batch of images, and enqueues the kernels of the stage files (orbfe_pyramid / fast / octree* / describe / stereo .hip).
// There is no CPU compute path here: without a HIP device orbfe_create fails.
#include "../../include/orbfe.h"
#include "orbfe_device.h"
#include "orbfe_host.h"

#include <algorithm>
#include <chrono>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

size_t orbfe_octree_lds_bytes(const DeviceConfig &cfg);
// bit_pattern_31_ (src/ORBextractor.cc:145-403, re-emitted by tools/extract_pattern.py): the default of every context's pattern copy
static const int8_t k_bit_pattern_31[1024] = {
#include "orb_pattern_31.inc"
};
#define ORBFE_MAX_GROUPS 8

struct orbfe_context {
    orbfe_params params;
    DeviceConfig cfg;
    DeviceBuffers buf;
    hipStream_t stream = nullptr;
    uint8_t *d_in = nullptr;      // staging for host-image entry points [max_images][w*h]
    float *d_depth_in = nullptr;  // staging for RGB-D depth
    // pinned host staging of the single-frame entry points (lazily allocated): packed input rows, then one block of
    // outputs per call so a frame costs one stream synchronisation instead of one blocking copy per array
    uint8_t *h_in = nullptr;      // [min(max_images,2)][w*h]
    float *h_depth_in = nullptr;  // [w*h]
    uint8_t *h_out = nullptr;     // see HostOut
    uint8_t *d_pack = nullptr;    // device staging of orbfe_fetch_batch_packed (lazily allocated for max_images)
    size_t d_pack_bytes = 0;
    hipEvent_t ev_pack = nullptr; // recorded behind the staging's device-to-host copy: the next packed fetch (whatever its stream) waits for it before it refills d_pack
    bool ev_pack_set = false;
    uint8_t *d_ham = nullptr;     // scratch for orbfe_hamming_matrix
    void *d_und = nullptr;        // scratch for the undistortion entry points
    size_t d_und_bytes = 0;
    size_t d_ham_bytes = 0;
    int last_images = 0;
    unsigned epoch = 0;       // extraction calls enqueued so far (device-resident frame caches key on it)
    std::vector<int> slot_cnt;    // keypoint counts of the slots of call `slot_cnt_epoch` (host copy, filled by the first fetch)
    unsigned slot_cnt_epoch = ~0u;
    bool fuse_blur = true;    // blur level l - 1 in the launch that resizes it into level l (ORBFE_NO_FUSE=1: separate launches)
    // The blur of level l only needs level l, is memory-bound and is first read by describe_kernel: levels >= blur_ride_from are
    // blurred by workgroups that ride in FAST's launch (issue-bound) instead of beside the resize that reads the level, for
    // batches of at least blur_ride_min_images images (smaller batches: whatever the pyramid launches leave unblurred rides).
    // Round 5, 64 pairs: every level riding (0) takes the pyramid's launches from 167 to 99 us and FAST's from 266 to 320 (+ 2 %).
    int blur_ride_from = 0, blur_ride_min_images = 64;
    // Level 0 read in place from the caller's packed CV_8UC1 images (no ingest launch, no copy): possible when level 1 is resized
    // by the LDS-free kernel and nothing stages level 0 through pyr_tail_kernel; ORBFE_NO_INPLACE=1 keeps the copy (A/B, tests).
    // Colour / rectified input always goes through ingest (it computes level 0).
    bool inplace_ok = false;
    const uint8_t *last_src = nullptr; // images of the latest enqueue when it ran in place (orbfe_fetch_pyramid's level 0), else null
    bool last_src_owned = false;       // ... and they live in the library's own staging (d_in: the host entry points), which outlives the call
    bool input_retained = false;       // orbfe_set_input_retained: the caller keeps the images of an enqueue call valid until its next call
    bool use_octree3 = false; // bucket-pyramid quadtree (orbfe_octree3.hip); preferred when its limits hold
    size_t ot3_lds = 0;
    bool ot3_nodes_in_hbm = false; // node tables of the bucket-pyramid quadtree in HBM scratch (large per-level quotas)
    int ot_sort_cap = 0;      // power of two >= max_nodes: the quadtree kernels' sort buffer
    // stage timing: ring of PROF_RING calls x (ORBFE_NUM_STAGES + 1) events
    bool profiling = false;
    int prof_every = 1;      // record events on every prof_every-th enqueue call only (orbfe_set_profiling_interval)
    unsigned prof_seq = 0;   // enqueue calls seen while profiling
    bool prof_now = false;   // the current call records
    int prof_only = -1; // >= 0: record only the two events around that stage
    std::vector<hipEvent_t> events;
    int prof_calls = 0;      // calls recorded since the last reset
    int prof_stages[64];     // number of stages recorded by each call in the ring
    int prof_groups = 1;
    // Recorded on the stream of every enqueue call when its work has been queued: what "the latest extraction" means to the
    // blocking fetches and to the matchers on the resident frame.  The caller's stream handle itself is NOT kept -- a caller
    // may enqueue, synchronise and destroy its stream before it fetches.
    hipEvent_t ev_latest = nullptr;
    bool latest_foreign = false; // the latest call ran on a caller's stream
    std::recursive_mutex mu;
    // stream groups (orbfe_set_streams)
    int groups = 1;
    hipStream_t gstreams[ORBFE_MAX_GROUPS] = {};
    hipEvent_t ev_fork = nullptr, ev_join[ORBFE_MAX_GROUPS] = {};
    float scale[ORBFE_MAX_LEVELS], inv_scale[ORBFE_MAX_LEVELS], sigma2[ORBFE_MAX_LEVELS], inv_sigma2[ORBFE_MAX_LEVELS];
    int32_t feats[ORBFE_MAX_LEVELS];
    int8_t pattern[1024];     // host copy of DeviceBuffers::pattern
    std::vector<void *> allocs;
    orbfe_match_state *match = nullptr;
    orbfe_bow_state *bow = nullptr;
    orbfe_pose_state *pose = nullptr;
    char err[512];
};

static thread_local char g_err[512] = "";

static int fail(orbfe_context *ctx, int code, const char *fmt, ...)
{
    char msg[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(msg, sizeof(msg), fmt, ap);
    va_end(ap);
    snprintf(g_err, sizeof(g_err), "%s", msg);
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s", msg);
    return code;
}

int orbfe_fail(orbfe_context *ctx, int code, const char *fmt, ...)
{
    char msg[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(msg, sizeof(msg), fmt, ap);
    va_end(ap);
    return fail(ctx, code, "%s", msg);
}
orbfe_match_state *orbfe_ctx_match_state(orbfe_context *ctx)
{
    if (!ctx->match) ctx->match = orbfe_match_state_create();
    return ctx->match;
}
hipStream_t orbfe_ctx_stream(orbfe_context *ctx) { return ctx->stream; }
int orbfe_ctx_device(const orbfe_context *ctx) { return ctx->params.device; }
const orbfe_params *orbfe_ctx_params(const orbfe_context *ctx) { return &ctx->params; }
const float *orbfe_ctx_scale_factors(const orbfe_context *ctx) { return ctx->scale; }
const float *orbfe_ctx_inv_sigma2(const orbfe_context *ctx) { return ctx->inv_sigma2; }
const DeviceConfig *orbfe_ctx_config(const orbfe_context *ctx) { return &ctx->cfg; }
const DeviceBuffers *orbfe_ctx_buffers(const orbfe_context *ctx) { return &ctx->buf; }
unsigned orbfe_ctx_epoch(const orbfe_context *ctx) { return ctx->epoch; }
int orbfe_ctx_wait_foreign_stream(orbfe_context *ctx)
{
    if (ctx->latest_foreign && hipStreamWaitEvent(ctx->stream, ctx->ev_latest, 0) != hipSuccess)
        return fail(ctx, ORBFE_ERR_HIP, "hipStreamWaitEvent on the latest extraction failed");
    return ORBFE_OK;
}
std::recursive_mutex &orbfe_ctx_mutex(orbfe_context *ctx) { return ctx->mu; }
// Keypoint count of image slot `slot` of the latest extraction call: from the host copy the frame entry points leave behind,
// else one blocking read of the counters per call (batched calls whose results were not fetched yet).
int orbfe_ctx_slot_count(orbfe_context *ctx, int slot, int *cnt)
{
    if (slot < 0 || slot >= ctx->last_images)
        return fail(ctx, ORBFE_ERR_INVALID, "device slot %d: the latest extraction call filled %d image slots", slot, ctx->last_images);
    if (ctx->slot_cnt_epoch != ctx->epoch || (int)ctx->slot_cnt.size() < ctx->last_images) {
        if (ctx->latest_foreign && hipEventSynchronize(ctx->ev_latest) != hipSuccess) return fail(ctx, ORBFE_ERR_HIP, "hipEventSynchronize failed");
        ctx->slot_cnt.resize(ctx->last_images);
        if (hipMemcpyAsync(ctx->slot_cnt.data(), ctx->buf.kp_cnt, sizeof(int) * ctx->last_images, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            return fail(ctx, ORBFE_ERR_HIP, "reading the keypoint counters failed");
        ctx->slot_cnt_epoch = ctx->epoch;
    }
    *cnt = ctx->slot_cnt[slot];
    return ORBFE_OK;
}
orbfe_pose_state *orbfe_ctx_pose_state(orbfe_context *ctx)
{
    if (!ctx->pose) ctx->pose = orbfe_pose_state_create();
    return ctx->pose;
}
orbfe_bow_state *orbfe_ctx_bow_state(orbfe_context *ctx)
{
    if (!ctx->bow) ctx->bow = orbfe_bow_state_create();
    return ctx->bow;
}

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(ctx, ORBFE_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" int orbfe_abi_version(void) { return ORBFE_ABI_VERSION; }
#include "build/orbfe_build_id.h" // generated by the Makefile: sha256 over the library's sources and compile flags
#ifdef ORBFE_PROFILE_CUTS
extern "C" const char *orbfe_build_id(void) { return ORBFE_BUILD_ID "+cuts"; }
#else
extern "C" const char *orbfe_build_id(void) { return ORBFE_BUILD_ID; }
#endif
extern "C" const char *orbfe_last_error(const orbfe_context *ctx) { return ctx ? ctx->err : g_err; }

static int cv_round_f(float v) { return (int)lrintf(v); }

// OpenCV 4.5.5 getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED (see oracle for provenance)
static void gaussian_taps_q8(int ksize, double sigma, int *taps)
{
    double g[64];
    double scale2x = -0.5 / (sigma * sigma);
    double sum = 0.0;
    for (int i = 0; i < ksize; i++) {
        double x = (double)i - (double)(ksize - 1) * 0.5;
        g[i] = exp(scale2x * x * x);
        sum += g[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < ksize; i++) g[i] *= sum;
    int n2 = ksize / 2;
    double err = 0.0;
    long acc = 0;
    for (int i = 0; i < n2; i++) {
        double adj = g[i] * 256.0 + err;
        long v0 = lrint(adj);
        err = adj - (double)v0;
        taps[i] = (int)v0;
        taps[ksize - 1 - i] = (int)v0;
        acc += 2 * v0;
    }
    taps[n2] = (int)(256 - acc);
}

template <typename T>
static int dev_alloc(orbfe_context *ctx, T **p, size_t count)
{
    void *q = nullptr;
    size_t bytes = sizeof(T) * (count ? count : 1);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail(ctx, ORBFE_ERR_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    ctx->allocs.push_back(q);
    *p = (T *)q;
    return ORBFE_OK;
}

// Tables of ORBextractor::ORBextractor (src/ORBextractor.cc:405-464) and the per-level
// geometry of ComputePyramid / ComputeKeyPointsOctTree / DistributeOctTree.
static int build_config(orbfe_context *ctx)
{
    const orbfe_params &p = ctx->params;
    DeviceConfig &c = ctx->cfg;
    memset(&c, 0, sizeof(c));
    c.nlevels = p.nlevels;
    c.width = p.width; c.height = p.height;
    c.edge_threshold = p.edge_threshold;
    c.min_border = p.edge_threshold - 3;
    c.ini_th = p.ini_th_fast; c.min_th = p.min_th_fast;
    {   // integers 0 .. 255 as IEEE half precision (exact): 0 -> 0, else exponent e = floor(log2 t) biased by 15, mantissa t's bits below the leading one
        auto half_bits = [](int t) -> uint32_t {
            if (t <= 0) return 0u;
            int e = 0;
            while ((t >> (e + 1)) != 0) e++;
            return (uint32_t)(((e + 15) << 10) | (((t << (10 - e)) & 0x3ff)));
        };
        c.ini_th_h2 = half_bits(c.ini_th) * 0x10001u; c.min_th_h2 = half_bits(c.min_th) * 0x10001u;
    }
    c.half_patch = p.half_patch_size;
    c.bf = p.bf; c.fx = p.fx;
    c.mb = p.fx != 0.f ? p.bf / p.fx : 0.f; // mb := mbf / fx
    c.in_cn = 1; c.in_coef[0] = c.in_coef[1] = c.in_coef[2] = 0; c.in_shift = 15;
    c.in_image_bytes = (size_t)p.width * p.height;
    c.rm_on = 0; c.rm_sw = c.rm_sh = 0; c.rm_xy[0] = c.rm_xy[1] = nullptr; c.rm_a[0] = c.rm_a[1] = nullptr;
    c.n_dist = 0; for (int i = 0; i < 5; i++) c.dist[i] = 0.f;
    c.cam[0] = p.fx; c.cam[1] = p.fy; c.cam[2] = p.cx; c.cam[3] = p.cy;

    const double sf_d = (double)p.scale_factor; // member is double, initialised from float
    ctx->scale[0] = 1.0f; ctx->sigma2[0] = 1.0f;
    for (int i = 1; i < p.nlevels; i++) {
        ctx->scale[i] = (float)((double)ctx->scale[i - 1] * sf_d);
        ctx->sigma2[i] = ctx->scale[i] * ctx->scale[i];
    }
    for (int i = 0; i < p.nlevels; i++) {
        ctx->inv_scale[i] = 1.0f / ctx->scale[i];
        ctx->inv_sigma2[i] = 1.0f / ctx->sigma2[i];
    }
    const float factor = (float)(1.0 / sf_d);
    float n_desired = (float)p.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)p.nlevels));
    int sum = 0;
    for (int l = 0; l < p.nlevels - 1; l++) {
        ctx->feats[l] = cv_round_f(n_desired);
        sum += ctx->feats[l];
        n_desired *= factor;
    }
    ctx->feats[p.nlevels - 1] = p.nfeatures - sum > 0 ? p.nfeatures - sum : 0;

    const int hp = p.half_patch_size;
    const int vmax = (int)floor((double)((float)hp * sqrtf(2.f) / 2 + 1));
    const int vmin = (int)ceil((double)((float)hp * sqrtf(2.f) / 2));
    const double hp2 = (double)(hp * hp);
    for (int v = 0; v <= vmax; ++v) c.umax[v] = (int)lrint(sqrt(hp2 - (double)(v * v)));
    for (int v = hp, v0 = 0; v >= vmin; --v) {
        while (c.umax[v0] == c.umax[v0 + 1]) ++v0;
        c.umax[v] = v0;
        ++v0;
    }
    gaussian_taps_q8(7, 2.0, c.taps);

    size_t pyr_off = 0, blur_off = 0;
    int cell_off = 0, cand_off = 0, sel_off = 0, tile_off = 0, cell_cap = 1, max_nodes = 8;
    for (int l = 0; l < p.nlevels; l++) {
        LevelInfo &L = c.lv[l];
        L.scale = ctx->scale[l]; L.inv_scale = ctx->inv_scale[l];
        L.w = cv_round_f((float)p.width * L.inv_scale);
        L.h = cv_round_f((float)p.height * L.inv_scale);
        if (L.w < 1 || L.h < 1) return fail(ctx, ORBFE_ERR_UNSUPPORTED, "level %d is empty (%dx%d)", l, L.w, L.h);
        // reflect-101 margin: 4 px left, >= 12 px right, 3 rows above/below (see orbfe_pyramid.hip)
        L.pitch = (L.w + 16 + 63) & ~63;
        L.pyr_off = (int)(pyr_off + (size_t)3 * L.pitch + 4);
        pyr_off += ((size_t)L.pitch * (L.h + 6) + 255) & ~(size_t)255;
        // blurred copy: 32 x 4 px tiles (one 128-B line each), only the image itself (describe never leaves it by more than
        // the 2 px its aligned 40-byte patch rows overshoot: one spare tile column)
        L.blur_off = (int)blur_off;
        L.blur_tx = (L.w + 31) / 32 + 1;
        blur_off += (size_t)L.blur_tx * ((L.h + 3) / 4 + 1) * 128; // + one tile row: describe stages 40 rows from a multiple of 4 (up to row h + 1; loaded, never used)
        if (l > 0) {
            L.rs_scale_x = 1.0 / ((double)L.w / (double)c.lv[l - 1].w);
            L.rs_scale_y = 1.0 / ((double)L.h / (double)c.lv[l - 1].h);
        }
        L.scaled_patch = (int)((float)p.patch_size * L.scale);
        L.quota = ctx->feats[l];
        const int min_b = c.min_border;
        const int max_bx = L.w - p.edge_threshold + 3, max_by = L.h - p.edge_threshold + 3;
        const float width = (float)(max_bx - min_b), height = (float)(max_by - min_b);
        L.n_cols = (int)(width / 30.f);
        L.n_rows = (int)(height / 30.f);
        L.cell_off = cell_off; L.cand_off = cand_off;
        int cand_cap = 0;
        if (L.n_cols >= 1 && L.n_rows >= 1 && max_bx > min_b && max_by > min_b) {
            L.w_cell = (int)ceilf(width / (float)L.n_cols);
            L.h_cell = (int)ceilf(height / (float)L.n_rows);
            L.n_cells = L.n_cols * L.n_rows;
            for (int i = 0; i < L.n_rows; i++) {
                const int ini_y = min_b + i * L.h_cell;
                int my = ini_y + L.h_cell + 6;
                if (ini_y >= max_by - 3) continue;
                if (my > max_by) my = max_by;
                for (int j = 0; j < L.n_cols; j++) {
                    const int ini_x = min_b + j * L.w_cell;
                    int mx = ini_x + L.w_cell + 6;
                    if (ini_x >= max_bx - 6) continue;
                    if (mx > max_bx) mx = max_bx;
                    const int iw = mx - ini_x - 6, ih = my - ini_y - 6;
                    if (iw <= 0 || ih <= 0) continue;
                    const int cc = ((iw + 1) / 2) * ((ih + 1) / 2); // 3x3 strict NMS survivors bound
                    cand_cap += cc;
                    if (cc > cell_cap) cell_cap = cc;
                }
            }
            L.n_ini = (int)roundf(width / height);
            if (L.n_ini < 1) L.n_ini = 1; // reference would index an empty vector; documented guard
            L.hx = width / (float)L.n_ini;
        } else {
            L.n_cols = L.n_rows = 0; L.w_cell = L.h_cell = 0; L.n_cells = 0;
            L.n_ini = 1; L.hx = 1.f;
        }
        L.cand_cap = cand_cap;
        cell_off += L.n_cells;
        cand_off += (cand_cap + 3) & ~3;
        L.sel_off = sel_off;
        L.sel_cap = (L.quota + 3 > 4 * L.n_ini ? L.quota + 3 : 4 * L.n_ini) + 1;
        sel_off += L.sel_cap;
        if (L.sel_cap + 1 > max_nodes) max_nodes = L.sel_cap + 1;
        L.blur_tile_off = tile_off;
        L.blur_tiles_x = (L.w + 255) / 256;
        L.blur_tiles_y = (L.h + ORBFE_BLUR_ROWS - 1) / ORBFE_BLUR_ROWS;
        tile_off += L.blur_tiles_x * L.blur_tiles_y;
    }
    c.pyr_bytes = pyr_off;
    c.blur_bytes = blur_off + 256;
    c.cells_total = cell_off > 0 ? cell_off : 1;
    c.cand_total = cand_off > 0 ? cand_off : 4;
    c.sel_total = sel_off;
    c.cell_cap = cell_cap;
    c.blur_tiles_total = tile_off;
    c.max_nodes = max_nodes;
    if (p.width > 32767 || p.height > 32767) return fail(ctx, ORBFE_ERR_UNSUPPORTED, "image larger than 32767 px");
    if (c.sel_total > 65535) return fail(ctx, ORBFE_ERR_UNSUPPORTED, "nfeatures too large (keypoint capacity %d > 65535)", c.sel_total);
    {
        int sc = 1;
        while (sc < c.max_nodes) sc <<= 1;
        bool roots_ok = true;
        for (int l = 0; l < p.nlevels; l++) roots_ok = roots_ok && c.lv[l].n_ini <= 4;
        int max_cand = 0;
        for (int l = 0; l < p.nlevels; l++) { roots_ok = roots_ok && c.lv[l].cand_cap <= (1 << 20); max_cand = c.lv[l].cand_cap > max_cand ? c.lv[l].cand_cap : max_cand; }
        ctx->ot_sort_cap = sc;
        {
            bool ok3 = roots_ok && c.cell_cap <= 4095; // key fields: 12-bit cell, 12-bit slot; bucket partials: 12-bit count
            for (int l = 0; l < p.nlevels; l++) ok3 = ok3 && c.lv[l].n_cells <= 4096 && c.lv[l].w_cell <= 64 && c.lv[l].h_cell <= 64; // one lane per cell column / row
            // the counts of the two deepest pyramid depths are 16 bit: a depth-4 node (1 / 256 of a root) holds at most one strict-NMS
            // survivor per 2 x 2 px (always true within the cell-count limit above: <= 4096 cells of <= 64 x 64 px per level)
            for (int l = 0; l < p.nlevels; l++) {
                const long rw = (long)(c.lv[l].w - p.edge_threshold + 3) - c.min_border, rh = (long)(c.lv[l].h - p.edge_threshold + 3) - c.min_border;
                if (rw > 0 && rh > 0) ok3 = ok3 && ((rw / c.lv[l].n_ini / 16 + 2) / 2 + 1) * ((rh / 16 + 2) / 2 + 1) <= 65535;
            }
            ctx->ot3_nodes_in_hbm = orbfe_octree3_lds_bytes(c.max_nodes, sc, false) > 150 * 1024;
            ctx->ot3_lds = orbfe_octree3_lds_bytes(c.max_nodes, sc, ctx->ot3_nodes_in_hbm);
            const char *force = getenv("ORBFE_OCTREE"); // test knob: 1 = the generic node-parallel kernel (the fallback beyond the bucket-pyramid kernel's limits)
            ctx->use_octree3 = ok3 && ctx->ot3_lds <= 150 * 1024 && !(force && atoi(force) == 1);
        }
        if (!ctx->use_octree3 && orbfe_octree_lds_bytes(c) > 150 * 1024)
            return fail(ctx, ORBFE_ERR_UNSUPPORTED, "nfeatures too large for the quadtree LDS budget");
    }
    return ORBFE_OK;
}

extern "C" int orbfe_create(const orbfe_params *params, orbfe_context **out)
try {
    if (!params || !out) return fail(nullptr, ORBFE_ERR_INVALID, "null argument");
    *out = nullptr;
    const orbfe_params &p = *params;
    if (p.nlevels < 1 || p.nlevels > ORBFE_MAX_LEVELS || p.nfeatures < 1 || !(p.scale_factor > 1.0f) ||
        p.half_patch_size < 1 || p.half_patch_size > 62 || p.edge_threshold < p.half_patch_size + 4 || p.edge_threshold < 19 ||
        p.width < 1 || p.height < 1 || p.max_images < 1 || p.min_th_fast < 1 || p.ini_th_fast < p.min_th_fast ||
        p.ini_th_fast > 254)
        return fail(nullptr, ORBFE_ERR_INVALID, "invalid orbfe_params"); // edge_threshold >= 19: the rotated test pattern reaches 18 px from a keypoint (describe_kernel stages +-18)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, ORBFE_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (p.device < 0 || p.device >= ndev) return fail(nullptr, ORBFE_ERR_INVALID, "device %d out of range (%d)", p.device, ndev);
    orbfe_context *ctx = new (std::nothrow) orbfe_context();
    if (!ctx) return fail(nullptr, ORBFE_ERR_INVALID, "out of host memory");
    ctx->err[0] = 0;
    ctx->params = p;
    int rc = build_config(ctx);
    if (rc != ORBFE_OK) { snprintf(g_err, sizeof(g_err), "%s", ctx->err); delete ctx; return rc; }
    // the XCD-aware block maps divide jb = blockIdx.x / 8 through a float reciprocal that is exact for jb < 2^21 (small_div,
    // orbfe_common.hpp), i.e. below 2^24 workgroups per launch; xcd_grid may round a launch up to twice blocks x images, so
    // the limit on blocks x images is 2^23
    if (((size_t)ctx->cfg.cells_total / 4 + 1 + (size_t)ctx->cfg.blur_tiles_total / 4 + 1) * (size_t)p.max_images >= ((size_t)1 << 23) || // FAST's launch may carry blur tiles too

        ((size_t)ctx->cfg.sel_total / 4 + 1) * (size_t)p.max_images >= ((size_t)1 << 23)) {
        delete ctx;
        return fail(nullptr, ORBFE_ERR_CAPACITY, "max_images %d: more than 2^23 workgroups per launch", p.max_images);
    }
    if (hipSetDevice(p.device) != hipSuccess) { delete ctx; return fail(nullptr, ORBFE_ERR_NO_DEVICE, "hipSetDevice failed"); }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return fail(nullptr, ORBFE_ERR_NO_DEVICE, "hipStreamCreate failed"); }
    if (hipEventCreateWithFlags(&ctx->ev_latest, hipEventDisableTiming) != hipSuccess) { orbfe_destroy(ctx); return fail(nullptr, ORBFE_ERR_NO_DEVICE, "hipEventCreate failed"); }
    if (ctx->use_octree3 && orbfe_octree3_prepare(ctx->ot3_lds, ctx->ot3_nodes_in_hbm) != 0) ctx->use_octree3 = false;
    const DeviceConfig &c = ctx->cfg;
    const size_t B = (size_t)p.max_images;
    DeviceBuffers &b = ctx->buf;
    KeyPointPOD *kps = nullptr;
#define A(ptr, count) do { rc = dev_alloc(ctx, &(ptr), (count)); if (rc != ORBFE_OK) { orbfe_destroy(ctx); return rc; } } while (0)
#define Z(ptr, bytes) do { if (hipMemset((ptr), 0, (bytes)) != hipSuccess) { orbfe_destroy(ctx); return fail(nullptr, ORBFE_ERR_HIP, "hipMemset failed"); } } while (0)
    A(b.pyr, B * c.pyr_bytes);
    A(b.blur, B * c.blur_bytes);
    A(b.cell_cnt, B * c.cells_total);
    A(b.cell_xy, B * c.cells_total * c.cell_cap);
    A(b.cell_sc, B * c.cells_total * c.cell_cap);
    A(b.cell_base, B * c.cells_total);
    A(b.cand_xy, B * c.cand_total);
    A(b.cand_sc, B * c.cand_total);
    A(b.ot_xy2, B * c.cand_total);
    A(b.idx0, B * c.cand_total);
    A(b.idx1, B * c.cand_total);
    A(b.bk_end, B * c.nlevels * 4097);
    b.bk_best = nullptr;
    if (ctx->use_octree3) A(b.bk_best, B * c.nlevels * ORBFE_BK_PYR);
    b.ot3_scratch = nullptr;
    if (ctx->use_octree3 && ctx->ot3_nodes_in_hbm) A(b.ot3_scratch, B * c.nlevels * orbfe_octree3_node_bytes(c.max_nodes, ctx->ot_sort_cap));
    A(b.lvl_ncand, B * c.nlevels);
    A(b.sel_cnt, B * c.nlevels);
    A(b.sel_xy, B * c.sel_total + 4); // + 4: stereo_rowlist_kernel reads whole quads of slots
    A(b.sel_sc, B * c.sel_total);
    A(b.proc_xy, B * c.sel_total);
    A(b.proc_meta, B * c.sel_total);
    Z(b.proc_xy, B * c.sel_total * sizeof(uint32_t));
    Z(b.proc_meta, B * c.sel_total * sizeof(uint32_t));
    A(kps, B * c.sel_total);
    b.kps = kps;
    A(b.desc, B * c.sel_total * 32);
    A(b.kp_cnt, B);
    A(b.u_right, B * c.sel_total);
    A(b.depth, B * c.sel_total);
    A(b.sad, B * c.sel_total);
    A(b.status, B);
    A(ctx->d_in, B * (size_t)p.width * p.height);
    A(b.dbg_ts, 4096);
    Z(b.dbg_ts, 4096 * sizeof(long long));
    {   // stereo row lists (vRowIndices, src/Frame.cc:474-491): fixed capacity per row, ~4x the mean occupancy
        // (a right keypoint is listed in ~2 * 2 * scale + 1 rows); a fuller row makes stereo_match_kernel scan all keypoints
        int cap = (int)(4.0 * c.sel_total * 10.0 / p.height);
        cap = cap < 64 ? 64 : cap;
        cap = cap > c.sel_total ? c.sel_total : cap;
        ctx->cfg.row_cap = cap;
        const size_t pairs = (B + 1) / 2;
        A(b.row_cnt, pairs * (size_t)p.height);
        A(b.row_ent, pairs * (size_t)p.height * cap);
        Z(b.row_cnt, pairs * (size_t)p.height * sizeof(int));
    }
    {   // cv::resize tables (resize.cpp: xofs/ialpha, yofs/ibeta) over the margin-extended domain of each level
        std::vector<uint32_t> tab;
        auto reflect = [](int q, int len) { if (len == 1) return 0; while (q < 0 || q >= len) q = q < 0 ? -q : 2 * len - 2 - q; return q; };
        for (int l = 1; l < p.nlevels; l++) {
            LevelInfo &D = ctx->cfg.lv[l];
            const LevelInfo &S = ctx->cfg.lv[l - 1];
            const int nx = (D.w + 12 + 3) & ~3, ny = D.h + 6;
            while (tab.size() % 4) tab.push_back(0);
            D.rs_xtab_off = (int)tab.size(); D.rs_xtab_n = nx;
            tab.resize(tab.size() + 2 * (size_t)nx, 0);
            for (int i = 0; i < nx; i++) {
                const int dx = reflect(i - 4, D.w);
                float fx = (float)(((double)dx + 0.5) * D.rs_scale_x - 0.5);
                int sx = (int)floorf(fx);
                fx -= (float)sx;
                if (sx < 0) { fx = 0.f; sx = 0; }
                if (sx >= S.w - 1) { fx = 0.f; sx = S.w - 1; }
                const int a0 = (int)lrintf((1.f - fx) * 2048.f), a1 = (int)lrintf(fx * 2048.f);
                const int sx1 = sx + 1 < S.w ? sx + 1 : S.w - 1;
                tab[D.rs_xtab_off + i] = (uint32_t)sx | ((uint32_t)sx1 << 16);
                tab[D.rs_xtab_off + nx + i] = (uint32_t)a0 | ((uint32_t)a1 << 16);
            }
            {   // pyr_resize_direct_kernel: the two source bytes of every column as a v_perm_b32 selector into the 8 bytes that
                // start at the leftmost source byte of the column's 4-column word, and that byte's offset per word
                D.rs_dtab_off = (int)tab.size();
                tab.resize(tab.size() + (size_t)nx + nx / 4, 0);
                bool direct = true;
                for (int wd = 0; wd < nx / 4; wd++) {
                    int lo = INT_MAX;
                    for (int j = 0; j < 4; j++) {
                        const uint32_t e = tab[D.rs_xtab_off + 4 * wd + j];
                        lo = std::min(lo, std::min((int)(e & 0xffffu), (int)(e >> 16)));
                    }
                    tab[D.rs_dtab_off + nx + wd] = (uint32_t)lo;
                    if (orbfe_resize_word_base_host(wd, D.w, D.rs_scale_x, S.w) != lo) direct = false; // the kernel computes this offset
                    for (int j = 0; j < 4; j++) {
                        const uint32_t e = tab[D.rs_xtab_off + 4 * wd + j];
                        const int o0 = (int)(e & 0xffffu) - lo, o1 = (int)(e >> 16) - lo;
                        if (o0 > 7 || o1 > 7) direct = false;
                        tab[D.rs_dtab_off + 4 * wd + j] = 0x0c000c00u | (uint32_t)(o0 & 7) | ((uint32_t)(o1 & 7) << 16);
                    }
                }
                const char *env = getenv("ORBFE_PYR_LDS");
                D.rs_direct = direct && ORBFE_PYR_RB > 0 && !(env && env[0] == '1');
                if (getenv("ORBFE_HOST_TRACE")) fprintf(stderr, "orbfe: level %d cv::resize: %s kernel\n", l, D.rs_direct ? "direct (aligned 96-bit row loads)" : "LDS-staged");
            }
            D.rs_ytab_off = (int)tab.size(); D.rs_ytab_n = ny;
            tab.resize(tab.size() + 2 * (size_t)ny, 0);
            for (int i = 0; i < ny; i++) {
                const int dy = reflect(i - 3, D.h);
                float fy = (float)(((double)dy + 0.5) * D.rs_scale_y - 0.5);
                int sy = (int)floorf(fy);
                fy -= (float)sy;
                const int b0 = (int)lrintf((1.f - fy) * 2048.f), b1 = (int)lrintf(fy * 2048.f);
                const int sy0 = sy < 0 ? 0 : (sy > S.h - 1 ? S.h - 1 : sy);
                const int sy1 = sy + 1 < 0 ? 0 : (sy + 1 > S.h - 1 ? S.h - 1 : sy + 1);
                tab[D.rs_ytab_off + i] = (uint32_t)sy0 | ((uint32_t)sy1 << 16);
                tab[D.rs_ytab_off + ny + i] = (uint32_t)b0 | ((uint32_t)b1 << 16);
            }
            for (int v = 0; v < 3; v++) { // source-row span of the worst block of 16 / 8 / 4 output rows
                const int rows = 16 >> v;
                int worst = 1;
                for (int y0 = 0; y0 < ny; y0 += rows) {
                    int lo = INT_MAX, hi = -1;
                    for (int i = y0; i < y0 + rows && i < ny; i++) {
                        const uint32_t e = tab[D.rs_ytab_off + i];
                        const int a = (int)(e & 0xffffu), b = (int)(e >> 16);
                        lo = std::min(lo, std::min(a, b)); hi = std::max(hi, std::max(a, b));
                    }
                    worst = std::max(worst, hi - lo + 1);
                }
                D.rs_src_rows[v] = worst;
            }
        }
        {   // per-block source-row range of the resize launches (pyr_resize_kernel stages it before it has read any row table)
            std::vector<uint32_t> blk;
            for (int l = 1; l < p.nlevels; l++) {
                LevelInfo &D = ctx->cfg.lv[l];
                const int src_words = (ctx->cfg.lv[l - 1].w + 3) / 4, ny = D.h + 6;
                const size_t rowp = ((size_t)src_words * 4 + 15) & ~(size_t)15; // staged row pitch (pyr_resize_kernel)
                D.rs_rw = D.rs_src_rows[0] * rowp <= 60 * 1024 ? 4 : (D.rs_src_rows[1] * rowp <= 60 * 1024 ? 2 : 1);
                D.rs_blk_off = (int)blk.size();
                const int rows = 4 * D.rs_rw;
                for (int y0 = 0; y0 < ny; y0 += rows) {
                    int lo = INT_MAX, hi = -1;
                    for (int i = y0; i < y0 + rows && i < ny; i++) {
                        const uint32_t e = tab[D.rs_ytab_off + i];
                        const int a = (int)(e & 0xffffu), b2 = (int)(e >> 16);
                        lo = std::min(lo, std::min(a, b2)); hi = std::max(hi, std::max(a, b2));
                    }
                    blk.push_back((uint32_t)lo | ((uint32_t)(hi - lo + 1) << 16));
                }
            }
            if (blk.empty()) blk.push_back(0);
            uint32_t *d_blk = nullptr;
            A(d_blk, blk.size());
            if (hipMemcpy(d_blk, blk.data(), blk.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
                orbfe_destroy(ctx);
                return fail(nullptr, ORBFE_ERR_HIP, "resize block table upload failed");
            }
            b.rs_blk = d_blk;
        }
        {   // plan of the fused pyramid tail (pyr_tail_kernel): a workgroup owns a strip of ORBFE_TAIL_COLS extended columns of
            // the LAST level over all its rows and computes, level by level in LDS, exactly the columns of the previous tail
            // levels that strip needs (plus the margin columns at the image's left / right, which no later level reads);
            // columns shared by neighbouring strips (one or two per level) are computed twice
            DeviceConfig &c2 = ctx->cfg;
            c2.tail_first = 0; c2.tail_n = 0; c2.tail_strips = 0; c2.tail_lds_bytes = 0;
            const int nst = p.nlevels >= 4 ? 3 : (p.nlevels == 3 ? 2 : 0);
            { const char *nf = getenv("ORBFE_NO_FUSE"); ctx->fuse_blur = !(nf && nf[0] == '1'); }
            { const char *bf = getenv("ORBFE_BLUR_RIDE_FROM"); if (bf && bf[0] >= '0' && bf[0] <= '9') { ctx->blur_ride_from = atoi(bf); ctx->blur_ride_min_images = 1; } } // given explicitly: for every batch size
            const char *env = getenv("ORBFE_NO_TAIL");
            c2.tail_max_images = env && env[0] == '0' ? INT_MAX : 63; // ORBFE_NO_TAIL=0: the tail at every batch size (A/B)
            if (nst >= 2 && !(env && env[0] == '1')) {
                const int F = p.nlevels - nst, Lz = p.nlevels - 1;
                const int strips = (c2.lv[Lz].rs_xtab_n + ORBFE_TAIL_COLS - 1) / ORBFE_TAIL_COLS;
                std::vector<int> plan((size_t)strips * ORBFE_TAIL_MAX * 4, 0);
                int max_wd[ORBFE_TAIL_MAX] = {0, 0, 0}, max_src = 0;
                bool ok = true;
                auto src_hull = [&](int l, int x0, int n, int &lo, int &hi) { // source columns of extended columns [x0, x0 + n) of level l
                    lo = INT_MAX; hi = -1;
                    for (int i = x0; i < x0 + n; i++) {
                        const uint32_t e = tab[c2.lv[l].rs_xtab_off + i];
                        const int a = (int)(e & 0xffffu), b2 = (int)(e >> 16);
                        lo = std::min(lo, std::min(a, b2)); hi = std::max(hi, std::max(a, b2));
                    }
                };
                for (int sj = 0; sj < strips; sj++) {
                    int x0 = sj * ORBFE_TAIL_COLS, n = std::min(ORBFE_TAIL_COLS, c2.lv[Lz].rs_xtab_n - x0);
                    for (int st = nst - 1; st >= 0; st--) {
                        const int l = F + st;
                        int *e = &plan[((size_t)sj * ORBFE_TAIL_MAX + st) * 4];
                        e[0] = x0; e[1] = n >> 2;
                        max_wd[st] = std::max(max_wd[st], n >> 2);
                        if (n >> 2 > 64) ok = false; // one lane per 4-pixel word of a strip row
                        int lo, hi;
                        src_hull(l, x0, n, lo, hi); // interior columns of level l - 1
                        if (st == 0) {
                            e[2] = lo & ~3; e[3] = (((hi | 3) + 1) - (lo & ~3)) >> 2; // staged words of level F - 1 (4-aligned interior start)
                            max_src = std::max(max_src, e[3]);
                            break;
                        }
                        // columns of level l - 1 this strip computes: the needed interior columns (+ PYR_MX as extended index)
                        // rounded out to words, and the margin columns for the first / last strip
                        x0 = sj == 0 ? 0 : ((lo + 4) & ~3);
                        const int x1 = sj == strips - 1 ? c2.lv[l - 1].rs_xtab_n : std::min(c2.lv[l - 1].rs_xtab_n, ((hi + 4) | 3) + 1);
                        n = x1 - x0;
                    }
                }
                for (int st = 0; st < nst && ok; st++) { // every extended column of every tail level is computed by some strip
                    int covered = 0;
                    for (int sj = 0; sj < strips; sj++) {
                        const int *e = &plan[((size_t)sj * ORBFE_TAIL_MAX + st) * 4];
                        if (e[0] > covered) ok = false;
                        covered = std::max(covered, e[0] + 4 * e[1]);
                    }
                    if (covered != c2.lv[F + st].rs_xtab_n) ok = false;
                }
                size_t off = 0;
                for (int st = 0; st < nst; st++) { c2.tail_lds_y[st] = (int)off; off += (size_t)2 * c2.lv[F + st].rs_ytab_n * 4; }
                off = (off + 15) & ~(size_t)15;
                c2.tail_lds_src = (int)off;
                off += (size_t)c2.lv[F - 1].h * max_src * 4;
                for (int st = 0; st + 1 < nst; st++) {
                    c2.tail_lds_buf[st] = (int)off;
                    off += (size_t)(c2.lv[F + st].h + 6) * max_wd[st] * 4;
                }
                if (ok && F >= 1 && off <= 64 * 1024) {
                    c2.tail_first = F; c2.tail_n = nst; c2.tail_strips = strips; c2.tail_src_words = max_src;
                    for (int st = 0; st < nst; st++) c2.tail_words[st] = max_wd[st];
                    c2.tail_lds_bytes = (int)off;
                    if (getenv("ORBFE_HOST_TRACE")) fprintf(stderr, "orbfe: pyramid tail: levels %d..%d, %d strips, %d bytes of LDS per workgroup\n", F, F + nst - 1, strips, (int)off);
                    int *d_plan = nullptr;
                    A(d_plan, plan.size());
                    if (hipMemcpy(d_plan, plan.data(), plan.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
                        orbfe_destroy(ctx);
                        return fail(nullptr, ORBFE_ERR_HIP, "pyramid tail plan upload failed");
                    }
                    b.tail_plan = d_plan;
                }
            }
        }
        {   // plan of pyr_pair_kernel (levels l and l + 1 in one launch): tiles of TW words x TR extended rows of level l + 1; per tile
            // column the words of level l (extended) its sources lie in and the words it stores, per tile row the same for rows.
            // The stored ranges partition level l's extended domain; a tile computes the hull of both.
            std::vector<int> plan;
            const char *np = getenv("ORBFE_NO_PAIR");
            const bool want = !(np && np[0] == '1');
            ctx->cfg.pp_max_images = np && np[0] == '0' ? INT_MAX : 63; // ORBFE_NO_PAIR=0: pairs at every batch size (A/B)
            for (int l = 1; l + 1 < p.nlevels; l++) {
                LevelInfo &D1 = ctx->cfg.lv[l];
                const LevelInfo &D2 = ctx->cfg.lv[l + 1];
                D1.pp_ok = 0;
                if (!want || !D1.rs_direct || !D2.rs_direct) continue;
                const int nw1 = D1.rs_xtab_n >> 2, ny1 = D1.rs_ytab_n, nw2 = D2.rs_xtab_n >> 2, ny2 = D2.rs_ytab_n;
                static const int k_tr[4] = {12, 11, 10, 8}, k_tw[6] = {48, 44, 40, 32, 24, 16};
                for (int ci = 0; ci < 24 && !D1.pp_ok; ci++) { // the largest tile whose level-l rectangle fits 64 words x 16 rows
                    const int TR = k_tr[ci / 6], TW = k_tw[ci % 6];
                    const int ntx = (nw2 + TW - 1) / TW, nty = (ny2 + TR - 1) / TR;
                    std::vector<int> px((size_t)ntx * 4), py((size_t)nty * 4);
                    bool ok = true;
                    int prev = 0;
                    for (int t = 0; t < ntx && ok; t++) { // columns: sources of extended columns [4 TW t, ...) of level l + 1
                        int lo = INT_MAX, hi = -1;
                        for (int i = 4 * TW * t; i < std::min(4 * TW * (t + 1), D2.rs_xtab_n); i++) {
                            const uint32_t e = tab[D2.rs_xtab_off + i];
                            lo = std::min(lo, (int)std::min(e & 0xffffu, e >> 16)); hi = std::max(hi, (int)std::max(e & 0xffffu, e >> 16));
                        }
                        // the kernel's windows start at the word's computed first source byte and read 12 bytes from the 4-byte boundary below
                        const int need0 = (lo + 4) >> 2, need1 = ((hi + 4) >> 2) + 1; // extended words [need0, need1) of level l (interior column c = extended byte c + 4)
                        const int s0 = t == 0 ? 0 : std::max(prev, std::min(need0, prev)); // stores continue where the previous tile's ended
                        const int s1 = t == ntx - 1 ? nw1 : need1;
                        const int c0 = std::min(need0, s0), c1 = std::max(need1, s1);
                        if (t > 0 && need0 > prev) ok = false; // a gap nobody would store
                        px[4 * t] = c0; px[4 * t + 1] = c1 - c0; px[4 * t + 2] = s0; px[4 * t + 3] = std::max(s1, s0);
                        if (c1 - c0 > 64 || c0 < 0 || c1 > nw1) ok = false;
                        prev = std::max(s1, s0);
                    }
                    if (prev != nw1) ok = false;
                    prev = 0;
                    for (int t = 0; t < nty && ok; t++) {
                        int lo = INT_MAX, hi = -1;
                        for (int i = TR * t; i < std::min(TR * (t + 1), ny2); i++) {
                            const uint32_t e = tab[D2.rs_ytab_off + i];
                            lo = std::min(lo, (int)std::min(e & 0xffffu, e >> 16)); hi = std::max(hi, (int)std::max(e & 0xffffu, e >> 16));
                        }
                        const int need0 = lo + 3, need1 = hi + 3 + 1; // extended rows of level l (interior row r = extended row r + 3)
                        const int s0 = t == 0 ? 0 : prev;
                        const int s1 = t == nty - 1 ? ny1 : need1;
                        if (t > 0 && need0 > prev) ok = false;
                        const int c0 = std::min(need0, s0), c1 = std::max(need1, s1);
                        py[4 * t] = c0; py[4 * t + 1] = c1 - c0; py[4 * t + 2] = s0; py[4 * t + 3] = std::max(s1, s0);
                        if (c1 - c0 > 16 || c0 < 0 || c1 > ny1) ok = false;
                        prev = std::max(s1, s0);
                    }
                    if (prev != ny1) ok = false;
                    if (!ok) continue;
                    D1.pp_ok = 1; D1.pp_ntx = ntx; D1.pp_nty = nty; D1.pp_tw = TW; D1.pp_tr = TR;
                    D1.pp_xoff = (int)plan.size() / 4; plan.insert(plan.end(), px.begin(), px.end());
                    D1.pp_yoff = (int)plan.size() / 4; plan.insert(plan.end(), py.begin(), py.end());
                }
                if (getenv("ORBFE_HOST_TRACE")) fprintf(stderr, "orbfe: levels %d + %d in one launch: %s (%d x %d tiles of %d words x %d rows)\n", l, l + 1, D1.pp_ok ? "yes" : "no", D1.pp_ntx, D1.pp_nty, D1.pp_tw, D1.pp_tr);
            }
            if (plan.empty()) plan.resize(4, 0);
            int *d_pp = nullptr;
            A(d_pp, plan.size());
            if (hipMemcpy(d_pp, plan.data(), plan.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
                orbfe_destroy(ctx);
                return fail(nullptr, ORBFE_ERR_HIP, "pyramid pair plan upload failed");
            }
            b.pair_plan = d_pp;
        }
        while (tab.size() % 4) tab.push_back(0);
        if (tab.empty()) tab.resize(4, 0);
        uint32_t *d_tab = nullptr;
        A(d_tab, tab.size());
        if (hipMemcpy(d_tab, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
            orbfe_destroy(ctx);
            return fail(nullptr, ORBFE_ERR_HIP, "resize table upload failed");
        }
        b.rs_tab = d_tab;
    }
    std::vector<uint32_t> bk_tab_host;
    {   // FAST cell table: what fast_cell_kernel's prologue would otherwise derive per wave from a chain of dependent scalar loads
        // (level search over lv[].cell_off, a division by n_cols, the clipping of src/ORBextractor.cc:783-800)
        std::vector<uint32_t> ci((size_t)c.cells_total * 4, 0u), ca((size_t)c.cells_total * 2, 0u);
        std::vector<uint32_t> shapes, lane_tab; // tile w | h << 8 of every distinct cell shape; [shape][64][8]
        const int pw = orbfe_fast_tile_pitch(c) >> 2;
        for (int l = 0; l < p.nlevels; l++) {
            const LevelInfo &L = c.lv[l];
            const int max_bx = L.w - c.edge_threshold + 3, max_by = L.h - c.edge_threshold + 3;
            for (int k = 0; k < L.n_cells; k++) {
                const int i = k / L.n_cols, j = k - i * L.n_cols;
                const int ini_y = c.min_border + i * L.h_cell, ini_x = c.min_border + j * L.w_cell;
                int max_y = ini_y + L.h_cell + 6, max_x = ini_x + L.w_cell + 6;
                uint32_t *e = &ci[(size_t)(L.cell_off + k) * 4];
                e[0] = (uint32_t)l; e[3] = (uint32_t)k;
                if (ini_y >= max_by - 3 || ini_x >= max_bx - 6) continue; // src/ORBextractor.cc:788-798: no FAST call
                if (max_y > max_by) max_y = max_by;
                if (max_x > max_bx) max_x = max_bx;
                const int tw = max_x - ini_x, th = max_y - ini_y;
                if (tw - 6 <= 0 || th - 6 <= 0) continue;
                e[0] |= 1u << 8;
                e[1] = (uint32_t)ini_x | ((uint32_t)ini_y << 16);
                e[2] = (uint32_t)tw | ((uint32_t)th << 8);
                // the divisions of fast_cell_kernel's lane maps (its prologue: tile staging by 16-byte chunks, phase A by 4-pixel groups)
                const int iw = tw - 6, ih = th - 6;
                const int ng = (iw + 3) >> 2, dr = 64 / ng, r0_last = ((ih - 1) / dr) * dr;
                const int cpr = ((((tw + 3) >> 2)) + 3) >> 2, rpi = 64 / cpr;
                uint32_t *a = &ca[(size_t)(L.cell_off + k) * 2];
                size_t shape = 0;
                while (shape < shapes.size() && shapes[shape] != e[2]) shape++;
                if (shape == shapes.size()) { // phase A's per-lane constants for this tile size (fast_cell_kernel documents them)
                    shapes.push_back(e[2]);
                    for (int lane = 0; lane < 64; lane++) {
                        const int rl = lane / ng, jg = lane - rl * ng, c0 = 4 * jg;
                        const bool act = rl < dr, in_last = r0_last + rl < ih;
                        const uint32_t vm01 = (act ? 0x8000u : 0u) | (act && c0 + 1 < iw ? 0x80000000u : 0u);
                        const uint32_t vm23 = (act && c0 + 2 < iw ? 0x8000u : 0u) | (act && c0 + 3 < iw ? 0x80000000u : 0u);
                        const uint32_t row[8] = {vm01, vm23, in_last ? vm01 : 0u, in_last ? vm23 : 0u,
                                                 (uint32_t)(((rl + 1) << 8) + c0) * 0x10001u + 0x10000u, (uint32_t)(rl * pw + jg), 0u, 0u};
                        lane_tab.insert(lane_tab.end(), row, row + 8);
                    }
                }
                a[0] = (uint32_t)shape | ((uint32_t)dr << 17) | ((uint32_t)r0_last << 24);
                a[1] = (uint32_t)((65536 + cpr - 1) / cpr) | ((uint32_t)rpi << 17);
            }
        }
        uint32_t *d_ca = nullptr;
        A(d_ca, ca.size());
        if (hipMemcpy(d_ca, ca.data(), ca.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
            orbfe_destroy(ctx);
            return fail(nullptr, ORBFE_ERR_HIP, "cell table upload failed");
        }
        b.cell_aux = (const uint2 *)d_ca;
        if (lane_tab.empty()) lane_tab.resize(512, 0u);
        uint32_t *d_lt = nullptr;
        A(d_lt, lane_tab.size());
        if (hipMemcpy(d_lt, lane_tab.data(), lane_tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
            orbfe_destroy(ctx);
            return fail(nullptr, ORBFE_ERR_HIP, "cell table upload failed");
        }
        b.fast_lane_tab = d_lt;
        uint32_t *d_ci = nullptr;
        A(d_ci, ci.size());
        if (hipMemcpy(d_ci, ci.data(), ci.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
            orbfe_destroy(ctx);
            return fail(nullptr, ORBFE_ERR_HIP, "cell table upload failed");
        }
        b.cell_info = (const uint4 *)d_ci;
        // bucket partials (fast_cell_kernel phase E -> octree3_kernel): the survivors of a cell fall into the bucket columns
        // X[first col] >> 16 .. X[last col] >> 16 and the rows Y[..] >> 16 likewise (the tables are monotone).  A cell whose
        // rectangle has at most 64 buckets accumulates them in LDS and stores the count and best key of each with plain
        // stores into its own entries of bk_part (one word each); bk_emap names the bucket and the cell of every entry.  Cells with larger rectangles
        // (the small levels, where a bucket is 3 px wide) have no entries: the quadtree kernel buckets their candidates.
        std::vector<uint32_t> off((size_t)c.cells_total, 0u);
        std::vector<uint32_t> emap;
        auto spread5 = [](unsigned v) { unsigned r = 0; for (int i = 0; i < 5; i++) r |= ((v >> i) & 1u) << (2 * i); return r; };
        // Quadtree bucket tables (orbfe_octree3.hip): root and path bits down to the level's bucket depth of every x / y of the
        // level's candidate region, with the reference's arithmetic (src/ORBextractor.cc:537-564 roots, :145-209 splits):
        //   X[x] = root << 2 depth | x path bits spread to the even positions | (root * 2^depth + column) << 16
        //   Y[y] = y path bits spread to the odd positions | row << 16
        // Bucket depth per level (round 4): 5, or 4 where some FAST cell of the level spans more than 64 depth-5 buckets (the small
        // levels, whose depth-5 buckets are ~3 px: their candidates were bucketed one by one inside the quadtree kernel, 14-18 us of
        // those workgroups, which ended the launch).  A quota of ~200 nodes over 4 roots splits down to depth 3-4; nodes deeper than
        // the level's bucket depth take the kernel's slow path as before.
        auto build_tabs = [&](const LevelInfo &L, int depth, std::vector<uint32_t> &X, std::vector<uint32_t> &Y) {
            const int region_w = (L.w - p.edge_threshold + 3) - c.min_border, region_h = (L.h - p.edge_threshold + 3) - c.min_border;
            X.clear(); Y.clear();
            for (int x = 0; x < region_w; x++) {
                int b = (int)((float)x / L.hx);
                b = b < 0 ? 0 : (b >= L.n_ini ? L.n_ini - 1 : b);
                int x0 = (int)(L.hx * (float)b), x1 = (int)(L.hx * (float)(b + 1));
                unsigned col = 0;
                for (int d = 0; d < depth; d++) {
                    const int mx = x0 + ((x1 - x0 + 1) >> 1);
                    const int cx = x < mx ? 0 : 1;
                    col = (col << 1) | (unsigned)cx;
                    if (cx) x0 = mx; else x1 = mx;
                }
                X.push_back(((unsigned)b << (2 * depth)) | spread5(col) | ((((unsigned)b << depth) + col) << 16));
            }
            for (int y = 0; y < region_h; y++) {
                int y0 = 0, y1 = region_h;
                unsigned row = 0;
                for (int d = 0; d < depth; d++) {
                    const int my = y0 + ((y1 - y0 + 1) >> 1);
                    const int cy = y < my ? 0 : 1;
                    row = (row << 1) | (unsigned)cy;
                    if (cy) y0 = my; else y1 = my;
                }
                Y.push_back((spread5(row) << 1) | (row << 16));
            }
        };
        auto max_cell_buckets = [&](const LevelInfo &L, const std::vector<uint32_t> &X, const std::vector<uint32_t> &Y) {
            int worst = 0;
            for (int k = 0; k < L.n_cells; k++) {
                const uint32_t *e = &ci[(size_t)(L.cell_off + k) * 4];
                if (!(e[0] & 0x100u)) continue;
                const int cx0 = (int)(e[1] & 0xffffu) - c.min_border, cy0 = (int)(e[1] >> 16) - c.min_border;
                const int iw = (int)(e[2] & 0xffu) - 6, ih = (int)((e[2] >> 8) & 0xffu) - 6;
                const int nb = ((int)(X[3 + cx0 + iw - 1] >> 16) - (int)(X[3 + cx0] >> 16) + 1) * ((int)(Y[3 + cy0 + ih - 1] >> 16) - (int)(Y[3 + cy0] >> 16) + 1);
                worst = std::max(worst, nb);
            }
            return worst;
        };
        {
            std::vector<uint32_t> X, Y;
            for (int l = 0; l < p.nlevels; l++) {
                LevelInfo &L = ctx->cfg.lv[l];
                L.bk_depth = ORBFE_BK_DEPTH;
                build_tabs(L, ORBFE_BK_DEPTH, X, Y);
                if (max_cell_buckets(L, X, Y) > 64) { // depth 4 where a FAST cell would span more than 64 depth-5 buckets (else the quadtree kernel buckets that level's candidates one by one)
                    std::vector<uint32_t> X4, Y4;
                    build_tabs(L, ORBFE_BK_DEPTH - 1, X4, Y4);
                    if (max_cell_buckets(L, X4, Y4) <= 64) { L.bk_depth = ORBFE_BK_DEPTH - 1; X.swap(X4); Y.swap(Y4); }
                }
                L.bk_xoff = (int)bk_tab_host.size(); bk_tab_host.insert(bk_tab_host.end(), X.begin(), X.end());
                L.bk_yoff = (int)bk_tab_host.size(); bk_tab_host.insert(bk_tab_host.end(), Y.begin(), Y.end());
                if (getenv("ORBFE_HOST_TRACE")) fprintf(stderr, "orbfe: level %d quadtree bucket depth %d\n", l, L.bk_depth);
            }
            if (bk_tab_host.empty()) bk_tab_host.resize(4, 0);
            uint32_t *d_tab = nullptr;
            A(d_tab, bk_tab_host.size());
            if (hipMemcpy(d_tab, bk_tab_host.data(), bk_tab_host.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
                orbfe_destroy(ctx);
                return fail(nullptr, ORBFE_ERR_HIP, "bucket table upload failed");
            }
            b.bk_tab = d_tab;
        }
        for (int l = 0; l < p.nlevels; l++) {
            LevelInfo &L = ctx->cfg.lv[l];
            while (emap.size() % 4) emap.push_back(0); // octree3_kernel reads a level's entries as 128-bit quads: 16-byte aligned start, padded end (the padding's bk_part words stay zero: no cell owns them)
            L.bk_part_off = (int)emap.size();
            L.bk_points = 0;
            for (int k = 0; k < L.n_cells; k++) {
                const uint32_t *e = &ci[(size_t)(L.cell_off + k) * 4];
                off[L.cell_off + k] = (uint32_t)emap.size();
                if (!(e[0] & 0x100u)) continue; // no FAST call: no entries
                const int cx0 = (int)(e[1] & 0xffffu) - c.min_border, cy0 = (int)(e[1] >> 16) - c.min_border;
                const int iw = (int)(e[2] & 0xffu) - 6, ih = (int)((e[2] >> 8) & 0xffu) - 6;
                const uint32_t *tx = &bk_tab_host[L.bk_xoff + 3 + cx0], *ty = &bk_tab_host[L.bk_yoff + 3 + cy0];
                const int gx0 = (int)(tx[0] >> 16), gx1 = (int)(tx[iw - 1] >> 16), by0 = (int)(ty[0] >> 16), by1 = (int)(ty[ih - 1] >> 16);
                const int ncols = gx1 - gx0 + 1, nb = ncols * (by1 - by0 + 1);
                if (nb > 64) { off[L.cell_off + k] = ~0u; L.bk_points = 1; continue; }
                for (int j = 0; j < nb; j++) {
                    const unsigned gx = (unsigned)(gx0 + j % ncols), by = (unsigned)(by0 + j / ncols);
                    const int dp = L.bk_depth;
                    emap.push_back(((gx >> dp) << (2 * dp)) | spread5(gx & ((1u << dp) - 1u)) | (spread5(by) << 1) | ((uint32_t)k << 16));
                }
            }
            while (emap.size() % 4) emap.push_back(0);
            L.bk_part_n = (int)emap.size() - L.bk_part_off;
        }
        ctx->cfg.bk_part_total = (int)emap.size();
        while (emap.size() % 8 || emap.empty()) emap.push_back(0);
        uint32_t *d_off = nullptr;
        uint32_t *d_emap = nullptr;
        A(d_off, off.size());
        A(d_emap, emap.size());
        A(b.bk_part, B * emap.size());
        Z(b.bk_part, B * emap.size() * sizeof(uint32_t));
        if (hipMemcpy(d_off, off.data(), off.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_emap, emap.data(), emap.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
            orbfe_destroy(ctx);
            return fail(nullptr, ORBFE_ERR_HIP, "bucket partial table upload failed");
        }
        b.bk_off = d_off;
        b.bk_emap = d_emap;
    }
    {   // blur tile table (blur_kernel): level, 256-column strip and first row of every wave's tile
        std::vector<uint32_t> ti((size_t)(c.blur_tiles_total > 0 ? c.blur_tiles_total : 1), 0u);
        for (int l = 0; l < p.nlevels; l++) {
            const LevelInfo &L = c.lv[l];
            for (int t = 0; t < L.blur_tiles_x * L.blur_tiles_y; t++)
                ti[L.blur_tile_off + t] = (uint32_t)l | ((uint32_t)(t % L.blur_tiles_x) << 8) | ((uint32_t)((t / L.blur_tiles_x) * ORBFE_BLUR_ROWS) << 16);
        }
        uint32_t *d_ti = nullptr;
        A(d_ti, ti.size());
        if (hipMemcpy(d_ti, ti.data(), ti.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
            orbfe_destroy(ctx);
            return fail(nullptr, ORBFE_ERR_HIP, "blur tile table upload failed");
        }
        b.blur_tile_info = d_ti;
    }
    {   // keypoint slot -> level
        std::vector<uint8_t> sl((size_t)c.sel_total + 4, 0); // + 4: read as whole 32-bit words of four slots (stereo_rowlist_kernel)
        for (int l = 0; l < p.nlevels; l++)
            for (int k = 0; k < c.lv[l].sel_cap; k++) sl[c.lv[l].sel_off + k] = (uint8_t)l;
        uint8_t *d_sl = nullptr;
        A(d_sl, sl.size());
        if (hipMemcpy(d_sl, sl.data(), sl.size(), hipMemcpyHostToDevice) != hipSuccess) {
            orbfe_destroy(ctx);
            return fail(nullptr, ORBFE_ERR_HIP, "slot table upload failed");
        }
        b.slot_level = d_sl;
    }
    {   // describe_kernel's processing order (orbfe_octree3.hip step 4a): a node's bin = row bits | root | column bits of its quadrant path,
        // rows ~40 px tall (a patch is 31 - 40 rows), the remaining bits of the 256 bins for columns
        const char *np = getenv("ORBFE_NO_PROC_ORDER");
        ctx->cfg.proc_order = !(np && np[0] == '1') && ctx->use_octree3 && !ctx->ot3_nodes_in_hbm;
        for (int l = 0; l < p.nlevels; l++) {
            LevelInfo &L = ctx->cfg.lv[l];
            const int region_h = (L.h - p.edge_threshold + 3) - c.min_border;
            int rb = 0;
            while (rb < 3 && (region_h >> (rb + 1)) >= 28) rb++; // rows of 28 - 55 px (measured: 20 / 40-px minima cost describe_kernel 1 us, 14 / 112 px 3 - 5 us)
            L.po_rb = rb; L.po_cb = 6 - rb;
        }
    }
    {   // circular patch of IC_Angle (src/ORBextractor.cc:79-96): |v| <= hp, |u| <= umax[|v|]
        std::vector<int16_t> uv;
        const int hp = p.half_patch_size;
        for (int v = -hp; v <= hp; v++) {
            const int d = c.umax[v < 0 ? -v : v];
            for (int u = -d; u <= d; u++) uv.push_back((int16_t)((u & 0xff) | ((v & 0xff) << 8)));
        }
        while (uv.size() % 64) uv.push_back(0);
        int16_t *d_uv = nullptr;
        A(d_uv, uv.size());
        if (hipMemcpy(d_uv, uv.data(), uv.size() * sizeof(int16_t), hipMemcpyHostToDevice) != hipSuccess) {
            orbfe_destroy(ctx);
            return fail(nullptr, ORBFE_ERR_HIP, "patch table upload failed");
        }
        b.patch_uv = d_uv;
        ctx->cfg.patch_n = (int)uv.size();
    }
    {   // the same patch as byte-dot-product weights for describe_kernel (hp == 15): lane = 2 * row + half holds the 16 pixels
        // u = -15 + 16 * half .. of row v = row - 15 as four words (its one 128-bit load of the raw patch); per lane 12 words:
        // [0..3] weights (u + 16) inside the circle else 0, [4..7] weights 1 / 0, [8] v (three 128-bit loads per lane).
        // Lanes 62 / 63 (no row 31) and u = 16 get zero weights.
        std::vector<uint32_t> mt(12 * 64, 0u);
        if (p.half_patch_size == 15)
            for (int lane = 0; lane < 62; lane++) {
                const int r = lane >> 1, half = lane & 1, v = r - 15, um = c.umax[v < 0 ? -v : v];
                for (int k = 0; k < 4; k++) {
                    uint32_t wu = 0, w1 = 0;
                    for (int j = 0; j < 4; j++) {
                        const int u = 16 * half + 4 * k + j - 15;
                        if (u >= -um && u <= um) { wu |= (uint32_t)(u + 16) << (8 * j); w1 |= 1u << (8 * j); }
                    }
                    mt[(size_t)lane * 12 + k] = wu;
                    mt[(size_t)lane * 12 + 4 + k] = w1;
                }
                mt[(size_t)lane * 12 + 8] = (uint32_t)v;
            }
        uint32_t *d_mt = nullptr;
        A(d_mt, mt.size());
        if (hipMemcpy(d_mt, mt.data(), mt.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
            orbfe_destroy(ctx);
            return fail(nullptr, ORBFE_ERR_HIP, "moment table upload failed");
        }
        b.mom_tab = d_mt;
    }
    {   // the context's copy of the 256 rBRIEF tests (ORBextractor's member `pattern`, src/ORBextractor.cc:442-444)
        uint32_t *d_pat = nullptr;
        A(d_pat, 256);
        if (hipMemcpy(d_pat, k_bit_pattern_31, 1024, hipMemcpyHostToDevice) != hipSuccess) {
            orbfe_destroy(ctx);
            return fail(nullptr, ORBFE_ERR_HIP, "pattern upload failed");
        }
        b.pattern = d_pat;
        memcpy(ctx->pattern, k_bit_pattern_31, 1024);
    }
    {
        const char *ni = getenv("ORBFE_NO_INPLACE");
        const DeviceConfig &cc = ctx->cfg;
        ctx->inplace_ok = !(ni && ni[0] == '1') && cc.nlevels >= 2 && cc.lv[1].rs_direct && cc.tail_first != 1 && p.width >= 16 && p.height >= 8;
        if (getenv("ORBFE_HOST_TRACE")) fprintf(stderr, "orbfe: level 0 of packed grey input: %s\n", ctx->inplace_ok ? "read in place (no ingest launch)" : "copied by ingest16_kernel");
        b.lv0 = b.pyr + cc.lv[0].pyr_off; b.lv0_stride = cc.pyr_bytes; b.lv0_pitch = cc.lv[0].pitch; b.lv0_packed = 0;
    }
    Z(b.kp_cnt, sizeof(int) * B);
    Z(b.sel_cnt, sizeof(int) * B * c.nlevels);
    Z(b.status, sizeof(int) * B);
#undef A
#undef Z
    *out = ctx;
    return ORBFE_OK;
} ORBFE_CATCH(nullptr)

extern "C" void orbfe_destroy(orbfe_context *ctx)
{
    if (!ctx) return;
    if (ctx->stream) { hipStreamSynchronize(ctx->stream); }
    for (void *q : ctx->allocs) hipFree(q);
    for (hipEvent_t e : ctx->events) hipEventDestroy(e);
    for (int g = 0; g < ORBFE_MAX_GROUPS; g++) {
        if (ctx->gstreams[g]) { hipStreamSynchronize(ctx->gstreams[g]); hipStreamDestroy(ctx->gstreams[g]); }
        if (ctx->ev_join[g]) hipEventDestroy(ctx->ev_join[g]);
    }
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_latest) hipEventDestroy(ctx->ev_latest);
    if (ctx->ev_pack) { if (ctx->ev_pack_set) hipEventSynchronize(ctx->ev_pack); hipEventDestroy(ctx->ev_pack); }
    if (ctx->match) orbfe_match_state_destroy(ctx->match);
    if (ctx->bow) orbfe_bow_state_destroy(ctx->bow);
    if (ctx->pose) orbfe_pose_state_destroy(ctx->pose);
    if (ctx->d_depth_in) hipFree(ctx->d_depth_in);
    if (ctx->h_in) hipHostFree(ctx->h_in);
    if (ctx->h_depth_in) hipHostFree(ctx->h_depth_in);
    if (ctx->h_out) hipHostFree(ctx->h_out);
    if (ctx->d_ham) hipFree(ctx->d_ham);
    if (ctx->d_pack) hipFree(ctx->d_pack);
    for (int sd = 0; sd < 2; sd++) { if (ctx->cfg.rm_xy[sd]) hipFree((void *)ctx->cfg.rm_xy[sd]); if (ctx->cfg.rm_a[sd]) hipFree((void *)ctx->cfg.rm_a[sd]); }
    if (ctx->d_und) hipFree(ctx->d_und);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int orbfe_get_camera(const orbfe_context *ctx, float *cam)
try {
    if (!ctx || !cam) return ORBFE_ERR_INVALID;
    cam[0] = ctx->params.fx; cam[1] = ctx->params.fy; cam[2] = ctx->params.cx; cam[3] = ctx->params.cy; cam[4] = ctx->params.bf;
    return ORBFE_OK;
} ORBFE_CATCH(nullptr)

static int wait_latest(orbfe_context *ctx);
// ORBextractor copies bit_pattern_31_ into its member `pattern` (src/ORBextractor.cc:442-444); a deployment that distributes the
// table (one broadcast from rank 0: orbslam2_amd/dist.py) hands it to every context here.  Takes effect for calls enqueued afterwards.
extern "C" int orbfe_set_pattern(orbfe_context *ctx, const int32_t *pattern)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !pattern) return fail(ctx, ORBFE_ERR_INVALID, "null argument");
    int8_t pk[1024];
    for (int i = 0; i < 512; i++) {
        const int x = pattern[2 * i], y = pattern[2 * i + 1];
        // describe_kernel stages +-18 px around a keypoint (what edge_threshold >= 19 guarantees inside the level): a rotated
        // test point lands cvRound(r * cos / sin) <= cvRound(r) px away, r^2 = x^2 + y^2; r^2 <= 342 <=> r <= 18.493 rounds to 18
        // (bit_pattern_31_ itself reaches r^2 = 338: the point (13, 13))
        if (x < -18 || x > 18 || y < -18 || y > 18 || x * x + y * y > 342)
            return fail(ctx, ORBFE_ERR_UNSUPPORTED, "pattern point %d = (%d, %d) can rotate to more than 18 px from the keypoint", i, x, y);
        pk[2 * i] = (int8_t)x; pk[2 * i + 1] = (int8_t)y;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->params.device));
    { const int rcw = wait_latest(ctx); if (rcw != ORBFE_OK) return rcw; } // calls already enqueued keep the table they were enqueued with
    HIP_TRY(ctx, hipMemcpy((void *)ctx->buf.pattern, pk, 1024, hipMemcpyHostToDevice));
    memcpy(ctx->pattern, pk, 1024);
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

extern "C" int orbfe_get_pattern(const orbfe_context *ctx, int32_t *pattern)
try {
    if (!ctx || !pattern) return ORBFE_ERR_INVALID;
    for (int i = 0; i < 1024; i++) pattern[i] = ctx->pattern[i];
    return ORBFE_OK;
} ORBFE_CATCH(nullptr)

extern "C" int orbfe_levels(const orbfe_context *ctx) { return ctx ? ctx->cfg.nlevels : ORBFE_ERR_INVALID; }
extern "C" int orbfe_keypoint_capacity(const orbfe_context *ctx) { return ctx ? ctx->cfg.sel_total : ORBFE_ERR_INVALID; }

extern "C" int orbfe_get_tables(const orbfe_context *ctx, float *scale, float *inv_scale, float *sigma2,
                                float *inv_sigma2, int32_t *features_per_level, int32_t *umax)
try {
    if (!ctx) return ORBFE_ERR_INVALID;
    const int n = ctx->cfg.nlevels;
    if (scale) memcpy(scale, ctx->scale, sizeof(float) * n);
    if (inv_scale) memcpy(inv_scale, ctx->inv_scale, sizeof(float) * n);
    if (sigma2) memcpy(sigma2, ctx->sigma2, sizeof(float) * n);
    if (inv_sigma2) memcpy(inv_sigma2, ctx->inv_sigma2, sizeof(float) * n);
    if (features_per_level) memcpy(features_per_level, ctx->feats, sizeof(int32_t) * n);
    if (umax) memcpy(umax, ctx->cfg.umax, sizeof(int32_t) * (ctx->cfg.half_patch + 1));
    return ORBFE_OK;
} ORBFE_CATCH(nullptr)

extern "C" int orbfe_level_size(const orbfe_context *ctx, int level, int *w, int *h)
try {
    if (!ctx || level < 0 || level >= ctx->cfg.nlevels) return ORBFE_ERR_INVALID;
    if (w) *w = ctx->cfg.lv[level].w;
    if (h) *h = ctx->cfg.lv[level].h;
    return ORBFE_OK;
} ORBFE_CATCH(nullptr)

static hipStream_t pick_stream(orbfe_context *ctx, void *stream) { return stream ? (hipStream_t)stream : ctx->stream; }

// The blocking fetch entry points copy on the null stream, which does not wait for the non-blocking streams the
// enqueue calls run on: wait for the stream of the latest enqueue (and the context's own) first.
static int wait_latest(orbfe_context *ctx)
{
    if (ctx->latest_foreign) HIP_TRY(ctx, hipEventSynchronize(ctx->ev_latest));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ORBFE_OK;
}

#define PROF_RING 64
static const char *k_stage_names[ORBFE_NUM_STAGES] = {"ingest", "pyramid", "blur", "fast", "octree", "describe",
                                                      "stereo_match", "stereo_median"};
extern "C" const char *orbfe_stage_name(int stage) { return stage >= 0 && stage < ORBFE_NUM_STAGES ? k_stage_names[stage] : ""; }

extern "C" int orbfe_set_profiling(orbfe_context *ctx, int enabled)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx) return ORBFE_ERR_INVALID;
    if (enabled && ctx->events.empty()) {
        ctx->events.resize((size_t)PROF_RING * ORBFE_MAX_GROUPS * (ORBFE_NUM_STAGES + 1));
        for (auto &e : ctx->events) HIP_TRY(ctx, hipEventCreate(&e));
    }
    if (enabled >= 2 + ORBFE_NUM_STAGES || enabled < 0) return fail(ctx, ORBFE_ERR_INVALID, "profiling mode must be 0, 1 or 2 + stage");
    ctx->profiling = enabled != 0;
    ctx->prof_seq = 0;
    ctx->prof_only = enabled >= 2 ? enabled - 2 : -1;
    ctx->prof_calls = 0;
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

extern "C" int orbfe_set_profiling_interval(orbfe_context *ctx, int every)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || every < 1) return fail(ctx, ORBFE_ERR_INVALID, "interval must be >= 1");
    ctx->prof_every = every;
    ctx->prof_seq = 0;
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// record event #idx of the current call for stream group `group` (idx 0 = before the first stage)
static inline void prof_mark(orbfe_context *ctx, int group, int idx, hipStream_t s)
{
    if (!ctx->prof_now) return;
    if (ctx->prof_only >= 0 && idx != ctx->prof_only && idx != ctx->prof_only + 1) {
        ctx->prof_stages[ctx->prof_calls % PROF_RING] = idx;
        return;
    }
    const int slot = ctx->prof_calls % PROF_RING;
    hipEventRecord(ctx->events[((size_t)slot * ORBFE_MAX_GROUPS + group) * (ORBFE_NUM_STAGES + 1) + idx], s);
    ctx->prof_stages[slot] = idx;
}

// Per-stage elapsed ms summed over the recorded calls AND over the stream groups of each call
// (with G groups a stage runs as G launches per call, or 7*G for the pyramid).
extern "C" int orbfe_stage_times(orbfe_context *ctx, float *ms, int *calls, int reset)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !ms) return ORBFE_ERR_INVALID;
    for (int i = 0; i < ORBFE_NUM_STAGES; i++) ms[i] = 0.f;
    int n = ctx->prof_calls < PROF_RING ? ctx->prof_calls : PROF_RING;
    if (calls) *calls = n;
    if (n > 0) {
        { const int rcw = wait_latest(ctx); if (rcw != ORBFE_OK) return rcw; }
        for (int c = 0; c < n; c++)
            for (int g = 0; g < ctx->prof_groups; g++) {
                const hipEvent_t *ev = &ctx->events[((size_t)c * ORBFE_MAX_GROUPS + g) * (ORBFE_NUM_STAGES + 1)];
                for (int st = 0; st < ctx->prof_stages[c]; st++) {
                    if (ctx->prof_only >= 0 && st != ctx->prof_only) continue;
                    float t = 0.f;
                    HIP_TRY(ctx, hipEventElapsedTime(&t, ev[st], ev[st + 1]));
                    ms[st] += t;
                }
            }
    }
    if (reset) { ctx->prof_calls = 0; ctx->prof_seq = 0; }
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

static inline int ot_sort_cap_of(const DeviceConfig &c) { int p = 1; while (p < c.max_nodes) p <<= 1; return p; }

// View of the per-image buffers starting at image img0 (kernels index images from 0).
static DeviceBuffers shift_buffers(const DeviceBuffers &b, const DeviceConfig &c, int img0)
{
    DeviceBuffers o = b;
    const size_t i = (size_t)img0;
    o.pyr += i * c.pyr_bytes; o.blur += i * c.blur_bytes;
    o.lv0 = o.pyr + c.lv[0].pyr_off; o.lv0_stride = c.pyr_bytes; o.lv0_pitch = c.lv[0].pitch; o.lv0_packed = 0;
    o.cell_cnt += i * c.cells_total; o.cell_base += i * c.cells_total;
    o.cell_xy += i * c.cells_total * c.cell_cap; o.cell_sc += i * c.cells_total * c.cell_cap;
    o.cand_xy += i * c.cand_total; o.cand_sc += i * c.cand_total;
    o.idx0 += i * c.cand_total; o.idx1 += i * c.cand_total; o.ot_xy2 += i * c.cand_total;
    o.lvl_ncand += i * c.nlevels; o.sel_cnt += i * c.nlevels;
    o.bk_part += i * c.bk_part_total; o.bk_end += i * c.nlevels * 4097;
    if (o.ot3_scratch) o.ot3_scratch += i * c.nlevels * orbfe_octree3_node_bytes(c.max_nodes, ot_sort_cap_of(c));
    if (o.bk_best) o.bk_best += i * c.nlevels * ORBFE_BK_PYR;
    o.sel_xy += i * c.sel_total; o.sel_sc += i * c.sel_total; o.proc_xy += i * c.sel_total; o.proc_meta += i * c.sel_total;
    o.kps = (KeyPointPOD *)o.kps + i * c.sel_total; o.desc += i * c.sel_total * 32;
    o.kp_cnt += i; o.status += i;
    o.u_right += i * c.sel_total; o.depth += i * c.sel_total; o.sad += i * c.sel_total;
    o.row_cnt += (i / 2) * (size_t)c.height; o.row_ent += (i / 2) * (size_t)c.height * c.row_cap;
    return o;
}

static int blur_ride_from_of(const orbfe_context *ctx, int n_images)
{
    if (!ctx->fuse_blur) return ctx->cfg.nlevels;
    return n_images >= ctx->blur_ride_min_images ? (ctx->blur_ride_from < ctx->cfg.nlevels ? ctx->blur_ride_from : ctx->cfg.nlevels) : ORBFE_MAX_LEVELS;
}

// First pyramid level whose Gaussian blur rides in FAST's launch for a batch of n_images images (nlevels: none rides; the value is
// an upper bound for batches below the threshold, whose pyramid launches blur as many levels as they reach): what bench.py
// attributes to the dominant kernel's launch.
extern "C" int orbfe_blur_ride_from(const orbfe_context *ctx, int n_images)
try {
    if (!ctx) return ORBFE_ERR_INVALID;
    const int r = blur_ride_from_of(ctx, n_images);
    return r < ctx->cfg.nlevels ? r : (ctx->fuse_blur && n_images < ctx->blur_ride_min_images ? ctx->cfg.nlevels - 1 : ctx->cfg.nlevels);
} ORBFE_CATCH(nullptr)

// One chain of stages over images [img0, img0 + n_images) on stream s; stereo stages if n_pairs > 0.
static void run_chain(orbfe_context *ctx, const uint8_t *d_images, int img0, int n_images, int n_pairs, hipStream_t s, int group)
{
    const DeviceConfig &cfg = ctx->cfg;
    DeviceBuffers buf = shift_buffers(ctx->buf, cfg, img0);
    const uint8_t *src = d_images + (size_t)img0 * cfg.in_image_bytes;
    prof_mark(ctx, group, 0, s);
    if (ctx->inplace_ok && cfg.in_cn == 1 && !cfg.rm_on) { // level 0 = the caller's images (the first pyramid launch clears the status words)
        buf.lv0 = src; buf.lv0_stride = cfg.in_image_bytes; buf.lv0_pitch = cfg.width; buf.lv0_packed = 1;
    } else {
        orbfe_launch_ingest(cfg, buf, src, n_images, s);
    }
    prof_mark(ctx, group, 1, s);
    // levels >= ride_from are left to FAST's launch (orbfe_context::blur_ride_from); the lower ones are blurred beside the resize
    // that reads them, as far as the pyramid's launches reach
    const int ride_from = blur_ride_from_of(ctx, n_images);
    const int blurred = orbfe_launch_pyramid(cfg, buf, n_images, ctx->fuse_blur, s, ride_from);
    prof_mark(ctx, group, 2, s);
    // the levels still unblurred ride in FAST's launch as the last workgroups of each image's block list: FAST is bound by
    // instruction issue, these waves by memory latency.  ORBFE_NO_FUSE=1 gives every blur a launch of its own
    if (!ctx->fuse_blur) orbfe_launch_blur(cfg, buf, n_images, blurred, s);
    prof_mark(ctx, group, 3, s);
    orbfe_launch_fast(cfg, buf, n_images, ctx->use_octree3, s, ctx->fuse_blur ? blurred : cfg.nlevels);
    prof_mark(ctx, group, 4, s);
    if (ctx->use_octree3) orbfe_launch_octree3(cfg, buf, n_images, ctx->ot_sort_cap, ctx->ot3_lds, ctx->ot3_nodes_in_hbm, s);
    else orbfe_launch_octree_generic(cfg, buf, n_images, s);
#ifdef ORBFE_PROFILE_CUTS
    { // EXPERIMENT: see dbg_sort_sel_kernel (orbfe_describe.hip)
        extern void orbfe_launch_dbg_sort_sel(const DeviceConfig &, const DeviceBuffers &, int, int, hipStream_t);
        static const int dbg_sort = getenv("ORBFE_DBG_SORT_SEL") ? atoi(getenv("ORBFE_DBG_SORT_SEL")) : 0;
        if (dbg_sort) orbfe_launch_dbg_sort_sel(cfg, buf, n_images, dbg_sort, s);
    }
#endif
    prof_mark(ctx, group, 5, s);
    orbfe_launch_describe(cfg, buf, n_images, n_pairs > 0, s);
    prof_mark(ctx, group, 6, s);
    if (n_pairs > 0) {
        if (cfg.half_patch != 15) orbfe_launch_stereo_rowlists(cfg, buf, n_pairs, s); // describe_kernel (the reference's patch size) builds them in its own launch
        orbfe_launch_stereo_match(cfg, buf, n_pairs, s);
        prof_mark(ctx, group, 7, s);
        orbfe_launch_stereo_median(cfg, buf, n_pairs, s);
        prof_mark(ctx, group, 8, s);
    }
}

// Images (or pairs) are independent, so a batch is cut into `groups` contiguous sub-batches whose stage
// chains run on separate streams: the latency / barrier-bound stages of one sub-batch (quadtree,
// describe) overlap the VALU-bound stages of another (FAST).  The caller's stream is the fork/join point.
static int enqueue_batch(orbfe_context *ctx, const uint8_t *d_images, int n_units, int imgs_per_unit, void *stream)
{
    hipStream_t s = pick_stream(ctx, stream);
    HIP_TRY(ctx, hipSetDevice(ctx->params.device));
    ctx->prof_now = ctx->profiling && (ctx->prof_seq++ % (unsigned)ctx->prof_every) == 0;
    const int n_images = n_units * imgs_per_unit;
    int G = ctx->groups < n_units ? ctx->groups : n_units;
    if (G < 1) G = 1;
    if (G == 1) {
        run_chain(ctx, d_images, 0, n_images, imgs_per_unit == 2 ? n_units : 0, s, 0);
    } else {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, s));
        int u0 = 0;
        for (int g = 0; g < G; g++) {
            const int nu = n_units / G + (g < n_units % G ? 1 : 0);
            hipStream_t sg = ctx->gstreams[g];
            HIP_TRY(ctx, hipStreamWaitEvent(sg, ctx->ev_fork, 0));
            run_chain(ctx, d_images, u0 * imgs_per_unit, nu * imgs_per_unit, imgs_per_unit == 2 ? nu : 0, sg, g);
            HIP_TRY(ctx, hipEventRecord(ctx->ev_join[g], sg));
            HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_join[g], 0));
            u0 += nu;
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    ctx->last_src = (ctx->inplace_ok && ctx->cfg.in_cn == 1 && !ctx->cfg.rm_on) ? d_images : nullptr;
    ctx->last_src_owned = ctx->last_src && ctx->last_src == ctx->d_in;
    ctx->last_images = n_images;
    ctx->epoch++;
    ctx->prof_groups = G;
    ctx->latest_foreign = s != ctx->stream;
    if (ctx->latest_foreign) HIP_TRY(ctx, hipEventRecord(ctx->ev_latest, s));
    if (ctx->prof_now) ctx->prof_calls++;
    return ORBFE_OK;
}

extern "C" int orbfe_quadtree_kernel(const orbfe_context *ctx)
try {
    if (!ctx) return 0;
    return ctx->use_octree3 ? 3 : 1;
} ORBFE_CATCH(nullptr)

extern "C" int orbfe_set_streams(orbfe_context *ctx, int groups)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || groups < 1 || groups > ORBFE_MAX_GROUPS) return fail(ctx, ORBFE_ERR_INVALID, "groups must be in [1, %d]", ORBFE_MAX_GROUPS);
    if (ctx->profiling) return fail(ctx, ORBFE_ERR_INVALID, "change the stream count before enabling profiling");
    HIP_TRY(ctx, hipSetDevice(ctx->params.device));
    if (!ctx->ev_fork) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    for (int g = 0; g < groups; g++) {
        if (!ctx->gstreams[g]) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->gstreams[g], hipStreamNonBlocking));
        if (!ctx->ev_join[g]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_join[g], hipEventDisableTiming));
    }
    ctx->groups = groups;
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// Level 0 of a batched call on packed grey images is the caller's buffer itself (round 4: no ingest copy).  The kernels need it until
// the call's work has finished; orbfe_fetch_pyramid(level 0) would need it AFTER that, which the library cannot know: it follows the
// pointer only for callers that state here that the images of an enqueue call stay valid (and unchanged) until their next call.
extern "C" int orbfe_set_input_retained(orbfe_context *ctx, int retained)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx) return ORBFE_ERR_INVALID;
    ctx->input_retained = retained != 0;
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

extern "C" int orbfe_enqueue_extract(orbfe_context *ctx, const uint8_t *d_images, int n_images, void *stream)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !d_images) return fail(ctx, ORBFE_ERR_INVALID, "null argument");
    if (n_images < 1 || n_images > ctx->params.max_images)
        return fail(ctx, ORBFE_ERR_CAPACITY, "n_images %d outside [1, %d]", n_images, ctx->params.max_images);
    return enqueue_batch(ctx, d_images, n_images, 1, stream);
} ORBFE_CATCH(ctx)

extern "C" int orbfe_enqueue_stereo(orbfe_context *ctx, const uint8_t *d_images, int n_pairs, void *stream)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !d_images) return fail(ctx, ORBFE_ERR_INVALID, "null argument");
    if (n_pairs < 1 || 2 * n_pairs > ctx->params.max_images)
        return fail(ctx, ORBFE_ERR_CAPACITY, "n_pairs %d needs max_images >= %d", n_pairs, 2 * n_pairs);
    return enqueue_batch(ctx, d_images, n_pairs, 2, stream);
} ORBFE_CATCH(ctx)

// the staging of the host entry points is sized by the input format: one packed image = image_bytes
static int resize_input_staging(orbfe_context *ctx, size_t image_bytes)
{
    uint8_t *nd = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&nd, (size_t)ctx->params.max_images * image_bytes));
    for (void *&q : ctx->allocs)
        if (q == ctx->d_in) q = nd;
    (void)hipFree(ctx->d_in);
    if (ctx->last_src == ctx->d_in) ctx->last_src = nullptr; // the latest call's level 0 lived there (orbfe_fetch_pyramid)
    ctx->d_in = nd;
    if (ctx->h_in) { (void)hipHostFree(ctx->h_in); ctx->h_in = nullptr; }
    ctx->cfg.in_image_bytes = image_bytes;
    return ORBFE_OK;
}

extern "C" int orbfe_set_rectification(orbfe_context *ctx, int side, const float *map_x, const float *map_y, int src_w, int src_h)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || side < 0 || side > 1) return fail(ctx, ORBFE_ERR_INVALID, "bad argument");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    DeviceConfig &c = ctx->cfg;
    if (!map_x || !map_y) { // clear this side; the left side switches rectification off
        if (c.rm_xy[side]) { (void)hipFree((void *)c.rm_xy[side]); (void)hipFree((void *)c.rm_a[side]); c.rm_xy[side] = nullptr; c.rm_a[side] = nullptr; }
        if (side == 0 && c.rm_on) {
            if (c.rm_xy[1]) { (void)hipFree((void *)c.rm_xy[1]); (void)hipFree((void *)c.rm_a[1]); c.rm_xy[1] = nullptr; c.rm_a[1] = nullptr; }
            c.rm_on = 0;
            return resize_input_staging(ctx, (size_t)ctx->params.width * ctx->params.height * c.in_cn);
        }
        return ORBFE_OK;
    }
    if (c.in_cn != 1) return fail(ctx, ORBFE_ERR_UNSUPPORTED, "rectification takes single-channel input");
    if (src_w < 1 || src_h < 1 || src_w > 32767 || src_h > 32767) return fail(ctx, ORBFE_ERR_INVALID, "bad source size");
    if (side == 1 && !c.rm_on) return fail(ctx, ORBFE_ERR_INVALID, "set the left (side 0) maps first");
    if (c.rm_on && (src_w != c.rm_sw || src_h != c.rm_sh)) {
        if (side == 1 || c.rm_xy[1]) return fail(ctx, ORBFE_ERR_UNSUPPORTED, "both sides must share one source size");
    }
    // RemapInvoker's conversion of the float maps (imgwarp.cpp): sx = cvRound(mapx * INTER_TAB_SIZE), integer part
    // saturated to short, fraction index = (sy & 31) * 32 + (sx & 31)
    const size_t n = (size_t)ctx->params.width * ctx->params.height;
    std::vector<uint32_t> xy(n);
    std::vector<uint16_t> al(n);
    for (size_t i = 0; i < n; i++) {
        const int ix = (int)lrintf(map_x[i] * 32.0f), iy = (int)lrintf(map_y[i] * 32.0f);
        const int sx = std::min(std::max(ix >> 5, -32768), 32767), sy = std::min(std::max(iy >> 5, -32768), 32767);
        xy[i] = (uint32_t)(uint16_t)(int16_t)sx | ((uint32_t)(uint16_t)(int16_t)sy << 16);
        al[i] = (uint16_t)((iy & 31) * 32 + (ix & 31));
    }
    uint32_t *dxy = nullptr; uint16_t *da = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&dxy, n * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc((void **)&da, n * sizeof(uint16_t)));
    HIP_TRY(ctx, hipMemcpy(dxy, xy.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(da, al.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice));
    if (c.rm_xy[side]) { (void)hipFree((void *)c.rm_xy[side]); (void)hipFree((void *)c.rm_a[side]); }
    c.rm_xy[side] = dxy; c.rm_a[side] = da;
    if (side == 0) {
        c.rm_on = 1; c.rm_sw = src_w; c.rm_sh = src_h;
        return resize_input_staging(ctx, (size_t)src_w * src_h);
    }
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

extern "C" int orbfe_set_input_format(orbfe_context *ctx, int channels, int rgb_order, int legacy_weights)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx) return ORBFE_ERR_INVALID;
    if (channels != 1 && channels != 3 && channels != 4) return fail(ctx, ORBFE_ERR_INVALID, "channels must be 1, 3 or 4");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->cfg.rm_on && channels != 1) return fail(ctx, ORBFE_ERR_UNSUPPORTED, "rectification takes single-channel input");
    if (channels != ctx->cfg.in_cn) {
        const int rc = resize_input_staging(ctx, (size_t)ctx->params.width * ctx->params.height * channels);
        if (rc != ORBFE_OK) return rc;
    }
    // color_rgb.simd.hpp RGB2Gray<uchar>: RY15 / GY15 / BY15 with 15 fraction bits; OpenCV 3.x: R2Y / G2Y / B2Y with 14
    const int cr = legacy_weights ? 4899 : 9798, cg = legacy_weights ? 9617 : 19235, cb = legacy_weights ? 1868 : 3735;
    ctx->cfg.in_cn = channels;
    ctx->cfg.in_coef[0] = rgb_order ? cr : cb;
    ctx->cfg.in_coef[1] = cg;
    ctx->cfg.in_coef[2] = rgb_order ? cb : cr;
    ctx->cfg.in_shift = legacy_weights ? 14 : 15;
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

extern "C" int orbfe_set_distortion(orbfe_context *ctx, const float *dist, int n)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || (n != 0 && n != 4 && n != 5) || (n > 0 && !dist)) return fail(ctx, ORBFE_ERR_INVALID, "distortion needs 0, 4 or 5 coefficients (k1 k2 p1 p2 [k3])");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cfg.n_dist = n;
    for (int i = 0; i < 5; i++) ctx->cfg.dist[i] = i < n ? dist[i] : 0.f;
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// scratch for the undistortion entry points: [0, n) input keypoints, [n, 2n) output
static int undistort_on_device(orbfe_context *ctx, const orbfe_keypoint *kps, const KeyPointPOD *d_src, int n, orbfe_keypoint *kps_un)
{
    if (n <= 0) return ORBFE_OK;
    const size_t need = sizeof(KeyPointPOD) * (size_t)n * 2;
    if (ctx->d_und_bytes < need) {
        if (ctx->d_und) (void)hipFree(ctx->d_und);
        ctx->d_und = nullptr; ctx->d_und_bytes = 0;
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_und, need));
        ctx->d_und_bytes = need;
    }
    KeyPointPOD *d_in = (KeyPointPOD *)ctx->d_und, *d_out = d_in + n;
    if (kps) { HIP_TRY(ctx, hipMemcpyAsync(d_in, kps, sizeof(KeyPointPOD) * n, hipMemcpyHostToDevice, ctx->stream)); d_src = d_in; }
    orbfe_launch_undistort(ctx->cfg, d_src, d_out, n, ctx->stream);
    HIP_TRY(ctx, hipMemcpyAsync(kps_un, d_out, sizeof(KeyPointPOD) * n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ORBFE_OK;
}

extern "C" int orbfe_undistort_keypoints(orbfe_context *ctx, const orbfe_keypoint *kps, int n, orbfe_keypoint *kps_un)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || n < 0 || (n > 0 && (!kps || !kps_un))) return fail(ctx, ORBFE_ERR_INVALID, "bad argument");
    return undistort_on_device(ctx, kps, nullptr, n, kps_un);
} ORBFE_CATCH(ctx)

extern "C" int orbfe_fetch_keys_un(orbfe_context *ctx, int image, orbfe_keypoint *kps_un, int cap, int *n)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || image < 0 || image >= ctx->params.max_images || !n) return fail(ctx, ORBFE_ERR_INVALID, "bad argument");
    { const int rcw = wait_latest(ctx); if (rcw != ORBFE_OK) return rcw; }
    int cnt = 0;
    HIP_TRY(ctx, hipMemcpy(&cnt, ctx->buf.kp_cnt + image, sizeof(int), hipMemcpyDeviceToHost));
    *n = cnt;
    if (cnt > cap) return fail(ctx, ORBFE_ERR_CAPACITY, "caller buffer holds %d keypoints, image has %d", cap, cnt);
    if (cnt > 0 && !kps_un) return fail(ctx, ORBFE_ERR_INVALID, "null output");
    return undistort_on_device(ctx, nullptr, (const KeyPointPOD *)ctx->buf.kps + (size_t)image * ctx->cfg.sel_total, cnt, kps_un);
} ORBFE_CATCH(ctx)

extern "C" int orbfe_image_bounds(orbfe_context *ctx, float *bounds)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !bounds) return fail(ctx, ORBFE_ERR_INVALID, "bad argument");
    const float cols = (float)ctx->params.width, rows = (float)ctx->params.height;
    if (ctx->cfg.n_dist == 0 || ctx->cfg.dist[0] == 0.0f) { // src/Frame.cc:455-461
        bounds[0] = 0.f; bounds[1] = cols; bounds[2] = 0.f; bounds[3] = rows;
        return ORBFE_OK;
    }
    orbfe_keypoint c[4] = {}, o[4];
    c[1].x = cols; c[2].y = rows; c[3].x = cols; c[3].y = rows; // :439-442
    const int rc = undistort_on_device(ctx, c, nullptr, 4, o);
    if (rc != ORBFE_OK) return rc;
    bounds[0] = std::min(o[0].x, o[2].x);
    bounds[1] = std::max(o[1].x, o[3].x);
    bounds[2] = std::min(o[0].y, o[1].y);
    bounds[3] = std::max(o[2].y, o[3].y);
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

extern "C" int orbfe_synchronize(orbfe_context *ctx, void *stream)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx) return ORBFE_ERR_INVALID;
    HIP_TRY(ctx, hipStreamSynchronize(pick_stream(ctx, stream)));
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

extern "C" int orbfe_fetch_counts(orbfe_context *ctx, int32_t *counts, int n_images)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !counts || n_images < 1 || n_images > ctx->params.max_images) return fail(ctx, ORBFE_ERR_INVALID, "bad argument");
    { const int rcw = wait_latest(ctx); if (rcw != ORBFE_OK) return rcw; }
    HIP_TRY(ctx, hipMemcpy(counts, ctx->buf.kp_cnt, sizeof(int32_t) * n_images, hipMemcpyDeviceToHost));
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

extern "C" int orbfe_fetch_batch_async(orbfe_context *ctx, int n_images, orbfe_keypoint *kps, uint8_t *desc, int32_t *counts,
                                       float *u_right, float *depth, void *stream)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || n_images < 1 || n_images > ctx->params.max_images) return fail(ctx, ORBFE_ERR_INVALID, "bad argument");
    hipStream_t s = pick_stream(ctx, stream);
    const size_t n = (size_t)n_images * ctx->cfg.sel_total;
    if (kps) HIP_TRY(ctx, hipMemcpyAsync(kps, ctx->buf.kps, sizeof(KeyPointPOD) * n, hipMemcpyDeviceToHost, s));
    if (desc) HIP_TRY(ctx, hipMemcpyAsync(desc, ctx->buf.desc, (size_t)32 * n, hipMemcpyDeviceToHost, s));
    if (counts) HIP_TRY(ctx, hipMemcpyAsync(counts, ctx->buf.kp_cnt, sizeof(int32_t) * n_images, hipMemcpyDeviceToHost, s));
    if (u_right) HIP_TRY(ctx, hipMemcpyAsync(u_right, ctx->buf.u_right, sizeof(float) * n, hipMemcpyDeviceToHost, s));
    if (depth) HIP_TRY(ctx, hipMemcpyAsync(depth, ctx->buf.depth, sizeof(float) * n, hipMemcpyDeviceToHost, s));
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// ---- packed result block (include/orbfe.h: orbfe_packed_layout) ----
static int packed_layout(const orbfe_context *ctx, int n_images, int flags, orbfe_packed_layout *o)
{
    if (!ctx || !o || n_images < 1 || n_images > ctx->params.max_images || (flags & ~(ORBFE_PACK_STEREO | ORBFE_PACK_LEFT_ONLY | ORBFE_PACK_DIRECT)))
        return ORBFE_ERR_INVALID;
    if ((flags & (ORBFE_PACK_STEREO | ORBFE_PACK_LEFT_ONLY)) && (n_images & 1)) return ORBFE_ERR_INVALID; // pairs L0 R0 L1 R1 ...
    const size_t cap = (size_t)ctx->cfg.sel_total, nl = (size_t)ctx->cfg.nlevels;
    const size_t n_out = (flags & ORBFE_PACK_LEFT_ONLY) ? (size_t)n_images / 2 : (size_t)n_images;
    const size_t n_pairs = (flags & ORBFE_PACK_STEREO) ? (size_t)n_images / 2 : 0;
    auto up = [](size_t v) { return (v + 63) & ~(size_t)63; };
    memset(o, 0, sizeof(*o));
    o->n_images_out = (int32_t)n_out; o->capacity = (int32_t)cap; o->nlevels = (int32_t)nl; o->n_pairs = (int32_t)n_pairs; o->flags = flags;
    size_t off = 0;
    o->counts_off = off; off = up(off + 4 * n_out);
    o->level_counts_off = off; off = up(off + 4 * n_out * nl);
    o->xy_off = off; off = up(off + 4 * n_out * cap);
    o->angle_off = off; off = up(off + 4 * n_out * cap);
    o->response_off = off; off = up(off + n_out * cap);
    o->desc_off = off; off = up(off + 32 * n_out * cap);
    o->u_right_off = off; off = up(off + 4 * n_pairs * cap);
    o->depth_off = off; off = up(off + 4 * n_pairs * cap);
    o->bytes = off;
    return ORBFE_OK;
}

extern "C" int orbfe_get_packed_layout(const orbfe_context *ctx, int n_images, int flags, orbfe_packed_layout *out)
try {
    return packed_layout(ctx, n_images, flags, out);
} ORBFE_CATCH(nullptr)

extern "C" int orbfe_fetch_batch_packed(orbfe_context *ctx, int n_images, int flags, void *host_block, size_t host_bytes, void *stream)
try {
    ORBFE_ENTRY(ctx);
    orbfe_packed_layout lay;
    if (packed_layout(ctx, n_images, flags, &lay) != ORBFE_OK || !host_block) return fail(ctx, ORBFE_ERR_INVALID, "bad argument");
    if (n_images > ctx->last_images) return fail(ctx, ORBFE_ERR_INVALID, "the latest extraction call filled %d image slots, %d asked for", ctx->last_images, n_images);
    if (host_bytes < lay.bytes) return fail(ctx, ORBFE_ERR_CAPACITY, "packed block needs %zu bytes, caller offers %zu", lay.bytes, host_bytes);
    HIP_TRY(ctx, hipSetDevice(ctx->params.device));
    PackedOffsets po = {lay.counts_off, lay.level_counts_off, lay.xy_off, lay.angle_off, lay.response_off, lay.desc_off, lay.u_right_off, lay.depth_off};
    if (flags & ORBFE_PACK_DIRECT) {
        // the caller's block is pinned host memory mapped into this device's address space: the gather kernel stores into it
        // across the link itself (posted writes), so no copy engine is involved -- on the measured link an upload and a download
        // queued on the copy engines take the SUM of their times, while a kernel's stores run beside an upload
        {   // checked on every call (a microsecond): the block a caller passes today may not be the pinned one it passed yesterday at the same address
            hipPointerAttribute_t at;
            void *dp = nullptr;
            if (hipPointerGetAttributes(&at, host_block) != hipSuccess || at.type != hipMemoryTypeHost || hipHostGetDevicePointer(&dp, host_block, 0) != hipSuccess || dp != host_block) {
                (void)hipGetLastError();
                return fail(ctx, ORBFE_ERR_INVALID, "ORBFE_PACK_DIRECT needs pinned host memory that the device addresses at the same pointer (hipHostMalloc / hipHostRegister)");
            }
            // ... and the pinned range must hold the whole block: the gather kernel stores lay.bytes from the pointer on
            hipDeviceptr_t base = nullptr;
            size_t range = 0;
            if (hipMemGetAddressRange(&base, &range, (hipDeviceptr_t)host_block) == hipSuccess) {
                if ((const uint8_t *)host_block + lay.bytes > (const uint8_t *)base + range)
                    return fail(ctx, ORBFE_ERR_CAPACITY, "ORBFE_PACK_DIRECT: the pinned range ends %zu bytes after the pointer, the block needs %zu", (size_t)((const uint8_t *)base + range - (const uint8_t *)host_block), lay.bytes);
            } else {
                (void)hipGetLastError(); // the runtime cannot tell (registered memory on some versions): the caller's host_bytes stands
            }
        }
        orbfe_launch_pack_results(ctx->cfg, ctx->buf, (uint8_t *)host_block, po, lay.n_images_out, (flags & ORBFE_PACK_LEFT_ONLY) ? 2 : 1, (flags & ORBFE_PACK_STEREO) != 0, pick_stream(ctx, stream));
        HIP_TRY(ctx, hipGetLastError());
        return ORBFE_OK;
    }
    if (ctx->d_pack_bytes < lay.bytes) { // sized once for the largest block this context can be asked for
        orbfe_packed_layout mx;
        packed_layout(ctx, ctx->params.max_images, 0, &mx); // every image slot, plus uRight / depth of half of them
        const size_t want = mx.bytes + 2 * (((size_t)4 * ((size_t)ctx->params.max_images / 2 + 1) * (size_t)ctx->cfg.sel_total + 63) & ~(size_t)63);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->ev_pack_set) HIP_TRY(ctx, hipEventSynchronize(ctx->ev_pack)); // a copy out of the old staging may be in flight on a caller's stream
        if (ctx->d_pack) (void)hipFree(ctx->d_pack);
        ctx->d_pack = nullptr; ctx->d_pack_bytes = 0;
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_pack, want));
        ctx->d_pack_bytes = want;
    }
    hipStream_t s = pick_stream(ctx, stream);
    // ONE staging buffer per context: a fetch queued on another stream while the previous copy is still in flight must not refill it
    if (!ctx->ev_pack) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_pack, hipEventDisableTiming));
    if (ctx->ev_pack_set) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_pack, 0));
    orbfe_launch_pack_results(ctx->cfg, ctx->buf, ctx->d_pack, po, lay.n_images_out, (flags & ORBFE_PACK_LEFT_ONLY) ? 2 : 1, (flags & ORBFE_PACK_STEREO) != 0, s);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(host_block, ctx->d_pack, lay.bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_pack, s));
    ctx->ev_pack_set = true;
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// Host side of the packed record: cv::KeyPoint from (level x, level y, octave, score, angle) with the operations of
// src/ORBextractor.cc:838 (size = scaledPatchSize, an int stored as float), :909-915 (pt *= mvScaleFactor[level] for level != 0;
// one IEEE float product each, as describe_kernel's __fmul_rn), cv::KeyPoint's response = (float)score (cv::FAST, :803-808),
// class_id = -1.  Pure host code: no device call, no lock.
extern "C" int orbfe_expand_packed(const orbfe_context *ctx, const void *host_block, const orbfe_packed_layout *lay, int out_image,
                                   orbfe_keypoint *kps, int cap, int *n)
try {
    if (!ctx || !host_block || !lay || !n || out_image < 0 || out_image >= lay->n_images_out || lay->capacity != ctx->cfg.sel_total || lay->nlevels != ctx->cfg.nlevels)
        return ORBFE_ERR_INVALID;
    const uint8_t *b = (const uint8_t *)host_block;
    const int cnt = ((const int32_t *)(b + lay->counts_off))[out_image];
    *n = cnt;
    if (cnt < 0 || cnt > lay->capacity) return ORBFE_ERR_INVALID;
    if (cnt > cap) return ORBFE_ERR_CAPACITY;
    if (cnt > 0 && !kps) return ORBFE_ERR_INVALID;
    const int32_t *lc = (const int32_t *)(b + lay->level_counts_off) + (size_t)out_image * lay->nlevels;
    const uint32_t *xy = (const uint32_t *)(b + lay->xy_off) + (size_t)out_image * lay->capacity;
    const float *ang = (const float *)(b + lay->angle_off) + (size_t)out_image * lay->capacity;
    const uint8_t *rs = b + lay->response_off + (size_t)out_image * lay->capacity;
    int j = 0;
    for (int l = 0; l < lay->nlevels && j < cnt; l++) {
        const float scale = ctx->scale[l], size = (float)ctx->cfg.lv[l].scaled_patch;
        int c = lc[l];
        if (c < 0 || j + c > cnt) return ORBFE_ERR_INVALID;
        for (; c > 0; c--, j++) {
            float px = (float)(xy[j] & 0xffffu), py = (float)(xy[j] >> 16);
            if (l != 0) { px = px * scale; py = py * scale; }
            orbfe_keypoint &k = kps[j];
            k.x = px; k.y = py; k.size = size; k.angle = ang[j]; k.response = (float)rs[j]; k.octave = l; k.class_id = -1;
        }
    }
    return j == cnt ? ORBFE_OK : ORBFE_ERR_INVALID;
} ORBFE_CATCH(nullptr)

extern "C" int orbfe_fetch_image(orbfe_context *ctx, int image, orbfe_keypoint *kps, uint8_t *desc,
                                 float *u_right, float *depth, int cap, int *n)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || image < 0 || image >= ctx->params.max_images || !n) return fail(ctx, ORBFE_ERR_INVALID, "bad argument");
    { const int rcw = wait_latest(ctx); if (rcw != ORBFE_OK) return rcw; }
    int cnt = 0, status = 0;
    HIP_TRY(ctx, hipMemcpy(&cnt, ctx->buf.kp_cnt + image, sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(&status, ctx->buf.status + image, sizeof(int), hipMemcpyDeviceToHost));
    if (status != 0) return fail(ctx, ORBFE_ERR_CAPACITY, "device-side capacity overflow (status %d) on image %d", status, image);
    *n = cnt;
    if (cnt > cap) return fail(ctx, ORBFE_ERR_CAPACITY, "caller buffers hold %d keypoints, image has %d", cap, cnt);
    const size_t st = (size_t)ctx->cfg.sel_total;
    if (cnt > 0) {
        if (kps) HIP_TRY(ctx, hipMemcpy(kps, (const KeyPointPOD *)ctx->buf.kps + image * st, sizeof(KeyPointPOD) * cnt, hipMemcpyDeviceToHost));
        if (desc) HIP_TRY(ctx, hipMemcpy(desc, ctx->buf.desc + image * st * 32, (size_t)32 * cnt, hipMemcpyDeviceToHost));
        if (u_right) HIP_TRY(ctx, hipMemcpy(u_right, ctx->buf.u_right + image * st, sizeof(float) * cnt, hipMemcpyDeviceToHost));
        if (depth) HIP_TRY(ctx, hipMemcpy(depth, ctx->buf.depth + image * st, sizeof(float) * cnt, hipMemcpyDeviceToHost));
    }
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

extern "C" int orbfe_debug_timestamps(orbfe_context *ctx, long long *dst, int n)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !dst || n < 0 || n > 4096) return ORBFE_ERR_INVALID;
    HIP_TRY(ctx, hipMemcpy(dst, ctx->buf.dbg_ts, sizeof(long long) * n, hipMemcpyDeviceToHost));
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

extern "C" int orbfe_device_buffers(orbfe_context *ctx, void **kps, void **desc, void **counts, void **u_right, void **depth)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx) return ORBFE_ERR_INVALID;
    if (kps) *kps = ctx->buf.kps;
    if (desc) *desc = ctx->buf.desc;
    if (counts) *counts = ctx->buf.kp_cnt;
    if (u_right) *u_right = ctx->buf.u_right;
    if (depth) *depth = ctx->buf.depth;
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// Layout of the pinned output block of the single-frame entry points: counts and status words of the (up to two)
// images, then the keypoint / descriptor / uRight / depth arrays at their device capacity (sel_total per image).
struct HostOut {
    size_t cnt, status, kps, desc, ur, depth, bytes;
};

static HostOut host_out_layout(const orbfe_context *ctx)
{
    const size_t st = (size_t)ctx->cfg.sel_total, ni = ctx->params.max_images < 2 ? 1 : 2;
    HostOut o;
    o.cnt = 0;
    o.status = 16;
    o.kps = 32;
    o.desc = o.kps + ((ni * st * sizeof(KeyPointPOD) + 15) & ~(size_t)15);
    o.ur = o.desc + ni * st * 32;
    o.depth = o.ur + ((st * sizeof(float) + 15) & ~(size_t)15);
    o.bytes = o.depth + st * sizeof(float);
    return o;
}

static int ensure_host_stage(orbfe_context *ctx, bool want_depth)
{
    const size_t px = (size_t)ctx->params.width * ctx->params.height, ni = ctx->params.max_images < 2 ? 1 : 2;
    if (!ctx->h_in) HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_in, ni * ctx->cfg.in_image_bytes, hipHostMallocDefault));
    if (!ctx->h_out) HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_out, host_out_layout(ctx).bytes, hipHostMallocDefault));
    if (want_depth && !ctx->h_depth_in) HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_depth_in, px * sizeof(float), hipHostMallocDefault));
    if (want_depth && !ctx->d_depth_in) HIP_TRY(ctx, hipMalloc((void **)&ctx->d_depth_in, px * sizeof(float)));
    return ORBFE_OK;
}

// Packs the caller's rows into the pinned block and queues one linear copy (a pitched copy from pageable memory is
// executed row by row by the runtime: 3 ms for a 1241x376 image; splitting the copy into chunks to overlap packing
// and DMA costs more in submissions, about 15 us each, than it hides).
static int stage_rows(orbfe_context *ctx, void *d_dst, uint8_t *h_stage, const void *src, size_t row, int h, size_t stride)
{
    if (stride == row) memcpy(h_stage, src, row * h);
    else for (int y = 0; y < h; y++) memcpy(h_stage + (size_t)y * row, (const uint8_t *)src + (size_t)y * stride, row);
    HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_stage, row * h, hipMemcpyHostToDevice, ctx->stream));
    return ORBFE_OK;
}

static int upload_image(orbfe_context *ctx, int slot, const uint8_t *img, int w, int h, size_t stride)
{
    const int ew = ctx->cfg.rm_on ? ctx->cfg.rm_sw : ctx->params.width, eh = ctx->cfg.rm_on ? ctx->cfg.rm_sh : ctx->params.height;
    if (w != ew || h != eh)
        return fail(ctx, ORBFE_ERR_UNSUPPORTED, "image is %dx%d, context expects %dx%d%s", w, h, ew, eh, ctx->cfg.rm_on ? " (unrectified source size)" : "");
    const size_t row = (size_t)w * ctx->cfg.in_cn; // bytes per packed row of the context's input format
    if (stride < row) return fail(ctx, ORBFE_ERR_INVALID, "stride smaller than a row (%d px x %d channels)", w, ctx->cfg.in_cn);
    const size_t px = row * h;
    return stage_rows(ctx, ctx->d_in + (size_t)slot * px, ctx->h_in + (size_t)slot * px, img, row, h, stride);
}

// Queues the device-to-host copies of `nimg` images' results into the pinned block, waits once, and hands the
// caller its arrays.  `with_depth`: image 0 carries uRight / depth.
static int download_frame(orbfe_context *ctx, int nimg, bool with_depth)
{
    const HostOut o = host_out_layout(ctx);
    const size_t st = (size_t)ctx->cfg.sel_total;
    uint8_t *ho = ctx->h_out;
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(ho + o.cnt, ctx->buf.kp_cnt, sizeof(int) * nimg, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(ho + o.status, ctx->buf.status, sizeof(int) * nimg, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(ho + o.kps, ctx->buf.kps, sizeof(KeyPointPOD) * st * nimg, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(ho + o.desc, ctx->buf.desc, (size_t)32 * st * nimg, hipMemcpyDeviceToHost, s));
    if (with_depth) {
        HIP_TRY(ctx, hipMemcpyAsync(ho + o.ur, ctx->buf.u_right, sizeof(float) * st, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipMemcpyAsync(ho + o.depth, ctx->buf.depth, sizeof(float) * st, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));
    ctx->slot_cnt.assign((const int *)(ho + o.cnt), (const int *)(ho + o.cnt) + nimg);
    ctx->slot_cnt_epoch = ctx->epoch;
    return ORBFE_OK;
}

static int hand_over(orbfe_context *ctx, int image, orbfe_keypoint *kps, uint8_t *desc, float *u_right, float *depth,
                     int cap, int *n)
{
    const HostOut o = host_out_layout(ctx);
    const size_t st = (size_t)ctx->cfg.sel_total;
    const uint8_t *ho = ctx->h_out;
    const int cnt = ((const int *)(ho + o.cnt))[image], status = ((const int *)(ho + o.status))[image];
    if (status != 0) return fail(ctx, ORBFE_ERR_CAPACITY, "device-side capacity overflow (status %d) on image %d", status, image);
    *n = cnt;
    if (cnt > cap) return fail(ctx, ORBFE_ERR_CAPACITY, "caller buffers hold %d keypoints, image has %d", cap, cnt);
    if (cnt <= 0) return ORBFE_OK;
    if (kps) memcpy(kps, ho + o.kps + image * st * sizeof(KeyPointPOD), sizeof(KeyPointPOD) * cnt);
    if (desc) memcpy(desc, ho + o.desc + image * st * 32, (size_t)32 * cnt);
    if (u_right) memcpy(u_right, ho + o.ur, sizeof(float) * cnt);
    if (depth) memcpy(depth, ho + o.depth, sizeof(float) * cnt);
    return ORBFE_OK;
}

extern "C" int orbfe_extract(orbfe_context *ctx, const uint8_t *img, int w, int h, size_t stride,
                             orbfe_keypoint *kps, uint8_t *desc, int cap, int *n)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !n) return fail(ctx, ORBFE_ERR_INVALID, "null argument");
    if (!img || w <= 0 || h <= 0) { *n = 0; return ORBFE_OK; } // _image.empty(): src/ORBextractor.cc:861-862
    int rc = ensure_host_stage(ctx, false);
    if (rc != ORBFE_OK) return rc;
    rc = upload_image(ctx, 0, img, w, h, stride);
    if (rc != ORBFE_OK) return rc;
    rc = orbfe_enqueue_extract(ctx, ctx->d_in, 1, nullptr);
    if (rc != ORBFE_OK) return rc;
    rc = download_frame(ctx, 1, false);
    if (rc != ORBFE_OK) return rc;
    return hand_over(ctx, 0, kps, desc, nullptr, nullptr, cap, n);
} ORBFE_CATCH(ctx)

extern "C" int orbfe_device_count(void)
try {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
} ORBFE_CATCH(nullptr)

// Host-fed batch in one call: what a single-process multi-device host (orbslam2_amd/host/multi_device.h) runs per context.
extern "C" int orbfe_stereo_batch(orbfe_context *ctx, const uint8_t *images, int n_pairs, orbfe_keypoint *kps, uint8_t *desc, int32_t *counts,
                                  float *u_right, float *depth)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !images || !counts) return fail(ctx, ORBFE_ERR_INVALID, "null argument");
    if (n_pairs < 1 || 2 * n_pairs > ctx->params.max_images) return fail(ctx, ORBFE_ERR_CAPACITY, "n_pairs %d needs max_images >= %d", n_pairs, 2 * n_pairs);
    HIP_TRY(ctx, hipSetDevice(ctx->params.device));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_in, images, (size_t)2 * n_pairs * ctx->cfg.in_image_bytes, hipMemcpyHostToDevice, ctx->stream));
    int rc = orbfe_enqueue_stereo(ctx, ctx->d_in, n_pairs, nullptr);
    if (rc != ORBFE_OK) return rc;
    rc = orbfe_fetch_batch_async(ctx, 2 * n_pairs, kps, desc, counts, u_right, depth, nullptr);
    if (rc != ORBFE_OK) return rc;
    // the results are in the caller's buffers once the stream is idle; the status words are read after that, synchronously (an
    // asynchronous copy into a local buffer would outlive it on an early return)
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int> status((size_t)2 * n_pairs);
    HIP_TRY(ctx, hipMemcpy(status.data(), ctx->buf.status, sizeof(int) * 2 * n_pairs, hipMemcpyDeviceToHost));
    for (int i = 0; i < 2 * n_pairs; i++)
        if (status[i] != 0) return fail(ctx, ORBFE_ERR_CAPACITY, "device-side capacity overflow (status %d) on image %d", status[i], i);
    ctx->slot_cnt.assign(counts, counts + 2 * n_pairs);
    ctx->slot_cnt_epoch = ctx->epoch;
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

// The same batch with the packed result block (one gather kernel + one copy, about two thirds the bytes; orbfe_expand_packed on the host).
extern "C" int orbfe_stereo_batch_packed(orbfe_context *ctx, const uint8_t *images, int n_pairs, int flags, void *host_block, size_t host_bytes)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !images || !host_block) return fail(ctx, ORBFE_ERR_INVALID, "null argument");
    if (n_pairs < 1 || 2 * n_pairs > ctx->params.max_images) return fail(ctx, ORBFE_ERR_CAPACITY, "n_pairs %d needs max_images >= %d", n_pairs, 2 * n_pairs);
    HIP_TRY(ctx, hipSetDevice(ctx->params.device));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_in, images, (size_t)2 * n_pairs * ctx->cfg.in_image_bytes, hipMemcpyHostToDevice, ctx->stream));
    int rc = orbfe_enqueue_stereo(ctx, ctx->d_in, n_pairs, nullptr);
    if (rc != ORBFE_OK) return rc;
    rc = orbfe_fetch_batch_packed(ctx, 2 * n_pairs, flags | ORBFE_PACK_STEREO, host_block, host_bytes, nullptr);
    if (rc != ORBFE_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int> status((size_t)2 * n_pairs);
    HIP_TRY(ctx, hipMemcpy(status.data(), ctx->buf.status, sizeof(int) * 2 * n_pairs, hipMemcpyDeviceToHost));
    for (int i = 0; i < 2 * n_pairs; i++)
        if (status[i] != 0) return fail(ctx, ORBFE_ERR_CAPACITY, "device-side capacity overflow (status %d) on image %d", status[i], i);
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

extern "C" int orbfe_stereo_frame(orbfe_context *ctx, const uint8_t *left, const uint8_t *right,
                                  int w, int h, size_t stride,
                                  orbfe_keypoint *kps_left, uint8_t *desc_left, int *n_left,
                                  orbfe_keypoint *kps_right, uint8_t *desc_right, int *n_right,
                                  float *u_right, float *depth, int cap)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !n_left || !n_right) return fail(ctx, ORBFE_ERR_INVALID, "null argument");
    if (!left || !right || w <= 0 || h <= 0) { *n_left = 0; *n_right = 0; return ORBFE_OK; }
    if (ctx->params.max_images < 2) return fail(ctx, ORBFE_ERR_CAPACITY, "stereo needs max_images >= 2");
    int rc = ensure_host_stage(ctx, false);
    if (rc != ORBFE_OK) return rc;
    rc = upload_image(ctx, 0, left, w, h, stride);
    if (rc != ORBFE_OK) return rc;
    rc = upload_image(ctx, 1, right, w, h, stride);
    if (rc != ORBFE_OK) return rc;
    rc = orbfe_enqueue_stereo(ctx, ctx->d_in, 1, nullptr);
    if (rc != ORBFE_OK) return rc;
    rc = download_frame(ctx, 2, true);
    if (rc != ORBFE_OK) return rc;
    rc = hand_over(ctx, 0, kps_left, desc_left, u_right, depth, cap, n_left);
    if (rc != ORBFE_OK) return rc;
    return hand_over(ctx, 1, kps_right, desc_right, nullptr, nullptr, cap, n_right);
} ORBFE_CATCH(ctx)

// Common body of the two RGB-D entry points.  The depth rows are packed while the extraction kernels already run:
// the map is only sampled at the keypoints, at the very end of the chain.
static double host_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int rgbd_frame_impl(orbfe_context *ctx, const uint8_t *gray, const void *depth_img, size_t px_bytes, float factor,
                           int w, int h, size_t gray_stride, size_t depth_stride,
                           orbfe_keypoint *kps, uint8_t *desc, int *n, float *u_right, float *depth, int cap)
{
    if (!ctx || !n) return fail(ctx, ORBFE_ERR_INVALID, "null argument");
    if (!gray || !depth_img || w <= 0 || h <= 0) { *n = 0; return ORBFE_OK; }
    const size_t row = px_bytes * (size_t)w;
    if (depth_stride < row) return fail(ctx, ORBFE_ERR_INVALID, "depth stride smaller than a row");
    if (ctx->cfg.rm_on) return fail(ctx, ORBFE_ERR_UNSUPPORTED, "RGB-D frames with rectification maps are not supported (the depth map would need the same warp)");
    static const bool trace = getenv("ORBFE_HOST_TRACE") != nullptr;
    double t[6] = {};
    t[0] = host_ms();
    int rc = ensure_host_stage(ctx, true);
    if (rc != ORBFE_OK) return rc;
    rc = upload_image(ctx, 0, gray, w, h, gray_stride);
    if (rc != ORBFE_OK) return rc;
    t[1] = host_ms();
    rc = orbfe_enqueue_extract(ctx, ctx->d_in, 1, nullptr);
    if (rc != ORBFE_OK) return rc;
    t[2] = host_ms();
    rc = stage_rows(ctx, ctx->d_depth_in, (uint8_t *)ctx->h_depth_in, depth_img, row, h, depth_stride);
    if (rc != ORBFE_OK) return rc;
    t[3] = host_ms();
    if (px_bytes == 2) orbfe_launch_rgbd_u16(ctx->cfg, ctx->buf, (const uint16_t *)ctx->d_depth_in, (size_t)w, factor, 0, ctx->stream);
    else orbfe_launch_rgbd(ctx->cfg, ctx->buf, ctx->d_depth_in, (size_t)w, 0, ctx->stream);
    rc = download_frame(ctx, 1, true);
    if (rc != ORBFE_OK) return rc;
    t[4] = host_ms();
    rc = hand_over(ctx, 0, kps, desc, u_right, depth, cap, n);
    t[5] = host_ms();
    if (trace)
        fprintf(stderr, "[orbfe] rgbd frame: gray upload %.3f  enqueue %.3f  depth upload %.3f  rgbd+download+wait %.3f  hand-over %.3f ms\n",
                t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4]);
    return rc;
}

extern "C" int orbfe_rgbd_frame(orbfe_context *ctx, const uint8_t *gray, const float *depth_img,
                                int w, int h, size_t gray_stride, size_t depth_stride,
                                orbfe_keypoint *kps, uint8_t *desc, int *n,
                                float *u_right, float *depth, int cap)
try {
    ORBFE_ENTRY(ctx);
    return rgbd_frame_impl(ctx, gray, depth_img, sizeof(float), 1.0f, w, h, gray_stride, depth_stride, kps, desc, n, u_right, depth, cap);
} ORBFE_CATCH(ctx)

extern "C" int orbfe_rgbd_frame_u16(orbfe_context *ctx, const uint8_t *gray, const uint16_t *depth_img, float depth_map_factor,
                                    int w, int h, size_t gray_stride, size_t depth_stride,
                                    orbfe_keypoint *kps, uint8_t *desc, int *n,
                                    float *u_right, float *depth, int cap)
try {
    ORBFE_ENTRY(ctx);
    return rgbd_frame_impl(ctx, gray, depth_img, sizeof(uint16_t), depth_map_factor, w, h, gray_stride, depth_stride, kps, desc, n, u_right, depth, cap);
} ORBFE_CATCH(ctx)

// N RGB-D frames in one chain (BASELINE config 5 batched; the reference builds a multi-camera RGB-D runner, CMakeLists.txt:145-146):
// extraction of the N grey images, then Frame::ComputeStereoFromRGBD (src/Frame.cc:645-666) for every image slot in one launch.
extern "C" int orbfe_enqueue_rgbd(orbfe_context *ctx, const uint8_t *d_gray, const void *d_depth, int depth_is_u16, float depth_map_factor,
                                  int n_images, void *stream)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !d_gray || !d_depth) return fail(ctx, ORBFE_ERR_INVALID, "null argument");
    if (n_images < 1 || n_images > ctx->params.max_images) return fail(ctx, ORBFE_ERR_CAPACITY, "n_images %d outside [1, %d]", n_images, ctx->params.max_images);
    if (ctx->cfg.rm_on) return fail(ctx, ORBFE_ERR_UNSUPPORTED, "RGB-D frames with rectification maps are not supported (the depth map would need the same warp)");
    const int rc = enqueue_batch(ctx, d_gray, n_images, 1, stream);
    if (rc != ORBFE_OK) return rc;
    orbfe_launch_rgbd_batch(ctx->cfg, ctx->buf, d_depth, depth_is_u16 != 0, depth_is_u16 ? depth_map_factor : 1.0f, n_images, pick_stream(ctx, stream));
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->latest_foreign) HIP_TRY(ctx, hipEventRecord(ctx->ev_latest, pick_stream(ctx, stream))); // "the latest call" now ends after the depth kernel
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

extern "C" int orbfe_fetch_pyramid(orbfe_context *ctx, int image, int level, int blurred, uint8_t *dst, size_t dst_stride)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !dst || image < 0 || image >= ctx->params.max_images || level < 0 || level >= ctx->cfg.nlevels)
        return fail(ctx, ORBFE_ERR_INVALID, "bad argument");
    { const int rcw = wait_latest(ctx); if (rcw != ORBFE_OK) return rcw; }
    const LevelInfo &L = ctx->cfg.lv[level];
    if (dst_stride < (size_t)L.w) return fail(ctx, ORBFE_ERR_INVALID, "dst_stride smaller than level width");
    if (blurred) { // tiled on the device: download the level's tiles and lay the rows out
        const size_t bytes = (size_t)L.blur_tx * ((L.h + 3) / 4) * 128;
        std::vector<uint8_t> t(bytes);
        HIP_TRY(ctx, hipMemcpy(t.data(), ctx->buf.blur + (size_t)image * ctx->cfg.blur_bytes + L.blur_off, bytes, hipMemcpyDeviceToHost));
        for (int y = 0; y < L.h; y++)
            for (int x = 0; x < L.w; x += 4) { // 32 x 4 px tiles of eight 4 x 4 px blocks
                const size_t off = ((size_t)(y >> 2) * L.blur_tx + (x >> 5)) * 128 + (size_t)((x & 31) >> 2) * 16 + (size_t)(y & 3) * 4;
                memcpy(dst + (size_t)y * dst_stride + x, t.data() + off, (size_t)std::min(4, L.w - x));
            }
        return ORBFE_OK;
    }
    if (level == 0 && !blurred && ctx->last_src) { // read in place by the latest call: level 0 IS the caller's image
        // ... which the library does not own: a caller may have freed or reused it once the call's work was done (legal since ABI 1), so
        // the raw pointer is only followed when it is the library's own staging or the caller has promised to keep its images
        if (!ctx->last_src_owned && !ctx->input_retained)
            return fail(ctx, ORBFE_ERR_UNSUPPORTED, "level 0 of the latest call is the caller's own image buffer (read in place, never copied): "
                                                    "call orbfe_set_input_retained(ctx, 1) if that buffer is still valid, or read the images there");
        HIP_TRY(ctx, hipMemcpy2D(dst, dst_stride, ctx->last_src + (size_t)image * ctx->cfg.in_image_bytes, (size_t)L.w, (size_t)L.w, (size_t)L.h, hipMemcpyDeviceToHost));
        return ORBFE_OK;
    }
    const uint8_t *src = ctx->buf.pyr + (size_t)image * ctx->cfg.pyr_bytes + L.pyr_off;
    HIP_TRY(ctx, hipMemcpy2D(dst, dst_stride, src, (size_t)L.pitch, (size_t)L.w, (size_t)L.h, hipMemcpyDeviceToHost));
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

extern "C" int orbfe_fetch_candidates(orbfe_context *ctx, int image, int level, int32_t *xs, int32_t *ys,
                                      int32_t *scores, int cap, int *n)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !n || image < 0 || image >= ctx->params.max_images || level < 0 || level >= ctx->cfg.nlevels)
        return fail(ctx, ORBFE_ERR_INVALID, "bad argument");
    { const int rcw = wait_latest(ctx); if (rcw != ORBFE_OK) return rcw; }
    const DeviceConfig &c = ctx->cfg;
    const LevelInfo &L = c.lv[level];
    if (ctx->use_octree3) { // this path never materialises the emission-order arrays; build them for the tap
        orbfe_launch_candidates_gather(c, ctx->buf, ctx->params.max_images, ctx->stream);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    int nc = 0;
    HIP_TRY(ctx, hipMemcpy(&nc, ctx->buf.lvl_ncand + (size_t)image * c.nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
    *n = nc;
    if (nc > cap) return fail(ctx, ORBFE_ERR_CAPACITY, "caller buffers hold %d candidates, level has %d", cap, nc);
    if (nc == 0) return ORBFE_OK;
    std::vector<uint32_t> xy(nc);
    std::vector<uint8_t> sc(nc);
    HIP_TRY(ctx, hipMemcpy(xy.data(), ctx->buf.cand_xy + (size_t)image * c.cand_total + L.cand_off, sizeof(uint32_t) * nc, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(sc.data(), ctx->buf.cand_sc + (size_t)image * c.cand_total + L.cand_off, (size_t)nc, hipMemcpyDeviceToHost));
    for (int i = 0; i < nc; i++) {
        if (xs) xs[i] = (int32_t)(xy[i] & 0xffffu);
        if (ys) ys[i] = (int32_t)(xy[i] >> 16);
        if (scores) scores[i] = sc[i];
    }
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

extern "C" int orbfe_hamming_matrix(orbfe_context *ctx, const uint8_t *desc_a, int na, const uint8_t *desc_b, int nb, int32_t *dist)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || !desc_a || !desc_b || !dist || na < 0 || nb < 0) return fail(ctx, ORBFE_ERR_INVALID, "bad argument");
    if (na == 0 || nb == 0) return ORBFE_OK;
    const size_t need = (size_t)32 * na + (size_t)32 * nb + sizeof(int) * (size_t)na * nb;
    if (need > ctx->d_ham_bytes) {
        if (ctx->d_ham) (void)hipFree(ctx->d_ham); // only this entry point's own scratch
        ctx->d_ham = nullptr; ctx->d_ham_bytes = 0;
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_ham, need));
        ctx->d_ham_bytes = need;
    }
    uint8_t *da = ctx->d_ham, *db = da + (size_t)32 * na;
    int *dd = (int *)(db + (size_t)32 * nb);
    HIP_TRY(ctx, hipMemcpyAsync(da, desc_a, (size_t)32 * na, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(db, desc_b, (size_t)32 * nb, hipMemcpyHostToDevice, ctx->stream));
    orbfe_launch_hamming_matrix(da, na, db, nb, dd, ctx->stream);
    HIP_TRY(ctx, hipMemcpyAsync(dist, dd, sizeof(int) * (size_t)na * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ORBFE_OK;
} ORBFE_CATCH(ctx)
