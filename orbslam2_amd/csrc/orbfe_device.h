// orbfe_device.h -- device-side configuration shared by the kernels and the host API.
// gfx950 only (wave64).  See DESIGN.md for the HBM layout.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#define ORBFE_MAX_LEVELS 16
#define ORBFE_TAIL_MAX 3   // levels fused by pyr_tail_kernel
#define ORBFE_TAIL_COLS 64 // extended columns of the last level per workgroup
#ifndef ORBFE_BLUR_ROWS
#define ORBFE_BLUR_ROWS 32
#endif
#define ORBFE_WAVE 64

// Profiling cut points (tools/*_phases.sh, tools/*_insts.sh): an extra kernel argument that makes a kernel return after a
// given phase so that phases can be timed / counted.  They exist ONLY in the -DORBFE_PROFILE_CUTS build (make cuts ->
// tools/ab/cuts.so); the shipped liborbfe.so has neither the argument nor the branches nor the getenv.
#ifdef ORBFE_PROFILE_CUTS
#define ORBFE_CUT_PARAM , int dbg
// the marker leaves "; ORBFE_PHASE_END n" in the kernel's assembly at the cut: tools/isa_mix.py splits the disassembly into phases there
template <int N> __device__ __forceinline__ bool orbfe_phase_end() { asm volatile("; ORBFE_PHASE_END %0" ::"n"(N)); return true; }
#define ORBFE_CUT(n) (orbfe_phase_end<n>() && dbg == (n))
#define ORBFE_CUT_ARG(env) , orbfe_cut_value(env)
static inline int orbfe_cut_value(const char *env) { const char *v = getenv(env); return v ? atoi(v) : 0; }
#else
#define ORBFE_CUT_PARAM
#define ORBFE_CUT(n) false
#define ORBFE_CUT_ARG(env)
#endif

#ifndef ORBFE_PYR_RB
#define ORBFE_PYR_RB 4 // extended rows per wave of pyr_resize_direct_kernel (0: never use it; A/B builds only)
#endif
// Per-level geometry (reference: src/ORBextractor.cc:759-781,925-926,533-557).
struct LevelInfo {
    int w, h, pitch;       // level image size and row pitch (bytes)
    int pyr_off;           // byte offset of the level inside one image's pyramid buffer
    int n_cols, n_rows;    // FAST cell grid
    int w_cell, h_cell;
    int cell_off, n_cells; // cell index range inside one image's cell arrays
    int quota;             // mnFeaturesPerLevel[level]
    int n_ini;             // quadtree roots
    float hx;              // root width (float, as the reference)
    int cand_off, cand_cap;// candidate array range inside one image's candidate arrays
    int sel_off, sel_cap;  // selected-keypoint slot range inside one image's slot arrays
    int blur_off, blur_tx; // blurred level: byte offset inside one image's blurred pyramid and 32-px tiles per tile row (tiled layout)
    int blur_tile_off;     // first blur tile of this level in the fused all-level grid
    int blur_tiles_x, blur_tiles_y;
    int scaled_patch;      // int(patchSize * scale)
    float scale, inv_scale;
    double rs_scale_x, rs_scale_y; // cv::resize scale from level-1 (1/(dw/sw))
    int rs_xtab_off, rs_xtab_n;    // resize column table (two planes of rs_xtab_n words) in DeviceBuffers::rs_tab
    int rs_ytab_off, rs_ytab_n;    // resize row table
    int rs_src_rows[3];            // source rows spanned by the worst block of 16 / 8 / 4 output rows (pyr_resize_kernel)
    int rs_rw, rs_blk_off;         // rows per wave the launch uses (4 / 2 / 1) and this level's entries in DeviceBuffers::rs_blk
    int rs_direct, rs_dtab_off;    // pyr_resize_direct_kernel usable (every 4-column word's sources lie within 8 bytes) and its table: rs_xtab_n byte selectors, then rs_xtab_n / 4 first-source-byte offsets
    // pyr_pair_kernel (this level and the next in one launch): usable, tile grid over the NEXT level's extended domain (tiles of
    // pp_tw words x pp_tr rows), and this level's entries in DeviceBuffers::pair_plan (int4 units)
    int pp_ok, pp_ntx, pp_nty, pp_tw, pp_tr, pp_xoff, pp_yoff;
    int bk_xoff, bk_yoff;          // quadtree bucket tables of this level in DeviceBuffers::bk_tab (orbfe_octree3.hip)
    int bk_part_off, bk_part_n;    // this level's per-cell bucket partials in DeviceBuffers::bk_part / bk_emap
    int bk_depth;                  // quadtree bucket depth of this level: 5, or 4 where a FAST cell would span more than 64 depth-5 buckets
    int po_rb, po_cb;              // processing order: row / column bits of a node's bin (orbfe_octree3.hip step 4a)
    int bk_points;                 // some cell of the level spans > 64 buckets: the quadtree kernel buckets its candidates itself
};

struct DeviceConfig {
    int nlevels;
    int width, height;
    int edge_threshold, min_border; // min_border = edge_threshold - 3
    int ini_th, min_th;
    uint32_t ini_th_h2, min_th_h2; // the thresholds as half-precision numbers in both halves of a word (fast_cell_kernel's score clamp)
    int half_patch;
    int cell_cap;          // slots per FAST cell
    int cells_total;       // per image
    int cand_total;        // per image
    int sel_total;         // per image == keypoint capacity
    int blur_tiles_total;
    int proc_order;        // octree3_kernel also writes proc_xy / proc_meta and describe_kernel walks those (0: describe_kernel walks sel_xy; ORBFE_NO_PROC_ORDER=1, other quadtree kernels)
    int fast_blur_t0;      // blur tiles [fast_blur_t0, blur_tiles_total) ride in the FAST launch (set per launch; blur_tiles_total: none)
    uint32_t xcd_magic;    // ceil(2^32 / workgroups per XCD and round of units) of the block map of the launch this copy is passed to (FAST, describe,
                           // stereo match: set by the launcher, xcd_map_magic_host; 0: divide in the kernel)
    int max_nodes;         // quadtree node capacity (LDS)
    int bk_part_total;     // per image: entries of DeviceBuffers::bk_part
    int row_cap;           // entries per image row in DeviceBuffers::row_ent
    int patch_n;           // entries in DeviceBuffers::patch_uv (multiple of 64)
    // fused pyramid tail (pyr_tail_kernel): the last tail_n levels (2 or 3) in one launch, 0 = not used
    int tail_first, tail_n, tail_strips;
    int pp_max_images;                 // pyr_pair_kernel (two levels per launch) serves batches of up to this many images
    int tail_max_images;               // the fused tail serves batches of up to this many images; larger ones run levels tail_first.. as single launches
    int tail_src_words;                // staged words per row of level tail_first - 1 (widest strip)
    int tail_words[ORBFE_TAIL_MAX];    // words per row of stage s computed by the widest strip
    int tail_lds_y[ORBFE_TAIL_MAX];    // LDS byte offsets: row tables of stage s, ...
    int tail_lds_buf[ORBFE_TAIL_MAX];  // ... and the columns of stage s kept for stage s + 1
    int tail_lds_src, tail_lds_bytes;
    int umax[64];
    int taps[7];           // Gaussian 8.8 fixed-point taps
    size_t pyr_bytes;      // per image
    size_t blur_bytes;     // per image: blurred pyramid, 32 x 4 px tiles of 128 B, each eight 4 x 4 px blocks (see orbfe_pyramid.hip)
    float bf, fx, mb;
    // input pixel format (orbfe_set_input_format): 1 = CV_8UC1; 3 / 4 = interleaved colour converted by ingest with
    // cv::cvtColor's fixed-point weights for channels 0, 1, 2 (in_coef) and in_shift fraction bits
    int in_cn, in_coef[3], in_shift;
    size_t in_image_bytes; // bytes of one packed input image (w * h * in_cn, or rm_sw * rm_sh with rectification)
    // rectification (orbfe_set_rectification): cv::remap's fixed-point form of the float maps, one pair per side
    // (index = image slot & 1 when the right map is set, else 0): rm_xy = sx | sy << 16 (int16 each), rm_a = fy * 32 + fx
    int rm_on, rm_sw, rm_sh;
    const uint32_t *rm_xy[2];
    const uint16_t *rm_a[2];
    // lens distortion of Frame::UndistortKeyPoints (orbfe_set_distortion): k1 k2 p1 p2 k3; n_dist == 0 or dist[0] == 0: none
    int n_dist;
    float dist[5];
    float cam[4];          // fx fy cx cy
    LevelInfo lv[ORBFE_MAX_LEVELS];
};

// Per-batch device buffers (all [image][...] with the per-image strides of DeviceConfig).
struct DeviceBuffers {
    uint8_t *pyr;        // raw pyramid
    // Level 0 as the kernels read it (level_image(), orbfe_common.hpp).  Copy mode: pyr + lv[0].pyr_off, per-image stride pyr_bytes,
    // pitch lv[0].pitch, reflect-101 margin materialised by ingest.  In-place mode (lv0_packed = 1, round 4): the caller's packed
    // CV_8UC1 images themselves -- no ingest launch, no copy; rows at any alignment, no margin (the blur of level 0 reflects by
    // itself, every other reader stays inside the image)
    const uint8_t *lv0;
    size_t lv0_stride;
    int lv0_pitch, lv0_packed;
    uint8_t *blur;       // blurred pyramid
    int *cell_cnt;       // [img][cells_total]
    uint32_t *cell_xy;   // [img][cells_total*cell_cap]  x | y<<16 (region relative)
    uint8_t *cell_sc;    // [img][cells_total*cell_cap]
    int *cell_base;      // [img][cells_total] scratch (exclusive scan)
    uint32_t *cand_xy;   // [img][cand_total]
    uint8_t *cand_sc;    // [img][cand_total]
    uint32_t *ot_xy2;    // [img][cand_total] quadtree ping-pong partner
    uint32_t *idx0, *idx1; // [img][cand_total] ping-pong permutation
    uint32_t *bk_part;   // [img][bk_part_total] count | best slot << 12 | best score << 24 (ORBFE_BK_PART) of every bucket a FAST cell's
                         // survivors can fall into, cell after cell; rewritten by fast_cell_kernel every frame, summed into the
                         // bucket arrays (LDS) by octree3_kernel
    int *bk_end;         // [img][nlevels][4097] quadtree deep path: bucket ends of the counting sort
    uint8_t *ot3_scratch; // [img][nlevels][node_bytes] quadtree node tables when they do not fit LDS (else null)
    uint32_t *bk_best;   // [img][nlevels][ORBFE_BK_PYR] best-key pyramid of every (image, level): written after the pyramid step, read by the final selection
    int *lvl_ncand;      // [img][nlevels]
    int *sel_cnt;        // [img][nlevels]
    uint32_t *sel_xy;    // [img][sel_total]
    uint8_t *sel_sc;     // [img][sel_total]
    uint32_t *proc_xy;   // [img][sel_total] describe_kernel's processing order (DeviceConfig::proc_order): the keypoints of a level in a spatial order, ...
    uint32_t *proc_meta; // ... and their slot | score << 24 (octree3_kernel writes both beside sel_xy / sel_sc, which stay in the reference's order)
    void *kps;           // [img][sel_total] orbfe_keypoint
    uint8_t *desc;       // [img][sel_total][32]
    int *kp_cnt;         // [img]
    float *u_right;      // [img][sel_total]
    float *depth;        // [img][sel_total]
    int *sad;            // [img][sel_total] best SAD (or -1)
    int *status;         // [img] non-zero = device-side capacity problem
    int *row_cnt;        // [pair][height] stereo row lists: right keypoints whose band covers the row (every row written by the row-list waves, orbfe_rowlist.hpp)
    uint2 *row_ent;      // [pair][height][row_cap] entries: (iR | octave << 16, x bits), written by stereo_rowlist_kernel
    const uint32_t *bk_tab; // quadtree bucket tables: per level X[region_w] then Y[region_h] (see ORBFE_BK_*)
    const uint32_t *bk_emap; // [bk_part_total] bucket index | level-local cell << 16 of every bk_part entry
    const uint32_t *bk_off;  // [cells_total] first bk_part entry of the cell; ~0u: the cell spans > 64 buckets (no partials)
    const uint32_t *rs_tab; // cv::resize offset/weight tables of every level (see pyr_resize_kernel)
    const uint4 *cell_info; // [cells_total] FAST cells: level | valid << 8, ini_x | ini_y << 16, tile w | h << 8, index inside the level (fast_cell_kernel)
    const uint2 *cell_aux;  // [cells_total] the lane maps of the cell's shape, divided on the host: x = shape index (fast_lane_tab) | (64 / ng) << 17 | last band's first row << 24
                            // (ng = 4-pixel groups per interior row), y = ceil(2^16 / cpr) | (64 / cpr) << 17 (cpr = 16-byte chunks per tile row)
    const uint32_t *fast_lane_tab; // [shapes][64 lanes][8]: phase A's per-lane constants of a cell shape (tile w x h): pixel masks of the pairs 0 1 / 2 3, the same in
                                   // the cell's last band, entry word of pixels 0 1, word offset of the lane's window in the LDS tile, 2 spare
    const uint32_t *blur_tile_info; // [blur_tiles_total] level | column strip << 8 | first row << 16 (blur_kernel)
    const uint32_t *rs_blk; // per (level, block of 4 * rs_rw output rows): first source row | source rows << 16 (pyr_resize_kernel)
    const int *pair_plan;   // pyr_pair_kernel: per tile column / row of a level pair {first word (row) of the LDS tile, words (rows), first, end word (row) this workgroup stores}
    const int *tail_plan;   // [tail_strips][ORBFE_TAIL_MAX][4]: first extended column, words, first staged source column, staged words (pyr_tail_kernel)
    long long *dbg_ts;   // 4096 timestamps for kernel bring-up (ORBFE_OT2_STOP=99); never read by product code
    const uint8_t *slot_level; // [sel_total] level of every keypoint slot
    const int16_t *patch_uv; // IC_Angle patch offsets: (u & 0xff) | (v << 8), padded with (0,0)
    const uint32_t *mom_tab; // [64 lanes][12] byte-dot-product weights of the same patch (hp == 15), see orbfe_api.hip
    const uint32_t *pattern; // [256] the extractor's copy of the rBRIEF tests (src/ORBextractor.cc:442-444): x0 | y0 << 8 | x1 << 16 | y1 << 24 as int8;
                             // the compiled bit_pattern_31_ unless orbfe_set_pattern replaced it
};

// Quadtree buckets (orbfe_octree3.hip).  A candidate's bucket = its root and quadrant path down to depth 5; the
// path is separable (x decides the x bits, y the y bits), so it is X[x] | Y[y] from two host-built tables:
//   X[x] = root << 10 | x bits spread to the even positions | (root * 32 + column) << 16
//   Y[y] = y bits spread to the odd positions | row << 16
// Best key of a bucket: score (8 bits) << 24 | ~(level-local cell (12 bits) << 12 | slot (12 bits)) -- maximum
// = best score, first in cv::FAST emission order.
#define ORBFE_BK_DEPTH 5
#define ORBFE_BK_BUCKETS 4096
#define ORBFE_BK_PYR 5460 // entries of one bucket pyramid: 4 roots x (1 + 4 + ... + 4^5)
#define ORBFE_BK_REF_MASK 0xffffffu
#define ORBFE_BK_KEY(sc, cell, slot) (((sc) << 24) | (ORBFE_BK_REF_MASK - (unsigned)(((cell) << 12) | (slot))))
// A cell's partial entry: the cell is implied by the entry's position, so count (<= cell_cap <= 1024) and the key's score and
// (inverted) slot fields fit one word; ORBFE_BK_PART_KEY rebuilds the key for level-local cell `cell`.
#define ORBFE_BK_PART(cnt, key) ((cnt) | (((key) & 0xfffu) << 12) | ((key) & 0xff000000u))
#define ORBFE_BK_PART_KEY(e, cell) (((e) & 0xff000000u) | ((4095u - (unsigned)(cell)) << 12) | (((e) >> 12) & 0xfffu))

struct KeyPointPOD {
    float x, y, size, angle, response;
    int32_t octave, class_id;
};
static_assert(sizeof(KeyPointPOD) == 28, "keypoint must match cv::KeyPoint");

// launchers (orbfe_pyramid.hip, orbfe_fast.hip, orbfe_octree*.hip, orbfe_describe.hip, orbfe_stereo.hip)
void orbfe_launch_ingest(const DeviceConfig &cfg, const DeviceBuffers &buf, const uint8_t *d_images,
                         int n_images, hipStream_t s);
int orbfe_launch_pyramid(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, bool fuse_blur, hipStream_t s, int ride_from = ORBFE_MAX_LEVELS); // returns the number of levels (from 0) whose blur it launched too
int orbfe_resize_word_base_host(int xw, int dst_w, double scale, int src_w); // pyr_resize_direct_kernel's first-source-byte formula, for orbfe_create's check
void orbfe_launch_blur(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, int first_level, hipStream_t s);
void orbfe_launch_fast(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, bool buckets, hipStream_t s, int blur_first_level);
int orbfe_fast_tile_pitch(const DeviceConfig &cfg); // bytes per row of fast_cell_kernel's LDS tile (orbfe_create builds fast_lane_tab with it)
void orbfe_launch_octree_generic(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, hipStream_t s);
// orbfe_octree3.hip
void orbfe_launch_octree3(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, int sort_cap, size_t lds, bool nodes_in_hbm, hipStream_t s);
void orbfe_launch_candidates_gather(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, hipStream_t s);
size_t orbfe_octree3_node_bytes(int max_nodes, int sort_cap);
size_t orbfe_octree3_lds_bytes(int max_nodes, int sort_cap, bool nodes_in_hbm);
int orbfe_octree3_prepare(size_t lds, bool nodes_in_hbm);
void orbfe_launch_describe(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, bool stereo, hipStream_t s);
void orbfe_launch_stereo_rowlists(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_pairs, hipStream_t s);
void orbfe_launch_stereo_match(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_pairs, hipStream_t s);
void orbfe_launch_stereo_median(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_pairs, hipStream_t s);
void orbfe_launch_rgbd(const DeviceConfig &cfg, const DeviceBuffers &buf, const float *d_depth,
                       size_t depth_pitch_floats, int image, hipStream_t s);
void orbfe_launch_undistort(const DeviceConfig &cfg, const void *d_keys_in, void *d_keys_out, int n, hipStream_t s);
void orbfe_launch_rgbd_u16(const DeviceConfig &cfg, const DeviceBuffers &buf, const uint16_t *d_depth, size_t depth_pitch_px,
                           float factor, int image, hipStream_t s);
void orbfe_launch_rgbd_batch(const DeviceConfig &cfg, const DeviceBuffers &buf, const void *d_depth, bool is_u16, float factor, int n_images, hipStream_t s);
// byte offsets of the arrays inside a packed result block (orbfe_packed_layout of include/orbfe.h mirrors it)
struct PackedOffsets { size_t counts, level_counts, xy, angle, response, desc, u_right, depth; };
void orbfe_launch_pack_results(const DeviceConfig &cfg, const DeviceBuffers &buf, uint8_t *d_out, const PackedOffsets &lay, int n_out, int img_step, bool stereo, hipStream_t s);
void orbfe_launch_hamming_matrix(const uint8_t *da, int na, const uint8_t *db, int nb, int *dist, hipStream_t s);
