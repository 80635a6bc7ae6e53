/*
 * orbfe.h -- C ABI of the MI355X (gfx950) ORB front-end.
 *
 * Drop-in boundary for the per-frame front-end of fabrizioromanelli/ORBSLAM2.
 * Every entry point names the reference interface it replaces (file:line under
 * the reference tree).  Plain pointers and sizes only; no C++/torch types; no
 * exceptions cross this boundary.  All functions return ORBFE_OK (0) or a
 * negative error code; orbfe_last_error() gives a message.
 *
 * Threading: different contexts may be used from different threads concurrently
 * (the reference runs its two extractor objects on two threads, src/Frame.cc:78-81).
 * Calls on the SAME context are serialised by a mutex inside the context, so the
 * reference's Tracking / LocalMapping / LoopClosing threads may share one (their
 * ORBmatcher objects do, through the compat shim); "the latest extraction call" that
 * the fetch functions and device_slot_plus1 refer to is then whichever call the
 * context saw last -- callers that interleave enqueue and fetch from several threads
 * on one context must order those pairs themselves.
 *
 * Environment (read once, by orbfe_create; meant for tests and A/B measurements):
 *   ORBFE_OCTREE=1     force the generic node-parallel DistributeOctTree kernel (the fallback beyond the
 *                      bucket-pyramid kernel's limits) instead of the bucket-pyramid one
 *                      (orbfe_quadtree_kernel() reports the choice);
 *   ORBFE_NO_TAIL=1|0  never / always run the last three pyramid levels in the fused tail
 *                      kernel (default: for batches of fewer than 64 images; larger
 *                      batches run them as single launches);
 *   ORBFE_PYR_LDS=1    keep cv::resize on the LDS-staged kernel (the path of scale factors
 *                      above ~2) instead of the direct one;
 *   ORBFE_NO_FUSE=1    blur every level in one launch after the pyramid instead of blurring
 *                      levels inside the pyramid's and FAST's launches;
 *   ORBFE_BLUR_RIDE_FROM=l for every batch size, blur levels >= l in FAST's launch and the lower ones beside the resize
 *                      that reads them (default: every level rides for batches of 64 images and more -- the blur is
 *                      memory-bound, FAST issue-bound -- and smaller batches blur beside the resize launches, FAST's
 *                      launch taking what they leave);
 *   ORBFE_NO_PROC_ORDER=1 describe_kernel walks the keypoints in slot order instead of the
 *                      spatial order the quadtree kernel writes beside its selection
 *                      (results are the same either way: only the order of processing differs);
 *   ORBFE_NO_INPLACE=1 copy packed grey input into the library's pitched level 0 (ingest16_kernel)
 *                      instead of reading the caller's images in place as pyramid level 0;
 *   ORBFE_NO_PAIR=1|0  never / always compute two pyramid levels per launch (pyr_pair_kernel;
 *                      default: for batches of fewer than 64 images);
 *   ORBFE_HOST_TRACE=1 print the context's geometry, kernel choices and LDS sizes to
 *                      stderr at create time.
 * Every alternative plan gives bit-identical results (tools/r05_fullsuite.sh runs the
 * frame-path tests under each).  Round 5 removed the plans that lost at every measured batch size (the blur in the quadtree
 * launch, depth-5 buckets everywhere, the resize table / formula switch) and the point-parallel quadtree kernel (the generic
 * kernel covers its geometries).
 *
 * No C++ exception leaves the library: a host-side failure (out of memory, ...) is
 * returned as ORBFE_ERR_HIP with its message in orbfe_last_error().
 *
 * There is NO CPU fallback: if no HIP device is present orbfe_create fails
 * with ORBFE_ERR_NO_DEVICE.
 */
#ifndef ORBFE_H
#define ORBFE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBFE_ABI_VERSION 6 /* 2: orbfe_frame_view.device_slot_plus1; 3: .keyframe; 4: orbfe_get_camera, orbfe_assign_features_to_grid, orbfe_stereo_batch, orbfe_device_count, orbfe_set_profiling_interval; 5: orbfe_get_packed_layout, orbfe_fetch_batch_packed, orbfe_expand_packed, orbfe_stereo_batch_packed, orbfe_enqueue_rgbd; 6: orbfe_build_id, orbfe_set_pattern, orbfe_get_pattern, orbfe_blur_ride_from, orbfe_set_input_retained (additive: no struct changed; orbfe_fetch_pyramid(level 0) of an in-place batched call now needs the latter) */

enum {
    ORBFE_OK = 0,
    ORBFE_ERR_INVALID = -1,    /* bad argument */
    ORBFE_ERR_NO_DEVICE = -2,  /* no HIP device / HIP runtime failure at create */
    ORBFE_ERR_HIP = -3,        /* HIP runtime error during a call */
    ORBFE_ERR_CAPACITY = -4,   /* caller buffer or context capacity too small */
    ORBFE_ERR_UNSUPPORTED = -5 /* image type / size outside what the context was built for */
};

/* Layout-identical to cv::KeyPoint (28 bytes) so compat shims can memcpy. */
typedef struct orbfe_keypoint {
    float x, y;      /* pt, level-0 pixel coordinates */
    float size;      /* int(patchSize * scale[octave]) */
    float angle;     /* degrees, [0,360) */
    float response;  /* FAST score */
    int32_t octave;
    int32_t class_id; /* -1 */
} orbfe_keypoint;

/* The 8 ORBextractor constructor arguments (include/ORBextractor.h:51,
 * src/ORBextractor.cc:405) + the camera numbers Frame needs (src/Frame.cc:104-114)
 * + sizing of the device context. */
typedef struct orbfe_params {
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
    int32_t patch_size;
    int32_t half_patch_size;
    int32_t edge_threshold; /* >= 19 (the descriptor pattern's reach + 1) and >= half_patch_size + 4 */
    float fx, fy, cx, cy;
    float bf;             /* baseline * fx (Frame::mbf) */
    int32_t device;       /* HIP device ordinal */
    int32_t width, height;/* image size the context is built for (one camera model per context) */
    int32_t max_images;   /* images in flight per batched call (2 per stereo pair) */
} orbfe_params;

typedef struct orbfe_context orbfe_context;

int orbfe_abi_version(void);
/* sha256 (64 hex digits) over the library's sources and compile flags, fixed at build time (orbslam2_amd/csrc/Makefile): measurement
 * files under profiles/ carry the id of the build they were taken on, and bench.py replays a counter only when it matches the
 * library it ran.  No counterpart in the reference. */
const char *orbfe_build_id(void);
const char *orbfe_last_error(const orbfe_context *ctx);

/* Replaces `new ORBextractor(...)` (src/Tracking.cc:125-131, src/ORBextractor.cc:405-464). */
int orbfe_create(const orbfe_params *params, orbfe_context **out);
void orbfe_destroy(orbfe_context *ctx);

/* The extractor's copy of the rBRIEF test table: ORBextractor::ORBextractor copies the 512 points of bit_pattern_31_ into its
 * member `pattern` (src/ORBextractor.cc:442-444, include/ORBextractor.h:93), computeOrbDescriptor reads it (:103-142).  A new
 * context holds the compiled-in table (orbslam2_amd/csrc/orb_pattern_31.inc); orbfe_set_pattern replaces it for every call
 * enqueued afterwards -- what a multi-GPU deployment does with the table rank 0 broadcasts (orbslam2_amd/dist.py:
 * broadcast_pattern).  pattern = 256 tests x (x0, y0, x1, y1), i.e. the reference's `int bit_pattern_31_[256 * 4]` layout; a
 * point with x^2 + y^2 > 342 (it could rotate to more than 18 px from the keypoint) is refused with ORBFE_ERR_UNSUPPORTED: the
 * descriptor stage reads +-18 px, which edge_threshold >= 19 keeps inside the level; the reference's table reaches x^2 + y^2 = 338. */
/* Lifetime promise for the images of orbfe_enqueue_* (see orbfe_fetch_pyramid): retained != 0 = they stay valid and unchanged until
 * this context's next enqueue call.  Default 0: valid until the call's work on the stream has finished (stream order suffices). */
int orbfe_set_input_retained(orbfe_context *ctx, int retained);
/* Launch-plan query for measurement tools: the first pyramid level whose Gaussian blur (src/ORBextractor.cc:899-900) is computed by
 * workgroups riding in the cell-FAST launch for a batch of n_images images (orbfe_levels(): none).  bench.py prices the
 * dominant kernel's launch with it. */
int orbfe_blur_ride_from(const orbfe_context *ctx, int n_images);
int orbfe_set_pattern(orbfe_context *ctx, const int32_t *pattern);
int orbfe_get_pattern(const orbfe_context *ctx, int32_t *pattern);

/* Getters of include/ORBextractor.h:61-82 (GetLevels/GetScaleFactors/...), plus
 * mnFeaturesPerLevel and umax for tests.  Arrays hold nlevels entries (umax: half_patch+1). */
int orbfe_levels(const orbfe_context *ctx);
/* cam[5] = fx, fy, cx, cy, bf the context was created with: every matcher / optimiser entry point projects with these
 * (the reference reads Frame::fx ... / pKF->fx ..., one camera model per process, src/Frame.cc:29-33); a shim that is handed
 * a KeyFrame checks them against pKF->fx ... instead of trusting that the right context was picked. */
int orbfe_get_camera(const orbfe_context *ctx, float *cam);
int orbfe_keypoint_capacity(const orbfe_context *ctx); /* max keypoints one image can yield */
int orbfe_get_tables(const orbfe_context *ctx, float *scale, float *inv_scale, float *sigma2,
                     float *inv_sigma2, int32_t *features_per_level, int32_t *umax);
int orbfe_level_size(const orbfe_context *ctx, int level, int *w, int *h);

/* ORBextractor::operator() (src/ORBextractor.cc:858-919): host image in, host
 * keypoints (level-major, quadtree-leaf order) + 32-byte descriptors out.
 * `cap` entries are available in kps/desc; *n receives the count.  An empty image
 * (img==NULL or w/h<=0) returns ORBFE_OK with *n = 0 and outputs untouched. */
int orbfe_extract(orbfe_context *ctx, const uint8_t *img, int w, int h, size_t stride,
                  orbfe_keypoint *kps, uint8_t *desc, int cap, int *n);

/* Frame::Frame(stereo) body (src/Frame.cc:61-117): ExtractORB(left) + ExtractORB(right)
 * + ComputeStereoMatches (src/Frame.cc:464-642).  u_right/depth have cap entries and
 * are filled for the first *n_left (-1 where unmatched).  mb := bf/fx. */
int orbfe_stereo_frame(orbfe_context *ctx, const uint8_t *left, const uint8_t *right,
                       int w, int h, size_t stride,
                       orbfe_keypoint *kps_left, uint8_t *desc_left, int *n_left,
                       orbfe_keypoint *kps_right, uint8_t *desc_right, int *n_right,
                       float *u_right, float *depth, int cap);

/* The same for n_pairs stereo pairs in one call (host memory in and out; pinned memory makes the copies asynchronous to other
 * contexts' work): images = [2 * n_pairs][h][w * channels] packed (L0, R0, L1, R1, ...), one upload, one stage chain, one
 * download.  Outputs are laid out like the device arrays, [image][orbfe_keypoint_capacity()] records (kps 28 B, desc 32 B,
 * u_right / depth float: left-image slots) and counts[2 * n_pairs]; kps / desc / u_right / depth may be NULL.  This is the
 * per-context call of a single-process multi-device host (orbslam2_amd/host/multi_device.h: N contexts, one feeder thread each). */
int orbfe_stereo_batch(orbfe_context *ctx, const uint8_t *images, int n_pairs, orbfe_keypoint *kps, uint8_t *desc, int32_t *counts,
                       float *u_right, float *depth);
/* HIP devices visible to the process (orbfe_params.device ranges over them); 0 without a device. */
int orbfe_device_count(void);

/* Input pixel format of every image entry point of this context (host and device-resident), default CV_8UC1.
 * channels = 3 / 4 makes ingest perform the grey conversion Tracking::GrabImageMonocular / Stereo / RGBD do before they
 * build the Frame (src/Tracking.cc:269-294,305-321,335-351): cv::cvtColor(im, im, COLOR_RGB2GRAY / BGR2GRAY / RGBA2GRAY /
 * BGRA2GRAY); rgb_order = Tracking's mbRGB (channel 0 is red).  Images are then w * channels bytes per packed row
 * (stride arguments count bytes of such rows; orbfe_enqueue_* read [image][h][w][channels]).  legacy_weights != 0 selects
 * OpenCV 3.x's 14-bit weights (4899, 9617, 1868) instead of 4.x's 15-bit ones (9798, 19235, 3735). */
int orbfe_set_input_format(orbfe_context *ctx, int channels, int rgb_order, int legacy_weights);

/* Lens distortion of the context's camera: Tracking's mDistCoef = k1 k2 p1 p2 [k3] (src/Tracking.cc:67-78), n = 0 / 4 / 5.
 * With k1 != 0 the RGB-D entry points build mvuRight from the UNDISTORTED keypoint x as Frame::ComputeStereoFromRGBD does
 * (src/Frame.cc:652-664), and the three functions below reproduce Frame::UndistortKeyPoints / ComputeImageBounds
 * (src/Frame.cc:402-462): cv::undistortPoints(pts, K, D, Mat(), K) -- five fixed-point iterations in double.
 * k1 == 0 means "no distortion" exactly as the reference tests it (:404,436). */
int orbfe_set_distortion(orbfe_context *ctx, const float *dist, int n);
/* mvKeysUn from mvKeys: copies every field, replaces pt (host arrays). */
int orbfe_undistort_keypoints(orbfe_context *ctx, const orbfe_keypoint *kps, int n, orbfe_keypoint *kps_un);
/* mvKeysUn of image slot `image` of the latest call, undistorted on the device before the download. */
int orbfe_fetch_keys_un(orbfe_context *ctx, int image, orbfe_keypoint *kps_un, int cap, int *n);
/* bounds[4] = mnMinX, mnMaxX, mnMinY, mnMaxY for the context's image size. */
int orbfe_image_bounds(orbfe_context *ctx, float *bounds);

/* Stereo rectification in front of the pipeline: level 0 = cv::remap(raw, map1, map2, INTER_LINEAR) as the EuRoC
 * drivers do before TrackStereo (Test/Replay/Stereo/stereo_euroc.cc:98-99,136-137).  map_x / map_y are the CV_32FC1 maps
 * of cv::initUndistortRectifyMap (width x height of the context, row major); src_w x src_h is the raw image size the
 * entry points then expect (host and device-resident: [image][src_h][src_w] bytes).  side 0 = left / monocular, side 1 =
 * right (slot parity of orbfe_*_stereo); without a right map every image uses the left one.  NULL maps clear a side
 * (side 0: rectification off).  Single-channel input only. */
int orbfe_set_rectification(orbfe_context *ctx, int side, const float *map_x, const float *map_y, int src_w, int src_h);

/* Frame::Frame(rgbd) body (src/Frame.cc:120-172) for an undistorted camera:
 * ExtractORB + ComputeStereoFromRGBD (src/Frame.cc:645-666).  depth_img is CV_32F
 * metres, row stride in bytes.  uRight uses the undistorted x when orbfe_set_distortion gave k1 != 0. */
int orbfe_rgbd_frame(orbfe_context *ctx, const uint8_t *gray, const float *depth_img,
                     int w, int h, size_t gray_stride, size_t depth_stride,
                     orbfe_keypoint *kps, uint8_t *desc, int *n,
                     float *u_right, float *depth, int cap);

/* The same with the sensor's raw CV_16U depth map: folds the conversion of Tracking::GrabImageRGBD
 * (src/Tracking.cc:323-324, imDepth.convertTo(imDepth, CV_32F, mDepthMapFactor)) into the sampling, so
 * half the bytes cross the bus and no full-image conversion runs.  depth_map_factor is Tracking's
 * already inverted mDepthMapFactor (src/Tracking.cc:151-155: 1.0f / DepthMapFactor, or 1 when unset);
 * depth_stride in bytes. */
int orbfe_rgbd_frame_u16(orbfe_context *ctx, const uint8_t *gray, const uint16_t *depth_img, float depth_map_factor,
                         int w, int h, size_t gray_stride, size_t depth_stride,
                         orbfe_keypoint *kps, uint8_t *desc, int *n,
                         float *u_right, float *depth, int cap);

/* mvImagePyramid[level] of image slot `image` of the latest call (include/ORBextractor.h:84;
 * read by src/Frame.cc:471,565,577,582).  blurred!=0 returns the Gaussian-blurred
 * working copy (src/ORBextractor.cc:899-900).  Copies w*h bytes into dst (row stride dst_stride).
 * Level 0 (unblurred) of a device-resident call on packed grey images IS the caller's buffer (read in place, never copied): the
 * library follows that pointer only after orbfe_set_input_retained(ctx, 1) and returns ORBFE_ERR_UNSUPPORTED otherwise -- the
 * caller may legally have freed or reused the buffer once the call's work was done.  The host entry points (orbfe_extract,
 * orbfe_stereo_frame, ...) stage their images inside the library and are not affected. */
int orbfe_fetch_pyramid(orbfe_context *ctx, int image, int level, int blurred,
                        uint8_t *dst, size_t dst_stride);

/* ---- batched / device-resident path (no reference counterpart: SURVEY.md §8e) ----
 * d_images: device pointer to n_images contiguous 8UC1 images (w*h bytes each, row
 * stride w; stereo pairs as L0,R0,L1,R1,...).  Work is enqueued on `stream`
 * (a hipStream_t; NULL = the context's own stream) and NOT synchronised.  Results
 * stay in context-owned device buffers until fetched.  Round 4: packed 8UC1 input is read IN PLACE as pyramid level 0 by every
 * stage of the call (no copy into the library's pyramid), so the images must stay valid and unchanged until the call's work on
 * `stream` has finished (stream order is enough: a copy into the same buffer queued on the same stream is fine).  ORBFE_NO_INPLACE=1
 * at orbfe_create keeps the round-3 copy (ingest16_kernel); colour / rectified input always goes through ingest. */
int orbfe_enqueue_extract(orbfe_context *ctx, const uint8_t *d_images, int n_images, void *stream);
int orbfe_enqueue_stereo(orbfe_context *ctx, const uint8_t *d_images, int n_pairs, void *stream);
int orbfe_synchronize(orbfe_context *ctx, void *stream);
/* Cut every batched call into `groups` (1..8) contiguous sub-batches whose stage chains run on internal
 * streams forked from / joined to the caller's stream (overlaps barrier-bound and ALU-bound stages). */
int orbfe_set_streams(orbfe_context *ctx, int groups);
/* Which DistributeOctTree kernel this context uses (src/ORBextractor.cc:533-757): 3 = bucket pyramid,
 * 1 = generic node-parallel (chosen at create time from the geometry / LDS limits; 2 was the point-parallel
 * kernel, removed in round 5). */
int orbfe_quadtree_kernel(const orbfe_context *ctx);
/* Copy the results of image slot `image` to host.  u_right/depth may be NULL.  The blocking fetch functions
 * (orbfe_fetch_image / _counts / _keys_un / _pyramid / _candidates) first wait for the stream of the latest
 * orbfe_enqueue_* call, so no orbfe_synchronize is needed in between; orbfe_fetch_batch_async does not wait. */
int orbfe_fetch_image(orbfe_context *ctx, int image, orbfe_keypoint *kps, uint8_t *desc,
                      float *u_right, float *depth, int cap, int *n);
/* Per-image keypoint counts of the latest batch (n_images ints). */
int orbfe_fetch_counts(orbfe_context *ctx, int32_t *counts, int n_images);
/* Results of the latest batched call for image slots 0 .. n_images-1, copied asynchronously on `stream` (NULL: the context's)
 * into caller buffers laid out like the device arrays: [image][orbfe_keypoint_capacity()] records (kps 28 B, desc 32 B,
 * u_right / depth float; left-image slots only carry the last two) and counts[n_images].  Pinned host memory makes the
 * copies overlap other streams' work; any pointer may be NULL.  Synchronise the stream before reading. */
int orbfe_fetch_batch_async(orbfe_context *ctx, int n_images, orbfe_keypoint *kps, uint8_t *desc, int32_t *counts,
                            float *u_right, float *depth, void *stream);

/* ---- packed results (round 4): the same batch in ONE device-to-host copy of about two thirds the bytes ----
 * A cv::KeyPoint's pt, size, octave and class_id are functions of (x, y on its level, octave): size = scaledPatchSize
 * (src/ORBextractor.cc:838), pt *= mvScaleFactor[level] (:909-915), and the octave follows from the per-level counts because a
 * frame's keypoints are stored octave by octave (:866-917).  The block carries per keypoint x | y << 16 on its level (4 B), the
 * angle (4 B) and the FAST score (1 B) -- 9 bytes instead of 28 -- plus per image the count and the per-level counts, the 32-byte
 * descriptors, and with ORBFE_PACK_STEREO uRight / depth of the LEFT images only (orbfe_fetch_batch_async also moves the unused
 * right-image slots).  ORBFE_PACK_LEFT_ONLY drops the right images' keypoints and descriptors too (nothing outside
 * ComputeStereoMatches, src/Frame.cc:464-642, reads mvKeysRight / mDescriptorsRight).  Arrays are [out image][capacity] at the byte
 * offsets of orbfe_packed_layout (64-byte aligned); out image o is image slot o, or slot 2 o with LEFT_ONLY. */
enum { ORBFE_PACK_STEREO = 1, ORBFE_PACK_LEFT_ONLY = 2,
       /* host_block is pinned host memory that the device addresses at the same pointer (hipHostMalloc / hipHostRegister; checked):
        * the gather kernel stores the block across the link itself and NO copy is queued -- a kernel's stores run beside an upload,
        * which two copy-engine transfers in opposite directions do not on the measured link (profiles/r04_pcie.json) */
       ORBFE_PACK_DIRECT = 4 };
typedef struct orbfe_packed_layout {
    int32_t n_images_out, capacity, nlevels, n_pairs, flags, reserved;
    size_t counts_off;       /* int32 [n_images_out] */
    size_t level_counts_off; /* int32 [n_images_out][nlevels]: keypoints per octave */
    size_t xy_off;           /* uint32 [n_images_out][capacity]: x | y << 16, level-image pixels */
    size_t angle_off;        /* float [n_images_out][capacity] */
    size_t response_off;     /* uint8 [n_images_out][capacity]: FAST score (cv::KeyPoint::response as an integer) */
    size_t desc_off;         /* uint8 [n_images_out][capacity][32] */
    size_t u_right_off;      /* float [n_pairs][capacity] (ORBFE_PACK_STEREO), pair p = image slots 2 p, 2 p + 1 */
    size_t depth_off;        /* float [n_pairs][capacity] */
    size_t bytes;            /* size of the block */
} orbfe_packed_layout;
int orbfe_get_packed_layout(const orbfe_context *ctx, int n_images, int flags, orbfe_packed_layout *out);
/* Gathers the latest batched call's results for image slots 0 .. n_images - 1 into the block (one small kernel on `stream`) and
 * copies it to host_block (>= layout.bytes; pinned memory for an asynchronous copy).  Does not wait; synchronise the stream. */
int orbfe_fetch_batch_packed(orbfe_context *ctx, int n_images, int flags, void *host_block, size_t host_bytes, void *stream);
/* Host-only: the cv::KeyPoint records of out image `out_image` of a fetched block, bit-identical to what orbfe_fetch_batch_async
 * delivers (the reference's own float operations: one product per coordinate).  Descriptors / uRight / depth are read in place
 * at the layout's offsets.  *n = keypoint count. */
int orbfe_expand_packed(const orbfe_context *ctx, const void *host_block, const orbfe_packed_layout *layout, int out_image,
                        orbfe_keypoint *kps, int cap, int *n);
/* orbfe_stereo_batch with the packed block as its result (ORBFE_PACK_STEREO is implied): upload, one stage chain, one gather
 * kernel, one copy into host_block (any host memory; ORBFE_PACK_DIRECT needs pinned memory), synchronised on return.
 * What orbslam2_amd/host/multi_device.h runs per context. */
int orbfe_stereo_batch_packed(orbfe_context *ctx, const uint8_t *images, int n_pairs, int flags, void *host_block, size_t host_bytes);
/* N RGB-D frames in one chain (BASELINE.json config 5 batched; multi-camera RGB-D, CMakeLists.txt:145-146): extraction of the
 * n_images grey device images + Frame::ComputeStereoFromRGBD (src/Frame.cc:645-666) for every slot.  d_depth: n_images depth maps
 * packed one after the other (w*h elements each), float metres or, depth_is_u16 != 0, raw uint16 scaled by depth_map_factor
 * (Tracking's inverted mDepthMapFactor, src/Tracking.cc:151-155,323-324).  Results as after orbfe_enqueue_extract, plus uRight /
 * depth for every image slot. */
int orbfe_enqueue_rgbd(orbfe_context *ctx, const uint8_t *d_gray, const void *d_depth, int depth_is_u16, float depth_map_factor,
                       int n_images, void *stream);
/* Device pointers to the result buffers (for consumers that stay on the GPU):
 * keypoints [max_images][capacity], descriptors [max_images][capacity][32],
 * counts [max_images], u_right/depth [max_images][capacity]. */
int orbfe_device_buffers(orbfe_context *ctx, void **kps, void **desc, void **counts,
                         void **u_right, void **depth);

/* ---- stage timing (HIP events recorded on the stream each enqueue call uses) ----
 * Stages: ingest, pyramid, blur, fast, octree, describe, stereo_match, stereo_median.
 * orbfe_stage_times synchronises, adds up the per-stage elapsed ms of the enqueue calls
 * recorded since the last reset (at most 64 are kept; summed over the stream groups of each call) and
 * reports how many calls that was.
 * orbfe_set_profiling: 0 = off, 1 = events at every stage boundary (each costs a few us of idle GPU),
 * 2 + k = only the two events around stage k (the other stages report 0). */
#define ORBFE_NUM_STAGES 8
int orbfe_set_profiling(orbfe_context *ctx, int enabled);
/* Record the events on every `every`-th enqueue call only (default 1): two events around a stage cost a few microseconds of
 * idle GPU per call, which a timed region then carries; sampling every 4th call keeps the stage's average and 3/4 of that cost out. */
int orbfe_set_profiling_interval(orbfe_context *ctx, int every);
const char *orbfe_stage_name(int stage);
int orbfe_stage_times(orbfe_context *ctx, float *ms, int *calls, int reset);

/* ---- stage taps for parity tests (results of the latest call) ---- */
/* FAST+NMS candidates of (image, level) in the reference's emission order
 * (src/ORBextractor.cc:783-823); coordinates relative to (minBorderX, minBorderY). */
int orbfe_fetch_candidates(orbfe_context *ctx, int image, int level, int32_t *xs, int32_t *ys,
                           int32_t *scores, int cap, int *n);

/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1643-1659) for every (a_i, b_j):
 * host descriptors in, host int32 matrix [na][nb] out. */
int orbfe_hamming_matrix(orbfe_context *ctx, const uint8_t *desc_a, int na,
                         const uint8_t *desc_b, int nb, int32_t *dist);

/* ---- Tracking-thread matchers ----
 * Pointer-rich reference state is passed flattened: a MapPoint* becomes an index into the caller's
 * arrays (GetWorldPos -> pos[i][3], GetDescriptor -> desc[i][32], Observations() -> obs[i]).  All
 * arrays are host memory; calls are synchronous.  Window queries and Hamming distances run on the
 * GPU, the sequentially greedy resolution runs in order on the host (see orbfe_match.hip).
 * Camera intrinsics, bf and the scale factors are those of the context.  Poses are 3x4 row-major
 * [R|t] (the top rows of Frame::mTcw). */
typedef struct orbfe_frame_view { /* what the matchers read from a Frame (include/Frame.h) */
    int32_t n;                      /* N */
    const orbfe_keypoint *keys_un;  /* mvKeysUn */
    const float *u_right;           /* mvuRight, or NULL for monocular */
    const uint8_t *descriptors;     /* mDescriptors, n x 32 */
    float min_x, max_x, min_y, max_y; /* mnMinX, mnMaxX, mnMinY, mnMaxY (ComputeImageBounds) */
    /* 0: the arrays above are uploaded and bucketed on every call (any frame or keyframe).
     * k + 1: the frame IS image slot k of this context's latest extraction call (orbfe_extract / _stereo_frame / _rgbd_frame:
     * slot 0; batched calls: any slot): keypoints (undistorted on the device when orbfe_set_distortion is active) and
     * descriptors are read where the extraction left them in HBM, and the 64 x 48 grid is built once per frame and reused by
     * every later matcher call on it.  n must be that slot's keypoint count (checked: a view of another frame is refused
     * with ORBFE_ERR_INVALID); keys_un must still point to the host copy (the
     * host-side accept rules read angles from it), u_right (n floats, or NULL) is uploaded with every call, descriptors may
     * be NULL.  What Tracking matches against
     * is always the current frame, so this is the Tracking-thread fast path; zero-initialise the struct to stay on the
     * upload path. */
    int32_t device_slot_plus1;
    /* nonzero: the view describes a KeyFrame.  A KeyFrame keeps the frame's grid (cells assigned with the frame's FLOAT bounds and
     * cell size, src/KeyFrame.cc:32-50) but its own bounds are ints initialised from those floats (include/KeyFrame.h:194-197), and
     * KeyFrame::GetFeaturesInArea / IsInImage (src/KeyFrame.cc:563-607) use the ints.  Pass the frame's float bounds
     * (Frame::mnMinX ... are static) in min_x .. max_y and set this flag: cells are assigned with the floats, windows and the
     * in-image test use (float)(int) of them.  Without distortion the bounds are whole numbers and the flag changes nothing. */
    int32_t keyframe;
} orbfe_frame_view;

/* what Frame::isInFrustum (src/Frame.cc:270-326) leaves in a MapPoint for SearchLocalPoints */
typedef struct orbfe_track_point {
    int32_t in_view;                /* mbTrackInView (and !isBad()) */
    float proj_x, proj_y, proj_xr;  /* mTrackProjX, mTrackProjY, mTrackProjXR */
    int32_t level;                  /* mnTrackScaleLevel */
    float view_cos;                 /* mTrackViewCos */
} orbfe_track_point;

/* Frame::GetFeaturesInArea (src/Frame.cc:328-381) over Frame::AssignFeaturesToGrid's 64x48 grid (:231-246),
 * result in the reference's order. */
int orbfe_features_in_area(orbfe_context *ctx, const orbfe_frame_view *frame, float x, float y, float r,
                           int min_level, int max_level, int32_t *out, int cap, int *n);
/* Frame::AssignFeaturesToGrid (src/Frame.cc:231-246) = the 64 x 48 grid itself, as CSR: mGrid[ix][iy] is
 * cell_idx[cell_off[ix * 48 + iy] .. cell_off[ix * 48 + iy + 1]) (ascending keypoint indices = push_back order);
 * cell_off has 64 * 48 + 1 entries, cell_idx fv->n.  For a device-resident frame the grid stays cached for the matcher
 * calls that follow. */
int orbfe_assign_features_to_grid(orbfe_context *ctx, const orbfe_frame_view *fv, int32_t *cell_off, int32_t *cell_idx);
/* nq queries against the same frame in one call (the frame is uploaded and bucketed once): query i's indices are
 * out[out_off[i] .. out_off[i + 1]), in GetFeaturesInArea's order; min_level / max_level may be NULL (-1 for all). */
int orbfe_features_in_area_batch(orbfe_context *ctx, const orbfe_frame_view *fv, int nq, const float *x, const float *y,
                                 const float *r, const int32_t *min_level, const int32_t *max_level,
                                 int32_t *out_off, int32_t *out, int cap);
/* ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:1597-1638) on the sizes of the rotation histogram bins. */
int orbfe_three_maxima(const int32_t *histo_sizes, int L, int *ind1, int *ind2, int *ind3);
/* ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) (src/ORBmatcher.cc:1324-1466).
 * last_valid[i] = LastFrame.mvpMapPoints[i] && !mvbOutlier[i]; cur_has_obs[k] (may be NULL) = CurrentFrame
 * keypoint k already holds a point with Observations() > 0.  cur_match[k] receives the last-frame index
 * assigned to keypoint k, or -1. */
int orbfe_search_by_projection_last(orbfe_context *ctx, const orbfe_frame_view *cur,
                                    const float *Tcw_cur, const float *Tcw_last, int n_last,
                                    const float *last_pos, const uint8_t *last_desc, const int32_t *last_valid,
                                    const int32_t *last_obs, const int32_t *last_octave, const float *last_angle,
                                    const uint8_t *cur_has_obs, float th, int mono, int check_ori,
                                    int32_t *cur_match, int *nmatches);
/* Frame::isInFrustum (src/Frame.cc:270-326) for n map points; max/min_distance are mfMaxDistance / mfMinDistance. */
int orbfe_is_in_frustum(orbfe_context *ctx, const float *Tcw, float min_x, float max_x, float min_y, float max_y,
                        int n, const float *pos, const float *normal, const float *max_distance,
                        const float *min_distance, float viewing_cos_limit, orbfe_track_point *out);
/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:43-135). */
int orbfe_search_by_projection_points(orbfe_context *ctx, const orbfe_frame_view *cur, int n_pts,
                                      const orbfe_track_point *pts, const uint8_t *pt_desc, const int32_t *pt_obs,
                                      const uint8_t *cur_has_obs, float th, float nnratio,
                                      int32_t *cur_match, int *nmatches);
/* ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:1468-1595).
 * kf_valid[i] = pMP && !isBad() && !sAlreadyFound.count(pMP); cur_has_point[k] = CurrentFrame.mvpMapPoints[k] != NULL. */
int orbfe_search_by_projection_kf(orbfe_context *ctx, const orbfe_frame_view *cur, const float *Tcw_cur, int n_kf,
                                  const float *kf_pos, const uint8_t *kf_desc, const int32_t *kf_valid,
                                  const float *kf_angle, const float *kf_max_distance, const float *kf_min_distance,
                                  const uint8_t *cur_has_point, float th, int orb_dist, int check_ori,
                                  int32_t *cur_match, int *nmatches);
/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:400-515); prev_matched is vbPrevMatched ([n1][2], in/out). */
int orbfe_search_for_initialization(orbfe_context *ctx, const orbfe_frame_view *f1, const orbfe_frame_view *f2,
                                    float *prev_matched, int window_size, float nnratio, int check_ori,
                                    int32_t *matches12, int *nmatches);

/* ---- bag of words ----
 * orbfe_vocab_load: fbow::Vocabulary::readFromFile / fromStream (Thirdparty/fbow/src/fbow.cpp:172-191) from a
 * memory blob in the fbow file format (u64 55824124, 120-byte params, block data); the tree stays in HBM. */
int orbfe_vocab_load(orbfe_context *ctx, const uint8_t *blob, size_t size);
/* Bytes of the vocabulary image the context holds, 0 when none is loaded. */
long long orbfe_vocab_bytes(orbfe_context *ctx);
/* Frame::ComputeFboW (src/Frame.cc:395-400) = Vocabulary::transform(desc, level, fBow, fBow2)
 * (Thirdparty/fbow/src/fbow.h:400-444): per descriptor the leaf word id, its weight and the id of the
 * node reached at `level` (4 in ORB-SLAM2).  n == 0 is an error, as in fbow (fbow.cpp:52). */
int orbfe_bow_transform(orbfe_context *ctx, const uint8_t *desc, int n, int level,
                        uint32_t *word_id, float *weight, uint32_t *node_id);
/* The two std::map results as sorted arrays: fBow = (words[i], word_w[i]) with weights summed in feature
 * order; fBow2 = nodes[k] -> node_feat[node_off[k] .. node_off[k+1]) (ascending feature indices).
 * All output arrays need n entries (node_off n+1). */
int orbfe_bow_maps(const uint32_t *word_id, const float *weight, const uint32_t *node_id, int n,
                   uint32_t *words, float *word_w, int *n_words,
                   uint32_t *nodes, int32_t *node_off, int32_t *node_feat, int *n_nodes);
/* ORBmatcher::SearchByFboW(KeyFrame*, Frame&, vpMapPointMatches) (src/ORBmatcher.cc:157-283) on the feature
 * vectors of orbfe_bow_maps.  kf_valid[i] = KF keypoint i has a map point that is not bad.  f_match[j] receives
 * the KF keypoint whose map point frame keypoint j got, or -1. */
int orbfe_search_by_bow(orbfe_context *ctx,
                        const uint32_t *kf_nodes, const int32_t *kf_off, const int32_t *kf_feat, int kf_nnodes,
                        const int32_t *kf_valid, const uint8_t *kf_desc, const float *kf_angle, int n_kf,
                        const uint32_t *f_nodes, const int32_t *f_off, const int32_t *f_feat, int f_nnodes,
                        const uint8_t *f_desc, const float *f_angle, int n_f,
                        float nnratio, int check_ori, int32_t *f_match, int *nmatches);
/* Search part of ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, th) (src/ORBmatcher.cc:821-971, called by
 * LocalMapping::SearchInNeighbors): best_idx[i] = keypoint of `kf` that map point i would be fused with, or -1.  The
 * caller then replaces / adds observations exactly as :943-964 (that mutation never feeds back into the search).
 * pt_valid = pMP && !isBad() && !IsInKeyFrame(pKF); max_distance / min_distance = mfMaxDistance / mfMinDistance; normal =
 * GetNormal(); camera and pyramid parameters are the context's. */
int orbfe_fuse(orbfe_context *ctx, const orbfe_frame_view *kf, const float *Tcw, int n_pts,
               const float *pos, const float *normal, const float *max_distance, const float *min_distance,
               const uint8_t *pt_desc, const int32_t *pt_valid, float th, int32_t *best_idx, int *n_fused);
/* LoopClosing matchers on a Sim3 pose Scw = [sR|t] (3x4 row major; decomposed as src/ORBmatcher.cc:293-298):
 * orbfe_search_by_projection_sim3 = ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (:285-398):
 *   pt_match[i] = keypoint of `kf` matched to point i or -1; kf_matched[k] != 0 marks keypoints that already hold a match
 *   (vpMatched[k] != NULL) -- those, and keypoints taken earlier in the loop, are skipped;
 * orbfe_fuse_sim3 = search part of ORBmatcher::Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint) (:973-1096).
 * pt_valid = !isBad() && not already in the keyframe / matched set. */
int orbfe_search_by_projection_sim3(orbfe_context *ctx, const orbfe_frame_view *kf, const float *Scw, int n_pts,
                                    const float *pos, const float *normal, const float *max_distance, const float *min_distance,
                                    const uint8_t *pt_desc, const int32_t *pt_valid, const uint8_t *kf_matched, float th,
                                    int32_t *pt_match, int *nmatches);
int orbfe_fuse_sim3(orbfe_context *ctx, const orbfe_frame_view *kf, const float *Scw, int n_pts,
                    const float *pos, const float *normal, const float *max_distance, const float *min_distance,
                    const uint8_t *pt_desc, const int32_t *pt_valid, float th, int32_t *best_idx, int *n_fused);
/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (src/ORBmatcher.cc:1098-1322): the map points of each
 * keyframe (arrays with one entry per keypoint slot; valid = pMP && !isBad() && not already matched) are moved into the other
 * camera with the Sim3 (R12 3x3 row major, t12, s12), searched there, and only mutually consistent pairs are kept:
 * match12[i1] = keypoint of KF2 or -1.  T1w / T2w = the keyframes' [R|t] (3x4). */
int orbfe_search_by_sim3(orbfe_context *ctx,
                         const orbfe_frame_view *kf1, const float *T1w, const float *pos1, const float *max_distance1,
                         const float *min_distance1, const uint8_t *pt_desc1, const int32_t *valid1,
                         const orbfe_frame_view *kf2, const float *T2w, const float *pos2, const float *max_distance2,
                         const float *min_distance2, const uint8_t *pt_desc2, const int32_t *valid2,
                         float s12, const float *R12, const float *t12, float th, int32_t *match12, int *n_found);
/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) (src/ORBmatcher.cc:652-819, called by
 * LocalMapping::CreateNewMapPoints): keypoints without a map point (has_mp == 0) paired inside shared vocabulary nodes
 * (feature vectors of orbfe_bow_maps), Hamming <= TH_LOW, monocular pairs away from the epipole, CheckDistEpipolarLine
 * (:138-155) against F12 (3x3 row major).  Cw1 = KF1's camera centre, T2w = KF2's [R|t] (3x4 row major), fx2.. = KF2's
 * intrinsics; u_right < 0 marks a monocular keypoint.  match12[i1] = KF2 keypoint or -1; the reference's pair list is the
 * non-negative entries in index order. */
int orbfe_search_for_triangulation(orbfe_context *ctx,
                                   const uint32_t *nodes1, const int32_t *off1, const int32_t *feat1, int nnodes1,
                                   const orbfe_keypoint *keys1, const float *u_right1, const uint8_t *has_mp1, const uint8_t *desc1, int n1,
                                   const uint32_t *nodes2, const int32_t *off2, const int32_t *feat2, int nnodes2,
                                   const orbfe_keypoint *keys2, const float *u_right2, const uint8_t *has_mp2, const uint8_t *desc2, int n2,
                                   const float *F12, const float *Cw1, const float *T2w, float fx2, float fy2, float cx2, float cy2,
                                   int only_stereo, int check_ori, int32_t *match12, int *nmatches);
/* ---- keyframe database (KeyFrameDatabase, src/KeyFrameDatabase.cc): the keyframes' BoW vectors stay in HBM ----
 * orbfe_kfdb_add = KeyFrameDatabase::add (:38-44) for one keyframe: its fBow as ascending word ids + weights
 * (orbfe_bow_maps); returns the keyframe's index (insertion order = inverted-file order).  orbfe_kfdb_erase (:46-62)
 * removes it from every query; indices are not reused.  orbfe_kfdb_clear (:64-70). */
int orbfe_kfdb_clear(orbfe_context *ctx);
int orbfe_kfdb_add(orbfe_context *ctx, const uint32_t *words, const float *weights, int n, int *kf_index);
int orbfe_kfdb_erase(orbfe_context *ctx, int kf_index);
int orbfe_kfdb_size(orbfe_context *ctx);
/* Per keyframe: number of words shared with the query and fbow::fBow::score(query, keyframe)
 * (Thirdparty/fbow/src/fbow.cpp:206-256: float products summed in double in word order).  Either output may be NULL. */
int orbfe_kfdb_score(orbfe_context *ctx, const uint32_t *q_words, const float *q_w, int nq, int32_t *common, float *score);
/* KeyFrameDatabase::DetectRelocalizationCandidates(Frame*) (:196-307).  covis_off / covis_idx = every keyframe's
 * GetBestCovisibilityKeyFrames(10) list in its order (CSR over the database indices).  reloc_score = the keyframes'
 * persistent mRelocScore, in/out: the reference never initialises it and updates it only for keyframes that pass the
 * common-word filter, so it is caller-owned state (start from zeros).  Candidates in the reference's order. */
int orbfe_detect_reloc_candidates(orbfe_context *ctx, const uint32_t *q_words, const float *q_w, int nq,
                                  const int32_t *covis_off, const int32_t *covis_idx, float *reloc_score,
                                  int32_t *cand, int cap, int *n_cand);
/* Optimizer::PoseOptimization(Frame *pFrame) (src/Optimizer.cc:283-495): motion-only bundle adjustment with g2o's
 * Levenberg solver (Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:59-157), 4 rounds x <= 10 iterations,
 * Huber kernel in rounds 0-2, chi2 classification after every round.  Called after every Tracking matcher call
 * (src/Tracking.cc:875,998,1040,1475,1555,1580).
 *   Tcw       4x4 row-major float, in/out: pFrame->mTcw before, what pFrame->SetPose receives after (untouched when
 *             fewer than 3 correspondences, :404-405)
 *   keys_un   pFrame->mvKeysUn (pt and octave are read), u_right pFrame->mvuRight (< 0: monocular edge)
 *   has_point pFrame->mvpMapPoints[i] != NULL; Xw[3*i..] = that point's GetWorldPos()
 *   outlier   pFrame->mvbOutlier, in/out: written for entries with a point, others keep their value
 *   n_inliers the return value, nInitialCorrespondences - nBad
 * Camera (fx, fy, cx, cy, bf) and mvInvLevelSigma2 are the context's.  FP64 throughout like g2o; sums are reduced in a
 * fixed tree order instead of edge order, so poses agree with the CPU path to about 1e-5 absolute (usually exactly; the
 * solver's stop rules can differ by one tiny iteration), not bit for bit.  The batch form runs one workgroup per problem (frames of independent sequences / relocalisation
 * candidates); offsets[n_problems + 1] delimits each problem's slice of the per-keypoint arrays, Tcw is n_problems x 16. */
int orbfe_pose_optimization(orbfe_context *ctx, float *Tcw, int n, const orbfe_keypoint *keys_un, const float *u_right,
                            const uint8_t *has_point, const float *Xw, uint8_t *outlier, int *n_inliers);
int orbfe_pose_optimization_batch(orbfe_context *ctx, int n_problems, const int32_t *offsets, float *Tcw,
                                  const orbfe_keypoint *keys_un, const float *u_right, const uint8_t *has_point,
                                  const float *Xw, uint8_t *outlier, int32_t *n_inliers);
/* The same on device-resident arrays, asynchronous on `stream` (NULL: the context's stream).  max_keypoints = an upper
 * bound of offsets[k + 1] - offsets[k] (the host cannot see the device-resident offsets; it selects the kernel variant
 * that keeps each problem's edges in LDS). */
int orbfe_enqueue_pose_optimization(orbfe_context *ctx, int n_problems, const int32_t *d_offsets,
                                    const orbfe_keypoint *d_keys_un, const float *d_u_right, const uint8_t *d_has_point,
                                    const float *d_Xw, float *d_Tcw, uint8_t *d_outlier, int32_t *d_n_inliers,
                                    int max_keypoints, void *stream);
/* KeyFrameDatabase::DetectLoopCandidates(KeyFrame *pKF, float minScore) (src/KeyFrameDatabase.cc:73-194; LoopClosing::DetectLoop,
 * src/LoopClosing.cc:131).  connected[k] != 0 marks the keyframes of pKF->GetConnectedKeyFrames() (may be NULL: none); covisibility
 * lists as for the relocalisation query.  Stateless: mLoopScore is only read for keyframes scored by the same call. */
int orbfe_detect_loop_candidates(orbfe_context *ctx, const uint32_t *q_words, const float *q_w, int nq,
                                 const uint8_t *connected, float min_score,
                                 const int32_t *covis_off, const int32_t *covis_idx,
                                 int32_t *cand, int cap, int *n_cand);
/* ORBmatcher::SearchByFboW(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12) (src/ORBmatcher.cc:517-650; LoopClosing and
 * relocalisation).  valid1 / valid2 = the keypoint has a map point that is not bad.  match12[i1] receives the KF2
 * keypoint whose map point KF1 keypoint i1 got, or -1 (n1 entries). */
int orbfe_search_by_bow_kf(orbfe_context *ctx,
                           const uint32_t *nodes1, const int32_t *off1, const int32_t *feat1, int nnodes1,
                           const int32_t *valid1, const uint8_t *desc1, const float *angle1, int n1,
                           const uint32_t *nodes2, const int32_t *off2, const int32_t *feat2, int nnodes2,
                           const int32_t *valid2, const uint8_t *desc2, const float *angle2, int n2,
                           float nnratio, int check_ori, int32_t *match12, int *nmatches);

/* ---- input side: cv::imread(path, cv::IMREAD_UNCHANGED) for PNG files, the per-frame call of every
 * replay driver (Test/Replay/Stereo/stereo_kitti.cc:69-70, stereo_euroc.cc:119-120, RGBD/rgbd_tum.cc:80-81).  Host code
 * (a PNG is one bit-serial DEFLATE stream + neighbour-dependent scanline filters); no context, no GPU needed, thread safe.
 * Output = what imread returns: grey -> 1 channel (1/2/4-bit scaled to 8), RGB -> B,G,R, RGBA and grey+alpha -> B,G,R,A,
 * palette -> B,G,R (B,G,R,A with tRNS), 16-bit samples stay 16 bit in host byte order; Adam7 interlacing supported.
 * A file imread would refuse (bad signature / CRC / stream) returns ORBFE_ERR_INVALID; orbfe_png_last_error() says why
 * (per thread). */
const char *orbfe_png_last_error(void);
int orbfe_png_info(const uint8_t *file, size_t size, int *width, int *height, int *channels, int *bit_depth);
/* dst receives height rows of width * channels * (bit_depth / 8) bytes, dst_stride bytes apart (0 = tightly packed). */
int orbfe_png_decode(const uint8_t *file, size_t size, uint8_t *dst, size_t dst_bytes, size_t dst_stride,
                     int *width, int *height, int *channels, int *bit_depth);
/* n files of one camera stream (common geometry, checked) decoded by `threads` host threads (0 = all cores) into
 * dst[i * image_bytes ..]: pinned memory here is the staging block of the upload that feeds orbfe_enqueue_*. */
int orbfe_png_decode_batch(const uint8_t *const *files, const size_t *sizes, int n, uint8_t *dst, size_t image_bytes,
                           int width, int height, int channels, int bit_depth, int threads);

#ifdef __cplusplus
}
#endif
#endif
