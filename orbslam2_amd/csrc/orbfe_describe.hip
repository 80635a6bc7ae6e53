// orbfe_describe.hip -- IC_Angle + computeOrbDescriptor + keypoint records (src/ORBextractor.cc:72-142,831-846,909-915) and the stereo row lists.
#include "orbfe_common.hpp"
#include "orbfe_rowlist.hpp"

// the 256 tests come from DeviceBuffers::pattern (the context's own copy, as ORBextractor keeps one: src/ORBextractor.cc:442-444)

// ---------------------------------------------------------------------------
// orientation + descriptor + final keypoint record: one wave per keypoint slot
// ---------------------------------------------------------------------------
// The kernel is bound by vector-memory INSTRUCTION issue: the texture addresser spends >= 16 cycles on a wave64 load
// whatever its width (TA busy 73 % with twelve dword loads per keypoint), so a keypoint's patches are fetched with THREE
// 128-bit loads and all per-pixel / per-sample accesses are LDS reads:
//   raw patch (IC_Angle, hp == 15): 31 rows x 32 bytes from column cx - 15 (unaligned 16-byte loads are fine on this memory
//     system): lane = (row, half), one load for the whole patch, stored lane-linear (row pitch 32 B = the 31 x 8 word grid of
//     the moment weights, no byte alignment step);
//   blurred patch (descriptor): 40 rows from a row that is a multiple of 4 = ten tile rows, in each of which the patch's ten
//     4 x 4 px blocks are 160 contiguous bytes (blur_kernel's layout): lane = (tile row, block), two loads.
// A wave handles DS_KPW consecutive keypoint slots: the table staging and the slot bookkeeping are paid once per 4 * DS_KPW
// keypoints, and the patch of keypoint i+1 is fetched into registers while keypoint i is computed from LDS.
#define DS_PATCH_W 40 // bytes per staged row of the blurred patch (10 words: 37 + 3 px at any alignment) and of the generic raw patch
#define DS_RAW_W 32   // bytes per staged row of the raw patch, hp == 15
#define DS_KPW 4      // keypoint slots per wave
#define DS_BLR_ROWS 40 // blurred patch rows staged: the 37 the descriptor can reach, from a row that is a multiple of 4


__global__ __launch_bounds__(256, 8) void describe_generic_kernel(DeviceConfig cfg, DeviceBuffers buf, int n_images, int stereo ORBFE_CUT_PARAM)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dm[];
    // XCD-aware block -> (image, block) map: workgroups are dealt round-robin over the 8 XCDs, so block
    // b runs on XCD b % 8 (placement is a speed assumption only).  All blocks of one image are given to
    // one XCD, whose 4 MiB L2 then holds that image's raw + blurred pyramid (3.3 MB) while its ~2000
    // overlapping 31x31 / 37x37 patches are read, instead of every patch row coming from the MALL.
    const int bpi = (cfg.sel_total + 4 * DS_KPW - 1) / (4 * DS_KPW); // blocks per image
    int img, blk;
    if (!xcd_map_magic(bpi, n_images, cfg.xcd_magic, img, blk)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform values must be provably so: they feed scalar addresses
    const int slot0 = blk * (4 * DS_KPW) + wave * DS_KPW;
    const int hp = cfg.half_patch;
    const int raw_rows = 2 * hp + 1;
    const int *sel_cnt = buf.sel_cnt + (size_t)img * cfg.nlevels;
    if (blk == 0 && tid == 0) {
        int tot = 0;
        for (int l = 0; l < cfg.nlevels; l++) tot += sel_cnt[l];
        buf.kp_cnt[img] = tot;
    }
    // block-shared tables: patch offsets (patch_n shorts) | pattern (256 words)
    int16_t *s_uv = (int16_t *)s_dm;
    int *s_pat = (int *)(s_dm + ((cfg.patch_n * 2 + 15) & ~15));
    const bool dot_moments = hp == 15; // the reference's HALF_PATCH_SIZE: raw patch by one 128-bit load, moments by byte dot products
    const int raw_bytes = dot_moments ? 32 * DS_RAW_W : ((raw_rows * DS_PATCH_W + 15) & ~15);
    uint8_t *s_raw = (uint8_t *)(s_pat + 256) + wave * (raw_bytes + DS_BLR_ROWS * DS_PATCH_W);
    uint8_t *s_blr = s_raw + raw_bytes;
    if (cfg.half_patch != 15) // the offset list is only read by the generic moment loop
        for (int i = tid; i < cfg.patch_n / 2; i += 256) ((int *)s_uv)[i] = ((const int *)buf.patch_uv)[i];
    s_pat[tid] = (int)buf.pattern[tid];
    // per-wave slot data, one slot per lane (lanes < DS_KPW), issued before the barrier
    const int my_slot = slot0 + (lane < DS_KPW ? lane : 0);
    const bool my_in = lane < DS_KPW && my_slot < cfg.sel_total;
    const int level_l = my_in ? buf.slot_level[my_slot] : 0;
    const uint32_t xy_l = my_in ? buf.sel_xy[(size_t)img * cfg.sel_total + my_slot] : 0u;
    const int score_l = my_in ? buf.sel_sc[(size_t)img * cfg.sel_total + my_slot] : 0;
    const int c_l = lane < cfg.nlevels ? sel_cnt[lane] : 0;
    __syncthreads();
    int inc = c_l;
#pragma unroll
    for (int o = 1; o < ORBFE_MAX_LEVELS; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    const int excl = inc - c_l; // keypoints of the lower levels
    if (ORBFE_CUT(1)) return;

    const int r0 = lane / (DS_PATCH_W / 4), c0 = lane - r0 * (DS_PATCH_W / 4); // generic raw path
    uint4 pr = {0u, 0u, 0u, 0u}, pb[2];
    // per-lane constants.  Raw: row (lanes 62, 63 repeat row 30: same bytes to LDS words the moment weights ignore) and 16 * half.
    // Blurred, k = 0, 1: block i = lane + 64 k of the 10 x 10 grid (i >= 100 repeats block 99: same value to the same place)
    // as tile row | 16 * block << 8 | LDS byte offset of the block's first row (tile row * 160 + 4 * block) << 16.
    const int raw_row = (lane >> 1) < raw_rows ? (lane >> 1) : raw_rows - 1, raw_h16 = (lane & 1) << 4;
    uint32_t wb[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int i = lane + 64 * k < 100 ? lane + 64 * k : 99;
        const int tr = (i * 6554) >> 16, bl = i - 10 * tr; // / 10
        wb[k] = (uint32_t)tr | ((uint32_t)(16 * bl) << 8) | ((uint32_t)(tr * 4 * DS_PATCH_W + 4 * bl) << 16);
    }
    auto fetch_raw = [&](const uint8_t *base /* uniform: pixel (cx - 15, cy - 15) */, int pitch) {
        pr = load16_unaligned(base + (unsigned)(__mul24(raw_row, pitch) + raw_h16)); // 16 bytes at any alignment: one global_load_dwordx4
    };
    // the blurred pyramid is stored in 32 x 4 px tiles of 128 B, each eight 4 x 4 px blocks of 16 B (blur_kernel): the block
    // of pixels X .. X + 3 (X a multiple of 4) x rows 4 T .. 4 T + 3 is at T * tile_row_bytes + 4 * X, its rows 4 bytes apart
    auto fetch_blr = [&](const uint8_t *base /* uniform: block (x0, y0 >> 2) of the level */, unsigned tile_row_bytes) {
#pragma unroll
        for (int k = 0; k < 2; k++)
            pb[k] = *(const uint4 *)(base + (__umul24(wb[k] & 0xffu, tile_row_bytes) + ((wb[k] >> 8) & 0xffu)));
    };
    // slot i of this wave: uniform keypoint data; returns false if the slot holds no keypoint
    int level = 0, cx = 0, cy = 0, score = 0, out = 0;
    auto slot_data = [&](int i) -> bool {
        if (slot0 + i >= cfg.sel_total) return false;
        level = __builtin_amdgcn_readlane(level_l, i);
        const uint32_t xy = (uint32_t)__builtin_amdgcn_readlane((int)xy_l, i);
        score = __builtin_amdgcn_readlane(score_l, i);
        const int k = slot0 + i - cfg.lv[level].sel_off;
        if (k >= __builtin_amdgcn_readlane(c_l, level)) return false; // readlane (not a shuffle): the result is a scalar
        out = k + __builtin_amdgcn_readlane(excl, level);
        cx = (int)(xy & 0xffffu) + cfg.min_border;
        cy = (int)(xy >> 16) + cfg.min_border;
        return true;
    };
    auto prefetch_raw = [&](int i) -> bool {
        if (i >= DS_KPW || !slot_data(i)) return false;
        if (dot_moments)
        {
            int lp;
            const uint8_t *li = level_image(cfg, buf, img, level, lp);
            fetch_raw(li + (ptrdiff_t)(cy - hp) * lp + (cx - hp), lp);
        }
        return true;
    };
    auto prefetch_blur = [&](int i) -> bool {
        if (i >= DS_KPW || !slot_data(i)) return false;
        const LevelInfo &L = cfg.lv[level];
        const unsigned trb = (unsigned)L.blur_tx << 7;
        fetch_blr(buf.blur + (size_t)img * cfg.blur_bytes + L.blur_off + (size_t)((cy - 18) >> 2) * trb + 4 * ((cx - 18) & ~3), trb);
        return true;
    };

    // Pass 1: raw patches -> IC_Angle moments of the wave's keypoints (lane i keeps keypoint i's);
    // then fastAtan2 and the sin / cos ONCE for all of them (lane i computes keypoint i's: those ~150 scalar-like
    // instructions, part of them double precision, would otherwise be repeated per keypoint by all 64 lanes);
    // pass 2: blurred patches -> descriptors and keypoint records.
    int m10_l = 0, m01_l = 0;
    // hp == 15 (the reference's HALF_PATCH_SIZE): the 31 x 31 patch is 31 rows x 8 four-pixel words (u = -15 .. 16) in LDS,
    // word s = lane + 64 k of that grid per lane, and the moments come from byte dot products against per-lane constant
    // weight words: m10 = sum (u + 16) I - 16 sum I, m01 = sum_rows v * (row sum I), the circle mask folded into the weights
    // (0 outside |u| <= umax[|v|]).  Integer sums: the order does not matter, the result is IC_Angle's exactly.
    uint32_t mw_u[4] = {0, 0, 0, 0}, mw_1[4] = {0, 0, 0, 0};
    int mw_v[4] = {0, 0, 0, 0};
    if (dot_moments) { // host-built per-lane constants (orbfe_api.hip), 48 bytes per lane: three 128-bit loads per wave
        const uint4 *mt = (const uint4 *)buf.mom_tab + 3 * lane;
        const uint4 a = mt[0], b = mt[1], c = mt[2];
        mw_u[0] = a.x; mw_u[1] = a.y; mw_u[2] = a.z; mw_u[3] = a.w;
        mw_1[0] = b.x; mw_1[1] = b.y; mw_1[2] = b.z; mw_1[3] = b.w;
#pragma unroll
        for (int k = 0; k < 4; k++) mw_v[k] = (int)(int8_t)(c.x >> (8 * k));
    }
    bool have = prefetch_raw(0);
    for (int i = 0; i < DS_KPW; i++) {
        const bool cur = have;
        int kx = 0, ky = 0, lv = 0;
        if (cur) {
            slot_data(i);
            kx = cx; ky = cy; lv = level;
            if (dot_moments) {
                ((uint4 *)s_raw)[lane] = pr;
            } else { // other patch sizes: aligned words straight through (no prefetch), row pitch DS_PATCH_W
                int lp;
                const uint8_t *gp = level_image(cfg, buf, img, lv, lp) + (ptrdiff_t)__mul24(ky - hp + r0, lp) + ((kx - hp) & ~3) + 4 * c0; // level 0 in place: the words may be unaligned (fine on this memory system)
                const int step = 6 * lp + 16, wrap = lp - DS_PATCH_W;
                int c = c0;
                for (int w = lane; w < raw_rows * (DS_PATCH_W / 4); w += 64) {
                    ((uint32_t *)s_raw)[w] = *(const uint32_t *)gp;
                    gp += step; c += 4;
                    if (c >= DS_PATCH_W / 4) { c -= DS_PATCH_W / 4; gp += wrap; }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0); // LDS writes of this wave are visible to its own later reads in order
        __builtin_amdgcn_wave_barrier();
        have = prefetch_raw(i + 1); // in flight while keypoint i is computed
        if (!cur || ORBFE_CUT(2)) continue;
        // IC_Angle (src/ORBextractor.cc:72-99): integer moments over the circular patch (host-built offset
        // table, padded with (0,0) entries that contribute nothing)
        const int xr = (kx - hp) & ~3;
        int m10 = 0, m01 = 0;
        if (dot_moments) {
            unsigned acc_u = 0;
            int acc_1 = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t px = ((const uint32_t *)s_raw)[lane + 64 * k];
                acc_u = __builtin_amdgcn_udot4(px, mw_u[k], acc_u, false);
                const int t = (int)__builtin_amdgcn_udot4(px, mw_1[k], 0u, false);
                acc_1 += t;
                m01 += mw_v[k] * t;
            }
            m10 = (int)acc_u - 16 * acc_1;
        } else {
            const uint8_t *pc = s_raw + hp * DS_PATCH_W + (kx - xr);
            for (int kk = lane; kk < cfg.patch_n; kk += 64) {
                const int uv = s_uv[kk];
                const int u = (int)(int8_t)(uv & 0xff), v = (int)(int8_t)((uv >> 8) & 0xff);
                const int I = pc[__mul24(v, DS_PATCH_W) + u];
                m10 += u * I;
                m01 += v * I;
            }
        }
        m10 = wave_sum_i32(m10);
        m01 = wave_sum_i32(m01);
        if (lane == i) { m10_l = m10; m01_l = m01; }
    }
    have = prefetch_blur(0); // in flight during the angle arithmetic
    const float angle_l = fast_atan2_deg((float)m01_l, (float)m10_l);
    const float factor_pi = __uint_as_float(0x3c8efa35u); // (float)(CV_PI/180.f)
    float a_l, b_l;
    sincos_det(__fmul_rn(angle_l, factor_pi), &b_l, &a_l);

    int rl_lv = -1;          // lane i < DS_KPW: level | index << 8 of the wave's i-th keypoint (-1: none), its x and y
    float rl_x = 0.f, rl_y = 0.f, size_keep = 0.f;
    unsigned long long dkeep = 0ull;
    for (int i = 0; i < DS_KPW; i++) {
        const bool cur = have;
        int lv = 0, kx = 0, ky = 0, ksc = 0, kout = 0;
        if (cur) {
            slot_data(i);
            lv = level; kx = cx; ky = cy; ksc = score; kout = out;
#pragma unroll
            for (int k = 0; k < 2; k++) { // the block's four rows, one LDS row apart
                uint32_t *d = (uint32_t *)(s_blr + (wb[k] >> 16));
                d[0] = pb[k].x; d[DS_PATCH_W / 4] = pb[k].y; d[2 * (DS_PATCH_W / 4)] = pb[k].z; d[3 * (DS_PATCH_W / 4)] = pb[k].w;
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        have = prefetch_blur(i + 1);
        if (!cur || ORBFE_CUT(2)) continue;
        const LevelInfo &L = cfg.lv[lv];
        const int xb = (kx - 18) & ~3;
        const float angle = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(angle_l), i));
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_l), i));
        const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b_l), i));
        if (ORBFE_CUT(3)) { if (lane == 0) buf.depth[(size_t)img * cfg.sel_total + kout] = angle; continue; }

        // computeOrbDescriptor (src/ORBextractor.cc:103-142)
        const uint8_t *center = s_blr + (18 + ((ky - 18) & 3)) * DS_PATCH_W + (kx - xb);
        unsigned long long bits[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int pw = s_pat[r * 64 + lane]; // (x0, y0, x1, y1) as 4 signed bytes
            const float x0 = (float)(int)(int8_t)(pw & 0xff), y0 = (float)(int)(int8_t)((pw >> 8) & 0xff);
            const float x1 = (float)(int)(int8_t)((pw >> 16) & 0xff), y1 = (float)(pw >> 24);
            const int rr0 = (int)rintf(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
            const int cc0 = (int)rintf(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
            const int rr1 = (int)rintf(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
            const int cc1 = (int)rintf(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
            const int t0 = center[__mul24(rr0, DS_PATCH_W) + cc0];
            const int t1 = center[__mul24(rr1, DS_PATCH_W) + cc1];
            bits[r] = __ballot(t0 < t1);
        }
        // results stay in registers until the wave's last keypoint is done (lane 4 i + w: descriptor word w of keypoint i; lane i:
        // its record): a store issued here would be waited for by the next keypoint's s_waitcnt (gfx9 counts stores in vmcnt)
        if ((lane >> 2) == i) dkeep = (lane & 3) == 0 ? bits[0] : ((lane & 3) == 1 ? bits[1] : ((lane & 3) == 2 ? bits[2] : bits[3]));
        float px = (float)kx, py = (float)ky;
        if (lv != 0) { px = __fmul_rn(px, L.scale); py = __fmul_rn(py, L.scale); }
        if (lane == i) { rl_lv = lv | (kout << 8); rl_x = px; rl_y = py; size_keep = (float)L.scaled_patch; } // also for the stereo row lists below
        (void)ksc; (void)angle;
    }
    {
        const int lvk = __shfl(rl_lv, lane >> 2, 64); // level | index << 8 of the keypoint this lane holds a descriptor word of
        if (lane < 4 * DS_KPW && lvk >= 0)
            *(unsigned long long *)(buf.desc + ((size_t)img * cfg.sel_total + (lvk >> 8)) * 32 + (lane & 3) * 8) = dkeep;
        if (lane < DS_KPW && rl_lv >= 0) {
            KeyPointPOD kp;
            kp.x = rl_x; kp.y = rl_y;
            kp.size = size_keep;
            kp.angle = angle_l;            // lane i computed keypoint i's angle
            kp.response = (float)score_l;  // and loaded its slot's score
            kp.octave = rl_lv & 255;
            kp.class_id = -1;
            ((KeyPointPOD *)buf.kps)[(size_t)img * cfg.sel_total + (rl_lv >> 8)] = kp;
        }
    }
    (void)stereo; // the stereo row lists come from stereo_rowlist_kernel (orbfe_stereo.hip) for this kernel's geometries
}


// ---------------------------------------------------------------------------
// describe_kernel: the reference's geometry (HALF_PATCH_SIZE 15), round 3.  Same outputs as describe_generic_kernel, bit for bit;
// what changed is where the time went:
//  * IC_Angle needs no LDS: a lane's one 128-bit load IS its share of the 31 x 31 patch (lane = 2 * row + half: 16 pixels of
//    one row), so the moments are eight v_dot4_u32_u8 of the loaded registers against per-lane weight words (host-built for
//    that layout; u-weights and mask, one v per lane) -- no LDS store, wait, barrier and four LDS reads per keypoint;
//  * all four raw patches of the wave are requested before the first is used, and the blurred patches are prefetched TWO
//    keypoints ahead (the per-keypoint arithmetic is ~500 cycles, a miss ~2 us: one patch ahead left the waves waiting 41 %
//    of their time, round-2 counters);
//  * the rBRIEF pattern sits in LDS as four floats per test (one ds_read_b128 instead of a word and eight unpack / convert
//    instructions per round), and cvRound is one fp32 add of 1.5 * 2^23 (round-half-even in the add's own rounding; exact for
//    |x| < 2^22): the sum's low bits are the integer, its bias goes into the LDS base address through a 24-bit multiply.
// ---------------------------------------------------------------------------
typedef const __attribute__((address_space(3))) uint8_t *ds_lds_cptr; // LDS pointers are 32 bits wide
#define DS_LGKM0 0xc07f // s_waitcnt lgkmcnt(0) only: vmcnt / expcnt fields at their maxima (prefetches stay in flight)
__global__ __launch_bounds__(256, 8) void describe_kernel(DeviceConfig cfg, DeviceBuffers buf, int n_images, int stereo /* row-list workgroups ahead of the descriptor ones; 0: no stereo stage */ ORBFE_CUT_PARAM)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Stereo: the first `stereo` workgroups (a multiple of 8: the XCD map of the rest) build the pairs' row lists, four independent
    // waves each (orbfe_rowlist.hpp: they need the quadtree's output only).  As a launch of its own that work takes 19.5 us, 8 of
    // them launch and first-load latency; here it rides beside the descriptor waves.
    int bid = blockIdx.x;
    if (stereo) {
        if (bid < stereo) {
            const int bpp = rowlist_blocks(cfg.height, RL_ROWS_FUSED);
            const int e = bid * 4 + wave;
            const int pair = __builtin_amdgcn_readfirstlane(small_div(e, bpp));
            if (pair < (n_images >> 1)) rowlist_wave<RL_ROWS_FUSED>(cfg, buf, pair, e - pair * bpp, s_dm + wave * RL_LDS_BYTES);
            return;
        }
        bid -= stereo;
    }
    const int bpi = (cfg.sel_total + 4 * DS_KPW - 1) / (4 * DS_KPW); // blocks per image (XCD-aware map: see describe_generic_kernel)
    int img, blk;
    if (!xcd_map_of_magic(bid, bpi, n_images, cfg.xcd_magic, img, blk)) return;
    const int slot0 = blk * (4 * DS_KPW) + wave * DS_KPW;
    const int *sel_cnt = buf.sel_cnt + (size_t)img * cfg.nlevels;
    if (blk == 0 && tid == 0) {
        int tot = 0;
        for (int l = 0; l < cfg.nlevels; l++) tot += sel_cnt[l];
        buf.kp_cnt[img] = tot;
    }
    // block-shared: the 256 tests as (x0, y0, x1, y1) floats; per wave: the blurred patch (40 rows x 40 bytes)
    float4 *s_patf = (float4 *)s_dm;
    uint8_t *s_blr = s_dm + 256 * sizeof(float4) + wave * (DS_BLR_ROWS * DS_PATCH_W);
    {
        const int pw = (int)buf.pattern[tid];
        s_patf[tid] = make_float4((float)(int)(int8_t)(pw & 0xff), (float)(int)(int8_t)((pw >> 8) & 0xff), (float)(int)(int8_t)((pw >> 16) & 0xff), (float)(pw >> 24));
    }
    const int my_slot = slot0 + (lane < DS_KPW ? lane : 0);
    const bool my_in = lane < DS_KPW && my_slot < cfg.sel_total;
    const int level_l = my_in ? buf.slot_level[my_slot] : 0;
    // DeviceConfig::proc_order: position -> keypoint through octree3_kernel's spatially ordered copy (proc_xy, proc_meta = slot | score << 24),
    // so that the keypoints of a workgroup share cache lines; otherwise the keypoint of the slot itself.  Positions beyond the level's
    // count are never used.
    uint32_t xy_l = 0u;
    int score_l = 0, pslot_l = my_slot;
    if (my_in) {
        if (cfg.proc_order) {
            xy_l = buf.proc_xy[(size_t)img * cfg.sel_total + my_slot];
            const uint32_t m = buf.proc_meta[(size_t)img * cfg.sel_total + my_slot];
            pslot_l = (int)(m & 0xffffffu); score_l = (int)(m >> 24);
        } else {
            xy_l = buf.sel_xy[(size_t)img * cfg.sel_total + my_slot];
            score_l = buf.sel_sc[(size_t)img * cfg.sel_total + my_slot];
        }
    }
    const int c_l = lane < cfg.nlevels ? sel_cnt[lane] : 0;
    // per-lane moment weights (host-built, orbfe_api.hip): 4 words (u + 16 inside the circle, else 0), 4 words (1 / 0), v
    const uint4 *mt = (const uint4 *)buf.mom_tab + 3 * lane;
    const uint4 mwu = mt[0], mw1 = mt[1];
    const int mv = (int)mt[2].x;
    __syncthreads();
    int inc = c_l;
#pragma unroll
    for (int o = 1; o < ORBFE_MAX_LEVELS; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    const int excl = inc - c_l; // keypoints of the lower levels
    if (ORBFE_CUT(1)) return;

    // the wave's keypoints: uniform data of slot i (hv false: the slot holds no keypoint)
    bool hv[DS_KPW];
    int lvv[DS_KPW], cxv[DS_KPW], cyv[DS_KPW], outv[DS_KPW];
#pragma unroll
    for (int i = 0; i < DS_KPW; i++) {
        hv[i] = false; lvv[i] = 0; cxv[i] = 0; cyv[i] = 0; outv[i] = 0;
        if (slot0 + i < cfg.sel_total) {
            const int level = __builtin_amdgcn_readlane(level_l, i);
            const uint32_t xy = (uint32_t)__builtin_amdgcn_readlane((int)xy_l, i);
            const int k = slot0 + i - cfg.lv[level].sel_off;
            if (k < __builtin_amdgcn_readlane(c_l, level)) { // readlane (not a shuffle): the result is a scalar
                hv[i] = true; lvv[i] = level;
                outv[i] = __builtin_amdgcn_readlane(pslot_l, i) - cfg.lv[level].sel_off + __builtin_amdgcn_readlane(excl, level);
                cxv[i] = (int)(xy & 0xffffu) + cfg.min_border;
                cyv[i] = (int)(xy >> 16) + cfg.min_border;
            }
        }
    }
    // Pass 1: every raw patch of the wave in flight at once; lane = 2 * row + half holds pixels u = -15 + 16 * half .. + 15 of
    // row v = row - 15 (lanes 62 / 63 repeat row 30 with zero weights)
    const int raw_row = (lane >> 1) < 31 ? (lane >> 1) : 30, raw_h16 = (lane & 1) << 4;
    uint4 pr[DS_KPW];
#pragma unroll
    for (int i = 0; i < DS_KPW; i++) {
        pr[i] = make_uint4(0u, 0u, 0u, 0u);
        if (hv[i]) {
            int lp;
            const uint8_t *base = level_image(cfg, buf, img, lvv[i], lp) + (ptrdiff_t)(cyv[i] - 15) * lp + (cxv[i] - 15);
            pr[i] = load16_unaligned(base + (unsigned)(__mul24(raw_row, lp) + raw_h16)); // 16 bytes at any alignment: one global_load_dwordx4
        }
    }
    // blurred patches: block i = lane + 64 k of the 10 x 10 grid of 4 x 4 px blocks (layout: describe_generic_kernel)
    uint32_t wb[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int i = lane + 64 * k < 100 ? lane + 64 * k : 99;
        const int tr = (i * 6554) >> 16, bl = i - 10 * tr; // / 10
        wb[k] = (uint32_t)tr | ((uint32_t)(16 * bl) << 8) | ((uint32_t)(tr * 4 * DS_PATCH_W + 4 * bl) << 16);
    }
    uint4 pb[2][2];
    auto fetch_blr = [&](int i, uint4 *dst) {
        const LevelInfo &L = cfg.lv[lvv[i]];
        const unsigned trb = (unsigned)L.blur_tx << 7;
        const uint8_t *base = buf.blur + (size_t)img * cfg.blur_bytes + L.blur_off + (size_t)((cyv[i] - 18) >> 2) * trb + 4 * ((cxv[i] - 18) & ~3);
#pragma unroll
        for (int k = 0; k < 2; k++)
            dst[k] = *(const uint4 *)(base + (__umul24(wb[k] & 0xffu, trb) + ((wb[k] >> 8) & 0xffu)));
    };
    if (ORBFE_CUT(2)) return;
    int m10_l = 0, m01_l = 0;
#pragma unroll
    for (int i = 0; i < DS_KPW; i++) {
        if (!hv[i]) continue;
        // IC_Angle (src/ORBextractor.cc:72-99): m10 = sum u I = sum (u + 16) I - 16 sum I, m01 = sum v I; integer sums, any order
        unsigned acc_u = __builtin_amdgcn_udot4(pr[i].x, mwu.x, 0u, false);
        acc_u = __builtin_amdgcn_udot4(pr[i].y, mwu.y, acc_u, false);
        acc_u = __builtin_amdgcn_udot4(pr[i].z, mwu.z, acc_u, false);
        acc_u = __builtin_amdgcn_udot4(pr[i].w, mwu.w, acc_u, false);
        unsigned acc_1 = __builtin_amdgcn_udot4(pr[i].x, mw1.x, 0u, false);
        acc_1 = __builtin_amdgcn_udot4(pr[i].y, mw1.y, acc_1, false);
        acc_1 = __builtin_amdgcn_udot4(pr[i].z, mw1.z, acc_1, false);
        acc_1 = __builtin_amdgcn_udot4(pr[i].w, mw1.w, acc_1, false);
        const int m10 = wave_sum_i32((int)acc_u - 16 * (int)acc_1);
        const int m01 = wave_sum_i32(mv * (int)acc_1);
        if (lane == i) { m10_l = m10; m01_l = m01; }
    }
    if (hv[0]) fetch_blr(0, pb[0]); // in flight during the angle arithmetic
    if (hv[1]) fetch_blr(1, pb[1]);
    const float angle_l = fast_atan2_deg((float)m01_l, (float)m10_l);
    const float factor_pi = __uint_as_float(0x3c8efa35u); // (float)(CV_PI/180.f)
    float a_l, b_l;
    sincos_det(__fmul_rn(angle_l, factor_pi), &b_l, &a_l);

    int rl_lv = -1;          // lane i < DS_KPW: level | index << 8 of the wave's i-th keypoint (-1: none), its x and y
    float rl_x = 0.f, rl_y = 0.f, size_keep = 0.f;
    unsigned long long dkeep = 0ull;
    const float magic = 12582912.0f; // 1.5 * 2^23: (x + magic) holds x rounded half-to-even in its low mantissa bits
#pragma unroll
    for (int i = 0; i < DS_KPW; i++) {
        if (hv[i]) {
            const uint4 *q = pb[i & 1];
#pragma unroll
            for (int k = 0; k < 2; k++) { // the block's four rows, one LDS row apart
                uint32_t *d = (uint32_t *)(s_blr + (wb[k] >> 16));
                d[0] = q[k].x; d[DS_PATCH_W / 4] = q[k].y; d[2 * (DS_PATCH_W / 4)] = q[k].z; d[3 * (DS_PATCH_W / 4)] = q[k].w;
            }
        }
        __builtin_amdgcn_s_waitcnt(DS_LGKM0); // this wave's LDS writes are done (its later reads follow in order); the prefetches keep flying
        __builtin_amdgcn_wave_barrier();
        if (i + 2 < DS_KPW) { if (hv[i + 2 < DS_KPW ? i + 2 : 0]) fetch_blr(i + 2 < DS_KPW ? i + 2 : 0, pb[i & 1]); } // two keypoints ahead, into the registers just stored
        if (!hv[i]) continue;
        const int lv = lvv[i], kx = cxv[i], ky = cyv[i];
        const LevelInfo &L = cfg.lv[lv];
        const int xb = (kx - 18) & ~3;
        const float angle = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(angle_l), i));
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_l), i));
        const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b_l), i));
        if (ORBFE_CUT(3)) { if (lane == 0) buf.depth[(size_t)img * cfg.sel_total + outv[i]] = angle; continue; }

        // computeOrbDescriptor (src/ORBextractor.cc:103-142): center[cvRound(x b + y a) * step + cvRound(x a - y b)].
        // With the biased integers ir = 0x4b400000 + r and ic = 0x4b400000 + c (bit patterns of the two sums):
        // (ir & 0xffffff) * 40 + ic = (0x400000 + r) * 40 + 0x4b400000 + c = r * 40 + c + 0x55400000 (mod 2^32)
        const unsigned center_biased = (unsigned)(uintptr_t)(ds_lds_cptr)(s_blr + (18 + ((ky - 18) & 3)) * DS_PATCH_W + (kx - xb)) - 0x55400000u;
        unsigned long long bits[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float4 pt = s_patf[r * 64 + lane];
            const unsigned ir0 = __float_as_uint(__fadd_rn(__fadd_rn(__fmul_rn(pt.x, b), __fmul_rn(pt.y, a)), magic));
            const unsigned ic0 = __float_as_uint(__fadd_rn(__fsub_rn(__fmul_rn(pt.x, a), __fmul_rn(pt.y, b)), magic));
            const unsigned ir1 = __float_as_uint(__fadd_rn(__fadd_rn(__fmul_rn(pt.z, b), __fmul_rn(pt.w, a)), magic));
            const unsigned ic1 = __float_as_uint(__fadd_rn(__fsub_rn(__fmul_rn(pt.z, a), __fmul_rn(pt.w, b)), magic));
            const unsigned a0 = __umul24(ir0, DS_PATCH_W) + ic0 + center_biased;
            const unsigned a1 = __umul24(ir1, DS_PATCH_W) + ic1 + center_biased;
            const int t0 = *(ds_lds_cptr)(uintptr_t)a0, t1 = *(ds_lds_cptr)(uintptr_t)a1; // ds_read_u8 at a 32-bit LDS address
            bits[r] = __ballot(t0 < t1);
        }
        // results stay in registers until the wave's last keypoint is done (lane 4 i + w: descriptor word w of keypoint i; lane i:
        // its record): a store issued here would be waited for by a later keypoint's vmcnt wait (gfx9 counts stores in vmcnt)
        if ((lane >> 2) == i) dkeep = (lane & 3) == 0 ? bits[0] : ((lane & 3) == 1 ? bits[1] : ((lane & 3) == 2 ? bits[2] : bits[3]));
        float px = (float)kx, py = (float)ky;
        if (lv != 0) { px = __fmul_rn(px, L.scale); py = __fmul_rn(py, L.scale); }
        if (lane == i) { rl_lv = lv | (outv[i] << 8); rl_x = px; rl_y = py; size_keep = (float)L.scaled_patch; } // also for the stereo row lists
        (void)angle;
    }
    {
        const int lvk = __shfl(rl_lv, lane >> 2, 64); // level | index << 8 of the keypoint this lane holds a descriptor word of
        if (lane < 4 * DS_KPW && lvk >= 0)
            *(unsigned long long *)(buf.desc + ((size_t)img * cfg.sel_total + (lvk >> 8)) * 32 + (lane & 3) * 8) = dkeep;
        if (lane < DS_KPW && rl_lv >= 0) {
            KeyPointPOD kp;
            kp.x = rl_x; kp.y = rl_y;
            kp.size = size_keep;
            kp.angle = angle_l;            // lane i computed keypoint i's angle
            kp.response = (float)score_l;  // and loaded its slot's score
            kp.octave = rl_lv & 255;
            kp.class_id = -1;
            ((KeyPointPOD *)buf.kps)[(size_t)img * cfg.sel_total + (rl_lv >> 8)] = kp;
        }
    }
}

void orbfe_launch_describe(const DeviceConfig &cfg_in, const DeviceBuffers &buf, int n_images, bool stereo, hipStream_t s)
{
    DeviceConfig cfg = cfg_in;
    cfg.xcd_magic = xcd_map_magic_host((cfg.sel_total + 4 * DS_KPW - 1) / (4 * DS_KPW), n_images);
    dim3 grid(xcd_grid((cfg.sel_total + 4 * DS_KPW - 1) / (4 * DS_KPW), n_images));
    if (cfg.half_patch == 15) { // the reference's HALF_PATCH_SIZE
        size_t lds = 256 * sizeof(float4) + 4 * (DS_BLR_ROWS * DS_PATCH_W);
        int rl_blocks = 0; // workgroups that build the stereo row lists, ahead of the descriptor ones
        if (stereo) {
            rl_blocks = (((n_images / 2) * rowlist_blocks(cfg.height, RL_ROWS_FUSED) + 3) / 4 + 7) & ~7;
            if (lds < 4 * (size_t)RL_LDS_BYTES) lds = 4 * (size_t)RL_LDS_BYTES;
        }
        dim3 grid_s(rl_blocks + grid.x);
        hipLaunchKernelGGL(describe_kernel, grid_s, dim3(256), lds, s, cfg, buf, n_images, rl_blocks ORBFE_CUT_ARG("ORBFE_DESC_DBG"));
        return;
    }
    const size_t raw_bytes = (((2 * cfg.half_patch + 1) * DS_PATCH_W + 15) & ~15);
    const size_t lds = ((cfg.patch_n * 2 + 15) & ~15) + 256 * 4 + 4 * (raw_bytes + DS_BLR_ROWS * DS_PATCH_W);
    hipLaunchKernelGGL(describe_generic_kernel, grid, dim3(256), lds, s, cfg, buf, n_images, stereo ? 1 : 0 ORBFE_CUT_ARG("ORBFE_DESC_DBG"));
}

#ifdef ORBFE_PROFILE_CUTS
// EXPERIMENT (cut-point build only, ORBFE_DBG_SORT_SEL=1|2): re-order every (image, level)'s selected keypoints spatially before
// describe_kernel runs, to measure what the kernel would gain from keypoints that share cache lines being processed together.
// The OUTPUT ORDER is then not the reference's: timing only.  (It led to DeviceConfig::proc_order; with that on, describe_kernel walks proc_xy
// and this kernel changes nothing for it: combine with ORBFE_NO_PROC_ORDER=1.)  1: 32-px-tile rows, then x; 2: Morton order of 16-px cells.
__global__ __launch_bounds__(256) void dbg_sort_sel_kernel(DeviceConfig cfg, DeviceBuffers buf, int mode)
{
    __shared__ uint32_t s_key[1024], s_xy[1024];
    __shared__ uint8_t s_sc[1024];
    const int img = blockIdx.x, level = blockIdx.y, tid = threadIdx.x;
    const LevelInfo &L = cfg.lv[level];
    int n = buf.sel_cnt[(size_t)img * cfg.nlevels + level];
    n = n > 1024 ? 0 : n;
    uint32_t *xy = buf.sel_xy + (size_t)img * cfg.sel_total + L.sel_off;
    uint8_t *sc = buf.sel_sc + (size_t)img * cfg.sel_total + L.sel_off;
    for (int i = tid; i < n; i += 256) {
        const uint32_t v = xy[i];
        const unsigned x = v & 0xffffu, y = v >> 16;
        uint32_t key;
        if (mode == 1) key = ((y >> 5) << 20) | (x << 8) | (y & 31u);
        else if (mode == 3) key = ((y >> 5) << 20) | (unsigned)i;              // tile row only, list order inside (a stable 12-bin sort)
        else if (mode == 4) key = ((y >> 4) << 20) | (x << 8) | (y & 15u);
        else if (mode == 5) key = ((y >> 6) << 20) | (x << 8) | (y & 63u);
        else if (mode == 6) key = ((x >> 5) << 20) | (y << 8) | (x & 31u);
        else if (mode == 7) key = ((y >> 5) << 20) | ((x >> 6) << 12) | (unsigned)i; // 32 x 64 px cells in row-major order, list order inside
        else {
            unsigned m = 0;
            for (int b = 0; b < 8; b++) m |= (((x >> (4 + b)) & 1u) << (2 * b)) | (((y >> (4 + b)) & 1u) << (2 * b + 1));
            key = (m << 12) | ((y & 15u) << 4) | (x & 15u);
        }
        s_key[i] = key; s_xy[i] = v; s_sc[i] = sc[i];
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const uint32_t k = s_key[i];
        int r = 0;
        for (int j = 0; j < n; j++) r += (s_key[j] < k || (s_key[j] == k && j < i)) ? 1 : 0;
        xy[r] = s_xy[i]; sc[r] = s_sc[i];
    }
}
void orbfe_launch_dbg_sort_sel(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, int mode, hipStream_t s)
{
    hipLaunchKernelGGL(dbg_sort_sel_kernel, dim3(n_images, cfg.nlevels), dim3(256), 0, s, cfg, buf, mode);
}
#endif
