// orbfe_stereo.hip -- Frame::ComputeStereoMatches (src/Frame.cc:464-642), ComputeStereoFromRGBD (:645-666), batched DescriptorDistance.
#include "orbfe_common.hpp"


// ---------------------------------------------------------------------------
// stereo: one wave per left keypoint (coarse Hamming band search + SAD + parabola)
// ---------------------------------------------------------------------------
// One 16-lane group per left keypoint (four keypoints per wave, sixteen per workgroup): a row list holds a few
// dozen candidates of which ~10 pass the octave / disparity filter, so a whole wave per keypoint idles most lanes
// and, with ~7 dependent global round trips per keypoint, needs 4x the waves to hide the same latency.
//   coarse search: lanes stride the row list; arg-min key dist << 16 | iR (= the reference's first minimum);
//   SAD: the 11 x 11 left window and the 11 x 21 right band go through LDS (three 128-bit loads by 11 lanes); lane handles window
//        pixels p = gl, gl + 16, .. < 121; the 11 shifted right-image bytes of a pixel are 12 contiguous bytes of its band row;
//        sums reduced over the group by xor shuffles.
#define SM_G 16
#define SM_WIN_BYTES (11 * 16 + 11 * 32 + 16) // left window rows (16 B each), right band rows (32 B each), + 16: a row's shifted read may touch the next word
__device__ __forceinline__ unsigned group_min_u32(unsigned v)
{
#pragma unroll
    for (int o = SM_G / 2; o > 0; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)v, o, 64);
        v = t < v ? t : v;
    }
    return v;
}
__device__ __forceinline__ int group_sum_i32(int v) { return row_sum_i32(v); } // SM_G == 16 == one DPP row
__device__ __forceinline__ uint32_t sad_u16(uint32_t a, uint32_t b, uint32_t c) { uint32_t d; asm("v_sad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

#include "orbfe_rowlist.hpp"

// four independent waves per workgroup: wave w of workgroup b builds block 4 b + w of pair blockIdx.y
__global__ __launch_bounds__(256) void stereo_rowlist_kernel(DeviceConfig cfg, DeviceBuffers buf)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_rl_all[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    rowlist_wave<RL_ROWS_ALONE>(cfg, buf, blockIdx.y, (int)blockIdx.x * 4 + wave, s_rl_all + wave * RL_LDS_BYTES);
}

__global__ __launch_bounds__(256) void stereo_match_kernel(DeviceConfig cfg, DeviceBuffers buf, int n_pairs)
{
    // XCD-aware block -> (pair, block) map: all blocks of a pair on one XCD (its L2 then holds the pair's
    // descriptors, keypoints and the pyramid rows the SAD windows touch)
    __shared__ uint2 s_cand[(256 / SM_G) * 4 * SM_G]; // per 16-lane group: the candidates of a 64-entry chunk that passed the filter
    __shared__ __attribute__((aligned(16))) uint8_t s_win[(256 / SM_G) * SM_WIN_BYTES]; // per group: the SAD windows
    const int kpb = 256 / SM_G;
    const int bpp = (cfg.sel_total + kpb - 1) / kpb;
    int pair, blk;
    if (!xcd_map_magic(bpp, n_pairs, cfg.xcd_magic, pair, blk)) return;
    const int imgL = 2 * pair, imgR = 2 * pair + 1;
    const int gl = threadIdx.x & (SM_G - 1);
    const int iL = blk * kpb + (threadIdx.x / SM_G);
    const int nL = buf.kp_cnt[imgL], nR = buf.kp_cnt[imgR];
    if (iL >= nL) return; // whole group; the groups of a wave only meet in xor shuffles below the group size
    const KeyPointPOD *kL = (const KeyPointPOD *)buf.kps + (size_t)imgL * cfg.sel_total;
    const KeyPointPOD *kR = (const KeyPointPOD *)buf.kps + (size_t)imgR * cfg.sel_total;
    const uint8_t *dL = buf.desc + (size_t)imgL * cfg.sel_total * 32;
    const uint8_t *dR = buf.desc + (size_t)imgR * cfg.sel_total * 32;
    float *u_right = buf.u_right + (size_t)imgL * cfg.sel_total;
    float *depth = buf.depth + (size_t)imgL * cfg.sel_total;
    int *sad_out = buf.sad + (size_t)imgL * cfg.sel_total;

    const KeyPointPOD kp = kL[iL];
    const int level_l = kp.octave;
    const float uL = kp.x, vL = kp.y;
    const int row = (int)vL;
    const float min_z = cfg.mb;
    const float max_d = __fdiv_rn(cfg.bf, min_z);
    const float min_u = __fsub_rn(uL, max_d);
    const float max_u = uL; // uL - minD, minD = 0

    uint32_t dl[8];
    {
        const uint4 *p = (const uint4 *)(dL + (size_t)iL * 32);
        const uint4 lo = p[0], hi = p[1];
        dl[0] = lo.x; dl[1] = lo.y; dl[2] = lo.z; dl[3] = lo.w; dl[4] = hi.x; dl[5] = hi.y; dl[6] = hi.z; dl[7] = hi.w;
    }
    unsigned best = (100u << 16) | 0xffffu; // TH_HIGH; the index field only matters below it
    float best_x = 0.f;
    auto consider = [&](int iR, int oct, float xr) {
        if (oct >= level_l - 1 && oct <= level_l + 1 && xr >= min_u && xr <= max_u) {
            const uint4 *p = (const uint4 *)(dR + (size_t)iR * 32);
            const uint4 lo = p[0], hi = p[1];
            const uint32_t dr[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const unsigned key = ((unsigned)hamming256(dl, dr) << 16) | (unsigned)iR;
            if (key < best) { best = key; best_x = xr; }
        }
    };
    // candidates = right keypoints whose row band covers int(vL) (vRowIndices[vL], src/Frame.cc:513), listed per row by the
    // row-list waves (orbfe_rowlist.hpp); the arg-min key (dist << 16 | iR) makes the result independent of the order inside a row list
    int cnt = 0;
    if (row >= 0 && row < cfg.height) cnt = buf.row_cnt[(size_t)pair * cfg.height + row];
    if (cnt <= cfg.row_cap) {
        // Two rounds of loads instead of two per candidate: (1) every lane fetches up to four list entries at once and applies
        // the octave / disparity filter (about a fifth pass); (2) the survivors are packed into a per-group LDS list through a
        // ballot, so that each lane then fetches ONE survivor's descriptor -- all in flight together.  A row list of ~50
        // entries used to cost four dependent entry -> descriptor round trips per lane.
        const uint2 *rent = buf.row_ent + ((size_t)pair * cfg.height + row) * cfg.row_cap;
        uint2 *s_list = s_cand + (threadIdx.x / SM_G) * (4 * SM_G);
        const int gshift = (threadIdx.x & 63) & ~(SM_G - 1); // first lane of this group inside its wave
        for (int j0 = 0; j0 < cnt; j0 += 4 * SM_G) {
            uint2 e[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = j0 + gl + SM_G * u;
                e[u] = rent[j < cnt ? j : cnt - 1];
            }
            int n_pass = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = j0 + gl + SM_G * u;
                const int oct = (int)(e[u].x >> 16);
                const float xr = __uint_as_float(e[u].y);
                ok[u] = j < cnt && oct >= level_l - 1 && oct <= level_l + 1 && xr >= min_u && xr <= max_u;
                const unsigned slice = (unsigned)(__ballot(ok[u]) >> gshift) & ((1u << SM_G) - 1u);
                if (ok[u]) s_list[n_pass + __popc(slice & ((1u << gl) - 1u))] = e[u];
                n_pass += __popc(slice);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f); // this wave's LDS writes have landed (the groups of a wave run in lockstep)
            __builtin_amdgcn_wave_barrier();
            for (int k = gl; k < n_pass; k += SM_G) {
                const uint2 c = s_list[k];
                const int iR = (int)(c.x & 0xffffu);
                const uint4 *p = (const uint4 *)(dR + (size_t)iR * 32);
                const uint4 lo = p[0], hi = p[1];
                const uint32_t dr[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                const unsigned key = ((unsigned)hamming256(dl, dr) << 16) | (unsigned)iR;
                if (key < best) { best = key; best_x = __uint_as_float(c.y); }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier(); // the list is rewritten by the next chunk
        }
    } else { // the row's list overflowed its capacity: test every right keypoint's band
        for (int iR = gl; iR < nR; iR += SM_G) {
            const KeyPointPOD kr = kR[iR];
            const float r = __fmul_rn(2.0f, cfg.lv[kr.octave].scale);
            const int maxr = (int)ceilf(__fadd_rn(kr.y, r));
            const int minr = (int)floorf(__fsub_rn(kr.y, r));
            if (row >= minr && row <= maxr) consider(iR, kr.octave, kr.x);
        }
    }
    const unsigned gbest = group_min_u32(best);
    const int best_dist = (int)(gbest >> 16);
    // x of the winning candidate: held by the lane whose key won (keys are unique per iR)
    float uR0 = best == gbest ? best_x : 0.f;
    {
        int bits = __float_as_int(uR0);
#pragma unroll
        for (int o = SM_G / 2; o > 0; o >>= 1) bits |= __shfl_xor(bits, o, 64); // one lane holds it, the others 0
        uR0 = __int_as_float(bits);
    }
    float out_u = -1.0f, out_d = -1.0f;
    int out_sad = -1;
    if (best_dist < 75) { // (TH_HIGH + TH_LOW) / 2
        const float sf = cfg.lv[level_l].inv_scale;
        const float s_uL = roundf(__fmul_rn(kp.x, sf));
        const float s_vL = roundf(__fmul_rn(kp.y, sf));
        const float s_uR0 = roundf(__fmul_rn(uR0, sf));
        const LevelInfo &L = cfg.lv[level_l];
        const int cu = (int)s_uL, cv = (int)s_vL, cr = (int)s_uR0;
        const float iniu = s_uR0;                        // scaleduR0 + L - w
        const float endu = __fadd_rn(s_uR0, 11.0f);      // scaleduR0 + L + w + 1
        const bool in_ref = !(iniu < 0 || endu >= (float)L.w);
        // the reference would throw on a window outside the level image; unreachable for
        // keypoints >= 19 px from the border, kept as a memory-safety guard
        const bool safe = cu - 5 >= 0 && cu + 5 < L.w && cv - 5 >= 0 && cv + 5 < L.h && cr - 10 >= 0 && cr + 10 < L.w;
        if (in_ref && safe) {
            int lpitch;
            const uint8_t *imL = level_image(cfg, buf, imgL, level_l, lpitch);
            const uint8_t *imR = level_image(cfg, buf, imgR, level_l, lpitch);
            // the group's windows through LDS: lane r < 11 fetches row r of the left window (11 bytes from column cu - 5: one
            // unaligned 128-bit load) and of the right band (21 bytes from column cr - 10: two), instead of every lane fetching a
            // byte and a 12-byte piece for each of its 8 window pixels (18 load instructions per lane; the texture addresser was
            // busy 62 % of this kernel).  The extra bytes (up to column cu + 10 / cr + 21) lie in the row's right margin
            // (level 0 read in place: in the first bytes of the next row; the window's last row is >= 14 rows above the image's).
            uint8_t *wl = s_win + (threadIdx.x / SM_G) * SM_WIN_BYTES, *wr = wl + 11 * 16;
            if (gl < 11) {
                const unsigned ro = (unsigned)__mul24(cv - 5 + gl, lpitch);
                const uint4 a = load16_unaligned(imL + ro + cu - 5);
                const uint4 b0 = load16_unaligned(imR + ro + cr - 10), b1 = load16_unaligned(imR + ro + cr + 6);
                *(uint4 *)(wl + 16 * gl) = a;
                *(uint4 *)(wr + 32 * gl) = b0;
                *(uint4 *)(wr + 32 * gl + 16) = b1;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f); // this wave's LDS writes have landed (the lanes of a group run in lockstep)
            __builtin_amdgcn_wave_barrier();
            // 12 bytes from byte o of a right-band row: four aligned words, shifted
            auto window = [&](int row, int o, uint32_t (&w)[3]) {
                const uint32_t *q = (const uint32_t *)(wr + 32 * row + (o & ~3));
                const uint32_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                const unsigned sh = (unsigned)(o & 3);
                w[0] = __builtin_amdgcn_alignbyte(q1, q0, sh); w[1] = __builtin_amdgcn_alignbyte(q2, q1, sh); w[2] = __builtin_amdgcn_alignbyte(q3, q2, sh);
            };
            const int lc = wl[16 * 5 + 5];
            // centre row of the right image: bytes cr-5 .. cr+5 (+1 spare) = rc of the 11 shifts
            uint32_t rcw[3];
            window(5, 5, rcw);
            // |(IL - lc) - (IR_i - rc_i)| = |(IL + rc_i) - (IR_i + lc)| for the 11 shifts i: both sides are in [0, 510], so TWO window
            // pixels per register (16-bit halves, plain 32-bit adds: no carry crosses) and one v_sad_u16 per shift and pixel pair,
            // which also accumulates: 4 ops per pair and shift (byte pair by v_perm_b32, two adds, the sad) instead of 12.  A
            // lane's sum over its 8 window pixels is below 4088.
            const uint32_t lc2 = (uint32_t)lc * 0x10001u;
            uint32_t rc2[11], acc[11];
#pragma unroll
            for (int i = 0; i < 11; i++) {
                const uint32_t word = rcw[i >> 2];
                const uint32_t sel = (i & 3) == 0 ? 0x0c000c00u : ((i & 3) == 1 ? 0x0c010c01u : ((i & 3) == 2 ? 0x0c020c02u : 0x0c030c03u));
                rc2[i] = __builtin_amdgcn_perm(0u, word, sel); // rc_i in both halves
                acc[i] = 0u;
            }
#pragma unroll
            for (int t = 0; t < 8; t += 2) {
                const int p0 = gl + SM_G * t, p1 = p0 + SM_G;
                const int py0 = (p0 * 745) >> 13, px0 = p0 - py0 * 11; // p / 11 for p < 128
                const int p1c = p1 < 121 ? p1 : p0;                    // only t = 6 of lanes 9 .. 15: masked out below
                const int py1 = (p1c * 745) >> 13, px1 = p1c - py1 * 11;
                const uint32_t il2 = (uint32_t)wl[16 * py0 + px0] | ((uint32_t)wl[16 * py1 + px1] << 16);
                uint32_t w0[3], w1[3];
                window(py0, px0, w0); // right bytes cr + dx - 5 .. cr + dx + 6, dx = px - 5
                window(py1, px1, w1);
                const uint32_t m = p1 < 121 ? 0xffffffffu : 0x0000ffffu;
#pragma unroll
                for (int i = 0; i < 11; i++) {
                    const uint32_t sel = (i & 3) == 0 ? 0x0c040c00u : ((i & 3) == 1 ? 0x0c050c01u : ((i & 3) == 2 ? 0x0c060c02u : 0x0c070c03u));
                    const uint32_t ir2 = __builtin_amdgcn_perm(w1[i >> 2], w0[i >> 2], sel); // byte i of pixel t | byte i of pixel t + 1 << 16
                    uint32_t x2 = il2 + rc2[i], y2 = ir2 + lc2;
                    if (t == 6) { x2 &= m; y2 &= m; }
                    acc[i] = sad_u16(x2, y2, acc[i]);
                }
            }
            int dists[11];
#pragma unroll
            for (int i = 0; i < 11; i++) dists[i] = (int)acc[i];
            int sad_best = 0x7fffffff, best_inc = 0;
#pragma unroll
            for (int i = 0; i < 11; i++) {
                dists[i] = group_sum_i32(dists[i]);
                if (dists[i] < sad_best) { sad_best = dists[i]; best_inc = i - 5; }
            }
            out_sad = -2 - sad_best; // coarse match without an accepted disparity (debug tap): negative
            if (best_inc != -5 && best_inc != 5) {
                float d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
                for (int t = 1; t < 10; t++)
                    if (t == best_inc + 5) { d1 = (float)dists[t - 1]; d2 = (float)dists[t]; d3 = (float)dists[t + 1]; }
                const float delta = __fdiv_rn(__fsub_rn(d1, d3), __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2))));
                if (!(delta < -1.0f || delta > 1.0f)) {
                    float best_ur = __fmul_rn(L.scale, __fadd_rn(__fadd_rn(s_uR0, (float)best_inc), delta));
                    float disparity = __fsub_rn(uL, best_ur);
                    if (disparity >= 0.0f && disparity < max_d) {
                        if (disparity <= 0.0f) {
                            disparity = 0.01f;
                            best_ur = (float)__dsub_rn((double)uL, 0.01);
                        }
                        out_d = __fdiv_rn(cfg.bf, disparity);
                        out_u = best_ur;
                        out_sad = sad_best;
                    }
                }
            }
        }
    }
    if (gl == 0) {
        u_right[iL] = out_u;
        depth[iL] = out_d;
        sad_out[iL] = out_sad;
    }
}

// Median of the accepted SADs, then cut at 1.5*1.4*median (src/Frame.cc:628-641; Q2: skip when empty).
// The reference sorts (SAD, iL) pairs and reads element size/2; only its SAD matters, so the median is
// found by a 2-level radix select (8 bits per level, LDS histograms; a SAD fits 16 bits) instead of a sort.
__global__ __launch_bounds__(256) void stereo_median_kernel(DeviceConfig cfg, DeviceBuffers buf)
{
    extern __shared__ int s_vals[]; // [sel_total] the pair's SADs, read from HBM once
    __shared__ int s_hist[256];
    __shared__ int s_sel[3]; // selected digit, rank inside the digit's bucket, count of valid entries
    const int pair = blockIdx.x;
    const int imgL = 2 * pair;
    const int tid = threadIdx.x, lane = tid & 63;
    const int nL = buf.kp_cnt[imgL];
    float *u_right = buf.u_right + (size_t)imgL * cfg.sel_total;
    float *depth = buf.depth + (size_t)imgL * cfg.sel_total;
    const int *sad = buf.sad + (size_t)imgL * cfg.sel_total;
    for (int i = tid; i < nL; i += 256) s_vals[i] = sad[i];
    unsigned prefix = 0, mask = 0;
    for (int shift = 8; shift >= 0; shift -= 8) { // SAD <= 121 px * 510 = 61 710 < 2^16: two 8-bit digits
        s_hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < nL; i += 256) {
            const int v = s_vals[i];
            if (v >= 0 && ((unsigned)v & mask) == prefix) atomicAdd(&s_hist[((unsigned)v >> shift) & 255u], 1);
        }
        __syncthreads();
        if (tid < 64) { // digit that holds the wanted rank: lane = 4 bins, wave scan, owner lane resolves its bins
            const int h0 = s_hist[4 * lane], h1 = s_hist[4 * lane + 1], h2 = s_hist[4 * lane + 2], h3 = s_hist[4 * lane + 3];
            const int sum = h0 + h1 + h2 + h3;
            int inc = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(inc, o, 64);
                if (lane >= o) inc += t;
            }
            const int total = __shfl(inc, 63, 64);
            if (shift == 8 && lane == 0) s_sel[2] = total;
            // vDistIdx[size/2] in ascending order; later digits continue with the rank left inside the chosen bucket
            const int rank = shift == 8 ? total / 2 : s_sel[1];
            const int before = inc - sum;
            if (total > 0 && rank >= before && rank < inc) {
                int r = rank - before, d = 4 * lane;
                if (r >= h0) { r -= h0; d++; if (r >= h1) { r -= h1; d++; if (r >= h2) { r -= h2; d++; } } }
                s_sel[0] = d;
                s_sel[1] = r;
            }
        }
        __syncthreads();
        if (s_sel[2] == 0) return;
        prefix |= (unsigned)s_sel[0] << shift;
        mask |= 255u << shift;
        __syncthreads();
    }
    const float median = (float)(int)prefix;
    const float th_dist = __fmul_rn(__fmul_rn(1.5f, 1.4f), median);
    for (int i = tid; i < nL; i += 256) {
        const int v = s_vals[i];
        if (v >= 0 && !((float)v < th_dist)) { u_right[i] = -1.0f; depth[i] = -1.0f; }
    }
}

// Frame::ComputeStereoFromRGBD (src/Frame.cc:645-666), undistorted camera
// cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK) of Frame::UndistortKeyPoints (src/Frame.cc:402-432):
// five fixed-point iterations in double (the overload's TermCriteria(MAX_ITER, 5, 0.01)), K and D widened from float,
// re-projection with P = K.  Plain IEEE double operations in the oracle's order: bit-exact.  OPENCV-4.5.5-SEMANTICS.
__device__ inline void undistort_point(const DeviceConfig &cfg, float uf, float vf, float &uo, float &vo)
{
    if (cfg.n_dist == 0 || cfg.dist[0] == 0.0f) { uo = uf; vo = vf; return; } // :404-408
    const double k0 = cfg.dist[0], k1 = cfg.dist[1], k2 = cfg.dist[2], k3 = cfg.dist[3], k4 = cfg.n_dist > 4 ? (double)cfg.dist[4] : 0.0;
    const double fx = cfg.cam[0], fy = cfg.cam[1], cx = cfg.cam[2], cy = cfg.cam[3], ifx = 1. / fx, ify = 1. / fy;
    const double u = uf, v = vf;
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0.0 * r2 + 0.0) * r2 + 0.0) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
        const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + 0.0 * r2 + 0.0 * r2 * r2;
        const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + 0.0 * r2 + 0.0 * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0.0 * y + cx, yy = 0.0 * x + fy * y + cy, ww = 1. / (0.0 * x + 0.0 * y + 1.0);
    uo = (float)(xx * ww);
    vo = (float)(yy * ww);
}

__global__ __launch_bounds__(256) void undistort_kernel(DeviceConfig cfg, const KeyPointPOD *__restrict__ in, KeyPointPOD *__restrict__ out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    KeyPointPOD kp = in[i];
    undistort_point(cfg, kp.x, kp.y, kp.x, kp.y);
    out[i] = kp;
}

void orbfe_launch_undistort(const DeviceConfig &cfg, const void *d_keys_in, void *d_keys_out, int n, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(undistort_kernel, dim3((n + 255) / 256), dim3(256), 0, s, cfg, (const KeyPointPOD *)d_keys_in, (KeyPointPOD *)d_keys_out, n);
}

// T = float: the CV_32F map Frame::Frame receives.  T = uint16_t: the sensor's raw map; the conversion of
// Tracking::GrabImageRGBD (src/Tracking.cc:323-324, convertTo(CV_32F, mDepthMapFactor): one rounded float multiply)
// is applied to the sampled pixel only.
template <typename T>
__global__ __launch_bounds__(256) void rgbd_kernel(DeviceConfig cfg, DeviceBuffers buf, const T *__restrict__ depth_img,
                                                   size_t pitch_floats, int img, float factor)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    img += (int)blockIdx.y;                                              // batched call (orbfe_enqueue_rgbd): one grid row per image,
    depth_img += (size_t)blockIdx.y * pitch_floats * (size_t)cfg.height; // the depth maps packed one after the other
    const int n = buf.kp_cnt[img];
    if (i >= n) return;
    const KeyPointPOD kp = ((const KeyPointPOD *)buf.kps)[(size_t)img * cfg.sel_total + i];
    float u = -1.0f, dp = -1.0f;
    const int v = (int)kp.y, uu = (int)kp.x;
    if (uu >= 0 && v >= 0 && uu < cfg.width && v < cfg.height) {
        float d;
        if constexpr (sizeof(T) == 2) d = __fmul_rn((float)depth_img[(size_t)v * pitch_floats + uu], factor);
        else d = depth_img[(size_t)v * pitch_floats + uu];
        if (d > 0) { // depth is read at the distorted keypoint, uRight is built from the undistorted x (src/Frame.cc:652-664)
            float xu, yu;
            undistort_point(cfg, kp.x, kp.y, xu, yu);
            dp = d;
            u = __fsub_rn(xu, __fdiv_rn(cfg.bf, d));
        }
    }
    buf.u_right[(size_t)img * cfg.sel_total + i] = u;
    buf.depth[(size_t)img * cfg.sel_total + i] = dp;
}

// all-pairs Hamming distance: block (64 b-columns) x (4 a-rows per block.y step)
__global__ __launch_bounds__(256) void hamming_matrix_kernel(const uint8_t *__restrict__ da, int na,
                                                             const uint8_t *__restrict__ db, int nb, int *__restrict__ dist)
{
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 16;
    if (j >= nb) return;
    uint32_t b[8];
    const uint32_t *pb = (const uint32_t *)(db + (size_t)j * 32);
#pragma unroll
    for (int k = 0; k < 8; k++) b[k] = pb[k];
    for (int i = i0; i < i0 + 16 && i < na; i++) {
        const uint32_t *pa = (const uint32_t *)(da + (size_t)i * 32);
        uint32_t a[8];
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = pa[k];
        dist[(size_t)i * nb + j] = hamming256(a, b);
    }
}


void orbfe_launch_stereo_rowlists(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_pairs, hipStream_t s)
{
    dim3 grid((rowlist_blocks(cfg.height, RL_ROWS_ALONE) + 3) / 4, n_pairs);
    hipLaunchKernelGGL(stereo_rowlist_kernel, grid, dim3(256), 4 * RL_LDS_BYTES, s, cfg, buf);
}

void orbfe_launch_stereo_match(const DeviceConfig &cfg_in, const DeviceBuffers &buf, int n_pairs, hipStream_t s)
{
    const int kpb = 256 / SM_G;
    DeviceConfig cfg = cfg_in;
    cfg.xcd_magic = xcd_map_magic_host((cfg.sel_total + kpb - 1) / kpb, n_pairs);
    dim3 grid(xcd_grid((cfg.sel_total + kpb - 1) / kpb, n_pairs));
    hipLaunchKernelGGL(stereo_match_kernel, grid, dim3(256), 0, s, cfg, buf, n_pairs);
}

void orbfe_launch_stereo_median(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_pairs, hipStream_t s)
{
    hipLaunchKernelGGL(stereo_median_kernel, dim3(n_pairs), dim3(256), (size_t)cfg.sel_total * sizeof(int), s, cfg, buf);
}

void orbfe_launch_rgbd(const DeviceConfig &cfg, const DeviceBuffers &buf, const float *d_depth, size_t depth_pitch_floats,
                       int image, hipStream_t s)
{
    hipLaunchKernelGGL(rgbd_kernel<float>, dim3((cfg.sel_total + 255) / 256), dim3(256), 0, s, cfg, buf, d_depth, depth_pitch_floats, image, 1.0f);
}

void orbfe_launch_rgbd_u16(const DeviceConfig &cfg, const DeviceBuffers &buf, const uint16_t *d_depth, size_t depth_pitch_px,
                           float factor, int image, hipStream_t s)
{
    hipLaunchKernelGGL(rgbd_kernel<uint16_t>, dim3((cfg.sel_total + 255) / 256), dim3(256), 0, s, cfg, buf, d_depth, depth_pitch_px, image, factor);
}

// n_images frames in one launch: depth map k (w x h elements, rows packed) belongs to image slot k
void orbfe_launch_rgbd_batch(const DeviceConfig &cfg, const DeviceBuffers &buf, const void *d_depth, bool is_u16, float factor, int n_images, hipStream_t s)
{
    dim3 grid((cfg.sel_total + 255) / 256, n_images);
    if (is_u16) hipLaunchKernelGGL(rgbd_kernel<uint16_t>, grid, dim3(256), 0, s, cfg, buf, (const uint16_t *)d_depth, (size_t)cfg.width, 0, factor);
    else hipLaunchKernelGGL(rgbd_kernel<float>, grid, dim3(256), 0, s, cfg, buf, (const float *)d_depth, (size_t)cfg.width, 0, 1.0f);
}

// ---------------------------------------------------------------------------
// Packed result block (orbfe_fetch_batch_packed, round 4).  A cv::KeyPoint is 28 bytes of which pt, size, octave and class_id are
// functions of (level x, level y, octave) -- src/ORBextractor.cc:838 (size = scaledPatchSize), :909-915 (pt *= scale) -- and the
// octave follows from the per-level counts, because a frame's keypoints are stored octave by octave (:866-917).  What has to cross
// the link per keypoint is x | y << 16 on its level (4 B), the angle (4 B) and the FAST score (1 B): 9 bytes instead of 28; the
// host expands with the reference's own float operations (orbfe_expand_packed).  One kernel gathers everything a step's results
// need -- these records, counts, descriptors, uRight / depth of the left images -- into one contiguous block: ONE device-to-host
// copy per step.  Out image o is device image slot o * img_step (img_step 2: left images only).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_results_kernel(DeviceConfig cfg, DeviceBuffers buf, uint8_t *__restrict__ out, PackedOffsets lay, int img_step, int stereo)
{
    const int oi = blockIdx.y, di = oi * img_step;
    const int cap = cfg.sel_total, j0 = (int)blockIdx.x * 256, j = j0 + (int)threadIdx.x;
    const int n = buf.kp_cnt[di];
    const int *lc = buf.sel_cnt + (size_t)di * cfg.nlevels;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ((int *)(out + lay.counts))[oi] = n;
        for (int l = 0; l < cfg.nlevels; l++) ((int *)(out + lay.level_counts))[oi * cfg.nlevels + l] = lc[l];
    }
    if (j < n) {
        int l = 0, first = 0; // the keypoint's octave: octave by octave, in the quadtree's list order (describe_kernel's `excl`)
        for (; l < cfg.nlevels - 1; l++) {
            const int c = lc[l];
            if (j < first + c) break;
            first += c;
        }
        const int slot = cfg.lv[l].sel_off + (j - first);
        const unsigned mb = (unsigned)cfg.min_border * 0x10001u;
        ((uint32_t *)(out + lay.xy))[(size_t)oi * cap + j] = buf.sel_xy[(size_t)di * cap + slot] + mb; // level-image coordinates (both halves < 32768: no carry)
        ((float *)(out + lay.angle))[(size_t)oi * cap + j] = ((const KeyPointPOD *)buf.kps)[(size_t)di * cap + j].angle;
        (out + lay.response)[(size_t)oi * cap + j] = buf.sel_sc[(size_t)di * cap + slot];
    }
    // descriptors: this block's 256 slots = 512 coalesced 16-byte chunks (slots past the count travel too: fixed layout)
    {
        const uint4 *src = (const uint4 *)(buf.desc + ((size_t)di * cap + j0) * 32);
        uint4 *dst = (uint4 *)(out + lay.desc + ((size_t)oi * cap + j0) * 32);
        const int chunks = 2 * (cap - j0 < 256 ? cap - j0 : 256);
        for (int c = threadIdx.x; c < chunks; c += 256) dst[c] = src[c];
    }
    if (stereo && (di & 1) == 0 && j < cap) { // uRight / depth of the left image of pair di / 2
        const int pr = di >> 1;
        ((float *)(out + lay.u_right))[(size_t)pr * cap + j] = buf.u_right[(size_t)di * cap + j];
        ((float *)(out + lay.depth))[(size_t)pr * cap + j] = buf.depth[(size_t)di * cap + j];
    }
}

void orbfe_launch_pack_results(const DeviceConfig &cfg, const DeviceBuffers &buf, uint8_t *d_out, const PackedOffsets &lay, int n_out, int img_step, bool stereo, hipStream_t s)
{
    dim3 grid((cfg.sel_total + 255) / 256, n_out);
    hipLaunchKernelGGL(pack_results_kernel, grid, dim3(256), 0, s, cfg, buf, d_out, lay, img_step, stereo ? 1 : 0);
}

void orbfe_launch_hamming_matrix(const uint8_t *da, int na, const uint8_t *db, int nb, int *dist, hipStream_t s)
{
    dim3 grid((nb + 63) / 64, (na + 63) / 64);
    hipLaunchKernelGGL(hamming_matrix_kernel, grid, dim3(256), 0, s, da, na, db, nb, dist);
}
