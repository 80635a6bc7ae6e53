// orbfe_rowlist.hpp -- the stereo row lists (vRowIndices, src/Frame.cc:474-491), built by independent waves; shared by
// orbfe_stereo.hip (a launch of its own: mono-sized patch geometries other than the reference's) and orbfe_describe.hip (the same
// waves riding in describe_kernel's launch).  Device code only.
#pragma once
#include "orbfe_common.hpp"

// Right keypoint iR is listed in rows floor(y - r) .. ceil(y + r), r = 2 * scale[octave].  One WAVE builds the lists of RL_ROWS
// consecutive image rows of a pair: the lists go to their place in runs of consecutive entries, the counts at the end.  It reads the keypoints as the
// quadtree kernel left them (one coalesced word per slot: level-local integer coordinates; final index, x and y follow exactly as
// describe_kernel derives them), all of an image's slots in flight at once; the per-level constants come from a 16-entry LDS table.
//   (1) every keypoint's band against the block's rows -- y + r > r0 - 1 and y - r < r1, the same rounded sums the reference takes
//       ceil / floor of -- and a ballot-ordered queue of the few per cent that meet them (every wave looks at all keypoints of the
//       image, so this pass is kept to ~16 instructions per slot);
//   (2) per queue chunk of 64 and block row: a ballot of the lanes whose band covers the row gives their list positions -- no
//       atomics, no divergent per-lane row loop.
// A row's count may exceed row_cap (entries beyond it are dropped): stereo_match_kernel then scans every right keypoint.
// History: rounds 1-2 appended from describe_kernel with one returning GLOBAL atomic and one scattered 8-byte store per (keypoint,
// row): ~1.1 M of each per 64-pair step, 24 us of describe_kernel (0.163 -> 0.139 ms without them) and most of its write
// amplification; issuing them at the start of the wave instead of its end changed nothing (their number, not their latency).  A
// first LDS kernel (one workgroup per 8-16 rows, LDS atomics, every slot's full arithmetic under a divergent test) took 20-29 us;
// these waves in a launch of their own 19.5 / 24.4 / 34.5 us at 4 / 8 / 16 rows per wave (7.8 of them launch + first-load
// latency), riding in describe_kernel's launch 8 / 5 / 4 us (describe 0.1466 / 0.1439 / 0.1425 ms against 0.1385 without lists; 32 rows: 0.146, its 32 counters spill;
// with the lists staged in LDS before a coalesced copy-out 0.1477: the runs a ballot writes are contiguous enough).
#define RL_ROWS_FUSED 16 // image rows per wave inside describe_kernel's launch (long waves are free there: only their issue slots count)
#define RL_ROWS_ALONE 4  // ... and in a launch of its own, which lives on the number of short waves
#define RL_QUEUE 128
#define RL_BATCH 4       // per lane and pass: 4 x 4 consecutive slots (one 128-bit load of coordinates, one 32-bit load of levels) in flight; 8 spills 13 VGPRs inside describe_kernel (64 allowed)
#define RL_LDS_BYTES (16 * 16 + RL_QUEUE * 16) // per wave: level constants, queue (entry.x, entry.y, y + r, y - r as bits)
__host__ __device__ __forceinline__ int rowlist_blocks(int height, int rows) { return (height + rows - 1) / rows; }

template <int ROWS>
__device__ __forceinline__ void rowlist_wave(const DeviceConfig &cfg, const DeviceBuffers &buf, int pair, int block, uint8_t *s_mem)
{
    int4 *s_lev = (int4 *)s_mem;                      // per level: first final index, keypoint count, first slot, scale bits
    uint4 *s_q = (uint4 *)(s_mem + 16 * 16);
    const int lane = threadIdx.x & 63;
    const int r0 = block * ROWS;
    if (r0 >= cfg.height) return;
    const int r1 = r0 + ROWS < cfg.height ? r0 + ROWS : cfg.height; // rows [r0, r1)
    const int imgR = 2 * pair + 1;
    const uint32_t *sxy = buf.sel_xy + (size_t)imgR * cfg.sel_total;
    const int *sel_cnt = buf.sel_cnt + (size_t)imgR * cfg.nlevels;
    const int cap = cfg.row_cap;
    int *rcnt = buf.row_cnt + (size_t)pair * cfg.height;
    uint2 *rent = buf.row_ent + ((size_t)pair * cfg.height + r0) * cap;
    const float lo_bound = (float)(r0 - 1), hi_bound = (float)r1;
    if (lane < 16) {
        int inc = lane < cfg.nlevels ? sel_cnt[lane] : 0;
        const int c = inc;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        int off = 0, sc = 0; // picked with a uniform level index: a per-lane index into the kernel arguments is a loop of dependent scalar loads
#pragma unroll
        for (int l = 0; l < ORBFE_MAX_LEVELS; l++)
            if (lane == l) { off = cfg.lv[l].sel_off; sc = __float_as_int(cfg.lv[l].scale); }
        s_lev[lane] = make_int4(inc - c, c, off, sc);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    int cnt[ROWS]; // list lengths (uniform)
#pragma unroll
    for (int k = 0; k < ROWS; k++) cnt[k] = 0;
    int nqd = 0; // queue fill (uniform)
    auto drain = [&]() {
        __builtin_amdgcn_s_waitcnt(0xc07f); // this wave's queue writes have landed
        __builtin_amdgcn_wave_barrier();
        for (int i0 = 0; i0 < nqd; i0 += 64) {
            const bool have = i0 + lane < nqd;
            const uint4 qe = s_q[have ? i0 + lane : 0];
            const uint2 e = make_uint2(qe.x, qe.y);
            const int maxr = (int)ceilf(__uint_as_float(qe.z)), minr = (int)floorf(__uint_as_float(qe.w));
#pragma unroll
            for (int k = 0; k < ROWS; k++) {
                const int row = r0 + k;
                const bool in = have && row < r1 && row >= minr && row <= maxr; // rows of the block only: that is the clamp of the band to the image
                const unsigned long long m = __ballot(in);
                if (m == 0ull) continue; // uniform
                const int p = cnt[k] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (in && p < cap) rent[(size_t)k * cap + p] = e; // the lanes of the ballot write consecutive entries of the row's list
                cnt[k] += __popcll(m);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        nqd = 0;
    };
    const int nq = (cfg.sel_total + 3) >> 2; // slot quads; sel_xy / slot_level are allocated with 4 spare entries
    for (int qb = 0; qb < nq; qb += RL_BATCH * 64) { // 1024 slots per pass
        uint4 xy[RL_BATCH];
        uint32_t lv4[RL_BATCH];
#pragma unroll
        for (int u = 0; u < RL_BATCH; u++) { // all in flight: the only global round trip of the scan
            const int q = qb + u * 64 + lane;
            xy[u] = q < nq ? load16_unaligned((const uint8_t *)(sxy + 4 * q)) : make_uint4(0u, 0u, 0u, 0u); // an image's slots start on a 4-byte boundary only
            lv4[u] = q < nq ? *(const uint32_t *)(buf.slot_level + 4 * q) : 0u;
        }
#pragma unroll
        for (int u = 0; u < RL_BATCH; u++) {
            const int q = qb + u * 64 + lane;
            const uint32_t xyv[4] = {xy[u].x, xy[u].y, xy[u].z, xy[u].w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int sl = 4 * q + j, lvl = (int)((lv4[u] >> (8 * j)) & 0xffu);
                const int4 lc = s_lev[lvl]; // first final index, keypoint count, first slot, scale bits
                const float scale = __int_as_float(lc.w);
                const int k = sl - lc.z;
                float y = (float)((int)(xyv[j] >> 16) + cfg.min_border);
                if (lvl != 0) y = __fmul_rn(y, scale);                      // describe_kernel's kp.y
                const float r = __fmul_rn(2.0f, scale);
                const float up = __fadd_rn(y, r), dn = __fsub_rn(y, r);     // the reference takes ceil / floor of these
                const bool pass = q < nq && k < lc.y && up > lo_bound && dn < hi_bound; // k < count: the slot holds a keypoint; ceil(up) >= r0, floor(dn) <= r1 - 1
                const unsigned long long m = __ballot(pass);
                if (m == 0ull) continue; // uniform
                if (nqd + 64 > RL_QUEUE) drain();
                if (pass) {
                    float x = (float)((int)(xyv[j] & 0xffffu) + cfg.min_border);
                    if (lvl != 0) x = __fmul_rn(x, scale);                  // kp.x
                    s_q[nqd + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] =
                        make_uint4((uint32_t)(k + lc.x) | ((uint32_t)lvl << 16), __float_as_uint(x), __float_as_uint(up), __float_as_uint(dn));
                }
                nqd += __popcll(m);
            }
        }
    }
    drain();
#pragma unroll
    for (int k = 0; k < ROWS; k++)
        if (lane == 0 && r0 + k < r1) rcnt[r0 + k] = cnt[k];
}
