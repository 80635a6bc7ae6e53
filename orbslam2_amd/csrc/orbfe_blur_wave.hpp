// orbfe_blur_wave.hpp -- the wave-level body of the 7x7 Gaussian (src/ORBextractor.cc:900), shared by orbfe_pyramid.hip (its own launch and the
// launches that also resize) and orbfe_octree3.hip (the levels that are still unblurred ride in the quadtree launch).  Device code only.
#pragma once
#include "orbfe_common.hpp"

// ---------------------------------------------------------------------------
// Gaussian 7x7 (8.8 fixed point, separable), all levels in one launch.
// Register sliding window: a lane owns 4 adjacent columns and walks down BL_ROWS rows; per input
// row it loads three aligned words (12 px), forms the four 7-tap row sums with v_dot4_u32_u8 against
// shifted tap words, keeps the row sums of the last eight rows as four row pairs and emits two 4-px
// output words per two input rows with v_dot2_u32_u16.  No LDS, no barriers; HBM traffic = one read
// of the level (+6/BL_ROWS row halo, L2-served) and one write.  Round 2: 872 -> 630 VALU instructions
// per wave (13.6 -> 9.8 per pixel), 0.109 -> 0.098 ms (the first 12 % of the cut bought all of that:
// the kernel then waits for memory), and whole-line 16-byte stores (tiles of 4 x 4 px blocks): -> 0.083 ms.
// ---------------------------------------------------------------------------
#define BL_ROWS ORBFE_BLUR_ROWS // rows per wave (a multiple of 4): 6 / BL_ROWS of the rows are loaded (and row-filtered) twice; 32 beats 16 by 4 us now that the kernel is memory-bound (no difference while it was issue-bound); 48 / 64: + 3 / + 9 us (too few waves)
#define BL_COLS 256 // per wave: 64 lanes x 4 px
// tile u of the flattened (level, row band, 256-column strip) list, by one wave
__device__ __forceinline__ void blur_wave(const DeviceConfig &cfg, const DeviceBuffers &buf, int img, int u)
{
    const int lane = threadIdx.x & 63;
    const uint32_t ti = buf.blur_tile_info[u]; // host-built: saves the per-wave level search (a chain of dependent scalar loads)
    const LevelInfo &L = cfg.lv[ti & 0xffu];
    const int x0 = (int)((ti >> 8) & 0xffu) * BL_COLS + lane * 4;
    const int r0 = (int)(ti >> 16);
    if (x0 >= L.w || r0 >= L.h) return;
    const uint8_t *src = buf.pyr + (size_t)img * cfg.pyr_bytes + L.pyr_off + x0;
    // output in 32 x 4 px tiles of 128 B (describe_kernel's 37-row patches then touch about half as many cache lines), a tile
    // being eight 4 x 4 px blocks of 16 B: the lane's four columns of four rows are ONE 16-byte store and a wave's store
    // instruction fills eight whole lines (one dword per row in row-major tiles meant 32-B pieces of eight lines per store:
    // 0.097 -> 0.084 ms)
    uint8_t *dst = buf.blur + (size_t)img * cfg.blur_bytes + L.blur_off + (((unsigned)(r0 >> 2) * L.blur_tx + (x0 >> 5)) << 7) + ((x0 & 31) << 2);
    const unsigned tile_row_bytes = (unsigned)L.blur_tx << 7;
    unsigned tw[4][3]; // tw[j][q]: taps against the bytes of word q for pixel j; byte 4 q + b meets tap 4 q + b - 1 - j
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int q = 0; q < 3; q++) {
            unsigned w = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int t = 4 * q + b - 1 - j;
                if (t >= 0 && t <= 6) w |= (unsigned)cfg.taps[t] << (8 * b);
            }
            tw[j][q] = w;
        }
    const int y_max = L.h + PYR_MY - 1; // last materialised row
    // column pass on row PAIRS starting at even window rows: Q[m % 4][j] = H_2m | H_(2m+1) << 16 (row sums fit 16 bits:
    // <= 255 * 256).  Output row 2a is k0 k1 | k2 k3 | k4 k5 | k6 0 against Q[a .. a+3], output row 2a + 1 is
    // 0 k0 | k1 k2 | k3 k4 | k5 k6 against the same four pairs: four v_dot2_u32_u16 per pixel either way, and only one pair is
    // formed per two rows (pairs at every row start, which the three-dot2-plus-mad form needs, cost twice the packing ops).
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const unsigned short k0 = (unsigned short)cfg.taps[0], k1 = (unsigned short)cfg.taps[1], k2 = (unsigned short)cfg.taps[2], k3 = (unsigned short)cfg.taps[3];
    const unsigned short k4 = (unsigned short)cfg.taps[4], k5 = (unsigned short)cfg.taps[5], k6 = (unsigned short)cfg.taps[6];
    const u16x2 te[4] = {{k0, k1}, {k2, k3}, {k4, k5}, {k6, 0}}, to[4] = {{0, k0}, {k1, k2}, {k3, k4}, {k5, k6}};
    unsigned Q[4][4], he[4] = {0, 0, 0, 0};
    uint4 og = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < BL_ROWS + 6; i++) {
        int y = r0 - 3 + i;
        y = y > y_max ? y_max : y; // rows past the margin only feed outputs that are never stored
        const uint32_t *row = (const uint32_t *)(src + __mul24(y, L.pitch)); // y >= -3: the margin rows above the image
        const unsigned w0 = row[-1], w1 = row[0], w2 = row[1];
        // row pass: pixel j of the lane's word is byte 4 + j of (w0, w1, w2) and its seven taps cover bytes 1 + j .. 7 + j, so
        // H_j is a byte dot product of the three aligned words with tap words shifted by j (wave-uniform, in scalar registers):
        // 2 + 3 + 3 + 2 v_dot4_u32_u8 per four pixels and no byte alignment ops (aligning the pixels instead costs 6 + 8)
        unsigned hn[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            unsigned acc = __builtin_amdgcn_udot4(w1, tw[j][1], 0u, false);
            if (j < 3) acc = __builtin_amdgcn_udot4(w0, tw[j][0], acc, false);
            if (j > 0) acc = __builtin_amdgcn_udot4(w2, tw[j][2], acc, false);
            hn[j] = acc;
        }
        if (!(i & 1)) {
#pragma unroll
            for (int j = 0; j < 4; j++) he[j] = hn[j];
            continue;
        }
        const int m = i >> 1;
#pragma unroll
        for (int j = 0; j < 4; j++) Q[m & 3][j] = he[j] | (hn[j] << 16);
        if (m < 3) continue;
        const int a = m - 3;
        unsigned o2[2];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            unsigned ob[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                unsigned acc = 32768u;
#pragma unroll
                for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, Q[(a + q) & 3][j]), half ? to[q] : te[q], acc, false);
                ob[j] = acc;
            }
            // byte 2 of the four sums -> one word: two v_perm_b32 and an or
            o2[half] = __builtin_amdgcn_perm(ob[1], ob[0], 0x0c0c0602u) | __builtin_amdgcn_perm(ob[3], ob[2], 0x06020c0cu);
        }
        if (!(a & 1)) { og.x = o2[0]; og.y = o2[1]; continue; }
        og.z = o2[0]; og.w = o2[1];
        // rows 2a - 2 .. 2a + 1 = one row of tiles; rows past the level in the last one are allocated and never read
        if (r0 + 2 * a - 2 < L.h) *(uint4 *)(dst + (unsigned)(a >> 1) * tile_row_bytes) = og;
    }
}
