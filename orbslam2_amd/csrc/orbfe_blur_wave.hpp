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
// tile u of the flattened (level, row band, 256-column strip) list, by one wave.  `fetch(y, w0, w1, w2)` delivers the three
// words of input row y around the lane's four columns (bytes x0 - 4 .. x0 + 7, reflect-101 outside the image).
template <class Fetch>
__device__ __forceinline__ void blur_wave_rows(const DeviceConfig &cfg, const LevelInfo &L, int r0, __amdgpu_buffer_rsrc_t dst, unsigned dst_off, Fetch fetch)
{
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
    typedef unsigned bw_v4 __attribute__((ext_vector_type(4)));
    const unsigned short k0 = (unsigned short)cfg.taps[0], k1 = (unsigned short)cfg.taps[1], k2 = (unsigned short)cfg.taps[2], k3 = (unsigned short)cfg.taps[3];
    const unsigned short k4 = (unsigned short)cfg.taps[4], k5 = (unsigned short)cfg.taps[5], k6 = (unsigned short)cfg.taps[6];
    const u16x2 te[4] = {{k0, k1}, {k2, k3}, {k4, k5}, {k6, 0}}, to[4] = {{0, k0}, {k1, k2}, {k3, k4}, {k5, k6}};
    unsigned Q[4][4], he[4] = {0, 0, 0, 0};
    uint4 og = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < BL_ROWS + 6; i++) {
        int y = r0 - 3 + i;
        y = y > y_max ? y_max : y; // rows past the margin only feed outputs that are never stored
        unsigned w0, w1, w2;
        fetch(y, w0, w1, w2); // y >= -3: the margin rows above the image
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
        if (r0 + 2 * a - 2 < L.h) __builtin_amdgcn_raw_buffer_store_b128((bw_v4){og.x, og.y, og.z, og.w}, dst, dst_off, (unsigned)(a >> 1) * tile_row_bytes, 0);
    }
}

__device__ __forceinline__ void blur_wave(const DeviceConfig &cfg, const DeviceBuffers &buf, int img, int u)
{
    const int lane = threadIdx.x & 63;
    const uint32_t ti = buf.blur_tile_info[u]; // host-built: saves the per-wave level search (a chain of dependent scalar loads)
    const int level = (int)(ti & 0xffu);
    const LevelInfo &L = cfg.lv[level];
    const int strip = (int)((ti >> 8) & 0xffu);
    const int x0 = strip * BL_COLS + lane * 4;
    const int r0 = (int)(ti >> 16);
    if (x0 >= L.w || r0 >= L.h) return;
    // output in 32 x 4 px tiles of 128 B (describe_kernel's 37-row patches then touch about half as many cache lines), a tile
    // being eight 4 x 4 px blocks of 16 B: the lane's four columns of four rows are ONE 16-byte store and a wave's store
    // instruction fills eight whole lines (one dword per row in row-major tiles meant 32-B pieces of eight lines per store:
    // 0.097 -> 0.084 ms)
    // Buffer addressing (round 5): descriptor (wave-uniform base) + scalar row offset + the lane's 32-bit offset: no 64-bit vector add per
    // row load / tile store (these waves ride in FAST's launch, where every vector instruction is paid for at FAST's issue rate).
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(buf.blur + (size_t)img * cfg.blur_bytes + L.blur_off + (((unsigned)(r0 >> 2) * L.blur_tx) << 7), 0,
                                                                       (BL_ROWS / 4) * (L.blur_tx << 7), ORBFE_RSRC_FLAGS);
    const unsigned dst_off = ((unsigned)(x0 >> 5) << 7) + (((unsigned)x0 & 31u) << 2);
    if (level == 0 && buf.lv0_packed) {
        // Level 0 read in place from the caller's packed image (round 4): rows at any alignment and no margin.  A lane's 12 bytes
        // come from ONE 128-bit load at the 4-byte boundary below them (byte-aligned loads cost the texture addresser about twice
        // an aligned one: the first version, an unaligned 96-bit load per lane, made this level's blur 50 % slower) and three
        // v_alignbyte_b32 by the ROW's misalignment, which is wave-uniform (the lane's columns start at a multiple of 4).  Rows
        // above / below the image are the reflected rows (a scalar index).  Columns: only lane 0 of the first strip (its left
        // neighbours are the reflection of its own pixels: one v_perm_b32 and three selects per row) and the last strip (a window
        // clamped into the row, bytes picked by per-lane selectors computed once per wave) differ from the interior.
        const uint8_t *simg = buf.lv0 + (size_t)img * buf.lv0_stride;
        const unsigned a0 = (unsigned)((uintptr_t)simg & 3u);
        const uint8_t *sbase = simg - a0; // 4-byte aligned; a window may start up to 3 bytes before the image, inside the word that holds its first pixel
        const int pitch = buf.lv0_pitch, w = L.w, h = L.h;
        auto row_off = [&](int y) { return (unsigned)__mul24(y < 0 ? -y : (y >= h ? 2 * h - 2 - y : y), pitch) + a0; }; // wave-uniform
        const bool last = (strip + 1) * BL_COLS + 8 > w; // wave-uniform: some lane's 12 bytes reach past the row's last pixel
        if (!last) {
            const bool lane0 = strip == 0 && lane == 0; // columns -4 .. -1 are columns 4 .. 1
            const unsigned xo = lane0 ? 0u : (unsigned)(x0 - 4);
            typedef unsigned bw_v4 __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t src0 = __builtin_amdgcn_make_buffer_rsrc((void *)sbase, 0, (unsigned)__mul24(h, pitch) + a0, ORBFE_RSRC_FLAGS); // every window of these strips ends inside its row
            blur_wave_rows(cfg, L, r0, dst, dst_off, [&](int y, unsigned &w0, unsigned &w1, unsigned &w2) {
                const unsigned ro = row_off(y), sh = ro & 3u;
                const bw_v4 q = __builtin_amdgcn_raw_buffer_load_b128(src0, xo, ro & ~3u, 0); // 4-byte aligned, not 16
                w0 = __builtin_amdgcn_alignbyte(q.y, q.x, sh); w1 = __builtin_amdgcn_alignbyte(q.z, q.y, sh); w2 = __builtin_amdgcn_alignbyte(q.w, q.z, sh);
                if (strip == 0) { // uniform
                    const unsigned t = __builtin_amdgcn_perm(w1, w0, 0x01020304u);
                    w2 = lane0 ? w1 : w2; w1 = lane0 ? w0 : w1; w0 = lane0 ? t : w0;
                }
            });
            return;
        }
        int xs = x0 - 4;
        xs = xs > w - 12 ? w - 12 : xs;
        xs = xs < 0 ? 0 : xs;
        unsigned selA[3], selB[3]; // word q = perm(A1, A0, selA[q]) | perm(0, A2, selB[q]) of the window's words A0 A1 A2
#pragma unroll
        for (int q = 0; q < 3; q++) {
            unsigned sa = 0u, sb = 0u;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                int c = reflect101(x0 - 4 + 4 * q + b, w) - xs; // byte of the window; columns no kept output needs may fall outside it
                c = c < 0 ? 0 : (c > 11 ? 11 : c);
                sa |= (c < 8 ? (unsigned)c : 0x0cu) << (8 * b);
                sb |= (c < 8 ? 0x0cu : (unsigned)(c - 8)) << (8 * b);
            }
            selA[q] = sa; selB[q] = sb;
        }
        const unsigned end16 = (unsigned)__mul24(h, pitch) + a0 - 16u; // last 16-byte load that stays inside the image
        blur_wave_rows(cfg, L, r0, dst, dst_off, [&](int y, unsigned &w0, unsigned &w1, unsigned &w2) {
            const unsigned want = row_off(y) + (unsigned)xs; // first byte of the window
            unsigned ld = want & ~3u;
            ld = ld > end16 ? end16 : ld;
            unsigned sh = want - ld; // 0 .. 4
            uint4 q = load16_any(sbase + ld); // the clamped window may start at any byte
            if (sh >= 4u) { q.x = q.y; q.y = q.z; q.z = q.w; sh -= 4u; }
            const unsigned A0 = __builtin_amdgcn_alignbyte(q.y, q.x, sh), A1 = __builtin_amdgcn_alignbyte(q.z, q.y, sh), A2 = __builtin_amdgcn_alignbyte(q.w, q.z, sh);
            w0 = __builtin_amdgcn_perm(A1, A0, selA[0]) | __builtin_amdgcn_perm(0u, A2, selB[0]);
            w1 = __builtin_amdgcn_perm(A1, A0, selA[1]) | __builtin_amdgcn_perm(0u, A2, selB[1]);
            w2 = __builtin_amdgcn_perm(A1, A0, selA[2]) | __builtin_amdgcn_perm(0u, A2, selB[2]);
        });
        return;
    }
    // extended row 0, extended column 0 of the level (rows - PYR_MY .. h + PYR_MY - 1 are materialised; y >= -3 here)
    const int pitch = L.pitch;
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(buf.pyr + (size_t)img * cfg.pyr_bytes + L.pyr_off - (PYR_MY * pitch + PYR_MX), 0,
                                                                       (L.h + 2 * PYR_MY) * pitch, ORBFE_RSRC_FLAGS);
    typedef unsigned bw_v3 __attribute__((ext_vector_type(3)));
    blur_wave_rows(cfg, L, r0, dst, dst_off, [&](int y, unsigned &w0, unsigned &w1, unsigned &w2) {
        const bw_v3 v = __builtin_amdgcn_raw_buffer_load_b96(src, (unsigned)x0, (unsigned)__mul24(y + PYR_MY, pitch), 0); // bytes x0 - 4 .. x0 + 7 of row y
        w0 = v.x; w1 = v.y; w2 = v.z;
    });
}
