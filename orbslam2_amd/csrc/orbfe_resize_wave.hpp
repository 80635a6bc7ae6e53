// orbfe_resize_wave.hpp -- the wave-level body of the LDS-free cv::resize (INTER_LINEAR, 8UC1 fixed point; src/ORBextractor.cc:934), shared by
// orbfe_pyramid.hip (the per-level launches, the two-level kernel) and orbfe_fast.hip (the pyramid levels that ride in FAST's launch
// behind ready counters).  Device code only; see the comment above pyr_resize_direct_kernel (orbfe_pyramid.hip) for the method.
#pragma once
#include "orbfe_common.hpp"

// first source column of extended column i (cv::resize's xofs, as orbfe_create builds the table; it checks this formula against it)
__host__ __device__ __forceinline__ int resize_first_source(int dx, double scale, int src_w)
{
    const float fx = (float)(((double)dx + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    sx = sx < 0 ? 0 : sx;
    return sx >= src_w - 1 ? src_w - 1 : sx;
}
__host__ __device__ __forceinline__ int resize_word_base(int xw, int dst_w, double scale, int src_w)
{
    int lo = 0x7fffffff;
    for (int j = 0; j < 4; j++) {
        int q = 4 * xw + j - PYR_MX;
        if (dst_w == 1) q = 0;
        else while (q < 0 || q >= dst_w) q = q < 0 ? -q : 2 * dst_w - 2 - q;
        lo = q < lo ? q : lo;
    }
    return resize_first_source(lo, scale, src_w);
}

// PACKED0 (round 4): the source is level 0 read IN PLACE from the caller's packed image (DeviceBuffers::lv0_packed) -- rows at any
// alignment, so the 96-bit window is loaded at the row's own 4-byte boundary and the byte shift is per row; the one window that
// could reach past the image's last byte (last source row, last words) is loaded 12 bytes before the image's end instead and
// shifted into place.  The first workgroup of every image also clears the image's status word (ingest's job in copy mode).
struct ResizeStoreGlobal { // the word of extended row y goes to the level's row in HBM
    uint8_t *dst; int pitch;
    __device__ __forceinline__ void operator()(int y, uint32_t out) const { *(uint32_t *)(dst + (ptrdiff_t)(y - PYR_MY) * pitch) = out; }
};
// the same as a write-through store (global_store_dword ... sc1): for the levels that are handed to other workgroups of the SAME
// launch (orbfe_fast.hip) -- the bytes leave the XCD's L2 with the store, so the producer needs no L2 write-back before its flag
struct ResizeStoreGlobalWT {
    uint8_t *dst; int pitch;
    __device__ __forceinline__ void operator()(int y, uint32_t out) const { __hip_atomic_store((uint32_t *)(dst + (ptrdiff_t)(y - PYR_MY) * pitch), out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};
// word xw of the extended rows y0 .. min(y0 + RB, y_end) - 1 of `level`; store(y, word) receives the results
// LOOKUP: the word's first source byte from the host table instead of the double-precision formula: 45 fewer VALU instructions per
// wave (a fifth of the kernel) for one more dependent load -- used for batches of 64 images and more, whose launches are bound by
// instruction issue (0.3 VALU per output pixel), not by the latency of a wave's load chain (round 4: pyramid + blur 174 -> 168 us in the stage table, value + 0.9 %)
template <int RB, bool PACKED0, class Store, bool LOOKUP = false>
__device__ __forceinline__ void resize_direct_rows(const DeviceConfig &cfg, const DeviceBuffers &buf, int level, int img, int xw, int y0, int y_end, Store store);

template <int RB, bool PACKED0 = false, bool LOOKUP = false, class Store = ResizeStoreGlobal>
__device__ __forceinline__ void resize_direct_wave(const DeviceConfig &cfg, const DeviceBuffers &buf, int level, int img, int strip, int band)
{
    const LevelInfo &D = cfg.lv[level];
    const int lane = threadIdx.x & 63;
    if (PACKED0 && strip == 0 && band == 0 && lane == 0) buf.status[img] = 0;
    const int xw = strip * 64 + lane;
    const Store st = {buf.pyr + (size_t)img * cfg.pyr_bytes + D.pyr_off + (xw * 4 - PYR_MX), D.pitch};
    resize_direct_rows<RB, PACKED0, Store, LOOKUP>(cfg, buf, level, img, xw, band * RB, D.rs_ytab_n, st);
}

template <int RB, bool PACKED0, class Store, bool LOOKUP>
__device__ __forceinline__ void resize_direct_rows(const DeviceConfig &cfg, const DeviceBuffers &buf, int level, int img, int xw, int y0, int y_end, Store store)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const LevelInfo &D = cfg.lv[level];
    const LevelInfo &S = cfg.lv[level - 1];
    const int ny = D.rs_ytab_n, nx = D.rs_xtab_n, nwords = nx >> 2;
    if (y0 >= y_end) return;
    if (xw >= nwords) return; // no barriers in here: the spare lanes of the last strip just leave
    const uint32_t *__restrict__ xt = buf.rs_tab + D.rs_xtab_off;
    const uint32_t *__restrict__ dt = buf.rs_tab + D.rs_dtab_off;
    // the row table through the constant address space: never written by a kernel, and only so does the compiler keep its
    // (wave-uniform) reads scalar loads whatever stores to the pyramid are around
    typedef const __attribute__((address_space(4))) uint32_t *rs_const_ptr;
    const rs_const_ptr yt = (rs_const_ptr)(uintptr_t)(buf.rs_tab + D.rs_ytab_off);
    const int base = LOOKUP ? (int)dt[nx + xw] : resize_word_base(xw, D.w, D.rs_scale_x, S.w); // orbfe_create checks the formula against the table
    // three words in ONE global_load_dwordx3 (a struct of three fields is split into two overlapping 64-bit loads as soon as its
    // fields are selected between, as the clamped-window fix-up below does)
    typedef uint32_t win_v __attribute__((ext_vector_type(3)));
    typedef win_v win_ld __attribute__((aligned(4)));
    struct win_t { uint32_t x, y, z; };
    auto ld_win = [](const uint8_t *p) { const win_v v = *(const win_ld *)p; win_t w; w.x = v.x; w.y = v.y; w.z = v.z; return w; };
    const unsigned spitch = PACKED0 ? (unsigned)buf.lv0_pitch : (unsigned)S.pitch;
    const uint8_t *simg = PACKED0 ? buf.lv0 + (size_t)img * buf.lv0_stride : buf.pyr + (size_t)img * cfg.pyr_bytes + S.pyr_off;
    // copy mode: pixel (0,0) and the pitch are 4-byte aligned, the window's shift is the lane's own constant.  In place: offsets from
    // the 4-byte boundary at or below the image's first byte; a window's shift depends on its row
    const unsigned a0 = PACKED0 ? (unsigned)((uintptr_t)simg & 3u) : 0u;
    const uint8_t *sp = PACKED0 ? simg - a0 : simg + (base & ~3);
    const unsigned sh_fixed = (unsigned)base & 3u;
    const unsigned lim = (unsigned)S.h * spitch + a0 - 12u; // PACKED0: the last window that ends inside the image
    uint32_t ye[RB], yb[RB];
#pragma unroll
    for (int k = 0; k < RB; k++) {
        const int yy = y0 + k < y_end ? y0 + k : y_end - 1;
        ye[k] = yt[yy]; yb[k] = yt[ny + yy];
    }
    // window of source row `row`: 12 bytes from the 4-byte boundary at or below the lane's first source byte, and that byte's
    // offset in them.  No branch here: every load of the band is issued before anything waits (a uniform "last row" branch
    // around the load made the compiler serialise them: level 1 took 44 us instead of 28).
    auto load_win = [&](unsigned row, win_t &w, unsigned &sh) {
        if (!PACKED0) { w = ld_win(sp + __umul24(row, spitch)); sh = sh_fixed; return; }
        const unsigned off = __umul24(row, spitch) + ((unsigned)base + a0);
        unsigned ld = off & ~3u;
        ld = ld > lim ? lim : ld; // only in the image's last row, last words: never read past the image (the caller's buffer may end there)
        sh = off - ld;            // 0 .. 3, or up to 11 for a clamped window (whose bytes end at the row's last pixel)
        w = ld_win(sp + ld);
    };
    const uint4 SEL = *(const uint4 *)(dt + 4 * xw);
    const uint4 WT = *(const uint4 *)(xt + nx + 4 * xw);
    win_t wa[RB], wb[RB];
    unsigned sa[RB], sb[RB];
#pragma unroll
    for (int k = 0; k < RB; k++) {
        sa[k] = 0u;
        if (k == 0 || (ye[k] & 0xffffu) != (ye[k - 1] >> 16)) // uniform: else the previous row's lower source row
            load_win(ye[k] & 0xffffu, wa[k], sa[k]);
        load_win(ye[k] >> 16, wb[k], sb[k]);
    }
    const uint32_t sel[4] = {SEL.x, SEL.y, SEL.z, SEL.w}, wt[4] = {WT.x, WT.y, WT.z, WT.w};
    auto hpass = [&](win_t w, unsigned sh, bool last_row, unsigned h[4]) {
        if (PACKED0 && last_row) { // uniform: a clamped window (load_win) brings the wanted bytes to the front
            if (sh >= 8u) { w.x = w.z; sh -= 8u; } else if (sh >= 4u) { w.x = w.y; w.y = w.z; sh -= 4u; }
        }
        const unsigned lo = __builtin_amdgcn_alignbyte(w.y, w.x, sh), hi = __builtin_amdgcn_alignbyte(w.z, w.y, sh);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned pp = __builtin_amdgcn_perm(hi, lo, sel[j]); // S[sx] | S[sx1] << 16
            h[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pp), __builtin_bit_cast(u16x2, wt[j]), 0u, false) & ~15u;
        }
    };
    unsigned hA[4], hB[4];
#pragma unroll
    for (int k = 0; k < RB; k++) {
        if (y0 + k >= y_end) break; // uniform
        if (k > 0 && (ye[k] & 0xffffu) == (ye[k - 1] >> 16)) { // uniform: this row's upper source row is the previous row's lower one
#pragma unroll
            for (int j = 0; j < 4; j++) hA[j] = hB[j];
        } else {
            hpass(wa[k], sa[k], (ye[k] & 0xffffu) + 1u == (unsigned)S.h, hA);
        }
        hpass(wb[k], sb[k], (ye[k] >> 16) + 1u == (unsigned)S.h, hB);
        const unsigned b0 = (yb[k] & 0xffffu) << 12, b1 = (yb[k] >> 16) << 12; // <= 2^23
        uint32_t out = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned v0 = (unsigned)(((unsigned long long)(b0 & 0xffffffu) * (unsigned long long)(hA[j] & 0xffffffu)) >> 32);
            const unsigned v1 = (unsigned)(((unsigned long long)(b1 & 0xffffffu) * (unsigned long long)(hB[j] & 0xffffffu)) >> 32);
            out |= ((v0 + v1 + 2u) >> 2) << (8 * j);
        }
        store(y0 + k, out);
    }
}

