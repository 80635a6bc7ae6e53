// orbfe_pyramid.hip -- ingest, cv::resize pyramid (src/ORBextractor.cc:921-946) and the 7x7 Gaussian (:899-900).
#include "orbfe_common.hpp"

// ---------------------------------------------------------------------------
// Pyramid storage.  Every level is stored with a reflect-101 margin (PYR_MX px on the left, at
// least 8 px on the right, PYR_MY rows above and below) written by its producer (ingest / resize),
// and pixel (0,0) sits on a 4-byte boundary.  Consumers can therefore use aligned 32-bit loads and
// need no border logic: the margin IS cv::GaussianBlur's BORDER_REFLECT_101 (src/ORBextractor.cc:900).
// ---------------------------------------------------------------------------

// ingest: packed images -> level 0 (+ margin); one thread = 4 px of the extended domain, block = 64 words x 4 rows.
// Interior words are one (unaligned) 32-bit load of the packed source; only the margin words gather reflected bytes.
// Also clears the image's status word (first kernel of every chain).
// CN = 3 / 4: interleaved colour input, converted on the fly with cv::cvtColor's 8-bit fixed-point arithmetic
// (Tracking::GrabImage*, src/Tracking.cc:269-294: gray = (c0*k0 + c1*k1 + c2*k2 + half) >> shift; alpha ignored).
// REMAP: the packed input is an unrectified rm_sw x rm_sh image and level 0 is cv::remap(src, map1, map2, INTER_LINEAR)
// of it (EuRoC rectification, Test/Replay/Stereo/stereo_euroc.cc:136-137): 5-bit fixed-point coordinates, the four taps
// weighted with {(32-fx)(32-fy), fx(32-fy), (32-fx)fy, fx*fy} * 32, (sum + 2^14) >> 15, taps outside the source read 0.
__device__ __forceinline__ uint32_t remap_px(const DeviceConfig &cfg, const uint8_t *__restrict__ simg, int side, int x, int y)
{
    const size_t idx = (size_t)y * cfg.width + x;
    const uint32_t xy = cfg.rm_xy[side][idx];
    const int a = cfg.rm_a[side][idx];
    const int sx = (int)(int16_t)(xy & 0xffffu), sy = (int)(int16_t)(xy >> 16), fx = a & 31, fy = a >> 5;
    const int sw = cfg.rm_sw, sh = cfg.rm_sh;
    const bool x0in = sx >= 0 && sx < sw, x1in = sx + 1 >= 0 && sx + 1 < sw, y0in = sy >= 0 && sy < sh, y1in = sy + 1 >= 0 && sy + 1 < sh;
    const uint8_t *r0 = simg + (ptrdiff_t)sy * sw + sx, *r1 = r0 + sw;
    const int v00 = (x0in && y0in) ? r0[0] : 0, v01 = (x1in && y0in) ? r0[1] : 0, v10 = (x0in && y1in) ? r1[0] : 0, v11 = (x1in && y1in) ? r1[1] : 0;
    const int s = v00 * (32 - fx) * (32 - fy) + v01 * fx * (32 - fy) + v10 * (32 - fx) * fy + v11 * fx * fy; // weights / 32
    return (uint32_t)((s * 32 + (1 << 14)) >> 15);
}

#define ING_ROWS 4
template <int CN, bool REMAP = false>
__global__ __launch_bounds__(256) void ingest_kernel(DeviceConfig cfg, DeviceBuffers buf, const uint8_t *__restrict__ src)
{
    const int img = blockIdx.z;
    const LevelInfo &L = cfg.lv[0];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) buf.status[img] = 0; // first kernel of every chain: clear the image's status word
    const int x0 = (int)(blockIdx.x * 64 + (threadIdx.x & 63)) * 4 - PYR_MX;
    if (x0 >= L.w + 8) return;
    // a wave copies ING_ROWS consecutive rows of its 64-word strip (one word per lane and row): four times fewer, longer
    // waves than one row per wave
#pragma unroll
    for (int rr = 0; rr < ING_ROWS; rr++) {
    const int y = (int)(blockIdx.y * (4 * ING_ROWS) + (threadIdx.x >> 6) * ING_ROWS + rr) - PYR_MY;
    if (y >= L.h + PYR_MY) break;
    uint32_t v = 0;
    if constexpr (REMAP) {
        const uint8_t *simg = src + (size_t)img * cfg.in_image_bytes;
        const int side = cfg.rm_xy[1] ? (img & 1) : 0, yy = reflect101(y, L.h);
#pragma unroll
        for (int j = 0; j < 4; j++) v |= remap_px(cfg, simg, side, reflect101(x0 + j, L.w), yy) << (8 * j);
        uint8_t *dr = buf.pyr + (size_t)img * cfg.pyr_bytes + L.pyr_off + (ptrdiff_t)y * L.pitch + x0;
        *(uint32_t *)dr = v;
        continue;
    }
    const uint8_t *s = src + ((size_t)img * L.h + (size_t)reflect101(y, L.h)) * L.w * CN;
    if constexpr (CN == 1) {
        if (x0 >= 0 && x0 + 3 < L.w) {
            __builtin_memcpy(&v, s + x0, 4);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) v |= (uint32_t)s[reflect101(x0 + j, L.w)] << (8 * j);
        }
    } else {
        const int k0 = cfg.in_coef[0], k1 = cfg.in_coef[1], k2 = cfg.in_coef[2], sh = cfg.in_shift, half = 1 << (sh - 1);
        if (x0 >= 0 && x0 + 3 < L.w) {
            uint32_t wds[CN]; // 4 px = 3 or 4 words
            __builtin_memcpy(wds, s + (size_t)x0 * CN, 4 * CN);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t c[3];
#pragma unroll
                for (int k = 0; k < 3; k++) { const int byte = j * CN + k; c[k] = (wds[byte >> 2] >> (8 * (byte & 3))) & 0xffu; }
                v |= ((c[0] * k0 + c[1] * k1 + c[2] * k2 + half) >> sh) << (8 * j);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint8_t *p = s + (size_t)reflect101(x0 + j, L.w) * CN;
                v |= (((uint32_t)p[0] * k0 + (uint32_t)p[1] * k1 + (uint32_t)p[2] * k2 + half) >> sh) << (8 * j);
            }
        }
    }
    uint8_t *d = buf.pyr + (size_t)img * cfg.pyr_bytes + L.pyr_off + (ptrdiff_t)y * L.pitch + x0;
    *(uint32_t *)d = v;
    } // rows
}

// ingest of packed CV_8UC1 images without rectification (the stereo / mono path of the benchmark): a plain copy with margins, so
// a thread moves 16 pixels of one extended row (one unaligned 128-bit load, one aligned 128-bit store) instead of 4; the
// (row, chunk) pairs of an 8-row band are dealt to the threads flat, because an extended row is rarely a multiple of 64 chunks
// (79 for KITTI's 1241 columns).  The chunks that touch the left / right margin gather reflected bytes, in a loop of their own.
#define ING16_ROWS 8
__global__ __launch_bounds__(256) void ingest16_kernel(DeviceConfig cfg, DeviceBuffers buf, const uint8_t *__restrict__ src)
{
    const int img = blockIdx.y;
    const LevelInfo &L = cfg.lv[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) buf.status[img] = 0; // first kernel of every chain: clear the image's status word
    const int ext_w = (L.w + 12 + 3) & ~3;            // extended row in bytes: PYR_MX + w + >= 8, whole words
    const int cpr = (ext_w + 15) >> 4;                // 16-byte chunks per row (the last one may be half used: the pitch covers it)
    const int y_first = (int)blockIdx.x * ING16_ROWS - PYR_MY;
    int rows = L.h + PYR_MY - y_first;
    rows = rows < ING16_ROWS ? rows : ING16_ROWS;
    const uint8_t *simg = src + (size_t)img * L.h * L.w;
    uint8_t *dimg = buf.pyr + (size_t)img * cfg.pyr_bytes + L.pyr_off;
    // interior chunks (all 16 bytes inside the image row): plain copies.  Kept apart from the few chunks that touch a margin:
    // dealt flat together, nearly every wave held one of those and ran their 16-byte-load gather path for all its lanes
    // (38 load instructions per wave instead of 3).
    const int c_hi_ = (L.w - 11 + 15) >> 4;           // first chunk that reaches past the row's last pixel (x0 + 15 >= w)
    const int c_hi = c_hi_ < 1 ? 1 : (c_hi_ > cpr ? cpr : c_hi_);
    const int ci = c_hi - 1;                          // interior chunks 1 .. c_hi - 1 per row
    if (ci > 0) {
        const int ni = rows * ci;
        for (int i = threadIdx.x; i < ni; i += 256) {
            const int r = small_div(i, ci), c = 1 + i - r * ci;
            const int y = y_first + r, x0 = c * 16 - PYR_MX;
            const uint4 v = load16_unaligned(simg + (size_t)reflect101(y, L.h) * L.w + x0);
            *(uint4 *)(dimg + (ptrdiff_t)y * L.pitch + x0) = v;
        }
    }
    const int ne = cpr - ci;                          // chunk 0 and chunks c_hi .. cpr - 1: reflected bytes gathered one by one
    for (int i = threadIdx.x; i < rows * ne; i += 256) {
        const int r = small_div(i, ne), e = i - r * ne;
        const int c = e == 0 ? 0 : c_hi + e - 1;
        const int y = y_first + r, x0 = c * 16 - PYR_MX;
        const uint8_t *s = simg + (size_t)reflect101(y, L.h) * L.w;
        uint32_t w4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t t = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) t |= (uint32_t)s[reflect101(x0 + 4 * k + j, L.w)] << (8 * j);
            w4[k] = t;
        }
        *(uint4 *)(dimg + (ptrdiff_t)y * L.pitch + x0) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
}

// pyramid: level l from level l-1 (cv::resize INTER_LINEAR, 8UC1 fixed point) over the extended (margin-
// included) domain of level l.  The per-column / per-row source offsets and 11-bit weights (cv::resize's
// xofs/ialpha, yofs/ibeta tables) are built once per context on the host with exactly the arithmetic of
// resize.cpp and read here as packed words:
//   X0 = sx | sx1 << 16, X1 = a0 | a1 << 16 (per extended column), Y0 = sy0 | sy1 << 16, Y1 = b0 | b1 << 16.
// A 256-thread workgroup produces 4 * RW extended rows: the source rows they touch (a contiguous range,
// margins index reflected rows) are staged in LDS with coalesced 32-bit loads; wave w then owns RW
// consecutive output rows and a lane owns 4 output columns, whose column table entries stay in registers.
// resize.cpp's horizontal pass of a source row, h = S[sx] * a0 + S[sx1] * a1, is kept in registers for the
// two most recent source rows, so a source row shared by consecutive output rows is filtered once (the
// 1.2 : 1 row ratio makes that 0.83 instead of 2 horizontal passes per output row).  The vertical pass
// ((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) is two 24-bit high multiplies: (b << 12) * (h & ~15)
// = b * (h >> 4) * 2^16, and the sum is <= 1020 so resize.cpp's saturating cast never clips.
template <int RW>
__global__ __launch_bounds__(256) void pyr_resize_kernel(DeviceConfig cfg, DeviceBuffers buf, int level, int src_words, int max_src_rows)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_src[]; // [max_src_rows][src_words * 4]
    const int img = blockIdx.y;
    const LevelInfo &D = cfg.lv[level];
    const LevelInfo &S = cfg.lv[level - 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int y0 = blockIdx.x * (4 * RW); // first extended row of this block
    const uint8_t *src = buf.pyr + (size_t)img * cfg.pyr_bytes + S.pyr_off;
    uint8_t *dst = buf.pyr + (size_t)img * cfg.pyr_bytes + D.pyr_off;
    const uint32_t *xt = buf.rs_tab + D.rs_xtab_off;
    const uint32_t *yt = buf.rs_tab + D.rs_ytab_off;
    const int total_rows = D.h + 2 * PYR_MY;
    const int nrows = (total_rows - y0) < 4 * RW ? (total_rows - y0) : 4 * RW;
    const int row_bytes = (src_words * 4 + 15) & ~15; // staged row pitch
    // table entries this wave needs after the barrier, requested now so that their latency overlaps the staging:
    // the vertical entries of its rows (uniform) and the column entries of its first pass
    const int nwords = (D.w + 12 + 3) >> 2; // extended row in 4-px words
    const int k0 = wave * RW;
    uint32_t Y0r[RW], Y1r[RW];
#pragma unroll
    for (int k = 0; k < RW; k++) {
        const int yy = y0 + (k0 + k < nrows ? k0 + k : nrows - 1);
        Y0r[k] = yt[yy]; Y1r[k] = yt[D.rs_ytab_n + yy];
    }
    const int xi_first = (lane < nwords ? lane : 0) * 4;
    uint4 X0 = *(const uint4 *)(xt + xi_first);
    uint4 X1 = *(const uint4 *)(xt + D.rs_xtab_n + xi_first);
    // source row range of the block: host-built (one scalar load), so the staging below does not wait for a row-table read
    const uint32_t be = buf.rs_blk[D.rs_blk_off + blockIdx.x];
    const int smin = (int)(be & 0xffffu), smax = smin + (int)(be >> 16) - 1;
    int n_src = smax - smin + 1;
    if (n_src > max_src_rows) n_src = max_src_rows; // cannot happen: the host sized max_src_rows from the same table
    {
        // 16-byte chunks, four in flight per thread: unaligned 128-bit loads (a source row starts 4 bytes into its 64-byte aligned
        // pitch), aligned 128-bit LDS stores (row pitch rounded up to 16; the last chunk of a row reads up to 12 bytes of the
        // right margin that no table entry points at)
        const int cpr = row_bytes >> 4;
        int r = small_div(tid, cpr), c = tid - r * cpr;
        const int dr = small_div(256, cpr), dc = 256 - dr * cpr;
        const uint8_t *sp = src + (size_t)smin * S.pitch;
        const int nc = n_src * cpr;
        for (int i0 = tid; i0 < nc; i0 += 1024) {
            uint4 v[4];
            int di[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int rr = r < n_src ? r : n_src - 1; // past the block: repeat a chunk of its last row (same bytes, same place)
                di[u] = __mul24(rr, row_bytes) + 16 * c;
                v[u] = load16_unaligned(sp + (unsigned)(__mul24(rr, S.pitch) + 16 * c));
                c += dc; r += dr;
                if (c >= cpr) { c -= cpr; r++; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) *(uint4 *)(s_src + di[u]) = v[u];
        }
    }
    __syncthreads();
    if (k0 >= nrows) return;
    for (int xw = lane; xw < nwords; xw += 64) {
        const int xi = xw * 4;
        if (xw != lane) {
            X0 = *(const uint4 *)(xt + xi);
            X1 = *(const uint4 *)(xt + D.rs_xtab_n + xi);
        }
        const uint32_t x0v[4] = {X0.x, X0.y, X0.z, X0.w}, x1v[4] = {X1.x, X1.y, X1.z, X1.w};
        int tagA = -1, tagB = -1;          // source rows held in hA / hB
        unsigned hA[4], hB[4];             // (S[sx] * a0 + S[sx1] * a1) & ~15
        auto hpass = [&](int sy, unsigned h[4]) {
            const uint8_t *r = s_src + __mul24(sy - smin, row_bytes);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned p0 = r[x0v[j] & 0xffffu], p1 = r[x0v[j] >> 16];
                h[j] = (__umul24(p0, x1v[j] & 0xffffu) + __umul24(p1, x1v[j] >> 16)) & ~15u;
            }
        };
#pragma unroll
        for (int k = 0; k < RW; k++) {
            if (k0 + k < nrows) {
                const int sy0 = (int)(Y0r[k] & 0xffffu), sy1 = (int)(Y0r[k] >> 16);
                // make hA = row sy0, hB = row sy1 (uniform branches: all lanes walk the same rows)
                if (tagA != sy0) {
                    if (tagB == sy0) {
#pragma unroll
                        for (int j = 0; j < 4; j++) { const unsigned t = hA[j]; hA[j] = hB[j]; hB[j] = t; }
                        tagB = tagA; tagA = sy0;
                    } else {
                        hpass(sy0, hA); tagA = sy0;
                    }
                }
                if (tagB != sy1) {
                    if (sy1 == sy0) {
#pragma unroll
                        for (int j = 0; j < 4; j++) hB[j] = hA[j];
                    } else {
                        hpass(sy1, hB);
                    }
                    tagB = sy1;
                }
                const unsigned b0 = (Y1r[k] & 0xffffu) << 12, b1 = (Y1r[k] >> 16) << 12; // <= 2^23
                uint32_t out = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const unsigned v0 = (unsigned)(((unsigned long long)(b0 & 0xffffffu) * (unsigned long long)(hA[j] & 0xffffffu)) >> 32);
                    const unsigned v1 = (unsigned)(((unsigned long long)(b1 & 0xffffffu) * (unsigned long long)(hB[j] & 0xffffffu)) >> 32);
                    out |= ((v0 + v1 + 2u) >> 2) << (8 * j);
                }
                *(uint32_t *)(dst + (ptrdiff_t)(y0 + k0 + k - PYR_MY) * D.pitch + (xi - PYR_MX)) = out;
            }
        }
    }
}

// The same resize without LDS (round 3).  With the 1.2 : 1 ratio of the pyramid the two source bytes of each of a lane's four
// output columns lie within 8 bytes of the leftmost one (b), so a lane reads a source row as ONE aligned 96-bit global load at
// b & ~3, shifts it to b with two v_alignbyte_b32 and picks its bytes with v_perm_b32 (host-built selectors,
// LevelInfo::rs_dtab_off); the horizontal pass is then a v_dot2_u32_u16 against the packed weights.  A wave owns 64 words x
// RB consecutive extended rows: the row-table entries are wave-uniform scalar loads, b is computed rather than looked up (the
// first loads then wait for nothing but those), every load of the band is issued before the first store, and there is no
// staging, no barrier and no byte gather through the LDS pipe, which is what bounded pyr_resize_kernel.  A source row shared
// by consecutive output rows is loaded and filtered once (a uniform branch).  Measured, KITTI levels 1-4, 128 images:
// 36.4 / 26.7 / 19.6 / 17.0 us (pyr_resize_kernel) -> 28.0 / 19.9 / 15.2 / 13.2 us; what is left is ~6 us per launch of
// dependent latency (kernel arguments -> row table -> source rows) plus ~4.6 TB/s of traffic.  Variants (A/B on one box,
// pyramid stage in ms, LDS kernel 0.131): byte-aligned 64-bit loads with the duplicates 0.118-0.120 (unaligned loads cost the
// texture addresser about twice an aligned one), the same without duplicates 0.107, aligned 0.107, RB 2 / 4 / 8: 0.124 / 0.108 /
// 0.110; a wave walking 16 / 32 / 64 rows with the next group's loads in flight: 0.124 / 0.130 / 0.154 (fewer, longer waves
// lose: these launches live on the number of independent waves).  Levels whose words reach further than 8 bytes (scale
// factors above ~2) keep pyr_resize_kernel, and ORBFE_PYR_LDS=1 forces it.
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
int orbfe_resize_word_base_host(int xw, int dst_w, double scale, int src_w) { return resize_word_base(xw, dst_w, scale, src_w); }

// PACKED0 (round 4): the source is level 0 read IN PLACE from the caller's packed image (DeviceBuffers::lv0_packed) -- rows at any
// alignment, so the 96-bit window is loaded at the row's own 4-byte boundary and the byte shift is per row; the one window that
// could reach past the image's last byte (last source row, last words) is loaded 12 bytes before the image's end instead and
// shifted into place.  The first workgroup of every image also clears the image's status word (ingest's job in copy mode).
// Buffer addressing (round 5): descriptor (wave-uniform base) + scalar row offset + the lane's own 32-bit offset, so a row's address costs
// no vector instruction (flat addressing made every load / store a 64-bit vector add, the stores a v_mad_i64_i32 as well).
struct ResizeStoreGlobal { // the word of extended row y goes to the level's row in HBM
    __amdgpu_buffer_rsrc_t dst; unsigned off, pitch; // dst = extended row 0 of the lane's image and level, off = the lane's word in a row
    __device__ __forceinline__ void operator()(int y, uint32_t out) const { __builtin_amdgcn_raw_buffer_store_b32(out, dst, off, (unsigned)y * pitch, 0); }
};
// word xw of the extended rows y0 .. min(y0 + RB, y_end) - 1 of `level`; store(y, word) receives the results
// LOOKUP: the word's first source byte from the host table instead of the double-precision formula: 45 fewer VALU instructions per
// wave (a fifth of the kernel) for one more dependent load -- used for batches of 64 images and more (round 4: pyramid + blur 174 -> 168 us in
// the stage table, value + 0.9 %; the blur still rode in these launches then and made them issue-bound.  Round 5: the blur rides in FAST's
// launch and what is left here is bound by the memory path -- a third fewer instructions bought 3 % -- but the table still wins by ~1 us)
template <int RB, bool PACKED0, class Store, bool LOOKUP = false>
__device__ __forceinline__ void resize_direct_rows(const DeviceConfig &cfg, const DeviceBuffers &buf, int level, int img, int xw, int y0, int y_end, Store store);

template <int RB, bool PACKED0 = false, bool LOOKUP = false>
__device__ __forceinline__ void resize_direct_wave(const DeviceConfig &cfg, const DeviceBuffers &buf, int level, int img, int strip, int band)
{
    const LevelInfo &D = cfg.lv[level];
    const int lane = threadIdx.x & 63;
    if (PACKED0 && strip == 0 && band == 0 && lane == 0) buf.status[img] = 0;
    const int xw = strip * 64 + lane;
    uint8_t *drow0 = buf.pyr + (size_t)img * cfg.pyr_bytes + D.pyr_off - (PYR_MY * D.pitch + PYR_MX); // extended row 0, extended column 0
    const ResizeStoreGlobal st = {__builtin_amdgcn_make_buffer_rsrc(drow0, 0, D.rs_ytab_n * D.pitch, ORBFE_RSRC_FLAGS), (unsigned)xw * 4u, (unsigned)D.pitch};
    resize_direct_rows<RB, PACKED0, ResizeStoreGlobal, LOOKUP>(cfg, buf, level, img, xw, band * RB, D.rs_ytab_n, st);
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
    struct win_t { uint32_t x, y, z; };
    auto ld_win = [](__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { const win_v v = __builtin_amdgcn_raw_buffer_load_b96(r, voff, soff, 0); win_t w; w.x = v.x; w.y = v.y; w.z = v.z; return w; };
    const unsigned spitch = PACKED0 ? (unsigned)buf.lv0_pitch : (unsigned)S.pitch;
    const uint8_t *simg = PACKED0 ? buf.lv0 + (size_t)img * buf.lv0_stride : buf.pyr + (size_t)img * cfg.pyr_bytes + S.pyr_off;
    // copy mode: pixel (0,0) and the pitch are 4-byte aligned, the window's shift is the lane's own constant.  In place: offsets from
    // the 4-byte boundary at or below the image's first byte; a window's shift depends on its row
    const unsigned a0 = PACKED0 ? (unsigned)((uintptr_t)simg & 3u) : 0u;
    // source descriptor: the image's level from the 4-byte boundary at or below its first byte, to the end of its last row
    const __amdgpu_buffer_rsrc_t sp = __builtin_amdgcn_make_buffer_rsrc((void *)(PACKED0 ? simg - a0 : simg), 0, (unsigned)S.h * spitch + (PACKED0 ? a0 : 64u), ORBFE_RSRC_FLAGS); // copy mode: a last-row window may run into the bottom margin
    const unsigned voff_fixed = (unsigned)base & ~3u, sh_fixed = (unsigned)base & 3u;
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
        if (!PACKED0) { w = ld_win(sp, voff_fixed, __umul24(row, spitch)); sh = sh_fixed; return; }
        const unsigned off = __umul24(row, spitch) + ((unsigned)base + a0);
        unsigned ld = off & ~3u;
        ld = ld > lim ? lim : ld; // only in the image's last row, last words: never read past the image (the caller's buffer may end there)
        sh = off - ld;            // 0 .. 3, or up to 11 for a clamped window (whose bytes end at the row's last pixel)
        w = ld_win(sp, ld, 0u);
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
            h[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pp), __builtin_bit_cast(u16x2, wt[j]), 0u, false) & ~15u; // < 2^20
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
        const unsigned b0 = (yb[k] & 0xffffu) << 12, b1 = (yb[k] >> 16) << 12; // <= 2^23, wave-uniform
        // (b * (h >> 4)) >> 16 = the high word of the 24 x 24-bit product (b << 12) * (h & ~15), stated as the instruction so that the
        // operands' masks (which the values already satisfy) are not re-applied after every select between hA and hB
        auto mulhi24 = [](unsigned b, unsigned h) { unsigned r; asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "s"(b), "v"(h)); return r; };
        unsigned t[4];
#pragma unroll
        for (int j = 0; j < 4; j++) t[j] = mulhi24(b0, hA[j]) + mulhi24(b1, hB[j]) + 2u; // <= 1022
        // bytes (t >> 2) of the four pixels: pixels 0 | 2 and 1 | 3 share a register 16 bits apart, one shift each, one byte pick
        const unsigned e = (t[0] + (t[2] << 16)) >> 2, o = (t[1] + (t[3] << 16)) >> 2;
        store(y0 + k, __builtin_amdgcn_perm(o, e, 0x06020400u));
    }
}

template <int RB, bool PACKED0 = false, bool LOOKUP = false>
__global__ __launch_bounds__(256) void pyr_resize_direct_kernel(DeviceConfig cfg, DeviceBuffers buf, int level)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    resize_direct_wave<RB, PACKED0, LOOKUP>(cfg, buf, level, blockIdx.z, blockIdx.x, (int)blockIdx.y * 4 + wave);
}

// ---------------------------------------------------------------------------
// Fused pyramid tail: the last NST levels (NST = 2 or 3) in ONE launch.  On their own these levels are latency bound: each
// launch is a chain of dependent table loads, staging, one barrier and a few microseconds of arithmetic (10 us per level for
// 2 us of work at 346 x 105), and each re-reads from HBM what the previous one just wrote.  Here a workgroup owns a STRIP of
// ORBFE_TAIL_COLS extended columns of the last level over all its rows and walks the chain in LDS: it stages the columns of
// level F - 1 it needs (all rows), computes from them its columns of level F (written to HBM and kept in LDS), from those its
// columns of level F + 1, and so on.  Which columns those are is a host-built plan (orbfe_api.hip, from the same column
// tables), including the margin columns at the left / right of each level, which no later level reads; a strip's neighbours
// share one or two columns per level with it (computed by both: same arithmetic, same bytes) -- cutting by rows instead costs
// 40 % in halo rows (measured: no gain over the separate launches).
// A lane owns one 4-pixel word of the strip (its column-table entries stay in registers for the whole level) and a run of
// consecutive rows: 64 / words lanes share a word and split the wave's rows between them, so the horizontal pass of a source
// row is still reused by the next output row.  Row tables sit in LDS.  The arithmetic is pyr_resize_kernel's.
// ---------------------------------------------------------------------------
#define TAIL_THREADS 512
template <int NST>
__global__ __launch_bounds__(TAIL_THREADS) void pyr_tail_kernel(DeviceConfig cfg, DeviceBuffers buf)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_tail[];
    const int img = blockIdx.y, strip = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int F = cfg.tail_first;
    const int *plan = buf.tail_plan + strip * (ORBFE_TAIL_MAX * 4);
#ifdef ORBFE_PROFILE_CUTS // tools/tail_timeline.py (`make cuts` build only): phase boundaries of image 0's workgroups, 100 MHz clock
    long long *ph = tid == 0 && img == 0 && strip < 8 ? buf.dbg_ts + 3072 + 16 * strip : nullptr;
    struct Stamp { // start / end of the first 1024 workgroups
        long long *p;
        __device__ Stamp(long long *q) : p(q) { if (p) p[0] = (long long)__builtin_amdgcn_s_memrealtime(); }
        __device__ ~Stamp() { if (p) p[1] = (long long)__builtin_amdgcn_s_memrealtime(); }
    } stamp(tid == 0 && (img * (int)gridDim.x + strip) < 1024 ? buf.dbg_ts + 2 * (img * (int)gridDim.x + strip) : nullptr);
    int ph_n = 0;
#define TAIL_PHASE() do { if (ph && ph_n < 15) ph[ph_n++] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TAIL_PHASE() do { } while (0)
#endif
    TAIL_PHASE();
    uint8_t *pyr = buf.pyr + (size_t)img * cfg.pyr_bytes;
    // this lane's word and row run of every stage, and the word's column-table entries (issued first: longest latency)
    int xi_[NST], yb_[NST], ye_[NST];
    uint4 X0_[NST], X1_[NST];
#pragma unroll
    for (int st = 0; st < NST; st++) {
        const LevelInfo &D = cfg.lv[F + st];
        const int x0 = plan[st * 4], nwd = plan[st * 4 + 1];
        const int nq = 64 / nwd;                       // lanes per word
        const int q = small_div(lane, nwd), w = lane - q * nwd;
        const int R = D.h + 2 * PYR_MY, rpw = (R + TAIL_THREADS / 64 - 1) / (TAIL_THREADS / 64); // rows per wave, then per lane
        const int wb = wave * rpw, we = wb + rpw < R ? wb + rpw : R;
        const int rpl = (rpw + nq - 1) / nq;
        yb_[st] = wb + q * rpl;
        ye_[st] = q < nq ? (yb_[st] + rpl < we ? yb_[st] + rpl : we) : yb_[st];
        xi_[st] = x0 + 4 * w;
        const uint32_t *xt = buf.rs_tab + D.rs_xtab_off;
        X0_[st] = *(const uint4 *)(xt + xi_[st]);
        X1_[st] = *(const uint4 *)(xt + D.rs_xtab_n + xi_[st]);
    }
    // row tables of every stage -> LDS (two planes of rs_ytab_n words each, contiguous in rs_tab)
#pragma unroll
    for (int st = 0; st < NST; st++) {
        const LevelInfo &D = cfg.lv[F + st];
        const uint32_t *g = buf.rs_tab + D.rs_ytab_off;
        uint32_t *d = (uint32_t *)(s_tail + cfg.tail_lds_y[st]);
        for (int i = tid; i < 2 * D.rs_ytab_n; i += TAIL_THREADS) d[i] = g[i];
    }
    // the strip's columns of level F - 1, every row (interior pixels, 4-aligned start); this copy is the kernel's longest
    // dependent chain
    const int src_x0 = plan[2], src_words = plan[3];
    {
        const LevelInfo &S = cfg.lv[F - 1];
        const uint8_t *sp = pyr + S.pyr_off + src_x0;
        uint32_t *d = (uint32_t *)(s_tail + cfg.tail_lds_src);
        // 16-byte chunks of a row (unaligned 128-bit loads), four in flight per thread; the LDS rows keep their pitch of
        // src_words words, so a chunk is stored as up to four words (the last chunk of a row may be partial)
        const int cpr = (src_words + 3) >> 2;
        int r = small_div(tid, cpr), c = tid - r * cpr;
        const int dr = small_div(TAIL_THREADS, cpr), dc = TAIL_THREADS - dr * cpr;
        const int nc = S.h * cpr;
        for (int i0 = tid; i0 < nc; i0 += 4 * TAIL_THREADS) {
            uint4 v[4];
            int di[4], left[4]; // first word of the chunk in LDS, words up to the row's end
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int rr = r < S.h ? r : S.h - 1; // past the level: repeat a chunk of its last row (same words, same place)
                left[u] = src_words - 4 * c;
                di[u] = __mul24(rr, src_words) + 4 * c;
                v[u] = load16_unaligned(sp + (unsigned)(__mul24(rr, S.pitch) + 16 * c));
                c += dc; r += dr;
                if (c >= cpr) { c -= cpr; r++; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
                d[di[u]] = v[u].x;
                if (left[u] > 1) d[di[u] + 1] = v[u].y;
                if (left[u] > 2) d[di[u] + 2] = v[u].z;
                if (left[u] > 3) d[di[u] + 3] = v[u].w;
            }
        }
    }
    __syncthreads();
    TAIL_PHASE();
#pragma unroll
    for (int st = 0; st < NST; st++) {
        const LevelInfo &D = cfg.lv[F + st];
        // source in LDS: interior pixel (sx, sy) of the source level at src[sy * src_pitch + sx]
        const uint8_t *src;
        int src_pitch;
        if (st == 0) {
            src_pitch = src_words * 4;
            src = s_tail + cfg.tail_lds_src - src_x0;
        } else {
            src_pitch = plan[(st - 1) * 4 + 1] * 4;
            src = s_tail + cfg.tail_lds_buf[st - 1] + PYR_MY * src_pitch + (PYR_MX - plan[(st - 1) * 4]);
        }
        const int keep_pitch = plan[st * 4 + 1] * 4;
        uint8_t *keep = st + 1 < NST ? s_tail + cfg.tail_lds_buf[st] + (xi_[st] - plan[st * 4]) : nullptr; // extended rows, the strip's columns
        uint8_t *dst = pyr + D.pyr_off + (xi_[st] - PYR_MX);
        const uint32_t *yt = (const uint32_t *)(s_tail + cfg.tail_lds_y[st]);
        const int ny = D.rs_ytab_n;
        const uint32_t x0v[4] = {X0_[st].x, X0_[st].y, X0_[st].z, X0_[st].w}, x1v[4] = {X1_[st].x, X1_[st].y, X1_[st].z, X1_[st].w};
        int tagA = -1, tagB = -1;          // source rows held in hA / hB (per lane: lanes of different runs are at different rows)
        unsigned hA[4], hB[4];
        auto hpass = [&](int sy, unsigned h[4]) {
            const uint8_t *r = src + __mul24(sy, src_pitch);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned p0 = r[x0v[j] & 0xffffu], p1 = r[x0v[j] >> 16];
                h[j] = (__umul24(p0, x1v[j] & 0xffffu) + __umul24(p1, x1v[j] >> 16)) & ~15u;
            }
        };
        uint32_t y0n = 0u, y1n = 0u; // row-table entries one row ahead
        if (yb_[st] < ye_[st]) { y0n = yt[yb_[st]]; y1n = yt[ny + yb_[st]]; }
        for (int y = yb_[st]; y < ye_[st]; y++) {
            const uint32_t y0e = y0n, y1e = y1n;
            if (y + 1 < ye_[st]) { y0n = yt[y + 1]; y1n = yt[ny + y + 1]; }
            const int sy0 = (int)(y0e & 0xffffu), sy1 = (int)(y0e >> 16);
            if (tagA != sy0) {
                if (tagB == sy0) {
#pragma unroll
                    for (int j = 0; j < 4; j++) { const unsigned t = hA[j]; hA[j] = hB[j]; hB[j] = t; }
                    tagB = tagA; tagA = sy0;
                } else {
                    hpass(sy0, hA); tagA = sy0;
                }
            }
            if (tagB != sy1) {
                if (sy1 == sy0) {
#pragma unroll
                    for (int j = 0; j < 4; j++) hB[j] = hA[j];
                } else {
                    hpass(sy1, hB);
                }
                tagB = sy1;
            }
            const unsigned b0 = (y1e & 0xffffu) << 12, b1 = (y1e >> 16) << 12;
            uint32_t out = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned v0 = (unsigned)(((unsigned long long)(b0 & 0xffffffu) * (unsigned long long)(hA[j] & 0xffffffu)) >> 32);
                const unsigned v1 = (unsigned)(((unsigned long long)(b1 & 0xffffffu) * (unsigned long long)(hB[j] & 0xffffffu)) >> 32);
                out |= ((v0 + v1 + 2u) >> 2) << (8 * j);
            }
            *(uint32_t *)(dst + (ptrdiff_t)(y - PYR_MY) * D.pitch) = out;
            if (keep) *(uint32_t *)(keep + __mul24(y, keep_pitch)) = out;
        }
        if (st + 1 < NST) __syncthreads();
        TAIL_PHASE();
    }
}

#include "orbfe_blur_wave.hpp"

// the waves of a workgroup are independent; tiles [tile_begin, tile_end) of the list (level-major: a range is a set of levels)
__global__ __launch_bounds__(256) void blur_kernel(DeviceConfig cfg, DeviceBuffers buf, int tile_begin, int tile_end)
{
    const int u = tile_begin + (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (u >= tile_end) return;
    blur_wave(cfg, buf, blockIdx.y, u);
}

// Level l from level l - 1 AND the blur of level l - 1 in one launch (both only read level l - 1), as independent waves of one
// grid: no second stream, no event (those cost more than they return: tools/experiments).  The blur workgroups of ALL images are
// dispatched first, the resize workgroups after them.  Measured (pyramid + blur stages, A/B on one box, separate launches
// 0.186 ms): this order 0.181; blur and resize workgroups interleaved image by image, either one first, 0.198-0.200 (mixed,
// the two kinds of waves slow each other down by more than the overlap returns: the blur streams whole rows at ~5 TB/s, the
// resize lives on many short waves); 16- / 8-row blur bands in the interleaved order 0.189 / 0.192; s_setprio 3 for the
// resize waves: no change.  So what the fusion buys is the launch boundary and the drain of the blur's last waves, 4 x ~1 us.
// ORBFE_NO_FUSE=1 (orbfe_create) keeps the launches apart.
template <int RB, bool PACKED0 = false, bool LOOKUP = false>
__global__ __launch_bounds__(256) void pyr_resize_blur_kernel(DeviceConfig cfg, DeviceBuffers buf, int level, int strips, int n_resize, int tile_begin, int tile_end)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // grid = (image, item): every image's blur items come before any resize item (dispatch order is x fastest)
    const int n_blur = (tile_end - tile_begin + 3) >> 2;
    const int img = blockIdx.x, x = (int)blockIdx.y < n_blur ? n_resize + (int)blockIdx.y : (int)blockIdx.y - n_blur;
    if (x < n_resize) {
        const int bg = __builtin_amdgcn_readfirstlane(small_div(x, strips));
        resize_direct_wave<RB, PACKED0, LOOKUP>(cfg, buf, level, img, x - bg * strips, bg * 4 + wave);
    } else {
        const int u = tile_begin + (x - n_resize) * 4 + wave;
        if (u < tile_end) blur_wave(cfg, buf, img, u);
    }
}

// ---------------------------------------------------------------------------
// Two pyramid levels per launch (round 4): level l from level l - 1 in HBM, and level l + 1 from the rows of level l the workgroup
// has just computed, kept in LDS -- the dependent chain of the pyramid has one launch per PAIR of levels, and level l is not read
// back from HBM for its own resize.  A workgroup owns a tile of PP_TW words x PP_TR extended rows of level l + 1.  Phase 1: its four
// waves compute the rectangle of level l (extended coordinates) that the tile's sources lie in -- with pyr_resize_direct's waves,
// results to an LDS tile (<= 64 words x 16 rows) and, for the part of the rectangle this workgroup is RESPONSIBLE for, to HBM: the
// responsibility rectangles of all workgroups partition level l's extended domain (margins go to the border tiles), so every
// pixel of level l is stored exactly once although neighbouring tiles' rectangles overlap by a column / row or two (computed by both:
// same arithmetic, same bytes; ~12 % more level-l work).  Phase 2 (after one barrier): the same resize from the LDS tile (three
// aligned LDS words per row and lane instead of a 96-bit global load, then the identical v_alignbyte / v_perm / v_dot2 passes).
// Both levels must be resizable by the LDS-free kernel (LevelInfo::rs_direct); the host plans the rectangles from the same tables
// (orbfe_api.hip: pair plan) and falls back to single-level launches when a rectangle would not fit.  ORBFE_NO_PAIR=1 disables it.
// ---------------------------------------------------------------------------
#define PP_LDS_ROWS 17 // 16 tile rows + one spare: a lane's 12-byte window may run past its row's last word
template <bool PACKED0>
__global__ __launch_bounds__(256) void pyr_pair_kernel(DeviceConfig cfg, DeviceBuffers buf, int l1, int n_pair, int tile_begin, int tile_end)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) uint32_t s_tile[PP_LDS_ROWS * 64];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int n_blur = (tile_end - tile_begin + 3) >> 2;
    const int img = blockIdx.x;
    if ((int)blockIdx.y < n_blur) { // the blur of the levels below rides in front (as in pyr_resize_blur_kernel)
        const int u = tile_begin + (int)blockIdx.y * 4 + wave;
        if (u < tile_end) blur_wave(cfg, buf, img, u);
        return;
    }
    const int x = (int)blockIdx.y - n_blur;
    if (x >= n_pair) return;
    const LevelInfo &D1 = cfg.lv[l1];
    const LevelInfo &D2 = cfg.lv[l1 + 1];
    const int ntx = D1.pp_ntx;
    const int ty = __builtin_amdgcn_readfirstlane(small_div(x, ntx)), tx = x - ty * ntx;
    if (PACKED0 && x == 0 && threadIdx.x == 0) buf.status[img] = 0; // first launch of the chain when level 0 is read in place
    const int4 px = ((const int4 *)buf.pair_plan)[D1.pp_xoff + tx]; // first word, words of the LDS tile; first, end word this group stores to HBM
    const int4 py = ((const int4 *)buf.pair_plan)[D1.pp_yoff + ty]; // the same for rows
    // ---- phase 1: level l1, rows py.x + 4 * wave ..., word px.x + lane ----
    {
        const int xw = px.x + lane, y0 = py.x + 4 * wave, y_end = py.x + py.y;
        uint8_t *g = buf.pyr + (size_t)img * cfg.pyr_bytes + D1.pyr_off + (xw * 4 - PYR_MX);
        const int pitch = D1.pitch;
        const bool mine_x = xw >= px.z && xw < px.w;
        uint32_t *t = s_tile + lane;
        auto st = [&](int y, uint32_t out) {
            t[(y - py.x) * 64] = out;
            if (mine_x && y >= py.z && y < py.w) *(uint32_t *)(g + (ptrdiff_t)(y - PYR_MY) * pitch) = out;
        };
        if (lane < px.y) resize_direct_rows<4, PACKED0>(cfg, buf, l1, img, xw, y0, y_end, st);
    }
    __syncthreads();
    // ---- phase 2: level l1 + 1 from the tile ----
    const int nw2 = D2.rs_xtab_n >> 2, ny2 = D2.rs_ytab_n, nx2 = D2.rs_xtab_n;
    const int xw2 = tx * D1.pp_tw + lane;
    if (lane >= D1.pp_tw || xw2 >= nw2) return;
    const int rb2 = (D1.pp_tr + 3) >> 2; // rows per wave
    const int y2_0 = ty * D1.pp_tr + wave * rb2;
    int y2_end = ty * D1.pp_tr + D1.pp_tr;
    y2_end = y2_end < y2_0 + rb2 ? y2_end : y2_0 + rb2;
    y2_end = y2_end < ny2 ? y2_end : ny2;
    if (y2_0 >= y2_end) return;
    typedef const __attribute__((address_space(4))) uint32_t *rs_const_ptr;
    const rs_const_ptr yt = (rs_const_ptr)(uintptr_t)(buf.rs_tab + D2.rs_ytab_off);
    const uint4 SEL = *(const uint4 *)(buf.rs_tab + D2.rs_dtab_off + 4 * xw2);
    const uint4 WT = *(const uint4 *)(buf.rs_tab + D2.rs_xtab_off + nx2 + 4 * xw2);
    const uint32_t sel[4] = {SEL.x, SEL.y, SEL.z, SEL.w}, wt[4] = {WT.x, WT.y, WT.z, WT.w};
    const int base = resize_word_base(xw2, D2.w, D2.rs_scale_x, D1.w); // first source column (interior of level l1)
    const int bo = base + PYR_MX - 4 * px.x;                            // ... as a byte of the tile row
    const uint32_t *tw = s_tile + (bo >> 2);
    const unsigned sh = (unsigned)bo & 3u;
    auto hpass = [&](int srow, unsigned h[4]) { // source row as an interior row of level l1
        const uint32_t *r = tw + (srow + PYR_MY - py.x) * 64;
        const uint32_t w0 = r[0], w1 = r[1], w2 = r[2];
        const unsigned lo = __builtin_amdgcn_alignbyte(w1, w0, sh), hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned pp = __builtin_amdgcn_perm(hi, lo, sel[j]);
            h[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pp), __builtin_bit_cast(u16x2, wt[j]), 0u, false) & ~15u; // < 2^20
        }
    };
    uint8_t *dst = buf.pyr + (size_t)img * cfg.pyr_bytes + D2.pyr_off + (xw2 * 4 - PYR_MX);
    unsigned hA[4], hB[4];
    int have = -1; // source row whose horizontal pass sits in hB
    for (int y = y2_0; y < y2_end; y++) { // uniform
        const uint32_t ye = yt[y], yb = yt[ny2 + y];
        const int s0 = (int)(ye & 0xffffu), s1 = (int)(ye >> 16);
        if (s0 == have) {
#pragma unroll
            for (int j = 0; j < 4; j++) hA[j] = hB[j];
        } else {
            hpass(s0, hA);
        }
        hpass(s1, hB);
        have = s1;
        const unsigned b0 = (yb & 0xffffu) << 12, b1 = (yb >> 16) << 12;
        uint32_t out = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned v0 = (unsigned)(((unsigned long long)(b0 & 0xffffffu) * (unsigned long long)(hA[j] & 0xffffffu)) >> 32);
            const unsigned v1 = (unsigned)(((unsigned long long)(b1 & 0xffffffu) * (unsigned long long)(hB[j] & 0xffffffu)) >> 32);
            out |= ((v0 + v1 + 2u) >> 2) << (8 * j);
        }
        *(uint32_t *)(dst + (ptrdiff_t)(y - PYR_MY) * D2.pitch) = out;
    }
}

void orbfe_launch_ingest(const DeviceConfig &cfg, const DeviceBuffers &buf, const uint8_t *d_images, int n_images, hipStream_t s)
{
    const int words = (cfg.lv[0].w + 12 + 3) / 4;
    dim3 grid((words + 63) / 64, (cfg.lv[0].h + 2 * PYR_MY + 4 * ING_ROWS - 1) / (4 * ING_ROWS), n_images);
    if (!cfg.rm_on && cfg.in_cn == 1) {
        dim3 grid16((cfg.lv[0].h + 2 * PYR_MY + ING16_ROWS - 1) / ING16_ROWS, n_images);
        hipLaunchKernelGGL(ingest16_kernel, grid16, dim3(256), 0, s, cfg, buf, d_images);
        return;
    }
    if (cfg.rm_on) hipLaunchKernelGGL((ingest_kernel<1, true>), grid, dim3(256), 0, s, cfg, buf, d_images);
    else if (cfg.in_cn == 3) hipLaunchKernelGGL(ingest_kernel<3>, grid, dim3(256), 0, s, cfg, buf, d_images);
    else if (cfg.in_cn == 4) hipLaunchKernelGGL(ingest_kernel<4>, grid, dim3(256), 0, s, cfg, buf, d_images);
    else hipLaunchKernelGGL(ingest_kernel<1>, grid, dim3(256), 0, s, cfg, buf, d_images);
}

int orbfe_launch_pyramid(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, bool fuse_blur, hipStream_t s, int ride_from)
{
    // ride_from: the blur of levels >= ride_from is left to the caller (FAST's launch carries it: orbfe_launch_fast); the lower
    // levels are blurred beside the resize that reads them, as far as the launches reach
    // Fused tail or one launch per level for the last levels: the tail is one launch instead of three (small batches are bound by
    // the chain's latency: 8 pairs per step 46.9 k pairs/s against 45.4 k), three direct launches carry the blur of the level below
    // and leave the last blur launch one level instead of four (64 pairs: 86.4 k -> 87.0 k, three chains in flight 96.2 -> 96.9 k)
    const bool tail = cfg.tail_first && n_images <= cfg.tail_max_images;
    const bool lookup = n_images >= 64; // large batches: the table (fewer instructions); small ones are latency-bound: the formula (one dependent load less)
    const int last_single = tail ? cfg.tail_first - 1 : cfg.nlevels - 1;
    int blurred = 0; // levels 0 .. blurred - 1 have had their blur launched (beside the resize that reads them)
    for (int l = 1; l <= last_single; l++) {
        const int src_words = (cfg.lv[l - 1].w + 3) / 4; // interior pixels of the source row (4-aligned start)
        const int total_rows = cfg.lv[l].h + 2 * PYR_MY;
        const size_t rowp = ((size_t)src_words * 4 + 15) & ~(size_t)15; // staged row pitch
        // rows per wave: the largest of 4, 2, 1 whose staged source rows fit 60 KB of LDS (rs_src_rows[i] = source
        // row span of the worst block of 16 / 8 / 4 output rows, from the host's row table)
        const int *span = cfg.lv[l].rs_src_rows;
        // Pairs serve the same batches as the fused tail (fewer than 64 images, where the chain of dependent launches is what a step
        // waits for: a single stereo pair's five pyramid launches 48 us -> three 34 us, 8 pairs per step +6 %); at 128 images every
        // per-level launch is throughput-bound and the pair kernel's ~15 % recomputed pixels and its barrier make the pyramid 15 us
        // SLOWER (170 -> 185 us), so large batches keep one launch per level.  ORBFE_NO_PAIR=0 forces pairs, =1 forbids them.
        if (cfg.lv[l].pp_ok && l + 1 <= last_single && n_images <= cfg.pp_max_images) { // this level and the next in one launch; the blur of every finished level below ride_from rides along
            const bool ride = fuse_blur && blurred <= l - 1 && blurred < ride_from;
            const int lb = l - 1 < ride_from - 1 ? l - 1 : ride_from - 1; // last level blurred here
            const int t0 = ride ? cfg.lv[blurred].blur_tile_off : 0, t1 = ride ? cfg.lv[lb].blur_tile_off + cfg.lv[lb].blur_tiles_x * cfg.lv[lb].blur_tiles_y : 0;
            const int n_pair = cfg.lv[l].pp_ntx * cfg.lv[l].pp_nty;
            dim3 grid(n_images, n_pair + (t1 - t0 + 3) / 4);
            if (l == 1 && buf.lv0_packed) hipLaunchKernelGGL(pyr_pair_kernel<true>, grid, dim3(256), 0, s, cfg, buf, l, n_pair, t0, t1);
            else hipLaunchKernelGGL(pyr_pair_kernel<false>, grid, dim3(256), 0, s, cfg, buf, l, n_pair, t0, t1);
            if (ride) blurred = lb + 1;
            l++; // level l + 1 is done too
            continue;
        }
        if (cfg.lv[l].rs_direct) {
            const int nwords = cfg.lv[l].rs_xtab_n >> 2;
            constexpr int rb = ORBFE_PYR_RB > 0 ? ORBFE_PYR_RB : 8;
            const int strips = (nwords + 63) / 64, groups = (total_rows + 4 * rb - 1) / (4 * rb);
            if (fuse_blur && blurred <= l - 1 && blurred < ride_from) { // every level finished by the earlier launches and not blurred yet
                const LevelInfo &P = cfg.lv[l - 1 < ride_from - 1 ? l - 1 : ride_from - 1];
                const int t0 = cfg.lv[blurred].blur_tile_off, t1 = P.blur_tile_off + P.blur_tiles_x * P.blur_tiles_y;
                dim3 grid(n_images, strips * groups + (t1 - t0 + 3) / 4);
                const bool packed = l == 1 && buf.lv0_packed;
                if (packed && lookup) hipLaunchKernelGGL((pyr_resize_blur_kernel<rb, true, true>), grid, dim3(256), 0, s, cfg, buf, l, strips, strips * groups, t0, t1);
                else if (packed) hipLaunchKernelGGL((pyr_resize_blur_kernel<rb, true, false>), grid, dim3(256), 0, s, cfg, buf, l, strips, strips * groups, t0, t1);
                else if (lookup) hipLaunchKernelGGL((pyr_resize_blur_kernel<rb, false, true>), grid, dim3(256), 0, s, cfg, buf, l, strips, strips * groups, t0, t1);
                else hipLaunchKernelGGL((pyr_resize_blur_kernel<rb, false, false>), grid, dim3(256), 0, s, cfg, buf, l, strips, strips * groups, t0, t1);
                blurred = l < ride_from ? l : ride_from;
            } else {
                dim3 grid(strips, groups, n_images);
                const bool packed = l == 1 && buf.lv0_packed;
                if (packed && lookup) hipLaunchKernelGGL((pyr_resize_direct_kernel<rb, true, true>), grid, dim3(256), 0, s, cfg, buf, l);
                else if (packed) hipLaunchKernelGGL((pyr_resize_direct_kernel<rb, true, false>), grid, dim3(256), 0, s, cfg, buf, l);
                else if (lookup) hipLaunchKernelGGL((pyr_resize_direct_kernel<rb, false, true>), grid, dim3(256), 0, s, cfg, buf, l);
                else hipLaunchKernelGGL((pyr_resize_direct_kernel<rb, false, false>), grid, dim3(256), 0, s, cfg, buf, l);
            }
        } else if (cfg.lv[l].rs_rw == 4) {
            dim3 grid((total_rows + 15) / 16, n_images);
            hipLaunchKernelGGL(pyr_resize_kernel<4>, grid, dim3(256), (size_t)span[0] * rowp, s, cfg, buf, l, src_words, span[0]);
        } else if (cfg.lv[l].rs_rw == 2) {
            dim3 grid((total_rows + 7) / 8, n_images);
            hipLaunchKernelGGL(pyr_resize_kernel<2>, grid, dim3(256), (size_t)span[1] * rowp, s, cfg, buf, l, src_words, span[1]);
        } else {
            dim3 grid((total_rows + 3) / 4, n_images);
            hipLaunchKernelGGL(pyr_resize_kernel<1>, grid, dim3(256), (size_t)span[2] * rowp, s, cfg, buf, l, src_words, span[2]);
        }
    }
    if (tail) {
        dim3 grid(cfg.tail_strips, n_images);
        if (cfg.tail_n == 3) hipLaunchKernelGGL(pyr_tail_kernel<3>, grid, dim3(TAIL_THREADS), (size_t)cfg.tail_lds_bytes, s, cfg, buf);
        else hipLaunchKernelGGL(pyr_tail_kernel<2>, grid, dim3(TAIL_THREADS), (size_t)cfg.tail_lds_bytes, s, cfg, buf);
    }
    return blurred;
}

// levels first_level .. nlevels - 1 (the lower ones were blurred beside the pyramid launches)
void orbfe_launch_blur(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, int first_level, hipStream_t s)
{
    if (first_level >= cfg.nlevels) return;
    const int t0 = cfg.lv[first_level].blur_tile_off, t1 = cfg.blur_tiles_total;
    if (t1 <= t0) return;
    dim3 grid((t1 - t0 + 3) / 4, n_images);
    hipLaunchKernelGGL(blur_kernel, grid, dim3(256), 0, s, cfg, buf, t0, t1);
}
