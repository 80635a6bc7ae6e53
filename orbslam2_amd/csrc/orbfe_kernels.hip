// orbfe_kernels.hip -- hand-written gfx950 kernels of the ORB front-end.
//
// Integer / bitwise HBM-bound work: no MFMA.  Wave = 64 lanes everywhere.
// Floating point follows contract Q4: compiled with -ffp-contract=off,
// every float product/sum below is individually rounded (IEEE), divisions are
// correctly rounded, cos/sin come from the shared deterministic routine.
//
// Reference routines replaced (paths under the reference tree):
//   pyr_resize_kernel   <- ORBextractor::ComputePyramid        src/ORBextractor.cc:921-946 (cv::resize INTER_LINEAR)
//   blur_kernel         <- cv::GaussianBlur 7x7 sigma 2        src/ORBextractor.cc:899-900
//   fast_cell_kernel    <- cell-wise cv::FAST, 2 thresholds    src/ORBextractor.cc:783-823
//   octree_kernel       <- ORBextractor::DistributeOctTree     src/ORBextractor.cc:533-757
//   describe_kernel     <- IC_Angle + computeOrbDescriptor     src/ORBextractor.cc:72-142,831-846,909-915
//   stereo_match_kernel <- Frame::ComputeStereoMatches         src/Frame.cc:464-626
//   stereo_median_kernel<- outlier cut                         src/Frame.cc:628-641
//   rgbd_kernel         <- Frame::ComputeStereoFromRGBD        src/Frame.cc:645-666
//   hamming_matrix_kernel <- ORBmatcher::DescriptorDistance    src/ORBmatcher.cc:1643-1659
#include "orbfe_device.h"
#include <cstdlib>

#define OT_THREADS 512
typedef short pk16 __attribute__((ext_vector_type(2))); // two int16 lanes in one VGPR (v_pk_* ops)
#define PYR_MX 4
#define PYR_MY 3

__device__ __attribute__((aligned(16))) const int8_t g_pattern[1024] = {
#include "orb_pattern_31.inc"
};

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Sum over the 64 lanes (all active) with DPP row operations: six v_add_u32_dpp and one v_readlane instead of six
// ds_bpermute round trips with their address arithmetic.  Quad xor 1, quad xor 2, row_half_mirror, row_mirror leave every
// 16-lane row holding its row sum; row_bcast:15 / row_bcast:31 then accumulate the rows into lane 63.
__device__ __forceinline__ int row_sum_i32(int v) // every lane of a 16-lane row gets the row's sum
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true); // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true); // row_mirror
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v)
{
    v = row_sum_i32(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned t = (unsigned)__shfl_xor((int)v, o, 64);
        v = t < v ? t : v;
    }
    return v;
}

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

// Exclusive scan of src[0..n) into dst[0..n) by the whole block (src may alias dst; LDS or
// global).  s_tmp: blockDim.x ints of LDS.  Returns the total.  All threads must call.
__device__ int block_excl_scan(const int *src, int *dst, int n, int *s_tmp)
{
    const int nt = blockDim.x, tid = threadIdx.x;
    const int per = (n + nt - 1) / nt;
    const int b = tid * per;
    const int e = (b + per < n) ? b + per : n;
    int sum = 0;
    for (int i = b; i < e; i++) sum += src[i];
    s_tmp[tid] = sum;
    __syncthreads();
    for (int off = 1; off < nt; off <<= 1) {
        int v = tid >= off ? s_tmp[tid - off] : 0;
        __syncthreads();
        s_tmp[tid] += v;
        __syncthreads();
    }
    const int total = s_tmp[nt - 1];
    int run = s_tmp[tid] - sum;
    for (int i = b; i < e; i++) {
        int v = src[i];
        dst[i] = run;
        run += v;
    }
    __syncthreads();
    return total;
}

// cv::fastAtan2 scalar path (see oracle/orb_oracle.c: orc_fast_atan2).
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float p1 = __uint_as_float(0x4265226fu);
    const float p3 = __uint_as_float(0xc19556eeu);
    const float p5 = __uint_as_float(0x410e9fbfu);
    const float p7 = __uint_as_float(0xc0228ad9u);
    const float eps = 2.220446049250313e-16f;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// Contract Q4 sin/cos (same algorithm as oracle orc_sincos_det): double reduction by pi/2 +
// fdlibm kernel polynomials, one rounding to float.
__device__ __forceinline__ void sincos_det(float rad, float *s, float *c)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00;
    const double PIO2_LO = 6.07710050650619224932e-11;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double x = (double)rad;
    double t = __dadd_rn(__dmul_rn(x, TWO_OVER_PI), 0.5);
    int q = (int)t;
    if (t < 0.0 && (double)q != t) q -= 1;
    double qd = (double)q;
    double r = __dsub_rn(__dsub_rn(x, __dmul_rn(qd, PIO2_HI)), __dmul_rn(qd, PIO2_LO));
    double z = __dmul_rn(r, r);
    double sp = __dadd_rn(S2, __dmul_rn(z, __dadd_rn(S3, __dmul_rn(z, __dadd_rn(S4, __dmul_rn(z, __dadd_rn(S5, __dmul_rn(z, S6))))))));
    double sr = __dadd_rn(r, __dmul_rn(__dmul_rn(z, r), __dadd_rn(S1, __dmul_rn(z, sp))));
    double cp = __dmul_rn(z, __dadd_rn(C1, __dmul_rn(z, __dadd_rn(C2, __dmul_rn(z, __dadd_rn(C3, __dmul_rn(z, __dadd_rn(C4, __dmul_rn(z, __dadd_rn(C5, __dmul_rn(z, C6)))))))))));
    double cr = __dsub_rn(1.0, __dsub_rn(__dmul_rn(0.5, z), __dmul_rn(z, cp)));
    double sv, cv;
    switch (q & 3) {
    case 0: sv = sr; cv = cr; break;
    case 1: sv = cr; cv = -sr; break;
    case 2: sv = -sr; cv = -cr; break;
    default: sv = -cr; cv = sr; break;
    }
    *s = (float)sv;
    *c = (float)cv;
}

__device__ __forceinline__ int hamming256(const uint32_t *a, const uint32_t *b)
{
    int d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d += __popc(a[i] ^ b[i]);
    return d;
}

// ---------------------------------------------------------------------------
// Pyramid storage.  Every level is stored with a reflect-101 margin (PYR_MX px on the left, at
// least 8 px on the right, PYR_MY rows above and below) written by its producer (ingest / resize),
// and pixel (0,0) sits on a 4-byte boundary.  Consumers can therefore use aligned 32-bit loads and
// need no border logic: the margin IS cv::GaussianBlur's BORDER_REFLECT_101 (src/ORBextractor.cc:900).
// ---------------------------------------------------------------------------

// ingest: packed images -> level 0 (+ margin); one thread = 4 px of the extended domain, block = 64 words x 4 rows.
// Interior words are one (unaligned) 32-bit load of the packed source; only the margin words gather reflected bytes.
// Also clears the image's status word (first kernel of every chain).
__global__ __launch_bounds__(256) void ingest_kernel(DeviceConfig cfg, DeviceBuffers buf, const uint8_t *__restrict__ src)
{
    const int img = blockIdx.z;
    const LevelInfo &L = cfg.lv[0];
    if (blockIdx.x == 0 && blockIdx.y == 0) { // first kernel of every chain: clear the image's status word and, for a right image, its pair's stereo row counters
        if (threadIdx.x == 0) buf.status[img] = 0;
        if (img & 1)
            for (int i = threadIdx.x; i < cfg.height; i += 256) buf.row_cnt[(size_t)(img >> 1) * cfg.height + i] = 0;
    }
    const int y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6)) - PYR_MY;
    const int x0 = (int)(blockIdx.x * 64 + (threadIdx.x & 63)) * 4 - PYR_MX;
    if (x0 >= L.w + 8 || y >= L.h + PYR_MY) return;
    const uint8_t *s = src + (size_t)img * L.w * L.h + (size_t)reflect101(y, L.h) * L.w;
    uint32_t v = 0;
    if (x0 >= 0 && x0 + 3 < L.w) {
        __builtin_memcpy(&v, s + x0, 4);
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) v |= (uint32_t)s[reflect101(x0 + j, L.w)] << (8 * j);
    }
    uint8_t *d = buf.pyr + (size_t)img * cfg.pyr_bytes + L.pyr_off + (ptrdiff_t)y * L.pitch + x0;
    *(uint32_t *)d = v;
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
    const int row_bytes = src_words * 4;
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
    // source row range of the block (every wave computes it: lanes < nrows hold one output row each)
    const uint32_t Yl = yt[y0 + (lane < nrows ? lane : 0)];
    int smin = (int)(Yl & 0xffffu), smax = (int)(Yl >> 16);
    { const int t = smin < smax ? smin : smax; smax = smin < smax ? smax : smin; smin = t; }
#pragma unroll
    for (int o = 1; o < 4 * RW; o <<= 1) {
        const int a = __shfl_xor(smin, o, 64), b = __shfl_xor(smax, o, 64);
        smin = a < smin ? a : smin; smax = b > smax ? b : smax;
    }
    smin = __builtin_amdgcn_readfirstlane(smin); smax = __builtin_amdgcn_readfirstlane(smax);
    int n_src = smax - smin + 1;
    if (n_src > max_src_rows) n_src = max_src_rows; // cannot happen: the host sized max_src_rows from the same table
    {
        int r = (int)(((float)tid + 0.5f) * (1.0f / (float)src_words)), c = tid - r * src_words;
        const int dr = 256 / src_words, dc = 256 - dr * src_words;
        const uint8_t *sp = src + (size_t)smin * S.pitch;
        const int nw = n_src * src_words;
        for (int i0 = tid; i0 < nw; i0 += 1024) { // 4 loads in flight per thread
            uint32_t v[4];
            int di[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                di[u] = __mul24(r, src_words) + c;
                if (i0 + 256 * u < nw) v[u] = *(const uint32_t *)(sp + (unsigned)(__mul24(r, S.pitch) + 4 * c));
                c += dc; r += dr;
                if (c >= src_words) { c -= src_words; r++; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (i0 + 256 * u < nw) ((uint32_t *)s_src)[di[u]] = v[u];
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

// ---------------------------------------------------------------------------
// Gaussian 7x7 (8.8 fixed point, separable), all levels in one launch.
// Register sliding window: a lane owns 4 adjacent columns and walks down BL_ROWS rows; per input
// row it loads three aligned words (12 px), forms the four 7-tap row sums with v_dot4_u32_u8, keeps
// the last seven row-sum vectors in registers and emits one 4-px output word.  No LDS, no barriers;
// HBM traffic = one read of the level (+6/BL_ROWS row halo, L2-served) and one write.
// ---------------------------------------------------------------------------
#define BL_ROWS 32  // rows per wave: 6 / BL_ROWS of the rows are loaded (and row-filtered) twice
#define BL_COLS 256 // per wave: 64 lanes x 4 px
__global__ __launch_bounds__(256) void blur_kernel(DeviceConfig cfg, DeviceBuffers buf)
{
    // the waves of a workgroup are independent: wave u of the flattened (level, row band, 256-column strip) list
    const int img = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int u = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (u >= cfg.blur_tiles_total) return;
    int level = 0;
    for (int l = 1; l < cfg.nlevels; l++)
        if (u >= cfg.lv[l].blur_tile_off) level = l;
    const LevelInfo &L = cfg.lv[level];
    const int t = u - L.blur_tile_off;
    const int x0 = (t % L.blur_tiles_x) * BL_COLS + lane * 4;
    const int r0 = (t / L.blur_tiles_x) * BL_ROWS;
    if (x0 >= L.w || r0 >= L.h) return;
    const uint8_t *src = buf.pyr + (size_t)img * cfg.pyr_bytes + L.pyr_off + x0;
    uint8_t *dst = buf.blur + (size_t)img * cfg.pyr_bytes + L.pyr_off + x0;
    const unsigned k_lo = (unsigned)cfg.taps[0] | ((unsigned)cfg.taps[1] << 8) | ((unsigned)cfg.taps[2] << 16) | ((unsigned)cfg.taps[3] << 24);
    const unsigned k_hi = (unsigned)cfg.taps[4] | ((unsigned)cfg.taps[5] << 8) | ((unsigned)cfg.taps[6] << 16);
    const unsigned k0 = cfg.taps[0], k1 = cfg.taps[1], k2 = cfg.taps[2], k3 = cfg.taps[3];
    const int y_max = L.h + PYR_MY - 1; // last materialised row
    unsigned H[7][4];
#pragma unroll
    for (int i = 0; i < BL_ROWS + 6; i++) {
        int y = r0 - 3 + i;
        y = y > y_max ? y_max : y; // rows past the margin only feed outputs that are never stored
        const uint32_t *row = (const uint32_t *)(src + (ptrdiff_t)y * L.pitch);
        const unsigned w0 = row[-1], w1 = row[0], w2 = row[1];
        unsigned hn[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned lo = j == 3 ? w1 : __builtin_amdgcn_alignbyte(w1, w0, j + 1);
            const unsigned hi = j == 3 ? w2 : __builtin_amdgcn_alignbyte(w2, w1, j + 1);
            hn[j] = __builtin_amdgcn_udot4(lo, k_lo, __builtin_amdgcn_udot4(hi, k_hi, 0u, false), false);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) H[i % 7][j] = hn[j];
        if (i >= 6) {
            const int yo = r0 + i - 6;
            if (yo < L.h) {
                // rows of the window in age order: oldest is slot (i+1)%7
                unsigned o = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const unsigned acc = k0 * (H[(i + 1) % 7][j] + H[i % 7][j]) + k1 * (H[(i + 2) % 7][j] + H[(i + 6) % 7][j]) +
                                         k2 * (H[(i + 3) % 7][j] + H[(i + 5) % 7][j]) + k3 * H[(i + 4) % 7][j];
                    o |= ((acc + 32768u) >> 16) << (8 * j);
                }
                *(uint32_t *)(dst + (ptrdiff_t)yo * L.pitch) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// FAST-9/16 per cell: score map + 3x3 NMS inside the cell + two-threshold select
// ---------------------------------------------------------------------------
__device__ __forceinline__ int fast_score16(const uint8_t *t, int pitch, int minth)
{
    const int v = t[0];
    int d[16];
    d[0] = v - t[3 * pitch];       d[1] = v - t[3 * pitch + 1];   d[2] = v - t[2 * pitch + 2];   d[3] = v - t[pitch + 3];
    d[4] = v - t[3];               d[5] = v - t[-pitch + 3];      d[6] = v - t[-2 * pitch + 2];  d[7] = v - t[-3 * pitch + 1];
    d[8] = v - t[-3 * pitch];      d[9] = v - t[-3 * pitch - 1];  d[10] = v - t[-2 * pitch - 2]; d[11] = v - t[-pitch - 3];
    d[12] = v - t[-3];             d[13] = v - t[pitch - 3];      d[14] = v - t[2 * pitch - 2];  d[15] = v - t[3 * pitch - 1];
    int mn2[16], mx2[16], mn4[16], mx4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { mn2[k] = min(d[k], d[(k + 1) & 15]); mx2[k] = max(d[k], d[(k + 1) & 15]); }
#pragma unroll
    for (int k = 0; k < 16; k++) { mn4[k] = min(mn2[k], mn2[(k + 2) & 15]); mx4[k] = max(mx2[k], mx2[(k + 2) & 15]); }
    int a = -512, b = 512;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int mn9 = min(min(mn4[k], mn4[(k + 4) & 15]), d[(k + 8) & 15]);
        const int mx9 = max(max(mx4[k], mx4[(k + 4) & 15]), d[(k + 8) & 15]);
        a = max(a, mn9);
        b = min(b, mx9);
    }
    return max(minth, max(a, -b)) - 1;
}

// One wave (64-thread workgroup) per FAST cell (src/ORBextractor.cc:783-810): no inter-wave barriers,
// and wave-ordered ballot compaction keeps every queue in row-major order, which is cv::FAST's
// emission order.  Phases:
//  0  aligned 32-bit loads of the (w+6)x(h+6) cell tile into LDS;
//  A  cheap necessary test on the 4 cardinal ring pixels (a 9-arc always holds two adjacent
//     cardinals)                                                        -> queue 1
//  B  OpenCV's 8-opposite-pairs necessary test on the full ring         -> queue 2
//  C  exact threshold-independent score (closed form of cornerScore<16>) for queue 2
//  D  strict 3x3 NMS inside the cell interior; iniThFAST, or minThFAST if that leaves nothing
//  E  ordered compaction of the survivors into the cell's slot.
__device__ __forceinline__ void load_ring(const uint8_t *t, int pitch, int d[16])
{
    const int v = t[0];
    d[0] = v - t[3 * pitch];       d[1] = v - t[3 * pitch + 1];   d[2] = v - t[2 * pitch + 2];   d[3] = v - t[pitch + 3];
    d[4] = v - t[3];               d[5] = v - t[-pitch + 3];      d[6] = v - t[-2 * pitch + 2];  d[7] = v - t[-3 * pitch + 1];
    d[8] = v - t[-3 * pitch];      d[9] = v - t[-3 * pitch - 1];  d[10] = v - t[-2 * pitch - 2]; d[11] = v - t[-pitch - 3];
    d[12] = v - t[-3];             d[13] = v - t[pitch - 3];      d[14] = v - t[2 * pitch - 2];  d[15] = v - t[3 * pitch - 1];
}

// number of set bits of m below this lane, plus acc (v_mbcnt_lo/hi accumulate form)
__device__ __forceinline__ unsigned mbcnt64(unsigned long long m, unsigned acc)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, acc));
}

// Two adjacent bytes P, P+1 of a 12-byte row window (w[0] | w[1] | w[2]), zero-extended into the two
// 16-bit halves of a register: one v_perm_b32 with a constant selector (0x0c selects 0x00).
template <int P> __device__ __forceinline__ pk16 row_pair(const unsigned (&w)[3])
{
    if constexpr (P + 1 <= 7)
        return __builtin_bit_cast(pk16, __builtin_amdgcn_perm(w[1], w[0], (unsigned)P | 0x0c00u | ((unsigned)(P + 1) << 16) | 0x0c000000u));
    else
        return __builtin_bit_cast(pk16, __builtin_amdgcn_perm(w[2], w[1], (unsigned)(P - 4) | 0x0c00u | ((unsigned)(P - 3) << 16) | 0x0c000000u));
}

// TP = tile pitch in bytes as a compile-time constant (0: run-time value): with it every ring / row offset folds
// into the immediate offset field of the LDS instructions instead of costing address VALU.
// BK: also accumulate the quadtree bucket counts / best keys of the survivors (orbfe_octree3.hip) -- aggregated per
// cell in LDS, then a few global atomics per cell.
// the waves of a workgroup are independent; a wave's own LDS traffic only needs its outstanding LDS operations retired
#define FAST_WAVE_SYNC() do { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); } while (0)
template <int TP, bool BK>
__global__ __launch_bounds__(256) void fast_cell_kernel(DeviceConfig cfg, DeviceBuffers buf, int n_images, int tile_pitch_rt, int tile_bytes, int sc_bytes, int q_bytes, int lds_per_wave, int dbg)
{
    const int tile_pitch = TP ? TP : tile_pitch_rt;
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem_all[];
    // XCD-aware block -> (image, cell) map (same scheme as describe_kernel): consecutive cells of one image run
    // on one XCD, so the 128-B lines that horizontally / vertically adjacent cell tiles share (a 37-row tile
    // uses ~44 B of each line) are served by that XCD's L2 instead of being re-fetched from HBM by 8 XCDs.
    // A workgroup is four independent waves = four consecutive cells (no workgroup barriers: FAST_WAVE_SYNC): horizontally
    // adjacent cells share the 128-B lines of their tile rows, and on one CU those lines are fetched from L2 once.
    const int bpi = (cfg.cells_total + 3) >> 2;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int img = (jb / bpi) * 8 + xcd;
    if (img >= n_images) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int cell = (jb % bpi) * 4 + wave;
    if (cell >= cfg.cells_total) return;
    uint8_t *s_mem = s_mem_all + wave * lds_per_wave;
    int level = 0;
    for (int l = 1; l < cfg.nlevels; l++)
        if (cell >= cfg.lv[l].cell_off) level = l;
    const LevelInfo &L = cfg.lv[level];
    const int ci = cell - L.cell_off;
    const int ci_i = ci / L.n_cols, ci_j = ci - ci_i * L.n_cols;
    const int lane = threadIdx.x & 63;
    int *cnt_out = buf.cell_cnt + (size_t)img * cfg.cells_total + cell;

    const int min_b = cfg.min_border;
    const int max_bx = L.w - cfg.edge_threshold + 3;
    const int max_by = L.h - cfg.edge_threshold + 3;
    const int ini_y = min_b + ci_i * L.h_cell;
    const int ini_x = min_b + ci_j * L.w_cell;
    int max_y = ini_y + L.h_cell + 6;
    int max_x = ini_x + L.w_cell + 6;
    if (ini_y >= max_by - 3 || ini_x >= max_bx - 6) { // src/ORBextractor.cc:788-798
        if (lane == 0) *cnt_out = 0;
        return;
    }
    if (max_y > max_by) max_y = max_by;
    if (max_x > max_bx) max_x = max_bx;
    const int tw = max_x - ini_x, th = max_y - ini_y;
    const int iw = tw - 6, ih = th - 6;
    if (iw <= 0 || ih <= 0) {
        if (lane == 0) *cnt_out = 0;
        return;
    }
    // bucket tables of this cell's columns / rows (BK): issued now, consumed in phase E
    unsigned tabx = 0u, taby = 0u;
    if (BK) {
        const uint32_t *bx_tab = buf.bk_tab + L.bk_xoff + 3 + ci_j * L.w_cell; // survivor x = c + 3 + j * wCell
        const uint32_t *by_tab = buf.bk_tab + L.bk_yoff + 3 + ci_i * L.h_cell;
        tabx = bx_tab[lane < iw ? lane : iw - 1];
        taby = by_tab[lane < ih ? lane : ih - 1];
    }
    // LDS layout (sizes fixed by the host from the largest cell): tile | scores | queue.  Queue 2 is
    // compacted in place over queue 1 (writes never pass the read cursor); the per-entry flags of
    // phase D reuse the tile, which is dead after phase C.  4.7 KB per wave keeps 32 waves per CU.
    uint8_t *s_tile = s_mem;                           // [th][tile_pitch], column 0 = pixel xa (4-aligned)
    uint8_t *s_sc = s_mem + tile_bytes;                // [(ih+2)][(iw+2)], zero border
    uint16_t *s_q1 = (uint16_t *)(s_sc + sc_bytes);    // packed (r << 8 | c), row-major ascending
    uint16_t *s_q2 = s_q1;
    uint8_t *s_qf = s_tile;                            // per queue-2 entry: 0 / 1 (local max, >= minTh) / 2 (>= iniTh)
    const int scp = iw + 2;

    const int xa = ini_x & ~3, ox = ini_x - xa;
    const int wpr = (max_x - xa + 3) >> 2; // words per tile row
    const uint8_t *src = buf.pyr + (size_t)img * cfg.pyr_bytes + L.pyr_off + (size_t)ini_y * L.pitch + xa;
    if (wpr <= 16) {
        // tile rows of at most 16 words (cells up to ~55 px): a wave-load covers 4 rows x 16 words; lanes beyond the row /
        // the last row repeat the last valid element (same value to the same LDS word), so nothing is predicated
        const int c4 = 4 * ((lane & 15) < wpr ? (lane & 15) : wpr - 1);
        const int rr = lane >> 4;
        const int nu = (th + 3) >> 2;
        for (int u0 = 0; u0 < nu; u0 += 4) { // 4 loads in flight per lane
            uint32_t v[4];
            int dst[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                int r = rr + 4 * (u0 + u);
                r = r < th ? r : th - 1;
                dst[u] = __mul24(r, tile_pitch) + c4;
                v[u] = *(const uint32_t *)(src + (unsigned)(__mul24(r, L.pitch) + c4));
            }
#pragma unroll
            for (int u = 0; u < 4; u++) *(uint32_t *)(s_tile + dst[u]) = v[u];
        }
    } else {
        // (row, word) of element i = lane, advanced by 64 per step; 32-bit offsets only (64-bit multiplies and a
        // per-element division cost more VALU issue than the copy itself)
        int r = (int)(((float)lane + 0.5f) * (1.0f / (float)wpr)), c = lane - r * wpr;
        const int dr = 64 / wpr, dc = 64 - dr * wpr;
        const int nw = th * wpr;
        for (int i0 = lane; i0 < nw; i0 += 256) {
            uint32_t v[4];
            int dst[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                dst[u] = __mul24(r, tile_pitch) + 4 * c;
                if (i0 + 64 * u < nw) v[u] = *(const uint32_t *)(src + (unsigned)(__mul24(r, L.pitch) + 4 * c));
                c += dc; r += dr;
                if (c >= wpr) { c -= wpr; r++; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (i0 + 64 * u < nw) *(uint32_t *)(s_tile + dst[u]) = v[u];
        }
    }
    for (int i = lane; i < sc_bytes / 4; i += 64) ((uint32_t *)s_sc)[i] = 0;
    FAST_WAVE_SYNC();
    const int t = cfg.min_th;
    if (dbg == 1) { if (lane == 0) *cnt_out = 0; return; }
    // Phases A and C work on TWO pixels per lane, one in each 16-bit half of a register, with packed
    // v_pk_{sub,min,max}_i16 (ring differences are in [-255, 255]): integer VALU issue is what bounds
    // this kernel (measured ~1 wave64 instruction / cycle / CU), so instructions are what is saved.
    int roff[16]; // ring offsets inside the LDS tile (uniform)
    roff[0] = 3 * tile_pitch;      roff[1] = 3 * tile_pitch + 1;   roff[2] = 2 * tile_pitch + 2;   roff[3] = tile_pitch + 3;
    roff[4] = 3;                   roff[5] = -tile_pitch + 3;      roff[6] = -2 * tile_pitch + 2;  roff[7] = -3 * tile_pitch + 1;
    roff[8] = -3 * tile_pitch;     roff[9] = -3 * tile_pitch - 1;  roff[10] = -2 * tile_pitch - 2; roff[11] = -tile_pitch - 3;
    roff[12] = -3;                 roff[13] = tile_pitch - 3;      roff[14] = 2 * tile_pitch - 2;  roff[15] = 3 * tile_pitch - 1;
    const pk16 tt = {(short)t, (short)t};
    // ---- A: cv::FAST's quick test on the 8 opposite ring pairs, for every interior pixel: a dark
    //      (bright) 9-arc needs one darker (brighter) pixel in every pair.  Passing BOTH polarities means
    //      every pair straddles the centre, which excludes any 9-arc, so such pixels are dropped;
    //      survivors are queued in row-major order with their polarity in bit 15.
    //      A lane owns FOUR horizontally adjacent pixels whose tile bytes are 3..6 of a 12-byte window
    //      (3 aligned LDS words per ring row, 21 per group): every ring column x-3..x+3 of the four pixels
    //      lies inside the window, so each ring position is two v_perm_b32 with constant selectors, and the
    //      test runs on raw ring values (with d = v - r: min_k max(d_k, d_k+8) = v - max_k min(r_k, r_k+8)). ----
    int n2 = 0;
    {
        const int ng = (iw + ox + 3) >> 2;       // groups per interior row; group j covers c = 4j - ox .. 4j - ox + 3
        const int G = ng * ih;
        const float rcp_ng = 1.0f / (float)ng;
        int rg = (int)(((float)lane + 0.5f) * rcp_ng), jg = lane - rg * ng;
        const int dr = 64 / ng, dj = 64 - dr * ng;
        const int pw = tile_pitch >> 2;
        for (int g0 = 0; g0 < G; g0 += 64) {
            const bool vg = g0 + lane < G;
            const uint32_t *tw4 = (const uint32_t *)s_tile + (vg ? __mul24(rg, pw) + jg : 0);
            unsigned w[7][3];
#pragma unroll
            for (int y = 0; y < 7; y++) {
#pragma unroll
                for (int i = 0; i < 3; i++) w[y][i] = tw4[y * pw + i];
            }
            pk16 A01, A23, B01, B23;
#define FAST_PAIR(first, ya, sa, yb, sb)                                                                          \
    {                                                                                                             \
        const pk16 a01 = row_pair<sa>(w[ya]), a23 = row_pair<sa + 2>(w[ya]);                                     \
        const pk16 b01 = row_pair<sb>(w[yb]), b23 = row_pair<sb + 2>(w[yb]);                                     \
        const pk16 n01 = __builtin_elementwise_min(a01, b01), x01 = __builtin_elementwise_max(a01, b01);          \
        const pk16 n23 = __builtin_elementwise_min(a23, b23), x23 = __builtin_elementwise_max(a23, b23);          \
        if (first) { A01 = n01; B01 = x01; A23 = n23; B23 = x23; }                                                \
        else {                                                                                                    \
            A01 = __builtin_elementwise_max(A01, n01); B01 = __builtin_elementwise_min(B01, x01);                 \
            A23 = __builtin_elementwise_max(A23, n23); B23 = __builtin_elementwise_min(B23, x23);                 \
        }                                                                                                         \
    }
            // ring pairs (k, k+8): (dx, dy) -> window start byte 3 + dx, tile row 3 + dy
            FAST_PAIR(true, 6, 3, 0, 3)   // ( 0, 3) / ( 0,-3)
            FAST_PAIR(false, 6, 4, 0, 2)  // ( 1, 3) / (-1,-3)
            FAST_PAIR(false, 5, 5, 1, 1)  // ( 2, 2) / (-2,-2)
            FAST_PAIR(false, 4, 6, 2, 0)  // ( 3, 1) / (-3,-1)
            FAST_PAIR(false, 3, 6, 3, 0)  // ( 3, 0) / (-3, 0)
            FAST_PAIR(false, 2, 6, 4, 0)  // ( 3,-1) / (-3, 1)
            FAST_PAIR(false, 1, 5, 5, 1)  // ( 2,-2) / (-2, 2)
            FAST_PAIR(false, 0, 4, 6, 2)  // ( 1,-3) / (-1, 3)
#undef FAST_PAIR
            const pk16 v01 = row_pair<3>(w[3]), v23 = row_pair<5>(w[3]);
            const pk16 lo01 = v01 - A01, hi01 = v01 - B01, lo23 = v23 - A23, hi23 = v23 - B23;
            const int c0 = 4 * jg - ox;
            const bool d0 = lo01.x > t, b0 = hi01.x < -t, d1 = lo01.y > t, b1 = hi01.y < -t;
            const bool d2 = lo23.x > t, b2 = hi23.x < -t, d3 = lo23.y > t, b3 = hi23.y < -t;
            const bool p0 = vg & (c0 >= 0) & (d0 != b0);
            const bool p1 = vg & (c0 + 1 >= 0) & (c0 + 1 < iw) & (d1 != b1);
            const bool p2 = vg & (c0 + 2 >= 0) & (c0 + 2 < iw) & (d2 != b2);
            const bool p3 = vg & (c0 + 3 < iw) & (d3 != b3);
            const unsigned long long m0 = __ballot(p0), m1 = __ballot(p1), m2 = __ballot(p2), m3 = __ballot(p3);
            int pos = n2 + (int)mbcnt64(m3, mbcnt64(m2, mbcnt64(m1, mbcnt64(m0, 0u))));
            const int e = (rg << 8) + c0; // c0 < 0 only for pixels that are never stored
            if (p0) s_q2[pos] = (uint16_t)(e | (b0 ? 0x8000 : 0));
            pos += p0;
            if (p1) s_q2[pos] = (uint16_t)((e + 1) | (b1 ? 0x8000 : 0));
            pos += p1;
            if (p2) s_q2[pos] = (uint16_t)((e + 2) | (b2 ? 0x8000 : 0));
            pos += p2;
            if (p3) s_q2[pos] = (uint16_t)((e + 3) | (b3 ? 0x8000 : 0));
            n2 += __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
            jg += dj; rg += dr;
            if (jg >= ng) { jg -= ng; rg++; }
        }
    }
    FAST_WAVE_SYNC();
    if (dbg == 3) { if (lane == 0) *cnt_out = 0; return; }
    // ---- C: exact score for the entry's polarity: max over the 16 arcs of the min of 9 (sign-normalised)
    //      differences; windows of 2, 4, 8 (+1) by doubling. ----
    for (int q0 = 0; q0 < n2; q0 += 128) {
        const int qa = q0 + lane, qb = q0 + 64 + lane;
        const bool va = qa < n2, vb = qb < n2;
        const unsigned ea = s_q2[va ? qa : 0], eb = s_q2[vb ? qb : 0];
        const int ra = (ea >> 8) & 127, ca = ea & 255, rb = (eb >> 8) & 127, cb = eb & 255;
        const uint8_t *pa = &s_tile[(ra + 3) * tile_pitch + ca + 3 + ox];
        const uint8_t *pb = &s_tile[(rb + 3) * tile_pitch + cb + 3 + ox];
        // sign-normalise by complementing bright entries (x -> -x-1 in both centre and ring keeps differences):
        // score = max_arcs min_arc (v' - r'_k) = v' - min_arcs max_arc r'_k, so the arcs run on raw ring values
        const pk16 sg = {(short)((ea & 0x8000u) ? -1 : 0), (short)((eb & 0x8000u) ? -1 : 0)};
        const pk16 vv = (pk16){(short)pa[0], (short)pb[0]} ^ sg;
        pk16 e[16];
#pragma unroll
        for (int k = 0; k < 16; k++) e[k] = (pk16){(short)pa[roff[k]], (short)pb[roff[k]]} ^ sg;
        pk16 m2[16], m4[16];
#pragma unroll
        for (int k = 0; k < 16; k++) m2[k] = __builtin_elementwise_max(e[k], e[(k + 1) & 15]);
#pragma unroll
        for (int k = 0; k < 16; k++) m4[k] = __builtin_elementwise_max(m2[k], m2[(k + 2) & 15]);
        pk16 worst = {512, 512};
#pragma unroll
        for (int k = 0; k < 16; k++)
            worst = __builtin_elementwise_min(worst, __builtin_elementwise_max(__builtin_elementwise_max(m4[k], m4[(k + 4) & 15]), e[(k + 8) & 15]));
        pk16 best = vv - worst;
        best = __builtin_elementwise_max(best, tt);
        const int sa = (int)best.x - 1, sb = (int)best.y - 1;
        if (va && sa >= t) s_sc[(ra + 1) * scp + ca + 1] = (uint8_t)sa;
        if (vb && sb >= t) s_sc[(rb + 1) * scp + cb + 1] = (uint8_t)sb;
    }
    FAST_WAVE_SYNC();
    if (dbg == 4) { if (lane == 0) *cnt_out = 0; return; }
    // ---- D: NMS + threshold choice.  The first 256 queue entries (all of them for ordinary cells) keep their flag and
    //      coordinates in registers for the compaction of phase E; later ones go through the LDS flag array. ----
    bool any = false;
    auto nms_flag = [&](unsigned rc) {
        const uint8_t *p = &s_sc[((rc >> 8) + 1) * scp + (rc & 255) + 1];
        const int s = p[0];
        // all nine reads issued together (a short-circuit chain would be nine dependent LDS round trips)
        const int n0 = p[-scp - 1], n1 = p[-scp], n2_ = p[-scp + 1], n3 = p[-1], n4 = p[1], n5 = p[scp - 1], n6 = p[scp], n7 = p[scp + 1];
        const int mx = max(max(max(n0, n1), max(n2_, n3)), max(max(n4, n5), max(n6, n7)));
        return ((s > 0) & (s > mx)) ? ((s >= cfg.ini_th) ? 2 : 1) : 0;
    };
    unsigned rcq[4];
    int fq[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int q = lane + 64 * k;
        rcq[k] = 0u; fq[k] = 0;
        if (q < n2) {
            rcq[k] = s_q2[q] & 0x7fffu;
            fq[k] = nms_flag(rcq[k]);
            any |= (fq[k] == 2);
        }
    }
    for (int q = lane + 256; q < n2; q += 64) {
        const int f = nms_flag(s_q2[q] & 0x7fffu);
        any |= (f == 2);
        s_qf[q] = (uint8_t)f;
    }
    const int need = __ballot(any) != 0ull ? 2 : 1;
    FAST_WAVE_SYNC();
    (void)q_bytes;
    // ---- E: ordered emission ----
    uint32_t *oxy = buf.cell_xy + ((size_t)img * cfg.cells_total + cell) * cfg.cell_cap;
    uint8_t *osc = buf.cell_sc + ((size_t)img * cfg.cells_total + cell) * cfg.cell_cap;
    // BK: the cell's survivors fall into the bucket columns gx0..gx1 and rows by0..by1 (tables are monotone); when
    // that rectangle has at most 64 buckets they are accumulated in LDS first
    unsigned *s_ac = (unsigned *)(s_mem + tile_bytes + sc_bytes + q_bytes), *s_ab = s_ac + 64;
    int gx0 = 0, by0 = 0, ncols = 1, nb = 0;
    uint32_t *g_cnt = nullptr, *g_best = nullptr;
    if (BK) {
        gx0 = (int)(__builtin_amdgcn_readfirstlane(tabx) >> 16);
        by0 = (int)(__builtin_amdgcn_readfirstlane(taby) >> 16);
        ncols = (int)(__builtin_amdgcn_readlane(tabx, 63) >> 16) - gx0 + 1; // lanes >= iw hold the last column / row
        nb = ncols * ((int)(__builtin_amdgcn_readlane(taby, 63) >> 16) - by0 + 1);
        if (nb <= 64) { s_ac[lane] = 0u; s_ab[lane] = 0u; }
        g_cnt = buf.bk_cnt + ((size_t)img * cfg.nlevels + level) * ORBFE_BK_BUCKETS;
        g_best = buf.bk_best + ((size_t)img * cfg.nlevels + level) * ORBFE_BK_BUCKETS;
        FAST_WAVE_SYNC();
    }
    // survivors are first compacted in place over the queue (a write never passes this iteration's reads), then
    // emitted densely: one pass of 64 lanes per 64 survivors instead of one per 64 queue entries
    int run = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (64 * k < n2) {
            const bool pred = fq[k] >= need; // fq = 0 beyond the queue
            const unsigned long long m = __ballot(pred);
            if (pred) s_q2[run + (int)mbcnt64(m, 0u)] = (uint16_t)rcq[k];
            run += __popcll(m);
        }
    }
    for (int q0 = 256; q0 < n2; q0 += 64) {
        const int q = q0 + lane;
        const bool pred = q < n2 && s_qf[q] >= need;
        const unsigned long long m = __ballot(pred);
        const uint16_t rc = q < n2 ? s_q2[q] : (uint16_t)0;
        if (pred) s_q2[run + (int)mbcnt64(m, 0u)] = rc;
        run += __popcll(m);
    }
    FAST_WAVE_SYNC();
    const int n_out = run < cfg.cell_cap ? run : cfg.cell_cap;
    for (int p0 = 0; p0 < n_out; p0 += 64) {
        const int pos = p0 + lane;
        const bool v = pos < n_out;
        const unsigned rc = v ? (s_q2[pos] & 0x7fffu) : 0u;
        const int r = rc >> 8, c = rc & 255;
        unsigned tx = 0u, ty = 0u;
        if (BK) { tx = (unsigned)__shfl((int)tabx, c, 64); ty = (unsigned)__shfl((int)taby, r, 64); }
        if (v) {
            // cell-local FAST coords (c+3, r+3) + (j*wCell, i*hCell): src/ORBextractor.cc:816-817
            const unsigned x = (unsigned)(c + 3 + ci_j * L.w_cell);
            const unsigned y = (unsigned)(r + 3 + ci_i * L.h_cell);
            const unsigned sc = s_sc[(r + 1) * scp + c + 1];
            oxy[pos] = x | (y << 16);
            osc[pos] = (uint8_t)sc;
            if (BK) {
                const unsigned key = ORBFE_BK_KEY(sc, (unsigned)ci, (unsigned)pos);
                if (nb <= 64) {
                    const int li = ((int)(ty >> 16) - by0) * ncols + ((int)(tx >> 16) - gx0);
                    atomicAdd(&s_ac[li], 1u);
                    atomicMax(&s_ab[li], key);
                } else {
                    const unsigned b = (tx | ty) & 0xfffu;
                    atomicAdd(&g_cnt[b], 1u);
                    atomicMax(&g_best[b], key);
                }
            }
        }
    }
    if (lane == 0) *cnt_out = run < cfg.cell_cap ? run : cfg.cell_cap;
    if (BK && nb <= 64 && dbg != 5) {
        FAST_WAVE_SYNC();
        const unsigned cnt = lane < nb ? s_ac[lane] : 0u;
        if (cnt) {
            const int ly = (int)(((float)lane + 0.5f) / (float)ncols), lx = lane - ly * ncols;
            const unsigned gx = (unsigned)(gx0 + lx), by = (unsigned)(by0 + ly);
            auto spread5 = [](unsigned v) { return (v & 1u) | ((v & 2u) << 1) | ((v & 4u) << 2) | ((v & 8u) << 3) | ((v & 16u) << 4); };
            const unsigned b = ((gx >> 5) << 10) | spread5(gx & 31u) | (spread5(by) << 1);
            atomicAdd(&g_cnt[b], cnt);
            atomicMax(&g_best[b], s_ab[lane]);
        }
    }
}

// ---------------------------------------------------------------------------
// DistributeOctTree, generic node-parallel kernel (any n_ini; fallback of orbfe_octree.hip): one workgroup per (image, level)
// ---------------------------------------------------------------------------
// Array formulation validated on the CPU by tests/octree_model.py:
//  * nodes live in an array kept in std::list order (front -> back);
//  * a pass splits a set of multi-point nodes; their non-empty children are written
//    n4,n3,n2,n1 at the front, blocks of later-processed parents nearer the front;
//  * every node owns a contiguous segment [beg, beg+cnt) of a candidate-index
//    permutation (two ping-pong buffers); a split is a stable 4-way partition of the
//    segment, so the per-node point order stays the FAST emission order;
//  * the "expand the biggest node first" phase sorts on (count desc, position asc),
//    which equals the reference's (size, pointer) ordering under contract Q3.
struct OtNodes {
    short *x0, *y0, *x1, *y1;
    int *beg, *cnt;
    uint8_t *bf;
};

__device__ __forceinline__ void ot_bind(OtNodes &n, uint8_t *&p, int cap)
{
    n.beg = (int *)p; p += sizeof(int) * cap;
    n.cnt = (int *)p; p += sizeof(int) * cap;
    n.x0 = (short *)p; p += sizeof(short) * cap;
    n.y0 = (short *)p; p += sizeof(short) * cap;
    n.x1 = (short *)p; p += sizeof(short) * cap;
    n.y1 = (short *)p; p += sizeof(short) * cap;
    n.bf = p; p += ((cap + 7) / 8) * 8;
}

__global__ __launch_bounds__(OT_THREADS) void octree_generic_kernel(DeviceConfig cfg, DeviceBuffers buf, int sort_cap)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_raw[];
    __shared__ int s_scan[OT_THREADS];
    __shared__ int s_n, s_total_k, s_nproc, s_nexpand, s_mode, s_done;
    const int level = blockIdx.x, img = blockIdx.y;
    const LevelInfo &L = cfg.lv[level];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = OT_THREADS / 64;
    const int MAXN = cfg.max_nodes;

    uint8_t *p = s_raw;
    unsigned long long *s_key = (unsigned long long *)p; p += sizeof(unsigned long long) * sort_cap;
    OtNodes A, B;
    ot_bind(A, p, MAXN);
    ot_bind(B, p, MAXN);
    int *s_ccnt = (int *)p; p += sizeof(int) * 4 * MAXN;   // child counts per old node
    int *s_rank = (int *)p; p += sizeof(int) * MAXN;       // processing rank of old node (-1: not processed)
    int *s_plist = (int *)p; p += sizeof(int) * MAXN;      // processing order -> old node
    int *s_kk = (int *)p; p += sizeof(int) * MAXN;         // scan scratch
    int *s_un = (int *)p; p += sizeof(int) * MAXN;         // scan scratch (unprocessed flags)

    const size_t ib = (size_t)img;
    int *cell_cnt = buf.cell_cnt + ib * cfg.cells_total + L.cell_off;
    int *cell_base = buf.cell_base + ib * cfg.cells_total + L.cell_off;
    const uint32_t *cell_xy = buf.cell_xy + (ib * cfg.cells_total + L.cell_off) * cfg.cell_cap;
    const uint8_t *cell_sc = buf.cell_sc + (ib * cfg.cells_total + L.cell_off) * cfg.cell_cap;
    uint32_t *cxy = buf.cand_xy + ib * cfg.cand_total + L.cand_off;
    uint8_t *csc = buf.cand_sc + ib * cfg.cand_total + L.cand_off;
    uint32_t *idx[2] = {buf.idx0 + ib * cfg.cand_total + L.cand_off, buf.idx1 + ib * cfg.cand_total + L.cand_off};
    int *sel_cnt = buf.sel_cnt + ib * cfg.nlevels + level;
    uint32_t *sel_xy = buf.sel_xy + ib * cfg.sel_total + L.sel_off;
    uint8_t *sel_sc = buf.sel_sc + ib * cfg.sel_total + L.sel_off;

    // ---- gather the per-cell candidates into emission order ----
    int nc = block_excl_scan(cell_cnt, cell_base, L.n_cells, s_scan);
    if (nc > L.cand_cap) { nc = L.cand_cap; if (tid == 0) buf.status[img] = 1; }
    if (tid == 0) buf.lvl_ncand[ib * cfg.nlevels + level] = nc;
    for (int i = tid; i < nc; i += OT_THREADS) {
        int lo = 0, hi = L.n_cells - 1; // last cell with cell_base <= i
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (cell_base[mid] <= i) lo = mid; else hi = mid - 1;
        }
        const int k = i - cell_base[lo];
        cxy[i] = cell_xy[(size_t)lo * cfg.cell_cap + k];
        csc[i] = cell_sc[(size_t)lo * cfg.cell_cap + k];
    }
    __syncthreads();
    if (nc == 0) {
        if (tid == 0) *sel_cnt = 0;
        return;
    }

    // ---- roots: stable partition by int(x / hX) (src/ORBextractor.cc:537-564) ----
    const int n_ini = L.n_ini;
    const int region_h = (L.h - cfg.edge_threshold + 3) - cfg.min_border;
    for (int i = tid; i < n_ini; i += OT_THREADS) s_kk[i] = 0;
    __syncthreads();
    for (int i = tid; i < nc; i += OT_THREADS) {
        int b = (int)__fdiv_rn((float)(cxy[i] & 0xffffu), L.hx);
        b = b < 0 ? 0 : (b >= n_ini ? n_ini - 1 : b);
        atomicAdd(&s_kk[b], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0, n = 0;
        for (int b = 0; b < n_ini; b++) {
            const int c = s_kk[b];
            s_un[b] = run; // segment begin of bucket b
            if (c > 0) {
                A.x0[n] = (short)(int)__fmul_rn(L.hx, (float)b);
                A.x1[n] = (short)(int)__fmul_rn(L.hx, (float)(b + 1));
                A.y0[n] = 0;
                A.y1[n] = (short)region_h;
                A.beg[n] = run; A.cnt[n] = c; A.bf[n] = 0;
                n++;
            }
            run += c;
        }
        s_n = n;
        s_done = 0;
    }
    __syncthreads();
    for (int b = wave; b < n_ini; b += nwaves) {
        int run = s_un[b];
        for (int i0 = 0; i0 < nc; i0 += 64) {
            const int i = i0 + lane;
            bool pred = false;
            if (i < nc) {
                int bb = (int)__fdiv_rn((float)(cxy[i] & 0xffffu), L.hx);
                bb = bb < 0 ? 0 : (bb >= n_ini ? n_ini - 1 : bb);
                pred = (bb == b);
            }
            const unsigned long long m = __ballot(pred);
            if (pred) idx[0][run + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)i;
            run += __popcll(m);
        }
    }
    __syncthreads();

    // ---- split passes ----
    OtNodes cur = A, nxt = B;
    int sorted_phase = 0;
    for (int iter = 0; iter < 100000; iter++) { // n grows every pass, so this ends at n >= quota at the latest
        const int n = s_n;
        // (A) child counts of every multi-point node
        for (int i = wave; i < n; i += nwaves) {
            const int cnt = cur.cnt[i];
            if (cnt > 1) {
                const int mx = cur.x0[i] + ((cur.x1[i] - cur.x0[i] + 1) >> 1);
                const int my = cur.y0[i] + ((cur.y1[i] - cur.y0[i] + 1) >> 1);
                const uint32_t *src = idx[cur.bf[i]] + cur.beg[i];
                int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
                for (int j = lane; j < cnt; j += 64) {
                    const uint32_t xy = cxy[src[j]];
                    const int cls = ((int)(xy & 0xffffu) < mx ? 0 : 1) + ((int)(xy >> 16) < my ? 0 : 2);
                    c0 += cls == 0; c1 += cls == 1; c2 += cls == 2; c3 += cls == 3;
                }
                c0 = wave_sum_i32(c0); c1 = wave_sum_i32(c1); c2 = wave_sum_i32(c2); c3 = wave_sum_i32(c3);
                if (lane == 0) { s_ccnt[4 * i] = c0; s_ccnt[4 * i + 1] = c1; s_ccnt[4 * i + 2] = c2; s_ccnt[4 * i + 3] = c3; }
            }
        }
        for (int i = tid; i < n; i += OT_THREADS) { s_rank[i] = -1; s_kk[i] = cur.cnt[i] > 1 ? 1 : 0; }
        __syncthreads();
        // (B) processing order
        int m;
        if (!sorted_phase) {
            m = block_excl_scan(s_kk, s_kk, n, s_scan); // s_kk[i] = rank among multi nodes
            for (int i = tid; i < n; i += OT_THREADS)
                if (cur.cnt[i] > 1) s_plist[s_kk[i]] = i;
            __syncthreads();
        } else {
            m = block_excl_scan(s_kk, s_kk, n, s_scan);
            int P = 1;
            while (P < m) P <<= 1;
            for (int i = tid; i < P; i += OT_THREADS) s_key[i] = ~0ull;
            __syncthreads();
            for (int i = tid; i < n; i += OT_THREADS)
                if (cur.cnt[i] > 1)
                    s_key[s_kk[i]] = ((unsigned long long)(0xffffffffu - (unsigned)cur.cnt[i]) << 32) | (unsigned)i;
            __syncthreads();
            for (int k = 2; k <= P; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = tid; i < P; i += OT_THREADS) {
                        const int ixj = i ^ j;
                        if (ixj > i) {
                            const unsigned long long a = s_key[i], b = s_key[ixj];
                            const bool up = ((i & k) == 0);
                            if ((a > b) == up) { s_key[i] = b; s_key[ixj] = a; }
                        }
                    }
                    __syncthreads();
                }
            }
            for (int i = tid; i < m; i += OT_THREADS) s_plist[i] = (int)(s_key[i] & 0xffffffffu);
            __syncthreads();
        }
        // k (non-empty children) per processing rank; inclusive prefix decides the stop
        for (int r = tid; r < m; r += OT_THREADS) {
            const int i = s_plist[r];
            s_kk[r] = (s_ccnt[4 * i] > 0) + (s_ccnt[4 * i + 1] > 0) + (s_ccnt[4 * i + 2] > 0) + (s_ccnt[4 * i + 3] > 0);
        }
        __syncthreads();
        block_excl_scan(s_kk, s_un, m, s_scan); // s_un[r] = sum of k over ranks < r
        if (tid == 0) {
            int nproc = m;
            if (sorted_phase) {
                // first r with n + sum_{r'<=r}(k-1) >= quota (src/ORBextractor.cc:724-725)
                nproc = m;
                for (int r = 0; r < m; r++) {
                    const int incl = s_un[r] + s_kk[r];
                    if (n + incl - (r + 1) >= L.quota) { nproc = r + 1; break; }
                }
            }
            s_nproc = nproc;
            s_total_k = nproc > 0 ? s_un[nproc - 1] + s_kk[nproc - 1] : 0;
        }
        __syncthreads();
        const int nproc = s_nproc, total_k = s_total_k;
        for (int r = tid; r < nproc; r += OT_THREADS) s_rank[s_plist[r]] = r;
        __syncthreads();
        // unprocessed old nodes keep their relative order behind the new blocks
        for (int i = tid; i < n; i += OT_THREADS) s_plist[i] = (s_rank[i] < 0) ? 1 : 0; // reuse as flag array
        __syncthreads();
        const int n_un = block_excl_scan(s_plist, s_plist, n, s_scan);
        const int n_new = total_k + n_un;
        if (n_new > MAXN) { // cannot happen for max_nodes >= max(quota+3, 4*n_ini); guard anyway
            if (tid == 0) { buf.status[img] = 2; *sel_cnt = 0; }
            return;
        }
        if (tid == 0) s_nexpand = 0;
        __syncthreads();
        // (D) emit new node array + scatter the points of processed nodes
        for (int i = wave; i < n; i += nwaves) {
            const int r = s_rank[i];
            if (r < 0) {
                if (lane == 0) {
                    const int q = total_k + s_plist[i];
                    nxt.x0[q] = cur.x0[i]; nxt.y0[q] = cur.y0[i]; nxt.x1[q] = cur.x1[i]; nxt.y1[q] = cur.y1[i];
                    nxt.beg[q] = cur.beg[i]; nxt.cnt[q] = cur.cnt[i]; nxt.bf[q] = cur.bf[i];
                }
                continue;
            }
            const int x0 = cur.x0[i], y0 = cur.y0[i], x1 = cur.x1[i], y1 = cur.y1[i];
            const int mx = x0 + ((x1 - x0 + 1) >> 1);
            const int my = y0 + ((y1 - y0 + 1) >> 1);
            const int cnt = cur.cnt[i], beg = cur.beg[i], sb = cur.bf[i];
            const int c0 = s_ccnt[4 * i], c1 = s_ccnt[4 * i + 1], c2 = s_ccnt[4 * i + 2], c3 = s_ccnt[4 * i + 3];
            const int k = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
            // block of this parent starts after the blocks of all later-processed parents
            int q = total_k - (s_un[r] + k);
            if (lane == 0) {
                int nexp = 0;
                const int b0 = beg, b1 = beg + c0, b2 = b1 + c1, b3 = b2 + c2;
                if (c3 > 0) { nxt.x0[q] = (short)mx; nxt.y0[q] = (short)my; nxt.x1[q] = (short)x1; nxt.y1[q] = (short)y1; nxt.beg[q] = b3; nxt.cnt[q] = c3; nxt.bf[q] = (uint8_t)(1 - sb); q++; nexp += c3 > 1; }
                if (c2 > 0) { nxt.x0[q] = (short)x0; nxt.y0[q] = (short)my; nxt.x1[q] = (short)mx; nxt.y1[q] = (short)y1; nxt.beg[q] = b2; nxt.cnt[q] = c2; nxt.bf[q] = (uint8_t)(1 - sb); q++; nexp += c2 > 1; }
                if (c1 > 0) { nxt.x0[q] = (short)mx; nxt.y0[q] = (short)y0; nxt.x1[q] = (short)x1; nxt.y1[q] = (short)my; nxt.beg[q] = b1; nxt.cnt[q] = c1; nxt.bf[q] = (uint8_t)(1 - sb); q++; nexp += c1 > 1; }
                if (c0 > 0) { nxt.x0[q] = (short)x0; nxt.y0[q] = (short)y0; nxt.x1[q] = (short)mx; nxt.y1[q] = (short)my; nxt.beg[q] = b0; nxt.cnt[q] = c0; nxt.bf[q] = (uint8_t)(1 - sb); q++; nexp += c0 > 1; }
                if (nexp) atomicAdd(&s_nexpand, nexp);
            }
            const uint32_t *src = idx[sb] + beg;
            uint32_t *dst = idx[1 - sb];
            int r0 = beg, r1 = beg + c0, r2 = r1 + c1, r3 = r2 + c2;
            const unsigned long long lt = (1ull << lane) - 1ull;
            for (int j0 = 0; j0 < cnt; j0 += 64) {
                const int j = j0 + lane;
                int cls = -1;
                uint32_t id = 0;
                if (j < cnt) {
                    id = src[j];
                    const uint32_t xy = cxy[id];
                    cls = ((int)(xy & 0xffffu) < mx ? 0 : 1) + ((int)(xy >> 16) < my ? 0 : 2);
                }
                const unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1),
                                         m2 = __ballot(cls == 2), m3 = __ballot(cls == 3);
                if (cls == 0) dst[r0 + __popcll(m0 & lt)] = id;
                else if (cls == 1) dst[r1 + __popcll(m1 & lt)] = id;
                else if (cls == 2) dst[r2 + __popcll(m2 & lt)] = id;
                else if (cls == 3) dst[r3 + __popcll(m3 & lt)] = id;
                r0 += __popcll(m0); r1 += __popcll(m1); r2 += __popcll(m2); r3 += __popcll(m3);
            }
        }
        __syncthreads();
        // (E) stop logic (src/ORBextractor.cc:661-731)
        if (tid == 0) {
            const int prev = n;
            s_n = n_new;
            if (n_new >= L.quota || n_new == prev) s_done = 1;
            else if (!sorted_phase && n_new + 3 * s_nexpand > L.quota) s_mode = 1;
            else s_mode = sorted_phase;
        }
        __syncthreads();
        { OtNodes t = cur; cur = nxt; nxt = t; }
        if (s_done) break;
        sorted_phase = s_mode;
        __syncthreads();
    }

    // ---- keep the best response per node, first wins (src/ORBextractor.cc:735-754) ----
    const int n = s_n;
    int n_out = n < L.sel_cap ? n : L.sel_cap;
    if (n > L.sel_cap && tid == 0) buf.status[img] = 3;
    for (int i = wave; i < n_out; i += nwaves) {
        const uint32_t *src = idx[cur.bf[i]] + cur.beg[i];
        const int cnt = cur.cnt[i];
        unsigned best = 0xffffffffu; // (255-score)<<24 | position in node  (cnt < 2^24)
        for (int j = lane; j < cnt; j += 64) {
            const unsigned key = ((unsigned)(255 - csc[src[j]]) << 24) | (unsigned)j;
            best = key < best ? key : best;
        }
        best = wave_min_u32(best);
        if (lane == 0) {
            const uint32_t id = src[best & 0xffffffu];
            sel_xy[i] = cxy[id];
            sel_sc[i] = csc[id];
        }
    }
    if (tid == 0) *sel_cnt = n_out;
}

// ---------------------------------------------------------------------------
// orientation + descriptor + final keypoint record: one wave per keypoint slot
// ---------------------------------------------------------------------------
// The kernel is bound by vector-memory INSTRUCTION issue (a wave64 byte gather costs the texture
// addresser ~16 cycles whatever it fetches), so everything is fetched as aligned dwords into LDS:
// the block stages the two tables once, each wave stages a keypoint's 31-row raw patch (5 loads) and
// 37-row blurred patch (6 loads), and all per-pixel / per-sample accesses become LDS byte reads.
// A wave handles DS_KPW consecutive keypoint slots: the table staging and the slot bookkeeping are paid
// once per 4 * DS_KPW keypoints, and the patch words of keypoint i+1 are fetched into registers while
// keypoint i is computed from LDS, so the global-load latency is off the critical path.
#define DS_PATCH_W 40 // bytes per staged patch row (10 words: covers 31+3 / 37+3 px at any alignment)
#define DS_KPW 4      // keypoint slots per wave
#define DS_RAW_REGS 5 // prefetch registers for the raw patch (raw_rows * 10 words <= 320, i.e. half_patch <= 15)
#define DS_BLR_REGS 6 // 37 rows * 10 words = 370 words


__global__ __launch_bounds__(256, 6) void describe_kernel(DeviceConfig cfg, DeviceBuffers buf, int n_images, int stereo, int dbg)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dm[];
    // XCD-aware block -> (image, block) map: workgroups are dealt round-robin over the 8 XCDs, so block
    // b runs on XCD b % 8 (placement is a speed assumption only).  All blocks of one image are given to
    // one XCD, whose 4 MiB L2 then holds that image's raw + blurred pyramid (3.3 MB) while its ~2000
    // overlapping 31x31 / 37x37 patches are read, instead of every patch row coming from the MALL.
    const int bpi = (cfg.sel_total + 4 * DS_KPW - 1) / (4 * DS_KPW); // blocks per image
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int img = (jb / bpi) * 8 + xcd;
    if (img >= n_images) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform values must be provably so: they feed scalar addresses
    const int slot0 = (jb % bpi) * (4 * DS_KPW) + wave * DS_KPW;
    const int hp = cfg.half_patch;
    const int raw_rows = 2 * hp + 1;
    const int raw_words = raw_rows * (DS_PATCH_W / 4);
    const int *sel_cnt = buf.sel_cnt + (size_t)img * cfg.nlevels;
    if (jb % bpi == 0 && tid == 0) {
        int tot = 0;
        for (int l = 0; l < cfg.nlevels; l++) tot += sel_cnt[l];
        buf.kp_cnt[img] = tot;
    }
    // block-shared tables: patch offsets (patch_n shorts) | pattern (256 words)
    int16_t *s_uv = (int16_t *)s_dm;
    int *s_pat = (int *)(s_dm + ((cfg.patch_n * 2 + 15) & ~15));
    uint8_t *s_raw = (uint8_t *)(s_pat + 256) + wave * ((raw_rows + 37) * DS_PATCH_W);
    uint8_t *s_blr = s_raw + raw_rows * DS_PATCH_W;
    for (int i = tid; i < cfg.patch_n / 2; i += 256) ((int *)s_uv)[i] = ((const int *)buf.patch_uv)[i];
    s_pat[tid] = ((const int *)g_pattern)[tid];
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
    if (dbg == 1) return;

    const int r0 = lane / (DS_PATCH_W / 4), c0 = lane - r0 * (DS_PATCH_W / 4);
    const bool raw_in_regs = raw_words <= 64 * DS_RAW_REGS;
    uint32_t pr[DS_RAW_REGS], pb[DS_BLR_REGS];
    // Word i = lane + 64*k of a staged patch is (row i / 10, word i % 10).  The (row, 4 * word) pairs of this lane's words are
    // computed once per wave and packed two per register (raw | blurred << 16, each row | 4 * word << 8); a patch word is then
    // one mad away from a wave-uniform base address (scalar registers), instead of 64-bit per-lane pointer stepping with
    // per-load predication -- that bookkeeping used to be 40 % of the kernel's VALU instructions.  Words past the end of a
    // patch repeat its last word (same value to the same LDS slot), so nothing is predicated.
    const int blr_words = 37 * (DS_PATCH_W / 4);
    uint32_t wtab[DS_BLR_REGS];
#pragma unroll
    for (int k = 0; k < DS_BLR_REGS; k++) {
        const int i = lane + 64 * k;
        const int ir = i < raw_words ? i : raw_words - 1, ib = i < blr_words ? i : blr_words - 1;
        const int rr = (ir * 6554) >> 16, rb = (ib * 6554) >> 16; // / 10 for i < 16384
        wtab[k] = (uint32_t)(rr | ((4 * (ir - 10 * rr)) << 8)) | ((uint32_t)(rb | ((4 * (ib - 10 * rb)) << 8)) << 16);
    }
    const int lds_last_raw = 4 * (lane + 64 * (DS_RAW_REGS - 1) < raw_words ? lane + 64 * (DS_RAW_REGS - 1) : raw_words - 1);
    const int lds_last_blr = 4 * (lane + 64 * (DS_BLR_REGS - 1) < blr_words ? lane + 64 * (DS_BLR_REGS - 1) : blr_words - 1);
    auto fetch_raw = [&](const uint8_t *base /* uniform: patch origin */, int pitch) {
#pragma unroll
        for (int k = 0; k < DS_RAW_REGS; k++) {
            const unsigned e = wtab[k] & 0xffffu;
            pr[k] = *(const uint32_t *)(base + ((unsigned)__mul24(e & 0xffu, pitch) + (e >> 8))); // one 32-bit offset: saddr + voffset
        }
    };
    auto fetch_blr = [&](const uint8_t *base, int pitch) {
#pragma unroll
        for (int k = 0; k < DS_BLR_REGS; k++) {
            const unsigned e = wtab[k] >> 16;
            pb[k] = *(const uint32_t *)(base + ((unsigned)__mul24(e & 0xffu, pitch) + (e >> 8)));
        }
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
        const LevelInfo &L = cfg.lv[level];
        if (raw_in_regs)
            fetch_raw(buf.pyr + (size_t)img * cfg.pyr_bytes + L.pyr_off + (ptrdiff_t)(cy - hp) * L.pitch + ((cx - hp) & ~3), L.pitch);
        return true;
    };
    auto prefetch_blur = [&](int i) -> bool {
        if (i >= DS_KPW || !slot_data(i)) return false;
        const LevelInfo &L = cfg.lv[level];
        fetch_blr(buf.blur + (size_t)img * cfg.pyr_bytes + L.pyr_off + (ptrdiff_t)(cy - 18) * L.pitch + ((cx - 18) & ~3), L.pitch);
        return true;
    };

    // Pass 1: raw patches -> IC_Angle moments of the wave's keypoints (lane i keeps keypoint i's);
    // then fastAtan2 and the sin / cos ONCE for all of them (lane i computes keypoint i's: those ~150 scalar-like
    // instructions, part of them double precision, would otherwise be repeated per keypoint by all 64 lanes);
    // pass 2: blurred patches -> descriptors and keypoint records.
    int m10_l = 0, m01_l = 0;
    bool have = prefetch_raw(0);
    for (int i = 0; i < DS_KPW; i++) {
        const bool cur = have;
        int kx = 0, ky = 0, lv = 0;
        if (cur) {
            slot_data(i);
            kx = cx; ky = cy; lv = level;
            if (raw_in_regs) {
#pragma unroll
                for (int k = 0; k < DS_RAW_REGS - 1; k++) ((uint32_t *)s_raw)[lane + 64 * k] = pr[k];
                *(uint32_t *)(s_raw + lds_last_raw) = pr[DS_RAW_REGS - 1];
            }
            if (!raw_in_regs) { // big patches: straight through (no prefetch)
                const LevelInfo &Lr = cfg.lv[lv];
                const uint8_t *gp = buf.pyr + (size_t)img * cfg.pyr_bytes + Lr.pyr_off + (ptrdiff_t)__mul24(ky - hp + r0, Lr.pitch) + ((kx - hp) & ~3) + 4 * c0;
                const int step = 6 * Lr.pitch + 16, wrap = Lr.pitch - DS_PATCH_W;
                int c = c0;
                for (int w = lane; w < raw_words; w += 64) {
                    ((uint32_t *)s_raw)[w] = *(const uint32_t *)gp;
                    gp += step; c += 4;
                    if (c >= DS_PATCH_W / 4) { c -= DS_PATCH_W / 4; gp += wrap; }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0); // LDS writes of this wave are visible to its own later reads in order
        __builtin_amdgcn_wave_barrier();
        have = prefetch_raw(i + 1); // in flight while keypoint i is computed
        if (!cur || dbg == 2) continue;
        // IC_Angle (src/ORBextractor.cc:72-99): integer moments over the circular patch (host-built offset
        // table, padded with (0,0) entries that contribute nothing)
        const int xr = (kx - hp) & ~3;
        int m10 = 0, m01 = 0;
        const uint8_t *pc = s_raw + hp * DS_PATCH_W + (kx - xr);
        for (int kk = lane; kk < cfg.patch_n; kk += 64) {
            const int uv = s_uv[kk];
            const int u = (int)(int8_t)(uv & 0xff), v = (int)(int8_t)((uv >> 8) & 0xff);
            const int I = pc[__mul24(v, DS_PATCH_W) + u];
            m10 += u * I;
            m01 += v * I;
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
    float rl_x = 0.f, rl_y = 0.f;
    for (int i = 0; i < DS_KPW; i++) {
        const bool cur = have;
        int lv = 0, kx = 0, ky = 0, ksc = 0, kout = 0;
        if (cur) {
            slot_data(i);
            lv = level; kx = cx; ky = cy; ksc = score; kout = out;
#pragma unroll
            for (int k = 0; k < DS_BLR_REGS - 1; k++) ((uint32_t *)s_blr)[lane + 64 * k] = pb[k];
            *(uint32_t *)(s_blr + lds_last_blr) = pb[DS_BLR_REGS - 1];
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        have = prefetch_blur(i + 1);
        if (!cur || dbg == 2) continue;
        const LevelInfo &L = cfg.lv[lv];
        const int xb = (kx - 18) & ~3;
        const float angle = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(angle_l), i));
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_l), i));
        const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b_l), i));
        if (dbg == 3) { if (lane == 0) buf.depth[(size_t)img * cfg.sel_total + kout] = angle; continue; }

        // computeOrbDescriptor (src/ORBextractor.cc:103-142)
        const uint8_t *center = s_blr + 18 * DS_PATCH_W + (kx - xb);
        unsigned long long *dout = (unsigned long long *)(buf.desc + ((size_t)img * cfg.sel_total + kout) * 32);
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
        if (lane < 4) dout[lane] = lane == 0 ? bits[0] : (lane == 1 ? bits[1] : (lane == 2 ? bits[2] : bits[3]));
        float px = (float)kx, py = (float)ky;
        if (lv != 0) { px = __fmul_rn(px, L.scale); py = __fmul_rn(py, L.scale); }
        if (lane == i) { rl_lv = lv | (kout << 8); rl_x = px; rl_y = py; } // for the stereo row lists below
        if (lane == 0) {
            KeyPointPOD kp;
            kp.x = px; kp.y = py;
            kp.size = (float)L.scaled_patch;
            kp.angle = angle;
            kp.response = (float)ksc;
            kp.octave = lv;
            kp.class_id = -1;
            ((KeyPointPOD *)buf.kps)[(size_t)img * cfg.sel_total + kout] = kp;
        }
    }
    if (stereo && (img & 1)) {
        // right image of a pair: list each keypoint in the rows its band covers (vRowIndices, src/Frame.cc:474-491: rows
        // floor(y - r) .. ceil(y + r), r = 2 * scale[octave]); the order inside a row list is irrelevant to
        // stereo_match_kernel's arg-min.  All of the wave's atomics are issued before the first dependent store.
        int *rcnt = buf.row_cnt + (size_t)(img >> 1) * cfg.height;
        uint2 *rent = buf.row_ent + (size_t)(img >> 1) * cfg.height * cfg.row_cap;
        int pos[DS_KPW], yy[DS_KPW];
        uint2 e[DS_KPW];
#pragma unroll
        for (int i = 0; i < DS_KPW; i++) {
            const int lvk = __builtin_amdgcn_readlane(rl_lv, i);
            const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rl_x), i));
            const float y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rl_y), i));
            pos[i] = -1; yy[i] = 0;
            e[i].x = (uint32_t)(lvk >> 8) | ((uint32_t)(lvk & 255) << 16); e[i].y = __float_as_uint(x);
            if (lvk >= 0) {
                const float r = __fmul_rn(2.0f, cfg.lv[lvk & 255].scale);
                int maxr = (int)ceilf(__fadd_rn(y, r)), minr = (int)floorf(__fsub_rn(y, r));
                minr = minr < 0 ? 0 : minr; maxr = maxr > cfg.height - 1 ? cfg.height - 1 : maxr;
                yy[i] = minr + lane;
                if (yy[i] <= maxr) pos[i] = atomicAdd(&rcnt[yy[i]], 1);
                for (int y2 = yy[i] + 64; y2 <= maxr; y2 += 64) { // bands taller than a wave (large scale factors only)
                    const int p2 = atomicAdd(&rcnt[y2], 1);
                    if (p2 < cfg.row_cap) rent[(size_t)y2 * cfg.row_cap + p2] = e[i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < DS_KPW; i++)
            if (pos[i] >= 0 && pos[i] < cfg.row_cap) rent[(size_t)yy[i] * cfg.row_cap + pos[i]] = e[i];
    }
}

// ---------------------------------------------------------------------------
// stereo: one wave per left keypoint (coarse Hamming band search + SAD + parabola)
// ---------------------------------------------------------------------------
// One 16-lane group per left keypoint (four keypoints per wave, sixteen per workgroup): a row list holds a few
// dozen candidates of which ~10 pass the octave / disparity filter, so a whole wave per keypoint idles most lanes
// and, with ~7 dependent global round trips per keypoint, needs 4x the waves to hide the same latency.
//   coarse search: lanes stride the row list; arg-min key dist << 16 | iR (= the reference's first minimum);
//   SAD: lane handles window pixels p = gl, gl + 16, .. < 121; the 11 shifted right-image bytes of a pixel are
//        12 contiguous bytes = three unaligned dword loads; sums reduced over the group by xor shuffles.
#define SM_G 16
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

__global__ __launch_bounds__(256) void stereo_match_kernel(DeviceConfig cfg, DeviceBuffers buf, int n_pairs)
{
    // XCD-aware block -> (pair, block) map: all blocks of a pair on one XCD (its L2 then holds the pair's
    // descriptors, keypoints and the pyramid rows the SAD windows touch)
    const int kpb = 256 / SM_G;
    const int bpp = (cfg.sel_total + kpb - 1) / kpb;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int pair = (jb / bpp) * 8 + xcd;
    if (pair >= n_pairs) return;
    const int imgL = 2 * pair, imgR = 2 * pair + 1;
    const int gl = threadIdx.x & (SM_G - 1);
    const int iL = (jb % bpp) * kpb + (threadIdx.x / SM_G);
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
    // candidates = right keypoints whose row band covers int(vL) (vRowIndices[vL], src/Frame.cc:513), listed per row by
    // describe_kernel; the arg-min key (dist << 16 | iR) makes the result independent of the order inside a row list
    int cnt = 0;
    if (row >= 0 && row < cfg.height) cnt = buf.row_cnt[(size_t)pair * cfg.height + row];
    if (cnt <= cfg.row_cap) {
        const uint2 *rent = buf.row_ent + ((size_t)pair * cfg.height + row) * cfg.row_cap;
        for (int j = gl; j < cnt; j += SM_G) {
            const uint2 e = rent[j];
            consider((int)(e.x & 0xffffu), (int)(e.x >> 16), __uint_as_float(e.y));
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
            const uint8_t *imL = buf.pyr + (size_t)imgL * cfg.pyr_bytes + L.pyr_off;
            const uint8_t *imR = buf.pyr + (size_t)imgR * cfg.pyr_bytes + L.pyr_off;
            const int lc = imL[__mul24(cv, L.pitch) + cu];
            // centre row of the right image: bytes cr-5 .. cr+5 (+1 spare) = rc of the 11 shifts
            uint32_t rcw[3];
            __builtin_memcpy(rcw, imR + __mul24(cv, L.pitch) + cr - 5, 12);
            int dists[11];
#pragma unroll
            for (int t = 0; t < 11; t++) dists[t] = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const int p = gl + SM_G * t;
                if (p < 121) {
                    const int py = (p * 745) >> 13, dy = py - 5, dx = p - py * 11 - 5; // p / 11 for p < 128
                    const int a = (int)imL[__mul24(cv + dy, L.pitch) + cu + dx] - lc;
                    uint32_t w[3];
                    __builtin_memcpy(w, imR + __mul24(cv + dy, L.pitch) + cr + dx - 5, 12);
#pragma unroll
                    for (int i = 0; i < 11; i++) {
                        const int rb = (int)((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
                        const int rc = (int)((rcw[i >> 2] >> (8 * (i & 3))) & 0xffu);
                        dists[i] += abs(a - (rb - rc));
                    }
                }
            }
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
// found by a 3-level radix select (8 bits per level, LDS histograms) instead of a sort.
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
    for (int shift = 16; shift >= 0; shift -= 8) { // SAD < 2^24 (121 px * 510)
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
            if (shift == 16 && lane == 0) s_sel[2] = total;
            // vDistIdx[size/2] in ascending order; later digits continue with the rank left inside the chosen bucket
            const int rank = shift == 16 ? total / 2 : s_sel[1];
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
__global__ __launch_bounds__(256) void rgbd_kernel(DeviceConfig cfg, DeviceBuffers buf, const float *__restrict__ depth_img,
                                                   size_t pitch_floats, int img)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = buf.kp_cnt[img];
    if (i >= n) return;
    const KeyPointPOD kp = ((const KeyPointPOD *)buf.kps)[(size_t)img * cfg.sel_total + i];
    float u = -1.0f, dp = -1.0f;
    const int v = (int)kp.y, uu = (int)kp.x;
    if (uu >= 0 && v >= 0 && uu < cfg.width && v < cfg.height) {
        const float d = depth_img[(size_t)v * pitch_floats + uu];
        if (d > 0) { dp = d; u = __fsub_rn(kp.x, __fdiv_rn(cfg.bf, d)); }
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

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static inline int max_cell_w(const DeviceConfig &cfg) { int m = 0; for (int l = 0; l < cfg.nlevels; l++) m = cfg.lv[l].w_cell > m ? cfg.lv[l].w_cell : m; return m; }
static inline int max_cell_h(const DeviceConfig &cfg) { int m = 0; for (int l = 0; l < cfg.nlevels; l++) m = cfg.lv[l].h_cell > m ? cfg.lv[l].h_cell : m; return m; }

void orbfe_launch_ingest(const DeviceConfig &cfg, const DeviceBuffers &buf, const uint8_t *d_images, int n_images, hipStream_t s)
{
    const int words = (cfg.lv[0].w + 12 + 3) / 4;
    dim3 grid((words + 63) / 64, (cfg.lv[0].h + 2 * PYR_MY + 3) / 4, n_images);
    hipLaunchKernelGGL(ingest_kernel, grid, dim3(256), 0, s, cfg, buf, d_images);
}

void orbfe_launch_pyramid(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, hipStream_t s)
{
    for (int l = 1; l < cfg.nlevels; l++) {
        const int src_words = (cfg.lv[l - 1].w + 3) / 4; // interior pixels of the source row (4-aligned start)
        const int total_rows = cfg.lv[l].h + 2 * PYR_MY;
        // rows per wave: the largest of 4, 2, 1 whose staged source rows fit 60 KB of LDS (rs_src_rows[i] = source
        // row span of the worst block of 16 / 8 / 4 output rows, from the host's row table)
        const int *span = cfg.lv[l].rs_src_rows;
        if ((size_t)span[0] * src_words * 4 <= 60 * 1024) {
            dim3 grid((total_rows + 15) / 16, n_images);
            hipLaunchKernelGGL(pyr_resize_kernel<4>, grid, dim3(256), (size_t)span[0] * src_words * 4, s, cfg, buf, l, src_words, span[0]);
        } else if ((size_t)span[1] * src_words * 4 <= 60 * 1024) {
            dim3 grid((total_rows + 7) / 8, n_images);
            hipLaunchKernelGGL(pyr_resize_kernel<2>, grid, dim3(256), (size_t)span[1] * src_words * 4, s, cfg, buf, l, src_words, span[1]);
        } else {
            dim3 grid((total_rows + 3) / 4, n_images);
            hipLaunchKernelGGL(pyr_resize_kernel<1>, grid, dim3(256), (size_t)span[2] * src_words * 4, s, cfg, buf, l, src_words, span[2]);
        }
    }
}

void orbfe_launch_blur(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, hipStream_t s)
{
    dim3 grid((cfg.blur_tiles_total + 3) / 4, n_images);
    hipLaunchKernelGGL(blur_kernel, grid, dim3(256), 0, s, cfg, buf);
}

void orbfe_launch_fast(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, bool buckets, hipStream_t s)
{
    const int mw = max_cell_w(cfg), mh = max_cell_h(cfg);
    const int tile_pitch = (mw + 6 + 3 + 3 + 3) & ~3; // + alignment slack on both sides
    const int tile_rows = mh + 6;
    const int tile_bytes = (tile_pitch * tile_rows + 15) & ~15;
    const int sc_bytes = ((mw + 2) * (mh + 2) + 15) & ~15;
    const int q_bytes = (2 * mw * mh + 15) & ~15;
    // flags alias the tile: it must hold one byte per interior pixel
    // flags alias the tile region: tile_bytes passed to the kernel covers both; + 2 x 64 words of bucket accumulators
    const int tile_region = tile_bytes > mw * mh ? tile_bytes : ((mw * mh + 15) & ~15);
    const int lds_per_wave = tile_region + sc_bytes + q_bytes + 512;
    const size_t lds = (size_t)4 * lds_per_wave;
    dim3 grid(((cfg.cells_total + 3) / 4) * ((n_images + 7) / 8) * 8);
    static const int dbg = getenv("ORBFE_FAST_DBG") ? atoi(getenv("ORBFE_FAST_DBG")) : 0; // profiling aid only
#define FAST_LAUNCH(TP)                                                                                                                   \
    do {                                                                                                                              \
        if (buckets) hipLaunchKernelGGL((fast_cell_kernel<TP, true>), grid, dim3(256), lds, s, cfg, buf, n_images, tile_pitch, tile_region, sc_bytes, q_bytes, lds_per_wave, dbg); \
        else hipLaunchKernelGGL((fast_cell_kernel<TP, false>), grid, dim3(256), lds, s, cfg, buf, n_images, tile_pitch, tile_region, sc_bytes, q_bytes, lds_per_wave, dbg); \
    } while (0)
    switch (tile_pitch) {
    case 44: FAST_LAUNCH(44); break;
    case 48: FAST_LAUNCH(48); break;
    case 52: FAST_LAUNCH(52); break;
    case 56: FAST_LAUNCH(56); break;
    case 60: FAST_LAUNCH(60); break;
    case 64: FAST_LAUNCH(64); break;
    default: FAST_LAUNCH(0); break;
    }
#undef FAST_LAUNCH
}

static inline int ot_sort_cap(const DeviceConfig &cfg) { int p = 1; while (p < cfg.max_nodes) p <<= 1; return p; }

size_t orbfe_octree_lds_bytes(const DeviceConfig &cfg)
{
    const int cap = cfg.max_nodes;
    const size_t node = 2 * sizeof(int) * cap + 4 * sizeof(short) * cap + ((cap + 7) / 8) * 8;
    return sizeof(unsigned long long) * ot_sort_cap(cfg) + 2 * node + sizeof(int) * 4 * cap + 4 * sizeof(int) * cap;
}

void orbfe_launch_octree_generic(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, hipStream_t s)
{
    dim3 grid(cfg.nlevels, n_images);
    hipLaunchKernelGGL(octree_generic_kernel, grid, dim3(OT_THREADS), orbfe_octree_lds_bytes(cfg), s, cfg, buf, ot_sort_cap(cfg));
}

void orbfe_launch_describe(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_images, bool stereo, hipStream_t s)
{
    dim3 grid(((cfg.sel_total + 4 * DS_KPW - 1) / (4 * DS_KPW)) * ((n_images + 7) / 8) * 8);
    const size_t lds = ((cfg.patch_n * 2 + 15) & ~15) + 256 * 4 + (size_t)4 * (2 * cfg.half_patch + 1 + 37) * DS_PATCH_W;
    static const int dbg = getenv("ORBFE_DESC_DBG") ? atoi(getenv("ORBFE_DESC_DBG")) : 0; // profiling aid only
    hipLaunchKernelGGL(describe_kernel, grid, dim3(256), lds, s, cfg, buf, n_images, stereo ? 1 : 0, dbg);
}

void orbfe_launch_stereo_match(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_pairs, hipStream_t s)
{
    const int kpb = 256 / SM_G;
    dim3 grid(((cfg.sel_total + kpb - 1) / kpb) * ((n_pairs + 7) / 8) * 8);
    hipLaunchKernelGGL(stereo_match_kernel, grid, dim3(256), 0, s, cfg, buf, n_pairs);
}
void orbfe_launch_stereo_median(const DeviceConfig &cfg, const DeviceBuffers &buf, int n_pairs, hipStream_t s)
{
    hipLaunchKernelGGL(stereo_median_kernel, dim3(n_pairs), dim3(256), (size_t)cfg.sel_total * sizeof(int), s, cfg, buf);
}

void orbfe_launch_rgbd(const DeviceConfig &cfg, const DeviceBuffers &buf, const float *d_depth, size_t depth_pitch_floats,
                       int image, hipStream_t s)
{
    hipLaunchKernelGGL(rgbd_kernel, dim3((cfg.sel_total + 255) / 256), dim3(256), 0, s, cfg, buf, d_depth, depth_pitch_floats, image);
}

void orbfe_launch_hamming_matrix(const uint8_t *da, int na, const uint8_t *db, int nb, int *dist, hipStream_t s)
{
    dim3 grid((nb + 63) / 64, (na + 63) / 64);
    hipLaunchKernelGGL(hamming_matrix_kernel, grid, dim3(256), 0, s, da, na, db, nb, dist);
}
