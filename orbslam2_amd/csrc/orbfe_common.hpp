// orbfe_common.hpp -- device helpers shared by the kernel translation units of liborbfe.so (gfx950, wave64).
//
// Floating point follows contract Q4: compiled with -ffp-contract=off, every float product / sum is
// individually rounded (IEEE), divisions are correctly rounded, cos / sin come from the deterministic routine below.
#pragma once

#include "orbfe_device.h"
#include <cstdlib>

typedef short pk16 __attribute__((ext_vector_type(2))); // two int16 lanes in one VGPR (v_pk_* ops)
#define PYR_MX 4 // reflect-101 margin of every pyramid level: pixels left of column 0 ...
#define PYR_MY 3 // ... and rows above row 0 / below the last row
#define ORBFE_RSRC_FLAGS 0x00020000 // word 3 of a gfx9 raw buffer descriptor (__builtin_amdgcn_make_buffer_rsrc): 32-bit data format, no swizzle, no stride

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

// XCD-aware block -> (unit, block-of-unit) map shared by the kernels whose workgroups of one image (or pair) should share
// an L2: workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8; placement is a speed assumption
// only).  With 5 or more units in flight every unit gets one XCD; with fewer, a unit is spread over 2 / 4 / 8 XCDs so
// that a single frame still uses the whole chip (a lone image on one XCD would leave 7/8 of the CUs idle).
__host__ __device__ __forceinline__ int xcd_split_log2(int n_units) { return n_units >= 5 ? 0 : n_units >= 3 ? 1 : n_units == 2 ? 2 : 3; }
__host__ __forceinline__ int xcd_grid(int blocks_per_unit, int n_units)
{
    const int lg = xcd_split_log2(n_units), per_xcd = (blocks_per_unit + (1 << lg) - 1) >> lg, side = 8 >> lg;
    return per_xcd * ((n_units + side - 1) / side) * 8;
}
// a / b for 0 <= a < 2^21, b >= 1: exact through the 1-ulp reciprocal (the quotient's distance to the next integer, 0.5 / b,
// exceeds (a / b) * 2^-22); six VALU operations instead of the compiler's integer-division sequence
__device__ __forceinline__ int small_div(int a, int b) { return (int)(((float)a + 0.5f) * __builtin_amdgcn_rcpf((float)b)); }
__device__ __forceinline__ bool xcd_map_of(int bid, int blocks_per_unit, int n_units, int &unit, int &blk); // the same for workgroup `bid` of a grid whose first (multiple of 8) workgroups do something else
__device__ __forceinline__ bool xcd_map(int blocks_per_unit, int n_units, int &unit, int &blk) { return xcd_map_of((int)blockIdx.x, blocks_per_unit, n_units, unit, blk); }
__device__ __forceinline__ bool xcd_map_of(int bid, int blocks_per_unit, int n_units, int &unit, int &blk)
{
    const int lg = xcd_split_log2(n_units), xcd = bid & 7, jb = bid >> 3;
    const int per_xcd = (blocks_per_unit + (1 << lg) - 1) >> lg;
    const int round = small_div(jb, per_xcd); // jb < 2^21: at most 2^24 blocks per launch (orbfe_create refuses blocks x images >= 2^23)
    unit = round * (8 >> lg) + (xcd >> lg);
    blk = ((jb - round * per_xcd) << lg) + (xcd & ((1 << lg) - 1));
    return unit < n_units && blk < blocks_per_unit;
}

// The same map with the division done in SCALAR arithmetic: magic = ceil(2^32 / per_xcd) from the host (xcd_map_magic_host; 0 = no magic:
// per_xcd == 1 or (grid / 8) * per_xcd >= 2^32), so round = (jb * magic) >> 32 exactly.  small_div costs every wave seven vector
// instructions, a quarter-rate v_rcp_f32 among them, and a v_readfirstlane; fast_cell_kernel's 180 k waves per launch pay for them at its issue rate.
__host__ __forceinline__ uint32_t xcd_map_magic_host(int blocks_per_unit, int n_units)
{
    const int lg = xcd_split_log2(n_units), per_xcd = (blocks_per_unit + (1 << lg) - 1) >> lg, side = 8 >> lg;
    const unsigned long long rounds = (unsigned long long)((n_units + side - 1) / side); // jb < per_xcd * rounds
    if (per_xcd <= 1 || (unsigned long long)per_xcd * rounds * (unsigned long long)per_xcd >= (1ull << 32)) return 0u;
    return (uint32_t)(((1ull << 32) + (unsigned long long)per_xcd - 1ull) / (unsigned long long)per_xcd);
}
__device__ __forceinline__ bool xcd_map_of_magic(int bid, int blocks_per_unit, int n_units, uint32_t magic, int &unit, int &blk)
{
    if (magic == 0u) return xcd_map_of(bid, blocks_per_unit, n_units, unit, blk);
    const int lg = xcd_split_log2(n_units), xcd = bid & 7, jb = bid >> 3;
    const int per_xcd = (blocks_per_unit + (1 << lg) - 1) >> lg;
    const int round = (int)__builtin_amdgcn_readfirstlane((int)(((unsigned long long)(unsigned)jb * (unsigned long long)magic) >> 32)); // uniform: s_mul_hi_u32
    unit = round * (8 >> lg) + (xcd >> lg);
    blk = ((jb - round * per_xcd) << lg) + (xcd & ((1 << lg) - 1));
    return unit < n_units && blk < blocks_per_unit;
}

__device__ __forceinline__ bool xcd_map_magic(int blocks_per_unit, int n_units, uint32_t magic, int &unit, int &blk) { return xcd_map_of_magic((int)blockIdx.x, blocks_per_unit, n_units, magic, unit, blk); }

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
__device__ inline int block_excl_scan(const int *src, int *dst, int n, int *s_tmp)
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

// Raw pyramid level `level` of image `img`: first pixel and row pitch.  Level 0 goes through DeviceBuffers::lv0 (the library's
// pitched copy, or the caller's packed images read in place); `level` is wave-uniform at every call site but the stereo
// refinement's, where it is a select per lane.
__device__ __forceinline__ const uint8_t *level_image(const DeviceConfig &cfg, const DeviceBuffers &buf, int img, int level, int &pitch)
{
    if (level == 0) { pitch = buf.lv0_pitch; return buf.lv0 + (size_t)img * buf.lv0_stride; }
    const LevelInfo &L = cfg.lv[level];
    pitch = L.pitch;
    return buf.pyr + (size_t)img * cfg.pyr_bytes + L.pyr_off;
}

// 16 bytes from an address that is only 4-byte aligned: one global_load_dwordx4 (fine on this memory system).  A plain struct
// load rather than __builtin_memcpy into an array element, which demotes the array to scratch memory.
struct __attribute__((packed, aligned(4))) orbfe_u4_unaligned { uint32_t x, y, z, w; };
__device__ __forceinline__ uint4 load16_unaligned(const uint8_t *p)
{
    const orbfe_u4_unaligned t = *(const orbfe_u4_unaligned *)p;
    return make_uint4(t.x, t.y, t.z, t.w);
}

// 16 bytes from ANY byte address (still one global_load_dwordx4: the target runs in unaligned access mode)
struct __attribute__((packed, aligned(1))) orbfe_u4_any { uint32_t x, y, z, w; };
__device__ __forceinline__ uint4 load16_any(const uint8_t *p)
{
    const orbfe_u4_any t = *(const orbfe_u4_any *)p;
    return make_uint4(t.x, t.y, t.z, t.w);
}

// 12 bytes from any address: one global_load_dwordx3
struct __attribute__((packed, aligned(4))) orbfe_u3_unaligned { uint32_t x, y, z; };

// Wave-uniform reads of host-built tables (never written by a kernel) through the constant address space: only so does the compiler
// keep them scalar loads whatever global stores the kernel holds elsewhere (a blur branch beside FAST's cells turned the cell table's
// s_load into a vector load + v_readfirstlane: + 7 us on the launch).
typedef uint32_t orbfe_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t const_load_u32(const uint32_t *p) { return *(const __attribute__((address_space(4))) uint32_t *)(uintptr_t)p; }
__device__ __forceinline__ uint4 const_load_u32x4(const uint4 *p)
{
    const orbfe_u32x4 v = *(const __attribute__((address_space(4))) orbfe_u32x4 *)(uintptr_t)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
