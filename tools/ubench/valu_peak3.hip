// valu_peak3.hip -- round 5 extension (SDWA / 16-bit forms: can FAST's byte extraction ride in the min / max operands?) of valu_peak2.hip, itself a round 3 extension of valu_peak.hip: issue cost of the VALU opcodes that the front-end's kernels contain
// (tools/isa_mix.py lists them from the disassembly) and that profiles/r02_valu_peak.json had not measured: logic / shift
// forms with three operands, carry adds, compares (VCC and SGPR destinations), lane reads / writes, conversions, 16-bit and
// 64-bit forms, DPP moves.  Same method: long streams of independent inline-asm instructions, 8 accumulators, every CU loaded
// with 4 or 8 waves per SIMD, rate from the chip-wide span of s_memrealtime, clock from s_memtime / s_memrealtime.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_peak3 tools/ubench/valu_peak3.hip && /tmp/valu_peak3 > profiles/r05_valu_peak_sdwa.json
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define UNROLL 16
// kind: 0 = 32-bit accumulator "+v"(a), inputs b, c;  1 = the same, clobbers vcc;  2 = 64-bit accumulator "+v"(d), inputs db, dc;
//       3 = reads the accumulator, writes s20 (or s[20:21]) -- no VGPR result
#define OPS(X)                                                                                   \
    X(v_add_u32, "v_add_u32 %0, %0, %1", 0)                     /* calibration: full rate */ \
    X(v_pk_max_i16, "v_pk_max_i16 %0, %0, %1", 0)               /* calibration: half rate */ \
    X(v_max_u16, "v_max_u16 %0, %0, %1", 0)                                                       \
    X(v_min_u16, "v_min_u16 %0, %0, %1", 0)                                                       \
    X(v_max_i16, "v_max_i16 %0, %0, %1", 0)                                                       \
    X(v_add_u16, "v_add_u16 %0, %0, %1", 0)                                                       \
    X(v_sub_u16, "v_sub_u16 %0, %0, %1", 0)                                                       \
    X(v_max_u16_sdwa_b, "v_max_u16_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2", 0) \
    X(v_min_u16_sdwa_b, "v_min_u16_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_3", 0) \
    X(v_max_u16_sdwa_w1p, "v_max_u16_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_2", 0) \
    X(v_min_u16_sdwa_w0p, "v_min_u16_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_0", 0) \
    X(v_max_u16_sdwa_acc, "v_max_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_2", 0) \
    X(v_sub_u16_sdwa_b, "v_sub_u16_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2", 0) \
    X(v_add_u32_sdwa_b, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2", 0) \
    X(v_and_b32_sdwa_b, "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2", 0) \
    X(v_mov_b32_sdwa_b, "v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2", 0) \
    X(v_cmp_gt_u16_sdwa_s, "v_cmp_gt_u16_sdwa s[20:21], %0, %1 src0_sel:BYTE_0 src1_sel:BYTE_1", 3) \
    X(v_cmp_gt_u16_sdwa_vcc, "v_cmp_gt_u16_sdwa vcc, %0, %1 src0_sel:BYTE_0 src1_sel:BYTE_1", 1) \
    X(v_and_b32, "v_and_b32 %0, %0, %1", 0)                                                       \
    X(v_bitop3_b32, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96", 0)                                 \
    X(v_alignbyte_b32, "v_alignbyte_b32 %0, %0, %1, 3", 0)                                        \
    X(v_perm_b32, "v_perm_b32 %0, %0, %1, %2", 0)                                                 \
    X(v_pk_maximum3_f16, "v_pk_maximum3_f16 %0, %0, %1, %2", 0)                                   \
    X(v_sad_u8, "v_sad_u8 %0, %1, %2, %0", 0)                                                     \
    X(v_msad_u8, "v_msad_u8 %0, %1, %2, %0", 0)                                                   \
    X(v_lerp_u8, "v_lerp_u8 %0, %0, %1, %2", 0)

enum Op {
#define X(n, s, k) OP_##n,
    OPS(X)
#undef X
    NUM_OPS
};
static const char *k_names[NUM_OPS] = {
#define X(n, s, k) #n,
    OPS(X)
#undef X
};

template <int OP>
__global__ __launch_bounds__(256) void issue_kernel(unsigned *sink, long long *stamps, int iters, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 11u, a5 = a0 + 13u, a6 = a0 + 17u, a7 = a0 + 19u;
    unsigned b = seed | 0x01010101u, c = 0x07060504u ^ (seed << 8);
    double d0 = (double)a0, d1 = (double)a1, d2 = (double)a2, d3 = (double)a3, d4 = (double)a4, d5 = (double)a5, d6 = (double)a6, d7 = (double)a7;
    double db = 1.0000001, dc = 1e-9;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#define K0(s, a) asm volatile(s : "+v"(a) : "v"(b), "v"(c));
#define K1(s, a) asm volatile(s : "+v"(a) : "v"(b), "v"(c) : "vcc");
#define K2(s, d) asm volatile(s : "+v"(d) : "v"(db), "v"(dc));
#define K3(s, a) asm volatile(s : : "v"(a), "v"(b), "v"(c) : "s20", "s21");
#define K4(s, d) asm volatile(s : "+v"(d) : "v"(b), "v"(c) : "vcc");
#define EIGHT_A(K, s) K(s, a0) K(s, a1) K(s, a2) K(s, a3) K(s, a4) K(s, a5) K(s, a6) K(s, a7)
#define EIGHT_D(K, s) K(s, d0) K(s, d1) K(s, d2) K(s, d3) K(s, d4) K(s, d5) K(s, d6) K(s, d7)
#define X(n, s, k)                                                      \
    if (OP == OP_##n) {                                                  \
        if (k == 0) { EIGHT_A(K0, s) }                                   \
        else if (k == 1) { EIGHT_A(K1, s) }                              \
        else if (k == 2) { EIGHT_D(K2, s) }                              \
        else if (k == 3) { EIGHT_A(K3, s) }                              \
        else { EIGHT_D(K4, s) }                                          \
    }
            OPS(X)
#undef X
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    sink[blockIdx.x * 256 + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[4 * w + 0] = t0; stamps[4 * w + 1] = t1; stamps[4 * w + 2] = r0; stamps[4 * w + 3] = r1;
    }
}

struct Result { double cyc_per_inst, clock_mhz, ginst_per_s; };

template <int OP>
static Result run(int waves_per_simd, unsigned *d_sink, long long *d_stamps, int n_cu)
{
    const int blocks = n_cu * waves_per_simd, iters = 512;
    std::vector<long long> st((size_t)blocks * 16);
    Result best = {1e30, 0, 0};
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(issue_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_sink, d_stamps, iters, 3u + rep);
        hipDeviceSynchronize();
        hipMemcpy(st.data(), d_stamps, sizeof(long long) * 16 * blocks, hipMemcpyDeviceToHost);
        long long r_first = st[2], r_last = st[3];
        std::vector<double> clk;
        for (int w = 0; w < 4 * blocks; w++) {
            r_first = std::min(r_first, st[4 * w + 2]); r_last = std::max(r_last, st[4 * w + 3]);
            const double dcy = (double)(st[4 * w + 1] - st[4 * w + 0]), dr = (double)(st[4 * w + 3] - st[4 * w + 2]);
            if (dr > 0) clk.push_back(dcy / dr * 100.0);
        }
        std::sort(clk.begin(), clk.end());
        const double med_clk = clk.empty() ? 0.0 : clk[clk.size() / 2];
        const double rate = (double)iters * UNROLL * 8 * 4.0 * blocks / ((double)(r_last - r_first) * 1e-8);
        const double cpi = (double)n_cu * 4.0 * med_clk * 1e6 / rate;
        if (rep > 0 && cpi < best.cyc_per_inst) best = {cpi, med_clk, rate / 1e9};
    }
    return best;
}

template <int OP>
static void report_from(unsigned *d_sink, long long *d_stamps, int n_cu)
{
    if constexpr (OP < NUM_OPS) {
        printf("  \"%s\": {", k_names[OP]);
        const int occ[2] = {4, 8};
        for (int i = 0; i < 2; i++) {
            const Result r = run<OP>(occ[i], d_sink, d_stamps, n_cu);
            printf("\"%d\": {\"cycles_per_wave_inst_per_simd\": %.3f, \"clock_mhz\": %.0f, \"chip_G_wave_inst_per_s\": %.1f}%s", occ[i], r.cyc_per_inst, r.clock_mhz,
                   r.ginst_per_s, i < 1 ? ", " : "");
        }
        printf("}%s\n", OP + 1 < NUM_OPS ? "," : "");
        fflush(stdout);
        report_from<OP + 1>(d_sink, d_stamps, n_cu);
    }
}

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int n_cu = prop.multiProcessorCount;
    unsigned *d_sink; long long *d_stamps;
    hipMalloc(&d_sink, (size_t)n_cu * 8 * 256 * 4);
    hipMalloc(&d_stamps, (size_t)n_cu * 8 * 16 * sizeof(long long));
    printf("{\n \"device\": \"%s\", \"compute_units\": %d, \"method\": \"as profiles/r02_valu_peak.json (tools/ubench/valu_peak.hip): independent inline-asm streams, 8 accumulators, "
           "%d instructions per wave; keys 4 / 8 = waves per SIMD on every CU; not under rocprofv3\",\n \"ops\": {\n", prop.name, n_cu, 512 * UNROLL * 8);
    report_from<0>(d_sink, d_stamps, n_cu);
    printf(" }\n}\n");
    return 0;
}
